"""Do two independent transformer chains overlap on the GPU (two streams)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from sculptmate_amd import ops, synth
dev = torch.device("cuda:0")
mA, sd = bench.build_model(dev, 0)
mB, _ = bench.build_model(dev, 0)
img = torch.from_numpy(synth.composite_rgb(synth.image_rgba(seed=100))).to(dev).contiguous()
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def fwd(m):
    ctx, _ = m.image_tokens(img); _, outb = m.backbone_tokens(ctx); return m.scene_code(outb)
def dens(m, planes):
    return ops.density_grid(planes, m.decoder, 256, out=m._b("vol", (256 ** 3,), torch.float32))
with torch.no_grad():
    for _ in range(3): fwd(mA); fwd(mB)
    torch.cuda.synchronize()
    def t(fn, n=10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    print("sequential 2x forward: %.2f ms" % t(lambda: (fwd(mA), fwd(mB))))
    def both():
        with torch.cuda.stream(sA): fwd(mA)
        with torch.cuda.stream(sB): fwd(mB)
    print("two streams 2x forward: %.2f ms" % t(both))
    def both_full():
        with torch.cuda.stream(sA): dens(mA, fwd(mA))
        with torch.cuda.stream(sB): dens(mB, fwd(mB))
    def seq_full():
        dens(mA, fwd(mA)); dens(mB, fwd(mB))
    # workspaces are shared between the two density calls here: results unused, timing only
    print("sequential 2x (forward+density): %.2f ms" % t(seq_full))
    print("two streams 2x (forward+density): %.2f ms" % t(both_full))
