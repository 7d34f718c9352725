"""Does hipGraph capture of the forward + density launches pay? (torch.cuda.CUDAGraph around the ctypes launches)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from sculptmate_amd import ops, synth
dev = torch.device("cuda:0")
model, sd = bench.build_model(dev, 0)
img = torch.from_numpy(synth.composite_rgb(synth.image_rgba(seed=100))).to(dev).contiguous()
with torch.no_grad():
    bench.calibrate(model, sd, img)
    r = model.renderer.cfg.radius
    def fwd():
        ctx, _ = model.image_tokens(img)
        _, outb = model.backbone_tokens(ctx)
        planes = model.scene_code(outb)
        return ops.density_grid(planes, model.decoder, 256, radius=r, density_bias=-1.0, out_add=-25.0)
    for _ in range(3): vol = fwd()
    torch.cuda.synchronize()
    def timeit(fn, n=20):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    print("eager forward+density: %.3f ms" % timeit(fwd))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): fwd()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        vol_g = fwd()
    g.replay(); torch.cuda.synchronize()
    print("graph  forward+density: %.3f ms" % timeit(g.replay))
    print("same result:", bool(torch.equal(vol_g, fwd())))
