"""Soak: many image -> mesh steps in one process; memory must stay flat and every mesh must be identical per image."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from sculptmate_amd import synth

dev = torch.device("cuda:0")
model, sd = bench.build_model(dev, seed=0)
imgs = [torch.from_numpy(synth.composite_rgb(synth.image_rgba(seed=100 + i))).to(dev) for i in range(2)]
with torch.no_grad():
    bench.calibrate(model, sd, imgs[0])
    ref = [bench.one_step(model, im) for im in imgs]
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    t0 = time.time()
    for i in range(n):
        v, f = bench.one_step(model, imgs[i % 2])
        if i % 10 == 0:
            assert torch.equal(v, ref[i % 2][0]) and torch.equal(f, ref[i % 2][1]), "result changed at step %d" % i
    torch.cuda.synchronize()
    dt = time.time() - t0
print("soak: %d steps, %.1f meshes/s, memory %.1f -> %.1f MB, peak %.1f MB" % (
    n, n / dt, base / 1e6, torch.cuda.memory_allocated() / 1e6, torch.cuda.max_memory_allocated() / 1e6))
