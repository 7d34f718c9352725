"""Where does the host-boundary step lose time against the resident-input step?  (bench.py 'boundary' vs 'value')"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from sculptmate_amd import synth

dev = torch.device("cuda:0")
model, sd = bench.build_model(dev, 0)
imgs_np = [synth.composite_rgb(synth.image_rgba(seed=100 + i)) for i in range(4)]
imgs = [torch.from_numpy(a).to(dev) for a in imgs_np]
bench.calibrate(model, sd, imgs[0])
N = 20

def timed(fn, n=N):
    fn(0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

with torch.no_grad():
    print("resident in, mesh stays in HBM        %.3f ms" % timed(lambda i: bench.one_step(model, imgs[i % 4])))
    print("host image in (pageable), mesh in HBM %.3f ms" % timed(lambda i: bench.one_step(model, imgs_np[i % 4])))
    pin = [torch.from_numpy(a).pin_memory() for a in imgs_np]
    print("host image in (pinned), mesh in HBM   %.3f ms" % timed(lambda i: bench.one_step(model, pin[i % 4])))
    def sync_out(i):
        v, f = bench.one_step(model, imgs[i % 4]); v.cpu(); f.cpu()
    print("resident in, mesh .cpu() (pageable)   %.3f ms" % timed(sync_out))
    prev = [None]
    def pipe(i):
        cur = model.run_async(imgs[i % 4], 256, 25.0)
        if prev[0] is not None: prev[0].result()
        prev[0] = cur
    print("resident in, run_async pipelined      %.3f ms" % timed(pipe))
    prev = [None]
    def pipe2(i):
        cur = model.run_async(imgs_np[i % 4], 256, 25.0)
        if prev[0] is not None: prev[0].result()
        prev[0] = cur
    print("host in, run_async pipelined          %.3f ms" % timed(pipe2))
    # how long does the pinned D2H alone take, and does it overlap a density kernel?
    v, f = bench.one_step(model, imgs[0]); torch.cuda.synchronize()
    hv = torch.empty(v.shape, dtype=v.dtype, pin_memory=True); hf = torch.empty(f.shape, dtype=f.dtype, pin_memory=True)
    t0 = time.perf_counter()
    for _ in range(10): hv.copy_(v, non_blocking=True); hf.copy_(f, non_blocking=True)
    torch.cuda.synchronize(); print("pinned D2H of one mesh alone          %.3f ms (%.1f GB/s)" % ((time.perf_counter() - t0) * 100, (v.nbytes + f.nbytes) / ((time.perf_counter() - t0) / 10) / 1e9))
    t0 = time.perf_counter()
    for _ in range(10): hv = torch.empty(v.shape, dtype=v.dtype, pin_memory=True); hf = torch.empty(f.shape, dtype=f.dtype, pin_memory=True)
    print("pinned alloc of both buffers          %.3f ms" % ((time.perf_counter() - t0) * 100))
