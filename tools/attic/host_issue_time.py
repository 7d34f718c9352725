import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from sculptmate_amd import synth
dev = torch.device("cuda:0")
model, sd = bench.build_model(dev, 0)
img = torch.from_numpy(synth.composite_rgb(synth.image_rgba(seed=100))).to(dev)
with torch.no_grad():
    for _ in range(3): model.forward(img)
    torch.cuda.synchronize()
    hs, ts = [], []
    for _ in range(10):
        t0 = time.perf_counter(); c = model.forward(img); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        hs.append((t1 - t0) * 1e3); ts.append((t2 - t0) * 1e3)
    print("forward: host issue %.2f ms (min %.2f), issue + wait %.2f ms" % (np.median(hs), min(hs), np.median(ts)))
