"""TSR.forward on B images as one batched pass vs B single-image passes: ms per image (HIP events, median), B = 1, 2, 4, 8.
    python tools/time_batched.py [--prof B]      (--prof B: a few batched passes of B only, for rocprofv3 --kernel-trace)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sculptmate_amd import synth
from sculptmate_amd.tsr import TSR

dev = torch.device("cuda:0")
m = TSR(pos_embed_mode="scale_factor"); m.load_state_dict(synth.tsr_state(seed=0)); m.to(dev)
imgs = [torch.from_numpy(synth.composite_rgb(synth.image_rgba(seed=100 + i))).to(dev) for i in range(8)]


def timed(fn, n=12, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts)), float(np.min(ts))


m.max_batch = 8   # forward() batches only when asked to (default 1 since round 5)
with torch.no_grad():
    if "--prof" in sys.argv:
        B = int(sys.argv[sys.argv.index("--prof") + 1])
        for _ in range(6):
            m(imgs[:B], device=dev)
        torch.cuda.synchronize()
        print("profiled B =", B)
        sys.exit(0)
    for B in (1, 2, 3, 4, 6, 8):
        med, mn = timed(lambda: m(imgs[:B], device=dev))
        m.max_batch = 1
        med1, mn1 = timed(lambda: m(imgs[:B], device=dev))
        m.max_batch = 8
        print("B=%d  batched forward %.3f ms (min %.3f) = %.3f ms/image | one by one %.3f ms = %.3f ms/image" %
              (B, med, mn, med / B, med1, med1 / B), flush=True)
