"""Race screen for the pipelined attention kernel: the same launch repeated many times must give the SAME bits every time (a stale
tile, a slot overwritten early or a missed wait shows up as run-to-run differences), with a second stream hammering the GPU to
perturb the timing.  Shapes around the staging edge cases (ragged last pair, one pair, many pairs, both workgroup sizes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sculptmate_amd import ops

dev = torch.device("cuda:0"); BF = torch.bfloat16
rng = np.random.default_rng(0)
g = torch.Generator().manual_seed(0)
noise_stream = torch.cuda.Stream(dev)
na = torch.randn(2048, 2048, device=dev)
shapes = [(3072, 3072, 16), (3072, 1025, 16), (1025, 1025, 12), (27648 // 4, 3089, 16), (200, 130, 4), (3072, 129, 16), (3072, 128, 16), (3072, 64, 16),
          (1536, 2047, 8), (1000, 385, 16)]
for _ in range(10):
    shapes.append((int(rng.integers(1, 4000)), int(rng.integers(1, 3000)), int(rng.integers(1, 17))))
bad = 0
for (Tq, Tk, heads) in shapes:
    D = heads * 64
    q = (torch.randn(Tq, D, generator=g) * 0.18).to(BF).to(dev); k = torch.randn(Tk, D, generator=g).to(BF).to(dev)
    vt = torch.zeros(D, ((Tk + 63) // 64) * 64, dtype=BF, device=dev); vt[:, :Tk] = torch.randn(D, Tk, generator=g).to(BF).to(dev)
    o = torch.empty(Tq, D, dtype=BF, device=dev)
    ops.attention(q, k, vt, o, Tq, Tk, heads, None)
    torch.cuda.synchronize()
    ref = o.clone()
    ndiff = 0
    for it in range(150):
        if it % 3 == 0:
            with torch.cuda.stream(noise_stream):
                torch.mm(na, na)
        o.fill_(float("nan"))
        ops.attention(q, k, vt, o, Tq, Tk, heads, None)
        if it % 10 == 9:
            torch.cuda.synchronize()
        if not torch.equal(o.view(torch.int16), ref.view(torch.int16)):
            ndiff += 1
    torch.cuda.synchronize()
    bad += ndiff
    print("Tq %5d Tk %5d heads %2d: %d of 150 launches differ from the first" % (Tq, Tk, heads, ndiff), flush=True)
print("TOTAL differing launches:", bad)
sys.exit(1 if bad else 0)
