"""Where do the ~14 us of the small one-image GEMMs go?  Launch time against K at fixed M x N (slope = the K loop, intercept = what
a launch costs before and after it), HIP events over back-to-back launches on one stream, beside a trivial torch kernel.
    python tools/gemm_k_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sculptmate_amd import ops

dev = torch.device("cuda:0")
BF = torch.bfloat16
g = torch.Generator().manual_seed(0)


def timed(f, n=50, rounds=5):
    f(); torch.cuda.synchronize()
    out = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / n * 1e3)
    return float(np.median(out))


x = torch.zeros(64, device=dev)
print("trivial torch kernel (64-element add_): %.1f us per launch" % timed(lambda: x.add_(1.0)))
for name, M, N, Ks in (("ViT o / f2 shape", 1025, 768, (64, 128, 256, 384, 768, 1536, 3072)),
                       ("ViT qkv shape", 1025, 2304, (64, 128, 256, 768)),
                       ("ViT f1 shape", 1025, 3072, (64, 128, 256, 768)),
                       ("backbone o / q / FF2 shape", 3072, 1024, (64, 128, 256, 512, 1024, 2048, 4096)),
                       ("backbone QKV shape", 3072, 3072, (64, 256, 1024)),
                       ("backbone FF1 shape (plain epilogue)", 3072, 8192, (64, 256, 1024))):
    row = []
    for K in Ks:
        A = torch.randn(M, K, generator=g).to(BF).to(dev)
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        out = torch.empty(M, N, dtype=BF, device=dev)
        row.append((K, timed(lambda: ops.gemm(A, W, bias=bias, out_bf16=out))))
    (k0, t0), (k1, t1) = row[0], row[-1]
    slope = (t1 - t0) / (k1 - k0)
    print("%-38s M=%d N=%d: " % (name, M, N) + "  ".join("K=%d %.1f us" % r for r in row)
          + "  | slope %.2f us per 64 of K, intercept %.1f us" % (slope * 64, t0 - slope * k0), flush=True)
