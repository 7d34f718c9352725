"""N = 1024 residual GEMM at K = 1024 / 4096 over M: the time staircase of 128 x 64 tiles per CU (1.0 -> 1.25 tiles per CU costs as much as
2.0), i.e. what a perfectly balanced 1.5-tiles-per-CU schedule (mixed tile heights, stream-K) could return on the backbone's M = 3072."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sculptmate_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    ts = []
    for _ in range(n):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 10)
    ts.sort(); return ts[len(ts) // 2] * 1e3
for K in (1024, 4096):
    for M in (1024, 2048, 2560, 3072, 3584, 4096, 6144):
        N = 1024
        A = torch.randn(M, K, device=dev).to(BF); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(BF)
        res = torch.randn(M, N, device=dev); out = torch.empty(M, N, device=dev); outb = torch.empty(M, N, dtype=BF, device=dev)
        bias = torch.randn(N, device=dev)
        us = timeit(lambda: ops.gemm(A, W, bias=bias, residual=res, out_f32=out, out_bf16=outb))
        tiles = (M // 128) * (N // 64)
        print("K %4d M %4d: %6.1f us  tiles(128x64) %4d = %.2f per CU   %.0f TF/s" % (K, M, us, tiles, tiles / 256, 2.0 * M * N * K / us / 1e6), flush=True)
