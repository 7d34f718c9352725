"""Time single bf16 GEMM shapes (HIP events); SCULPT_GEMM_NW8S=0/1 forces the 4- / 8-wave 64-row kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sculptmate_amd import ops

dev = torch.device("cuda:0")
shapes = [(1025, 768, 3072), (3072, 1024, 4096), (1025, 768, 768), (1025, 2304, 768), (1025, 3072, 768), (1297, 1024, 4096)]
for M, N, K in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.float32)
    res = torch.randn(M, N, device=dev)
    bias = torch.randn(N, device=dev)
    f = lambda: ops.gemm(A, W, bias=bias, residual=res, out_f32=out)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 50 * 1e3
    print("M=%d N=%d K=%d: %.1f us (%.0f TFLOP/s)" % (M, N, K, t, 2.0 * M * N * K / t / 1e6))
