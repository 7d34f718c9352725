"""GPU-side duration (rocprofv3 --kernel-trace, host overhead excluded) of one GEMM shape over K: what is fixed per launch and what
a K-tile costs.  Run under:  rocprofv3 --kernel-trace --stats -d OUT -- python3 tools/gemm_ksweep.py [residual]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sculptmate_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
M, N = 3072, 1024
res_mode = len(sys.argv) > 1 and sys.argv[1] == "residual"
if len(sys.argv) > 1 and sys.argv[1] in ("g256", "g256full"):
    # the 256 x 256 tile kernel: a 48-tile launch (every workgroup has a CU to itself) or FF1's shape with every CU on it
    os.environ["SCULPT_GEMM_256"], os.environ["SCULPT_GEMM_192"] = "2", "0"
    M, N = (768, 4096) if sys.argv[1] == "g256" else (3072, 8192)
g = torch.Generator().manual_seed(0)
for K in (64, 128, 256, 512, 1024, 2048, 4096):
    A = torch.randn(M, K, generator=g).to(BF).to(dev); W = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF).to(dev)
    o32 = torch.empty(M, N, device=dev); o16 = torch.empty(M, N, dtype=BF, device=dev); res = torch.randn(M, N, device=dev)
    stats = torch.zeros(N // 64, M, 2, device=dev); bias = torch.randn(N, device=dev)
    for _ in range(12):
        if res_mode: ops.gemm(A, W, bias=bias, residual=res, out_f32=o32, out_bf16=o16, stats_out=stats)
        else: ops.gemm(A, W, out_bf16=o16)
    torch.cuda.synchronize()
    # marker: a distinct tiny kernel between K values (a fill of K elements) so that the trace can be split
    torch.zeros(K, device=dev)
torch.cuda.synchronize()
