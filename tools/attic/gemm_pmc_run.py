"""Launch the transformer's GEMM shapes a few times each (for rocprofv3 --pmc: tools/gemm_pmc.sh)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sculptmate_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
M = 3072
for (N, K, epi) in ((1024, 1024, 0), (3072, 1024, 0), (4096, 1024, 2), (1024, 4096, 0)):
    A = torch.randn(M, K, device=dev).to(BF); W = (torch.randn((2 * N if epi == 2 else N), K, device=dev) / math.sqrt(K)).to(BF)
    out = torch.empty(M, N, dtype=BF, device=dev)
    for _ in range(12): ops.gemm(A, W, out_bf16=out, epilogue=epi)
    torch.cuda.synchronize()
