import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sculptmate_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
M, N, K = 1025, 768, 3072
A = torch.randn(M, K, device=dev).to(BF); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(BF)
out = torch.empty(M, N, dtype=BF, device=dev)
for _ in range(20): ops.gemm(A, W, out_bf16=out)
torch.cuda.synchronize()
