"""Launch the backbone's self-attention shape a few times (for rocprofv3 --pmc: tools/attn_pmc.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sculptmate_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
SHAPES = ((3072, 3072, 16), (3072, 1025, 16))
if len(sys.argv) > 1: SHAPES = (tuple(int(x) for x in sys.argv[1].split(",")),)
for (Tq, Tk, heads) in SHAPES:
    D = heads * 64
    q = (torch.randn(Tq, D, device=dev) * 0.18).to(BF); k = torch.randn(Tk, D, device=dev).to(BF)
    vt = torch.zeros(D, ((Tk + 63) // 64) * 64, dtype=BF, device=dev); vt[:, :Tk] = torch.randn(D, Tk, device=dev).to(BF)
    o = torch.empty(Tq, D, dtype=BF, device=dev)
    for _ in range(12): ops.attention(q, k, vt, o, Tq, Tk, heads, None)
    torch.cuda.synchronize()
