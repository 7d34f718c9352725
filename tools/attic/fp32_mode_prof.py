"""Three full-size forwards in TSR(precision="fp32") (the parity mode): run under `rocprofv3 --kernel-trace --stats` to see where its
~160 ms go (measured: gemm_f32_kernel 143 ms incl. the per-head attention products, softmax_rows 16 ms)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sculptmate_amd import synth
from sculptmate_amd.tsr import TSR
dev = torch.device("cuda:0")
sd = synth.tsr_state(seed=0)
m = TSR(pos_embed_mode="scale_factor", precision="fp32"); m.load_state_dict(sd); m.to(dev)
img = synth.composite_rgb(synth.image_rgba(seed=100))
with torch.no_grad():
    for _ in range(3):
        c = m([img], device=dev)
    torch.cuda.synchronize()
print("done")
