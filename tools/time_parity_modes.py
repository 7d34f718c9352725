"""Full-size TSR.forward in the two parity modes -- precision="fp32" (exact-fp32 matrix instruction) and "bf16l3" (three-limb bf16
split) -- ms per forward (HIP events, median) and the scene code against each other and the bf16 mode.
    python tools/time_parity_modes.py [--prof MODE]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sculptmate_amd import synth
from sculptmate_amd.tsr import TSR

dev = torch.device("cuda:0")
sd = synth.tsr_state(seed=0)
img = torch.from_numpy(synth.composite_rgb(synth.image_rgba(seed=100))).to(dev)
codes = {}
# "bf16l3-split": the same mode with SCULPT_L3_TILE=split, every GEMM splitting its operands while staging them (the form before "limbs once")
modes = [sys.argv[sys.argv.index("--prof") + 1]] if "--prof" in sys.argv else ["fp16l2", "bf16l3", "bf16l3-split", "fp32", "bf16"]
with torch.no_grad():
    for mode in modes:
        os.environ["SCULPT_L3_TILE"] = "split" if mode.endswith("-split") else ""
        m = TSR(pos_embed_mode="scale_factor", precision=mode.split("-")[0]); m.load_state_dict(sd); m.to(dev)
        for _ in range(2):
            c = m([img], device=dev)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3 if "--prof" in sys.argv else 5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); c = m([img], device=dev); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        codes[mode] = c[0].clone()
        print("precision=%-12s forward %.2f ms (min %.2f)" % (mode, float(np.median(ts)), min(ts)), flush=True)
        del m
        torch.cuda.empty_cache()
rel = lambda a, b: float((a - b).norm() / b.norm())
if "bf16l3" in codes and "bf16l3-split" in codes:
    print("scene code, limbs once vs split in every GEMM: identical %s" % torch.equal(codes["bf16l3"], codes["bf16l3-split"]))
if "fp32" in codes and "fp16l2" in codes:
    print("scene code: fp16l2 vs exact fp32 rel %.3e" % rel(codes["fp16l2"], codes["fp32"]))
if "fp32" in codes and "bf16l3" in codes:
    print("scene code: bf16l3 vs exact fp32 rel %.3e; bf16 vs exact fp32 rel %.3e" % (rel(codes["bf16l3"], codes["fp32"]), rel(codes["bf16"], codes["fp32"])))
