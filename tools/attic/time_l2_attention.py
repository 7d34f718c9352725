import os, sys, math
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from sculptmate_amd import ops
dev = torch.device("cuda:0"); g = torch.Generator().manual_seed(0)
def timed(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for Tq, Tk, heads in ((1025, 1025, 12), (3072, 3072, 16), (3072, 1025, 16)):
    D = heads * 64
    Q, K = torch.randn(Tq, D, generator=g).to(dev), torch.randn(Tk, D, generator=g).to(dev)
    ldv = ((Tk + 63) // 64) * 64
    Vt = torch.zeros(D, ldv, device=dev); Vt[:, :Tk] = torch.randn(D, Tk, generator=g).to(dev)
    O = ops.Limbs(Tq, D, dev, zero=True, fmt="f16x2")
    r = {}
    for name, pipe, h2 in (("3 limbs default", None, False), ("2 limbs default", None, True), ("2 limbs, pipelined form forced", "1", True)):
        if pipe is None: os.environ.pop("SCULPT_L3_ATTN_PIPE", None)
        else: os.environ["SCULPT_L3_ATTN_PIPE"] = pipe
        r[name] = timed(lambda: ops.attention_f32(Q, K, Vt, O, Tq, Tk, heads, 0.125, None, l3=True, two_fp16_limbs=h2))
    print(Tq, Tk, heads, " | ".join("%s %.1f us" % kv for kv in r.items()), flush=True)
