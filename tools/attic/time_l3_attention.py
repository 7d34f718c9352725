"""Fused three-limb attention (attention_l3.hip): the pipelined 8-wave form against the plain 4-wave form (SCULPT_L3_ATTN_PIPE = 1 / 0),
interleaved rounds in one process, on the transformer's shapes; executed FLOPs = 6 x the algorithmic 4 Tq Tk 64 heads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sculptmate_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for Tq, Tk, heads in ((3072, 3072, 16), (3072, 1025, 16), (1025, 1025, 12)):
    D = heads * 64
    Q = torch.randn(Tq, D, generator=g).to(dev); K = torch.randn(Tk, D, generator=g).to(dev)
    Vt = torch.zeros(D, (Tk + 63) // 64 * 64); Vt[:, :Tk] = torch.randn(D, Tk, generator=g); Vt = Vt.to(dev)
    O = torch.empty(Tq, D, device=dev)
    f = lambda: ops.attention_f32(Q, K, Vt, O, Tq, Tk, heads, 0.125, None, l3=True)
    res, outs = {}, {}
    for rnd in range(4):
        for mode in ("0", "1"):
            os.environ["SCULPT_L3_ATTN_PIPE"] = mode
            for _ in range(2): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(mode, []).append(e0.elapsed_time(e1) / 10 * 1e3)
            outs[mode] = O.clone()
    fl = 6 * 4.0 * Tq * Tk * 64 * heads
    t0, t1 = np.median(res["0"]), np.median(res["1"])
    print("Tq %d Tk %d heads %d: plain 4-wave %.1f us (%.0f TF/s executed) | pipelined 8-wave %.1f us (%.0f TF/s executed, %.3f of the bf16 peak) | identical: %s"
          % (Tq, Tk, heads, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6, fl / t1 / 1e6 / 2500, torch.equal(outs["0"], outs["1"])), flush=True)
os.environ.pop("SCULPT_L3_ATTN_PIPE", None)
