"""attention, pre-scaled queries: the software-pipelined tile loop (default) against the phase-separated one (SCULPT_ATTN_PIPE=0):
time per call and agreement of each with an fp64 softmax(QK^T)V of the same bf16 operands, and with each other."""
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

SHAPES = ((3072, 3072, 16, 1.0), (3072, 1025, 16, 1.0), (1025, 1025, 12, 1.0), (3072, 3072, 16, 6.0), (3089, 3089, 16, 1.0),
                              (3089, 1297, 16, 1.0), (1297, 1297, 16, 1.0), (3089, 27648, 16, 1.0), (200, 130, 4, 1.0), (64, 64, 1, 1.0), (33, 1, 2, 1.0))
if len(sys.argv) > 1:   # "Tq,Tk,heads": one shape (e.g. under rocprofv3 --kernel-trace --stats: the two kernels' own durations)
    SHAPES = (tuple(int(x) for x in sys.argv[1].split(",")) + (1.0,),)
for (Tq, Tk, heads, gain) in SHAPES:
    D = heads * 64
    g = torch.Generator(device="cpu").manual_seed(Tq + Tk)
    qf = (gain * torch.randn(Tq, D, generator=g)).to(dev); kf = torch.randn(Tk, D, generator=g).to(dev); vf = torch.randn(Tk, D, generator=g).to(dev)
    c = 0.125 * 1.4426950408889634
    qs = (qf * c).to(BF); k = kf.to(BF)
    vt = torch.zeros(D, ((Tk + 63) // 64) * 64, dtype=BF, device=dev); vt[:, :Tk] = vf.to(BF).t()
    o0 = torch.empty(Tq, D, dtype=BF, device=dev); o1 = torch.empty_like(o0)
    os.environ["SCULPT_ATTN_PIPE"] = "0"
    t0 = timeit(lambda: ops.attention(qs, k, vt, o0, Tq, Tk, heads, None))
    os.environ["SCULPT_ATTN_PIPE"] = "1"
    t1 = timeit(lambda: ops.attention(qs, k, vt, o1, Tq, Tk, heads, None))
    qh = qs.double().view(Tq, heads, 64).transpose(0, 1); kh = k.double().view(Tk, heads, 64).transpose(0, 1)
    vh = vt[:, :Tk].t().double().view(Tk, heads, 64).transpose(0, 1)
    r = (torch.softmax(qh @ kh.transpose(1, 2) * math.log(2.0), -1) @ vh).transpose(0, 1).reshape(Tq, D)
    e0 = float((o0.double() - r).norm() / r.norm()); e1 = float((o1.double() - r).norm() / r.norm())
    print("Tq %5d Tk %5d H %2d gain %.0f: phases %6.1f us (rel err %.2e) | pipelined %6.1f us (rel err %.2e)  x%.3f  max |diff| %.3e  finite %s"
          % (Tq, Tk, heads, gain, t0, e0, t1, e1, t0 / t1, float((o0.float() - o1.float()).abs().max()), bool(torch.isfinite(o1.float()).all())), flush=True)
