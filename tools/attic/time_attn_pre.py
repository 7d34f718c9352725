"""attention_kernel: scores scaled in the kernel (scale > 0) vs Q pre-scaled by scale*log2(e) (scale = 0): time and agreement."""
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

for (Tq, Tk, heads, gain) in ((3072, 3072, 16, 1.0), (3072, 1025, 16, 1.0), (1025, 1025, 12, 1.0), (3072, 3072, 16, 6.0), (3089, 27648, 16, 1.0), (27648, 3089, 16, 1.0), (3089, 3089, 16, 1.0), (3089, 1297, 16, 1.0), (1297, 1297, 16, 1.0)):
    D = heads * 64
    g = torch.Generator(device="cpu").manual_seed(Tq + Tk)
    qf = (gain * torch.randn(Tq, D, generator=g)).to(dev); kf = torch.randn(Tk, D, generator=g).to(dev); vf = torch.randn(Tk, D, generator=g).to(dev)
    c = 0.125 * 1.4426950408889634
    q = qf.to(BF); qs = (qf * c).to(BF); k = kf.to(BF)
    vt = torch.zeros(D, ((Tk + 63) // 64) * 64, dtype=BF, device=dev); vt[:, :Tk] = vf.to(BF).t()
    o0 = torch.empty(Tq, D, dtype=BF, device=dev); o1 = torch.empty_like(o0)
    t0 = timeit(lambda: ops.attention(q, k, vt, o0, Tq, Tk, heads, 0.125))
    t1 = timeit(lambda: ops.attention(qs, k, vt, o1, Tq, Tk, heads, None))
    # fp64 reference from the bf16 operands each kernel really gets
    def ref(qq, scale):
        qh = qq.double().view(Tq, heads, 64).transpose(0, 1); kh = k.double().view(Tk, heads, 64).transpose(0, 1)
        vh = vt[:, :Tk].t().double().view(Tk, heads, 64).transpose(0, 1)
        return (torch.softmax(qh @ kh.transpose(1, 2) * scale, -1) @ vh).transpose(0, 1).reshape(Tq, D)
    r0 = ref(q, 0.125); r1 = ref(qs, math.log(2.0))
    e0 = float((o0.double() - r0).norm() / r0.norm()); e1 = float((o1.double() - r1).norm() / r1.norm())
    print("Tq %5d Tk %5d H %2d gain %.0f: in-kernel scale %6.1f us (rel err %.2e) | pre-scaled Q %6.1f us (rel err %.2e)  x%.3f"
          % (Tq, Tk, heads, gain, t0, e0, t1, e1, t0 / t1))
