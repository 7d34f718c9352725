"""GEMM launch variants A/B on the transformer's shapes, interleaved rounds in ONE process (every switch is read per call):
   tile kernel   SCULPT_GEMM_256 = 0 (128-row tiles) / 2 (256 x 256) / 2 + SCULPT_GEMM_192=1 (192 x 256)
   tile order    SCULPT_GEMM_GM  = 0 (1-D band per XCD) / unset (grouped, auto height) / n
Shapes: the backbone's Linears at B = 1 and as one launch over B = 4 images.  Random operands, the real epilogues.
    python tools/gemm_order_ab.py [quick]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sculptmate_amd import _lib, ops

dev = torch.device("cuda:0")
BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
VARIANTS = [("128/band", "0", "0", "0"), ("128/grouped", "0", "0", None), ("256/band", "2", "0", "0"), ("256/grouped", "2", "0", None),
            ("192/band", "2", "1", "0"), ("192/grouped", "2", "1", None)]


RES_VARIANTS = [("128/band", "0", "0", "0"), ("128/grouped", "0", "0", None), ("192-row residual form", "1", "1", None)]


def setenv(v):
    os.environ["SCULPT_GEMM_256"], os.environ["SCULPT_GEMM_192"] = v[1], v[2]
    os.environ["SCULPT_GEMM_RES256"] = "1" if v[0].startswith("192-row residual") else "0"
    if v[3] is None:
        os.environ.pop("SCULPT_GEMM_GM", None)
    else:
        os.environ["SCULPT_GEMM_GM"] = v[3]


def case(name, M, K, N, epi, split=0, residual=False, ln=True):
    rows = 2 * N if epi == _lib.EPI_GEGLU else N
    A = torch.randn(M, K, generator=g).to(BF).to(dev)
    W = (torch.randn(rows, K, generator=g) / K ** 0.5).to(BF).to(dev)
    bias = torch.randn(rows, generator=g).to(dev)
    kw = {}
    if ln:
        stats = torch.zeros(K // 64, M, 2, device=dev); stats[..., 1] = 64.0
        kw.update(ln_stats=stats, ln_colsum=W.float().sum(1).contiguous(), ln_eps=1e-5)
    Mp = (M + 63) // 64 * 64
    if residual:   # h += A W^T + b: fp32 in / out, bf16 copy, slice statistics
        h = torch.randn(M, N, generator=g).to(dev)
        kw.update(residual=h, out_f32=h, out_bf16=torch.empty(M, N, dtype=BF, device=dev), stats_out=torch.zeros(N // 64, M, 2, device=dev))
        o, ot = kw["out_bf16"], None
    elif split:
        o = torch.empty(M, split, dtype=BF, device=dev); ot = torch.zeros(N - split, Mp, dtype=BF, device=dev)
        kw.update(out_bf16=o, out_t=ot, n_split=split)
    else:
        o = torch.empty(M, N, dtype=BF, device=dev); ot = None
        kw.update(out_bf16=o)
    f = lambda: ops.gemm(A, W, bias=bias, epilogue=epi, **kw)
    res, outs = {}, {}
    variants = RES_VARIANTS if residual else VARIANTS
    for rnd in range(3 if "quick" in sys.argv else 5):
        for v in variants:
            setenv(v)
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(v[0], []).append(e0.elapsed_time(e1) / 20 * 1e3)
            if not residual:
                outs[v[0]] = (o.clone(), None if ot is None else ot.clone())
    fl = 2.0 * M * rows * K
    same = (not residual) and all(torch.equal(outs[variants[0][0]][0], outs[v[0]][0]) and
                                  (ot is None or torch.equal(outs[variants[0][0]][1], outs[v[0]][1])) for v in variants)
    print("%-26s M=%-5d K=%-4d N=%-5d " % (name, M, K, N) +
          " | ".join("%s %.1f us (%.0f TF/s)" % (v[0], np.median(res[v[0]]), fl / np.median(res[v[0]]) / 1e6) for v in variants) +
          ("" if residual else " | identical: %s" % same), flush=True)


for B in (1, 4, 8):
    T = 3072 * B
    case("FF1 + GEGLU  B=%d" % B, T, 1024, 4096, _lib.EPI_GEGLU)
    case("fused Q|K|V^T B=%d" % B, T, 1024, 3072, _lib.EPI_NONE, split=2048)
    case("cross-attn q  B=%d" % B, T, 1024, 1024, _lib.EPI_NONE)
    case("to_out (res)  B=%d" % B, T, 1024, 1024, _lib.EPI_NONE, residual=True, ln=False)
    case("FF2 (res)     B=%d" % B, T, 4096, 1024, _lib.EPI_NONE, residual=True, ln=False)
    case("K/V all layers B=%d" % B, 1025 if B == 1 else 1032 * B, 768, 32768, _lib.EPI_NONE, split=16384, ln=False)
os.environ.pop("SCULPT_GEMM_RES256", None)
