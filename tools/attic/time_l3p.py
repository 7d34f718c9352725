"""The Linears of one forward in the tolerance mode: the three-limb GEMM that splits while staging (gemm_l3.hip) against the one
whose operands arrive split (gemm_l3p.hip); interleaved rounds, HIP events, outputs compared bit for bit.
    python tools/time_l3p.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sculptmate_amd import _lib, ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def timed(f, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot_old = tot_new = 0.0
for name, M, N, K, epi, per_fwd in (("backbone QKV", 3072, 3072, 1024, 0, 16), ("backbone o / q", 3072, 1024, 1024, 0, 48),
                                    ("backbone FF1 GEGLU", 3072, 4096, 1024, _lib.EPI_GEGLU, 16), ("backbone FF2", 3072, 1024, 4096, 0, 16),
                                    ("ViT qkv", 1025, 2304, 768, 0, 12), ("ViT o", 1025, 768, 768, 0, 12),
                                    ("ViT f1 GELU", 1025, 3072, 768, _lib.EPI_GELU, 12), ("ViT f2", 1025, 768, 3072, 0, 12)):
    rows = 2 * N if epi == _lib.EPI_GEGLU else N
    A = torch.randn(M, K, generator=g).to(dev); W = (torch.randn(rows, K, generator=g) / K ** 0.5).to(dev)
    bias = torch.randn(rows, generator=g).to(dev)
    o1, o2 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    A_lt = ops.limbs_split(A)
    W_lt = ops.limbs_split(ops.geglu_row_blocks(W) if epi == _lib.EPI_GEGLU else W)
    f_old = lambda: ops.gemm_f32(A, W, bias=bias, out=o1, epilogue=epi, l3=True)
    res = {}
    variants = {"old": f_old}
    for bm in ("0", "1"):
        def f_new(bm=bm):
            os.environ["SCULPT_L3P_BM64"] = bm; os.environ["SCULPT_L3P_NW8"] = "0"
            ops.gemm_l3p(A_lt, W_lt, M, N, K, bias=bias, out=o2, epilogue=epi)
        variants["new bm%s" % ("64" if bm == "1" else "128")] = f_new
    def f_nw8():
        os.environ["SCULPT_L3P_BM64"] = "0"; os.environ["SCULPT_L3P_NW8"] = "1"
        ops.gemm_l3p(A_lt, W_lt, M, N, K, bias=bias, out=o2, epilogue=epi)
        os.environ.pop("SCULPT_L3P_NW8", None)
    variants["new bm128 8 waves"] = f_nw8
    out_lt = ops.limbs_empty(M, N, dev)
    def f_lt():
        os.environ.pop("SCULPT_L3P_BM64", None); os.environ.pop("SCULPT_L3P_NW8", None)
        ops.gemm_l3p(A_lt, W_lt, M, N, K, bias=bias, out_lt=out_lt, epilogue=epi)
    variants["new, limb output"] = f_lt
    variants["split of A alone"] = lambda: ops.limbs_split(A, out=A_lt.data if hasattr(A_lt, "data") else A_lt)
    A_h = ops.Limbs.of(A, fmt="f16x2")
    W_h = ops.Limbs.of(ops.geglu_row_blocks(W) if epi == _lib.EPI_GEGLU else W, fmt="f16x2", weight=True)
    o3 = torch.empty(M, N, device=dev)
    def f_h():
        os.environ.pop("SCULPT_L3P_BM64", None); os.environ.pop("SCULPT_L3P_NW8", None)
        ops.gemm_l3p(A_h, W_h, M, N, K, bias=bias, out=o3, epilogue=epi)
    variants["two fp16 limbs"] = f_h
    for f in variants.values(): f()
    torch.cuda.synchronize()
    same = torch.equal(o1, o2)
    for rnd in range(5):
        for k, f in variants.items():
            res.setdefault(k, []).append(timed(f))
    med = {k: float(np.median(v)) for k, v in res.items()}
    fl = 2.0 * M * rows * K * 6
    tot_h = globals().get("tot_h", 0.0) + per_fwd * float(np.median(res["two fp16 limbs"])); globals()["tot_h"] = tot_h
    err_h = float((o3.double() - o1.double()).abs().max() / o1.double().abs().max())
    best = min(v for k, v in med.items() if k.startswith("new") and "limb" not in k)
    tot_old += per_fwd * med["old"]; tot_new += per_fwd * best
    print("%-20s M=%d N=%d K=%d: " % (name, M, rows, K) + " | ".join("%s %.1f us (%.2f PF/s)" % (k, v, (fl / 2 if "fp16" in k else fl) / v / 1e9) if "split" not in k else "%s %.1f us" % (k, v) for k, v in med.items())
          + " | identical %s | two fp16 limbs vs three bf16 limbs: max rel dev %.1e" % (same, err_h), flush=True)
print("per forward: old %.2f ms, new (better tile) %.2f ms, two fp16 limbs %.2f ms" % (tot_old / 1e3, tot_new / 1e3, tot_h / 1e3))
