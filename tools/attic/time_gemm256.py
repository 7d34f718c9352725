"""The 256 x 256 tile kernel (gemm256_kernel) against the 128-row tiles on the backbone's big projections, interleaved rounds in one
process: SCULPT_GEMM_256 = 0 / 2 is read per call.  Operands are random; the LayerNorm fold and the epilogues are the real ones."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sculptmate_amd import _lib, ops

dev = torch.device("cuda:0")
BF = torch.bfloat16
g = torch.Generator().manual_seed(0)


def case(name, M, K, N, epi, split=0):
    rows = 2 * N if epi == _lib.EPI_GEGLU else N
    A = torch.randn(M, K, generator=g).to(BF).to(dev)
    W = (torch.randn(rows, K, generator=g) / K ** 0.5).to(BF).to(dev)
    bias = torch.randn(rows, generator=g).to(dev)
    cs = W.float().sum(1).contiguous()
    stats = torch.zeros(K // 64, M, 2, device=dev)
    stats[..., 1] = 64.0
    Mp = (M + 63) // 64 * 64
    if split:
        o = torch.empty(M, split, dtype=BF, device=dev)
        ot = torch.zeros(N - split, Mp, dtype=BF, device=dev)
        kw = dict(out_bf16=o, out_t=ot, n_split=split)
    else:
        o = torch.empty(M, N, dtype=BF, device=dev)
        ot = None
        kw = dict(out_bf16=o)
    f = lambda: ops.gemm(A, W, bias=bias, epilogue=epi, ln_stats=stats, ln_colsum=cs, ln_eps=1e-5, **kw)
    res = {}
    outs = {}
    for rnd in range(4):
        for mode in ("0", "2", "192"):
            os.environ["SCULPT_GEMM_256"] = "2" if mode == "192" else mode
            os.environ["SCULPT_GEMM_192"] = "1" if mode == "192" else "0"
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(mode, []).append(e0.elapsed_time(e1) / 20 * 1e3)
            outs[mode] = (o.clone(), None if ot is None else ot.clone())
    same = all(torch.equal(outs["0"][0], outs[m][0]) and (ot is None or torch.equal(outs["0"][1], outs[m][1])) for m in ("2", "192"))
    fl = 2.0 * M * rows * K
    t0, t2, t3 = np.median(res["0"]), np.median(res["2"]), np.median(res["192"])
    print("%-28s M=%d K=%d N=%d: 128-row tiles %.1f us (%.0f TF/s) | 256 x 256 %.1f us (%.0f TF/s) | 192 x 256 %.1f us (%.0f TF/s) | identical outputs: %s"
          % (name, M, K, N, t0, fl / t0 / 1e6, t2, fl / t2 / 1e6, t3, fl / t3 / 1e6, same))


case("FF1 + GEGLU", 3072, 1024, 4096, _lib.EPI_GEGLU)
case("fused Q|K|V^T", 3072, 1024, 3072, _lib.EPI_NONE, split=2048)
case("cross-attention q", 3072, 1024, 1024, _lib.EPI_NONE)
case("K/V of all layers", 1025, 768, 32768, _lib.EPI_NONE, split=16384)
case("SF3D FF1 + GEGLU", 27648, 1024, 4096, _lib.EPI_GEGLU)
