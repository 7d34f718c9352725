"""The N = 1024 projections of the backbone at B = 1 (to_out, cross-attention q, FF2, proj) on 128 x 64 tiles (384 workgroups, 1.5
per CU) against ONE round of 192 x 64 tiles (256 workgroups): interleaved rounds in one process, outputs compared bit for bit.
    python tools/gemm_bm192_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sculptmate_amd import _lib, ops

dev = torch.device("cuda:0")
BF = torch.bfloat16
g = torch.Generator().manual_seed(0)


def case(name, M, K, N, residual, ln):
    A = torch.randn(M, K, generator=g).to(BF).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    kw = {}
    if ln:
        stats = torch.zeros(K // 64, M, 2, device=dev); stats[..., 1] = 64.0
        kw.update(ln_stats=stats, ln_colsum=W.float().sum(1).contiguous(), ln_eps=1e-5)
    h0 = torch.randn(M, N, generator=g).to(dev)
    res, outs = {}, {}
    for rnd in range(6):
        for v in ("0", "1", "96", "96w4", "ks"):
            os.environ["SCULPT_GEMM_BM192"] = "1" if v in ("1", "ks") else "0"
            os.environ["SCULPT_GEMM_KS"] = "1" if v == "ks" else "0"
            os.environ["SCULPT_GEMM_BM96"] = {"96": "1", "96w4": "2"}.get(v, "0")
            if residual:
                h = h0.clone()
                kk = dict(kw, residual=h, out_f32=h, out_bf16=torch.empty(M, N, dtype=BF, device=dev), stats_out=torch.zeros(N // 64, M, 2, device=dev))
            else:
                kk = dict(kw, out_bf16=torch.empty(M, N, dtype=BF, device=dev))
            f = lambda: ops.gemm(A, W, bias=bias, **kk)
            f()
            torch.cuda.synchronize()
            if rnd == 0:
                outs[v] = (kk["out_bf16"].clone(), kk.get("out_f32", kk["out_bf16"]).clone(), None if not residual else kk["stats_out"].clone())
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    same = all(torch.equal(outs["0"][0], outs[v][0]) and torch.equal(outs["0"][1], outs[v][1]) for v in ("1", "96", "96w4"))
    st = "" if not residual else " stats max dev %.2e / %.2e" % tuple(float((outs["0"][2] - outs[v][2]).abs().max()) for v in ("1", "96"))
    fl = 2.0 * M * N * K
    print("%-22s M=%d K=%d N=%d  128x64 tiles %.1f us (%.0f TF/s) | one round of 192x64 %.1f us (%.0f TF/s) | one round of 96x128 %.1f us (%.0f TF/s) | the same with 4 waves of 64x48 %.1f us | 192x64 with k-split pairs %.1f us (max rel dev of the fp32 / bf16 output %.1e) | outputs identical (all but k-split): %s%s"
          % (name, M, K, N, np.median(res["0"]), fl / np.median(res["0"]) / 1e6, np.median(res["1"]), fl / np.median(res["1"]) / 1e6,
             np.median(res["96"]), fl / np.median(res["96"]) / 1e6, np.median(res["96w4"]), np.median(res["ks"]),
             float((outs["ks"][1].float() - outs["0"][1].float()).abs().max() / outs["0"][1].float().abs().max()), same, st), flush=True)


case("cross-attn q (LN fold)", 3072, 1024, 1024, False, True)
case("to_out (residual)", 3072, 1024, 1024, True, False)
case("FF2 (residual)", 3072, 4096, 1024, True, False)
case("plain", 3072, 1024, 1024, False, False)
case("plain K=4096", 3072, 4096, 1024, False, False)
