"""What the END of a GEMM epilogue costs: each of the backbone's launches with and without its output stores
(SCULPT_GEMM_DBG_NOSTORE=1: no row passes the store guard), interleaved rounds in one process.  Timing only.
READ WITH CARE: where the stores come straight from the accumulator layout (128-row kernel; 256-row kernel with SCULPT_GEMM_STAGE=0)
hipcc sinks the arithmetic that only feeds a store -- bias, LayerNorm fold, erf-GELU, bf16 conversion -- behind the guard, so
"without" drops that arithmetic too (FF1: 13 us = ~11 us of GEGLU math + ~2 us of stores).  With the staged epilogue the math feeds
the LDS writes and stays: there the difference is the stores alone (FF1 1.9 us, fused Q|K|V^T 3.4 us)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sculptmate_amd import _lib, ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
def case(name, M, K, N, epi=0, split=0, residual=False, ln=True):
    rows = 2 * N if epi == _lib.EPI_GEGLU else N
    A = torch.randn(M, K, generator=g).to(BF).to(dev); W = (torch.randn(rows, K, generator=g) / K ** 0.5).to(BF).to(dev)
    bias = torch.randn(rows, generator=g).to(dev)
    kw = {}
    if ln:
        stats = torch.zeros(K // 64, M, 2, device=dev); stats[..., 1] = 64.0
        kw.update(ln_stats=stats, ln_colsum=W.float().sum(1).contiguous(), ln_eps=1e-5)
    Mp = (M + 63) // 64 * 64
    if residual:
        h = torch.randn(M, N, generator=g).to(dev)
        kw.update(residual=h, out_f32=h, out_bf16=torch.empty(M, N, dtype=BF, device=dev), stats_out=torch.zeros(N // 64, M, 2, device=dev))
    elif split:
        kw.update(out_bf16=torch.empty(M, split, dtype=BF, device=dev), out_t=torch.zeros(N - split, Mp, dtype=BF, device=dev), n_split=split)
    else:
        kw.update(out_bf16=torch.empty(M, N, dtype=BF, device=dev))
    f = lambda: ops.gemm(A, W, bias=bias, epilogue=epi, **kw)
    res = {}
    for rnd in range(5):
        for mode in ("0", "1"):
            os.environ["SCULPT_GEMM_DBG_NOSTORE"] = mode
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(mode, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    os.environ["SCULPT_GEMM_DBG_NOSTORE"] = "0"
    t0, t1 = np.median(res["0"]), np.median(res["1"])
    print("%-24s M=%-5d K=%-4d N=%-5d with stores %.1f us | without %.1f us | stores cost %.1f us (%.0f %%)" % (name, M, K, N, t0, t1, t0 - t1, 100 * (t0 - t1) / t0), flush=True)
for B in (1, 4):
    T = 3072 * B
    case("FF1 + GEGLU B=%d" % B, T, 1024, 4096, _lib.EPI_GEGLU)
    case("fused Q|K|V^T B=%d" % B, T, 1024, 3072, 0, split=2048)
    case("cross-attn q B=%d" % B, T, 1024, 1024)
    case("to_out (res) B=%d" % B, T, 1024, 1024, residual=True, ln=False)
    case("FF2 (res) B=%d" % B, T, 4096, 1024, residual=True, ln=False)
case("K/V all layers B=1", 1025, 768, 32768, 0, split=16384, ln=False)
case("ViT qkv", 1025, 768, 2304, 0, split=1536)
case("ViT f1 GELU", 1025, 768, 3072, _lib.EPI_GELU)
case("ViT f2 (res)", 1025, 3072, 768, residual=True, ln=False)
