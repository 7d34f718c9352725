"""The three-limb GEMM (gemm_l3.hip) on the backbone's shapes: pipelined K loop (split of K-step kt + 1 in the MFMA shadows of kt)
against the plain one (SCULPT_L3_PIPE=0), interleaved rounds in one process; executed FLOPs = 6 x algorithmic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sculptmate_amd import _lib, ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
def case(name, M, K, N, epi=0, residual=False):
    rows = 2 * N if epi == _lib.EPI_GEGLU else N
    A = torch.randn(M, K, generator=g).to(dev); W = (torch.randn(rows, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(rows, generator=g).to(dev)
    out = torch.randn(M, N, generator=g).to(dev)
    f = lambda: ops.gemm_f32(A, W, bias=b, residual=out if residual else None, out=out, epilogue=epi, l3=True)
    res = {}
    outs = {}
    for rnd in range(4):
        for mode in ("0", "1", "64"):
            os.environ["SCULPT_L3_PIPE"] = "1" if mode == "64" else mode
            os.environ["SCULPT_L3_BM64"] = "1" if mode == "64" else "0"
            for _ in range(2): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(mode, []).append(e0.elapsed_time(e1) / 10 * 1e3)
            if not residual:
                outs[mode] = out.clone()
    fl = 2.0 * M * rows * K
    t0, t1, t2 = np.median(res["0"]), np.median(res["1"]), np.median(res["64"])
    same = (not residual) and torch.equal(outs["0"], outs["1"]) and torch.equal(outs["0"], outs["64"])
    print("%-22s M=%-5d K=%-4d N=%-5d 128-row plain %.1f us (%.0f TF/s executed) | 128-row pipelined %.1f us (%.0f TF/s, %.3f of the bf16 peak) | 64-row tiles %.1f us (%.0f TF/s) | identical: %s"
          % (name, M, K, N, t0, 6 * fl / t0 / 1e6, t1, 6 * fl / t1 / 1e6, 6 * fl / t1 / 1e6 / 2500, t2, 6 * fl / t2 / 1e6, same), flush=True)
case("FF1 + GEGLU", 3072, 1024, 4096, _lib.EPI_GEGLU)
case("fused Q|K|V", 3072, 1024, 3072)
case("to_out (residual)", 3072, 1024, 1024, residual=True)
case("FF2 (residual)", 3072, 4096, 1024, residual=True)
case("K/V all layers", 1025, 768, 32768)
case("ViT qkv", 1025, 768, 2304)
case("ViT f2 (residual)", 1025, 3072, 768, residual=True)
case("ViT o (residual)", 1025, 768, 768, residual=True)
case("ViT f1 GELU", 1025, 768, 3072, _lib.EPI_GELU)
os.environ.pop("SCULPT_L3_BM64", None); os.environ.pop("SCULPT_L3_PIPE", None)
