"""Per-launch cost of the LayerNorm fold (producer statistics / consumer epilogue) against the plain GEMMs + stand-alone LN."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sculptmate_amd import ops, _lib
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

M, D = 3072, 1024
h = torch.randn(M, D, device=dev); hb = torch.empty(M, D, dtype=BF, device=dev); stats = torch.zeros(D // 64, M, 2, device=dev)
ops.row_slice_stats(h, stats, hb)
g = torch.ones(D, device=dev); xn = torch.empty(M, D, dtype=BF, device=dev)
print("layernorm                         %6.1f us" % timeit(lambda: ops.layernorm(h, g, g, 1e-5, y=xn)))
print("row_slice_stats                   %6.1f us" % timeit(lambda: ops.row_slice_stats(h, stats, hb)))
for K in (1024, 4096):
    A = torch.randn(M, K, device=dev).to(BF); W = (torch.randn(D, K, device=dev) / math.sqrt(K)).to(BF); b = torch.randn(D, device=dev)
    print("residual gemm K=%4d plain        %6.1f us" % (K, timeit(lambda: ops.gemm(A, W, bias=b, residual=h, out_f32=h))))
    print("residual gemm K=%4d +bf16 copy   %6.1f us" % (K, timeit(lambda: ops.gemm(A, W, bias=b, residual=h, out_f32=h, out_bf16=hb))))
    print("residual gemm K=%4d +copy +stats %6.1f us" % (K, timeit(lambda: ops.gemm(A, W, bias=b, residual=h, out_f32=h, out_bf16=hb, stats_out=stats))))
h.normal_(); ops.row_slice_stats(h, stats, hb)
for name, N, epi, split in (("ca_q", 1024, 0, 0), ("qkv", 3072, 0, 2048), ("ff1", 4096, 2, 0)):
    rows = 2 * N if epi == 2 else N
    W = (torch.randn(rows, D, device=dev) / 32).to(BF); b = torch.randn(rows, device=dev); cs = torch.randn(rows, device=dev)
    out = torch.empty(M, N if not split else split, dtype=BF, device=dev)
    ot = torch.zeros(N - split, M, dtype=BF, device=dev) if split else None
    kw = dict(out_bf16=out, out_t=ot, n_split=split, epilogue=epi)
    print("%-5s N=%4d plain                 %6.1f us" % (name, N, timeit(lambda: ops.gemm(hb, W, bias=b, **kw))))
    print("%-5s N=%4d LayerNorm folded      %6.1f us" % (name, N, timeit(lambda: ops.gemm(hb, W, bias=b, ln_stats=stats, ln_colsum=cs, **kw))))
