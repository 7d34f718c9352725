"""Calibration only: our bf16 GEMM vs torch.matmul (hipBLASLt) on the TripoSR shapes (never used by the product path)."""
import sys, os, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sculptmate_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16

def timeit(fn, n=20, warm=5):
    for _ in range(warm): fn()
    ts = []
    for _ in range(n):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 10)
    ts.sort(); return ts[len(ts) // 2] * 1e3

for name, M, N, K in [("sa_qkv", 3072, 3072, 1024), ("o/q", 3072, 1024, 1024), ("ff1", 3072, 8192, 1024), ("ff2", 3072, 1024, 4096),
                      ("ca_kv_all", 1025, 32768, 768), ("vit qkv", 1025, 2304, 768), ("vit o", 1025, 768, 768),
                      ("vit f1", 1025, 3072, 768), ("vit f2", 1025, 768, 3072)]:
    A = torch.randn(M, K, device=dev).to(BF); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(BF)
    out = torch.empty(M, N, dtype=BF, device=dev)
    Wt = W.t()
    t_ours = timeit(lambda: ops.gemm(A, W, out_bf16=out)) if N % 128 == 0 else float("nan")
    t_blas = timeit(lambda: torch.matmul(A, Wt, out=out))
    fl = 2.0 * M * N * K
    print("%-10s M%5d N%6d K%5d  ours %7.1f us %6.1f TF/s | hipblaslt %7.1f us %6.1f TF/s" % (name, M, N, K, t_ours, fl / t_ours / 1e6, t_blas, fl / t_blas / 1e6))
q = torch.randn(1, 16, 3072, 64, device=dev).to(BF); k = torch.randn(1, 16, 3072, 64, device=dev).to(BF); v = torch.randn_like(k)
t = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v))
print("torch sdpa self 3072x3072x16x64: %.1f us %.1f TF/s" % (t, 4.0 * 3072 * 3072 * 1024 / t / 1e6))
k2 = torch.randn(1, 16, 1025, 64, device=dev).to(BF); v2 = torch.randn_like(k2)
t = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k2, v2))
print("torch sdpa cross 3072x1025x16x64: %.1f us %.1f TF/s" % (t, 4.0 * 3072 * 1025 * 1024 / t / 1e6))
