"""GPU micro-timing of the transformer primitives at the TripoSR shapes (HIP events, median of 20)."""
import sys, os, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    ts.sort(); return ts[len(ts) // 2] * 1e3  # us (back-to-back launches, like the real pipeline)

def gemm_case(name, M, N, K, epi=0, count=1):
    A = torch.randn(M, K, device=dev).to(BF); W = (torch.randn((2 * N if epi == 2 else N), K, device=dev) / math.sqrt(K)).to(BF)
    out = torch.empty(M, N, dtype=BF, device=dev)
    us = timeit(lambda: ops.gemm(A, W, out_bf16=out, epilogue=epi))
    fl = 2.0 * M * K * (2 * N if epi == 2 else N)
    print("%-28s M%5d N%5d K%5d  %7.1f us  %6.1f TF/s   x%d = %.2f ms" % (name, M, N, K, us, fl / us / 1e6, count, us * count / 1e3))
    return us * count

def attn_case(name, Tq, Tk, heads, count):
    D = heads * 64
    q = torch.randn(Tq, D, device=dev).to(BF); k = torch.randn(Tk, D, device=dev).to(BF)
    vt = torch.zeros(D, ((Tk + 63) // 64) * 64, dtype=BF, device=dev); vt[:, :Tk] = torch.randn(D, Tk, device=dev).to(BF)
    o = torch.empty(Tq, D, dtype=BF, device=dev)
    us = timeit(lambda: ops.attention(q, k, vt, o, Tq, Tk, heads, 0.125))
    fl = 4.0 * Tq * Tk * D
    print("%-28s Tq%5d Tk%5d H%3d       %7.1f us  %6.1f TF/s   x%d = %.2f ms" % (name, Tq, Tk, heads, us, fl / us / 1e6, count, us * count / 1e3))
    return us * count

tot = 0
tot += gemm_case("bb self qkv", 3072, 3072, 1024, 0, 16)
tot += gemm_case("bb cross kv all", 1025, 32768, 768, 0, 1)
tot += gemm_case("bb self v / q / o (x5)", 3072, 1024, 1024, 0, 16 * 5)
tot += gemm_case("bb cross k,v", 1025, 1024, 768, 0, 32)
tot += gemm_case("bb ff1 geglu", 3072, 4096, 1024, 2, 16)
tot += gemm_case("bb ff2", 3072, 1024, 4096, 0, 16)
tot += gemm_case("vit qk", 1025, 1536, 768, 0, 12)
tot += gemm_case("vit v / o", 1025, 768, 768, 0, 24)
tot += gemm_case("vit f1 gelu", 1025, 3072, 768, 1, 12)
tot += gemm_case("vit f2", 1025, 768, 3072, 0, 12)
tot += attn_case("self attn", 3072, 3072, 16, 16)
tot += attn_case("cross attn", 3072, 1025, 16, 16)
tot += attn_case("vit attn", 1025, 1025, 12, 12)
x = torch.randn(3072, 1024, device=dev); w = torch.ones(1024, device=dev); y = torch.empty(3072, 1024, dtype=BF, device=dev)
us = timeit(lambda: ops.layernorm(x, w, w, 1e-5, y=y)); print("layernorm 3072x1024 %.1f us x48 = %.2f ms" % (us, us * 48 / 1e3)); tot += us * 48
print("total transformer kernels: %.2f ms" % (tot / 1e3))
