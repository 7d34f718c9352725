import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from sculptmate_amd import _lib, ops
dev = torch.device("cuda:0"); g = torch.Generator().manual_seed(0)
def timed(f, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for name, M, N, K, epi in (("QKV", 3072, 3072, 1024, 0), ("o/q", 3072, 1024, 1024, 0), ("FF1", 3072, 4096, 1024, _lib.EPI_GEGLU), ("FF2", 3072, 1024, 4096, 0),
                           ("ViT qkv", 1025, 2304, 768, 0), ("ViT o", 1025, 768, 768, 0), ("ViT f1", 1025, 3072, 768, _lib.EPI_GELU), ("ViT f2", 1025, 768, 3072, 0)):
    rows = 2 * N if epi == _lib.EPI_GEGLU else N
    A = torch.randn(M, K, generator=g).to(dev); W = (torch.randn(rows, K, generator=g) / K ** 0.5).to(dev)
    A_h = ops.Limbs.of(A, fmt="f16x2"); W_h = ops.Limbs.of(ops.geglu_row_blocks(W) if epi == _lib.EPI_GEGLU else W, fmt="f16x2", weight=True)
    o = torch.empty(M, N, device=dev)
    res = {}
    forms = {"4w128": ("0", "0"), "4w64": ("1", "0"), "8w128": ("0", "1"), "default": (None, None)}
    for rnd in range(5):
        for k, (b64, n8) in forms.items():
            for kk, v in (("SCULPT_L3P_BM64", b64), ("SCULPT_L3P_NW8", n8)):
                if v is None: os.environ.pop(kk, None)
                else: os.environ[kk] = v
            f = lambda: ops.gemm_l3p(A_h, W_h, M, N, K, out=o, epilogue=epi)
            f(); torch.cuda.synchronize()
            res.setdefault(k, []).append(timed(f))
    fl = 2.0 * M * rows * K * 3
    print("%-8s" % name + " | ".join("%s %.1f us (%.2f PF/s)" % (k, np.median(v), fl / np.median(v) / 1e9) for k, v in res.items()), flush=True)
