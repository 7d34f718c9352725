"""Where the time of one 192 x 256-tile GEMM launch goes, per workgroup and per CU (experiment builds only).

    SCULPT_EXTRA_HIPCC_FLAGS=-DSCULPT_EXPERIMENTS python -m sculptmate_amd.build
    SCULPT_EXTRA_HIPCC_FLAGS=-DSCULPT_EXPERIMENTS python tools/gemm_timeline.py [ff1|qkv|plain|o|ff2|l3o|l3ff2|l3qkv|l3ff1]

gemm256_kernel / gemm_bf16_kernel stamp s_memrealtime at entry / first K-tile landed / K loop done / epilogue arithmetic done / exit, and the CU it ran
on (csrc/gemm.hip, GEMM_STAMP).  Printed: the medians of the four phases, and per CU the order of its workgroups with the gaps
between one's exit and the next one's entry.
"""
import ctypes
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from sculptmate_amd import _lib, ops

dev = torch.device("cuda:0")
BF = torch.bfloat16
what = sys.argv[1] if len(sys.argv) > 1 else "ff1"
lib = _lib.lib
lib.sculpt_experiment_gemm_stamps.argtypes = [ctypes.c_void_p]
lib.sculpt_experiment_gemm_stamps.restype = None

M, K = 3072, 1024
TILE = None     # (weight rows, activation rows) of the launch's tiles when they are not 256 x 192
res = stats = None
if what in ("o", "ff2"):     # the N = 1024 projections: + bias + residual -> fp32, bf16 copy, slice statistics (192 x 64 tiles)
    N, epi, K = 1024, 0, (1024 if what == "o" else 4096)
    W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(BF)
    bias = torch.randn(N, device=dev) * 0.1
    res = torch.randn(M, N, device=dev)
    stats = torch.zeros(N // 64, M, 2, device=dev)
    TILE = (64, 192)
elif what == "ff1":
    N, epi = 4096, _lib.EPI_GEGLU
    W = (torch.randn(2 * N, K, device=dev) / math.sqrt(K)).to(BF)
    bias = torch.randn(2 * N, device=dev) * 0.1
elif what == "qkv":
    N, epi = 3072, 0
    W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(BF)
    bias = torch.randn(N, device=dev) * 0.1
else:
    N, epi = 8192, 0
    W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(BF)
    bias = None
L3 = what.startswith("l3")     # the limb GEMM of the tolerance mode (two fp16 limbs): l3o, l3ff2 (+ residual -> fp32), l3qkv, l3ff1 (-> limbs)
if L3:
    N, K, epi = {"l3o": (1024, 1024, 0), "l3ff2": (1024, 4096, 0), "l3qkv": (3072, 1024, 0), "l3ff1": (4096, 1024, _lib.EPI_GEGLU)}[what]
    rows = 2 * N if epi else N
    Wf = torch.randn(rows, K, device=dev) / math.sqrt(K)
    A_h = ops.Limbs.of(torch.randn(M, K, device=dev), fmt="f16x2")
    W_h = ops.Limbs.of(ops.geglu_row_blocks(Wf) if epi else Wf, fmt="f16x2", weight=True)
    bias = torch.randn(rows, device=dev) * 0.1
    res = torch.randn(M, N, device=dev) if what in ("l3o", "l3ff2") else None
    out_lt = ops.Limbs(M, N, dev, fmt="f16x2") if res is None else None
    stats = None
    TILE = (64 if epi else 128, 128)
A = torch.randn(M, K, device=dev).to(BF)
out = torch.empty(M, N, dtype=BF, device=dev)
outf = torch.empty(M, N, device=dev) if res is not None else None
ntiles = (N // (128 if epi else 256)) * (M // 192) if TILE is None else (N // TILE[0]) * (M // TILE[1])
stamps = torch.zeros(ntiles * 16, dtype=torch.int64, device=dev)


def run():
    if L3:
        ops.gemm_l3p(A_h, W_h, M, N, K, bias=bias, residual=res, out=outf, out_lt=out_lt, epilogue=epi)
    elif res is not None:
        ops.gemm(A, W, bias=bias, residual=res, out_f32=outf, out_bf16=out, stats_out=stats)
    else:
        ops.gemm(A, W, bias=bias, out_bf16=out, epilogue=epi)


for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
print("%s: M %d N %d K %d, %d tiles; launch %.1f us (events, 20 back to back)" % (what, M, N, K, ntiles, e0.elapsed_time(e1) * 50.0))
lib.sculpt_experiment_gemm_stamps(stamps.data_ptr())
run()
torch.cuda.synchronize()
stamps.zero_()
run()
torch.cuda.synchronize()
lib.sculpt_experiment_gemm_stamps(None)
s = stamps.cpu().numpy().reshape(ntiles, 16)
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/stamps_%s.npy" % what, s)
assert (s[:, 4] != 0).all(), "no stamps: not an experiment build, or the launch took another kernel"
t = (s[:, :5] - s[:, 0].min()) / 100.0                       # us since the launch's first entry
clk = np.median((s[:, 12] - s[:, 8]) / np.maximum(t[:, 4] - t[:, 0], 1e-9))
print("shader clock during the launch: %.0f MHz (s_memtime ticks per s_memrealtime us, median over the workgroups)" % clk)
names = ("entry -> first K-tile landed (incl. LayerNorm prologue)", "K loop", "epilogue arithmetic (+ staging writes)", "copy-out")
ph = np.diff(t, axis=1)
for k, nme in enumerate(names):
    print("  %-58s median %6.2f us   p10 %6.2f   p90 %6.2f" % (nme, np.median(ph[:, k]), np.quantile(ph[:, k], 0.1), np.quantile(ph[:, k], 0.9)))
print("  %-58s median %6.2f us" % ("workgroup, entry -> exit", np.median(t[:, 4] - t[:, 0])))
print("  launch, first entry -> last exit: %.2f us" % t[:, 4].max())
# per CU: HW_ID bits 8-11 cu, 12 sh, 13-15 se; XCC_ID bits 0-3
cu_key = ((s[:, 6] & 15) << 16) | (s[:, 5] & 0xFF00)
cus = {}
for w in range(ntiles):
    cus.setdefault(int(cu_key[w]), []).append(w)
print("CUs used: %d; workgroups per CU: %s" % (len(cus), dict(zip(*[x.tolist() for x in np.unique([len(v) for v in cus.values()], return_counts=True)]))))
gaps, firsts, lasts = [], [], []
rounds = {}
for key, ws in cus.items():
    ws.sort(key=lambda w: t[w, 0])
    firsts.append(t[ws[0], 0])
    lasts.append(t[ws[-1], 4])
    for r, w in enumerate(ws):
        rounds.setdefault(r, []).append(w)
    for a, b in zip(ws[:-1], ws[1:]):
        gaps.append(t[b, 0] - t[a, 4])
print("first entry on a CU: median %.2f us, max %.2f" % (np.median(firsts), max(firsts)))
if gaps:
    print("gap between a workgroup's exit and the next one's entry on the same CU: median %.2f us, p10 %.2f, p90 %.2f" % (
        np.median(gaps), np.quantile(gaps, 0.1), np.quantile(gaps, 0.9)))
print("last exit on a CU: median %.2f us, min %.2f, max %.2f" % (np.median(lasts), min(lasts), max(lasts)))
for r, ws in sorted(rounds.items()):
    print("  round %d workgroups (%d): fill %.2f, K loop %.2f, epilogue %.2f, copy-out %.2f us (medians)" % (
        (r, len(ws)) + tuple(np.median(ph[ws, k]) for k in range(4))))
