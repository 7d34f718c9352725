"""Experiment: the backbone's token rows in TWO halves on two HIP streams (every Linear / cross-attention / FF of a half is
independent of the other half; only the self-attention needs both halves' K / V), so that one half's launch latencies sit under the
other half's kernels -- as one hipGraph (twice the launches would otherwise be host-bound: 12 us of Python per launch).
    python tools/try_split_rows.py"""
import math, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from sculptmate_amd import _lib, ops, synth

dev = torch.device("cuda:0")
BF16 = torch.bfloat16


def run_blocks_split(self, st, ctx, first_self_attention_done=False, batch=1, ctx_tokens=None):
    assert batch == 1 and self.precision == "bf16"
    b, w = self.cfg["backbone"], self._w
    nh, hd = b["num_attention_heads"], b["attention_head_dim"]
    D = nh * hd
    M, Mc = st["h"].shape[0], ctx.shape[0]
    Tc = Mc if ctx_tokens is None else ctx_tokens
    ldc = ((Tc + 63) // 64) * 64
    nL = len(w["blocks"])
    ck_all = self._b("bb_ck", (Mc, nL * D), self.adt)
    cvt_all = self._b("bb_cvt", (nL * D, ldc), self.adt, zero=True)
    self._gemm(ctx, w["ca_kv_all"], out_bf16=ck_all, out_t=cvt_all, n_split=nL * D, M=Mc)
    Mp = ((M + 63) // 64) * 64
    q = self._b("bb_q", (M, D), self.adt)
    qk = self._b("bb_qk", (M, 2 * D), self.adt)
    vt = self._b("bb_vt", (D, Mp), self.adt, zero=True)
    att = self._b("bb_att", (M, D), self.adt)
    ff = self._b("bb_ff", (M, 4 * D), self.adt)
    H = M // 2
    halves = [(0, H), (H, M)]
    s0 = torch.cuda.current_stream(self.device)
    s1 = getattr(self, "_split_stream", None)
    if s1 is None:
        s1 = self._split_stream = torch.cuda.Stream(self.device)
    streams = [s0, s1]
    sub = [{"h": st["h"][a:z], "hb": st["hb"][a:z], "stats": st["stats"][:, a:z], "name": st["name"]} for a, z in halves]
    fork = torch.cuda.Event(); fork.record(s0); s1.wait_event(fork)
    att_done = [None, None]
    for li, L in enumerate(w["blocks"]):
        ck, cvt = ck_all[:, li * D:(li + 1) * D], cvt_all[li * D:(li + 1) * D]
        sa = li > 0 or not first_self_attention_done
        qkv_done = [None, None]
        if sa:
            for i, (a, z) in enumerate(halves):
                with torch.cuda.stream(streams[i]):
                    if att_done[1 - i] is not None:
                        streams[i].wait_event(att_done[1 - i])   # the other half still reads the previous layer's K / V
                    self._ln_gemm(sub[i], L, "sa_qkv", 1e-5, out_bf16=qk[a:z], out_t=vt[:, a:], n_split=2 * D)
                    qkv_done[i] = torch.cuda.Event(); qkv_done[i].record(streams[i])
        for i, (a, z) in enumerate(halves):
            with torch.cuda.stream(streams[i]):
                if sa:
                    streams[i].wait_event(qkv_done[1 - i])
                    ops.attention(qk[a:z, :D], qk[:, D:], vt, att[a:z], z - a, M, nh, None)
                    att_done[i] = torch.cuda.Event(); att_done[i].record(streams[i])
                    self._res_gemm(sub[i], att[a:z], L["sa_o"], L["sa_ob"])
                self._ln_gemm(sub[i], L, "ca_q", 1e-5, out_bf16=q[a:z])
                ops.attention(q[a:z], ck, cvt, att[a:z], z - a, Tc, nh, None)
                self._res_gemm(sub[i], att[a:z], L["ca_o"], L["ca_ob"])
                self._ln_gemm(sub[i], L, "ff1", 1e-5, out_bf16=ff[a:z], epilogue=_lib.EPI_GEGLU)
                self._res_gemm(sub[i], ff[a:z], L["ff2"], L["ff2_b"])
    join = torch.cuda.Event(); join.record(s1); s0.wait_event(join)
    return st


def main():
    model, sd = bench.build_model(dev, 0)
    img = torch.from_numpy(synth.composite_rgb(synth.image_rgba(seed=100))).to(dev).contiguous()
    with torch.no_grad():
        def timeit(fn, n=20):
            fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): fn()
            torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

        fwd = lambda: model.forward(img)
        ref = fwd().clone()
        print("eager, one chain:            %.3f ms" % timeit(fwd))

        def graph_of(fn):
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2): fn()
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = fn()
            return g, out

        g1, o1 = graph_of(fwd)
        print("graph, one chain:            %.3f ms   same %s" % (timeit(g1.replay), torch.equal(o1, ref)))
        orig = model._run_blocks
        model._run_blocks = types.MethodType(run_blocks_split, model)
        split = fwd().clone()
        rel = float((split - ref).norm() / ref.norm())
        print("eager, two half-row chains:  %.3f ms   rel diff of the scene code %.2e" % (timeit(fwd), rel))
        g2, o2 = graph_of(fwd)
        print("graph, two half-row chains:  %.3f ms   same as eager split %s" % (timeit(g2.replay), torch.equal(o2, split)))
        model._run_blocks = orig


main()
