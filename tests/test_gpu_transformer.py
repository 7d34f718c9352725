"""HIP transformer primitives + TSR.forward vs the torch-fp32 oracle (MI355X)."""
import math
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import tsr_ref
from sculptmate_amd import synth
from sculptmate_amd.tsr.spec import SMALL_CFG, make_cfg

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12)), float((a - b).abs().max())


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (1025, 768, 768), (3072, 1024, 1024), (300, 256, 4096), (1, 128, 64),
                                   (3072, 1024, 4096), (2950, 1024, 512), (3073, 1024, 128)])  # deep K, ragged M
def test_gemm_bias_residual_vs_fp32_reference(cuda, M, N, K):
    from sculptmate_amd import ops

    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(BF)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    ref = A.float() @ W.float().t() + bias + res  # exact products of bf16 inputs, fp32 accumulate
    out = torch.empty(M, N, device=cuda)
    outb = torch.empty(M, N, dtype=BF, device=cuda)
    Mp = ((M + 63) // 64) * 64
    outt = torch.zeros(N, Mp, dtype=BF, device=cuda)
    ops.gemm(A.to(cuda), W.to(cuda), bias=bias.to(cuda), residual=res.to(cuda), out_f32=out, out_bf16=outb, out_t=outt)
    rel, mx = _rel(out, ref)
    assert rel < 2e-6 * math.sqrt(K) and mx < 1e-3, (rel, mx)  # fp32 accumulation error only
    assert _rel(outb, ref)[0] < 4e-3  # bf16 rounding of the result
    assert _rel(outt[:, :M].t(), ref)[0] < 4e-3
    assert (outt[:, M:] == 0).all()


@pytest.mark.parametrize("M,D,N,epi", [(3072, 1024, 1024, 0), (1025, 768, 2304, 0), (1025, 768, 3072, 1), (3072, 1024, 4096, 2),
                                       (77, 256, 128, 0)])
def test_gemm_layernorm_fold_vs_fp32_reference(cuda, M, D, N, epi):
    """sculpt_gemm_bf16_ln: a residual GEMM writes h (fp32), bf16(h) and the 32-column slice statistics; the next GEMM
    consumes bf16(h) with LayerNorm folded in == Linear(LayerNorm(h)) of the reference (basic_transformer_block.py:149-206),
    within bf16 operand rounding.  Rows get a large common offset (|mean| ~ 3 sigma) to exercise the mean term."""
    import torch.nn.functional as F

    from sculptmate_amd import _lib, ops

    g = torch.Generator().manual_seed(M + D + N + epi)
    K0 = 256
    A0 = torch.randn(M, K0, generator=g).to(BF)
    W0 = (torch.randn(D, K0, generator=g) / math.sqrt(K0)).to(BF)
    b0 = torch.randn(D, generator=g)
    res = torch.randn(M, D, generator=g) + 3.0 * torch.randn(M, 1, generator=g)
    h_ref = A0.float() @ W0.float().t() + b0 + res
    h = torch.empty(M, D, device=cuda)
    hb = torch.empty(M, D, dtype=BF, device=cuda)
    stats = torch.zeros(D // 64, M, 2, device=cuda)
    ops.gemm(A0.to(cuda), W0.to(cuda), bias=b0.to(cuda), residual=res.to(cuda), out_f32=h, out_bf16=hb, stats_out=stats)
    assert _rel(h, h_ref)[0] < 1e-5
    sl = h.cpu().view(M, D // 64, 64)
    mean_ref = sl.mean(-1)
    m2_ref = ((sl - mean_ref[..., None]) ** 2).sum(-1)
    assert (stats[..., 0].cpu().t() - mean_ref).abs().max() < 1e-5 and _rel(stats[..., 1].t(), m2_ref)[0] < 1e-5
    # the same statistics from the stand-alone kernel (rows that do not come out of a GEMM)
    stats2 = torch.zeros_like(stats)
    hb2 = torch.empty_like(hb)
    ops.row_slice_stats(h, stats2, hb2)
    assert torch.equal(hb2, hb) and _rel(stats2, stats)[0] < 1e-6  # same numbers, different summation order
    # consumer
    gamma = 1.0 + 0.3 * torch.randn(D, generator=g)
    beta = 0.2 * torch.randn(D, generator=g)
    rows = 2 * N if epi == 2 else N
    W = torch.randn(rows, D, generator=g) / math.sqrt(D)
    bias = torch.randn(rows, generator=g)
    eps = 1e-5
    pre = F.linear(F.layer_norm(h.cpu(), (D,), gamma, beta, eps), W, bias)
    ref = pre if epi == 0 else (F.gelu(pre) if epi == 1 else pre[:, :N] * F.gelu(pre[:, N:]))
    Wp, bp, cs = ops.fold_layernorm(W, bias, gamma, beta)
    out = torch.empty(M, N, device=cuda)
    ops.gemm(hb, Wp.to(BF).to(cuda), bias=bp.to(cuda), out_f32=out, epilogue=epi, ln_stats=stats, ln_colsum=cs.to(cuda), ln_eps=eps)
    rel, mx = _rel(out, ref)
    assert rel < 6e-3, (rel, mx)  # bf16 rounding of both operands (2^-9 each), fp32 accumulate
    # tight: against the same arithmetic on the host (bf16 operands, fp32 everything else)
    hq, Wq = hb.float().cpu(), Wp.to(BF).float()
    mu = h.cpu().mean(-1, keepdim=True)
    rstd = torch.rsqrt(h.cpu().var(-1, unbiased=False, keepdim=True) + eps)
    pre2 = rstd * (hq @ Wq.t() - mu * cs[None, :]) + bp
    ref2 = pre2 if epi == 0 else (F.gelu(pre2) if epi == 1 else pre2[:, :N] * F.gelu(pre2[:, N:]))
    assert _rel(out, ref2)[0] < 2e-5, _rel(out, ref2)
    if epi == 0 and N % 256 == 0:  # fused [Q|K|V] form: column split + transposed output
        Mp = ((M + 63) // 64) * 64
        o1 = torch.empty(M, N // 2, dtype=BF, device=cuda)
        ot = torch.zeros(N // 2, Mp, dtype=BF, device=cuda)
        ops.gemm(hb, Wp.to(BF).to(cuda), bias=bp.to(cuda), out_bf16=o1, out_t=ot, n_split=N // 2, ln_stats=stats,
                 ln_colsum=cs.to(cuda), ln_eps=eps)
        assert _rel(o1, ref2[:, :N // 2])[0] < 4e-3 and _rel(ot[:, :M].t(), ref2[:, N // 2:])[0] < 4e-3


@pytest.mark.parametrize("Tq,Tk,heads", [(3072, 3072, 16), (3072, 1025, 16), (1025, 1025, 12), (200, 77, 2), (257, 640, 3)])
def test_attention_prescaled_q_with_forced_rescales(cuda, Tq, Tk, heads):
    """sculpt_attention_bf16_prescaled (Q carries scale * log2 e; the running maximum is subtracted inside the MFMA and
    may lag by up to 2^10) against an fp64 softmax over the SAME bf16 operands.  The rescale branch is rare on random data,
    so it is forced: single keys are spiked against single queries so that the row maximum jumps by far more than the
    threshold at chosen tiles (first, middle, last, and in the ragged tail), one query sees a huge NEGATIVE first tile, and
    the un-spiked rows keep exercising the lagging-maximum path."""
    from sculptmate_amd import ops

    D = heads * 64
    g = torch.Generator().manual_seed(Tq * 7 + Tk)
    c = 0.125 * 1.4426950408889634
    qf = torch.randn(Tq, D, generator=g)
    kf = torch.randn(Tk, D, generator=g)
    vf = torch.randn(Tk, D, generator=g)
    # spikes: key j aligned with query i in head hh -> score ~ +-|q|^2 * amp / 8 (hundreds of log2 units above the rest)
    spikes = [(0, min(5, Tk - 1), 0, 30.0), (1, Tk // 2, heads - 1, 25.0), (2 % Tq, Tk - 1, 0, 40.0), (3 % Tq, max(Tk - 70, 0), 1 % heads, 35.0),
              (min(130, Tq - 1), min(200, Tk - 1), 0, 30.0), (Tq - 1, 0, 0, 20.0)]
    for (i, j, hh, amp) in spikes:
        kf[j, hh * 64:(hh + 1) * 64] = amp * qf[i, hh * 64:(hh + 1) * 64] / qf[i, hh * 64:(hh + 1) * 64].norm() * 8.0
    qf[min(7, Tq - 1)] *= 0.0
    qf[min(7, Tq - 1), :64] = -kf[:64, :64].mean(0) * 50.0   # very negative scores on the first tile, head 0
    qs = (qf * c).to(BF).to(cuda)
    k = kf.to(BF).to(cuda)
    vt = torch.zeros(D, ((Tk + 63) // 64) * 64, dtype=BF, device=cuda)
    vt[:, :Tk] = vf.to(BF).t().to(cuda)
    o = torch.empty(Tq, D, dtype=BF, device=cuda)
    ops.attention(qs, k, vt, o, Tq, Tk, heads, None)
    qh = qs.double().cpu().view(Tq, heads, 64).transpose(0, 1)
    kh = k.double().cpu().view(Tk, heads, 64).transpose(0, 1)
    vh = vt[:, :Tk].t().double().cpu().view(Tk, heads, 64).transpose(0, 1)
    ref = (torch.softmax(qh @ kh.transpose(1, 2) * math.log(2.0), -1) @ vh).transpose(0, 1).reshape(Tq, D)
    got = o.double().cpu()
    assert torch.isfinite(got).all()
    err = (got - ref).abs()
    assert float(err.max()) < 0.05, float(err.max())                # |O| <= max |v| ~ 4.5: bf16 rounding of p and of O
    assert float((got - ref).norm() / ref.norm()) < 4e-3
    # the spiked rows are (almost) one-hot: O = v[j]
    for (i, j, hh, amp) in spikes[:3]:
        assert float((got[i, hh * 64:(hh + 1) * 64] - vf[j, hh * 64:(hh + 1) * 64].to(BF).double()).abs().max()) < 0.05
    # and the same rows as the in-kernel-scale entry, within bf16 noise
    o2 = torch.empty_like(o)
    ops.attention((qf).to(BF).to(cuda), k, vt, o2, Tq, Tk, heads, 0.125)
    assert float((o2.double().cpu() - ref).norm() / ref.norm()) < 8e-3


@pytest.mark.parametrize("Tq,heads", [(3072, 16), (1025, 12), (130, 2)])
def test_attention_pipelined_loop_against_the_phase_separated_loop(cuda, monkeypatch, Tq, heads):
    """attention_pipe_kernel (pre-scaled queries: softmax of tile t in the MFMA shadows of the scores of t + 1 and of P.V of t, K
    rows read in a permuted order, one score tile computed ahead) against attention_kernel<., true> (SCULPT_ATTN_FORM=nopipe) and the
    fp64 softmax of the same bf16 operands -- at every key count around the tile boundaries: the look-ahead tile is missing,
    ragged, belongs to the other key half only, or is a complete pair."""
    from sculptmate_amd import ops

    D = heads * 64
    g = torch.Generator().manual_seed(Tq + heads)
    c = 0.125 * 1.4426950408889634
    for Tk in (1, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 320, 384, 385, 1025):
        qs = (torch.randn(Tq, D, generator=g) * c).to(BF).to(cuda)
        kf = torch.randn(Tk, D, generator=g)
        kf[Tk - 1, :64] *= 6.0                     # a late maximum jump in head 0: the rescale decided one tile ahead
        k = kf.to(BF).to(cuda)
        vt = torch.zeros(D, ((Tk + 63) // 64) * 64, dtype=BF, device=cuda)
        vt[:, :Tk] = torch.randn(Tk, D, generator=g).to(BF).t().to(cuda)
        outs = []
        for form in ("", "nopipe"):
            monkeypatch.setenv("SCULPT_ATTN_FORM", form)   # read per call
            o = torch.full((Tq, D), float("nan"), dtype=BF, device=cuda)
            ops.attention(qs, k, vt, o, Tq, Tk, heads, None)
            outs.append(o.double().cpu())
        qh = qs.double().cpu().view(Tq, heads, 64).transpose(0, 1)
        kh = k.double().cpu().view(Tk, heads, 64).transpose(0, 1)
        vh = vt[:, :Tk].t().double().cpu().view(Tk, heads, 64).transpose(0, 1)
        ref = (torch.softmax(qh @ kh.transpose(1, 2) * math.log(2.0), -1) @ vh).transpose(0, 1).reshape(Tq, D)
        for o in outs:
            assert torch.isfinite(o).all(), Tk
            assert float((o - ref).norm() / ref.norm()) < 4e-3, Tk
            assert float((o - ref).abs().max()) < 0.05, Tk
        assert float((outs[0] - outs[1]).abs().max()) < 0.04, Tk   # both round p and O to bf16; the row sums differ in order


@pytest.mark.parametrize("M,K,N,epi,split", [(3072, 1024, 4096, 2, 0), (3072, 1024, 3072, 0, 2048), (1025, 768, 2304, 0, 1536),
                                             (300, 128, 512, 1, 0), (517, 64, 256, 0, 0)])
def test_gemm256_tile_kernel_is_bit_identical_to_the_128_row_tiles(cuda, monkeypatch, M, K, N, epi, split):
    """gemm256_kernel (256 x 256 tile, fragment-level pipeline, DMA in flight across barriers by counted vmcnt) accumulates in
    the same k order with the same MFMA and runs the same epilogue arithmetic as gemm_bf16_kernel: every output bit must match --
    LayerNorm fold, GELU / GEGLU, the Q|K / V^T column split, ragged M, K of 1 / 2 / many K-tiles.  Repeated launches screen the
    counted-wait structure for races (a stale tile shows up as a mismatch)."""
    from sculptmate_amd import _lib, ops

    g = torch.Generator().manual_seed(M + K + N)
    rows = 2 * N if epi == 2 else N
    A = torch.randn(M, K, generator=g).to(BF).to(cuda)
    W = (torch.randn(rows, K, generator=g) / math.sqrt(K)).to(BF).to(cuda)
    bias = torch.randn(rows, generator=g).to(cuda)
    cs = W.float().sum(1).contiguous()
    stats = torch.randn(K // 64, M, 2, generator=g).abs().to(cuda) + 0.5
    Mp = (M + 63) // 64 * 64

    def run(tile):
        monkeypatch.setenv("SCULPT_GEMM_TILE", tile)   # read per call: one variable, comma-separated tokens (csrc/common.h)
        if split:
            o = torch.zeros(M, split, dtype=BF, device=cuda)
            ot = torch.zeros(N - split, Mp, dtype=BF, device=cuda)
            ops.gemm(A, W, bias=bias, out_bf16=o, out_t=ot, n_split=split, epilogue=epi, ln_stats=stats, ln_colsum=cs, ln_eps=1e-5)
            return o, ot
        o = torch.zeros(M, N, dtype=BF, device=cuda)
        of = torch.zeros(M, N, device=cuda)
        ops.gemm(A, W, bias=bias, out_bf16=o, out_f32=of, epilogue=epi, ln_stats=stats, ln_colsum=cs, ln_eps=1e-5)
        return o, of

    ref = run("no256")
    assert torch.isfinite(ref[1].float()).all()
    for bm192 in ("no192", "192"):  # 256- and 192-row tiles
        for rep in range(5):
            got = run("256," + bm192)
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), (bm192, rep)


@pytest.mark.parametrize("M,K", [(12288, 1024), (3072, 4096), (960, 128)])
def test_gemm_residual_form_of_the_192_row_tile_kernel(cuda, monkeypatch, M, K):
    """gemm256_kernel<NONE, 192, RES>: h += A W^T + b in place, bf16(h) and the 64-column slice statistics -- against the 128-row
    kernel on the same operands: h and bf16(h) bit for bit (same k order), the statistics to fp32 summation order."""
    from sculptmate_amd import ops

    N = 1024
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, K, generator=g).to(BF).to(cuda)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(cuda)
    b = torch.randn(N, generator=g).to(cuda)
    h0 = (torch.randn(M, N, generator=g) + 2.0 * torch.randn(M, 1, generator=g)).to(cuda)

    def run(flag):
        # ks0: the 128-row side in its k order (the k-split pairs differ by fp32 rounding)
        monkeypatch.setenv("SCULPT_GEMM_TILE", "ks0," + flag)
        h = h0.clone()
        hb = torch.empty(M, N, dtype=BF, device=cuda)
        st = torch.zeros(N // 64, M, 2, device=cuda)
        for _ in range(2):     # twice: in place, so the second launch reads what the first wrote
            ops.gemm(A, W, bias=b, residual=h, out_f32=h, out_bf16=hb, stats_out=st)
        return h, hb, st

    h1, hb1, st1 = run("nores")
    h2, hb2, st2 = run("res")
    assert torch.equal(h1, h2) and torch.equal(hb1, hb2)
    assert _rel(st2, st1)[0] < 1e-6
    ref = h0.double().cpu() + 2 * (A.double().cpu() @ W.double().cpu().t() + b.double().cpu())
    assert _rel(h2, ref.float())[0] < 1e-5
    sl = h2.cpu().view(M, N // 64, 64)
    assert (st2[..., 0].cpu().t() - sl.mean(-1)).abs().max() < 2e-5


@pytest.mark.parametrize("M,K,N,residual", [(3072, 1024, 1024, True), (3072, 4096, 1024, True), (3072, 1024, 1024, False),
                                            (1536, 256, 2048, True), (960, 128, 1024, False)])
def test_gemm_one_round_of_192_row_tiles_is_bit_identical(cuda, monkeypatch, M, K, N, residual):
    """gemm_bf16_kernel<NONE, 64, 8, false, 192> (round 5: M = 3072, N = 1024 as ONE round of 256 tiles of 192 x 64; token ks0: the
    weight-row split) against the 128 x 64 tiles on the same operands: the same k order per output, so h, bf16(h), the slice
    statistics (residual form) or the LayerNorm-folded bf16 output are equal bit for bit, over repeated in-place launches."""
    from sculptmate_amd import ops

    g = torch.Generator().manual_seed(M + K + N)
    A = torch.randn(M, K, generator=g).to(BF).to(cuda)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(cuda)
    b = torch.randn(N, generator=g).to(cuda)
    h0 = (torch.randn(M, N, generator=g) + 2.0 * torch.randn(M, 1, generator=g)).to(cuda)
    stats_in = torch.zeros(K // 64, M, 2, device=cuda)
    stats_in[..., 0] = 0.1 * torch.randn(K // 64, M, device=cuda)
    stats_in[..., 1] = 64.0 + torch.rand(K // 64, M, device=cuda)
    cs = W.float().sum(1).contiguous()

    def run(tile):
        monkeypatch.setenv("SCULPT_GEMM_TILE", tile)
        if residual:
            h = h0.clone()
            hb = torch.empty(M, N, dtype=BF, device=cuda)
            st = torch.zeros(N // 64, M, 2, device=cuda)
            for _ in range(2):
                ops.gemm(A, W, bias=b, residual=h, out_f32=h, out_bf16=hb, stats_out=st)
            return h, hb, st
        o = torch.empty(M, N, dtype=BF, device=cuda)
        ops.gemm(A, W, bias=b, out_bf16=o, ln_stats=stats_in, ln_colsum=cs, ln_eps=1e-5)
        return (o,)

    a = run("nobm192")
    c = run("bm192,ks0")   # 192 x 64 tiles, weight-row split
    for x, y in zip(a, c):
        assert torch.equal(x, y)
    # the shipped form of the 192 x 64 tiles: the two waves of a band split the K-tile instead of the weight rows (30 % fewer LDS
    # fragment bytes) -- the sum is (even k-steps) + (odd k-steps): equal to fp32 rounding, bf16 outputs to one rounding flip
    c = run("bm192")
    assert _rel(c[0].float(), a[0].float())[0] < (1e-6 if residual else 1e-3)   # fp32 h: rounding of two partial sums; bf16: flips
    if residual:
        assert (c[1].float() - a[1].float()).abs().max() <= 2.0 ** -7 * a[1].float().abs().max()
        assert _rel(c[2], a[2])[0] < 1e-5
        ref = h0.double().cpu() + 2 * (A.double().cpu() @ W.double().cpu().t() + b.double().cpu())
        assert _rel(a[0], ref.float())[0] < 1e-5 and _rel(c[0], ref.float())[0] < 1e-5
        for _ in range(2):   # and it is reproducible
            c2 = run("bm192")
            assert all(torch.equal(x, y) for x, y in zip(c, c2))


@pytest.mark.parametrize("M,K,N,epi,split,bm192", [(3072, 1024, 4096, 2, 0, 1), (3072, 1024, 3072, 0, 2048, 1), (12288, 1024, 1024, 0, 0, 1),
                                                     (3000, 256, 512, 0, 0, 1), (3000, 256, 512, 1, 0, 0), (2048, 128, 1024, 0, 512, 0),
                                                     (1032, 768, 2304, 0, 1536, 0), (520, 128, 256, 2, 0, 1),
                                                     (1025, 768, 4096, 0, 2048, 1), (203, 128, 512, 0, 256, 0)])   # M % 8 != 0: ragged V^T pieces
def test_gemm256_staged_stores_equal_direct_stores(cuda, monkeypatch, M, K, N, epi, split, bm192):
    """gemm256_kernel's epilogue through LDS (whole rows, 16 bytes per lane; the V^T part staged transposed) against its direct
    stores from the accumulator layout: the same bits, ragged last row tiles, a column-sliced output buffer, several launches
    (the staging reuses the K loop's LDS ring), and nothing written outside the outputs."""
    from sculptmate_amd import _lib, ops

    g = torch.Generator().manual_seed(M + K + N + epi)
    rows = 2 * N if epi == 2 else N
    A = torch.randn(M, K, generator=g).to(BF).to(cuda)
    W = (torch.randn(rows, K, generator=g) / math.sqrt(K)).to(BF).to(cuda)
    b = torch.randn(rows, generator=g).to(cuda)
    stats = torch.zeros(K // 64, M, 2, device=cuda); stats[..., 1] = 64.0 + torch.rand(K // 64, M, device=cuda)
    cs = W.float().sum(1).contiguous()
    Mp = ((M + 63) // 64) * 64
    def run(stage):
        monkeypatch.setenv("SCULPT_GEMM_TILE", "256,%s%s" % ("192" if bm192 else "no192", "" if stage == "1" else ",nostage"))
        ncol = split if split else N
        buf = torch.full((M + 3, ncol + 8), 7.0, dtype=BF, device=cuda)     # the output is a column slice of a wider buffer
        o = buf[:M, :ncol]
        ot = torch.full((N - split, Mp), 5.0, dtype=BF, device=cuda) if split else None
        for _ in range(3):
            ops.gemm(A, W, bias=b, out_bf16=o, out_t=ot, n_split=split, epilogue=epi, ln_stats=stats, ln_colsum=cs, ln_eps=1e-5)
        return buf.clone(), None if ot is None else ot.clone()

    d, dt = run("0")
    s_, st = run("1")
    assert torch.equal(d, s_)
    assert (s_[M:] == 7.0).all() and (s_[:, (split if split else N):] == 7.0).all()
    if split:
        assert torch.equal(dt, st) and (st[:, M:] == 5.0).all()


def test_bf16_conversion_rounds_to_nearest_even_like_the_integer_rule(cuda):
    """common.h f32_to_bf16 / pack_bf16x2 are v_cvt_pk_bf16_f32 since round 6.  Against the integer rule every kernel used before
    (u + 0x7fff + ((u >> 16) & 1)) >> 16: the same bits for every number -- ties, denormals, the largest finite value rounding to
    infinity, signed zeros, infinities; a NaN stays a NaN with its sign (its payload may differ)."""
    from sculptmate_amd import ops

    g = torch.Generator().manual_seed(77)
    bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (1 << 20,), generator=g, dtype=torch.int64).to(torch.int32)
    special = torch.tensor([0x00000000, -0x80000000, 0x7f800000, -0x00800000, 0x7f7fffff, 0x7f7f8000, 0x7f7f7fff, 0x00000001, 0x00007fff,
                            0x00008000, 0x00008001, 0x00018000, 0x00017fff, 0x3f808000, 0x3f818000, 0x3f807fff, 0x3f808001, 0x007fffff,
                            0x00800000, 0x7fc00000, 0x7f800001, -0x00400000, 0x7fffffff], dtype=torch.int64).to(torch.int32)
    ties = (torch.randint(0, 1 << 16, (4096,), generator=g, dtype=torch.int64) << 16 | 0x8000).to(torch.int32)   # exact ties, every exponent
    u = torch.cat([bits, special, ties]).to(cuda)
    x = u.view(torch.float32)
    y = torch.empty(x.numel(), dtype=torch.int16, device=cuda)
    ops.cast_bf16(x, y)
    got = y.cpu().to(torch.int64) & 0xffff
    uu = u.cpu().to(torch.int64) & 0xffffffff
    want = ((uu + 0x7fff + ((uu >> 16) & 1)) >> 16) & 0xffff
    nan = (uu & 0x7fffffff) > 0x7f800000
    assert torch.equal(got[~nan], want[~nan])
    assert nan.sum() > 1000 and (((got[nan] & 0x7fff) > 0x7f80).all()) and torch.equal(got[nan] >> 15, (uu[nan] >> 31) & 1)
    assert torch.equal(got[~nan], (x.cpu().to(BF).view(torch.int16).to(torch.int64) & 0xffff)[~nan])   # = torch's own conversion


def test_gemm_gelu_and_geglu_epilogues(cuda):
    from sculptmate_amd import _lib, ops

    g = torch.Generator().manual_seed(5)
    M, N, K = 200, 256, 128
    A = torch.randn(M, K, generator=g).to(BF)
    W = (torch.randn(2 * N, K, generator=g) / math.sqrt(K)).to(BF)
    b = torch.randn(2 * N, generator=g)
    pre = A.float() @ W.float().t() + b
    out = torch.empty(M, 2 * N, device=cuda)
    ops.gemm(A.to(cuda), W.to(cuda), bias=b.to(cuda), out_f32=out, epilogue=_lib.EPI_GELU)
    assert _rel(out, torch.nn.functional.gelu(pre))[1] < 2e-5
    o2 = torch.empty(M, N, device=cuda)
    o2b = torch.empty(M, N, dtype=BF, device=cuda)
    ops.gemm(A.to(cuda), W.to(cuda), bias=b.to(cuda), out_f32=o2, out_bf16=o2b, epilogue=_lib.EPI_GEGLU)
    ref = pre[:, :N] * torch.nn.functional.gelu(pre[:, N:])  # chunk(2): first half value, second gate
    assert _rel(o2, ref)[1] < 3e-5
    assert _rel(o2b, ref)[0] < 4e-3


@pytest.mark.parametrize("Tq,Tk,heads", [(128, 64, 1), (3072, 3072, 16), (3072, 1025, 16), (1025, 1025, 12), (37, 5, 2)])
def test_attention_vs_fp32_reference(cuda, Tq, Tk, heads):
    from sculptmate_amd import ops

    g = torch.Generator().manual_seed(Tq + Tk)
    D = heads * 64
    q = torch.randn(Tq, D, generator=g).to(BF)
    k = torch.randn(Tk, D, generator=g).to(BF)
    v = torch.randn(Tk, D, generator=g).to(BF)
    Tkp = ((Tk + 63) // 64) * 64
    vt = torch.zeros(D, Tkp, dtype=BF)
    vt[:, :Tk] = v.t()
    o = torch.empty(Tq, D, dtype=BF, device=cuda)
    ops.attention(q.to(cuda), k.to(cuda), vt.to(cuda), o, Tq, Tk, heads, 0.125)
    qh = q.float().view(Tq, heads, 64).transpose(0, 1)
    kh = k.float().view(Tk, heads, 64).transpose(0, 1)
    vh = v.float().view(Tk, heads, 64).transpose(0, 1)
    ref = (torch.softmax(qh @ kh.transpose(1, 2) * 0.125, -1) @ vh).transpose(0, 1).reshape(Tq, D)
    rel, mx = _rel(o, ref)
    # bf16 probabilities and bf16 output: ~2^-8 relative per element, averaged down by the sum
    assert rel < 6e-3 and mx < 0.06, (rel, mx)


@pytest.mark.parametrize("Tq,Tk,heads,B,prescaled", [(3072, 3072, 16, 3, True), (3072, 1025, 16, 4, True), (1025, 1025, 12, 2, True),
                                                       (1025, 1025, 12, 5, False), (200, 77, 2, 3, True), (64, 130, 1, 7, False)])
def test_batched_attention_equals_per_entry_launches(cuda, Tq, Tk, heads, B, prescaled):
    """sculpt_attention_bf16_batched: B independent attentions in one launch (entries stacked row-wise, V^T side by side at a
    column stride that is a multiple of 8 but not of 64) == B single launches on the same operands, bit for bit where the
    launcher picks the same workgroup shape, and always within the kernel's own error against fp32."""
    from sculptmate_amd import ops

    g = torch.Generator().manual_seed(Tq + Tk + B)
    D = heads * 64
    Tqs, Tks = ((Tq + 7) // 8) * 8, ((Tk + 7) // 8) * 8           # row stride between entries (image_tokens' stacking)
    ldv = (((B - 1) * Tks + ((Tk + 63) // 64) * 64 + 63) // 64) * 64
    q = torch.randn(B * Tqs, D, generator=g).to(BF).to(cuda)
    k = torch.randn(B * Tks, D, generator=g).to(BF).to(cuda)
    v = torch.randn(B * Tks, D, generator=g).to(BF).to(cuda)
    vt = torch.zeros(D, ldv, dtype=BF, device=cuda)
    vt[:, :B * Tks] = v.t()
    scale = None if prescaled else 0.125
    o = torch.zeros(B * Tqs, D, dtype=BF, device=cuda)
    ops.attention(q, k, vt, o, Tq, Tk, heads, scale, batch=B, q_bs=Tqs * D, k_bs=Tks * D, vt_bs=Tks, o_bs=Tqs * D)
    same = True
    for b in range(B):
        ob = torch.zeros(Tq, D, dtype=BF, device=cuda)
        ops.attention(q[b * Tqs:], k[b * Tks:], vt[:, b * Tks:], ob, Tq, Tk, heads, scale)
        got = o[b * Tqs:b * Tqs + Tq]
        same = same and torch.equal(got, ob)
        qh = q[b * Tqs:b * Tqs + Tq].float().view(Tq, heads, 64).transpose(0, 1)
        kh = k[b * Tks:b * Tks + Tk].float().view(Tk, heads, 64).transpose(0, 1)
        vh = v[b * Tks:b * Tks + Tk].float().view(Tk, heads, 64).transpose(0, 1)
        sc = qh @ kh.transpose(1, 2)
        p = torch.softmax(sc * math.log(2.0), -1) if prescaled else torch.softmax(sc * 0.125, -1)
        ref = (p @ vh).transpose(0, 1).reshape(Tq, D)
        rel, mx = _rel(got, ref)
        assert rel < 6e-3 and mx < 0.08, (b, rel, mx)
        assert _rel(got, ob)[0] < 2e-3
        assert (o[b * Tqs + Tq:(b + 1) * Tqs] == 0).all()        # pad rows between entries are not written
    print("batched == per-entry launches bit for bit:", same)


def test_attention_forced_rescale_branch(cuda):
    """One key spikes late in the sequence so the running max jumps in a late tile (online softmax)."""
    from sculptmate_amd import ops

    Tq, Tk, D = 128, 512, 64
    g = torch.Generator().manual_seed(0)
    q = torch.randn(Tq, D, generator=g).to(BF)
    k = (0.1 * torch.randn(Tk, D, generator=g))
    k[400] = 4.0 * q[7].float()  # huge score for query 7 at key 400 (tile 6)
    k = k.to(BF)
    v = torch.randn(Tk, D, generator=g).to(BF)
    o = torch.empty(Tq, D, dtype=BF, device=cuda)
    ops.attention(q.to(cuda), k.to(cuda), v.t().contiguous().to(cuda), o, Tq, Tk, 1, 0.125)
    ref = torch.softmax(q.float() @ k.float().t() * 0.125, -1) @ v.float()
    assert _rel(o, ref)[1] < 0.06


@pytest.mark.parametrize("cols,eps", [(768, 1e-12), (1024, 1e-5), (256, 1e-5)])
def test_layernorm(cuda, cols, eps):
    from sculptmate_amd import ops

    g = torch.Generator().manual_seed(cols)
    x = torch.randn(1025, cols, generator=g) * 3 + 0.5
    w, b = torch.randn(cols, generator=g), torch.randn(cols, generator=g)
    ref = torch.nn.functional.layer_norm(x, (cols,), w, b, eps)
    y = torch.empty(1025, cols, dtype=BF, device=cuda)
    y32 = torch.empty(1025, cols, device=cuda)
    ops.layernorm(x.to(cuda), w.to(cuda), b.to(cuda), eps, y=y, y_f32=y32)
    assert _rel(y32, ref)[1] < 2e-5
    assert _rel(y, ref)[0] < 4e-3
    yb = torch.empty(1025, cols, dtype=BF, device=cuda)
    ops.layernorm(x.to(BF).to(cuda), w.to(cuda), b.to(cuda), eps, y=yb)
    assert _rel(yb, torch.nn.functional.layer_norm(x.to(BF).float(), (cols,), w, b, eps))[0] < 4e-3


def test_groupnorm_tokens_and_transpose_add(cuda):
    from sculptmate_amd import ops

    g = torch.Generator().manual_seed(3)
    C, T, G = 1024, 3072, 32
    x = torch.randn(C, T, generator=g) * 2 + 1
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ref = torch.nn.functional.group_norm(x[None], G, w, b, 1e-6)[0].t()
    y = torch.empty(T, C, dtype=BF, device=cuda)
    st = torch.empty(2 * G, device=cuda)
    ops.groupnorm_tokens(x.to(cuda), G, w.to(cuda), b.to(cuda), 1e-6, y, st)
    assert _rel(y, ref)[0] < 4e-3
    xt = torch.randn(T, C, generator=g)
    out = torch.empty(C, T, device=cuda)
    ops.transpose_add(xt.to(cuda), x.to(cuda), out)
    assert torch.equal(out.cpu(), xt.t() + x)


def test_upsample_vs_reference_golden(cuda):
    """TriplaneUpsampleNetwork through the GEMM + scatter kernels vs the reference's own output (G6)."""
    from sculptmate_amd import ops

    g = np.load(os.path.join(GOLDEN, "upsample.npz"))
    rng = np.random.default_rng([7, 15])
    w = synth._uniform(rng, (1024, 40, 2, 2), 1.0 / np.sqrt(160.0))
    b = synth._uniform(rng, (40,), 1.0 / np.sqrt(160.0))
    x = np.random.default_rng(8).standard_normal((1, 3, 1024, 32, 32), dtype=np.float32)
    tokens = torch.from_numpy(x[0]).permute(0, 2, 3, 1).reshape(3072, 1024).to(BF).to(cuda)  # [t][c]
    upw = torch.zeros(256, 1024)
    upw[:160] = torch.from_numpy(w).permute(1, 2, 3, 0).reshape(160, 1024)
    gg = torch.empty(3072, 256, device=cuda)
    ops.gemm(tokens, upw.to(BF).to(cuda), out_f32=gg)
    planes = torch.empty(3, 40, 64, 64, device=cuda)
    ops.upsample_scatter(gg, torch.from_numpy(b).to(cuda), planes, 32, 40)
    got = planes.cpu().numpy().reshape(-1)[g["idx"]]
    # inputs and weights rounded to bf16 (K=1024): documented bf16 tolerance
    assert np.abs(got - g["y"]).max() < 0.05 and np.linalg.norm(got - g["y"]) / np.linalg.norm(g["y"]) < 6e-3


def _small_model(cuda, seed=31):
    from sculptmate_amd.tsr import TSR

    sd = synth.tsr_state(seed, SMALL_CFG)
    m = TSR(SMALL_CFG, pos_embed_mode="size")
    m.load_state_dict(sd)
    m.to(cuda)
    return m, sd


def test_small_tsr_forward_vs_oracle(cuda):
    """Whole TSR.forward (ViT + backbone + upsample) on a small kernel-compatible model."""
    m, sd = _small_model(cuda)
    S = SMALL_CFG["cond_image_size"]
    img = synth.composite_rgb(synth.image_rgba(seed=32, size=S))
    codes = m([img], device=cuda)
    assert codes.shape == (1, 3, 40, 16, 16) and codes.dtype == torch.float32
    ref32 = tsr_ref.tsr_forward(sd, img, SMALL_CFG, pos_mode="size")
    refbf = tsr_ref.tsr_forward(sd, img, SMALL_CFG, pos_mode="size", bf16=True)
    r_bf, _ = _rel(codes[0], refbf)
    r_32, _ = _rel(codes[0], ref32)
    # vs the oracle with the same bf16 storage points: only accumulation order / exp differences
    assert r_bf < 8e-3, r_bf
    # vs the fp32 oracle: the bf16 tolerance of BASELINE config 2 (bf16 transformer): 2 % of the norm
    assert r_32 < 2e-2, r_32


def test_small_tsr_intermediates_vs_oracle(cuda):
    m, sd = _small_model(cuda)
    S = SMALL_CFG["cond_image_size"]
    img = synth.composite_rgb(synth.image_rgba(seed=33, size=S))
    col = {}
    tsr_ref.tsr_forward(sd, img, SMALL_CFG, pos_mode="size", bf16=True, collect=col)
    ctx, ctx32 = m.image_tokens(torch.from_numpy(img).to(cuda))
    assert _rel(ctx32, col["ctx"])[0] < 5e-3
    out, _ = m.backbone_tokens(ctx)
    assert _rel(out.t(), col["tokens"])[0] < 8e-3


def test_full_size_block_vs_reference_golden(cuda):
    """One full-size backbone block (3072 x 1024, ctx 1025 x 768) through the HIP kernels vs the
    reference's own BasicTransformerBlock output (G3), bf16 tolerance."""
    from sculptmate_amd.tsr import TSR

    g = np.load(os.path.join(GOLDEN, "tsr_block.npz"))
    cfg = make_cfg(vit_layers=1, layers=1)
    sd = synth.tsr_state(seed=23, cfg=cfg)
    m = TSR(cfg, pos_embed_mode="size")
    m.load_state_dict(sd)
    m.to(cuda)
    h = torch.from_numpy(np.random.default_rng(24).standard_normal((3072, 1024), dtype=np.float32)).to(cuda)
    ctx = torch.from_numpy(np.random.default_rng(25).standard_normal((1025, 768), dtype=np.float32)).to(BF).to(cuda)
    y = m._run_blocks(m._state_from(h.clone()), ctx)["h"].cpu().numpy()
    got = y.reshape(-1)[g["idx"]]
    rel = np.linalg.norm(got - g["y"]) / np.linalg.norm(g["y"])
    assert rel < 1e-2, rel


def test_end_to_end_run_returns_meshes(cuda):
    """TSR.run: image -> mesh, and the mesh equals oracle marching cubes of the GPU density grid."""
    from oracle import capi
    from sculptmate_amd import ops

    m, sd = _small_model(cuda, seed=41)
    S = SMALL_CFG["cond_image_size"]
    img = synth.composite_rgb(synth.image_rgba(seed=42, size=S))
    codes = m([img], device=cuda)
    R = 48
    dens = ops.density_grid(codes[0].contiguous(), m.decoder, R, precision=m.decoder_precision)  # the mode m.run uses
    # random weights never reach the threshold 25: use the median density as the iso level
    thr = float(dens.median())
    meshes = m.run([img], mc_resolution=R, threshold=thr, enable_texture=True)
    assert len(meshes) == 1
    v, f, c = meshes[0].vertices, meshes[0].faces, meshes[0].vertex_colors
    assert v.dtype == np.float32 and f.dtype == np.int64 and c.shape == (len(v), 3)
    assert np.abs(v).max() <= 0.87 + 1e-6 and f.max() == len(v) - 1 and (c >= 0).all() and (c <= 1).all()
    vol = (dens - thr).view(R, R, R).cpu().numpy()
    rv, rf = capi.reference_isosurface(-vol, R)
    rv = rv * np.float32(0.87 - (-0.87)) + np.float32(-0.87)
    assert np.array_equal(f, rf)
    assert np.array_equal(v.view(np.uint32), rv.astype(np.float32).view(np.uint32))


def test_batch_of_images_equals_one_at_a_time(cuda):
    """BASELINE config 3 per GPU: TSR.forward / TSR.run on a LIST of images.  forward() runs the list as ONE batched pass like
    the reference (system.py:82-115: every Linear over the stacked token rows, attention over batch x heads).  Both give, image
    by image, what single-image calls give: the batched scene codes bit for bit, the meshes of run() bit for bit."""
    m, sd = _small_model(cuda, seed=43)
    S = SMALL_CFG["cond_image_size"]
    imgs = [synth.composite_rgb(synth.image_rgba(seed=60 + i, size=S)) for i in range(3)]
    codes = m(imgs, device=cuda)
    assert codes.shape == (3, 3, 40, 16, 16)
    m.max_batch = 1
    serial = m(imgs, device=cuda)
    m.max_batch = 8
    assert serial.shape == codes.shape
    from sculptmate_amd import ops

    thr = float(ops.density_grid(codes[0].contiguous(), m.decoder, 32).median())
    batch = m.run(imgs, mc_resolution=32, threshold=thr)
    assert len(batch) == 3
    for i, im in enumerate(imgs):
        one = m([im], device=cuda)
        assert torch.equal(one[0], serial[i])
        assert torch.equal(one[0], codes[i]), _rel(codes[i], one[0])
        single = m.run([im], mc_resolution=32, threshold=thr)[0]
        assert np.array_equal(single.vertices.view(np.uint32), batch[i].vertices.view(np.uint32))
        assert np.array_equal(single.faces, batch[i].faces)
    assert not np.array_equal(batch[0].vertices[:50], batch[1].vertices[:50])  # different images, different meshes
    # a batch larger than max_batch is cut into passes; a ragged last pass and a pass of one
    m.max_batch = 2
    assert torch.equal(m(imgs, device=cuda), codes)
    m.max_batch = 8
    # TSR.run(images, batch=2): batched transformer passes behind the headless entry point (on this small model the batched
    # scene codes are bit-identical, so the meshes are too); max_batch is restored afterwards
    for got, want in zip(m.run(imgs, mc_resolution=32, threshold=thr, batch=2), batch):
        assert np.array_equal(got.vertices.view(np.uint32), want.vertices.view(np.uint32)) and np.array_equal(got.faces, want.faces)
    assert m.max_batch == 8


def test_run_batches_by_default_and_returns_the_serial_meshes(cuda):
    """TSR.run on a list (the entry north_star names): in the bf16 mode it stacks RUN_BATCH images per transformer pass by
    default -- each image gets the scene code of its own pass bit for bit, so the meshes are those of one-at-a-time calls -- over
    a list that is not a multiple of four; batch=1 is the tokenizer look-ahead path of round 5; the limb modes stay on it."""
    from sculptmate_amd import ops
    from sculptmate_amd.tsr import TSR

    m, sd = _small_model(cuda, seed=47)
    S = SMALL_CFG["cond_image_size"]
    imgs = [synth.composite_rgb(synth.image_rgba(seed=70 + i, size=S)) for i in range(6)]
    m.RUN_BATCH = 4          # two passes over the six images: four, then a ragged one of two
    thr = float(ops.density_grid(m([imgs[0]], device=cuda)[0].contiguous(), m.decoder, 32).median())
    calls = []
    rb, rp = m.run_batched, m.run_pipelined
    m.run_batched = lambda *a, **k: (calls.append(("batched", a[1])), rb(*a, **k))[1]
    m.run_pipelined = lambda *a, **k: (calls.append(("pipelined", None)), rp(*a, **k))[1]
    got = m.run(imgs, mc_resolution=32, threshold=thr)
    assert calls == [("batched", 4)] and len(got) == 6
    serial = m.run(imgs, mc_resolution=32, threshold=thr, batch=1)
    assert calls[-1] == ("pipelined", None)
    for a, b, im in zip(got, serial, imgs):
        one = m.run([im], mc_resolution=32, threshold=thr)[0]
        for x in (a, b):
            assert np.array_equal(x.vertices.view(np.uint32), one.vertices.view(np.uint32)) and np.array_equal(x.faces, one.faces)
    m3 = TSR(SMALL_CFG, pos_embed_mode="size", precision="bf16l3")
    m3.load_state_dict(sd)
    m3.to(cuda)
    calls3 = []
    rp3 = m3.run_pipelined
    m3.run_pipelined = lambda *a, **k: (calls3.append("pipelined"), rp3(*a, **k))[1]
    assert len(m3.run(imgs[:2], mc_resolution=32, threshold=thr)) == 2 and calls3 == ["pipelined"]


def test_full_size_batched_forward_equals_single_image_passes(cuda):
    """The full-size model: TSR.forward on B = 3, 4 and 7 images in one batched pass (max_batch = B: M = B x 3072 / B x 1032 stacked
    token rows, attention over B x 16 heads per launch) against the single-image passes: bit-identical scene codes.
    Round 6: every GEMM of the stacked pass takes the tile form a single image takes (sculpt_ln_fold_t::rows_per_image), so the
    k-split pairs of the N = 1024 launches, the half-slice merges of their LayerNorm statistics and the upsampler's k order are
    the single-image ones (rounds 4-5: 2.7e-3 apart -- a bf16 rounding flipped now and then and sixteen blocks amplified it; with
    only the backbone hinted, B = 3 and 4 still differed by 2e-7: the upsampler's GEMM took the k-split tile at exactly those)."""
    from sculptmate_amd.tsr import TSR

    sd = synth.tsr_state(seed=0)
    m = TSR(pos_embed_mode="scale_factor")
    m.load_state_dict(sd)
    m.to(cuda)
    imgs = [torch.from_numpy(synth.composite_rgb(synth.image_rgba(seed=100 + i))).to(cuda) for i in range(7)]
    with torch.no_grad():
        assert m.max_batch == 1
        singles = [m([im], device=cuda)[0].clone() for im in imgs]
        for B in (3, 4, 7):
            m.max_batch = B
            codes = m(imgs[:B], device=cuda).clone()
            again = m(imgs[:B], device=cuda)
            assert codes.shape == (B, 3, 40, 64, 64) and torch.isfinite(codes).all()
            assert torch.equal(again, codes)                       # deterministic, buffers reused
            for i in range(B):
                assert torch.equal(codes[i], singles[i]), (B, i, _rel(codes[i], singles[i]))
    assert _rel(singles[0], singles[1])[0] > 1e-2              # different images


def test_tokenizer_lookahead_gives_the_serial_meshes(cuda):
    """TSR.run on several images queues the image tokenizer of image i + 1 on a second stream beside the backbone / density grid
    / marching cubes of image i (tokens_async, two token slots).  Same kernels on the same operands: every mesh must equal the
    one-image-at-a-time result bit for bit -- over more images than slots, in a different order, with serial forward() calls mixed
    in (they share the tokenizer's work buffers), and for token sets used out of order."""
    m, sd = _small_model(cuda, seed=45)
    S = SMALL_CFG["cond_image_size"]
    imgs = [synth.composite_rgb(synth.image_rgba(seed=90 + i, size=S)) for i in range(7)]
    from sculptmate_amd import ops

    thr = float(ops.density_grid(m([imgs[0]], device=cuda)[0].contiguous(), m.decoder, 32).median())
    want = []
    for im in imgs:
        w = m.run([im], mc_resolution=32, threshold=thr, enable_texture=True)[0]   # one image: no lookahead
        want.append((w.vertices.copy(), w.faces.copy(), w.vertex_colors.copy()))
    assert len({w[0].shape for w in want}) > 1                                      # the images do give different meshes

    def same(got, idx):
        for g, i in zip(got, idx):
            assert np.array_equal(want[i][0], g.vertices) and np.array_equal(want[i][1], g.faces) and np.array_equal(want[i][2], g.vertex_colors), i

    same(m.run(imgs, mc_resolution=32, threshold=thr, enable_texture=True, batch=1), range(7))   # batch=1: the look-ahead path
    order = [5, 2, 6, 0, 3]
    same(m.run([imgs[i] for i in order], mc_resolution=32, threshold=thr, enable_texture=True, batch=1), order)
    # serial calls between pipelined ones, and a scene code computed the serial way while tokens are in flight
    t3 = m.tokens_async(imgs[3])
    codes_serial = m([imgs[4]], device=cuda)
    t1 = m.tokens_async(imgs[1])
    got1 = m.run_async(imgs[1], 32, thr, enable_texture=True, tokens=t1).result()   # the younger token set first
    got3 = m.run_async(imgs[3], 32, thr, enable_texture=True, tokens=t3).result()
    same([got1, got3], [1, 3])
    mesh4 = m.extract_meshes(codes_serial, True, 32, thr)[0]
    assert np.array_equal(mesh4.vertices.cpu().numpy(), want[4][0]) and np.array_equal(mesh4.faces.cpu().numpy(), want[4][1])
    # device-resident images take the same path
    dev_imgs = [torch.from_numpy(im).to(cuda) for im in imgs[:3]]
    same(m.run(dev_imgs, mc_resolution=32, threshold=thr, enable_texture=True, batch=1), range(3))
    same(m.run(dev_imgs, mc_resolution=32, threshold=thr, enable_texture=True), range(3))             # ... and the stacked default


def test_run_async_pipeline_and_pinned_buffer_lifetime(cuda):
    """TSR.run_async: several images in flight, results collected later == the one-at-a-time results; and the pinned host
    buffers behind the returned arrays are recycled only when the ARRAYS are gone -- an array kept after its Mesh object was
    dropped must survive any number of later runs unchanged."""
    import gc

    m, sd = _small_model(cuda, seed=44)
    S = SMALL_CFG["cond_image_size"]
    imgs = [synth.composite_rgb(synth.image_rgba(seed=70 + i, size=S)) for i in range(4)]
    from sculptmate_amd import ops

    thr = float(ops.density_grid(m([imgs[0]], device=cuda)[0].contiguous(), m.decoder, 32).median())
    want = [m.run([im], mc_resolution=32, threshold=thr, enable_texture=True)[0] for im in imgs]
    want = [(w.vertices.copy(), w.faces.copy(), w.vertex_colors.copy()) for w in want]
    pend = [m.run_async(im, 32, thr, enable_texture=True) for im in imgs]          # nothing waited for in between
    got = [p.result() for p in reversed(pend)][::-1]
    for (wv, wf, wc), g in zip(want, got):
        assert np.array_equal(wv, g.vertices) and np.array_equal(wf, g.faces) and np.array_equal(wc, g.vertex_colors)
    # keep ONE array, drop everything else, run many more images through the same pool
    keep = m.run([imgs[1]], mc_resolution=32, threshold=thr)[0].vertices          # the Mesh object dies right here
    snapshot = keep.copy()
    del got, pend
    gc.collect()
    for rep in range(6):
        for im in imgs:
            out = m.run([im], mc_resolution=32, threshold=thr, enable_texture=True)
            del out
        gc.collect()
    assert np.array_equal(keep, snapshot) and np.array_equal(keep, want[1][0])
    pool = m._pin_pool
    n_free = sum(len(v) for v in pool.free.values())
    assert 1 <= n_free <= 16, n_free  # buffers do come back (no leak): at most the 4 x 3 that were in flight at once, plus the kept ones
    view = keep[:10]
    del keep
    gc.collect()
    m.run([imgs[2]], mc_resolution=32, threshold=thr)
    assert np.array_equal(view, snapshot[:10])  # a slice keeps its buffer out of the pool just as well
    # result() twice hands out the SAME Mesh (a second set of views would not own its buffers), and it survives later runs
    p = m.run_async(imgs[3], 32, thr)
    r1 = p.result()
    r2 = p.result()
    assert r1 is r2
    snap = r1.vertices.copy()
    del r1
    gc.collect()
    for im in imgs:
        m.run([im], mc_resolution=32, threshold=thr)
    assert np.array_equal(r2.vertices, snap)
    # the pool hands out zero-element arrays of any dtype (a mesh without colours / an empty slab)
    from sculptmate_amd.tsr.system import _PinnedPool

    lease, h = _PinnedPool().take((0, 3), torch.int64)
    assert h.shape == (0, 3) and h.dtype == torch.int64 and h.is_pinned()
    # a pageable host image goes through the pinned staging ring: more images than slots, results unchanged
    more = [m.run_async(imgs[i % 4], 32, thr) for i in range(7)]
    for i, q in enumerate(more):
        assert np.array_equal(q.result().vertices, want[i % 4][0])


@pytest.mark.parametrize("precision", ["fp32", "bf16l3", "fp16l2", "bf16"])
def test_full_size_forward_is_reproducible_call_after_call(cuda, precision, monkeypatch):
    """The same image through the same model gives the same bits on every call, in every precision mode -- with the attention
    score scratch capped so that the backbone's heads go through it in chunks (round 5: the exact-fp32 mode's scratch is one
    buffer PER STREAM; shared between the image tokenizer and the backbone head, which run on two streams, it raced)."""
    from sculptmate_amd.engine import KernelEngine
    from sculptmate_amd.tsr import TSR

    monkeypatch.setattr(KernelEngine, "ATTN_SCRATCH_BYTES", 256 << 20)
    junk = [torch.full((128 << 20,), float("nan"), device=cuda) for _ in range(8)]   # what torch.empty() hands out next is NaN
    del junk
    m = TSR(pos_embed_mode="scale_factor", precision=precision)
    m.load_state_dict(synth.tsr_state(seed=0))
    m.to(cuda)
    img = torch.from_numpy(synth.composite_rgb(synth.image_rgba(seed=100))).to(cuda)
    with torch.no_grad():
        codes = [m([img], device=cuda)[0].clone() for _ in range(3)]
    assert torch.isfinite(codes[0]).all()
    assert torch.equal(codes[0], codes[1]) and torch.equal(codes[0], codes[2])


def test_generator_facade_end_to_end(cuda, tmp_path):
    """TripoGenerator (the add-on's entry point): initiate_model -> generate_mesh, return codes 0,
    meshes delivered to the sink with the reference's array types (system.py:200)."""
    from test_host_logic import _write_checkpoint

    from sculptmate_amd.generate import TripoGenerator

    _write_checkpoint(str(tmp_path), SMALL_CFG, seed=61)
    g = TripoGenerator(cuda)
    g.checkpoint_dir = str(tmp_path)
    g.mc_resolution = 32
    assert g.initiate_model() == 0
    assert g.initiate_model() is None  # already loaded (generate.py:18)
    assert g.model.renderer.chunk_size == 8192
    got = []
    g.model.mesh_sink = lambda v, f, c, name: got.append((v, f, c, name))
    img = (synth.composite_rgb(synth.image_rgba(seed=62, size=SMALL_CFG["cond_image_size"])) * 255).astype(np.uint8)
    # random weights never reach the default threshold 25 -> skimage raises -> facade returns 2, like the reference
    assert g.generate_mesh(img, "probe") == 2
    # with a reachable threshold the mesh arrives
    import types

    orig = g.model.extract_mesh
    g.model.extract_mesh = types.MethodType(
        lambda self, codes, **kw: orig(codes, **dict(kw, threshold=0.0 + float(_median_density(self, codes, kw["resolution"])))), g.model)
    assert g.generate_mesh(img, "mesh0", enable_texture=True) == 0
    v, f, c, name = got[-1]
    assert name == "mesh0" and v.dtype == np.float32 and f.dtype == np.int64 and c.shape == (len(v), 3)


def test_generator_facade_in_the_tolerance_mode(cuda, tmp_path):
    """the mode that meets north_star's 1e-4 (precision="bf16l3") is selectable at the add-on's own entry
    point -- TripoGenerator.precision -> TSR.from_pretrained(..., precision=) -- and delivers the fp32 model's mesh: same
    topology, vertices within 1e-4 of the extent of TSR(precision="fp32") on the same checkpoint."""
    import types

    from _meshcmp import assert_mesh_close
    from test_host_logic import _write_checkpoint

    from sculptmate_amd.generate import TripoGenerator

    _write_checkpoint(str(tmp_path), SMALL_CFG, seed=61)
    img = (synth.composite_rgb(synth.image_rgba(seed=62, size=SMALL_CFG["cond_image_size"])) * 255).astype(np.uint8)
    meshes = {}
    for prec in ("bf16l3", "fp16l2", "fp32"):
        g = TripoGenerator(cuda)
        assert g.precision == "bf16"
        g.precision = prec
        g.checkpoint_dir = str(tmp_path)
        g.mc_resolution = 32
        assert g.initiate_model() == 0 and g.model.precision == prec
        got = []
        g.model.mesh_sink = lambda v, f, c, name, got=got: got.append((v, f))
        orig = g.model.extract_mesh
        if prec == "bf16l3":
            thr = float(_median_density(g.model, g.model([img], device=cuda), 32))
        g.model.extract_mesh = types.MethodType(lambda self, codes, orig=orig, **kw: orig(codes, **dict(kw, threshold=thr)), g.model)
        assert g.generate_mesh(img, "m") == 0
        meshes[prec] = got[-1]
    assert_mesh_close(meshes["bf16l3"][0], meshes["bf16l3"][1], meshes["fp32"][0], meshes["fp32"][1], tol=1e-4 * 1.74)
    # the faster tolerance mode (two fp16 limbs in the Linears, round 5): the same bound
    assert_mesh_close(meshes["fp16l2"][0], meshes["fp16l2"][1], meshes["fp32"][0], meshes["fp32"][1], tol=1e-4 * 1.74)


def _median_density(model, codes, R):
    from sculptmate_amd import ops

    return ops.density_grid(codes[0].contiguous(), model.decoder, R).median()


def test_gpu_preprocessor_resize_matches_reference(cuda):
    """ImagePreprocessor 1024^2 -> 512^2 (antialiased bilinear) on the GPU vs the reference's own output (G1)."""
    from sculptmate_amd import ops

    g = np.load(os.path.join(GOLDEN, "preproc.npz"))
    img = synth.composite_rgb(synth.image_rgba(seed=27, size=1024))
    y = ops.resize_aa_bilinear(torch.from_numpy(img).to(cuda), 512).cpu().numpy()
    assert y.shape == (512, 512, 3)
    np.testing.assert_allclose(y.reshape(-1)[g["idx"]], g["y"], rtol=0, atol=2e-6)
    # generic ratios vs torch on the host
    for (h, w, s) in ((300, 300, 128), (200, 200, 256)):
        x = torch.rand(h, w, 3)
        ref = torch.nn.functional.interpolate(x.permute(2, 0, 1)[None], (s, s), mode="bilinear", align_corners=False,
                                              antialias=True)[0].permute(1, 2, 0)
        got = ops.resize_aa_bilinear(x.to(cuda), s).cpu()
        assert (got - ref).abs().max() < 3e-6


def test_forward_accepts_large_uint8_images(cuda):
    m, sd = _small_model(cuda)
    S = SMALL_CFG["cond_image_size"]
    big = (synth.composite_rgb(synth.image_rgba(seed=35, size=2 * S)) * 255).astype(np.uint8)
    codes = m([big, big], device=cuda)
    assert codes.shape[0] == 2 and torch.equal(codes[0], codes[1])
    ref_in = torch.nn.functional.interpolate(torch.from_numpy(big.astype(np.float32) / 255.0).permute(2, 0, 1)[None], (S, S),
                                             mode="bilinear", align_corners=False, antialias=True)[0].permute(1, 2, 0)
    ref = tsr_ref.tsr_forward(sd, ref_in.numpy(), SMALL_CFG, pos_mode="size", bf16=True)
    assert _rel(codes[0], ref)[0] < 1e-2


# bounds of the bf16-mode mesh against the fp32 CPU mesh at 128^3, fractions of the scene extent (see the test below)
BF16_MESH_MEAN, BF16_MESH_P99, BF16_MESH_P999, BF16_MESH_MAX = 4e-4, 2e-3, 5e-2, 1e-1


def test_full_size_tsr_forward_vs_oracle(cuda):
    """BASELINE config 2 size: the real architecture (ViT-B/16 @ 1025 tokens, 16 blocks @ 3072 tokens,
    419 M parameters, seeded random init) through the HIP kernels vs the torch-fp32 oracle on the host."""
    from sculptmate_amd import ops
    from sculptmate_amd.tsr import TSR
    from sculptmate_amd.tsr.spec import DEFAULT_CFG

    sd = synth.tsr_state(seed=0)
    m = TSR(pos_embed_mode="scale_factor")  # the transformers-4.38 form the reference pins
    m.load_state_dict(sd)
    m.to(cuda)
    img = synth.composite_rgb(synth.image_rgba(seed=100))
    codes = m([img], device=cuda)
    assert codes.shape == (1, 3, 40, 64, 64)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    col = {}
    ref32 = tsr_ref.tsr_forward(sd, img, DEFAULT_CFG, pos_mode="scale_factor", collect=col)
    r32, _ = _rel(codes[0], ref32)
    # bf16 storage of weights/activations through 12 + 16 layers vs the fp32 reference: documented tolerance
    assert r32 < 3e-2, r32
    ctx, ctx32 = m.image_tokens(torch.from_numpy(img).to(cuda))
    assert _rel(ctx32, col["ctx"])[0] < 2e-2
    # the batched pass (TSR.forward on a list, system.py:82-115) against the SAME fp32 oracle: the same bound, and as close to
    # the oracle as the one-image pass is (the two differ from each other by bf16 rounding flips, 2.7e-3: below)
    codes_b = m([img, synth.composite_rgb(synth.image_rgba(seed=101)), synth.composite_rgb(synth.image_rgba(seed=102))], device=cuda)
    r32_b, _ = _rel(codes_b[0], ref32)
    print("full-size scene code vs fp32 oracle: one-image pass rel %.3e, image 0 of a batched pass of three rel %.3e" % (r32, r32_b))
    assert r32_b < 3e-2 and r32_b < 1.25 * r32 + 1e-3, (r32_b, r32)
    # the tokenizer look-ahead at full size (real launch shapes and timings): several rounds of TSR.run on six images must
    # reproduce the one-image-at-a-time meshes bit for bit (a race between the two streams would show as a different mesh)
    six = [synth.composite_rgb(synth.image_rgba(seed=100 + i)) for i in range(6)]
    thr_l = float(ops.density_grid(codes[0].contiguous(), m.decoder, 64).quantile(0.97))
    serial = [m.run([im], mc_resolution=64, threshold=thr_l)[0] for im in six]
    serial = [(w.vertices.copy(), w.faces.copy()) for w in serial]
    for rnd in range(3):
        for (wv, wf), got in zip(serial, m.run(six, mc_resolution=64, threshold=thr_l)):
            assert np.array_equal(wv, got.vertices) and np.array_equal(wf, got.faces), rnd
    # the fp32 parity mode at full size: fp32 rounding only (28 layers deep)
    del m
    torch.cuda.empty_cache()
    m32 = TSR(pos_embed_mode="scale_factor", precision="fp32")
    m32.load_state_dict(sd)
    m32.to(cuda)
    c32 = m32([img], device=cuda)
    rel32, _ = _rel(c32[0], ref32)
    print("full-size scene code: bf16 mode rel %.3e, fp32 mode rel %.3e" % (r32, rel32))
    assert rel32 < 1e-4, rel32
    # image -> mesh against the CPU path (north-star: vertices within 1e-4 relative of the CPU reference)
    from oracle import capi

    # at BASELINE config 1's resolution (128^3; was tests/tools/parity_e2e.py): oracle forward -> oracle density ->
    # oracle marching cubes on the host vs HIP forward (fp32 mode) -> HIP density -> HIP marching cubes
    from _meshcmp import assert_mesh_close

    R = 128
    Ws, bs = synth.decoder_lists(sd)
    capi.set_threads(min(32, os.cpu_count() or 1))
    dref = capi.density_grid(ref32.numpy(), Ws, bs, R)
    thr = float(np.quantile(dref, 0.97))
    dens = ops.density_grid(c32[0].contiguous(), m32.decoder, R)
    assert float(np.abs(np.log(dens.cpu().numpy()) - np.log(dref)).max()) < 2e-4
    mesh = m32.extract_meshes(c32, resolution=R, threshold=thr)[0]
    rv, rf = capi.reference_isosurface(-(dref - np.float32(thr)), R)
    rv = rv * np.float32(0.87 - (-0.87)) + np.float32(-0.87)
    v, f = mesh.vertices.cpu().numpy(), mesh.faces.cpu().numpy()
    info = assert_mesh_close(v, f, rv, rf, tol=1e-4 * 1.74)  # unconditional: same topology or not
    print("image -> mesh at %d^3: %d vertices, %d faces, %s" % (R, len(v), len(f), info))
    # the FAST parity mode (precision="bf16l3": fp32 storage, every matrix product with three-limb bf16 operands and fp32
    # accumulation, fused three-limb attention): the same two assertions as the exact-fp32 mode -- scene code within fp32 rounding
    # of the oracle, image -> mesh within the north-star tolerance with the oracle's topology -- at ~1/6 of its time
    del m32
    torch.cuda.empty_cache()
    ml3 = TSR(pos_embed_mode="scale_factor", precision="bf16l3")
    ml3.load_state_dict(sd)
    ml3.to(cuda)
    cl3 = ml3([img], device=cuda)
    rell3, _ = _rel(cl3[0], ref32)
    print("full-size scene code, bf16l3 mode: rel %.3e (exact-fp32 mode %.3e)" % (rell3, rel32))
    assert rell3 < 1e-4 and rell3 < 2.0 * rel32 + 1e-6, (rell3, rel32)
    densl3 = ops.density_grid(cl3[0].contiguous(), ml3.decoder, R)
    assert float(np.abs(np.log(densl3.cpu().numpy()) - np.log(dref)).max()) < 2e-4
    meshl3 = ml3.extract_meshes(cl3, resolution=R, threshold=thr)[0]
    infol3 = assert_mesh_close(meshl3.vertices.cpu().numpy(), meshl3.faces.cpu().numpy(), rv, rf, tol=1e-4 * 1.74)
    print("image -> mesh at %d^3 in bf16l3 mode: %s" % (R, infol3))
    # ... and at BASELINE's own grid, 256^3, through the product's default extraction (the two-pass filtered density grid +
    # marching cubes) against the oracle's dense fp32 grid + its marching cubes on the host: the same 1e-4 and the same topology
    R2 = 256
    dref2 = capi.density_grid(ref32.numpy(), Ws, bs, R2)
    thr2 = float(np.quantile(dref2, 0.97))
    assert ml3.decoder_filter
    mesh2 = ml3.extract_meshes(cl3, resolution=R2, threshold=thr2)[0]
    assert ml3.filter_info["filtered"] >= 1 and ml3.filter_info["fallbacks"] == 0, ml3.filter_info
    rv2, rf2 = capi.reference_isosurface(-(dref2 - np.float32(thr2)), R2)
    rv2 = rv2 * np.float32(0.87 - (-0.87)) + np.float32(-0.87)
    info2 = assert_mesh_close(mesh2.vertices.cpu().numpy(), mesh2.faces.cpu().numpy(), rv2, rf2, tol=1e-4 * 1.74)
    print("image -> mesh at %d^3 in bf16l3 mode, filtered grid: %d vertices, %s" % (R2, mesh2.vertices.shape[0], info2))
    del mesh2
    # the faster tolerance mode, TSR(precision="fp16l2") -- the Linears of the two transformers on two fp16 limbs per operand (22
    # bits, three products per multiply): the same three assertions -- scene code within fp32 rounding of the oracle, image -> mesh
    # within 1e-4 at 128^3 and, through the filtered grid, at 256^3
    del ml3
    torch.cuda.empty_cache()
    ml2 = TSR(pos_embed_mode="scale_factor", precision="fp16l2")
    ml2.load_state_dict(sd)
    ml2.to(cuda)
    cl2 = ml2([img], device=cuda)
    rell2, _ = _rel(cl2[0], ref32)
    print("full-size scene code, fp16l2 mode: rel %.3e (bf16l3 %.3e, exact fp32 %.3e)" % (rell2, rell3, rel32))
    assert rell2 < 1e-4 and rell2 < 2.0 * rel32 + 1e-6, (rell2, rel32)
    meshl2 = ml2.extract_meshes(cl2, resolution=R, threshold=thr)[0]
    infol2 = assert_mesh_close(meshl2.vertices.cpu().numpy(), meshl2.faces.cpu().numpy(), rv, rf, tol=1e-4 * 1.74)
    mesh2 = ml2.extract_meshes(cl2, resolution=R2, threshold=thr2)[0]
    info2b = assert_mesh_close(mesh2.vertices.cpu().numpy(), mesh2.faces.cpu().numpy(), rv2, rf2, tol=1e-4 * 1.74)
    print("image -> mesh in fp16l2 mode: 128^3 %s; 256^3 (filtered grid) %s" % (infol2, info2b))
    del dref2, mesh2, rv2, rf2
    ml3 = ml2
    m32 = ml3   # the bf16-mode comparison below only needs a decoder
    # The DEFAULT mode (bf16 transformer, what bench.py times) against the same fp32 CPU mesh: the scene code is 0.8 % away
    # (bf16 weights and activations through 28 layers), so the iso-surface moves; how far is stated here and in bench.py's
    # `parity.bf16_mesh_vs_fp32_cpu` as two-sided nearest-vertex distances over the 1.74 extent.  Measured on MI355X
    # (random-init weights, iso level = the 97 % quantile): mean 1.5e-4, p99 7.9e-4, p99.9 2.2e-2, max 3.9e-2 -- 99 % of the
    # vertices sit within a tenth of a 128^3 voxel (7.9e-3 of the extent) of the fp32 mesh; the tail is small closed components
    # of this rough random field that exist on one side only (a few voxels away from anything).  Bounds = those with ~2.5x margin.
    # The 1e-4 vertex tolerance of north_star holds in fp32 mode (above) or given the same scene code -- not in this mode.
    from _meshcmp import mesh_distance

    mesh_bf = m32.extract_meshes(codes, resolution=R, threshold=thr)[0]
    dist = mesh_distance(mesh_bf.vertices.cpu().numpy(), rv)
    print("bf16-mode mesh vs fp32 CPU mesh at %d^3 (fractions of the 1.74 extent): %s" % (R, dist))
    assert dist["mean"] < BF16_MESH_MEAN and dist["p99"] < BF16_MESH_P99 and dist["p999"] < BF16_MESH_P999 and dist["max"] < BF16_MESH_MAX, dist


def test_gemm_f32_and_softmax(cuda):
    from sculptmate_amd import _lib, ops

    g = torch.Generator().manual_seed(11)
    for (M, N, K) in ((200, 256, 64), (1025, 768, 768), (3072, 1028, 64)):
        A = torch.randn(M, K, generator=g)
        W = torch.randn(N, K, generator=g) / math.sqrt(K)
        b = torch.randn(N, generator=g)
        r = torch.randn(M, N, generator=g)
        ref = (A.double() @ W.double().t() * 0.5 + b + r).float()
        out = torch.empty(M, N, device=cuda)
        ops.gemm_f32(A.to(cuda), W.to(cuda), bias=b.to(cuda), residual=r.to(cuda), out=out, alpha=0.5)
        assert _rel(out, ref)[0] < 5e-7, (M, N, K)
    A = torch.randn(130, 128, generator=g); W = torch.randn(256, 128, generator=g) / 11; b = torch.randn(256, generator=g)
    pre = A @ W.t() + b
    o = torch.empty(130, 128, device=cuda)
    ops.gemm_f32(A.to(cuda), W.to(cuda), bias=b.to(cuda), out=o, epilogue=_lib.EPI_GEGLU)
    assert _rel(o, pre[:, :128] * torch.nn.functional.gelu(pre[:, 128:]))[1] < 2e-5
    x = torch.randn(37, 112, generator=g)
    xs = x.clone().to(cuda)
    ops.softmax_rows_f32(xs, 37, 100, 112)
    assert _rel(xs[:, :100], torch.softmax(x[:, :100], -1))[1] < 1e-6 and (xs[:, 100:] == 0).all()


@pytest.mark.parametrize("M,N,K", [(200, 256, 64), (1025, 768, 768), (3072, 1028, 64), (3072, 1024, 4096), (77, 64, 3072), (1, 4, 32)])
def test_gemm_three_limb_bf16_is_fp32_equivalent(cuda, M, N, K):
    """sculpt_gemm_f32_ex(SCULPT_F32_BF16L3): both fp32 operands split exactly into three bf16 limbs, six exact products, fp32
    accumulate.  Against an fp64 product of the SAME fp32 operands it must be as close as the exact-fp32 matrix instruction is
    (both errors are fp32 accumulation error; the dropped limb products are < 2^-23 relative) -- on plain data, on data with a
    huge dynamic range inside a row (limbs of very different exponents), and with operands that are exact bf16 values."""
    from sculptmate_amd import ops

    g = torch.Generator().manual_seed(M * 7 + N + K)
    for case in ("normal", "range", "bf16"):
        A = torch.randn(M, K, generator=g)
        W = torch.randn(N, K, generator=g) / math.sqrt(K)
        if case == "range":
            A = A * torch.exp2(torch.randint(-20, 20, (M, K), generator=g).float())
            W = W * torch.exp2(torch.randint(-12, 12, (N, K), generator=g).float())
        if case == "bf16":
            A, W = A.to(BF).float(), W.to(BF).float()
        b = torch.randn(N, generator=g)
        r = torch.randn(M, N, generator=g)
        ref = A.double() @ W.double().t() * 0.5 + b.double() + r.double()
        mag = (A.double().abs() @ W.double().abs().t() * 0.5 + b.abs().double() + r.abs().double())   # sum |a||w|: the error scale
        out3 = torch.empty(M, N, device=cuda)
        out1 = torch.empty(M, N, device=cuda)
        ops.gemm_f32(A.to(cuda), W.to(cuda), bias=b.to(cuda), residual=r.to(cuda), out=out3, alpha=0.5, l3=True)
        ops.gemm_f32(A.to(cuda), W.to(cuda), bias=b.to(cuda), residual=r.to(cuda), out=out1, alpha=0.5) if K % 16 == 0 else None
        e3 = float(((out3.cpu().double() - ref).abs() / mag).max())
        e1 = float(((out1.cpu().double() - ref).abs() / mag).max()) if K % 16 == 0 else 1e-6
        # relative to sum |a||w|: fp32 accumulation of K terms; measured 1e-7 .. 5e-7 for BOTH kernels (the wide-range rows,
        # where single products dominate a sum, sit at the top: 4.9e-7 here, 5.4e-7 for the exact instruction)
        assert e3 < 5e-6, (case, e3, e1)                       # (1.3e-6 / 1.4e-6 at K = 768 on the wide-range rows)
        assert e3 < 2.0 * e1 + 2.5 * 2.0 ** -23, (case, e3, e1)
        if case == "bf16":   # one limb each: the single product W1 x1 is exact, only the accumulation order differs from fp32
            assert _rel(out3, ref.float())[0] < 2e-7


def test_gemm_three_limb_epilogues_batch_and_split(cuda):
    """The l3 kernel shares gemm_f32's epilogue: GEGLU / GELU, the Q|K / V^T column split, w_rows clamping, and grid-z batches
    (the heads of an attention as one launch) against torch on the host."""
    from sculptmate_amd import _lib, ops

    g = torch.Generator().manual_seed(5)
    A = torch.randn(130, 128, generator=g); W = torch.randn(256, 128, generator=g) / 11; b = torch.randn(256, generator=g)
    pre = (A.double() @ W.double().t() + b.double())
    o = torch.empty(130, 128, device=cuda)
    ops.gemm_f32(A.to(cuda), W.to(cuda), bias=b.to(cuda), out=o, epilogue=_lib.EPI_GEGLU, l3=True)
    assert _rel(o, (pre[:, :128] * torch.nn.functional.gelu(pre[:, 128:])).float())[1] < 2e-5
    o2 = torch.empty(130, 256, device=cuda)
    ops.gemm_f32(A.to(cuda), W.to(cuda), bias=b.to(cuda), out=o2, epilogue=_lib.EPI_GELU, l3=True)
    assert _rel(o2, torch.nn.functional.gelu(pre).float())[1] < 2e-5
    # column split: first 128 columns token-major, the rest transposed
    o3 = torch.empty(130, 128, device=cuda); ot = torch.zeros(128, 192, device=cuda)
    ops.gemm_f32(A.to(cuda), W.to(cuda), bias=b.to(cuda), out=o3, out_t=ot, n_split=128, l3=True)
    assert _rel(o3, pre[:, :128].float())[0] < 3e-7 and _rel(ot[:, :130].t(), pre[:, 128:].float())[0] < 3e-7 and (ot[:, 130:] == 0).all()
    # heads of an attention as one launch: scores[h] = 0.125 Q_h K_h^T (w_rows clamp on a ragged key count), then P V per head
    heads, Tq, Tk = 3, 70, 45
    D = heads * 64
    Q = torch.randn(Tq, D, generator=g); K = torch.randn(Tk, D, generator=g); V = torch.randn(Tk, D, generator=g)
    Vt = torch.zeros(D, 64); Vt[:, :Tk] = V.t()
    O = torch.empty(Tq, D, device=cuda)
    scores = torch.empty(heads, Tq, 64, device=cuda)
    ops.attention_f32(Q.to(cuda), K.to(cuda), Vt.to(cuda), O, Tq, Tk, heads, 0.125, scores, l3=True)
    qh = Q.double().view(Tq, heads, 64).transpose(0, 1); kh = K.double().view(Tk, heads, 64).transpose(0, 1)
    vh = V.double().view(Tk, heads, 64).transpose(0, 1)
    ref = (torch.softmax(qh @ kh.transpose(1, 2) * 0.125, -1) @ vh).transpose(0, 1).reshape(Tq, D)
    assert _rel(O, ref.float())[0] < 1e-6
    O1 = torch.empty(Tq, D, device=cuda)
    ops.attention_f32(Q.to(cuda), K.to(cuda), Vt.to(cuda), O1, Tq, Tk, heads, 0.125, torch.empty(Tq, 64, device=cuda), l3=True)
    assert torch.equal(O1, O)      # head by head == all heads in one launch
    O2 = torch.empty(Tq, D, device=cuda)
    ops.attention_f32(Q.to(cuda), K.to(cuda), Vt.to(cuda), O2, Tq, Tk, heads, 0.125, torch.empty(heads, Tq, 48, device=cuda))
    assert _rel(O2, ref.float())[0] < 1e-6   # the exact-fp32 instruction through the same batched launches


@pytest.mark.parametrize("Tq,Tk,heads", [(3072, 3072, 16), (3072, 1025, 16), (1025, 1025, 12), (200, 77, 2), (37, 5, 1), (130, 640, 3)])
def test_fused_three_limb_attention_vs_fp64(cuda, Tq, Tk, heads):
    """sculpt_attention_f32_l3 (one launch: three-limb QK^T, fp32 online softmax in registers, three-limb PV) against an fp64
    softmax of the same fp32 operands -- the bound of an fp32 evaluation (1e-6 of the norm), three orders below the bf16 kernel;
    one key is spiked so that the running maximum jumps late in the sequence, and one query row sees only tiny scores."""
    from sculptmate_amd import ops

    g = torch.Generator().manual_seed(Tq * 3 + Tk)
    D = heads * 64
    Q = torch.randn(Tq, D, generator=g); K = torch.randn(Tk, D, generator=g); V = torch.randn(Tk, D, generator=g)
    K[Tk - 1, :64] = 6.0 * Q[min(3, Tq - 1), :64]            # a late maximum jump for one query of head 0
    Q[min(5, Tq - 1)] *= 1e-3
    ldv = ((Tk + 63) // 64) * 64
    Vt = torch.zeros(D, ldv); Vt[:, :Tk] = V.t()
    Vt[:, Tk:] = 7.0                                            # padding columns: finite, multiplied by exact zeros
    O = torch.empty(Tq, D, device=cuda)
    ops.attention_f32(Q.to(cuda), K.to(cuda), Vt.to(cuda), O, Tq, Tk, heads, 0.125, None, l3=True)
    qh = Q.double().view(Tq, heads, 64).transpose(0, 1); kh = K.double().view(Tk, heads, 64).transpose(0, 1)
    vh = V.double().view(Tk, heads, 64).transpose(0, 1)
    ref = (torch.softmax(qh @ kh.transpose(1, 2) * 0.125, -1) @ vh).transpose(0, 1).reshape(Tq, D)
    rel, mx = _rel(O, ref.float())
    assert torch.isfinite(O).all() and rel < 1e-6 and mx < 2e-5, (rel, mx)
    # the composition (scores in HBM) on the same limb arithmetic agrees to fp32 rounding
    O2 = torch.empty(Tq, D, device=cuda)
    ops.attention_f32(Q.to(cuda), K.to(cuda), Vt.to(cuda), O2, Tq, Tk, heads, 0.125,
                      torch.empty(heads, Tq, ((Tk + 31) // 32) * 32, device=cuda), l3=True)
    assert _rel(O2, ref.float())[0] < 1e-6 and _rel(O, O2)[0] < 1e-6


def test_three_limb_kernels_repeated_launches_are_bit_identical(cuda):
    """Race screen for the single-buffer LDS pipelines of gemm_l3.hip / attention_l3.hip (limbs written between two barriers while
    the previous K-step's fragments were just read): a dozen launches each of the pipelined GEMM, the 64-row GEMM and both attention
    forms on full-size shapes must reproduce the first launch bit for bit, with other kernels running in between."""
    from sculptmate_amd import ops

    g = torch.Generator().manual_seed(9)
    A = torch.randn(3072, 1024, generator=g).to(cuda); W = (torch.randn(1024, 1024, generator=g) / 32).to(cuda)
    Wb = (torch.randn(3072, 1024, generator=g) / 32).to(cuda)
    Q = torch.randn(3072, 1024, generator=g).to(cuda); K = torch.randn(1025, 1024, generator=g).to(cuda)
    Vt = torch.zeros(1024, 1088, device=cuda); Vt[:, :1025] = torch.randn(1024, 1025, generator=g).to(cuda)
    noise = torch.randn(4096, 4096, device=cuda)

    def variants():
        o1 = torch.empty(3072, 1024, device=cuda); ops.gemm_f32(A, W, out=o1, l3=True)          # 64-row tiles (192 tiles of 128)
        o2 = torch.empty(3072, 3072, device=cuda); ops.gemm_f32(A, Wb, out=o2, l3=True)
        o3 = torch.empty(3072, 1024, device=cuda); ops.attention_f32(Q, K, Vt, o3, 3072, 1025, 16, 0.125, None, l3=True)
        o4 = torch.empty(1025, 1024, device=cuda); ops.attention_f32(K, K, Vt, o4, 1025, 1025, 16, 0.125, None, l3=True)   # plain 4-wave form
        return o1, o2, o3, o4

    first = variants()
    for it in range(12):
        noise.mul_(1.0001)           # something else on the GPU between the launches
        for a, b in zip(first, variants()):
            assert torch.equal(a, b), it


def test_bf16l3_parity_mode_small_and_mesh(cuda):
    """TSR(precision='bf16l3'): fp32 storage, every matrix product on the bf16 matrix pipe through the exact three-limb split --
    the same bounds as the exact-fp32 mode: scene code within fp32 rounding of the oracle, mesh within the north-star 1e-4."""
    from oracle import capi
    from sculptmate_amd.tsr import TSR

    sd = synth.tsr_state(31, SMALL_CFG)
    m = TSR(SMALL_CFG, pos_embed_mode="size", precision="bf16l3")
    m.load_state_dict(sd)
    m.to(cuda)
    img = synth.composite_rgb(synth.image_rgba(seed=32, size=SMALL_CFG["cond_image_size"]))
    codes = m([img], device=cuda)
    ref = tsr_ref.tsr_forward(sd, img, SMALL_CFG, pos_mode="size")
    rel, mx = _rel(codes[0], ref)
    assert rel < 2e-5, rel
    m32 = TSR(SMALL_CFG, pos_embed_mode="size", precision="fp32")
    m32.load_state_dict(sd)
    m32.to(cuda)
    rel32, _ = _rel(m32([img], device=cuda)[0], ref)
    print("small model scene code vs fp32 oracle: bf16l3 %.3e, exact fp32 %.3e" % (rel, rel32))
    assert rel < 2.0 * rel32 + 2e-7
    R = 40
    Ws, bs = synth.decoder_lists(sd)
    dens_ref = capi.density_grid(ref.numpy(), Ws, bs, R)
    thr = float(np.median(dens_ref))
    mesh = m.run([img], mc_resolution=R, threshold=thr)[0]
    rv, rf = capi.reference_isosurface(-(dens_ref - np.float32(thr)), R)
    rv = rv * np.float32(0.87 - (-0.87)) + np.float32(-0.87)
    from _meshcmp import assert_mesh_close

    assert_mesh_close(mesh.vertices, mesh.faces, rv, rf, tol=1e-4 * 1.74)
    # a batch in this mode runs the entries' attentions one after the other (engine._attn) on the same stacked Linears
    two = m([img, synth.composite_rgb(synth.image_rgba(seed=33, size=SMALL_CFG["cond_image_size"]))], device=cuda)
    assert _rel(two[0], ref)[0] < 2e-5


def test_fp32_parity_mode_small_and_mesh(cuda):
    """TSR(precision='fp32'): the whole forward on the exact-fp32 matrix pipe reproduces the fp32 oracle to
    fp32 rounding, and the mesh extracted from it matches the mesh of the oracle's scene code."""
    from oracle import capi
    from sculptmate_amd.tsr import TSR

    sd = synth.tsr_state(31, SMALL_CFG)
    m = TSR(SMALL_CFG, pos_embed_mode="size", precision="fp32")
    m.load_state_dict(sd)
    m.to(cuda)
    img = synth.composite_rgb(synth.image_rgba(seed=32, size=SMALL_CFG["cond_image_size"]))
    codes = m([img], device=cuda)
    ref = tsr_ref.tsr_forward(sd, img, SMALL_CFG, pos_mode="size")
    rel, mx = _rel(codes[0], ref)
    assert rel < 2e-5, rel
    # image -> mesh, end to end, against the CPU path (oracle forward -> oracle density -> oracle marching cubes)
    R = 40
    Ws, bs = synth.decoder_lists(sd)
    dens_ref = capi.density_grid(ref.numpy(), Ws, bs, R)
    thr = float(np.median(dens_ref))
    mesh = m.run([img], mc_resolution=R, threshold=thr)[0]
    rv, rf = capi.reference_isosurface(-(dens_ref - np.float32(thr)), R)
    rv = rv * np.float32(0.87 - (-0.87)) + np.float32(-0.87)
    from _meshcmp import assert_mesh_close

    # vertices within the north-star tolerance (1e-4 relative to the scene extent), same topology or not
    assert_mesh_close(mesh.vertices, mesh.faces, rv, rf, tol=1e-4 * 1.74)


def test_encode_image_two_stream_overlap_is_bit_identical(cuda):
    """TSR.encode_image issues the image-independent head of the backbone on a second HIP stream under the ViT: same kernels,
    same operands -> the same bits as the sequential calls, also back to back without host synchronisation."""
    m, sd = _small_model(cuda)
    S = SMALL_CFG["cond_image_size"]
    imgs = [torch.from_numpy(synth.composite_rgb(synth.image_rgba(seed=40 + i, size=S))).to(cuda) for i in range(3)]
    want = []
    for im in imgs:
        ctx, _ = m.image_tokens(im)
        out, outb = m.backbone_tokens(ctx)
        want.append((out.clone(), outb.clone()))
    got = []
    for rep in range(3):
        for im in imgs:                       # no synchronisation between images: the fork event orders the buffer reuse
            out, outb = m.encode_image(im)
            got.append((out.clone(), outb.clone()))
    torch.cuda.synchronize()
    for k, (o, ob) in enumerate(got):
        assert torch.equal(o, want[k % 3][0]) and torch.equal(ob, want[k % 3][1]), k
    # forward() goes through the same path
    codes = m([synth.composite_rgb(synth.image_rgba(seed=40, size=S))], device=cuda)
    assert torch.equal(codes[0], m.scene_code(want[0][1]))


def test_image_file_to_glb_chain(cuda, tmp_path):
    """The add-on's whole chain headless: image file -> preprocess_image (U^2-Net cut-out on the GPU, the reference's framing)
    -> TSR.run -> Mesh.export('.glb') -> read back."""
    from PIL import Image

    from sculptmate_amd import meshio, ops, preprocessing
    from sculptmate_amd.rembg import session

    path = str(tmp_path / "photo.png")
    Image.fromarray(synth.image_rgba(seed=70, size=384)[..., :3], mode="RGB").save(path)
    sess = session.U2netSession(device=cuda, state_dict=synth.u2net_state(0))
    framed = preprocessing.preprocess_image(path, ratio=0.75, session=sess)
    assert framed is not None and framed.size == (1024, 1024) and framed.mode == "RGB"
    m, sd = _small_model(cuda, seed=71)
    codes = m([framed], device=cuda)                       # PIL 1024^2 in, resized on the GPU like the add-on's input
    dens = ops.density_grid(codes[0].contiguous(), m.decoder, 40)
    mesh = m.run([framed], mc_resolution=40, threshold=float(dens.median()), enable_texture=True)[0]
    out = str(tmp_path / "mesh.glb")
    mesh.export(out)
    back = meshio.read_glb(out)
    assert np.array_equal(back["vertices"], mesh.vertices) and np.array_equal(back["faces"], mesh.faces)
    assert np.array_equal(back["vertex_colors"], mesh.vertex_colors)
    mesh.export(str(tmp_path / "mesh.obj"))
    mesh.export(str(tmp_path / "mesh.ply"))
    v2, f2, c2 = meshio.read_ply(str(tmp_path / "mesh.ply"))
    assert np.array_equal(v2, mesh.vertices) and np.array_equal(f2, mesh.faces)
    with pytest.raises(ValueError):
        mesh.export(str(tmp_path / "mesh.stl"))
