"""The "limbs once" GEMM of the tolerance mode (csrc/gemm_l3p.hip) against the kernel that splits while staging (gemm_l3.hip):
same products in the same order, so every comparison here is BIT for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _rand(shape, g, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(_dev())


def test_limbs_split_is_exact_and_tiled_as_documented():
    from sculptmate_amd import ops

    g = torch.Generator().manual_seed(1)
    R, K = 77, 96
    x = _rand((R, K), g) * torch.exp2(torch.randint(-30, 30, (R, K), generator=g).float()).to(_dev())
    x[3, 5] = 0.0
    x[4, 6] = -0.0
    lt = ops.limbs_split(x)
    assert lt.numel() == ops.limbs_bytes(R, K) == 3 * 96 * 192
    assert torch.equal(ops.limbs_join(lt, R, K), x)
    # the documented byte offset of limb l of X[r][k], element by element on a few positions
    v = lt.view(torch.bfloat16)
    for r, k in ((0, 0), (31, 7), (32, 8), (76, 95), (45, 50)):
        limbs = [float(v[((((r // 32) * (K // 8) + k // 8) * 3 + l) * 512 + (r % 32) * 16 + (k % 8) * 2) // 2]) for l in range(3)]
        assert limbs[0] == float(x[r, k].to(torch.bfloat16))
        assert (limbs[0] + limbs[1]) + limbs[2] == float(x[r, k])
    # pad rows of the last block are zeros
    assert float(v.view(-1, K // 8, 3, 32, 8)[2, :, :, 77 - 64:].abs().max()) == 0.0
    # a strided source
    big = _rand((R, 2 * K), g)
    assert torch.equal(ops.limbs_join(ops.limbs_split(big[:, K:]), R, K), big[:, K:])


@pytest.mark.parametrize("M,N,K,bm64", [(1025, 768, 768, None), (3072, 1024, 1024, "0"), (3072, 1024, 1024, "1"), (200, 256, 64, None),
                                         (1025, 2304, 768, None)])
def test_l3p_gemm_equals_the_splitting_kernel_bit_for_bit(M, N, K, bm64, monkeypatch):
    from sculptmate_amd import ops

    if bm64 is not None:
        monkeypatch.setenv("SCULPT_L3_TILE", "bm64" if bm64 == "1" else "nobm64")
    g = torch.Generator().manual_seed(M + N)
    A, W, bias = _rand((M, K), g), _rand((N, K), g, K ** -0.5), _rand((N,), g)
    res = _rand((M, N), g)
    A_lt, W_lt = ops.limbs_split(A), ops.limbs_split(W)
    # plain + bias
    want = torch.empty(M, N, device=_dev()); got = torch.full((M, N), float("nan"), device=_dev())
    ops.gemm_f32(A, W, bias=bias, out=want, l3=True)
    ops.gemm_l3p(A_lt, W_lt, M, N, K, bias=bias, out=got)
    assert torch.equal(got, want)
    # residual in place, no bias
    want = res.clone(); got = res.clone()
    ops.gemm_f32(A, W, residual=want, out=want, l3=True)
    ops.gemm_l3p(A_lt, W_lt, M, N, K, residual=got, out=got)
    assert torch.equal(got, want)
    # column split with a transposed part (Q | K token-major, V^T)
    ns = N // 2
    ldt = ((M + 63) // 64) * 64
    w1, w2 = torch.zeros(M, ns, device=_dev()), torch.zeros(N - ns, ldt, device=_dev())
    g1, g2 = torch.zeros(M, ns, device=_dev()), torch.zeros(N - ns, ldt, device=_dev())
    ops.gemm_f32(A, W, bias=bias, out=w1, out_t=w2, n_split=ns, l3=True)
    ops.gemm_l3p(A_lt, W_lt, M, N, K, bias=bias, out=g1, out_t=g2, n_split=ns)
    assert torch.equal(g1, w1) and torch.equal(g2, w2)
    # the result as limbs: exactly the fp32 result, split
    out_lt = ops.limbs_empty(M, N, _dev(), zero=True)
    ops.gemm_l3p(A_lt, W_lt, M, N, K, bias=bias, out_lt=out_lt)
    ops.gemm_f32(A, W, bias=bias, out=want, l3=True)
    assert torch.equal(ops.limbs_join(out_lt, M, N), want)


@pytest.mark.parametrize("M,N,K", [(1025, 3072, 768), (3072, 4096, 1024), (96, 128, 64)])
def test_l3p_gelu_and_geglu_epilogues(M, N, K):
    from sculptmate_amd import _lib, ops

    g = torch.Generator().manual_seed(N)
    A, bias = _rand((M, K), g), _rand((N,), g)
    W = _rand((N, K), g, K ** -0.5)
    A_lt = ops.limbs_split(A)
    want = torch.empty(M, N, device=_dev()); got = torch.empty(M, N, device=_dev())
    ops.gemm_f32(A, W, bias=bias, out=want, epilogue=_lib.EPI_GELU, l3=True)
    ops.gemm_l3p(A_lt, ops.limbs_split(W), M, N, K, bias=bias, out=got, epilogue=_lib.EPI_GELU)
    assert torch.equal(got, want)
    out_lt = ops.limbs_empty(M, N, _dev(), zero=True)
    ops.gemm_l3p(A_lt, ops.limbs_split(W), M, N, K, bias=bias, out_lt=out_lt, epilogue=_lib.EPI_GELU)
    assert torch.equal(ops.limbs_join(out_lt, M, N), want)
    # GEGLU: W [2 No][K] value rows then gate rows; the limb-tiled weight carries its row blocks in tile order
    No = N // 2
    want = torch.empty(M, No, device=_dev()); got = torch.empty(M, No, device=_dev())
    ops.gemm_f32(A, W, bias=bias, out=want, epilogue=_lib.EPI_GEGLU, l3=True)
    W_lt = ops.limbs_split(ops.geglu_row_blocks(W))
    ops.gemm_l3p(A_lt, W_lt, M, No, K, bias=bias, out=got, epilogue=_lib.EPI_GEGLU)
    assert torch.equal(got, want)
    out_lt = ops.limbs_empty(M, No, _dev(), zero=True)
    ops.gemm_l3p(A_lt, W_lt, M, No, K, bias=bias, out_lt=out_lt, epilogue=_lib.EPI_GEGLU)
    assert torch.equal(ops.limbs_join(out_lt, M, No), want)


def test_l3p_refuses_what_it_cannot_do():
    from sculptmate_amd import ops

    big = ops.limbs_empty(128, 64, _dev(), zero=True)    # enough bytes for every shape below: the C entry is what refuses
    out = torch.empty(64, 128, device=_dev())
    with pytest.raises(ops.SculptError):
        ops.gemm_l3p(big, big, 64, 100, 64, out=out)     # N % 128
    with pytest.raises(ops.SculptError):
        ops.gemm_l3p(big, big, 64, 128, 48, out=out)     # K % 32
    with pytest.raises(ops.SculptError):
        ops.gemm_l3p(big, big, 64, 128, 64)              # no output
    with pytest.raises(ops.SculptError):
        ops.limbs_split(torch.zeros(8, 40, device=_dev()))
    # the Python front-end checks what the C entry cannot see: the sizes of the limb arrays
    small = ops.Limbs(32, 64, _dev(), zero=True)
    with pytest.raises(AssertionError):
        ops.gemm_l3p(small, big, 64, 128, 64, out=out)   # A holds 32 rows, the call needs 64
    with pytest.raises(AssertionError):
        ops.gemm_l3p(ops.Limbs(64, 32, _dev(), zero=True), big, 64, 128, 64, out=out)   # K mismatch


def test_layernorm_and_attention_write_the_limbs_of_their_fp32_results():
    import math

    from sculptmate_amd import ops

    g = torch.Generator().manual_seed(5)
    rows, cols = 1025, 768
    x, gamma, beta = _rand((rows, cols), g, 3.0), _rand((cols,), g), _rand((cols,), g)
    want = torch.empty(rows, cols, device=_dev())
    ops.layernorm(x, gamma, beta, 1e-5, y_f32=want)
    lt, also = ops.Limbs(rows, cols, _dev(), zero=True), torch.empty(rows, cols, device=_dev())
    ops.layernorm(x, gamma, beta, 1e-5, y_lt=lt, y_f32=also)
    assert torch.equal(lt.float(), want) and torch.equal(also, want)
    # fused three-limb attention: both kernel forms (4-wave: the tokenizer's shape; pipelined 8-wave: the backbone's), with a row offset
    for Tq, Tk, heads in ((1025, 1025, 12), (3072, 1025, 16)):
        D = heads * 64
        Q, K = _rand((Tq, D), g), _rand((Tk, D), g)
        ldv = ((Tk + 63) // 64) * 64
        Vt = torch.zeros(D, ldv, device=_dev()); Vt[:, :Tk] = _rand((D, Tk), g)
        want = torch.empty(Tq, D, device=_dev())
        ops.attention_f32(Q, K, Vt, want, Tq, Tk, heads, 1.0 / math.sqrt(64), None, l3=True)
        row0 = 40
        O = ops.Limbs(row0 + Tq, D, _dev(), zero=True)
        ops.attention_f32(Q, K, Vt, O, Tq, Tk, heads, 1.0 / math.sqrt(64), None, l3=True, o_row0=row0)
        got = O.float()
        assert torch.equal(got[row0:], want) and float(got[:row0].abs().max()) == 0.0


def test_limbs_once_forward_is_bit_identical_to_the_splitting_kernels(monkeypatch):
    """TSR(precision="bf16l3") with the operands split once (default) against SCULPT_L3_TILE=split (every GEMM splits while staging): the
    same products in the same order -> the same scene code, bit for bit; one image and a batch of two."""
    from sculptmate_amd import synth
    from sculptmate_amd.tsr.spec import SMALL_CFG
    from sculptmate_amd.tsr.system import TSR

    sd = synth.tsr_state(3, SMALL_CFG)
    imgs = [torch.from_numpy(synth.composite_rgb(synth.image_rgba(seed=s, size=SMALL_CFG["cond_image_size"]))).to(_dev()) for s in (1, 2)]
    codes = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("SCULPT_L3_TILE", "" if flag == "1" else "split")
        m = TSR(SMALL_CFG, pos_embed_mode="size", precision="bf16l3")
        m.load_state_dict(sd)
        m.to(_dev())
        assert m.l3p == (flag == "1")
        m.max_batch = 2
        with torch.no_grad():
            codes[flag] = (m.forward(imgs[0]).clone(), m.forward(imgs).clone())
    for a, b in zip(codes["1"], codes["0"]):
        assert torch.isfinite(a).all() and torch.equal(a, b)


def test_batched_three_limb_attention_equals_the_per_entry_launches():
    """sculpt_attention_f32_l3_batched (grid z = batch entry) against one launch per entry: token rows stacked, V^T side by side
    in one array (a column offset per entry), fp32 and limb outputs; both kernel forms."""
    import math

    from sculptmate_amd import ops

    g = torch.Generator().manual_seed(9)
    for T, Tk, heads, B in ((520, 520, 4, 3), (3072, 1032, 16, 2)):
        D = heads * 64
        Ts = ((Tk + 7) // 8) * 8
        Q, K = _rand((B * T, D), g), _rand((B * Ts, D), g)
        ldv = (B - 1) * Ts + ((Tk + 63) // 64) * 64
        Vt = _rand((D, ldv), g)
        scale = 1.0 / math.sqrt(64)
        want = torch.empty(B * T, D, device=_dev())
        for b in range(B):
            ops.attention_f32(Q[b * T:], K[b * Ts:], Vt[:, b * Ts:], want[b * T:], T, Tk, heads, scale, None, l3=True)
        got = torch.full((B * T, D), float("nan"), device=_dev())
        ops.attention_f32_l3_batched(Q, K, Vt, got, T, Tk, heads, scale, B, T * D, Ts * D, Ts, T * D)
        assert torch.equal(got, want)
        O = ops.Limbs(B * T, D, _dev(), zero=True)
        ops.attention_f32_l3_batched(Q, K, Vt, O, T, Tk, heads, scale, B, T * D, Ts * D, Ts, T * D)
        assert torch.equal(O.float(), want)


@pytest.mark.parametrize("M,N,K", [(1, 128, 32), (33, 128, 32), (31, 256, 96), (129, 384, 160), (64, 128, 4096)])
def test_l3p_smallest_and_ragged_shapes(M, N, K, monkeypatch):
    """One row, one K-tile pair, a last row block with a single row, a tile whose row blocks run past the matrix (clamped reads,
    unwritten rows), a long K: every tile form, all bit-identical to the splitting kernel; GEGLU at its smallest width."""
    from sculptmate_amd import _lib, ops

    g = torch.Generator().manual_seed(M * 7 + N + K)
    A, W, bias, res = _rand((M, K), g), _rand((N, K), g, K ** -0.5), _rand((N,), g), _rand((M, N), g)
    A_lt, W_lt = ops.Limbs.of(A), ops.Limbs.of(W)
    want = res.clone()
    ops.gemm_f32(A, W, bias=bias, residual=want, out=want, l3=True)
    for bm64, nw8 in (("0", "0"), ("1", "0"), ("0", "1")):
        monkeypatch.setenv("SCULPT_L3_TILE", "%s,%s" % ("bm64" if bm64 == "1" else "nobm64", "nw8" if nw8 == "1" else "nonw8"))
        got = res.clone()
        ops.gemm_l3p(A_lt, W_lt, M, N, K, bias=bias, residual=got, out=got)
        assert torch.equal(got, want), (bm64, nw8)
        out_lt = ops.Limbs(M, N, _dev(), zero=True)
        ops.gemm_l3p(A_lt, W_lt, M, N, K, bias=bias, out_lt=out_lt, epilogue=_lib.EPI_GELU)
        w2 = torch.empty(M, N, device=_dev())
        ops.gemm_f32(A, W, bias=bias, out=w2, epilogue=_lib.EPI_GELU, l3=True)
        assert torch.equal(out_lt.float(), w2), (bm64, nw8)
    # GEGLU with 64 output columns (one tile)
    Wg, bg = _rand((128, K), g, K ** -0.5), _rand((128,), g)
    want = torch.empty(M, 64, device=_dev()); got = torch.empty(M, 64, device=_dev())
    ops.gemm_f32(A, Wg, bias=bg, out=want, epilogue=_lib.EPI_GEGLU, l3=True)
    ops.gemm_l3p(A_lt, ops.Limbs.of(ops.geglu_row_blocks(Wg)), M, 64, K, bias=bg, out=got, epilogue=_lib.EPI_GEGLU)
    assert torch.equal(got, want)


def test_two_fp16_limbs_format():
    """LIMBS_F16X2: the split keeps 22 bits (an absolute 2^-25 for small values), a weight travels pre-scaled into the fp16 range,
    the GEMM's three products are fp32-equivalent on operands that already fit 22 bits and within 2^-21 of Sum |a||w| otherwise;
    LayerNorm / attention / GELU epilogues write the same limbs as the stand-alone split of their fp32 results."""
    import math

    from sculptmate_amd import _lib, ops

    g = torch.Generator().manual_seed(21)
    M, N, K = 300, 256, 160
    A, W, bias = _rand((M, K), g), _rand((N, K), g, 0.03), _rand((N,), g)
    A_lt, W_lt = ops.Limbs.of(A, fmt="f16x2"), ops.Limbs.of(W, fmt="f16x2", weight=True)
    assert W_lt.scale >= 2.0 ** 14 and math.log2(W_lt.scale).is_integer()
    ja, jw = A_lt.float(), W_lt.float()
    assert float((ja - A).abs().max()) <= 2.0 ** -22 * float(A.abs().max()) and float(((ja - A).abs() / A.abs().clamp_min(0.125)).max()) <= 2.0 ** -22
    assert float(((jw - W).abs() / W.abs().clamp_min(1e-30)).max()) <= 2.0 ** -21      # scaled: full precision down to tiny weights
    assert ops.limbs_bytes(M, K, "f16x2") == 10 * K * 128 and A_lt.data.numel() == ops.limbs_bytes(M, K, "f16x2")
    # the GEMM against fp64 on the SAME 22-bit operands: only the dropped h2.h2 term and fp32 accumulation remain
    ref = (ja.double() @ jw.double().t() + bias.double())
    for bm64, nw8 in (("0", "0"), ("1", "0"), ("0", "1")):
        import os
        os.environ["SCULPT_L3_TILE"] = "%s,%s" % ("bm64" if bm64 == "1" else "nobm64", "nw8" if nw8 == "1" else "nonw8")
        try:
            got = torch.empty(M, N, device=_dev())
            ops.gemm_l3p(A_lt, W_lt, M, N, K, bias=bias, out=got)
        finally:
            os.environ.pop("SCULPT_L3_TILE")
        bound = (ja.abs().double() @ jw.abs().double().t())
        assert float(((got.double() - ref).abs() / bound).max()) < 3e-7, (bm64, nw8)
    # and against fp64 on the fp32 operands: 22-bit operand rounding
    ref32 = A.double() @ W.double().t() + bias.double()
    bound = A.abs().double() @ W.abs().double().t()
    assert float(((got.double() - ref32).abs() / bound).max()) < 2.0 ** -20
    # producers: GELU epilogue, LayerNorm, attention write split(fp32 result)
    want = torch.empty(M, N, device=_dev())
    ops.gemm_l3p(A_lt, W_lt, M, N, K, bias=bias, out=want, epilogue=_lib.EPI_GELU)
    o_lt = ops.Limbs(M, N, _dev(), zero=True, fmt="f16x2")
    ops.gemm_l3p(A_lt, W_lt, M, N, K, bias=bias, out_lt=o_lt, epilogue=_lib.EPI_GELU)
    assert torch.equal(o_lt.float(), ops.Limbs.of(want, fmt="f16x2").float())   # (pad rows of the last block differ: never read)
    x, gamma, beta = _rand((77, 256), g, 3.0), _rand((256,), g), _rand((256,), g)
    y = torch.empty(77, 256, device=_dev())
    ops.layernorm(x, gamma, beta, 1e-5, y_f32=y)
    y_lt = ops.Limbs(77, 256, _dev(), zero=True, fmt="f16x2")
    ops.layernorm(x, gamma, beta, 1e-5, y_lt=y_lt)
    assert torch.equal(y_lt.float(), ops.Limbs.of(y, fmt="f16x2").float())
    Tq, Tk, heads = 200, 130, 4
    D = heads * 64
    Q, Kk = _rand((Tq, D), g), _rand((Tk, D), g)
    Vt = torch.zeros(D, 192, device=_dev()); Vt[:, :Tk] = _rand((D, Tk), g)
    o = torch.empty(Tq, D, device=_dev())
    ops.attention_f32(Q, Kk, Vt, o, Tq, Tk, heads, 0.125, None, l3=True)
    O = ops.Limbs(Tq, D, _dev(), zero=True, fmt="f16x2")
    ops.attention_f32(Q, Kk, Vt, O, Tq, Tk, heads, 0.125, None, l3=True)
    assert torch.equal(O.float(), ops.Limbs.of(o, fmt="f16x2").float())
    # overflow is loud, not silent: a value beyond the fp16 range becomes inf
    big = A.clone(); big[0, 0] = 1e5
    assert not bool(torch.isfinite(ops.Limbs.of(big, fmt="f16x2").float()).all())


def test_fp16l2_mode_is_fp32_equivalent_on_the_small_model(monkeypatch):
    """TSR(precision="fp16l2") against the exact-fp32 mode and the three-limb mode: the scene code differs from fp32 by fp32
    rounding noise (the bound the three-limb mode meets), one image and a batch."""
    from sculptmate_amd import synth
    from sculptmate_amd.tsr.spec import SMALL_CFG
    from sculptmate_amd.tsr.system import TSR

    sd = synth.tsr_state(3, SMALL_CFG)
    imgs = [torch.from_numpy(synth.composite_rgb(synth.image_rgba(seed=s, size=SMALL_CFG["cond_image_size"]))).to(_dev()) for s in (1, 2)]
    codes = {}
    for prec in ("fp32", "bf16l3", "fp16l2"):
        m = TSR(SMALL_CFG, pos_embed_mode="size", precision=prec)
        m.load_state_dict(sd)
        m.to(_dev())
        m.max_batch = 2
        with torch.no_grad():
            codes[prec] = (m.forward(imgs[0]).clone(), m.forward(imgs).clone())
        assert (m.limb_format == "f16x2") == (prec == "fp16l2")
    rel = lambda a, b: float((a - b).norm() / b.norm())
    for i in range(2):
        r3, r2 = rel(codes["bf16l3"][i], codes["fp32"][i]), rel(codes["fp16l2"][i], codes["fp32"][i])
        print("scene code vs exact fp32: bf16l3 %.2e, fp16l2 %.2e" % (r3, r2))
        assert r2 < 1e-5 and r2 < 4.0 * r3 + 1e-6


@pytest.mark.parametrize("Tq,Tk,heads,gain", [(3072, 3072, 16, 1.0), (3072, 1025, 16, 1.0), (600, 333, 4, 1.0), (300, 64, 2, 6.0), (257, 1, 1, 1.0)])
def test_two_fp16_limb_attention_vs_fp64(Tq, Tk, heads, gain, monkeypatch):
    """csrc/attention_l2.hip (both products on two fp16 limbs per operand, three limb products) against an fp64 softmax(QK^T s)V of
    the same fp32 operands: the error of the three-limb kernel's class (fp32 rounding of the softmax), ragged key tiles, one key,
    sharp distributions; fp32 and limb outputs; a batch."""
    import math

    from sculptmate_amd import ops

    monkeypatch.setenv("SCULPT_ATTN_FORM", "l3pipe")     # the pipelined form on every shape (the small ones default to the 4-wave kernel)
    g = torch.Generator().manual_seed(Tq + Tk)
    D = heads * 64
    Q, K = _rand((Tq, D), g, gain), _rand((Tk, D), g)
    ldv = ((Tk + 63) // 64) * 64
    Vt = torch.zeros(D, ldv, device=_dev()); Vt[:, :Tk] = _rand((D, Tk), g)
    scale = 1.0 / math.sqrt(64)
    qh = Q.double().view(Tq, heads, 64).transpose(0, 1); kh = K.double().view(Tk, heads, 64).transpose(0, 1)
    vh = Vt[:, :Tk].t().double().view(Tk, heads, 64).transpose(0, 1)
    ref = (torch.softmax(qh @ kh.transpose(1, 2) * scale, -1) @ vh).transpose(0, 1).reshape(Tq, D)
    o3 = torch.empty(Tq, D, device=_dev()); o2 = torch.full((Tq, D), float("nan"), device=_dev())
    ops.attention_f32(Q, K, Vt, o3, Tq, Tk, heads, scale, None, l3=True)
    ops.attention_f32(Q, K, Vt, o2, Tq, Tk, heads, scale, None, l3=True, two_fp16_limbs=True)
    e3, e2 = float((o3.double() - ref).norm() / ref.norm()), float((o2.double() - ref).norm() / ref.norm())
    print("Tq %d Tk %d: rel err vs fp64: three bf16 limbs %.2e, two fp16 limbs %.2e" % (Tq, Tk, e3, e2))
    assert torch.isfinite(o2).all() and e2 < 2e-6 and e2 < 4.0 * e3 + 2e-7
    O = ops.Limbs(Tq + 32, D, _dev(), zero=True, fmt="f16x2")
    ops.attention_f32(Q, K, Vt, O, Tq, Tk, heads, scale, None, l3=True, o_row0=32, two_fp16_limbs=True)
    assert torch.equal(O.float()[32:], ops.Limbs.of(o2, fmt="f16x2").float())
    if Tq == 600:   # two entries in one launch, V^T side by side
        Ts = ((Tk + 7) // 8) * 8
        Q2, K2 = torch.cat([Q, Q.flip(0)]), torch.zeros(2 * Ts, D, device=_dev())
        K2[:Tk], K2[Ts:Ts + Tk] = K, K.flip(0)
        Vt2 = torch.zeros(D, Ts + ldv, device=_dev()); Vt2[:, :Tk] = Vt[:, :Tk]; Vt2[:, Ts:Ts + Tk] = Vt[:, :Tk].flip(1)
        got = torch.empty(2 * Tq, D, device=_dev())
        ops.attention_f32_l3_batched(Q2, K2, Vt2, got, Tq, Tk, heads, scale, 2, Tq * D, Ts * D, Ts, Tq * D, two_fp16_limbs=True)
        assert torch.equal(got[:Tq], o2)
        assert float((got[Tq:].flip(0).double() - ref).norm() / ref.norm()) < 2e-6


def test_fp16l2_range_fallback():
    """An activation beyond the fp16 range (here: a LayerNorm gain of 3e5 in one block) makes the two-limb result non-finite;
    forward() answers with the three-limb twin's scene code -- bit for bit what TSR(precision="bf16l3") gives -- and counts it."""
    from sculptmate_amd import synth
    from sculptmate_amd.tsr.spec import SMALL_CFG
    from sculptmate_amd.tsr.system import TSR

    sd = dict(synth.tsr_state(5, SMALL_CFG))
    key = "backbone.transformer_blocks.1.norm3.weight"
    sd[key] = sd[key] * 3e5
    img = torch.from_numpy(synth.composite_rgb(synth.image_rgba(seed=4, size=SMALL_CFG["cond_image_size"]))).to(_dev())
    got = {}
    for prec in ("fp16l2", "bf16l3"):
        m = TSR(SMALL_CFG, pos_embed_mode="size", precision=prec)
        m.load_state_dict(sd)
        m.to(_dev())
        with torch.no_grad():
            got[prec] = m.forward(img).clone()
            if prec == "fp16l2":
                assert m.range_fallbacks == 1
                again = m.forward(img)
                assert m.range_fallbacks == 2 and torch.equal(again, got[prec])
    assert torch.isfinite(got["bf16l3"]).all() and torch.equal(got["fp16l2"], got["bf16l3"])


def test_fp16l2_through_the_pipelined_entry(monkeypatch):
    """TSR.run on a list of host images (tokens of image i + 1 on a second stream; the tokens travel as limbs) in the fp16l2 mode:
    the same meshes as forward + extract_meshes one image at a time, and the range fallback works there as well."""
    from sculptmate_amd import synth
    from sculptmate_amd.tsr.spec import SMALL_CFG
    from sculptmate_amd.tsr.system import TSR

    sd = dict(synth.tsr_state(7, SMALL_CFG))
    imgs = [synth.composite_rgb(synth.image_rgba(seed=s, size=SMALL_CFG["cond_image_size"])) for s in (11, 12, 13)]
    m = TSR(SMALL_CFG, pos_embed_mode="size", precision="fp16l2")
    m.load_state_dict(sd)
    m.to(_dev())
    with torch.no_grad():
        thr = float(torch.median(__import__("sculptmate_amd").ops.density_grid(m([imgs[0]], device=_dev())[0].contiguous(), m.decoder, 32)))
        one = [m.extract_meshes(m([im], device=_dev()), False, 32, thr)[0] for im in imgs]
        many = m.run(imgs, mc_resolution=32, threshold=thr)
    for a, b in zip(one, many):
        assert np.array_equal(np.asarray(a.vertices.cpu()), np.asarray(b.vertices)) and np.array_equal(np.asarray(a.faces.cpu()), np.asarray(b.faces))
    assert m.range_fallbacks == 0
    key = "backbone.transformer_blocks.1.norm3.weight"
    sd[key] = sd[key] * 3e5
    m2 = TSR(SMALL_CFG, pos_embed_mode="size", precision="fp16l2"); m2.load_state_dict(sd); m2.to(_dev())
    m3 = TSR(SMALL_CFG, pos_embed_mode="size", precision="bf16l3"); m3.load_state_dict(sd); m3.to(_dev())
    with torch.no_grad():
        tok = m2.tokens_async(imgs[0])
        got = m2.forward_tokens(tok)
        want = m3([imgs[0]], device=_dev())
    assert m2.range_fallbacks == 1 and torch.equal(got, want)


def test_reloading_weights_drops_the_range_twin_and_the_filter_calibration():
    """state derived from the weights -- the bf16l3 twin an fp16l2 model falls back to, the two-pass grid's margin -- must
    not survive load_state_dict: after a reload the fallback has to answer with the NEW weights' scene code."""
    from sculptmate_amd import synth
    from sculptmate_amd.tsr.spec import SMALL_CFG
    from sculptmate_amd.tsr.system import TSR

    key = "backbone.transformer_blocks.1.norm3.weight"
    img = synth.composite_rgb(synth.image_rgba(seed=11, size=SMALL_CFG["cond_image_size"]))
    sds = []
    for seed in (7, 8):
        sd = dict(synth.tsr_state(seed, SMALL_CFG))
        sd[key] = sd[key] * 3e5          # leaves the fp16 range: every forward goes through the twin
        sds.append(sd)
    m = TSR(SMALL_CFG, pos_embed_mode="size", precision="fp16l2")
    m.load_state_dict(sds[0])
    m.to(_dev())
    with torch.no_grad():
        a = m([img], device=_dev())
        assert m.range_fallbacks == 1 and m._range_twin is not None
        m.filter_info.update(margin=0.123, usable=False)
        m.load_state_dict(sds[1])
        assert m._range_twin is None and m.filter_info["margin"] is None and m.filter_info["usable"]
        b = m([img], device=_dev())
        ref = TSR(SMALL_CFG, pos_embed_mode="size", precision="bf16l3")
        ref.load_state_dict(sds[1])
        ref.to(_dev())
        want = ref([img], device=_dev())
    assert m.range_fallbacks == 2 and torch.equal(b, want) and not torch.equal(a, b)


def test_bf16l3_falls_back_to_the_splitting_gemm_for_widths_the_limb_gemm_cannot_tile():
    """limbs-once needs Linear widths that are multiples of 128 (gemm_l3p's tiles); a model with another width (here the
    image tokenizer's MLP: 320; the hidden sizes are multiples of 256 for the LayerNorm kernel anyway) runs "bf16l3" on the
    splitting GEMM as before (l3p False), and "fp16l2" -- which exists on limbs only -- says so."""
    import pytest

    from sculptmate_amd import synth
    from sculptmate_amd.tsr.spec import make_cfg
    from sculptmate_amd.tsr.system import TSR

    cfg = make_cfg(vit_hidden=256, vit_layers=1, vit_heads=4, vit_mlp=320, channels=256, plane_size=8, heads=4, head_dim=64, layers=1,
                   cond_image_size=64)
    sd = synth.tsr_state(5, cfg)
    m = TSR(cfg, pos_embed_mode="size", precision="bf16l3")
    m.load_state_dict(sd)
    m.to(_dev())
    assert m.l3p is False
    img = synth.composite_rgb(synth.image_rgba(seed=3, size=64))
    with torch.no_grad():
        code = m([img], device=_dev())
    assert bool(torch.isfinite(code).all())
    m2 = TSR(cfg, pos_embed_mode="size", precision="fp16l2")
    m2.load_state_dict(sd)
    with pytest.raises(ValueError):
        m2.to(_dev())
