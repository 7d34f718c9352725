"""Precision dispatch and buffer pool shared by the host-side model mirrors (TSR, SF3D).

bf16:   bf16 storage / fp32 accumulate on the MFMA kernels (what bench.py times).
fp32:   every GEMM / attention / norm on the exact-fp32 parity kernels (the reference's own precision).
fp16l2: "bf16l3" with the hot Linears (limbs once, below) on TWO fp16 limbs per operand: 22 significant bits, three products per
        multiply instead of six; weights are stored pre-scaled into the fp16 range, activations must stay below 65504 in magnitude
        (they are LayerNorm outputs, attention outputs, GELU / GEGLU products; a value beyond it makes the result non-finite, which
        the model checks for and answers with its three-limb twin).  The attention products run on two fp16 limbs as well
        (csrc/attention_l2.hip: scores and P.V, three products each) wherever the pipelined limb attention runs -- the backbone;
        the image tokenizer's small launches and the cold Linears stay on three bf16 limbs.
bf16l3: fp32 storage and fp32 norms / softmax like "fp32", but every matrix product (Linears, QK^T, PV) on the bf16 matrix pipe
        through the exact three-limb split of both operands, fp32 accumulate (csrc/gemm_l3.hip): fp32-equivalent, ~6x faster.
        "Limbs once" (default in this mode, SCULPT_L3_TILE=split restores the form that splits inside every GEMM): a subclass that
        stores its hot Linear weights as ops.Limbs (split at load time) and allocates the activations that only feed such a
        Linear with _lt() gets them written as limbs by their producers -- LayerNorm, attention, the GELU / GEGLU epilogue -- and
        multiplied by csrc/gemm_l3p.hip: the same products in the same order, bit-identical results, no split in any K loop.
A subclass provides self.precision ("bf16" | "fp32" | "bf16l3" | "fp16l2"), self.device and self._buf = {}.
"""
import os

import torch

from . import _lib, ops

BF16 = torch.bfloat16


def prepare_ln_linear(L, key, W, bias, gamma, beta, fold, to_weight, to_f32, q_rows=0, q_scale=1.0):
    """A Linear fed by a LayerNorm, prepared for KernelEngine._ln_gemm.  fold (bf16 mode): the LayerNorm is folded into the
    GEMM (ops.fold_layernorm): L[key] = bf16(W * gamma), L[key_b] = bias + W . beta, L[key_cs] = column sums.
    Otherwise (fp32 mode): plain weights + the LayerNorm's own parameters for the stand-alone kernel.
    q_rows / q_scale (fold only): the first q_rows output rows are an attention's query projection and carry
    softmax_scale * log2(e) -- multiplied in BEFORE the bf16 rounding of the weights, so q is stored as bf16(c q), one rounding as
    before -- which lets the attention kernel take its scores as exponents of 2 (scale = 0 entry)."""
    W = torch.as_tensor(W)
    if fold:
        if q_rows:
            W = W.clone().to(torch.float32)
            W[:q_rows] *= q_scale
            if bias is not None:
                bias = torch.as_tensor(bias).clone().to(torch.float32)
                bias[:q_rows] *= q_scale
        Wp, bp, cs = ops.fold_layernorm(W, bias, gamma, beta)
        L[key], L[key + "_b"], L[key + "_cs"] = to_weight(Wp), to_f32(bp), to_f32(cs)
    else:
        L[key], L[key + "_b"] = to_weight(W), (None if bias is None else to_f32(bias))
        L[key + "_ln"] = (to_f32(gamma), to_f32(beta))


class KernelEngine:
    def _b(self, name, shape, dtype, zero=False):
        key = (name, tuple(shape), dtype)
        t = self._buf.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
            self._buf[key] = t
        return t

    def _lt(self, name, rows, cols):
        """A cached limb-tiled activation buffer (ops.Limbs), zero-initialised once (pad rows stay finite)."""
        fmt = getattr(self, "limb_format", "bf16x3")
        key = (name, "limbs", rows, cols, fmt)
        t = self._buf.get(key)
        if t is None:
            t = self._buf[key] = ops.Limbs(rows, cols, self.device, zero=True, fmt=fmt)
        return t

    # ------------------------------------------------------------------ precision dispatch
    def _gemm(self, A, W, bias=None, residual=None, out_f32=None, out_bf16=None, out_t=None, M=None, epilogue=0,
              n_split=0):
        """One Linear: bf16 MFMA kernel (out_bf16 = activation buffer) or the fp32 parity kernel (every buffer fp32)."""
        if self.precision == "bf16":
            return ops.gemm(A, W, bias=bias, residual=residual, out_f32=out_f32, out_bf16=out_bf16, out_t=out_t, M=M,
                            epilogue=epilogue, n_split=n_split)
        out = out_f32 if out_f32 is not None else out_bf16
        if isinstance(W, ops.Limbs):   # limbs once: both operands arrive split (csrc/gemm_l3p.hip)
            assert isinstance(A, ops.Limbs) and A.cols == W.cols, "a limb-tiled weight needs a limb-tiled activation of the same K"
            N = W.rows // 2 if epilogue == ops._lib.EPI_GEGLU else W.rows
            lt = isinstance(out, ops.Limbs)
            return ops.gemm_l3p(A, W, A.rows if M is None else M, N, W.cols, bias=bias, residual=residual, out=None if lt else out,
                                out_t=out_t, n_split=n_split, out_lt=out if lt else None, epilogue=epilogue)
        return ops.gemm_f32(A, W, bias=bias, residual=residual, out=out, out_t=out_t, M=M, epilogue=epilogue,
                            n_split=n_split, l3=self.l3)

    @property
    def l3(self):
        """fp32 storage with the matrix products on the 16-bit matrix pipe through limbs ("bf16l3": three bf16 limbs everywhere;
        "fp16l2": the hot Linears on two fp16 limbs -- half the products --, everything else as "bf16l3")."""
        return self.precision in ("bf16l3", "fp16l2")

    def _attn(self, Q, K, Vt, O, Tq, Tk, heads, scale, batch=1, q_bs=0, k_bs=0, vt_bs=0, o_bs=0):
        """batch > 1: `batch` independent attentions, entry b at Q + b*q_bs, K + b*k_bs, Vt + b*vt_bs (a column offset), O + b*o_bs
        (elements): one launch in bf16 mode, a loop over the entries in the fp32 parity modes (there one attention is three
        launches over all heads: scores = alpha Q K^T into an fp32 [heads][Tq][Tk] scratch, row softmax, P V)."""
        if self.precision == "bf16":
            return ops.attention(Q, K, Vt, O, Tq, Tk, heads, scale, batch, q_bs, k_bs, vt_bs, o_bs)
        # bf16l3: one fused launch, no score matrix; SCULPT_ATTN_FORM=l3unfused keeps the three-launch composition (A/B)
        fused = self.l3 and not _lib.form_has("SCULPT_ATTN_FORM", "l3unfused")
        if fused:
            chunks = [(0, heads, None)]
        else:
            # the [heads][Tq][Tk] fp32 score scratch is capped (ATTN_SCRATCH_BYTES): heads go through it in chunks, and ONE
            # grow-only buffer per stream serves every attention shape of the model (SF3D's 27 648 x 3 089 fuse attentions would
            # otherwise keep 2 x 5.6 GB resident)
            ld = ((Tk + 31) // 32) * 32
            per_head = Tq * ld
            n = max(1, min(heads, self.ATTN_SCRATCH_BYTES // (4 * per_head)))
            # ... per HIP stream: TSR.encode_image runs the image-independent head of the backbone on a second stream beside the
            # image tokenizer, and both come through here
            key = ("attn_scores_flat", torch.cuda.current_stream(self.device).cuda_stream)
            flat = self._buf.get(key)
            if flat is None or flat.numel() < n * per_head:
                flat = self._buf[key] = torch.empty(n * per_head, dtype=torch.float32, device=self.device)
            chunks = [(h0, min(n, heads - h0), flat[:min(n, heads - h0) * per_head].view(min(n, heads - h0), Tq, ld))
                      for h0 in range(0, heads, n)]
        if batch > 1:
            assert q_bs % Q.stride(0) == 0 and k_bs % K.stride(0) == 0 and vt_bs < Vt.stride(0)
            assert o_bs % (O.cols if isinstance(O, ops.Limbs) else O.stride(0)) == 0
            Tkp = ((Tk + 31) // 32) * 32 if self.l3 else ((Tk + 15) // 16) * 16
            assert (batch - 1) * vt_bs + Tkp <= Vt.shape[1], "V^T too narrow for the last batch entry (%d + %d > %d)" % (
                (batch - 1) * vt_bs, Tkp, Vt.shape[1])
        h2 = self.precision == "fp16l2"   # both products on two fp16 limbs where the pipelined form runs
        if fused and batch > 1:   # one launch over batch x heads (grid z), fp32 or limb output
            return ops.attention_f32_l3_batched(Q, K, Vt, O, Tq, Tk, heads, scale, batch, q_bs, k_bs, vt_bs, o_bs, two_fp16_limbs=h2)
        for b in range(batch):
            q, k = Q[b * q_bs // Q.stride(0):] if b else Q, K[b * k_bs // K.stride(0):] if b else K
            vt = Vt[:, b * vt_bs:] if b else Vt
            if isinstance(O, ops.Limbs):   # fused three-limb attention writing limbs: entry b starts at row b * o_bs / cols
                assert fused
                ops.attention_f32(q, k, vt, O, Tq, Tk, heads, scale, None, l3=True, o_row0=b * o_bs // O.cols, two_fp16_limbs=h2)
                continue
            o = O[b * o_bs // O.stride(0):] if b else O
            for h0, nh, scores in chunks:
                ops.attention_f32(q[:, 64 * h0:], k[:, 64 * h0:], vt[64 * h0:], o[:, 64 * h0:], Tq, Tk, nh, scale, scores, l3=self.l3)

    ATTN_SCRATCH_BYTES = 512 << 20

    def _ln(self, x, gamma, beta, eps, y):
        if self.precision == "bf16":
            return ops.layernorm(x, gamma, beta, eps, y=y)
        return ops.layernorm(x, gamma, beta, eps, y_f32=y)

    # ---- residual stream with the LayerNorm fold (bf16 mode) ------------------------------------------------------
    # Every LayerNorm of the two transformers sits between a GEMM that writes the residual stream h and a GEMM that
    # consumes LN(h).  In bf16 mode the producer also writes bf16(h) and per-row statistics of 32-column slices, and the
    # consumer applies mean / rstd in its epilogue with gamma / beta folded into its weights (sculpt_gemm_bf16_ln): no
    # LayerNorm launch, no normalised copy.  fp32 parity mode keeps the stand-alone LayerNorm kernel.
    def _stream_state(self, name, h):
        T, D = h.shape
        st = {"h": h, "name": name}
        if self.precision == "bf16":
            st["hb"] = self._b(name + "_hb", (T, D), BF16)
            st["stats"] = self._b(name + "_stats", (D // ops.LN_SLOT, T, 2), torch.float32)  # slice-major
        else:
            st["xn"] = self._b(name + "_xn", (T, D), torch.float32)
        return st

    def _state_from(self, h, name="bb"):
        """Stream state for a residual stream given as a plain fp32 tensor (tests, external callers)."""
        st = self._stream_state(name, h)
        self._stats_of(st)
        return st

    def _stats_of(self, st):
        if self.precision == "bf16":
            ops.row_slice_stats(st["h"], st["stats"], st["hb"])

    def _ln_gemm(self, st, L, key, eps, **kw):
        """Linear(LayerNorm(h)) with the weights prepared by ln_linear()."""
        if self.precision == "bf16":
            return ops.gemm(st["hb"], L[key], bias=L[key + "_b"], ln_stats=st["stats"], ln_colsum=L[key + "_cs"], ln_eps=eps, **kw)
        g, b = L[key + "_ln"]
        if isinstance(L[key], ops.Limbs):   # limbs once: the LayerNorm writes the GEMM's operand as limbs
            xn = self._lt(st["name"] + "_xn", st["h"].shape[0], st["h"].shape[1])
            ops.layernorm(st["h"], g, b, eps, y_lt=xn)
            return self._gemm(xn, L[key], bias=L[key + "_b"], **kw)
        ops.layernorm(st["h"], g, b, eps, y_f32=st["xn"])
        return self._gemm(st["xn"], L[key], bias=L[key + "_b"], **kw)

    def _res_gemm(self, st, A, W, bias):
        """h += A . W^T + bias (in place); bf16 mode also refreshes bf16(h) and the slice statistics."""
        h = st["h"]
        if self.precision == "bf16":
            return ops.gemm(A, W, bias=bias, residual=h, out_f32=h, out_bf16=st["hb"], stats_out=st["stats"])
        return self._gemm(A, W, bias=bias, residual=h, out_f32=h)
