"""Precision dispatch and buffer pool shared by the host-side model mirrors (TSR, SF3D).

bf16: bf16 storage / fp32 accumulate on the MFMA kernels (what bench.py times).
fp32: every GEMM / attention / norm on the exact-fp32 parity kernels (the reference's own precision).
A subclass provides self.precision ("bf16" | "fp32"), self.device and self._buf = {}.
"""
import torch

from . import ops


class KernelEngine:
    def _b(self, name, shape, dtype, zero=False):
        key = (name, tuple(shape), dtype)
        t = self._buf.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
            self._buf[key] = t
        return t

    # ------------------------------------------------------------------ precision dispatch
    def _gemm(self, A, W, bias=None, residual=None, out_f32=None, out_bf16=None, out_t=None, M=None, epilogue=0,
              n_split=0):
        """One Linear: bf16 MFMA kernel (out_bf16 = activation buffer) or the fp32 parity kernel (every buffer fp32)."""
        if self.precision == "bf16":
            return ops.gemm(A, W, bias=bias, residual=residual, out_f32=out_f32, out_bf16=out_bf16, out_t=out_t, M=M,
                            epilogue=epilogue, n_split=n_split)
        out = out_f32 if out_f32 is not None else out_bf16
        return ops.gemm_f32(A, W, bias=bias, residual=residual, out=out, out_t=out_t, M=M, epilogue=epilogue,
                            n_split=n_split)

    def _attn(self, Q, K, Vt, O, Tq, Tk, heads, scale):
        if self.precision == "bf16":
            return ops.attention(Q, K, Vt, O, Tq, Tk, heads, scale)
        scores = self._b("attn_scores", (Tq, ((Tk + 15) // 16) * 16), torch.float32)
        return ops.attention_f32(Q, K, Vt, O, Tq, Tk, heads, scale, scores)

    def _ln(self, x, gamma, beta, eps, y):
        if self.precision == "bf16":
            return ops.layernorm(x, gamma, beta, eps, y=y)
        return ops.layernorm(x, gamma, beta, eps, y_f32=y)
