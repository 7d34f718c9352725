"""ViT position-embedding interpolation (host side, once per image size, input independent).

HF `ViTEmbeddings.interpolate_pos_encoding` resizes the 14x14 grid of patch position embeddings to
the actual patch grid with torch's bicubic `F.interpolate(align_corners=False)` (cubic convolution,
A = -0.75, border indices clamped).  Two variants exist (SURVEY.md section 7 "HF version skew"):
  * transformers 4.38 (what the reference pins, __init__.py:37): scale_factor = (n + 0.1) / 14, the
    coordinate map uses that scale factor           -> mode "scale_factor"  (default for checkpoints)
  * transformers >= 4.4x / 5.x: size=(n, n), scale = n / 14 -> mode "size"  (matches goldens made here)
Called from DINOSingleImageTokenizer (reference: TripoSR/tsr/models/tokenizers/image.py:49-51).
"""
import math

import numpy as np


def _cubic_weights(t, A=-0.75):
    # torch's cubic_convolution1 / cubic_convolution2 (upsample bicubic)
    def c1(x):
        return ((A + 2) * x - (A + 3)) * x * x + 1

    def c2(x):
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A

    return np.stack([c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t)], -1)


def _resize_axis(x, out_size, scale, axis):
    """x float32; 1-D bicubic resample along `axis` with src = (dst + 0.5)/scale - 0.5."""
    in_size = x.shape[axis]
    dst = np.arange(out_size, dtype=np.float32)
    src = (dst + np.float32(0.5)) * np.float32(1.0 / scale) - np.float32(0.5)
    i0 = np.floor(src)
    t = (src - i0).astype(np.float32)
    w = _cubic_weights(t).astype(np.float32)  # [out, 4]
    idx = np.clip(i0.astype(np.int64)[:, None] + np.arange(-1, 3)[None, :], 0, in_size - 1)  # [out, 4]
    xm = np.moveaxis(x, axis, -1)  # [..., in]
    g = xm[..., idx]  # [..., out, 4]
    out = (g * w).sum(-1, dtype=np.float32)
    return np.moveaxis(out, -1, axis)


def interpolate_pos_embedding(pos, n_side, mode="scale_factor"):
    """pos [1, 1 + g*g, D] float32 -> [1 + n_side*n_side, D] float32 (CLS position kept)."""
    pos = np.asarray(pos, np.float32)
    D = pos.shape[-1]
    n_pos = pos.shape[1] - 1
    g = int(math.sqrt(n_pos))
    assert g * g == n_pos
    cls = pos[0, :1]
    if n_side == g:
        return np.concatenate([cls, pos[0, 1:]], 0)
    grid = pos[0, 1:].reshape(g, g, D)
    if mode == "scale_factor":
        scale = (n_side + 0.1) / g
    elif mode == "size":
        scale = n_side / g
    else:
        raise ValueError("mode must be 'scale_factor' (transformers 4.38) or 'size' (>= 4.4x)")
    # torch computes the separable bicubic as rows-then-columns of the 4x4 neighbourhood; the
    # interpolation is separable so the order only changes fp32 rounding
    y = _resize_axis(grid, n_side, scale, 0)
    y = _resize_axis(y, n_side, scale, 1)
    return np.concatenate([cls, y.reshape(n_side * n_side, D)], 0).astype(np.float32)
