"""Background removal entry point with the signature and results of the reference's `rembg.bg.remove`
(rembg/bg.py:149-238), running U^2-Net on the MI355X.  Cut-outs are assembled with the same PIL primitives the
reference uses, so given the same mask the pixels are identical (tests/golden/rembg_prepost.npz).  `post_process_mask` is
restated in numpy from the three OpenCV calls the reference makes (OpenCV is not in this image: that piece is parity-unpinned,
tested against scipy.ndimage); alpha matting (pymatting's closed-form solver) is outside this package and raises."""
import io
from enum import Enum
from typing import Any, List, Optional, Tuple, Union

import numpy as np
from PIL import Image, ImageOps

from .session import U2netSession, new_session

RGBA = Tuple[int, int, int, int]


class ReturnType(Enum):
    BYTES = 0
    PILLOW = 1
    NDARRAY = 2


# ------------------------------------------------------------------------------------------------ cut-out primitives
def naive_cutout(img: Image.Image, mask: Image.Image) -> Image.Image:
    """The picture where the mask is white, transparent black elsewhere (mask-weighted blend against an empty RGBA)."""
    transparent = Image.new("RGBA", img.size, 0)
    return Image.composite(img, transparent, mask)


def putalpha_cutout(img: Image.Image, mask: Image.Image) -> Image.Image:
    """The mask becomes the picture's alpha channel (in place, like the reference)."""
    img.putalpha(mask)
    return img


def get_concat_v(top: Image.Image, bottom: Image.Image) -> Image.Image:
    sheet = Image.new("RGBA", (top.width, top.height + bottom.height))
    for tile, y in ((top, 0), (bottom, top.height)):
        sheet.paste(tile, (0, y))
    return sheet


def get_concat_v_multi(imgs: List[Image.Image]) -> Image.Image:
    """Stack the cut-outs of all masks vertically (one mask -> that cut-out); consumes the list like the reference."""
    stacked = imgs.pop(0)
    while imgs:
        stacked = get_concat_v(stacked, imgs.pop(0))
    return stacked


def apply_background_color(img: Image.Image, color: RGBA) -> Image.Image:
    backdrop = Image.new("RGBA", img.size, tuple(color))
    backdrop.paste(img, mask=img)
    return backdrop


def fix_image_orientation(img: Image.Image) -> Image.Image:
    return ImageOps.exif_transpose(img)


# ------------------------------------------------------------------------------------------------ mask post-processing
_GAUSS5_SIGMA2_Q8 = np.array([39, 57, 64, 57, 39], dtype=np.int64)  # exp(-x^2 / 8) / sum, x = -2..2, in 1/256 (sums to 256)


def _shift_fill(a: np.ndarray, dy: int, dx: int, fill: int) -> np.ndarray:
    """a moved by (dy, dx) with `fill` where the source lies outside the image."""
    out = np.full_like(a, fill)
    h, w = a.shape
    ys, yd = (slice(0, h - dy), slice(dy, h)) if dy >= 0 else (slice(-dy, h), slice(0, h + dy))
    xs, xd = (slice(0, w - dx), slice(dx, w)) if dx >= 0 else (slice(-dx, w), slice(0, w + dx))
    out[yd, xd] = a[ys, xs]
    return out


def post_process(mask: np.ndarray) -> np.ndarray:
    """The reference's mask clean-up (rembg/bg.py:98-108), uint8 [H, W] -> uint8 {0, 255}:
      1. morphological opening with the 3 x 3 elliptic structuring element -- which at that size is the 4-neighbour cross --
         erosion then dilation, pixels outside the image never winning (OpenCV's default morphology border);
      2. 5 x 5 Gaussian blur, sigma 2, mirrored border without the edge pixel (BORDER_REFLECT_101), in the 8-bit fixed-point form
         OpenCV uses for uint8 images: weights [39 57 64 57 39] / 256 per axis, one rounding at the end;
      3. threshold: below 127 -> 0, else 255.
    Restated from the published behaviour of morphologyEx / GaussianBlur; cv2 is not available here to pin it."""
    m = np.ascontiguousarray(mask)
    if m.ndim != 2:
        raise ValueError("post_process expects a single-channel mask")
    m = m.astype(np.uint8)
    cross = ((0, 0), (-1, 0), (1, 0), (0, -1), (0, 1))
    er = m
    for dy, dx in cross[1:]:
        er = np.minimum(er, _shift_fill(m, dy, dx, 255))
    op = er
    for dy, dx in cross[1:]:
        op = np.maximum(op, _shift_fill(er, dy, dx, 0))
    acc = op.astype(np.int64)
    for axis in (1, 0):  # rows, then columns; reflect-101 padding of two pixels
        pad = [(0, 0), (0, 0)]
        pad[axis] = (2, 2)
        padded = np.pad(acc, pad, mode="reflect") if acc.shape[axis] > 1 else np.pad(acc, pad, mode="edge")
        n = acc.shape[axis]
        acc = sum(int(w) * np.take(padded, range(k, k + n), axis=axis) for k, w in enumerate(_GAUSS5_SIGMA2_Q8))
    blurred = (acc + 32768) >> 16
    return np.where(blurred < 127, 0, 255).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ entry point
def _decode(data) -> Tuple[ReturnType, Image.Image]:
    if isinstance(data, Image.Image):
        return ReturnType.PILLOW, data
    if isinstance(data, bytes):
        return ReturnType.BYTES, Image.open(io.BytesIO(data))
    if isinstance(data, np.ndarray):
        return ReturnType.NDARRAY, Image.fromarray(data)
    raise ValueError("Input type {} is not supported.".format(type(data)))


def _encode(kind: ReturnType, picture: Image.Image):
    if kind is ReturnType.PILLOW:
        return picture
    if kind is ReturnType.NDARRAY:
        return np.asarray(picture)
    buffer = io.BytesIO()
    picture.save(buffer, "PNG")
    return buffer.getvalue()


def remove(data: Union[bytes, Image.Image, np.ndarray], alpha_matting: bool = False,
           alpha_matting_foreground_threshold: int = 240, alpha_matting_background_threshold: int = 10,
           alpha_matting_erode_size: int = 10, session: Optional[U2netSession] = None, only_mask: bool = False,
           post_process_mask: bool = False, bgcolor: Optional[RGBA] = None, *args: Optional[Any], **kwargs: Optional[Any]):
    """Same arguments and return types as the reference: the result comes back in the form the input came in
    (PIL image, ndarray, or PNG bytes)."""
    kind, img = _decode(data)
    if alpha_matting:
        raise NotImplementedError("alpha matting needs pymatting (not part of this package)")
    as_alpha = kwargs.pop("putalpha", False)
    img = fix_image_orientation(img)
    session = session if session is not None else new_session("u2net", *args, **kwargs)
    pieces = []
    for mask in session.predict(img, *args, **kwargs):
        if post_process_mask:
            mask = Image.fromarray(post_process(np.array(mask)))
        if only_mask:
            pieces.append(mask)
        else:
            pieces.append(putalpha_cutout(img, mask) if as_alpha else naive_cutout(img, mask))
    result = get_concat_v_multi(pieces) if pieces else img
    if bgcolor is not None and not only_mask:
        result = apply_background_color(result, bgcolor)
    return _encode(kind, result)
