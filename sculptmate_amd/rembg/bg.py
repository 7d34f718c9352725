"""Background removal entry point with the signature and results of the reference's `rembg.bg.remove`
(rembg/bg.py:149-238), running U^2-Net on the MI355X.  Cut-outs are assembled with the same PIL primitives the
reference uses, so given the same mask the pixels are identical (tests/golden/rembg_prepost.npz).  Alpha matting
(pymatting) and `post_process_mask` (OpenCV morphology) are outside this package and raise."""
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
    if post_process_mask:
        raise NotImplementedError("post_process_mask needs OpenCV morphology (not part of this package)")
    as_alpha = kwargs.pop("putalpha", False)
    img = fix_image_orientation(img)
    session = session if session is not None else new_session("u2net", *args, **kwargs)
    pieces = []
    for mask in session.predict(img, *args, **kwargs):
        if only_mask:
            pieces.append(mask)
        else:
            pieces.append(putalpha_cutout(img, mask) if as_alpha else naive_cutout(img, mask))
    result = get_concat_v_multi(pieces) if pieces else img
    if bgcolor is not None and not only_mask:
        result = apply_background_color(result, bgcolor)
    return _encode(kind, result)
