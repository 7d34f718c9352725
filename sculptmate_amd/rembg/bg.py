"""remove() -- the reference's background removal entry point (rembg/bg.py:149-238) on top of the HIP U^2-Net session.
Cut-out helpers are the reference's PIL calls (bg.py:33-95, 110-125).  alpha matting (pymatting) and
post_process_mask (OpenCV morphology) are not available here and raise."""
import io
from enum import Enum
from typing import Any, List, Optional, Tuple, Union

import numpy as np
from PIL import Image, ImageOps

from .session import U2netSession, new_session


class ReturnType(Enum):
    BYTES = 0
    PILLOW = 1
    NDARRAY = 2


def naive_cutout(img, mask):
    empty = Image.new("RGBA", (img.size), 0)
    return Image.composite(img, empty, mask)


def putalpha_cutout(img, mask):
    img.putalpha(mask)
    return img


def get_concat_v(img1, img2):
    dst = Image.new("RGBA", (img1.width, img1.height + img2.height))
    dst.paste(img1, (0, 0))
    dst.paste(img2, (0, img1.height))
    return dst


def get_concat_v_multi(imgs: List):
    pivot = imgs.pop(0)
    for im in imgs:
        pivot = get_concat_v(pivot, im)
    return pivot


def apply_background_color(img, color: Tuple[int, int, int, int]):
    r, g, b, a = color
    colored_image = Image.new("RGBA", img.size, (r, g, b, a))
    colored_image.paste(img, mask=img)
    return colored_image


def fix_image_orientation(img):
    return ImageOps.exif_transpose(img)


def remove(data: Union[bytes, Image.Image, np.ndarray], alpha_matting: bool = False,
           alpha_matting_foreground_threshold: int = 240, alpha_matting_background_threshold: int = 10,
           alpha_matting_erode_size: int = 10, session: Optional[U2netSession] = None, only_mask: bool = False,
           post_process_mask: bool = False, bgcolor: Optional[Tuple[int, int, int, int]] = None, *args: Optional[Any],
           **kwargs: Optional[Any]):
    if isinstance(data, Image.Image):
        return_type, img = ReturnType.PILLOW, data
    elif isinstance(data, bytes):
        return_type, img = ReturnType.BYTES, Image.open(io.BytesIO(data))
    elif isinstance(data, np.ndarray):
        return_type, img = ReturnType.NDARRAY, Image.fromarray(data)
    else:
        raise ValueError("Input type {} is not supported.".format(type(data)))
    if alpha_matting:
        raise NotImplementedError("alpha matting needs pymatting (not part of this package)")
    if post_process_mask:
        raise NotImplementedError("post_process_mask needs OpenCV morphology (not part of this package)")
    putalpha = kwargs.pop("putalpha", False)
    img = fix_image_orientation(img)
    if session is None:
        session = new_session("u2net", *args, **kwargs)
    masks = session.predict(img, *args, **kwargs)
    cutouts = []
    for mask in masks:
        if only_mask:
            cutout = mask
        elif putalpha:
            cutout = putalpha_cutout(img, mask)
        else:
            cutout = naive_cutout(img, mask)
        cutouts.append(cutout)
    cutout = img
    if len(cutouts) > 0:
        cutout = get_concat_v_multi(cutouts)
    if bgcolor is not None and not only_mask:
        cutout = apply_background_color(cutout, bgcolor)
    if ReturnType.PILLOW == return_type:
        return cutout
    if ReturnType.NDARRAY == return_type:
        return np.asarray(cutout)
    bio = io.BytesIO()
    cutout.save(bio, "PNG")
    bio.seek(0)
    return bio.read()
