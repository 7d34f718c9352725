"""Host-side helpers with the names and behaviour of /root/reference/TripoSR/tsr/utils.py:
`scale_tensor` (utils.py:222-231) and `ImagePreprocessor` (utils.py:62-112)."""
from typing import Any, List, Sequence, Union

import numpy as np
import torch
import torch.nn.functional as F

try:  # PIL is optional for headless use
    import PIL.Image as _PILImage
except Exception:  # pragma: no cover
    _PILImage = None


def scale_tensor(dat, inp_scale, tgt_scale):
    """Affine map of `dat` from the interval inp_scale to tgt_scale (None means (0, 1)).  The four operations --
    subtract, divide, multiply, add -- are applied in exactly this order: the fp32 rounding of the lattice
    coordinates is part of the parity contract."""
    lo_in, hi_in = (0, 1) if inp_scale is None else (inp_scale[0], inp_scale[1])
    lo_out, hi_out = (0, 1) if tgt_scale is None else (tgt_scale[0], tgt_scale[1])
    unit = (dat - lo_in) / (hi_in - lo_in)
    return unit * (hi_out - lo_out) + lo_out


def _as_float_hwc(image) -> torch.Tensor:
    """PIL image or uint8 array -> float32 in [0, 1]; float arrays / tensors pass through unchanged."""
    if _PILImage is not None and isinstance(image, _PILImage.Image):
        image = np.array(image)
        return torch.from_numpy(image.astype(np.float32) / 255.0)
    if isinstance(image, np.ndarray):
        return torch.from_numpy(image.astype(np.float32) / 255.0) if image.dtype == np.uint8 else torch.from_numpy(image)
    return image


class ImagePreprocessor:
    """Conditioning images -> float32 [B, size, size, C] on the host, like the reference does before `.to(device)`
    (system.py:94-96): antialiased bilinear resampling (align_corners=False); an image already at `size` is untouched."""

    def convert_and_resize(self, image, size: int) -> torch.Tensor:
        x = _as_float_hwc(image)
        single = x.ndim == 3
        x = x.unsqueeze(0) if single else x
        if tuple(x.shape[1:3]) != (size, size):
            nchw = x.permute(0, 3, 1, 2)
            nchw = F.interpolate(nchw, (size, size), mode="bilinear", align_corners=False, antialias=True)
            x = nchw.permute(0, 2, 3, 1)
        return x[0] if single else x

    def __call__(self, image: Union[Any, Sequence[Any]], size: int) -> torch.Tensor:
        already_batched = isinstance(image, (np.ndarray, torch.Tensor)) and image.ndim == 4
        if already_batched:
            return self.convert_and_resize(image, size)
        items: List[Any] = list(image) if isinstance(image, list) else [image]
        return torch.stack([self.convert_and_resize(item, size) for item in items], dim=0)
