"""Host-side helpers mirroring /root/reference/TripoSR/tsr/utils.py (same names, same behaviour)."""
from typing import Any, List, Union

import numpy as np
import torch
import torch.nn.functional as F

try:  # PIL is optional for headless use
    import PIL.Image
except Exception:  # pragma: no cover
    PIL = None


def scale_tensor(dat, inp_scale, tgt_scale):
    """utils.py:222-231 -- kept operation for operation (fp32 rounding matters for parity)."""
    if inp_scale is None:
        inp_scale = (0, 1)
    if tgt_scale is None:
        tgt_scale = (0, 1)
    dat = (dat - inp_scale[0]) / (inp_scale[1] - inp_scale[0])
    dat = dat * (tgt_scale[1] - tgt_scale[0]) + tgt_scale[0]
    return dat


class ImagePreprocessor:
    """utils.py:62-112: uint8/PIL -> float32/255, antialiased bilinear resize to `size` on the host
    (the reference also does this on the CPU before `.to(device)`, system.py:94-96)."""

    def convert_and_resize(self, image, size: int):
        if PIL is not None and isinstance(image, PIL.Image.Image):
            image = torch.from_numpy(np.array(image).astype(np.float32) / 255.0)
        elif isinstance(image, np.ndarray):
            if image.dtype == np.uint8:
                image = torch.from_numpy(image.astype(np.float32) / 255.0)
            else:
                image = torch.from_numpy(image)
        elif isinstance(image, torch.Tensor):
            pass
        batched = image.ndim == 4
        if not batched:
            image = image[None, ...]
        if image.shape[1] != size or image.shape[2] != size:
            image = F.interpolate(image.permute(0, 3, 1, 2), (size, size), mode="bilinear",
                                  align_corners=False, antialias=True).permute(0, 2, 3, 1)
        if not batched:
            image = image[0]
        return image

    def __call__(self, image: Union[Any, List[Any]], size: int) -> torch.Tensor:
        if isinstance(image, (np.ndarray, torch.Tensor)) and image.ndim == 4:
            image = self.convert_and_resize(image, size)
        else:
            if not isinstance(image, list):
                image = [image]
            image = [self.convert_and_resize(im, size) for im in image]
            image = torch.stack(image, dim=0)
        return image
