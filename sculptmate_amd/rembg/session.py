"""U2netSession -- the reference's rembg session (rembg/sessions/base.py:10-69, rembg/sessions/u2net.py:10-46) with the
onnxruntime InferenceSession replaced by the HIP U^2-Net.  Pre- and post-processing are the reference's, on the host:
LANCZOS resize to 320x320, divide by the image maximum, ImageNet mean/std, network, min-max normalisation of d0,
8-bit mask resized back with LANCZOS."""
import os
from typing import List

import numpy as np
import torch
from PIL import Image

from .u2net import U2Net

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)
SIZE = (320, 320)


def normalize(img, mean=MEAN, std=STD, size=SIZE) -> np.ndarray:
    """base.py:44-69 -> float32 [1, 3, H, W]"""
    im = img.convert("RGB").resize(size, Image.LANCZOS)
    im_ary = np.array(im)
    im_ary = im_ary / np.max(im_ary)
    tmp = np.zeros((im_ary.shape[0], im_ary.shape[1], 3))
    for c in range(3):
        tmp[:, :, c] = (im_ary[:, :, c] - mean[c]) / std[c]
    return np.expand_dims(tmp.transpose((2, 0, 1)), 0).astype(np.float32)


def prediction_to_mask(pred: np.ndarray, size) -> Image.Image:
    """u2net.py:34-44: pred = network output [1, H, W] (channel 0 of d0)"""
    ma, mi = np.max(pred), np.min(pred)
    pred = (pred - mi) / (ma - mi)
    pred = np.squeeze(pred)
    mask = Image.fromarray((pred * 255).astype("uint8"), mode="L")
    return mask.resize(size, Image.LANCZOS)


def load_weights(weights_path=None):
    """-> state dict (authors' parameter names) from .onnx / .safetensors / torch file; see U2netSession.__init__."""
    if weights_path is None:
        here = os.path.dirname(os.path.abspath(__file__))
        ckpt = os.path.join(os.path.dirname(os.path.dirname(here)), "checkpoints")
        tried = [os.path.join(ckpt, "u2net" + ext) for ext in (".onnx", ".pth", ".safetensors")]
        weights_path = next((p for p in tried if os.path.isfile(p)), None)
        if weights_path is None:
            raise FileNotFoundError("U2netSession: no weights at %s" % " or ".join(tried))
    elif not os.path.isfile(weights_path):
        raise FileNotFoundError("U2netSession: no weights at %s" % weights_path)
    if weights_path.endswith(".onnx"):
        from .onnx_weights import u2net_state_dict

        return u2net_state_dict(weights_path)
    if weights_path.endswith(".safetensors"):
        from safetensors.torch import load_file

        return load_file(weights_path)
    return torch.load(weights_path, map_location="cpu")


class U2netSession:
    def __init__(self, model_name: str = "u2net", device=None, state_dict=None, weights_path=None, *args, **kwargs):
        """device: a HIP device (default cuda:0).  Weights: `state_dict` (authors' parameter names, see spec.py) or a file
        at `weights_path`: the reference's own `u2net.onnx` (initialisers read by onnx_weights.py, no onnx package
        needed), a torch checkpoint or a safetensors file.  Default: <package>/../checkpoints/u2net.onnx -- the place the
        reference opens (rembg/sessions/base.py:38-42) -- then u2net.pth / u2net.safetensors beside it."""
        self.model_name = model_name
        self.device = torch.device(device if device is not None else "cuda:0")
        self.net = U2Net()
        if state_dict is None:
            state_dict = load_weights(weights_path)
        self.net.load_state_dict(state_dict)
        self.net.to(self.device)

    def predict(self, img, *args, **kwargs) -> List[Image.Image]:
        x = torch.from_numpy(normalize(img)[0]).to(self.device)
        d0 = self.net.forward(x).cpu().numpy()[None]
        return [prediction_to_mask(d0, img.size)]

    @classmethod
    def name(cls, *args, **kwargs):
        return "u2net"


def new_session(model_name: str = "u2net", providers=None, *args, **kwargs) -> U2netSession:
    """rembg/session_factory.py:10-44 (only the u2net session exists here)."""
    if model_name != "u2net":
        raise ValueError("only the 'u2net' session is implemented (the reference's default)")
    return U2netSession(model_name, *args, **kwargs)
