"""The add-on's input-side caller of the generators, with the reference's name and behaviour:
`preprocess_image(img_path, ratio=0.85, use_alpha=False)` (/root/reference/preprocessing.py:73-127; called at
GUIPanel.py:158 with ratio=0.75 for TripoSR and at :160 with ratio=0.85, use_alpha=True for StableFast-3D).

Background removal runs on the MI355X U^2-Net (sculptmate_amd.rembg); the rest is a handful of host-side array
operations on one image (bounding box of alpha > 0 with the reference's exclusive max, square padding, border so the
object fills `ratio` of the side, grey composite, LANCZOS to 1024^2).  The reference opens a new onnxruntime session on
every call (rembg/bg.py:200-201); here one session per device is kept for the process lifetime.
"""
import numpy as np
from PIL import Image

image_size = (1024, 1024)
_sessions = {}


def _session(device):
    from .rembg.session import new_session

    key = str(device)
    if key not in _sessions:
        _sessions[key] = new_session("u2net", device=device)
    return _sessions[key]


def _cutout(raw, session=None, device="cuda:0"):
    """rembg's remove() on the HIP U^2-Net -> RGBA PIL image."""
    from .rembg.bg import remove

    return remove(raw, session=session if session is not None else _session(device))


def _centre_pad(a, side):
    """Zero-pad H x W x C to side x side; an odd remainder goes to the bottom / right."""
    top, left = (side - a.shape[0]) // 2, (side - a.shape[1]) // 2
    out = np.zeros((side, side, a.shape[2]), a.dtype)
    out[top:top + a.shape[0], left:left + a.shape[1]] = a
    return out


def frame_foreground(rgba: np.ndarray, ratio: float) -> np.ndarray:
    """uint8 RGBA cut-out -> square RGBA with the object's bounding box scaled to `ratio` of the side
    (preprocessing.py:81-110).  The box is [min, max) on both axes, as the reference slices it."""
    ys, xs = np.nonzero(rgba[..., 3] > 0)
    if ys.size == 0:
        raise ValueError("preprocess_image: the cut-out is empty (alpha is zero everywhere)")
    fg = rgba[ys.min():ys.max(), xs.min():xs.max()]
    side = max(fg.shape[0], fg.shape[1])
    return _centre_pad(_centre_pad(fg, side), int(side / ratio))


def preprocess_image(img_path, ratio=0.85, use_alpha=False, session=None, device="cuda:0"):
    """Image file -> the generator's input.  use_alpha=True: framed RGBA at native size (StableFast-3D composites it
    itself); otherwise RGB composited on 0.5 grey and resized to 1024 x 1024, or None when the framed image is
    narrower than 250 px.  `session`: a rembg session to reuse (default: one U2netSession per device)."""
    raw = Image.open(img_path)
    if use_alpha:
        raw = raw.convert("RGBA")
    cut = _cutout(raw, session, device)
    framed = frame_foreground(np.array(cut), ratio)
    if use_alpha:
        return Image.fromarray(framed, mode="RGBA")
    x = framed.astype(np.float32) / 255.0
    rgb = x[:, :, :3] * x[:, :, 3:4] + (1 - x[:, :, 3:4]) * 0.5
    out = Image.fromarray((rgb * 255.0).astype(np.uint8))
    if out.size[0] < 250:
        return None
    return out.resize(image_size, Image.Resampling.LANCZOS)
