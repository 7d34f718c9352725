"""Golden vectors for the add-on's input-side caller of the hot path, preprocess_image (/root/reference/preprocessing.py:
73-127), produced by IMPORTING the reference module (build container only).

Stand-ins (no arithmetic): `cv2` (imported at module level, only used by the SAM helpers) and the package-relative
`.rembg.remove`, replaced by a function that returns the RGBA cut-out stored in the fixture -- background removal has
its own fixtures (rembg_prepost.npz); here the crop / pad / composite / LANCZOS arithmetic is pinned.
"""
import importlib.util
import os
import sys
import tempfile
import types
import zlib

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def cutout(seed, h, w):
    """A smooth RGBA cut-out with an off-centre elliptical alpha (soft edge) -- what remove() hands back."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    rgb = np.stack([127 + 100 * np.sin(xx / 17.0 + seed), 127 + 100 * np.cos(yy / 23.0), 127 + 90 * np.sin((xx + yy) / 31.0)], -1)
    rgb = np.clip(rgb + np.kron(rng.integers(-20, 21, (h // 8 + 1, w // 8 + 1, 3)), np.ones((8, 8, 1)))[:h, :w], 0, 255)  # blocky texture
    d = ((xx - 0.55 * w) / (0.30 * w)) ** 2 + ((yy - 0.40 * h) / (0.22 * h)) ** 2
    a = np.clip((1.0 - d) * 6.0, 0, 1) * 255
    return np.concatenate([rgb, a[..., None]], -1).astype(np.uint8)


def load_reference(cut):
    pkg = types.ModuleType("refaddon")
    pkg.__path__ = ["/root/reference"]
    sys.modules["refaddon"] = pkg
    rb = types.ModuleType("refaddon.rembg")
    rb.remove = lambda img, *a, **k: Image.fromarray(cut["rgba"], mode="RGBA")
    sys.modules["refaddon.rembg"] = rb
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    spec = importlib.util.spec_from_file_location("refaddon.preprocessing", "/root/reference/preprocessing.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["refaddon.preprocessing"] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    cut = {}
    ref = load_reference(cut)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "in.png")
        Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(path)           # content irrelevant: remove() is the stand-in
        for name, (h, w) in (("a", (300, 380)), ("b", (640, 400))):
            cut["rgba"] = cutout(len(name) + h, h, w)
            out[name + ".cutout"] = cut["rgba"]
            rgb = np.asarray(ref.preprocess_image(path, ratio=0.75))         # GUIPanel.py:158 (TripoSR)
            out[name + ".tripo_sub"] = rgb[::4, ::4]                         # every 4th pixel + a CRC of the whole 1024^2 image
            out[name + ".tripo_crc"] = np.uint32(zlib.crc32(np.ascontiguousarray(rgb).tobytes()))
            rgba = ref.preprocess_image(path, ratio=0.85, use_alpha=True)    # GUIPanel.py:160 (StableFast)
            out[name + ".sf3d"] = np.asarray(rgba)
        cut["rgba"] = cutout(9, 120, 100)                                    # too small: < 250 px after padding -> None
        assert ref.preprocess_image(path, ratio=0.75) is None
        out["small.cutout"] = cut["rgba"]
    np.savez_compressed(os.path.join(HERE, "preprocess.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
