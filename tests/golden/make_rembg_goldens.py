"""Golden vectors for the pre/post-processing around U^2-Net, produced by running the reference's own code
(rembg/sessions/base.py:44-69 `normalize`, rembg/sessions/u2net.py:16-46 `predict`, rembg/bg.py:33-47 `naive_cutout`)
in the build container.  Run: python tests/golden/make_rembg_goldens.py

onnxruntime is not installed and `checkpoints/u2net.onnx` is not in the checkout, so the ONNX session is replaced by a
stand-in that (a) records the tensor `predict` feeds to the network and (b) returns a fixed synthetic prediction --
everything around that call is the reference's code.  rembg/bg.py needs cv2 at import (absent); `naive_cutout` is four
lines of PIL calls and is exercised by loading bg.py with a cv2 stand-in that provides only the names it imports."""
import importlib.util
import os
import sys
import types

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/rembg"


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=None)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def main():
    ort = types.ModuleType("onnxruntime")
    ort.SessionOptions = object
    ort.InferenceSession = object
    ort.get_available_providers = lambda: []
    ort.set_default_logger_severity = lambda *_: None
    sys.modules["onnxruntime"] = ort
    pkg = types.ModuleType("refrembg")
    pkg.__path__ = [REF]
    sys.modules["refrembg"] = pkg
    spkg = types.ModuleType("refrembg.sessions")
    spkg.__path__ = [REF + "/sessions"]
    sys.modules["refrembg.sessions"] = spkg
    base = _load("refrembg.sessions.base", REF + "/sessions/base.py")
    u2 = _load("refrembg.sessions.u2net", REF + "/sessions/u2net.py")

    rng = np.random.default_rng(0)
    img_np = rng.integers(0, 256, (97, 131, 3), dtype=np.uint8)
    img_np[20:70, 30:100] = (img_np[20:70, 30:100] // 2 + 100).astype(np.uint8)
    img = Image.fromarray(img_np, mode="RGB")
    yy, xx = np.mgrid[0:320, 0:320]
    pred = (0.1 + 0.8 * np.exp(-(((yy - 150) / 70.0) ** 2 + ((xx - 170) / 90.0) ** 2))).astype(np.float32)
    pred = pred[None, None] + 0.01 * rng.standard_normal((1, 1, 320, 320)).astype(np.float32)
    captured = {}

    class _Input:
        name = "input.1"

    class _Inner:
        def get_inputs(self):
            return [_Input()]

        def run(self, _outs, feed):
            captured["x"] = feed["input.1"].copy()
            return [pred]

    sess = u2.U2netSession.__new__(u2.U2netSession)
    sess.inner_session = _Inner()
    masks = sess.predict(img)
    mask = np.asarray(masks[0])
    # naive_cutout (bg.py:33-47)
    cv2 = types.ModuleType("cv2")
    for n in ("BORDER_DEFAULT", "MORPH_ELLIPSE", "MORPH_OPEN"):
        setattr(cv2, n, 0)
    cv2.GaussianBlur = cv2.morphologyEx = lambda *a, **k: None
    cv2.getStructuringElement = lambda *a, **k: None
    sys.modules["cv2"] = cv2
    sf = types.ModuleType("refrembg.session_factory")
    sf.new_session = lambda *a, **k: None
    sys.modules["refrembg.session_factory"] = sf
    spkg.sessions_class = []
    bg = _load("refrembg.bg", REF + "/bg.py")
    cut = np.asarray(bg.naive_cutout(img, masks[0]))
    colored = np.asarray(bg.apply_background_color(bg.naive_cutout(img, masks[0]), (10, 200, 30, 255)))
    np.savez_compressed(os.path.join(HERE, "rembg_prepost.npz"), image=img_np, net_input=captured["x"], pred=pred, mask=mask,
                        cutout=cut, cutout_bg=colored)
    print("rembg_prepost:", captured["x"].shape, mask.shape, cut.shape)


if __name__ == "__main__":
    main()
