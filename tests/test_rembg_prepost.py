"""Pre/post-processing around U^2-Net vs outputs of the reference's own code (tests/golden/rembg_prepost.npz,
made by tests/golden/make_rembg_goldens.py).  CPU only: the modules under test are pure PIL / numpy host code, loaded by
file so that the HIP library is not needed."""
import importlib.util
import os
import sys
import types

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden", "rembg_prepost.npz")


def _load_host_modules():
    """session.py / bg.py import the HIP network at module level; give them a stand-in for that one import."""
    pkg = types.ModuleType("rb")
    pkg.__path__ = [os.path.join(ROOT, "sculptmate_amd", "rembg")]
    sys.modules["rb"] = pkg
    u = types.ModuleType("rb.u2net")
    u.U2Net = object
    sys.modules["rb.u2net"] = u
    mods = {}
    for n in ("session", "bg"):
        spec = importlib.util.spec_from_file_location("rb." + n, os.path.join(pkg.__path__[0], n + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules["rb." + n] = m
        spec.loader.exec_module(m)
        mods[n] = m
    return mods


def test_normalize_and_mask_postprocessing_match_reference():
    z = np.load(G)
    m = _load_host_modules()
    img = Image.fromarray(z["image"], mode="RGB")
    x = m["session"].normalize(img)
    assert x.shape == (1, 3, 320, 320) and x.dtype == np.float32
    assert np.array_equal(x, z["net_input"])
    mask = m["session"].prediction_to_mask(z["pred"][:, 0], img.size)
    assert np.array_equal(np.asarray(mask), z["mask"])
    cut = m["bg"].naive_cutout(img, mask)
    assert np.array_equal(np.asarray(cut), z["cutout"])
    assert np.array_equal(np.asarray(m["bg"].apply_background_color(cut, (10, 200, 30, 255))), z["cutout_bg"])
