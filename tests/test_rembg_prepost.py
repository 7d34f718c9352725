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


def test_post_process_mask_against_scipy_morphology_and_fixed_point_blur():
    """bg.post_process (the reference's rembg/bg.py:98-108: opening with the 3 x 3 elliptic element, 5 x 5 Gaussian sigma 2,
    threshold 127) restated in numpy.  OpenCV is not in this image (parity unpinned); the three steps are checked against
    scipy.ndimage: grey opening with the 4-neighbour cross and a border that never wins, the blur with the 8-bit weights
    [39 57 64 57 39] / 256 on a mirrored (reflect-101) border with one rounding, and against the ideal Gaussian away from the
    threshold."""
    from scipy import ndimage as ndi

    bg = _load_host_modules()["bg"]
    rng = np.random.default_rng(0)
    cross = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], bool)
    w8 = np.array([39, 57, 64, 57, 39]) / 256.0
    wg = np.exp(-np.arange(-2, 3) ** 2 / 8.0)
    wg /= wg.sum()
    assert np.array_equal(np.round(wg * 256), [39, 57, 64, 57, 39])
    for shape in ((64, 80), (33, 7), (5, 5), (2, 9), (320, 320)):
        soft = ndi.gaussian_filter(rng.random(shape), 2)
        soft = ((soft - soft.min()) / (soft.max() - soft.min()) * 255).astype(np.uint8)
        binary = np.where(soft > 128, 255, 0).astype(np.uint8)
        binary[rng.integers(0, shape[0], 6), rng.integers(0, shape[1], 6)] ^= 255      # speckle: isolated pixels
        for a in (soft, binary):
            out = bg.post_process(a)
            assert out.dtype == np.uint8 and out.shape == a.shape and set(np.unique(out)) <= {0, 255}
            opened = ndi.grey_dilation(ndi.grey_erosion(a, footprint=cross, mode="constant", cval=255), footprint=cross,
                                       mode="constant", cval=0)
            blur = ndi.correlate1d(ndi.correlate1d(opened.astype(np.float64), w8, axis=1, mode="mirror"), w8, axis=0, mode="mirror")
            assert np.array_equal(out, np.where(np.floor(blur + 0.5) < 127, 0, 255))
            ideal = ndi.correlate1d(ndi.correlate1d(opened.astype(np.float64), wg, axis=1, mode="mirror"), wg, axis=0, mode="mirror")
            far = np.abs(ideal - 126.5) > 0.5                                          # fixed-point weights move a value by < 0.5
            assert np.array_equal(out[far], np.where(ideal < 126.5, 0, 255)[far])
    # an isolated white pixel and a one-pixel line do not survive the opening; a solid block keeps its interior
    m = np.zeros((20, 20), np.uint8)
    m[3, 3] = 255
    m[10, 2:18] = 255
    assert not bg.post_process(m).any()
    m[12:19, 5:15] = 255
    out = bg.post_process(m)
    assert out[14:17, 7:13].all() and not out[:9].any()
    # constant masks are fixed points
    assert (bg.post_process(np.full((9, 11), 255, np.uint8)) == 255).all() and not bg.post_process(np.zeros((9, 11), np.uint8)).any()


def test_remove_applies_the_post_processed_mask():
    """remove(post_process_mask=True) hands the cleaned mask on like the reference (rembg/bg.py:206-208): only_mask returns it."""
    bg = _load_host_modules()["bg"]

    class OneMask:
        def predict(self, img, *a, **k):
            m = np.zeros((img.size[1], img.size[0]), np.uint8)
            m[8:40, 10:50] = 200
            m[2, 2] = 255
            return [Image.fromarray(m, mode="L")]

    img = Image.fromarray(np.full((48, 64, 3), 90, np.uint8), mode="RGB")
    raw = np.asarray(bg.remove(img, session=OneMask(), only_mask=True))
    cleaned = np.asarray(bg.remove(img, session=OneMask(), only_mask=True, post_process_mask=True))
    assert raw[2, 2] == 255 and cleaned[2, 2] == 0 and set(np.unique(cleaned)) == {0, 255}
    assert np.array_equal(cleaned, bg.post_process(raw))
    cut = np.asarray(bg.remove(img, session=OneMask(), post_process_mask=True))
    assert cut.shape == (48, 64, 4) and cut[20, 30, 3] == 255 and cut[2, 2, 3] == 0
