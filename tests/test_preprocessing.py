"""sculptmate_amd/preprocessing.py against outputs of the reference's own preprocess_image
(tests/golden/make_preprocess_goldens.py); background removal is replaced by the stored cut-out on both sides."""
import os
import zlib

import numpy as np
import pytest
from PIL import Image

from conftest import GOLDEN
from sculptmate_amd import preprocessing


class _FixedCutout:
    """A rembg session stand-in whose mask is the stored alpha: bg.remove() then reproduces the stored cut-out."""

    def __init__(self, rgba):
        self.rgba = rgba

    def predict(self, img, *a, **k):
        return [Image.fromarray(self.rgba[..., 3], mode="L")]


@pytest.fixture()
def fixture_image(tmp_path):
    z = np.load(os.path.join(GOLDEN, "preprocess.npz"))

    def make(name):
        rgba = z[name + ".cutout"]
        path = str(tmp_path / (name + ".png"))
        Image.fromarray(rgba[..., :3], mode="RGB").save(path)          # the colour image remove() cuts out again
        return path, rgba

    return z, make


def _patched_remove(monkeypatch, rgba):
    monkeypatch.setattr(preprocessing, "_cutout", lambda raw, session=None, device=None: Image.fromarray(rgba, mode="RGBA"))


@pytest.mark.parametrize("name", ["a", "b"])
def test_preprocess_image_matches_reference(fixture_image, monkeypatch, name):
    z, make = fixture_image
    path, rgba = make(name)
    _patched_remove(monkeypatch, rgba)
    rgb = preprocessing.preprocess_image(path, ratio=0.75)
    assert rgb.size == (1024, 1024) and rgb.mode == "RGB"
    arr = np.asarray(rgb)
    assert np.array_equal(arr[::4, ::4], z[name + ".tripo_sub"])
    assert np.uint32(zlib.crc32(np.ascontiguousarray(arr).tobytes())) == z[name + ".tripo_crc"]
    framed = preprocessing.preprocess_image(path, ratio=0.85, use_alpha=True)
    assert framed.mode == "RGBA" and np.array_equal(np.asarray(framed), z[name + ".sf3d"])


def test_preprocess_image_small_and_empty(fixture_image, monkeypatch):
    z, make = fixture_image
    path, rgba = make("small")
    _patched_remove(monkeypatch, rgba)
    assert preprocessing.preprocess_image(path, ratio=0.75) is None        # < 250 px after framing, like the reference
    _patched_remove(monkeypatch, np.zeros_like(rgba))
    with pytest.raises(ValueError):
        preprocessing.preprocess_image(path, ratio=0.75)


def test_preprocess_image_through_bg_remove_with_a_session(fixture_image):
    """The real bg.remove() path (PIL compositing) with a session object: no GPU needed for a session stand-in."""
    z, make = fixture_image
    path, rgba = make("a")
    out = preprocessing.preprocess_image(path, ratio=0.85, use_alpha=True, session=_FixedCutout(rgba))
    a = np.asarray(out)
    assert out.mode == "RGBA" and a.shape[0] == a.shape[1]
    ys, xs = np.nonzero(a[..., 3] > 0)
    # the object spans about 0.85 of the side and is centred
    span = max(ys.max() - ys.min(), xs.max() - xs.min()) + 1
    assert abs(span / a.shape[0] - 0.85) < 0.02
