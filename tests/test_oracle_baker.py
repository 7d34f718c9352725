"""Texture-baker oracle (oracle/baker.c) against the reference's own Python restatement (common.py)."""
import os

import numpy as np

from conftest import GOLDEN
from oracle import capi


def test_rasterize_and_interpolate_match_reference_common_py():
    g = np.load(os.path.join(GOLDEN, "baker.npz"))
    res = g["rast"].shape[0]
    rast = capi.bake_rasterize(g["uv"], g["faces"], res)
    # same triangle everywhere (non-overlapping charts), barycentrics to fp32 rounding
    assert np.array_equal(rast[..., 3], g["rast"][..., 3])
    np.testing.assert_allclose(rast[..., :3], g["rast"][..., :3], rtol=0, atol=2e-5)
    inter = capi.bake_interpolate(g["attr"], g["faces"], rast)
    np.testing.assert_allclose(inter, g["inter"], rtol=0, atol=5e-5)
    assert (rast[..., 3] >= 0).sum() > 0.5 * res * res


def test_empty_pixels_and_pixel_centre_convention():
    # one triangle covering the lower-left half of the unit square: pixel (x, y) samples (x/W, 1 - y/H)
    uv = np.array([[0, 0], [1, 0], [0, 1]], np.float32)
    f = np.array([[0, 1, 2]], np.int32)
    r = capi.bake_rasterize(uv, f, 8)
    assert r[0, 0, 3] == 0 and np.allclose(r[0, 0, :3], [0, 0, 1])      # sample (0, 1) = vertex 2
    assert r[7, 7, 3] == 0                                              # sample (7/8, 1/8): u = 0, on the edge
    assert r[1, 7, 3] == -1 and (r[1, 7, :3] == 0).all()                # sample (7/8, 7/8) outside
    inter = capi.bake_interpolate(np.eye(3, dtype=np.float32), f, r)
    assert (inter[1, 7] == 0).all() and np.allclose(inter[0, 0], [0, 0, 1])


def test_overlap_takes_lowest_triangle_index():
    uv = np.array([[0, 0], [1, 0], [0, 1], [1, 1]], np.float32)
    f = np.array([[0, 1, 2], [0, 1, 3], [0, 3, 2]], np.int32)
    r = capi.bake_rasterize(uv, f, 16)
    assert r[12, 2, 3] == 0  # covered by triangles 0 and 2 -> 0
