"""oracle/sf3d_unwrap_ref.py against golden vectors produced stage by stage by the reference's own Unwrapper methods
(tests/golden/make_sf3d_unwrap_goldens.py)."""
import os

import numpy as np
import pytest

from oracle import sf3d_unwrap_ref as U

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sf3d_unwrap.npz")


@pytest.fixture(scope="module")
def z():
    return np.load(G)


@pytest.mark.parametrize("name", ["ell", "tor"])
def test_stages_match_reference(z, name):
    g = lambda k: z[name + "." + k]  # noqa: E731
    rp, rn, faces = g("rot_pos"), g("rot_nrm"), g("faces")
    bbox = np.stack([rp.min(0), rp.max(0)], 0)
    uv, chart = U.box_project(rp, rn, faces, bbox)
    assert np.array_equal(chart, g("face_index"))
    assert np.abs(uv - g("uv_box")).max() < 1e-6
    tang = U.vertex_tangents(rp, rn, faces, g("uv_box"))
    assert np.abs(tang - g("tangents")).max() < 2e-5
    uv_rot, _ = U.rotate_charts(rp, rn, faces, g("uv_box"), chart)
    assert np.abs(uv_rot - g("uv_rot")).max() < 2e-5
    ox, oy, dx, dy = U.slice_offset_and_scale(g("assigned"))
    for a, k in ((ox, "offset_x"), (oy, "offset_y"), (dx, "div_x"), (dy, "div_y")):
        assert np.array_equal(a, g(k)), k
    placed = U.place_in_atlas(g("uv_rot"), g("assigned"), 0.02)
    assert placed.shape == g("placed").shape and np.abs(placed - g("placed")).max() < 2e-6
    assert placed.min() >= 0 and placed.max() <= 1


@pytest.mark.parametrize("name", ["ell", "tor"])
def test_axis_alignment_is_close_to_the_reference(z, name):
    """The reference's axes come from a RANDOMISED rank-2 PCA (torch.pca_lowrank, q=2, two power iterations, manual_seed(0)),
    which is only roughly the principal frame (15 degrees off on the 1 : 0.7 : 0.5 ellipsoid).  An exact PCA lands on the same
    canonical-axis assignment with every axis within ~25 degrees of the reference's, up to sign: unpinned by construction."""
    g = lambda k: z[name + "." + k]  # noqa: E731
    v = g("v_pos").astype(np.float64)
    c = v - v.mean(0)
    w, vec = np.linalg.eigh(c.T @ c)
    rot = U.axis_rotation(vec[:, 2], vec[:, 1])
    ref_rot = np.linalg.lstsq(g("v_pos").astype(np.float64), g("rot_pos").astype(np.float64), rcond=None)[0].T
    assert np.abs(ref_rot @ ref_rot.T - np.eye(3)).max() < 1e-4          # the reference's matrix is a rotation / reflection
    assert np.abs(rot @ rot.T - np.eye(3)).max() < 1e-5
    assert np.all(np.abs(np.diag(rot @ ref_rot.T)) > 0.9), np.diag(rot @ ref_rot.T)


def test_overlap_property_checker_finds_the_torus_back_layer(z):
    uv, chart = z["tor.uv_rot"], z["tor.face_index"]
    pairs = U.overlapping_pairs(uv, chart)
    assert len(pairs) > 50                      # inner and outer wall of the torus project onto each other
    assert U.overlapping_pairs(z["ell.uv_rot"], z["ell.face_index"]) == []
