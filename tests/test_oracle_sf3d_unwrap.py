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
def test_axis_alignment_matches_reference_up_to_sign(z, name):
    """The reference's axes come from a randomised rank-2 PCA (torch.pca_lowrank, q=2, two power iterations): an exact PCA
    gives the same canonical-axis assignment and axes within ~2 degrees of it, up to sign."""
    g = lambda k: z[name + "." + k]  # noqa: E731
    v = g("v_pos").astype(np.float64)
    c = v - v.mean(0)
    w, vec = np.linalg.eigh(c.T @ c)
    main, second = vec[:, 2], vec[:, 1]
    rot = U.axis_rotation(main, second)
    rp = np.einsum("ij,nj->ni", rot, g("v_pos"))
    ref = g("rot_pos")
    for k in range(3):     # every rotated coordinate equals the reference's up to a global sign
        err = min(np.abs(rp[:, k] - ref[:, k]).max(), np.abs(rp[:, k] + ref[:, k]).max())
        assert err < 5e-2, (k, err)
    assert abs(abs(np.linalg.det(rot)) - 1) < 1e-5


def test_overlap_property_checker_finds_the_torus_back_layer(z):
    uv, chart = z["tor.uv_rot"], z["tor.face_index"]
    pairs = U.overlapping_pairs(uv, chart)
    assert len(pairs) > 50                      # inner and outer wall of the torus project onto each other
    assert U.overlapping_pairs(z["ell.uv_rot"], z["ell.face_index"]) == []
