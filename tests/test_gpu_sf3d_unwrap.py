"""Box-projection UV unwrapper on the MI355X: every stage against goldens from the reference's own Unwrapper methods, the
own atlas assignment by properties."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import sf3d_unwrap_ref as U
from sculptmate_amd.sf3d.unwrap import BoxProjectionUnwrapper, axis_rotation, principal_axes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def z():
    return np.load(os.path.join(GOLDEN, "sf3d_unwrap.npz"))


def _ref_rot(z, name):
    a, b = z[name + ".v_pos"].astype(np.float64), z[name + ".rot_pos"].astype(np.float64)
    return np.linalg.lstsq(a, b, rcond=None)[0].T.astype(np.float32)


@pytest.mark.parametrize("name", ["ell", "tor"])
@pytest.mark.parametrize("idx_dtype", [torch.int64, torch.int32])
def test_stages_vs_reference_goldens(cuda, z, name, idx_dtype):
    g = lambda k: z[name + "." + k]  # noqa: E731
    uw = BoxProjectionUnwrapper(256)
    # feed the reference's rotated mesh with an identity rotation: the stage inputs are then the reference's own
    rp_ref = torch.from_numpy(g("rot_pos")).to(cuda)
    rn_ref = torch.from_numpy(g("rot_nrm")).to(cuda)
    faces = torch.from_numpy(g("faces")).to(cuda).to(idx_dtype)
    rp, rn, uv, chart, st = uw.box_project(rp_ref, rn_ref, faces, np.eye(3, dtype=np.float32))
    assert torch.equal(rp, rp_ref) and torch.equal(rn, rn_ref)
    assert np.array_equal(chart.cpu().numpy(), g("face_index"))
    assert np.abs(uv.cpu().numpy() - g("uv_box")).max() < 1e-6
    angles, vt = uw.chart_angles(rp, rn, faces, uv, chart)
    assert np.abs(vt[:, :3].cpu().numpy() - g("tangents")).max() < 3e-5
    _, ref_angles = U.rotate_charts(g("rot_pos"), g("rot_nrm"), g("faces"), g("uv_box"), g("face_index"))
    assert np.abs(angles - ref_angles).max() < 2e-5
    uw.rotate_charts(uv, chart, angles, st)
    assert np.abs(uv.cpu().numpy() - g("uv_rot")).max() < 3e-5
    # placement with the fixture's hand-made assignment, from the reference's rotated charts
    uv_ref = torch.from_numpy(g("uv_rot")).to(cuda).contiguous()
    assigned = torch.from_numpy(g("assigned").astype(np.int32)).to(cuda)
    placed = uw.place(uv_ref, assigned, 0.02, st).cpu().numpy()
    assert placed.shape == g("placed").shape and np.abs(placed - g("placed")).max() < 1e-6


@pytest.mark.parametrize("name", ["ell", "tor"])
def test_whole_unwrap_vs_oracle_with_shared_rotation_and_assignment(cuda, z, name):
    """Unwrapper.forward end to end: same rotation, the library's own atlas assignment handed to the oracle."""
    g = lambda k: z[name + "." + k]  # noqa: E731
    uw = BoxProjectionUnwrapper(512)
    v, n, f = (torch.from_numpy(g(k)).to(cuda) for k in ("v_pos", "v_nrm", "faces"))
    rot = _ref_rot(z, name)
    uv, idx = uw(v, n, f, 0.02, rot=rot)
    assigned = uw.last["assigned"].cpu().numpy().astype(np.int64)
    ref, chart, _ = U.unwrap(g("v_pos"), g("v_nrm"), g("faces"), rot, lambda *a: assigned, 0.02)
    assert np.array_equal(uw.last["chart"].cpu().numpy(), chart)
    got = uv[idx].reshape(-1, 2).cpu().numpy()
    assert got.shape == ref.shape and np.abs(got - ref).max() < 5e-5
    assert got.min() >= 0 and got.max() <= 1
    assert torch.equal(idx.reshape(-1), torch.arange(3 * f.shape[0], device=cuda))


def test_own_atlas_assignment_properties(cuda, z):
    """Contract of assign_faces_uv_to_atlas_index: index in {c, c + 6, 12}; within the front layer and within the overlap
    slice of a chart no two triangles overlap; a convex body needs no second layer; the torus does, and its hidden wall goes
    to the overlap slice rather than to 'remaining'."""
    for name, res in (("ell", 512), ("tor", 512)):
        g = lambda k: z[name + "." + k]  # noqa: E731
        uw = BoxProjectionUnwrapper(res)
        rp = torch.from_numpy(g("rot_pos")).to(cuda)
        faces = torch.from_numpy(g("faces")).to(cuda)
        uv = torch.from_numpy(g("uv_rot")).to(cuda).contiguous()
        chart = torch.from_numpy(g("face_index").astype(np.int32)).to(cuda)
        a = uw.assign_atlas(rp, faces, uv, chart).cpu().numpy()
        c = g("face_index")
        assert np.all((a == c) | (a == c + 6) | (a == 12))
        if name == "ell":
            assert np.array_equal(a, c)
            continue
        moved = (a != c).mean()
        assert 0.2 < moved < 0.6, moved                      # roughly the hidden half of the tube
        assert (a == 12).mean() < 0.1
        keep = a < 12
        pairs = U.overlapping_pairs(g("uv_rot")[keep], a[keep])
        assert len(pairs) == 0, pairs[:5]
        # the front layer is the outer one: in the +x chart the kept triangles lie further out than the moved ones (the far
        # wall of the hole and the far side of the ring face +x too, behind the near wall)
        side = c == 0
        xc = g("rot_pos")[g("faces")].mean(1)[:, 0]
        assert (side & (a == 6)).sum() > 5
        assert xc[side & (a == 0)].mean() > xc[side & (a == 6)].mean()
        assert not np.any((c == 4) & (a != 4))                # the top of the tube is a single layer


def test_rotation_from_device_moments(cuda, z):
    for name in ("ell", "tor"):
        v = z[name + ".v_pos"]
        uw = BoxProjectionUnwrapper(64)
        rot = uw.rotation(torch.from_numpy(v).to(cuda))
        c = v.astype(np.float64) - v.astype(np.float64).mean(0)
        w, vec = np.linalg.eigh(c.T @ c)
        want = U.axis_rotation(vec[:, 2], vec[:, 1])
        assert np.abs(np.abs(rot) - np.abs(want)).max() < 1e-4       # same axes, sign convention aside
        assert np.abs(rot @ rot.T - np.eye(3)).max() < 1e-5
        ref = _ref_rot(z, name)
        assert np.all(np.abs(np.diag(rot @ ref.T)) > 0.9)             # and close to the reference's randomised PCA frame
    # host helpers on their own
    m, s = principal_axes([0, 0, 0, 4.0, 0, 0, 1.0, 0, 0.25], 1)
    assert np.allclose(np.abs(m), [1, 0, 0]) and np.allclose(np.abs(s), [0, 1, 0])
    assert np.allclose(axis_rotation([0, 0, 1], [1, 0, 0]), [[1, 0, 0], [0, 1, 0], [0, 0, 1]])


def test_unwrapper_in_sf3d_mesh_and_bake(cuda):
    """Mesh.unwrap_uv + texture bake with the box-projection unwrapper on a marching-tetrahedra sphere."""
    from sculptmate_amd import ops
    from sculptmate_amd.sf3d.system import Mesh

    n = 40
    lin = np.linspace(-1, 1, n, dtype=np.float32)
    x, y, zz = np.meshgrid(lin, lin, lin, indexing="ij")
    vol = torch.from_numpy((0.7 - np.sqrt(x * x + 1.4 * y * y + 2.0 * zz * zz)).astype(np.float32)).to(cuda)
    v, f = ops.marching_cubes(vol, 0.0)
    mesh = Mesh(v, f, unwrapper=BoxProjectionUnwrapper(512))
    mesh.unwrap_uv()
    nf = f.shape[0]
    assert mesh.v_pos.shape == (3 * nf, 3) and mesh.v_tex.shape == (3 * nf, 2)
    uv = mesh.v_tex.cpu().numpy()
    assert uv.min() >= 0 and uv.max() <= 1
    a = mesh.unwrapper.last["assigned"].cpu().numpy()
    assert (a < 6).mean() > 0.97                                          # a convex blob: (almost) everything in the front layer
    rast = ops.bake_rasterize(mesh.v_tex, mesh.t_pos_idx, 256)
    cover = float((rast[..., 3] >= 0).float().mean())
    assert cover > 0.25, cover                                            # six large charts fill a good part of the 3x2 atlas
    with pytest.raises(Exception):
        BoxProjectionUnwrapper()(v.cpu(), v.cpu(), f.cpu(), 0.02)


def test_dll_entry_point_called_like_the_reference_does(cuda, z):
    """uv_unwrapper.dll's export under its own name and host-pointer signature (unwrap.py:143-172): the ctypes call the
    reference makes, pointed at libsculpt_hip.so, returns the same assignment as the device entry point."""
    import ctypes

    from sculptmate_amd import _lib

    dll = ctypes.CDLL(_lib.SO_PATH)
    fn = dll.assign_faces_uv_to_atlas_index
    fn.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_size_t, ctypes.POINTER(ctypes.c_longlong), ctypes.c_size_t,
                   ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong)]
    fn.restype = None
    g = lambda k: z["tor." + k]  # noqa: E731
    v = np.ascontiguousarray(g("rot_pos").reshape(-1), np.float32)
    f = np.ascontiguousarray(g("faces").reshape(-1), np.int64)
    uv = np.ascontiguousarray(g("uv_rot").reshape(-1), np.float32)
    fi = np.ascontiguousarray(g("face_index"), np.int64)
    out = np.zeros(fi.shape[0], np.int64)
    fn(v.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), v.size // 3, f.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), fi.shape[0],
       uv.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), fi.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)),
       out.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)))
    uw = BoxProjectionUnwrapper(1024)
    want = uw.assign_atlas(torch.from_numpy(g("rot_pos")).to(cuda), torch.from_numpy(g("faces")).to(cuda),
                           torch.from_numpy(g("uv_rot")).to(cuda).contiguous(), torch.from_numpy(fi.astype(np.int32)).to(cuda))
    assert np.array_equal(out, want.cpu().numpy().astype(np.int64))
    assert np.all((out == fi) | (out == fi + 6) | (out == 12)) and (out != fi).any()
