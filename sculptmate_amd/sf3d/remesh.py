"""Triangle remeshing for SF3D.run_image(remesh="triangle") -- Mesh.triangle_remesh of the reference
(StableFast/sf3d/models/mesh.py:175-237).

The reference does this step on the CPU through the third-party package gpytoolbox: optional `subdivide`, `decimate` to the
requested vertex count, then `remesh_botsch`.  Here the three operations are native host code behind the C ABI
(`sculpt_mesh_subdivide / _decimate / _remesh_botsch`, csrc/remesh_host.h) with gpytoolbox's call signatures:

    subdivide(v, f, iters=1)                 midpoint subdivision
    decimate(v, f, face_ratio=0.1)           shortest-edge collapse (libigl's default) -> (v, f, None, None)
    remesh_botsch(v, f, i=10, h=None)        Botsch-Kobbelt isotropic remeshing, projected onto the input surface

gpytoolbox is absent from the reference tree and from the build image, so the outputs cannot be compared with it:
PARITY UNPINNED.  tests/test_remesh.py checks the invariants the reference relies on (closed manifold stays a closed
manifold of the same genus and orientation, the face / vertex budget is met, edge lengths end in the [4/5 h, 4/3 h] band,
vertices stay on the input surface).  `gpytoolbox_remesher` still drives the real package when a caller injects it.
"""
import ctypes
import math

import numpy as np
import torch

from .. import _lib


def _take(handle):
    """Copy a sculpt_host_mesh_t result out and release it."""
    try:
        nv, nf = _lib.lib.sculpt_mesh_num_vertices(handle), _lib.lib.sculpt_mesh_num_faces(handle)
        v = np.empty((nv, 3), np.float64)
        f = np.empty((nf, 3), np.int32)
        _lib.check(_lib.lib.sculpt_mesh_read(handle, v.ctypes.data, f.ctypes.data))
        return v, f
    finally:
        _lib.lib.sculpt_mesh_free(handle)


def _inputs(v, f):
    v = np.ascontiguousarray(np.asarray(v, np.float64).reshape(-1, 3))
    f = np.ascontiguousarray(np.asarray(f).reshape(-1, 3).astype(np.int32, copy=False))
    return v, f


def subdivide(v, f, iters=1):
    """gpytoolbox.subdivide(v, f, method='upsample', iters=...) -> (v, f)."""
    v, f = _inputs(v, f)
    out = ctypes.c_void_p()
    _lib.check(_lib.lib.sculpt_mesh_subdivide(v.ctypes.data, len(v), f.ctypes.data, len(f), int(iters), ctypes.byref(out)))
    return _take(out)


def decimate(v, f, face_ratio=0.1, num_faces=None):
    """gpytoolbox.decimate(v, f, face_ratio=..., num_faces=None) -> (v, f, None, None): the birth-index outputs of the
    original are not produced (the reference discards them, mesh.py:195)."""
    v, f = _inputs(v, f)
    if num_faces is None:
        num_faces = int(math.floor(face_ratio * len(f)))
    out = ctypes.c_void_p()
    _lib.check(_lib.lib.sculpt_mesh_decimate(v.ctypes.data, len(v), f.ctypes.data, len(f), max(0, int(num_faces)), ctypes.byref(out)))
    vo, fo = _take(out)
    return vo, fo, None, None


def remesh_botsch(v, f, i=10, h=None, project=True):
    """gpytoolbox.remesh_botsch(v, f, i, h) -> (v, f); h None = mean edge length of the input."""
    v, f = _inputs(v, f)
    out = ctypes.c_void_p()
    _lib.check(_lib.lib.sculpt_mesh_remesh_botsch(v.ctypes.data, len(v), f.ctypes.data, len(f), int(i),
                                                  -1.0 if h is None else float(h), 1 if project else 0, ctypes.byref(out)))
    return _take(out)


class _Native:
    """The three calls under the names Mesh.triangle_remesh uses."""
    subdivide = staticmethod(subdivide)
    decimate = staticmethod(decimate)
    remesh_botsch = staticmethod(remesh_botsch)


def triangle_remesh(mesh, vertex_count=-1, remesh_steps: int = 10, edge_length_multiplier=None, toolbox=_Native):
    """Mesh.triangle_remesh(triangle_average_edge_length_multiplier, triangle_remesh_steps, triangle_vertex_count)
    (mesh.py:175-237), statement for statement: with a vertex budget, subdivide while the mesh has fewer vertices than
    asked for, decimate with face_ratio = budget / vertices, and remesh at the decimated mesh's own mean edge length;
    without one, remesh at mean edge length x multiplier (or the mean itself)."""
    from .system import Mesh

    v = mesh.v_pos.detach().cpu().numpy().astype(np.float32)
    f = mesh.t_pos_idx.detach().cpu().numpy().astype(np.int32)
    if vertex_count > 0:
        ratio = vertex_count / v.shape[0]
        if ratio > 1.0:
            v, f = toolbox.subdivide(v, f, iters=int(math.ceil(math.log(ratio) / math.log(2))))
            ratio = vertex_count / v.shape[0]
        v, f, _, _ = toolbox.decimate(v, f, face_ratio=ratio)
        # the reference round-trips the decimated mesh through the fp32 / integer tensors of the Mesh
        v = np.asarray(v).astype(np.float32)
        edge_length_multiplier = None
    h = None
    if edge_length_multiplier is not None:
        vd = np.asarray(v, np.float64)
        fe = np.asarray(f).reshape(-1, 3)
        e = np.unique(np.sort(np.concatenate([fe[:, [0, 1]], fe[:, [1, 2]], fe[:, [2, 0]]], 0), 1), axis=0)  # Mesh.edges
        h = float(np.linalg.norm(vd[e[:, 0]] - vd[e[:, 1]], axis=1).astype(np.float32).mean() * edge_length_multiplier)
    v, f = toolbox.remesh_botsch(np.asarray(v, np.float64), np.asarray(f, np.int32), remesh_steps, h)
    dev = mesh.v_pos.device
    return Mesh(torch.from_numpy(np.ascontiguousarray(v)).to(dev, mesh.v_pos.dtype).contiguous(),
                torch.from_numpy(np.ascontiguousarray(f)).to(dev, mesh.t_pos_idx.dtype).contiguous(), unwrapper=mesh.unwrapper)


def native_remesher(mesh, mode, vertex_count, remesh_steps: int = 10):
    """(Mesh, "triangle", target vertex count) -> Mesh: what SF3D.generate_mesh calls (system.py:353-356)."""
    if mode != "triangle":
        raise NotImplementedError("remesh=%r: only 'triangle' exists (the reference's quad path is commented out, "
                                  "mesh.py:152-171)" % mode)
    return triangle_remesh(mesh, vertex_count, remesh_steps)


def gpytoolbox_remesher(mesh, mode, vertex_count, remesh_steps: int = 10, gpytoolbox=None):
    """The same sequence through the real gpytoolbox package (or a stand-in the caller injects)."""
    if mode != "triangle":
        raise NotImplementedError("remesh=%r: only 'triangle' exists" % mode)
    if gpytoolbox is None:
        import gpytoolbox  # noqa: F811  (ImportError tells the caller what is missing)
    return triangle_remesh(mesh, vertex_count, remesh_steps, toolbox=gpytoolbox)


def default_remesher():
    return native_remesher
