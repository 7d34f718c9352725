"""Thin torch-tensor front-ends over the C ABI (include/sculpt_hip.h).

torch is plumbing here: it owns HBM allocations and the current HIP stream; every function below
passes raw device pointers + sizes into libsculpt_hip.so.  No function has a CPU path.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import SculptError, check, lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _req(t, dtype, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise SculptError("%s must be a CUDA/HIP tensor (no CPU fallback)" % name)
    if t.dtype != dtype:
        raise SculptError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise SculptError("%s must be contiguous" % name)
    return t


# ----------------------------------------------------------------------------------------------
# NeRF decoder
# ----------------------------------------------------------------------------------------------
class PackedMLP:
    """Decoder weights re-ordered for the MFMA kernels (sculpt_mlp_pack), resident in HBM."""

    def __init__(self, weights, biases, device):
        Ws = [np.ascontiguousarray(w.detach().cpu().numpy() if isinstance(w, torch.Tensor) else w, np.float32)
              for w in weights]
        bs = [np.ascontiguousarray(b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else b, np.float32)
              for b in biases]
        n = len(Ws)
        dims = np.array([Ws[0].shape[1]] + [w.shape[0] for w in Ws], np.int32)
        self.in_channels = int(dims[0])
        self.n_hidden = n - 2
        nbytes = lib.sculpt_mlp_packed_bytes(self.in_channels, self.n_hidden)
        host = np.zeros(nbytes // 4, np.float32)
        WP = (ctypes.c_void_p * n)(*[w.ctypes.data for w in Ws])
        BP = (ctypes.c_void_p * n)(*[b.ctypes.data for b in bs])
        check(lib.sculpt_mlp_pack(WP, BP, n, dims.ctypes.data, host.ctypes.data, nbytes))
        self.blob = torch.from_numpy(host).to(device)


def triplane_query(planes, mlp, points, radius=0.87, density_bias=-1.0,
                   want=("density", "features", "density_act", "color")):
    """query_triplane (nerf_renderer.py:41-91) at arbitrary points -> dict of [N,1]/[N,3] tensors."""
    planes = _req(planes, torch.float32, "planes")
    shape = points.shape[:-1]
    pts = _req(points.reshape(-1, 3).contiguous(), torch.float32, "points")
    N = pts.shape[0]
    _, C, H, W = planes.shape
    out = {}
    for k, w in (("density", 1), ("features", 3), ("density_act", 1), ("color", 3)):
        out[k] = torch.empty((N, w), dtype=torch.float32, device=planes.device) if k in want else None
    check(lib.sculpt_triplane_query(_ptr(planes), C, H, W, _ptr(mlp.blob), mlp.n_hidden, _ptr(pts), N,
                                    float(radius), float(density_bias), _ptr(out["density"]),
                                    _ptr(out["features"]), _ptr(out["density_act"]), _ptr(out["color"]),
                                    _stream()))
    return {k: v.view(*shape, v.shape[-1]) for k, v in out.items() if v is not None}


def grid_axis_coords(resolution, radius):
    """Per-axis lattice coordinate table, computed on the host exactly as the reference does:
    torch.linspace(0, 1, R) (isosurface.py:28-32) then scale_tensor(., (0,1), (-r, r))
    (system.py:177-181, utils.py:222-231).  R floats -- the lattice is separable."""
    g = torch.linspace(0, 1, resolution)
    g = (g - 0) / (1 - 0)
    return g * (radius - (-radius)) + (-radius)


_ws_cache = {}


def _workspace(key, nbytes, device):
    t = _ws_cache.get(key)
    if t is None or t.numel() < nbytes or t.device != device:
        t = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _ws_cache[key] = t
    return t


def density_grid(planes, mlp, resolution, radius=0.87, density_bias=-1.0, x_begin=0, x_end=None, out=None):
    """density_act over the lattice slab ix in [x_begin, x_end): f32 [(x_end-x_begin)*R*R]
    (TSR.extract_mesh's dense query, system.py:171-183)."""
    planes = _req(planes, torch.float32, "planes")
    R = int(resolution)
    x_end = R if x_end is None else int(x_end)
    nx = x_end - x_begin
    _, C, H, W = planes.shape
    axis = grid_axis_coords(R, radius).to(planes.device)
    ws = _workspace(("dg", planes.device), lib.sculpt_density_grid_workspace_bytes(R, nx), planes.device)
    if out is None:
        out = torch.empty(nx * R * R, dtype=torch.float32, device=planes.device)
    check(lib.sculpt_density_grid(_ptr(planes), C, H, W, _ptr(mlp.blob), mlp.n_hidden, _ptr(axis), R,
                                  int(x_begin), x_end, float(radius), float(density_bias), _ptr(ws),
                                  _ptr(out), _stream()))
    return out


# ----------------------------------------------------------------------------------------------
# marching cubes
# ----------------------------------------------------------------------------------------------
def marching_cubes(vol, level=0.0, reference_order=False, vert_div=1.0, vert_mul=1.0, vert_add=0.0,
                   use_classic=False):
    """skimage.measure.marching_cubes(vol, level) on the GPU.

    reference_order=False: (verts f32[nv,3] voxel units, faces i32[nf,3]) exactly as skimage returns.
    reference_order=True : faces int64 with columns [1,0,2] and verts/(vert_div)*vert_mul+vert_add
                           (MarchingCubeHelper.forward + scale_tensor, isosurface.py:49-53, system.py:185-189).
    Raises ValueError / RuntimeError like skimage for an out-of-range level / empty surface.
    """
    vol = _req(vol, torch.float32, "vol")
    assert vol.dim() == 3
    n0, n1, n2 = vol.shape
    ws = _workspace(("mc", vol.device), lib.sculpt_mc_workspace_bytes(n0, n1, n2), vol.device)
    flags = 0
    if reference_order:
        flags |= _lib.MC_REFERENCE_ORDER | _lib.MC_FACES_I64
    if use_classic:
        flags |= _lib.MC_USE_CLASSIC
    nv, nf = ctypes.c_int64(), ctypes.c_int64()
    rc = lib.sculpt_mc_count(_ptr(vol), n0, n1, n2, float(level), flags, _ptr(ws), ctypes.byref(nv),
                             ctypes.byref(nf), _stream())
    if rc == _lib.ERR_MC_LEVEL:
        raise ValueError(_lib.last_error())
    if rc == _lib.ERR_MC_EMPTY:
        raise RuntimeError(_lib.last_error())
    check(rc)
    verts = torch.empty((nv.value, 3), dtype=torch.float32, device=vol.device)
    faces = torch.empty((nf.value, 3), dtype=torch.int64 if reference_order else torch.int32, device=vol.device)
    check(lib.sculpt_mc_emit(_ptr(vol), n0, n1, n2, float(level), flags, _ptr(ws), float(vert_div),
                             float(vert_mul), float(vert_add), _ptr(verts), _ptr(faces), _stream()))
    return verts, faces
