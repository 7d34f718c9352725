"""ctypes front-end to the C oracle (oracle/_build/liboracle.so).

ORACLE -- TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never by sculptmate_amd/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("mc_lewiner.c", "triplane_query.c", "baker.c", "mc_luts.h", "Makefile")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.oracle_marching_cubes.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int,
            ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int),
            ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int)]
        L.oracle_marching_cubes.restype = ctypes.c_int
        L.oracle_mc_free.argtypes = [ctypes.c_void_p]
        L.oracle_mc_classify.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int)] * 3
        L.oracle_mc_classify.restype = ctypes.c_int
        fpp = ctypes.POINTER(ctypes.c_void_p)
        L.oracle_query_triplane.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
            ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, fpp, fpp,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.oracle_query_triplane.restype = None
        L.oracle_density_grid.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
            ctypes.c_float, ctypes.c_int, ctypes.c_void_p, fpp, fpp, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_void_p]
        L.oracle_density_grid.restype = None
        L.oracle_grid_point.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
        L.oracle_grid_point.restype = None
        L.oracle_bake_rasterize.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
        L.oracle_bake_rasterize.restype = None
        L.oracle_bake_interpolate.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.oracle_bake_interpolate.restype = None
        L.oracle_set_threads.argtypes = [ctypes.c_int]
        L.oracle_max_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def set_threads(n):
    lib().oracle_set_threads(int(n))
    return lib().oracle_max_threads()


class MCError(ValueError):
    pass


def marching_cubes(vol, level=0.0, use_classic=False):
    """skimage.measure.marching_cubes(vol, level) -> (verts f32[nv,3], faces i32[nf,3])."""
    vol = np.ascontiguousarray(vol, np.float32)
    assert vol.ndim == 3
    pv, pf = ctypes.c_void_p(), ctypes.c_void_p()
    nv, nf = ctypes.c_int(), ctypes.c_int()
    rc = lib().oracle_marching_cubes(vol.ctypes.data, vol.shape[0], vol.shape[1], vol.shape[2],
                                     float(level), int(use_classic), ctypes.byref(pv), ctypes.byref(nv),
                                     ctypes.byref(pf), ctypes.byref(nf))
    if rc == 1:
        raise MCError("Surface level must be within volume data range.")
    if rc == 2:
        raise RuntimeError("No surface found at the given iso value.")
    if rc == 3:
        raise MCError("Input array must be at least 2x2x2.")
    verts = np.ctypeslib.as_array(ctypes.cast(pv, ctypes.POINTER(ctypes.c_float)), (nv.value, 3)).copy()
    faces = np.ctypeslib.as_array(ctypes.cast(pf, ctypes.POINTER(ctypes.c_int32)), (nf.value, 3)).copy()
    lib().oracle_mc_free(pv)
    lib().oracle_mc_free(pf)
    return verts, faces


def reference_isosurface(level_arg, resolution):
    """MarchingCubeHelper.forward (isosurface.py:41-54): level_arg is what the reference passes
    (= -(density - threshold)); returns (v_pos f32[nv,3] in [0,1], t_pos_idx int64[nf,3])."""
    vol = -np.asarray(level_arg, np.float32).reshape(resolution, resolution, resolution)
    v, f = marching_cubes(vol, 0.0)
    f = f[:, [1, 0, 2]].astype(np.int64)
    v = (v / np.float32(resolution - 1.0)).astype(np.float32)
    return v, f


def _mlp_args(weights, biases):
    Ws = [np.ascontiguousarray(w, np.float32) for w in weights]
    bs = [np.ascontiguousarray(b, np.float32) for b in biases]
    dims = np.array([Ws[0].shape[1]] + [w.shape[0] for w in Ws], np.int32)
    n = len(Ws)
    WP = (ctypes.c_void_p * n)(*[w.ctypes.data for w in Ws])
    BP = (ctypes.c_void_p * n)(*[b.ctypes.data for b in bs])
    return Ws, bs, dims, n, WP, BP


def query_triplane(planes, points, weights, biases, radius=0.87, density_bias=-1.0):
    """query_triplane (nerf_renderer.py:41-91) -> dict(density, features, density_act, color)."""
    planes = np.ascontiguousarray(planes, np.float32)
    pts = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
    N = pts.shape[0]
    Ws, bs, dims, n, WP, BP = _mlp_args(weights, biases)
    d = np.empty((N, 1), np.float32); f = np.empty((N, 3), np.float32)
    da = np.empty((N, 1), np.float32); c = np.empty((N, 3), np.float32)
    lib().oracle_query_triplane(planes.ctypes.data, planes.shape[1], planes.shape[2], planes.shape[3],
                                pts.ctypes.data, N, radius, density_bias, n, dims.ctypes.data, WP, BP,
                                d.ctypes.data, f.ctypes.data, da.ctypes.data, c.ctypes.data)
    return {"density": d, "features": f, "density_act": da, "color": c}


def density_grid(planes, weights, biases, R, radius=0.87, density_bias=-1.0, begin=0, end=None):
    """density_act over the flat lattice range [begin, end) of an R^3 grid (system.py:171-183)."""
    planes = np.ascontiguousarray(planes, np.float32)
    end = R ** 3 if end is None else end
    Ws, bs, dims, n, WP, BP = _mlp_args(weights, biases)
    out = np.empty(end - begin, np.float32)
    lib().oracle_density_grid(planes.ctypes.data, planes.shape[1], planes.shape[2], planes.shape[3], R,
                              radius, density_bias, n, dims.ctypes.data, WP, BP, begin, end,
                              out.ctypes.data)
    return out


def grid_points(R, radius=0.87, idx=None):
    idx = np.arange(R ** 3, dtype=np.int64) if idx is None else np.asarray(idx, np.int64)
    out = np.empty((idx.size, 3), np.float32)
    tmp = np.empty(3, np.float32)
    L = lib()
    for j, i in enumerate(idx):
        L.oracle_grid_point(int(i), R, radius, tmp.ctypes.data)
        out[j] = tmp
    return out


def bake_rasterize(uv, faces, res):
    """rasterize_cpu (texture_baker/common.py:123-142) -> f32 [res, res, 4] (u, v, w, triangle id | -1)."""
    uv = np.ascontiguousarray(uv, np.float32); faces = np.ascontiguousarray(faces, np.int32)
    out = np.empty((res, res, 4), np.float32)
    lib().oracle_bake_rasterize(uv.ctypes.data, uv.shape[0], faces.ctypes.data, faces.shape[0], res, out.ctypes.data)
    return out


def bake_interpolate(attr, faces, rast):
    """interpolate_cpu (texture_baker/common.py:214-229) -> f32 [res, res, 3]."""
    attr = np.ascontiguousarray(attr, np.float32); faces = np.ascontiguousarray(faces, np.int32)
    rast = np.ascontiguousarray(rast, np.float32)
    out = np.empty(rast.shape[:2] + (3,), np.float32)
    lib().oracle_bake_interpolate(attr.ctypes.data, faces.ctypes.data, rast.ctypes.data, rast.shape[0], out.ctypes.data)
    return out
