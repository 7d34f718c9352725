"""Thin torch-tensor front-ends over the C ABI (include/sculpt_hip.h).

torch is plumbing here: it owns HBM allocations and the current HIP stream; every function below
passes raw device pointers + sizes into libsculpt_hip.so.  No function has a CPU path.
"""
import ctypes
import threading
import os

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


class ChannelLastPlanes:
    """A scene code re-laid as [3][H][W][C] for the point-query kernel (sculpt_planes_channel_last)."""

    def __init__(self, planes):
        planes = _req(planes.contiguous(), torch.float32, "planes")
        _, self.C, self.H, self.W = planes.shape
        self.data = torch.empty((3, self.H, self.W, self.C), dtype=torch.float32, device=planes.device)
        check(lib.sculpt_planes_channel_last(_ptr(planes), self.C, self.H, self.W, _ptr(self.data), _stream()))


def triplane_query(planes, mlp, points, radius=0.87, density_bias=-1.0,
                   want=("density", "features", "density_act", "color"), align_corners=False):
    """query_triplane (nerf_renderer.py:41-91) at arbitrary points -> dict of [N,1]/[N,3] tensors.
    align_corners=True is SF3D.query_triplane's sampling (StableFast/sf3d/system.py:170-199).
    planes: [3,C,H,W] (reference layout) or a ChannelLastPlanes (much faster for many points)."""
    shape = points.shape[:-1]
    pts = _req(points.reshape(-1, 3).contiguous(), torch.float32, "points")
    N = pts.shape[0]
    flags = _lib.QUERY_ALIGN_CORNERS if align_corners else 0
    if isinstance(planes, ChannelLastPlanes):
        C, H, W = planes.C, planes.H, planes.W
        planes = planes.data
        flags |= _lib.QUERY_CHANNEL_LAST
    else:
        planes = _req(planes, torch.float32, "planes")
        _, C, H, W = planes.shape
    out = {}
    for k, w in (("density", 1), ("features", 3), ("density_act", 1), ("color", 3)):
        out[k] = torch.empty((N, w), dtype=torch.float32, device=pts.device) if k in want else None
    check(lib.sculpt_triplane_query_ex(_ptr(planes), C, H, W, _ptr(mlp.blob), mlp.n_hidden, _ptr(pts), N,
                                       float(radius), float(density_bias), flags, _ptr(out["density"]),
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


_DENSITY_FLAGS = {"fp32": 0, "bf16l3": _lib.DENSITY_BF16L3}


def lattice_decode(planes, mlp, axis, radius, density_bias=0.0, out_add=0.0, want=("density_act",), align_corners=True,
                   out=None):
    """One decoder head on the full R^3 lattice point(ix, iy, iz) = (axis[ix], axis[iy], axis[iz]), flat order
    (ix*R + iy)*R + iz, through the separable layer-0 tables (sculpt_plane_features_ex + sculpt_grid_decode): what
    triplane_query + the head MLP give at those points, with the first layer's sum regrouped per plane.
    axis f32 [R] device (world coordinates exactly as the caller's positions have them); planes [3,C,H,W] channel-first.
    want: "density_act" -> f32 [R^3] = exp(row 0 + density_bias) + out_add; "features" -> f32 [R^3, 3] = rows 1..3, raw."""
    planes = _req(planes, torch.float32, "planes")
    axis = _req(axis, torch.float32, "axis")
    R = int(axis.numel())
    _, C, H, W = planes.shape
    ws = _workspace(("dg", planes.device), lib.sculpt_density_grid_workspace_bytes(R, R), planes.device)
    res = {}
    if "density_act" in want:
        res["density_act"] = out if out is not None else torch.empty(R * R * R, dtype=torch.float32, device=planes.device)
    if "features" in want:
        res["features"] = torch.empty((R * R * R, 3), dtype=torch.float32, device=planes.device)
    if not res:
        raise SculptError("lattice_decode: want must name density_act and / or features")
    flags = _lib.QUERY_ALIGN_CORNERS if align_corners else 0
    check(lib.sculpt_plane_features_ex(_ptr(planes), C, H, W, _ptr(mlp.blob), _ptr(axis), R, 0, R, float(radius), flags,
                                       _ptr(ws), _stream()))
    check(lib.sculpt_grid_decode(_ptr(mlp.blob), mlp.n_hidden, R, 0, R, float(density_bias), float(out_add), _ptr(ws),
                                 _ptr(res["density_act"]) if "density_act" in res else None,
                                 _ptr(res["features"]) if "features" in res else None, _stream()))
    return res


def density_grid(planes, mlp, resolution, radius=0.87, density_bias=-1.0, x_begin=0, x_end=None, out=None,
                 out_add=0.0, events=None, precision="fp32"):
    """density_act (+ out_add) over the lattice slab ix in [x_begin, x_end): f32 [(x_end-x_begin)*R*R]
    (TSR.extract_mesh's dense query, system.py:171-183; out_add=-threshold folds system.py:184).
    events: optional (start, stop) torch.cuda.Event pair recorded around the fused MLP launch only.
    precision: "fp32" (exact fp32 MFMA; this function's default) or "bf16l3" (three exact bf16 limbs per operand, six products,
    fp32 accumulate: fp32-equivalent, what TSR.extract_meshes uses)."""
    if precision not in _DENSITY_FLAGS:
        raise SculptError("density_grid: precision must be one of %s" % sorted(_DENSITY_FLAGS))
    planes = _req(planes, torch.float32, "planes")
    R = int(resolution)
    x_end = R if x_end is None else int(x_end)
    nx = x_end - x_begin
    _, C, H, W = planes.shape
    axis = _axis_table(R, radius, planes.device)
    ws = _workspace(("dg", planes.device), lib.sculpt_density_grid_workspace_bytes(R, nx), planes.device)
    if out is None:
        out = torch.empty(nx * R * R, dtype=torch.float32, device=planes.device)
    check(lib.sculpt_plane_features(_ptr(planes), C, H, W, _ptr(mlp.blob), _ptr(axis), R, int(x_begin), x_end,
                                    float(radius), _ptr(ws), _stream()))
    if events is not None:
        events[0].record()
    check(lib.sculpt_density_grid_ex(_ptr(mlp.blob), mlp.n_hidden, R, int(x_begin), x_end, float(density_bias),
                                     float(out_add), _ptr(ws), _ptr(out),
                                     _DENSITY_FLAGS[precision], _stream()))
    if events is not None:
        events[1].record()
    return out


def density_grid_filtered(planes, mlp, resolution, margin, radius=0.87, density_bias=-1.0, x_begin=0, x_end=None, out=None,
                          out_add=0.0, events=None, coarse="fp16", mark_all=False, stats_host=None, passes="ABC", tables=True):
    """density_grid(precision="bf16l3") for marching cubes, filtered (sculpt_density_grid_filtered): every lattice point with one
    16-bit product per hidden layer (`coarse`: "fp16" or "bf16"; pass A), then the exact three-limb arithmetic at the points
    within `margin` (natural-log units of density_act) of the level -out_add (pass B: every sign certain) and at the values
    marching cubes reads -- end points of sign-changing lattice edges, all corners of ambiguous cells (pass C).  Returns (volume,
    stats): the volume holds the bits of the full evaluation wherever marching cubes reads a value and a value of the right sign
    elsewhere as long as no coarse error reaches the margin; stats = device int32[FILTER_STATS_WORDS] view of the call's
    statistics (include/sculpt_hip.h; filter_stats() decodes them, filter_guard_error() is the figure to hold against the
    margin), valid once the stream has passed the call -- stats_host (a pinned int32[FILTER_STATS_WORDS] tensor) receives an
    asynchronous copy.  mark_all=True re-evaluates every point and stats[1] is the largest coarse error (calibration).
    passes / tables: timing aids -- run only the named passes ("A", "B", "C" one after the other on the same workspace give what
    "ABC" gives), skip the plane tables when an earlier call with the same arguments has left them in the workspace."""
    if coarse not in ("fp16", "bf16"):
        raise SculptError("density_grid_filtered: coarse must be 'fp16' or 'bf16'")
    planes = _req(planes, torch.float32, "planes")
    R = int(resolution)
    x_end = R if x_end is None else int(x_end)
    nx = x_end - x_begin
    _, C, H, W = planes.shape
    axis = _axis_table(R, radius, planes.device)
    ws = _workspace(("dg", planes.device), lib.sculpt_density_grid_workspace_bytes(R, nx), planes.device)
    fws = _workspace(("dgf", planes.device), lib.sculpt_density_filter_workspace_bytes(R, nx), planes.device)
    if out is None:
        out = torch.empty(nx * R * R, dtype=torch.float32, device=planes.device)
    if tables:
        check(lib.sculpt_plane_features(_ptr(planes), C, H, W, _ptr(mlp.blob), _ptr(axis), R, int(x_begin), x_end,
                                        float(radius), _ptr(ws), _stream()))
    flags = _lib.DENSITY_BF16L3 | (_lib.FILTER_COARSE_FP16 if coarse == "fp16" else 0) | (_lib.FILTER_MARK_ALL if mark_all else 0)
    if passes != "ABC":
        flags |= sum({"A": _lib.FILTER_PASS_A, "B": _lib.FILTER_PASS_B, "C": _lib.FILTER_PASS_C}[c] for c in passes)
    if events is not None:
        events[0].record()
    check(lib.sculpt_density_grid_filtered(_ptr(mlp.blob), mlp.n_hidden, R, int(x_begin), x_end, float(density_bias),
                                           float(out_add), float(margin), _ptr(ws), _ptr(fws), _ptr(out), flags, _stream()))
    if events is not None:
        events[1].record()
    # whose sign planes the workspace now holds (filter_sign_planes / marching_cubes check it: a probe or a slab call in between
    # rewrites them)
    gen = _filter_last.get(planes.device, (0,))[0] + 1
    _filter_last[planes.device] = (gen, R, int(x_begin), x_end, "C" in passes and not mark_all)
    stats = fws[:4 * FILTER_STATS_WORDS].view(torch.int32)
    if stats_host is not None:
        stats_host.copy_(stats, non_blocking=True)
    return out, stats


_filter_last = {}   # device -> (generation, R, x_begin, x_end, complete) of the last density_grid_filtered call


def filter_sign_planes(resolution, device, x_begin=0, x_end=None):
    """The sign planes the LAST density_grid_filtered call on this device left in the filter workspace -- which must be a call
    of these arguments: int32 [nx * R][ceil(R / 32)], bit iz % 32 of word iz / 32 = (final volume value > 0), what
    marching_cubes(sign_planes=) takes for level 0.  A view, stamped with the call's generation: marching_cubes refuses it once
    another filtered call (a calibration probe, a slab) has rewritten the workspace."""
    R = int(resolution)
    x_end = R if x_end is None else int(x_end)
    nx = x_end - int(x_begin)
    fws = _ws_cache.get(("dgf", device))
    last = _filter_last.get(device)
    if fws is None or last is None or last[1:] != (R, int(x_begin), x_end, True):
        raise SculptError("filter_sign_planes: the last filtered density grid on %s was not a complete call of R=%d, x in [%d, %d)"
                          % (device, R, int(x_begin), x_end))
    off, nzb = int(lib.sculpt_density_filter_sign_offset(R, nx)), (R + 31) // 32
    view = fws[off:off + nx * R * nzb * 4].view(torch.int32).view(nx * R, nzb)
    view._sculpt_filter_generation = last[0]
    return view


def _check_sign_planes_current(sign_planes):
    gen = getattr(sign_planes, "_sculpt_filter_generation", None)
    if gen is not None and _filter_last.get(sign_planes.device, (None,))[0] != gen:
        raise SculptError("marching_cubes: these sign planes are a view of the filter workspace, and a later "
                          "density_grid_filtered call on %s has rewritten it" % sign_planes.device)


FILTER_STATS_WORDS = 12   # SCULPT_FILTER_STATS_WORDS


def filter_stats(stats):
    """int32[FILTER_STATS_WORDS] statistics of density_grid_filtered (host or device tensor; a device tensor is read back
    here) -> dict.  max_err: largest coarse error over every re-evaluated point (inf: an unmarked point's sign was wrong, or a
    NaN); audit_err: the same over the audit sample of otherwise untouched points; n_mismatch: unmarked points with a wrong
    coarse sign; n_sign_fixed: marked points whose sign pass B corrected (what the margin is for)."""
    s = stats.cpu().numpy() if isinstance(stats, torch.Tensor) else np.asarray(stats)
    return {"n_refined": int(s[0]), "max_err": float(s[1:2].view(np.float32)[0]), "n_marked": int(s[2]),
            "n_nonfinite": int(s[3]), "n_cells": int(s[4]), "n_points": int(s[5]), "n_first": int(s[6]), "n_second": int(s[7]),
            "audit_err": float(s[8:9].view(np.float32)[0]), "n_audit": int(s[9]), "n_mismatch": int(s[10]),
            "n_sign_fixed": int(s[11])}


def filter_guard_error(st):
    """The one figure of a filter_stats dict the run-time guard holds against the margin: the largest coarse error the call saw
    anywhere -- re-evaluated points and audit sample -- and inf when any unmarked sign was wrong."""
    if st["n_mismatch"]:
        return float("inf")
    return max(st["max_err"], st["audit_err"])


_axis_cache = {}


def _axis_table(R, radius, device):
    key = (R, float(radius), device)
    if key not in _axis_cache:
        _axis_cache[key] = grid_axis_coords(R, radius).to(device)
    return _axis_cache[key]


# ----------------------------------------------------------------------------------------------
# marching cubes
# ----------------------------------------------------------------------------------------------
def marching_cubes(vol, level=0.0, reference_order=False, vert_div=1.0, vert_mul=1.0, vert_add=0.0,
                   use_classic=False, slab=None, sign_planes=None):
    """skimage.measure.marching_cubes(vol, level) on the GPU.

    reference_order=False: (verts f32[nv,3] voxel units, faces i32[nf,3]) exactly as skimage returns.
    reference_order=True : faces int64 with columns [1,0,2] and verts/(vert_div)*vert_mul+vert_add
                           (MarchingCubeHelper.forward + scale_tensor, isosurface.py:49-53, system.py:185-189).
    Raises ValueError / RuntimeError like skimage for an out-of-range level / empty surface.
    slab=dict(axis0_offset=int, halo_low=bool): slab mode for the axis-0 split (sculptmate_amd/slab.py);
        returns (verts, faces, top_plane_map int32[2,n1,n2], (min, max)) and never raises on an empty slab.
    sign_planes (uint32 / int32 [n0*n1][words >= ceil(n2/32)], bit i2%32 of word i2/32 = vol > level, bits past n2 zero; not in
        slab mode): the count phase takes the signs from them and reads the volume only where a cell is active
        (sculpt_mc_count_launch_signed) -- what density_grid_filtered leaves behind for level 0.  Same mesh; an empty result is
        re-run without the planes so that skimage's two errors stay apart.
    """
    vol = _req(vol, torch.float32, "vol")
    assert vol.dim() == 3
    n0, n1, n2 = vol.shape
    # The workspace holds 8 bytes per ACTIVE cell in a record pool (default: one active cell per 8 cells).  A shape whose count
    # phase ran out of pool once (SCULPT_ERR_MC_WORKSPACE: a noisy volume) keeps the larger capacity.
    rec_cap = _MC_REC_CAPACITY.get((vol.device, n0, n1, n2), 0)
    ws = _workspace(("mc", vol.device), lib.sculpt_mc_workspace_bytes_for(n0, n1, n2, rec_cap), vol.device)
    flags = 0
    if reference_order:
        flags |= _lib.MC_REFERENCE_ORDER | _lib.MC_FACES_I64
    if use_classic:
        flags |= _lib.MC_USE_CLASSIC
    off = 0
    if slab is not None:
        flags |= _lib.MC_SLAB
        if slab.get("halo_low"):
            flags |= _lib.MC_SLAB_HALO_LOW
        off = int(slab.get("axis0_offset", 0))
    nv, nf = ctypes.c_int64(), ctypes.c_int64()
    mm = (ctypes.c_float * 2)()
    fdt = torch.int64 if reference_order else torch.int32
    # Speculative emit: a call of a shape / flag combination seen before sizes its outputs by the largest mesh so far + 25 % and
    # queues count AND emit before it reads the counts back, so the stream does not idle through the host's read-allocate-launch
    # round trip; the emit kernels write nothing when the mesh does not fit, and the exact path below runs instead.
    key = (vol.device, n0, n1, n2, flags)
    cap = _MC_CAPACITY.get(key) if (slab is None and _MC_SPECULATE) else None
    verts = faces = None
    rflags = flags
    if sign_planes is not None:
        assert slab is None and sign_planes.dim() == 2 and sign_planes.shape[0] == n0 * n1 and sign_planes.stride(1) == 1
        assert sign_planes.element_size() == 4 and sign_planes.shape[1] >= (n2 + 31) // 32
        _check_sign_planes_current(sign_planes)
        rflags = flags | _lib.MC_SIGNED

    def launch_count():
        if sign_planes is not None:
            check(lib.sculpt_mc_count_launch_for(_ptr(vol), _ptr(sign_planes), sign_planes.stride(0), n0, n1, n2, float(level), rflags,
                                                 rec_cap, _ptr(ws), _stream()))
        else:
            check(lib.sculpt_mc_count_launch_for(_ptr(vol), None, 0, n0, n1, n2, float(level), flags, rec_cap, _ptr(ws), _stream()))

    launch_count()
    if cap is not None:
        verts = torch.empty((cap[0], 3), dtype=torch.float32, device=vol.device)
        faces = torch.empty((cap[1], 3), dtype=fdt, device=vol.device)
        check(lib.sculpt_mc_emit_capped(_ptr(vol), n0, n1, n2, float(level), flags, _ptr(ws), float(vert_div), float(vert_mul),
                                        float(vert_add), off, _ptr(verts), cap[0], _ptr(faces), cap[1], None, _stream()))
    nact = ctypes.c_int64()
    rc = lib.sculpt_mc_count_read_ex(n0, n1, n2, float(level), rflags, _ptr(ws), ctypes.byref(nv), ctypes.byref(nf),
                                     ctypes.cast(mm, ctypes.c_void_p), ctypes.byref(nact), _stream())
    while rc == _lib.ERR_MC_WORKSPACE:
        # more active cells than the pool holds (the speculative emit above wrote nothing): a larger workspace, the count again.
        # (The pool is handed out in up to 64 parts: half as much again covers their uneven loads; a pool that holds every cell
        # of the grid cannot overflow, so the loop ends.)
        rec_cap = max(2 * rec_cap, int(nact.value + nact.value // 2 + 4096))
        _MC_REC_CAPACITY[(vol.device, n0, n1, n2)] = rec_cap
        ws = _workspace(("mc", vol.device), lib.sculpt_mc_workspace_bytes_for(n0, n1, n2, rec_cap), vol.device)
        cap = None
        launch_count()
        rc = lib.sculpt_mc_count_read_ex(n0, n1, n2, float(level), rflags, _ptr(ws), ctypes.byref(nv), ctypes.byref(nf),
                                         ctypes.cast(mm, ctypes.c_void_p), ctypes.byref(nact), _stream())
    if sign_planes is not None and rc == _lib.ERR_MC_EMPTY:
        # no surface: whether that is skimage's ValueError (level outside the data range) or its RuntimeError needs the range
        return marching_cubes(vol, level, reference_order, vert_div, vert_mul, vert_add, use_classic, slab)
    if rc == _lib.ERR_MC_LEVEL:
        raise ValueError(_lib.last_error())
    if rc == _lib.ERR_MC_EMPTY:
        raise RuntimeError(_lib.last_error())
    check(rc)
    if slab is None and _MC_SPECULATE:
        old = _MC_CAPACITY.get(key, (0, 0))
        _MC_CAPACITY[key] = (max(old[0], nv.value + nv.value // 4 + 1024), max(old[1], nf.value + nf.value // 4 + 1024))
    if cap is not None and nv.value <= cap[0] and nf.value <= cap[1]:
        return verts[:nv.value], faces[:nf.value]      # views of the capacity-sized buffers (fresh per call)
    verts = torch.empty((nv.value, 3), dtype=torch.float32, device=vol.device)
    faces = torch.empty((nf.value, 3), dtype=fdt, device=vol.device)
    top = torch.empty((2, n1, n2), dtype=torch.int32, device=vol.device) if slab is not None else None
    if nv.value > 0 or slab is not None:
        # dummy non-null pointers for empty outputs
        vp = verts if nv.value else torch.empty(3, device=vol.device)
        fp = faces if nf.value else torch.empty(3, dtype=faces.dtype, device=vol.device)
        check(lib.sculpt_mc_emit(_ptr(vol), n0, n1, n2, float(level), flags, _ptr(ws), float(vert_div),
                                 float(vert_mul), float(vert_add), off, _ptr(vp), _ptr(fp), _ptr(top), _stream()))
    if slab is not None:
        return verts, faces, top, (float(mm[0]), float(mm[1]))
    return verts, faces


_MC_CAPACITY = {}     # (device, n0, n1, n2, flags) -> (vertex capacity, face capacity) of the speculative emit
_MC_REC_CAPACITY = {} # (device, n0, n1, n2) -> active-cell records of the workspace once the default pool was too small
_MC_SPECULATE = not _lib.form_has("SCULPT_MC_FORM", "nospeculate")   # A/B: the two-phase path every time


# ----------------------------------------------------------------------------------------------
# transformer primitives (bf16 storage as torch.bfloat16 tensors; fp32 accumulate)
# ----------------------------------------------------------------------------------------------
BF16 = torch.bfloat16
LN_SLOT = 64  # columns per LayerNorm statistics slice (csrc/gemm.hip)


def gemm(A, W, bias=None, residual=None, out_f32=None, out_bf16=None, out_t=None, M=None, epilogue=0, n_split=0,
         ln_stats=None, ln_colsum=None, ln_eps=1e-5, stats_out=None):
    """out[m][n] = epi(A[m][:] . W[n][:] + bias[n]) (+ residual[m][n]); see sculpt_gemm_bf16.
    A [>=M][K] bf16, W [N or 2N][K] bf16 (row stride = K).  Outputs are caller-allocated.
    LayerNorm fold (sculpt_gemm_bf16_ln): ln_stats [K/64][rows][2] + ln_colsum [N] -> A holds the un-normalised rows, W / bias
    have gamma / beta folded in (fold_layernorm); stats_out [N/64][rows][2] receives the slice statistics of out_f32."""
    K = A.shape[1]
    N = W.shape[0] // 2 if epilogue == _lib.EPI_GEGLU else W.shape[0]
    M = A.shape[0] if M is None else M
    ldo = (out_f32 if out_f32 is not None else out_bf16).stride(0) if (out_f32 is not None or out_bf16 is not None) else 0
    if out_f32 is not None and out_bf16 is not None:
        assert out_f32.stride(0) == out_bf16.stride(0)
    ln = None
    if _TILE_ROWS[0] and ln_stats is None and stats_out is None:
        ln = _lib.LnFold(None, 0, None, float(ln_eps), None, 0, int(_TILE_ROWS[0]))
    if ln_stats is not None or stats_out is not None:
        # statistics arrays are slice-major: [K/64 or N/64][rows][2]
        # (a view of a band of rows, [:, r0:r1], keeps the plane stride of the whole array: ld comes from the stride)
        ref = ln_stats if ln_stats is not None else stats_out
        ld = ref.stride(0) // 2
        ln = _lib.LnFold(_ptr(ln_stats), K // LN_SLOT if ln_stats is not None else 0, _ptr(ln_colsum), float(ln_eps), _ptr(stats_out), ld,
                         int(_TILE_ROWS[0]))
        if ln_stats is not None:
            assert ln_stats.dtype == torch.float32 and ln_stats.shape[0] >= K // LN_SLOT and ln_stats.stride(0) == 2 * ld and ln_colsum is not None
            assert ln_stats.shape[1] >= M and ln_stats.stride(1) == 2
        if stats_out is not None:
            assert stats_out.dtype == torch.float32 and stats_out.shape[0] >= N // LN_SLOT and stats_out.stride(0) == 2 * ld
            assert stats_out.shape[1] >= M and stats_out.stride(1) == 2
    check(lib.sculpt_gemm_bf16_ln(_ptr(A), A.stride(0), _ptr(W), W.stride(0), _ptr(bias), _ptr(residual),
                                  residual.stride(0) if residual is not None else 0, _ptr(out_f32), _ptr(out_bf16),
                                  ldo, _ptr(out_t), out_t.stride(0) if out_t is not None else 0, int(n_split), 0, M, N, K,
                                  epilogue, ctypes.byref(ln) if ln is not None else None, _stream()))


class _TileRows(threading.local):
    def __init__(self):
        self.v = 0

    def __getitem__(self, i):
        return self.v

    def __setitem__(self, i, value):
        self.v = value


_TILE_ROWS = _TileRows()   # per thread: 0, or the rows of one image while a stacked pass is being issued


class single_image_tiles:
    """with ops.single_image_tiles(rows): every bf16 GEMM issued inside chooses its tile form as for `rows` activation rows
    (sculpt_ln_fold_t::rows_per_image) -- a batched pass over stacked token rows then gives each image the bits of its own pass."""

    def __init__(self, rows):
        self.rows = int(rows)

    def __enter__(self):
        self.prev = _TILE_ROWS[0]
        _TILE_ROWS[0] = self.rows

    def __exit__(self, *exc):
        _TILE_ROWS[0] = self.prev
        return False


def fold_layernorm(W, bias, gamma, beta):
    """Host-side (load-time) fold of y = LayerNorm(x) * gamma + beta into the Linear that consumes it (fp32 tensors):
       W' = W * gamma (rounded to bf16 by the caller's upload), bias' = bias + W . beta, colsum = sum_k bf16(W')[n][k].
    Returns (W' fp32, bias' fp32, colsum fp32)."""
    W = W.detach().to(torch.float32).cpu()
    Wp = W * gamma.detach().to(torch.float32).cpu()[None, :]
    bp = (W.double() @ beta.detach().double().cpu())
    if bias is not None:
        bp = bp + bias.detach().double().cpu()
    colsum = Wp.to(BF16).double().sum(1)  # of the values the GEMM really multiplies by: the mean term cancels exactly
    return Wp, bp.to(torch.float32), colsum.to(torch.float32)


def row_slice_stats(x, stats, x_bf16=None, rows=None):
    """(mean, M2) of every 64-column slice of the fp32 rows x -> stats [cols/64][rows][2] (+ the bf16 copy of x)."""
    rows = x.shape[0] if rows is None else rows
    check(lib.sculpt_row_slice_stats(_ptr(x), x.stride(0), rows, x.shape[1], _ptr(stats), stats.shape[1], _ptr(x_bf16),
                                     x_bf16.stride(0) if x_bf16 is not None else 0, _stream()))


def attention(Q, K, Vt, O, Tq, Tk, heads, scale, batch=1, q_bs=0, k_bs=0, vt_bs=0, o_bs=0):
    """O = softmax(Q K^T scale) V per head (D=64); Vt is V transposed [heads*64][ld >= round_up(Tk,64)].
    scale=None: the queries already carry softmax_scale * log2(e) (sculpt_attention_bf16_prescaled).
    batch > 1: that many independent attentions in one launch (sculpt_attention_bf16_batched); entry b reads Q / K rows
    b*q_bs / b*k_bs ELEMENTS further on, V^T b*vt_bs elements (a column offset inside one array, or a whole array) and writes
    O + b*o_bs."""
    if batch > 1:
        check(lib.sculpt_attention_bf16_batched(_ptr(Q), Q.stride(0), int(q_bs), _ptr(K), K.stride(0), int(k_bs), _ptr(Vt),
                                                Vt.stride(0), int(vt_bs), _ptr(O), O.stride(0), int(o_bs), Tq, Tk, heads, int(batch),
                                                1 if scale is None else 0, 0.0 if scale is None else float(scale), _stream()))
    elif scale is None:
        check(lib.sculpt_attention_bf16_prescaled(_ptr(Q), Q.stride(0), _ptr(K), K.stride(0), _ptr(Vt), Vt.stride(0), _ptr(O),
                                                  O.stride(0), Tq, Tk, heads, _stream()))
    else:
        check(lib.sculpt_attention_bf16(_ptr(Q), Q.stride(0), _ptr(K), K.stride(0), _ptr(Vt), Vt.stride(0), _ptr(O),
                                        O.stride(0), Tq, Tk, heads, float(scale), _stream()))


def layernorm(x, gamma, beta, eps, y=None, y_f32=None, rows=None, y_lt=None):
    """y_lt (a Limbs, fp32 x only): the normalised rows as three bf16 limbs, limb-tiled (sculpt_layernorm_limbs); y_f32 optional."""
    rows = x.shape[0] if rows is None else rows
    if y_lt is not None:
        assert x.dtype == torch.float32 and y is None and y_lt.cols == x.shape[1] and y_lt.rows >= rows
        check(lib.sculpt_layernorm_limbs(_ptr(x), x.stride(0), _ptr(gamma), _ptr(beta), float(eps), _ptr(y_lt.data), y_lt.code, _ptr(y_f32),
                                         y_f32.stride(0) if y_f32 is not None else 0, rows, x.shape[1], _stream()))
        return
    xf = x if x.dtype == torch.float32 else None
    xb = x if x.dtype == BF16 else None
    ldy = (y if y is not None else y_f32).stride(0)
    check(lib.sculpt_layernorm(_ptr(xf), _ptr(xb), x.stride(0), _ptr(gamma), _ptr(beta), float(eps), _ptr(y), ldy,
                               _ptr(y_f32), rows, x.shape[1], _stream()))


def groupnorm_tokens(x_ct, groups, gamma, beta, eps, y_tc, stats_ws):
    C, T = x_ct.shape
    yb = y_tc if y_tc.dtype == BF16 else None
    yf = y_tc if y_tc.dtype == torch.float32 else None
    check(lib.sculpt_groupnorm_tokens(_ptr(x_ct), C, T, groups, _ptr(gamma), _ptr(beta), float(eps), _ptr(yb), _ptr(yf),
                                      _ptr(stats_ws), _stream()))


def transpose_add(x_tc, residual_ct, out_ct):
    T, C = x_tc.shape
    check(lib.sculpt_transpose_add(_ptr(x_tc), _ptr(residual_ct), _ptr(out_ct), T, C, _stream()))


def vit_patchify(image_hwc, patch, mean, std, patches):
    S = image_hwc.shape[0]
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    pb = patches if patches.dtype == BF16 else None
    pf = patches if patches.dtype == torch.float32 else None
    check(lib.sculpt_vit_patchify(_ptr(image_hwc), S, patch, ctypes.cast(m, ctypes.c_void_p),
                                  ctypes.cast(s, ctypes.c_void_p), _ptr(pb), _ptr(pf), patches.stride(0), _stream()))


def vit_assemble(patch_out, cls, pos, tokens):
    n_patches, hidden = patch_out.shape[0], patch_out.shape[1]
    check(lib.sculpt_vit_assemble(_ptr(patch_out), _ptr(cls), _ptr(pos), _ptr(tokens), n_patches, hidden, _stream()))


def upsample_scatter(g, bias, planes, S, Co):
    check(lib.sculpt_upsample_scatter(_ptr(g), g.stride(0), _ptr(bias), _ptr(planes), S, Co, _stream()))


def cast_bf16(x, y):
    check(lib.sculpt_cast_bf16(_ptr(x), _ptr(y), x.numel(), _stream()))


# ----------------------------------------------------------------------------------------------
# UV-space texture baker (StableFast)
# ----------------------------------------------------------------------------------------------
def bake_rasterize(uv, face_indices, bake_resolution):
    """TextureBaker.rasterize (sf3d/texture_baker/baker.py:12-59) on the GPU -> f32 [res, res, 4]."""
    uv = _req(uv.contiguous(), torch.float32, "uv")
    idx = _req(face_indices.to(torch.int32).contiguous(), torch.int32, "face_indices")
    res = int(bake_resolution)
    ws = _workspace(("bake", uv.device), lib.sculpt_bake_workspace_bytes(res), uv.device)
    out = torch.empty((res, res, 4), dtype=torch.float32, device=uv.device)
    check(lib.sculpt_bake_rasterize(_ptr(uv), uv.shape[0], _ptr(idx), idx.shape[0], res, _ptr(ws), _ptr(out), _stream()))
    return out


def bake_interpolate(attr, rast, face_indices):
    """TextureBaker.interpolate (baker.py:71-120) on the GPU -> f32 [res, res, 3]."""
    attr = _req(attr.contiguous(), torch.float32, "attr")
    rast = _req(rast.contiguous(), torch.float32, "rast")
    idx = _req(face_indices.to(torch.int32).contiguous(), torch.int32, "face_indices")
    res = rast.shape[0]
    out = torch.empty((res, res, 3), dtype=torch.float32, device=attr.device)
    check(lib.sculpt_bake_interpolate(_ptr(attr), attr.shape[0], _ptr(idx), idx.shape[0], _ptr(rast), res, _ptr(out), _stream()))
    return out


def resize_aa_bilinear(img_hwc, size):
    """F.interpolate(bilinear, align_corners=False, antialias=True) to (size, size) on an HWC fp32 device image
    (ImagePreprocessor.convert_and_resize, tsr/utils.py:82-88)."""
    img = _req(img_hwc.contiguous(), torch.float32, "image")
    H, W, C = img.shape
    tmp = torch.empty((H, size, C), dtype=torch.float32, device=img.device)
    out = torch.empty((size, size, C), dtype=torch.float32, device=img.device)
    check(lib.sculpt_resize_aa_bilinear(_ptr(img), H, W, C, _ptr(tmp), _ptr(out), size, size, _stream()))
    return out


# ----------------------------------------------------------------------------------------------
# StableFast geometry tail
# ----------------------------------------------------------------------------------------------
def dilate_fill(img, mask, iterations=10):
    """dilate_fill (sf3d/models/utils.py:96-133): img [1,3,H,W] f32, mask [1,1,H,W] (bool or float) -> [1,3,H,W]."""
    img = _req(img.contiguous(), torch.float32, "img")
    H, W = img.shape[-2:]
    m = _req(mask.to(torch.float32).contiguous(), torch.float32, "mask")
    scratch = torch.empty(8 * H * W, dtype=torch.float32, device=img.device)
    out = torch.empty_like(img)
    check(lib.sculpt_dilate_fill(_ptr(img), _ptr(m), H, W, int(iterations), _ptr(scratch), _ptr(out), _stream()))
    return out


def vertex_normals(v_pos, faces):
    """Mesh._compute_vertex_normal (sf3d/models/mesh.py:66-92)."""
    v = _req(v_pos.contiguous(), torch.float32, "v_pos")
    f = faces.contiguous()
    out = torch.empty_like(v)
    check(lib.sculpt_vertex_normals(_ptr(v), v.shape[0], _ptr(f), int(f.dtype == torch.int64), f.shape[0], _ptr(out), _stream()))
    return out


def vertex_tangents(v_pos, v_tex, v_nrm, faces):
    """Mesh._compute_vertex_tangent (sf3d/models/mesh.py:94-139)."""
    v = _req(v_pos.contiguous(), torch.float32, "v_pos")
    t = _req(v_tex.contiguous(), torch.float32, "v_tex")
    n = _req(v_nrm.contiguous(), torch.float32, "v_nrm")
    f = faces.contiguous()
    cnt = torch.empty(v.shape[0], dtype=torch.float32, device=v.device)
    out = torch.empty_like(v)
    check(lib.sculpt_vertex_tangents(_ptr(v), _ptr(t), _ptr(n), v.shape[0], _ptr(f), int(f.dtype == torch.int64), f.shape[0],
                                     _ptr(cnt), _ptr(out), _stream()))
    return out


# ----------------------------------------------------------------------------------------------
# StableFast-3D networks: 3x3 conv as im2col + GEMM, pixel shuffle, marching tetrahedra
# ----------------------------------------------------------------------------------------------
def im2col3x3(act, n_planes, S, out):
    """act [n_planes*S*S][C] channel-last (bf16 or f32) -> out [n_planes*S*S][9*C], k = (ky*3+kx)*C + c."""
    C = act.shape[1]
    assert act.is_contiguous() and out.is_contiguous() and out.shape == (n_planes * S * S, 9 * C) and out.dtype == act.dtype
    check(lib.sculpt_im2col3x3(_ptr(act), n_planes, S, C, act.element_size(), _ptr(out), _stream()))


def conv3x3_planes(act, n_planes, S, W2, bias, out_f32=None, out_bf16=None, relu=False):
    """3x3 / pad 1 convolution of n_planes S x S channel-last images act [n_planes*S*S][C] (C % 64 == 0), implicit GEMM."""
    C = act.shape[1]
    assert act.is_contiguous() and W2.shape[1] == 9 * C
    o = out_f32 if out_f32 is not None else out_bf16
    check(lib.sculpt_conv3x3_bf16(_ptr(act), act.stride(0), n_planes, S, S, C, 1, _ptr(W2), _ptr(bias), _ptr(out_f32), _ptr(out_bf16),
                                  o.stride(0), 0, W2.shape[0], _lib.EPI_RELU if relu else _lib.EPI_NONE, _stream()))


def resize_bilinear_hwc(image_hwc, size, mul_hw=None, out=None):
    """F.interpolate(bilinear, align_corners=False, no antialias) of an fp32 [H][W][C] image to size x size; every source
    pixel is first multiplied by mul_hw [H][W] when given (sculpt_resize_bilinear_hwc)."""
    H, W, C = image_hwc.shape
    assert image_hwc.dtype == torch.float32 and image_hwc.is_contiguous()
    if mul_hw is not None:
        mul_hw = mul_hw.reshape(H, W)
        assert mul_hw.dtype == torch.float32 and mul_hw.is_contiguous()
    if out is None:
        out = torch.empty((size, size, C), dtype=torch.float32, device=image_hwc.device)
    check(lib.sculpt_resize_bilinear_hwc(_ptr(image_hwc), _ptr(mul_hw), H, W, C, _ptr(out), size, size, _stream()))
    return out


def im2col3x3_strided(act, n_groups, S, stride, out):
    """act [n_groups*S*S][C] channel-last groups (bf16 or f32) -> out [So*So][9*n_groups*C] of the 3x3 / padding 0 /
    stride convolution over the channel-concatenated image, k = (ky*3+kx)*n_groups*C + g*C + c."""
    C = act.shape[1]
    So = (S - 3) // stride + 1
    assert act.is_contiguous() and act.shape[0] == n_groups * S * S and out.is_contiguous() and out.dtype == act.dtype
    assert out.shape == (So * So, 9 * n_groups * C)
    check(lib.sculpt_im2col3x3_strided(_ptr(act), n_groups, S, C, act.element_size(), stride, _ptr(out), _stream()))
    return So


def col_reduce(x, rows, out, mean=False):
    """out[c] = max (or mean) over the first `rows` rows of fp32 x [*, cols] (sculpt_col_reduce_f32)."""
    assert x.dtype == torch.float32 and out.dtype == torch.float32 and x.stride(1) == 1
    check(lib.sculpt_col_reduce_f32(_ptr(x), x.stride(0), rows, x.shape[1], 1 if mean else 0, _ptr(out), _stream()))
    return out


def pixel_shuffle(g, planes, n_planes, S, Co, r):
    """nn.PixelShuffle(r) of g f32 [n_planes*S*S][>= Co*r*r] into planes f32 [n_planes][Co][S*r][S*r]."""
    g = _req(g, torch.float32, "g")
    check(lib.sculpt_pixel_shuffle(_ptr(g), g.stride(0), _ptr(planes), n_planes, S, Co, r, _stream()))


def normalize_rows3(x, eps=1e-7):
    """F.normalize(x, dim=-1, eps=eps) for [N,3] fp32."""
    x = _req(x.contiguous(), torch.float32, "x")
    y = torch.empty_like(x)
    check(lib.sculpt_normalize_rows3(_ptr(x), x.shape[0], float(eps), _ptr(y), _stream()))
    return y


def bake_material(rast, color, perturb_normal=None, nrm=None, tng=None):
    """system.py:375-440 per texel -> (albedo [res,res,3], bump [res,res,3] or None)."""
    res = rast.shape[0]
    rast = _req(rast.contiguous(), torch.float32, "rast")
    albedo = torch.empty((res, res, 3), dtype=torch.float32, device=rast.device)
    bump = torch.empty_like(albedo) if perturb_normal is not None else None
    args = [None if t is None else _req(t.reshape(-1, 3).contiguous(), torch.float32, "texel image")
            for t in (color, perturb_normal, nrm, tng)]
    check(lib.sculpt_bake_material(_ptr(rast), res, _ptr(args[0]), _ptr(args[1]), _ptr(args[2]), _ptr(args[3]),
                                   _ptr(albedo), _ptr(bump), _stream()))
    return albedo, bump


def uv_cell_atlas(v_pos, faces, padding=0.05):
    """One grid cell per triangle (stand-in unwrapper) -> (uv f32 [3*Nf, 2], indices i64 [Nf, 3] = arange)."""
    v = _req(v_pos.contiguous(), torch.float32, "v_pos")
    f = faces.contiguous()
    nf = f.shape[0]
    cols = max(1, int(np.ceil(np.sqrt(nf))))
    rows = max(1, -(-nf // cols))
    uv = torch.empty((3 * nf, 2), dtype=torch.float32, device=v.device)
    check(lib.sculpt_uv_cell_atlas(_ptr(v), _ptr(f), int(f.dtype == torch.int64), nf, cols, rows, float(padding), _ptr(uv),
                                   _stream()))
    return uv, torch.arange(3 * nf, device=v.device, dtype=torch.int64).reshape(-1, 3)


class TetGrid:
    """Static tetrahedral grid tables resident in HBM (see sculptmate_amd/sf3d/tets.py)."""

    def __init__(self, vertices, indices, device, edges=None, tet_edges=None):
        from .sf3d.tets import edge_tables

        v = np.ascontiguousarray(vertices, np.float32)
        t = np.ascontiguousarray(indices, np.int64)
        if edges is None or tet_edges is None:
            edges, tet_edges = edge_tables(t, v.shape[0])
        self.n_vertices, self.n_tets, self.n_edges = v.shape[0], t.shape[0], edges.shape[0]
        self.vertices = torch.from_numpy(v).to(device)
        self.tets = torch.from_numpy(t.astype(np.int32)).to(device)
        self.edges = torch.from_numpy(np.ascontiguousarray(edges, np.int32)).to(device)
        self.tet_edges = torch.from_numpy(np.ascontiguousarray(tet_edges, np.int32)).to(device)
        self.ws = torch.empty(lib.sculpt_mtet_workspace_bytes(self.n_edges, self.n_tets), dtype=torch.uint8, device=device)


def mtet_deform(grid: TetGrid, offsets, resolution):
    """grid_vertices + (1/resolution) * tanh(offsets)  (isosurface.py:108-115)."""
    off = _req(offsets.reshape(-1, 3).contiguous(), torch.float32, "offsets")
    out = torch.empty_like(grid.vertices)
    check(lib.sculpt_mtet_deform(_ptr(grid.vertices), _ptr(off), grid.n_vertices, float((1 - 0) / resolution), _ptr(out),
                                 _stream()))
    return out


def marching_tets(grid: TetGrid, pos, sdf, vert_mul=1.0, vert_add=0.0):
    """MarchingTetrahedraHelper._forward (isosurface.py:142-209) -> (verts f32 [Nv,3] * vert_mul + vert_add, faces i64)."""
    pos = _req(pos.contiguous(), torch.float32, "pos")
    sdf = _req(sdf.reshape(-1).contiguous(), torch.float32, "sdf")
    assert pos.shape == (grid.n_vertices, 3) and sdf.shape[0] == grid.n_vertices
    nv, nf = ctypes.c_int64(), ctypes.c_int64()
    check(lib.sculpt_mtet_count(_ptr(sdf), _ptr(grid.tets), grid.n_tets, _ptr(grid.edges), grid.n_edges, _ptr(grid.ws),
                                ctypes.byref(nv), ctypes.byref(nf), _stream()))
    verts = torch.empty((nv.value, 3), dtype=torch.float32, device=pos.device)
    faces = torch.empty((nf.value, 3), dtype=torch.int64, device=pos.device)
    if nv.value and nf.value:
        check(lib.sculpt_mtet_emit(_ptr(pos), _ptr(sdf), _ptr(grid.tets), grid.n_tets, _ptr(grid.edges), grid.n_edges,
                                   _ptr(grid.tet_edges), _ptr(grid.ws), float(vert_mul), float(vert_add), _ptr(verts),
                                   _ptr(faces), _stream()))
    return verts, faces


# ----------------------------------------------------------------------------------------------
# fp32 parity mode primitives
# ----------------------------------------------------------------------------------------------
def gemm_f32(A, W, bias=None, residual=None, out=None, out_t=None, M=None, N=None, epilogue=0, n_split=0, w_rows=0,
             alpha=1.0, l3=False, batch=1, a_bs=0, w_bs=0, o_bs=0):
    """out[m][n] = epi(alpha * A[m][:] . W[n][:] + bias[n]) (+ residual); fp32 in and out (sculpt_gemm_f32_ex).
    l3=False: the exact-fp32 matrix instruction.  l3=True: fp32 arithmetic on the bf16 matrix pipe through the exact
    three-limb split of both operands (K % 32 == 0).  batch > 1: grid z, entry z at A + z*a_bs, W + z*w_bs, out + z*o_bs."""
    K = A.shape[1]
    if N is None:
        N = W.shape[0] // 2 if epilogue == _lib.EPI_GEGLU else W.shape[0]
    M = A.shape[0] if M is None else M
    check(lib.sculpt_gemm_f32_ex(_ptr(A), A.stride(0), _ptr(W), W.stride(0), _ptr(bias), _ptr(residual),
                                 residual.stride(0) if residual is not None else 0, _ptr(out),
                                 out.stride(0) if out is not None else 0, _ptr(out_t),
                                 out_t.stride(0) if out_t is not None else 0, int(n_split), int(w_rows), M, N, K,
                                 float(alpha), epilogue, _lib.F32_BF16L3 if l3 else _lib.F32_EXACT, int(batch), int(a_bs), int(w_bs),
                                 int(o_bs), _stream()))


LIMB_FORMATS = {"bf16x3": _lib.LIMBS_BF16X3, "f16x2": _lib.LIMBS_F16X2}


class Limbs:
    """An fp32 matrix [rows][cols] held as 16-bit limbs in the limb-tiled layout of csrc/limbs.h: the operand form of `gemm_l3p`
    (weights split once at load time, activations written this way by the kernel that produces them).
    fmt "bf16x3": three bf16 limbs, exact; "f16x2": two fp16 limbs (22 bits, |x| < 65504) -- a weight is then stored times the
    power of two `scale` that puts its largest magnitude in [2^14, 2^15) and the GEMM multiplies by alpha = 1 / scale."""
    __slots__ = ("data", "rows", "cols", "fmt", "scale")

    def __init__(self, rows, cols, device=None, data=None, zero=False, fmt="bf16x3", scale=1.0):
        self.rows, self.cols, self.fmt, self.scale = int(rows), int(cols), fmt, float(scale)
        self.data = data if data is not None else limbs_empty(rows, cols, device, zero=zero, fmt=fmt)

    @staticmethod
    def of(x, fmt="bf16x3", weight=False):
        """weight=True (f16x2 only): stored pre-scaled into the fp16 range (see the class)."""
        scale = 1.0
        if weight and fmt == "f16x2":
            mx = float(x.abs().max())
            if mx > 0.0 and np.isfinite(mx):
                scale = 2.0 ** (14 - int(np.floor(np.log2(mx))))
        return Limbs(x.shape[0], x.shape[1], data=limbs_split(x, fmt=fmt, scale=scale), fmt=fmt, scale=scale)

    @property
    def code(self):
        return LIMB_FORMATS[self.fmt]

    def float(self):
        return limbs_join(self.data, self.rows, self.cols, self.fmt) / self.scale


def limbs_bytes(rows, K, fmt="bf16x3"):
    return int(lib.sculpt_limbs_bytes(int(rows), int(K), LIMB_FORMATS[fmt]))


def limbs_empty(rows, K, device, zero=False, fmt="bf16x3"):
    """An uninitialised limb-tiled [rows][K] matrix (sculpt_limbs_split's layout; a flat uint8 tensor)."""
    return (torch.zeros if zero else torch.empty)(limbs_bytes(rows, K, fmt), dtype=torch.uint8, device=device)


def limbs_split(x, out=None, fmt="bf16x3", scale=1.0):
    """scale * fp32 [rows][K] -> the limb-tiled form the `gemm_l3p` operands take (bf16x3: exact; f16x2: 22 bits)."""
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    rows, K = x.shape
    if out is None:
        out = limbs_empty(rows, K, x.device, fmt=fmt)
    check(lib.sculpt_limbs_split(_ptr(x), x.stride(0), rows, K, float(scale), LIMB_FORMATS[fmt], _ptr(out), _stream()))
    return out


def limbs_join(lt, rows, K, fmt="bf16x3"):
    """The fp32 matrix a limb-tiled array stands for (tests): the sum of the limbs per element."""
    nl = 2 if fmt == "f16x2" else 3
    v = lt.view(torch.float16 if fmt == "f16x2" else torch.bfloat16).view(-1, K // 8, nl, 32, 8).to(torch.float32)   # [block][chunk][limb][row][k]
    x = v[:, :, 0] + v[:, :, 1]
    if nl == 3:
        x = x + v[:, :, 2]                                                         # exact: the limbs do not overlap
    return x.permute(0, 2, 1, 3).reshape(-1, K)[:rows].contiguous()


def geglu_row_blocks(W):
    """A GEGLU weight [2N][K] (value rows, then gate rows) with its 32-row blocks in gemm_l3p's tile order."""
    N = W.shape[0] // 2
    assert N % 64 == 0
    idx = torch.arange(N, device=W.device).view(N // 64, 2, 32)                    # [tile][half][row]
    order = torch.stack([idx[:, 0], idx[:, 0] + N, idx[:, 1], idx[:, 1] + N], 1).reshape(-1)
    return W.index_select(0, order)


def gemm_l3p(A_lt, W_lt, M, N, K, bias=None, residual=None, out=None, out_t=None, n_split=0, out_lt=None, epilogue=0, fmt=None,
             alpha=None):
    """sculpt_gemm_l3p: the limb GEMM on operands split once (A_lt [M][K], W_lt [N or 2N][K] limb-tiled; Limbs or raw tensors
    with fmt / alpha given).  A Limbs weight brings its format and alpha = 1 / scale."""
    if fmt is None:
        fmt = W_lt.fmt if isinstance(W_lt, Limbs) else "bf16x3"
    if alpha is None:
        alpha = 1.0 / (W_lt.scale * (A_lt.scale if isinstance(A_lt, Limbs) else 1.0)) if isinstance(W_lt, Limbs) else 1.0
    out_fmt = out_lt.fmt if isinstance(out_lt, Limbs) else fmt
    rows_w = 2 * N if epilogue == _lib.EPI_GEGLU else N
    for t, r, c, what, f in ((A_lt, M, K, "A", fmt), (W_lt, rows_w, K, "W", fmt), (out_lt, M, N, "out", out_fmt)):
        if isinstance(t, Limbs):
            assert t.cols == c and t.rows >= r and t.fmt == f, "gemm_l3p: %s is a limb-tiled %s [%d][%d], the call needs %s [>= %d][%d]" % (
                what, t.fmt, t.rows, t.cols, f, r, c)
        elif t is not None:
            assert t.numel() * t.element_size() >= limbs_bytes(r, c, f), "gemm_l3p: %s holds fewer bytes than a limb-tiled [%d][%d]" % (what, r, c)
    A_lt, W_lt = getattr(A_lt, "data", A_lt), getattr(W_lt, "data", W_lt)
    out_lt = getattr(out_lt, "data", out_lt)
    check(lib.sculpt_gemm_l3p(_ptr(A_lt), _ptr(W_lt), LIMB_FORMATS[fmt], float(alpha), _ptr(bias), _ptr(residual),
                              residual.stride(0) if residual is not None else 0, _ptr(out), out.stride(0) if out is not None else 0,
                              _ptr(out_t), out_t.stride(0) if out_t is not None else 0, int(n_split), _ptr(out_lt),
                              LIMB_FORMATS[out_fmt], int(M), int(N), int(K), int(epilogue), _stream()))


def softmax_rows_f32(x, rows, cols, pad_cols):
    check(lib.sculpt_softmax_rows_f32(_ptr(x), x.stride(0), rows, cols, pad_cols, _stream()))


def attention_f32_l3_batched(Q, K, Vt, O, Tq, Tk, heads, scale, batch, q_bs, k_bs, vt_bs, o_bs, two_fp16_limbs=False):
    """`batch` fused three-limb attentions in one launch (sculpt_attention_f32_l3_batched); O an fp32 tensor (o_bs in elements) or a
    Limbs (o_bs in elements of its logical [rows][cols] matrix: entry b starts at row b * o_bs / cols)."""
    lt = isinstance(O, Limbs)
    if lt:
        assert o_bs % O.cols == 0
    check(lib.sculpt_attention_f32_l3_batched(_ptr(Q), Q.stride(0), int(q_bs), _ptr(K), K.stride(0), int(k_bs), _ptr(Vt), Vt.stride(0),
                                              int(vt_bs), None if lt else _ptr(O), 0 if lt else O.stride(0), 0 if lt else int(o_bs),
                                              _ptr(O.data) if lt else None, O.code if lt else 0, 0, int(o_bs // O.cols) if lt else 0,
                                              O.cols if lt else 0,
                                              Tq, Tk, heads, int(batch), float(scale), 1 if two_fp16_limbs else 0, _stream()))


def attention_f32(Q, K, Vt, O, Tq, Tk, heads, scale, scores, l3=False, o_row0=0, two_fp16_limbs=False):
    """softmax(Q K^T scale) V per head in fp32: two GEMMs and a row softmax per head.
    Q [Tq][*], K [Tk][*] with head h at columns 64h..; Vt [heads*64][>= round_up(Tk,32)] (zero padded).
    scores: fp32 scratch.  [Tq][>= round_up(Tk,32)]: the heads run one after the other (three launches each);
    [heads][Tq][>= round_up(Tk,32)]: all heads at once, three launches per attention (grid z = head).
    l3: the two products on the three-limb bf16 matrix pipe (gemm_f32 l3=True) instead of the exact-fp32 instruction;
    with scores=None (l3 only) the whole attention is ONE fused launch (sculpt_attention_f32_l3): what TSR(precision="bf16l3") runs."""
    Tkp = ((Tk + 31) // 32) * 32 if l3 else ((Tk + 15) // 16) * 16
    N4 = ((Tk + 3) // 4) * 4
    if scores is None:   # the fused kernel of the three-limb mode: no score matrix in HBM
        assert l3, "the exact-fp32 attention is a composition: it needs the scores scratch"
        if isinstance(O, Limbs):   # the output as limbs: row o_row0 + q of the limb-tiled matrix O
            check(lib.sculpt_attention_f32_l3_limbs(_ptr(Q), Q.stride(0), _ptr(K), K.stride(0), _ptr(Vt), Vt.stride(0), _ptr(O.data),
                                                    O.code, int(o_row0), O.cols, Tq, Tk, heads, float(scale), 1 if two_fp16_limbs else 0,
                                                    _stream()))
            return
        if two_fp16_limbs:   # fp32 output through the batched entry (one entry)
            return attention_f32_l3_batched(Q, K, Vt, O, Tq, Tk, heads, scale, 1, 0, 0, 0, 0, two_fp16_limbs=True)
        check(lib.sculpt_attention_f32_l3(_ptr(Q), Q.stride(0), _ptr(K), K.stride(0), _ptr(Vt), Vt.stride(0), _ptr(O), O.stride(0),
                                          Tq, Tk, heads, float(scale), _stream()))
        return
    if scores.dim() == 3:
        assert scores.shape[0] >= heads and scores.shape[1] == Tq and scores.is_contiguous()
        ld = scores.shape[2]
        flat = scores.view(-1, ld)
        gemm_f32(Q[:, :64], K[:, :64], out=flat, M=Tq, N=N4, w_rows=Tk, alpha=scale, l3=l3, batch=heads, a_bs=64, w_bs=64, o_bs=Tq * ld)
        softmax_rows_f32(flat, heads * Tq, Tk, Tkp)
        gemm_f32(flat[:, :Tkp], Vt[:64, :Tkp], out=O[:, :64], M=Tq, N=64, l3=l3, batch=heads, a_bs=Tq * ld, w_bs=64 * Vt.stride(0), o_bs=64)
        return
    for h in range(heads):
        q, k = Q[:, 64 * h:64 * h + 64], K[:, 64 * h:64 * h + 64]
        gemm_f32(q, k, out=scores, M=Tq, N=N4, w_rows=Tk, alpha=scale, l3=l3)
        softmax_rows_f32(scores, Tq, Tk, Tkp)
        gemm_f32(scores[:, :Tkp], Vt[64 * h:64 * h + 64, :Tkp], out=O[:, 64 * h:64 * h + 64], M=Tq, N=64, l3=l3)


# ----------------------------------------------------------------------------------------------
# channel-last bf16 conv-net blocks (U^2-Net)
# ----------------------------------------------------------------------------------------------
class Act:
    """A slice [H*W][C] of a channel-last bf16 activation buffer (buf: [H*W, ld] bf16, channels off..off+C)."""

    def __init__(self, buf, off, C, H, W):
        assert buf.dtype == BF16 and buf.is_contiguous() and buf.shape[0] == H * W and off % 8 == 0 and C % 8 == 0
        assert off + C <= buf.shape[1]
        self.buf, self.off, self.C, self.H, self.W = buf, off, C, H, W

    @property
    def ptr(self):
        return ctypes.c_void_p(self.buf.data_ptr() + 2 * self.off)

    @property
    def ld(self):
        return self.buf.stride(0)


def conv3x3_bf16(x: Act, W2, bias, out, n_store, dilation, relu, col=None, implicit=None):
    """Conv2d(3x3, padding=dilation, dilation) of the slice x with packed weights W2 [Npad][9*C_pad] (+bias [Npad]).
    out: an Act (bf16 slice, n_store columns written) or an fp32 tensor [H*W, Npad].
    implicit (default: whenever the slice has C_pad readable channels per pixel): no im2col rows, the GEMM fetches the
    shifted pixels itself (sculpt_conv3x3_bf16); otherwise im2col into `col` (bf16 scratch >= H*W*9*C_pad) + GEMM."""
    K = W2.shape[1]
    C_pad = K // 9
    M = x.H * x.W
    assert W2.dtype == BF16
    epi = _lib.EPI_RELU if relu else _lib.EPI_NONE
    can = x.off + C_pad <= x.buf.shape[1] and C_pad % 64 == 0
    if implicit is None:
        implicit = can
    if implicit:
        assert can, "implicit conv needs C_pad readable channels in the slice's rows"
        if isinstance(out, Act):
            check(lib.sculpt_conv3x3_bf16(x.ptr, x.ld, 1, x.H, x.W, C_pad, int(dilation), _ptr(W2), _ptr(bias), None, out.ptr,
                                          out.ld, int(n_store), W2.shape[0], epi, _stream()))
        else:
            check(lib.sculpt_conv3x3_bf16(x.ptr, x.ld, 1, x.H, x.W, C_pad, int(dilation), _ptr(W2), _ptr(bias), _ptr(out), None,
                                          out.stride(0), 0, W2.shape[0], epi, _stream()))
        return
    assert col is not None and col.numel() >= M * K
    check(lib.sculpt_im2col3x3_dilated(x.ptr, x.ld, x.H, x.W, x.C, C_pad, int(dilation), _ptr(col), _stream()))
    if isinstance(out, Act):
        check(lib.sculpt_gemm_bf16_ex(_ptr(col), K, _ptr(W2), K, _ptr(bias), None, 0, None, out.ptr, out.ld, None, 0, 0,
                                      int(n_store), M, W2.shape[0], K, epi, _stream()))
    else:
        check(lib.sculpt_gemm_bf16_ex(_ptr(col), K, _ptr(W2), K, _ptr(bias), None, 0, _ptr(out), None, out.stride(0), None, 0,
                                      0, 0, M, W2.shape[0], K, epi, _stream()))


def maxpool2x2_ceil(x: Act, out: Act):
    assert out.H == (x.H + 1) // 2 and out.W == (x.W + 1) // 2 and out.C == x.C
    check(lib.sculpt_maxpool2x2_ceil(x.ptr, x.ld, x.H, x.W, x.C, out.ptr, out.ld, _stream()))


def upsample_bilinear(x: Act, out: Act):
    assert out.C == x.C
    check(lib.sculpt_upsample_bilinear_bf16(x.ptr, x.ld, x.H, x.W, x.C, out.ptr, out.ld, out.H, out.W, _stream()))


def upsample_bilinear_f32(x, ld, h, w, out, H, W):
    check(lib.sculpt_upsample_bilinear_f32(_ptr(x), int(ld), h, w, _ptr(out), H, W, _stream()))


def add_bf16(a: Act, b: Act, out: Act):
    assert a.C == b.C == out.C and a.H * a.W == out.H * out.W
    check(lib.sculpt_add_bf16(a.ptr, a.ld, b.ptr, b.ld, out.ptr, out.ld, a.H * a.W, a.C, _stream()))


def fuse_sigmoid(maps, w, bias, out):
    check(lib.sculpt_fuse_sigmoid(_ptr(maps), maps.shape[0], maps.shape[1], _ptr(w), float(bias), _ptr(out), _stream()))
