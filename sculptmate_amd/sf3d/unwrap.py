"""Box-projection UV unwrapping on the MI355X: the reference's `Unwrapper` (StableFast/sf3d/uv_unwrapper/unwrap.py:12-697)
as a callable with the hook signature Mesh.unwrap_uv expects,

    unwrapper(v_pos [Nv,3], v_nrm [Nv,3], faces [Nf,3], island_padding) -> (uv [3*Nf, 2], indices [Nf, 3]).

Stages (each one HIP entry point, csrc/uv_unwrap.hip; the reference lines are cited there and in include/sculpt_hip.h):
principal-axis alignment -> projection onto the six cube faces -> per-chart rotation into a consistent tangent frame ->
atlas assignment (front layer / overlap slice / remaining) -> placement in the 3x2 (+ half-size 6x... ) atlas.

Differences from the reference, stated once:
  * the principal axes are the exact eigenvectors of the vertex covariance (3x3 eigen-decomposition on the host from
    nine device-side sums); the reference's randomised rank-2 torch.pca_lowrank lands within ~15 degrees of them.  Signs:
    each axis points to its larger-magnitude side (deterministic);
  * the atlas assignment is the library's own z-buffer algorithm (the reference's lives in uv_unwrapper.dll);
  * (uv, indices) are returned per corner -- indices = arange(3*Nf) -- instead of torch.unique's sorted unique rows plus
    inverse (:529-544): Mesh.unwrap_uv only ever forms uv[indices], which is identical, and the sort is skipped.
"""
import ctypes
import math

import numpy as np
import torch

from .. import _lib
from ..ops import _ptr, _stream, check, lib


def principal_axes(sums9, n):
    """Nine sums (x, y, z, xx, xy, xz, yy, yz, zz) -> (main, secondary) unit eigenvectors of the covariance, float64."""
    s = np.asarray(sums9, np.float64)
    mu = s[:3] / n
    m2 = np.array([[s[3], s[4], s[5]], [s[4], s[6], s[7]], [s[5], s[7], s[8]]]) / n
    cov = m2 - np.outer(mu, mu)
    w, vec = np.linalg.eigh(cov)
    axes = []
    for k in (2, 1):
        a = vec[:, k]
        if a[np.abs(a).argmax()] < 0:
            a = -a
        axes.append(a)
    return axes[0], axes[1]


def axis_rotation(main_axis, secondary_axis):
    """unwrap.py:566-617: orthonormalise (main, secondary, main x secondary) in float32 and sort the three onto the
    canonical axes they point along (a clash is resolved in favour of the stronger axes) -> row-major 3x3 float32."""
    f = np.float32

    def unit(v):
        return (v / max(f(np.sqrt((v * v).sum(dtype=f))), f(1e-6))).astype(f)

    a = unit(np.asarray(main_axis, f))
    b = np.asarray(secondary_axis, f)
    b = unit(b - (b * a).sum(dtype=f) * a)
    c = unit(np.cross(a, b).astype(f))
    where = [int(np.abs(v).argmax()) for v in (a, b, c)]
    for fix in (2, 1):                      # move the third axis first, then the second (it contributed less to the shape)
        if len(set(where)) == 3:
            break
        where[fix] = ({0, 1, 2} - set(where)).pop()
    if len(set(where)) != 3:
        raise ValueError("Could not find 3 unique axis")
    rows = [None] * 3
    for v, k in zip((a, b, c), where):
        rows[k] = v
    return np.stack(rows, 0).astype(f)


class BoxProjectionUnwrapper:
    """Callable UV unwrapper; keeps its scratch buffers between calls.  `raster_resolution`: pixels per chart side of the
    overlap test's z-buffer."""

    def __init__(self, raster_resolution: int = 1024):
        self.raster_resolution = int(raster_resolution)
        self._buf = {}
        self.last = {}     # per-stage tensors of the last call (chart, assigned, rotation, angles): read by tests / tools

    def _b(self, name, shape, dtype, dev):
        key = (name, tuple(shape), dtype, str(dev))
        t = self._buf.get(key)
        if t is None:
            t = torch.empty(shape, dtype=dtype, device=dev)
            self._buf = {k: v for k, v in self._buf.items() if k[0] != name}
            self._buf[key] = t
        return t

    # ---- stages -------------------------------------------------------------------------------------------------
    def rotation(self, v_pos):
        sums = self._b("sums9", (9,), torch.float64, v_pos.device)
        check(lib.sculpt_uv_moments(_ptr(v_pos), v_pos.shape[0], _ptr(sums), _stream()))
        main, second = principal_axes(sums.cpu().numpy(), v_pos.shape[0])
        return axis_rotation(main, second)

    def box_project(self, v_pos, v_nrm, faces, rot):
        dev, nv, nf = v_pos.device, v_pos.shape[0], faces.shape[0]
        st = self._b("stats", (int(lib.sculpt_uv_stats_words()),), torch.int32, dev)
        rp = self._b("rot_pos", (nv, 3), torch.float32, dev)
        rn = self._b("rot_nrm", (nv, 3), torch.float32, dev)
        uv = self._b("face_uv", (nf, 3, 2), torch.float32, dev)
        chart = self._b("chart", (nf,), torch.int32, dev)
        r9 = (ctypes.c_float * 9)(*[float(x) for x in np.asarray(rot, np.float32).reshape(-1)])
        check(lib.sculpt_uv_box_project(_ptr(v_pos), _ptr(v_nrm), nv, _ptr(faces), int(faces.dtype == torch.int64), nf,
                                        ctypes.cast(r9, ctypes.c_void_p), _ptr(rp), _ptr(rn), _ptr(uv), _ptr(chart), _ptr(st), _stream()))
        return rp, rn, uv, chart, st

    def chart_angles(self, rp, rn, faces, uv, chart):
        dev, nv, nf = rp.device, rp.shape[0], faces.shape[0]
        vt = self._b("vtan", (nv, 4), torch.float32, dev)
        sums = self._b("sums42", (6, 7), torch.float64, dev)
        check(lib.sculpt_uv_chart_tangents(_ptr(rp), _ptr(rn), nv, _ptr(faces), int(faces.dtype == torch.int64), nf, _ptr(uv), _ptr(chart),
                                           _ptr(vt), _ptr(sums), _stream()))
        s = sums.cpu().numpy()
        angles = np.zeros(6, np.float32)
        for c in range(6):
            if s[c, 6] == 0:
                continue
            a = (s[c, 0:3] / s[c, 6]).astype(np.float32)
            e = (s[c, 3:6] / s[c, 6]).astype(np.float32)
            # :350-355: atan2(cross_z, dot) of the two mean tangents (3-D dot, 2-D cross)
            angles[c] = np.float32(math.atan2(float(a[0] * e[1] - a[1] * e[0]), float((a * e).sum(dtype=np.float32))))
        return angles, vt

    def rotate_charts(self, uv, chart, angles, st):
        co = (ctypes.c_float * 6)(*[float(np.float32(math.cos(float(a)))) for a in angles])
        si = (ctypes.c_float * 6)(*[float(np.float32(math.sin(float(a)))) for a in angles])
        check(lib.sculpt_uv_rotate_charts(_ptr(uv), _ptr(chart), uv.shape[0], ctypes.cast(co, ctypes.c_void_p),
                                          ctypes.cast(si, ctypes.c_void_p), _ptr(st), _stream()))
        return uv

    def assign_atlas(self, rp, faces, uv, chart):
        dev, nf, res = rp.device, faces.shape[0], self.raster_resolution
        zbuf = self._b("zbuf", (6 * res * res,), torch.int64, dev)
        assigned = self._b("assigned", (nf,), torch.int32, dev)
        check(lib.sculpt_uv_assign_atlas(_ptr(rp), _ptr(faces), int(faces.dtype == torch.int64), nf, _ptr(uv), _ptr(chart), res,
                                         _ptr(zbuf), _ptr(assigned), _stream()))
        return assigned

    def place(self, uv, assigned, island_padding, st, out=None):
        dev, nf = uv.device, uv.shape[0]
        scratch = self._b("blocks", ((nf + 255) // 256,), torch.int32, dev)
        if out is None:
            out = torch.empty((nf * 3, 2), dtype=torch.float32, device=dev)
        check(lib.sculpt_uv_place(_ptr(uv), _ptr(assigned), nf, float(island_padding), _ptr(st), _ptr(scratch), _ptr(out), _stream()))
        return out

    # ---- Unwrapper.forward ---------------------------------------------------------------------------------------
    def __call__(self, vertex_positions, vertex_normals, triangle_idxs, island_padding, rot=None, assigned=None):
        """`rot` (3x3) / `assigned` ([Nf] int32 on the device) override the two stages that are not pinned to the reference
        (used by the parity tests to feed both sides the same values)."""
        v = vertex_positions
        if v.device.type != "cuda":
            raise _lib.SculptError("BoxProjectionUnwrapper runs on an MI355X only (device %s; there is no CPU fallback)" % v.device)
        v = v.to(torch.float32).contiguous()
        n = vertex_normals.to(torch.float32).contiguous()
        f = triangle_idxs.contiguous()
        if f.dtype not in (torch.int32, torch.int64):
            raise _lib.SculptError("BoxProjectionUnwrapper: faces must be int32 or int64")
        if f.shape[0] == 0:
            return torch.zeros((0, 2), dtype=torch.float32, device=v.device), f.reshape(0, 3)
        rot = self.rotation(v) if rot is None else np.asarray(rot, np.float32)
        rp, rn, uv, chart, st = self.box_project(v, n, f, rot)
        angles, _ = self.chart_angles(rp, rn, f, uv, chart)
        self.rotate_charts(uv, chart, angles, st)
        if assigned is None:
            assigned = self.assign_atlas(rp, f, uv, chart)
        placed = self.place(uv, assigned, island_padding, st)
        self.last = dict(rot=rot, angles=angles, chart=chart, assigned=assigned, face_uv=uv, rot_pos=rp, rot_nrm=rn)
        indices = torch.arange(3 * f.shape[0], device=v.device, dtype=f.dtype).reshape(-1, 3)
        return placed, indices
