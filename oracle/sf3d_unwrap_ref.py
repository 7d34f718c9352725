"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by sculptmate_amd/).

NumPy (float32) restatement of StableFast-3D's box-projection UV unwrapper, stage by stage
(StableFast/sf3d/uv_unwrapper/unwrap.py); each function cites the lines it follows:

  axis_rotation            :546-623  _align_mesh_with_main_axis, given the two principal axes (the reference gets them
                                     from a RANDOMISED torch.pca_lowrank under manual_seed(0): the axes themselves are
                                     compared up to sign, not bit for bit)
  box_project              :16-122   _box_assign_vertex_to_cube_face
  vertex_tangents          :239-305  _calculate_tangents
  rotate_charts            :307-381  _rotate_uv_slices_consistent_space
  slice_offset_and_scale   :177-237  _find_slice_offset_and_scale
  place_in_atlas           :383-527  _handle_slice_uvs, _handle_remaining_uvs, _distribute_individual_uvs_in_atlas
  unwrap                   :625-697  forward, with the atlas assignment as an argument

NOT restated: assign_faces_uv_to_atlas_index (:124-175) lives in uv_unwrapper.dll, whose source is not in /root/reference
-> `assigned` is an input here; the product's own overlap resolution (csrc/uv_unwrap.hip) is checked by properties
(no two front-layer triangles overlap), "parity unpinned" against the DLL.

PARITY PIN: tests/golden/sf3d_unwrap.npz, every stage produced by the reference's own methods
(tests/golden/make_sf3d_unwrap_goldens.py); tests/test_oracle_sf3d_unwrap.py.
"""
import math

import numpy as np

F32 = np.float32
AXES = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], F32)


def _normalize(x, eps, axis=-1):
    n = np.sqrt((x * x).sum(axis, keepdims=True, dtype=F32)).astype(F32)
    return (x / np.maximum(n, F32(eps))).astype(F32)


def axis_rotation(main_axis, secondary_axis):
    """Two principal axes -> the 3x3 matrix whose rows are the axes sorted onto the canonical axes they point along."""
    a = _normalize(np.asarray(main_axis, F32), 1e-6)
    b = np.asarray(secondary_axis, F32)
    b = _normalize(b - (b * a).sum(dtype=F32) * a, 1e-6)
    c = _normalize(np.cross(a, b).astype(F32), 1e-6)
    ia, ib, ic = int(np.abs(a).argmax()), int(np.abs(b).argmax()), int(np.abs(c).argmax())
    step = 1
    while len({ia, ib, ic}) != 3:
        missing = ({0, 1, 2} - {ia, ib, ic}).pop()
        if step == 1:
            ic = missing
        elif step == 2:
            ib = missing
        else:
            raise ValueError("Could not find 3 unique axis")
        step += 1
    rows = [None] * 3
    rows[ia], rows[ib], rows[ic] = a, b, c
    return np.stack(rows, 1).T.astype(F32)


def box_project(pos, nrm, faces, bbox):
    """-> (uv [Nf,3,2], chart [Nf]): chart = cube face the summed corner normals point at; uv = the two other coordinates."""
    p = (pos - bbox[:1]) / (bbox[1:] - bbox[:1])
    p = (F32(2.0) * p - F32(1.0)).astype(F32)
    tri = p[faces]                                             # [Nf, 3, 3]
    fn = _normalize(nrm[faces].sum(1, dtype=F32), 1e-6)
    chart = (fn[:, None, :] * AXES[None]).sum(-1, dtype=F32).argmax(-1)
    ax = chart // 2
    absax = np.abs(np.take_along_axis(tri, ax[:, None, None].repeat(3, 1), 2)[..., 0])      # [Nf, 3]
    u_src = np.where(ax == 0, 1, 0)                             # x-charts take y, the others x
    v_src = np.where(ax == 2, 1, 2)                             # z-charts take y, the others z
    uc = np.take_along_axis(tri, u_src[:, None, None].repeat(3, 1), 2)[..., 0]
    vc = np.take_along_axis(tri, v_src[:, None, None].repeat(3, 1), 2)[..., 0]
    vc = np.where((chart == 4)[:, None], vc, -vc)               # only +z keeps the sign
    div = absax.max(0, keepdims=True)                           # per CORNER SLOT, over all faces (:114)
    uc = np.clip((uc / div + F32(1.0)) * F32(0.5), 0, 1)
    vc = np.clip((vc / div + F32(1.0)) * F32(0.5), 0, 1)
    return np.stack([uc, vc], -1).astype(F32), chart.astype(np.int64)


def vertex_tangents(pos, nrm, faces, uv):
    """Per-vertex tangent: mean of the incident faces' UV tangents, made perpendicular to the normal."""
    p = [pos[faces[:, i]] for i in range(3)]
    duv1, duv2 = uv[:, 1] - uv[:, 0], uv[:, 2] - uv[:, 0]
    dp1, dp2 = p[1] - p[0], p[2] - p[0]
    nom = dp1 * duv2[:, 1:2] - dp2 * duv1[:, 1:2]
    den = duv1[:, 0:1] * duv2[:, 1:2] - duv1[:, 1:2] * duv2[:, 0:1]
    tang = (nom / np.maximum(den, F32(1e-6))).astype(F32)       # clip(1e-6): negative denominators are replaced too
    acc = np.zeros_like(nrm, dtype=np.float64)
    cnt = np.zeros((nrm.shape[0], 1), np.float64)
    for i in range(3):
        np.add.at(acc, faces[:, i], tang.astype(np.float64))
        np.add.at(cnt, faces[:, i], 1.0)
    with np.errstate(invalid="ignore", divide="ignore"):
        t = (acc / cnt).astype(F32)
    t = _normalize(t, 1e-12)
    return _normalize(t - (t * nrm).sum(-1, keepdims=True, dtype=F32) * nrm, 1e-12)


def expected_tangents(pos, nrm):
    """:326-341.  NB the reference writes F.normalize(x, -1): the second positional argument of F.normalize is the norm's
    ORDER p, not the axis -- so this divides by the p = -1 "norm" 1 / (1/|x| + 1/|y| + 1/|z|) along dim 1 (clamped at
    1e-12).  The direction is that of n x (side x n), but the per-vertex lengths (>= 3) weight the chart means below."""
    side = np.stack([-pos[:, 1], pos[:, 0], np.zeros_like(pos[:, 0])], -1)
    t = np.cross(nrm, np.cross(side, nrm)).astype(F32)
    with np.errstate(divide="ignore"):
        inv = (F32(1.0) / np.abs(t)).sum(-1, keepdims=True, dtype=F32)
        norm = (F32(1.0) / inv).astype(F32)
    return (t / np.maximum(norm, F32(1e-12))).astype(F32)


def rotate_charts(pos, nrm, faces, uv, chart):
    """Rotate every chart so its mean tangent points along the expected one, then stretch it to [0, 1] (joint min / max)."""
    uv = uv.copy()
    actual = vertex_tangents(pos, nrm, faces, uv)[faces]
    expect = expected_tangents(pos, nrm)[faces]
    angles = np.zeros(6, F32)
    for c in range(6):
        m = (chart % 6) == c
        if not m.any():
            continue
        a = actual[m].mean((0, 1), dtype=F32)
        e = expect[m].mean((0, 1), dtype=F32)
        ang = F32(math.atan2(float(a[0] * e[1] - a[1] * e[0]), float((a * e).sum(dtype=F32))))
        angles[c] = ang
        co, si = F32(math.cos(ang)), F32(math.sin(ang))
        rot = np.array([[co, -si], [si, co]], F32)
        cur = uv[m] * F32(2) - F32(1)
        r = np.einsum("ij,nfj->nfi", rot, cur).astype(F32)
        uv[m] = (r - r.min()) / (r.max() - r.min())
    return uv, angles


def slice_offset_and_scale(assigned):
    third, sixth = 1 / 3, 1 / 6
    xs, ys = [0, 1, 2, 0, 1, 2], [0, 0, 0, 1, 1, 1]
    ox = np.zeros(assigned.shape, F32)
    oy = np.zeros(assigned.shape, F32)
    for i in range(int(assigned.max()) + 1):
        m = assigned == i
        if not m.any():
            continue
        lvl = i // 6
        ox[m] = third * xs[i % 6] if lvl == 0 else sixth * xs[i % 6] + min(lvl - 1, 1) * 0.5
        oy[m] = third * ys[i % 6] if lvl == 0 else sixth * ys[i % 6] + third * 2
    dx = np.full(assigned.shape, 3, F32)
    dx[assigned >= 6] = 6
    dy = dx.copy()
    dx[assigned >= 12] = 2
    dy[assigned >= 12] = 3
    return ox, oy, dx, dy


def place_in_atlas(uv, assigned, pad):
    """uv [Nf,3,2] chart coordinates in [0,1] -> atlas coordinates [3*Nf, 2]."""
    pad = float(pad)
    uc, vc = uv[..., 0].copy(), uv[..., 1].copy()
    for i in range(6, 12):                                      # overlap slices fill their patch, at most 2x magnified
        m = assigned == i
        if m.sum() > 0:
            uc[m] = (uc[m] - uc[m].min()) / max(uc[m].max() - uc[m].min(), F32(0.5))
            vc[m] = (vc[m] - vc[m].min()) / max(vc[m].max() - vc[m].min(), F32(0.5))
    uc = np.clip(uc * F32(1 - 2 * pad) + F32(pad), 0, 1).astype(F32)
    vc = np.clip(vc * F32(1 - 2 * pad) + F32(pad), 0, 1).astype(F32)
    rem = assigned >= 12
    left = int(rem.sum())
    if left:
        nw = int(math.ceil(0.5 * math.sqrt(left / (0.5 * (1 / 3)))))
        nh = int(math.ceil(left / nw))
        w, h = 1 / nw, 1 / nh
        lim = F32(min(w, h) * 1.5)
        ru, rv = uc[rem], vc[rem]
        ru = (ru - ru.min(1, keepdims=True)) / np.maximum(ru.max(1, keepdims=True) - ru.min(1, keepdims=True), lim)
        rv = (rv - rv.min(1, keepdims=True)) / np.maximum(rv.max(1, keepdims=True) - rv.min(1, keepdims=True), lim)
        ru = np.clip(ru * F32(1 - pad * nw * 0.5) + F32(pad * nw * 0.25), 0, 1)
        rv = np.clip(rv * F32(1 - pad * nh * 0.5) + F32(pad * nh * 0.25), 0, 1)
        k = np.arange(left, dtype=np.int32)
        ru = ru * F32(w) + (k % nw)[:, None] * F32(w)
        rv = rv * F32(h) + (k // nw)[:, None] * F32(h)
        uc[rem] = np.clip(ru * F32(1 - 2 * pad * 0.5) + F32(pad * 0.5), 0, 1)
        vc[rem] = np.clip(rv * F32(1 - 2 * pad * 0.5) + F32(pad * 0.5), 0, 1)
    ox, oy, dx, dy = slice_offset_and_scale(assigned)
    uc = uc / dx[:, None] + ox[:, None]
    vc = vc / dy[:, None] + oy[:, None]
    return np.stack([uc, vc], -1).reshape(-1, 2).astype(F32)


def unwrap(pos, nrm, faces, rot, assign_fn, pad):
    """Unwrapper.forward with the rotation and the atlas assignment supplied -> (placed uv [3Nf,2], chart, assigned)."""
    rp = np.einsum("ij,nj->ni", rot, pos).astype(F32)
    rn = np.einsum("ij,nj->ni", rot, nrm).astype(F32)
    bbox = np.stack([rp.min(0), rp.max(0)], 0)
    uv, chart = box_project(rp, rn, faces, bbox)
    uv, _ = rotate_charts(rp, rn, faces, uv, chart)
    assigned = assign_fn(rp, faces, uv, chart)
    return place_in_atlas(uv, assigned, pad), chart, assigned


# ----------------------------------------------------------------------------- property checks for any assignment
def _tri_overlap_area_samples(a, b, n=6):
    """True if an interior sample point of triangle a (barycentric lattice) lies strictly inside triangle b."""
    w = []
    for i in range(1, n):
        for j in range(1, n - i):
            w.append((i / n, j / n, 1 - i / n - j / n))
    w = np.array(w)
    pts = w @ a
    d = (b[1, 0] - b[0, 0]) * (b[2, 1] - b[0, 1]) - (b[2, 0] - b[0, 0]) * (b[1, 1] - b[0, 1])
    if abs(d) < 1e-14:
        return False
    l1 = ((pts[:, 0] - b[0, 0]) * (b[2, 1] - b[0, 1]) - (b[2, 0] - b[0, 0]) * (pts[:, 1] - b[0, 1])) / d
    l2 = ((b[1, 0] - b[0, 0]) * (pts[:, 1] - b[0, 1]) - (pts[:, 0] - b[0, 0]) * (b[1, 1] - b[0, 1])) / d
    return bool(np.any((l1 > 1e-3) & (l2 > 1e-3) & (l1 + l2 < 1 - 1e-3)))


def overlapping_pairs(uv, group):
    """Pairs (i, j) of triangles of the same group whose UV interiors overlap (O(n^2) per group: small meshes only)."""
    out = []
    for gval in np.unique(group):
        ids = np.nonzero(group == gval)[0]
        lo, hi = uv[ids].min(1), uv[ids].max(1)
        for x, i in enumerate(ids):
            cand = np.nonzero((lo[:, 0] < hi[x, 0]) & (hi[:, 0] > lo[x, 0]) & (lo[:, 1] < hi[x, 1]) & (hi[:, 1] > lo[x, 1]))[0]
            for y in cand:
                j = ids[y]
                if j > i and (_tri_overlap_area_samples(uv[i], uv[j]) or _tri_overlap_area_samples(uv[j], uv[i])):
                    out.append((int(i), int(j)))
    return out
