"""Host-side triangle remeshing (csrc/remesh_host.h behind sculpt_mesh_*): the gpytoolbox calls of the reference's
Mesh.triangle_remesh (/root/reference/StableFast/sf3d/models/mesh.py:175-237).

gpytoolbox is not available (not in the reference tree, not installed): PARITY UNPINNED.  These tests check what the
reference relies on from the three calls -- a closed oriented manifold stays one (same genus, no flipped faces), the face /
vertex budget is met, the result stays on the input surface, edge lengths end in the Botsch-Kobbelt band -- on analytic
shapes where the surface is known exactly.  No GPU needed: the operations run on the host.
"""
import collections

import numpy as np
import pytest

from sculptmate_amd import _lib
from sculptmate_amd.sf3d import remesh as rm


# ------------------------------------------------------------------------------------------------------------ shapes
def icosahedron():
    t = (1 + 5 ** 0.5) / 2
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
                  [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10],
                  [8, 6, 7], [9, 8, 1]], np.int32)
    return v / np.linalg.norm(v[0]), f


def icosphere(levels):
    v, f = icosahedron()
    for _ in range(levels):
        v, f = rm.subdivide(v, f, 1)
        v /= np.linalg.norm(v, axis=1, keepdims=True)
    return v, f


def torus(nu, nv, R=1.0, r=0.35):
    u = np.arange(nu) * 2 * np.pi / nu
    w = np.arange(nv) * 2 * np.pi / nv
    U, W = np.meshgrid(u, w, indexing="ij")
    v = np.stack([(R + r * np.cos(W)) * np.cos(U), (R + r * np.cos(W)) * np.sin(U), r * np.sin(W)], -1).reshape(-1, 3)
    idx = lambda i, j: (i % nu) * nv + (j % nv)  # noqa: E731
    f = []
    for i in range(nu):
        for j in range(nv):
            f.append([idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)])
            f.append([idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)])
    return v, np.array(f, np.int32)


def open_sheet(n, jitter=0.0, seed=0):
    """An n x n height field over [0,1]^2 with a boundary."""
    g = np.linspace(0, 1, n)
    X, Y = np.meshgrid(g, g, indexing="ij")
    rng = np.random.default_rng(seed)
    X = X + jitter * rng.uniform(-1, 1, X.shape) / n * (X > 0) * (X < 1)
    Y = Y + jitter * rng.uniform(-1, 1, Y.shape) / n * (Y > 0) * (Y < 1)
    v = np.stack([X, Y, 0.1 * np.sin(3 * X) * np.cos(2 * Y)], -1).reshape(-1, 3)
    f = []
    for i in range(n - 1):
        for j in range(n - 1):
            a, b, c, d = i * n + j, (i + 1) * n + j, (i + 1) * n + j + 1, i * n + j + 1
            f += [[a, b, c], [a, c, d]]
    return v, np.array(f, np.int32)


# -------------------------------------------------------------------------------------------------------- invariants
def topology(v, f):
    """(euler characteristic, #boundary edges); asserts an oriented manifold: every directed edge once, every undirected
    edge in at most two faces, every vertex with a single fan of faces, no unreferenced vertices, no degenerate faces."""
    assert f.min() >= 0 and f.max() < len(v)
    assert (f[:, 0] != f[:, 1]).all() and (f[:, 1] != f[:, 2]).all() and (f[:, 0] != f[:, 2]).all()
    d = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0).astype(np.int64)
    key = d[:, 0] * (len(v) + 1) + d[:, 1]
    assert len(np.unique(key)) == len(key), "a directed edge is used by two faces (flipped or duplicated face)"
    und = np.sort(d, 1)
    uk, cnt = np.unique(und[:, 0] * (len(v) + 1) + und[:, 1], return_counts=True)
    assert cnt.max() <= 2, "non-manifold edge"
    assert len(np.unique(f)) == len(v), "unreferenced vertices in the output"
    # vertex fans: the link of every vertex (edges opposite to it) is one path or one cycle
    link = collections.defaultdict(list)
    for a, b, c in f:
        link[a].append((b, c))
        link[b].append((c, a))
        link[c].append((a, b))
    for x, es in link.items():
        nxt = dict(es)
        assert len(nxt) == len(es), "vertex %d: two faces leave through the same edge" % x
        starts = set(nxt) - set(nxt.values())
        assert len(starts) <= 1, "vertex %d touches the surface in two separate fans" % x
        s = next(iter(starts)) if starts else es[0][0]
        n, cur = 0, s
        while cur in nxt and n < len(es):
            cur = nxt[cur]
            n += 1
        assert n == len(es), "vertex %d: its faces do not form one fan" % x
    return len(v) - len(uk) + len(f), int((cnt == 1).sum())


def edge_lengths(v, f):
    d = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0), 1)
    e = np.unique(d, axis=0)
    return np.linalg.norm(v[e[:, 0]] - v[e[:, 1]], axis=1)


def signed_volume(v, f):
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    return float(np.einsum("ij,ij->i", a, np.cross(b, c)).sum() / 6)


def torus_distance(p, R=1.0, r=0.35):
    return np.abs(np.hypot(np.hypot(p[:, 0], p[:, 1]) - R, p[:, 2]) - r)


# ------------------------------------------------------------------------------------------------------------- tests
def test_subdivide_counts_positions_and_orientation():
    v, f = icosahedron()
    v2, f2 = rm.subdivide(v, f, 2)
    assert len(f2) == 16 * len(f) and len(v2) == 12 + 30 + 120  # V + E per round
    assert np.array_equal(v2[:12], v)  # old vertices first, untouched
    chi, nb = topology(v2, f2)
    assert chi == 2 and nb == 0
    # midpoint subdivision keeps every point on the original faces: volume unchanged, all normals outward
    assert abs(signed_volume(v2, f2) - signed_volume(v, f)) < 1e-12
    v1, f1 = rm.subdivide(v, f, 1)
    # every new vertex is the midpoint of an old edge
    e = np.unique(np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0), 1), axis=0)
    mids = {tuple(np.round(0.5 * (v[a] + v[b]), 12)) for a, b in e}
    assert {tuple(np.round(p, 12)) for p in v1[12:]} == mids
    v0, f0 = rm.subdivide(v, f, 0)
    assert np.array_equal(v0, v) and np.array_equal(f0, f)


@pytest.mark.parametrize("ratio", [0.5, 0.1, 0.02])
def test_decimate_sphere_meets_budget_and_stays_a_sphere(ratio):
    v, f = icosphere(4)  # 2562 vertices, 5120 faces
    vo, fo, _, _ = rm.decimate(v, f, face_ratio=ratio)
    target = int(np.floor(ratio * len(f)))
    assert target - 1 <= len(fo) <= target  # a collapse removes two faces of a closed surface
    chi, nb = topology(vo, fo)
    assert chi == 2 and nb == 0
    assert signed_volume(vo, fo) > 0  # still oriented outward
    # midpoints of chords fall inside the sphere, by at most ~ (edge/2)^2 / 2 per collapse level
    rad = np.linalg.norm(vo, axis=1)
    assert rad.max() <= 1 + 1e-12 and rad.min() > (0.55 if ratio < 0.05 else 0.8)
    # shortest-edge-first keeps the triangles even: no sliver edges
    el = edge_lengths(vo, fo)
    assert el.max() / el.min() < 6


def test_decimate_keeps_genus_and_boundary():
    v, f = torus(48, 24)
    vo, fo, _, _ = rm.decimate(v, f, face_ratio=0.25)
    chi, nb = topology(vo, fo)
    assert chi == 0 and nb == 0 and len(fo) <= len(f) // 4
    assert torus_distance(vo).max() < 0.08
    v, f = open_sheet(33)
    chi0, nb0 = topology(v, f)
    vo, fo, _, _ = rm.decimate(v, f, face_ratio=0.2)
    chi, nb = topology(vo, fo)
    assert chi == chi0 == 1 and 0 < nb < nb0 and len(fo) <= int(0.2 * len(f))
    # midpoint placement pulls the rim inwards a little (libigl's default does not pin the boundary), never outwards,
    # and no triangle of the height field turns over
    assert vo[:, :2].min() >= -1e-12 and vo[:, :2].max() <= 1 + 1e-12
    a, b, c = vo[fo[:, 0]], vo[fo[:, 1]], vo[fo[:, 2]]
    area = 0.5 * ((b - a)[:, 0] * (c - a)[:, 1] - (b - a)[:, 1] * (c - a)[:, 0])
    assert (area > 0).all() and 0.85 < area.sum() <= 1 + 1e-12


def test_decimate_stops_when_nothing_can_collapse():
    v, f = icosahedron()
    f4 = np.array([[0, 1, 2], [0, 3, 1], [1, 3, 2], [2, 3, 0]], np.int32)  # a tetrahedron: no edge passes the link condition
    v4 = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float64)
    vo, fo, _, _ = rm.decimate(v4, f4, num_faces=0)
    assert len(fo) == 4 and len(vo) == 4
    vo, fo, _, _ = rm.decimate(v, f, num_faces=0)
    chi, nb = topology(vo, fo)
    assert chi == 2 and nb == 0 and len(fo) == 4  # ends at a tetrahedron, never at a degenerate pillow


def test_remesh_botsch_sphere_edge_band_valence_and_surface():
    rng = np.random.default_rng(5)
    v, f = icosphere(3)
    # irregular input: a decimated finer sphere has uneven edges and valences
    v, f, _, _ = rm.decimate(*icosphere(5), face_ratio=0.12)
    v = v / np.linalg.norm(v, axis=1, keepdims=True)
    h = float(edge_lengths(v, f).mean())
    vo, fo = rm.remesh_botsch(v, f, 10, None)
    chi, nb = topology(vo, fo)
    assert chi == 2 and nb == 0 and signed_volume(vo, fo) > 0.95 * signed_volume(v, f)
    el = edge_lengths(vo, fo)
    assert abs(el.mean() / h - 1) < 0.15
    assert ((el > 0.8 * h * 0.9) & (el < 4 / 3 * h * 1.1)).mean() > 0.97 and el.min() > 0.45 * h and el.max() < 1.6 * h
    val = np.bincount(fo.ravel())
    assert ((val >= 5) & (val <= 7)).mean() > 0.95 and val.min() >= 4 and val.max() <= 8
    # projected onto the INPUT surface (the faceted sphere, edges up to ~1.5 h): never outside the unit sphere, inside by at
    # most the sagitta of its largest facet, circumradius^2 / 2 ~ (1.5 h)^2 / 6
    rad = np.linalg.norm(vo, axis=1)
    assert rad.max() <= 1 + 1e-9 and rad.min() >= 1 - 0.6 * h * h
    # explicit h: halving it quadruples the face count
    vf, ff = rm.remesh_botsch(v, f, 10, 0.5 * h)
    assert 3.2 < len(ff) / len(fo) < 4.8
    assert abs(edge_lengths(vf, ff).mean() / (0.5 * h) - 1) < 0.15
    del rng


def test_remesh_botsch_torus_and_open_sheet():
    v, f = torus(64, 16)  # anisotropic: long thin triangles
    vo, fo = rm.remesh_botsch(v, f, 10, None)
    chi, nb = topology(vo, fo)
    assert chi == 0 and nb == 0
    assert torus_distance(vo).max() < 0.02
    el = edge_lengths(vo, fo)
    assert el.max() / el.min() < 3.0 and edge_lengths(v, f).max() / edge_lengths(v, f).min() > 2.5
    v, f = open_sheet(25, jitter=0.4)
    chi0, nb0 = topology(v, f)
    bverts = lambda vv, ff: {tuple(np.round(vv[i], 12)) for i in boundary_vertices(ff)}  # noqa: E731
    vo, fo = rm.remesh_botsch(v, f, 10, None)
    chi, nb = topology(vo, fo)
    assert chi == 1
    # boundary vertices are features: none moves or disappears (splits may add new ones on boundary edges)
    assert bverts(v, f) <= bverts(vo, fo)
    assert np.abs(vo[:, 2] - 0.1 * np.sin(3 * vo[:, 0]) * np.cos(2 * vo[:, 1])).max() < 2e-3


def boundary_vertices(f):
    d = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0), 1)
    e, cnt = np.unique(d, axis=0, return_counts=True)
    return np.unique(e[cnt == 1])


def test_remesh_is_deterministic_and_zero_iterations_is_identity():
    v, f = torus(32, 12)
    a = rm.remesh_botsch(v, f, 3, None)
    b = rm.remesh_botsch(v, f, 3, None)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    v0, f0 = rm.remesh_botsch(v, f, 0, None)
    assert np.array_equal(v0, v) and np.array_equal(f0, f)


def test_bad_input_is_refused_with_a_message():
    v, f = icosahedron()
    bad = f.copy()
    bad[3, 1] = 99
    for fn in (lambda: rm.decimate(v, bad, 0.5), lambda: rm.remesh_botsch(v, bad, 1), lambda: rm.subdivide(v, bad, 1)):
        with pytest.raises(_lib.SculptError, match="out of range"):
            fn()
    deg = f.copy()
    deg[0] = [1, 1, 2]
    with pytest.raises(_lib.SculptError, match="degenerate"):
        rm.decimate(v, deg, 0.5)
    nanv = v.copy()
    nanv[2, 0] = np.nan
    with pytest.raises(_lib.SculptError, match="non-finite"):
        rm.remesh_botsch(nanv, f, 1)
    # empty mesh: empty result, no error
    vo, fo = rm.remesh_botsch(np.zeros((0, 3)), np.zeros((0, 3), np.int32), 2)
    assert vo.shape == (0, 3) and fo.shape == (0, 3)


def test_triangle_remesh_follows_the_reference_sequence():
    """Mesh.triangle_remesh (mesh.py:175-237) with a recording toolbox: subdivide only when the budget exceeds the vertex
    count, decimate with face_ratio = budget / vertices, remesh with h = None after a decimation."""
    import torch
    from sculptmate_amd.sf3d.system import Mesh

    calls = []

    class Recorder:
        @staticmethod
        def subdivide(v, f, iters=1):
            calls.append(("subdivide", iters, v.dtype, f.dtype))
            return rm.subdivide(v, f, iters)

        @staticmethod
        def decimate(v, f, face_ratio=0.1):
            calls.append(("decimate", round(float(face_ratio), 6), len(v)))
            return rm.decimate(v, f, face_ratio)

        @staticmethod
        def remesh_botsch(v, f, i, h):
            calls.append(("remesh_botsch", i, h, v.dtype, f.dtype))
            return rm.remesh_botsch(v, f, i, h)

    v, f = icosphere(3)  # 642 vertices
    mesh = Mesh(torch.from_numpy(v).float(), torch.from_numpy(f).long())
    out = rm.triangle_remesh(mesh, vertex_count=200, toolbox=Recorder)
    assert [c[0] for c in calls] == ["decimate", "remesh_botsch"]
    assert calls[0][1] == round(200 / 642, 6) and calls[1][1:3] == (10, None) and calls[1][3] == np.float64 and calls[1][4] == np.int32
    assert out.v_pos.dtype == torch.float32 and out.t_pos_idx.dtype == torch.int64
    assert 140 < out.v_pos.shape[0] < 300  # remeshing at the decimated mesh's edge length keeps the budget roughly
    del calls[:]
    out = rm.triangle_remesh(mesh, vertex_count=2000, toolbox=Recorder)  # more than it has: subdivide first
    assert [c[0] for c in calls] == ["subdivide", "decimate", "remesh_botsch"] and calls[0][1] == 2  # ceil(log2(2000 / 642))
    assert calls[0][2] == np.float32 and calls[0][3] == np.int32
    assert calls[1][2] == 642 + 1920 + 7680 and calls[1][1] == round(2000 / 10242, 6)
    del calls[:]
    out = rm.triangle_remesh(mesh, vertex_count=-1, edge_length_multiplier=2.0, toolbox=Recorder)
    mean = float(edge_lengths(v.astype(np.float32).astype(np.float64), f).astype(np.float32).mean())
    assert [c[0] for c in calls] == ["remesh_botsch"] and abs(calls[0][2] - 2.0 * mean) < 1e-6
    chi, nb = topology(out.v_pos.numpy().astype(np.float64), out.t_pos_idx.numpy())
    assert chi == 2 and nb == 0
    with pytest.raises(NotImplementedError):
        rm.native_remesher(mesh, "quad", 100)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_isosurface_meshes_with_borders_keep_their_topology(seed):
    """Marching-cubes surfaces of a random smooth field (several components, handles, open rims where the surface leaves the
    grid) through the whole triangle_remesh sequence: manifold in -> manifold out, Euler characteristic and the number of
    boundary loops' worth of rim unchanged in kind (closed stays closed, open stays open), every vertex near the input."""
    from scipy.spatial import cKDTree

    from oracle import capi

    rng = np.random.default_rng(seed)
    n = 28
    g = np.linspace(-1, 1, n)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    vol = np.zeros((n, n, n))
    for _ in range(6):
        k = rng.uniform(1.0, 3.5, 3)
        ph = rng.uniform(0, 2 * np.pi, 3)
        vol += rng.uniform(0.5, 1.0) * np.sin(k[0] * X + ph[0]) * np.sin(k[1] * Y + ph[1]) * np.sin(k[2] * Z + ph[2])
    v, f = capi.marching_cubes(vol.astype(np.float32), 0.15)[:2]
    v, f = v.astype(np.float64), f.astype(np.int32)
    chi0, nb0 = topology(v, f)
    vd, fd, _, _ = rm.decimate(v, f, face_ratio=0.4)
    chi1, nb1 = topology(vd, fd)
    assert chi1 == chi0 and (nb1 > 0) == (nb0 > 0) and len(fd) <= int(0.4 * len(f)) + 1
    vr, fr = rm.remesh_botsch(vd, fd, 10, None)
    chi2, nb2 = topology(vr, fr)
    assert chi2 == chi0 and (nb2 > 0) == (nb0 > 0)
    # on the decimated surface; the decimated surface within a cell or so of the isosurface
    h = edge_lengths(vd, fd).mean()
    tree = cKDTree(vd)
    assert tree.query(vr)[0].max() < 1.5 * h
    assert cKDTree(v).query(vr)[0].max() < 2.5 * h
    el = edge_lengths(vr, fr)
    assert ((el > 0.7 * h) & (el < 1.45 * h)).mean() > 0.9


def test_small_components_never_disappear():
    """An isolated triangle, a two-triangle quad and a fan tip next to a closed body: decimation towards zero faces keeps
    every component (libigl's boundary-at-infinity link condition refuses to collapse the last triangle of a patch)."""
    vs, fs = icosahedron()
    tri_v = np.array([[5, 0, 0], [6, 0, 0], [5, 1, 0]], np.float64)
    quad_v = np.array([[8, 0, 0], [9, 0, 0], [9, 1, 0], [8, 1, 0]], np.float64)
    v = np.concatenate([vs, tri_v, quad_v], 0)
    f = np.concatenate([fs, [[12, 13, 14]], [[15, 16, 17], [15, 17, 18]]], 0).astype(np.int32)
    chi0, nb0 = topology(v, f)
    assert chi0 == 2 + 1 + 1 and nb0 == 3 + 4
    vo, fo, _, _ = rm.decimate(v, f, num_faces=0)
    chi, nb = topology(vo, fo)
    assert chi == chi0 and nb > 0
    assert len(fo) == 4 + 1 + 1  # tetrahedron + one triangle + one triangle (the quad loses its diagonal, never its last face)
    vr, fr = rm.remesh_botsch(v, f, 5, 3.0)  # a target far above every edge: collapses wherever they are allowed
    assert topology(vr, fr)[0] == chi0


@pytest.mark.parametrize("seed", range(1000, 1012))
def test_random_isosurfaces_stress(seed):
    """The bug class the three cases above came from, kept as a sweep: marching-cubes surfaces with many small components and
    open rims through decimation (mild, strong, to exhaustion) and remeshing (default, fine, coarse target): the Euler
    characteristic never changes and open stays open."""
    from oracle import capi

    rng = np.random.default_rng(seed)
    n = int(rng.integers(14, 24))
    g = np.linspace(-1, 1, n)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    vol = np.zeros((n, n, n))
    for _ in range(int(rng.integers(3, 9))):
        k = rng.uniform(0.8, 5.0, 3)
        ph = rng.uniform(0, 2 * np.pi, 3)
        vol += rng.uniform(0.3, 1.0) * np.sin(k[0] * X + ph[0]) * np.sin(k[1] * Y + ph[1]) * np.sin(k[2] * Z + ph[2])
    v, f = capi.marching_cubes(vol.astype(np.float32), float(rng.uniform(-0.3, 0.3)))[:2]
    v, f = v.astype(np.float64), f.astype(np.int32)
    chi0, nb0 = topology(v, f)
    for ratio in (0.7, 0.08, 0.0):
        vd, fd, _, _ = rm.decimate(v, f, face_ratio=ratio)
        chi, nb = topology(vd, fd)
        assert chi == chi0 and (nb > 0) == (nb0 > 0), (ratio, chi0, chi)
        if ratio == 0.0:
            continue
        h = edge_lengths(vd, fd).mean()
        for target in (None, 0.5 * h, 2.5 * h):
            vr, fr = rm.remesh_botsch(vd, fd, 4, target)
            chi, nb = topology(vr, fr)
            assert chi == chi0 and (nb > 0) == (nb0 > 0), (ratio, target, chi0, chi)
