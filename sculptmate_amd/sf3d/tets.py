"""Tetrahedral background grid for marching tetrahedra.

The reference loads `load/tets/{resolution}_tets.npz` (StableFast/sf3d/system.py:125-134, keys "vertices" [Nv,3] in
[0,1] and "indices" [Nt,4]; models/isosurface.py:74-85) -- a file its checkout does not ship.  `load_tets` reads
that file when it exists; `kuhn_tet_grid` builds a stand-in of the same format: the (res+1)^3 lattice with every cube
cut into six positively oriented tetrahedra around its main diagonal (Kuhn / Freudenthal triangulation, which is
face-compatible between neighbouring cubes, so the extracted surface is a closed manifold).

Also derives, once per grid, the static topology tables the GPU kernel needs instead of a per-call sort/unique:
the lexicographically sorted unique edge list (== the reference's `all_edges`, isosurface.py:117-131) and each
tetrahedron's six edge ids in `base_tet_edges` order (isosurface.py:64-69).
"""
import os

import numpy as np

BASE_TET_EDGES = (0, 1, 0, 2, 0, 3, 1, 2, 1, 3, 2, 3)


def kuhn_tet_grid(res: int):
    n = res + 1
    ax = np.linspace(0.0, 1.0, n, dtype=np.float64)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    verts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1).astype(np.float32)
    I, J, K = np.meshgrid(np.arange(res), np.arange(res), np.arange(res), indexing="ij")
    base = [I.ravel(), J.ravel(), K.ravel()]
    tets = []
    for perm, odd in (((0, 1, 2), 0), ((0, 2, 1), 1), ((1, 0, 2), 1), ((1, 2, 0), 0), ((2, 0, 1), 0), ((2, 1, 0), 1)):
        cur = [b.copy() for b in base]
        path = [(cur[0] * n + cur[1]) * n + cur[2]]
        for axis in perm:
            cur[axis] = cur[axis] + 1
            path.append((cur[0] * n + cur[1]) * n + cur[2])
        if odd:  # odd permutations walk a mirrored simplex: swap two corners so every tet has positive volume
            path[2], path[3] = path[3], path[2]
        tets.append(np.stack(path, 1))
    idx = np.stack(tets, 1).reshape(-1, 4).astype(np.int64)
    return verts, idx


def load_tets(resolution: int, tets_path=None):
    """(vertices f32 [Nv,3], indices i64 [Nt,4], source) -- the reference's file if present, else the Kuhn grid."""
    if tets_path is not None and os.path.isfile(tets_path):
        z = np.load(tets_path)
        return z["vertices"].astype(np.float32), z["indices"].astype(np.int64), tets_path
    v, i = kuhn_tet_grid(resolution)
    return v, i, "kuhn(%d)" % resolution


def edge_tables(indices: np.ndarray, n_vertices: int):
    """-> (edges i32 [Ne,2] sorted unique with a<b, tet_edges i32 [Nt,6]) for the GPU kernel."""
    e = indices[:, list(BASE_TET_EDGES)].reshape(-1, 2)
    lo, hi = np.minimum(e[:, 0], e[:, 1]), np.maximum(e[:, 0], e[:, 1])
    key = lo.astype(np.int64) * n_vertices + hi
    uniq, inv = np.unique(key, return_inverse=True)
    edges = np.stack([uniq // n_vertices, uniq % n_vertices], 1).astype(np.int32)
    return edges, inv.reshape(-1, 6).astype(np.int32)
