"""Mesh comparison used by the image -> mesh parity tests (test infrastructure).

north_star: "vertices within 1e-4 relative of the CPU reference".  Two meshes extracted from scene codes that
differ by fp32 rounding can differ in a handful of cells (a lattice value that sits on the iso level within
rounding), so the check has two unconditional legs:
  * identical face arrays  -> every vertex within `tol` of its counterpart (same index);
  * otherwise              -> vertex counts within `count_slack`, and every vertex of either mesh has a vertex of
                              the other within `tol` (cKDTree nearest neighbour, both directions) -- except at most
                              `count_slack` vertices per direction: a lattice value within rounding of the iso level
                              that is a local extremum gives a tiny closed component that exists in one mesh only.
"""
import numpy as np


def assert_mesh_close(v, f, rv, rf, tol, count_slack=None):
    v, rv = np.asarray(v, np.float64), np.asarray(rv, np.float64)
    f, rf = np.asarray(f), np.asarray(rf)
    assert len(v) > 0 and len(rv) > 0, "empty mesh"
    if count_slack is None:
        count_slack = max(4, len(rv) // 500)
    assert abs(len(v) - len(rv)) <= count_slack, (len(v), len(rv))
    assert abs(len(f) - len(rf)) <= 3 * count_slack, (len(f), len(rf))
    if f.shape == rf.shape and np.array_equal(f, rf):
        d = float(np.abs(v - rv).max())
        assert d < tol, "same topology, max vertex difference %.3e >= %.3e" % (d, tol)
        return {"same_topology": True, "max_dist": d}
    from scipy.spatial import cKDTree

    d_ab, _ = cKDTree(rv).query(v)
    d_ba, _ = cKDTree(v).query(rv)
    far = int((d_ab >= tol).sum()), int((d_ba >= tol).sum())
    assert max(far) <= count_slack, "topology differs; %s vertices have no counterpart within %.3e (slack %d)" % (far, tol, count_slack)
    inl = np.concatenate([d_ab[d_ab < tol], d_ba[d_ba < tol]])
    return {"same_topology": False, "max_dist": float(inl.max()), "unmatched": far}


def mesh_distance(v, rv, extent=1.74):
    """Two-sided nearest-vertex distance between two meshes, as fractions of `extent` (the scene box edge): what the
    bf16-transformer mode is quoted with against the fp32 CPU mesh."""
    from scipy.spatial import cKDTree

    v, rv = np.asarray(v, np.float64), np.asarray(rv, np.float64)
    d_ab, _ = cKDTree(rv).query(v)
    d_ba, _ = cKDTree(v).query(rv)
    d = np.concatenate([d_ab, d_ba]) / extent
    return {"max": float(d.max()), "p999": float(np.quantile(d, 0.999)), "p99": float(np.quantile(d, 0.99)), "mean": float(d.mean()),
            "vertices": [int(len(v)), int(len(rv))]}
