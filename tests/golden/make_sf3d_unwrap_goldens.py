"""Golden vectors for StableFast-3D's box-projection UV unwrapper, produced by IMPORTING the reference
(build container only):  python tests/golden/make_sf3d_unwrap_goldens.py

  sf3d_unwrap.npz   every stage of Unwrapper.forward (StableFast/sf3d/uv_unwrapper/unwrap.py:625-697) that is Python in the
                    reference, each run through the reference's own method on the SAME inputs:
                      _align_mesh_with_main_axis           :546-623  (randomised torch.pca_lowrank under manual_seed(0))
                      _box_assign_vertex_to_cube_face      :16-122
                      _calculate_tangents                  :239-305
                      _rotate_uv_slices_consistent_space   :307-381
                      _find_slice_offset_and_scale         :177-237
                      _distribute_individual_uvs_in_atlas  :383-527 (_handle_slice_uvs, _handle_remaining_uvs)
                    The one stage that is NOT Python -- assign_faces_uv_to_atlas_index, inside uv_unwrapper.dll (:124-175)
                    -- cannot run here: the fixture feeds the later stages a hand-made atlas assignment that exercises
                    all three classes (first slice 0..5, overlap slice 6..11, remaining 12).
Two meshes: an ellipsoid (convex, no overlaps) and a torus (every chart has a hidden back layer).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _reference_shims as shims  # noqa: E402

shims.install()
import make_reference_goldens as mrg  # noqa: E402

mrg._sf3d_shims()


def ellipsoid(n=300, seed=0):
    from scipy.spatial import ConvexHull

    rng = np.random.default_rng(seed)
    p = rng.standard_normal((n, 3))
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    f = ConvexHull(p).simplices.copy()
    c = p[f].mean(1)
    nrm = np.cross(p[f[:, 1]] - p[f[:, 0]], p[f[:, 2]] - p[f[:, 0]])
    flip = (nrm * c).sum(1) < 0
    f[flip] = f[flip][:, [0, 2, 1]]
    scale = np.array([1.0, 0.7, 0.5])
    v = p * scale
    vn = p / scale
    vn /= np.linalg.norm(vn, axis=1, keepdims=True)
    rot = np.linalg.qr(rng.standard_normal((3, 3)))[0]            # not axis aligned: the PCA step has work to do
    return (v @ rot.T).astype(np.float32), (vn @ rot.T).astype(np.float32), f.astype(np.int64)


def torus(nu=28, nv=14, R=1.0, r=0.35):
    u, v = np.meshgrid(np.arange(nu) * 2 * np.pi / nu, np.arange(nv) * 2 * np.pi / nv, indexing="ij")
    p = np.stack([(R + r * np.cos(v)) * np.cos(u), (R + r * np.cos(v)) * np.sin(u) * 0.8, r * np.sin(v)], -1).reshape(-1, 3)
    n = np.stack([np.cos(v) * np.cos(u), np.cos(v) * np.sin(u), np.sin(v)], -1).reshape(-1, 3)
    idx = lambda i, j: (i % nu) * nv + (j % nv)  # noqa: E731
    f = []
    for i in range(nu):
        for j in range(nv):
            f.append([idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)])
            f.append([idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)])
    return p.astype(np.float32), n.astype(np.float32), np.array(f, np.int64)


def main():
    from sf3d.uv_unwrapper.unwrap import Unwrapper

    u = Unwrapper()
    out = {}
    for name, (v, vn, f) in (("ell", ellipsoid()), ("tor", torus())):
        vp, vnn, fi = torch.from_numpy(v), torch.from_numpy(vn), torch.from_numpy(f)
        rp, rn = u._align_mesh_with_main_axis(vp.clone(), vnn.clone())
        bbox = torch.stack([rp.min(0).values, rp.max(0).values], 0)
        uv0, idx = u._box_assign_vertex_to_cube_face(rp, rn, fi, bbox)
        tang = u._calculate_tangents(rp, rn, fi, uv0)
        uv1 = u._rotate_uv_slices_consistent_space(rp, rn, fi, uv0.clone(), idx)
        # a hand-made assignment (the DLL's job): every 7th face to the overlap slice, every 19th to "remaining"
        k = torch.arange(idx.shape[0])
        assigned = idx.clone()
        assigned[k % 7 == 3] += 6
        assigned[k % 19 == 5] = 12
        ox, oy, dx, dy = u._find_slice_offset_and_scale(assigned)
        placed = u._distribute_individual_uvs_in_atlas(uv1.clone(), assigned, ox, oy, dx, dy, 0.02)
        out.update({name + ".v_pos": v, name + ".v_nrm": vn, name + ".faces": f, name + ".rot_pos": rp.numpy(),
                    name + ".rot_nrm": rn.numpy(), name + ".uv_box": uv0.numpy(), name + ".face_index": idx.numpy(),
                    name + ".tangents": tang.numpy(), name + ".uv_rot": uv1.numpy(), name + ".assigned": assigned.numpy(),
                    name + ".offset_x": ox.numpy(), name + ".offset_y": oy.numpy(), name + ".div_x": dx.numpy(),
                    name + ".div_y": dy.numpy(), name + ".placed": placed.numpy()})
        print(name, "faces", f.shape[0], "charts", np.bincount(idx.numpy(), minlength=6), "placed", tuple(placed.shape))
    np.savez_compressed(os.path.join(HERE, "sf3d_unwrap.npz"), **out)


if __name__ == "__main__":
    main()
