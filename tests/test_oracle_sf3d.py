"""The SF3D oracle (oracle/sf3d_ref.py) pinned to outputs of the reference's own classes (tests/golden/sf3d_*.npz,
made by tests/golden/make_sf3d_goldens.py in the build container).  CPU only."""
import os

import numpy as np
import torch

from oracle import sf3d_ref as R

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(G, name))
    return {k: z[k] for k in z.files}


def weights(z, prefix):
    return {prefix + k[2:]: v for k, v in z.items() if k.startswith("w.")}


def test_camera_embedding_matches_reference():
    z = load("sf3d_camera.npz")
    K, Kn = R.intrinsic_from_fov_deg(40.0, 512, 512)
    assert np.array_equal(K.numpy(), z["intrinsic"]) and np.array_equal(Kn.numpy(), z["intrinsic_normed"])
    assert np.array_equal(R.default_cond_c2w(1.6).numpy(), z["c2w"])
    e = R.camera_embedding(weights(z, "camera_embedder."), "camera_embedder.")
    np.testing.assert_allclose(e.numpy(), z["embedding"].reshape(-1), rtol=0, atol=1e-6)


def test_dinov2_with_modulation_matches_reference():
    z = load("sf3d_dino.npz")
    cfg = dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=2, patch_size=14, layer_norm_eps=1e-6)
    img = torch.from_numpy(z["images"][0, 0]).permute(1, 2, 0)
    with torch.no_grad():
        out = R.dino_forward(weights(z, "image_tokenizer."), img, z["cond"].reshape(-1), cfg)
    ref = z["out"][0, 0].T  # [Nt, Ct]
    assert out.shape == ref.shape == (17, 64)
    np.testing.assert_allclose(out.numpy(), ref, rtol=0, atol=2e-5)


def test_two_stream_backbone_matches_reference():
    z = load("sf3d_backbone.npz")
    cfg = dict(num_attention_heads=2, num_blocks=2, num_basic_blocks=2, norm_num_groups=32)
    with torch.no_grad():
        out = R.backbone_forward(weights(z, "backbone."), z["tokens"][0], z["image_tokens"][0], cfg)
    np.testing.assert_allclose(out.numpy(), z["out"][0], rtol=0, atol=3e-5)


def test_pixel_shuffle_upsampler_matches_reference():
    z = load("sf3d_post.npz")
    with torch.no_grad():
        out = R.post_forward(weights(z, "post_processor."), z["x"][0], dict(conv_layers=4, scale_factor=2))
    assert out.shape == (3, 40, 10, 10)
    np.testing.assert_allclose(out.numpy(), z["out"][0], rtol=0, atol=2e-5)


def test_query_and_material_heads_match_reference():
    z = load("sf3d_decoder.npz")
    with torch.no_grad():
        feats = R.query_triplane(z["points"], z["planes"], 0.87)
        np.testing.assert_allclose(feats.numpy(), z["feats"], rtol=0, atol=1e-6)
        dec = R.decoder_forward(weights(z, "decoder."), feats)
    assert set(dec) == {"density", "features", "perturb_normal", "vertex_offset"}
    for k, v in dec.items():
        np.testing.assert_allclose(v.numpy(), z["out." + k], rtol=2e-5, atol=2e-6, err_msg=k)
    inc = R.decoder_forward(weights(z, "decoder."), feats, include=["density"])
    assert list(inc) == ["density"]


def test_marching_tetrahedra_matches_reference_bit_for_bit():
    z = load("sf3d_mtet.npz")
    res = int(z["res"])
    grid = R.deform_grid(z["vertices"], z["deform"], res)
    np.testing.assert_allclose(grid.numpy(), z["grid_vertices"], rtol=0, atol=1e-7)
    # topology and interpolation on the reference's own deformed grid: exact
    v, f = R.marching_tets(z["grid_vertices"], z["sdf"], z["indices"])
    assert np.array_equal(f, z["faces"])
    assert np.array_equal(v.view(np.uint32), z["v_pos"].view(np.uint32))
    v, f = R.marching_tets(z["vertices"], z["sdf"], z["indices"])
    assert np.array_equal(f, z["faces_nodef"]) and np.array_equal(v.view(np.uint32), z["v_pos_nodef"].view(np.uint32))
    assert np.array_equal(R.all_edges(z["indices"]), z["tet_edges"])


def test_marching_tetrahedra_mesh_is_closed():
    z = load("sf3d_mtet.npz")
    f = z["faces"]
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    und = np.sort(e, 1)
    _, cnt = np.unique(und, axis=0, return_counts=True)
    assert (cnt == 2).all()  # closed 2-manifold: the surface stays inside the grid
    d, dc = np.unique(e, axis=0, return_counts=True)
    assert (dc == 1).all()  # consistently oriented
