"""oracle/sf3d_est_ref.py against the golden vectors produced by the reference's own estimator classes
(tests/golden/make_sf3d_est_goldens.py) and, for the CLIP tower, by transformers' CLIP implementation."""
import os

import numpy as np
import torch

from conftest import golden_weights
from oracle import sf3d_est_ref as E

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

GLOBAL_CFG = dict(triplane_features=8, n_layers=2, hidden_features=16, activation="relu", pool="max",
                  heads=[dict(name="sg_amplitudes", out_channels=6, n_hidden_layers=3, output_activation="softplus",
                              output_bias=1.0, add_to_decoder_features=False, shape=[-1, 6, 1]),
                         dict(name="tint", out_channels=3, n_hidden_layers=1, output_activation="sigmoid",
                              output_bias=0.0, add_to_decoder_features=True, shape=None)])
_HEAD = dict(out_channels=1, n_hidden_layers=3, output_activation="linear", add_to_decoder_features=True, output_bias=1.0,
             shape=[-1, 1, 1])
IMAGE_CFG = dict(distribution="beta", distribution_eval="mode", activation="relu", hidden_features=128,
                 heads=[dict(name="roughness", **_HEAD), dict(name="metallic", **_HEAD)])


def _load(name):
    z = np.load(os.path.join(G, name))
    return z, golden_weights(z)


def test_global_estimator_matches_reference():
    z, sd = _load("sf3d_global_est.npz")
    out = E.global_estimator_forward(sd, "", GLOBAL_CFG, z["triplane"])
    assert set(out) == {"sg_amplitudes", "decoder_tint"}
    for k, v in out.items():
        ref = z["out." + k]
        assert v.shape == ref.shape
        assert np.abs(v.numpy() - ref).max() < 2e-6, k


def test_image_estimator_resize_heads_and_beta_mode_match_reference():
    z, sd = _load("sf3d_image_est.npz")
    cond = (z["rgb_u8"].astype(np.float32) / 255.0) * z["mask"][..., None].astype(np.float32)
    resized = E.resize_for_clip(cond)
    assert np.abs(resized.numpy() - z["resized"]).max() < 1e-6
    assert np.abs(E.clip_normalize(resized).numpy()[:, :, ::7, ::7] - z["clip_input_sample"]).max() < 1e-5
    out, dists = E.image_estimator_heads(sd, "", IMAGE_CFG, z["features"])
    for name in ("roughness", "metallic"):
        a, b = dists[name]
        assert np.allclose(a.numpy(), z["alpha." + name], rtol=2e-6, atol=1e-6)
        assert np.allclose(b.numpy(), z["beta." + name], rtol=2e-6, atol=1e-6)
        got, ref = out["decoder_" + name].numpy(), z["out.decoder_" + name]
        assert got.shape == ref.shape == (1, 1, 1) and np.abs(got - ref).max() < 2e-6


def test_beta_mode_matches_torch_distribution():
    rng = np.random.default_rng(0)
    a = torch.from_numpy(np.concatenate([rng.uniform(0.05, 4.0, 200), [0.5, 0.5, 1.0, 1.0, 0.3, 2.0]]).astype(np.float32))
    b = torch.from_numpy(np.concatenate([rng.uniform(0.05, 4.0, 200), [0.5, 0.7, 1.0, 2.0, 2.0, 0.3]]).astype(np.float32))
    ref = torch.distributions.Beta(a, b).mode
    got = E.beta_mode(a, b)
    both_nan = torch.isnan(ref) & torch.isnan(got)
    assert torch.equal(torch.isnan(ref), torch.isnan(got))
    assert torch.equal(ref[~both_nan], got[~both_nan])


def test_clip_tower_matches_transformers_clip():
    z, sd = _load("sf3d_clip.npz")
    W, L, NH, P, S, Edim, MLP = [int(x) for x in z["cfg"]]
    feats, tokens = E.clip_visual_forward(sd, "visual.", z["image"], NH, return_tokens=True)
    assert feats.shape == (2, Edim)
    assert np.abs(tokens.numpy() - z["hidden_last"]).max() < 2e-5
    assert np.abs(feats.numpy() - z["out"]).max() < 2e-5
    # bf16 rounding points stay close (what the HIP path is compared with)
    fb = E.clip_visual_forward(sd, "visual.", z["image"], NH, bf16=True)
    assert np.abs(fb.numpy() - z["out"]).max() < 0.15
