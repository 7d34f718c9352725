"""The torch-fp32 transformer oracle (oracle/tsr_ref.py) against vectors produced by the reference itself."""
import os

import numpy as np
import torch

from conftest import GOLDEN
from oracle import tsr_ref
from sculptmate_amd import synth
from sculptmate_amd.tsr.spec import TINY_CFG, make_cfg


def test_tiny_tsr_forward_matches_reference():
    g = np.load(os.path.join(GOLDEN, "tsr_tiny.npz"))
    sd = synth.tsr_state(seed=21, cfg=TINY_CFG)
    img = synth.composite_rgb(synth.image_rgba(seed=22, size=TINY_CFG["cond_image_size"]))
    col = {}
    codes = tsr_ref.tsr_forward(sd, img, TINY_CFG, pos_mode="size", collect=col).numpy()
    # fp32 vs fp32, different op order only
    np.testing.assert_allclose(col["ctx"].numpy(), g["ctx"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(codes, g["scene_codes"][0], rtol=1e-4, atol=2e-5)


def test_full_size_block_matches_reference():
    g = np.load(os.path.join(GOLDEN, "tsr_block.npz"))
    cfg = make_cfg(vit_layers=1, layers=1)
    sd = synth.tsr_state(seed=23, cfg=cfg)
    h = torch.from_numpy(np.random.default_rng(24).standard_normal((3072, 1024), dtype=np.float32))
    ctx = torch.from_numpy(np.random.default_rng(25).standard_normal((1025, 768), dtype=np.float32))
    with torch.no_grad():
        y = tsr_ref.block_forward(sd, "backbone.transformer_blocks.0.", h, ctx, 16).numpy()
    np.testing.assert_allclose(y.reshape(-1)[g["idx"]], g["y"], rtol=1e-4, atol=5e-5)
    assert abs(y.astype(np.float64).sum() - g["ysum"][0]) < 1e-3 * g["ysum"][1]


def test_upsample_matches_reference():
    g = np.load(os.path.join(GOLDEN, "upsample.npz"))
    rng = np.random.default_rng([7, 15])
    w = synth._uniform(rng, (1024, 40, 2, 2), 1.0 / np.sqrt(160.0))
    b = synth._uniform(rng, (40,), 1.0 / np.sqrt(160.0))
    x = np.random.default_rng(8).standard_normal((1, 3, 1024, 32, 32), dtype=np.float32)
    cfg = make_cfg()
    tokens_ct = torch.from_numpy(x[0]).permute(1, 0, 2, 3).reshape(1024, 3072)
    with torch.no_grad():
        y = tsr_ref.upsample_forward({"post_processor.upsample.weight": w, "post_processor.upsample.bias": b},
                                     tokens_ct, cfg).numpy()
    np.testing.assert_allclose(y.reshape(-1)[g["idx"]], g["y"], rtol=1e-4, atol=1e-5)


def test_pos_embedding_host_interpolation_matches_torch_bicubic():
    """Product host code (numpy bicubic) vs torch F.interpolate in both HF modes."""
    from sculptmate_amd.tsr.posemb import interpolate_pos_embedding

    pos = np.random.default_rng(1).standard_normal((1, 197, 48)).astype(np.float32)
    for mode in ("size", "scale_factor"):
        for n in (32, 4, 14, 20):
            a = interpolate_pos_embedding(pos, n, mode)
            b = tsr_ref.interpolate_pos(pos, n, mode).numpy()
            np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-5)


def test_pos_embedding_438_form_matches_reference_vendored_code():
    """The transformers-4.38 `scale_factor` (+0.1) form -- what the reference's pinned transformers runs for TripoSR's ViT
    and what StableFast vendors at sf3d/models/tokenizers/dinov2.py:89-133 -- pinned by outputs of that vendored code
    (tests/golden/make_posemb_goldens.py): the oracle and the product's host interpolation both reproduce it."""
    from sculptmate_amd.tsr.posemb import interpolate_pos_embedding

    g = np.load(os.path.join(GOLDEN, "posemb_438.npz"))
    for name in ("tsr", "sf3d"):
        table, n, want = g[name + ".table"], int(g[name + ".n_side"]), g[name + ".out"]
        a = tsr_ref.interpolate_pos(table, n, "scale_factor").numpy()
        np.testing.assert_allclose(a, want, rtol=0, atol=1e-6)
        b = interpolate_pos_embedding(table, n, "scale_factor")
        # fp32 rounding of the cubic coefficients: the reference's own fp32 result is 8e-6 from the exact (fp64) interpolation
        np.testing.assert_allclose(b, want, rtol=0, atol=1e-5)
        # and it is NOT the `size=` form: the two differ by far more than the tolerance above
        c = interpolate_pos_embedding(table, n, "size")
        assert np.abs(c - want).max() > 1e-3


def test_preprocessor_matches_reference():
    from sculptmate_amd.tsr.utils import ImagePreprocessor

    g = np.load(os.path.join(GOLDEN, "preproc.npz"))
    img = synth.composite_rgb(synth.image_rgba(seed=27, size=1024))
    y = ImagePreprocessor()(img, 512).numpy()
    assert y.shape == (1, 512, 512, 3)
    assert np.array_equal(y.reshape(-1)[g["idx"]], g["y"])
    # PIL / uint8 / list inputs and the identity case
    u8 = (img[:512, :512] * 255).astype(np.uint8)
    a = ImagePreprocessor()([u8, u8], 512)
    assert a.shape == (2, 512, 512, 3) and torch.equal(a[0], torch.from_numpy(u8.astype(np.float32) / 255.0))


def test_bf16_variant_stays_close_to_fp32_oracle():
    sd = synth.tsr_state(seed=21, cfg=TINY_CFG)
    img = synth.composite_rgb(synth.image_rgba(seed=22, size=TINY_CFG["cond_image_size"]))
    a = tsr_ref.tsr_forward(sd, img, TINY_CFG, pos_mode="size").numpy()
    b = tsr_ref.tsr_forward(sd, img, TINY_CFG, pos_mode="size", bf16=True).numpy()
    assert np.abs(a - b).max() < 0.05 * np.abs(a).max()
