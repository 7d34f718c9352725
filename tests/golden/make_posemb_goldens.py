"""Golden vectors for the transformers-4.38 form of the ViT position-embedding interpolation (scale_factor with the
+0.1 offset), produced by IMPORTING the reference's vendored copy of that code:

    /root/reference/StableFast/sf3d/models/tokenizers/dinov2.py:89-133  Dinov2Embeddings.interpolate_pos_encoding

which is statement for statement the 4.38 `ViTEmbeddings.interpolate_pos_encoding` the TripoSR tokenizer runs under the
reference's pinned transformers==4.38.0 (TripoSR/tsr/models/tokenizers/image.py:49-51 with interpolate_pos_encoding=True).
The installed transformers (5.x) only has the `size=` form, so this is the one place the 4.38 arithmetic can be pinned.

Run in the build container only (needs /root/reference):  python tests/golden/make_posemb_goldens.py
Output: tests/golden/posemb_438.npz -- inputs (seeded tables) and the reference's outputs for
  * TripoSR's case: 14x14 (+CLS) table, 512 px image, patch 16 -> 32x32
  * SF3D's case:    37x37 (+CLS) table, 512 px image, patch 14 -> 36x36
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
sys.path.insert(0, "/root/reference/StableFast")
import transformers.pytorch_utils as _pu  # noqa: E402

for _n in ("find_pruneable_heads_and_indices", "prune_linear_layer"):
    if not hasattr(_pu, _n):
        setattr(_pu, _n, lambda *a, **k: None)


def main():
    from sf3d.models.tokenizers import dinov2
    from transformers.models.dinov2.configuration_dinov2 import Dinov2Config

    out = {}
    for name, grid, patch, image, dim, seed in (("tsr", 14, 16, 512, 8, 0), ("sf3d", 37, 14, 512, 8, 1)):
        cfg = Dinov2Config(hidden_size=dim, num_hidden_layers=1, num_attention_heads=1, image_size=grid * patch, patch_size=patch)
        emb = dinov2.Dinov2Embeddings(cfg).eval()
        g = torch.Generator().manual_seed(seed)
        table = torch.randn(1, grid * grid + 1, dim, generator=g)
        with torch.no_grad():
            emb.position_embeddings.copy_(table)
            n = image // patch
            tokens = torch.zeros(1, n * n + 1, dim)  # only its shape is read
            ref = emb.interpolate_pos_encoding(tokens, image, image)
        assert ref.shape == (1, n * n + 1, dim)
        out[name + ".table"] = table.numpy()
        out[name + ".n_side"] = np.int64(n)
        out[name + ".out"] = ref[0].numpy()
    np.savez_compressed(os.path.join(HERE, "posemb_438.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
