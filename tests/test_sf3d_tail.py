"""StableFast geometry tail: oracle vs the reference's own output (CPU) and HIP vs oracle (GPU)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import sf3d_tail as ref



def _inputs():
    # same construction as tests/golden/make_reference_goldens.py::sf3d_tail_inputs (the generator itself needs
    # /root/reference, which does not exist on the GPU box)
    rng = np.random.default_rng(33)
    H = W = 48
    yy, xx = np.mgrid[0:H, 0:W]
    mask = ((xx - 20) ** 2 + (yy - 26) ** 2 < 12 ** 2) | ((xx > 30) & (xx < 36) & (yy > 5) & (yy < 40))
    img = rng.random((1, 3, H, W)).astype(np.float32) * mask[None, None]
    # same construction as tests/golden/make_reference_goldens.py::uv_test_mesh(seed=2, n=8)
    n = 8
    r2 = np.random.default_rng(2)
    gg = np.linspace(0.06, 0.94, n)
    u, v = np.meshgrid(gg, gg, indexing="ij")
    uv = np.stack([u, v], -1).astype(np.float64)
    uv[1:-1, 1:-1] += (r2.random((n - 2, n - 2, 2)) - 0.5) * 0.04
    uv = uv.reshape(-1, 2).astype(np.float32)
    faces = []
    for i in range(n - 1):
        for j in range(n - 1):
            a, b, c, d = i * n + j, (i + 1) * n + j, (i + 1) * n + j + 1, i * n + j + 1
            faces += [[a, b, c], [a, c, d]] if (i + j) % 2 == 0 else [[a, b, d], [b, c, d]]
    faces = np.array(faces, np.int64)
    v_pos = np.concatenate([uv * 2 - 1, (0.3 * np.sin(uv[:, :1] * 6) * np.cos(uv[:, 1:] * 5))], 1).astype(np.float32)
    return img.astype(np.float32), mask[None, None], v_pos, uv, faces


def test_oracle_matches_reference():
    g = np.load(os.path.join(GOLDEN, "sf3d_tail.npz"))
    img, mask, v_pos, uv, faces = _inputs()
    np.testing.assert_allclose(ref.dilate_fill(img, mask, 10).numpy(), g["dilate"], rtol=0, atol=2e-6)
    n = ref.vertex_normals(v_pos, faces)
    np.testing.assert_allclose(n.numpy(), g["v_nrm"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(ref.vertex_tangents(v_pos, uv, n, faces).numpy(), g["v_tng"], rtol=0, atol=5e-6)


@pytest.mark.gpu
def test_hip_matches_oracle_and_reference(cuda):
    from sculptmate_amd import ops

    g = np.load(os.path.join(GOLDEN, "sf3d_tail.npz"))
    img, mask, v_pos, uv, faces = _inputs()
    d = ops.dilate_fill(torch.from_numpy(img).to(cuda), torch.from_numpy(mask).to(cuda), 10).cpu().numpy()
    np.testing.assert_allclose(d, g["dilate"], rtol=0, atol=3e-6)
    n = ops.vertex_normals(torch.from_numpy(v_pos).to(cuda), torch.from_numpy(faces).to(cuda))
    np.testing.assert_allclose(n.cpu().numpy(), g["v_nrm"], rtol=0, atol=3e-6)
    t = ops.vertex_tangents(torch.from_numpy(v_pos).to(cuda), torch.from_numpy(uv).to(cuda), n, torch.from_numpy(faces).to(cuda))
    np.testing.assert_allclose(t.cpu().numpy(), g["v_tng"], rtol=0, atol=1e-5)
    # larger random case vs the oracle (int32 faces, 512^2 texture)
    rng = np.random.default_rng(5)
    m = torch.from_numpy(rng.random((1, 1, 512, 512)) > 0.7)
    im = torch.from_numpy(rng.random((1, 3, 512, 512)).astype(np.float32)) * m
    np.testing.assert_allclose(ops.dilate_fill(im.to(cuda), m.to(cuda), 10).cpu().numpy(), ref.dilate_fill(im, m, 10).numpy(), rtol=0, atol=5e-6)
    n32 = ops.vertex_normals(torch.from_numpy(v_pos).to(cuda), torch.from_numpy(faces.astype(np.int32)).to(cuda))
    assert torch.allclose(n32, n, atol=1e-6)
