"""HIP UV-space texture baker vs the oracle / the reference's common.py golden (MI355X)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import capi

pytestmark = pytest.mark.gpu


def _mesh(seed, n):
    import sys

    sys.path.insert(0, GOLDEN)
    rng = np.random.default_rng(seed)
    g = np.linspace(0.03, 0.97, n)
    u, v = np.meshgrid(g, g, indexing="ij")
    uv = np.stack([u, v], -1)
    uv[1:-1, 1:-1] += (rng.random((n - 2, n - 2, 2)) - 0.5) * (0.4 / n)
    uv = uv.reshape(-1, 2).astype(np.float32)
    f = []
    for i in range(n - 1):
        for j in range(n - 1):
            a, b, c, d = i * n + j, (i + 1) * n + j, (i + 1) * n + j + 1, i * n + j + 1
            f += [[a, b, c], [a, c, d]] if (i + j) % 2 == 0 else [[a, b, d], [b, c, d]]
    return uv, np.array(f, np.int32), rng.standard_normal((n * n, 3)).astype(np.float32)


def test_vs_reference_golden(cuda):
    from sculptmate_amd import ops

    g = np.load(os.path.join(GOLDEN, "baker.npz"))
    res = g["rast"].shape[0]
    rast = ops.bake_rasterize(torch.from_numpy(g["uv"]).to(cuda), torch.from_numpy(g["faces"]).to(cuda), res)
    assert np.array_equal(rast[..., 3].cpu().numpy(), g["rast"][..., 3])
    np.testing.assert_allclose(rast[..., :3].cpu().numpy(), g["rast"][..., :3], rtol=0, atol=2e-5)
    inter = ops.bake_interpolate(torch.from_numpy(g["attr"]).to(cuda), rast, torch.from_numpy(g["faces"]).to(cuda))
    np.testing.assert_allclose(inter.cpu().numpy(), g["inter"], rtol=0, atol=5e-5)


@pytest.mark.parametrize("n,res", [(5, 16), (33, 128), (80, 257)])
def test_bit_exact_vs_oracle(cuda, n, res):
    from sculptmate_amd import ops

    uv, f, attr = _mesh(n, n)
    ref = capi.bake_rasterize(uv, f, res)
    rast = ops.bake_rasterize(torch.from_numpy(uv).to(cuda), torch.from_numpy(f).to(cuda), res)
    assert np.array_equal(rast.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    ri = capi.bake_interpolate(attr, f, ref)
    inter = ops.bake_interpolate(torch.from_numpy(attr).to(cuda), rast, torch.from_numpy(f).to(cuda))
    assert np.array_equal(inter.cpu().numpy().view(np.uint32), ri.view(np.uint32))


def test_overlap_and_empty(cuda):
    from sculptmate_amd import ops

    uv = torch.tensor([[0, 0], [1, 0], [0, 1], [1, 1]], dtype=torch.float32, device=cuda)
    f = torch.tensor([[0, 1, 2], [0, 1, 3], [0, 3, 2]], dtype=torch.int32, device=cuda)
    r = ops.bake_rasterize(uv, f, 16).cpu().numpy()
    ref = capi.bake_rasterize(uv.cpu().numpy(), f.cpu().numpy(), 16)
    assert np.array_equal(r.view(np.uint32), ref.view(np.uint32))
    e = ops.bake_rasterize(uv, f[:0], 8).cpu().numpy()
    assert (e[..., 3] == -1).all() and (e[..., :3] == 0).all()


def test_dll_compatible_host_entry_points(cuda):
    """rasterize_cpu / interpolate_cpu with the exact ctypes declarations of baker.py:34-41, 94-101."""
    from sculptmate_amd import _lib

    dll = ctypes.CDLL(_lib.SO_PATH)
    dll.rasterize_cpu.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_size_t, ctypes.POINTER(ctypes.c_int),
                                  ctypes.c_size_t, ctypes.c_longlong, ctypes.POINTER(ctypes.c_float)]
    dll.rasterize_cpu.restype = None
    dll.interpolate_cpu.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_size_t, ctypes.POINTER(ctypes.c_int),
                                    ctypes.c_size_t, ctypes.POINTER(ctypes.c_float), ctypes.c_longlong,
                                    ctypes.POINTER(ctypes.c_float)]
    dll.interpolate_cpu.restype = None
    uv, f, attr = _mesh(3, 12)
    res = 64
    uvf, idx = uv.flatten(), f.flatten()
    out = np.zeros(res * res * 4, np.float32)
    dll.rasterize_cpu(uvf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), uv.shape[0],
                      idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), f.shape[0], res,
                      out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    ref = capi.bake_rasterize(uv, f, res)
    assert np.array_equal(out.reshape(res, res, 4).view(np.uint32), ref.view(np.uint32))
    inter = np.zeros(res * res * 3, np.float32)
    af = attr.flatten()
    dll.interpolate_cpu(af.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), attr.shape[0],
                        idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), f.shape[0],
                        out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), res,
                        inter.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    assert np.array_equal(inter.reshape(res, res, 3).view(np.uint32), capi.bake_interpolate(attr, f, ref).view(np.uint32))
