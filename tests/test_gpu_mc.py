"""HIP marching cubes vs the C oracle / scikit-image goldens: bit-exact vertices, faces and order (MI355X)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import capi

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(GOLDEN, "mc_skimage.npz"))
NAMES = sorted(k[:-4] for k in G.files if k.endswith("_vol"))


def _same(v, f, rv, rf):
    assert tuple(f.shape) == rf.shape and tuple(v.shape) == rv.shape, (f.shape, rf.shape, v.shape, rv.shape)
    assert np.array_equal(f.cpu().numpy(), rf)
    assert np.array_equal(v.cpu().numpy().view(np.uint32), rv.view(np.uint32))


@pytest.mark.parametrize("name", NAMES)
def test_bit_exact_vs_skimage_golden(cuda, name):
    from sculptmate_amd import ops

    v, f = ops.marching_cubes(torch.from_numpy(G[name + "_vol"]).to(cuda), 0.0)
    assert f.dtype == torch.int32
    _same(v, f, G[name + "_verts"], G[name + "_faces"])


@pytest.mark.parametrize("shape,seed", [((2, 2, 2), 0), ((3, 2, 5), 1), ((17, 9, 33), 2), ((40, 41, 42), 3), ((64, 64, 64), 4)])
def test_noise_volumes_vs_oracle(cuda, shape, seed):
    """White noise hits every ambiguous Lewiner case and the centre vertex."""
    from sculptmate_amd import ops

    vol = np.random.default_rng(seed).standard_normal(shape).astype(np.float32)
    rv, rf = capi.marching_cubes(vol, 0.0)
    v, f = ops.marching_cubes(torch.from_numpy(vol).to(cuda), 0.0)
    _same(v, f, rv, rf)


def test_integer_volume_with_exact_zeros_and_degenerate_saddles(cuda):
    from sculptmate_amd import ops

    vol = np.random.default_rng(7).integers(-2, 3, (20, 21, 19)).astype(np.float32)
    rv, rf = capi.marching_cubes(vol, 0.0)
    v, f = ops.marching_cubes(torch.from_numpy(vol).to(cuda), 0.0)
    _same(v, f, rv, rf)


def test_nonzero_level_and_classic_mode(cuda):
    from sculptmate_amd import ops

    vol = np.random.default_rng(8).standard_normal((15, 16, 17)).astype(np.float32)
    rv, rf = capi.marching_cubes(vol, 0.25)
    v, f = ops.marching_cubes(torch.from_numpy(vol).to(cuda), 0.25)
    _same(v, f, rv, rf)
    rv, rf = capi.marching_cubes(vol, 0.0, use_classic=True)
    v, f = ops.marching_cubes(torch.from_numpy(vol).to(cuda), 0.0, use_classic=True)
    _same(v, f, rv, rf)


def test_reference_order_output(cuda):
    """MarchingCubeHelper.forward + scale_tensor (isosurface.py:49-53, system.py:185-189)."""
    from sculptmate_amd import ops

    R = 24
    vol = G["density_vol"] if G["density_vol"].shape[0] == R else G["sphere_vol"]
    R = vol.shape[0]
    rv, rf = capi.reference_isosurface(-vol, R)
    rv = rv * np.float32(0.87 - (-0.87)) + np.float32(-0.87)
    v, f = ops.marching_cubes(torch.from_numpy(vol).to(cuda), 0.0, reference_order=True, vert_div=R - 1.0,
                              vert_mul=0.87 - (-0.87), vert_add=-0.87)
    assert f.dtype == torch.int64
    assert np.array_equal(f.cpu().numpy(), rf)
    assert np.array_equal(v.cpu().numpy().view(np.uint32), rv.astype(np.float32).view(np.uint32))


def test_errors_like_skimage(cuda):
    from sculptmate_amd import ops

    with pytest.raises(ValueError):
        ops.marching_cubes(torch.ones(4, 4, 4, device=cuda), 0.0)
    vol = -torch.ones(3, 3, 3, device=cuda)
    vol[1, 1, 1] = 0.0
    with pytest.raises(RuntimeError):
        ops.marching_cubes(vol, 0.0)
    with pytest.raises(ops.SculptError):
        ops.marching_cubes(torch.ones(1, 4, 4, device=cuda), 0.0)


def _closed_manifold(f):
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    ue, cnt = np.unique(np.sort(e, 1), axis=0, return_counts=True)
    return (cnt == 2).all(), len(ue)


def test_full_size_256_properties_and_oracle(cuda):
    """BASELINE size: analytic blob field at 256^3 -- closed 2-manifold, Euler characteristic 2,
    and (the oracle finishes 256^3 in seconds) bit-exact against the oracle."""
    from sculptmate_amd import ops

    R = 256
    g = torch.linspace(-0.87, 0.87, R, device=cuda)
    x, y, z = torch.meshgrid(g, g, g, indexing="ij")
    dens = 25.0 * torch.exp(9.0 * (0.5 - torch.sqrt(x * x + 1.3 * y * y + 0.8 * z * z)))
    vol = (dens - 25.0).contiguous()
    v, f = ops.marching_cubes(vol, 0.0)
    fn = f.cpu().numpy()
    ok, ne = _closed_manifold(fn)
    assert ok
    assert len(v) - ne + len(fn) == 2
    rv, rf = capi.marching_cubes(vol.cpu().numpy(), 0.0)
    _same(v, f, rv, rf)
    # idempotent / deterministic
    v2, f2 = ops.marching_cubes(vol, 0.0)
    assert torch.equal(v, v2) and torch.equal(f, f2)


def test_dense_noisy_density_field_256_vs_oracle(cuda):
    """The bench-like case: a 256^3 density volume from the fused MLP kernel (random decoder, ~1.5 % of the
    voxels inside) -- a ~1 M vertex mesh with every Lewiner case -- bit-exact against the oracle."""
    from sculptmate_amd import ops, synth

    R = 256
    Ws, bs = synth.decoder_lists(synth.decoder_state(seed=71))
    mlp = ops.PackedMLP(Ws, bs, cuda)
    planes = torch.from_numpy(synth.triplane(seed=72, scale=2.0)).to(cuda)
    dens = ops.density_grid(planes, mlp, R)
    thr = float(dens.float().quantile(torch.tensor(0.985, device=cuda)) if dens.numel() < 2 ** 24 else
                torch.quantile(dens[:: 7].float(), 0.985))
    vol = (dens - thr).view(R, R, R).contiguous()
    v, f = ops.marching_cubes(vol, 0.0)
    assert len(v) > 200000
    rv, rf = capi.marching_cubes(vol.cpu().numpy(), 0.0)
    _same(v, f, rv, rf)


def test_nan_volume_is_reported(cuda):
    """fminf/fmaxf drop NaN, so the count pass looks for it explicitly (error 13) instead of meshing garbage."""
    import torch

    from sculptmate_amd import _lib, ops

    vol = torch.randn(12, 12, 12, device=cuda)
    vol[5, 6, 7] = float("nan")
    with pytest.raises(_lib.SculptError) as ei:
        ops.marching_cubes(vol, 0.0)
    assert ei.value.code == _lib.ERR_MC_NAN
    vol[5, 6, 7] = 0.5
    v, f = ops.marching_cubes(vol, 0.0)
    assert v.shape[0] > 0


def test_speculative_emit_equals_the_two_phase_path(cuda, monkeypatch):
    """ops.marching_cubes queues count and emit before it reads the counts back once it has seen a shape (buffers sized by the
    largest mesh so far + 25 %): the same vertices and faces as the count -> allocate -> emit path, whether the mesh fits the
    estimate, is far smaller, or does not fit (nothing written, exact path instead); the error cases keep their exceptions."""
    from sculptmate_amd import ops

    shape = (48, 50, 52)
    rng = np.random.default_rng(11)
    zz, yy, xx = np.meshgrid(*[np.linspace(-1, 1, n, dtype=np.float32) for n in shape], indexing="ij")
    small = (0.3 - np.sqrt(xx ** 2 + yy ** 2 + zz ** 2)).astype(np.float32)            # one small sphere
    medium = (0.7 - np.sqrt(xx ** 2 + yy ** 2 + zz ** 2)).astype(np.float32)
    noise = rng.standard_normal(shape).astype(np.float32)                                # far more surface than any estimate
    vols = [torch.from_numpy(v).to(cuda) for v in (medium, medium * 1.01, small, noise, noise, medium)]
    monkeypatch.setattr(ops, "_MC_SPECULATE", False)
    want = [ops.marching_cubes(v, 0.0, reference_order=True, vert_div=51.0) for v in vols]
    monkeypatch.setattr(ops, "_MC_SPECULATE", True)
    monkeypatch.setattr(ops, "_MC_CAPACITY", {})
    for i, v in enumerate(vols):
        gv, gf = ops.marching_cubes(v, 0.0, reference_order=True, vert_div=51.0)
        assert gf.dtype == torch.int64 and torch.equal(gv, want[i][0]) and torch.equal(gf, want[i][1]), i
        assert gv.is_contiguous() and gf.is_contiguous()
    (cap,) = ops._MC_CAPACITY.values()
    assert cap[0] >= want[3][0].shape[0] and cap[1] >= want[3][1].shape[0]              # grown by the overflowing call
    # error semantics on the speculative path (the shape has a capacity now)
    with pytest.raises(ValueError):
        ops.marching_cubes(torch.ones(shape, device=cuda), 0.0, reference_order=True, vert_div=51.0)
    e = -torch.ones(shape, device=cuda)
    e[1, 1, 1] = 0.0
    with pytest.raises(RuntimeError):
        ops.marching_cubes(e, 0.0, reference_order=True, vert_div=51.0)
    bad = vols[0].clone()
    bad[5, 5, 5] = float("nan")
    with pytest.raises(ops.SculptError):
        ops.marching_cubes(bad, 0.0, reference_order=True, vert_div=51.0)
    # and a good call afterwards is still right
    gv, gf = ops.marching_cubes(vols[0], 0.0, reference_order=True, vert_div=51.0)
    assert torch.equal(gv, want[0][0]) and torch.equal(gf, want[0][1])


def _planes_of(vol, level=0.0):
    """uint32 words along the last axis, bit i % 32 of word i / 32 = vol > level (bits past the row are zero), as int32 [n0*n1][words]."""
    n0, n1, n2 = vol.shape
    words = (n2 + 31) // 32
    bits = np.zeros((n0 * n1, words * 32), np.uint8)
    bits[:, :n2] = (vol.reshape(n0 * n1, n2) > np.float32(level))
    packed = np.packbits(bits.reshape(n0 * n1, words, 32), axis=-1, bitorder="little").view(np.uint32).reshape(n0 * n1, words)
    return torch.from_numpy(packed.view(np.int32))


@pytest.mark.parametrize("shape,seed", [((2, 2, 2), 0), ((17, 9, 33), 2), ((40, 41, 42), 3), ((64, 64, 64), 4), ((9, 10, 300), 5),
                                         ((12, 7, 513), 6)])
def test_count_phase_from_sign_planes_equals_the_plain_one(cuda, shape, seed, monkeypatch):
    """sculpt_mc_count_launch_signed: the cell signs come from caller-supplied planes and the volume is read only in bricks with an
    active cell -- same vertices, faces and order as the plain count phase (noise: every Lewiner case; rows longer than one
    256-cell segment; sizes that are not multiples of 32 / 64), with and without the speculative emit."""
    from sculptmate_amd import ops

    rng = np.random.default_rng(seed)
    vol = rng.standard_normal(shape).astype(np.float32)
    if min(shape) > 4:
        vol[2:4] = np.abs(vol[2:4]) + 1.0          # slabs without any sign change: bricks that never read the volume
    volt, planes = torch.from_numpy(vol).to(cuda), _planes_of(vol).to(cuda)
    for spec in (False, True, True):
        monkeypatch.setattr(ops, "_MC_SPECULATE", spec)
        v, f = ops.marching_cubes(volt, 0.0)
        sv, sf = ops.marching_cubes(volt, 0.0, sign_planes=planes)
        assert torch.equal(sv, v) and torch.equal(sf, f)
    rv, rf = capi.marching_cubes(vol, 0.0)
    _same(sv, sf, rv, rf)
    # integer values with exact zeros (value == level is "not above")
    ivol = rng.integers(-2, 3, shape).astype(np.float32)
    if (ivol > 0).any() and (ivol <= 0).any():
        it = torch.from_numpy(ivol).to(cuda)
        v, f = ops.marching_cubes(it, 0.0, reference_order=True, vert_div=3.0)
        sv, sf = ops.marching_cubes(it, 0.0, reference_order=True, vert_div=3.0, sign_planes=_planes_of(ivol).to(cuda))
        assert torch.equal(sv, v) and torch.equal(sf, f)


def test_sign_plane_count_keeps_the_error_semantics(cuda):
    from sculptmate_amd import ops

    shape = (20, 21, 70)
    ones = torch.ones(shape, device=cuda)
    with pytest.raises(ValueError):       # level below the data range: the planes say "no surface", the plain form names the error
        ops.marching_cubes(ones, 0.0, sign_planes=_planes_of(ones.cpu().numpy()).to(cuda))
    e = -torch.ones(shape, device=cuda)
    e[1, 1, 1] = 0.0
    with pytest.raises(RuntimeError):
        ops.marching_cubes(e, 0.0, sign_planes=_planes_of(e.cpu().numpy()).to(cuda))
    vol = np.random.default_rng(1).standard_normal(shape).astype(np.float32)
    bad = vol.copy()
    bad[5, 5, 5] = np.nan                 # inside the surface band of a noise volume: its brick is staged, the NaN is seen
    with pytest.raises(ops.SculptError):
        ops.marching_cubes(torch.from_numpy(bad).to(cuda), 0.0, sign_planes=_planes_of(bad).to(cuda))
    with pytest.raises(AssertionError):   # not in slab mode
        ops.marching_cubes(torch.from_numpy(vol).to(cuda), 0.0, slab=dict(axis0_offset=0), sign_planes=_planes_of(vol).to(cuda))


# ------------------------------------------------------------------------------------------------- the record pool (round 6)
def test_workspace_is_a_record_pool_and_overflow_is_reported_then_retried(cuda):
    """The workspace holds 8 bytes per ACTIVE cell in a pool of the caller's capacity.  With a pool of 1000 records a noise volume
    (every cell active) overflows: the count phase still returns the right totals, with SCULPT_ERR_MC_WORKSPACE and the number of
    active cells; a speculative emit after it writes nothing; ops.marching_cubes repeats the count with a larger workspace and
    returns the oracle's mesh."""
    import ctypes

    from sculptmate_amd import _lib, ops

    shape = (24, 25, 26)
    vol_np = np.random.default_rng(11).standard_normal(shape).astype(np.float32)
    rv, rf = capi.marching_cubes(vol_np, 0.0)
    vol = torch.from_numpy(vol_np).to(cuda)
    lib = _lib.lib
    small = lib.sculpt_mc_workspace_bytes_for(*shape, 1000)
    assert small < lib.sculpt_mc_workspace_bytes_for(*shape, 100000) and lib.sculpt_mc_workspace_bytes(*shape) == lib.sculpt_mc_workspace_bytes_for(*shape, 0)
    ws = torch.empty(small, dtype=torch.uint8, device=cuda)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    vp, wp = ctypes.c_void_p(vol.data_ptr()), ctypes.c_void_p(ws.data_ptr())
    _lib.check(lib.sculpt_mc_count_launch_for(vp, None, 0, *shape, 0.0, 0, 1000, wp, st))
    verts = torch.full((rv.shape[0], 3), -7.0, dtype=torch.float32, device=cuda)
    faces = torch.full((rf.shape[0], 3), -7, dtype=torch.int32, device=cuda)
    _lib.check(lib.sculpt_mc_emit_capped(vp, *shape, 0.0, 0, wp, 1.0, 1.0, 0.0, 0, ctypes.c_void_p(verts.data_ptr()), rv.shape[0],
                                         ctypes.c_void_p(faces.data_ptr()), rf.shape[0], None, st))
    nv, nf, na = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    rc = lib.sculpt_mc_count_read_ex(*shape, 0.0, 0, wp, ctypes.byref(nv), ctypes.byref(nf), None, ctypes.byref(na), st)
    assert rc == _lib.ERR_MC_WORKSPACE and "record pool" in _lib.last_error()
    assert (nv.value, nf.value) == (rv.shape[0], rf.shape[0]) and 1000 < na.value <= 23 * 24 * 25
    assert bool((verts == -7.0).all()) and bool((faces == -7).all())          # the emit after an overflow wrote nothing
    # the same through a pool that holds them (a pool for every cell of the grid is never split: it cannot overflow)
    big = 23 * 24 * 25
    ws = torch.empty(lib.sculpt_mc_workspace_bytes_for(*shape, big), dtype=torch.uint8, device=cuda)
    wp = ctypes.c_void_p(ws.data_ptr())
    _lib.check(lib.sculpt_mc_count_launch_for(vp, None, 0, *shape, 0.0, 0, big, wp, st))
    _lib.check(lib.sculpt_mc_count_read_ex(*shape, 0.0, 0, wp, ctypes.byref(nv), ctypes.byref(nf), None, ctypes.byref(na), st))
    _lib.check(lib.sculpt_mc_emit(vp, *shape, 0.0, 0, wp, 1.0, 1.0, 0.0, 0, ctypes.c_void_p(verts.data_ptr()),
                                  ctypes.c_void_p(faces.data_ptr()), None, st))
    _same(verts, faces, rv, rf)
    # ops.marching_cubes: a pool that is too small for this shape, then the retry (and the capacity is remembered)
    key = (vol.device,) + shape
    ops._MC_REC_CAPACITY[key] = 500
    try:
        v, f = ops.marching_cubes(vol, 0.0)
        _same(v, f, rv, rf)
        assert ops._MC_REC_CAPACITY[key] >= na.value
        v, f = ops.marching_cubes(vol, 0.0)
        _same(v, f, rv, rf)
        # ... and in slab mode (sculpt_mc_count_read returns early there for an empty slab: the overflow must come first)
        ops._MC_REC_CAPACITY[key] = 500
        v, f, top, mm = ops.marching_cubes(vol, 0.0, slab=dict(axis0_offset=0, halo_low=False))
        _same(v, f, rv, rf)
    finally:
        ops._MC_REC_CAPACITY.pop(key, None)


def test_workspace_footprint(cuda):
    """<= 100 MB at 256^3 and <= 0.8 GB at 512^3 (the dense records + lattice-edge map took 403 MB / 3.2 GB)."""
    from sculptmate_amd import _lib

    assert _lib.lib.sculpt_mc_workspace_bytes(256, 256, 256) <= 100 * 2 ** 20
    assert _lib.lib.sculpt_mc_workspace_bytes(512, 512, 512) <= 0.8 * 2 ** 30


@pytest.mark.parametrize("shape,seed", [((6, 7, 300), 21), ((5, 300, 9), 22), ((40, 3, 520), 23)])
def test_vertex_ids_across_row_segments_and_on_the_low_faces(cuda, shape, seed):
    """Vertex ids come from the OWNER cell's record (a neighbour at offset {0,-1}^3, found through the row's active mask): rows
    longer than one 256-cell segment (the owner may sit in the previous segment), thin volumes where every cell lies on a low
    face (the rank walk instead of the two record bits), noise (every Lewiner case)."""
    from sculptmate_amd import ops

    vol = np.random.default_rng(seed).standard_normal(shape).astype(np.float32)
    rv, rf = capi.marching_cubes(vol, 0.0)
    v, f = ops.marching_cubes(torch.from_numpy(vol).to(cuda), 0.0)
    _same(v, f, rv, rf)
    rv, rf = capi.marching_cubes(vol, 0.0, use_classic=True)
    v, f = ops.marching_cubes(torch.from_numpy(vol).to(cuda), 0.0, use_classic=True)
    _same(v, f, rv, rf)


@pytest.mark.parametrize("tilt", [0.0, 0.013])
def test_surfaces_lying_flat_in_a_brick(cuda, tilt):
    """A plane across the long axis fills whole rows of cells: the emit pass then takes the brick through LDS one layer of rows
    (or one row) at a time instead of all at once; same mesh as the oracle's.  (White noise at 256^3 -- every cell active -- is
    test_dense_noisy_density_field_256_vs_oracle.)"""
    from sculptmate_amd import ops

    n0, n1, n2 = 11, 37, 300
    z, y, x = np.meshgrid(np.arange(n0), np.arange(n1), np.arange(n2), indexing="ij")
    vol = (z - 4.37 + tilt * x + 0.02 * np.sin(0.9 * y) + 0.4 * np.sin(0.05 * x) * (tilt > 0)).astype(np.float32)
    rv, rf = capi.marching_cubes(vol, 0.0)
    v, f = ops.marching_cubes(torch.from_numpy(vol).to(cuda), 0.0)
    _same(v, f, rv, rf)
    # two sheets, one directly above the other: two full layers of rows in the same bricks
    vol2 = np.minimum(vol, 3.1 - vol + 4.0).astype(np.float32)
    rv, rf = capi.marching_cubes(vol2, 0.0)
    v, f = ops.marching_cubes(torch.from_numpy(vol2).to(cuda), 0.0)
    _same(v, f, rv, rf)
