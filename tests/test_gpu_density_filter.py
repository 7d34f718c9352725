"""The two-pass ("filtered") dense density grid (csrc/density_filter.hip, sculpt_density_grid_filtered) against the full three-limb
evaluation (sculpt_density_grid_ex, itself pinned to the C oracle in test_gpu_triplane.py): marching cubes must see the same bits
at every corner of every active cell and the same sign everywhere, i.e. give the same mesh bit for bit (MI355X)."""
import numpy as np
import pytest
import torch

from oracle import capi
from sculptmate_amd import synth

pytestmark = pytest.mark.gpu

THR = 25.0


def _field(cuda, seed, inside, scale=3.0, size=64, n_hidden_layers=9):
    """(planes, PackedMLP, Ws, bs): a decoder whose last bias is shifted so that `inside` of a 16^3 oracle probe exceeds THR."""
    from sculptmate_amd import ops

    Ws, bs = synth.decoder_lists(synth.decoder_state(seed=seed, n_hidden_layers=n_hidden_layers))
    tri_np = synth.smooth_triplane(seed=seed + 1, scale=scale, size=size)
    pre = np.log(capi.density_grid(tri_np, Ws, bs, 16)) + 1.0
    bs[-1] = bs[-1].copy()
    bs[-1][0] += synth.calibrate_density_bias(pre, inside_fraction=inside, threshold=THR)
    return torch.from_numpy(tri_np).to(cuda), ops.PackedMLP(Ws, bs, cuda), Ws, bs


def _margin(tri, mlp, coarse, probe=32):
    from sculptmate_amd import ops

    _, st = ops.density_grid_filtered(tri, mlp, probe, 0.0, out_add=0.0, coarse=coarse, mark_all=True)
    s = ops.filter_stats(st)
    assert s["n_refined"] == s["n_points"] == probe ** 3 and s["n_nonfinite"] == 0
    return max(8.0 * s["max_err"], 1e-3), s["max_err"]


def _needed(full, R, nx=None):
    """bool [nx*R*R]: lattice points whose VALUE marching cubes reads in `full` (tests/_mcneeds.py), and the active-cell count."""
    from _mcneeds import needed_points

    nx = R if nx is None else nx
    need, n_active = needed_points(full.view(nx, R, R) > 0)
    return need.view(-1), n_active


def _assert_same_for_marching_cubes(vol, full, R, nx=None):
    assert int(((vol > 0) != (full > 0)).sum()) == 0, "a lattice point changed its side of the level"
    need, n_active = _needed(full, R, nx)
    assert n_active > 0
    assert torch.equal(vol.view(torch.int32)[need], full.view(torch.int32)[need]), "a value marching cubes reads carries other bits"
    return need


def _same_mesh(a, b):
    return a[0].shape == b[0].shape and a[1].shape == b[1].shape and torch.equal(a[0].view(torch.int32), b[0].view(torch.int32)) \
        and torch.equal(a[1], b[1])


@pytest.mark.parametrize("coarse", ["fp16", "bf16"])
@pytest.mark.parametrize("R", [33, 64, 100])
def test_filtered_grid_gives_marching_cubes_the_full_evaluations_bits(cuda, R, coarse):
    from sculptmate_amd import ops

    tri, mlp, _, _ = _field(cuda, 21, inside=0.1)
    margin, _ = _margin(tri, mlp, coarse)
    full = ops.density_grid(tri, mlp, R, out_add=-THR, precision="bf16l3").clone()
    vol, st = ops.density_grid_filtered(tri, mlp, R, margin, out_add=-THR, coarse=coarse)
    s = ops.filter_stats(st)
    need = _assert_same_for_marching_cubes(vol, full, R)
    assert s["n_points"] == R ** 3 and int(need.sum()) <= s["n_refined"] < R ** 3
    assert s["n_marked"] == s["n_first"] <= s["n_refined"] == s["n_first"] + s["n_second"] and s["n_nonfinite"] == 0
    assert s["max_err"] <= margin / 3.0, (s, margin)   # the run-time guard TSR applies
    assert _same_mesh(ops.marching_cubes(vol.view(R, R, R), 0.0), ops.marching_cubes(full.view(R, R, R), 0.0))


@pytest.mark.parametrize("R", [8, 40, 64])
def test_mark_all_reproduces_the_full_volume_bit_for_bit(cuda, R):
    """Every point through pass C: the list kernel runs the dense kernel's device function, so the volumes are equal bit for bit
    (ragged last z word at R = 8 / 40), whatever tile a point lands in; the recorded error is the coarse pass's largest."""
    from sculptmate_amd import ops

    tri, mlp, _, _ = _field(cuda, 23, inside=0.2)
    full = ops.density_grid(tri, mlp, R, precision="bf16l3").clone()
    coarse_only, _ = ops.density_grid_filtered(tri, mlp, R, 0.0, coarse="fp16", mark_all=True, passes="A")
    coarse_only = coarse_only.clone()
    vol, st = ops.density_grid_filtered(tri, mlp, R, 0.0, coarse="fp16", mark_all=True)
    s = ops.filter_stats(st)
    assert s["n_refined"] == R ** 3 == s["n_marked"] == s["n_first"] and s["n_second"] == 0
    assert torch.equal(vol.view(torch.int32), full.view(torch.int32))
    err = (torch.log(coarse_only.double()) - torch.log(full.double())).abs().max().item()
    assert abs(err - s["max_err"]) <= 1e-5 + 1e-3 * err, (err, s)
    assert 0 < err < 0.2   # one fp16 product per layer: ~1e-2 in log density


def test_refined_set_from_signs_alone_matches_a_host_restatement(cuda):
    """Pass B with a margin that marks nothing: the refined points are exactly the values marching cubes reads in the COARSE
    volume of pass A -- end points of its sign-changing lattice edges and all corners of its cells with an ambiguous sign pattern
    (tests/_mcneeds.py, from the case table) -- counted here on the host, equal to the statistics; the possibly active cells are
    the cells whose coarse corner signs differ."""
    from sculptmate_amd import ops

    R = 48
    tri, mlp, _, _ = _field(cuda, 25, inside=0.15)
    coarse, _ = ops.density_grid_filtered(tri, mlp, R, 1e-30, out_add=-THR, coarse="bf16", passes="A")
    coarse = coarse.clone()
    full = ops.density_grid(tri, mlp, R, out_add=-THR, precision="bf16l3").clone()
    vol, st = ops.density_grid_filtered(tri, mlp, R, 1e-30, out_add=-THR, coarse="bf16")
    s = ops.filter_stats(st)
    need, n_active = _needed(coarse, R)
    assert s["n_marked"] == 0 == s["n_first"] and s["n_cells"] == n_active and s["n_refined"] == s["n_second"] == int(need.sum())
    # ... and pass C rewrote exactly those points
    assert torch.equal(vol.view(torch.int32)[need], full.view(torch.int32)[need])
    assert torch.equal(vol.view(torch.int32)[~need], coarse.view(torch.int32)[~need])


def test_passes_one_by_one_equal_the_single_call(cuda):
    from sculptmate_amd import ops

    R = 64
    tri, mlp, _, _ = _field(cuda, 27, inside=0.1)
    margin, _ = _margin(tri, mlp, "fp16")
    a, st = ops.density_grid_filtered(tri, mlp, R, margin, out_add=-THR)
    a, sa = a.clone(), ops.filter_stats(st)
    out = torch.empty_like(a)
    for k, p in enumerate("ABC"):
        _, st = ops.density_grid_filtered(tri, mlp, R, margin, out_add=-THR, out=out, passes=p, tables=(k == 0))
    assert torch.equal(out.view(torch.int32), a.view(torch.int32)) and ops.filter_stats(st) == sa


def test_filtered_slab_of_the_lattice(cuda):
    """x_begin / x_end (the slabs of BASELINE config 5): cells inside the slab only, same identity against the full slab."""
    from sculptmate_amd import ops

    R, x0, x1 = 64, 19, 41
    tri, mlp, _, _ = _field(cuda, 29, inside=0.1)
    margin, _ = _margin(tri, mlp, "fp16")
    full = ops.density_grid(tri, mlp, R, out_add=-THR, precision="bf16l3", x_begin=x0, x_end=x1).clone()
    vol, st = ops.density_grid_filtered(tri, mlp, R, margin, out_add=-THR, x_begin=x0, x_end=x1)
    assert ops.filter_stats(st)["n_points"] == (x1 - x0) * R * R
    _assert_same_for_marching_cubes(vol, full, R, nx=x1 - x0)


def test_filtered_grid_at_full_size_over_thresholds(cuda):
    """256^3 (BASELINE config 2's grid): one calibration, then a sweep of levels on the same field -- no sign mismatch, every
    corner of every active cell bit-equal, the mesh (vertices, faces, order) equal to the full evaluation's."""
    from sculptmate_amd import ops

    R = 256
    tri, mlp, _, _ = _field(cuda, 13, inside=0.015)
    margin, probe_err = _margin(tri, mlp, "fp16", probe=64)
    base = ops.density_grid(tri, mlp, R, precision="bf16l3").clone()   # density_act; the level only shifts it
    out = torch.empty_like(base)
    for thr in (25.0, 11.0, 40.0, 3.0):
        full = base - np.float32(thr)
        ref = ops.density_grid(tri, mlp, R, out_add=-thr, precision="bf16l3")
        assert torch.equal(ref.view(torch.int32), full.view(torch.int32))   # exp(.) + out_add is one fp32 add
        vol, st = ops.density_grid_filtered(tri, mlp, R, margin, out_add=-thr, out=out)
        s = ops.filter_stats(st)
        _assert_same_for_marching_cubes(vol, full, R)
        assert s["max_err"] <= margin / 3.0 and s["max_err"] <= 3.0 * probe_err, (s, margin, probe_err)
        assert s["n_refined"] < 0.5 * R ** 3
        assert _same_mesh(ops.marching_cubes(vol.view(R, R, R), 0.0), ops.marching_cubes(full.view(R, R, R), 0.0))


def test_statistics_through_the_c_entry_point(cuda):
    """sculpt_density_filter_stats (device -> host copy + wait, for callers without torch) returns the words of the header."""
    import ctypes

    from sculptmate_amd import _lib, ops

    tri, mlp, _, _ = _field(cuda, 21, inside=0.1)
    margin, _ = _margin(tri, mlp, "fp16")
    _, st = ops.density_grid_filtered(tri, mlp, 40, margin, out_add=-THR)
    want = st.cpu().numpy()
    fws = ops._ws_cache[("dgf", tri.device)]
    got = np.zeros(8, np.int32)
    _lib.check(_lib.lib.sculpt_density_filter_stats(ctypes.c_void_p(fws.data_ptr()), ctypes.c_void_p(got.ctypes.data),
                                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    assert np.array_equal(got, want) and got[5] == 40 ** 3 and 0 < got[0] < 40 ** 3


def test_bad_arguments_are_refused(cuda):
    from sculptmate_amd import _lib, ops

    tri, mlp, _, _ = _field(cuda, 21, inside=0.1)
    with pytest.raises(_lib.SculptError):
        ops.density_grid_filtered(tri, mlp, 32, 0.1, out_add=0.0)       # no positive level
    with pytest.raises(_lib.SculptError):
        ops.density_grid_filtered(tri, mlp, 32, 0.0, out_add=-THR)      # no margin
    with pytest.raises(_lib.SculptError):
        ops.density_grid_filtered(tri, mlp, 32, float("inf"), out_add=-THR)
    with pytest.raises(_lib.SculptError):
        ops.density_grid_filtered(tri, mlp, 32, 0.1, out_add=-THR, coarse="fp8")
    _, mlp0, _, _ = _field(cuda, 21, inside=0.1, n_hidden_layers=1)     # no 64x64 hidden layer: nothing to do coarsely
    with pytest.raises(_lib.SculptError):
        ops.density_grid_filtered(tri, mlp0, 32, 0.1, out_add=-THR)


# ------------------------------------------------------------------------------------------------- through TSR.extract_meshes
def _small_tsr(cuda, sd, **kw):
    from sculptmate_amd.tsr import TSR
    from sculptmate_amd.tsr.spec import SMALL_CFG

    m = TSR(SMALL_CFG, **kw)
    m.load_state_dict(sd)
    return m.to(cuda)


def _small_scene(cuda, seed=5):
    return torch.from_numpy(synth.smooth_triplane(seed=seed, size=16, scale=2.0)).to(cuda)[None]


def test_tsr_extract_meshes_filtered_equals_unfiltered(cuda):
    from sculptmate_amd import ops
    from sculptmate_amd.tsr.spec import SMALL_CFG

    sd = synth.tsr_state(3, SMALL_CFG)
    planes = _small_scene(cuda)
    a = _small_tsr(cuda, sd, decoder_filter=False)
    b = _small_tsr(cuda, sd)
    assert b.decoder_filter and not a.decoder_filter
    dens = ops.density_grid(planes[0], a.decoder, 48, precision="bf16l3")
    for q in (0.9, 0.7, 0.97):
        thr = float(np.quantile(dens.cpu().numpy(), q))
        ma = a.extract_meshes(planes, resolution=48, threshold=thr, enable_texture=True)[0]
        mb = b.extract_meshes(planes, resolution=48, threshold=thr, enable_texture=True)[0]
        assert torch.equal(ma.faces, mb.faces) and torch.equal(ma.vertices.view(torch.int32), mb.vertices.view(torch.int32))
        assert torch.equal(ma.vertex_colors.view(torch.int32), mb.vertex_colors.view(torch.int32))
    info = b.filter_info
    assert info["filtered"] == 3 and info["fallbacks"] == 0 and info["calibrations"] == 1 and info["coarse"] == "fp16"
    assert info["last"]["max_err"] <= info["margin"] / 3 and 0 < info["last"]["n_refined"] < 48 ** 3
    assert a.filter_info["filtered"] == 0


def test_tsr_guard_redoes_the_grid_in_full(cuda):
    """A margin far below the coarse error (as if the calibration had been done on an unrepresentative scene): the guard sees it
    at the re-evaluated points, the grid is redone with the full kernel (same mesh as the unfiltered model) and the margin is
    re-calibrated to cover what was seen."""
    from sculptmate_amd import ops
    from sculptmate_amd.tsr.spec import SMALL_CFG

    sd = synth.tsr_state(3, SMALL_CFG)
    planes = _small_scene(cuda)
    a = _small_tsr(cuda, sd, decoder_filter=False)
    b = _small_tsr(cuda, sd)
    thr = float(np.quantile(ops.density_grid(planes[0], a.decoder, 48, precision="bf16l3").cpu().numpy(), 0.9))
    b.calibrate_decoder_filter(planes[0])
    good = b.filter_info["margin"]
    b.filter_info["margin"] = good / 400.0
    ma = a.extract_meshes(planes, resolution=48, threshold=thr)[0]
    mb = b.extract_meshes(planes, resolution=48, threshold=thr)[0]
    assert torch.equal(ma.faces, mb.faces) and torch.equal(ma.vertices.view(torch.int32), mb.vertices.view(torch.int32))
    assert b.filter_info["fallbacks"] == 1 and b.filter_info["filtered"] == 0
    assert b.filter_info["margin"] >= 0.5 * good
    mb = b.extract_meshes(planes, resolution=48, threshold=thr)[0]     # and the next call is filtered again
    assert b.filter_info["filtered"] == 1 and torch.equal(ma.faces, mb.faces)


def test_tsr_filter_with_activations_beyond_the_fp16_range(cuda):
    """Hidden activations of ~1e6: the IEEE-half coarse pass overflows on the probe, so the calibration takes bf16 operands (fp32
    exponent range) or switches the filter off -- either way the mesh is the unfiltered model's."""
    from sculptmate_amd import ops
    from sculptmate_amd.tsr.spec import SMALL_CFG

    sd = synth.tsr_state(3, SMALL_CFG)
    sd["decoder.layers.4.weight"] = (sd["decoder.layers.4.weight"] * np.float32(1e6)).astype(np.float32)
    sd["decoder.layers.4.bias"] = (sd["decoder.layers.4.bias"] * np.float32(1e6)).astype(np.float32)
    sd["decoder.layers.6.weight"] = (sd["decoder.layers.6.weight"] * np.float32(1e-6)).astype(np.float32)
    planes = _small_scene(cuda)
    a = _small_tsr(cuda, sd, decoder_filter=False)
    b = _small_tsr(cuda, sd)
    thr = float(np.quantile(ops.density_grid(planes[0], a.decoder, 32, precision="bf16l3").cpu().numpy(), 0.9))
    ma = a.extract_meshes(planes, resolution=32, threshold=thr)[0]
    mb = b.extract_meshes(planes, resolution=32, threshold=thr)[0]
    assert torch.equal(ma.faces, mb.faces) and torch.equal(ma.vertices.view(torch.int32), mb.vertices.view(torch.int32))
    assert b.filter_info["coarse"] == "bf16" or not b.filter_info["usable"]
    assert b.filter_info["fallbacks"] == 0


@pytest.mark.parametrize("world", [1, 3, 8])
def test_tsr_slabs_through_the_filtered_grid_equal_the_full_evaluation(cuda, world):
    """BASELINE config 5's per-rank work (TSR.extract_mesh_sharded's slabs, emulated one after the other) with every slab through
    the two-pass grid: the assembled mesh equals the one from fully evaluated slabs bit for bit, every slab under the guard."""
    from sculptmate_amd import ops, slab
    from sculptmate_amd.tsr.spec import SMALL_CFG

    R = 72
    sd = synth.tsr_state(3, SMALL_CFG)
    planes = _small_scene(cuda)[0]
    m = _small_tsr(cuda, sd)
    thr = float(np.quantile(ops.density_grid(planes, m.decoder, 48, precision="bf16l3").cpu().numpy(), 0.9))
    cfg = m.renderer.cfg
    kw = dict(radius=cfg.radius, density_bias=cfg.density_bias, threshold=thr)
    dkw = dict(radius=cfg.radius, density_bias=cfg.density_bias, out_add=-thr)
    fv, ff = slab.extract_mesh_slabs_local(planes, m.decoder, R, world, **kw)
    v, f = slab.extract_mesh_slabs_local(planes, m.decoder, R, world,
                                         run=lambda x0, x1, mc: m._extract_filtered(planes, R, mc, dkw, None, x0, x1), **kw)
    assert torch.equal(f, ff) and torch.equal(v.view(torch.int32), fv.view(torch.int32)) and v.shape[0] > 1000
    assert m.filter_info["filtered"] == world and m.filter_info["fallbacks"] == 0
    # ... and through the model's own entry point (no process group: one slab)
    mesh = m.extract_mesh_sharded(planes, resolution=R, threshold=thr)
    m2 = _small_tsr(cuda, sd, decoder_filter=False)
    ref = m2.extract_mesh_sharded(planes, resolution=R, threshold=thr)
    assert torch.equal(mesh.faces, ref.faces) and torch.equal(mesh.vertices.view(torch.int32), ref.vertices.view(torch.int32))
    assert m.filter_info["filtered"] == world + 1 and m2.filter_info["filtered"] == 0


def test_tsr_filter_leaves_other_decoder_modes_alone(cuda):
    from sculptmate_amd.tsr.spec import SMALL_CFG

    sd = synth.tsr_state(3, SMALL_CFG)
    from sculptmate_amd import ops

    m = _small_tsr(cuda, sd, decoder_precision="fp32")
    planes = _small_scene(cuda)
    thr = float(np.quantile(ops.density_grid(planes[0], m.decoder, 32).cpu().numpy(), 0.9))
    m.extract_meshes(planes, resolution=32, threshold=thr)
    assert m.filter_info["filtered"] == 0 and m.filter_info["calibrations"] == 0
