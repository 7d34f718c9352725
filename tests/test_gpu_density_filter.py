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
    assert s["n_marked"] == s["n_first"] <= s["n_refined"] == s["n_first"] + s["n_second"] + s["n_audit"] and s["n_nonfinite"] == 0
    assert ops.filter_guard_error(s) <= margin / 3.0, (s, margin)   # the run-time guard TSR applies
    assert s["n_mismatch"] == 0 and 0 < s["n_audit"] < 0.0055 * R ** 3
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
    assert s["n_refined"] == R ** 3 == s["n_marked"] == s["n_first"] and s["n_second"] == 0 == s["n_audit"]
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
    assert s["n_marked"] == 0 == s["n_first"] and s["n_cells"] == n_active
    assert s["n_second"] == int(need.sum()) and s["n_refined"] == s["n_second"] + s["n_audit"]
    # ... and pass C rewrote exactly those points, plus its audit sample of the others
    assert torch.equal(vol.view(torch.int32)[need], full.view(torch.int32)[need])
    touched = (vol.view(torch.int32) != coarse.view(torch.int32)) & ~need
    assert int(touched.sum()) <= s["n_audit"] and torch.equal(vol.view(torch.int32)[touched], full.view(torch.int32)[touched])
    # (an audit point whose coarse and exact bits agree is not counted by `touched`; there are few of those)
    assert int(touched.sum()) >= 0.9 * s["n_audit"]


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
        assert ops.filter_guard_error(s) <= margin / 3.0 and s["max_err"] <= 3.0 * probe_err, (s, margin, probe_err)
        # the audit sample: ~0.5 % of the lattice (one candidate in 16 % of the z words), minus the candidates that are refined anyway
        assert 0.003 * R ** 3 < s["n_audit"] < 0.0052 * R ** 3 and s["n_mismatch"] == 0
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
    got = np.zeros(ops.FILTER_STATS_WORDS, np.int32)
    assert want.shape == got.shape == (12,)
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



# ------------------------------------------------------------------------------------------------- the run-time guard
def _planted(cuda, seed, A_of, inside=0.15, deep=0.03, R=48):
    """A decoder whose density is d = u . a (a: the activations entering the last hidden layer, u > 0) plus a pair of identical
    neurons that fire only where d is in its top `deep` quantile (deep inside the object) and enter the density with weights
    +A, -A: they cancel in every evaluation.  In the PLANTED model the IEEE-half copy of the second neuron's weight row -- what
    pass A multiplies by -- is zero, so the coarse pass alone is off by A silu(g (d - T)) there, and nowhere else.
    -> (planes, clean PackedMLP, planted PackedMLP, thr, facts)"""
    from sculptmate_amd import ops

    rng = np.random.default_rng([seed, 77])
    Ws, bs = synth.decoder_lists(synth.decoder_state(seed=seed))
    Ws, bs = [w.copy() for w in Ws], [b.copy() for b in bs]
    tri_np = synth.smooth_triplane(seed=seed + 1, scale=3.0)
    tri = torch.from_numpy(tri_np).to(cuda)
    u = (np.abs(rng.standard_normal(64)) * 0.3).astype(np.float32)
    W, b, WL, bL = Ws[-2], bs[-2], Ws[-1], bs[-1]
    W[0], W[1], W[2], W[3] = u, -u, 0.0, 0.0
    b[0:4] = 0.0
    WL[0] = 0.0
    WL[0, 0], WL[0, 1] = 1.0, -1.0     # silu(x) - silu(-x) = x: the density row reads u . a
    bL[0] = 0.0
    probe = ops.density_grid(tri, ops.PackedMLP(Ws, bs, cuda), R, density_bias=0.0, precision="bf16l3")
    s = torch.log(probe).double().cpu().numpy()
    level, T, hi = np.quantile(s, 1.0 - inside), np.quantile(s, 1.0 - deep), np.quantile(s, 1.0 - deep / 6)
    assert T > 0 and hi > T > level
    g = np.float32(max(25.0 / T, 6.0 / (hi - T), 20.0 / (T - level)))
    A = np.float32(A_of(level, T, hi))
    W[2] = W[3] = g * u
    b[2] = b[3] = -g * np.float32(T)
    WL[0, 2], WL[0, 3] = A, -A
    clean = ops.PackedMLP(Ws, bs, cuda)
    W_bad = [w.copy() for w in Ws]
    W_bad[-2][3] = 0.0
    bad = ops.PackedMLP(W_bad, bs, cuda)
    # the planted model: the clean blob with the leading IEEE-half part of the last hidden layer taken from `bad`
    hd = clean.blob[:16].view(torch.int32).cpu().numpy()
    NH, off_x3h = int(hd[2]), int(hd[12])
    assert NH == len(Ws) - 2 and bad.blob[:16].view(torch.int32).cpu().numpy()[12] == off_x3h
    lo = off_x3h + (NH - 1) * 4096
    planted = ops.PackedMLP(Ws, bs, cuda)
    assert torch.equal(planted.blob, clean.blob)
    planted.blob[lo:lo + 2048] = bad.blob[lo:lo + 2048]
    assert not torch.equal(planted.blob, clean.blob)
    thr = float(np.exp(level))      # density_bias = 0 in these tests
    return tri, clean, planted, thr, dict(level=level, T=T, hi=hi, g=float(g), A=float(A), s=s)


def _far_and_near_errors(tri, planted, R, thr, margin):
    """Largest coarse error |log d~ - log d| of the planted model near the level (what the round-5 guard looked at) and far."""
    from sculptmate_amd import ops

    full = ops.density_grid(tri, planted, R, density_bias=0.0, precision="bf16l3").clone()
    coarse, _ = ops.density_grid_filtered(tri, planted, R, 0.0, density_bias=0.0, coarse="fp16", mark_all=True, passes="A")
    err = (torch.log(coarse.double()) - torch.log(full.double())).abs()
    dist = (torch.log(full.double()) - np.log(thr)).abs()
    return float(err[dist < 2 * margin].max()), float(err[dist >= 2 * margin].max()), full


def test_guard_audit_sees_a_coarse_error_planted_far_from_the_level(cuda):
    """A coarse error of ~1 in log density in the deepest 3 % of the object, none within reach of the level, no sign changed: the
    points marching cubes reads are all fine (what the guard of round 5 measured), the audit sample is what reports it."""
    from sculptmate_amd import ops

    R = 96
    tri, clean, planted, thr, f = _planted(cuda, 41, lambda level, T, hi: 0.2)
    _, st = ops.density_grid_filtered(tri, clean, 32, 0.0, density_bias=0.0, out_add=0.0, coarse="fp16", mark_all=True)
    margin = max(8.0 * ops.filter_stats(st)["max_err"], 1e-3)
    near, far, full = _far_and_near_errors(tri, planted, R, thr, margin)
    assert near <= margin / 3 and far > 10 * margin, (near, far, margin)
    # the clean model passes ...
    vol, st = ops.density_grid_filtered(tri, clean, R, margin, density_bias=0.0, out_add=-thr)
    s = ops.filter_stats(st)
    assert ops.filter_guard_error(s) <= margin / 3 and s["n_mismatch"] == 0
    # ... the planted one is caught by the audit sample alone: no sign is wrong, the re-evaluated points show nothing
    vol, st = ops.density_grid_filtered(tri, planted, R, margin, density_bias=0.0, out_add=-thr)
    s = ops.filter_stats(st)
    assert int(((vol > 0) != (full - np.float32(thr) > 0)).sum()) == 0 and s["n_mismatch"] == 0
    assert 3 * margin < s["audit_err"] <= far * 1.001 + 1e-5, (s, margin, far)
    assert ops.filter_guard_error(s) > margin / 3
    # (the field is steep at this resolution: some end points of sign-changing edges lie deep inside, so the re-evaluated points
    # report the planted error too -- max_err now covers every one of them, not only those within two margins of the level)


def test_guard_counts_unmarked_sign_errors_planted_deep_inside(cuda):
    """The same neuron with a negative weight: the coarse pass puts the deepest part of the object OUTSIDE.  The rim of that hole
    is a set of sign-changing lattice edges of the planes, pass C re-evaluates their end points and finds unmarked points on the
    wrong side: n_mismatch > 0, the call is void whatever the audit saw."""
    from sculptmate_amd import ops

    R = 96
    tri, clean, planted, thr, f = _planted(cuda, 43, lambda level, T, hi: -(hi - level) / 3.0)
    _, st = ops.density_grid_filtered(tri, clean, 32, 0.0, density_bias=0.0, out_add=0.0, coarse="fp16", mark_all=True)
    margin = max(8.0 * ops.filter_stats(st)["max_err"], 1e-3)
    near, far, full = _far_and_near_errors(tri, planted, R, thr, margin)
    assert near <= margin / 3 and far > 10 * margin, (near, far, margin)
    coarse, _ = ops.density_grid_filtered(tri, planted, R, margin, density_bias=0.0, out_add=-thr, passes="A")
    wrong = int(((coarse > 0) != (full - np.float32(thr) > 0)).sum())
    assert wrong > 100          # the hole exists in the coarse volume
    vol, st = ops.density_grid_filtered(tri, planted, R, margin, density_bias=0.0, out_add=-thr)
    s = ops.filter_stats(st)
    assert s["n_mismatch"] > 0 and s["max_err"] == float("inf") and ops.filter_guard_error(s) == float("inf")
    assert s["audit_err"] > 3 * margin      # (the audit sample sees the hole too)
    vol, st = ops.density_grid_filtered(tri, clean, R, margin, density_bias=0.0, out_add=-thr)
    assert ops.filter_stats(st)["n_mismatch"] == 0


def test_guard_trips_on_a_margin_below_the_coarse_error(cuda):
    """A deliberately small margin (a quarter of the largest coarse error of the probe): marked points come out further from the
    level than the margin, or unmarked ones on the other side of it -- either way the call reports it."""
    from sculptmate_amd import ops

    R = 64
    tri, mlp, _, _ = _field(cuda, 21, inside=0.1)
    margin, probe_err = _margin(tri, mlp, "fp16")
    small = probe_err / 4.0
    full = ops.density_grid(tri, mlp, R, out_add=-THR, precision="bf16l3").clone()
    vol, st = ops.density_grid_filtered(tri, mlp, R, small, out_add=-THR)
    s = ops.filter_stats(st)
    assert ops.filter_guard_error(s) > small / 3.0, (s, small)
    # every marked point's error is recorded wherever its exact value lies: at least the errors up to the margin are seen
    assert s["max_err"] > small / 3.0
    # the right margin on the same field: passes, and then the volume is the full evaluation's for marching cubes
    vol, st = ops.density_grid_filtered(tri, mlp, R, margin, out_add=-THR)
    assert ops.filter_guard_error(ops.filter_stats(st)) <= margin / 3.0
    _assert_same_for_marching_cubes(vol, full, R)


def test_sign_planes_view_is_refused_once_the_workspace_is_rewritten(cuda):
    from sculptmate_amd import _lib, ops

    R = 40
    tri, mlp, _, _ = _field(cuda, 21, inside=0.1)
    margin, _ = _margin(tri, mlp, "fp16")
    vol, _ = ops.density_grid_filtered(tri, mlp, R, margin, out_add=-THR)
    signs = ops.filter_sign_planes(R, tri.device)
    ref = ops.marching_cubes(vol.view(R, R, R), 0.0)
    assert _same_mesh(ops.marching_cubes(vol.view(R, R, R), 0.0, sign_planes=signs), ref)
    ops.density_grid_filtered(tri, mlp, 32, 0.0, out_add=0.0, mark_all=True)      # a calibration probe in between
    with pytest.raises(_lib.SculptError):
        ops.marching_cubes(vol.view(R, R, R), 0.0, sign_planes=signs)
    with pytest.raises(_lib.SculptError):
        ops.filter_sign_planes(R, tri.device)                                      # the last call was not an R = 40 grid


def test_tsr_guard_catches_the_planted_error_and_returns_the_full_evaluations_mesh(cuda):
    """Through TSR.extract_meshes: calibrated on the clean decoder, then the planted one (either kind) -- one fallback, the mesh
    of the unfiltered model, bit for bit."""
    from sculptmate_amd.tsr.spec import SMALL_CFG

    sd = synth.tsr_state(3, SMALL_CFG)
    for seed, A_of in ((41, lambda level, T, hi: 0.2), (43, lambda level, T, hi: -(hi - level) / 3.0)):
        tri, clean, planted, thr, f = _planted(cuda, seed, A_of)
        a = _small_tsr(cuda, sd, decoder_filter=False)
        b = _small_tsr(cuda, sd)
        a.renderer.cfg.density_bias = b.renderer.cfg.density_bias = 0.0
        a.decoder, b.decoder = clean, clean
        b.calibrate_decoder_filter(tri)
        assert b.filter_info["usable"] and b.filter_info["coarse"] == "fp16"
        ma = a.extract_meshes(tri[None], resolution=96, threshold=thr)[0]
        mb = b.extract_meshes(tri[None], resolution=96, threshold=thr)[0]
        assert b.filter_info["filtered"] == 1 and b.filter_info["fallbacks"] == 0
        assert torch.equal(ma.faces, mb.faces) and torch.equal(ma.vertices.view(torch.int32), mb.vertices.view(torch.int32))
        b.decoder = planted
        mb = b.extract_meshes(tri[None], resolution=96, threshold=thr)[0]
        assert b.filter_info["fallbacks"] == 1 and b.filter_info["filtered"] == 1, b.filter_info
        assert torch.equal(ma.faces, mb.faces) and torch.equal(ma.vertices.view(torch.int32), mb.vertices.view(torch.int32))
        # the re-calibration on this scene sees the planted error on its probe: the margin now covers it, or the filter is off
        assert (not b.filter_info["usable"]) or b.filter_info["margin"] >= 8 * 0.5 * abs(f["A"])


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


# ------------------------------------------------------------------------------------------------- trained-like weight statistics
@pytest.mark.parametrize("factor,expect", [(30.0, "fp16"), (100.0, "off")])
def test_trained_like_outliers_through_both_16_bit_shortcuts(cuda, factor, expect):
    """synth.tsr_state(outliers=factor): massive-activation channels in both residual streams, heavy-tailed decoder rows (what a
    trained checkpoint has and the initialiser does not; tools/stress_trained_like.py runs the full-size model).  The two-limb
    fp16 transformer stays fp32-equivalent (its operands are scaled per tensor: no range fallback up to x 10^4); the two-pass
    grid keeps IEEE-half operands at x 30 and switches itself off at x 100 (margin beyond FILTER_MAX_MARGIN) -- the mesh is the
    unfiltered model's bit for bit either way."""
    from sculptmate_amd import ops
    from sculptmate_amd.tsr import TSR
    from sculptmate_amd.tsr.spec import SMALL_CFG

    sd = synth.tsr_state(7, SMALL_CFG, outliers=factor)
    img = torch.from_numpy(synth.composite_rgb(synth.image_rgba(seed=400, size=SMALL_CFG["cond_image_size"]))).to(cuda)
    models = {}
    for prec in ("fp16l2", "bf16l3"):
        models[prec] = TSR(SMALL_CFG, pos_embed_mode="scale_factor", precision=prec)
        models[prec].load_state_dict(sd)
        models[prec].to(cuda)
    a, b = models["fp16l2"].forward(img), models["bf16l3"].forward(img)
    assert bool(torch.isfinite(a).all()) and float((a - b).norm() / b.norm()) < 1e-5
    assert models["fp16l2"].range_fallbacks == 0
    nof = _small_tsr(cuda, sd, precision="bf16l3", decoder_filter=False, pos_embed_mode="scale_factor")
    dens = ops.density_grid(b[0].contiguous(), nof.decoder, 48, precision="bf16l3").double().cpu().numpy()
    thr = float(np.quantile(dens, 0.97))
    ma = models["bf16l3"].extract_meshes(b, resolution=80, threshold=thr)[0]
    mb = nof.extract_meshes(b, resolution=80, threshold=thr)[0]
    assert torch.equal(ma.faces, mb.faces) and torch.equal(ma.vertices.view(torch.int32), mb.vertices.view(torch.int32))
    info = models["bf16l3"].filter_info
    if expect == "off":
        assert not info["usable"] and info["filtered"] == 0 and info["fallbacks"] == 0
    else:
        assert info["usable"] and info["coarse"] == expect and info["filtered"] == 1 and info["fallbacks"] == 0
        assert ops.filter_guard_error(info["last"]) <= info["margin"] / 3
