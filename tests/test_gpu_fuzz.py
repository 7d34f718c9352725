"""Randomised shape sweeps on the GPU (seeded, no external state): ragged sizes, tiny sizes, odd aspect ratios."""
import math

import numpy as np
import pytest
import torch

from oracle import capi

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def test_marching_cubes_many_random_volumes_vs_oracle(cuda):
    from sculptmate_amd import ops

    rng = np.random.default_rng(2024)
    for it in range(60):
        shape = tuple(int(x) for x in rng.integers(2, 40, 3))
        kind = it % 3
        if kind == 0:
            vol = rng.standard_normal(shape).astype(np.float32)
        elif kind == 1:
            vol = rng.integers(-2, 3, shape).astype(np.float32)      # exact zeros, degenerate saddles
        else:
            g = [np.linspace(-1, 1, n) for n in shape]
            x, y, z = np.meshgrid(*g, indexing="ij")
            vol = (rng.uniform(0.3, 0.9) - np.sqrt(x * x + y * y + z * z) + 0.05 * rng.standard_normal(shape)).astype(np.float32)
        level = float(rng.choice([0.0, 0.0, 0.1, -0.25]))
        try:
            rv, rf = capi.marching_cubes(vol, level)
        except (ValueError, RuntimeError) as e:
            with pytest.raises(type(e) if not isinstance(e, capi.MCError) else ValueError):
                ops.marching_cubes(torch.from_numpy(vol).to(cuda), level)
            continue
        v, f = ops.marching_cubes(torch.from_numpy(vol).to(cuda), level)
        assert np.array_equal(f.cpu().numpy(), rf), (it, shape)
        assert np.array_equal(v.cpu().numpy().view(np.uint32), rv.view(np.uint32)), (it, shape)


def test_slab_assembly_random_partitions(cuda):
    from sculptmate_amd import ops, slab

    rng = np.random.default_rng(7)
    for it in range(12):
        shape = (int(rng.integers(3, 30)), int(rng.integers(2, 20)), int(rng.integers(2, 20)))
        world = int(rng.integers(1, 9))
        vol = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).to(cuda)
        parts = []
        for (c0, c1) in slab.slab_ranges(shape[0], world):
            if c1 <= c0:
                continue
            v, f, top, mm = ops.marching_cubes(vol[c0:c1 + 1].contiguous(), 0.0, reference_order=True,
                                               slab=dict(axis0_offset=c0, halo_low=c0 > 0))
            parts.append(dict(verts=v, faces=f, top=top, minmax=mm))
        v, f = slab.assemble(parts)
        fv, ff = ops.marching_cubes(vol, 0.0, reference_order=True)
        assert torch.equal(v, fv) and torch.equal(f, ff), (shape, world)


def test_gemm_random_shapes(cuda):
    from sculptmate_amd import _lib, ops

    rng = np.random.default_rng(3)
    g = torch.Generator().manual_seed(3)
    for it in range(34):
        M = int(rng.integers(1, 700))
        N = 128 * int(rng.integers(1, 9))
        K = 64 * int(rng.integers(1, 17))
        if it >= 24:  # launches of more tiles than CUs: the 128-row tiles (the small shapes above take the 64-row form)
            M, N = int(rng.integers(2000, 3300)), 128 * int(rng.choice([8, 16, 24]))
        A = torch.randn(M, K, generator=g).to(BF)
        W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF)
        b = torch.randn(N, generator=g)
        ref = A.float() @ W.float().t() + b
        out = torch.empty(M, N, device=cuda)
        split = int(rng.choice([0, N // 2])) if N >= 256 else 0
        Mp = ((M + 63) // 64) * 64
        outt = torch.zeros(N - split if split else N, Mp, dtype=BF, device=cuda)
        ops.gemm(A.to(cuda), W.to(cuda), bias=b.to(cuda), out_f32=out, out_t=outt, n_split=split)
        lim = split if split else N
        err = (out[:, :lim].cpu() - ref[:, :lim]).abs().max()
        assert err < 2e-3, (M, N, K, float(err))
        tref = ref[:, split:] if split else ref
        assert (outt[:, :M].t().float().cpu() - tref).norm() / tref.norm() < 5e-3


def test_attention_random_shapes(cuda):
    from sculptmate_amd import ops

    rng = np.random.default_rng(5)
    g = torch.Generator().manual_seed(5)
    for it in range(28):
        Tq, Tk, heads = int(rng.integers(1, 400)), int(rng.integers(1, 400)), int(rng.integers(1, 5))
        if it >= 16:  # the launch shapes that pick 192- and 256-query workgroups (every third with many queries)
            Tq, heads = int(rng.integers(1500, 3200)) if it % 3 else int(rng.integers(5000, 9000)), int(rng.integers(8, 17))
        D = heads * 64
        q = torch.randn(Tq, D, generator=g).to(BF)
        k = torch.randn(Tk, D, generator=g).to(BF)
        v = torch.randn(Tk, D, generator=g).to(BF)
        vt = torch.zeros(D, ((Tk + 63) // 64) * 64, dtype=BF)
        vt[:, :Tk] = v.t()
        o = torch.empty(Tq, D, dtype=BF, device=cuda)
        if it % 2:  # the pre-scaled entry (scale = 0): q carries softmax_scale * log2(e), rounded to bf16 once
            q = (q.float() * (0.125 * 1.4426950408889634)).to(BF)
            ops.attention(q.to(cuda), k.to(cuda), vt.to(cuda), o, Tq, Tk, heads, None)
            q = (q.float() / (0.125 * 1.4426950408889634))  # the reference sees the same rounded queries
        else:
            ops.attention(q.to(cuda), k.to(cuda), vt.to(cuda), o, Tq, Tk, heads, 0.125)
        qh = q.float().view(Tq, heads, 64).transpose(0, 1)
        kh = k.float().view(Tk, heads, 64).transpose(0, 1)
        vh = v.float().view(Tk, heads, 64).transpose(0, 1)
        ref = (torch.softmax(qh @ kh.transpose(1, 2) * 0.125, -1) @ vh).transpose(0, 1).reshape(Tq, D)
        assert (o.float().cpu() - ref).abs().max() < 0.06, (Tq, Tk, heads)


def test_density_grid_random_resolutions(cuda):
    from sculptmate_amd import ops, synth

    Ws, bs = synth.decoder_lists(synth.decoder_state(seed=81))
    mlp = ops.PackedMLP(Ws, bs, cuda)
    tri_np = synth.smooth_triplane(seed=82, scale=3.0)
    tri = torch.from_numpy(tri_np).to(cuda)
    for R in (2, 3, 17, 31, 32, 45):
        ref = capi.density_grid(tri_np, Ws, bs, R)
        out = ops.density_grid(tri, mlp, R).cpu().numpy()
        assert np.abs(np.log(out) - np.log(ref)).max() < 5e-5, R


def test_attention_repeated_launches_are_bit_identical(cuda):
    """Race screen for the pipelined attention loop (tile DMA issued through inline asm, ring slots refilled one iteration ahead,
    one barrier per iteration): the same launch repeated must give the same bits every time, while a second stream perturbs the
    timing.  A stale tile, a slot overwritten early or a missed wait shows up as run-to-run differences (tools/stress_attention.py
    is the long form)."""
    from sculptmate_amd import ops

    rng = np.random.default_rng(3)
    g = torch.Generator().manual_seed(3)
    noise_stream = torch.cuda.Stream(cuda)
    na = torch.randn(1024, 1024, device=cuda)
    shapes = [(3072, 3072, 16), (3072, 1025, 16), (1025, 1025, 12), (3072, 129, 16), (3072, 64, 16), (200, 130, 4)]
    shapes += [(int(rng.integers(1, 4000)), int(rng.integers(1, 3000)), int(rng.integers(1, 17))) for _ in range(4)]
    for (Tq, Tk, heads) in shapes:
        D = heads * 64
        q = (torch.randn(Tq, D, generator=g) * 0.18).to(BF).to(cuda)
        k = torch.randn(Tk, D, generator=g).to(BF).to(cuda)
        vt = torch.zeros(D, ((Tk + 63) // 64) * 64, dtype=BF, device=cuda)
        vt[:, :Tk] = torch.randn(D, Tk, generator=g).to(BF).to(cuda)
        o = torch.empty(Tq, D, dtype=BF, device=cuda)
        ops.attention(q, k, vt, o, Tq, Tk, heads, None)
        torch.cuda.synchronize()
        ref = o.clone()
        for it in range(40):
            if it % 3 == 0:
                with torch.cuda.stream(noise_stream):
                    torch.mm(na, na)
            o.fill_(float("nan"))
            ops.attention(q, k, vt, o, Tq, Tk, heads, None)
            assert torch.equal(o.view(torch.int16), ref.view(torch.int16)), (Tq, Tk, heads, it)
        torch.cuda.synchronize()
