#!/usr/bin/env python
"""bench.py -- meshes/sec of the TripoSR generation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one image -> mesh through the reference's own entry points (TripoSR/generate.py:32-43 ->
tsr/system.py:82-115,171-200): TSR.forward (ViT-B/16 + 16-block triplane transformer, bf16 MFMA)
+ TSR.extract_meshes at 256^3 (fused triplane sample + NeRF-MLP: fp32 arithmetic on the bf16 matrix pipe through an exact
three-limb split of both operands, fp32 accumulate; Lewiner marching cubes).
`value` times those two calls with the image resident in HBM and the mesh left in HBM (the contract's
definition); the extra key "boundary" times TSR.run_async(host image) -> host mesh (H2D of the image,
pinned D2H of vertices / faces on a copy stream under the next image's kernels).
Workload = BASELINE.json configs[1] "TripoSR single image, mc_resolution=256, bf16 on 1xMI355X";
with N GPUs each rank runs the same per-GPU workload on its own images (configs[2]: one image per
GPU, no collectives on the data path) -> "scaling": "weak".
Inputs are resident in HBM when the timed region starts (synthetic 512x512 RGBA composited on grey,
random-init weights of the reference architecture; the decoder's density bias is calibrated so
that ~1.5 % of the voxels are inside the iso-surface).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FLOP_PER_POINT = 82.4e3          # 81 408 MLP + ~960 sampling (ALGORITHMIC: `roofline.algorithmic_*`)
# FLOPs the dominant kernel's MFMA instructions EXECUTE per lattice point (what `roofline.achieved` / `frac` use):
#   bf16l3: 8 hidden layers x 48 v_mfma_f32_32x32x16_bf16 (32 768 FLOP each) per 32 points = six 64x64 products per layer
#   fp32  : 8 hidden layers x 64 v_mfma_f32_32x32x2_f32 (4 096 FLOP each) per 32 points   = one 64x64 product per layer
# (layer 0 is hoisted into the separable plane tables, the last layer is a VALU dot: neither is on the matrix pipe)
EXECUTED_FLOP_PER_POINT = {"bf16l3": 8 * 48 * 32768 / 32, "fp32": 8 * 64 * 4096 / 32}
# the two-pass ("filtered") grid, TSR's default since round 5 (csrc/density_filter.hip): pass A = one 16-bit product per hidden layer at
# EVERY point (8 MFMAs per layer and 32 points), pass C = the six-product arithmetic at the re-evaluated points only
COARSE_FLOP_PER_POINT = 8 * 8 * 32768 / 32
FILTER_KERNEL_NAME = ("density_coarse_kernel<f16> (every lattice point: fused triplane-sum + NeRF-MLP, one fp16 product per hidden layer "
                      "on v_mfma_f32_32x32x16_f16, fp32 accumulate, fp32 SiLU) + filter_cells / filter_points (bit arithmetic) + 2 x "
                      "density_list_l3k_kernel (six exact bf16-limb products at the points near the level, then at the values "
                      "marching cubes reads)")
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32 matrix peak (dense)
PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense bf16 / fp16 matrix peak
KERNEL_NAME = {"bf16l3": "density_grid_l3k_kernel (fused triplane-sum + NeRF-MLP; hidden layers as six exact bf16-limb products on "
                         "v_mfma_f32_32x32x16_bf16, fp32 accumulate)",
               "fp32": "density_grid_kernel (fused triplane-sum + NeRF-MLP, exact fp32 on v_mfma_f32_32x32x2_f32)"}
DTYPE = {"bf16l3": "fp32-equivalent (bf16 3-limb split, 24 bits, fp32 accumulate) density MLP / f32 tables, SiLU, marching cubes / "
                   "bf16 transformer",
         "fp32": "f32 (density MLP + marching cubes) / bf16 (transformer)"}
MC_RES = 256
THRESHOLD = 25.0


def build_model(device, seed):
    from sculptmate_amd import ops, synth
    from sculptmate_amd.tsr import TSR

    sd = synth.tsr_state(seed=seed)
    model = TSR(pos_embed_mode="scale_factor")
    model.load_state_dict(sd)
    model.to(device)
    return model, sd


def calibrate(model, sd, img_dev, inside=0.015):
    """Shift decoder.layers.18.bias[0] so that `inside` of the voxels exceed the threshold."""
    from sculptmate_amd import synth

    return synth.calibrate_tsr_density_bias(model, sd, img_dev, inside, THRESHOLD)


DECODER_PRECISION = "bf16l3"  # TSR's default; --decoder-precision fp32 times the exact-fp32 kernel instead


def one_step(model, img_dev, events=None):
    """TSR.forward + TSR.extract_meshes (the two calls TripoGenerator.generate_mesh makes, generate.py:36-40), image
    resident in HBM, mesh left in HBM."""
    codes = model([img_dev], device=model.device)
    mesh = model.extract_meshes(codes, False, MC_RES, THRESHOLD, density_events=events)[0]
    return mesh.vertices, mesh.faces


def boundary_rate(model, imgs_host, steps):
    """The same work through the host boundary, by the entry north_star names: TSR.run(list of host fp32 HWC images) -> list of
    Mesh with host (NumPy, pinned) arrays.  Inside, per image: H2D of the 3 MB image + ImagePreprocessor + tokenizer on a second
    stream one image ahead (TSR.tokens_async), backbone / density grid / marching cubes on the current stream, the mesh (the
    reference's `.cpu().numpy()`, system.py:200) device -> pinned host on a copy stream under the next image's kernels.
    `meshes_per_s` = images / wall time of run() calls on lists of up to 20 images; `run_async_loop` = the hand-rolled loop of
    TSR.run_async without the look-ahead (round 3's figure); `latency_ms_single_image` = one image alone, nothing to overlap."""
    chunk = max(1, min(steps, 20))
    lists = [[imgs_host[(c * chunk + i) % len(imgs_host)] for i in range(chunk)] for c in range(max(1, steps // chunk))]
    nbytes = 0
    for _ in range(2):   # warm-up: the pinned-buffer pool (a pinned allocation of a 54 MB mesh is a ~1 ms driver call), streams, slots
        model.run(lists[0], MC_RES, THRESHOLD)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_run = 0
    for lst in lists:
        meshes = model.run(lst, MC_RES, THRESHOLD)
        n_run += len(meshes)
        nbytes = meshes[-1].vertices.nbytes + meshes[-1].faces.nbytes
        del meshes
    dt_run = time.perf_counter() - t0
    # the same entry one image per transformer pass (round 5's default: tokenizer look-ahead only).  Since round 6 TSR.run stacks
    # four images per pass by default in the bf16 mode: every GEMM keeps the single-image tile form, so the meshes are those of
    # one-at-a-time calls bit for bit (checked below on the first list)
    serial = None
    try:
        ref = model.run(lists[0][:9], MC_RES, THRESHOLD, batch=1)
        got = model.run(lists[0][:9], MC_RES, THRESHOLD)     # a full stacked pass and a ragged one
        same = all(np.array_equal(a.vertices.view(np.uint32), b.vertices.view(np.uint32)) and np.array_equal(a.faces, b.faces)
                   for a, b in zip(ref, got))
        del ref, got
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nb = 0
        for lst in lists:
            nb += len(model.run(lst, MC_RES, THRESHOLD, batch=1))
        serial = {"entry": "TSR.run(images, batch=1): one image per transformer pass, tokenizer look-ahead",
                  "meshes_per_s": nb / (time.perf_counter() - t0), "default_run_meshes_identical_to_these": bool(same)}
    except Exception as e:
        serial = {"error": "%s: %s" % (type(e).__name__, e)}
    # the hand-rolled loop: one run_async per image, mesh i collected while image i + 1 runs, no tokenizer look-ahead
    prev = model.run_async(imgs_host[0], MC_RES, THRESHOLD)
    for i in range(1, 5):
        cur = model.run_async(imgs_host[i % len(imgs_host)], MC_RES, THRESHOLD)
        prev.result()
        prev = cur
    prev.result()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        cur = model.run_async(imgs_host[i % len(imgs_host)], MC_RES, THRESHOLD)
        prev.result()
        prev = cur
    prev.result()
    dt = time.perf_counter() - t0
    # latency of ONE image with nothing to overlap: host image in -> host mesh out
    lat = []
    for i in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        model.run_async(imgs_host[i % len(imgs_host)], MC_RES, THRESHOLD).result()
        lat.append((time.perf_counter() - t1) * 1e3)
    return {"entry": "TSR.run([host fp32 HWC 512x512 images]) -> host (pinned) vertices + int64 faces per image; lists of %d" % chunk,
            "meshes_per_s": n_run / dt_run, "ms_per_step": dt_run / n_run * 1e3, "images_timed": n_run,
            "images_per_transformer_pass": int(getattr(model, "RUN_BATCH", 1)) if model.precision == "bf16" else 1,
            "one_image_per_pass": serial,
            "run_async_loop": {"entry": "TSR.run_async(image) per image, mesh i collected under image i + 1, no tokenizer look-ahead",
                               "meshes_per_s": steps / dt, "ms_per_step": dt / steps * 1e3},
            "latency_ms_single_image": float(np.median(lat)),
            "h2d_bytes_per_image": int(imgs_host[0].nbytes), "d2h_bytes_per_mesh": int(nbytes)}


def filtered_active(model):
    """True when extract_meshes evaluates the grid in two passes (TSR.decoder_filter, calibrated and usable)."""
    return bool(model.decoder_filter and model.decoder_precision == "bf16l3" and model.filter_info["usable"]
                and model.filter_info["margin"] is not None)


def filter_identity_check(model, img_dev, rounds=5):
    """OUTSIDE the timed region, before it: the two-pass grid of the bench image against the full three-limb evaluation -- lattice
    points whose side of the level differs (must be 0), marching-cubes output equal bit for bit (vertices, faces, order), the
    fraction of points re-evaluated, and the three passes timed one by one (HIP events, median).  `value` uses the filter only
    when this check passes; otherwise the model is switched to the full evaluation and the line says so."""
    from sculptmate_amd import ops

    _, outb = model.encode_image(img_dev)
    planes = model.scene_code(outb)
    cfg = model.renderer.cfg
    kw = dict(radius=cfg.radius, density_bias=cfg.density_bias, out_add=-THRESHOLD)
    info = model.calibrate_decoder_filter(planes)
    res = {"usable": bool(info["usable"]), "coarse_operands": info["coarse"], "margin_log_density": info["margin"],
           "probe_max_err_log_density": info.get("probe_max_err"),
           "margin_rule": "%g x the largest |log d~ - log d| on a %d^3 probe of the scene code; a call is redone in full when ANY "
                          "re-evaluated point or any point of the audit sample (~0.5 %% of the otherwise untouched lattice) shows "
                          "more than %.2f x margin, or an unmarked point's exact sign differs from the coarse one"
                          % (model.FILTER_SAFETY, model.FILTER_PROBE, model.FILTER_GUARD)}
    if not info["usable"]:
        return res
    R = MC_RES
    full = ops.density_grid(planes, model.decoder, R, precision="bf16l3", **kw).clone()
    vol, st = ops.density_grid_filtered(planes, model.decoder, R, info["margin"], coarse=info["coarse"], **kw)
    stt = ops.filter_stats(st)
    mism = int(((vol > 0) != (full > 0)).sum())
    mca = ops.marching_cubes(full.view(R, R, R), 0.0, reference_order=True, vert_div=R - 1.0, vert_mul=2 * cfg.radius, vert_add=-cfg.radius)
    mcb = ops.marching_cubes(vol.view(R, R, R), 0.0, reference_order=True, vert_div=R - 1.0, vert_mul=2 * cfg.radius, vert_add=-cfg.radius)
    same = (mca[0].shape == mcb[0].shape and mca[1].shape == mcb[1].shape
            and bool(torch.equal(mca[0].view(torch.int32), mcb[0].view(torch.int32))) and bool(torch.equal(mca[1], mcb[1])))
    times = {"A": [], "B": [], "C": [], "ABC": [], "full": []}
    for rnd in range(rounds + 1):
        evs = {}
        for k, p in enumerate(("A", "B", "C")):
            evs[p] = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ops.density_grid_filtered(planes, model.decoder, R, info["margin"], coarse=info["coarse"], out=vol, passes=p,
                                      tables=(k == 0), events=evs[p], **kw)
        evs["ABC"] = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ops.density_grid_filtered(planes, model.decoder, R, info["margin"], coarse=info["coarse"], out=vol, events=evs["ABC"], **kw)
        evs["full"] = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ops.density_grid(planes, model.decoder, R, precision="bf16l3", out=full, events=evs["full"], **kw)
        torch.cuda.synchronize()
        if rnd:
            for k, (a, b) in evs.items():
                times[k].append(a.elapsed_time(b))
    med = {k: (float(np.median(v)) if v else None) for k, v in times.items()}
    res.update({"pass_a_ms": med["A"], "pass_b_ms": med["B"], "pass_c_ms": med["C"], "all_passes_ms": med["ABC"],
                "full_evaluation_ms": med["full"],
                "refined_fraction": stt["n_refined"] / stt["n_points"], "marked_fraction": stt["n_marked"] / stt["n_points"],
                "active_cells": stt["n_cells"], "nonfinite_coarse_values": stt["n_nonfinite"],
                "passes": "A = one fp16 product per layer at every point; B = the marked points exactly (every sign certain after it); "
                          "C = the values marching cubes reads (end points of sign-changing edges, corners of ambiguous cells)",
                "guard_max_err_log_density": stt["max_err"], "guard_over_margin": ops.filter_guard_error(stt) / info["margin"],
                "audit": {"points": stt["n_audit"], "fraction_of_lattice": stt["n_audit"] / stt["n_points"],
                          "max_err_log_density": stt["audit_err"], "unmarked_sign_mismatches": stt["n_mismatch"],
                          "marked_signs_corrected_by_pass_b": stt["n_sign_fixed"]},
                "sign_mismatches_vs_full": mism, "mesh_identical": bool(same),
                "mesh": {"vertices": int(mcb[0].shape[0]), "faces": int(mcb[1].shape[0])},
                "checked_on": "256^3 volume of bench image 0, before the timed region"})
    del full, vol
    return res


def full_density_sibling(model, imgs, steps):
    """The same step with TSR(decoder_filter=False): every lattice point through the six-product kernel (round 4's headline path)."""
    keep = model.decoder_filter
    try:
        model.decoder_filter = False
        one_step(model, imgs[0])
        torch.cuda.synchronize()
        n = max(3, min(steps, 10))
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        t0 = time.perf_counter()
        for i in range(n):
            one_step(model, imgs[i % len(imgs)], events=ev[i])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    finally:
        model.decoder_filter = keep
    ex = EXECUTED_FLOP_PER_POINT["bf16l3"] * MC_RES ** 3
    return {"kernel": KERNEL_NAME["bf16l3"], "launch_ms": ms, "bound": "mfma", "executed_flop_per_launch": ex,
            "achieved": ex / (ms * 1e-3) / 1e12, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": ex / (ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, "meshes_per_s": n / dt, "ms_per_step": dt / n * 1e3}


def stage_split(model, img_dev, n=5):
    """ms per stage (HIP events on torch's stream, median of n untimed extra steps): SURVEY 8d's stage split."""
    from sculptmate_amd import ops

    r = model.renderer.cfg.radius
    rows = []
    for _ in range(n):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        e[0].record()
        ctx, _ = model.image_tokens(img_dev)
        e[1].record()
        _, outb = model.backbone_tokens(ctx)
        e[2].record()
        planes = model.scene_code(outb)
        e[3].record()
        if filtered_active(model):
            vol, _ = ops.density_grid_filtered(planes, model.decoder, MC_RES, model.filter_info["margin"], radius=r,
                                               density_bias=model.renderer.cfg.density_bias, out_add=-THRESHOLD,
                                               coarse=model.filter_info["coarse"])
        else:
            vol = ops.density_grid(planes, model.decoder, MC_RES, radius=r, density_bias=model.renderer.cfg.density_bias,
                                   out_add=-THRESHOLD, precision=DECODER_PRECISION)
        e[4].record()
        ops.marching_cubes(vol.view(MC_RES, MC_RES, MC_RES), 0.0, reference_order=True, vert_div=MC_RES - 1.0,
                           vert_mul=r - (-r), vert_add=-r)
        e[5].record()
        torch.cuda.synchronize()
        rows.append([e[i].elapsed_time(e[i + 1]) for i in range(5)])
    med = np.median(np.array(rows), 0)
    return dict(zip(("image_tokenizer", "backbone", "upsample", "density_grid", "marching_cubes"), [round(float(x), 3) for x in med]))


def density_deviation(model, img_dev, mode, R=96):
    """max |density_act(mode) - density_act(fp32 kernel)| / max(|density_act(fp32 kernel)|, 1e-3) over an R^3 lattice of the
    bench's own scene code (the fp32 kernel itself is within 2e-5 of the CPU oracle, tests/test_gpu_triplane.py)."""
    from sculptmate_amd import ops

    _, outb = model.encode_image(img_dev)
    planes = model.scene_code(outb)
    cfg = model.renderer.cfg
    a = ops.density_grid(planes, model.decoder, R, radius=cfg.radius, density_bias=cfg.density_bias).clone()
    b = ops.density_grid(planes, model.decoder, R, radius=cfg.radius, density_bias=cfg.density_bias, precision=mode)
    return float(((a - b).abs() / a.abs().clamp_min(1e-3)).max())


def batched_rates(model, imgs, steps):
    """TSR.forward on B images in ONE pass (the reference's batch dimension, system.py:82-115: every Linear over B x 3072 /
    B x 1025 stacked token rows, attention over B x heads) + extract_meshes of the B scene codes.  `value` stays the strict
    one-image step; this is the throughput form."""
    res = {"entry": "TSR.forward([B images resident in HBM]) as one batched pass + TSR.extract_meshes(codes, 256); forward_ms = HIP "
                    "events around forward() (tokenizer + backbone + upsampler of the B images).  B2 / B4 / B8: every GEMM keeps the "
                    "tile form of the single-image pass (TSR.batch_exact, the default since round 6): scene codes bit-identical to "
                    "one image at a time.  B4_fastest_tiles: batch_exact = False, the tile forms that are fastest for the stacked "
                    "rows (rounds 4-5: scene codes 2.7e-3 from the single-image ones)"}
    keep, keep_exact = model.max_batch, model.batch_exact
    for B, exact in ((2, True), (4, True), (8, True), (4, False)):
        model.max_batch = B   # forward() batches only when asked to (default 1: image by image)
        model.batch_exact = exact
        group = [imgs[i % len(imgs)] for i in range(B)]
        n = max(2, min(steps, 24) // B + 1)
        for _ in range(2):
            model.extract_meshes(model(group, device=model.device), False, MC_RES, THRESHOLD)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        t0 = time.perf_counter()
        for it in range(n):
            ev[it][0].record()
            codes = model(group, device=model.device)
            ev[it][1].record()
            model.extract_meshes(codes, False, MC_RES, THRESHOLD)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        fwd = float(np.median([a.elapsed_time(b) for a, b in ev]))
        res["B%d" % B if exact else "B%d_fastest_tiles" % B] = {"ms_per_image": dt / (n * B) * 1e3, "meshes_per_s": n * B / dt, "transformer_ms_per_image": fwd / B,
                          "forward_ms": fwd, "passes_timed": n,
                          # 2.96 TFLOP per image over the forward time: the transformer's fraction of the bf16 peak
                          "transformer_tflops": 2.96 * B / (fwd * 1e-3), "transformer_frac": 2.96 * B / (fwd * 1e-3) / PEAK_BF16_MFMA_TFLOPS}
    model.max_batch, model.batch_exact = keep, keep_exact
    torch.cuda.empty_cache()
    return res


def parity_mode_extra(sd, model, imgs, imgs_np, cpu_verts, steps, precision="fp16l2"):
    """The tolerance modes: fp32 storage, fp32 norms / softmax, every matrix product of the transformer on the 16-bit matrix pipe
    through limbs with fp32 accumulation -- the modes that meet north_star's 1e-4 vertex tolerance against the fp32 CPU reference.
    precision="bf16l3": both operands of every product (Linears, QK^T, PV) split exactly into three bf16 limbs, six products;
    "fp16l2" (round 5, the faster one): the Linears on two fp16 limbs per operand (22 bits), three products, the rest as bf16l3.
    Timed like `value` (image resident, mesh left in HBM); the 128^3 mesh against the oracle's fp32 CPU mesh of the same image
    (the same comparison as parity.bf16_mesh_vs_fp32_cpu)."""
    from sculptmate_amd.tsr import TSR

    m = TSR(pos_embed_mode="scale_factor", precision=precision, decoder_precision=model.decoder_precision)
    m.load_state_dict(model.state_dict())   # the calibrated density bias included
    m.to(model.device)
    n = max(3, min(steps, 10))
    for _ in range(2):
        one_step(m, imgs[0])
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    t0 = time.perf_counter()
    for i in range(n):
        ev[i][0].record()
        codes = m([imgs[i % len(imgs)]], device=m.device)
        ev[i][1].record()
        m.extract_meshes(codes, False, MC_RES, THRESHOLD)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dtype = ("f32 storage; matrix products: bf16 3-limb split of both operands (24 bits), fp32 accumulate; f32 norms / softmax"
             if precision == "bf16l3" else
             "f32 storage; Linears AND attention products: fp16 2-limb operands (22 bits), 3 products (the image tokenizer's small "
             "attention launches: bf16 3-limb, 6 products); fp32 accumulate; f32 norms / softmax")
    out = {"mode": 'TSR(precision="%s")' % precision, "forward_ms": float(np.median([a.elapsed_time(b) for a, b in ev])),
           "ms_per_step": dt / n * 1e3, "meshes_per_s": n / dt, "dtype": dtype,
           # 2.96 TFLOP algorithmic per image (2.10 in the Linears, 0.86 in the attention products, 0.03 of them the image
           # tokenizer's); bf16l3 executes 6 limb products per product, fp16l2 3 (6 in the tokenizer's attention)
           "transformer_tflops_algorithmic": None, "steps_timed": n}
    out["transformer_tflops_algorithmic"] = 2.96 / (out["forward_ms"] * 1e-3)
    out["mfma_executed_tflops"] = (6 * 2.96 if precision == "bf16l3" else 3 * (2.10 + 0.83) + 6 * 0.03) / (out["forward_ms"] * 1e-3)
    # operands split once (weights at load time, activations by their producers; csrc/gemm_l3p.hip) -- SCULPT_L3_TILE=split: in every GEMM
    out["limbs_once"] = bool(getattr(m, "l3p", False))
    try:   # the same mode with four images per transformer pass (TSR.forward on a list; every attention one launch over batch x heads)
        keep = m.max_batch
        m.max_batch = 4
        group = [imgs[i % len(imgs)] for i in range(4)]
        m(group, device=m.device)
        torch.cuda.synchronize()
        tb = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            m(group, device=m.device)
            b.record()
            torch.cuda.synchronize()
            tb.append(a.elapsed_time(b))
        out["batched4_forward_ms_per_image"] = float(np.median(tb)) / 4
        m.max_batch = keep
    except Exception as e:  # noqa: BLE001
        out["batched4_forward_ms_per_image"] = "%s: %s" % (type(e).__name__, e)
    if cpu_verts is not None:
        gm = m.run_async(imgs_np[0], 128, THRESHOLD).result()
        d = mesh_distance(gm.vertices, cpu_verts, 1.74)
        d["what"] = ("128^3, bench image 0, threshold 25: %s transformer + %s decoder vs the oracle's fp32 CPU path; two-sided "
                     "nearest-vertex distance / 1.74 (north_star: 1e-4)" % (precision, model.decoder_precision))
        out["mesh_vs_fp32_cpu"] = d
    del m
    torch.cuda.empty_cache()
    return out


def fp32_exact_sibling(model, imgs, steps):
    """The exact-fp32 kernel (TSR(decoder_precision="fp32"), the parity mode) beside the default: its launch time by HIP events,
    the fraction of the fp32 matrix peak its MFMA instructions reach, and the whole-step rate with it."""
    keep = model.decoder_precision
    try:
        model.decoder_precision = "fp32"
        one_step(model, imgs[0])
        torch.cuda.synchronize()
        n = max(3, min(steps, 10))
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        t0 = time.perf_counter()
        for i in range(n):
            one_step(model, imgs[i % len(imgs)], events=ev[i])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    finally:
        model.decoder_precision = keep
    ex = EXECUTED_FLOP_PER_POINT["fp32"] * MC_RES ** 3
    return {"kernel": KERNEL_NAME["fp32"], "launch_ms": ms, "bound": "mfma", "executed_flop_per_launch": ex,
            "achieved": ex / (ms * 1e-3) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": ex / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "meshes_per_s": n / dt, "ms_per_step": dt / n * 1e3}


def kernel_parity(model, img_dev):
    """Default-mode volume against the exact-fp32 kernel's on the bench's own 256^3 field: lattice points on different sides of
    the threshold (= cells whose marching-cubes case can differ), max relative deviation of density_act."""
    from sculptmate_amd import ops

    _, outb = model.encode_image(img_dev)
    planes = model.scene_code(outb)
    cfg = model.renderer.cfg
    a = ops.density_grid(planes, model.decoder, MC_RES, radius=cfg.radius, density_bias=cfg.density_bias, out_add=-THRESHOLD).clone()
    b = ops.density_grid(planes, model.decoder, MC_RES, radius=cfg.radius, density_bias=cfg.density_bias, out_add=-THRESHOLD,
                         precision=DECODER_PRECISION)
    flips = torch.nonzero((a > 0) != (b > 0)).reshape(-1)
    rel = float((((a - b).abs()) / (a + THRESHOLD).abs().clamp_min(1e-3)).max())
    return {"field": "256^3 volume of the bench image, default mode vs exact-fp32 kernel",
            "lattice_points_on_different_sides_of_threshold": int(flips.numel()),
            "flipped_points": [[int(i), float(a[i]), float(b[i])] for i in flips[:8]],
            "density_max_rel_dev": rel}


def mesh_distance(v, rv, extent):
    """Two-sided nearest-vertex distances between two meshes as fractions of `extent`."""
    from scipy.spatial import cKDTree

    v, rv = np.asarray(v, np.float64), np.asarray(rv, np.float64)
    d = np.concatenate([cKDTree(rv).query(v)[0], cKDTree(v).query(rv)[0]]) / extent
    return {"max": float(d.max()), "p999": float(np.quantile(d, 0.999)), "p99": float(np.quantile(d, 0.99)), "mean": float(d.mean()),
            "vertices": [int(len(v)), int(len(rv))]}


def _median_time(fn, warm=1, n=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def _physical_cores():
    """Physical cores of the host: distinct (socket, core) pairs of lscpu -p; None when lscpu is not there."""
    import subprocess

    try:
        out = subprocess.run(["lscpu", "-p=SOCKET,CORE"], capture_output=True, text=True, timeout=10).stdout
        pairs = {ln.strip() for ln in out.splitlines() if ln.strip() and not ln.startswith("#")}
        return len(pairs) or None
    except Exception:
        return None


def cpu_baseline(sd, img_np):
    """the oracle (CPU restatement of the reference, kind "port") on this box's host cores,
    configured like BASELINE config 1 -- ONE 512x512 image, mc_resolution=128, fp32 -- every stage measured whole
    (1 warm-up + 3 timed runs, median), all cores; then the metric's own size, mc_resolution=256, with the query and marching
    cubes measured whole once (about 15 s).  Only the single-thread figure is a bounded sample, labelled as an extrapolation.
    Returns (json dict, CPU mesh vertices at 128^3 in scene units) -- the mesh feeds `parity.bf16_mesh_vs_fp32_cpu`."""
    from oracle import capi, tsr_ref
    from sculptmate_amd import synth
    from sculptmate_amd.tsr.spec import DEFAULT_CFG

    ncpu = os.cpu_count() or 1
    cores = min(ncpu, 32)  # more threads than this only oversubscribes torch's CPU GEMMs
    R = 128
    Ws, bs = synth.decoder_lists(sd)
    torch.set_num_threads(cores)
    capi.set_threads(cores)
    ref = {}

    def fwd():
        ref["code"] = tsr_ref.tsr_forward(sd, img_np, DEFAULT_CFG, pos_mode="scale_factor")

    t_fwd = _median_time(fwd)
    planes = ref["code"].numpy()
    dens = {}

    def query():
        dens["d"] = capi.density_grid(planes, Ws, bs, R)

    t_q = _median_time(query)
    level = -(dens["d"] - np.float32(THRESHOLD))   # system.py:184 with the bench's calibrated model: the surface exists at 25
    mesh = {}

    def mc():
        mesh["v"], mesh["f"] = capi.reference_isosurface(level, R)

    t_mc = _median_time(mc)
    cpu_verts = mesh["v"] * np.float32(0.87 - (-0.87)) + np.float32(-0.87)
    total = t_fwd + t_q + t_mc
    # the metric's own size: mc_resolution = 256, query + marching cubes measured whole, once (forward is resolution-independent)
    t0 = time.perf_counter()
    d256 = capi.density_grid(planes, Ws, bs, MC_RES)
    t_q256 = time.perf_counter() - t0
    t0 = time.perf_counter()
    v256, f256 = capi.reference_isosurface(-(d256 - np.float32(THRESHOLD)), MC_RES)
    t_mc256 = time.perf_counter() - t0
    del d256
    # single thread: bounded sample (one of 12 ViT layers incl. patch embedding, one of 16 backbone blocks, 1/16 of the
    # 128^3 lattice), extrapolated by the layer / lattice counts
    torch.set_num_threads(1)
    capi.set_threads(1)
    cfg1 = dict(DEFAULT_CFG, image_tokenizer=dict(DEFAULT_CFG["image_tokenizer"], num_hidden_layers=1))
    ctx = {}
    t0 = time.perf_counter()
    with torch.no_grad():
        ctx["c"] = tsr_ref.vit_forward(sd, img_np, cfg1, "scale_factor")
    t_vit1 = time.perf_counter() - t0
    h = torch.zeros(3 * 32 * 32, 1024)
    t0 = time.perf_counter()
    with torch.no_grad():
        tsr_ref.block_forward(sd, "backbone.transformer_blocks.0.", h, ctx["c"], 16)
    t_blk1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    capi.density_grid(planes, Ws, bs, R, begin=0, end=R ** 3 // 16)
    t_q1 = (time.perf_counter() - t0) * 16
    total_1t = t_vit1 * 12 + t_blk1 * 16 + t_q1 + t_mc
    # BASELINE.md section 3 asks for "all physical cores" beside the 32-thread figure: the same stages on as many threads as
    # lscpu reports physical cores (one warm-up + one timed run each: torch's CPU GEMMs stop scaling long before, see above)
    phys = _physical_cores() or ncpu
    all_cores = None
    if phys != cores:
        torch.set_num_threads(phys)
        capi.set_threads(phys)
        t_fwd_p = _median_time(fwd, warm=1, n=1)
        t_q_p = _median_time(query, warm=0, n=1)
        all_cores = {"value": 1.0 / (t_fwd_p + t_q_p + t_mc), "unit": "meshes/s", "cores": phys, "extrapolated": False,
                     "stages_s": {"forward": t_fwd_p, "query_128": t_q_p, "marching_cubes_128": t_mc},
                     "sample": "config 1 again on every physical core lscpu reports (%d; one timed run per stage); the headline "
                               "cpu_baseline figure keeps the faster of the two thread counts' settings only if it is this one" % phys}
    torch.set_num_threads(cores)
    capi.set_threads(cores)
    out = {"value": 1.0 / total, "unit": "meshes/s", "cores": cores, "kind": "port",
           "sample": "BASELINE config 1 exactly: one 512x512 image, mc_resolution=128, fp32, all stages whole, 1 warm-up + 3 timed "
                     "(median): TSR.forward %.2fs + query_triplane 128^3 %.2fs + marching cubes %.3fs (single thread, like "
                     "scikit-image) = %.2fs on %d threads" % (t_fwd, t_q, t_mc, total, cores),
           "lscpu_logical_cpus": ncpu, "lscpu_physical_cores": phys, "all_physical_cores": all_cores, "ms_per_image": total * 1e3,
           "stages_s": {"forward": t_fwd, "query_128": t_q, "marching_cubes_128": t_mc},
           "mesh_128": {"vertices": int(len(mesh["v"])), "faces": int(len(mesh["f"]))},
           "at_256": {"value": 1.0 / (t_fwd + t_q256 + t_mc256), "unit": "meshes/s", "cores": cores, "extrapolated": False,
                      "stages_s": {"forward": t_fwd, "query_256": t_q256, "marching_cubes_256": t_mc256},
                      "mesh": {"vertices": int(len(v256)), "faces": int(len(f256))},
                      "sample": "the metric's own configuration (mc_resolution=256): query_triplane over 256^3 and marching cubes "
                                "measured whole, one run each, + the forward time above"},
           "single_thread": {"value": 1.0 / total_1t, "unit": "meshes/s", "cores": 1, "extrapolated": True,
                             "sample": "the only extrapolated figure: 1 of 12 ViT layers %.2fs x12 + 1 of 16 backbone blocks %.2fs x16 "
                                       "+ 1/16 of the 128^3 query x16 = %.2fs + marching cubes %.3fs" % (t_vit1, t_blk1, t_q1, t_mc)}}
    return out, cpu_verts


def slab512_extra(model, img_dev, world=8, iters=2):
    """BASELINE config 5 on ONE GPU: the 512^3 grid of the bench's own scene code cut into `world` slabs evaluated one after the
    other here and assembled (what the ranks + the RCCL gather produce; tests/test_gpu_slab.py checks bit equality)."""
    from sculptmate_amd import slab

    codes = model([img_dev], device=model.device)
    planes = codes[0].contiguous()
    cfg = model.renderer.cfg
    kw = dict(radius=cfg.radius, density_bias=cfg.density_bias, threshold=THRESHOLD)
    res = {}
    variants = [("", kw)]
    if filtered_active(model):   # every slab through the two-pass grid, as TSR.extract_mesh_sharded runs it (the default)
        dkw = dict(radius=cfg.radius, density_bias=cfg.density_bias, out_add=-THRESHOLD)
        fkw = dict(kw, run=lambda x0, x1, mc: model._extract_filtered(planes, 512, mc, dkw, None, x0, x1))
        variants = [("", fkw), ("full_evaluation_", kw)]
    meshes = {}
    for tag, k in variants:
        for w in (1, world):
            v, f = slab.extract_mesh_slabs_local(planes, model.decoder, 512, w, **k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                v, f = slab.extract_mesh_slabs_local(planes, model.decoder, 512, w, **k)
            torch.cuda.synchronize()
            res["%sslabs_%d_ms" % (tag, w)] = (time.perf_counter() - t0) / iters * 1e3
            res["vertices"], res["faces"] = int(v.shape[0]), int(f.shape[0])
            meshes[(tag, w)] = (v, f)
    if len(variants) == 2:
        res["decoder_filter"] = True
        res["mesh_identical_to_full_evaluation"] = bool(all(
            torch.equal(meshes[("", w)][0].view(torch.int32), meshes[("full_evaluation_", w)][0].view(torch.int32))
            and torch.equal(meshes[("", w)][1], meshes[("full_evaluation_", w)][1]) for w in (1, world)))
    del meshes
    res["what"] = ("extract_mesh at 512^3 (density grid + marching cubes) on 1 GPU: single pass vs %d slabs one after the other + "
                   "assemble (the per-rank work and the assembly of config 5; the RCCL gather itself needs %d GPUs)" % (world, world))
    torch.cuda.empty_cache()
    return res


def _sf3d_tail_stages(m, mesh, planes, img, device):
    """The rest of BASELINE config 4 (StableFast/sf3d/system.py:308-528 as the add-on calls it, generate.py:33-48: remesh
    'triangle', texture 512): image estimator, triangle remesh (host), UV unwrap, texture bake -- each at full size on the
    Kuhn-grid mesh, except the host remesher, whose cost is linear in the faces and is timed on a 600k-face slab of it."""
    from sculptmate_amd.sf3d.bake import bake_textures
    from sculptmate_amd.sf3d.system import Mesh

    def wall(fn, n=2):
        ts = []
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return min(ts), r

    res = {}
    if m.image_estimator is not None:
        mask = torch.ones(1, img.shape[0], img.shape[1], device=device)
        res["image_estimator"] = round(wall(lambda: m.image_estimator(img[None], mask=mask), 3)[0], 3)
    mm = Mesh(mesh.v_pos.clone(), mesh.t_pos_idx.clone(), unwrapper=m.unwrapper)
    res["uv_unwrap"] = round(wall(lambda: Mesh(mesh.v_pos.clone(), mesh.t_pos_idx.clone(), unwrapper=m.unwrapper).unwrap_uv())[0], 3)
    mm.unwrap_uv()
    res["texture_bake_512"] = round(wall(lambda: bake_textures(m, mm, planes, 512, {}, 0))[0], 3)
    # host remesh on a slab: the first 600k faces, vertices compacted
    nf = min(600_000, mesh.t_pos_idx.shape[0])
    f = mesh.t_pos_idx[:nf]
    used, inv = torch.unique(f.reshape(-1), return_inverse=True)
    slab = Mesh(mesh.v_pos[used].contiguous(), inv.reshape(-1, 3).contiguous())
    t_r, rm = wall(lambda: m.remesher(slab, "triangle", round(0.75 * slab.v_pos.shape[0])), 1)
    res["remesh_host_slab"] = {"ms": round(t_r, 1), "faces_in": int(nf), "vertices_in": int(used.numel()), "vertices_out": int(rm.v_pos.shape[0]),
                               "us_per_face": round(t_r * 1e3 / nf, 2),
                               "full_mesh_extrapolated_s": round(t_r * 1e-3 * mesh.t_pos_idx.shape[0] / nf, 1)}
    gpu_ms = sum(v for k, v in res.items() if isinstance(v, float))
    return {"stages_ms": res, "gpu_tail_ms": round(gpu_ms, 3),
            "what": "tail of config 4 at full size on the Kuhn-grid mesh (5.0 M vertices / 9.6 M faces -- the shipped 160_tets.npz grid is "
                    "absent and would give a mesh several times smaller): CLIP image estimator, box-projection UV unwrap, 512^2 texture "
                    "bake; the triangle remesh runs on the HOST like the reference's gpytoolbox (linear in faces: timed on a 600k-face "
                    "slab, extrapolation labelled)"}


def sf3d_extra(device, n=5):
    """BASELINE config 4: StableFast-3D single image, full-size networks (random init, 834.8 M parameters), stage times by HIP
    events (median of n).  The shipped 160_tets.npz is absent: a Kuhn 6-tets-per-cube grid of the same resolution stands in
    (4.17 M vertices / 24.6 M tets -- denser than the shipped grid, so the mesh stage is an upper bound).  Estimators skipped."""
    from sculptmate_amd import synth
    from sculptmate_amd.sf3d.spec import DEFAULT_CFG as SF_CFG
    from sculptmate_amd.sf3d.system import SF3D

    cfg = dict(SF_CFG)
    sd = synth.sf3d_state(0, cfg)
    sd.update(synth.sf3d_estimator_state(0))   # CLIP image estimator + illumination estimator, like the shipped checkpoint
    m = SF3D(cfg)
    m.load_state_dict(sd)
    m.to(device)
    img = torch.from_numpy(synth.composite_rgb(synth.image_rgba(0, 512))).to(device)
    codes = m.scene_code(img)
    q = m.query_triplane(m._grid_world, codes)
    pre = m.decoder(q, include=["density"])["density"].reshape(-1).log().cpu().numpy()
    shift = np.log(cfg["isosurface_threshold"]) - np.quantile(pre.astype(np.float64), 0.9)
    sd["decoder.heads.density.4.bias"] = (sd["decoder.heads.density.4.bias"] + np.float32(shift)).astype(np.float32)
    m.load_state_dict(sd)
    rows, mesh = [], None
    for it in range(n + 2):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        e[0].record()
        tok = m.image_tokens(img); e[1].record()
        direct = m.backbone_tokens(tok); e[2].record()
        planes = m.post_process(direct); e[3].record()
        mesh = m.triplane_to_meshes(planes[None])[0]; e[4].record()
        torch.cuda.synchronize()
        if it >= 2:
            rows.append([e[i].elapsed_time(e[i + 1]) for i in range(4)])
    t = np.median(np.array(rows), 0)
    full = None
    try:
        full = _sf3d_tail_stages(m, mesh, planes, img, device)
    except Exception as e:  # the tail is informational: never lose the network stages over it
        full = {"error": "%s: %s" % (type(e).__name__, e)}
    out = {"stages_ms": dict(zip(("dinov2", "backbone", "upsampler", "query_mtet"), [round(float(x), 3) for x in t])),
           "ms_per_image": float(t.sum()), "meshes_per_s": 1e3 / float(t.sum()),
           "mesh": {"vertices": int(mesh.v_pos.shape[0]), "faces": int(mesh.t_pos_idx.shape[0])},
           "what": "SF3D image -> mesh (DINOv2-L + two-stream backbone + pixel-shuffle upsampler + density / deformation query + "
                   "marching tetrahedra), bf16 networks, Kuhn tet grid (160_tets.npz absent); the rest of config 4 (estimator, remesh, unwrap, "
                   "bake) is under 'full'",
           "full": full}
    del m, sd
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optional-modes", action="store_true",
                    help="accepted for old command lines: the two-limb decoder modes it used to skip were removed in round 6")
    ap.add_argument("--no-siblings", action="store_true",
                    help="skip the exact-fp32 kernel's sibling measurement and the kernel-parity pass (profiling runs)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra keys measured after the timed region (boundary, slab512, sf3d)")
    ap.add_argument("--decoder-precision", choices=("bf16l3", "fp32"), default="bf16l3",
                    help="bf16l3 (default = TSR's default: fp32-equivalent three-limb bf16 split), fp32 (exact-fp32 MFMA kernel), "
                         "or a two-limb experiment mode")
    ap.add_argument("--check-rounds", type=int, default=5,
                    help="timing rounds of the three passes inside the identity check of the two-pass grid (0: profiling runs, the "
                         "check alone)")
    ap.add_argument("--no-decoder-filter", action="store_true",
                    help="every lattice point through the six-product kernel (TSR(decoder_filter=False), round 4's path) as the headline")
    args = ap.parse_args()
    global DECODER_PRECISION
    DECODER_PRECISION = args.decoder_precision

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d needs %d ranks (WORLD_SIZE=%d): launch with python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node %d --master-addr 127.0.0.1 bench.py --gpus %d ..." % (args.gpus, args.gpus, world, args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    # Test plumbing for 1-GPU boxes (tests/test_gpu_bench_ranks.py): SCULPT_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and
    # SCULPT_BENCH_BACKEND=gloo runs the two collectives of this script on the host -- RCCL refuses two ranks on one device.
    # The driver's multi-GPU runs set neither: one rank per GPU over RCCL.
    share_gpu = bool(os.environ.get("SCULPT_BENCH_SHARE_GPU"))
    backend = os.environ.get("SCULPT_BENCH_BACKEND", "nccl")
    dev_index = 0 if share_gpu else local_rank
    if not share_gpu and local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d has no GPU of its own (%d visible): one rank per GPU" % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    from sculptmate_amd import batch, parallel, synth

    if world > 1:
        batch.cap_host_threads(world)  # N ranks synthesise 1.7 GB of weights on one host at once
    if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
        os.environ["NCCL_DEBUG"] = "WARN"  # keep stdout to the one JSON line (RCCL prints its banner there)

    # SCULPT_FORCE_DIST=1 exercises the RCCL path with a single rank (used to validate it on a 1-GPU box)
    dist = parallel.init(backend, device) if (world > 1 or os.environ.get("SCULPT_FORCE_DIST")) else None

    model, sd = build_model(device, seed=0)  # every rank holds a full replica (no weight sharding)
    model.decoder_precision = DECODER_PRECISION
    imgs_np = [synth.composite_rgb(synth.image_rgba(seed=100 + rank * 8 + i)) for i in range(4)]
    imgs = [torch.from_numpy(a).to(device).contiguous() for a in imgs_np]
    filt = None
    with torch.no_grad():
        calibrate(model, sd, imgs[0])
        if args.no_decoder_filter:
            model.decoder_filter = False
        if model.decoder_filter and DECODER_PRECISION == "bf16l3":
            # the two-pass grid may carry `value` only if it reproduces the full evaluation on the bench's own field
            filt = filter_identity_check(model, imgs[0], rounds=args.check_rounds)
            if not (filt.get("usable") and filt.get("sign_mismatches_vs_full") == 0 and filt.get("mesh_identical")):
                model.decoder_filter = False
                filt["used_for_value"] = False
            else:
                filt["used_for_value"] = True
        for i in range(args.warmup):
            one_step(model, imgs[i % len(imgs)])
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nv = nf = 0
        for i in range(args.steps):
            v, f = one_step(model, imgs[i % len(imgs)], events=ev[i])
            nv, nf = v.shape[0], f.shape[0]
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    elapsed = parallel.max_over_ranks(elapsed, device if backend == "nccl" else "cpu")

    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    if rank == 0:
        mode = DECODER_PRECISION
        use_filter = filtered_active(model)
        if use_filter:
            # executed MFMA FLOPs of BOTH passes: one product per layer at every point + six at the re-evaluated ones (of the last step)
            n_ref = int((model.filter_info["last"] or {}).get("n_refined", 0))
            executed = COARSE_FLOP_PER_POINT * MC_RES ** 3 + EXECUTED_FLOP_PER_POINT["bf16l3"] * n_ref
        else:
            executed = EXECUTED_FLOP_PER_POINT[mode] * MC_RES ** 3
        achieved = executed / (kern_ms * 1e-3) / 1e12
        peak = PEAK_F32_MFMA_TFLOPS if mode == "fp32" else PEAK_BF16_MFMA_TFLOPS
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_density_grid.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                from sculptmate_amd import _lib as _l

                have = _l.lib.sculpt_source_digest().decode()
                if j.get("mode", "fp32") != (mode + "+filter" if use_filter else mode):
                    traffic_src = "profiles/pmc_density_grid.json is for mode %s: not used" % j.get("mode")
                elif j.get("source_digest") != have:
                    # the counters were collected on other kernels than the ones timed here: a stale constant is no measurement
                    traffic_src = ("profiles/pmc_density_grid.json was measured on library %s, this run loaded %s: not used "
                                   "(tools/profile_bench.sh collects it again)" % (j.get("source_digest"), have))
                else:
                    traffic, traffic_src = j.get("hbm_bytes_per_launch"), "profiles/pmc_density_grid.json (%s)" % j.get("source", "rocprofv3 --pmc passes")
            except Exception as e:
                traffic, traffic_src = None, "profiles/pmc_density_grid.json unreadable: %r" % (e,)
        out = {
            "metric": "meshes/sec + ms/image, TripoSR 256^3 grid, at 1/2/4/8 MI355X",
            "value": args.gpus * args.steps / elapsed,
            "unit": "meshes/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": DTYPE[mode] + (" [filtered grid: an fp16 one-product pass decides the SIGN away from the level, every value marching "
                                    "cubes reads carries the three-limb value: mesh bit-identical to the full evaluation, checked "
                                    "in-bench under density.filtered]" if filtered_active(model) else ""),
            "data": "synthetic 512x512 RGBA composited on grey; random-init weights (seeded), calibrated density bias",
            "config": {"workload": "TripoSR single image -> mesh, mc_resolution=256, 1 image per GPU per step",
                       "entry_points_timed": "TSR.forward([image resident in HBM]) + TSR.extract_meshes(codes, resolution=256): mesh "
                                             "left in HBM; the host-boundary rate is the extra key 'boundary'",
                       "mc_resolution": MC_RES, "threshold": THRESHOLD, "images_per_gpu_per_step": 1,
                       "decoder_precision": mode, "decoder_filter": use_filter,
                       # test plumbing made visible (tests/test_gpu_bench_ranks.py): ranks sharing one device are NOT a multi-GPU figure
                       "shared_gpu": share_gpu, "collective_backend": backend if dist is not None else None,
                       # what the process group itself reports (RCCL's communicator when the backend is nccl): a SCALE line
                       # whose collective_ranks != n_gpus did not run the job it claims
                       "collective_ranks": int(dist.get_world_size()) if dist is not None else None,
                       "mesh": {"vertices": nv, "faces": nf}, "parallelism": "dp%d (replicas, no collectives)" % args.gpus},
            # achieved / frac: the FLOPs the kernel's MFMA instructions execute (cross-check: SQ_INSTS_VALU_MFMA_MOPS_* x 512 in
            # profiles/round3/pmc_summary.txt) over the live HIP-event launch time, against the dense peak of that pipe
            "roofline": {"kernel": FILTER_KERNEL_NAME if use_filter else KERNEL_NAME[mode],
                         "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
                         # the same two numbers under names that say what they count: FLOPs the MFMA instructions
                         # EXECUTE (six limb products per useful product in bf16l3 mode), i.e. matrix-pipe utilisation; the
                         # useful work is `algorithmic_*` below
                         "meaning": "achieved / frac = executed MFMA FLOPs over the launch time (matrix-pipe utilisation; since "
                                    "round 3); useful fp32-equivalent work = algorithmic_tflops / algorithmic_frac",
                         "mfma_executed_tflops": achieved, "mfma_util": achieved / peak,
                         "launch_ms": kern_ms, "launch_ms_covers": ("all launches of sculpt_density_grid_filtered (pass A + bit passes + "
                                                                    "pass C), HIP events on their stream" if use_filter else "the one launch"),
                         "executed_flop_per_launch": executed,
                         "algorithmic_flop_per_launch": FLOP_PER_POINT * MC_RES ** 3,
                         "algorithmic_tflops": FLOP_PER_POINT * MC_RES ** 3 / (kern_ms * 1e-3) / 1e12,
                         # SURVEY 8(d)'s 82.4 kFLOP/point (includes the layer-0 work the separable tables remove) over the SAME
                         # time, against the fp32 matrix peak the reference's arithmetic would be priced at: can pass 1
                         "algorithmic_frac": FLOP_PER_POINT * MC_RES ** 3 / (kern_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS},
            "latency_ms_per_image": elapsed / args.steps * 1e3,  # steps run back to back, one image in flight: step time = latency
        }
        with torch.no_grad():
            st = stage_split(model, imgs[0])  # outside the timed region
            out["stages_ms"] = st
            out["transformer_ms"] = round(st["image_tokenizer"] + st["backbone"], 3)
            # bf16 MFMA fraction of the transformer stack: 2.96 TFLOP / image over the two stage times
            out["transformer_roofline"] = {"bound": "mfma", "achieved": 2.96 / (out["transformer_ms"] * 1e-3), "peak": PEAK_BF16_MFMA_TFLOPS,
                                           "unit": "TFLOP/s", "frac": 2.96 / (out["transformer_ms"] * 1e-3) / PEAK_BF16_MFMA_TFLOPS}
            single = args.gpus == 1 and DECODER_PRECISION == "bf16l3"
            if filt is not None:
                out["density"] = {"filtered": filt,
                                  "guard": {k: model.filter_info[k] for k in ("filtered", "fallbacks", "calibrations")}}
                if use_filter and single and not args.no_siblings:
                    out["density"]["full"] = full_density_sibling(model, imgs, args.steps)
            if single and not args.no_siblings:
                out["fp32_exact"] = fp32_exact_sibling(model, imgs, args.steps)
                out["parity"] = {"kernel_vs_fp32_kernel": kernel_parity(model, imgs[0])}
            if single and not args.no_extras:
                out["boundary"] = boundary_rate(model, imgs_np, args.steps)
                try:
                    out["batched"] = batched_rates(model, imgs, args.steps)
                except Exception as e:  # an extra must never take the headline line down
                    out["batched"] = {"error": "%s: %s" % (type(e).__name__, e)}
            if single and not args.no_extras:
                for key, fn in (("slab512", lambda: slab512_extra(model, imgs[0])), ("sf3d", lambda: sf3d_extra(device))):
                    t0 = time.perf_counter()
                    try:
                        out[key] = fn()
                        out[key]["bench_wall_s"] = round(time.perf_counter() - t0, 1)
                    except Exception as e:  # an extra must never take the headline line down
                        out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
                    torch.cuda.empty_cache()
        if args.gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], cpu_verts = cpu_baseline(sd, imgs_np[0])
            # the headline mode's mesh (bf16 transformer, default decoder mode) against the oracle's fp32 CPU mesh of the same
            # image at BASELINE config 1's resolution: two-sided nearest-vertex distances over the 1.74 extent
            with torch.no_grad():
                gm = model.run_async(imgs_np[0], 128, THRESHOLD).result()
            d = mesh_distance(gm.vertices, cpu_verts, 1.74)
            d["what"] = ("128^3, bench image 0, threshold 25: default mode (bf16 transformer + %s decoder) vs the oracle's fp32 CPU path; "
                         "two-sided nearest-vertex distance / 1.74; bounds asserted in tests/test_gpu_transformer.py: mean < 4e-4, "
                         "p99 < 2e-3, p99.9 < 5e-2, max < 1e-1" % DECODER_PRECISION)
            out.setdefault("parity", {})["bf16_mesh_vs_fp32_cpu"] = d
            if DECODER_PRECISION == "bf16l3" and not args.no_extras:
                try:
                    with torch.no_grad():
                        out["parity_mode"] = parity_mode_extra(sd, model, imgs, imgs_np, cpu_verts, args.steps, "fp16l2")
                        # the all-three-limb mode of round 4 beside it (same measurement)
                        out["parity_mode"]["bf16l3"] = parity_mode_extra(sd, model, imgs, imgs_np, cpu_verts, args.steps, "bf16l3")
                except Exception as e:  # an extra must never take the headline line down
                    out["parity_mode"] = {"error": "%s: %s" % (type(e).__name__, e)}
        else:
            out["cpu_baseline"] = None
        # on its own line whatever was written before it (RCCL warnings go to stdout without a trailing newline)
        sys.stdout.write("\n" + json.dumps(out) + "\n")
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
