#!/usr/bin/env python
"""The two-pass ("filtered") density grid against the full three-limb evaluation on one field (tools/time_density.py's, with the
last bias calibrated so that ~1.5 % of the lattice is inside, like bench.py's): launch times by HIP events (interleaved rounds),
coarse error (mark-all calibration at 64^3), refined fraction, and the identity checks -- every value marching cubes reads bit-equal
to the full volume, no sign mismatch anywhere, marching-cubes output bit-equal.

    python tools/time_density_filter.py [--R 256] [--rounds 5] [--inside 0.015] [--safety 8]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--R", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--inside", type=float, default=0.015)
    ap.add_argument("--safety", type=float, default=8.0)
    ap.add_argument("--threshold", type=float, default=25.0)
    ap.add_argument("--scale", type=float, default=3.0)
    args = ap.parse_args()
    from sculptmate_amd import ops, synth

    dev = torch.device("cuda:0")
    R, thr = args.R, args.threshold
    Ws, bs = synth.decoder_lists(synth.decoder_state(seed=13))
    tri = torch.from_numpy(synth.smooth_triplane(seed=14, scale=args.scale)).to(dev)
    # calibrate the last bias on a 64^3 probe of the full kernel
    mlp = ops.PackedMLP(Ws, bs, dev)
    probe = ops.density_grid(tri, mlp, 64, precision="bf16l3").cpu().numpy().astype(np.float64)
    pre = np.log(probe) + 1.0
    bs[-1] = bs[-1].copy()
    bs[-1][0] += float(np.log(thr) + 1.0 - np.quantile(pre, 1.0 - args.inside))
    mlp = ops.PackedMLP(Ws, bs, dev)
    full = ops.density_grid(tri, mlp, R, out_add=-thr, precision="bf16l3")
    torch.cuda.synchronize()
    print("field: inside fraction %.4f at %d^3" % (float((full > 0).float().mean()), R))
    ev = lambda: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))  # noqa: E731
    for coarse in ("fp16", "bf16"):
        _, st = ops.density_grid_filtered(tri, mlp, 64, 0.0, out_add=0.0, coarse=coarse, mark_all=True)
        cal = ops.filter_stats(st)
        margin = max(args.safety * cal["max_err"], 1e-3)
        print("%s: calibration at 64^3: max |dlog d| %.3e (nonfinite %d) -> margin %.3e" % (coarse, cal["max_err"], cal["n_nonfinite"], margin))
        out = torch.empty(R ** 3, dtype=torch.float32, device=dev)
        vol, st = ops.density_grid_filtered(tri, mlp, R, margin, out_add=-thr, coarse=coarse, out=out)
        s = ops.filter_stats(st)
        print("   refined %.2f %% of %d points (marked %.2f %%, possibly active cells %.2f %%), guard max err %.3e (%.2f of margin)"
              % (100.0 * s["n_refined"] / s["n_points"], s["n_points"], 100.0 * s["n_marked"] / s["n_points"],
                 100.0 * s["n_cells"] / s["n_points"], s["max_err"], s["max_err"] / margin))
        sign_mismatch = int(((vol > 0) != (full > 0)).sum())
        # every value marching cubes READS in the full volume (end points of sign-changing edges, all corners of cells with an
        # ambiguous sign pattern: tests/_mcneeds.py) must carry the full evaluation's bits
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from _mcneeds import needed_points

        need3, n_active = needed_points(full.view(R, R, R) > 0, luts=os.path.join(ROOT, "sculptmate_amd", "csrc", "mc_luts.h"))
        need = need3.view(-1)
        diff_at_need = int((vol.view(torch.int32)[need] != full.view(torch.int32)[need]).sum())
        print("   sign mismatches vs full %d; active cells %d (%.2f %%), values marching cubes reads %d (%.2f %% of the lattice), of them with other bits %d"
              % (sign_mismatch, n_active, 100.0 * n_active / (R - 1) ** 3, int(need.sum()), 100.0 * float(need.float().mean()), diff_at_need))
        v1, f1 = ops.marching_cubes(full.view(R, R, R), 0.0)
        v2, f2 = ops.marching_cubes(vol.view(R, R, R), 0.0)
        same = v1.shape == v2.shape and f1.shape == f2.shape and bool((v1.view(torch.int32) == v2.view(torch.int32)).all()) and bool((f1 == f2).all())
        print("   marching cubes: %d verts / %d faces, mesh identical to the full evaluation's: %s" % (v2.shape[0], f2.shape[0], same))
        tf, tg = [], []
        for rnd in range(args.rounds + 1):
            e1 = ev()
            ops.density_grid(tri, mlp, R, out_add=-thr, precision="bf16l3", out=full, events=e1)
            e2 = ev()
            ops.density_grid_filtered(tri, mlp, R, margin, out_add=-thr, coarse=coarse, out=out, events=e2)
            torch.cuda.synchronize()
            if rnd:
                tf.append(e1[0].elapsed_time(e1[1]))
                tg.append(e2[0].elapsed_time(e2[1]))
        print("   full %.3f ms (min %.3f)   filtered %.3f ms (min %.3f)   ratio %.3f"
              % (np.median(tf), min(tf), np.median(tg), min(tg), np.median(tg) / np.median(tf)))
        tp = {"A": [], "B": [], "C": []}
        for rnd in range(3 * args.rounds + 1):
            es = {}
            for k, ps in enumerate("ABC"):
                es[ps] = ev()
                ops.density_grid_filtered(tri, mlp, R, margin, out_add=-thr, coarse=coarse, out=out, events=es[ps], passes=ps, tables=(k == 0))
            torch.cuda.synchronize()
            if rnd:
                for ps in "ABC":
                    tp[ps].append(es[ps][0].elapsed_time(es[ps][1]))
        print("   pass A %.3f ms (min %.3f)   pass B %.3f ms   pass C %.3f ms (min %.3f)"
              % (np.median(tp["A"]), min(tp["A"]), np.median(tp["B"]), np.median(tp["C"]), min(tp["C"])))


if __name__ == "__main__":
    main()
