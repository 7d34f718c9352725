#!/usr/bin/env python
"""Dense density grid at 256^3, every decoder mode side by side in ONE process (interleaved rounds, guide rule 24):
launch time by HIP events and max |log d - log d_oracle| on a sample of lattice points.

    python tools/time_density.py [--rounds 5] [--R 256] [--l3-threads 1024,768,512] [--l3-kstep 0,1] [--l3-grid 256,128,64]
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
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--R", type=int, default=256)
    ap.add_argument("--l3-threads", default="1024")
    ap.add_argument("--l3-kstep", default="1")
    ap.add_argument("--l3-grid", default="256")
    ap.add_argument("--modes", default="fp32,fp16x3,bf16x3,bf16l3")
    args = ap.parse_args()
    from oracle import capi
    from sculptmate_amd import ops, synth

    dev = torch.device("cuda:0")
    R = args.R
    Ws, bs = synth.decoder_lists(synth.decoder_state(seed=13))
    tri_np = synth.smooth_triplane(seed=14, scale=3.0)
    mlp = ops.PackedMLP(Ws, bs, dev)
    tri = torch.from_numpy(tri_np).to(dev)
    rng = np.random.default_rng(0)
    idx = np.unique(rng.integers(0, R ** 3, 200000))
    ref = np.log(capi.query_triplane(tri_np, capi.grid_points(R, 0.87, idx), Ws, bs)["density_act"][:, 0].astype(np.float64))
    variants = []
    for m in args.modes.split(","):
        if m == "bf16l3":
            for nt in args.l3_threads.split(","):
                for ks in args.l3_kstep.split(","):
                    for gr in args.l3_grid.split(","):
                        variants.append((m, {"SCULPT_DENSITY_L3_THREADS": nt, "SCULPT_DENSITY_L3_KSTEP": ks, "SCULPT_DENSITY_L3_GRID": gr}))
        else:
            variants.append((m, {}))
    out = torch.empty(R ** 3, dtype=torch.float32, device=dev)
    times = {i: [] for i in range(len(variants))}
    err = {}
    for rnd in range(args.rounds + 1):
        for i, (m, env) in enumerate(variants):
            os.environ.update(env)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ops.density_grid(tri, mlp, R, out=out, precision=m, events=ev)
            torch.cuda.synchronize()
            if rnd:
                times[i].append(ev[0].elapsed_time(ev[1]))
            else:
                got = np.log(out.cpu().numpy()[idx].astype(np.float64))
                d = np.abs(got - ref)
                err[i] = (float(d.max()), float(d.mean()), bool(np.isfinite(got).all()))
    for i, (m, env) in enumerate(variants):
        t = np.array(times[i])
        print("%-8s %-40s median %.3f ms  min %.3f ms   |dlog d| max %.2e mean %.2e finite=%s"
              % (m, " ".join("%s=%s" % (k[15:], v) for k, v in env.items()), np.median(t), t.min(), *err[i]))


if __name__ == "__main__":
    main()
