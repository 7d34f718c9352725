"""Can pass A of the two-pass density grid (VALU-bound, matrix pipe 78 % idle) run BESIDE the transformer's GEMMs / attention of
the next image on the same CUs?  Two HIP streams: stream D loops pass A at 256^3 (workgroup form from SCULPT_COARSE_WG =
"threads,lds_kb": 1024 threads own a CU's whole register file; 512 threads leave half of it and, with the LDS padded past 80 KB,
only one such workgroup fits a CU), stream F loops the model's forward.  Prints serial and concurrent times.
    SCULPT_COARSE_WG=512,84 python tools/try_coresident.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from sculptmate_amd import ops, synth

dev = torch.device("cuda:0")
model, sd = bench.build_model(dev, 0)
img = torch.from_numpy(synth.composite_rgb(synth.image_rgba(seed=100))).to(dev).contiguous()
R, cfg = 256, model.renderer.cfg
N = int(os.environ.get("N", "10"))
with torch.no_grad():
    bench.calibrate(model, sd, img)
    codes = model([img], device=dev)
    planes = codes[0].contiguous()
    model.calibrate_decoder_filter(planes)
    margin, coarse = model.filter_info["margin"], model.filter_info["coarse"]
    vol = torch.empty(R ** 3, dtype=torch.float32, device=dev)
    kw = dict(radius=cfg.radius, density_bias=cfg.density_bias, out_add=-25.0, coarse=coarse, out=vol)
    ops.density_grid_filtered(planes, model.decoder, R, margin, **kw)     # tables in the workspace

    def pass_a(n):
        for _ in range(n):
            ops.density_grid_filtered(planes, model.decoder, R, margin, passes="A", tables=False, **kw)

    def fwd(n):
        for _ in range(n):
            ctx, _ = model.image_tokens(img)
            _, outb = model.backbone_tokens(ctx)
            model.scene_code(outb)

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    pass_a(2); fwd(2)
    ta = timed(lambda: pass_a(N)) / N
    tf = timed(lambda: fwd(N)) / N
    sD, sF = torch.cuda.Stream(), torch.cuda.Stream()

    def both():
        with torch.cuda.stream(sF):
            fwd(N)
        with torch.cuda.stream(sD):
            pass_a(N)

    def both_interleaved():   # host queues one forward, one pass A, ... so that neither stream runs dry
        for _ in range(N):
            with torch.cuda.stream(sF):
                fwd(1)
            with torch.cuda.stream(sD):
                pass_a(1)

    both()
    tb = timed(both) / N
    ti = timed(both_interleaved) / N
    print("SCULPT_COARSE_WG=%s: pass A alone %.3f ms, forward alone %.3f ms, serial sum %.3f; concurrent %.3f ms (queued in bulk) / %.3f ms "
          "(interleaved) per pair -> hidden %.3f ms of pass A" % (os.environ.get("SCULPT_COARSE_WG", "1024 (default)"), ta, tf, ta + tf, tb, ti,
                                                                 ta + tf - min(tb, ti)))
