"""Image i+1's transformer beside image i's density grid + marching cubes on DISJOINT CUs.

Two HIP streams with CU masks (sculpt_stream_create_cu_mask): A = the first `d` CUs (density grid, marching cubes of image i),
B = the remaining 256 - d CUs (ViT + backbone of image i+1).  Reports meshes/s per split against the serial pipeline, and
checks that the meshes are bit-identical to the serial ones.
"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from sculptmate_amd import _lib, ops, synth

dev = torch.device("cuda:0")
model, sd = bench.build_model(dev, 0)
imgs = [torch.from_numpy(synth.composite_rgb(synth.image_rgba(seed=100 + i))).to(dev) for i in range(4)]
N = 24


def masked_stream(first, n):
    h = ctypes.c_void_p()
    _lib.check(_lib.lib.sculpt_stream_create_cu_mask(first, n, ctypes.byref(h)))
    return torch.cuda.ExternalStream(h.value, device=dev), h


def forward_plain(img):
    """ViT + backbone + upsample on the CURRENT stream only (no second stream inside)."""
    ctx, _ = model.image_tokens(img)
    _, outb = model.backbone_tokens(ctx)
    return model.scene_code(outb)


def mesh_of(planes):
    r = model.renderer.cfg.radius
    vol = ops.density_grid(planes, model.decoder, 256, radius=r, density_bias=model.renderer.cfg.density_bias, out_add=-25.0,
                           precision=model.decoder_precision)
    return ops.marching_cubes(vol.view(256, 256, 256), 0.0, reference_order=True, vert_div=255.0, vert_mul=2 * r, vert_add=-r)


with torch.no_grad():
    bench.calibrate(model, sd, imgs[0])
    # serial reference (one stream, all CUs)
    ref = []
    for i in range(4):
        v, f = mesh_of(forward_plain(imgs[i]))
        ref.append((v.clone(), f.clone()))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        mesh_of(forward_plain(imgs[i % 4]))
    torch.cuda.synchronize()
    serial = (time.perf_counter() - t0) / N * 1e3
    print("serial, one stream, 256 CUs: %.3f ms/image  %.1f meshes/s" % (serial, 1e3 / serial))
    for d in (256, 224, 192, 176, 160, 144, 128, 112, 96):
        os.environ["SCULPT_DENSITY_L3_GRID"] = str(d)   # one density workgroup per CU of the mask (the kernel splits its tiles over any grid)
        if d == 256:   # two unmasked streams: the overlap r1 measured (tools/try_pipeline.py)
            sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
            ha = hb = None
        else:
            (sa, ha), (sb, hb) = masked_stream(0, d), masked_stream(d, 256 - d)
        ok = True

        def run(n, check):
            global ok
            with torch.cuda.stream(sb):
                planes = forward_plain(imgs[0])
                ev = torch.cuda.Event(); ev.record(sb)
            for i in range(n):
                with torch.cuda.stream(sb):           # image i+1's transformer: queued first, runs on B's CUs
                    nxt = forward_plain(imgs[(i + 1) % 4])
                    ev_n = torch.cuda.Event(); ev_n.record(sb)
                with torch.cuda.stream(sa):           # image i's density grid + marching cubes on A's CUs
                    sa.wait_event(ev)
                    v, f = mesh_of(planes)
                    planes.record_stream(sa)
                if check:
                    sa.synchronize()                  # the emit kernels of image i (stream A) before reading v, f here
                    rv, rf = ref[i % 4]
                    ok = ok and torch.equal(v, rv) and torch.equal(f, rf)
                planes, ev = nxt, ev_n
            torch.cuda.synchronize()

        run(4, True)
        t0 = time.perf_counter()
        run(N, False)
        dt = (time.perf_counter() - t0) / N * 1e3
        print("density+MC on %3d CUs | transformer on %3d CUs: %.3f ms/image  %.1f meshes/s  (x%.3f vs serial)  meshes identical: %s"
              % (d, 256 - d if d < 256 else 256, dt, 1e3 / dt, serial / dt, ok))
        if ha is not None:
            torch.cuda.synchronize()
            _lib.lib.sculpt_stream_destroy(ha); _lib.lib.sculpt_stream_destroy(hb)
