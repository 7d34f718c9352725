"""Timing experiment: ViT of image i+1 on a second stream under the BACKBONE of image i (density + MC stay alone on the GPU)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from sculptmate_amd import ops, synth

dev = torch.device("cuda:0")
model, sd = bench.build_model(dev, 0)
imgs = [torch.from_numpy(synth.composite_rgb(synth.image_rgba(seed=100 + i))).to(dev).contiguous() for i in range(4)]
R, r = 256, model.renderer.cfg.radius
N = 24
with torch.no_grad():
    bench.calibrate(model, sd, imgs[0])
    for _ in range(3):
        bench.one_step(model, imgs[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        bench.one_step(model, imgs[i % 4])
    torch.cuda.synchronize()
    print("sequential (head of the backbone under the ViT): %.2f ms/image" % ((time.perf_counter() - t0) / N * 1e3))

    side = torch.cuda.Stream()
    ctx0, _ = model.image_tokens(imgs[0])
    ctx_saved = ctx0.clone()
    vol = torch.empty(R ** 3, dtype=torch.float32, device=dev)

    def step(i):
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            model.image_tokens(imgs[(i + 1) % 4])          # ViT(i+1), result unused here (timing only)
        _, outb = model.backbone_tokens(ctx_saved)          # backbone(i) on the main stream
        planes = model.scene_code(outb)
        ops.density_grid(planes, model.decoder, R, radius=r, density_bias=model.renderer.cfg.density_bias, out_add=-25.0, out=vol)
        torch.cuda.current_stream().wait_stream(side)
        return ops.marching_cubes(vol.view(R, R, R), 0.0, reference_order=True, vert_div=R - 1.0, vert_mul=2 * r, vert_add=-r)

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        step(i)
    torch.cuda.synchronize()
    print("ViT(i+1) under backbone(i): %.2f ms/image" % ((time.perf_counter() - t0) / N * 1e3))
