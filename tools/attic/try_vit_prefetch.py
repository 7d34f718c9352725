"""Experiment: the image tokenizer (ViT, ~1.0 ms of launches that leave most CUs idle) of image i + 1 on a second stream beside the
backbone + density grid + marching cubes of image i.  Serial reference = TSR.forward + extract_meshes per image (the bench's step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sculptmate_amd import synth

dev = torch.device("cuda:0")
model, sd = bench.build_model(dev, 0)
imgs = [torch.from_numpy(synth.composite_rgb(synth.image_rgba(seed=100 + i))).to(dev) for i in range(4)]
N = 40


def tail(ctx):
    st = model._run_blocks(model._backbone_head(), ctx)
    _, outb = model._backbone_tail(st)
    codes = model.scene_code(outb)[None]
    return model.extract_meshes(codes, False, 256, 25.0)


with torch.no_grad():
    bench.calibrate(model, sd, imgs[0])
    for _ in range(3):
        ref = model.extract_meshes(model.forward(imgs[0]), False, 256, 25.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        model.extract_meshes(model.forward(imgs[i % 4]), False, 256, 25.0)
    torch.cuda.synchronize()
    serial = (time.perf_counter() - t0) / N * 1e3
    print("serial (forward + extract_meshes): %.3f ms / image" % serial)

    main = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(dev)
    ctxs = [torch.empty(1025, 768, dtype=torch.bfloat16, device=dev) for _ in range(2)]

    def vit(i):   # on `side`: tokens of image i into ctxs[i & 1]
        with torch.cuda.stream(side):
            c, _ = model.image_tokens(imgs[i % 4])
            ctxs[i & 1].copy_(c)
            ev = torch.cuda.Event(); ev.record(side)
        return ev

    def run(n, check=False, late=False):
        """late: the tokens of image i + 1 start when the density grid of image i starts (beside it, on the CUs a smaller density
        launch leaves free) instead of when its backbone starts"""
        ev = vit(0)
        for i in range(n):
            main.wait_event(ev)
            if not late:
                fork = torch.cuda.Event(); fork.record(main)
                side.wait_event(fork)
                ev_next = vit(i + 1)
            st = model._run_blocks(model._backbone_head(), ctxs[i & 1])
            _, outb = model._backbone_tail(st)
            codes = model.scene_code(outb)[None]
            if late:
                fork = torch.cuda.Event(); fork.record(main)
                side.wait_event(fork)
                ev_next = vit(i + 1)
            m = model.extract_meshes(codes, False, 256, 25.0)
            if check and i == 0:
                assert torch.equal(m[0].vertices, ref[0].vertices) and torch.equal(m[0].faces, ref[0].faces), "mesh differs"
            ev = ev_next
        torch.cuda.synchronize()

    for late in (False, True):
        for grid in (256, 248, 240, 232, 224, 208, 192):
            os.environ["SCULPT_DENSITY_L3_GRID"] = str(grid)
            run(4, check=True, late=late)
            t0 = time.perf_counter()
            run(N, late=late)
            dt = (time.perf_counter() - t0) / N * 1e3
            print("tokens of image i+1 from the start of image i's %s, density grid on %3d workgroups: %.3f ms / image (x%.3f)"
                  % ("density grid" if late else "backbone    ", grid, dt, serial / dt), flush=True)
