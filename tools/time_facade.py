"""Stage times of the add-on's TripoSR call at full size with the options the bench does not cover: PIL image in (host
preprocessing), vertex colours, meshes to host numpy (what import_obj_blender consumes).  python tools/time_facade.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
import bench
from sculptmate_amd import synth

dev = torch.device("cuda:0")
model, sd = bench.build_model(dev, 0)
img_np = synth.composite_rgb(synth.image_rgba(seed=100))
img_dev = torch.from_numpy(img_np).to(dev)
bench.calibrate(model, sd, img_dev)
pil = Image.fromarray((img_np * 255).astype(np.uint8))

def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3, r

with torch.no_grad():
    ms, codes = t(lambda: model([pil], device=dev));                    print("forward([PIL 512x512])            %7.2f ms" % ms)
    ms, codes = t(lambda: model([img_dev], device=dev));                print("forward([device tensor])          %7.2f ms" % ms)
    ms, m0 = t(lambda: model.extract_meshes(codes, False, 256, 25.0));  print("extract_meshes(colours off)       %7.2f ms" % ms)
    ms, m1 = t(lambda: model.extract_meshes(codes, True, 256, 25.0));   print("extract_meshes(vertex colours)    %7.2f ms  (%d vertices)" % (ms, m1[0].vertices.shape[0]))
    ms, _ = t(lambda: (m1[0].vertices.cpu().numpy(), m1[0].faces.cpu().numpy(), m1[0].vertex_colors.cpu().numpy()))
    print("mesh + colours -> host numpy      %7.2f ms" % ms)
    model.mesh_sink = lambda v, f, c, name: None
    ms, _ = t(lambda: model.extract_mesh(codes, True, "x", 256));       print("extract_mesh(enable_texture=True) %7.2f ms  (through the sink hand-off)" % ms)
