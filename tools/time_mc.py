"""Time sculpt_mc_count's kernels on the bench's own density volume (HIP events around the classify launch are not
available from Python: run under rocprofv3 --kernel-trace --stats, tools/time_mc.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sculptmate_amd import ops, synth
dev = torch.device("cuda:0")
model, sd = bench.build_model(dev, 0)
img = torch.from_numpy(synth.composite_rgb(synth.image_rgba(seed=100))).to(dev)
with torch.no_grad():
    bench.calibrate(model, sd, img)
    codes = model([img], device=dev)
    r = model.renderer.cfg.radius
    vol = ops.density_grid(codes[0].contiguous(), model.decoder, 256, radius=r, density_bias=model.renderer.cfg.density_bias, out_add=-25.0)
    for i in range(6):
        try:
            v, f = ops.marching_cubes(vol.view(256, 256, 256), 0.0, reference_order=True, vert_div=255.0, vert_mul=1.74, vert_add=-0.87)
            n = (v.shape[0], f.shape[0])
        except Exception as e:
            n = str(e)[:60]
    torch.cuda.synchronize()
    print("mesh", n)
