"""The bench's step alone (TSR.forward + extract_meshes on a resident image), 2 x N times, for rocprofv3 (tools/step_gaps.sh):
the first N images are warm-up; the census is taken over the last N."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sculptmate_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
model, sd = bench.build_model(dev, 0)
img = torch.from_numpy(synth.composite_rgb(synth.image_rgba(seed=100))).to(dev)
with torch.no_grad():
    bench.calibrate(model, sd, img)
    for phase in range(2):
        for _ in range(N):
            codes = model.forward(img)
            model.extract_meshes(codes, False, 256, 25.0)
        torch.cuda.synchronize()
        if phase == 0:
            torch.zeros(12345, device=dev).sum().item()   # marker
print("done", N)
