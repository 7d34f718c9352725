"""BASELINE config 5 timing: one scene code at 512^3, slabs over WORLD_SIZE ranks (torchrun) or emulated
locally (--emulate N: N slabs one after the other on one GPU, checks equality with the single pass).
    python tools/bench_slab512.py --res 512 --emulate 8
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_slab512.py --res 512
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sculptmate_amd import ops, parallel, slab, synth

ap = argparse.ArgumentParser(); ap.add_argument("--res", type=int, default=512); ap.add_argument("--emulate", type=int, default=0)
ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
rank, local, world = parallel.env_rank_world()
torch.cuda.set_device(local); dev = torch.device("cuda", local)
if world > 1: parallel.init("nccl", dev)
Ws, bs = synth.decoder_lists(synth.decoder_state(seed=1))
planes = torch.from_numpy(synth.smooth_triplane(seed=2, scale=3.0)).to(dev)
mlp = ops.PackedMLP(Ws, bs, dev)
thr = float(ops.density_grid(planes, mlp, 64).quantile(0.97))
R = a.res
def run():
    if world > 1:
        return slab.gather_and_assemble(slab.extract_slab(planes, mlp, R, rank, world, threshold=thr), dev)
    return slab.extract_mesh_slabs_local(planes, mlp, R, max(a.emulate, 1), threshold=thr)
v, f = run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters): v, f = run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.iters
dt = parallel.max_over_ranks(dt, dev)
if rank == 0:
    print("res %d world %d emulate %d: %.2f ms  (%d verts, %d faces)" % (R, world, a.emulate, dt * 1e3, len(v), len(f)))
    if a.emulate > 1:
        v1, f1 = slab.extract_mesh_slabs_local(planes, mlp, R, 1, threshold=thr)
        print("equal to single pass:", bool(torch.equal(v, v1) and torch.equal(f, f1)))
