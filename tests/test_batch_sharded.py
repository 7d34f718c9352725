"""sculptmate_amd.batch.run_sharded: a batch of images over the ranks (one image per GPU at a time, no data-path collective).
CPU leg: two gloo processes with a stand-in model exercise the sharding, the pipelining order, the count exchange and the
writers; GPU leg: two ranks sharing one MI355X (gloo for the one exchange) with the real small model against a single-process run,
bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

FAKE = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from sculptmate_amd import batch, parallel, meshio
from sculptmate_amd.tsr.system import Mesh

class Pending:
    def __init__(self, m, log): self.m, self.log = m, log
    def result(self):
        self.log.append(("result", self.m.tag)); return self.m

class FakeModel:
    """run_async(image) -> a mesh that is a pure function of the image; records the call order"""
    device = "cpu"
    def __init__(self): self.log = []
    def run_async(self, im, res, thr, tex):
        n = 3 + int(im.sum()) % 5
        rng = np.random.default_rng(int(im.sum() * 1000) % 2**31)
        m = Mesh(rng.random((n, 3), dtype=np.float32), rng.integers(0, n, (2 * n, 3)).astype(np.int64), None)
        m.tag = int(im[0, 0, 0] * 100)
        self.log.append(("run", m.tag))
        return Pending(m, self.log)

images = [np.full((4, 4, 3), i / 100.0, np.float32) for i in range(7)]
out = sys.argv[2]
rank, local, world = parallel.env_rank_world()
if world > 1:
    parallel.init("gloo")
m = FakeModel()
loc, summary = batch.run_sharded(m, images, 32, 1.0, out_dir=out, fmt="ply")
assert sorted(loc) == parallel.shard_indices(7, rank, world)
assert [s[0] for s in summary] == list(range(7)) and all(s[1] == s[0] % world for s in summary)
# pipelining: mesh i is collected after image i + 1 was queued
runs = [t for k, t in m.log if k == "run"]
for a, b in zip(runs, runs[1:]):
    assert m.log.index(("run", b)) < m.log.index(("result", a))
for i, r, nv, nf in summary:
    if r == rank:
        assert loc[i].vertices.shape[0] == nv and loc[i].faces.shape[0] == nf
# callables are only invoked by the owning rank
called = []
lazy = [(lambda i=i: (called.append(i), images[i])[1]) for i in range(7)]
batch.run_sharded(FakeModel(), lazy, 32, 1.0, keep=False)
assert called == parallel.shard_indices(7, rank, world)
print("rank", rank, "summary", summary)
parallel.barrier()
'''


def _launch(script, args, world, extra_env=None, timeout=600):
    port = 29000 + (os.getpid() * 7) % 900
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, str(script)] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=timeout)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return outs


def test_run_sharded_two_ranks_gloo_equals_one_rank(tmp_path):
    from sculptmate_amd import meshio

    script = tmp_path / "w.py"
    script.write_text(FAKE)
    one, two = tmp_path / "one", tmp_path / "two"
    o1 = _launch(script, [ROOT, str(one)], 1)
    o2 = _launch(script, [ROOT, str(two)], 2)
    s1 = o1[0].split("summary ")[1].strip()
    # same (index, vertices, faces) everywhere; only the owning rank differs
    strip = lambda s: [(a, c, d) for a, b, c, d in eval(s)]
    assert strip(o2[0].split("summary ")[1].strip()) == strip(o2[1].split("summary ")[1].strip()) == strip(s1)
    files = sorted(os.listdir(one))
    assert files == sorted(os.listdir(two)) == ["mesh_%05d.ply" % i for i in range(7)]  # every image exactly once
    for f in files:
        a, b = meshio.read_ply(str(one / f)), meshio.read_ply(str(two / f))
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


REAL = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from sculptmate_amd import batch, parallel, synth, ops
from sculptmate_amd.tsr import TSR
from sculptmate_amd.tsr.spec import SMALL_CFG
rank, local, world = parallel.env_rank_world()
batch.cap_host_threads(world)
dev = torch.device("cuda", 0)            # both ranks share the one GPU of the test box; the exchange runs over gloo
torch.cuda.set_device(dev)
if world > 1:
    parallel.init("gloo")
m = TSR(SMALL_CFG); m.load_state_dict(synth.tsr_state(seed=21, cfg=SMALL_CFG)); m.to(dev)
S = SMALL_CFG["cond_image_size"]
imgs = [synth.composite_rgb(synth.image_rgba(seed=300 + i, size=S)) for i in range(5)]
thr = float(ops.density_grid(m([imgs[0]], device=dev)[0].contiguous(), m.decoder, 32).median())
loc, summary = m.run_sharded(imgs, 32, thr, enable_texture=True, out_dir=sys.argv[2], fmt="ply")
assert sorted(loc) == parallel.shard_indices(5, rank, world)
print("rank", rank, "summary", summary)
parallel.barrier()
'''


@pytest.mark.gpu
def test_run_sharded_real_model_two_ranks_on_one_gpu(tmp_path):
    from sculptmate_amd import meshio

    script = tmp_path / "w.py"
    script.write_text(REAL)
    one, two = tmp_path / "one", tmp_path / "two"
    env = {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    _launch(script, [ROOT, str(one)], 1, env)
    o2 = _launch(script, [ROOT, str(two)], 2, env)
    assert "summary" in o2[0] and "summary" in o2[1]
    files = sorted(os.listdir(one))
    assert files == sorted(os.listdir(two)) == ["mesh_%05d.ply" % i for i in range(5)]
    for f in files:
        a, b = meshio.read_ply(str(one / f)), meshio.read_ply(str(two / f))
        assert len(a[0]) > 50
        for x, y in zip(a, b):
            assert (x is None and y is None) or np.array_equal(x, y)
