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


FAIL = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from sculptmate_amd import batch, parallel
from sculptmate_amd.tsr.system import Mesh

class Pending:
    def __init__(self, m): self.m = m
    def result(self): return self.m

class FailingModel:
    device = "cpu"
    def run_async(self, im, res, thr, tex):
        if int(round(float(im[0, 0, 0]) * 100)) == 3:
            raise RuntimeError("boom on image 3")
        return Pending(Mesh(np.zeros((3, 3), np.float32), np.zeros((1, 3), np.int64), None))

images = [np.full((4, 4, 3), i / 100.0, np.float32) for i in range(6)]
rank, local, world = parallel.env_rank_world()
parallel.init("gloo")
try:
    batch.run_sharded(FailingModel(), images, 32, 1.0, out_dir=sys.argv[2], fmt="npz")
except RuntimeError as e:
    print("rank", rank, "raised:", e)
    sys.exit(3)
sys.exit(0)
'''


def test_run_sharded_failure_on_one_rank_fails_every_rank_without_hanging(tmp_path):
    """a rank that raises inside run_sharded must not leave the others waiting in a collective.  The failing rank
    puts an error marker into the ONE exchange; every rank raises (the owner with its own exception), nobody times out."""
    script = tmp_path / "f.py"
    script.write_text(FAIL)
    port = 29000 + (os.getpid() * 11) % 900
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path / "out")], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=120)[0] for p in procs]       # a hang would be a TimeoutExpired here
    assert [p.returncode for p in procs] == [3, 3], outs
    assert "boom on image 3" in outs[1] and "rank 1 failed" in outs[0], outs   # image 3 belongs to rank 1


def test_mesh_writer_pool_writes_the_synchronous_bytes(tmp_path):
    """MeshWriter (files written by threads off the submit loop) == the synchronous writer, byte for byte, for every format;
    an error in a worker surfaces from close()."""
    from sculptmate_amd import batch, meshio
    from sculptmate_amd.tsr.system import Mesh

    rng = np.random.default_rng(0)
    meshes = []
    for i in range(6):
        nv = 2000 + 1000 * i
        meshes.append(Mesh(rng.random((nv, 3), dtype=np.float32), rng.integers(0, nv, (2 * nv + 7, 3)).astype(np.int64),
                           rng.random((nv, 3), dtype=np.float32) if i % 2 else None))
    big = Mesh(rng.random((300000, 3), dtype=np.float32), rng.integers(0, 300000, (600001, 3)).astype(np.int64), None)  # several face chunks
    meshes.append(big)
    for fmt in ("ply", "npz", "obj"):
        w = batch.MeshWriter(workers=3, max_pending=2)
        for i, m in enumerate(meshes if fmt != "obj" else meshes[:3]):
            w.submit(str(tmp_path / ("a%d.%s" % (i, fmt))), m, fmt)
            batch._write(str(tmp_path / ("s%d.%s" % (i, fmt))), m, fmt)
        w.close()
        for i in range(len(meshes) if fmt != "obj" else 3):
            assert (tmp_path / ("a%d.%s" % (i, fmt))).read_bytes() == (tmp_path / ("s%d.%s" % (i, fmt))).read_bytes(), (fmt, i)
    v, f, c = meshio.read_ply(str(tmp_path / "a6.ply"))
    assert np.array_equal(v, big.vertices) and np.array_equal(f, big.faces) and c is None
    w = batch.MeshWriter(workers=2)
    w.submit(str(tmp_path / "no_such_dir" / "x.ply"), meshes[0], "ply")
    with pytest.raises(OSError):
        w.close()


BATCH_CLI = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from sculptmate_amd import batch, synth
from sculptmate_amd.tsr import TSR
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
sd = synth.tsr_state(seed=0)
m = TSR(pos_embed_mode="scale_factor"); m.load_state_dict(sd); m.to(dev)
imgs = [synth.composite_rgb(synth.image_rgba(seed=100 + i)) for i in range(8)]
with torch.no_grad():
    synth.calibrate_tsr_density_bias(m, sd, torch.from_numpy(imgs[0]).to(dev), threshold=25.0)
    def plain():       # the same 8 images through run_async with the tokenizer look-ahead, nothing written
        t0 = time.perf_counter(); ms = m.run(imgs, 256, 25.0, batch=1); return time.perf_counter() - t0, ms
    def sharded(out, writers):
        t0 = time.perf_counter(); batch.run_sharded(m, imgs, 256, 25.0, out_dir=out, fmt="ply", keep=False, writers=writers)
        return time.perf_counter() - t0
    plain(); sharded(sys.argv[2] + "/warm", 4)          # pinned pool, streams, page cache of the output directory
    t_plain = min(plain()[0] for _ in range(3))
    t_pool = min(sharded(sys.argv[2] + "/pool", 4) for _ in range(3))
    t_sync = sharded(sys.argv[2] + "/sync", 0)
print("RESULT", t_plain, t_pool, t_sync)
'''


@pytest.mark.gpu
def test_batch_cli_writer_keeps_up_with_the_gpu(tmp_path):
    """8 synthetic images at 256^3 (0.95 M vertices / 1.76 M faces each, 34 MB of PLY) through run_sharded with
    the files written off the submit loop take at most 1.3 x the time of the same images through TSR.run without writing
    anything; the files are byte-identical to the synchronous writer's.  The output directory is RAM-backed where there is one
    (/dev/shm): the test is about the writer, not the box's disk."""
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else str(tmp_path)
    out = os.path.join(base, "sculpt_batch_test_%d" % os.getpid())
    script = tmp_path / "b.py"
    script.write_text(BATCH_CLI)
    try:
        o = _launch(script, [ROOT, out], 1, {"HSA_ENABLE_IPC_MODE_LEGACY": "0"})[0]
        t_plain, t_pool, t_sync = [float(x) for x in o.split("RESULT")[1].split()]
        print("8 images at 256^3: TSR.run %.1f ms, run_sharded + writer pool %.1f ms (%.2fx), synchronous writer %.1f ms"
              % (t_plain * 1e3, t_pool * 1e3, t_pool / t_plain, t_sync * 1e3))
        files = sorted(os.listdir(os.path.join(out, "pool")))
        assert files == ["mesh_%05d.ply" % i for i in range(8)] == sorted(os.listdir(os.path.join(out, "sync")))
        for f in files:
            with open(os.path.join(out, "pool", f), "rb") as a, open(os.path.join(out, "sync", f), "rb") as b:
                assert a.read() == b.read(), f
        assert t_pool <= 1.3 * t_plain, (t_plain, t_pool, t_sync)
    finally:
        import shutil

        shutil.rmtree(out, ignore_errors=True)


@pytest.mark.gpu
def test_batch_cli_two_ranks_batched_passes_write_the_one_rank_files(tmp_path):
    """`python -m sculptmate_amd.batch --synthetic 16 --batch 4` (the small test configuration of the model: the
    full-size CLI run is test_batch_cli_writer_keeps_up_with_the_gpu) launched as two ranks (both on the one GPU of
    the box, the count exchange over gloo) writes, file for file and byte for byte, what the one-rank run writes: every image
    exactly once, each in a 4-image transformer pass whose other members differ between the two runs (an image's rows do not see
    its neighbours'), the two-pass density grid calibrated by each rank on its own first image."""
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else str(tmp_path)
    out = os.path.join(base, "sculpt_batch_cli_%d" % os.getpid())
    common = [sys.executable, "-m", "sculptmate_amd.batch", "--synthetic", "16", "--synthetic-model", "small", "--batch", "4",
              "--resolution", "96", "--backend", "gloo"]
    try:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SCULPT_BATCH_SHARE_GPU="1", PYTHONPATH=ROOT)
        p = subprocess.run(common + ["--out", os.path.join(out, "one")], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
        port = 29100 + (os.getpid() * 11) % 800
        procs = []
        for r in range(2):
            e = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen(common + ["--out", os.path.join(out, "two")], env=e, stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT, text=True, cwd=ROOT))
        outs = [q.communicate(timeout=900)[0] for q in procs]
        for q, o in zip(procs, outs):
            assert q.returncode == 0, o[-3000:]
        lines = [ln for ln in outs[0].splitlines() if ln.startswith("synthetic_")]
        assert len(lines) == 16 and sum("rank 1" in ln for ln in lines) == 8     # rank 0 prints the summary of all images
        files = sorted(os.listdir(os.path.join(out, "one")))
        assert files == ["synthetic_%05d.ply" % i for i in range(16)] == sorted(os.listdir(os.path.join(out, "two")))
        for f in files:
            with open(os.path.join(out, "one", f), "rb") as a, open(os.path.join(out, "two", f), "rb") as b:
                x, y = a.read(), b.read()
            assert len(x) > 20000 and x == y, f
    finally:
        import shutil

        shutil.rmtree(out, ignore_errors=True)


def test_run_sharded_batched_passes_with_a_stand_in_model(tmp_path):
    """run_sharded(batch=3): this rank's images go through model.run_batched in groups, the previous group is collected while the
    next is queued, every image is written exactly once and the summary is complete (single process, stand-in model)."""
    from sculptmate_amd import batch, meshio
    from sculptmate_amd.tsr.system import Mesh

    calls = []

    class Pending:
        def __init__(self, m):
            self.m = m

        def result(self):
            calls.append(("result", int(self.m.vertices[0, 0])))
            return self.m

    class Model:
        device = "cpu"

        def run_batched(self, images, batch_, res, thr, tex):
            calls.append(("pass", [int(im[0, 0, 0]) for im in images]))
            out = []
            for im in images:
                k = int(im[0, 0, 0])
                out.append(Pending(Mesh(np.full((3 + k, 3), k, np.float32), np.zeros((k + 1, 3), np.int64), None)))
            return out

        def run_async(self, *a, **k):
            raise AssertionError("batched mode must not fall back to run_async")

    images = [np.full((2, 2, 3), i, np.float32) for i in range(8)]
    local, summary = batch.run_sharded(Model(), images, 32, 1.0, out_dir=str(tmp_path), fmt="npz", batch=3)
    assert sorted(local) == list(range(8))
    assert summary == [(i, 0, 3 + i, i + 1) for i in range(8)]
    passes = [c[1] for c in calls if c[0] == "pass"]
    assert passes == [[0, 1, 2], [3, 4, 5], [6, 7]]
    # the first group's meshes are collected only after the second pass was queued
    assert calls.index(("pass", [3, 4, 5])) < calls.index(("result", 0))
    assert sorted(os.listdir(tmp_path)) == ["mesh_%05d.npz" % i for i in range(8)]
    z = np.load(tmp_path / "mesh_00005.npz")
    assert z["vertices"].shape == (8, 3) and z["faces"].shape == (6, 3)
