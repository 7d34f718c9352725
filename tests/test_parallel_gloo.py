"""N>1 path on CPU: world_size-2 gloo processes exercise the sharding and the timing reduction of bench.py."""
import os
import subprocess
import sys

from conftest import ROOT

WORKER = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1])
from sculptmate_amd import parallel
rank, local, world = parallel.env_rank_world()
dist = parallel.init("gloo")
assert dist.get_world_size() == world == 2
mine = parallel.shard_indices(7, rank, world)
assert mine == ([0, 2, 4, 6] if rank == 0 else [1, 3, 5])
parallel.barrier()
t = parallel.max_over_ranks(1.0 + rank)          # slowest rank defines the step time
assert t == 2.0, t
counts = parallel.gather_counts((100 + rank, 200 + rank))
assert counts == [[100, 200], [101, 201]], counts
all_items = sorted(i for r in range(world) for i in parallel.shard_indices(7, r, world))
assert all_items == list(range(7))               # every image processed exactly once, no exchange needed
parallel.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_two_rank_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = 29000 + os.getpid() % 1000
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=120)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % r in o


def test_single_process_fallbacks():
    from sculptmate_amd import parallel

    assert parallel.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]
    assert parallel.max_over_ranks(3.5) == 3.5
    assert parallel.gather_counts((1, 2)) == [[1, 2]]


SLAB_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch
from sculptmate_amd import parallel, slab
rank, local, world = parallel.env_rank_world()
dist = parallel.init("gloo")
# two slabs of a tiny hand-made mesh: rank 1 references two boundary vertices of rank 0
top0 = torch.full((2, 2, 3), -1, dtype=torch.int32); top0[0, 1, 2] = 1; top0[1, 0, 0] = 2
parts = [dict(verts=torch.arange(9.).view(3, 3), faces=torch.tensor([[0, 1, 2]]), top=top0, minmax=(-1.0, 1.0)),
         dict(verts=10 + torch.arange(6.).view(2, 3), faces=torch.tensor([[0, 1, -(1 + 0 * 6 + 1 * 3 + 2)], [1, -(1 + 1 * 6 + 0), 0]]),
              top=torch.full((2, 2, 3), -1, dtype=torch.int32), minmax=(-2.0, 0.5))]
v, f = slab.gather_and_assemble(parts[rank], "cpu")
ev, ef = slab.assemble(parts)
assert torch.equal(v, ev) and torch.equal(f, ef)
assert f.tolist() == [[0, 1, 2], [3, 4, 1], [4, 2, 3]], f.tolist()
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_slab_gather_two_rank_gloo(tmp_path):
    """The one exchange step of the 512^3 split (padded all_gather + boundary-vertex resolution) over gloo."""
    script = tmp_path / "slab_worker.py"
    script.write_text(SLAB_WORKER)
    port = 29000 + (os.getpid() + 7) % 1000
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=120)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o


def test_slab_ranges_cover_all_cell_layers():
    from sculptmate_amd import slab

    for R, w in ((512, 8), (256, 3), (9, 16), (2, 2)):
        rs = slab.slab_ranges(R, w)
        assert rs[0][0] == 0 and rs[-1][1] == R - 1 and all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
    assert slab.slab_ranges(512, 8)[0] == (0, 64) and slab.slab_ranges(512, 8)[-1] == (448, 511)
