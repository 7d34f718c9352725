"""bench.py's multi-rank control flow on a 1-GPU box: two ranks launched exactly as the driver launches them
(python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ...), both on cuda:0, the script's barrier / MAX collectives
over gloo (RCCL refuses two ranks on one device; the RCCL calls themselves are covered with one rank in test_gpu_slab.py).
Checks what the scaling run relies on: every rank gets through, rank 0 alone prints ONE JSON line, value = 2 x steps / time."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_two_ranks_one_json_line(cuda):
    env = dict(os.environ, SCULPT_BENCH_SHARE_GPU="1", SCULPT_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    assert d["cpu_baseline"] is None and "boundary" not in d and "sf3d" not in d   # N = 1 extras stay out of the N > 1 line
    assert d["config"]["parallelism"].startswith("dp2")
    assert d["config"]["collective_ranks"] == 2 and d["config"]["collective_backend"] == "gloo" and d["config"]["shared_gpu"] is True
    assert d["config"]["decoder_filter"] is True and d["density"]["filtered"]["mesh_identical"] is True
    # refusing a mismatched launch: --gpus 2 without two ranks
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=dict(os.environ), cwd=ROOT)
    assert q.returncode != 0 and "needs 2 ranks" in (q.stderr + q.stdout)
