"""The one collective of the path on the GPU box: a world-size-1 "nccl" (= RCCL) communicator whose gather really runs, so that
the RCCL code path is exercised even though gpurun boxes have a single MI355X; and bench.py's own multi-rank
launcher refusing to report more GPUs than exist."""
import json
import os
import subprocess
import sys

import pytest

from tests.test_parallel_cpu import run_single_rank

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gather_single_rank_group_rccl():
    run_single_rank("nccl")


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` spawns N ranks itself; with fewer than N devices it must fail loudly, never print a line"""
    import torch
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "n_gpus" not in r.stdout
    assert "visible" in (r.stderr + r.stdout)


def test_bench_single_rank_collective_line():
    """bench.py --rccl-single: N = 1 with a size-1 RCCL communicator; the gather is inside the timed region and the line says so"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--rccl-single",
                        "--no-cpu-baseline", "--no-kernel-events"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["result_gather"] == "rccl gather (communicator of size 1)"
    assert line["config"]["graph_replay_equals_eager"] is True
    assert line["value"] > 0
