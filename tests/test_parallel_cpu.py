"""Frame-batch data parallelism on CPU: world_size 2, gloo.  Covers the sharding rule and the one
collective of the path (the result gather), with uneven frame counts."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _fake_frame(par, f):
    """A recognisable per-frame result row: boxes filled with f + small offsets, count = f + 1."""
    row = torch.zeros(par.ROW)
    boxes = torch.arange(4500, dtype=torch.float32).reshape(500, 9) * 1e-3 + f
    par.pack_result(boxes, torch.tensor([f + 1], dtype=torch.int32), row)
    return row


def _worker(rank, world, port, n_frames, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("par", os.path.join(ROOT, "dsvt-ai-trt_amd", "parallel.py"))
    par = importlib.util.module_from_spec(spec); spec.loader.exec_module(par)
    r, _, w = par.init(backend="gloo")
    ids = par.shard_frames(n_frames, r, w)
    local = torch.stack([_fake_frame(par, f) for f in ids]) if ids else torch.zeros((0, par.ROW))
    par.barrier()
    out = par.gather_results(local, n_frames, r, w)
    t = par.max_over_ranks(float(r + 1), torch.device("cpu"))
    ok = abs(t - w) < 1e-9
    # bench.py's timing protocol: the per-repeat times of every rank meet in ONE all-reduce (element-wise maximum)
    tv = par.max_over_ranks_vec([1.0 + r, 5.0 - 2 * r, 0.25], torch.device("cpu"))
    ok = ok and tv == [float(w), 5.0, 0.25]
    # ... and since round 5 in ONE all-gather that keeps the per-rank values (which rank was slow)
    av = par.all_ranks_vec([1.0 + r, 5.0 - 2 * r], torch.device("cpu"))
    ok = ok and av == [[1.0 + k, 5.0 - 2 * k] for k in range(w)]
    if r == 0:
        ok = ok and par.own_rows_match(out, local, n_frames, r, w)
        bad = out.clone(); bad[0, 7] += 1.0
        ok = ok and not par.own_rows_match(bad, local, n_frames, r, w)
    # the product loop's form (round 6): buffers allocated once (parallel.GatherBuffers), a gather after every batch -- three batches whose rows differ --,
    # through dist.gather and through the all-gather form; every call returns the same rows as the allocating call
    gb = par.GatherBuffers(n_frames, r, w, torch.device("cpu"))
    for coll in ("gather", "all_gather"):
        for batch in range(3):
            shifted = local + float(batch)
            got = par.gather_results(shifted, n_frames, r, w, buffers=gb, collective=coll)
            ref = par.gather_results(shifted, n_frames, r, w)
            ok = ok and ((got is None and ref is None) if r != 0 else (got is gb.out and torch.equal(got, ref)))
    if r == 0:
        for f in range(n_frames):
            b, c = par.unpack_result(out[f])
            ok = ok and c == f + 1 and abs(float(b[0, 0]) - f) < 1e-6 and abs(float(b[499, 8]) - (f + 4.499)) < 1e-3
        ok = ok and out.shape == (n_frames, par.ROW)
    else:
        ok = ok and out is None
    q.put((r, bool(ok)))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_frames", [5, 8, 1])
def test_gather_world2_gloo(n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == {0: True, 1: True}


def _single_rank_worker(backend, port, q):
    """world size 1, a real communicator: the gather goes through the collective library (gloo here, RCCL in
    tests/test_parallel_gpu.py) instead of the single-process shortcut"""
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import importlib.util
    spec = importlib.util.spec_from_file_location("par", os.path.join(ROOT, "dsvt-ai-trt_amd", "parallel.py"))
    par = importlib.util.module_from_spec(spec); spec.loader.exec_module(par)
    r, _, w = par.init(backend=backend, single_rank_group=True)
    dev = torch.device("cuda:0" if backend == "nccl" else "cpu")
    ok = torch.distributed.is_initialized() and torch.distributed.get_backend() == backend and (r, w) == (0, 1)
    local = torch.stack([_fake_frame(par, f) for f in range(3)]).to(dev)
    par.barrier()
    out = par.gather_results(local, 3, r, w, force_collective=True)
    ok = ok and out is not local and torch.equal(out.cpu(), local.cpu())
    ok = ok and par.max_over_ranks(2.5, dev) == 2.5 and par.max_over_ranks_vec([1.5, 0.5], dev) == [1.5, 0.5]
    q.put((0, bool(ok)))
    torch.distributed.destroy_process_group()


def run_single_rank(backend):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_single_rank_worker, args=(backend, _free_port(), q))
    p.start()
    res = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0 and res == (0, True)


def test_gather_single_rank_group_gloo():
    run_single_rank("gloo")


def test_shard_frames_partition():
    import importlib.util
    spec = importlib.util.spec_from_file_location("par", os.path.join(ROOT, "dsvt-ai-trt_amd", "parallel.py"))
    par = importlib.util.module_from_spec(spec); spec.loader.exec_module(par)
    for n in (0, 1, 7, 32):
        for w in (1, 2, 4, 8):
            parts = [par.shard_frames(n, r, w) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert par.ROW == 4501


def test_bench_launcher_fails_loudly_without_gpus():
    """`python bench.py --gpus N` launches its own ranks; where fewer than N GPUs are visible (none in the build container) it must exit non-zero
    with a message and print no JSON line (never an `n_gpus` that is not the number of ranks that ran)"""
    import subprocess, sys
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("needs a host with fewer than two GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "visible" in (r.stdout + r.stderr) and "n_gpus" not in r.stdout
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "n_gpus" not in r.stdout          # WORLD_SIZE disagrees with --gpus (or no GPU at all): refused either way
