"""The C++ host executable above the C ABI (dsvt-ai-trt_amd/host/dsvt_detect.cpp: loadWeights + createEngine order + the `-d` loop of
src/dsvt-ai-trt.cpp:532-1970, using only include/dsvt_plugin.h and the HIP runtime) against the Python host (pipeline.py) on the
same weights and frames: the boxes must be the same BITS -- both hosts fold BatchNorm / re-lay out weights in fp32 in the same
order and call the same plugins -- and the .txt files the same text as hostio.save_txt's."""
import os
import subprocess

import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "dsvt-ai-trt_amd", "dsvt_detect")
DEV = "cuda:0"


@pytest.fixture(scope="module")
def wts_file(pkg, tmp_path_factory):
    path = str(tmp_path_factory.mktemp("wts") / "dsvt.wts")
    pkg.synth.write_wts(path, pkg.synth.make_weights())
    return path


def _read_rows(path):
    raw = open(path, "rb").read()
    k = int(np.frombuffer(raw[:4], np.int32)[0])
    return k, np.frombuffer(raw[4:], np.float32).reshape(k, 9)


MODES = {   # dsvt_detect flag -> the Python host's configuration of the same arithmetic
    "--fp32": lambda P: dict(linear_compute=P.COMPUTE_SPLIT),                                   # the default of both hosts: fp32 grade, three fp16 products everywhere
    "--fp8-head": lambda P: dict(linear_compute=P.COMPUTE_SPLIT, head_mx=True),
    "--fp16": lambda P: dict(linear_compute=P.COMPUTE_F16, head_dtype=torch.float16),
}
_PIPES = {}


def _python_boxes(pkg, caps, pts, n, mode="--fp32"):
    key = (mode, caps.N, caps.P)
    if key not in _PIPES:
        _PIPES.clear()                                  # (one Python pipeline alive at a time: the 468 x 468 triple maps are large)
        _PIPES[key] = pkg.pipeline.DsvtPipeline(pkg.synth.make_weights(), caps=caps, device=DEV, device_nms=True, **MODES[mode](pkg.plugin))
    rows, cnt = _PIPES[key].forward(torch.from_numpy(pts[None]).to(DEV), torch.tensor([n], dtype=torch.int32, device=DEV))
    torch.cuda.synchronize()
    k = int(cnt[0])
    return k, rows[0, :k].cpu().numpy().copy()


@pytest.mark.parametrize("mode,graph", [("--fp32", True), ("--fp32", False), ("--fp16", True), ("--fp16", False), ("--fp8-head", True)])
def test_cpp_host_equals_python_host_on_reference_frames(pkg, wts_file, tmp_path, mode, graph):
    """the reference's `-d` loop (src/dsvt-ai-trt.cpp:1884-1970) in C++ above the C ABI, in the precision that passes parity (--fp32, the default:
    the reference's arithmetic is fp32, include/params.h:332) and in the two faster ones: the same BITS as the Python host on the three reference frames"""
    assert os.path.exists(EXE), "dsvt_detect was not built (__graft_entry__.build())"
    out = tmp_path / "out"; out.mkdir()
    cmd = [EXE, "--wts", wts_file, "--data", cases.GOLDEN, "--out", str(out), "--ref-caps", "--dump-raw"] + ([] if graph else ["--no-graph"]) + ([] if mode == "--fp32" else [mode])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    caps = pkg.pipeline.Caps.reference()
    for name in ("000000", "000003", "000004"):
        pts, n = cases.load_frame(name, caps.N)
        k, rows = _python_boxes(pkg, caps, pts, n, mode)
        kc, rc = _read_rows(str(out / f"{name}.rows"))
        assert kc == k and k > 0
        assert np.array_equal(rc.view(np.uint32), rows.view(np.uint32)), float(np.abs(rc - rows).max())      # bit for bit
        # the text file is save_txt's (include/helper.h:441-481): same lines as the Python writer, apart from the time on line 1
        txt = open(out / f"{name}.txt").read().splitlines()
        assert txt[1:] == pkg.hostio.format_results(rows, 0.0).splitlines()[1:]
        assert float(txt[0]) > 0


@pytest.mark.parametrize("mode", ["--fp32", "--fp16"])
def test_cpp_host_on_the_bench_frame(pkg, wts_file, tmp_path, mode):
    """BASELINE configs[2] size through the C++ host: lidar_like(180000, 0) written as a .bin, default (Waymo-sized) caps; the fp32-grade default
    one frame at a time, upload and download included"""
    data = tmp_path / "data"; data.mkdir(); out = tmp_path / "out"; out.mkdir()
    p = pkg.synth.lidar_like(180000, 0)
    p.tofile(data / "000000.bin")
    r = subprocess.run([EXE, "--wts", wts_file, "--data", str(data), "--out", str(out), "--dump-raw", "--repeat", "20", mode], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    caps = pkg.pipeline.Caps()
    pts, n = cases.pad_points(p, caps.N)
    k, rows = _python_boxes(pkg, caps, pts, n, mode)
    kc, rc = _read_rows(str(out / "000000.rows"))
    assert kc == k and np.array_equal(rc.view(np.uint32), rows.view(np.uint32))
    ms = float(open(out / "000000.txt").readline())
    print("dsvt_detect", mode, "180k-point frame, upload + graph launch + download:", ms, "ms;", r.stdout.strip().splitlines()[-1])
    assert ms < (7.0 if mode == "--fp32" else 4.0)


def test_cpp_host_four_frames_per_forward(pkg, wts_file, tmp_path):
    """dsvt_detect --frames 4 (BASELINE configs[3]: four frames per GPU and batch): six 180k / 120k-point clouds = one full forward and a partial one
    (two empty slots); every frame's boxes are the bits of the Python host's single-frame fp32-grade run"""
    data = tmp_path / "data"; data.mkdir(); out = tmp_path / "out"; out.mkdir()
    clouds = [pkg.synth.lidar_like(180000 if i % 2 == 0 else 120000, 30 + i) for i in range(6)]
    for i, p in enumerate(clouds):
        p.tofile(data / f"{i:06d}.bin")
    r = subprocess.run([EXE, "--wts", wts_file, "--data", str(data), "--out", str(out), "--dump-raw", "--frames", "4", "--repeat", "5"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    print(r.stdout.strip().splitlines()[-1])
    caps = pkg.pipeline.Caps()
    for i, p in enumerate(clouds):
        pts, n = cases.pad_points(p, caps.N)
        k, rows = _python_boxes(pkg, caps, pts, n, "--fp32")
        kc, rc = _read_rows(str(out / f"{i:06d}.rows"))
        assert kc == k and k > 0 and np.array_equal(rc.view(np.uint32), rows.view(np.uint32)), i


def test_cpp_host_two_forwards_in_flight(pkg, wts_file, tmp_path):
    """dsvt_detect --in-flight 2 (two engines on two streams, groups round-robin: the copies of one group under the forward of the other): ten clouds in groups of four
    (two full groups + a partial one, so a slot is reused) and in groups of one -- every frame's rows are the bits the synchronous loop writes"""
    data = tmp_path / "data"; data.mkdir()
    for i in range(10):
        pkg.synth.lidar_like(180000 if i % 3 else 90000, 50 + i).tofile(data / f"{i:06d}.bin")
    outs = {}
    for tag, extra in (("sync4", ["--frames", "4"]), ("pipe4", ["--frames", "4", "--in-flight", "2", "--repeat", "3"]), ("pipe1", ["--in-flight", "3", "--repeat", "2"])):
        out = tmp_path / tag; out.mkdir()
        r = subprocess.run([EXE, "--wts", wts_file, "--data", str(data), "--out", str(out), "--dump-raw"] + extra, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        print(tag, r.stdout.strip().splitlines()[-1])
        outs[tag] = [_read_rows(str(out / f"{i:06d}.rows")) for i in range(10)]
        assert all(os.path.exists(out / f"{i:06d}.txt") for i in range(10))
    for i in range(10):
        k, rows = outs["sync4"][i]
        assert k > 0
        for tag in ("pipe4", "pipe1"):
            kc, rc = outs[tag][i]
            assert kc == k and np.array_equal(rc.view(np.uint32), rows.view(np.uint32)), (tag, i)


def test_cpp_host_fails_loudly(wts_file, tmp_path):
    r = subprocess.run([EXE, "--wts", wts_file, "--data", str(tmp_path), "--out", str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no .bin frames" in r.stderr
    r = subprocess.run([EXE, "--wts", str(tmp_path / "missing.wts"), "--data", cases.GOLDEN, "--out", str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "cannot open" in r.stderr


def test_cpp_host_multi_gpu_path_on_one_device(pkg, wts_file, tmp_path):
    """dsvt_detect --gpus N (round 6): frame-batch data parallelism inside the C++ host -- an engine per device, file j of a batch on device j mod N, ONE RCCL gather of
    the result rows to device 0 per batch (the loop shape of BASELINE configs[3]; the reference binds one device, src/dsvt-ai-trt.cpp:1783).  A gpurun box has one GPU:
    --rccl-gather runs that path with a communicator of size 1, four frames per forward, the gather after every batch -- and must write the same BITS as the plain
    loop for every file (six clouds of different sizes: a partial last batch).  --gpus 2 on a one-GPU box must fail loudly, never share a device."""
    data = tmp_path / "data"; data.mkdir()
    for i in range(6):
        pkg.synth.lidar_like(180000 if i % 3 else 90000, 70 + i).tofile(data / f"{i:06d}.bin")
    outs = {}
    for tag, extra in (("plain", []), ("gather", ["--rccl-gather"]), ("gather_no_graph", ["--rccl-gather", "--no-graph"])):
        o = tmp_path / tag; o.mkdir()
        r = subprocess.run([EXE, "--wts", wts_file, "--data", str(data), "--out", str(o), "--frames", "4", "--dump-raw", "--repeat", "3"] + extra, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        if extra:
            assert "one RCCL gather per batch of 4 frames" in r.stdout
        outs[tag] = [_read_rows(str(o / f"{i:06d}.rows")) for i in range(6)]
    for tag in ("gather", "gather_no_graph"):
        for (k0, r0), (k1, r1) in zip(outs["plain"], outs[tag]):
            assert k0 == k1 and k0 > 0 and np.array_equal(r0.view(np.uint32), r1.view(np.uint32)), tag
    if torch.cuda.device_count() < 2:
        r = subprocess.run([EXE, "--wts", wts_file, "--data", str(data), "--out", str(tmp_path / "plain"), "--gpus", "2"], capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "visible" in r.stderr and "frames/s" not in r.stdout
