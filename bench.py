"""bench.py -- frames/s + p50 per-frame ms of the DSVT hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step = one frame through the whole pipeline (BASELINE.json configs[2]: Waymo-shaped 180k-point synthetic cloud
`lidar_like(180000, seed)`, 0.32 m pillars, 468x468 BEV grid, full 4-block DSVT pillar backbone + BEV backbone + CenterHead +
top-K decode + FilterBoxByScore + rotated NMS), inputs already resident in HBM when the timed region starts.  Four frames per
forward() (their pillar rows concatenated: one launch per backbone layer for all of them, `--batch`; BASELINE configs[3] puts four
frames on each GPU per batch) on each of two streams (`--streams`) are the default; `--batch 1 --streams 1` is the latency mode (the
reference's own: one frame at a time).  Frame-batch data parallelism: every rank processes K frames of its own (weak scaling) and the
per-frame results are gathered to rank 0 with ONE collective inside the timed region.  Rank 0 prints one JSON line.

Three precision modes are timed by a default run (N = 1):
  value / ms_per_step / p50_ms   `--dtype split` (default): the fp32-grade mode -- every GEMM and convolution on (hi, lo) fp16 operand pairs (three
                v_mfma_f32_16x16x32_f16 per product), fp32 accumulate, fp32 tensors in the DSVT stage.  The reference's arithmetic is fp32
                (include/params.h:332); ALL NINE box columns (yaw included) sit within 1e-3 of the fp32 oracle on every check cloud (measured ~1e-4 worst):
                the mode that answers north_star's joint target (>= 200 frames/s AND 1e-3).  Round 4's headline ran the convolutions' correction products
                on the fp8 scaled MFMA: its yaw tail crosses 1e-3 on ill-conditioned boxes, so since round 5 it is `fp8_head_mode` below, not `value`.
  fast_mode     the same frames through the fp16 pipeline (BASELINE configs[2] says "fp16"): 2.2 x the frames/s, boxes 2e-3 .. 4e-3 from the oracle on
                z / size -- OUTSIDE the 1e-3 bar, so it is reported beside the headline, not as it.
  fp8_head_mode  the fp32-grade frame with one fp16 product + two e4m3 correction products (v_mfma_scale_f32_16x16x128_f8f6f4) in the convolutions
                (`DsvtPipeline(head_mx=True)`): ~20 % more frames/s, centres / sizes / scores ~1e-4, yaw above 1e-3 on single boxes -- reported, not claimed.
  box_err_vs_oracle (inside cpu_baseline, where the oracle runs as the checker) the WORST box error of each mode over eight check clouds
                (seeds 21, 9, 3, 1, 0, 7, 16, 23), per column and over all nine, run through the timed pipelines.
  targets       which operating point meets which bar (>= 200 frames/s; <= 5 ms p50 per frame), and whether one point meets both.

Timing: K steps per repeat, every repeat between two torch.cuda.synchronize() on every rank; R = max(3, min(15, ceil(300 / K))) repeats (a function
of K only).  Collectives: one all-gather of the ranks' device identities at start-up (n_gpus = DISTINCT devices, checked), one barrier before the first
repeat, the RESULT GATHER AFTER EVERY BATCH (one forward on every rank = frames_per_forward frames per rank: configs[3]'s 32 frames over 8 GPUs; `--gather-every-batch`, the default
whenever a communicator exists: N > 1 and `--rccl-single`) inside every repeat -- the product loop of BASELINE configs[3], buffers allocated once (round 5
gathered once, in the last repeat: `--gather-once`) --, rank 0 checks its own rows in the gathered tensor bit for bit, one barrier after the last repeat, ONE
all-gather of the R per-repeat times (`repeat_values_per_rank`); `value` = total frames / median over the repeats of the max-over-ranks time -- the same protocol for
every N, so N = 8 / N = 1 compares like with like (a communicator-free N = 1 run has no collective to time: `gathers_per_repeat` says which it was).  The roofline sample (one eager forward with HIP events around every launch) is a pre-pass
outside every timed region.

Extra objects in the line:
  single_frame_mode  (N = 1) the same kernels with ONE frame per forward and one in flight -- the reference's own mode: frames/s and p50, both modes.
  host_input_mode    (N = 1) the reference's own timed bracket (src/dsvt-ai-trt.cpp:1918-1956): upload from pinned host memory + the frame + download
                of the final boxes, one frame at a time, host clock around each frame -- the PCIe-inclusive figure, never `value`.
  roofline      the kernel family with the LARGEST share of the frame (the convolutions): algorithmic flops (bytes) per launch / average launch
                duration, measured with HIP events around every launch of the sampled forward, against the dense fp16 MFMA peak (8 TB/s HBM peak).
  roofline_hot_path  the same object for the largest of SURVEY 8(a)'s own rows (the DSVT stage: encoder MLP / QKV / set attention / PFN).
  roofline_other_kernels  the other kernel families, incl. the scatter / gather stages north_star names (voxelizer chain, set partition,
                Map2Bev) priced with SURVEY 8(d)'s algorithmic bytes.
  cpu_baseline  SURVEY 8(d): the reference's HOST path -- loadData + save_result + nms_cpu (include/helper.h:28-72, 257-283, 470-481;
                restated in oracle/dsvt_oracle.c) -- timed single-threaded (the reference is) on the host cores of the same box, fed the
                FilterBoxByScore rows the GPU produced for the same frames; plus a frame-parallel variant, the CPU voxelize + partition
                restatement and the whole network on the CPU oracle (which also yields the reference boxes for box_err_vs_oracle).

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches the N ranks itself (torch.distributed.run, one
process per GPU) and fails if fewer than N GPUs are visible.  `--share-gpu` lets N ranks share the visible GPU(s) (a launcher / RCCL
dry run: the line says so and is not a scaling number).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

PEAK_F32_MATRIX_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_F16_MATRIX_TFLOPS = 2500.0     # same guide: "Peak BF16/FP16 MFMA ~2.5 PF dense"
PEAK_HBM_GBS = 8000.0               # same guide: "HBM3E peak BW 8.0 TB/s spec" (6.29 TB/s measured float4 copy)
N_POINTS = 180000
# FETCH_SIZE / WRITE_SIZE passes (profiles/), by (mode, frames per forward())
PMC_FILES = {("f16", 1): "r02_g_batch1_pmc_traffic.json", ("f16", 2): "r02_g_pmc_traffic.json", ("f16", 4): "r04_f16_pmc_traffic.json",
             ("split", 4): "r06_split_pmc_traffic.json", ("split", 1): "r05_split_batch1_pmc_traffic.json",
             ("splitmx", 4): "r04_split_pmc_traffic.json"}          # (round 4's split frame had the fp8 head: today's `splitmx`)
FRAME_POOL = 4                      # distinct synthetic clouds cycled through by the steps
MIN_TIMED_S = 0.5                   # K steps shorter than this are repeated
MAX_REPEATS = 15


def cpu_model():
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                return l.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def box_errors(got, n_got, exp, n_exp):
    """per-field max abs error over rows matched by (class, nearest centre within 0.2 m): xy, z, size, score, yaw; fraction matched"""
    got, exp = got[:n_got], exp[:n_exp]
    used = np.zeros(n_got, bool)
    errs, matched = [], 0
    for e in exp:
        if n_got == 0:
            break
        d = np.abs(got[:, :2] - e[:2]).max(1) + (got[:, 7] != e[7]) * 1e3 + used * 1e3
        j = int(np.argmin(d))
        if d[j] > 0.2:
            continue
        used[j] = True; matched += 1
        d = np.abs(got[j] - e)
        if min(abs(got[j][6]), abs(e[6])) > np.pi / 2 - 0.02:      # yaw = atan(sin/cos) lives in (-pi/2, pi/2): +-pi/2 are the same heading (wrapped only AT the ends)
            d[6] = min(d[6], abs(np.pi - d[6]))
        errs.append(d)
    if not errs:
        return None
    m = np.array(errs).max(0)
    return dict(xy=float(m[:2].max()), z=float(m[2]), size=float(m[3:6].max()), yaw=float(m[6]), score=float(m[8]),
                max_xyz_size_score=float(max(m[:6].max(), m[8])), matched=round(matched / max(n_exp, 1), 4), boxes=int(n_got), oracle_boxes=int(n_exp))


def cpu_baseline(caps, frames, n_parallel_frames=32, whole_network=None, mode_rows=None, check_clouds=None, n_points_key=N_POINTS, live_seed=0):
    """SURVEY 8(d).  frames: [(points [n,4] float32 numpy, FilterBoxByScore rows [500,9] float32 numpy from the GPU, count)].
    value = frames/s of the reference's host path, ONE thread: loadData (read the .bin, size check, zero-pad to the cap) +
    save_result + nms_cpu on the GPU's own rows.  whole_network = (weights,): the whole network of frame 0 on the CPU oracle;
    check_clouds = [(seed, points)], mode_rows = {mode: [(rows, count) per check cloud]}: FilterBoxByScore rows of every precision mode on the CHECK clouds
    (run through the timed pipelines) -> box_err_vs_oracle = the worst error per column over all of them against the oracle's rows."""
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.lib()
    tmp = tempfile.mkdtemp(prefix="dsvt_bench_")
    paths = []
    for i, (pts, _, _) in enumerate(frames):
        paths.append(os.path.join(tmp, f"{i:06d}.bin"))
        pts.astype(np.float32).tofile(paths[-1])

    def host_path(i):
        with open(paths[i % len(frames)], "rb") as fh:
            raw = fh.read()
        O.load_data(raw, caps.N)                                        # helper.h:28-72 + the zero-padded copy (:1909)
        _, rows, cnt = frames[i % len(frames)]
        return len(O.nms_cpu(rows, cnt, 0.01)[1])                         # save_result + nms_cpu (helper.h:257-283, 470-481)

    host_path(0)
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 3.0 or reps < len(frames):
        kept = host_path(reps); reps += 1
    t_1 = (time.perf_counter() - t0) / reps
    cores = os.cpu_count() or 1
    nthr = min(cores, n_parallel_frames)
    with ThreadPoolExecutor(nthr) as ex:                                # ctypes releases the GIL inside the C restatement
        list(ex.map(host_path, range(nthr)))
        t0 = time.perf_counter()
        rounds = 0
        while time.perf_counter() - t0 < 3.0:
            list(ex.map(host_path, range(n_parallel_frames))); rounds += 1
        t_par = (time.perf_counter() - t0) / (rounds * n_parallel_frames)
    # context: the CPU restatement of the voxelizer + both partitions, one core
    from oracle import dense_ref as D
    cfg = D.OracleCfg(max_points=caps.N, max_points_filter=caps.Nk, max_pillars=caps.P, max_win=caps.W,
                      max_vox_per_win=caps.Vw, max_sets=caps.S)
    pts0 = np.zeros((caps.N, 4), np.float32); n0 = frames[0][0].shape[0]; pts0[:n0] = frames[0][0]
    t0 = time.perf_counter()
    vox = O.points2features(pts0, n0, cfg.p2f)
    for wc, gc in zip(cfg.wp, cfg.gs):
        wp = O.window_partition(vox["coords"], vox["P"], wc)
        O.get_set(wp["gidx"], wp["cinw"], wp["vcnt"], wp["W"], gc)
    t_pre = time.perf_counter() - t0
    out = dict(value=round(1.0 / t_1, 2), unit="frames/s", cores=1, kind="port", cpu_model=cpu_model(), host_cores=cores,
               definition="v2 (rounds 2+): the reference's HOST path per SURVEY 8(d) -- loadData + save_result + nms_cpu, one thread; round 1's key of the same "
                          "name timed the whole network on the CPU oracle, which is `whole_network_port` here",
               sample=f"reference host path (loadData of a {frames[0][0].shape[0]}-point .bin zero-padded to {caps.N} + save_result + "
                      f"nms_cpu on the {frames[0][2]} FilterBoxByScore rows the GPU produced), {reps} frames, 1 thread",
               ms_per_frame=round(1e3 * t_1, 3), nms_kept=int(kept),
               frame_parallel=dict(value=round(1.0 / t_par, 1), unit="frames/s", threads=nthr, frames=n_parallel_frames,
                                   note="same host path, one frame per thread"),
               preprocess_voxelize_partition_ms_1core=round(1e3 * t_pre, 2))
    if whole_network is not None:
        weights, = whole_network
        t0 = time.perf_counter()
        boxes, cnt = D.forward(pts0, n0, weights, cfg)
        t_frame = time.perf_counter() - t0
        out["whole_network_port"] = dict(value=round(1.0 / t_frame, 4), unit="frames/s", frame_ms=round(1e3 * t_frame, 1),
                                         note=f"the whole network of frame 0 in fp32 on the CPU oracle, dense layers on {torch.get_num_threads()} "
                                              "torch threads, plugin restatement on 1 core", boxes=int(cnt))
        if mode_rows and check_clouds:
            # the oracle as the CHECKER of the timed modes: FilterBoxByScore rows (before NMS, like the reference engine's output) of every check
            # cloud -- the 24-cloud sweep's ill-conditioned one (seed 21) among them --, worst error per column over all of them
            errs = {m: [] for m in mode_rows}
            from tests import golden_oracle as GO
            src = {"golden": 0, "live": 0}
            for ci, (seed, cp) in enumerate(check_clouds):
                pc = np.zeros((caps.N, 4), np.float32); pc[:cp.shape[0]] = cp
                # the oracle's rows for this cloud: the committed fixture (tests/golden/oracle_boxes.npz: made by tools/make_golden.py from the live oracle, refused when
                # its recorded inputs differ from these points / weights / caps); a cloud without a fixture runs the live oracle (~10 s)
                try:
                    ob, oc = GO.forward(f"lidar{n_points_key}s{seed}", pc, cp.shape[0], weights, caps); src["golden"] += 1
                except (KeyError, FileNotFoundError):
                    ob, oc = D.forward(pc, cp.shape[0], weights, cfg); src["live"] += 1
                if seed == live_seed:          # frame 0 ran on the LIVE oracle above: the fixture of the same cloud must be its rows
                    e_ = box_errors(np.asarray(ob), int(oc), np.asarray(boxes), int(cnt))       # (rows matched by class + centre: two candidates whose scores differ by 1e-7 may swap ranks between hosts)
                    out["whole_network_port"]["golden_fixture_vs_live_oracle"] = None if e_ is None else dict(max_abs_all_nine_columns=max(e_["max_xyz_size_score"], e_["yaw"]), matched=e_["matched"],
                                                                                                               rows=(int(oc), int(cnt)))
                for m, rows in mode_rows.items():
                    errs[m].append(box_errors(rows[ci][0], rows[ci][1], ob, int(oc)))
            cols = ("xy", "z", "size", "yaw", "score", "max_xyz_size_score")
            out["box_err_vs_oracle"] = {}
            for m, es in errs.items():
                es_ = [e for e in es if e]
                w_ = {k: max(e[k] for e in es_) for k in cols} if es_ else None
                if w_:
                    w_.update(all_nine_columns=max(w_["max_xyz_size_score"], w_["yaw"]), matched=min(e["matched"] for e in es_), clouds=len(es_),
                              per_cloud_worst=[round(max(e["max_xyz_size_score"], e["yaw"]), 7) for e in es_])
                out["box_err_vs_oracle"][m] = w_
            out["box_err_vs_oracle"]["seeds"] = [sd for sd, _ in check_clouds]
            out["box_err_vs_oracle"]["oracle_rows_from"] = src
            out["box_err_vs_oracle"]["note"] = ("WORST absolute error per column over the check clouds (lidar_like(points, seed) for the listed seeds, run through the timed "
                                                "pipelines: same kernels, same frames per forward) of the FilterBoxByScore rows against the fp32 CPU oracle, rows matched by "
                                                "class + nearest centre; all_nine_columns = max over x, y, z, the sizes, yaw, score (the class matches by construction); "
                                                "north_star's bar: centres / sizes / scores within 1e-3 -- the headline mode holds it on all nine")
    for p_ in paths:
        os.remove(p_)
    os.rmdir(tmp)
    return out


def cpp_host_mode(pkg, weights, clouds, repeat=12):
    """the reference's `-d` loop in C++ above the C ABI (dsvt-ai-trt_amd/host/dsvt_detect.cpp; nothing but include/dsvt_plugin.h + the HIP runtime), in the
    headline precision, on the bench clouds written as .bin files: per frame the upload of n x 16 bytes from pinned memory + one HIP-graph launch + the
    download of the final boxes, host clock around it (src/dsvt-ai-trt.cpp:1918-1956) -- one frame at a time and four frames per forward"""
    import re, shutil, subprocess, tempfile
    exe = os.path.join(G.PKG_DIR, "dsvt_detect")
    if not os.path.exists(exe):
        return {"error": "dsvt_detect is not built"}
    tmp = tempfile.mkdtemp(prefix="dsvt_cpp_")
    try:
        wts = os.path.join(tmp, "dsvt.wts"); data = os.path.join(tmp, "data"); os.makedirs(data)
        pkg.synth.write_wts(wts, weights)
        for i, cl in enumerate(clouds):
            cl.astype(np.float32).tofile(os.path.join(data, f"{i:06d}.bin"))
        out = {"binary": "dsvt-ai-trt_amd/dsvt_detect --fp32 (default): f16x3 everywhere, the headline's arithmetic; boxes bit-identical to the Python host "
                         "(tests/test_host_executor_gpu.py)", "clouds": len(clouds), "repeat": repeat}
        for key, extra in (("one_frame_per_forward", []), ("four_frames_per_forward", ["--frames", "4"])):
            o = os.path.join(tmp, "out_" + key); os.makedirs(o)
            r = subprocess.run([exe, "--wts", wts, "--data", data, "--out", o, "--repeat", str(repeat)] + extra, capture_output=True, text=True, timeout=600)
            m = re.search(r"([0-9.]+) ms per frame, ([0-9.]+) frames/s", r.stdout)
            out[key] = ({"ms_per_frame": float(m.group(1)), "value": float(m.group(2)), "unit": "frames/s"} if (r.returncode == 0 and m)
                        else {"error": (r.stdout + r.stderr)[-300:]})
        # --in-flight 2: two engines on two streams, groups round-robin (the copies of one group under the forward of the other); sixteen files = the clouds four times
        data2 = os.path.join(tmp, "data16"); os.makedirs(data2)
        for j in range(16):
            os.symlink(os.path.join(data, f"{j % len(clouds):06d}.bin"), os.path.join(data2, f"{j:06d}.bin"))
        for key, extra in (("one_frame_per_forward_two_in_flight", ["--in-flight", "2"]), ("four_frames_per_forward_two_in_flight", ["--frames", "4", "--in-flight", "2"])):
            o = os.path.join(tmp, "out_" + key); os.makedirs(o)
            r = subprocess.run([exe, "--wts", wts, "--data", data2, "--out", o, "--repeat", "4"] + extra, capture_output=True, text=True, timeout=600)
            m = re.search(r"([0-9.]+) ms per frame, ([0-9.]+) frames/s", r.stdout)
            out[key] = ({"ms_per_frame": float(m.group(1)), "value": float(m.group(2)), "unit": "frames/s"} if (r.returncode == 0 and m)
                        else {"error": (r.stdout + r.stderr)[-300:]})
        out["note"] = ("PCIe upload / download and the host's launch inside every figure; the first two: one forward in flight (the reference's loop is synchronous), host clock around each group; "
                       "*_two_in_flight: dsvt_detect --in-flight 2, wall clock of a whole pass over sixteen files (host copy into pinned memory inside too)")
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def other_configs(pkg, weights, dev):
    """The single-GPU configurations of BASELINE.json besides the one `value` is quoted on, timed in the same driver run (missing 4):
      configs[1]  lidar_like(60000, 0): voxelize + WindowPartition_0 + GetSet_0 + ONE DSVT block, fp32 (v_mfma_f32_16x16x4_f32), HIP-graph replay: ms per cloud, and the
                  set-attention kernel priced against the fp32-matrix peak (parity: tests/test_pipeline_gpu.py::test_config1_60k_cloud_one_block_fp32)
      configs[4]  lidar_like(300000, 0): the two-stage 3-D voxel backbone (pipeline3d.Dsvt3dBackbone, 468 x 468 x 32 -> 468 x 468 x 8), HIP-graph replay: ms per
                  cloud, and the voxelizer / partition launches priced in GB/s (parity: tests/test_voxel3d_gpu.py)"""
    P = pkg.plugin
    out = {}

    def replay_ms(fwd, ins, reps=24):
        sin = tuple(torch.zeros_like(t) for t in ins[0])
        for t, v in zip(sin, ins[0]):
            t.copy_(v)
        for _ in range(3):
            fwd(*sin)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fwd(*sin)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(reps):
            for t, v in zip(sin, ins[i % len(ins)]):
                t.copy_(v)
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def plugin_rows(fwd, ins):
        P.PROFILE = {k: [] for k in P.plugin_types()}
        fwd(*ins); torch.cuda.synchronize()
        prof, P.PROFILE = P.PROFILE, None
        return {k: (sum(e0.elapsed_time(e1) for e0, e1, _ in v) * 1e3, len(v)) for k, v in prof.items() if v}

    # configs[1]
    caps = pkg.pipeline.Caps(65536, 65536, 32768, 2048, 576)
    pipe = pkg.pipeline.DsvtPipeline(weights, caps=caps, blocks=1, with_head=False, device=dev, linear_compute=P.COMPUTE_F32)
    ins = []
    for sd in range(2):
        p = pkg.synth.lidar_like(60000, sd); buf = np.zeros((1, caps.N, 4), np.float32); buf[0, :len(p)] = p
        ins.append((torch.from_numpy(buf).to(dev), torch.tensor([len(p)], dtype=torch.int32, device=dev)))
    fwd = lambda pts, n: pipe.backbone(pipe.voxel_stage(pts, n))
    st = pipe.voxel_stage(*ins[0]); torch.cuda.synchronize()
    Pn, Nk, S = int(st["P"][0]), int(st["Nk"][0]), [int(g[2][0]) for g in st["gss"]]
    ms = replay_ms(fwd, ins)
    rows = plugin_rows(fwd, ins[0])
    att_us, att_n = rows.get("DsvtSetAttentionPlugin", (0.0, 0))
    att_fl = sum(4.0 * 36 * 36 * 192 * S[l % len(S)] for l in range(att_n))                      # QK^T + AV of every set, both layers (layer l on window configuration l)
    out["configs[1]"] = dict(workload="lidar_like(60000, 0): voxelize + WindowPartition_0 + GetSet_0 + one DSVT block (two encoder layers), fp32 (COMPUTE_F32: v_mfma_f32_16x16x4_f32), HIP-graph replay",
                             ms_per_cloud=round(ms, 4), clouds_per_s=round(1e3 / ms, 1), counts=dict(P=Pn, Nk=Nk, S=S),
                             set_attention=dict(kernel="set_attention_kernel (v_mfma_f32_16x16x4_f32, fp32 I/O)", launches=att_n, us_per_launch=round(att_us / max(att_n, 1), 2),
                                                achieved=round(att_fl / max(att_us, 1e-9) / 1e6, 3), peak=PEAK_F32_MATRIX_TFLOPS, unit="TFLOP/s",
                                                frac=round(att_fl / max(att_us, 1e-9) / 1e6 / PEAK_F32_MATRIX_TFLOPS, 4),
                                                hbm_gbs=round(att_n * Pn * 192 * 4 * 4 / max(att_us, 1e-9) / 1e3, 1),
                                                note="algorithmic flops = 4 x 36 x 36 x 192 per set (QK^T + AV); bound by HBM / latency at this size, not by the fp32 matrix pipe"),
                             plugin_us={k: round(v[0], 1) for k, v in sorted(rows.items(), key=lambda kv: -kv[1][0])})
    del pipe
    # configs[4]
    net = pkg.pipeline3d.Dsvt3dBackbone(pkg.synth.make_weights_3d(), device=dev)
    ins = []
    for sd in range(2):
        p = pkg.synth.lidar_like(300000, sd); buf = np.zeros((1, net.N, 4), np.float32); buf[0, :len(p)] = p
        ins.append((torch.from_numpy(buf).to(dev), torch.tensor([len(p)], dtype=torch.int32, device=dev)))
    x, coords, Pl = net.forward(*ins[0]); torch.cuda.synchronize()
    ms = replay_ms(net.forward, ins)
    rows = plugin_rows(net.forward, ins[0])
    npts = int(ins[0][1][0])
    p2f_us = rows.get("Points2FeaturesPlugin", (0.0, 0))[0]
    out["configs[4]"] = dict(workload="lidar_like(300000, 0): two-stage 3-D voxel DSVT backbone (468 x 468 x 32 voxels -> block over 12 x 12 x 32 windows -> pooling by (1, 1, 4) -> block over "
                                      "12 x 12 x 8 windows), split precision, HIP-graph replay",
                             ms_per_cloud=round(ms, 4), clouds_per_s=round(1e3 / ms, 1), voxels_last_stage=int(Pl[0]),
                             voxelizer=dict(us=round(p2f_us, 1), algorithmic_mb=round((16.0 * npts + 44.0 * npts) / 1e6, 2), gbs=round((16.0 * npts + 44.0 * npts) / max(p2f_us, 1e-9) / 1e3, 1),
                                            peak=PEAK_HBM_GBS, note="16 N bytes read + 44 Nk written (Nk <= N: upper bound on the bytes, so on the GB/s)"),
                             plugin_us={k: round(v[0], 1) for k, v in sorted(rows.items(), key=lambda kv: -kv[1][0])})
    return out


def spawn_ranks(n, share_gpu):
    """`python bench.py --gpus N` outside torchrun: launch the N ranks (one process per GPU) and relay rank 0's JSON line"""
    import socket
    import subprocess
    ngpu = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ngpu < n and not (share_gpu and ngpu >= 1):
        raise SystemExit(f"bench.py --gpus {n}: only {ngpu} GPU(s) visible")
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


class ModeRun:
    """NS pipelines of one precision mode on NS streams: warm-up, graph capture, replay == eager check, the timed K-step loop."""

    def __init__(self, pkg, par, args, mode, dev, caps, FB, weights, pool, host_pool, world, rank):
        P = pkg.plugin
        self.pkg, self.par, self.args, self.mode, self.dev, self.caps, self.FB, self.pool, self.host_pool = pkg, par, args, mode, dev, caps, FB, pool, host_pool
        self.world, self.rank = world, rank
        self.NS = NS = max(1, args.streams)
        self.use_graph = not args.no_graph
        kw = {"f16": dict(linear_compute=P.COMPUTE_F16, head_dtype=torch.float16), "split": dict(linear_compute=P.COMPUTE_SPLIT),
              "splitmx": dict(linear_compute=P.COMPUTE_SPLIT, head_mx=True), "f32": dict(linear_compute=P.COMPUTE_F32)}[mode]
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
        self.pipes = [pkg.pipeline.DsvtPipeline(weights, caps=caps, device=dev, device_nms=not args.no_nms, frames=FB, **kw) for _ in range(NS)]
        self.static_in = [(torch.zeros_like(pool[0][0]), torch.zeros_like(pool[0][1])) for _ in range(NS)]
        self.scratch = [torch.zeros((FB, par.ROW), dtype=torch.float32, device=dev) for _ in range(NS)]
        self.replay_equals_eager = None
        self.collective = world > 1 or args.rccl_single
        # --gather-every-batch: the buffers of the per-batch gather exist before any graph is captured
        self.batch_buffers = par.GatherBuffers(FB * world, rank, world, dev, staged=(torch.distributed.is_initialized() and torch.distributed.get_backend() == "gloo")) if (self.collective and args.gather_every_batch) else None
        self.gathered_all = torch.zeros((args.steps * world, par.ROW), dtype=torch.float32, device=dev) if (self.batch_buffers is not None and rank == 0) else None

    def pack(self, boxes, cnt, rows):
        """boxes [FB,500,9], cnt [FB] -> FB rows of the result buffer (two device ops, no host sync)"""
        if self.FB == 1:
            self.par.pack_result(boxes[0], cnt, rows[0])
        else:
            rows[:, :self.par.ROW - 1].copy_(boxes.reshape(self.FB, -1)); rows[:, self.par.ROW - 1].copy_(cnt.to(torch.float32))

    def run_frame(self, i, row, eager=False):
        """forward() call i (FB frames) on pipeline/stream i % NS (must be called with that stream current)"""
        s = i % self.NS
        pts, n = self.pool[i % len(self.pool)]
        pipe, sin = self.pipes[s], self.static_in[s]
        if self.host_pool is not None:
            hp, hn, k = self.host_pool[i % len(self.pool)]
            sin[0][0, :k].copy_(hp, non_blocking=True); sin[1].copy_(hn, non_blocking=True)
            boxes, cnt = pipe.replay() if (self.use_graph and not eager) else pipe.forward(*sin)
        elif self.use_graph and not eager:
            sin[0].copy_(pts); sin[1].copy_(n)        # device-to-device refill of the graph's inputs
            boxes, cnt = pipe.replay()
        else:
            boxes, cnt = pipe.forward(pts, n)
        self.pack(boxes, cnt, row)

    def prepare(self):
        a = self.args
        # every pipeline's FIRST forward runs on the process's current stream, before its side stream sees any work.  Round 6 (tools/bisect_gather_fault.sh,
        # DESIGN 5): with a communicator alive, pipelines whose first forward ran on a side stream (capture()'s warm-up) fault at the first two-stream graph
        # replay ("illegal memory access" / "write access to a read-only page"), with or without a collective in the loop; one eager forward on the
        # current stream first and thirty batches with a gather after each run clean.  Not understood below that level (the library holds no per-stream state).
        for s in range(self.NS):
            self.run_frame(s, self.scratch[s], eager=True)
        torch.cuda.synchronize()
        for s in range(self.NS):
            with torch.cuda.stream(self.streams[s]):
                for i in range(max(a.warmup, 1)):
                    self.run_frame(i * self.NS + s, self.scratch[s], eager=True)
                torch.cuda.synchronize()
                if self.use_graph:
                    self.static_in[s][0].copy_(self.pool[0][0]); self.static_in[s][1].copy_(self.pool[0][1])
                    self.pipes[s].capture(*self.static_in[s])
                    for i in range(a.warmup):
                        self.run_frame(i * self.NS + s, self.scratch[s])
                torch.cuda.synchronize()
        # the graph replay does the frame's work: one replay against one eager (op-by-op) run of the same frame, bit for bit
        # (every kernel is deterministic), outside the timed region
        if self.use_graph:
            ok = True
            for s in range(self.NS):
                with torch.cuda.stream(self.streams[s]):
                    pts, n = self.pool[(s + 1) % len(self.pool)]
                    eb, ec = [t.clone() for t in self.pipes[s].forward(pts, n)]
                    self.static_in[s][0].copy_(pts); self.static_in[s][1].copy_(n)
                    gb, gc = self.pipes[s].replay()
                    torch.cuda.synchronize()
                    ok = ok and bool(torch.equal(eb, gb)) and bool(torch.equal(ec, gc)) and int(ec[0]) > 0
            self.replay_equals_eager = ok
            if not ok:
                raise SystemExit(f"bench.py: the HIP-graph replay of a frame differs from its eager run ({self.mode})")

    def sample(self, results, prof):
        """the roofline sample: ONE forward launched op by op, alone on the GPU, with HIP events around every launch (events cannot bracket
        kernels inside a graph replay).  A pre-pass outside every timed region, for every N (round 3 put it inside the first repeat, which
        a one-repeat N > 1 run then carried in its only region: scaling read 3-5 % low before any real effect)."""
        torch.cuda.synchronize()
        self.pkg.plugin.PROFILE = prof
        with torch.cuda.stream(self.streams[0]):
            self.run_frame(0, results[0:self.FB], eager=True)
        torch.cuda.synchronize()
        self.pkg.plugin.PROFILE = None
        return 1

    def gather_forward(self, j, results):
        """the product loop of BASELINE configs[3] (src/dsvt-ai-trt.cpp:1884-1970: a result per frame, every frame): the rows of forward j -- FB frames on every rank:
        configs[3]'s 32 frames over 8 GPUs -- meet on rank 0; static buffers (parallel.GatherBuffers), no allocation.  The HOST waits for the forward's stream first (the
        other stream's forward keeps the GPU busy meanwhile): a collective that is only stream-ordered behind graph replays on side streams is one of the two triggers of
        the fault bisected in round 6 (profiles/r06_gather_fault_bisect.txt; the other, the order of the pipelines' first forwards, is removed in prepare()) -- with one
        GPU both forms ran clean for thousands of gathers, no N > 1 run exists yet, and the host wait costs nothing measurable."""
        par, FB = self.par, self.FB
        self.streams[j % self.NS].synchronize()
        g = par.gather_results(results[j * FB:(j + 1) * FB], FB * self.world, self.rank, self.world, force_collective=self.args.rccl_single, buffers=self.batch_buffers)
        if g is not None:
            self.gathered_all[j * FB * self.world:(j + 1) * FB * self.world].copy_(g)

    def timed(self, results, K, gather):
        """exactly K steps (K / FB forwards), timed on THIS rank between two stream synchronisations; no collective inside unless `gather`
        (the last repeat: the path's one collective, the result gather).  Returns (seconds, per-forward ms, gathered rows, gather ms)."""
        par, FB, NS = self.par, self.FB, self.NS
        KB = K // FB
        marks = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(KB)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        every = self.collective and self.args.gather_every_batch
        gathered, gather_ms = None, None
        for i in range(KB):
            with torch.cuda.stream(self.streams[i % NS]):
                marks[i][0].record()
                self.run_frame(i, results[i * FB:(i + 1) * FB])
                marks[i][1].record()
            if every and i >= 1:
                self.gather_forward(i - 1, results)                # (one forward behind: the host waits for forward i - 1 while forward i runs on the other stream)
        if every:
            self.gather_forward(KB - 1, results)
        for s in self.streams:
            torch.cuda.current_stream().wait_stream(s)
        if every:
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            frame_ms = [marks[i][0].elapsed_time(marks[i][1]) for i in range(KB)]
            return dt, frame_ms, (self.gathered_all if self.rank == 0 else None), None
        if gather:
            torch.cuda.synchronize()
            tg = time.perf_counter()
            gathered = par.gather_results(results, K * self.world, self.rank, self.world, force_collective=self.args.rccl_single)
            torch.cuda.synchronize()
            gather_ms = 1e3 * (time.perf_counter() - tg)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        frame_ms = [marks[i][0].elapsed_time(marks[i][1]) for i in range(KB)]
        return dt, frame_ms, gathered, gather_ms

    def measure(self, results, K):
        """R repeats of the K-step loop, the same R on every rank (it depends on K only).  Collectives: one barrier before the first repeat, the
        result gather inside the LAST repeat, one barrier after it, then ONE all-gather of the R-vector of per-repeat times (the maximum over the ranks is taken on the host) -- two
        collective rounds however many repeats, because on this stack (ROCm 7.2, RCCL 2.26 of torch 2.10) a third barrier / gather / barrier /
        all-reduce round interleaved with HIP-graph replays ends in "Memory access fault ... write access to a read-only page" (DESIGN 5).
        Every repeat is bracketed by torch.cuda.synchronize() on both sides; value = median over the repeats of total frames / max-over-ranks time."""
        R = self.args.repeats if self.args.repeats > 0 else max(3, min(MAX_REPEATS, -(-300 // K)))
        self.par.barrier(); torch.cuda.synchronize()
        dts, fms, gathered, gather_ms = [], [], None, None
        for r in range(R):
            last = r == R - 1
            dt, frame_ms, g, gm = self.timed(results, K, gather=last)
            dts.append(dt); fms.extend(frame_ms)
            if last:
                gathered, gather_ms = g, gm
        self.par.barrier(); torch.cuda.synchronize()
        per_rank = self.par.all_ranks_vec(dts, self.dev)                 # ONE all-gather: [world][R]
        dts = [max(pr[i] for pr in per_rank) for i in range(len(dts))]
        self.per_rank_dts = per_rank
        return dts, fms, gathered, gather_ms


def roofline_rows(prof, sampled, counts, pool_len, FB, mode, n_points_per_launch, head_mx=False):
    """per plugin family: algorithmic work per launch (SURVEY 8d formulas) / measured launch duration"""
    f16, split = mode == "f16", mode in ("split", "splitmx")
    pm = {}
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", PMC_FILES[(mode, FB)])))
    except Exception:
        pass

    def pmc_traffic(substrs, chain):
        """HBM bytes per plugin enqueue from the PMC passes, over ALL kernels of the family (round 3 took the first match).  chain: every kernel of
        the family runs once per enqueue (voxelizer, set partition) -> the sum of their per-launch means; otherwise the enqueue launches ONE of
        several instantiations (convolutions, linears) -> the launch-weighted mean."""
        ks = [k for k in pm if any(x in k for x in substrs)]
        if not ks:
            return None
        if chain:
            return round(sum(pm[k]["traffic_mb_per_launch"] for k in ks) * 1e6)
        # (round 6: at four frames per forward a 468-row 128-channel convolution is TWO launches -- the whole tile rows on conv_rows_kernel<8, 2>, the partial last tile row on
        # conv_rows_kernel<4, 1>, csrc/conv_rows.hip launchRows128 --; the second launch's bytes belong to the same enqueue, its launch count does not)
        second = [k for k in ks if FB == 4 and "conv_rows_kernelILi4ELi1" in k]
        n = sum(pm[k]["launches"] for k in ks if k not in second)
        return round(sum(pm[k]["traffic_mb_per_launch"] * pm[k]["launches"] for k in ks) / max(n, 1) * 1e6)

    def work(pl, c):
        """(flops, algorithmic HBM bytes) of one launch of plugin `pl` on a forward with counts c"""
        f = pl.fields
        t = pl.plugin_type
        if t == "DsvtLinearPlugin":
            rows = c[pl.rows_kind]          # "Nk" for the two PFN linears, "P" for everything on voxel rows
            K_, N_ = f["in_features"], f["out_features"]
            ct = f.get("compute_type", 0)
            esz = 2 if f.get("input_half") else 4
            out_b = {0: 4, 1: 2, 2: 6}[f.get("output_mode", 0)]
            return (2.0 * rows * K_ * N_,
                    rows * (K_ * esz * (2 if (f.get("add_cols") and not f.get("add_gather_width")) else 1) + 12 * bool(f.get("add_gather_width"))
                            + 4 * N_ * f.get("num_layer_norms", 0) + out_b * N_)
                    + K_ * N_ * (2 if ct == 1 else 4))
        if t == "DsvtEncoderMlpPlugin":
            rows = c["P"]                   # att + x (+ xb) in, x' out (fp32 + fp16 copy in fp16 mode), weights once
            sp = f.get("split_precision", 0)
            per_row = 192 * ((4 + 4 + 4 * f.get("has_block_norm", 0) + 4) if sp else (2 + 4 + 4 * f.get("has_block_norm", 0) + 4 + 2))
            return (2.0 * rows * (192 * 192 + 2 * 192 * 384), rows * per_row + (4 if sp else 2) * (192 * 192 + 2 * 192 * 384))
        if t == "DsvtSetAttentionPlugin":
            # io_half: 1 = fp16 rows, 0 / 2 = fp32 rows (2 = the split-precision kernel; round 4 priced those at 2 bytes).  Algorithmic bytes = every
            # voxel's q, k, v row in ONCE and its output row out once (P x 3C + P x C values): two thirds of the S x 36 slots repeat voxels of the same
            # window, and the PMC fetch of the kernel IS P x 576 x 4 B (profiles/r04_split_pmc_traffic.json) -- the slot formula S x 36 x 3C counted
            # every repeat as HBM traffic (kept as slot_gather_mb in the row)
            S = c["S"][0 if pl.win == 0 else 1]
            esz = 2 if f.get("io_half") == 1 else 4
            return (4.0 * 36 * 36 * 192 * S, c["P"] * 192 * esz * 3 + c["P"] * 192 * esz)
        if t == "DsvtPosEmbedPlugin":
            L = f["num_layers"]
            return (2.0 * L * c["P"] * (2 * 192 + 192 * 192), c["P"] * 16 + L * (c["P"] * 192 * 2 + 2 * 192 * 192))
        if t == "DsvtPillarFeatureNetPlugin":
            return (2.0 * c["Nk"] * (10 * 96 + 96 * 192) + 2.0 * c["P"] * 96 * 192, c["Nk"] * 40 + c["P"] * 192 * (4 if f.get("split_precision") else 6))
        if t == "DsvtConv2dPlugin":
            Ho = (f["in_height"] + 2 * f["padding"] - f["kernel_size"]) // f["stride"] + 1
            up = f.get("pixel_shuffle", 1)
            taps = f["kernel_size"] ** 2
            sp = bool(getattr(pl, "split_in", False))                 # [hi | lo | hi] operands: the real channel count is a third
            cin = f["in_channels"] // 3 if sp else f["in_channels"]
            esz = 4 if sp else 2                                      # a split activation / weight is hi + lo = 4 bytes per element
            return (FB * 2.0 * Ho * Ho * up * up * f["out_channels"] * taps * cin,
                    FB * (esz * f["in_height"] ** 2 * cin + (4 if (f.get("out_f32") or sp) else 2) * Ho * Ho * up * up * f["out_channels"]
                          + esz * up * up * f["out_channels"] * taps * cin))
        # the scatter / gather stages north_star names, SURVEY 8(d) "algorithmic work per frame"
        if t == "Points2FeaturesPlugin":      # read 16 N; write 40 Nk (features) + 4 Nk (pidx) + 20 P (coords, count)
            return (0.0, 16.0 * n_points_per_launch + 44.0 * c["Nk"] + 20.0 * c["P"])
        if t == "DsvtSetPartitionPlugin":     # per window configuration: read 16 P; write 2 x 36 x 4 S (inds) + 2 x 36 x 4 S (mask) + 12 P (in-window coordinates)
            return (0.0, sum(16.0 * c["P"] + 2 * 2 * 36 * 4.0 * s_ + 12.0 * c["P"] for s_ in c["S"]))
        if t == "Map2BevPlugin":              # write GX GY C e (the dense map, zero fill included) + read P C e_in
            e_out, e_in = (6, 4) if f.get("split_output") else (2, 2)
            if f.get("persistent_output"):    # the map persists: write the P live cells + zero the P cells of the call before (no fill of the whole map)
                return (0.0, c["P"] * 192.0 * (e_in + 2 * e_out))
            return (0.0, FB * 468.0 * 468 * 192 * e_out + c["P"] * 192.0 * e_in)
        return (0.0, 0.0)

    resident_qkv = f16 and FB >= 3          # (csrc/linear.hip: row capacity of three or more frames -> the resident-weights kernel)
    qkv_name = ("linear_split_resident_kernel (QKV at fp32 grade: (hi, lo) fp16 operands, 3 x v_mfma_f32_16x16x32_f16 per product; a third of (w_hi, w_lo) resident in LDS per CU, waves walk 16-row tiles)" if split else
                "linear_f16_resident_kernel (QKV: half of W_qkv resident in LDS per CU, waves walk 16-row tiles, v_mfma_f32_16x16x32_f16)" if resident_qkv else
                "linear_f16_rows_kernel (QKV: all column chunks of a row tile per workgroup, v_mfma_f32_16x16x32_f16, weights by LDS-DMA)" if f16 else
                "linear_f32_kernel (v_mfma_f32_16x16x4_f32)")
    sp_ = " <SPLIT>: (hi, lo) fp16 operand pairs, fp32 tensors" if split else ""
    # matrix peak of one fp32-grade product: three fp16 MFMAs (2.5 PF / 3), or on the fp16 + fp8 K loop one fp16 MFMA + two e4m3 products at the
    # 5 PF dense MX-fp8 rate with ten tap slots for nine taps: 2.5 PF / (1 + 2 (10 / 9) / 2)
    PEAK_SPLIT3, PEAK_MX = PEAK_F16_MATRIX_TFLOPS / 3.0, PEAK_F16_MATRIX_TFLOPS / (1.0 + 10.0 / 9.0)
    meta = {"DsvtLinearPlugin": (qkv_name, "mfma" if mode == "f32" else "hbm", "linear_split_resident_kernel" if split else "linear_f16_resident_kernel" if resident_qkv else "linear_f16_rows_kernel" if f16 else "linear_f32_kernel<true>"),
            "DsvtEncoderMlpPlugin": ("encoder_mlp_stream_kernel (out-proj+LN -> FC1+GELU -> FC2+LN+LN, v_mfma_f32_16x16x32_f16, weights by LDS-DMA)" + sp_, "hbm", "encoder_mlp_stream_kernel"),
            "DsvtSetAttentionPlugin": ("set_attention_f16_kernel (v_mfma_f32_16x16x32_f16)" if f16 else
                                       "set_attention_split_kernel ((hi, lo) images of Q, K, V^T in LDS, 3 x v_mfma_f32_16x16x32_f16 per product, fp32 I/O)" if split else
                                       "set_attention_kernel (v_mfma_f32_16x16x4_f32, fp32 I/O)", "hbm",
                                       "set_attention_f16_kernel" if f16 else "set_attention_split_kernel" if split else "set_attention_kernel("),
            "DsvtPosEmbedPlugin": ("posembed_batched_kernel (8 position-embedding MLPs, v_mfma_f32_16x16x32_f16)", "hbm", "posembed_batched_kernel"),
            "DsvtPillarFeatureNetPlugin": ("pfn_kernel (both PFN layers + scatter-max, v_mfma_f32_16x16x4_f32 + 16x16x32_f16; peak = the mix of the two matrix rates; a wave owns a work-balanced group of up to 16 pillars; bound by dependent LDS / L2 round trips at two waves per SIMD, not by either)" + sp_, "mfma", "pfn_kernel"),
            "DsvtConv2dPlugin": ("conv_rows_kernel<8, 2> / <4, 3> / <4, 2> / <4, 1> (round 6: the 3 x 3 stride-1 layers of the fp32-grade frame) / conv_wide_kernel / conv_halo_kernel / conv_f16_kernel / conv1x1_resident(_split / _mx)_kernel / conv3x3_grouped_narrow(_split)_kernel (implicit GEMM, v_mfma_f32_16x16x32_f16)" +
                                 (" -- 3 x 3 stride-1 layers with > 32 output channels (93 % of the products) on the fp16 + fp8 K loop over [hi | x8]: one fp16 MFMA product + "
                                  "two e4m3 correction products (v_mfma_scale_f32_16x16x128_f8f6f4) per fp32-grade product; the other layers walk [hi | lo | hi] x "
                                  "[w_hi | w_hi | w_lo], three fp16 MFMAs per product; peak = the launch-weighted mix of the two" if (split and head_mx) else
                                  " on [hi | lo | hi] x [w_hi | w_hi | w_lo]: three MFMAs per fp32-grade product" if split else ""), "mfma", "conv"),
            "Points2FeaturesPlugin": ("p2f_partition -> p2f_bins -> p2f_pillar: the voxelizer, SURVEY 8a-1", "hbm", "dsvt::p2f_"),
            "DsvtSetPartitionPlugin": ("sp_count -> sp_scan -> sp_scatter -> sp_window (+ one memset): WindowPartition + GetSet of both window configurations, SURVEY 8a-3/4", "hbm", ("dsvt::sp_", "sp_window")),
            "Map2BevPlugin": ("map2bev_kernel + map2bev_clear_kernel: the scatter and the zeroing of the cells the previous call wrote (persistent_output; the stateless plugin fills the whole map, plugins/src/map2bev.cu:250-310)", "hbm", "map2bev")}
    rows_out = []
    for ptype, lst in prof.items():
        if not lst or ptype not in meta:
            continue
        per_frame = len(lst) // sampled
        tot_ms = tot_fl = tot_by = tot_peak_s = 0.0
        for j, (e0, e1, pl) in enumerate(lst):
            fl, by = work(pl, counts[(j // per_frame) % pool_len])
            tot_ms += e0.elapsed_time(e1); tot_fl += fl; tot_by += by
            tot_peak_s += fl / 1e12 / (PEAK_MX if getattr(pl, "mx_in", False) else PEAK_SPLIT3)       # (the split convolutions' matrix-peak time)
        n_l = len(lst)
        avg_ms = tot_ms / n_l
        tfl, gbs = tot_fl / n_l / (avg_ms * 1e-3) / 1e12, tot_by / n_l / (avg_ms * 1e-3) / 1e9
        kname, bound, pmk = meta[ptype]
        # split precision: an fp32-grade product IS three fp16 MFMAs, so the matrix peak of that arithmetic is a third of the fp16 peak
        peak_tf = (PEAK_F32_MATRIX_TFLOPS if mode == "f32" else PEAK_F16_MATRIX_TFLOPS / (3 if split else 1))
        if ptype == "DsvtPillarFeatureNetPlugin":
            # layer 0 (10 -> 96, 5 % of the flops) runs on the fp32 matrix instruction, layer 1 on fp16 (three per product in split precision): the
            # peak of that mix is flops / (time of each part at its own peak).  (Rounds 1-2 priced the whole kernel against the fp32 peak, which
            # made a latency-bound kernel look MFMA-bound.)
            f32_part = 10.0 / (10.0 + 192.0)
            peak_tf = 1.0 / (f32_part / PEAK_F32_MATRIX_TFLOPS + (1.0 - f32_part) / (PEAK_F16_MATRIX_TFLOPS / (3 if split else 1)))
        if split and ptype == "DsvtConv2dPlugin" and tot_peak_s > 0:
            peak_tf = tot_fl / 1e12 / tot_peak_s
        r = dict(kernel=kname, bound=bound)
        if bound == "hbm":
            r.update(achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4))
            if tfl:
                r["mfma_tflops"] = round(tfl, 2)
                # north_star asks for the MFMA utilisation of the attention GEMMs against the gfx950 peak: algorithmic flops / the peak of the arithmetic
                mp = PEAK_F32_MATRIX_TFLOPS if mode == "f32" else PEAK_SPLIT3 if split else PEAK_F16_MATRIX_TFLOPS
                r["mfma_frac_of_peak"] = round(tfl / mp, 4); r["mfma_peak_tflops"] = round(mp, 1)
        else:
            # peak = the guide's dense peak of the matrix instruction the kernel issues (fp16: 2.5 PF; fp32 matrix: 157 TF); achieved = ALGORITHMIC flops
            # (2 M N K of the fp32-grade product) per second.  A split-precision product is three fp16 MFMAs (or fp16 + two fp8), so the rate the matrix
            # pipe is asked for is 3 x that: frac_of_arithmetic_peak prices the achieved rate against the peak OF THAT ARITHMETIC (dense peak / 3, ...)
            dense = PEAK_F32_MATRIX_TFLOPS if mode == "f32" else PEAK_F16_MATRIX_TFLOPS
            r.update(achieved=round(tfl, 2), peak=dense, unit="TFLOP/s", frac=round(tfl / dense, 4), hbm_gbs=round(gbs, 1),
                     peak_of_the_arithmetic=round(peak_tf, 1), frac_of_arithmetic_peak=round(tfl / peak_tf, 4))
            if split and ptype == "DsvtConv2dPlugin" and not head_mx:
                r["mfma_issue_tflops"] = round(3 * tfl, 1)
        chain = ptype in ("Points2FeaturesPlugin", "DsvtSetPartitionPlugin")
        r.update(traffic=pmc_traffic([pmk] if isinstance(pmk, str) else list(pmk), chain), launches_per_frame=per_frame, sampled_frames=sampled, avg_launch_us=round(1e3 * avg_ms, 2),
                 ms_per_forward=round(tot_ms / sampled, 3), algorithmic_mb_per_launch=round(tot_by / n_l / 1e6, 2))
        if tot_fl:
            r["algorithmic_gflop_per_launch"] = round(tot_fl / n_l / 1e9, 3)
        if r["traffic"] is not None:
            r["traffic_source"] = "profiles/" + PMC_FILES[(mode, FB)]
        if ptype == "DsvtSetAttentionPlugin":        # the slot formula of rounds 1-4 (every slot's q, k, v row counted as HBM traffic), for comparison
            esz_ = 2 if lst[0][2].fields.get("io_half") == 1 else 4
            r["slot_gather_mb"] = round(sum(counts[(j // per_frame) % pool_len]["S"][0 if pl.win == 0 else 1] * 36 * 192 * esz_ * 3 for j, (_, _, pl) in enumerate(lst)) / n_l / 1e6, 2)
        rows_out.append((ptype, r))
    return rows_out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--dtype", choices=["f32", "f16", "split"], default="split",
                    help="precision of the headline value.  split (default): (hi, lo) fp16 operand pairs on the DSVT GEMMs, fp16 + fp8 correction "
                         "products in the convolutions, fp32 tensors -- fp32 grade, boxes within 1e-3 of the oracle (north_star's bar); f16: fp16 MFMA "
                         "operands / fp16 dense head, fp32 accumulate (BASELINE configs[2] says fp16, but its boxes miss the 1e-3 bar on z / size: timed as "
                         "`fast_mode` beside the headline); f32: v_mfma_f32_16x16x4_f32 linears, the slow exact cross-check")
    ap.add_argument("--streams", type=int, default=2,
                    help="forwards in flight per GPU: independent pipeline instances on separate HIP streams")
    ap.add_argument("--batch", type=int, default=4,
                    help="frames per forward(): their pillar rows are concatenated and every backbone layer is ONE launch for all of them "
                         "(DsvtPipeline(frames=B)); a step is still one frame, --steps must be a multiple of B")
    ap.add_argument("--no-graph", action="store_true", help="launch every op from the host instead of replaying a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-mode", "--no-fast-mode", dest="no_fast_mode", action="store_true", help="skip the run of the OTHER precision mode (fp16 beside a split headline) that follows the headline (N = 1 only)")
    ap.add_argument("--no-latency-mode", action="store_true", help="skip the single-frame-mode measurement (N = 1 only) that follows the timed region")
    ap.add_argument("--host-input", action="store_true", help="frames start in pinned host memory and are uploaded (n x 16 B) inside the timed region: the PCIe-inclusive rate quoted in DESIGN.md, never the headline value")
    ap.add_argument("--no-nms", action="store_true", help="stop at FilterBoxByScore (the reference engine's output) instead of the final boxes")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the per-launch HIP events (roofline = null)")
    ap.add_argument("--rccl-single", action="store_true", help="N = 1 only: create a communicator of size 1 so that the result gather "
                                                               "really goes through RCCL (SURVEY 8e: exercising the collective on one device)")
    ap.add_argument("--share-gpu", action="store_true", help="N > visible GPUs: rank r uses GPU r mod visible (a launcher / RCCL dry run on one device; "
                                                             "the line is marked and is NOT a scaling number)")
    ap.add_argument("--gather-every-batch", dest="gather_every_batch", action="store_true", default=None,
                    help="the result gather runs after EVERY batch (one forward per stream on every rank) inside every repeat, with buffers allocated once -- the product loop of "
                         "BASELINE configs[3]; default for N > 1 and with --rccl-single.  --gather-once: round 5's protocol (the gather inside the last repeat only)")
    ap.add_argument("--gather-once", dest="gather_every_batch", action="store_false")
    ap.add_argument("--repeats", type=int, default=0, help="repeats of the K-step timed loop (0 = max(3, min(15, ceil(300 / K))): a function of K only, so every rank runs the same number)")
    ap.add_argument("--dump-rows", default=None, help="rank 0 saves the gathered result rows [K * N, 4501] of the headline mode as .npy (tests: the gather against single-process rows)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip other_configs (BASELINE configs[1] and configs[4] timed beside the headline, ~10 s)")
    ap.add_argument("--no-cpp-host", action="store_true", help="skip cpp_host_mode (the C++ host dsvt_detect on the same clouds: writes an ~80 MB .wts file, ~25 s)")
    ap.add_argument("--oracle-clouds", type=int, default=8, help="how many 180k-point clouds the timed modes' boxes are checked on against the CPU oracle (~10 s of host time each on the GPU box; cpu_baseline.box_err_vs_oracle = the worst over them)")
    ap.add_argument("--no-whole-network-cpu", action="store_true", help="cpu_baseline skips the whole network on the CPU oracle (~10 s; also drops box_err_vs_oracle)")
    args = ap.parse_args()
    if args.gather_every_batch is None:
        args.gather_every_batch = args.gpus > 1 or args.rccl_single
    # the product library reads no environment switch (csrc/plugin_base.h ablateEnv), and a timed run must not load another build either
    stray = sorted(k for k in os.environ if k.startswith("DSVT_"))
    if stray:
        raise SystemExit(f"bench.py: refusing to run with {stray} set (ablation / A-B switches belong to tools/, not to a timed run)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus, args.share_gpu)

    # one rank builds (the in-tree .so normally travels with the snapshot and this is a no-op); the others wait
    rank0 = int(os.environ.get("RANK", "0")) == 0
    if rank0:
        G.build()
    par = G._load_file("dsvt_parallel_boot", os.path.join(G.PKG_DIR, "parallel.py"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: n_gpus would not be the ranks that ran")
    ngpu = torch.cuda.device_count()
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    shared = args.share_gpu and ngpu < args.gpus
    if ngpu < lr + 1 and not shared:
        raise SystemExit(f"bench.py: rank {os.environ.get('RANK')} has no GPU (only {ngpu} visible)")
    dev_index = lr % ngpu
    # RCCL refuses two ranks of a communicator on one device ("Duplicate GPU detected", probed with tools/rccl_same_device.py), so the
    # shared-device dry run gathers through gloo (rows staged through the host); every other run is nccl = RCCL
    rank, local_rank, world = par.init(single_rank_group=args.rccl_single, device_index=dev_index, backend="gloo" if shared else None)
    par.barrier()
    pkg = G.load_package()
    par = pkg.parallel
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    # n_gpus must be the number of DISTINCT physical devices that took part: every rank's device identity (UUID / PCI address) meets in one
    # all-gather BEFORE any HIP graph exists; a job whose ranks doubled up on a device (a wrong HIP_VISIBLE_DEVICES / LOCAL_RANK mapping) is refused
    # unless it is the declared dry run (--share-gpu)
    dev_ids = par.gather_device_identities(dev)
    n_distinct = len({d for d, _ in dev_ids})
    if n_distinct != world and not shared:
        raise SystemExit(f"bench.py: {world} ranks on {n_distinct} distinct device(s) {sorted({d for d, _ in dev_ids})}: n_gpus would not be the GPUs that ran "
                         "(use --share-gpu for a declared dry run)")

    FB = max(1, args.batch) if args.dtype in ("f16", "split") else 1          # (the exact-fp32 cross-check mode has no multi-frame path)
    if args.host_input:
        FB = 1
    while args.steps % FB:          # exactly K timed frames: the largest frames-per-forward <= --batch that divides K
        FB -= 1
    caps = pkg.pipeline.Caps() if FB == 1 else pkg.pipeline.Caps.for_frames(FB)     # 196608 points per frame; pillar / window / set capacities are totals
    weights = pkg.synth.make_weights()

    # synthetic frames of this rank, resident in HBM before the timed region
    K = args.steps
    # distinct clouds of this rank: whole forwards, at least one per stream when the steps allow it (so that the streams do not replay the
    # same four clouds), seeds rank * C + i
    nclouds = -(-max(FB, min(FRAME_POOL, max(K, 1))) // FB) * FB
    nclouds = max(nclouds, (min(K, FB * max(1, args.streams)) // FB) * FB)
    clouds = [pkg.synth.lidar_like(args.points, seed=rank * nclouds + i) for i in range(nclouds)]
    pool = []                                       # entries = the inputs of one forward(): FB consecutive clouds, frame f in rows f * caps.N ...
    for j in range(max(1, len(clouds) // FB)):
        buf = np.zeros((1, FB * caps.N, 4), np.float32); ns = []
        for f in range(FB):
            p = clouds[(j * FB + f) % len(clouds)]
            buf[0, f * caps.N:f * caps.N + p.shape[0]] = p; ns.append(p.shape[0])
        pool.append((torch.from_numpy(buf).to(dev), torch.tensor(ns, dtype=torch.int32, device=dev)))
    results = torch.zeros((K, par.ROW), dtype=torch.float32, device=dev)
    # --host-input: the frames wait in pinned host memory (where a loader thread would have read the .bin files) and only
    # the n x 16 bytes that exist + the count cross PCIe, asynchronously on the frame's stream
    host_pool = [(p_[0, :int(n_[0])].cpu().pin_memory(), n_.cpu().pin_memory(), int(n_[0])) for p_, n_ in pool] if args.host_input else None

    def run_mode(mode):
        """build, warm, capture and time one precision mode; returns its part of the JSON line (rank 0) or None"""
        run = ModeRun(pkg, par, args, mode, dev, caps, FB, weights, pool, host_pool, world, rank)
        run.prepare()
        # device-side counts of each pooled forward (for the algorithmic byte / flop counts), read outside the timed region
        counts = []
        for pts, n in pool:
            st = run.pipes[0].voxel_stage(pts, n)
            counts.append(dict(P=int(st["P"][0]), Nk=int(st["Nk"][0]), S=[int(g[2][0]) for g in st["gss"]]))
        torch.cuda.synchronize()
        prof = None
        if not args.no_kernel_events:
            prof = {k: [] for k in ("DsvtLinearPlugin", "DsvtEncoderMlpPlugin", "DsvtSetAttentionPlugin", "DsvtConv2dPlugin", "DsvtPillarFeatureNetPlugin",
                                    "DsvtPosEmbedPlugin", "Points2FeaturesPlugin", "DsvtSetPartitionPlugin", "Map2BevPlugin")}
        # warm-up of the one collective: RCCL sets up its channels lazily at the first call of each kind (tens of ms: measured 45 ms on a size-1
        # communicator, as much as 23 frames), which is start-up cost, not a property of the frame path
        if world > 1 or args.rccl_single:
            par.gather_results(results, K * world, rank, world, force_collective=args.rccl_single)
            if run.batch_buffers is not None:
                par.gather_results(results[:FB], FB * world, rank, world, force_collective=args.rccl_single, buffers=run.batch_buffers)
        sampled = run.sample(results, prof) if prof is not None else 0
        dts, frame_ms, gathered, gather_ms = run.measure(results, K)
        if rank != 0:
            return None
        total = K * world
        own_ok = None
        if world > 1 or args.rccl_single:
            assert gathered is not None and gathered.shape[0] == total
            # the gather, checked where it can be: rank 0's own shard sits in the gathered tensor bit for bit, in global frame order f -> rank f mod N
            own_ok = par.own_rows_match(gathered, results, total, rank, world)
            if not own_ok:
                raise SystemExit("bench.py: the gathered result rows of rank 0's own shard differ from its local rows")
        if args.dump_rows and mode == args.dtype:
            np.save(args.dump_rows, gathered.cpu().numpy())
        med = float(np.median(dts))
        out = dict(value=round(total / med, 3), ms_per_step=round(1e3 * med / K, 4), p50_ms=round(float(np.median(frame_ms)), 4), repeats=len(dts),
                   repeat_values=[round(total / d, 1) for d in dts],
                   gather_ms=None if gather_ms is None or not run.collective else round(gather_ms, 3),
                   value_of_the_repeat_with_the_gather=round(total / dts[-1], 3),
                   gathers_per_repeat=(K // FB) if (run.collective and args.gather_every_batch) else (1 if run.collective else 0),
                   graph_replay_equals_eager=run.replay_equals_eager, frame0=counts[0], gather_own_rows_bit_identical=own_ok,
                   repeat_values_per_rank=[[round(K / d, 1) for d in pr] for pr in getattr(run, "per_rank_dts", [])] if world > 1 else None)
        out["_run"] = run
        if prof is not None and sampled:
            npl = sum(int(v) for v in pool[0][1].cpu())
            rows = roofline_rows(prof, sampled, counts, len(pool), FB, mode, npl, head_mx=bool(getattr(run.pipes[0], "head_mx", False)))
            # the headline roofline object = the kernel family with the LARGEST share of the frame, whichever it is (round 4 left the convolutions
            # out of the candidates although they are two thirds of the frame); roofline_hot_path = the largest of SURVEY 8(a)'s own rows (the DSVT stage)
            tot = sum(x[1]["ms_per_forward"] for x in rows)
            for _, r_ in rows:
                r_["share_of_sampled_forward"] = round(r_["ms_per_forward"] / max(tot, 1e-9), 4)
            top = max(rows, key=lambda x: x[1]["ms_per_forward"])[1]
            hot = [x for x in rows if x[0] not in ("DsvtConv2dPlugin", "Points2FeaturesPlugin", "DsvtSetPartitionPlugin", "Map2BevPlugin")] or rows
            out["roofline"] = top
            out["roofline_hot_path"] = max(hot, key=lambda x: x[1]["ms_per_forward"])[1]
            out["roofline_other_kernels"] = [r for _, r in rows if r is not top]
        else:
            out["roofline"], out["roofline_hot_path"], out["roofline_other_kernels"] = None, None, []
        return out

    head = run_mode(args.dtype)
    if rank == 0:
        run = head.pop("_run")
        line = {
            "metric": "frames/sec (p50 per-frame ms in p50_ms), 180k-pt Waymo pillar DSVT",
            "value": head["value"], "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "p50_ms": head["p50_ms"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f16": "f16 (fp16 MFMA operands, fp32 accumulate: boxes OUTSIDE the 1e-3 bar on z / size)",
                      "split": "f16x3 (fp32 grade: every GEMM / convolution operand a (hi, lo) fp16 pair, three v_mfma_f32_16x16x32_f16 per product, fp32 "
                               "accumulate, fp32 tensors in the DSVT stage; all nine box columns within 1e-3 of the fp32 oracle: cpu_baseline.box_err_vs_oracle.split)",
                      "f32": "f32"}[args.dtype],
            "data": "synthetic" + (" (uploaded from pinned host memory inside the timed region)" if args.host_input else ""),
            "repeats": head["repeats"], "repeat_values": head["repeat_values"], "gather_ms": head["gather_ms"],
            "value_of_the_repeat_with_the_gather": head["value_of_the_repeat_with_the_gather"],
            "gathers_per_repeat": head["gathers_per_repeat"],
            "repeat_values_per_rank": head["repeat_values_per_rank"],
            "config": {"workload": f"BASELINE configs[2]: lidar_like({args.points}, seed) Waymo-shaped cloud, 0.32 m pillars, "
                                   "468x468 BEV, full 4-block DSVT pillar backbone + BEV ResNet + CenterHead + top-K decode + "
                                   "FilterBoxByScore" + ("" if args.no_nms else " + rotated NMS (final boxes)") + "; seeded random weights (dsvt.wts is not shipped)",
                       "frames_per_gpu": K, "parallelism": f"frame-batch dp{world}, one result gather",
                       "frames": f"{len(clouds)} distinct clouds per rank, seeds rank * {len(clouds)} + i, cycled (BASELINE configs[3]: "
                                 f"32 frames lidar_like(180000, 0..31) over 8 GPUs, four frames per forward)",
                       "result_gather": ("gloo gather through the host (ranks share a device: RCCL refuses duplicate GPUs)" if shared else
                                         "rccl gather" if world > 1 else "rccl gather (communicator of size 1)" if args.rccl_single
                                         else "none (single process)"),
                       "graph_replay_equals_eager": head["graph_replay_equals_eager"],
                       "gather_own_rows_bit_identical": head["gather_own_rows_bit_identical"],
                       "devices": {"ranks": world, "distinct": n_distinct, "identities_md5": [d for d, _ in dev_ids]},
                       "launch": "hip-graph replay per forward" if not args.no_graph else "host launch per op",
                       "frames_in_flight": run.NS * FB, "frames_per_forward": FB,
                       "timing": (f"K = {K} steps per repeat, each repeat between two torch.cuda.synchronize() on every rank; one barrier before the first repeat, " +
                                  ("the result gather after EVERY forward (frames_per_forward frames per rank) inside every repeat (static buffers, rank 0 checks its own rows of the last repeat bit for bit), "
                                   if (run.collective and args.gather_every_batch) else "the result gather inside the LAST repeat (gather_ms), ") +
                                  "one barrier after the last repeat, ONE all-gather of the per-repeat times (max over ranks on the host); "
                                  "value = total frames / median over repeats of the max-over-ranks time; the roofline sample is a pre-pass outside every timed region"),
                       "caps": dict(points=caps.N, pillars=caps.P, windows=caps.W, sets=caps.S, overflow_free=caps.overflow_free()),
                       "frame0": head["frame0"]},
            "roofline": head["roofline"],
            "roofline_hot_path": head["roofline_hot_path"],
            "roofline_other_kernels": head["roofline_other_kernels"],
        }
        if shared:
            line["config"]["shared_gpu"] = (f"{world} ranks on {ngpu} visible GPU(s) (--share-gpu): a launcher / sharding / gather dry run, NOT a scaling number -- "
                                            "n_gpus counts ranks, the ranks time-share the device")
            line["metric"] += " [DRY RUN: ranks share a device]"
    mode_rows = {}
    # the clouds every timed mode is CHECKED on against the CPU oracle (cpu_baseline.box_err_vs_oracle): eight seeds, the 24-cloud sweep's worst yaw
    # cases among them (21, 9, 3, 1: profiles/r04_mx_box_sweep.txt), run through the timed pipelines themselves (FB frames per forward)
    CHECK_SEEDS = [21, 9, 3, 1, 0, 7, 16, 23][:max(1, args.oracle_clouds)] if args.oracle_clouds > 0 else []
    check_clouds = [(sd, pkg.synth.lidar_like(args.points, seed=sd)) for sd in CHECK_SEEDS] if (rank == 0 and world == 1 and not args.no_cpu_baseline) else []

    def fb_rows(run_, tag):
        """FilterBoxByScore rows (before NMS: the reference engine's output) of the check clouds in this mode, through the timed pipeline"""
        if not check_clouds:
            return
        pipe = run_.pipes[0]
        nms_op, pipe.nms = pipe.nms, None
        rows_ = []
        for j in range(0, len(check_clouds), FB):
            buf = np.zeros((1, FB * caps.N, 4), np.float32); ns = [0] * FB
            grp = check_clouds[j:j + FB]
            for f, (_, cp) in enumerate(grp):
                buf[0, f * caps.N:f * caps.N + cp.shape[0]] = cp; ns[f] = cp.shape[0]
            fb = pipe.forward(torch.from_numpy(buf).to(dev), torch.tensor(ns, dtype=torch.int32, device=dev))
            torch.cuda.synchronize()
            for f in range(len(grp)):
                rows_.append((fb[0][f].cpu().numpy().copy(), int(fb[1][f])))
        mode_rows[tag] = rows_
        pipe.nms = nms_op

    if rank == 0 and world == 1:
        if not args.no_cpu_baseline:
            fb_rows(run, args.dtype)
        other = {"split": "f16", "f16": "split"}.get(args.dtype)
        if other and not args.no_fast_mode and not args.host_input:
            # the OTHER precision mode on the same frames with the same protocol.  Beside the default (split) headline: `fast_mode`, the fp16 frame
            # BASELINE configs[2] names -- 2.2 x the frames/s, but its boxes sit 2e-3 .. 4e-3 from the fp32 oracle on z / size, outside north_star's
            # 1e-3 (cpu_baseline.box_err_vs_oracle.f16), so it is reported, not claimed.  Beside an f16 headline: `parity_mode`, as in rounds 1-3.
            pm = run_mode(other)
            prun = pm.pop("_run")
            if not args.no_cpu_baseline:
                fb_rows(prun, other)
            key = "fast_mode" if other == "f16" else "parity_mode"
            line[key] = dict(
                dtype=("f16: fp16 MFMA operands / fp16 BEV maps, fp32 accumulate + LayerNorm / softmax / box decode (BASELINE configs[2] 'fp16'; the reference's "
                       "own arithmetic is fp32: include/params.h:332)" if other == "f16" else
                       "f16x3: the fp32-grade mode (see --dtype split)"),
                value=pm["value"], unit="frames/s", ms_per_step=pm["ms_per_step"], p50_ms=pm["p50_ms"], repeats=pm["repeats"], repeat_values=pm["repeat_values"],
                frames_per_forward=FB, frames_in_flight=prun.NS * FB,
                graph_replay_equals_eager=pm["graph_replay_equals_eager"], roofline=pm["roofline"], roofline_other_kernels=pm["roofline_other_kernels"],
                note=("OUT OF TOLERANCE: boxes 2e-3 (z) .. 3.5e-3 (size) from the fp32 oracle (cpu_baseline.box_err_vs_oracle.f16; every one of ~60 fp16 rounding "
                      "sites adds ~4e-4, DESIGN 2) -- north_star's bar is 1e-3, which only the headline mode meets" if other == "f16" else
                      "boxes within 1e-3 of the fp32 oracle (cpu_baseline.box_err_vs_oracle.split)"))
            del prun, pm
        if args.dtype == "split" and not args.no_fast_mode and not args.host_input:
            # round 4's headline, now reported beside it: the fp32-grade frame with the convolutions' two correction products on the fp8 scaled MFMA
            # (`DsvtPipeline(head_mx=True)`): ~20 % more frames/s, centres / sizes / scores ~1e-4 from the oracle, but the yaw of boxes with a short rot
            # vector lands above 1e-3 on about one cloud in thirty (tools/head_variant_sweep.py) -- a nine-column 1e-3 bar rejects it, so it is not `value`
            em = run_mode("splitmx")
            erun = em.pop("_run")
            if not args.no_cpu_baseline:
                fb_rows(erun, "splitmx")
            line["fp8_head_mode"] = dict(dtype="f16x3 in the DSVT stage; fp16 product + two e4m3 correction products (v_mfma_scale_f32_16x16x128_f8f6f4) in the 3 x 3 and 1 x 1 "
                                               "stride-1 convolutions (DsvtPipeline(head_mx=True))",
                                         value=em["value"], unit="frames/s", ms_per_step=em["ms_per_step"], p50_ms=em["p50_ms"], repeats=em["repeats"],
                                         repeat_values=em["repeat_values"], graph_replay_equals_eager=em["graph_replay_equals_eager"],
                                         note="NOT CLAIMED: cpu_baseline.box_err_vs_oracle.splitmx -- centres / sizes / scores ~1e-4, yaw above 1e-3 on the worst "
                                              "box of the check clouds")
            del erun, em
        if FB > 1 and not args.no_latency_mode and args.dtype in ("f16", "split"):
            # the reference's own mode beside the headline: ONE frame per forward, one in flight (graph replay), same clouds --
            # what a caller who wants latency, not throughput, gets from the same kernels; measured after the timed region
            c1 = pkg.pipeline.Caps()
            one = []
            for cl in clouds[:FRAME_POOL]:
                b1 = np.zeros((1, c1.N, 4), np.float32); b1[0, :cl.shape[0]] = cl
                one.append((torch.from_numpy(b1).to(dev), torch.tensor([cl.shape[0]], dtype=torch.int32, device=dev)))
            line["single_frame_mode"] = {"frames_per_forward": 1, "frames_in_flight": 1,
                                         "note": "same kernels, one frame at a time (the reference's mode, src/dsvt-ai-trt.cpp:1884-1956; BASELINE.md: >= 200 frames/s, <= 5 ms p50); "
                                                 "top-level value / p50_ms of this object = the split (fp32-grade) mode"}
            KL = 64
            for tag, kw1 in (("split", dict(linear_compute=pkg.plugin.COMPUTE_SPLIT)), ("f16", dict(linear_compute=pkg.plugin.COMPUTE_F16, head_dtype=torch.float16))):
                p1 = pkg.pipeline.DsvtPipeline(weights, caps=c1, device=dev, device_nms=not args.no_nms, **kw1)
                sin = (torch.zeros_like(one[0][0]), torch.zeros_like(one[0][1]))
                for pts1, n1 in one[:2]:
                    p1.forward(pts1, n1)
                torch.cuda.synchronize()
                sin[0].copy_(one[0][0]); sin[1].copy_(one[0][1])
                p1.capture(*sin)
                ev1 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(KL)]
                for i in range(-8, KL):
                    pts1, n1 = one[i % len(one)]
                    if i >= 0:
                        ev1[i][0].record()
                    sin[0].copy_(pts1); sin[1].copy_(n1)
                    p1.replay()
                    if i >= 0:
                        ev1[i][1].record()
                    if i == -1:
                        torch.cuda.synchronize(); tl0 = time.perf_counter()
                torch.cuda.synchronize()
                dl = time.perf_counter() - tl0
                line["single_frame_mode"][tag] = {"frames": KL, "value": round(KL / dl, 1), "unit": "frames/s",
                                                  "p50_ms": round(float(np.median([a.elapsed_time(b) for a, b in ev1])), 4)}
                if tag == "split":
                    # the reference's timed bracket (src/dsvt-ai-trt.cpp:1918-1956): H2D of the frame + enqueue + D2H of the result + NMS, one frame at a
                    # time, synchronously.  The points wait in pinned host memory (n x 16 bytes cross PCIe, not the zero-padded cap the reference
                    # copies), the final boxes + count come back to pinned host memory; the host clock brackets each frame.  Never the headline value.
                    hp = [(p_[0, :int(n_[0])].cpu().pin_memory(), n_.cpu().pin_memory(), int(n_[0])) for p_, n_ in one]
                    gb, gc = p1.graph_out
                    hb, hc = torch.empty(gb.shape, dtype=gb.dtype).pin_memory(), torch.empty(gc.shape, dtype=gc.dtype).pin_memory()
                    lat = []
                    for i in range(-8, KL):
                        h_pts, h_n, k_ = hp[i % len(hp)]
                        torch.cuda.synchronize(); th = time.perf_counter()
                        sin[0][0, :k_].copy_(h_pts, non_blocking=True); sin[1].copy_(h_n, non_blocking=True)
                        p1.replay()
                        hb.copy_(gb, non_blocking=True); hc.copy_(gc, non_blocking=True)
                        torch.cuda.synchronize()
                        if i >= 0:
                            lat.append(1e3 * (time.perf_counter() - th))
                    line["host_input_mode"] = {"dtype": "split (fp32 grade)", "frames": KL, "value": round(1e3 * len(lat) / sum(lat), 1), "unit": "frames/s",
                                               "p50_ms": round(float(np.median(lat)), 4), "pcie_bytes_per_frame": int(np.mean([h[2] for h in hp])) * 16 + 4 + gb.numel() * 4 + 4,
                                               "note": "the reference's own bracket: upload of the frame's points from pinned host memory + graph replay (network + decode + "
                                                       "FilterBoxByScore + device NMS) + download of the final boxes, one frame at a time, host clock around each frame "
                                                       "(src/dsvt-ai-trt.cpp:1918-1956); the PCIe-inclusive figure, never `value`"}
                del p1
            line["single_frame_mode"].update(value=line["single_frame_mode"]["split"]["value"], unit="frames/s", p50_ms=line["single_frame_mode"]["split"]["p50_ms"])
        # which operating point meets which bar (>= 200 frames/s and <= 5 ms p50 per frame; north_star: boxes within 1e-3), in ONE place:
        # the headline (FB frames per forward x NS streams) is the throughput point -- its p50 is per FORWARD of FB frames with NS in flight, not a frame
        # latency --, single_frame_mode is the latency point; `both_at_one_operating_point` says whether one of them meets both bars by itself
        sf = line.get("single_frame_mode", {}).get(args.dtype if args.dtype in ("split", "f16") else "split")
        hl_ok = line["value"] >= 200.0
        line["targets"] = {
            "bars": ">= 200 frames/s on one MI355X, <= 5 ms p50 per frame, boxes within 1e-3 of the fp32 oracle (all nine columns: cpu_baseline.box_err_vs_oracle)",
            "throughput_point": {"mode": f"{args.dtype}, {FB} frames per forward x {run.NS} streams", "frames_per_s": line["value"], "meets_200_frames_per_s": bool(hl_ok),
                                 "p50_ms_per_forward": line["p50_ms"], "frame_latency_ms": round(line["p50_ms"], 3),
                                 "meets_5_ms_p50": bool(line["p50_ms"] <= 5.0)},
            "latency_point": None if not sf else {"mode": f"{args.dtype}, one frame per forward, one in flight", "frames_per_s": sf["value"], "p50_ms": sf["p50_ms"],
                                                  "meets_200_frames_per_s": bool(sf["value"] >= 200.0), "meets_5_ms_p50": bool(sf["p50_ms"] <= 5.0)},
        }
        line["targets"]["both_at_one_operating_point"] = (
            "latency_point" if sf and sf["value"] >= 200.0 and sf["p50_ms"] <= 5.0 else
            "throughput_point" if hl_ok and line["p50_ms"] <= 5.0 else "neither: throughput and latency bars are met at different operating points" if (hl_ok and sf and sf["p50_ms"] <= 5.0) else "neither")
        if not args.no_other_configs and not args.host_input:
            try:
                line["other_configs"] = other_configs(pkg, weights, dev)
            except Exception as e:                                   # (never lose the headline line to a side measurement)
                line["other_configs"] = {"error": repr(e)[:300]}
        if not args.no_cpp_host and not args.host_input and args.dtype == "split":
            line["cpp_host_mode"] = cpp_host_mode(pkg, weights, clouds[:FRAME_POOL])
        if not args.no_cpu_baseline:
            # the FilterBoxByScore rows of the pooled frames as the GPU produced them (the reference's D2H payload)
            frames = []
            pipe = run.pipes[0]
            nms_op, pipe.nms = pipe.nms, None
            for pts, n in pool:
                fb = pipe.forward(pts, n)
                torch.cuda.synchronize()
                for f in range(FB):
                    k = int(n[f])
                    frames.append((pts[0, f * caps.N:f * caps.N + k].cpu().numpy(), fb[0][f].cpu().numpy().copy(), int(fb[1][f])))
            pipe.nms = nms_op
            c_one = pkg.pipeline.Caps()              # (the oracle runs ONE frame: per-frame capacities)
            line["cpu_baseline"] = cpu_baseline(c_one, frames, whole_network=None if args.no_whole_network_cpu else (weights,), mode_rows=mode_rows, check_clouds=check_clouds,
                                                n_points_key=args.points, live_seed=0)
        else:
            line["cpu_baseline"] = None
    elif rank == 0:
        line["cpu_baseline"] = None
    if rank == 0:
        # the JSON line is the LAST thing on stdout: whatever native libraries left in C stdio buffers (RCCL's version banner) goes out first
        try:
            import ctypes
            sys.stdout.flush(); ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
