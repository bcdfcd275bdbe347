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

Two precision modes are timed by a default run (N = 1):
  value / ms_per_step / p50_ms   `--dtype f16`: fp16 MFMA operands, fp32 accumulate (BASELINE configs[2] "fp16").  Boxes sit 2e-3 .. 4e-3 from
                the fp32 oracle: OUTSIDE north_star's 1e-3.
  parity_mode   the same frames through the split-precision pipeline (every GEMM / convolution operand a (hi, lo) fp16 pair, three MFMAs per
                product, fp32 tensors): the reference's fp32 arithmetic, boxes within 1e-3 of the oracle -- the mode that answers
                north_star's joint target (>= 200 frames/s AND 1e-3).  Same timing protocol, its own roofline rows.
  box_err_vs_oracle (inside cpu_baseline, where the oracle runs as the checker) the maximum box error of each mode on the bench frame.

Timing: K steps are timed between barrier + synchronize on both sides.  When K steps take less than half a second the K-step loop is
repeated (`repeats`); `value` is the median repeat.  The first repeat carries the roofline sample (its first forward() runs eagerly, alone,
with HIP events around every launch): `value_with_sample` is that repeat, `value_without_sample` the median of the others.

Extra objects in the line:
  single_frame_mode  (N = 1) the same kernels with ONE frame per forward and one in flight -- the reference's own mode: frames/s and p50.
  roofline      the hot path's kernel with the largest share of the frame: algorithmic bytes per launch / average launch
                duration, measured with HIP events around every launch of the sampled forward, against the 8 TB/s HBM peak of gfx950.
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
PMC_FILES = {("f16", 1): "r02_g_batch1_pmc_traffic.json", ("f16", 2): "r02_g_pmc_traffic.json", ("f16", 4): "r03_f16_pmc_traffic.json",
             ("split", 4): "r03_split_pmc_traffic.json"}
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
        d[6] = min(d[6], abs(np.pi - d[6]))      # yaw = atan(sin/cos) lives in (-pi/2, pi/2): +-pi/2 are the same heading
        errs.append(d)
    if not errs:
        return None
    m = np.array(errs).max(0)
    return dict(xy=float(m[:2].max()), z=float(m[2]), size=float(m[3:6].max()), yaw=float(m[6]), score=float(m[8]),
                max_xyz_size_score=float(max(m[:6].max(), m[8])), matched=round(matched / max(n_exp, 1), 4), boxes=int(n_got), oracle_boxes=int(n_exp))


def cpu_baseline(caps, frames, n_parallel_frames=32, whole_network=None, mode_rows=None):
    """SURVEY 8(d).  frames: [(points [n,4] float32 numpy, FilterBoxByScore rows [500,9] float32 numpy from the GPU, count)].
    value = frames/s of the reference's host path, ONE thread: loadData (read the .bin, size check, zero-pad to the cap) +
    save_result + nms_cpu on the GPU's own rows.  whole_network = (weights,): the whole network of frame 0 on the CPU oracle;
    mode_rows = {mode: (rows, count)} FilterBoxByScore rows of frame 0 per precision mode -> box_err_vs_oracle."""
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
        if mode_rows:
            # the oracle as the CHECKER of the timed modes: FilterBoxByScore rows of frame 0 (before NMS, like the reference engine's output)
            out["box_err_vs_oracle"] = {m: box_errors(r, c, boxes, int(cnt)) for m, (r, c) in mode_rows.items()}
            out["box_err_vs_oracle"]["note"] = ("max abs error of the FilterBoxByScore rows of pool frame 0 against the fp32 CPU oracle, rows matched by class + "
                                                "nearest centre; north_star's bar: centres / sizes / scores within 1e-3")
    for p_ in paths:
        os.remove(p_)
    os.rmdir(tmp)
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
              "f32": dict(linear_compute=P.COMPUTE_F32)}[mode]
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
        self.pipes = [pkg.pipeline.DsvtPipeline(weights, caps=caps, device=dev, device_nms=not args.no_nms, frames=FB, **kw) for _ in range(NS)]
        self.static_in = [(torch.zeros_like(pool[0][0]), torch.zeros_like(pool[0][1])) for _ in range(NS)]
        self.scratch = [torch.zeros((FB, par.ROW), dtype=torch.float32, device=dev) for _ in range(NS)]
        self.replay_equals_eager = None
        self.collective = world > 1 or args.rccl_single

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

    def timed(self, results, K, prof, sample):
        """exactly K steps (K / FB forwards) between barrier + synchronize; returns (seconds (max over ranks), per-forward ms, sampled forwards)"""
        par, FB, NS = self.par, self.FB, self.NS
        KB = K // FB
        marks = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(KB)]
        par.barrier(); torch.cuda.synchronize()
        sampled = 0
        t0 = time.perf_counter()
        for i in range(KB):
            # roofline sample: the FIRST forward of a sampled repeat is launched op by op, alone on the GPU, with HIP events around each
            # launch (events cannot bracket kernels inside a graph replay); nothing is in flight yet, so it drains no other stream
            ev = sample and prof is not None and (not self.use_graph or i == 0)
            if ev and (self.use_graph or NS > 1):
                torch.cuda.synchronize()
            self.pkg.plugin.PROFILE = prof if ev else None
            with torch.cuda.stream(self.streams[i % NS]):
                marks[i][0].record()
                self.run_frame(i, results[i * FB:(i + 1) * FB], eager=ev)
                marks[i][1].record()
            if ev and (self.use_graph or NS > 1):
                torch.cuda.synchronize()
            sampled += ev
        for s in self.streams:
            torch.cuda.current_stream().wait_stream(s)
        gathered = par.gather_results(results, K * self.world, self.rank, self.world, force_collective=self.args.rccl_single)   # the one collective of the path
        par.barrier(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        self.pkg.plugin.PROFILE = None
        dt = par.max_over_ranks(dt, self.dev)
        frame_ms = [marks[i][0].elapsed_time(marks[i][1]) for i in range(KB)]
        return dt, frame_ms, sampled, gathered

    def measure(self, results, K, prof):
        """repeat the K-step loop until MIN_TIMED_S is covered; the first repeat carries the roofline sample"""
        dt0, fm0, sampled, gathered = self.timed(results, K, prof, sample=True)
        dts, fms = [dt0], []
        R = 1 if dt0 >= MIN_TIMED_S else min(MAX_REPEATS, max(3, int(np.ceil(MIN_TIMED_S / dt0))))
        if self.collective:
            # ONE timed region when a communicator exists (N > 1, or --rccl-single): measured on this stack (ROCm 7.2, RCCL of torch 2.10), a
            # third barrier / gather / barrier / all-reduce round interleaved with HIP-graph replays hangs (one stream) or faults (two) --
            # `--rccl-single --repeats 3`; one or two rounds, round 2's protocol, are fine.  The N = 1 line without a communicator -- the
            # one a short --steps makes noisy -- repeats.
            R = 1
        if self.args.repeats > 0:
            R = self.args.repeats
        for _ in range(1, R):
            dt, frame_ms, _sm, g = self.timed(results, K, prof, sample=False)
            dts.append(dt); fms.extend(frame_ms)
            gathered = g if g is not None else gathered
        if not fms:
            fms = fm0
        return dts, fms, sampled, gathered


def roofline_rows(prof, sampled, counts, pool_len, FB, mode, n_points_per_launch):
    """per plugin family: algorithmic work per launch (SURVEY 8d formulas) / measured launch duration"""
    f16, split = mode == "f16", mode == "split"
    pm = {}
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", PMC_FILES[(mode, FB)])))
    except Exception:
        pass

    def pmc_traffic(substr):
        ks = [k for k in pm if substr in k]
        return round(pm[ks[0]]["traffic_mb_per_launch"] * 1e6) if ks else None

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
            S = c["S"][0 if pl.win == 0 else 1]
            esz = 2 if f.get("io_half") else 4
            return (4.0 * 36 * 36 * 192 * S, S * 36 * 192 * esz * 3 + c["P"] * 192 * esz)
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
            e_out = 6 if f.get("split_output") else 2
            return (0.0, FB * 468.0 * 468 * 192 * e_out + c["P"] * 192.0 * (4 if f.get("split_output") else 2))
        return (0.0, 0.0)

    resident_qkv = f16 and FB >= 3          # (csrc/linear.hip: row capacity of three or more frames -> the resident-weights kernel)
    qkv_name = ("linear_split_rows_kernel (QKV at fp32 grade: (hi, lo) fp16 operands, 3 x v_mfma_f32_16x16x32_f16 per product, weights by LDS-DMA)" if split else
                "linear_f16_resident_kernel (QKV: half of W_qkv resident in LDS per CU, waves walk 16-row tiles, v_mfma_f32_16x16x32_f16)" if resident_qkv else
                "linear_f16_rows_kernel (QKV: all column chunks of a row tile per workgroup, v_mfma_f32_16x16x32_f16, weights by LDS-DMA)" if f16 else
                "linear_f32_kernel (v_mfma_f32_16x16x4_f32)")
    sp_ = " <SPLIT>: (hi, lo) fp16 operand pairs, fp32 tensors" if split else ""
    meta = {"DsvtLinearPlugin": (qkv_name, "mfma" if mode == "f32" else "hbm", "linear_split_rows_kernel" if split else "linear_f16_resident_kernel" if resident_qkv else "linear_f16_rows_kernel" if f16 else "linear_f32_kernel<true>"),
            "DsvtEncoderMlpPlugin": ("encoder_mlp_stream_kernel (out-proj+LN -> FC1+GELU -> FC2+LN+LN, v_mfma_f32_16x16x32_f16, weights by LDS-DMA)" + sp_, "hbm", "encoder_mlp_stream_kernel"),
            "DsvtSetAttentionPlugin": ("set_attention_f16_kernel (v_mfma_f32_16x16x32_f16)" if f16 else "set_attention_kernel (v_mfma_f32_16x16x4_f32, fp32 I/O)", "hbm",
                                       "set_attention_f16_kernel" if f16 else "set_attention_kernel"),
            "DsvtPosEmbedPlugin": ("posembed_batched_kernel (8 position-embedding MLPs, v_mfma_f32_16x16x32_f16)", "hbm", "posembed_batched_kernel"),
            "DsvtPillarFeatureNetPlugin": ("pfn_kernel (both PFN layers + scatter-max, v_mfma_f32_16x16x4_f32 + 16x16x32_f16; peak = the mix of the two matrix rates; the kernel is bound by the round trips of its 16-pillar groups, not by either)" + sp_, "mfma", "pfn_kernel"),
            "DsvtConv2dPlugin": ("conv_wide_kernel / conv_halo_kernel / conv_f16_kernel (implicit GEMM, v_mfma_f32_16x16x32_f16)" + (" on [hi | lo | hi] x [w_hi | w_hi | w_lo]: three MFMAs per fp32-grade product" if split else ""), "mfma", "conv_wide_kernelILi8ELi8"),
            "Points2FeaturesPlugin": ("p2f_partition -> p2f_bins -> p2f_pillar: the voxelizer, SURVEY 8a-1", "hbm", "p2f_"),
            "DsvtSetPartitionPlugin": ("sp_count -> sp_scan -> sp_scatter -> sp_window (+ one memset): WindowPartition + GetSet of both window configurations, SURVEY 8a-3/4", "hbm", "sp_"),
            "Map2BevPlugin": ("map2bev_kernel + the zero fill of the dense map (plugins/src/map2bev.cu:250-310)", "hbm", "map2bev")}
    rows_out = []
    for ptype, lst in prof.items():
        if not lst or ptype not in meta:
            continue
        per_frame = len(lst) // sampled
        tot_ms = tot_fl = tot_by = 0.0
        for j, (e0, e1, pl) in enumerate(lst):
            fl, by = work(pl, counts[(j // per_frame) % pool_len])
            tot_ms += e0.elapsed_time(e1); tot_fl += fl; tot_by += by
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
        r = dict(kernel=kname, bound=bound)
        if bound == "hbm":
            r.update(achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4))
            if tfl:
                r["mfma_tflops"] = round(tfl, 2)
        else:
            r.update(achieved=round(tfl, 2), peak=round(peak_tf, 1), unit="TFLOP/s", frac=round(tfl / peak_tf, 4), hbm_gbs=round(gbs, 1))
            if split and ptype == "DsvtConv2dPlugin":
                r["mfma_issue_tflops"] = round(3 * tfl, 1)
        r.update(traffic=pmc_traffic(pmk), launches_per_frame=per_frame, sampled_frames=sampled, avg_launch_us=round(1e3 * avg_ms, 2),
                 ms_per_forward=round(tot_ms / sampled, 3), algorithmic_mb_per_launch=round(tot_by / n_l / 1e6, 2))
        if tot_fl:
            r["algorithmic_gflop_per_launch"] = round(tot_fl / n_l / 1e9, 3)
        if r["traffic"] is not None:
            r["traffic_source"] = "profiles/" + PMC_FILES[(mode, FB)]
        rows_out.append((ptype, r))
    return rows_out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--dtype", choices=["f32", "f16", "split"], default="f16",
                    help="precision of the headline value.  f16: fp16 MFMA operands / fp16 dense head, fp32 accumulate + LayerNorm/softmax/decode "
                         "(BASELINE configs[2]); split: (hi, lo) fp16 operand pairs, fp32 tensors (fp32 grade: boxes within 1e-3; also timed as "
                         "`parity_mode` beside an f16 headline); f32: v_mfma_f32_16x16x4_f32 linears, the slow exact cross-check")
    ap.add_argument("--streams", type=int, default=2,
                    help="forwards in flight per GPU: independent pipeline instances on separate HIP streams")
    ap.add_argument("--batch", type=int, default=4,
                    help="frames per forward(): their pillar rows are concatenated and every backbone layer is ONE launch for all of them "
                         "(DsvtPipeline(frames=B)); a step is still one frame, --steps must be a multiple of B")
    ap.add_argument("--no-graph", action="store_true", help="launch every op from the host instead of replaying a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the split-precision run that follows an f16 headline (N = 1 only)")
    ap.add_argument("--no-latency-mode", action="store_true", help="skip the single-frame-mode measurement (N = 1 only) that follows the timed region")
    ap.add_argument("--host-input", action="store_true", help="frames start in pinned host memory and are uploaded (n x 16 B) inside the timed region: the PCIe-inclusive rate quoted in DESIGN.md, never the headline value")
    ap.add_argument("--no-nms", action="store_true", help="stop at FilterBoxByScore (the reference engine's output) instead of the final boxes")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the per-launch HIP events (roofline = null)")
    ap.add_argument("--rccl-single", action="store_true", help="N = 1 only: create a communicator of size 1 so that the result gather "
                                                               "really goes through RCCL (SURVEY 8e: exercising the collective on one device)")
    ap.add_argument("--share-gpu", action="store_true", help="N > visible GPUs: rank r uses GPU r mod visible (a launcher / RCCL dry run on one device; "
                                                             "the line is marked and is NOT a scaling number)")
    ap.add_argument("--repeats", type=int, default=0, help="repeats of the K-step timed loop (0 = as many as cover half a second, at least 3 when one is shorter than that)")
    ap.add_argument("--dump-rows", default=None, help="rank 0 saves the gathered result rows [K * N, 4501] of the headline mode as .npy (tests: the gather against single-process rows)")
    ap.add_argument("--no-whole-network-cpu", action="store_true", help="cpu_baseline skips the whole network on the CPU oracle (~10 s; also drops box_err_vs_oracle)")
    args = ap.parse_args()
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
        dts, frame_ms, sampled, gathered = run.measure(results, K, prof)
        if rank != 0:
            return None
        total = K * world
        if world > 1:
            assert gathered is not None and gathered.shape[0] == total
        if args.dump_rows and mode == args.dtype:
            np.save(args.dump_rows, gathered.cpu().numpy())
        med = float(np.median(dts))
        rest = dts[1:] if (sampled and len(dts) > 1) else dts
        out = dict(value=round(total / med, 3), ms_per_step=round(1e3 * med / K, 4), p50_ms=round(float(np.median(frame_ms)), 4), repeats=len(dts),
                   repeat_values=[round(total / d, 1) for d in dts],
                   value_with_sample=round(total / dts[0], 3) if sampled else None,
                   value_without_sample=round(total / float(np.median(rest)), 3),
                   graph_replay_equals_eager=run.replay_equals_eager, frame0=counts[0])
        out["_run"] = run
        if prof is not None and sampled:
            npl = sum(int(v) for v in pool[0][1].cpu())
            rows = roofline_rows(prof, sampled, counts, len(pool), FB, mode, npl)
            # the headline roofline object = the hot path's kernel with the largest share of the frame
            hot = [x for x in rows if x[0] not in ("DsvtConv2dPlugin", "Points2FeaturesPlugin", "DsvtSetPartitionPlugin", "Map2BevPlugin")] or rows
            top = max(hot, key=lambda x: x[1]["ms_per_forward"])[1]
            out["roofline"] = top
            out["roofline_other_kernels"] = [r for _, r in rows if r is not top]
        else:
            out["roofline"], out["roofline_other_kernels"] = None, []
        return out

    head = run_mode(args.dtype)
    if rank == 0:
        run = head.pop("_run")
        line = {
            "metric": "frames/sec (p50 per-frame ms in p50_ms), 180k-pt Waymo pillar DSVT",
            "value": head["value"], "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "p50_ms": head["p50_ms"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": {"f16": "f16", "split": "f16x3 (split-precision fp16 MFMA, fp32 grade)", "f32": "f32"}[args.dtype],
            "data": "synthetic" + (" (uploaded from pinned host memory inside the timed region)" if args.host_input else ""),
            "repeats": head["repeats"], "repeat_values": head["repeat_values"], "value_with_sample": head["value_with_sample"],
            "value_without_sample": head["value_without_sample"],
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
                       "launch": "hip-graph replay per forward" if not args.no_graph else "host launch per op",
                       "frames_in_flight": run.NS * FB, "frames_per_forward": FB,
                       "timing": f"K = {K} steps per repeat between barrier + synchronize; value = median of `repeats` repeats; repeat 0 carries the roofline sample"
                                 + ("; ONE repeat when a communicator exists (RCCL rounds interleaved with graph replays beyond two hang on this stack)" if run.collective else ""),
                       "caps": dict(points=caps.N, pillars=caps.P, windows=caps.W, sets=caps.S, overflow_free=caps.overflow_free()),
                       "frame0": head["frame0"]},
            "roofline": head["roofline"],
            "roofline_other_kernels": head["roofline_other_kernels"],
        }
        if shared:
            line["config"]["shared_gpu"] = (f"{world} ranks on {ngpu} visible GPU(s) (--share-gpu): a launcher / sharding / gather dry run, NOT a scaling number -- "
                                            "n_gpus counts ranks, the ranks time-share the device")
            line["metric"] += " [DRY RUN: ranks share a device]"
    mode_rows = {}

    def fb_rows(run_, tag):
        """FilterBoxByScore rows (before NMS: the reference engine's output) of pool frame 0 in this mode"""
        pipe = run_.pipes[0]
        nms_op, pipe.nms = pipe.nms, None
        fb = pipe.forward(*pool[0])
        torch.cuda.synchronize()
        mode_rows[tag] = (fb[0][0].cpu().numpy().copy(), int(fb[1][0]))
        pipe.nms = nms_op
        return fb

    if rank == 0 and world == 1:
        if not args.no_cpu_baseline:
            fb_rows(run, args.dtype)
        if args.dtype == "f16" and not args.no_parity_mode and not args.host_input:
            # north_star's joint target: >= 200 frames/s AND boxes within 1e-3.  Same frames, same protocol, split-precision kernels.
            pm = run_mode("split")
            prun = pm.pop("_run")
            if not args.no_cpu_baseline:
                fb_rows(prun, "split")
            line["parity_mode"] = dict(
                dtype="f16x3: every GEMM / convolution operand a (hi, lo) fp16 pair, three v_mfma_f32_16x16x32_f16 per product, fp32 accumulate, fp32 tensors "
                      "between the DSVT kernels, [hi | lo | hi] fp16 triples between the convolutions (the reference's arithmetic is fp32: include/params.h:332)",
                value=pm["value"], unit="frames/s", ms_per_step=pm["ms_per_step"], p50_ms=pm["p50_ms"], repeats=pm["repeats"], repeat_values=pm["repeat_values"],
                value_with_sample=pm["value_with_sample"], value_without_sample=pm["value_without_sample"], frames_per_forward=FB, frames_in_flight=prun.NS * FB,
                graph_replay_equals_eager=pm["graph_replay_equals_eager"], roofline=pm["roofline"], roofline_other_kernels=pm["roofline_other_kernels"],
                note="boxes within 1e-3 of the fp32 oracle (cpu_baseline.box_err_vs_oracle.split; tests/test_split_kernels_gpu.py::test_boxes_split_mode): "
                     "north_star's target is >= 200 frames/s AND 1e-3, which the f16 headline does not meet on z / size")
            del prun, pm
        if FB > 1 and not args.no_latency_mode and args.dtype == "f16":
            # the reference's own mode beside the headline: ONE frame per forward, one in flight (graph replay), same clouds --
            # what a caller who wants latency, not throughput, gets from the same kernels; measured after the timed region
            c1 = pkg.pipeline.Caps()
            one = []
            for cl in clouds[:FRAME_POOL]:
                b1 = np.zeros((1, c1.N, 4), np.float32); b1[0, :cl.shape[0]] = cl
                one.append((torch.from_numpy(b1).to(dev), torch.tensor([cl.shape[0]], dtype=torch.int32, device=dev)))
            line["single_frame_mode"] = {"frames_per_forward": 1, "frames_in_flight": 1,
                                         "note": "same kernels, one frame at a time (the reference's mode); not the headline value"}
            for tag, kw1 in (("f16", dict(linear_compute=pkg.plugin.COMPUTE_F16, head_dtype=torch.float16)), ("split", dict(linear_compute=pkg.plugin.COMPUTE_SPLIT))):
                p1 = pkg.pipeline.DsvtPipeline(weights, caps=c1, device=dev, device_nms=not args.no_nms, **kw1)
                sin = (torch.zeros_like(one[0][0]), torch.zeros_like(one[0][1]))
                for pts1, n1 in one[:2]:
                    p1.forward(pts1, n1)
                torch.cuda.synchronize()
                sin[0].copy_(one[0][0]); sin[1].copy_(one[0][1])
                p1.capture(*sin)
                KL = 64
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
                del p1
            line["single_frame_mode"].update(value=line["single_frame_mode"]["f16"]["value"], unit="frames/s", p50_ms=line["single_frame_mode"]["f16"]["p50_ms"])
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
            line["cpu_baseline"] = cpu_baseline(c_one, frames, whole_network=None if args.no_whole_network_cpu else (weights,), mode_rows=mode_rows)
        else:
            line["cpu_baseline"] = None
    elif rank == 0:
        line["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
