"""bench.py -- frames/s + p50 per-frame ms of the DSVT hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Four frames per forward() (their pillar rows concatenated: one launch per backbone layer for all of them, `--batch`; BASELINE
configs[3] puts four frames on each GPU per batch) on each of two streams (`--streams`) are the default.  Measured on one MI355X
(frames/s, p50 per-frame ms; a frame is done when its forward() is), frames per forward x streams:
    1 x 1  471 / 2.1      1 x 2  577 / 3.4      2 x 2  603 / 6.6
    3 x 2  644 / 9.2      4 x 1  595 / 6.7      4 x 2  663 / 11.9 (default)
`--batch 1 --streams 1` is the latency mode (the reference's own: one frame at a time).

A step = one frame through the whole pipeline (BASELINE.json configs[2]: Waymo-shaped
180k-point synthetic cloud `lidar_like(180000, seed)`, 0.32 m pillars, 468x468 BEV grid, full
4-block DSVT pillar backbone + BEV backbone + CenterHead + FilterBoxByScore), inputs already
resident in HBM when the timed region starts.  Frame-batch data parallelism: every rank
processes K frames of its own (weak scaling) and the per-frame results are gathered to rank 0
with ONE collective inside the timed region.  Rank 0 prints one JSON line.

Extra objects in the line:
  single_frame_mode  (N = 1) the same kernels with ONE frame per forward and one in flight -- the reference's own mode: frames/s and p50,
                measured live after the timed region; never the headline value.
  roofline      the dominant hand-written kernel (the MFMA linear kernel): algorithmic flops per
                launch / average launch duration, measured with HIP events around every launch
                during the timed steps, against the dense fp32-matrix peak of gfx950.
  cpu_baseline  SURVEY 8(d): the reference's HOST path -- loadData + save_result + nms_cpu (include/helper.h:28-72,
                257-283, 470-481; restated in oracle/dsvt_oracle.c) -- timed single-threaded (the reference is) on the
                host cores of the same box, fed the FilterBoxByScore rows the GPU produced for the same frames; plus a
                frame-parallel variant (one frame per thread, 32 frames), the CPU voxelize + partition restatement and the
                whole network on the CPU oracle as extra keys.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches the N ranks itself
(torch.distributed.run, one process per GPU) and fails if fewer than N GPUs are visible.
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
PMC_FILES = {1: "r02_g_batch1_pmc_traffic.json", 2: "r02_g_pmc_traffic.json", 4: "r02_h_pmc_traffic.json"}      # FETCH_SIZE / WRITE_SIZE passes, by frames per forward()
FRAME_POOL = 4                      # distinct synthetic clouds cycled through by the steps


def cpu_model():
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                return l.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(caps, frames, n_parallel_frames=32, whole_network=None):
    """SURVEY 8(d).  frames: [(points [n,4] float32 numpy, FilterBoxByScore rows [500,9] float32 numpy from the GPU, count)].
    value = frames/s of the reference's host path, ONE thread: loadData (read the .bin, size check, zero-pad to the cap) +
    save_result + nms_cpu on the GPU's own rows."""
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
                                         note=f"the whole network in fp32 on the CPU oracle, dense layers on {torch.get_num_threads()} "
                                              "torch threads, plugin restatement on 1 core", boxes=int(cnt))
    for p_ in paths:
        os.remove(p_)
    os.rmdir(tmp)
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` outside torchrun: launch the N ranks (one process per GPU) and relay rank 0's JSON line"""
    import socket
    import subprocess
    if not torch.cuda.is_available() or torch.cuda.device_count() < n:
        raise SystemExit(f"bench.py --gpus {n}: only {torch.cuda.device_count() if torch.cuda.is_available() else 0} GPU(s) visible")
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240)      # (timed frames; the un-graphed roofline sample is the last forward: 60 steps read 1.5 % lower)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--dtype", choices=["f32", "f16"], default="f16",
                    help="f16: fp16 MFMA operands / fp16 dense head, fp32 accumulate + LayerNorm/softmax/decode "
                         "(BASELINE configs[2]); f32: fp32 everywhere (the mode the 1e-3 box-parity tests run in)")
    ap.add_argument("--streams", type=int, default=2,
                    help="frames in flight per GPU: independent pipeline instances on separate HIP streams (a single "
                         "180k-point frame leaves most kernels one wave per SIMD; overlapping two frames fills the gaps)")
    ap.add_argument("--batch", type=int, default=4,
                    help="frames per forward(): their pillar rows are concatenated and every backbone layer is ONE launch for all of them "
                         "(DsvtPipeline(frames=B)); a step is still one frame, --steps must be a multiple of B (the table in this file's docstring)")
    ap.add_argument("--no-graph", action="store_true", help="launch every op from the host instead of replaying a HIP graph")
    ap.add_argument("--event-every", type=int, default=0,
                    help="roofline sample: every N-th timed step runs un-graphed, alone on the GPU, with HIP events around each launch; "
                         "0 (default) = only the FIRST forward() of the timed region (it runs alone before the streams fill: no pipeline drain; "
                         "sampling the last forward cost 8 %% at --steps 20, three samples 5-8 %% at 60)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency-mode", action="store_true", help="skip the single-frame-mode measurement (N = 1 only) that follows the timed region")
    ap.add_argument("--host-input", action="store_true", help="frames start in pinned host memory and are uploaded (n x 16 B) inside the timed region: the PCIe-inclusive rate quoted in DESIGN.md, never the headline value")
    ap.add_argument("--no-nms", action="store_true", help="stop at FilterBoxByScore (the reference engine's output) instead of the final boxes")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the per-launch HIP events (roofline = null)")
    ap.add_argument("--rccl-single", action="store_true", help="N = 1 only: create a communicator of size 1 so that the result gather "
                                                               "really goes through RCCL (SURVEY 8e: exercising the collective on one device)")
    ap.add_argument("--whole-network-cpu", action="store_true", help="cpu_baseline also runs the whole network on the CPU oracle (~10 s)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)

    # one rank builds (the in-tree .so normally travels with the snapshot and this is a no-op); the others wait
    rank0 = int(os.environ.get("RANK", "0")) == 0
    if rank0:
        G.build()
    par = G._load_file("dsvt_parallel_boot", os.path.join(G.PKG_DIR, "parallel.py"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: n_gpus would not be the ranks that ran")
    if torch.cuda.device_count() < int(os.environ.get("LOCAL_RANK", "0")) + 1:
        raise SystemExit(f"bench.py: rank {os.environ.get('RANK')} has no GPU (only {torch.cuda.device_count()} visible)")
    rank, local_rank, world = par.init(single_rank_group=args.rccl_single)
    par.barrier()
    pkg = G.load_package()
    par = pkg.parallel
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    if args.batch > 4:
        raise SystemExit("bench.py: --batch <= 4 (Points2Features packs the running count of occupied cells into 20 bits of its scan state: "
                         "4 x 468 x 468 cells fit, 5 do not, and the plugin rejects the fields)")
    FB = max(1, args.batch) if args.dtype == "f16" else 1          # (the fp32 mode has no multi-frame path)
    if args.host_input:
        FB = 1
    while args.steps % FB:          # exactly K timed frames: the largest frames-per-forward <= --batch that divides K
        FB -= 1
    caps = pkg.pipeline.Caps() if FB == 1 else pkg.pipeline.Caps.for_frames(FB)     # 196608 points per frame; pillar / window / set capacities are totals
    weights = pkg.synth.make_weights()
    f16 = args.dtype == "f16"
    use_graph = not args.no_graph
    NS = max(1, args.streams)
    streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
    pipes = [pkg.pipeline.DsvtPipeline(weights, caps=caps, device=dev,
                                       linear_compute=pkg.plugin.COMPUTE_F16 if f16 else pkg.plugin.COMPUTE_F32,
                                       head_dtype=torch.float16 if f16 else torch.float32, device_nms=not args.no_nms, frames=FB)
             for _ in range(NS)]
    pipe = pipes[0]

    # synthetic frames of this rank, resident in HBM before the timed region
    K = args.steps
    KB = K // FB                                    # forward() calls of this rank (FB frames each)
    clouds = [pkg.synth.lidar_like(args.points, seed=rank * FRAME_POOL + i) for i in range(max(FB, min(FRAME_POOL, max(K, 1))))]
    pool = []                                       # entries = the inputs of one forward(): FB consecutive clouds, frame f in rows f * caps.N ...
    for j in range(max(1, len(clouds) // FB)):
        buf = np.zeros((1, FB * caps.N, 4), np.float32); ns = []
        for f in range(FB):
            p = clouds[(j * FB + f) % len(clouds)]
            buf[0, f * caps.N:f * caps.N + p.shape[0]] = p; ns.append(p.shape[0])
        pool.append((torch.from_numpy(buf).to(dev), torch.tensor(ns, dtype=torch.int32, device=dev)))
    results = torch.zeros((K, par.ROW), dtype=torch.float32, device=dev)
    static_in = [(torch.zeros_like(pool[0][0]), torch.zeros_like(pool[0][1])) for _ in range(NS)]

    # --host-input: the frames wait in pinned host memory (where a loader thread would have read the .bin files) and only
    # the n x 16 bytes that exist + the count cross PCIe, asynchronously on the frame's stream
    host_pool = [(p_[0, :int(n_[0])].cpu().pin_memory(), n_.cpu().pin_memory(), int(n_[0])) for p_, n_ in pool] if args.host_input else None

    def pack(boxes, cnt, rows):
        """boxes [FB,500,9], cnt [FB] -> FB rows of the result buffer (two device ops, no host sync)"""
        if FB == 1:
            par.pack_result(boxes[0], cnt, rows[0])
        else:
            rows[:, :par.ROW - 1].copy_(boxes.reshape(FB, -1)); rows[:, par.ROW - 1].copy_(cnt.to(torch.float32))

    def run_frame(i, row, eager=False):
        """forward() call i (FB frames) on pipeline/stream i % NS (must be called with that stream current)"""
        s = i % NS
        pts, n = pool[i % len(pool)]
        if host_pool is not None:
            hp, hn, k = host_pool[i % len(pool)]
            static_in[s][0][0, :k].copy_(hp, non_blocking=True); static_in[s][1].copy_(hn, non_blocking=True)
            pts, n = static_in[s]
            boxes, cnt = pipes[s].replay() if (use_graph and not eager) else pipes[s].forward(pts, n)
        elif use_graph and not eager:
            static_in[s][0].copy_(pts); static_in[s][1].copy_(n)        # device-to-device refill of the graph's inputs
            boxes, cnt = pipes[s].replay()
        else:
            boxes, cnt = pipes[s].forward(pts, n)
        pack(boxes, cnt, row)

    scratch = [torch.zeros((FB, par.ROW), dtype=torch.float32, device=dev) for _ in range(NS)]
    for s in range(NS):
        with torch.cuda.stream(streams[s]):
            for i in range(max(args.warmup, 1)):
                run_frame(i * NS + s, scratch[s], eager=True)
            torch.cuda.synchronize()
            if use_graph:
                static_in[s][0].copy_(pool[0][0]); static_in[s][1].copy_(pool[0][1])
                pipes[s].capture(*static_in[s])
                for i in range(args.warmup):
                    run_frame(i * NS + s, scratch[s])
            torch.cuda.synchronize()
    # the graph replay does the frame's work: one replay against one eager (op-by-op) run of the same frame, bit for bit
    # (every kernel is deterministic), outside the timed region
    replay_equals_eager = None
    if use_graph:
        replay_equals_eager = True
        for s in range(NS):
            with torch.cuda.stream(streams[s]):
                pts, n = pool[(s + 1) % len(pool)]
                eb, ec = [t.clone() for t in pipes[s].forward(pts, n)]
                static_in[s][0].copy_(pts); static_in[s][1].copy_(n)
                gb, gc = pipes[s].replay()
                torch.cuda.synchronize()
                replay_equals_eager = replay_equals_eager and bool(torch.equal(eb, gb)) and bool(torch.equal(ec, gc)) and int(ec[0]) > 0
        if not replay_equals_eager:
            raise SystemExit("bench.py: the HIP-graph replay of a frame differs from its eager run")
    # device-side counts of each pooled frame (for the algorithmic flop count), read outside the timed region
    counts = []                                     # per pool entry: totals over its FB frames (what one launch processes)
    for pts, n in pool:
        st = pipe.voxel_stage(pts, n)
        counts.append(dict(P=int(st["P"][0]), Nk=int(st["Nk"][0]), S=[int(g[2][0]) for g in st["gss"]]))
    torch.cuda.synchronize()

    prof = None if args.no_kernel_events else {"DsvtLinearPlugin": [], "DsvtEncoderMlpPlugin": [], "DsvtSetAttentionPlugin": [],
                                               "DsvtConv2dPlugin": [], "DsvtPillarFeatureNetPlugin": [], "DsvtPosEmbedPlugin": []}
    marks = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(KB)]
    # warm-up of the one collective: RCCL sets up its channels lazily at the first call of each kind (tens of ms: measured 45 ms on a size-1
    # communicator, as much as 23 frames), which is start-up cost, not a property of the frame path
    if world > 1 or args.rccl_single:
        par.gather_results(results, K * world, rank, world, force_collective=args.rccl_single)
    par.barrier(); torch.cuda.synchronize()
    sampled = 0
    t0 = time.perf_counter()
    for i in range(KB):
        # roofline sample: every event_every-th step is launched op by op, alone on the GPU, with HIP events
        # around each linear launch (events cannot bracket kernels inside a graph replay); it stays inside the
        # timed region
        if args.event_every > 0:
            ev = prof is not None and (not use_graph or (i * FB) % args.event_every == (args.event_every // 2) // FB * FB)
        else:
            ev = prof is not None and (not use_graph or i == 0)       # the FIRST forward: nothing is in flight yet, so running it alone drains no other stream
        if ev and (use_graph or NS > 1):
            torch.cuda.synchronize()
        pkg.plugin.PROFILE = prof if ev else None
        with torch.cuda.stream(streams[i % NS]):
            marks[i][0].record()
            run_frame(i, results[i * FB:(i + 1) * FB], eager=ev)
            marks[i][1].record()
        if ev and (use_graph or NS > 1):
            torch.cuda.synchronize()
        sampled += ev
    for s in streams:
        torch.cuda.current_stream().wait_stream(s)
    gathered = par.gather_results(results, K * world, rank, world, force_collective=args.rccl_single)          # the one collective of the path
    par.barrier(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pkg.plugin.PROFILE = None
    dt = par.max_over_ranks(dt, dev)

    frame_ms = np.array([marks[i][0].elapsed_time(marks[i][1]) for i in range(KB)])      # a frame is done when its forward() is
    roofline, roofline_all = None, []
    if prof is not None and sampled:
        pm = {}
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", PMC_FILES[FB])))
        except Exception:
            pass

        def pmc_traffic(substr):
            ks = [k for k in pm if substr in k]
            return round(pm[ks[0]]["traffic_mb_per_launch"] * 1e6) if ks else None

        def work(pl, c):
            """(flops, algorithmic HBM bytes) of one launch of plugin `pl` on a frame with counts c"""
            f = pl.fields
            if pl.plugin_type == "DsvtLinearPlugin":
                rows = c[pl.rows_kind]          # "Nk" for the two PFN linears, "P" for everything on voxel rows
                K_, N_ = f["in_features"], f["out_features"]
                esz = 2 if f.get("input_half") else 4
                out_b = {0: 4, 1: 2, 2: 6}[f.get("output_mode", 0)]
                return (2.0 * rows * K_ * N_,
                        rows * (K_ * esz * (2 if (f.get("add_cols") and not f.get("add_gather_width")) else 1) + 12 * bool(f.get("add_gather_width"))
                                + 4 * N_ * f.get("num_layer_norms", 0) + out_b * N_)
                        + K_ * N_ * (2 if f16 else 4))
            if pl.plugin_type == "DsvtEncoderMlpPlugin":
                rows = c["P"]                   # att16 + x (+ xb) in, x' fp32 + fp16 out, weights once
                return (2.0 * rows * (192 * 192 + 2 * 192 * 384),
                        rows * 192 * (2 + 4 + 4 * f.get("has_block_norm", 0) + 4 + 2) + 2 * (192 * 192 + 2 * 192 * 384))
            if pl.plugin_type == "DsvtSetAttentionPlugin":
                S = c["S"][0 if pl.win == 0 else 1]
                esz = 2 if f.get("io_half") else 4
                return (4.0 * 36 * 36 * 192 * S, S * 36 * 192 * esz * 3 + c["P"] * 192 * esz)
            if pl.plugin_type == "DsvtPosEmbedPlugin":
                L = f["num_layers"]
                return (2.0 * L * c["P"] * (2 * 192 + 192 * 192), c["P"] * 16 + L * (c["P"] * 192 * 2 + 2 * 192 * 192))
            if pl.plugin_type == "DsvtPillarFeatureNetPlugin":
                return (2.0 * c["Nk"] * (10 * 96 + 96 * 192) + 2.0 * c["P"] * 96 * 192, c["Nk"] * 40 + c["P"] * 192 * 6)
            if pl.plugin_type == "DsvtConv2dPlugin":
                Ho = (f["in_height"] + 2 * f["padding"] - f["kernel_size"]) // f["stride"] + 1
                up = f.get("pixel_shuffle", 1)
                taps = f["kernel_size"] ** 2
                # (one enqueue = FB per-frame launches through the C ABI's batched enqueue)
                return (FB * 2.0 * Ho * Ho * up * up * f["out_channels"] * taps * f["in_channels"],
                        FB * (2 * f["in_height"] ** 2 * f["in_channels"] + (4 if f.get("out_f32") else 2) * Ho * Ho * up * up * f["out_channels"]
                              + 2 * up * up * f["out_channels"] * taps * f["in_channels"]))
            return (0.0, 0.0)

        resident_qkv = f16 and FB >= 3          # (csrc/linear.hip: row capacity of three or more frames -> the resident-weights kernel)
        meta = {"DsvtLinearPlugin": (("linear_f16_resident_kernel (QKV: half of W_qkv resident in LDS per CU, waves walk 16-row tiles, v_mfma_f32_16x16x32_f16)" if resident_qkv else
                                      "linear_f16_rows_kernel (QKV: all column chunks of a row tile per workgroup, v_mfma_f32_16x16x32_f16, weights by LDS-DMA)") if f16 else "linear_f32_kernel (v_mfma_f32_16x16x4_f32)",
                                     "hbm" if f16 else "mfma", ("linear_f16_resident_kernel" if resident_qkv else "linear_f16_rows_kernel") if f16 else "linear_f32_kernel<true>"),
                "DsvtEncoderMlpPlugin": ("encoder_mlp_stream_kernel (out-proj+LN -> FC1+GELU -> FC2+LN+LN, v_mfma_f32_16x16x32_f16, weights by LDS-DMA)", "hbm", "encoder_mlp_stream_kernel"),
                "DsvtSetAttentionPlugin": ("set_attention_f16_kernel (v_mfma_f32_16x16x32_f16)" if f16 else "set_attention_kernel (v_mfma_f32_16x16x4_f32)", "hbm",
                                           "set_attention_f16_kernel" if f16 else "set_attention_kernel"),
                "DsvtPosEmbedPlugin": ("posembed_batched_kernel (8 position-embedding MLPs, v_mfma_f32_16x16x32_f16)", "hbm", "posembed_batched_kernel"),
                "DsvtPillarFeatureNetPlugin": ("pfn_kernel (both PFN layers + scatter-max, v_mfma_f32_16x16x4_f32 + 16x16x32_f16)", "mfma", "pfn_kernel"),
                "DsvtConv2dPlugin": ("conv_wide_kernel / conv_halo_kernel / conv_f16_kernel (implicit GEMM, v_mfma_f32_16x16x32_f16)", "mfma", "conv_wide_kernelILi8ELi8")}
        for ptype, lst in prof.items():
            if not lst:
                continue
            per_frame = len(lst) // sampled
            tot_ms = tot_fl = tot_by = 0.0
            for j, (e0, e1, pl) in enumerate(lst):
                fl, by = work(pl, counts[(j // per_frame) % len(pool)])
                tot_ms += e0.elapsed_time(e1); tot_fl += fl; tot_by += by
            n_l = len(lst)
            avg_ms = tot_ms / n_l
            tfl, gbs = tot_fl / n_l / (avg_ms * 1e-3) / 1e12, tot_by / n_l / (avg_ms * 1e-3) / 1e9
            kname, bound, pmk = meta[ptype]
            peak_tf = (PEAK_F32_MATRIX_TFLOPS if (ptype == "DsvtPillarFeatureNetPlugin" or not f16) else PEAK_F16_MATRIX_TFLOPS)      # pfn: 2/3 of its MFMA cycles are fp32
            r = dict(kernel=kname, bound=bound)
            if bound == "hbm":
                r.update(achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4), mfma_tflops=round(tfl, 2))
            else:
                r.update(achieved=round(tfl, 2), peak=peak_tf, unit="TFLOP/s", frac=round(tfl / peak_tf, 4), hbm_gbs=round(gbs, 1))
            r.update(traffic=pmc_traffic(pmk), launches_per_frame=per_frame, sampled_frames=sampled, avg_launch_us=round(1e3 * avg_ms, 2),
                     ms_per_frame=round(tot_ms / sampled, 3), algorithmic_gflop_per_launch=round(tot_fl / n_l / 1e9, 3),
                     algorithmic_mb_per_launch=round(tot_by / n_l / 1e6, 2))
            if r["traffic"] is not None:
                r["traffic_source"] = "profiles/" + PMC_FILES[FB]
            # the weight matrices are counted ONCE in the algorithmic bytes, but every workgroup streams its own copy L2 -> LDS: what the
            # memory system carries per launch beside the activations (DESIGN.md "what bounds the backbone kernels")
            wg_weights = {"DsvtEncoderMlpPlugin": 2 * (192 * 192 + 2 * 192 * 384), "DsvtLinearPlugin": 2 * 192 * 576 if f16 else 0}.get(ptype, 0)
            if wg_weights and f16:
                c0 = counts[0]
                need = -(-c0["P"] // (16 * 256))                       # the kernels' tile plan: 8 .. 10 live waves of 16 rows, two rounds / two per CU beyond
                if need > 10:
                    need = -(-c0["P"] // (32 * 256))
                nwg = -(-c0["P"] // (16 * (max(8, need) if need <= 10 else 8)))
                if FB >= 3:                                            # three or more frames per launch: the other kernel of each pair
                    nwg = 256 // 2 if ptype == "DsvtLinearPlugin" else -(-c0["P"] // 128)      # resident QKV: each CU loads its half once; MLP <2,4>: 128-row workgroups
                r["weights_restreamed_mb_per_launch"] = round(nwg * wg_weights / 1e6, 1)
                r["fabric_gbs_incl_weight_stream"] = round((tot_by / n_l + nwg * wg_weights) / (avg_ms * 1e-3) / 1e9, 1)
            roofline_all.append((ptype, r))
        # the headline roofline object = the hot path's kernel with the largest share of the frame
        hot = [x for x in roofline_all if x[0] != "DsvtConv2dPlugin"] or roofline_all
        roofline = max(hot, key=lambda x: x[1]["ms_per_frame"])[1]

    if rank == 0:
        total_frames = K * world
        if world > 1:
            assert gathered is not None and gathered.shape[0] == total_frames
        line = {
            "metric": "frames/sec (p50 per-frame ms in p50_ms), 180k-pt Waymo pillar DSVT",
            "value": round(total_frames / dt, 3), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / K, 4), "p50_ms": round(float(np.median(frame_ms)), 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic" + (" (uploaded from pinned host memory inside the timed region)" if args.host_input else ""),
            "config": {"workload": f"BASELINE configs[2]: lidar_like({args.points}, seed) Waymo-shaped cloud, 0.32 m pillars, "
                                   "468x468 BEV, full 4-block DSVT pillar backbone + BEV ResNet + CenterHead + top-K decode + "
                                   "FilterBoxByScore" + ("" if args.no_nms else " + rotated NMS (final boxes)") + "; seeded random weights (dsvt.wts is not shipped)",
                       "frames_per_gpu": K, "parallelism": f"frame-batch dp{world}, one result gather",
                       "frames": f"{min(FRAME_POOL, max(K, 1))} distinct clouds per rank, seeds rank * {FRAME_POOL} + i, cycled (BASELINE configs[3]: "
                                 f"32 frames lidar_like(180000, 0..31), 4 per GPU on 8 GPUs)",
                       "result_gather": ("rccl gather" if world > 1 else "rccl gather (communicator of size 1)" if args.rccl_single
                                         else "none (single process)"),
                       "graph_replay_equals_eager": replay_equals_eager,
                       "launch": "hip-graph replay per frame" if use_graph else "host launch per op",
                       "frames_in_flight": NS * FB, "frames_per_forward": FB,
                       "caps": dict(points=caps.N, pillars=caps.P, windows=caps.W, sets=caps.S, overflow_free=caps.overflow_free()),
                       "frame0": counts[0]},
            "roofline": roofline,
            "roofline_other_kernels": [r for _, r in roofline_all if r is not roofline],
        }
        if world == 1 and f16 and FB > 1 and not args.no_latency_mode:
            # the reference's own mode beside the headline: ONE frame per forward, one in flight (graph replay), same clouds --
            # what a caller who wants latency, not throughput, gets from the same kernels; measured after the timed region
            c1 = pkg.pipeline.Caps()
            p1 = pkg.pipeline.DsvtPipeline(weights, caps=c1, device=dev, linear_compute=pkg.plugin.COMPUTE_F16, head_dtype=torch.float16,
                                           device_nms=not args.no_nms)
            one = []
            for cl in clouds[:FRAME_POOL]:
                b1 = np.zeros((1, c1.N, 4), np.float32); b1[0, :cl.shape[0]] = cl
                one.append((torch.from_numpy(b1).to(dev), torch.tensor([cl.shape[0]], dtype=torch.int32, device=dev)))
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
            line["single_frame_mode"] = {"frames_per_forward": 1, "frames_in_flight": 1, "frames": KL, "value": round(KL / dl, 1), "unit": "frames/s",
                                         "p50_ms": round(float(np.median([a.elapsed_time(b) for a, b in ev1])), 4),
                                         "note": "same kernels, one frame at a time (the reference's mode); not the headline value"}
            del p1
        if not args.no_cpu_baseline and world == 1:
            # the FilterBoxByScore rows of the pooled frames as the GPU produced them (the reference's D2H payload)
            frames = []
            nms_op, pipe.nms = pipe.nms, None
            for pts, n in pool:
                fb = pipe.forward(pts, n)
                torch.cuda.synchronize()
                for f in range(FB):
                    k = int(n[f])
                    frames.append((pts[0, f * caps.N:f * caps.N + k].cpu().numpy(), fb[0][f].cpu().numpy().copy(), int(fb[1][f])))
            pipe.nms = nms_op
            line["cpu_baseline"] = cpu_baseline(caps, frames, whole_network=(weights,) if args.whole_network_cpu else None)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))


if __name__ == "__main__":
    main()
