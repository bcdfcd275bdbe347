"""The bench's throughput loop (NS pipelines on NS streams, FB frames per forward, HIP-graph replay) as an A/B tool: any build through
DSVT_HIP_LIB, the ablation switches of the ablate build through their DSVT_* variables (bench.py itself refuses both).
    [DSVT_HIP_LIB=...] python tools/two_stream_fps.py [mode: split | splitmx | f16] [FB] [NS] [forwards] [repeats]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "split"
FB = int(sys.argv[2]) if len(sys.argv) > 2 else 4
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
KB = int(sys.argv[4]) if len(sys.argv) > 4 else 60
REP = int(sys.argv[5]) if len(sys.argv) > 5 else 5
kw = {"split": dict(linear_compute=P.COMPUTE_SPLIT), "splitmx": dict(linear_compute=P.COMPUTE_SPLIT, head_mx=True), "f16": dict(linear_compute=P.COMPUTE_F16, head_dtype=torch.float16)}[mode]
caps = pkg.pipeline.Caps() if FB == 1 else pkg.pipeline.Caps.for_frames(FB)
w = pkg.synth.make_weights()
pool = []
for c in range(NS * 2):
    buf = np.zeros((1, FB * caps.N, 4), np.float32); ns = []
    for f in range(FB):
        p = pkg.synth.lidar_like(180000, c * FB + f); buf[0, f * caps.N:f * caps.N + len(p)] = p; ns.append(len(p))
    pool.append((torch.from_numpy(buf).to(dev), torch.tensor(ns, dtype=torch.int32, device=dev)))
streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
pipes = [pkg.pipeline.DsvtPipeline(w, caps=caps, device=dev, device_nms=True, frames=FB, **kw) for _ in range(NS)]
sin = [(torch.zeros_like(pool[0][0]), torch.zeros_like(pool[0][1])) for _ in range(NS)]
for s in range(NS):
    with torch.cuda.stream(streams[s]):
        for _ in range(2): pipes[s].forward(*pool[s])
        torch.cuda.synchronize()
        sin[s][0].copy_(pool[s][0]); sin[s][1].copy_(pool[s][1])
        pipes[s].capture(*sin[s])
        for _ in range(3): pipes[s].replay()
        torch.cuda.synchronize()
vals = []
for rep in range(REP):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(KB):
        s = i % NS
        with torch.cuda.stream(streams[s]):
            pts, n = pool[i % len(pool)]
            sin[s][0].copy_(pts); sin[s][1].copy_(n)
            pipes[s].replay()
    torch.cuda.synchronize()
    vals.append(KB * FB / (time.perf_counter() - t0))
print(f"{mode} {FB} frames x {NS} streams, {KB} forwards: " + " / ".join(f"{v:.1f}" for v in vals) + f" frames/s (median {sorted(vals)[len(vals) // 2]:.1f})  lib={os.path.basename(os.environ.get('DSVT_HIP_LIB', 'libdsvt_hip.so'))} DSVT_CONV_DBG={os.environ.get('DSVT_CONV_DBG', '0')}", flush=True)
