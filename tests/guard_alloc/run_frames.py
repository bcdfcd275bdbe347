"""Child process of tests/test_guard_pages_gpu.py: every device allocation of this process comes from guard_alloc.so (unmapped pages around each
buffer), then the frame pipeline runs eagerly in the requested modes.  Prints GUARD-OK at the end; an out-of-bounds access aborts the process."""
import os, sys, subprocess
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np, torch
so = os.path.join(HERE, "guard_alloc.so")
if os.environ.get("DSVT_GUARD_OFF", "0") == "0":
    alloc = torch.cuda.memory.CUDAPluggableAllocator(so, "guard_malloc", "guard_free")
    torch.cuda.memory.change_current_allocator(alloc)
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
if os.environ.get("DSVT_GUARD_OFF", "0") == "0" and os.environ.get("DSVT_GUARD_PLUGINS", "1") != "0":
    # the torch allocator only covers TENSORS.  The plugins' own device memory (packed weights incl. their "+ slack" rows, position /
    # scale tables, LayerNorm parameters, Map2Bev's coordinate list, the convolutions' zero rows) comes from the same guard allocator through
    # dsvtSetGpuAllocator (the C ABI's IGpuAllocator): an LDS-DMA or prefetch that runs past the end of one of THOSE buffers dies on its guard page too
    import ctypes as C
    ga = C.CDLL(so)
    ga.guard_malloc.restype = C.c_void_p; ga.guard_malloc.argtypes = [C.c_ssize_t, C.c_int, C.c_void_p]
    ga.guard_free.restype = None; ga.guard_free.argtypes = [C.c_void_p, C.c_ssize_t, C.c_int, C.c_void_p]
    n_plugin_blocks = [0]
    def _alloc(n):
        n_plugin_blocks[0] += 1
        return ga.guard_malloc(n, 0, None)
    P.set_gpu_allocator(_alloc, lambda p_: ga.guard_free(p_, 0, 0, None))
trace = os.environ.get("DSVT_GUARD_TRACE", "0") != "0"
if trace:        # name the plugin whose launch faults: synchronise after every enqueue
    call = P.Plugin.__call__
    def traced(self, *a, **k):
        print("enqueue", self.plugin_type, {k_: v for k_, v in self.fields.items() if not hasattr(v, "__len__")}, flush=True)
        r = call(self, *a, **k); torch.cuda.synchronize()
        import zlib
        print("   out crc", [f"{zlib.crc32(t.cpu().contiguous().view(torch.uint8).numpy().tobytes()):08x}" for t in r], flush=True)
        return r
    P.Plugin.__call__ = traced
dev = torch.device("cuda:0")
if len(sys.argv) > 1 and sys.argv[1] == "selftest":
    # the harness must be able to fail: a GELU launch told to read 4096 rows of a 16-row buffer has to die on the guard page (a READ)
    import ctypes as C
    op = P.add_gelu_op(4096, 384)
    x = torch.zeros((1, 16, 384), dtype=torch.float32, device=dev); out = torch.zeros((1, 4096, 384), dtype=torch.float32, device=dev)
    cnt = torch.tensor([4096], dtype=torch.int32, device=dev)
    ind = (P.PluginTensorDesc * 2)(P._desc((1, 4096, 384), P.DT_FLOAT), P._desc((1,), P.DT_INT32))
    outd = (P.PluginTensorDesc * 1)(P._desc((1, 4096, 384), P.DT_FLOAT))
    rc = P.LIB.dsvtPluginEnqueue(op._h, ind, outd, (C.c_void_p * 2)(x.data_ptr(), cnt.data_ptr()), (C.c_void_p * 1)(out.data_ptr()), None,
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    print("SELFTEST-SURVIVED rc", rc, flush=True)
    sys.exit(0)
w = pkg.synth.make_weights()
modes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["f16", "split"]
frames = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 4]
npts = int(sys.argv[3]) if len(sys.argv) > 3 else 180000
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
for mode in modes:
    kw = dict(linear_compute=P.COMPUTE_F16, head_dtype=torch.float16) if mode == "f16" else dict(linear_compute=P.COMPUTE_SPLIT) if mode == "split" else dict(linear_compute=P.COMPUTE_SPLIT, head_mx=False) if mode == "split3" else dict(linear_compute=P.COMPUTE_F32)
    for FB in frames:
        caps = pkg.pipeline.Caps() if FB == 1 else pkg.pipeline.Caps.for_frames(FB)
        pipe = pkg.pipeline.DsvtPipeline(w, caps=caps, device=dev, device_nms=True, frames=FB, **kw)
        clouds = [pkg.synth.lidar_like(npts, seed=s) for s in range(FB)]
        if FB == 1:         # the reference's frames too (small clouds: other tile regimes)
            clouds += [cases.load_frame(nm, caps.N)[0][:cases.load_frame(nm, caps.N)[1]] for nm in ("000000", "000004")]
        for j in range(0, len(clouds), FB):
            buf = np.zeros((1, FB * caps.N, 4), np.float32); ns = []
            for f in range(FB):
                p = clouds[(j + f) % len(clouds)]; buf[0, f * caps.N:f * caps.N + len(p)] = p; ns.append(len(p))
            boxes, cnt = pipe.forward(torch.from_numpy(buf).to(dev), torch.tensor(ns, dtype=torch.int32, device=dev))
            torch.cuda.synchronize()
            import zlib
            print(f"mode {mode} frames {FB} cloud {j}: boxes {[int(c) for c in cnt]} crc {zlib.crc32(boxes.cpu().numpy().tobytes()):08x}", flush=True)
        del pipe
print("GUARD-OK", "plugin-owned blocks behind guard pages:", n_plugin_blocks[0] if "n_plugin_blocks" in dir() else 0, flush=True)
