"""Builds libdsvt_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so sits next to this
file so that it travels with the source snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libdsvt_hip.so")
OBJ = os.path.join(HERE, "build")

# (source, extra flags).  -ffp-contract=off where results must follow the reference's
# expression order (cell indices, features, LayerNorm); the MFMA kernels use explicit fma.
SOURCES = [
    ("c_api.hip", []),
    ("points2features.hip", ["-ffp-contract=off"]),
    ("pillar_ops.hip", ["-ffp-contract=off"]),
    ("partition_ops.hip", ["-ffp-contract=off"]),
    ("linear.hip", []),
    ("attention.hip", []),
    ("conv.hip", []),
    ("conv_rows.hip", []),
    ("mlp.hip", []),
    # -fno-honor-nans: fmaxf() is llvm.maxnum, which without it costs THREE v_max_f32 (both operands canonicalised first); pfn_kernel is bound
    # by VALU issue and its maxima run over MFMA sums of finite inputs
    ("pfn.hip", ["-fno-honor-nans"]),
    ("decode.hip", []),
    ("nms.hip", ["-ffp-contract=off"]),
    ("voxel_pool.hip", ["-ffp-contract=off"]),
]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math",
          "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, ablate=False):
    """ablate=True: libdsvt_hip_ablate.so (-DDSVT_ABLATE: the DSVT_* trace / ablation / A-B switches of csrc/ are read from the environment).
    The product library reads none; tools/ load the other one through DSVT_HIP_LIB."""
    out, obj_dir = (os.path.join(HERE, "libdsvt_hip_ablate.so"), os.path.join(HERE, "build_ablate")) if ablate else (OUT, OBJ)
    os.makedirs(obj_dir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "dsvt_plugin.h"))
    objs = []
    procs = []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(obj_dir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc()] + COMMON + extra + (["-DDSVT_ABLATE"] if ablate else []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        log, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{log.decode()}")
        if verbose and log:
            print(log.decode(), file=sys.stderr)
    if force or procs or _stale(out, objs):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
        subprocess.check_call(cmd)
    if not ablate:
        build_host(force)
    return out


HOST_SRC = os.path.join(HERE, "host", "dsvt_detect.cpp")
HOST_EXE = os.path.join(HERE, "dsvt_detect")


def build_host(force=False):
    """The C++ host executable (the reference's `dsvt-ai-trt -d`, src/dsvt-ai-trt.cpp:1771-1970) above the C ABI: only
    include/dsvt_plugin.h + the HIP runtime.  -ffp-contract=off: its fp32 weight folding must round like numpy's."""
    hdr = os.path.join(HERE, "..", "include", "dsvt_plugin.h")
    if force or _stale(HOST_EXE, [HOST_SRC, hdr, OUT]):
        subprocess.check_call([hipcc(), "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-I", os.path.join(HERE, "..", "include"),
                               HOST_SRC, "-o", HOST_EXE, "-L", HERE, "-l:libdsvt_hip.so", "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"])
    return HOST_EXE


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, ablate="--ablate" in sys.argv))
