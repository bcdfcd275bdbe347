"""Out-of-bounds READS (and writes) of the frame pipeline, caught by unmapped guard pages.

tests/guard_alloc/guard_alloc.cpp is a torch pluggable allocator that gives every device tensor of a child process its own virtual range
[unmapped | pages | unmapped] (HIP virtual-memory API) with the tensor flush against one of the guards (16-byte alignment): an access past
that side of ANY buffer -- inputs, outputs, workspaces, the pipeline's own tensors -- is a GPU memory access fault that kills the child,
instead of landing unnoticed in a neighbouring block of the caching allocator (which is all test_no_plugin_writes_outside_its_buffers'
0xA5 bands can see, and only for writes).  The children run the reference frames and the 180k-point cloud eagerly, one and four frames
per forward, in the fp16 and the split-precision (fp32-grade) modes, once with the end of every buffer on a guard and once with its start.
Since round 5 the plugins' own device memory (packed weights and their slack rows, tables, LayerNorm parameters ...) comes from the same allocator
through the C ABI's dsvtSetGpuAllocator, so a prefetch or LDS-DMA past the end of one of those buffers is caught as well.
A freed range is never mapped again (a use after free faults too; re-mapping a translated address is not safe on this stack: DESIGN 5)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GA = os.path.join(HERE, "guard_alloc")


def _build():
    so, src = os.path.join(GA, "guard_alloc.so"), os.path.join(GA, "guard_alloc.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", src, "-o", so])
    return so


def _child(args, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(GA, "run_frames.py")] + args, capture_output=True, text=True, timeout=timeout, env=e)


def test_the_guard_pages_catch_an_out_of_bounds_read():
    _build()
    r = _child(["selftest"], timeout=300)
    assert r.returncode != 0 and "SELFTEST-SURVIVED" not in r.stdout, r.stdout[-400:] + r.stderr[-400:]
    # (the runtime reports the fault as an abort "Memory access fault by GPU ..." or, on other boxes / with core dumps off, as hipErrorIllegalAddress)
    assert "Memory access fault" in r.stderr or "fault" in r.stderr.lower() or "illegal memory access" in r.stderr.lower(), r.stderr[-600:]


@pytest.mark.parametrize("front", [0, 1])
@pytest.mark.parametrize("mode", ["f16", "split"])
def test_no_kernel_of_the_frame_touches_memory_outside_its_buffers(mode, front):
    _build()
    r = _child([mode, "1,4"], env={"DSVT_GUARD_FRONT": str(front)})
    assert r.returncode == 0 and "GUARD-OK" in r.stdout, (r.stdout[-600:], r.stderr[-1200:])
    # (round 5) the plugins' OWN device memory -- packed weights, tables, parameters -- sat behind guard pages too (dsvtSetGpuAllocator)
    assert int(r.stdout.strip().splitlines()[-1].rsplit(":", 1)[1]) > 100, r.stdout[-300:]
