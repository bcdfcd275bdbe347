"""dsvt-ai-trt_amd -- MI355X-native (gfx950) implementation of the DSVT-AI-TRT hot path.

The directory name carries a hyphen (it is the reference's name + `_amd`), so it is loaded
through `__graft_entry__.load_package()` / `tests/conftest.py`, which register it under the
importable module name `dsvt_ai_trt_amd`.

Importing the package loads libdsvt_hip.so (hand-written HIP kernels behind the C ABI of
include/dsvt_plugin.h).  If the library has not been built the import fails: there is no
CPU or PyTorch fallback for the product path.
"""
from . import plugin          # noqa: F401  (loads libdsvt_hip.so or raises)
from . import synth           # noqa: F401
from . import pipeline        # noqa: F401
from . import pipeline3d      # noqa: F401
from . import parallel        # noqa: F401
from . import hostio          # noqa: F401
from . import detect          # noqa: F401
