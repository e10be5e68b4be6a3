import os
import sys

# before anything imports torch (MIOpen reads its environment when the library is loaded): fp32 Winograd convolutions off, see
# <package>/__init__.py.  fp32 is the parity mode; the 16-bit modes do not reach MIOpen for 3x3 stride-1 convolutions.
os.environ.setdefault("MIOPEN_DEBUG_CONV_WINOGRAD", "0")
# library-GEMM plans time the heuristic's candidates once per shape; the tests see hundreds of shapes and need no speed record
os.environ.setdefault("DGTD_GEMM_CANDIDATES", "4")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # fp16 op cases run through ONE host binding layer (the C++ one the training step uses): both layers call the same C ABI and the
    # bf16 / fp32 cases already cover the Python layer, so the duplicate would only lengthen the GPU run
    keep, drop = [], []
    for item in items:
        (drop if ("python-bindings" in item.nodeid and "float16" in item.nodeid) else keep).append(item)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    # the filler's shared value file (oracle/filler.py: 458 MB in /dev/shm for the processes the 2-rank tests spawn) lives for one session
    try:
        from oracle import filler
        for f in filler._disk_paths():
            if os.path.exists(f):
                os.remove(f)
    except Exception:
        pass
