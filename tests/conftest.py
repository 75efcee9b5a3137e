import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "torch-rgcn_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")       # before the first HIP call: see torch_rgcn/__init__.py


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("RGCN_TEST_FILL_UNINITIALIZED") == "1":
        # the thorough (slower) form for global memory: every torch.empty() comes back full of NaN, so a kernel that reads a buffer (or
        # the padding columns of one) before anything wrote it shows up as NaN instead of depending on what the allocator handed out
        import torch
        torch.use_deterministic_algorithms(True, warn_only=True)
        torch.utils.deterministic.fill_uninitialized_memory = True
    # a fresh checkout has no built artefacts (they are git-ignored): build the HIP library and the C oracle once
    lib = os.path.join(PKG, "torch_rgcn", "lib", "librgcn_hip.so")
    ora = os.path.join(ROOT, "oracle", "_build")
    if not os.path.isfile(lib) or not os.path.isdir(ora):
        import subprocess
        if not os.path.isfile(lib):
            subprocess.check_call(["make", "-C", os.path.join(PKG, "csrc")])
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


@pytest.fixture(autouse=True)
def _routes_restored():
    """route choices made through routes.patch(monkeypatch, ...) are undone by monkeypatch; the integers that were pushed into the
    loaded library (routes.NATIVE) are pushed again from the restored table after every test"""
    yield
    from torch_rgcn import routes
    if routes._native_sink is not None:
        routes.push_native()


@pytest.fixture(autouse=True)
def _poisoned_lds(request):
    """GPU tests start with every CU's LDS full of NaN patterns (rgcn_poison_lds, ~10 us): LDS keeps its contents between kernels, and a
    kernel that reads a word it never wrote -- and gets away with it because yesterday's garbage was finite and met a zero -- fails here
    instead of once in 25 runs (round 5: the fused tile backward).  No-op for CPU tests and without a GPU."""
    undo = None
    if request.node.get_closest_marker("gpu") is not None:
        import torch
        if torch.cuda.is_available():
            from torch_rgcn import _native
            _native.poison_lds()
            if os.environ.get("RGCN_TEST_POISON_EVERY_LAUNCH") == "1":      # the thorough (slower) form: before EVERY launch of the library
                orig, busy = _native._stream, [False]

                def stream_after_poison(device):
                    if not busy[0] and not torch.cuda.is_current_stream_capturing():
                        busy[0] = True
                        try:
                            _native.poison_lds(device)
                        finally:
                            busy[0] = False
                    return orig(device)
                _native._stream = stream_after_poison
                undo = lambda: setattr(_native, "_stream", orig)
    yield
    if undo is not None:
        undo()
