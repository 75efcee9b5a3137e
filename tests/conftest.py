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
