"""Route selection as ONE object instead of environment reads scattered over the call paths.

Every switch that picks between kernels / plans (they all compute the same result) lives in `routes`: a flat table
name -> string value (or None = "not set: the code's own default / heuristic").  The environment only SEEDS the table, once, when
this module is imported (`RGCN_BWD_KERNEL=lean python ...` still works for tools/ and for bisecting); after that the process
environment is never looked at again -- behaviour does not depend on when somebody changed os.environ.

    from torch_rgcn import routes
    routes.get("bwd_kernel", "blk")            # what the library reads
    routes.set("deterministic", "1")           # programmatic choice
    with routes.override(bwd="split"): ...     # scoped (tests, experiments)

Names are the old variable names without the RGCN_ prefix, lower case; docs: DESIGN.md section 7.  NATIVE names are tuning
integers of librgcn_hip.so itself: they are pushed through rgcn_set_option (the library has no getenv).  The two
timing-only switch with WRONG results (bwd_abl) exists in the ablation build only
(make -C torch-rgcn_amd/csrc abl, loaded through RGCN_HIP_LIB by tools/); the shipped library refuses them.
"""
import contextlib
import os

NAMES = (
    "deterministic", "relabel", "tile_rows", "bwd_tile_rows", "bwd", "bwd_kernel", "bwd_blk_cap", "twopass", "pad16",
    "featureless_csr", "dist_comm", "dist_slabs", "deferred_checks", "block_path", "block_fwd", "basis_path",
    "wgrad_tiles", "wgrad_item_chunks", "wgrad", "spmm_csr", "sparse_path", "graph_build", "fbasis_inplace_mb", "fbasis",
    "distmult_bwd", "diag_path", "capture", "fbasis_tile", "pad16_view", "softwin", "softwin_build", "bwd_own", "own_rows_cap", "fwd_rows_cap",
)
NATIVE = ("bwd_nw", "gemm_bm", "spmm_u", "wgrad_rg", "wgrad_u", "bwd_abl")

_values = {}
_native_sink = None        # set by _native.lib(): callable(name, int) -> pushes one option into the loaded library


def _seed():
    _values.clear()
    for n in NAMES + NATIVE:
        v = os.environ.get("RGCN_" + n.upper())
        if v is not None and v != "":
            _values[n] = v


_seed()


def get(name, default=None):
    assert name in NAMES or name in NATIVE, f"unknown route {name!r}"
    return _values.get(name, default)


def is_set(name):
    return get(name) is not None


def flag(name, default=False):
    v = get(name)
    return default if v is None else v == "1"


def push_native(name=None):
    """hand the NATIVE options (one, or all that are set) to the loaded library"""
    if _native_sink is None:
        return
    for n in ((name,) if name else NATIVE):
        v = _values.get(n)
        _native_sink(n, None if v is None else int(v))


def set(name, value):     # noqa: A001  (module-level API: routes.set)
    assert name in NAMES or name in NATIVE, f"unknown route {name!r}"
    if value is None:
        _values.pop(name, None)
    else:
        _values[name] = str(value)
    if name in NATIVE:
        push_native(name)


@contextlib.contextmanager
def override(**kw):
    saved = {k: _values.get(k) for k in kw}
    try:
        for k, v in kw.items():
            set(k, v)
        yield
    finally:
        for k, v in saved.items():
            set(k, v)


def patch(monkeypatch, name, value):
    """tests: routes.patch(monkeypatch, "bwd_kernel", "lean") -- undone with the test (NATIVE ones by the autouse fixture in
    tests/conftest.py, which pushes the table again)"""
    assert name in NAMES or name in NATIVE, f"unknown route {name!r}"
    if value is None:
        monkeypatch.delitem(_values, name, raising=False)
    else:
        monkeypatch.setitem(_values, name, str(value))
    if name in NATIVE:
        push_native(name)


def snapshot():
    return dict(_values)
