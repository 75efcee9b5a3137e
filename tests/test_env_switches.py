"""routing is an API (torch_rgcn.routes), the environment only seeds it, and no switch of the shipped library
produces wrong results.  Checked here: every route is documented in DESIGN.md section 7, the package reads no RGCN_* variable
outside routes.py (plus the few plain ones section 7 names), csrc/ has no getenv, librgcn_hip.so exports no ablation entry point and
refuses the ablation options."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "torch-rgcn_amd")
PLAIN = {"RGCN_HIP_LIB", "RGCN_SYNTHETIC", "RGCN_DATA", "RGCN_CPU_THREADS", "RGCN_DIST_BACKEND", "RGCN_FORCE_DIST", "RGCN_BENCH_ONE_DEVICE", "RGCN_BENCH_DETAIL"}
SEMANTIC = {"RGCN_DEFERRED_CHECKS", "RGCN_DETERMINISTIC", "RGCN_SYNTHETIC", "RGCN_BWD_ABL"}
ENV_READ = re.compile(r'(?:os\.environ\.get\(\s*|os\.environ\[\s*|getenv\(\s*|setdefault\(\s*)"(RGCN_[A-Z0-9_]+)"|"(RGCN_[A-Z0-9_]+)"\s+(?:not\s+)?in\s+os\.environ')


def _design_section7():
    text = open(os.path.join(ROOT, "DESIGN.md"), encoding="utf-8").read()
    a = text.index("## 7. Routes")
    return text[a:text.index("\n## 8.", a)]


def _files(base, exts):
    path = os.path.join(ROOT, base)
    if os.path.isfile(path):
        yield path
        return
    for d, _, files in os.walk(path):
        if os.sep + "build" in d:
            continue
        for f in files:
            if f.endswith(exts):
                yield os.path.join(d, f)


def test_every_route_is_documented_and_nothing_documented_is_dead():
    from torch_rgcn import routes
    assert len(routes.NAMES) + len(routes.NATIVE) <= 40, "VERDICT r4 #9: the route table stays at 40 switches or fewer"
    sec = _design_section7()
    sec = sec[sec.index("| route (seeded by)"):]           # (the prose above the table also names the switches round 5 removed)
    names = {"RGCN_" + n.upper() for n in routes.NAMES + routes.NATIVE}
    listed = set(re.findall(r"`(RGCN_[A-Z0-9_]+)(?:=[^`]*)?`", sec))
    assert not (names - listed), f"routes missing from DESIGN.md section 7: {sorted(names - listed)}"
    assert not (listed - names - PLAIN), f"DESIGN.md section 7 lists switches that are neither routes nor plain variables: {sorted(listed - names - PLAIN)}"
    used = set()
    for path in _files("torch-rgcn_amd", (".py",)):
        used |= set(re.findall(r'routes\.(?:get|flag|is_set|set|patch)\(\s*(?:monkeypatch,\s*)?"([a-z0-9_]+)"', open(path).read()))
        used |= set(re.findall(r'routes\.override\(([a-z0-9_]+)=', open(path).read()))
    assert not (used - set(routes.NAMES) - set(routes.NATIVE)), f"route names used but not declared: {sorted(used - set(routes.NAMES) - set(routes.NATIVE))}"
    dead = set(routes.NAMES) - used
    assert not dead, f"routes nothing reads: {sorted(dead)}"


def test_the_package_reads_the_environment_in_routes_only():
    offenders = {}
    for base in ("torch-rgcn_amd", "bench.py", "__graft_entry__.py"):
        for path in _files(base, (".py",)):
            rel = os.path.relpath(path, ROOT)
            if rel.endswith("torch_rgcn/routes.py"):
                continue
            for m in ENV_READ.finditer(open(path, encoding="utf-8", errors="replace").read()):
                name = m.group(1) or m.group(2)
                if name not in PLAIN:
                    offenders.setdefault(rel, set()).add(name)
    assert not offenders, f"RGCN_* variables read outside torch_rgcn/routes.py: {offenders}"
    text = open(os.path.join(PKG, "torch_rgcn", "functional.py")).read()
    assert text.count("os.environ") == 0


def test_the_library_has_no_getenv():
    for path in _files(os.path.join("torch-rgcn_amd", "csrc"), (".hip", ".cpp", ".h")):
        code = re.sub(r"//[^\n]*", "", open(path, encoding="utf-8", errors="replace").read())
        assert "getenv" not in code, os.path.relpath(path, ROOT)


def test_semantic_switches_are_flagged():
    sec = _design_section7()
    a = sec.index("change semantics")
    para = sec[a:a + 1500]
    for k in SEMANTIC:
        assert k in para, f"{k} changes results or error behaviour and must be named in the 'change semantics' paragraph"


def test_shipped_library_has_no_wrong_result_switch():
    lib_path = os.path.join(PKG, "torch_rgcn", "lib", "librgcn_hip.so")
    if not os.path.isfile(lib_path):
        pytest.skip("library not built")
    L = ctypes.CDLL(lib_path)
    L.rgcn_last_error.restype = ctypes.c_char_p
    for name in (b"bwd_abl",):
        assert L.rgcn_set_option(name, ctypes.c_int32(2)) != 0, name
        assert L.rgcn_set_option(name, ctypes.c_int32(0)) == 0, name
    assert L.rgcn_set_option(b"gemm_bm", ctypes.c_int32(64)) == 0 and L.rgcn_set_option(b"gemm_bm", ctypes.c_int32(0)) == 0
    assert L.rgcn_set_option(b"no_such_option", ctypes.c_int32(1)) != 0
    syms = subprocess.run(["nm", "-D", "--defined-only", lib_path], capture_output=True, text=True).stdout
    assert "rgcn_blk_debug_read" not in syms and "abl" not in syms.lower().replace("available", ""), "ablation entry points in the shipped library"


def test_tile_kernel_mode_follows_the_largest_source_degree(monkeypatch):
    """featureless basis layer on tables beyond the caches (rgcn_fbasis_tile.hip): one wave per node on the matrix cores unless a source
    node is a hub (its messages would be ONE wave's loop) -- then a tile's messages are dealt over the waves; routes force either form or
    turn the tile kernels off; shapes outside the LDS budget fall back (host logic only: no GPU needed)"""
    from torch_rgcn import _native, routes
    am = dict(R=267, B=40, d=10, n_nodes=1_666_764)          # AM as shipped
    assert _native.fbasis_tile_ok(**am, max_degree=25) == (True, 1)
    assert _native.fbasis_tile_ok(**am, max_degree=_native.TILE_NODE_MODE_MAX_DEGREE + 1) == (True, 0)
    assert _native.fbasis_tile_ok(**am, max_degree=None) == (True, 0)
    routes.patch(monkeypatch, "fbasis_tile", "ranges")
    assert _native.fbasis_tile_ok(**am, max_degree=25) == (True, 0)
    routes.patch(monkeypatch, "fbasis_tile", "nodes")
    assert _native.fbasis_tile_ok(**am, max_degree=10 ** 6) == (True, 1)
    routes.patch(monkeypatch, "fbasis_tile", "0")
    assert _native.fbasis_tile_ok(**am, max_degree=25)[0] is False
    routes.patch(monkeypatch, "fbasis_tile", None)
    routes.patch(monkeypatch, "deterministic", "1")
    assert _native.fbasis_tile_ok(**am, max_degree=25)[0] is False
    routes.patch(monkeypatch, "deterministic", None)
    assert _native.fbasis_tile_ok(R=2000, B=64, d=16, n_nodes=10 ** 6, max_degree=5)[0] is False      # R x B doubles beyond the LDS
    assert _native.fbasis_tile_ok(R=13, B=5, d=10, n_nodes=8, max_degree=5)[0] is False               # fewer nodes than a tile
