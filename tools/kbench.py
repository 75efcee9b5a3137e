#!/usr/bin/env python3
"""Kernel micro-benchmark on the S1 graph: times rgcn_spmm_f32 / rgcn_wgrad_f32 launches alone
(HIP events on the launch stream).  Tuning knobs come from the environment (RGCN_SPMM_U,
RGCN_TILE_ROWS, RGCN_WGRAD_ITEM_CHUNKS).  Usage: python tools/kbench.py [--nodes N --edges E]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-rgcn_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from torch_rgcn import _native  # noqa: E402
from torch_rgcn.graph import graph_from_nc_triples  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=1_000_000)
ap.add_argument("--edges", type=int, default=10_000_000)
ap.add_argument("--rels", type=int, default=50)
ap.add_argument("--d", type=int, default=16)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--what", default="spmm,wgrad")
ap.add_argument("--skew", type=float, default=0.0, help="Zipf exponent for subjects/objects/relations (0 = uniform S1 graph)")
ap.add_argument("--local", type=int, default=0, help="objects within +-LOCAL rows of their subject (0 = uniform S1 graph): shows "
                "what the same kernels do when the graph has node locality")
a = ap.parse_args()
dev = torch.device("cuda:0")
N, R0, E, d = a.nodes, a.rels, a.edges, a.d
R = 2 * R0 + 1
t0 = time.time()
if a.skew > 0:
    rng = np.random.default_rng(0)
    def zipf(n, size):   # heavy-tailed ids: a few hubs / dominant relations
        w = 1.0 / np.arange(1, n + 1) ** a.skew
        return rng.choice(n, size=size, p=w / w.sum())
    T = np.stack([zipf(N, E), zipf(R0, E), rng.permutation(N)[zipf(N, E)]], axis=1).astype(np.int64)
    print("skewed graph: max in-degree", np.bincount(T[:, 0]).max(), "largest relation", np.bincount(T[:, 1]).max(), flush=True)
elif a.local > 0:
    T = _native.synthetic_triples_host(N, R0, E, 0)
    rng = np.random.default_rng(0)
    T[:, 2] = (T[:, 0] + rng.integers(-a.local, a.local + 1, size=E)) % N
    print(f"local graph: |subject - object| <= {a.local}", flush=True)
else:
    T = _native.synthetic_triples_host(N, R0, E, 0)
tp = _native.add_inverse_and_self_host(T, N, R0)
g = graph_from_nc_triples(tp, N, R, False, dev)
X = torch.randn(N, d, device=dev)
G = torch.randn(N, d, device=dev)
W = torch.randn(R, d, d, device=dev) * 0.1
b = torch.zeros(d, device=dev)
M = tp.shape[0]
alg = M * (4 * d + 8) + N * 4 * d


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return float(np.median(ts)), float(np.min(ts))


tag = f"U={os.environ.get('RGCN_SPMM_U', '-')} T={os.environ.get('RGCN_TILE_ROWS', '-')} IC={os.environ.get('RGCN_WGRAD_ITEM_CHUNKS', '-')}"
if "spmm" in a.what:
    plan = g.fwd_plan(d)
    med, mn = timeit(lambda: _native.spmm(X, W, b, plan), a.iters)
    print(f"[{tag}] spmm  tile={plan.tile_rows} pad={plan.m_pad / M:.3f} med {med:.3f} ms min {mn:.3f} ms -> {alg / med / 1e6:.0f} GB/s algorithmic", flush=True)
if "wgrad" in a.what:
    wp = g.wgt_plan()
    med, mn = timeit(lambda: _native.wgrad(X, G, wp, R), a.iters)
    print(f"[{tag}] wgrad items={wp.n_items} pad={wp.m_pad / M:.3f} med {med:.3f} ms min {mn:.3f} ms", flush=True)
if "wgradtile" in a.what:
    plan = g.fwd_plan(d)
    med, mn = timeit(lambda: _native.wgrad(X, G, plan, R), a.iters)
    print(f"[{tag}] wgrad(tile-major plan) items={plan.n_items} med {med:.3f} ms min {mn:.3f} ms", flush=True)
if "twopass" in a.what:
    sp, csr = g.scatter_plan("fwd"), g.csr("fwd")
    ref = _native.spmm(X, W, b, g.fwd_plan(d))
    got = _native.spmm_two_pass(X, W, b, sp, csr)
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    _native.profile_start()
    med, mn = timeit(lambda: _native.spmm_two_pass(X, W, b, sp, csr), a.iters)
    prof = _native.profile_stop()
    print(f"[{tag}] spmm_two_pass relerr={err:.2e} items={sp.n_items} med {med:.3f} ms min {mn:.3f} ms; "
          + " ".join(f"{k} {np.median(v):.3f}" for k, v in prof.items()), flush=True)
if "wtiled" in a.what:
    plan = g.fwd_plan(d)
    tpi = int(os.environ.get("RGCN_WGRAD_TILES", "4"))
    ref = _native.wgrad(X, G, g.wgt_plan(), R)
    got = _native.wgrad_tiled(X, G, plan, R, tpi)
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    med, mn = timeit(lambda: _native.wgrad_tiled(X, G, plan, R, tpi), a.iters)
    print(f"[{tag}] wgrad_tiled tiles/item={tpi} relerr_vs_relmajor={err:.2e} med {med:.3f} ms min {mn:.3f} ms", flush=True)
if "bwd" in a.what:
    bp, fp = g.bwd_blk_plan() or g.bwd_plan(d), g.fwd_plan(d)
    Wt = W.transpose(1, 2).contiguous()
    bp64 = g._plan("bwd", min(64, bp.tile_rows))
    ref_dx = _native.spmm(G, Wt, None, bp64)
    ref_dw = _native.wgrad_tiled(X, G, fp, R, 8)
    bp_tall = bp
    for atomic in (False, True):
        bp = bp_tall if (atomic or not _native._bwd_blk_plan(bp_tall)) else bp64    # the block-tile kernel has no deterministic form
        dx, dw = _native.bwd_fused(G, X, W, bp, atomic=atomic)
        e1 = ((dx - ref_dx).abs().max() / ref_dx.abs().max()).item()
        e2 = ((dw - ref_dw).abs().max() / ref_dw.abs().max()).item()
        dbg = getattr(_native.lib(), "rgcn_blk_debug_read", None) if atomic and os.environ.get("RGCN_HIP_LIB") else None
        if dbg is not None:
            import ctypes
            buf = (ctypes.c_uint64 * 4)()
            dbg(buf, 1)
        _native.profile_start()
        med, mn = timeit(lambda: _native.bwd_fused(G, X, W, bp, atomic=atomic), a.iters)
        _native.profile_stop()
        if dbg is not None:
            dbg(buf, 1)
            w = max(buf[3], 1)
            print(f"   per wave (100 MHz ticks -> us): barrier wait {buf[0] / w / 100:.1f} us, epilogue {buf[1] / w / 100:.1f} us, kernel {buf[2] / w / 100:.1f} us "
                  f"({buf[3] // (a.iters + 3)} waves per launch)", flush=True)
        balg = M * (4 * d + 8) + 2 * N * 4 * d
        print(f"[{tag} K={os.environ.get('RGCN_BWD_KERNEL', 'blk')} NW={os.environ.get('RGCN_BWD_NW', '-')} BP={os.environ.get('RGCN_BWD_BPERM', '-')}] bwd_fused {'atomic' if atomic else 'partial'} tile={bp.tile_rows} relerr dX {e1:.2e} dW {e2:.2e} "
              f"med {med:.3f} ms min {mn:.3f} ms -> {balg / med / 1e6:.0f} GB/s algorithmic (backward bytes)", flush=True)
    if _native.bwd_fused_relu_ok(bp):        # ReLU mask fused into the dX epilogue (window kernel)
        Xr = torch.relu(X)
        dxr, dwr = _native.bwd_fused(G, Xr, W, bp, atomic=True, relu=True)
        dx0, dw0 = _native.bwd_fused(G, Xr, W, bp, atomic=True)
        print("relu-masked dX == mask(dX):", bool(torch.allclose(dxr, dx0 * (Xr > 0), rtol=1e-5, atol=1e-6 * dx0.abs().max().item())), "dW relerr", ((dwr - dw0).abs().max() / dw0.abs().max()).item(), flush=True)
    d2 = _native.bwd_fused(G, X, W, bp64)
    d3 = _native.bwd_fused(G, X, W, bp64)
    print("bwd_fused (partial) bitwise reproducible:", bool(torch.equal(d2[0], d3[0]) and torch.equal(d2[1], d3[1])), flush=True)
print(f"setup {time.time() - t0:.1f}s", flush=True)
