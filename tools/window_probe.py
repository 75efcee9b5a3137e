#!/usr/bin/env python3
"""VERDICT r5 #2, step B probe: the block-tile forward kernel (rgcn_spmm_blk_f32, tall workgroup-owned tiles, chunk records that carry
their relation) on a plan whose chunks are ordered WINDOW-MAJOR inside every tile: bucket key (tile, source window, relation) instead of
(tile, relation) -- built with the existing plan builder by handing it the pseudo relation  window(src) * R + p  and restoring the true
relation in chunk_rel afterwards.  All workgroups then walk the source windows in the same order (tools/micro/gather_window.hip: the
gather alone goes from 0.375 to ~0.2 ms).  Prints per (tile rows, windows): padding, chunks, ms per launch, and the error against spmm."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-rgcn_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from torch_rgcn import _native  # noqa: E402
from torch_rgcn._native import _check, _dp, _on, _stream, c_i32, c_i64, lib  # noqa: E402
from torch_rgcn.graph import graph_from_nc_triples  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=1_000_000)
ap.add_argument("--edges", type=int, default=10_000_000)
ap.add_argument("--rels", type=int, default=50)
ap.add_argument("--rows", default="977,489")
ap.add_argument("--windows", default="1,2,4,8,16")
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
N, R0, E = a.nodes, a.rels, a.edges
R = 2 * R0 + 1
T = _native.synthetic_triples_host(N, R0, E, 0)
tp = _native.add_inverse_and_self_host(T, N, R0)
g = graph_from_nc_triples(tp, N, R, False, dev)
M = tp.shape[0]
torch.manual_seed(0)
X = torch.randn(N, 16, device=dev)
W = torch.randn(R, 16, 16, device=dev) * 0.1
b = torch.randn(16, device=dev)
ref = _native.spmm(X, W, b, g.fwd_plan(16))
Wp = _native.pack_w16(W)


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


med, mn = timeit(lambda: _native.spmm(X, W, b, g.fwd_plan(16)), a.iters)
print(f"spmm_d16 (wave-owned 128-row tiles): med {med:.3f} min {mn:.3f} ms", flush=True)
s, p, o, val, alive = g._dev
for rows in [int(v) for v in a.rows.split(",")]:
    for nw in [int(v) for v in a.windows.split(",")]:
        wrows = -(-N // nw)
        rel = (torch.div(o, wrows, rounding_mode="floor") * R + p).to(torch.int32)
        plan = _native.build_plan_device(s, o, rel, val, alive, N, N, R * nw, rows, M, want_runs=True, want_pack=False)
        plan.chunk_rel.remainder_(R)
        rec = _native._blk_rec(plan)
        out = torch.empty(N, 16, device=dev)

        def run():
            with _on(dev):
                _check(lib().rgcn_spmm_blk_f32(_dp(X), _dp(Wp), _dp(b), _dp(out), _dp(rec), _dp(plan.run_ptr), c_i64(plan.n_tiles),
                                               c_i32(plan.tile_rows), c_i64(N), c_i32(R * nw), c_i32(0), None, c_i64(0), c_i64(0), _stream(dev)), "spmm_blk")
        med, mn = timeit(run, a.iters)
        err = float((out - ref).abs().max() / ref.abs().max())
        print(f"spmm_blk rows={rows} tiles={plan.n_tiles} windows={nw} ({wrows * 64 / 1e6:.1f} MB): pad {plan.m_pad / M:.3f} chunks {plan.n_chunks} "
              f"med {med:.3f} min {mn:.3f} ms  err {err:.1e}", flush=True)
        del plan, rec, rel
        torch.cuda.empty_cache()
