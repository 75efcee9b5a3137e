#!/usr/bin/env python3
"""rgcn_bwd_own_f32 (relation-owner backward on the soft-window plan) against rgcn_bwd_blk_f32 on the S1 graph: results and time"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-rgcn_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from torch_rgcn import _native, routes  # noqa: E402
from torch_rgcn.graph import graph_from_nc_triples  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=1_000_000)
ap.add_argument("--edges", type=int, default=10_000_000)
ap.add_argument("--rels", type=int, default=50)
ap.add_argument("--rows", default="")
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
N, R0, E = a.nodes, a.rels, a.edges
R = 2 * R0 + 1
T = _native.synthetic_triples_host(N, R0, E, 0)
tp = _native.add_inverse_and_self_host(T, N, R0)
g = graph_from_nc_triples(tp, N, R, False, dev)
torch.manual_seed(0)
X = torch.randn(N, 16, device=dev)
G = torch.randn(N, 16, device=dev)
W = torch.randn(R, 16, 16, device=dev) * 0.1


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


bp = g.bwd_blk_plan()
dX0, dW0, db0 = _native.bwd_fused(G, X, W, bp, atomic=True, relu=True, want_db=True)
med, mn = timeit(lambda: _native.bwd_fused(G, X, W, bp, atomic=True, relu=True, want_db=True), a.iters)
print(f"bwd_blk rows={bp.tile_rows}: med {med:.3f} min {mn:.3f} ms", flush=True)
for rows in [int(v) for v in a.rows.split(",") if v] or [None]:
    op = g.win_plan("bwd_own", rows)
    if op is None:
        print("no owner plan for rows", rows)
        continue
    dX, dW, db = _native.bwd_own(G, X, W, op, relu=True, want_db=True)
    torch.cuda.synchronize()
    e = [float((x - y).abs().max() / y.abs().max()) for x, y in ((dX, dX0), (dW, dW0), (db, db0))]
    med, mn = timeit(lambda: _native.bwd_own(G, X, W, op, relu=True, want_db=True), a.iters)
    print(f"bwd_own rows={op.tile_rows} tiles={op.n_tiles} pad {op.m_pad / op.n_messages:.3f} balance {op.own_balance:.3f}: med {med:.3f} min {mn:.3f} ms  "
          f"err dX {e[0]:.1e} dW {e[1]:.1e} db {e[2]:.1e}", flush=True)
