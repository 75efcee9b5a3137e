#!/usr/bin/env python3
"""Source-locality pass: the forward gather kernel and the fused backward on the plans of a graph whose nodes were
relabelled for locality (torch_rgcn.graph.node_order: degree / rcm), against the original numbering.
  python tools/locality_bench.py --graph s1|zipf [--orders none,degree,rcm] [--only ORDER]   -> one JSON line per order
Graphs: s1 = the uniform S1 graph (N = 1 M, E = 10 M, R0 = 50); zipf = the AM-shaped graph with Zipf(0.9) subjects / objects /
relations of DESIGN.md section 5 (N = 1,666,764, E = 5,988,321, R0 = 133 -> sparse buckets: forward on the two-pass path)."""
import argparse
import json
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
ap.add_argument("--graph", default="s1")
ap.add_argument("--orders", default="none,degree,rcm")
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
d = 16
if a.graph == "s1":
    N, R0, E = 1_000_000, 50, 10_000_000
    T = _native.synthetic_triples_host(N, R0, E, 0)
else:
    N, R0, E = 1_666_764, 133, 5_988_321
    rng = np.random.default_rng(0)

    def zipf(n, size):
        w = 1.0 / np.arange(1, n + 1) ** 0.9
        return rng.choice(n, size=size, p=w / w.sum())
    T = np.stack([zipf(N, E), zipf(R0, E), rng.permutation(N)[zipf(N, E)]], axis=1).astype(np.int64)
R = 2 * R0 + 1
tp = _native.add_inverse_and_self_host(T, N, R0)
M = tp.shape[0]
X = torch.randn(N, d, device=dev)
G = torch.randn(N, d, device=dev)
W = torch.randn(R, d, d, device=dev) * 0.1
b = torch.zeros(d, device=dev)


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
    return float(np.median(ts))


ref = None
for how in a.orders.split(","):
    t0 = time.time()
    g = graph_from_nc_triples(tp, N, R, False, dev, relabel=how)
    t_order = time.time() - t0
    Xp, Gp = (X, G) if g.perm is None else (X.index_select(0, g.inv), G.index_select(0, g.inv))
    fp = g.fwd_plan(d)
    dense = fp.m_pad > 0 and fp.n_messages >= 0.5 * fp.m_pad
    res = {"graph": a.graph, "order": how, "N": N, "messages": M, "plan_and_order_s": round(t_order, 2), "tile_path": bool(dense)}
    if dense:
        out = _native.spmm(Xp, W, b, fp)
        res["spmm_ms"] = round(timeit(lambda: _native.spmm(Xp, W, b, fp), a.iters), 4)
        bp = g.bwd_blk_plan() or g.bwd_plan(d)
        if _native.bwd_fused_ok(bp):
            res["bwd_fused_ms"] = round(timeit(lambda: _native.bwd_fused(Gp, Xp, W, bp, atomic=True), a.iters), 4)
    else:
        sp, csr = g.scatter_plan("fwd"), g.csr("fwd")
        out = _native.spmm_two_pass(Xp, W, b, sp, csr)
        res["spmm_two_pass_ms"] = round(timeit(lambda: _native.spmm_two_pass(Xp, W, b, sp, csr), a.iters), 4)
        spb, csrb = g.scatter_plan("bwd"), g.csr("bwd")
        res["bwd_two_pass_fused_ms"] = round(timeit(lambda: _native.bwd_two_pass_fused(Gp, Xp, W, spb, csrb), a.iters), 4)
    out = out if g.perm is None else out.index_select(0, g.perm)
    if ref is None:
        ref = out
    res["max_rel_diff_vs_original_numbering"] = float(((out - ref).abs().max() / ref.abs().max()).item())
    print(json.dumps(res), flush=True)
    del g, fp
    torch.cuda.empty_cache()
