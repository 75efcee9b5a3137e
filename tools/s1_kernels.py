#!/usr/bin/env python3
"""The S1 launches of the shipped route, alone (profiling target of tools/prof.sh pmc / trace): the hidden-16 forward and the fused backward
as torch_rgcn.functional picks them (round 6: rgcn_spmm_blk_f32 on the soft-window plan; backward per RGCN_BWD_OWN)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-rgcn_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from torch_rgcn import _native, functional  # noqa: E402
from torch_rgcn.graph import graph_from_nc_triples  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=1_000_000)
ap.add_argument("--edges", type=int, default=10_000_000)
ap.add_argument("--rels", type=int, default=50)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--what", default="fwd,bwd")
a = ap.parse_args()
dev = torch.device("cuda:0")
N, R0, E = a.nodes, a.rels, a.edges
R = 2 * R0 + 1
tp = _native.add_inverse_and_self_host(_native.synthetic_triples_host(N, R0, E, 0), N, R0)
g = graph_from_nc_triples(tp, N, R, False, dev)
torch.manual_seed(0)
X = torch.randn(N, 16, device=dev)
G = torch.randn(N, 16, device=dev)
W = torch.randn(R, 16, 16, device=dev) * 0.1
b = torch.zeros(16, device=dev)


def timeit(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return float(np.median(ts))


if "fwd" in a.what:
    plan = functional._fwd_win(g, True)
    fn = (lambda: _native.spmm_blk(X, W, b, plan, relu=True)) if plan is not None else (lambda: _native.spmm(X, W, b, g.fwd_plan(16), relu=True))
    print("forward", "spmm_blk on the soft-window plan" if plan is not None else "spmm_d16", f"{timeit(fn):.3f} ms", flush=True)
if "bwd" in a.what:
    print("backward", f"{timeit(lambda: functional._fused_backward(X, W, G, g, relu_in=True, want_db=True)):.3f} ms", flush=True)
