#!/usr/bin/env python3
"""The dense step of the basis path at WN18 size (N = 40,943, d = 200, B = 2): kernel times (HIP events) and achieved
TFLOP/s of the fused aggregate-in-LDS + MFMA forward and of the two backward GEMMs (d_ag = g flat^T, dbases = ag^T g).
Run under `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` for the matrix-core utilisation (tools/prof.sh pmc)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-rgcn_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from torch_rgcn import _native  # noqa: E402
from torch_rgcn.graph import graph_from_lp_triples  # noqa: E402

dev = torch.device("cuda:0")
N, R0, d, B = 40_943, 18, 200, 2
R = 2 * R0 + 1
PEAK = 157.3   # TFLOP/s, fp32 MFMA (MI355X_MICROARCH.md)
res = {}
for tag, E in (("train graph (15,000 triples)", 15_000), ("eval graph (141,442 triples)", 141_442)):
    T = torch.from_numpy(_native.synthetic_triples_host(N, R0, E, 3)).to(dev)
    g = graph_from_lp_triples(T, N, R, False, None, dev)
    X, G = torch.randn(N, d, device=dev), torch.randn(N, d, device=dev)
    comps, bases, bias = torch.randn(R, B, device=dev), torch.randn(B, d, d, device=dev) * 0.1, torch.zeros(d, device=dev)
    csr = g.csr("fwd")
    flat = bases.view(B * d, d)
    ag = _native.basis_aggregate(X, comps, csr, B, d, 1)

    def t(fn, iters=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(iters):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return float(np.median(ts))
    flops = 2.0 * N * (B * d) * d
    M = int(csr.rowptr[-1].item())
    r = {"messages": M}
    for name, fn, fl in (("basis_aggregate (to HBM) + gemm NN", lambda: _native.gemm(_native.basis_aggregate(X, comps, csr, B, d, 1), flat, bias=bias), flops),
                         ("gemm NN alone: ag @ flat", lambda: _native.gemm(ag, flat, bias=bias), flops),
                         ("torch (rocBLAS) addmm: ag @ flat", lambda: torch.addmm(bias, ag, flat), flops),
                         ("gemm NT: d_ag = g @ flat^T", lambda: _native.gemm(G, flat, trans_b=True), flops),
                         ("gemm TN split-K 64: dbases = ag^T @ g", lambda: _native.gemm(ag, G, trans_a=True, split_k=64), flops)):
        ms = t(fn)
        r[name] = {"ms": round(ms, 4), "TFLOPs": round(fl / ms / 1e9, 1), "frac_of_fp32_mfma_peak": round(fl / ms / 1e9 / PEAK, 3)}
    res[tag] = r
print(json.dumps({"workload": f"WN18-shaped basis layer N={N} d={d} B={B} R={R}", "dense_flops_per_product": 2.0 * N * B * d * d, "results": res}, indent=1))
