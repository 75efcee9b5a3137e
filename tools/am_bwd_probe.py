#!/usr/bin/env python3
"""AM-shaped block-diagonal layer (N = 1.67 M, R = 267, d = 16, nb = 4): per-kernel times of forward + backward on the default
routes and on round 2's backward (RGCN_BWD_KERNEL=lean), with SURVEY 8(d)'s byte model next to them."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-rgcn_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from torch_rgcn import _native  # noqa: E402
from torch_rgcn.layers import RelationalGraphConvolutionNC  # noqa: E402

dev = torch.device("cuda:0")
N, R0, E, d = 1_666_764, 133, 5_988_321, 16
T = _native.synthetic_triples_host(N, R0, E, 2)
tp = torch.from_numpy(_native.add_inverse_and_self_host(T, N, R0))
M = tp.shape[0]
res = {}
for tag, env in (("round 3 default (block CSR forward, block-tile backward DIAG4)", {}), ("round 2 backward (relation-major fused pass + row sums)", {"RGCN_BWD_KERNEL": "lean"})):
    for k, v in env.items():
        os.environ[k] = v
    torch.manual_seed(0)
    layer = RelationalGraphConvolutionNC(triples=tp, num_nodes=N, num_relations=2 * R0 + 1, in_features=d, out_features=d,
                                         decomposition={"type": "block", "num_blocks": 4}).to(dev)
    X = torch.randn(N, d, device=dev, requires_grad=True)
    for _ in range(3):
        out = layer(X); out.backward(out.detach())
    torch.cuda.synchronize()
    _native.profile_start()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        X.grad = None
        out = layer(X); out.backward(out.detach())
    b.record(); torch.cuda.synchronize()
    prof = _native.profile_stop()
    fb, bb = M * (4 * d + 8) + N * 4 * d, M * (4 * d + 8) + 2 * N * 4 * d
    kern = {k: round(float(np.mean(v)), 4) for k, v in prof.items()}
    bwd_ms = sum(v for k, v in kern.items() if k in ("bwd_fused", "bwd_scatter_dw", "segment_gather_sum", "segment_sum", "colsum"))
    res[tag] = {"ms_per_layer_fwd_bwd": round(a.elapsed_time(b) / 10, 4), "kernels_ms": kern, "backward_kernels_ms": round(bwd_ms, 4),
                "backward_frac_of_hbm_roof_on_8d_bytes": round(bb / (bwd_ms * 1e-3) / 8e12, 4),
                "forward_frac": round(fb / (kern.get("block_spmm", 1e9) * 1e-3) / 8e12, 4)}
    for k in env:
        del os.environ[k]
    del layer
print(json.dumps({"workload": f"AM-shaped N={N} R={2 * R0 + 1} M={M} d={d} block nb=4, ONE layer fwd+bwd", "results": res}, indent=1))
