#!/usr/bin/env python3
"""Featureless basis layer at AM scale (N = 1,666,764, R = 267, M = 13.6 M messages, B = 40, d = 10): kernel times of the
source-major path (csrc/rgcn_basis.hip) and of the destination-major fallback.   python tools/fbasis_bench.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "torch-rgcn_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from torch_rgcn import _native  # noqa: E402
from torch_rgcn.layers import RelationalGraphConvolutionNC  # noqa: E402

N, R0, E, B, d = 1_666_764, 133, 5_988_321, 40, 10
if len(sys.argv) > 1:
    N, R0, E, B, d = (int(x) for x in sys.argv[1:6])
dev = torch.device("cuda:0")
T = _native.synthetic_triples_host(N, R0, E, 2)
tp = torch.from_numpy(_native.add_inverse_and_self_host(T, N, R0))
res = {"N": N, "R": 2 * R0 + 1, "M": int(tp.shape[0]), "B": B, "d": d}
for path in ("src", "csr"):
    os.environ["RGCN_FBASIS"] = path
    layer = RelationalGraphConvolutionNC(triples=tp, num_nodes=N, num_relations=2 * R0 + 1, in_features=None, out_features=d,
                                         decomposition={"type": "basis", "num_bases": B}).to(dev)
    for it in range(6):
        if it == 2:
            torch.cuda.synchronize()
            _native.profile_start()
        for p in layer.parameters():
            p.grad = None
        layer().pow(2).mean().backward()
    torch.cuda.synchronize()
    prof = _native.profile_stop()
    res[path] = {k: round(float(np.mean(v)), 3) for k, v in prof.items()}
    del layer
print(json.dumps(res))
