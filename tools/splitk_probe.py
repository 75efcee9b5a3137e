#!/usr/bin/env python3
"""dbases = ag^T g of the WN18-shaped basis layer (M = 400, N = 200, K = 40,943): rgcn_gemm_f32 with the K dimension cut into split_k slices
(partial products to scratch + one fixed-order sum): time per split -- what functional._split_k should pick"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-rgcn_amd"))
from torch_rgcn import _native  # noqa: E402

dev = torch.device("cuda")
for K, M, N in ((40_943, 400, 200), (14_541, 200, 100), (40_943, 200, 200)):
    A = torch.randn(K, M, device=dev)
    G = torch.randn(K, N, device=dev)
    ref = (A.double().t() @ G.double())
    row = {}
    for s in (8, 16, 24, 32, 48, 64, 79, 128):
        for _ in range(3):
            out = _native.gemm(A, G, trans_a=True, split_k=s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            out = _native.gemm(A, G, trans_a=True, split_k=s)
        e1.record()
        torch.cuda.synchronize()
        err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
        row[s] = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
        assert err < 1e-5, (s, err)
    print(f"K={K} M={M} N={N}: us per product by split_k {row}", flush=True)
