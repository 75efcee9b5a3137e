"""rgcn_gemm_f32 against torch.mm (rocBLAS) on the dense products of the LP block layer's self-loop relation
(X @ blocks_self, its two gradients) and a few neighbours.  One JSON line per shape."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "torch-rgcn_amd"))
from torch_rgcn import _native  # noqa: E402
from torch_rgcn.functional import _split_k  # noqa: E402

dev = torch.device("cuda")


def timed(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


for tag, M, N, K, ta, tb in (("X @ W", 14545, 500, 500, False, False), ("g @ W^T", 14545, 500, 500, False, True),
                             ("X^T @ g", 500, 500, 14545, True, False), ("X @ W (WN18 N)", 40943, 200, 200, False, False),
                             ("X^T @ g (WN18 N)", 200, 200, 40943, True, False), ("square 4096", 4096, 4096, 4096, False, False),
                             # the WN18 step's three products: ag @ flat(bases), g @ flat^T, ag^T @ g
                             ("WN18 ag @ flat", 40943, 200, 400, False, False), ("WN18 g @ flat^T", 40943, 400, 200, False, True),
                             ("WN18 ag^T @ g", 400, 200, 40943, True, False)):
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    sk = _split_k(K, M, N) if ta else 1
    ms_hand = timed(lambda: _native.gemm(A, B, trans_a=ta, trans_b=tb, split_k=sk))
    ms_lib = timed(lambda: torch.mm(A.t() if ta else A, B.t() if tb else B))
    ref = torch.mm(A.t() if ta else A, B.t() if tb else B)
    out = _native.gemm(A, B, trans_a=ta, trans_b=tb, split_k=sk)
    fl = 2.0 * M * N * K
    print(json.dumps({"product": tag, "M": M, "N": N, "K": K, "split_k": sk, "hand_ms": round(ms_hand, 4), "rocblas_ms": round(ms_lib, 4),
                      "hand_TFLOPs": round(fl / ms_hand / 1e9, 1), "rocblas_TFLOPs": round(fl / ms_lib / 1e9, 1),
                      "rel_diff": float((out - ref).abs().max() / ref.abs().max())}), flush=True)
