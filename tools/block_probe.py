"""Block-diagonal layers: the block kernels (csrc/rgcn_block.hip) against the two earlier routes -- the R x N x d_out message
table and the expanded dense R x d x d weights (gather-GEMM above width 16, the hidden-16 kernels at width 16).
One JSON line per workload: forward + backward of one layer, median wall time, peak memory, per-kernel times.

  python tools/block_probe.py > profiles/r02_block_probe.jsonl
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "torch-rgcn_amd"))
from torch_rgcn import _native  # noqa: E402
from torch_rgcn.layers import RelationalGraphConvolutionLP, RelationalGraphConvolutionNC  # noqa: E402

dev = torch.device("cuda")
ROUTES = {"block_kernels": ("2", "1"), "dense": ("0", "0")}     # RGCN_BLOCK_PATH (the einsum message-table route was removed in round 3)


def timed(fn, iters=7, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


def run(tag, make, messages, extra):
    res, ref = {}, None
    for route, (bp, bt) in ROUTES.items():
        os.environ["RGCN_BLOCK_PATH"], os.environ["RGCN_BLOCK_TABLE"] = bp, bt
        try:
            torch.manual_seed(0)
            step, grads = make()
            ms = timed(step)
            torch.cuda.reset_peak_memory_stats()
            step()
            torch.cuda.synchronize()
            peak = torch.cuda.max_memory_allocated() / 1e9
            _native.profile_start()
            step()
            k = {n: round(sum(v), 4) for n, v in _native.profile_stop().items()}
            res[route] = {"ms": round(ms, 3), "peak_GB": round(peak, 2), "kernels_ms": k}
            g = [t.clone() for t in grads()]
            if ref is None:
                ref = g
            else:
                res[route]["rel_diff_vs_block_kernels"] = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
                                                              for a, b in zip(g, ref))
        except Exception as exc:  # noqa: BLE001
            res[route] = {"error": f"{type(exc).__name__}: {exc}"[:160]}
        torch.cuda.empty_cache()
    print(json.dumps({"workload": tag, "messages": messages, **extra, **res}), flush=True)


def lp_case(tag, N, R0, E, d, nb):
    T = torch.from_numpy(_native.synthetic_triples_host(N, R0, E, 3)).to(dev)

    def make():
        layer = RelationalGraphConvolutionLP(num_nodes=N, num_relations=2 * R0 + 1, in_features=d, out_features=d,
                                             edge_dropout={"general": 0.5, "self_loop": 0.2, "self_loop_type": "schlichtkrull-dropout"},
                                             decomposition={"type": "block", "num_blocks": nb}, b_init="zeros").to(dev).eval()
        X = torch.randn(N, d, device=dev, requires_grad=True)

        def step():
            X.grad = None
            for p in layer.parameters():
                p.grad = None
            layer(T, X).pow(2).mean().backward()
        return step, lambda: (X.grad, layer.blocks.grad, layer.blocks_self.grad)
    run(tag, make, 3 * E + N, {"N": N, "R": 2 * R0 + 1, "d": d, "nb": nb, "layer": "LP (graph built per call)"})


def nc_case(tag, N, R0, E, d, nb):
    tp = torch.from_numpy(_native.add_inverse_and_self_host(_native.synthetic_triples_host(N, R0, E, 2), N, R0))

    def make():
        layer = RelationalGraphConvolutionNC(triples=tp, num_nodes=N, num_relations=2 * R0 + 1, in_features=d, out_features=d,
                                             decomposition={"type": "block", "num_blocks": nb}).to(dev)
        X = torch.randn(N, d, device=dev, requires_grad=True)

        def step():
            X.grad = None
            for p in layer.parameters():
                p.grad = None
            layer(X).pow(2).mean().backward()
        return step, lambda: (X.grad, layer.blocks.grad)
    run(tag, make, 2 * E + N, {"N": N, "R": 2 * R0 + 1, "d": d, "nb": nb, "layer": "NC (static graph)"})


if __name__ == "__main__":
    nc_case("AM-shaped, d=16, 4 blocks of 4x4 (BASELINE config 2)", 1_666_764, 133, 5_988_321, 16, 4)
    nc_case("S1-shaped, d=16, 4 blocks of 4x4", 1_000_000, 50, 10_000_000, 16, 4)
    lp_case("FB15k-237-shaped LP layer, d=500, 100 blocks of 5x5", 14_545, 237, 30_000, 500, 100)
    lp_case("FB-toy-shaped LP layer, d=500, 100 blocks of 5x5", 280, 112, 300, 500, 100)
    nc_case("AM-shaped, d=64, 8 blocks of 8x8", 1_666_764, 133, 5_988_321, 64, 8)
