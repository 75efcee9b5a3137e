"""Diagonal-weight layer (e-rgcn, reference models.py:272-280): the diagonal kernels against the embedded R x d x d route,
AM-shaped and AIFB-shaped graphs at the embedding sizes of configs/e-rgcn/*.yaml.  One JSON line per case.

  python tools/diag_bench.py > profiles/r02_diag_bench.jsonl
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-rgcn_amd"))
from torch_rgcn import _native  # noqa: E402
from torch_rgcn.layers import RelationalGraphConvolutionNC  # noqa: E402

DEV = torch.device("cuda:0")


def timed(fn, iters=10, warm=3):
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


def case(tag, N, R0, E, d):
    T = _native.synthetic_triples_host(N, R0, E, 5)
    tp = torch.from_numpy(_native.add_inverse_and_self_host(T, N, R0))
    out = {"workload": tag, "N": N, "R0": R0, "E": E, "d": d, "step": "diag layer forward + backward (dX, dw)"}
    ref = None
    for path in ("diag_kernels", "embedded"):
        os.environ["RGCN_DIAG_PATH"] = "1" if path == "diag_kernels" else "0"
        torch.manual_seed(0)
        layer = RelationalGraphConvolutionNC(triples=tp, num_nodes=N, num_relations=2 * R0 + 1, in_features=d, out_features=d,
                                             diag_weight_matrix=True).to(DEV)
        X = torch.randn(N, d, device=DEV, requires_grad=True)

        def step():
            X.grad = None
            layer.weights.grad = None
            layer(X).pow(2).mean().backward()
        try:
            ms = timed(step)
            torch.cuda.reset_peak_memory_stats()
            step()
            _native.profile_start()
            step()
            k = {n: round(sum(v), 4) for n, v in _native.profile_stop().items()}
            out[path] = {"ms_per_step": round(ms, 3), "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2), "kernels_ms": k}
            g = (X.grad.clone(), layer.weights.grad.clone())
            if ref is None:
                ref = g
            else:
                out["rel_diff_dX"] = float((g[0] - ref[0]).abs().max() / ref[0].abs().max())
                out["rel_diff_dw"] = float((g[1] - ref[1]).abs().max() / ref[1].abs().max())
        except Exception as exc:  # noqa: BLE001
            out[path] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
        del layer, X
        torch.cuda.empty_cache()
    M = 2 * E + N
    out["algorithmic_bytes_fwd"] = M * (4 * d + 8) + N * 4 * d
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    case("AIFB-shaped e-rgcn layer 1 (emb 32)", 8285, 45, 29_043, 32)
    case("AM-shaped e-rgcn layer 1 (emb 32)", 1_666_764, 133, 5_988_321, 32)
    case("AM-shaped, emb 128", 1_666_764, 133, 5_988_321, 128)
    case("S1-shaped, emb 16", 1_000_000, 50, 10_000_000, 16)
