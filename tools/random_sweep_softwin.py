"""Randomised parity sweep of round 6's routes (soft-window forward, relation-owner backward) against the oracle: large static graphs with
hidden-16 layers -- sizes around the routes' thresholds (32,768 nodes; 4 chunks per (tile, relation) bucket; 108 relations; owner balance 1.25),
hubs, skewed relation sizes, ragged last tiles, both stackings, basis / block weights on top.    python tools/random_sweep_softwin.py SEED [CASES]"""
import sys, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "torch-rgcn_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import test_gpu_parity as T
from oracle import oracle
from torch_rgcn import _native
from torch_rgcn.layers import RelationalGraphConvolutionNC

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
fails, took = 0, {"win_fwd": 0, "own_bwd": 0}
for case in range(cases):
    N = int(rng.choice([32_768, 32_769, 40_001, 65_536, 100_003, 200_000, 262_145]))
    R0 = int(rng.choice([1, 2, 5, 20, 50, 53, 54, 60]))
    R = 2 * R0 + 1
    per_bucket = float(rng.choice([20, 64, 70, 150, 400]))                 # messages per (256-tile, relation) bucket, roughly
    E = int(min(2_000_000, max(1000, per_bucket * 256 * R / 2)))
    mode = str(rng.choice(["none", "none", "none", "basis", "block"]))
    vertical = bool(rng.random() < 0.5)
    T_ = oracle.synthetic_triples(N, R0, E, seed=9000 + case)
    kind = str(rng.choice(["uniform", "uniform", "hub", "skewrel", "local"]))
    if kind == "hub":
        T_[: E // 6, 0] = int(rng.integers(0, N))
    elif kind == "skewrel":
        T_[: E // 2, 1] = 0                                              # half of the triples in one relation: the owner waves are unbalanced
    elif kind == "local":
        T_[:, 2] = (T_[:, 0] + rng.integers(-500, 501, size=E)) % N
    tp = oracle.add_inverse_and_self(T_, N, R0)
    decomp = {"none": None, "basis": {"type": "basis", "num_bases": int(rng.integers(1, 6))}, "block": {"type": "block", "num_blocks": int(rng.choice([2, 4]))}}[mode]
    try:
        layer = RelationalGraphConvolutionNC(triples=torch.from_numpy(tp), num_nodes=N, num_relations=R, in_features=16, out_features=16,
                                             decomposition=decomp, vertical_stacking=vertical).to(T.DEV)
        with torch.no_grad():
            for p in layer.parameters():
                p.copy_(torch.from_numpy(rng.standard_normal(tuple(p.shape)).astype(np.float32) * 0.3))
        Xh = rng.standard_normal((N, 16)).astype(np.float32)
        gh = rng.standard_normal((N, 16)).astype(np.float32)
        X = torch.from_numpy(Xh).to(T.DEV).requires_grad_(True)
        _native.profile_start()
        out = layer(X)
        out.backward(torch.from_numpy(gh).to(T.DEV))
        prof = _native.profile_stop()
        g = layer._graph
        win = g._plans.get(("win", "fwd", _native.spmm_blk_rows(N)))
        own = g._plans.get(("win", "bwd_own", _native.bwd_own_rows(N)))
        took["win_fwd"] += win is not None and "spmm_blk" in prof
        took["own_bwd"] += own is not None
        params = {n: p.detach().cpu().numpy() for n, p in layer.named_parameters() if n != "bias"}
        ref = oracle.nc_layer(tp, N, R, Xh, params, mode, layer.bias.detach().cpu().numpy(), vertical, gh)
        errs = {"out": T.rel_err(out, ref["out"]), "dX": T.rel_err(X.grad, ref["dX"]), "db": T.rel_err(layer.bias.grad, ref["db"])}
        for n, gv in ref["grads"].items():
            errs[n] = T.rel_err(getattr(layer, n).grad, gv)
        bad = {k: v for k, v in errs.items() if not v < T.TOL}
        if bad:
            raise AssertionError(f"errors {bad}")
        print(f"ok case {case}: N={N} R={R} E={E} {mode} {kind} vert={vertical} win={win is not None} own={own is not None} max err {max(errs.values()):.1e}", flush=True)
    except Exception as exc:
        fails += 1
        print(f"FAIL case {case}: N={N} R0={R0} E={E} mode={mode} kind={kind} vert={vertical}: {type(exc).__name__}: {str(exc)[:300]}", flush=True)
    torch.cuda.empty_cache()
print("done, failures:", fails, "routes taken:", took)
