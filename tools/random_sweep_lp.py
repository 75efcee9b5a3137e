"""Randomised parity sweep of the link-prediction layer (graph built per call) against the oracle: python tools/random_sweep_lp.py SEED [CASES]
Block counts 1..25 with blocks 1 x 1 .. 9 x 9 (the block kernels take <= 8 x 8), basis 1..4, hubs, self-loop dropout in
training mode (mask replayed from the same torch seed), widths 1..200."""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "torch-rgcn_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import oracle  # noqa: E402
from torch_rgcn.layers import RelationalGraphConvolutionLP  # noqa: E402

DEV, TOL = "cuda:0", 1e-4


def rel_err(a, b):
    a = a.detach().cpu().numpy().astype(np.float64)
    den = np.abs(b).max()
    return float(np.abs(a - b).max() / den) if den > 0 else float(np.abs(a).max())


seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(seed)
fails = 0
for case in range(cases):
    N = int(rng.choice([1, 2, 5, 30, 64, 65, 129, 500, 1500]))
    R0 = int(rng.integers(1, 7))
    E = min(int(rng.choice([0, 1, 2, 17, 60, 900, 4000, 12000])), 200 * N)     # (thousands of parallel edges between 1-2 nodes: fp32 sums of
    #                                                                              12,000 cancelling terms differ from the fp64 oracle by > 1e-4)
    mode = str(rng.choice(["none", "basis", "block"]))
    vertical = bool(rng.random() < 0.4) and mode != "block"
    nb = 2
    if mode == "block":
        nb = int(rng.choice([1, 2, 3, 4, 5, 12, 25]))
        d_in, d_out = nb * int(rng.integers(1, 10)), nb * int(rng.integers(1, 10))
    else:
        d_in, d_out = (int(rng.choice([1, 2, 8, 16, 17, 20, 40, 66, 128, 200])) for _ in range(2))
    training = bool(rng.random() < 0.5)
    p_self = float(rng.choice([0.0, 0.3, 0.7]))
    sl_type = str(rng.choice(["other", "schlichtkrull-dropout"])) if mode != "block" else "other"
    R = 2 * R0 + 1
    T = oracle.synthetic_triples(N, R0, E, seed=9100 + case) if E else np.zeros((0, 3), np.int64)
    if E > 50 and N > 2 and rng.random() < 0.3:
        T[: E // 4, 0] = 1                                              # hub
    decomp = {"none": None, "basis": {"type": "basis", "num_bases": int(rng.integers(1, 5))},
              "block": {"type": "block", "num_blocks": nb}}[mode]
    tag = f"case {case}: N={N} R0={R0} E={E} mode={mode} nb={nb} vertical={vertical} d=({d_in},{d_out}) training={training} p={p_self} {sl_type}"
    if os.environ.get("SWEEP_VERBOSE"):        # (a case that takes the process down is the last one printed)
        print(tag, flush=True)
    try:
        layer = RelationalGraphConvolutionLP(num_nodes=N, num_relations=R, in_features=d_in, out_features=d_out,
                                             edge_dropout={"general": 0.5, "self_loop": p_self, "self_loop_type": sl_type},
                                             decomposition=decomp, vertical_stacking=vertical, b_init="zeros").to(DEV)
        with torch.no_grad():
            for prm in layer.parameters():
                prm.copy_(torch.from_numpy(rng.standard_normal(tuple(prm.shape)).astype(np.float32) * 0.3))
        layer.train(training)
        X = torch.from_numpy(rng.standard_normal((N, d_in)).astype(np.float32)).to(DEV).requires_grad_(True)
        torch.manual_seed(9000 + case)
        out = layer(torch.from_numpy(T), X)
        torch.manual_seed(9000 + case)
        keep = (1.0 - p_self) if (training and sl_type == "other") else 1.0
        mask = torch.bernoulli(torch.full((N,), keep, dtype=torch.float, device=DEV)).to(torch.bool).cpu().numpy()
        g = rng.standard_normal(tuple(out.shape)).astype(np.float32)
        out.backward(torch.from_numpy(g).to(DEV))
        params = {n: prm.detach().cpu().numpy() for n, prm in layer.named_parameters() if n != "bias"}
        ref = oracle.lp_layer(T, N, R, X.detach().cpu().numpy(), params, mode, layer.bias.detach().cpu().numpy(), vertical, mask, g)
        errs = {"out": rel_err(out, ref["out"]), "dX": rel_err(X.grad, ref["dX"])}
        errs.update({n: rel_err(getattr(layer, n).grad, gv) for n, gv in ref["grads"].items()})
        bad = {k: v for k, v in errs.items() if not v < TOL}
        if bad:
            fails += 1
            print("FAIL", tag, bad, flush=True)
    except Exception as exc:  # noqa: BLE001
        fails += 1
        print("FAIL", tag, f"{type(exc).__name__}: {str(exc)[:160]}", flush=True)
print("done, cases:", cases, "failures:", fails)
