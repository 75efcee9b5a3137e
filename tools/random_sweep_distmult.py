"""Randomised parity sweep of DistMult (scores, the three backward routes, biases, 2-D and 3-D triple tensors, hubs, unscored entities,
0 / 1 / many triples, widths 1..300) against the oracle: python tools/random_sweep_distmult.py SEED [CASES]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "torch-rgcn_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import oracle  # noqa: E402
from torch_rgcn import routes  # noqa: E402
from torch_rgcn.layers import DistMult  # noqa: E402

DEV, TOL = "cuda:0", 1e-4


def rel_err(a, b):
    a = a.detach().cpu().numpy().astype(np.float64)
    den = np.abs(b).max() if b.size else 0.0
    return float(np.abs(a - b).max() / den) if den > 0 else (float(np.abs(a).max()) if a.size else 0.0)


seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(seed)
fails = 0
for case in range(cases):
    N = int(rng.choice([1, 2, 7, 64, 65, 300, 2000, 40943]))
    R = int(rng.integers(1, 20))
    Tn = int(rng.choice([0, 1, 2, 63, 64, 65, 700, 5000, 60000]))
    d = int(rng.choice([1, 2, 3, 4, 7, 8, 16, 31, 32, 50, 52, 64, 100, 128, 200, 300]))
    bias = bool(rng.random() < 0.5)
    three_d = bool(rng.random() < 0.3) and Tn >= 2 and Tn % 2 == 0
    route = str(rng.choice(["auto", "auto", "split", "atomic"]))
    tr = np.stack([rng.integers(0, N, Tn), rng.integers(0, R, Tn), rng.integers(0, N, Tn)], axis=1).astype(np.int64)
    if Tn > 50 and N > 2 and rng.random() < 0.4:
        tr[: Tn // 3, 0] = 1                                            # hub subject
        tr[Tn // 3: Tn // 2, 2] = 1                                     # ... and hub object
    tag = f"case {case}: N={N} R={R} T={Tn} d={d} bias={bias} 3d={three_d} route={route}"
    if os.environ.get("SWEEP_VERBOSE"):
        print(tag, flush=True)
    try:
        if route != "auto":
            routes.set("distmult_bwd", route)
        else:
            routes.set("distmult_bwd", None)
        dm = DistMult(R, d, N, R, b_init="normal" if bias else None).to(DEV)
        with torch.no_grad():
            for prm in dm.parameters():
                prm.copy_(torch.from_numpy(rng.standard_normal(tuple(prm.shape)).astype(np.float32)))
        nodes = torch.from_numpy(rng.standard_normal((N, d)).astype(np.float32)).to(DEV).requires_grad_(True)
        shaped = tr.reshape(2, Tn // 2, 3) if three_d else tr
        sc = dm(torch.from_numpy(shaped).to(DEV), nodes)
        g = rng.standard_normal(tuple(sc.shape)).astype(np.float32)
        sc.backward(torch.from_numpy(g).to(DEV))
        P = {k: v.detach().cpu().numpy() for k, v in dm.named_parameters()}
        ref = oracle.distmult_forward(shaped, nodes.detach().cpu().numpy(), P["relations"], P.get("sbias"), P.get("pbias"), P.get("obias"))
        dn, dr, dsb, dpb, dob = oracle.distmult_backward(tr, nodes.detach().cpu().numpy(), P["relations"], g, bias)
        errs = {"scores": rel_err(sc, ref)}
        if Tn:
            errs["nodes"], errs["relations"] = rel_err(nodes.grad, dn), rel_err(dm.relations.grad, dr)
            if bias:
                errs.update(sbias=rel_err(dm.sbias.grad, dsb), pbias=rel_err(dm.pbias.grad, dpb), obias=rel_err(dm.obias.grad, dob))
        else:
            assert nodes.grad is None or float(nodes.grad.abs().max()) == 0.0
        bad = {k: v for k, v in errs.items() if not v < TOL}
        if bad:
            fails += 1
            print("FAIL", tag, bad, flush=True)
    except Exception as exc:  # noqa: BLE001
        fails += 1
        print("FAIL", tag, f"{type(exc).__name__}: {str(exc)[:200]}", flush=True)
print("done, cases:", cases, "failures:", fails)
