"""Randomised end-to-end sweep of the node-classification models (NodeClassifier featureless, EmbeddingNodeClassifier; none / basis /
block decomposition; hidden widths 4..64; 2..16 classes; the fused ReLU between the layers and the one-launch masked cross-entropy head, as
experiments/classify_nodes.py runs them) against the oracle's two layers composed by hand in float64 around them: logits, loss and
every parameter gradient.  python tools/random_sweep_models.py SEED [CASES]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "torch-rgcn_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import oracle  # noqa: E402
from torch_rgcn.functional import MaskedCrossEntropy, unit_gradient  # noqa: E402
from torch_rgcn.models import EmbeddingNodeClassifier, NodeClassifier  # noqa: E402

DEV, TOL = "cuda:0", 1e-4


def rel_err(a, b):
    a = a.detach().cpu().numpy().astype(np.float64)
    den = np.abs(b).max()
    return float(np.abs(a - b).max() / den) if den > 0 else float(np.abs(a).max())


def layer_params(layer):
    P = {n: p.detach().cpu().numpy() for n, p in layer.named_parameters()}
    return P, P.pop("bias", None)


def mode_of(P, diag=False):
    return "diag" if diag else "basis" if "bases" in P else "block" if "blocks" in P else "none"


seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rng = np.random.default_rng(seed)
fails = 0
for case in range(cases):
    N = int(rng.choice([8, 30, 64, 100, 332, 1000, 3000]))
    R0 = int(rng.choice([1, 2, 5, 12, 40]))
    E = int(rng.choice([0, 1, 5, 20])) * N
    emb = bool(rng.random() < 0.35)
    nhid = int(rng.choice([4, 8, 10, 16, 17, 32, 64]))
    nclass = int(rng.choice([2, 3, 4, 8, 11, 16]))
    kind = str(rng.choice(["none", "basis", "block"]))
    decomp = None
    if kind == "basis":
        decomp = {"type": "basis", "num_bases": int(rng.choice([1, 2, 5, 30, 40]))}
    elif kind == "block":
        nb = int(rng.choice([2, 4]))
        if not emb and N % nb:          # (the featureless block layer splits the NODES into blocks: the reference asserts divisibility)
            nb = 2
        nhid, nclass = nb * max(1, nhid // nb), nb * max(1, nclass // nb)
        decomp = {"type": "block", "num_blocks": nb}
    n_lab = int(rng.integers(1, max(2, N // 2)))
    tag = f"case {case}: N={N} R0={R0} E={E} emb={emb} nhid={nhid} nclass={nclass} decomp={decomp} labelled={n_lab}"
    if os.environ.get("SWEEP_VERBOSE"):
        print(tag, flush=True)
    try:
        T = oracle.synthetic_triples(N, R0, E, seed=8000 + case) if E else np.zeros((0, 3), np.int64)
        kw = dict(triples=torch.from_numpy(T), nnodes=N, nrel=R0, nhid=nhid, nclass=nclass, decomposition=decomp)
        model = (EmbeddingNodeClassifier(nemb=nhid, **kw) if emb else NodeClassifier(**kw)).to(DEV)
        with torch.no_grad():
            for prm in model.parameters():
                prm.copy_(torch.from_numpy(rng.standard_normal(tuple(prm.shape)).astype(np.float32) * 0.3))
        idx = rng.choice(N, n_lab, replace=False)
        lab = rng.integers(0, nclass, n_lab)
        crit = MaskedCrossEntropy(torch.from_numpy(idx).to(DEV), torch.from_numpy(lab).to(DEV), N)
        first = model.rgcn_no_hidden if emb else model.rgc1
        seen, inner = {}, first.forward_activated

        def spy(*a, **k):            # the hidden activation of the step itself: its > 0 pattern is the ReLU mask of the backward pass
            seen["a"] = inner(*a, **k)
            return seen["a"]
        first.forward_activated = spy
        logits = model()
        loss = crit(logits)
        loss.backward(gradient=unit_gradient(loss.device))
        # ---- the same step from the oracle's layers
        tp = oracle.add_inverse_and_self(T, N, R0)
        R = 2 * R0 + 1
        first, second = (model.rgcn_no_hidden, model.rgc1) if emb else (model.rgc1, model.rgc2)
        P1, b1 = layer_params(first)
        P2, b2 = layer_params(second)
        X = model.node_embeddings.detach().cpu().numpy() if emb else None
        m1, m2 = mode_of(P1, diag=emb), mode_of(P2)
        v2 = not emb            # NodeClassifier: layer 2 stacks vertically; e-rgcn's rgc1 is the horizontal (nlayers = 1) layer
        h = oracle.nc_layer(tp, N, R, X, P1, m1, b1, False, None)["out"]
        a = np.maximum(h, 0)
        lg = oracle.nc_layer(tp, N, R, a, P2, m2, b2, v2, None)["out"]
        t = torch.from_numpy(lg).double().requires_grad_(True)
        ref_loss = torch.nn.functional.cross_entropy(t[torch.from_numpy(idx)], torch.from_numpy(lab))
        ref_loss.backward()
        g = t.grad.float().numpy()
        r2 = oracle.nc_layer(tp, N, R, a, P2, m2, b2, v2, g)
        # (an element of h within round-off of 0 may sit on the other side in the float64 composition: the mask is the step's own)
        r1 = oracle.nc_layer(tp, N, R, X, P1, m1, b1, False, (r2["dX"] * (seen["a"].detach().cpu().numpy()[:, :h.shape[1]] > 0)).astype(np.float32))
        gin = {"l2": g, "l1": r2["dX"] * (seen["a"].detach().cpu().numpy()[:, :h.shape[1]] > 0)}
        errs = {"logits": rel_err(logits, lg), "loss": abs(loss.item() - ref_loss.item()) / max(abs(ref_loss.item()), 1.0)}      # (one labelled node: a loss near 0)
        for lname, layer, res in (("l1", first, r1), ("l2", second, r2)):
            for n, gv in res["grads"].items():
                errs[f"{lname}.{n}"] = rel_err(getattr(layer, n).grad, gv)
            if layer.bias is not None:       # a column sum of signed terms (two classes: +a and -a with a ~ 1e-5 of the terms): measured against the sum of |terms|
                errs[f"{lname}.bias"] = float(np.abs(layer.bias.grad.cpu().numpy() - res["db"]).max() / max(np.abs(gin[lname]).sum(0).max(), 1e-30))
        if emb:
            errs["embeddings"] = rel_err(model.node_embeddings.grad, r1["dX"])
        bad = {k: v for k, v in errs.items() if not v < TOL}
        if bad:
            fails += 1
            print("FAIL", tag, bad, flush=True)
    except Exception as exc:  # noqa: BLE001
        fails += 1
        print("FAIL", tag, f"{type(exc).__name__}: {str(exc)[:200]}", flush=True)
print("done, cases:", cases, "failures:", fails)
