#!/usr/bin/env python3
"""One training step (forward + backward) of every BASELINE.json config on dataset-shaped synthetic graphs
(the datasets themselves are not available offline).  Prints one JSON line per config.
    python tools/config_bench.py [--cpu]     # --cpu also times the PyTorch-CPU port where it fits"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "torch-rgcn_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from torch_rgcn import _native  # noqa: E402
from torch_rgcn.layers import DistMult, RelationalGraphConvolutionLP, RelationalGraphConvolutionNC  # noqa: E402
from torch_rgcn.models import NodeClassifier  # noqa: E402

DEV = torch.device("cuda:0")


def timed(fn, iters=10, warm=3):
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


def nc_model(name, N, R0, E, nhid, ncls, decomp, labelled):
    T = _native.synthetic_triples_host(N, R0, E, 1)
    t0 = time.time()
    model = NodeClassifier(triples=T, nnodes=N, nrel=R0, nhid=nhid, nclass=ncls, decomposition=decomp).to(DEV)
    idx = torch.arange(labelled, device=DEV)
    y = torch.randint(0, ncls, (labelled,), device=DEV)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(model()[idx], y)
        loss.backward()
        opt.step()
    step()
    torch.cuda.synchronize()
    build = time.time() - t0
    ms = timed(step)
    # the same step captured in a hipGraph (static NC graph, static shapes): launch-bound at this size
    ms_graph = None
    try:
        opt_g = torch.optim.Adam(model.parameters(), lr=0.01, capturable=True)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                opt_g.zero_grad(set_to_none=True)
                torch.nn.functional.cross_entropy(model()[idx], y).backward()
                opt_g.step()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        opt_g.zero_grad(set_to_none=True)
        with torch.cuda.graph(g):
            loss_g = torch.nn.functional.cross_entropy(model()[idx], y)
            loss_g.backward()
            opt_g.step()
        ms_graph = timed(g.replay)
    except Exception as exc:  # noqa: BLE001
        ms_graph = f"capture failed: {type(exc).__name__}: {exc}"[:200]
    print(json.dumps({"config": name, "N": N, "R0": R0, "E": E, "ms_per_train_step": round(ms, 3),
                      "ms_per_train_step_hipgraph": round(ms_graph, 3) if isinstance(ms_graph, float) else ms_graph,
                      "edges_per_s": round(E / ms * 1e3), "first_step_incl_graph_build_s": round(build, 2),
                      "params": sum(p.numel() for p in model.parameters())}), flush=True)


def s2_featureless_basis():
    """S1 with a featureless first layer (weight-table gather), basis B = 2."""
    N, R0, E, d = 1_000_000, 50, 10_000_000, 16
    T = _native.synthetic_triples_host(N, R0, E, 0)
    tp = torch.from_numpy(_native.add_inverse_and_self_host(T, N, R0))
    kw = dict(triples=tp, num_nodes=N, num_relations=2 * R0 + 1, out_features=d)
    l1 = RelationalGraphConvolutionNC(in_features=None, vertical_stacking=False,
                                      decomposition={"type": "basis", "num_bases": 2}, **kw).to(DEV)
    l2 = RelationalGraphConvolutionNC(in_features=d, vertical_stacking=True, **kw).to(DEV)

    def step():
        for p in list(l1.parameters()) + list(l2.parameters()):
            p.grad = None
        l2(torch.relu(l1())).pow(2).mean().backward()
    ms = timed(step, iters=5, warm=2)
    print(json.dumps({"config": "S2: S1 graph, featureless layer 1 with basis B=2 (no R x N x 16 table is materialised), layer 2 16->16",
                      "N": N, "R0": R0, "E": E, "ms_per_fwd_bwd": round(ms, 3), "edges_per_s": round(E / ms * 1e3)}), flush=True)


def featured_layers(title, N, R0, E, d, decomposition, seed):
    T = _native.synthetic_triples_host(N, R0, E, seed)
    tp = torch.from_numpy(_native.add_inverse_and_self_host(T, N, R0))
    kw = dict(triples=tp, num_nodes=N, num_relations=2 * R0 + 1, in_features=d, out_features=d,
              decomposition=decomposition)
    l1 = RelationalGraphConvolutionNC(vertical_stacking=False, **kw).to(DEV)
    l2 = RelationalGraphConvolutionNC(vertical_stacking=True, **kw).to(DEV)
    X = torch.randn(N, d, device=DEV, requires_grad=True)

    def step():
        for p in [X] + list(l1.parameters()) + list(l2.parameters()):
            p.grad = None
        l2(torch.relu(l1(X))).pow(2).mean().backward()
    ms = timed(step)
    print(json.dumps({"config": title, "N": N, "R0": R0, "E": E,
                      "ms_per_fwd_bwd": round(ms, 3), "edges_per_s": round(E / ms * 1e3)}), flush=True)


def wn18_lp():
    N, R0, d = 40_943, 18, 200
    ed = {"general": 0.5, "self_loop": 0.2, "self_loop_type": "schlichtkrull-dropout"}
    layer = RelationalGraphConvolutionLP(num_nodes=N, num_relations=2 * R0 + 1, in_features=d, out_features=d,
                                         edge_dropout=ed, decomposition={"type": "basis", "num_bases": 2},
                                         w_init="glorot-normal", b_init="zeros").to(DEV)
    dm = DistMult(R0, d, N, R0).to(DEV)
    emb = torch.randn(N, d, device=DEV, requires_grad=True)
    for tag, E in (("train graph 15k", 15_000), ("eval graph 141k", 141_442)):
        graph = torch.from_numpy(_native.synthetic_triples_host(N, R0, E, 3))
        batch = torch.from_numpy(_native.synthetic_triples_host(N, R0, 330_000, 4)).to(DEV)
        y = torch.rand(330_000, device=DEV).round()

        def step():
            for p in [emb] + list(layer.parameters()) + list(dm.parameters()):
                p.grad = None
            x = layer(graph, torch.relu(emb))
            loss = torch.nn.functional.binary_cross_entropy_with_logits(dm(batch, x), y)
            loss.backward()
        ms = timed(step, iters=5, warm=2)
        t0 = time.perf_counter()
        with torch.no_grad():
            layer(graph, emb)
        torch.cuda.synchronize()
        fwd = 1e3 * (time.perf_counter() - t0)
        print(json.dumps({"config": f"WN18-shaped LP layer d=200 basis 2 + DistMult 330k triples, {tag}", "N": N,
                          "graph_triples": E, "ms_per_fwd_bwd": round(ms, 2), "encoder_fwd_ms_incl_host_graph_build": round(fwd, 2)}),
              flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    todo = a.only.split(",") if a.only else ["aifb", "mutag", "am", "wn18", "s2", "s1"]
    if "s2" in todo:
        s2_featureless_basis()
    if "aifb" in todo:
        nc_model("AIFB-shaped NodeClassifier (featureless L1, hidden 16, 4 classes)", 8285, 45, 29043, 16, 4, None, 176)
    if "mutag" in todo:
        nc_model("MUTAG-shaped NodeClassifier (basis 30, hidden 16, 2 classes)", 23644, 23, 74227, 16, 2,
                 {"type": "basis", "num_bases": 30}, 340)
    if "am" in todo:
        featured_layers("AM-shaped, block-diagonal (nb=4), 2 featured layers d=16", 1_666_764, 133, 5_988_321, 16,
                        {"type": "block", "num_blocks": 4}, 2)
    if "s1" in todo:   # SURVEY.md 8(d) S1 variants (i)-(iii); (i) is what bench.py times
        featured_layers("S1(i): no decomposition, W 101x16x16", 1_000_000, 50, 10_000_000, 16, None, 0)
        featured_layers("S1(ii): basis B=10", 1_000_000, 50, 10_000_000, 16, {"type": "basis", "num_bases": 10}, 0)
        featured_layers("S1(iii): block nb=4", 1_000_000, 50, 10_000_000, 16, {"type": "block", "num_blocks": 4}, 0)
    if "amreal" in todo:
        nc_model("AM-shaped NodeClassifier as shipped (featureless L1, basis 40, hidden 10, 11 classes)", 1666764, 133,
                 5988321, 10, 11, {"type": "basis", "num_bases": 40}, 802)
    if "wn18" in todo:
        wn18_lp()
