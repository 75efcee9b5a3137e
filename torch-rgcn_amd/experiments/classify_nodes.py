#!/usr/bin/env python3
"""Node classification with R-GCN on MI355X -- counterpart of the reference's experiments/classify_nodes.py (:19-175)
without sacred.  Reads the reference's config schema unchanged:

    python experiments/classify_nodes.py configs/rgcn/nc-AIFB.yaml [--data DIR] [--epochs N] [--repeats K] [--eager]

The training step (forward, loss, backward, optimiser) and the evaluation forward are captured ONCE as hipGraphs and replayed every
epoch BY DEFAULT (full-batch node classification issues the same launches every epoch, and on the benchmark graphs the eager step is
bound by launch overhead: AIFB 0.68 ms eager against 0.39 replayed for 0.2 ms of kernels); --eager / route `capture=0` runs the
reference's loop literally.

dataset.{name,prune}  training.{epochs,optimiser.{algorithm,learn_rate,weight_decay},use_cuda}
rgcn.{model,hidden_size,num_layers,decomposition,layer1_l2_penalty,node_embeddings,node_embedding_l2_penalty}
evaluation.final_run.   The HIP layers have no CPU path, so a GPU is required whatever `use_cuda` says."""
import argparse
import os
import sys
import time
from statistics import stdev

import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_rgcn  # noqa: E402
from torch_rgcn import routes  # noqa: E402
from torch_rgcn.functional import MaskedCrossEntropy, unit_gradient  # noqa: E402
from torch_rgcn.models import EmbeddingNodeClassifier, NodeClassifier  # noqa: E402
from utils.data import load_node_classification_data  # noqa: E402

OPTIMISERS = {"adam": torch.optim.Adam, "adamw": torch.optim.AdamW, "adagrad": torch.optim.Adagrad}


def _first_layer_l2(model, decomposition):
    kind = decomposition["type"] if decomposition is not None else None
    if kind == "basis":
        return model.rgc1.bases.pow(2).sum() + model.rgc1.comps.pow(2).sum()
    if kind == "block":
        return model.rgc1.blocks.pow(2).sum()
    return model.rgc1.weights.pow(2).sum()


def _capture(fn, warmup=3, model=None, optimiser=None):
    """capture `fn` (static shapes, static graph) in a hipGraph after `warmup` eager runs on a side stream.  When `fn` trains
    (model / optimiser given) the warm-up runs are real optimiser steps: parameters and optimiser state are put back afterwards, in
    place (the captured graph holds their addresses) -- a captured run starts epoch 1 from the same state as the eager one."""
    saved = [p.detach().clone() for p in model.parameters()] if model is not None else None
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = fn()
    finally:
        # also when the capture FAILS: the eager fallback must start epoch 1 from the initial state, not after the warm-up's steps
        if saved is not None:
            torch.cuda.synchronize()
            with torch.no_grad():
                for p, q in zip(model.parameters(), saved):
                    p.copy_(q)
                for st in optimiser.state.values():              # Adam state after zero steps: all zeros (step counter included)
                    for v in st.values():
                        if torch.is_tensor(v):
                            v.zero_()
    return graph, out


def run(cfg, data_dir=None, epochs=None, quiet=False, hipgraph=None, synthetic=None):
    """one training run -> [(loss, train accuracy, test accuracy) per epoch] (accuracies in [0, 1]).
    hipgraph: None (default) = replay the whole training step (forward, loss, backward, optimiser) and the evaluation forward as
    two captured hipGraphs whenever the optimiser can be captured (adam / adamw) and route `capture` is not "0" -- falling back
    to the eager loop, with a warning, if the capture fails; True = insist (errors surface); False = the eager loop."""
    dataset, training, rgcn, evaluation = cfg["dataset"], cfg["training"], cfg["rgcn"], cfg.get("evaluation", {})
    assert training is not None, "Training configuration is not specified!"
    epochs = epochs or training.get("epochs", 50)
    decomposition = rgcn.get("decomposition")
    l2_first, l2_emb = rgcn.get("layer1_l2_penalty", 0.0), rgcn.get("node_embedding_l2_penalty", 0.0)

    # the validation split is the test set unless this is a final run (classify_nodes.py:40-43)
    triples, (n2i, i2n), (r2i, i2r), train, test = load_node_classification_data(
        dataset["name"], use_test_set=evaluation.get("final_run", False), prune=dataset.get("prune", False),
        directory=data_dir, synthetic=synthetic)
    device = torch.device("cuda")
    train_idx = torch.tensor([n2i[name] for name in train], dtype=torch.long, device=device)
    train_lbl = torch.tensor(list(train.values()), dtype=torch.long, device=device)
    test_idx = torch.tensor([n2i[name] for name in test], dtype=torch.long, device=device)
    test_lbl = torch.tensor(list(test.values()), dtype=torch.long, device=device)
    num_classes = len(set(train.values()) | set(test.values()))

    if rgcn.get("model", "rgcn") == "rgcn":
        kind = NodeClassifier
    elif rgcn["model"] == "e-rgcn":
        kind = EmbeddingNodeClassifier
    else:
        raise NotImplementedError(f"'{rgcn['model']}' model has not been implemented!")
    model = kind(triples=triples, nnodes=len(n2i), nrel=len(r2i), nclass=num_classes, nhid=rgcn.get("hidden_size", 16),
                 nlayers=rgcn.get("num_layers", 2), decomposition=decomposition,
                 nemb=rgcn.get("node_embeddings", 10)).to(device)

    opt_cfg = training.get("optimiser", {"algorithm": "adam", "learn_rate": 0.01, "weight_decay": 0.0})
    if opt_cfg["algorithm"] not in OPTIMISERS:
        raise NotImplementedError(f"'{opt_cfg['algorithm']}' optimiser has not been implemented!")
    adam_like = opt_cfg["algorithm"] in ("adam", "adamw")
    if hipgraph and not adam_like:
        raise NotImplementedError("hipgraph=True needs a capturable optimiser (adam / adamw)")
    insist = hipgraph is True
    if insist and not torch_rgcn.REPLAY_SAFE:
        import warnings
        warnings.warn("hipgraph=True although DEBUG_CLR_GRAPH_PACKET_CAPTURE was not 0 when the HIP runtime started: replays of a captured "
                      "step are known to go wrong on this runtime (torch_rgcn/__init__.py)")
    if hipgraph is None:
        hipgraph = adam_like and routes.get("capture", "1") != "0" and torch_rgcn.REPLAY_SAFE
    # one fused multi-tensor kernel per step instead of ~10 elementwise passes over every parameter (AM: 667 M of them)
    extra = {"fused": True, **({"capturable": True} if hipgraph else {})} if adam_like else {}
    optimiser = OPTIMISERS[opt_cfg["algorithm"]](model.parameters(), lr=opt_cfg["learn_rate"],
                                                  weight_decay=opt_cfg["weight_decay"], **extra)
    # criterion(model()[train_idx, :], train_lbl) with nn.CrossEntropyLoss() (classify_nodes.py:107-110) as one launch for loss + gradient
    criterion = MaskedCrossEntropy(train_idx, train_lbl, len(n2i))
    if l2_emb > 0.0 and rgcn.get("model") != "e-rgcn":
        raise ValueError(f"Cannot apply L2-regularisation on node embeddings for {rgcn.get('model')} model")

    def objective():
        loss = criterion(model())
        if l2_first > 0.0:
            loss = loss + l2_first * _first_layer_l2(model, decomposition)
        if l2_emb > 0.0:
            loss = loss + l2_emb * model.node_embeddings.pow(2).sum()
        return loss

    def train_step():
        optimiser.zero_grad(set_to_none=True)
        loss = objective()
        loss.backward(gradient=unit_gradient(loss.device))      # (no ones_like() fill per step; MaskedCrossEntropy skips the multiplication)
        optimiser.step()
        return loss

    def predict():
        with torch.no_grad():
            return model()

    if hipgraph:
        try:
            model.train()
            train_graph, static_loss = _capture(train_step, model=model, optimiser=optimiser)
            model.eval()
            eval_graph, static_logits = _capture(predict)
        except Exception as exc:  # noqa: BLE001  (the eager loop runs the same HIP kernels: a slower path, not another implementation)
            if insist:
                raise
            import warnings
            warnings.warn(f"hipGraph capture of the training step failed ({type(exc).__name__}: {exc}); running the eager loop")
            torch.cuda.synchronize()
            hipgraph = False
            # a fresh optimiser without capturable=True (its device-side step counters are the captured step's business); parameters
            # were put back by _capture
            optimiser = OPTIMISERS[opt_cfg["algorithm"]](model.parameters(), lr=opt_cfg["learn_rate"], weight_decay=opt_cfg["weight_decay"],
                                                          **({"fused": True} if adam_like else {}))

    history = []
    for epoch in range(1, epochs + 1):
        t1 = time.time()
        model.train()
        if hipgraph:
            train_graph.replay()
            loss, t2 = static_loss, t1
        else:
            optimiser.zero_grad()
            loss = objective()
            t2 = time.time()
            loss.backward(gradient=unit_gradient(loss.device))
            optimiser.step()
        torch.cuda.synchronize()
        t3 = time.time()
        with torch.no_grad():
            model.eval()
            if hipgraph:
                eval_graph.replay()
                out = static_logits
            else:
                out = model()
            train_acc = (out[train_idx].argmax(dim=-1) == train_lbl).float().mean().item()
            test_acc = (out[test_idx].argmax(dim=-1) == test_lbl).float().mean().item()
        history.append((loss.item(), train_acc, test_acc))
        if not quiet:
            print(f"[Epoch {epoch}] Loss: {loss.item():.5f} Forward: {t2 - t1:.3f}s Backward: {t3 - t2:.3f}s "
                  f"Train Accuracy: {100 * train_acc:.2f} Test Accuracy: {100 * test_acc:.2f}")
    if not quiet:
        print(f"[Evaluation] Test Accuracy: {100 * history[-1][2]:.2f}")
    return history


def repeat(cfg, repeats=1, **kw):
    """average test accuracy (percent) and its standard error over `repeats` runs (classify_nodes.py:156-175)"""
    accs = [100 * run(cfg, **kw)[-1][2] for _ in range(repeats)]
    std = stdev(accs) if len(accs) != 1 else 0
    return round(sum(accs) / len(accs), 2), round(std / len(accs) ** 0.5, 2)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--data", default=None, help="directory holding data/<name>/...")
    ap.add_argument("--synthetic", action="store_true", help="when the dataset files are absent, train on a random graph with "
                    "the dataset's node / relation / edge counts (timing and plumbing only: the accuracies mean nothing)")
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--repeats", type=int, default=1)
    ap.add_argument("--hipgraph", action="store_true", help="insist on the captured step (the default tries it and falls back to eager)")
    ap.add_argument("--eager", action="store_true", help="the reference's loop literally: no hipGraph capture")
    a = ap.parse_args()
    avg, ste = repeat(yaml.safe_load(open(a.config)), a.repeats, data_dir=a.data, epochs=a.epochs,
                      hipgraph=False if a.eager else (True if a.hipgraph else None),
                      synthetic=True if a.synthetic else None)
    print(f"test accuracy {avg} +- {ste}" + (" [SYNTHETIC DATA]" if a.synthetic else ""))
