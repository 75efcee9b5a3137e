#!/usr/bin/env python3
"""Node classification with R-GCN on MI355X -- counterpart of the reference's experiments/classify_nodes.py
(:20-160) without sacred: `python experiments/classify_nodes.py configs/rgcn/nc-AIFB.yaml [--data DIR] [--epochs N]`.
Config keys follow the reference's YAML files (dataset.name, training.{epochs,learn_rate,weight_decay,optimiser},
rgcn.{node_embedding,hidden_size,num_layers,decomposition,edge_dropout}, l2 penalty on the first layer)."""
import argparse
import os
import sys
import time

import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_rgcn.models import EmbeddingNodeClassifier, NodeClassifier  # noqa: E402
from utils.data import load_node_classification_data  # noqa: E402


def run(cfg, data_dir=None, epochs=None, quiet=False):
    ds, tr, enc = cfg["dataset"], cfg["training"], cfg.get("rgcn", cfg.get("encoder", {}))
    triples, (n, r, c), tr_idx, tr_y, te_idx, te_y = load_node_classification_data(ds["name"], data_dir)
    dev = torch.device("cuda")
    kind = EmbeddingNodeClassifier if enc.get("model", "rgcn") == "e-rgcn" else NodeClassifier
    model = kind(triples=triples, nnodes=n, nrel=r, nfeat=None, nhid=enc.get("hidden_size", 16),
                 nlayers=enc.get("num_layers", 2), nclass=c, edge_dropout=enc.get("edge_dropout"),
                 decomposition=enc.get("decomposition"), nemb=enc.get("node_embedding")).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=tr.get("learn_rate", 0.01), weight_decay=tr.get("weight_decay", 0.0))
    tr_idx_t, te_idx_t = torch.as_tensor(tr_idx, device=dev), torch.as_tensor(te_idx, device=dev)
    tr_y_t, te_y_t = torch.as_tensor(tr_y, device=dev), torch.as_tensor(te_y, device=dev)
    l2 = tr.get("l2_penalty", enc.get("l2_penalty", 0.0))
    hist = []
    for epoch in range(epochs or tr.get("epochs", 50)):
        t0 = time.time()
        model.train()
        opt.zero_grad(set_to_none=True)
        logits = model()
        loss = torch.nn.functional.cross_entropy(logits[tr_idx_t], tr_y_t)
        if l2:   # reference classify_nodes.py:111-118: penalty on the first layer's (decomposed) weights
            first = model.rgc1
            for name in ("weights", "bases", "comps", "blocks"):
                if hasattr(first, name):
                    loss = loss + l2 * getattr(first, name).pow(2).sum()
        t1 = time.time()
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        t2 = time.time()
        with torch.no_grad():
            model.eval()
            out = model()
            acc_tr = (out[tr_idx_t].argmax(1) == tr_y_t).float().mean().item()
            acc_te = (out[te_idx_t].argmax(1) == te_y_t).float().mean().item()
        hist.append((loss.item(), acc_tr, acc_te))
        if not quiet:
            print(f"[Epoch {epoch + 1}] loss {loss.item():.5f} forward {t1 - t0:.4f}s backward {t2 - t1:.4f}s "
                  f"train acc {acc_tr:.3f} test acc {acc_te:.3f}")
    return hist


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--data", default=None)
    ap.add_argument("--epochs", type=int, default=None)
    a = ap.parse_args()
    run(yaml.safe_load(open(a.config)), a.data, a.epochs)
