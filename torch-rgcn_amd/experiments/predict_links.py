#!/usr/bin/env python3
"""Link prediction with an R-GCN encoder + DistMult decoder on MI355X -- counterpart of the reference's
experiments/predict_links.py (:19-228) without sacred:
`python experiments/predict_links.py configs/rgcn/lp-WN18.yaml [--data DIR] [--epochs N]`."""
import argparse
import os
import sys
import time

import numpy as np
import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_rgcn.models import CompressionRelationPredictor, LinkPredictor  # noqa: E402
from utils.data import load_link_prediction_data  # noqa: E402
from utils.misc import evaluate, generate_true_dict, negative_sampling, sample_edges  # noqa: E402


def run(cfg, data_dir=None, epochs=None, quiet=False, seed=0):
    ds, tr, enc, dec = cfg["dataset"], cfg["training"], cfg["encoder"], cfg.get("decoder", {})
    (n, r), train, valid, test = load_link_prediction_data(ds["name"], data_dir)
    heads, tails = generate_true_dict(np.concatenate([train, valid, test]))
    dev = torch.device("cuda")
    rng = np.random.default_rng(seed)
    kind = CompressionRelationPredictor if enc.get("model") == "c-rgcn" else LinkPredictor
    model = kind(nnodes=n, nrel=r, encoder_config=enc, decoder_config=dec).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=tr.get("learn_rate", 0.01), weight_decay=tr.get("weight_decay", 0.0))
    drop = (enc.get("edge_dropout") or {}).get("general", 0.0)
    hist = []
    for epoch in range(epochs or tr.get("epochs", 10)):
        t0 = time.time()
        model.train()
        positives = sample_edges(train, tr.get("graph_batch_size", 30000), rng)
        batch, labels = negative_sampling(positives, n, tr.get("negative_sampling", {}).get("sampling_rate", 10), rng)
        keep = rng.random(len(positives)) >= drop          # general edge dropout on the message graph
        graph = torch.as_tensor(positives[keep])
        opt.zero_grad(set_to_none=True)
        scores, penalty = model(graph, torch.as_tensor(batch, device=dev))
        loss = torch.nn.functional.binary_cross_entropy_with_logits(scores, torch.as_tensor(labels, device=dev))
        loss = loss + (dec.get("l2_penalty", 0.0) or 0.0) * penalty
        t1 = time.time()
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        hist.append(loss.item())
        if not quiet:
            print(f"[Epoch {epoch + 1}] loss {loss.item():.5f} forward {t1 - t0:.4f}s backward {time.time() - t1:.4f}s")
    metrics = evaluate(model, torch.as_tensor(train), test[: cfg.get("evaluation", {}).get("max_test", 2000)], heads, tails, n,
                       batch_size=cfg.get("evaluation", {}).get("batch_size", 64))
    if not quiet:
        print("test:", {k: round(v, 4) for k, v in metrics.items()})
    return hist, metrics


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--data", default=None)
    ap.add_argument("--epochs", type=int, default=None)
    a = ap.parse_args()
    run(yaml.safe_load(open(a.config)), a.data, a.epochs)
