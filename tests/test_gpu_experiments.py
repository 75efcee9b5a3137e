"""End-to-end: the experiment counterparts train on dataset-shaped synthetic graphs through the HIP layers."""
import os
import sys

import pytest
import yaml

from conftest import PKG

pytestmark = pytest.mark.gpu


def cfg(name):
    return yaml.safe_load(open(os.path.join(PKG, "configs", "rgcn", name)))


def test_classify_nodes_aifb_shaped_learns():
    sys.path.insert(0, os.path.join(PKG, "experiments"))
    import classify_nodes
    hist = classify_nodes.run(cfg("nc-AIFB.yaml"), epochs=25, quiet=True)
    assert hist[-1][0] < 0.6 * hist[0][0]          # loss goes down
    assert hist[-1][1] > 0.9                         # and the (featureless, 12 M parameter) model fits the train labels


def test_classify_nodes_mutag_shaped_basis():
    sys.path.insert(0, os.path.join(PKG, "experiments"))
    import classify_nodes
    hist = classify_nodes.run(cfg("nc-MUTAG.yaml"), epochs=15, quiet=True)
    assert hist[-1][0] < hist[0][0]


def test_predict_links_small_graph_trains_and_ranks():
    sys.path.insert(0, os.path.join(PKG, "experiments"))
    import predict_links
    c = cfg("lp-WN18.yaml")
    c["dataset"]["name"] = "fb-toy"
    c["encoder"].update(node_embedding=32, hidden1_size=32)
    c["training"].update(graph_batch_size=2000)
    c["evaluation"] = {"batch_size": 32, "max_test": 100}
    hist, metrics = predict_links.run(c, epochs=30, quiet=True)
    assert hist[-1] < hist[0]
    assert 0.0 < metrics["mrr"] <= 1.0 and metrics["hits@10"] >= metrics["hits@1"]
