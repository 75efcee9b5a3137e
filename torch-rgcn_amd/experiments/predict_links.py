#!/usr/bin/env python3
"""Link prediction with an R-GCN encoder + DistMult decoder on MI355X -- counterpart of the reference's
experiments/predict_links.py (:19-228) without sacred.  Reads the reference's config schema unchanged:

    python experiments/predict_links.py configs/rgcn/lp-WN18.yaml [--data DIR] [--epochs N]

training.{epochs,graph_batch_size,sampling_method,negative_sampling.{sampling_rate,head_prob},optimiser.*}
encoder.{model,...,edge_dropout.general}  decoder.l2_penalty  evaluation.{final_run,filtered,check_every,batch_size,verbose}.
Differences: the positives are sampled by the native edge-neighbourhood sampler (milliseconds instead of minutes per
epoch), negatives are drawn on the GPU, and evaluation encodes the training graph once (utils/misc.py)."""
import argparse
import os
import sys
import time

import numpy as np
import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_rgcn  # noqa: E402
from torch_rgcn import routes  # noqa: E402
from torch_rgcn.functional import bce_with_logits, unit_gradient  # noqa: E402
from torch_rgcn.models import CompressionRelationPredictor, LinkPredictor  # noqa: E402
from utils.data import load_link_prediction_data  # noqa: E402
from utils.misc import evaluate, generate_true_dict, negative_sampling, select_sampling  # noqa: E402

OPTIMISERS = {"adam": torch.optim.Adam, "adamw": torch.optim.AdamW, "adagrad": torch.optim.Adagrad, "sgd": torch.optim.SGD}


def run(cfg, data_dir=None, epochs=None, quiet=False, max_test=None, synthetic=None, hipgraph=None):
    """-> (loss per epoch, {"mrr", "hits@1", "hits@3", "hits@10"} of the final evaluation).
    The training step (per-step graph build, encoder, decoder, loss, backward, optimiser) is captured once in a hipGraph and replayed
    every epoch on freshly sampled inputs copied into static buffers -- BY DEFAULT (hipgraph=None: unless route `capture` is "0";
    falls back to the eager loop, with a warning, if the capture fails; True insists, False is the reference's loop literally).
    The capture runs under the sync-free plan builder (route deferred_checks = 1, scoped to the warm-up and the capture)."""
    dataset, training, encoder = cfg["dataset"], cfg["training"], cfg["encoder"]
    decoder, evaluation = cfg.get("decoder", {}), cfg.get("evaluation", {})
    max_epochs = epochs or training.get("epochs", 5000)
    graph_batch_size = training.get("graph_batch_size")
    sampling_function = select_sampling(training.get("sampling_method", "uniform"))
    neg_sample_rate = training.get("negative_sampling", {}).get("sampling_rate")
    head_corrupt_prob = training.get("negative_sampling", {}).get("head_prob")
    edge_dropout = encoder.get("edge_dropout", {}).get("general", 0.0) if encoder.get("edge_dropout") else 0.0
    decoder_l2_penalty = decoder.get("l2_penalty", 0.0)
    filtered = evaluation.get("filtered", False)
    eval_every = evaluation.get("check_every", 2000)
    eval_kw = dict(batch_size=evaluation.get("batch_size", 16), verbose=evaluation.get("verbose", False) and not quiet,
                   filter_candidates=filtered)

    (n2i, nodes), (r2i, relations), train, test, all_triples = load_link_prediction_data(
        dataset["name"], use_test_set=evaluation.get("final_run", False), directory=data_dir, synthetic=synthetic)
    true_triples = generate_true_dict(all_triples)
    if max_test:
        test = test[:max_test]

    # block decomposition: pad the node list to a multiple of the block size (predict_links.py:54-67)
    if encoder.get("decomposition") and encoder["decomposition"]["type"] == "block":
        if "node_embedding" not in encoder:
            raise ValueError()
        block_size = encoder["node_embedding"] / encoder["decomposition"]["num_blocks"]
        added = 0
        while len(nodes) % block_size != 0:
            label = "null" + str(added)
            nodes.append(label)
            n2i[label] = len(nodes) - 1
            added += 1
        if not quiet:
            print(f"nodes padded to {len(nodes)} to make it divisible by {block_size} (added {added} null nodes).")

    device = torch.device("cuda")
    num_nodes, num_relations = len(n2i), len(r2i)
    test = torch.tensor(test, dtype=torch.long)
    train = np.asarray(train, dtype=np.int64)          # sampled from every epoch: keep it an array
    train_graph = torch.from_numpy(train)
    if encoder["model"] == "rgcn":
        kind = LinkPredictor
    elif encoder["model"] == "c-rgcn":
        kind = CompressionRelationPredictor
    else:
        raise NotImplementedError(f"'{encoder['model']}' encoder has not been implemented!")
    model = kind(nnodes=num_nodes, nrel=num_relations, encoder_config=encoder, decoder_config=decoder).to(device)
    opt_cfg = training.get("optimiser", {"algorithm": "adam", "learn_rate": 0.01, "weight_decay": 0.0})
    if opt_cfg["algorithm"] not in OPTIMISERS:
        raise NotImplementedError(f"'{opt_cfg['algorithm']}' optimiser has not been implemented!")
    extra = {"fused": True} if opt_cfg["algorithm"] in ("adam", "adamw") else {}
    insist = hipgraph is True
    if insist and not torch_rgcn.REPLAY_SAFE:
        import warnings
        warnings.warn("hipgraph=True although DEBUG_CLR_GRAPH_PACKET_CAPTURE was not 0 when the HIP runtime started: replays of a captured "
                      "step are known to go wrong on this runtime (torch_rgcn/__init__.py)")
    if hipgraph is None:
        hipgraph = routes.get("capture", "1") != "0" and torch_rgcn.REPLAY_SAFE
    if hipgraph:
        # a captured step cannot read the device-side range-check flags back: validate the triples the sampler draws from HERE,
        # once, on the host (the kernels additionally clamp and kill out-of-range triples, csrc/rgcn_build.hip)
        tr = np.asarray(train)
        assert tr.size == 0 or (tr[:, [0, 2]].min() >= 0 and tr[:, [0, 2]].max() < num_nodes and tr[:, 1].min() >= 0
                                and tr[:, 1].max() < num_relations), "training triples: node or relation index out of range"
        extra = {"capturable": True} if opt_cfg["algorithm"] in ("adam", "adamw") else {}
    optimiser = OPTIMISERS[opt_cfg["algorithm"]](model.parameters(), lr=opt_cfg["learn_rate"],
                                                  weight_decay=opt_cfg["weight_decay"], **extra)

    def report(tag, mrr, hits):
        kind_ = "filtered" if filtered else "raw"
        print(f"{tag} MRR({kind_}): {mrr:.4f} \tHits@1({kind_}): {hits[0]:.4f} \tHits@3({kind_}): {hits[1]:.4f} \t"
              f"Hits@10({kind_}): {hits[2]:.4f}")

    def sample_inputs():
        """one epoch's positives / negatives / labels / message graph (all sizes are the same every epoch)"""
        nonlocal graph_batch_size
        with torch.no_grad():
            if graph_batch_size is None:          # the whole graph
                positives, graph_batch_size = train, len(train)
            else:
                positives = sampling_function(train, sample_size=graph_batch_size, entities=n2i)
            positives = torch.as_tensor(positives, dtype=torch.long).to(device)
            negatives = positives[:, None, :].expand(graph_batch_size, neg_sample_rate, 3).contiguous()
            negatives = negative_sampling(negatives, num_nodes, head_corrupt_prob, device=device)
            batch_idx = torch.cat([positives, negatives], dim=0)
            train_lbl = torch.cat([torch.ones(graph_batch_size, device=device),
                                   torch.zeros(graph_batch_size * neg_sample_rate, device=device)])
            graph = positives
            if edge_dropout > 0.0:                # self-loop dropout happens inside the layer
                graph = graph[torch.randperm(graph.size(0), device=device)]
                graph = graph[round((1 - edge_dropout) * graph.size(0)):, :]     # (keeps the edge_dropout share, as upstream)
        return graph, batch_idx, train_lbl

    def train_step(graph, batch_idx, train_lbl):
        optimiser.zero_grad(set_to_none=False)
        predictions, penalty = model(graph, batch_idx)
        loss = bce_with_logits(predictions, train_lbl) + decoder_l2_penalty * penalty
        loss.backward(gradient=unit_gradient(loss.device))      # (no ones_like() fill per step; MaskedCrossEntropy skips the multiplication)
        optimiser.step()
        return loss

    captured = None
    if hipgraph:
        model.train()
        # the step is captured once on static input buffers; every epoch samples as the eager loop does (on the GPU) and copies
        # into them.  (Round 2 history: replays used to fault whenever eager kernels ran between them -- the HIP runtime replays
        # hipMemsetAsync NODES with stale arguments; the library zero-fills with kernels now, tools/hipgraph_repro/.)
        static = [t.clone() for t in sample_inputs()]
        # the warm-up steps below are real optimiser steps: parameters and optimiser state are put back afterwards (in place -- the
        # captured graph holds their addresses), so that a captured run starts epoch 1 from the same state as the eager run
        saved_params = [p.detach().clone() for p in model.parameters()]
        try:
            with routes.override(deferred_checks="1"):       # no device -> host reads inside the captured step
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):                    # warm-up off the capture: allocator pools, lazy inits
                    for _ in range(3):
                        train_step(*static)
                torch.cuda.current_stream().wait_stream(side)
                captured = torch.cuda.CUDAGraph()
                with torch.cuda.graph(captured):
                    static_loss = train_step(*static)
        except Exception as exc:  # noqa: BLE001  (the eager loop runs the same HIP kernels: a slower path, not another implementation)
            if insist:
                raise
            import warnings
            warnings.warn(f"hipGraph capture of the training step failed ({type(exc).__name__}: {exc}); running the eager loop")
            torch.cuda.synchronize()
            captured = None
        with torch.no_grad():
            for p, q in zip(model.parameters(), saved_params):
                p.copy_(q)
            for st in optimiser.state.values():              # Adam / SGD-momentum state after zero steps: all zeros
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
        del saved_params

    losses = []
    for epoch in range(1, max_epochs + 1):
        t1 = time.time()
        model.train()
        if captured is not None:
            for buf, fresh in zip(static, sample_inputs()):
                buf.copy_(fresh)
            captured.replay()
            loss = static_loss
            t2 = time.time()
        else:
            optimiser.zero_grad()
            graph, batch_idx, train_lbl = sample_inputs()
            predictions, penalty = model(graph, batch_idx)
            loss = bce_with_logits(predictions, train_lbl) + decoder_l2_penalty * penalty
            t2 = time.time()
            loss.backward(gradient=unit_gradient(loss.device))
            optimiser.step()
        torch.cuda.synchronize()
        t3 = time.time()
        losses.append(loss.item())
        if not quiet:
            print(f"[Epoch {epoch}] Loss: {loss.item():.5f} Forward: {t2 - t1:.3f}s Backward: {t3 - t2:.3f}s ")
        if epoch % eval_every == 0 and epoch != max_epochs:
            model.eval()
            mrr, hits, _ = evaluate(model=model, graph=train_graph, test_set=test, true_triples=true_triples,
                                    num_nodes=num_nodes, **eval_kw)
            if not quiet:
                report(f"[Epoch {epoch}]", mrr, hits)

    model.eval()
    t0 = time.time()
    mrr, hits, ranks = evaluate(model=model, graph=train_graph, test_set=test, true_triples=true_triples,
                                num_nodes=num_nodes, **eval_kw)
    if not quiet:
        report(f"[Final Scores] Total Epoch {max_epochs} ({len(test)} test triples ranked in {time.time() - t0:.2f}s)", mrr, hits)
    return losses, {"mrr": mrr, "hits@1": hits[0], "hits@3": hits[1], "hits@10": hits[2]}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--data", default=None, help="directory holding data/<name>/{train,valid,test}.txt")
    ap.add_argument("--synthetic", action="store_true", help="when the dataset files are absent, train on a random graph with "
                    "the dataset's entity / relation / triple counts (timing and plumbing only: MRR / Hits mean nothing)")
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--max-test", type=int, default=None)
    ap.add_argument("--hipgraph", action="store_true", help="insist on the captured training step (the default tries it and falls back to eager)")
    ap.add_argument("--eager", action="store_true", help="the reference's loop literally: no hipGraph capture")
    a = ap.parse_args()
    run(yaml.safe_load(open(a.config)), a.data, a.epochs, max_test=a.max_test, synthetic=True if a.synthetic else None,
        hipgraph=False if a.eager else (True if a.hipgraph else None))
