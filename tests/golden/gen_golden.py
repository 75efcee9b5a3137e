#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ by running the
REFERENCE ITSELF (thiviyanT/torch-rgcn, mounted read-only at /root/reference).

Runs only in the build container -- the reference cannot travel to the GPU box;
the .npz files it writes (data only: inputs, expected outputs, expected grads) do.

    python tests/golden/gen_golden.py            # rewrites tests/golden/*.npz

Sets:
  g1_nc_*    RelationalGraphConvolutionNC, tiny graph with duplicates + isolated node
  g2_lp_*    RelationalGraphConvolutionLP, eval and deterministic train mode
  g3_utils   add_inverse_and_self / stack_matrices / sum_sparse on random input
  g4_model_* NodeClassifier / EmbeddingNodeClassifier: logits, loss, grads, 3 Adam steps
  g5_distmult DistMult.forward (2-D, 3-D, +-bias) and s_penalty
  g6_mid     N=2000, R0=10, E=20000, d=16 NC layer pair (int32 triples, fp32 tensors)
  g7_eval_*  utils/misc.py evaluate(): filtered and raw ranks (ties included) of an LP encoder + DistMult model
  g8_sampler utils/misc.py edge_neighborhood(): picks under two np.random seeds
  g9_lp_loader utils/data.py load_link_prediction_data() on a tiny text dataset
  g10_nc_loader utils/data.py load_node_classification_data() on a tiny AIFB-layout dataset
  g11_dup_*  NC layers on graphs of thousands of PARALLEL edges between one or two nodes: inputs on which the reference's own fp32
             round-off (a chain of equal terms per output element) is above 1e-4 of the exact result
"""
import os
import sys
import warnings

import numpy as np
import torch

REF = os.environ.get("RGCN_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
warnings.filterwarnings("ignore")

from torch_rgcn.layers import (DistMult, RelationalGraphConvolutionLP,  # noqa: E402
                               RelationalGraphConvolutionNC)
from torch_rgcn.models import EmbeddingNodeClassifier, NodeClassifier  # noqa: E402
from torch_rgcn.utils import add_inverse_and_self, stack_matrices, sum_sparse  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def seeded_params(module, gen):
    """Overwrite every parameter with seeded N(0, 0.5) values (bias included)."""
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * 0.5)


def tiny_graph():
    """N=8 (divisible by 2 and 4), R0=3, 15 distinct edges + 2 duplicates; node 7 isolated."""
    base = [(0, 0, 1), (1, 0, 2), (2, 0, 3), (3, 0, 0), (0, 1, 4), (4, 1, 5), (5, 1, 6), (6, 1, 0),
            (1, 2, 3), (3, 2, 5), (5, 2, 1), (2, 2, 6), (0, 0, 2), (0, 0, 3), (4, 1, 6),
            (0, 0, 1), (5, 2, 1)]  # last two duplicate earlier edges
    return torch.tensor(base, dtype=torch.long), 8, 3


def grads_of(module, out, g, extra=()):
    module.zero_grad()
    for t in extra:
        if t.grad is not None:
            t.grad = None
    out.backward(g)
    res = {f"grad_{n}": p.grad.detach().numpy().copy() for n, p in module.named_parameters() if p.grad is not None}
    return res


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"{name}: {os.path.getsize(path)} bytes")


def params_np(module):
    return {f"param_{n}": p.detach().numpy().copy() for n, p in module.named_parameters()}


# ------------------------------------------------------------------ G1
def g1():
    T, N, R0 = tiny_graph()
    Tp = add_inverse_and_self(T, N, R0)
    R = 2 * R0 + 1
    gen = torch.Generator().manual_seed(101)
    cases = []
    for vertical in (False, True):
        for decomp in (None, {"type": "basis", "num_bases": 2}, {"type": "block", "num_blocks": 2}):
            for featureless in (False, True):
                if vertical and featureless:
                    continue  # invalid in the reference (shape error)
                cases.append((vertical, decomp, featureless, False))
    cases.append((False, None, False, True))  # diag
    for vertical, decomp, featureless, diag in cases:
        d_in, d_out = 4, 6
        layer = RelationalGraphConvolutionNC(triples=Tp, num_nodes=N, num_relations=R,
                                             in_features=None if featureless else d_in, out_features=d_out,
                                             decomposition=decomp, vertical_stacking=vertical,
                                             diag_weight_matrix=diag)
        seeded_params(layer, gen)
        X = None if featureless else torch.randn(N, d_in, generator=gen).requires_grad_(True)
        out = layer(X) if X is not None else layer()
        g = torch.randn(out.shape, generator=gen)
        grads = grads_of(layer, out, g)
        tag = "{}_{}_{}".format("v" if vertical else "h",
                                "diag" if diag else (decomp["type"] if decomp else "none"),
                                "fl" if featureless else "ft")
        arrs = dict(triples=T.numpy(), triples_plus=Tp.numpy(), num_nodes=N, num_rels=R0, vertical=int(vertical),
                    out=out.detach().numpy(), g=g.numpy(), **params_np(layer), **grads)
        if X is not None:
            arrs["X"] = X.detach().numpy()
            arrs["grad_X"] = X.grad.numpy()
        save("g1_nc_" + tag, **arrs)


# ------------------------------------------------------------------ G2
def g2():
    T, N, R0 = tiny_graph()
    R = 2 * R0 + 1
    gen = torch.Generator().manual_seed(202)
    d_in, d_out = 4, 6
    for train, sl_type in ((False, "schlichtkrull-dropout"), (True, "schlichtkrull-dropout"), (True, "other")):
        for vertical in (False, True):
            for decomp in (None, {"type": "basis", "num_bases": 2}, {"type": "block", "num_blocks": 2}):
                if vertical and decomp is not None and decomp["type"] == "block":
                    continue  # reference bug: cat of 3-D block_diag with 2-D blocks_self raises (layers.py:526-528)
                ed = {"general": 0.5, "self_loop": 0.0, "self_loop_type": sl_type}
                layer = RelationalGraphConvolutionLP(num_nodes=N, num_relations=R, in_features=d_in,
                                                     out_features=d_out, edge_dropout=ed, decomposition=decomp,
                                                     vertical_stacking=vertical, w_init="glorot-normal",
                                                     b_init="zeros")
                seeded_params(layer, gen)
                layer.train(train)
                X = torch.randn(N, d_in, generator=gen).requires_grad_(True)
                out = layer(T, X)
                g = torch.randn(out.shape, generator=gen)
                grads = grads_of(layer, out, g)
                tag = "{}_{}_{}_{}".format("train" if train else "eval", "sd" if sl_type.startswith("schl") else "ot",
                                           "v" if vertical else "h", decomp["type"] if decomp else "none")
                save("g2_lp_" + tag, triples=T.numpy(), num_nodes=N, num_rels=R0, vertical=int(vertical),
                     train=int(train), self_loop_type=sl_type, X=X.detach().numpy(), grad_X=X.grad.numpy(),
                     out=out.detach().numpy(), g=g.numpy(), **params_np(layer), **grads)


# ------------------------------------------------------------------ G3
def g3():
    gen = torch.Generator().manual_seed(303)
    N, R0, E = 11, 4, 40
    T = torch.stack([torch.randint(0, N, (E,), generator=gen), torch.randint(0, R0, (E,), generator=gen),
                     torch.randint(0, N, (E,), generator=gen)], dim=1)
    Tp = add_inverse_and_self(T, N, R0)
    R = 2 * R0 + 1
    vi, vs = stack_matrices(Tp, N, R, vertical_stacking=True)
    hi, hs = stack_matrices(Tp, N, R, vertical_stacking=False)
    ones = torch.ones(Tp.size(0))
    vsum = sum_sparse(vi, ones, vs, row_normalisation=True)
    hsum = sum_sparse(hi, ones, hs, row_normalisation=False)
    save("g3_utils", triples=T.numpy(), num_nodes=N, num_rels=R0, triples_plus=Tp.numpy(),
         ver_idx=vi.numpy(), ver_size=np.array(vs), hor_idx=hi.numpy(), hor_size=np.array(hs),
         ver_sums=vsum.numpy(), hor_sums=hsum.numpy())


# ------------------------------------------------------------------ G4
def g4():
    gen = torch.Generator().manual_seed(404)
    N, R0, E, ncls = 12, 3, 40, 2
    T = torch.stack([torch.randint(0, N, (E,), generator=gen), torch.randint(0, R0, (E,), generator=gen),
                     torch.randint(0, N, (E,), generator=gen)], dim=1)
    labels = torch.randint(0, ncls, (N,), generator=gen)
    train_idx = torch.arange(0, N, 2)
    for name, decomp in (("none", None), ("basis", {"type": "basis", "num_bases": 2}),
                         ("block", {"type": "block", "num_blocks": 2})):
        model = NodeClassifier(triples=T.tolist(), nnodes=N, nrel=R0, nfeat=None, nhid=4, nlayers=2, nclass=ncls,
                               edge_dropout=None, decomposition=decomp)
        seeded_params(model, gen)
        init = params_np(model)
        logits = model()
        loss = torch.nn.functional.cross_entropy(logits[train_idx], labels[train_idx])
        model.zero_grad()
        loss.backward()
        grads = {f"grad_{n}": p.grad.numpy().copy() for n, p in model.named_parameters()}
        opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=0.0)
        traj = []
        for _ in range(3):
            opt.zero_grad()
            l = torch.nn.functional.cross_entropy(model()[train_idx], labels[train_idx])
            l.backward()
            opt.step()
            traj.append(l.item())
        save("g4_model_nc_" + name, triples=T.numpy(), num_nodes=N, num_rels=R0, nclass=ncls, nhid=4,
             labels=labels.numpy(), train_idx=train_idx.numpy(), logits=logits.detach().numpy(),
             loss=loss.item(), adam_losses=np.array(traj), **init, **grads)
    # e-rgcn
    model = EmbeddingNodeClassifier(triples=T.tolist(), nnodes=N, nrel=R0, nfeat=None, nhid=4, nlayers=2,
                                    nclass=ncls, edge_dropout=None, decomposition=None, nemb=4)
    seeded_params(model, gen)
    init = params_np(model)
    logits = model()
    loss = torch.nn.functional.cross_entropy(logits[train_idx], labels[train_idx])
    model.zero_grad()
    loss.backward()
    grads = {f"grad_{n}": p.grad.numpy().copy() for n, p in model.named_parameters() if p.grad is not None}
    save("g4_model_enc", triples=T.numpy(), num_nodes=N, num_rels=R0, nclass=ncls, nemb=4,
         labels=labels.numpy(), train_idx=train_idx.numpy(), logits=logits.detach().numpy(), loss=loss.item(),
         **init, **grads)


# ------------------------------------------------------------------ G5
def g5():
    gen = torch.Generator().manual_seed(505)
    N, R0, d = 9, 4, 5
    nodes = torch.randn(N, d, generator=gen).requires_grad_(True)
    tr2 = torch.stack([torch.randint(0, N, (14,), generator=gen), torch.randint(0, R0, (14,), generator=gen),
                       torch.randint(0, N, (14,), generator=gen)], dim=1)
    tr3 = torch.stack([torch.randint(0, N, (3, 7), generator=gen), torch.randint(0, R0, (3, 7), generator=gen),
                       torch.randint(0, N, (3, 7), generator=gen)], dim=2)
    arrs = dict(nodes=nodes.detach().numpy(), triples2=tr2.numpy(), triples3=tr3.numpy(), num_nodes=N, num_rels=R0)
    for bias in (None, "normal"):
        dm = DistMult(R0, d, N, R0, w_init="standard-normal", w_gain=False, b_init=bias)
        seeded_params(dm, gen)
        tag = "b" if bias else "nb"
        for k, v in params_np(dm).items():
            arrs[f"{tag}_{k}"] = v
        for nm, tr in (("2", tr2), ("3", tr3)):
            sc = dm(tr, nodes)
            g = torch.randn(sc.shape, generator=gen)
            dm.zero_grad()
            nodes.grad = None
            sc.backward(g)
            arrs[f"{tag}_scores{nm}"] = sc.detach().numpy()
            arrs[f"{tag}_g{nm}"] = g.numpy()
            arrs[f"{tag}_grad_nodes{nm}"] = nodes.grad.numpy().copy()
            for n, p in dm.named_parameters():
                arrs[f"{tag}_grad_{n}{nm}"] = p.grad.numpy().copy()
        arrs[f"{tag}_penalty2"] = dm.s_penalty(tr2, nodes).item()
    save("g5_distmult", **arrs)


# ------------------------------------------------------------------ G6
def g6():
    torch.manual_seed(606)              # the layers keep their own xavier init: make it independent of what ran before
    gen = torch.Generator().manual_seed(606)
    N, R0, E, d = 2000, 10, 20000, 16
    s = torch.randint(0, N, (E,), generator=gen)
    # skew: a hub node receives many edges so long segments are exercised
    s[:1500] = 7
    T = torch.stack([s, torch.randint(0, R0, (E,), generator=gen), torch.randint(0, N, (E,), generator=gen)], dim=1)
    T[100:200] = T[0:100]  # duplicates
    Tp = add_inverse_and_self(T, N, R0)
    R = 2 * R0 + 1
    l1 = RelationalGraphConvolutionNC(triples=Tp, num_nodes=N, num_relations=R, in_features=d, out_features=d,
                                      vertical_stacking=False)
    l2 = RelationalGraphConvolutionNC(triples=Tp, num_nodes=N, num_relations=R, in_features=d, out_features=d,
                                      vertical_stacking=True)
    with torch.no_grad():
        l1.bias.copy_(torch.randn(d, generator=gen) * 0.1)
        l2.bias.copy_(torch.randn(d, generator=gen) * 0.1)
    X = torch.randn(N, d, generator=gen).requires_grad_(True)
    h = l1(X)
    out = l2(torch.relu(h))
    loss = out.pow(2).mean()
    loss.backward()
    save("g6_mid", triples=T.numpy().astype(np.int32), num_nodes=N, num_rels=R0, X=X.detach().numpy(),
         w1=l1.weights.detach().numpy(), b1=l1.bias.detach().numpy(), w2=l2.weights.detach().numpy(),
         b2=l2.bias.detach().numpy(), h=h.detach().numpy(), out=out.detach().numpy(), loss=loss.item(),
         grad_X=X.grad.numpy(), grad_w1=l1.weights.grad.numpy(), grad_b1=l1.bias.grad.numpy(),
         grad_w2=l2.weights.grad.numpy(), grad_b2=l2.bias.grad.numpy())


def _reference_misc():
    """utils/misc.py imports sacred (absent here) at module level only for create_experiment; stub it."""
    import types
    for name in ("sacred", "sacred.observers"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["sacred"].Experiment = object
    sys.modules["sacred.observers"].MongoObserver = object
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_utils_misc", os.path.join(REF, "utils", "misc.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def g7():
    misc = _reference_misc()
    gen = torch.Generator().manual_seed(7)
    # (a) LP encoder (basis) + DistMult without biases, float embeddings
    N, R0, d = 30, 4, 8
    def rand_triples(n):
        return torch.stack([torch.randint(0, N, (n,), generator=gen), torch.randint(0, R0, (n,), generator=gen),
                            torch.randint(0, N, (n,), generator=gen)], dim=1)
    train, valid, test = rand_triples(90), rand_triples(15), rand_triples(20)
    test[5] = train[3]                     # a test triple that is also a training triple
    test[6, :2] = test[7, :2]              # two test triples sharing (s, p): each filters the other's tail
    all_triples = torch.cat([train, valid, test]).tolist()
    true_triples = misc.generate_true_dict(all_triples)
    layer = RelationalGraphConvolutionLP(num_nodes=N, num_relations=2 * R0 + 1, in_features=d, out_features=d,
                                         edge_dropout={"general": 0.5, "self_loop": 0.2, "self_loop_type": "schlichtkrull-dropout"},
                                         decomposition={"type": "basis", "num_bases": 2}, w_init="glorot-normal",
                                         b_init="zeros")
    dm = DistMult(R0, d, N, R0, w_init="standard-normal")
    seeded_params(layer, gen)
    seeded_params(dm, gen)
    emb = torch.randn(N, d, generator=gen)
    layer.eval()

    class Model(torch.nn.Module):
        def forward(self, graph, triples):
            return dm(triples, layer(graph, torch.relu(emb))), 0
    res = {}
    with torch.no_grad():
        for tag, filt in (("filtered", True), ("raw", False)):
            mrr, hits, ranks = misc.evaluate(Model(), train, test, true_triples, N, batch_size=7, hits_at_k=[1, 3, 10],
                                             filter_candidates=filt, verbose=False)
            res.update({f"mrr_{tag}": mrr, f"hits_{tag}": np.asarray(hits), f"ranks_{tag}": np.asarray(ranks)})
        x = layer(train, torch.relu(emb))
    save("g7_eval_lp", num_nodes=N, num_rels=R0, train=train.numpy(), valid=valid.numpy(), test=test.numpy(),
         emb=emb.numpy(), nodes=x.numpy(), relations=dm.relations.detach().numpy(), batch_size=7,
         **{f"layer_{k}": v for k, v in params_np(layer).items()}, **res)

    # (b) ties: small-integer embeddings with repeated rows (all arithmetic exact in fp32), DistMult with biases
    N, R0, d = 24, 3, 6
    emb = torch.randint(-2, 3, (N, d), generator=gen).float()
    emb[8:16] = emb[0:8]                   # entity n+8 ties with entity n wherever their biases agree
    dm = DistMult(R0, d, N, R0, w_init="standard-normal", b_init="ones")
    with torch.no_grad():
        dm.relations.copy_(torch.randint(-2, 3, (R0, d), generator=gen).float())
        dm.sbias.copy_(torch.randint(0, 2, (N,), generator=gen).float())
        dm.obias.copy_(torch.randint(0, 2, (N,), generator=gen).float())
        dm.pbias.copy_(torch.randint(-1, 2, (R0,), generator=gen).float())
    def rand_triples(n):
        return torch.stack([torch.randint(0, N, (n,), generator=gen), torch.randint(0, R0, (n,), generator=gen),
                            torch.randint(0, N, (n,), generator=gen)], dim=1)
    known, test = rand_triples(60), rand_triples(25)
    true_triples = misc.generate_true_dict(torch.cat([known, known[:10], test]).tolist())   # duplicated entries too

    class Model2(torch.nn.Module):
        def forward(self, graph, triples):
            return dm(triples, emb), 0
    res = {}
    with torch.no_grad():
        for tag, filt in (("filtered", True), ("raw", False)):
            mrr, hits, ranks = misc.evaluate(Model2(), known, test, true_triples, N, batch_size=25, hits_at_k=[1, 3, 10],
                                             filter_candidates=filt, verbose=False)
            res.update({f"mrr_{tag}": mrr, f"hits_{tag}": np.asarray(hits), f"ranks_{tag}": np.asarray(ranks)})
        toscore = torch.cat([torch.arange(N).view(1, N, 1).expand(25, N, 1), test[:, 1:].view(25, 1, 2).expand(25, N, 2)], dim=2)
        head_scores = dm(toscore, emb)
    save("g7_eval_ties", num_nodes=N, num_rels=R0, known=known.numpy(), test=test.numpy(), nodes=emb.numpy(),
         relations=dm.relations.detach().numpy(), sbias=dm.sbias.detach().numpy(), pbias=dm.pbias.detach().numpy(),
         obias=dm.obias.detach().numpy(), head_scores=head_scores.numpy(), batch_size=25, **res)


def g8():
    """utils/misc.py edge_neighborhood under np.random.seed: the picks the oracle restatement must reproduce"""
    misc = _reference_misc()
    gen = torch.Generator().manual_seed(8)
    N, E = 14, 40
    T = torch.stack([torch.randint(0, N - 2, (E,), generator=gen), torch.randint(0, 3, (E,), generator=gen),
                     torch.randint(0, N - 2, (E,), generator=gen)], dim=1)     # nodes 12, 13 isolated
    T[3, 2] = T[3, 0]            # a self loop
    T[5] = T[4]                  # a duplicated edge
    T[30:36, 0] = N - 3          # a component that the sampler has to jump to
    T[30:36, 2] = N - 3
    entities = {f"e{i}": i for i in range(N)}
    res = {}
    for seed, size in ((123, 25), (7, 40)):
        np.random.seed(seed)
        picked = misc.edge_neighborhood(T.tolist(), sample_size=size, entities=entities)
        res[f"picked_seed{seed}"] = np.asarray(picked)
    save("g8_sampler", triples=T.numpy(), num_nodes=N, **res)


def g9():
    """utils/data.py load_link_prediction_data on a tiny text dataset (rdflib stubbed: only the module import needs it).
    The reference numbers nodes in set-iteration order, so the fixture stores label-level facts: the decoded train /
    test triples, the label sets and the number of distinct triples."""
    import importlib.util
    import tempfile
    import types
    sys.modules.setdefault("rdflib", types.ModuleType("rdflib"))
    sys.modules["rdflib"].URIRef = str
    spec = importlib.util.spec_from_file_location("ref_utils_data", os.path.join(REF, "utils", "data.py"))
    data = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(data)
    files = {
        "train": "a likes b\nb likes c\nc hates a\na likes b\nd  knows\ta\ne knows d\n",
        "valid": "a hates c\nb knows e\n",
        "test": "c likes d\nf knows a\nb likes c\n",
    }
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "data", "wn18"))
        for part, text in files.items():
            with open(os.path.join(tmp, "data", "wn18", part + ".txt"), "w") as f:
                f.write(text)
        data.locate_file = lambda rel: os.path.join(tmp, rel)
        for tag, kw in (("valid", {}), ("test", {"use_test_set": True}), ("limit", {"limit": 4})):
            (n2i, n), (r2i, r), train, test, all_triples = data.load_link_prediction_data("WN18", **kw)
            dec = lambda ts: np.asarray([[n[s_], r[p_], n[o_]] for s_, p_, o_ in ts])  # noqa: E731
            res.update({f"{tag}_train": dec(train), f"{tag}_test": dec(test), f"{tag}_nodes": np.asarray(sorted(n)),
                        f"{tag}_rels": np.asarray(sorted(r)), f"{tag}_all": np.asarray(sorted(dec(all_triples).tolist()))})
            assert sorted(n2i, key=n2i.get) == n and sorted(r2i, key=r2i.get) == r
    save("g9_lp_loader", **{f"file_{k}": np.asarray(v) for k, v in files.items()}, **res)


def _rdflib_standin():
    """rdflib (a dependency of the reference's node-classification loader, utils/data.py:1-5) is not installed in this image and
    cannot be (no network).  This stand-in provides exactly what the loader touches (utils/data.py:15-25 st(), :28-43
    add_neighbors(), :131-160): a Graph that is a SET of triples with .parse(file=, format='nt'), iteration and
    .triples((s|None, p|None, o|None)); URIRef / BNode / Literal terms with .n3().  Term parsing restates the W3C N-Triples
    grammar.  The G10 fixture keeps to terms whose rdflib .n3() is the input token itself (ASCII IRIs, blank nodes, plain /
    language-tagged / datatyped literals without escapes), so what G10 pins is the REFERENCE's loader logic around the parser:
    label category codes, validation split, two-hop pruning, node / relation inventories, the relation limit."""
    import re
    import types

    class URIRef(str):
        def n3(self):
            return f"<{self}>"

    class BNode(str):
        def n3(self):
            return f"_:{self}"

    class Literal(str):
        def __new__(cls, lex, suffix=""):
            obj = str.__new__(cls, lex + "\0" + suffix)     # identity = lexical form + language / datatype
            obj.lex, obj.suffix = lex, suffix
            return obj

        def n3(self):
            return f'"{self.lex}"{self.suffix}'

    term = r'(<[^>]*>|_:[A-Za-z0-9]+|"[^"\\]*"(?:@[A-Za-z]+(?:-[A-Za-z0-9]+)*|\^\^<[^>]*>)?)'
    line_re = re.compile(r"^\s*" + term + r"\s*" + term + r"\s*" + term + r"\s*\.\s*(#.*)?$")

    def make(tok):
        if tok.startswith("<"):
            return URIRef(tok[1:-1])
        if tok.startswith("_:"):
            return BNode(tok[2:])
        end = tok.rindex('"')
        return Literal(tok[1:end], tok[end + 1:])

    class Graph:
        def __init__(self):
            self._t = set()

        def parse(self, file=None, format=None):
            assert format == "nt"
            for raw in file:
                line = raw.decode("utf8") if isinstance(raw, bytes) else raw
                if not line.strip() or line.lstrip().startswith("#"):
                    continue
                m = line_re.match(line)
                assert m, line
                self._t.add(tuple(make(m.group(i)) for i in (1, 2, 3)))
            return self

        def __iter__(self):
            return iter(self._t)

        def __len__(self):
            return len(self._t)

        def triples(self, pattern):
            s, p, o = pattern
            for t in self._t:
                if (s is None or t[0] == s) and (p is None or t[1] == p) and (o is None or t[2] == o):
                    yield t

    mod = types.ModuleType("rdflib")
    mod.URIRef, mod.BNode, mod.Literal, mod.Graph = URIRef, BNode, Literal, Graph
    mod.util = types.SimpleNamespace(guess_format=lambda f: "nt")
    return mod


G10_NT = """# tiny AIFB-layout graph for G10
<http://ex.org/a> <http://ex.org/p> <http://ex.org/b> .
<http://ex.org/a> <http://ex.org/p> <http://ex.org/b> .
<http://ex.org/b> <http://ex.org/p> <http://ex.org/c> .
<http://ex.org/c> <http://ex.org/p> <http://ex.org/a> .
<http://ex.org/d> <http://ex.org/p> <http://ex.org/a> .
<http://ex.org/b> <http://ex.org/q> _:blank1 .
_:blank1 <http://ex.org/q> "a literal with spaces" .
<http://ex.org/b> <http://ex.org/q> "Bob"@en-GB .
<http://ex.org/b> <http://ex.org/age> "42"^^<http://www.w3.org/2001/XMLSchema#integer> .
<http://ex.org/c> <http://ex.org/age> "42" .
<http://ex.org/far> <http://ex.org/p> <http://ex.org/farther> .
<http://ex.org/farther> <http://ex.org/p> <http://ex.org/farthest> .
<http://ex.org/farthest> <http://ex.org/r> <http://ex.org/beyond> .
<http://ex.org/beyond> <http://ex.org/r> <http://ex.org/outer> .
"""
G10_TRAIN = [("http://ex.org/a", "x"), ("http://ex.org/b", "y"), ("http://ex.org/c", "x"), ("http://ex.org/far", "z"),
             ("http://ex.org/d", "y"), ("http://ex.org/a", "x")]
G10_TEST = [("http://ex.org/c", "y"), ("http://ex.org/b", "x")]


def g10():
    """utils/data.py:50-186 load_node_classification_data -- the REFERENCE's loader, run on a tiny AIFB-layout dataset with the
    rdflib stand-in above.  The reference numbers nodes and relations in set / dict iteration order, so the fixture stores
    label-level facts: decoded edges (sorted), node and relation label sets, the train / test dictionaries."""
    import gzip
    import importlib.util
    import tempfile
    sys.modules["rdflib"] = _rdflib_standin()
    spec = importlib.util.spec_from_file_location("ref_utils_data_nc", os.path.join(REF, "utils", "data.py"))
    data = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(data)
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        d = os.path.join(tmp, "data", "aifb")
        os.makedirs(d)
        with gzip.open(os.path.join(d, "aifb_stripped.nt.gz"), "wb") as f:
            f.write(G10_NT.encode("utf8"))
        for fname, rows in (("trainingSet.tsv", G10_TRAIN), ("testSet.tsv", G10_TEST)):
            with open(os.path.join(d, fname), "w") as f:
                f.write("id\tperson\tlabel_affiliation\n" + "".join(f"{i}\t{n}\t{c}\n" for i, (n, c) in enumerate(rows)))
        data.locate_file = lambda rel: os.path.join(tmp, rel)
        for tag, kw in (("valid", {}), ("test", {"use_test_set": True}), ("prune", {"use_test_set": True, "prune": True}),
                        ("limit", {"use_test_set": True, "limit": 1}), ("valprop", {"val_prop": 0.5})):
            edges, (n2i, i2n), (r2i, i2r), train, test = data.load_node_classification_data("AIFB", enable_cache=False, **kw)
            assert [n2i[x] for x in i2n] == list(range(len(i2n))) and [r2i[x] for x in i2r] == list(range(len(i2r)))
            dec = sorted([i2n[s_], i2r[p_], i2n[o_]] for s_, p_, o_ in edges)
            res.update({f"{tag}_edges": np.asarray(dec), f"{tag}_nodes": np.asarray(sorted(i2n)),
                        f"{tag}_rels": np.asarray(i2r if "limit" in kw else sorted(i2r)),
                        f"{tag}_train": np.asarray(sorted((k, str(int(v))) for k, v in train.items())),
                        f"{tag}_test": np.asarray(sorted((k, str(int(v))) for k, v in test.items()))})
    save("g10_nc_loader", file_nt=np.asarray(G10_NT), file_train=np.asarray([list(r) for r in G10_TRAIN]),
         file_test=np.asarray([list(r) for r in G10_TEST]), **res)


def g11():
    """Thousands of parallel edges (the same triple repeated): every output element of torch.sparse.mm is ONE chain of that many fp32
    additions of equal terms, whose rounding residue repeats at every step.  The fixture records what the reference returns; the tests
    measure how far that -- and the HIP path -- is from the oracle's doubles (tests/test_oracle_golden.py, tests/test_gpu_parity.py)."""
    cases = (("wide", 1, 1, 9000, 3, 100, None), ("basis", 2, 1, 20000, 100, 100, {"type": "basis", "num_bases": 2}),
             ("featureless", 2, 1, 20000, None, 100, None))
    for tag, N, R0, E, d_in, d_out, decomp in cases:
        gen = torch.Generator().manual_seed(1100 + E + d_out)
        T = torch.stack([torch.randint(0, N, (E,), generator=gen), torch.randint(0, R0, (E,), generator=gen),
                         torch.randint(0, N, (E,), generator=gen)], dim=1)
        Tp = add_inverse_and_self(T, N, R0)
        layer = RelationalGraphConvolutionNC(triples=Tp, num_nodes=N, num_relations=2 * R0 + 1, in_features=d_in, out_features=d_out,
                                             decomposition=decomp, vertical_stacking=False)
        seeded_params(layer, gen)
        X = None if d_in is None else torch.randn(N, d_in, generator=gen).requires_grad_(True)
        out = layer(X) if X is not None else layer()
        g = torch.randn(out.shape, generator=gen)
        grads_of(layer, out, g, extra=() if X is None else (X,))
        arrs = {"triples": T.numpy().astype(np.int32), "num_nodes": N, "num_rels": R0, "g": g.numpy(), "out": out.detach().numpy()}
        if X is not None:
            arrs.update(X=X.detach().numpy(), grad_X=X.grad.numpy())
        for n, p_ in layer.named_parameters():
            arrs[f"param_{n}"] = p_.detach().numpy()
            arrs[f"grad_{n}"] = p_.grad.numpy()
        save(f"g11_dup_{tag}", **arrs)


if __name__ == "__main__":
    torch.set_num_threads(4)
    only = sys.argv[1:]
    for fn in (g1, g2, g3, g4, g5, g6, g7, g8, g9, g10, g11):
        if only and fn.__name__ not in only:
            continue
        fn()
