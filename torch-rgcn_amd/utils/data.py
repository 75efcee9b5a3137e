"""Dataset access for the experiment scripts.

The reference parses RDF with rdflib and downloads AIFB / MUTAG / BGS / AM / WN18 / FB15k from Dropbox
(utils/data.py:50-256, get_data.sh) -- neither the package nor the network is available here, so this module reads
plain TSV files when a directory is given and otherwise produces dataset-SHAPED synthetic graphs (same N, R, E,
class / label counts as SURVEY.md 8(d)), deterministic through splitmix64.

TSV layout:  <dir>/triples.tsv  (s \t p \t o, integer ids)  [+ train.tsv / valid.tsv / test.tsv for link prediction,
             labels_train.tsv / labels_test.tsv (node \t class) for node classification]
"""
import os

import numpy as np

SHAPES = {  # name: (nodes, base relations, triples, classes, labelled train, labelled test)
    "aifb": (8285, 45, 29043, 4, 140, 36),
    "mutag": (23644, 23, 74227, 2, 272, 68),
    "bgs": (333845, 103, 916199, 2, 117, 29),
    "am": (1666764, 133, 5988321, 11, 802, 198),
    "wn18": (40943, 18, 141442, 0, 0, 0),
    "fb-toy": (280, 112, 4565, 0, 0, 0),
}


def _splitmix(seed, n):
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def synthetic_triples(n, r, e, seed=0):
    z = _splitmix(seed, 3 * e).reshape(e, 3)
    return np.stack([z[:, 0] % np.uint64(n), z[:, 1] % np.uint64(r), z[:, 2] % np.uint64(n)], axis=1).astype(np.int64)


def _read_tsv(path, cols):
    return np.loadtxt(path, dtype=np.int64, delimiter="\t", ndmin=2)[:, :cols]


def load_node_classification_data(name, directory=None, seed=0):
    """-> triples [E,3], (num_nodes, num_rels, num_classes), train_idx, train_y, test_idx, test_y"""
    if directory and os.path.isdir(directory):
        t = _read_tsv(os.path.join(directory, "triples.tsv"), 3)
        tr = _read_tsv(os.path.join(directory, "labels_train.tsv"), 2)
        te = _read_tsv(os.path.join(directory, "labels_test.tsv"), 2)
        n = int(max(t[:, 0].max(), t[:, 2].max(), tr[:, 0].max(), te[:, 0].max())) + 1
        return t, (n, int(t[:, 1].max()) + 1, int(max(tr[:, 1].max(), te[:, 1].max())) + 1), tr[:, 0], tr[:, 1], te[:, 0], te[:, 1]
    n, r, e, c, ntr, nte = SHAPES[name.lower()]
    t = synthetic_triples(n, r, e, seed)
    rng = np.random.default_rng(seed)
    nodes = rng.permutation(n)[: ntr + nte]
    # labels correlated with the graph so that training has something to learn: class = f(first relation seen)
    first_rel = np.zeros(n, np.int64)
    first_rel[t[::-1, 0]] = t[::-1, 1]
    y = first_rel[nodes] % c
    return t, (n, r, c), nodes[:ntr], y[:ntr], nodes[ntr:], y[ntr:]


def load_link_prediction_data(name, directory=None, seed=0):
    """-> (num_nodes, num_rels), train, valid, test triples"""
    if directory and os.path.isdir(directory):
        tr, va, te = (_read_tsv(os.path.join(directory, f + ".tsv"), 3) for f in ("train", "valid", "test"))
        allt = np.concatenate([tr, va, te])
        return (int(max(allt[:, 0].max(), allt[:, 2].max())) + 1, int(allt[:, 1].max()) + 1), tr, va, te
    n, r, e, *_ = SHAPES[name.lower()]
    t = np.unique(synthetic_triples(n, r, e + e // 10, seed), axis=0)
    rng = np.random.default_rng(seed)
    t = t[rng.permutation(len(t))]
    k = max(1, len(t) // 30)
    return (n, r), t[2 * k:], t[:k], t[k:2 * k]
