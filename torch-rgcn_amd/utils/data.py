"""Dataset access for the experiment scripts -- counterpart of the reference's utils/data.py:50-256, same function
names, arguments and return values.

The reference parses the node-classification graphs with rdflib and downloads AIFB / MUTAG / BGS / AM / WN18 / FB15k
with get_data.sh; neither the package nor the network exists here.  So:
  * the files are looked up under `data/<name>/...` (same relative names as the reference) below the directory given
    by the `directory` argument, $RGCN_DATA, or this package's root;
  * N-Triples are read by the line parser below (W3C N-Triples grammar: IRIs, blank nodes, literals with language tag or
    datatype, comments) instead of rdflib;
  * when a dataset's files are absent, a dataset-SHAPED synthetic graph is produced (same N, R, E, class and label
    counts as SURVEY.md 8(d)), deterministic through splitmix64, so that the experiments and benchmarks still run.
Node and relation numbering: the reference numbers them in Python-set iteration order (different on every run);
here it is first-seen order, i.e. the same graph up to a relabelling.
"""
import gzip
import os
import re
from collections import Counter

import numpy as np

S = os.sep

SHAPES = {  # name: (nodes, base relations, triples, classes, labelled train, labelled test)
    "aifb": (8285, 45, 29043, 4, 140, 36),
    "mutag": (23644, 23, 74227, 2, 272, 68),
    "bgs": (333845, 103, 916199, 2, 117, 29),
    "am": (1666764, 133, 5988321, 11, 802, 198),
    "wn18": (40943, 18, 141442, 0, 0, 0),
    "wn18rr": (40943, 11, 86835, 0, 0, 0),
    "fb15k": (14951, 1345, 483142, 0, 0, 0),
    "fb15k-237": (14541, 237, 272115, 0, 0, 0),
    "fb-toy": (280, 112, 4565, 0, 0, 0),
}

NC_FILES = {  # name: (graph, training labels, test labels, label column, node column)   utils/data.py:80-107
    "aifb": ("aifb_stripped.nt.gz", "trainingSet.tsv", "testSet.tsv", "label_affiliation", "person"),
    "am": ("am_stripped.nt.gz", "trainingSet.tsv", "testSet.tsv", "label_cateogory", "proxy"),
    "bgs": ("bgs_stripped.nt.gz", "trainingSet(lith).tsv", "testSet(lith).tsv", "label_lithogenesis", "rock"),
    "mutag": ("mutag_stripped.nt.gz", "trainingSet.tsv", "testSet.tsv", "label_mutagenic", "bond"),
}
LP_DIRS = {"fb15k": "fb15k", "fb15k-237": "fB15k-237", "fb-toy": "fb-toy", "wn18": "wn18", "wn18rr": "wn18rr"}


def locate_file(filepath, directory=None):
    root = directory or os.environ.get("RGCN_DATA") or os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
    return os.path.join(root, filepath.lstrip("/"))


# ------------------------------------------------------------------ synthetic stand-ins
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


def _labels_of(n, prefix):
    i2x = [f"{prefix}{i}" for i in range(n)]
    return {x: i for i, x in enumerate(i2x)}, i2x


# ------------------------------------------------------------------ N-Triples
_IRI = r'<([^<>"{}|^`\\\x00-\x20]*(?:\\[uU][0-9A-Fa-f]+[^<>"{}|^`\\\x00-\x20]*)*)>'
_BNODE = r'(_:[^\s]+?)'
_LITERAL = r'("(?:[^"\\\n\r]|\\.)*"(?:\^\^<[^<>]*>|@[A-Za-z]+(?:-[A-Za-z0-9]+)*)?)'
_TRIPLE = re.compile(rf'^\s*(?:{_IRI}|{_BNODE})\s*{_IRI}\s*(?:{_IRI}|{_BNODE}|{_LITERAL})\s*\.\s*(?:#.*)?$')
_UCHAR = re.compile(r'\\u([0-9A-Fa-f]{4})|\\U([0-9A-Fa-f]{8})')


def _iri(text):
    return _UCHAR.sub(lambda m: chr(int(m.group(1) or m.group(2), 16)), text) if "\\" in text else text


def parse_ntriples(lines):
    """-> list of distinct (s, p, o) label triples in first-seen order.  Labels follow the reference's `st()`
    (utils/data.py:15-25): the bare IRI for IRIs, the N-Triples token itself for blank nodes and literals."""
    seen, out = set(), []
    for ln, line in enumerate(lines, 1):
        if isinstance(line, bytes):
            line = line.decode("utf8")
        if not line.strip() or line.lstrip().startswith("#"):
            continue
        m = _TRIPLE.match(line)
        if m is None:
            raise ValueError(f"N-Triples syntax error on line {ln}: {line.strip()[:80]!r}")
        s_iri, s_b, p, o_iri, o_b, o_lit = m.groups()
        t = (_iri(s_iri) if s_iri is not None else s_b, _iri(p),
             _iri(o_iri) if o_iri is not None else (o_b if o_b is not None else o_lit))
        if t not in seen:       # an rdflib Graph is a set of triples
            seen.add(t)
            out.append(t)
    return out


def _prune(triples, targets, depth=2):
    """triples within `depth` hops of a labelled node, both directions (add_neighbors, utils/data.py:27-41)"""
    outgoing, incoming = {}, {}
    for i, (s, _, o) in enumerate(triples):
        outgoing.setdefault(s, []).append(i)
        incoming.setdefault(o, []).append(i)
    keep, frontier = set(), set(targets)
    for _ in range(depth):
        nxt = set()
        for node in frontier:
            for i in outgoing.get(node, ()):
                keep.add(i)
                nxt.add(triples[i][2])
            for i in incoming.get(node, ()):
                keep.add(i)
                nxt.add(triples[i][0])
        frontier = nxt
    return [triples[i] for i in sorted(keep)]


def load_strings(file):
    """whitespace-separated string triples, one per line (utils/data.py:44-47)"""
    with open(file, "r") as f:
        return [line.split() for line in f]


# ------------------------------------------------------------------ node classification
def _label_codes(frame, label_header, nodes_header):
    codes = frame[label_header].astype("category").cat.codes        # codes in sorted-category order, per split (as upstream)
    return {node: int(code) for node, code in zip(frame[nodes_header].values, codes)}


def _want_synthetic(synthetic, what, path):
    """Dataset files are missing: a dataset-SHAPED random graph is only handed out when the caller asked for it
    (synthetic=True / RGCN_SYNTHETIC=1 / the experiments' --synthetic); otherwise fail like the reference does."""
    if synthetic is None:
        synthetic = os.environ.get("RGCN_SYNTHETIC", "0") == "1"
    if not synthetic:
        raise FileNotFoundError(f"{what}: {path} not found (pass synthetic=True / --synthetic / RGCN_SYNTHETIC=1 to train on a "
                                "random graph with the dataset's node, relation and edge counts instead)")
    import warnings
    warnings.warn(f"{what}: files not found, using a SYNTHETIC dataset-shaped random graph -- accuracies / MRR are meaningless",
                  stacklevel=3)


def load_node_classification_data(name, use_test_set=False, limit=None, enable_cache=True, val_prop=0.4, prune=False,
                                  directory=None, seed=0, synthetic=None):
    """-> edges [[s, p, o], ...], (n2i, i2n), (r2i, i2r), train {node label: class}, test {node label: class}
    (utils/data.py:50-200; `enable_cache` is accepted and ignored: parsing takes seconds, nothing is pickled)."""
    REST, INV = ".rest", "inv."
    key = name.lower()
    if key not in NC_FILES:
        raise ValueError(f"Could not find '{name}' dataset")
    graph_file, train_file, test_file, label_header, nodes_header = NC_FILES[key]
    graph_path = locate_file(f"data{S}{key}{S}{graph_file}", directory)
    if not os.path.isfile(graph_path):
        _want_synthetic(synthetic, f"node classification dataset '{name}'", graph_path)
        return _synthetic_node_classification(key, use_test_set, val_prop, seed)
    import pandas as pd
    labels_train = pd.read_csv(locate_file(f"data{S}{key}{S}{train_file}", directory), sep="\t", encoding="utf8")
    if use_test_set:
        labels_test = pd.read_csv(locate_file(f"data{S}{key}{S}{test_file}", directory), sep="\t", encoding="utf8")
    else:
        pivot = int(len(labels_train) * val_prop)
        labels_test, labels_train = labels_train[:pivot], labels_train[pivot:]
    train = _label_codes(labels_train, label_header, nodes_header)
    test = _label_codes(labels_test, label_header, nodes_header)
    opener = gzip.open if graph_path.endswith(".gz") else open
    with opener(graph_path, "rb") as f:
        triples = parse_ntriples(f)
    if prune:
        triples = _prune(triples, list(train) + list(test), depth=2)
    n2i, relations = {}, Counter()
    for s, p, o in triples:
        n2i.setdefault(s, len(n2i))
        n2i.setdefault(o, len(n2i))
        relations[p] += 1
    i2n = list(n2i)
    i2r = ([r for r, _ in relations.most_common(limit)] + [REST, INV + REST]) if limit is not None else list(relations)
    r2i = {r: i for i, r in enumerate(i2r)}
    edges = [[n2i[s], r2i[p] if p in r2i else r2i[REST], n2i[o]] for s, p, o in triples]
    return edges, (n2i, i2n), (r2i, i2r), train, test


def _synthetic_node_classification(key, use_test_set, val_prop, seed):
    n, r, e, c, ntr, nte = SHAPES[key]
    t = synthetic_triples(n, r, e, seed)
    rng = np.random.default_rng(seed)
    nodes = rng.permutation(n)[: ntr + nte]
    first_rel = np.zeros(n, np.int64)      # labels correlated with the graph, so that training has something to learn
    first_rel[t[::-1, 0]] = t[::-1, 1]
    y = first_rel[nodes] % c
    (n2i, i2n), (r2i, i2r) = _labels_of(n, "n"), _labels_of(r, "r")
    if use_test_set:
        tr_nodes, te_nodes, tr_y, te_y = nodes[:ntr], nodes[ntr:], y[:ntr], y[ntr:]
    else:
        pivot = int(ntr * val_prop)
        tr_nodes, te_nodes, tr_y, te_y = nodes[pivot:ntr], nodes[:pivot], y[pivot:ntr], y[:pivot]
    train = {i2n[v]: int(k) for v, k in zip(tr_nodes, tr_y)}
    test = {i2n[v]: int(k) for v, k in zip(te_nodes, te_y)}
    return t.tolist(), (n2i, i2n), (r2i, i2r), train, test


# ------------------------------------------------------------------ link prediction
def load_link_prediction_data(name, use_test_set=False, limit=None, directory=None, seed=0, synthetic=None):
    """-> (n2i, nodes), (r2i, relations), train [[s, p, o], ...], test [[s, p, o], ...], all_triples {(s, p, o), ...}
    (utils/data.py:202-256: the validation file is the test set unless `use_test_set`; `limit` keeps the first triples)."""
    key = name.lower()
    if key not in LP_DIRS:
        raise ValueError(f"Could not find '{name}' dataset")
    paths = [locate_file(f"data{S}{LP_DIRS[key]}{S}{part}.txt", directory) for part in ("train", "valid", "test")]
    if not all(os.path.isfile(p) for p in paths):
        _want_synthetic(synthetic, f"link prediction dataset '{name}'", next(p for p in paths if not os.path.isfile(p)))
        return _synthetic_link_prediction(key, use_test_set, limit, seed)
    train, val, test = (load_strings(p) for p in paths)
    if not use_test_set:
        test = val
    if limit:
        train, test = train[:limit], test[:limit]
    n2i, r2i = {}, {}
    for s, p, o in train + val + test:
        n2i.setdefault(s, len(n2i))
        r2i.setdefault(p, len(r2i))
        n2i.setdefault(o, len(n2i))
    all_triples = {(n2i[s], r2i[p], n2i[o]) for s, p, o in train + val + test}
    train = [[n2i[s], r2i[p], n2i[o]] for s, p, o in train]
    test = [[n2i[s], r2i[p], n2i[o]] for s, p, o in test]
    return (n2i, list(n2i)), (r2i, list(r2i)), train, test, all_triples


def _synthetic_link_prediction(key, use_test_set, limit, seed):
    n, r, e, *_ = SHAPES[key]
    t = np.unique(synthetic_triples(n, r, e + e // 10, seed), axis=0)
    t = t[np.random.default_rng(seed).permutation(len(t))]
    k = max(1, len(t) // 30)
    val, test, train = t[:k], t[k:2 * k], t[2 * k:]
    if not use_test_set:
        test = val
    if limit:
        train, test = train[:limit], test[:limit]
    all_triples = {tuple(x) for x in np.concatenate([train, val, test]).tolist()}
    (n2i, i2n), (r2i, i2r) = _labels_of(n, "n"), _labels_of(r, "r")
    return (n2i, i2n), (r2i, i2r), train.tolist(), test.tolist(), all_triples
