"""Sampling and ranking evaluation for link prediction -- counterpart of the reference's utils/misc.py:29-189,
same function names, arguments and return values.

evaluate(): the reference expands every batch of test triples to a [bn, N, 3] candidate tensor and calls the whole
model on it, which re-runs the encoder for each of the ~2 * len(test) / batch_size batches (misc.py:78-85).  Here the
graph is encoded ONCE and the candidates are scored by the MFMA kernel behind `_native.distmult_score_all`
("encode once, score many"); filtering and rank counting are HIP kernels too.  Ranks are defined
exactly as in the reference: known true completions other than the target are set to -inf (misc.py:40-58), and the
target sits halfway down its ties (misc.py:93-101).
"""
import random

import numpy as np
import torch

from torch_rgcn import _native
from torch_rgcn.layers import DistMult

_SCORE_BYTES = 1 << 30     # score-matrix budget per chunk of queries


def create_experiment(name='exp', database=None):
    """sacred Experiment for the REFERENCE's own experiment scripts (misc.py:10-24), so that they keep running with this
    package first on the path; sacred is imported here, not at module level -- this package's experiments/ do not use it.
    A MongoDB observer is attached only when MONGO_DB_USER / MONGO_DB_PASS / MONGO_DB_HOST are all set."""
    try:
        from sacred import Experiment
    except ImportError as exc:
        raise ImportError("sacred is not installed: run this package's experiments/*.py (argparse + YAML) instead") from exc
    import os
    ex = Experiment(name)
    user, password, host = (os.environ.get(k) for k in ("MONGO_DB_USER", "MONGO_DB_PASS", "MONGO_DB_HOST"))
    if user and password and host:
        from sacred.observers import MongoObserver
        ex.observers.append(MongoObserver(url=f"mongodb+srv://{user}:{password}@{host}", db_name=database))
    return ex


def generate_true_dict(all_triples):
    """(p, o) -> known heads and (s, p) -> known tails, duplicates kept (misc.py:29-38)"""
    heads, tails = {}, {}
    for s, p, o in (all_triples.tolist() if hasattr(all_triples, "tolist") else all_triples):
        heads.setdefault((p, o), []).append(s)
        tails.setdefault((s, p), []).append(o)
    return heads, tails


class _FilterIndex:
    """The true-triple dictionaries flattened once into two sorted key -> candidates tables (numpy), so that the filter
    lists of a whole batch of queries are a searchsorted + repeat instead of a Python loop over the queries."""

    def __init__(self, true_triples, num_nodes):
        self.n = int(num_nodes)
        self.tables = []
        for table, key_of in ((true_triples[0], lambda k: k[0] * self.n + k[1]),      # heads: (p, o) -> [s]
                              (true_triples[1], lambda k: k[1] * self.n + k[0])):     # tails: (s, p) -> [o]
            keys = np.fromiter((key_of(k) for k in table), np.int64, len(table))
            lens = np.fromiter((len(v) for v in table.values()), np.int64, len(table))
            vals = np.fromiter((x for v in table.values() for x in v), np.int64, int(lens.sum()))
            order = np.argsort(keys, kind="stable")
            ptr = np.zeros(len(keys) + 1, np.int64)
            np.cumsum(lens[order], out=ptr[1:])
            starts = np.concatenate([[0], np.cumsum(lens)])[:-1][order]
            gather = np.repeat(starts - ptr[:-1], lens[order]) + np.arange(int(lens.sum()))
            self.tables.append((keys[order], ptr, vals[gather]))

    def lists(self, batch, head):
        """-> (rows, cols) int32 arrays: for query i every known completion except its own target"""
        b = np.asarray(batch, np.int64)
        keys, ptr, vals = self.tables[0 if head else 1]
        q = b[:, 1] * self.n + (b[:, 2] if head else b[:, 0])
        pos = np.searchsorted(keys, q)
        pos_c = np.minimum(pos, max(len(keys) - 1, 0))
        hit = (pos < len(keys)) & (keys[pos_c] == q) if len(keys) else np.zeros(len(q), bool)
        lo = np.where(hit, ptr[pos_c], 0)
        cnt = np.where(hit, ptr[pos_c + 1] - ptr[pos_c], 0) if len(keys) else np.zeros(len(q), np.int64)
        rows = np.repeat(np.arange(len(q)), cnt)
        cols = vals[np.repeat(lo - np.concatenate([[0], np.cumsum(cnt)])[:-1], cnt) + np.arange(int(cnt.sum()))]
        keep = cols != (b[:, 0] if head else b[:, 2])[rows]
        return rows[keep].astype(np.int32), cols[keep].astype(np.int32)


_FILTER_CACHE = []      # [(true_triples object, num_nodes, _FilterIndex)]: one entry per experiment in practice


def _filter_index(true_triples, num_nodes):
    for obj, n, idx in _FILTER_CACHE:
        if obj is true_triples and n == num_nodes:
            return idx
    idx = _FilterIndex(true_triples, num_nodes)
    del _FILTER_CACHE[:-3]
    _FILTER_CACHE.append((true_triples, num_nodes, idx))
    return idx


def filter_scores(scores, batch, true_triples, head=True):
    """scores of known true triples that are not the target -> -inf, in place (misc.py:40-58)"""
    rows, cols = _filter_index(true_triples, scores.shape[1]).lists(batch.cpu().numpy(), head)
    if len(rows):   # (the reference indexes an empty tensor and raises here)
        _native.rank_filter(scores, torch.from_numpy(rows).to(scores.device), torch.from_numpy(cols).to(scores.device))


def _rank_chunk(scores, batch, true_triples, head, filter_candidates):
    if filter_candidates:
        filter_scores(scores, batch, true_triples, head=head)
    raw, ties = _native.rank_count(scores, batch, head)
    return (raw + (ties - 1) // 2 + 1).tolist()


@torch.no_grad()
def evaluate(model, graph, test_set, true_triples, num_nodes, batch_size=16, hits_at_k=[1, 3, 10],
             filter_candidates=True, verbose=True):
    """(mrr, hits tuple, ranks): head queries for the whole test set first, then tail queries (misc.py:60-110).

    `batch_size` only bounds the score matrix of models without an `encode`/DistMult pair; the fast path scores as many
    queries at once as fit in 1 GiB -- ranks do not depend on the batching."""
    device = next(model.parameters()).device
    test_set = torch.as_tensor(test_set, dtype=torch.long)
    decoder = getattr(model, "scoring_function", None)
    fast = hasattr(model, "encode") and isinstance(decoder, DistMult)
    if fast:
        x = model.encode(graph).contiguous()
        assert x.shape[0] == num_nodes, "num_nodes differs from the encoder output"
        batch_size = max(batch_size, min(len(test_set), max(64, _SCORE_BYTES // (4 * num_nodes))))
    ranks = []
    for head in (True, False):
        for fr in range(0, len(test_set), batch_size):
            batch = test_set[fr:fr + batch_size].to(device).contiguous()
            bn = batch.shape[0]
            if fast:
                scores = _native.distmult_score_all(batch, head, x, decoder.relations.detach(),
                                                    *((decoder.sbias.detach(), decoder.pbias.detach(), decoder.obias.detach())
                                                      if decoder.b_init else ()))
            else:
                ar = torch.arange(num_nodes, device=device).view(1, num_nodes, 1).expand(bn, num_nodes, 1)
                bexp = (batch[:, 1:] if head else batch[:, :2]).view(bn, 1, 2).expand(bn, num_nodes, 2)
                scores, _ = model(graph, torch.cat([ar, bexp] if head else [bexp, ar], dim=2))
                scores = scores.float().contiguous()
            assert scores.shape == (bn, num_nodes)
            ranks.extend(_rank_chunk(scores, batch, true_triples, head, filter_candidates))
        if verbose:
            print(f"  ranked {len(test_set)} {'head' if head else 'tail'} queries")
    mrr = sum(1.0 / r for r in ranks) / len(ranks)
    hits = tuple(sum(1.0 if r <= k else 0.0 for r in ranks) / len(ranks) for k in hits_at_k)
    return mrr, hits, ranks


def select_sampling(method):
    method = method.lower()
    if method == 'uniform':
        return uniform_sampling
    if method == 'edge-neighborhood':
        return edge_neighborhood
    raise NotImplementedError(f'{method} sampling method has not been implemented!')


def uniform_sampling(graph, sample_size=30000, entities=None, train_triplets=None):
    """sample_size triples without replacement (misc.py:120-122); an ndarray / tensor of triples is indexed, not copied"""
    if isinstance(graph, list):
        return random.sample(graph, sample_size)
    return graph[random.sample(range(len(graph)), sample_size)]


def edge_neighborhood(train_triples, sample_size=30000, entities=None, seed=None):
    """Edge-neighbourhood sampling (misc.py:125-172): grow the sample along edges of already-visited vertices, a vertex
    drawn with probability proportional to its number of still-unpicked incident edges.

    The reference rebuilds an O(N) probability vector and calls np.random.choice for every one of the sample_size
    draws (O(sample_size * N)); the native sampler keeps the weights in a Fenwick tree (O(log N) per draw).  Same
    distribution, its own random stream (`seed`; drawn from Python's `random` when None)."""
    triples = train_triples if isinstance(train_triples, np.ndarray) and train_triples.dtype == np.int64 else \
        np.asarray(train_triples, dtype=np.int64)
    triples = np.ascontiguousarray(triples.reshape(-1, 3))
    num_nodes = len(entities) if entities is not None else int(max(triples[:, 0].max(), triples[:, 2].max())) + 1
    if seed is None:
        seed = random.getrandbits(63)
    picked = _native.edge_neighborhood_host(triples, num_nodes, sample_size, seed)
    rows = triples[picked]
    return [train_triples[e] for e in picked.tolist()] if isinstance(train_triples, list) else rows


def negative_sampling(batch, num_nodes, head_corrupt_prob, device='cpu', generator=None):
    """Corrupt the head (probability head_corrupt_prob) or else the tail of every triple of `batch` [bs, ns, 3], in
    place; returns the [bs * ns, 3] view (misc.py:174-189).  `generator` (extension): a torch.Generator to draw from
    instead of the device's default one."""
    bs, ns, _ = batch.size()
    rows = batch.view(bs * ns, 3)
    column = torch.where(torch.rand(bs * ns, device=device, generator=generator) < head_corrupt_prob, 0, 2)     # 0: head, 2: tail
    rows[torch.arange(bs * ns, device=device), column] = torch.randint(0, num_nodes, (bs * ns,), device=device,
                                                                       generator=generator)
    return rows
