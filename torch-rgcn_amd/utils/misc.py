"""Sampling and ranking evaluation for link prediction (counterpart of the reference's utils/misc.py:60-189).

evaluate(): the reference re-runs the whole encoder for every batch of 32 test triples (misc.py:86); here the graph is
encoded ONCE and every (s, p, ?) / (?, p, o) query is scored against all entities with the DistMult kernel
("encode once, score many").  Filtered ranks as in the reference: known true triples are
removed from the candidate list before ranking.
"""
import numpy as np
import torch


def negative_sampling(positive, num_nodes, neg_sample_rate, rng=None):
    """corrupt head or tail of each positive `neg_sample_rate` times -> (triples [B*(1+rate), 3], labels)"""
    rng = rng or np.random.default_rng()
    b = len(positive)
    neg = np.tile(positive, (neg_sample_rate, 1))
    corrupt = rng.integers(0, num_nodes, size=len(neg))
    head = rng.random(len(neg)) < 0.5
    neg[head, 0] = corrupt[head]
    neg[~head, 2] = corrupt[~head]
    labels = np.concatenate([np.ones(b, np.float32), np.zeros(len(neg), np.float32)])
    return np.concatenate([positive, neg]), labels


def sample_edges(train, batch_size, rng=None):
    """uniform edge sample (the reference's default; its neighbourhood sampler is O(batch * N) numpy)"""
    rng = rng or np.random.default_rng()
    return train[rng.choice(len(train), size=min(batch_size, len(train)), replace=False)]


def generate_true_dict(all_triples):
    heads, tails = {}, {}
    for s, p, o in np.asarray(all_triples).tolist():
        heads.setdefault((p, o), []).append(s)
        tails.setdefault((s, p), []).append(o)
    return heads, tails


@torch.no_grad()
def evaluate(model, graph, test, true_heads, true_tails, num_nodes, batch_size=64, hits_at=(1, 3, 10), filtered=True):
    """MRR and hits@k over head and tail queries; one encoder pass."""
    model.eval()
    device = next(model.parameters()).device
    x = model.encode(graph)
    test = np.asarray(test)
    ranks = []
    cand = torch.arange(num_nodes, device=device)
    for head_query in (True, False):
        for a in range(0, len(test), batch_size):
            b = torch.as_tensor(test[a:a + batch_size], device=device)
            q = b[:, None, :].expand(len(b), num_nodes, 3).clone()
            q[:, :, 0 if head_query else 2] = cand[None, :]
            scores = model.scoring_function(q, x)                       # [batch, N] via the DistMult kernel
            target = b[:, 0 if head_query else 2]
            true_score = scores.gather(1, target[:, None])
            if filtered:
                for i, (s, p, o) in enumerate(b.tolist()):
                    known = true_heads.get((p, o), []) if head_query else true_tails.get((s, p), [])
                    if known:
                        scores[i, torch.as_tensor(known, device=device)] = float("-inf")
            ranks.append(((scores > true_score).sum(1) + 1).cpu())
    ranks = torch.cat(ranks).float()
    out = {"mrr": (1.0 / ranks).mean().item()}
    for k in hits_at:
        out[f"hits@{k}"] = (ranks <= k).float().mean().item()
    return out
