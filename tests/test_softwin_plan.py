"""The soft-window plan (torch_rgcn._native.build_softwin_plan, round 6) is made with torch ops, so its invariants are checked on the CPU:
every live message sits in exactly one slot, a chunk holds ONE relation and ONE tile, the slots of a (tile, relation) bucket are sorted by
source, the chunks of a tile by first source, pads are (dst -1, val 0), the tile pointers count chunks."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-rgcn_amd"))


def random_messages(N, R, M, masked, seed):
    rng = np.random.default_rng(seed)
    dst = torch.from_numpy(rng.integers(0, N, M).astype(np.int32))
    src = torch.from_numpy(rng.integers(0, N, M).astype(np.int32))
    rel = torch.from_numpy(rng.integers(0, R, M).astype(np.int32))
    val = torch.from_numpy(rng.random(M).astype(np.float32) + 0.5)
    alive = torch.from_numpy((rng.random(M) < 0.7).astype(np.uint8)) if masked else None
    return dst, src, rel, val, alive


def check_plan(p, dst, src, rel, val, alive, N, R, rows):
    """the invariants of a soft-window plan without owners (the plan's tensors may live on the GPU)"""
    M = dst.shape[0]
    live = np.ones(M, bool) if alive is None else alive.cpu().numpy() != 0
    assert p.n_messages == int(live.sum()) and p.m_pad == 16 * p.n_chunks and p.n_tiles == -(-N // rows)
    S, D, V = (t.cpu().numpy()[:p.m_pad] for t in (p.src, p.dst, p.val))
    real = D >= 0
    assert int(real.sum()) == p.n_messages and np.all(V[~real] == 0.0)
    crel = p.chunk_rel.cpu().numpy()[:p.n_chunks]
    # the multiset of (dst, src, rel, val) is the live message list
    got = sorted(zip(D[real].tolist(), S[real].tolist(), np.repeat(crel, 16)[real].tolist(), V[real].tolist()))
    want = sorted(zip(dst.cpu().numpy()[live].tolist(), src.cpu().numpy()[live].tolist(), rel.cpu().numpy()[live].tolist(), val.cpu().numpy()[live].tolist()))
    assert got == want
    tp = p.tile_ptr.cpu().numpy()
    assert tp[0] == 0 and tp[p.n_tiles] == p.n_chunks and np.all(np.diff(tp) >= 0)
    rp = p.run_ptr.cpu().numpy().reshape(p.n_tiles, R + 1)
    assert np.array_equal(rp[:, 0], tp[:-1]) and np.array_equal(rp[:, R], tp[1:])
    for t in range(p.n_tiles):
        firsts = []
        seen = {}
        for c in range(tp[t], tp[t + 1]):
            d, s = D[16 * c:16 * c + 16], S[16 * c:16 * c + 16]
            k = int((d >= 0).sum())
            assert k >= 1 and np.all(d[:k] >= 0) and np.all(d[k:] < 0)            # real slots first, a chunk never starts with a pad
            assert np.all(d[:k] // rows == t)
            assert np.all(np.diff(s[:k]) >= 0)                                    # sorted by source inside the chunk
            firsts.append(int(s[0]))
            r = int(crel[c])
            assert seen.get(r, -1) <= int(s[0])                                   # ... and from chunk to chunk of one bucket
            seen[r] = int(s[k - 1])
        assert firsts == sorted(firsts)                                           # the tile's chunks: by first source


@pytest.mark.parametrize("N,R,M,rows,masked", [(1000, 7, 20000, 128, False), (1000, 7, 20000, 128, True), (37, 3, 11, 16, False),
                                                (500, 1, 3000, 977, False), (64, 5, 0, 16, False)])
def test_softwin_plan_invariants(N, R, M, rows, masked):
    from torch_rgcn import _native
    dst, src, rel, val, alive = random_messages(N, R, M, masked, N + M)
    check_plan(_native.build_softwin_plan(dst, src, rel, val, alive, N, N, R, rows), dst, src, rel, val, alive, N, R, rows)


def test_own_relations_lpt_packing():
    from torch_rgcn import _native
    counts = [1000] + [200] * 100                     # S1's shape: the self-loop relation is five times the others
    parts, base, owner, local, unit_rel, balance = _native.own_relations(counts, 12, 9)
    assert balance < 1.06 and parts.sum() <= 108 and parts[0] >= 1 and all(parts[1:] == 1)
    for r in range(101):
        for q in range(parts[r]):
            u = base[r] + q
            assert unit_rel[owner[u] * 9 + local[u]] == r and 0 <= local[u] < 9
    assert sorted(set(int(u) for u in unit_rel if u >= 0)) == list(range(101)) and int((unit_rel >= 0).sum()) == parts.sum()
    assert _native.own_relations([1] * 109, 12, 9) is None      # more relations than slots
    # one dominant relation: cut into parts, so that its chunks spread over the waves
    parts, base, owner, local, unit_rel, balance = _native.own_relations([5000, 1, 1, 1], 12, 9)
    assert parts[0] >= 12 and balance < 1.3
    # a graph with ONE relation (self loops only): every wave owns a part
    parts, base, owner, local, unit_rel, balance = _native.own_relations([4096], 12, 9)
    assert parts[0] >= 12 and len(set(owner.tolist())) == 12 and balance < 1.2


def check_owner_plan(p, dst, src, rel, val, alive, N, R, rows, NW, K):
    M = dst.shape[0]
    live = np.ones(M, bool) if alive is None else alive.cpu().numpy() != 0
    D, S, V = (t.cpu().numpy()[:p.m_pad] for t in (p.dst, p.src, p.val))
    packed = p.chunk_rel.cpu().numpy()[:p.n_chunks]
    crel, cloc = packed & 0xFFFF, packed >> 16
    unit_rel = p.unit_rel.cpu().numpy()
    op = p.own_ptr.cpu().numpy()
    assert op.shape[0] == p.n_tiles * NW + 1 and op[0] == 0 and op[-1] == p.n_chunks and np.all(np.diff(op) >= 0)
    real = D >= 0
    got = sorted(zip(D[real].tolist(), S[real].tolist(), np.repeat(crel, 16)[real].tolist(), V[real].tolist()))
    want = sorted(zip(dst.cpu().numpy()[live].tolist(), src.cpu().numpy()[live].tolist(), rel.cpu().numpy()[live].tolist(), val.cpu().numpy()[live].tolist()))
    assert got == want
    assert sorted(set(int(r) for r in unit_rel if r >= 0)) == list(range(R))
    tp = p.tile_ptr.cpu().numpy()
    for t in range(p.n_tiles):
        assert op[t * NW] == tp[t]
        for w in range(NW):
            firsts = []
            for c in range(op[t * NW + w], op[t * NW + w + 1]):
                assert unit_rel[w * K + int(cloc[c])] == crel[c]           # the chunk is in the range of a wave that owns (a part of) its relation
                assert D[16 * c] >= 0 and D[16 * c] // rows == t
                k = int((D[16 * c:16 * c + 16] >= 0).sum())
                assert np.all(np.diff(S[16 * c:16 * c + k]) >= 0)
                firsts.append(int(S[16 * c]))
            assert firsts == sorted(firsts)


@pytest.mark.parametrize("masked,R", [(False, 20), (True, 20), (False, 2), (True, 1)])
def test_softwin_plan_with_relation_owners(masked, R):
    from torch_rgcn import _native
    N, M, rows, NW, K = 900, 30000, 128, 12, 9
    dst, src, rel, val, alive = random_messages(N, R, M, masked, 7)
    p = _native.build_softwin_plan(dst, src, rel, val, alive, N, N, R, rows, own_waves=NW, own_per_wave=K)
    check_owner_plan(p, dst, src, rel, val, alive, N, R, rows, NW, K)
