"""Host-side graph preparation of librgcn_hip.so (C++, no GPU) against the oracle: index
work is bit-exact; the relation-tile plan is checked through its invariants."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle
from torch_rgcn import _native as nat


def rand_triples(rng, N, R0, E):
    return np.stack([rng.integers(0, N, E), rng.integers(0, R0, E), rng.integers(0, N, E)], axis=1).astype(np.int64)


@pytest.mark.parametrize("N,R0,E", [(1, 1, 0), (5, 2, 1), (17, 3, 60), (400, 11, 5000)])
def test_augmentation_bit_exact(N, R0, E):
    rng = np.random.default_rng(N * 1000 + E)
    T = rand_triples(rng, N, R0, E)
    assert np.array_equal(nat.add_inverse_and_self_host(T, N, R0), oracle.add_inverse_and_self(T, N, R0))
    keep = rng.integers(0, 2, N).astype(np.uint8)
    for k in (None, keep):
        a, na = nat.lp_augment_host(T, N, R0, k)
        b, nb = oracle.lp_augment(T, N, R0, k)
        assert na == nb and np.array_equal(a, b)


def test_synthetic_generator_matches_oracle():
    a = nat.synthetic_triples_host(1000, 7, 5000, seed=42)
    b = oracle.synthetic_triples(1000, 7, 5000, seed=42)
    assert np.array_equal(a, b)
    assert a[:, 0].max() < 1000 and a[:, 1].max() < 7


@pytest.mark.parametrize("N,R0,E", [(5, 2, 1), (17, 3, 60), (400, 11, 5000), (3000, 50, 30000)])
def test_edge_norm_bit_exact(N, R0, E):
    rng = np.random.default_rng(7 * N + E)
    T = rand_triples(rng, N, R0, E)
    T[: E // 10] = T[E // 10: 2 * (E // 10)]  # duplicates
    R = 2 * R0 + 1
    tp = oracle.add_inverse_and_self(T, N, R0)
    M = len(tp)
    for vertical in (True, False):
        a = nat.edge_norm_host(tp, N, R, vertical, (M - N) // 2, N)
        b = oracle.edge_norm(tp, N, R, vertical, (M - N) // 2, N)
        assert np.array_equal(a, b)
    keep = rng.integers(0, 2, N).astype(np.uint8)
    tl, ns = oracle.lp_augment(T, N, R0, keep)
    for vertical in (True, False):
        assert np.array_equal(nat.edge_norm_host(tl, N, R, vertical, E, ns), oracle.edge_norm(tl, N, R, vertical, E, ns))
    # arbitrary (ill-formed) row order still follows the literal count / swap / divide procedure
    perm = rng.permutation(M)
    assert np.array_equal(nat.edge_norm_host(tp[perm], N, R, False, (M - N) // 2, N),
                          oracle.edge_norm(tp[perm], N, R, False, (M - N) // 2, N))


def test_edge_norm_golden_g3():
    d = load_golden("g3_utils")
    N, R0 = int(d["num_nodes"]), int(d["num_rels"])
    tp = d["triples_plus"]
    M = len(tp)
    v = nat.edge_norm_host(tp, N, 2 * R0 + 1, True, (M - N) // 2, N)
    assert np.array_equal(v, np.float32(1.0) / d["ver_sums"])
    h = nat.edge_norm_host(tp, N, 2 * R0 + 1, False, (M - N) // 2, N)
    n = (M - N) // 2
    hs = d["hor_sums"]
    assert np.array_equal(h, np.float32(1.0) / np.concatenate([hs[n:2 * n], hs[:n], hs[-N:]]))


def test_edge_norm_errors():
    tp = np.array([[0, 0, 1], [1, 1, 0], [0, 2, 0], [1, 2, 1]])
    with pytest.raises(AssertionError):  # node id out of range -> stack_matrices' assert
        nat.edge_norm_host(np.array([[0, 0, 5]]), 2, 3, True, 0, 0)
    with pytest.raises(AssertionError):  # relation id out of range
        nat.edge_norm_host(np.array([[0, 3, 1]]), 2, 3, False, 0, 1)
    with pytest.raises(AssertionError):  # 2n + i != M: the reference's cat() yields a shape error
        nat.edge_norm_host(tp, 2, 3, False, 2, 2)
    assert nat.edge_norm_host(np.zeros((0, 3), np.int64), 2, 3, True, 0, 0).shape == (0,)


def check_plan(hp, dst, src, rel, val, n_dst, R, tile_rows, max_item_chunks):
    C = nat.CHUNK
    M = len(dst)
    assert hp.m_pad == hp.n_chunks * C and hp.n_tiles == -(-n_dst // tile_rows)
    perm = hp.perm[:hp.m_pad]
    real = perm >= 0
    assert real.sum() == M and np.array_equal(np.sort(perm[real]), np.arange(M))  # each message exactly once
    assert np.array_equal(hp.src[:hp.m_pad][real], src[perm[real]])
    assert np.array_equal(hp.dst[:hp.m_pad][real], dst[perm[real]]) and np.all(hp.dst[:hp.m_pad][~real] == -1)
    assert np.array_equal(hp.val[:hp.m_pad][real], val[perm[real]])
    assert np.all(hp.val[:hp.m_pad][~real] == 0)
    tp = hp.tile_ptr
    assert tp[0] == 0 and tp[-1] == hp.n_chunks and np.all(np.diff(tp) >= 0)
    pd = hp.dst[:hp.m_pad].reshape(-1, C)
    pv = hp.val[:hp.m_pad].reshape(-1, C)
    pr = perm.reshape(-1, C)
    for t in range(hp.n_tiles):
        for c in range(tp[t], tp[t + 1]):
            rl = pr[c] >= 0
            assert np.all(pd[c][rl] // tile_rows == t)           # one destination tile per chunk
            r = hp.chunk_rel[c]
            assert np.all(rel[pr[c][pr[c] >= 0]] == r)          # one relation per chunk
            assert pr[c][0] >= 0                                 # never an all-pad chunk
            assert np.all(np.diff(pd[c][rl]) >= 0) and np.all(np.diff(rl.astype(int)) <= 0)  # sorted; pads trail
            assert np.all(pv[c][pr[c] < 0] == 0)
        rels = hp.chunk_rel[tp[t]:tp[t + 1]]
        assert np.all(np.diff(rels) >= 0)                        # relation-grouped inside the tile
    # work items partition the chunks into constant-relation ranges
    it = hp.items[:hp.n_items]
    if hp.n_items:
        assert it[0, 0] == 0 and it[-1, 1] == hp.n_chunks and np.array_equal(it[1:, 0], it[:-1, 1])
        for c0, c1 in it:
            assert 0 < c1 - c0 <= max_item_chunks and len(set(hp.chunk_rel[c0:c1])) == 1
    else:
        assert hp.n_chunks == 0


@pytest.mark.parametrize("N,R,M,tile", [(1, 1, 0, 4), (7, 3, 1, 4), (50, 5, 700, 16), (50, 5, 700, 64),
                                        (1000, 21, 20000, 512), (333, 9, 5000, 1000)])
def test_plan_invariants(N, R, M, tile):
    rng = np.random.default_rng(N + M)
    dst = rng.integers(0, N, M).astype(np.int32)
    if M > 100:
        dst[:M // 4] = 3  # a hub
    src = rng.integers(0, N, M).astype(np.int32)
    rel = rng.integers(0, R, M).astype(np.int32)
    val = (rng.random(M).astype(np.float32) + 0.1)
    hp = nat.build_plan_host(dst, src, rel, val, N, N, R, tile, max_item_chunks=5, want_perm=True, want_runs=True)
    check_plan(hp, dst, src, rel, val, N, R, tile, 5)
    # run_ptr[t][r] .. run_ptr[t][r+1] = the chunks of (tile t, relation r); rows tile the chunk list
    rp = hp.run_ptr.reshape(hp.n_tiles, R + 1) if hp.n_tiles else hp.run_ptr.reshape(0, R + 1)
    for t in range(hp.n_tiles):
        assert rp[t, 0] == hp.tile_ptr[t] and rp[t, R] == hp.tile_ptr[t + 1] and np.all(np.diff(rp[t]) >= 0)
        for r in range(R):
            assert np.all(hp.chunk_rel[rp[t, r]:rp[t, r + 1]] == r)


def test_plan_errors():
    one = np.zeros(1, np.int32)
    with pytest.raises(AssertionError):
        nat.build_plan_host(np.array([9], np.int32), one, one, np.ones(1, np.float32), 4, 4, 2, 4)
    with pytest.raises(AssertionError):
        nat.build_plan_host(one, np.array([-1], np.int32), one, np.ones(1, np.float32), 4, 4, 2, 4)
    with pytest.raises(AssertionError):
        nat.build_plan_host(one, one, np.array([2], np.int32), np.ones(1, np.float32), 4, 4, 2, 4)


def test_work_units_split_hub_tiles():
    rng = np.random.default_rng(3)
    N, R, M = 600, 5, 40000
    dst = rng.integers(0, N, M).astype(np.int32)
    dst[: M // 2] = 7   # hub
    src = rng.integers(0, N, M).astype(np.int32)
    rel = rng.integers(0, R, M).astype(np.int32)
    hp = nat.build_plan_host(dst, src, rel, np.ones(M, np.float32), N, N, R, 64, max_unit_chunks=50)
    u = hp.units[:hp.n_units]
    assert hp.n_split > 0 and hp.n_units > hp.n_tiles
    # units tile the chunk list in order; whole-tile units carry flags 0; pieces are <= 50 chunks, first piece marked
    assert u[0, 1] == 0 and u[-1, 2] == hp.n_chunks and np.array_equal(u[1:, 1], u[:-1, 2])
    for t, c0, c1, fl in u:
        assert hp.tile_ptr[t] <= c0 < c1 <= hp.tile_ptr[t + 1] or (c0 == c1 == hp.tile_ptr[t])
        if fl == 0:
            assert (c0, c1) == (hp.tile_ptr[t], hp.tile_ptr[t + 1])
        else:
            assert c1 - c0 <= 50 and bool(fl & 2) == (c0 == hp.tile_ptr[t])
    assert int((u[:, 3] != 0).sum()) == hp.n_split
    assert hp.max_run_chunks >= 50
    # every tile appears (also empty ones)
    assert set(u[:, 0].tolist()) == set(range(hp.n_tiles))


def test_slab_bounds_are_rank_independent():
    """relation-sharded ranks hold different messages but must all-reduce identical row ranges"""
    rng = np.random.default_rng(0)
    N, R, T = 1000, 7, 64
    rows = []
    for seed, M in ((1, 3000), (2, 9000), (3, 50)):
        r = np.random.default_rng(seed)
        dst = r.integers(0, N, M).astype(np.int32)
        if seed == 2:
            dst[:4000] = 5   # a hub: this "rank" has a split tile
        hp = nat.build_plan_host(dst, r.integers(0, N, M).astype(np.int32), r.integers(0, R, M).astype(np.int32),
                                 np.ones(M, np.float32), N, N, R, T, max_unit_chunks=20)
        b = nat.slab_bounds(hp, 3)
        rows.append([(r0, r1) for _, _, r0, r1 in b])
        # units of a slab are exactly the units of its tiles, in order, covering everything once
        assert b[0][0] == 0 and b[-1][1] == hp.n_units and all(x[1] == y[0] for x, y in zip(b[:-1], b[1:]))
        for u0, u1, r0, r1 in b:
            t = hp.units[u0:u1, 0]
            assert np.all(t * T >= r0) and np.all(t * T < r1)
    assert rows[0] == rows[1] == rows[2] and rows[0][0][0] == 0 and rows[0][-1][1] == N


# ------------------------------------------------------------------ edge-neighbourhood sampler (utils/misc.py:125-172)
def _walk_is_legal(T, N, picked):
    """every pick comes from a touched vertex whenever some touched vertex still has an unpicked edge"""
    seen = np.zeros(N, bool)
    left = np.bincount(np.concatenate([T[:, 0], T[:, 2]]), minlength=N)
    for e in picked:
        s, o = T[e, 0], T[e, 2]
        if (left * seen).sum() > 0:
            assert seen[s] or seen[o]
        left[s] -= 1
        left[o] -= 1
        seen[s] = seen[o] = True


def test_edge_neighborhood_native_properties():
    d = load_golden("g8_sampler")
    T, N = d["triples"], int(d["num_nodes"])
    for size in (0, 1, 25, len(T)):
        picked = nat.edge_neighborhood_host(T, N, size, seed=size + 3)
        assert len(picked) == size and len(set(picked.tolist())) == size
        assert picked.size == 0 or (picked.min() >= 0 and picked.max() < len(T))
        _walk_is_legal(T, N, picked)
    assert np.array_equal(nat.edge_neighborhood_host(T, N, 20, seed=5), nat.edge_neighborhood_host(T, N, 20, seed=5))
    big = nat.synthetic_triples_host(40_943, 18, 141_442, 1)          # WN18-sized: milliseconds, not minutes
    picked = nat.edge_neighborhood_host(big, 40_943, 30_000, seed=11)
    assert len(set(picked.tolist())) == 30_000
    _walk_is_legal(big, 40_943, picked[:2000])
    with pytest.raises(AssertionError):
        nat.edge_neighborhood_host(T, N, len(T) + 1, seed=0)          # the reference would spin forever here
    with pytest.raises(AssertionError):
        nat.edge_neighborhood_host(T, 3, 5, seed=0)                    # node index out of range


def test_edge_neighborhood_native_distribution_matches_oracle():
    """same sampling distribution as the restated reference algorithm: frequency of every edge as first pick, as second
    pick and anywhere in the sample, over 3000 runs each (tolerance ~5 sigma)."""
    T = np.array([[0, 0, 1], [1, 0, 2], [2, 0, 0], [3, 0, 4], [4, 0, 5], [1, 0, 3], [2, 0, 2], [0, 0, 1]], np.int64)
    N, size, runs = 6, 3, 3000
    rng = np.random.RandomState(0)
    freq = np.zeros((2, 3, len(T)))
    for r in range(runs):
        for k, picked in enumerate((oracle.edge_neighborhood(T, size, N, rng), nat.edge_neighborhood_host(T, N, size, seed=1000 + r))):
            freq[k, 0, picked[0]] += 1
            freq[k, 1, picked[1]] += 1
            freq[k, 2, picked] += 1
    freq /= runs
    assert np.abs(freq[0] - freq[1]).max() < 0.05, freq
