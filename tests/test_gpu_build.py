"""Device-side graph preparation (csrc/rgcn_build.hip) against the host builder / the oracle:
integer work is bit-exact; plans are equal up to the (irrelevant) order inside one (relation, dst) cell."""
import numpy as np
import pytest
import torch
from torch_rgcn import routes  # noqa: E402

from oracle import oracle
from torch_rgcn import _native as nat

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def graph(N, R0, E, seed, hub=True):
    T = oracle.synthetic_triples(N, R0, E, seed)
    if hub and E > 100:
        T[: E // 3, 0] = 2
        T[E // 3: E // 3 + E // 10] = T[0]   # duplicates
    return T


@pytest.mark.parametrize("N,R0,E", [(1, 1, 0), (40, 2, 7), (700, 5, 9000), (5000, 40, 60000)])
def test_device_lp_expand_and_norm_bit_exact(N, R0, E):
    T = graph(N, R0, E, N + E)
    rng = np.random.default_rng(N)
    keep = rng.integers(0, 2, N).astype(np.uint8)
    R = 2 * R0 + 1
    for k in (None, keep):
        s, p, o, alive, err = nat.dev_lp_expand(torch.from_numpy(T).to(DEV), N, R0, None if k is None else torch.from_numpy(k).to(DEV))
        assert int(err.item()) == 0
        tp, n_self = oracle.lp_augment(T, N, R0, k)
        live = alive.cpu().numpy().astype(bool)
        got = np.stack([s.cpu().numpy(), p.cpu().numpy(), o.cpu().numpy()], 1)[live]
        assert np.array_equal(got, tp)
        for vertical in (True, False):
            val = nat.dev_edge_norm(s, p, o, alive, N, R, vertical, E).cpu().numpy()
            assert np.array_equal(val[live], oracle.edge_norm(tp, N, R, vertical, E, n_self))
            assert np.all(val[~live] == 0)


def test_device_nc_norm_bit_exact_and_range_error():
    N, R0, E = 3000, 12, 40000
    tp = oracle.add_inverse_and_self(graph(N, R0, E, 5), N, R0)
    R = 2 * R0 + 1
    s, p, o, err = nat.dev_split_triples(torch.from_numpy(tp).to(DEV), N, R)
    assert int(err.item()) == 0
    for vertical in (True, False):
        val = nat.dev_edge_norm(s, p, o, None, N, R, vertical, (len(tp) - N) // 2).cpu().numpy()
        assert np.array_equal(val, oracle.nc_edge_norm(tp, N, R, vertical))
    bad = tp.copy()
    bad[17, 2] = N
    _, _, _, err = nat.dev_split_triples(torch.from_numpy(bad).to(DEV), N, R)
    with pytest.raises(AssertionError):
        nat.dev_check_err(err, "range")


def canon(src, dst, val, run_starts, m_pad):
    """slots sorted inside every (tile, relation) run by (dst, src, val): order inside a (relation, dst) cell is free"""
    run_of_chunk = np.searchsorted(run_starts, np.arange(m_pad // 16), side="right") - 1
    bucket = np.repeat(run_of_chunk, 16)
    order = np.lexsort((val[:m_pad], src[:m_pad], dst[:m_pad].astype(np.int64) % (1 << 31), bucket))
    return src[:m_pad][order], dst[:m_pad][order], val[:m_pad][order]


@pytest.mark.parametrize("N,R,M,tile", [(1, 1, 0, 8), (50, 5, 700, 16), (3000, 21, 50000, 128), (3000, 21, 50000, 3000),
                                        (20000, 101, 300000, 128)])
def test_device_plan_equals_host_plan(N, R, M, tile):
    rng = np.random.default_rng(N + M)
    dst = rng.integers(0, N, M).astype(np.int32)
    if M > 100:
        dst[:M // 4] = 3
    src = rng.integers(0, N, M).astype(np.int32)
    rel = rng.integers(0, R, M).astype(np.int32)
    val = rng.random(M).astype(np.float32) + 0.1
    alive = np.ones(M, np.uint8)
    if M > 10:
        alive[rng.integers(0, M, M // 7)] = 0
    live = alive.astype(bool)
    hp = nat.build_plan_host(dst[live], src[live], rel[live], val[live], N, N, R, tile, 64, want_runs=True,
                             want_pack=tile <= 255)
    t = lambda a: torch.from_numpy(a).to(DEV)
    dp = nat.build_plan_device(t(dst), t(src), t(rel), t(val), t(alive), N, N, R, tile, int(live.sum()), 64,
                               want_runs=True, want_pack=tile <= 255)
    assert (dp.m_pad, dp.n_chunks, dp.n_tiles, dp.n_units, dp.n_split) == (hp.m_pad, hp.n_chunks, hp.n_tiles, hp.n_units, hp.n_split)
    assert np.array_equal(dp.tile_ptr.cpu().numpy()[:hp.n_tiles + 1], hp.tile_ptr)
    assert np.array_equal(dp.chunk_rel.cpu().numpy()[:hp.n_chunks], hp.chunk_rel[:hp.n_chunks])
    assert np.array_equal(dp.run_ptr.cpu().numpy()[:hp.n_tiles * (R + 1)], hp.run_ptr[:hp.n_tiles * (R + 1)])
    assert np.array_equal(dp.units.cpu().numpy()[:hp.n_units], hp.units[:hp.n_units])
    assert dp.max_run_chunks == hp.max_run_chunks
    starts = hp.run_ptr.reshape(-1, R + 1)[:, :R].ravel() if hp.n_tiles else np.zeros(0, np.int32)
    a = canon(dp.src.cpu().numpy(), dp.dst.cpu().numpy(), dp.val.cpu().numpy(), starts, hp.m_pad)
    b = canon(hp.src, hp.dst, hp.val, starts, hp.m_pad)
    # pads copy "the last real source of the bucket", which depends on the free order: compare real slots only
    real = b[2] != 0
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert np.array_equal(a[0][real], b[0][real])
    if tile <= 255 and hp.m_pad:
        pk = dp.pack.cpu().numpy()[:hp.m_pad]
        d = dp.dst.cpu().numpy()[:hp.m_pad]
        assert np.array_equal(pk[:, 0] & 0xFFFFFF, dp.src.cpu().numpy()[:hp.m_pad])
        assert np.array_equal((pk[:, 0].view(np.uint32) >> 24), np.where(d < 0, 255, d % tile).astype(np.uint32))
        assert np.array_equal(pk[:, 1].view(np.float32), dp.val.cpu().numpy()[:hp.m_pad])
    if tile >= N and hp.n_items:
        assert np.array_equal(dp.items.cpu().numpy()[:hp.n_items], hp.items[:hp.n_items])


def test_layers_give_same_result_with_host_and_device_build(monkeypatch):
    from torch_rgcn.layers import RelationalGraphConvolutionNC
    N, R0, E = 4000, 9, 50000
    tp = oracle.add_inverse_and_self(graph(N, R0, E, 3), N, R0)
    outs = []
    for mode in ("device", "host"):
        routes.patch(monkeypatch, "graph_build", mode)
        torch.manual_seed(0)
        layer = RelationalGraphConvolutionNC(triples=torch.from_numpy(tp), num_nodes=N, num_relations=2 * R0 + 1,
                                             in_features=16, out_features=16).to(DEV)
        X = torch.randn(N, 16, device=DEV, requires_grad=True)
        out = layer(X)
        out.backward(torch.ones_like(out))
        outs.append((out.detach().cpu().numpy(), X.grad.cpu().numpy(), layer.weights.grad.cpu().numpy()))
    for a, b in zip(*outs):
        assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max()
