"""Ranking evaluator and samplers of the link-prediction experiments on the GPU:
HIP score-all / filter / rank-count kernels behind utils.misc.evaluate against the reference's own ranks (golden
g7_*), the CPU oracle, and size-independent checks at WN18 size.  Everything calls through librgcn_hip.so."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import oracle

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL = 1e-4


class _Model(torch.nn.Module):
    """encoder + DistMult pair with the attribute names utils.misc.evaluate looks for (models.py LinkPredictor)"""

    def __init__(self, decoder, nodes=None, layer=None, emb=None):
        super().__init__()
        self.scoring_function, self.layer = decoder, layer
        self.nodes, self.emb = nodes, emb
        self.encoder_calls = 0

    def encode(self, graph):
        self.encoder_calls += 1
        return self.nodes if self.layer is None else self.layer(graph, torch.relu(self.emb))

    def forward(self, graph, triples):
        return self.scoring_function(triples, self.encode(graph)), 0


class _Opaque(torch.nn.Module):
    """the same model without `encode`: evaluate() has to go through model(graph, toscore) like the reference"""

    def __init__(self, inner):
        super().__init__()
        self.inner = inner

    def forward(self, graph, triples):
        return self.inner(graph, triples)


def _decoder(d, N, R0, biased):
    from torch_rgcn.layers import DistMult
    dm = DistMult(R0, d["nodes"].shape[1], N, R0, b_init="ones" if biased else None).to(DEV)
    with torch.no_grad():
        dm.relations.copy_(torch.from_numpy(d["relations"]))
        if biased:
            for n in ("sbias", "pbias", "obias"):
                getattr(dm, n).copy_(torch.from_numpy(d[n]))
    return dm


def test_g7_ties_scores_and_ranks_exact():
    from torch_rgcn import _native
    from utils import misc
    d = load_golden("g7_eval_ties")
    N, R0 = int(d["num_nodes"]), int(d["num_rels"])
    dm = _decoder(d, N, R0, True)
    nodes = torch.from_numpy(d["nodes"]).to(DEV)
    test = torch.from_numpy(d["test"])
    sc = _native.distmult_score_all(test.to(DEV), True, nodes, dm.relations.detach(), dm.sbias.detach(), dm.pbias.detach(),
                                    dm.obias.detach())
    assert np.array_equal(sc.cpu().numpy(), d["head_scores"])          # small integers: exact in any summation order
    true_triples = misc.generate_true_dict(np.concatenate([d["known"], d["known"][:10], d["test"]]))
    model = _Model(dm, nodes=nodes)
    for tag, filt in (("filtered", True), ("raw", False)):
        for bs in (25, 7, 1):
            mrr, hits, ranks = misc.evaluate(model, None, test, true_triples, N, batch_size=bs, filter_candidates=filt, verbose=False)
            assert ranks == d[f"ranks_{tag}"].tolist(), (tag, bs)
            assert abs(mrr - float(d[f"mrr_{tag}"])) < 1e-12 and np.allclose(hits, d[f"hits_{tag}"], atol=1e-12)
        # a model without encode(): the reference's own route, candidate tensor through model.forward
        assert misc.evaluate(_Opaque(model), None, test, true_triples, N, batch_size=6, filter_candidates=filt, verbose=False)[2] \
            == d[f"ranks_{tag}"].tolist()


def test_g7_lp_encoder_evaluate_vs_reference():
    from torch_rgcn.layers import RelationalGraphConvolutionLP
    from utils import misc
    d = load_golden("g7_eval_lp")
    N, R0 = int(d["num_nodes"]), int(d["num_rels"])
    dim = d["emb"].shape[1]
    layer = RelationalGraphConvolutionLP(num_nodes=N, num_relations=2 * R0 + 1, in_features=dim, out_features=dim,
                                         edge_dropout={"general": 0.5, "self_loop": 0.2, "self_loop_type": "schlichtkrull-dropout"},
                                         decomposition={"type": "basis", "num_bases": 2}, w_init="glorot-normal",
                                         b_init="zeros").to(DEV)
    with torch.no_grad():
        for n, p in layer.named_parameters():
            p.copy_(torch.from_numpy(d["layer_param_" + n]))
    model = _Model(_decoder(d, N, R0, False), layer=layer, emb=torch.from_numpy(d["emb"]).to(DEV)).eval()
    true_triples = misc.generate_true_dict(np.concatenate([d["train"], d["valid"], d["test"]]))
    train = torch.from_numpy(d["train"])
    x = model.encode(train).detach()
    assert np.abs(x.cpu().numpy() - d["nodes"]).max() < TOL * np.abs(d["nodes"]).max()
    for tag, filt in (("filtered", True), ("raw", False)):
        model.encoder_calls = 0
        mrr, hits, ranks = misc.evaluate(model, train, torch.from_numpy(d["test"]), true_triples, N, batch_size=7,
                                         filter_candidates=filt, verbose=False)
        assert model.encoder_calls == 1                                 # encode once, score many
        assert ranks == d[f"ranks_{tag}"].tolist(), tag
        assert abs(mrr - float(d[f"mrr_{tag}"])) < 1e-9


@pytest.mark.parametrize("N,Q,dim,biased", [(1, 1, 1, False), (63, 65, 6, True), (130, 3, 8, False), (1000, 200, 200, True),
                                            (257, 129, 50, False), (64, 64, 16, True), (77, 10, 500, False)])
def test_score_all_vs_oracle(N, Q, dim, biased):
    from torch_rgcn import _native
    rng = np.random.default_rng(N + Q + dim)
    R0 = 5
    nodes = rng.standard_normal((N, dim)).astype(np.float32)
    rel = rng.standard_normal((R0, dim)).astype(np.float32)
    bias = [rng.standard_normal(n).astype(np.float32) for n in (N, R0, N)] if biased else [None] * 3
    batch = np.stack([rng.integers(0, N, Q), rng.integers(0, R0, Q), rng.integers(0, N, Q)], 1)
    dev = lambda a: None if a is None else torch.from_numpy(a).to(DEV)  # noqa: E731
    for head in (True, False):
        toscore = np.repeat(batch[:, None, :], N, axis=1)
        toscore[:, :, 0 if head else 2] = np.arange(N)[None, :]
        want = oracle.distmult_forward(toscore, nodes, rel, *bias)
        got = _native.distmult_score_all(dev(batch), head, dev(nodes), dev(rel), *[dev(b) for b in bias]).cpu().numpy()
        assert got.shape == (Q, N)
        assert np.abs(got - want).max() < TOL * max(np.abs(want).max(), 1e-30), head
        # filter + count against a numpy restatement on the SAME score matrix (exact)
        filt = np.unique(np.stack([rng.integers(0, Q, 3 * Q), rng.integers(0, N, 3 * Q)], 1), axis=0)
        target = batch[:, 0 if head else 2]
        filt = filt[filt[:, 1] != target[filt[:, 0]]]
        sc = torch.from_numpy(got).to(DEV)
        if len(filt):
            _native.rank_filter(sc, dev(filt[:, 0].astype(np.int32)), dev(filt[:, 1].astype(np.int32)))
        ref = got.copy()
        ref[filt[:, 0], filt[:, 1]] = -np.inf
        assert np.array_equal(sc.cpu().numpy(), ref)
        g, t = _native.rank_count(sc, dev(batch), head)
        true = ref[np.arange(Q), target][:, None]
        assert np.array_equal(g.cpu().numpy(), (ref > true).sum(1)) and np.array_equal(t.cpu().numpy(), (ref == true).sum(1))


def test_score_all_argument_errors():
    from torch_rgcn import _native
    nodes = torch.randn(10, 8, device=DEV)
    rel = torch.randn(3, 8, device=DEV)
    ok = torch.tensor([[0, 1, 2]], device=DEV)
    with pytest.raises(AssertionError):
        _native.distmult_score_all(torch.tensor([[0, 3, 2]], device=DEV), True, nodes, rel)      # relation out of range
    with pytest.raises(AssertionError):
        _native.distmult_score_all(torch.tensor([[10, 1, 2]], device=DEV), True, nodes, rel)     # node out of range
    with pytest.raises(RuntimeError):
        _native.distmult_score_all(ok.cpu(), True, nodes, rel)                                    # no CPU path
    with pytest.raises(AssertionError):
        _native.distmult_score_all(ok, True, nodes, rel, torch.zeros(10, device=DEV), None, None)  # biases: all or none
    assert _native.distmult_score_all(ok[:0], True, nodes, rel).shape == (0, 10)


def test_evaluate_wn18_sized_matches_dense_torch_ranks():
    """N = 40,943, d = 200, 2,000 test triples: MFMA scores against an fp32 matmul, ranks against torch on the same scores"""
    from torch_rgcn import _native
    from utils import misc
    N, R0, dim, Q = 40_943, 18, 200, 2_000
    g = torch.Generator(device=DEV).manual_seed(0)
    nodes = torch.randn(N, dim, device=DEV, generator=g)
    d = {"nodes": nodes.cpu().numpy(), "relations": torch.randn(R0, dim, generator=torch.Generator().manual_seed(1)).numpy()}
    dm = _decoder(d, N, R0, False)
    test = torch.from_numpy(_native.synthetic_triples_host(N, R0, Q, 5))
    known = _native.synthetic_triples_host(N, R0, 150_000, 6)
    known[:5000, :2] = test[:, :2].numpy()[np.arange(5000) % Q]            # make sure filters are hit
    true_triples = misc.generate_true_dict(np.concatenate([known, test.numpy()]))
    model = _Model(dm, nodes=nodes)
    mrr, hits, ranks = misc.evaluate(model, None, test, true_triples, N, batch_size=16, verbose=False)
    assert len(ranks) == 2 * Q and 0 < mrr < 1
    old = torch.backends.cuda.matmul.allow_tf32
    torch.backends.cuda.matmul.allow_tf32 = False
    try:
        for head in (True, False):
            batch = test.to(DEV)
            qv = nodes[batch[:, 2 if head else 0]] * dm.relations[batch[:, 1]]
            dense = qv.double() @ nodes.double().t()
            sc = _native.distmult_score_all(batch, head, nodes, dm.relations.detach())
            assert (sc.double() - dense).abs().max().item() < TOL * dense.abs().max().item()
            misc.filter_scores(sc, batch, true_triples, head=head)
            true = sc.gather(1, batch[:, 0 if head else 2][:, None])
            want = ((sc > true).sum(1) + ((sc == true).sum(1) - 1) // 2 + 1).tolist()
            assert ranks[(0 if head else Q):(Q if head else 2 * Q)] == want
    finally:
        torch.backends.cuda.matmul.allow_tf32 = old


def test_negative_sampling_and_samplers_on_device():
    from utils import misc
    N, bs, ns = 500, 2000, 10
    pos = torch.from_numpy(np.stack([np.arange(bs) % N, np.arange(bs) % 7, (np.arange(bs) * 3) % N], 1)).to(DEV)
    neg = pos.clone()[:, None, :].expand(bs, ns, 3).contiguous()
    out = misc.negative_sampling(neg, N, 0.3, device=DEV)
    assert out.shape == (bs * ns, 3) and out.data_ptr() == neg.data_ptr()
    ref = pos[:, None, :].expand(bs, ns, 3).reshape(-1, 3)
    assert torch.equal(out[:, 1], ref[:, 1]) and int(out.min()) >= 0 and int(out[:, [0, 2]].max()) < N
    changed_h, changed_t = out[:, 0] != ref[:, 0], out[:, 2] != ref[:, 2]
    assert not bool((changed_h & changed_t).any())                         # head or tail, never both
    frac_head = changed_h.float().sum() / (changed_h | changed_t).float().sum()
    assert abs(frac_head.item() - 0.3) < 0.02
    assert misc.select_sampling("uniform") is misc.uniform_sampling
    assert misc.select_sampling("Edge-Neighborhood") is misc.edge_neighborhood
    with pytest.raises(NotImplementedError):
        misc.select_sampling("snowball")
    triples = [[i % 50, i % 3, (7 * i) % 50] for i in range(300)]
    sub = misc.uniform_sampling(triples, sample_size=100)
    assert len(sub) == 100 and all(t in triples for t in sub)
    sub = misc.edge_neighborhood(triples, sample_size=100, entities={str(i): i for i in range(50)}, seed=3)
    assert len(sub) == 100 and all(t in triples for t in sub)


def test_s_penalty_value_and_gradients_match_the_gather_formulation():
    """layers.py:77-85 literally (three gathers, autograd scatters) against the histogram form used here"""
    from torch_rgcn.layers import DistMult
    N, R0, dim, T = 500, 7, 24, 4000
    g = torch.Generator().manual_seed(3)
    dm = DistMult(R0, dim, N, R0).to(DEV)
    nodes = torch.randn(N, dim, generator=g).to(DEV).requires_grad_(True)
    for shape in ((T, 3), (40, 100, 3)):
        tr = torch.stack([torch.randint(0, N, (T,), generator=g), torch.randint(0, R0, (T,), generator=g),
                          torch.randint(0, N, (T,), generator=g)], dim=1).reshape(shape).to(DEV)
        pen = dm.s_penalty(tr, nodes)
        gn, gr = torch.autograd.grad(pen, [nodes, dm.relations])
        s, p, o = tr[..., 0], tr[..., 1], tr[..., 2]
        ref = nodes[s, :].pow(2).mean() + dm.relations[p, :].pow(2).mean() + nodes[o, :].pow(2).mean()
        rn, rr = torch.autograd.grad(ref, [nodes, dm.relations])
        assert abs(pen.item() - ref.item()) < 1e-5 * abs(ref.item())
        assert (gn - rn).abs().max().item() < 1e-5 * rn.abs().max().item()
        assert (gr - rr).abs().max().item() < 1e-5 * rr.abs().max().item()


def test_bce_head_of_nothing_is_nan_like_aten():
    """F.binary_cross_entropy_with_logits of zero elements is NaN with an empty gradient, not an error"""
    from torch_rgcn import _native
    dev = torch.device("cuda:0")
    loss, ds = _native.bce_head(torch.empty(0, device=dev), torch.empty(0, device=dev))
    assert torch.isnan(loss).all() and ds.shape == (0,)
    want = torch.nn.functional.binary_cross_entropy_with_logits(torch.empty(0, device=dev), torch.empty(0, device=dev))
    assert torch.isnan(want)


def test_bce_head_workspaces_are_per_stream():
    """two streams launching the one-launch BCE head concurrently must not share a ticket / partial workspace"""
    from torch_rgcn import _native
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x, y = torch.randn(300_000, device=dev), (torch.rand(300_000, device=dev) > 0.5).float()
    want = torch.nn.functional.binary_cross_entropy_with_logits(x, y)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    got = []
    for _ in range(20):
        for st in (s1, s2):
            with torch.cuda.stream(st):
                got.append(_native.bce_head(x, y)[0])
    torch.cuda.synchronize()
    assert len({k for k in _native._BCE_WS if k[0] == dev}) >= 2
    for g in got:
        assert abs(g.item() - want.item()) < 1e-5 * abs(want.item())
