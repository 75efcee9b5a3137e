"""BASELINE.json configs 2 and 3 inside `pytest -m gpu` (`configs_untested` named AM / block-diagonal):

  AM    nc-AM.yaml:3,15-20 -- N = 1,666,764, R0 = 133 (layer R = 267), E = 5,988,321; BASELINE's block-diagonal variant is
        run at layer level (featured, d = 16, nb = 4) -- reference layers.py:243-244 + utils.py:168-196 --
        and the shipped first layer (featureless, basis 40, hidden 10) without the 17.8 GB R x N x d table.
  MUTAG nc-MUTAG.yaml:15-20 -- N = 23,644, R0 = 23, E = 74,227, basis 30, hidden 16, 2 classes, featureless layer 1.

Dataset-shaped synthetic graphs (the files are not available offline).  1/10-AM and full MUTAG go against the
oracle (out, dX, every parameter gradient, 1e-4); full-size AM goes through size-independent properties (linearity,
(relation, subject) counts, checksum), both the sparse-bucket two-pass path and the tile path."""
import numpy as np
import pytest
import torch
from torch_rgcn import routes  # noqa: E402

from oracle import oracle
from test_gpu_parity import DEV, TOL, rel_err, run_layer_vs_oracle

pytestmark = pytest.mark.gpu

AM = dict(N=1_666_764, R0=133, E=5_988_321)
MUTAG = dict(N=23_644, R0=23, E=74_227)


@pytest.mark.parametrize("vertical", [False, True])
@pytest.mark.parametrize("route", ["block", "hybrid", "hybrid_r2", "fwdblk", "twopass", "tile"])
def test_am_tenth_scale_block_diagonal_layer_vs_oracle(monkeypatch, route, vertical):
    """1/10 of AM, block-diagonal nb = 4, d = 16, both stackings, every route: out / dX / dblocks / db against the oracle.
      hybrid   the default for sparse (tile, relation) buckets (267 relations): forward on the block CSR kernel (4 x 4 blocks as
               they are), backward on the block-tile kernel (round 3: one 255-row tile per workgroup, dX + the DIAGONAL blocks of
               dW + db from one gather per message; RGCN_F_DIAG4)
      hybrid_r2  round 2's default: the backward on the expanded 16 x 16 weights, relation-major fused pass + row sums
      fwdblk   forward on the expanded weights in ONE launch of the block-tile forward kernel (round 5, rgcn_spmm_blk_f32: what a dense-weight
               layer with 267 relations takes), round 2's backward
      twopass  forward too on the expanded weights in two passes (transform in relation-major order, sum per destination; RGCN_SPMM_CSR=0),
               round 2's backward
      tile     the (tile, relation) kernels of dense-bucket graphs
      block    forward and backward on the block kernels (RGCN_BLOCK_PATH=2; at width 16 not the default)"""
    from torch_rgcn import _native
    routes.patch(monkeypatch, "block_path", "2" if route == "block" else "0")
    routes.patch(monkeypatch, "block_fwd", "1" if route in ("hybrid", "hybrid_r2") else "0")
    if route in ("hybrid_r2", "twopass", "fwdblk"):
        routes.patch(monkeypatch, "bwd_kernel", "lean")       # no block-tile kernel: the sparse graph's backward is the two-pass one
    if route in ("twopass", "fwdblk", "tile"):
        routes.patch(monkeypatch, "sparse_path", "0" if route == "tile" else "1")
    if route == "twopass":
        routes.patch(monkeypatch, "spmm_csr", "0")
    _native.profile_start()
    run_layer_vs_oracle(N=166_676, R0=133, E=598_832, d_in=16, d_out=16, mode="block", num_blocks=4, vertical=vertical,
                        seed=301 + int(vertical))
    prof = _native.profile_stop()
    assert ("block_spmm" in prof) == (route in ("block", "hybrid", "hybrid_r2")) and ("block_wgrad" in prof) == (route == "block")
    assert ("spmm_scatter" in prof) == (route == "twopass") and ("spmm_blk" in prof) == (route == "fwdblk")
    # backward: relation-major fused pass (dX rows + dW from one walk) on the sparse path, tile-walk fused kernel otherwise
    assert ("bwd_scatter_dw" in prof) == (route in ("hybrid_r2", "twopass", "fwdblk")) and ("bwd_fused" in prof) == (route in ("tile", "hybrid"))


def test_am_tenth_scale_default_path_is_the_sparse_one():
    """without any switch the 267-relation graph must pick the sparse-bucket routes by itself (fill of the 16-slot chunks < 50 %): the
    block-tile forward kernel on tall tiles (more relations than the one-pass CSR kernel holds), the relation-major fused backward"""
    from torch_rgcn import _native
    _native.profile_start()
    run_layer_vs_oracle(N=166_676, R0=133, E=598_832, d_in=16, d_out=16, mode="none", seed=303)
    prof = _native.profile_stop()
    assert "spmm_blk" in prof and "bwd_scatter_dw" in prof and "spmm" not in prof, sorted(prof)


def _am_graph():
    T = oracle.synthetic_triples(AM["N"], AM["R0"], AM["E"], seed=2)
    return oracle.add_inverse_and_self(T, AM["N"], AM["R0"])


def test_am_full_size_block_layer_properties():
    """full-size AM, featured block-diagonal layer: linearity in X, per-subject (relation, subject) group counts,
    checksum -- through the counting workspace of the full graph (445 M cells) and the max_degree() guard"""
    from torch_rgcn import _native
    from torch_rgcn.layers import RelationalGraphConvolutionNC
    N, R0 = AM["N"], AM["R0"]
    tp = _am_graph()
    layer = RelationalGraphConvolutionNC(triples=torch.from_numpy(tp), num_nodes=N, num_relations=2 * R0 + 1, in_features=16,
                                         out_features=16, bias=False, decomposition={"type": "block", "num_blocks": 4}).to(DEV)
    _native.profile_start()
    with torch.no_grad():
        a, b = torch.randn(N, 16, device=DEV), torch.randn(N, 16, device=DEV)
        ya, yb, yab = layer(a), layer(b), layer(0.5 * a + 2.0 * b)
        assert rel_err(yab, (0.5 * ya + 2.0 * yb).cpu().numpy()) < 1e-5
        layer.blocks.fill_(0.0)
        layer.blocks[:, 0, 0, 0] = 1.0                      # W_r[0, 0] = 1
        y = layer(torch.ones(N, 16, device=DEV))[:, 0]
        key = torch.from_numpy(tp[:, 1] * N + tp[:, 0]).to(DEV)
        cnt = torch.bincount(torch.unique(key) % N, minlength=N).float()
        assert torch.allclose(y, cnt, rtol=1e-5, atol=1e-5)
        assert abs(y.sum().item() - cnt.sum().item()) < 1e-3 * cnt.sum().item()
    prof = _native.profile_stop()
    assert "block_spmm" in prof and "spmm" not in prof, "full-size AM: sparse buckets -> the forward reads the 4 x 4 blocks on the CSR kernel"
    # gradients at full size: dX of a constant upstream gradient with W_r[0,0] = 1 is the count of (relation, object)
    # groups each node SENDS into, weighted by 1/c of the receiving group -- checked as a checksum: sum(dX[:,0]) = sum(out[:,0])
    X = torch.ones(N, 16, device=DEV, requires_grad=True)
    out = layer(X)
    out.backward(torch.ones_like(out))
    assert abs(X.grad[:, 0].sum().item() - out[:, 0].sum().item()) < 1e-3 * abs(out[:, 0].sum().item())
    assert torch.isfinite(layer.blocks.grad).all()
    # the same forward with the block table read from L2 instead of LDS (68 KB: above the default 64 KB LDS limit) and on the
    # two-pass kernels
    ref = out.detach()
    with routes.override(block_fwd="0"):
        assert rel_err(layer(X).detach(), ref.cpu().numpy()) < 1e-5


def test_am_full_size_dense_layer_one_launch_forward_properties():
    """full-size AM, featured layer with DENSE weights at R = 267 (the shipped model's second layer, at width 16): the forward is ONE launch of
    the block-tile forward kernel on 930-row tiles (four float4 of the tile per thread at the hand-over: the shape only this graph reaches) --
    linear in X, equal to the two-pass route, and the (relation, subject) group counts with W_r[0, 0] = 1"""
    from torch_rgcn import _native
    from torch_rgcn.layers import RelationalGraphConvolutionNC
    N, R0 = AM["N"], AM["R0"]
    tp = _am_graph()
    layer = RelationalGraphConvolutionNC(triples=torch.from_numpy(tp), num_nodes=N, num_relations=2 * R0 + 1, in_features=16, out_features=16).to(DEV)
    with torch.no_grad():
        layer.bias.normal_()
        a, b = torch.randn(N, 16, device=DEV), torch.randn(N, 16, device=DEV)
        _native.profile_start()
        ya, yb, yab = layer(a), layer(b), layer(0.5 * a + 2.0 * b)
        prof = _native.profile_stop()
        assert "spmm_blk" in prof and "spmm_scatter" not in prof, sorted(prof)
        assert rel_err(yab - layer.bias, (0.5 * (ya - layer.bias) + 2.0 * (yb - layer.bias)).cpu().numpy()) < 1e-5
        with routes.override(spmm_csr="0"):
            two = RelationalGraphConvolutionNC(triples=torch.from_numpy(tp), num_nodes=N, num_relations=2 * R0 + 1, in_features=16,
                                               out_features=16).to(DEV)
            two.load_state_dict(layer.state_dict())
            _native.profile_start()
            yt = two(a)
            assert "spmm_scatter" in _native.profile_stop()
        assert rel_err(ya, yt.cpu().numpy()) < 1e-5
        assert torch.equal(layer.forward_activated(a, "relu"), torch.relu(ya))
        layer.weights.fill_(0.0)
        layer.weights[:, 0, 0] = 1.0
        layer.bias.zero_()
        y = layer(torch.ones(N, 16, device=DEV))[:, 0]
        key = torch.from_numpy(tp[:, 1] * N + tp[:, 0]).to(DEV)
        cnt = torch.bincount(torch.unique(key) % N, minlength=N).float()
        assert torch.allclose(y, cnt, rtol=1e-5, atol=1e-5)


def test_am_full_size_featureless_basis40_properties():
    """AM as shipped (nc-AM.yaml: featureless first layer, basis 40, hidden 10): the source-major kernels at full size.
    With comps = 1/B and bases = 1 the output row is the number of (relation, subject) groups of the node; linear in bases."""
    from torch_rgcn.layers import RelationalGraphConvolutionNC
    N, R0, B, d = AM["N"], AM["R0"], 40, 10
    tp = _am_graph()
    layer = RelationalGraphConvolutionNC(triples=torch.from_numpy(tp), num_nodes=N, num_relations=2 * R0 + 1, in_features=None,
                                         out_features=d, bias=False, decomposition={"type": "basis", "num_bases": B}).to(DEV)
    with torch.no_grad():
        layer.comps.fill_(1.0 / B)
        layer.bases.fill_(1.0)
        y = layer()
        key = torch.from_numpy(tp[:, 1] * N + tp[:, 0]).to(DEV)
        cnt = torch.bincount(torch.unique(key) % N, minlength=N).float()
        assert torch.allclose(y[:, 0], cnt, rtol=1e-4, atol=1e-4) and torch.allclose(y[:, d - 1], cnt, rtol=1e-4, atol=1e-4)
        layer.comps.normal_()
        b1 = torch.randn_like(layer.bases)
        layer.bases.copy_(b1)
        y1 = layer()
        layer.bases.mul_(-0.5)
        assert rel_err(layer(), (-0.5 * y1).cpu().numpy()) < 1e-5
    out = layer()
    out.backward(torch.ones_like(out))
    assert torch.isfinite(layer.bases.grad).all() and torch.isfinite(layer.comps.grad).all()
    # d out / d bases[b, o, j] = sum over messages sent by o of val * comps[r, b]: same for every j
    assert torch.allclose(layer.bases.grad[:, :, 0], layer.bases.grad[:, :, d - 1], rtol=1e-4, atol=1e-5)


def _am_samples(seed):
    rng = np.random.default_rng(seed)
    N, R = AM["N"], 2 * AM["R0"] + 1
    return (np.sort(rng.choice(N, 20_000, replace=False)), np.sort(rng.choice(N, 2_000, replace=False)),
            np.sort(np.concatenate([rng.choice(R - 1, 31, replace=False), [R - 1]])))          # 31 relations and the self-loop relation


@pytest.mark.parametrize("mode", ["block", "none"])
def test_am_full_size_featured_layer_sampled_rows_vs_oracle(mode):
    """full-size AM against the ORACLE, not against properties: the featured block-diagonal layer (BASELINE configs[2]:
    block CSR forward, block-tile backward with RGCN_F_DIAG4 on ~400-row tiles) and the dense R = 267 layer (block-tile forward on 930-row
    tiles, relation-major backward) -- out on 20,000 sampled rows, dX on 2,000 sampled sources, the weight-gradient rows of 32 sampled
    relations (self loops included), db; each sampled row sums ALL the messages that touch it (oracle.nc_layer_rows)."""
    from torch_rgcn.layers import RelationalGraphConvolutionNC
    N, R0 = AM["N"], AM["R0"]
    R = 2 * R0 + 1
    tp = _am_graph()
    out_rows, src_rows, rel_rows = _am_samples(41)
    torch.manual_seed(41)
    layer = RelationalGraphConvolutionNC(triples=torch.from_numpy(tp), num_nodes=N, num_relations=R, in_features=16, out_features=16,
                                         decomposition={"type": "block", "num_blocks": 4} if mode == "block" else None).to(DEV)
    with torch.no_grad():
        layer.bias.normal_()
    rng = np.random.default_rng(42)
    Xh = rng.random((N, 16), dtype=np.float32) * 2 - 1
    gh = rng.random((N, 16), dtype=np.float32) * 2 - 1
    X = torch.from_numpy(Xh).to(DEV).requires_grad_(True)
    out = layer(X)
    out.backward(torch.from_numpy(gh).to(DEV))
    params = {n: p.detach().cpu().numpy() for n, p in layer.named_parameters() if n != "bias"}
    ref = oracle.nc_layer_rows(tp, N, R, Xh, params, mode, layer.bias.detach().cpu().numpy(), False, gh, out_rows, src_rows, rel_rows)
    assert rel_err(out[torch.from_numpy(out_rows).to(DEV)], ref["out"]) < TOL
    assert rel_err(X.grad[torch.from_numpy(src_rows).to(DEV)], ref["dX"]) < TOL
    assert rel_err(layer.bias.grad, ref["db"]) < TOL
    name = "blocks" if mode == "block" else "weights"
    assert rel_err(getattr(layer, name).grad[torch.from_numpy(rel_rows).to(DEV)], ref["grads_rows"][name]) < TOL


def test_am_full_size_featureless_basis40_sampled_rows_vs_oracle():
    """AM as shipped, first layer (featureless, basis 40, hidden 10: the 2.67 GB bases table through the in-place tile
    kernels) at FULL size against the oracle on sampled rows: out (20,000 rows), dbases (the rows of 2,000 sampled nodes, all 40 bases),
    dcomps (32 relations incl. self loops), db."""
    from torch_rgcn.layers import RelationalGraphConvolutionNC
    N, R0, B, d = AM["N"], AM["R0"], 40, 10
    R = 2 * R0 + 1
    tp = _am_graph()
    out_rows, src_rows, rel_rows = _am_samples(43)
    torch.manual_seed(43)
    layer = RelationalGraphConvolutionNC(triples=torch.from_numpy(tp), num_nodes=N, num_relations=R, in_features=None, out_features=d,
                                         decomposition={"type": "basis", "num_bases": B}).to(DEV)
    with torch.no_grad():
        layer.bias.normal_()
        layer.comps.normal_()
        layer.bases.uniform_(-1.0, 1.0)
    gh = np.random.default_rng(44).random((N, d), dtype=np.float32) * 2 - 1
    out = layer()
    out.backward(torch.from_numpy(gh).to(DEV))
    params = {"comps": layer.comps.detach().cpu().numpy(), "bases": layer.bases.detach().cpu().numpy()}
    ref = oracle.nc_layer_rows(tp, N, R, None, params, "basis", layer.bias.detach().cpu().numpy(), False, gh, out_rows, src_rows, rel_rows)
    assert rel_err(out[torch.from_numpy(out_rows).to(DEV)], ref["out"]) < TOL
    assert rel_err(layer.bases.grad[:, torch.from_numpy(src_rows).to(DEV), :], ref["dbases_rows"]) < TOL
    assert rel_err(layer.comps.grad[torch.from_numpy(rel_rows).to(DEV)], ref["dcomps_rows"]) < TOL
    assert rel_err(layer.bias.grad, ref["db"]) < TOL


def test_mutag_full_shape_layers_vs_oracle():
    """MUTAG at its full shape: the NodeClassifier's two layers (featureless basis-30 16-wide; featured basis-30 16 -> 2,
    vertical) against the oracle"""
    run_layer_vs_oracle(N=MUTAG["N"], R0=MUTAG["R0"], E=MUTAG["E"], d_in=None, d_out=16, mode="basis", featureless=True,
                        num_bases=30, seed=401)
    run_layer_vs_oracle(N=MUTAG["N"], R0=MUTAG["R0"], E=MUTAG["E"], d_in=16, d_out=2, mode="basis", vertical=True,
                        num_bases=30, seed=402)


def test_mutag_full_shape_node_classifier_matches_oracle_composition():
    """the whole MUTAG-shaped NodeClassifier (featureless basis layer -> fused ReLU -> vertical basis layer): logits and
    the gradient of every parameter against the two oracle layers chained by hand"""
    from torch_rgcn.models import NodeClassifier
    N, R0, E = MUTAG["N"], MUTAG["R0"], MUTAG["E"]
    R = 2 * R0 + 1
    T = oracle.synthetic_triples(N, R0, E, seed=403)
    model = NodeClassifier(triples=T.tolist(), nnodes=N, nrel=R0, nfeat=None, nhid=16, nlayers=2, nclass=2,
                           decomposition={"type": "basis", "num_bases": 30}).to(DEV)
    rng = np.random.default_rng(5)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(torch.from_numpy(rng.standard_normal(tuple(p.shape)).astype(np.float32) * 0.2))
    logits = model()
    g = rng.standard_normal(tuple(logits.shape)).astype(np.float32)
    logits.backward(torch.from_numpy(g).to(DEV))
    tp = oracle.add_inverse_and_self(T, N, R0)
    P = {n: p.detach().cpu().numpy() for n, p in model.named_parameters()}
    l1 = oracle.nc_layer(tp, N, R, None, {"bases": P["rgc1.bases"], "comps": P["rgc1.comps"]}, "basis", P["rgc1.bias"], False, None)
    a = np.maximum(l1["out"], 0)
    l2 = oracle.nc_layer(tp, N, R, a, {"bases": P["rgc2.bases"], "comps": P["rgc2.comps"]}, "basis", P["rgc2.bias"], True, g)
    assert rel_err(logits, l2["out"]) < TOL
    for n, gv in l2["grads"].items():
        assert rel_err(getattr(model.rgc2, n).grad, gv) < TOL, n
    assert rel_err(model.rgc2.bias.grad, l2["db"]) < TOL
    # the ReLU's mask as the GPU saw it: a pre-activation within rounding of 0 may come out on the other side there (this seed has ONE such
    # entry, -2.98e-8 against +2.98e-8 with the tile kernels' summation order) -- then the two gradients legitimately differ by that row
    with torch.no_grad():
        mask = model.rgc1.forward_activated(None, "relu").cpu().numpy() > 0
    flipped = mask != (l1["out"] > 0)
    assert flipped.sum() <= 3 and (np.abs(l1["out"][flipped]) < 1e-6).all(), (int(flipped.sum()), l1["out"][flipped])
    l1b = oracle.nc_layer(tp, N, R, None, {"bases": P["rgc1.bases"], "comps": P["rgc1.comps"]}, "basis", P["rgc1.bias"], False,
                          (l2["dX"] * mask).astype(np.float32))
    for n, gv in l1b["grads"].items():
        assert rel_err(getattr(model.rgc1, n).grad, gv) < TOL, n


def test_more_than_2_31_relation_node_cells_vs_oracle():
    """num_nodes x num_relations past 2^31 (17 M nodes x 129 relations = 2.19e9 cells of the counting
    tables, 64-bit cell indices; also past the 24-bit source id of the packed slots -> unpacked slot arrays): one featured
    layer, forward and backward, against the oracle.  (27 M x 81 until round 5: the same two limits crossed with 37 % fewer rows of
    random numbers, oracle zero-fills and float64 comparisons -- 27 s of the suite's 184.)"""
    run_layer_vs_oracle(N=17_000_000, R0=64, E=3_000_000, d_in=16, d_out=16, mode="none", seed=77)


@pytest.mark.parametrize("tile_mode", ["1", "ranges"])
def test_am_as_shipped_tenth_scale_first_layer_vs_oracle(monkeypatch, tile_mode):
    """AM as the reference ships it (nc-AM.yaml: featureless first layer, basis 40, hidden 10) at 1/10 scale -- 166,676 nodes, 267
    relations, a 267 MB table: the in-place tile kernels (one wave per node on the matrix cores by default on this hub-free graph;
    `ranges`: a tile's messages dealt over the waves), ~650 tiles per workgroup pipeline, against the oracle: output and both gradients"""
    from torch_rgcn import _native, routes
    routes.patch(monkeypatch, "fbasis_tile", tile_mode)
    _native.profile_start()
    run_layer_vs_oracle(N=AM["N"] // 10, R0=AM["R0"], E=AM["E"] // 10, d_in=None, d_out=10, mode="basis", featureless=True, num_bases=40, seed=410)
    prof = _native.profile_stop()
    assert "fbasis_tile_fwd" in prof and "fbasis_tile_bwd" in prof, sorted(prof)


def test_am_as_shipped_tenth_scale_node_classifier_matches_oracle_composition():
    """the WHOLE AM-as-shipped NodeClassifier at 1/10 scale (nc-AM.yaml: featureless basis-40 layer, hidden 10 ->
    fused ReLU -> 10 -> 11 layer at R = 267, zero-padded to 16 x 16, on the sparse-bucket route -> logits) chained like the MUTAG test above:
    logits and the gradient of every parameter of both layers against the two oracle layers composed by hand
    (/root/reference/torch_rgcn/models.py:186-196, layers.py:241-242,286-306)."""
    from torch_rgcn import _native
    from torch_rgcn.models import NodeClassifier
    N, R0, E = AM["N"] // 10, AM["R0"], AM["E"] // 10
    R = 2 * R0 + 1
    T = oracle.synthetic_triples(N, R0, E, seed=411)
    model = NodeClassifier(triples=T.tolist(), nnodes=N, nrel=R0, nfeat=None, nhid=10, nlayers=2, nclass=11,
                           decomposition={"type": "basis", "num_bases": 40}).to(DEV)
    rng = np.random.default_rng(6)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(torch.from_numpy(rng.standard_normal(tuple(p.shape)).astype(np.float32) * 0.2))
    _native.profile_start()
    logits = model()
    g = rng.standard_normal(tuple(logits.shape)).astype(np.float32)
    logits.backward(torch.from_numpy(g).to(DEV))
    prof = _native.profile_stop()
    assert "fbasis_tile_fwd" in prof and "fbasis_tile_bwd" in prof, sorted(prof)          # layer 1 on the in-place tile kernels
    tp = oracle.add_inverse_and_self(T, N, R0)
    P = {n: p.detach().cpu().numpy() for n, p in model.named_parameters()}
    l1 = oracle.nc_layer(tp, N, R, None, {"bases": P["rgc1.bases"], "comps": P["rgc1.comps"]}, "basis", P["rgc1.bias"], False, None)
    a = np.maximum(l1["out"], 0)
    l2 = oracle.nc_layer(tp, N, R, a, {"bases": P["rgc2.bases"], "comps": P["rgc2.comps"]}, "basis", P["rgc2.bias"], True, g)
    assert tuple(logits.shape) == (N, 11) and rel_err(logits, l2["out"]) < TOL
    for n, gv in l2["grads"].items():
        assert rel_err(getattr(model.rgc2, n).grad, gv) < TOL, n
    assert rel_err(model.rgc2.bias.grad, l2["db"]) < TOL
    with torch.no_grad():       # the ReLU's mask as the GPU saw it (see the MUTAG test): entries within rounding of 0 may flip
        mask = model.rgc1.forward_activated(None, "relu").cpu().numpy() > 0
    flipped = mask != (l1["out"] > 0)
    assert flipped.sum() <= 3 and (np.abs(l1["out"][flipped]) < 1e-6).all(), (int(flipped.sum()), l1["out"][flipped])
    l1b = oracle.nc_layer(tp, N, R, None, {"bases": P["rgc1.bases"], "comps": P["rgc1.comps"]}, "basis", P["rgc1.bias"], False,
                          (l2["dX"] * mask).astype(np.float32))
    for n, gv in l1b["grads"].items():
        assert rel_err(getattr(model.rgc1, n).grad, gv) < TOL, n
    assert rel_err(model.rgc1.bias.grad, l1b["db"]) < TOL


WN18 = dict(N=40_943, R0=18, d=200, B=2, train_graph=15_000, eval_graph=141_442, scored=330_000)


def _wn18_layer(seed, self_loop_type="schlichtkrull-dropout"):
    from torch_rgcn.layers import RelationalGraphConvolutionLP
    N, R0, d = WN18["N"], WN18["R0"], WN18["d"]
    torch.manual_seed(seed)
    ed = {"general": 0.5, "self_loop": 0.2, "self_loop_type": self_loop_type}
    layer = RelationalGraphConvolutionLP(num_nodes=N, num_relations=2 * R0 + 1, in_features=d, out_features=d, edge_dropout=ed,
                                         decomposition={"type": "basis", "num_bases": WN18["B"]}, w_init="glorot-normal",
                                         b_init="zeros").to(DEV)
    with torch.no_grad():
        layer.bias.normal_(0.0, 0.1)
    return layer


@pytest.mark.parametrize("self_loop_type", ["schlichtkrull-dropout", "bernoulli"])
def test_wn18_full_shape_train_step_vs_oracle(self_loop_type):
    """VERDICT r4 missing #4: lp-WN18.yaml's training step at its REAL size against the oracle -- RelationalGraphConvolutionLP at
    d = 200, basis 2, N = 40,943 on a 15,000-triple graph (training mode; the shipped `schlichtkrull-dropout` keeps every self loop of a
    basis layer, any other type drops 20 % of them: the layer's Bernoulli draw is replayed for the oracle's keep mask) feeding DistMult
    on 330,000 scored triples: encoder output, scores, and -- from one backward through
    both -- d embeddings, dbases, dcomps, dbias of the encoder and the decoder's relation gradient
    (/root/reference/torch_rgcn/layers.py:450-565, :77-98; experiments/predict_links.py:117-157)."""
    from torch_rgcn.layers import DistMult
    N, R0, d = WN18["N"], WN18["R0"], WN18["d"]
    R = 2 * R0 + 1
    layer = _wn18_layer(11, self_loop_type)
    layer.train()
    dm = DistMult(R0, d, N, R0).to(DEV)
    T = oracle.synthetic_triples(N, R0, WN18["train_graph"], seed=3)
    batch = oracle.synthetic_triples(N, R0, WN18["scored"], seed=4)
    X = (torch.randn(N, d, device=DEV) * 0.5).requires_grad_(True)
    from torch_rgcn import _native
    _native.profile_start()
    torch.manual_seed(321)
    H = layer(torch.from_numpy(T), X)
    torch.manual_seed(321)      # the layer's Bernoulli draw again: which self loops it kept
    keep = torch.bernoulli(torch.full((N,), 1.0 if self_loop_type == "schlichtkrull-dropout" else 0.8, dtype=torch.float,
                                      device=DEV)).to(torch.bool).cpu().numpy()
    assert keep.all() == (self_loop_type == "schlichtkrull-dropout")
    H.retain_grad()
    scores = dm(torch.from_numpy(batch).to(DEV), H)
    gs = torch.randn(WN18["scored"], device=DEV) / WN18["scored"]
    scores.backward(gs)
    prof = _native.profile_stop()
    assert "basis_dcomps_csr" in prof and "basis_dcomps" not in prof, sorted(prof)      # per-step graph: dcomps on the forward's CSR, no relation-major plan
    # decoder against the oracle on the GPU's own encoder output ...
    Hn, rel = H.detach().cpu().numpy(), dm.relations.detach().cpu().numpy()
    s_ref = oracle.distmult_forward(batch, Hn, rel)
    assert rel_err(scores, s_ref) < TOL
    dn_ref, dr_ref = oracle.distmult_backward(batch, Hn, rel, gs.cpu().numpy())[:2]
    assert rel_err(H.grad, dn_ref) < TOL and rel_err(dm.relations.grad, dr_ref) < TOL
    # ... and the encoder, forward and backward (upstream gradient = the oracle's d scores / d H), at the full shape
    params = {"bases": layer.bases.detach().cpu().numpy(), "comps": layer.comps.detach().cpu().numpy()}
    ref = oracle.lp_layer(T, N, R, X.detach().cpu().numpy(), params, "basis", layer.bias.detach().cpu().numpy(), False, keep,
                          dn_ref.astype(np.float32))
    assert rel_err(H, ref["out"]) < TOL and rel_err(X.grad, ref["dX"]) < TOL
    assert rel_err(layer.bases.grad, ref["grads"]["bases"]) < TOL and rel_err(layer.comps.grad, ref["grads"]["comps"]) < TOL
    assert rel_err(layer.bias.grad, ref["db"]) < TOL


def test_wn18_full_shape_eval_graph_forward_vs_oracle():
    """lp-WN18.yaml's evaluation encoder pass at its real size: the 141,442-triple graph (/root/reference/utils/misc.py:60-110 encodes
    it once per evaluation, eval mode: every self loop kept, no backward) against the oracle -- 324 k messages at d = 200"""
    N, R0 = WN18["N"], WN18["R0"]
    layer = _wn18_layer(12)
    layer.eval()
    T = oracle.synthetic_triples(N, R0, WN18["eval_graph"], seed=5)
    X = torch.randn(N, WN18["d"], device=DEV) * 0.5
    with torch.no_grad():
        H = layer(torch.from_numpy(T), X)
    params = {"bases": layer.bases.detach().cpu().numpy(), "comps": layer.comps.detach().cpu().numpy()}
    ref = oracle.lp_layer(T, N, 2 * R0 + 1, X.cpu().numpy(), params, "basis", layer.bias.detach().cpu().numpy(), False, None, None)
    assert rel_err(H, ref["out"]) < TOL
