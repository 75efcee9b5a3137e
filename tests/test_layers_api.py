"""Drop-in surface (CPU): constructor signatures, parameter names/shapes (the reference's
tests/test_nn.py pins exactly these sizes), helper utilities' known answers, and the
'no CPU fallback' rule."""
import inspect

import pytest
import torch

from torch_rgcn.layers import DistMult, RelationalGraphConvolutionLP, RelationalGraphConvolutionNC
from torch_rgcn.models import EmbeddingNodeClassifier, NodeClassifier
from torch_rgcn.utils import (add_inverse_and_self, block_diag, drop_edges, generate_inverses, generate_self_loops,
                              select_b_init, select_w_init, stack_matrices, sum_sparse)

TRIPLES = torch.tensor([[1, 0, 2], [3, 0, 0], [3, 1, 0], [0, 1, 3], [2, 2, 1], [1, 2, 3],   # general
                        [0, 3, 0], [1, 3, 1], [2, 3, 2], [3, 3, 3]])                      # self loops


def test_nc_signature_matches_reference():
    sig = inspect.signature(RelationalGraphConvolutionNC.__init__)
    assert list(sig.parameters)[1:] == ["triples", "num_nodes", "num_relations", "in_features", "out_features",
                                        "edge_dropout", "edge_dropout_self_loop", "bias", "decomposition",
                                        "vertical_stacking", "diag_weight_matrix", "reset_mode"]
    assert list(inspect.signature(RelationalGraphConvolutionNC.forward).parameters) == ["self", "features"]
    sig = inspect.signature(RelationalGraphConvolutionLP.__init__)
    assert list(sig.parameters)[1:] == ["num_nodes", "num_relations", "in_features", "out_features", "edge_dropout",
                                        "edge_dropout_self_loop", "decomposition", "vertical_stacking", "w_init",
                                        "w_gain", "b_init"]
    assert list(inspect.signature(RelationalGraphConvolutionLP.forward).parameters) == ["self", "triples", "features"]
    assert list(inspect.signature(DistMult.__init__).parameters)[1:] == ["indim", "outdim", "num_nodes", "num_rel",
                                                                        "w_init", "w_gain", "b_init"]


def test_nc_parameter_shapes():  # reference tests/test_nn.py:22-121
    nn_, nr = 4, 3 * 2 + 1
    a = RelationalGraphConvolutionNC(triples=TRIPLES, num_nodes=nn_, num_relations=nr, in_features=None, out_features=16)
    b = RelationalGraphConvolutionNC(triples=TRIPLES, num_nodes=nn_, num_relations=nr, in_features=16, out_features=16)
    assert a.weights.shape == (7, 4, 16) and b.weights.shape == (7, 16, 16) and a.bias.shape == (16,)
    dec = {'type': 'basis', 'num_bases': 2}
    a = RelationalGraphConvolutionNC(triples=TRIPLES, num_nodes=nn_, num_relations=nr, out_features=16, decomposition=dec)
    b = RelationalGraphConvolutionNC(triples=TRIPLES, num_nodes=nn_, num_relations=nr, in_features=16, out_features=16,
                                     decomposition=dec)
    assert a.bases.shape == (2, 4, 16) and b.bases.shape == (2, 16, 16) and a.comps.shape == (7, 2)
    dec = {'type': 'block', 'num_blocks': 2}
    a = RelationalGraphConvolutionNC(triples=TRIPLES, num_nodes=nn_, num_relations=nr, out_features=16, decomposition=dec)
    b = RelationalGraphConvolutionNC(triples=TRIPLES, num_nodes=nn_, num_relations=nr, in_features=16, out_features=16,
                                     decomposition=dec)
    assert a.blocks.shape == (7, 2, 2, 8) and b.blocks.shape == (7, 2, 8, 8)
    d = RelationalGraphConvolutionNC(triples=TRIPLES, num_nodes=nn_, num_relations=nr, in_features=6, out_features=9,
                                     diag_weight_matrix=True)
    assert d.weights.shape == (7, 6) and d.bias is None and d.out_features == 6
    assert [n for n, _ in a.named_parameters()] == ["blocks", "bias"]


def test_nc_init_and_errors():
    l = RelationalGraphConvolutionNC(triples=TRIPLES, num_nodes=4, num_relations=7, in_features=8, out_features=8)
    assert torch.all(l.bias == 0) and l.weights.abs().max() > 0
    with pytest.raises(NotImplementedError):
        RelationalGraphConvolutionNC(triples=TRIPLES, num_nodes=4, num_relations=7, in_features=8, out_features=8,
                                     decomposition={'type': 'tucker'})
    with pytest.raises(NotImplementedError):
        RelationalGraphConvolutionNC(triples=TRIPLES, num_nodes=4, num_relations=7, in_features=8, out_features=8,
                                     reset_mode='nope')
    with pytest.raises(AssertionError):
        RelationalGraphConvolutionNC(triples=TRIPLES, num_nodes=4, num_relations=7, in_features=9, out_features=8,
                                     decomposition={'type': 'block', 'num_blocks': 2})
    with pytest.raises(AttributeError):  # upstream: 'uniform' reads self.weights, absent under decomposition
        RelationalGraphConvolutionNC(triples=TRIPLES, num_nodes=4, num_relations=7, in_features=8, out_features=8,
                                     decomposition={'type': 'basis', 'num_bases': 2}, reset_mode='uniform')
    with pytest.raises(AssertionError):
        l()  # in_features given but no features


def test_lp_parameter_shapes_and_inits():
    ed = {"general": 0.5, "self_loop": 0.2, "self_loop_type": "schlichtkrull-dropout"}
    l = RelationalGraphConvolutionLP(num_nodes=10, num_relations=7, in_features=8, out_features=6, edge_dropout=ed,
                                     decomposition={'type': 'block', 'num_blocks': 2}, b_init='zeros')
    assert l.blocks.shape == (6, 2, 4, 3) and l.blocks_self.shape == (8, 6) and l.bias.shape == (6,)
    l = RelationalGraphConvolutionLP(num_nodes=10, num_relations=7, in_features=8, out_features=6, edge_dropout=ed,
                                     decomposition={'type': 'basis', 'num_bases': 3})
    assert l.bases.shape == (3, 8, 6) and l.comps.shape == (7, 3) and l.bias is None
    l = RelationalGraphConvolutionLP(num_nodes=10, num_relations=7, in_features=None, out_features=6, edge_dropout=ed)
    assert l.weights.shape == (7, 10, 6) and l.in_features == 10  # never None upstream
    with pytest.raises(AssertionError):
        l(torch.zeros(1, 3, dtype=torch.long))  # features missing


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the CPU-tensor refusal")
def test_no_cpu_fallback():
    l = RelationalGraphConvolutionNC(triples=TRIPLES, num_nodes=4, num_relations=7, in_features=8, out_features=8)
    with pytest.raises(RuntimeError, match="no CPU path"):
        l(torch.randn(4, 8))
    ed = {"general": 0.5, "self_loop": 0.2, "self_loop_type": "schlichtkrull-dropout"}
    lp = RelationalGraphConvolutionLP(num_nodes=4, num_relations=7, in_features=8, out_features=6, edge_dropout=ed)
    with pytest.raises(RuntimeError, match="no CPU path"):
        lp(TRIPLES[:6], torch.randn(4, 8))
    dm = DistMult(3, 8, 4, 3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        dm(TRIPLES[:6], torch.randn(4, 8))


def test_models_construct():
    T = [[0, 0, 1], [1, 1, 2], [2, 0, 3]]
    m = NodeClassifier(triples=T, nnodes=4, nrel=2, nhid=8, nclass=3)
    assert m.triples_plus.shape == (2 * 3 + 4, 3) and m.rgc1.weights.shape == (5, 4, 8) and m.rgc2.weights.shape == (5, 8, 3)
    assert m.rgc1.vertical_stacking is False and m.rgc2.vertical_stacking is True
    e = EmbeddingNodeClassifier(triples=T, nnodes=4, nrel=2, nclass=3, nemb=6)
    assert e.rgcn_no_hidden.weights.shape == (5, 6) and e.rgc1.weights.shape == (5, 6, 3) and e.node_embeddings.shape == (4, 6)


# ---- utilities: known answers of the reference's tests/test_utils.py, restated -------------------

def test_utils_add_inverse_and_self():
    t = torch.tensor([[0, 0, -1], [1, 1, -2], [2, 2, -3]])
    exp = torch.tensor([[0, 0, -1], [1, 1, -2], [2, 2, -3], [-1, 3, 0], [-2, 4, 1], [-3, 5, 2],
                        [0, 6, 0], [1, 6, 1], [2, 6, 2]])
    assert torch.equal(add_inverse_and_self(t, 3, 3), exp)
    assert torch.equal(generate_inverses(t, 3), exp[3:6])
    assert torch.equal(generate_self_loops(t, 3, 3, 1.0), torch.cat([t, exp[6:]]))
    assert generate_self_loops(t, 3, 3, 0.0).shape == (3, 3)


def test_utils_stack_and_sum():
    t = torch.tensor([[0, 0, 3], [1, 1, 4], [2, 2, 5], [3, 3, 0], [4, 4, 1], [5, 5, 2],
                      [0, 6, 0], [1, 6, 1], [2, 6, 2], [3, 6, 3], [4, 6, 4], [5, 6, 5]])
    vi, vs = stack_matrices(t, 9, 7, vertical_stacking=True)
    assert vs == (63, 9) and vi[:, 0].tolist() == [0, 10, 20, 30, 40, 50, 54, 55, 56, 57, 58, 59]
    hi, hs = stack_matrices(t, 9, 7, vertical_stacking=False)
    assert hs == (9, 63) and hi[:, 1].tolist() == [3, 13, 23, 27, 37, 47, 54, 55, 56, 57, 58, 59]
    with pytest.raises(AssertionError):
        stack_matrices(t, 5, 7)
    ver = torch.tensor([[0, 0], [0, 1], [0, 2], [4, 1], [8, 2], [7, 2]])
    v = torch.ones(6) / sum_sparse(ver, torch.ones(6), (9, 3), row_normalisation=True)
    assert torch.equal(v, torch.tensor([1 / 3, 1 / 3, 1 / 3, 1, 1, 1]))
    hor = torch.tensor([[0, 0], [1, 0], [2, 0], [3, 0], [1, 4], [2, 8], [2, 7]])
    v = torch.ones(7) / sum_sparse(hor, torch.ones(7), (4, 9), row_normalisation=False)
    assert torch.equal(v, torch.tensor([.25, .25, .25, .25, 1, 1, 1]))


def test_utils_drop_edges_counts():  # reference tests/test_utils.py:126-167 (counts only)
    t = add_inverse_and_self(torch.tensor([[0, 0, 1], [1, 1, 2], [2, 0, 0]]), 6, 2)
    out = drop_edges(t, 6, 0.5, 0.5)
    assert out.shape == (3 + 3, 3) and int((out[:, 1] == 4).sum()) == 3


def test_utils_block_diag_and_inits():
    m = torch.arange(24.).view(2, 2, 3, 2)
    bd = block_diag(m)
    assert bd.shape == (2, 6, 4)
    assert torch.equal(bd[1, :3, :2], m[1, 0]) and torch.equal(bd[1, 3:, 2:], m[1, 1]) and bd[1, :3, 2:].abs().sum() == 0
    assert block_diag(torch.ones(4, 3, 2)).shape == (12, 8)
    assert select_w_init('Glorot-Uniform') is torch.nn.init.xavier_uniform_
    assert select_b_init('zeros') is torch.nn.init.zeros_
    with pytest.raises(NotImplementedError):
        select_w_init('he')


def test_forward_activated_goes_through_module_call_hooks():
    """the fused-activation entry the models use must fire forward hooks like `F.relu(self.rgc1(...))` does
    (checked on the CPU: the hook fires before the layer refuses CPU parameters)."""
    import torch
    from torch_rgcn.layers import RelationalGraphConvolutionNC
    tp = torch.tensor([[0, 0, 1], [1, 1, 0], [0, 2, 0], [1, 2, 1]])
    layer = RelationalGraphConvolutionNC(triples=tp, num_nodes=2, num_relations=3, in_features=4, out_features=4)
    seen = []
    layer.register_forward_pre_hook(lambda mod, args, kwargs: seen.append(("pre", kwargs.get("features") is not None)), with_kwargs=True)
    try:
        layer.forward_activated(torch.zeros(2, 4), "relu")
    except RuntimeError:
        pass          # no GPU here: the layer has no CPU path -- the hook must have fired first
    assert seen == [("pre", True)] and "_fused_activation" not in layer.__dict__
