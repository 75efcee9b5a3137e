"""Host-side helpers with the names `torch_rgcn.utils` exposes in the reference.

The message-passing layers themselves do NOT go through the index helpers below
(they use the native relation-tile plans, see graph.py); the helpers are kept so
that callers and tests written against the reference's module keep working:

    add_inverse_and_self   reference utils.py:127-141      generate_inverses   :100-107
    generate_self_loops    :110-124                        stack_matrices      :143-166
    sum_sparse             :71-97                          block_diag          :168-196
    split_spo              :201-206                        drop_edges          :57-69
    select_w_init / select_b_init / schlichtkrull_*        :6-55
"""
import math
import random

import torch


# ----------------------------------------------------------------------------- initialisers

def schlichtkrull_std(shape, gain):
    """gain * 3 / sqrt(fan_in + fan_out) with (fan_in, fan_out) = shape[:2]."""
    return gain * 3.0 / math.sqrt(float(shape[0] + shape[1]))


def schlichtkrull_normal_(tensor, shape, gain=1.):
    with torch.no_grad():
        return tensor.normal_(0.0, schlichtkrull_std(shape, gain))


def schlichtkrull_uniform_(tensor, gain=1.):
    # As upstream, the tensor itself is handed over where a (fan_in, fan_out) pair is
    # expected, so this raises for real tensors; kept for parity.
    bound = schlichtkrull_std(tensor, gain)
    with torch.no_grad():
        return tensor.uniform_(-bound, bound)


_W_INITS = {
    'glorot-uniform': torch.nn.init.xavier_uniform_, 'xavier-uniform': torch.nn.init.xavier_uniform_,
    'glorot-normal': torch.nn.init.xavier_normal_, 'xavier-normal': torch.nn.init.xavier_normal_,
    'schlichtkrull-uniform': schlichtkrull_uniform_, 'schlichtkrull-normal': schlichtkrull_normal_,
    'normal': torch.nn.init.normal_, 'standard-normal': torch.nn.init.normal_,
    'uniform': torch.nn.init.uniform_,
}
_B_INITS = {
    'zeros': torch.nn.init.zeros_, 'zero': torch.nn.init.zeros_,
    'ones': torch.nn.init.ones_, 'one': torch.nn.init.ones_,
    'uniform': torch.nn.init.uniform_, 'normal': torch.nn.init.normal_,
}


def select_w_init(init):
    try:
        return _W_INITS[init.lower()]
    except KeyError:
        raise NotImplementedError(f'{init} initialisation has not been implemented!')


def select_b_init(init):
    try:
        return _B_INITS[init.lower()]
    except KeyError:
        raise NotImplementedError(f'{init} initialisation has not been implemented!')


# ----------------------------------------------------------------------------- triples

def split_spo(triples):
    """(..., 3) -> subject, predicate, object views"""
    return triples[..., 0], triples[..., 1], triples[..., 2]


def generate_inverses(triples, num_rels):
    s, p, o = split_spo(triples)
    return torch.stack((o, p + num_rels, s), dim=1)


def _self_loop_block(num_nodes, num_rels, device):
    ids = torch.arange(num_nodes, device=device, dtype=torch.long)
    return torch.stack((ids, torch.full_like(ids, 2 * num_rels), ids), dim=1)


def generate_self_loops(triples, num_nodes, num_rels, self_loop_keep_prob, device='cpu'):
    """`triples` followed by the Bernoulli-kept self loops (the original block is part of
    the result, which is what makes the LP normalisation non-standard: SURVEY.md F5)."""
    loops = _self_loop_block(num_nodes, num_rels, device)
    probs = torch.full((num_nodes,), float(self_loop_keep_prob), dtype=torch.float, device=device)
    kept = torch.bernoulli(probs).to(torch.bool)
    return torch.cat((triples, loops[kept]), dim=0)


def add_inverse_and_self(triples, num_nodes, num_rels, device='cpu'):
    """[T | inverse(T) | (i, 2R, i) for every node] in this fixed block order."""
    return torch.cat((triples, generate_inverses(triples, num_rels),
                      _self_loop_block(num_nodes, num_rels, device).to(triples.device)), dim=0)


def drop_edges(triples, num_nodes, general_edo, self_loop_edo):
    """Keeps floor(keep * count) random rows of the general block and of the trailing
    self-loop block (self loops are the last num_nodes rows)."""
    n_general = triples.size(0) - num_nodes
    keep_g = random.sample(range(n_general), k=int(math.floor((1.0 - general_edo) * n_general)))
    keep_s = random.sample(range(n_general, n_general + num_nodes),
                           k=int(math.floor((1.0 - self_loop_edo) * num_nodes)))
    return triples[keep_g + keep_s, :]


def stack_matrices(triples, num_nodes, num_rels, vertical_stacking=True, device='cpu'):
    """COO indices of the R stacked adjacency matrices: rows are subjects.
    vertical: (p*N + s, o), size (R*N, N); horizontal: (s, p*N + o), size (N, R*N)."""
    assert triples.dtype == torch.long
    s, p, o = split_spo(triples)
    shift = p * num_nodes
    if vertical_stacking:
        size, rows, cols = (num_rels * num_nodes, num_nodes), shift + s, o
    else:
        size, rows, cols = (num_nodes, num_rels * num_nodes), s, shift + o
    indices = torch.stack((rows, cols), dim=1).to(device)
    assert indices.size(0) == triples.size(0)
    if indices.numel():
        assert indices[:, 0].max() < size[0], f'{indices[:, 0].max()}, {size}, {num_rels}'
        assert indices[:, 1].max() < size[1], f'{indices[:, 1].max()}, {size}, {num_rels}'
    return indices, size


def sum_sparse(indices, values, size, row_normalisation=True, device='cpu'):
    """For every entry, the sum of `values` over the entries of its row (or column).
    Done as a segmented sum (index_add) instead of the reference's sparse @ ones."""
    assert indices.dim() == values.dim() + 1
    key = indices[:, 0] if row_normalisation else indices[:, 1]
    n = size[0] if row_normalisation else size[1]
    totals = torch.zeros(n, dtype=values.dtype, device=values.device).index_add_(0, key.to(values.device), values)
    return totals[key.to(values.device)].view(indices.size(0))


def block_diag(m):
    """(..., nb, bi, bo) -> (..., nb*bi, nb*bo) with the nb blocks on the diagonal."""
    if isinstance(m, (list, tuple)):
        m = torch.stack(list(m), dim=-3)
    *lead, nb, bi, bo = m.shape
    out = m.new_zeros(*lead, nb, bi, nb, bo)
    # the diagonal blocks as a strided VIEW (..., bi, bo, nb) of the zeros: one copy forward, one strided read backward -- the advanced-index
    # assignment this replaces (out[..., idx, :, idx, :] = m) cost an arange, two index computations, an index_put and, backward, an index
    out.diagonal(dim1=-4, dim2=-2).copy_(m.movedim(-3, -1))
    return out.reshape(*lead, nb * bi, nb * bo)


def attach_dim(v, n_dim_to_prepend=0, n_dim_to_append=0):
    """view of `v` with singleton dimensions added in front / at the back (utils.py:198-199)"""
    return v.reshape((1,) * n_dim_to_prepend + tuple(v.shape) + (1,) * n_dim_to_append)
