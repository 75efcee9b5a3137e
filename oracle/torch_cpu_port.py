"""PyTorch-CPU port of the reference's op sequence for one RGC layer.

TEST INFRASTRUCTURE ONLY (see oracle/rgcn_oracle.c).  This is what bench.py times
as `cpu_baseline` (kind "port"): the reference's Python files cannot travel to the
GPU box, so the same ATen calls are issued here in the same order --

    stacked COO indices            torch_rgcn/utils.py:143-166
    degree = sparse @ ones, gather torch_rgcn/utils.py:71-97
    block swap (horizontal)        torch_rgcn/layers.py:267-273
    horizontal: einsum('ni,rio->rno') then sparse(N, R*N) @ dense(R*N, d_out)
                                   torch_rgcn/layers.py:298-301
    vertical:   sparse(R*N, N) @ X then einsum('rio,rni->no')
                                   torch_rgcn/layers.py:293-297
    featureless: sparse(N, R*N) @ W.view(R*N, d_out)   torch_rgcn/layers.py:286-288

tests/test_cpu_port.py checks it against the golden vectors (and, in the build
container, against the imported reference for value AND wall time).
"""
import torch


def _coo(rows, cols, vals, shape):
    return torch.sparse_coo_tensor(torch.stack([rows, cols]), vals, shape, check_invariants=False)


def layer_cpu(triples_plus, num_nodes, num_rels, features, weights, bias=None, vertical=False,
              n_swap=None, i_tail=None):
    """One layer, dense (R, d_in, d_out) weights (or (R, N, d_out) when featureless)."""
    N, R = num_nodes, num_rels
    s, p, o = triples_plus[:, 0], triples_plus[:, 1], triples_plus[:, 2]
    M = triples_plus.size(0)
    if vertical:
        rows, cols, shape = p * N + s, o, (R * N, N)
    else:
        rows, cols, shape = s, p * N + o, (N, R * N)
    ones = torch.ones(M)
    if vertical:
        deg = torch.mm(_coo(rows, cols, ones, shape), torch.ones(shape[1], 1))[rows, 0]
    else:
        degT = torch.mm(_coo(cols, rows, ones, (shape[1], shape[0])), torch.ones(shape[0], 1))[cols, 0]
        n = int((M - N) / 2) if n_swap is None else n_swap
        i = N if i_tail is None else i_tail
        deg = torch.cat([degT[n:2 * n], degT[:n], degT[M - i:]])
    adj = _coo(rows, cols, ones / deg, shape)
    d_out = weights.size(-1)
    if features is None:
        out = torch.mm(adj, weights.reshape(R * N, d_out))
    elif vertical:
        agg = torch.mm(adj, features).view(R, N, -1)
        out = torch.einsum('rio,rni->no', weights, agg)
    else:
        xw = torch.einsum('ni,rio->rno', features, weights).contiguous()
        out = torch.mm(adj, xw.view(R * N, d_out))
    if bias is not None:
        out = out + bias
    return out


def two_layer_step(triples_plus, num_nodes, num_rels, X, w1, b1, w2, b2):
    """The S1 step: layer 1 horizontal -> relu -> layer 2 vertical -> mean(out^2) -> backward.
    Tensors that need gradients must have requires_grad set by the caller."""
    h = layer_cpu(triples_plus, num_nodes, num_rels, X, w1, b1, vertical=False)
    out = layer_cpu(triples_plus, num_nodes, num_rels, torch.relu(h), w2, b2, vertical=True)
    loss = out.pow(2).mean()
    loss.backward()
    return loss.detach()
