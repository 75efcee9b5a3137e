"""numpy front-end of the CPU oracle (oracle/rgcn_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of rgcn_oracle.c.  The shipped package
under torch-rgcn_amd/ never imports this module; tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg do, as the checker.

What lives here rather than in C: the expansion of decomposed weights into the
dense (R, d_in, d_out) tensor the C loops consume, and the chain rule back to the
decomposed parameters.  Reference lines:
  basis  W_r = sum_b comps[r,b] bases[b]        torch_rgcn/layers.py:241-242, :468-469
  block  W_r = blockdiag(blocks[r])             torch_rgcn/layers.py:243-244, utils.py:168-196
         LP: [blockdiag(blocks) ; blocks_self]  torch_rgcn/layers.py:375-378, :520-528
  diag   W_r = diag(weights[r])                 torch_rgcn/layers.py:147-151, :289-292
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "librgcn_oracle.so")
_lib = None

_I64P = ctypes.POINTER(ctypes.c_int64)
_F32P = ctypes.POINTER(ctypes.c_float)
_U8P = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile rgcn_oracle.c with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "rgcn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "_build/librgcn_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


def _check(rc, what):
    if rc == 3:
        raise AssertionError(f"{what}: index out of range")
    if rc == 1:
        raise AssertionError(f"{what}: invalid argument / shape mismatch")
    if rc != 0:
        raise RuntimeError(f"{what}: oracle error {rc}")


# --------------------------------------------------------------------------- index utilities

def add_inverse_and_self(triples, num_nodes, num_rels):
    T = _i64(triples).reshape(-1, 3)
    out = np.empty((2 * T.shape[0] + num_nodes, 3), np.int64)
    _check(lib().oracle_add_inverse_and_self(_p(T, _I64P), ctypes.c_int64(T.shape[0]), ctypes.c_int64(num_nodes),
                                            ctypes.c_int64(num_rels), _p(out, _I64P)), "add_inverse_and_self")
    return out


def lp_augment(triples, num_nodes, num_rels, keep_mask=None):
    """-> (triples_plus [T|inv|T|SL], n_self = E + #kept)."""
    T = _i64(triples).reshape(-1, 3)
    E = T.shape[0]
    out = np.empty((3 * E + num_nodes, 3), np.int64)
    keep = None if keep_mask is None else np.ascontiguousarray(keep_mask, dtype=np.uint8)
    M = ctypes.c_int64(0)
    ns = ctypes.c_int64(0)
    _check(lib().oracle_lp_augment(_p(T, _I64P), ctypes.c_int64(E), ctypes.c_int64(num_nodes),
                                  ctypes.c_int64(num_rels), _p(keep, _U8P), _p(out, _I64P),
                                  ctypes.byref(M), ctypes.byref(ns)), "lp_augment")
    return out[:M.value].copy(), ns.value


def stack_matrices(triples_plus, num_nodes, num_rels, vertical_stacking=True):
    Tp = _i64(triples_plus).reshape(-1, 3)
    idx = np.empty((Tp.shape[0], 2), np.int64)
    size = np.empty(2, np.int64)
    _check(lib().oracle_stack_matrices(_p(Tp, _I64P), ctypes.c_int64(Tp.shape[0]), ctypes.c_int64(num_nodes),
                                      ctypes.c_int64(num_rels), ctypes.c_int(int(vertical_stacking)),
                                      _p(idx, _I64P), _p(size, _I64P)), "stack_matrices")
    return idx, (int(size[0]), int(size[1]))


def sum_sparse(indices, values, size=None, row_normalisation=True):
    idx = _i64(indices).reshape(-1, 2)
    vals = None if values is None else _f32(values)
    sums = np.empty(idx.shape[0], np.float32)
    _check(lib().oracle_sum_sparse(_p(idx, _I64P), _p(vals, _F32P), ctypes.c_int64(idx.shape[0]),
                                  ctypes.c_int(int(row_normalisation)), _p(sums, _F32P)), "sum_sparse")
    return sums


def edge_norm(triples_plus, num_nodes, num_rels, vertical, n_swap, i_tail):
    Tp = _i64(triples_plus).reshape(-1, 3)
    val = np.empty(Tp.shape[0], np.float32)
    _check(lib().oracle_edge_norm(_p(Tp, _I64P), ctypes.c_int64(Tp.shape[0]), ctypes.c_int64(num_nodes),
                                 ctypes.c_int64(num_rels), ctypes.c_int(int(vertical)), ctypes.c_int64(n_swap),
                                 ctypes.c_int64(i_tail), _p(val, _F32P)), "edge_norm")
    return val


def nc_edge_norm(triples_plus, num_nodes, num_rels, vertical):
    """NC layer's (n, i): layers.py:235-236."""
    M = len(triples_plus)
    return edge_norm(triples_plus, num_nodes, num_rels, vertical, int((M - num_nodes) / 2), num_nodes)


# --------------------------------------------------------------------------- dense-W layer

def rgcn_forward(triples_plus, val, N, R, X, W, bias=None):
    Tp = _i64(triples_plus).reshape(-1, 3)
    val = _f32(val)
    W = _f32(W)
    if X is None:
        d_in, d_out = N, W.shape[-1]
        assert W.size == R * N * d_out
    else:
        X = _f32(X)
        d_in, d_out = X.shape[1], W.shape[-1]
        assert W.size == R * d_in * d_out and X.shape[0] == N
    b = None if bias is None else _f32(bias)
    out = np.empty((N, d_out), np.float32)
    _check(lib().oracle_rgcn_forward(_p(Tp, _I64P), _p(val, _F32P), ctypes.c_int64(Tp.shape[0]), ctypes.c_int64(N),
                                    ctypes.c_int64(R), ctypes.c_int64(d_in), ctypes.c_int64(d_out), _p(X, _F32P),
                                    _p(W, _F32P), _p(b, _F32P), _p(out, _F32P)), "rgcn_forward")
    return out


def rgcn_backward(triples_plus, val, N, R, X, W, g, need_dx=True):
    """-> (dX or None, dW shaped like W, db)."""
    Tp = _i64(triples_plus).reshape(-1, 3)
    val = _f32(val)
    W = _f32(W)
    g = _f32(g)
    d_out = W.shape[-1]
    d_in = N if X is None else X.shape[1]
    Xc = None if X is None else _f32(X)
    dX = np.zeros((N, d_in), np.float32) if (X is not None and need_dx) else None
    dW = np.zeros(W.shape, np.float32)
    db = np.zeros(d_out, np.float32)
    _check(lib().oracle_rgcn_backward(_p(Tp, _I64P), _p(val, _F32P), ctypes.c_int64(Tp.shape[0]), ctypes.c_int64(N),
                                     ctypes.c_int64(R), ctypes.c_int64(d_in), ctypes.c_int64(d_out), _p(Xc, _F32P),
                                     _p(W, _F32P), _p(g, _F32P), _p(dX, _F32P), _p(dW, _F32P), _p(db, _F32P)),
           "rgcn_backward")
    return dX, dW, db


# --------------------------------------------------------------------------- weight expansion + chain rule

def block_diag_np(blocks):
    """(R, nb, bi, bo) -> (R, nb*bi, nb*bo)."""
    R, nb, bi, bo = blocks.shape
    W = np.zeros((R, nb * bi, nb * bo), np.float32)
    for b in range(nb):
        W[:, b * bi:(b + 1) * bi, b * bo:(b + 1) * bo] = blocks[:, b]
    return W


def expand_weights(params, mode):
    """params: dict of numpy arrays named as the module's parameters -> dense W."""
    if mode == "none":
        return _f32(params["weights"])
    if mode == "basis":
        # einsum("rb,bio->rio") spelled as one matrix product [R, B] x [B, i o]: the same float64 sums on BLAS (np.einsum walks this
        # contraction element by element -- 45 s for AM's table at 1/10 scale against 7)
        comps, bases = params["comps"].astype(np.float64), params["bases"].astype(np.float64)
        return (comps @ bases.reshape(bases.shape[0], -1)).reshape((comps.shape[0],) + bases.shape[1:]).astype(np.float32)
    if mode == "block":
        W = block_diag_np(_f32(params["blocks"]))
        if "blocks_self" in params:  # LP variant: dense self-loop weight appended last
            W = np.concatenate([W, _f32(params["blocks_self"])[None]], axis=0)
        return W
    if mode == "diag":
        w = _f32(params["weights"])
        W = np.zeros((w.shape[0], w.shape[1], w.shape[1]), np.float32)
        i = np.arange(w.shape[1])
        W[:, i, i] = w
        return W
    raise NotImplementedError(mode)


def contract_weight_grads(dW, params, mode):
    """dense dW -> grads of the decomposed parameters (float64 maths, fp32 out)."""
    dW64 = dW.astype(np.float64)
    if mode == "none":
        return {"weights": dW}
    if mode == "basis":
        comps = params["comps"].astype(np.float64)
        bases = params["bases"].astype(np.float64)
        flat = dW64.reshape(dW64.shape[0], -1)              # einsum("rb,rio->bio") and ("rio,bio->rb") as matrix products
        return {"bases": (comps.T @ flat).reshape(bases.shape).astype(np.float32),
                "comps": (flat @ bases.reshape(bases.shape[0], -1).T).astype(np.float32)}
    if mode == "block":
        blocks = params["blocks"]
        Rb, nb, bi, bo = blocks.shape
        gb = np.empty_like(blocks, dtype=np.float32)
        for b in range(nb):
            gb[:, b] = dW[:Rb, b * bi:(b + 1) * bi, b * bo:(b + 1) * bo]
        out = {"blocks": gb}
        if "blocks_self" in params:
            out["blocks_self"] = dW[Rb].copy()
        return out
    if mode == "diag":
        i = np.arange(dW.shape[1])
        return {"weights": dW[:, i, i].copy()}
    raise NotImplementedError(mode)


def layer(triples_plus, val, N, R, X, params, mode, bias=None, g=None, need_dx=True):
    """One RGC layer on an explicit edge list.  Returns dict(out=..., and, when g is
    given, dX, db and one grad per parameter)."""
    W = expand_weights(params, mode)
    res = {"out": rgcn_forward(triples_plus, val, N, R, X, W, bias)}
    if g is not None:
        dX, dW, db = rgcn_backward(triples_plus, val, N, R, X, W, g, need_dx)
        res["dX"] = dX
        res["db"] = db
        res["grads"] = contract_weight_grads(dW, params, mode)
    return res


def nc_layer(triples_plus, N, R, X, params, mode, bias=None, vertical=False, g=None):
    """RelationalGraphConvolutionNC.forward (layers.py:222-308) + autograd duals."""
    val = nc_edge_norm(triples_plus, N, R, vertical)
    return layer(triples_plus, val, N, R, X, params, mode, bias, g)


def nc_layer_rows(triples_plus, N, R, X, params, mode, bias=None, vertical=False, g=None, out_rows=None, src_rows=None, rel_rows=None):
    """RelationalGraphConvolutionNC.forward (layers.py:222-308) + autograd duals on SAMPLED rows of a graph too large to walk whole in a
    test (full-size AM: 13.6 M messages, a 2.67 GB bases table): the normalisation is the full graph's (nc_edge_norm), then

      out[out_rows]                          from the messages whose subject is sampled,
      dX[src_rows] / dbases[:, src_rows]     from the messages whose object is sampled,
      the gradient rows of rel_rows          (weights / blocks / comps) from the messages of those relations,
      db                                     = column sums of g.

    Every returned row is exact -- it sums ALL the messages that touch it.  Featured layers go through the same C loops as nc_layer
    (rgcn_forward / rgcn_backward on the filtered message list); the featureless basis layer (layers.py:241-242, 286-288: W = the
    R x N x d table einsum('rb,bio->rio')) is evaluated in float64 numpy without the table.  -> dict of the sampled rows."""
    Tp = _i64(triples_plus).reshape(-1, 3)
    val = nc_edge_norm(Tp, N, R, vertical)
    s, p, o = Tp[:, 0], Tp[:, 1], Tp[:, 2]
    res = {}
    featureless = X is None
    if featureless:
        assert mode == "basis", "featureless sampled rows: basis decomposition (the dense table is what the sampling avoids)"
        comps, bases = params["comps"].astype(np.float64), params["bases"].astype(np.float64)          # [R, B], [B, N, d]
    else:
        W = expand_weights(params, mode)
    if out_rows is not None:
        out_rows = np.asarray(out_rows, np.int64)
        sel = np.isin(s, out_rows)
        if featureless:
            pos = np.searchsorted(np.sort(out_rows), s[sel])
            order = np.argsort(out_rows)
            msg = np.einsum("mb,bmd->md", comps[p[sel]] * val[sel, None].astype(np.float64), bases[:, o[sel], :])
            tmp = np.zeros((len(out_rows), bases.shape[2]), np.float64)
            np.add.at(tmp, pos, msg)
            outv = np.empty_like(tmp)
            outv[order] = tmp
            if bias is not None:
                outv += np.asarray(bias, np.float64)
            res["out"] = outv.astype(np.float32)
        else:
            res["out"] = rgcn_forward(Tp[sel], val[sel], N, R, X, W, bias)[out_rows]
    if g is not None:
        g = _f32(g)
        res["db"] = g.astype(np.float64).sum(0).astype(np.float32)
        if src_rows is not None:
            src_rows = np.asarray(src_rows, np.int64)
            sel = np.isin(o, src_rows)
            if featureless:      # dbases[b, o, :] = sum over the messages sent by o of comps[r, b] val g[s, :]
                order = np.argsort(src_rows)
                pos = np.searchsorted(src_rows[order], o[sel])
                contrib = (comps[p[sel]] * val[sel, None].astype(np.float64))[:, :, None] * g[s[sel]].astype(np.float64)[:, None, :]   # [m, B, d]
                tmp = np.zeros((len(src_rows),) + contrib.shape[1:], np.float64)
                np.add.at(tmp, pos, contrib)
                db_rows = np.empty_like(tmp)
                db_rows[order] = tmp
                res["dbases_rows"] = np.transpose(db_rows, (1, 0, 2)).astype(np.float32)             # [B, rows, d]
            else:
                res["dX"] = rgcn_backward(Tp[sel], val[sel], N, R, X, W, g, True)[0][src_rows]
        if rel_rows is not None:
            rel_rows = np.asarray(rel_rows, np.int64)
            sel = np.isin(p, rel_rows)
            if featureless:      # dcomps[r, b] = sum over the messages of r of val <bases[b, o, :], g[s, :]>
                dc = np.zeros((R, comps.shape[1]), np.float64)
                idx = np.nonzero(sel)[0]
                for a in range(0, len(idx), 1 << 18):
                    ii = idx[a:a + (1 << 18)]
                    part = np.einsum("bmd,md->mb", bases[:, o[ii], :], g[s[ii]].astype(np.float64) * val[ii, None].astype(np.float64))
                    np.add.at(dc, p[ii], part)
                res["dcomps_rows"] = dc[rel_rows].astype(np.float32)
            else:
                dW = rgcn_backward(Tp[sel], val[sel], N, R, X, W, g, False)[1]
                grads = contract_weight_grads(dW, params, mode)
                if mode == "basis":     # (dbases sums over ALL relations: not a per-relation row)
                    res["grads_rows"] = {"comps": grads["comps"][rel_rows]}
                else:
                    res["grads_rows"] = {k: v[rel_rows] for k, v in grads.items() if v.shape[0] >= R - 1}
    return res


def lp_layer(triples, N, R, X, params, mode, bias=None, vertical=False, keep_mask=None, g=None):
    """RelationalGraphConvolutionLP.forward (layers.py:450-565), deterministic parts:
    the self-loop Bernoulli mask is an input; the dense dropout of the block path
    (layers.py:545-546) is not modelled (tests run it with p = 0)."""
    R0 = (R - 1) // 2
    Tp, n_self = lp_augment(triples, N, R0, keep_mask)
    E = len(_i64(triples).reshape(-1, 3))
    val = edge_norm(Tp, N, R, vertical, E, n_self)
    return layer(Tp, val, N, R, X, params, mode, bias, g)


# --------------------------------------------------------------------------- DistMult

def distmult_forward(triples, nodes, relations, sbias=None, pbias=None, obias=None):
    tr = _i64(triples)
    shape = tr.shape[:-1]
    tr = tr.reshape(-1, 3)
    nodes = _f32(nodes)
    rel = _f32(relations)
    sb, pb, ob = (None if b is None else _f32(b) for b in (sbias, pbias, obias))
    sc = np.empty(tr.shape[0], np.float32)
    _check(lib().oracle_distmult_forward(_p(tr, _I64P), ctypes.c_int64(tr.shape[0]), ctypes.c_int64(nodes.shape[0]),
                                        ctypes.c_int64(rel.shape[0]), ctypes.c_int64(nodes.shape[1]),
                                        _p(nodes, _F32P), _p(rel, _F32P), _p(sb, _F32P), _p(pb, _F32P), _p(ob, _F32P),
                                        _p(sc, _F32P)), "distmult_forward")
    return sc.reshape(shape)


def distmult_backward(triples, nodes, relations, gscores, with_bias=False):
    tr = _i64(triples).reshape(-1, 3)
    nodes = _f32(nodes)
    rel = _f32(relations)
    gs = _f32(gscores).reshape(-1)
    dn = np.empty_like(nodes)
    dr = np.empty_like(rel)
    dsb = np.empty(nodes.shape[0], np.float32) if with_bias else None
    dob = np.empty(nodes.shape[0], np.float32) if with_bias else None
    dpb = np.empty(rel.shape[0], np.float32) if with_bias else None
    _check(lib().oracle_distmult_backward(_p(tr, _I64P), ctypes.c_int64(tr.shape[0]), ctypes.c_int64(nodes.shape[0]),
                                         ctypes.c_int64(rel.shape[0]), ctypes.c_int64(nodes.shape[1]),
                                         _p(nodes, _F32P), _p(rel, _F32P), _p(gs, _F32P), _p(dn, _F32P),
                                         _p(dr, _F32P), _p(dsb, _F32P), _p(dpb, _F32P), _p(dob, _F32P)),
           "distmult_backward")
    return dn, dr, dsb, dpb, dob


# --------------------------------------------------------------------------- ranking evaluator
# Restatement of the reference's utils/misc.py:29-110.  `score_fn(toscore)` stands for `model(graph, toscore)[0]`:
# it receives the expanded [bn, N, 3] candidate tensor exactly as the reference builds it (misc.py:78-83).

def generate_true_dict(all_triples):
    """misc.py:29-38: (p, o) -> list of heads, (s, p) -> list of tails, duplicates kept, input order"""
    heads, tails = {}, {}
    for s, p, o in np.asarray(all_triples).tolist():
        heads.setdefault((p, o), []).append(s)
        tails.setdefault((s, p), []).append(o)
    return heads, tails


def filter_indices(batch, true_triples, head=True):
    """misc.py:46-54: (row, entity) pairs of known true completions other than the target"""
    heads, tails = true_triples
    idx = []
    for i, (s, p, o) in enumerate(np.asarray(batch).tolist()):
        if head:
            idx.extend((i, si) for si in heads.get((p, o), []) if si != s)
        else:
            idx.extend((i, oi) for oi in tails.get((s, p), []) if oi != o)
    return np.asarray(idx, np.int64).reshape(-1, 2)


def rank_batch(scores, targets):
    """misc.py:93-101: optimistic rank + half of the other ties, 1-based"""
    scores = np.asarray(scores)
    true = scores[np.arange(len(scores)), targets][:, None]
    raw = (scores > true).sum(1)
    ties = (scores == true).sum(1)
    return raw + (ties - 1) // 2 + 1


def evaluate(score_fn, test_set, true_triples, num_nodes, batch_size=16, hits_at_k=(1, 3, 10), filter_candidates=True):
    """misc.py:60-110 -> (mrr, hits tuple, ranks list); head queries for the whole test set first, then tails"""
    test_set = _i64(test_set)
    ranks = []
    for head in (True, False):
        for fr in range(0, len(test_set), batch_size):
            batch = test_set[fr:fr + batch_size]
            bn = len(batch)
            toscore = np.repeat(batch[:, None, :], num_nodes, axis=1)
            toscore[:, :, 0 if head else 2] = np.arange(num_nodes)[None, :]
            scores = np.array(score_fn(toscore), np.float32).reshape(bn, num_nodes)
            if filter_candidates:
                idx = filter_indices(batch, true_triples, head)
                scores[idx[:, 0], idx[:, 1]] = -np.inf     # misc.py:58 (raises on an empty list in the reference)
            ranks.extend(rank_batch(scores, batch[:, 0] if head else batch[:, 2]).tolist())
    mrr = sum(1.0 / r for r in ranks) / len(ranks)
    hits = tuple(sum(1.0 if r <= k else 0.0 for r in ranks) / len(ranks) for k in hits_at_k)
    return mrr, hits, ranks


# --------------------------------------------------------------------------- edge-neighbourhood sampler
def edge_neighborhood(train_triples, sample_size, num_nodes, rng=np.random):
    """misc.py:125-172, the same sequence of `rng.choice` calls (so the reference's np.random.seed stream reproduces
    its picks); returns the indices of the sampled triples."""
    tr = np.asarray(train_triples)
    ends = [[] for _ in range(num_nodes)]                     # (edge, other vertex) per edge end
    for i, (s, _, o) in enumerate(tr.tolist()):
        ends[s].append((i, o))
        ends[o].append((i, s))
    left = np.array([len(a) for a in ends])                   # sample_counts
    seen = np.zeros(num_nodes, bool)
    picked = np.zeros(len(tr), bool)
    out = np.zeros(sample_size, np.int64)
    for i in range(sample_size):
        w = left * seen
        if w.sum() == 0:
            w = np.ones_like(w)
            w[left == 0] = 0
        v = rng.choice(np.arange(num_nodes), p=w / w.sum())
        seen[v] = True
        e, other = ends[v][rng.choice(np.arange(len(ends[v])))]
        while picked[e]:
            e, other = ends[v][rng.choice(np.arange(len(ends[v])))]
        out[i] = e
        picked[e] = True
        left[v] -= 1
        left[other] -= 1
        seen[other] = True
    return out


# --------------------------------------------------------------------------- deterministic graph generator

_MASK = (1 << 64) - 1


def splitmix64_stream(seed, count):
    """count uint64 values of the splitmix64 sequence (vectorised; identical to the
    C++ generator in the package so graphs reproduce across torch versions)."""
    idx = np.arange(1, count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def synthetic_triples(num_nodes, num_rels, num_edges, seed=0):
    """s, o ~ U[0,N), p ~ U[0,R0): three consecutive stream values per triple
    (value mod range).  Duplicates are kept.  SURVEY.md section 8(d) S1."""
    z = splitmix64_stream(seed, 3 * num_edges).reshape(num_edges, 3)
    out = np.empty((num_edges, 3), np.int64)
    out[:, 0] = (z[:, 0] % np.uint64(num_nodes)).astype(np.int64)
    out[:, 1] = (z[:, 1] % np.uint64(num_rels)).astype(np.int64)
    out[:, 2] = (z[:, 2] % np.uint64(num_nodes)).astype(np.int64)
    return out
