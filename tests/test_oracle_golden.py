"""Pins the CPU oracle (oracle/) to the reference: every golden vector under
tests/golden/ was produced by the reference itself (tests/golden/gen_golden.py);
here the oracle must reproduce them.  CPU only."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from oracle import oracle

TOL = 2e-5  # fp32 reference vs double-accumulating oracle, tiny graphs


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def params_of(d):
    return {k[len("param_"):]: v for k, v in d.items() if k.startswith("param_")}


def mode_of(name, params):
    if "diag" in name:
        return "diag"
    if "bases" in params:
        return "basis"
    if "blocks" in params:
        return "block"
    return "none"


# ---- reference known-answer vectors (tests/test_utils.py of the reference) restated ----

def test_kat_add_inverse_and_self():
    # reference tests/test_utils.py:5-25 -- negative ids prove it is pure index shuffling
    t = np.array([[0, 0, -1], [1, 1, -2], [2, 2, -3]])
    exp = np.array([[0, 0, -1], [1, 1, -2], [2, 2, -3], [-1, 3, 0], [-2, 4, 1], [-3, 5, 2],
                    [0, 6, 0], [1, 6, 1], [2, 6, 2]])
    assert np.array_equal(oracle.add_inverse_and_self(t, 3, 3), exp)


def test_kat_stack_matrices():
    # reference tests/test_utils.py:28-84
    t = np.array([[0, 0, 3], [1, 1, 4], [2, 2, 5], [3, 3, 0], [4, 4, 1], [5, 5, 2],
                  [0, 6, 0], [1, 6, 1], [2, 6, 2], [3, 6, 3], [4, 6, 4], [5, 6, 5]])
    vi, vs = oracle.stack_matrices(t, 9, 7, True)
    assert vs == (63, 9)
    assert np.array_equal(vi, np.array([[0, 3], [10, 4], [20, 5], [30, 0], [40, 1], [50, 2],
                                        [54, 0], [55, 1], [56, 2], [57, 3], [58, 4], [59, 5]]))
    hi, hs = oracle.stack_matrices(t, 9, 7, False)
    assert hs == (9, 63)
    assert np.array_equal(hi, np.array([[0, 3], [1, 13], [2, 23], [3, 27], [4, 37], [5, 47],
                                        [0, 54], [1, 55], [2, 56], [3, 57], [4, 58], [5, 59]]))


def test_kat_sum_sparse():
    # reference tests/test_utils.py:87-123
    ver = np.array([[0, 0], [0, 1], [0, 2], [4, 1], [8, 2], [7, 2]])
    v = 1.0 / oracle.sum_sparse(ver, np.ones(6), (9, 3), True)
    assert np.array_equal(v, np.array([1 / 3, 1 / 3, 1 / 3, 1, 1, 1], np.float32))
    hor = np.array([[0, 0], [1, 0], [2, 0], [3, 0], [1, 4], [2, 8], [2, 7]])
    v = 1.0 / oracle.sum_sparse(hor, np.ones(7), (4, 9), False)
    assert np.array_equal(v, np.array([.25, .25, .25, .25, 1, 1, 1], np.float32))


def test_kat_swap_trick():
    # reference tests/test_utils.py:170-220 (not collected upstream; float values make it pass)
    t = np.array([[0, 0, 1], [0, 0, 2], [1, 0, 2], [1, 1, 0], [2, 1, 0], [2, 1, 1]])
    N, R0 = 3, 2
    tp = oracle.add_inverse_and_self(t, N, R0)
    for vertical in (True, False):
        val = oracle.nc_edge_norm(tp, N, 2 * R0 + 1, vertical)
        # per-(relation, receiving node) mean: counts of edges sharing (p, s)
        cnt = {}
        for s, p, o in tp:
            cnt[(p, s)] = cnt.get((p, s), 0) + 1
        exp = np.array([1.0 / cnt[(p, s)] for s, p, o in tp], np.float32)
        assert np.array_equal(val, exp)


# ---- G3 ----

def test_g3_utils():
    d = load_golden("g3_utils")
    N, R0 = int(d["num_nodes"]), int(d["num_rels"])
    tp = oracle.add_inverse_and_self(d["triples"], N, R0)
    assert np.array_equal(tp, d["triples_plus"])
    R = 2 * R0 + 1
    vi, vs = oracle.stack_matrices(tp, N, R, True)
    hi, hs = oracle.stack_matrices(tp, N, R, False)
    assert np.array_equal(vi, d["ver_idx"]) and tuple(d["ver_size"]) == vs
    assert np.array_equal(hi, d["hor_idx"]) and tuple(d["hor_size"]) == hs
    assert np.array_equal(oracle.sum_sparse(vi, None, vs, True), d["ver_sums"])
    assert np.array_equal(oracle.sum_sparse(hi, None, hs, False), d["hor_sums"])


# ---- G1 / G2 ----

G1 = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "g1_nc_*.npz")))
G2 = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "g2_lp_*.npz")))


def check_layer(res, d, params):
    assert rel_err(res["out"], d["out"]) < TOL
    if "grad_X" in d:
        assert rel_err(res["dX"], d["grad_X"]) < TOL
    if "grad_bias" in d:
        assert rel_err(res["db"], d["grad_bias"]) < TOL
    for k, gv in res["grads"].items():
        assert rel_err(gv, d["grad_" + k]) < TOL, k
    assert set(res["grads"]) == {k[5:] for k in d if k.startswith("grad_")} - {"X", "bias"}


@pytest.mark.parametrize("name", G1)
def test_g1_nc_layer(name):
    d = load_golden(name)
    N, R0 = int(d["num_nodes"]), int(d["num_rels"])
    params = params_of(d)
    bias = params.pop("bias", None)
    res = oracle.nc_layer(d["triples_plus"], N, 2 * R0 + 1, d.get("X"), params, mode_of(name, params), bias,
                          bool(d["vertical"]), d["g"])
    check_layer(res, d, params)


@pytest.mark.parametrize("name", G2)
def test_g2_lp_layer(name):
    d = load_golden(name)
    N, R0 = int(d["num_nodes"]), int(d["num_rels"])
    params = params_of(d)
    bias = params.pop("bias", None)
    res = oracle.lp_layer(d["triples"], N, 2 * R0 + 1, d["X"], params, mode_of(name, params), bias,
                          bool(d["vertical"]), None, d["g"])
    check_layer(res, d, params)


def test_lp_horizontal_quirk_formula():
    """with horizontal stacking the LP layer's adjacency is NOT the per-(p,s) mean.
    Net weight of a forward edge (s,p,o) = 1/out(s,p) + 1/(2 in(p,o)); inverse edge 1/(2 in(p,o));
    self loop 1 -- because the original triples enter twice.  The oracle must reproduce that."""
    d = load_golden("g2_lp_eval_sd_h_none")
    N, R0 = int(d["num_nodes"]), int(d["num_rels"])
    T = d["triples"]
    E = len(T)
    tp, n_self = oracle.lp_augment(T, N, R0)
    assert n_self == E + N and len(tp) == 3 * E + N
    val = oracle.edge_norm(tp, N, 2 * R0 + 1, False, E, n_self)
    out_c, in_c = {}, {}
    for s, p, o in T:
        out_c[(s, p)] = out_c.get((s, p), 0) + 1
        in_c[(p, o)] = in_c.get((p, o), 0) + 1
    for e, (s, p, o) in enumerate(T):
        assert val[e] == np.float32(1.0) / np.float32(out_c[(s, p)])
        assert val[E + e] == np.float32(1.0) / np.float32(2 * in_c[(p, o)])
        assert val[2 * E + e] == np.float32(1.0) / np.float32(2 * in_c[(p, o)])
    assert np.all(val[3 * E:] == 1.0)


# ---- G6 mid-size two-layer ----

def test_g6_mid():
    d = load_golden("g6_mid")
    N, R0 = int(d["num_nodes"]), int(d["num_rels"])
    R = 2 * R0 + 1
    tp = oracle.add_inverse_and_self(d["triples"].astype(np.int64), N, R0)
    v1 = oracle.nc_edge_norm(tp, N, R, False)
    v2 = oracle.nc_edge_norm(tp, N, R, True)
    h = oracle.rgcn_forward(tp, v1, N, R, d["X"], d["w1"], d["b1"])
    assert rel_err(h, d["h"]) < TOL
    a = np.maximum(h, 0)
    out = oracle.rgcn_forward(tp, v2, N, R, a, d["w2"], d["b2"])
    assert rel_err(out, d["out"]) < TOL
    g = (2.0 / out.size) * out
    da, dw2, db2 = oracle.rgcn_backward(tp, v2, N, R, a, d["w2"], g)
    dh = da * (h > 0)
    dx, dw1, db1 = oracle.rgcn_backward(tp, v1, N, R, d["X"], d["w1"], dh)
    for got, key in ((dw2, "grad_w2"), (db2, "grad_b2"), (dw1, "grad_w1"), (db1, "grad_b1"), (dx, "grad_X")):
        assert rel_err(got, d[key]) < 5e-5, key


# ---- G11 thousands of parallel edges: where the REFERENCE's own fp32 round-off passes 1e-4 ----

G11 = ("g11_dup_wide", "g11_dup_basis", "g11_dup_featureless")


def g11_case(name):
    """-> (reference tensors, oracle tensors, inputs) of one G11 fixture, keyed out / dX / <parameter>"""
    d = load_golden(name)
    N, R0 = int(d["num_nodes"]), int(d["num_rels"])
    R = 2 * R0 + 1
    tp = oracle.add_inverse_and_self(d["triples"].astype(np.int64), N, R0)
    params = params_of(d)
    bias = params.pop("bias", None)
    X = d["X"] if "X" in d else None
    res = oracle.nc_layer(tp, N, R, X, params, mode_of(name, params), bias, False, d["g"])
    ref = {"out": d["out"], **{n: d[f"grad_{n}"] for n in params}}
    got = {"out": res["out"], **res["grads"]}
    if X is not None:
        ref["dX"], got["dX"] = d["grad_X"], res["dX"]
    if bias is not None:
        ref["bias"], got["bias"] = d["grad_bias"], res["db"]
    return ref, got, (d, tp, N, R)


def test_g11_parallel_edges_reference_round_off():
    """9,000 .. 20,000 copies of the same few triples: every output element of the reference is one chain of that many fp32 additions
    of equal terms (torch.sparse.mm), and its distance from the oracle's doubles is ABOVE the 1e-4 the parity tests otherwise use --
    the fixture pins that observation, and bounds it: parity on such an input is only defined to the reference's own round-off."""
    worst = {}
    for name in G11:
        ref, got, _ = g11_case(name)
        assert set(ref) == set(got)
        worst[name] = max(rel_err(ref[k], got[k]) for k in ref)
        assert worst[name] < 5e-3, (name, worst[name])
    assert max(worst.values()) > 1e-4, worst


# ---- G5 DistMult ----

def test_g5_distmult():
    d = load_golden("g5_distmult")
    for tag in ("nb", "b"):
        rel = d[f"{tag}_param_relations"]
        biases = (d[f"{tag}_param_sbias"], d[f"{tag}_param_pbias"], d[f"{tag}_param_obias"]) if tag == "b" else (None,) * 3
        for nm, tr in (("2", d["triples2"]), ("3", d["triples3"])):
            sc = oracle.distmult_forward(tr, d["nodes"], rel, *biases)
            assert rel_err(sc, d[f"{tag}_scores{nm}"]) < TOL
            dn, dr, dsb, dpb, dob = oracle.distmult_backward(tr, d["nodes"], rel, d[f"{tag}_g{nm}"], tag == "b")
            assert rel_err(dn, d[f"{tag}_grad_nodes{nm}"]) < TOL
            assert rel_err(dr, d[f"{tag}_grad_relations{nm}"]) < TOL
            if tag == "b":
                assert rel_err(dsb, d[f"{tag}_grad_sbias{nm}"]) < TOL
                assert rel_err(dpb, d[f"{tag}_grad_pbias{nm}"]) < TOL
                assert rel_err(dob, d[f"{tag}_grad_obias{nm}"]) < TOL


@pytest.mark.parametrize("name", ["g7_eval_lp", "g7_eval_ties"])
def test_g7_ranking_evaluator(name):
    """oracle.evaluate (utils/misc.py:60-110 restated) on the encoder output the reference produced"""
    d = load_golden(name)
    N = int(d["num_nodes"])
    bias = [d[k] for k in ("sbias", "pbias", "obias")] if "sbias" in d else [None] * 3
    if name == "g7_eval_lp":
        known = np.concatenate([d["train"], d["valid"], d["test"]])
    else:
        known = np.concatenate([d["known"], d["known"][:10], d["test"]])
    true_triples = oracle.generate_true_dict(known)
    score = lambda toscore: oracle.distmult_forward(toscore, d["nodes"], d["relations"], *bias)  # noqa: E731
    for tag, filt in (("filtered", True), ("raw", False)):
        mrr, hits, ranks = oracle.evaluate(score, d["test"], true_triples, N, batch_size=int(d["batch_size"]),
                                           filter_candidates=filt)
        assert ranks == d[f"ranks_{tag}"].tolist(), tag
        assert abs(mrr - float(d[f"mrr_{tag}"])) < 1e-12 and np.allclose(hits, d[f"hits_{tag}"], atol=1e-12)
    if "head_scores" in d:   # integer data: the score matrix itself is exact
        toscore = np.repeat(d["test"][:, None, :], N, axis=1)
        toscore[:, :, 0] = np.arange(N)[None, :]
        assert np.array_equal(score(toscore), d["head_scores"])


def test_g8_edge_neighborhood_sampler_reproduces_reference_picks():
    d = load_golden("g8_sampler")
    T, N = d["triples"], int(d["num_nodes"])
    for seed in (123, 7):
        want = d[f"picked_seed{seed}"]
        idx = oracle.edge_neighborhood(T, len(want), N, np.random.RandomState(seed))
        assert len(set(idx.tolist())) == len(want) and np.array_equal(T[idx], want)
