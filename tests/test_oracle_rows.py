"""oracle.nc_layer_rows (sampled rows of a graph too large to walk whole: the full-size AM tests) against oracle.nc_layer on a graph
small enough for both: every sampled row must be the full oracle's row."""
import numpy as np
import pytest

from oracle import oracle


@pytest.mark.parametrize("mode", ["none", "block", "basis", "featureless-basis"])
def test_sampled_rows_are_the_full_oracles_rows(mode):
    rng = np.random.default_rng(5)
    N, R0, E = 300, 4, 3000
    R = 2 * R0 + 1
    tp = oracle.add_inverse_and_self(oracle.synthetic_triples(N, R0, E, 1), N, R0)
    g = rng.standard_normal((N, 6)).astype(np.float32)
    out_rows, src_rows, rel_rows = rng.choice(N, 40, replace=False), rng.choice(N, 30, replace=False), np.array([0, 3, 8])
    X = rng.standard_normal((N, 6)).astype(np.float32)
    if mode == "none":
        params = {"weights": rng.standard_normal((R, 6, 6)).astype(np.float32)}
    elif mode == "block":
        params = {"blocks": rng.standard_normal((R, 2, 3, 3)).astype(np.float32)}
    elif mode == "basis":
        params = {"comps": rng.standard_normal((R, 3)).astype(np.float32), "bases": rng.standard_normal((3, 6, 6)).astype(np.float32)}
    else:
        params, X = {"comps": rng.standard_normal((R, 3)).astype(np.float32), "bases": rng.standard_normal((3, N, 6)).astype(np.float32)}, None
    bias = rng.standard_normal(6).astype(np.float32)
    m = mode.replace("featureless-", "")
    err = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    for vertical in ((False, True) if X is not None else (False,)):
        full = oracle.nc_layer(tp, N, R, X, params, m, bias, vertical, g)
        part = oracle.nc_layer_rows(tp, N, R, X, params, m, bias, vertical, g, out_rows, src_rows, rel_rows)
        assert err(part["out"], full["out"][out_rows]) < 1e-6 and err(part["db"], full["db"]) < 1e-6
        if X is not None:
            assert err(part["dX"], full["dX"][src_rows]) < 1e-6
            for k, v in part["grads_rows"].items():
                assert err(v, full["grads"][k][rel_rows]) < 1e-6, k
            assert set(part["grads_rows"]) == ({"comps"} if m == "basis" else set(full["grads"]))
        else:
            assert err(part["dbases_rows"], full["grads"]["bases"][:, src_rows, :]) < 1e-6
            assert err(part["dcomps_rows"], full["grads"]["comps"][rel_rows]) < 1e-6
