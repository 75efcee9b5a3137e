"""The PyTorch-CPU port timed as cpu_baseline must give the reference's numbers
(golden vectors) -- otherwise its timings could not be quoted as 'the reference CPU path'."""
import numpy as np
import torch

from conftest import load_golden
from oracle import oracle, torch_cpu_port


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_port_matches_g6():
    d = load_golden("g6_mid")
    N, R0 = int(d["num_nodes"]), int(d["num_rels"])
    tp = torch.from_numpy(oracle.add_inverse_and_self(d["triples"].astype(np.int64), N, R0))
    X = torch.from_numpy(d["X"]).requires_grad_(True)
    ps = [torch.from_numpy(d[k]).requires_grad_(True) for k in ("w1", "b1", "w2", "b2")]
    loss = torch_cpu_port.two_layer_step(tp, N, 2 * R0 + 1, X, *ps)
    assert abs(loss.item() - float(d["loss"])) < 1e-5 * abs(float(d["loss"]))
    assert rel_err(X.grad.numpy(), d["grad_X"]) < 1e-5
    for t, k in zip(ps, ("grad_w1", "grad_b1", "grad_w2", "grad_b2")):
        assert rel_err(t.grad.numpy(), d[k]) < 1e-5, k


def test_port_featureless_and_vertical_g1():
    for name, vertical in (("g1_nc_h_none_fl", False), ("g1_nc_v_none_ft", True), ("g1_nc_h_none_ft", False)):
        d = load_golden(name)
        N, R0 = int(d["num_nodes"]), int(d["num_rels"])
        X = torch.from_numpy(d["X"]) if "X" in d else None
        out = torch_cpu_port.layer_cpu(torch.from_numpy(d["triples_plus"]), N, 2 * R0 + 1, X,
                                       torch.from_numpy(d["param_weights"]), torch.from_numpy(d["param_bias"]),
                                       vertical=vertical)
        assert rel_err(out.numpy(), d["out"]) < 1e-5, name
