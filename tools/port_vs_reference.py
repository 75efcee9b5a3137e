#!/usr/bin/env python3
"""Build-container check behind bench.py's `cpu_baseline` (kind "port"): the PyTorch-CPU port (oracle/torch_cpu_port.py)
against the IMPORTED reference (/root/reference, thiviyanT/torch-rgcn) on the same inputs -- same loss and gradients, and
the wall-time ratio.  The reference cannot travel to the GPU box, so this runs here and its record is committed:

    python tools/port_vs_reference.py > profiles/r02_port_vs_reference.json

S1 at 1/10 scale (N = 100,000, E = 1,000,000, R0 = 50, d = 16), layer 1 horizontal -> ReLU -> layer 2 vertical,
loss = mean(out^2), forward + backward; 1 warm-up + `--steps` timed steps each, interleaved (port, reference, port, ...)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("RGCN_REFERENCE", "/root/reference")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import oracle, torch_cpu_port  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--nodes", type=int, default=100_000)
ap.add_argument("--edges", type=int, default=1_000_000)
a = ap.parse_args()
N, R0, E, d = a.nodes, 50, a.edges, 16
R = 2 * R0 + 1
threads = int(os.environ.get("RGCN_CPU_THREADS", os.cpu_count() or 1))
torch.set_num_threads(threads)
tp = torch.from_numpy(oracle.add_inverse_and_self(oracle.synthetic_triples(N, R0, E, 0), N, R0))
g = torch.Generator().manual_seed(0)
base = [torch.randn(N, d, generator=g), torch.randn(R, d, d, generator=g) * 0.2, torch.zeros(d),
        torch.randn(R, d, d, generator=g) * 0.2, torch.zeros(d)]

sys.path.insert(0, REF)
import warnings  # noqa: E402
warnings.filterwarnings("ignore")
from torch_rgcn.layers import RelationalGraphConvolutionNC as RefLayer  # noqa: E402  (the reference's own class)


def ref_step(ts):
    X, w1, b1, w2, b2 = ts
    l1 = RefLayer(triples=tp, num_nodes=N, num_relations=R, in_features=d, out_features=d, vertical_stacking=False)
    l2 = RefLayer(triples=tp, num_nodes=N, num_relations=R, in_features=d, out_features=d, vertical_stacking=True)
    l1.weights, l1.bias, l2.weights, l2.bias = (torch.nn.Parameter(t) for t in (w1, b1, w2, b2))
    t0 = time.perf_counter()
    loss = l2(torch.relu(l1(X))).pow(2).mean()
    loss.backward()
    dt = time.perf_counter() - t0
    return dt, loss.detach(), X.grad, l1.weights.grad, l2.weights.grad


def port_step(ts):
    t0 = time.perf_counter()
    loss = torch_cpu_port.two_layer_step(tp, N, R, *ts)
    dt = time.perf_counter() - t0
    return dt, loss, ts[0].grad, ts[1].grad, ts[3].grad


times = {"port": [], "reference": []}
last = {}
for it in range(a.steps + 1):
    for name, fn in (("port", port_step), ("reference", ref_step)):
        ts = [t.clone().requires_grad_(True) for t in base]
        dt, *res = fn(ts)
        if it:
            times[name].append(dt)
        last[name] = res
same = {k: bool(torch.equal(x, y)) for k, x, y in zip(("loss", "dX", "dW1", "dW2"), last["port"], last["reference"])}
rel = {k: float((x - y).abs().max() / y.abs().max()) for k, x, y in zip(("loss", "dX", "dW1", "dW2"), last["port"], last["reference"])}
med = {k: float(np.median(v)) for k, v in times.items()}
print(json.dumps({"what": "oracle/torch_cpu_port.py vs the imported reference (RelationalGraphConvolutionNC, /root/reference), CPU",
                  "workload": f"N={N} E={E} R0={R0} d={d}, 2 layers fwd+bwd", "threads": threads, "torch": torch.__version__,
                  "timed_steps_each": a.steps, "seconds_per_step": {k: [round(x, 3) for x in v] for k, v in times.items()},
                  "median_s": {k: round(v, 3) for k, v in med.items()}, "port_over_reference_time": round(med["port"] / med["reference"], 3),
                  "identical_bits": same, "rel_max_diff": rel}, indent=1))
