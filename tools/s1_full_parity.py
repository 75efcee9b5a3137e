"""Full-size S1 parity, once: one NC layer (N = 1 M, 10 M triples, R = 101, M = 21 M messages, 16 -> 16) forward + backward on
the GPU against the C oracle on the host (~1 min of oracle time); the test suite checks this size through size-independent
properties only.   python tools/s1_full_parity.py > profiles/r01_s1_full_parity.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "torch-rgcn_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import oracle  # noqa: E402
from torch_rgcn.layers import RelationalGraphConvolutionNC  # noqa: E402

N, R0, E, d = 1_000_000, 50, 10_000_000, 16
R = 2 * R0 + 1
T = oracle.synthetic_triples(N, R0, E, seed=0)
tp = oracle.add_inverse_and_self(T, N, R0)
rng = np.random.default_rng(0)
for vertical in (False, True):
    layer = RelationalGraphConvolutionNC(triples=torch.from_numpy(tp), num_nodes=N, num_relations=R, in_features=d,
                                         out_features=d, vertical_stacking=vertical).cuda()
    with torch.no_grad():
        layer.bias.normal_()
    X = torch.from_numpy(rng.standard_normal((N, d)).astype(np.float32)).cuda().requires_grad_(True)
    out = layer(X)
    g = rng.standard_normal((N, d)).astype(np.float32)
    out.backward(torch.from_numpy(g).cuda())
    torch.cuda.synchronize()
    t0 = time.time()
    ref = oracle.nc_layer(tp, N, R, X.detach().cpu().numpy(), {"weights": layer.weights.detach().cpu().numpy()}, "none",
                          layer.bias.detach().cpu().numpy(), vertical, g)
    rel = lambda a, b: float(np.abs(a.detach().cpu().numpy().astype(np.float64) - b).max() / np.abs(b).max())  # noqa: E731
    print(f"S1 full size, {'vertical' if vertical else 'horizontal'} normalisation (oracle {time.time() - t0:.0f} s): rel-max errors "
          f"out {rel(out, ref['out']):.2e}  dX {rel(X.grad, ref['dX']):.2e}  dW {rel(layer.weights.grad, ref['grads']['weights']):.2e}  "
          f"db {rel(layer.bias.grad, ref['db']):.2e}   (bound 1e-4)", flush=True)
