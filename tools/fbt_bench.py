#!/usr/bin/env python3
"""AM-as-shipped first layer (featureless, basis 40, hidden 10) alone: per-kernel times of the tile kernels (rgcn_fbasis_tile.hip).
With the ablation library (RGCN_HIP_LIB=.../librgcn_hip_abl.so) RGCN_BWD_ABL = 1 no message loop, 2 no tile loads, 4 no row gathers,
8 no LDS adds (wrong results, timing only)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-rgcn_amd"))
from torch_rgcn import _native, routes  # noqa: E402
from torch_rgcn.layers import RelationalGraphConvolutionNC  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_666_764)
ap.add_argument("--r0", type=int, default=133)
ap.add_argument("--e", type=int, default=5_988_321)
ap.add_argument("--bases", type=int, default=40)
ap.add_argument("--d", type=int, default=10)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--hub", type=int, default=0, help="make node 7 the object of the first HUB triples (a hub source of the layer)")
a = ap.parse_args()
dev = torch.device("cuda")
T = torch.from_numpy(np.asarray(_native.synthetic_triples_host(a.n, a.r0, a.e, 1)))
if a.hub:
    T[: a.hub, 2] = 7
from torch_rgcn.utils import add_inverse_and_self  # noqa: E402
tp = add_inverse_and_self(T, a.n, a.r0)
layer = RelationalGraphConvolutionNC(triples=tp, num_nodes=a.n, num_relations=2 * a.r0 + 1, in_features=None, out_features=a.d,
                                     decomposition={"type": "basis", "num_bases": a.bases}).to(dev)
g = torch.randn(a.n, a.d, device=dev)
layer.zero_grad(set_to_none=True)
layer().backward(g)                 # builds the plan
plan = layer._graph.fbasis_plan()
bases, comps, bias = layer.bases.detach(), layer.comps.detach(), layer.bias.detach()


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / a.iters, 4)


mode = int(os.environ.get("FBT_MODE", "1"))
res = {"mode": mode, "max_degree": plan.max_src_degree,
       "fwd+gather": timed(lambda: _native.fbasis_tile_fwd(bases, comps, bias, plan, mode=mode)),
       "dbases": timed(lambda: _native.fbasis_tile_bwd(bases, comps, g, plan, True, False, mode=mode)),
       "dcomps": timed(lambda: _native.fbasis_tile_bwd(bases, comps, g, plan, False, True, mode=mode)),
       # both gradients: ONE walk in mode 1 when its LDS image fits (round 5: fbn_bwd_kernel), the two kernels in mode 3
       "bwd_both": timed(lambda: _native.fbasis_tile_bwd(bases, comps, g, plan, True, True, mode=mode)),
       "fused_strip_rows": int(_native.lib().rgcn_fbasis_tile_bwd_fused_gn(2 * a.r0 + 1, a.bases, a.d, a.n))}
L = _native.lib()
if hasattr(L, "rgcn_fbt_debug_read"):
    import ctypes
    buf = (ctypes.c_ulonglong * 8)()
    L.rgcn_fbt_debug_read(buf, 1)
    _native.fbasis_tile_fwd(bases, comps, bias, plan, mode=0)
    L.rgcn_fbt_debug_read(buf, 1)
    w = max(buf[5], 1)
    res["fwd_us_per_wave"] = {k: round(buf[i] / w / 100.0, 1) for i, k in enumerate(("arrive+store+flush", "issue", "messages", "barrier", "rotate"))}
    if mode == 1:
        _native.fbasis_tile_bwd(bases, comps, g, plan, True, False, mode=1)
        L.rgcn_fbt_debug_read(buf, 1)
        w = max(buf[6], 1)
        res["dbases_cycles_per_wave_and_tile"] = {k: round(buf[i] / w / (plan.n_nodes / 16 / 512), 0) for i, k in
                                                  enumerate(("issue", "message loops", "barrier 1", "write-out", "barrier 2", "rotate"))}
if hasattr(L, "rgcn_fbt_debug_read") and mode == 1 and res["fused_strip_rows"]:
    import ctypes
    buf = (ctypes.c_ulonglong * 8)()
    L.rgcn_fbt_debug_read(buf, 1)
    _native.fbasis_tile_bwd(bases, comps, g, plan, True, True, mode=1)
    L.rgcn_fbt_debug_read(buf, 1)
    w = max(buf[6], 1)          # (s_memtime ticks at 100 MHz)
    res["bwd_fused_us_per_wave_and_tile"] = {k: round(buf[i] / w / 100.0, 3) for i, k in
                                             enumerate(("issue", "messages", "wait for the loads", "barrier 1", "hand-over", "barrier 2"))}
_native.profile_start()
_native.fbasis_tile_fwd(bases, comps, bias, plan, mode=mode)
prof = _native.profile_stop()
res.update({k: round(float(np.mean(v)), 4) for k, v in prof.items()})
print("abl", routes.get("bwd_abl", "0"), "messages", plan.n_messages, res)
