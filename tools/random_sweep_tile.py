"""Randomised parity sweep of the featureless basis layer on the in-place tile kernels (rgcn_fbasis_tile.hip, forced for tables of any size):
node counts around the 16-node tile grid, 1..70 bases, widths 1..16, up to 81 relations, hubs, message counts 0..30 N, both kernel forms.
python tools/random_sweep_tile.py SEED [CASES]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "torch-rgcn_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import test_gpu_parity as T  # noqa: E402
from torch_rgcn import _native, routes  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rng = np.random.default_rng(seed)
routes.set("fbasis_inplace_mb", "0")
fails = tiled = 0
for case in range(cases):
    N = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 100, 257, 1000, 1023, 3001, 5000]))
    R0 = int(rng.choice([1, 2, 3, 6, 20, 40]))
    E = min(int(rng.choice([0, 1, 17, 400, 3000, 30000, 100000])), 30 * N)
    B = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 13, 16, 30, 40, 41, 64, 70]))
    d = int(rng.choice([2, 3, 4, 5, 7, 8, 9, 10, 11, 12, 15, 16]))      # (d = 1: the bias gradient is ONE sum of N cancelling terms, ill-conditioned for a relative bar)
    mode = str(rng.choice(["ranges", "nodes"]))
    hub = bool(rng.random() < 0.4) and N > 2
    tag = f"case {case}: N={N} R0={R0} E={E} B={B} d={d} mode={mode} hub={hub}"
    if os.environ.get("SWEEP_VERBOSE"):
        print(tag, flush=True)
    routes.set("fbasis_tile", mode)
    _native.profile_start()
    try:
        T.run_layer_vs_oracle(N=N, R0=R0, E=E, d_in=None, d_out=d, mode="basis", featureless=True, seed=7000 + case, hub=hub, num_bases=B)
    except Exception as exc:  # noqa: BLE001
        fails += 1
        print("FAIL", tag, f"{type(exc).__name__}: {str(exc)[:200]}", flush=True)
    tiled += "fbasis_tile_bwd" in _native.profile_stop()
print("done, cases:", cases, "on the tile kernels:", tiled, "failures:", fails)
