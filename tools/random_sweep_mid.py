"""Randomised parity sweep at the sizes where the block-tile and sparse-bucket kernels take over (33 k .. 300 k nodes, 7 .. 401 relations, hidden 16
and the widths around it): the routes the small sweep (tools/random_sweep.py, <= 2049 nodes) never reaches.  python tools/random_sweep_mid.py SEED [CASES]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "torch-rgcn_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import test_gpu_parity as T  # noqa: E402
from torch_rgcn import _native  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
fails = 0
seen = set()
for case in range(cases):
    N = int(rng.choice([32768, 32769, 40001, 65536, 100003, 300000]))
    R0 = int(rng.choice([3, 20, 55, 60, 133, 200]))
    E = int(rng.choice([1, 4, 10])) * N
    mode = str(rng.choice(["none", "none", "basis", "block"]))
    d_in, d_out = [(16, 16), (16, 16), (16, 16), (10, 11), (16, 4), (12, 16), (32, 32), (16, 48)][int(rng.integers(0, 8))]
    nb = 4 if mode == "block" else 2
    if mode == "block":
        d_in, d_out = 4 * max(1, d_in // 4), 4 * max(1, d_out // 4)
    vertical = bool(rng.random() < 0.5)
    hub = bool(rng.random() < 0.4)
    tag = f"case {case}: N={N} R0={R0} E={E} mode={mode} d=({d_in},{d_out}) vertical={vertical} hub={hub}"
    if os.environ.get("SWEEP_VERBOSE"):
        print(tag, flush=True)
    _native.profile_start()
    try:
        T.run_layer_vs_oracle(N=N, R0=R0, E=E, d_in=d_in, d_out=d_out, mode=mode, vertical=vertical, seed=6000 + case, hub=hub,
                              num_bases=int(rng.integers(1, 12)), num_blocks=nb)
    except Exception as exc:  # noqa: BLE001
        fails += 1
        print("FAIL", tag, f"{type(exc).__name__}: {str(exc)[:200]}", flush=True)
    seen.update(k for k in _native.profile_stop())
    T._ORACLE_MEMO.clear()
print("done, cases:", cases, "failures:", fails, "kernels timed:", sorted(seen))
