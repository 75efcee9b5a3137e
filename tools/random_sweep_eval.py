"""Randomised sweep of the ranking evaluator's kernels (score-all on the matrix cores, filter, rank counting) through the suite's own checker,
tests/test_gpu_eval.py::test_score_all_vs_oracle, at random (candidates, queries, width, biases): python tools/random_sweep_eval.py SEED [CASES]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "torch-rgcn_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import test_gpu_eval as T  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rng = np.random.default_rng(seed)
fails = 0
for case in range(cases):
    N = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 127, 129, 500, 1023, 1025, 3000]))
    Q = int(rng.choice([1, 2, 7, 15, 16, 17, 31, 33, 64, 100, 257]))
    dim = int(rng.choice([1, 2, 3, 4, 5, 8, 15, 16, 17, 32, 50, 64, 100, 128, 200, 256, 300, 512, 520]))
    while N * Q * dim > 60_000_000:
        Q = max(1, Q // 2)
    biased = bool(rng.random() < 0.5)
    tag = f"case {case}: N={N} Q={Q} dim={dim} biased={biased}"
    if os.environ.get("SWEEP_VERBOSE"):
        print(tag, flush=True)
    try:
        T.test_score_all_vs_oracle(N, Q, dim, biased)
    except Exception as exc:  # noqa: BLE001
        fails += 1
        print("FAIL", tag, f"{type(exc).__name__}: {str(exc)[:200]}", flush=True)
print("done, cases:", cases, "failures:", fails)
