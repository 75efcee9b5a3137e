"""Thread sweep of the CPU baseline: the reference's op sequence (oracle/torch_cpu_port.two_layer_step) at 1/10 of S1
(N = 100 k, E = 1 M, R0 = 50, d = 16) for several torch thread counts on the GPU box's host -> one JSON line.  Why bench.py caps the
baseline at 32 threads: ATen's sparse kernels stop scaling long before a 256-thread host is used up, and then regress.
  python tools/cpu_threads_probe.py > profiles/r04_cpu_threads.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import oracle, torch_cpu_port  # noqa: E402

N, R0, E, d = 100_000, 50, 1_000_000, 16
tp = torch.from_numpy(oracle.add_inverse_and_self(oracle.synthetic_triples(N, R0, E, 0), N, R0))
R = 2 * R0 + 1
g = torch.Generator().manual_seed(0)
base = [torch.randn(N, d, generator=g), torch.randn(R, d, d, generator=g) * 0.2, torch.zeros(d), torch.randn(R, d, d, generator=g) * 0.2, torch.zeros(d)]
out = {"workload": f"S1 at 1/10 scale (N={N}, E={E}, R0={R0}, d={d}), oracle/torch_cpu_port.two_layer_step, 1 warm-up + 2 timed steps, best",
       "cores_available": os.cpu_count(), "torch": torch.__version__, "s_per_step_by_threads": {}}
for th in (1, 4, 8, 16, 32, 64, 128, os.cpu_count() or 1):
    if th > (os.cpu_count() or 1) or str(th) in out["s_per_step_by_threads"]:
        continue
    torch.set_num_threads(th)
    ts = []
    for _ in range(3):
        args = [t.clone().requires_grad_(True) for t in base]
        t0 = time.perf_counter()
        torch_cpu_port.two_layer_step(tp, N, R, *args)
        ts.append(time.perf_counter() - t0)
    out["s_per_step_by_threads"][str(th)] = round(min(ts[1:]), 3)
best = min(out["s_per_step_by_threads"], key=lambda k: out["s_per_step_by_threads"][k])
out["best_threads"] = int(best)
print(json.dumps(out))
