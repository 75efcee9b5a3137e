#!/usr/bin/env python3
"""bench.py -- 2-layer R-GCN forward+backward throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload S1 (SURVEY.md 8d / BASELINE.md): N = 1,000,000 nodes, R0 = 50 relations (layer R = 101),
E = 10,000,000 base triples from a splitmix64 stream (seed 0), augmented with inverses and self loops
(M = 21,000,000 messages per layer); layer 1 16->16 (horizontal normalisation), ReLU, layer 2 16->16
(vertical), learnable input features, loss = mean(out^2); fp32.  One step = forward + backward of
both layers (optimiser excluded, graph preprocessing excluded: the NC graph is static).

N > 1: relation-sharded, weak scaling -- every rank owns its own 50 relations / 10 M triples over the
SAME 1 M nodes (rank r uses seed r), and the partial N x 16 outputs / feature gradients are summed with
an RCCL all-reduce between layers.  value = edges of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "torch-rgcn_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

# This image exports NCCL_DEBUG=VERSION, which makes RCCL print a banner on STDOUT when the communicator is torn
# down, i.e. after our result line.  The contract is "rank 0 prints ONE JSON line": keep it the last line.
if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
    del os.environ["NCCL_DEBUG"]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL across processes)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

METRIC = "edges/s/GPU (fwd+bwd) 2-layer RGCN, 1M nodes/10M edges/50 rels, h=16"
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes(M, N, d_in, d_out):
    """forward of one layer = M*(4*d_in + 8) + N*4*d_out."""
    return M * (4 * d_in + 8) + N * 4 * d_out


PMC_FILE = os.path.join(ROOT, "profiles", "r01_pmc_kernels.json")
PMC_NOTE = ("profiles/r01_pmc_kernels.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/kbench.py "
            "on the same S1 launch; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE tallies 128-B requests "
            "at 64 B, MI355X_MICROARCH.md HBM section). Calibration on a known random-64B-row pattern "
            "(profiles/r01_pmc_gather_probe_calibration.json) reads 1.0x: if the row gathers are 64-B requests the "
            "HBM-side bytes are (FETCH_SIZE + WRITE_SIZE)*1024 (see traffic_if_64B_requests).")


def pmc_traffic(kernel_substr, doubled=True):
    """HBM-side bytes per launch of a kernel from the committed PMC summary (None when absent)."""
    try:
        with open(PMC_FILE) as f:
            data = json.load(f)
    except OSError:
        return None
    for name, c in data.items():
        if kernel_substr in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            return int(((2.0 if doubled else 1.0) * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024)
    return None


def pmc_detail(kernel_key):
    """fabric request counters / MFMA busy of a kernel from profiles/r01_pmc_detail.json (None when absent)"""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_detail.json")) as f:
            c = json.load(f)[kernel_key]
        return {"fabric_read_requests": int(c["TCC_EA0_RDREQ_sum"]),
                "share_of_128B_requests": round(c["TCC_EA0_RDREQ_128B_sum"] / c["TCC_EA0_RDREQ_sum"], 4),
                "l2_hit_rate": round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 3),
                "mfma_busy_frac": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 1024), 4),
                "note": "profiles/r01_pmc_detail.json (tools/pmc_passes.sh): every 64-B row gather is a 128-B fabric request "
                        "(profiles/r01_gather_probe2.txt), so traffic ~ 1.75x the algorithmic bytes is the floor of this "
                        "access pattern, not re-reads"}
    except (OSError, KeyError, ZeroDivisionError):
        return None


def build_layers(N, R0, E, d, seed, device, group):
    from torch_rgcn import _native
    from torch_rgcn.dist import shard_layer
    from torch_rgcn.layers import RelationalGraphConvolutionNC
    T = _native.synthetic_triples_host(N, R0, E, seed)
    tp = torch.from_numpy(_native.add_inverse_and_self_host(T, N, R0))
    R = 2 * R0 + 1
    torch.manual_seed(1234 + seed)
    kw = dict(triples=tp, num_nodes=N, num_relations=R, in_features=d, out_features=d)
    l1 = RelationalGraphConvolutionNC(vertical_stacking=False, **kw).to(device)
    l2 = RelationalGraphConvolutionNC(vertical_stacking=True, **kw).to(device)
    if group is not None:
        shard_layer(l1, group)
        shard_layer(l2, group)
    return l1, l2, tp.size(0)


def cpu_baseline(steps=2):
    """The reference's op sequence (oracle/torch_cpu_port.py) on the host cores, 1/10-scale S1."""
    from oracle import oracle, torch_cpu_port
    N, R0, E, d = 100_000, 50, 1_000_000, 16
    # ATen's sparse kernels stop scaling (and then regress) long before 256 host threads: 13.1 s/step at
    # 256 threads vs the figures below; the baseline gets the best thread count we measured, not the worst.
    threads = int(os.environ.get("RGCN_CPU_THREADS", min(os.cpu_count() or 1, 32)))
    torch.set_num_threads(threads)
    tp = torch.from_numpy(oracle.add_inverse_and_self(oracle.synthetic_triples(N, R0, E, 0), N, R0))
    R = 2 * R0 + 1
    g = torch.Generator().manual_seed(0)
    base = [torch.randn(N, d, generator=g), torch.randn(R, d, d, generator=g) * 0.2, torch.zeros(d),
            torch.randn(R, d, d, generator=g) * 0.2, torch.zeros(d)]
    times = []
    for it in range(steps + 1):
        ts = [t.clone().requires_grad_(True) for t in base]
        t0 = time.perf_counter()
        torch_cpu_port.two_layer_step(tp, N, R, *ts)
        times.append(time.perf_counter() - t0)
    best = min(times[1:])
    return {"value": E / best, "unit": "edges/s", "cores": threads, "kind": "port",
            "sample": f"S1 at 1/10 scale (N={N}, E={E}, R0={R0}, d={d}), 1 warm-up + {steps} timed steps, best; "
                      f"{best:.2f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--rels", type=int, default=50)
    ap.add_argument("--edges", type=int, default=10_000_000)
    ap.add_argument("--hidden", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    if os.environ.get("RGCN_BENCH_ONE_DEVICE"):   # test hook: every rank on cuda:0 (with RGCN_DIST_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    backend = os.environ.get("RGCN_DIST_BACKEND", "nccl")      # "nccl" is RCCL on ROCm
    group = None
    if world > 1 or os.environ.get("RGCN_FORCE_DIST"):   # RGCN_FORCE_DIST=1: exercise the RCCL path on one GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        extra = {"device_id": device} if backend == "nccl" else {}
        if "RANK" not in os.environ:
            dist.init_process_group(backend, rank=0, world_size=1, **extra)
        else:
            dist.init_process_group(backend, **extra)
        group = dist.group.WORLD
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from torch_rgcn import _native
    N, R0, E, d = args.nodes, args.rels, args.edges, args.hidden
    l1, l2, M = build_layers(N, R0, E, d, seed=rank, device=device, group=group)
    torch.manual_seed(99)
    X = torch.randn(N, d, device=device, requires_grad=True)

    def step():
        for p in (X, l1.weights, l1.bias, l2.weights, l2.bias):
            p.grad = None
        out = l2(torch.relu(l1(X)))
        loss = out.pow(2).mean()
        loss.backward()
        return loss

    def fence():
        torch.cuda.synchronize()
        if group is not None:
            dist.barrier()
            torch.cuda.synchronize()

    slabs = None
    if group is not None and "RGCN_DIST_SLABS" not in os.environ:
        # Untimed set-up: how many slabs should a sharded spmm be cut into so that the all-reduce of slab k hides behind
        # the kernels of slab k+1?  That depends on the link (xGMI ring vs one rank vs gloo), so it is measured here, once:
        # 3 steps per candidate, max over ranks, the same choice on every rank.
        best = None
        for cand in ("0", "2", "4"):
            os.environ["RGCN_DIST_SLABS"] = cand
            step()
            fence()
            t_c = time.perf_counter()
            for _ in range(3):
                step()
            fence()
            tc = torch.tensor([time.perf_counter() - t_c], device=device, dtype=torch.float64)
            dist.all_reduce(tc, op=dist.ReduceOp.MAX)
            if best is None or tc.item() < best[0]:
                best = (tc.item(), cand)
        slabs = best[1]
        os.environ["RGCN_DIST_SLABS"] = slabs
    for _ in range(args.warmup):
        step()
    fence()
    _native.profile_start()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        marks[k].record()                 # HIP events on the launch stream: per-step GPU time (median / min below)
        step()
    marks[args.steps].record()
    fence()
    elapsed = time.perf_counter() - t0
    prof = _native.profile_stop()
    per_step = [marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps)]

    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if group is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()
    ms = 1e3 * elapsed / args.steps

    if rank == 0:
        kern = prof.get("spmm", [])
        spmm_ms = float(np.mean(kern)) if kern else None
        slabbed = False
        if spmm_ms is None and prof.get("spmm_slab"):
            # relation-sharded path: every spmm is launched in slabs of whole tiles (the all-reduce of slab k overlaps the
            # kernels of slab k+1); one launch = the slabs of one spmm, 4 spmm per step (2 forward, 2 feature-gradient)
            kern = prof["spmm_slab"]
            spmm_ms = float(np.sum(kern)) / (4 * args.steps)
            slabbed = True
        alg = algorithmic_bytes(M, N, d, d)
        roof = None
        traffic = pmc_traffic("spmm_d16_kernel")
        if spmm_ms:
            ach = alg / (spmm_ms * 1e-3) / 1e9
            roof = {"kernel": "spmm_d16_kernel (forward and feature-gradient launches)" +
                              (f", each launched in {len(kern) // (4 * args.steps)} slabs on rank 0" if slabbed else ""),
                    "bound": "hbm",
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "traffic_if_64B_requests": pmc_traffic("spmm_d16_kernel", doubled=False),
                    "traffic_rate_GBs": round(traffic / (spmm_ms * 1e-3) / 1e9, 1) if traffic else None,
                    "traffic_source": PMC_NOTE if traffic else None, "pmc": pmc_detail("spmm"), "avg_launch_ms": round(spmm_ms, 4), "launches_per_step": 4.0 if slabbed else len(kern) / args.steps,
                    "algorithmic_bytes_per_launch": alg,
                    "other_kernels_ms": {k: round(float(np.mean(v)), 4) for k, v in prof.items() if k != "spmm"}}
        res = {"metric": METRIC, "value": world * E / (ms * 1e-3), "unit": "edges/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "per_gpu_edges_per_s": E / (ms * 1e-3),
               "step_ms_median": round(float(np.median(per_step)), 4), "step_ms_min": round(float(np.min(per_step)), 4),
               "config": {"workload": f"S1: N={N} nodes, E={E} base triples/GPU, R0={R0} relations/GPU "
                                      f"(layer R={2 * R0 + 1}), M={M} messages/layer, hidden={d}, 2 NC layers "
                                      "(horizontal, vertical), fwd+bwd, learnable X",
                          "sharding": "single GPU" if world == 1 else f"relation-sharded x{world}, all-reduce N x {d} fp32"
                                      + (f", {slabs} slabs per spmm (picked in the untimed set-up)" if slabs else "")},
               "step_hbm_algorithmic_GBs": round(4 * (alg + 0) / (ms * 1e-3) / 1e9, 1),
               "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
    else:
        res = None
    if group is not None:
        dist.destroy_process_group()   # RCCL prints its banner here; keep the JSON line last
    if res is not None:
        sys.stdout.flush()
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
