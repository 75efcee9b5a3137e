#!/usr/bin/env python3
"""bench.py -- 2-layer R-GCN forward+backward throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload S1 (SURVEY.md 8d / BASELINE.md): N = 1,000,000 nodes, R0 = 50 relations (layer R = 101),
E = 10,000,000 base triples from a splitmix64 stream (seed 0), augmented with inverses and self loops
(M = 21,000,000 messages per layer); layer 1 16->16 (horizontal normalisation), ReLU, layer 2 16->16
(vertical), learnable input features, loss = mean(out^2); fp32.  One step = forward + backward of
both layers (optimiser excluded, graph preprocessing excluded: the NC graph is static; its one-off
cost is reported as graph_build_ms).

N > 1 (BASELINE.json configs[4]): STRONG scaling of that ONE graph -- every rank generates the same
seed-0 S1 graph, normalises it on the full graph, keeps the relations the LPT packer assigns to it
(torch_rgcn.dist.shard_layer(keep="lpt")) and the partial N x 16 outputs / feature gradients are summed
over RCCL between layers.  value = E / max-over-ranks step time (whole job).
`--weak` keeps round 1's weak-scaling mode (every rank its own 50 relations / 10 M triples over the
same 1 M nodes, seed = rank; value = N_gpus * E / time).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# the secondary config lines replay captured hipGraphs: this runtime's graph packet capture replays memset nodes with stale arguments
# (torch_rgcn/__init__.py); it has to be off before the process' first HIP call
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

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
PROFILE_ROUND = "r06"


def fwd_bytes(M, N, d_in, d_out):
    """forward of one layer = M*(4*d_in + 8) + N*4*d_out."""
    return M * (4 * d_in + 8) + N * 4 * d_out


def bwd_bytes(M, N, d_in, d_out, x_needs_grad=True):
    """backward of one layer = M*(4*d_out + 8) + N*4*d_in*(1 + [X needs grad])."""
    return M * (4 * d_out + 8) + N * 4 * d_in * (2 if x_needs_grad else 1)


def _profile_json(name):
    for rnd in (PROFILE_ROUND, "r05", "r04", "r03", "r02", "r01"):
        try:
            with open(os.path.join(ROOT, "profiles", f"{rnd}_{name}.json")) as f:
                return json.load(f), f"profiles/{rnd}_{name}.json"
        except OSError:
            continue
    return None, None


def pmc_traffic(kernel_substr, doubled=True):
    """HBM-side bytes per launch of a kernel from the committed PMC summary (None when absent)."""
    data, src = _profile_json("pmc_kernels")
    if data is None:
        return None, None
    for name, c in data.items():
        if name != "_meta" and kernel_substr in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            return int(((2.0 if doubled else 1.0) * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024), src
    return None, None


def static_profile_identity():
    """csrc_sha the committed counter files were taken on (their _meta block; None for files older than round 4)"""
    data, src = _profile_json("pmc_kernels")
    return ((data or {}).get("_meta") or {}).get("csrc_sha"), src


def pmc_detail(kernel_key):
    """fabric request counters / MFMA busy / LDS conflicts of a kernel from the committed PMC detail file"""
    data, src = _profile_json("pmc_detail")
    try:
        c = data[kernel_key]
        out = {"fabric_read_requests": int(c["TCC_EA0_RDREQ_sum"]),
               "share_of_128B_requests": round(c["TCC_EA0_RDREQ_128B_sum"] / c["TCC_EA0_RDREQ_sum"], 4),
               "l2_hit_rate": round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 3),
               # GRBM_GUI_ACTIVE is summed over the 8 XCDs (8 x 2.3 GHz x kernel time): busy SIMD-cycles / (cycles x 1024 SIMDs)
               "mfma_busy_frac": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), 4),
               "source": src}
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
            out["lds_bank_conflict_frac"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 3)
        return out
    except (TypeError, KeyError, ZeroDivisionError):
        return None


def _loaded_csrc_sha():
    from torch_rgcn import _native
    return _native.csrc_sha()


def kernel_roofline(kernel, avg_ms, alg_bytes, bytes_model, step_ms, launches_per_step, pmc_substrs, pmc_key):
    """roofline block of one kernel (group): achieved = SURVEY 8(d) ALGORITHMIC bytes / the average launch time measured
    live with HIP events inside the timed region; traffic / pmc are STATIC (the committed rocprofv3 --pmc summaries of
    the same S1 launch under profiles/), tagged as such."""
    ach = alg_bytes / (avg_ms * 1e-3) / 1e9
    traffic = traffic64 = tsrc = None
    for sub in pmc_substrs:
        traffic, tsrc = pmc_traffic(sub)
        traffic64, _ = pmc_traffic(sub, doubled=False)
        if traffic:
            break
    return {"kernel": kernel, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "avg_launch_ms": round(avg_ms, 4), "launches_per_step": launches_per_step,
            "share_of_step": round(launches_per_step * avg_ms / step_ms, 4),
            "algorithmic_bytes_per_launch": alg_bytes, "bytes_model": bytes_model,
            "traffic": traffic, "traffic_if_64B_requests": traffic64,
            "traffic_rate_GBs": round(traffic / (avg_ms * 1e-3) / 1e9, 1) if traffic else None,
            "traffic_static_csrc_sha": static_profile_identity()[0] if traffic else None,
            "traffic_static_csrc_match": (static_profile_identity()[0] == _loaded_csrc_sha()) if traffic else None,
            "traffic_static": (f"{tsrc}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on the same S1 launch, NOT measured in "
                               "this run; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B, "
                               "MI355X_MICROARCH.md HBM section)") if traffic else None,
            "pmc": pmc_detail(pmc_key) if pmc_key else None,
            # context, STATIC: the random gather of one 64-byte row per message is served as whole 128-byte lines whatever the load
            # flavour (tools/micro/gather64.hip) -- a kernel that does nothing but S1's gather takes this long per pass
            "gather_only_floor_ms_static": 0.373, "gather_only_floor_source": "profiles/r04_gather64.txt",
            # ... in RANDOM order.  In window-major order (the whole chip reading from 1-4 MB of the table at a time) the same reads take 0.19-0.22 ms
            # (tools/micro/gather_window.hip): the order round 6's soft-window plans approximate with ~5 MB spans
            "gather_window_major_floor_ms_static": 0.2, "gather_window_major_source": "profiles/r06_gather_window.txt"}


class _MeanSquare(torch.autograd.Function):
    """loss = mean(out^2) as two kernels (a dot product; one scaled copy in backward) instead of the five elementwise /
    reduction kernels of out.pow(2).mean() -- the loss is part of the timed step but not the thing measured."""

    @staticmethod
    def forward(ctx, out):
        ctx.save_for_backward(out)
        return torch.linalg.vector_norm(out.reshape(-1)).square() / out.numel()      # (an ATen reduction; torch.dot is a rocBLAS call)

    @staticmethod
    def backward(ctx, g):
        out, = ctx.saved_tensors
        return out * (g * (2.0 / out.numel()))


def _backend_version(backend):
    """what the line says about the collective library -- never worth the headline number: any failure here is reported as text"""
    try:
        if backend == "nccl":
            v = torch.cuda.nccl.version()
            return "RCCL " + (".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v))
        return f"{backend} (torch {torch.__version__})"
    except Exception as exc:  # noqa: BLE001
        return f"{backend} (version query failed: {type(exc).__name__})"


def make_step(l1, l2, X):
    """THE timed step: layer 1 (horizontal flag) with its ReLU fused, layer 2 (vertical flag), loss = mean(out^2), backward to X and
    to both layers' parameters (tests/test_gpu_parity.py::test_bench_step_composition_vs_cpu_port checks exactly this callable)"""
    def step():
        for p in (X, l1.weights, l1.bias, l2.weights, l2.bias):
            p.grad = None
        out = l2(l1.forward_activated(X, "relu", private=True))
        loss = _MeanSquare.apply(out)
        loss.backward()
        return loss
    return step


def build_layers(N, R0, E, d, seed, device, group, keep):
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
        shard_layer(l1, group, keep=keep)
        shard_layer(l2, group, keep=keep)
    return l1, l2, tp.size(0)


def cpu_baseline(full_scale):
    """The reference's op sequence (oracle/torch_cpu_port.py) on the host cores: S1 at 1/10 scale (3 timed steps) and,
    when the host has the memory for the dense R x N x d intermediates (~15 GB), S1 itself (1 warm-up + 2 timed)."""
    from oracle import oracle, torch_cpu_port
    # ATen's sparse kernels stop scaling -- and then regress -- long before a many-core host is used up (thread sweep at 1/10 of S1 on
    # the GPU box's host: profiles/r04_cpu_threads.json, tools/cpu_threads_probe.py): the baseline gets the best thread count measured
    # there (4 .. 16 threads: 2.42 s per step; 32: 2.70; 128: 4.19; all 256: 12.7), not all cores; `cores_available` rides along.
    threads = int(os.environ.get("RGCN_CPU_THREADS", min(os.cpu_count() or 1, 16)))
    torch.set_num_threads(threads)

    def leg(N, R0, E, d, steps):
        tp = torch.from_numpy(oracle.add_inverse_and_self(oracle.synthetic_triples(N, R0, E, 0), N, R0))
        R = 2 * R0 + 1
        g = torch.Generator().manual_seed(0)
        base = [torch.randn(N, d, generator=g), torch.randn(R, d, d, generator=g) * 0.2, torch.zeros(d),
                torch.randn(R, d, d, generator=g) * 0.2, torch.zeros(d)]
        times = []
        for _ in range(steps + 1):
            ts = [t.clone().requires_grad_(True) for t in base]
            t0 = time.perf_counter()
            torch_cpu_port.two_layer_step(tp, N, R, *ts)
            times.append(time.perf_counter() - t0)
        return E / min(times[1:]), min(times[1:])

    small_v, small_t = leg(100_000, 50, 1_000_000, 16, 3)
    res = {"value": small_v, "unit": "edges/s", "cores": threads, "cores_available": os.cpu_count(), "kind": "port",
           "threads_note": "min(cores, 16): ATen's sparse kernels are flat from 4 to 16 threads and regress past that (profiles/r04_cpu_threads.json)",
           "sample": f"S1 at 1/10 scale (N=100000, E=1000000, R0=50, d=16), 1 warm-up + 3 timed steps, best; {small_t:.2f} s/step",
           "port_vs_reference": "profiles/r02_port_vs_reference.json (tools/port_vs_reference.py, build container)"}
    if full_scale:
        v, t = leg(1_000_000, 50, 10_000_000, 16, 2)
        res.update({"value": v, "sample": f"S1 itself (N=1000000, E=10000000, R0=50, d=16), 1 warm-up + 2 timed steps, best; "
                                          f"{t:.2f} s/step", "tenth_scale": {"value": small_v, "s_per_step": round(small_t, 3)}})
    return res


def self_launch(n):
    """`python bench.py --gpus N` without a launcher around it: re-run this command line under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` (one process per GPU, RCCL over xGMI) on a free local
    port.  The ranks' stderr passes through; of their stdout only rank 0's JSON line is printed, last."""
    import socket
    import subprocess
    if not os.environ.get("RGCN_BENCH_ONE_DEVICE"):
        have = torch.cuda.device_count()
        assert have >= n, f"--gpus {n}: this node shows {have} GPU(s)"
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    result = [ln for ln in lines if ln.lstrip().startswith("{") and '"metric"' in ln]
    for ln in lines:
        if not result or ln is not result[-1]:
            print(ln, file=sys.stderr)           # anything else the ranks wrote to stdout (RCCL banners ...) is not the result
    if proc.returncode != 0 or not result:
        print(f"bench.py: the {n}-rank run failed (exit code {proc.returncode})", file=sys.stderr)
        return proc.returncode or 1
    print(result[-1], flush=True)
    return 0


LINE_TARGET_BYTES = 4096     # what the line aims for
LINE_LIMIT_BYTES = 8192      # the driver keeps an 8 KB tail of stdout: a longer line cannot be parsed (round 5's was 27 KB: unmeasured)

_ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "launches_per_step", "share_of_step",
              "algorithmic_bytes_per_launch", "traffic", "traffic_over_algorithmic", "traffic_source", "step_frac",
              "step_algorithmic_bytes")
_CPU_KEYS = ("value", "unit", "cores", "kind", "sample")
_COMM_KEYS = ("collective", "collectives_per_step", "bytes_per_collective", "compute_alone_ms_per_step", "collectives_alone_ms_per_step",
              "exposed_ms_per_step", "allreduce_algbw_GBs", "world_size_seen_by_backend", "backend", "backend_version", "overlap")


def _pick(d, keys):
    return {k: d[k] for k in keys if k in d}


def _kernel_summary(r):
    """the few numbers of a roofline block that belong on the line (the rest lives in the detail file)"""
    out = _pick(r, _ROOF_KEYS)
    if out.get("traffic") and out.get("algorithmic_bytes_per_launch"):
        out["traffic_over_algorithmic"] = round(out["traffic"] / out["algorithmic_bytes_per_launch"], 3)
    if r.get("traffic_static"):
        out["traffic_source"] = str(r["traffic_static"]).split(":")[0] + " (static, csrc match: " + str(r.get("traffic_static_csrc_match")) + ")"
    return out


def compact_line(res, detail_path=None):
    """The ONE stdout line: the contract's keys + `roofline` + `cpu_baseline` (+ `comm` for N > 1), nothing that grows with the number of
    kernels or configs.  Everything else -- per-config lines, per-kernel counters, the forward/backward sub-blocks, notes -- is the detail
    file's.  Asserted < LINE_LIMIT_BYTES (tests/test_bench_line.py runs this on a canned result)."""
    line = _pick(res, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                       "dtype", "data", "per_gpu_edges_per_s", "step_ms_median", "step_ms_min", "graph_build_ms", "csrc_sha"))
    line["config"] = _pick(res.get("config") or {}, ("workload", "sharding"))
    roof = res.get("roofline")
    if roof:
        r = _kernel_summary(roof)
        for side in ("forward", "backward"):
            if side in roof:
                r[side] = _pick(_kernel_summary(roof[side]), ("kernel", "avg_launch_ms", "frac", "algorithmic_bytes_per_launch", "traffic",
                                                             "traffic_over_algorithmic"))
                r[side]["kernel"] = str(r[side].get("kernel", "")).split(" ")[0]
        r["kernel"] = str(r.get("kernel", ""))[:120]
        line["roofline"] = r
    else:
        line["roofline"] = None
    if res.get("sustained"):
        line["sustained_ms_per_step"] = res["sustained"].get("ms_per_step")
    if res.get("comm"):
        line["comm"] = _pick(res["comm"], _COMM_KEYS)
        line["comm"]["candidates_ms_per_step"] = res["comm"].get("candidates_ms_per_step")
    if res.get("sharded_vs_unsharded"):
        line["sharded_vs_unsharded"] = _pick(res["sharded_vs_unsharded"], ("max_rel_err",))
    if isinstance(res.get("cpu_baseline"), dict):
        line["cpu_baseline"] = _pick(res["cpu_baseline"], _CPU_KEYS)
        line["cpu_baseline"]["sample"] = str(line["cpu_baseline"].get("sample", ""))[:200]
    if isinstance(res.get("configs"), list):     # one short entry per secondary config; their full lines are in the detail file
        line["configs_ms"] = {f"{i}:" + " ".join(str(c.get("baseline_config", "")).split()[:3])[:28]: c.get("ms_per_step", c.get("failed"))
                              for i, c in enumerate(res["configs"]) if isinstance(c, dict)}
    if detail_path:
        line["detail"] = detail_path
    text = json.dumps(line, allow_nan=False)
    if len(text) >= LINE_TARGET_BYTES:           # past the target: drop the optional blocks, largest first, until it fits
        for k in ("configs_ms", "sustained_ms_per_step", "step_ms_median", "step_ms_min", "per_gpu_edges_per_s"):
            line.pop(k, None)
            text = json.dumps(line, allow_nan=False)
            if len(text) < LINE_TARGET_BYTES:
                break
    assert len(text) < LINE_LIMIT_BYTES, f"bench line is {len(text)} bytes (limit {LINE_LIMIT_BYTES})"
    assert "\n" not in text
    return text


def _no_nan(o):
    """json.dumps(allow_nan=False)-safe copy: NaN / inf become None (a strict parser must read both the line and the detail file)"""
    if isinstance(o, float):
        return o if np.isfinite(o) else None
    if isinstance(o, dict):
        return {str(k): _no_nan(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_no_nan(v) for v in o]
    if isinstance(o, (np.floating, np.integer)):
        return _no_nan(o.item())
    return o


def emit(res):
    """detail -> bench_detail.json (+ stderr); the compact line -> stdout, LAST"""
    res = _no_nan(res)
    path = os.environ.get("RGCN_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))
    rel = None
    try:
        with open(path, "w") as f:
            json.dump(res, f, indent=1, allow_nan=False)
        rel = os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
    except OSError as exc:
        print(f"bench.py: detail file not written ({exc})", file=sys.stderr)
    print("bench.py detail: " + json.dumps(res, allow_nan=False), file=sys.stderr, flush=True)
    sys.stdout.flush()
    print(compact_line(res, rel), flush=True)


def host_ram_gb():
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--rels", type=int, default=50)
    ap.add_argument("--edges", type=int, default=10_000_000)
    ap.add_argument("--hidden", type=int, default=16)
    ap.add_argument("--weak", action="store_true", help="N > 1: every rank its own relations/triples (round 1's mode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-s1", action="store_true", help="CPU baseline at 1/10 scale only")
    ap.add_argument("--no-configs", action="store_true", help="skip the secondary lines for BASELINE configs 1-4")
    ap.add_argument("--check-unsharded", action="store_true", help="N > 1, strong mode: rank 0 also runs the step on the whole graph and the line "
                    "reports the largest relative difference of loss / dX / dW (tests; not part of the timed region)")
    ap.add_argument("--sustained-steps", type=int, default=400,
                    help="N = 1: untimed-for-the-headline extra leg after the K timed steps (clocks / thermals over ~1 s); 0: skip")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- one rank per GPU under torch.distributed.run, and pass
        # rank 0's ONE JSON line through as the last line of stdout
        return self_launch(args.gpus)

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
        seen = dist.get_world_size(group)
        if seen != args.gpus:      # the line's n_gpus must be what the collective library ran with, not what the command line asked for
            print(f"bench.py: --gpus {args.gpus} but the {backend} backend sees {seen} rank(s)", file=sys.stderr)
            dist.destroy_process_group()
            return 3
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    mode = "single" if group is None else ("weak" if args.weak else "strong")

    from torch_rgcn import _native, routes
    from torch_rgcn.dist import set_transport
    N, R0, E, d = args.nodes, args.rels, args.edges, args.hidden
    l1, l2, M = build_layers(N, R0, E, d, seed=rank if mode == "weak" else 0, device=device, group=group,
                             keep="all" if mode == "weak" else "lpt")
    torch.manual_seed(99)
    X = torch.randn(N, d, device=device, requires_grad=True)

    # one-off graph preprocessing (NC graphs are static): normalisation + the plans one step needs, both layers
    torch.cuda.synchronize()
    t_b = time.perf_counter()
    for layer in (l1, l2):
        g = layer._graph_on(device)
        if group is None and g.win_plan("fwd") is not None and g.win_plan("bwd_own") is not None:
            continue          # round 6, one GPU: the soft-window plans are the only ones the step walks (forward, relation-owner backward)
        g.fwd_plan(d)
        if group is not None or g.bwd_blk_plan() is None:      # (sharded ranks may fall back to spmm on the wave-owned plan)
            g.bwd_plan(d)
    torch.cuda.synchronize()
    graph_build_ms = 1e3 * (time.perf_counter() - t_b)
    my_messages = l1._graph.num_messages

    step = make_step(l1, l2, X)

    def fence():
        torch.cuda.synchronize()
        if group is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_steps(k):
        step()
        fence()
        t_c = time.perf_counter()
        for _ in range(k):
            step()
        fence()
        tc = torch.tensor([time.perf_counter() - t_c], device=device, dtype=torch.float64)
        if group is not None:
            dist.all_reduce(tc, op=dist.ReduceOp.MAX)
        return 1e3 * tc.item() / k

    comm = None
    if group is not None:
        # Untimed set-up: which form of the collective is fastest on THIS link (xGMI ring / direct, one rank, gloo)?
        # 3 steps per candidate, max over ranks, the same choice on every rank.
        cands = [("allreduce", "0"), ("rs_ag", "0"), ("a2a", "0"), ("allreduce", "2")] + ([("allreduce", "4")] if mode == "weak" else [])
        if routes.is_set("dist_comm") or routes.is_set("dist_slabs"):
            cands = [(routes.get("dist_comm", "allreduce"), routes.get("dist_slabs", "0"))]
        tried = {}
        for c, s in cands:
            set_transport(l1, c, int(s)), set_transport(l2, c, int(s))
            try:
                tried[f"{c}/slabs={s}"] = round(timed_steps(3), 4)
            except RuntimeError as exc:       # a collective this backend lacks fails on every rank alike: drop the candidate
                tried[f"{c}/slabs={s}"] = None
                if rank == 0:
                    print(f"bench.py: transport {c}/slabs={s} unavailable: {str(exc)[:120]}", file=sys.stderr)
        best = min((k for k in tried if tried[k] is not None), key=tried.get)
        chosen = (best.split("/slabs=")[0], int(best.split("/slabs=")[1]))
        set_transport(l1, *chosen), set_transport(l2, *chosen)
        comm = {"collective": best, "candidates_ms_per_step": tried}

    for _ in range(args.warmup):
        step()
    fence()
    # The per-launch timers (HIP events on the launch stream around every launch of the path's kernels: the roofline's `avg_launch_ms`) are
    # SAMPLED: every `sample_every`-th step of the timed region carries them.  An event pair costs ~9 us of GPU time -- the launches on either
    # side cannot overlap their tails and heads -- and with all four launches of every step bracketed the timed region measured the timers
    # (1.845 ms per step against 1.81 for the same loop without them, tools/prof_overhead.py).
    sample_every = max(1, min(4, args.steps // 4))
    n_sampled = len(range(0, args.steps, sample_every))
    _native.profile_start()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        _native.profile_enable(k % sample_every == 0)
        marks[k].record()                 # HIP events on the launch stream: per-step GPU time (median / min below)
        step()
    marks[args.steps].record()
    fence()
    elapsed = time.perf_counter() - t0
    _native.profile_enable(True)
    prof = _native.profile_stop()
    per_step = [marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps)]

    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if group is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()
    ms = 1e3 * elapsed / args.steps

    shard_check = None
    if group is not None and args.check_unsharded and mode == "strong":
        # sharded == unsharded on the very step that was timed: dX is complete on every rank after the join, the weight rows live on
        # their owners (disjoint: their sum over ranks is the whole gradient); rank 0 repeats the step on the whole graph
        loss_s = step().detach().clone()
        got = [X.grad.detach().clone(), l1.weights.grad.detach().clone(), l2.weights.grad.detach().clone()]
        for gr in got[1:]:
            dist.all_reduce(gr)
        if rank == 0:
            u1, u2, _ = build_layers(N, R0, E, d, seed=0, device=device, group=None, keep="all")
            Xu = X.detach().clone().requires_grad_(True)
            loss_u = make_step(u1, u2, Xu)().detach()
            ref = [Xu.grad, u1.weights.grad, u2.weights.grad]
            errs = {n: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for n, a, b in zip(("dX", "dW1", "dW2"), got, ref)}
            errs["loss"] = float((loss_s - loss_u).abs() / loss_u.abs().clamp_min(1e-30))
            shard_check = {"max_rel_err": max(errs.values()), **{k: float(f"{v:.3e}") for k, v in errs.items()}}
            del u1, u2, Xu
        fence()

    if group is not None:
        # what the collectives cost: (a) the step without them (RGCN_DIST_COMM=none: same kernels, wrong numbers),
        # (b) the four N x d collectives of one step on their own
        # (a bench-local patch of the join -- there is no such transport in torch_rgcn.dist: nothing else can switch the sums off)
        import torch_rgcn.functional as _fn
        join = _fn._join_shards
        _fn._join_shards = lambda partial, grp, mode="allreduce": partial
        set_transport(l1, "allreduce", 0), set_transport(l2, "allreduce", 0)
        try:
            compute_ms = timed_steps(3)
        finally:
            _fn._join_shards = join
            set_transport(l1, *chosen), set_transport(l2, *chosen)
        _join_shards = join
        buf = torch.zeros(N, d, device=device)
        fence()
        t_c = time.perf_counter()
        for _ in range(3 * 4):
            _join_shards(buf, group, chosen[0])
        fence()
        comm_ms = 1e3 * (time.perf_counter() - t_c) / 3
        counts = torch.zeros(world, device=device, dtype=torch.float64)
        counts[rank] = my_messages
        dist.all_reduce(counts)
        bytes_per_coll = N * d * 4
        comm.update({"collectives_per_step": 4, "bytes_per_collective": bytes_per_coll,
                     "compute_alone_ms_per_step": round(compute_ms, 4),
                     "collectives_alone_ms_per_step": round(comm_ms, 4),
                     "exposed_ms_per_step": round(ms - compute_ms, 4),
                     "allreduce_algbw_GBs": round(4 * bytes_per_coll / (comm_ms * 1e-3) / 1e9, 1) if comm_ms > 0 else None,
                     "messages_per_rank": [int(c) for c in counts.tolist()],
                     # what the backend itself saw (the driver's SCALE record can be checked for "RCCL ran with N ranks")
                     "world_size_seen_by_backend": dist.get_world_size(group), "backend": str(dist.get_backend(group)),
                     "backend_version": _backend_version(backend),
                     "overlap": "side-stream" if chosen[1] > 0 else "none",
                     "model": "per step 4 x all-reduce of N x d fp32 (2 forward outputs, 2 feature gradients); a ring moves "
                              "2(G-1)/G x 64 MB over each rank's slowest link, a direct reduce-scatter + all-gather "
                              "2(G-1)/G x 64 MB spread over G-1 links (DESIGN.md section 6)"})

    sustained = None
    if world == 1 and group is None and args.sustained_steps > 0:
        # the timed region is K steps (~45 ms at the defaults): a second, longer leg of the SAME step shows whether the figure holds
        # once clocks and temperature have settled.  Reported next to the headline, never instead of it.
        n_s = args.sustained_steps
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        fence()
        for q in range(4):
            ev[q].record()
            for _ in range(n_s // 4):
                step()
        ev[4].record()
        fence()
        quarters = [ev[q].elapsed_time(ev[q + 1]) / (n_s // 4) for q in range(4)]
        sustained = {"steps": 4 * (n_s // 4), "ms_per_step": round(float(np.mean(quarters)), 4),
                     "ms_per_step_by_quarter": [round(v, 4) for v in quarters],
                     "note": "HIP events around four consecutive quarters of the leg, after the timed region; same step, same inputs"}

    if rank == 0:
        total_edges = E * (world if mode == "weak" else 1)
        value = total_edges / (ms * 1e-3)
        alg = fwd_bytes(M, N, d, d)
        launches = {k: (float(np.mean(v)), len(v) / n_sampled) for k, v in prof.items()}
        spmm_key = next((k for k in ("spmm_blk", "spmm", "spmm_slab") if k in launches), None)
        fwd_kernel = "spmm_blk_d16_kernel" if spmm_key == "spmm_blk" else "spmm_d16_kernel"
        roof = None
        step_alg = 2 * (fwd_bytes(M, N, d, d) + bwd_bytes(M, N, d, d))
        if spmm_key and mode != "strong":
            spmm_ms = launches[spmm_key][0]
            slabbed = spmm_key == "spmm_slab"
            n_spmm = 2 if "bwd_fused" in launches else 4       # launches per step: 2 forward (+ 2 feature-gradient without the fused backward)
            if slabbed:   # one spmm = the slabs of one launch group
                spmm_ms = float(np.sum(prof[spmm_key])) / (n_spmm * n_sampled)
            fwd = kernel_roofline(fwd_kernel + " (forward launches" + ("" if "bwd_fused" in launches else " and feature-gradient launches") + ")",
                                  spmm_ms, alg, "SURVEY 8(d) forward, one layer: M(4 d_in + 8) + N 4 d_out", ms, n_spmm,
                                  (fwd_kernel,), "spmm_blk" if spmm_key == "spmm_blk" else "spmm")
            # the backward of one layer, whatever kernels it is made of (SURVEY 8d backward bytes)
            balg = bwd_bytes(M, N, d, d)
            bmodel = "SURVEY 8(d) backward, one layer: M(4 d_out + 8) + 2 N 4 d_in (X needs a gradient)"
            bwd = None
            if "bwd_fused" in launches:
                bms = launches["bwd_fused"][0] + launches.get("dw_reduce", (0.0, 0))[0]
                route = _native.bwd_route()
                if route == "blk" and not _native.bwd_blk_rows(N, 2 * R0 + 1, routes.flag("deterministic")):
                    route = "lean"
                kname, kkey = {"blk": ("bwd_blk_d16_kernel", "bwd_blk"), "lean": ("bwd_lean_d16_kernel", "bwd_lean")}.get(route, ("bwd_fused_d16_kernel", "bwd_fused"))
                if route == "blk" and routes.get("bwd_own", "1") != "0" and l2._graph.win_plan("bwd_own") is not None:
                    kname, kkey = "bwd_own_d16_kernel", "bwd_own"       # round 6: relation-owner kernel on the soft-window plan
                bwd = kernel_roofline(kname + " (dX + dW of one layer from one gather per message"
                                      + (" + dw_reduce)" if "dw_reduce" in launches else ")"), bms, balg, bmodel, ms, 2,
                                      (kname, "bwd_fused_d16_kernel") if kname != "bwd_fused_d16_kernel" else (kname,), kkey)
            elif "wgrad" in launches and not slabbed:
                bms = spmm_ms + launches["wgrad"][0]
                bwd = kernel_roofline("spmm_d16_kernel (dX) + wgrad_tiled_d16_kernel (dW)", bms, balg, bmodel, ms, 2, (), None)
            # headline = the kernel with the largest share of the timed step; the other one and the whole step ride along
            roof = dict(bwd if (bwd is not None and bwd["share_of_step"] >= fwd["share_of_step"]) else fwd)
            roof["forward"] = fwd
            if bwd is not None:
                roof["backward"] = bwd
            roof["step_algorithmic_bytes"] = step_alg
            roof["step_frac"] = round(step_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            path_ms = n_spmm * spmm_ms + (2 * bwd["avg_launch_ms"] if "bwd_fused" in launches else (2 * launches["wgrad"][0] if "wgrad" in launches else 0.0))
            roof["step_minus_path_kernels_ms"] = round(ms - path_ms, 4)
            roof["other_kernels_ms"] = {k: round(v[0], 4) for k, v in launches.items() if k != spmm_key}
        elif spmm_key:   # strong scaling: a rank's launches cover ITS share of the messages only (rank 0's shard here)
            spmm_ms = launches[spmm_key][0]
            m_local = my_messages

            def local(kernel, t_ms, alg_b, model, per_step):
                ach = alg_b / (t_ms * 1e-3) / 1e9
                return {"kernel": kernel + " on rank 0's relation shard", "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "avg_launch_ms": round(t_ms, 4),
                        "launches_per_step": per_step, "algorithmic_bytes_per_launch": int(alg_b), "bytes_model": model,
                        "share_of_step": round(per_step * t_ms / ms, 4)}
            n_spmm = 2 if "bwd_fused" in launches else 4
            fwd = local(fwd_kernel, spmm_ms, fwd_bytes(m_local, N, d, d), "SURVEY 8(d) forward on the local messages: M_local(4 d_in + 8) + N 4 d_out", n_spmm)
            bwd = None
            if "bwd_fused" in launches:
                bwd = local("fused backward (dX + dW from one gather per local message)", launches["bwd_fused"][0], bwd_bytes(m_local, N, d, d),
                            "SURVEY 8(d) backward on the local messages: M_local(4 d_out + 8) + 2 N 4 d_in", 2)
            elif "wgrad" in launches:
                bwd = local("spmm_d16_kernel (dX) + wgrad (dW)", spmm_ms + launches["wgrad"][0], bwd_bytes(m_local, N, d, d),
                            "SURVEY 8(d) backward on the local messages: M_local(4 d_out + 8) + 2 N 4 d_in", 2)
            roof = dict(bwd if (bwd is not None and bwd["share_of_step"] >= fwd["share_of_step"]) else fwd)
            roof["forward"] = fwd
            if bwd is not None:
                roof["backward"] = bwd
            roof["local_messages"] = int(m_local)
            roof["other_kernels_ms"] = {k: round(v[0], 4) for k, v in launches.items() if k != spmm_key}
        res = {"metric": METRIC, "value": value, "unit": "edges/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
               "scaling": "weak" if mode == "weak" else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "per_gpu_edges_per_s": value / world,
               "step_ms_median": round(float(np.median(per_step)), 4), "step_ms_min": round(float(np.min(per_step)), 4),
               "step_ms_p95": round(float(np.percentile(per_step, 95)), 4), "sustained": sustained,
               "graph_build_ms": round(graph_build_ms, 2), "csrc_sha": _native.csrc_sha(),
               "config": {"workload": (f"S1: N={N} nodes, E={E} base triples" + ("/GPU" if mode == "weak" else "") +
                                       f", R0={R0} relations" + ("/GPU" if mode == "weak" else "") +
                                       f" (layer R={2 * R0 + 1}), M={M} messages/layer, hidden={d}, 2 NC layers "
                                       "(horizontal, vertical), fwd+bwd, learnable X"),
                          "sharding": "single GPU" if world == 1 and group is None else
                                      (f"relation-sharded x{world} (LPT on message counts, ONE seed-0 graph), "
                                       if mode == "strong" else f"relation-sharded x{world} (each rank its own relations), ")
                                      + f"N x {d} fp32 partials joined by {comm['collective']}"},
               "step_hbm_algorithmic_GBs": round(step_alg / (ms * 1e-3) / 1e9, 1),
               "launch_timers": {"sampled_steps": n_sampled, "of_steps": args.steps,
                                 "note": "HIP event pairs around the path's launches (avg_launch_ms of the rooflines) in every "
                                         f"{sample_every}-th step of the timed region: a pair costs ~9 us of GPU time"},
               "roofline": roof}
        if comm is not None:
            res["comm"] = comm
        if shard_check is not None:
            res["sharded_vs_unsharded"] = shard_check
        if world == 1 and group is None and not args.no_configs:
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import config_bench
                res["configs"] = config_bench.secondary_lines()
            except Exception as exc:  # noqa: BLE001  (secondary lines must never cost the headline number)
                res["configs"] = f"failed: {type(exc).__name__}: {exc}"[:300]
        if world == 1 and not args.no_cpu_baseline:
            del l1, l2, X
            torch.cuda.empty_cache()
            res["cpu_baseline"] = cpu_baseline(full_scale=not args.no_cpu_s1 and host_ram_gb() >= 32.0 and
                                               (N, R0, E, d) == (1_000_000, 50, 10_000_000, 16))
    else:
        res = None
    if group is not None:
        dist.destroy_process_group()   # RCCL prints its banner here; keep the JSON line last
    if res is not None:
        emit(res)


if __name__ == "__main__":
    sys.exit(main())
