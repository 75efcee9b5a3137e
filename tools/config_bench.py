#!/usr/bin/env python3
"""One training step (forward + backward) of every BASELINE.json config on dataset-shaped synthetic graphs
(the datasets themselves are not available offline).  Prints one JSON line per config.
    python tools/config_bench.py [--cpu]     # --cpu also times the PyTorch-CPU port where it fits"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "torch-rgcn_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from torch_rgcn import _native  # noqa: E402
from torch_rgcn.layers import DistMult, RelationalGraphConvolutionLP, RelationalGraphConvolutionNC  # noqa: E402
from torch_rgcn.models import NodeClassifier  # noqa: E402

DEV = torch.device("cuda:0")
QUICK = False      # --quick: few iterations (profiling runs under rocprofv3: tools/prof.sh lines)


def timed(fn, iters=10, warm=3):
    if QUICK:
        iters, warm = min(iters, 3), 1
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


def nc_model(name, N, R0, E, nhid, ncls, decomp, labelled):
    T = _native.synthetic_triples_host(N, R0, E, 1)
    t0 = time.time()
    model = NodeClassifier(triples=T, nnodes=N, nrel=R0, nhid=nhid, nclass=ncls, decomposition=decomp).to(DEV)
    idx = torch.arange(labelled, device=DEV)
    y = torch.randint(0, ncls, (labelled,), device=DEV)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, fused=True)      # as experiments/classify_nodes.py does

    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(model()[idx], y)
        loss.backward()
        opt.step()
    step()
    torch.cuda.synchronize()
    build = time.time() - t0
    ms = timed(step)
    # the same step captured in a hipGraph (static NC graph, static shapes): launch-bound at this size
    ms_graph = None
    try:
        opt_g = torch.optim.Adam(model.parameters(), lr=0.01, fused=True, capturable=True)       # as experiments/classify_nodes.py does
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                opt_g.zero_grad(set_to_none=True)
                torch.nn.functional.cross_entropy(model()[idx], y).backward()
                opt_g.step()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        opt_g.zero_grad(set_to_none=True)
        with torch.cuda.graph(g):
            loss_g = torch.nn.functional.cross_entropy(model()[idx], y)
            loss_g.backward()
            opt_g.step()
        ms_graph = timed(g.replay)
    except Exception as exc:  # noqa: BLE001
        ms_graph = f"capture failed: {type(exc).__name__}: {exc}"[:200]
    print(json.dumps({"config": name, "N": N, "R0": R0, "E": E, "ms_per_train_step": round(ms, 3),
                      "ms_per_train_step_hipgraph": round(ms_graph, 3) if isinstance(ms_graph, float) else ms_graph,
                      "edges_per_s": round(E / ms * 1e3), "first_step_incl_graph_build_s": round(build, 2),
                      "params": sum(p.numel() for p in model.parameters())}), flush=True)


def s2_featureless_basis():
    """S1 with a featureless first layer (weight-table gather), basis B = 2."""
    N, R0, E, d = 1_000_000, 50, 10_000_000, 16
    T = _native.synthetic_triples_host(N, R0, E, 0)
    tp = torch.from_numpy(_native.add_inverse_and_self_host(T, N, R0))
    kw = dict(triples=tp, num_nodes=N, num_relations=2 * R0 + 1, out_features=d)
    l1 = RelationalGraphConvolutionNC(in_features=None, vertical_stacking=False,
                                      decomposition={"type": "basis", "num_bases": 2}, **kw).to(DEV)
    l2 = RelationalGraphConvolutionNC(in_features=d, vertical_stacking=True, **kw).to(DEV)

    def step():
        for p in list(l1.parameters()) + list(l2.parameters()):
            p.grad = None
        l2(torch.relu(l1())).pow(2).mean().backward()
    ms = timed(step, iters=5, warm=2)
    print(json.dumps({"config": "S2: S1 graph, featureless layer 1 with basis B=2 (no R x N x 16 table is materialised), layer 2 16->16",
                      "N": N, "R0": R0, "E": E, "ms_per_fwd_bwd": round(ms, 3), "edges_per_s": round(E / ms * 1e3)}), flush=True)


def featured_layers(title, N, R0, E, d, decomposition, seed):
    T = _native.synthetic_triples_host(N, R0, E, seed)
    tp = torch.from_numpy(_native.add_inverse_and_self_host(T, N, R0))
    kw = dict(triples=tp, num_nodes=N, num_relations=2 * R0 + 1, in_features=d, out_features=d,
              decomposition=decomposition)
    l1 = RelationalGraphConvolutionNC(vertical_stacking=False, **kw).to(DEV)
    l2 = RelationalGraphConvolutionNC(vertical_stacking=True, **kw).to(DEV)
    X = torch.randn(N, d, device=DEV, requires_grad=True)

    def step():
        for p in [X] + list(l1.parameters()) + list(l2.parameters()):
            p.grad = None
        l2(torch.relu(l1(X))).pow(2).mean().backward()
    ms = timed(step)
    print(json.dumps({"config": title, "N": N, "R0": R0, "E": E,
                      "ms_per_fwd_bwd": round(ms, 3), "edges_per_s": round(E / ms * 1e3)}), flush=True)


def wn18_lp():
    N, R0, d = 40_943, 18, 200
    ed = {"general": 0.5, "self_loop": 0.2, "self_loop_type": "schlichtkrull-dropout"}
    layer = RelationalGraphConvolutionLP(num_nodes=N, num_relations=2 * R0 + 1, in_features=d, out_features=d,
                                         edge_dropout=ed, decomposition={"type": "basis", "num_bases": 2},
                                         w_init="glorot-normal", b_init="zeros").to(DEV)
    dm = DistMult(R0, d, N, R0).to(DEV)
    emb = torch.randn(N, d, device=DEV, requires_grad=True)
    for tag, E in (("train graph 15k", 15_000), ("eval graph 141k", 141_442)):
        graph = torch.from_numpy(_native.synthetic_triples_host(N, R0, E, 3))
        batch = torch.from_numpy(_native.synthetic_triples_host(N, R0, 330_000, 4)).to(DEV)
        y = torch.rand(330_000, device=DEV).round()

        def step():
            for p in [emb] + list(layer.parameters()) + list(dm.parameters()):
                p.grad = None
            x = layer(graph, torch.relu(emb))
            loss = torch.nn.functional.binary_cross_entropy_with_logits(dm(batch, x), y)
            loss.backward()
        ms = timed(step, iters=5, warm=2)
        t0 = time.perf_counter()
        with torch.no_grad():
            layer(graph, emb)
        torch.cuda.synchronize()
        fwd = 1e3 * (time.perf_counter() - t0)
        print(json.dumps({"config": f"WN18-shaped LP layer d=200 basis 2 + DistMult 330k triples, {tag}", "N": N,
                          "graph_triples": E, "ms_per_fwd_bwd": round(ms, 2), "encoder_fwd_ms_incl_host_graph_build": round(fwd, 2)}),
              flush=True)


# ------------------------------------------------------------------ secondary bench lines (bench.py `configs`)
HBM_PEAK_GBS = 8000.0


def _dominant(step, iters=3):
    """per-kernel launch times (HIP events on the launch stream) of `iters` steps -> (name, avg ms per launch, launches per step, all)"""
    _native.profile_start()
    for _ in range(iters):
        step()
    prof = _native.profile_stop()
    if not prof:
        return None, None, None, {}
    tot = {k: float(np.sum(v)) for k, v in prof.items()}
    name = max(tot, key=tot.get)
    return name, float(np.mean(prof[name])), len(prof[name]) / iters, {k: round(float(np.mean(v)), 4) for k, v in prof.items()}


def _launch_counts(step, iters=2):
    """launches per step of every timed kernel name"""
    _native.profile_start()
    for _ in range(iters):
        step()
    prof = _native.profile_stop()
    return {k: len(v) / iters for k, v in prof.items()}


def _count_launches(step):
    """GPU kernel launches of ONE step, counted by torch.profiler (roctracer: every kernel of the process, this library's included)"""
    try:
        from torch.profiler import ProfilerActivity, profile
        import warnings
        step()
        torch.cuda.synchronize()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                step()
                torch.cuda.synchronize()
        evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
        copies = sum(1 for e in evs if e.name.startswith(("Memcpy", "Memset")))
        return {"kernels": len(evs) - copies, "memcpy_memset": copies}
    except Exception as exc:  # noqa: BLE001
        return {"failed": f"{type(exc).__name__}: {exc}"[:120]}


PROFILE_ROUND = "r06"
# the kernels behind a timed tag of torch_rgcn._native (HIP-event timers) as rocprofv3 names them: (substrings, kernels per call of the tag)
TAG_KERNELS = {
    "fbasis_tile_fwd": (("fbn_fwd_kernel", "fbt_fwd_kernel"), 1), "gather_rows_sum4": (("gather_rows_sum4_kernel",), 1),
    "fbasis_tile_bwd": (("fbn_bwd_kernel", "fbn_dbases_kernel", "fbn_dcomps_kernel", "fbt_dbases_kernel", "fbt_dcomps_kernel"), None),
    "spmm": (("spmm_d16_kernel", "spmm_wide_kernel", "spmm_generic_kernel"), 1), "spmm_csr": (("spmm_csr_d16_kernel",), 1),
    "spmm_scatter": (("spmm_scatter_d16_kernel",), 1), "segment_sum": (("segment_gather_sum", "segment_sum"), 1),
    "bwd_scatter_dw": (("bwd_scatter_dw_d16_kernel",), 1), "bwd_fused": (("bwd_blk_d16_kernel", "bwd_lean_d16_kernel"), 1),
    "block_spmm": (("block44_csr_kernel", "block_csr"), 1), "basis_aggregate": (("basis_aggregate",), 1),
    "fbasis_small_bwd": (("fbasis_small_bwd_kernel",), 1), "fbasis_fwd": (("fbasis_fwd_kernel",), 1), "fbasis_bwd": (("fbasis_bwd",), 1),
    "featureless_csr_fwd": (("featureless_csr_fwd",), 1), "featureless_csr_wgrad": (("featureless_csr_wgrad",), 1),
    "gemm": (("gemm_kernel", "gemm_panel_kernel"), 1), "distmult_fwd": (("distmult_fwd_kernel",), 1), "distmult_bwd_all": (("distmult_bwd_all_kernel",), 2),
    "basis_dcomps_csr": (("basis_dcomps_csr_kernel",), 1), "basis_dcomps": (("basis_dcomps_kernel",), 1),
    "ce_head": (("ce_head_kernel",), 1), "bce_head": (("bce_head_kernel",), 1), "colsum": (("colsum",), None),
}
_PMC_CACHE = {}


def _pmc_file(line_key):
    """the committed counter summary of this line (tools/prof.sh lines -> profiles/r06_<line>_pmc.json), or None"""
    if line_key not in _PMC_CACHE:
        try:
            with open(os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_{line_key}_pmc.json")) as f:
                _PMC_CACHE[line_key] = json.load(f)
        except OSError:
            _PMC_CACHE[line_key] = None
    return _PMC_CACHE[line_key]


def _tag_traffic(line_key, tag):
    """HBM-side bytes per CALL of a timed tag from the committed rocprofv3 --pmc summary of this line (STATIC: not measured in this run;
    bytes = (FETCH_SIZE x (1 + share of 128-byte requests) + WRITE_SIZE) x 1024, MI355X_MICROARCH.md) -> dict or None"""
    data = _pmc_file(line_key)
    if not data or tag not in TAG_KERNELS:
        return None
    subs, kpc = TAG_KERNELS[tag]
    hits = [(k, v) for k, v in data.items() if k != "_meta" and any(sub in k for sub in subs) and "hbm_bytes_per_launch" in v]
    if not hits:
        return None
    launches = sum(max(v.get("launches_traced", 1), 1) for _, v in hits)
    if kpc is None:
        kpc = len(hits) if tag != "colsum" else 1
    calls = launches / kpc
    total = sum(v["hbm_bytes_per_launch"] * max(v.get("launches_traced", 1), 1) for _, v in hits)
    us = sum(v.get("avg_us", 0.0) * max(v.get("launches_traced", 1), 1) for _, v in hits) / calls
    out = {"traffic": int(total / calls), "kernels": sorted(k.split("(")[0][-60:] for k, _ in hits), "avg_us_under_rocprof": round(us, 1)}
    for extra in ("l2_hit_rate", "lds_bank_conflict_frac", "mfma_busy_frac", "share_of_128B_read_requests"):
        vals = [v[extra] for _, v in hits if extra in v]
        if vals:
            out[extra] = round(sum(vals) / len(vals), 3)
    return out


def _attach_traffic(line, line_key, models=None):
    """roofline.traffic of the line's dominant kernel + the traffic of every timed kernel of the step, tagged with the binary it was taken on"""
    data = _pmc_file(line_key)
    roof = line.get("roofline")
    if not data or not isinstance(roof, dict):
        if isinstance(roof, dict):
            roof.setdefault("traffic", None)
        return line
    meta = data.get("_meta", {})
    per = {}
    for tag in line.get("kernels_ms", {}):
        t = _tag_traffic(line_key, tag)
        if t:
            if models and tag in models:
                t["algorithmic_bytes"] = int(models[tag])
                t["traffic_over_algorithmic"] = round(t["traffic"] / models[tag], 3)
            per[tag] = t
    roof["kernels_traffic"] = per
    dom = roof.get("dominant_tag")
    roof["traffic"] = per[dom]["traffic"] if dom in per else (sum(v["traffic"] for v in per.values()) if roof.get("traffic_is_step_sum") else None)
    roof["traffic_static"] = f"profiles/{PROFILE_ROUND}_{line_key}_pmc.json: rocprofv3 --pmc (separate passes, tools/prof.sh lines {line_key}), NOT measured in this run"
    roof["traffic_static_csrc_sha"] = meta.get("csrc_sha")
    roof["traffic_static_csrc_match"] = meta.get("csrc_sha") == _native.csrc_sha()
    return line


def _roof(name, ms, alg_bytes, note):
    if not name or not ms:
        return None
    ach = alg_bytes / (ms * 1e-3) / 1e9
    return {"kernel": name, "dominant_tag": name, "bound": "hbm", "avg_launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": int(alg_bytes),
            "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "bytes_model": note}


def line_node_classifier(tag, N, R0, E, nhid, ncls, decomp, labelled, baseline_config, key=None):
    T = _native.synthetic_triples_host(N, R0, E, 1)
    model = NodeClassifier(triples=T, nnodes=N, nrel=R0, nhid=nhid, nclass=ncls, decomposition=decomp).to(DEV)
    idx = torch.arange(labelled, device=DEV)
    y = torch.randint(0, ncls, (labelled,), device=DEV)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, fused=True)      # as experiments/classify_nodes.py does

    from torch_rgcn.functional import MaskedCrossEntropy, unit_gradient
    head = MaskedCrossEntropy(idx, y, N)        # what experiments/classify_nodes.py uses (one launch for loss + gradient)
    unit = unit_gradient(DEV)

    def step():
        opt.zero_grad(set_to_none=True)
        head(model()).backward(gradient=unit)
        opt.step()
    ms = timed(step, iters=10, warm=3)
    name, kms, per_step, allk = _dominant(step)
    # the same step as a hipGraph replay (static graph, static shapes: these small graphs are launch-bound in eager mode)
    ms_graph = None
    try:
        opt_g = torch.optim.Adam(model.parameters(), lr=0.01, fused=True, capturable=True)       # as experiments/classify_nodes.py does

        def gstep():
            opt_g.zero_grad(set_to_none=True)
            head(model()).backward(gradient=unit)
            opt_g.step()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                gstep()
        torch.cuda.current_stream().wait_stream(side)
        hg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(hg):
            gstep()
        ms_graph = round(timed(hg.replay, iters=10, warm=3), 3)
    except Exception as exc:  # noqa: BLE001
        ms_graph = f"capture failed: {type(exc).__name__}: {exc}"[:200]
    M = 2 * E + N
    B = (decomp or {}).get("num_bases")
    # featureless first layer: one weight-table row (basis: the node's B x d block) per message + index, one output row per node
    row = (B or 1) * nhid * 4
    fwd = M * (row + 8) + N * nhid * 4
    bwd_basis = M * (nhid * 4 + 8) + 2 * N * row
    alg = {"featureless_fwd": fwd, "fbasis_fwd": fwd, "fbasis_tile_fwd": fwd, "featureless_wgrad": M * (nhid * 4 + 8) + (2 * R0 + 1) * N * nhid * 4,
           "fbasis_bwd": bwd_basis, "fbasis_tile_bwd": bwd_basis}.get(name, fwd)
    return _attach_traffic({"baseline_config": baseline_config, "workload": tag, "N": N, "R0": R0, "E": E, "params": sum(p.numel() for p in model.parameters()),
            "step": "NodeClassifier forward + cross-entropy + backward + Adam",
            # the experiments replay the captured step by default (experiments/classify_nodes.py); the eager loop's time is the host's
            "ms_per_step": ms_graph if isinstance(ms_graph, float) else round(ms, 3), "ms_per_step_eager": round(ms, 3),
            "ms_per_step_hipgraph_replay": ms_graph,
            "edges_per_s": round(E / (ms_graph if isinstance(ms_graph, float) else ms) * 1e3), "launches_per_step": _count_launches(step), "kernels_ms": allk, "library_launches_per_step": round(sum(_launch_counts(step).values()), 1),
            "roofline": _roof(name, kms, alg, "messages x (weight-table row + 8 B index) + node rows written")}, key, {name: alg} if name else None)


class _MeanSquare(torch.autograd.Function):
    """loss = mean(out^2) as a dot product and one scaled copy (bench.py's loss): out.pow(2).mean() costs five elementwise /
    reduction kernels and a device-to-device copy per step (PowBackward) -- 0.2 ms at AM size that are not the layer's"""

    @staticmethod
    def forward(ctx, out):
        ctx.save_for_backward(out)
        return torch.linalg.vector_norm(out.reshape(-1)).square() / out.numel()      # (an ATen reduction; torch.dot is a rocBLAS call)

    @staticmethod
    def backward(ctx, g):
        out, = ctx.saved_tensors
        return out * (g * (2.0 / out.numel()))


def _replay_ms(step, iters=20):
    """the same step captured once as a hipGraph and replayed (what the experiments do by default): nothing but the replay between the timing
    points -- the eager figure of a step made of 0.3 .. 0.5 ms kernels carries 0.1 .. 0.3 ms of host time that differs from box to box"""
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        return round(timed(g.replay, iters=3 if QUICK else iters, warm=3), 3)
    except Exception as exc:  # noqa: BLE001
        return f"capture failed: {type(exc).__name__}: {exc}"[:200]


def line_featured(tag, N, R0, E, d, decomposition, seed, baseline_config, key=None):
    T = _native.synthetic_triples_host(N, R0, E, seed)
    tp = torch.from_numpy(_native.add_inverse_and_self_host(T, N, R0))
    kw = dict(triples=tp, num_nodes=N, num_relations=2 * R0 + 1, in_features=d, out_features=d, decomposition=decomposition)
    l1 = RelationalGraphConvolutionNC(vertical_stacking=False, **kw).to(DEV)
    l2 = RelationalGraphConvolutionNC(vertical_stacking=True, **kw).to(DEV)
    X = torch.randn(N, d, device=DEV, requires_grad=True)

    def step():
        for p in [X] + list(l1.parameters()) + list(l2.parameters()):
            p.grad = None
        _MeanSquare.apply(l2(l1.forward_activated(X, "relu", private=True))).backward()
    ms = timed(step, iters=10, warm=3)
    name, kms, per_step, allk = _dominant(step)
    M = 2 * E + N
    # SURVEY 8(d), literally: forward of one layer M (4 d + 8) + N 4 d; backward M (4 d + 8) + 2 N 4 d (X needs a gradient).  The
    # forward / backward of a layer may be ONE kernel (tile path) or TWO (sparse buckets: transform + per-destination sum): the
    # fraction is taken over the SUM of the kernels that make up the pass, never over one of them with its own byte model.
    fwd_alg, bwd_alg = M * (4 * d + 8) + N * 4 * d, M * (4 * d + 8) + 2 * N * 4 * d
    fwd_kernels = [k for k in ("spmm", "block_spmm", "spmm_scatter") if k in allk]
    if "spmm_scatter" in allk and "segment_sum" in allk:
        fwd_kernels.append("segment_sum")
    bwd_kernels = [k for k in ("bwd_fused", "bwd_scatter_dw", "wgrad", "block_wgrad") if k in allk]
    if "bwd_scatter_dw" in allk and "segment_sum" in allk and "segment_sum" not in fwd_kernels:
        bwd_kernels.append("segment_sum")
    # per-kernel launch counts per step tell a shared kernel name's passes apart (segment_sum serves both directions on the two-pass path)
    counts = _launch_counts(step)
    def pass_ms(kernels, direction):
        tot = 0.0
        for k in kernels:
            per_layer = counts.get(k, 0) / 2.0                 # launches of this kernel per layer and step (both directions together)
            share = 1.0 if k != "segment_sum" else (0.5 if ("spmm_scatter" in allk and "bwd_scatter_dw" in allk) else 1.0)
            tot += allk[k] * per_layer * share
        return tot
    fwd_ms, bwd_ms = pass_ms(fwd_kernels, "fwd"), pass_ms(bwd_kernels, "bwd")

    def blk(kernels, t, alg, model):
        if not kernels or not t:
            return None
        ach = alg / (t * 1e-3) / 1e9
        return {"kernels": " + ".join(kernels), "ms_per_layer": round(t, 4), "algorithmic_bytes_per_layer": int(alg), "achieved": round(ach, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "bound": "hbm", "bytes_model": model}
    fwd = blk(fwd_kernels, fwd_ms, fwd_alg, "SURVEY 8(d) forward, one layer: M (4 d_in + 8) + N 4 d_out")
    bwd = blk(bwd_kernels, bwd_ms, bwd_alg, "SURVEY 8(d) backward, one layer: M (4 d_out + 8) + 2 N 4 d_in")
    dom = bwd if (bwd and (not fwd or bwd_ms >= fwd_ms)) else fwd
    roof = dict(dom) if dom else None
    if roof:
        roof["kernel"] = roof.pop("kernels")
        roof["dominant_tag"] = name
        roof["avg_launch_ms"] = roof["ms_per_layer"]
        roof["forward"], roof["backward"] = fwd, bwd
        step_alg = 2 * (fwd_alg + bwd_alg)
        roof["step_frac"] = round(step_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        # the dominant kernel with its OWN access pattern's bytes, as a separately named figure (not the roofline fraction)
        own = {"spmm": M * (4 * d + 8) + N * 4 * d, "spmm_scatter": M * (4 * d + 8) + M * 4 * d, "segment_sum": M * (4 * d + 4) + N * 4 * d,
               "bwd_fused": bwd_alg, "wgrad": M * (4 * d + 8) + N * 4 * d, "bwd_scatter_dw": M * (2 * 4 * d + 12) + M * 4 * d,
               "block_spmm": M * (4 * d + 12) + N * 4 * d, "block_wgrad": M * (2 * 4 * d + 12)}.get(name)
        if own and kms:
            roof["dominant_kernel_own_access_pattern"] = {"kernel": name, "avg_launch_ms": round(kms, 4), "bytes_per_launch": int(own),
                                                          "GBs": round(own / (kms * 1e-3) / 1e9, 1),
                                                          "note": "what this kernel itself moves (two gathered rows / the transformed rows once more), NOT SURVEY 8(d)"}
    models = {"spmm": fwd_alg, "block_spmm": fwd_alg, "spmm_csr": fwd_alg, "bwd_fused": bwd_alg}
    return _attach_traffic({"baseline_config": baseline_config, "workload": tag, "N": N, "R0": R0, "E": E, "step": "2 featured layers, forward + backward",
                            "ms_per_step": round(ms, 3), "ms_per_step_hipgraph_replay": _replay_ms(step), "edges_per_s": round(E / ms * 1e3),
                            "kernels_ms": allk, "roofline": roof}, key, models)


def line_wn18(baseline_config):
    N, R0, d, E, Tn = 40_943, 18, 200, 15_000, 330_000
    ed = {"general": 0.5, "self_loop": 0.2, "self_loop_type": "schlichtkrull-dropout"}
    layer = RelationalGraphConvolutionLP(num_nodes=N, num_relations=2 * R0 + 1, in_features=d, out_features=d, edge_dropout=ed,
                                         decomposition={"type": "basis", "num_bases": 2}, w_init="glorot-normal",
                                         b_init="zeros").to(DEV)
    dm = DistMult(R0, d, N, R0).to(DEV)
    emb = torch.randn(N, d, device=DEV, requires_grad=True)
    graph = torch.from_numpy(_native.synthetic_triples_host(N, R0, E, 3)).to(DEV)
    batch = torch.from_numpy(_native.synthetic_triples_host(N, R0, Tn, 4)).to(DEV)
    y = torch.rand(Tn, device=DEV).round()

    from torch_rgcn.functional import bce_with_logits, unit_gradient
    unit = unit_gradient(DEV)

    def step():      # (the loss as experiments/predict_links.py computes it: one launch for the BCE and its gradient)
        for p in [emb] + list(layer.parameters()) + list(dm.parameters()):
            p.grad = None
        x = layer(graph, torch.relu(emb))
        bce_with_logits(dm(batch, x), y).backward(gradient=unit)
    ms = timed(step, iters=10, warm=3)
    name, kms, per_step, allk = _dominant(step)
    alg = {"distmult_bwd": Tn * (3 * d * 4 + 28) + Tn * 2 * d * 4, "distmult_fwd": Tn * (3 * d * 4 + 28)}.get(name, Tn * 3 * d * 4)
    # the same step without any host synchronisation (plans sized by upper bounds, deferred range checks), and replayed
    # from a hipGraph (nothing but the replay between the timing points)
    ms_nosync = ms_graph = None
    from torch_rgcn import routes
    try:
        with routes.override(deferred_checks="1"):
            ms_nosync = timed(step, iters=10, warm=3)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            hg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(hg):
                step()
        ms_graph = timed(hg.replay, iters=10, warm=3)
    except Exception as exc:  # noqa: BLE001
        ms_graph = f"failed: {type(exc).__name__}: {exc}"[:200]
    roof = _roof(name, kms, alg, "scored triples x (three d-wide rows + 24 B of indices [+ two gradient rows])")
    if name == "gemm":      # the dense (B d) x d contraction and its two backward products: bound by the matrix cores
        flops = 2.0 * N * 2 * d * d
        roof = {"kernel": "gemm_panel_kernel (ag @ flat(bases); d_ag = g @ flat^T; dbases = ag^T @ g)", "bound": "mfma", "avg_launch_ms": round(kms, 4),
                "algorithmic_flops_per_launch": flops, "achieved": round(flops / (kms * 1e-3) / 1e12, 1), "peak": 157.3, "unit": "TFLOP/s",
                "frac": round(flops / (kms * 1e-3) / 1e12 / 157.3, 4)}
    # the largest single kernel of the step after the GEMMs: all DistMult gradients from one pass (rgcn_distmult_bwd_all_f32)
    if roof is not None and "distmult_bwd_all" in allk:
        dm_alg = Tn * (3 * d * 4 + 28) + (N + R0) * d * 4       # the three embedding rows + indices + upstream scalar per scored triple; every
        roof["distmult_bwd_all"] = _roof("distmult_bwd_all_kernel", allk["distmult_bwd_all"], dm_alg,       # entity / relation gradient row written once
                                         "scored triples x (three d-wide rows + 28 B) + (entities + relations) x one gradient row")
        roof["distmult_fwd"] = _roof("distmult_fwd_kernel", allk.get("distmult_fwd"), Tn * (3 * d * 4 + 28), "scored triples x (three d-wide rows + 28 B)")
    if roof is not None:
        roof["dominant_tag"] = name
    return _attach_traffic({"baseline_config": baseline_config, "workload": "WN18-shaped: LP layer d=200 basis 2 (graph of 15,000 triples built per step) + "
            "DistMult on 330,000 triples", "N": N, "R0": R0, "graph_triples": E, "scored_triples": Tn,
            "step": "encoder + decoder forward + BCE + backward (per-step graph build included)", "ms_per_step": round(ms, 3),
            "ms_per_step_sync_free": None if ms_nosync is None else round(ms_nosync, 3),
            "ms_per_step_hipgraph_replay": round(ms_graph, 3) if isinstance(ms_graph, float) else ms_graph,
            "scored_triples_per_s": round(Tn / ms * 1e3), "kernels_ms": allk, "launches_per_step": _count_launches(step),
            "roofline": roof}, "wn18", {"distmult_fwd": Tn * (3 * d * 4 + 28), "distmult_bwd_all": Tn * (3 * d * 4 + 28) + (N + R0) * d * 4})


def line_s2(baseline_config):
    """the S1 graph with a FEATURELESS first layer (faithful to NodeClassifier), basis B = 2 -- the weight-table gather
    path; layer 2 is S1's (16 -> 16, vertical flag).  Roofline on a stated bytes model for the whole step and for the dominant kernel."""
    N, R0, E, d, B = 1_000_000, 50, 10_000_000, 16, 2
    T = _native.synthetic_triples_host(N, R0, E, 0)
    tp = torch.from_numpy(_native.add_inverse_and_self_host(T, N, R0))
    kw = dict(triples=tp, num_nodes=N, num_relations=2 * R0 + 1, out_features=d)
    l1 = RelationalGraphConvolutionNC(in_features=None, vertical_stacking=False, decomposition={"type": "basis", "num_bases": B}, **kw).to(DEV)
    l2 = RelationalGraphConvolutionNC(in_features=d, vertical_stacking=True, **kw).to(DEV)

    def step():
        for p in list(l1.parameters()) + list(l2.parameters()):
            p.grad = None
        _MeanSquare.apply(l2(l1.forward_activated(None, "relu", private=True))).backward()
    ms = timed(step, iters=10, warm=3)
    name, kms, per_step, allk = _dominant(step)
    counts = _launch_counts(step)
    M = 2 * E + N
    row = B * d * 4
    # layer 1 forward: one B x d block of the bases table + 8 B of index per message (the block belongs to the message's SOURCE node: a
    # destination-major walk reads it per message), one output row per node; backward: one upstream row + 8 B per message, every node's
    # block read once and its gradient written once (the source-major walk); layer 2: SURVEY 8(d) S1 bytes
    l1_fwd, l1_bwd = M * (row + 8) + N * d * 4, M * (4 * d + 8) + 2 * N * row
    l2_fwd, l2_bwd = M * (4 * d + 8) + N * 4 * d, M * (4 * d + 8) + 2 * N * 4 * d
    step_alg = l1_fwd + l1_bwd + l2_fwd + l2_bwd
    own = {"fbasis_fwd": l1_fwd, "basis_aggregate": l1_fwd, "fbasis_bwd": l1_bwd, "fbasis_small_bwd": l1_bwd, "spmm": l2_fwd, "bwd_fused": l2_bwd}.get(name)
    roof = _roof(name, kms, own, "layer 1 forward: messages x (B x d table block + 8 B) + node rows; backward: messages x (upstream row + 8 B) + 2 x table; layer 2: SURVEY 8(d)") if own else None
    if roof:
        roof["step_algorithmic_bytes"] = int(step_alg)
        roof["step_frac"] = round(step_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    if roof:
        roof["dominant_tag"] = name
    return _attach_traffic({"baseline_config": baseline_config, "workload": "S2: S1 graph, featureless layer 1 with basis B=2 (no R x N x 16 table), ReLU, layer 2 16->16",
                            "N": N, "R0": R0, "E": E, "step": "2 layers, forward + backward", "ms_per_step": round(ms, 3),
                            "ms_per_step_hipgraph_replay": _replay_ms(step), "edges_per_s": round(E / ms * 1e3),
                            "kernels_ms": allk, "launches_per_step": counts, "roofline": roof}, "s2",
                           {"basis_aggregate": l1_fwd, "fbasis_small_bwd": l1_bwd, "spmm": l2_fwd, "bwd_fused": l2_bwd})


def line_am_shipped(baseline_config):
    """AM as the reference ships it (configs/rgcn/nc-AM.yaml: featureless layer 1 with basis 40, hidden 10, 11 classes, Adam): the
    forward + backward and the optimiser step are timed SEPARATELY -- Adam over 667 M parameters is ~2.3 ms of pure streaming --
    and the roofline is taken for forward + backward on a stated bytes model"""
    N, R0, E, nhid, ncls, B, labelled = 1_666_764, 133, 5_988_321, 10, 11, 40, 802
    T = _native.synthetic_triples_host(N, R0, E, 1)
    model = NodeClassifier(triples=T, nnodes=N, nrel=R0, nhid=nhid, nclass=ncls, decomposition={"type": "basis", "num_bases": B}).to(DEV)
    idx = torch.arange(labelled, device=DEV)
    y = torch.randint(0, ncls, (labelled,), device=DEV)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, fused=True)

    from torch_rgcn.functional import MaskedCrossEntropy, unit_gradient
    head = MaskedCrossEntropy(idx, y, N)        # what experiments/classify_nodes.py uses (one launch for loss + gradient)
    unit = unit_gradient(DEV)

    def fwd_bwd():
        opt.zero_grad(set_to_none=True)
        head(model()).backward(gradient=unit)

    def step():
        fwd_bwd()
        opt.step()
    ms_step = timed(step, iters=5, warm=2)
    ms_fb = timed(fwd_bwd, iters=5, warm=2)
    fwd_bwd()
    ms_adam = timed(opt.step, iters=5, warm=1)
    name, kms, per_step, allk = _dominant(fwd_bwd)
    M = 2 * E + N
    n_par = sum(p.numel() for p in model.parameters())
    # forward + backward of the step: the bases table (B x N x nhid floats) is READ once by the forward gather and its gradient WRITTEN
    # once by the backward -- per message only the 8 B of index are inherent (a node's B x nhid block is shared by all its messages);
    # layer 2 (10 -> 11, padded to 16): SURVEY 8(d).  Adam: 4 reads + 3 writes of 4 B per parameter.
    table = B * N * nhid * 4
    l2 = 2 * (M * (4 * 16 + 8) + N * 4 * 16) + N * 4 * 16
    fb_alg = 2 * table + 2 * M * 8 + 2 * N * nhid * 4 + l2
    adam_alg = 7 * 4 * n_par
    fb = {"ms": round(ms_fb, 3), "algorithmic_bytes": int(fb_alg), "achieved": round(fb_alg / (ms_fb * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
          "frac": round(fb_alg / (ms_fb * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "bound": "hbm",
          "bytes_model": "bases table read ONCE (by the forward) + its gradient written once + 2 x 8 B of index per message + layer 2 per SURVEY 8(d); "
                         "the backward's own read of the table (the comps gradient needs it) is NOT counted: with it the figure is frac_with_second_table_read",
          "frac_with_second_table_read": round((fb_alg + table) / (ms_fb * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    adam = {"ms": round(ms_adam, 3), "algorithmic_bytes": int(adam_alg), "achieved": round(adam_alg / (ms_adam * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(adam_alg / (ms_adam * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "bound": "hbm",
            "bytes_model": "torch.optim.Adam(fused=True): param, grad, exp_avg, exp_avg_sq read; param, exp_avg, exp_avg_sq written"}
    roof = dict(fb)
    roof["kernel"] = f"forward + backward (dominant kernel: {name}, {round(kms, 4) if kms else None} ms per launch)"
    roof["dominant_tag"] = name
    roof["avg_launch_ms"] = fb["ms"]
    roof["adam"] = adam
    # per-kernel models for the wasted-traffic ratios (traffic / model): the tile kernels' own access patterns
    ys = 16
    models = {"fbasis_tile_fwd": table + M * 8 + M * ys * 4, "gather_rows_sum4": M * (ys * 4 + 4) + N * 16 * 4,
              "fbasis_tile_bwd": 2 * table + M * (12 + 64), "spmm_scatter": M * (4 * 16 + 8) + M * 64, "segment_sum": M * (64 + 4) + N * 64,
              "bwd_scatter_dw": M * (2 * 64 + 12) + M * 64}
    return _attach_traffic({"baseline_config": baseline_config, "workload": "AM-shaped NodeClassifier as shipped (featureless L1, basis 40, hidden 10, 11 classes)",
                            "N": N, "R0": R0, "E": E, "params": n_par, "step": "NodeClassifier forward + cross-entropy + backward + Adam",
                            "ms_per_step": round(ms_step, 3), "ms_forward_backward": round(ms_fb, 3), "ms_adam": round(ms_adam, 3),
                            "edges_per_s": round(E / ms_step * 1e3), "kernels_ms": allk, "roofline": roof}, "amshipped", models)


def line_eval(cfg):
    """the filtered ranking evaluator at WN18 size (reference utils/misc.py:60-110: 10,000 head / tail queries against 40,943
    candidates each): whole evaluate() wall time, the score-all kernel against the fp32 MFMA peak, the oracle on a small sample"""
    import eval_bench
    r = eval_bench.run(test=5000, cpu_sample=8 if QUICK else 16)
    return {"baseline_config": cfg, "workload": r["workload"], "step": "one filtered ranking evaluation (encode once, score all candidates, rank)",
            "ms_per_step": round(1e3 * r["evaluate_wall_s"], 2), "queries_per_s": r["queries_per_s"], "mrr": r["mrr"],
            "kernels_ms": r["kernels_ms_in_evaluate"], "roofline": r["roofline"], "cpu_baseline": r.get("cpu_baseline")}


def secondary_lines():
    """one dict per BASELINE.json config 1-4 (dataset-shaped synthetic graphs), each with the dominant kernel's
    roofline; bounded to a few seconds each"""
    out = []
    for fn in (lambda: line_node_classifier("AIFB-shaped NodeClassifier (featureless L1, hidden 16, 4 classes)", 8285, 45, 29043, 16, 4, None, 176,
                                            "configs[0] AIFB (reference config nc-AIFB.yaml; here on the GPU, the CPU run is cpu_baseline's business)", key="aifb"),
               lambda: line_node_classifier("MUTAG-shaped NodeClassifier (basis 30, hidden 16, 2 classes)", 23644, 23, 74227, 16, 2,
                                            {"type": "basis", "num_bases": 30}, 340, "configs[1] MUTAG, basis decomposition", key="mutag"),
               lambda: line_featured("AM-shaped, block-diagonal (nb=4), 2 featured layers d=16 (layer-level, SURVEY 8d)", 1_666_764, 133,
                                     5_988_321, 16, {"type": "block", "num_blocks": 4}, 2, "configs[2] AM, block-diagonal", key="am"),
               lambda: line_wn18("configs[3] WN18 link prediction, DistMult decoder"),
               lambda: line_featured("S1(ii): S1 graph, basis decomposition B=10, 2 featured layers d=16", 1_000_000, 50, 10_000_000, 16,
                                     {"type": "basis", "num_bases": 10}, 0, "SURVEY 8(d) S1 variant (ii)", key="s1ii"),
               lambda: line_featured("S1(iii): S1 graph, block-diagonal nb=4, 2 featured layers d=16", 1_000_000, 50, 10_000_000, 16,
                                     {"type": "block", "num_blocks": 4}, 0, "SURVEY 8(d) S1 variant (iii)", key="s1iii"),
               lambda: line_s2("SURVEY 8(d) S2 (secondary)"),
               lambda: line_am_shipped("configs[2] AM as the reference ships it (nc-AM.yaml: featureless, basis 40, hidden 10)"),
               lambda: line_eval("configs[3] WN18 ranking evaluator (SURVEY 8 f-1)")):
        try:
            out.append(fn())
        except Exception as exc:  # noqa: BLE001
            out.append({"failed": f"{type(exc).__name__}: {exc}"[:300]})
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--lines", default="", help="bench.py's secondary lines by name: s2,amshipped,s1ii,s1iii (JSON, one per line)")
    ap.add_argument("--quick", action="store_true", help="few iterations per measurement (profiling runs)")
    a = ap.parse_args()
    QUICK = a.quick
    if a.lines:
        for nm in a.lines.split(","):
            fn = {"s2": lambda: line_s2("S2"), "amshipped": lambda: line_am_shipped("AM shipped"),
                  "aifb": lambda: line_node_classifier("AIFB-shaped", 8285, 45, 29043, 16, 4, None, 176, "AIFB", key="aifb"),
                  "mutag": lambda: line_node_classifier("MUTAG-shaped", 23644, 23, 74227, 16, 2, {"type": "basis", "num_bases": 30}, 340, "MUTAG", key="mutag"),
                  "wn18": lambda: line_wn18("WN18"), "eval": lambda: line_eval("WN18 ranking evaluator"), "am": lambda: line_featured("AM block", 1_666_764, 133, 5_988_321, 16, {"type": "block", "num_blocks": 4}, 2, "AM", key="am"),
                  "s1ii": lambda: line_featured("S1(ii)", 1_000_000, 50, 10_000_000, 16, {"type": "basis", "num_bases": 10}, 0, "S1(ii)", key="s1ii"),
                  "s1iii": lambda: line_featured("S1(iii)", 1_000_000, 50, 10_000_000, 16, {"type": "block", "num_blocks": 4}, 0, "S1(iii)", key="s1iii")}[nm]
            print(json.dumps(fn()), flush=True)
            torch.cuda.empty_cache()
        sys.exit(0)
    todo = a.only.split(",") if a.only else ["aifb", "mutag", "am", "wn18", "s2", "s1"]
    if "s2" in todo:
        s2_featureless_basis()
    if "aifb" in todo:
        nc_model("AIFB-shaped NodeClassifier (featureless L1, hidden 16, 4 classes)", 8285, 45, 29043, 16, 4, None, 176)
    if "mutag" in todo:
        nc_model("MUTAG-shaped NodeClassifier (basis 30, hidden 16, 2 classes)", 23644, 23, 74227, 16, 2,
                 {"type": "basis", "num_bases": 30}, 340)
    if "am" in todo:
        featured_layers("AM-shaped, block-diagonal (nb=4), 2 featured layers d=16", 1_666_764, 133, 5_988_321, 16,
                        {"type": "block", "num_blocks": 4}, 2)
    if "s1" in todo:   # SURVEY.md 8(d) S1 variants (i)-(iii); (i) is what bench.py times
        featured_layers("S1(i): no decomposition, W 101x16x16", 1_000_000, 50, 10_000_000, 16, None, 0)
        featured_layers("S1(ii): basis B=10", 1_000_000, 50, 10_000_000, 16, {"type": "basis", "num_bases": 10}, 0)
        featured_layers("S1(iii): block nb=4", 1_000_000, 50, 10_000_000, 16, {"type": "block", "num_blocks": 4}, 0)
    if "amreal" in todo:
        nc_model("AM-shaped NodeClassifier as shipped (featureless L1, basis 40, hidden 10, 11 classes)", 1666764, 133,
                 5988321, 10, 11, {"type": "basis", "num_bases": 40}, 802)
    if "wn18" in todo:
        wn18_lp()
