#!/usr/bin/env python3
"""Step B probe 2: "soft windows" at no padding cost.  Inside a (tile, relation) bucket the slots are sorted by SOURCE row (the block-tile
kernels add with ds_add_f64 and carry every slot's tile row in the chunk record: the slot order inside a bucket is free), so a chunk's 16
sources span 1 / (chunks per bucket) of the table; the tile's chunks are then ordered by their first source.  Every workgroup sweeps the
source table once per tile, all workgroups roughly together -- the temporal locality tools/micro/gather_window.hip measures -- with the
(tile, relation) padding of the plain plan.  Plans are made here with torch ops (a probe), records by rgcn_bwd_blk_prepare_f32.
    --order dst   the shipped order (relation-major, destination-sorted buckets) through the same code path (control)
    --order src   buckets sorted by source, chunks in relation-major order (control: sorting alone)
    --order win   buckets sorted by source, chunks ordered by first source (soft windows)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-rgcn_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from torch_rgcn import _native  # noqa: E402
from torch_rgcn._native import F_RELU, _check, _dp, _on, _stream, c_i32, c_i64, lib  # noqa: E402
from torch_rgcn.graph import graph_from_nc_triples  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=1_000_000)
ap.add_argument("--edges", type=int, default=10_000_000)
ap.add_argument("--rels", type=int, default=50)
ap.add_argument("--fwd-rows", default="977,489")
ap.add_argument("--bwd-rows", default="218")
ap.add_argument("--orders", default="dst,src,win")
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
N, R0, E = a.nodes, a.rels, a.edges
R = 2 * R0 + 1
T = _native.synthetic_triples_host(N, R0, E, 0)
tp = _native.add_inverse_and_self_host(T, N, R0)
g = graph_from_nc_triples(tp, N, R, False, dev)
M = tp.shape[0]
torch.manual_seed(0)
X = torch.randn(N, 16, device=dev)
G = torch.randn(N, 16, device=dev)
W = torch.randn(R, 16, 16, device=dev) * 0.1
b = torch.randn(16, device=dev)


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return float(np.median(ts)), float(np.min(ts))


def make_plan(dst, src, rel, val, rows, order):
    """-> (src[m_pad], dst[m_pad] (-1: pad), val[m_pad], chunk_rel[n_chunks], run_ptr[n_tiles (R + 1)], n_tiles)"""
    n_tiles = -(-N // rows)
    dst, src, rel = dst.long(), src.long(), rel.long()
    bucket = torch.div(dst, rows, rounding_mode="floor") * R + rel
    inner = dst if order == "dst" else src
    perm = torch.argsort(bucket * N + inner)
    bs = bucket[perm]
    nbk = n_tiles * R
    cnt = torch.bincount(bs, minlength=nbk)
    padded = (cnt + 15) // 16 * 16
    base = torch.cumsum(padded, 0) - padded
    first = torch.cumsum(cnt, 0) - cnt
    slot = base[bs] + (torch.arange(M, device=dev) - first[bs])
    m_pad = int(padded.sum().item())
    n_chunks = m_pad // 16
    S = torch.zeros(m_pad, dtype=torch.int32, device=dev)
    D = torch.full((m_pad,), -1, dtype=torch.int32, device=dev)
    V = torch.zeros(m_pad, dtype=torch.float32, device=dev)
    S[slot], D[slot], V[slot] = src[perm].int(), dst[perm].int(), val[perm]
    cb = torch.repeat_interleave(torch.arange(nbk, device=dev), padded // 16)          # bucket of every chunk
    crel = (cb % R).int()
    ctile = torch.div(cb, R, rounding_mode="floor")
    if order == "win":
        cperm = torch.argsort(ctile * N + S[::16].long())                                  # inside a tile: by the chunk's first source
        idx = (cperm[:, None] * 16 + torch.arange(16, device=dev)[None, :]).reshape(-1)
        S, D, V, crel = S[idx].contiguous(), D[idx].contiguous(), V[idx].contiguous(), crel[cperm].contiguous()
    tcnt = torch.bincount(ctile, minlength=n_tiles)
    tend = torch.cumsum(tcnt, 0)
    run_ptr = torch.zeros(n_tiles * (R + 1), dtype=torch.int32, device=dev)
    run_ptr[0::R + 1] = (tend - tcnt).int()
    run_ptr[R::R + 1] = tend.int()
    return S, D, V, crel, run_ptr, n_tiles, m_pad, n_chunks


def records(S, D, V, crel, rows, n_chunks):
    rec = torch.empty(int(lib().rgcn_bwd_blk_rec_bytes(c_i64(n_chunks))) + 16, device=dev, dtype=torch.uint8)
    with _on(dev):
        _check(lib().rgcn_bwd_blk_prepare_f32(None, _dp(S), _dp(D), _dp(V), c_i32(rows), _dp(crel), c_i64(n_chunks), _dp(rec), _stream(dev)), "prep")
    return rec


s, p, o, val, alive = g._dev
assert alive is None or bool((alive != 0).all())
ref = _native.spmm(X, W, b, g.fwd_plan(16))
med, mn = timeit(lambda: _native.spmm(X, W, b, g.fwd_plan(16)), a.iters)
print(f"forward, shipped spmm_d16 (wave-owned tiles): med {med:.3f} min {mn:.3f} ms", flush=True)
Wp = _native.pack_w16(W)
for rows in [int(v) for v in a.fwd_rows.split(",") if v]:
    for order in a.orders.split(","):
        S, D, V, crel, run_ptr, n_tiles, m_pad, n_chunks = make_plan(s, o, p, val, rows, order)
        rec = records(S, D, V, crel, rows, n_chunks)
        out = torch.empty(N, 16, device=dev)

        def run():
            with _on(dev):
                _check(lib().rgcn_spmm_blk_f32(_dp(X), _dp(Wp), _dp(b), _dp(out), _dp(rec), _dp(run_ptr), c_i64(n_tiles), c_i32(rows), c_i64(N),
                                               c_i32(R), c_i32(0), None, c_i64(0), c_i64(0), _stream(dev)), "spmm_blk")
        med, mn = timeit(run, a.iters)
        err = float((out - ref).abs().max() / ref.abs().max())
        print(f"forward spmm_blk rows={rows} order={order}: pad {m_pad / M:.3f} chunks {n_chunks} med {med:.3f} min {mn:.3f} ms err {err:.1e}", flush=True)
        del S, D, V, crel, rec
        torch.cuda.empty_cache()

bp = g.bwd_blk_plan()
dX0, dW0 = _native.bwd_fused(G, X, W, bp, atomic=True)
med, mn = timeit(lambda: _native.bwd_fused(G, X, W, bp, atomic=True), a.iters)
print(f"backward, shipped bwd_blk rows={bp.tile_rows}: med {med:.3f} min {mn:.3f} ms (incl. the dW fill)", flush=True)
Wtp = _native.pack_w16t(W)
for rows in [int(v) for v in a.bwd_rows.split(",") if v]:
    for order in a.orders.split(","):
        S, D, V, crel, run_ptr, n_tiles, m_pad, n_chunks = make_plan(o, s, p, val, rows, order)
        rec = records(S, D, V, crel, rows, n_chunks)
        dX = torch.empty(N, 16, device=dev)
        dW = torch.empty_like(W)

        def run():
            with _on(dev):
                _check(lib().rgcn_bwd_blk_f32(_dp(G), _dp(X), _dp(Wtp), _dp(dX), _dp(dW), _dp(rec), _dp(run_ptr), c_i64(n_tiles), c_i32(rows), c_i64(N),
                                              c_i32(R), c_i32(0), None, c_i64(N), None, c_i64(0), c_i64(0), _stream(dev)), "bwd_blk")
        med, mn = timeit(run, a.iters)
        e1 = float((dX - dX0).abs().max() / dX0.abs().max())
        e2 = float((dW - dW0).abs().max() / dW0.abs().max())
        print(f"backward bwd_blk rows={rows} order={order}: pad {m_pad / M:.3f} chunks {n_chunks} med {med:.3f} min {mn:.3f} ms err dX {e1:.1e} dW {e2:.1e}", flush=True)
        del S, D, V, crel, rec
        torch.cuda.empty_cache()
