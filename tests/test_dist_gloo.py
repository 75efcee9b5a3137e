"""Relation-sharded path on CPU with gloo, world_size 2: the collectives and the partitioning are
the product's (torch_rgcn.dist); the per-rank compute is the oracle (injected as local_fn), since
the HIP kernels cannot run here.  Checks that the all-reduced sum of the two ranks' partial
outputs / feature gradients equals the unsharded oracle, and that every relation has one owner."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT
from oracle import oracle


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _OracleMP(torch.autograd.Function):
    """val-weighted message passing on an explicit edge list, computed by the CPU oracle"""

    @staticmethod
    def forward(ctx, X, W, tp, val, N, R):
        ctx.args = (tp, val, N, R)
        ctx.save_for_backward(X, W)
        return torch.from_numpy(oracle.rgcn_forward(tp, val, N, R, X.detach().numpy(), W.detach().numpy()))

    @staticmethod
    def backward(ctx, g):
        X, W = ctx.saved_tensors
        tp, val, N, R = ctx.args
        dX, dW, _ = oracle.rgcn_backward(tp, val, N, R, X.detach().numpy(), W.detach().numpy(), g.numpy())
        return torch.from_numpy(dX), torch.from_numpy(dW), None, None, None, None


def _worker(rank, world, port, q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch_rgcn.dist import partition_relations, sharded_apply
    N, R0, E, d = 300, 6, 4000, 8
    R = 2 * R0 + 1
    tp = oracle.add_inverse_and_self(oracle.synthetic_triples(N, R0, E, 7), N, R0)
    val = oracle.nc_edge_norm(tp, N, R, False)           # normalise on the FULL graph, then shard
    owner = partition_relations(np.bincount(tp[:, 1], minlength=R), world)
    mine = owner[tp[:, 1]] == rank
    g = torch.Generator().manual_seed(0)
    X = torch.randn(N, d, generator=g).requires_grad_(True)  # replicated
    W = torch.randn(R, d, d, generator=g).requires_grad_(True)
    gout = torch.randn(N, d, generator=g)
    out = sharded_apply(lambda x: _OracleMP.apply(x, W, tp[mine], val[mine], N, R), X, dist.group.WORLD)
    out.backward(gout)
    # relation weights: gradient lives on the owner only; sum over ranks reassembles dW
    dW = W.grad.clone()
    dist.all_reduce(dW)
    if rank == 0:
        q.put((out.detach().numpy(), X.grad.numpy(), dW.numpy(), owner))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_relation_sharding_matches_unsharded():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out, dX, dW, owner = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    N, R0, E, d = 300, 6, 4000, 8
    R = 2 * R0 + 1
    tp = oracle.add_inverse_and_self(oracle.synthetic_triples(N, R0, E, 7), N, R0)
    val = oracle.nc_edge_norm(tp, N, R, False)
    g = torch.Generator().manual_seed(0)
    X = torch.randn(N, d, generator=g).numpy()
    W = torch.randn(R, d, d, generator=g).numpy()
    gout = torch.randn(N, d, generator=g).numpy()
    ref = oracle.rgcn_forward(tp, val, N, R, X, W)
    rdX, rdW, _ = oracle.rgcn_backward(tp, val, N, R, X, W, gout)
    assert np.abs(out - ref).max() < 1e-5 * np.abs(ref).max()
    assert np.abs(dX - rdX).max() < 1e-5 * np.abs(rdX).max()
    assert np.abs(dW - rdW).max() < 1e-5 * np.abs(rdW).max()
    assert set(owner.tolist()) == {0, 1}


def test_partition_relations_lpt():
    from torch_rgcn.dist import partition_relations
    counts = [100, 90, 50, 40, 10, 10, 0]
    owner = partition_relations(counts, 2)
    loads = [sum(c for c, o in zip(counts, owner) if o == k) for k in range(2)]
    assert sorted(loads) == [150, 150]
    assert partition_relations(counts, 1).tolist() == [0] * 7
    o8 = partition_relations(np.full(101, 1000), 8)
    assert np.bincount(o8, minlength=8).max() - np.bincount(o8, minlength=8).min() <= 1


def _join_worker(rank, world, port, q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch_rgcn.functional import _join_shards
    out = {}
    for mode in ("allreduce", "rs_ag", "a2a"):
        for n in (10, 11):                                   # 11 rows: not divisible by the world size (padded blocks)
            part = torch.arange(n * 3, dtype=torch.float32).view(n, 3) * (rank + 1)
            out[(mode, n)] = _join_shards(part.clone(), dist.group.WORLD, mode).numpy()
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_join_shards_collective_variants_agree():
    """the transports of the partial-sum join (torch_rgcn.functional._join_shards): all-reduce, reduce-scatter + all-gather
    and the direct exchange (all-to-all + local sum + all-gather) give the sum over ranks (there is no "no collective" transport:
    bench.py's compute-alone leg patches the join locally)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_join_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for n in (10, 11):
        base = np.arange(n * 3, dtype=np.float32).reshape(n, 3)
        assert np.array_equal(out[("allreduce", n)], 3 * base)
        assert np.array_equal(out[("rs_ag", n)], 3 * base) and out[("rs_ag", n)].shape == (n, 3)
        assert np.array_equal(out[("a2a", n)], 3 * base) and out[("a2a", n)].shape == (n, 3)


# ----------------------------------------------------------------------------- world 8 (the shape of the driver's 8-GPU run)
def _worker8(rank, world, port, q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch_rgcn.dist import partition_relations
    from torch_rgcn.functional import _join_shards
    N, R0, E, d = 301, 2, 3000, 8             # 301 rows: not divisible by 8; R = 5 relations < 8 ranks: three ranks own nothing
    R = 2 * R0 + 1
    tp = oracle.add_inverse_and_self(oracle.synthetic_triples(N, R0, E, 11), N, R0)
    val = oracle.nc_edge_norm(tp, N, R, False)
    owner = partition_relations(np.bincount(tp[:, 1], minlength=R), world)
    mine = owner[tp[:, 1]] == rank
    g = torch.Generator().manual_seed(0)
    X = torch.randn(N, d, generator=g).numpy()
    W = torch.randn(R, d, d, generator=g).numpy()
    gout = torch.randn(N, d, generator=g).numpy()
    part = oracle.rgcn_forward(tp[mine], val[mine], N, R, X, W) if mine.any() else np.zeros((N, d), np.float32)
    dpart = oracle.rgcn_backward(tp[mine], val[mine], N, R, X, W, gout)[0] if mine.any() else np.zeros((N, d), np.float32)
    res = {}
    for mode in ("allreduce", "rs_ag", "a2a"):
        res[mode] = (_join_shards(torch.from_numpy(np.ascontiguousarray(part, np.float32)).clone(), dist.group.WORLD, mode).numpy(),
                     _join_shards(torch.from_numpy(np.ascontiguousarray(dpart, np.float32)).clone(), dist.group.WORLD, mode).numpy())
    if rank == 0:
        q.put((res, owner, int(mine.sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_sharding_every_transport_matches_unsharded():
    """world 8 on gloo: LPT packing of FEWER relations than ranks (idle ranks contribute zeros), a node count that 8 does not
    divide (padded row blocks in rs_ag / a2a), every transport of the partial-sum join against the unsharded oracle"""
    from torch_rgcn.dist import partition_relations
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res, owner, n0 = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    N, R0, E, d = 301, 2, 3000, 8
    R = 2 * R0 + 1
    tp = oracle.add_inverse_and_self(oracle.synthetic_triples(N, R0, E, 11), N, R0)
    val = oracle.nc_edge_norm(tp, N, R, False)
    g = torch.Generator().manual_seed(0)
    X = torch.randn(N, d, generator=g).numpy()
    W = torch.randn(R, d, d, generator=g).numpy()
    gout = torch.randn(N, d, generator=g).numpy()
    ref = oracle.rgcn_forward(tp, val, N, R, X, W)
    rdX = oracle.rgcn_backward(tp, val, N, R, X, W, gout)[0]
    for mode, (out, dX) in res.items():
        assert out.shape == ref.shape and dX.shape == rdX.shape, mode
        assert np.abs(out - ref).max() < 1e-5 * np.abs(ref).max(), mode
        assert np.abs(dX - rdX).max() < 1e-5 * np.abs(rdX).max(), mode
    assert len(set(owner.tolist())) == R, "five relations, five different owners"
    # S1's shape: 101 relations (100 of ~200 k messages, the self loops 1 M) on 8 ranks -- within 7 % of the mean
    counts = np.array([200_000] * 100 + [1_000_000])
    o = partition_relations(counts, 8)
    loads = np.bincount(o, weights=counts, minlength=8)
    assert loads.max() <= 1.07 * loads.mean(), loads      # (200 k granules: 2.8 M on one rank against a mean of 2.625 M is the optimum)
