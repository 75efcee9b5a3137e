"""The soft-window plans as the layers get them on the GPU -- built through the C ABI (rgcn_softwin_order / rgcn_softwin_fill: rocPRIM radix sorts
+ small kernels) -- satisfy the invariants the torch-op builder is checked for on the CPU, and have the same shape as its plans."""
import numpy as np
import pytest
import torch

from test_softwin_plan import check_owner_plan, check_plan, random_messages

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("N,R,M,rows,masked", [(1000, 7, 20000, 128, False), (1000, 7, 20000, 128, True), (37, 3, 11, 16, False),
                                                (500, 1, 3000, 977, False), (64, 5, 0, 16, False), (70_000, 11, 400_000, 274, True),
                                                (6400, 100, 10_000, 64, False)])        # (as many chunks as messages: one nearly empty bucket each)
def test_device_built_plan_invariants(N, R, M, rows, masked):
    from torch_rgcn import _native, routes
    dst, src, rel, val, alive = random_messages(N, R, M, masked, N + M)
    g = [t if t is None else t.to(DEV) for t in (dst, src, rel, val, alive)]
    p = _native.build_softwin_plan(*g, N, N, R, rows)
    assert p.src.is_cuda
    check_plan(p, dst, src, rel, val, alive, N, R, rows)
    with routes.override(softwin_build="torch"):
        q = _native.build_softwin_plan(*g, N, N, R, rows)
    assert (q.m_pad, q.n_chunks, q.n_messages, q.max_run_chunks) == (p.m_pad, p.n_chunks, p.n_messages, p.max_run_chunks)
    assert torch.equal(q.tile_ptr.cpu(), p.tile_ptr.cpu()) and torch.equal(q.run_ptr.cpu(), p.run_ptr.cpu())
    assert torch.equal(q.chunk_rel.cpu(), p.chunk_rel.cpu())                      # the chunk order (ties: same first source AND relation only)
    assert torch.equal(q.src.cpu()[:p.m_pad], p.src.cpu()[:p.m_pad])              # sources: the order among equal keys does not show


@pytest.mark.parametrize("masked,R,N,M,rows", [(False, 20, 900, 30000, 128), (True, 20, 900, 30000, 128), (False, 2, 900, 30000, 128), (True, 1, 900, 30000, 128),
                                               (False, 101, 60_000, 900_000, 235)])
def test_device_built_owner_plan_invariants(masked, R, N, M, rows):
    from torch_rgcn import _native
    NW, K = 12, 9
    dst, src, rel, val, alive = random_messages(N, R, M, masked, 7)
    g = [t if t is None else t.to(DEV) for t in (dst, src, rel, val, alive)]
    p = _native.build_softwin_plan(*g, N, N, R, rows, own_waves=NW, own_per_wave=K)
    assert p.src.is_cuda and p.own_waves == NW
    check_owner_plan(p, dst, src, rel, val, alive, N, R, rows, NW, K)
