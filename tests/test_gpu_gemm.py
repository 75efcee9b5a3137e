"""The hand-written MFMA contractions of the basis path (csrc/rgcn_gemm.hip) against float64 numpy: every operand layout,
ragged sizes (edges of the 128 x 128 x 16 tiles, K tails, unaligned leading dimensions), split-K; and the fused
aggregate-in-LDS + contract kernel against the CSR aggregation followed by a float64 product."""
import numpy as np
import pytest
import torch
from torch_rgcn import routes  # noqa: E402

from oracle import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (7, 5, 3), (128, 128, 16), (130, 127, 33), (400, 200, 513), (37, 256, 200),
                                   (300, 2, 64), (16, 400, 40), (257, 129, 4), (404, 200, 400), (200, 400, 204), (1000, 16, 36),
                                   (68, 100, 52), (132, 144, 20), (64, 208, 16), (60, 212, 24)])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("bm", ["128", "64", "0"])
def test_gemm_layouts_and_edges(monkeypatch, M, N, K, ta, tb, bm):
    """bm: rows of the C tile per workgroup (rgcn_gemm_f32 picks 64-row tiles per launch when they balance better over the
    CUs; RGCN_GEMM_BM forces either form); 0 = the launcher's own choice, which includes the 64-row PANELS of up to 208 columns
    (gemm_panel_kernel: aligned operands whose N the 128-wide tiles would round up by more than the panels do -- 200, 400, 16, 100,
    144, 208, 212 here; every operand layout, K tails, rows past M, split-K)"""
    from torch_rgcn import _native
    routes.patch(monkeypatch, "gemm_bm", bm)
    rng = np.random.default_rng(M * 31 + N * 7 + K)
    A = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    B = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    ref = (A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64)
    At, Bt = torch.from_numpy(A).to(DEV), torch.from_numpy(B).to(DEV)
    assert _rel(_native.gemm(At, Bt, trans_a=ta, trans_b=tb).cpu().numpy(), ref) < 2e-6
    got = _native.gemm(At, Bt, bias=torch.from_numpy(bias).to(DEV), trans_a=ta, trans_b=tb, split_k=5)
    assert _rel(got.cpu().numpy(), ref + bias) < 2e-6


def test_gemm_split_k_is_bitwise_reproducible_and_skinny_k():
    from torch_rgcn import _native
    A = torch.randn(40_943, 400, device=DEV)       # dbases = ag^T g at WN18 size: K = nodes
    G = torch.randn(40_943, 200, device=DEV)
    a = _native.gemm(A, G, trans_a=True, split_k=64)
    b = _native.gemm(A, G, trans_a=True, split_k=64)
    assert torch.equal(a, b)
    ref = A.double().t() @ G.double()
    assert ((a.double() - ref).abs().max() / ref.abs().max()).item() < 1e-5


def test_matmul_mfma_autograd_matches_torch():
    from torch_rgcn import functional as F_
    A = torch.randn(37, 2, device=DEV, requires_grad=True)
    B = torch.randn(2, 200 * 200, device=DEV, requires_grad=True)
    g = torch.randn(37, 200 * 200, device=DEV)
    F_.matmul_mfma(A, B).backward(g)
    ga, gb = A.grad.clone(), B.grad.clone()
    A.grad = B.grad = None
    (A.double() @ B.double()).backward(g.double())
    assert ((ga.double() - A.grad.double()).abs().max() / A.grad.abs().max()).item() < 1e-5
    assert ((gb.double() - B.grad.double()).abs().max() / B.grad.abs().max()).item() < 1e-5


@pytest.mark.parametrize("N,R0,E,d_in,d_out,B", [(40_943, 18, 15_000, 200, 200, 2), (3000, 5, 40_000, 100, 100, 3), (700, 3, 5000, 64, 72, 5),
                                                 (257, 2, 900, 130, 7, 1), (5000, 4, 30_000, 68, 200, 7), (900, 3, 8000, 200, 16, 2)])
def test_basis_forward_aggregate_then_product(N, R0, E, d_in, d_out, B):
    """the basis layer's forward as it runs -- rgcn_basis_aggregate_f32 followed by rgcn_gemm_f32 (+ bias) -- against a float64
    aggregation by hand and a float64 product (the fused aggregate-in-LDS kernel of rounds 1-4 this test used to compare with is gone)"""
    from torch_rgcn import _native
    from torch_rgcn.graph import graph_from_nc_triples
    R = 2 * R0 + 1
    tp = oracle.add_inverse_and_self(oracle.synthetic_triples(N, R0, E, seed=N % 97), N, R0)
    g = graph_from_nc_triples(tp, N, R, False, torch.device(DEV))
    X = torch.randn(N, d_in, device=DEV)
    comps = torch.randn(R, B, device=DEV)
    bases = torch.randn(B, d_in, d_out, device=DEV) * 0.1
    bias = torch.randn(d_out, device=DEV)
    csr = g.csr("fwd")
    ag = _native.basis_aggregate(X, comps, csr, B, d_in, 1)
    rp = csr.rowptr[: N + 1].long()
    rows = torch.repeat_interleave(torch.arange(N, device=DEV), rp[1:] - rp[:-1])
    M = rows.numel()
    msg = (comps[csr.rel[:M].long()].double()[:, :, None] * (X[csr.src[:M].long()].double() * csr.val[:M, None].double())[:, None, :]).reshape(M, B * d_in)
    ag_ref = torch.zeros(N, B * d_in, device=DEV, dtype=torch.float64).index_add_(0, rows, msg)
    assert ((ag.double() - ag_ref).abs().max() / ag_ref.abs().max()).item() < 1e-5
    out = _native.gemm(ag, bases.view(B * d_in, d_out), bias=bias)
    ref = ag_ref @ bases.view(B * d_in, d_out).double() + bias.double()
    assert ((out.double() - ref).abs().max() / ref.abs().max()).item() < 1e-5
