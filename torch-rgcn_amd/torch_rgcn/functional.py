"""autograd bindings of the HIP kernels.

`relational_mp`  : out = sum_e val_e X[o_e] W[p_e] + b      (featured layers)
`featureless_mp` : out = sum_e val_e W[p_e, o_e, :] + b     (X = I)

Both are the custom-Function form of what the reference leaves to autograd
(torch_rgcn/layers.py:286-306; duals in SURVEY.md 8 a-9).  backward runs on the
autograd engine's thread and launches on that thread's current HIP stream; it uses
nothing but the tensors saved on ctx and the immutable RelGraph.
"""
import os

import torch

from . import routes

from . import _native


def _wgrad_tiles(plan):
    """destination tiles per work item of the tile-major weight gradient: more tiles amortise the item's set-up and its
    256 atomics (8 measured best at S1), fewer keep small graphs parallel"""
    env = routes.get("wgrad_tiles")
    return int(env) if env else (8 if plan.n_tiles >= 4096 else 4)


def _sparse_buckets(graph, W):
    """hidden 16 and (tile, relation) buckets so small that the 16-slot chunks are mostly padding"""
    if W.shape[1] != 16 or W.shape[2] != 16 or getattr(graph, "_dev", None) is None or getattr(graph, "sync_free", False) or \
            getattr(graph, "per_call", False):
        return False            # (the two-pass path sizes its scratch by a message count read back from the device)
    mode = routes.get("sparse_path", "auto")
    if mode != "auto":
        return mode == "1"
    fp = graph.fwd_plan(16)
    if not (fp.m_pad > 0 and fp.n_messages < 0.5 * fp.m_pad):
        return False
    # pass 2 walks a destination's messages with 4 lanes: keep mega-hubs on the tile path (which splits them)
    return graph.max_degree() <= 4096


def _fwd_blk(graph, relu):
    """the forward plan of tall tiles for rgcn_spmm_blk_f32, or None (graph too small, route off, ReLU epilogue on a plan with hub pieces,
    deterministic mode: the tile is summed in arrival order)"""
    if deterministic():
        return None
    plan = graph.fwd_blk_plan()
    if plan is None or (relu and _native._blk_units(plan)[2]):
        return None
    return plan


def _fwd_win(graph, relu):
    """the forward plan in soft-window order (graph.win_plan) for rgcn_spmm_blk_f32, or None; the ReLU epilogue needs tiles that are not cut
    into hub pieces (as _fwd_blk)"""
    plan = graph.win_plan("fwd") if hasattr(graph, "win_plan") else None
    if plan is None or (relu and _native._blk_units(plan)[2]):
        return None
    return plan


def _pad_blocks(X, W, bias, graph=None):
    """Widths up to 64 run on the MFMA block kernels (hidden-16 scheme over blocks of 16 features) with operands
    zero-padded to multiples of 16: a 40-byte row costs the same 128-byte fabric request as a 64-byte one, and these
    kernels (packed slots, pre-swizzled weight fragments, DPP folds) are several times faster than the generic-width
    one (AM: 10 -> 11 layer 2.9 -> 0.9 ms per launch, 32 -> 32 layer 7.5 -> 2.2 ms).
    -> (X', W', bias', (d_in, d_out)) or the inputs unchanged and None."""
    d_in, d_out = W.shape[1], W.shape[2]
    pi, po = -d_in % 16, -d_out % 16
    if max(d_in, d_out) > _BLOCKED_MAX or (pi == 0 and po == 0) or routes.get("pad16", "1") == "0" or \
            (graph is not None and _wide_gemm_path(graph, d_in, d_out)):       # (the gather-GEMM takes ragged widths as they are)
        return X, W, bias, None
    # one launch for W and the bias together (torch.nn.functional.pad: a fill and a copy each)
    if pi and _zero_padded_rows(X, d_in + pi):
        Xp = torch.as_strided(X, (X.shape[0], d_in + pi), (d_in + pi, 1))      # the producer wrote the padded rows already: no copy
    else:
        Xp = X if pi == 0 else _native.resize3(dense(X), (X.shape[0], d_in + pi))
    if bias is None:
        Wp, bp = _native.resize3(dense(W), (d_in + pi, d_out + po)), None
    else:
        Wp, bp = _native.resize3(dense(W), (d_in + pi, d_out + po), dense(bias), d_out + po)
    return Xp, Wp, bp, (d_in, d_out)


_BLOCKED_MAX = 512     # widest layer that is cut into 64-wide blocks of the MFMA block kernel


def _wide_gemm_path(graph, d_in, d_out):
    """undecomposed weights above width 64: relation-grouped gather-GEMM on the matrix cores + per-destination row sum
    (csrc/rgcn_gemm.hip); needs the device-side graph and host-known message counts (not the sync-free per-call build)"""
    return max(d_in, d_out) > 64 and getattr(graph, "_dev", None) is not None and not getattr(graph, "sync_free", False)


def _spmm_blocked(X, W, bias, plan_of, relu=False, graph=None, kind="fwd"):
    """spmm for any width.  Up to 64 x 64 it is one launch of the block kernels.  Above (undecomposed weights at d = 100,
    200, ...): relation-grouped gather-GEMM (X gathered once per message, every W_r through LDS once per 128 messages) + row
    sum; graphs without a device-side build keep round 1's (d_in / 64) x (d_out / 64) launches of the block kernel."""
    d_in, d_out = W.shape[1], W.shape[2]
    if graph is not None and _wide_gemm_path(graph, d_in, d_out):
        return _native.spmm_wide_two_pass(X, W, bias, graph.scatter_plan(kind, 8), graph.csr(kind), relu=relu)
    if max(d_in, d_out) <= 64 or d_in % 16 or d_out % 16 or max(d_in, d_out) > _BLOCKED_MAX or \
            routes.get("pad16", "1") == "0":
        return _native.spmm(X, W, bias, plan_of(d_out), relu=relu and max(d_in, d_out) <= 64)
    xs = [X[:, i:i + 64].contiguous() for i in range(0, d_in, 64)]
    cols = []
    for j in range(0, d_out, 64):
        wj = min(64, d_out - j)
        acc = None
        for bi, i in enumerate(range(0, d_in, 64)):
            part = _native.spmm(xs[bi], W[:, i:i + 64, j:j + wj].contiguous(),
                                bias[j:j + wj].contiguous() if (bias is not None and bi == 0) else None, plan_of(wj))
            acc = part if acc is None else acc.add_(part)
        cols.append(acc)
    return torch.cat(cols, dim=1)


def _zero_padded_rows(t, wide):
    """t = the first columns of a contiguous, 16-byte aligned [N, wide] buffer whose other columns are ZERO (the producer said so)?"""
    if not (getattr(t, "_rgcn_zero_padded", False) and _rows16(t) and t.stride(0) == wide):
        return False
    # the mark is a Python attribute: it survives set_() / resize_() / .data swaps of the tensor object -- the storage must still hold the rows
    return t.untyped_storage().nbytes() >= (t.storage_offset() + t.shape[0] * wide) * t.element_size()


def _unpad_blocks(dims, dX, dW, db, dx_view=False):
    if dims is None:
        return dX, dW, db
    d_in, d_out = dims
    if dX is not None and dX.shape[1] != d_in:
        if dx_view:       # X was an intermediate that arrived as the first columns of zero-padded rows: its gradient goes back the same way
            dX = dX[:, :d_in]                                  # (the padded columns of dX are exact zeros: W's padding rows are)
            dX._rgcn_zero_padded = True
        else:
            dX = _native.resize3(dense(dX), (dX.shape[0], d_in))
    if dW is not None and (dW.shape[1] != d_in or dW.shape[2] != d_out):      # contiguous gradients in one launch (a sliced view costs
        if db is not None:                                                     # AccumulateGrad a strided copy per tensor)
            dW, db = _native.resize3(dense(dW), (d_in, d_out), dense(db), d_out)
        else:
            dW = _native.resize3(dense(dW), (d_in, d_out))
    elif db is not None and db.shape[0] != d_out:
        db = db[:d_out]
    return dX, dW, db


def deterministic():
    """RGCN_DETERMINISTIC=1: bit-reproducible gradients.  The hidden-16 backward then runs the lean window kernel on 64-row tiles
    (wave-owned dX tiles: a fixed summation order), writes its per-workgroup dW partials with plain stores and two small kernels
    sum them in a fixed order (+0.1-0.15 ms per layer at S1) -- instead of the block-tile kernel, whose waves add to the shared
    dX tile, to dW and to the bias gradient in arrival order (fp32 sums differ in the last bits from run to run).  Outputs are
    reproducible either way."""
    return routes.get("deterministic", "0") == "1"


def _dense_buckets(plan):
    return plan.m_pad > 0 and plan.n_messages >= 0.5 * plan.m_pad


def dense(t):
    """contiguous AND 16-byte aligned (what the C ABI asks for): a contiguous view at an odd storage offset is copied"""
    t = t.contiguous()
    return t if t.data_ptr() % 16 == 0 else t.clone()


class _ReluToken:
    """Links a layer whose kernel applied ReLU in its epilogue (the producer of H = relu(pre)) with the layer that consumes
    H.  The consumer's fused backward can mask its feature gradient with H > 0 in the kernel's epilogue -- that IS the ReLU's
    backward -- and then records here WHICH tensor it returned; the producer's backward skips its own masking launch
    (aten.threshold_backward) only when the gradient it receives is that very tensor object, unmodified (same Python object
    through a weak reference, same version counter).  Anything else -- H had other consumers and autograd summed their
    gradients (a new tensor, or an in-place add that bumps the version), a hook replaced the tensor, the consumer took
    another route -- falls back to masking, which is idempotent on the pre-masked part: always exact.

    OPT-IN (round 5).  What the consumer hands autograd for H is then dL/d(pre-activation), not dL/dH -- they differ exactly where
    H == 0 -- and `torch.autograd.grad(loss, H)` (or a grad_fn pre-hook) reads that value without leaving any trace the backward could
    see.  So the consumer pre-masks only when the PRODUCER was told that H is private: `layer.forward_activated(x, "relu",
    private=True)` = "H goes into layers of this library and nobody asks autograd for its gradient" (models.py, whose hidden
    activation never leaves forward(), and bench.py do).  A bare `l2(l1.forward_activated(x, "relu"))` -- like the reference's
    `F.relu(self.rgc1(...))` -- returns the exact dL/dH; the producer masks in its own backward (one elementwise launch)."""
    __slots__ = ("ref", "version", "consumer_input", "private")

    def __init__(self, private=False):
        self.ref, self.version, self.consumer_input, self.private = None, -1, None, bool(private)

    def observed(self):
        """someone can SEE the gradient that arrives at H -- H.retain_grad() or a tensor hook on H: then it has to be dL/dH, not the
        pre-masked dL/d(pre-activation), and the consumer leaves the masking to the producer.  (torch.autograd.grad(loss, H) cannot
        be seen from here: it returns the pre-masked gradient, which differs from dL/dH exactly where H == 0.)"""
        h = self.consumer_input() if self.consumer_input is not None else None
        return h is None or h.retains_grad or bool(getattr(h, "_backward_hooks", None))

    def mark(self, dX):
        import weakref
        self.ref, self.version = weakref.ref(dX), dX._version

    def premasked(self, g):
        return self.ref is not None and self.ref() is g and g._version == self.version


def _fused_backward(X, W, g, graph, relu_in=False, want_db=False, diag4=False, sparse=False):
    """hidden 16, both gradients wanted: ONE walk of the transposed plan gathers G[s] once per message and produces dX
    and dW together (csrc/rgcn_bwd.hip).  None when the plan does not qualify (hub-split tiles, unpacked slots) or
    RGCN_BWD=split asks for round 1's two-pass backward.  relu_in: X is the output of a ReLU and dX is wanted before it
    (masked with X > 0 in the kernel's epilogue); returns (dX, dW, masked, db) -- db: the bias gradient when want_db and the kernel
    sums G's columns on the side (block-tile kernel), else None."""
    if W.shape[1] != 16 or W.shape[2] != 16 or routes.get("bwd", "fused") == "split":
        return None
    if not diag4 and not sparse and routes.get("bwd_own", "1") != "0" and _native.bwd_route() == "blk" and hasattr(graph, "win_plan"):
        # large static graph with dense buckets (S1): tall tiles in soft-window order, every relation's dW in the registers of its owner wave
        op = graph.win_plan("bwd_own")
        if op is not None and not _native._blk_units(op)[2]:
            dX, dW, db = _native.bwd_own(g, X, W, op, relu=relu_in, want_db=True)
            return dX, dW, relu_in, (db if want_db else None)
    bp = graph.bwd_blk_plan(diag4, sparse)                  # tall tiles, one per workgroup -- or the wave-owned 64-row plan
    diag4 = diag4 and bp is not None and _native._bwd_blk_plan(bp, True)   # block-diagonal W (4 x 4 blocks): only on the block-tile kernel
    if bp is None or not _native._bwd_blk_plan(bp, diag4):
        bp = graph.bwd_plan(16)
    if not _native.bwd_fused_ok(bp, diag4):
        return None
    masked = relu_in and _native.bwd_fused_relu_ok(bp, diag4)
    dX, dW, db = _native.bwd_fused(g, X, W, bp, atomic=not deterministic(), relu=masked, want_db=True, diag4=diag4)
    return dX, dW, masked, (db if want_db else None)


def _weight_gradient(X, W, g, graph):
    if _wide_gemm_path(graph, W.shape[1], W.shape[2]):
        # (the same 128-slot work items as the row GEMM: longer items flush fewer partial blocks but leave too few
        # workgroups on small per-call graphs -- measured 0.104 -> 0.184 ms at FB15k-237 shape -- and cost a second plan)
        return _native.wgrad_wide(X, g, graph.scatter_plan("fwd", 8), W.shape[0])
    fp = graph.fwd_plan(min(W.shape[2], 64))
    # tile-major walk (one random gather per message) unless a (tile, relation) run is so long that
    # one wave would serialise it (hub nodes): then the relation-major kernel with bounded work items
    # and unless the (tile, relation) buckets are so sparse that a work item is a fraction of a chunk
    tiled_ok = W.shape[1] == 16 and W.shape[2] == 16 and fp.max_run_chunks <= 64 and _dense_buckets(fp)
    if tiled_ok and routes.get("wgrad", "tiled") == "tiled":
        return _native.wgrad_tiled(X, g, fp, W.shape[0], _wgrad_tiles(fp))
    return _native.wgrad(X, g, graph.wgt_plan(), W.shape[0])


class _RelationalMP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, W, bias, graph, relu=False, blocks=None, in_token=None):
        ctx.in_token = in_token                      # X = relu(...) of a layer that fused the activation (see _ReluToken)
        # H itself, kept alive until the backward: at padded widths (AM: hidden 10) the tensor saved below is the [N, 16] buffer behind it, the
        # caller's H is a temporary, and _ReluToken.observed() must still be able to ask it whether anybody watches its gradient
        ctx.h_ref = X if in_token is not None else None
        ctx.out_token = _ReluToken(private=relu == "private") if relu else None
        relu = bool(relu)
        x_padded_view = W.shape[1] % 16 != 0 and _zero_padded_rows(X, W.shape[1] + (-W.shape[1] % 16))
        X, W, bias, ctx.dims = _pad_blocks(X, W, bias, graph)
        # the feature gradient of such an input returns as a view of the padded rows too (no crop launch) -- to an intermediate only: a
        # leaf's .grad stays a dense tensor of its own shape
        ctx.dx_view = bool(x_padded_view and ctx.dims is not None and routes.get("pad16_view", "1") != "0")
        X = dense(X)
        W = dense(W)
        b = None if bias is None else dense(bias)
        fused_relu = relu and (max(W.shape[1], W.shape[2]) <= 64 or _wide_gemm_path(graph, W.shape[1], W.shape[2]))   # kernel epilogues
        with _native.w16_scope() as w16:        # [R,16,16] weights: both fragment orders packed once, handed to the backward below
            if _sparse_buckets(graph, W) and blocks is not None and ctx.dims is None and tuple(blocks.shape[2:]) == (4, 4) and \
                    routes.get("block_fwd", "1") != "0":
                # W = block_diag(blocks), 4 x 4 blocks, sparse buckets (AM): the forward reads the blocks themselves on the CSR
                # kernel (block table in LDS; one gather per message, no transformed-row buffer): 0.46 ms against 0.65 ms for the
                # two passes below.  The backward stays on the dense W (dX rows + dW from one relation-major walk; autograd
                # through block_diag() picks the blocks' gradient out of dW).
                out = _native.block_spmm(X, blocks.detach().contiguous(), b, graph.csr("fwd"), relu=fused_relu)
            elif _sparse_buckets(graph, W) and routes.get("spmm_csr", "1") != "0" and \
                    _native.spmm_csr_d16_ok(graph.csr("fwd"), W.shape[0]):
                # sparse buckets, up to 120 relations: ONE pass over the destination-major CSR, messages of mixed relations, W in LDS
                out = _native.spmm_csr_d16(X, W, b, graph.csr("fwd"), relu=fused_relu)
            elif _sparse_buckets(graph, W) and _fwd_blk(graph, fused_relu) is not None:
                # sparse buckets, more relations than the CSR kernel's LDS holds (AM as shipped, layer 2: R = 267): the forward plan cut into
                # tall workgroup-owned tiles (31 messages per (tile, relation) bucket instead of ~2), one launch, no [M, 16] intermediate
                out = _native.spmm_blk(X, W, b, _fwd_blk(graph, fused_relu), relu=fused_relu)
            elif _sparse_buckets(graph, W):
                out = _native.spmm_two_pass(X, W, b, graph.scatter_plan("fwd"), graph.csr("fwd"), relu=fused_relu)
            elif W.shape[1] == 16 and W.shape[2] == 16 and _fwd_win(graph, fused_relu) is not None:
                # dense buckets on a large static graph (S1): tall workgroup-owned tiles walked in soft-window order -- the whole chip gathers
                # from a few MB of X at any time (0.33 ms against 0.42 on the wave-owned tiles, tools/softwin_probe.py)
                out = _native.spmm_blk(X, W, b, _fwd_win(graph, fused_relu), relu=fused_relu)
            else:
                out = _spmm_blocked(X, W, b, graph.fwd_plan, relu=fused_relu, graph=graph, kind="fwd")
        ctx.w16 = w16.pair(W)
        if relu and not fused_relu:
            out = torch.relu_(out)
        ctx.graph = graph
        ctx.has_bias = bias is not None
        ctx.relu = relu
        # W = block_diag(4 x 4 blocks) at width 16: the backward may keep only dW's diagonal blocks (block-tile kernel, R <= 447)
        ctx.diag4 = blocks is not None and ctx.dims is None and tuple(blocks.shape[2:]) == (4, 4) and W.shape[1] == 16 and W.shape[2] == 16
        if relu:
            ctx.save_for_backward(X, W, out)
        else:
            ctx.save_for_backward(X, W)
        if ctx.dims is None or out.shape[1] == ctx.dims[1]:
            return out
        if routes.get("pad16_view", "1") != "0":
            return out[:, :ctx.dims[1]]       # the first columns of the padded rows, no copy (MaskedCrossEntropy reads them in place)
        return _native.resize3(out, (out.shape[0], ctx.dims[1]))

    @staticmethod
    def backward(ctx, g):
        X, W = ctx.saved_tensors[:2]
        with _native.w16_scope(seed=(W, ctx.w16)):
            return _RelationalMP._backward(ctx, g, X, W)

    @staticmethod
    def _backward(ctx, g, X, W):
        graph = ctx.graph
        if ctx.dims is not None and ctx.dims[1] % 16:
            wide = ctx.dims[1] + (-ctx.dims[1] % 16)
            if _zero_padded_rows(g, wide):
                g = torch.as_strided(g, (g.shape[0], wide), (wide, 1))       # MaskedCrossEntropy wrote the zero-padded rows already
            else:
                g = _native.resize3(dense(g), (g.shape[0], wide))
        g = dense(g)
        if ctx.relu and not ctx.out_token.premasked(g):     # out = relu(pre): the gradient passes where the stored output is positive
            g = torch.ops.aten.threshold_backward(g, ctx.saved_tensors[2], 0.0)
        dX = dW = db = None
        sparse = _sparse_buckets(graph, W)
        both = None
        masked = False
        # sparse (tile, relation) buckets normally leave the tile plan -- except on a graph the block-tile kernel takes: its tall tiles
        # (128 .. 512 rows) hold several messages per bucket where a 64-row tile holds one or two, and dW of all relations (block-diagonal
        # weights: its diagonal blocks) fits the workgroup's LDS.  One launch instead of the two-pass backward's three.
        blk_sparse = sparse and routes.get("bwd", "fused") != "split" and \
            _native.bwd_blk_rows(graph.num_nodes, graph.num_rels, deterministic(), graph.device, ctx.diag4, True) > 0
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and (not sparse or blk_sparse):
            both = _fused_backward(X, W, g, graph, relu_in=ctx.in_token is not None and ctx.in_token.private and (ctx.dims is None or ctx.dims[0] % 16 == 0) and not ctx.in_token.observed(),
                                   want_db=ctx.has_bias and ctx.needs_input_grad[2], diag4=ctx.diag4, sparse=sparse)
            if both is not None:
                both, masked, db = both[:2], both[2], both[3]
        if both is None and sparse and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and routes.get("bwd", "fused") != "split" \
                and routes.get("twopass", "gather") == "gather" and not deterministic():
            # sparse buckets: relation-major walk, G[s] and X[o] gathered once each for dX's rows and dW together
            # (the producer's ReLU mask rides on the transformed rows: same condition as for the fused kernels above)
            masked = ctx.in_token is not None and ctx.in_token.private and not ctx.in_token.observed()
            both = _native.bwd_two_pass_fused(g, X, W, graph.scatter_plan("bwd"), graph.csr("bwd"), relu=masked)
        if both is not None:
            dX, dW = both
        else:
            if ctx.needs_input_grad[0]:
                Wt = W.transpose(1, 2).contiguous()
                if sparse:
                    dX = _native.spmm_two_pass(g, Wt, None, graph.scatter_plan("bwd"), graph.csr("bwd"))
                else:
                    dX = _spmm_blocked(g, Wt, None, graph.bwd_plan, graph=graph, kind="bwd")
            if ctx.needs_input_grad[1]:
                dW = _weight_gradient(X, W, g, graph)
        if ctx.has_bias and ctx.needs_input_grad[2] and db is None:
            db = _native.colsum(g)
        res = _unpad_blocks(ctx.dims, dX, dW, db, getattr(ctx, "dx_view", False))
        if masked:
            ctx.in_token.mark(res[0])                 # the tensor autograd is handed (a view or a crop of dX for padded widths) already is the
        return (*res, None, None, None, None)         # gradient BEFORE the producer's ReLU


def _join_shards(partial, group, mode="allreduce"):
    """Sum the ranks' partial N x d matrices (in place; returns the tensor to use).  mode (an argument -- the sharded
    layer's transport, torch_rgcn.dist.set_transport -- not process-global state):
      allreduce (default)  one RCCL all-reduce of the whole matrix
      rs_ag                reduce-scatter of row blocks + all-gather (the two halves of an all-reduce as separate
                           collectives: lets RCCL pick its direct algorithms per half; same bytes on every link)
      a2a                  direct exchange: all-to-all of the row blocks (every rank sends block j straight to rank j: on the
                           fully connected xGMI mesh all 7 links carry 1/8 of the matrix at once, a ring carries 7/8 of it
                           through every link in turn), local sum of the world_size received blocks, all-gather"""
    import torch.distributed as dist
    assert mode in ("allreduce", "rs_ag", "a2a"), f"unknown transport {mode!r}"
    if mode == "a2a":
        world = dist.get_world_size(group)
        n, d = partial.shape
        rows = -(-n // world)
        buf = partial if rows * world == n else torch.nn.functional.pad(partial, (0, 0, 0, rows * world - n))
        recv = torch.empty_like(buf)                         # recv[j] = rank j's partial of MY row block
        dist.all_to_all_single(recv, buf, group=group)
        shard = recv.view(world, rows * d).sum(dim=0).view(rows, d)      # fixed order: rank 0 .. world-1
        dist.all_gather_into_tensor(buf, shard, group=group)
        return buf if rows * world == n else buf[:n].contiguous()
    if mode == "rs_ag":
        world = dist.get_world_size(group)
        n, d = partial.shape
        rows = -(-n // world)
        buf = partial if rows * world == n else torch.nn.functional.pad(partial, (0, 0, 0, rows * world - n))
        shard = torch.empty((rows, d), device=partial.device, dtype=partial.dtype)
        dist.reduce_scatter_tensor(shard, buf, group=group)
        dist.all_gather_into_tensor(buf, shard, group=group)
        return buf if rows * world == n else buf[:n].contiguous()
    dist.all_reduce(partial, group=group)
    return partial


def _all_ranks_agree(mine, graph, group, key, device):
    """A route whose collectives differ from the fallback's (other row ranges, other slab cuts) may only be taken when EVERY rank of the
    group can take it: hub pieces, for one, exist on the rank that owns a hub relation and nowhere else (ranks deciding on their
    own would post mismatched all-reduces -- a hang or silently wrong dX).  One MIN all-reduce of the local eligibility per (graph, route),
    cached on the graph: every rank reaches this call on its first backward (the conditions in front of it are rank-invariant)."""
    import torch.distributed as dist
    cache = graph.__dict__.setdefault("_group_routes", {})
    if key not in cache:
        flag = torch.tensor([1 if mine else 0], device=device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        cache[key] = bool(int(flag.item()))
    return cache[key]


class _ShardedRelationalMP(torch.autograd.Function):
    """Relation shard of the featured layer.  Features and upstream gradient are replicated; every rank computes the
    partial output (and, in backward, the partial feature gradient) of ITS relations and the partials are summed over
    the group; rank 0 alone adds the bias (it is summed once).  n_slabs > 0: the partial is produced slab by slab and
    every finished slab is all-reduced asynchronously (RCCL on its own stream) while the next slab's kernels run --
    pays when the kernels are long against the collective (weak scaling); n_slabs = 0: one collective after the kernel
    (strong scaling at 8 GPUs: the kernel is ~8x shorter than the collective, nothing to hide behind)."""

    @staticmethod
    def forward(ctx, X, W, bias, graph, group, n_slabs, comm="allreduce"):
        import torch.distributed as dist
        X, W, bias, ctx.dims = _pad_blocks(X, W, bias)
        X, W = dense(X), dense(W)
        rank = dist.get_rank(group)
        b = dense(bias) if (bias is not None and rank == 0) else None
        with _native.w16_scope() as w16:
            if n_slabs > 0 and comm == "allreduce":
                works = []
                out = _native.spmm_slabs(X, W, b, graph.fwd_plan(W.shape[2]), n_slabs,
                                         lambda o, r0, r1: works.append(dist.all_reduce(o[r0:r1], group=group, async_op=True)))
                for w in works:
                    w.wait()
            else:
                out = _join_shards(_native.spmm(X, W, b, graph.fwd_plan(W.shape[2])), group, comm)
        ctx.w16 = w16.pair(W)
        ctx.graph, ctx.group, ctx.n_slabs, ctx.has_bias, ctx.comm = graph, group, n_slabs, bias is not None, comm
        ctx.save_for_backward(X, W)
        return out if ctx.dims is None else out[:, :ctx.dims[1]]

    @staticmethod
    def backward(ctx, g):
        X, W = ctx.saved_tensors
        with _native.w16_scope(seed=(W, ctx.w16)):
            return _ShardedRelationalMP._backward(ctx, g, X, W)

    @staticmethod
    def _backward(ctx, g, X, W):
        import torch.distributed as dist
        graph = ctx.graph
        if ctx.dims is not None and ctx.dims[1] % 16:
            wide = ctx.dims[1] + (-ctx.dims[1] % 16)
            if _zero_padded_rows(g, wide):
                g = torch.as_strided(g, (g.shape[0], wide), (wide, 1))       # MaskedCrossEntropy wrote the zero-padded rows already
            else:
                g = _native.resize3(dense(g), (g.shape[0], wide))
        g = dense(g)
        dX = dW = db = None
        works = []
        slabbed = ctx.n_slabs > 0 and ctx.comm == "allreduce"
        both = None
        joined = False
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and slabbed and not deterministic() and routes.get("bwd", "fused") != "split":
            # the fused backward slab by slab (round 5): the all-reduce of slab k's dX rows -- RCCL on its own stream -- runs under slab
            # k + 1's kernel; dW (owner-local rows) keeps adding across the slabs.  Plans with hub pieces keep the unfused slab path below.
            bp = graph.bwd_blk_plan()
            mine = bp is not None and W.shape[1] == 16 and W.shape[2] == 16 and _native.bwd_fused_slabs_ok(bp)
            if _all_ranks_agree(mine, graph, ctx.group, ("bwd_fused_slabs", ctx.n_slabs), g.device):
                dX, dW, _ = _native.bwd_fused_slabs(g, X, W, bp, ctx.n_slabs,
                                                    lambda o, r0, r1: works.append(dist.all_reduce(o[r0:r1], group=ctx.group, async_op=True)))
                both, joined = (dX, dW), True
        if both is None and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and not slabbed:
            both = _fused_backward(X, W, g, graph)
        if both is not None:
            dX, dW = both[:2]
        else:
            if ctx.needs_input_grad[0]:
                Wt = W.transpose(1, 2).contiguous()
                if slabbed:
                    dX = _native.spmm_slabs(g, Wt, None, graph.bwd_plan(W.shape[1]), ctx.n_slabs,
                                            lambda o, r0, r1: works.append(dist.all_reduce(o[r0:r1], group=ctx.group, async_op=True)))
                else:
                    dX = _native.spmm(g, Wt, None, graph.bwd_plan(W.shape[1]))
            if ctx.needs_input_grad[1]:   # owner-local: runs while the last slabs are still being reduced
                dW = _weight_gradient(X, W, g, graph)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _native.colsum(g)
        if dX is not None and not slabbed and not joined:
            dX = _join_shards(dX, ctx.group, ctx.comm)
        for w in works:
            w.wait()
        return (*_unpad_blocks(ctx.dims, dX, dW, db), None, None, None, None)


def sharded_relational_mp(features, weights, bias, graph, group, n_slabs=0, comm="allreduce"):
    return _ShardedRelationalMP.apply(features, weights, bias, graph, group, n_slabs, comm)


def _featureless_csr(graph, width):
    """the featureless layer's route: destination-major CSR kernels (one lane group per message) when the (tile, relation) buckets
    of the tile plan are mostly padding (AIFB: 91 relations on 8-row tiles, 23 slots per message) and the graph is static"""
    if getattr(graph, "_dev", None) is None or getattr(graph, "sync_free", False) or getattr(graph, "per_call", False) or \
            routes.get("featureless_csr", "auto") == "0":
        return False
    if routes.get("featureless_csr", "auto") == "1":
        return True
    fp = graph.fwd_plan(width)
    return not _dense_buckets(fp) and graph.max_degree() <= 4096


def _relu_epilogue(ctx, res, relu):
    """res: what a native forward returned for relu=True -- (out, applied in the kernel's epilogue) -- or the plain output; finishes the
    activation where the kernel could not, and keeps what the backward needs (see _ReluToken)"""
    ctx.relu = bool(relu)
    ctx.out_token = _ReluToken(private=relu == "private") if relu else None
    if not relu:
        return res
    out, applied = res if isinstance(res, tuple) else (res, False)
    return out if applied else torch.relu_(out)


def _relu_backward(ctx, g, out):
    """the gradient before the fused ReLU: masked here unless the consumer's backward kernel already did it (_ReluToken)"""
    if ctx.relu and not ctx.out_token.premasked(g):
        g = torch.ops.aten.threshold_backward(g, out, 0.0)
    return g


class _FeaturelessMP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, bias, graph, relu=False):
        table = dense(table)
        b = None if bias is None else dense(bias)
        ctx.csr = _featureless_csr(graph, table.shape[2])
        if ctx.csr:
            out = _relu_epilogue(ctx, _native.featureless_csr_fwd(table, b, graph.csr("fwd"), relu=bool(relu)), relu)
        else:
            out = _relu_epilogue(ctx, _native.featureless_fwd(table, b, graph.fwd_plan(table.shape[2])), relu)
        if relu:
            ctx.save_for_backward(out)
        ctx.graph = graph
        ctx.has_bias = bias is not None
        ctx.num_rels = table.shape[0]
        ctx.n_src = table.shape[1]
        ctx.width = table.shape[2]
        return out

    @staticmethod
    def backward(ctx, g):
        g = dense(g)
        if ctx.relu:
            g = _relu_backward(ctx, g, ctx.saved_tensors[0])
        dT = db = None
        if ctx.needs_input_grad[0]:
            if ctx.csr:
                dT = _native.featureless_csr_wgrad(g, ctx.graph.csr("fwd"), ctx.num_rels, ctx.n_src)
            else:
                dT = _native.featureless_wgrad(g, ctx.graph.fwd_plan(ctx.width), ctx.num_rels)
        if ctx.has_bias and ctx.needs_input_grad[1]:
            db = _native.colsum(g)
        return dT, db, None, None


class _BlockMP(torch.autograd.Function):
    """Block-diagonal per-relation weights without expanding them (reference layers.py:243-244 / :520-527 build
    block_diag(blocks) and run the dense path): out = sum_r A_r X blockdiag(B_r) + b on the CSR kernels of csrc/rgcn_block.hip.
    Relations past blocks.shape[0] (the LP layer's dense self-loop relation) are skipped -- the caller adds them."""

    @staticmethod
    def forward(ctx, X, blocks, bias, graph, relu):
        X, blocks = dense(X), dense(blocks)
        b = None if bias is None else dense(bias)
        out = _native.block_spmm(X, blocks, b, graph.csr("fwd"), relu=relu)
        ctx.save_for_backward(X, blocks, out if relu else None)
        ctx.graph = graph
        ctx.has_bias = bias is not None
        ctx.relu = relu
        return out

    @staticmethod
    def backward(ctx, g):
        X, blocks, out = ctx.saved_tensors
        g = dense(g)
        if ctx.relu:
            g = torch.ops.aten.threshold_backward(g, out, 0.0)
        graph = ctx.graph
        dX = dB = db = None
        if ctx.needs_input_grad[0]:
            dX = _native.block_spmm(g, blocks, None, graph.csr("bwd"), transposed=True)
        if ctx.needs_input_grad[1]:
            dB = _native.block_wgrad(X, g, graph.wgt_plan(), tuple(blocks.shape))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _native.colsum(g)
        return dX, dB, db, None, None


def block_mp(features, blocks, bias, graph, relu=False):
    """features [N, nb * bi], blocks [R', nb, bi, bo] -> [N, nb * bo]"""
    return _BlockMP.apply(features, blocks, bias, graph, relu)


def use_block_path(graph, blocks):
    """The block kernels need the device-side graph build (CSR + relation-major plan) and blocks of at most 8 x 8.  At
    width 16 the expanded 16 x 16 weights on the matrix-core kernels are faster (AM shape, 4 x 4 blocks: 1.96 ms per layer
    forward + backward against 2.43 -- every message re-reads its blocks from L2, 4x the bytes of its feature row), so
    RGCN_BLOCK_PATH=1 (default) takes the block kernels only above width 16; 2 = whenever supported; 0 = never."""
    mode = routes.get("block_path", "1")
    if mode == "0" or getattr(graph, "_dev", None) is None or not _native.block_supported(blocks.shape[2], blocks.shape[3]):
        return False
    wide = blocks.shape[1] * blocks.shape[2] > 16 or blocks.shape[1] * blocks.shape[3] > 16
    return wide or mode == "2"


class _DiagMP(torch.autograd.Function):
    """Diagonal per-relation weights (reference layers.py:289-292): out = sum_r A_r (X * w_r) + b, backward
    dX = sum_r A_r^T (G * w_r) (the same kernel on the transposed plan), dw_r = sum over the messages of r of val X[src] * G[dst]."""

    @staticmethod
    def forward(ctx, X, w, bias, graph):
        X, w = dense(X), dense(w)
        b = None if bias is None else dense(bias)
        out = _native.diag_spmm(X, w, b, graph.csr("fwd"))
        ctx.save_for_backward(X, w)
        ctx.graph = graph
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        X, w = ctx.saved_tensors
        g = dense(g)
        graph = ctx.graph
        dX = dw = db = None
        if ctx.needs_input_grad[0]:
            dX = _native.diag_spmm(g, w, None, graph.csr("bwd"))
        if ctx.needs_input_grad[1]:
            dw = _native.diag_wgrad(X, g, graph.wgt_plan(), w.shape[0])
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _native.colsum(g)
        return dX, dw, db, None


def diag_mp(features, w, bias, graph):
    """features [N, d], w [R, d] -> [N, d]"""
    return _DiagMP.apply(features, w, bias, graph)


def use_diag_path(graph, d):
    """the diagonal kernels need the device-side graph build (CSR + relation-major plan)"""
    return getattr(graph, "_dev", None) is not None and routes.get("diag_path") != "0"


def _split_k(K, M, N):
    """slices of the K dimension so that a skinny product (K = number of nodes, M x N = a weight matrix) still fills the chip"""
    tiles = -(-M // 128) * -(-N // 128)
    # (at most 64 slices: past that the partial products' round trip through HBM costs more than the extra workgroups buy --
    # tools/splitk_probe.py, WN18's dbases = ag^T g, K = 40,943: 64 slices 122.7 us, the former choice of 79 133.7, 128 142.0)
    # short K (the weight assembly's adjoint: K = relations, a few hundred) is ONE workgroup's serial loop of K / 16 steps, each a global
    # round trip with nothing to overlap it (32 us for K = 267): slices of at least three steps
    # slice length: 48 steps of K for short products, growing to 512 for long ones -- continuous in K (K // 48 below 8192 and K // 512
    # from there on gave K = 8191 64 slices and K = 8192 16)
    return int(max(1, min(64, (4 * 256) // max(tiles, 1), K // max(48, min(512, K // 16)))))


class _MatmulMFMA(torch.autograd.Function):
    """A @ B on rgcn_gemm_f32 (the weight assembly einsum('rb,bio->rio') of layers.py:241-242 as an [R, B] x [B, d_i d_o]
    product, and the other small dense products of the path): no rocBLAS."""

    @staticmethod
    def forward(ctx, A, B):
        A, B = dense(A), dense(B)
        ctx.save_for_backward(A, B)
        return _native.gemm(A, B)

    @staticmethod
    def backward(ctx, g):
        A, B = ctx.saved_tensors
        g = dense(g)
        dA = _native.gemm(g, B, trans_b=True) if ctx.needs_input_grad[0] else None          # g B^T
        # A^T g: K = the rows of A (nodes) -- split so that a small M x N output still fills the chip (fixed-order reduction)
        dB = _native.gemm(A, g, trans_a=True, split_k=_split_k(A.shape[0], A.shape[1], g.shape[1])) \
            if ctx.needs_input_grad[1] else None
        return dA, dB


def matmul_mfma(A, B):
    return _MatmulMFMA.apply(A, B)


class _BasisMP(torch.autograd.Function):
    """W_r = sum_b comps[r,b] bases[b] at large width: aggregate per basis, then contract (B d_in) x d_out on the matrix
    cores -- never touches an R x d x d weight tensor (reference: layers.py:241-242, :468-469).  Forward: aggregation
    kernel + hand-written MFMA GEMM (rgcn_gemm_f32; 57 TFLOP/s at WN18 size, rocBLAS addmm 50).  Backward:
    d_ag = g flat^T and dbases = ag^T g on rgcn_gemm_f32, dX by the same aggregation on the source-major CSR, dcomps by the
    relation-major dot-product kernel."""

    @staticmethod
    def forward(ctx, X, bases, comps, bias, graph):
        X, bases, comps = dense(X), dense(bases), dense(comps)
        B, d_in, d_out = bases.shape
        b = None if bias is None else dense(bias)
        ag = _native.basis_aggregate(X, comps, graph.csr("fwd"), B, d_in, 1)          # [N, B*d_in]
        out = _native.gemm(ag, bases.view(B * d_in, d_out), bias=b)
        ctx.graph, ctx.has_bias = graph, bias is not None
        ctx.save_for_backward(X, bases, comps, ag)
        return out

    @staticmethod
    def backward(ctx, g):
        X, bases, comps, ag = ctx.saved_tensors
        B, d_in, d_out = bases.shape
        g = dense(g)
        flat = bases.view(B * d_in, d_out)
        dX = dB = dC = db = None
        d_ag = _native.gemm(g, flat, trans_b=True) if (ctx.needs_input_grad[0] or ctx.needs_input_grad[2]) else None   # [N, B*d_in]
        if ctx.needs_input_grad[1]:
            dB = _native.gemm(ag, g, trans_a=True, split_k=_split_k(ag.shape[0], B * d_in, d_out)).view(B, d_in, d_out)
        if ctx.needs_input_grad[0]:
            dX = _native.basis_aggregate(d_ag, comps, ctx.graph.csr("bwd"), B, d_in, B)
        if ctx.needs_input_grad[2]:
            if getattr(ctx.graph, "per_call", False) and not deterministic() and _native.basis_dcomps_csr_ok(comps.shape[0], B, d_in):
                # per-step (LP) graphs: on the CSR the forward walked -- building a relation-major plan for this one kernel costs a dozen launches
                dC = _native.basis_dcomps_csr(X, d_ag, ctx.graph.csr("fwd"), comps.shape[0], B, d_in)
            else:
                dC = _native.basis_dcomps(X, d_ag, ctx.graph.wgt_plan(), comps.shape[0], B, d_in)
        if ctx.has_bias and ctx.needs_input_grad[3]:
            db = _native.colsum(g)
        return dX, dB, dC, db, None


class _FeaturelessBasisMP(torch.autograd.Function):
    """Featureless layer with basis decomposition WITHOUT the R x N x d_out weight table the reference
    materialises (layers.py:242 + :288; 17.8 GB for AM):  out[s] = sum_e val_e sum_b comps[r_e,b] bases[b,o_e,:]."""

    @staticmethod
    def forward(ctx, bases, comps, bias, graph, relu=False):
        out = _FeaturelessBasisMP._forward(ctx, bases, comps, bias, graph, relu)
        ctx.out_padded = bool(getattr(out, "_rgcn_zero_padded", False))     # (the saved copy comes back as another Python object)
        to_save = ctx.to_save
        del ctx.to_save
        ctx.save_for_backward(*to_save, *((out,) if relu else ()))
        return out

    @staticmethod
    def _forward(ctx, bases, comps, bias, graph, relu):
        B, N, d = bases.shape
        ctx.src_major = _native.fbasis_supported(B, d) and routes.get("fbasis", "src") == "src"
        # Layout of the table the source-major kernels walk.  Node-major [N, B, d] (a transposed copy per step, and a transposed
        # gradient back): a node's B rows are ONE contiguous run -- what a table far beyond the caches needs (AM as shipped: 2.7 GB,
        # rows of 40 bytes: reading them in place, 40 half-used lines per node, cost 33.5 ms per step against 24).  Basis-major
        # [B, N, d] = the parameter itself, no copies: wins while the table stays cache-resident (MUTAG: 45 MB, step 0.56 -> 0.51 ms).
        # Round 4: tables beyond the caches are walked in place too, by the tile kernels (rgcn_fbasis_tile.hip: 16 source nodes per tile,
        # staged through LDS with aligned 16-byte accesses, software-pipelined) -- no transposed copy, no transposed gradient.  They also
        # beat the wave-per-node kernels on a cache-resident table of MUTAG's size (45 MB: forward 42 -> 40 us, backward 65 -> 49 us), hence
        # the 32 MB threshold -- which only holds where the tile kernels can take over: a shape or mode they refuse (deterministic mode,
        # B > 64 ...) keeps round 3's 256 MB before it pays for the transposed copy.
        tile_ok, tile_mode = _native.fbasis_tile_ok(comps.shape[0], B, d, N, graph.fbasis_plan().max_src_degree) if ctx.src_major else (False, 0)
        limit_mb = int(routes.get("fbasis_inplace_mb")) if routes.is_set("fbasis_inplace_mb") else (32 if tile_ok else 256)
        ctx.in_place = ctx.src_major and B * N * d * 4 <= limit_mb << 20
        tiled, ctx.tile_mode = (tile_ok, tile_mode) if (ctx.src_major and not ctx.in_place) else (False, 0)
        ctx.tile_bwd = tiled
        if tiled:
            comps, bases = dense(comps), dense(bases)
            ctx.graph, ctx.has_bias = graph, bias is not None
            ctx.in_place = True
            ctx.to_save = (bases, comps)
            return _relu_epilogue(ctx, _native.fbasis_tile_fwd(bases, comps, bias, graph.fbasis_plan(), relu=bool(relu), mode=ctx.tile_mode,
                                                               padded=routes.get("pad16_view", "1") != "0"), relu)
        if ctx.src_major and ctx.in_place:
            comps, bases = dense(comps), dense(bases)
            ctx.graph, ctx.has_bias = graph, bias is not None
            ctx.to_save = (bases, comps)
            return _relu_epilogue(ctx, _native.fbasis_fwd(bases, comps, bias, graph.fbasis_plan(), basis_major=True, relu=bool(relu)), relu)
        table = bases.permute(1, 0, 2).contiguous()                   # [N, B, d]: one contiguous block per source node
        if ctx.src_major:   # every node's B x d block is read once
            comps = dense(comps)
            ctx.graph, ctx.has_bias = graph, bias is not None
            ctx.to_save = (table, comps)
            return _relu_epilogue(ctx, _native.fbasis_fwd(table, comps, bias, graph.fbasis_plan(), relu=bool(relu)), relu)
        comps = dense(comps)
        out = _native.basis_aggregate(table.view(N, B * d), comps, graph.csr("fwd"), B, d, B)
        if bias is not None:
            out += bias
        ctx.graph, ctx.has_bias = graph, bias is not None
        # small blocks (S2: B = 2): the backward walks the sources in order and reads / writes their blocks as B sequential streams -- straight
        # from the PARAMETER's [B, N, d] layout and into a gradient of that layout (round 5: no strided view for AccumulateGrad to copy);
        # only the forward's per-message gather wants the node-major copy (one 128-byte line per message instead of B half-used ones)
        ctx.small_in_place = (not deterministic()) and _native.fbasis_small_ok(comps.shape[0], B, d)
        ctx.to_save = (dense(bases), comps) if ctx.small_in_place else (table, comps)
        return _relu_epilogue(ctx, out, relu)

    @staticmethod
    def backward(ctx, g):
        if getattr(ctx, "tile_bwd", False) and g.dim() == 2 and g.shape[1] % 16 and _zero_padded_rows(g, g.shape[1] + (-g.shape[1] % 16)):
            # the consumer handed its feature gradient back as the first columns of zero-padded [N, 16] rows: the tile kernels gather those rows
            # in place (64-byte aligned: one line per row instead of 40-byte rows straddling two), the ReLU's mask and the bias' column sum run
            # on the padded buffer (its extra columns stay zero)
            wide = g.stride(0)
            g16 = torch.as_strided(g, (g.shape[0], wide), (wide, 1))
            if ctx.relu and not ctx.out_token.premasked(g):
                out = ctx.saved_tensors[2]
                out16 = torch.as_strided(out, (out.shape[0], wide), (wide, 1)) if (ctx.out_padded and _rows16(out) and out.stride(0) == wide) else \
                    _native.resize3(dense(out), (out.shape[0], wide))
                g16 = torch.ops.aten.threshold_backward(g16, out16, 0.0)
            table, comps = ctx.saved_tensors[:2]
            dB, dC = _native.fbasis_tile_bwd(table, comps, g16[:, :g.shape[1]], ctx.graph.fbasis_plan(), ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                             mode=ctx.tile_mode)
            db = _native.colsum(g16)[:g.shape[1]].contiguous() if ctx.has_bias and ctx.needs_input_grad[2] else None
            return dB, dC, db, None, None
        g = dense(g)
        if ctx.relu:
            g = _relu_backward(ctx, g, ctx.saved_tensors[2])
        if ctx.src_major:
            table, comps = ctx.saved_tensors[:2]
            if getattr(ctx, "tile_bwd", False):
                dB, dC = _native.fbasis_tile_bwd(table, comps, g, ctx.graph.fbasis_plan(), ctx.needs_input_grad[0], ctx.needs_input_grad[1], mode=ctx.tile_mode)
            else:
                dB, dC = _native.fbasis_bwd(table, comps, g, ctx.graph.fbasis_plan(), ctx.needs_input_grad[0],
                                            ctx.needs_input_grad[1], basis_major=ctx.in_place)
            if dB is not None and not ctx.in_place:
                dB = dB.permute(1, 0, 2)      # a view: autograd accumulates it into the [B, N, d] parameter gradient
            db = _native.colsum(g) if ctx.has_bias and ctx.needs_input_grad[2] else None
            return dB, dC, db, None, None
        table, comps = ctx.saved_tensors[:2]
        dB = dC = db = None
        if getattr(ctx, "small_in_place", False):
            # small blocks (S2: B = 2, d = 16): one walk of the source-major CSR for both gradients (one gather per message), in the parameter's layout
            B, N, d = table.shape
            dB, dC = _native.fbasis_small_bwd(g, table, comps, ctx.graph.csr("bwd"), B, d, basis_major=True)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = _native.colsum(g)
            return (dB if ctx.needs_input_grad[0] else None), (dC if ctx.needs_input_grad[1] else None), db, None, None
        N, B, d = table.shape
        if ctx.needs_input_grad[0]:
            dB = _native.basis_aggregate(g, comps, ctx.graph.csr("bwd"), B, d, 1).view(N, B, d).permute(1, 0, 2)
        if ctx.needs_input_grad[1]:
            dC = _native.basis_dcomps(g, table.view(N, B * d), ctx.graph.wgt_plan(), comps.shape[0], B, d, swap=True)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _native.colsum(g)
        return dB, dC, db, None, None


def featureless_basis_mp(bases, comps, bias, graph, relu=False):
    return _FeaturelessBasisMP.apply(bases, comps, bias, graph, relu)


def basis_mp(features, bases, comps, bias, graph):
    return _BasisMP.apply(features, bases, comps, bias, graph)


def use_basis_path(num_bases, d_in, d_out, graph):
    """aggregate-then-contract pays when the per-message d_in x d_out product is large and B is small"""
    if routes.get("basis_path") == "0" or getattr(graph, "_dev", None) is None:
        return False
    return d_in * d_out >= 64 * 64 and num_bases <= 8


def relational_mp(features, weights, bias, graph, relu=False, blocks=None):
    """features [N, d_in], weights [R, d_in, d_out] (dense), bias [d_out] or None -> [N, d_out]; relu=True applies the
    activation in the kernel's epilogue (the backward masks the upstream gradient with the stored output), relu="private" also lets
    the layer that consumes the output mask on this layer's behalf (see _ReluToken: the caller promises that nobody asks autograd for
    the output's gradient); blocks: the
    [R, nb, bi, bo] parameter when weights = block_diag(blocks) (a hint: lets the forward skip the zero entries)"""
    # features = the output of a layer that applied ReLU in its kernel's epilogue: its backward node carries the token
    in_token = getattr(getattr(features, "grad_fn", None), "out_token", None)
    if in_token is not None:
        import weakref
        in_token.consumer_input = weakref.ref(features)     # looked at again in the backward: see _ReluToken.observed
    return _RelationalMP.apply(features, weights, bias, graph, relu, blocks, in_token)


def featureless_mp(table, bias, graph, relu=False):
    """table [R, N, d_out] -> [N, d_out]; relu: the activation the models apply right after the layer, in the kernel's epilogue"""
    return _FeaturelessMP.apply(table, bias, graph, relu)


class _DistMultScore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, triples, nodes, relations, sbias, pbias, obias):
        shape = triples.shape[:-1]
        tr = triples.reshape(-1, 3).contiguous()
        nodes = dense(nodes)
        relations = dense(relations)
        # the backward pass walks the scored triples as two CSRs (unless routed to the atomic scatter): their counting pass rides on
        # the scoring kernel
        ctx.ranks = None
        if any(ctx.needs_input_grad[1:]) and tr.shape[0] and routes.get("distmult_bwd", "csr") != "atomic":
            scores, ctx.ranks = _native.distmult_fwd(tr, nodes, relations, sbias, pbias, obias, ranks=True)
        else:
            scores = _native.distmult_fwd(tr, nodes, relations, sbias, pbias, obias)
        ctx.save_for_backward(tr, nodes, relations)
        ctx.with_bias = sbias is not None
        ctx.shape = shape
        return scores.view(shape)

    @staticmethod
    def backward(ctx, gs):
        tr, nodes, relations = ctx.saved_tensors
        gs = gs.reshape(-1)
        gs = dense(gs)
        mode = routes.get("distmult_bwd", "csr")
        scatter = ctx.ranks is None
        if not scatter and mode != "split" and _native.distmult_bwd_all_supported(relations.shape[0], nodes.shape[1]):
            # small relation tables (WN18: 18 x 200): every gradient from the two CSR walks, no predicate sort
            dn, dr, dsb, dpb, dob = _native.distmult_bwd_all(tr, ctx.ranks, nodes, relations, gs, ctx.with_bias)
            return None, dn, dr, dsb, dpb, dob
        order = torch.argsort(tr[:, 1], stable=True)   # predicate runs -> relation gradient accumulates in registers
        dn, dr, dsb, dpb, dob = _native.distmult_bwd(tr[order].contiguous(), nodes, relations, gs[order].contiguous(),
                                                     ctx.with_bias, nodes_grad=scatter)
        if not scatter:      # entity gradients: CSR by subject / by object, one wave per entity, no atomics
            dn = _native.distmult_bwd_nodes(tr, ctx.ranks, nodes, relations, gs)
        return None, dn, dr, dsb, dpb, dob


def distmult_score(triples, nodes, relations, sbias=None, pbias=None, obias=None):
    return _DistMultScore.apply(triples, nodes, relations, sbias, pbias, obias)


_UNIT = {}


def unit_gradient(device):
    """the constant 1.0 to start a backward pass with -- `loss.backward(gradient=unit_gradient(loss.device))` -- instead of the ones_like()
    autograd fills per call; MaskedCrossEntropy's backward recognises it and skips the multiplication by it (two launches per step of a
    launch-bound graph).  Never written to."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if device not in _UNIT:
        _UNIT[device] = torch.ones((), device=device)
    return _UNIT[device]


def _is_unit(g):
    u = _UNIT.get(g.device)
    return u is not None and g.dim() == 0 and g.data_ptr() == u.data_ptr() and u._version == 0


def _rows16(t):
    """t = the first columns of a contiguous [N, 16 k] buffer (what a layer of fewer than 16 k outputs returns, see _RelationalMP.forward)?"""
    return t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 16 == 0 and t.stride(0) > t.shape[1] and t.storage_offset() == 0 and \
        t.data_ptr() % 16 == 0


class _MaskedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, row_label, lab_rows):
        logits = logits if _rows16(logits) else dense(logits)      # the padded rows are read in place
        loss, dl, dl_full = _native.ce_head(logits, row_label, lab_rows)
        ctx.save_for_backward(dl)
        ctx.dl_full = dl_full is not None
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        dl, = ctx.saved_tensors
        if not _is_unit(g):
            return dl * g, None, None
        if ctx.dl_full:      # dl = the first columns of a zero-padded [N, 16] buffer: the layer's backward takes the buffer as it is
            dl._rgcn_zero_padded = True
        return dl, None, None


class _BCEWithLogits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, labels):
        loss, ds = _native.bce_head(dense(scores.reshape(-1)), dense(labels.reshape(-1).to(torch.float32)))
        ctx.save_for_backward(ds)
        ctx.shape = scores.shape
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        ds, = ctx.saved_tensors
        return (ds if _is_unit(g) else ds * g).view(ctx.shape), None


def bce_with_logits(scores, labels):
    """`F.binary_cross_entropy_with_logits(scores, labels)` (mean reduction; reference experiments/predict_links.py:152-153) as ONE kernel for the
    loss and its gradient instead of ATen's ~13 launches around a few hundred thousand scalars (rgcn_bce_head_f32); labels get no gradient"""
    return _BCEWithLogits.apply(scores, labels)


class MaskedCrossEntropy(torch.nn.Module):
    """`criterion(logits[idx, :], labels)` with nn.CrossEntropyLoss() (reference experiments/classify_nodes.py:107-110) for a FIXED
    set of labelled nodes, as one kernel for the loss and its gradient instead of ATen's ~13 launches (gather, log-softmax, nll,
    their backwards and an index_put through a sort): `MaskedCrossEntropy(idx, labels, num_nodes)(logits)`; idx must not repeat."""

    def __init__(self, idx, labels, num_nodes):
        super().__init__()
        idx, labels = idx.reshape(-1).long(), labels.reshape(-1).long()
        assert idx.numel() == labels.numel() and idx.numel() > 0 and torch.unique(idx).numel() == idx.numel(), "labelled nodes: one label each"
        # (one read-back at construction) the kernel indexes a row of logits with the label: checked against the class count in forward,
        # as ATen's CrossEntropyLoss device-asserts
        self._label_range = (int(labels.min().item()), int(labels.max().item()))
        row_label = torch.full((num_nodes,), -1, dtype=torch.int32, device=idx.device)
        row_label[idx] = labels.to(torch.int32)
        self.register_buffer("row_label", row_label, persistent=False)
        self.register_buffer("lab_rows", idx.to(torch.int32).contiguous(), persistent=False)

    def forward(self, logits):
        assert logits.dim() == 2 and logits.shape[0] == self.row_label.shape[0]
        assert int(logits.shape[1]) <= 64, "at most 64 classes"
        lo, hi = self._label_range
        if lo < 0 or hi >= int(logits.shape[1]):
            raise IndexError(f"MaskedCrossEntropy: labels span {lo} .. {hi} but the logits have {int(logits.shape[1])} classes")
        return _MaskedCE.apply(logits, self.row_label, self.lab_rows)
