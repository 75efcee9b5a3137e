"""Relation-sharded multi-GPU message passing (one process per GPU, RCCL over xGMI).

out = sum_r A_r X W_r is a sum over relations and the normalisation is local to a
(relation, node) pair, so relation buckets are independent units:
every rank holds the full node-feature matrix, the messages and weights of ITS
relations, and one sum all-reduce of the partial N x d output joins them (forward);
the feature gradient is the mirror image (all-reduce of the partial N x d_in dX).
The reference has no multi-GPU code at all.

    _CopyToShards      forward identity          backward all-reduce(sum)
    _ReduceFromShards  forward all-reduce(sum)   backward identity

`backend "nccl"` is RCCL on ROCm; the same code runs on gloo for the CPU tests.
"""
import numpy as np
import torch

from . import routes
import torch.distributed as dist


def _owned(t):
    """the tensor itself when it is a dense buffer of its own (what the HIP kernels return: nobody else holds it), else a dense copy --
    the all-reduces below run IN PLACE on it (round 6: no 64 MB clone in front of every collective at S1's size)"""
    return t if (t.is_contiguous() and t._base is None) else t.contiguous().clone()


class _CopyToShards(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        # g is this rank's partial feature gradient, fresh from the local function's backward (autograd sums several consumers into a
        # buffer of its own): reduced in place
        g = _owned(g)
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        return g, None


class _ReduceFromShards(torch.autograd.Function):
    @staticmethod
    def forward(ctx, partial, group):
        # in place on the local function's output (declared to autograd: a local function that saved its output for its own backward
        # fails loudly there instead of reading the reduced values)
        out = _owned(partial)
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
        if out is partial:
            ctx.mark_dirty(partial)
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None


def sharded_apply(local_fn, features, group):
    """local_fn(features) -> this rank's partial output; returns the all-reduced sum.
    `features` must hold the same values on every rank (replicated)."""
    x = features if features is None else _CopyToShards.apply(features, group)
    return _ReduceFromShards.apply(local_fn(x), group)


def partition_relations(message_counts, world_size):
    """Greedy longest-processing-time packing of relations onto ranks by message count.
    Returns owner[r] in [0, world_size).  Deterministic (ties -> lower relation id, lower rank)."""
    counts = np.asarray(message_counts, dtype=np.int64)
    order = sorted(range(len(counts)), key=lambda r: (-int(counts[r]), r))
    load = [0] * world_size
    owner = np.zeros(len(counts), dtype=np.int64)
    for r in order:
        k = min(range(world_size), key=lambda i: (load[i], i))
        owner[r] = k
        load[k] += int(counts[r])
    return owner


def _sum_over_group(group):
    def hook(grad):
        g = grad.contiguous().clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
        return g
    return hook


# parameters whose gradient is a SUM over relations (every rank holds a partial sum) vs. parameters with one row per
# relation (the gradient of a relation's row lives on the rank that owns the relation)
_SUMMED_PARAMS = ("bases",)
_PER_RELATION_PARAMS = ("weights", "comps", "blocks")


TRANSPORTS = ("allreduce", "rs_ag", "a2a")


def set_transport(layer, comm="allreduce", slabs=0):
    """How a sharded layer sums its partial N x d matrices over the group (functional._join_shards): "allreduce", "rs_ag"
    (reduce-scatter + all-gather), "a2a" (direct exchange: all-to-all of row blocks + local sum + all-gather); slabs > 0
    (with "allreduce"): the partial is produced slab by slab and every slab is reduced asynchronously while the next one's
    kernels run.  State of the layer, not of the process.  (There is no "no collective" transport: bench.py's compute-alone leg
    patches functional._join_shards locally.)"""
    assert comm in TRANSPORTS, f"unknown transport {comm!r}"
    layer._shard_transport = (comm, int(slabs))
    return layer


def shard_layer(layer, group=None, keep="all", comm=None, slabs=None):
    """Turn a RelationalGraphConvolutionNC into one shard of a relation-sharded layer.

    keep="all"  : the layer's triples already ARE this rank's relations (weak scaling: every rank
                  was built on its own relation bucket); only the collectives are added.
    keep="lpt"  : the layer holds the FULL graph; this rank keeps the relations LPT-assigned to it
                  (normalisation is computed on the full graph first, then messages are filtered).

    Parameter gradients: `bias` is computed from the replicated upstream gradient (identical on every rank);
    `bases` (basis decomposition: W_r = sum_b comps[r,b] bases[b]) receives contributions from every relation, so its
    gradient is all-reduced here (a hook; B x d_in x d_out floats; both keep modes) and the replicas stay in step;
    `weights` / `comps` / `blocks` have one row per relation and the row's gradient exists on the owner only.  Rows of
    relations a rank does not own get a ZERO gradient there, but an optimiser with weight decay or momentum still moves
    them: they are not authoritative -- CHECKPOINTS MUST GO THROUGH `gather_owned_parameters()` (a plain state_dict()
    of one rank holds stale rows for the relations it does not own), and `sync_owned_parameters()` copies the owners'
    rows back into every replica (call it before evaluation on a single rank or every few hundred steps).
    """
    group = group if group is not None else dist.group.WORLD
    layer._shard_group = group
    layer._shard_keep = keep
    layer._graph = None  # rebuild with the filter
    # transport: argument, else the routes' default (dist_comm / dist_slabs), else all-reduce
    set_transport(layer, comm if comm is not None else routes.get("dist_comm", "allreduce"),
                  slabs if slabs is not None else int(routes.get("dist_slabs", "0")))
    # `bases` sums over ALL relations whichever way the relations were split: its gradient is all-reduced in both modes
    if not getattr(layer, "_shard_hooks", None):
        layer._shard_hooks = [getattr(layer, n).register_hook(_sum_over_group(group))
                              for n in _SUMMED_PARAMS if getattr(layer, n, None) is not None]
    return layer


@torch.no_grad()
def gather_owned_parameters(layer):
    """{name: complete tensor} of the per-relation parameters of a keep="lpt" shard: every relation's row is taken from
    its owner (sum all-reduce of the rows masked by ownership).  Collective: call it on every rank."""
    group = layer._shard_group
    graph = layer._graph
    assert graph is not None and getattr(graph, "owned_relations", None) is not None, \
        "run one forward (the shard's graph is built lazily) before gathering parameters"
    out = {}
    for name in _PER_RELATION_PARAMS:
        p = getattr(layer, name, None)
        if p is None:
            continue
        mask = torch.zeros(p.shape[0], dtype=p.dtype, device=p.device)
        owned = torch.as_tensor(graph.owned_relations, device=p.device, dtype=torch.long)
        mask[owned[owned < p.shape[0]]] = 1        # LP-style `blocks` has R-1 rows (the self-loop relation is separate)
        full = p.detach() * mask.view(-1, *([1] * (p.dim() - 1)))
        dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group)
        out[name] = full
    return out


@torch.no_grad()
def sync_owned_parameters(layer):
    """Overwrite every replica's per-relation parameters with the owners' rows (collective).  After it, state_dict() is
    the same, complete model on every rank."""
    for name, full in gather_owned_parameters(layer).items():
        getattr(layer, name).copy_(full)


def filter_graph_for_rank(graph, group):
    """Drop the messages of relations owned by other ranks (in place, before any plan is built)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if getattr(graph, "_dev", None) is not None:   # message list lives on the GPU: shard through the alive mask
        s, p, o, val, alive = graph._dev
        live = torch.ones_like(p, dtype=torch.bool) if alive is None else alive != 0
        counts = torch.bincount(p[live].long(), minlength=graph.num_rels).cpu().numpy()
        owner = partition_relations(counts, world)
        mine = torch.from_numpy(owner == rank).to(p.device)[p.long()] & live
        graph._dev = (s, p, o, val, mine.to(torch.uint8))
        graph.num_messages = int(mine.sum().item())
        graph.owned_relations = np.nonzero(owner == rank)[0]
        return graph
    counts = np.bincount(graph._p, minlength=graph.num_rels)
    owner = partition_relations(counts, world)
    m = owner[graph._p] == rank
    graph._s, graph._p, graph._o, graph._val = graph._s[m], graph._p[m], graph._o[m], graph._val[m]
    graph.num_messages = int(m.sum())
    graph.owned_relations = np.nonzero(owner == rank)[0]
    return graph
