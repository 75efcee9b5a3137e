"""Device-resident graph layout for the HIP message-passing kernels.

A `RelGraph` is what replaces the per-forward `stack_matrices -> sum_sparse ->
sparse COO` pipeline of the reference (torch_rgcn/utils.py:143-166, :71-97;
torch_rgcn/layers.py:255-279, :490-516): the augmented triples are normalised once
(literal procedure, so the horizontal/LP quirks are preserved), bucketed into
relation-tile plans and uploaded.  NC layers build it once; LP layers per call.

Three plans, all produced by the same host routine (csrc/rgcn_host.cpp):
  fwd    destination = subject s, tiles over s      -> out = sum val X[o] W_p
  bwd    destination = object  o, tiles over o      -> dX  = sum val G[s] W_p^T
  wgt    one tile (relation-major)                  -> dW_p = sum val X[o]^T G[s]
"""
import os

import numpy as np
import torch

from . import routes

from . import _native

_LDS_WAVE_FLOATS = 2048  # 8 KiB of LDS per wave-owned destination tile (4 waves per workgroup)


def pick_tile_rows(width, n_rows=None):
    """Rows of a (wave-owned) destination tile for an output width in floats.  Padding of the
    (tile, relation) buckets is measured to be free in the gather kernels, so small tiles
    (more waves in flight) win; the LDS row stride is the width rounded up to 4 floats.
    Small graphs get smaller tiles still: a tile is one wave, and a few dozen waves cannot fill 256 CUs."""
    ld = (max(1, width) + 3) & ~3
    env = routes.get("tile_rows")
    if env:
        return max(1, min(int(env), 4096 // ld))
    rows = max(1, min(128, _LDS_WAVE_FLOATS // ld))        # widths up to 4096 floats fit the 64 KiB workgroup budget
    if n_rows is not None and n_rows < rows * 2048:
        rows = max(min(rows, 8), min(rows, n_rows // 2048))
    return int(rows)


_MAX_CELLS = 2 ** 34      # 64 GiB per counting table
_SOFTWIN_MIN_CHUNKS = 4   # soft-window plans: chunks per (tile, relation) bucket, on average, from which the source order pays
_OWN_MAX_IMBALANCE = 1.25  # relation-owner backward: largest (messages of the busiest wave) / (mean over the waves) it is used at


class RelGraph:
    def __init__(self, triples_plus, val, num_nodes, num_rels, device):
        """triples_plus: int64 numpy [M,3] (s,p,o); val: float32 numpy [M].  (host-built plans)"""
        self.num_nodes, self.num_rels = int(num_nodes), int(num_rels)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("RelGraph lives on a GPU: torch_rgcn runs on HIP kernels only (no CPU fallback)")
        # the graph build counts messages in dense (relation, node) tables: 4 bytes x num_nodes x num_relations each (64-bit
        # cell indices on the device).  The limit is memory, not addressing: 2^36 cells = 256 GiB is past one MI355X.
        if self.num_nodes * self.num_rels >= _MAX_CELLS:
            raise NotImplementedError(f"num_nodes * num_relations = {self.num_nodes * self.num_rels:,} needs a "
                                      f"{4 * self.num_nodes * self.num_rels / 2**30:.0f} GiB counting table (limit {_MAX_CELLS:,} cells)")
        self._dev = None
        self._plans = {}
        self.perm = self.inv = None   # locality relabelling of the nodes (graph_from_nc_triples(relabel=...)): perm[old] = new, inv[new] = old
        self.sync_free = False      # True: plans are sized by upper bounds and finished on the device (no host read-back)
        if triples_plus is None:
            return
        tp = np.ascontiguousarray(triples_plus, dtype=np.int64).reshape(-1, 3)
        self._s = tp[:, 0].astype(np.int32)
        self._p = tp[:, 1].astype(np.int32)
        self._o = tp[:, 2].astype(np.int32)
        self._val = np.ascontiguousarray(val, dtype=np.float32)
        self.num_messages = tp.shape[0]

    @classmethod
    def on_device(cls, s, p, o, val, alive, n_live, num_nodes, num_rels):
        """Message list already on the GPU (int32 s/p/o, fp32 val, optional uint8 alive): plans are built by
        the device-side counting sort (csrc/rgcn_build.hip)."""
        g = cls(None, None, num_nodes, num_rels, s.device)
        g._dev = (s, p, o, val, alive)
        g.num_messages = int(n_live)
        return g

    # -- plans are built lazily and cached per tile height
    def _plan(self, kind, tile_rows, max_item_chunks=64):
        # RGCN_DETERMINISTIC=1: no hub pieces (a tile cut into several work units merges its pieces with fp32 atomics, in
        # arrival order): a hub tile is one long unit for one wave -- slow on skewed graphs, bit-reproducible.  Part of the cache
        # key: switching the route after a first forward must not reuse plans of the other kind.
        whole = routes.get("deterministic", "0") == "1"
        key = (kind, tile_rows, max_item_chunks, whole)
        if key not in self._plans and self._dev is not None:
            N, R = self.num_nodes, self.num_rels
            s, p, o, val, alive = self._dev
            dst, src = (s, o) if kind == "fwd" else (o, s)
            # plans of per-call (LP) graphs are always finished on the device: 4 read-backs per plan cost more than they buy on
            # graphs that live for one step (upper-bound sizes: at most 16 slots per message; one work unit per tile, no hub
            # splitting -- a sampled graph's hub is a few thousand messages).  Static (NC) graphs keep the exact path.
            nosync = self.sync_free or getattr(self, "per_call", False)
            self._plans[key] = _native.build_plan_device(dst, src, p, val, alive, N, N, R, tile_rows, self.num_messages,
                                                         max_item_chunks, want_runs=True, want_pack=True, sync_free=nosync,
                                                         **({"max_unit_chunks": 1 << 30} if whole else {}))
        if key not in self._plans:
            N, R = self.num_nodes, self.num_rels
            if kind == "fwd":
                hp = _native.build_plan_host(self._s, self._o, self._p, self._val, N, N, R, tile_rows, max_item_chunks,
                                             want_runs=True, want_pack=True, **({"max_unit_chunks": 1 << 30} if whole else {}))
            elif kind == "bwd":
                hp = _native.build_plan_host(self._o, self._s, self._p, self._val, N, N, R, tile_rows, max_item_chunks,
                                             want_runs=True, want_pack=True, **({"max_unit_chunks": 1 << 30} if whole else {}))
            else:
                raise KeyError(kind)
            self._plans[key] = _native.DevicePlan(hp, self.device)
        return self._plans[key]

    def fwd_plan(self, d_out):
        return self._plan("fwd", pick_tile_rows(d_out, self.num_nodes))

    def bwd_plan(self, d_in):
        """transposed plan with wave-owned tiles (spmm on the transposed graph, the wave-owned fused backward kernels)"""
        rows = pick_tile_rows(d_in, self.num_nodes)
        if d_in == 16 and not routes.is_set("tile_rows"):
            # hidden 16: the fused backward kernels (dX + dW in one walk) keep a dX tile, an X tile, a transposition scratch and
            # the dW hand-over slots in LDS: 64-row tiles (measured in round 2: 0.71 ms at 64 rows, 0.80 at 128)
            rows = int(routes.get("bwd_tile_rows")) if routes.is_set("bwd_tile_rows") else min(rows, 64)
        return self._plan("bwd", rows)

    def bwd_blk_plan(self, diag4=False, sparse=False):
        """transposed plan of TALL tiles (one per workgroup, up to 255 / 512 rows) for the block-tile backward kernel, or None
        when that kernel does not apply (small graph, too many relations, RGCN_DETERMINISTIC=1, RGCN_BWD_KERNEL != blk).
        Only rgcn_bwd_blk_f32 can walk it -- every other kernel gets bwd_plan()."""
        if routes.is_set("tile_rows"):
            return None
        if routes.is_set("bwd_tile_rows") and _native.bwd_route() == "blk":
            rows = int(routes.get("bwd_tile_rows"))         # experiments (tools/r3_blk.sh)
            return self._plan("bwd", rows) if rows > 64 else None
        rows = _native.bwd_blk_rows(self.num_nodes, self.num_rels, routes.get("deterministic", "0") == "1", self.device, diag4,
                                    sparse)
        return self._plan("bwd", rows) if rows else None

    def fwd_blk_plan(self):
        """FORWARD plan of tall tiles (one per workgroup, up to 1000 rows) for the block-tile forward kernel (rgcn_spmm_blk_f32), or None
        when it does not apply (small graph, forced tile height, no run pointers)"""
        if routes.is_set("tile_rows"):
            return None
        rows = _native.spmm_blk_rows(self.num_nodes, self.device)
        if not rows:
            return None
        plan = self._plan("fwd", rows)
        return plan if plan.run_ptr is not None and plan.n_src < (1 << 26) else None

    def win_plan(self, kind, rows=None):
        """plan of tall workgroup-owned tiles in SOFT-WINDOW order (_native.build_softwin_plan: buckets sorted by source, a tile's chunks
        ordered by first source) for the block-tile kernels, or None where it does not apply or does not pay: static graphs with a
        device-side message list, large enough for one tile per workgroup, with at least _SOFTWIN_MIN_CHUNKS (4) chunks per
        (tile, relation) bucket on average -- the span of a chunk's sources is 1 / that of the table -- and without forced tile heights,
        RGCN_SOFTWIN=0 or RGCN_DETERMINISTIC=1 (the tile is summed in arrival order)."""
        if routes.get("softwin", "auto") == "0" or routes.get("deterministic", "0") == "1" or routes.is_set("tile_rows") or \
                self._dev is None or self.sync_free or getattr(self, "per_call", False):
            return None
        if rows is None:
            rows = _native.bwd_own_rows(self.num_nodes, self.device) if kind == "bwd_own" else _native.spmm_blk_rows(self.num_nodes, self.device)
        if not rows or self.num_nodes >= (1 << 26):
            return None
        n_tiles = -(-self.num_nodes // rows)
        if routes.get("softwin", "auto") != "1" and \
                self.num_messages < _SOFTWIN_MIN_CHUNKS * 16 * n_tiles * self.num_rels:
            return None
        key = ("win", kind, rows)
        if key not in self._plans:
            s, p, o, val, alive = self._dev
            dst, src = (s, o) if kind == "fwd" else (o, s)
            own = {}
            if kind == "bwd_own":     # transposed plan, a tile's chunks grouped by the wave that owns their relation (rgcn_bwd_own_f32)
                nw, per_wave, max_rows = _native.bwd_own_geometry()
                if rows > max_rows or self.num_rels > nw * per_wave:
                    self._plans[key] = None
                    return None
                own = {"own_waves": nw, "own_per_wave": per_wave}
            plan = _native.build_softwin_plan(dst, src, p, val, alive, self.num_nodes, self.num_nodes, self.num_rels, rows, **own)
            # a wave that owns far more messages than the others holds every tile back: such graphs keep the block-tile kernel
            if plan is not None and own and plan.own_balance > _OWN_MAX_IMBALANCE:
                plan = None
            self._plans[key] = plan
        return self._plans[key]

    def wgt_plan(self):
        """relation-major (single tile): long runs per relation for the weight gradient"""
        return self._plan("fwd", max(self.num_nodes, 1), int(routes.get("wgrad_item_chunks", "64")))

    def csr(self, kind, need_slot=False):
        """destination-major ("fwd": rows = subjects) or source-major ("bwd") CSR for the basis kernels.  need_slot: with the
        msg_slot array (input message -> CSR position) the two-pass routes read; per-call graphs build it only on demand"""
        key = ("csr", kind)
        if key in self._plans and need_slot and self._plans[key].msg_slot is None:
            del self._plans[key]
        if key not in self._plans:
            if self._dev is None:
                raise RuntimeError("the basis-aggregation path needs the device-side graph build")
            s, p, o, val, alive = self._dev
            if getattr(self, "per_call", False) and not need_slot and ("csr", "fwd") not in self._plans \
                    and ("csr", "bwd") not in self._plans:
                # per-call graphs: both directions at once (a training step walks both), five launches for the pair
                self._plans[("csr", "fwd")], self._plans[("csr", "bwd")] = \
                    _native.build_csr_pair_device(s, o, p, val, alive, self.num_nodes)
                return self._plans[key]
            dst, src = (s, o) if kind == "fwd" else (o, s)
            self._plans[key] = _native.build_csr_device(dst, src, p, val, alive, self.num_nodes, sync_free=self.sync_free,
                                                        want_slot=need_slot or not getattr(self, "per_call", False))
            self._plans[key].per_call = getattr(self, "per_call", False)
            self._plans[key].num_rels = self.num_rels
        return self._plans[key]

    def fbasis_plan(self):
        """source-major message list + the permutations back to destination / relation order (featureless basis layer)"""
        if "fbasis" not in self._plans:
            self._plans["fbasis"] = _native.build_fbasis_plan(self.csr("bwd"), self.csr("fwd"), self.num_nodes, self.num_rels)
        return self._plans["fbasis"]

    def max_degree(self):
        """largest number of messages received or sent by one node (cached; device graphs only)"""
        if self.sync_free:
            return 0                 # unknown without a read-back; callers treat the graph as hub-free
        if "maxdeg" not in self._plans:
            m = 0
            for kind in ("fwd", "bwd"):
                rp = self.csr(kind).rowptr
                m = max(m, int((rp[1:] - rp[:-1]).max().item()) if rp.numel() > 1 else 0)
            self._plans["maxdeg"] = m
        return self._plans["maxdeg"]

    def scatter_plan(self, kind, item_chunks=64):
        """relation-major plan whose slots know their position in the destination-major CSR (sparse-bucket path; with
        item_chunks = 8 its work items are the 128-row blocks of the wide-layer gather-GEMM)"""
        key = ("scatter", kind, item_chunks)
        if key not in self._plans:
            if self._dev is None:
                raise RuntimeError("the relation-major two-pass paths need the device-side graph build")
            s, p, o, val, alive = self._dev
            dst, src = (s, o) if kind == "fwd" else (o, s)
            csr = self.csr(kind, need_slot=True)
            if csr.n_messages is None:
                csr.n_messages = int(csr.rowptr[-1].item())
            N, R = self.num_nodes, self.num_rels
            self._plans[key] = _native.build_plan_device(dst, src, p, val, alive, N, N, R, max(N, 1), self.num_messages,
                                                         item_chunks, aux=csr.msg_slot)
        return self._plans[key]

    def selfloop_edges(self, self_rel):
        """(s, o, val) device tensors of the messages of one relation (used by the LP block-dropout branch)."""
        if self._dev is not None:
            s, p, o, val, alive = self._dev
            m = p == self_rel
            if alive is not None:
                m = m & (alive != 0)
            return s[m].long(), o[m].long(), val[m]
        m = self._p == self_rel
        dev = self.device
        return (torch.from_numpy(self._s[m].astype(np.int64)).to(dev), torch.from_numpy(self._o[m].astype(np.int64)).to(dev),
                torch.from_numpy(self._val[m]).to(dev))


def _device_build_enabled():
    return routes.get("graph_build", "device") == "device"


def node_order(s, o, num_nodes, how):
    """Locality relabelling of the nodes (SURVEY 8d / DESIGN: every gathered 64-byte row costs a 128-byte line, and a line is only
    shared when its two rows are wanted close together in time).  s, o: int arrays of the messages' endpoints.  Returns
    perm (int64 [N]): new id of every old id.
      "degree"  nodes by descending message count (in + out): hub rows become neighbours, a handful of lines carries most gathers
                and stays in L2 (pays on skewed graphs; on a uniform graph every row is as cold as every other)
      "rcm"     reverse Cuthill-McKee on the symmetrised adjacency (scipy): neighbours get nearby ids, so a destination tile's
                sources cluster (pays when the graph has community structure / small separators; a uniform random graph has none)
      "bfs"     breadth-first order from the highest-degree node"""
    s = np.asarray(s, dtype=np.int64)
    o = np.asarray(o, dtype=np.int64)
    if how == "degree":
        deg = np.bincount(s, minlength=num_nodes) + np.bincount(o, minlength=num_nodes)
        order = np.argsort(-deg, kind="stable")                      # order[new] = old
    elif how in ("rcm", "bfs"):
        import scipy.sparse as sp
        from scipy.sparse import csgraph
        keep = s != o
        A = sp.coo_matrix((np.ones(int(keep.sum()), np.int8), (s[keep], o[keep])), shape=(num_nodes, num_nodes)).tocsr()
        A = ((A + A.T) > 0).astype(np.int8).tocsr()
        if how == "rcm":
            order = np.asarray(csgraph.reverse_cuthill_mckee(A, symmetric_mode=True), dtype=np.int64)
        else:
            deg = np.asarray(A.sum(axis=1)).ravel()
            seen = np.zeros(num_nodes, bool)
            parts = []
            for start in np.argsort(-deg, kind="stable"):
                if seen[start]:
                    continue
                comp = csgraph.breadth_first_order(A, int(start), directed=False, return_predecessors=False)
                seen[comp] = True
                parts.append(comp)
                if seen.all():
                    break
            order = np.concatenate(parts).astype(np.int64)
    else:
        raise NotImplementedError(f"node order {how!r} (degree, rcm, bfs)")
    perm = np.empty(num_nodes, dtype=np.int64)
    perm[order] = np.arange(num_nodes, dtype=np.int64)
    return perm


_AUTO_MIN_NODES = 500_000        # below: the feature matrix (64 B rows) sits in L2 / the first MB of the Infinity Cache anyway
_AUTO_HUB_SHARE = 0.30           # the busiest 1 % of the nodes touch at least this share of the messages' endpoints


def _auto_order(s, o, num_nodes):
    """relabel="auto" (the default): "degree" for LARGE SKEWED static graphs, "none" otherwise.  Measured (profiles/r03_locality.json,
    r04_locality_auto.json): on a uniform graph (S1) no order helps -- every row is as cold as every other --, on an AM-sized graph
    with Zipf(0.9) endpoints the degree order makes the hubs' rows neighbours (a handful of lines carries most gathers and stays in
    L2): -13 % kernel time; rcm / bfs cost seconds of host time per graph and gain less.  The test is one device-side sort of the
    node degrees, once per graph."""
    if num_nodes < _AUTO_MIN_NODES or s.numel() == 0:
        return "none"
    deg = torch.bincount(s.long(), minlength=num_nodes) + torch.bincount(o.long(), minlength=num_nodes)
    top = max(1, num_nodes // 100)
    share = deg.topk(top).values.sum().item() / max(1, int(deg.sum().item()))
    return "degree" if share >= _AUTO_HUB_SHARE else "none"


def graph_from_nc_triples(triples_plus, num_nodes, num_rels, vertical, device, relabel=None):
    """NC layer: n = int((M - N) / 2), i = N  (torch_rgcn/layers.py:235-236, :269-271).
    relabel ("degree" / "rcm" / "bfs" / "none" / "auto"; default: route `relabel`, else "auto" = degree order for large skewed
    graphs, _auto_order): the plans are built on locality-relabelled node ids
    (node_order); graph.perm[old] = new, graph.inv[new] = old on the device -- the layer reads its features through inv and
    returns its output through perm, so callers never see the relabelling.  The normalisation is computed on the original
    ids first (it only counts equal (relation, node) keys: invariant under a relabelling)."""
    relabel = relabel if relabel is not None else (routes.get("relabel") or "auto")
    if _device_build_enabled() and num_nodes * num_rels < _MAX_CELLS:
        t = torch.as_tensor(triples_plus, dtype=torch.long).reshape(-1, 3).to(device)
        M = t.shape[0]
        n_swap = int((M - num_nodes) / 2)
        assert vertical or (n_swap >= 0 and 2 * n_swap + num_nodes == M), \
            f"edge_norm: horizontal swap needs 2n+i == M (n={n_swap} i={num_nodes} M={M})"
        s, p, o, err = _native.dev_split_triples(t, num_nodes, num_rels)
        _native.dev_check_err(err, "stack_matrices")
        val = _native.dev_edge_norm(s, p, o, None, num_nodes, num_rels, vertical, max(n_swap, 0))
        perm = None
        if relabel == "auto":
            relabel = _auto_order(s, o, num_nodes)
        if relabel and relabel != "none":
            perm = torch.from_numpy(node_order(s.cpu().numpy(), o.cpu().numpy(), num_nodes, relabel)).to(device)
            p32 = perm.to(torch.int32)
            s, o = p32[s.long()].contiguous(), p32[o.long()].contiguous()
        g = RelGraph.on_device(s, p, o, val, None, M, num_nodes, num_rels)
        if perm is not None:
            g.perm = perm
            g.inv = torch.empty_like(perm)
            g.inv[perm] = torch.arange(num_nodes, device=perm.device)
        return g
    tp = triples_plus.detach().cpu().numpy() if torch.is_tensor(triples_plus) else np.asarray(triples_plus)
    tp = np.ascontiguousarray(tp, dtype=np.int64).reshape(-1, 3)
    M = tp.shape[0]
    val = _native.edge_norm_host(tp, num_nodes, num_rels, vertical, int((M - num_nodes) / 2), num_nodes)
    return RelGraph(tp, val, num_nodes, num_rels, device)


def graph_from_lp_triples(triples, num_nodes, num_rels, vertical, keep_mask, device):
    """LP layer: [T | inv | T | kept self loops], n = E, i = E + #kept (layers.py:481-487, :505-510)."""
    if _device_build_enabled() and num_nodes * num_rels < _MAX_CELLS:
        t = torch.as_tensor(triples, dtype=torch.long).reshape(-1, 3).to(device)
        E = t.shape[0]
        s, p, o, alive, err = _native.dev_lp_expand(t, num_nodes, (num_rels - 1) // 2, keep_mask)
        _native.dev_check_err(err, "stack_matrices")
        val = _native.dev_edge_norm(s, p, o, alive, num_nodes, num_rels, vertical, E)
        sync_free = _native._deferred_mode()
        # live messages: the upper bound (all self loops kept) -- it only feeds sizing heuristics, not worth a read-back
        n_live = 3 * E + num_nodes
        g = RelGraph.on_device(s, p, o, val, alive, n_live, num_nodes, num_rels)
        g.sync_free = sync_free
        g.per_call = True
        return g
    t = triples.detach().cpu().numpy() if torch.is_tensor(triples) else np.asarray(triples)
    t = np.ascontiguousarray(t, dtype=np.int64).reshape(-1, 3)
    R0 = (num_rels - 1) // 2
    keep = None if keep_mask is None else keep_mask.detach().cpu().numpy().astype(np.uint8)
    tp, n_self = _native.lp_augment_host(t, num_nodes, R0, keep)
    val = _native.edge_norm_host(tp, num_nodes, num_rels, vertical, t.shape[0], n_self)
    return RelGraph(tp, val, num_nodes, num_rels, device)
