"""ctypes binding of librgcn_hip.so (C ABI in include/rgcn_hip.h).

There is deliberately NO fallback: if the shared library is missing or a kernel
launch fails, the caller gets an exception.  The library is built in-tree by
`__graft_entry__.build()` / `make -C torch-rgcn_amd/csrc`.
"""
import ctypes
import threading
import os

import numpy as np
import torch

from . import routes

# RGCN_HIP_LIB: tools/ only -- the ablation build (make -C csrc abl -> lib/librgcn_hip_abl.so), whose kernels can be told to skip work
_LIB_PATH = os.environ.get("RGCN_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "librgcn_hip.so")
_lib = None

OK, EINVAL, ENOMEM, ERANGE, EHIP, EUNSUPPORTED = range(6)
CHUNK = 16

c_i64, c_i32, c_int, c_void_p = ctypes.c_int64, ctypes.c_int32, ctypes.c_int, ctypes.c_void_p


class NativeLibraryError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise NativeLibraryError(
                f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). torch_rgcn has no CPU/eager fallback.")
        L = ctypes.CDLL(_LIB_PATH)
        L.rgcn_version.restype = ctypes.c_char_p
        L.rgcn_csrc_sha.restype = ctypes.c_char_p
        L.rgcn_bwd_fused_scratch_floats.restype = ctypes.c_int64
        L.rgcn_bwd_lean_slot_bytes.restype = ctypes.c_int64
        L.rgcn_bwd_blk_rec_bytes.restype = ctypes.c_int64
        L.rgcn_softwin_tmp_bytes.restype = ctypes.c_int64
        L.rgcn_colsum_scratch_floats.restype = ctypes.c_int64
        L.rgcn_gemm_scratch_floats.restype = ctypes.c_int64
        L.rgcn_basis_sum_workspace_bytes.restype = ctypes.c_int64
        L.rgcn_last_error.restype = ctypes.c_char_p
        _lib = L
        defaults = {}

        def sink(name, value):      # routes -> the library's option table (None: back to the library's default)
            if name not in defaults:
                cur = ctypes.c_int32(0)
                L.rgcn_get_option(name.encode(), ctypes.byref(cur))
                defaults[name] = cur.value
            v = defaults[name] if value is None else value
            if L.rgcn_set_option(name.encode(), c_i32(v)) != OK:
                raise NativeLibraryError(f"route {name} = {v}: {L.rgcn_last_error().decode()}")
        routes._native_sink = sink
        routes.push_native()
    return _lib


def version():
    return lib().rgcn_version().decode()


def csrc_sha():
    """identity of the kernel sources the loaded library was built from (rgcn_csrc_sha)"""
    return lib().rgcn_csrc_sha().decode()


def _check(rc, what):
    if rc == OK:
        return
    msg = lib().rgcn_last_error().decode()
    if rc in (ERANGE, EINVAL):
        # the reference signals these with Python asserts (utils.py:148,162-164; layers.py:282-284,303)
        raise AssertionError(f"{what}: {msg}")
    if rc == ENOMEM:
        raise MemoryError(f"{what}: {msg}")
    raise NativeLibraryError(f"{what}: {msg} (code {rc})")


def _np(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _hp(a):
    """host pointer of a numpy array (or None)"""
    return None if a is None else c_void_p(a.ctypes.data)


def _dp(t):
    """device pointer of a torch tensor (or None)"""
    return None if t is None else c_void_p(t.data_ptr())


# ----------------------------------------------------------------------------- per-kernel timing
# HIP events on the stream the kernels are launched on (torch's current stream), recorded
# around every launch while profiling is on; bench.py uses this for the roofline figures.
_PROF = None
_PROF_ON = True


def profile_start():
    global _PROF, _PROF_ON
    _PROF, _PROF_ON = {}, True


def profile_enable(on):
    """timers on / off without ending the profile: a HIP event pair around a launch costs ~9 us of GPU time (the launches before and after
    it cannot overlap their tails and heads), so a caller that times a whole loop as well samples the per-launch timers -- bench.py brackets
    the launches of every fourth step"""
    global _PROF_ON
    _PROF_ON = bool(on)


def profile_stop():
    """-> {kernel name: [ms per launch, ...]} (synchronises)."""
    global _PROF
    rec, _PROF = _PROF or {}, None
    torch.cuda.synchronize()
    return {k: [a.elapsed_time(b) for a, b in v] for k, v in rec.items()}


class _Timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.a.record()

    def __exit__(self, *exc):
        b = torch.cuda.Event(enable_timing=True)
        b.record()
        if _PROF is not None:
            _PROF.setdefault(self.name, []).append((self.a, b))


def _timed(name):
    # (no events inside a stream capture: recording them there is an error -- a captured step is timed as a whole, by its replay)
    return _NOCTX if (_PROF is None or not _PROF_ON or torch.cuda.is_current_stream_capturing()) else _Timed(name)


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device):
    """the current HIP stream of `device` as a void* (torch's raw-stream query when it exists: no Stream object per launch)"""
    if _RAW_STREAM is not None:
        idx = device.index
        return c_void_p(_RAW_STREAM(torch.cuda.current_device() if idx is None else idx))
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _NoCtx:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NOCTX = _NoCtx()


def _on(device):
    """`with _on(dev):` = torch.cuda.device(dev), skipped when dev is already the current device (one process per GPU: always) --
    a step of the LP layer makes ~60 native calls, the guard object and its two device switches were a measurable part of it"""
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NOCTX
    return torch.cuda.device(device)


# ----------------------------------------------------------------------------- host side

def add_inverse_and_self_host(triples, num_nodes, num_rels):
    t = _np(triples, np.int64).reshape(-1, 3)
    out = np.empty((2 * t.shape[0] + num_nodes, 3), np.int64)
    _check(lib().rgcn_add_inverse_and_self_host(_hp(t), c_i64(t.shape[0]), c_i64(num_nodes), c_i64(num_rels),
                                                _hp(out)), "add_inverse_and_self")
    return out


def lp_augment_host(triples, num_nodes, num_rels, keep=None):
    t = _np(triples, np.int64).reshape(-1, 3)
    E = t.shape[0]
    out = np.empty((3 * E + num_nodes, 3), np.int64)
    k = None if keep is None else _np(keep, np.uint8)
    M, ns = c_i64(0), c_i64(0)
    _check(lib().rgcn_lp_augment_host(_hp(t), c_i64(E), c_i64(num_nodes), c_i64(num_rels), _hp(k), _hp(out),
                                      ctypes.byref(M), ctypes.byref(ns)), "lp_augment")
    return out[:M.value], ns.value


def edge_norm_host(triples_plus, num_nodes, num_rels, vertical, n_swap, i_tail):
    t = _np(triples_plus, np.int64).reshape(-1, 3)
    val = np.empty(t.shape[0], np.float32)
    _check(lib().rgcn_edge_norm_host(_hp(t), c_i64(t.shape[0]), c_i64(num_nodes), c_i64(num_rels),
                                     c_int(int(bool(vertical))), c_i64(n_swap), c_i64(i_tail), _hp(val)),
           "edge_norm")
    return val


def synthetic_triples_host(num_nodes, num_rels, num_edges, seed=0):
    out = np.empty((num_edges, 3), np.int64)
    _check(lib().rgcn_synthetic_triples_host(c_i64(num_nodes), c_i64(num_rels), c_i64(num_edges),
                                             ctypes.c_uint64(seed), _hp(out)), "synthetic_triples")
    return out


def edge_neighborhood_host(triples, num_nodes, sample_size, seed):
    """indices of `sample_size` triples drawn by edge-neighbourhood sampling (utils/misc.py:125-172)"""
    triples = np.ascontiguousarray(triples, np.int64)
    out = np.empty(sample_size, np.int64)
    _check(lib().rgcn_edge_neighborhood_host(_hp(triples), c_i64(triples.shape[0]), c_i64(num_nodes),
                                             c_i64(sample_size), ctypes.c_uint64(seed), _hp(out)), "edge_neighborhood")
    return out


class HostPlan:
    """Relation-tile plan as numpy arrays (see rgcn_plan_fill_host)."""
    __slots__ = ("src", "dst", "val", "perm", "chunk_rel", "tile_ptr", "items", "run_ptr", "pack", "units", "units_host", "_cache", "n_units", "n_split", "max_run_chunks", "n_dst", "n_src", "num_rels",
                 "tile_rows", "n_tiles", "n_chunks", "m_pad", "n_items", "n_messages")


def build_plan_host(dst, src, rel, val, n_dst, n_src, num_rels, tile_rows, max_item_chunks=64, want_perm=False,
                    want_runs=False, want_pack=False, max_unit_chunks=256):
    dst = _np(dst, np.int32)
    src = _np(src, np.int32)
    rel = _np(rel, np.int32)
    val = _np(val, np.float32)
    M = dst.shape[0]
    assert src.shape[0] == M and rel.shape[0] == M and val.shape[0] == M
    m_pad, n_chunks, n_tiles, n_items = c_i64(0), c_i64(0), c_i64(0), c_i64(0)
    L = lib()
    _check(L.rgcn_plan_count_host(_hp(dst), _hp(rel), c_i64(M), c_i64(n_dst), c_i32(num_rels), c_i32(tile_rows),
                                  c_i32(max_item_chunks), ctypes.byref(m_pad), ctypes.byref(n_chunks),
                                  ctypes.byref(n_tiles), ctypes.byref(n_items)), "plan_count")
    p = HostPlan()
    p.n_dst, p.n_src, p.num_rels, p.tile_rows = n_dst, n_src, num_rels, tile_rows
    p.n_tiles, p.n_chunks, p.m_pad, p.n_items, p.n_messages = n_tiles.value, n_chunks.value, m_pad.value, n_items.value, M
    p.src = np.empty(max(p.m_pad, 1), np.int32)
    p.dst = np.empty(max(p.m_pad, 1), np.int32)
    p.val = np.empty(max(p.m_pad, 1), np.float32)
    p.perm = np.empty(max(p.m_pad, 1), np.int32) if want_perm else None
    p.chunk_rel = np.empty(max(p.n_chunks, 1), np.int32)
    p.tile_ptr = np.zeros(p.n_tiles + 1, np.int32)
    p.items = np.empty((max(p.n_items, 1), 2), np.int32)
    p.run_ptr = np.zeros(max(p.n_tiles, 1) * (num_rels + 1), np.int32) if want_runs else None
    can_pack = want_pack and n_src < (1 << 24) and tile_rows <= 255
    p.pack = np.empty((max(p.m_pad, 1), 2), np.int32) if can_pack else None
    _check(L.rgcn_plan_fill_host(_hp(dst), _hp(src), _hp(rel), _hp(val), c_i64(M), c_i64(n_dst), c_i64(n_src),
                                 c_i32(num_rels), c_i32(tile_rows), c_i32(max_item_chunks), _hp(p.src), _hp(p.dst),
                                 _hp(p.val), _hp(p.perm), _hp(p.chunk_rel), _hp(p.tile_ptr), _hp(p.items),
                                 _hp(p.run_ptr), _hp(p.pack)), "plan_fill")
    # work units of the tile kernels (hub tiles are cut into pieces)
    nu, ns = c_i64(0), c_i64(0)
    _check(L.rgcn_plan_units_host(_hp(p.tile_ptr), c_i64(p.n_tiles), c_i32(max_unit_chunks), None, ctypes.byref(nu),
                                  ctypes.byref(ns)), "plan_units")
    p.n_units, p.n_split = nu.value, ns.value
    p.units = np.zeros((max(p.n_units, 1), 4), np.int32)
    p.units_host = p.units
    p._cache = {}
    _check(L.rgcn_plan_units_host(_hp(p.tile_ptr), c_i64(p.n_tiles), c_i32(max_unit_chunks), _hp(p.units),
                                  ctypes.byref(nu), ctypes.byref(ns)), "plan_units")
    if p.n_chunks:
        cr = p.chunk_rel[:p.n_chunks]
        edges = np.flatnonzero(np.diff(cr) != 0)
        runs = np.diff(np.concatenate(([-1], edges, [p.n_chunks - 1])))
        p.max_run_chunks = int(runs.max())
    else:
        p.max_run_chunks = 0
    return p


# ----------------------------------------------------------------------------- device side

class DevicePlan:
    """HostPlan uploaded to one GPU (int32 / fp32 tensors)."""

    def __init__(self, hp, device):
        self.device = torch.device(device)
        up = lambda a: torch.from_numpy(a).to(self.device, non_blocking=False)
        self.src, self.dst, self.val = up(hp.src), up(hp.dst), up(hp.val)
        self.chunk_rel, self.tile_ptr, self.items = up(hp.chunk_rel), up(hp.tile_ptr), up(hp.items)
        self.run_ptr = None if hp.run_ptr is None else up(hp.run_ptr)
        self.pack = None if hp.pack is None else up(hp.pack)
        self.units = up(hp.units)
        self.units_host = hp.units
        for k in ("n_dst", "n_src", "num_rels", "tile_rows", "n_tiles", "n_chunks", "m_pad", "n_items", "n_messages",
                  "n_units", "n_split", "max_run_chunks"):
            setattr(self, k, getattr(hp, k))

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in (self.src, self.dst, self.val, self.chunk_rel,
                                                           self.tile_ptr, self.items))


# ----------------------------------------------------------------------------- device-side graph build

def _i32(n, device):
    return torch.empty(max(int(n), 1), dtype=torch.int32, device=device)


# Device-side range checks leave a flag in device memory.  By default the flag is read back at once (one 4-byte
# device -> host copy, i.e. a synchronisation) and a bad index raises where the reference raises.  A training loop that
# must not synchronise (hipGraph capture, torch.cuda.set_sync_debug_mode) sets RGCN_DEFERRED_CHECKS=1: the flags are then
# queued and looked at on later calls once the GPU has passed them (event query, no wait) or in check_deferred_errors().
_DEFERRED = []
_PINNED_FREE = []        # recycled 1-int pinned host buffers (allocating pinned memory per call would stall the stream)


def _deferred_mode():
    return routes.get("deferred_checks", "0") == "1"


def dev_check_err(err_flag, what, exc=AssertionError):
    if not _deferred_mode():
        if int(err_flag.item()):
            raise exc(f"{what}: node or relation index out of range")
        return
    if torch.cuda.is_current_stream_capturing():
        return                                   # inside a hipGraph capture nothing can be read back: validated before capture
    host = _PINNED_FREE.pop() if _PINNED_FREE else torch.empty(1, dtype=torch.int32, pin_memory=True)
    host.copy_(err_flag[:1], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _DEFERRED.append((ev, host, what, exc))
    check_deferred_errors(wait=False)


def check_deferred_errors(wait=True):
    """raise for any queued device-side range check that failed; wait=False only looks at checks the GPU has finished"""
    keep = []
    err = None
    for ev, host, what, exc in _DEFERRED:
        if wait:
            ev.synchronize()
        if ev.query():
            if int(host[0]) and err is None:
                err = exc(f"{what}: node or relation index out of range")
            _PINNED_FREE.append(host)
        else:
            keep.append((ev, host, what, exc))
    _DEFERRED[:] = keep
    if err is not None:
        raise err


def dev_split_triples(triples_plus, num_nodes, num_rels):
    """int64 [M,3] device tensor -> (s, p, o) int32 device tensors (range-checked)"""
    t = triples_plus.contiguous()
    M, dev = t.shape[0], t.device
    s, p, o, err = _i32(M, dev), _i32(M, dev), _i32(M, dev), _i32(1, dev)
    with _on(dev):
        _check(lib().rgcn_dev_split_triples(_dp(t), c_i64(M), c_i64(num_nodes), c_i32(num_rels), _dp(s), _dp(p), _dp(o),
                                            _dp(err), _stream(dev)), "dev_split_triples")
    return s[:M], p[:M], o[:M], err


def dev_lp_expand(triples, num_nodes, num_rels0, keep):
    """[T | inv | T | self loops] on the device; keep: uint8/bool [N] device tensor or None"""
    t = triples.contiguous()
    E, dev = t.shape[0], t.device
    M = 3 * E + num_nodes
    s, p, o, err = _i32(M, dev), _i32(M, dev), _i32(M, dev), _i32(1, dev)
    alive = torch.empty(M, dtype=torch.uint8, device=dev)
    k = None if keep is None else keep.to(torch.uint8).contiguous()
    with _on(dev):
        _check(lib().rgcn_dev_lp_expand(_dp(t), c_i64(E), c_i64(num_nodes), c_i32(num_rels0), _dp(k), _dp(s), _dp(p),
                                        _dp(o), _dp(alive), _dp(err), _stream(dev)), "dev_lp_expand")
    return s, p, o, alive, err


def dev_edge_norm(s, p, o, alive, num_nodes, num_rels, vertical, n_swap):
    M, dev = s.shape[0], s.device
    table = _i32(num_nodes * num_rels, dev)
    val = torch.empty(max(M, 1), dtype=torch.float32, device=dev)
    with _on(dev):
        _check(lib().rgcn_dev_edge_norm(_dp(s), _dp(p), _dp(o), _dp(alive), c_i64(M), c_i64(num_nodes), c_i32(num_rels),
                                        c_int(int(bool(vertical))), c_i64(n_swap), _dp(table), _dp(val), _stream(dev)),
               "dev_edge_norm")
    return val[:M]


class BuiltPlan:
    """Same fields as DevicePlan, produced on the device."""


def build_plan_device(dst, src, rel, val, alive, n_dst, n_src, num_rels, tile_rows, n_live, max_item_chunks=64,
                      want_runs=False, want_pack=False, max_unit_chunks=256, aux=None, sync_free=False):
    """sync_free: size everything by an upper bound and finish the plan on the device (rgcn_dev_plan_finish_nosync): no
    device -> host read at all -- for the per-call graphs of the LP layer inside a step that must not synchronise.  Costs
    memory (up to 16 slots per message) and gives one work unit per tile (no hub splitting), so static graphs keep the
    exact path."""
    dev = dst.device
    M = dst.shape[0]
    n_tiles = (n_dst + tile_rows - 1) // tile_rows
    nbk = n_tiles * num_rels
    cells = _i32(nbk * tile_rows, dev)
    tall = tile_rows > 1024
    cells_tmp = _i32(nbk * tile_rows, dev) if tall else None
    bucket_cnt, bucket_base = _i32(nbk, dev), _i32(nbk + 1, dev)
    scan_tmp = _i32(max(nbk, nbk * tile_rows if tall else 0) // 1024 + 4, dev)
    L = lib()
    with _on(dev):
        _check(L.rgcn_dev_plan_count(_dp(dst), _dp(rel), _dp(alive), c_i64(M), c_i64(n_dst), c_i32(num_rels),
                                     c_i32(tile_rows), _dp(cells), _dp(bucket_cnt), _dp(bucket_base), _dp(scan_tmp),
                                     _dp(cells_tmp), _stream(dev)), "dev_plan_count")
    if sync_free:
        m_pad = (M + 15 * min(nbk, M) + CHUNK - 1) // CHUNK * CHUNK      # every non-empty bucket pads by < 16 slots
    else:
        m_pad = int(bucket_base[nbk].item())        # the one host round trip: output sizes
    p = BuiltPlan()
    p.device = dev
    p.n_dst, p.n_src, p.num_rels, p.tile_rows = n_dst, n_src, num_rels, tile_rows
    p.n_tiles, p.m_pad, p.n_chunks, p.n_messages = n_tiles, m_pad, m_pad // CHUNK, int(n_live)
    p.src, p.dst = _i32(m_pad, dev), _i32(m_pad, dev)
    p.val = torch.empty(max(m_pad, 1), dtype=torch.float32, device=dev)
    can_pack = want_pack and n_src < (1 << 24) and tile_rows <= 255
    p.pack = torch.empty((max(m_pad, 1), 2), dtype=torch.int32, device=dev) if can_pack else None
    p.chunk_rel = _i32(p.n_chunks, dev)
    p.tile_ptr = _i32(n_tiles + 1, dev)
    p.run_ptr = _i32(n_tiles * (num_rels + 1), dev) if want_runs else None
    p.aux = _i32(m_pad, dev) if aux is not None else None
    with _on(dev):
        _check(L.rgcn_dev_plan_fill(_dp(dst), _dp(src), _dp(rel), _dp(val), _dp(alive), c_i64(M), c_i64(n_dst),
                                    c_i64(n_src), c_i32(num_rels), c_i32(tile_rows), _dp(cells), _dp(bucket_cnt),
                                    _dp(bucket_base), _dp(p.src), _dp(p.dst), _dp(p.val), _dp(p.pack), _dp(p.chunk_rel),
                                    _dp(p.tile_ptr), _dp(p.run_ptr), _dp(aux), _dp(p.aux), None, c_i64(p.n_chunks),
                                    _stream(dev)), "dev_plan_fill")
    if sync_free:
        p.n_units, p.n_split, p.units_host = n_tiles, 0, None
        p.units = torch.empty((max(n_tiles, 1), 4), dtype=torch.int32, device=dev)
        p.max_run_chunks = 1 << 30                  # unknown on the host: callers must not rely on short runs
        if n_tiles == 1:
            p.n_items = p.n_chunks // max_item_chunks + num_rels
            p.items = torch.empty((max(p.n_items, 1), 2), dtype=torch.int32, device=dev)
        else:
            p.n_items, p.items = 0, None
        with _on(dev):
            _check(L.rgcn_dev_plan_finish_nosync(_dp(bucket_base), c_i64(n_tiles), c_i32(num_rels), c_i64(m_pad), _dp(p.src),
                                                 _dp(p.dst), _dp(p.val), _dp(p.pack), _dp(p.aux), _dp(p.tile_ptr), _dp(p.units),
                                                 _dp(p.items), c_i64(max(p.n_items, 1)), c_i32(max_item_chunks), _stream(dev)),
                   "dev_plan_finish_nosync")
        if p.items is None:
            p.items = _i32(2, dev).view(1, 2)
        return p
    # work units (hub tiles split) and the relation-major work list are tiny: host side
    tp_host = p.tile_ptr[:n_tiles + 1].cpu().numpy()
    nu, ns = c_i64(0), c_i64(0)
    _check(L.rgcn_plan_units_host(_hp(tp_host), c_i64(n_tiles), c_i32(max_unit_chunks), None, ctypes.byref(nu),
                                  ctypes.byref(ns)), "plan_units")
    units = np.zeros((max(nu.value, 1), 4), np.int32)
    _check(L.rgcn_plan_units_host(_hp(tp_host), c_i64(n_tiles), c_i32(max_unit_chunks), _hp(units), ctypes.byref(nu),
                                  ctypes.byref(ns)), "plan_units")
    p.n_units, p.n_split = nu.value, ns.value
    p.units_host = units
    p.units = torch.from_numpy(units).to(dev)
    p.max_run_chunks = int((int(bucket_cnt[:nbk].max().item()) + CHUNK - 1) // CHUNK) if nbk else 0
    if n_tiles == 1:   # relation-major plan: work items = chunk ranges of one relation, at most max_item_chunks long
        base = (bucket_base[:nbk + 1].cpu().numpy() // CHUNK).astype(np.int64)
        items = [(c, min(c + max_item_chunks, base[r + 1])) for r in range(num_rels)
                 for c in range(base[r], base[r + 1], max_item_chunks)]
        p.n_items = len(items)
        p.items = torch.tensor(items if items else [[0, 0]], dtype=torch.int32, device=dev)
    else:
        p.n_items, p.items = 0, _i32(2, dev).view(1, 2)
    return p


def own_relations(counts, n_waves, per_wave):
    """relation -> owner waves for the relation-owner backward (rgcn_bwd_own_f32).  The unit of ownership is a PART of a relation: a
    relation much larger than a wave's fair share (the self-loop relation; the one relation of a graph with few of them) is cut into parts,
    part q takes the relation's chunks q, q + parts, ... of every (tile, relation) bucket; every part has its own accumulator and they all
    leave the CU into the same dW_r.  Parts are split greedily (largest part first) until no part exceeds half a wave's fair share or the
    n_waves * per_wave slots are used up, then packed onto the waves by longest-processing-time, at most per_wave per wave.
    -> (parts[R], unit_base[R], owner[U], local[U], unit_rel[n_waves * per_wave] (-1: unused), max load / mean load), U = sum(parts);
    None when there are more relations than slots."""
    import heapq
    counts = np.asarray(counts, dtype=np.int64)
    R = len(counts)
    slots = n_waves * per_wave
    if R > slots:
        return None
    parts = np.ones(R, np.int64)
    target = max(1.0, float(counts.sum()) / n_waves)
    heap = [(-float(counts[r]), r) for r in range(R)]
    heapq.heapify(heap)
    units = R
    while units < slots and heap:
        size, r = heap[0]
        if -size <= 0.5 * target:
            break
        parts[r] += 1
        units += 1
        heapq.heapreplace(heap, (-float(counts[r]) / parts[r], r))
    unit_base = np.cumsum(parts) - parts
    U = int(parts.sum())
    usize = np.concatenate([np.full(parts[r], counts[r] / parts[r]) for r in range(R)]) if R else np.zeros(0)
    urel = np.repeat(np.arange(R), parts)
    load, used = [0.0] * n_waves, [0] * n_waves
    owner, local = np.zeros(U, np.int64), np.zeros(U, np.int64)
    unit_rel = np.full(slots, -1, np.int32)
    for u in sorted(range(U), key=lambda u: (-usize[u], u)):
        w = min((i for i in range(n_waves) if used[i] < per_wave), key=lambda i: (load[i], i))
        owner[u], local[u] = w, used[w]
        unit_rel[w * per_wave + used[w]] = urel[u]
        used[w] += 1
        load[w] += usize[u]
    mean = max(1.0, float(sum(load)) / n_waves)
    return parts, unit_base, owner, local, unit_rel, max(load) / mean


def build_softwin_plan_device(dst, src, rel, val, alive, n_dst, n_src, num_rels, tile_rows, own_waves=0, own_per_wave=0):
    """build_softwin_plan through the C ABI (rgcn_softwin_order / rgcn_softwin_fill: two rocPRIM radix sorts, a histogram, three scans, four small
    kernels): the form the layers use on the GPU.  Two host reads (m_pad / live messages / the relations' message counts; nothing else)."""
    dev = dst.device
    M = dst.shape[0]
    n_tiles = (n_dst + tile_rows - 1) // tile_rows
    nbk = n_tiles * num_rels
    n_groups = n_tiles * max(own_waves, 1)
    L = lib()
    # (the chunks sorted in the second step number at most M / 16 + one per non-empty bucket)
    tmp_bytes = int(L.rgcn_softwin_tmp_bytes(c_i64(max(M, nbk + 1, n_groups + 1, M // CHUNK + min(nbk, M) + 1))))
    tmp = torch.empty(tmp_bytes, dtype=torch.uint8, device=dev)
    keys = torch.empty(2 * max(M, 1), dtype=torch.int64, device=dev)
    order = torch.empty(2 * max(M, 1), dtype=torch.int32, device=dev)
    bucket_cnt, bucket_base, bucket_first = _i32(nbk + 1, dev), _i32(nbk + 1, dev), _i32(nbk + 1, dev)
    with _on(dev):
        _check(L.rgcn_softwin_order(_dp(dst), _dp(src), _dp(rel), _dp(alive), c_i64(M), c_i64(n_dst), c_i64(n_src), c_i32(num_rels), c_i32(tile_rows),
                                    _dp(keys[:max(M, 1)]), _dp(keys[max(M, 1):]), _dp(order[:max(M, 1)]), _dp(order[max(M, 1):]), _dp(bucket_cnt),
                                    _dp(bucket_base), _dp(bucket_first), _dp(tmp), c_i64(tmp_bytes), _stream(dev)), "softwin_order")
    rel_counts = bucket_cnt[:nbk].view(n_tiles, num_rels).sum(0, dtype=torch.int64)
    host = torch.cat([bucket_base[nbk:nbk + 1].long(), bucket_first[nbk:nbk + 1].long(), bucket_cnt[:nbk].max().view(1).long(), rel_counts]).cpu().numpy()
    m_pad, n_live, max_cnt = int(host[0]), int(host[1]), int(host[2])
    own = None
    if own_waves:
        own = own_relations(host[3:], own_waves, own_per_wave)
        if own is None:
            return None
    n_chunks = m_pad // CHUNK
    p = BuiltPlan()
    p.device = dev
    p.n_dst, p.n_src, p.num_rels, p.tile_rows = n_dst, n_src, num_rels, tile_rows
    p.n_tiles, p.m_pad, p.n_chunks, p.n_messages = n_tiles, m_pad, n_chunks, n_live
    p.src, p.dst = _i32(m_pad, dev), _i32(m_pad, dev)
    p.val = torch.empty(max(m_pad, 1), dtype=torch.float32, device=dev)
    p.chunk_rel = _i32(n_chunks, dev)
    group_ptr = _i32(n_groups + 1, dev)
    stage = _i32(2 * max(m_pad, 1), dev)
    stage_val = torch.empty(max(m_pad, 1), dtype=torch.float32, device=dev)
    ckeys = torch.empty(2 * max(n_chunks, 1), dtype=torch.int64, device=dev)
    cidx, crel, group_cnt = _i32(2 * max(n_chunks, 1), dev), _i32(n_chunks, dev), _i32(n_groups + 1, dev)
    tabs = [None] * 4
    if own is not None:
        tabs = [torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev) for a in own[:4]]
    nc1, mp1 = max(n_chunks, 1), max(m_pad, 1)
    with _on(dev):
        _check(L.rgcn_softwin_fill(_dp(dst), _dp(src), _dp(val), _dp(keys[max(M, 1):]), _dp(order[max(M, 1):]), c_i64(n_live), c_i64(n_dst), c_i64(n_src),
                                   c_i32(num_rels), c_i32(tile_rows), _dp(bucket_base), _dp(bucket_first), c_i64(m_pad), _dp(tabs[0]), _dp(tabs[1]),
                                   _dp(tabs[2]), _dp(tabs[3]), c_i32(own_waves), _dp(stage[:mp1]), _dp(stage[mp1:]), _dp(stage_val), _dp(ckeys[:nc1]),
                                   _dp(ckeys[nc1:]), _dp(cidx[:nc1]), _dp(cidx[nc1:]), _dp(crel), _dp(group_cnt), _dp(p.src), _dp(p.dst), _dp(p.val),
                                   _dp(p.chunk_rel), _dp(group_ptr), _dp(tmp), c_i64(tmp_bytes), _stream(dev)), "softwin_fill")
    p.tile_ptr = group_ptr[::max(own_waves, 1)].contiguous()
    p.run_ptr = torch.zeros(n_tiles * (num_rels + 1), dtype=torch.int32, device=dev)
    p.run_ptr[0::num_rels + 1] = p.tile_ptr[:-1]
    p.run_ptr[num_rels::num_rels + 1] = p.tile_ptr[1:]
    if own is not None:
        p.own_ptr = group_ptr
        p.unit_rel = torch.from_numpy(own[4]).to(dev)
        p.own_waves, p.own_per_wave, p.own_balance = own_waves, own_per_wave, float(own[5])
    p.pack, p.aux = None, None
    p.soft_windows = True
    p.units_host = np.zeros((0, 4), np.int32)      # (not None: _blk_units cuts hub tiles into pieces from tile_ptr)
    p.n_units, p.n_split, p.units = n_tiles, 0, None
    p.n_items, p.items = 0, _i32(2, dev).view(1, 2)
    p.max_run_chunks = (max_cnt + CHUNK - 1) // CHUNK
    return p


def build_softwin_plan(dst, src, rel, val, alive, n_dst, n_src, num_rels, tile_rows, own_waves=0, own_per_wave=0):
    """Plan of tall workgroup-owned tiles for the block-tile kernels (rgcn_spmm_blk_f32, rgcn_bwd_own_f32) in SOFT-WINDOW order (round 6):

      * inside a (tile, relation) bucket the slots are sorted by SOURCE row -- the block-tile kernels add with ds_add_f64 and the chunk
        records carry every slot's tile row, so the order of the slots of a bucket is free -- hence the 16 sources of a chunk span
        1 / (chunks per bucket) of the table (S1, 977-row tiles: 203 messages per bucket, 13 chunks, ~5 MB of X);
      * the chunks of a tile are ordered by their first source, whatever their relation.

    Every workgroup then sweeps the source table once per tile and all workgroups sweep roughly together: the row gathers of the whole
    chip fall into a few MB at any time.  tools/micro/gather_window.hip: S1's 21 M row reads take 0.19-0.22 ms in window-major order
    against 0.375 ms uniformly random (profiles/r06_gather_window.txt); the forward kernel on this plan 0.327 ms against 0.398 on the same
    tiles in destination order (tools/softwin_probe.py) -- at the plain (tile, relation) padding (1.04), no window buckets to pad.

    own_waves > 0 (the relation-owner backward, rgcn_bwd_own_f32): every relation -- a large one: every PART of it -- belongs to one of
    own_waves waves (own_relations: LPT over the message counts, at most own_per_wave units per wave); a tile's chunks are grouped by owner
    wave, each wave's chunks ordered by first source: own_ptr[tile * own_waves + wave] = the wave's first chunk; chunk_rel carries the unit's
    local number in its high half (rel | local << 16); unit_rel[wave * own_per_wave + local] = relation.  None when the relations do not
    fit the slots.

    On the GPU this is build_softwin_plan_device (the C ABI: rgcn_softwin_order / rgcn_softwin_fill); the body below is the same procedure in
    torch ops (two sorts, a histogram, two scans) -- CPU tensors (tests/test_softwin_plan.py checks the invariants on it) and
    RGCN_SOFTWIN_BUILD=torch.  One-off preprocessing of STATIC graphs; per-call graphs keep build_plan_device.  Same fields as BuiltPlan; run_ptr holds only a tile's first and end chunk (entries 0 and R of its row) -- all the
    block-tile kernels read -- and there is no packed slot array."""
    dev = dst.device
    if dev.type == "cuda" and routes.get("softwin_build", "device") != "torch":
        return build_softwin_plan_device(dst, src, rel, val, alive, n_dst, n_src, num_rels, tile_rows, own_waves, own_per_wave)
    M = dst.shape[0]
    n_tiles = (n_dst + tile_rows - 1) // tile_rows
    nbk = n_tiles * num_rels
    live = None if alive is None else (alive != 0)
    d64, s64, r64 = dst.long(), src.long(), rel.long()
    bucket = torch.div(d64, tile_rows, rounding_mode="floor") * num_rels + r64
    key = bucket * n_src + s64
    if live is not None:
        key = torch.where(live, key, torch.full_like(key, nbk * n_src))       # dropped messages sort behind everything
    perm = torch.argsort(key)
    n_live = M if live is None else int(live.sum().item())
    perm = perm[:n_live]
    bs = bucket[perm]
    cnt = torch.bincount(bs, minlength=nbk)
    padded = (cnt + (CHUNK - 1)) // CHUNK * CHUNK
    base = torch.cumsum(padded, 0) - padded
    first = torch.cumsum(cnt, 0) - cnt
    slot = base[bs] + (torch.arange(n_live, device=dev) - first[bs])
    m_pad = int(padded.sum().item())
    n_chunks = m_pad // CHUNK
    S = torch.zeros(max(m_pad, 1), dtype=torch.int32, device=dev)
    D = torch.full((max(m_pad, 1),), -1, dtype=torch.int32, device=dev)
    V = torch.zeros(max(m_pad, 1), dtype=torch.float32, device=dev)
    S[slot], D[slot], V[slot] = src[perm].to(torch.int32), dst[perm].to(torch.int32), val[perm]
    cb = torch.repeat_interleave(torch.arange(nbk, device=dev), padded // CHUNK, output_size=n_chunks)    # bucket of every chunk
    ctile = torch.div(cb, num_rels, rounding_mode="floor")
    own = None
    if own_waves:
        live_rel = r64 if live is None else r64[live]
        own = own_relations(torch.bincount(live_rel, minlength=num_rels).cpu().numpy(), own_waves, own_per_wave)
        if own is None:
            return None
        parts_t, ubase_t = torch.from_numpy(own[0]).to(dev), torch.from_numpy(own[1]).to(dev)
        owner_t, local_t = torch.from_numpy(own[2]).to(dev), torch.from_numpy(own[3]).to(dev)
        crel0 = cb % num_rels
        first_chunk = torch.cumsum(padded // CHUNK, 0) - padded // CHUNK          # first chunk of every bucket (bucket-major layout)
        cunit = ubase_t[crel0] + (torch.arange(n_chunks, device=dev) - first_chunk[cb]) % parts_t[crel0]      # chunk j of its bucket -> part j % parts
        cgroup = ctile * own_waves + owner_t[cunit]                             # (tile, owner wave) of every chunk
    else:
        cgroup = ctile
    cperm = torch.argsort(cgroup * n_src + S[:m_pad:CHUNK].long())             # inside a tile (a wave's share of it): by the chunk's first source (a real slot)
    idx = (cperm[:, None] * CHUNK + torch.arange(CHUNK, device=dev)[None, :]).reshape(-1)
    p = BuiltPlan()
    p.device = dev
    p.n_dst, p.n_src, p.num_rels, p.tile_rows = n_dst, n_src, num_rels, tile_rows
    p.n_tiles, p.m_pad, p.n_chunks, p.n_messages = n_tiles, m_pad, n_chunks, n_live
    if m_pad:
        p.src, p.dst, p.val = S[idx].contiguous(), D[idx].contiguous(), V[idx].contiguous()
    else:
        p.src, p.dst, p.val = S, D, V
    crel = cb % num_rels
    if own is not None:
        crel = crel | (local_t[cunit] << 16)
        gcnt = torch.bincount(cgroup, minlength=n_tiles * own_waves)
        p.own_ptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(gcnt, 0)]).to(torch.int32)
        p.unit_rel = torch.from_numpy(own[4]).to(dev)
        p.own_waves, p.own_per_wave, p.own_balance = own_waves, own_per_wave, float(own[5])
    p.chunk_rel = crel[cperm].to(torch.int32).contiguous() if n_chunks else _i32(0, dev)
    tcnt = torch.bincount(ctile, minlength=n_tiles)
    tend = torch.cumsum(tcnt, 0)
    p.tile_ptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), tend]).to(torch.int32)
    p.run_ptr = torch.zeros(n_tiles * (num_rels + 1), dtype=torch.int32, device=dev)
    p.run_ptr[0::num_rels + 1] = (tend - tcnt).to(torch.int32)
    p.run_ptr[num_rels::num_rels + 1] = tend.to(torch.int32)
    p.pack, p.aux = None, None
    p.soft_windows = True
    p.units_host = np.zeros((0, 4), np.int32)      # (not None: _blk_units cuts hub tiles into pieces from tile_ptr)
    p.n_units, p.n_split, p.units = n_tiles, 0, None
    p.n_items, p.items = 0, _i32(2, dev).view(1, 2)
    p.max_run_chunks = int(padded.max().item()) // CHUNK if nbk else 0
    return p


class CsrPlan:
    """destination-major CSR (rowptr, src, rel, val) built on the device: the layout of the basis kernels"""


def build_csr_device(dst, src, rel, val, alive, n_rows, sync_free=False, want_slot=True):
    """want_slot=False: no msg_slot array (input message -> CSR position; only the two-pass / featureless-basis routes read it)"""
    dev = dst.device
    M = dst.shape[0]
    # `cells` (one counter per row) lives at rowbuf[1:]: the count pass leaves the rows' exclusive offsets there and the fill
    # pass advances every counter to the END of its row -- which is the start of the next one, so rowbuf (rowbuf[0] = 0)
    # IS the CSR row pointer afterwards.  No device-to-device copy: inside a captured hipGraph such copies become memcpy
    # nodes, and those fault on this ROCm build when pageable host-to-device copies run between replays (tools/hipgraph_repro/).
    rowbuf = torch.zeros(n_rows + 2, dtype=torch.int32, device=dev)
    cells, cells_tmp = rowbuf[1:], _i32(n_rows + 1, dev)
    bucket_cnt, bucket_base, scan_tmp = _i32(1, dev), _i32(2, dev), _i32(n_rows // 1024 + 4, dev)
    L = lib()
    with _on(dev):         # (one bucket, one "relation": the relation array of the builder is NULL)
        _check(L.rgcn_dev_plan_count(_dp(dst), None, _dp(alive), c_i64(M), c_i64(n_rows), c_i32(1), c_i32(n_rows),
                                     _dp(cells), _dp(bucket_cnt), _dp(bucket_base), _dp(scan_tmp), _dp(cells_tmp),
                                     _stream(dev)), "dev_plan_count")
    # one bucket: the padded size is the live message count rounded up to 16 -- sized by its bound, the list length (dead
    # messages are few: dropped self loops), so that building a CSR never reads anything back from the device
    m_pad = (M + CHUNK - 1) // CHUNK * CHUNK
    p = CsrPlan()
    p.n_rows = n_rows
    # CSR position of every live input message
    p.msg_slot = torch.full((max(M, 1),), -1, dtype=torch.int32, device=dev)[:M] if want_slot else None
    p.n_messages = None
    p.src, pdst, p.rel = _i32(m_pad, dev), _i32(m_pad, dev), _i32(m_pad, dev)
    p.val = torch.empty(max(m_pad, 1), dtype=torch.float32, device=dev)
    tile_ptr = _i32(2, dev)
    with _on(dev):
        _check(L.rgcn_dev_plan_fill(_dp(dst), _dp(src), None, _dp(val), _dp(alive), c_i64(M), c_i64(n_rows),
                                    c_i64(n_rows), c_i32(1), c_i32(n_rows), _dp(cells), _dp(bucket_cnt), _dp(bucket_base),
                                    _dp(p.src), _dp(pdst), _dp(p.val), None, None, _dp(tile_ptr), None,
                                    _dp(rel), _dp(p.rel), _dp(p.msg_slot), c_i64(m_pad // CHUNK), _stream(dev)),
               "dev_plan_fill")
    p.rowptr = rowbuf[: n_rows + 1]
    p.sync_free = sync_free
    p.units = None
    return p


def build_csr_pair_device(a, b, rel, val, alive, n_rows):
    """(csr_by_a, csr_by_b): rows = a with entries (b, rel, val) and rows = b with entries (a, rel, val), built together in five
    launches without a read-back (rgcn_dev_csr_pair); no msg_slot arrays -- for per-call graphs and the DistMult backward"""
    dev = a.device
    M = a.shape[0]
    rowbuf = _i32(2 * n_rows + 2, dev)
    scan_tmp = _i32((2 * n_rows + 1) // 1024 + 4, dev)
    e_other, e_rel = _i32(2 * M, dev), _i32(2 * M, dev)
    e_val = torch.empty(max(2 * M, 1), dtype=torch.float32, device=dev)
    with _on(dev):
        _check(lib().rgcn_dev_csr_pair(_dp(a), _dp(b), _dp(rel), _dp(val), _dp(alive), c_i64(M), c_i64(n_rows), _dp(rowbuf),
                                       _dp(scan_tmp), _dp(e_other), _dp(e_rel), _dp(e_val), _stream(dev)), "dev_csr_pair")
    out = []
    for k in range(2):
        p = CsrPlan()
        p.n_rows = n_rows
        p.msg_slot, p.n_messages = None, None
        p.src, p.rel, p.val = e_other, e_rel, e_val
        p.rowptr = rowbuf[k * (n_rows + 1): k * (n_rows + 1) + n_rows + 1]
        p.sync_free, p.per_call, p.units = True, True, None
        out.append(p)
    return out[0], out[1]


def row_units(rowptr, n_rows, max_len):
    """work units {row, first entry, end entry, flags} over a CSR: one per row, rows longer than max_len cut into pieces
    (flags RGCN_U_SHARED, first piece also RGCN_U_FIRST).  -> (units int32 [n_units, 4] on the device, n_units, n_split)"""
    dev = rowptr.device
    rp = rowptr[: n_rows + 1].to(torch.int64)
    deg = rp[1:] - rp[:-1]
    pieces = torch.clamp((deg + max_len - 1) // max_len, min=1)
    n_units = int(pieces.sum().item())
    row = torch.repeat_interleave(torch.arange(n_rows, device=dev), pieces, output_size=n_units)
    first = torch.cumsum(pieces, 0) - pieces
    k = torch.arange(n_units, device=dev) - first[row]
    begin = rp[row] + k * max_len
    end = torch.minimum(begin + max_len, rp[row + 1])
    shared = pieces[row] > 1
    flags = shared.to(torch.int64) * 1 + (shared & (k == 0)).to(torch.int64) * 2
    units = torch.stack([row, begin, end, flags], dim=1).to(torch.int32).contiguous()
    return units, n_units, int(shared.sum().item())


class FBasisPlan:
    """source-major view of a graph for the featureless basis layer (see csrc/rgcn_basis.hip)"""
    __slots__ = ("e_dst", "e_rel", "e_val", "n_messages", "units_src", "perm_dst", "units_dst", "perm_rel", "units_rel",
                 "n_nodes", "num_rels", "rowptr_src", "max_src_degree")


def build_fbasis_plan(csr_src, csr_dst, n_nodes, num_rels, max_len=1024):
    """csr_src: rows = source nodes (entries: destination, relation, val); csr_dst: rows = destinations.  Both were
    built from the same message list, so their msg_slot arrays link the two orders."""
    p = FBasisPlan()
    M = int(csr_src.rowptr[n_nodes].item())
    p.n_messages, p.n_nodes, p.num_rels = M, n_nodes, num_rels
    p.e_dst, p.e_rel, p.e_val = csr_src.src, csr_src.rel, csr_src.val
    p.rowptr_src = csr_src.rowptr
    rp = csr_src.rowptr[: n_nodes + 1]
    p.max_src_degree = int((rp[1:] - rp[:-1]).max().item()) if n_nodes > 0 else 0       # static graph: read once (the tile kernels' mode)
    p.units_src = row_units(csr_src.rowptr, n_nodes, max_len)
    live = csr_src.msg_slot >= 0
    perm = torch.zeros(max(M, 1), dtype=torch.int32, device=csr_src.rowptr.device)
    perm[csr_dst.msg_slot[live].long()] = csr_src.msg_slot[live]
    p.perm_dst = perm                                         # destination-major position -> source-major position
    p.units_dst = row_units(csr_dst.rowptr, n_nodes, max_len)
    rel = csr_src.rel[:M].long()
    p.perm_rel = torch.argsort(rel, stable=True).to(torch.int32)     # relation-major order of source-major positions
    rel_ptr = torch.zeros(num_rels + 1, dtype=torch.int64, device=rel.device)
    rel_ptr[1:] = torch.cumsum(torch.bincount(rel, minlength=num_rels), 0)
    # a relation holds M / R messages: cut it into enough pieces (>= ~16 k units overall) to fill the chip
    p.units_rel = row_units(rel_ptr, num_rels, int(min(2 * max_len, max(64, M // 16384))))
    return p


def fbasis_supported(B, d):
    """source-major kernels: within their register limits, and worth it only when a node's B x d block is big enough
    that re-reading it per message (destination-major fallback) costs more than the Y / T round trip (measured: B >= 4)"""
    if B < 4:
        return False
    dp = 4
    while dp < d and dp < 64:
        dp <<= 1
    ngrp = 64 // dp
    return d <= 16 and B <= 64 and 4 * ((B + 4 * ngrp - 1) // (4 * ngrp)) <= 16


def fbasis_fwd(table, comps, bias, plan, basis_major=False, relu=False):
    """table: the bases, node-major [N, B, d] or -- basis_major -- in the parameter's own [B, N, d] layout (no transposed copy);
    relu: applied in the row sum's epilogue (-> (out, True)) unless hub rows are cut into shared pieces (-> (out, False))"""
    bases = table
    _req(bases, "bases"); _req(comps, "comps"); _req(bias, "bias")
    N, B, d = (bases.shape[1], bases.shape[0], bases.shape[2]) if basis_major else bases.shape
    dev = bases.device
    Y = torch.empty(max(plan.n_messages, 1), d, device=dev, dtype=torch.float32)
    out = torch.empty(N, d, device=dev, dtype=torch.float32)
    units, n_units, _ = plan.units_src
    with _on(dev), _timed("fbasis_fwd"):
        _check(lib().rgcn_fbasis_fwd_f32(_dp(bases), _dp(comps), _dp(Y), _dp(plan.e_rel), _dp(plan.e_val), _dp(units),
                                         c_i64(n_units), c_i64(N), c_i32(comps.shape[0]), c_i32(B), c_i32(d),
                                         c_i32(1 if basis_major else 0), _stream(dev)), "fbasis_fwd")
        units, n_units, n_split = plan.units_dst
        fused = bool(relu) and n_split == 0
        _check(lib().rgcn_gather_rows_sum_f32(_dp(Y), _dp(plan.perm_dst), _dp(units), c_i64(n_units), c_i64(n_split),
                                              _dp(bias), _dp(out), c_i64(N), c_i32(d), c_i32(1 if fused else 0), _stream(dev)), "gather_rows_sum")
    return (out, fused) if relu else out


def fbasis_bwd(table, comps, g, plan, need_bases=True, need_comps=True, basis_major=False):
    """-> (d table in the layout of `table`: node-major [N, B, d] or, basis_major, [B, N, d]; dcomps [R, B])"""
    bases = table
    _req(bases, "bases"); _req(comps, "comps"); _req(g, "grad")
    N, B, d = (bases.shape[1], bases.shape[0], bases.shape[2]) if basis_major else bases.shape
    R = comps.shape[0]
    dev = bases.device
    dB = torch.empty_like(bases) if need_bases else None
    dC = torch.empty(R, B, device=dev, dtype=torch.float32) if need_comps else None
    units, n_units, n_split = plan.units_src
    if need_comps and not routes.flag("deterministic") and \
            lib().rgcn_fbasis_bwd_dc_supported(c_i32(R), c_i32(B), c_i32(d)):
        # dcomps summed in an LDS table of doubles inside the walk: no [M, B] scratch, no second pass
        with _on(dev), _timed("fbasis_bwd"):
            _check(lib().rgcn_fbasis_bwd_dc_f32(_dp(bases), _dp(comps), _dp(g), _dp(dB), _dp(dC), _dp(plan.e_dst), _dp(plan.e_rel),
                                                _dp(plan.e_val), _dp(units), c_i64(n_units), c_i64(n_split), c_i64(N), c_i32(R),
                                                c_i32(B), c_i32(d), c_i32(1 if basis_major else 0), _stream(dev)), "fbasis_bwd_dc")
        return dB, dC
    T = torch.empty(max(plan.n_messages, 1), B, device=dev, dtype=torch.float32) if need_comps else None
    with _on(dev), _timed("fbasis_bwd"):
        _check(lib().rgcn_fbasis_bwd_f32(_dp(bases), _dp(comps), _dp(g), _dp(dB), _dp(T), _dp(plan.e_dst), _dp(plan.e_rel),
                                         _dp(plan.e_val), _dp(units), c_i64(n_units), c_i64(n_split), c_i64(N), c_i32(R),
                                         c_i32(B), c_i32(d), c_i32(1 if basis_major else 0), _stream(dev)), "fbasis_bwd")
        if need_comps:
            units, n_units, n_split = plan.units_rel
            _check(lib().rgcn_gather_rows_sum_f32(_dp(T), _dp(plan.perm_rel), _dp(units), c_i64(n_units), c_i64(n_split),
                                                  None, _dp(dC), c_i64(R), c_i32(B), c_i32(0), _stream(dev)), "gather_rows_sum")
    return dB, dC


TILE_NODE_MODE_MAX_DEGREE = 4096


def fbasis_tile_ok(R, B, d, n_nodes, max_degree=None):
    """-> (available, mode) of the tile kernels (rgcn_fbasis_tile.hip: the table walked IN the parameter's [B, N, d] layout, 16 source
    nodes per tile).  mode 1 = one wave per node on the matrix cores -- for graphs without hub sources (largest source degree <=
    TILE_NODE_MODE_MAX_DEGREE: a hub is ONE wave's work there); mode 0 = a tile's messages dealt evenly over the 16 waves.
    Route fbasis_tile: 0 off, ranges / nodes force a mode; the deterministic mode keeps the wave-per-node kernels of rgcn_basis.hip."""
    route = routes.get("fbasis_tile", "1")
    if route == "0" or routes.flag("deterministic"):
        return False, 0
    m = lib().rgcn_fbasis_tile_supported(c_i32(R), c_i32(B), c_i32(d), c_i64(n_nodes))
    ranges, nodes = (m & 3) == 3, (m & 12) == 12
    if route == "nodes":
        return nodes, 1
    if route == "nodes2":           # one wave per node, dbases and dcomps as two kernels (rounds 4's backward; comparisons)
        return nodes, 3
    if route == "ranges":
        return ranges, 0
    if nodes and max_degree is not None and max_degree <= TILE_NODE_MODE_MAX_DEGREE:
        return True, 1
    return ranges, 0


def fbasis_tile_fwd(bases, comps, bias, plan, relu=False, mode=0, padded=False):
    """bases [B, N, d] (the parameter itself) -> out [N, d]; relu: as fbasis_fwd; mode: fbasis_tile_ok.  padded: the rows are written into
    a zero-padded [N, 16 k] buffer and its first d columns returned as a VIEW (row stride 16 k: what the width-16 kernels of the next
    layer read in place -- no pad copy there, no crop copy of its feature gradient on the way back)"""
    _req(bases, "bases"); _req(comps, "comps"); _req(bias, "bias")
    B, N, d = bases.shape
    dev = bases.device
    ys = int(lib().rgcn_fbasis_tile_ystride(c_i32(d)))
    Y = torch.empty(max(plan.n_messages, 1), ys, device=dev, dtype=torch.float32)
    ow = d + (-d % 16) if (padded and d % 16 and d + (-d % 16) <= ys) else d
    out = torch.empty(N, ow, device=dev, dtype=torch.float32)
    with _on(dev), _timed("fbasis_tile_fwd"):
        _check(lib().rgcn_fbasis_tile_fwd_f32(_dp(bases), _dp(comps), _dp(Y), _dp(plan.rowptr_src), _dp(plan.e_rel), _dp(plan.e_val),
                                              c_i64(plan.n_messages), c_i64(N), c_i32(comps.shape[0]), c_i32(B), c_i32(d), c_i32(1 if mode else 0), _stream(dev)),
               "fbasis_tile_fwd")
    units, n_units, n_split = plan.units_dst
    fused = bool(relu) and n_split == 0
    with _on(dev), _timed("gather_rows_sum4"):
        _check(lib().rgcn_gather_rows_sum4_f32(_dp(Y), c_i32(ys), _dp(plan.perm_dst), _dp(units), c_i64(n_units), c_i64(n_split),
                                               _dp(bias), _dp(out), c_i64(N), c_i32(d), c_i32(ow), c_i32(1 if fused else 0), _stream(dev)), "gather_rows_sum4")
    if ow != d:
        out = out[:, :d]
        out._rgcn_zero_padded = True
    return (out, fused) if relu else out


def fbasis_tile_bwd(bases, comps, g, plan, need_bases=True, need_comps=True, mode=0):
    """-> (dbases [B, N, d] in the parameter's layout, dcomps [R, B]); g [N, d], rows contiguous or at a stride of 16 k floats (the
    first columns of the zero-padded rows a width-16 layer hands back: taken in place)"""
    _req(bases, "bases"); _req(comps, "comps")
    B, N, d = bases.shape
    if g.dim() == 2 and g.stride(1) == 1 and g.stride(0) >= g.shape[1] and g.stride(0) % 4 == 0 and g.data_ptr() % 16 == 0 and not g.is_contiguous():
        g_stride = int(g.stride(0))
    else:
        _req(g, "grad")
        g_stride = d
    R = comps.shape[0]
    dev = bases.device
    dB = torch.empty_like(bases) if need_bases else None
    dC = torch.empty(R, B, device=dev, dtype=torch.float32) if need_comps else None
    if dB is None and dC is None:
        return None, None
    with _on(dev), _timed("fbasis_tile_bwd"):
        _check(lib().rgcn_fbasis_tile_bwd_f32(_dp(bases), _dp(comps), _dp(g), c_i32(g_stride), _dp(dB), _dp(dC), _dp(plan.rowptr_src), _dp(plan.e_dst),
                                              _dp(plan.e_rel), _dp(plan.e_val), c_i64(plan.n_messages), c_i64(N), c_i32(R), c_i32(B), c_i32(d), c_i32(mode),
                                              _stream(dev)), "fbasis_tile_bwd")
    return dB, dC


def basis_aggregate(X, comps, csr, B, d, n_b_in):
    _req(X, "features"); _req(comps, "comps")
    out = torch.empty((csr.n_rows, B * d) if n_b_in == 1 else (csr.n_rows, d), device=X.device, dtype=torch.float32)
    with _on(X.device), _timed("basis_aggregate"):
        _check(lib().rgcn_basis_aggregate_f32(_dp(X), _dp(comps), _dp(out), _dp(csr.rowptr), _dp(csr.src), _dp(csr.rel),
                                              _dp(csr.val), c_i64(csr.n_rows), c_i32(comps.shape[0]), c_i32(B), c_i32(d),
                                              c_i32(n_b_in), _stream(X.device)), "basis_aggregate")
    return out


def fbasis_small_ok(R, B, d):
    return bool(lib().rgcn_fbasis_small_supported(c_i32(R), c_i32(B), c_i32(d)))


def fbasis_small_bwd(G, table, comps, csr, B, d, basis_major=False):
    """featureless basis layer with small blocks (B <= 4), both gradients from ONE walk of the source-major CSR
    (rgcn_fbasis_small_bwd_f32): table [N, B, d] node-major -> (dbases [N, B, d], dcomps [R, B]); basis_major: table and its gradient in
    the parameter's own [B, N, d] layout (a node's B blocks are then B sequential streams: no transposed copy of the gradient)"""
    _req(G, "grad_output"); _req(table, "bases"); _req(comps, "comps")
    R = comps.shape[0]
    dB = torch.empty_like(table)
    dC = torch.empty((R, B), device=G.device, dtype=torch.float32)
    with _on(G.device), _timed("fbasis_small_bwd"):
        _check(lib().rgcn_fbasis_small_bwd_f32(_dp(G), _dp(table), _dp(comps), _dp(dB), _dp(dC), _dp(csr.rowptr), _dp(csr.src), _dp(csr.rel),
                                               _dp(csr.val), c_i64(csr.n_rows), c_i32(R), c_i32(B), c_i32(d), c_i32(1 if basis_major else 0),
                                               _stream(G.device)),
               "fbasis_small_bwd")
    return dB, dC


def basis_dcomps(X, D, plan, R, B, d, swap=False):
    """dcomps[r,b] = sum_e val <X[src_e], D[dst_e, b]>; plan: relation-major plan (graph.wgt_plan()).
    swap=True exchanges the roles of the two index arrays (featureless layers: X = grad rows by destination,
    D = the basis table by source)."""
    _req(X, "features"); _req(D, "grad")
    copies = 16 if R * B <= 4096 else 1      # few addresses: spread the pieces' atomics over copies (summed below)
    dc = torch.empty((copies, R, B), device=X.device, dtype=torch.float32)
    a, b = (plan.dst, plan.src) if swap else (plan.src, plan.dst)
    with _on(X.device), _timed("basis_dcomps"):
        _check(lib().rgcn_basis_dcomps_f32(_dp(X), _dp(D), _dp(dc), _dp(a), _dp(b), _dp(plan.val),
                                           _dp(plan.chunk_rel), _dp(plan.items), c_i64(plan.n_items), c_i32(R), c_i32(B),
                                           c_i32(d), c_i32(copies), _stream(X.device)), "basis_dcomps")
    return dc.sum(0) if copies > 1 else dc[0]


_TABLE_WS = {}
_WS_KEEP = []


def _table_workspace(dev, R, B):
    """workspace of the kernels that reduce per-workgroup R x B tables in a fixed order: one per (device, stream) -- launches on one stream
    are ordered, two streams must not share the ticket and the partials --, zeroed at creation (every launch leaves its ticket
    zeroed), replaced by a larger one when a larger table comes along; a capture has its own (the capture stream's, allocated from the
    graph's pool) and a buffer a captured graph may still point at is never released"""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)      # per stream: two streams' launches would race on the ticket and the partials
    need = int(lib().rgcn_basis_sum_workspace_bytes(c_i32(R), c_i32(B)))
    ws = _TABLE_WS.get(key)
    if ws is None or ws.numel() < need:
        if ws is not None and torch.cuda.is_current_stream_capturing():
            _WS_KEEP.append(ws)       # an earlier capture on this stream has the smaller buffer's address baked into its kernels: it stays
        ws = _TABLE_WS[key] = torch.zeros(need, dtype=torch.uint8, device=dev)
    return ws


def basis_dcomps_csr_ok(R, B, d):
    return bool(lib().rgcn_basis_dcomps_csr_supported(c_i32(R), c_i32(B), c_i32(d)))


def basis_dcomps_csr(X, D, csr, R, B, d):
    """dcomps[r,b] = sum_e val <X[src_e], D[dst_e, b, :]> on the destination-major CSR (rgcn_basis_dcomps_csr_f32): X [N, d], D [N, B*d]"""
    _req(X, "features"); _req(D, "grad")
    dc = torch.empty((R, B), device=X.device, dtype=torch.float32)
    with _on(X.device), _timed("basis_dcomps_csr"):
        _check(lib().rgcn_basis_dcomps_csr_f32(_dp(X), _dp(D), _dp(dc), _dp(csr.rowptr), _dp(csr.src), _dp(csr.rel), _dp(csr.val),
                                               c_i64(csr.n_rows), c_i32(R), c_i32(B), c_i32(d), _dp(_table_workspace(X.device, R, B)),
                                               _stream(X.device)), "basis_dcomps_csr")
    return dc


G_TRANS_A, G_TRANS_B = 1, 2


def gemm(A, B, bias=None, trans_a=False, trans_b=False, split_k=1):
    """C = op(A) @ op(B) (+ bias) on the matrix cores (rgcn_gemm_f32; fp32, exact FMA chains).  A: [M, K] or, trans_a,
    stored [K, M]; B: [K, N] or, trans_b, stored [N, K]."""
    _req(A, "A"); _req(B, "B"); _req(bias, "bias")
    assert A.dim() == 2 and B.dim() == 2
    M, K = (A.shape[1], A.shape[0]) if trans_a else A.shape
    N = B.shape[0] if trans_b else B.shape[1]
    assert (B.shape[1] if trans_b else B.shape[0]) == K, f"gemm: inner dimensions {tuple(A.shape)} x {tuple(B.shape)}"
    C = torch.empty((M, N), device=A.device, dtype=torch.float32)
    if M == 0 or N == 0:
        return C
    scratch = None
    if split_k > 1:
        scratch = torch.empty(int(lib().rgcn_gemm_scratch_floats(c_i64(M), c_i64(N), c_i64(K), c_i32(split_k))),
                              device=A.device, dtype=torch.float32)
    flags = (G_TRANS_A if trans_a else 0) | (G_TRANS_B if trans_b else 0)
    with _on(A.device), _timed("gemm"):
        _check(lib().rgcn_gemm_f32(_dp(A), _dp(B), _dp(bias), _dp(C), _dp(scratch), c_i64(M), c_i64(N), c_i64(K),
                                   c_i64(A.shape[1]), c_i64(B.shape[1]), c_i64(N), c_i32(flags), c_i32(split_k),
                                   _stream(A.device)), "gemm")
    return C


def _req(t, name, dtype=torch.float32):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on a GPU: torch_rgcn runs on HIP kernels only (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if t.data_ptr() % (16 if t.is_floating_point() else t.element_size()):
        # the kernels pick their 16-byte load / store forms from the row width alone; a contiguous view at an odd storage offset
        # (a slice of a flattened tensor) would hand them a misaligned base -- functional.py's dense() copies such views.
        # Index tensors (int64 triples: 24-byte rows, read element by element) only need their natural alignment: triples[1:] is fine.
        raise ValueError(f"{name} must be 16-byte aligned (got a view at storage offset {t.storage_offset()}): pass x.clone()")


F_RELU, F_WPACKED = 1, 2
F_DIAG4 = 16      # rgcn_bwd_blk_f32: block-diagonal weights, 4 x 4 blocks
F_PARTIAL, F_ACCUMULATE = 32, 64      # rgcn_bwd_blk_f32: a slab of whole tiles (first / further slabs of one backward)


# Both fragment orders (Wp, Wtp) of a [R,16,16] weight are packed in ONE launch the first time either is asked for.  They are
# remembered only inside a `w16_scope` -- one autograd call: the forward packs once and hands the pair to ITS backward through ctx
# -- never in a process-wide cache: a parameter can change without its version counter moving (hipGraph replays run the
# optimiser on the device, `W.data` writes), and a cache keyed on (identity, version) would then serve stale fragments
# (an eager evaluation after replayed training steps scored with the weights of the first evaluation).
_W16_TLS = threading.local()


class w16_scope:
    """`with w16_scope() as sc:` -- packed fragments made inside are reused inside (same tensor OBJECT only) and dropped at exit;
    `sc.pair(W)` = what was packed for W (or None); `w16_scope(seed=(W, pair))` starts with a pair made earlier (the forward's)."""

    def __init__(self, seed=None):
        self.map = {}
        if seed is not None and seed[1] is not None:
            self.map[id(seed[0])] = (seed[0], seed[1])

    def __enter__(self):
        self.prev = getattr(_W16_TLS, "scope", None)
        _W16_TLS.scope = self
        return self

    def __exit__(self, *exc):
        _W16_TLS.scope = self.prev
        self.map = {k: v for k, v in self.map.items()}     # (kept readable after exit: forward reads sc.pair(W))
        return False

    def pair(self, W):
        hit = self.map.get(id(W))
        return hit[1] if hit is not None and hit[0] is W else None


def _packed_w16(W):
    sc = getattr(_W16_TLS, "scope", None)
    if sc is not None:
        hit = sc.pair(W)
        if hit is not None:
            return hit
    Wp, Wtp = torch.empty_like(W), torch.empty_like(W)
    with _on(W.device):
        _check(lib().rgcn_pack_w16_pair_f32(_dp(W), _dp(Wp), _dp(Wtp), c_i32(W.shape[0]), _stream(W.device)), "pack_w16_pair")
    if sc is not None:
        sc.map[id(W)] = (W, (Wp, Wtp))
    return Wp, Wtp


def pack_w16(W):
    """[R,16,16] weights -> MFMA fragment order (one float4 per lane), see rgcn_pack_w16_f32 (packed together with the
    transposed fragments, rgcn_pack_w16_pair_f32; remembered inside a w16_scope only)"""
    return _packed_w16(W)[0]


def _spmm_launch(X, W, bias, plan, out, flags, u0, u1, n_split, tag):
    R, d_in, d_out = plan.num_rels, X.shape[1], out.shape[1]
    units = plan.units if u0 == 0 else plan.units[u0:]
    with _on(X.device), _timed(tag):
        _check(lib().rgcn_spmm_f32(_dp(X), _dp(W), _dp(bias), _dp(out), _dp(plan.src), _dp(plan.dst), _dp(plan.val),
                                   _dp(plan.pack), _dp(plan.chunk_rel), _dp(units), c_i64(u1 - u0), c_i64(n_split),
                                   c_i32(plan.tile_rows), c_i64(plan.n_dst), c_i64(plan.n_src), c_i32(R), c_i32(d_in),
                                   c_i32(d_out), c_i32(flags), _stream(X.device)), "spmm")


def pack_w_blocks(W):
    """[R, 16 NI, 16 NJ] weights -> per-(input block, output block) MFMA fragments for the wide spmm kernel"""
    Wp = torch.empty_like(W)
    with _on(W.device):
        _check(lib().rgcn_pack_w_blocks_f32(_dp(W), _dp(Wp), c_i32(W.shape[0]), c_i32(W.shape[1]), c_i32(W.shape[2]),
                                            _stream(W.device)), "pack_w_blocks")
    return Wp


def _spmm_prepare(X, W, bias, plan):
    _req(X, "features"); _req(W, "weights"); _req(bias, "bias")
    R, d_in, d_out = W.shape
    assert X.shape == (plan.n_src, d_in), f"features {tuple(X.shape)} vs ({plan.n_src}, {d_in})"
    assert R == plan.num_rels
    flags = 0
    if plan.pack is not None:
        if d_in == 16 and d_out == 16:
            W = pack_w16(W)
            flags |= F_WPACKED
        elif d_in % 16 == 0 and d_out % 16 == 0 and d_in <= 64 and d_out <= 64:
            W = pack_w_blocks(W)
            flags |= F_WPACKED
    return W, flags


def spmm(X, W, bias, plan, relu=False):
    """out[n_dst, d_out] = bias + sum_slots val * X[src] @ W[rel]"""
    W, flags = _spmm_prepare(X, W, bias, plan)
    out = torch.empty((plan.n_dst, W.shape[2] if W.dim() == 3 else 16), device=X.device, dtype=torch.float32)
    if relu and not plan.n_split:
        flags |= F_RELU
    _spmm_launch(X, W, bias, plan, out, flags, 0, plan.n_units, plan.n_split, "spmm")
    return torch.relu_(out) if (relu and plan.n_split) else out


def slab_bounds(plan, n_slabs):
    """Cut the destination tiles into <= n_slabs contiguous groups of EQUAL TILE COUNT:
    [(unit_begin, unit_end, row_begin, row_end)].  The row boundaries depend only on (n_dst, tile_rows), so every
    rank of a relation-sharded layer -- each with its own messages -- all-reduces identical row ranges."""
    key = ("slabs", n_slabs)
    if getattr(plan, "_cache", None) is None:
        plan._cache = {}
    cache = plan._cache
    if key not in cache:
        u = plan.units_host[:plan.n_units]
        tile_cuts = sorted({(plan.n_tiles * k) // n_slabs for k in range(n_slabs + 1)})
        out = []
        for ta, tb in zip(tile_cuts[:-1], tile_cuts[1:]):
            a = int(np.searchsorted(u[:, 0], ta, side="left"))
            b = int(np.searchsorted(u[:, 0], tb, side="left"))
            out.append((a, b, ta * plan.tile_rows, min(plan.n_dst, tb * plan.tile_rows)))
        cache[key] = out
    return cache[key]


def spmm_slabs(X, W, bias, plan, n_slabs, after_slab):
    """spmm launched slab by slab (whole tiles); after_slab(out, row_begin, row_end) is called right after each
    launch -- the relation-sharded layer starts an asynchronous all-reduce of those rows there, so the
    collective of slab k overlaps the kernels of slab k+1."""
    W, flags = _spmm_prepare(X, W, bias, plan)
    alloc = torch.zeros if plan.n_split else torch.empty
    out = alloc((plan.n_dst, W.shape[2] if W.dim() == 3 else 16), device=X.device, dtype=torch.float32)
    for (u0, u1, r0, r1) in slab_bounds(plan, n_slabs):
        _spmm_launch(X, W, bias, plan, out, flags, u0, u1, 0, "spmm_slab")
        after_slab(out, r0, r1)
    return out


def _slot_perm(p, n_msg, dev):
    """destination-major position -> slot of the relation-major plan `p` (cached on the plan)"""
    if getattr(p, "_inv", None) is None:
        dst, aux = p.dst[:p.m_pad], p.aux[:p.m_pad]        # (a plan without messages keeps one-element placeholders nobody wrote)
        live = dst >= 0
        inv = torch.zeros(max(n_msg, 1), dtype=torch.int32, device=dev)
        inv[aux[live].long()] = torch.arange(p.m_pad, device=dev, dtype=torch.int32)[live]
        p._inv = inv
    return p._inv


def _segment_gather_sum(Y, perm, csr, bias, out, relu):
    """pass 2 of the two-pass routes: out[row] = bias + sum of the row's Y rows (through perm).  Static graphs with hub rows (longer
    than 512 entries) go over work units -- the pieces of a hub row merge with atomics; returns whether a ReLU is still owed"""
    units, n_units, n_split = _csr_units(csr)
    if units is not None and n_split:
        _check(lib().rgcn_segment_gather_sum_units_f32(_dp(Y), _dp(perm), _dp(units), c_i64(n_units), c_i64(n_split), _dp(bias), _dp(out),
                                                       c_i64(csr.n_rows), c_i32(16), c_i32(0), _stream(Y.device)), "segment_gather_sum_units")
        return relu
    _check(lib().rgcn_segment_gather_sum_f32(_dp(Y), _dp(perm), _dp(csr.rowptr), _dp(bias), _dp(out), c_i64(csr.n_rows), c_i32(16),
                                             c_i32(F_RELU if relu else 0), _stream(Y.device)), "segment_gather_sum")
    return False


def spmm_two_pass(X, W, bias, scatter_plan, csr, relu=False):
    """sparse-bucket path (d = 16): relation-major transform, then per-destination sum.  Default: pass 1 writes its rows
    in slot order (sequential, full lines) and pass 2 gathers them through a permutation; RGCN_TWOPASS=scatter: pass 1
    scatters the rows to destination-major positions and pass 2 streams them."""
    _req(X, "features"); _req(W, "weights"); _req(bias, "bias")
    Wp = pack_w16(W)
    p = scatter_plan
    n_msg = int(csr.rowptr[-1].item()) if csr.n_messages is None else csr.n_messages
    out = torch.empty((csr.n_rows, 16), device=X.device, dtype=torch.float32)
    gather = routes.get("twopass", "gather") == "gather"
    if gather:
        _slot_perm(p, n_msg, X.device)
        Y = torch.empty((max(p.dst.shape[0], 1), 16), device=X.device, dtype=torch.float32)
    else:
        Y = torch.empty((max(n_msg, 1), 16), device=X.device, dtype=torch.float32)
    with _on(X.device), _timed("spmm_scatter"):
        _check(lib().rgcn_spmm_scatter_f32(_dp(X), _dp(Wp), _dp(Y), _dp(p.src), _dp(p.val), None if gather else _dp(p.aux),
                                           _dp(p.chunk_rel), _dp(p.items), c_i64(p.n_items), c_i32(16), _stream(X.device)),
               "spmm_scatter")
    with _on(X.device), _timed("segment_sum"):
        if gather:
            relu = _segment_gather_sum(Y, p._inv, csr, bias, out, relu)
        else:
            _check(lib().rgcn_segment_sum_f32(_dp(Y), _dp(csr.rowptr), _dp(bias), _dp(out), c_i64(csr.n_rows), c_i32(16),
                                              c_i32(F_RELU if relu else 0), _stream(X.device)), "segment_sum")
            relu = False
    return out.relu_() if relu else out


def spmm_wide_two_pass(X, W, bias, scatter_plan, csr, relu=False):
    """out = bias + sum val X[src] @ W[rel] for undecomposed weights of any width: relation-grouped gather-GEMM on the
    matrix cores (rgcn_rel_rows_f32; work items of <= 128 slots) + per-destination sum of the rows"""
    _req(X, "features"); _req(W, "weights"); _req(bias, "bias")
    p, dev = scatter_plan, X.device
    R, d_in, d_out = W.shape
    n_msg = int(csr.rowptr[-1].item()) if csr.n_messages is None else csr.n_messages
    perm = _slot_perm(p, n_msg, dev)
    Y = torch.empty((max(p.dst.shape[0], 1), d_out), device=dev, dtype=torch.float32)
    out = torch.empty((csr.n_rows, d_out), device=dev, dtype=torch.float32)
    with _on(dev), _timed("rel_rows"):
        _check(lib().rgcn_rel_rows_f32(_dp(X), _dp(W), _dp(Y), _dp(p.src), _dp(p.val), _dp(p.chunk_rel), _dp(p.items),
                                       c_i64(p.n_items), c_i32(R), c_i32(d_in), c_i32(d_out), _stream(dev)), "rel_rows")
    with _on(dev), _timed("segment_sum_wide"):
        _check(lib().rgcn_segment_gather_sum_wide_f32(_dp(Y), _dp(perm), _dp(csr.rowptr), _dp(bias), _dp(out), c_i64(csr.n_rows),
                                                      c_i32(d_out), c_i32(F_RELU if relu else 0), _stream(dev)),
               "segment_gather_sum_wide")
    return out


def wgrad_wide(X, G, scatter_plan, num_rels):
    """dW[R, d_in, d_out] = sum_slots val X[src]^T G[dst] on the matrix cores (rgcn_rel_wgrad_f32), any width"""
    _req(X, "features"); _req(G, "grad_output")
    p = scatter_plan
    dW = torch.empty((num_rels, X.shape[1], G.shape[1]), device=X.device, dtype=torch.float32)
    with _on(X.device), _timed("rel_wgrad"):
        _check(lib().rgcn_rel_wgrad_f32(_dp(X), _dp(G), _dp(dW), _dp(p.src), _dp(p.dst), _dp(p.val), _dp(p.chunk_rel),
                                        _dp(p.items), c_i64(p.n_items), c_i32(num_rels), c_i32(X.shape[1]), c_i32(G.shape[1]),
                                        _stream(X.device)), "rel_wgrad")
    return dW


def bwd_two_pass_fused(G, X, W, scatter_plan, csr, relu=False):
    """(dX, dW) of the hidden-16 layer on a sparse-bucket graph: relation-major pass producing the transformed rows AND dW
    (rgcn_bwd_scatter_dw_f32), then the per-destination sum of the rows.  relu: X = relu(.) of the producing layer -- dX comes back
    masked with X > 0 (RGCN_F_RELU)."""
    _req(G, "grad_output"); _req(X, "features"); _req(W, "weights")
    p = scatter_plan
    dev = G.device
    Wtp = pack_w16t(W)
    n_msg = int(csr.rowptr[-1].item()) if csr.n_messages is None else csr.n_messages
    _slot_perm(p, n_msg, dev)
    Y = torch.empty((max(p.dst.shape[0], 1), 16), device=dev, dtype=torch.float32)
    dW = torch.empty_like(W)
    dX = torch.empty((csr.n_rows, 16), device=dev, dtype=torch.float32)
    with _on(dev), _timed("bwd_scatter_dw"):
        _check(lib().rgcn_bwd_scatter_dw_f32(_dp(G), _dp(X), _dp(Wtp), _dp(Y), _dp(dW), _dp(p.src), _dp(p.dst), _dp(p.val),
                                             _dp(p.chunk_rel), _dp(p.items), c_i64(p.n_items), c_i32(W.shape[0]), c_i32(16),
                                             c_i32(F_RELU if relu else 0), _stream(dev)), "bwd_scatter_dw")
    with _on(dev), _timed("segment_sum"):
        _segment_gather_sum(Y, p._inv, csr, None, dX, False)
    return dX, dW


def wgrad(X, G, plan, num_rels):
    """dW[R, d_in, d_out] = sum_slots val * X[src]^T G[dst], grouped by relation"""
    _req(X, "features"); _req(G, "grad_output")
    d_in, d_out = X.shape[1], G.shape[1]
    assert X.shape[0] == plan.n_src and G.shape[0] == plan.n_dst
    dW = torch.empty((num_rels, d_in, d_out), device=X.device, dtype=torch.float32)
    with _on(X.device), _timed("wgrad"):
        _check(lib().rgcn_wgrad_f32(_dp(X), _dp(G), _dp(dW), _dp(plan.src), _dp(plan.dst), _dp(plan.val),
                                    _dp(plan.chunk_rel), _dp(plan.items), c_i64(plan.n_items), c_i64(plan.n_dst),
                                    c_i64(plan.n_src), c_i32(num_rels), c_i32(d_in), c_i32(d_out),
                                    _stream(X.device)), "wgrad")
    return dW


def wgrad_tiled(X, G, plan, num_rels, tiles_per_item=4):
    """same result as wgrad(), walking the forward (destination-tile) plan; d_in = d_out = 16 only"""
    _req(X, "features"); _req(G, "grad_output")
    assert plan.run_ptr is not None and X.shape[0] == plan.n_src and G.shape[0] == plan.n_dst
    dW = torch.empty((num_rels, X.shape[1], G.shape[1]), device=X.device, dtype=torch.float32)
    with _on(X.device), _timed("wgrad"):
        _check(lib().rgcn_wgrad_tiled_f32(_dp(X), _dp(G), _dp(dW), _dp(plan.src), _dp(plan.dst), _dp(plan.val),
                                          _dp(plan.chunk_rel), _dp(plan.run_ptr), c_i64(plan.n_tiles),
                                          c_i32(plan.tile_rows), c_i64(plan.n_dst), c_i64(plan.n_src),
                                          c_i32(num_rels), c_i32(X.shape[1]), c_i32(G.shape[1]),
                                          c_i32(tiles_per_item), _stream(X.device)), "wgrad_tiled")
    return dW


F_DW_ATOMIC = 4
F_TRANSPOSE_W = 8


def pack_w16t(W):
    """[R,16,16] weights -> fragments of W^T (what the feature-gradient kernels multiply by), see rgcn_pack_w16t_f32 (cached
    with the forward fragments)"""
    return _packed_w16(W)[1]


def bwd_route():
    """RGCN_BWD_KERNEL: blk (default: block-tile kernel where it applies, else lean) | lean -- DESIGN.md 4.2"""
    return routes.get("bwd_kernel", "blk")


_BLK_MIN_NODES = 32768      # below: a tile per workgroup leaves most CUs idle; the wave-owned 64-row (or smaller) tiles stay
_BLK_MIN_NODES_SPARSE = 4096   # graphs with sparse (tile, relation) buckets: the alternative is the two-pass backward (three launches)


def bwd_blk_rows(n_nodes, num_rels, deterministic=False, device=None, diag4=False, sparse=False):
    """tile height of the transposed plan for the block-tile backward kernel (rgcn_bwd_blk_f32), 0 when it does not apply:
    the tallest tile the kernel's LDS holds next to the relations' dW (rgcn_bwd_blk_max_rows: 192 bytes per row) that gives every CU
    the same number of tiles (S1, R = 101: up to 227 rows -> 218 rows, 4588 tiles, 17.9 per CU).
    diag4: the weights are block_diag() of 4 x 4 blocks (only the diagonal blocks of dW are kept: AM's 267 relations leave 406 rows)"""
    if bwd_route() != "blk" or deterministic or n_nodes < (_BLK_MIN_NODES_SPARSE if sparse else _BLK_MIN_NODES):
        return 0
    cap = int(lib().rgcn_bwd_blk_max_rows(c_i32(num_rels), c_i32(F_DIAG4 if diag4 else 0)))
    cap = min(cap, int(routes.get("bwd_blk_cap", "512")))
    if cap < 64:
        return 0
    n_cu = torch.cuda.get_device_properties(device if device is not None else torch.cuda.current_device()).multi_processor_count
    per_cu = -(-n_nodes // (n_cu * cap))
    rows = -(-n_nodes // (n_cu * per_cu))
    return max(rows, min(cap, 128))      # small graphs: fewer, taller tiles (fuller buckets, fewer dW flushes) rather than one per CU


def poison_lds(device=None):
    """test helper: every CU's LDS filled with NaN patterns (rgcn_poison_lds) -- a kernel that reads LDS words it never wrote shows it"""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with _on(dev):
        _check(lib().rgcn_poison_lds(_stream(dev)), "poison_lds")


def spmm_blk_rows(n_nodes, device=None):
    """tile height of the forward plan for the block-tile FORWARD kernel (rgcn_spmm_blk_f32): the tallest tile up to 1000 rows that gives
    every CU the same number of tiles; 0 for graphs too small to fill the chip with one tile per workgroup"""
    if n_nodes < _BLK_MIN_NODES or routes.get("spmm_csr", "1") == "0":
        return 0
    cap = min(1000, int(lib().rgcn_spmm_blk_max_rows()), int(routes.get("fwd_rows_cap", "1000")))
    n_cu = torch.cuda.get_device_properties(device if device is not None else torch.cuda.current_device()).multi_processor_count
    per_cu = -(-n_nodes // (n_cu * cap))
    return -(-n_nodes // (n_cu * per_cu))


def spmm_blk(X, W, bias, plan, relu=False):
    """out [n_dst, 16] = bias + sum val X[src] W_r on a forward plan of tall tiles, one launch (rgcn_spmm_blk_f32)"""
    _req(X, "features"); _req(W, "weights"); _req(bias, "bias")
    assert W.shape[1:] == (16, 16) and X.shape == (plan.n_src, 16)
    dev = X.device
    Wp = pack_w16(W)
    out = torch.empty((plan.n_dst, 16), device=dev, dtype=torch.float32)
    rec = _blk_rec(plan)
    units, n_units, n_split = _blk_units(plan)
    if relu and n_split:
        raise NativeLibraryError("spmm_blk: relu in the epilogue needs a plan without hub pieces")
    with _on(dev), _timed("spmm_blk"):
        _check(lib().rgcn_spmm_blk_f32(_dp(X), _dp(Wp), _dp(bias), _dp(out), _dp(rec), _dp(plan.run_ptr), c_i64(plan.n_tiles),
                                       c_i32(plan.tile_rows), c_i64(plan.n_dst), c_i32(W.shape[0]), c_i32(F_RELU if relu else 0),
                                       _dp(units), c_i64(n_units), c_i64(n_split), _stream(dev)), "spmm_blk")
    return out


def _bwd_blk_plan(plan, diag4=False):
    return (plan.tile_rows > 160 or (plan.tile_rows > 64 and bwd_route() == "blk")) and \
        bool(lib().rgcn_bwd_blk_supported(c_i32(plan.tile_rows), c_i32(plan.num_rels), c_i32(F_DIAG4 if diag4 else 0)))


def bwd_fused_ok(plan, diag4=False):
    """the fused backward kernels walk the transposed plan tile by tile: packed slots and run pointers.  The wave-owned forms
    take a whole tile per wave (no hub-split work units, at most 160 rows: dX tiles + scratch + staging within the LDS); the
    block-tile form deals a tile's chunks to 16 waves itself and ignores the work units"""
    if plan.run_ptr is None or plan.n_tiles <= 0 or plan.n_src >= (1 << 24):
        return False
    if _bwd_blk_plan(plan, diag4):
        return True             # (tiles above 255 rows have no packed slots: the lean slots are made from the unpacked arrays)
    # (the lean window kernel: its LDS holds the dX tile + X tile + scratch of at least 8 waves; round 2's staging kernel, which took the
    # taller wave-owned tiles up to 160 rows, is gone -- such plans only come from experimental routes and take the two-pass backward)
    return plan.pack is not None and plan.n_split == 0 and plan.n_units == plan.n_tiles and \
        bool(lib().rgcn_bwd_lean_supported(c_i32(plan.tile_rows))) and plan.num_rels < 65536


def _blk_units(plan):
    """work units of the block-tile backward: one per tile, tiles holding hub rows (more than 2x the mean chunk count, at least
    256 chunks) cut into pieces that different workgroups walk -> (units or None, n_units, n_split); plans built without host
    statistics (sync-free per-call graphs) have none (one unit per tile)"""
    cached = getattr(plan, "_blk_units", None)
    if cached is None:
        if getattr(plan, "units_host", None) is None:
            cached = (None, plan.n_tiles, 0)
        else:
            mean = max(1, plan.n_chunks // max(plan.n_tiles, 1))
            cached = row_units(plan.tile_ptr, plan.n_tiles, max(256, 2 * mean))   # (pieces of 1 / 2 / 4 / 8 means measured: 2 is best)
            if cached[2] == 0:
                cached = (None, plan.n_tiles, 0)        # no hub tile: the kernel reads the run pointers itself
        plan._blk_units = cached
    return cached


def _lean_plan(plan):
    """(slots, hdr) of the lean backward kernel: the packed transposed plan reformatted once (rgcn_bwd_lean_prepare_f32), cached
    on the plan object -- static (NC) graphs pay it once, per-call (LP) graphs one small launch per step"""
    lean = getattr(plan, "_lean", None)
    if lean is None:
        n_chunks = plan.chunk_rel.shape[0]
        dev = plan.chunk_rel.device
        slots = torch.empty(int(lib().rgcn_bwd_lean_slot_bytes(c_i64(n_chunks))) + 16, device=dev, dtype=torch.uint8)
        hdr = torch.empty(max(n_chunks, 1), device=dev, dtype=torch.int32)
        with _on(dev):
            if plan.pack is not None:
                _check(lib().rgcn_bwd_lean_prepare_f32(_dp(plan.pack), _dp(plan.chunk_rel), c_i64(n_chunks), _dp(slots), _dp(hdr), _stream(dev)),
                       "bwd_lean_prepare")
            else:
                _check(lib().rgcn_bwd_lean_prepare_unpacked_f32(_dp(plan.src), _dp(plan.dst), _dp(plan.val), c_i32(plan.tile_rows),
                                                                _dp(plan.chunk_rel), c_i64(n_chunks), _dp(slots), _dp(hdr), _stream(dev)),
                       "bwd_lean_prepare_unpacked")
        lean = plan._lean = (slots, hdr)
    return lean


def _blk_rec(plan):
    """chunk records of the block-tile backward kernel (rgcn_bwd_blk_prepare_f32: 176 bytes per chunk), made once per plan and
    cached on it -- static (NC) graphs pay it once, per-call (LP) graphs one small launch per step"""
    rec = getattr(plan, "_blk_rec", None)
    if rec is None:
        n_chunks = plan.chunk_rel.shape[0]
        dev = plan.chunk_rel.device
        rec = torch.empty(int(lib().rgcn_bwd_blk_rec_bytes(c_i64(n_chunks))) + 16, device=dev, dtype=torch.uint8)
        packed = plan.pack is not None and plan.tile_rows <= 255
        with _on(dev):
            _check(lib().rgcn_bwd_blk_prepare_f32(_dp(plan.pack) if packed else None, _dp(plan.src), _dp(plan.dst), _dp(plan.val),
                                                  c_i32(plan.tile_rows), _dp(plan.chunk_rel), c_i64(n_chunks), _dp(rec), _stream(dev)),
                   "bwd_blk_prepare")
        plan._blk_rec = rec
    return rec


def bwd_own_geometry():
    """(waves per workgroup, relation slots per wave, tallest tile) of the relation-owner backward kernel (rgcn_bwd_own_f32)"""
    L = lib()
    nw = int(L.rgcn_bwd_own_waves())
    return nw, int(L.rgcn_bwd_own_units()) // nw, int(L.rgcn_bwd_own_max_rows())


def bwd_own_rows(n_nodes, device=None):
    """tile height of the relation-owner backward's plan: the tallest tile the kernel's LDS holds that gives every CU the same number of
    tiles (S1: 782 rows, 5 tiles per CU); 0 for graphs too small to fill the chip with one tile per workgroup"""
    if n_nodes < _BLK_MIN_NODES:
        return 0
    cap = min(bwd_own_geometry()[2], int(routes.get("own_rows_cap", "789")))
    n_cu = torch.cuda.get_device_properties(device if device is not None else torch.cuda.current_device()).multi_processor_count
    per_cu = -(-n_nodes // (n_cu * cap))
    return -(-n_nodes // (n_cu * per_cu))


def bwd_own(G, X, W, plan, relu=False, want_db=False):
    """(dX, dW[, db]) of the hidden-16 layer from one walk of a soft-window plan with relation ownership (build_softwin_plan(own_waves=...)):
    rgcn_bwd_own_f32 -- dX tile in LDS doubles, every relation's dW in the registers of its owner wave, X rows of the tile from global memory.
    relu: X is the output of a ReLU and dX is wanted BEFORE it.  Sums in arrival order (not bit-reproducible)."""
    _req(G, "grad_output"); _req(X, "features"); _req(W, "weights")
    assert W.shape[1:] == (16, 16) and G.shape == (plan.n_src, 16) and X.shape == (plan.n_dst, 16) and W.shape[0] == plan.num_rels
    assert getattr(plan, "own_ptr", None) is not None, "bwd_own: a plan made with build_softwin_plan(own_waves=...)"
    dev = G.device
    Wtp = pack_w16t(W)
    dX = torch.empty((plan.n_dst, 16), device=dev, dtype=torch.float32)
    buf = torch.empty(W.numel() + 16, device=dev, dtype=torch.float32)       # dW and db back to back: one fill zeroes both
    dW, db = buf[:W.numel()].view_as(W), buf[W.numel():]
    rec = _blk_rec(plan)
    with _on(dev), _timed("bwd_fused"):
        _check(lib().rgcn_bwd_own_f32(_dp(G), _dp(X), _dp(Wtp), _dp(dX), _dp(dW), _dp(rec), _dp(plan.own_ptr), _dp(plan.unit_rel),
                                      c_i64(plan.n_tiles), c_i32(plan.tile_rows), c_i64(plan.n_dst), c_i32(W.shape[0]),
                                      c_i32(F_RELU if relu else 0), _dp(db) if want_db else None, c_i64(plan.n_src), _stream(dev)), "bwd_own")
    return (dX, dW, db if want_db else None)


def bwd_fused_relu_ok(plan, diag4=False):
    """RGCN_F_RELU (dX masked with X > 0 in the epilogue): both fused kernels have it"""
    return bool(_bwd_blk_plan(plan, diag4)) or bwd_fused_ok(plan, diag4)


def bwd_fused(G, X, W, plan, atomic=False, relu=False, want_db=False, diag4=False):
    """(dX [n, 16], dW [R, 16, 16]) of the hidden-16 layer from one walk of the transposed plan:
    G upstream gradient, X the layer's input, W [R, 16, 16] (rgcn_bwd_blk_f32 on a plan of tall tiles, rgcn_bwd_lean_f32 on wave-owned ones).  relu: X is the output of a ReLU and dX is wanted BEFORE it
    (rows masked with X > 0 in the kernel's epilogue).  want_db: returns (dX, dW, db) -- db [16] = column sums of G when the
    kernel that ran computes them on the side (block-tile kernel), else None (the caller launches colsum).  diag4: W is
    block_diag() of 4 x 4 blocks and only the diagonal blocks of dW are wanted (block-tile kernel only: up to 447 relations)."""
    _req(G, "grad_output"); _req(X, "features"); _req(W, "weights")
    assert W.shape[1:] == (16, 16) and G.shape == (plan.n_src, 16) and X.shape == (plan.n_dst, 16)
    dev = G.device
    Wtp = pack_w16t(W)
    dX = torch.empty((plan.n_dst, 16), device=dev, dtype=torch.float32)
    blk = _bwd_blk_plan(plan, diag4)
    if diag4 and not blk:
        raise NativeLibraryError("bwd_fused: diag4 needs the block-tile kernel (plan of tall tiles, non-deterministic mode)")
    db = None
    if blk and want_db:      # dW and db back to back: one fill zeroes both
        buf = torch.empty(W.numel() + 16, device=dev, dtype=torch.float32)
        dW, db = buf[:W.numel()].view_as(W), buf[W.numel():]
    else:
        dW = torch.empty_like(W)
    ret = (lambda: (dX, dW, db)) if want_db else (lambda: (dX, dW))
    scratch = None
    if not atomic:
        n = int(lib().rgcn_bwd_fused_scratch_floats(c_i64(plan.n_tiles), c_i32(W.shape[0])))
        scratch = torch.empty(n, device=dev, dtype=torch.float32)
    if blk:     # a plan of tall tiles (graph.bwd_plan asked bwd_blk_rows): one tile per workgroup
        if not atomic:
            raise NativeLibraryError("bwd_fused: the block-tile plan (tall tiles) has no bit-reproducible kernel")
        rec = _blk_rec(plan)
        units, n_units, n_split = _blk_units(plan)
        with _on(dev), _timed("bwd_fused"):
            _check(lib().rgcn_bwd_blk_f32(_dp(G), _dp(X), _dp(Wtp), _dp(dX), _dp(dW), _dp(rec), _dp(plan.run_ptr),
                                          c_i64(plan.n_tiles), c_i32(plan.tile_rows), c_i64(plan.n_dst), c_i32(W.shape[0]),
                                          c_i32((F_RELU if relu else 0) | (F_DIAG4 if diag4 else 0)), _dp(db), c_i64(plan.n_src),
                                          _dp(units), c_i64(n_units), c_i64(n_split), _stream(dev)), "bwd_blk")
        return ret()
    if not (lib().rgcn_bwd_lean_supported(c_i32(plan.tile_rows)) and W.shape[0] < 65536):
        raise NativeLibraryError(f"bwd_fused: no fused kernel for wave-owned tiles of {plan.tile_rows} rows (bwd_fused_ok)")
    slots, hdr = _lean_plan(plan)
    with _on(dev), _timed("bwd_fused"):
        _check(lib().rgcn_bwd_lean_f32(_dp(G), _dp(X), _dp(Wtp), _dp(dX), _dp(dW), _dp(scratch), _dp(slots), _dp(hdr),
                                       _dp(plan.run_ptr), c_i64(plan.n_tiles), c_i32(plan.tile_rows), c_i64(plan.n_dst),
                                       c_i32(W.shape[0]), c_i32((F_DW_ATOMIC if atomic else 0) | (1 if relu else 0)),
                                       _stream(dev)), "bwd_lean")
    return ret()


def bwd_fused_slabs_ok(plan):
    """the slab-by-slab form of the fused backward: the block-tile kernel on a plan of tall tiles without hub pieces"""
    return bool(_bwd_blk_plan(plan)) and _blk_units(plan)[2] == 0 and getattr(plan, "tile_ptr", None) is not None


def bwd_fused_slabs(G, X, W, plan, n_slabs, after_slab):
    """bwd_fused launched slab by slab (whole tiles of the transposed plan): dX rows [r0, r1) are complete after slab k's launch and
    after_slab(dX, r0, r1) is called right there -- the relation-sharded layer starts the asynchronous all-reduce of those rows while
    slab k + 1's kernel runs; dW and the bias gradient keep accumulating over the slabs (RGCN_F_PARTIAL, then RGCN_F_ACCUMULATE).
    -> (dX, dW, db)"""
    _req(G, "grad_output"); _req(X, "features"); _req(W, "weights")
    assert W.shape[1:] == (16, 16) and bwd_fused_slabs_ok(plan)
    dev = G.device
    Wtp = pack_w16t(W)
    dX = torch.empty((plan.n_dst, 16), device=dev, dtype=torch.float32)
    buf = torch.empty(W.numel() + 16, device=dev, dtype=torch.float32)
    dW, db = buf[:W.numel()].view_as(W), buf[W.numel():]
    rec = _blk_rec(plan)
    tiles = getattr(plan, "_blk_tile_units", None)
    if tiles is None:        # one explicit work unit per tile (the default launch lets the kernel read the run pointers itself)
        tiles = plan._blk_tile_units = row_units(plan.tile_ptr, plan.n_tiles, 1 << 30)[0]
    cuts = sorted({(plan.n_tiles * k) // n_slabs for k in range(n_slabs + 1)})
    for i, (ta, tb) in enumerate(zip(cuts[:-1], cuts[1:])):
        with _on(dev), _timed("bwd_fused"):
            _check(lib().rgcn_bwd_blk_f32(_dp(G), _dp(X), _dp(Wtp), _dp(dX), _dp(dW), _dp(rec), _dp(plan.run_ptr),
                                          c_i64(plan.n_tiles), c_i32(plan.tile_rows), c_i64(plan.n_dst), c_i32(W.shape[0]),
                                          c_i32(F_PARTIAL if i == 0 else F_ACCUMULATE), _dp(db) if i == 0 else None, c_i64(plan.n_src),
                                          c_void_p(tiles.data_ptr() + 16 * ta), c_i64(tb - ta), c_i64(0), _stream(dev)), "bwd_blk")
        after_slab(dX, ta * plan.tile_rows, min(plan.n_dst, tb * plan.tile_rows))
    return dX, dW, db


def featureless_fwd(table, bias, plan):
    """out[n_dst, d] = bias + sum_slots val * table[rel, src, :]   (table: [R, n_src, d])"""
    _req(table, "weights"); _req(bias, "bias")
    R, n_src, d = table.shape
    assert R == plan.num_rels and n_src == plan.n_src
    out = torch.empty((plan.n_dst, d), device=table.device, dtype=torch.float32)
    with _on(table.device), _timed("featureless_fwd"):
        _check(lib().rgcn_featureless_fwd_f32(_dp(table), _dp(bias), _dp(out), _dp(plan.src), _dp(plan.dst),
                                              _dp(plan.val), _dp(plan.chunk_rel), _dp(plan.units),
                                              c_i64(plan.n_units), c_i64(plan.n_split), c_i32(plan.tile_rows), c_i64(plan.n_dst),
                                              c_i64(n_src), c_i32(R), c_i32(d), _stream(table.device)),
               "featureless_fwd")
    return out


def featureless_csr_fwd(table, bias, csr, relu=False):
    """out[n_rows, d] = bias + sum over the row's CSR entries of val * table[rel, src, :] (rgcn_featureless_csr_fwd_f32): the
    featureless layer on graphs whose (tile, relation) buckets are sparse -- one lane group per MESSAGE, not per padded plan slot"""
    _req(table, "weights"); _req(bias, "bias")
    R, n_src, d = table.shape
    units, n_units, n_split = _csr_units(csr)
    assert units is not None, "the CSR featureless kernels serve static graphs"
    out = torch.empty((csr.n_rows, d), device=table.device, dtype=torch.float32)
    fused = bool(relu) and n_split == 0          # relu in the epilogue unless hub rows are cut into shared pieces
    with _on(table.device), _timed("featureless_csr_fwd"):
        _check(lib().rgcn_featureless_csr_fwd_f32(_dp(table), _dp(bias), _dp(out), _dp(units), c_i64(n_units), c_i64(n_split),
                                                  _dp(csr.src), _dp(csr.rel), _dp(csr.val), c_i64(csr.n_rows), c_i64(n_src),
                                                  c_i32(R), c_i32(d), c_i32(1 if fused else 0), _stream(table.device)), "featureless_csr_fwd")
    return (out, fused) if relu else out


def featureless_csr_wgrad(G, csr, num_rels, n_src):
    """dtable[R, n_src, d]: += val * G[row, :] at [rel, src] for every CSR entry (rgcn_featureless_csr_wgrad_f32)"""
    _req(G, "grad_output")
    d = G.shape[1]
    if getattr(csr, "n_entries", None) is None:
        csr.n_entries = int(csr.rowptr[-1].item())       # static graph: read once
    dT = torch.empty((num_rels, n_src, d), device=G.device, dtype=torch.float32)
    with _on(G.device), _timed("featureless_csr_wgrad"):
        _check(lib().rgcn_featureless_csr_wgrad_f32(_dp(G), _dp(dT), _dp(csr.rowptr), _dp(csr.src), _dp(csr.rel), _dp(csr.val),
                                                    c_i64(csr.n_entries), c_i64(csr.n_rows), c_i64(n_src), c_i32(num_rels), c_i32(d),
                                                    _stream(G.device)), "featureless_csr_wgrad")
    return dT


def _csr_units(csr):
    """(units, n_units, n_split) of a CSR: one unit per row, hub rows cut into 512-entry pieces -- computed once per static
    graph (host statistics); a CSR of a per-call LP graph has none: (None, n_rows, 0) = every row is one unit.
    RGCN_DETERMINISTIC=1: no pieces (they merge with fp32 atomics, in arrival order) -- a hub row is then one long unit"""
    if getattr(csr, "sync_free", False) or getattr(csr, "per_call", False):
        return None, csr.n_rows, 0              # (per-call graphs of the LP layer: the statistics would cost two read-backs a step)
    if routes.get("deterministic", "0") == "1":
        if getattr(csr, "units_whole", None) is None:
            csr.units_whole = row_units(csr.rowptr, csr.n_rows, 1 << 30)
        return csr.units_whole
    if getattr(csr, "units", None) is None:
        csr.units = row_units(csr.rowptr, csr.n_rows, 512)
    return csr.units


def _empty_csr_out(csr, width, bias, relu, device):
    """a CSR without a single entry (rows, no messages): out = bias on every row -- the kernels prefetch entry 0 unconditionally"""
    out = torch.zeros((csr.n_rows, width), device=device, dtype=torch.float32)
    if bias is not None:
        out += bias
    return out.relu_() if relu else out


def block_supported(bi, bo):
    return bool(lib().rgcn_block_supported(c_i32(bi), c_i32(bo)))


def block_spmm(X, blocks, bias, csr, transposed=False, relu=False):
    """out[n_rows, nb * bo] = bias + sum over the row's CSR entries of val * X[src] . blockdiag(blocks[rel])   (blocks:
    [R', nb, bi, bo]; entries with rel >= R' are skipped; transposed: X is [., nb * bo] and the blocks are applied
    transposed -> [n_rows, nb * bi])"""
    _req(X, "features"); _req(blocks, "blocks"); _req(bias, "bias")
    Rb, nb, bi, bo = blocks.shape
    assert X.shape[1] == nb * (bo if transposed else bi)
    if csr.src.numel() == 0:
        return _empty_csr_out(csr, nb * (bi if transposed else bo), bias, relu, X.device)
    units, n_units, n_split = _csr_units(csr)
    fuse_relu = relu and n_split == 0
    out = torch.empty((csr.n_rows, nb * (bi if transposed else bo)), device=X.device, dtype=torch.float32)
    flags = (F_TRANSPOSE_W if transposed else 0) | (F_RELU if fuse_relu else 0)
    with _on(X.device), _timed("block_spmm"):
        _check(lib().rgcn_block_spmm_f32(_dp(X), _dp(blocks), _dp(bias), _dp(out), _dp(units), _dp(csr.rowptr), c_i64(n_units),
                                         c_i64(n_split), _dp(csr.src), _dp(csr.rel), _dp(csr.val), c_i64(csr.n_rows), c_i32(Rb),
                                         c_i32(nb), c_i32(bi), c_i32(bo), c_i32(flags), _stream(X.device)), "block_spmm")
    if relu and not fuse_relu:
        out.relu_()
    return out


def spmm_csr_d16_ok(csr, R):
    """dense 16 x 16 weights on the destination-major CSR in one pass: the table of R relations has to fit the LDS (R <= 120) and
    the CSR needs its work units (static graphs)"""
    return not getattr(csr, "sync_free", False) and not getattr(csr, "per_call", False) and \
        bool(lib().rgcn_spmm_csr_d16_supported(c_i32(R)))


def spmm_csr_d16(X, W, bias, csr, relu=False):
    """out[n_rows, 16] = bias + sum over the row's CSR entries of val * X[src] @ W[rel]   (W: [R, 16, 16]; rgcn_spmm_csr_d16_f32)"""
    _req(X, "features"); _req(W, "weights"); _req(bias, "bias")
    assert X.shape[1] == 16 and tuple(W.shape[1:]) == (16, 16)
    num_rels = getattr(csr, "num_rels", None)
    assert num_rels is None or W.shape[0] == num_rels, f"weights of {W.shape[0]} relations on a graph of {num_rels}: the kernel indexes its LDS table by the entries' relation"
    if csr.src.numel() == 0:
        return _empty_csr_out(csr, 16, bias, relu, X.device)
    units, n_units, n_split = _csr_units(csr)
    fuse_relu = relu and n_split == 0
    out = torch.empty((csr.n_rows, 16), device=X.device, dtype=torch.float32)
    with _on(X.device), _timed("spmm_csr"):
        _check(lib().rgcn_spmm_csr_d16_f32(_dp(X), _dp(W), _dp(bias), _dp(out), _dp(units), c_i64(n_units), c_i64(n_split),
                                           _dp(csr.src), _dp(csr.rel), _dp(csr.val), c_i64(csr.n_rows), c_i32(W.shape[0]),
                                           c_i32(F_RELU if fuse_relu else 0), _stream(X.device)), "spmm_csr_d16")
    if relu and not fuse_relu:
        out.relu_()
    return out


def block_wgrad(X, G, scatter_plan, shape):
    """dblocks[R', nb, bi, bo] = sum_slots val * X[src, b, :]^T G[dst, b, :] grouped by relation (relation-major plan)"""
    _req(X, "features"); _req(G, "grad_output")
    p = scatter_plan
    Rb, nb, bi, bo = shape
    dB = torch.empty(shape, device=X.device, dtype=torch.float32)
    with _on(X.device), _timed("block_wgrad"):
        _check(lib().rgcn_block_wgrad_f32(_dp(X), _dp(G), _dp(dB), _dp(p.src), _dp(p.dst), _dp(p.val), _dp(p.chunk_rel),
                                          _dp(p.items), c_i64(p.n_items), c_i32(Rb), c_i32(nb), c_i32(bi), c_i32(bo),
                                          _stream(X.device)), "block_wgrad")
    return dB


def diag_spmm(X, w, bias, csr):
    """out[n_rows, d] = bias + sum over the row's CSR entries of val * X[src, :] * w[rel, :]   (diagonal weights, w: [R, d])"""
    _req(X, "features"); _req(w, "weights"); _req(bias, "bias")
    R, d = w.shape
    assert X.shape[1] == d
    units, n_units, n_split = _csr_units(csr)
    assert units is not None, "the diagonal layer is not part of the sync-free LP step"
    out = torch.empty((csr.n_rows, d), device=X.device, dtype=torch.float32)
    with _on(X.device), _timed("diag_spmm"):
        _check(lib().rgcn_diag_spmm_f32(_dp(X), _dp(w), _dp(bias), _dp(out), _dp(units), c_i64(n_units), c_i64(n_split),
                                        _dp(csr.src), _dp(csr.rel), _dp(csr.val), c_i64(csr.n_rows), c_i32(R), c_i32(d),
                                        _stream(X.device)), "diag_spmm")
    return out


def diag_wgrad(X, G, scatter_plan, num_rels):
    """dw[R, d] = sum_slots val * X[src, :] * G[dst, :] grouped by relation (relation-major plan)"""
    _req(X, "features"); _req(G, "grad_output")
    p, d = scatter_plan, X.shape[1]
    dw = torch.empty((num_rels, d), device=X.device, dtype=torch.float32)
    with _on(X.device), _timed("diag_wgrad"):
        _check(lib().rgcn_diag_wgrad_f32(_dp(X), _dp(G), _dp(dw), _dp(p.src), _dp(p.dst), _dp(p.val), _dp(p.chunk_rel),
                                         _dp(p.items), c_i64(p.n_items), c_i32(num_rels), c_i32(d), _stream(X.device)),
               "diag_wgrad")
    return dw


def featureless_wgrad(G, plan, num_rels):
    _req(G, "grad_output")
    d = G.shape[1]
    dT = torch.empty((num_rels, plan.n_src, d), device=G.device, dtype=torch.float32)
    with _on(G.device), _timed("featureless_wgrad"):
        _check(lib().rgcn_featureless_wgrad_f32(_dp(G), _dp(dT), _dp(plan.src), _dp(plan.dst), _dp(plan.val),
                                                _dp(plan.chunk_rel), c_i64(plan.n_chunks), c_i64(plan.n_dst),
                                                c_i64(plan.n_src), c_i32(num_rels), c_i32(d), _stream(G.device)),
               "featureless_wgrad")
    return dT


def colsum(G):
    _req(G, "grad_output")
    db = torch.empty(G.shape[1], device=G.device, dtype=torch.float32)
    scratch = torch.empty(int(lib().rgcn_colsum_scratch_floats(c_i64(G.shape[0]), c_i32(G.shape[1]))), device=G.device,
                          dtype=torch.float32)
    with _on(G.device), _timed("colsum"):
        _check(lib().rgcn_colsum_f32(_dp(G), _dp(db), _dp(scratch), c_i64(G.shape[0]), c_i32(G.shape[1]),
                                     _stream(G.device)), "colsum")
    return db


def resize3(src, shape, src1=None, n1d=None):
    """zero-pad / crop the two trailing dimensions of `src` ([A, B, C] or [B, C]) to `shape`, and optionally a 1-D tensor to n1d
    elements, in ONE launch (rgcn_resize3_f32) -> dst or (dst, dst1)"""
    _req(src, "tensor"); _req(src1, "vector")
    s3 = src if src.dim() == 3 else src.unsqueeze(0)
    A, B, C = s3.shape
    Bd, Cd = shape[-2], shape[-1]
    dst = torch.empty((A, Bd, Cd) if src.dim() == 3 else (Bd, Cd), device=src.device, dtype=torch.float32)
    dst1 = None if src1 is None else torch.empty(n1d, device=src.device, dtype=torch.float32)
    with _on(src.device):
        _check(lib().rgcn_resize3_f32(_dp(s3), _dp(dst), c_i64(A), c_i32(B), c_i32(C), c_i32(Bd), c_i32(Cd), _dp(src1), _dp(dst1),
                                      c_i32(0 if src1 is None else src1.numel()), c_i32(0 if src1 is None else n1d), _stream(src.device)),
               "resize3")
    return dst if src1 is None else (dst, dst1)


def ce_head(logits, row_label, lab_rows):
    """(loss [1], dlogits [N, C], the zero-padded [N, ld] buffer dlogits is the first columns of -- or None) of the mean cross-entropy
    over the labelled rows (rgcn_ce_head_f32).  logits: rows of C floats at a stride of ld >= C floats (a layer's padded output)."""
    _req(row_label, "row_label", torch.int32); _req(lab_rows, "lab_rows", torch.int32)
    assert logits.dtype == torch.float32 and logits.is_cuda and logits.dim() == 2 and logits.stride(1) == 1 and logits.data_ptr() % 16 == 0
    N, C = logits.shape
    ld = logits.stride(0) if N > 1 else C
    loss = torch.empty(1, device=logits.device, dtype=torch.float32)
    full = torch.empty((N, ld), device=logits.device, dtype=torch.float32)
    with _on(logits.device), _timed("ce_head"):
        _check(lib().rgcn_ce_head_f32(_dp(logits), _dp(row_label), _dp(lab_rows), _dp(loss), _dp(full), c_i64(N), c_i32(C), c_i32(ld),
                                      c_i32(lab_rows.shape[0]), _stream(logits.device)), "ce_head")
    return (loss, full, None) if ld == C else (loss, full[:, :C], full)


_BCE_WS = {}


def bce_head(scores, labels):
    """(loss [1], dscores [T]) of the mean binary cross-entropy with logits (rgcn_bce_head_f32: one launch)"""
    _req(scores, "scores"); _req(labels, "labels")
    assert scores.dim() == 1 and labels.shape == scores.shape
    dev = scores.device
    if scores.shape[0] == 0:      # F.binary_cross_entropy_with_logits of nothing: the mean over zero elements is NaN, the gradient is empty
        return torch.full((1,), float("nan"), device=dev, dtype=torch.float32), torch.empty_like(scores)
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)      # per stream (the ticket and the partials are the launch's own)
    ws = _BCE_WS.get(key)
    if ws is None:        # zeroed once; every launch leaves it zeroed (allocated outside any capture: the first call is the warm-up's)
        ws = _BCE_WS[key] = torch.zeros(int(lib().rgcn_bce_head_workspace_bytes()), dtype=torch.uint8, device=dev)
    loss = torch.empty(1, device=dev, dtype=torch.float32)
    ds = torch.empty_like(scores)
    with _on(dev), _timed("bce_head"):
        _check(lib().rgcn_bce_head_f32(_dp(scores), _dp(labels), _dp(loss), _dp(ds), _dp(ws), c_i64(scores.shape[0]), _stream(dev)), "bce_head")
    return loss, ds


def distmult_fwd(triples, nodes, rel, sbias, pbias, obias, ranks=False):
    """scores [T]; ranks=True: (scores, [row sizes, ranks]) -- the counting pass of the backward's two CSRs done by the scoring kernel
    (see distmult_csrs; the list is consumed by it)"""
    _req(nodes, "nodes"); _req(rel, "relations"); _req(triples, "triples", torch.int64)
    for b, n in ((sbias, "sbias"), (pbias, "pbias"), (obias, "obias")):
        _req(b, n)
    T = triples.shape[0]
    scores = torch.empty(T, device=nodes.device, dtype=torch.float32)
    err = _i32(1, nodes.device)
    counts = rk = None
    if ranks and T:
        counts, rk = _i32(2 * nodes.shape[0] + 3, nodes.device), _i32(2 * T, nodes.device)
    with _on(nodes.device), _timed("distmult_fwd"):
        _check(lib().rgcn_distmult_fwd_f32(_dp(triples), c_i64(T), _dp(nodes), _dp(rel), _dp(sbias), _dp(pbias),
                                           _dp(obias), _dp(scores), c_i64(nodes.shape[0]), c_i32(rel.shape[0]),
                                           c_i32(nodes.shape[1]), _dp(err), _dp(counts), _dp(rk), _stream(nodes.device)),
               "distmult_fwd")
    # the reference indexes nodes[s], relations[p], nodes[o] (layers.py:89-93) and raises IndexError on a bad index
    dev_check_err(err, "DistMult triples (s, o < num_nodes, p < num_relations)", IndexError)
    return (scores, [counts, rk, False]) if ranks else scores


def distmult_score_all(batch, head, nodes, rel, sbias=None, pbias=None, obias=None, out=None):
    """scores [Q, N] of every entity as head (head=True) or tail of each triple of `batch` (int64 [Q,3], device);
    utils/misc.py:71-88 without the [bn, N, 3] candidate tensor."""
    _req(nodes, "nodes"); _req(rel, "relations"); _req(batch, "batch", torch.int64)
    for b, n in ((sbias, "sbias"), (pbias, "pbias"), (obias, "obias")):
        _req(b, n)
    Q, (N, d) = batch.shape[0], nodes.shape
    assert batch.dim() == 2 and batch.shape[1] == 3, "batch must be [Q, 3]"
    assert rel.shape[1] == d, "relation and node embeddings differ in width"
    if Q:
        lo, hi = batch.amin(0), batch.amax(0)
        assert int(lo.min()) >= 0 and int(hi[0]) < N and int(hi[2]) < N and int(hi[1]) < rel.shape[0], \
            "triple index out of range"
    scores = out if out is not None else torch.empty(Q, N, device=nodes.device, dtype=torch.float32)
    assert scores.shape == (Q, N) and scores.is_contiguous() and scores.dtype == torch.float32
    qvec = torch.empty(Q, d, device=nodes.device, dtype=torch.float32)
    qb = torch.empty(2 * Q, device=nodes.device, dtype=torch.float32) if sbias is not None else None
    with _on(nodes.device), _timed("score_all"):
        _check(lib().rgcn_distmult_score_all_f32(_dp(batch), c_i64(Q), c_i32(1 if head else 0), _dp(nodes), _dp(rel),
                                                 _dp(sbias), _dp(pbias), _dp(obias), _dp(qvec), _dp(qb), _dp(scores),
                                                 c_i64(N), c_i32(rel.shape[0]), c_i32(d), _stream(nodes.device)),
               "distmult_score_all")
    return scores


def rank_filter(scores, filt_q, filt_n):
    """scores[filt_q[e], filt_n[e]] = -inf (utils/misc.py:40-58); int32 device index lists"""
    _req(scores, "scores"); _req(filt_q, "filt_q", torch.int32); _req(filt_n, "filt_n", torch.int32)
    assert filt_q.shape == filt_n.shape and filt_q.dim() == 1
    with _on(scores.device):
        _check(lib().rgcn_rank_filter_f32(_dp(scores), c_i64(scores.shape[0]), c_i64(scores.shape[1]), _dp(filt_q),
                                          _dp(filt_n), c_i64(filt_q.shape[0]), _stream(scores.device)), "rank_filter")
    return scores


def rank_count(scores, batch, head):
    """(#scores > target score, #scores == target score) per query, int64 (utils/misc.py:93-96)"""
    _req(scores, "scores"); _req(batch, "batch", torch.int64)
    Q = scores.shape[0]
    assert batch.shape == (Q, 3)
    greater = torch.empty(Q, device=scores.device, dtype=torch.int64)
    ties = torch.empty(Q, device=scores.device, dtype=torch.int64)
    with _on(scores.device), _timed("rank_count"):
        _check(lib().rgcn_rank_count_f32(_dp(scores), _dp(batch), c_i64(Q), c_i32(1 if head else 0),
                                         c_i64(scores.shape[1]), _dp(greater), _dp(ties), _stream(scores.device)),
               "rank_count")
    return greater, ties


def distmult_bwd(triples, nodes, rel, gs, with_bias, nodes_grad=True):
    """relation (and bias) gradients from predicate-sorted triples; nodes_grad=True also scatters the entity gradients with
    atomics (round 1's form; the default path computes them with distmult_bwd_nodes instead)"""
    _req(gs, "grad_scores")
    dn = torch.empty_like(nodes) if nodes_grad else None
    dr = torch.empty_like(rel)
    dsb = torch.empty(nodes.shape[0], device=nodes.device) if with_bias else None
    dob = torch.empty(nodes.shape[0], device=nodes.device) if with_bias else None
    dpb = torch.empty(rel.shape[0], device=nodes.device) if with_bias else None
    with _on(nodes.device), _timed("distmult_bwd"):
        _check(lib().rgcn_distmult_bwd_f32(_dp(triples), c_i64(triples.shape[0]), _dp(nodes), _dp(rel), _dp(gs),
                                           _dp(dn), _dp(dr), _dp(dsb), _dp(dpb), _dp(dob), c_i64(nodes.shape[0]),
                                           c_i32(rel.shape[0]), c_i32(nodes.shape[1]), _stream(nodes.device)),
               "distmult_bwd")
    return dn, dr, dsb, dpb, dob


def distmult_bwd_all_supported(n_rel, d):
    return bool(lib().rgcn_distmult_bwd_all_supported(c_i32(n_rel), c_i32(d)))


def distmult_csrs(triples, ranks, nodes, rel, gs):
    """(rowptr by subject, rowptr by object, entries): the two CSRs of the scored triples the backward kernels walk, 16-byte entries
    {other end, predicate, gs[t], -}, from the row sizes and ranks the scoring kernel left (distmult_fwd(..., ranks=True)): a scan
    and one pass without atomics (rgcn_distmult_csr_place).  The row sizes are scanned in place, once: ranks[2] remembers."""
    counts, rk, scanned = ranks
    N, R, T = nodes.shape[0], rel.shape[0], triples.shape[0]
    dev = nodes.device
    scan_tmp = None if scanned else _i32((2 * N + 2) // 1024 + 4, dev)
    ranks[2] = True
    entries = torch.empty((max(2 * T, 1), 4), dtype=torch.int32, device=dev)
    with _on(dev):
        _check(lib().rgcn_distmult_csr_place(_dp(triples), c_i64(T), c_i64(N), c_i32(R), _dp(rk), _dp(counts), _dp(scan_tmp), _dp(gs),
                                             _dp(entries), _stream(dev)), "distmult_csr_place")
    return counts[1: N + 2], counts[N + 2: 2 * N + 3], entries


def distmult_bwd_all(triples, ranks, nodes, rel, gs, with_bias):
    """(dnodes, drel, dsbias, dpbias, dobias) of DistMult from two CSRs of the scored triples -- no predicate sort, no atomics on
    the entity rows (rgcn_distmult_bwd_all_f32); bias gradients None unless with_bias"""
    _req(gs, "grad_scores"); _req(nodes, "nodes"); _req(rel, "relations")
    N, d = nodes.shape
    R = rel.shape[0]
    dev = nodes.device
    rp_s, rp_o, entries = distmult_csrs(triples, ranks, nodes, rel, gs)
    dn, dr = torch.empty_like(nodes), torch.empty_like(rel)
    dsb = dpb = dob = None
    if with_bias:
        dsb, dob = torch.empty(N, device=dev, dtype=torch.float32), torch.empty(N, device=dev, dtype=torch.float32)
        dpb = torch.empty(R, device=dev, dtype=torch.float32)
    with _on(dev), _timed("distmult_bwd_all"):
        _check(lib().rgcn_distmult_bwd_all_f32(_dp(rp_s), _dp(rp_o), _dp(entries), _dp(nodes), _dp(rel), _dp(dn), _dp(dr), _dp(dsb),
                                               _dp(dpb), _dp(dob), c_i64(N), c_i32(R), c_i32(d), _stream(dev)), "distmult_bwd_all")
    return dn, dr, dsb, dpb, dob


def distmult_bwd_nodes(triples, ranks, nodes, rel, gs):
    """entity gradients of DistMult without atomics: the two CSRs of the scored triples (distmult_csrs) and one wave per entity
    (rgcn_distmult_bwd_nodes_f32)"""
    _req(gs, "grad_scores"); _req(nodes, "nodes"); _req(rel, "relations")
    N, d = nodes.shape
    dev = nodes.device
    rp_s, rp_o, entries = distmult_csrs(triples, ranks, nodes, rel, gs)
    dn = torch.empty_like(nodes)
    with _on(dev), _timed("distmult_bwd_nodes"):
        _check(lib().rgcn_distmult_bwd_nodes_f32(_dp(rp_s), _dp(rp_o), _dp(entries), _dp(nodes), _dp(rel), _dp(dn), c_i64(N), c_i32(d),
                                                 _stream(dev)), "distmult_bwd_nodes")
    return dn
