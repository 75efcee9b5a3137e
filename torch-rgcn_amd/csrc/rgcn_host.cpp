// Host half of librgcn_hip.so: graph preparation for the R-GCN hot path.
// Pure C++ (no HIP calls) so it loads and runs on a box without a GPU.
// See include/rgcn_hip.h for the contract of every entry point.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "rgcn_hip.h"
#include "rgcn_options.h"

namespace {
thread_local char g_err[512] = "";
}

extern "C" void rgcn_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *rgcn_last_error(void) { return g_err; }
extern "C" const char *rgcn_version(void) { return "rgcn-hip 0.1 (gfx950)"; }
#ifndef RGCN_CSRC_SHA
#define RGCN_CSRC_SHA "unknown"
#endif
extern "C" const char *rgcn_csrc_sha(void) { return RGCN_CSRC_SHA; }

// ---- tuning options: name, value (= default until rgcn_set_option).  No getenv anywhere in the library: torch_rgcn/routes.py (or any
// other host) decides and says so through this table.
namespace {
struct OptEntry { const char *name; int32_t value; };
OptEntry g_options[] = {
    {"bwd_nw", 16},               // rgcn_bwd_lean_f32: waves per workgroup (16 / 8)
    {"gemm_bm", 0},               // rgcn_gemm_f32: 0 auto, 64 / 128 rows per tile
    {"spmm_u", 4},                // rgcn_spmm_f32 hidden-16 kernel: chunks per loop iteration
    {"wgrad_rg", 1},              // rgcn_wgrad_tiled_f32 variants
    {"wgrad_u", 2},
    {"bwd_abl", 0},               // ablation build only: timing experiments with WRONG results
};
static_assert(sizeof(g_options) / sizeof(g_options[0]) == RGCN_OPT_COUNT, "option table out of step with enum RgcnOpt (rgcn_device.h)");
}  // namespace

extern "C" int32_t rgcn_option_value(int index) { return (index >= 0 && index < RGCN_OPT_COUNT) ? g_options[index].value : 0; }

extern "C" int rgcn_set_option(const char *name, int32_t value) {
  for (auto &o : g_options)
    if (name && !strcmp(o.name, name)) {
#ifndef RGCN_ABLATIONS
      if (!strcmp(name, "bwd_abl") && value) {
        rgcn_set_error("set_option: %s exists in the ablation build only (make -C torch-rgcn_amd/csrc abl)", name);
        return RGCN_EUNSUPPORTED;
      }
#endif
      o.value = value;
      return RGCN_OK;
    }
  rgcn_set_error("set_option: unknown option %s", name ? name : "(null)");
  return RGCN_EINVAL;
}

extern "C" int rgcn_get_option(const char *name, int32_t *value) {
  for (auto &o : g_options)
    if (name && value && !strcmp(o.name, name)) { *value = o.value; return RGCN_OK; }
  rgcn_set_error("get_option: unknown option %s", name ? name : "(null)");
  return RGCN_EINVAL;
}

namespace {

// LSD radix sort of (key, payload) pairs on the low `bits` bits of key, 11 bits a pass.
void radix_sort_pairs(std::vector<uint64_t> &key, std::vector<int64_t> &val, int bits) {
  const size_t n = key.size();
  std::vector<uint64_t> k2(n);
  std::vector<int64_t> v2(n);
  const int RB = 11;
  std::vector<size_t> hist(size_t(1) << RB);
  for (int shift = 0; shift < bits; shift += RB) {
    std::fill(hist.begin(), hist.end(), 0);
    const uint64_t mask = (uint64_t(1) << RB) - 1;
    for (size_t i = 0; i < n; ++i) ++hist[(key[i] >> shift) & mask];
    size_t run = 0;
    for (auto &h : hist) { size_t c = h; h = run; run += c; }
    for (size_t i = 0; i < n; ++i) {
      size_t p = hist[(key[i] >> shift) & mask]++;
      k2[p] = key[i];
      v2[p] = val[i];
    }
    key.swap(k2);
    val.swap(v2);
  }
}

int bits_for(uint64_t maxv) {
  int b = 1;
  while (b < 64 && (maxv >> b)) ++b;
  return b;
}

// count[e] = number of entries sharing key[e]
void count_by_key(const std::vector<uint64_t> &keys, uint64_t maxkey, std::vector<int64_t> &count) {
  const size_t n = keys.size();
  count.assign(n, 0);
  if (!n) return;
  std::vector<uint64_t> k(keys);
  std::vector<int64_t> idx(n);
  for (size_t i = 0; i < n; ++i) idx[i] = int64_t(i);
  radix_sort_pairs(k, idx, bits_for(maxkey));
  size_t a = 0;
  while (a < n) {
    size_t b = a;
    while (b < n && k[b] == k[a]) ++b;
    for (size_t t = a; t < b; ++t) count[size_t(idx[t])] = int64_t(b - a);
    a = b;
  }
}

}  // namespace

extern "C" int rgcn_add_inverse_and_self_host(const int64_t *T, int64_t E, int64_t N, int64_t R0, int64_t *out) {
  if (E < 0 || N < 0 || (!T && E) || !out) { rgcn_set_error("add_inverse_and_self: bad argument"); return RGCN_EINVAL; }
  int64_t *fwd = out, *inv = out + 3 * E, *slf = out + 6 * E;
  for (int64_t e = 0; e < E; ++e) {
    const int64_t s = T[3 * e], p = T[3 * e + 1], o = T[3 * e + 2];
    fwd[3 * e] = s; fwd[3 * e + 1] = p; fwd[3 * e + 2] = o;
    inv[3 * e] = o; inv[3 * e + 1] = p + R0; inv[3 * e + 2] = s;
  }
  for (int64_t n = 0; n < N; ++n) { slf[3 * n] = n; slf[3 * n + 1] = 2 * R0; slf[3 * n + 2] = n; }
  return RGCN_OK;
}

extern "C" int rgcn_lp_augment_host(const int64_t *T, int64_t E, int64_t N, int64_t R0, const uint8_t *keep,
                                    int64_t *out, int64_t *M_out, int64_t *n_self) {
  if (E < 0 || N < 0 || (!T && E) || !out || !M_out || !n_self) { rgcn_set_error("lp_augment: bad argument"); return RGCN_EINVAL; }
  int64_t *b0 = out, *b1 = out + 3 * E, *b2 = out + 6 * E, *sl = out + 9 * E;
  for (int64_t e = 0; e < E; ++e) {
    const int64_t s = T[3 * e], p = T[3 * e + 1], o = T[3 * e + 2];
    b0[3 * e] = s; b0[3 * e + 1] = p; b0[3 * e + 2] = o;
    b1[3 * e] = o; b1[3 * e + 1] = p + R0; b1[3 * e + 2] = s;
    b2[3 * e] = s; b2[3 * e + 1] = p; b2[3 * e + 2] = o;
  }
  int64_t kept = 0;
  for (int64_t n = 0; n < N; ++n) {
    if (keep && !keep[n]) continue;
    sl[3 * kept] = n; sl[3 * kept + 1] = 2 * R0; sl[3 * kept + 2] = n;
    ++kept;
  }
  *M_out = 3 * E + kept;
  *n_self = E + kept;
  return RGCN_OK;
}

extern "C" int rgcn_edge_norm_host(const int64_t *Tp, int64_t M, int64_t N, int64_t R, int vertical,
                                   int64_t n_swap, int64_t i_tail, float *val) {
  if (M < 0 || N <= 0 || R <= 0 || (!Tp && M) || (!val && M)) { rgcn_set_error("edge_norm: bad argument"); return RGCN_EINVAL; }
  try {
    std::vector<uint64_t> keys(static_cast<size_t>(M));
    for (int64_t e = 0; e < M; ++e) {
      const int64_t s = Tp[3 * e], p = Tp[3 * e + 1], o = Tp[3 * e + 2];
      // stack_matrices' asserts: both stacked indices below the matrix size (utils.py:163-164);
      // negative ids would make the sparse constructor throw, so they are range errors too.
      if (s < 0 || s >= N || o < 0 || o >= N || p < 0 || p >= R) {
        rgcn_set_error("edge_norm: triple %lld = (%lld,%lld,%lld) out of range for N=%lld R=%lld", (long long)e,
                       (long long)s, (long long)p, (long long)o, (long long)N, (long long)R);
        return RGCN_ERANGE;
      }
      keys[size_t(e)] = uint64_t(p) * uint64_t(N) + uint64_t(vertical ? s : o);
    }
    std::vector<int64_t> cnt;
    count_by_key(keys, uint64_t(R) * uint64_t(N), cnt);
    if (vertical) {
      for (int64_t e = 0; e < M; ++e) val[e] = 1.0f / float(cnt[size_t(e)]);
    } else {
      if (n_swap < 0 || i_tail < 0 || 2 * n_swap + i_tail != M) {
        rgcn_set_error("edge_norm: horizontal swap needs 2n+i == M (n=%lld i=%lld M=%lld)", (long long)n_swap,
                       (long long)i_tail, (long long)M);
        return RGCN_EINVAL;
      }
      const int64_t n = n_swap;
      for (int64_t e = 0; e < n; ++e) val[e] = 1.0f / float(cnt[size_t(n + e)]);
      for (int64_t e = 0; e < n; ++e) val[n + e] = 1.0f / float(cnt[size_t(e)]);
      for (int64_t e = 0; e < i_tail; ++e) val[2 * n + e] = 1.0f / float(cnt[size_t(M - i_tail + e)]);
    }
  } catch (const std::bad_alloc &) {
    rgcn_set_error("edge_norm: out of host memory");
    return RGCN_ENOMEM;
  }
  return RGCN_OK;
}

namespace {

struct PlanShape {
  int64_t n_tiles = 0, n_chunks = 0, m_pad = 0, n_items = 0;
};

// bucket = tile * R + rel.  Fills per-bucket counts; validates ranges.
int bucket_counts(const int32_t *dst, const int32_t *rel, int64_t M, int64_t n_dst, int32_t R, int32_t tile_rows,
                  std::vector<int64_t> &cnt, int64_t &n_tiles) {
  if (M < 0 || n_dst < 0 || R <= 0 || tile_rows <= 0 || (M && (!dst || !rel))) {
    rgcn_set_error("plan: bad argument");
    return RGCN_EINVAL;
  }
  n_tiles = (n_dst + tile_rows - 1) / tile_rows;
  if (n_tiles * int64_t(R) > (int64_t(1) << 31)) { rgcn_set_error("plan: too many (tile, relation) buckets"); return RGCN_EUNSUPPORTED; }
  cnt.assign(size_t(n_tiles * R), 0);
  for (int64_t e = 0; e < M; ++e) {
    const int32_t d = dst[e], r = rel[e];
    if (d < 0 || d >= n_dst || r < 0 || r >= R) {
      rgcn_set_error("plan: message %lld (dst=%d rel=%d) out of range", (long long)e, d, r);
      return RGCN_ERANGE;
    }
    ++cnt[size_t(int64_t(d / tile_rows) * R + r)];
  }
  return RGCN_OK;
}

PlanShape shape_of(const std::vector<int64_t> &cnt, int64_t n_tiles, int32_t max_item_chunks) {
  PlanShape s;
  s.n_tiles = n_tiles;
  const int64_t mic = max_item_chunks > 0 ? max_item_chunks : (int64_t(1) << 40);
  for (int64_t c : cnt) {
    if (!c) continue;
    const int64_t ch = (c + RGCN_CHUNK - 1) / RGCN_CHUNK;
    s.n_chunks += ch;
    s.n_items += (ch + mic - 1) / mic;
  }
  s.m_pad = s.n_chunks * RGCN_CHUNK;
  return s;
}

}  // namespace

extern "C" int rgcn_plan_count_host(const int32_t *dst, const int32_t *rel, int64_t M, int64_t n_dst, int32_t R,
                                    int32_t tile_rows, int32_t max_item_chunks, int64_t *m_pad, int64_t *n_chunks,
                                    int64_t *n_tiles, int64_t *n_items) {
  try {
    std::vector<int64_t> cnt;
    int64_t nt = 0;
    int rc = bucket_counts(dst, rel, M, n_dst, R, tile_rows, cnt, nt);
    if (rc) return rc;
    PlanShape s = shape_of(cnt, nt, max_item_chunks);
    if (s.m_pad >= (int64_t(1) << 31)) { rgcn_set_error("plan: more than 2^31 slots"); return RGCN_EUNSUPPORTED; }
    if (m_pad) *m_pad = s.m_pad;
    if (n_chunks) *n_chunks = s.n_chunks;
    if (n_tiles) *n_tiles = s.n_tiles;
    if (n_items) *n_items = s.n_items;
  } catch (const std::bad_alloc &) {
    rgcn_set_error("plan: out of host memory");
    return RGCN_ENOMEM;
  }
  return RGCN_OK;
}

extern "C" int rgcn_plan_fill_host(const int32_t *dst, const int32_t *src, const int32_t *rel, const float *val,
                                   int64_t M, int64_t n_dst, int64_t n_src, int32_t R, int32_t tile_rows,
                                   int32_t max_item_chunks, int32_t *p_src, int32_t *p_dst, float *p_val,
                                   int32_t *p_perm, int32_t *chunk_rel, int32_t *tile_ptr, int32_t *items,
                                   int32_t *run_ptr, int32_t *p_pack) {
  try {
    std::vector<int64_t> cnt;
    int64_t nt = 0;
    int rc = bucket_counts(dst, rel, M, n_dst, R, tile_rows, cnt, nt);
    if (rc) return rc;
    if (M && (!src || !val || !p_src || !p_dst || !p_val || !chunk_rel)) { rgcn_set_error("plan_fill: null array"); return RGCN_EINVAL; }
    if (!tile_ptr) { rgcn_set_error("plan_fill: null tile_ptr"); return RGCN_EINVAL; }
    for (int64_t e = 0; e < M; ++e)
      if (src[e] < 0 || src[e] >= n_src) {
        rgcn_set_error("plan: message %lld src=%d out of range (n_src=%lld)", (long long)e, src[e], (long long)n_src);
        return RGCN_ERANGE;
      }
    // order messages by (bucket, dst): stable counting sort by dst first, then by bucket
    std::vector<int64_t> byd(size_t(M) ? size_t(M) : 1);
    {
      std::vector<int64_t> head(size_t(n_dst) + 1, 0);
      for (int64_t e = 0; e < M; ++e) ++head[size_t(dst[e]) + 1];
      for (int64_t d = 0; d < n_dst; ++d) head[size_t(d) + 1] += head[size_t(d)];
      for (int64_t e = 0; e < M; ++e) byd[size_t(head[size_t(dst[e])]++)] = e;
    }
    const size_t nb = cnt.size();
    std::vector<int64_t> slot0(nb + 1, 0);  // first slot of every bucket (padded layout)
    for (size_t b = 0; b < nb; ++b) slot0[b + 1] = slot0[b] + (cnt[b] + RGCN_CHUNK - 1) / RGCN_CHUNK * RGCN_CHUNK;
    std::vector<int64_t> cur(slot0.begin(), slot0.end() - 1);
    for (int64_t t = 0; t < M; ++t) {
      const int64_t e = byd[size_t(t)];
      const size_t b = size_t(int64_t(dst[e] / tile_rows) * R + rel[e]);
      const int64_t p = cur[b]++;
      p_src[p] = src[e];
      p_dst[p] = dst[e];
      p_val[p] = val[e];
      if (p_perm) p_perm[p] = int32_t(e);
    }
    const int64_t mic = max_item_chunks > 0 ? max_item_chunks : (int64_t(1) << 40);
    int64_t n_items = 0;
    for (size_t b = 0; b < nb; ++b) {
      if (b % size_t(R) == 0) tile_ptr[b / size_t(R)] = int32_t(slot0[b] / RGCN_CHUNK);
      if (run_ptr) {
        const size_t t = b / size_t(R), r = b % size_t(R);
        run_ptr[t * size_t(R + 1) + r] = int32_t(slot0[b] / RGCN_CHUNK);
        if (r + 1 == size_t(R)) run_ptr[t * size_t(R + 1) + size_t(R)] = int32_t(slot0[b + 1] / RGCN_CHUNK);
      }
      if (!cnt[b]) continue;
      const int64_t last = cur[b] - 1;
      for (int64_t p = cur[b]; p < slot0[b + 1]; ++p) {  // pads: val 0, dst -1, source row of the last real message
        p_src[p] = p_src[last];
        p_dst[p] = -1;
        p_val[p] = 0.0f;
        if (p_perm) p_perm[p] = -1;
      }
      const int64_t c0 = slot0[b] / RGCN_CHUNK, c1 = slot0[b + 1] / RGCN_CHUNK;
      for (int64_t c = c0; c < c1; ++c) chunk_rel[c] = int32_t(b % size_t(R));
      if (items)
        for (int64_t c = c0; c < c1; c += mic) {
          items[2 * n_items] = int32_t(c);
          items[2 * n_items + 1] = int32_t(std::min(c + mic, c1));
          ++n_items;
        }
    }
    tile_ptr[nt] = int32_t(slot0[nb] / RGCN_CHUNK);
    if (p_pack) {
      if (n_src >= (int64_t(1) << 24) || tile_rows > 255) {
        rgcn_set_error("plan_fill: packed slots need n_src < 2^24 and tile_rows <= 255");
        return RGCN_EUNSUPPORTED;
      }
      const int64_t m_pad = slot0[nb];
      for (int64_t p = 0; p < m_pad; ++p) {
        const uint32_t dl = p_dst[p] < 0 ? 0xFFu : uint32_t(p_dst[p] % tile_rows);   // 0xFF marks a pad
        p_pack[2 * p] = int32_t(uint32_t(p_src[p]) | (dl << 24));
        std::memcpy(&p_pack[2 * p + 1], &p_val[p], sizeof(float));
      }
    }
  } catch (const std::bad_alloc &) {
    rgcn_set_error("plan: out of host memory");
    return RGCN_ENOMEM;
  }
  return RGCN_OK;
}

extern "C" int rgcn_plan_units_host(const int32_t *tile_ptr, int64_t n_tiles, int32_t max_unit_chunks, int32_t *units,
                                    int64_t *n_units, int64_t *n_split) {
  if (n_tiles < 0 || (n_tiles && !tile_ptr) || max_unit_chunks <= 0 || !n_units) { rgcn_set_error("plan_units: bad argument"); return RGCN_EINVAL; }
  int64_t n = 0, ns = 0;
  for (int64_t t = 0; t < n_tiles; ++t) {
    const int64_t c0 = tile_ptr[t], c1 = tile_ptr[t + 1];
    if (c1 - c0 <= max_unit_chunks) {
      if (units) { units[4 * n] = int32_t(t); units[4 * n + 1] = int32_t(c0); units[4 * n + 2] = int32_t(c1); units[4 * n + 3] = 0; }
      ++n;
    } else {
      for (int64_t c = c0; c < c1; c += max_unit_chunks, ++n, ++ns)
        if (units) {
          units[4 * n] = int32_t(t);
          units[4 * n + 1] = int32_t(c);
          units[4 * n + 2] = int32_t(std::min<int64_t>(c + max_unit_chunks, c1));
          units[4 * n + 3] = RGCN_U_SHARED | (c == c0 ? RGCN_U_FIRST : 0);
        }
    }
  }
  *n_units = n;
  if (n_split) *n_split = ns;
  return RGCN_OK;
}

extern "C" int rgcn_synthetic_triples_host(int64_t N, int64_t R0, int64_t E, uint64_t seed, int64_t *out) {
  if (N <= 0 || R0 <= 0 || E < 0 || (!out && E)) { rgcn_set_error("synthetic_triples: bad argument"); return RGCN_EINVAL; }
  uint64_t x = seed;
  auto next = [&x]() {
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  };
  for (int64_t e = 0; e < E; ++e) {
    out[3 * e] = int64_t(next() % uint64_t(N));
    out[3 * e + 1] = int64_t(next() % uint64_t(R0));
    out[3 * e + 2] = int64_t(next() % uint64_t(N));
  }
  return RGCN_OK;
}

// ------------------------------------------------------------------ edge-neighbourhood sampler
// utils/misc.py:125-172 draws sample_size edges; every draw picks a vertex with probability proportional to
// (#unpicked incident edge ends) x (vertex already touched), falling back to "any vertex that still has an unpicked
// edge" when no touched vertex has one, then one of that vertex's unpicked edge ends uniformly.  The reference
// rebuilds the N-long probability vector per draw; here both weight vectors live in Fenwick trees.
namespace {
struct Fenwick {
  std::vector<int64_t> t;
  int64_t n, top, total = 0;
  explicit Fenwick(int64_t n_) : t(size_t(n_) + 1, 0), n(n_), top(1) { while (top * 2 <= n) top *= 2; }
  void add(int64_t i, int64_t d) { total += d; for (++i; i <= n; i += i & -i) t[size_t(i)] += d; }
  int64_t find(int64_t u) const {   // smallest i with prefix_sum(i) > u, 0 <= u < total
    int64_t pos = 0;
    for (int64_t step = top; step; step >>= 1)
      if (pos + step <= n && t[size_t(pos + step)] <= u) { pos += step; u -= t[size_t(pos)]; }
    return pos;
  }
};
}  // namespace

extern "C" int rgcn_edge_neighborhood_host(const int64_t *triples, int64_t E, int64_t N, int64_t sample_size,
                                           uint64_t seed, int64_t *picked_edges) {
  if (E < 0 || N <= 0 || sample_size < 0 || (E && !triples) || (sample_size && !picked_edges)) {
    rgcn_set_error("edge_neighborhood: bad argument");
    return RGCN_EINVAL;
  }
  if (sample_size > E) { rgcn_set_error("edge_neighborhood: sample_size %lld exceeds the %lld edges", (long long)sample_size, (long long)E); return RGCN_EINVAL; }
  for (int64_t e = 0; e < E; ++e)
    if (triples[3 * e] < 0 || triples[3 * e] >= N || triples[3 * e + 2] < 0 || triples[3 * e + 2] >= N) {
      rgcn_set_error("edge_neighborhood: node index out of range at triple %lld", (long long)e);
      return RGCN_ERANGE;
    }
  uint64_t x = seed;
  auto below = [&x](uint64_t n) {   // uniform in [0, n): splitmix64 + multiply-shift
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return uint64_t(((unsigned __int128)z * n) >> 64);
  };
  // edge ends per vertex (CSR): a self loop contributes two ends to its vertex, as in the reference's adj_list
  std::vector<int64_t> ptr((size_t)N + 1, 0);
  for (int64_t e = 0; e < E; ++e) { ++ptr[size_t(triples[3 * e]) + 1]; ++ptr[size_t(triples[3 * e + 2]) + 1]; }
  for (int64_t v = 0; v < N; ++v) ptr[size_t(v) + 1] += ptr[size_t(v)];
  std::vector<int64_t> end_edge((size_t)(2 * E), 0), end_other((size_t)(2 * E), 0), fill(ptr.begin(), ptr.end() - 1);
  for (int64_t e = 0; e < E; ++e) {
    const int64_t s = triples[3 * e], o = triples[3 * e + 2];
    end_edge[size_t(fill[size_t(s)])] = e; end_other[size_t(fill[size_t(s)]++)] = o;
    end_edge[size_t(fill[size_t(o)])] = e; end_other[size_t(fill[size_t(o)]++)] = s;
  }
  std::vector<int64_t> left((size_t)N, 0);             // unpicked edge ends per vertex (the reference's sample_counts)
  std::vector<char> seen((size_t)N, 0), picked((size_t)E, 0);
  Fenwick touched(N), any(N);
  for (int64_t v = 0; v < N; ++v) { left[size_t(v)] = ptr[size_t(v) + 1] - ptr[size_t(v)]; if (left[size_t(v)]) any.add(v, 1); }
  auto touch = [&](int64_t v) { if (!seen[size_t(v)]) { seen[size_t(v)] = 1; touched.add(v, left[size_t(v)]); } };
  auto drop_end = [&](int64_t v) {
    if (seen[size_t(v)]) touched.add(v, -1);
    if (--left[size_t(v)] == 0) any.add(v, -1);
  };
  for (int64_t i = 0; i < sample_size; ++i) {
    const int64_t v = touched.total > 0 ? touched.find(int64_t(below(uint64_t(touched.total))))
                                        : any.find(int64_t(below(uint64_t(any.total))));
    touch(v);
    const int64_t b = ptr[size_t(v)], deg = ptr[size_t(v) + 1] - b;
    int64_t j = b + int64_t(below(uint64_t(deg)));
    while (picked[size_t(end_edge[size_t(j)])]) j = b + int64_t(below(uint64_t(deg)));   // misc.py:156-159
    const int64_t e = end_edge[size_t(j)], other = end_other[size_t(j)];
    picked[size_t(e)] = 1;
    picked_edges[i] = e;
    drop_end(v);
    drop_end(other);
    touch(other);
  }
  return RGCN_OK;
}
