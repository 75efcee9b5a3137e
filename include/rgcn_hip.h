/*
 * rgcn_hip.h -- C ABI of librgcn_hip.so: the MI355X (gfx950) implementation of the
 * R-GCN relational message-passing hot path
 *
 *     out[s,:] = sum_{(s,p,o) in T+} val_e * X[o,:] @ W_p  (+ b)        and its backward.
 *
 * The reference (thiviyanT/torch-rgcn) has no native layer: the path is Python over
 * ATen.  Each entry point therefore names the reference *Python*
 * lines whose work it takes over; the Python modules under
 * torch-rgcn_amd/torch_rgcn/ (same class names and signatures as
 * torch_rgcn/layers.py) are the only callers.  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / HIP types in signatures
 *     (`stream` is a hipStream_t passed as void*; NULL = default stream)
 *   - *_host functions run on the CPU and need no GPU; every other function takes
 *     DEVICE pointers and only enqueues work on `stream` (no synchronisation)
 *   - all floating point is fp32, indices are int32 on the device side and int64
 *     where the reference hands over LongTensors
 *   - return value: RGCN_OK or an error code; rgcn_last_error() gives the text.
 *     The Python layer turns RGCN_ERANGE / RGCN_EINVAL into AssertionError the way
 *     the reference's asserts do (layers.py:121,225,282-284,303; utils.py:148,162-164).
 */
#ifndef RGCN_HIP_H
#define RGCN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGCN_API __attribute__((visibility("default")))

#define RGCN_OK 0
#define RGCN_EINVAL 1       /* bad argument / shape mismatch */
#define RGCN_ENOMEM 2
#define RGCN_ERANGE 3       /* node / relation index out of range */
#define RGCN_EHIP 4         /* a HIP runtime call failed */
#define RGCN_EUNSUPPORTED 5

#define RGCN_CHUNK 16       /* slots per chunk: one MFMA 16x16x4 row block */

RGCN_API const char *rgcn_version(void);
RGCN_API const char *rgcn_last_error(void);
/* first 16 hex digits of the SHA-256 over the library sources (every .hip / .cpp / .h under csrc plus this header) at build time: profiles/ carry
 * it, bench.py reports whether the committed counter files were taken on THIS binary (no reference counterpart) */
RGCN_API const char *rgcn_csrc_sha(void);
/* Test helper (no counterpart in the reference; changes no result): fills the LDS of every CU with quiet-NaN bit patterns -- one
 * workgroup per CU writes its whole 160 KiB allocation and exits.  LDS keeps its contents between kernels, so a kernel that reads a word
 * it never wrote (and relies on multiplying it by zero) computes NaN afterwards instead of getting away with yesterday's finite
 * garbage: tests/test_gpu_parity.py runs the tile kernels behind it (round 5: that is how the fused tile backward lost its dcomps in
 * 1 of 25 runs). */
RGCN_API int rgcn_poison_lds(void *stream);
/* Tuning options: the library never reads the environment -- whoever hosts it (torch_rgcn/routes.py) decides and says so here.
 * Names: bwd_nw, gemm_bm, spmm_u, wgrad_rg, wgrad_u (all of them choose between kernels that compute the same result), and bwd_abl:
 * timing experiments with WRONG results that exist in the ablation build only (make -C torch-rgcn_amd/csrc abl) -- the shipped
 * library refuses a non-zero value (RGCN_EUNSUPPORTED).  Process-wide; set them before launching from several threads.  No
 * reference counterpart (the reference has no native layer). */
RGCN_API int rgcn_set_option(const char *name, int32_t value);
RGCN_API int rgcn_get_option(const char *name, int32_t *value);

/* ------------------------------------------------------------------ graph preparation (host) */

/* [T | inverse(T) | self loops]; out has (2E+N) rows of 3.
 * Replaces torch_rgcn/utils.py:127-141 (add_inverse_and_self). */
RGCN_API int rgcn_add_inverse_and_self_host(const int64_t *triples, int64_t E, int64_t N, int64_t R0,
                                            int64_t *out);

/* Link-prediction augmentation [T | inverse(T) | T | kept self loops] (the original
 * block enters twice -- SURVEY.md F5).  keep may be NULL (= keep all).  out must
 * hold 3E+N rows.  Replaces utils.py:100-124 + layers.py:481-487. */
RGCN_API int rgcn_lp_augment_host(const int64_t *triples, int64_t E, int64_t N, int64_t R0,
                                  const uint8_t *keep, int64_t *out, int64_t *M_out, int64_t *n_self);

/* Per-edge adjacency value by the layer's literal procedure: count edges sharing
 * (p,s) [vertical] or (p,o) [horizontal], apply the block swap
 * c = [k[n:2n] | k[0:n] | k[M-i:M]] when horizontal, val = 1/c.
 * Replaces utils.py:143-166 (stack_matrices), utils.py:71-97 (sum_sparse) and
 * layers.py:263-273 / :498-510.  RGCN_ERANGE where stack_matrices' asserts fire,
 * RGCN_EINVAL where the reference hits a shape error (2n+i != M). */
RGCN_API int rgcn_edge_norm_host(const int64_t *triples_plus, int64_t M, int64_t N, int64_t R, int vertical,
                                 int64_t n_swap, int64_t i_tail, float *val);

/* Relation-tile plan.  Messages (dst <- src, relation rel, weight val) are bucketed
 * by (dst / tile_rows, rel), sorted by dst inside a bucket, and every bucket is
 * padded to a multiple of RGCN_CHUNK slots (pad: val = 0, dst = -1, src = the bucket's
 * last source row), so that each chunk of 16 slots has ONE relation and one dst tile.
 * This replaces the sparse COO constructor + coalesce of layers.py:276-279 /
 * :513-516 as the device-side graph layout.
 *   rgcn_plan_count_host : sizes only
 *   rgcn_plan_fill_host  : fills caller-allocated arrays
 *       p_src, p_dst [m_pad] int32; p_val [m_pad] f32; p_perm [m_pad] (original
 *       message index, -1 for pads; may be NULL); chunk_rel [n_chunks];
 *       tile_ptr [n_tiles+1] (chunk offsets);  items [2*n_items] = (c0,c1) chunk
 *       ranges of constant relation, at most max_item_chunks long (weight-gradient
 *       work list; may be NULL);  run_ptr [n_tiles*(R+1)] = first chunk of every
 *       (tile, relation) run, row t ends with tile_ptr[t+1] (may be NULL);
 *       p_pack [2*m_pad] = 8-byte slots { src | (dst - tile_row0) << 24 , val bits } for the
 *       hidden-16 kernels (dst field 0xFF = pad; needs n_src < 2^24 and tile_rows <= 255; may be NULL). */
RGCN_API int rgcn_plan_count_host(const int32_t *dst, const int32_t *rel, int64_t M, int64_t n_dst, int32_t R,
                                  int32_t tile_rows, int32_t max_item_chunks, int64_t *m_pad,
                                  int64_t *n_chunks, int64_t *n_tiles, int64_t *n_items);
RGCN_API int rgcn_plan_fill_host(const int32_t *dst, const int32_t *src, const int32_t *rel, const float *val,
                                 int64_t M, int64_t n_dst, int64_t n_src, int32_t R, int32_t tile_rows,
                                 int32_t max_item_chunks, int32_t *p_src, int32_t *p_dst, float *p_val,
                                 int32_t *p_perm, int32_t *chunk_rel, int32_t *tile_ptr, int32_t *items,
                                 int32_t *run_ptr, int32_t *p_pack);

/* Work units of the tile kernels: normally one per destination tile (flags 0: the wave owns the tile and
 * writes its rows); a tile with more than max_unit_chunks chunks (a hub node) is cut into several units
 * (flags RGCN_U_SHARED: partial sums are added to `out` with fp32 atomics; RGCN_U_FIRST marks the piece
 * that also adds the bias).  units = int32 [n_units][4] = {tile, chunk_begin, chunk_end, flags}; pass
 * units = NULL to get the count only.  *n_split receives the number of shared units. */
#define RGCN_U_SHARED 1
#define RGCN_U_FIRST 2
RGCN_API int rgcn_plan_units_host(const int32_t *tile_ptr, int64_t n_tiles, int32_t max_unit_chunks,
                                  int32_t *units, int64_t *n_units, int64_t *n_split);

/* splitmix64 synthetic graph: s,o ~ U[0,N), p ~ U[0,R0), three
 * consecutive stream values per triple.  Same stream as oracle.synthetic_triples. */
RGCN_API int rgcn_synthetic_triples_host(int64_t N, int64_t R0, int64_t E, uint64_t seed, int64_t *out);

/* Edge-neighbourhood sampler of the link-prediction experiments (utils/misc.py:125-172):
 * picked_edges[i] = index of the i-th sampled triple, sample_size <= E distinct edges.  Same sampling distribution
 * as the reference, O(log N) per draw (Fenwick trees) instead of O(N); its own splitmix64 stream from `seed`. */
RGCN_API int rgcn_edge_neighborhood_host(const int64_t *triples, int64_t E, int64_t N, int64_t sample_size,
                                         uint64_t seed, int64_t *picked_edges);

/* ------------------------------------------------------------------ graph preparation (device)
 * The same work as the *_host functions above without leaving the GPU (the LP layer builds its graph on
 * every call, layers.py:481-516).  Dense counting: workspaces are one int per (tile, relation, row) cell.
 * All pointers are device pointers; nothing synchronises.  err_flag (1 int) becomes 1 when an index is out
 * of range (the caller turns that into the reference's AssertionError). */
RGCN_API int rgcn_dev_split_triples(const int64_t *triples_plus, int64_t M, int64_t N, int32_t R, int32_t *s,
                                    int32_t *p, int32_t *o, int32_t *err_flag, void *stream);
/* [T | inverse(T) | T | self loops] as s/p/o int32 arrays of length 3E+N; dropped self loops (keep[n] == 0)
 * stay in the list with alive = 0 and are ignored by everything downstream. */
RGCN_API int rgcn_dev_lp_expand(const int64_t *triples, int64_t E, int64_t N, int32_t R0, const uint8_t *keep,
                                int32_t *s, int32_t *p, int32_t *o, uint8_t *alive, int32_t *err_flag, void *stream);
/* val by the literal procedure (see rgcn_edge_norm_host); table = R*N ints of workspace; alive may be NULL.
 * The caller checks 2*n_swap + i_tail == number of live messages (the reference's shape error). */
RGCN_API int rgcn_dev_edge_norm(const int32_t *s, const int32_t *p, const int32_t *o, const uint8_t *alive, int64_t M,
                                int64_t N, int32_t R, int vertical, int64_t n_swap, int32_t *table, float *val,
                                void *stream);
/* Relation-tile plan in two steps.  count: cells [n_tiles*R*tile_rows], bucket_cnt [n_tiles*R],
 * bucket_base [n_tiles*R + 1] (last entry = m_pad, which the host reads back to size the outputs),
 * scan_tmp [n_tiles*R/1024 + 2].  cells_tmp (may be NULL): a second cell table; with it and tile_rows > 1024
 * (relation-major plan, CSR) the cells are scanned globally instead of one wave per bucket, and scan_tmp must
 * hold n_cells/1024 + 3 ints.  n_chunks (fill) = m_pad / 16 as read back from bucket_base.  fill: same layout as rgcn_plan_fill_host (message order inside one
 * (relation, destination) cell is arbitrary). */
RGCN_API int rgcn_dev_plan_count(const int32_t *dst, const int32_t *rel, const uint8_t *alive, int64_t M, int64_t n_dst,
                                 int32_t R, int32_t tile_rows, int32_t *cells, int32_t *bucket_cnt,
                                 int32_t *bucket_base, int32_t *scan_tmp, int32_t *cells_tmp, void *stream);
RGCN_API int rgcn_dev_plan_fill(const int32_t *dst, const int32_t *src, const int32_t *rel, const float *val,
                                const uint8_t *alive, int64_t M, int64_t n_dst, int64_t n_src, int32_t R,
                                int32_t tile_rows, int32_t *cells, const int32_t *bucket_cnt,
                                const int32_t *bucket_base, int32_t *p_src, int32_t *p_dst, float *p_val,
                                int32_t *p_pack, int32_t *chunk_rel, int32_t *tile_ptr, int32_t *run_ptr,
                                const int32_t *aux, int32_t *p_aux, int32_t *msg_slot, int64_t n_chunks,
                                void *stream);
/* Two CSRs of one message list in five launches and without any read-back: rows = a[e] with entries (b[e], rel[e], val[e]) and
 * rows = b[e] with entries (a[e], rel[e], val[e]) -- what the basis / diagonal / block kernels (forward and backward walks) and
 * the DistMult backward (by subject, by object) read.  rowbuf: 2 N + 2 ints; afterwards rowbuf[0 .. N] is the row pointer of the
 * first CSR and rowbuf[N + 1 .. 2 N + 1] of the second, both indexing the SAME entry arrays e_other / e_rel / e_val (2 M entries).
 * scan_tmp: (2 N + 1) / 1024 + 3 ints.  rel may be NULL (0).  alive (may be NULL): uint8 per message, 0 = skipped. */
RGCN_API int rgcn_dev_csr_pair(const int32_t *a, const int32_t *b, const int32_t *rel, const float *val, const uint8_t *alive,
                               int64_t M, int64_t N, int32_t *rowbuf, int32_t *scan_tmp, int32_t *e_other, int32_t *e_rel,
                               float *e_val, void *stream);
/* Completion of a plan WITHOUT any device -> host read (per-call graphs of the link-prediction layer, layers.py:481-516,
 * inside a training step that must not synchronise / is captured in a hipGraph): the caller sizes p_src / p_dst / p_val /
 * p_pack / chunk_rel by the upper bound m_pad_ub >= M + 15 * min(n_buckets, M) (rounded up to 16), runs rgcn_dev_plan_count
 * and rgcn_dev_plan_fill with n_chunks = m_pad_ub / 16, then this: slots past the real end become pads, `units` gets one
 * work unit per tile (no hub splitting) and, for the relation-major plan (n_tiles == 1), `items` the chunk ranges of each
 * relation cut at max_item_chunks, padded with empty items up to n_items_ub >= m_pad_ub / 16 / max_item_chunks + R. */
RGCN_API int rgcn_dev_plan_finish_nosync(const int32_t *bucket_base, int64_t n_tiles, int32_t R, int64_t m_pad_ub,
                                         int32_t *p_src, int32_t *p_dst, float *p_val, int32_t *p_pack, int32_t *p_aux,
                                         const int32_t *tile_ptr, int32_t *units, int32_t *items, int64_t n_items_ub,
                                         int32_t max_item_chunks, void *stream);
/* (msg_slot, may be NULL: slot index of every input message.)
 * (aux / p_aux, may be NULL: one extra int32 per message carried into slot order -- with R = 1 and
 * tile_rows >= n_dst the plan degenerates to a destination-major CSR, aux = relation, and `cells` holds the
 * row pointers: that is the layout of the basis-aggregation kernels below.) */

/* Basis decomposition at large width (W_r = sum_b comps[r,b] bases[b], layers.py:241-242 / :468-469):
 * aggregate first, contract afterwards --  ag[s, b, :] = sum_{e -> s} comps[rel_e, b] * val_e * X[src_e, :],
 * then ONE dense GEMM  out = ag.view(N, B*d) @ bases.view(B*d, d_out)  (rocBLAS; MFMA).  rowptr/p_src/p_rel/
 * p_val: destination-major CSR (see rgcn_dev_plan_fill).  n_b_in = 1: X rows are [d] and the output rows
 * [B*d] (forward);  n_b_in = B: X rows are [B*d] and the B blocks are summed into [d] rows (feature gradient
 * on the source-major CSR). */
RGCN_API int rgcn_basis_aggregate_f32(const float *X, const float *comps, float *out, const int32_t *rowptr,
                                      const int32_t *p_src, const int32_t *p_rel, const float *p_val,
                                      int64_t n_rows, int32_t R, int32_t B, int32_t d, int32_t n_b_in,
                                      void *stream);
/* dcomps[r, b] = sum_{e in r} val_e <X[src_e], D[dst_e, b, :]>   (D = d ag, rows of B*d), over the work
 * items of the relation-major plan (rgcn_plan_fill_host / rgcn_dev_plan_fill with tile_rows >= n_dst).
 * dcomps is [n_copies][R][B] (zeroed here): the pieces an item is cut into add to different copies -- with few relations
 * thousands of atomics on R * B addresses serialise at L2 (WN18: 15,000 on 74 addresses, 0.1 ms) -- and the caller sums the
 * copies; n_copies = 1 gives the plain [R][B] result. */
RGCN_API int rgcn_basis_dcomps_f32(const float *X, const float *D, float *dcomps, const int32_t *p_src,
                                   const int32_t *p_dst, const float *p_val, const int32_t *chunk_rel,
                                   const int32_t *items, int64_t n_items, int32_t R, int32_t B, int32_t d,
                                   int32_t n_copies, void *stream);
/* The same sum on the destination-major CSR of the forward walk (rows = destinations; entries: source, relation, val), without the
 * relation-major plan: one wave per row, D's row in registers, dcomps summed in an LDS table of doubles per workgroup (round 5: per-call
 * LP graphs no longer build a relation-major plan for this one kernel).  B <= 8, d a multiple of 4, R x B doubles within 60 KB. */
RGCN_API int rgcn_basis_dcomps_csr_supported(int32_t R, int32_t B, int32_t d);
RGCN_API int rgcn_basis_dcomps_csr_f32(const float *X, const float *D, float *dcomps, const int32_t *rowptr, const int32_t *p_src,
                                       const int32_t *p_rel, const float *p_val, int64_t n_rows, int32_t R, int32_t B, int32_t d,
                                       void *workspace, void *stream);
/* workspace of rgcn_basis_dcomps_csr_f32 (a kernel that sums an R x B table per workgroup): bytes for one row of doubles (whole 128-byte
 * lines) per workgroup + a ticket; its first 128 bytes ZEROED once by the caller and left zeroed by every launch.  The workgroups' tables
 * meet in a fixed order (agent-scope atomic exchanges into the rows, the last workgroup adds them up): bit-reproducible, no zeroing launch. */
RGCN_API int64_t rgcn_basis_sum_workspace_bytes(int32_t R, int32_t B);

/* ------------------------------------------------------------------ device kernels */

/* out[n_dst, d_out] = bias + sum over plan slots  val * X[src,:] @ W[rel]   (W: [R, d_in, d_out]).
 * Forward of layers.py:293-301 / :524-551 (both stackings are the same function of
 * (T+, val, X, W)); with the transposed plan, G in place of X and W^T in place of W
 * it is the feature gradient dX.  bias may be NULL.
 * flags: RGCN_F_RELU fuses max(.,0) into the epilogue (caller-side F.relu of models.py:196).
 * p_pack (may be NULL): packed slots from rgcn_plan_fill_host; with d_in = d_out = 16 the kernel then
 * reads 8 bytes of index data per message instead of 12. */
#define RGCN_F_RELU 1
#define RGCN_F_WPACKED 2   /* W is in MFMA fragment order (rgcn_pack_w16_f32 / rgcn_pack_w_blocks_f32); widths multiples of 16, <= 64 */
RGCN_API int rgcn_spmm_f32(const float *X, const float *W, const float *bias, float *out, const int32_t *p_src,
                           const int32_t *p_dst, const float *p_val, const int32_t *p_pack,
                           const int32_t *chunk_rel, const int32_t *units, int64_t n_units, int64_t n_split,
                           int32_t tile_rows, int64_t n_dst, int64_t n_src, int32_t R, int32_t d_in,
                           int32_t d_out, int32_t flags, void *stream);

/* Wp[r][16k+o][c] = W[r][4k+c][o]: the per-lane float4 the hidden-16 kernel feeds to the matrix cores
 * (weight assembly step of layers.py:239-244, device side).  W, Wp: [R,16,16]. */
RGCN_API int rgcn_pack_w16_f32(const float *W, float *Wp, int32_t R, void *stream);
/* The same for widths that are multiples of 16 (d_in, d_out <= 64; rgcn_spmm_f32 with RGCN_F_WPACKED):
 * Wp[r][ib][jb][16k+o][c] = W[r][16 ib + 4k + c][16 jb + o].  W, Wp: [R, d_in, d_out]. */
RGCN_API int rgcn_pack_w_blocks_f32(const float *W, float *Wp, int32_t R, int32_t d_in, int32_t d_out, void *stream);

/* Sparse-bucket variant of rgcn_spmm_f32 for d_in = d_out = 16 (graphs with many relations per tile, e.g. AM:
 * (tile, relation) buckets of a few messages would leave the 16-slot chunks mostly empty).  Two passes:
 *   1. relation-major plan (dense chunks): Y[pos[slot], :] = val * X[src,:] @ W[rel]   -- rgcn_spmm_scatter_f32
 *      (p_pos = position of the message in destination-major order; Wp packed by rgcn_pack_w16_f32)
 *   2. out[row,:] = bias + sum_{j in rowptr[row] .. rowptr[row+1]} Y[j,:]              -- rgcn_segment_sum_f32
 * Y is a [n_messages, 16] scratch the caller provides. */
RGCN_API int rgcn_spmm_scatter_f32(const float *X, const float *Wp, float *Y, const int32_t *p_src,
                                   const float *p_val, const int32_t *p_pos, const int32_t *chunk_rel,
                                   const int32_t *items, int64_t n_items, int32_t d, void *stream);
RGCN_API int rgcn_segment_sum_f32(const float *Y, const int32_t *rowptr, const float *bias, float *out,
                                  int64_t n_rows, int32_t d, int32_t flags, void *stream);
/* Variant: pass 1 called with p_pos = NULL leaves Y in relation-major SLOT order (sequential writes); pass 2 then
 * gathers a destination's rows through perm[destination-major position] = slot. */
RGCN_API int rgcn_segment_gather_sum_f32(const float *Y, const int32_t *perm, const int32_t *rowptr, const float *bias,
                                         float *out, int64_t n_rows, int32_t d, int32_t flags, void *stream);
/* Pass 2 over work units (int32 [n_units][4] = {row, first entry, end entry, flags}, as rgcn_block_spmm_f32's): hub rows are cut into
 * pieces that merge with fp32 atomics (n_split = number of such pieces; out is zeroed first; RGCN_F_RELU only with n_split = 0). */
RGCN_API int rgcn_segment_gather_sum_units_f32(const float *Y, const int32_t *perm, const int32_t *units, int64_t n_units,
                                               int64_t n_split, const float *bias, float *out, int64_t n_rows, int32_t d,
                                               int32_t flags, void *stream);

/* dW[rel] += sum over slots val * X[src,:]^T G[dst,:]  for every work item; dW
 * ([R, d_in, d_out]) is zeroed first.  Autograd dual of the einsum / sparse mm pair
 * (dW_r = (A_r X)^T g). */
RGCN_API int rgcn_wgrad_f32(const float *X, const float *G, float *dW, const int32_t *p_src,
                            const int32_t *p_dst, const float *p_val, const int32_t *chunk_rel,
                            const int32_t *items, int64_t n_items, int64_t n_dst, int64_t n_src, int32_t R,
                            int32_t d_in, int32_t d_out, void *stream);

/* Same gradient, tile-major: walks the FORWARD plan (destination tiles) so that the G rows of a
 * tile are staged once in LDS and only X[src] is gathered from HBM (one random row read per
 * message instead of two).  A wave owns 16 consecutive relations x tiles_per_item tiles and keeps
 * their 16 gradient blocks in MFMA accumulators.  d_in = d_out = 16 only (RGCN_EUNSUPPORTED
 * otherwise -- callers then use rgcn_wgrad_f32). */
RGCN_API int rgcn_wgrad_tiled_f32(const float *X, const float *G, float *dW, const int32_t *p_src,
                                  const int32_t *p_dst, const float *p_val, const int32_t *chunk_rel,
                                  const int32_t *run_ptr, int64_t n_tiles, int32_t tile_rows, int64_t n_dst,
                                  int64_t n_src, int32_t R, int32_t d_in, int32_t d_out,
                                  int32_t tiles_per_item, void *stream);

/* Fused backward of the hidden-16 layer: dX AND dW from ONE walk of the transposed plan (destination tile = tile of
 * source rows of the forward: the rows X[o]), i.e. one random row gather (G[s]) per message instead of two.  Autograd
 * duals of layers.py:293-301: dX[o] += val G[s] W_r^T, dW_r += val X[o]^T G[s].
 *   G            upstream gradient [n_rows, 16] (gathered);  X  layer input [n_rows, 16] (tile-local reads)
 *   Wt_packed    W^T in fragment order (rgcn_pack_w16t_f32)
 *   dX [n_rows,16], dW [R,16,16] outputs (both fully written)
 *   p_pack / chunk_rel / run_ptr / tile_rows: the transposed relation-tile plan, built with run pointers and packed
 *   slots (tile_rows <= 255), NO hub-split tiles (one work unit per tile)
 *   scratch      rgcn_bwd_fused_scratch_floats(n_tiles, R) floats: per-workgroup dW partials, summed in a fixed order
 *                by two small kernels (bit-reproducible); with RGCN_F_DW_ATOMIC the partials are added to dW with
 *                fp32 atomics instead and scratch may be NULL. */
#define RGCN_F_DW_ATOMIC 4
#define RGCN_F_TRANSPOSE_W 8   /* rgcn_block_spmm_f32: multiply by the transposed blocks */
#define RGCN_F_DIAG4 16        /* rgcn_bwd_blk_f32: W_r is block-diagonal with 4 x 4 blocks; only the diagonal blocks of dW_r are computed */
#define RGCN_F_PARTIAL 32      /* rgcn_bwd_blk_f32: `units` lists SOME whole tiles (a slab): their dX rows are written, dW / dbias zeroed and summed */
#define RGCN_F_ACCUMULATE 64   /* rgcn_bwd_blk_f32: a further slab of the same backward: as RGCN_F_PARTIAL, but dW / dbias are NOT zeroed first */
RGCN_API int64_t rgcn_bwd_fused_scratch_floats(int64_t n_tiles, int32_t R);
/* (rgcn_bwd_fused_f32 itself -- round 2's staging kernel, four wave-owned tiles per workgroup -- was the fallback of rounds 3-4 for
 * wave-owned tiles of 65 .. 160 rows; nothing selected it by default and round 5 removed it.  The argument conventions above are those
 * of the two kernels that follow.) */
/* The same backward on a reformatted plan (round 3, "lean" kernel): per-chunk bookkeeping -- unpacking, duplicate / tail flags of the
 * fold, LDS row offsets -- is done once per plan instead of once per launch.
 *   rgcn_bwd_lean_slot_bytes(n_chunks)   size of the slot array (12 bytes per slot: source row << 6 | flags, val, tile row << 6)
 *   rgcn_bwd_lean_prepare_f32            p_pack / chunk_rel of the transposed plan (n_chunks = m_pad / 16) -> slots, hdr [n_chunks]
 *                                        (relation | fold flags); once per static graph, per call for per-call graphs
 *   rgcn_bwd_lean_supported(tile_rows)   1 when the kernel's LDS (dX tile + X tile + scratch per wave, 8 or 16 waves) fits
 *   rgcn_bwd_lean_f32                    arguments as described above with (slots, hdr) in place of (p_pack, chunk_rel);
 *                                        flags RGCN_F_DW_ATOMIC, RGCN_F_RELU (dX masked with X > 0: X is a ReLU's output and
 *                                        dX is wanted before it -- the caller's F.relu backward, models.py:196 / autograd)
 * Same autograd duals of layers.py:293-301 as above; R < 65536. */
RGCN_API int64_t rgcn_bwd_lean_slot_bytes(int64_t n_chunks);
RGCN_API int rgcn_bwd_lean_prepare_f32(const int32_t *p_pack, const int32_t *chunk_rel, int64_t n_chunks, void *slots, int32_t *hdr,
                                       void *stream);
/* The same from a plan WITHOUT packed slots (tiles taller than 255 rows, up to 512): p_src / p_dst (global destination row, < 0 = pad) /
 * p_val as rgcn_dev_plan_fill wrote them. */
RGCN_API int rgcn_bwd_lean_prepare_unpacked_f32(const int32_t *p_src, const int32_t *p_dst, const float *p_val, int32_t tile_rows,
                                                const int32_t *chunk_rel, int64_t n_chunks, void *slots, int32_t *hdr, void *stream);
RGCN_API int rgcn_bwd_lean_supported(int32_t tile_rows);
RGCN_API int rgcn_bwd_lean_f32(const float *G, const float *X, const float *Wt_packed, float *dX, float *dW, float *scratch,
                               const void *slots, const int32_t *hdr, const int32_t *run_ptr, int64_t n_tiles, int32_t tile_rows,
                               int64_t n_dst, int32_t R, int32_t flags, void *stream);
/* The same backward, block-tile form (the default when it applies): ONE destination tile per workgroup (plan built with that tile
 * height: 15 % bucket padding instead of 42 %), its chunks dealt to the 16 waves from an LDS counter; X tile (fp32) and dX tile (fp64,
 * updated with ds_add_f64 -- round 4; round 3: fp32 with compare-and-swap loops) shared, dW of ALL relations resident in LDS for the
 * workgroup's life and flushed once (dirty relations only).  LDS: 192 bytes per tile row + 16 KiB + R KiB (R / 4 KiB with
 * RGCN_F_DIAG4):
 *   rgcn_bwd_blk_max_rows(R, flags)      tallest tile that fits (0: none; at most 512) -- S1 (R = 101): 227, AM with diag4 (R = 267): 406
 *   rgcn_bwd_blk_supported(tile_rows, R, flags)
 *   rgcn_bwd_blk_rec_bytes(n_chunks)     size of the chunk records (176 bytes per chunk: 16 x {source row << 6, val}, 16 x u16 tile
 *                                        row << 6, relation)
 *   rgcn_bwd_blk_prepare_f32             the transposed plan -> records, once per plan: p_pack (tiles of up to 255 rows) or, with
 *                                        p_pack == NULL, p_src / p_dst (global destination row, < 0 = pad) / p_val as
 *                                        rgcn_dev_plan_fill wrote them (tiles of up to 1023 rows: 512 for the backward kernel, 1023 for
 *                                        rgcn_spmm_blk_f32); chunk_rel; n_chunks = m_pad / 16
 * Atomic flush only, sums in arrival order (dX: fp64 sums rounded once, so run-to-run differences are rare but possible;
 * RGCN_DETERMINISTIC=1 takes the lean kernel on 64-row tiles).
 * flags: RGCN_F_RELU; RGCN_F_DIAG4: the weights are block_diag() of 4 x 4 blocks (layers.py:243-244 at width 16) -- only the
 * diagonal blocks of dW_r are accumulated (the rest of dW stays zero), 256 bytes of LDS per relation instead of 1 KiB.
 * dbias (may be NULL): 16 floats, the bias gradient = column sums of
 * G's n_src rows, read on the side of the tile walk (replaces an rgcn_colsum_f32 launch; one fill zeroes dW and dbias when
 * dbias == dW + R * 256).  units (may be NULL = one unit per tile): [n_units][4] = {tile, first chunk, end chunk, flags} as
 * rgcn_plan_units_host makes them from the plan's tile pointer -- tiles of hub rows cut into pieces (RGCN_U_SHARED; n_split of them)
 * that different workgroups walk and whose dX rows are added to a zeroed dX.
 * Same autograd duals of layers.py:293-301 as above. */
RGCN_API int32_t rgcn_bwd_blk_max_rows(int32_t R, int32_t flags);
RGCN_API int rgcn_bwd_blk_supported(int32_t tile_rows, int32_t R, int32_t flags);
RGCN_API int64_t rgcn_bwd_blk_rec_bytes(int64_t n_chunks);
RGCN_API int rgcn_bwd_blk_prepare_f32(const int32_t *p_pack, const int32_t *p_src, const int32_t *p_dst, const float *p_val,
                                      int32_t tile_rows, const int32_t *chunk_rel, int64_t n_chunks, void *rec, void *stream);
RGCN_API int rgcn_bwd_blk_f32(const float *G, const float *X, const float *Wt_packed, float *dX, float *dW, const void *rec,
                              const int32_t *run_ptr, int64_t n_tiles, int32_t tile_rows, int64_t n_dst, int32_t R, int32_t flags,
                              float *dbias, int64_t n_src, const int32_t *units, int64_t n_units, int64_t n_split, void *stream);
/* The same backward, RELATION-OWNER form (round 6; the default on large static graphs with dense buckets -- S1): tall tiles (up to
 * rgcn_bwd_own_max_rows() = 767 rows) walked in soft-window order.  LDS holds the dX tile (doubles, ds_add_f64), the X tile and 1 KiB of
 * scratch per wave, no weight-gradient table: every relation belongs to ONE of the workgroup's rgcn_bwd_own_waves() waves, which walks
 * that relation's chunks only and keeps its dW in registers for the life of the kernel (rgcn_bwd_own_units() relation slots per
 * workgroup; one flush of global atomics at the end).
 *   rec       chunk records of rgcn_bwd_blk_prepare_f32 whose relation word is  rel | local << 16  (local = the relation's slot in its
 *             owner wave); inside a tile the chunks are grouped by owner wave
 *   own_ptr   [n_tiles * waves + 1]: own_ptr[tile * waves + w] = first chunk of wave w in the tile
 *   unit_rel  [units]: unit_rel[w * (units / waves) + local] = relation, -1 = unused slot
 * flags: RGCN_F_RELU; dbias as rgcn_bwd_blk_f32.  No hub pieces.  Sums in arrival order.  Same autograd duals of layers.py:293-301. */
RGCN_API int32_t rgcn_bwd_own_waves(void);
RGCN_API int32_t rgcn_bwd_own_units(void);
RGCN_API int32_t rgcn_bwd_own_max_rows(void);
RGCN_API int rgcn_bwd_own_f32(const float *G, const float *X, const float *Wt_packed, float *dX, float *dW, const void *rec,
                              const int32_t *own_ptr, const int32_t *unit_rel, int64_t n_tiles, int32_t tile_rows, int64_t n_dst,
                              int32_t R, int32_t flags, float *dbias, int64_t n_src, void *stream);
/* Soft-window plans (round 6), built on the device: the relation-tile plan of rgcn_spmm_blk_f32 / rgcn_bwd_own_f32 in the order those
 * kernels are fast on -- what replaces the reference's per-forward stack_matrices -> sum_sparse -> sparse COO pipeline (utils.py:143-166, :71-97;
 * layers.py:255-279) for them.  Messages (dst <- src, rel, val; alive may be NULL) are bucketed by (dst / tile_rows, rel), every bucket padded to a
 * multiple of 16 slots (pad: dst = -1, val = 0), the slots of a bucket sorted by src, the chunks (16 slots) of a tile ordered by their first source.
 *   rgcn_softwin_tmp_bytes(n)   device scratch for sorting / scanning up to n elements (pass it as tmp to both calls; n >= M, buckets + 1, groups + 1)
 *   rgcn_softwin_order          keys [M], keys_sorted [M], order [M], order_sorted [M]: scratch; bucket_cnt / bucket_base / bucket_first
 *                               [n_tiles * R + 1]: live messages per bucket, first slot of every bucket (entry n_tiles * R = m_pad), first live
 *                               message of every bucket in sorted order (last entry = number of live messages).  The caller reads m_pad and n_live back.
 *   rgcn_softwin_fill           s_src / s_dst / s_val [m_pad], ckeys / ckeys_sorted / cidx / cidx_sorted / crel [m_pad / 16]: scratch; group_cnt
 *                               [groups + 1]: scratch.  Outputs p_src / p_dst / p_val [m_pad], chunk_rel [m_pad / 16], group_ptr [groups + 1]
 *                               (first chunk of every group).  own_waves = 0: a group is a tile (group_ptr = the tile pointer).  own_waves > 0
 *                               (rgcn_bwd_own_f32): a group is (tile, owner wave); relation r is cut into parts[r] units unit_base[r] .. , chunk j of a
 *                               bucket belongs to unit unit_base[r] + j % parts[r], owned by wave unit_owner[u] under the local number unit_local[u];
 *                               chunk_rel = rel | local << 16. */
RGCN_API int64_t rgcn_softwin_tmp_bytes(int64_t n);
RGCN_API int rgcn_softwin_order(const int32_t *dst, const int32_t *src, const int32_t *rel, const uint8_t *alive, int64_t M, int64_t n_dst,
                                int64_t n_src, int32_t R, int32_t tile_rows, uint64_t *keys, uint64_t *keys_sorted, int32_t *order,
                                int32_t *order_sorted, int32_t *bucket_cnt, int32_t *bucket_base, int32_t *bucket_first, void *tmp,
                                int64_t tmp_bytes, void *stream);
RGCN_API int rgcn_softwin_fill(const int32_t *dst, const int32_t *src, const float *val, const uint64_t *keys_sorted, const int32_t *order_sorted,
                               int64_t n_live, int64_t n_dst, int64_t n_src, int32_t R, int32_t tile_rows, const int32_t *bucket_base,
                               const int32_t *bucket_first, int64_t m_pad, const int32_t *parts, const int32_t *unit_base,
                               const int32_t *unit_owner, const int32_t *unit_local, int32_t own_waves, int32_t *s_src, int32_t *s_dst, float *s_val,
                               uint64_t *ckeys, uint64_t *ckeys_sorted, int32_t *cidx, int32_t *cidx_sorted, int32_t *crel, int32_t *group_cnt,
                               int32_t *p_src, int32_t *p_dst, float *p_val, int32_t *chunk_rel, int32_t *group_ptr, void *tmp, int64_t tmp_bytes,
                               void *stream);
/* The same walk as a FORWARD kernel (round 5): out[dst] = bias + sum val X[src] W_r (layers.py:293-301 at width 16) on the FORWARD plan cut
 * into tall tiles (one per workgroup, up to rgcn_spmm_blk_max_rows() = 1023 rows: 128 bytes of LDS per row, no weight table) with the chunk
 * records of rgcn_bwd_blk_prepare_f32.  For layers whose (tile, relation) buckets on the wave-owned tiles of rgcn_spmm_f32 are mostly
 * padding and whose relations do not fit rgcn_spmm_csr_d16_f32's LDS (AM as shipped, layer 2: R = 267); one launch instead of
 * rgcn_spmm_scatter_f32 + rgcn_segment_gather_sum_f32.  W_packed: rgcn_pack_w16_f32 fragments; bias may be NULL; flags: RGCN_F_RELU
 * (the activation in the epilogue; not with hub pieces).  units / n_units / n_split as above (pieces add into a zeroed out; the
 * RGCN_U_FIRST piece adds the bias). */
RGCN_API int32_t rgcn_spmm_blk_max_rows(void);
RGCN_API int rgcn_spmm_blk_f32(const float *X, const float *W_packed, const float *bias, float *out, const void *rec,
                               const int32_t *run_ptr, int64_t n_tiles, int32_t tile_rows, int64_t n_dst, int32_t R, int32_t flags,
                               const int32_t *units, int64_t n_units, int64_t n_split, void *stream);
/* The same for graphs whose (tile, relation) buckets are sparse (AM: 267 relations), on the RELATION-major plan of the
 * two-pass path: one wave per work item gathers G[p_src] and X[p_dst] once per message and produces
 *   Y[slot, :] = val G[p_src] W_r^T   (slot order; pass 2 = rgcn_segment_gather_sum_f32 sums them per destination -> dX)
 *   dW[r]    += val X[p_dst]^T G[p_src]   (one flush of 256 fp32 atomics per item; dW is zeroed first)
 * i.e. two random row reads per message instead of the three of rgcn_spmm_scatter_f32 + rgcn_wgrad_f32.  d = 16 only.
 * flags: RGCN_F_RELU -- X = relu(.) of the producing layer (models.py:194): every transformed row is masked with X[p_dst] > 0 as it is
 * written, so the summed dX already is the gradient BEFORE that ReLU (the producer skips its threshold_backward launch). */
RGCN_API int rgcn_bwd_scatter_dw_f32(const float *G, const float *X, const float *Wt_packed, float *Y, float *dW,
                                     const int32_t *p_src, const int32_t *p_dst, const float *p_val,
                                     const int32_t *chunk_rel, const int32_t *items, int64_t n_items, int32_t R, int32_t d,
                                     int32_t flags, void *stream);
/* Wp[r][16k+f][c] = W[r][f][4k+c]: fragments of W_r^T straight from W (the feature-gradient kernels multiply by W^T). */
RGCN_API int rgcn_pack_w16t_f32(const float *W, float *Wp, int32_t R, void *stream);
/* rgcn_pack_w16_f32 and rgcn_pack_w16t_f32 in ONE launch (a training step needs both: forward and fused backward of the
 * hidden-16 layer; the Python wrapper caches the pair per weight tensor and version). */
RGCN_API int rgcn_pack_w16_pair_f32(const float *W, float *Wp, float *Wtp, int32_t R, void *stream);

/* Featureless layer (X = I, d_in = N): out[dst,:] = bias + sum val * table[rel*n_src + src, :].
 * Replaces torch.mm(adj, weights.view(R*N, d_out)) of layers.py:286-288 / :518-523. */
RGCN_API int rgcn_featureless_fwd_f32(const float *table, const float *bias, float *out, const int32_t *p_src,
                                      const int32_t *p_dst, const float *p_val, const int32_t *chunk_rel,
                                      const int32_t *units, int64_t n_units, int64_t n_split, int32_t tile_rows,
                                      int64_t n_dst, int64_t n_src, int32_t R, int32_t d_out, void *stream);

/* Block-diagonal weights (decomposition {type: block}): blocks [n_rel_blocks][nb][bi][bo], X [n_src][nb*bi] ->
 * out[row, b, :] = bias + sum over the row's messages of val * X[src, b, :] . blocks[rel, b].  Replaces
 * block_diag(self.blocks) (layers.py:243-244, :520-527: dense R x d_in x d_out) + the dense message passing (:297-300,
 * :536-541): 1/nb of the flops, no expanded weights.  Destination-major CSR as for rgcn_diag_spmm_f32; `units` may be NULL
 * (one unit per row, n_units = n_rows: the sync-free LP build has no host-side row statistics) -- then `rowptr` is read.
 * Messages with rel >= n_rel_blocks are skipped (the LP layer's dense self-loop relation, layers.py:514-527, is the
 * caller's).  flags: RGCN_F_TRANSPOSE_W = multiply by the transposed blocks (X is [.][nb*bo], out [.][nb*bi]: the feature
 * gradient on the transposed CSR), RGCN_F_RELU (not with shared units).  bi, bo <= 8 (rgcn_block_supported). */
/* Dense 16 x 16 weights W [R][16][16] on the same destination-major CSR, one pass: out[row, :] = bias + sum over the row's messages of
 * val * X[src, :] W[rel] (layers.py:293-301 at hidden 16) for graphs whose (tile, relation) buckets are sparse -- messages of MIXED
 * relations, the weight table (transposed, padded) resident in LDS: R <= 120 (rgcn_spmm_csr_d16_supported).  flags: RGCN_F_RELU
 * (not with shared units).  The two-pass route it replaces: rgcn_spmm_scatter_f32 + rgcn_segment_gather_sum_f32. */
RGCN_API int rgcn_spmm_csr_d16_supported(int32_t R);
RGCN_API int rgcn_spmm_csr_d16_f32(const float *X, const float *W, const float *bias, float *out, const int32_t *units, int64_t n_units,
                                   int64_t n_split, const int32_t *e_src, const int32_t *e_rel, const float *e_val, int64_t n_rows,
                                   int32_t R, int32_t flags, void *stream);
RGCN_API int rgcn_block_supported(int32_t bi, int32_t bo);
RGCN_API int rgcn_block_spmm_f32(const float *X, const float *blocks, const float *bias, float *out, const int32_t *units,
                                 const int32_t *rowptr, int64_t n_units, int64_t n_split, const int32_t *e_src,
                                 const int32_t *e_rel, const float *e_val, int64_t n_rows, int32_t n_rel_blocks, int32_t nb,
                                 int32_t bi, int32_t bo, int32_t flags, void *stream);

/* Its weight gradient: dblocks[rel, b] = sum over the messages of rel of val * X[src, b, :]^T G[dst, b, :] over a
 * RELATION-MAJOR plan (items as for rgcn_rel_wgrad_f32); dblocks is zeroed first, fp32 atomics across items. */
RGCN_API int rgcn_block_wgrad_f32(const float *X, const float *G, float *dblocks, const int32_t *p_src, const int32_t *p_dst,
                                  const float *p_val, const int32_t *chunk_rel, const int32_t *items, int64_t n_items,
                                  int32_t n_rel_blocks, int32_t nb, int32_t bi, int32_t bo, void *stream);

/* Diagonal-weight layer (diag_weight_matrix=True, the first layer of the reference's EmbeddingNodeClassifier,
 * models.py:272-280): out[row, :] = bias + sum over the row's messages of val * X[src, :] * w[rel, :].  Replaces
 * einsum('ij,kj->kij') + reshape + torch.mm(adj, fw) of layers.py:289-292 (which materialises the [R*N, d] product) --
 * d multiplies per message, nothing materialised.  Destination-major CSR: e_src / e_rel / e_val are the entries,
 * `units` = int32 [n_units][4] = {row, first entry, end entry, flags (RGCN_U_SHARED | RGCN_U_FIRST)}, one per row with
 * long rows cut into pieces (n_split = number of shared units; > 0 zeroes `out` first).  On the transposed CSR with G in
 * place of X it is the feature gradient of the same layer.  Any d. */
RGCN_API int rgcn_diag_spmm_f32(const float *X, const float *w, const float *bias, float *out, const int32_t *units,
                                int64_t n_units, int64_t n_split, const int32_t *e_src, const int32_t *e_rel,
                                const float *e_val, int64_t n_rows, int32_t R, int32_t d, void *stream);

/* Its weight gradient: dw[rel, j] = sum_{slots of rel} val * X[src, j] * G[dst, j] over a RELATION-MAJOR plan (items =
 * chunk ranges of one relation, as for rgcn_rel_wgrad_f32); dw ([R, d]) is zeroed first, fp32 atomics across items. */
RGCN_API int rgcn_diag_wgrad_f32(const float *X, const float *G, float *dw, const int32_t *p_src, const int32_t *p_dst,
                                 const float *p_val, const int32_t *chunk_rel, const int32_t *items, int64_t n_items,
                                 int32_t R, int32_t d, void *stream);

/* Its weight gradient: dtable[rel*n_src + src, :] += val * G[dst,:]; dtable
 * ([R*n_src, d_out]) is zeroed first. */
RGCN_API int rgcn_featureless_wgrad_f32(const float *G, float *dtable, const int32_t *p_src,
                                        const int32_t *p_dst, const float *p_val, const int32_t *chunk_rel,
                                        int64_t n_chunks, int64_t n_dst, int64_t n_src, int32_t R,
                                        int32_t d_out, void *stream);
/* The featureless layer on a destination-major CSR (graphs whose (tile, relation) buckets are sparse: the tile plan of an AIFB-sized
 * graph with 91 relations is 23 padded slots per message).  Same results as rgcn_featureless_fwd_f32 / rgcn_featureless_wgrad_f32
 * (reference layers.py:293-301 with the one-hot input folded into the weight table; relu != 0: the models' F.relu(self.rgc1()) --
 * models.py:194 -- in the epilogue, rows not cut into shared pieces only -- the same flag on the rgcn_gather_rows_sum*_f32).  units: [n_units][4] = {row, e0, e1, flags}
 * (hub rows cut into pieces, RGCN_U_SHARED); the weight-gradient form walks the n_entries CSR entries, rowptr gives an entry's row. */
RGCN_API int rgcn_featureless_csr_fwd_f32(const float *table, const float *bias, float *out, const int32_t *units, int64_t n_units,
                                          int64_t n_split, const int32_t *e_src, const int32_t *e_rel, const float *e_val,
                                          int64_t n_rows, int64_t n_src, int32_t R, int32_t d, int32_t relu, void *stream);
RGCN_API int rgcn_featureless_csr_wgrad_f32(const float *G, float *dtable, const int32_t *rowptr, const int32_t *e_src,
                                            const int32_t *e_rel, const float *e_val, int64_t n_entries, int64_t n_rows, int64_t n_src,
                                            int32_t R, int32_t d, void *stream);

/* db[j] = sum_n G[n, j]  (bias gradient: autograd dual of the `+ bias` of layers.py:305-306).  Two stages through
 * `scratch` (rgcn_colsum_scratch_floats(n, d) floats), no atomics, fixed summation order: bit-reproducible. */
RGCN_API int64_t rgcn_colsum_scratch_floats(int64_t n, int32_t d);
RGCN_API int rgcn_colsum_f32(const float *G, float *db, float *scratch, int64_t n, int32_t d, void *stream);

/* Featureless layer with basis decomposition, source-major (layers.py:241-242 + :286-288 without the R x N x d_out
 * table): out[s,:] = sum_{e=(s,r,o)} val_e sum_b comps[r,b] bases[b,o,:].  `bases` / `dbases`: basis_major != 0 -- the parameter's
 * own [B, N, d] layout (round 3: no transposed copy of the table, no transposed gradient); 0 -- node-major [N, B, d].  comps [R, B].  Messages in SOURCE-major CSR order: e_dst / e_rel / e_val [M]; `units` = int32 [n_units][4]
 * {row, first entry, end entry, flags (RGCN_U_SHARED | RGCN_U_FIRST)}: one unit per source node, long rows cut into
 * several (n_split = number of shared units; their results merge with fp32 atomics into a zeroed output).
 *   rgcn_fbasis_fwd_f32:  Y[e,:] = val_e * comps[r_e,:] . bases[o,:,:]            (d <= 64, ceil(B / (64/pow2(d))) <= 16)
 *   rgcn_fbasis_bwd_f32:  dbases[o,:,:] = sum_e val_e comps[r_e,:]^T (x) G[s_e,:];  T[e,b] = val_e <bases[o,b,:], G[s_e,:]>
 *                         (B <= 64, d <= 16; dbases or T may be NULL)
 *   rgcn_gather_rows_sum_f32: out[row,:] = (bias) + sum_{j in unit ranges of row} Y[perm[j],:]   (perm NULL = identity; w <= 64)
 *     -- rows of Y by destination give the layer output, rows of T by relation give dcomps.
 * RGCN_EUNSUPPORTED outside the stated limits (the caller falls back to rgcn_basis_aggregate_f32). */
RGCN_API int rgcn_fbasis_fwd_f32(const float *bases, const float *comps, float *Y, const int32_t *e_rel,
                                 const float *e_val, const int32_t *units, int64_t n_units, int64_t n_nodes,
                                 int32_t R, int32_t B, int32_t d, int32_t basis_major, void *stream);
RGCN_API int rgcn_fbasis_bwd_f32(const float *bases, const float *comps, const float *G, float *dbases, float *T,
                                 const int32_t *e_dst, const int32_t *e_rel, const float *e_val,
                                 const int32_t *units, int64_t n_units, int64_t n_split, int64_t n_nodes, int32_t R,
                                 int32_t B, int32_t d, int32_t basis_major, void *stream);
/* The same layer with SMALL blocks (B <= 4: S2 of SURVEY 8d has B = 2, d = 16 -- a node's block is one 128-byte line), backward in ONE
 * walk of the source-major CSR (rowptr / p_src = the messages' destination rows, whose G rows are gathered / p_rel / p_val as
 * rgcn_basis_aggregate_f32 takes them): dbases[o, b, :] = sum_e val_e comps[r_e, b] G[s_e, :] (node-major [N, B, d], written once per
 * node) AND dcomps[r, b] = sum_e val_e <table[o_e, b, :], G[s_e, :]> (summed in an LDS table of doubles per workgroup, one flush each;
 * dcomps is zeroed here).  table: node-major [N, B, d].  d a power of two, 4 .. 64.  Replaces rgcn_basis_aggregate_f32 +
 * rgcn_basis_dcomps_f32 (two gathers per message in the latter) on that route: the autograd duals of layers.py:241-242 + :286-288. */
RGCN_API int rgcn_fbasis_small_supported(int32_t R, int32_t B, int32_t d);
RGCN_API int rgcn_fbasis_small_bwd_f32(const float *G, const float *table, const float *comps, float *dbases, float *dcomps,
                                       const int32_t *rowptr, const int32_t *p_src, const int32_t *p_rel, const float *p_val,
                                       int64_t n_rows, int32_t R, int32_t B, int32_t d, int32_t basis_major, void *stream);
/* rgcn_fbasis_bwd_f32 with dcomps finished on the chip (round 4): no T scratch, no relation-major pass 2 -- t_e[b] is added to a per-workgroup
 * LDS table of doubles (R x B x 8 bytes <= 120 KiB: rgcn_fbasis_bwd_dc_supported) and every workgroup flushes once into dcomps (zeroed here). */
RGCN_API int rgcn_fbasis_bwd_dc_supported(int32_t R, int32_t B, int32_t d);
RGCN_API int rgcn_fbasis_bwd_dc_f32(const float *bases, const float *comps, const float *G, float *dbases, float *dcomps,
                                    const int32_t *e_dst, const int32_t *e_rel, const float *e_val, const int32_t *units,
                                    int64_t n_units, int64_t n_split, int64_t n_nodes, int32_t R, int32_t B, int32_t d,
                                    int32_t basis_major, void *stream);
RGCN_API int rgcn_gather_rows_sum_f32(const float *Y, const int32_t *perm, const int32_t *units, int64_t n_units,
                                      int64_t n_split, const float *bias, float *out, int64_t n_rows, int32_t w,
                                      int32_t relu, void *stream);

/* The same layer on a table far beyond the caches, IN the parameter's own [B, N, d] layout and software-pipelined (round 4,
 * rgcn_fbasis_tile.hip; reference layers.py:241-242 + :286-288 with nc-AM.yaml's B = 40, d = 10: a 2.67 GB table): persistent
 * 1024-thread workgroups walk TILES of 16 consecutive source nodes, whose blocks are staged into LDS (or whose gradient is written
 * out of LDS) with aligned 16-byte accesses -- no node-major copy of the table, no transposed gradient.  rowptr [N + 1] = the
 * source-major CSR's row pointers; e_dst / e_rel / e_val [M = n_messages] its entries.
 * mode 0: a tile's messages are dealt evenly over the workgroup's 16 waves (any degree distribution); mode 1: wave w takes node w of the tile and
 * runs its messages through the matrix cores sixteen at a time (a node costs about the same for 1 or 16 messages: the waves stay in step) --
 * for graphs without hub sources (the caller decides from the largest source degree; torch_rgcn/_native.py: <= 4096).
 *   rgcn_fbasis_tile_supported -> bit 0: forward, bit 1: backward, bits 2 / 3: the same in mode 1 (LDS budget: forward R x 64 floats + 2 tiles; backward R x B doubles + 2 tiles
 *                                 and 2 tiles of doubles + R x B floats; B <= 64, d <= 16, N >= 16)
 *   rgcn_fbasis_tile_fwd_f32:     Y[e, 0..ys) = val_e * comps[r_e,:] . bases[:,o,:], zero padded to ys = rgcn_fbasis_tile_ystride(d) = pow2(d) >= 4 columns
 *   rgcn_gather_rows_sum4_f32:    out[row, 0..w) = (bias) + sum over the row's units of Y[perm[j], 0..w); Y rows ys = 4 / 8 / 16 floats; out rows
 *                                 out_stride floats apart (w <= out_stride <= ys), columns w .. out_stride written as zeros
 *   rgcn_fbasis_tile_bwd_f32:     dbases [B, N, d] (written once, no pre-zeroing) and / or dcomps [R, B] (either may be NULL); sums in LDS
 *                                 doubles (ds_add_f64): not bit-reproducible.  G [N, >= d] with a row stride of g_stride floats (the zero-padded
 *                                 [N, 16] rows the width-16 kernels of the next layer hand back are taken as they are).  Round 5: in mode 1
 *                                 with both gradients wanted the two sums come from ONE walk (one gather of G, one staged tile that is also
 *                                 the gradient tile) whenever R x B doubles + R x B floats + one tile + the waves' strips fit the LDS:
 *                                 rgcn_fbasis_tile_bwd_fused_gn -> rows per strip (16 / 12 / 8) or 0; mode 3 = mode 1 on the two kernels */
RGCN_API int rgcn_fbasis_tile_supported(int32_t R, int32_t B, int32_t d, int64_t n_nodes);
RGCN_API int rgcn_fbasis_tile_ystride(int32_t d);
RGCN_API int rgcn_fbasis_tile_fwd_f32(const float *bases, const float *comps, float *Y, const int32_t *rowptr, const int32_t *e_rel,
                                      const float *e_val, int64_t n_messages, int64_t n_nodes, int32_t R, int32_t B, int32_t d,
                                      int32_t mode, void *stream);
RGCN_API int rgcn_gather_rows_sum4_f32(const float *Y, int32_t ys, const int32_t *perm, const int32_t *units, int64_t n_units,
                                       int64_t n_split, const float *bias, float *out, int64_t n_rows, int32_t w, int32_t out_stride,
                                       int32_t relu, void *stream);
RGCN_API int rgcn_fbasis_tile_bwd_f32(const float *bases, const float *comps, const float *G, int32_t g_stride, float *dbases, float *dcomps,
                                      const int32_t *rowptr, const int32_t *e_dst, const int32_t *e_rel, const float *e_val,
                                      int64_t n_messages, int64_t n_nodes, int32_t R, int32_t B, int32_t d, int32_t mode, void *stream);
RGCN_API int rgcn_fbasis_tile_bwd_fused_gn(int32_t R, int32_t B, int32_t d, int64_t n_nodes);

/* Classifier head of the node-classification experiments, one launch: loss = mean cross-entropy of the logits' LABELLED rows and
 * dlogits [N, C] = d loss / d logits (zero rows for unlabelled nodes).  Replaces `criterion(model()[train_idx, :], train_lbl)` with
 * nn.CrossEntropyLoss() and its autograd graph (reference experiments/classify_nodes.py:107-110, :129): row_label [N] = class of the
 * node or -1, lab_rows [n_lab] = the labelled nodes (each once).  C <= 64.  Rows of logits and dlogits are ld >= C floats apart (a layer's
 * zero-padded output read in place; columns C .. ld of dlogits are written 0). */
RGCN_API int rgcn_ce_head_f32(const float *logits, const int32_t *row_label, const int32_t *lab_rows, float *loss, float *dlogits,
                              int64_t N, int32_t C, int32_t ld, int32_t n_lab, void *stream);

/* Loss head of the link-prediction experiments, one launch: loss = mean binary cross-entropy with logits over the T scored triples and
 * dscores [T] = d loss / d scores = (sigmoid(score) - label) / T.  Replaces F.binary_cross_entropy_with_logits(predictions, train_lbl)
 * and its autograd graph (reference experiments/predict_links.py:152-153).  workspace: rgcn_bce_head_workspace_bytes() bytes, ZEROED
 * once by the caller before the first call and kept between calls (the launch leaves it zeroed again); partial sums in double, added in
 * block order by the last block to finish: bit-reproducible. */
RGCN_API int rgcn_bce_head_workspace_bytes(void);
RGCN_API int rgcn_bce_head_f32(const float *scores, const float *labels, float *loss, float *dscores, void *workspace, int64_t T,
                               void *stream);

/* Zero-padding / cropping of the two trailing dimensions of a [A][B][C] tensor into [A][Bd][Cd], with an optional 1-D tensor (n1 -> n1d
 * elements) in the same launch: how widths that are no multiple of 16 (classifier outputs: layers.py weights [R, 16, 4], bias [4]) reach
 * the MFMA block kernels and how their gradients come back -- one launch where torch.nn.functional.pad and the gradient slices cost
 * two each (a fill and a copy).  No reference counterpart (plumbing). */
RGCN_API int rgcn_resize3_f32(const float *src, float *dst, int64_t A, int32_t B, int32_t C, int32_t Bd, int32_t Cd, const float *src1,
                              float *dst1, int32_t n1, int32_t n1d, void *stream);

/* ------------------------------------------------------------------ dense contractions on the matrix cores
 * Basis decomposition (layers.py:241-242, :468-469: W_r = sum_b comps[r,b] bases[b]) at large width: the layer is
 * out = ag @ flat(bases) + bias with ag[s, b, :] = sum_e comps[r_e, b] val_e X[o_e, :].
 * Forward = rgcn_basis_aggregate_f32 + rgcn_gemm_f32 (the fused aggregate-in-LDS kernel of rounds 1-4 was measured slower and is gone). */
/* C[M,N] = op(A) op(B) (+ bias[n]); fp32 MFMA, LDS-tiled (128 x 128 x 16).  A is [M,K] (lda) or, with RGCN_G_TRANS_A,
 * stored [K,M]; B is [K,N] (ldb) or, with RGCN_G_TRANS_B, stored [N,K].  split_k > 1 cuts K into slices whose partial
 * products go to `scratch` (rgcn_gemm_scratch_floats) and are summed in a fixed order.  The backward of the basis path
 * (d_ag = g flat^T, dbases = ag^T g) and the weight assembly / wide-block products of layers.py:241-244 run on it. */
#define RGCN_G_TRANS_A 1
#define RGCN_G_TRANS_B 2
RGCN_API int64_t rgcn_gemm_scratch_floats(int64_t M, int64_t N, int64_t K, int32_t split_k);
RGCN_API int rgcn_gemm_f32(const float *A, const float *B, const float *bias, float *C, float *scratch, int64_t M,
                           int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int32_t flags, int32_t split_k,
                           void *stream);

/* Undecomposed weights above width 64 (layers.py:293-301 with W [R, d_in, d_out], d = 100, 200, ...): the relation-major
 * plan (rgcn_dev_plan_fill with tile_rows >= n_dst; work items of <= 8 chunks = 128 slots of one relation) as the row
 * blocks of an LDS-tiled MFMA GEMM with gathered rows:
 *   rgcn_rel_rows_f32               Y[slot, :] = val[slot] * Xs[p_src[slot], :] @ W[rel]        (Y: [slots, d_out], slot order)
 *   rgcn_segment_gather_sum_wide_f32  out[row, :] = bias + sum_j Y[perm[j], :] over the row's CSR range (any width; the sum of a row is
 *                                   carried in doubles and rounded once: a hub row is a chain of as many additions as it has messages)
 *   rgcn_rel_wgrad_f32              dW[rel] += sum_slots Xs[p_src[slot], :]^T (val[slot] G[p_dst[slot], :])   (dW zeroed first,
 *                                   fp32 atomics across the items of a relation)
 * The feature gradient is rgcn_rel_rows_f32 on the transposed relation-major plan with G and W^T. */
RGCN_API int rgcn_rel_rows_f32(const float *Xs, const float *W, float *Y, const int32_t *p_src, const float *p_val,
                               const int32_t *chunk_rel, const int32_t *items, int64_t n_items, int32_t R, int32_t d_in,
                               int32_t d_out, void *stream);
RGCN_API int rgcn_rel_wgrad_f32(const float *Xs, const float *G, float *dW, const int32_t *p_src, const int32_t *p_dst,
                                const float *p_val, const int32_t *chunk_rel, const int32_t *items, int64_t n_items,
                                int32_t R, int32_t d_in, int32_t d_out, void *stream);
RGCN_API int rgcn_segment_gather_sum_wide_f32(const float *Y, const int32_t *perm, const int32_t *rowptr, const float *bias,
                                              float *out, int64_t n_rows, int32_t d, int32_t flags, void *stream);

/* DistMult decoder (torch_rgcn/layers.py:86-98):
 * scores[t] = sum_k nodes[s,k] rel[p,k] nodes[o,k] (+ sbias[s] + pbias[p] + obias[o]);
 * triples int64 [T,3] on the device.  Biases may all be NULL.  Triples whose s / o are outside [0, n_nodes) or whose p
 * is outside [0, n_rel) are NOT touched (the reference's fancy indexing raises IndexError): their score is 0 and
 * *err_flag (device int32, may be NULL) is set to 1 -- the caller turns it into the exception.
 * rank_counts / ranks (both NULL or both set): the counting pass of the backward's two CSRs of the scored triples (by subject, by
 * object) rides along -- rank_counts (2 n_nodes + 3 ints, zeroed here) receives the row sizes (subject row k at [1 + k], object row k at
 * [n_nodes + 2 + k]) and ranks (2 T ints) each valid triple's rank within its subject row and its object row (one returning atomic
 * each, hidden behind the wave's row loads); rgcn_distmult_csr_place finishes the CSRs once the score gradients exist. */
RGCN_API int rgcn_distmult_fwd_f32(const int64_t *triples, int64_t T, const float *nodes, const float *rel,
                                   const float *sbias, const float *pbias, const float *obias, float *scores,
                                   int64_t n_nodes, int32_t n_rel, int32_t d, int32_t *err_flag, int32_t *rank_counts,
                                   int32_t *ranks, void *stream);
/* The two CSRs rgcn_distmult_bwd_all_f32 / rgcn_distmult_bwd_nodes_f32 walk, from the ranks and row sizes the scoring kernel left
 * (rgcn_distmult_fwd_f32) and the score gradients gs [T]: an exclusive scan of rank_counts in place (scan_tmp: (2 N + 2) / 1024 + 3
 * ints; NULL = rank_counts already holds the scanned row starts of an earlier call) and one pass over the triples without atomics.
 * Afterwards rank_counts + 1 is the row pointer by subject (N + 1 ints; entries = (object, predicate, gs)), rank_counts + N + 2 the row
 * pointer by object (entries = (subject, predicate, gs)), both indexing the same array of 2 T entries of four int32 {other end,
 * predicate, gs[t] (bit pattern), 0} (16 bytes: one store here, one load per lane in the walk; 16-byte aligned).  Triples outside the
 * ranges are skipped as the scoring kernel skipped them. */
RGCN_API int rgcn_distmult_csr_place(const int64_t *triples, int64_t T, int64_t N, int32_t n_rel, const int32_t *ranks,
                                     int32_t *rank_counts, int32_t *scan_tmp, const float *gs, int32_t *entries, void *stream);
/* Gradients of sum_t gs[t] * scores[t]; dnodes / drel (and the bias grads when
 * non-NULL) are zeroed first and accumulated with fp32 atomics.  Out-of-range triples are skipped.  dnodes may be NULL
 * (see rgcn_distmult_bwd_nodes_f32). */
RGCN_API int rgcn_distmult_bwd_f32(const int64_t *triples, int64_t T, const float *nodes, const float *rel,
                                   const float *gs, float *dnodes, float *drel, float *dsbias, float *dpbias,
                                   float *dobias, int64_t n_nodes, int32_t n_rel, int32_t d, void *stream);
/* Entity gradients without atomics: dnodes[n] = sum_{t: s_t = n} g_t rel[p_t] * nodes[o_t] + sum_{t: o_t = n} g_t rel[p_t] *
 * nodes[s_t], one wave per entity over the two CSRs of the scored triples (rgcn_distmult_csr_place: rows = subject, entries =
 * (object, predicate, g); rows = object, entries = (subject, predicate, g)).
 * dnodes is fully written.  rgcn_distmult_bwd_f32 with dnodes = NULL then yields the relation / bias gradients only. */
RGCN_API int rgcn_distmult_bwd_nodes_f32(const int32_t *rowptr_s, const int32_t *rowptr_o, const int32_t *entries, const float *nodes,
                                         const float *rel, float *dnodes, int64_t n_nodes, int32_t d, void *stream);

/* All DistMult gradients from the two CSRs of the scored triples (by subject: entries = (object, predicate, g); by object:
 * entries = (subject, predicate, g)) -- entity gradient as above, relation gradient d_rel[p] += g x_s x_o accumulated in
 * one LDS table of doubles per workgroup (ds_add_f64; no predicate sort, no second pass over the triples), bias gradients
 * (d_sbias[n] / d_obias[n] = the row sums of g, d_pbias via the LDS table) when the three pointers are set.  The autograd dual of
 * layers.py:87-101.
 * Needs n_rel (d + 1) <= 4096 (rgcn_distmult_bwd_all_supported); d_rel / d_pbias are zeroed first. */
RGCN_API int rgcn_distmult_bwd_all_supported(int32_t n_rel, int32_t d);
RGCN_API int rgcn_distmult_bwd_all_f32(const int32_t *rowptr_s, const int32_t *rowptr_o, const int32_t *entries, const float *nodes,
                                       const float *rel, float *dnodes, float *drel, float *dsbias, float *dpbias, float *dobias,
                                       int64_t n_nodes, int32_t n_rel, int32_t d, void *stream);

/* Ranking evaluator (utils/misc.py:60-110 + torch_rgcn/layers.py:87-98 on the expanded
 * [bn, N, 3] candidate tensor, which is never built here).  For each of the Q test triples in `batch` (int64
 * [Q,3], device) every entity n is scored as its head (head != 0: (n, p, o)) or tail ((s, p, n)):
 *   scores[q, n] = sum_k (nodes[fixed_q,k] * rel[p_q,k]) * nodes[n,k]  (+ sbias + pbias + obias as layers.py:96)
 * fp32 MFMA NT product.  qvec [Q,d] and qbias [2Q] (only with biases) are caller-provided scratch; scores is
 * [Q, n_nodes] row-major.  Indices are NOT range-checked on the device (the Python wrapper asserts). */
RGCN_API int rgcn_distmult_score_all_f32(const int64_t *batch, int64_t Q, int32_t head, const float *nodes,
                                         const float *rel, const float *sbias, const float *pbias,
                                         const float *obias, float *qvec, float *qbias, float *scores,
                                         int64_t n_nodes, int32_t n_rel, int32_t d, void *stream);
/* filter_scores (utils/misc.py:40-58): scores[filt_q[e], filt_n[e]] = -inf for the F known true completions that
 * are not the target (list built by the caller; duplicates allowed). */
RGCN_API int rgcn_rank_filter_f32(float *scores, int64_t Q, int64_t n_nodes, const int32_t *filt_q,
                                  const int32_t *filt_n, int64_t F, void *stream);
/* utils/misc.py:93-96: per query, greater[q] = #{n : scores[q,n] > scores[q,target_q]} and
 * ties[q] = #{n : scores[q,n] == scores[q,target_q]} (the target included); target = batch[q,0] (head) or
 * batch[q,2].  rank = greater + (ties - 1) / 2 + 1 is left to the caller (misc.py:99-101). */
RGCN_API int rgcn_rank_count_f32(const float *scores, const int64_t *batch, int64_t Q, int32_t head,
                                 int64_t n_nodes, int64_t *greater, int64_t *ties, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RGCN_HIP_H */
