/*
 * rgcn_oracle.c -- CPU oracle for the R-GCN message-passing hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package (torch-rgcn_amd/) may
 * link, load or call this file.  Legitimate users: tests/, __graft_entry__.smoke()
 * and the cpu_baseline leg of bench.py -- always as the checker, never as the
 * thing that is measured or shipped.
 *
 * It restates, as scalar loops over an edge list, what thiviyanT/torch-rgcn does
 * with ATen sparse ops.  Each function names the reference lines it follows
 * (paths relative to the reference checkout):
 *
 *   oracle_add_inverse_and_self   torch_rgcn/utils.py:127-141
 *   oracle_lp_augment             torch_rgcn/utils.py:100-124, layers.py:481-487
 *   oracle_stack_matrices         torch_rgcn/utils.py:143-166
 *   oracle_sum_sparse             torch_rgcn/utils.py:71-97
 *   oracle_edge_norm              torch_rgcn/layers.py:263-273 (NC), :498-510 (LP)
 *   oracle_rgcn_forward           torch_rgcn/layers.py:286-306, :518-556
 *   oracle_rgcn_backward          autograd duals of the above
 *   oracle_distmult_*             torch_rgcn/layers.py:77-98
 *
 * Parity pin: the .npz fixtures under tests/golden/ were produced by importing the
 * reference itself (tests/golden/gen_golden.py) and tests/test_oracle_golden.py
 * checks every function here against them.
 *
 * Arithmetic: inputs/outputs fp32 (as the reference); sums are carried in double
 * and rounded once, so the oracle sits at or below the reference's own fp32
 * round-off (about 1e-7 relative) from the exact result.
 *
 * Threads (round 5): the two layer loops run "owner computes" over a few OpenMP
 * threads -- thread t walks the WHOLE edge list in order and takes the messages
 * whose accumulator row it owns (destination row for the forward, source row for
 * dX, relation -- or table row -- for dW).  Every accumulator still receives its
 * terms in edge-list order: the results are bit for bit those of the one-thread
 * loop (ORACLE_THREADS=1 in the environment runs it), the full-size parity tests
 * just stop waiting 15 s per case for it.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* threads of the layer loops: min(16, cores), ORACLE_THREADS overrides; small edge lists stay on one */
static int oracle_threads(int64_t M) {
#ifdef _OPENMP
  const char *e = getenv("ORACLE_THREADS");
  int t = e ? atoi(e) : omp_get_max_threads();
  if (t > 16) t = 16;
  if (t < 1 || M < 200000) t = 1;
  return t;
#else
  (void)M;
  return 1;
#endif
}

#define ORACLE_OK 0
#define ORACLE_EINVAL 1
#define ORACLE_ENOMEM 2
#define ORACLE_ERANGE 3
#define NZ(x) ((size_t)((x) > 0 ? (x) : 1))

/* ---- index shuffles ------------------------------------------------------ */

/* [T | inverse(T) | self loops], fixed block order; pure index arithmetic so
 * it is deliberately oblivious to the sign/range of node ids (the reference
 * test uses negative ids).  out has (2E+N) rows of 3. */
int oracle_add_inverse_and_self(const int64_t *T, int64_t E, int64_t N,
                                int64_t R0, int64_t *out) {
  if (E < 0 || N < 0) return ORACLE_EINVAL;
  for (int64_t e = 0; e < E; ++e) {
    out[3 * e + 0] = T[3 * e + 0];
    out[3 * e + 1] = T[3 * e + 1];
    out[3 * e + 2] = T[3 * e + 2];
    int64_t *inv = out + 3 * (E + e);
    inv[0] = T[3 * e + 2];
    inv[1] = T[3 * e + 1] + R0;
    inv[2] = T[3 * e + 0];
  }
  for (int64_t i = 0; i < N; ++i) {
    int64_t *sl = out + 3 * (2 * E + i);
    sl[0] = i;
    sl[1] = 2 * R0;
    sl[2] = i;
  }
  return ORACLE_OK;
}

/* Link-prediction augmentation.  generate_self_loops returns `triples ++ kept
 * self loops`, so the layer ends up with [T | inv(T) | T | SL]: the original
 * block appears twice.  keep[i] != 0 keeps self loop i (the Bernoulli draw is
 * made by the caller).  out must hold (3E + N) rows; *M_out receives the rows
 * actually written, *n_self the size of the third block (E + #kept). */
int oracle_lp_augment(const int64_t *T, int64_t E, int64_t N, int64_t R0,
                      const uint8_t *keep, int64_t *out, int64_t *M_out,
                      int64_t *n_self) {
  if (E < 0 || N < 0) return ORACLE_EINVAL;
  int64_t w = 0;
  for (int64_t e = 0; e < E; ++e, ++w) {
    out[3 * w + 0] = T[3 * e + 0];
    out[3 * w + 1] = T[3 * e + 1];
    out[3 * w + 2] = T[3 * e + 2];
  }
  for (int64_t e = 0; e < E; ++e, ++w) {
    out[3 * w + 0] = T[3 * e + 2];
    out[3 * w + 1] = T[3 * e + 1] + R0;
    out[3 * w + 2] = T[3 * e + 0];
  }
  for (int64_t e = 0; e < E; ++e, ++w) {
    out[3 * w + 0] = T[3 * e + 0];
    out[3 * w + 1] = T[3 * e + 1];
    out[3 * w + 2] = T[3 * e + 2];
  }
  int64_t kept = 0;
  for (int64_t i = 0; i < N; ++i) {
    if (keep && !keep[i]) continue;
    out[3 * w + 0] = i;
    out[3 * w + 1] = 2 * R0;
    out[3 * w + 2] = i;
    ++w;
    ++kept;
  }
  *M_out = w;
  *n_self = E + kept;
  return ORACLE_OK;
}

/* Sparse indices of the stacked adjacency.  Row is always the subject;
 * vertical: (p*N + s, o) in an (R*N, N) matrix; horizontal: (s, p*N + o) in
 * (N, R*N).  Returns ORACLE_ERANGE where the reference's asserts would fire. */
int oracle_stack_matrices(const int64_t *Tp, int64_t M, int64_t N, int64_t R,
                          int vertical, int64_t *idx, int64_t *size2) {
  size2[0] = vertical ? R * N : N;
  size2[1] = vertical ? N : R * N;
  int bad = 0;
  for (int64_t e = 0; e < M; ++e) {
    int64_t s = Tp[3 * e + 0], p = Tp[3 * e + 1], o = Tp[3 * e + 2];
    int64_t fr = vertical ? p * N + s : s;
    int64_t to = vertical ? o : p * N + o;
    idx[2 * e + 0] = fr;
    idx[2 * e + 1] = to;
    if (fr >= size2[0] || to >= size2[1]) bad = 1;
  }
  return bad ? ORACLE_ERANGE : ORACLE_OK;
}

static int cmp_i64(const void *a, const void *b) {
  int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
  return (x > y) - (x < y);
}

/* For every entry: the sum of `vals` over all entries that share its row
 * (row_norm) or its column (!row_norm).  The reference builds a sparse matrix
 * and multiplies by ones; this is the same number computed by sort + run sum.
 * vals == NULL means all ones. */
int oracle_sum_sparse(const int64_t *idx, const float *vals, int64_t M,
                      int row_norm, float *sums) {
  if (M == 0) return ORACLE_OK;
  int64_t *key = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)M);
  if (!key) return ORACLE_ENOMEM;
  for (int64_t e = 0; e < M; ++e) {
    key[2 * e + 0] = idx[2 * e + (row_norm ? 0 : 1)];
    key[2 * e + 1] = e;
  }
  qsort(key, (size_t)M, 2 * sizeof(int64_t), cmp_i64);
  int64_t a = 0;
  while (a < M) {
    int64_t b = a;
    double acc = 0.0;
    while (b < M && key[2 * b] == key[2 * a]) {
      acc += vals ? (double)vals[key[2 * b + 1]] : 1.0;
      ++b;
    }
    for (int64_t t = a; t < b; ++t) sums[key[2 * t + 1]] = (float)acc;
    a = b;
  }
  free(key);
  return ORACLE_OK;
}

/* The per-edge adjacency value, by the literal procedure of the layer:
 *   vertical   : val = 1 / #{edges with the same (p, s)}
 *   horizontal : k = #{edges with the same (p, o)};  then the "transpose trick"
 *                c = [k[n:2n] | k[0:n] | k[M-i:M]],  val = 1 / c
 * n = n_swap and i = i_tail are what the layer passes:
 *   NC: n = (M - N) / 2 (integer division), i = N            layers.py:235-236,269-271
 *   LP: n = E,          i = E + #kept self loops             layers.py:507-509
 * If 2n + i != M the reference raises a shape error; so do we. */
int oracle_edge_norm(const int64_t *Tp, int64_t M, int64_t N, int64_t R,
                     int vertical, int64_t n_swap, int64_t i_tail, float *val) {
  int64_t *idx = (int64_t *)malloc(sizeof(int64_t) * 2 * NZ(M));
  float *k = (float *)malloc(sizeof(float) * NZ(M));
  if (!idx || !k) { free(idx); free(k); return ORACLE_ENOMEM; }
  int64_t size2[2];
  int rc = oracle_stack_matrices(Tp, M, N, R, vertical, idx, size2);
  if (rc == ORACLE_OK) rc = oracle_sum_sparse(idx, NULL, M, vertical, k);
  if (rc == ORACLE_OK) {
    if (vertical) {
      for (int64_t e = 0; e < M; ++e) val[e] = 1.0f / k[e];
    } else if (2 * n_swap + i_tail != M || n_swap < 0 || i_tail < 0) {
      rc = ORACLE_EINVAL;
    } else {
      int64_t n = n_swap;
      for (int64_t e = 0; e < n; ++e) val[e] = 1.0f / k[n + e];
      for (int64_t e = 0; e < n; ++e) val[n + e] = 1.0f / k[e];
      for (int64_t e = 0; e < i_tail; ++e) val[2 * n + e] = 1.0f / k[M - i_tail + e];
    }
  }
  free(idx);
  free(k);
  return rc;
}

/* ---- the layer ----------------------------------------------------------- */

/* out[s,:] = sum_e val_e * X[o_e,:] @ W[p_e] + bias.
 * X == NULL is the featureless layer: d_in == N, X = I, so the message is the
 * weight-table row W[p_e, o_e, :].  W is dense (R, d_in, d_out) row-major; any
 * basis / block / diagonal structure is expanded by the caller (oracle.py).
 * Message direction object -> subject (row index = subject, utils.py:153-158). */
int oracle_rgcn_forward(const int64_t *Tp, const float *val, int64_t M,
                        int64_t N, int64_t R, int64_t d_in, int64_t d_out,
                        const float *X, const float *W, const float *bias,
                        float *out) {
  double *acc = (double *)calloc(NZ(N * d_out), sizeof(double));
  if (!acc) return ORACLE_ENOMEM;
  int rc = ORACLE_OK;
  const int nt = oracle_threads(M);
#ifdef _OPENMP
#pragma omp parallel num_threads(nt)
#endif
  {
#ifdef _OPENMP
    const int me = omp_get_thread_num();
#else
    const int me = 0;
#endif
    for (int64_t e = 0; e < M; ++e) {
      int64_t s = Tp[3 * e + 0], p = Tp[3 * e + 1], o = Tp[3 * e + 2];
      if (s < 0 || s >= N || o < 0 || o >= N || p < 0 || p >= R) { rc = ORACLE_ERANGE; break; }   /* (every thread meets it) */
      if ((int)(s % nt) != me) continue;            /* owner computes: the destination row */
      double v = (double)val[e];
      double *dst = acc + s * d_out;
      if (!X) {
        const float *w = W + (p * N + o) * d_out;
        for (int64_t j = 0; j < d_out; ++j) dst[j] += v * (double)w[j];
      } else {
        const float *x = X + o * d_in;
        const float *w = W + p * d_in * d_out;
        for (int64_t i = 0; i < d_in; ++i) {
          double xv = v * (double)x[i];
          const float *wr = w + i * d_out;
          for (int64_t j = 0; j < d_out; ++j) dst[j] += xv * (double)wr[j];
        }
      }
    }
  }
  if (rc == ORACLE_OK) {         /* (independent elements: the rounding loop may be split over the threads) */
#ifdef _OPENMP
#pragma omp parallel for num_threads(nt) schedule(static)
#endif
    for (int64_t n = 0; n < N; ++n)
      for (int64_t j = 0; j < d_out; ++j)
        out[n * d_out + j] = (float)(acc[n * d_out + j] + (bias ? (double)bias[j] : 0.0));
  }
  free(acc);
  return rc;
}

/* Given g = dL/dout (N, d_out):
 *   dX[o,:]   = sum_e val_e * g[s_e,:] @ W[p_e]^T           (skipped if dX NULL or X NULL)
 *   dW[p]     = sum_{e in p} val_e * X[o_e,:]^T g[s_e,:]    (featureless: dW[p,o,:] += val*g[s])
 *   db[j]     = sum_n g[n,j]                                (skipped if db NULL)
 * val carries no gradient (it is built under no_grad / from constants). */
int oracle_rgcn_backward(const int64_t *Tp, const float *val, int64_t M,
                         int64_t N, int64_t R, int64_t d_in, int64_t d_out,
                         const float *X, const float *W, const float *g,
                         float *dX, float *dW, float *db) {
  int64_t wsz = R * d_in * d_out;
  double *aW = dW ? (double *)calloc(NZ(wsz), sizeof(double)) : NULL;
  double *aX = (dX && X) ? (double *)calloc(NZ(N * d_in), sizeof(double)) : NULL;
  if ((dW && !aW) || (dX && X && !aX)) { free(aW); free(aX); return ORACLE_ENOMEM; }
  int rc = ORACLE_OK;
  const int nt = oracle_threads(M);
  /* the one loop of the one-thread form, walked by every thread: a thread adds to dX rows it owns (source row o) and to dW
   * entries it owns (relation p; featureless: table row o) -- two owners per message, each accumulator filled in edge order */
#ifdef _OPENMP
#pragma omp parallel num_threads(nt)
#endif
  {
#ifdef _OPENMP
    const int me = omp_get_thread_num();
#else
    const int me = 0;
#endif
    for (int64_t e = 0; e < M; ++e) {
      int64_t s = Tp[3 * e + 0], p = Tp[3 * e + 1], o = Tp[3 * e + 2];
      if (s < 0 || s >= N || o < 0 || o >= N || p < 0 || p >= R) { rc = ORACLE_ERANGE; break; }   /* (every thread meets it) */
      double v = (double)val[e];
      const float *gs = g + s * d_out;
      if (!X) {
        if (aW && (int)(o % nt) == me) {
          double *w = aW + (p * N + o) * d_out;
          for (int64_t j = 0; j < d_out; ++j) w[j] += v * (double)gs[j];
        }
        continue;
      }
      const int own_x = aX && (int)(o % nt) == me, own_w = aW && (int)(p % nt) == me;
      if (!own_x && !own_w) continue;
      const float *x = X + o * d_in;
      const float *w = W + p * d_in * d_out;
      for (int64_t i = 0; i < d_in; ++i) {
        const float *wr = w + i * d_out;
        double xv = v * (double)x[i];
        double dot = 0.0;
        for (int64_t j = 0; j < d_out; ++j) {
          dot += (double)gs[j] * (double)wr[j];
          if (own_w) aW[p * d_in * d_out + i * d_out + j] += xv * (double)gs[j];
        }
        if (own_x) aX[o * d_in + i] += v * dot;
      }
    }
  }
  if (rc == ORACLE_OK) {
    if (aW) for (int64_t t = 0; t < wsz; ++t) dW[t] = (float)aW[t];
    if (aX) {
#ifdef _OPENMP
#pragma omp parallel for num_threads(nt) schedule(static)
#endif
      for (int64_t t = 0; t < N * d_in; ++t) dX[t] = (float)aX[t];
    }
    if (db) {                    /* one pass over g, every column's sum still in row order (was: one strided pass per column) */
      double *a = (double *)calloc(NZ(d_out), sizeof(double));
      if (!a) rc = ORACLE_ENOMEM;
      else {
        for (int64_t n = 0; n < N; ++n)
          for (int64_t j = 0; j < d_out; ++j) a[j] += (double)g[n * d_out + j];
        for (int64_t j = 0; j < d_out; ++j) db[j] = (float)a[j];
        free(a);
      }
    }
  }
  free(aW);
  free(aX);
  return rc;
}

/* ---- DistMult decoder (next-row f-1) ------------------------------------- */

/* score_t = sum_k nodes[s_t,k] * rel[p_t,k] * nodes[o_t,k] (+ sb[s]+pb[p]+ob[o]).
 * triples: T rows of (s,p,o); works for the flattened 3-D case too. */
int oracle_distmult_forward(const int64_t *tr, int64_t T, int64_t N, int64_t R,
                            int64_t d, const float *nodes, const float *rel,
                            const float *sb, const float *pb, const float *ob,
                            float *scores) {
  for (int64_t t = 0; t < T; ++t) {
    int64_t s = tr[3 * t], p = tr[3 * t + 1], o = tr[3 * t + 2];
    if (s < 0 || s >= N || o < 0 || o >= N || p < 0 || p >= R) return ORACLE_ERANGE;
    double a = 0.0;
    for (int64_t k = 0; k < d; ++k)
      a += (double)nodes[s * d + k] * (double)rel[p * d + k] * (double)nodes[o * d + k];
    if (sb) a += (double)sb[s] + (double)pb[p] + (double)ob[o];
    scores[t] = (float)a;
  }
  return ORACLE_OK;
}

/* Gradients of sum_t gs_t * score_t w.r.t. nodes, rel and the three biases. */
int oracle_distmult_backward(const int64_t *tr, int64_t T, int64_t N, int64_t R,
                             int64_t d, const float *nodes, const float *rel,
                             const float *gs, float *dnodes, float *drel,
                             float *dsb, float *dpb, float *dob) {
  double *an = (double *)calloc(NZ(N * d), sizeof(double));
  double *ar = (double *)calloc(NZ(R * d), sizeof(double));
  if (!an || !ar) { free(an); free(ar); return ORACLE_ENOMEM; }
  if (dsb) { memset(dsb, 0, sizeof(float) * (size_t)N); memset(dob, 0, sizeof(float) * (size_t)N); memset(dpb, 0, sizeof(float) * (size_t)R); }
  for (int64_t t = 0; t < T; ++t) {
    int64_t s = tr[3 * t], p = tr[3 * t + 1], o = tr[3 * t + 2];
    double gv = (double)gs[t];
    for (int64_t k = 0; k < d; ++k) {
      double ns = nodes[s * d + k], rp = rel[p * d + k], no = nodes[o * d + k];
      an[s * d + k] += gv * rp * no;
      an[o * d + k] += gv * rp * ns;
      ar[p * d + k] += gv * ns * no;
    }
    if (dsb) { dsb[s] += (float)gv; dpb[p] += (float)gv; dob[o] += (float)gv; } /* few terms per slot */
  }
  for (int64_t t = 0; t < N * d; ++t) dnodes[t] = (float)an[t];
  for (int64_t t = 0; t < R * d; ++t) drel[t] = (float)ar[t];
  free(an);
  free(ar);
  return ORACLE_OK;
}
