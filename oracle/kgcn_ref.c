/*
 * kgcn_ref.c -- plain-C CPU restatement ("port") of the reference's GraphConv hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE: used by tests/ to cross-check the numpy oracle and by
 * bench.py's `cpu_baseline` leg (kind "port") to time the reference algorithm on the GPU box's
 * host cores.  The product path never links or calls it.
 *
 * PARITY STATUS: parity unpinned at the TensorFlow boundary (see oracle/kgcn_oracle.py): the
 * reference's arithmetic is TF 1.15 ops that cannot run here.  This file follows the op
 * structure of the reference's default branch, kgcn/layers.py:105-116 -- per graph b:
 *     fw = matmul(X[b], W) + bias                      (:112)
 *     el = sparse_tensor_dense_matmul(A[b], fw)        (:113)   COO, stored-entry order
 * and for the backward the gradient definitions of kgcn/bspmm_call.py:22-57 plus TF's
 * MatMul/Add gradients:
 *     dfw = A[b]^T g[b];  dW += X[b]^T dfw;  dbias += colsum(dfw);  dX[b] = dfw W^T
 * fp32 throughout, accumulation order as written.  Graphs are independent, so the loop over the
 * batch is an OpenMP parallel-for (the TF executor also runs the per-graph ops concurrently);
 * dW/dbias are reduced per thread, then over threads in thread order.
 *
 * Adjacency input = the reference's COO layout (kgcn/data_util.py:40-45): per graph t the entries
 * off[t]..off[t+1]-1 of idx[2*e] = row, idx[2*e+1] = col, val[e].
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static void dense_xw_bias(int n, int din, int dout, const float* x, const float* w,
                          const float* bias, float* fw) {
  for (int i = 0; i < n; ++i) {
    float* o = fw + (size_t)i * dout;
    for (int j = 0; j < dout; ++j) o[j] = 0.f;
    for (int k = 0; k < din; ++k) {
      const float a = x[(size_t)i * din + k];
      const float* wr = w + (size_t)k * dout;
      for (int j = 0; j < dout; ++j) o[j] += a * wr[j];
    }
    if (bias)
      for (int j = 0; j < dout; ++j) o[j] += bias[j];
  }
}

int kgcn_ref_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* out[t] (+)= A[t] @ (x[t] @ w + bias) */
void kgcn_ref_graphconv_fwd(int T, int n, int din, int dout, const int64_t* off, const int32_t* idx,
                            const float* val, const float* x, const float* w, const float* bias,
                            float* out, int accumulate, int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    float* fw = (float*)malloc((size_t)n * dout * sizeof(float));
#pragma omp for schedule(static)
    for (int t = 0; t < T; ++t) {
      dense_xw_bias(n, din, dout, x + (size_t)t * n * din, w, bias, fw);
      float* o = out + (size_t)t * n * dout;
      if (!accumulate) memset(o, 0, (size_t)n * dout * sizeof(float));
      for (int64_t e = off[t]; e < off[t + 1]; ++e) {
        const int r = idx[2 * e], c = idx[2 * e + 1];
        const float v = val[e];
        const float* src = fw + (size_t)c * dout;
        float* dst = o + (size_t)r * dout;
        for (int j = 0; j < dout; ++j) dst[j] += v * src[j];
      }
    }
    free(fw);
  }
}

/* dx[t] (+)= (A[t]^T g[t]) w^T ; dw += sum_t x[t]^T (A[t]^T g[t]) ; db += sum_t colsum(.)
 * dw [din*dout] and db [dout] are overwritten unless accumulate. dx may be NULL. */
void kgcn_ref_graphconv_bwd(int T, int n, int din, int dout, const int64_t* off, const int32_t* idx,
                            const float* val, const float* x, const float* w, const float* g,
                            float* dx, float* dw, float* db, int accumulate, int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
  const int nt = omp_get_max_threads();
#else
  const int nt = 1;
#endif
  const size_t wsz = (size_t)din * dout;
  float* part = (float*)calloc((size_t)nt * (wsz + dout), sizeof(float));
#pragma omp parallel
  {
#ifdef _OPENMP
    const int me = omp_get_thread_num();
#else
    const int me = 0;
#endif
    float* pw = part + (size_t)me * (wsz + dout);
    float* pb = pw + wsz;
    float* dfw = (float*)malloc((size_t)n * dout * sizeof(float));
#pragma omp for schedule(static)
    for (int t = 0; t < T; ++t) {
      const float* gt = g + (size_t)t * n * dout;
      const float* xt = x + (size_t)t * n * din;
      memset(dfw, 0, (size_t)n * dout * sizeof(float));
      for (int64_t e = off[t]; e < off[t + 1]; ++e) { /* adjoint_a=True: rows <-> cols */
        const int r = idx[2 * e], c = idx[2 * e + 1];
        const float v = val[e];
        const float* src = gt + (size_t)r * dout;
        float* dst = dfw + (size_t)c * dout;
        for (int j = 0; j < dout; ++j) dst[j] += v * src[j];
      }
      for (int i = 0; i < n; ++i) {
        const float* dr = dfw + (size_t)i * dout;
        for (int j = 0; j < dout; ++j) pb[j] += dr[j];
        for (int k = 0; k < din; ++k) {
          const float a = xt[(size_t)i * din + k];
          float* wr = pw + (size_t)k * dout;
          for (int j = 0; j < dout; ++j) wr[j] += a * dr[j];
        }
      }
      if (dx) {
        float* dxt = dx + (size_t)t * n * din;
        for (int i = 0; i < n; ++i) {
          const float* dr = dfw + (size_t)i * dout;
          for (int k = 0; k < din; ++k) {
            const float* wr = w + (size_t)k * dout;
            float s = 0.f;
            for (int j = 0; j < dout; ++j) s += dr[j] * wr[j];
            if (accumulate) dxt[(size_t)i * din + k] += s;
            else dxt[(size_t)i * din + k] = s;
          }
        }
      }
    }
    free(dfw);
  }
  if (!accumulate) {
    memset(dw, 0, wsz * sizeof(float));
    memset(db, 0, (size_t)dout * sizeof(float));
  }
  for (int p = 0; p < nt; ++p) {
    const float* pw = part + (size_t)p * (wsz + dout);
    for (size_t i = 0; i < wsz; ++i) dw[i] += pw[i];
    for (int j = 0; j < dout; ++j) db[j] += pw[wsz + j];
  }
  free(part);
}

/* plain batched SpMM (op Bspmm, kgcn/bspmm_call.py:16): out[t] = op(A[t]) @ rhs[t] */
void kgcn_ref_bspmm(int T, int m, int k, int d, const int64_t* off, const int32_t* idx,
                    const float* val, const float* rhs, float* out, int adjoint_a, int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  const int om = adjoint_a ? k : m, ok = adjoint_a ? m : k;
#pragma omp parallel for schedule(static)
  for (int t = 0; t < T; ++t) {
    float* o = out + (size_t)t * om * d;
    const float* r = rhs + (size_t)t * ok * d;
    memset(o, 0, (size_t)om * d * sizeof(float));
    for (int64_t e = off[t]; e < off[t + 1]; ++e) {
      const int row = adjoint_a ? idx[2 * e + 1] : idx[2 * e];
      const int col = adjoint_a ? idx[2 * e] : idx[2 * e + 1];
      const float v = val[e];
      for (int j = 0; j < d; ++j) o[(size_t)row * d + j] += v * r[(size_t)col * d + j];
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * Dense part at BASELINE sizes (round 3): Keras Dense on [m, din] rows (kgcn/layers.py:255-262 GraphDense; the
 * [B*N, Din] x [Din, Dout] contraction of GraphConv's batched branch :99-100) with the model files' activation, and its
 * TF gradients (MatMul / BiasAdd / activation grads).  fp32 products like the path under test; the SUMS over the m rows
 * (dW, dbias: 2e5 rows at config 4 / 5) accumulate in fp64 per thread so that the checker's own rounding stays far below
 * the 1e-5 it is used to certify.  act: 0 none, 1 sigmoid, 2 relu, 3 tanh (KGCN_ACT_* of include/kgcn_hip.h).
 * ------------------------------------------------------------------------------------------------ */
#include <math.h>

static float ref_act(float v, int act) {
  if (act == 1) return 1.0f / (1.0f + expf(-v));
  if (act == 2) return v > 0.f ? v : 0.f;
  if (act == 3) return tanhf(v);
  return v;
}
static float ref_dact(float a, int act) { /* derivative expressed in the activation OUTPUT */
  if (act == 1) return a * (1.0f - a);
  if (act == 2) return a > 0.f ? 1.0f : 0.f;
  if (act == 3) return 1.0f - a * a;
  return 1.0f;
}

/* y = act(x @ w + bias), x [m, din], w [din, dout], bias [dout] or NULL */
void kgcn_ref_dense_fwd(int64_t m, int din, int dout, const float* x, const float* w, const float* bias, int act,
                        float* y, int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < m; ++i) {
    float* o = y + (size_t)i * dout;
    for (int j = 0; j < dout; ++j) o[j] = 0.f;
    for (int k = 0; k < din; ++k) {
      const float a = x[(size_t)i * din + k];
      const float* wr = w + (size_t)k * dout;
      for (int j = 0; j < dout; ++j) o[j] += a * wr[j];
    }
    for (int j = 0; j < dout; ++j) o[j] = ref_act(o[j] + (bias ? bias[j] : 0.f), act);
  }
}

/* dpre = g * act'(y);  dx = dpre @ w^T (may be NULL);  dw = x^T dpre;  db = colsum(dpre)   (dw, db: fp64 out) */
void kgcn_ref_dense_bwd(int64_t m, int din, int dout, const float* x, const float* w, const float* y, const float* g,
                        int act, float* dx, double* dw, double* db, int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
  const int nt = omp_get_max_threads();
#else
  const int nt = 1;
#endif
  const size_t wsz = (size_t)din * dout;
  double* part = (double*)calloc((size_t)nt * (wsz + dout), sizeof(double));
#pragma omp parallel
  {
#ifdef _OPENMP
    const int me = omp_get_thread_num();
#else
    const int me = 0;
#endif
    double* pw = part + (size_t)me * (wsz + dout);
    double* pb = pw + wsz;
    float* dpre = (float*)malloc((size_t)dout * sizeof(float));
#pragma omp for schedule(static)
    for (int64_t i = 0; i < m; ++i) {
      const float* gr = g + (size_t)i * dout;
      for (int j = 0; j < dout; ++j) dpre[j] = gr[j] * (act ? ref_dact(y[(size_t)i * dout + j], act) : 1.0f);
      for (int j = 0; j < dout; ++j) pb[j] += dpre[j];
      for (int k = 0; k < din; ++k) {
        const float a = x[(size_t)i * din + k];
        double* wr = pw + (size_t)k * dout;
        for (int j = 0; j < dout; ++j) wr[j] += (double)(a * dpre[j]);
        if (dx) {
          const float* wk = w + (size_t)k * dout;
          float s = 0.f;
          for (int j = 0; j < dout; ++j) s += dpre[j] * wk[j];
          dx[(size_t)i * din + k] = s;
        }
      }
    }
    free(dpre);
  }
  for (size_t i = 0; i < wsz; ++i) dw[i] = 0.0;
  for (int j = 0; j < dout; ++j) db[j] = 0.0;
  for (int p = 0; p < nt; ++p) {
    const double* pw = part + (size_t)p * (wsz + dout);
    for (size_t i = 0; i < wsz; ++i) dw[i] += pw[i];
    for (int j = 0; j < dout; ++j) db[j] += pw[wsz + j];
  }
  free(part);
}

/* GINAggregate, one channel (kgcn/layers.py:461-472): out[t] = eps * x[t] + op(A[t]) @ x[t]; op = transpose for the
 * backward (d x = eps * g + A^T g).  *deps_out (may be NULL) = <x, g2> in fp64 when g2 is given (d eps = <g, x>, :469). */
void kgcn_ref_gin_aggregate(int T, int n, int d, const int64_t* off, const int32_t* idx, const float* val, const float* x,
                            float eps, int adjoint, float* out, const float* g2, double* deps_out, int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  double dot = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : dot)
  for (int t = 0; t < T; ++t) {
    const float* xt = x + (size_t)t * n * d;
    float* o = out + (size_t)t * n * d;
    for (size_t i = 0; i < (size_t)n * d; ++i) o[i] = eps * xt[i];
    for (int64_t e = off[t]; e < off[t + 1]; ++e) {
      const int row = adjoint ? idx[2 * e + 1] : idx[2 * e];
      const int col = adjoint ? idx[2 * e] : idx[2 * e + 1];
      const float v = val[e];
      for (int j = 0; j < d; ++j) o[(size_t)row * d + j] += v * xt[(size_t)col * d + j];
    }
    if (g2) {
      const float* gt = g2 + (size_t)t * n * d;
      for (size_t i = 0; i < (size_t)n * d; ++i) dot += (double)xt[i] * (double)gt[i];
    }
  }
  if (deps_out) *deps_out = dot;
}
