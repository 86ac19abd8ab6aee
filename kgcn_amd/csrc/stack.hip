// Cross-layer kernels for the small-graph regime: the WHOLE node-level body of example_model/model.py:42-54
//     GraphConv -> act -> GraphConv -> act -> GraphConv -> [BatchNormalization (inference statistics) -> act] ->
//     GraphDense -> act -> GraphGather
// in ONE forward and ONE backward launch.
//
// Why: at the reference's own batch sizes (example_config/synth.json: 30 graphs of 10 nodes; 4,096 at BASELINE config 4)
// a step of that network is ~45 launches of kernels that each move a few KB..MB -- launch-latency-bound even as a
// hipGraph replay (0.22 ms at 30 graphs, 0.32 ms at 4,096, profiles/r03_train_bench.json): every layer is a launch and an
// HBM round trip although graphs are independent across LAYERS too.  For N <= 32 nodes and widths <= 64 a graph's
// activations (N x 64 floats = 8 KB) and all weights (<= 4 x 64 x 64 floats) fit LDS, so a workgroup walks the layer list
// for its graphs with the activations in LDS and the weights staged once.  HBM traffic = inputs + the per-layer outputs the
// backward needs (written once) + outputs; the arithmetic (<= 0.8 MFLOP per graph) runs as plain fp32 FMAs -- no MFMA: at
// these sizes the kernel is bound by launch / latency, and plain fp32 keeps the results within one rounding of the
// layer-by-layer kernels.
//
// Layer kinds (struct kgcn_stack_layer):  0  H <- act(A (H W + b))      GraphConv, one channel (kgcn/layers.py:105-116)
//                                         1  H <- act(H W + b)          GraphDense (:255-262)
//                                         2  H <- act(gamma (H - mean) / sqrt(var + eps) + beta) on the rows < enabled[t],
//                                                 act(0) on the others   GraphBatchNormalization, moving statistics (:196-216)
// followed by an optional GraphGather (sum over ALL N rows, quirk Q4).
//
// Workgroup = 4 waves; thread (wave w, lane j) owns rows r = w, w+4, ... of column j.  Backward: dW / dbias / dgamma / dbeta
// accumulate over the workgroup's graphs in LDS slots owned by exactly one thread (no atomics), one partial per workgroup,
// reduce_partials adds the partials in a fixed order: deterministic.
#include "stack_common.h"

namespace kgcn {

int launch_reduce_partials(const float* part, int nparts, long n, float* out, hipStream_t s);
int reduce_or_defer(const float* part, int nparts, long n, float* out, hipStream_t s);   // (dense.hip: queued inside a deferral scope)

// LDS floats of one weight block: kind 0/1: W [din x 65] + b [64]; kind 2: scale [64] + shift [64]
__host__ __device__ inline int sk_wblock(int kind, int din) { return kind == 2 ? 128 : din * SK_WLD + 64; }

__device__ __forceinline__ void sk_stage_weights(const StackArgs& a, float* wl) {
  for (int l = 0; l < a.nl; ++l) {
    float* blk = wl + a.woff[l];
    if (a.kind[l] == 2) {
      for (int j = threadIdx.x; j < 64; j += 256) {
        float sc = 0.f, sh = 0.f;
        if (j < a.dout[l]) {
          const float rs = 1.0f / __builtin_sqrtf(a.var[l][j] + a.eps[l]);
          sc = a.w[l][j] * rs;                               // gamma * rstd
          sh = a.b[l][j] - a.mean[l][j] * sc;                // beta - mean * gamma * rstd
        }
        blk[j] = sc;
        blk[64 + j] = sh;
      }
    } else {
      const int din = a.din[l], dout = a.dout[l];
      for (int i = threadIdx.x; i < din * 64; i += 256) {
        const int k = i >> 6, j = i & 63;
        blk[k * SK_WLD + j] = j < dout ? a.w[l][(long)k * dout + j] : 0.f;
      }
      for (int j = threadIdx.x; j < 64; j += 256) blk[din * SK_WLD + j] = (j < dout && a.b[l]) ? a.b[l][j] : 0.f;
    }
  }
}

__device__ __forceinline__ void sk_load_tile(const float* __restrict__ src, int N, int d, float* tile) {
  for (int i = threadIdx.x; i < N * 64; i += 256) {
    const int r = i >> 6, j = i & 63;
    tile[i] = j < d ? src[(long)r * d + j] : 0.f;
  }
}

__device__ __forceinline__ void sk_load_csr(const int* __restrict__ rowptr, const int2* __restrict__ cv, long t, int N,
                                            int max_nnz, int* rp, int2* ent) {
  const int* g = rowptr + t * N;
  const int base = g[0];
  for (int i = threadIdx.x; i <= N; i += 256) rp[i] = g[i] - base;
  int cnt = g[N] - base;
  if (cnt > max_nnz) cnt = max_nnz;
  for (int i = threadIdx.x; i < cnt; i += 256) ent[i] = cv[base + i];
}

// acc[q] = sum_k Hin[r_q, k] W[k, j] + b[j] for the wave's rows r_q = w + 4 q
template <int NQ>
__device__ __forceinline__ void sk_matmul(const float* Hin, const float* W, int din, int N, int w, int j, float (&acc)[NQ]) {
  const float bj = W[din * SK_WLD + j];
  const float* hr[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    acc[q] = bj;
    const int r = w + 4 * q;
    hr[q] = Hin + (r < N ? r : 0) * SK_LD;             // rows past N read row 0 (their results are never stored)
  }
  int k = 0;
  for (; k + 3 < din; k += 4) {                        // four k-values per trip: one 16-byte broadcast read per row
    const float w0 = W[k * SK_WLD + j], w1 = W[(k + 1) * SK_WLD + j], w2 = W[(k + 2) * SK_WLD + j], w3 = W[(k + 3) * SK_WLD + j];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const f32x4 h = *reinterpret_cast<const f32x4*>(hr[q] + k);
      acc[q] = __builtin_fmaf(h[0], w0, acc[q]);
      acc[q] = __builtin_fmaf(h[1], w1, acc[q]);
      acc[q] = __builtin_fmaf(h[2], w2, acc[q]);
      acc[q] = __builtin_fmaf(h[3], w3, acc[q]);
    }
  }
  for (; k < din; ++k) {
    const float wv = W[k * SK_WLD + j];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = __builtin_fmaf(hr[q][k], wv, acc[q]);
  }
}

// out[q] = sum over the stored entries of row r_q of val * T[col, j]
template <int NQ>
__device__ __forceinline__ void sk_aggregate(const float* T, const int* rp, const int2* ent, int N, int w, int j,
                                             float (&out)[NQ]) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int r = w + 4 * q;
    float s = 0.f;
    if (r < N) {
      for (int e = rp[r]; e < rp[r + 1]; ++e) {
        const int2 p = ent[e];
        s = __builtin_fmaf(__int_as_float(p.y), T[p.x * SK_LD + j], s);
      }
    }
    out[q] = s;
  }
}

template <int NQ>
__global__ __launch_bounds__(256) void stack_fwd_kernel(StackArgs a, const int* __restrict__ rowptr, const int2* __restrict__ cv,
                                                        const float* __restrict__ x, const int* __restrict__ enabled, long T,
                                                        float* __restrict__ pooled) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* wl = sm;
  const int tile = a.N * SK_LD;
  float* Ha = wl + a.wtotal;
  float* Hb = Ha + tile;
  float* Tt = Hb + tile;
  float* red = Tt + tile;                             // [4][64]
  int* rp = reinterpret_cast<int*>(red + 4 * 64);
  int2* ent = reinterpret_cast<int2*>(rp + 36);
  const int N = a.N, w = threadIdx.x >> 6, j = threadIdx.x & 63;
  sk_stage_weights(a, wl);
  for (long t = blockIdx.x; t < T; t += gridDim.x) {
    __syncthreads();                                   // previous graph's readers are done (and the weights are staged)
    sk_load_tile(x + t * N * a.din[0], N, a.din[0], Ha);
    sk_load_csr(rowptr, cv, t, N, a.max_nnz, rp, ent);
    __syncthreads();
    float* Hin = Ha;
    float* Hout = Hb;
    const int nvalid = enabled ? enabled[t] : N;
    for (int l = 0; l < a.nl; ++l) {
      const float* W = wl + a.woff[l];
      const int dout = a.dout[l], act = a.act[l];
      float y[NQ];
      if (a.kind[l] == 2) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int r = w + 4 * q;
          y[q] = (r < N && r < nvalid) ? __builtin_fmaf(Hin[r * SK_LD + j], W[j], W[64 + j]) : 0.f;
        }
      } else {
        sk_matmul(Hin, W, a.din[l], N, w, j, y);
        if (a.kind[l] == 0) {
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const int r = w + 4 * q;
            if (r < N) Tt[r * SK_LD + j] = y[q];
          }
          __syncthreads();
          sk_aggregate(Tt, rp, ent, N, w, j, y);
        }
      }
      float* og = a.out[l] + t * N * dout;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int r = w + 4 * q;
        if (r < N) {
          const float v = j < dout ? act_fwd(y[q], act) : 0.f;
          Hout[r * SK_LD + j] = v;
          if (j < dout) og[(long)r * dout + j] = v;
        }
      }
      __syncthreads();                                 // Hout complete; Tt / Hin free
      float* tmp = Hin; Hin = Hout; Hout = tmp;
    }
    if (a.gather) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int r = w + 4 * q;
        if (r < N) s += Hin[r * SK_LD + j];
      }
      red[w * 64 + j] = s;
      __syncthreads();
      const int dl = a.dout[a.nl - 1];
      if (w == 0 && j < dl) pooled[t * dl + j] = (red[j] + red[64 + j]) + (red[128 + j] + red[192 + j]);
    }
  }
}

// Backward.  LDS: weights | gradient accumulators (same layout as the weights: W slots -> dW, b slots -> 4 wave-private
// rows of dbias folded at the end ... see below) | tiles.
//   dacc layout per layer: kind 0/1: dW [din x 65] then db [4 x 64] (one row per wave);  kind 2: dgamma [4 x 64], dbeta [4 x 64]
__host__ __device__ inline int sk_gblock(int kind, int din) { return kind == 2 ? 512 : din * SK_WLD + 256; }

template <int NQ>
__global__ __launch_bounds__(256) void stack_bwd_kernel(StackArgs a, const int* __restrict__ rowptr_t, const int2* __restrict__ cv_t,
                                                        const float* __restrict__ x, const int* __restrict__ enabled, long T,
                                                        const float* __restrict__ dlast, float* __restrict__ dx,
                                                        float* __restrict__ part, int gtotal) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* wl = sm;
  float* ga = wl + a.wtotal;                          // gradient accumulators, gtotal floats
  const int tile = a.N * SK_LD;
  float* Dy = ga + gtotal;                            // d(layer output) of the layer being processed
  float* Yt = Dy + tile;                              // saved layer output (for act') / d pre-activation
  float* Ht = Yt + tile;                              // saved layer input
  float* Tt = Ht + tile;                              // dT = A^T d pre-activation
  int* rp = reinterpret_cast<int*>(Tt + tile);
  int2* ent = reinterpret_cast<int2*>(rp + 36);
  const int N = a.N, w = threadIdx.x >> 6, j = threadIdx.x & 63;
  sk_stage_weights(a, wl);
  for (int i = threadIdx.x; i < gtotal; i += 256) ga[i] = 0.f;
  for (long t = blockIdx.x; t < T; t += gridDim.x) {
    __syncthreads();
    const int dl = a.dout[a.nl - 1];
    for (int i = threadIdx.x; i < N * 64; i += 256) {
      const int r = i >> 6, c = i & 63;
      Dy[i] = c < dl ? (a.gather ? dlast[t * dl + c] : dlast[(t * N + r) * dl + c]) : 0.f;
    }
    sk_load_csr(rowptr_t, cv_t, t, N, a.max_nnz, rp, ent);
    const int nvalid = enabled ? enabled[t] : N;
    // the saved output of the LAST layer; every other saved tensor is loaded once, as the input of the layer above it, and
    // then serves as the output of the layer below (Yt / Ht swap roles)
    sk_load_tile(a.out[a.nl - 1] + t * N * dl, N, dl, Yt);
    for (int l = a.nl - 1; l >= 0; --l) {
      const int din = a.din[l], dout = a.dout[l], act = a.act[l], kind = a.kind[l];
      const float* W = wl + a.woff[l];
      float* G = ga + a.goff[l];
      // this layer's input tile: requested now (registers), landed in LDS behind the first barrier
      const float* hsrc = l == 0 ? x + t * N * a.din[0] : a.out[l - 1] + t * N * a.dout[l - 1];
      const int hd = l == 0 ? a.din[0] : a.dout[l - 1];
      float hp[8];                                     // N <= 32: at most 8 tile elements per thread
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = threadIdx.x + 256 * q;
        const int r = i >> 6, c = i & 63;
        hp[q] = (i < N * 64 && c < hd) ? hsrc[(long)r * hd + c] : 0.f;
      }
      __syncthreads();                                 // Dy and Yt of this layer complete; tiles of the previous iteration free
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = threadIdx.x + 256 * q;
        if (i < N * 64) Ht[i] = hp[q];
      }
      __syncthreads();
      if (kind == 2) {
        // y = act(sc * x + sh) on valid rows; d sc x-term: dgamma = sum dpre * xhat, xhat = (x - mean) rstd = (sc x + sh - beta) / gamma
        // is avoided: accumulate S1 = sum dpre * x and S0 = sum dpre, the host-free conversion happens in the finishing step
        float s1 = 0.f, s0 = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int r = w + 4 * q;
          if (r < N) {
            float d = 0.f;
            if (r < nvalid && j < dout) {
              d = Dy[r * SK_LD + j] * act_dout(Yt[r * SK_LD + j], act);
              s1 = __builtin_fmaf(d, Ht[r * SK_LD + j], s1);
              s0 += d;
            }
            Tt[r * SK_LD + j] = d * W[j];              // dx = dpre * gamma * rstd (0 on padded rows)
          }
        }
        G[w * 64 + j] += s1;
        G[256 + w * 64 + j] += s0;
        __syncthreads();
        for (int i = threadIdx.x; i < N * 64; i += 256) Dy[i] = Tt[i];
        { float* tmp = Yt; Yt = Ht; Ht = tmp; }        // this layer's input is the output of the layer below
        continue;
      }
      // d pre-activation into Yt (in place)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int r = w + 4 * q;
        if (r < N) Yt[r * SK_LD + j] = j < dout ? Dy[r * SK_LD + j] * act_dout(Yt[r * SK_LD + j], act) : 0.f;
      }
      __syncthreads();
      float dt[NQ];
      if (kind == 0) {
        sk_aggregate(Yt, rp, ent, N, w, j, dt);        // A^T d pre-activation
      } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int r = w + 4 * q;
          dt[q] = r < N ? Yt[r * SK_LD + j] : 0.f;
        }
      }
      float bs = 0.f;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int r = w + 4 * q;
        if (r < N) { Tt[r * SK_LD + j] = dt[q]; bs += dt[q]; }
      }
      G[din * SK_WLD + w * 64 + j] += bs;              // dbias, wave-private row
      __syncthreads();
      // dW[k, j] += sum_i Hin[i, k] dT[i, j] for k = w, w+4, ...: slot (k, j) is owned by this thread alone
      for (int k = w; k < din; k += 4) {
        float s = 0.f;
        for (int i = 0; i < N; ++i) s = __builtin_fmaf(Ht[i * SK_LD + k], Tt[i * SK_LD + j], s);
        G[k * SK_WLD + j] += s;
      }
      // d input[r, k'] = sum_j dT[r, j] W[k', j]  (lane = k'), written over Dy for the next (earlier) layer
      if (l > 0 || dx) {
        float di[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) di[q] = 0.f;
        if (j < din) {
          for (int c = 0; c < dout; ++c) {
            const float wv = W[j * SK_WLD + c];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
              const int r = w + 4 * q;
              if (r < N) di[q] = __builtin_fmaf(Tt[r * SK_LD + c], wv, di[q]);
            }
          }
        }
        // Dy is only read at the top of an iteration (all threads passed two barriers since): safe to overwrite now
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int r = w + 4 * q;
          if (r < N) Dy[r * SK_LD + j] = di[q];
        }
      }
      { float* tmp = Yt; Yt = Ht; Ht = tmp; }          // this layer's input is the output of the layer below
    }
    if (dx) {
      __syncthreads();
      const int d0 = a.din[0];
      float* o = dx + t * N * d0;
      for (int i = threadIdx.x; i < N * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        if (c < d0) o[(long)r * d0 + c] = Dy[i];
      }
    }
  }
  __syncthreads();
  // this workgroup's partial in the flat parameter-gradient layout: per layer dW [din, dout] | db [dout]  (kind 2: dgamma | dbeta)
  float* pp = part + (long)blockIdx.x * a.ptotal;
  for (int l = 0; l < a.nl; ++l) {
    const float* G = ga + a.goff[l];
    float* o = pp + a.poff[l];
    const int din = a.din[l], dout = a.dout[l];
    if (a.kind[l] == 2) {
      // dgamma = rstd * (S1 - mean * S0), dbeta = S0  with S1 = sum dpre x, S0 = sum dpre
      for (int c = threadIdx.x; c < dout; c += 256) {
        const float s1 = (G[c] + G[64 + c]) + (G[128 + c] + G[192 + c]);
        const float s0 = (G[256 + c] + G[320 + c]) + (G[384 + c] + G[448 + c]);
        const float rs = 1.0f / __builtin_sqrtf(a.var[l][c] + a.eps[l]);
        o[c] = rs * (s1 - a.mean[l][c] * s0);
        o[dout + c] = s0;
      }
    } else {
      for (int i = threadIdx.x; i < din * dout; i += 256) {
        const int k = i / dout, c = i - k * dout;
        o[i] = G[k * SK_WLD + c];
      }
      const float* B = G + din * SK_WLD;
      for (int c = threadIdx.x; c < dout; c += 256) o[din * dout + c] = (B[c] + B[64 + c]) + (B[128 + c] + B[192 + c]);
    }
  }
}

struct StackPlan { StackArgs a; int gtotal; size_t lds_fwd, lds_bwd; Stack2Plan tile; int route; };

// Route of a launch: layers[0].route 1 / 2 force the one-graph / the tile kernels (tests, measurements); 0: the tile kernels
// whenever they take the layer list (model.py step, 10-node graphs: 30 graphs 0.099 vs 0.118 ms, 512 graphs 0.103 vs 0.169 ms,
// 4,096 graphs 0.208 vs 0.867 ms), the one-graph kernels for what they do not (more than four weight matrices, LDS).
static bool use_tiles(const StackPlan& p, long T) {
  if (!p.tile.ok || p.route == 1) return false;
  if (p.route == 2) return true;
  if (p.lds_fwd > (size_t)kLdsBytes || p.lds_bwd > (size_t)kLdsBytes) return true;     // the one-graph kernels do not fit
  return true;
}

static int stack_plan(int n_nodes, int max_nnz, const kgcn_stack_layer* layers, int nl, StackPlan* p, const char* who) {
  if (nl <= 0 || nl > SK_MAXL) return fail("%s: %d layers (1..%d)", who, nl, SK_MAXL);
  if (n_nodes <= 0 || n_nodes > 32) return fail("%s: %d nodes per graph (1..32)", who, n_nodes);
  if (max_nnz < 0 || max_nnz > 2048) return fail("%s: %d stored entries per graph (<= 2048)", who, max_nnz);
  StackArgs& a = p->a;
  a.nl = nl; a.N = n_nodes; a.max_nnz = max_nnz > 0 ? max_nnz : 1;
  int wo = 0, po = 0, go = 0;
  for (int l = 0; l < nl; ++l) {
    const kgcn_stack_layer& L = layers[l];
    if (L.kind < 0 || L.kind > 2) return fail("%s: layer %d: unknown kind %d", who, l, L.kind);
    if (L.act < KGCN_ACT_NONE || L.act > KGCN_ACT_TANH) return fail("%s: layer %d: unknown activation %d", who, l, L.act);
    if (L.din <= 0 || L.din > 64 || L.dout <= 0 || L.dout > 64) return fail("%s: layer %d: widths %d -> %d (1..64)", who, l, L.din, L.dout);
    if (L.kind == 2 && L.din != L.dout) return fail("%s: layer %d: a normalisation keeps the width", who, l);
    if (l > 0 && L.din != layers[l - 1].dout) return fail("%s: layer %d: input width %d != previous output %d", who, l, L.din, layers[l - 1].dout);
    if (!L.w || (L.kind == 2 && (!L.b || !L.mean || !L.var))) return fail("%s: layer %d: NULL parameter", who, l);
    a.kind[l] = L.kind; a.act[l] = L.act; a.din[l] = L.din; a.dout[l] = L.dout;
    a.w[l] = L.w; a.b[l] = L.b; a.mean[l] = L.mean; a.var[l] = L.var; a.eps[l] = L.eps;
    a.woff[l] = wo; a.poff[l] = po; a.goff[l] = go;
    wo += (sk_wblock(L.kind, L.din) + 3) & ~3;          // blocks start 16-byte aligned (the CSR entries behind them are int2)
    go += (sk_gblock(L.kind, L.din) + 3) & ~3;
    po += L.kind == 2 ? 2 * L.dout : L.din * L.dout + L.dout;
  }
  a.wtotal = wo; a.ptotal = po; p->gtotal = go;
  const size_t csr = (size_t)36 * 4 + (size_t)a.max_nnz * 8;
  p->lds_fwd = ((size_t)wo + 3 * (size_t)n_nodes * SK_LD + 4 * 64) * 4 + csr;
  p->lds_bwd = ((size_t)wo + (size_t)go + 4 * (size_t)n_nodes * SK_LD) * 4 + csr;
  p->route = layers[0].route;
  if (p->route < 0 || p->route > 2) return fail("%s: route %d of the first layer (0 automatic, 1 one graph per trip, 2 tiles)", who, p->route);
  p->tile = stack2_plan(a);
  if (p->route == 2 && !p->tile.ok) return fail("%s: the tile kernels do not take this layer list", who);
  return 0;
}

}  // namespace kgcn

using namespace kgcn;

extern "C" int kgcn_gcn_stack_supported(int32_t n_nodes, int32_t max_nnz_per_graph, const kgcn_stack_layer* layers,
                                        int32_t num_layers) {
  StackPlan p;
  if (!layers || stack_plan(n_nodes, max_nnz_per_graph, layers, num_layers, &p, "kgcn_gcn_stack_supported")) return 0;
  if (p.route != 1 && p.tile.ok) return 1;
  return p.lds_fwd <= (size_t)kLdsBytes && p.lds_bwd <= (size_t)kLdsBytes ? 1 : 0;
}

extern "C" int64_t kgcn_gcn_stack_param_floats(const kgcn_stack_layer* layers, int32_t num_layers) {
  int64_t n = 0;
  for (int l = 0; layers && l < num_layers; ++l)
    n += layers[l].kind == 2 ? 2 * (int64_t)layers[l].dout : (int64_t)layers[l].din * layers[l].dout + layers[l].dout;
  return n;
}

static int stack_blocks(long T, size_t lds) {
  int per_cu = (int)((size_t)kLdsBytes / (lds ? lds : 1));
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 4) per_cu = 4;
  long b = (long)kNumCU * per_cu;
  if (b > T) b = T;
  return (int)(b < 1 ? 1 : b);
}

extern "C" int kgcn_gcn_stack_fwd_f32(const kgcn_csr_batch* a, const float* x, const int32_t* enabled,
                                      const kgcn_stack_layer* layers, int32_t num_layers, float* const* layer_out,
                                      float* pooled, void* stream) {
  if (int rc = validate_csr(a, "kgcn_gcn_stack_fwd_f32")) return rc;
  if (a->rows != a->cols) return fail("kgcn_gcn_stack_fwd_f32: adjacency must be square");
  if (!layers || !layer_out) return fail("kgcn_gcn_stack_fwd_f32: NULL layer list");
  StackPlan p;
  if (int rc = stack_plan(a->rows, a->max_nnz_per_graph, layers, num_layers, &p, "kgcn_gcn_stack_fwd_f32")) return rc;
  const bool tiles = use_tiles(p, a->num_graphs);
  if (!tiles && p.lds_fwd > (size_t)kLdsBytes) return fail("kgcn_gcn_stack_fwd_f32: %zu bytes of LDS needed", p.lds_fwd);
  if (a->num_graphs == 0) return 0;
  if (!x) return fail("kgcn_gcn_stack_fwd_f32: x is NULL");
  for (int l = 0; l < num_layers; ++l) {
    if (!layer_out[l]) return fail("kgcn_gcn_stack_fwd_f32: layer_out[%d] is NULL", l);
    p.a.out[l] = layer_out[l];
  }
  p.a.gather = pooled ? 1 : 0;
  if (tiles)
    return launch_stack2_fwd(p.a, p.tile, a->rowptr, reinterpret_cast<const int2*>(a->cv), x, enabled, (long)a->num_graphs,
                             pooled, as_stream(stream));
  const dim3 grid(stack_blocks(a->num_graphs, p.lds_fwd));
  const int2* cvp = reinterpret_cast<const int2*>(a->cv);
#define KGCN_SK_FWD(NQ)                                                                                               \
  {                                                                                                                   \
    static thread_local bool attr = false;                                                                            \
    if (!attr) {                                                                                                      \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stack_fwd_kernel<NQ>),                                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);                              \
      attr = true;                                                                                                    \
    }                                                                                                                 \
    hipLaunchKernelGGL(stack_fwd_kernel<NQ>, grid, dim3(256), p.lds_fwd, as_stream(stream), p.a, a->rowptr, cvp, x,   \
                       enabled, (long)a->num_graphs, pooled);                                                         \
  }
  switch ((a->rows + 3) / 4) {
    case 1: KGCN_SK_FWD(1) break;
    case 2: KGCN_SK_FWD(2) break;
    case 3: KGCN_SK_FWD(3) break;
    case 4: KGCN_SK_FWD(4) break;
    case 5: KGCN_SK_FWD(5) break;
    case 6: KGCN_SK_FWD(6) break;
    case 7: KGCN_SK_FWD(7) break;
    default: KGCN_SK_FWD(8) break;
  }
#undef KGCN_SK_FWD
  return check_launch("stack_fwd_kernel");
}

extern "C" int64_t kgcn_gcn_stack_bwd_workspace_bytes(int32_t num_graphs, const kgcn_stack_layer* layers, int32_t num_layers) {
  const int64_t n = kgcn_gcn_stack_param_floats(layers, num_layers);
  long blocks = (long)kNumCU * 4;
  if (blocks > num_graphs) blocks = num_graphs > 0 ? num_graphs : 1;
  return blocks * n * 4;
}

extern "C" int kgcn_gcn_stack_bwd_f32(const kgcn_csr_batch* at, const float* x, const int32_t* enabled,
                                      const kgcn_stack_layer* layers, int32_t num_layers, float* const* layer_out,
                                      const float* dlast, int32_t gather, float* dx, float* dparams, void* workspace,
                                      int64_t workspace_bytes, void* stream) {
  if (int rc = validate_csr(at, "kgcn_gcn_stack_bwd_f32")) return rc;
  if (at->rows != at->cols) return fail("kgcn_gcn_stack_bwd_f32: adjacency must be square");
  if (!layers || !layer_out) return fail("kgcn_gcn_stack_bwd_f32: NULL layer list");
  StackPlan p;
  if (int rc = stack_plan(at->rows, at->max_nnz_per_graph, layers, num_layers, &p, "kgcn_gcn_stack_bwd_f32")) return rc;
  const bool tiles = use_tiles(p, at->num_graphs);
  if (!tiles && p.lds_bwd > (size_t)kLdsBytes) return fail("kgcn_gcn_stack_bwd_f32: %zu bytes of LDS needed", p.lds_bwd);
  if (!dparams) return fail("kgcn_gcn_stack_bwd_f32: dparams is NULL");
  hipStream_t s = as_stream(stream);
  if (at->num_graphs == 0) {
    hipError_t e = hipMemsetAsync(dparams, 0, (size_t)p.a.ptotal * 4, s);
    return e == hipSuccess ? 0 : fail("kgcn_gcn_stack_bwd_f32: memset failed");
  }
  if (!x || !dlast) return fail("kgcn_gcn_stack_bwd_f32: NULL operand");
  for (int l = 0; l < num_layers; ++l) {
    if (!layer_out[l]) return fail("kgcn_gcn_stack_bwd_f32: layer_out[%d] is NULL", l);
    p.a.out[l] = layer_out[l];
  }
  p.a.gather = gather ? 1 : 0;
  const int blocks = tiles ? stack2_blocks(at->num_graphs, p.a.G) : stack_blocks(at->num_graphs, p.lds_bwd);
  const int64_t need = (int64_t)blocks * p.a.ptotal * 4;
  if (!workspace || workspace_bytes < need)
    return fail("kgcn_gcn_stack_bwd_f32: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
  float* part = static_cast<float*>(workspace);
  const int2* cvp = reinterpret_cast<const int2*>(at->cv);
  if (tiles) {
    if (int rc = launch_stack2_bwd(p.a, p.tile, at->rowptr, cvp, x, enabled, (long)at->num_graphs, dlast, dx, part, blocks, s))
      return rc;
    return reduce_or_defer(part, blocks, p.a.ptotal, dparams, s);
  }
#define KGCN_SK_BWD(NQ)                                                                                               \
  {                                                                                                                   \
    static thread_local bool attr = false;                                                                            \
    if (!attr) {                                                                                                      \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stack_bwd_kernel<NQ>),                                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);                              \
      attr = true;                                                                                                    \
    }                                                                                                                 \
    hipLaunchKernelGGL(stack_bwd_kernel<NQ>, dim3(blocks), dim3(256), p.lds_bwd, s, p.a, at->rowptr, cvp, x, enabled, \
                       (long)at->num_graphs, dlast, dx, part, p.gtotal);                                              \
  }
  switch ((at->rows + 3) / 4) {
    case 1: KGCN_SK_BWD(1) break;
    case 2: KGCN_SK_BWD(2) break;
    case 3: KGCN_SK_BWD(3) break;
    case 4: KGCN_SK_BWD(4) break;
    case 5: KGCN_SK_BWD(5) break;
    case 6: KGCN_SK_BWD(6) break;
    case 7: KGCN_SK_BWD(7) break;
    default: KGCN_SK_BWD(8) break;
  }
#undef KGCN_SK_BWD
  if (int rc = check_launch("stack_bwd_kernel")) return rc;
  return reduce_or_defer(part, blocks, p.a.ptotal, dparams, s);
}
