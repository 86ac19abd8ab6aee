// Dense layers with a very narrow side: the read-out Dense of the model files -- K.layers.Dense(info.label_dim) on the
// gathered [B, D] features (example_model/model.py:55: 50 -> 2; model_multitask.py:64: 50 -> 12; model_gin.py:62: 512 -> 2
// at width 256; sparse.py:96: 256 -> 10).  The panel kernels of dense.hip move 64-column blocks: for 2..16 columns they spend
// 20-45 us on tensors that take 5-10 us to stream (profiles/r03_c_cfg5_rocprof.txt: dense_wgrad_persist 39 us at 0.19 of HBM
// peak, dense_fwd_persist 32 us at 0.17).  Here a wave owns a row at a time:
//   skinny_n_fwd     y[m, n] = act(x[m, k] W[k, n] + b), n <= 16: lanes stride over k (coalesced row reads), n partial sums
//                    per lane, butterfly reduction, lane j writes column j
//   skinny_k_fwd     y[m, n] = x[m, k] op(W) (+ b), k <= 16 (the dX of such a layer: op(W) = W^T): lanes own output
//                    columns, their W entries live in registers, the k inputs of a row are wave-uniform
//   skinny_n_wgrad   dW[k, n] = x^T g, db = colsum g, n <= 16: lanes own k (their dW rows in registers across the wave's
//                    rows), one partial per workgroup, fixed-order second stage (reduce_partials)
#include "kgcn_common.h"

namespace kgcn {

int launch_reduce_pair(const float* part_dw, long n_dw, float* dw, const float* part_db, long n_db, float* dbias, int nparts,
                       hipStream_t s);

constexpr int SKN = 16;          // narrow side at most
constexpr int SK_KPL = 16;       // k values per lane at most (wide side <= 1024)

// N = template upper bound of the narrow width n (columns j >= n are never read or written)
template <int N>
__global__ __launch_bounds__(256) void skinny_n_fwd_kernel(const float* __restrict__ x, long m, int k, long x_ld,
                                                           const float* __restrict__ w, long w_ld, const float* __restrict__ bias,
                                                           float* __restrict__ y, long y_ld, int act, int n) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * 256) >> 6;
  for (long r = wave; r < m; r += nwaves) {
    float acc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = 0.f;
    const float* xr = x + r * x_ld;
    for (int c = lane; c < k; c += 64) {
      const float xv = xr[c];
      const float* wr = w + (long)c * w_ld;
#pragma unroll
      for (int j = 0; j < N; ++j) acc[j] = __builtin_fmaf(xv, j < n ? wr[j] : 0.f, acc[j]);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc[j] += __shfl_xor(acc[j], o, 64);
    }
    float out = 0.f;
#pragma unroll
    for (int j = 0; j < N; ++j) out = lane == j ? acc[j] : out;
    if (lane < n) y[r * y_ld + lane] = act_fwd(out + (bias ? bias[lane] : 0.f), act);
  }
}

// y[r, c] = sum_j x[r, j] Wop[j, c] (+ bias[c]);  trans: Wop[j, c] = w[c * w_ld + j], else w[j * w_ld + c];  K = k <= 16
template <int K>
__global__ __launch_bounds__(256) void skinny_k_fwd_kernel(const float* __restrict__ x, long m, long x_ld,
                                                           const float* __restrict__ w, long w_ld, int trans, int n,
                                                           const float* __restrict__ bias, float* __restrict__ y, long y_ld,
                                                           int act) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * 256) >> 6;
  for (int c0 = 0; c0 < n; c0 += 64 * 4) {              // four column groups per pass: 4 x K weights per lane in registers
    float wv[4][K], bv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c0 + 64 * q + lane;
      bv[q] = (bias && c < n) ? bias[c] : 0.f;
#pragma unroll
      for (int j = 0; j < K; ++j) wv[q][j] = c < n ? (trans ? w[(long)c * w_ld + j] : w[(long)j * w_ld + c]) : 0.f;
    }
    for (long r = wave; r < m; r += nwaves) {
      float xv[K];
#pragma unroll
      for (int j = 0; j < K; ++j) xv[j] = x[r * x_ld + j];            // wave-uniform addresses: one broadcast load each
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = c0 + 64 * q + lane;
        float s = bv[q];
#pragma unroll
        for (int j = 0; j < K; ++j) s = __builtin_fmaf(xv[j], wv[q][j], s);
        if (c < n) y[r * y_ld + c] = act_fwd(s, act);
      }
    }
  }
}

// part_dw[block][k, n] = sum over this workgroup's rows of x[r, k] g[r, n];  part_db[block][n] = sum g[r, n]
template <int N, int QK>
__global__ __launch_bounds__(256) void skinny_n_wgrad_kernel(const float* __restrict__ x, long m, int k, long x_ld,
                                                             const float* __restrict__ g, long g_ld,
                                                             float* __restrict__ part_dw, float* __restrict__ part_db, int n) {
  extern __shared__ float red[];                          // [4 waves][k * N + N]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * 256) >> 6;
  float acc[QK][N], bs[N];
#pragma unroll
  for (int q = 0; q < QK; ++q)
#pragma unroll
    for (int j = 0; j < N; ++j) acc[q][j] = 0.f;
#pragma unroll
  for (int j = 0; j < N; ++j) bs[j] = 0.f;
  // two rows per trip: their loads are independent (a wave is latency-bound on one row at a time)
  for (long r = wave; r < m; r += 2 * nwaves) {
    const long r2 = r + nwaves;
    const bool two = r2 < m;
    float gv[N], gw[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      gv[j] = j < n ? g[r * g_ld + j] : 0.f;                           // wave-uniform
      gw[j] = (two && j < n) ? g[r2 * g_ld + j] : 0.f;
    }
    const float* xr = x + r * x_ld;
    const float* xs = x + (two ? r2 : r) * x_ld;
    float xa[QK], xb[QK];
#pragma unroll
    for (int q = 0; q < QK; ++q) {
      const int c = lane + 64 * q;
      xa[q] = c < k ? xr[c] : 0.f;
      xb[q] = c < k ? xs[c] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < QK; ++q) {
#pragma unroll
      for (int j = 0; j < N; ++j) acc[q][j] = __builtin_fmaf(xb[q], gw[j], __builtin_fmaf(xa[q], gv[j], acc[q][j]));
    }
#pragma unroll
    for (int j = 0; j < N; ++j) bs[j] += gv[j] + gw[j];
  }
  float* mine = red + (long)wv * (k * n + n);
#pragma unroll
  for (int q = 0; q < QK; ++q) {
    const int c = lane + 64 * q;
    if (c < k) {
#pragma unroll
      for (int j = 0; j < N; ++j)
        if (j < n) mine[c * n + j] = acc[q][j];
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < N; ++j)
      if (j < n) mine[k * n + j] = bs[j];
  }
  __syncthreads();
  const int tot = k * n;
  for (int i = threadIdx.x; i < tot + n; i += 256) {
    const float s = (red[i] + red[(tot + n) + i]) + (red[2 * (tot + n) + i] + red[3 * (tot + n) + i]);
    if (i < tot) part_dw[(long)blockIdx.x * tot + i] = s;
    else if (part_db) part_db[(long)blockIdx.x * n + (i - tot)] = s;
  }
}

static int skinny_blocks(long m) {
  long b = (m + 3) / 4;                                  // 4 waves per workgroup, a row per wave and trip
  if (b > (long)kNumCU * 8) b = (long)kNumCU * 8;
  return (int)(b < 1 ? 1 : b);
}

static int round_n(int n) { return n <= 2 ? 2 : (n <= 4 ? 4 : (n <= 8 ? 8 : (n <= 12 ? 12 : 16))); }
static int round_qk(int k) { const int q = (k + 63) / 64; return q <= 1 ? 1 : (q <= 2 ? 2 : (q <= 4 ? 4 : (q <= 8 ? 8 : 16))); }

bool skinny_n_ok(int din, int dout, int trans_w) {
  if (trans_w || dout > SKN || din < 32 || din > 64 * SK_KPL) return false;
  // weight gradient: QK x N accumulators per lane, 4 waves x (din dout + dout) floats of LDS
  return round_qk(din) * round_n(dout) <= 128 && (size_t)4 * ((size_t)din * dout + dout) * 4 <= 64 * 1024;
}
bool skinny_k_ok(int din, int dout) { return din <= SKN && dout >= 32; }

int launch_skinny_n_fwd(const float* x, long m, int din, long x_ld, const float* w, long w_ld, const float* bias, float* y,
                        int dout, long y_ld, int act, hipStream_t s) {
  const dim3 grid(skinny_blocks(m));
#define KGCN_CASE(NN) case NN: hipLaunchKernelGGL(skinny_n_fwd_kernel<NN>, grid, dim3(256), 0, s, x, m, din, x_ld, w, w_ld, bias, y, y_ld, act, dout); break;
  switch (round_n(dout)) {
    KGCN_CASE(2) KGCN_CASE(4) KGCN_CASE(8) KGCN_CASE(12) KGCN_CASE(16)
    default: return fail("skinny_n_fwd: dout=%d", dout);
  }
#undef KGCN_CASE
  return check_launch("skinny_n_fwd_kernel");
}

int launch_skinny_k_fwd(const float* x, long m, int din, long x_ld, const float* w, long w_ld, int trans_w, const float* bias,
                        float* y, int dout, long y_ld, int act, hipStream_t s) {
  const dim3 grid(skinny_blocks(m));
#define KGCN_CASE(KK) case KK: hipLaunchKernelGGL(skinny_k_fwd_kernel<KK>, grid, dim3(256), 0, s, x, m, x_ld, w, w_ld, trans_w, dout, bias, y, y_ld, act); break;
  switch (din) {
    KGCN_CASE(1) KGCN_CASE(2) KGCN_CASE(3) KGCN_CASE(4) KGCN_CASE(5) KGCN_CASE(6) KGCN_CASE(7) KGCN_CASE(8)
    KGCN_CASE(9) KGCN_CASE(10) KGCN_CASE(11) KGCN_CASE(12) KGCN_CASE(13) KGCN_CASE(14) KGCN_CASE(15) KGCN_CASE(16)
    default: return fail("skinny_k_fwd: din=%d", din);
  }
#undef KGCN_CASE
  return check_launch("skinny_k_fwd_kernel");
}

int skinny_wgrad_parts(long m) {
  long b = (m + 15) / 16;                                // >= 4 rows per wave
  if (b > 4L * kNumCU) b = 4L * kNumCU;                  // <= 1,024 partials (the workspace query sizes for them)
  return (int)(b < 1 ? 1 : b);
}

int launch_skinny_n_wgrad(const float* x, long m, int din, long x_ld, const float* g, long g_ld, int dout, float* part_dw,
                          float* part_db, int nparts, hipStream_t s) {
  const dim3 grid(nparts);
  const size_t lds = (size_t)4 * ((size_t)din * dout + dout) * 4;
  const int nn = round_n(dout), qk = round_qk(din);
#define KGCN_W(NN, QQ)                                                                                                \
  if (nn == NN && qk == QQ) {                                                                                         \
    hipLaunchKernelGGL((skinny_n_wgrad_kernel<NN, QQ>), grid, dim3(256), lds, s, x, m, din, x_ld, g, g_ld, part_dw,   \
                       part_db, dout);                                                                                \
    return check_launch("skinny_n_wgrad_kernel");                                                                     \
  }
  KGCN_W(2, 1) KGCN_W(2, 2) KGCN_W(2, 4) KGCN_W(2, 8) KGCN_W(2, 16)
  KGCN_W(4, 1) KGCN_W(4, 2) KGCN_W(4, 4) KGCN_W(4, 8) KGCN_W(4, 16)
  KGCN_W(8, 1) KGCN_W(8, 2) KGCN_W(8, 4) KGCN_W(8, 8) KGCN_W(8, 16)
  KGCN_W(12, 1) KGCN_W(12, 2) KGCN_W(12, 4) KGCN_W(12, 8)
  KGCN_W(16, 1) KGCN_W(16, 2) KGCN_W(16, 4) KGCN_W(16, 8)
#undef KGCN_W
  return fail("skinny_n_wgrad: unsupported shape %d x %d", din, dout);
}

}  // namespace kgcn
