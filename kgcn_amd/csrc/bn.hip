// GraphBatchNormalization (kgcn/layers.py:170-220): Keras BatchNormalization over the VALID node rows of a padded
// batch -- the reference gathers the first enabled_node_nums[b] rows of every graph, normalises the stacked [rows, D]
// matrix per feature and pads the result back with zeros (:196-210); without enabled_node_nums every row is valid
// (:211-216).  Nothing is gathered here: the kernels walk the padded [T, N, D] tensor and mask rows n >= enabled[t].
//
//   statistics (training mode)   mean[c] = sum_valid x / n ;  var[c] = sum_valid (x - mean)^2 / n      (two passes, as
//                                tf.nn.moments computes them: population variance, no single-pass cancellation)
//   forward                      y = gamma (x - mean) / sqrt(var + eps) + beta   on valid rows, 0 on padding rows
//   backward                     dbeta = sum g ; dgamma = sum g xhat ;
//                                training:  dx = gamma rstd (g - dbeta / n - xhat dgamma / n)
//                                inference: dx = gamma rstd g                       (valid rows; 0 on padding rows)
// All of them are HBM-bound streaming passes; reductions are deterministic (per-workgroup partials, fixed-order second stage).
#include "kgcn_common.h"

namespace kgcn {

int launch_reduce_partials(const float* part, int nparts, long n, float* out, hipStream_t s);
int launch_reduce_pair_now(const float* part_a, long n_a, float* out_a, const float* part_b, long n_b, float* out_b, int nparts,
                       hipStream_t s);
int launch_reduce_pair(const float* part_a, long n_a, float* out_a, const float* part_b, long n_b, float* out_b, int nparts,
                       hipStream_t s);            // (dense.hip: queued inside a deferral scope, kgcn_reduce_defer)

constexpr int BN_BLOCKS = 1024;    // partial rows of the first reduction stage

// MODE 0: acc0 = sum x                       (mean numerator)
// MODE 1: acc0 = sum (x - mean)^2            (variance numerator)
// MODE 2: acc0 = sum g, acc1 = sum g * xhat  (dbeta, dgamma), xhat = (x - mean) * rstd
// MODE 3: MODE 2 and, in the same pass, the inference-phase dx = gamma rstd g (0 on padding rows): x and g are read once
// Thread layout: column c = tid % cb of a block of cb <= 256 columns, row lane tid / cb of 256 / cb rows per pass:
// consecutive threads read consecutive floats of a row.
template <int MODE>
__global__ __launch_bounds__(256) void bn_colreduce_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                           long rows, int n_nodes, int d, const int* __restrict__ enabled,
                                                           const float* __restrict__ mean, const float* __restrict__ var,
                                                           float eps, float* __restrict__ part0, float* __restrict__ part1,
                                                           const float* __restrict__ gamma = nullptr,
                                                           float* __restrict__ dx = nullptr,
                                                           const float* __restrict__ aout = nullptr, int dact = 0) {
  __shared__ float red[2][256];
  const int tid = threadIdx.x;
  for (int c0 = 0; c0 < d; c0 += 256) {
    const int cb = d - c0 < 256 ? d - c0 : 256;
    const int rpp = 256 / cb;                         // rows per pass of this workgroup
    const int c = tid % cb, rl = tid / cb;
    float a0 = 0.f, a1 = 0.f;
    if (rl < rpp) {
      float mu = 0.f, rs = 0.f, gr = 0.f;
      if constexpr (MODE >= 1) mu = mean[c0 + c];
      if constexpr (MODE >= 2) rs = 1.0f / __builtin_sqrtf(var[c0 + c] + eps);
      if constexpr (MODE == 3) gr = gamma[c0 + c] * rs;
      for (long r = (long)blockIdx.x * rpp + rl; r < rows; r += (long)gridDim.x * rpp) {
        if (enabled) {
          const long t = r / n_nodes;
          if ((int)(r - t * n_nodes) >= enabled[t]) {
            if constexpr (MODE == 3) dx[r * d + c0 + c] = 0.f;
            continue;
          }
        }
        const float v = x[r * d + c0 + c];
        if constexpr (MODE == 0) a0 += v;
        if constexpr (MODE == 1) { const float dv = v - mu; a0 += dv * dv; }
        if constexpr (MODE >= 2) {
          float gv = g[r * d + c0 + c];
          if (dact != KGCN_ACT_NONE) gv *= act_dout(aout[r * d + c0 + c], dact);   // backward of act(bn(x)): uniform branch
          a0 += gv;
          a1 += gv * ((v - mu) * rs);
          if constexpr (MODE == 3) dx[r * d + c0 + c] = gr * gv;
        }
      }
    }
    red[0][tid] = a0;
    red[1][tid] = a1;
    __syncthreads();
    if (tid < cb) {
      float s0 = 0.f, s1 = 0.f;
      for (int k = 0; k < rpp; ++k) { s0 += red[0][k * cb + tid]; s1 += red[1][k * cb + tid]; }
      part0[(long)blockIdx.x * d + c0 + tid] = s0;
      if constexpr (MODE >= 2) part1[(long)blockIdx.x * d + c0 + tid] = s1;
    }
    __syncthreads();
  }
}

// out[c] = in[c] * scale (the reduced sums -> mean / variance)
__global__ void bn_scale_kernel(float* __restrict__ v, int d, const int* __restrict__ enabled, long graphs, int n_nodes,
                                long* __restrict__ count_out) {
  // n = number of valid rows: sum of min(enabled[t], N) (or T * N); computed by one workgroup, deterministic
  __shared__ long red[256];
  long n = 0;
  if (enabled) {
    for (long t = threadIdx.x; t < graphs; t += 256) {
      const int e = enabled[t];
      n += e < 0 ? 0 : (e > n_nodes ? n_nodes : e);
    }
  } else if (threadIdx.x == 0) {
    n = graphs * n_nodes;
  }
  red[threadIdx.x] = n;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  n = red[0];
  if (threadIdx.x == 0 && count_out) *count_out = n;
  const float inv = n > 0 ? 1.0f / (float)n : 0.f;
  for (int c = threadIdx.x; c < d; c += 256) v[c] *= inv;
}

template <bool VEC>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, long rows, int n_nodes, int d,
                                                       const int* __restrict__ enabled, const float* __restrict__ mean,
                                                       const float* __restrict__ var, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, float* __restrict__ y,
                                                       int act) {
  const int dv = VEC ? d >> 2 : d;
  const long total = rows * dv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / dv;
    const int c = (int)(i - r * dv) * (VEC ? 4 : 1);
    bool valid = true;
    if (enabled) {
      const long t = r / n_nodes;
      valid = (int)(r - t * n_nodes) < enabled[t];
    }
    if constexpr (VEC) {
      // the activation of the model follows the zero padding (tf.sigmoid(bn(...)): padding rows become act(0))
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      if (valid) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * d + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float rs = 1.0f / __builtin_sqrtf(var[c + j] + eps);
          o[j] = (v[j] - mean[c + j]) * rs * gamma[c + j] + beta[c + j];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = act_fwd(o[j], act);
      *reinterpret_cast<f32x4*>(y + r * d + c) = o;
    } else {
      float o = 0.f;
      if (valid) o = (x[r * d + c] - mean[c]) * (1.0f / __builtin_sqrtf(var[c] + eps)) * gamma[c] + beta[c];
      y[r * d + c] = act_fwd(o, act);
    }
  }
}

// The same pass with a FIXED column group per thread: lane (row sub-index, column group of V floats), dvp = d / V rounded up to
// a power of two (lanes beyond d / V idle: 7 of 32 at d = 50), so a thread's statistics / scale / shift are loop constants and
// no index is divided -- the kernel above spends ~100 vector instructions per element group on `i / dv`, a square root and
// four parameter loads (d = 50, 117,888 rows: 22 us for 47 MB).  Same arithmetic, same order: bit-identical results.
template <int V>
__global__ __launch_bounds__(256) void bn_apply_cols_kernel(const float* __restrict__ x, long rows, int n_nodes, int d, int lg,
                                                            const int* __restrict__ enabled, const float* __restrict__ mean,
                                                            const float* __restrict__ var, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, float* __restrict__ y,
                                                            int act) {
  typedef float fv __attribute__((ext_vector_type(V)));
  const int cg = threadIdx.x & ((1 << lg) - 1), rsub = threadIdx.x >> lg, rpp = 256 >> lg;
  const int c = cg * V;
  if (c >= d) return;
  float mu[V], rs[V], ga[V], be[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    mu[j] = mean[c + j];
    rs[j] = 1.0f / __builtin_sqrtf(var[c + j] + eps);
    ga[j] = gamma[c + j];
    be[j] = beta[c + j];
  }
  for (long r = (long)blockIdx.x * rpp + rsub; r < rows; r += (long)gridDim.x * rpp) {
    bool valid = true;
    if (enabled) {
      const long t = r / n_nodes;
      valid = (int)(r - t * n_nodes) < enabled[t];
    }
    fv o;
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = 0.f;
    if (valid) {
      const fv v = *reinterpret_cast<const fv*>(x + r * d + c);
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = (v[j] - mu[j]) * rs[j] * ga[j] + be[j];
    }
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = act_fwd(o[j], act);
    *reinterpret_cast<fv*>(y + r * d + c) = o;
  }
}

// dx; dgamma / dbeta are final (reduced) here.  inv_n = 0 selects the inference form.
__global__ __launch_bounds__(256) void bn_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ g, long rows,
                                                        int n_nodes, int d, const int* __restrict__ enabled,
                                                        const float* __restrict__ mean, const float* __restrict__ var,
                                                        const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                                        const float* __restrict__ dbeta, float eps, int training,
                                                        const long* __restrict__ count, float* __restrict__ dx,
                                                        const float* __restrict__ aout, int dact) {
  const long total = rows * d;
  const float inv_n = (training && *count > 0) ? 1.0f / (float)*count : 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / d;
    const int c = (int)(i - r * d);
    bool valid = true;
    if (enabled) {
      const long t = r / n_nodes;
      valid = (int)(r - t * n_nodes) < enabled[t];
    }
    float o = 0.f;
    if (valid) {
      const float rs = 1.0f / __builtin_sqrtf(var[c] + eps);
      const float xh = (x[i] - mean[c]) * rs;
      float gv = g[i];
      if (dact != KGCN_ACT_NONE) gv *= act_dout(aout[i], dact);
      o = gamma[c] * rs * (gv - inv_n * (dbeta[c] + xh * dgamma[c]));
    }
    dx[i] = o;
  }
}

static int bn_grid(long work) {
  long b = (work + 255) / 256;
  if (b > (long)kNumCU * 16) b = (long)kNumCU * 16;
  return b < 1 ? 1 : (int)b;
}

// workgroups of a column reduction: ~8 row passes each (a pass covers 256 / min(d, 256) rows), at most BN_BLOCKS
static int bn_blocks(long rows, int d) {
  const int rpp = 256 / (d < 256 ? d : 256);
  long nb = (rows + 8L * rpp - 1) / (8L * rpp);
  if (nb > BN_BLOCKS) nb = BN_BLOCKS;
  return nb < 1 ? 1 : (int)nb;
}

static int bn_check(const char* who, const float* x, int64_t graphs, int32_t n_nodes, int32_t d) {
  if (graphs < 0 || n_nodes <= 0 || d <= 0) return fail("%s: bad shape T=%lld N=%d D=%d", who, (long long)graphs, n_nodes, d);
  if (graphs > 0 && !x) return fail("%s: x is NULL", who);
  return 0;
}

}  // namespace kgcn

using namespace kgcn;

extern "C" int64_t kgcn_graph_bn_workspace_bytes(int32_t d) {
  if (d <= 0) return 0;
  return ((int64_t)2 * BN_BLOCKS * d + 8) * 4 + 16;      // two partial arrays + the valid-row count
}

extern "C" int kgcn_graph_bn_stats_f32(const float* x, int64_t graphs, int32_t n_nodes, int32_t d, const int32_t* enabled,
                                       float* mean, float* var, void* workspace, int64_t workspace_bytes, void* stream) {
  if (int rc = bn_check("kgcn_graph_bn_stats_f32", x, graphs, n_nodes, d)) return rc;
  if (!mean || !var) return fail("kgcn_graph_bn_stats_f32: mean/var is NULL");
  if (!workspace || workspace_bytes < kgcn_graph_bn_workspace_bytes(d))
    return fail("kgcn_graph_bn_stats_f32: workspace %lld < %lld bytes", (long long)workspace_bytes,
                (long long)kgcn_graph_bn_workspace_bytes(d));
  hipStream_t s = as_stream(stream);
  const long rows = (long)graphs * n_nodes;
  float* part = static_cast<float*>(workspace);
  int nb = bn_blocks(rows, d);
  hipLaunchKernelGGL(bn_colreduce_kernel<0>, dim3(nb), dim3(256), 0, s, x, nullptr, rows, n_nodes, d, enabled, nullptr,
                     nullptr, 0.f, part, nullptr);
  if (int rc = launch_reduce_partials(part, nb, d, mean, s)) return rc;
  hipLaunchKernelGGL(bn_scale_kernel, dim3(1), dim3(256), 0, s, mean, d, enabled, (long)graphs, n_nodes, nullptr);
  hipLaunchKernelGGL(bn_colreduce_kernel<1>, dim3(nb), dim3(256), 0, s, x, nullptr, rows, n_nodes, d, enabled, mean,
                     nullptr, 0.f, part, nullptr);
  if (int rc = launch_reduce_partials(part, nb, d, var, s)) return rc;
  hipLaunchKernelGGL(bn_scale_kernel, dim3(1), dim3(256), 0, s, var, d, enabled, (long)graphs, n_nodes, nullptr);
  return check_launch("bn_stats");
}

static int bn_apply_impl(const float* x, int64_t graphs, int32_t n_nodes, int32_t d, const int32_t* enabled,
                         const float* mean, const float* var, const float* gamma, const float* beta, float eps, float* y,
                         int act, void* stream) {
  if (act < KGCN_ACT_NONE || act > KGCN_ACT_TANH) return fail("kgcn_graph_bn_apply_f32: unknown activation code %d", act);
  if (int rc = bn_check("kgcn_graph_bn_apply_f32", x, graphs, n_nodes, d)) return rc;
  if (graphs == 0) return 0;
  if (!mean || !var || !gamma || !beta || !y) return fail("kgcn_graph_bn_apply_f32: NULL operand");
  const long rows = (long)graphs * n_nodes;
  const bool vec = (d % 4 == 0) && aligned16(x) && aligned16(y);
  {
    // fixed column group per thread (d / V <= 256 groups): V = 4, or 2 for even widths like the 50 of model.py / model_multitask.py
    const int V = vec ? 4 : ((d % 2 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) % 8 == 0) ? 2 : 1);
    const int dv = d / V;
    if (dv <= 256) {
      int lg = 0;
      while ((1 << lg) < dv) ++lg;
      const int rpp = 256 >> lg;
      long nb = (rows + rpp - 1) / rpp;
      if (nb > (long)kNumCU * 8) nb = (long)kNumCU * 8;
      const dim3 grid((unsigned)(nb < 1 ? 1 : nb));
      if (V == 4)
        hipLaunchKernelGGL(bn_apply_cols_kernel<4>, grid, dim3(256), 0, as_stream(stream), x, rows, n_nodes, d, lg, enabled, mean,
                           var, gamma, beta, eps, y, act);
      else if (V == 2)
        hipLaunchKernelGGL(bn_apply_cols_kernel<2>, grid, dim3(256), 0, as_stream(stream), x, rows, n_nodes, d, lg, enabled, mean,
                           var, gamma, beta, eps, y, act);
      else
        hipLaunchKernelGGL(bn_apply_cols_kernel<1>, grid, dim3(256), 0, as_stream(stream), x, rows, n_nodes, d, lg, enabled, mean,
                           var, gamma, beta, eps, y, act);
      return check_launch("bn_apply_cols_kernel");
    }
  }
  if (vec)
    hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(bn_grid(rows * (d / 4))), dim3(256), 0, as_stream(stream), x, rows, n_nodes,
                       d, enabled, mean, var, gamma, beta, eps, y, act);
  else
    hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(bn_grid(rows * d)), dim3(256), 0, as_stream(stream), x, rows, n_nodes, d,
                       enabled, mean, var, gamma, beta, eps, y, act);
  return check_launch("bn_apply_kernel");
}

extern "C" int kgcn_graph_bn_apply_f32(const float* x, int64_t graphs, int32_t n_nodes, int32_t d, const int32_t* enabled,
                                       const float* mean, const float* var, const float* gamma, const float* beta,
                                       float eps, float* y, void* stream) {
  return bn_apply_impl(x, graphs, n_nodes, d, enabled, mean, var, gamma, beta, eps, y, KGCN_ACT_NONE, stream);
}

extern "C" int kgcn_graph_bn_apply_act_f32(const float* x, int64_t graphs, int32_t n_nodes, int32_t d,
                                           const int32_t* enabled, const float* mean, const float* var, const float* gamma,
                                           const float* beta, float eps, int32_t act, float* y, void* stream) {
  return bn_apply_impl(x, graphs, n_nodes, d, enabled, mean, var, gamma, beta, eps, y, act, stream);
}

static int bn_bwd_impl(const float* x, const float* grad, const float* aout, int dact, int64_t graphs, int32_t n_nodes,
                       int32_t d, const int32_t* enabled, const float* mean, const float* var, const float* gamma, float eps,
                       int32_t training, float* dx, float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes,
                       void* stream) {
  if (dact < KGCN_ACT_NONE || dact > KGCN_ACT_TANH) return fail("kgcn_graph_bn_bwd_f32: unknown activation code %d", dact);
  if (dact != KGCN_ACT_NONE && !aout) return fail("kgcn_graph_bn_bwd_f32: act_out is NULL");
  if (int rc = bn_check("kgcn_graph_bn_bwd_f32", x, graphs, n_nodes, d)) return rc;
  if (!grad || !mean || !var || !gamma || !dgamma || !dbeta) return fail("kgcn_graph_bn_bwd_f32: NULL operand");
  if (!workspace || workspace_bytes < kgcn_graph_bn_workspace_bytes(d))
    return fail("kgcn_graph_bn_bwd_f32: workspace %lld < %lld bytes", (long long)workspace_bytes,
                (long long)kgcn_graph_bn_workspace_bytes(d));
  hipStream_t s = as_stream(stream);
  const long rows = (long)graphs * n_nodes;
  float* part0 = static_cast<float*>(workspace);
  float* part1 = part0 + (long)BN_BLOCKS * d;
  long* count = reinterpret_cast<long*>(reinterpret_cast<char*>(workspace) +
                                        ((((size_t)2 * BN_BLOCKS * d) * 4 + 15) & ~(size_t)15));
  int nb = bn_blocks(rows, d);
  if (dx && !training) {
    // inference phase: dx does not depend on the reductions -> one pass over x and g
    hipLaunchKernelGGL(bn_colreduce_kernel<3>, dim3(nb), dim3(256), 0, s, x, grad, rows, n_nodes, d, enabled, mean, var, eps,
                       part0, part1, gamma, dx, aout, dact);
    // (nothing in this call reads d gamma / d beta back: their second stage may wait for the step's one reduction launch)
    return launch_reduce_pair(part0, d, dbeta, part1, d, dgamma, nb, s);
  }
  hipLaunchKernelGGL(bn_colreduce_kernel<2>, dim3(nb), dim3(256), 0, s, x, grad, rows, n_nodes, d, enabled, mean, var, eps,
                     part0, part1, nullptr, nullptr, aout, dact);
  if (int rc = launch_reduce_pair_now(part0, d, dbeta, part1, d, dgamma, nb, s)) return rc;
  if (dx) {
    // the valid-row count (scale kernel with no array to scale: d = 0)
    hipLaunchKernelGGL(bn_scale_kernel, dim3(1), dim3(256), 0, s, part0, 0, enabled, (long)graphs, n_nodes, count);
    if (rows > 0)
      hipLaunchKernelGGL(bn_bwd_dx_kernel, dim3(bn_grid(rows * d)), dim3(256), 0, s, x, grad, rows, n_nodes, d, enabled, mean,
                         var, gamma, dgamma, dbeta, eps, training, count, dx, aout, dact);
  }
  return check_launch("bn_bwd");
}

extern "C" int kgcn_graph_bn_bwd_f32(const float* x, const float* grad, int64_t graphs, int32_t n_nodes, int32_t d,
                                     const int32_t* enabled, const float* mean, const float* var, const float* gamma,
                                     float eps, int32_t training, float* dx, float* dgamma, float* dbeta, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
  return bn_bwd_impl(x, grad, nullptr, KGCN_ACT_NONE, graphs, n_nodes, d, enabled, mean, var, gamma, eps, training, dx, dgamma,
                     dbeta, workspace, workspace_bytes, stream);
}

extern "C" int kgcn_graph_bn_bwd_dact_f32(const float* x, const float* grad, const float* act_out, int32_t act,
                                          int64_t graphs, int32_t n_nodes, int32_t d, const int32_t* enabled,
                                          const float* mean, const float* var, const float* gamma, float eps,
                                          int32_t training, float* dx, float* dgamma, float* dbeta, void* workspace,
                                          int64_t workspace_bytes, void* stream) {
  return bn_bwd_impl(x, grad, act_out, act, graphs, n_nodes, d, enabled, mean, var, gamma, eps, training, dx, dgamma, dbeta,
                     workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------------
// Activations that have no producer kernel to ride in (after the fused GraphConv kernels, after BN in training mode) and
// the backward of every fused activation:  dpre = grad (.) act'(act_out), in place when dpre == grad.
// ------------------------------------------------------------------------------------------------
namespace kgcn {
template <bool BWD>
__global__ __launch_bounds__(256) void act_kernel(const float* __restrict__ a, const float* __restrict__ g, long n, int act,
                                                  float* __restrict__ o) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 av = reinterpret_cast<const f32x4*>(a)[i];
    f32x4 r;
    if constexpr (BWD) {
      const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = gv[j] * act_dout(av[j], act);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = act_fwd(av[j], act);
    }
    reinterpret_cast<f32x4*>(o)[i] = r;
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    o[i] = BWD ? g[i] * act_dout(a[i], act) : act_fwd(a[i], act);
}
}  // namespace kgcn

extern "C" int kgcn_act_fwd_f32(const float* x, int64_t n, int32_t act, float* y, void* stream) {
  if (act < KGCN_ACT_NONE || act > KGCN_ACT_TANH) return fail("kgcn_act_fwd_f32: unknown activation code %d", act);
  if (n <= 0) return 0;
  if (!x || !y) return fail("kgcn_act_fwd_f32: NULL operand");
  if (!aligned16(x) || !aligned16(y)) return fail("kgcn_act_fwd_f32: tensors not 16-byte aligned");
  hipLaunchKernelGGL(act_kernel<false>, dim3(bn_grid(n / 4 + 1)), dim3(256), 0, as_stream(stream), x, nullptr, (long)n, act, y);
  return check_launch("act_kernel");
}

extern "C" int kgcn_act_bwd_f32(const float* act_out, const float* grad, int64_t n, int32_t act, float* dpre, void* stream) {
  if (act < KGCN_ACT_NONE || act > KGCN_ACT_TANH) return fail("kgcn_act_bwd_f32: unknown activation code %d", act);
  if (n <= 0) return 0;
  if (!act_out || !grad || !dpre) return fail("kgcn_act_bwd_f32: NULL operand");
  if (!aligned16(act_out) || !aligned16(grad) || !aligned16(dpre)) return fail("kgcn_act_bwd_f32: tensors not 16-byte aligned");
  hipLaunchKernelGGL(act_kernel<true>, dim3(bn_grid(n / 4 + 1)), dim3(256), 0, as_stream(stream), act_out, grad, (long)n, act,
                     dpre);
  return check_launch("act_kernel");
}
