// Library bookkeeping (error text, ABI version) and the small reductions of the path:
// GraphGather (kgcn/layers.py:163-164) forward/backward and the dot product behind d eps of
// GINAggregate (kgcn/layers.py:469).
#include "kgcn_common.h"

namespace kgcn {

char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return 1;
}

// out[b, c] = sum_n x[b, n, c].  One thread per (b, 4 columns) when d % 4 == 0: consecutive
// threads read consecutive 16-byte words of a row, the n loop strides by d.
template <int VEC>
__global__ __launch_bounds__(256) void gather_fwd_kernel(const float* __restrict__ x, long batch,
                                                         int n_nodes, int d,
                                                         float* __restrict__ out, long out_ld) {
  const int dv = d / VEC;
  const long total = batch * dv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / dv;
    const int c = (int)(i - b * dv) * VEC;
    const float* src = x + (b * n_nodes) * (long)d + c;
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
#pragma unroll 4
    for (int n = 0; n < n_nodes; ++n) {
      if constexpr (VEC == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (long)n * d);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += v[j];
      } else if constexpr (VEC == 2) {
        const f32x2 v = *reinterpret_cast<const f32x2*>(src + (long)n * d);
        acc[0] += v[0];
        acc[1] += v[1];
      } else {
        acc[0] += src[(long)n * d];
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) out[b * out_ld + c + j] = acc[j];
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void gather_bwd_kernel(const float* __restrict__ g, long batch,
                                                         int n_nodes, int d,
                                                         float* __restrict__ dx) {
  const int dv = d / VEC;
  const long total = batch * n_nodes * dv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long bn = i / dv;
    const int c = (int)(i - bn * dv) * VEC;
    const long b = bn / n_nodes;
    if constexpr (VEC == 4) {
      *reinterpret_cast<f32x4*>(dx + bn * d + c) = *reinterpret_cast<const f32x4*>(g + b * d + c);
    } else if constexpr (VEC == 2) {
      *reinterpret_cast<f32x2*>(dx + bn * d + c) = *reinterpret_cast<const f32x2*>(g + b * d + c);
    } else {
      dx[bn * d + c] = g[b * d + c];
    }
  }
}

// dx[b, n, :] = dx_in[b, n, :] + g[b, :]: the gradient of a tensor that feeds BOTH a GraphGather read-out and the next layer
// (example_model/model_gin.py:45-60: every block's output is gathered and passed on) in one pass instead of a broadcast
// pass plus autograd's accumulation pass
__global__ __launch_bounds__(256) void gather_bwd_add_kernel(const float* __restrict__ g, const float* __restrict__ dx_in,
                                                             long batch, int n_nodes, int d, float* __restrict__ dx) {
  const int dv = d >> 2;
  const long total = batch * n_nodes * dv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long bn = i / dv;
    const int c = (int)(i - bn * dv) * 4;
    const long b = bn / n_nodes;
    const f32x4 a = *reinterpret_cast<const f32x4*>(dx_in + bn * d + c);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + b * d + c);
    *reinterpret_cast<f32x4*>(dx + bn * d + c) = a + gv;
  }
}

constexpr int kDotBlocks = 1024;

// 16-byte loads, four independent accumulators per thread and four loads of each operand in flight (the scalar form
// with one accumulator reached 3.4 TB/s); the summation order is fixed by (grid, n) alone.
__global__ __launch_bounds__(256) void dot_partial_kernel(const float* __restrict__ a,
                                                          const float* __restrict__ b, long n,
                                                          float* __restrict__ part, int vec) {
  __shared__ float red[4];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const long tid = (long)blockIdx.x * 256 + threadIdx.x, nthr = (long)gridDim.x * 256;
  long done = 0;
  if (vec) {
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(b);
    const long n4 = n >> 2;
    long i = tid;
    for (; i + 3 * nthr < n4; i += 4 * nthr) {
      const f32x4 x0 = a4[i], x1 = a4[i + nthr], x2 = a4[i + 2 * nthr], x3 = a4[i + 3 * nthr];
      const f32x4 y0 = b4[i], y1 = b4[i + nthr], y2 = b4[i + 2 * nthr], y3 = b4[i + 3 * nthr];
      s0 += (x0[0] * y0[0] + x0[1] * y0[1]) + (x0[2] * y0[2] + x0[3] * y0[3]);
      s1 += (x1[0] * y1[0] + x1[1] * y1[1]) + (x1[2] * y1[2] + x1[3] * y1[3]);
      s2 += (x2[0] * y2[0] + x2[1] * y2[1]) + (x2[2] * y2[2] + x2[3] * y2[3]);
      s3 += (x3[0] * y3[0] + x3[1] * y3[1]) + (x3[2] * y3[2] + x3[3] * y3[3]);
    }
    for (; i < n4; i += nthr) {
      const f32x4 x0 = a4[i], y0 = b4[i];
      s0 += (x0[0] * y0[0] + x0[1] * y0[1]) + (x0[2] * y0[2] + x0[3] * y0[3]);
    }
    done = n4 << 2;
  }
  for (long i = done + tid; i < n; i += nthr) s1 += a[i] * b[i];
  float s = (s0 + s1) + (s2 + s3);
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void dot_final_kernel(const float* __restrict__ part, int nparts,
                                                        float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  // eight loads in flight per lane, added in the order of the plain loop (same bits): 20,000 per-graph partials of the fused d eps
  // took 23.6 us as 78 dependent load-add steps (rocprofv3, cfg5 step)
  int i = threadIdx.x;
  for (; i + 7 * 256 < nparts; i += 8 * 256) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[i + u * 256];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; i < nparts; i += 256) s += part[i];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

static unsigned grid_for(long total_threads) {
  long blocks = (total_threads + 255) / 256;
  const long cap = (long)kNumCU * 16;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

// Streaming probe of the box's HBM (measurement aid of bench.py: "slow box" vs "regression"): grid-stride float4 streams in the
// read : write mix of the priced kernel.  mix 0: b = a (1 : 1, the SpMM's mix); 1: b = a + a2 (2 : 1, the fused backward's mix);
// 2: read only; 3: write only.
template <int MIX>
__global__ __launch_bounds__(256) void hbm_probe_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ a2,
                                                         f32x4* __restrict__ b, size_t n) {
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  const f32x4 one = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if constexpr (MIX == 0) b[i] = a[i];
    else if constexpr (MIX == 1) b[i] = a[i] + a2[i];
    else if constexpr (MIX == 2) s += a[i];
    else b[i] = one;
  }
  if constexpr (MIX == 2) {
    if (s[0] + s[1] + s[2] + s[3] == 123.456f) b[0] = s;
  }
}

// Several strided 2-D copies in one launch (grid.y = job): the concatenation of the C kernels / biases of a multi-channel GraphConv
// into the operand of its ONE GEMM, and the split of that operand's gradient into C contiguous tensors (layers.py:68-78 builds C
// MatMul ops; here it is one GEMM over [W_0 | W_1 | ...]).
struct Copy2dJobs {
  const float* src[KGCN_COPY2D_MAX_JOBS];
  float* dst[KGCN_COPY2D_MAX_JOBS];
  long rows[KGCN_COPY2D_MAX_JOBS], cols[KGCN_COPY2D_MAX_JOBS], src_ld[KGCN_COPY2D_MAX_JOBS], dst_ld[KGCN_COPY2D_MAX_JOBS];
};
__global__ __launch_bounds__(256) void copy2d_multi_kernel(Copy2dJobs jb) {
  const int q = blockIdx.y;
  const float* __restrict__ src = jb.src[q];
  float* __restrict__ dst = jb.dst[q];
  const long cols = jb.cols[q], total = jb.rows[q] * cols, sl = jb.src_ld[q], dl = jb.dst_ld[q];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / cols, c = i - r * cols;
    dst[r * dl + c] = src[r * sl + c];
  }
}

}  // namespace kgcn

using namespace kgcn;

extern "C" int kgcn_copy2d_multi_f32(const kgcn_copy2d_job* jobs, int32_t num_jobs, void* stream) {
  if (num_jobs < 0 || (num_jobs > 0 && !jobs)) return fail("kgcn_copy2d_multi_f32: bad job list");
  for (int base = 0; base < num_jobs; base += KGCN_COPY2D_MAX_JOBS) {
    const int n = num_jobs - base < KGCN_COPY2D_MAX_JOBS ? num_jobs - base : KGCN_COPY2D_MAX_JOBS;
    Copy2dJobs jb{};
    long most = 0;
    for (int q = 0; q < n; ++q) {
      const kgcn_copy2d_job& j = jobs[base + q];
      if (j.rows < 0 || j.cols < 0 || j.src_ld < j.cols || j.dst_ld < j.cols)
        return fail("kgcn_copy2d_multi_f32: job %d: rows=%lld cols=%lld src_ld=%lld dst_ld=%lld", base + q, (long long)j.rows,
                    (long long)j.cols, (long long)j.src_ld, (long long)j.dst_ld);
      if (j.rows * j.cols > 0 && (!j.src || !j.dst)) return fail("kgcn_copy2d_multi_f32: job %d: NULL operand", base + q);
      jb.src[q] = j.src; jb.dst[q] = j.dst; jb.rows[q] = (long)j.rows; jb.cols[q] = (long)j.cols;
      jb.src_ld[q] = (long)j.src_ld; jb.dst_ld[q] = (long)j.dst_ld;
      if (j.rows * j.cols > most) most = (long)(j.rows * j.cols);
    }
    if (most == 0) continue;
    long blocks = (most + 255) / 256;
    if (blocks > 4L * kNumCU) blocks = 4L * kNumCU;
    hipLaunchKernelGGL(copy2d_multi_kernel, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, static_cast<hipStream_t>(stream), jb);
    if (int rc = check_launch("copy2d_multi_kernel")) return rc;
  }
  return 0;
}

extern "C" int kgcn_hbm_probe(int32_t mix, const void* a, const void* a2, void* b, int64_t bytes, void* stream) {
  if (mix < 0 || mix > 3) return fail("kgcn_hbm_probe: mix %d", mix);
  if (bytes <= 0 || (bytes & 15)) return fail("kgcn_hbm_probe: bytes=%lld (a positive multiple of 16)", (long long)bytes);
  if (!b || (mix != 3 && !a) || (mix == 1 && !a2)) return fail("kgcn_hbm_probe: NULL operand");
  if (!aligned16(a) || !aligned16(a2) || !aligned16(b)) return fail("kgcn_hbm_probe: operands must be 16-byte aligned");
  const size_t n = (size_t)bytes / 16;
  const dim3 grid((unsigned)kNumCU * 32), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const f32x4* pa = static_cast<const f32x4*>(a);
  const f32x4* pa2 = static_cast<const f32x4*>(a2);
  f32x4* pb = static_cast<f32x4*>(b);
  if (mix == 0) hipLaunchKernelGGL(hbm_probe_kernel<0>, grid, block, 0, s, pa, pa2, pb, n);
  else if (mix == 1) hipLaunchKernelGGL(hbm_probe_kernel<1>, grid, block, 0, s, pa, pa2, pb, n);
  else if (mix == 2) hipLaunchKernelGGL(hbm_probe_kernel<2>, grid, block, 0, s, pa, pa2, pb, n);
  else hipLaunchKernelGGL(hbm_probe_kernel<3>, grid, block, 0, s, pa, pa2, pb, n);
  return check_launch("hbm_probe_kernel");
}

extern "C" int kgcn_abi_version(void) { return KGCN_HIP_ABI_VERSION; }
extern "C" int64_t kgcn_csr_batch_size(void) { return (int64_t)sizeof(kgcn_csr_batch); }
extern "C" const char* kgcn_last_error(void) { return error_buffer(); }
extern "C" const char* kgcn_build_arch(void) { return "gfx950"; }

static int gather_fwd_impl(const char* who, const float* x, int64_t batch, int32_t n_nodes, int32_t d, float* out, int64_t out_ld,
                           void* stream) {
  if (batch < 0 || n_nodes < 0 || d < 0) return fail("%s: negative shape", who);
  if (batch == 0 || d == 0) return 0;
  if (!out || (!x && n_nodes > 0)) return fail("%s: NULL operand", who);
  if (out_ld < d) return fail("%s: out_ld=%lld smaller than d=%d", who, (long long)out_ld, d);
  const bool v4 = (d % 4 == 0) && (out_ld % 4 == 0) && aligned16(x) && aligned16(out);
  const bool v2 = (d % 2 == 0) && (out_ld % 2 == 0) &&
                  ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) % 8 == 0);
  if (v4)
    hipLaunchKernelGGL((gather_fwd_kernel<4>), dim3(grid_for(batch * (d / 4))), dim3(256), 0,
                       as_stream(stream), x, (long)batch, n_nodes, d, out, (long)out_ld);
  else if (v2)
    hipLaunchKernelGGL((gather_fwd_kernel<2>), dim3(grid_for(batch * (d / 2))), dim3(256), 0,
                       as_stream(stream), x, (long)batch, n_nodes, d, out, (long)out_ld);
  else
    hipLaunchKernelGGL((gather_fwd_kernel<1>), dim3(grid_for(batch * d)), dim3(256), 0,
                       as_stream(stream), x, (long)batch, n_nodes, d, out, (long)out_ld);
  return check_launch("gather_fwd_kernel");
}

extern "C" int kgcn_graph_gather_fwd_f32(const float* x, int64_t batch, int32_t n_nodes, int32_t d,
                                         float* out, void* stream) {
  return gather_fwd_impl("kgcn_graph_gather_fwd_f32", x, batch, n_nodes, d, out, d, stream);
}

extern "C" int kgcn_graph_gather_fwd_ld_f32(const float* x, int64_t batch, int32_t n_nodes, int32_t d, float* out,
                                            int64_t out_ld, void* stream) {
  return gather_fwd_impl("kgcn_graph_gather_fwd_ld_f32", x, batch, n_nodes, d, out, out_ld, stream);
}

extern "C" int kgcn_graph_gather_bwd_f32(const float* dout_grad, int64_t batch, int32_t n_nodes,
                                         int32_t d, float* dx, void* stream) {
  if (batch < 0 || n_nodes < 0 || d < 0) return fail("kgcn_graph_gather_bwd_f32: negative shape");
  if (batch == 0 || d == 0 || n_nodes == 0) return 0;
  if (!dout_grad || !dx) return fail("kgcn_graph_gather_bwd_f32: NULL operand");
  const bool v4 = (d % 4 == 0) && aligned16(dout_grad) && aligned16(dx);
  const bool v2 = (d % 2 == 0) && ((reinterpret_cast<uintptr_t>(dout_grad) | reinterpret_cast<uintptr_t>(dx)) % 8 == 0);
  if (v4)
    hipLaunchKernelGGL((gather_bwd_kernel<4>), dim3(grid_for(batch * n_nodes * (d / 4))),
                       dim3(256), 0, as_stream(stream), dout_grad, (long)batch, n_nodes, d, dx);
  else if (v2)
    hipLaunchKernelGGL((gather_bwd_kernel<2>), dim3(grid_for(batch * n_nodes * (d / 2))),
                       dim3(256), 0, as_stream(stream), dout_grad, (long)batch, n_nodes, d, dx);
  else
    hipLaunchKernelGGL((gather_bwd_kernel<1>), dim3(grid_for(batch * n_nodes * d)), dim3(256), 0,
                       as_stream(stream), dout_grad, (long)batch, n_nodes, d, dx);
  return check_launch("gather_bwd_kernel");
}

extern "C" int kgcn_graph_gather_bwd_add_f32(const float* dout_grad, const float* dx_in, int64_t batch, int32_t n_nodes,
                                             int32_t d, float* dx, void* stream) {
  if (batch < 0 || n_nodes < 0 || d < 0) return fail("kgcn_graph_gather_bwd_add_f32: negative shape");
  if (batch == 0 || d == 0 || n_nodes == 0) return 0;
  if (!dout_grad || !dx_in || !dx) return fail("kgcn_graph_gather_bwd_add_f32: NULL operand");
  if (d % 4 != 0 || !aligned16(dout_grad) || !aligned16(dx_in) || !aligned16(dx))
    return fail("kgcn_graph_gather_bwd_add_f32: d must be a multiple of 4 and the tensors 16-byte aligned");
  hipLaunchKernelGGL(gather_bwd_add_kernel, dim3(grid_for(batch * n_nodes * (d / 4))), dim3(256), 0, as_stream(stream),
                     dout_grad, dx_in, (long)batch, n_nodes, d, dx);
  return check_launch("gather_bwd_add_kernel");
}

extern "C" int64_t kgcn_dot_workspace_bytes(int64_t n) {
  (void)n;
  return (int64_t)kDotBlocks * 4;
}

extern "C" int kgcn_dot_f32(const float* a, const float* b, int64_t n, float* out, void* workspace,
                            int64_t workspace_bytes, void* stream) {
  if (n < 0) return fail("kgcn_dot_f32: n < 0");
  if (!out) return fail("kgcn_dot_f32: out is NULL");
  hipStream_t s = as_stream(stream);
  if (n == 0) {
    (void)hipMemsetAsync(out, 0, 4, s);
    return 0;
  }
  if (!a || !b) return fail("kgcn_dot_f32: NULL operand");
  if (!workspace || workspace_bytes < kgcn_dot_workspace_bytes(n))
    return fail("kgcn_dot_f32: workspace too small");
  long blocks = (n + 255) / 256;
  if (blocks > kDotBlocks) blocks = kDotBlocks;
  float* part = static_cast<float*>(workspace);
  const int vec = aligned16(a) && aligned16(b) ? 1 : 0;
  hipLaunchKernelGGL(dot_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, b, (long)n,
                     part, vec);
  if (int rc = check_launch("dot_partial_kernel")) return rc;
  hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(256), 0, s, part, (int)blocks, out);
  return check_launch("dot_final_kernel");
}

// ------------------------------------------------------------------------------------------------
// Decoders (kgcn/layers.py:268-361): per graph the weighted Gram matrix
//   out[t, i, j] = sum_k w[k] x[t, i, k] x[t, j, k]           (w = NULL: plain inner products)
// GraphDecoderInnerProd (:268-282), GraphDecoderDistMult (:285-305), DistMult.call (:347-354, one channel
// per call).  Tiny per-graph contractions (N <= ~50 nodes): one workgroup stages the graph's x tile in LDS
// and loops over graphs; backward  dx = w * ((g + g^T) x),  dw[k] = 1/2 sum_i x[i,k] ((g + g^T) x)[i,k]
// with one partial per workgroup (deterministic second stage).
// ------------------------------------------------------------------------------------------------
namespace kgcn {
int launch_reduce_partials(const float* part, int nparts, long n, float* out, hipStream_t s);

__global__ __launch_bounds__(256) void gram_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        float* __restrict__ out, int T, int N, int D) {
  extern __shared__ float gsm[];
  float* xs = gsm;                 // [N][D + 1]
  float* ws = gsm + (size_t)N * (D + 1);
  const int ld = D + 1;
  for (int k = threadIdx.x; k < D; k += 256) ws[k] = w ? w[k] : 1.f;
  for (int t = blockIdx.x; t < T; t += gridDim.x) {
    __syncthreads();
    const float* xt = x + (long)t * N * D;
    for (int e = threadIdx.x; e < N * D; e += 256) xs[(e / D) * ld + e % D] = xt[e];
    __syncthreads();
    float* ot = out + (long)t * N * N;
    for (int p = threadIdx.x; p < N * N; p += 256) {
      const int i = p / N, j = p - i * N;
      const float* a = xs + i * ld;
      const float* b = xs + j * ld;
      float s = 0.f;
      for (int k = 0; k < D; ++k) s = __builtin_fmaf(ws[k] * a[k], b[k], s);
      ot[p] = s;
    }
  }
}

__global__ __launch_bounds__(256) void gram_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ g, float* __restrict__ dx,
                                                        float* __restrict__ part_dw, int T, int N, int D, float beta) {
  extern __shared__ float gsm[];
  float* xs = gsm;                               // [N][D + 1]
  float* gs = gsm + (size_t)N * (D + 1);         // [N][N + 1]: g + g^T
  float* dws = gs + (size_t)N * (N + 1);         // [D] partial of this workgroup
  const int ld = D + 1, lg = N + 1;
  for (int k = threadIdx.x; k < D; k += 256) dws[k] = 0.f;
  for (int t = blockIdx.x; t < T; t += gridDim.x) {
    __syncthreads();
    const float* xt = x + (long)t * N * D;
    const float* gt = g + (long)t * N * N;
    for (int e = threadIdx.x; e < N * D; e += 256) xs[(e / D) * ld + e % D] = xt[e];
    for (int p = threadIdx.x; p < N * N; p += 256) {
      const int i = p / N, j = p - i * N;
      gs[i * lg + j] = gt[p] + gt[j * N + i];
    }
    __syncthreads();
    float* dxt = dx + (long)t * N * D;
    // thread <-> column k, all rows in order: the dw partial of a column is a fixed-order sum in ONE thread
    // (deterministic, no atomics); g + g^T is read as a broadcast, x along rows without bank conflicts
    for (int k = threadIdx.x; k < D; k += 256) {
      const float wk = w ? w[k] : 1.f;
      float dwk = 0.f;
      for (int i = 0; i < N; ++i) {
        float h = 0.f;
        for (int j = 0; j < N; ++j) h = __builtin_fmaf(gs[i * lg + j], xs[j * ld + k], h);
        float* o = dxt + i * D + k;
        *o = (beta != 0.f ? *o : 0.f) + wk * h;
        dwk = __builtin_fmaf(0.5f * xs[i * ld + k], h, dwk);
      }
      dws[k] += dwk;
    }
  }
  __syncthreads();
  if (part_dw)
    for (int k = threadIdx.x; k < D; k += 256) part_dw[(long)blockIdx.x * D + k] = dws[k];
}
}  // namespace kgcn

extern "C" int64_t kgcn_gram_workspace_bytes(int32_t d) { return d > 0 ? (int64_t)512 * d * 4 : 0; }

static size_t gram_lds(int N, int D, bool bwd) {
  return ((size_t)N * (D + 1) + (bwd ? (size_t)N * (N + 1) + D : (size_t)D)) * 4;
}

extern "C" int kgcn_gram_fwd_f32(const float* x, int32_t num_graphs, int32_t n_nodes, int32_t d, const float* w,
                                 float* out, void* stream) {
  if (num_graphs < 0 || n_nodes <= 0 || d <= 0) return kgcn::fail("kgcn_gram_fwd_f32: bad shape");
  if (num_graphs == 0) return 0;
  if (!x || !out) return kgcn::fail("kgcn_gram_fwd_f32: NULL operand");
  const size_t lds = gram_lds(n_nodes, d, false);
  if (lds > (size_t)kgcn::kLdsBytes) return kgcn::fail("kgcn_gram_fwd_f32: graph tile %d x %d does not fit LDS", n_nodes, d);
  static thread_local bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kgcn::gram_fwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kgcn::kLdsBytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kgcn::gram_bwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kgcn::kLdsBytes);
    attr_set = true;
  }
  const int blocks = num_graphs < 2048 ? num_graphs : 2048;
  hipLaunchKernelGGL(kgcn::gram_fwd_kernel, dim3(blocks), dim3(256), lds, kgcn::as_stream(stream), x, w, out, num_graphs,
                     n_nodes, d);
  return kgcn::check_launch("gram_fwd_kernel");
}

extern "C" int kgcn_gram_bwd_f32(const float* x, int32_t num_graphs, int32_t n_nodes, int32_t d, const float* w,
                                 const float* dout_grad, float* dx, float beta, float* dw, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
  if (num_graphs < 0 || n_nodes <= 0 || d <= 0) return kgcn::fail("kgcn_gram_bwd_f32: bad shape");
  hipStream_t s = kgcn::as_stream(stream);
  if (num_graphs == 0) {
    if (dw) (void)hipMemsetAsync(dw, 0, (size_t)d * 4, s);
    return 0;
  }
  if (!x || !dout_grad || !dx) return kgcn::fail("kgcn_gram_bwd_f32: NULL operand");
  if (beta != 0.f && beta != 1.f) return kgcn::fail("kgcn_gram_bwd_f32: beta must be 0 or 1");
  const size_t lds = gram_lds(n_nodes, d, true);
  if (lds > (size_t)kgcn::kLdsBytes) return kgcn::fail("kgcn_gram_bwd_f32: graph tile %d x %d does not fit LDS", n_nodes, d);
  const int blocks = num_graphs < 512 ? num_graphs : 512;
  if (dw && (!workspace || workspace_bytes < (int64_t)blocks * d * 4))
    return kgcn::fail("kgcn_gram_bwd_f32: workspace %lld < %lld bytes", (long long)workspace_bytes,
                      (long long)blocks * d * 4);
  static thread_local bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kgcn::gram_bwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kgcn::kLdsBytes);
    attr_set = true;
  }
  float* part = dw ? static_cast<float*>(workspace) : nullptr;
  hipLaunchKernelGGL(kgcn::gram_bwd_kernel, dim3(blocks), dim3(256), lds, s, x, w, dout_grad, dx, part, num_graphs,
                     n_nodes, d, beta);
  if (int rc = kgcn::check_launch("gram_bwd_kernel")) return rc;
  return dw ? kgcn::launch_reduce_partials(part, blocks, d, dw, s) : 0;
}
