// Fused GraphConv layer for small molecular graphs on gfx950:
//
//   forward   out[t] = A[t] @ (x[t] @ W + bias)                     kgcn/layers.py:64-116
//   backward  dfw[t] = A[t]^T @ g[t];  dx[t] = dfw[t] @ W^T;
//             dW = sum_t x[t]^T dfw[t];  dbias = sum_t colsum(dfw[t])   SURVEY 3.3, bspmm_call.py:45
//
// The reference materialises FW = X.W + b per graph and channel (B*C tiny MatMul ops) and then
// aggregates it with B*C sparse ops.  Here one persistent wave owns one graph at a time:
//   * the graph's node-feature tile [N<=32 x D<=64] is streamed HBM -> LDS once (dwordx4),
//   * the dense contraction runs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32)
//     with the weight fragments resident in registers (forward) / in LDS (backward, W^T),
//   * FW (resp. dFW) lives only in LDS, where the sparse aggregation gathers neighbour rows with
//     conflict-free ds_read_b128 -- X.W and dFW never touch HBM,
//   * dW / dbias accumulate in MFMA accumulators across ALL graphs a wave processes and leave the
//     chip once per wave (deterministic second-stage reduction).
// Algorithmic HBM bytes per graph (N=32, D=64, nnz=100): forward 17,316, backward 25,508.
//
// Shape support (kgcn_graphconv_fused_supported): N <= 32, din,dout <= 64 and multiples of 4.
// Everything else goes through kgcn_dense_* + kgcn_bconv_f32.
#include "kgcn_common.h"

namespace kgcn {

int launch_reduce_partials(const float* part, int nparts, long n, float* out, hipStream_t s);

constexpr int FN = 32;    // node tile (MFMA M)
constexpr int FD = 64;    // feature tile (K of the forward GEMM, two 32-wide output tiles)
constexpr int ALD = 68;   // padded row stride (floats) of tiles read as MFMA A fragments (b128)
constexpr int MAX_WPB = 8;

__device__ __forceinline__ void wave_sync() {
  // LDS operations of one wave execute in order; this only stops the compiler from moving LDS
  // accesses of different lanes across the hand-off point.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ f32x4 lds4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

struct WaveSlice {
  float* a;     // [FN][ALD]  A-fragment source tile
  float* b;     // [FN][FD]   gather source tile
  int2* ecv;    // [max_nnz]
  int* rp;      // [FN + 4]
};

__host__ __device__ inline size_t slice_bytes(int max_nnz) {
  size_t e = ((size_t)max_nnz * 8 + 15) & ~(size_t)15;
  return (size_t)FN * ALD * 4 + (size_t)FN * FD * 4 + e + (FN + 4) * 4;
}

__device__ __forceinline__ WaveSlice carve(unsigned char* base, int wave, int max_nnz) {
  unsigned char* p = base + (size_t)wave * slice_bytes(max_nnz);
  WaveSlice s;
  s.a = reinterpret_cast<float*>(p);
  s.b = s.a + FN * ALD;
  s.ecv = reinterpret_cast<int2*>(s.b + FN * FD);
  s.rp = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(s.ecv) +
                                (((size_t)max_nnz * 8 + 15) & ~(size_t)15));
  return s;
}

// stage one graph's CSR slice (rowptr rebased to 0, interleaved col/val pairs) into LDS
__device__ __forceinline__ void stage_csr(const int* __restrict__ rowptr,
                                          const int2* __restrict__ cv, int t, int N, int lane,
                                          const WaveSlice& ws) {
  const int* grp = rowptr + (long)t * N;
  const int base = grp[0];
  const int cnt = grp[N] - base;
  for (int i = lane; i < cnt; i += 64) ws.ecv[i] = cv[base + i];
  if (lane <= N) ws.rp[lane] = grp[lane] - base;
}

// out rows r0..r0+3 (16 lanes x float4 each) = sum of gathered rows of `src` (row stride FD)
template <typename Sink>
__device__ __forceinline__ void aggregate_rows(const WaveSlice& ws, const float* src, int N,
                                               int dcols, int lane, Sink&& sink) {
  const int sub = lane >> 4, cl = lane & 15;
  const bool col_ok = cl * 4 < dcols;
  for (int r0 = 0; r0 < N; r0 += 4) {
    const int r = r0 + sub;
    if (r < N && col_ok) {
      const int s = ws.rp[r], e = ws.rp[r + 1];
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int k = s; k < e; ++k) {
        const int2 p = ws.ecv[k];
        acc += __int_as_float(p.y) * lds4(src + p.x * FD + cl * 4);
      }
      sink(r, cl, acc);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void graphconv_fwd_kernel(
    const int* __restrict__ rowptr, const int2* __restrict__ cv, const float* __restrict__ x,
    const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ out, int T,
    int N, int din, int dout, int max_nnz) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, hi = lane >> 5;
  const int wpb = blockDim.x >> 6;
  WaveSlice ws = carve(smem, wave, max_nnz);

  // zero the A tile once: padding rows (>= N) and columns (>= din) stay zero for every graph
  for (int i = lane; i < FN * ALD; i += 64) ws.a[i] = 0.f;

  // weight fragments: B[k][j] with k = hi*32 + s (the K permutation matches the A fragments)
  float wr0[32], wr1[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    const int k = hi * 32 + s;
    wr0[s] = (k < din && li < dout) ? w[(long)k * dout + li] : 0.f;
    wr1[s] = (k < din && 32 + li < dout) ? w[(long)k * dout + 32 + li] : 0.f;
  }
  const float b0 = (bias && li < dout) ? bias[li] : 0.f;
  const float b1 = (bias && 32 + li < dout) ? bias[32 + li] : 0.f;
  wave_sync();

  const int din4 = din >> 2;
  const int n4 = N * din4;
  const int nwaves = gridDim.x * wpb;
  for (int t = blockIdx.x * wpb + wave; t < T; t += nwaves) {
    // ---- stage x[t] (contiguous N*din floats) into the padded A tile, and the CSR slice ------
    const float* xt = x + (long)t * N * din;
    for (int i = lane; i < n4; i += 64) {
      const int r = i / din4, c4 = i - r * din4;
      *reinterpret_cast<f32x4*>(ws.a + r * ALD + c4 * 4) = lds4(xt + (long)i * 4);
    }
    stage_csr(rowptr, cv, t, N, lane, ws);
    wave_sync();

    // ---- FW = x @ W + bias on the matrix cores ------------------------------------------------
    f32x4 a4[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a4[q] = lds4(ws.a + li * ALD + hi * 32 + q * 4);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = b0; acc1[r] = b1; }
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const float a = a4[s >> 2][s & 3];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wr0[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wr1[s], acc1, 0, 0, 0);
    }
    // C layout -> LDS gather tile (bank = column: conflict free)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
      ws.b[row * FD + li] = acc0[r];
      ws.b[row * FD + 32 + li] = acc1[r];
    }
    wave_sync();

    // ---- out[t] = A[t] @ FW : 4 rows per pass, 1 KiB coalesced store per pass ---------------
    float* ot = out + (long)t * N * dout;
    aggregate_rows(ws, ws.b, N, dout, lane, [&](int r, int cl, f32x4 acc) {
      *reinterpret_cast<f32x4*>(ot + (long)r * dout + cl * 4) = acc;
    });
    wave_sync();
  }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void graphconv_bwd_kernel(
    const int* __restrict__ rowptr_t, const int2* __restrict__ cv_t, const float* __restrict__ x,
    const float* __restrict__ w, const float* __restrict__ g, float* __restrict__ dx,
    float* __restrict__ part_dw, float* __restrict__ part_db, int T, int N, int din, int dout,
    int max_nnz) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, hi = lane >> 5;
  const int wpb = blockDim.x >> 6;
  float* Wt = reinterpret_cast<float*>(smem);  // [FD][FD]: Wt[k][j] = W[j][k] (zero padded)
  WaveSlice ws = carve(smem + FD * FD * 4, wave, max_nnz);

  for (int i = tid; i < FD * FD; i += blockDim.x) {
    const int k = i >> 6, j = i & 63;  // coalesced over k for fixed j would be strided; W is tiny
    Wt[i] = (j < din && k < dout) ? w[(long)j * dout + k] : 0.f;
  }
  for (int i = lane; i < FN * ALD; i += 64) ws.a[i] = 0.f;
  for (int i = lane; i < FN * FD; i += 64) ws.b[i] = 0.f;
  __syncthreads();

  f32x16 dw00, dw01, dw10, dw11;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dw00[r] = 0.f; dw01[r] = 0.f; dw10[r] = 0.f; dw11[r] = 0.f; }
  f32x4 dbacc = {0.f, 0.f, 0.f, 0.f};

  const int din4 = din >> 2, dout4 = dout >> 2;
  const int nx4 = N * din4, ng4 = N * dout4;
  const int nwaves = gridDim.x * wpb;
  for (int t = blockIdx.x * wpb + wave; t < T; t += nwaves) {
    // ---- stage g[t] into the gather tile, CSR(A^T) slice ------------------------------------
    const float* gt = g + (long)t * N * dout;
    for (int i = lane; i < ng4; i += 64) {
      const int r = i / dout4, c4 = i - r * dout4;
      *reinterpret_cast<f32x4*>(ws.b + r * FD + c4 * 4) = lds4(gt + (long)i * 4);
    }
    stage_csr(rowptr_t, cv_t, t, N, lane, ws);
    wave_sync();

    // ---- dFW = A^T @ g -> A tile (padded stride), dbias partial ------------------------------
    aggregate_rows(ws, ws.b, N, dout, lane, [&](int r, int cl, f32x4 acc) {
      *reinterpret_cast<f32x4*>(ws.a + r * ALD + cl * 4) = acc;
      dbacc += acc;
    });
    wave_sync();

    // ---- x[t] on its way (registers) while dX runs ---------------------------------------------
    const float* xt = x + (long)t * N * din;
    f32x4 xpf[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = lane + q * 64;
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      xpf[q] = (i < nx4) ? lds4(xt + (long)i * 4) : z;
    }

    // ---- dX = dFW @ W^T ------------------------------------------------------------------------
    if (dx) {
      f32x4 a4[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) a4[q] = lds4(ws.a + li * ALD + hi * 32 + q * 4);
      f32x16 c0, c1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        const float a = a4[s >> 2][s & 3];
        const int k = hi * 32 + s;
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Wt[k * FD + li], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Wt[k * FD + 32 + li], c1, 0, 0, 0);
      }
      float* dxt = dx + (long)t * N * din;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (row < N) {
          if (li < din) dxt[(long)row * din + li] = c0[r];
          if (32 + li < din) dxt[(long)row * din + 32 + li] = c1[r];
        }
      }
    }

    // ---- x[t] -> gather tile (g is dead), then dW += x^T @ dFW ---------------------------------
    wave_sync();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = lane + q * 64;
      if (i < nx4) {
        const int r = i / din4, c4 = i - r * din4;
        *reinterpret_cast<f32x4*>(ws.b + r * FD + c4 * 4) = xpf[q];
      }
    }
    wave_sync();
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int n = hi * 16 + s;
      const float a0 = ws.b[n * FD + li], a1 = ws.b[n * FD + 32 + li];
      const float f0 = ws.a[n * ALD + li], f1 = ws.a[n * ALD + 32 + li];
      dw00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, f0, dw00, 0, 0, 0);
      dw01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, f1, dw01, 0, 0, 0);
      dw10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, f0, dw10, 0, 0, 0);
      dw11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, f1, dw11, 0, 0, 0);
    }
    wave_sync();
  }

  // ---- per-wave partials leave the chip once -----------------------------------------------------
  const int gw = blockIdx.x * wpb + wave;
  float* pw = part_dw + (long)gw * din * dout;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    if (row < din) {
      if (li < dout) pw[(long)row * dout + li] = dw00[r];
      if (32 + li < dout) pw[(long)row * dout + 32 + li] = dw01[r];
    }
    if (32 + row < din) {
      if (li < dout) pw[(long)(32 + row) * dout + li] = dw10[r];
      if (32 + li < dout) pw[(long)(32 + row) * dout + 32 + li] = dw11[r];
    }
  }
  // dbacc: lane (sub, cl) holds the column-4-group cl summed over rows == sub (mod 4)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float v = dbacc[j];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    dbacc[j] = v;
  }
  if (lane < 16 && lane * 4 < dout) {
    float* pb = part_db + (long)gw * dout + lane * 4;
    pb[0] = dbacc[0]; pb[1] = dbacc[1]; pb[2] = dbacc[2]; pb[3] = dbacc[3];
  }
}

// waves per block that fit the LDS budget (0 = does not fit at all)
static int fused_wpb(int max_nnz, size_t shared_bytes) {
  const size_t per = slice_bytes(max_nnz);
  long fit = ((long)kLdsBytes - (long)shared_bytes) / (long)per;
  if (fit > MAX_WPB) fit = MAX_WPB;
  return fit < 1 ? 0 : (int)fit;
}

static int fused_grid(int T, int wpb) {
  int blocks = (T + wpb - 1) / wpb;
  if (blocks > kNumCU) blocks = kNumCU;  // one persistent workgroup per CU
  return blocks < 1 ? 1 : blocks;
}

static bool fused_shape_ok(int n, int din, int dout, int max_nnz) {
  if (n <= 0 || n > FN) return false;
  if (din <= 0 || din > FD || (din & 3)) return false;
  if (dout <= 0 || dout > FD || (dout & 3)) return false;
  if (max_nnz < 0) return false;
  return fused_wpb(max_nnz, FD * FD * 4) >= 4;
}

}  // namespace kgcn

using namespace kgcn;

extern "C" int kgcn_graphconv_fused_supported(int32_t n_nodes, int32_t din, int32_t dout,
                                              int32_t max_nnz_per_graph) {
  return fused_shape_ok(n_nodes, din, dout, max_nnz_per_graph) ? 1 : 0;
}

extern "C" int kgcn_graphconv_fwd_f32(const kgcn_csr_batch* a, const float* x, const float* w,
                                      const float* bias, int32_t din, int32_t dout, float* out,
                                      void* stream) {
  if (int rc = validate_csr(a, "kgcn_graphconv_fwd_f32")) return rc;
  if (a->rows != a->cols) return fail("kgcn_graphconv_fwd_f32: adjacency must be square");
  if (!fused_shape_ok(a->rows, din, dout, a->max_nnz_per_graph))
    return fail("kgcn_graphconv_fwd_f32: shape N=%d din=%d dout=%d max_nnz=%d not supported by the "
                "fused kernel (use kgcn_dense_fwd_f32 + kgcn_bconv_f32)",
                a->rows, din, dout, a->max_nnz_per_graph);
  if (a->num_graphs == 0) return 0;
  if (!x || !w || !out) return fail("kgcn_graphconv_fwd_f32: NULL operand");
  if (!aligned16(x) || !aligned16(out)) return fail("kgcn_graphconv_fwd_f32: x/out not 16-byte aligned");
  const int wpb = fused_wpb(a->max_nnz_per_graph, 0);
  const size_t lds = (size_t)wpb * slice_bytes(a->max_nnz_per_graph);
  static thread_local bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(graphconv_fwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    attr_set = true;
  }
  hipLaunchKernelGGL(graphconv_fwd_kernel, dim3(fused_grid(a->num_graphs, wpb)), dim3(64 * wpb),
                     lds, as_stream(stream), a->rowptr, reinterpret_cast<const int2*>(a->cv), x, w,
                     bias, out, a->num_graphs, a->rows, din, dout, a->max_nnz_per_graph);
  return check_launch("graphconv_fwd_kernel");
}

extern "C" int64_t kgcn_graphconv_bwd_workspace_bytes(int32_t num_graphs, int32_t din,
                                                      int32_t dout) {
  if (num_graphs <= 0 || din <= 0 || dout <= 0) return 0;
  // sized for the largest grid the launcher can pick (MAX_WPB waves x one workgroup per CU)
  return (int64_t)kNumCU * MAX_WPB * ((int64_t)din * dout + dout) * 4;
}

extern "C" int kgcn_graphconv_bwd_f32(const kgcn_csr_batch* at, const float* x, const float* w,
                                      const float* dout_grad, int32_t din, int32_t dout, float* dx,
                                      float* dw, float* dbias, void* workspace,
                                      int64_t workspace_bytes, void* stream) {
  if (int rc = validate_csr(at, "kgcn_graphconv_bwd_f32")) return rc;
  if (at->rows != at->cols) return fail("kgcn_graphconv_bwd_f32: adjacency must be square");
  if (!fused_shape_ok(at->rows, din, dout, at->max_nnz_per_graph))
    return fail("kgcn_graphconv_bwd_f32: shape N=%d din=%d dout=%d max_nnz=%d not supported by the "
                "fused kernel", at->rows, din, dout, at->max_nnz_per_graph);
  if (!dw || !dbias) return fail("kgcn_graphconv_bwd_f32: dw/dbias is NULL");
  hipStream_t s = as_stream(stream);
  if (at->num_graphs == 0) {
    (void)hipMemsetAsync(dw, 0, (size_t)din * dout * 4, s);
    (void)hipMemsetAsync(dbias, 0, (size_t)dout * 4, s);
    return 0;
  }
  if (!x || !w || !dout_grad) return fail("kgcn_graphconv_bwd_f32: NULL operand");
  if (!aligned16(x) || !aligned16(dout_grad) || (dx && !aligned16(dx)))
    return fail("kgcn_graphconv_bwd_f32: tensors not 16-byte aligned");
  const int wpb = fused_wpb(at->max_nnz_per_graph, FD * FD * 4);
  const int blocks = fused_grid(at->num_graphs, wpb);
  const int nparts = blocks * wpb;
  const int64_t need = (int64_t)nparts * ((int64_t)din * dout + dout) * 4;
  if (!workspace || workspace_bytes < need)
    return fail("kgcn_graphconv_bwd_f32: workspace %lld < %lld bytes", (long long)workspace_bytes,
                (long long)need);
  float* part_dw = static_cast<float*>(workspace);
  float* part_db = part_dw + (long)nparts * din * dout;
  const size_t lds = FD * FD * 4 + (size_t)wpb * slice_bytes(at->max_nnz_per_graph);
  static thread_local bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(graphconv_bwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    attr_set = true;
  }
  hipLaunchKernelGGL(graphconv_bwd_kernel, dim3(blocks), dim3(64 * wpb), lds, s, at->rowptr,
                     reinterpret_cast<const int2*>(at->cv), x, w, dout_grad, dx, part_dw, part_db,
                     at->num_graphs, at->rows, din, dout, at->max_nnz_per_graph);
  if (int rc = check_launch("graphconv_bwd_kernel")) return rc;
  if (int rc = launch_reduce_partials(part_dw, nparts, (long)din * dout, dw, s)) return rc;
  return launch_reduce_partials(part_db, nparts, dout, dbias, s);
}
