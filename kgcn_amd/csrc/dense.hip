// Dense feature x weight contraction on the fp32 matrix cores of gfx950.
//
//   y[m,dout] = x[m,din] @ W (+ bias)         kgcn/layers.py:99-100, :112 (X.W + b), :255-262 (GraphDense)
//   dx        = dy @ W^T                      (same kernel, trans_w = 1)
//   dW, db    = x^T @ dy, colsum(dy)          (TF MatMul / BiasAdd gradients)
//
// fp32 in, fp32 accumulate: v_mfma_f32_32x32x2_f32 is an exact k-ordered fmaf chain at the fp32
// vector rate (157 TF), so the 1e-5 parity budget is not spent on reduced-precision inputs
// (gfx950 has no xf32/TF32 path anyway).  m = B*N is huge and din/dout are small (3..256): a
// tall-skinny GEMM, HBM-bound for d <= 64 and MFMA-bound above.
//
// MFMA operand map used throughout (32x32x2, one VGPR per operand):
//   A: lane l holds A[i = l&31][k = l>>5];  B: lane l holds B[k = l>>5][j = l&31]
//   C/D: lane l, reg r holds C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
// The K index is permuted (half hi = l>>5 owns k in [16*hi, 16*hi+16) of every 32-wide chunk) so
// that a lane's 16 A values are contiguous in LDS (4 x ds_read_b128 instead of 16 x ds_read_b32);
// A and B use the same permutation, so the sum over k is unchanged.
#include "kgcn_common.h"

namespace kgcn {

constexpr int BM = 128;      // rows per workgroup (32 per wave)
constexpr int BN = 64;       // output columns per workgroup
constexpr int BK = 32;       // k chunk
constexpr int XS_LD = BK + 4;  // +16 B pad: conflict-free ds_read_b128 of A fragments

__global__ __launch_bounds__(256) void dense_fwd_kernel(
    const float* __restrict__ x, long m, int din, long x_ld, const float* __restrict__ w,
    long w_ld, int trans_w, const float* __restrict__ bias, float* __restrict__ y, int dout,
    long y_ld) {
  __shared__ __attribute__((aligned(16))) float Xs[BM * XS_LD];
  __shared__ __attribute__((aligned(16))) float Ws[BK * BN];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, hi = lane >> 5;
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const bool x_vec = ((din & 3) == 0) && ((x_ld & 3) == 0) &&
                     ((reinterpret_cast<uintptr_t>(x) & 15u) == 0);

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

  for (int k0 = 0; k0 < din; k0 += BK) {
    // ---- stage X chunk [BM x BK] (zero filled outside m / din) ------------------------------
    if (x_vec) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = tid + it * 256;  // 1024 float4
        const int r = idx >> 3, c4 = idx & 7;
        const long row = m0 + r;
        const int col = k0 + c4 * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < m && col < din) v = *reinterpret_cast<const f32x4*>(x + row * x_ld + col);
        *reinterpret_cast<f32x4*>(&Xs[r * XS_LD + c4 * 4]) = v;
      }
    } else {
#pragma unroll 4
      for (int it = 0; it < 16; ++it) {
        const int idx = tid + it * 256;  // 4096 floats
        const int r = idx >> 5, c = idx & 31;
        const long row = m0 + r;
        const int col = k0 + c;
        Xs[r * XS_LD + c] = (row < m && col < din) ? x[row * x_ld + col] : 0.f;
      }
    }
    // ---- stage W chunk [BK x BN] --------------------------------------------------------------
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = tid + it * 256;  // 2048 floats
      int k, j;
      if (trans_w) { k = idx & 31; j = idx >> 5; } else { k = idx >> 6; j = idx & 63; }
      const int kk = k0 + k, jj = n0 + j;
      float v = 0.f;
      if (kk < din && jj < dout) v = trans_w ? w[(long)jj * w_ld + kk] : w[(long)kk * w_ld + jj];
      Ws[k * BN + j] = v;
    }
    __syncthreads();

    // ---- 16 k-steps x 2 column tiles ---------------------------------------------------------
    const float* arow = &Xs[(wave * 32 + li) * XS_LD + hi * 16];
    f32x4 a4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a4[q] = *reinterpret_cast<const f32x4*>(arow + q * 4);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float a = a4[s >> 2][s & 3];
      const float b0 = Ws[(s + 16 * hi) * BN + li];
      const float b1 = Ws[(s + 16 * hi) * BN + 32 + li];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue: + bias, store (lanes 0..31 write 128 contiguous bytes of one row) ------------
  const int c0 = n0 + li, c1 = n0 + 32 + li;
  const float bv0 = (bias && c0 < dout) ? bias[c0] : 0.f;
  const float bv1 = (bias && c1 < dout) ? bias[c1] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const long row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    if (row < m) {
      if (c0 < dout) y[row * y_ld + c0] = acc0[r] + bv0;
      if (c1 < dout) y[row * y_ld + c1] = acc1[r] + bv1;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// dW / dbias partials: block (chunk of rows, 64-wide din block, 64-wide dout block); wave (ti,tj)
// owns one 32x32 tile of the 64x64 output block.  K dimension of the MFMA = the m rows.
// ------------------------------------------------------------------------------------------------
constexpr int WG_ROWS = 32;  // rows staged per step

__global__ __launch_bounds__(256) void dense_wgrad_kernel(
    const float* __restrict__ x, long x_ld, const float* __restrict__ dy, long dy_ld, long m,
    int din, int dout, long rows_per_chunk, float* __restrict__ part_dw,
    float* __restrict__ part_db) {
  __shared__ __attribute__((aligned(16))) float Xs[WG_ROWS * 64];
  __shared__ __attribute__((aligned(16))) float Gs[WG_ROWS * 64];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, hi = lane >> 5;
  const int ti = wave >> 1, tj = wave & 1;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.z * 64;
  const long r_begin = (long)blockIdx.x * rows_per_chunk;
  long r_end = r_begin + rows_per_chunk;
  if (r_end > m) r_end = m;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float colsum = 0.f;

  for (long r0 = r_begin; r0 < r_end; r0 += WG_ROWS) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = tid + it * 256;  // 2048 floats each
      const int rr = idx >> 6, c = idx & 63;
      const long row = r0 + rr;
      const bool rok = row < r_end;
      Xs[idx] = (rok && i0 + c < din) ? x[row * x_ld + i0 + c] : 0.f;
      Gs[idx] = (rok && j0 + c < dout) ? dy[row * dy_ld + j0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int rr = s + 16 * hi;
      const float a = Xs[rr * 64 + ti * 32 + li];   // A[i = din idx][k = row]
      const float b = Gs[rr * 64 + tj * 32 + li];   // B[k = row][j = dout idx]
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      colsum += b;
    }
    __syncthreads();
  }

  // partial dW tile
  float* pw = part_dw + (long)blockIdx.x * din * dout;
  const int col = j0 + tj * 32 + li;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = i0 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    if (row < din && col < dout) pw[(long)row * dout + col] = acc[r];
  }
  // partial dbias: only the blocks/waves of the first din block contribute
  if (part_db && blockIdx.y == 0 && ti == 0) {
    colsum += __shfl_xor(colsum, 32, 64);
    if (hi == 0 && col < dout) part_db[(long)blockIdx.x * dout + col] = colsum;
  }
}

// out[i] = sum_p part[p*n + i]   (deterministic second stage).  A workgroup owns 32 consecutive
// outputs (128 contiguous bytes per partial); its 8 lane-groups stride over the partials with 8
// independent loads in flight each, then combine through LDS in a fixed order.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part,
                                                              int nparts, long n,
                                                              float* __restrict__ out) {
  __shared__ float red[8][33];
  const int oi = threadIdx.x & 31, pg = threadIdx.x >> 5;
  const long o = (long)blockIdx.x * 32 + oi;
  float s[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) s[u] = 0.f;
  if (o < n) {
    int p = pg;
    for (; p + 56 < nparts; p += 64) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += part[(long)(p + 8 * u) * n + o];
    }
    for (; p < nparts; p += 8) s[0] += part[(long)p * n + o];
  }
  red[pg][oi] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (pg == 0 && o < n) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][oi];
    out[o] = t;
  }
}

static void wgrad_plan(long m, long* rows_per_chunk, int* nchunks) {
  // ~4 workgroups per CU worth of row chunks, each a multiple of WG_ROWS rows
  long target = (long)kNumCU * 4;
  long rpc = (m + target - 1) / target;
  rpc = ((rpc + WG_ROWS - 1) / WG_ROWS) * WG_ROWS;
  if (rpc < WG_ROWS) rpc = WG_ROWS;
  *rows_per_chunk = rpc;
  *nchunks = (int)((m + rpc - 1) / rpc);
  if (*nchunks < 1) *nchunks = 1;
}

int launch_reduce_partials(const float* part, int nparts, long n, float* out, hipStream_t s) {
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, s,
                     part, nparts, n, out);
  return check_launch("reduce_partials_kernel");
}

}  // namespace kgcn

using namespace kgcn;

extern "C" int kgcn_dense_fwd_f32(const float* x, int64_t m, int32_t din, int64_t x_ld,
                                  const float* w, int64_t w_ld, int32_t trans_w, const float* bias,
                                  float* y, int32_t dout, int64_t y_ld, void* stream) {
  if (m < 0 || din <= 0 || dout <= 0)
    return fail("kgcn_dense_fwd_f32: bad shape m=%lld din=%d dout=%d", (long long)m, din, dout);
  if (m == 0) return 0;
  if (!x || !w || !y) return fail("kgcn_dense_fwd_f32: NULL operand");
  if (x_ld < din || y_ld < dout) return fail("kgcn_dense_fwd_f32: leading dimension too small");
  if (w_ld < (trans_w ? din : dout)) return fail("kgcn_dense_fwd_f32: w_ld too small");
  const long gx = (m + BM - 1) / BM;
  if (gx > 0x7fffffffL) return fail("kgcn_dense_fwd_f32: m too large");
  dim3 grid((unsigned)gx, (unsigned)((dout + BN - 1) / BN));
  hipLaunchKernelGGL(dense_fwd_kernel, grid, dim3(256), 0, as_stream(stream), x, (long)m, din,
                     (long)x_ld, w, (long)w_ld, trans_w, bias, y, dout, (long)y_ld);
  return check_launch("dense_fwd_kernel");
}

extern "C" int64_t kgcn_dense_wgrad_workspace_bytes(int64_t m, int32_t din, int32_t dout) {
  if (m <= 0 || din <= 0 || dout <= 0) return 0;
  long rpc;
  int nchunks;
  wgrad_plan(m, &rpc, &nchunks);
  return (int64_t)nchunks * ((int64_t)din * dout + dout) * 4;
}

extern "C" int kgcn_dense_wgrad_f32(const float* x, int64_t x_ld, const float* dy, int64_t dy_ld,
                                    int64_t m, int32_t din, int32_t dout, float* dw, float* dbias,
                                    void* workspace, int64_t workspace_bytes, void* stream) {
  if (m < 0 || din <= 0 || dout <= 0)
    return fail("kgcn_dense_wgrad_f32: bad shape m=%lld din=%d dout=%d", (long long)m, din, dout);
  if (!dw && !dbias) return 0;
  hipStream_t s = as_stream(stream);
  if (m == 0) {
    if (dw) (void)hipMemsetAsync(dw, 0, (size_t)din * dout * 4, s);
    if (dbias) (void)hipMemsetAsync(dbias, 0, (size_t)dout * 4, s);
    return 0;
  }
  if (!x || !dy) return fail("kgcn_dense_wgrad_f32: NULL operand");
  const int64_t need = kgcn_dense_wgrad_workspace_bytes(m, din, dout);
  if (!workspace || workspace_bytes < need)
    return fail("kgcn_dense_wgrad_f32: workspace %lld < %lld bytes", (long long)workspace_bytes,
                (long long)need);
  long rpc;
  int nchunks;
  wgrad_plan(m, &rpc, &nchunks);
  float* part_dw = static_cast<float*>(workspace);
  float* part_db = part_dw + (long)nchunks * din * dout;
  dim3 grid((unsigned)nchunks, (unsigned)((din + 63) / 64), (unsigned)((dout + 63) / 64));
  hipLaunchKernelGGL(dense_wgrad_kernel, grid, dim3(256), 0, s, x, (long)x_ld, dy, (long)dy_ld,
                     (long)m, din, dout, rpc, part_dw, part_db);
  if (int rc = check_launch("dense_wgrad_kernel")) return rc;
  if (dw)
    if (int rc = launch_reduce_partials(part_dw, nchunks, (long)din * dout, dw, s)) return rc;
  if (dbias)
    if (int rc = launch_reduce_partials(part_db, nchunks, dout, dbias, s)) return rc;
  return 0;
}
