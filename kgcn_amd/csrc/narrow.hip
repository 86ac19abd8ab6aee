// Narrow dense layers (din <= 64 and dout <= 64 whose rows are NOT a whole number of float4: the 50-wide layers of
// example_model/model_multitask.py:57-62, GraphDense 50 -> 50, and the 12-task read-out) as HBM streams.
//
//   y[m, dout] = act(x[m, din] @ W + bias)  /  dx = dy @ W^T          narrow_fwd_kernel
//   dW = x^T @ dy, dbias = colsum(dy)  (per-workgroup partials)        narrow_wgrad_kernel
//
// These layers move 200-byte rows.  The row-wise kernels of dense.hip fetch such a row with one 4-byte load per lane
// (64 lanes -> one row: 256 bytes per wave instruction) and store y the same way; measured 0.2 of the HBM rate at
// m = 204,800.  A [32 rows x d] tile of a dense matrix (ld == d) is, however, ONE contiguous 128 d-byte block whose start
// is 16-byte aligned for every d, so here a tile moves as flat dwordx4 (1 KiB per wave instruction) into the wave's LDS
// tile, keeps its row stride d there (odd / 2-mod-4 strides are conflict-free for the MFMA operand reads), and y leaves
// the same way: accumulators -> LDS tile [32 x dout] -> flat dwordx4 stores.
// The contraction itself stays on v_mfma_f32_32x32x2_f32 (exact fp32 products): <= 2 x 32 MFMAs of 64 cycles per 32-row
// tile is below the tile's HBM time.
#include "kgcn_common.h"

namespace kgcn {

constexpr int NR_WAVES = 8;
constexpr int NR_TILE = 32 * 64 + 64;      // floats of one wave's LDS tile (+ pad: operand reads of unused lanes run past)

struct FlatTile { f32x4 v[8]; };           // 32 rows x <= 64 floats = <= 512 float4 over 64 lanes

// rows [row0, row0 + 32) of a dense [m x d] matrix with ld == d; rows beyond m arrive as zeros (only the one float4 that
// straddles the end of the matrix is fetched float by float)
__device__ __forceinline__ void flat_issue(FlatTile& f, const float* __restrict__ base, long m, int d, long row0,
                                           int lane) {
  const float* tb = base + row0 * d;
  const int rows = m - row0 < 32 ? (int)(m - row0) : 32;
  const int cnt = rows * d;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int i4 = 4 * (lane + 64 * q);
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (i4 + 3 < cnt) {
      z = *reinterpret_cast<const f32x4*>(tb + i4);
    } else if (i4 < cnt) {
      z[0] = tb[i4];
      if (i4 + 1 < cnt) z[1] = tb[i4 + 1];
      if (i4 + 2 < cnt) z[2] = tb[i4 + 2];
    }
    f.v[q] = z;
  }
}

__device__ __forceinline__ void flat_land(const FlatTile& f, float* tile, int d, int lane) {
  const int n4 = 8 * d;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int i = lane + 64 * q;
    if (i < n4) *reinterpret_cast<f32x4*>(tile + 4 * i) = f.v[q];
  }
}

__device__ __forceinline__ void nr_handoff() {   // intra-wave LDS hand-off: compiler barrier only
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// One workgroup (8 waves) per CU and launch, every wave owns whole 32-row tiles; the weight panel [din x 64] (zero
// padded) stays in LDS.  MFMA k index of lane half hi, step s: k = hi * kh + s with kh = ceil(din / 2) -- both halves
// walk contiguous floats of their row.
#ifdef NR_WPE4
__attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
__global__ __launch_bounds__(512) void narrow_fwd_kernel(
    const float* __restrict__ x, long m, int din, const float* __restrict__ w, long w_ld, int trans_w,
    const float* __restrict__ bias, float* __restrict__ y, int dout, int act, int tile_floats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  float* Wp = reinterpret_cast<float*>(dsm);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, hi = lane >> 5;
  const int kh = (din + 1) >> 1;
  float* ts = Wp + (size_t)2 * kh * 64 + (size_t)wave * tile_floats;
  for (int i = tid; i < 2 * kh * 64; i += blockDim.x) {
    const int k = i >> 6, j = i & 63;
    float v = 0.f;
    if (k < din && j < dout) v = trans_w ? w[(long)j * w_ld + k] : w[(long)k * w_ld + j];
    Wp[i] = v;
  }
  __syncthreads();
  const float b0 = (bias && li < dout) ? bias[li] : 0.f;
  const float b1 = (bias && 32 + li < dout) ? bias[32 + li] : 0.f;
  const bool two = dout > 32;

  const long ntiles = (m + 31) / 32;
  const long nwaves = (long)gridDim.x * NR_WAVES;
  long tile = (long)blockIdx.x * NR_WAVES + wave;
  if (tile >= ntiles) return;
  FlatTile fx;
  flat_issue(fx, x, m, din, tile * 32, lane);
  const float* xa = ts + li * din + hi * kh;
  const float* wb = Wp + (size_t)hi * kh * 64 + li;
  const int kvalid = din - hi * kh;                 // steps s < kvalid carry a real k
  for (;;) {
    flat_land(fx, ts, din, lane);
    nr_handoff();
    const long tn = tile + nwaves;
    const bool more = tn < ntiles;
    flat_issue(fx, x, m, din, (more ? tn : tile) * 32, lane);     // the wave's next tile flies during the MFMAs
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = b0; acc1[r] = b1; }
#ifndef NR_ABL_NOMFMA
    if (two) {
      for (int s = 0; s < kh; ++s) {
        float a = xa[s];
        a = s < kvalid ? a : 0.f;                   // k == din of an odd din: the neighbour row's float, not ours
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb[s * 64], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb[s * 64 + 32], acc1, 0, 0, 0);
      }
    } else {
      for (int s = 0; s < kh; ++s) {
        float a = xa[s];
        a = s < kvalid ? a : 0.f;
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb[s * 64], acc0, 0, 0, 0);
      }
    }
#endif
    nr_handoff();
    if (act != KGCN_ACT_NONE) {                     // one uniform branch around the whole activation
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = act_fwd(acc0[r], act); acc1[r] = act_fwd(acc1[r], act); }
    }
    // accumulators -> the wave's tile as [32 x dout] (lane = column: consecutive banks)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (li < dout) ts[row * dout + li] = acc0[r];
      if (32 + li < dout) ts[row * dout + 32 + li] = acc1[r];
    }
    nr_handoff();
    {
      const long row0 = tile * 32;
      const long rows = m - row0 < 32 ? m - row0 : 32;
      const long cnt = rows * dout;
      float* dst = y + row0 * dout;
      const int n4 = 8 * dout;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = lane + 64 * q;
        if (i < n4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(ts + 4 * i);
          if (4L * i + 3 < cnt) {
            *reinterpret_cast<f32x4*>(dst + 4 * i) = v;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (4L * i + j < cnt) dst[4 * i + j] = v[j];
          }
        }
      }
    }
    nr_handoff();
    if (!more) break;
    tile = tn;
  }
}

// dW / dbias partials of one workgroup: rows are the MFMA K (k = hi * 16 + s inside a 32-row tile), every wave keeps the
// four 32 x 32 blocks of dW over its whole row range, the 8 waves are summed through LDS in a fixed order.
__global__ __launch_bounds__(512, 1) void narrow_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, long m, int din, int dout,
    float* __restrict__ part_dw, float* __restrict__ part_db) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, hi = lane >> 5;
  float* xs = reinterpret_cast<float*>(dsm) + (size_t)wave * 2 * NR_TILE;
  float* gs = xs + NR_TILE;
  const bool i2 = din > 32, j2 = dout > 32;

  f32x16 d00, d01, d10, d11;
#pragma unroll
  for (int r = 0; r < 16; ++r) { d00[r] = 0.f; d01[r] = 0.f; d10[r] = 0.f; d11[r] = 0.f; }
  float cs0 = 0.f, cs1 = 0.f;

  const long ntiles = (m + 31) / 32;
  const long nwaves = (long)gridDim.x * NR_WAVES;
  long tile = (long)blockIdx.x * NR_WAVES + wave;
  if (tile < ntiles) {
    FlatTile fx, fg;
    flat_issue(fx, x, m, din, tile * 32, lane);
    flat_issue(fg, dy, m, dout, tile * 32, lane);
    for (;;) {
      flat_land(fx, xs, din, lane);
      flat_land(fg, gs, dout, lane);
      nr_handoff();
      const long tn = tile + nwaves;
      const bool more = tn < ntiles;
      flat_issue(fx, x, m, din, (more ? tn : tile) * 32, lane);
      flat_issue(fg, dy, m, dout, (more ? tn : tile) * 32, lane);
      // lanes with li (+32) beyond din / dout read a neighbour's floats: they only reach dW rows / columns nobody stores
      const float* xa = xs + hi * 16 * din + li;
      const float* ga = gs + hi * 16 * dout + li;
#pragma unroll 4
      for (int s = 0; s < 16; ++s) {
        const float a0 = xa[s * din], f0 = ga[s * dout];
        d00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, f0, d00, 0, 0, 0);
        cs0 += f0;
        if (j2) {
          const float f1 = ga[s * dout + 32];
          d01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, f1, d01, 0, 0, 0);
          cs1 += f1;
          if (i2) d11 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[s * din + 32], f1, d11, 0, 0, 0);
        }
        if (i2) d10 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[s * din + 32], f0, d10, 0, 0, 0);
      }
      nr_handoff();
      if (!more) break;
      tile = tn;
    }
  }
  // the 8 waves' blocks -> LDS, summed in wave order
  __syncthreads();
  float* park = reinterpret_cast<float*>(dsm) + (size_t)wave * (64 * 64 + 64);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    park[row * 64 + li] = d00[r];
    park[row * 64 + 32 + li] = d01[r];
    park[(32 + row) * 64 + li] = d10[r];
    park[(32 + row) * 64 + 32 + li] = d11[r];
  }
  cs0 += __shfl_xor(cs0, 32, 64);
  cs1 += __shfl_xor(cs1, 32, 64);
  if (hi == 0) { park[64 * 64 + li] = cs0; park[64 * 64 + 32 + li] = cs1; }
  __syncthreads();
  const float* base = reinterpret_cast<const float*>(dsm);
  float* pw = part_dw + (long)blockIdx.x * din * dout;
  for (int i = tid; i < 64 * 64; i += blockDim.x) {
    float sum = 0.f;
#pragma unroll
    for (int wv = 0; wv < NR_WAVES; ++wv) sum += base[(size_t)wv * (64 * 64 + 64) + i];
    const int row = i >> 6, col = i & 63;
    if (row < din && col < dout) pw[(long)row * dout + col] = sum;
  }
  if (part_db && tid < 64) {
    float sum = 0.f;
#pragma unroll
    for (int wv = 0; wv < NR_WAVES; ++wv) sum += base[(size_t)wv * (64 * 64 + 64) + 64 * 64 + tid];
    if (tid < dout) part_db[(long)blockIdx.x * dout + tid] = sum;
  }
}

// ---- host side ------------------------------------------------------------------------------------------------
bool narrow_fwd_ok(const float* x, int din, long x_ld, const float* y, int dout, long y_ld) {
  return din <= 64 && dout <= 64 && x_ld == din && y_ld == dout && aligned16(x) && aligned16(y) && din % 4 != 0;
}

int launch_narrow_fwd(const float* x, long m, int din, const float* w, long w_ld, int trans_w, const float* bias,
                      float* y, int dout, int act, hipStream_t s) {
  const int kh = (din + 1) / 2;
  const int tile_floats = 32 * (din > dout ? din : dout) + 64;   // x tile, then y tile (+ pad, see NR_TILE); 16-byte multiple
  const size_t lds = ((size_t)2 * kh * 64 + (size_t)NR_WAVES * tile_floats) * 4;
  static thread_local bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(narrow_fwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    attr_set = true;
  }
  const long tiles = (m + 31) / 32;
  long blocks = (tiles + NR_WAVES - 1) / NR_WAVES;
#ifdef NR_WPE4
  if (blocks > 2 * kNumCU) blocks = 2 * kNumCU;
#else
  if (blocks > kNumCU) blocks = kNumCU;              // 256 VGPRs: one workgroup per CU
#endif
  hipLaunchKernelGGL(narrow_fwd_kernel, dim3((unsigned)blocks), dim3(64 * NR_WAVES), lds, s, x, m, din, w, w_ld,
                     trans_w, bias, y, dout, act, tile_floats);
  return check_launch("narrow_fwd_kernel");
}

bool narrow_wgrad_ok(const float* x, int din, long x_ld, const float* dy, int dout, long dy_ld) {
  return din <= 64 && dout <= 64 && x_ld == din && dy_ld == dout && aligned16(x) && aligned16(dy) &&
         (din % 4 != 0 || dout % 4 != 0);
}

// nblocks partials ([nblocks][din*dout], [nblocks][dout]); nblocks <= kNumCU
int launch_narrow_wgrad(const float* x, const float* dy, long m, int din, int dout, float* part_dw, float* part_db,
                        int nblocks, hipStream_t s) {
  const size_t tiles_b = (size_t)NR_WAVES * 2 * NR_TILE * 4, park_b = (size_t)NR_WAVES * (64 * 64 + 64) * 4;
  const size_t lds = tiles_b > park_b ? tiles_b : park_b;
  static thread_local bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(narrow_wgrad_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    attr_set = true;
  }
  hipLaunchKernelGGL(narrow_wgrad_kernel, dim3((unsigned)nblocks), dim3(64 * NR_WAVES), lds, s, x, dy, m, din, dout,
                     part_dw, part_db);
  return check_launch("narrow_wgrad_kernel");
}

}  // namespace kgcn
