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
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "kgcn_common.h"

namespace kgcn {

int launch_gemm3_fwd(const float* x, long m, int din, long x_ld, const float* w, long w_ld, int trans_w,
                     const float* bias, float* y, int dout, long y_ld, int act, const void* table, hipStream_t s);
bool narrow_fwd_ok(const float* x, int din, long x_ld, const float* y, int dout, long y_ld);
int launch_narrow_fwd(const float* x, long m, int din, const float* w, long w_ld, int trans_w, const float* bias,
                      float* y, int dout, int act, hipStream_t s);
bool narrow_wgrad_ok(const float* x, int din, long x_ld, const float* dy, int dout, long dy_ld);
int launch_narrow_wgrad(const float* x, const float* dy, long m, int din, int dout, float* part_dw, float* part_db,
                        int nblocks, hipStream_t s);
int launch_gemm3_dx_dact(const float* grad, const float* act_out, float* dpre, long m, int k, long ld, const void* table,
                         float* dx, int n, long dx_ld, int dact, hipStream_t s, const float* pooled_grad = nullptr,
                         int n_nodes = 0, long pooled_ld = 0);
bool gemmn_pays(const float* x, int din, long x_ld, int dout);
int launch_gemmn_fwd(const float* x, long m, int din, long x_ld, const void* table, const float* bias, float* y, int dout,
                     long y_ld, int act, hipStream_t s);
bool wgradn_ok(const float* x, int din, long x_ld, int dout);
int launch_wgradn(const float* x, long x_ld, const float* dy, long dy_ld, long m, int din, int dout, float* part_dw,
                  float* part_db, int nblocks, hipStream_t s);
int64_t wtable_bytes(int din, int dout);
void launch_wtable_split(const float* w, long w_ld, int trans_w, int din, int dout, void* workspace, hipStream_t s);
int launch_gemm3_wgrad(const float* x, long x_ld, const float* dy, long dy_ld, long m, int din, int dout,
                       float* part_dw, float* part_db, int nblocks, hipStream_t s, const float* yact = nullptr,
                       int act = KGCN_ACT_NONE);
bool gemmh_wgrad_ok(int din, int dout, long m);
int launch_gemmh_wgrad(const float* x, long x_ld, const float* dy, long dy_ld, long m, int din, int dout, float* part_dw,
                       float* part_db, int nblocks, hipStream_t s, const float* yact, int act);

constexpr int BM = 128;      // rows per workgroup (32 per wave)
constexpr int BN = 64;       // output columns per workgroup
constexpr int BK = 32;       // k chunk
constexpr int XS_LD = BK + 4;  // +16 B pad: conflict-free ds_read_b128 of A fragments

__global__ __launch_bounds__(256) void dense_fwd_kernel(
    const float* __restrict__ x, long m, int din, long x_ld, const float* __restrict__ w,
    long w_ld, int trans_w, const float* __restrict__ bias, float* __restrict__ y, int dout,
    long y_ld, int act) {
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
  for (int r = 0; r < 16; ++r) { acc0[r] += bv0; acc1[r] += bv1; }
  if (act != KGCN_ACT_NONE) {                          // one uniform branch around the whole activation
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = act_fwd(acc0[r], act); acc1[r] = act_fwd(acc1[r], act); }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const long row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    if (row < m) {
      if (c0 < dout) y[row * y_ld + c0] = acc0[r];
      if (c1 < dout) y[row * y_ld + c1] = acc1[r];
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
// Two partial arrays (e.g. dW and dbias of one layer) are reduced by ONE launch: outputs [0, n) come from
// `part`, outputs [n, n + n2) from `part2`.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part,
                                                              int nparts, long n,
                                                              float* __restrict__ out,
                                                              const float* __restrict__ part2, long n2,
                                                              float* __restrict__ out2) {
  __shared__ float red[8][33];
  const int oi = threadIdx.x & 31, pg = threadIdx.x >> 5;
  long o = (long)blockIdx.x * 32 + oi;
  const bool second = o >= n;
  const float* src = second ? part2 : part;
  float* dst = second ? out2 : out;
  const long nn = second ? n2 : n;
  if (second) o -= n;
  float s[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) s[u] = 0.f;
  if (o < nn) {
    int p = pg;
    for (; p + 56 < nparts; p += 64) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += src[(long)(p + 8 * u) * nn + o];
    }
    for (; p < nparts; p += 8) s[0] += src[(long)p * nn + o];
  }
  red[pg][oi] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (pg == 0 && o < nn) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][oi];
    dst[o] = t;
  }
}

// ------------------------------------------------------------------------------------------------
// Persistent variants (the fast path): one workgroup per CU, 8 waves, every wave owns whole 32-row
// tiles.  The weight panel [din x 64] of the block's 64 output columns lives in LDS for the whole
// launch, the next x tile is already in flight (registers) while the current one is multiplied,
// MFMAs stay in clusters (the f32 MFMA shares the VALU datapath, see fused.hip).
// ------------------------------------------------------------------------------------------------
constexpr int PT_LD = 68;                 // padded row stride of the per-wave x tile (b128 A reads)
constexpr int PT_FLOATS = 32 * PT_LD;
constexpr int P_WAVES = 8;

// one [32 rows x 64 cols] chunk of x in flight: VEC: 8 float4 per lane, else 32 floats per lane
template <bool VEC>
struct XChunk { f32x4 v[8]; };
template <>
struct XChunk<false> { float v[32]; };

template <bool VEC>
__device__ __forceinline__ void issue_xchunk(XChunk<VEC>& f, const float* __restrict__ x, long m,
                                             int din, long x_ld, long row0, int k0, int lane) {
  if constexpr (VEC) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = lane + q * 64, r = i >> 4, c = (i & 15) * 4;
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      f.v[q] = (row0 + r < m && k0 + c < din) ? *reinterpret_cast<const f32x4*>(x + (row0 + r) * x_ld + k0 + c) : z;
    }
  } else {
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      const int i = lane + q * 64, r = i >> 6, c = i & 63;
      f.v[q] = (row0 + r < m && k0 + c < din) ? x[(row0 + r) * x_ld + k0 + c] : 0.f;
    }
  }
}
template <bool VEC>
__device__ __forceinline__ void land_xchunk(const XChunk<VEC>& f, float* tile, int ld, int lane) {
  if constexpr (VEC) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = lane + q * 64, r = i >> 4, c = (i & 15) * 4;
      *reinterpret_cast<f32x4*>(tile + r * ld + c) = f.v[q];
    }
  } else {
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      const int i = lane + q * 64, r = i >> 6, c = i & 63;
      tile[r * ld + c] = f.v[q];
    }
  }
}

__device__ __forceinline__ void lds_handoff() {   // intra-wave LDS hand-off: compiler barrier only
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

template <bool VEC>
__global__ __launch_bounds__(512, 2) void dense_fwd_persist_kernel(
    const float* __restrict__ x, long m, int din, long x_ld, const float* __restrict__ w, long w_ld,
    int trans_w, const float* __restrict__ bias, float* __restrict__ y, int dout, long y_ld, int kp, int act) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  float* Wp = reinterpret_cast<float*>(dsm);                      // [kp][64], zero padded
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, hi = lane >> 5;
  float* xs = Wp + (size_t)kp * 64 + (size_t)wave * PT_FLOATS;    // this wave's x tile
  const int n0 = blockIdx.y * 64;
  for (int i = tid; i < kp * 64; i += blockDim.x) {
    const int k = i >> 6, j = n0 + (i & 63);
    float v = 0.f;
    if (k < din && j < dout) v = trans_w ? w[(long)j * w_ld + k] : w[(long)k * w_ld + j];
    Wp[i] = v;
  }
  __syncthreads();
  const int c0 = n0 + li, c1 = n0 + 32 + li;
  const float b0 = (bias && c0 < dout) ? bias[c0] : 0.f;
  const float b1 = (bias && c1 < dout) ? bias[c1] : 0.f;

  const long ntiles = (m + 31) / 32;
  const int nkc = kp >> 6;
  const long nwaves = (long)gridDim.x * P_WAVES;
  long tile = (long)blockIdx.x * P_WAVES + wave;
  if (tile >= ntiles) return;
  // flattened (tile, k chunk) sequence with one chunk of lookahead
  XChunk<VEC> fx;
  issue_xchunk<VEC>(fx, x, m, din, x_ld, tile * 32, 0, lane);
  f32x16 acc0, acc1;
  int kc = 0;
  for (;;) {
    land_xchunk<VEC>(fx, xs, PT_LD, lane);
    lds_handoff();
    // next chunk (same tile, or first chunk of the wave's next tile; clamped on the last step)
    const bool last_kc = (kc + 1 == nkc);
    const long tn = last_kc ? tile + nwaves : tile;
    const bool more = tn < ntiles;
    issue_xchunk<VEC>(fx, x, m, din, x_ld, (more ? tn : tile) * 32, more ? (last_kc ? 0 : (kc + 1) * 64) : kc * 64, lane);
    if (kc == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = b0; acc1[r] = b1; }
    }
    {
      f32x4 a4[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) a4[q] = *reinterpret_cast<const f32x4*>(xs + li * PT_LD + hi * 32 + q * 4);
      const float* wb = Wp + (size_t)(kc * 64 + hi * 32) * 64 + li;
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        const float a = a4[s >> 2][s & 3];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb[s * 64], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb[s * 64 + 32], acc1, 0, 0, 0);
      }
    }
    lds_handoff();
    if (last_kc) {
      const long row0 = tile * 32;
      if (act != KGCN_ACT_NONE) {                      // one uniform branch around the whole activation
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = act_fwd(acc0[r], act); acc1[r] = act_fwd(acc1[r], act); }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long row = row0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (row < m) {
          if (c0 < dout) y[row * y_ld + c0] = acc0[r];
          if (c1 < dout) y[row * y_ld + c1] = acc1[r];
        }
      }
      if (!more) break;
      tile = tn;
      kc = 0;
    } else {
      ++kc;
    }
  }
}

// dW[64x64 block] / dbias partials: every wave owns whole 32-row tiles and all four 32x32 output
// tiles (64 accumulator registers live across the wave's whole row range), rows are the MFMA K.
template <bool VEC>
__global__ __launch_bounds__(512, 2) void dense_wgrad_persist_kernel(
    const float* __restrict__ x, long x_ld, const float* __restrict__ dy, long dy_ld, long m, int din,
    int dout, float* __restrict__ part_dw, float* __restrict__ part_db) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, hi = lane >> 5;
  float* xs = reinterpret_cast<float*>(dsm) + (size_t)wave * (2 * 32 * 64);   // [32][64]
  float* gs = xs + 32 * 64;                                                    // [32][64]
  const int i0 = blockIdx.y * 64, j0 = blockIdx.z * 64;

  f32x16 d00, d01, d10, d11;
#pragma unroll
  for (int r = 0; r < 16; ++r) { d00[r] = 0.f; d01[r] = 0.f; d10[r] = 0.f; d11[r] = 0.f; }
  float cs0 = 0.f, cs1 = 0.f;

  const long ntiles = (m + 31) / 32;
  const long nwaves = (long)gridDim.x * P_WAVES;
  long tile = (long)blockIdx.x * P_WAVES + wave;
  if (tile < ntiles) {
    XChunk<VEC> fx, fg;
    issue_xchunk<VEC>(fx, x, m, din, x_ld, tile * 32, i0, lane);
    issue_xchunk<VEC>(fg, dy, m, dout, dy_ld, tile * 32, j0, lane);
    for (;;) {
      land_xchunk<VEC>(fx, xs, 64, lane);
      land_xchunk<VEC>(fg, gs, 64, lane);
      lds_handoff();
      const long tn = tile + nwaves;
      const bool more = tn < ntiles;
      issue_xchunk<VEC>(fx, x, m, din, x_ld, (more ? tn : tile) * 32, i0, lane);
      issue_xchunk<VEC>(fg, dy, m, dout, dy_ld, (more ? tn : tile) * 32, j0, lane);
#pragma unroll 4
      for (int s = 0; s < 16; ++s) {
        const int n = hi * 16 + s;
        const float a0 = xs[n * 64 + li], a1 = xs[n * 64 + 32 + li];
        const float f0 = gs[n * 64 + li], f1 = gs[n * 64 + 32 + li];
        d00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, f0, d00, 0, 0, 0);
        d01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, f1, d01, 0, 0, 0);
        d10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, f0, d10, 0, 0, 0);
        d11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, f1, d11, 0, 0, 0);
        cs0 += f0;
        cs1 += f1;
      }
      lds_handoff();
      if (!more) break;
      tile = tn;
    }
  }
  // reduce the 8 waves through LDS (each wave parks 64x64 + 64 floats in its own 16 KB + ...)
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
    for (int wv = 0; wv < P_WAVES; ++wv) sum += base[(size_t)wv * (64 * 64 + 64) + i];
    const int row = i0 + (i >> 6), col = j0 + (i & 63);
    if (row < din && col < dout) pw[(long)row * dout + col] = sum;
  }
  if (part_db && blockIdx.y == 0 && tid < 64) {
    float sum = 0.f;
#pragma unroll
    for (int wv = 0; wv < P_WAVES; ++wv) sum += base[(size_t)wv * (64 * 64 + 64) + 64 * 64 + tid];
    if (j0 + tid < dout) part_db[(long)blockIdx.x * dout + j0 + tid] = sum;
  }
}

static int persist_blocks(long m) {
  long tiles = (m + 31) / 32;
  long b = (tiles + P_WAVES - 1) / P_WAVES;
  if (b > kNumCU) b = kNumCU;
  return b < 1 ? 1 : (int)b;
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
                     part, nparts, n, out, nullptr, 0L, nullptr);
  return check_launch("reduce_partials_kernel");
}

// both reductions of a layer (dW [n], dbias [n2]) in one launch; n is rounded up to whole 32-output groups
// of the first array, so no workgroup straddles the two
int launch_reduce_partials2(const float* part, int nparts, long n, float* out, const float* part2, long n2,
                            float* out2, hipStream_t s) {
  if (n % 32) {                         // keep the simple kernel: two launches when dW is not a multiple of 32
    if (int rc = launch_reduce_partials(part, nparts, n, out, s)) return rc;
    return launch_reduce_partials(part2, nparts, n2, out2, s);
  }
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((n + n2 + 31) / 32)), dim3(256), 0, s,
                     part, nparts, n, out, part2, n2, out2);
  return check_launch("reduce_partials_kernel");
}

// ---- deferred second stages ------------------------------------------------------------------------------------------
// The weight / bias gradients of a training step are read once, by the optimiser at its end, but every weight-gradient call
// finished its own partials with a 5-7 us launch: 9 per step of model_multitask.py (51 us of 1.31 ms), 7 of sparse.py (42 us of
// 0.34 ms).  With kgcn_reduce_defer(1) those second stages are QUEUED (host side, per thread) and kgcn_reduce_flush adds all of
// them in ONE launch; the caller keeps the workspaces alive until then and reads no gradient before it.
constexpr int kMaxReduceJobs = 24;
struct ReduceJobs {
  const float* part[kMaxReduceJobs];
  float* out[kMaxReduceJobs];
  long n[kMaxReduceJobs];
  int nparts[kMaxReduceJobs];
  int block0[kMaxReduceJobs + 1];       // first workgroup of every job (32 outputs per workgroup)
  int njobs;
};

__global__ __launch_bounds__(256) void reduce_partials_multi_kernel(ReduceJobs jb) {
  __shared__ float red[8][33];
  int j = 0;
  while (j + 1 < jb.njobs && (int)blockIdx.x >= jb.block0[j + 1]) ++j;          // uniform
  const float* __restrict__ src = jb.part[j];
  const long nn = jb.n[j];
  const int nparts = jb.nparts[j];
  const int oi = threadIdx.x & 31, pg = threadIdx.x >> 5;
  const long o = (long)((int)blockIdx.x - jb.block0[j]) * 32 + oi;
  float s[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) s[u] = 0.f;
  if (o < nn) {                          // the same order of additions as reduce_partials_kernel: bit-identical results
    int p = pg;
    for (; p + 56 < nparts; p += 64) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += src[(long)(p + 8 * u) * nn + o];
    }
    for (; p < nparts; p += 8) s[0] += src[(long)p * nn + o];
  }
  red[pg][oi] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (pg == 0 && o < nn) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][oi];
    jb.out[j][o] = t;
  }
}

// PROCESS-wide, not per thread: PyTorch runs the backward nodes of a device on its autograd worker thread, the caller switches
// deferral on from another one.  One training loop per process (one process per GPU) is the contract; the mutex only keeps the
// queue itself consistent.
struct PendingReduce { const float* part; int nparts; long n; float* out; };
static std::atomic<bool> g_defer_reduce{false};
static std::mutex g_pending_mutex;
static PendingReduce g_pending[256];
static int g_npending = 0;

static int flush_pending_locked(hipStream_t s) {
  int done = 0;
  while (done < g_npending) {
    ReduceJobs jb{};
    int blocks = 0, nj = 0;
    for (; nj < kMaxReduceJobs && done + nj < g_npending; ++nj) {
      const PendingReduce& p = g_pending[done + nj];
      jb.part[nj] = p.part; jb.out[nj] = p.out; jb.n[nj] = p.n; jb.nparts[nj] = p.nparts; jb.block0[nj] = blocks;
      blocks += (int)((p.n + 31) / 32);
    }
    jb.block0[nj] = blocks;
    jb.njobs = nj;
    if (blocks > 0) {
      hipLaunchKernelGGL(reduce_partials_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, s, jb);
      if (int rc = check_launch("reduce_partials_multi_kernel")) { g_npending = 0; return rc; }
    }
    done += nj;
  }
  g_npending = 0;
  return 0;
}

// the second stage of a PARAMETER gradient: queued while deferral is on, launched otherwise
int reduce_or_defer(const float* part, int nparts, long n, float* out, hipStream_t s) {
  if (!g_defer_reduce.load()) return launch_reduce_partials(part, nparts, n, out, s);
  std::lock_guard<std::mutex> lock(g_pending_mutex);
  if (g_npending == 256)
    if (int rc = flush_pending_locked(s)) return rc;
  g_pending[g_npending++] = PendingReduce{part, nparts, n, out};
  return 0;
}
static int flush_pending(hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_pending_mutex);
  return flush_pending_locked(s);
}

}  // namespace kgcn

using namespace kgcn;

extern "C" int kgcn_reduce_defer(int32_t on) {
  return g_defer_reduce.exchange(on != 0) ? 1 : 0;
}
extern "C" int kgcn_reduce_pending(void) { return g_npending; }
extern "C" int kgcn_reduce_flush(void* stream) { return flush_pending(as_stream(stream)); }

// wide layers take the bf16-split GEMM of gemm3.hip (W pre-split into the fragment table of wtable.hip when the caller
// provides the workspace for it)
static bool wide_layer(int din, int dout) { return dout > 128 && din >= 32; }
// where reading W pre-split from the fragment table measured ahead of splitting it inside the kernel (tools/gemm_bench.py,
// 204,800 rows: 256 -> 256 +15% forward / +5% dX, 512 -> 256 +5% / -3%, 256 -> 512 +2%; 128 -> 256 -14%: few k-steps)
static bool table_pays(int din, int dout) { return wide_layer(din, dout) && din >= 32; }

namespace kgcn {
bool skinny_n_ok(int din, int dout, int trans_w);
bool skinny_k_ok(int din, int dout);
int launch_skinny_n_fwd(const float* x, long m, int din, long x_ld, const float* w, long w_ld, const float* bias, float* y,
                        int dout, long y_ld, int act, hipStream_t s);
int launch_skinny_k_fwd(const float* x, long m, int din, long x_ld, const float* w, long w_ld, int trans_w, const float* bias,
                        float* y, int dout, long y_ld, int act, hipStream_t s);
int skinny_wgrad_parts(long m);
int launch_skinny_n_wgrad(const float* x, long m, int din, long x_ld, const float* g, long g_ld, int dout, float* part_dw,
                          float* part_db, int nparts, hipStream_t s);
bool wgradx_ok(int din, int dout, long x_ld, long dy_ld);
int launch_wgradx(const float* x, long x_ld, const float* dy, long dy_ld, long m, int din, int dout, float* part_dw, float* part_db,
                  int nparts, hipStream_t s, const float* yact, int act);
}  // namespace kgcn

// table_ready: `workspace` already holds the fragment table of (w, trans_w) (kgcn_wtable_split_multi at the start of the step)
static int dense_fwd_impl(const float* x, int64_t m, int32_t din, int64_t x_ld, const float* w, int64_t w_ld,
                          int32_t trans_w, const float* bias, float* y, int32_t dout, int64_t y_ld, int act,
                          void* workspace, int64_t workspace_bytes, void* stream, bool table_ready = false) {
  if (act < KGCN_ACT_NONE || act > KGCN_ACT_TANH) return fail("kgcn_dense_fwd_f32: unknown activation code %d", act);
  if (m < 0 || din <= 0 || dout <= 0)
    return fail("kgcn_dense_fwd_f32: bad shape m=%lld din=%d dout=%d", (long long)m, din, dout);
  if (m == 0) return 0;
  if (!x || !w || !y) return fail("kgcn_dense_fwd_f32: NULL operand");
  if (x_ld < din || y_ld < dout) return fail("kgcn_dense_fwd_f32: leading dimension too small");
  if (w_ld < (trans_w ? din : dout)) return fail("kgcn_dense_fwd_f32: w_ld too small");
  // read-out layers (2..16 columns on one side): a row per wave, no panels (skinny.hip)
  if (skinny_n_ok(din, dout, trans_w))
    return launch_skinny_n_fwd(x, (long)m, din, (long)x_ld, w, (long)w_ld, bias, y, dout, (long)y_ld, act, as_stream(stream));
  if (skinny_k_ok(din, dout))
    return launch_skinny_k_fwd(x, (long)m, din, (long)x_ld, w, (long)w_ld, trans_w, bias, y, dout, (long)y_ld, act,
                               as_stream(stream));
  // wide layers: f32-MFMA bound -> the bf16-split GEMM (gemm3.hip).  With a workspace W is split ONCE into a fragment
  // table (wtable.hip) that the waves read from L2, and the kernel's staging only splits x.
  if (wide_layer(din, dout)) {
    static const char* route = dev_knob("KGCN_DENSE_ROUTE");         // development: "gemm3" = never use the table
    const void* table = nullptr;
    if (!(route && !strcmp(route, "gemm3")) && table_pays(din, dout) && workspace &&
        workspace_bytes >= wtable_bytes(din, dout) && m >= 1024) {
      if (!table_ready) launch_wtable_split(w, (long)w_ld, trans_w, din, dout, workspace, as_stream(stream));
      table = workspace;
    }
    return launch_gemm3_fwd(x, (long)m, din, (long)x_ld, w, (long)w_ld, trans_w, bias, y, dout, (long)y_ld, act, table,
                            as_stream(stream));
  }
  // wide input, narrow output (256 -> 50): one 64-column block per wave on the bf16 pipe (gemmn.hip), W from the table
  if (gemmn_pays(x, din, (long)x_ld, dout) && workspace && workspace_bytes >= wtable_bytes(din, dout) && m >= 1024) {
    static const char* route = dev_knob("KGCN_DENSE_ROUTE");
    if (!(route && !strcmp(route, "gemm3"))) {
      if (!table_ready) launch_wtable_split(w, (long)w_ld, trans_w, din, dout, workspace, as_stream(stream));
      return launch_gemmn_fwd(x, (long)m, din, (long)x_ld, workspace, bias, y, dout, (long)y_ld, act, as_stream(stream));
    }
  }
  // 50-wide layers: flat tile movement (narrow.hip)
  if (narrow_fwd_ok(x, din, (long)x_ld, y, dout, (long)y_ld))
    return launch_narrow_fwd(x, (long)m, din, w, (long)w_ld, trans_w, bias, y, dout, act, as_stream(stream));
  {
    // fast path: weight panel [din_pad x 64] resident in LDS next to 8 per-wave x tiles
    const int kp = ((din + 63) / 64) * 64;
    const size_t lds = (size_t)kp * 64 * 4 + (size_t)P_WAVES * PT_FLOATS * 4;
    if (lds <= (size_t)kLdsBytes) {
      const bool vec = (din % 4 == 0) && (x_ld % 4 == 0) && aligned16(x);
      static thread_local bool attr_set = false;
      if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_fwd_persist_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_fwd_persist_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        attr_set = true;
      }
      // two workgroups per CU when the panel is small enough (4 waves per SIMD in total)
      int blocks = persist_blocks(m);
      if (2 * lds <= (size_t)kLdsBytes && blocks == kNumCU) blocks = 2 * kNumCU;
      dim3 grid((unsigned)blocks, (unsigned)((dout + 63) / 64));
      if (vec)
        hipLaunchKernelGGL(dense_fwd_persist_kernel<true>, grid, dim3(64 * P_WAVES), lds, as_stream(stream), x,
                           (long)m, din, (long)x_ld, w, (long)w_ld, trans_w, bias, y, dout, (long)y_ld, kp, act);
      else
        hipLaunchKernelGGL(dense_fwd_persist_kernel<false>, grid, dim3(64 * P_WAVES), lds, as_stream(stream), x,
                           (long)m, din, (long)x_ld, w, (long)w_ld, trans_w, bias, y, dout, (long)y_ld, kp, act);
      return check_launch("dense_fwd_persist_kernel");
    }
  }
  const long gx = (m + BM - 1) / BM;
  if (gx > 0x7fffffffL) return fail("kgcn_dense_fwd_f32: m too large");
  dim3 grid((unsigned)gx, (unsigned)((dout + BN - 1) / BN));
  hipLaunchKernelGGL(dense_fwd_kernel, grid, dim3(256), 0, as_stream(stream), x, (long)m, din,
                     (long)x_ld, w, (long)w_ld, trans_w, bias, y, dout, (long)y_ld, act);
  return check_launch("dense_fwd_kernel");
}

extern "C" int kgcn_dense_fwd_f32(const float* x, int64_t m, int32_t din, int64_t x_ld, const float* w, int64_t w_ld,
                                  int32_t trans_w, const float* bias, float* y, int32_t dout, int64_t y_ld,
                                  void* stream) {
  return dense_fwd_impl(x, m, din, x_ld, w, w_ld, trans_w, bias, y, dout, y_ld, KGCN_ACT_NONE, nullptr, 0, stream);
}

static int dense_dx_dact_impl(const float* grad, const float* act_out, int64_t m, int32_t dout, int64_t ld,
                             const float* w, int64_t w_ld, int32_t din, float* dx, int64_t dx_ld, int32_t act,
                             float* dpre, void* workspace, int64_t workspace_bytes, void* stream, bool table_ready) {
  if (act <= KGCN_ACT_NONE || act > KGCN_ACT_TANH) return fail("kgcn_dense_dx_dact_f32: activation code %d", act);
  if (m < 0 || din <= 0 || dout <= 0)
    return fail("kgcn_dense_dx_dact_f32: bad shape m=%lld din=%d dout=%d", (long long)m, din, dout);
  if (m == 0) return 0;
  if (!grad || !act_out || !w || !dx || !dpre) return fail("kgcn_dense_dx_dact_f32: NULL operand");
  if (dpre == grad) return fail("kgcn_dense_dx_dact_f32: dpre must not alias grad");
  if (ld < dout || dx_ld < din || w_ld < dout) return fail("kgcn_dense_dx_dact_f32: leading dimension too small");
  // the contraction runs over the layer's OUTPUT width: K = dout, N = din, W used transposed
  static const char* route = dev_knob("KGCN_DENSE_ROUTE");
  if (!(route && !strcmp(route, "gemm3")) && table_pays(dout, din) && workspace && workspace_bytes >= wtable_bytes(dout, din) &&
      m >= 1024) {
    if (!table_ready) launch_wtable_split(w, (long)w_ld, 1, dout, din, workspace, as_stream(stream));
    const int rc = launch_gemm3_dx_dact(grad, act_out, dpre, (long)m, dout, (long)ld, workspace, dx, din, (long)dx_ld, act,
                                        as_stream(stream));
    if (rc >= 0) return rc;
  }
  if (ld != dout) return fail("kgcn_dense_dx_dact_f32: rows must be contiguous (ld == dout) outside the fused form");
  // (50-wide layers: act' inside the operand tile of narrow.hip's kernel -- which then also writes d pre-activation -- was built
  // and measured: 34.2 us against 11.2 + 22.2 us of the two launches at 117,888 rows, 35-43 spilled registers; not kept)
  if (int rc = kgcn_act_bwd_f32(act_out, grad, m * dout, act, dpre, stream)) return rc;
  return dense_fwd_impl(dpre, m, dout, ld, w, w_ld, 1, nullptr, dx, din, dx_ld, KGCN_ACT_NONE, workspace, workspace_bytes,
                        stream, table_ready);
}

// 1 when kgcn_dense_dx_dact_gather_f32 takes this shape (the fused table form of the wide layers)
extern "C" int kgcn_dense_dx_dact_gather_supported(int64_t m, int32_t din, int32_t dout) {
  return (m >= 1024 && din > 0 && dout > 0 && dout % 4 == 0 && table_pays(dout, din)) ? 1 : 0;
}

extern "C" int kgcn_dense_dx_dact_gather_f32(const float* grad, const float* pooled_grad, int64_t pooled_ld, int32_t n_nodes,
                                             const float* act_out,
                                             int64_t m, int32_t dout, int64_t ld, const float* w, int64_t w_ld, int32_t din,
                                             float* dx, int64_t dx_ld, int32_t act, float* dpre, void* table, int64_t table_bytes,
                                             int32_t table_ready, void* stream) {
  if (act <= KGCN_ACT_NONE || act > KGCN_ACT_TANH) return fail("kgcn_dense_dx_dact_gather_f32: activation code %d", act);
  if (!kgcn_dense_dx_dact_gather_supported(m, din, dout))
    return fail("kgcn_dense_dx_dact_gather_f32: shape m=%lld %d -> %d has no fused form (kgcn_graph_gather_bwd_f32 + "
                "kgcn_dense_dx_dact_f32)", (long long)m, din, dout);
  if (!pooled_grad || n_nodes <= 0 || m % n_nodes != 0)
    return fail("kgcn_dense_dx_dact_gather_f32: pooled gradient / %d nodes per graph do not match %lld rows", n_nodes, (long long)m);
  if (pooled_ld < dout || pooled_ld % 4 != 0) return fail("kgcn_dense_dx_dact_gather_f32: pooled_ld=%lld (>= dout, a multiple of 4)", (long long)pooled_ld);
  if (!act_out || !w || !dx || !dpre) return fail("kgcn_dense_dx_dact_gather_f32: NULL operand");
  if (dpre == grad) return fail("kgcn_dense_dx_dact_gather_f32: dpre must not alias grad");
  if (ld < dout || dx_ld < din || w_ld < dout) return fail("kgcn_dense_dx_dact_gather_f32: leading dimension too small");
  if (!table || table_bytes < wtable_bytes(dout, din)) return fail("kgcn_dense_dx_dact_gather_f32: table / workspace too small");
  if (!table_ready) launch_wtable_split(w, (long)w_ld, 1, dout, din, table, as_stream(stream));
  const int rc = launch_gemm3_dx_dact(grad, act_out, dpre, (long)m, dout, (long)ld, table, dx, din, (long)dx_ld, act,
                                      as_stream(stream), pooled_grad, n_nodes, (long)pooled_ld);
  if (rc < 0) return fail("kgcn_dense_dx_dact_gather_f32: operands must be 16-byte aligned with ld %% 4 == 0");
  return rc;
}

// d epsilon of a GINAggregate in front of an activated wide layer whose INPUT needs no gradient (the first block of model_gin.py):
// <(grad (.) act'(act_out)) W^T, dotx> with the product never stored (gemmh.hip, dot form); d pre-activation is written as usual.
namespace kgcn {
int launch_gemmh_dx_dact(const float* grad, const float* act_out, float* dpre, long m, int k, long ld, const void* tabh, float* dx,
                         int n, long dx_ld, int dact, hipStream_t s, const float* pooled_grad, int n_nodes, long pooled_ld,
                         float* dot_part);
int gemmh_dot_parts(long m, int dout);
int64_t wtable_bf16_bytes(int din, int dout);
__global__ __launch_bounds__(256) void dx_dot_final_kernel(const float* __restrict__ part, int nparts, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}
}  // namespace kgcn

extern "C" int kgcn_dense_dx_dact_dot_supported(int64_t m, int32_t din, int32_t dout) {
  // (the f16 weight-stationary kernel only: wide layer, >= 16,384 rows; dout = the contraction here)
  return (din > 128 && din <= 256 && din % 4 == 0 && dout % 4 == 0 && dout >= 32 && dout <= 256 && m >= (int64_t)kNumCU * 64) ? 1 : 0;
}

extern "C" int64_t kgcn_dense_dx_dact_dot_workspace_bytes(int64_t m, int32_t din) {
  return (int64_t)gemmh_dot_parts((long)m, din) * 4;
}

extern "C" int kgcn_dense_dx_dact_dot_f32(const float* grad, const float* act_out, int64_t m, int32_t dout, int64_t ld, const float* w,
                                          int64_t w_ld, int32_t din, const float* dotx, int64_t dotx_ld, int32_t act, float* dpre,
                                          void* table, int64_t table_bytes, int32_t table_ready, float* dot_out, void* workspace,
                                          int64_t workspace_bytes, void* stream) {
  if (act <= KGCN_ACT_NONE || act > KGCN_ACT_TANH) return fail("kgcn_dense_dx_dact_dot_f32: activation code %d", act);
  if (!kgcn_dense_dx_dact_dot_supported(m, din, dout))
    return fail("kgcn_dense_dx_dact_dot_f32: shape m=%lld %d -> %d has no fused form (kgcn_dense_dx_dact_f32 + kgcn_dot_f32)",
                (long long)m, din, dout);
  if (!grad || !act_out || !w || !dotx || !dpre || !dot_out) return fail("kgcn_dense_dx_dact_dot_f32: NULL operand");
  if (dpre == grad) return fail("kgcn_dense_dx_dact_dot_f32: dpre must not alias grad");
  if (ld < dout || dotx_ld < din || w_ld < dout || dotx_ld % 4 != 0 || !aligned16(dotx))
    return fail("kgcn_dense_dx_dact_dot_f32: leading dimension too small / dotx not 16-byte aligned rows");
  if (!table || table_bytes < wtable_bytes(dout, din)) return fail("kgcn_dense_dx_dact_dot_f32: table / workspace too small");
  if (!workspace || workspace_bytes < kgcn_dense_dx_dact_dot_workspace_bytes(m, din))
    return fail("kgcn_dense_dx_dact_dot_f32: workspace %lld < %lld bytes", (long long)workspace_bytes,
                (long long)kgcn_dense_dx_dact_dot_workspace_bytes(m, din));
  hipStream_t s = as_stream(stream);
  if (!table_ready) launch_wtable_split(w, (long)w_ld, 1, dout, din, table, s);
  float* part = static_cast<float*>(workspace);
  const int rc = launch_gemmh_dx_dact(grad, act_out, dpre, (long)m, dout, (long)ld,
                                      static_cast<const char*>(table) + wtable_bf16_bytes(dout, din), const_cast<float*>(dotx), din,
                                      (long)dotx_ld, act, s, nullptr, 0, 0, part);
  if (rc < 0) return fail("kgcn_dense_dx_dact_dot_f32: operands must be 16-byte aligned with ld %% 4 == 0");
  if (rc) return rc;
  hipLaunchKernelGGL(dx_dot_final_kernel, dim3(1), dim3(256), 0, s, part, gemmh_dot_parts((long)m, din), dot_out);
  return check_launch("dx_dot_final_kernel");
}

// ---- one-pass backward of a wide dense layer (gemmb.hip): dX, dW and dbias from ONE sweep over (grad, act_out, x); the
// d pre-activation tensor is never written.  The weight-gradient partials go through the same second stage as every other
// weight gradient (deferrable: kgcn_reduce_defer).
namespace kgcn {
int launch_gemmb(const float* grad, const float* act_out, long m, int din, int dout, long ld, const float* x, long x_ld,
                 const void* tabh, float* dx, long dx_ld, float* part_dw, float* part_db, int dact, const float* pooled_grad,
                 int n_nodes, long pooled_ld, hipStream_t s, float* dot_part);
int launch_reduce_pair(const float* part_dw, long n_dw, float* dw, const float* part_db, long n_db, float* dbias, int nparts,
                       hipStream_t s);
}  // namespace kgcn

extern "C" int kgcn_dense_bwd_supported(int64_t m, int32_t din, int32_t dout) {
  return (din > 128 && din <= 256 && dout == 256 && din % 4 == 0 && m >= (int64_t)kNumCU * 64) ? 1 : 0;
}

extern "C" int kgcn_dense_bwd_f32(const float* grad, const float* pooled_grad, int64_t pooled_ld, int32_t n_nodes,
                                  const float* act_out, int32_t act, int64_t ld, const float* x, int64_t x_ld, int64_t m,
                                  int32_t din, int32_t dout, const float* w, int64_t w_ld, float* dx, int64_t dx_ld, float* dw,
                                  float* dbias, void* table, int64_t table_bytes, int32_t table_ready, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
  if (act < KGCN_ACT_NONE || act > KGCN_ACT_TANH) return fail("kgcn_dense_bwd_f32: activation code %d", act);
  if (!kgcn_dense_bwd_supported(m, din, dout))
    return fail("kgcn_dense_bwd_f32: shape m=%lld %d -> %d has no one-pass form (kgcn_dense_dx_dact_f32 / kgcn_dense_fwd_f32 + "
                "kgcn_dense_wgrad_f32)", (long long)m, din, dout);
  if ((!grad && !pooled_grad) || !x || !w || !dx || !dw) return fail("kgcn_dense_bwd_f32: NULL operand");
  if (act != KGCN_ACT_NONE && !act_out) return fail("kgcn_dense_bwd_f32: act_out is NULL with activation code %d", act);
  if (pooled_grad && (act == KGCN_ACT_NONE || n_nodes < 8 || m % n_nodes != 0 || pooled_ld < dout || pooled_ld % 4 != 0))
    return fail("kgcn_dense_bwd_f32: pooled gradient needs an activated layer, %d (>= 8) nodes per graph dividing %lld rows and "
                "pooled_ld=%lld a multiple of 4 >= dout", n_nodes, (long long)m, (long long)pooled_ld);
  if (ld < dout || x_ld < din || dx_ld < din || w_ld < dout) return fail("kgcn_dense_bwd_f32: leading dimension too small");
  if (dx == x || dx == grad || dx == act_out) return fail("kgcn_dense_bwd_f32: dx must not alias an input");
  if (!table || table_bytes < wtable_bytes(dout, din)) return fail("kgcn_dense_bwd_f32: table / workspace too small");
  const int64_t need = kgcn_dense_wgrad_workspace_bytes(m, din, dout);
  if (!workspace || workspace_bytes < need)
    return fail("kgcn_dense_bwd_f32: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
  hipStream_t s = as_stream(stream);
  if (!table_ready) launch_wtable_split(w, (long)w_ld, 1, dout, din, table, s);
  const int pairs = kNumCU / 2;
  float* part_dw = static_cast<float*>(workspace);
  float* part_db = part_dw + (long)pairs * din * dout;
  const int rc = launch_gemmb(grad, act_out, (long)m, din, dout, (long)ld, x, (long)x_ld,
                              static_cast<const char*>(table) + wtable_bf16_bytes(dout, din), dx, (long)dx_ld, part_dw,
                              dbias ? part_db : nullptr, act, pooled_grad, n_nodes, (long)pooled_ld, s, nullptr);
  if (rc == -1) return fail("kgcn_dense_bwd_f32: operands must be 16-byte aligned with ld %% 4 == 0");
  if (rc < 0) return 1;
  return launch_reduce_pair(part_dw, (long)din * dout, dw, part_db, dout, dbias, rc, s);
}

// the one-pass backward where the d-input product is only needed for an inner product (kgcn_dense_dx_dact_dot_f32's case): dW, dbias
// and dot_out[0] = <dpre @ w^T, dotx> from ONE sweep; neither d pre-activation nor the [m, din] product is ever written
extern "C" int kgcn_dense_bwd_dot_f32(const float* grad, const float* act_out, int32_t act, int64_t ld, const float* x, int64_t x_ld,
                                      int64_t m, int32_t din, int32_t dout, const float* w, int64_t w_ld, const float* dotx,
                                      int64_t dotx_ld, float* dw, float* dbias, float* dot_out, void* table, int64_t table_bytes,
                                      int32_t table_ready, void* workspace, int64_t workspace_bytes, void* stream) {
  if (act <= KGCN_ACT_NONE || act > KGCN_ACT_TANH) return fail("kgcn_dense_bwd_dot_f32: activation code %d", act);
  if (!kgcn_dense_bwd_supported(m, din, dout))
    return fail("kgcn_dense_bwd_dot_f32: shape m=%lld %d -> %d has no one-pass form (kgcn_dense_dx_dact_dot_f32 + kgcn_dense_wgrad_f32)",
                (long long)m, din, dout);
  if (!grad || !act_out || !x || !w || !dotx || !dw || !dot_out) return fail("kgcn_dense_bwd_dot_f32: NULL operand");
  if (ld < dout || x_ld < din || dotx_ld < din || w_ld < dout) return fail("kgcn_dense_bwd_dot_f32: leading dimension too small");
  if (!table || table_bytes < wtable_bytes(dout, din)) return fail("kgcn_dense_bwd_dot_f32: table / workspace too small");
  const int pairs = kNumCU / 2;
  const int64_t need = kgcn_dense_wgrad_workspace_bytes(m, din, dout);
  if (!workspace || workspace_bytes < need || need < (int64_t)(pairs * ((int64_t)din * dout + dout) + kNumCU) * 4)
    return fail("kgcn_dense_bwd_dot_f32: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
  hipStream_t s = as_stream(stream);
  if (!table_ready) launch_wtable_split(w, (long)w_ld, 1, dout, din, table, s);
  float* part_dw = static_cast<float*>(workspace);
  float* part_db = part_dw + (long)pairs * din * dout;
  float* part_dot = part_db + (long)pairs * dout;                // one partial per workgroup: kNumCU floats
  const int rc = launch_gemmb(grad, act_out, (long)m, din, dout, (long)ld, x, (long)x_ld,
                              static_cast<const char*>(table) + wtable_bf16_bytes(dout, din), const_cast<float*>(dotx), (long)dotx_ld,
                              part_dw, dbias ? part_db : nullptr, act, nullptr, 0, 0, s, part_dot);
  if (rc == -1) return fail("kgcn_dense_bwd_dot_f32: operands must be 16-byte aligned with ld %% 4 == 0");
  if (rc < 0) return 1;
  hipLaunchKernelGGL(dx_dot_final_kernel, dim3(1), dim3(256), 0, s, part_dot, kNumCU, dot_out);
  if (int rc2 = check_launch("dx_dot_final_kernel")) return rc2;
  return launch_reduce_pair(part_dw, (long)din * dout, dw, part_db, dout, dbias, rc, s);
}

extern "C" int kgcn_dense_dx_dact_f32(const float* grad, const float* act_out, int64_t m, int32_t dout, int64_t ld,
                                      const float* w, int64_t w_ld, int32_t din, float* dx, int64_t dx_ld, int32_t act,
                                      float* dpre, void* workspace, int64_t workspace_bytes, void* stream) {
  return dense_dx_dact_impl(grad, act_out, m, dout, ld, w, w_ld, din, dx, dx_ld, act, dpre, workspace, workspace_bytes, stream,
                            false);
}

extern "C" int kgcn_dense_dx_dact_tab_f32(const float* grad, const float* act_out, int64_t m, int32_t dout, int64_t ld,
                                          const float* w, int64_t w_ld, int32_t din, float* dx, int64_t dx_ld, int32_t act,
                                          float* dpre, const void* table, int64_t table_bytes, void* stream) {
  return dense_dx_dact_impl(grad, act_out, m, dout, ld, w, w_ld, din, dx, dx_ld, act, dpre, const_cast<void*>(table),
                            table_bytes, stream, table != nullptr);
}

extern "C" int64_t kgcn_dense_fwd_workspace_bytes(int32_t din, int32_t dout) {
  if (din <= 0 || dout <= 0 || !(table_pays(din, dout) || (dout <= 64 && din >= 128 && din % 4 == 0))) return 0;
  return wtable_bytes(din, dout);
}

extern "C" int kgcn_dense_fwd_tab_f32(const float* x, int64_t m, int32_t din, int64_t x_ld, const float* w, int64_t w_ld,
                                      int32_t trans_w, const float* bias, float* y, int32_t dout, int64_t y_ld,
                                      int32_t act, const void* table, int64_t table_bytes, void* stream) {
  return dense_fwd_impl(x, m, din, x_ld, w, w_ld, trans_w, bias, y, dout, y_ld, act, const_cast<void*>(table), table_bytes,
                        stream, table != nullptr);
}

extern "C" int kgcn_dense_fwd_ws_f32(const float* x, int64_t m, int32_t din, int64_t x_ld, const float* w, int64_t w_ld,
                                     int32_t trans_w, const float* bias, float* y, int32_t dout, int64_t y_ld,
                                     int32_t act, void* workspace, int64_t workspace_bytes, void* stream) {
  return dense_fwd_impl(x, m, din, x_ld, w, w_ld, trans_w, bias, y, dout, y_ld, act, workspace, workspace_bytes, stream);
}

// dW and dbias partials of one layer in ONE launch (either output may be NULL)
namespace kgcn {
// ... now, whatever the deferral state: for results the SAME call reads back (batch normalisation's d gamma / d beta enter its dx)
int launch_reduce_pair_now(const float* part_dw, long n_dw, float* dw, const float* part_db, long n_db, float* dbias,
                           int nparts, hipStream_t s);
int launch_reduce_pair(const float* part_dw, long n_dw, float* dw, const float* part_db, long n_db, float* dbias,
                       int nparts, hipStream_t s) {
  if (g_defer_reduce.load()) {                           // a training step: all second stages in one launch at its end
    if (dw) if (int rc = reduce_or_defer(part_dw, nparts, n_dw, dw, s)) return rc;
    if (dbias) if (int rc = reduce_or_defer(part_db, nparts, n_db, dbias, s)) return rc;
    return 0;
  }
  return launch_reduce_pair_now(part_dw, n_dw, dw, part_db, n_db, dbias, nparts, s);
}
int launch_reduce_pair_now(const float* part_dw, long n_dw, float* dw, const float* part_db, long n_db, float* dbias,
                           int nparts, hipStream_t s) {
  if (dw && dbias) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((n_dw + n_db + 31) / 32)), dim3(256), 0, s, part_dw,
                       nparts, n_dw, dw, part_db, n_db, dbias);
    return check_launch("reduce_partials_kernel");
  }
  if (dw) return launch_reduce_partials(part_dw, nparts, n_dw, dw, s);
  if (dbias) return launch_reduce_partials(part_db, nparts, n_db, dbias, s);
  return 0;
}
}  // namespace kgcn

extern "C" int kgcn_dense_fwd_act_f32(const float* x, int64_t m, int32_t din, int64_t x_ld, const float* w,
                                      int64_t w_ld, int32_t trans_w, const float* bias, float* y, int32_t dout,
                                      int64_t y_ld, int32_t act, void* stream) {
  return dense_fwd_impl(x, m, din, x_ld, w, w_ld, trans_w, bias, y, dout, y_ld, act, nullptr, 0, stream);
}

namespace kgcn { bool gemmh_fwd_ok(const float* x, long m, int din, long x_ld, int dout); }

extern "C" int kgcn_dense_mfma_products(int32_t kind, int64_t m, int32_t din, int32_t dout) {
  if (m <= 0 || din <= 0 || dout <= 0) return 0;
  const float* aligned = reinterpret_cast<const float*>(uintptr_t(256));      // stands for a 16-byte aligned operand
  const long ld = (din + 3) & ~3;
  if (kind == 0 || kind == 1) {
    if (kind == 0 && (skinny_n_ok(din, dout, 0) || skinny_k_ok(din, dout))) return 0;
    if (wide_layer(din, dout)) return (m >= 1024 && din % 4 == 0 && gemmh_fwd_ok(aligned, (long)m, din, ld, dout)) ? 3 : 6;
    if (gemmn_pays(aligned, din, ld, dout) && m >= 1024) return 6;
    return 1;
  }
  if (skinny_n_ok(din, dout, 0)) return 0;
  if (wgradx_ok(din, dout, ld, dout) && m >= 4096) return 6;
  if (din > 64 && dout > 128) return gemmh_wgrad_ok(din, dout, (long)m) ? 3 : 6;
  if (wgradn_ok(aligned, din, ld, dout) && m >= 4096) return 6;
  return 1;
}

extern "C" int64_t kgcn_dense_wgrad_workspace_bytes(int64_t m, int32_t din, int32_t dout) {
  if (m <= 0 || din <= 0 || dout <= 0) return 0;
  long rpc;
  int nchunks;
  wgrad_plan(m, &rpc, &nchunks);
  if (nchunks < kNumCU) nchunks = kNumCU;   // the persistent kernel writes one partial per workgroup
  if (skinny_n_ok(din, dout, 0) && nchunks < 4 * kNumCU) nchunks = 4 * kNumCU;     // skinny.hip: up to 1,024 partials
  return (int64_t)nchunks * ((int64_t)din * dout + dout) * 4;
}

static int dense_wgrad_impl(const float* x, int64_t x_ld, const float* dy, int64_t dy_ld, int64_t m, int32_t din,
                            int32_t dout, float* dw, float* dbias, void* workspace, int64_t workspace_bytes, void* stream,
                            const float* yact, int act);

extern "C" int kgcn_dense_wgrad_f32(const float* x, int64_t x_ld, const float* dy, int64_t dy_ld,
                                    int64_t m, int32_t din, int32_t dout, float* dw, float* dbias,
                                    void* workspace, int64_t workspace_bytes, void* stream) {
  return dense_wgrad_impl(x, x_ld, dy, dy_ld, m, din, dout, dw, dbias, workspace, workspace_bytes, stream, nullptr,
                          KGCN_ACT_NONE);
}

// 1 when kgcn_dense_wgrad_dact_f32 forms d pre-activation inside the weight-gradient GEMM for this shape (else the caller
// runs kgcn_act_bwd_f32 first)
extern "C" int kgcn_dense_wgrad_dact_supported(int32_t din, int32_t dout) { return din > 64 && dout > 128 ? 1 : 0; }

extern "C" int kgcn_dense_wgrad_dact_f32(const float* x, int64_t x_ld, const float* dy, const float* act_out, int64_t dy_ld,
                                         int32_t act, int64_t m, int32_t din, int32_t dout, float* dw, float* dbias,
                                         void* workspace, int64_t workspace_bytes, void* stream) {
  if (act <= KGCN_ACT_NONE || act > KGCN_ACT_TANH) return fail("kgcn_dense_wgrad_dact_f32: activation code %d", act);
  if (!kgcn_dense_wgrad_dact_supported(din, dout))
    return fail("kgcn_dense_wgrad_dact_f32: shape %d x %d is not a wide layer (run kgcn_act_bwd_f32 + kgcn_dense_wgrad_f32)",
                din, dout);
  if (m > 0 && !act_out) return fail("kgcn_dense_wgrad_dact_f32: act_out is NULL");
  return dense_wgrad_impl(x, x_ld, dy, dy_ld, m, din, dout, dw, dbias, workspace, workspace_bytes, stream, act_out, act);
}

static int dense_wgrad_impl(const float* x, int64_t x_ld, const float* dy, int64_t dy_ld, int64_t m, int32_t din,
                            int32_t dout, float* dw, float* dbias, void* workspace, int64_t workspace_bytes, void* stream,
                            const float* yact, int act) {
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
  if (!yact && skinny_n_ok(din, dout, 0)) {
    // read-out layers: lanes own k, the 2..16 gradient columns of a row are wave-uniform (skinny.hip)
    nchunks = skinny_wgrad_parts(m);
    float* part_dw = static_cast<float*>(workspace);
    float* part_db = part_dw + (long)nchunks * din * dout;
    if (int rc = launch_skinny_n_wgrad(x, (long)m, din, (long)x_ld, dy, (long)dy_ld, dout, part_dw, part_db, nchunks, s))
      return rc;
    return launch_reduce_pair(part_dw, (long)din * dout, dw, part_db, dout, dbias, nchunks, s);
  }
  if (wgradx_ok(din, dout, (long)x_ld, (long)dy_ld) && m >= 4096) {
    // narrow input, wide output (81 -> 256): operands straight from memory into the f32 MFMA (wgradx.hip); one workgroup per
    // CU (four per CU: 101 instead of 94 us and a four times larger second stage)
    nchunks = kNumCU;
    float* part_dw = static_cast<float*>(workspace);
    float* part_db = part_dw + (long)nchunks * din * dout;
    if (int rc = launch_wgradx(x, (long)x_ld, dy, (long)dy_ld, (long)m, din, dout, part_dw, part_db, nchunks, s, yact, act))
      return rc;
    return launch_reduce_pair(part_dw, (long)din * dout, dw, part_db, dout, dbias, nchunks, s);
  }
  if (din > 64 && dout > 128) {
    // wide layers: bf16-split GEMM (gemm3.hip); one partial per row-range workgroup, <= kNumCU of them
    const int tiles = ((din + 127) / 128) * ((dout + 255) / 256);
    long nb = kNumCU / tiles;
    if (nb < 1) nb = 1;
    // at least 64 rows per row range: a range's partial is a whole [din x dout] block (256 KB at 256 x 256) -- with one 32-row
    // chunk per workgroup sparse.py's 4,457 rows wrote 128 partials = 32 MB per layer and the step's reduction launch read 119 MB
    // (profiles/r05_h_cfg3_rocprof.txt).  A/B on one box, cfg3 whole step: 32 rows 0.308 ms, 64 rows 0.301, 128 rows 0.318 (too few
    // workgroups), 256 rows 0.364
#ifndef KGCN_WGRAD_MIN_ROWS
#define KGCN_WGRAD_MIN_ROWS 64
#endif
    const long chunks = (m + KGCN_WGRAD_MIN_ROWS - 1) / KGCN_WGRAD_MIN_ROWS;
    if (nb > chunks) nb = chunks;
    float* part_dw = static_cast<float*>(workspace);
    float* part_db = part_dw + nb * din * dout;
    static const char* hknob = dev_knob("KGCN_GEMMH");         // development: see gemm3.hip ("w" = the f16 weight gradient)
    if (!(hknob && !strchr(hknob, 'w')) && gemmh_wgrad_ok(din, dout, (long)m)) {
      if (int rc = launch_gemmh_wgrad(x, (long)x_ld, dy, (long)dy_ld, (long)m, din, dout, part_dw, part_db, (int)nb, s, yact, act))
        return rc;
      return launch_reduce_pair(part_dw, (long)din * dout, dw, part_db, dout, dbias, (int)nb, s);
    }
    if (int rc = launch_gemm3_wgrad(x, (long)x_ld, dy, (long)dy_ld, (long)m, din, dout, part_dw, part_db, (int)nb, s, yact,
                                    act))
      return rc;
    return launch_reduce_pair(part_dw, (long)din * dout, dw, part_db, dout, dbias, (int)nb, s);
  }
  if (wgradn_ok(x, din, (long)x_ld, dout) && m >= 4096) {
    // wide input, narrow output (256 x 50): the whole dW block per wave on the bf16 pipe (wgradn.hip)
    nchunks = kNumCU;
    float* part_dw = static_cast<float*>(workspace);
    float* part_db = part_dw + (long)nchunks * din * dout;
    if (int rc = launch_wgradn(x, (long)x_ld, dy, (long)dy_ld, (long)m, din, dout, part_dw, part_db, nchunks, s)) return rc;
    return launch_reduce_pair(part_dw, (long)din * dout, dw, part_db, dout, dbias, nchunks, s);
  }
  if (narrow_wgrad_ok(x, din, (long)x_ld, dy, dout, (long)dy_ld)) {
    // 50-wide layers: flat tile movement (narrow.hip), one partial per workgroup
    nchunks = persist_blocks(m);
    float* part_dw = static_cast<float*>(workspace);
    float* part_db = part_dw + (long)nchunks * din * dout;
    if (int rc = launch_narrow_wgrad(x, dy, (long)m, din, dout, part_dw, part_db, nchunks, s)) return rc;
    return launch_reduce_pair(part_dw, (long)din * dout, dw, part_db, dout, dbias, nchunks, s);
  }
  {
    // persistent kernel: one workgroup (8 waves) per CU and per 64x64 output block
    nchunks = persist_blocks(m);
    const size_t lds = (size_t)P_WAVES * (2 * 32 * 64) * 4;      // >= P_WAVES * (64*64+64)*4 ? no: park needs more
    const size_t lds_park = (size_t)P_WAVES * (64 * 64 + 64) * 4;
    const size_t lds_use = lds > lds_park ? lds : lds_park;
    static thread_local bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_wgrad_persist_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_wgrad_persist_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
      attr_set = true;
    }
    float* part_dw = static_cast<float*>(workspace);
    float* part_db = part_dw + (long)nchunks * din * dout;
    const bool vec = (din % 4 == 0) && (dout % 4 == 0) && (x_ld % 4 == 0) && (dy_ld % 4 == 0) &&
                     aligned16(x) && aligned16(dy);
    dim3 grid((unsigned)nchunks, (unsigned)((din + 63) / 64), (unsigned)((dout + 63) / 64));
    if (vec)
      hipLaunchKernelGGL(dense_wgrad_persist_kernel<true>, grid, dim3(64 * P_WAVES), lds_use, s, x, (long)x_ld, dy,
                         (long)dy_ld, (long)m, din, dout, part_dw, part_db);
    else
      hipLaunchKernelGGL(dense_wgrad_persist_kernel<false>, grid, dim3(64 * P_WAVES), lds_use, s, x, (long)x_ld, dy,
                         (long)dy_ld, (long)m, din, dout, part_dw, part_db);
    if (int rc = check_launch("dense_wgrad_persist_kernel")) return rc;
    return launch_reduce_pair(part_dw, (long)din * dout, dw, part_db, dout, dbias, nchunks, s);
  }
  wgrad_plan(m, &rpc, &nchunks);
  float* part_dw = static_cast<float*>(workspace);
  float* part_db = part_dw + (long)nchunks * din * dout;
  dim3 grid((unsigned)nchunks, (unsigned)((din + 63) / 64), (unsigned)((dout + 63) / 64));
  hipLaunchKernelGGL(dense_wgrad_kernel, grid, dim3(256), 0, s, x, (long)x_ld, dy, (long)dy_ld,
                     (long)m, din, dout, rpc, part_dw, part_db);
  if (int rc = check_launch("dense_wgrad_kernel")) return rc;
  if (dw)
    if (int rc = reduce_or_defer(part_dw, nchunks, (long)din * dout, dw, s)) return rc;
  if (dbias)
    if (int rc = reduce_or_defer(part_db, nchunks, dout, dbias, s)) return rc;
  return 0;
}
