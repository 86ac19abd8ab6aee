// Weight gradient of a layer with a WIDE input and a NARROW output (din 128..256, dout <= 64: GraphConv(50) behind the
// 256-wide layers, example_model/model_multitask.py:57) on the bf16 matrix pipe:
//
//   dW[din, dout] = x^T @ dy,   dbias[dout] = colsum(dy)           (per-workgroup partials, reduced in fixed order)
//
// The f32-MFMA kernel of dense.hip cuts din into four 64-column blocks (each re-reading dy, x in 256-byte segments) and
// spends 256 MFMAs of 64 cycles per 32 rows on a pipe it shares with the VALU: 146 us at m = 204,800 (HBM time 31 us).
// Here one wave (one per SIMD, 512 registers) owns the WHOLE [256 x 64] block of dW in 256 accumulator registers:
//   * the batch rows are the MFMA K dimension, 16 rows per k-step; x rows are loaded WHOLE (one 1 KiB row per instruction,
//     a lane holds 4 columns of 8 consecutive rows = four 8-k fragments after an exact 3-way bf16 split), written to the
//     wave's own 24 KB of LDS in fragment order (XOR-rotated by the m-tile: 4 lanes per bank group, the optimum for a
//     1 KiB write) and read back as A fragments at the start of the k-step -- wave-local in-order LDS traffic, no barrier;
//   * dy is loaded lane = column, 8 consecutive rows per lane: B fragments without any data movement; its column sums are
//     accumulated on the way (dbias);
//   * the four waves of a workgroup are summed through LDS (two rounds), one [din x dout] partial per workgroup.
#include "kgcn_common.h"

namespace kgcn {

constexpr int WN_LDS_WAVE = 8 * 3 * 64 * 16;               // (m-tile, piece) x 64 lanes x 16 bytes = 24 KB

__global__ __launch_bounds__(256, 1) void wgradn_kernel(const float* __restrict__ x, long x_ld,
                                                        const float* __restrict__ dy, long dy_ld, long m, int din,
                                                        int dout, float* __restrict__ part_dw, float* __restrict__ part_db) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wn_smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 31, hi = lane >> 5;
  unsigned char* xp = wn_smem + (size_t)wave * WN_LDS_WAVE;
  const long nsteps = (m + 15) / 16;                         // k-steps of 16 rows
  const long nwaves = (long)gridDim.x * 4;
  const long gw = (long)blockIdx.x * 4 + wave;
  // contiguous range of k-steps per wave
  const long per = (nsteps + nwaves - 1) / nwaves;
  const long s_begin = gw * per;
  long s_end = s_begin + per;
  if (s_end > nsteps) s_end = nsteps;

  // x staging: this lane's 4 columns (4 lane .. 4 lane + 3; beyond din: clamped, masked when split), m-tile lane >> 3
  const int xcol = 4 * lane < din ? 4 * lane : 0;
  const bool xok = 4 * lane < din;
  const int mt_w = lane >> 3;
  // entry of column e of this lane inside its m-tile block, XOR-rotated by the m-tile: 4 (lane & 7) + (e ^ (mt & 3))
  const int wr_base = mt_w * 3072 + ((lane & 7) * 4) * 16;
  // dy staging: column 32 nt + li (clamped / masked)
  const bool dok0 = li < dout, dok1 = 32 + li < dout;
  const int dc0 = dok0 ? li : 0, dc1 = dok1 ? 32 + li : 0;

  f32x16 acc[8][2];
#pragma unroll
  for (int mt = 0; mt < 8; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  float cs0 = 0.f, cs1 = 0.f;

  f32x4 rawx[16];                                            // rows 8 g + j of the k-step: rawx[8 g + j]
  float rawd[2][8];                                          // dy[row 8 hi + j][column of tile nt]
  auto load_step = [&](long s) __attribute__((always_inline)) {
    const long row0 = s * 16;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const long r = row0 + q;
      rawx[q] = *reinterpret_cast<const f32x4*>(x + (r < m ? r : m - 1) * x_ld + xcol);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long r = row0 + 8 * hi + j;
      const float* dr = dy + (r < m ? r : m - 1) * dy_ld;
      rawd[0][j] = dr[dc0];
      rawd[1][j] = dr[dc1];
    }
  };
  // rawx -> pieces in LDS (fragment order); rawd -> B fragments
  auto stage_x = [&](long s) __attribute__((always_inline)) {
    const long row0 = s * 16;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (xok && row0 + 8 * g + j < m) ? rawx[8 * g + j][e] : 0.f;
        Frag3 f;
        split8(v, f);
        unsigned char* d = xp + wr_base + ((e ^ (mt_w & 3)) + 32 * g) * 16;
        *reinterpret_cast<u32x4*>(d) = f.p1;
        *reinterpret_cast<u32x4*>(d + 1024) = f.p2;
        *reinterpret_cast<u32x4*>(d + 2048) = f.p3;
      }
  };
  auto stage_d = [&](long s, Frag3 (&B)[2]) __attribute__((always_inline)) {
    const long row0 = s * 16 + 8 * hi;
    float v0[8], v1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool rok = row0 + j < m;
      v0[j] = (dok0 && rok) ? rawd[0][j] : 0.f;
      v1[j] = (dok1 && rok) ? rawd[1][j] : 0.f;
      cs0 += v0[j];
      cs1 += v1[j];
    }
    split8(v0, B[0]);
    split8(v1, B[1]);
  };

  if (s_begin < s_end) {
    Frag3 B[2];
    load_step(s_begin);
    stage_x(s_begin);
    stage_d(s_begin, B);
    if (s_begin + 1 < s_end) load_step(s_begin + 1);
    for (long s = s_begin; s < s_end; ++s) {
      // A fragments of this k-step -> registers (then the LDS buffer is free for the next one)
      u32x4 A[8][3];
#pragma unroll
      for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
          A[mt][pc] = *reinterpret_cast<const u32x4*>(xp + (mt * 3 + pc) * 1024 + (32 * hi + (li ^ (mt & 3))) * 16);
      static_for<96>([&](auto sc) __attribute__((always_inline)) {
        constexpr int q = decltype(sc)::value, mt = q / 12, r12 = q % 12, pr = r12 >> 1, nt = r12 & 1;
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};   // smallest terms first
        const u32x4 bv = PB[pr] == 0 ? B[nt].p1 : PB[pr] == 1 ? B[nt].p2 : B[nt].p3;
        acc[mt][nt] = mfma_bf16(A[mt][PA[pr]], bv, acc[mt][nt]);
      });
      // the next k-step: pieces -> LDS (its A reads above are through: in-order LDS) / B registers; the one after that is
      // requested (a whole k-step of lead)
      if (s + 1 < s_end) {
        stage_x(s + 1);
        stage_d(s + 1, B);
        if (s + 2 < s_end) load_step(s + 2);
      }
    }
  }

  // ---- the four waves' blocks summed through LDS: waves 2, 3 -> waves 0, 1; wave 1 -> wave 0 ------------------------
  cs0 += __shfl_xor(cs0, 32, 64);
  cs1 += __shfl_xor(cs1, 32, 64);
  float* red = reinterpret_cast<float*>(wn_smem);            // two slabs of [8][2][16][64] floats (64 KB each)
  float* csr = red + 2 * 16384;                              // [4][64] column sums
  if (hi == 0) { csr[wave * 64 + li] = cs0; csr[wave * 64 + 32 + li] = cs1; }
  __syncthreads();                                           // every wave is through with its fragment region
  auto put = [&](float* slab) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[((mt * 2 + nt) * 16 + r) * 64 + lane] = acc[mt][nt][r];
  };
  auto add = [&](const float* slab) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] += slab[((mt * 2 + nt) * 16 + r) * 64 + lane];
  };
  if (wave >= 2) put(red + (wave - 2) * 16384);
  __syncthreads();
  if (wave < 2) add(red + wave * 16384);
  __syncthreads();
  if (wave == 1) put(red);
  __syncthreads();
  if (wave == 0) {
    add(red);
    float* pw = part_dw + (long)blockIdx.x * din * dout;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int col = 32 * nt + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (row < din && col < dout) pw[(long)row * dout + col] = acc[mt][nt][r];
        }
      }
    if (part_db && lane < dout)
      part_db[(long)blockIdx.x * dout + lane] = (csr[lane] + csr[64 + lane]) + (csr[128 + lane] + csr[192 + lane]);
  }
}

bool wgradn_ok(const float* x, int din, long x_ld, int dout) {
  return dout <= 64 && din >= 128 && din <= 256 && din % 4 == 0 && x_ld % 4 == 0 && aligned16(x);
}

// nblocks partials ([nblocks][din*dout], [nblocks][dout]); nblocks <= kNumCU
int launch_wgradn(const float* x, long x_ld, const float* dy, long dy_ld, long m, int din, int dout, float* part_dw,
                  float* part_db, int nblocks, hipStream_t s) {
  const size_t frag_b = 4 * (size_t)WN_LDS_WAVE, red_b = (size_t)(2 * 16384 + 256) * 4;
  const size_t lds = frag_b > red_b ? frag_b : red_b;
  static thread_local bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgradn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              kLdsBytes);
    attr_set = true;
  }
  hipLaunchKernelGGL(wgradn_kernel, dim3((unsigned)nblocks), dim3(256), lds, s, x, x_ld, dy, dy_ld, m, din, dout, part_dw,
                     part_db);
  return check_launch("wgradn_kernel");
}

}  // namespace kgcn
