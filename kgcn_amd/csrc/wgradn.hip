// Weight gradient of a layer with a WIDE input and a NARROW output (din 128..256, dout <= 64: GraphConv(50) behind the
// 256-wide layers, example_model/model_multitask.py:57) on the bf16 matrix pipe:
//
//   dW[din, dout] = x^T @ dy,   dbias[dout] = colsum(dy)           (per-workgroup partials, reduced in fixed order)
//
// Shipped: wgradnb_kernel (below) -- the register-split form of wgradx.hip with the roles exchanged; 49 us per call incl. the
// second stage at m = 117,888 (the kernel above it in the file: 58; cfg4 step 1.477 against 1.491 ms).
// DEV_KNOBS builds keep the first kernel (KGCN_WGRADN=lds) for A/B:
// The f32-MFMA kernel of dense.hip cuts din into four 64-column blocks (each re-reading dy, x in 256-byte segments) and
// spends 256 MFMAs of 64 cycles per 32 rows on a pipe it shares with the VALU: 146 us at m = 204,800 (HBM time 31 us).
// There a wave owns a [128 x 64] half of dW in 128 accumulator registers, two waves per SIMD (the two column halves of the
// same rows) hide each other's vector work behind the other's MFMAs:
//   * the batch rows are the MFMA K dimension, 16 rows per k-step with k = (row parity, row / 2) -- so that the lane layout
//     of a coalesced load IS the fragment layout: a load instruction fetches two whole 512-byte half rows (lanes 0-31 the
//     even row, 32-63 the odd one), eight of them give a lane 4 columns x 8 rows of its parity = four 8-k fragments after
//     an exact 3-way bf16 split;
//   * the fragments go through the wave's own 12 KB of LDS into operand order (entries XOR-rotated by the m-tile: 4 lanes
//     per bank group, the optimum for a 1 KiB write) and are read back at the start of the k-step -- wave-local in-order LDS
//     traffic, no barrier;
//   * dy is loaded lane = column, the 8 rows of the lane's parity: B fragments without any data movement; the column-half-0
//     waves sum it for dbias;
//   * the row-range groups of a workgroup are summed through LDS (two rounds), one [din x dout] partial per workgroup.
#include "kgcn_common.h"

namespace kgcn {

#ifdef KGCN_DEV_KNOBS
constexpr int WN_LDS_WAVE = 4 * 3 * 64 * 16;               // (m-tile, piece) x 64 lanes x 16 bytes = 12 KB
constexpr int WN_WAVES = 8;                                 // 4 row-range groups x 2 column halves

__global__ __launch_bounds__(512, 1) void wgradn_kernel(const float* __restrict__ x, long x_ld,
                                                        const float* __restrict__ dy, long dy_ld, long m, int din,
                                                        int dout, float* __restrict__ part_dw, float* __restrict__ part_db) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wn_smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 31, hi = lane >> 5;
  const int grp = wave >> 1, ch = wave & 1;                  // row-range group, column half (x columns 128 ch ..)
  unsigned char* xp = wn_smem + (size_t)wave * WN_LDS_WAVE;
  const long nsteps = (m + 15) / 16;                         // k-steps of 16 rows
  const long ngroups = (long)gridDim.x * 4;
  const long gg = (long)blockIdx.x * 4 + grp;
  const long per = (nsteps + ngroups - 1) / ngroups;         // contiguous range of k-steps per group
  const long s_begin = gg * per;
  long s_end = s_begin + per;
  if (s_end > nsteps) s_end = nsteps;

  // x staging: lane = (row parity hi, 16-byte column group li): columns 128 ch + 4 li .. + 3 (beyond din: clamped, masked)
  const int c0x = 128 * ch + 4 * li;
  const bool xok = c0x < din;
  const int xcol = xok ? c0x : 0;
  const int mt_w = li >> 3;
  // entry of column e inside its m-tile block, XOR-rotated by the m-tile: 32 hi + 4 (li & 7) + (e ^ (mt & 3))
  const int wr_base = mt_w * 3072 + (32 * hi + (li & 7) * 4) * 16;
  // dy staging: column 32 nt + li (clamped / masked), rows of parity hi
  const bool dok0 = li < dout, dok1 = 32 + li < dout;
  const int dc0 = dok0 ? li : 0, dc1 = dok1 ? 32 + li : 0;

  f32x16 acc[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  float cs0 = 0.f, cs1 = 0.f;

  f32x4 rawx[8];                                             // rows 2 j + hi of the k-step
  float rawd[2][8];                                          // dy[row 2 j + hi][column of tile nt]
  auto load_step = [&](long s) __attribute__((always_inline)) {
    const long row0 = s * 16 + hi;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long r = row0 + 2 * j;
      const long rc = r < m ? r : m - 1;
      rawx[j] = *reinterpret_cast<const f32x4*>(x + rc * x_ld + xcol);
      const float* dr = dy + rc * dy_ld;
      rawd[0][j] = dr[dc0];
      rawd[1][j] = dr[dc1];
    }
  };
  // rawx -> pieces in LDS (fragment order); rawd -> B fragments
  auto stage = [&](long s, Frag3 (&B)[2]) __attribute__((always_inline)) {
    const long row0 = s * 16 + hi;
    const bool full = row0 + 14 < m;                        // every row of this lane's parity exists (uniform per half wave)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (xok && (full || row0 + 2 * j < m)) ? rawx[j][e] : 0.f;
      Frag3 f;
      split8(v, f);
      unsigned char* d = xp + wr_base + (e ^ (mt_w & 3)) * 16;
      *reinterpret_cast<u32x4*>(d) = f.p1;
      *reinterpret_cast<u32x4*>(d + 1024) = f.p2;
      *reinterpret_cast<u32x4*>(d + 2048) = f.p3;
    }
    float v0[8], v1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool rok = full || row0 + 2 * j < m;
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
    stage(s_begin, B);
    if (s_begin + 1 < s_end) load_step(s_begin + 1);
    for (long s = s_begin; s < s_end; ++s) {
      // A fragments of this k-step -> registers (then the LDS buffer is free for the next one)
      u32x4 A[4][3];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
          A[mt][pc] = *reinterpret_cast<const u32x4*>(xp + (mt * 3 + pc) * 1024 + (32 * hi + (li ^ (mt & 3))) * 16);
      static_for<48>([&](auto sc) __attribute__((always_inline)) {
        constexpr int q = decltype(sc)::value, mt = q / 12, r12 = q % 12, pr = r12 >> 1, nt = r12 & 1;
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};   // smallest terms first
        const u32x4 bv = PB[pr] == 0 ? B[nt].p1 : PB[pr] == 1 ? B[nt].p2 : B[nt].p3;
        acc[mt][nt] = mfma_bf16(A[mt][PA[pr]], bv, acc[mt][nt]);
      });
      // the next k-step: pieces -> LDS (its A reads above are through: in-order LDS) / B registers; the one after that is
      // requested.  The other wave of the SIMD issues its MFMAs meanwhile.
      if (s + 1 < s_end) {
        stage(s + 1, B);
        if (s + 2 < s_end) load_step(s + 2);
      }
    }
  }

  // ---- the four row-range groups summed through LDS: groups 2, 3 -> 0, 1; group 1 -> 0 -----------------------------------
  cs0 += __shfl_xor(cs0, 32, 64);
  cs1 += __shfl_xor(cs1, 32, 64);
  float* red = reinterpret_cast<float*>(wn_smem);            // four slabs of [4][2][16][64] floats (32 KB each)
  float* csr = red + 4 * 8192;                               // [4 groups][64] column sums (column-half-0 waves)
  if (ch == 0 && hi == 0) { csr[grp * 64 + li] = cs0; csr[grp * 64 + 32 + li] = cs1; }
  __syncthreads();                                           // every wave is through with its fragment region
  auto put = [&](float* slab) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[((mt * 2 + nt) * 16 + r) * 64 + lane] = acc[mt][nt][r];
  };
  auto add = [&](const float* slab) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] += slab[((mt * 2 + nt) * 16 + r) * 64 + lane];
  };
  if (grp >= 2) put(red + ((grp - 2) * 2 + ch) * 8192);
  __syncthreads();
  if (grp < 2) add(red + (grp * 2 + ch) * 8192);
  __syncthreads();
  if (grp == 1) put(red + ch * 8192);
  __syncthreads();
  if (grp == 0) {
    add(red + ch * 8192);
    float* pw = part_dw + (long)blockIdx.x * din * dout;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int col = 32 * nt + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 128 * ch + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (row < din && col < dout) pw[(long)row * dout + col] = acc[mt][nt][r];
        }
      }
    if (part_db && ch == 0 && lane < dout)
      part_db[(long)blockIdx.x * dout + lane] = (csr[lane] + csr[64 + lane]) + (csr[128 + lane] + csr[192 + lane]);
  }
}

#endif  // KGCN_DEV_KNOBS

// The register-split form (wgradx.hip: wgradxb_kernel) with the roles exchanged: wave w owns x block w (32 of the up to 256
// input columns -- its own slice of the HBM stream, 8 coalesced row loads per 16 rows) and BOTH dy blocks (64 output columns,
// the narrow operand: every wave of the workgroup loads and splits it, L1 / L2 hits); no LDS, no reduction across row-range
// groups, one partial per workgroup.  k = (row parity, row / 2): the lane layout of the loads is the MFMA operand layout.
__global__ __launch_bounds__(512, 2) void wgradnb_kernel(const float* __restrict__ x, long x_ld, const float* __restrict__ dy,
                                                         long dy_ld, long m, int din, int dout, long steps_per_block,
                                                         float* __restrict__ part_dw, float* __restrict__ part_db) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, hi = lane >> 5;
  const int i0 = 32 * wave;                            // first x column (dW row) of this wave
  const int kcol = i0 + li < din ? i0 + li : 0;
  int ncol[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) ncol[nb] = 32 * nb + li < dout ? 32 * nb + li : 0;
  const long nsteps = (m + 15) / 16;
  const long s0 = (long)blockIdx.x * steps_per_block;
  long s1 = s0 + steps_per_block;
  if (s1 > nsteps) s1 = nsteps;
  f32x16 acc[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
  float bsum[2] = {0.f, 0.f};
  struct Raw { float a[8], b[2][8]; };
  // uniform row pointers + one per-lane offset per operand block; full steps need no row clamp and no masks at all (columns
  // beyond din / dout land in accumulator rows / columns that are never stored); the ragged last step of the tensors is peeled off
  const unsigned offx = (unsigned)(hi * x_ld + kcol);
  unsigned offy[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) offy[nb] = (unsigned)(hi * dy_ld + ncol[nb]);
  auto load = [&](long s, Raw& r) __attribute__((always_inline)) {
    const float* xs = x + s * 16 * x_ld;
    const float* gs = dy + s * 16 * dy_ld;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      r.a[j] = xs[2 * j * x_ld + offx];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) r.b[nb][j] = gs[2 * j * dy_ld + offy[nb]];
    }
  };
  auto load_tail = [&](long s, Raw& r) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      long row = s * 16 + 2 * j + hi;
      row = row < m ? row : m - 1;
      r.a[j] = x[row * x_ld + kcol];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) r.b[nb][j] = dy[row * dy_ld + ncol[nb]];
    }
  };
  auto mma = [&](long s, Raw& r, auto tailc) __attribute__((always_inline)) {
    constexpr bool TAIL = decltype(tailc)::value;
    Frag3 fa, fb[2];
    {
      float u[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) u[j] = (!TAIL || s * 16 + 2 * j + hi < m) ? r.a[j] : 0.f;
      split8(u, fa);
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j] = (!TAIL || s * 16 + 2 * j + hi < m) ? r.b[nb][j] : 0.f;
        bsum[nb] += v[j];
      }
      split8(v, fb[nb]);
    }
#define KGCN_WNB(PA, PB)                             \
  acc[0] = mfma_bf16(fa.PA, fb[0].PB, acc[0]);      \
  acc[1] = mfma_bf16(fa.PA, fb[1].PB, acc[1]);
    KGCN_SPLIT_PRODUCTS(KGCN_WNB)
#undef KGCN_WNB
  };
  const bool ragged_last = (m % 16 != 0) && s1 == nsteps && s0 < s1;
  const long s1f = ragged_last ? s1 - 1 : s1;
  if (s0 < s1f) {
    // one step of lookahead; sched_barrier keeps the requests of step s + 1 in front of the arithmetic of step s
    Raw r0, r1;
    const long sl = s1f - 1;
    load(s0, r0);
    long s = s0;
    for (; s + 1 < s1f; s += 2) {
      load(s + 1, r1);
      __builtin_amdgcn_sched_barrier(0);
      mma(s, r0, std::false_type{});
      __builtin_amdgcn_sched_barrier(0);
      load(s + 2 < sl ? s + 2 : sl, r0);
      __builtin_amdgcn_sched_barrier(0);
      mma(s + 1, r1, std::false_type{});
      __builtin_amdgcn_sched_barrier(0);
    }
    if (s < s1f) mma(s, r0, std::false_type{});
  }
  if (ragged_last) {
    Raw rt;
    load_tail(s1 - 1, rt);
    mma(s1 - 1, rt, std::true_type{});
  }
  float* pw = part_dw + (long)blockIdx.x * din * dout;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int n = 32 * nb + li;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = i0 + (i & 3) + 8 * (i >> 2) + 4 * hi;
      if (k < din && n < dout) pw[(long)k * dout + n] = acc[nb][i];
    }
    if (part_db && wave == 0) {
      const float b = bsum[nb] + __shfl_xor(bsum[nb], 32, 64);      // the two row parities of a column
      if (hi == 0 && n < dout) part_db[(long)blockIdx.x * dout + n] = b;
    }
  }
}

bool wgradn_ok(const float* x, int din, long x_ld, int dout) {
  return dout <= 64 && din >= 128 && din <= 256 && din % 4 == 0 && x_ld % 4 == 0 && aligned16(x);
}

// nblocks partials ([nblocks][din*dout], [nblocks][dout]); nblocks <= kNumCU
int launch_wgradn(const float* x, long x_ld, const float* dy, long dy_ld, long m, int din, int dout, float* part_dw,
                  float* part_db, int nblocks, hipStream_t s) {
#ifdef KGCN_DEV_KNOBS
  static const char* route = dev_knob("KGCN_WGRADN");            // development: "lds" = the kernel with the LDS transposition
  const bool lds_route = route && route[0] == 'l';
#else
  const bool lds_route = false;
#endif
  if (!lds_route) {
    const long nsteps = (m + 15) / 16, spb = (nsteps + nblocks - 1) / nblocks;
    hipLaunchKernelGGL(wgradnb_kernel, dim3((unsigned)nblocks), dim3(512), 0, s, x, x_ld, dy, dy_ld, m, din, dout, spb, part_dw,
                       part_db);
    return check_launch("wgradnb_kernel");
  }
#ifdef KGCN_DEV_KNOBS
  const size_t frag_b = WN_WAVES * (size_t)WN_LDS_WAVE, red_b = (size_t)(4 * 8192 + 256) * 4;
  const size_t lds = frag_b > red_b ? frag_b : red_b;
  static thread_local bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgradn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              kLdsBytes);
    attr_set = true;
  }
  hipLaunchKernelGGL(wgradn_kernel, dim3((unsigned)nblocks), dim3(64 * WN_WAVES), lds, s, x, x_ld, dy, dy_ld, m, din, dout,
                     part_dw, part_db);
  return check_launch("wgradn_kernel");
#else
  return fail("wgradn: no route");
#endif
}

}  // namespace kgcn
