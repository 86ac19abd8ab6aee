// fp32 GEMM on the bf16 matrix pipe for layers with a WIDE input and a NARROW output (din >= 128, dout <= 64: the
// 256 -> 50 contraction of GraphConv(50) behind the 256-wide layers, example_model/model_multitask.py:57):
//
//   y[m, dout] = act(x[m, din] @ W + bias)
//
// With one 64-column block there is nothing to share between waves (gemm3.hip shares the split x pieces of a row tile
// between FOUR column waves), and the f32-MFMA kernel of dense.hip needs 256 MFMAs of 64 cycles per 32 rows on a pipe it
// shares with the VALU (107 us at m = 204,800; HBM time 31 us).  Here every wave owns [64 rows x 64 columns] by itself:
//   * x arrives in coalesced row segments (a lane loads 16 bytes, eight lanes one 128-byte line, one instruction eight
//     rows of a 32-wide k chunk), is split in registers (exact 3-way bf16 split, kgcn_common.h) and goes through the wave's
//     OWN 12 KB of LDS into MFMA fragment order -- wave-local, in-order LDS traffic: no barrier anywhere.  (Loading the
//     fragments directly, lane = row, made every request a lone 64-byte piece of 131k concurrent row streams: HBM-page
//     bound at 105 us, profiles/r02_gemm_experiments.txt.)
//   * the fragments of a chunk are read into registers at its start, so the single LDS buffer is free for the pieces of the
//     next chunk, which are written behind the MFMAs of this one; the raw x of the chunk after that is already in flight;
//   * W comes pre-split from the fragment table of wtable.hip (L2 / L1 resident), one k-step ahead, straight into registers;
//   * two waves per SIMD (<= 256 registers) hide each other's vector work; bias is the initial accumulator value, the
//     activation is applied in registers, a lane owns one COLUMN of its tiles.
#include "kgcn_common.h"

namespace kgcn {

constexpr int GN_BM = 64, GN_BN = 64, GN_BK = 32;
constexpr int GN_LDS_WAVE = 2 * 2 * 3 * 64 * 16;            // (m-tile, k-step, piece) x 64 lanes x 16 bytes = 12 KB

template <bool KMASK>
__global__ __launch_bounds__(256, 2) void gemmn_fwd_kernel(const float* __restrict__ x, long m, int din, long x_ld,
                                                           const u32x4* __restrict__ table, int tab_nt, int tab_ks,
                                                           const float* __restrict__ bias, float* __restrict__ y, int dout,
                                                           long y_ld, int act) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gn_smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 31, hi = lane >> 5;
  unsigned char* xp = gn_smem + (size_t)wave * GN_LDS_WAVE;
  const int ncb = (dout + GN_BN - 1) / GN_BN;               // column blocks
  const long nrb = (m + GN_BM - 1) / GN_BM;                 // row blocks
  const long items = nrb * ncb;
  const int nchunks = (din + GN_BK - 1) / GN_BK;
  const long nwaves = (long)gridDim.x * 4;
  // staging coordinates of this lane: 16-byte group c8 of the chunk's 128-byte row segment, rows r8 + 8 i (i = 0..7)
  const int c8 = lane & 7, r8 = lane >> 3;
  // piece write address of load i: (m-tile i >> 2, k-step c8 >> 2, piece) block, entry 32 (c8 >> 1 & 1) + r8 + 8 (i & 3),
  // half c8 & 1 of the entry
  const int wr_lane = (c8 >> 2) * 3072 + ((c8 >> 1) & 1) * 512 + r8 * 16 + (c8 & 1) * 8;

  for (long item = (long)blockIdx.x * 4 + wave; item < items; item += nwaves) {
    const long rb = item / ncb;
    const int cb = (int)(item - rb * ncb);
    const long row0 = rb * GN_BM;

    f32x16 acc[2][2];                                       // [nt][mt]: lane (li, hi) = column 32 nt + li
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int c = GN_BN * cb + 32 * nt + li;
      const float bv = (bias && c < dout) ? bias[c] : 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][mt][r] = bv;
    }
    // this lane's eight rows, clamped (rows beyond m are never stored), at its 16-byte column group
    const float* xrow[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long r = row0 + r8 + 8 * i;
      xrow[i] = x + (r < m ? r : m - 1) * x_ld + 4 * c8;
    }
    const u32x4* tb = table + ((long)(2 * cb) * 3) * 64 + lane;       // + ks * tab_nt * 192

    f32x4 raw[8];                                           // raw x of the chunk after the one being multiplied
    u32x4 B0[2][3], B1[2][3];                               // W fragments of the chunk's two k-steps
    auto load_raw1 = [&](int i, int chunk) __attribute__((always_inline)) {
      const int k = GN_BK * chunk + 4 * c8;
      const int off = (!KMASK || k < din) ? GN_BK * chunk : 0;          // clamped: masked when split
      raw[i] = *reinterpret_cast<const f32x4*>(xrow[i] + off);
    };
    auto load_raw = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 8; ++i) load_raw1(i, chunk);
    };
    auto load_b = [&](u32x4 (&Bd)[2][3], int g) __attribute__((always_inline)) {
      const int gg = g < tab_ks ? g : tab_ks - 1;           // a k-step beyond the table meets zero x pieces
      const u32x4* p = tb + (long)gg * tab_nt * 192;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) Bd[nt][pc] = p[(nt * 3 + pc) * 64];
    };
    // raw[i] -> three 8-byte piece halves in fragment order
    auto split_store = [&](int i, int chunk) __attribute__((always_inline)) {
      f32x4 v = raw[i];
      if constexpr (KMASK) {
        const int k = GN_BK * chunk + 4 * c8;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = k + j < din ? v[j] : 0.f;
      }
      unsigned a1, a2, a3, b1, b2, b3;
      split_pair(v[0], v[1], a1, a2, a3);
      split_pair(v[2], v[3], b1, b2, b3);
      unsigned char* d = xp + wr_lane + (i >> 2) * 6144 + (i & 3) * 128;
      *reinterpret_cast<u32x2*>(d) = u32x2{a1, b1};
      *reinterpret_cast<u32x2*>(d + 1024) = u32x2{a2, b2};
      *reinterpret_cast<u32x2*>(d + 2048) = u32x2{a3, b3};
    };

    // prologue: chunk 0 split into LDS, chunk 1 in flight, W fragments of chunk 0
    load_raw(0);
    load_b(B0, 0);
    load_b(B1, 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) split_store(i, 0);
    load_raw(nchunks > 1 ? 1 : 0);

    for (int c = 0; c < nchunks; ++c) {
      // fragments of chunk c -> registers (the LDS buffer is free again once these reads are through: the LDS executes a
      // wave's operations in order)
      u32x4 A[2][2][3];                                     // [k-step][m-tile][piece]
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int pc = 0; pc < 3; ++pc)
            A[ks][mt][pc] = *reinterpret_cast<const u32x4*>(xp + ((mt * 2 + ks) * 3 + pc) * 1024 + lane * 16);
      const int cn = c + 1 < nchunks ? c + 1 : c;           // chunk being split (a re-split of c at the very end)
      const int cl = c + 2 < nchunks ? c + 2 : cn;          // chunk being requested
      static_for<48>([&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value, ks = s / 24, q = s % 24, pr = q >> 2, nt = (q >> 1) & 1, mt = q & 1;
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};   // smallest terms first
        const u32x4 bv = ks == 0 ? B0[nt][PB[pr]] : B1[nt][PB[pr]];
        acc[nt][mt] = mfma_bf16(A[ks][mt][PA[pr]], bv, acc[nt][mt]);
        if constexpr (s >= 4 && s < 36 && (s & 3) == 0) {
          split_store((s - 4) >> 2, cn);                    // one raw register (two pairs) every fourth MFMA ...
          load_raw1((s - 4) >> 2, cl);                      // ... and straight away its refill: a whole chunk of lead
        } else if constexpr (s == 25) {
          load_b(B0, 2 * (c + 1));                          // k-step 0 of this chunk is through with B0
        }
      });
      load_b(B1, 2 * (c + 1) + 1);
    }

    if (act != KGCN_ACT_NONE) {                             // one uniform branch around the whole activation
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[nt][mt][r] = act_fwd(acc[nt][mt][r], act);
    }
    const int c0 = GN_BN * cb + li, c1 = c0 + 32;
    if (row0 + GN_BM <= m && GN_BN * cb + GN_BN <= dout) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        float* p = y + (row0 + 32 * mt + 4 * hi) * y_ld + c0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          p[0] = acc[0][mt][r];
          p[32] = acc[1][mt][r];
          p += ((r & 3) == 3) ? 5 * y_ld : y_ld;              // rows 0-3, 8-11, 16-19, 24-27 (+ 4 hi)
        }
      }
    } else {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long row = row0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (row < m) {
            if (c0 < dout) y[row * y_ld + c0] = acc[0][mt][r];
            if (c1 < dout) y[row * y_ld + c1] = acc[1][mt][r];
          }
        }
    }
  }
}

bool gemmn_pays(const float* x, int din, long x_ld, int dout) {
  return dout <= 64 && din >= 128 && din % 4 == 0 && x_ld % 4 == 0 && aligned16(x);
}

// `table`: fragment table of wtable.hip for (w, trans_w)
int launch_gemmn_fwd(const float* x, long m, int din, long x_ld, const void* table, const float* bias, float* y, int dout,
                     long y_ld, int act, hipStream_t s) {
  const long items = ((m + GN_BM - 1) / GN_BM) * ((dout + GN_BN - 1) / GN_BN);
  long blocks = (items + 3) / 4;
  if (blocks > 2 * kNumCU) blocks = 2 * kNumCU;
  const int tab_nt = ((dout + 63) / 64) * 2, tab_ks = (din + 15) / 16;
  const u32x4* tab = static_cast<const u32x4*>(table);
  const size_t lds = 4 * (size_t)GN_LDS_WAVE;
  if (din % GN_BK == 0)
    hipLaunchKernelGGL((gemmn_fwd_kernel<false>), dim3((unsigned)blocks), dim3(256), lds, s, x, m, din, x_ld, tab, tab_nt,
                       tab_ks, bias, y, dout, y_ld, act);
  else
    hipLaunchKernelGGL((gemmn_fwd_kernel<true>), dim3((unsigned)blocks), dim3(256), lds, s, x, m, din, x_ld, tab, tab_nt,
                       tab_ks, bias, y, dout, y_ld, act);
  return check_launch("gemmn_fwd_kernel");
}

}  // namespace kgcn
