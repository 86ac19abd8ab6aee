// Weight gradient of a layer with a NARROW input and a wide output (65..96 -> 129..256 columns: 81 features -> 256 of
// example_model/model_multitask.py:51, with or without the aggregate-first ones column) on the f32 MFMA.
//
// dW [din x dout] = sum over the rows of x[r, :]^T (dy[r, :] (.) act'(a[r, :])) is 5 GFLOP over 118,000 rows but 280 MB of
// operands: HBM-bound (56 us at 5 TB/s).  The bf16-split kernel of gemm3.hip spends its time splitting both operands into
// three bf16 pieces -- (84 + 256) values per row against (256 + 256) of the square layer: 99 us against 115 us, 2.8 TB/s.
// v_mfma_f32_32x32x2_f32 takes its operands as they lie in memory: lane l of the A operand holds x[r + l / 32][32 mb + l % 32],
// lane l of the B operand dy[r + l / 32][32 w + l % 32] -- consecutive floats of two rows, one coalesced global load each,
// no LDS, no split, exact f32 products; the derivative of the activation is one multiply on the B operand.  At 64 cycles per
// MFMA and ceil(din / 32) <= 3 MFMAs per row pair and wave (wave w owns output columns 32 w .. 32 w + 31, all rows of dW) the
// matrix work would be 37 us, under the memory time.  Measured (profiles/r03_g_cfg4_rocprof.txt): 94 us = 3.2 TB/s -- 5 % under
// the split kernel, not the 40 % the arithmetic promises: more workgroups per CU, operand prefetch (ping-pong registers),
// scalar address arithmetic and wide loads (one dwordx3 + one dwordx2 per operand for six MFMAs: 105 us) all left the time
// where it is or worse.  What the measurements say: the loads alone (MFMAs replaced by plain FMAs) take 58 us = 5.2 TB/s, the
// 1.41 M MFMAs alone 38 us (27.5 ns each, tools/probes), together 96 us -- the sum, not the maximum -- with two or eight waves
// per SIMD and also with the next chunk's loads issued in front of the current chunk's MFMAs (verified in the ISA: partial
// vmcnt waits), and with the operands travelling global -> LDS directly (global_load_lds, no VGPR write on arrival: 110 us):
// while the f32 MFMA works, the memory stream does not; what holds the f32 MFMA at ~150 cycles per instruction here, while
// tools/probes/mfma_f32_probe2.hip issues one per 64 cycles from registers, is not understood.  Kept for the exact products.
// Eight row pairs of operands are requested before their MFMAs (two to four waves per SIMD; requesting the next chunk before
// the current chunk's MFMAs -- ping-pong registers -- was slower: 102 vs 94 us).  One partial per workgroup, reduce_partials adds them in a fixed order: deterministic.
#include "kgcn_common.h"

namespace kgcn {


constexpr int WX_U = 8;            // row pairs in flight per wave (2: 134 us, 4: 113 us, 8: 103 us, 16: 102 us with the second stage)

template <int MB, bool DACT>
__global__ __launch_bounds__(512) void wgradx_kernel(const float* __restrict__ x, long x_ld, const float* __restrict__ dy,
                                                     const float* __restrict__ yact, long dy_ld, long m, int din, int dout,
                                                     int act, long rows_per_block, float* __restrict__ part_dw,
                                                     float* __restrict__ part_db) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int n = 32 * wave + li;                        // this lane's output column
  const bool nok = n < dout;
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > m) r1 = m;
  f32x16 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[mb][i] = 0.f;
  float bsum = 0.f;
  bool kok[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) kok[mb] = 32 * mb + li < din;
  // The f32 MFMA shares the VALU datapath (tools/probes/mfma_f32_probe2.hip: every vector instruction between two MFMAs adds
  // its cycles to theirs), so the loop carries no vector address arithmetic: uniform row pointers (scalar registers) plus ONE
  // per-lane offset per operand, full chunks without row masks, and only the ragged last chunk of a workgroup masked.
  const long lane_x = (long)lh * x_ld + li, lane_y = (long)lh * dy_ld + n;
  struct Ops { float a[WX_U][MB], b[WX_U], ya[WX_U]; };
  auto mma = [&](Ops& o) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < WX_U; ++u) {
      if constexpr (DACT) o.b[u] *= act_dout(o.ya[u], act);
      bsum += o.b[u];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[u][mb], o.b[u], acc[mb], 0, 0, 0);
    }
  };
  const long step = 2 * WX_U;
  long r = r0;
  const long rfull = r1 < m - 1 ? r1 : m - 1;          // full (unmasked) chunks end below the tensors' last row
  auto load_full = [&](long rr, Ops& o) __attribute__((always_inline)) {
    const float* xr = x + rr * x_ld;
    const float* gr = dy + rr * dy_ld;
    const float* yr = DACT ? yact + rr * dy_ld : nullptr;
    // UNCONDITIONAL loads (a lane beyond the matrix edge reads into the next row -- the caller keeps full chunks away from the
    // last row of the tensors -- and is zeroed by a select afterwards): a masked load is a branch around the load, every branch a
    // basic block, and the counters are drained to zero at every join -- nothing stays in flight across the MFMAs
#pragma unroll
    for (int u = 0; u < WX_U; ++u) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) o.a[u][mb] = xr[2 * u * x_ld + lane_x + 32 * mb];
      o.b[u] = gr[2 * u * dy_ld + lane_y];
      if constexpr (DACT) o.ya[u] = yr[2 * u * dy_ld + lane_y];
    }
  };
  auto mask_full = [&](Ops& o) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < WX_U; ++u) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) o.a[u][mb] = kok[mb] ? o.a[u][mb] : 0.f;
      o.b[u] = nok ? o.b[u] : 0.f;
    }
  };
  for (; r + step <= rfull; r += step) {
    Ops o;
    load_full(r, o);
    mask_full(o);
    mma(o);
  }
  for (; r < r1; r += step) {
    Ops o;
#pragma unroll
    for (int u = 0; u < WX_U; ++u) {
      const long row = r + 2 * u + lh;
      const bool rok = row < r1;
      const float* xr = x + row * x_ld + li;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) o.a[u][mb] = (rok && kok[mb]) ? xr[32 * mb] : 0.f;
      o.b[u] = (rok && nok) ? dy[row * dy_ld + n] : 0.f;
      if constexpr (DACT) o.ya[u] = (rok && nok) ? yact[row * dy_ld + n] : 0.f;
    }
    mma(o);
  }
  float* pw = part_dw + (long)blockIdx.x * din * dout;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = 32 * mb + (i & 3) + 8 * (i >> 2) + 4 * lh;
      if (k < din && nok) pw[(long)k * dout + n] = acc[mb][i];
    }
  bsum += __shfl_xor(bsum, 32, 64);                    // the two row parities of a column
  if (lh == 0 && nok) part_db[(long)blockIdx.x * dout + n] = bsum;
}

bool wgradx_ok(int din, int dout, long x_ld, long dy_ld) { return din > 64 && din <= 96 && dout > 128 && dout <= 256; }

int launch_wgradx(const float* x, long x_ld, const float* dy, long dy_ld, long m, int din, int dout, float* part_dw, float* part_db,
                  int nparts, hipStream_t s, const float* yact, int act) {
  long rpb = (m + nparts - 1) / nparts;
  rpb = (rpb + 1) & ~1L;                               // row pairs never straddle two workgroups
  const int mb = (din + 31) / 32;
  if (mb != 3) return fail("wgradx: %d input columns", din);     // 65..96: three 32-row blocks of dW
  if (yact && act != KGCN_ACT_NONE)
    hipLaunchKernelGGL((wgradx_kernel<3, true>), dim3(nparts), dim3(512), 0, s, x, x_ld, dy, yact, dy_ld, m, din, dout, act, rpb,
                       part_dw, part_db);
  else
    hipLaunchKernelGGL((wgradx_kernel<3, false>), dim3(nparts), dim3(512), 0, s, x, x_ld, dy, yact, dy_ld, m, din, dout, act, rpb,
                       part_dw, part_db);
  return check_launch("wgradx_kernel");
}

}  // namespace kgcn
