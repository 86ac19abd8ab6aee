// Weight gradient of a layer with a NARROW input and a wide output (65..96 -> 129..256 columns: 81 features -> 256 of
// example_model/model_multitask.py:51, with or without the aggregate-first ones column).
//
// dW [din x dout] = sum over the rows of x[r, :]^T (dy[r, :] (.) act'(a[r, :])) is 5 GFLOP over 118,000 rows but 280 MB of
// operands: HBM-bound (56 us at 5 TB/s).  The LDS-staged bf16-split kernel of gemm3.hip takes 99 us for it (it spends its time
// splitting and transposing (84 + 256) values per row).  Two kernels without LDS live here:
//   * wgradxb_kernel (shipped): with k = (row parity, row / 2) over 16 rows the lane layout of coalesced row loads IS the operand
//     layout of v_mfma_f32_32x32x16_bf16; the lane's eight values are split into bf16 pieces in registers.  71 us per call incl.
//     the second stage (0.49 of the HBM peak on algorithmic bytes; the loads alone: 57 us).
//   * wgradx_kernel (DEV_KNOBS builds, KGCN_WGRADX=f32): v_mfma_f32_32x32x2_f32 takes the operands as they lie in memory, no split
//     at all -- and 93-99 us: the f32 MFMA runs on the vector ALU's datapath and nothing overlaps it (loads alone 58 us, MFMAs
//     alone 38 us, together their SUM, with 2 or 8 waves per SIMD, ping-pong registers, LDS-direct operands, wide loads).
// What made the first kernel fast, in the order it was found (tools/wgradx_bench.py, all numbers kernel + second stage):
//   104 us  first version: per-element masks `(full || row < m) ? v : 0` on a uniform condition -- each became a branch;
//    90 us  no masks in the main path (columns beyond din / dout land in rows / columns of the accumulators that are never
//           stored; rows beyond m exist only in the tensors' last step, which is peeled off);
//    90 us  sched_barrier between the requests of step s + 1 and the arithmetic of step s (hipcc had sunk the loads BEHIND the
//           arithmetic: the ISA showed wait -> split -> MFMA -> 40 loads per step) -- necessary, not sufficient;
//    83 us  uniform row pointers + one per-lane offset per operand (no vector address arithmetic in front of the 40 loads), the
//           six products issued product-major over the three accumulators (a chain of six MFMAs on ONE accumulator waits for
//           each result);
//    88 / 86-90 us  two steps of lookahead for dy / those requests spread over the MFMAs: more requests in flight is not better
//           (profiles/r03_i_hbm_patterns.txt says the same of plain streams).
// The same design for the SQUARE wide layers (256 x 256, gemm3_wgrad_kernel's job: a wave owns one x block and four dy blocks, 24
// MFMAs per 16 rows, 40 loads of which each dy block is also loaded -- and split -- by three other waves) was built and measured:
// 150 us against 139 us of the LDS-staged kernel at 117,888 rows, 260 against 236 at 200,000: splitting every dy value four times costs more than the LDS round trip that shares the pieces.
// One partial per workgroup, reduce_partials adds them in a fixed order: deterministic.
#include "kgcn_common.h"

namespace kgcn {


#ifdef KGCN_DEV_KNOBS
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

#endif  // KGCN_DEV_KNOBS

// The same contraction on the bf16 matrix pipe WITHOUT LDS: with k = (row parity, row / 2) over 16 rows, the lane layout of
// coalesced row loads -- lane (li, hi) reads x[r + 2 j + hi][32 mb + li] and dy[r + 2 j + hi][32 w + li], j = 0..7 -- IS the
// operand layout of v_mfma_f32_32x32x16_bf16 (lane (i, hi) holds k = 8 hi + j): the eight values of a lane are split into their
// three bf16 pieces in registers (exact, kgcn_common.h) and feed six products per 32 x 32 block.  Against the f32 MFMA above:
// 18 MFMAs of 32 cycles on a pipe that runs BESIDE the vector ALU and the memory stream instead of 24 of 64 cycles that stop
// both; ~230 vector instructions per 16 rows and wave pay for it.
template <bool DACT, int MODE = 0>   // MODE (development): 1 = loads only, 2 = loads + split, no MFMA
__global__ __launch_bounds__(512, 2) void wgradxb_kernel(const float* __restrict__ x, long x_ld, const float* __restrict__ dy,
                                                         const float* __restrict__ yact, long dy_ld, long m, int din, int dout,
                                                         float c0, float c1, float c2, int relu, long steps_per_block,
                                                         float* __restrict__ part_dw, float* __restrict__ part_db) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, hi = lane >> 5;
  const int n = 32 * wave + li;                        // this lane's output column
  const bool nok = n < dout;
  const int nc = nok ? n : 0;
  int kc[3];
#pragma unroll
  for (int mb = 0; mb < 3; ++mb) kc[mb] = 32 * mb + li < din ? 32 * mb + li : 0;
  const long nsteps = (m + 15) / 16;
  const long s0 = (long)blockIdx.x * steps_per_block;
  long s1 = s0 + steps_per_block;
  if (s1 > nsteps) s1 = nsteps;
  f32x16 acc[3];
#pragma unroll
  for (int mb = 0; mb < 3; ++mb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[mb][i] = 0.f;
  float bsum = 0.f;
  // x (narrow, shared by the eight waves of the workgroup: L2 hits after the first of them) is requested one k-step ahead,
  // dy / the saved activations (the HBM streams) TWO: with one step of prefetch a wave has nothing in flight but the next step
  // while it splits and multiplies, and the time of a step is latency + arithmetic (measured: loads 62 us, + split 22, + MFMA 14)
  struct RawX { float a[3][8]; };
  struct RawY { float b[8], y[8]; };
  // Uniform row pointers (scalar registers) + ONE per-lane element offset per operand block: no vector address arithmetic in
  // front of the 40 loads of a step.  Full steps need no row clamp; the ragged last step of the tensors takes the clamped form.
  unsigned offx[3];
#pragma unroll
  for (int mb = 0; mb < 3; ++mb) offx[mb] = (unsigned)(hi * x_ld + kc[mb]);
  const unsigned offy = (unsigned)(hi * dy_ld + nc);
  auto load_x = [&](long s, RawX& r) __attribute__((always_inline)) {
    const float* xs = x + s * 16 * x_ld;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float* xr = xs + 2 * j * x_ld;
#pragma unroll
      for (int mb = 0; mb < 3; ++mb) r.a[mb][j] = (MODE == 3 && wave != 0) ? 1.f : xr[offx[mb]];
    }
  };
  auto load_y = [&](long s, RawY& r) __attribute__((always_inline)) {
    const float* gs = dy + s * 16 * dy_ld;
    const float* ys = DACT ? yact + s * 16 * dy_ld : nullptr;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      r.b[j] = gs[2 * j * dy_ld + offy];
      if constexpr (DACT) r.y[j] = ys[2 * j * dy_ld + offy];
    }
  };
  auto load_tail = [&](long s, RawX& rx, RawY& ry) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      long row = s * 16 + 2 * j + hi;
      row = row < m ? row : m - 1;
#pragma unroll
      for (int mb = 0; mb < 3; ++mb) rx.a[mb][j] = x[row * x_ld + kc[mb]];
      ry.b[j] = dy[row * dy_ld + nc];
      if constexpr (DACT) ry.y[j] = yact[row * dy_ld + nc];
    }
  };
  // Columns need no masks: a lane beyond din / dout reads a clamped (valid) column, and what it contributes lands in rows / columns
  // of the accumulators that are never stored.  Rows beyond m exist only in the last k-step of the tensors: `tail` (uniform) takes
  // the masked form once; the selects of a per-element mask on a uniform condition would each become a branch.
  auto mma = [&](long s, RawX& rx, RawY& ry, auto tailc) __attribute__((always_inline)) {
    constexpr bool TAIL = decltype(tailc)::value;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float g = ry.b[j];
      if constexpr (DACT) {
        const float a = ry.y[j];
        float d = __builtin_fmaf(__builtin_fmaf(c2, a, c1), a, c0);
        d = relu ? (a > 0.f ? 1.f : 0.f) : d;
        g *= d;
      }
      if constexpr (TAIL) g = (s * 16 + 2 * j + hi < m) ? g : 0.f;
      v[j] = g;
      bsum += g;
    }
    if constexpr (MODE == 1 || MODE == 3) {
#pragma unroll
      for (int mb = 0; mb < 3; ++mb)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[mb][j] += rx.a[mb][j] * v[j];
      return;
    }
    Frag3 fb, fa[3];
    split8(v, fb);
#pragma unroll
    for (int mb = 0; mb < 3; ++mb) {
      float u[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) u[j] = (!TAIL || s * 16 + 2 * j + hi < m) ? rx.a[mb][j] : 0.f;
      split8(u, fa[mb]);
      if constexpr (MODE == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          acc[mb][q] += __uint_as_float((fa[mb].p1[q] ^ fb.p3[q]) + (fa[mb].p2[q] ^ fb.p2[q]) + (fa[mb].p3[q] ^ fb.p1[q]));
      }
    }
    if constexpr (MODE == 0) {
      // product-major over the three accumulators: consecutive MFMAs never wait for each other's result
#define KGCN_WXB(PA, PB)                                  \
  acc[0] = mfma_bf16(fa[0].PA, fb.PB, acc[0]);           \
  acc[1] = mfma_bf16(fa[1].PA, fb.PB, acc[1]);           \
  acc[2] = mfma_bf16(fa[2].PA, fb.PB, acc[2]);
      KGCN_SPLIT_PRODUCTS(KGCN_WXB)
#undef KGCN_WXB
    }
  };
  // the last k-step of the tensors (if ragged) belongs to the last workgroup that has steps: peel it off the pipelined loop
  const bool ragged_last = (m % 16 != 0) && s1 == nsteps && s0 < s1;
  const long s1f = ragged_last ? s1 - 1 : s1;
  if (s0 < s1f) {
    // One step of lookahead, the requests of step s + 1 IN FRONT of the arithmetic of step s (sched_barrier: hipcc sinks them
    // behind it to shorten their live ranges -- a step then costs latency + arithmetic).  Measured on this shape, kernel + second
    // stage: 83 us; two steps of lookahead for dy (three register sets): 88; those requests spread over the MFMAs: 86-90; the
    // per-element masks of the first version (a branch each on a uniform condition): 104; the f32-MFMA kernel above: 96-99.
    // Requests past the end repeat the last step (valid, unused).
    RawX x0, x1;
    RawY y0, y1;
    const long sl = s1f - 1;
    load_y(s0, y0);
    load_x(s0, x0);
    long s = s0;
    for (; s + 1 < s1f; s += 2) {
      load_y(s + 1, y1);
      load_x(s + 1, x1);
      __builtin_amdgcn_sched_barrier(0);
      mma(s, x0, y0, std::false_type{});
      __builtin_amdgcn_sched_barrier(0);
      load_y(s + 2 < sl ? s + 2 : sl, y0);
      load_x(s + 2 < sl ? s + 2 : sl, x0);
      __builtin_amdgcn_sched_barrier(0);
      mma(s + 1, x1, y1, std::false_type{});
      __builtin_amdgcn_sched_barrier(0);
    }
    if (s < s1f) mma(s, x0, y0, std::false_type{});
  }
  if (ragged_last) {
    RawX xt;
    RawY yt;
    load_tail(s1 - 1, xt, yt);
    mma(s1 - 1, xt, yt, std::true_type{});
  }
  float* pw = part_dw + (long)blockIdx.x * din * dout;
#pragma unroll
  for (int mb = 0; mb < 3; ++mb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = 32 * mb + (i & 3) + 8 * (i >> 2) + 4 * hi;
      if (k < din && nok) pw[(long)k * dout + n] = acc[mb][i];
    }
  bsum += __shfl_xor(bsum, 32, 64);                    // the two row parities of a column
  if (hi == 0 && nok) part_db[(long)blockIdx.x * dout + n] = bsum;
}

bool wgradx_ok(int din, int dout, long x_ld, long dy_ld) { return din > 64 && din <= 96 && dout > 128 && dout <= 256; }

int launch_wgradx(const float* x, long x_ld, const float* dy, long dy_ld, long m, int din, int dout, float* part_dw, float* part_db,
                  int nparts, hipStream_t s, const float* yact, int act) {
  long rpb = (m + nparts - 1) / nparts;
  rpb = (rpb + 1) & ~1L;                               // row pairs never straddle two workgroups
  const int mb = (din + 31) / 32;
  if (mb != 3) return fail("wgradx: %d input columns", din);     // 65..96: three 32-row blocks of dW
#ifdef KGCN_DEV_KNOBS
  static const char* route = dev_knob("KGCN_WGRADX");          // development: "f32" = the f32-MFMA kernel
  const bool f32_route = route && route[0] == 'f';
#else
  const bool f32_route = false;
#endif
  if (!f32_route) {
    const long nsteps = (m + 15) / 16, spb = (nsteps + nparts - 1) / nparts;
    const bool dact = yact && act != KGCN_ACT_NONE;
    const float c0 = act == KGCN_ACT_TANH ? 1.f : 0.f, c1 = act == KGCN_ACT_SIGMOID ? 1.f : 0.f, c2 = -1.f;
#ifdef KGCN_DEV_KNOBS
    static const char* mode = dev_knob("KGCN_WGRADX_MODE");
    if (dact && mode && mode[0] == '1') {
      hipLaunchKernelGGL((wgradxb_kernel<true, 1>), dim3(nparts), dim3(512), 0, s, x, x_ld, dy, yact, dy_ld, m, din, dout, c0, c1, c2,
                         act == KGCN_ACT_RELU ? 1 : 0, spb, part_dw, part_db);
      return check_launch("wgradxb_kernel");
    }
    if (dact && mode && mode[0] == '3') {
      hipLaunchKernelGGL((wgradxb_kernel<true, 3>), dim3(nparts), dim3(512), 0, s, x, x_ld, dy, yact, dy_ld, m, din, dout, c0, c1, c2,
                         act == KGCN_ACT_RELU ? 1 : 0, spb, part_dw, part_db);
      return check_launch("wgradxb_kernel");
    }
    if (dact && mode && mode[0] == '2') {
      hipLaunchKernelGGL((wgradxb_kernel<true, 2>), dim3(nparts), dim3(512), 0, s, x, x_ld, dy, yact, dy_ld, m, din, dout, c0, c1, c2,
                         act == KGCN_ACT_RELU ? 1 : 0, spb, part_dw, part_db);
      return check_launch("wgradxb_kernel");
    }
#endif
    if (dact)
      hipLaunchKernelGGL((wgradxb_kernel<true>), dim3(nparts), dim3(512), 0, s, x, x_ld, dy, yact, dy_ld, m, din, dout, c0, c1, c2,
                         act == KGCN_ACT_RELU ? 1 : 0, spb, part_dw, part_db);
    else
      hipLaunchKernelGGL((wgradxb_kernel<false>), dim3(nparts), dim3(512), 0, s, x, x_ld, dy, yact, dy_ld, m, din, dout, c0, c1, c2,
                         0, spb, part_dw, part_db);
    return check_launch("wgradxb_kernel");
  }
#ifdef KGCN_DEV_KNOBS
  if (yact && act != KGCN_ACT_NONE)
    hipLaunchKernelGGL((wgradx_kernel<3, true>), dim3(nparts), dim3(512), 0, s, x, x_ld, dy, yact, dy_ld, m, din, dout, act, rpb,
                       part_dw, part_db);
  else
    hipLaunchKernelGGL((wgradx_kernel<3, false>), dim3(nparts), dim3(512), 0, s, x, x_ld, dy, yact, dy_ld, m, din, dout, act, rpb,
                       part_dw, part_db);
  return check_launch("wgradx_kernel");
#else
  (void)rpb;
  return fail("wgradx: no route");
#endif
}

}  // namespace kgcn
