// Tall-skinny fp32 GEMMs of the wide layers (BASELINE configs 4 and 5: 256 -> 256, kgcn/layers.py:99-100, :248-260 Keras Dense,
// example_model/model_multitask.py:51-57, example_model/model_gin.py:45-54) on the f16 matrix pipe with a TWO-piece split and
// exact power-of-two scaling -- three MFMA products per fp32 product instead of the six of the bf16 split (gemm3.hip), which
// turns these GEMMs from matrix-pipe-bound (0.36-0.46 of the pipe at 81-110 us, profiles/r03_n_cfg4_rocprof.txt) into
// HBM-bound streams:
//
//   x' = x * 2^kr   (kr per ROW of x: the row maximum lands in [2^14, 2^15))        W' = W * 2^kc (kc per COLUMN of W)
//   x' = hx + lx,  hx = f16(x') (round to nearest: 11 bits),  lx = f16(x' - hx) (x' - hx is exact in fp32 and has <= 13
//   significant bits: lx loses at most its last one, |x' - hx - lx| <= 2^-24 |x'|)
//   x'w' = hx hw + (hx lw + lx hw) + [lx lw + the two representation errors] ;  |[...]| <= 3 * 2^-24 |x'w'|
//   y = 2^-(kr + kc) * sum_k (...) + bias
//
// f16 x f16 products (22 bits) are exact in the fp32 accumulator, power-of-two scaling is exact, so the result differs from
// the exact product sum by ~2^-23 sum_k |x w| -- the class of ONE fp32 rounding per product, the same as the six-product
// bf16 split (measured against fp64 next to numpy float32 and the bf16 split: tests/test_gpu_dense_edges.py,
// profiles/r04_accuracy.json).  What the narrow f16 exponent costs: an element more than 2^14 below its row's (column's)
// maximum keeps fewer than 22 bits -- its ABSOLUTE error stays below 2^-25 * 2^-15 of that maximum (f16 denormal spacing),
// i.e. the error bound gains a term K * 2^-40 * max_k|x[r,k]| * max_k|W[k,n]| next to 2^-23 sum_k |x w|.
// Non-finite inputs: a row of x (column of W) that holds +-inf / NaN gives non-finite outputs in that whole row (column) --
// every one of them depends on the non-finite input --, all other outputs are untouched.
//
// gemmh_fwd_kernel (y = act(x W + b), dx = dy W^T, and the backward form with g (.) act'(a) [+ d pooled] as the operand):
//   WEIGHT-STATIONARY.  One workgroup = 8 waves, persistent over 64-row tiles; wave w owns the 32 output columns 32 w .. +31 and
//   keeps its slice of W' -- the h and l fragments of all 16 k-steps, 128 registers, read once from the fragment table of
//   wtable.hip -- for the whole launch.  (The first two forms re-read W' fragments from L2 for every tile: 86-90 us with the TCP
//   miss path moving 591 MB per launch; vmcnt counts in order, so a wave whose x tile is in flight cannot wait for W' fragments
//   requested later.  profiles/r04_gemmh_history.txt)
//   * a tile's rows travel HBM -> registers as 1 KiB rows (lane = 4 consecutive columns, one dwordx4 buffer load per row and lane:
//     no address arithmetic, rows past the end read as 0), eight rows per wave, requested before the previous tile is multiplied;
//   * a row belongs to one wave, so its maximum is a DPP wave reduction and its scale a SCALAR; split = v_ldexp,
//     v_cvt_pk_f16_f32, 2 x v_cvt_f32_f16, v_pk_add_f32, v_cvt_pk_f16_f32 per pair of values (3.5 per value; bf16 x 3: 5.5);
//   * pieces land in LDS ONCE per tile in MFMA A-operand order (1 KiB block per (m-tile, k-step, piece), slots XOR-rotated so that
//     the 8-byte writes of a row and the 16-byte fragment reads are both conflict-free) next to the rows' exponents; every wave
//     then multiplies the whole tile with its columns: 2 m-tiles x 16 k-steps x 3 products (smallest terms first);
//   * epilogue: v_ldexp by -(kr + kc), + bias, activation, buffer stores.  The LDS hand-over is s_waitcnt lgkmcnt(0) + s_barrier
//     (gh_barrier_lds): __syncthreads() would also drain vmcnt -- the next tile's rows and this tile's stores.
// gemmh_wgrad_kernel<DACT> (dW = x^T dy, db = colsum dy): no LDS.  The batch rows are the contraction index: with lane (li, hi)
//   reading x[r + 8 hi + j][c + li], j = 0..7, a coalesced dword load IS the MFMA operand layout (wgradx.hip), so a wave
//   splits its own operands in registers and runs without barriers.  Scales are per COLUMN here and not known in advance:
//   every lane keeps the scale of its column(s) and a limit; when a value exceeds the limit (first step, rarely later) the
//   wave rescales its accumulators (exact) and goes on -- the online form of the row scale above.  One partial dW per
//   workgroup (written unscaled), fixed-order second stage as everywhere.  Shipped for the fused act' form; the plain weight
//   gradient takes gemmh_wgradl_kernel below (every value split once, shared through LDS).
#include "gemmh.h"

namespace kgcn {

#ifdef KGCN_PROBE   // development: per-wave cycle sums per phase of gemmh_wgradl_kernel (tools/gemmh_probe.py)
__device__ long long* gh_probe = nullptr;
#define GHP_DECL long long pt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long pc_ = __builtin_readcyclecounter();
#define GHP(k) { const long long n_ = __builtin_readcyclecounter(); pt_[k] += n_ - pc_; pc_ = n_; }
#define GHP_FLUSH if (gh_probe && lane == 0) { for (int k_ = 0; k_ < 8; ++k_) gh_probe[(((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 8 + k_] = pt_[k_]; }
#else
#define GHP_DECL
#define GHP(k)
#define GHP_FLUSH
#endif
#ifndef GH_INTERLEAVE
#define GH_INTERLEAVE 1          // 0: the next tile staged AFTER the multiplication (rounds 4 form; build/variants A/B)
#endif
#ifndef GH_TWO_PHASES
#define GH_TWO_PHASES 1          // 1: waves 0-3 and 4-7 (one of each per SIMD) run half a tile apart (see the forward kernel's main loop)
#endif
#ifndef GH_VARIANT
#define GH_VARIANT 0             // development (tools/gemmh_variants.sh): 2 no y stores, 3 no x loads (forward); 5 no MFMAs, 6 no
#endif                           // split, 7 no loads (gemmh_wgradl) -- what each part of the kernels costs
constexpr int GH_BM = 64;        // rows per tile
constexpr int GH_KMAX = 256;     // widest x row one lane quad layout covers (64 lanes x 4 columns)

// LDS slot of row rr32 (0..31 of its m-tile), lane half hi, in the block of k-step ks: XOR-rotated by (2 ks + hi) mod 16
__device__ __forceinline__ int gh_slot(int rr32, int ks, int hi) { return (rr32 ^ ((2 * ks + hi) & 15)) + 32 * hi; }

// DK: 0 plain; 1 operand = g (.) (c0 + c1 a + c2 a^2) (sigmoid / tanh derivative in the layer OUTPUT a); 2 relu (a > 0).
// With DK the d pre-activation is stored `pdiff` elements away from the gradient row it was formed from (column block 0 only).
// NKS: k-steps held in registers -- 16 (K <= 256: 128 registers of W') or 8 (K <= 128).
//
// WEIGHT-STATIONARY: wave w of the 8 owns the output columns 32 w .. +31 and keeps ITS slice of W' -- NKS k-steps x (high,
// low) x 16 bytes per lane -- in registers for the whole launch; x tiles stream through LDS.  Two earlier forms of this kernel
// fetched W' fragments from the L2-resident table inside the k loop (64-row tiles: 4 KB of W' per KB of x): 591 MB through
// the vector L1s per launch, TCP_PENDING_STALL 43 % of the time, 75-86 us at 117,888 rows (profiles/r04_gemmh_history.txt) --
// and a wave with a tile of x on its way from HBM cannot wait for such a fragment without waiting for the whole tile first
// (vmcnt counts in order).  With the weights in registers the only vector-memory traffic of the loop is x in, y out.
// Row-block buffer descriptors: every tile access is buffer_{load,store} v, voffset, s[rsrc], soffset -- the descriptor (scalar
// registers) points at the wave's first row of the tile and is `rows` rows long, the row within it is the scalar soffset, the
// lane's column its ONE 32-bit voffset.  No vector address arithmetic, one address register per tensor instead of a 64-bit
// pointer per access (32 y pointers next to 128 registers of W' were the spill source of the first build), and rows beyond m
// need no clamps or masks: loads outside the descriptor return 0, stores outside it are dropped.
constexpr int GH_PIECES = 2 * 16 * 2 * 64;     // u32x4 entries of one LDS tile buffer: (m-tile, k-step, piece, slot)
constexpr size_t GH_LDS = 2 * (size_t)GH_PIECES * 16 + 4 * 64 * 4;     // two piece buffers + (up to three) sets of row exponents
template <int DK, int NKS>
__global__ __launch_bounds__(512, 2) void gemmh_fwd_kernel(const float* __restrict__ x, long m, int din, long x_ld,
                                                           const u32x4* __restrict__ tab, const float* __restrict__ bias,
                                                           float* __restrict__ y, int dout, long y_ld, int act, GhDact da) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: row indices stay on the SALU
  int* rowk_base = reinterpret_cast<int*>(dsm + 2 * (size_t)GH_PIECES * 16);     // [2 or 3][64] row exponents
  GHP_DECL                                                 // (development: tools/gemmh_fwd_probe.py)
  const long ntiles = (m + GH_BM - 1) / GH_BM;
  const long G = gridDim.x;
  long t = blockIdx.x;
  if (t >= ntiles) return;                              // uniform for the whole workgroup
  const long ntw = (ntiles - t + G - 1) / G;            // tiles of this workgroup

  // ---- this wave's slice of W' and its column constants -------------------------------------------------------------
  const int nt32 = gh_nt32(dout), kse = gh_kse(din);
  const int* kctab = reinterpret_cast<const int*>(tab + (long)kse * nt32 * 2 * 64);
  const int n0 = blockIdx.y * 256 + 32 * wave;          // first output column
  const int ntc = (n0 / 32) < nt32 ? (n0 / 32) : nt32 - 1;      // clamped: the columns of such a wave are never stored
  // (compile-time indices everywhere: an array indexed through a lambda argument or initialised under a branch stays in
  // scratch memory -- the first build of this kernel kept W' there)
  u32x4 Bh[NKS], Bl[NKS];
  static_for<NKS>([&](auto kc) __attribute__((always_inline)) {
    constexpr int ks = decltype(kc)::value;
    const int kk = ks < kse ? ks : 0;                   // k-steps beyond the table: zero operands (selected below)
    const u32x4* e = tab + ((long)(kk * nt32 + ntc) * 2) * 64 + lane;
    const u32x4 h = e[0], l = e[64];
    const u32x4 z = {0u, 0u, 0u, 0u};
    Bh[ks] = ks < kse ? h : z;
    Bl[ks] = ks < kse ? l : z;
  });
  const int col = n0 + li;
  const int ldy4 = (int)(y_ld * 4);
  const unsigned voff_y = 4u * (unsigned)(4 * hi * y_ld + (col < dout ? col : 0));     // rows 4 hi + ... of the tile, this column
  const float bcol = (bias && col < dout) ? bias[col] : 0.f;
  const int kcol = kctab[32 * ntc + li];

  // ---- staging: rows 8 w .. 8 w + 7 of a tile; lane = 4 consecutive columns -----------------------------------------
  const int c4 = 4 * lane;
  const bool cok = c4 < din;
  const unsigned voff_x = 16u * (unsigned)(cok ? lane : 0);       // byte offset of the lane's 4 columns in a row
  const int ldx4 = (int)(x_ld * 4);
  const int wks = lane >> 2, whi = (lane >> 1) & 1, wsub = lane & 1;    // LDS coordinates of this lane's 8 bytes
  f32x4 raw[8];
  auto load_tile = [&](long tt) __attribute__((always_inline)) {
    if (DK != 0 && da.bc_only) {                        // uniform: the gradient is the read-out's broadcast alone
#pragma unroll
      for (int i = 0; i < 8; ++i) raw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      return;
    }
    const __amdgpu_buffer_rsrc_t rx = gh_rows(x, tt * GH_BM + 8 * wave, 8, m, x_ld);    // tiles past the end: empty, zeros
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (GH_VARIANT == 3) raw[i] = f32x4{1.f + (float)i, 2.f, 3.f, (float)lane};
      else raw[i] = gh_ld4(rx, voff_x, i * ldx4);
    }
  };
  auto stage = [&](long tt, int buf) __attribute__((always_inline)) {
    const long r0 = tt * GH_BM + 8 * wave;
    if constexpr (DK != 0) {
      const __amdgpu_buffer_rsrc_t ra = gh_rows(x + da.ydiff, r0, 8, m, x_ld);
      const __amdgpu_buffer_rsrc_t rp = gh_rows(x + da.pdiff, r0, (blockIdx.y == 0 && cok) ? 8 : 0, m, x_ld);
      f32x4 ya[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) ya[i] = gh_ld4(ra, voff_x, i * ldx4);
      // (graph, node) of the wave's consecutive rows, kept incrementally: one division per tile
      long gq = 0;
      int grem = 0;
      const long gmax = da.bc ? (m - 1) / da.bc_n : 0;   // rows beyond m are zero anyway; their address stays valid
      if (da.bc) { gq = r0 / da.bc_n; grem = (int)(r0 - gq * da.bc_n); }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        f32x4 g = raw[i];
        if (da.bc) {                                    // uniform
          g += *reinterpret_cast<const f32x4*>(da.bc + (gq < gmax ? gq : gmax) * da.bc_ld + (cok ? c4 : 0));
          if (++grem == da.bc_n) { grem = 0; ++gq; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = ya[i][e];
          if constexpr (DK == 1) g[e] *= __builtin_fmaf(__builtin_fmaf(da.c2, a, da.c1), a, da.c0);
          else g[e] = a > 0.f ? g[e] : 0.f;
        }
        raw[i] = g;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, g), rp, (int)voff_x, i * ldx4, 0);   // rows >= m: dropped
      }
    }
    unsigned mx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (!(cok && (DK == 0 || r0 + i < m))) raw[i] = f32x4{0.f, 0.f, 0.f, 0.f};    // (plain form: rows >= m were loaded as 0)
      float a;
      asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(a) : "v"(raw[i][0]), "v"(raw[i][1]), "v"(raw[i][2]));
      asm("v_max_f32 %0, %1, |%2|" : "=v"(a) : "v"(a), "v"(raw[i][3]));
      mx[i] = __float_as_uint(a);
    }
    wave_umax4(mx[0], mx[1], mx[2], mx[3]);
    wave_umax4(mx[4], mx[5], mx[6], mx[7]);
    unsigned char* pbuf = dsm + (size_t)buf * GH_PIECES * 16;
    int* rowk = rowk_base + 64 * buf;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rr = 8 * wave + i;                       // row of the tile
      const int k = scale_exp(__uint_as_float(mx[i]));   // wave-uniform
      unsigned h0, l0, h1, l1;
      splith_pair(__builtin_ldexpf(raw[i][0], k), __builtin_ldexpf(raw[i][1], k), h0, l0);
      splith_pair(__builtin_ldexpf(raw[i][2], k), __builtin_ldexpf(raw[i][3], k), h1, l1);
      // every lane writes (lanes beyond din write the zeros of the padded k-steps); all 16 k-step blocks exist
      unsigned char* e = pbuf + ((size_t)(((rr >> 5) * 16 + wks) * 2) * 64 + gh_slot(rr & 31, wks, whi)) * 16 + 8 * wsub;
      *reinterpret_cast<u32x2*>(e) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(e + 1024) = u32x2{l0, l1};
      rowk[rr] = k;                                      // same value from every lane
    }
  };
  // ONE row of the next tile (plain form): what `stage` does for the wave's eight rows at once, as a piece the multiplication
  // lays between its MFMAs (round 5: staged after the multiplication the 8 rows were a vector-ALU phase with the matrix pipe idle,
  // and the multiplication a matrix phase with the vector ALU idle -- 0.47 of the HBM floor at 65 / 107 us; gemmb.hip's history)
  auto stage_row = [&](int buf, int kslot, auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    if (!cok) raw[i] = f32x4{0.f, 0.f, 0.f, 0.f};              // (rows >= m were loaded as 0)
    float a;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(a) : "v"(raw[i][0]), "v"(raw[i][1]), "v"(raw[i][2]));
    asm("v_max_f32 %0, %1, |%2|" : "=v"(a) : "v"(a), "v"(raw[i][3]));
    const int k = scale_exp(__uint_as_float(wave_umax1(__float_as_uint(a))));           // wave-uniform
    const int rr = 8 * wave + i;
    unsigned h0, l0, h1, l1;
    splith_pair(__builtin_ldexpf(raw[i][0], k), __builtin_ldexpf(raw[i][1], k), h0, l0);
    splith_pair(__builtin_ldexpf(raw[i][2], k), __builtin_ldexpf(raw[i][3], k), h1, l1);
    unsigned char* e = dsm + (size_t)buf * GH_PIECES * 16 +
                       ((size_t)(((rr >> 5) * 16 + wks) * 2) * 64 + gh_slot(rr & 31, wks, whi)) * 16 + 8 * wsub;
    *reinterpret_cast<u32x2*>(e) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(e + 1024) = u32x2{l0, l1};
    (rowk_base + 64 * kslot)[rr] = k;                          // same value from every lane
  };
  float dotacc = 0.f;
  // kslot / kslot_next: where the row exponents of this tile lie / those of the tile staged meanwhile go (== buf / buf ^ 1 but for the
  // two-phase schedule, whose group B reads a tile's exponents while group A already stages the tile after the next one: three sets)
  auto compute = [&](long tt, int buf, long tt_load, int kslot, int kslot_next) __attribute__((always_inline)) {
    const u32x4* lds = reinterpret_cast<const u32x4*>(dsm) + (size_t)buf * GH_PIECES;
    const int* rowk = rowk_base + 64 * kslot;
    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    // A fragments one k-step ahead, scheduling scope bounded per k-step (left alone, the scheduler hoists all 64 fragment
    // reads of the straight-line loop to its head: 254 spilled registers)
    auto load_a = [&](u32x4 (&A)[4], int ks) __attribute__((always_inline)) {
      const int slot = gh_slot(li, ks, hi);
      const u32x4* e0 = lds + (size_t)((0 * 16 + ks) * 2) * 64 + slot;
      const u32x4* e1 = lds + (size_t)((1 * 16 + ks) * 2) * 64 + slot;
      A[0] = e0[0]; A[1] = e0[64]; A[2] = e1[0]; A[3] = e1[64];
    };
    u32x4 A0[4], A1[4];
    load_a(A0, 0);
    static_for<NKS / 2>([&](auto kc) __attribute__((always_inline)) {
      constexpr int ks = 2 * decltype(kc)::value;
      __builtin_amdgcn_sched_barrier(0);
      load_a(A1, ks + 1);
      __builtin_amdgcn_sched_barrier(0);
      // smallest terms first
      acc[0] = mfma_f16(A0[1], Bh[ks], acc[0]);
      acc[1] = mfma_f16(A0[3], Bh[ks], acc[1]);
      acc[0] = mfma_f16(A0[0], Bl[ks], acc[0]);
      acc[1] = mfma_f16(A0[2], Bl[ks], acc[1]);
      acc[0] = mfma_f16(A0[0], Bh[ks], acc[0]);
      acc[1] = mfma_f16(A0[2], Bh[ks], acc[1]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ks + 2 < NKS) load_a(A0, ks + 2);
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = mfma_f16(A1[1], Bh[ks + 1], acc[0]);
      acc[1] = mfma_f16(A1[3], Bh[ks + 1], acc[1]);
      acc[0] = mfma_f16(A1[0], Bl[ks + 1], acc[0]);
      acc[1] = mfma_f16(A1[2], Bl[ks + 1], acc[1]);
      acc[0] = mfma_f16(A1[0], Bh[ks + 1], acc[0]);
      acc[1] = mfma_f16(A1[2], Bh[ks + 1], acc[1]);
      if constexpr (DK == 0 && GH_INTERLEAVE != 0 && NKS == 16) {
        // between these six MFMAs: row ks / 2 of the NEXT tile (already in the registers) -> the other buffer
        stage_row(buf ^ 1, kslot_next, std::integral_constant<int, ks / 2>{});
        // ... and its registers take the same row of the tile after that: a whole multiplication ahead of its use
        raw[ks / 2] = gh_ld4(gh_rows(x, tt_load * GH_BM + 8 * wave, 8, m, x_ld), voff_x, (ks / 2) * ldx4);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
      }
    });
    GHP(1)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DK == 0 && GH_INTERLEAVE != 0 && NKS == 16 && GH_TWO_PHASES != 0) {
      gh_barrier_lds();                                  // (the other group's multiplication starts / has ended here: see the main loop)
      GHP(4)
    }
    // ---- y <- act(2^-(kr + kc) acc + bias) ---------------------------------------------------------------------------
    const long row0 = tt * GH_BM;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const u32x4 kr4 = *reinterpret_cast<const u32x4*>(rowk + 32 * mt + 8 * rq + 4 * hi);
#pragma unroll
        for (int rj = 0; rj < 4; ++rj)
          acc[mt][4 * rq + rj] = __builtin_ldexpf(acc[mt][4 * rq + rj], -((int)kr4[rj] + kcol)) + bcol;
      }
    // the activation as a compile-time constant inside each arm: hipcc does not unswitch the element loop on a runtime code
    auto apply = [&](auto code) __attribute__((always_inline)) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = act_fwd(acc[mt][r], decltype(code)::value);
    };
    if (act == KGCN_ACT_SIGMOID) apply(std::integral_constant<int, KGCN_ACT_SIGMOID>{});
    else if (act == KGCN_ACT_RELU) apply(std::integral_constant<int, KGCN_ACT_RELU>{});
    else if (act == KGCN_ACT_TANH) apply(std::integral_constant<int, KGCN_ACT_TANH>{});
    if (GH_VARIANT == 2 && acc[0][0] != 1.2345e-30f) return;
    GHP(2)
    if (DK != 0 && da.dot_part) {                        // uniform: <product, y operand> instead of the stores
      if (col < dout) {
        const __amdgpu_buffer_rsrc_t rz = gh_rows(y, row0, GH_BM, m, y_ld);   // rows >= m: read as 0
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            float z[4];
#pragma unroll
            for (int rj = 0; rj < 4; ++rj)
              z[rj] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rz, (int)voff_y, (32 * mt + rj + 8 * rq) * ldy4, 0));
#pragma unroll
            for (int rj = 0; rj < 4; ++rj) dotacc = __builtin_fmaf(acc[mt][4 * rq + rj], z[rj], dotacc);
          }
      }
    } else if (col < dout) {
      const __amdgpu_buffer_rsrc_t ry = gh_rows(y, row0, GH_BM, m, y_ld);     // rows >= m: dropped by the descriptor
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[mt][r];        // (a copy: __builtin_bit_cast of the vector-element lvalue read element 0 sixteen times)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, (int)voff_y, (32 * mt + (r & 3) + 8 * (r >> 2)) * ldy4, 0);
        }
    }
  };

  load_tile(t);
  stage(t, 0);
  if constexpr (DK == 0 && GH_INTERLEAVE != 0 && NKS == 16) load_tile(t + G);      // the rows the first multiplication stages
  gh_barrier_lds();
  // No exit between a request and its use (hipcc proves a requested tile dead on an exit path and sinks its loads behind
  // everything in between -- seen in the ISA of an earlier form): the tile after the last one clamps to the last row (cached)
  // and is staged into the buffer nobody reads.
  // TWO PHASES (plain form): a tile's 16 k-steps keep the matrix pipe busy only while BOTH waves of a SIMD are inside them -- 6,421 of
  // the 13,095 cycles of a tile with the eight waves in lock-step; head, scaling, activation, stores and the barrier (51 %) left it idle
  // (tools/gemmh_fwd_probe.py).  Waves 0-3 (group A) and 4-7 (group B) now alternate: every wave passes TWO barriers per tile, one
  // behind its k-steps and one in front of them, and the instances pair up as  X: A has multiplied tile t, B starts to;  Y: B has, A
  // starts on tile t + 1 -- one group's epilogue and loop head run under the other group's MFMAs.  Who stages what and which buffer is
  // free when is unchanged: a group's rows of tile t + 1 are staged during ITS k-steps of tile t, both before either group reads them.
  constexpr bool kTwoPhases = DK == 0 && GH_INTERLEAVE != 0 && NKS == 16 && GH_TWO_PHASES != 0;
#ifndef GH_GROUP_SHIFT
#define GH_GROUP_SHIFT 2
#endif
  const bool group_b = ((wave >> GH_GROUP_SHIFT) & 1) != 0;    // uniform; waves w and w + 4 share a SIMD
  int k3 = 0;                                              // the tile's set of row exponents (two phases: one of three)
  for (long i = 0; i < ntw; ++i) {
    const int buf = (int)(i & 1);
    if constexpr (kTwoPhases) {
      if (group_b) gh_barrier_lds();                     // X: group A's k-steps of this tile are done
    }
    if constexpr (DK == 0 && !(GH_INTERLEAVE != 0 && NKS == 16)) {
      __builtin_amdgcn_sched_barrier(0);
      load_tile(t + G);                                  // on its way from HBM while this tile is multiplied
      __builtin_amdgcn_sched_barrier(0);
    }
    GHP(0)
    const int kslot = kTwoPhases ? k3 : buf;
    if constexpr (kTwoPhases) k3 = k3 == 2 ? 0 : k3 + 1;
    compute(t, buf, t + 2 * G, kslot, kTwoPhases ? k3 : (buf ^ 1));
    GHP(3)
    if constexpr (DK != 0) load_tile(t + G);             // backward form: gradient and saved output travel together below
    if constexpr (!(DK == 0 && GH_INTERLEAVE != 0 && NKS == 16)) stage(t + G, buf ^ 1);
    if constexpr (kTwoPhases) {
      if (!group_b) gh_barrier_lds();                    // Y: group B's k-steps of this tile are done
    } else {
      gh_barrier_lds();
    }
    GHP(4)
#ifndef KGCN_ABL_HOT                          // (development: with it every workgroup re-reads its first tile -- the kernel without HBM traffic)
    t += G;
#endif
  }
  GHP_FLUSH
  if constexpr (DK != 0) {
    if (da.dot_part) {                                   // uniform.  Fixed order: lanes (butterfly), then the eight waves
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) dotacc += __shfl_xor(dotacc, o, 64);
      float* red = reinterpret_cast<float*>(dsm);        // (every wave is behind the loop's last barrier: the tile buffers are free)
      if (lane == 0) red[wave] = dotacc;
      __syncthreads();
      if (tid == 0) {
        float s8 = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) s8 += red[w8];
        da.dot_part[(long)blockIdx.y * gridDim.x + blockIdx.x] = s8;
      }
    }
  }
}

// workgroups of a launch over m rows and `dout` output columns (= partials of the dot form)
int gemmh_dot_parts(long m, int dout) {
  const long ntiles = (m + GH_BM - 1) / GH_BM;
  return (int)((ntiles < kNumCU ? ntiles : kNumCU) * ((dout + 255) / 256));
}

// x: 16-byte aligned rows of <= 256 columns (din % 4 == 0, x_ld % 4 == 0); the table holds the f16 section for (din, dout)
bool gemmh_fwd_ok(const float* x, long m, int din, long x_ld, int dout) {
  // one 64-row tile per workgroup and fewer workgroups than CUs is the regime of gemm3's 64 x 64 column cut (sparse.py's 4,457
  // rows: 0.320 ms per step with it, 0.332 ms through this kernel)
  return din % 4 == 0 && din <= GH_KMAX && din >= 32 && x_ld % 4 == 0 && aligned16(x) && dout > 128 && m >= (long)kNumCU * GH_BM;
}

template <int DK>
static int gh_fwd_launch(const float* x, long m, int din, long x_ld, const void* tabh, const float* bias, float* y, int dout,
                         long y_ld, int act, const GhDact& da, hipStream_t s) {
  static thread_local bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemmh_fwd_kernel<DK, 16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              kLdsBytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemmh_fwd_kernel<DK, 8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              kLdsBytes);
    attr_set = true;
  }
  const long ntiles = (m + GH_BM - 1) / GH_BM, cap = kNumCU;
  const dim3 grid((unsigned)(ntiles < cap ? ntiles : cap), (unsigned)((dout + 255) / 256));
  if (din > 128)
    hipLaunchKernelGGL((gemmh_fwd_kernel<DK, 16>), grid, dim3(512), GH_LDS, s, x, m, din, x_ld, static_cast<const u32x4*>(tabh),
                       bias, y, dout, y_ld, act, da);
  else
    hipLaunchKernelGGL((gemmh_fwd_kernel<DK, 8>), grid, dim3(512), GH_LDS, s, x, m, din, x_ld, static_cast<const u32x4*>(tabh),
                       bias, y, dout, y_ld, act, da);
  return check_launch("gemmh_fwd_kernel");
}

int launch_gemmh_fwd(const float* x, long m, int din, long x_ld, const void* tabh, const float* bias, float* y, int dout,
                     long y_ld, int act, hipStream_t s) {
  return gh_fwd_launch<0>(x, m, din, x_ld, tabh, bias, y, dout, y_ld, act, GhDact{}, s);
}

// dx = ((grad [+ d pooled of the row's graph]) (.) act'(act_out)) @ W^T, d pre-activation written on the way.  k = the layer's
// output width (the contraction), n = its input width; `tabh` = the f16 table of W^T.  Returns -1 when the operands do not
// fit the kernel (the caller falls back to gemm3 / the unfused route).
int launch_gemmh_dx_dact(const float* grad, const float* act_out, float* dpre, long m, int k, long ld, const void* tabh,
                         float* dx, int n, long dx_ld, int dact, hipStream_t s, const float* pooled_grad, int n_nodes, long pooled_ld,
                         float* dot_part) {
  // dot_part != nullptr: `dx` is READ ([m, n], row stride dx_ld) and <product, dx> goes to dot_part[workgroups of the launch]
  // (gemmh_dot_parts(m, n) floats) instead of the product being stored
  const float* base = grad ? grad : act_out;
  if (!(gemmh_fwd_ok(base, m, k, ld, n) && (!grad || aligned16(grad)) && aligned16(act_out) && aligned16(dpre) && tabh &&
        dact != KGCN_ACT_NONE && dpre != grad && (grad || pooled_grad) &&
        (!pooled_grad || (aligned16(pooled_grad) && n_nodes > 0 && pooled_ld % 4 == 0))))
    return -1;
  GhDact da;
  da.ydiff = act_out - base;
  da.pdiff = dpre - base;
  da.bc = pooled_grad;
  da.bc_ld = pooled_ld;
  da.bc_n = n_nodes > 0 ? n_nodes : 1;
  da.bc_only = grad ? 0 : 1;
  da.c0 = dact == KGCN_ACT_TANH ? 1.f : 0.f;
  da.c1 = dact == KGCN_ACT_SIGMOID ? 1.f : 0.f;
  da.c2 = -1.f;
  da.dot_part = dot_part;
  if (dact == KGCN_ACT_RELU) return gh_fwd_launch<2>(base, m, k, ld, tabh, nullptr, dx, n, dx_ld, KGCN_ACT_NONE, da, s);
  return gh_fwd_launch<1>(base, m, k, ld, tabh, nullptr, dx, n, dx_ld, KGCN_ACT_NONE, da, s);
}


// ------------------------------------------------------------------------------------------------------------------
// Weight gradient dW[din, dout] = x^T dy, dbias = colsum(dy) for the wide layers: the batch rows are the contraction index.
// Workgroup (8 waves) = a [128 x 256] block of dW over a range of 16-row k-steps; wave (wr, wc) owns the [64 x 64] block of
// x columns 64 wr .. and dy columns 64 wc .. (2 x 2 tiles): per k-step 32 coalesced dword loads (lane (li, hi), j: row
// 16 s + 8 hi + j, column c + li -- the operand layout), split in registers, 12 MFMAs.  No LDS, no barrier.
// Per-column scales, online: lane li keeps for each of its four columns the exponent k (values enter as v * 2^k) and the
// limit 2^(15 - k); a step with a value above its limit takes the slow path: new exponents from the running column maxima
// (two bits of headroom), accumulators rescaled by the exact powers of two (x columns are accumulator ROWS: their deltas are
// fetched across lanes).  Partials are written unscaled.
// ------------------------------------------------------------------------------------------------------------------
template <bool DACT>
__global__ __launch_bounds__(512, 2) void gemmh_wgrad_kernel(const float* __restrict__ x, long x_ld, const float* __restrict__ dy,
                                                             long dy_ld, long m, int din, int dout, long steps_per_block,
                                                             float* __restrict__ part_dw, float* __restrict__ part_db,
                                                             const float* __restrict__ yact, float c0, float c1, float c2,
                                                             int relu) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, li = lane & 31, hi = lane >> 5;
  const int wr = wave >> 2, wc = wave & 3;
  const int i0 = blockIdx.y * 128 + 64 * wr, j0 = blockIdx.z * 256 + 64 * wc;
  const long nsteps = (m + 15) / 16;
  const long s0 = (long)blockIdx.x * steps_per_block;
  long s1 = s0 + steps_per_block;
  if (s1 > nsteps) s1 = nsteps;

  // columns beyond din / dout read a clamped (valid) column: what they contribute lands in accumulator rows / columns that
  // are never stored
  unsigned offa[2], offb[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int ca = i0 + 32 * t + li, cb = j0 + 32 * t + li;
    offa[t] = 4u * (unsigned)(8 * hi * x_ld + (ca < din ? ca : din - 1));      // BYTE offsets: scalar base + 32-bit lane offset
    offb[t] = 4u * (unsigned)(8 * hi * dy_ld + (cb < dout ? cb : dout - 1));    // is an addressing mode of global_load
  }
  struct Raw { float a[2][8], b[2][8], y[DACT ? 2 : 1][DACT ? 8 : 1]; };
  auto at = [](const float* base, unsigned byte_off) __attribute__((always_inline)) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
  };
  auto load = [&](long s, Raw& r) __attribute__((always_inline)) {
    const float* xs = x + s * 16 * x_ld;
    const float* gs = dy + s * 16 * dy_ld;
    const float* ys = DACT ? yact + s * 16 * dy_ld : nullptr;
    // uniform row pointers + ONE per-lane byte offset per operand block.  (hipcc still spends a 64-bit vector add per load on
    // them -- loop strength reduction makes per-lane pointer inductions out of `uniform base + lane offset`; buffer loads with a
    // scalar row offset, as the forward kernel uses them, have no vector address arithmetic at all but cost 75 spilled
    // registers in THIS kernel in both forms tried -- descriptor per step, descriptor per launch: measured, not kept)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float* xr = xs + j * x_ld;
      const float* gr = gs + j * dy_ld;
      const float* yr = DACT ? ys + j * dy_ld : nullptr;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        r.a[t][j] = at(xr, offa[t]);
        r.b[t][j] = at(gr, offb[t]);
        if constexpr (DACT) r.y[t][j] = at(yr, offb[t]);
      }
    }
  };
  auto load_tail = [&](long s, Raw& r) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      long row = s * 16 + 8 * hi + j;
      row = row < m ? row : m - 1;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        r.a[t][j] = x[row * x_ld + (offa[t] / 4u - (unsigned)(8 * hi * x_ld))];
        r.b[t][j] = dy[row * dy_ld + (offb[t] / 4u - (unsigned)(8 * hi * dy_ld))];
        if constexpr (DACT) r.y[t][j] = yact[row * dy_ld + (offb[t] / 4u - (unsigned)(8 * hi * dy_ld))];
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  GhCol ca[2], cb[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) { ca[t] = GhCol{0, 0.f, 0.f}; cb[t] = GhCol{0, 0.f, 0.f}; }
  float bsum[2] = {0.f, 0.f};
  const bool want_bsum = part_db && blockIdx.y == 0 && wr == 0;

  // four instructions for eight values (fmaxf is an IEEE maxNum: hipcc quiets every operand with an extra v_max first)
  auto max8 = [&](const float (&v)[8]) __attribute__((always_inline)) {
    float t;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(t) : "v"(v[0]), "v"(v[1]), "v"(v[2]));
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(t) : "v"(t), "v"(v[3]), "v"(v[4]));
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(t) : "v"(t), "v"(v[5]), "v"(v[6]));
    asm("v_max_f32 %0, %1, |%2|" : "=v"(t) : "v"(t), "v"(v[7]));
    return t;
  };
  // slow path: new scale of one column from its running maximum; returns the change of the exponent
  auto rescale_col = [&](GhCol& c, float stepmax) __attribute__((always_inline)) {
    float mx = fmaxf(stepmax, __shfl_xor(stepmax, 32, 64));      // both lane halves hold rows of the same column
    mx = fmaxf(c.run, mx);
    c.run = mx;
    // A column that has only seen zeros keeps (k, lim) = (0, 0): its limit must NOT be recomputed from k = 0 when another
    // column of the wave brings it here -- 2^15 would then wave through its first real values unscaled (f16 denormals: the
    // cfg4 batch, whose padded gradient rows are zero, lost four digits of dW this way).
    if (!(mx > c.lim)) return 0;
    const int kn = 13 - __builtin_amdgcn_frexp_expf(mx);         // the maximum lands in [2^12, 2^13): two bits of headroom
    const int d = kn - c.k;
    c.k = kn;
    c.lim = __builtin_ldexpf(0.99951171875f, 16 - kn);      // |v| 2^k <= 65504, the largest f16: 8x above the new maximum
    return d;
  };
  auto mma = [&](long s, Raw& r, auto tailc) __attribute__((always_inline)) {
    constexpr bool TAIL = decltype(tailc)::value;
    if constexpr (DACT || TAIL) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float g = r.b[t][j];
          if constexpr (DACT) {
            const float a = r.y[t][j];
            float d = __builtin_fmaf(__builtin_fmaf(c2, a, c1), a, c0);
            d = relu ? (a > 0.f ? 1.f : 0.f) : d;
            g *= d;
          }
          if constexpr (TAIL) {
            const bool ok = s * 16 + 8 * hi + j < m;
            g = ok ? g : 0.f;
            r.a[t][j] = ok ? r.a[t][j] : 0.f;
          }
          r.b[t][j] = g;
        }
    }
    float ma[2], mb[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      ma[t] = max8(r.a[t]);
      mb[t] = max8(r.b[t]);
      if (want_bsum) {                       // uniform: the waves whose column sums are stored
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum[t] += r.b[t][j];
      }
    }
    const bool over = ma[0] > ca[0].lim || ma[1] > ca[1].lim || mb[0] > cb[0].lim || mb[1] > cb[1].lim;
    if (__builtin_amdgcn_ballot_w64(over) != 0) {                 // wave-uniform, rare after the first step
      int da_[2], db_[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) { da_[t] = rescale_col(ca[t], ma[t]); db_[t] = rescale_col(cb[t], mb[t]); }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r16 = 0; r16 < 16; ++r16) {
          const int dr = __shfl(da_[mt], (r16 & 3) + 8 * (r16 >> 2) + 4 * hi, 64);    // the x column of this accumulator row
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt][r16] = __builtin_ldexpf(acc[mt][nt][r16], dr + db_[nt]);
        }
    }
    u32x4 Ah[2], Al[2], Bh[2], Bl[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned h, l;
        splith_pair(__builtin_ldexpf(r.a[t][2 * q], ca[t].k), __builtin_ldexpf(r.a[t][2 * q + 1], ca[t].k), h, l);
        Ah[t][q] = h; Al[t][q] = l;
        splith_pair(__builtin_ldexpf(r.b[t][2 * q], cb[t].k), __builtin_ldexpf(r.b[t][2 * q + 1], cb[t].k), h, l);
        Bh[t][q] = h; Bl[t][q] = l;
      }
    // product-major over the four accumulators, smallest terms first
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma_f16(Al[mt], Bh[nt], acc[mt][nt]);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma_f16(Ah[mt], Bl[nt], acc[mt][nt]);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma_f16(Ah[mt], Bh[nt], acc[mt][nt]);
  };

  // the ragged last k-step of the tensors (rows beyond m) belongs to the last workgroup that has steps: peeled off
  const bool ragged_last = (m % 16 != 0) && s1 == nsteps && s0 < s1;
  const long s1f = ragged_last ? s1 - 1 : s1;
  if (s0 < s1f) {
    Raw ra, rb;
    const long sl = s1f - 1;
    load(s0, ra);
    long s = s0;
    for (; s + 1 < s1f; s += 2) {
      load(s + 1, rb);
      __builtin_amdgcn_sched_barrier(0);       // the requests of step s + 1 stay IN FRONT of the arithmetic of step s
      mma(s, ra, std::false_type{});
      __builtin_amdgcn_sched_barrier(0);
      load(s + 2 < sl ? s + 2 : sl, ra);
      __builtin_amdgcn_sched_barrier(0);
      mma(s + 1, rb, std::false_type{});
      __builtin_amdgcn_sched_barrier(0);
    }
    if (s < s1f) mma(s, ra, std::false_type{});
  }
  if (ragged_last) {
    Raw rt;
    load_tail(s1 - 1, rt);
    mma(s1 - 1, rt, std::true_type{});
  }

  // ---- the workgroup's partial dW block, unscaled -----------------------------------------------------------------
  float* pw = part_dw + (long)blockIdx.x * din * dout;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r16 = 0; r16 < 16; ++r16) {
      const int rr = (r16 & 3) + 8 * (r16 >> 2) + 4 * hi;
      const int kr = __shfl(ca[mt].k, rr, 64);
      const int row = i0 + 32 * mt + rr;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int col = j0 + 32 * nt + li;
        if (row < din && col < dout) pw[(long)row * dout + col] = __builtin_ldexpf(acc[mt][nt][r16], -(kr + cb[nt].k));
      }
    }
  if (want_bsum) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const float v = bsum[nt] + __shfl_xor(bsum[nt], 32, 64);
      const int col = j0 + 32 * nt + li;
      if (hi == 0 && col < dout) part_db[(long)blockIdx.x * dout + col] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same weight gradient with every value split ONCE (gemmh_wgradl_kernel, the shipped route without a fused activation
// derivative).  In the register kernel above each x fragment is split by the four waves that multiply with it and each dy
// fragment by two: ~200 vector instructions per wave and 16 rows, 2.4 * 10^7 per launch at 117,888 rows -- VALU-bound at 88-94 us
// (profiles/r04_gemmh_history.txt).  Here a STAGE of 32 rows is split once and shared through LDS:
//   * wave w loads dy tile w (32 columns; waves 4-7 also x tile w - 4) of the stage as coalesced dword fragments, keeps the
//     online column scales of ITS tiles, splits, and writes the (high, low) fragments lane-linearly to the stage buffer
//     [12 tiles][2 k-steps][2 pieces][1 KiB] (48 KB, two buffers) next to the tiles' current exponents;
//   * every wave then multiplies its [64 x 64] block out of the other buffer: 8 fragment reads and 12 MFMAs per k-step; a wave
//     that finds an exponent of one of its tiles changed rescales its accumulators first (exact, rare);
//   * raw fragments travel two stages ahead of the split (three register sets), one barrier per stage.
// ------------------------------------------------------------------------------------------------------------------
constexpr int GWL_BUF = 12 * 2 * 2 * 64;             // u32x4 entries of one stage buffer
constexpr size_t GWL_LDS = 2 * (size_t)GWL_BUF * 16 + 2 * 12 * 32 * 4;

__global__ __launch_bounds__(512, 2) void gemmh_wgradl_kernel(const float* __restrict__ x, long x_ld, const float* __restrict__ dy,
                                                              long dy_ld, long m, int din, int dout, long stages_per_block,
                                                              float* __restrict__ part_dw, float* __restrict__ part_db) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  u32x4* lbuf = reinterpret_cast<u32x4*>(dsm);
  int* kbuf = reinterpret_cast<int*>(dsm + 2 * (size_t)GWL_BUF * 16);           // [2][12][32]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, li = lane & 31, hi = lane >> 5;
  const int wr = wave >> 2, wc = wave & 3;
  const int i0 = blockIdx.y * 128, j0 = blockIdx.z * 256;
  const bool has_x = wave >= 4;                         // uniform: this wave also stages an x tile
  const long nfull = m / 32;                            // stages without a row mask
  const long s0 = (long)blockIdx.x * stages_per_block;
  long s1 = s0 + stages_per_block;
  if (s1 > nfull) s1 = nfull;
  const bool tail_here = (m % 32 != 0) && blockIdx.x == gridDim.x - 1;           // the ragged last stage: last row range

  // ---- staging side: columns of this wave's tiles (clamped: what a column beyond the matrix contributes is never stored) ----
  const int cy = j0 + 32 * wave + li, cx = i0 + 32 * (wave & 3) + li;
  const unsigned offy = 4u * (unsigned)(8 * hi * dy_ld + (cy < dout ? cy : dout - 1));
  const unsigned offx = 4u * (unsigned)(8 * hi * x_ld + (cx < din ? cx : din - 1));
  struct Raw { float y[2][8], a[2][8]; };
  auto at = [](const float* base, unsigned byte_off) __attribute__((always_inline)) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
  };
  auto load = [&](long st, Raw& r) __attribute__((always_inline)) {
    if constexpr (GH_VARIANT == 7) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) { r.y[q][j] = 1e-3f * (float)(lane + j + (int)st); r.a[q][j] = 0.5f + (float)j; }
      return;
    }
    // (plain pointers: uniform row pointer + one per-lane byte offset.  Buffer loads with scalar row offsets, as the forward
    // kernel uses them, cost this kernel 185 spilled registers -- as they did the register-split kernel above)
    const float* gs = dy + st * 32 * dy_ld;
    const float* xs = x + st * 32 * x_ld;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) r.y[q][j] = at(gs + (16 * q + j) * dy_ld, offy);
    if (has_x) {                                         // ONE uniform branch around the sixteen x requests
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) r.a[q][j] = at(xs + (16 * q + j) * x_ld, offx);
    }
  };
  auto load_tail = [&](long st, Raw& r) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const long row = st * 32 + 16 * q + 8 * hi + j;
        const bool ok = row < m;
        const long rc = ok ? row : m - 1;
        const float vy = dy[rc * dy_ld + (cy < dout ? cy : dout - 1)];
        const float vx = x[rc * x_ld + (cx < din ? cx : din - 1)];
        r.y[q][j] = ok ? vy : 0.f;
        r.a[q][j] = ok ? vx : 0.f;
      }
  };
  auto max8 = [&](const float (&v)[8]) __attribute__((always_inline)) {
    float t;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(t) : "v"(v[0]), "v"(v[1]), "v"(v[2]));
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(t) : "v"(t), "v"(v[3]), "v"(v[4]));
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(t) : "v"(t), "v"(v[5]), "v"(v[6]));
    asm("v_max_f32 %0, %1, |%2|" : "=v"(t) : "v"(t), "v"(v[7]));
    return t;
  };
  auto rescale_col = [&](GhCol& c, float stepmax) __attribute__((always_inline)) {
    float mx = fmaxf(stepmax, __shfl_xor(stepmax, 32, 64));      // both lane halves hold rows of the same column
    mx = fmaxf(c.run, mx);
    c.run = mx;
    if (!(mx > c.lim)) return;                                   // (a column of zeros keeps (k, lim) = (0, 0): see above)
    const int kn = 13 - __builtin_amdgcn_frexp_expf(mx);
    c.k = kn;
    c.lim = __builtin_ldexpf(0.99951171875f, 16 - kn);      // |v| 2^k <= 65504, the largest f16: 8x above the new maximum
  };
  GhCol sy{0, 0.f, 0.f}, sx{0, 0.f, 0.f};
  float bsum = 0.f;
  const bool want_bsum = part_db && blockIdx.y == 0;
  // staging in two parts: the scale check (vector maxima, a rare wave-uniform branch) and the emission of one fragment
  auto split_check = [&](Raw& r) __attribute__((always_inline)) {
    const float my = fmaxf(max8(r.y[0]), max8(r.y[1]));
    float mx = 0.f;
    if (has_x) mx = fmaxf(max8(r.a[0]), max8(r.a[1]));
    const bool over = my > sy.lim || (has_x && mx > sx.lim);
    if (__builtin_amdgcn_ballot_w64(over) != 0) {                 // wave-uniform, rare after the first stage
      rescale_col(sy, my);
      if (has_x) rescale_col(sx, mx);
    }
  };
  auto emit = [&](const float (&v)[8], int k, int tile, int q, int buf) __attribute__((always_inline)) {
    if constexpr (GH_VARIANT == 6) { if (v[0] != 1.2345e-30f) return; }
    u32x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned hh, ll;
      splith_pair(__builtin_ldexpf(v[2 * e], k), __builtin_ldexpf(v[2 * e + 1], k), hh, ll);
      h[e] = hh; l[e] = ll;
    }
    u32x4* d = lbuf + (size_t)buf * GWL_BUF + (size_t)((tile * 2 + q) * 2) * 64 + lane;
    d[0] = h; d[64] = l;
  };
  auto emit_y = [&](Raw& r, int q, int buf) __attribute__((always_inline)) {
    emit(r.y[q], sy.k, 4 + wave, q, buf);
    if (want_bsum) {
#pragma unroll
      for (int j = 0; j < 8; ++j) bsum += r.y[q][j];
    }
    if (q == 1) kbuf[buf * 12 * 32 + (4 + wave) * 32 + li] = sy.k;
  };
  auto emit_x = [&](Raw& r, int q, int buf) __attribute__((always_inline)) {
    if (has_x) {                                                  // uniform
      emit(r.a[q], sx.k, wave & 3, q, buf);
      if (q == 1) kbuf[buf * 12 * 32 + (wave & 3) * 32 + li] = sx.k;
    }
  };
  auto split = [&](Raw& r, int buf) __attribute__((always_inline)) {
    split_check(r);
    emit_y(r, 0, buf); emit_y(r, 1, buf); emit_x(r, 0, buf); emit_x(r, 1, buf);
  };

  // ---- multiplying side: the [64 x 64] block of x tiles 2 wr, 2 wr + 1 and dy tiles 2 wc, 2 wc + 1 ----------------------------
  f32x16 acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  int ka[2] = {0, 0}, kb2[2] = {0, 0};                  // exponents the accumulators are scaled with
  auto mult_check = [&](int buf) __attribute__((always_inline)) {
    const int* kb = kbuf + buf * 12 * 32;
    int na[2], nb[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { na[t] = kb[(2 * wr + t) * 32 + li]; nb[t] = kb[(4 + 2 * wc + t) * 32 + li]; }
    const bool changed = na[0] != ka[0] || na[1] != ka[1] || nb[0] != kb2[0] || nb[1] != kb2[1];
    if (__builtin_amdgcn_ballot_w64(changed) != 0) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int da = na[mt] - ka[mt];
#pragma unroll
        for (int r16 = 0; r16 < 16; ++r16) {
          const int dr = __shfl(da, (r16 & 3) + 8 * (r16 >> 2) + 4 * hi, 64);    // the x column of this accumulator row
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt][r16] = __builtin_ldexpf(acc[mt][nt][r16], dr + (nb[nt] - kb2[nt]));
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) { ka[t] = na[t]; kb2[t] = nb[t]; }
    }
  };
  struct Frags { u32x4 ah[2], al[2], bh[2], bl[2]; };
  auto read_frags = [&](Frags& f, int q, int buf) __attribute__((always_inline)) {
    const u32x4* b = lbuf + (size_t)buf * GWL_BUF;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const u32x4* ea = b + (size_t)(((2 * wr + t) * 2 + q) * 2) * 64 + lane;
      const u32x4* eb = b + (size_t)(((4 + 2 * wc + t) * 2 + q) * 2) * 64 + lane;
      f.ah[t] = ea[0]; f.al[t] = ea[64]; f.bh[t] = eb[0]; f.bl[t] = eb[64];
    }
  };
  // the twelve MFMAs of a k-step in two halves (smallest terms first)
  auto mma_lo = [&](const Frags& f) __attribute__((always_inline)) {
    if constexpr (GH_VARIANT == 5) { acc[0][0][0] += __uint_as_float(f.al[0][0] ^ f.bh[0][0] ^ f.ah[1][1] ^ f.bl[1][1]); return; }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma_f16(f.al[mt], f.bh[nt], acc[mt][nt]);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[0][nt] = mfma_f16(f.ah[0], f.bl[nt], acc[0][nt]);
  };
  auto mma_hi = [&](const Frags& f) __attribute__((always_inline)) {
    if constexpr (GH_VARIANT == 5) { acc[1][1][0] += __uint_as_float(f.ah[0][2] ^ f.bh[1][3] ^ f.al[1][2] ^ f.bl[0][3]); return; }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[1][nt] = mfma_f16(f.ah[1], f.bl[nt], acc[1][nt]);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma_f16(f.ah[mt], f.bh[nt], acc[mt][nt]);
  };
  auto multiply = [&](int buf) __attribute__((always_inline)) {
    mult_check(buf);
    Frags f;
    read_frags(f, 0, buf); mma_lo(f); mma_hi(f);
    read_frags(f, 1, buf); mma_lo(f); mma_hi(f);
  };

  int buf = 0;
  GHP_DECL
  if (s0 < s1) {
    const long sl = s1 - 1;                              // requests past the range repeat its last stage (valid, unused)
    Raw r0, r1, r2;
    load(s0, r0);
    load(s0 + 1 < sl ? s0 + 1 : sl, r1);
    load(s0 + 2 < sl ? s0 + 2 : sl, r2);
    split(r0, 0);
    gh_barrier_lds();
    // iteration for stage s: request stage s + 3 into the set stage s came from, split stage s + 1 into the other buffer,
    // multiply stage s.  No exit between a request and its use; stages past the range are requested (clamped) but neither
    // split nor multiplied.
    // Steady state (stages s, s + 1 both inside the range): the vector work of staging stage s + 1 is laid BETWEEN the
    // MFMAs of stage s, six at a time -- the matrix pipe runs beside the vector ALU, and with the halves of a stage one after
    // the other both waves of a SIMD collided on each pipe in turn (75 us at 117,888 rows; WAIT_ANY 42 %).
    auto iter_fast = [&](long s, Raw& rnext, Raw& rfree) __attribute__((always_inline)) {
      __builtin_amdgcn_sched_barrier(0);
      GHP(7)                                              // (loop overhead / what the previous iteration left)
      load(s + 3 < sl ? s + 3 : sl, rfree);
      __builtin_amdgcn_sched_barrier(0);
      GHP(0)
      split_check(rnext);
      GHP(1)
      mult_check(buf);
      Frags f;
      read_frags(f, 0, buf);
      __builtin_amdgcn_sched_barrier(0);
      GHP(2)
      mma_lo(f);
      emit_y(rnext, 0, buf ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_hi(f);
      emit_y(rnext, 1, buf ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      GHP(3)
      read_frags(f, 1, buf);
      __builtin_amdgcn_sched_barrier(0);
      GHP(4)
      mma_lo(f);
      emit_x(rnext, 0, buf ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_hi(f);
      emit_x(rnext, 1, buf ^ 1);
      GHP(5)
      gh_barrier_lds();
      GHP(6)
      buf ^= 1;
    };
    auto iter = [&](long s, Raw& rnext, Raw& rfree) __attribute__((always_inline)) {
      __builtin_amdgcn_sched_barrier(0);
      load(s + 3 < sl ? s + 3 : sl, rfree);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < s1) split(rnext, buf ^ 1);             // uniform
      if (s < s1) multiply(buf);
      gh_barrier_lds();
      buf ^= 1;
    };
    long s = s0;
    for (; s + 3 < s1; s += 3) {                          // stages s .. s + 3 exist: no guards
      iter_fast(s, r1, r0);
      iter_fast(s + 1, r2, r1);
      iter_fast(s + 2, r0, r2);
    }
    for (; s < s1; s += 3) {
      iter(s, r1, r0);
      iter(s + 1, r2, r1);
      iter(s + 2, r0, r2);
    }
    // (the loop leaves `buf` wherever its last, possibly idle, iterations put it: every wave agrees, nothing is pending)
  }
  if (tail_here) {
    Raw rt;
    load_tail(nfull, rt);
    if (!has_x) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) rt.a[q][j] = 0.f;
    }
    split(rt, buf);
    gh_barrier_lds();
    multiply(buf);
  }

  GHP_FLUSH
  // ---- the workgroup's partial dW block, unscaled -----------------------------------------------------------------
  float* pw = part_dw + (long)blockIdx.x * din * dout;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r16 = 0; r16 < 16; ++r16) {
      const int rr = (r16 & 3) + 8 * (r16 >> 2) + 4 * hi;
      const int kr = __shfl(ka[mt], rr, 64);
      const int row = i0 + 64 * wr + 32 * mt + rr;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int col = j0 + 64 * wc + 32 * nt + li;
        if (row < din && col < dout) pw[(long)row * dout + col] = __builtin_ldexpf(acc[mt][nt][r16], -(kr + kb2[nt]));
      }
    }
  if (want_bsum) {
    const float v = bsum + __shfl_xor(bsum, 32, 64);
    if (hi == 0 && cy < dout) part_db[(long)blockIdx.x * dout + cy] = v;
  }
}

// (below ~16k rows the 128 row ranges are a handful of stages each: sparse.py's 4,457 rows ran 15.6 us here, ~10 in gemm3's kernel)
bool gemmh_wgrad_ok(int din, int dout, long m) { return din > 96 && dout > 128 && m >= 16384; }

// nblocks row ranges; partial layout as gemm3_wgrad: part_dw[nblocks][din][dout], part_db[nblocks][dout]
int launch_gemmh_wgrad(const float* x, long x_ld, const float* dy, long dy_ld, long m, int din, int dout, float* part_dw,
                       float* part_db, int nblocks, hipStream_t s, const float* yact, int act) {
  const dim3 grid((unsigned)nblocks, (unsigned)((din + 127) / 128), (unsigned)((dout + 255) / 256));
  static const char* lknob = dev_knob("KGCN_WGRADL");          // development: "0" = the register-split kernel only
  if (!(yact && act != KGCN_ACT_NONE) && !(lknob && lknob[0] == '0')) {
    static thread_local bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemmh_wgradl_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                kLdsBytes);
      attr_set = true;
    }
    const long nfull = m / 32, spb = (nfull + nblocks - 1) / nblocks;
    hipLaunchKernelGGL(gemmh_wgradl_kernel, grid, dim3(512), GWL_LDS, s, x, x_ld, dy, dy_ld, m, din, dout, spb, part_dw, part_db);
    return check_launch("gemmh_wgradl_kernel");
  }
  const long nsteps = (m + 15) / 16, spb = (nsteps + nblocks - 1) / nblocks;
  const float c0 = act == KGCN_ACT_TANH ? 1.f : 0.f, c1 = act == KGCN_ACT_SIGMOID ? 1.f : 0.f, c2 = -1.f;
  if (yact && act != KGCN_ACT_NONE)
    hipLaunchKernelGGL(gemmh_wgrad_kernel<true>, grid, dim3(512), 0, s, x, x_ld, dy, dy_ld, m, din, dout, spb, part_dw, part_db,
                       yact, c0, c1, c2, act == KGCN_ACT_RELU ? 1 : 0);
  else
    hipLaunchKernelGGL(gemmh_wgrad_kernel<false>, grid, dim3(512), 0, s, x, x_ld, dy, dy_ld, m, din, dout, spb, part_dw, part_db,
                       nullptr, 0.f, 0.f, 0.f, 0);
  return check_launch("gemmh_wgrad_kernel");
}

}  // namespace kgcn

#ifdef KGCN_PROBE
extern "C" int kgcn_gh_probe_set(void* buf) {
  long long* p = static_cast<long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(kgcn::gh_probe), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif
