// One-pass backward of a wide dense layer (round 5):  y = act(x W + b), W [din x dout], 128 < din, dout <= 256
// (kgcn/layers.py:248,260 Keras Dense inside GraphDense; :99-100 / :112 the X.W part of GraphConv; example_model/model_gin.py:45-54,
// example_model/model_multitask.py:51-57).  Until round 4 this was two kernels making SIX passes over [m, 256] tensors:
//   gemmh_fwd_kernel<DK>  reads dY and the saved activation, WRITES d pre-activation and dX        (4 passes)
//   gemmh_wgradl_kernel   READS d pre-activation back, reads x                                     (2 passes)
// -- 782 + 444 MB of HBM traffic per layer at 200,000 rows (profiles/r04_s_cfg5_rocprof.txt).  Here ONE sweep over (dY, a, x)
// produces dX, dW and dbias; d pre-activation lives only in LDS (as f16 pieces) and is never written: four passes.
//
// Why it is not simply "both products in the old kernel": the weight-stationary dX product keeps W' (h and l f16 pieces of
// 256 x 256 values: 256 KB) in registers and the weight gradient needs 256 x 256 fp32 accumulators (256 KB) -- together the whole
// 512 KB register file of a CU.  So the work is cut in two along the layer's INPUT columns and given to a PAIR of workgroups
// (b, b + 8: the same XCD, whose L2 serves the second read of the shared rows -- the old weight-gradient kernel already read dY
// twice this way, PMC traffic x1.14):
//   workgroup `half` of a pair:  dX[:, 128 half .. +128)  =  dpre . W^T[:, that column range]      (needs all of dpre: dY, a)
//                                dW[128 half .. +128, :]  =  x[:, that column range]^T . dpre       (needs HALF of x)
//   HBM: dY + a + x in, dX out (the pair's second read of dY and a hits L2 / the memory-side cache);  per workgroup W' is
//   128 KB of registers and dW 128 KB of accumulators: 8 waves x 256 registers, two per SIMD, in two roles (below).
//
// One stage = 32 rows.  Waves 0..3 multiply for dX, waves 4..7 for dW (roles: see gemmb_kernel), all eight stage:
//   staging    rows 4 w .. 4 w + 3 of the stage travel HBM -> registers as 1 KiB rows (lane = 4 consecutive columns; dY and the
//              saved activation, one stage ahead), dpre = dY (.) act'(a) [+ the read-out's broadcast gradient], row maximum by a
//              DPP wave reduction -> row exponent kr (scalar), pieces h = f16(dpre 2^kr), l = f16(dpre 2^kr - h) -> LDS ONCE,
//              in an image that serves both products (below); dbias = column sums on the way.
//   dX         [32 rows x 32 columns] per wave: A operand = dpre pieces as 16-byte ROW reads, B = its slice of W' (128 registers,
//              whole launch), 16 k-steps x 3 products, two accumulator chains; epilogue 2^-(kr + kc), buffer stores.
//   dW         [32 x-columns x 256] per wave = 8 accumulator tiles (128 registers, whole launch).  The contraction runs over the
//              ROWS, so the B operand (dpre, k = row) is read with ds_read_b64_tr_b16 -- the transpose read of gfx950 -- out
//              of the same image, and the A operand (x, k = row) is loaded from HBM directly in fragment layout (lane (li, hi)
//              = x[r + 8 hi + j][c + li]: a coalesced dword load IS the operand, lesson 25) and split in registers.
//   scales     dpre carries a ROW scale 2^kr (needed by dX).  Inside the weight gradient a row scale would not factor out
//              of the sum over rows -- so x is counter-scaled: x''[r, i] = x[r, i] 2^(K_i - kr[r]) and x''[r, i] dpre'[r, j]
//              = x dpre 2^K_i exactly; K_i is the ONLINE per-column exponent of x'' (the wave owns its 32 x columns: when a
//              value would leave the f16 range the exponent drops and the accumulator rows are rescaled by the exact power
//              of two, as in gemmh_wgrad_kernel).  A row whose dpre is all zero gets kr = 120: its x'' vanishes instead of
//              setting the column scale; a row with +-inf / NaN takes kr from its FINITE values (the non-finite element stays
//              non-finite at any scale, the others keep their places in dW).  Error: the products l.H + h.L + h.H as in gemmh.hip -- 2^-23 sum |x dpre| plus the
//              f16-denormal tail m 2^-40 max_r|x''[r, i]| max|dpre'| -- the class documented in include/kgcn_hip.h (route 3).
// LDS image of a stage (per piece 20,480 bytes; tools/gemmb_layout_check.py emulates it and the three access patterns):
//   byte(r, f) = (r >> 2) 2560 + (r & 3) 64 + (f >> 5) 320 + ((((f & 31) >> 3) ^ ((r >> 2) & 3)) << 4) + (f & 7) 2
//   -- four consecutive rows' 64-byte segments of a 32-column block are 256 contiguous bytes (one bank row for a transpose read
//   of 4 rows x 32 columns), consecutive column blocks advance by 320 = 256 + 64 so that they rotate through the four
//   quarters (the 8-byte row writes of 16 lanes cover 32 distinct banks), 16-byte chunks are rotated by the row quad (the
//   16-lane groups of ds_read_b128 see 16 distinct bank quads).  Everything that depends on the k-step / column tile is an
//   IMMEDIATE offset on one of two lane bases.
#include "gemmh.h"

namespace kgcn {

constexpr int GB_R = 32;                      // rows per stage
#ifdef KGCN_ABL_HOT                           // development: every workgroup re-reads (and re-writes) its first stage -- the kernel without HBM traffic
#define GB_HOT_STEP 0
#else
#define GB_HOT_STEP G
#endif
constexpr int GB_PLANE = 20480;               // bytes of one piece plane
constexpr int GB_BUF = 2 * GB_PLANE;          // h | l
// k-steps of W' held in registers; the others live in LDS (below).  The form that adds the read-out's gradient to a row gradient
// (BC = 1) keeps eight more staging registers alive through the multiplication: four k-steps fewer in registers there.
#ifndef GB_DOT_WREG
#define GB_DOT_WREG 8
#endif
#ifndef GB_DOT_KS
#define GB_DOT_KS 0                             // the k-step of a stage at which the dot form requests its dot operand
#endif
// W' k-steps a dX wave keeps in registers (the rest wait in LDS).  The dot form gives two more of them to LDS: its sixteen values of
// the dot operand are requested at the HEAD of a stage -- a whole multiplication ahead of the epilogue that uses them -- and live in
// the registers that frees (requested at k-step 5 they were ~1,500 cycles ahead of their use, under the loaded HBM latency:
// 248 us per launch at 200,000 rows against 196 us of the plain form that moves the same bytes, profiles/r05_p_cfg5_rocprof.txt)
__host__ __device__ constexpr int gb_wreg(bool dot) { return dot ? GB_DOT_WREG : 10; }
__host__ __device__ constexpr size_t gb_lds(bool dot) { return 2 * (size_t)GB_BUF + 2 * GB_R * 4 + 4 * (size_t)(16 - gb_wreg(dot)) * 2 * 1024; }
#ifndef GB_DW_ROWS
#define GB_DW_ROWS 0                            // rows (of a wave quarter's eight per stage) staged by the dW wave
#endif
constexpr int GB_ZERO_ROW_K = 120;            // row exponent of an all-zero dpre row: x 2^(K - 120) vanishes

#ifdef KGCN_PROBE   // development: per-wave cycle sums per phase of gemmb_kernel (tools/gemmb_probe.py)
__device__ long long* gb_probe = nullptr;
#define GBP_DECL long long pt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long pc_ = __builtin_readcyclecounter();
#define GBP(k) { const long long n_ = __builtin_readcyclecounter(); pt_[k] += n_ - pc_; pc_ = n_; }
#define GBP_FLUSH if (gb_probe && lane == 0) { for (int k_ = 0; k_ < 8; ++k_) gb_probe[((long)blockIdx.x * 8 + wave) * 8 + k_] = pt_[k_]; }
#else
#define GBP_DECL
#define GBP(k)
#define GBP_FLUSH
#endif

typedef short gb_i16x4 __attribute__((ext_vector_type(4)));
#define GB_LDS_AS __attribute__((address_space(3)))
__device__ __forceinline__ unsigned gb_lds_off(const void* p) { return (unsigned)(uintptr_t)(const GB_LDS_AS unsigned char*)p; }
__device__ __forceinline__ u32x4 gb_ld128(unsigned a) { return *(const GB_LDS_AS u32x4*)(uintptr_t)a; }
__device__ __forceinline__ void gb_st64(unsigned a, unsigned lo, unsigned hi) {
  const u32x2 v = {lo, hi};
  *(GB_LDS_AS u32x2*)(uintptr_t)a = v;
}
__device__ __forceinline__ u32x2 gb_ld_tr16(unsigned a) {
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((GB_LDS_AS gb_i16x4*)(uintptr_t)a));
}

// wave maximum of the FINITE magnitudes of a row (cold path of the staging: a row that holds +-inf / NaN).  Out of line: inlined into
// each of the eight rows of a stage it was ~150 instructions of never-taken code per row inside the loop body.
__device__ __attribute__((noinline)) unsigned gb_finite_row_max(f32x4 v) {
  unsigned q = 0u;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned av = __float_as_uint(v[e]) & 0x7fffffffu;
    q = (av < 0x7f800000u && av > q) ? av : q;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned u = (unsigned)__shfl_xor((int)q, o, 64);
    q = u > q ? u : q;
  }
  return (unsigned)__builtin_amdgcn_readfirstlane((int)q);
}

// DK: 0 plain (dpre = the incoming gradient); 1 dpre = g (.) (c0 + c1 a + c2 a^2) (sigmoid / tanh derivative in the layer
// OUTPUT a); 2 relu (a > 0).  da.bc: the read-out's pooled gradient, broadcast over the bc_n node rows of a graph and added
// to (BC = 1) or standing for (BC = 2) the row gradient -- kgcn_dense_dx_dact_gather_f32's operand; BC = 0: none.  A template
// parameter, not a uniform branch: the first build tested da.bc per row at run time -- two dozen extra basic blocks in the loop
// body, registers live across all of them, 172 spilled VGPRs.
// DOT: the dX product is not stored; its inner product with `dx` (READ as an [m, ndim] operand) goes to da.dot_part, one partial per
// workgroup -- d epsilon of a GINAggregate whose input needs no gradient (kgcn/layers.py:469), as in gemmh_fwd_kernel's dot form.
//
// EIGHT waves, two per SIMD, in two ROLES (wave w and w + 4 share a SIMD):
//   waves 0..3 ("dX")  STAGE the rows (eight per wave and stage: dY / a rows -> dpre -> row exponent -> pieces -> LDS, requested a
//                      stage ahead), keep W' (k-steps 0..9 in 80 accumulator-file registers, 10..15 in LDS) and multiply the
//                      stage's 32 rows with their 32 dX columns: 48 MFMAs, epilogue, 16 stores per stage
//   waves 4..7 ("dW")  keep the eight dW accumulator tiles of their 32 x columns (128 accumulator-file registers), load x in
//                      fragment layout, counter-scale and split it, and multiply with the stage's dpre read transposed: 48 MFMAs
// The first build of this kernel ran both roles in ONE wave per SIMD (4 waves x 512 registers): with nobody to fill its
// issue slots a lone wave spent 6,400 cycles of every stage on ~900 non-MFMA instructions next to 3,400 cycles of MFMAs, one
// after the other (tools/gemmb_probe.py, profiles/r05_gemmb_history.txt: 248 us at 200,000 rows).  Two waves of 256 registers
// each hold HALF of the stationary data and fill each other's gaps.  Registers decide who stages: a wave has 128 VGPRs next to
// its 128 accumulator-file registers; the dW role's accumulators fill its accumulator file, so its x fragments, their pieces and
// the transposed fragments are all it has room for -- with four staged rows per wave on top the allocator spilled ACCUMULATORS to
// scratch memory (181-413 spilled registers); the dX role leaves 32 accumulator-file registers free for the rows in flight.
template <int DK, int BC, bool DOT>
__global__ __launch_bounds__(512, 2) void gemmb_kernel(const float* __restrict__ g, long m, int kdim, long ld,
                                                       const float* __restrict__ x, int ndim, long x_ld,
                                                       const u32x4* __restrict__ tab, float* __restrict__ dx, long dx_ld,
                                                       float* __restrict__ part_dw, float* __restrict__ part_db, GhDact da,
                                                       int niter) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w4 = wave & 3;                                    // the wave's column tile inside its role
  // pair (q0, half): workgroups b and b + 8 land on the same XCD (round-robin dispatch), i.e. behind the same L2
  const int b = (int)blockIdx.x;
  const int half = (b >> 3) & 1;
  const long q0 = ((b >> 4) << 3) | (b & 7);
  const long G = gridDim.x >> 1;
  const unsigned lds0 = gb_lds_off(dsm);
  int* rowk_base = reinterpret_cast<int*>(dsm + 2 * (size_t)GB_BUF);          // [2][32] row exponents
  GBP_DECL

  // ---- staging (dX role): rows 8 w .. 8 w + 7 of a stage; lane = 4 consecutive columns --------------------------------------
  const int c4 = 4 * lane;
  constexpr bool cok = true;                                  // kdim == 256 (launch_gemmb): every lane holds four valid columns
  unsigned wbase[2];                                          // LDS write bases of rows 8 w + 0..3 / 8 w + 4..7
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int Rr = 2 * w4 + p;
    wbase[p] = lds0 + (unsigned)(Rr * 2560 + (lane >> 3) * 320 + ((((lane >> 1) & 3) ^ (Rr & 3)) << 4) + 8 * (lane & 1));
  }
  f32x4 raw[8], ya[DK != 0 ? 8 : 1];
  // row i of this wave's share of stage `st` (dY and the saved activation), requested as soon as the registers of the same row
  // of the previous stage are free
  // (buffer loads with a row-block descriptor: rows past the end read as 0.  global_load_dwordx4 with a uniform row pointer --
  // ~3-7 cycles of issue in tools/issue_cost.hip against ~30 for a buffer load -- was measured in THIS kernel and lost: 243 us
  // against 218 us at 200,000 rows with x, dY, a and dX all moved to global accesses, every phase slower; profiles/r05_gemmb_history.txt)
  const unsigned voff_g = 16u * (unsigned)(cok ? lane : 0);
  const int ld4 = (int)(ld * 4);
  auto load_ga_row = [&](long st, auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    const long r0 = st * GB_R + 8 * w4;
    if constexpr (BC == 2) raw[i] = f32x4{0.f, 0.f, 0.f, 0.f};          // the gradient is the read-out's broadcast alone
    else raw[i] = gh_ld4(gh_rows(g, r0, 8, m, ld), voff_g, i * ld4);     // stages past the end: empty descriptor, zeros
    if constexpr (DK != 0) ya[i] = gh_ld4(gh_rows(g + da.ydiff, r0, 8, m, ld), voff_g, i * ld4);
  };
  // the read-out's gradient rows of the (at most two: bc_n >= 8) graphs this wave's eight rows of stage `st` belong to, and how
  // many of the eight belong to the first one.  Requested a whole multiplication ahead: read where they are needed they were
  // the YOUNGEST loads in flight, and vmcnt -- which counts in order -- made each of them wait for everything requested before.
  f32x4 bcv[BC != 0 ? 2 : 1];
  int bc_first = 8;
  auto load_bc = [&](long st) __attribute__((always_inline)) {
    if constexpr (BC != 0) {
      const long r0 = st * GB_R + 8 * w4;
      const long gmax = (m - 1) / da.bc_n;                    // rows beyond m are zeroed in the staging; their address stays valid
      const long gq = r0 / da.bc_n;
      bc_first = da.bc_n - (int)(r0 - gq * da.bc_n);
      const long ga = gq < gmax ? gq : gmax, gb = gq + 1 < gmax ? gq + 1 : gmax;
      bcv[0] = *reinterpret_cast<const f32x4*>(da.bc + ga * da.bc_ld + (cok ? c4 : 0));
      bcv[1] = *reinterpret_cast<const f32x4*>(da.bc + gb * da.bc_ld + (cok ? c4 : 0));
    }
  };

  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  // Staging of ONE row of stage `st` (in the registers) into buffer `buf`, in two parts that the dX loop lays between the MFMAs of
  // two consecutive k-steps (a wave's MFMAs leave its vector-ALU slots free: staged after the multiplication -- the second build --
  // the eight rows were 3,700 cycles at the END of every stage with the dW waves idle at the barrier, tools/gemmb_probe.py):
  //   part A  dpre = dY (.) act'(a) [+ read-out gradient], column sums for dbias, the row's largest magnitude (DPP wave reduction)
  //   part B  row exponent, pieces -> LDS, and the request for the same row of stage `st_next` into the registers this one leaves
  unsigned rmx = 0;
  auto row_max = [&](unsigned v) __attribute__((always_inline)) {       // wave maximum of a non-negative float pattern -> scalar
    asm volatile("s_nop 1\n"
                 "v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n"
                 "v_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n"
                 "v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n"
                 "v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n"
                 "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n"
                 "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1\n"
                 : "+v"(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
  };
  auto stage_row_a = [&](long st, auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    const long r0 = st * GB_R + 8 * w4;
    f32x4 v = raw[i];
    if constexpr (DK != 0) {
      if constexpr (BC != 0) v += i < bc_first ? bcv[0] : bcv[1];        // uniform select: the row's graph
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = ya[i][e];
        if constexpr (DK == 1) v[e] *= __builtin_fmaf(__builtin_fmaf(da.c2, a, da.c1), a, da.c0);
        else v[e] = a > 0.f ? v[e] : 0.f;
      }
    }
    // Rows >= m need no mask: dY and a were loaded as 0 through the descriptors, so relu' (a > 0) and sigmoid' (a (1 - a)) make
    // the row zero whatever the broadcast gradient added; only tanh' (1 - a^2 = 1) with a broadcast gradient keeps it alive
    if constexpr (DK == 1 && BC != 0) {
      if (r0 + i >= m) v = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) bsum[e] += v[e];
    float a;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(a) : "v"(v[0]), "v"(v[1]), "v"(v[2]));
    asm("v_max_f32 %0, %1, |%2|" : "=v"(a) : "v"(a), "v"(v[3]));
    raw[i] = v;
    rmx = row_max(__float_as_uint(a));
    unsigned mxr = rmx;                                       // (part A ends with the one branch of a row's staging: part B is straight-line code between its MFMAs)
    // A row that holds +-inf / NaN: its exponent comes from its FINITE values, so that those keep their places in the weight
    // gradient (dW[:, j] of a finite column j must not turn non-finite -- or lose the row -- because ANOTHER column of that
    // row is; the non-finite element itself splits into non-finite pieces at any scale).  Cold, wave-uniform.
    if (__builtin_expect(mxr >= 0x7f800000u, 0)) mxr = gb_finite_row_max(raw[i]);
    rmx = mxr;
  };
  auto stage_row_b = [&](long st_next, int buf, auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    const unsigned mxr = rmx;
    const int k = mxr == 0u ? GB_ZERO_ROW_K : scale_exp(__uint_as_float(mxr));          // wave-uniform
    unsigned h0, l0, h1, l1;
    splith_pair(__builtin_ldexpf(raw[i][0], k), __builtin_ldexpf(raw[i][1], k), h0, l0);
    splith_pair(__builtin_ldexpf(raw[i][2], k), __builtin_ldexpf(raw[i][3], k), h1, l1);
    const unsigned ad = wbase[i >> 2] + (unsigned)(buf * GB_BUF) + (unsigned)((i & 3) * 64);
    gb_st64(ad, h0, h1);
    gb_st64(ad + GB_PLANE, l0, l1);
    (rowk_base + GB_R * buf)[8 * w4 + i] = k;                  // same value from every lane
    load_ga_row(st_next, ic);
  };

  long t = q0;
  // Rows 0 .. GB_DX_ROWS-1 of a wave quarter's eight are staged by its dX wave, the rest by its dW wave (same w4): the dX waves are
  // the stage's critical path, the dW waves waited ~1,400 cycles at its barrier (tools/gemmb_probe.py)
  constexpr int GB_DX_ROWS = 8 - GB_DW_ROWS;

  if (wave < 4) {
    // =============================== dX role ====================================================================
    // Register files: hipcc gives every MFMA of a function its C / D in the accumulator file; W' is a B operand, which the
    // matrix pipe reads from either file.  Its fragments are pinned to accumulator-file registers (an empty asm with an "a"
    // constraint at the definition: the value's register class); left to itself the allocator kept W' in VGPRs, used the AGPRs
    // as spill slots and copied every fragment back in front of its MFMAs.  The fragments of the last k-steps wait in LDS
    // (4 KB per wave and k-step pair, written once) and are read with the stage's other fragments.
    const int nt32 = gh_nt32(ndim), kse = gh_kse(kdim);
    const int* kctab = reinterpret_cast<const int*>(tab + (long)kse * nt32 * 2 * 64);
    const int nt = 4 * half + w4;
    const int ntc = nt < nt32 ? nt : nt32 - 1;               // clamped: the columns of such a wave are never stored
    constexpr int GB_WREG = gb_wreg(DOT);
    u32x4 Wh[GB_WREG], Wl[GB_WREG];
    const unsigned wl_base = lds0 + (unsigned)(2 * GB_BUF + 2 * GB_R * 4 + w4 * (16 - GB_WREG) * 2048 + lane * 16);
    static_for<16>([&](auto kc) __attribute__((always_inline)) {
      constexpr int ks = decltype(kc)::value;
      const int kk = ks < kse ? ks : 0;
      const u32x4* e = tab + ((long)(kk * nt32 + ntc) * 2) * 64 + lane;
      const u32x4 z = {0u, 0u, 0u, 0u};
      const u32x4 h = ks < kse ? e[0] : z, l = ks < kse ? e[64] : z;
      if constexpr (ks < GB_WREG) {
        Wh[ks] = h;
        Wl[ks] = l;
        asm volatile("" : "+a"(Wh[ks]));
        asm volatile("" : "+a"(Wl[ks]));
      } else {
        *(GB_LDS_AS u32x4*)(uintptr_t)(wl_base + (unsigned)((ks - GB_WREG) * 2048)) = h;
        *(GB_LDS_AS u32x4*)(uintptr_t)(wl_base + (unsigned)((ks - GB_WREG) * 2048 + 1024)) = l;
      }
    });
    const int col = 32 * nt + li;                            // dX column of this lane
    const int ldy4 = (int)(dx_ld * 4);
    const unsigned voff_y = 4u * (unsigned)(4 * hi * dx_ld + (col < ndim ? col : 0));
    const int kcol = kctab[32 * ntc + li];
    float dotacc = 0.f;
    // A-operand rows of dX (row li): k-step even / odd
    const unsigned abaseE = lds0 + (unsigned)((li >> 2) * 2560 + (li & 3) * 64 + ((hi ^ ((li >> 2) & 3)) << 4));
    const unsigned abaseO = lds0 + (unsigned)((li >> 2) * 2560 + (li & 3) * 64 + (((2 + hi) ^ ((li >> 2) & 3)) << 4));

    auto compute_dx = [&](long st, long st_next, long st_load, int buf) __attribute__((always_inline)) {
      const int* rowk = rowk_base + GB_R * buf;
      const unsigned boff = (unsigned)(buf * GB_BUF);
      const unsigned abE = abaseE + boff, abO = abaseO + boff;
      float zz[DOT ? 16 : 1];                                 // dot form: this stage's elements of the dot operand
      // ONE accumulator chain: the wave's dependent MFMAs leave gaps on the matrix pipe that the SIMD's other wave (the dW
      // role, eight independent tiles) fills -- a second chain would cost 16 of the 128 accumulator-file registers W' needs
      f32x16 ax;
#pragma unroll
      for (int r = 0; r < 16; ++r) ax[r] = 0.f;
      struct P { u32x4 ph, pl, wh, wl; };
      auto read_p = [&](P& f, auto kc) __attribute__((always_inline)) {
        constexpr int ks = decltype(kc)::value;
        const unsigned ab = ((ks & 1) ? abO : abE) + (unsigned)((ks >> 1) * 320);
        f.pl = gb_ld128(ab + GB_PLANE);
        f.ph = gb_ld128(ab);
        if constexpr (ks >= GB_WREG) {
          f.wh = gb_ld128(wl_base + (unsigned)((ks - GB_WREG) * 2048));
          f.wl = gb_ld128(wl_base + (unsigned)((ks - GB_WREG) * 2048 + 1024));
        }
      };
      P F[17];                                                // (SSA names: F[k] dies inside k-step k)
      read_p(F[0], std::integral_constant<int, 0>{});
      static_for<16>([&](auto kc) __attribute__((always_inline)) {
        constexpr int ks = decltype(kc)::value;
        u32x4 wh, wl;
        if constexpr (ks < GB_WREG) { wh = Wh[ks]; wl = Wl[ks]; } else { wh = F[ks].wh; wl = F[ks].wl; }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ks + 1 < 16) read_p(F[ks + 1], std::integral_constant<int, (ks + 1 < 16 ? ks + 1 : 0)>{});
        if constexpr (DOT && ks == GB_DOT_KS) {
          // (with ten k-steps of W' in registers these sixteen values, requested at the head of the stage, were sixteen more registers
          // alive through the whole multiplication: 18 spilled, reloaded from scratch memory inside the loop -- see gb_wreg)
          const __amdgpu_buffer_rsrc_t rz = gh_rows(dx, st * GB_R, GB_R, m, dx_ld);
#pragma unroll
          for (int r = 0; r < 16; ++r)
            zz[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rz, (int)voff_y, ((r & 3) + 8 * (r >> 2)) * ldy4, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
        ax = mfma_f16(F[ks].pl, wh, ax);
        ax = mfma_f16(F[ks].ph, wl, ax);
        ax = mfma_f16(F[ks].ph, wh, ax);
        // BETWEEN the k-step's MFMAs: half a row of the NEXT stage's staging (row ks / 2 -> the other buffer).  The three
        // MFMAs form one dependent chain: issued back to back they hold the wave's issue port until the last one starts, and
        // vector work placed behind them overlaps only that one (the first interleaved build: 6,200 cycles for the phase
        // against 1,770 + 3,740 apart) -- the group barriers ask the scheduler for MFMA, 7 vector instructions, MFMA, ...
        if constexpr (ks / 2 < GB_DX_ROWS) {
          if constexpr ((ks & 1) == 0) stage_row_a(st_next, std::integral_constant<int, ks / 2>{});
          else stage_row_b(st_load, buf ^ 1, std::integral_constant<int, ks / 2>{});
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
      });
      load_bc(st_load);                                      // (the rows staged above used the previous request)
      GBP(1)
      __builtin_amdgcn_sched_barrier(0);
      // ---- dX <- 2^-(kr + kc) ax -----------------------------------------------------------------------------------------
      // (dword buffer stores, 2 rows x 128 bytes each.  The transposed form -- dX^T = W' dpre^T, a lane then owns 4 consecutive
      // columns of a row and stores 16 bytes -- was measured with global_store_dwordx4 and lost together with the global loads)
      if (col < ndim) {
        const __amdgpu_buffer_rsrc_t ry = gh_rows(dx, st * GB_R, GB_R, m, dx_ld);        // rows >= m: dropped / read as 0
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const u32x4 kr4 = *reinterpret_cast<const u32x4*>(rowk + 8 * rq + 4 * hi);
#pragma unroll
          for (int rj = 0; rj < 4; ++rj) {
            const float v = __builtin_ldexpf(ax[4 * rq + rj], -((int)kr4[rj] + kcol));
            if constexpr (DOT) dotacc = __builtin_fmaf(v, zz[4 * rq + rj], dotacc);
            else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, (int)voff_y, (rj + 8 * rq) * ldy4, 0);
          }
        }
      }
      GBP(2)
    };

    static_for<GB_DX_ROWS>([&](auto ic) __attribute__((always_inline)) { load_ga_row(t, ic); });
    load_bc(t);
    static_for<GB_DX_ROWS>([&](auto ic) __attribute__((always_inline)) {          // the first stage: staged without a multiplication to hide in
      stage_row_a(t, ic);
      stage_row_b(t + G, 0, ic);
    });
    load_bc(t + G);
    gh_barrier_lds();
    for (int it = 0; it < niter; ++it) {
      const int buf = it & 1;
      compute_dx(t, t + G, t + 2 * G, buf);
      gh_barrier_lds();
      GBP(5)
      t += GB_HOT_STEP;
    }
    if constexpr (DOT) {                                      // fixed order: lanes (butterfly), then the four dX waves (below)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) dotacc += __shfl_xor(dotacc, o, 64);
      if (lane == 0) reinterpret_cast<float*>(dsm + 2 * (size_t)GB_BUF)[wave] = dotacc;       // (the row exponents are done with)
    }
  } else {
    // =============================== dW role ====================================================================
    // transpose reads of dW's B operand: 16-lane group g16 = (hi, column half), lane i16 addresses row 4 t + (i16 >> 2) of
    // its 8-row half, columns 16 half16 + 4 (i16 & 3) .. + 3
    unsigned tbase[2];
    {
      const int g16 = lane >> 4, i16 = lane & 15, half16 = g16 & 1;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
        tbase[tt] = lds0 + (unsigned)(2 * hi * 2560 + (i16 >> 2) * 64 + ((((2 * half16) + ((i16 & 3) >> 1)) ^ (2 * hi + tt)) << 4) +
                                      8 * (i16 & 1));
    }
    // x in fragment layout: lane (li, hi) = x[row 16 q + 8 hi + j][i0 + li]
    const int i0 = 128 * half + 32 * w4;
    const int ca = i0 + li < ndim ? i0 + li : ndim - 1;      // clamped: what such a column contributes is never stored
    float xr[2][8];                                           // x of the stage in flight, fragment layout
    const unsigned voff_x = 4u * (unsigned)(8 * hi * x_ld + ca);
    const int xld4 = (int)(x_ld * 4);
    auto load_x = [&](long st) __attribute__((always_inline)) {
      const __amdgpu_buffer_rsrc_t rx = gh_rows(x, st * GB_R, GB_R, m, x_ld);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          xr[q][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, (int)voff_x, (16 * q + j) * xld4, 0));
    };
    f32x16 acc[8];
#pragma unroll
    for (int jt = 0; jt < 8; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[jt][r] = 0.f;
    GhCol sx{0, 0.f, 0.f};
    auto max8 = [&](const float (&v)[8]) __attribute__((always_inline)) {
      float tmx;
      asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(tmx) : "v"(v[0]), "v"(v[1]), "v"(v[2]));
      asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(tmx) : "v"(tmx), "v"(v[3]), "v"(v[4]));
      asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(tmx) : "v"(tmx), "v"(v[5]), "v"(v[6]));
      asm("v_max_f32 %0, %1, |%2|" : "=v"(tmx) : "v"(tmx), "v"(v[7]));
      return tmx;
    };

    auto compute_dw = [&](long st, long st_next, long st_load, int buf) __attribute__((always_inline)) {
      const int* rowk = rowk_base + GB_R * buf;
      const unsigned boff = (unsigned)(buf * GB_BUF);
      // ---- x'' = x 2^(K - kr[row]) in fragment layout, split once per stage ----------------------------------------------
      u32x4 Xh[2], Xl[2];
      {
        float tv[2][8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const u32x4 k0 = *reinterpret_cast<const u32x4*>(rowk + 16 * q + 8 * hi);
          const u32x4 k1 = *reinterpret_cast<const u32x4*>(rowk + 16 * q + 8 * hi + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            tv[q][j] = __builtin_ldexpf(xr[q][j], -(int)k0[j]);
            tv[q][4 + j] = __builtin_ldexpf(xr[q][4 + j], -(int)k1[j]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        load_x(st_next);                                      // the next stage's x: in flight behind this stage's arithmetic
        __builtin_amdgcn_sched_barrier(0);
        const float cm = fmaxf(max8(tv[0]), max8(tv[1]));
#ifndef GB_NO_RESCALE
        if (__builtin_amdgcn_ballot_w64(cm > sx.lim) != 0) {  // wave-uniform, rare after the first stages
          float mxc = fmaxf(cm, __shfl_xor(cm, 32, 64));      // both lane halves hold rows of the same column
          mxc = fmaxf(sx.run, mxc);
          sx.run = mxc;
          int d = 0;
          if (mxc > sx.lim) {                                 // (a column that has only seen zeros keeps (k, lim) = (0, 0))
            const int kn = 13 - __builtin_amdgcn_frexp_expf(mxc);        // the maximum lands in [2^12, 2^13)
            d = kn - sx.k;
            sx.k = kn;
            sx.lim = __builtin_ldexpf(0.99951171875f, 16 - kn);
          }
          int dr[16];                                         // the x column of accumulator row r16: its change of exponent
#pragma unroll
          for (int r16 = 0; r16 < 16; ++r16) dr[r16] = __shfl(d, (r16 & 3) + 8 * (r16 >> 2) + 4 * hi, 64);
          // one accumulator tile at a time (scheduling scope = a tile: left alone, the scheduler reads all 128 accumulator
          // registers out of the accumulator file first -- a register-pressure peak in this COLD block)
          static_for<8>([&](auto jc) __attribute__((always_inline)) {
            constexpr int jt = decltype(jc)::value;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r16 = 0; r16 < 16; ++r16) {
              // in place in the accumulator file (as C++ on the tile's elements the block cost 70-100 spilled registers: the
              // allocator moved whole tiles through VGPRs and scratch memory to merge the rescaled and the untouched values)
              float e = acc[jt][r16], tmp;
              asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_ldexp_f32 %1, %1, %2\n\ts_nop 1\n\tv_accvgpr_write_b32 %0, %1"
                           : "+a"(e), "=&v"(tmp) : "v"(dr[r16]));
              acc[jt][r16] = e;
            }
            __builtin_amdgcn_sched_barrier(0);
          });
#ifdef GB_PIN_ACC
#pragma unroll
          for (int jt = 0; jt < 8; ++jt) asm volatile("" : "+a"(acc[jt]));
#endif
        }
#endif
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            unsigned h, l;
            splith_pair(__builtin_ldexpf(tv[q][2 * e], sx.k), __builtin_ldexpf(tv[q][2 * e + 1], sx.k), h, l);
            Xh[q][e] = h; Xl[q][e] = l;
          }
      }
      GBP(0)
      // ---- 8 groups of 6 MFMAs over two tiles each ---------------------------------------------------------------------------
      const unsigned tb0 = tbase[0] + boff, tb1 = tbase[1] + boff;
      // fragments of tile (q, jt): requested one tile ahead; th is read twice (l.H first, h.H last)
      struct T { u32x4 th, tl; };
      auto read_t = [&](T& f, auto gc) __attribute__((always_inline)) {
        constexpr int gi = decltype(gc)::value, q = gi >> 3, jt = gi & 7;
        constexpr unsigned jo = (unsigned)(jt * 320);
        const u32x2 h0 = gb_ld_tr16(tb0 + (unsigned)((4 * q + 0) * 2560) + jo);
        const u32x2 h1 = gb_ld_tr16(tb1 + (unsigned)((4 * q + 1) * 2560) + jo);
        const u32x2 l0 = gb_ld_tr16(tb0 + GB_PLANE + (unsigned)((4 * q + 0) * 2560) + jo);
        const u32x2 l1 = gb_ld_tr16(tb1 + GB_PLANE + (unsigned)((4 * q + 1) * 2560) + jo);
        f.th = u32x4{h0[0], h0[1], h1[0], h1[1]};
        f.tl = u32x4{l0[0], l0[1], l1[0], l1[1]};
      };
      T F[17];
      read_t(F[0], std::integral_constant<int, 0>{});
      read_t(F[1], std::integral_constant<int, 1>{});
      static_for<8>([&](auto pc) __attribute__((always_inline)) {
        constexpr int g0 = 2 * decltype(pc)::value, g1 = g0 + 1, q = g0 >> 3, j0 = g0 & 7, j1 = g1 & 7;
        // product-major over two independent tiles
        __builtin_amdgcn_sched_barrier(0);
        acc[j0] = mfma_f16(Xl[q], F[g0].th, acc[j0]);
        acc[j1] = mfma_f16(Xl[q], F[g1].th, acc[j1]);
        acc[j0] = mfma_f16(Xh[q], F[g0].tl, acc[j0]);
        acc[j1] = mfma_f16(Xh[q], F[g1].tl, acc[j1]);
        acc[j0] = mfma_f16(Xh[q], F[g0].th, acc[j0]);
        acc[j1] = mfma_f16(Xh[q], F[g1].th, acc[j1]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g0 + 2 < 16) {
          read_t(F[g0 + 2 < 16 ? g0 + 2 : 16], std::integral_constant<int, (g0 + 2 < 16 ? g0 + 2 : 0)>{});
          read_t(F[g0 + 3 < 16 ? g0 + 3 : 16], std::integral_constant<int, (g0 + 3 < 16 ? g0 + 3 : 0)>{});
        }
        // this wave's share of the NEXT stage's staging: a row's two parts behind two consecutive groups
        if constexpr (GB_DW_ROWS > 0) {
          constexpr int pp = decltype(pc)::value, sp = 8 / (GB_DW_ROWS > 0 ? GB_DW_ROWS : 1), jr = pp / sp;
          if constexpr (pp % sp == 0) stage_row_a(st_next, std::integral_constant<int, GB_DX_ROWS + jr>{});
          else if constexpr (pp % sp == 1) stage_row_b(st_load, buf ^ 1, std::integral_constant<int, GB_DX_ROWS + jr>{});
        }
      });
      if constexpr (GB_DW_ROWS > 0) load_bc(st_load);
      GBP(1)
    };

    load_x(t);
    if constexpr (GB_DW_ROWS > 0) {
      static_for<GB_DW_ROWS>([&](auto ic) __attribute__((always_inline)) {
        load_ga_row(t, std::integral_constant<int, GB_DX_ROWS + decltype(ic)::value>{});
      });
      load_bc(t);
      static_for<GB_DW_ROWS>([&](auto ic) __attribute__((always_inline)) {
        stage_row_a(t, std::integral_constant<int, GB_DX_ROWS + decltype(ic)::value>{});
        stage_row_b(t + G, 0, std::integral_constant<int, GB_DX_ROWS + decltype(ic)::value>{});
      });
      load_bc(t + G);
    }
    gh_barrier_lds();
    for (int it = 0; it < niter; ++it) {
      const int buf = it & 1;
      compute_dw(t, t + G, t + 2 * G, buf);
      GBP(2)
      gh_barrier_lds();
      GBP(5)
      t += GB_HOT_STEP;
    }
    // ---- the wave's partial dW rows [128 half + 32 w4 .. + 32) x all columns, unscaled; one partial per PAIR --------------------
    float* pw = part_dw + q0 * (long)ndim * kdim;
#pragma unroll
    for (int r16 = 0; r16 < 16; ++r16) {
      const int rr = (r16 & 3) + 8 * (r16 >> 2) + 4 * hi;
      const int kr = __shfl(sx.k, rr, 64);
      const int row = i0 + rr;
#pragma unroll
      for (int jt = 0; jt < 8; ++jt) {
        const int cj = 32 * jt + li;
        if (row < ndim && cj < kdim) pw[(long)row * kdim + cj] = __builtin_ldexpf(acc[jt][r16], -kr);
      }
    }
  }
  GBP_FLUSH
  if constexpr (DOT) {
    __syncthreads();
    if (tid == 0) {
      const float* red = reinterpret_cast<const float*>(dsm + 2 * (size_t)GB_BUF);
      da.dot_part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
    __syncthreads();
  }
  // dbias: column sums of dpre over the pair's rows (both workgroups hold the same sums; half 0 stores them)
  if (part_db && half == 0) {                                  // uniform
    float* red = reinterpret_cast<float*>(dsm);                // (every wave is behind the loop's last barrier: the image is free)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave * 256 + c4 + e] = bsum[e];
    __syncthreads();
    if (tid < kdim) {
      float sum = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < (GB_DW_ROWS > 0 ? 8 : 4); ++w8) sum += red[w8 * 256 + tid];
      part_db[q0 * kdim + tid] = sum;
    }
  }
}

bool gemmb_ok(const float* g, const float* act_out, const float* x, long m, int din, int dout, long ld, long x_ld, long dx_ld,
              const float* dx) {
  return din > 128 && din <= 256 && dout == 256 && din % 4 == 0 && ld % 4 == 0 && x_ld >= din &&
         dx_ld >= din && (!g || aligned16(g)) && (!act_out || aligned16(act_out)) && m >= (long)kNumCU * 64 &&
         ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx)) & 3u) == 0;
}

// dx = dpre W^T, part_dw[pairs][din][dout] / part_db[pairs][dout] = per-pair partials of x^T dpre / colsum dpre with
// dpre = (grad [+ pooled gradient of the row's graph]) (.) act'(act_out)  (dact = KGCN_ACT_NONE: dpre = grad).
// tabh: the f16 table of W^T (contraction over dout).  Returns the number of partials (> 0) or -1 when the operands do not fit.
int launch_gemmb(const float* grad, const float* act_out, long m, int din, int dout, long ld, const float* x, long x_ld,
                 const void* tabh, float* dx, long dx_ld, float* part_dw, float* part_db, int dact, const float* pooled_grad,
                 int n_nodes, long pooled_ld, hipStream_t s, float* dot_part) {
  // dot_part != nullptr: `dx` is READ ([m, din], row stride dx_ld) and <dpre W^T, dx> goes to dot_part[workgroups] (256 floats)
  const float* base = grad ? grad : act_out;
  if (!base || !gemmb_ok(grad, act_out, x, m, din, dout, ld, x_ld, dx_ld, dx) || !tabh || (!grad && !pooled_grad) ||
      (dact != KGCN_ACT_NONE && !act_out) || (dact == KGCN_ACT_NONE && pooled_grad) || (dot_part && (pooled_grad || dact == KGCN_ACT_NONE)) ||
      (pooled_grad && !(aligned16(pooled_grad) && n_nodes >= 8 && pooled_ld % 4 == 0)))     // (eight rows of a wave: <= 2 graphs)
    return -1;
  GhDact da{};
  da.ydiff = act_out ? act_out - base : 0;
  da.bc = pooled_grad;
  da.bc_ld = pooled_ld;
  da.bc_n = n_nodes > 0 ? n_nodes : 1;
  da.bc_only = grad ? 0 : 1;
  da.c0 = dact == KGCN_ACT_TANH ? 1.f : 0.f;
  da.c1 = dact == KGCN_ACT_SIGMOID ? 1.f : 0.f;
  da.c2 = -1.f;
  da.dot_part = dot_part;
  const int pairs = kNumCU / 2;
  const long stages = (m + GB_R - 1) / GB_R;
  const int niter = (int)((stages + pairs - 1) / pairs);
  const dim3 grid((unsigned)(2 * pairs));
  const u32x4* tab = static_cast<const u32x4*>(tabh);
  const int bc = pooled_grad ? (grad ? 1 : 2) : 0;
  const int dk = dact == KGCN_ACT_NONE ? 0 : (dact == KGCN_ACT_RELU ? 2 : 1);
  auto go = [&](auto dkc, auto bcc, auto dotc) {
    constexpr int DKc = decltype(dkc)::value, BCc = decltype(bcc)::value;
    constexpr bool DOTc = decltype(dotc)::value;
    static thread_local bool attr_set = false;                 // (one flag per instantiation of this lambda)
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemmb_kernel<DKc, BCc, DOTc>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                kLdsBytes);
      attr_set = true;
    }
    hipLaunchKernelGGL((gemmb_kernel<DKc, BCc, DOTc>), grid, dim3(512), gb_lds(DOTc), s, base, m, dout, ld, x, din, x_ld, tab, dx, dx_ld,
                       part_dw, part_db, da, niter);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using F = std::false_type;
  using Tt = std::true_type;
  if (dot_part) { if (dk == 1) go(I1{}, I0{}, Tt{}); else go(I2{}, I0{}, Tt{}); }
  else if (dk == 0) go(I0{}, I0{}, F{});
  else if (dk == 1) { if (bc == 0) go(I1{}, I0{}, F{}); else if (bc == 1) go(I1{}, I1{}, F{}); else go(I1{}, I2{}, F{}); }
  else { if (bc == 0) go(I2{}, I0{}, F{}); else if (bc == 1) go(I2{}, I1{}, F{}); else go(I2{}, I2{}, F{}); }
  if (check_launch("gemmb_kernel")) return -2;
  return pairs;
}

}  // namespace kgcn

#ifdef KGCN_PROBE
extern "C" int kgcn_gb_probe_set(void* buf) {
  long long* p = static_cast<long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(kgcn::gb_probe), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif
