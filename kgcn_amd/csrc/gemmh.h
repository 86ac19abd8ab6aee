// Shared by gemmh.hip (the f16 two-piece GEMMs) and wtable.hip (which writes their weight table): the split, the scale
// exponent and the table layout.  See gemmh.hip for the arithmetic and its error bound.
#pragma once

#include "kgcn_common.h"

namespace kgcn {

typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_f16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}

// two (already scaled) fp32 values -> packed high pieces, packed low pieces
__device__ __forceinline__ void splith_pair(float a, float b, unsigned& h, unsigned& l) {
  const f32x2 v = {a, b};
  const h16x2 hh = __builtin_convertvector(v, h16x2);              // v_cvt_pk_f16_f32 (round to nearest even)
  const f32x2 r = v - __builtin_convertvector(hh, f32x2);          // exact
  const h16x2 ll = __builtin_convertvector(r, h16x2);
  h = __builtin_bit_cast(unsigned, hh);
  l = __builtin_bit_cast(unsigned, ll);
}

// exponent k such that |v| * 2^k lies in [2^14, 2^15) (v = 0, inf, NaN: 15)
__device__ __forceinline__ int scale_exp(float maxabs) {
  return 15 - __builtin_amdgcn_frexp_expf(maxabs);
}

// ---- fragment table of W' (wtable.hip writes it; layout shared through these helpers) --------------------------------
// blocks of 64 x 16 bytes: [(ks * nt32 + nt) * 2 + piece][lane], lane (li, hi) = column 32 nt + li, k = 16 ks + 8 hi + 0..7;
// then the column exponents kc[32 * nt32] (int32).  nt32 = number of 32-column tiles, columns padded to whole 64-blocks;
// k-steps are rounded up to an even number (zero blocks): the consumer's loop is unrolled by two without a tail.
__host__ __device__ inline int gh_nt32(int dout) { return ((dout + 63) / 64) * 2; }
__host__ __device__ inline int gh_kse(int din) { return (((din + 15) / 16) + 1) & ~1; }
__host__ __device__ inline long gh_table_blocks(int din, int dout) { return (long)gh_kse(din) * gh_nt32(dout) * 2; }
__host__ __device__ inline long gh_table_bytes(int din, int dout) {
  return gh_table_blocks(din, dout) * 1024 + (long)gh_nt32(dout) * 32 * 4;
}


// ---- shared by the kernels of gemmh.hip and gemmb.hip ---------------------------------------------------------------------
// max over the wave of four non-negative floats per lane (as bit patterns: unsigned order = float order; a NaN pattern wins,
// which only makes the scale of a row that is non-finite anyway meaningless).  Four rows per asm block: the DPP read of a
// register is three instructions behind its last write (the hazard needs two wait states).
#define KGCN_DPP_MAX4(a, b, c, d, ctrl)                                   \
  "v_max_u32_dpp %0, %0, %0 " ctrl "\n v_max_u32_dpp %1, %1, %1 " ctrl   \
  "\n v_max_u32_dpp %2, %2, %2 " ctrl "\n v_max_u32_dpp %3, %3, %3 " ctrl "\n"
__device__ __forceinline__ void wave_umax4(unsigned& a, unsigned& b, unsigned& c, unsigned& d) {
  asm volatile("s_nop 1\n" KGCN_DPP_MAX4(a, b, c, d, "row_shr:1 row_mask:0xf bank_mask:0xf")
               KGCN_DPP_MAX4(a, b, c, d, "row_shr:2 row_mask:0xf bank_mask:0xf")
               KGCN_DPP_MAX4(a, b, c, d, "row_shr:4 row_mask:0xf bank_mask:0xf")
               KGCN_DPP_MAX4(a, b, c, d, "row_shr:8 row_mask:0xf bank_mask:0xf")
               KGCN_DPP_MAX4(a, b, c, d, "row_bcast:15 row_mask:0xa bank_mask:0xf")
               KGCN_DPP_MAX4(a, b, c, d, "row_bcast:31 row_mask:0xc bank_mask:0xf")
               "s_nop 1\n"
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  a = (unsigned)__builtin_amdgcn_readlane((int)a, 63);
  b = (unsigned)__builtin_amdgcn_readlane((int)b, 63);
  c = (unsigned)__builtin_amdgcn_readlane((int)c, 63);
  d = (unsigned)__builtin_amdgcn_readlane((int)d, 63);
}


// the same for ONE value per lane (a row's maximum where rows are staged one at a time between MFMAs: gemmb.hip, gemmh_fwd_kernel)
__device__ __forceinline__ unsigned wave_umax1(unsigned v) {
  asm volatile("s_nop 1\n"
               "v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n"
               "v_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n"
               "v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n"
               "v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n"
               "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n"
               "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1\n"
               : "+v"(v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// dot_part != nullptr (backward forms): the product is NOT stored; its inner product with `y` (read as an [m, dout] operand of the
// same row stride) is accumulated instead, one partial per workgroup -- d epsilon of a GINAggregate whose input needs no gradient
// (kgcn/layers.py:469: <d out, x>; the d out tensor then never exists in HBM).
struct GhDact { long ydiff, pdiff; float c0, c1, c2; const float* bc; long bc_ld; int bc_n, bc_only; float* dot_part; };


// Row-block buffer descriptors (see gemmh_fwd_kernel): loads outside the descriptor return 0, stores outside it are dropped.
constexpr int kBufFlags = 0x00020000;          // raw dword buffer, gfx9 family (DST_SEL / formats unused by raw accesses)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t gh_rows(const float* base, long row0, long rows, long m, long ld) {
  long n = m - row0;
  n = n < 0 ? 0 : (n < rows ? n : rows);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + (n > 0 ? row0 : 0) * ld), 0, (int)(n * ld * 4), kBufFlags);
}
__device__ __forceinline__ f32x4 gh_ld4(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
}


// The hand-over through LDS needs the wave's LDS operations done (lgkmcnt), NOT its vector-memory operations: __syncthreads()
// also drains vmcnt -- the next tile's rows on their way from HBM and the stores of this one.
__device__ __forceinline__ void gh_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


struct GhCol {           // per lane and fragment: one column of an operand
  int k;                 // scale exponent
  float lim;             // |v| <= lim  <=>  |v| 2^k <= 65504 (the largest f16)
  float run;             // running maximum of |v| over the rows seen so far
};


}  // namespace kgcn
