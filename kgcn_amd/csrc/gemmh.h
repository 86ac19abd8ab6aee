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

}  // namespace kgcn
