// Internal helpers shared by the HIP translation units of libkgcn_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "../../include/kgcn_hip.h"

namespace kgcn {

// thread-local error text behind kgcn_last_error()
char* error_buffer();
int fail(const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("%s: launch failed: %s", what, hipGetErrorString(e));
  return 0;
}

// Development routing knobs (KGCN_DENSE_ROUTE, KGCN_GEMM3_MW) exist only in a -DKGCN_DEV_KNOBS build (make DEV_KNOBS=1,
// what tools/variant_bench.py / tools/gemm_route_bench.py use): the shipped library never reads the environment, so a
// stray variable cannot change kernel routing (and with it summation order) behind the parity tests' back.
#ifdef KGCN_DEV_KNOBS
inline const char* dev_knob(const char* name) { return getenv(name); }
#else
inline const char* dev_knob(const char*) { return nullptr; }
#endif

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int validate_csr(const kgcn_csr_batch* a, const char* who, bool allow_row_pad = false) {
  if (!a) return fail("%s: csr descriptor is NULL", who);
  if (a->row_pad != 0 && !(allow_row_pad && a->row_pad == 4))
    return fail("%s: row_pad=%d batches are only accepted by the fused GraphConv kernels", who,
                a->row_pad);
  if (a->num_graphs < 0 || a->rows < 0 || a->cols < 0 || a->nnz < 0)
    return fail("%s: negative size in csr descriptor (T=%d M=%d K=%d nnz=%lld)", who,
                a->num_graphs, a->rows, a->cols, (long long)a->nnz);
  if (a->num_graphs > 0 && a->rows > 0 && !a->rowptr) return fail("%s: rowptr is NULL", who);
  if (a->nnz > 0 && !a->cv) return fail("%s: cv is NULL with nnz=%lld", who, (long long)a->nnz);
  if ((int64_t)a->num_graphs * a->rows >= (int64_t)INT32_MAX)
    return fail("%s: T*M exceeds int32 row indexing", who);
  if (a->nnz >= (int64_t)INT32_MAX) return fail("%s: nnz exceeds int32 offsets", who);
  return 0;
}

constexpr int kWave = 64;        // CDNA wavefront
constexpr int kNumCU = 256;      // MI355X
constexpr int kLdsBytes = 160 * 1024;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- shared device helpers ---------------------------------------------------------------------------
template <typename Fn, int... I>
__device__ __forceinline__ void static_for_impl(Fn&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): straight-line code with compile-time
// indices (register arrays indexed inside `#pragma unroll` loops ended up in scratch memory)
template <int N, typename Fn>
__device__ __forceinline__ void static_for(Fn&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Values that are only consumed much later (fragments of the next k-step / phase, running sums): without a use at the
// place of their definition, machine sinking moves the whole computation across intervening blocks into ONE clump in
// front of the consumer.  An empty volatile asm with the values as in/out operands is such a use.
#define KGCN_PIN3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))
#define KGCN_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))

// ---- activations fused into producer epilogues (KGCN_ACT_* of include/kgcn_hip.h) ----------------------------
// reciprocals through v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division sequence: an activation epilogue is
// 1-2 instructions per element next to the exponential, not a dozen (the row-chunk SpMM is VALU-bound otherwise)
__device__ __forceinline__ float act_fwd(float v, int act) {
  if (act == KGCN_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.0f + __expf(-v));
  if (act == KGCN_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == KGCN_ACT_TANH) {
    const float e = __expf(-2.0f * fabsf(v));
    const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    return v < 0.f ? -t : t;
  }
  return v;
}
// derivative expressed in the activation OUTPUT a (what a backward pass has at hand)
__device__ __forceinline__ float act_dout(float a, int act) {
  if (act == KGCN_ACT_SIGMOID) return a * (1.0f - a);
  if (act == KGCN_ACT_RELU) return a > 0.f ? 1.0f : 0.f;
  if (act == KGCN_ACT_TANH) return 1.0f - a * a;
  return 1.0f;
}

// ---- fp32 contraction on the bf16 matrix pipe: exact 3-way split ------------------------------------
// gfx950 runs v_mfma_f32_32x32x2_f32 at the VALU rate AND on the VALU datapath (nothing overlaps it:
// tools/mfma_shadow.hip), while v_mfma_f32_32x32x16_bf16 is 16x faster per flop and runs beside VALU
// work (tools/mfma_shadow_bf16.hip).  An fp32 value is the EXACT sum of three bf16 values obtained by
// truncation -- v = p1 + p2 + p3, 24 significand bits = 8 + 8 + 8 -- and bf16 x bf16 products are exact
// in the fp32 accumulator, so  a*b = sum_{i+j<=4} a_i b_j + O(2^-23 |ab|): six bf16 MFMAs replace eight
// f32 MFMAs (K = 16 vs 2) at fp32 accuracy (the dropped terms a2b3 + a3b2 + a3b3 are below one fp32 ulp
// of the product).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Frag3 { u32x4 p1, p2, p3; };   // 8 k-values of one MFMA operand row/column, three bf16 pieces

__device__ __forceinline__ void split_pair(float v0, float v1, unsigned& q1, unsigned& q2, unsigned& q3) {
  const unsigned m = 0xffff0000u;
  const float h0 = __uint_as_float(__float_as_uint(v0) & m), h1 = __uint_as_float(__float_as_uint(v1) & m);
  const float r0 = v0 - h0, r1 = v1 - h1;                     // exact: the low 16 significand bits
  const float g0 = __uint_as_float(__float_as_uint(r0) & m), g1 = __uint_as_float(__float_as_uint(r1) & m);
  const float s0 = r0 - g0, s1 = r1 - g1;                     // exact, <= 8 significant bits: a bf16 value
  q1 = __builtin_amdgcn_perm(__float_as_uint(v1), __float_as_uint(v0), 0x07060302u);   // high halves
  q2 = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
  q3 = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}
__device__ __forceinline__ void split8(const float* v, Frag3& f) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned q1, q2, q3;
    split_pair(v[2 * j], v[2 * j + 1], q1, q2, q3);
    f.p1[j] = q1; f.p2[j] = q2; f.p3[j] = q3;
  }
}
__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c,
                                                 0, 0, 0);
}
// the six products of one (A fragment, B fragment) pair, smallest terms first
#define KGCN_SPLIT_PRODUCTS(F) F(p3, p1) F(p2, p2) F(p1, p3) F(p2, p1) F(p1, p2) F(p1, p1)

}  // namespace kgcn
