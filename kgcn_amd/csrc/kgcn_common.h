// Internal helpers shared by the HIP translation units of libkgcn_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

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
typedef float f32x16 __attribute__((ext_vector_type(16)));

}  // namespace kgcn
