/* A non-Python consumer of the C ABI (every "through the C-ABI" test was ctypes from the process that also
 * defines the ctypes mirror of the structs, which cannot see layout drift).  Plain C against include/kgcn_hip.h, built by
 * __graft_entry__.build() / tests with gcc, linked to libkgcn_hip.so and libamdhip64 only -- the binding a maintainer of
 * the reference would write for kgcn/bspmm_call.py:9-19 starts exactly like this.
 *
 *   abi_consumer layout   no GPU needed: sizeof / offsetof of kgcn_csr_batch as THIS compiler sees the header, the
 *                         library's kgcn_csr_batch_size() / kgcn_abi_version(), and the error convention
 *                         (a NULL descriptor -> non-zero status + kgcn_last_error() text)
 *   abi_consumer bspmm    on the GPU box: one kgcn_bspmm_f32 call on a 2-graph batch built here by hand, checked against
 *                         the products computed in C (exact: small integers), then a refused call (row_pad batch)
 * Prints one "key value" line per fact and "OK" last; exit status 0 only if every check held. */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kgcn_hip.h"

/* the four HIP runtime entry points this program needs (it is compiled with gcc, without the HIP headers) */
extern int hipMalloc(void** p, size_t n);
extern int hipFree(void* p);
extern int hipMemcpy(void* dst, const void* src, size_t n, int kind);
extern int hipDeviceSynchronize(void);
enum { H2D = 1, D2H = 2 };

static int fails = 0;
#define CHECK(cond, what) do { if (!(cond)) { printf("FAIL %s\n", what); ++fails; } } while (0)

static int layout(void) {
  printf("header_abi_version %d\n", KGCN_HIP_ABI_VERSION);
  printf("library_abi_version %d\n", kgcn_abi_version());
  printf("sizeof_csr_batch %zu\n", sizeof(kgcn_csr_batch));
  printf("library_csr_batch_size %lld\n", (long long)kgcn_csr_batch_size());
  printf("offsetof_nnz %zu\n", offsetof(kgcn_csr_batch, nnz));
  printf("offsetof_rowptr %zu\n", offsetof(kgcn_csr_batch, rowptr));
  printf("offsetof_cv %zu\n", offsetof(kgcn_csr_batch, cv));
  printf("offsetof_slots %zu\n", offsetof(kgcn_csr_batch, slots));
  printf("offsetof_graph_ptr %zu\n", offsetof(kgcn_csr_batch, graph_ptr));
  printf("offsetof_block_ptr %zu\n", offsetof(kgcn_csr_batch, block_ptr));
  printf("offsetof_num_blocks %zu\n", offsetof(kgcn_csr_batch, num_blocks));
  printf("offsetof_block_rows_max %zu\n", offsetof(kgcn_csr_batch, block_rows_max));
  printf("sizeof_wtable_job %zu\n", sizeof(kgcn_wtable_job));
  printf("offsetof_wtable_job_table %zu\n", offsetof(kgcn_wtable_job, table));
  printf("offsetof_wtable_job_extra_row %zu\n", offsetof(kgcn_wtable_job, extra_row));
  CHECK(kgcn_abi_version() == KGCN_HIP_ABI_VERSION, "library and header ABI versions differ");
  CHECK(kgcn_csr_batch_size() == (int64_t)sizeof(kgcn_csr_batch), "kgcn_csr_batch size differs between header and library");
  CHECK(strcmp(kgcn_build_arch(), "gfx950") == 0, "build arch");
  /* error convention: status + thread-local text, nothing launched */
  int rc = kgcn_bspmm_f32(NULL, NULL, 0, 0, 4, NULL, 0, 0, 0.0f, NULL);
  CHECK(rc != 0, "NULL descriptor accepted");
  CHECK(strstr(kgcn_last_error(), "NULL") != NULL, "kgcn_last_error text");
  printf("last_error %s\n", kgcn_last_error());
  return fails;
}

static int bspmm(void) {
  /* two graphs of 3 nodes, d = 4: graph 0 = path 0-1-2 with self loops on 0 (duplicate entry on row 0: accumulates like
   * tf.sparse_tensor_dense_matmul), graph 1 = one entry (2,0) with value 3 and two empty rows */
  enum { T = 2, M = 3, D = 4 };
  const int32_t rowptr[T * M + 1] = {0, 3, 5, 6, 6, 6, 7};
  const int32_t col[7] = {0, 0, 1, 0, 2, 1, 0};
  const float val[7] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 3.f};
  int32_t cv[14];
  for (int e = 0; e < 7; ++e) { cv[2 * e] = col[e]; memcpy(&cv[2 * e + 1], &val[e], 4); }
  float rhs[T * M * D], want[T * M * D], got[T * M * D];
  for (int i = 0; i < T * M * D; ++i) { rhs[i] = (float)(i % 7) - 2.f; want[i] = 0.f; got[i] = -99.f; }
  for (int t = 0; t < T; ++t)
    for (int r = 0; r < M; ++r)
      for (int e = rowptr[t * M + r]; e < rowptr[t * M + r + 1]; ++e)
        for (int k = 0; k < D; ++k) want[(t * M + r) * D + k] += val[e] * rhs[(t * M + col[e]) * D + k];
  void *d_rowptr = 0, *d_cv = 0, *d_rhs = 0, *d_out = 0;
  CHECK(hipMalloc(&d_rowptr, sizeof rowptr) == 0 && hipMalloc(&d_cv, sizeof cv) == 0 && hipMalloc(&d_rhs, sizeof rhs) == 0 &&
        hipMalloc(&d_out, sizeof got) == 0, "hipMalloc");
  if (fails) return fails;
  hipMemcpy(d_rowptr, rowptr, sizeof rowptr, H2D);
  hipMemcpy(d_cv, cv, sizeof cv, H2D);
  hipMemcpy(d_rhs, rhs, sizeof rhs, H2D);
  hipMemcpy(d_out, got, sizeof got, H2D);
  kgcn_csr_batch a;
  memset(&a, 0, sizeof a);
  a.num_graphs = T; a.rows = M; a.cols = M; a.max_nnz_per_graph = 6; a.nnz = 7;
  a.rowptr = (const int32_t*)d_rowptr; a.cv = (const int32_t*)d_cv;
  int rc = kgcn_bspmm_f32(&a, (const float*)d_rhs, D, M * D, D, (float*)d_out, D, M * D, 0.0f, NULL);
  CHECK(rc == 0, "kgcn_bspmm_f32 status");
  if (rc != 0) printf("last_error %s\n", kgcn_last_error());
  CHECK(hipDeviceSynchronize() == 0, "hipDeviceSynchronize");
  hipMemcpy(got, d_out, sizeof got, D2H);
  int bad = 0;
  for (int i = 0; i < T * M * D; ++i) bad += got[i] != want[i];      /* small integers: exact */
  printf("bspmm_mismatches %d of %d\n", bad, T * M * D);
  CHECK(bad == 0, "kgcn_bspmm_f32 result");
  a.row_pad = 4;                                                      /* the plain kernels refuse the row-padded layout */
  rc = kgcn_bspmm_f32(&a, (const float*)d_rhs, D, M * D, D, (float*)d_out, D, M * D, 0.0f, NULL);
  CHECK(rc != 0 && strstr(kgcn_last_error(), "row_pad") != NULL, "row_pad batch refused with a message");
  hipFree(d_rowptr); hipFree(d_cv); hipFree(d_rhs); hipFree(d_out);
  return fails;
}

int main(int argc, char** argv) {
  const char* what = argc > 1 ? argv[1] : "layout";
  int f = strcmp(what, "bspmm") == 0 ? (layout(), bspmm()) : layout();
  if (f == 0) printf("OK\n");
  return f == 0 ? 0 : 1;
}
