// COO -> batched CSR on the device (the feed side of the path).
//
// The reference feeds every mini-batch as B x C SparseTensorValues assembled entry by entry in Python
// (kgcn/feed.py:112-126, indices [row, col] per graph and channel); kgcn_amd's host packer
// (BatchedCSR.from_arrays) turns such COO triples into the batched-CSR container with numpy sorts.  A caller whose
// triples already live in HBM (generated, augmented or gathered there) packs them here without leaving the device:
//
//   kgcn_coo_pack_f32   (graph, row, col, val)[nnz] in ANY order -> rowptr [T*M+1], cv [nnz] (col, value bits) with the
//                       entries of a row in feed order (a stable radix sort on graph*M + row: duplicates stay duplicates and
//                       are summed by the kernels in feed order, exactly like the host packer), or -- transposed != 0 --
//                       the container of A^T with the entries of a transposed row in ascending original row
//                       (key (graph*K + col)*M + row), the order BatchedCSR.transpose() produces
//   kgcn_csr_pad4       plain container -> the row-padded layout of the fused GraphConv kernels (row_pad = 4: every row a
//                       positive multiple of 4 entries, padding = (KGCN_PAD_COL, 0); slot table; graph_ptr)
//
// Sort and scan are rocPRIM device primitives (radix_sort_pairs is stable); everything else is a handful of one-pass
// kernels.  Results are bit-identical to the host packer's (tests/test_gpu_parity.py).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "kgcn_common.h"

namespace kgcn {

constexpr int kPackBlock = 256;

static inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

static unsigned key_bits(int64_t max_key) {
  unsigned b = 1;
  while (b < 64 && (max_key >> b) != 0) ++b;
  return b;
}

__global__ __launch_bounds__(kPackBlock) void coo_key_kernel(
    const int* __restrict__ graph, const int* __restrict__ row, const int* __restrict__ col, long nnz, int T, int M, int K,
    int transposed, unsigned long long* __restrict__ keys, int* __restrict__ idx, int* __restrict__ stats) {
  const long e = (long)blockIdx.x * kPackBlock + threadIdx.x;
  if (e >= nnz) return;
  const int g = graph[e], r = row[e], c = col[e];
  unsigned long long key = 0;
  if (g < 0 || g >= T || r < 0 || r >= M || c < 0 || c >= K) {
    atomicAdd(stats + 1, 1);                         // reported to the caller; the entry lands in row 0 of graph 0
  } else {
    key = transposed ? ((unsigned long long)g * K + c) * M + r : (unsigned long long)g * M + r;
  }
  keys[e] = key;
  idx[e] = (int)e;
}

// rowptr[i] = first sorted position whose key is >= the first key of output row i
__global__ __launch_bounds__(kPackBlock) void coo_rowptr_kernel(const unsigned long long* __restrict__ keys, long nnz,
                                                                 long nrows, unsigned long long key_stride,
                                                                 int* __restrict__ rowptr) {
  const long i = (long)blockIdx.x * kPackBlock + threadIdx.x;
  if (i > nrows) return;
  const unsigned long long want = (unsigned long long)i * key_stride;
  long lo = 0, hi = nnz;
  while (lo < hi) {
    const long mid = (lo + hi) >> 1;
    if (keys[mid] < want) lo = mid + 1; else hi = mid;
  }
  rowptr[i] = (int)lo;
}

__global__ __launch_bounds__(kPackBlock) void coo_emit_kernel(
    const int* __restrict__ idx, const int* __restrict__ row, const int* __restrict__ col, const float* __restrict__ val,
    long nnz, int transposed, int2* __restrict__ cv, int* __restrict__ perm) {
  const long p = (long)blockIdx.x * kPackBlock + threadIdx.x;
  if (p >= nnz) return;
  const int src = idx[p];
  const float v = val ? val[src] : 1.f;
  cv[p] = make_int2(transposed ? row[src] : col[src], __float_as_int(v));
  if (perm) perm[p] = src;
}

__global__ __launch_bounds__(kPackBlock) void graph_max_kernel(const int* __restrict__ rowptr, int T, int R,
                                                                int* __restrict__ out_max) {
  const int t = blockIdx.x * kPackBlock + threadIdx.x;
  if (t >= T) return;
  atomicMax(out_max, rowptr[(long)(t + 1) * R] - rowptr[(long)t * R]);
}

// ---- row padding ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kPackBlock) void pad_count_kernel(const int* __restrict__ rowptr, long nrows,
                                                                int* __restrict__ padded) {
  const long i = (long)blockIdx.x * kPackBlock + threadIdx.x;
  if (i > nrows) return;
  int p = 0;
  if (i < nrows) {
    const int cnt = rowptr[i + 1] - rowptr[i];
    p = cnt <= 4 ? 4 : (cnt + 3) & ~3;
  }
  padded[i] = p;                                      // element nrows = 0: the exclusive scan ends with the total
}

__global__ __launch_bounds__(kPackBlock) void pad_fill_kernel(const int* __restrict__ rowptr, const int2* __restrict__ cv,
                                                               const int* __restrict__ rowptr4, long nrows, long capacity,
                                                               int2* __restrict__ cv4, int* __restrict__ stats) {
  const long i = (long)blockIdx.x * kPackBlock + threadIdx.x;
  if (i >= nrows) return;
  const int s = rowptr[i], cnt = rowptr[i + 1] - s;
  const int s4 = rowptr4[i], n4 = rowptr4[i + 1] - s4;
  if ((long)s4 + n4 > capacity) {
    atomicAdd(stats + 2, 1);
    return;
  }
  for (int j = 0; j < n4; ++j) cv4[s4 + j] = j < cnt ? cv[s + j] : make_int2(KGCN_PAD_COL, 0);
}

// one wave per graph (M <= 64 rows): rows ranked by decreasing padded length, ties by row index
__global__ __launch_bounds__(64) void pad_slots_kernel(const int* __restrict__ rowptr4, int T, int M,
                                                        int* __restrict__ slots, int* __restrict__ graph_ptr,
                                                        int* __restrict__ stats) {
  const int t = blockIdx.x, lane = threadIdx.x;
  const long base = (long)t * M;
  const int g0 = rowptr4[base];
  const bool live = lane < M;
  const int start = live ? rowptr4[base + lane] - g0 : 0;
  const int plen = live ? rowptr4[base + lane + 1] - rowptr4[base + lane] : -1;
  int rank = 0;
  for (int j = 0; j < M; ++j) {
    const int pj = __shfl(plen, j, 64);
    rank += (pj > plen || (pj == plen && j < lane)) ? 1 : 0;
  }
  if (live) {
    if (plen > 252 || start >= 65536) atomicAdd(stats + 2, 1);
    slots[base + rank] = (int)((unsigned)start | ((unsigned)plen << 16) | ((unsigned)lane << 24));
  }
  if (lane == 0) {
    graph_ptr[t] = g0;
    const int tot = rowptr4[base + M] - g0;
    atomicMax(stats, tot);
    if (t == T - 1) {
      graph_ptr[T] = rowptr4[base + M];
      stats[1] = rowptr4[base + M];
    }
  }
}

static size_t sort_temp_bytes(long nnz, unsigned bits) {
  size_t b = 0;
  (void)rocprim::radix_sort_pairs(nullptr, b, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (int*)nullptr,
                                  (int*)nullptr, (size_t)nnz, 0u, bits, (hipStream_t)0);
  return b;
}

static size_t scan_temp_bytes(long n) {
  size_t b = 0;
  (void)rocprim::exclusive_scan(nullptr, b, (int*)nullptr, (int*)nullptr, 0, (size_t)n, rocprim::plus<int>(),
                                (hipStream_t)0);
  return b;
}

}  // namespace kgcn

using namespace kgcn;

extern "C" int64_t kgcn_coo_pack_workspace_bytes(int64_t nnz, int32_t num_graphs, int32_t rows, int32_t cols) {
  if (nnz <= 0) return 256;
  const int64_t mx = (int64_t)(num_graphs > 0 ? num_graphs : 1) * (rows > 0 ? rows : 1) * (cols > 0 ? cols : 1);
  return (int64_t)(2 * align256((size_t)nnz * 8) + 2 * align256((size_t)nnz * 4) + align256(sort_temp_bytes(nnz, key_bits(mx))) +
                   256);
}

extern "C" int kgcn_coo_pack_f32(const int32_t* graph, const int32_t* row, const int32_t* col, const float* val,
                                 int64_t nnz, int32_t num_graphs, int32_t rows, int32_t cols, int32_t transposed,
                                 int32_t* rowptr_out, void* cv_out, int32_t* perm_out, int32_t* stats_out,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
  if (nnz < 0 || num_graphs < 0 || rows < 0 || cols < 0)
    return fail("kgcn_coo_pack_f32: negative size (nnz=%lld T=%d M=%d K=%d)", (long long)nnz, num_graphs, rows, cols);
  const int R = transposed ? cols : rows;                 // rows of the output container
  const int64_t nrows = (int64_t)num_graphs * R;
  if (nrows + 1 >= (int64_t)INT32_MAX || nnz >= (int64_t)INT32_MAX)
    return fail("kgcn_coo_pack_f32: batch too large for int32 offsets");
  if (!rowptr_out || !stats_out) return fail("kgcn_coo_pack_f32: rowptr_out / stats_out is NULL");
  hipStream_t s = as_stream(stream);
  (void)hipMemsetAsync(stats_out, 0, 2 * sizeof(int32_t), s);
  if (nnz == 0) {
    (void)hipMemsetAsync(rowptr_out, 0, (size_t)(nrows + 1) * 4, s);
    return 0;
  }
  if (!graph || !row || !col || !cv_out) return fail("kgcn_coo_pack_f32: NULL operand");
  const int64_t need = kgcn_coo_pack_workspace_bytes(nnz, num_graphs, rows, cols);
  if (!workspace || workspace_bytes < need)
    return fail("kgcn_coo_pack_f32: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
  unsigned char* w = static_cast<unsigned char*>(workspace);
  w = reinterpret_cast<unsigned char*>(align256(reinterpret_cast<size_t>(w)));
  unsigned long long* keys_in = reinterpret_cast<unsigned long long*>(w); w += align256((size_t)nnz * 8);
  unsigned long long* keys_out = reinterpret_cast<unsigned long long*>(w); w += align256((size_t)nnz * 8);
  int* idx_in = reinterpret_cast<int*>(w); w += align256((size_t)nnz * 4);
  int* idx_out = reinterpret_cast<int*>(w); w += align256((size_t)nnz * 4);
  const int64_t max_key = transposed ? (int64_t)num_graphs * cols * rows : (int64_t)num_graphs * rows;
  const unsigned bits = key_bits(max_key);
  size_t temp = sort_temp_bytes(nnz, bits);
  const unsigned nb = (unsigned)((nnz + kPackBlock - 1) / kPackBlock);
  hipLaunchKernelGGL(coo_key_kernel, dim3(nb), dim3(kPackBlock), 0, s, graph, row, col, (long)nnz, num_graphs, rows, cols,
                     transposed, keys_in, idx_in, stats_out);
  if (rocprim::radix_sort_pairs(w, temp, keys_in, keys_out, idx_in, idx_out, (size_t)nnz, 0u, bits, s) != hipSuccess)
    return fail("kgcn_coo_pack_f32: rocprim::radix_sort_pairs failed");
  hipLaunchKernelGGL(coo_rowptr_kernel, dim3((unsigned)((nrows + 1 + kPackBlock - 1) / kPackBlock)), dim3(kPackBlock), 0, s,
                     keys_out, (long)nnz, (long)nrows, (unsigned long long)(transposed ? rows : 1), rowptr_out);
  hipLaunchKernelGGL(coo_emit_kernel, dim3(nb), dim3(kPackBlock), 0, s, idx_out, row, col, val, (long)nnz, transposed,
                     static_cast<int2*>(cv_out), perm_out);
  if (num_graphs > 0 && R > 0)
    hipLaunchKernelGGL(graph_max_kernel, dim3((unsigned)((num_graphs + kPackBlock - 1) / kPackBlock)), dim3(kPackBlock), 0,
                       s, rowptr_out, num_graphs, R, stats_out);
  return check_launch("kgcn_coo_pack_f32");
}

extern "C" int64_t kgcn_csr_pad4_workspace_bytes(int32_t num_graphs, int32_t rows) {
  const int64_t n = (int64_t)(num_graphs > 0 ? num_graphs : 0) * (rows > 0 ? rows : 0) + 1;
  return (int64_t)(align256((size_t)n * 4) + align256(scan_temp_bytes(n)) + 256);
}

extern "C" int kgcn_csr_pad4(const kgcn_csr_batch* a, int32_t* rowptr4_out, void* cv4_out, int64_t cv4_capacity,
                             int32_t* slots_out, int32_t* graph_ptr_out, int32_t* stats_out, void* workspace,
                             int64_t workspace_bytes, void* stream) {
  if (int rc = validate_csr(a, "kgcn_csr_pad4")) return rc;
  if (a->rows != a->cols) return fail("kgcn_csr_pad4: adjacency must be square");
  if (a->rows > KGCN_PAD_COL)
    return fail("kgcn_csr_pad4: row padding is defined for graphs of at most %d nodes, got %d", KGCN_PAD_COL, a->rows);
  if (!rowptr4_out || !graph_ptr_out || !stats_out || (a->num_graphs > 0 && a->rows > 0 && (!slots_out || !cv4_out)))
    return fail("kgcn_csr_pad4: NULL output");
  hipStream_t s = as_stream(stream);
  (void)hipMemsetAsync(stats_out, 0, 3 * sizeof(int32_t), s);
  const int T = a->num_graphs, M = a->rows;
  const long nrows = (long)T * M;
  if (nrows == 0) {
    (void)hipMemsetAsync(rowptr4_out, 0, 4, s);
    (void)hipMemsetAsync(graph_ptr_out, 0, (size_t)(T + 1) * 4, s);
    return 0;
  }
  const int64_t need = kgcn_csr_pad4_workspace_bytes(T, M);
  if (!workspace || workspace_bytes < need)
    return fail("kgcn_csr_pad4: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
  unsigned char* w = reinterpret_cast<unsigned char*>(align256(reinterpret_cast<size_t>(workspace)));
  int* padded = reinterpret_cast<int*>(w); w += align256((size_t)(nrows + 1) * 4);
  size_t temp = scan_temp_bytes(nrows + 1);
  const unsigned nb = (unsigned)((nrows + 1 + kPackBlock - 1) / kPackBlock);
  hipLaunchKernelGGL(pad_count_kernel, dim3(nb), dim3(kPackBlock), 0, s, a->rowptr, nrows, padded);
  if (rocprim::exclusive_scan(w, temp, padded, rowptr4_out, 0, (size_t)(nrows + 1), rocprim::plus<int>(), s) != hipSuccess)
    return fail("kgcn_csr_pad4: rocprim::exclusive_scan failed");
  hipLaunchKernelGGL(pad_fill_kernel, dim3(nb), dim3(kPackBlock), 0, s, a->rowptr, reinterpret_cast<const int2*>(a->cv),
                     rowptr4_out, nrows, (long)cv4_capacity, static_cast<int2*>(cv4_out), stats_out);
  hipLaunchKernelGGL(pad_slots_kernel, dim3((unsigned)T), dim3(64), 0, s, rowptr4_out, T, M, slots_out, graph_ptr_out,
                     stats_out);
  return check_launch("kgcn_csr_pad4");
}
