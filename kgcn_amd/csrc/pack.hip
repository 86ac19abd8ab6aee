// Mini-batch assembly on the device.
//
// The reference assembles every mini-batch entry by entry in Python and re-feeds B x C
// SparseTensorValues per step (kgcn/feed.py:112-133, core.py:267-269).  Here the adjacency of the whole
// dataset stays resident in HBM as ONE batched-CSR container per channel; a mini-batch is the list of
// selected graph indices (-1 = the empty dummy graph that pads a short batch, feed.py:123-126) and the
// batch container is produced by a segmented copy:
//   gather_count_kernel      per selected graph: entry count, block-local exclusive scan, block totals
//   gather_scan_blocks_kernel one workgroup scans the block totals (<= 4096 of them per pass)
//   gather_copy_kernel       one wave per graph: re-based rowptr slice, cv slice (8-byte entries,
//                            coalesced), slot table slice; writes graph_ptr
// A row-padded source (row_pad = 4) yields a row-padded batch; its dummy graphs are synthesised (every row
// = 4 padding entries), exactly what BatchedCSR.padded4() builds for an empty graph.
#include "kgcn_common.h"

namespace kgcn {

constexpr int kScanBlock = 256;

__device__ __forceinline__ int graph_entries(const int* __restrict__ src_rowptr, int g, int M, int dummy_cnt) {
  return g < 0 ? dummy_cnt : src_rowptr[(long)(g + 1) * M] - src_rowptr[(long)g * M];
}

// exclusive scan of one value per thread over a 256-thread workgroup; returns the exclusive prefix,
// *total = workgroup sum.  Wave-level shuffles + one LDS hop over the 4 wave totals.
__device__ __forceinline__ int block_exclusive_scan(int v, int* total) {
  __shared__ int wave_tot[kScanBlock / kWave];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  int inc = v;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    int up = __shfl_up(inc, o, kWave);
    if (lane >= o) inc += up;
  }
  if (lane == kWave - 1) wave_tot[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kScanBlock / kWave; ++w) {
    const int t = wave_tot[w];
    if (w < wave) base += t;
    tot += t;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ __launch_bounds__(kScanBlock) void gather_count_kernel(
    const int* __restrict__ src_rowptr, const int* __restrict__ sel, int T, int M, int dummy_cnt,
    int* __restrict__ graph_ptr, int* __restrict__ block_sums) {
  const int t = blockIdx.x * kScanBlock + threadIdx.x;
  const int cnt = t < T ? graph_entries(src_rowptr, sel[t], M, dummy_cnt) : 0;
  int tot;
  const int ex = block_exclusive_scan(cnt, &tot);
  if (t < T) graph_ptr[t] = ex;                      // block-local prefix, finalised by the copy kernel
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kScanBlock) void gather_scan_blocks_kernel(int* __restrict__ block_sums, int nb,
                                                                         int* __restrict__ total_out) {
  int carry = 0;
  for (int b0 = 0; b0 < nb; b0 += kScanBlock) {
    const int i = b0 + threadIdx.x;
    const int v = i < nb ? block_sums[i] : 0;
    int tot;
    const int ex = block_exclusive_scan(v, &tot);
    if (i < nb) block_sums[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(256) void gather_copy_kernel(
    const int* __restrict__ src_rowptr, const int2* __restrict__ src_cv, const int* __restrict__ src_slots,
    const int* __restrict__ sel, int T, int M, int row_pad, const int* __restrict__ block_sums,
    int* __restrict__ graph_ptr, int* __restrict__ dst_rowptr, int2* __restrict__ dst_cv,
    int* __restrict__ dst_slots) {
  const int lane = threadIdx.x & (kWave - 1);
  const long wave0 = ((long)blockIdx.x * 256 + threadIdx.x) / kWave;
  const long nwaves = (long)gridDim.x * 256 / kWave;
  for (long t = wave0; t < T; t += nwaves) {
    const int g = sel[t];
    const int out_base = graph_ptr[t] + block_sums[t / kScanBlock];
    int* rp_out = dst_rowptr + t * M;
    int cnt;
    if (g >= 0) {
      const int* rp_in = src_rowptr + (long)g * M;
      const int src_base = rp_in[0];
      cnt = rp_in[M] - src_base;
      for (int r = lane; r < M; r += kWave) rp_out[r] = rp_in[r] - src_base + out_base;
      const int2* in = src_cv + src_base;
      int2* out = dst_cv + out_base;
      for (int i = lane; i < cnt; i += kWave) out[i] = in[i];
      if (row_pad) {
        const int* sl = src_slots + (long)g * M;
        for (int r = lane; r < M; r += kWave) dst_slots[t * M + r] = sl[r];
      }
    } else if (row_pad) {                               // dummy graph of a row-padded batch: M rows x 4 pads
      cnt = 4 * M;
      for (int r = lane; r < M; r += kWave) {
        rp_out[r] = out_base + 4 * r;
        dst_slots[t * M + r] = (4 * r) | (4 << 16) | (r << 24);
      }
      int2* out = dst_cv + out_base;
      for (int i = lane; i < cnt; i += kWave) out[i] = make_int2(KGCN_PAD_COL, 0);
    } else {
      cnt = 0;
      for (int r = lane; r < M; r += kWave) rp_out[r] = out_base;
    }
    if (lane == 0) {
      graph_ptr[t] = out_base;                        // no other wave reads entry t
      if (t == T - 1) dst_rowptr[(long)T * M] = out_base + cnt;
    }
  }
}

}  // namespace kgcn

using namespace kgcn;

extern "C" int64_t kgcn_csr_gather_workspace_bytes(int32_t num_sel) {
  if (num_sel <= 0) return 0;
  return (int64_t)((num_sel + kScanBlock - 1) / kScanBlock) * 4;
}

extern "C" int kgcn_csr_gather_graphs(const kgcn_csr_batch* src, const int32_t* sel, int32_t num_sel,
                                      int32_t* dst_rowptr, int32_t* dst_cv, int64_t dst_cv_capacity,
                                      int32_t* dst_slots, int32_t* dst_graph_ptr, void* workspace,
                                      int64_t workspace_bytes, void* stream) {
  if (int rc = validate_csr(src, "kgcn_csr_gather_graphs", /*allow_row_pad=*/true)) return rc;
  if (num_sel < 0) return fail("kgcn_csr_gather_graphs: negative num_sel");
  if (!dst_rowptr || !dst_graph_ptr) return fail("kgcn_csr_gather_graphs: dst_rowptr / dst_graph_ptr is NULL");
  if (src->row_pad == 4 && (!src->slots || (num_sel > 0 && src->rows > 0 && !dst_slots)))
    return fail("kgcn_csr_gather_graphs: a row-padded source needs slots and dst_slots");
  if ((int64_t)num_sel * src->rows >= (int64_t)INT32_MAX)
    return fail("kgcn_csr_gather_graphs: T*M exceeds int32 row indexing");
  hipStream_t s = as_stream(stream);
  if (num_sel == 0) {                                  // empty batch: rowptr = [0], graph_ptr = [0]
    hipError_t e = hipMemsetAsync(dst_rowptr, 0, 4, s);
    if (e == hipSuccess) e = hipMemsetAsync(dst_graph_ptr, 0, 4, s);
    return e == hipSuccess ? 0 : fail("kgcn_csr_gather_graphs: memset failed: %s", hipGetErrorString(e));
  }
  if (!sel) return fail("kgcn_csr_gather_graphs: sel is NULL");
  const int64_t need = kgcn_csr_gather_workspace_bytes(num_sel);
  if (!workspace || workspace_bytes < need)
    return fail("kgcn_csr_gather_graphs: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
  // dst_cv holds dst_cv_capacity entries: the exact total (the caller knows every graph's entry count) or
  // the worst case num_sel * max(max_nnz_per_graph, 4*rows for dummy graphs of a row-padded batch)
  if (dst_cv_capacity < 0 || (dst_cv_capacity > 0 && !dst_cv))
    return fail("kgcn_csr_gather_graphs: dst_cv is NULL with capacity %lld", (long long)dst_cv_capacity);
  const int nb = (num_sel + kScanBlock - 1) / kScanBlock;
  int* block_sums = static_cast<int*>(workspace);
  const int dummy_cnt = src->row_pad ? 4 * src->rows : 0;
  hipLaunchKernelGGL(gather_count_kernel, dim3(nb), dim3(kScanBlock), 0, s, src->rowptr, sel, num_sel, src->rows,
                     dummy_cnt, dst_graph_ptr, block_sums);
  if (int rc = check_launch("gather_count_kernel")) return rc;
  hipLaunchKernelGGL(gather_scan_blocks_kernel, dim3(1), dim3(kScanBlock), 0, s, block_sums, nb,
                     dst_graph_ptr + num_sel);
  if (int rc = check_launch("gather_scan_blocks_kernel")) return rc;
  long blocks = ((long)num_sel + 3) / 4;               // 4 waves per workgroup, one wave per graph
  if (blocks > (long)kNumCU * 32) blocks = (long)kNumCU * 32;
  hipLaunchKernelGGL(gather_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src->rowptr,
                     reinterpret_cast<const int2*>(src->cv), src->slots, sel, num_sel, src->rows, src->row_pad,
                     block_sums, dst_graph_ptr, dst_rowptr, reinterpret_cast<int2*>(dst_cv), dst_slots);
  return check_launch("gather_copy_kernel");
}
