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


// ---- one mini-batch, all of its tensors, two launches ---------------------------------------------------------------------
// kgcn_csr_gather_graphs is three launches per container, and a step that assembles its batch needs A, A^T (or their row-
// padded copies) plus the feature / label / mask / size rows of the selected graphs: 12+ launches of a few microseconds of
// work each -- at the reference's batch sizes that is as long as the model's own kernels.  Here ONE count launch covers every
// container (grid.y = container) and ONE copy launch covers every container and every per-graph table (grid.y = container or
// table); the scan over the <= T / 256 block totals is redone by each wave of the copy kernel (a strided sum + a wave
// reduction) instead of a launch of its own.
struct AssemblePlan {
  int num_csr, num_tables, nb;
  const int* src_rowptr[KGCN_ASSEMBLE_MAX_CSR];
  const int2* src_cv[KGCN_ASSEMBLE_MAX_CSR];
  const int* src_slots[KGCN_ASSEMBLE_MAX_CSR];
  int M[KGCN_ASSEMBLE_MAX_CSR], row_pad[KGCN_ASSEMBLE_MAX_CSR];
  int* graph_ptr[KGCN_ASSEMBLE_MAX_CSR];
  int* dst_rowptr[KGCN_ASSEMBLE_MAX_CSR];
  int2* dst_cv[KGCN_ASSEMBLE_MAX_CSR];
  int* dst_slots[KGCN_ASSEMBLE_MAX_CSR];
  const float* table[KGCN_ASSEMBLE_MAX_TABLES];
  float* table_out[KGCN_ASSEMBLE_MAX_TABLES];
  long row_floats[KGCN_ASSEMBLE_MAX_TABLES];
};

__global__ __launch_bounds__(kScanBlock) void assemble_count_kernel(AssemblePlan p, const int* __restrict__ sel, int T,
                                                                    int* __restrict__ block_sums) {
  const int c = blockIdx.y;
  const int t = blockIdx.x * kScanBlock + threadIdx.x;
  const int cnt = t < T ? graph_entries(p.src_rowptr[c], sel[t], p.M[c], p.row_pad[c] ? 4 * p.M[c] : 0) : 0;
  int tot;
  const int ex = block_exclusive_scan(cnt, &tot);
  if (t < T) p.graph_ptr[c][t] = ex;                 // block-local prefix, finalised by the copy kernel
  if (threadIdx.x == 0) block_sums[c * p.nb + blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void assemble_copy_kernel(AssemblePlan p, const int* __restrict__ sel, int T,
                                                            const int* __restrict__ block_sums) {
  const int lane = threadIdx.x & (kWave - 1);
  const long wave0 = ((long)blockIdx.x * 256 + threadIdx.x) / kWave;
  const long nwaves = (long)gridDim.x * 256 / kWave;
  if ((int)blockIdx.y >= p.num_csr) {
    // per-graph table: row sel[t] (a row of zeros for a dummy graph)
    const int k = blockIdx.y - p.num_csr;
    const long rf = p.row_floats[k];
    const float* __restrict__ src = p.table[k];
    float* __restrict__ dst = p.table_out[k];
    for (long t = wave0; t < T; t += nwaves) {
      const int g = sel[t];
      const float* in = src + (long)(g < 0 ? 0 : g) * rf;
      float* out = dst + t * rf;
      for (long i = lane; i < rf; i += kWave) out[i] = g < 0 ? 0.f : in[i];
    }
    return;
  }
  const int c = blockIdx.y, M = p.M[c], row_pad = p.row_pad[c];
  const int* __restrict__ src_rowptr = p.src_rowptr[c];
  const int* bs = block_sums + c * p.nb;
  int* graph_ptr = p.graph_ptr[c];
  int* dst_rowptr = p.dst_rowptr[c];
  int* dst_slots = p.dst_slots[c];
  for (long t = wave0; t < T; t += nwaves) {
    const int g = sel[t];
    // entries of the blocks before this graph's block: strided sum over the block totals + wave reduction
    const int b = (int)(t / kScanBlock);
    int off = 0;
    for (int i = lane; i < b; i += kWave) off += bs[i];
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) off += __shfl_xor(off, o, kWave);
    const int out_base = graph_ptr[t] + off;
    int* rp_out = dst_rowptr + t * M;
    int cnt;
    if (g >= 0) {
      const int* rp_in = src_rowptr + (long)g * M;
      const int src_base = rp_in[0];
      cnt = rp_in[M] - src_base;
      for (int r = lane; r < M; r += kWave) rp_out[r] = rp_in[r] - src_base + out_base;
      const int2* in = p.src_cv[c] + src_base;
      int2* out = p.dst_cv[c] + out_base;
      for (int i = lane; i < cnt; i += kWave) out[i] = in[i];
      if (row_pad) {
        const int* sl = p.src_slots[c] + (long)g * M;
        for (int r = lane; r < M; r += kWave) dst_slots[t * M + r] = sl[r];
      }
    } else if (row_pad) {                               // dummy graph of a row-padded batch: M rows x 4 pads
      cnt = 4 * M;
      for (int r = lane; r < M; r += kWave) {
        rp_out[r] = out_base + 4 * r;
        dst_slots[t * M + r] = (4 * r) | (4 << 16) | (r << 24);
      }
      int2* out = p.dst_cv[c] + out_base;
      for (int i = lane; i < cnt; i += kWave) out[i] = make_int2(KGCN_PAD_COL, 0);
    } else {
      cnt = 0;
      for (int r = lane; r < M; r += kWave) rp_out[r] = out_base;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      graph_ptr[t] = out_base;                        // no other wave reads entry t
      if (t == T - 1) {
        dst_rowptr[(long)T * M] = out_base + cnt;
        graph_ptr[T] = out_base + cnt;
      }
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

extern "C" int64_t kgcn_batch_assemble_workspace_bytes(int32_t num_sel) {
  if (num_sel <= 0) return 0;
  return (int64_t)KGCN_ASSEMBLE_MAX_CSR * ((num_sel + kScanBlock - 1) / kScanBlock) * 4;
}

extern "C" int kgcn_batch_assemble(const kgcn_assemble_plan* plan, const int32_t* sel, int32_t num_sel, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
  if (!plan) return fail("kgcn_batch_assemble: plan is NULL");
  if (plan->num_csr < 0 || plan->num_csr > KGCN_ASSEMBLE_MAX_CSR || plan->num_tables < 0 ||
      plan->num_tables > KGCN_ASSEMBLE_MAX_TABLES)
    return fail("kgcn_batch_assemble: %d containers / %d tables (at most %d / %d)", plan->num_csr, plan->num_tables,
                KGCN_ASSEMBLE_MAX_CSR, KGCN_ASSEMBLE_MAX_TABLES);
  if (num_sel < 0) return fail("kgcn_batch_assemble: negative num_sel");
  AssemblePlan p{};
  p.num_csr = plan->num_csr;
  p.num_tables = plan->num_tables;
  p.nb = (num_sel + kScanBlock - 1) / kScanBlock;
  for (int c = 0; c < plan->num_csr; ++c) {
    const kgcn_csr_batch* src = plan->src[c];
    if (int rc = validate_csr(src, "kgcn_batch_assemble", /*allow_row_pad=*/true)) return rc;
    if (!plan->dst_rowptr[c] || !plan->dst_graph_ptr[c]) return fail("kgcn_batch_assemble: container %d: dst_rowptr / dst_graph_ptr is NULL", c);
    if (src->row_pad == 4 && (!src->slots || (num_sel > 0 && src->rows > 0 && !plan->dst_slots[c])))
      return fail("kgcn_batch_assemble: container %d: a row-padded source needs slots and dst_slots", c);
    if ((int64_t)num_sel * src->rows >= (int64_t)INT32_MAX) return fail("kgcn_batch_assemble: T*M exceeds int32 row indexing");
    // the capacity must cover the worst case: the selection lives on the device
    const int64_t worst = (int64_t)num_sel * (src->row_pad && 4 * src->rows > src->max_nnz_per_graph ? 4 * src->rows
                                                                                                       : src->max_nnz_per_graph);
    if (plan->dst_cv_capacity[c] < worst || (worst > 0 && !plan->dst_cv[c]))
      return fail("kgcn_batch_assemble: container %d: dst_cv holds %lld entries, the worst case is %lld", c,
                  (long long)plan->dst_cv_capacity[c], (long long)worst);
    p.src_rowptr[c] = src->rowptr;
    p.src_cv[c] = reinterpret_cast<const int2*>(src->cv);
    p.src_slots[c] = src->slots;
    p.M[c] = src->rows;
    p.row_pad[c] = src->row_pad;
    p.graph_ptr[c] = plan->dst_graph_ptr[c];
    p.dst_rowptr[c] = plan->dst_rowptr[c];
    p.dst_cv[c] = reinterpret_cast<int2*>(plan->dst_cv[c]);
    p.dst_slots[c] = plan->dst_slots[c];
  }
  for (int k = 0; k < plan->num_tables; ++k) {
    if (plan->row_floats[k] < 0) return fail("kgcn_batch_assemble: table %d: negative row length", k);
    if (num_sel > 0 && plan->row_floats[k] > 0 && (!plan->table[k] || !plan->table_out[k]))
      return fail("kgcn_batch_assemble: table %d: NULL pointer", k);
    p.table[k] = plan->table[k];
    p.table_out[k] = plan->table_out[k];
    p.row_floats[k] = plan->row_floats[k];
  }
  hipStream_t s = as_stream(stream);
  if (num_sel == 0) {
    for (int c = 0; c < plan->num_csr; ++c) {
      hipError_t e = hipMemsetAsync(plan->dst_rowptr[c], 0, 4, s);
      if (e == hipSuccess) e = hipMemsetAsync(plan->dst_graph_ptr[c], 0, 4, s);
      if (e != hipSuccess) return fail("kgcn_batch_assemble: memset failed: %s", hipGetErrorString(e));
    }
    return 0;
  }
  if (!sel) return fail("kgcn_batch_assemble: sel is NULL");
  if (plan->num_csr + plan->num_tables == 0) return 0;
  const int64_t need = kgcn_batch_assemble_workspace_bytes(num_sel);
  if (!workspace || workspace_bytes < need)
    return fail("kgcn_batch_assemble: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
  int* block_sums = static_cast<int*>(workspace);
  if (plan->num_csr > 0) {
    hipLaunchKernelGGL(assemble_count_kernel, dim3(p.nb, plan->num_csr), dim3(kScanBlock), 0, s, p, sel, num_sel, block_sums);
    if (int rc = check_launch("assemble_count_kernel")) return rc;
  }
  long blocks = ((long)num_sel + 3) / 4;               // 4 waves per workgroup, one wave per graph
  if (blocks > (long)kNumCU * 8) blocks = (long)kNumCU * 8;
  hipLaunchKernelGGL(assemble_copy_kernel, dim3((unsigned)blocks, plan->num_csr + plan->num_tables), dim3(256), 0, s, p, sel,
                     num_sel, block_sums);
  return check_launch("assemble_copy_kernel");
}
