// Ragged-compact batches: only the VALID node rows of a padded batch exist in HBM.
//
// The reference pads every graph to max_node_num rows (kgcn/data_util.py:30-37, feed.py:127-133) and its ragged
// layers gather the first enabled_node_nums[b] rows of every graph, compute on the stacked [sum n_b, D] matrix and
// pad the result back (GraphDense kgcn/layers.py:243-254, GraphBatchNormalization :196-210).  On a Tox21-shaped batch
// (true sizes 5..50 of 50) 45 % of the rows every GEMM / aggregation / normalisation pass touches are padding.
// Here the WHOLE layer stack runs on the stacked matrix:
//
//   rows [graph_ptr[t], graph_ptr[t+1])   the n_t valid rows of graph t (graph_ptr = exclusive scan of the sizes)
//   rows [R, capacity)                    "padding representatives": zero features, no adjacency entries.  Every one of
//                                         them goes through the layers like a padded row of the reference's layout does
//                                         (GraphConv -> 0, activation -> act(0), un-ragged GraphDense -> the constant row
//                                         act(act(0) colsum(K) + bias), ragged BN -> 0), so row capacity-1 always HOLDS the
//                                         value every padded row of the padded formulation would have, and
//                                         GraphGather = sum of the valid rows + (N - n_t) x that row (quirk Q4) -- forward
//                                         and, through the same row, the gradient into the kernels / biases above it.
//   adjacency                             ONE block-diagonal [capacity x capacity] CSR (column = graph_ptr[t] + local
//                                         column), the container form of the kgcn-sparse path (data_util.py:698-845)
//
// capacity is fixed per batch SHAPE (>= R + 1), so a captured hipGraph can be replayed on batches of different R: only
// graph_ptr / rowptr contents change, never a launch dimension.
//
//   ragged_plan_*         per selected graph: n_t and its stored entries inside the valid block; multi-block exclusive scans
//   ragged_csr_kernel     one wave per graph: re-based rowptr slice, cv slice with global columns; tail rows
//   ragged_rows_kernel    one wave per graph: its n_t x d feature rows are ONE contiguous run in source and destination
//   ragged_gather_*       GraphGather over graph_ptr with the padding multiplicity
#include "kgcn_common.h"

namespace kgcn {

constexpr int kRScan = 256;

// exclusive scan of two values per thread over a 256-thread workgroup (wave shuffles + one LDS hop)
__device__ __forceinline__ int2 block_exclusive_scan2(int2 v, int2* total) {
  __shared__ int2 wave_tot[kRScan / kWave];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  int2 inc = v;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const int ux = __shfl_up(inc.x, o, kWave), uy = __shfl_up(inc.y, o, kWave);
    if (lane >= o) { inc.x += ux; inc.y += uy; }
  }
  if (lane == kWave - 1) wave_tot[wave] = inc;
  __syncthreads();
  int2 base = make_int2(0, 0), tot = make_int2(0, 0);
#pragma unroll
  for (int w = 0; w < kRScan / kWave; ++w) {
    const int2 t = wave_tot[w];
    if (w < wave) { base.x += t.x; base.y += t.y; }
    tot.x += t.x; tot.y += t.y;
  }
  __syncthreads();
  *total = tot;
  return make_int2(base.x + inc.x - v.x, base.y + inc.y - v.y);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// stage 1: (rows, entries) of every selected graph, block-local exclusive prefixes, block totals
__global__ __launch_bounds__(kRScan) void ragged_plan_count_kernel(
    const int* __restrict__ src_rowptr, const int* __restrict__ sizes, const int* __restrict__ sel, int T, int M,
    int* __restrict__ graph_ptr, int* __restrict__ entry_ptr, int2* __restrict__ block_sums) {
  const int t = blockIdx.x * kRScan + threadIdx.x;
  int2 v = make_int2(0, 0);
  if (t < T) {
    const int g = sel ? sel[t] : t;
    if (g >= 0) {
      const int n = clampi(sizes[g], 0, M);
      const int* rp = src_rowptr + (long)g * M;
      v = make_int2(n, rp[n] - rp[0]);
    }
  }
  int2 tot;
  const int2 ex = block_exclusive_scan2(v, &tot);
  if (t < T) { graph_ptr[t] = ex.x; entry_ptr[t] = ex.y; }
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// (One launch for the whole plan -- a single 1,024-thread workgroup that walks selection -> size -> row offsets for 8 graphs per
// thread and scans -- was tried for batches of <= 8,192 graphs: 10.6 us warm against 12.8 for the three launches, but 34 us inside
// the step, where the dataset's row offsets are cold: 8,192 scattered loads through ONE CU's address translation.  The count
// stage's 16+ workgroups spread them over as many CUs.)
// stage 2: one workgroup scans the block totals; the grand totals go to graph_ptr[T] / entry_ptr[T]
__global__ __launch_bounds__(kRScan) void ragged_plan_scan_kernel(int2* __restrict__ block_sums, int nb,
                                                                  int* __restrict__ rows_total, int* __restrict__ entries_total) {
  int2 carry = make_int2(0, 0);
  for (int b0 = 0; b0 < nb; b0 += kRScan) {
    const int i = b0 + threadIdx.x;
    const int2 v = i < nb ? block_sums[i] : make_int2(0, 0);
    int2 tot;
    const int2 ex = block_exclusive_scan2(v, &tot);
    if (i < nb) block_sums[i] = make_int2(carry.x + ex.x, carry.y + ex.y);
    carry.x += tot.x; carry.y += tot.y;
  }
  if (threadIdx.x == 0) { *rows_total = carry.x; *entries_total = carry.y; }
}

// stage 3: add the block offsets
__global__ __launch_bounds__(kRScan) void ragged_plan_finish_kernel(const int2* __restrict__ block_sums, int T,
                                                                    int* __restrict__ graph_ptr, int* __restrict__ entry_ptr) {
  const int t = blockIdx.x * kRScan + threadIdx.x;
  if (t < T) {
    const int2 b = block_sums[blockIdx.x];
    graph_ptr[t] += b.x;
    entry_ptr[t] += b.y;
  }
}

// stages 2 + 3 in one launch (<= 1,024 count blocks): every workgroup adds up the totals of the count blocks in front of it itself
// (<= 4 int2 per thread, one LDS reduction) -- the one-workgroup scan between two grid launches was a 4.6 us launch of its own per
// step of BASELINE config 4 (profiles/r05_k_cfg4_rocprof.txt).  block_sums stays as the count stage wrote it.
__global__ __launch_bounds__(kRScan) void ragged_plan_finish2_kernel(const int2* __restrict__ block_sums, int nb, int T,
                                                                     int* __restrict__ graph_ptr, int* __restrict__ entry_ptr) {
  __shared__ int red[4][kRScan];
  int px = 0, py = 0, tx = 0, ty = 0;
  for (int i = threadIdx.x; i < nb; i += kRScan) {
    const int2 v = block_sums[i];
    tx += v.x; ty += v.y;
    if (i < (int)blockIdx.x) { px += v.x; py += v.y; }
  }
  red[0][threadIdx.x] = px; red[1][threadIdx.x] = py; red[2][threadIdx.x] = tx; red[3][threadIdx.x] = ty;
  __syncthreads();
  for (int o = kRScan / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
#pragma unroll
      for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
    }
    __syncthreads();
  }
  const int t = blockIdx.x * kRScan + threadIdx.x;
  if (t < T) {
    graph_ptr[t] += red[0][0];
    entry_ptr[t] += red[1][0];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { graph_ptr[T] = red[2][0]; entry_ptr[T] = red[3][0]; }
}

// block-diagonal CSR of the valid blocks.  One wave per graph; the tail rows [R, capacity] are filled by the same grid.
// status[0] += entries of the selected graphs that lie outside their valid n_t x n_t block (rows >= n_t that store
// entries, columns >= n_t, or a count that disagrees with the plan): they have no place in the compact layout.
struct RaggedCsrJob {          // one container of a launch (blockIdx.y): A, A^T of a channel share the plan
  const int* src_rowptr; const int2* src_cv; long cv_capacity; int* dst_rowptr; int2* dst_cv;
};
struct RaggedCsrJobs { RaggedCsrJob j[2]; };
__global__ __launch_bounds__(256) void ragged_csr_kernel(
    RaggedCsrJobs jobs, const int* __restrict__ sel, int T, int M,
    const int* __restrict__ graph_ptr, const int* __restrict__ entry_ptr, int capacity_rows, int* __restrict__ status) {
  const RaggedCsrJob& jb = jobs.j[blockIdx.y];
  const int* __restrict__ src_rowptr = jb.src_rowptr;
  const int2* __restrict__ src_cv = jb.src_cv;
  const long cv_capacity = jb.cv_capacity;
  int* __restrict__ dst_rowptr = jb.dst_rowptr;
  int2* __restrict__ dst_cv = jb.dst_cv;
  const int lane = threadIdx.x & (kWave - 1);
  const long gtid = (long)blockIdx.x * 256 + threadIdx.x;
  const long nthreads = (long)gridDim.x * 256;
  const long wave0 = gtid / kWave, nwaves = nthreads / kWave;
  const int R = graph_ptr[T], E = entry_ptr[T];
  for (long i = R + gtid; i <= capacity_rows; i += nthreads) dst_rowptr[i] = E;
  for (long t = wave0; t < T; t += nwaves) {
    const int g = sel ? sel[t] : (int)t;
    if (g < 0) continue;
    const int r0 = graph_ptr[t], n = graph_ptr[t + 1] - r0;
    const int e0 = entry_ptr[t], cnt_plan = entry_ptr[t + 1] - e0;
    const int* rp = src_rowptr + (long)g * M;
    const int sbase = rp[0];
    const int cnt = rp[n] - sbase;
    const int ncopy = cnt < cnt_plan ? cnt : cnt_plan;
    for (int r = lane; r < n; r += kWave) {
      const int o = rp[r] - sbase;
      dst_rowptr[r0 + r] = e0 + (o < ncopy ? o : ncopy);
    }
    const bool fits = (long)e0 + ncopy <= cv_capacity;
    int bad = 0;
    if (fits) {
      for (int i = lane; i < ncopy; i += kWave) {
        int2 p = src_cv[sbase + i];
        if (p.x < 0 || p.x >= n) { ++bad; p.x = 0; p.y = 0; }      // keeps the container well formed; flagged
        p.x += r0;
        dst_cv[e0 + i] = p;
      }
    }
    if (status) {
#pragma unroll
      for (int o = kWave / 2; o > 0; o >>= 1) bad += __shfl_down(bad, o, kWave);
      if (lane == 0) {
        // entries stored in rows >= n_t, a count that disagrees with the plan (A^T of a batch with stray entries), or
        // a destination that is too small
        bad += (rp[M] - rp[n]) + (cnt != cnt_plan ? 1 : 0) + (fits ? 0 : (ncopy > 0 ? ncopy : 1));
        if (bad) atomicAdd(status, bad);
      }
    }
  }
}

// feature rows: graph t's n_t x d valid rows are contiguous in the padded source ([g, 0..n_t) x d) and in the compact
// destination (rows graph_ptr[t]..): one run of n_t*d floats per graph; tail rows are zeroed
__global__ __launch_bounds__(256) void ragged_rows_kernel(const float* __restrict__ src, const int* __restrict__ sel, int T,
                                                          int M, int d, const int* __restrict__ graph_ptr,
                                                          int capacity_rows, float* __restrict__ dst) {
  const int lane = threadIdx.x & (kWave - 1);
  const long gtid = (long)blockIdx.x * 256 + threadIdx.x;
  const long nthreads = (long)gridDim.x * 256;
  const int R = graph_ptr[T];
  const long tail0 = (long)R * d, tail1 = (long)capacity_rows * d;
  for (long i = tail0 + gtid; i < tail1; i += nthreads) dst[i] = 0.f;
  for (long t = gtid / kWave; t < T; t += nthreads / kWave) {
    const int g = sel ? sel[t] : (int)t;
    if (g < 0) continue;
    const int r0 = graph_ptr[t];
    const long len = (long)(graph_ptr[t + 1] - r0) * d;
    const float* s = src + (long)g * M * d;
    float* o = dst + (long)r0 * d;
    // the run starts at an arbitrary 4-byte offset (d = 81: 324-byte rows): dword accesses, four independent loads in
    // flight per lane
    long i = lane;
    for (; i + 3 * kWave < len; i += 4 * kWave) {
      const float a0 = s[i], a1 = s[i + kWave], a2 = s[i + 2 * kWave], a3 = s[i + 3 * kWave];
      o[i] = a0; o[i + kWave] = a1; o[i + 2 * kWave] = a2; o[i + 3 * kWave] = a3;
    }
    for (; i < len; i += kWave) o[i] = s[i];
  }
}

// the same copy into rows of dp >= d + 1 floats that end in [1 | 0 ...]: the [x | 1 | 0] operand of the aggregate-first GraphConv
// (layers.py; kgcn_augment_ones_f32 was a pass of its own over the compact rows: 17.9 us and 83 MB per step of BASELINE config 4,
// profiles/r05_h_cfg4_rocprof.txt).  A graph's n_t x dp destination floats are still ONE contiguous run.
__global__ __launch_bounds__(256) void ragged_rows_aug_kernel(const float* __restrict__ src, const int* __restrict__ sel, int T,
                                                              int M, int d, int dp, const int* __restrict__ graph_ptr,
                                                              int capacity_rows, float* __restrict__ dst) {
  const int lane = threadIdx.x & (kWave - 1);
  const long gtid = (long)blockIdx.x * 256 + threadIdx.x;
  const long nthreads = (long)gridDim.x * 256;
  const int R = graph_ptr[T];
  const float inv = 1.0f / (float)dp;
  // (j + 0.5) / dp in fp32: j < 2^22 keeps the quotient's error far below the 0.5 / dp that separates two rows
  auto split = [&](long j, int& row, int& c) __attribute__((always_inline)) {
    row = (int)(((float)j + 0.5f) * inv);
    c = (int)(j - (long)row * dp);
  };
  const long tail0 = (long)R * dp, tail1 = (long)capacity_rows * dp;
  for (long i = tail0 + gtid; i < tail1; i += nthreads) dst[i] = (int)(i % dp) == d ? 1.f : 0.f;
  for (long t = gtid / kWave; t < T; t += nthreads / kWave) {
    const int g = sel ? sel[t] : (int)t;
    if (g < 0) continue;
    const int r0 = graph_ptr[t];
    const long len = (long)(graph_ptr[t + 1] - r0) * dp;
    const float* s = src + (long)g * M * d;
    float* o = dst + (long)r0 * dp;
    auto value = [&](long j) __attribute__((always_inline)) {
      int row, c;
      split(j, row, c);
      return c < d ? s[(long)row * d + c] : (c == d ? 1.f : 0.f);
    };
    long i = lane;
    for (; i + 3 * kWave < len; i += 4 * kWave) {
      const float a0 = value(i), a1 = value(i + kWave), a2 = value(i + 2 * kWave), a3 = value(i + 3 * kWave);
      o[i] = a0; o[i + kWave] = a1; o[i + 2 * kWave] = a2; o[i + 3 * kWave] = a3;
    }
    for (; i < len; i += kWave) o[i] = value(i);
  }
}

// inverse of ragged_rows_kernel for results that are wanted in the padded layout: padded[t, r, :] = compact[graph_ptr[t]+r]
// for r < n_t, `fill` (the padding representative row, or zeros when fill_row < 0) elsewhere
__global__ __launch_bounds__(256) void ragged_expand_kernel(const float* __restrict__ src, int T, int M, int d,
                                                            const int* __restrict__ graph_ptr, int fill_row,
                                                            float* __restrict__ dst) {
  const long total = (long)T * M * d;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / d;
    const int c = (int)(i - row * d);
    const long t = row / M;
    const int r = (int)(row - t * M);
    const int r0 = graph_ptr[t], n = graph_ptr[t + 1] - r0;
    dst[i] = r < n ? src[(long)(r0 + r) * d + c] : (fill_row >= 0 ? src[(long)fill_row * d + c] : 0.f);
  }
}

// GraphGather (kgcn/layers.py:163-164) on the compact layout: out[b] = sum of the valid rows + (N - n_b) * x[pad_row]
template <int VEC>
__global__ __launch_bounds__(256) void ragged_gather_fwd_kernel(const float* __restrict__ x, const int* __restrict__ graph_ptr,
                                                                long B, int N, int d, int pad_row, float* __restrict__ out) {
  const int dv = d / VEC;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < B * dv; i += (long)gridDim.x * 256) {
    const long b = i / dv;
    const int c = (int)(i - b * dv) * VEC;
    const int r0 = graph_ptr[b], r1 = graph_ptr[b + 1];
    // the rows in the reference's order (one running sum per column: the fp32 result does not depend on this kernel's shape),
    // four loads in flight per thread -- a thread walks up to N rows, and one dependent load at a time is all latency
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    int r = r0;
    for (; r + 4 <= r1; r += 4) {
      float v[4][VEC];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[u][j] = x[(long)(r + u) * d + c + j];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += v[u][j];
    }
    for (; r < r1; ++r) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += x[(long)r * d + c + j];
    }
    const float mult = (float)(N - (r1 - r0));
#pragma unroll
    for (int j = 0; j < VEC; ++j) out[b * d + c + j] = acc[j] + mult * x[(long)pad_row * d + c + j];
  }
}

// dx[r] = g[graph of r] on valid rows (one wave per graph), 0 on the tail rows
__global__ __launch_bounds__(256) void ragged_gather_bwd_kernel(const float* __restrict__ g, const int* __restrict__ graph_ptr,
                                                                int B, int d, int capacity_rows, float* __restrict__ dx) {
  const int lane = threadIdx.x & (kWave - 1);
  const long gtid = (long)blockIdx.x * 256 + threadIdx.x;
  const long nthreads = (long)gridDim.x * 256;
  const int R = graph_ptr[B];
  for (long i = (long)R * d + gtid; i < (long)capacity_rows * d; i += nthreads) dx[i] = 0.f;
  for (long b = gtid / kWave; b < B; b += nthreads / kWave) {
    const int r0 = graph_ptr[b];
    const long len = (long)(graph_ptr[b + 1] - r0) * d;
    float* o = dx + (long)r0 * d;
    const float* gb = g + b * d;
    if (d <= kWave) {
      // lane -> (row offset, column) walked without dividing per element
      const int rows_per_pass = kWave / d;
      const int rl = lane / d, c = lane - rl * d;
      if (rl < rows_per_pass) {
        const float v = gb[c];
        for (long i = (long)rl * d + c; i < len; i += (long)rows_per_pass * d) o[i] = v;
      }
    } else {
      for (long i = lane; i < len; i += kWave) o[i] = gb[i % d];
    }
  }
}

// part[blockIdx.x, c] = sum over this workgroup's slice of graphs of (N - n_b) g[b, c]; blockIdx.y = block of <= 256
// columns.  A fixed-order second stage (reduce_partials) adds the slices into dx[pad_row]: deterministic.
constexpr int kPadParts = 128;
__global__ __launch_bounds__(256) void ragged_gather_pad_bwd_kernel(const float* __restrict__ g, const int* __restrict__ graph_ptr,
                                                                    int B, int N, int d, float* __restrict__ part) {
  __shared__ float red[256];
  const int tid = threadIdx.x;
  const int c0 = blockIdx.y * 256;
  const int cb = d - c0 < 256 ? d - c0 : 256;
  const int rpp = 256 / cb;
  const int c = tid % cb, rl = tid / cb;
  const int per = (B + gridDim.x - 1) / gridDim.x;
  const int b0 = blockIdx.x * per, b1 = b0 + per < B ? b0 + per : B;
  float a = 0.f;
  if (rl < rpp) {
    for (int b = b0 + rl; b < b1; b += rpp) a += (float)(N - (graph_ptr[b + 1] - graph_ptr[b])) * g[(long)b * d + c0 + c];
  }
  red[tid] = a;
  __syncthreads();
  if (tid < cb) {
    float s = 0.f;
    for (int k = 0; k < rpp; ++k) s += red[k * cb + tid];
    part[(long)blockIdx.x * d + c0 + tid] = s;
  }
}

int launch_reduce_partials(const float* part, int nparts, long n, float* out, hipStream_t s);

// out[r, 0:din] = x[r, 0:din], out[r, din] = 1, out[r, din+1:out_ld] = 0: the operand of an aggregate-FIRST GraphConv,
//   A (X W + 1 b) = (A [X | 1]) [W ; b]        (kgcn/layers.py:112-113 evaluated in the cheaper order when din + 1 < dout)
// -- the column of ones turns the bias term rowsum(A) (x) b into one more row of the contraction.
// BWD: dx[r, 0:din] = g[r, 0:din] (the gradient of the ones / padding columns is dropped).
template <bool BWD>
__global__ __launch_bounds__(256) void augment_ones_kernel(const float* __restrict__ src, long m, int din, long src_ld,
                                                           float* __restrict__ dst, long dst_ld) {
  const int w = BWD ? din : (int)dst_ld;
  const long total = m * w;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / w;
    const int c = (int)(i - r * w);
    if (BWD) dst[r * dst_ld + c] = src[r * src_ld + c];
    else dst[r * dst_ld + c] = c < din ? src[r * src_ld + c] : (c == din ? 1.f : 0.f);
  }
}

static unsigned grid_cap(long work_items, long cap = (long)kNumCU * 16) {
  long b = (work_items + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace kgcn

using namespace kgcn;

extern "C" int64_t kgcn_ragged_workspace_bytes(int32_t num_sel) {
  if (num_sel <= 0) return 0;
  return (int64_t)((num_sel + kRScan - 1) / kRScan) * 8;
}

extern "C" int kgcn_ragged_plan(const kgcn_csr_batch* src, const int32_t* sizes, const int32_t* sel, int32_t num_sel,
                                int32_t* graph_ptr, int32_t* entry_ptr, void* workspace, int64_t workspace_bytes,
                                void* stream) {
  if (int rc = validate_csr(src, "kgcn_ragged_plan")) return rc;
  if (num_sel < 0) return fail("kgcn_ragged_plan: negative num_sel");
  if (!graph_ptr || !entry_ptr) return fail("kgcn_ragged_plan: graph_ptr / entry_ptr is NULL");
  if (src->rows != src->cols) return fail("kgcn_ragged_plan: adjacency must be square (M=%d K=%d)", src->rows, src->cols);
  hipStream_t s = as_stream(stream);
  if (num_sel == 0) {
    hipError_t e = hipMemsetAsync(graph_ptr, 0, 4, s);
    if (e == hipSuccess) e = hipMemsetAsync(entry_ptr, 0, 4, s);
    return e == hipSuccess ? 0 : fail("kgcn_ragged_plan: memset failed: %s", hipGetErrorString(e));
  }
  if (!sizes) return fail("kgcn_ragged_plan: sizes is NULL");
  if (!sel && num_sel != src->num_graphs)
    return fail("kgcn_ragged_plan: sel is NULL (identity) but num_sel=%d != %d graphs", num_sel, src->num_graphs);
  const int64_t need = kgcn_ragged_workspace_bytes(num_sel);
  if (!workspace || workspace_bytes < need)
    return fail("kgcn_ragged_plan: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
  const int nb = (num_sel + kRScan - 1) / kRScan;
  int2* bs = static_cast<int2*>(workspace);
  hipLaunchKernelGGL(ragged_plan_count_kernel, dim3(nb), dim3(kRScan), 0, s, src->rowptr, sizes, sel, num_sel, src->rows,
                     graph_ptr, entry_ptr, bs);
  if (int rc = check_launch("ragged_plan_count_kernel")) return rc;
  if (nb <= 1024) {
    hipLaunchKernelGGL(ragged_plan_finish2_kernel, dim3(nb), dim3(kRScan), 0, s, bs, nb, num_sel, graph_ptr, entry_ptr);
    return check_launch("ragged_plan_finish2_kernel");
  }
  hipLaunchKernelGGL(ragged_plan_scan_kernel, dim3(1), dim3(kRScan), 0, s, bs, nb, graph_ptr + num_sel, entry_ptr + num_sel);
  if (int rc = check_launch("ragged_plan_scan_kernel")) return rc;
  if (nb > 1) {
    hipLaunchKernelGGL(ragged_plan_finish_kernel, dim3(nb), dim3(kRScan), 0, s, bs, num_sel, graph_ptr, entry_ptr);
    if (int rc = check_launch("ragged_plan_finish_kernel")) return rc;
  }
  return 0;
}

extern "C" int kgcn_ragged_compact_csr(const kgcn_csr_batch* src, const int32_t* sel, int32_t num_sel,
                                       const int32_t* graph_ptr, const int32_t* entry_ptr, int32_t capacity_rows,
                                       int32_t* dst_rowptr, int32_t* dst_cv, int64_t dst_cv_capacity, int32_t* status,
                                       void* stream) {
  if (int rc = validate_csr(src, "kgcn_ragged_compact_csr")) return rc;
  if (num_sel < 0 || capacity_rows < 0) return fail("kgcn_ragged_compact_csr: negative size");
  if (!graph_ptr || !entry_ptr || !dst_rowptr) return fail("kgcn_ragged_compact_csr: NULL operand");
  if (src->rows != src->cols) return fail("kgcn_ragged_compact_csr: adjacency must be square");
  if (!sel && num_sel != src->num_graphs) return fail("kgcn_ragged_compact_csr: sel is NULL but num_sel != graphs");
  if (dst_cv_capacity < 0 || (dst_cv_capacity > 0 && !dst_cv)) return fail("kgcn_ragged_compact_csr: dst_cv is NULL");
  const long waves_needed = num_sel > 0 ? num_sel : 1;
  const unsigned blocks = grid_cap(waves_needed * kWave, (long)kNumCU * 32);
  RaggedCsrJobs jobs{};
  jobs.j[0] = RaggedCsrJob{src->rowptr, reinterpret_cast<const int2*>(src->cv), (long)dst_cv_capacity, dst_rowptr,
                           reinterpret_cast<int2*>(dst_cv)};
  hipLaunchKernelGGL(ragged_csr_kernel, dim3(blocks, 1), dim3(256), 0, as_stream(stream), jobs, sel, num_sel, src->rows, graph_ptr,
                     entry_ptr, capacity_rows, status);
  return check_launch("ragged_csr_kernel");
}

extern "C" int kgcn_ragged_compact_csr_pair(const kgcn_csr_batch* src, const kgcn_csr_batch* src_t, const int32_t* sel,
                                            int32_t num_sel, const int32_t* graph_ptr, const int32_t* entry_ptr,
                                            int32_t capacity_rows, int32_t* dst_rowptr, int32_t* dst_cv, int64_t dst_cv_capacity,
                                            int32_t* dst_t_rowptr, int32_t* dst_t_cv, int64_t dst_t_cv_capacity, int32_t* status,
                                            void* stream) {
  if (int rc = validate_csr(src, "kgcn_ragged_compact_csr_pair")) return rc;
  if (int rc = validate_csr(src_t, "kgcn_ragged_compact_csr_pair")) return rc;
  if (num_sel < 0 || capacity_rows < 0) return fail("kgcn_ragged_compact_csr_pair: negative size");
  if (!graph_ptr || !entry_ptr || !dst_rowptr || !dst_t_rowptr) return fail("kgcn_ragged_compact_csr_pair: NULL operand");
  if (src->rows != src->cols || src_t->rows != src->rows || src_t->cols != src->cols || src_t->num_graphs != src->num_graphs)
    return fail("kgcn_ragged_compact_csr_pair: the two containers must be square and of one shape");
  if (!sel && num_sel != src->num_graphs) return fail("kgcn_ragged_compact_csr_pair: sel is NULL but num_sel != graphs");
  if (dst_cv_capacity < 0 || (dst_cv_capacity > 0 && !dst_cv) || dst_t_cv_capacity < 0 || (dst_t_cv_capacity > 0 && !dst_t_cv))
    return fail("kgcn_ragged_compact_csr_pair: dst_cv is NULL");
  const long waves_needed = num_sel > 0 ? num_sel : 1;
  const unsigned blocks = grid_cap(waves_needed * kWave, (long)kNumCU * 16);
  RaggedCsrJobs jobs{};
  jobs.j[0] = RaggedCsrJob{src->rowptr, reinterpret_cast<const int2*>(src->cv), (long)dst_cv_capacity, dst_rowptr,
                           reinterpret_cast<int2*>(dst_cv)};
  jobs.j[1] = RaggedCsrJob{src_t->rowptr, reinterpret_cast<const int2*>(src_t->cv), (long)dst_t_cv_capacity, dst_t_rowptr,
                           reinterpret_cast<int2*>(dst_t_cv)};
  hipLaunchKernelGGL(ragged_csr_kernel, dim3(blocks, 2), dim3(256), 0, as_stream(stream), jobs, sel, num_sel, src->rows, graph_ptr,
                     entry_ptr, capacity_rows, status);
  return check_launch("ragged_csr_kernel");
}

// Row blocks of whole molecules (kgcn_csr_batch.block_ptr): block k starts at the first molecule whose first row is >= k * S.
// Thread t <= T owns boundary b_t = graph_ptr[t] (b_T = R) and writes it to every k in (floor(b_{t-1} / S), floor(b_t / S)]
// (t = 0: k = 0) -- each k <= floor(R / S) has exactly one writer; thread T also lays the blocks of the padding rows behind R.
__global__ __launch_bounds__(256) void ragged_blocks_kernel(const int* __restrict__ graph_ptr, int T, int capacity_rows, int nblocks,
                                                            int* __restrict__ block_ptr) {
  constexpr int S = KGCN_RAGGED_BLOCK_ROWS;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t > T) return;
  const int b = graph_ptr[t];
  const int k0 = t == 0 ? 0 : graph_ptr[t - 1] / S + 1;
  for (int k = k0; k <= b / S && k <= nblocks; ++k) block_ptr[k] = b < capacity_rows ? b : capacity_rows;
  if (t == T) {
    int k = b / S + 1;
    for (long row = b; k <= nblocks; ++k, row += S) block_ptr[k] = row < capacity_rows ? (int)row : capacity_rows;
  }
}

extern "C" int32_t kgcn_ragged_block_rows(void) { return KGCN_RAGGED_BLOCK_ROWS; }

extern "C" int32_t kgcn_ragged_num_blocks(int32_t capacity_rows) {
  return capacity_rows <= 0 ? 0 : (capacity_rows + KGCN_RAGGED_BLOCK_ROWS - 1) / KGCN_RAGGED_BLOCK_ROWS + 2;
}

extern "C" int kgcn_ragged_blocks(const int32_t* graph_ptr, int32_t num_sel, int32_t capacity_rows, int32_t* block_ptr,
                                  void* stream) {
  if (num_sel < 0 || capacity_rows < 0) return fail("kgcn_ragged_blocks: negative size");
  if (capacity_rows == 0) return 0;
  if (!graph_ptr || !block_ptr) return fail("kgcn_ragged_blocks: NULL operand");
  const int nb = kgcn_ragged_num_blocks(capacity_rows);
  hipLaunchKernelGGL(ragged_blocks_kernel, dim3((unsigned)((num_sel + 1 + 255) / 256)), dim3(256), 0, as_stream(stream), graph_ptr,
                     num_sel, capacity_rows, nb, block_ptr);
  return check_launch("ragged_blocks_kernel");
}

extern "C" int kgcn_ragged_compact_rows_f32(const float* src, const int32_t* sel, int32_t num_sel, int32_t n_nodes,
                                            int32_t d, const int32_t* graph_ptr, int32_t capacity_rows, float* dst,
                                            void* stream) {
  if (num_sel < 0 || n_nodes < 0 || d < 0 || capacity_rows < 0) return fail("kgcn_ragged_compact_rows_f32: negative size");
  if (capacity_rows == 0 || d == 0) return 0;
  if (!graph_ptr || !dst || (!src && num_sel > 0)) return fail("kgcn_ragged_compact_rows_f32: NULL operand");
  const unsigned blocks = grid_cap((long)(num_sel > 0 ? num_sel : 1) * kWave, (long)kNumCU * 32);
  hipLaunchKernelGGL(ragged_rows_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), src, sel, num_sel, n_nodes, d,
                     graph_ptr, capacity_rows, dst);
  return check_launch("ragged_rows_kernel");
}

extern "C" int kgcn_ragged_compact_rows_aug_f32(const float* src, const int32_t* sel, int32_t num_sel, int32_t n_nodes,
                                                int32_t d, const int32_t* graph_ptr, int32_t capacity_rows, float* dst,
                                                int32_t dst_ld, void* stream) {
  if (num_sel < 0 || n_nodes < 0 || d < 0 || capacity_rows < 0) return fail("kgcn_ragged_compact_rows_aug_f32: negative size");
  if (dst_ld < d + 1) return fail("kgcn_ragged_compact_rows_aug_f32: dst_ld %d leaves no room for the ones column behind %d features", dst_ld, d);
  if ((long)n_nodes * dst_ld >= (1L << 22)) return fail("kgcn_ragged_compact_rows_aug_f32: n_nodes * dst_ld = %ld >= 2^22", (long)n_nodes * dst_ld);
  if (capacity_rows == 0) return 0;
  if (!graph_ptr || !dst || (!src && num_sel > 0 && d > 0)) return fail("kgcn_ragged_compact_rows_aug_f32: NULL operand");
  const unsigned blocks = grid_cap((long)(num_sel > 0 ? num_sel : 1) * kWave, (long)kNumCU * 32);
  hipLaunchKernelGGL(ragged_rows_aug_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), src, sel, num_sel, n_nodes, d, dst_ld,
                     graph_ptr, capacity_rows, dst);
  return check_launch("ragged_rows_aug_kernel");
}

extern "C" int kgcn_ragged_expand_rows_f32(const float* src, int32_t num_graphs, int32_t n_nodes, int32_t d,
                                           const int32_t* graph_ptr, int32_t fill_row, float* dst, void* stream) {
  if (num_graphs < 0 || n_nodes < 0 || d < 0) return fail("kgcn_ragged_expand_rows_f32: negative size");
  if (num_graphs == 0 || n_nodes == 0 || d == 0) return 0;
  if (!src || !graph_ptr || !dst) return fail("kgcn_ragged_expand_rows_f32: NULL operand");
  hipLaunchKernelGGL(ragged_expand_kernel, dim3(grid_cap((long)num_graphs * n_nodes * d)), dim3(256), 0, as_stream(stream),
                     src, num_graphs, n_nodes, d, graph_ptr, fill_row, dst);
  return check_launch("ragged_expand_kernel");
}

extern "C" int kgcn_ragged_gather_fwd_f32(const float* x, const int32_t* graph_ptr, int64_t batch, int32_t n_nodes,
                                          int32_t d, int32_t pad_row, float* out, void* stream) {
  if (batch < 0 || n_nodes < 0 || d < 0 || pad_row < 0) return fail("kgcn_ragged_gather_fwd_f32: negative argument");
  if (batch == 0 || d == 0) return 0;
  if (!x || !graph_ptr || !out) return fail("kgcn_ragged_gather_fwd_f32: NULL operand");
  if (d % 4 == 0)
    hipLaunchKernelGGL((ragged_gather_fwd_kernel<4>), dim3(grid_cap(batch * (d / 4))), dim3(256), 0, as_stream(stream), x,
                       graph_ptr, (long)batch, n_nodes, d, pad_row, out);
  else if (d % 2 == 0)
    hipLaunchKernelGGL((ragged_gather_fwd_kernel<2>), dim3(grid_cap(batch * (d / 2))), dim3(256), 0, as_stream(stream), x,
                       graph_ptr, (long)batch, n_nodes, d, pad_row, out);
  else
    hipLaunchKernelGGL((ragged_gather_fwd_kernel<1>), dim3(grid_cap(batch * d)), dim3(256), 0, as_stream(stream), x,
                       graph_ptr, (long)batch, n_nodes, d, pad_row, out);
  return check_launch("ragged_gather_fwd_kernel");
}

extern "C" int64_t kgcn_ragged_gather_bwd_workspace_bytes(int32_t d) {
  return d > 0 ? (int64_t)kPadParts * d * 4 : 0;
}

extern "C" int kgcn_ragged_gather_bwd_f32(const float* dout_grad, const int32_t* graph_ptr, int64_t batch, int32_t n_nodes,
                                          int32_t d, int32_t pad_row, int32_t capacity_rows, float* dx, void* workspace,
                                          int64_t workspace_bytes, void* stream) {
  if (batch < 0 || n_nodes < 0 || d < 0 || capacity_rows < 0) return fail("kgcn_ragged_gather_bwd_f32: negative argument");
  if (capacity_rows == 0 || d == 0) return 0;
  if (pad_row < 0 || pad_row >= capacity_rows) return fail("kgcn_ragged_gather_bwd_f32: pad_row outside the capacity");
  if (!graph_ptr || !dx || (!dout_grad && batch > 0)) return fail("kgcn_ragged_gather_bwd_f32: NULL operand");
  if (batch >= INT32_MAX) return fail("kgcn_ragged_gather_bwd_f32: batch too large");
  if (!workspace || workspace_bytes < kgcn_ragged_gather_bwd_workspace_bytes(d))
    return fail("kgcn_ragged_gather_bwd_f32: workspace %lld < %lld bytes", (long long)workspace_bytes,
                (long long)kgcn_ragged_gather_bwd_workspace_bytes(d));
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(ragged_gather_bwd_kernel, dim3(grid_cap((batch > 0 ? batch : 1) * kWave, (long)kNumCU * 32)), dim3(256),
                     0, s, dout_grad, graph_ptr, (int)batch, d, capacity_rows, dx);
  if (int rc = check_launch("ragged_gather_bwd_kernel")) return rc;
  int nparts = (int)((batch + 31) / 32);
  if (nparts > kPadParts) nparts = kPadParts;
  if (nparts < 1) nparts = 1;
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(ragged_gather_pad_bwd_kernel, dim3(nparts, (d + 255) / 256), dim3(256), 0, s, dout_grad, graph_ptr,
                     (int)batch, n_nodes, d, part);
  if (int rc = check_launch("ragged_gather_pad_bwd_kernel")) return rc;
  return launch_reduce_partials(part, nparts, d, dx + (long)pad_row * d, s);
}

extern "C" int kgcn_augment_ones_f32(const float* x, int64_t m, int32_t din, int64_t x_ld, float* out, int64_t out_ld,
                                     void* stream) {
  if (m < 0 || din <= 0) return fail("kgcn_augment_ones_f32: bad shape m=%lld din=%d", (long long)m, din);
  if (m == 0) return 0;
  if (!x || !out) return fail("kgcn_augment_ones_f32: NULL operand");
  if (x_ld < din || out_ld < din + 1) return fail("kgcn_augment_ones_f32: leading dimension too small (out_ld >= din + 1)");
  hipLaunchKernelGGL(augment_ones_kernel<false>, dim3(grid_cap(m * out_ld, (long)kNumCU * 32)), dim3(256), 0,
                     as_stream(stream), x, (long)m, din, (long)x_ld, out, (long)out_ld);
  return check_launch("augment_ones_kernel");
}

extern "C" int kgcn_augment_ones_bwd_f32(const float* dout_grad, int64_t m, int32_t din, int64_t g_ld, float* dx,
                                         int64_t dx_ld, void* stream) {
  if (m < 0 || din <= 0) return fail("kgcn_augment_ones_bwd_f32: bad shape m=%lld din=%d", (long long)m, din);
  if (m == 0) return 0;
  if (!dout_grad || !dx) return fail("kgcn_augment_ones_bwd_f32: NULL operand");
  if (g_ld < din + 1 || dx_ld < din) return fail("kgcn_augment_ones_bwd_f32: leading dimension too small");
  hipLaunchKernelGGL(augment_ones_kernel<true>, dim3(grid_cap(m * din, (long)kNumCU * 32)), dim3(256), 0, as_stream(stream),
                     dout_grad, (long)m, din, (long)g_ld, dx, (long)dx_ld);
  return check_launch("augment_ones_kernel<bwd>");
}
