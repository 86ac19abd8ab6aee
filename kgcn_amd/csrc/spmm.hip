// Batched sparse x dense aggregation for gfx950 (MI355X):  out[t] = beta*out[t] + A[t] @ rhs[t].
//
// Replaces the reference's custom ops Bspmm / Bspmdt / Bconv (kgcn/bspmm_call.py:16,
// kgcn/batched_call.py:27, kgcn/bconv_call.py:23) and the default-branch
// tf.sparse_tensor_dense_matmul loop (kgcn/layers.py:105-116).  The op is HBM-bound
// (~0.7 flop/B): the design goal is to touch every rhs/out byte exactly once with 16-byte
// coalesced accesses and to keep all irregular (gather) traffic inside LDS.
//
// Two kernels:
//  * spmm_tile_kernel   -- small graphs (the molecular case): ONE WAVE PER GRAPH.  The graph's
//    rhs block [K x d] is streamed HBM -> LDS with 1 KiB-per-instruction dwordx4 loads, its CSR
//    slice (rowptr + interleaved col/val pairs) is staged next to it, then 64/LPR rows are
//    aggregated concurrently (LPR lanes x float4 cover one row), neighbours gathered from LDS
//    with conflict-free ds_read_b128, and 64/LPR consecutive output rows leave as ONE fully
//    coalesced dwordx4 store instruction.  A one-wave workgroup needs no cross-wave barrier.
//  * spmm_gather_kernel -- any shape (big graphs, the block-diagonal [sumN x sumN] matrix of
//    kgcn/data_util.py:698-845, d not a multiple of 4): classic CSR-vector, LPR lanes per row,
//    neighbours gathered through L1/L2.
#include "kgcn_common.h"

namespace kgcn {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// ------------------------------------------------------------------------------------------------
// What one launch aggregates:  out[t] = act( beta*out[t] + sum_c A_c[t] @ (rhs_c[t] (.) act'(aout[t])) ).
//   * up to MAX_CH adjacency channels in ONE launch (tf.add_n over the channels, kgcn/layers.py:115; the degree split
//     of kgcn/data_util.py:76-122 has 6): the output is written once instead of read-modify-written per channel,
//   * act (KGCN_ACT_*): the activation the model applies to the layer output (tf.sigmoid / tf.nn.relu / tf.tanh around
//     GraphConv in example_model/*.py) as an epilogue -- saves one read + one write of the activation tensor,
//   * dact + aout: backward of such a layer.  The incoming gradient is multiplied by act'(.) expressed in the saved
//     layer OUTPUT (sigmoid: a(1-a), relu: a > 0, tanh: 1 - a^2) while it is gathered, so d pre-activation never
//     exists in HBM.
// ------------------------------------------------------------------------------------------------
constexpr int MAX_CH = 8;
struct SpmmChannels {
  const int* rowptr[MAX_CH];
  const int2* cv[MAX_CH];
  int max_nnz[MAX_CH];
  int n;
  long rhs_cs;           // element offset between the rhs operands of consecutive channels
};

// ------------------------------------------------------------------------------------------------
// LDS-staged tile kernel: one wave (= one 64-thread workgroup) per graph.
// LDS layout per channel: tile[K*d] | ecv[max_nnz] (int2) | rp[M+1]   (all channels staged, then one row loop)
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline size_t tile_chan_bytes(int M, int K, int d, int max_nnz) {
  return ((((size_t)K * d * 4 + 7) & ~(size_t)7) + (size_t)max_nnz * 8 + (size_t)(M + 1) * 4 + 15) & ~(size_t)15;
}

// NW waves per workgroup: 1 for tiles up to 20 KiB (8 workgroups per CU); 4 for wide operands (d = 256, N = 50: one 51 KB
// tile, three workgroups per CU -- every load and store instruction then covers whole 1 KiB rows).
// VEC floats per lane (4: dwordx4 everywhere; 2: widths like 50 that are even but not a multiple of 4).  blockIdx.y selects
// a slice of ds columns of a wide operand (d = 256: four slices of 64 keep the tile within the LDS budget of 8 waves per
// CU; each row of a slice is still one contiguous 256-byte segment).
template <int VEC> struct SpVec;
template <> struct SpVec<4> { using T = f32x4; };
template <> struct SpVec<2> { using T = f32x2; };

// DOT (GINAggregate backward, kgcn/layers.py:469): while the gradient tile of channel 0 is staged, the matching tile of `dotx`
// (the layer input) is read next to it and <grad, x> of this graph is accumulated: d epsilon without a pass of its own
// (one partial per workgroup in dot_part, fixed-order second stage).  Contiguous operands only (rhs_ld == ds).
template <int LPR, int VEC, int NW, bool DOT = false>
__global__ __launch_bounds__(64 * NW) void spmm_tile_kernel(
    SpmmChannels ch, const float* __restrict__ rhs, long rhs_ld, long rhs_gs, float* __restrict__ out, long out_ld,
    long out_gs, int M, int K, int ds, int nslices, float beta, const float* __restrict__ self_scale, int act,
    const float* __restrict__ aout, int dact, const float* __restrict__ dotx = nullptr, float* __restrict__ dot_part = nullptr) {
  using V = typename SpVec<VEC>::T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // slices of one graph are neighbours in the grid: they run at the same time and share the DRAM pages of its rows
  const int t = blockIdx.x / nslices;
  const int col0 = (blockIdx.x - t * nslices) * ds;
  const int lane = threadIdx.x;                 // thread of the workgroup (NW waves)
  constexpr int NT = 64 * NW;
  const int dv = ds / VEC;
  const int nv = K * dv;
  const int step_r = NT / dv, step_c = NT - step_r * dv;
  auto ldv = [](const float* p) { return *reinterpret_cast<const V*>(p); };
  auto stv = [](float* p, V v) { *reinterpret_cast<V*>(p) = v; };

  // ---- stage: per channel the rhs block (vector loads, fully coalesced; times act'(aout) in a backward launch) and
  // the CSR slice -----------------------------------------------------------------------------
  float dot_acc = 0.f;
  size_t off = 0;
  for (int c = 0; c < ch.n; ++c) {
    float* tile = reinterpret_cast<float*>(smem + off);
    int2* ecv = reinterpret_cast<int2*>(smem + off + (((size_t)K * ds * 4 + 7) & ~(size_t)7));
    int* rp = reinterpret_cast<int*>(ecv + ch.max_nnz[c]);
    const int* grp = ch.rowptr[c] + (long)t * M;
    const int base = grp[0];
    const int cnt = grp[M] - base;
    const float* rb = rhs + c * ch.rhs_cs + (long)t * rhs_gs + col0;
    if (dact == KGCN_ACT_NONE) {
      if (rhs_ld == ds) {
        const V* src = reinterpret_cast<const V*>(rb);
        V* dst = reinterpret_cast<V*>(tile);
        if constexpr (DOT) {
          if (c == 0) {
            const V* xs = reinterpret_cast<const V*>(dotx + (long)t * rhs_gs + col0);
#pragma unroll 4
            for (int i = lane; i < nv; i += NT) {
              const V v = src[i], xv = xs[i];
              dst[i] = v;
#pragma unroll
              for (int j = 0; j < VEC; ++j) dot_acc = __builtin_fmaf(v[j], xv[j], dot_acc);
            }
          } else {
#pragma unroll 4
            for (int i = lane; i < nv; i += NT) dst[i] = src[i];
          }
        } else {
#pragma unroll 4
          for (int i = lane; i < nv; i += NT) dst[i] = src[i];
        }
      } else {
        int r = lane / dv, cc = lane - r * dv;             // (row, vector) of element i, walked without dividing
        int i = lane;
        for (; i + 3 * NT < nv; i += 4 * NT) {                   // four loads in flight per lane
          V v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            v[u] = ldv(rb + (long)r * rhs_ld + cc * VEC);
            r += step_r; cc += step_c;
            if (cc >= dv) { cc -= dv; ++r; }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) stv(tile + (size_t)(i + NT * u) * VEC, v[u]);
        }
        for (; i < nv; i += NT) {
          stv(tile + (size_t)i * VEC, ldv(rb + (long)r * rhs_ld + cc * VEC));
          r += step_r; cc += step_c;
          if (cc >= dv) { cc -= dv; ++r; }
        }
      }
    } else {
      const float* ab = aout + (long)t * rhs_gs + col0;    // same layout as the gradient
      int r = lane / dv, cc = lane - r * dv;
      int i = lane;
      for (; i + 3 * NT < nv; i += 4 * NT) {                     // eight loads in flight per lane
        V v[4], a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          v[u] = ldv(rb + (long)r * rhs_ld + cc * VEC);
          a[u] = ldv(ab + (long)r * rhs_ld + cc * VEC);
          r += step_r; cc += step_c;
          if (cc >= dv) { cc -= dv; ++r; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) v[u][j] *= act_dout(a[u][j], dact);
          stv(tile + (size_t)(i + NT * u) * VEC, v[u]);
        }
      }
      for (; i < nv; i += NT) {
        V v = ldv(rb + (long)r * rhs_ld + cc * VEC);
        const V a = ldv(ab + (long)r * rhs_ld + cc * VEC);
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] *= act_dout(a[j], dact);
        stv(tile + (size_t)i * VEC, v);
        r += step_r; cc += step_c;
        if (cc >= dv) { cc -= dv; ++r; }
      }
    }
    for (int i = lane; i < cnt; i += NT) ecv[i] = ch.cv[c][base + i];
    for (int i = lane; i <= M; i += NT) rp[i] = grp[i] - base;
    off += tile_chan_bytes(M, K, ds, ch.max_nnz[c]);
  }
  __syncthreads();  // orders the LDS writes before the gathers (NW = 1: a compiler-level barrier only)

  // ---- aggregate: 64/LPR rows at a time, LPR lanes x VEC floats per row, channels innermost ------
  constexpr int RPW = NT / LPR;
  const int sub = lane / LPR;
  const int cl = lane % LPR;
  const bool col_ok = cl * VEC < ds;
  const float sscale = self_scale ? self_scale[0] : 0.f;
  float* ob = out + (long)t * out_gs + col0;
  for (int r0 = 0; r0 < M; r0 += RPW) {
    const int r = r0 + sub;
    if (r < M && col_ok) {
      V acc;
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
      size_t o2 = 0;
      for (int c = 0; c < ch.n; ++c) {
        const float* tile = reinterpret_cast<const float*>(smem + o2);
        const int2* ecv = reinterpret_cast<const int2*>(smem + o2 + (((size_t)K * ds * 4 + 7) & ~(size_t)7));
        const int* rp = reinterpret_cast<const int*>(ecv + ch.max_nnz[c]);
        const int s = rp[r], e = rp[r + 1];
        for (int k = s; k < e; ++k) {
          const int2 p = ecv[k];
          const float v = __int_as_float(p.y);
          const V x = ldv(tile + (size_t)p.x * ds + cl * VEC);
          acc += v * x;
        }
        if (self_scale && c == 0) acc += sscale * ldv(tile + (size_t)r * ds + cl * VEC);
        o2 += tile_chan_bytes(M, K, ds, ch.max_nnz[c]);
      }
      float* o = ob + (long)r * out_ld + cl * VEC;
      if (beta != 0.f) acc += ldv(o);
      if (act != KGCN_ACT_NONE) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = act_fwd(acc[j], act);
      }
      stv(o, acc);
    }
  }
  if constexpr (DOT) {
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) dot_acc += __shfl_xor(dot_acc, o2, 64);
    if constexpr (NW == 1) {
      if (lane == 0) dot_part[blockIdx.x] = dot_acc;
    } else {
      __syncthreads();                                  // every wave is done with the tiles: reuse the first floats
      float* red = reinterpret_cast<float*>(smem);
      if ((lane & 63) == 0) red[lane >> 6] = dot_acc;
      __syncthreads();
      if (lane == 0) {
        float s2 = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) s2 += red[w2];
        dot_part[blockIdx.x] = s2;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The 32-column slices of ONE graph as the waves of ONE workgroup (round 5).  Since round 3 a graph of
// 17..32 nodes is aggregated in 32-column slices (the kernel lives off occupancy); as separate one-wave workgroups every
// slice read the graph's CSR for itself: PMC traffic 1.920 GB for 1.7316 GB of algorithmic bytes (x1.11) on BASELINE config 2,
// and scalar bookkeeping as heavy as the arithmetic (SQ_INSTS_SALU 7.1e7 against SQ_INSTS_VALU 7.8e7 per launch: the generic
// kernel walks (row, vector) pairs with run-time strides).  Here wave w stages and aggregates slice w, the CSR slice is staged
// ONCE by all waves, sizes are compile-time (ds = 32: a row of a slice = 8 lanes x 16 bytes, 8 rows per pass), and the
// occupancy is unchanged (NS waves and NS tiles + one CSR copy per workgroup).  One channel; the activation epilogue, the
// act' prologue and the GIN self term as in spmm_tile_kernel.
// ------------------------------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(64 * NS) void spmm_slices_kernel(const int* __restrict__ rowptr, const int2* __restrict__ cv, int max_nnz,
                                                              const float* __restrict__ rhs, long rhs_ld, long rhs_gs,
                                                              float* __restrict__ out, long out_ld, long out_gs, int M, int K,
                                                              float beta, const float* __restrict__ self_scale, int act,
                                                              const float* __restrict__ aout, int dact) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int DS = 32;                               // columns per slice
  const int t = blockIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int tile_floats = K * DS;
  float* tile = reinterpret_cast<float*>(smem) + wv * tile_floats;
  int2* ecv = reinterpret_cast<int2*>(smem + (size_t)NS * tile_floats * 4);
  int* rp = reinterpret_cast<int*>(ecv + max_nnz);
  const int* grp = rowptr + (long)t * M;
  const int base = grp[0];
  const int cnt = grp[M] - base;
  const int col0 = wv * DS;
  // ---- this wave's slice of the rhs block: row i >> 3, 16-byte chunk i & 7 (128-byte row segments) ----------------------------
  const float* rb = rhs + (long)t * rhs_gs + col0 + (lane >> 3) * rhs_ld + (lane & 7) * 4;
  const int nv = K * 8;
  if (dact == KGCN_ACT_NONE) {
    for (int i = lane; i < nv; i += 256) {                  // four loads in flight per lane (K = 32: the whole tile)
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i + 64 * u < nv) v[u] = ld4(rb + (long)(8 * u + (i >> 3) - (lane >> 3)) * rhs_ld);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i + 64 * u < nv) st4(tile + (size_t)(i + 64 * u) * 4, v[u]);
    }
  } else {
    const float* ab = aout + (long)t * rhs_gs + col0 + (lane >> 3) * rhs_ld + (lane & 7) * 4;     // same layout as the gradient
    for (int i = lane; i < nv; i += 256) {
      f32x4 v[4], a[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i + 64 * u < nv) {
          v[u] = ld4(rb + (long)(8 * u + (i >> 3) - (lane >> 3)) * rhs_ld);
          a[u] = ld4(ab + (long)(8 * u + (i >> 3) - (lane >> 3)) * rhs_ld);
        }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i + 64 * u < nv) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[u][j] *= act_dout(a[u][j], dact);
          st4(tile + (size_t)(i + 64 * u) * 4, v[u]);
        }
    }
  }
  // ---- the graph's CSR slice, once for all slices ---------------------------------------------------------------------------
  for (int i = threadIdx.x; i < cnt; i += 64 * NS) ecv[i] = cv[base + i];
  for (int i = threadIdx.x; i <= M; i += 64 * NS) rp[i] = grp[i] - base;
  __syncthreads();
  // ---- aggregate: 8 rows per pass, 8 lanes x float4 per row ----------------------------------------------------------------------
  const int sub = lane >> 3, cl4 = (lane & 7) * 4;
  const float sscale = self_scale ? self_scale[0] : 0.f;
  float* ob = out + (long)t * out_gs + col0 + cl4;
  for (int r = sub; r < M; r += 8) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int s0 = rp[r], e0 = rp[r + 1];
    for (int k = s0; k < e0; ++k) {
      const int2 p = ecv[k];
      acc += __int_as_float(p.y) * ld4(tile + (size_t)p.x * DS + cl4);
    }
    if (self_scale) acc += sscale * ld4(tile + (size_t)r * DS + cl4);
    float* o = ob + (long)r * out_ld;
    if (beta != 0.f) acc += ld4(o);
    if (act != KGCN_ACT_NONE) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = act_fwd(acc[j], act);
    }
    st4(o, acc);
  }
}

// ------------------------------------------------------------------------------------------------
// row-chunk kernel: big matrices whose rhs block does not fit LDS -- the block-diagonal [sum N x sum N] batch of the
// kgcn-sparse path (kgcn/data_util.py:698-845) and the ragged-compact batches (ragged.hip), VEC in {4, 2}.
// The generic gather kernel below walks one row per lane group with THREE dependent loads per entry (rowptr -> cv ->
// rhs row): latency-bound (0.42 of HBM peak at d = 256, 0.12 at d = 50, profiles/r03a).  Here a group of LPR lanes owns
// ROWS = 8 consecutive rows: one coalesced load brings their 9 row offsets, one more the (column, value) pairs of all 8
// rows (a sliding LPR-entry window held across the group's lanes, handed out with ds_bpermute), and the rhs rows of up
// to four entries are requested back to back before the first is consumed -- two latency steps per 8 rows instead of
// three per entry.  Neighbour rows of a block-diagonal matrix lie within one graph (<= N rows away): consecutive row
// chunks are mapped to the SAME XCD (blockIdx -> chunk remap below), so those re-reads hit that XCD's L2.
// ------------------------------------------------------------------------------------------------
// (graph, local row) of global row `row` without a 64-bit division per row: one 32-bit division per 8-row chunk
struct RowWalk {
  int t, r;
  __device__ __forceinline__ RowWalk(long row0, int M, bool one_graph) {
    if (one_graph) { t = 0; r = (int)row0; }
    else { t = (int)(row0 / M); r = (int)(row0 - (long)t * M); }
  }
  __device__ __forceinline__ void step(int M) { if (++r >= M) { r = 0; ++t; } }
};

// LPR == 64: a WAVE owns the 8-row chunk, so row offsets, entry indices, columns and values are wave-uniform: they are
// pulled into SGPRs with v_readlane (no ds_bpermute), loop control and address bases run on the scalar unit, and entries
// past the end of a row are skipped by scalar branches.
// ROWS: consecutive rows per lane group: 8 for big matrices (amortises the two index loads), 2 for small ones (a matrix of a
// few thousand rows -- the 128-molecule batch of the kgcn-sparse path -- is latency-bound: 8 sequential rows per wave on
// 140 workgroups took 16 us for 4.5 MB; with 2 rows per group four times as many waves share the latency)
template <int VEC, int LPR, bool DACT, int ROWS>
__global__ __launch_bounds__(256) void spmm_rows_kernel(
    SpmmChannels ch, const float* __restrict__ rhs, int rhs_ld, long rhs_gs, float* __restrict__ out, int out_ld,
    long out_gs, int M, long total_rows, int d, float beta, const float* __restrict__ self_scale, int act,
    const float* __restrict__ aout, int dact, int blocks_per_xcd) {
  using V = typename SpVec<VEC>::T;
  constexpr int GPB = 256 / LPR;
  constexpr bool WAVE = LPR == 64;
  auto ldv = [](const float* p) { return *reinterpret_cast<const V*>(p); };
  auto stv = [](float* p, V v) { *reinterpret_cast<V*>(p) = v; };
  // XCD-aware: workgroups are dealt round-robin to the 8 XCDs; logical chunk = (xcd, position inside the xcd)
  const long lb = (long)(blockIdx.x & 7) * blocks_per_xcd + (blockIdx.x >> 3);
  const int gl = threadIdx.x % LPR, grp = threadIdx.x / LPR;
  const long r0 = (lb * GPB + grp) * ROWS;
  if (r0 >= total_rows) return;                      // whole lane group leaves together
  const bool one_graph = (long)M == total_rows;
  const float sscale = self_scale ? self_scale[0] : 0.f;
  auto bcast = [](int v, int src) __attribute__((always_inline)) {
    if constexpr (WAVE) return __builtin_amdgcn_readlane(v, src);
    else return __shfl(v, src, LPR);
  };
  for (int c0 = gl * VEC; c0 - gl * VEC < d; c0 += LPR * VEC) {     // one trip unless d > LPR * VEC; group-uniform count
    const bool ok = c0 < d;
    V acc[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i)
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[i][j] = 0.f;
    for (int c = 0; c < ch.n; ++c) {
      const int* rp = ch.rowptr[c];
      const int2* cv = ch.cv[c];
      const long ri = r0 + (gl <= ROWS ? gl : ROWS);
      const int my = rp[ri < total_rows ? ri : total_rows];          // rows past the end: empty
      const int e_all = bcast(my, ROWS);
      int wbase = bcast(my, 0);
      int2 w = (wbase + gl < e_all) ? cv[wbase + gl] : make_int2(0, 0);
      const float* rc = rhs + c * ch.rhs_cs + c0;
      const float* ac = aout + c0;
      RowWalk rw(r0, M, one_graph);
      static_for<ROWS>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int s = bcast(my, i), e = bcast(my, i + 1);
        const float* rb = rc + rw.t * rhs_gs;
        const float* ab = ac + rw.t * rhs_gs;
        for (int k = s; k < e; k += 4) {
          const int last = (k + 4 < e ? k + 4 : e);
          if (last > wbase + LPR) {                   // slide the window (group-uniform)
            wbase = k;
            w = (wbase + gl < e_all) ? cv[wbase + gl] : make_int2(0, 0);
          }
          V x[4], a[4];
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) { x[j][q] = 0.f; a[j][q] = 0.f; }
            v[j] = 0.f;
            if constexpr (WAVE) {
              if (k + j < e) {                        // scalar branch
                const int col = __builtin_amdgcn_readlane(w.x, k + j - wbase);
                v[j] = __int_as_float(__builtin_amdgcn_readlane(w.y, k + j - wbase));
                if (ok) {
                  x[j] = ldv(rb + (long)col * rhs_ld);
                  if constexpr (DACT) a[j] = ldv(ab + (long)col * rhs_ld);
                }
              }
            } else {
              const bool valid = k + j < e;
              const int idx = (k + j - wbase) & (LPR - 1);
              const int col = __shfl(w.x, idx, LPR);
              const float vv = __int_as_float(__shfl(w.y, idx, LPR));
              if (valid) v[j] = vv;
              if (valid && ok) {
                x[j] = ldv(rb + (long)col * rhs_ld);
                if constexpr (DACT) a[j] = ldv(ab + (long)col * rhs_ld);
              }
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if constexpr (DACT) {
#pragma unroll
              for (int q = 0; q < VEC; ++q) x[j][q] *= act_dout(a[j][q], dact);
            }
            acc[i] += v[j] * x[j];
          }
        }
        if (self_scale && c == 0 && ok && r0 + i < total_rows) acc[i] += sscale * ldv(rb + (long)rw.r * rhs_ld);
        rw.step(M);
      });
    }
    RowWalk rw(r0, M, one_graph);
    static_for<ROWS>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if (ok && r0 + i < total_rows) {
        float* o = out + rw.t * out_gs + (long)rw.r * out_ld + c0;
        V y = acc[i];
        if (beta != 0.f) y += ldv(o);
        if (act != KGCN_ACT_NONE) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) y[j] = act_fwd(y[j], act);
        }
        stv(o, y);
      }
      rw.step(M);
    });
  }
}

// ------------------------------------------------------------------------------------------------
// Block kernel: the ragged-compact batches (ONE block-diagonal matrix of whole molecules, ragged.hip) with their block
// structure (kgcn_csr_batch.block_ptr): a workgroup stages the rhs rows of one row block -- the molecules starting inside
// 64 consecutive rows, <= 64 + N - 1 rows -- times act'(aout) in a backward launch, its CSR entries and row offsets in LDS
// ONCE, then every row gathers from LDS.  The row-chunk kernel above re-reads a row from L2 / HBM once per neighbour and,
// in a backward launch, the saved activations as well: 617 MB for 361 MB of work at 117,888 x 256 (x 1.71; forward x 1.37,
// profiles/r03_n_cfg4_rocprof.txt); here every rhs byte is requested once.
// blockIdx.x = block * nslices + slice; a slice is ds columns (one 128 / 256-byte segment per row), so the slices of a block
// land on different XCDs at the same time and share the DRAM pages of its rows.
// Entries whose column lies outside the staged rows (never in a batch ragged.hip built) and rows beyond the LDS capacity
// (blocks longer than rows_cap) are gathered from memory.
// ------------------------------------------------------------------------------------------------
// One workgroup per (block, slice).  The requests go out in dependency order: the block's extent (two scalar loads), then its rhs
// rows -- the bulk -- and row offsets at once, and the (column, value) pairs, whose address needs a row offset, while those are in
// flight.  Tried and dropped (profiles/r04_spmm_block.txt): several items per workgroup with the next item's rows prefetched into
// registers behind the aggregation of the current one -- 100 / 135 us instead of 68 / 83 (forward / adjoint at d = 256): with the
// registers that costs only two workgroups fit a CU, and the aggregation itself (two dependent LDS reads per entry), not the
// load latency, is what a workgroup spends its time in.
constexpr int SPB_MAXE = 3;              // (column, value) pairs per lane on their way to LDS: ecap <= 768

template <int VEC, int LPR, bool DACT>
__global__ __launch_bounds__(256) void spmm_block_kernel(
    const int* __restrict__ rowptr, const int2* __restrict__ cv, const int* __restrict__ block_ptr, int nitems, int nslices,
    int ds, int rows_cap, int ecap, const float* __restrict__ rhs, long rhs_ld, float* __restrict__ out, long out_ld,
    float beta, const float* __restrict__ self_scale, int act, const float* __restrict__ aout, int dact) {
  using V = typename SpVec<VEC>::T;
  constexpr int MAXV = VEC == 4 ? 8 : 12;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* tile = reinterpret_cast<float*>(smem);
  int2* ecv = reinterpret_cast<int2*>(smem + (((size_t)rows_cap * ds * 4 + 15) & ~(size_t)15));
  int* rp = reinterpret_cast<int*>(ecv + ecap);
  auto ldv = [](const float* p) { return *reinterpret_cast<const V*>(p); };
  auto stv = [](float* p, V v) { *reinterpret_cast<V*>(p) = v; };
  const int tid = threadIdx.x;
  const int it = blockIdx.x;                           // item = (block, slice)
  const int k = it / nslices;
  const int col0 = (it - k * nslices) * ds;
  const int row_lo = block_ptr[k], n = block_ptr[k + 1] - row_lo;      // uniform: scalar loads
  if (n <= 0) return;
  const int nt = n < rows_cap ? n : rows_cap;          // rows of the block that live in LDS
  const int dv = ds / VEC;
  const int step_r = 256 / dv, step_c = 256 - step_r * dv;
  const int r_first = tid / dv, c_first = tid - r_first * dv;
  auto value = [&](long row, int col) __attribute__((always_inline)) {    // rhs (.) act'(aout) at (row, col)
    const long o = row * rhs_ld + col;
    V v = ldv(rhs + o);
    if constexpr (DACT) {
      const V a = ldv(aout + o);
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[j] *= act_dout(a[j], dact);
    }
    return v;
  };
  V pv[MAXV], pa[DACT ? MAXV : 1];
  int2 pe[SPB_MAXE];
  int pr = 0;
  // ---- stage: requests into registers first (rows and row offsets, then the entries), LDS writes after -----------------------
  const int nv = nt * dv;
  {
    int r = r_first, cc = c_first;
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
      // (lanes past the tile's end repeat its last vector: no predicated register, whose undefined half cost the forward
      // kernel 120 registers of copies)
      const bool in = tid + 256 * u < nv;
      const long o = (long)(row_lo + (in ? r : nt - 1)) * rhs_ld + col0 + (in ? cc : dv - 1) * VEC;
      pv[u] = ldv(rhs + o);
      if constexpr (DACT) pa[u] = ldv(aout + o);
      r += step_r; cc += step_c;
      if (cc >= dv) { cc -= dv; ++r; }
    }
  }
  if (tid <= nt) pr = rowptr[row_lo + tid];
  const int e_lo = rowptr[row_lo], cnt = rowptr[row_lo + n] - e_lo;    // uniform: scalar loads, beside the vector requests
  const bool ecv_lds = cnt <= ecap;
  if (ecv_lds) {
#pragma unroll
    for (int u = 0; u < SPB_MAXE; ++u)
      if (tid + 256 * u < cnt) pe[u] = cv[e_lo + tid + 256 * u];
  }
#pragma unroll
  for (int u = 0; u < MAXV; ++u) {
    if (tid + 256 * u < nv) {
      V v = pv[u];
      if constexpr (DACT) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] *= act_dout(pa[u][j], dact);
      }
      stv(tile + (size_t)(tid + 256 * u) * VEC, v);
    }
  }
  if (nv > MAXV * 256) {                               // what the registers do not hold: straight from memory
    int i = tid + MAXV * 256;
    int r = i / dv, cc = i - r * dv;
#pragma unroll 1
    for (; i < nv; i += 256) {
      stv(tile + (size_t)i * VEC, value(row_lo + r, col0 + cc * VEC));
      r += step_r; cc += step_c;
      if (cc >= dv) { cc -= dv; ++r; }
    }
  }
  if (ecv_lds) {
#pragma unroll
    for (int u = 0; u < SPB_MAXE; ++u)
      if (tid + 256 * u < cnt) ecv[tid + 256 * u] = pe[u];
  }
  if (tid <= nt) rp[tid] = pr - e_lo;
  __syncthreads();

  constexpr int RPW = 256 / LPR;
  const int sub = tid / LPR, cl = tid % LPR;
  const float sscale = self_scale ? self_scale[0] : 0.f;
  const int c = cl * VEC;
  auto finish = [&](V acc, long row, int col) __attribute__((always_inline)) {
    float* o = out + row * out_ld + col;
    if (beta != 0.f) acc += ldv(o);
    if (act != KGCN_ACT_NONE) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] = act_fwd(acc[j], act);
    }
    stv(o, acc);
  };
  // a row entirely from memory, entries in stored order (the same additions as the LDS path): rows beyond the LDS capacity,
  // rows with an entry that leaves the staged block, items with more entries than the LDS buffer holds
  auto slow_row = [&](long row, int col) __attribute__((always_inline)) {
    V acc;
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    const int s = rowptr[row], e = rowptr[row + 1];
#pragma unroll 1
    for (int q = s; q < e; ++q) {
      const int2 p = cv[q];
      acc += __int_as_float(p.y) * value(p.x, col);
    }
    if (self_scale) acc += sscale * value(row, col);
    finish(acc, row, col);
  };
  // aggregate out of LDS: 256 / LPR rows at a time, two rows per lane group and four entries per row in flight (the dependent
  // pair of LDS reads per entry -- (column, value), then the row -- is what a lane group waits for)
  {
    if (c >= ds) return;
    if (!ecv_lds) {                                                    // item-uniform
#pragma unroll 1
      for (int r = sub; r < n; r += RPW) slow_row(row_lo + r, col0 + c);
      return;
    }
#pragma unroll 1
    for (int r0 = sub; r0 < nt; r0 += 2 * RPW) {
      int s[2], len[2];
      V acc[2];
      bool left[2] = {false, false};                                   // an entry of the row leaves the staged block
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int r = r0 + u * RPW;
        s[u] = 0; len[u] = 0;
        if (r < nt) { s[u] = rp[r]; len[u] = rp[r + 1] - s[u]; }
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[u][j] = 0.f;
      }
      const int lmax = len[0] > len[1] ? len[0] : len[1];
#pragma unroll 1
      for (int q = 0; q < lmax; q += 4) {
        int2 p[2][4];
        V xv[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            p[u][j] = make_int2(row_lo, 0);
            if (q + j < len[u]) p[u][j] = ecv[s[u] + q + j];
          }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            unsigned lc = (unsigned)(p[u][j].x - row_lo);
            if (lc >= (unsigned)nt) { left[u] = true; lc = 0; }
            xv[u][j] = ldv(tile + (size_t)lc * ds + c);
          }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            // past the row's end: value 0 times the block's first row -- a finite row contributes exactly 0; a row with a
            // non-finite entry there must not: select instead of multiplying
            const float v = __int_as_float(p[u][j].y);
            const V t = acc[u] + v * xv[u][j];
            if (q + j < len[u]) acc[u] = t;
          }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int r = r0 + u * RPW;
        if (r >= nt) continue;
        if (left[u]) { slow_row(row_lo + r, col0 + c); continue; }
        if (self_scale) acc[u] += sscale * ldv(tile + (size_t)r * ds + c);
        finish(acc[u], row_lo + r, col0 + c);
      }
    }
#pragma unroll 1
    for (int r = nt + sub; r < n; r += RPW) slow_row(row_lo + r, col0 + c);
  }
}

// ------------------------------------------------------------------------------------------------
// generic CSR-vector gather kernel: LPR = 2^lpr_log2 lanes per row, VEC floats per lane
// ------------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void spmm_gather_kernel(
    SpmmChannels ch, const float* __restrict__ rhs, long rhs_ld, long rhs_gs, float* __restrict__ out, long out_ld,
    long out_gs, int M, long total_rows, int d, int lpr_log2, float beta, const float* __restrict__ self_scale, int act,
    const float* __restrict__ aout, int dact) {
  const int lpr = 1 << lpr_log2;
  const int cl = threadIdx.x & (lpr - 1);
  const long nworkers = ((long)gridDim.x * 256) >> lpr_log2;
  const float sscale = self_scale ? self_scale[0] : 0.f;
  for (long row = ((long)blockIdx.x * 256 + threadIdx.x) >> lpr_log2; row < total_rows;
       row += nworkers) {
    const long t = row / M;
    const int r = (int)(row - t * M);
    float* o = out + t * out_gs + (long)r * out_ld;
    for (int c0 = cl * VEC; c0 < d; c0 += lpr * VEC) {
      float acc[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
      for (int c = 0; c < ch.n; ++c) {
        const int s = ch.rowptr[c][row], e = ch.rowptr[c][row + 1];
        const float* rb = rhs + c * ch.rhs_cs + t * rhs_gs;
        const int2* cv = ch.cv[c];
        for (int k = s; k < e; ++k) {
          const int2 p = cv[k];
          const float v = __int_as_float(p.y);
          const float* src = rb + (long)p.x * rhs_ld + c0;
          if constexpr (VEC == 4) {
            f32x4 x = ld4(src);
            if (dact != KGCN_ACT_NONE) {
              const f32x4 a = ld4(aout + t * rhs_gs + (long)p.x * rhs_ld + c0);
#pragma unroll
              for (int j = 0; j < 4; ++j) x[j] *= act_dout(a[j], dact);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += v * x[j];
          } else {
            float x = src[0];
            if (dact != KGCN_ACT_NONE) x *= act_dout(aout[t * rhs_gs + (long)p.x * rhs_ld + c0], dact);
            acc[0] += v * x;
          }
        }
        if (self_scale && c == 0) {
          const float* src = rb + (long)r * rhs_ld + c0;
#pragma unroll
          for (int j = 0; j < VEC; ++j) acc[j] += sscale * src[j];
        }
      }
      if (beta != 0.f) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += o[c0 + j];
      }
      if (act != KGCN_ACT_NONE) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = act_fwd(acc[j], act);
      }
      if constexpr (VEC == 4) {
        f32x4 v4 = {acc[0], acc[1], acc[2], acc[3]};
        st4(o + c0, v4);
      } else {
        o[c0] = acc[0];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// d values[e] = <grad[row_e, :], rhs[col_e, :]>   (kgcn/bspmm_call.py:50-55)
// LPR lanes cooperate on one row's entries; butterfly reduction inside the lane group.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spmm_values_grad_kernel(
    const int* __restrict__ rowptr, const int2* __restrict__ cv, const float* __restrict__ grad,
    long g_ld, long g_gs, const float* __restrict__ rhs, long rhs_ld, long rhs_gs,
    float* __restrict__ dval, int M, long total_rows, int d, int lpr_log2) {
  const int lpr = 1 << lpr_log2;
  const int cl = threadIdx.x & (lpr - 1);
  const long nworkers = ((long)gridDim.x * 256) >> lpr_log2;
  // all lanes of a group run the same trip counts, so the shuffles below are convergent
  for (long row = ((long)blockIdx.x * 256 + threadIdx.x) >> lpr_log2; row < total_rows;
       row += nworkers) {
    const long t = row / M;
    const int r = (int)(row - t * M);
    const int s = rowptr[row], e = rowptr[row + 1];
    const float* g = grad + t * g_gs + (long)r * g_ld;
    const float* rb = rhs + t * rhs_gs;
    for (int k = s; k < e; ++k) {
      const int2 p = cv[k];
      const float* src = rb + (long)p.x * rhs_ld;
      float part = 0.f;
      for (int c = cl; c < d; c += lpr) part += g[c] * src[c];
      for (int off = lpr >> 1; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
      if (cl == 0) dval[k] = part;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// GraphMaxPooling (kgcn/layers.py:122-150): the segmented-max twin of the gather kernel.
//   fwd:    out[row,k] (+)= max( {a_e * x[col_e,k]} , 0 if the row stores fewer than K entries )
//   count:  m[row,k] = that maximum, inv[row,k] = 1 / #maximal elements of the densified row
//   bwd:    dx[j,k] (+)= sum over entries (i,j) of A^T's row j:  a * [a*x[j,k] == m[i,k]] * g[i,k]*inv[i,k]
// ------------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(
    const int* __restrict__ rowptr, const int2* __restrict__ cv, const float* __restrict__ x,
    float* __restrict__ out, float* __restrict__ mm, float* __restrict__ inv, int M, int K,
    long total_rows, int d, int lpr_log2, float beta) {
  const int lpr = 1 << lpr_log2;
  const int cl = threadIdx.x & (lpr - 1);
  const long nworkers = ((long)gridDim.x * 256) >> lpr_log2;
  for (long row = ((long)blockIdx.x * 256 + threadIdx.x) >> lpr_log2; row < total_rows;
       row += nworkers) {
    const long t = row / M;
    const int s = rowptr[row], e = rowptr[row + 1];
    const float* xb = x + t * (long)K * d;
    const bool has_zero = (e - s) < K;            // the densified row contains implicit zeros
    for (int c = cl; c < d; c += lpr) {
      float m = has_zero ? 0.f : -INFINITY;
      for (int k = s; k < e; ++k) {
        const int2 p = cv[k];
        m = fmaxf(m, __int_as_float(p.y) * xb[(long)p.x * d + c]);
      }
      if constexpr (COUNT) {
        int cnt = (has_zero && m == 0.f) ? K - (e - s) : 0;
        for (int k = s; k < e; ++k) {
          const int2 p = cv[k];
          cnt += (__int_as_float(p.y) * xb[(long)p.x * d + c] == m) ? 1 : 0;
        }
        mm[row * d + c] = m;
        inv[row * d + c] = 1.0f / (float)cnt;
      } else {
        float* o = out + row * d + c;
        *o = (beta != 0.f ? *o : 0.f) + m;
      }
    }
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(
    const int* __restrict__ rowptr_t, const int2* __restrict__ cv_t, const float* __restrict__ x,
    const float* __restrict__ g, const float* __restrict__ mm, const float* __restrict__ inv,
    float* __restrict__ dx, int M, int K, long total_cols, int d, int lpr_log2, float beta) {
  // rows of A^T = columns j of A: K per graph; entries (j, i) carry a_ij
  const int lpr = 1 << lpr_log2;
  const int cl = threadIdx.x & (lpr - 1);
  const long nworkers = ((long)gridDim.x * 256) >> lpr_log2;
  for (long col = ((long)blockIdx.x * 256 + threadIdx.x) >> lpr_log2; col < total_cols;
       col += nworkers) {
    const long t = col / K;
    const int s = rowptr_t[col], e = rowptr_t[col + 1];
    const float* xr = x + col * d;                 // x[t][j,:]
    const long rbase = t * (long)M;
    for (int c = cl; c < d; c += lpr) {
      const float xv = xr[c];
      float acc = 0.f;
      for (int k = s; k < e; ++k) {
        const int2 p = cv_t[k];                    // p.x = original row i
        const float av = __int_as_float(p.y);
        const long ri = (rbase + p.x) * d + c;
        if (av * xv == mm[ri]) acc += av * (g[ri] * inv[ri]);
      }
      float* o = dx + col * d + c;
      *o = (beta != 0.f ? *o : 0.f) + acc;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Multi-channel aggregation with a CHANNEL LOOP (round 6; kgcn/bconv_call.py:11-23, the split_adj_flag case of
// kgcn/data_util.py:76-122 -- one adjacency channel per bond type):   out[t] = act(beta out[t] + sum_c A_c[t] rhs_c[t]).
// spmm_tile_kernel stages ALL channels of a graph before it aggregates: with C = 6 channels of a 32 x 64 operand that is a 48 KB
// tile, three 4-wave workgroups per CU, 0.33 of the HBM peak (10 x 50: 0.23; tools/bconv_bench.py).  Here a ONE-wave workgroup
// stages one channel's [K x ds] block at a time -- the next channel's block and CSR slice are already on their way in registers --
// and the rows' sums stay in registers across the channels: lane (sub, cl) owns columns cl VEC .. of the rows p RPW + sub.  LDS per
// workgroup = one block + one CSR slice (8-10 KB: the occupancy of the single-channel kernel).
// bconv_fanout_kernel is the adjoint: out_c[t] = A_c[t] (g[t] (.) act'(aout[t])) for every channel c from ONE staging of g (the
// reference's addn_grad fans the gradient out to every channel, bconv_call.py:45-53; until now: one launch per channel, each of them
// reading g again).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void sp_wave_sync() {     // LDS hand-over inside ONE wave: program order is enough (see fused.hip)
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

template <int VEC, int NVL, int NP>
__global__ __launch_bounds__(64) void bconv_loop_kernel(
    SpmmChannels ch, const float* __restrict__ rhs, long rhs_ld, long rhs_gs, float* __restrict__ out, long out_ld, long out_gs,
    int M, int K, int ds, int nslices, float beta, int act) {
  using V = typename SpVec<VEC>::T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = blockIdx.x / nslices;
  const int col0 = (blockIdx.x - t * nslices) * ds;
  const int lane = threadIdx.x;
  const int dv = ds / VEC, nv = K * dv;
  const int step_r = 64 / dv, step_c = 64 - step_r * dv;
  const int rpw = 64 / dv;                                   // rows per pass
  const int sub = lane / dv, cl = lane - sub * dv;
  const bool active = sub < rpw;
  int max_nnz = 0;
  for (int c = 0; c < ch.n; ++c) max_nnz = ch.max_nnz[c] > max_nnz ? ch.max_nnz[c] : max_nnz;
  float* tile = reinterpret_cast<float*>(smem);
  int2* ecv = reinterpret_cast<int2*>(smem + (((size_t)K * ds * 4 + 15) & ~(size_t)15));
  int* rp = reinterpret_cast<int*>(ecv + max_nnz);
  auto ldv = [](const float* p) { return *reinterpret_cast<const V*>(p); };

  V pre[NVL];                 // the next channel's block in flight
  int2 pe0, pe1;              // ... its first 128 CSR entries
  int prp, pbase, pcnt;       // ... its row pointers (lane <= M), first entry, entry count
  auto issue = [&](int c) __attribute__((always_inline)) {
    const float* rb = rhs + c * ch.rhs_cs + (long)t * rhs_gs + col0;
    int r = lane / dv, cc = lane - r * dv;
#pragma unroll
    for (int u = 0; u < NVL; ++u) {
      const bool ok = lane + 64 * u < nv;
      const int rr = ok ? r : 0, c2 = ok ? cc : 0;           // clamped: always a valid address
      pre[u] = ldv(rb + (long)rr * rhs_ld + c2 * VEC);
      r += step_r; cc += step_c;
      if (cc >= dv) { cc -= dv; ++r; }
    }
    const int* grp = ch.rowptr[c] + (long)t * M;
    pbase = grp[0];
    pcnt = grp[M] - pbase;
    prp = grp[lane <= M ? lane : M] - pbase;
    const int2 z = {0, 0};
    pe0 = lane < pcnt ? ch.cv[c][pbase + lane] : z;
    pe1 = 64 + lane < pcnt ? ch.cv[c][pbase + 64 + lane] : z;
  };
  V acc[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[p][j] = 0.f;

  issue(0);
  for (int c = 0; c < ch.n; ++c) {
    // ---- registers -> LDS (the aggregation of channel c - 1 is through with the block: same wave, program order) ----
#pragma unroll
    for (int u = 0; u < NVL; ++u)
      if (lane + 64 * u < nv) *reinterpret_cast<V*>(tile + (size_t)(lane + 64 * u) * VEC) = pre[u];
    const int cnt = pcnt, base = pbase;
    if (lane < cnt) ecv[lane] = pe0;
    if (64 + lane < cnt) ecv[64 + lane] = pe1;
    for (int i = 128 + lane; i < cnt; i += 64) ecv[i] = ch.cv[c][base + i];      // rare: more than 128 entries in one channel
    if (lane <= M) rp[lane] = prp;
    if (M >= 64 && lane == 0) rp[M] = cnt;
    if (c + 1 < ch.n) issue(c + 1);                          // uniform
    sp_wave_sync();
    // ---- aggregate channel c into the rows' sums ----
    if (active) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int r = p * rpw + sub;
        if (r < M) {
          const int s0 = rp[r], e0 = rp[r + 1];
          for (int k = s0; k < e0; ++k) {
            const int2 q = ecv[k];
            acc[p] += __int_as_float(q.y) * ldv(tile + (size_t)q.x * ds + cl * VEC);
          }
        }
      }
    }
    sp_wave_sync();
  }
  if (active) {
    float* ob = out + (long)t * out_gs + col0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int r = p * rpw + sub;
      if (r < M) {
        float* o = ob + (long)r * out_ld + cl * VEC;
        V a = acc[p];
        if (beta != 0.f) a += ldv(o);
        if (act != KGCN_ACT_NONE) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) a[j] = act_fwd(a[j], act);
        }
        *reinterpret_cast<V*>(o) = a;
      }
    }
  }
}

// ch: the TRANSPOSED containers (rows of A_c^T = columns of A_c); M = their rows, K = their columns (= rows of g's block)
template <int VEC, int NVL>
__global__ __launch_bounds__(64) void bconv_fanout_kernel(
    SpmmChannels ch, const float* __restrict__ g, long g_ld, long g_gs, const float* __restrict__ aout, int dact,
    float* __restrict__ out, long out_ld, long out_gs, long out_cs, int M, int K, int ds, int nslices) {
  using V = typename SpVec<VEC>::T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = blockIdx.x / nslices;
  const int col0 = (blockIdx.x - t * nslices) * ds;
  const int lane = threadIdx.x;
  const int dv = ds / VEC, nv = K * dv;
  const int step_r = 64 / dv, step_c = 64 - step_r * dv;
  const int rpw = 64 / dv;
  const int sub = lane / dv, cl = lane - sub * dv;
  const bool active = sub < rpw;
  int max_nnz = 0;
  for (int c = 0; c < ch.n; ++c) max_nnz = ch.max_nnz[c] > max_nnz ? ch.max_nnz[c] : max_nnz;
  float* tile = reinterpret_cast<float*>(smem);
  int2* ecv = reinterpret_cast<int2*>(smem + (((size_t)K * ds * 4 + 15) & ~(size_t)15));
  int* rp = reinterpret_cast<int*>(ecv + max_nnz);
  auto ldv = [](const float* p) { return *reinterpret_cast<const V*>(p); };

  int2 pe0, pe1;
  int prp, pbase, pcnt;
  auto issue_csr = [&](int c) __attribute__((always_inline)) {
    const int* grp = ch.rowptr[c] + (long)t * M;
    pbase = grp[0];
    pcnt = grp[M] - pbase;
    prp = grp[lane <= M ? lane : M] - pbase;
    const int2 z = {0, 0};
    pe0 = lane < pcnt ? ch.cv[c][pbase + lane] : z;
    pe1 = 64 + lane < pcnt ? ch.cv[c][pbase + 64 + lane] : z;
  };
  issue_csr(0);
  {                                                          // the gradient block, once (times act'(aout) in an activated layer)
    const float* gb = g + (long)t * g_gs + col0;
    const float* ab = aout ? aout + (long)t * g_gs + col0 : nullptr;
    V v[NVL], a[NVL];
    int r = lane / dv, cc = lane - r * dv;
#pragma unroll
    for (int u = 0; u < NVL; ++u) {
      const bool ok = lane + 64 * u < nv;
      const int rr = ok ? r : 0, c2 = ok ? cc : 0;
      v[u] = ldv(gb + (long)rr * g_ld + c2 * VEC);
      if (dact != KGCN_ACT_NONE) a[u] = ldv(ab + (long)rr * g_ld + c2 * VEC);
      r += step_r; cc += step_c;
      if (cc >= dv) { cc -= dv; ++r; }
    }
#pragma unroll
    for (int u = 0; u < NVL; ++u) {
      if (dact != KGCN_ACT_NONE) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[u][j] *= act_dout(a[u][j], dact);
      }
      if (lane + 64 * u < nv) *reinterpret_cast<V*>(tile + (size_t)(lane + 64 * u) * VEC) = v[u];
    }
  }
  for (int c = 0; c < ch.n; ++c) {
    const int cnt = pcnt, base = pbase;
    if (lane < cnt) ecv[lane] = pe0;
    if (64 + lane < cnt) ecv[64 + lane] = pe1;
    for (int i = 128 + lane; i < cnt; i += 64) ecv[i] = ch.cv[c][base + i];
    if (lane <= M) rp[lane] = prp;
    if (M >= 64 && lane == 0) rp[M] = cnt;
    if (c + 1 < ch.n) issue_csr(c + 1);
    sp_wave_sync();
    if (active) {
      float* ob = out + c * out_cs + (long)t * out_gs + col0;
      for (int r = sub; r < M; r += rpw) {
        V acc;
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
        const int s0 = rp[r], e0 = rp[r + 1];
        for (int k = s0; k < e0; ++k) {
          const int2 q = ecv[k];
          acc += __int_as_float(q.y) * ldv(tile + (size_t)q.x * ds + cl * VEC);
        }
        *reinterpret_cast<V*>(ob + (long)r * out_ld + cl * VEC) = acc;
      }
    }
    sp_wave_sync();
  }
}

// ------------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------------
static int ilog2_ceil(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

// Tile kernel is used when one graph's working set (all channels of the launch) leaves >= 8 waves per CU resident.
// Can the LDS-staged kernel take this launch, and how?  vec = floats per lane (4, or 2 for even widths), slices = column
// slices of d/slices floats each (one workgroup per graph and slice) so that a tile stays within 20 KiB (8 waves / CU).
struct TilePlan { bool ok; int vec; int slices; int nw; };
static TilePlan tile_plan(const kgcn_csr_batch* a, int nch, const float* rhs, long rhs_ld, long rhs_gs, long rhs_cs,
                          int d, const float* out, long out_ld, long out_gs, const float* aout, bool whole_rows = false) {
  TilePlan p = {false, 4, 1, 1};
  if (d <= 0 || d > 1024 || a->rows <= 0 || a->cols <= 0) return p;
  const long all = rhs_ld | rhs_gs | out_ld | out_gs | rhs_cs | d;
  const uintptr_t ptrs = reinterpret_cast<uintptr_t>(rhs) | reinterpret_cast<uintptr_t>(out) |
                         reinterpret_cast<uintptr_t>(aout);
  if (all % 4 == 0 && ptrs % 16 == 0) p.vec = 4;
  else if (all % 2 == 0 && ptrs % 8 == 0) p.vec = 2;
  else return p;
  // This kernel lives off occupancy (a wave per graph runs load -> LDS -> gather -> store with nothing overlapped inside it): for
  // 17..32-node graphs 32-column slices (128-byte row segments, the CSR read once per slice) put 31 instead of 17 / 9 workgroups on
  // a CU: d = 64: 0.62 -> 0.645 of the HBM peak, d = 128: 0.56 -> 0.64.  Not a general rule: 10-, 16- and 50-node graphs and
  // 256-wide operands lose 4-25 % with narrower slices (profiles/r03_i_spmm_slices.txt).
  const int sl0 = (!whole_rows && nch == 1 && p.vec == 4 && a->rows > 16 && a->rows <= 32 && (d == 64 || d == 128)) ? d / 32 : 1;
  for (int sl = sl0; sl <= 16; sl *= 2) {
    if (d % sl != 0 || (d / sl) % p.vec != 0 || (sl > 1 && d / sl < 32)) break;
    if (d / sl > 64 * p.vec) continue;                  // one wave covers a row of the slice
    size_t lds = 0;
    for (int c = 0; c < nch; ++c) lds += tile_chan_bytes(a[c].rows, a[c].cols, d / sl, a[c].max_nnz_per_graph);
    if (lds <= 20 * 1024 || (sl == 1 && lds <= 53 * 1024)) {
      p.ok = true;
      p.slices = sl;
      p.nw = lds <= 20 * 1024 ? 1 : 4;
      return p;
    }
  }
  return p;
}

// Channel-loop kernels (bconv_loop_kernel / bconv_fanout_kernel): vec and the column slice ds of one wave (<= 64 vectors per row,
// the block of ONE channel within 16 KiB, <= 16 block vectors and <= 16 row passes per lane); 0 = not this route.
struct LoopPlan { int vec, ds, nvl, np; size_t lds; };
static LoopPlan loop_plan(const kgcn_csr_batch* a, int nch, long ld_all, uintptr_t ptrs, int d) {
  LoopPlan p = {0, 0, 0, 0, 0};
  const int M = a->rows, K = a->cols;
  if (M <= 0 || K <= 0 || M > 64 || K > 64 || d <= 0) return p;
  const int vec = (ld_all % 4 == 0 && ptrs % 16 == 0) ? 4 : (ld_all % 2 == 0 && ptrs % 8 == 0) ? 2 : 0;
  if (!vec) return p;
  int ds = d;
  while (ds / vec > 64 || (long)K * ds * 4 > 16 * 1024) {     // halve while the halves stay whole vectors of >= 32 columns
    if (ds % 2 != 0 || (ds / 2) % vec != 0 || ds / 2 < 32) return p;
    ds /= 2;
  }
  if (d % ds != 0) return p;
  const int dv = ds / vec, nv = K * dv, rpw = 64 / dv;
  const int nvl = (nv + 63) / 64, np = (M + rpw - 1) / rpw;
  if (nvl > 16 || np > 16) return p;
  int max_nnz = 0;
  for (int c = 0; c < nch; ++c) max_nnz = a[c].max_nnz_per_graph > max_nnz ? a[c].max_nnz_per_graph : max_nnz;
  p.vec = vec; p.ds = ds; p.nvl = nvl <= 8 ? 8 : 16; p.np = np <= 8 ? 8 : 16;
  p.lds = (((size_t)K * ds * 4 + 15) & ~(size_t)15) + (size_t)max_nnz * 8 + (size_t)(M + 2) * 4;
  if (p.lds > 40 * 1024) p.vec = 0;
  return p;
}

// out[t] = act(beta*out[t] + sum_c A_c[t] @ (rhs_c[t] (.) act'(aout[t])));  a: nch channel descriptors of one batch shape
int launch_spmm_multi(const kgcn_csr_batch* a, int nch, const float* rhs, long rhs_ld, long rhs_gs, long rhs_cs, int d,
                      float* out, long out_ld, long out_gs, float beta, const float* self_scale, int act,
                      const float* aout, int dact, hipStream_t stream, const float* dotx = nullptr,
                      float* dot_part = nullptr, bool* dot_done = nullptr) {
  const int T = a->num_graphs, M = a->rows, K = a->cols;
  if (T == 0 || M == 0 || d == 0) return 0;
  if (nch > MAX_CH) {                       // more channels than one launch takes: groups of MAX_CH, accumulate
    for (int c0 = 0; c0 < nch; c0 += MAX_CH) {
      const int n = nch - c0 < MAX_CH ? nch - c0 : MAX_CH;
      const bool last = c0 + n >= nch;      // the activation belongs to the last group only
      int rc = launch_spmm_multi(a + c0, n, rhs + c0 * rhs_cs, rhs_ld, rhs_gs, rhs_cs, d, out, out_ld, out_gs,
                                 c0 == 0 ? beta : 1.f, c0 == 0 ? self_scale : nullptr, last ? act : KGCN_ACT_NONE, aout,
                                 dact, stream);
      if (rc) return rc;
    }
    return 0;
  }
  SpmmChannels ch;
  ch.n = nch;
  ch.rhs_cs = rhs_cs;
  for (int c = 0; c < nch; ++c) {
    ch.rowptr[c] = a[c].rowptr;
    ch.cv[c] = reinterpret_cast<const int2*>(a[c].cv);
    ch.max_nnz[c] = a[c].max_nnz_per_graph;
  }
#ifndef KGCN_BCONV_NO_LOOP
  if (nch >= 2 && dact == KGCN_ACT_NONE && !self_scale && !dotx) {
    // several channels: one channel's block in LDS at a time, the sums in registers (bconv_loop_kernel)
    const LoopPlan lp = loop_plan(a, nch, rhs_ld | rhs_gs | out_ld | out_gs | rhs_cs | d,
                                  reinterpret_cast<uintptr_t>(rhs) | reinterpret_cast<uintptr_t>(out), d);
    if (lp.vec && (long)T * (d / lp.ds) <= 0x7fffffffL) {
      const int nsl = d / lp.ds;
      const dim3 grid((unsigned)(T * nsl));
#define KGCN_LOOP(VEC, NVL, NP)                                                                                            \
  hipLaunchKernelGGL((bconv_loop_kernel<VEC, NVL, NP>), grid, dim3(64), lp.lds, stream, ch, rhs, rhs_ld, rhs_gs, out, out_ld, \
                     out_gs, M, K, lp.ds, nsl, beta, act)
      if (lp.vec == 4) {
        if (lp.nvl == 8 && lp.np == 8) KGCN_LOOP(4, 8, 8); else if (lp.nvl == 8) KGCN_LOOP(4, 8, 16);
        else if (lp.np == 8) KGCN_LOOP(4, 16, 8); else KGCN_LOOP(4, 16, 16);
      } else {
        if (lp.nvl == 8 && lp.np == 8) KGCN_LOOP(2, 8, 8); else if (lp.nvl == 8) KGCN_LOOP(2, 8, 16);
        else if (lp.np == 8) KGCN_LOOP(2, 16, 8); else KGCN_LOOP(2, 16, 16);
      }
#undef KGCN_LOOP
      return check_launch("bconv_loop_kernel");
    }
  }
#endif
  // ragged-compact batches with their block structure: every rhs row staged once (spmm_block_kernel)
  if (nch == 1 && T == 1 && a->block_ptr && a->num_blocks > 0 && a->block_rows_max > 0 && !dotx) {
    const long all = rhs_ld | out_ld | d;
    const uintptr_t ptrs = reinterpret_cast<uintptr_t>(rhs) | reinterpret_cast<uintptr_t>(out) |
                           (dact ? reinterpret_cast<uintptr_t>(aout) : 0);
    const int vec = (all % 4 == 0 && ptrs % 16 == 0) ? 4 : (all % 2 == 0 && ptrs % 8 == 0) ? 2 : 0;
    const int rows_cap = a->block_rows_max < 255 ? a->block_rows_max : 255;      // (one row offset per lane)
    // slice width: whole rows while the tile stays within 24 KiB (>= 6 workgroups per CU), else 64 / 32 columns
    int ds = 0;
    if (vec) {
      if ((long)rows_cap * d * 4 <= 24 * 1024 && d / vec <= 64) ds = d;
#ifndef SPB_SLICE32
      else if (vec == 4 && d % 64 == 0 && (long)rows_cap * 64 * 4 <= 40 * 1024) ds = 64;
#endif
      else if (vec == 4 && d % 32 == 0 && (long)rows_cap * 32 * 4 <= 40 * 1024) ds = 32;
    }
    if (ds && (long)a->num_blocks * (d / ds) <= 0x7fffffffL) {
      const int nslices = d / ds, dv = ds / vec;
      int ecap = rows_cap * 6;
      if (ecap > SPB_MAXE * 256) ecap = SPB_MAXE * 256;
      const size_t lds = (((size_t)rows_cap * ds * 4 + 15) & ~(size_t)15) + (size_t)ecap * 8 +
                         (size_t)(rows_cap + 1) * 4;
      const int nitems = a->num_blocks * nslices;
      const dim3 grid((unsigned)nitems);
#define KGCN_BLK2(VEC, LPR, DACT)                                                                                       \
  hipLaunchKernelGGL((spmm_block_kernel<VEC, LPR, DACT>), grid, dim3(256), lds, stream, a->rowptr,                       \
                     reinterpret_cast<const int2*>(a->cv), a->block_ptr, nitems, nslices, ds, rows_cap, ecap, rhs, rhs_ld, out, \
                     out_ld, beta, self_scale, act, aout, dact)
#define KGCN_BLK(VEC, LPR)                                                                                              \
  {                                                                                                                     \
    if (dact != KGCN_ACT_NONE) KGCN_BLK2(VEC, LPR, true); else KGCN_BLK2(VEC, LPR, false);                              \
  }
      if (vec == 4) {
        if (dv <= 8) KGCN_BLK(4, 8) else if (dv <= 16) KGCN_BLK(4, 16) else if (dv <= 32) KGCN_BLK(4, 32) else KGCN_BLK(4, 64)
      } else {
        if (dv <= 8) KGCN_BLK(2, 8) else if (dv <= 16) KGCN_BLK(2, 16) else if (dv <= 32) KGCN_BLK(2, 32) else KGCN_BLK(2, 64)
      }
#undef KGCN_BLK
#undef KGCN_BLK2
      return check_launch("spmm_block_kernel");
    }
  }
  const TilePlan plan = tile_plan(a, nch, rhs, rhs_ld, rhs_gs, rhs_cs, d, out, out_ld, out_gs, dact ? aout : nullptr,
                                  dotx != nullptr);      // the fused <rhs, dotx> needs a graph's whole rows in one workgroup
  if (plan.ok && (long)T * plan.slices <= 0x7fffffffL) {
    const int ds = d / plan.slices;
#ifndef KGCN_SPMM_NO_SLICE_WAVES
    if (nch == 1 && plan.vec == 4 && ds == 32 && plan.nw == 1 && (plan.slices == 2 || plan.slices == 4) && !dotx) {
      // the slices of a graph as the waves of one workgroup: its CSR staged once (spmm_slices_kernel)
      const size_t lds2 = (size_t)plan.slices * K * 32 * 4 + (size_t)a->max_nnz_per_graph * 8 + (size_t)(M + 1) * 4;
      const int2* cvp = reinterpret_cast<const int2*>(a->cv);
      // the tile plan budgets its LDS per SLICE workgroup; this kernel holds all slices of the graph in one: beyond the
      // 64 KiB a launch gets without an attribute (K around 121..150 at d = 128) the slice-per-workgroup kernel runs instead
      if (lds2 > 64 * 1024) goto tile_route;
      if (plan.slices == 2)
        hipLaunchKernelGGL(spmm_slices_kernel<2>, dim3((unsigned)T), dim3(128), lds2, stream, a->rowptr, cvp, a->max_nnz_per_graph, rhs,
                           rhs_ld, rhs_gs, out, out_ld, out_gs, M, K, beta, self_scale, act, aout, dact);
      else
        hipLaunchKernelGGL(spmm_slices_kernel<4>, dim3((unsigned)T), dim3(256), lds2, stream, a->rowptr, cvp, a->max_nnz_per_graph, rhs,
                           rhs_ld, rhs_gs, out, out_ld, out_gs, M, K, beta, self_scale, act, aout, dact);
      return check_launch("spmm_slices_kernel");
    }
  tile_route:
#endif
    size_t lds = 0;
    for (int c = 0; c < nch; ++c) lds += tile_chan_bytes(M, K, ds, a[c].max_nnz_per_graph);
    const int lanes = ds / plan.vec;
    const dim3 grid((unsigned)(T * plan.slices));
    if (dotx && dot_part && dot_done && plan.vec == 4 && plan.slices == 1 && rhs_ld == ds && dact == KGCN_ACT_NONE) {
      // <rhs, dotx> of every graph rides in the staging of channel 0 (one partial per workgroup = per graph)
#define KGCN_TILE_DOT2(LPR, NW)                                                                                       \
  hipLaunchKernelGGL((spmm_tile_kernel<LPR, 4, NW, true>), grid, dim3(64 * NW), lds, stream, ch, rhs, rhs_ld, rhs_gs, \
                     out, out_ld, out_gs, M, K, ds, plan.slices, beta, self_scale, act, aout, dact, dotx, dot_part)
#define KGCN_TILE_DOT(LPR)                                                                                            \
  {                                                                                                                   \
    if (plan.nw == 1) KGCN_TILE_DOT2(LPR, 1);                                                                         \
    else {                                                                                                            \
      static thread_local bool big = false;                                                                           \
      if (!big) {                                                                                                     \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spmm_tile_kernel<LPR, 4, 4, true>),                   \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);                            \
        big = true;                                                                                                   \
      }                                                                                                               \
      KGCN_TILE_DOT2(LPR, 4);                                                                                         \
    }                                                                                                                 \
  }
      if (lanes <= 8) KGCN_TILE_DOT(8)
      else if (lanes <= 16) KGCN_TILE_DOT(16)
      else if (lanes <= 32) KGCN_TILE_DOT(32)
      else KGCN_TILE_DOT(64)
#undef KGCN_TILE_DOT
#undef KGCN_TILE_DOT2
      *dot_done = true;
      return check_launch("spmm_tile_kernel<dot>");
    }
#define KGCN_TILE2(LPR, VEC, NW)                                                                                      \
  hipLaunchKernelGGL((spmm_tile_kernel<LPR, VEC, NW>), grid, dim3(64 * NW), lds, stream, ch, rhs, rhs_ld, rhs_gs, out,   \
                     out_ld, out_gs, M, K, ds, plan.slices, beta, self_scale, act, aout, dact)
#define KGCN_TILE(LPR, VEC)                                                                                           \
  {                                                                                                                   \
    if (plan.nw == 1) KGCN_TILE2(LPR, VEC, 1);                                                                        \
    else {                                                                                                            \
      static thread_local bool big = false;                                                                           \
      if (!big) {                                                                                                     \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spmm_tile_kernel<LPR, VEC, 4>),                       \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);                            \
        big = true;                                                                                                   \
      }                                                                                                               \
      KGCN_TILE2(LPR, VEC, 4);                                                                                        \
    }                                                                                                                 \
  }
    if (plan.vec == 4) {
      if (lanes <= 8) KGCN_TILE(8, 4)
      else if (lanes <= 16) KGCN_TILE(16, 4)
      else if (lanes <= 32) KGCN_TILE(32, 4)
      else KGCN_TILE(64, 4)
    } else {
      if (lanes <= 8) KGCN_TILE(8, 2)
      else if (lanes <= 16) KGCN_TILE(16, 2)
      else if (lanes <= 32) KGCN_TILE(32, 2)
      else KGCN_TILE(64, 2)
    }
#undef KGCN_TILE2
#undef KGCN_TILE
    return check_launch("spmm_tile_kernel");
  }
  const long total_rows = (long)T * M;
  {
    // rows that do not fit an LDS tile: the row-chunk kernel when 8- or 16-byte vectors are possible
    const long all = rhs_ld | rhs_gs | out_ld | out_gs | rhs_cs | d;
    const uintptr_t ptrs = reinterpret_cast<uintptr_t>(rhs) | reinterpret_cast<uintptr_t>(out) |
                           (dact ? reinterpret_cast<uintptr_t>(aout) : 0);
    const int vec = (all % 4 == 0 && ptrs % 16 == 0) ? 4 : ((all % 2 == 0 && ptrs % 8 == 0) ? 2 : 0);
    if (vec && total_rows < (1L << 31) && rhs_ld < (1L << 31) && out_ld < (1L << 31)) {
      const int lanes = d / vec;
      const int lpr = lanes <= 16 ? 16 : (lanes <= 32 ? 32 : 64);
      const int rows_per_group = total_rows >= 32768 ? 8 : 2;
      const long rows_per_block = (long)(256 / lpr) * rows_per_group;
      const long nblocks = (total_rows + rows_per_block - 1) / rows_per_block;
      const int per_xcd = (int)((nblocks + 7) / 8);
      const dim3 grid((unsigned)(per_xcd * 8));
#define KGCN_ROWS3(VEC, LPR, DACT, R)                                                                                 \
  hipLaunchKernelGGL((spmm_rows_kernel<VEC, LPR, DACT, R>), grid, dim3(256), 0, stream, ch, rhs, (int)rhs_ld, rhs_gs, \
                     out, (int)out_ld, out_gs, M, total_rows, d, beta, self_scale, act, aout, dact, per_xcd)
#define KGCN_ROWS2(VEC, LPR, DACT)                                                                                    \
  {                                                                                                                   \
    if (rows_per_group == 8) KGCN_ROWS3(VEC, LPR, DACT, 8);                                                           \
    else KGCN_ROWS3(VEC, LPR, DACT, 2);                                                                               \
  }
#define KGCN_ROWS(VEC, LPR)                                                                                           \
  {                                                                                                                   \
    if (dact != KGCN_ACT_NONE) KGCN_ROWS2(VEC, LPR, true)                                                             \
    else KGCN_ROWS2(VEC, LPR, false)                                                                                  \
  }
      if (vec == 4) {
        if (lpr == 16) KGCN_ROWS(4, 16) else if (lpr == 32) KGCN_ROWS(4, 32) else KGCN_ROWS(4, 64)
      } else {
        if (lpr == 16) KGCN_ROWS(2, 16) else if (lpr == 32) KGCN_ROWS(2, 32) else KGCN_ROWS(2, 64)
      }
#undef KGCN_ROWS
#undef KGCN_ROWS2
#undef KGCN_ROWS3
      return check_launch("spmm_rows_kernel");
    }
  }
  const bool vec4 = (d % 4 == 0) && (rhs_ld % 4 == 0) && (rhs_gs % 4 == 0) && (out_ld % 4 == 0) &&
                    (out_gs % 4 == 0) && (rhs_cs % 4 == 0) && aligned16(rhs) && aligned16(out) &&
                    (dact == KGCN_ACT_NONE || aligned16(aout));
  const int per_row = vec4 ? d / 4 : d;
  int lpr_log2 = ilog2_ceil(per_row);
  if (lpr_log2 > 6) lpr_log2 = 6;
  long blocks = ((total_rows << lpr_log2) + 255) / 256;
  const long max_blocks = (long)kNumCU * 64;
  if (blocks > max_blocks) blocks = max_blocks;
  if (blocks < 1) blocks = 1;
  if (vec4)
    hipLaunchKernelGGL((spmm_gather_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, stream, ch, rhs, rhs_ld, rhs_gs,
                       out, out_ld, out_gs, M, total_rows, d, lpr_log2, beta, self_scale, act, aout, dact);
  else
    hipLaunchKernelGGL((spmm_gather_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, stream, ch, rhs, rhs_ld, rhs_gs,
                       out, out_ld, out_gs, M, total_rows, d, lpr_log2, beta, self_scale, act, aout, dact);
  return check_launch("spmm_gather_kernel");
}

int launch_spmm(const kgcn_csr_batch* a, const float* rhs, long rhs_ld, long rhs_gs, int d,
                float* out, long out_ld, long out_gs, float beta, const float* self_scale,
                hipStream_t stream) {
  return launch_spmm_multi(a, 1, rhs, rhs_ld, rhs_gs, 0, d, out, out_ld, out_gs, beta, self_scale, KGCN_ACT_NONE,
                           nullptr, KGCN_ACT_NONE, stream);
}

}  // namespace kgcn

using namespace kgcn;

extern "C" int kgcn_bspmm_f32(const kgcn_csr_batch* a, const float* rhs, int64_t rhs_ld,
                              int64_t rhs_graph_stride, int32_t d, float* out, int64_t out_ld,
                              int64_t out_graph_stride, float beta, void* stream) {
  if (int rc = validate_csr(a, "kgcn_bspmm_f32")) return rc;
  if (d < 0) return fail("kgcn_bspmm_f32: d=%d < 0", d);
  if (a->num_graphs == 0 || a->rows == 0 || d == 0) return 0;
  if (!rhs || !out) return fail("kgcn_bspmm_f32: rhs/out is NULL");
  if (rhs_ld < d || out_ld < d) return fail("kgcn_bspmm_f32: leading dimension smaller than d");
  if (beta != 0.f && beta != 1.f) return fail("kgcn_bspmm_f32: beta must be 0 or 1");
  return launch_spmm(a, rhs, rhs_ld, rhs_graph_stride, d, out, out_ld, out_graph_stride, beta,
                     nullptr, as_stream(stream));
}

extern "C" int kgcn_bconv_f32(const kgcn_csr_batch* a_ch, int32_t num_channels, const float* rhs,
                              int64_t rhs_ld, int64_t rhs_graph_stride,
                              int64_t rhs_channel_stride, int32_t d, float* out, int64_t out_ld,
                              int64_t out_graph_stride, void* stream) {
  if (num_channels <= 0) return fail("kgcn_bconv_f32: num_channels=%d", num_channels);
  if (!a_ch) return fail("kgcn_bconv_f32: a_ch is NULL");
  for (int c = 0; c < num_channels; ++c) {
    if (int rc = validate_csr(a_ch + c, "kgcn_bconv_f32")) return rc;
    if (a_ch[c].num_graphs != a_ch[0].num_graphs || a_ch[c].rows != a_ch[0].rows ||
        a_ch[c].cols != a_ch[0].cols)
      return fail("kgcn_bconv_f32: channel %d has a different batch shape", c);
  }
  if (a_ch[0].num_graphs == 0 || a_ch[0].rows == 0 || d == 0) return 0;
  if (!rhs || !out) return fail("kgcn_bconv_f32: rhs/out is NULL");
  if (rhs_ld < d || out_ld < d) return fail("kgcn_bconv_f32: leading dimension smaller than d");
  // channel add-n (tf.add_n, kgcn/layers.py:115) inside the kernel: every output row is written once
  return launch_spmm_multi(a_ch, num_channels, rhs, rhs_ld, rhs_graph_stride, rhs_channel_stride, d, out, out_ld,
                           out_graph_stride, 0.f, nullptr, KGCN_ACT_NONE, nullptr, KGCN_ACT_NONE, as_stream(stream));
}

static int check_act(const char* who, int act) {
  if (act < KGCN_ACT_NONE || act > KGCN_ACT_TANH) return fail("%s: unknown activation code %d", who, act);
  return 0;
}

extern "C" int kgcn_bconv_fanout_f32(const kgcn_csr_batch* at_ch, int32_t num_channels, const float* grad, const float* act_out,
                                     int64_t ld, int64_t graph_stride, int32_t d, int32_t act, float* out, int64_t out_ld,
                                     int64_t out_graph_stride, int64_t out_channel_stride, void* stream) {
  if (num_channels <= 0) return fail("kgcn_bconv_fanout_f32: num_channels=%d", num_channels);
  if (!at_ch) return fail("kgcn_bconv_fanout_f32: at_ch is NULL");
  if (int rc = check_act("kgcn_bconv_fanout_f32", act)) return rc;
  for (int c = 0; c < num_channels; ++c) {
    if (int rc = validate_csr(at_ch + c, "kgcn_bconv_fanout_f32")) return rc;
    if (at_ch[c].num_graphs != at_ch[0].num_graphs || at_ch[c].rows != at_ch[0].rows || at_ch[c].cols != at_ch[0].cols)
      return fail("kgcn_bconv_fanout_f32: channel %d has a different batch shape", c);
  }
  if (d < 0) return fail("kgcn_bconv_fanout_f32: d=%d < 0", d);
  const int T = at_ch[0].num_graphs, M = at_ch[0].rows, K = at_ch[0].cols;
  if (T == 0 || M == 0 || d == 0) return 0;
  if (!grad || !out || (act != KGCN_ACT_NONE && !act_out)) return fail("kgcn_bconv_fanout_f32: NULL operand");
  if (ld < d || out_ld < d) return fail("kgcn_bconv_fanout_f32: leading dimension smaller than d");
  hipStream_t s = as_stream(stream);
  const uintptr_t ptrs = reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(out) |
                         (act != KGCN_ACT_NONE ? reinterpret_cast<uintptr_t>(act_out) : 0);
  const LoopPlan lp = num_channels <= MAX_CH
      ? loop_plan(at_ch, num_channels, ld | graph_stride | out_ld | out_graph_stride | out_channel_stride | d, ptrs, d)
      : LoopPlan{0, 0, 0, 0, 0};
  if (lp.vec && (long)T * (d / lp.ds) <= 0x7fffffffL) {
    SpmmChannels ch;
    ch.n = num_channels;
    ch.rhs_cs = 0;
    for (int c = 0; c < num_channels; ++c) {
      ch.rowptr[c] = at_ch[c].rowptr;
      ch.cv[c] = reinterpret_cast<const int2*>(at_ch[c].cv);
      ch.max_nnz[c] = at_ch[c].max_nnz_per_graph;
    }
    const int nsl = d / lp.ds;
    const dim3 grid((unsigned)(T * nsl));
    const float* ao = act != KGCN_ACT_NONE ? act_out : nullptr;
#define KGCN_FAN(VEC, NVL)                                                                                                 \
  hipLaunchKernelGGL((bconv_fanout_kernel<VEC, NVL>), grid, dim3(64), lp.lds, s, ch, grad, (long)ld, (long)graph_stride, ao, \
                     (int)act, out, (long)out_ld, (long)out_graph_stride, (long)out_channel_stride, M, K, lp.ds, nsl)
    if (lp.vec == 4) { if (lp.nvl == 8) KGCN_FAN(4, 8); else KGCN_FAN(4, 16); }
    else { if (lp.nvl == 8) KGCN_FAN(2, 8); else KGCN_FAN(2, 16); }
#undef KGCN_FAN
    return check_launch("bconv_fanout_kernel");
  }
  for (int c = 0; c < num_channels; ++c) {                  // shapes the channel-loop kernel does not take: one launch per channel
    int rc = launch_spmm_multi(at_ch + c, 1, grad, ld, graph_stride, 0, d, out + c * out_channel_stride, out_ld, out_graph_stride,
                               0.f, nullptr, KGCN_ACT_NONE, act_out, act, s);
    if (rc) return rc;
  }
  return 0;
}

extern "C" int kgcn_bconv_act_f32(const kgcn_csr_batch* a_ch, int32_t num_channels, const float* rhs, int64_t rhs_ld,
                                  int64_t rhs_graph_stride, int64_t rhs_channel_stride, int32_t d, float* out,
                                  int64_t out_ld, int64_t out_graph_stride, int32_t act, void* stream) {
  if (num_channels <= 0) return fail("kgcn_bconv_act_f32: num_channels=%d", num_channels);
  if (!a_ch) return fail("kgcn_bconv_act_f32: a_ch is NULL");
  if (int rc = check_act("kgcn_bconv_act_f32", act)) return rc;
  for (int c = 0; c < num_channels; ++c) {
    if (int rc = validate_csr(a_ch + c, "kgcn_bconv_act_f32")) return rc;
    if (a_ch[c].num_graphs != a_ch[0].num_graphs || a_ch[c].rows != a_ch[0].rows || a_ch[c].cols != a_ch[0].cols)
      return fail("kgcn_bconv_act_f32: channel %d has a different batch shape", c);
  }
  if (a_ch[0].num_graphs == 0 || a_ch[0].rows == 0 || d == 0) return 0;
  if (!rhs || !out) return fail("kgcn_bconv_act_f32: rhs/out is NULL");
  if (rhs_ld < d || out_ld < d) return fail("kgcn_bconv_act_f32: leading dimension smaller than d");
  return launch_spmm_multi(a_ch, num_channels, rhs, rhs_ld, rhs_graph_stride, rhs_channel_stride, d, out, out_ld,
                           out_graph_stride, 0.f, nullptr, act, nullptr, KGCN_ACT_NONE, as_stream(stream));
}

extern "C" int kgcn_bspmm_dact_f32(const kgcn_csr_batch* a, const float* grad, const float* act_out, int64_t ld,
                                   int64_t graph_stride, int32_t d, int32_t act, float* out, int64_t out_ld,
                                   int64_t out_graph_stride, float beta, void* stream) {
  if (int rc = validate_csr(a, "kgcn_bspmm_dact_f32")) return rc;
  if (int rc = check_act("kgcn_bspmm_dact_f32", act)) return rc;
  if (d < 0) return fail("kgcn_bspmm_dact_f32: d=%d < 0", d);
  if (a->num_graphs == 0 || a->rows == 0 || d == 0) return 0;
  if (!grad || !out || (act != KGCN_ACT_NONE && !act_out)) return fail("kgcn_bspmm_dact_f32: NULL operand");
  if (ld < d || out_ld < d) return fail("kgcn_bspmm_dact_f32: leading dimension smaller than d");
  if (beta != 0.f && beta != 1.f) return fail("kgcn_bspmm_dact_f32: beta must be 0 or 1");
  return launch_spmm_multi(a, 1, grad, ld, graph_stride, 0, d, out, out_ld, out_graph_stride, beta, nullptr,
                           KGCN_ACT_NONE, act_out, act, as_stream(stream));
}

extern "C" int kgcn_gin_aggregate_f32(const kgcn_csr_batch* a_ch, int32_t num_channels,
                                      const float* x, int32_t d, const float* eps, float* out,
                                      void* stream) {
  if (num_channels <= 0) return fail("kgcn_gin_aggregate_f32: num_channels=%d", num_channels);
  if (!a_ch) return fail("kgcn_gin_aggregate_f32: a_ch is NULL");
  for (int c = 0; c < num_channels; ++c) {
    if (int rc = validate_csr(a_ch + c, "kgcn_gin_aggregate_f32")) return rc;
    if (a_ch[c].rows != a_ch[c].cols)
      return fail("kgcn_gin_aggregate_f32: adjacency must be square (M=%d K=%d)", a_ch[c].rows,
                  a_ch[c].cols);
    if (a_ch[c].num_graphs != a_ch[0].num_graphs || a_ch[c].rows != a_ch[0].rows)
      return fail("kgcn_gin_aggregate_f32: channel %d has a different batch shape", c);
  }
  if (a_ch[0].num_graphs == 0 || a_ch[0].rows == 0 || d == 0) return 0;
  if (!x || !out) return fail("kgcn_gin_aggregate_f32: x/out is NULL");
  const long gs = (long)a_ch[0].rows * d;
  for (int c = 0; c < num_channels; ++c) {
    int rc = launch_spmm(a_ch + c, x, d, gs, d, out, d, gs, c == 0 ? 0.f : 1.f,
                         eps ? eps + c : nullptr, as_stream(stream));
    if (rc) return rc;
  }
  return 0;
}

namespace kgcn {
__global__ void dot_final_kernel(const float* __restrict__ part, int nparts, float* __restrict__ out);
}

extern "C" int64_t kgcn_gin_aggregate_bwd_workspace_bytes(int32_t num_graphs, int32_t n_nodes, int32_t d) {
  const int64_t a = (int64_t)(num_graphs > 0 ? num_graphs : 1) * 4;
  const int64_t b = kgcn_dot_workspace_bytes((int64_t)num_graphs * n_nodes * d);
  return a > b ? a : b;
}

// Backward of GINAggregate (kgcn/layers.py:461-472) in one go: dx = sum_c (eps_c g + A_c^T g) and d eps = <g, x>.
extern "C" int kgcn_gin_aggregate_bwd_f32(const kgcn_csr_batch* at_ch, int32_t num_channels, const float* grad, int32_t d,
                                          const float* eps, const float* x, float* dx, float* deps, void* workspace,
                                          int64_t workspace_bytes, void* stream) {
  if (num_channels <= 0) return fail("kgcn_gin_aggregate_bwd_f32: num_channels=%d", num_channels);
  if (!at_ch) return fail("kgcn_gin_aggregate_bwd_f32: at_ch is NULL");
  for (int c = 0; c < num_channels; ++c) {
    if (int rc = validate_csr(at_ch + c, "kgcn_gin_aggregate_bwd_f32")) return rc;
    if (at_ch[c].rows != at_ch[c].cols) return fail("kgcn_gin_aggregate_bwd_f32: adjacency must be square");
    if (at_ch[c].num_graphs != at_ch[0].num_graphs || at_ch[c].rows != at_ch[0].rows)
      return fail("kgcn_gin_aggregate_bwd_f32: channel %d has a different batch shape", c);
  }
  hipStream_t s = as_stream(stream);
  const int T = at_ch[0].num_graphs, N = at_ch[0].rows;
  if (T == 0 || N == 0 || d == 0) {
    if (deps) (void)hipMemsetAsync(deps, 0, 4, s);
    return 0;
  }
  if (!grad || (!dx && !deps)) return fail("kgcn_gin_aggregate_bwd_f32: NULL operand");
  if (deps && (!x || !workspace || workspace_bytes < kgcn_gin_aggregate_bwd_workspace_bytes(T, N, d)))
    return fail("kgcn_gin_aggregate_bwd_f32: d eps needs x and a workspace of %lld bytes",
                (long long)kgcn_gin_aggregate_bwd_workspace_bytes(T, N, d));
  const long gs = (long)N * d;
  bool dot_done = false;
  if (dx) {
    for (int c = 0; c < num_channels; ++c) {
      const bool try_dot = deps && c == 0;
      int rc = launch_spmm_multi(at_ch + c, 1, grad, d, gs, 0, d, dx, d, gs, c == 0 ? 0.f : 1.f, eps ? eps + c : nullptr,
                                 KGCN_ACT_NONE, nullptr, KGCN_ACT_NONE, s, try_dot ? x : nullptr,
                                 try_dot ? static_cast<float*>(workspace) : nullptr, try_dot ? &dot_done : nullptr);
      if (rc) return rc;
    }
  }
  if (!deps) return 0;
  if (dot_done) {
    hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(256), 0, s, static_cast<const float*>(workspace), T, deps);
    return check_launch("dot_final_kernel");
  }
  return kgcn_dot_f32(grad, x, (int64_t)T * N * d, deps, workspace, workspace_bytes, stream);
}

extern "C" int kgcn_spmm_values_grad_f32(const kgcn_csr_batch* a, const float* grad,
                                         int64_t grad_ld, int64_t grad_graph_stride,
                                         const float* rhs, int64_t rhs_ld,
                                         int64_t rhs_graph_stride, int32_t d, float* dval,
                                         void* stream) {
  if (int rc = validate_csr(a, "kgcn_spmm_values_grad_f32")) return rc;
  if (a->nnz == 0 || a->num_graphs == 0 || a->rows == 0) return 0;
  if (!grad || !rhs || !dval) return fail("kgcn_spmm_values_grad_f32: NULL operand");
  if (d <= 0) return fail("kgcn_spmm_values_grad_f32: d=%d", d);
  int lpr_log2 = ilog2_ceil(d);
  if (lpr_log2 > 6) lpr_log2 = 6;
  const long total_rows = (long)a->num_graphs * a->rows;
  long blocks = ((total_rows << lpr_log2) + 255) / 256;
  const long max_blocks = (long)kNumCU * 64;
  if (blocks > max_blocks) blocks = max_blocks;
  hipLaunchKernelGGL(spmm_values_grad_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     as_stream(stream), a->rowptr, reinterpret_cast<const int2*>(a->cv), grad,
                     grad_ld, grad_graph_stride, rhs, rhs_ld, rhs_graph_stride, dval, a->rows,
                     total_rows, d, lpr_log2);
  return check_launch("spmm_values_grad_kernel");
}

extern "C" int kgcn_graph_maxpool_fwd_f32(const kgcn_csr_batch* a, const float* x, int32_t d,
                                          float* out, float beta, void* stream) {
  if (int rc = validate_csr(a, "kgcn_graph_maxpool_fwd_f32")) return rc;
  if (a->num_graphs == 0 || a->rows == 0 || d <= 0) return 0;
  if (!x || !out) return fail("kgcn_graph_maxpool_fwd_f32: NULL operand");
  if (beta != 0.f && beta != 1.f) return fail("kgcn_graph_maxpool_fwd_f32: beta must be 0 or 1");
  int lpr_log2 = ilog2_ceil(d);
  if (lpr_log2 > 6) lpr_log2 = 6;
  const long total_rows = (long)a->num_graphs * a->rows;
  long blocks = ((total_rows << lpr_log2) + 255) / 256;
  if (blocks > (long)kNumCU * 64) blocks = (long)kNumCU * 64;
  hipLaunchKernelGGL((maxpool_fwd_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                     a->rowptr, reinterpret_cast<const int2*>(a->cv), x, out, nullptr, nullptr, a->rows,
                     a->cols, total_rows, d, lpr_log2, beta);
  return check_launch("maxpool_fwd_kernel");
}

extern "C" int64_t kgcn_graph_maxpool_bwd_workspace_bytes(int32_t num_graphs, int32_t rows, int32_t d) {
  if (num_graphs <= 0 || rows <= 0 || d <= 0) return 0;
  return (int64_t)2 * num_graphs * rows * d * 4;
}

extern "C" int kgcn_graph_maxpool_bwd_f32(const kgcn_csr_batch* a, const kgcn_csr_batch* at,
                                          const float* x, const float* dout_grad, int32_t d, float* dx,
                                          float beta, void* workspace, int64_t workspace_bytes,
                                          void* stream) {
  if (int rc = validate_csr(a, "kgcn_graph_maxpool_bwd_f32")) return rc;
  if (int rc = validate_csr(at, "kgcn_graph_maxpool_bwd_f32")) return rc;
  if (at->num_graphs != a->num_graphs || at->rows != a->cols || at->cols != a->rows || at->nnz != a->nnz)
    return fail("kgcn_graph_maxpool_bwd_f32: `at` is not the transposed batch of `a`");
  if (a->num_graphs == 0 || a->rows == 0 || a->cols == 0 || d <= 0) return 0;
  if (!x || !dout_grad || !dx) return fail("kgcn_graph_maxpool_bwd_f32: NULL operand");
  if (beta != 0.f && beta != 1.f) return fail("kgcn_graph_maxpool_bwd_f32: beta must be 0 or 1");
  const int64_t need = kgcn_graph_maxpool_bwd_workspace_bytes(a->num_graphs, a->rows, d);
  if (!workspace || workspace_bytes < need)
    return fail("kgcn_graph_maxpool_bwd_f32: workspace %lld < %lld bytes", (long long)workspace_bytes,
                (long long)need);
  float* mm = static_cast<float*>(workspace);
  float* inv = mm + (long)a->num_graphs * a->rows * d;
  int lpr_log2 = ilog2_ceil(d);
  if (lpr_log2 > 6) lpr_log2 = 6;
  hipStream_t s = as_stream(stream);
  const long total_rows = (long)a->num_graphs * a->rows;
  long blocks = ((total_rows << lpr_log2) + 255) / 256;
  if (blocks > (long)kNumCU * 64) blocks = (long)kNumCU * 64;
  hipLaunchKernelGGL((maxpool_fwd_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, s, a->rowptr,
                     reinterpret_cast<const int2*>(a->cv), x, nullptr, mm, inv, a->rows, a->cols, total_rows,
                     d, lpr_log2, 0.f);
  if (int rc = check_launch("maxpool_fwd_kernel<count>")) return rc;
  const long total_cols = (long)at->num_graphs * at->rows;
  blocks = ((total_cols << lpr_log2) + 255) / 256;
  if (blocks > (long)kNumCU * 64) blocks = (long)kNumCU * 64;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, at->rowptr,
                     reinterpret_cast<const int2*>(at->cv), x, dout_grad, mm, inv, dx, a->rows, a->cols,
                     total_cols, d, lpr_log2, beta);
  return check_launch("maxpool_bwd_kernel");
}
