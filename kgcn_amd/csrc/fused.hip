// Fused GraphConv layer for small molecular graphs on gfx950:
//
//   forward   out[t] = A[t] @ (x[t] @ W + bias)                     kgcn/layers.py:64-116
//   backward  dfw[t] = A[t]^T @ g[t];  dx[t] = dfw[t] @ W^T;
//             dW = sum_t x[t]^T dfw[t];  dbias = sum_t colsum(dfw[t])   SURVEY 3.3, bspmm_call.py:45
//
// The reference materialises FW = X.W + b per graph and channel (B*C tiny MatMul ops) and then
// aggregates it with B*C sparse ops.  Here one persistent wave owns one graph at a time:
//   * the graph's node-feature tile [N<=32 x D<=64] is streamed HBM -> registers -> LDS once
//     (dwordx4, all loads of a tile in flight together), and the NEXT graph's tile + CSR slice are
//     already in flight (register prefetch, branch-free) while the current graph is computed,
//   * the dense contraction of the forward runs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32) with
//     the weight fragments resident in registers; the two contractions of the backward run on the bf16
//     matrix pipe with an exact 3-way split of the fp32 operands (six v_mfma_f32_32x32x16_bf16 products
//     per k-step, fp32-accurate -- see split_pair below),
//   * FW (resp. dFW) lives only in LDS, where the sparse aggregation gathers neighbour rows with
//     conflict-free ds_read_b128 -- X.W and dFW never touch HBM,
//   * dW / dbias accumulate in MFMA accumulators across ALL graphs a wave processes, are reduced
//     across the workgroup's waves through LDS and leave the chip once per workgroup
//     (deterministic second-stage reduction).
// Algorithmic HBM bytes per graph (N=32, D=64, nnz=100): forward 17,316, backward 25,508.
//
// Two instantiations of every kernel: FULL (N = 32, din = dout = 64: the benchmark shape; all
// sizes are compile-time constants, the 8 aggregation passes are straight-line code) and generic
// (N <= 32, din,dout <= 64; small graphs share a tile, widths that are not multiples of 4 take the flat-float4 or the
// element-wise tile path).  Other shapes use
// kgcn_dense_* + kgcn_bconv_f32.
#include <type_traits>

#include "kgcn_common.h"

namespace kgcn {

int launch_reduce_partials(const float* part, int nparts, long n, float* out, hipStream_t s);
int launch_reduce_pair(const float* part_dw, long n_dw, float* dw, const float* part_db, long n_db, float* dbias, int nparts,
                       hipStream_t s);
int launch_reduce_partials2(const float* part, int nparts, long n, float* out, const float* part2, long n2,
                            float* out2, hipStream_t s);

constexpr int FN = 32;    // node tile (MFMA M)
constexpr int FD = 64;    // feature tile (K of the forward GEMM, two 32-wide output tiles)
constexpr int ALD = 68;   // padded row stride (floats) of the forward's A-fragment tile (b128 reads)
constexpr int BLD = 65;   // ODD row stride of the backward's dFW tile: conflict-free ds_read_b32 for
                          // both MFMA operand patterns (down a column for dX, along a row for dW)
constexpr int MAX_WPB = 8;


// Phase probe (tools/phase_probe.py builds a second library with -DKGCN_PROBE): per-wave sums of
// s_memtime deltas per phase, written to a device buffer.  Compiled out of the product library.
#ifdef KGCN_PROBE
__device__ long long* g_probe = nullptr;
#define PROBE_DECL long long pt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long pc_ = __builtin_readcyclecounter();
#define PROBE(k) { const long long n_ = __builtin_readcyclecounter(); pt_[k] += n_ - pc_; pc_ = n_; }
#define PROBE_FLUSH(gw) if (g_probe && lane == 0) { for (int k_ = 0; k_ < 8; ++k_) g_probe[(long)(gw) * 8 + k_] = pt_[k_]; }
#else
#define PROBE_DECL
#define PROBE(k)
#define PROBE_FLUSH(gw)
#endif

__device__ __forceinline__ void wave_sync() {
  // Hand-off of LDS data between lanes of ONE wave.  The LDS unit executes a wave's DS
  // instructions in issue order, so no hardware wait is needed -- only the compiler must not move
  // LDS accesses across this point.  Deliberately NOT an atomic fence: a release/acquire fence
  // lowers to s_waitcnt vmcnt(0), which drains the next graph's prefetch loads the moment they
  // are issued.
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

typedef int i32x4 __attribute__((ext_vector_type(4)));   // native vector: HIP's int4 class type went
                                                        // through scratch memory in selects

__device__ __forceinline__ f32x4 ldv4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void stv4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// acc += v * x, component by component as plain v_fma_f32.  Beside f32 MFMAs the packed forms
// (v_pk_fma_f32 / v_pk_mul_f32, which hipcc forms from float4 arithmetic) cost ~22-26 extra cycles
// each (MI355X_MICROARCH.md, "price of one filler beside MFMAs"); the file is built with
// -fno-slp-vectorize so these stay scalar.
__device__ __forceinline__ void fma4(f32x4& acc, float v, const f32x4& x) {
  acc[0] = __builtin_fmaf(v, x[0], acc[0]);
  acc[1] = __builtin_fmaf(v, x[1], acc[1]);
  acc[2] = __builtin_fmaf(v, x[2], acc[2]);
  acc[3] = __builtin_fmaf(v, x[3], acc[3]);
}
__device__ __forceinline__ void add4(f32x4& acc, const f32x4& x) {
  acc[0] += x[0]; acc[1] += x[1]; acc[2] += x[2]; acc[3] += x[3];
}

// (split_pair / split8 / mfma_bf16 -- the exact 3-way bf16 split -- live in kgcn_common.h)
struct WaveSlice {
  float* a;     // forward: x tile [FN][ALD] (A-fragment source); backward: dFW tile [FN][BLD]
  float* b;     // [FN+1][FD] gather source tile; row FN (= KGCN_PAD_COL) stays zero
  int2* ecv;    // [max_nnz] (col,val) pairs of the graph, row-padded layout (x2 when double buffered)
  int* rp;      // [FN + 4] row pointers rebased to 0 (x2 when double buffered)
};

static_assert(KGCN_PAD_COL == FN, "padding entries must address the zero row of the gather tile");

__host__ __device__ inline size_t ecv_bytes(int max_nnz) { return ((size_t)max_nnz * 8 + 15) & ~(size_t)15; }
constexpr size_t RP_BYTES = (FN + 4) * 4;

// a_floats: FN*ALD (forward) or FN*BLD rounded up to 16 bytes (backward); ncsr: 1 or 2 CSR buffers
__host__ __device__ inline size_t slice_bytes(int max_nnz, int a_floats, int ncsr) {
  return (((size_t)a_floats * 4 + 15) & ~(size_t)15) + (size_t)(FN + 1) * FD * 4 +
         (size_t)ncsr * (ecv_bytes(max_nnz) + RP_BYTES);
}

__device__ __forceinline__ WaveSlice carve(unsigned char* base, int wave, int max_nnz, int a_floats,
                                           int ncsr) {
  unsigned char* p = base + (size_t)wave * slice_bytes(max_nnz, a_floats, ncsr);
  WaveSlice s;
  s.a = reinterpret_cast<float*>(p);
  s.b = reinterpret_cast<float*>(p + (((size_t)a_floats * 4 + 15) & ~(size_t)15));
  s.ecv = reinterpret_cast<int2*>(s.b + (FN + 1) * FD);
  s.rp = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(s.ecv) + (size_t)ncsr * ecv_bytes(max_nnz));
  return s;
}

// One feature tile in flight (<= 512 float4 = 8 per lane) / the first 128 CSR entries (2 per lane)
struct TileRegs { f32x4 v[8]; };
struct CsrRegs { i32x4 e0, e1; };   // lane l: entries 2l, 2l+1 and 128+2l, 128+2l+1 (plain members:
                                   // an array member ended up in scratch memory)

template <bool FULL>
__device__ __forceinline__ void issue_tile(TileRegs& f, const float* __restrict__ src, int n4,
                                           int lane) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int i = lane + q * 64;
    if constexpr (FULL) {
      f.v[q] = ldv4(src + (long)i * 4);
    } else {
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      f.v[q] = (i < n4) ? ldv4(src + (long)i * 4) : z;
    }
  }
}

template <bool FULL>
__device__ __forceinline__ void land_tile(const TileRegs& f, float* tile, int ld, int n4, int d4, int lane);

// Feature widths that are not multiples of 4 (the 50 and 3 of example_model/model.py:42-46): the tile is
// moved element by element, e = lane + 64 q over the N*d contiguous floats of the graph.
__device__ __forceinline__ void issue_tile_s(TileRegs& f, const float* __restrict__ src, int n, int lane) {
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int e = lane + q * 64;
    f.v[q >> 2][q & 3] = (e < n) ? src[e] : 0.f;
  }
}
__device__ __forceinline__ void land_tile_s(const TileRegs& f, float* tile, int ld, int n, int d, int lane) {
  int r = lane / d, c = lane - r * d;
  const int dr = 64 / d, dc = 64 - dr * d;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    if (lane + q * 64 < n) tile[r * ld + c] = f.v[q >> 2][q & 3];
    c += dc; r += dr;
    if (c >= d) { c -= d; ++r; }
  }
}
// Tile movement mode of the generic kernels:  2 = every width a multiple of 4 (float4 rows);  1 = widths are
// not, but every tile (rows * d floats, contiguous) is a whole number of aligned float4: the tile is LOADED
// as float4 (a quarter of the instructions) and scattered to its rows element by element;  0 = scalar.
__device__ __forceinline__ void land_tile_f4(const TileRegs& f, float* tile, int ld, int n4, int d, int lane) {
  int e = lane * 4;
  int r = e / d, c = e - r * d;
  const int dr = 256 / d, dc = 256 - dr * d;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    if (lane + q * 64 < n4) {
      int rr = r, cc = c;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        tile[rr * ld + cc] = f.v[q][j];
        if (++cc == d) { cc = 0; ++rr; }
      }
    }
    c += dc; r += dr;
    if (c >= d) { c -= d; ++r; }
  }
}
template <int MODE>
__device__ __forceinline__ void issue_tile_g(TileRegs& f, const float* __restrict__ src, int n, int lane) {
  if constexpr (MODE >= 1) issue_tile<false>(f, src, n >> 2, lane);
  else issue_tile_s(f, src, n, lane);
}
template <int MODE>
__device__ __forceinline__ void land_tile_g(const TileRegs& f, float* tile, int ld, int n, int d, int lane) {
  if constexpr (MODE == 2) land_tile<false>(f, tile, ld, n >> 2, d >> 2, lane);
  else if constexpr (MODE == 1) land_tile_f4(f, tile, ld, n >> 2, d, lane);
  else land_tile_s(f, tile, ld, n, d, lane);
}

// Row-padded layout: the graph's entry count is a multiple of 4 and its first entry index too, so
// entries can be moved two at a time as aligned 16-byte words.
__device__ __forceinline__ void issue_cv(CsrRegs& f, const int2* __restrict__ cv, int base, int cnt,
                                         int lane) {
  const i32x4 z = {0, 0, 0, 0};
  const i32x4* src = reinterpret_cast<const i32x4*>(cv + base);
  f.e0 = (2 * lane < cnt) ? src[lane] : z;
  f.e1 = (128 + 2 * lane < cnt) ? src[64 + lane] : z;
}

// registers -> LDS tile with row stride `ld` floats (d4 = float4 per source row)
template <bool FULL>
__device__ __forceinline__ void land_tile(const TileRegs& f, float* tile, int ld, int n4, int d4,
                                          int lane) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int i = lane + q * 64;
    if constexpr (FULL) {
      stv4(tile + (i >> 4) * ld + (i & 15) * 4, f.v[q]);
    } else {
      if (i < n4) {
        const int r = i / d4, c4 = i - r * d4;
        stv4(tile + r * ld + c4 * 4, f.v[q]);
      }
    }
  }
}

__device__ __forceinline__ void land_csr(const CsrRegs& f, int2* ecv, int* tab,
                                         const int2* __restrict__ cv, int slot_val, int base, int cnt,
                                         int N, int lane) {
  i32x4* dst = reinterpret_cast<i32x4*>(ecv);
  if (2 * lane < cnt) dst[lane] = f.e0;
  if (128 + 2 * lane < cnt) dst[64 + lane] = f.e1;
  for (int i = 256 + lane; i < cnt; i += 64) ecv[i] = cv[base + i];  // rare: > 256 entries
  if (lane < N) tab[lane] = slot_val;   // slot table of the graph (offsets are graph-local already)
}

// Per-graph metadata in flight: lane l < N holds slots[t*N + l]; `gp` holds graph_ptr[t + (l & 1)]
// (lane 0: first entry of the graph, lane 1: one past its last entry).
struct MetaRegs { int slot, gp; };
__device__ __forceinline__ void issue_meta(MetaRegs& m, const int* __restrict__ slots,
                                           const int* __restrict__ gptr, int t, int N, int lane) {
  m.slot = slots[(long)t * N + (lane < N ? lane : N - 1)];
  m.gp = gptr[t + (lane & 1)];
}
__device__ __forceinline__ int meta_base(const MetaRegs& m) { return __builtin_amdgcn_readlane(m.gp, 0); }
__device__ __forceinline__ int meta_cnt(const MetaRegs& m) {
  return __builtin_amdgcn_readlane(m.gp, 1) - __builtin_amdgcn_readlane(m.gp, 0);
}

// ---- graph packing (generic kernels) --------------------------------------------------------------------
// A graph of N <= 16 nodes leaves most of the 32-row tile (and of its MFMAs) empty, so P = min(4, 32 / N)
// CONSECUTIVE graphs share one tile: their features, CSR slices and slot tables are contiguous in the batch
// containers, i.e. the pack is read like one graph of P*N rows; only the graph-local column indices, row
// numbers and entry offsets are shifted when the slices are landed in LDS (padding entries keep col = FN).
constexpr int MAX_PACK = 4;
__device__ __forceinline__ void issue_meta_p(MetaRegs& m, const int* __restrict__ slots,
                                             const int* __restrict__ gptr, int t, int P, int N, int lane) {
  const int rows = P * N;
  m.slot = slots[(long)t * N + (lane < rows ? lane : rows - 1)];
  m.gp = gptr[t + (lane < P ? lane : P)];       // lane p <= P: first entry of graph t + p
}
__device__ __forceinline__ int meta_cnt_p(const MetaRegs& m, int P) {
  return __builtin_amdgcn_readlane(m.gp, P) - __builtin_amdgcn_readlane(m.gp, 0);
}
__device__ __forceinline__ void land_csr_p(const CsrRegs& f, int2* ecv, int* tab, const int2* __restrict__ cv,
                                           const MetaRegs& m, int base, int cnt, int N, int P, int lane) {
  // entry index (inside the pack) at which graphs 1..3 start; INT_MAX for graphs the pack does not hold
  const int big = 0x7fffffff;
  const int s1 = P > 1 ? __builtin_amdgcn_readlane(m.gp, 1) - base : big;
  const int s2 = P > 2 ? __builtin_amdgcn_readlane(m.gp, 2) - base : big;
  const int s3 = P > 3 ? __builtin_amdgcn_readlane(m.gp, 3) - base : big;
  auto shift = [&](i32x4 e, int idx) {             // both entries of a 16-byte pair belong to the same graph
    const int off = ((idx >= s1) + (idx >= s2) + (idx >= s3)) * N;
    e.x = (e.x == FN) ? FN : e.x + off;
    e.z = (e.z == FN) ? FN : e.z + off;
    return e;
  };
  i32x4* dst = reinterpret_cast<i32x4*>(ecv);
  if (2 * lane < cnt) dst[lane] = shift(f.e0, 2 * lane);
  if (128 + 2 * lane < cnt) dst[64 + lane] = shift(f.e1, 128 + 2 * lane);
  for (int i = 256 + lane; i < cnt; i += 64) {     // rare: > 256 entries
    int2 e = cv[base + i];
    if (e.x != FN) e.x += ((i >= s1) + (i >= s2) + (i >= s3)) * N;
    ecv[i] = e;
  }
  if (lane < P * N) {
    const int pg = (lane >= N) + (lane >= 2 * N) + (lane >= 3 * N);
    const int st = pg == 0 ? 0 : pg == 1 ? s1 : pg == 2 ? s2 : s3;
    tab[lane] = m.slot + st + ((pg * N) << 24);    // offset field += start of the graph, row field += pg * N
  }
}

// slot = (offset of the row's first entry inside the graph) | (entry count << 16) | (row << 24)
__device__ __forceinline__ void unpack_slot(int v, int& s, int& len, int& row) {
  const unsigned u = (unsigned)v;
  s = (int)(u & 0xffffu);
  len = (int)((u >> 16) & 0xffu);
  row = (int)(u >> 24);
}

// Gather of the four entries k..k+3 of one CSR row (k is a multiple of 4 in the row-padded layout):
// 2 ds_read_b128 fetch the (col,val) pairs, 4 ds_read_b128 the neighbour rows.  Padding entries
// are (col = FN, val = 0): they read the all-zero row, 0 * 0, never 0 * inf -- no masks.
__device__ __forceinline__ f32x4 gather4(const int2* ecv, const float* srcl, int k) {
  const i32x4 q0 = *reinterpret_cast<const i32x4*>(ecv + k);
  const i32x4 q1 = *reinterpret_cast<const i32x4*>(ecv + k + 2);
  const f32x4 x0 = ldv4(srcl + q0.x * FD);
  const f32x4 x1 = ldv4(srcl + q0.z * FD);
  const f32x4 x2 = ldv4(srcl + q1.x * FD);
  const f32x4 x3 = ldv4(srcl + q1.z * FD);
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f};
  fma4(a0, __int_as_float(q0.y), x0);
  fma4(a0, __int_as_float(q0.w), x1);
  fma4(a0, __int_as_float(q1.y), x2);
  fma4(a0, __int_as_float(q1.w), x3);
  return a0;
}

// One aggregation pass (4 slots j = 4p + sub of the graph's slot table, i.e. 4 rows) cut into
// micro-steps, so that the FULL kernels can place one step behind every MFMA pair of the dense
// contraction (the compiler keeps MFMAs in one clump otherwise, and an in-order wave cannot overlap
// a clump with what follows it):   slot() -> ecv() -> tile() -> fma() -> [tail()] -> a / row ready.
// Rows are ordered by decreasing length in the slot table, so tail() (rows with more than 4
// entries) fires only in the first pass or two of a graph.
struct PassSteps {
  int s, len, row;
  i32x4 q0, q1;
  f32x4 x0, x1, x2, x3;
  f32x4 a;
  __device__ __forceinline__ void slot(const int* tab, int j) { unpack_slot(tab[j], s, len, row); }
  __device__ __forceinline__ void ecv(const int2* ecv_) {
    q0 = *reinterpret_cast<const i32x4*>(ecv_ + s);
    q1 = *reinterpret_cast<const i32x4*>(ecv_ + s + 2);
  }
  __device__ __forceinline__ void tile(const float* srcl) {
    x0 = ldv4(srcl + q0.x * FD); x1 = ldv4(srcl + q0.z * FD);
    x2 = ldv4(srcl + q1.x * FD); x3 = ldv4(srcl + q1.z * FD);
  }
  __device__ __forceinline__ void fma() {
    a[0] = __int_as_float(q0.y) * x0[0]; a[1] = __int_as_float(q0.y) * x0[1];
    a[2] = __int_as_float(q0.y) * x0[2]; a[3] = __int_as_float(q0.y) * x0[3];
    fma4(a, __int_as_float(q0.w), x1);
    fma4(a, __int_as_float(q1.y), x2);
    fma4(a, __int_as_float(q1.w), x3);
  }
  __device__ __forceinline__ void tail(const int2* ecv_, const float* srcl) {
    if (__builtin_amdgcn_ballot_w64(len > 4)) {
      for (int k = 4; __builtin_amdgcn_ballot_w64(k < len); k += 4)
        if (k < len) add4(a, gather4(ecv_, srcl, s + k));
    }
  }
};

// Sparse aggregation of one graph out of a gather tile (row stride FD) without MFMA overlap: lane
// (sub, cl) produces the float4 [4cl, 4cl+4) of the rows of slots j = 4p + sub.
// emit(row, cl, acc) receives every row exactly once.
template <bool FULL, typename Emit>
__device__ __forceinline__ void aggregate_rows(const int2* ecv, const int* tab, const float* tile,
                                               int N, int dcols, int lane, Emit&& emit) {
  const int sub = lane >> 4, cl = lane & 15;
  const float* srcl = tile + cl * 4;
  if constexpr (FULL) {
#pragma unroll
    for (int p8 = 0; p8 < 8; ++p8) {
      PassSteps ps;
      ps.slot(tab, 4 * p8 + sub);
      ps.ecv(ecv);
      ps.tile(srcl);
      ps.fma();
      ps.tail(ecv, srcl);
      emit(ps.row, cl, ps.a);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    const bool col_ok = cl * 4 < dcols;
    for (int j0 = 0; j0 < N; j0 += 4) {
      const int j = j0 + sub;
      const bool ok = (j < N) && col_ok;
      int s_, len, row;
      unpack_slot(tab[ok ? j : 0], s_, len, row);
      if (!ok) len = 0;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int k = 0; __builtin_amdgcn_ballot_w64(k < len); k += 4)
        if (k < len) add4(acc, gather4(ecv, srcl, s_ + k));
      if (ok) emit(row, cl, acc);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
constexpr int A_FWD = FN * ALD;   // floats of the forward's A tile
constexpr int A_BWD = FN * BLD;   // floats of the backward's dFW tile

// weight fragments: B[k][j] with k = hi*32 + s (the K permutation matches the A fragments)
template <bool FULL>
__device__ __forceinline__ void load_w_frags(float (&wr0)[32], float (&wr1)[32],
                                             const float* __restrict__ w, int din, int dout, int li,
                                             int hi) {
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    const int k = hi * 32 + s;
    if constexpr (FULL) {
      wr0[s] = w[k * FD + li];
      wr1[s] = w[k * FD + 32 + li];
    } else {
      wr0[s] = (k < din && li < dout) ? w[(long)k * dout + li] : 0.f;
      wr1[s] = (k < din && 32 + li < dout) ? w[(long)k * dout + 32 + li] : 0.f;
    }
  }
}

// MFMA C layout (two 32x32 tiles) -> gather tile, bank = column: conflict free
__device__ __forceinline__ void store_c_tiles(float* tile, const f32x16& c0, const f32x16& c1, int li,
                                              int hi) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    tile[row * FD + li] = c0[r];
    tile[row * FD + 32 + li] = c1[r];
  }
}

// Generic shapes (N <= 32, din/dout <= 64; VEC: both multiples of 4): prefetch + phase-sequential per graph.
template <int MODE>
__global__ __launch_bounds__(512, 2) void graphconv_fwd_kernel(
    const int* __restrict__ slots, const int* __restrict__ gptr, const int2* __restrict__ cv,
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, int T, int N, int din, int dout, int max_nnz, int pack) {
  constexpr bool VEC = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, hi = lane >> 5;
  const int wpb = blockDim.x >> 6;
  WaveSlice ws = carve(smem, wave, max_nnz * pack, A_FWD, 1);

  const int stride = gridDim.x * wpb * pack;          // graphs between two tiles of this wave
  int t = __builtin_amdgcn_readfirstlane((blockIdx.x * wpb + wave) * pack);   // first graph of the tile
  if (t >= T) return;  // no workgroup barrier below: idle waves may leave

  // zero the A tile once: rows and columns no tile ever writes stay zero; rows a SHORTER last pack leaves
  // behind only produce FW rows that nothing gathers and nobody stores
  for (int i = lane; i < A_FWD; i += 64) ws.a[i] = 0.f;
  for (int i = lane; i < FD; i += 64) ws.b[FN * FD + i] = 0.f;

  float wr0[32], wr1[32];
  load_w_frags<false>(wr0, wr1, w, din, dout, li, hi);
  const float b0 = (bias && li < dout) ? bias[li] : 0.f;
  const float b1 = (bias && 32 + li < dout) ? bias[32 + li] : 0.f;

  auto graphs_at = [&](int tt) { return T - tt < pack ? T - tt : pack; };
  int P = graphs_at(t);

  TileRegs fx;
  CsrRegs fc;
  MetaRegs m_cur, m_nxt;
  issue_meta_p(m_cur, slots, gptr, t, P, N, lane);
  int base = meta_base(m_cur), cnt = meta_cnt_p(m_cur, P);
  issue_tile_g<MODE>(fx, x + (long)t * N * din, P * N * din, lane);
  issue_cv(fc, cv, base, cnt, lane);
  int tn = t + stride;
  // unconditional (index clamped): a conditional load becomes a phi whose copy forces
  // s_waitcnt vmcnt(0) right behind the prefetch
  int Pn = graphs_at(tn < T ? tn : t);
  issue_meta_p(m_nxt, slots, gptr, tn < T ? tn : t, Pn, N, lane);
  wave_sync();

  for (;;) {
    const int rows = P * N;
    land_tile_g<MODE>(fx, ws.a, ALD, rows * din, din, lane);
    land_csr_p(fc, ws.ecv, ws.rp, cv, m_cur, base, cnt, N, P, lane);
    wave_sync();

    // next tile in flight while this one is computed.  Branch-free on purpose: values defined
    // under `if (has_next)` become phis whose copies make the compiler wait (vmcnt(0)) for the
    // prefetch right after issuing it; on the last iteration the current tile is re-read (L2 hit).
    const bool has_next = tn < T;
    const int tp = has_next ? tn : t;
    const int base_n = meta_base(m_nxt), cnt_n = meta_cnt_p(m_nxt, Pn);
    issue_tile_g<MODE>(fx, x + (long)tp * N * din, Pn * N * din, lane);
    issue_cv(fc, cv, base_n, cnt_n, lane);
    m_cur = m_nxt;
    const int P_next = Pn;
    {
      const int tnn = tn + stride;
      const int tq = tnn < T ? tnn : tp;
      Pn = graphs_at(tq);
      issue_meta_p(m_nxt, slots, gptr, tq, Pn, N, lane);
    }

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = b0; acc1[r] = b1; }
    {
      f32x4 a4[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) a4[q] = ldv4(ws.a + li * ALD + hi * 32 + q * 4);
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        const float a = a4[s >> 2][s & 3];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wr0[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wr1[s], acc1, 0, 0, 0);
      }
    }
    store_c_tiles(ws.b, acc0, acc1, li, hi);
    wave_sync();

    float* ot = out + (long)t * N * dout;
    aggregate_rows<false>(ws.ecv, ws.rp, ws.b, rows, dout, lane, [&](int r, int cl, f32x4 acc) {
      if constexpr (VEC) {
        stv4(ot + (long)r * dout + cl * 4, acc);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (cl * 4 + j < dout) ot[(long)r * dout + cl * 4 + j] = acc[j];
      }
    });
    wave_sync();

    if (!has_next) break;
    t = tn;
    tn += stride;
    P = P_next;
    base = base_n;
    cnt = cnt_n;
  }
}

// FULL shape (N = 32, din = dout = 64): all sizes compile-time, straight-line aggregation, one graph
// of lag between the contraction and the aggregation (FW(t) is produced in the step in which graph
// t - nwaves is aggregated out of the gather tile), two waves per SIMD.
//
// FW = X W runs on the bf16 matrix pipe (exact 3-way split of both operands, 6 products per k-step, see
// split_pair): 48 MFMAs of 32 cycles that execute BESIDE the vector ALU -- while one wave of the SIMD
// multiplies, the other aggregates -- instead of 64 f32 MFMAs of 64 cycles on the VALU datapath.  The W
// pieces are resident: p1 and p2 of every B fragment in 64 registers, p3 (used by one product in six) in
// an 8 KB LDS table shared by the workgroup (register budget: 256 per wave, a spill would put scratch
// loads -- and their vmcnt waits -- in front of the prefetch loads).
//
// LDS: p3 table | per wave: x tile [32][68], gather tile [33][64], ONE CSR slice (landed at the end of the
// step, after the aggregation of the previous graph released it).
constexpr size_t FWD_FULL_SHARED = 2 * 4 * 64 * 16;   // W p3 fragments: [tile][k-step][lane] x 16 bytes

__global__ __launch_bounds__(512, 2) void graphconv_fwd_full_kernel(
    const int* __restrict__ slots, const int* __restrict__ gptr, const int2* __restrict__ cv,
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, int T, int max_nnz) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int N = FN, D = FD;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, hi = lane >> 5;
  const int wpb = blockDim.x >> 6;
  u32x4* w3 = reinterpret_cast<u32x4*>(smem);
  WaveSlice ws = carve(smem + FWD_FULL_SHARED, wave, max_nnz, A_FWD, 1);

  // B fragments of FW = X W: lane (li, hi), tile nt, k-step ks holds W[k][32 nt + li] for
  // k = 16 ks + 8 hi + j as three bf16 pieces
  u32x4 WF[2][4][2];
  static_for<8>([&](auto c) __attribute__((always_inline)) {
    constexpr int nt = decltype(c)::value >> 2, ks = decltype(c)::value & 3;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = w[(16 * ks + 8 * hi + j) * D + 32 * nt + li];
    Frag3 f;
    split8(v, f);
    WF[nt][ks][0] = f.p1; WF[nt][ks][1] = f.p2;
    if (wave == 0) w3[(nt * 4 + ks) * 64 + lane] = f.p3;
  });
  __syncthreads();

  const int nwaves = gridDim.x * wpb;
  int t = __builtin_amdgcn_readfirstlane(blockIdx.x * wpb + wave);
  if (t >= T) return;   // no workgroup barrier below: idle waves may leave
#ifdef KGCN_ABL_HOT       // development: every wave re-reads (and re-writes) its first two graphs -- the kernel without HBM traffic
  const int t_first = t;
  auto hot = [&](int tt) { return t_first + (((tt - t_first) / nwaves) & 1) * nwaves < T ? t_first + (((tt - t_first) / nwaves) & 1) * nwaves : t_first; };
#else
  auto hot = [&](int tt) { return tt; };
#endif

  for (int i = lane; i < D; i += 64) ws.b[FN * FD + i] = 0.f;   // zero row for padding entries
  const float b0 = bias ? bias[li] : 0.f;
  const float b1 = bias ? bias[32 + li] : 0.f;

  // ---- prologue ------------------------------------------------------------------------------
  TileRegs fx;
  CsrRegs fc;
  MetaRegs m_cur, m_nxt;
  issue_meta(m_cur, slots, gptr, hot(t), N, lane);
  int base = meta_base(m_cur), cnt = meta_cnt(m_cur);
  issue_tile<true>(fx, x + (long)hot(t) * N * D, 512, lane);
  issue_cv(fc, cv, base, cnt, lane);
  int tn = t + nwaves;
  issue_meta(m_nxt, slots, gptr, hot(tn < T ? tn : t), N, lane);
  wave_sync();

  int t_prev = t;      // graph whose FW sits in the gather tile (valid from the second step on)
  bool has_next = true;
  PROBE_DECL
  auto aggregate_prev = [&]() __attribute__((always_inline)) {
    float* ot = out + (long)hot(t_prev) * N * D;
    aggregate_rows<true>(ws.ecv, ws.rp, ws.b, N, D, lane, [&](int r, int c4, f32x4 v) {
      stv4(ot + r * D + c4 * 4, v);
    });
  };
  auto step = [&](auto agg_tag) __attribute__((always_inline)) {
    constexpr bool AGG = decltype(agg_tag)::value;
    PROBE(0)
    // ---- 1. x(t): registers -> LDS; x(t + nwaves) in flight (branch-free: values defined under
    // `if (has_next)` become phis whose copies make the compiler wait for the prefetch right after
    // issuing it; on the last step the current graph is re-read (L2 hit), unused) ---------------------
    land_tile<true>(fx, ws.a, ALD, 512, 16, lane);
    has_next = tn < T;
    const int tp = has_next ? tn : t;
    issue_tile<true>(fx, x + (long)hot(tp) * N * D, 512, lane);
    wave_sync();
    PROBE(1)

    // ---- 2. aggregation of graph t_prev (its FW sits in the gather tile, its CSR slice in LDS) ------
    if constexpr (AGG) aggregate_prev();
    wave_sync();
    PROBE(2)

    // ---- 3. CSR(t): registers -> LDS (the slice of t_prev is dead); CSR(t + nwaves) in flight ---------
    land_csr(fc, ws.ecv, ws.rp, cv, m_cur.slot, base, cnt, N, lane);
    const int base_n = meta_base(m_nxt), cnt_n = meta_cnt(m_nxt);
    issue_cv(fc, cv, base_n, cnt_n, lane);
    // a REAL register move before the reload of m_nxt (a plain copy becomes a loop phi whose copy lands
    // behind the new load: s_waitcnt vmcnt(~0) on the prefetch at the top of the next step)
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3"
                 : "=&v"(m_cur.slot), "=&v"(m_cur.gp) : "v"(m_nxt.slot), "v"(m_nxt.gp));
    {
      const int tnn = tn + nwaves;
      issue_meta(m_nxt, slots, gptr, hot(tnn < T ? tnn : tp), N, lane);
    }

    // ---- 4. FW(t) = X(t) W + b: per k-step 8 values of row li (k = 16 ks + 8 hi + j) are split, then
    // 6 products x 2 output tiles ------------------------------------------------------------------------
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = b0; acc1[r] = b1; }
    static_for<4>([&](auto ksc) __attribute__((always_inline)) {
      constexpr int ks = decltype(ksc)::value;
      const float* src = ws.a + li * ALD + 16 * ks + 8 * hi;
      const f32x4 lo = ldv4(src), hi4 = ldv4(src + 4);
      const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
      Frag3 fa;
      split8(v, fa);
      const u32x4 w30 = w3[ks * 64 + lane], w31 = w3[(4 + ks) * 64 + lane];
      acc0 = mfma_bf16(fa.p3, WF[0][ks][0], acc0);      // smallest terms first
      acc1 = mfma_bf16(fa.p3, WF[1][ks][0], acc1);
      acc0 = mfma_bf16(fa.p2, WF[0][ks][1], acc0);
      acc1 = mfma_bf16(fa.p2, WF[1][ks][1], acc1);
      acc0 = mfma_bf16(fa.p1, w30, acc0);
      acc1 = mfma_bf16(fa.p1, w31, acc1);
      acc0 = mfma_bf16(fa.p2, WF[0][ks][0], acc0);
      acc1 = mfma_bf16(fa.p2, WF[1][ks][0], acc1);
      acc0 = mfma_bf16(fa.p1, WF[0][ks][1], acc0);
      acc1 = mfma_bf16(fa.p1, WF[1][ks][1], acc1);
      acc0 = mfma_bf16(fa.p1, WF[0][ks][0], acc0);
      acc1 = mfma_bf16(fa.p1, WF[1][ks][0], acc1);
      __builtin_amdgcn_sched_barrier(0);   // one k-step of fragments live at a time (register budget)
    });
    PROBE(3)
    // ---- 5. FW(t) replaces the gather tile ------------------------------------------------------
    store_c_tiles(ws.b, acc0, acc1, li, hi);
    wave_sync();
    PROBE(4)

    t_prev = t;
    t = tn;
    tn += nwaves;
    base = base_n;
    cnt = cnt_n;
  };

#ifdef KGCN_FWD_STAGGER                          // development: the SIMD's second wave starts half a step late
  if (wave >= wpb / 2) {
#pragma unroll 1
    for (int k = 0; k < KGCN_FWD_STAGGER; ++k) __builtin_amdgcn_s_sleep(16);
  }
#endif
  step(std::false_type{});                       // first graph: nothing to aggregate yet
  while (has_next) step(std::true_type{});
  aggregate_prev();                              // epilogue: the last graph
  PROBE_FLUSH(blockIdx.x * wpb + wave)
}

// ------------------------------------------------------------------------------------------------
// backward, generic shapes: prefetch + phase-sequential per graph, 2 waves per SIMD
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(512, 2) void graphconv_bwd_kernel(
    const int* __restrict__ slots_t, const int* __restrict__ gptr_t, const int2* __restrict__ cv_t,
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ g,
    float* __restrict__ dx, float* __restrict__ part_dw, float* __restrict__ part_db, int T, int N,
    int din, int dout, int max_nnz, int pack) {
  constexpr bool FULL = false;
  constexpr bool VEC = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, hi = lane >> 5;
  const int wpb = blockDim.x >> 6;
  float* Wt = reinterpret_cast<float*>(smem);  // [FD][FD]: Wt[k][j] = W[j][k] (zero padded)
  WaveSlice ws = carve(smem + FD * FD * 4, wave, max_nnz * pack, A_BWD, 1);

  for (int i = tid; i < FD * FD; i += blockDim.x) {
    const int k = i >> 6, j = i & 63;
    Wt[i] = (j < din && k < dout) ? w[(long)j * dout + k] : 0.f;
  }
  for (int i = lane; i < A_BWD; i += 64) ws.a[i] = 0.f;
  for (int i = lane; i < (FN + 1) * FD; i += 64) ws.b[i] = 0.f;
  __syncthreads();

  f32x16 dw00, dw01, dw10, dw11;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dw00[r] = 0.f; dw01[r] = 0.f; dw10[r] = 0.f; dw11[r] = 0.f; }
  f32x4 dbacc = {0.f, 0.f, 0.f, 0.f};

  const int din4 = din >> 2;
  const int stride = gridDim.x * wpb * pack;           // graphs between two tiles of this wave (see the forward)
  int t = __builtin_amdgcn_readfirstlane((blockIdx.x * wpb + wave) * pack);

  if (t < T) {
    auto graphs_at = [&](int tt) { return T - tt < pack ? T - tt : pack; };
    int P = graphs_at(t);
    TileRegs fg, fx;
    CsrRegs fc;
    MetaRegs m_cur, m_nxt;
    issue_meta_p(m_cur, slots_t, gptr_t, t, P, N, lane);
    int base = meta_base(m_cur), cnt = meta_cnt_p(m_cur, P);
    issue_tile_g<MODE>(fg, g + (long)t * N * dout, P * N * dout, lane);
    issue_cv(fc, cv_t, base, cnt, lane);
    issue_tile_g<MODE>(fx, x + (long)t * N * din, P * N * din, lane);
    int tn = t + stride;
    int Pn = graphs_at(tn < T ? tn : t);
    issue_meta_p(m_nxt, slots_t, gptr_t, tn < T ? tn : t, Pn, N, lane);

    PROBE_DECL
    for (;;) {
      PROBE(0)
      // ---- 1. g[t], CSR(A^T) slice: registers -> LDS ------------------------------------------
      const int rows = P * N;
      const int nxe = rows * din, nge = rows * dout, nx4 = rows * din4;
      land_tile_g<MODE>(fg, ws.b, FD, nge, dout, lane);
      land_csr_p(fc, ws.ecv, ws.rp, cv_t, m_cur, base, cnt, N, P, lane);
      if (P < pack) {
        // a shorter last pack: the dFW rows a full pack of this wave left behind must not reach dW
        for (int i = rows * BLD + lane; i < pack * N * BLD; i += 64) ws.a[i] = 0.f;
      }
      wave_sync();

      PROBE(1)
      // ---- 2. dFW = A^T @ g -> dFW tile (odd stride), dbias partial ----------------------------
      aggregate_rows<FULL>(ws.ecv, ws.rp, ws.b, rows, dout, lane, [&](int r, int cl, f32x4 acc) {
        float* d = ws.a + r * BLD + cl * 4;
        d[0] = acc[0]; d[1] = acc[1]; d[2] = acc[2]; d[3] = acc[3];
        dbacc += acc;
      });
      wave_sync();
      PROBE(2)

      // ---- 3. x[t] -> gather tile (g is dead) --------------------------------------------------
      land_tile_g<MODE>(fx, ws.b, FD, nxe, din, lane);

      wave_sync();
      PROBE(3)
      // ---- 4. dW += x^T @ dFW on the bf16 matrix pipe (exact 3-way split, 6 products) ------------------
      // lane (li, hi): columns li / 32+li of x and dFW over its 16 nodes n = hi*16 + s; bf16 k-step ks
      // takes nodes hi*16 + 8ks .. +7 (the same node set for both operands).  The second wave of the
      // SIMD runs its VALU/LDS phases beside these MFMAs.
      static_for<2>([&](auto ksc) __attribute__((always_inline)) {
        constexpr int ks = decltype(ksc)::value;
        u32x4 F[4][3];                             // [operand: x lo, x hi, dFW lo, dFW hi][piece]
        static_for<16>([&](auto qc) __attribute__((always_inline)) {
          constexpr int Q = decltype(qc)::value, OP = Q >> 2, J = Q & 3;
          const int n = hi * 16 + 8 * ks + 2 * J;
          const float* src = (OP < 2 ? ws.b + n * FD : ws.a + n * BLD) + (OP & 1) * 32 + li;
          unsigned q1, q2, q3;
          split_pair(src[0], src[OP < 2 ? FD : BLD], q1, q2, q3);
          F[OP][0][J] = q1; F[OP][1][J] = q2; F[OP][2][J] = q3;
        });
        static_for<6>([&](auto pc) __attribute__((always_inline)) {
          constexpr int pr = decltype(pc)::value;
          constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};   // smallest terms first
          dw00 = mfma_bf16(F[0][PA[pr]], F[2][PB[pr]], dw00);
          dw01 = mfma_bf16(F[0][PA[pr]], F[3][PB[pr]], dw01);
          dw10 = mfma_bf16(F[1][PA[pr]], F[2][PB[pr]], dw10);
          dw11 = mfma_bf16(F[1][PA[pr]], F[3][PB[pr]], dw11);
        });
      });

      // ---- 5. next graph (g, CSR, x) in flight during the dX phase and the other wave's work (issued
      // AFTER the dW phase: its 72 registers would not fit beside the split fragments) ---------------
      const bool has_next = tn < T;
      const int tp = has_next ? tn : t;
      const int base_n = meta_base(m_nxt), cnt_n = meta_cnt_p(m_nxt, Pn);
      issue_tile_g<MODE>(fg, g + (long)tp * N * dout, Pn * N * dout, lane);
      issue_cv(fc, cv_t, base_n, cnt_n, lane);
      issue_tile_g<MODE>(fx, x + (long)tp * N * din, Pn * N * din, lane);
      m_cur = m_nxt;
      const int P_next = Pn;
      {
        const int tnn = tn + stride;
        const int tq = tnn < T ? tnn : tp;
        Pn = graphs_at(tq);
        issue_meta_p(m_nxt, slots_t, gptr_t, tq, Pn, N, lane);
      }

      PROBE(4)
      // ---- 6. dX = dFW @ W^T (A operand straight from the odd-stride LDS tile) ------------------
      if (FULL || dx) {
        f32x16 c0, c1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
#pragma unroll 8
        for (int s = 0; s < 32; ++s) {
          const int k = hi * 32 + s;
          const float a = ws.a[li * BLD + k];
          c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Wt[k * FD + li], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Wt[k * FD + 32 + li], c1, 0, 0, 0);
        }
        // C layout -> LDS (the x tile is dead) -> whole rows as dwordx4
        PROBE(5)
        wave_sync();
        store_c_tiles(ws.b, c0, c1, li, hi);
        wave_sync();
        float* dxt = dx + (long)t * N * din;
        if constexpr (VEC) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int i = lane + q * 64;
            if (i < nx4) {
              const int r = i / din4, c4 = i - r * din4;
              stv4(dxt + (long)i * 4, ldv4(ws.b + r * FD + c4 * 4));
            }
          }
        } else if constexpr (MODE == 1) {
          int e0 = lane * 4;
          int r = e0 / din, c = e0 - r * din;
          const int dr = 256 / din, dc = 256 - dr * din;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int i = lane + q * 64;
            if (i < (nxe >> 2)) {
              f32x4 v4;
              int rr = r, cc = c;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                v4[j] = ws.b[rr * FD + cc];
                if (++cc == din) { cc = 0; ++rr; }
              }
              stv4(dxt + (long)i * 4, v4);
            }
            c += dc; r += dr;
            if (c >= din) { c -= din; ++r; }
          }
        } else {
          int r = lane / din, c = lane - r * din;
          const int dr = 64 / din, dc = 64 - dr * din;
#pragma unroll
          for (int q = 0; q < 32; ++q) {
            const int e = lane + q * 64;
            if (e < nxe) dxt[e] = ws.b[r * FD + c];
            c += dc; r += dr;
            if (c >= din) { c -= din; ++r; }
          }
        }
      }
      wave_sync();
      PROBE(6)

      if (!has_next) break;
      t = tn;
      tn += stride;
      P = P_next;
      base = base_n;
      cnt = cnt_n;
    }
    PROBE_FLUSH(blockIdx.x * wpb + wave)
  }

  // ---- reduce the workgroup's waves through LDS; one partial per workgroup leaves the chip -----
  __syncthreads();
  float* park = ws.a;  // dFW tile (8320 B) and gather tile (8448 B) are contiguous: 4160 floats fit
  static_assert(((A_BWD * 4 + 15) & ~15) + (FN + 1) * FD * 4 >= (FD * FD + FD) * 4, "park area");
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    park[row * FD + li] = dw00[r];
    park[row * FD + 32 + li] = dw01[r];
    park[(32 + row) * FD + li] = dw10[r];
    park[(32 + row) * FD + 32 + li] = dw11[r];
  }
  // dbacc: lane (sub, cl) holds the column-4-group cl summed over its rows
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float v = dbacc[j];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    dbacc[j] = v;
  }
  if (lane < 16) stv4(park + FD * FD + lane * 4, dbacc);
  __syncthreads();
  const size_t slice_f = slice_bytes(max_nnz * pack, A_BWD, 1) / 4;
  const float* slice0 = reinterpret_cast<const float*>(smem + FD * FD * 4);
  float* pw = part_dw + (long)blockIdx.x * din * dout;
  for (int i = tid; i < FD * FD; i += blockDim.x) {
    float s = 0.f;
    for (int wv = 0; wv < wpb; ++wv) s += slice0[wv * slice_f + i];
    const int row = i >> 6, col = i & 63;
    if (row < din && col < dout) pw[(long)row * dout + col] = s;
  }
  if (tid < dout) {
    float s = 0.f;
    for (int wv = 0; wv < wpb; ++wv) s += slice0[wv * slice_f + FD * FD + tid];
    part_db[(long)blockIdx.x * dout + tid] = s;
  }
}

constexpr int BWD_FULL_WPB = 4;

// ------------------------------------------------------------------------------------------------
// backward, FULL shape, "planes" version: every fp32 -> 3 x bf16 split happens ONCE, at the producer.
//   * dFW(i+1) = A^T g(i+1) leaves the aggregation as three bf16 planes [32 nodes x 64 features] in LDS (two
//     split_pair + three 8-byte stores per row and lane).  Both contractions read their operand fragments straight
//     out of the planes: dX = dFW W^T wants 8 consecutive features of a node  -> one ds_read_b128 per piece and
//     k-step; dW = x^T dFW wants 8 consecutive NODES of a feature column -> two ds_read_b64_tr_b16 (the LDS
//     transpose read of gfx950: within 16 lanes, lane l element j receives element l&3 of what lane 4j + (l>>2)&3
//     addressed) per piece, tile and k-step.  No operand is split twice and no split sits in front of an MFMA group.
//   * x never touches LDS: its tile is loaded from HBM directly in the A-fragment layout of dW (lane (li, hi): feature
//     32 mt + li of nodes 16 ks + 8 hi + j -- 32 dword loads, every one of them two full 128-byte lines), a whole
//     iteration ahead, and is split into resident fragments behind the MFMAs of phase B.
//   * dX is accumulated TRANSPOSED (dX^T = W dFW^T: operands swapped), so a lane owns 4 consecutive floats of a row per
//     register quad and dX leaves as 16-byte stores (tools/bwd_skeleton.hip: the memory system is indifferent to the
//     store shape of this kernel; the instruction stream is not).
// Plane layout (bytes inside one 4 KiB plane; R = r >> 2, a = r & 3, h = f >> 5, c = (f & 31) >> 3):
//     ((2R + h) << 8) | ((a ^ h) << 6) | ((c ^ (R & 3)) << 4) | ((f & 7) << 1)
// i.e. four consecutive rows share a 256-byte bank row (one 64-byte quarter each), the two 32-feature halves sit in
// different quarters, 16-byte chunks are rotated by the row quad: the 8-byte row stores of the aggregation, the
// transpose reads (4 rows x 64 bytes per 32 lanes) and the 16-byte row reads (16-lane groups of ds_read_b128) are all
// bank-conflict free, and the address splits into (row word from the slot table) XOR (lane constant).
// Instruction count per graph 1,580 -> ~1,150 (VALU 990 -> ~700).
// ------------------------------------------------------------------------------------------------
constexpr int PL_BYTES = FN * FD * 2;          // one bf16 plane
constexpr int DFWP_BYTES = 3 * PL_BYTES;       // p1 | p2 | p3

// LDS of the workgroup: [4 waves x 2 plane buffers (4 KiB aligned: XOR addressing)] [W p2 / p3 fragment table]
// [4 waves x (gather tile, 2 CSR slices)]
constexpr int BWD_WPB_ = 4;
constexpr size_t PLANES_ALL = (size_t)BWD_WPB_ * 2 * DFWP_BYTES;            // 98,304
constexpr size_t WTAB_BYTES = 2 * 8 * 64 * 16;                             // [piece p2,p3][nt][ks][lane] x 16 bytes
__host__ __device__ inline size_t bwd_planes_wave_bytes(int max_nnz) {
  return (size_t)(FN + 1) * FD * 4 + 2 * (ecv_bytes(max_nnz) + RP_BYTES);
}
__host__ __device__ inline size_t bwd_planes_lds_bytes(int max_nnz) {
  return PLANES_ALL + WTAB_BYTES + BWD_WPB_ * bwd_planes_wave_bytes(max_nnz);
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short i16x4 __attribute__((ext_vector_type(4)));
#define KGCN_LDS __attribute__((address_space(3)))
__device__ __forceinline__ unsigned lds_off(const void* p) {
  return (unsigned)(uintptr_t)(const KGCN_LDS unsigned char*)p;
}
__device__ __forceinline__ u32x4 lds_ld128(unsigned a) { return *(const KGCN_LDS u32x4*)(uintptr_t)a; }
__device__ __forceinline__ void lds_st64(unsigned a, unsigned lo, unsigned hi) {
  u32x2 v = {lo, hi};
  *(KGCN_LDS u32x2*)(uintptr_t)a = v;
}
__device__ __forceinline__ u32x2 lds_ld_tr16(unsigned a) {
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((KGCN_LDS i16x4*)(uintptr_t)a));
}

// slot word of the packer (offset | count << 16 | row << 24) -> offset (12 bits) | count << 12 | row word << 20, where
// the row word is the row's share of the plane address (>> 4):  R << 5 | a << 2 | (R & 3)
__device__ __forceinline__ int slot_plane_word(int v) {
  const unsigned u = (unsigned)v, row = u >> 24, R = row >> 2;
  const unsigned rw = (R << 5) | ((row & 3) << 2) | (R & 3);
  return (int)((u & 0xfffu) | (((u >> 16) & 0xffu) << 12) | (rw << 24));
}

struct PlaneSteps {       // PassSteps on the plane slot word
  int s, len;
  unsigned rw;            // row word << 4 (byte address share of the row)
  i32x4 q0, q1;
  f32x4 x0, x1, x2, x3;
  f32x4 a;
  __device__ __forceinline__ void slot(const int* tab, int j) {
    const unsigned u = (unsigned)tab[j];
    s = (int)(u & 0xfffu);
    len = (int)((u >> 12) & 0xffu);
    rw = (u >> 24) << 4;
  }
  __device__ __forceinline__ void tail(const int2* ecv_, const float* srcl) {
    if (__builtin_amdgcn_ballot_w64(len > 4)) {
      for (int k = 4; __builtin_amdgcn_ballot_w64(k < len); k += 4)
        if (k < len) add4(a, gather4(ecv_, srcl, s + k));
    }
  }
};
template <typename Emit>
__device__ __forceinline__ void plane_half(PlaneSteps& P, int h, bool do_emit, const int* tab, int j,
                                           const int2* ecv, const float* srcl, Emit&& emit) {
  if (h == 0) { if (do_emit) emit(P); }
  else if (h == 1) P.slot(tab, j);
  else if (h == 2) P.q0 = *reinterpret_cast<const i32x4*>(ecv + P.s);
  else if (h == 3) P.q1 = *reinterpret_cast<const i32x4*>(ecv + P.s + 2);
  else if (h == 4) { P.x0 = ldv4(srcl + P.q0.x * FD); P.x1 = ldv4(srcl + P.q0.z * FD); }
  else if (h == 5) { P.x2 = ldv4(srcl + P.q1.x * FD); P.x3 = ldv4(srcl + P.q1.z * FD); }
  else if (h == 6) {
    const float v = __int_as_float(P.q0.y);
    P.a[0] = v * P.x0[0]; P.a[1] = v * P.x0[1]; P.a[2] = v * P.x0[2]; P.a[3] = v * P.x0[3];
    fma4(P.a, __int_as_float(P.q0.w), P.x1);
  } else {
    fma4(P.a, __int_as_float(P.q1.y), P.x2);
    fma4(P.a, __int_as_float(P.q1.w), P.x3);
  }
}

__global__ __launch_bounds__(256, 1) void graphconv_bwd_planes_kernel(
    const int* __restrict__ slots_t, const int* __restrict__ gptr_t, const int2* __restrict__ cv_t,
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ g,
    float* __restrict__ dx, float* __restrict__ part_dw, float* __restrict__ part_db, int T,
    int max_nnz) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int N = FN, D = FD;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, hi = lane >> 5;
  const int sub = lane >> 4, cl = lane & 15;
  static_assert(BWD_WPB_ == BWD_FULL_WPB, "waves per workgroup");
  unsigned char* sl = smem + (size_t)wave * 2 * DFWP_BYTES;
  const unsigned pl0 = lds_off(sl);                               // dFW planes, buffer 0 (buffer 1: + DFWP_BYTES)
  if (pl0 & 4095u) __builtin_trap();                              // XOR addressing needs 4 KiB aligned plane buffers
  const unsigned wtab = lds_off(smem + PLANES_ALL) + (unsigned)lane * 16;
  float* gt = reinterpret_cast<float*>(smem + PLANES_ALL + WTAB_BYTES +
                                       (size_t)wave * bwd_planes_wave_bytes(max_nnz));   // [FN+1][FD], row FN stays zero
  int2* ecv0 = reinterpret_cast<int2*>(gt + (FN + 1) * FD);
  const size_t ecv_stride = ecv_bytes(max_nnz) / 8;
  int* tab0 = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(ecv0) + 2 * ecv_bytes(max_nnz));

  for (int i = lane; i < D; i += 64) gt[FN * FD + i] = 0.f;

  // ---- lane constants of the plane addressing ---------------------------------------------------------------
  // aggregation rows (lane (sub, cl): features 4cl .. 4cl+3 of the row): address = row word ^ this, per buffer
  const unsigned h_e = (unsigned)cl >> 3;
  const unsigned lanexor = (h_e << 8) | (h_e << 6) | ((((unsigned)cl & 7) >> 1) << 4) | (((unsigned)cl & 1) << 3);
  // transpose reads: lane l addresses 4 features (16 nb + 4 a4 ..) of node 16 ks + 8 hi + 4 rd + jj
  const unsigned nb = ((unsigned)lane >> 4) & 1, jj = ((unsigned)lane >> 2) & 3, a4 = (unsigned)lane & 3;
  unsigned TRB[2][2];                                            // [feature half nt][read rd]
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd)
      TRB[nt][rd] = pl0 + (((unsigned)hi << 10) | ((jj ^ (unsigned)nt) << 6) |
                           (((2 * nb + (a4 >> 1)) ^ (2 * (unsigned)hi + (unsigned)rd)) << 4) | ((a4 & 1) << 3));
  // row reads: lane (li, hi) addresses features 16 ks + 8 hi .. + 7 of node li
  unsigned LB[4];
  {
    const unsigned R = (unsigned)li >> 2, a = (unsigned)li & 3;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      LB[ks] = pl0 + ((R << 9) | ((a ^ (unsigned)(ks >> 1)) << 6) |
                      (((2 * (unsigned)(ks & 1) + (unsigned)hi) ^ (R & 3)) << 4));
  }

  // A fragments of dX^T = W dFW^T: lane (li, hi), tile nt, k-step ks holds W[32 nt + li][16 ks + 8 hi + j].  The p1 pieces
  // (three products in six) are resident in 32 registers, p2 / p3 sit in a 16 KiB table shared by the workgroup and are
  // read one k-step ahead in phase B (the only phase that uses W: 64 registers less in phase A, the register peak).
  u32x4 WF1[2][4];
  static_for<8>([&](auto c) __attribute__((always_inline)) {
    constexpr int nt = decltype(c)::value >> 2, ks = decltype(c)::value & 3;
    const float* src = w + (32 * nt + li) * D + 16 * ks + 8 * hi;
    const f32x4 lo = ldv4(src), hi4 = ldv4(src + 4);
    const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    Frag3 f;
    split8(v, f);
    WF1[nt][ks] = f.p1;
    if (wave == (decltype(c)::value >> 1)) {
      *(KGCN_LDS u32x4*)(uintptr_t)(wtab + (0 * 8 + nt * 4 + ks) * 1024) = f.p2;
      *(KGCN_LDS u32x4*)(uintptr_t)(wtab + (1 * 8 + nt * 4 + ks) * 1024) = f.p3;
    }
  });
  __syncthreads();

  f32x16 dw00, dw01, dw10, dw11;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dw00[r] = 0.f; dw01[r] = 0.f; dw10[r] = 0.f; dw11[r] = 0.f; }
  f32x4 dbacc = {0.f, 0.f, 0.f, 0.f};

  const int nwaves = gridDim.x * BWD_FULL_WPB;
  const int t0 = __builtin_amdgcn_readfirstlane(blockIdx.x * BWD_FULL_WPB + wave);
  if (t0 < T) {
    const int cntw = (T - 1 - t0) / nwaves + 1;              // graphs of this wave
    const int tl = t0 + (cntw - 1) * nwaves;                  // its last graph
#ifdef KGCN_ABL_HOT
    auto gidx = [&](int k) { const int t = t0 + (k & 1) * nwaves; return t < tl ? t : tl; };
#else
    auto gidx = [&](int k) { const int t = t0 + k * nwaves; return t < tl ? t : tl; };  // clamped
#endif
    const float* srcl = gt + cl * 4;
    const int xlane = (8 * hi) * D + 2 * li;                  // lane share of the x fragment addresses (floats)

    TileRegs gpf;
    CsrRegs cpf;
    MetaRegs m_a, m_b;   // m_a: graph whose g/CSR are in flight; m_b: the one after it
    // x tile in flight in the A-fragment layout of dW.  Row li of M tile mt is feature 2 li + mt (the rows of dW come out
    // interleaved; undone when the accumulators are parked), so a lane's two tiles are ADJACENT floats: one 8-byte
    // load per node, xr[8 ks + j] = x[16 ks + 8 hi + j][2 li .. 2 li + 1] -- 16 loads, each two full 256-byte rows
    f32x2 xr[16];
    float sr0[16], ss0[16], sr1[16], sv0[16], sv1[16];   // split_pair of unit u = 2 (4 ks + J) + mt cut in two halves (one per MFMA slot)
    u32x4 XF[2][2][3];   // x(i) fragments [k-step][feature half mt][piece]
    u32x4 BF0[2][3];     // dFW(i) fragments of k-step 0 [feature half nt][piece] (read one phase ahead)

    // xb: this lane's pointer into the tile (tile base + xlane); the two 4 KiB halves get their own 64-bit lane pointer,
    // everything else is an immediate (a 32-bit unsigned lane offset per load made the compiler hoist sixteen 64-bit
    // indices out of the loop -- and spill them)
    auto load_x = [&](const float* xb, auto qc) __attribute__((always_inline)) {
      constexpr int q = decltype(qc)::value;
      // two wave-uniform bases 4 KiB apart + one lane offset + an immediate (13-bit signed) per load: no address registers
      xr[q] = *reinterpret_cast<const f32x2*>((q >> 3 ? xb + 16 * D : xb) + (q & 7) * D);
    };
    auto issue_x = [&](int k) __attribute__((always_inline)) {
      const float* xb = x + (long)gidx(k) * N * D + xlane;
      static_for<16>([&](auto qc) __attribute__((always_inline)) { load_x(xb, qc); });
    };
    // half hh of unit u: nodes (2J, 2J+1) of k-step ks, tile mt -> XF[ks][mt][.][J]
    auto split_x_half = [&](auto uc, auto hc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value, hh = decltype(hc)::value, P = u >> 1, mt = u & 1, ks = P >> 2, J = P & 3;
      const unsigned msk = 0xffff0000u;
      if constexpr (hh == 0) {
        // The tile in flight is loop carried (loaded one iteration ahead) and must not compete for the 256 VALU-addressable
        // registers: with it there the allocator split its live range -- new loads into fresh registers, copies on the
        // back edge -- and the copies drew s_waitcnt vmcnt(10) right behind the loads (~1,000 cycles per graph).  The
        // "a" operand keeps it in the accumulator file, which loads can target; one v_accvgpr_read per value moves it out.
        float v0, v1;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v0) : "a"(xr[8 * ks + 2 * J][mt]));
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v1) : "a"(xr[8 * ks + 2 * J + 1][mt]));
        sv0[u] = v0; sv1[u] = v1;
        const float h0 = __uint_as_float(__float_as_uint(v0) & msk), h1 = __uint_as_float(__float_as_uint(v1) & msk);
        sr0[u] = v0 - h0;
        sr1[u] = v1 - h1;
        ss0[u] = sr0[u] - __uint_as_float(__float_as_uint(sr0[u]) & msk);
        KGCN_PIN3(sr0[u], sr1[u], ss0[u]);
      } else {
        const float v0 = sv0[u], v1 = sv1[u];
        const float s1 = sr1[u] - __uint_as_float(__float_as_uint(sr1[u]) & msk);
        XF[ks][mt][0][J] = __builtin_amdgcn_perm(__float_as_uint(v1), __float_as_uint(v0), 0x07060302u);
        XF[ks][mt][1][J] = __builtin_amdgcn_perm(__float_as_uint(sr1[u]), __float_as_uint(sr0[u]), 0x07060302u);
        XF[ks][mt][2][J] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(ss0[u]), 0x07060302u);
        KGCN_PIN3(XF[ks][mt][0][J], XF[ks][mt][1][J], XF[ks][mt][2][J]);
      }
    };
    auto read_bf = [&](u32x4 (&BF)[2][3], auto ksc, auto ntc, auto pcc, unsigned buf_off) __attribute__((always_inline)) {
      constexpr int ks = decltype(ksc)::value, nt = decltype(ntc)::value, pc = decltype(pcc)::value;
      const u32x2 r0 = lds_ld_tr16(TRB[nt][0] + buf_off + (ks << 11) + (nt << 8) + pc * PL_BYTES);
      const u32x2 r1 = lds_ld_tr16(TRB[nt][1] + buf_off + (ks << 11) + (1 << 9) + (nt << 8) + pc * PL_BYTES);
      BF[nt][pc][0] = r0[0]; BF[nt][pc][1] = r0[1]; BF[nt][pc][2] = r1[0]; BF[nt][pc][3] = r1[1];
    };
    // finished aggregation pass -> planes of buffer `eb` (lane constant ^ row word), dbias
    auto emit_to = [&](unsigned lx, const PlaneSteps& q) __attribute__((always_inline)) {
      unsigned q1, q2, q3, r1, r2, r3;
      split_pair(q.a[0], q.a[1], q1, q2, q3);
      split_pair(q.a[2], q.a[3], r1, r2, r3);
      const unsigned ad = q.rw ^ lx;
      lds_st64(ad, q1, r1);
      lds_st64(ad + PL_BYTES, q2, r2);
      lds_st64(ad + 2 * PL_BYTES, q3, r3);
      add4(dbacc, q.a);
      KGCN_PIN4(dbacc[0], dbacc[1], dbacc[2], dbacc[3]);
    };

    // ---- prologue: graph 0 aggregated without overlap; the pipeline state of iteration 0 -------
    issue_meta(m_a, slots_t, gptr_t, gidx(0), N, lane);
    int base_a = meta_base(m_a), cnt_a = meta_cnt(m_a);
    issue_tile<true>(gpf, g + (long)gidx(0) * N * D, 512, lane);
    issue_cv(cpf, cv_t, base_a, cnt_a, lane);
    issue_x(0);
    issue_meta(m_b, slots_t, gptr_t, gidx(1), N, lane);
    land_tile<true>(gpf, gt, FD, 512, 16, lane);
    land_csr(cpf, ecv0, tab0, cv_t, slot_plane_word(m_a.slot), base_a, cnt_a, N, lane);
    wave_sync();
    // g(1), CSR(1) in flight
    m_a = m_b;
    base_a = meta_base(m_a);
    cnt_a = meta_cnt(m_a);
    issue_tile<true>(gpf, g + (long)gidx(1) * N * D, 512, lane);
    issue_cv(cpf, cv_t, base_a, cnt_a, lane);
    issue_meta(m_b, slots_t, gptr_t, gidx(2), N, lane);
    {
      const unsigned lx0 = lanexor | pl0;
#pragma unroll
      for (int p8 = 0; p8 < 8; ++p8) {
        PlaneSteps ps;
        ps.slot(tab0, 4 * p8 + sub);
        ps.q0 = *reinterpret_cast<const i32x4*>(ecv0 + ps.s);
        ps.q1 = *reinterpret_cast<const i32x4*>(ecv0 + ps.s + 2);
        ps.x0 = ldv4(srcl + ps.q0.x * FD); ps.x1 = ldv4(srcl + ps.q0.z * FD);
        ps.x2 = ldv4(srcl + ps.q1.x * FD); ps.x3 = ldv4(srcl + ps.q1.z * FD);
        const float v = __int_as_float(ps.q0.y);
        ps.a[0] = v * ps.x0[0]; ps.a[1] = v * ps.x0[1]; ps.a[2] = v * ps.x0[2]; ps.a[3] = v * ps.x0[3];
        fma4(ps.a, __int_as_float(ps.q0.w), ps.x1);
        fma4(ps.a, __int_as_float(ps.q1.y), ps.x2);
        fma4(ps.a, __int_as_float(ps.q1.w), ps.x3);
        ps.tail(ecv0, srcl);
        emit_to(lx0, ps);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    wave_sync();
    // x(0) -> fragments, x(1) in flight; g(1), CSR(1) -> LDS (buffer 1); g(2), CSR(2) in flight
    static_for<16>([&](auto uc) __attribute__((always_inline)) {
      split_x_half(uc, std::integral_constant<int, 0>{});
      split_x_half(uc, std::integral_constant<int, 1>{});
    });
    __builtin_amdgcn_sched_barrier(0);
    issue_x(1);
    land_tile<true>(gpf, gt, FD, 512, 16, lane);
    land_csr(cpf, ecv0 + ecv_stride, tab0 + (FN + 4), cv_t, slot_plane_word(m_a.slot), base_a, cnt_a, N, lane);
    m_a = m_b;
    base_a = meta_base(m_a);
    cnt_a = meta_cnt(m_a);
    issue_tile<true>(gpf, g + (long)gidx(2) * N * D, 512, lane);
    issue_cv(cpf, cv_t, base_a, cnt_a, lane);
    issue_meta(m_b, slots_t, gptr_t, gidx(3), N, lane);
    static_for<6>([&](auto c) __attribute__((always_inline)) {
      constexpr int v = decltype(c)::value;
      read_bf(BF0, std::integral_constant<int, 0>{}, std::integral_constant<int, v / 3>{},
              std::integral_constant<int, v % 3>{}, 0u);
    });
    wave_sync();

    f32x16 c0 = {}, c1 = {};   // dX(i)^T, produced in phase B(i), stored behind the first MFMAs of phase A(i+1)
    u32x4 FA[4][3];            // phase B: dFW(i) fragments [k-step][piece]
    u32x4 WL[4][2][2];         // phase B: W p2 / p3 fragments [k-step][piece - 1][tile]
    auto read_fa_at = [&](auto ksc, unsigned buf_off) __attribute__((always_inline)) {
      constexpr int ks = decltype(ksc)::value;
      static_for<3>([&](auto pc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        FA[ks][p] = lds_ld128(LB[ks] + buf_off + ((ks >> 1) << 8) + p * PL_BYTES);
      });
      static_for<4>([&](auto vc) __attribute__((always_inline)) {
        constexpr int pc = decltype(vc)::value >> 1, nt = decltype(vc)::value & 1;
        WL[ks][pc][nt] = lds_ld128(wtab + (pc * 8 + nt * 4 + ks) * 1024);
      });
    };
    int cur = 0;               // plane / CSR buffer of the MFMA graph i; the aggregated graph i+1 uses cur^1

    PROBE_DECL
    // ---- phase A: dW(i) MFMAs  ||  aggregation of graph i+1 -> planes(cur^1)  ||  stores of dX(i-1) ----------
    auto phase_a = [&](auto agg_tag, auto st_tag, int iprev) __attribute__((always_inline)) {
      constexpr bool AGG = decltype(agg_tag)::value, ST = decltype(st_tag)::value;
      float* dxp = dx + (long)gidx(iprev) * N * D + li * D + 4 * hi;
      const unsigned cur_off = cur ? (unsigned)DFWP_BYTES : 0u;
      const unsigned lx = lanexor | (pl0 + (cur ? 0u : (unsigned)DFWP_BYTES));
      const int2* ecv_n = ecv0 + (cur ^ 1) * ecv_stride;
      const int* tab_n = tab0 + (cur ^ 1) * (FN + 4);
      PlaneSteps qa, qb;
      auto emit = [&](const PlaneSteps& q) __attribute__((always_inline)) { emit_to(lx, q); };
      auto agg_half = [&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value, gq = j >> 4, wq = j & 15;
        if constexpr ((wq & 1) == 0) plane_half(qa, wq >> 1, gq > 0, tab_n, 8 * gq + sub, ecv_n, srcl, emit);
        else plane_half(qb, wq >> 1, gq > 0, tab_n, 8 * gq + 4 + sub, ecv_n, srcl, emit);
        if constexpr (wq == 14) qa.tail(ecv_n, srcl);
        if constexpr (wq == 15) qb.tail(ecv_n, srcl);
      };
      u32x4 BF1[2][3];
      static_for<2>([&](auto ksc) __attribute__((always_inline)) {
        constexpr int ks = decltype(ksc)::value;
        static_for<24>([&](auto mc) __attribute__((always_inline)) {
          constexpr int m = decltype(mc)::value, pr = m >> 2, tile = m & 3;
          // smallest terms first; k-step 1 takes the dFW pieces in the order p3, p2, p1 so that its fragments can be read
          // into the registers the pieces of k-step 0 leave (p3 after product 2, p2 after product 4)
          constexpr int PA0[6] = {2, 1, 0, 1, 0, 0}, PB0[6] = {0, 1, 2, 0, 1, 0};
          constexpr int PA1[6] = {0, 1, 2, 0, 1, 0}, PB1[6] = {2, 1, 0, 1, 0, 0};
          constexpr int pa = ks == 0 ? PA0[pr] : PA1[pr], pb = ks == 0 ? PB0[pr] : PB1[pr];
          const u32x4 av = XF[ks][tile >> 1][pa];
          const u32x4 bv = ks == 0 ? BF0[tile & 1][pb] : BF1[tile & 1][pb];
          if constexpr (tile == 0) dw00 = mfma_bf16(av, bv, dw00);
          else if constexpr (tile == 1) dw01 = mfma_bf16(av, bv, dw01);
          else if constexpr (tile == 2) dw10 = mfma_bf16(av, bv, dw10);
          else dw11 = mfma_bf16(av, bv, dw11);
          if constexpr (ks == 0 && (m == 12 || m == 13 || m == 20 || m == 21 || m == 22 || m == 23)) {
            constexpr int pc = m < 14 ? 2 : m < 22 ? 1 : 0;   // fragments of k-step 1: p3, p2, p1 (one tile per MFMA)
            read_bf(BF1, std::integral_constant<int, 1>{}, std::integral_constant<int, (m & 1)>{},
                    std::integral_constant<int, pc>{}, cur_off);
          }
          if constexpr (ST && ks == 0 && m >= 2 && m < 10) {   // dX(i-1): one 16-byte store per MFMA
            constexpr int v = m - 2, q = v & 3, nt = v >> 2;
            const f32x16& c = nt ? c1 : c0;
            const f32x4 val = {c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};
            stv4(dxp + 32 * nt + 8 * q, val);
          }
          if constexpr (AGG) agg_half(std::integral_constant<int, 32 * ks + m>{});
          __builtin_amdgcn_sched_barrier(0);
        });
#ifndef KGCN_PROBE_TAIL
        if constexpr (ks == 0) { PROBE(6) }
#endif
        if constexpr (AGG) {
          static_for<8>([&](auto rc) __attribute__((always_inline)) {
            agg_half(std::integral_constant<int, 32 * ks + 24 + decltype(rc)::value>{});
          });
        }
#ifndef KGCN_PROBE_TAIL
        if constexpr (ks == 0) { PROBE(7) }
#endif
      });
      read_fa_at(std::integral_constant<int, 0>{}, cur_off);   // phase B(i), k-step 0: in flight behind the last emits
      if constexpr (AGG) {
        emit(qa);
        emit(qb);
      }
      wave_sync();
    };

    // ---- phase B: dX(i)^T MFMAs  ||  x(i+1) -> fragments, x(i+2) loads  ||  g(i+2), CSR(i+2) -> LDS, g(i+3), CSR(i+3)
    //      loads  ||  fragments of dFW(i+1), k-step 0 ---------------------------------------------------------------
    auto phase_b = [&](auto mv_tag, int i) __attribute__((always_inline)) {
      constexpr bool MV = decltype(mv_tag)::value;
      const unsigned cur_off = cur ? (unsigned)DFWP_BYTES : 0u;
      const unsigned nxt_off = cur ? 0u : (unsigned)DFWP_BYTES;
      int2* ecv_c = ecv0 + cur * ecv_stride;     // CSR(i) is dead: receives CSR(i+2)
      int* tab_c = tab0 + cur * (FN + 4);
      const float* xb = x + (long)gidx(i + 2) * N * D + xlane;
      const float* gsrc = g + (long)gidx(i + 3) * N * D + 4 * lane;
      int base_n = 0, cnt_n = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
      auto read_fa = [&](auto ksc) __attribute__((always_inline)) { read_fa_at(ksc, cur_off); };
      static_for<48>([&](auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value, ks = m / 12, pr = (m % 12) >> 1, nt = m & 1;
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
        constexpr int wl = PB[pr] == 0 ? 0 : PB[pr] - 1;
        const u32x4 wv = PB[pr] == 0 ? WF1[nt][ks] : WL[ks][wl][nt];
        if constexpr (nt == 0) c0 = mfma_bf16(wv, FA[ks][PA[pr]], c0);
        else c1 = mfma_bf16(wv, FA[ks][PA[pr]], c1);
        if constexpr (m % 12 == 1 && ks < 3) read_fa(std::integral_constant<int, ks + 1>{});   // one k-step ahead
        // x(i+1) -> fragments: one half split_pair per MFMA (a whole one is 11 VALU ops: more than the 32 cycles of an
        // MFMA cover); the two 8-byte registers of a node pair are reloaded with x(i+2) as soon as both tiles are split
        if constexpr (MV && m < 32)
          split_x_half(std::integral_constant<int, (m >> 1)>{}, std::integral_constant<int, (m & 1)>{});
        if constexpr (MV && m >= 4 && m < 36 && ((m & 3) < 2)) {
          constexpr int P = (m >> 2) - 1, ks_ = P >> 2, J = P & 3;
          load_x(xb, std::integral_constant<int, 8 * ks_ + 2 * J + (m & 3)>{});
        }
        if constexpr (MV && m >= 36 && m < 44) {          // g(i+2) -> gather tile; g(i+3) in flight
          constexpr int q = m - 36;
          stv4(gt + 4 * lane + q * 256, gpf.v[q]);
          gpf.v[q] = ldv4((q >> 2 ? gsrc + 1024 : gsrc) + (q & 3) * 256);
        }
        if constexpr (MV && m == 44)
          land_csr(cpf, ecv_c, tab_c, cv_t, slot_plane_word(m_a.slot), base_a, cnt_a, N, lane);
        if constexpr (MV && m == 45) {
          base_n = meta_base(m_b);
          cnt_n = meta_cnt(m_b);
          issue_cv(cpf, cv_t, base_n, cnt_n, lane);
        }
        if constexpr (MV && m == 46) {
          // a REAL register move (a plain `m_a = m_b` becomes a loop phi whose copy lands behind the new load of m_b:
          // s_waitcnt vmcnt(~0) on every prefetch load of this phase at the top of the next iteration)
          asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3"
                       : "=&v"(m_a.slot), "=&v"(m_a.gp) : "v"(m_b.slot), "v"(m_b.gp));
          issue_meta(m_b, slots_t, gptr_t, gidx(i + 4), N, lane);
        }
#ifdef KGCN_PROBE_TAIL
        if constexpr (m == 35) { PROBE(1) }
        if constexpr (m == 39) { PROBE(2) }
        if constexpr (m == 43) { PROBE(3) }
        if constexpr (m == 44) { PROBE(4) }
        if constexpr (m == 45) { PROBE(6) }
        if constexpr (m == 46) { PROBE(7) }
#else
        if constexpr (m == 11) { PROBE(2) }
        if constexpr (m == 23) { PROBE(3) }
        if constexpr (m == 35) { PROBE(4) }
#endif
        if constexpr (MV && m >= 42) {                    // dFW(i+1), k-step 0: fragments for phase A(i+1)
          constexpr int v = m - 42;
          read_bf(BF0, std::integral_constant<int, 0>{}, std::integral_constant<int, v / 3>{},
                  std::integral_constant<int, v % 3>{}, nxt_off);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (MV) {
        base_a = base_n;
        cnt_a = cnt_n;
      }
      wave_sync();
    };

    // iterations: MFMA graph i, aggregated graph i+1, stores of dX(i-1)
    using Y = std::true_type;
    using Nn = std::false_type;
    if (cntw > 1) {
      phase_a(Y{}, Nn{}, 0);
      phase_b(Y{}, 0);
      cur ^= 1;
      PROBE(0)
      // The peeled first iteration needs more registers than the loop, so the loop's fragment registers come back from
      // scratch in the preheader.  A reload that is still pending at the loop header makes the compiler put its wait THERE,
      // where the back edge shares it -- as vmcnt(0): it drained the prefetch of every iteration.  A use in front of the
      // loop moves the wait out of it.
      asm volatile("" : "+v"(XF[0][0][0]), "+v"(XF[0][0][1]), "+v"(XF[0][0][2]), "+v"(XF[0][1][0]), "+v"(XF[0][1][1]),
                        "+v"(XF[0][1][2]), "+v"(XF[1][0][0]), "+v"(XF[1][0][1]), "+v"(XF[1][0][2]), "+v"(XF[1][1][0]),
                        "+v"(XF[1][1][1]), "+v"(XF[1][1][2]));
      asm volatile("" : "+v"(BF0[0][0]), "+v"(BF0[0][1]), "+v"(BF0[0][2]), "+v"(BF0[1][0]), "+v"(BF0[1][1]),
                        "+v"(BF0[1][2]), "+v"(FA[0][0]), "+v"(FA[0][1]), "+v"(FA[0][2]));
      for (int i = 1; i < cntw - 1; ++i) {
        phase_a(Y{}, Y{}, i - 1);
        PROBE(1)
        phase_b(Y{}, i);
        PROBE(5)
        cur ^= 1;
      }
      // last graph i = cntw-1: its dFW is ready, nothing left to aggregate or prefetch
      phase_a(Nn{}, Y{}, cntw - 2);
      phase_b(Nn{}, cntw - 1);
    } else {
      phase_a(Nn{}, Nn{}, 0);
      phase_b(Nn{}, 0);
    }
    PROBE_FLUSH(blockIdx.x * BWD_FULL_WPB + wave)
    {                                           // dX of the last graph
      float* dxp = dx + (long)tl * N * D + li * D + 4 * hi;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v0 = {c0[4 * q], c0[4 * q + 1], c0[4 * q + 2], c0[4 * q + 3]};
        const f32x4 v1 = {c1[4 * q], c1[4 * q + 1], c1[4 * q + 2], c1[4 * q + 3]};
        stv4(dxp + 8 * q, v0);
        stv4(dxp + 32 + 8 * q, v1);
      }
    }
  }

  // ---- reduce the workgroup's 4 waves through LDS; one partial per workgroup ---------------------
  __syncthreads();
  float* park = reinterpret_cast<float*>(sl);   // the plane buffers: 24,576 B >= (4096 + 64) floats
  static_assert(2 * DFWP_BYTES >= (FD * FD + FD) * 4, "park area");
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    park[(2 * row) * FD + li] = dw00[r];            // row `row` of M tile mt is input feature 2 row + mt
    park[(2 * row) * FD + 32 + li] = dw01[r];
    park[(2 * row + 1) * FD + li] = dw10[r];
    park[(2 * row + 1) * FD + 32 + li] = dw11[r];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float v = dbacc[j];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    dbacc[j] = v;
  }
  if (lane < 16) stv4(park + FD * FD + lane * 4, dbacc);
  __syncthreads();
  const size_t slice_f = 2 * DFWP_BYTES / 4;
  const float* slice0 = reinterpret_cast<const float*>(smem);
  float* pw = part_dw + (long)blockIdx.x * D * D;
  for (int i = tid; i < D * D; i += blockDim.x) {
    float s = 0.f;
    for (int wv = 0; wv < BWD_FULL_WPB; ++wv) s += slice0[wv * slice_f + i];
    pw[i] = s;
  }
  if (tid < D) {
    float s = 0.f;
    for (int wv = 0; wv < BWD_FULL_WPB; ++wv) s += slice0[wv * slice_f + D * D + tid];
    part_db[(long)blockIdx.x * D + tid] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// backward, FULL shape, "pairs" version (round 6): TWO waves share one graph slot -- the planes kernel's LDS
// (plane buffers, gather tile, CSR slices: ~36 KB per graph in flight) now serves a PAIR, eight waves per CU, two per SIMD, 256
// registers each, in two roles:
//   wave A ("dX")  g + CSR in (one iteration ahead), the adjoint aggregation dFW(i+1) = A^T g(i+1) -> three bf16 planes + dbias,
//                  dX(i)^T = W dFW(i)^T (48 MFMAs: W p1 resident, p2 / p3 from the workgroup's table), the dX stores
//   wave B ("dW")  x in (fragment layout, one iteration ahead), its 3-way split into resident fragments, dW += x(i)^T dFW(i)
//                  (48 MFMAs against the planes read transposed), the four dW accumulator tiles
// and ONE workgroup barrier per graph (s_waitcnt lgkmcnt(0) + s_barrier: vector-memory operations stay in flight across it): at the
// barrier that ends iteration i, planes[(i+1)&1] hold dFW(i+1) for both roles and nobody reads planes[i&1] any more.
// Why (tools/bwd_skeleton.hip, profiles/r06_microbench_bwd_memory_skeleton.txt): the memory system streams this byte pattern at
// 0.71 of 8 TB/s when the two read streams come from different waves that meet at a barrier (H: 0.447 ms per 100,000 graphs),
// at 0.66-0.685 when every wave moves its graph alone (P / PX: 0.466-0.48 ms, the planes kernel's structure, whatever the
// occupancy) -- and a lone wave per SIMD has nobody to fill its issue slots (SQ_WAIT_INST_ANY 0.33 of the wave cycles).
// With two instruction streams per SIMD the phases are NOT interleaved by hand any more: A aggregates (vector ALU + LDS) while
// B multiplies, A multiplies while B splits x.
// Iteration counts are uniform (every wave passes the same barriers): the launcher takes this kernel for T >= 2 pairs-in-flight
// graphs only, so a pair owns cnt_max or cnt_max - 1 >= 2 graphs; the one iteration a short pair lacks is skipped under uniform
// branches that hold no vector-memory instruction.
// ------------------------------------------------------------------------------------------------
constexpr int BP_PAIRS = 4;
#ifndef KGCN_BWD_PAIRS
#define KGCN_BWD_PAIRS 1        // 0: the one-wave-per-graph planes kernel for every batch
#endif
#ifndef KGCN_BP_B_GROUPS
#define KGCN_BP_B_GROUPS 5      // bit mask of the row groups (four of eight rows each, longest rows first) role B aggregates; role A the others
#endif
#ifndef KGCN_BP_AGG_SCOPE
#define KGCN_BP_AGG_SCOPE 0     // 0: one scheduling scope per row group (two passes in flight); 1: a role's groups in ONE scope
#endif
#ifndef KGCN_BP_PAIR_SYNC
#define KGCN_BP_PAIR_SYNC 0     // 1 (development): pair-local rendezvous instead of the workgroup barrier per graph
#endif
#ifndef KGCN_BP_A_ORDER
#define KGCN_BP_A_ORDER 1       // role A: 0 aggregate, multiply, land; 1 multiply, aggregate, land
#endif
#ifndef KGCN_BP_B_ORDER
#define KGCN_BP_B_ORDER 1       // role B: 0 multiply, aggregate, split; 1 aggregate, multiply, split
#endif
#ifndef KGCN_BP_ROLE_BIT
#define KGCN_BP_ROLE_BIT 2      // 2: pair = wave & 3, role = wave >> 2 (the two roles of a pair share a SIMD); 0: pair = wave >> 1, role = wave & 1
#endif
__device__ __forceinline__ void bp_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(512, 2) void graphconv_bwd_pairs_kernel(
    const int* __restrict__ slots_t, const int* __restrict__ gptr_t, const int2* __restrict__ cv_t,
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ g,
    float* __restrict__ dx, float* __restrict__ part_dw, float* __restrict__ part_db, int T,
    int max_nnz) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int N = FN, D = FD;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int pair = KGCN_BP_ROLE_BIT == 2 ? (wave & 3) : (wave >> 1);
  const int role = KGCN_BP_ROLE_BIT == 2 ? (wave >> 2) : (wave & 1);
  const int li = lane & 31, hi = lane >> 5;
  const int sub = lane >> 4, cl = lane & 15;
  static_assert(BWD_WPB_ == BP_PAIRS, "one graph slot of the planes layout per pair");
  unsigned char* sl = smem + (size_t)pair * 2 * DFWP_BYTES;
  const unsigned pl0 = lds_off(sl);                               // dFW planes, buffer 0 (buffer 1: + DFWP_BYTES)
  if (pl0 & 4095u) __builtin_trap();                              // XOR addressing needs 4 KiB aligned plane buffers
  const unsigned wtab = lds_off(smem + PLANES_ALL) + (unsigned)lane * 16;
  float* gt = reinterpret_cast<float*>(smem + PLANES_ALL + WTAB_BYTES +
                                       (size_t)pair * bwd_planes_wave_bytes(max_nnz));   // [FN+1][FD], row FN stays zero
  int2* ecv0 = reinterpret_cast<int2*>(gt + (FN + 1) * FD);
  const size_t ecv_stride = ecv_bytes(max_nnz) / 8;
  int* tab0 = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(ecv0) + 2 * ecv_bytes(max_nnz));

  const int npairs = gridDim.x * BP_PAIRS;
  const int t0 = blockIdx.x * BP_PAIRS + pair;
  const int cnt_max = (T - 1) / npairs + 1;                       // iterations (= barriers) of EVERY wave; >= 2 (launcher)
  const int cntw = (T - 1 - t0) / npairs + 1;                     // graphs of this pair: cnt_max or cnt_max - 1
  const int tl = t0 + (cntw - 1) * npairs;                        // its last graph
  const bool whole = cntw == cnt_max;                             // uniform
#ifdef KGCN_ABL_HOT                                                 // development: two graphs per pair, cache resident -- the kernel without HBM
  auto gidx = [&](int k) { const int t = t0 + (k & 1) * npairs; return t < tl ? t : tl; };
#else
  auto gidx = [&](int k) { const int t = t0 + k * npairs; return t < tl ? t : tl; };  // clamped
#endif
  PROBE_DECL

  f32x16 dw00, dw01, dw10, dw11;                                  // role B
  f32x4 dbacc = {0.f, 0.f, 0.f, 0.f};                             // both roles (each aggregates half of a graph's rows)
#pragma unroll
  for (int r = 0; r < 16; ++r) { dw00[r] = 0.f; dw01[r] = 0.f; dw10[r] = 0.f; dw11[r] = 0.f; }

  // ---- the adjoint aggregation, shared by the roles: role B takes the row groups {0, 2} of graph i + 1 (the longest rows: the slot
  //      table is ordered by decreasing length), role A {1, 3}; both write the planes of buffer (i + 1) & 1 ----------------------
  const unsigned h_e = (unsigned)cl >> 3;
  const unsigned lanexor = (h_e << 8) | (h_e << 6) | ((((unsigned)cl & 7) >> 1) << 4) | (((unsigned)cl & 1) << 3);
  const float* srcl = gt + cl * 4;
  auto emit_to = [&](unsigned lx, const PlaneSteps& q) __attribute__((always_inline)) {
    unsigned q1, q2, q3, r1, r2, r3;
    split_pair(q.a[0], q.a[1], q1, q2, q3);
    split_pair(q.a[2], q.a[3], r1, r2, r3);
    const unsigned ad = q.rw ^ lx;
    lds_st64(ad, q1, r1);
    lds_st64(ad + PL_BYTES, q2, r2);
    lds_st64(ad + 2 * PL_BYTES, q3, r3);
    add4(dbacc, q.a);
  };
  // row groups gq = G0 .. G1 - 1 of the graph in the gather tile -> planes at `plane_off`, two passes in flight
  auto aggregate = [&](auto maskc, const int2* ecv, const int* tab, unsigned plane_off) __attribute__((always_inline)) {
    const unsigned lx = lanexor | (pl0 + plane_off);
    static_for<4>([&](auto gqc) __attribute__((always_inline)) {
      constexpr int gq = decltype(gqc)::value;
      if constexpr (((decltype(maskc)::value >> gq) & 1) != 0) {
      PlaneSteps qa, qb;
      qa.slot(tab, 8 * gq + sub);
      qb.slot(tab, 8 * gq + 4 + sub);
      qa.q0 = *reinterpret_cast<const i32x4*>(ecv + qa.s);
      qa.q1 = *reinterpret_cast<const i32x4*>(ecv + qa.s + 2);
      qb.q0 = *reinterpret_cast<const i32x4*>(ecv + qb.s);
      qb.q1 = *reinterpret_cast<const i32x4*>(ecv + qb.s + 2);
      qa.x0 = ldv4(srcl + qa.q0.x * FD); qa.x1 = ldv4(srcl + qa.q0.z * FD);
      qa.x2 = ldv4(srcl + qa.q1.x * FD); qa.x3 = ldv4(srcl + qa.q1.z * FD);
      qb.x0 = ldv4(srcl + qb.q0.x * FD); qb.x1 = ldv4(srcl + qb.q0.z * FD);
      qb.x2 = ldv4(srcl + qb.q1.x * FD); qb.x3 = ldv4(srcl + qb.q1.z * FD);
      {
        const float v = __int_as_float(qa.q0.y);
        qa.a[0] = v * qa.x0[0]; qa.a[1] = v * qa.x0[1]; qa.a[2] = v * qa.x0[2]; qa.a[3] = v * qa.x0[3];
        fma4(qa.a, __int_as_float(qa.q0.w), qa.x1);
        fma4(qa.a, __int_as_float(qa.q1.y), qa.x2);
        fma4(qa.a, __int_as_float(qa.q1.w), qa.x3);
      }
      {
        const float v = __int_as_float(qb.q0.y);
        qb.a[0] = v * qb.x0[0]; qb.a[1] = v * qb.x0[1]; qb.a[2] = v * qb.x0[2]; qb.a[3] = v * qb.x0[3];
        fma4(qb.a, __int_as_float(qb.q0.w), qb.x1);
        fma4(qb.a, __int_as_float(qb.q1.y), qb.x2);
        fma4(qb.a, __int_as_float(qb.q1.w), qb.x3);
      }
      qa.tail(ecv, srcl);
      qb.tail(ecv, srcl);
      emit_to(lx, qa);
      emit_to(lx, qb);
      if constexpr (KGCN_BP_AGG_SCOPE == 0) __builtin_amdgcn_sched_barrier(0);
      }
    });
    if constexpr (KGCN_BP_AGG_SCOPE != 0) __builtin_amdgcn_sched_barrier(0);
  };
  using MALL = std::integral_constant<int, 15>;
  using MB = std::integral_constant<int, KGCN_BP_B_GROUPS>;                  // role B's row groups (bit mask), role A's: the others
  using MA = std::integral_constant<int, 15 & ~KGCN_BP_B_GROUPS>;
  constexpr int NB = KGCN_BP_B_GROUPS;                                          // (0: role A aggregates alone)
  constexpr int NA = 15 & ~KGCN_BP_B_GROUPS;
  // hand-over inside the pair: role B has finished READING the gather tile for graph i + 1 (flag = i + 1) before role A lands
  // g(i + 2) in it.  LDS executes a wave's operations in order, so the flag write follows B's last gather read.
  KGCN_LDS volatile int* const agg_flag = (KGCN_LDS volatile int*)(uintptr_t)lds_off(tab0 + FN + 2);
  // development (KGCN_BP_PAIR_SYNC): the end-of-iteration rendezvous between the TWO waves of a pair only (arrival counters in LDS)
  // instead of the workgroup barrier that also ties the four pairs together
  KGCN_LDS volatile int* const arrive_mine = (KGCN_LDS volatile int*)(uintptr_t)lds_off(tab0 + (FN + 4) + FN + 2 + role);
  KGCN_LDS volatile int* const arrive_other = (KGCN_LDS volatile int*)(uintptr_t)lds_off(tab0 + (FN + 4) + FN + 2 + (role ^ 1));
  auto iter_sync = [&](int i) __attribute__((always_inline)) {
    if constexpr (KGCN_BP_PAIR_SYNC != 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) *arrive_mine = i + 1;
      while (*arrive_other < i + 1) __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
    } else {
      bp_barrier();
    }
  };

  if (role == 0) {
    // =========================================== role A ===========================================================
    for (int i = lane; i < D; i += 64) gt[FN * FD + i] = 0.f;
    unsigned LB[4];                                               // row reads: lane (li, hi): features 16 ks + 8 hi .. + 7 of node li
    {
      const unsigned R = (unsigned)li >> 2, a = (unsigned)li & 3;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        LB[ks] = pl0 + ((R << 9) | ((a ^ (unsigned)(ks >> 1)) << 6) |
                        (((2 * (unsigned)(ks & 1) + (unsigned)hi) ^ (R & 3)) << 4));
    }
    // A fragments of dX^T = W dFW^T (as in the planes kernel): p1 resident, p2 / p3 in the workgroup's table
    u32x4 WF1[2][4];
    static_for<8>([&](auto c) __attribute__((always_inline)) {
      constexpr int nt = decltype(c)::value >> 2, ks = decltype(c)::value & 3;
      const float* src = w + (32 * nt + li) * D + 16 * ks + 8 * hi;
      const f32x4 lo = ldv4(src), hi4 = ldv4(src + 4);
      const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
      Frag3 f;
      split8(v, f);
      WF1[nt][ks] = f.p1;
      if (pair == (decltype(c)::value >> 1)) {
        *(KGCN_LDS u32x4*)(uintptr_t)(wtab + (0 * 8 + nt * 4 + ks) * 1024) = f.p2;
        *(KGCN_LDS u32x4*)(uintptr_t)(wtab + (1 * 8 + nt * 4 + ks) * 1024) = f.p3;
      }
    });
    __syncthreads();                                              // barrier 0: the W table

    TileRegs gpf;
    CsrRegs cpf;
    MetaRegs m_a, m_b;   // m_a: graph whose g / CSR are in flight; m_b: the one after it
    // ---- prologue: graph 0 aggregated into planes 0; g(1), CSR(1) in LDS; g(2), CSR(2) in flight ---------------
    issue_meta(m_a, slots_t, gptr_t, gidx(0), N, lane);
    int base_a = meta_base(m_a), cnt_a = meta_cnt(m_a);
    issue_tile<true>(gpf, g + (long)gidx(0) * N * D, 512, lane);
    issue_cv(cpf, cv_t, base_a, cnt_a, lane);
    issue_meta(m_b, slots_t, gptr_t, gidx(1), N, lane);
    land_tile<true>(gpf, gt, FD, 512, 16, lane);
    land_csr(cpf, ecv0, tab0, cv_t, slot_plane_word(m_a.slot), base_a, cnt_a, N, lane);
    wave_sync();
    m_a = m_b;
    base_a = meta_base(m_a);
    cnt_a = meta_cnt(m_a);
    issue_tile<true>(gpf, g + (long)gidx(1) * N * D, 512, lane);
    issue_cv(cpf, cv_t, base_a, cnt_a, lane);
    issue_meta(m_b, slots_t, gptr_t, gidx(2), N, lane);
    aggregate(MALL{}, ecv0, tab0, 0u);                        // graph 0: all four row groups
    wave_sync();
    land_tile<true>(gpf, gt, FD, 512, 16, lane);
    land_csr(cpf, ecv0 + ecv_stride, tab0 + (FN + 4), cv_t, slot_plane_word(m_a.slot), base_a, cnt_a, N, lane);
    m_a = m_b;
    base_a = meta_base(m_a);
    cnt_a = meta_cnt(m_a);
    issue_tile<true>(gpf, g + (long)gidx(2) * N * D, 512, lane);
    issue_cv(cpf, cv_t, base_a, cnt_a, lane);
    issue_meta(m_b, slots_t, gptr_t, gidx(3), N, lane);
    if (lane == 0) *arrive_mine = 0;
    bp_barrier();                                                 // barrier 1: dFW(0) in planes 0

    f32x16 c0, c1;
    // g(i+2), CSR(i+2): registers -> LDS (the aggregation of graph i+1 has read the gather tile: same wave, in order);
    // g(i+3), CSR(i+3), slots(i+4) requested -- a whole iteration ahead of their landing
    auto land_and_prefetch = [&](int i, int cur) __attribute__((always_inline)) {
      const float* gsrc = g + (long)gidx(i + 3) * N * D + 4 * lane;
      int2* ecv_c = ecv0 + cur * ecv_stride;                      // CSR(i) is dead: receives CSR(i+2)
      int* tab_c = tab0 + cur * (FN + 4);
      static_for<8>([&](auto qc) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value;
        stv4(gt + 4 * lane + q * 256, gpf.v[q]);
        gpf.v[q] = ldv4((q >> 2 ? gsrc + 1024 : gsrc) + (q & 3) * 256);
      });
      land_csr(cpf, ecv_c, tab_c, cv_t, slot_plane_word(m_a.slot), base_a, cnt_a, N, lane);
      const int base_n = meta_base(m_b), cnt_n = meta_cnt(m_b);
      issue_cv(cpf, cv_t, base_n, cnt_n, lane);
      // a REAL register move (see the planes kernel: a plain `m_a = m_b` becomes a phi whose copy lands behind the new load)
      asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3"
                   : "=&v"(m_a.slot), "=&v"(m_a.gp) : "v"(m_b.slot), "v"(m_b.gp));
      issue_meta(m_b, slots_t, gptr_t, gidx(i + 4), N, lane);
      base_a = base_n;
      cnt_a = cnt_n;
    };
    // dX(i)^T = W dFW(i)^T out of planes `cur`, then its 16-byte stores
    auto dx_of = [&](int i, int cur) __attribute__((always_inline)) {
      const unsigned cur_off = cur ? (unsigned)DFWP_BYTES : 0u;
      u32x4 FA[4][3];            // dFW(i) fragments [k-step][piece]
      u32x4 WL[4][2][2];         // W p2 / p3 fragments [k-step][piece - 1][tile]
      auto read_fa = [&](auto ksc) __attribute__((always_inline)) {
        constexpr int ks = decltype(ksc)::value;
        static_for<3>([&](auto pc) __attribute__((always_inline)) {
          constexpr int p = decltype(pc)::value;
          FA[ks][p] = lds_ld128(LB[ks] + cur_off + ((ks >> 1) << 8) + p * PL_BYTES);
        });
        static_for<4>([&](auto vc) __attribute__((always_inline)) {
          constexpr int pc = decltype(vc)::value >> 1, nt = decltype(vc)::value & 1;
          WL[ks][pc][nt] = lds_ld128(wtab + (pc * 8 + nt * 4 + ks) * 1024);
        });
      };
      const f32x16 zero16 = {};                                   // (the first product of a chain takes the inline constant 0 as its C operand)
      read_fa(std::integral_constant<int, 0>{});
      static_for<48>([&](auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value, ks = m / 12, pr = (m % 12) >> 1, nt = m & 1;
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
        constexpr int wl = PB[pr] == 0 ? 0 : PB[pr] - 1;
        const u32x4 wv = PB[pr] == 0 ? WF1[nt][ks] : WL[ks][wl][nt];
        if constexpr (m == 0) c0 = mfma_bf16(wv, FA[ks][PA[pr]], zero16);
        else if constexpr (m == 1) c1 = mfma_bf16(wv, FA[ks][PA[pr]], zero16);
        else if constexpr (nt == 0) c0 = mfma_bf16(wv, FA[ks][PA[pr]], c0);
        else c1 = mfma_bf16(wv, FA[ks][PA[pr]], c1);
        if constexpr (m % 12 == 1 && ks < 3) {                     // one k-step ahead
          read_fa(std::integral_constant<int, ks + 1>{});
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      float* dxp = dx + (long)gidx(i) * N * D + li * D + 4 * hi;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v0 = {c0[4 * q], c0[4 * q + 1], c0[4 * q + 2], c0[4 * q + 3]};
        const f32x4 v1 = {c1[4 * q], c1[4 * q + 1], c1[4 * q + 2], c1[4 * q + 3]};
        stv4(dxp + 8 * q, v0);
        stv4(dxp + 32 + 8 * q, v1);
      }
    };

    int cur = 0;
    PROBE(0)
    for (int i = 0; i < cnt_max - 2; ++i) {                       // graphs i and i + 1 exist for every pair
      if constexpr (KGCN_BP_A_ORDER == 0) {
        if constexpr (NA != 0)
          aggregate(MA{}, ecv0 + (cur ^ 1) * ecv_stride, tab0 + (cur ^ 1) * (FN + 4), cur ? 0u : (unsigned)DFWP_BYTES);
        PROBE(1)
        dx_of(i, cur);
        PROBE(3)
      } else {                                                    // the multiplications of both roles first
        dx_of(i, cur);
        PROBE(3)
        if constexpr (NA != 0)
          aggregate(MA{}, ecv0 + (cur ^ 1) * ecv_stride, tab0 + (cur ^ 1) * (FN + 4), cur ? 0u : (unsigned)DFWP_BYTES);
        PROBE(1)
      }
      if constexpr (NB != 0) {
        while (*agg_flag < i + 1) __builtin_amdgcn_s_sleep(1);    // role B is through with the gather tile (normally long ago)
      }
      wave_sync();
      PROBE(6)
      land_and_prefetch(i, cur);
      PROBE(2)
      iter_sync(i);
      PROBE(4)
      cur ^= 1;
    }
    // iteration cnt_max - 2: graph cnt_max - 1 exists for a whole pair only
    if constexpr (NA != 0) {
      if (whole) aggregate(MA{}, ecv0 + (cur ^ 1) * ecv_stride, tab0 + (cur ^ 1) * (FN + 4), cur ? 0u : (unsigned)DFWP_BYTES);
    }
    wave_sync();
    dx_of(cnt_max - 2, cur);
    iter_sync(cnt_max - 2);
    cur ^= 1;
    // iteration cnt_max - 1
    if (whole) dx_of(cnt_max - 1, cur);
    PROBE(5)
  } else {
    // =========================================== role B ===========================================================
    // transpose reads: lane l addresses 4 features (16 nb + 4 a4 ..) of node 16 ks + 8 hi + 4 rd + jj
    const unsigned nb = ((unsigned)lane >> 4) & 1, jj = ((unsigned)lane >> 2) & 3, a4 = (unsigned)lane & 3;
    unsigned TRB[2][2];                                            // [feature half nt][read rd]
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int rd = 0; rd < 2; ++rd)
        TRB[nt][rd] = pl0 + (((unsigned)hi << 10) | ((jj ^ (unsigned)nt) << 6) |
                             (((2 * nb + (a4 >> 1)) ^ (2 * (unsigned)hi + (unsigned)rd)) << 4) | ((a4 & 1) << 3));
    const int xlane = (8 * hi) * D + 2 * li;                      // lane share of the x fragment addresses (floats)
    f32x2 xr[16];        // x tile in flight in the A-fragment layout of dW (see the planes kernel)
    float sr0[16], ss0[16], sr1[16], sv0[16], sv1[16];
    u32x4 XF[2][2][3];   // x(i) fragments [k-step][feature half mt][piece]
    auto load_x = [&](const float* xb, auto qc) __attribute__((always_inline)) {
      constexpr int q = decltype(qc)::value;
      xr[q] = *reinterpret_cast<const f32x2*>((q >> 3 ? xb + 16 * D : xb) + (q & 7) * D);
    };
    auto split_x_half = [&](auto uc, auto hc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value, hh = decltype(hc)::value, P = u >> 1, mt = u & 1, ks = P >> 2, J = P & 3;
      const unsigned msk = 0xffff0000u;
      if constexpr (hh == 0) {
        float v0, v1;                                             // (the tile in flight lives in the accumulator file: planes kernel)
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v0) : "a"(xr[8 * ks + 2 * J][mt]));
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v1) : "a"(xr[8 * ks + 2 * J + 1][mt]));
        sv0[u] = v0; sv1[u] = v1;
        const float h0 = __uint_as_float(__float_as_uint(v0) & msk), h1 = __uint_as_float(__float_as_uint(v1) & msk);
        sr0[u] = v0 - h0;
        sr1[u] = v1 - h1;
        ss0[u] = sr0[u] - __uint_as_float(__float_as_uint(sr0[u]) & msk);
      } else {
        const float v0 = sv0[u], v1 = sv1[u];
        const float s1 = sr1[u] - __uint_as_float(__float_as_uint(sr1[u]) & msk);
        XF[ks][mt][0][J] = __builtin_amdgcn_perm(__float_as_uint(v1), __float_as_uint(v0), 0x07060302u);
        XF[ks][mt][1][J] = __builtin_amdgcn_perm(__float_as_uint(sr1[u]), __float_as_uint(sr0[u]), 0x07060302u);
        XF[ks][mt][2][J] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(ss0[u]), 0x07060302u);
      }
    };
    // x(k) -> fragments; every node pair's two registers take x(k + 1) as soon as both of their tiles are split
    auto split_and_reload = [&](int knext) __attribute__((always_inline)) {
      const float* xb = x + (long)gidx(knext) * N * D + xlane;
      static_for<8>([&](auto pc) __attribute__((always_inline)) {
        constexpr int P = decltype(pc)::value, ks = P >> 2, J = P & 3;
        split_x_half(std::integral_constant<int, 2 * P>{}, std::integral_constant<int, 0>{});
        split_x_half(std::integral_constant<int, 2 * P + 1>{}, std::integral_constant<int, 0>{});
        split_x_half(std::integral_constant<int, 2 * P>{}, std::integral_constant<int, 1>{});
        split_x_half(std::integral_constant<int, 2 * P + 1>{}, std::integral_constant<int, 1>{});
        load_x(xb, std::integral_constant<int, 8 * ks + 2 * J>{});
        load_x(xb, std::integral_constant<int, 8 * ks + 2 * J + 1>{});
      });
    };
    auto read_bf = [&](u32x4 (&BF)[2][3], auto ksc, auto ntc, auto pcc, unsigned buf_off) __attribute__((always_inline)) {
      constexpr int ks = decltype(ksc)::value, nt = decltype(ntc)::value, pc = decltype(pcc)::value;
      const u32x2 r0 = lds_ld_tr16(TRB[nt][0] + buf_off + (ks << 11) + (nt << 8) + pc * PL_BYTES);
      const u32x2 r1 = lds_ld_tr16(TRB[nt][1] + buf_off + (ks << 11) + (1 << 9) + (nt << 8) + pc * PL_BYTES);
      BF[nt][pc][0] = r0[0]; BF[nt][pc][1] = r0[1]; BF[nt][pc][2] = r1[0]; BF[nt][pc][3] = r1[1];
    };
    // dW += x(i)^T dFW(i) out of planes `cur`
    auto dw_of = [&](int cur) __attribute__((always_inline)) {
      const unsigned cur_off = cur ? (unsigned)DFWP_BYTES : 0u;
      u32x4 BF0[2][3], BF1[2][3];
      static_for<6>([&](auto c) __attribute__((always_inline)) {
        constexpr int v = decltype(c)::value;
        read_bf(BF0, std::integral_constant<int, 0>{}, std::integral_constant<int, v / 3>{},
                std::integral_constant<int, v % 3>{}, cur_off);
      });
      static_for<2>([&](auto ksc) __attribute__((always_inline)) {
        constexpr int ks = decltype(ksc)::value;
        static_for<24>([&](auto mc) __attribute__((always_inline)) {
          constexpr int m = decltype(mc)::value, pr = m >> 2, tile = m & 3;
          constexpr int PA0[6] = {2, 1, 0, 1, 0, 0}, PB0[6] = {0, 1, 2, 0, 1, 0};
          constexpr int PA1[6] = {0, 1, 2, 0, 1, 0}, PB1[6] = {2, 1, 0, 1, 0, 0};
          constexpr int pa = ks == 0 ? PA0[pr] : PA1[pr], pb = ks == 0 ? PB0[pr] : PB1[pr];
          const u32x4 av = XF[ks][tile >> 1][pa];
          const u32x4 bv = ks == 0 ? BF0[tile & 1][pb] : BF1[tile & 1][pb];
          if constexpr (tile == 0) dw00 = mfma_bf16(av, bv, dw00);
          else if constexpr (tile == 1) dw01 = mfma_bf16(av, bv, dw01);
          else if constexpr (tile == 2) dw10 = mfma_bf16(av, bv, dw10);
          else dw11 = mfma_bf16(av, bv, dw11);
          if constexpr (ks == 0 && (m == 12 || m == 13 || m == 20 || m == 21 || m == 22 || m == 23)) {
            constexpr int pc = m < 14 ? 2 : m < 22 ? 1 : 0;   // fragments of k-step 1: p3, p2, p1 (one tile per MFMA)
            read_bf(BF1, std::integral_constant<int, 1>{}, std::integral_constant<int, (m & 1)>{},
                    std::integral_constant<int, pc>{}, cur_off);
          }
        });
      });
    };

    __syncthreads();                                              // barrier 0: the W table (role A)
    // ---- prologue: x(0) -> fragments, x(1) in flight ---------------------------------------------------------------
    {
      const float* xb = x + (long)gidx(0) * N * D + xlane;
      static_for<16>([&](auto qc) __attribute__((always_inline)) { load_x(xb, qc); });
    }
    split_and_reload(1);
    if (lane == 0) { *agg_flag = 0; *arrive_mine = 0; }
    bp_barrier();                                                 // barrier 1: dFW(0) in planes 0

    int cur = 0;
    PROBE(0)
    for (int i = 0; i < cnt_max - 1; ++i) {                       // graph i exists for every pair
      if constexpr (KGCN_BP_B_ORDER == 0) {
        dw_of(cur);                                               // (matrix pipe) while role A aggregates (vector ALU)
        PROBE(1)
      }
      if (NB != 0 && (i < cnt_max - 2 || whole)) {                // uniform; no vector-memory instruction inside
        aggregate(MB{}, ecv0 + (cur ^ 1) * ecv_stride, tab0 + (cur ^ 1) * (FN + 4), cur ? 0u : (unsigned)DFWP_BYTES);
        wave_sync();
        if (lane == 0) *agg_flag = i + 1;
      }
      PROBE(3)
      if constexpr (KGCN_BP_B_ORDER != 0) {
        dw_of(cur);
        PROBE(1)
      }
      split_and_reload(i + 2);                                    // x(i+1) -> fragments, x(i+2) requested
      PROBE(2)
      iter_sync(i);
      PROBE(4)
      cur ^= 1;
    }
    if (whole) dw_of(cur);                                        // iteration cnt_max - 1
    PROBE(5)
  }
  PROBE_FLUSH(blockIdx.x * 8 + wave)

  // ---- reduce the workgroup's pairs through LDS; one partial per workgroup ---------------------
  __syncthreads();
  float* park = reinterpret_cast<float*>(sl);   // the pair's plane buffers: 24,576 B >= (4096 + 128) floats
  if (role == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
      park[(2 * row) * FD + li] = dw00[r];            // row `row` of M tile mt is input feature 2 row + mt
      park[(2 * row) * FD + 32 + li] = dw01[r];
      park[(2 * row + 1) * FD + li] = dw10[r];
      park[(2 * row + 1) * FD + 32 + li] = dw11[r];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float v = dbacc[j];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    dbacc[j] = v;
  }
  if (lane < 16) stv4(park + FD * FD + role * FD + lane * 4, dbacc);      // [role A's rows | role B's rows]
  __syncthreads();
  const size_t slice_f = 2 * DFWP_BYTES / 4;
  const float* slice0 = reinterpret_cast<const float*>(smem);
  float* pw = part_dw + (long)blockIdx.x * D * D;
  for (int i = tid; i < D * D; i += blockDim.x) {
    float s = 0.f;
    for (int wv = 0; wv < BP_PAIRS; ++wv) s += slice0[wv * slice_f + i];
    pw[i] = s;
  }
  if (tid < D) {
    float s = 0.f;
    for (int wv = 0; wv < BP_PAIRS; ++wv) s += slice0[wv * slice_f + D * D + tid] + slice0[wv * slice_f + D * D + D + tid];
    part_db[(long)blockIdx.x * D + tid] = s;
  }
}

// waves per block that fit the LDS budget (0 = does not fit at all)
static int fused_wpb(size_t per_wave, size_t shared_bytes) {
  long fit = ((long)kLdsBytes - (long)shared_bytes) / (long)per_wave;
  if (fit > MAX_WPB) fit = MAX_WPB;
  return fit < 1 ? 0 : (int)fit;
}

static int fused_grid(int T, int wpb) {
  int blocks = (T + wpb - 1) / wpb;
  if (blocks > kNumCU) blocks = kNumCU;  // one persistent workgroup per CU
  return blocks < 1 ? 1 : blocks;
}

static bool is_full(int n, int din, int dout) { return n == FN && din == FD && dout == FD; }

// graphs per 32-row tile in the generic kernels (see "graph packing"): as many as fit, at most MAX_PACK, and only
// while the packed CSR slices still leave room for >= 4 waves per workgroup
static int pack_factor(int n, int max_nnz, size_t a_floats_shared) {
  int p = FN / n;
  if (p > MAX_PACK) p = MAX_PACK;
  while (p > 1) {
    const size_t per = slice_bytes(max_nnz * p, a_floats_shared == 0 ? A_FWD : A_BWD, 1);
    long fit = ((long)kLdsBytes - (long)a_floats_shared) / (long)per;
    if (fit >= 4 && (long)max_nnz * p < 65536) break;
    --p;
  }
  return p < 1 ? 1 : p;
}

// LDS per wave of the forward / backward kernel for a (row-padded) batch
static size_t fwd_slice(int, int, int, int max_nnz) { return slice_bytes(max_nnz, A_FWD, 1); }
static size_t fwd_shared(int n, int din, int dout) { return is_full(n, din, dout) ? FWD_FULL_SHARED : 0; }
static size_t bwd_slice(int max_nnz) { return slice_bytes(max_nnz, A_BWD, 1); }

static bool fused_shape_ok(int n, int din, int dout, int max_nnz) {
  if (n <= 0 || n > FN) return false;
  if (din <= 0 || din > FD) return false;
  if (dout <= 0 || dout > FD) return false;
  if (max_nnz < 0 || (max_nnz & 3)) return false;
  return fused_wpb(fwd_slice(n, din, dout, max_nnz), fwd_shared(n, din, dout)) >= 4 &&
         fused_wpb(bwd_slice(max_nnz), FD * FD * 4) >= 4;
}

template <typename K>
static void allow_big_lds(K kernel) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
}

static int check_padded(const kgcn_csr_batch* a, const char* who) {
  if (int rc = validate_csr(a, who, true)) return rc;
  if (a->row_pad != 4)
    return fail("%s: the fused kernels read the row-padded layout (row_pad = 4), got row_pad=%d", who,
                a->row_pad);
  if (a->num_graphs > 0 && (!a->slots || !a->graph_ptr))
    return fail("%s: row-padded batch without slots / graph_ptr", who);
  if (a->rows != a->cols) return fail("%s: adjacency must be square", who);
  return 0;
}

}  // namespace kgcn

using namespace kgcn;

#ifdef KGCN_PROBE
extern "C" int kgcn_probe_set(void* buf) {
  long long* p = static_cast<long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int kgcn_graphconv_fused_supported(int32_t n_nodes, int32_t din, int32_t dout,
                                              int32_t max_nnz_per_graph) {
  // max_nnz_per_graph of the ROW-PADDED batch (a multiple of 4); rounded up here for convenience
  return fused_shape_ok(n_nodes, din, dout, (max_nnz_per_graph + 3) & ~3) ? 1 : 0;
}

extern "C" int kgcn_graphconv_fwd_f32(const kgcn_csr_batch* a, const float* x, const float* w,
                                      const float* bias, int32_t din, int32_t dout, float* out,
                                      void* stream) {
  if (int rc = check_padded(a, "kgcn_graphconv_fwd_f32")) return rc;
  if (!fused_shape_ok(a->rows, din, dout, a->max_nnz_per_graph))
    return fail("kgcn_graphconv_fwd_f32: shape N=%d din=%d dout=%d max_nnz=%d not supported by the "
                "fused kernel (use kgcn_dense_fwd_f32 + kgcn_bconv_f32)",
                a->rows, din, dout, a->max_nnz_per_graph);
  if (a->num_graphs == 0) return 0;
  if (!x || !w || !out) return fail("kgcn_graphconv_fwd_f32: NULL operand");
  const bool vec = !(din & 3) && !(dout & 3);
  if ((vec && (!aligned16(x) || !aligned16(out))) || !aligned16(a->cv))
    return fail("kgcn_graphconv_fwd_f32: x/out/cv not 16-byte aligned");
  // tile movement mode (see land_tile_f4): 1 when every graph's [N x d] block is a whole number of aligned float4
  const int mode = vec ? 2 : (((a->rows * din) & 3) == 0 && ((a->rows * dout) & 3) == 0 && aligned16(x)) ? 1 : 0;
  const bool full_shape = is_full(a->rows, din, dout);
  const int pack = full_shape ? 1 : pack_factor(a->rows, a->max_nnz_per_graph, 0);
  const size_t per = fwd_slice(a->rows, din, dout, a->max_nnz_per_graph * pack);
  const size_t shared = fwd_shared(a->rows, din, dout);
  const int wpb = fused_wpb(per, shared);
  const size_t lds = shared + (size_t)wpb * per;
  const int tiles = (a->num_graphs + pack - 1) / pack;
  static thread_local bool attr_set = false;
  if (!attr_set) {
    allow_big_lds(graphconv_fwd_full_kernel);
    allow_big_lds(graphconv_fwd_kernel<2>);
    allow_big_lds(graphconv_fwd_kernel<1>);
    allow_big_lds(graphconv_fwd_kernel<0>);
    attr_set = true;
  }
  const dim3 grid(fused_grid(tiles, wpb)), block(64 * wpb);
  const int2* cv = reinterpret_cast<const int2*>(a->cv);
  if (full_shape)
    hipLaunchKernelGGL(graphconv_fwd_full_kernel, grid, block, lds, as_stream(stream), a->slots,
                       a->graph_ptr, cv, x, w, bias, out, a->num_graphs, a->max_nnz_per_graph);
  else if (mode == 2)
    hipLaunchKernelGGL(graphconv_fwd_kernel<2>, grid, block, lds, as_stream(stream), a->slots,
                       a->graph_ptr, cv, x, w, bias, out, a->num_graphs, a->rows, din, dout,
                       a->max_nnz_per_graph, pack);
  else if (mode == 1)
    hipLaunchKernelGGL(graphconv_fwd_kernel<1>, grid, block, lds, as_stream(stream), a->slots,
                       a->graph_ptr, cv, x, w, bias, out, a->num_graphs, a->rows, din, dout,
                       a->max_nnz_per_graph, pack);
  else
    hipLaunchKernelGGL(graphconv_fwd_kernel<0>, grid, block, lds, as_stream(stream), a->slots,
                       a->graph_ptr, cv, x, w, bias, out, a->num_graphs, a->rows, din, dout,
                       a->max_nnz_per_graph, pack);
  return check_launch("graphconv_fwd_kernel");
}

extern "C" int64_t kgcn_graphconv_bwd_workspace_bytes(int32_t num_graphs, int32_t din,
                                                      int32_t dout) {
  if (num_graphs <= 0 || din <= 0 || dout <= 0) return 0;
  // one partial per persistent workgroup (at most one workgroup per CU)
#ifdef KGCN_DEV_KNOBS
  return (int64_t)16 * kNumCU * ((int64_t)din * dout + dout) * 4;         // (room for the KGCN_BWD_GRID_MULT experiment)
#else
  return (int64_t)kNumCU * ((int64_t)din * dout + dout) * 4;
#endif
}

extern "C" int kgcn_graphconv_bwd_f32(const kgcn_csr_batch* at, const float* x, const float* w,
                                      const float* dout_grad, int32_t din, int32_t dout, float* dx,
                                      float* dw, float* dbias, void* workspace,
                                      int64_t workspace_bytes, void* stream) {
  if (int rc = check_padded(at, "kgcn_graphconv_bwd_f32")) return rc;
  if (!fused_shape_ok(at->rows, din, dout, at->max_nnz_per_graph))
    return fail("kgcn_graphconv_bwd_f32: shape N=%d din=%d dout=%d max_nnz=%d not supported by the "
                "fused kernel", at->rows, din, dout, at->max_nnz_per_graph);
  if (!dw || !dbias) return fail("kgcn_graphconv_bwd_f32: dw/dbias is NULL");
  hipStream_t s = as_stream(stream);
  if (at->num_graphs == 0) {
    (void)hipMemsetAsync(dw, 0, (size_t)din * dout * 4, s);
    (void)hipMemsetAsync(dbias, 0, (size_t)dout * 4, s);
    return 0;
  }
  if (!x || !w || !dout_grad) return fail("kgcn_graphconv_bwd_f32: NULL operand");
  const bool vec = !(din & 3) && !(dout & 3);
  if ((vec && (!aligned16(x) || !aligned16(dout_grad) || (dx && !aligned16(dx)))) || !aligned16(at->cv))
    return fail("kgcn_graphconv_bwd_f32: tensors not 16-byte aligned");
  const int mode = vec ? 2 : (((at->rows * din) & 3) == 0 && ((at->rows * dout) & 3) == 0 && aligned16(x) &&
                              aligned16(dout_grad) && (!dx || aligned16(dx))) ? 1 : 0;
  const size_t full_lds = bwd_planes_lds_bytes(at->max_nnz_per_graph);
  const bool full = dx != nullptr && is_full(at->rows, din, dout) && full_lds <= (size_t)kLdsBytes &&
                    at->max_nnz_per_graph < 4096;
  const int pack = (full || is_full(at->rows, din, dout)) ? 1 : pack_factor(at->rows, at->max_nnz_per_graph, FD * FD * 4);
  const size_t per = full ? full_lds / BWD_FULL_WPB : bwd_slice(at->max_nnz_per_graph * pack);
  const int wpb = full ? BWD_FULL_WPB : fused_wpb(per, FD * FD * 4);
  int blocks = fused_grid((at->num_graphs + pack - 1) / pack, wpb);
  // two waves per graph slot (graphconv_bwd_pairs_kernel): every pair must own at least two graphs (uniform barrier counts)
  const bool pairs = KGCN_BWD_PAIRS != 0 && full && at->num_graphs >= 2 * BP_PAIRS * blocks;
  // development (the non-persistent form, measured): KGCN_BWD_GRID_MULT = k launches k workgroups per CU slot,
  // each walking 1 / k of the graphs with its own pipeline fill, W' split and dW partial (k x 256 partials in the second stage) --
  // the form the kernel's LDS footprint (one 4-wave workgroup per CU) allows.  profiles/r05_headline_experiments.txt
  static const char* gm = dev_knob("KGCN_BWD_GRID_MULT");
  if (!pairs && gm && atoi(gm) > 1 && atoi(gm) <= 16 && (long)blocks * atoi(gm) * wpb * 2 <= at->num_graphs) blocks *= atoi(gm);
  const int64_t need = (int64_t)blocks * ((int64_t)din * dout + dout) * 4;
  if (!workspace || workspace_bytes < need)
    return fail("kgcn_graphconv_bwd_f32: workspace %lld < %lld bytes", (long long)workspace_bytes,
                (long long)need);
  float* part_dw = static_cast<float*>(workspace);
  float* part_db = part_dw + (long)blocks * din * dout;
  const size_t lds = full ? full_lds : FD * FD * 4 + (size_t)wpb * per;
  static thread_local bool attr_set = false;
  if (!attr_set) {
    allow_big_lds(graphconv_bwd_planes_kernel);
    allow_big_lds(graphconv_bwd_pairs_kernel);
    allow_big_lds(graphconv_bwd_kernel<2>);
    allow_big_lds(graphconv_bwd_kernel<1>);
    allow_big_lds(graphconv_bwd_kernel<0>);
    attr_set = true;
  }
  const int2* cv = reinterpret_cast<const int2*>(at->cv);
  if (pairs)
    hipLaunchKernelGGL(graphconv_bwd_pairs_kernel, dim3(blocks), dim3(512), lds, s, at->slots,
                       at->graph_ptr, cv, x, w, dout_grad, dx, part_dw, part_db, at->num_graphs,
                       at->max_nnz_per_graph);
  else if (full)
    hipLaunchKernelGGL(graphconv_bwd_planes_kernel, dim3(blocks), dim3(64 * wpb), lds, s, at->slots,
                       at->graph_ptr, cv, x, w, dout_grad, dx, part_dw, part_db, at->num_graphs,
                       at->max_nnz_per_graph);
  else if (mode == 2)
    hipLaunchKernelGGL(graphconv_bwd_kernel<2>, dim3(blocks), dim3(64 * wpb), lds, s, at->slots,
                       at->graph_ptr, cv, x, w, dout_grad, dx, part_dw, part_db, at->num_graphs,
                       at->rows, din, dout, at->max_nnz_per_graph, pack);
  else if (mode == 1)
    hipLaunchKernelGGL(graphconv_bwd_kernel<1>, dim3(blocks), dim3(64 * wpb), lds, s, at->slots,
                       at->graph_ptr, cv, x, w, dout_grad, dx, part_dw, part_db, at->num_graphs,
                       at->rows, din, dout, at->max_nnz_per_graph, pack);
  else
    hipLaunchKernelGGL(graphconv_bwd_kernel<0>, dim3(blocks), dim3(64 * wpb), lds, s, at->slots,
                       at->graph_ptr, cv, x, w, dout_grad, dx, part_dw, part_db, at->num_graphs,
                       at->rows, din, dout, at->max_nnz_per_graph, pack);
  if (int rc = check_launch("graphconv_bwd_kernel")) return rc;
  return launch_reduce_pair(part_dw, (long)din * dout, dw, part_db, dout, dbias, blocks, s);      // (queued inside a deferral scope)
}
