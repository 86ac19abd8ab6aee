// Cross-layer kernels, tile route: the layer list of stack.hip for 64-row tiles of WHOLE graphs on the f32 MFMA.
//
// stack.hip walks the layers for ONE graph per workgroup trip: right for a few hundred graphs (one graph per CU, the
// latency of one graph), but at the batch sizes BASELINE configs 1/4 name (4,096 graphs of 10 nodes) every CU runs 16
// latency-bound trips of ~3 rows per wave.  Here a workgroup takes G = floor(64 / N) consecutive graphs -- their rows AND
// their stored entries are contiguous in the batched CSR -- as one [64 x 64] activation tile:
//   * H W, H^T dT and dT W^T run as v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: exact f32 products, no bf16 split), one
//     32 x 32 block per wave, operands straight from LDS: A and B as ds_read_b64 along k out of [.][66] tiles (32 rows x one
//     column pair cover the 64 banks once), two MFMAs per read pair (the first contracts k, k+2, the second k+1, k+3);
//   * the aggregation runs from an ELL copy of the tile's adjacency built once per tile (step s of every row side by side,
//     short rows padded with value 0 at a row of zeros behind the tile: 0 * 0, never 0 * inf): thread (wave w, lane j) owns
//     column j of the rows w, w+4, ..., and step s of its sixteen rows is sixteen independent LDS reads -- with ONE wave per
//     SIMD (the tiles and all weights fill the LDS) nothing else hides an LDS round trip;
//   * activations are dispatched per layer OUTSIDE the element loops (a branch per element serialises the sixteen
//     independent chains of a thread: measured 250-450 cycles per element);
//   * backward: d pre-activation of the layer below is produced by the dX epilogue itself (dX (.) act'(input tile)), dW of
//     every layer stays in the wave's accumulator registers over all of the workgroup's tiles (16 VGPRs per layer), dbias /
//     dgamma / dbeta in one register per thread and layer; one partial per workgroup, reduce_partials adds them in a fixed
//     order: deterministic.
// Results differ from the one-graph route only by the summation order of the contractions.
#include "stack_common.h"

namespace kgcn {

__host__ __device__ inline int s2_wblock(int kind) { return kind == 2 ? 128 : 64 * S2_LD + 64; }

Stack2Plan stack2_plan(StackArgs& a) {
  Stack2Plan p{};
  int wo = 0, nmat = 0;
  for (int l = 0; l < a.nl; ++l) {
    a.woff2[l] = wo;
    wo += s2_wblock(a.kind[l]);
    a.mslot[l] = a.kind[l] == 2 ? -1 : nmat++;
    if (a.kind[l] != 2 && a.mslot[l] < S2_MAXM) a.mlayer[a.mslot[l]] = l;
  }
  for (int i = nmat; i < S2_MAXM; ++i) a.mlayer[i] = -1;
  a.wtotal2 = wo;
  a.G = S2_R / a.N;
  const long ent = (long)a.G * a.max_nnz;
  a.max_ent = (int)(ent < 1 ? 1 : ent);
  static const char* abl = dev_knob("KGCN_S2_ABL");
  a.abl = abl ? atoi(abl) : 0;
  const size_t graphs = (size_t)S2_AUX * 4 + ((size_t)a.max_ent + (size_t)64 * a.N) * 8;
  p.lds_fwd = ((size_t)wo + 3 * (size_t)S2_TILE) * 4 + graphs;
  p.lds_bwd = ((size_t)wo + 4 * (size_t)S2_TILE) * 4 + graphs;
  p.ok = nmat <= S2_MAXM && a.G >= 1 && ent <= 8192 && p.lds_fwd <= (size_t)kLdsBytes && p.lds_bwd <= (size_t)kLdsBytes;
  return p;
}

int stack2_blocks(long T, int G) {
  const long tiles = (T + G - 1) / G;
  return (int)(tiles < kNumCU ? (tiles < 1 ? 1 : tiles) : kNumCU);
}

template <int L> struct S2IC { static constexpr int value = L; };
template <int L, class F>
__device__ __forceinline__ void s2_layers_down(F&& f) {
  f(S2IC<L>{});
  if constexpr (L > 0) s2_layers_down<L - 1>(f);
}
// f(activation as a compile-time constant): the element loops behind it are branch-free
template <class F>
__device__ __forceinline__ void s2_with_act(int act, F&& f) {
  switch (act) {
    case KGCN_ACT_SIGMOID: f(S2IC<KGCN_ACT_SIGMOID>{}); break;
    case KGCN_ACT_RELU: f(S2IC<KGCN_ACT_RELU>{}); break;
    case KGCN_ACT_TANH: f(S2IC<KGCN_ACT_TANH>{}); break;
    default: f(S2IC<KGCN_ACT_NONE>{}); break;
  }
}
// act'(activation output), selects only
__device__ __forceinline__ float s2_dact(float y, int act) {
  const float sg = y * (1.0f - y), th = 1.0f - y * y, re = y > 0.f ? 1.0f : 0.f;
  return act == KGCN_ACT_SIGMOID ? sg : act == KGCN_ACT_RELU ? re : act == KGCN_ACT_TANH ? th : 1.0f;
}

struct S2Lds {
  float* wl;
  int* rp;        // [68]  row pointers of the tile, relative to its first entry
  int* rowbase;   // [64]  first tile row of the row's graph
  int* nval;      // [64]  valid rows of the row's graph
  int* gidx;      // [64]  index of the row's graph in the tile
  int* misc;      // [4]   longest row of the tile
  int2* ent;      // [max_ent]  (column within the graph, value bits)
  int2* ell;      // [N][64]    step s of tile row r: (byte offset of the operand row in a tile, value bits)
};

__device__ __forceinline__ S2Lds s2_carve(float* after_tiles, float* wl, int max_ent) {
  S2Lds s;
  s.wl = wl;
  s.rp = reinterpret_cast<int*>(after_tiles);
  s.rowbase = s.rp + 68;
  s.nval = s.rowbase + 64;
  s.gidx = s.nval + 64;
  s.misc = s.gidx + 64;
  s.ent = reinterpret_cast<int2*>(s.misc + 4);
  s.ell = s.ent + max_ent;
  return s;
}

// Every weight matrix [din x dout] (row-major) -> its [64][66] block, zero padded: transposed (W^T, forward) or as is (backward).
// The loads of ALL matrices are in flight together (sixteen per thread and matrix): one global round trip for the staging
// instead of one per matrix (no measurable difference in the release build: 0.0895 / 0.1985 ms per step either way).
__device__ __forceinline__ void s2_stage_matrices(const StackArgs& a, float* wl, bool transposed) {
  float v[S2_MAXM][16];
#pragma unroll
  for (int mi = 0; mi < S2_MAXM; ++mi) {
    const int l = a.mlayer[mi];
    if (l < 0) continue;
    const int din = a.din[l], dout = a.dout[l];
    const float* __restrict__ wsrc = a.w[l];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int k = (threadIdx.x >> 6) + 4 * q, c = threadIdx.x & 63;
      v[mi][q] = (k < din && c < dout) ? wsrc[(long)k * dout + c] : 0.f;
    }
  }
#pragma unroll
  for (int mi = 0; mi < S2_MAXM; ++mi) {
    const int l = a.mlayer[mi];
    if (l < 0) continue;
    float* blk = wl + a.woff2[l];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int k = (threadIdx.x >> 6) + 4 * q, c = threadIdx.x & 63;
      blk[transposed ? c * S2_LD + k : k * S2_LD + c] = v[mi][q];
    }
  }
}

// Rows / entries / valid counts of a tile's graphs travel global -> registers -> LDS in two steps so that the NEXT tile's
// loads are in flight while the current tile computes: s2_fetch_rows (row pointers, valid counts: independent loads), then --
// once those have landed -- s2_fetch_entries (the first 512 stored entries; the rest is read directly at commit time).
struct S2Graphs {
  int base, last, mine, nv, nrows;
  int2 ev[2];
  bool have_ev;
};

__device__ __forceinline__ void s2_fetch_rows(const StackArgs& a, S2Graphs& g, const int* __restrict__ rowptr,
                                              const int* __restrict__ enabled, long t0, int nrows) {
  const int* rp = rowptr + t0 * a.N;
  const int tid = threadIdx.x;
  g.nrows = nrows;
  g.have_ev = false;
  g.base = g.last = g.mine = g.nv = 0;
  if (nrows > 0) {
    g.base = rp[0];
    g.last = rp[nrows];
    if (tid <= nrows) g.mine = rp[tid];
    if (tid < nrows) g.nv = enabled ? enabled[t0 + tid / a.N] : a.N;
  }
}

__device__ __forceinline__ void s2_fetch_entries(const StackArgs& a, S2Graphs& g, const int2* __restrict__ cv) {
  int cnt = g.last - g.base;
  if (cnt > a.max_ent) cnt = a.max_ent;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = threadIdx.x + 256 * u;
    g.ev[u] = make_int2(0, 0);
    if (i < cnt) g.ev[u] = cv[g.base + i];
  }
  g.have_ev = true;
}

// registers -> LDS (first LDS pass; s2_build_ell behind a barrier is the second)
__device__ __forceinline__ void s2_commit_graphs(const StackArgs& a, const S2Lds& s, const S2Graphs& g, const int2* __restrict__ cv) {
  const int tid = threadIdx.x;
  if (tid <= g.nrows) s.rp[tid] = g.mine - g.base;
  if (tid < S2_R) s.nval[tid] = g.nv;
  int cnt = g.last - g.base;
  if (cnt > a.max_ent) cnt = a.max_ent;
  int from = tid;
  if (g.have_ev) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (tid + 256 * u < cnt) s.ent[tid + 256 * u] = g.ev[u];
    from = tid + 512;
  }
  for (int i = from; i < cnt; i += 256) s.ent[i] = cv[g.base + i];
}

// Called by all threads behind the barrier that follows s2_commit_graphs.  Returns the longest row of the tile -- every wave
// finds it from its own copy of the row lengths (lane = row, ballots: no LDS atomics, no barrier) -- and fills the ELL copy when
// it has at most N steps (always, unless a row stores duplicate columns).
__device__ __forceinline__ int s2_build_ell(const StackArgs& a, const S2Lds& s, int nrows) {
  const int tid = threadIdx.x, w = tid >> 6, r = tid & 63;
  const int b = r < nrows ? s.rp[r] : 0, e = r < nrows ? s.rp[r + 1] : 0;
  int L = 0;
  while (__ballot(e - b > L) != 0) ++L;
  if (L > a.N) return L;
  const int rb = s.rowbase[r];
  for (int st = w; st < L; st += 4) {
    const bool in = b + st < e;
    const int2 p = s.ent[in ? b + st : 0];
    int2 o;
    o.x = (in ? rb + p.x : S2_R) * (S2_LD * 4);       // padding: the zero row behind the tile, value 0
    o.y = in ? p.y : 0;
    s.ell[st * 64 + r] = o;
  }
  return L;
}

// y[q] = sum over the stored entries of tile row r = w + 4 (q0 + q) of val * T[first row of its graph + col][j], q < NR, entries
// in stored order.  ELL form: step s of the NR rows = NR broadcast entry reads at immediate offsets + NR operand reads.
template <int NR>
__device__ __forceinline__ void s2_aggregate(const float* T, const S2Lds& s, int L, int nrows, int N, int w, int j, int q0, float* y) {
#pragma unroll
  for (int q = 0; q < NR; ++q) y[q] = 0.f;
  if (L <= N) {
    const char* tj = reinterpret_cast<const char*>(T + j);
    const int2* er = s.ell + w + 4 * q0;
#pragma unroll 2
    for (int st = 0; st < L; ++st) {
      int2 p[NR];
#pragma unroll
      for (int q = 0; q < NR; ++q) p[q] = er[st * 64 + 4 * q];
#pragma unroll
      for (int q = 0; q < NR; ++q)
        y[q] = __builtin_fmaf(__int_as_float(p[q].y), *reinterpret_cast<const float*>(tj + p[q].x), y[q]);
    }
    return;
  }
  // rows with duplicate columns (more stored entries than nodes): the rows advance in lockstep over the CSR itself
  int beg[NR], len[NR];
  const float* tb[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    const int r = w + 4 * (q0 + q);
    const bool ok = r < nrows;
    const int b = s.rp[ok ? r : 0], e = s.rp[ok ? r + 1 : 0];
    beg[q] = b;
    len[q] = ok ? e - b : 0;
    tb[q] = T + s.rowbase[r] * S2_LD + j;
  }
  for (int st = 0; st < L; ++st) {
    int2 p[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) p[q] = s.ent[st < len[q] ? beg[q] + st : 0];
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      const bool in = st < len[q];
      const float t = tb[q][(in ? p[q].x : 0) * S2_LD];
      y[q] = in ? __builtin_fmaf(__int_as_float(p[q].y), t, y[q]) : y[q];
    }
  }
}

// acc (32 x 32 block at rows r0, columns c0) = sum_k A[r0 + i][k] B[c0 + j][k] over k < kdim rounded up to 8 (both tiles hold
// zeros there); A, B: [.][66] tiles.  Straight loop over 8-wide k chunks: four ds_read_b64, four MFMAs on two accumulator
// chains (93 cycles per MFMA instead of 64, tools/probes/mfma_f32_probe.hip; ping-pong operand registers pinned behind the
// MFMAs reach 64 in the probe but spill in the backward kernel and bought 4 % in the forward: not kept).
__device__ __forceinline__ f32x16 s2_mm_kk(const float* A, const float* B, int r0, int c0, int kdim) {
  const int li = threadIdx.x & 31, lh = (threadIdx.x >> 5) & 1;
  const float* ap = A + (r0 + li) * S2_LD + 2 * lh;
  const float* bp = B + (c0 + li) * S2_LD + 2 * lh;
  const int nch = (kdim + 7) >> 3;
  f32x16 acc = {0}, acc2 = {0};
  for (int c = 0; c < nch; ++c) {
    const f32x2 a0 = *reinterpret_cast<const f32x2*>(ap + 8 * c), a1 = *reinterpret_cast<const f32x2*>(ap + 8 * c + 4);
    const f32x2 b0 = *reinterpret_cast<const f32x2*>(bp + 8 * c), b1 = *reinterpret_cast<const f32x2*>(bp + 8 * c + 4);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[0], b0[0], acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[1], b0[1], acc2, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0], b1[0], acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[1], b1[1], acc2, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] += acc2[i];
  return acc;
}

__device__ __forceinline__ int s2_acc_row(int i) { return (i & 3) + 8 * (i >> 2) + 4 * ((threadIdx.x >> 5) & 1); }

// bit q: tile row w + 4 q is a valid node row of its graph (normalisation layers); lane = row, one ballot
__device__ __forceinline__ unsigned s2_valid_mask(const S2Lds& s, int nrows, int w) {
  const int r = threadIdx.x & 63;
  const unsigned long long vb = __ballot(r < nrows && (r - s.rowbase[r]) < s.nval[r]);
  unsigned m = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) m |= (unsigned)((vb >> (w + 4 * q)) & 1ull) << q;
  return m;
}

#ifdef KGCN_DEV_KNOBS
#define S2_PROF_DECL unsigned long long pt_[12] = {0}, pl_ = __builtin_readcyclecounter()
#define S2_STAMP(i) do { if (a.abl & 256) { const unsigned long long n_ = __builtin_readcyclecounter(); pt_[i] += n_ - pl_; pl_ = n_; } } while (0)
#define S2_PROF_DUMP(name) do { if ((a.abl & 256) && blockIdx.x == 0 && threadIdx.x == 0) { printf(name ":"); for (int i_ = 0; i_ < 12; ++i_) printf(" %llu", pt_[i_]); printf("\n"); } } while (0)
#else
#define S2_PROF_DECL
#define S2_STAMP(i)
#define S2_PROF_DUMP(name)
#endif

__global__ __launch_bounds__(256) void stack2_fwd_kernel(StackArgs a, const int* __restrict__ rowptr, const int2* __restrict__ cv,
                                                         const float* __restrict__ x, const int* __restrict__ enabled, long T,
                                                         float* __restrict__ pooled) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ha = sm + a.wtotal2;
  float* Hb = Ha + S2_TILE;
  float* Tt = Hb + S2_TILE;
  const S2Lds s = s2_carve(Tt + S2_TILE, sm, a.max_ent);
  const int N = a.N, G = a.G, tid = threadIdx.x, w = tid >> 6, j = tid & 63;
  const int wr = w >> 1, wc = w & 1, li = tid & 31;
  S2_PROF_DECL;
  // weights: W^T [j][k] (the B operand is read along k), bias behind it
  for (int l = 0; l < a.nl; ++l) {
    float* blk = s.wl + a.woff2[l];
    if (a.kind[l] == 2) {
      if (tid < 64) {
        float sc = 0.f, sh = 0.f;
        if (tid < a.dout[l]) {
          const float rs = 1.0f / __builtin_sqrtf(a.var[l][tid] + a.eps[l]);
          sc = a.w[l][tid] * rs;
          sh = a.b[l][tid] - a.mean[l][tid] * sc;
        }
        blk[tid] = sc;
        blk[64 + tid] = sh;
      }
    } else {
      if (tid < 64) blk[64 * S2_LD + tid] = (tid < a.dout[l] && a.b[l]) ? a.b[l][tid] : 0.f;
    }
  }
  s2_stage_matrices(a, s.wl, true);
  if (tid < S2_R) { s.gidx[tid] = tid / N; s.rowbase[tid] = (tid / N) * N; }
  if (tid < S2_LD) Tt[S2_R * S2_LD + tid] = 0.f;      // the row of zeros the ELL padding points at
  const long ntiles = (T + G - 1) / G;
  const int d0 = a.din[0];
  auto rows_of = [&](long tile) { const long left = T - tile * G; return tile < ntiles ? (int)(left < G ? left : G) * N : 0; };
  // input rows and graph structure of a tile: global -> registers
  float xv[16];
  S2Graphs gr;
  auto fetch_tile = [&](long tile) __attribute__((always_inline)) {
    const int nr = rows_of(tile);
    const float* xs = x + tile * G * N * d0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int r = w + 4 * q;
      xv[q] = (r < nr && j < d0) ? xs[(long)r * d0 + j] : 0.f;
    }
    s2_fetch_rows(a, gr, rowptr, enabled, tile * G, nr);
  };
  fetch_tile(blockIdx.x);
  S2_STAMP(0);
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long t0 = tile * G;
    const int gt = (int)((T - t0) < G ? (T - t0) : G), nrows = gt * N;
    __syncthreads();                                   // the previous tile's readers are done (and the weights are staged)
#pragma unroll
    for (int q = 0; q < 16; ++q) Ha[(w + 4 * q) * S2_LD + j] = xv[q];
    s2_commit_graphs(a, s, gr, cv);
    fetch_tile(tile + gridDim.x);                      // the next tile's rows travel while this one computes
    __syncthreads();
    const int L = s2_build_ell(a, s, nrows);
    const unsigned vmask = s2_valid_mask(s, nrows, w);
    S2_STAMP(1);
    float* Hin = Ha;
    float* Hout = Hb;
    for (int l = 0; l < a.nl; ++l) {
      const float* W = s.wl + a.woff2[l];
      const int dout = a.dout[l], act = a.act[l], kind = a.kind[l];
      float* og = a.out[l] + t0 * N * dout;
      if (kind == 2) {
        float h[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) h[q] = Hin[(w + 4 * q) * S2_LD + j];
        const float sc = W[j], sh = W[64 + j];
        s2_with_act(act, [&](auto ac) __attribute__((always_inline)) {
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float y = ((vmask >> q) & 1) ? __builtin_fmaf(h[q], sc, sh) : 0.f;
            h[q] = (j < dout && w + 4 * q < nrows) ? act_fwd(y, decltype(ac)::value) : 0.f;
          }
        });
#pragma unroll
        for (int q = 0; q < 16; ++q) Hout[(w + 4 * q) * S2_LD + j] = h[q];
        if (j < dout) {
#pragma unroll
          for (int q = 0; q < 16; ++q)
            if (w + 4 * q < nrows) og[(long)(w + 4 * q) * dout + j] = h[q];
        }
        S2_STAMP(2);
      } else {
        f32x16 acc = {0};
        if (wr * 32 < nrows && wc * 32 < dout) acc = s2_mm_kk(Hin, W, wr * 32, wc * 32, a.din[l]);
        const int col = wc * 32 + li;
        const float bias = W[64 * S2_LD + col];
        S2_STAMP(3);
        if (kind == 1) {
          s2_with_act(act, [&](auto ac) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int row = wr * 32 + s2_acc_row(i);
              acc[i] = (col < dout && row < nrows) ? act_fwd(acc[i] + bias, decltype(ac)::value) : 0.f;
            }
          });
#pragma unroll
          for (int i = 0; i < 16; ++i) Hout[(wr * 32 + s2_acc_row(i)) * S2_LD + col] = acc[i];
          if (col < dout) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int row = wr * 32 + s2_acc_row(i);
              if (row < nrows) og[(long)row * dout + col] = acc[i];
            }
          }
          S2_STAMP(4);
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) Tt[(wr * 32 + s2_acc_row(i)) * S2_LD + col] = acc[i] + bias;
          __syncthreads();
          S2_STAMP(5);
          float y[16];
          s2_aggregate<16>(Tt, s, L, nrows, N, w, j, 0, y);
          S2_STAMP(6);
          s2_with_act(act, [&](auto ac) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 16; ++q) y[q] = (j < dout && w + 4 * q < nrows) ? act_fwd(y[q], decltype(ac)::value) : 0.f;
          });
#pragma unroll
          for (int q = 0; q < 16; ++q) Hout[(w + 4 * q) * S2_LD + j] = y[q];
          if (j < dout) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
              if (w + 4 * q < nrows) og[(long)(w + 4 * q) * dout + j] = y[q];
          }
          S2_STAMP(7);
        }
      }
      __syncthreads();                                 // Hout complete; Tt / Hin free
      S2_STAMP(8);
      if (l == 0) s2_fetch_entries(a, gr, cv);         // the next tile's row pointers have landed by now
      float* tmp = Hin; Hin = Hout; Hout = tmp;
    }
    if (a.gather) {
      const int dl = a.dout[a.nl - 1];
      for (int g = w; g < gt; g += 4) {
        float sum = 0.f;
        for (int r = 0; r < N; ++r) sum += Hin[(g * N + r) * S2_LD + j];
        if (j < dl) pooled[(t0 + g) * dl + j] = sum;
      }
    }
    S2_STAMP(9);
  }
  S2_PROF_DUMP("s2fwd stage load+ell bn mm dense_epi ttwrite+bar agg actstore endbar gather");
}

__global__ __launch_bounds__(256) void stack2_bwd_kernel(StackArgs a, const int* __restrict__ rowptr_t, const int2* __restrict__ cv_t,
                                                         const float* __restrict__ x, const int* __restrict__ enabled, long T,
                                                         const float* __restrict__ dlast, float* __restrict__ dx,
                                                         float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* P = sm + a.wtotal2;                           // d pre-activation of the layer being processed
  float* Pn = P + S2_TILE;                             // ... of the layer below, written by this layer's dX epilogue
  float* Ht = Pn + S2_TILE;                            // saved layer input
  float* Tt = Ht + S2_TILE;                            // A^T d pre-activation
  const S2Lds s = s2_carve(Tt + S2_TILE, sm, a.max_ent);
  const int N = a.N, G = a.G, tid = threadIdx.x, w = tid >> 6, j = tid & 63;
  const int wr = w >> 1, wc = w & 1, li = tid & 31, lh = (tid >> 5) & 1;
  S2_PROF_DECL;
  // weights row-major W [k][j] (dX reads the B operand along j); kind 2: scale
  for (int l = 0; l < a.nl; ++l) {
    float* blk = s.wl + a.woff2[l];
    if (a.kind[l] == 2) {
      if (tid < 64) blk[tid] = tid < a.dout[l] ? a.w[l][tid] * (1.0f / __builtin_sqrtf(a.var[l][tid] + a.eps[l])) : 0.f;
    }
  }
  s2_stage_matrices(a, s.wl, false);
  if (tid < S2_R) { s.gidx[tid] = tid / N; s.rowbase[tid] = (tid / N) * N; }
  if (tid < S2_LD) { P[S2_R * S2_LD + tid] = 0.f; Pn[S2_R * S2_LD + tid] = 0.f; }   // the aggregation reads P or Pn
  f32x16 dW[S2_MAXM];
  float db[S2_MAXL], s1[S2_MAXL];                      // dbias (kind 2: S0 = sum dpre) | kind 2: S1 = sum dpre * x
#pragma unroll
  for (int m = 0; m < S2_MAXM; ++m) {
#pragma unroll
    for (int i = 0; i < 16; ++i) dW[m][i] = 0.f;
  }
#pragma unroll
  for (int l = 0; l < S2_MAXL; ++l) {
    db[l] = 0.f;
    s1[l] = 0.f;
  }
  const long ntiles = (T + G - 1) / G;
  const int dl = a.dout[a.nl - 1], act_top = a.act[a.nl - 1];
  auto rows_of = [&](long tile) { const long left = T - tile * G; return tile < ntiles ? (int)(left < G ? left : G) * N : 0; };
  S2Graphs gr;
  s2_fetch_rows(a, gr, rowptr_t, enabled, (long)blockIdx.x * G, rows_of(blockIdx.x));
  float hp[16];
  const int gmul = 65536 / N + 1;                      // (r * gmul) >> 16 == r / N for r < 64
  auto fetch_input = [&](long tile, int l) __attribute__((always_inline)) {
    const int nr = rows_of(tile), din = a.din[l];
    const float* hsrc = (l == 0 ? x : a.out[l > 0 ? l - 1 : 0]) + tile * G * N * din;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int r = w + 4 * q;
      hp[q] = (r < nr && j < din) ? hsrc[(long)r * din + j] : 0.f;
    }
  };
  S2_STAMP(0);
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long t0 = tile * G;
    const int gt = (int)((T - t0) < G ? (T - t0) : G), nrows = gt * N;
    __syncthreads();
    {
      // d pre-activation of the top layer: d(output) (.) act'(saved output)
      const float* ys = a.out[a.nl - 1] + t0 * N * dl;
      float yv[16], dv[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int r = w + 4 * q;
        const bool ok = r < nrows && j < dl;
        yv[q] = ok ? ys[(long)r * dl + j] : 0.f;
        dv[q] = ok ? (a.gather ? dlast[(t0 + ((r * gmul) >> 16)) * dl + j] : dlast[(t0 * N + r) * dl + j]) : 0.f;
      }
      s2_commit_graphs(a, s, gr, cv_t);
      s2_fetch_rows(a, gr, rowptr_t, enabled, (tile + gridDim.x) * G, rows_of(tile + gridDim.x));
#pragma unroll
      for (int q = 0; q < 16; ++q) P[(w + 4 * q) * S2_LD + j] = dv[q] * s2_dact(yv[q], act_top);
    }
    S2_STAMP(10);
    __syncthreads();
    const int L = s2_build_ell(a, s, nrows);
    const unsigned vmask = s2_valid_mask(s, nrows, w);
    S2_STAMP(1);
    // The layer loop is a RUNTIME loop (one copy of the layer code: unrolled per layer it was 170 KB of instructions, more
    // than the instruction cache, with one wave per SIMD to hide the misses); only the statements that touch the per-layer
    // register accumulators are dispatched on the layer index.
    for (int l = a.nl - 1; l >= 0; --l) {
      const int din = a.din[l], dout = a.dout[l], kind = a.kind[l];
      const int act_below = l > 0 ? a.act[l > 0 ? l - 1 : 0] : KGCN_ACT_NONE;
      const float* W = s.wl + a.woff2[l];
      // this layer's input tile (= the saved output of the layer below): requested now, stored behind the barrier (requesting it
      // one layer ahead keeps 16 more registers live across the MFMA phases: the kernel then spills and gains nothing)
      fetch_input(tile, l);
      __syncthreads();                                 // P (dX epilogue of the layer above) complete; its operand tiles free
      S2_STAMP(2);
      if (kind == 2) {
        // column-owner: dgamma / dbeta sums and dx = dpre * gamma * rstd (0 on padded rows), in place
        const float sc = W[j];
        float a1 = 0.f, a0 = 0.f;
        float d[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) d[q] = P[(w + 4 * q) * S2_LD + j];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          d[q] = (((vmask >> q) & 1) && j < dout) ? d[q] : 0.f;
          a1 = __builtin_fmaf(d[q], hp[q], a1);
          a0 += d[q];
          d[q] = d[q] * sc * s2_dact(hp[q], act_below);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) P[(w + 4 * q) * S2_LD + j] = d[q];
        s2_layers_down<S2_MAXL - 1>([&](auto lc) __attribute__((always_inline)) {
          constexpr int ll = decltype(lc)::value;
          if (ll == l) { s1[ll] += a1; db[ll] += a0; }
        });
        S2_STAMP(3);
        continue;
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) Ht[(w + 4 * q) * S2_LD + j] = hp[q];
      float bs = 0.f;
      const float* Tp = P;
      if (kind == 0) {
        float y[16];
        s2_aggregate<16>(P, s, L, nrows, N, w, j, 0, y);
        S2_STAMP(4);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          Tt[(w + 4 * q) * S2_LD + j] = y[q];
          bs += y[q];
        }
        Tp = Tt;
      } else {
        float d[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) d[q] = P[(w + 4 * q) * S2_LD + j];
#pragma unroll
        for (int q = 0; q < 16; ++q) bs += d[q];
      }
      S2_STAMP(4);
      __syncthreads();                                 // input tile and A^T dpre complete
      S2_STAMP(5);
      // dW[k, c] += sum_r Hin[r, k] dT[r, c]: wave block k0 = 32 wr, c0 = 32 wc; both operands read along the row index
      const bool mine = wr * 32 < din && wc * 32 < dout;
      const float* ap = Ht + lh * S2_LD + wr * 32 + li;
      const float* bp = Tp + lh * S2_LD + wc * 32 + li;
      const int rend = mine ? (nrows + 1) & ~1 : 0;
      s2_layers_down<S2_MAXL - 1>([&](auto lc) __attribute__((always_inline)) {
        constexpr int ll = decltype(lc)::value;
        if (ll == l) db[ll] += bs;
      });
      const int slot = a.mslot[l];
      s2_layers_down<S2_MAXM - 1>([&](auto mc) __attribute__((always_inline)) {
        constexpr int ll = decltype(mc)::value;
        if (ll != slot) return;
        f32x16 acc = dW[ll], acc2 = {0};
        for (int r = 0; r < rend; r += 16) {           // rows past the tile's graphs are zero in both operands
          float av[8], bv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            av[u] = ap[(r + 2 * u) * S2_LD];
            bv[u] = bp[(r + 2 * u) * S2_LD];
          }
#pragma unroll
          for (int u = 0; u < 8; u += 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u + 1], bv[u + 1], acc2, 0, 0, 0);
          }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += acc2[i];
        dW[ll] = acc;
      });
      S2_STAMP(6);
      // d input[r, k] = sum_c dT[r, c] W[k, c], times act'(input) = d pre-activation of the layer below
      if (l > 0 || dx) {
        f32x16 di = {0};
        if (wr * 32 < nrows && wc * 32 < din) di = s2_mm_kk(Tp, W, wr * 32, wc * 32, dout);
        const int col = wc * 32 + li;
        if (l > 0) {
          float hv[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) hv[i] = Ht[(wr * 32 + s2_acc_row(i)) * S2_LD + col];
#pragma unroll
          for (int i = 0; i < 16; ++i) di[i] *= s2_dact(hv[i], act_below);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) Pn[(wr * 32 + s2_acc_row(i)) * S2_LD + col] = di[i];
        { float* tmp = P; P = Pn; Pn = tmp; }
      }
      if (l == a.nl - 1) s2_fetch_entries(a, gr, cv_t);   // the next tile's row pointers have landed by now
      S2_STAMP(7);
    }
    if (dx) {
      __syncthreads();
      const int d0 = a.din[0];
      float* o = dx + t0 * N * d0;
      if (j < d0) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (w + 4 * q < nrows) o[(long)(w + 4 * q) * d0 + j] = P[(w + 4 * q) * S2_LD + j];
      }
    }
    S2_STAMP(8);
  }
  // this workgroup's partial in the flat parameter-gradient layout: per layer dW [din, dout] | db [dout]  (kind 2: dgamma | dbeta)
  float* pp = part + (long)blockIdx.x * a.ptotal;
  float* red = Ht;                                     // [2][4][64]
  s2_layers_down<S2_MAXL - 1>([&](auto lc) __attribute__((always_inline)) {
    constexpr int l = decltype(lc)::value;
    if (l >= a.nl) return;
    const int din = a.din[l], dout = a.dout[l];
    float* o = pp + a.poff[l];
    __syncthreads();
    red[w * 64 + j] = db[l];
    red[256 + w * 64 + j] = s1[l];
    __syncthreads();
    if (a.kind[l] == 2) {
      if (w == 0 && j < dout) {
        const float v1 = (red[256 + j] + red[320 + j]) + (red[384 + j] + red[448 + j]);
        const float v0 = (red[j] + red[64 + j]) + (red[128 + j] + red[192 + j]);
        const float rs = 1.0f / __builtin_sqrtf(a.var[l][j] + a.eps[l]);
        o[j] = rs * (v1 - a.mean[l][j] * v0);          // dgamma = rstd (S1 - mean S0)
        o[dout + j] = v0;                              // dbeta
      }
    } else {
      const int c = wc * 32 + li;
      s2_layers_down<S2_MAXM - 1>([&](auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value;
        if (m != a.mslot[l]) return;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int k = wr * 32 + s2_acc_row(i);
          if (k < din && c < dout) o[(long)k * dout + c] = dW[m][i];
        }
      });
      if (w == 0 && j < dout) o[(long)din * dout + j] = (red[j] + red[64 + j]) + (red[128 + j] + red[192 + j]);
    }
  });
  S2_PROF_DUMP("s2bwd stage ell+vmask topbar+Htstore bn agg bar dW dX tail fetchissue loadP ttwrite");
}

static void s2_attr(const void* fn) {
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
}

int launch_stack2_fwd(const StackArgs& a, const Stack2Plan& p, const int* rowptr, const int2* cv, const float* x,
                      const int* enabled, long T, float* pooled, hipStream_t s) {
  static thread_local bool attr = false;
  if (!attr) { s2_attr(reinterpret_cast<const void*>(stack2_fwd_kernel)); attr = true; }
  hipLaunchKernelGGL(stack2_fwd_kernel, dim3(stack2_blocks(T, a.G)), dim3(256), p.lds_fwd, s, a, rowptr, cv, x, enabled, T, pooled);
  return check_launch("stack2_fwd_kernel");
}

int launch_stack2_bwd(const StackArgs& a, const Stack2Plan& p, const int* rowptr_t, const int2* cv_t, const float* x,
                      const int* enabled, long T, const float* dlast, float* dx, float* part, int blocks, hipStream_t s) {
  static thread_local bool attr = false;
  if (!attr) { s2_attr(reinterpret_cast<const void*>(stack2_bwd_kernel)); attr = true; }
  hipLaunchKernelGGL(stack2_bwd_kernel, dim3(blocks), dim3(256), p.lds_bwd, s, a, rowptr_t, cv_t, x, enabled, T, dlast, dx, part);
  return check_launch("stack2_bwd_kernel");
}

}  // namespace kgcn
