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
//     already in flight (register prefetch) while the current graph is being computed,
//   * the dense contraction runs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32)
//     with the weight fragments resident in registers (forward) / in LDS (backward, W^T),
//   * FW (resp. dFW) lives only in LDS, where the sparse aggregation gathers neighbour rows with
//     conflict-free ds_read_b128, four entries per lane group in flight -- X.W and dFW never
//     touch HBM,
//   * dW / dbias accumulate in MFMA accumulators across ALL graphs a wave processes, are reduced
//     across the workgroup's waves through LDS and leave the chip once per workgroup
//     (deterministic second-stage reduction).
// Algorithmic HBM bytes per graph (N=32, D=64, nnz=100): forward 17,316, backward 25,508.
//
// Shape support (kgcn_graphconv_fused_supported): N <= 32, din,dout <= 64 and multiples of 4.
// Everything else goes through kgcn_dense_* + kgcn_bconv_f32.
#include "kgcn_common.h"

namespace kgcn {

int launch_reduce_partials(const float* part, int nparts, long n, float* out, hipStream_t s);

constexpr int FN = 32;    // node tile (MFMA M)
constexpr int FD = 64;    // feature tile (K of the forward GEMM, two 32-wide output tiles)
constexpr int ALD = 68;   // padded row stride (floats) of tiles read as MFMA A fragments (b128)
constexpr int MAX_WPB = 8;
constexpr int ECV_PAD = 4;  // the 4-entry batched gather may read (not use) 3 entries past the end

__device__ __forceinline__ void wave_sync() {
  // LDS operations of one wave execute in order; this only stops the compiler from moving LDS
  // accesses of different lanes across the hand-off point.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ f32x4 ldv4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

struct WaveSlice {
  float* a;     // [FN][ALD]  A-fragment source tile
  float* b;     // [FN+1][FD] gather source tile; row FN stays zero (target of masked gathers)
  int2* ecv;    // [max_nnz + ECV_PAD]
  int* rp;      // [FN + 4]
};

__host__ __device__ inline size_t ecv_bytes(int max_nnz) {
  return ((size_t)(max_nnz + ECV_PAD) * 8 + 15) & ~(size_t)15;
}

__host__ __device__ inline size_t slice_bytes(int max_nnz) {
  return (size_t)FN * ALD * 4 + (size_t)(FN + 1) * FD * 4 + ecv_bytes(max_nnz) + (FN + 4) * 4;
}

__device__ __forceinline__ WaveSlice carve(unsigned char* base, int wave, int max_nnz) {
  unsigned char* p = base + (size_t)wave * slice_bytes(max_nnz);
  WaveSlice s;
  s.a = reinterpret_cast<float*>(p);
  s.b = s.a + FN * ALD;
  s.ecv = reinterpret_cast<int2*>(s.b + (FN + 1) * FD);
  s.rp = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(s.ecv) + ecv_bytes(max_nnz));
  return s;
}

// One graph's inputs in flight: the feature tile (<= 512 float4 = 8 per lane), the first 128 CSR
// entries (2 per lane) and the row pointers (lane l holds rowptr[t*N + min(l, N)]).
struct InFlight {
  f32x4 tile[8];
  int2 cv[2];
};

__device__ __forceinline__ void issue_tile(InFlight& f, const float* __restrict__ src, int n4,
                                           int lane) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int i = lane + q * 64;
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f.tile[q] = (i < n4) ? ldv4(src + (long)i * 4) : z;
  }
}

__device__ __forceinline__ void issue_cv(InFlight& f, const int2* __restrict__ cv, int base,
                                         int cnt, int lane) {
  const int2 z = {0, 0};
  f.cv[0] = (lane < cnt) ? cv[base + lane] : z;
  f.cv[1] = (lane + 64 < cnt) ? cv[base + lane + 64] : z;
}

// registers -> LDS: tile rows get row stride `ld` (floats); CSR slice rebased to 0
__device__ __forceinline__ void land(const InFlight& f, float* tile, int ld, int n4, int d4,
                                     const WaveSlice& ws, const int2* __restrict__ cv, int rp_val,
                                     int base, int cnt, int N, int lane) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int i = lane + q * 64;
    if (i < n4) {
      const int r = i / d4, c4 = i - r * d4;
      *reinterpret_cast<f32x4*>(tile + r * ld + c4 * 4) = f.tile[q];
    }
  }
  if (lane < cnt) ws.ecv[lane] = f.cv[0];
  if (lane + 64 < cnt) ws.ecv[lane + 64] = f.cv[1];
  for (int i = 128 + lane; i < cnt; i += 64) ws.ecv[i] = cv[base + i];  // rare: > 128 entries
  if (lane <= N) ws.rp[lane] = rp_val - base;
}

// Rows r0..r0+3 of the output (16 lanes x float4 each) = sum over the row's entries of
// val * src[col].  Entries are consumed four at a time: 4 ds_read_b64 (col,val) + 4 ds_read_b128
// in flight per lane instead of a dependent chain per entry.
template <typename Sink>
__device__ __forceinline__ void aggregate_rows(const WaveSlice& ws, const float* src, int src_ld,
                                               int N, int dcols, int lane, Sink&& sink) {
  const int sub = lane >> 4, cl = lane & 15;
  const bool col_ok = cl * 4 < dcols;
  const float* srcl = src + cl * 4;
  for (int r0 = 0; r0 < N; r0 += 4) {
    const int r = r0 + sub;
    const bool ok = (r < N) && col_ok;
    const int rr = ok ? r : 0;
    const int s = ws.rp[rr];
    const int e = ok ? ws.rp[rr + 1] : s;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    for (int k = s; k < e; k += 4) {
      const int2 p0 = ws.ecv[k], p1 = ws.ecv[k + 1], p2 = ws.ecv[k + 2], p3 = ws.ecv[k + 3];
      const bool h1 = k + 1 < e, h2 = k + 2 < e, h3 = k + 3 < e;
      const f32x4 x0 = ldv4(srcl + p0.x * src_ld);
      // masked slots gather the all-zero row FN with value 0 (0 * 0, never 0 * inf)
      const f32x4 x1 = ldv4(srcl + (h1 ? p1.x : FN) * src_ld);
      const f32x4 x2 = ldv4(srcl + (h2 ? p2.x : FN) * src_ld);
      const f32x4 x3 = ldv4(srcl + (h3 ? p3.x : FN) * src_ld);
      acc0 += __int_as_float(p0.y) * x0;
      acc1 += (h1 ? __int_as_float(p1.y) : 0.f) * x1;
      acc0 += (h2 ? __int_as_float(p2.y) : 0.f) * x2;
      acc1 += (h3 ? __int_as_float(p3.y) : 0.f) * x3;
    }
    if (ok) sink(r, cl, acc0 + acc1);
  }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void graphconv_fwd_kernel(
    const int* __restrict__ rowptr, const int2* __restrict__ cv, const float* __restrict__ x,
    const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ out, int T,
    int N, int din, int dout, int max_nnz) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, hi = lane >> 5;
  const int wpb = blockDim.x >> 6;
  WaveSlice ws = carve(smem, wave, max_nnz);

  const int nwaves = gridDim.x * wpb;
  int t = __builtin_amdgcn_readfirstlane(blockIdx.x * wpb + wave);
  if (t >= T) return;  // no workgroup barrier below: idle waves may leave

  // zero the A tile once: padding rows (>= N) and columns (>= din) stay zero for every graph
  for (int i = lane; i < FN * ALD; i += 64) ws.a[i] = 0.f;
  for (int i = lane; i < FD; i += 64) ws.b[FN * FD + i] = 0.f;

  // weight fragments: B[k][j] with k = hi*32 + s (the K permutation matches the A fragments)
  float wr0[32], wr1[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    const int k = hi * 32 + s;
    wr0[s] = (k < din && li < dout) ? w[(long)k * dout + li] : 0.f;
    wr1[s] = (k < din && 32 + li < dout) ? w[(long)k * dout + 32 + li] : 0.f;
  }
  const float b0 = (bias && li < dout) ? bias[li] : 0.f;
  const float b1 = (bias && 32 + li < dout) ? bias[32 + li] : 0.f;

  const int din4 = din >> 2;
  const int n4 = N * din4;
  const int lrp = lane < N ? lane : N;

  // ---- prologue: first graph's inputs, second graph's row pointers ---------------------------
  InFlight fl;
  int rp_cur = rowptr[(long)t * N + lrp];
  int base = __builtin_amdgcn_readlane(rp_cur, 0);
  int cnt = __builtin_amdgcn_readlane(rp_cur, N) - base;
  issue_tile(fl, x + (long)t * N * din, n4, lane);
  issue_cv(fl, cv, base, cnt, lane);
  int tn = t + nwaves;
  int rp_nxt = (tn < T) ? rowptr[(long)tn * N + lrp] : 0;
  wave_sync();

  for (;;) {
    // ---- 1. graph t: registers -> LDS ----------------------------------------------------------
    land(fl, ws.a, ALD, n4, din4, ws, cv, rp_cur, base, cnt, N, lane);
    wave_sync();

    // ---- 2. put graph t+nwaves in flight (lands while graph t is computed) ------------------
    const bool has_next = tn < T;
    int base_n = 0, cnt_n = 0;
    if (has_next) {
      base_n = __builtin_amdgcn_readlane(rp_nxt, 0);
      cnt_n = __builtin_amdgcn_readlane(rp_nxt, N) - base_n;
      issue_tile(fl, x + (long)tn * N * din, n4, lane);
      issue_cv(fl, cv, base_n, cnt_n, lane);
      rp_cur = rp_nxt;
      const int tnn = tn + nwaves;
      rp_nxt = (tnn < T) ? rowptr[(long)tnn * N + lrp] : 0;
    }

    // ---- 3. FW = x @ W + bias on the matrix cores ---------------------------------------------
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
    // C layout -> LDS gather tile (bank = column: conflict free)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
      ws.b[row * FD + li] = acc0[r];
      ws.b[row * FD + 32 + li] = acc1[r];
    }
    wave_sync();

    // ---- 4. out[t] = A[t] @ FW : 4 rows per pass, 1 KiB coalesced store per pass -------------
    float* ot = out + (long)t * N * dout;
    aggregate_rows(ws, ws.b, FD, N, dout, lane, [&](int r, int cl, f32x4 acc) {
      *reinterpret_cast<f32x4*>(ot + (long)r * dout + cl * 4) = acc;
    });
    wave_sync();

    if (!has_next) break;
    t = tn;
    tn += nwaves;
    base = base_n;
    cnt = cnt_n;
  }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// dFW tile row stride in the backward: ODD, so that both MFMA operand patterns read it with
// conflict-free ds_read_b32 (lanes along a column for dX's A operand, lanes along a row for dW's B
// operand) and no A-fragment registers are needed.
constexpr int BLD = 65;

__global__ __launch_bounds__(512, 2) void graphconv_bwd_kernel(
    const int* __restrict__ rowptr_t, const int2* __restrict__ cv_t, const float* __restrict__ x,
    const float* __restrict__ w, const float* __restrict__ g, float* __restrict__ dx,
    float* __restrict__ part_dw, float* __restrict__ part_db, int T, int N, int din, int dout,
    int max_nnz) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, hi = lane >> 5;
  const int wpb = blockDim.x >> 6;
  float* Wt = reinterpret_cast<float*>(smem);  // [FD][FD]: Wt[k][j] = W[j][k] (zero padded)
  WaveSlice ws = carve(smem + FD * FD * 4, wave, max_nnz);

  for (int i = tid; i < FD * FD; i += blockDim.x) {
    const int k = i >> 6, j = i & 63;
    Wt[i] = (j < din && k < dout) ? w[(long)j * dout + k] : 0.f;
  }
  for (int i = lane; i < FN * ALD; i += 64) ws.a[i] = 0.f;
  for (int i = lane; i < (FN + 1) * FD; i += 64) ws.b[i] = 0.f;
  __syncthreads();

  f32x16 dw00, dw01, dw10, dw11;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dw00[r] = 0.f; dw01[r] = 0.f; dw10[r] = 0.f; dw11[r] = 0.f; }
  f32x4 dbacc = {0.f, 0.f, 0.f, 0.f};

  const int din4 = din >> 2, dout4 = dout >> 2;
  const int nx4 = N * din4, ng4 = N * dout4;
  const int nwaves = gridDim.x * wpb;
  const int lrp = lane < N ? lane : N;
  int t = __builtin_amdgcn_readfirstlane(blockIdx.x * wpb + wave);

  if (t < T) {
    // ---- prologue: g, CSR(A^T) and x of the first graph in flight ------------------------------
    InFlight fg;       // g tile + CSR entries
    f32x4 fx[8];       // x tile
    int rp_cur = rowptr_t[(long)t * N + lrp];
    int base = __builtin_amdgcn_readlane(rp_cur, 0);
    int cnt = __builtin_amdgcn_readlane(rp_cur, N) - base;
    issue_tile(fg, g + (long)t * N * dout, ng4, lane);
    issue_cv(fg, cv_t, base, cnt, lane);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = lane + q * 64;
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      fx[q] = (i < nx4) ? ldv4(x + (long)t * N * din + (long)i * 4) : z;
    }
    int tn = t + nwaves;
    int rp_nxt = (tn < T) ? rowptr_t[(long)tn * N + lrp] : 0;

    for (;;) {
      // ---- 1. g[t], CSR(A^T) slice: registers -> LDS ------------------------------------------
      land(fg, ws.b, FD, ng4, dout4, ws, cv_t, rp_cur, base, cnt, N, lane);
      wave_sync();

      // ---- 2. dFW = A^T @ g -> dFW tile (odd stride), dbias partial ----------------------------
      aggregate_rows(ws, ws.b, FD, N, dout, lane, [&](int r, int cl, f32x4 acc) {
        float* d = ws.a + r * BLD + cl * 4;
        d[0] = acc[0]; d[1] = acc[1]; d[2] = acc[2]; d[3] = acc[3];
        dbacc += acc;
      });
      wave_sync();

      // ---- 3. x[t] -> gather tile (g is dead) --------------------------------------------------
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = lane + q * 64;
        if (i < nx4) {
          const int r = i / din4, c4 = i - r * din4;
          *reinterpret_cast<f32x4*>(ws.b + r * FD + c4 * 4) = fx[q];
        }
      }

      // ---- 4. next graph (g, CSR, x) in flight during the whole MFMA phase ---------------------
      const bool has_next = tn < T;
      int base_n = 0, cnt_n = 0;
      if (has_next) {
        base_n = __builtin_amdgcn_readlane(rp_nxt, 0);
        cnt_n = __builtin_amdgcn_readlane(rp_nxt, N) - base_n;
        issue_tile(fg, g + (long)tn * N * dout, ng4, lane);
        issue_cv(fg, cv_t, base_n, cnt_n, lane);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int i = lane + q * 64;
          f32x4 z = {0.f, 0.f, 0.f, 0.f};
          fx[q] = (i < nx4) ? ldv4(x + (long)tn * N * din + (long)i * 4) : z;
        }
        rp_cur = rp_nxt;
        const int tnn = tn + nwaves;
        rp_nxt = (tnn < T) ? rowptr_t[(long)tnn * N + lrp] : 0;
      }
      wave_sync();

      // ---- 5. dW += x^T @ dFW --------------------------------------------------------------------
#pragma unroll 2
      for (int s = 0; s < 16; ++s) {
        const int n = hi * 16 + s;
        const float a0 = ws.b[n * FD + li], a1 = ws.b[n * FD + 32 + li];
        const float f0 = ws.a[n * BLD + li], f1 = ws.a[n * BLD + 32 + li];
        dw00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, f0, dw00, 0, 0, 0);
        dw01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, f1, dw01, 0, 0, 0);
        dw10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, f0, dw10, 0, 0, 0);
        dw11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, f1, dw11, 0, 0, 0);
      }

      // ---- 6. dX = dFW @ W^T (A operand straight from the odd-stride LDS tile) ------------------
      if (dx) {
        f32x16 c0, c1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
#pragma unroll 4
        for (int s = 0; s < 32; ++s) {
          const int k = hi * 32 + s;
          const float a = ws.a[li * BLD + k];
          c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Wt[k * FD + li], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Wt[k * FD + 32 + li], c1, 0, 0, 0);
        }
        float* dxt = dx + (long)t * N * din;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (row < N) {
            if (li < din) dxt[(long)row * din + li] = c0[r];
            if (32 + li < din) dxt[(long)row * din + 32 + li] = c1[r];
          }
        }
      }
      wave_sync();

      if (!has_next) break;
      t = tn;
      tn += nwaves;
      base = base_n;
      cnt = cnt_n;
    }
  }

  // ---- reduce the workgroup's waves through LDS; one partial per workgroup leaves the chip -----
  // every wave parks its 64x64 dW tile (C layout -> row major) + dbias in its own slice
  __syncthreads();
  float* park = ws.a;  // a (8704 B) and b (8448 B) are contiguous: 4096 + 64 floats fit
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    park[row * FD + li] = dw00[r];
    park[row * FD + 32 + li] = dw01[r];
    park[(32 + row) * FD + li] = dw10[r];
    park[(32 + row) * FD + 32 + li] = dw11[r];
  }
  // dbacc: lane (sub, cl) holds the column-4-group cl summed over rows == sub (mod 4)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float v = dbacc[j];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    dbacc[j] = v;
  }
  if (lane < 16) *reinterpret_cast<f32x4*>(park + FD * FD + lane * 4) = dbacc;
  __syncthreads();
  const size_t slice_f = slice_bytes(max_nnz) / 4;
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

// waves per block that fit the LDS budget (0 = does not fit at all)
static int fused_wpb(int max_nnz, size_t shared_bytes) {
  const size_t per = slice_bytes(max_nnz);
  long fit = ((long)kLdsBytes - (long)shared_bytes) / (long)per;
  if (fit > MAX_WPB) fit = MAX_WPB;
  return fit < 1 ? 0 : (int)fit;
}

static int fused_grid(int T, int wpb) {
  int blocks = (T + wpb - 1) / wpb;
  if (blocks > kNumCU) blocks = kNumCU;  // one persistent workgroup per CU
  return blocks < 1 ? 1 : blocks;
}

static bool fused_shape_ok(int n, int din, int dout, int max_nnz) {
  if (n <= 0 || n > FN) return false;
  if (din <= 0 || din > FD || (din & 3)) return false;
  if (dout <= 0 || dout > FD || (dout & 3)) return false;
  if (max_nnz < 0) return false;
  return fused_wpb(max_nnz, FD * FD * 4) >= 4;
}

}  // namespace kgcn

using namespace kgcn;

extern "C" int kgcn_graphconv_fused_supported(int32_t n_nodes, int32_t din, int32_t dout,
                                              int32_t max_nnz_per_graph) {
  return fused_shape_ok(n_nodes, din, dout, max_nnz_per_graph) ? 1 : 0;
}

extern "C" int kgcn_graphconv_fwd_f32(const kgcn_csr_batch* a, const float* x, const float* w,
                                      const float* bias, int32_t din, int32_t dout, float* out,
                                      void* stream) {
  if (int rc = validate_csr(a, "kgcn_graphconv_fwd_f32")) return rc;
  if (a->rows != a->cols) return fail("kgcn_graphconv_fwd_f32: adjacency must be square");
  if (!fused_shape_ok(a->rows, din, dout, a->max_nnz_per_graph))
    return fail("kgcn_graphconv_fwd_f32: shape N=%d din=%d dout=%d max_nnz=%d not supported by the "
                "fused kernel (use kgcn_dense_fwd_f32 + kgcn_bconv_f32)",
                a->rows, din, dout, a->max_nnz_per_graph);
  if (a->num_graphs == 0) return 0;
  if (!x || !w || !out) return fail("kgcn_graphconv_fwd_f32: NULL operand");
  if (!aligned16(x) || !aligned16(out)) return fail("kgcn_graphconv_fwd_f32: x/out not 16-byte aligned");
  const int wpb = fused_wpb(a->max_nnz_per_graph, 0);
  const size_t lds = (size_t)wpb * slice_bytes(a->max_nnz_per_graph);
  static thread_local bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(graphconv_fwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    attr_set = true;
  }
  hipLaunchKernelGGL(graphconv_fwd_kernel, dim3(fused_grid(a->num_graphs, wpb)), dim3(64 * wpb),
                     lds, as_stream(stream), a->rowptr, reinterpret_cast<const int2*>(a->cv), x, w,
                     bias, out, a->num_graphs, a->rows, din, dout, a->max_nnz_per_graph);
  return check_launch("graphconv_fwd_kernel");
}

extern "C" int64_t kgcn_graphconv_bwd_workspace_bytes(int32_t num_graphs, int32_t din,
                                                      int32_t dout) {
  if (num_graphs <= 0 || din <= 0 || dout <= 0) return 0;
  // one partial per persistent workgroup (at most one workgroup per CU)
  return (int64_t)kNumCU * ((int64_t)din * dout + dout) * 4;
}

extern "C" int kgcn_graphconv_bwd_f32(const kgcn_csr_batch* at, const float* x, const float* w,
                                      const float* dout_grad, int32_t din, int32_t dout, float* dx,
                                      float* dw, float* dbias, void* workspace,
                                      int64_t workspace_bytes, void* stream) {
  if (int rc = validate_csr(at, "kgcn_graphconv_bwd_f32")) return rc;
  if (at->rows != at->cols) return fail("kgcn_graphconv_bwd_f32: adjacency must be square");
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
  if (!aligned16(x) || !aligned16(dout_grad) || (dx && !aligned16(dx)))
    return fail("kgcn_graphconv_bwd_f32: tensors not 16-byte aligned");
  const int wpb = fused_wpb(at->max_nnz_per_graph, FD * FD * 4);
  const int blocks = fused_grid(at->num_graphs, wpb);
  const int64_t need = (int64_t)blocks * ((int64_t)din * dout + dout) * 4;
  if (!workspace || workspace_bytes < need)
    return fail("kgcn_graphconv_bwd_f32: workspace %lld < %lld bytes", (long long)workspace_bytes,
                (long long)need);
  float* part_dw = static_cast<float*>(workspace);
  float* part_db = part_dw + (long)blocks * din * dout;
  const size_t lds = FD * FD * 4 + (size_t)wpb * slice_bytes(at->max_nnz_per_graph);
  static thread_local bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(graphconv_bwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    attr_set = true;
  }
  hipLaunchKernelGGL(graphconv_bwd_kernel, dim3(blocks), dim3(64 * wpb), lds, s, at->rowptr,
                     reinterpret_cast<const int2*>(at->cv), x, w, dout_grad, dx, part_dw, part_db,
                     at->num_graphs, at->rows, din, dout, at->max_nnz_per_graph);
  if (int rc = check_launch("graphconv_bwd_kernel")) return rc;
  if (int rc = launch_reduce_partials(part_dw, blocks, (long)din * dout, dw, s)) return rc;
  return launch_reduce_partials(part_db, blocks, dout, dbias, s);
}
