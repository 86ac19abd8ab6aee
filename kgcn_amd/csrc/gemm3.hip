// Tall-skinny fp32 GEMMs on the bf16 matrix pipe (exact 3-way split, kgcn_common.h) for the wide layers
// of the path (BASELINE configs 3-5: 128- and 256-dim features, example_model/model_multitask.py:51-57,
// example_model/sparse.py:30):
//
//   y[m, dout] = x[m, din] @ W (+ bias)   /   dx = dy @ W^T (trans_w)      gemm3_fwd_kernel
//   dW[din, dout] = x^T @ dy, dbias = colsum(dy)                           gemm3_wgrad_kernel
//
// Above ~128 x 128 weights the v_mfma_f32_32x32x2_f32 kernels of dense.hip are bound by the f32 matrix
// rate (157 TF: measured 74-96 TF at 256 x 256, hipBLASLt 82-119).  Six v_mfma_f32_32x32x16_bf16 products
// per 16 k-values cost 192 matrix-pipe cycles against 512 for eight f32 MFMAs and leave the vector ALU
// free for the splitting.
//
// Forward: one workgroup (8 waves) per 128 rows x 256 columns, persistent over row tiles, so x is read
// from HBM exactly once.  Per 32-wide k chunk the 512 threads stage x [128 x 32] and W [32 x 256]:
// global -> registers (two raw sets alternate: chunk i+2 is requested while chunk i+1 is split) -> exact
// 3-way split -> LDS, already in MFMA fragment order
//   XP[(mt, ks, piece)][lane] / WP[(nt, ks, piece)][lane]   (16 bytes per lane: 8 k-values as bf16)
// so that every operand read of the inner loop is one conflict-free ds_read_b128.  Wave (wr, wc) owns
// rows 64 wr .. +63 and columns 64 wc .. +63: per k-step 6 + 6 fragment reads feed 24 MFMAs (2 x 2 tiles x
// 6 products), and behind each MFMA sits one slice of the staging of the next chunk (a split_pair, a
// fragment write, the next loads).  LDS: 2 buffers x (24 KB + 48 KB); one workgroup barrier per chunk.
// Measured (tools/dense_shapes.py, 1M rows, 256 x 256): 1.00 ms = 131 TF (dense.hip f32 MFMA: 1.41 ms,
// hipBLASLt fp32: 1.07-1.13 ms); compiled-out experiments: MFMAs + y stores alone 0.68 ms, staging alone
// 0.54 ms -- the workgroup-wide lockstep (barrier per chunk, y stores of all waves at once) keeps the
// two from overlapping fully; that is where the remaining time is.
#include <cstdlib>
#include <cstring>

#include "kgcn_common.h"

namespace kgcn {

// the f16 two-piece kernels (gemmh.hip) take the shapes they were built for; their W' lives behind the bf16 section of the table
bool gemmh_fwd_ok(const float* x, long m, int din, long x_ld, int dout);
int launch_gemmh_fwd(const float* x, long m, int din, long x_ld, const void* tabh, const float* bias, float* y, int dout,
                     long y_ld, int act, hipStream_t s);
int launch_gemmh_dx_dact(const float* grad, const float* act_out, float* dpre, long m, int k, long ld, const void* tabh,
                         float* dx, int n, long dx_ld, int dact, hipStream_t s, const float* pooled_grad, int n_nodes, long pooled_ld,
                         float* dot_part);
int64_t wtable_bf16_bytes(int din, int dout);

#ifdef KGCN_PROBE   // development: per-workgroup cycle sums per phase (tools/gemm3_probe.py)
__device__ long long* g3_probe = nullptr;
#define G3P_DECL long long pt_[4] = {0, 0, 0, 0}; long long pc_ = __builtin_readcyclecounter();
#define G3P(k) { const long long n_ = __builtin_readcyclecounter(); pt_[k] += n_ - pc_; pc_ = n_; }
#define G3P_FLUSH if (g3_probe && (threadIdx.x & 63) == 0) { for (int k_ = 0; k_ < 4; ++k_) g3_probe[((long)blockIdx.x * 8 + (threadIdx.x >> 6)) * 4 + k_] = pt_[k_]; }
#else
#define G3P_DECL
#define G3P(k)
#define G3P_FLUSH
#endif

constexpr int G3_BM = 128;            // rows per workgroup tile
constexpr int G3_BN = 256;            // columns per workgroup
constexpr int G3_BK = 32;             // k chunk = 2 bf16 k-steps
// One fragment block = the 64 lanes' 16-byte entries of one (tile, k-step, piece).  The staging threads of
// a wave write four (k-step, half) groups at once, 512 / 3072 bytes apart, i.e. on the same banks: entry
// li of group g = 2 ks + hi is therefore stored at li ^ 4g (a 16-bank rotation per group) -- conflict-free
// ds_write_b128, and the fragment reads still cover one contiguous 512-byte half block per 32 lanes.
constexpr int G3_XP = 4 * 2 * 3 * 64;   // u32x4 entries of one x-piece buffer  (m-tile, k-step, piece, lane)
constexpr int G3_WP = 8 * 2 * 3 * 64;   // u32x4 entries of one W-piece buffer  (n-tile, k-step, piece, lane)
constexpr int G3_FB = 64;
__device__ __forceinline__ int g3_slot(int tile, int ks, int piece, int li, int hi) {
  return ((tile * 2 + ks) * 3 + piece) * G3_FB + (li ^ (4 * (2 * ks + hi))) + 32 * hi;
}
constexpr size_t G3_LDS = 2 * (size_t)(G3_XP + G3_WP) * 16;

// Raw (unsplit) data of one k chunk in flight: what one thread stages.  Two sets alternate by chunk parity
// so that chunk i+2 can be requested while chunk i+1 is being split.
struct G3Raw {
  f32x4 xa, xb;        // x[row][k0 + 8 qx .. +7]
  f32x4 ya, yb;        // DK != 0: the saved layer output at the same positions (x is then the incoming gradient)
  f32x4 ba, bb;        // DK != 0 with a gathered gradient: d pooled of the row's graph at the same columns
  float w0[8], w1[8];  // W^(T)[k0 + 8 q + j][n] for the thread's two (n, q) tasks
};

// per-thread staging coordinates (fixed for the whole launch / for one row tile)
struct G3Coord {
  const float* xrow;   // clamped row of the current tile
  const float* brow;   // gathered gradient: d pooled row of the row's graph (else nullptr)
  bool bc_only;        // ... and no per-row gradient besides it
  bool rowok;
  int qx;              // x task: 8-k group 0..3
  unsigned wbase[2];   // W tasks: clamped column offset (elements)
  bool nok[2];
  int qw[2];
  unsigned sk;         // element stride of k in W (w_ld, or 1 when transposed)
  bool wvec;           // transposed source with 16-byte aligned rows and din % 4 == 0
};

template <bool XVEC, bool WTAB = false, int DK = 0>
__device__ __forceinline__ void g3_issue(G3Raw& r, const G3Coord& c, const float* __restrict__ w, int din, int k0,
                                         long ydiff = 0) {
  // loads read clamped (always valid) addresses; masking happens when the data is split (g3_land), so no
  // select sits between a load and its first real use
  const int k = k0 + 8 * c.qx;
  if constexpr (XVEC) {
    if (DK != 0 && c.bc_only) {                          // uniform: the gradient is the broadcast alone
      r.xa = f32x4{0.f, 0.f, 0.f, 0.f};
      r.xb = r.xa;
    } else {
      r.xa = *reinterpret_cast<const f32x4*>(c.xrow + (k < din ? k : 0));
      r.xb = *reinterpret_cast<const f32x4*>(c.xrow + (k + 4 < din ? k + 4 : 0));
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      r.xa[j] = c.xrow[k + j < din ? k + j : 0];
      r.xb[j] = c.xrow[k + 4 + j < din ? k + 4 + j : 0];
    }
  }
  if constexpr (DK != 0) {         // XVEC only
    r.ya = *reinterpret_cast<const f32x4*>(c.xrow + ydiff + (k < din ? k : 0));
    r.yb = *reinterpret_cast<const f32x4*>(c.xrow + ydiff + (k + 4 < din ? k + 4 : 0));
    if (c.brow) {                                        // uniform
      r.ba = *reinterpret_cast<const f32x4*>(c.brow + (k < din ? k : 0));
      r.bb = *reinterpret_cast<const f32x4*>(c.brow + (k + 4 < din ? k + 4 : 0));
    }
  }
  if constexpr (WTAB) return;      // W arrives pre-split from the fragment table
  if (c.wvec) {          // transposed source, rows 16-byte aligned: 8 k-values = 2 x 16 bytes
    const int ka = k0 + 8 * c.qw[0], kb = k0 + 8 * c.qw[1];
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(w + c.wbase[0] + (ka < din ? ka : 0));
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(w + c.wbase[0] + (ka + 4 < din ? ka + 4 : 0));
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(w + c.wbase[1] + (kb < din ? kb : 0));
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(w + c.wbase[1] + (kb + 4 < din ? kb + 4 : 0));
#pragma unroll
    for (int j = 0; j < 4; ++j) { r.w0[j] = a0[j]; r.w0[4 + j] = a1[j]; r.w1[j] = b0[j]; r.w1[4 + j] = b1[j]; }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k0j = k0 + 8 * c.qw[0] + j, k1j = k0 + 8 * c.qw[1] + j;
      r.w0[j] = w[c.wbase[0] + (unsigned)(k0j < din ? k0j : 0) * c.sk];
      r.w1[j] = w[c.wbase[1] + (unsigned)(k1j < din ? k1j : 0) * c.sk];
    }
  }
}

// split step `step` (0..11) of the staged chunk: 0-3 x pairs, 4-7 W task 0, 8-11 W task 1
template <int STEP>
__device__ __forceinline__ void g3_split_step(const G3Raw& r, const G3Coord& c, int din, int k0, Frag3& fx, Frag3& f0,
                                              Frag3& f1) {
  constexpr int J = STEP & 3;
  unsigned q1, q2, q3;
  if constexpr (STEP < 4) {
    const int k = k0 + 8 * c.qx + 2 * J;
    const float a = (J < 2) ? r.xa[2 * J] : r.xb[2 * J - 4], b = (J < 2) ? r.xa[2 * J + 1] : r.xb[2 * J - 3];
    split_pair((c.rowok && k < din) ? a : 0.f, (c.rowok && k + 1 < din) ? b : 0.f, q1, q2, q3);
    fx.p1[J] = q1; fx.p2[J] = q2; fx.p3[J] = q3;
  } else if constexpr (STEP < 8) {
    const int k = k0 + 8 * c.qw[0] + 2 * J;
    split_pair((c.nok[0] && k < din) ? r.w0[2 * J] : 0.f, (c.nok[0] && k + 1 < din) ? r.w0[2 * J + 1] : 0.f, q1, q2, q3);
    f0.p1[J] = q1; f0.p2[J] = q2; f0.p3[J] = q3;
  } else {
    const int k = k0 + 8 * c.qw[1] + 2 * J;
    split_pair((c.nok[1] && k < din) ? r.w1[2 * J] : 0.f, (c.nok[1] && k + 1 < din) ? r.w1[2 * J + 1] : 0.f, q1, q2, q3);
    f1.p1[J] = q1; f1.p2[J] = q2; f1.p3[J] = q3;
  }
}

__device__ __forceinline__ void g3_write(u32x4* table, int tile, int q, int li, const Frag3& f) {
  u32x4* d = table + g3_slot(tile, q >> 1, 0, li, q & 1);
  d[0] = f.p1; d[G3_FB] = f.p2; d[2 * G3_FB] = f.p3;
}

// WTAB: W comes pre-split from the fragment table of wtable.hip (wtable_split_kernel: one 1 KiB lane-linear block per
// (k-step, 32-column tile, piece), L2 resident): every wave reads the B fragments of its two column tiles straight into
// registers one k-step ahead -- no W loads, no W split (2/3 of the staging arithmetic), no W LDS traffic; LDS holds the
// x pieces only (48 KB).  `w` is then the table.
// MW: row-waves of the workgroup.  2: 8 waves, 128-row tile (one workgroup per CU).  1 (table variant only): 4 waves, 64-row
// tile, TWO workgroups per CU with their own barriers -- while one workgroup waits at its chunk barrier the other one's
// waves own the pipes.
// DK (table variant, XVEC): backward of an activated dense layer in ONE pass -- x is the incoming gradient g, the operand
// of the contraction is dpre = g (.) act'(a) with a = the saved layer output (same layout, `ydiff` elements away): the
// staging threads multiply while they split and write dpre (`pdiff` elements away from x) for the weight-gradient GEMM, so
// the stand-alone activation-backward pass (3 x 4 bytes per element) disappears.  DK = 1: act' = c0 + c1 a + c2 a^2
// (sigmoid 0, 1, -1; tanh 1, 0, -1), DK = 2: relu (a > 0).  Every store goes to the clamped address its load came from:
// threads whose row or k is out of range re-store values another thread stores too.
// Gathered gradient (bc != nullptr): the incoming gradient of row r is g[r] + bc[(r / bc_n) * bc_ld + .] -- the layer's output was
// read out by GraphGather (kgcn/layers.py:163-164: d pooled reaches every node row of its graph) and, unless bc_only, also handed
// on (g): the broadcast never exists in HBM.
struct G3Dact { long ydiff, pdiff; float c0, c1, c2; const float* bc; long bc_ld; int bc_n, bc_only; };

// MT x NT (table variant, 4-wave workgroups): 32 x 32 tiles per wave.  2 x 2: the workgroup covers 64 rows x 256 columns
// (waves side by side).  When a launch has fewer such tiles than a quarter of the chip's workgroup slots (2 per CU) --
// sparse.py's 4,457 node rows are 70 tiles -- the columns are cut instead: 1 x 1 = 64 rows x 64 columns (waves 2 x 2),
// blockIdx.y = the column block.  Every column block stages the x pieces of its rows again (latency-bound launches: the
// staging is not what they wait for); d pre-activation is written by column block 0 only.
template <bool XVEC, bool WTAB, int MW, int DK = 0, int MT = 2, int NT = 2>
__global__ __launch_bounds__(256 * MW, 2) void gemm3_fwd_kernel(
    const float* __restrict__ x, long m, int din, long x_ld, const float* __restrict__ w, long w_ld, int trans_w,
    const float* __restrict__ bias, float* __restrict__ y, int dout, long y_ld, int act, G3Dact da) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  u32x4* lds = reinterpret_cast<u32x4*>(dsm);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, hi = lane >> 5;
  static_assert(MW == 2 || WTAB, "the 4-wave workgroup stages x only");
  static_assert((MT == 2 && NT == 2) || (MW == 1 && WTAB && (MT == 2 || NT == 1)), "narrow column blocks: table variant, 4 waves");
  constexpr int WC = (MT == 1) ? 2 : 4;                    // waves side by side
  constexpr int BN = WC * NT * 32;                         // columns per workgroup
  static_assert(DK == 0 || (WTAB && XVEC), "the activation-derivative prologue lives in the table variant");
  // g (.) act'(a) for the pair J of a raw set, in place (before the pair is split)
  auto dact_pair = [&](G3Raw& R, auto jc) __attribute__((always_inline)) {
    constexpr int J = decltype(jc)::value;
    if constexpr (DK != 0) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float a = (J < 2) ? R.ya[2 * J + e] : R.yb[2 * J - 4 + e];
        float g = (J < 2) ? R.xa[2 * J + e] : R.xb[2 * J - 4 + e];
        if (da.bc) g += (J < 2) ? R.ba[2 * J + e] : R.bb[2 * J - 4 + e];
        if constexpr (DK == 1) g *= __builtin_fmaf(__builtin_fmaf(da.c2, a, da.c1), a, da.c0);
        else g = a > 0.f ? g : 0.f;
        if constexpr (J < 2) R.xa[2 * J + e] = g;
        else R.xb[2 * J - 4 + e] = g;
      }
    }
  };
  auto store_dpre = [&](const G3Raw& R, const float* xrow, int k0) __attribute__((always_inline)) {
    if constexpr (DK != 0) {
      if (blockIdx.y != 0) return;                         // uniform: ONE column block writes d pre-activation
      const int k = k0 + 8 * (tid & 3);
      float* p = const_cast<float*>(xrow) + da.pdiff;
      *reinterpret_cast<f32x4*>(p + (k < din ? k : 0)) = R.xa;
      *reinterpret_cast<f32x4*>(p + (k + 4 < din ? k + 4 : 0)) = R.xb;
    }
  };
  constexpr int BMT = 64 * MW;                             // rows per workgroup tile
  const int wr = wave / WC, wc = wave % WC;
  const int n0 = blockIdx.y * BN;
  const long ntiles = (m + BMT - 1) / BMT;
  const int nkc = (din + G3_BK - 1) / G3_BK;
  if ((long)blockIdx.x >= ntiles) return;      // uniform for the whole workgroup

  float bcol[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int c = n0 + 32 * NT * wc + 32 * nt + li;
    bcol[nt] = (bias && c < dout) ? bias[c] : 0.f;
  }

  // staging tasks: x (row = tid / 4, q = tid % 4): 8 consecutive k of one row (coalesced 128-byte row
  // segments); W (n = id % 256, q = id / 256) for id = tid and tid + 512: 8 k-values of one column
  const int xr = tid >> 2;
  G3Coord co;
  co.qx = tid & 3;
  co.brow = nullptr;
  co.bc_only = DK != 0 && da.bc_only != 0;
  co.sk = trans_w ? 1u : (unsigned)w_ld;
  co.wvec = trans_w && (w_ld % 4 == 0) && (din % 4 == 0) && ((reinterpret_cast<uintptr_t>(w) & 15u) == 0);
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2) {
    const int id = tid + 512 * t2, n = n0 + (id & 255);
    co.qw[t2] = id >> 8;
    co.nok[t2] = n < dout;
    co.wbase[t2] = (unsigned)(n < dout ? n : dout - 1) * (trans_w ? (unsigned)w_ld : 1u);
  }
  auto set_tile = [&](long tile) __attribute__((always_inline)) {
    const long row = tile * BMT + xr;
    co.rowok = row < m;
    co.xrow = x + (row < m ? row : m - 1) * x_ld;
    if constexpr (DK != 0) co.brow = da.bc ? da.bc + ((row < m ? row : m - 1) / da.bc_n) * da.bc_ld : nullptr;
  };

  // flattened (tile, k chunk) sequence: coordinates of the chunks being multiplied (0), split (1), loaded (2)
  long t0 = blockIdx.x, t1, t2c;
  int k0c = 0, k1c, k2c;
  auto advance = [&](long t, int kc, long& tn, int& kn) __attribute__((always_inline)) {
    if (kc + 1 < nkc) { tn = t; kn = kc + 1; } else { tn = t + gridDim.x; kn = 0; }
  };
  advance(t0, k0c, t1, k1c);
  advance(t1, k1c, t2c, k2c);

  G3Raw ra, rb;
  Frag3 fx, f0, f1;
  // prologue: chunk 0 staged without overlap, chunk 1 requested
  // WTAB: B fragments of (k-step g, this wave's column tiles); k-steps beyond the table (din % 32 in 1..16) and column
  // tiles beyond it (dout % 64 in 1..32 under a 256-column block) are clamped: their products meet zero x pieces or
  // unstored columns
  constexpr int LBUF = WTAB ? G3_XP : G3_XP + G3_WP;     // u32x4 entries of one LDS buffer
  const u32x4* wtab = reinterpret_cast<const u32x4*>(w);
  const int tab_nt = ((dout + 63) / 64) * 2, tab_ks = (din + 15) / 16;
  u32x4 B0[NT][3], B1[NT][3];
  auto load_b = [&](u32x4 (&Bd)[NT][3], int g) __attribute__((always_inline)) {
    const int gg = g < tab_ks ? g : tab_ks - 1;
#pragma unroll
    for (int t2 = 0; t2 < NT; ++t2) {
      int nt = (n0 + 32 * NT * wc) / 32 + t2;
      nt = nt < tab_nt ? nt : tab_nt - 1;
      const u32x4* e = wtab + ((long)(gg * tab_nt + nt) * 3) * 64 + lane;
#pragma unroll
      for (int p = 0; p < 3; ++p) Bd[t2][p] = e[64 * p];
    }
  };
  set_tile(t0);
  g3_issue<XVEC, WTAB, DK>(ra, co, w, din, k0c * G3_BK, da.ydiff);
  static_for<(WTAB ? 4 : 12)>([&](auto sc) __attribute__((always_inline)) {
    if constexpr (decltype(sc)::value < 4) dact_pair(ra, sc);
    g3_split_step<decltype(sc)::value>(ra, co, din, k0c * G3_BK, fx, f0, f1);
  });
  store_dpre(ra, co.xrow, k0c * G3_BK);
  g3_write(lds, xr >> 5, co.qx, xr & 31, fx);
  if constexpr (!WTAB) {
    g3_write(lds + G3_XP, (tid & 255) >> 5, co.qw[0], tid & 31, f0);
    g3_write(lds + G3_XP, (tid & 255) >> 5, co.qw[1], tid & 31, f1);
  } else {
    load_b(B0, 0);
  }
  set_tile(t1 < ntiles ? t1 : t0);
  g3_issue<XVEC, WTAB, DK>(rb, co, w, din, (t1 < ntiles ? k1c : k0c) * G3_BK, da.ydiff);
  __syncthreads();

  f32x16 acc[MT][NT];
  int buf = 0;
  bool done = false;
  float touch = 0.f, sink = 0.f;
  G3P_DECL
  // one pipeline step: multiply chunk (t0, k0c) out of LDS buffer `buf` || split chunk (t1, k1c) (raw set RS)
  // into buffer buf^1 || request chunk (t2c, k2c) into raw set RL
  auto step = [&](G3Raw& RS, G3Raw& RL) __attribute__((always_inline)) {
    const bool have1 = t1 < ntiles;                      // uniform
    const int ks1 = (have1 ? k1c : k0c) * G3_BK;
    // coordinates of the chunk being SPLIT: its row mask (co.rowok) was set when it was requested; keep a copy
    const bool rowok_split = co.rowok;
    const bool have2 = t2c < ntiles;
    // request chunk 2 (clamped to a valid chunk when the sequence ends)
    const long tl = have2 ? t2c : (have1 ? t1 : t0);
    const int kl = have2 ? k2c : (have1 ? k1c : k0c);
    const long rowl = tl * BMT + xr;
    const float* xrow_l = x + (rowl < m ? rowl : m - 1) * x_ld;
    const bool rowok_l = rowl < m;
    const float* brow_l = co.brow;                       // gathered gradient: the row's graph changes with the tile
    if constexpr (DK != 0) {
      if (kl == 0 && da.bc) brow_l = da.bc + ((rowl < m ? rowl : m - 1) / da.bc_n) * da.bc_ld;      // uniform
    }
    if (k0c == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][nt][r] = bcol[nt];
    }
    const u32x4* xp = lds + buf * LBUF;
    const u32x4* wp = xp + G3_XP;
    u32x4* xq = lds + (buf ^ 1) * LBUF;
    u32x4* wq = xq + G3_XP;
    // k-step after this chunk's second one: first of the next chunk of the tile, or of the next tile (k-step 0)
    const int gnext = (k0c + 1 < nkc) ? 2 * (k0c + 1) : 0;
    G3Coord cs = co;
    cs.rowok = rowok_split;
    G3P(0)
    static_for<2>([&](auto ksc) __attribute__((always_inline)) {
      constexpr int ks = decltype(ksc)::value;
      u32x4 A[MT][3], B[NT][3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int t2 = 0; t2 < MT; ++t2) A[t2][p] = xp[g3_slot(MT * wr + t2, ks, p, li, hi)];
#pragma unroll
        for (int t2 = 0; t2 < NT; ++t2) {
          if constexpr (!WTAB) B[t2][p] = wp[g3_slot(2 * wc + t2, ks, p, li, hi)];
          else B[t2][p] = ks == 0 ? B0[t2][p] : B1[t2][p];
        }
      }
      constexpr int NMF = 6 * MT * NT;                     // MFMAs per k-step
      constexpr int SL = 24 / NMF;                         // staging slices behind each of them (the slice plan has 48 slots)
      static_for<NMF * SL>([&](auto mc) __attribute__((always_inline)) {
        constexpr int mm = decltype(mc)::value / SL, pr = mm / (MT * NT), tl4 = mm % (MT * NT), slot = 24 * ks + decltype(mc)::value;
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};   // smallest terms first
        if constexpr (decltype(mc)::value % SL == 0)
          acc[tl4 / NT][tl4 % NT] = mfma_bf16(A[tl4 / NT][PA[pr]], B[tl4 % NT][PB[pr]], acc[tl4 / NT][tl4 % NT]);
        // ---- one slice of the staging work behind every MFMA (the matrix pipe runs beside the VALU) ----
        if constexpr (WTAB && decltype(mc)::value == SL) {
          // the other fragment set <- the k-step after this one (its previous contents were consumed one k-step ago)
          if constexpr (ks == 0) load_b(B1, 2 * k0c + 1);
          else load_b(B0, gnext);
        }
        if constexpr (WTAB && slot == 5) {
          store_dpre(RS, cs.xrow, ks1);
        } else if constexpr (WTAB && slot >= 4 && slot < 12) {
        } else if constexpr (WTAB && (slot == 13 || slot == 14)) {
        } else if constexpr (slot < 12) {
          if constexpr (slot < 4) dact_pair(RS, std::integral_constant<int, slot>{});
          g3_split_step<slot>(RS, cs, din, ks1, fx, f0, f1);
        } else if constexpr (slot == 12) {
          g3_write(xq, xr >> 5, co.qx, xr & 31, fx);
        } else if constexpr (slot == 13) {
          g3_write(wq, (tid & 255) >> 5, co.qw[0], tid & 31, f0);
        } else if constexpr (slot == 14) {
          g3_write(wq, (tid & 255) >> 5, co.qw[1], tid & 31, f1);
        } else if constexpr (slot == 16) {
          co.xrow = xrow_l;
          co.rowok = rowok_l;
          co.brow = brow_l;
          // pinned: hipcc sinks these requests to the END of the chunk (shorter live ranges), i.e. right in front of the step that
          // splits them -- two thirds of a chunk of lead time gone (the ISA showed the loads behind the last MFMAs and vmcnt(0)
          // waits at the head of the next chunk)
          __builtin_amdgcn_sched_barrier(0);
          g3_issue<XVEC, WTAB, DK>(RL, co, w, din, kl * G3_BK, da.ydiff);
          __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (slot == 18) {
          // one more chunk of x on its way from HBM: a single k chunk per workgroup in flight (16 KB) caps the
          // read rate at ~2 TB/s (latency x bytes in flight); this 4-byte load per 32-byte piece pulls the line
          // of the chunk AFTER the requested one into L2 and is consumed only as a dead add, late
          long tp; int kp;
          if (kl + 1 < nkc) { tp = tl; kp = kl + 1; } else { tp = tl + gridDim.x; kp = 0; }
          if (tp >= ntiles) { tp = tl; kp = kl; }
          const long rowp = tp * BMT + xr;
          const int kq = kp * G3_BK + 8 * co.qx;
          touch = x[(rowp < m ? rowp : m - 1) * x_ld + (kq < din ? kq : 0)];
        } else if constexpr (slot == 47) {
          sink += touch;
        }
      });
    });
    G3P(1)
    if (k0c + 1 == nkc) {                                // last chunk of the tile: y <- act(acc)
      if (act != KGCN_ACT_NONE) {                        // ONE uniform branch: the plain epilogue stays what it was
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = act_fwd(acc[mt][nt][r], act);
      }
      const long row0 = t0 * BMT + 32 * MT * wr;
      const int cb = n0 + 32 * NT * wc;
      if (row0 + 32 * MT <= m && cb + 32 * NT <= dout) {
        // interior block (wave-uniform test): no masks, one running row pointer (the masked form below costs
        // ~12k cycles per tile in compares, branches and 64-bit multiplies)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          float* p = y + (row0 + 32 * mt + 4 * hi) * y_ld + cb + li;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) p[32 * nt] = acc[mt][nt][r];
            p += ((r & 3) == 3) ? 5 * y_ld : y_ld;          // rows 0-3, 8-11, 16-19, 24-27 (+ 4 hi)
          }
        }
      } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int c = cb + 32 * nt + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const long row = row0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hi;
              if (row < m && c < dout) y[row * y_ld + c] = acc[mt][nt][r];
            }
          }
      }
    }
    done = !have1;
    G3P(2)
    __syncthreads();
    G3P(3)
    buf ^= 1;
    t0 = t1; k0c = k1c;
    t1 = t2c; k1c = k2c;
    advance(t1, k1c, t2c, k2c);
  };
  for (;;) {
    step(rb, ra);        // chunk 1 sits in rb (requested in the prologue / previous step)
    if (done) break;
    step(ra, rb);
    if (done) break;
  }
  if (sink == 1.2345e-30f && tid == 4097) y[0] = sink;   // never true: keeps the touch loads alive
  G3P_FLUSH
}

// Table variant, 4-wave workgroups: how a launch is cut into workgroup jobs.  Fewer 64-row tiles than a quarter of the
// workgroups the chip runs at once (two per CU): 64-column blocks (sparse.py's 4,457 node rows: 70 tiles -> 280 workgroups;
// 20.3 against 23.8 us per 256 x 256 layer, its training step 0.335 against 0.362 ms).  Measured and NOT kept
// (tools/gemm_cut_bench.py, profiles/r03_i_gemm_cut.txt): 128-column blocks up to half of the slots (equal or slower), and
// a second, narrow-block launch over the rows of a last, partial round (200,000 rows = 6 rounds + 53 tiles: 190.7 against
// 186.6 us -- the lone workgroups of the seventh round run faster than the launch gap + a latency-bound launch cost).
template <int DK, int MT, int NT>
static void g3_table_launch_shape(const float* x, long m, int din, long x_ld, const float* tw, long w_ld, int trans_w,
                                  const float* bias, float* y, int dout, long y_ld, int act, const G3Dact& da, hipStream_t s) {
  constexpr int BN = (MT == 1 ? 2 : 4) * NT * 32;
  const long nt64 = (m + 63) / 64;
  const long cap = 2L * kNumCU;
  const dim3 grid((unsigned)(nt64 < cap ? nt64 : cap), (unsigned)((dout + BN - 1) / BN));
  const size_t lds = 2 * (size_t)G3_XP * 16;
  hipLaunchKernelGGL((gemm3_fwd_kernel<true, true, 1, DK, MT, NT>), grid, dim3(256), lds, s, x, m, din, x_ld, tw, w_ld, trans_w,
                     bias, y, dout, y_ld, act, da);
}

template <int DK>
static void g3_table_launch(const float* x, long m, int din, long x_ld, const float* tw, long w_ld, int trans_w, const float* bias,
                            float* y, int dout, long y_ld, int act, const G3Dact& da, hipStream_t s) {
  const long jobs = ((m + 63) / 64) * ((dout + G3_BN - 1) / G3_BN), slots = 2L * kNumCU;
  static const char* knob = dev_knob("KGCN_GEMM3_CUT");       // development: "0" = whole column blocks only
  // the backward form carries more staging per row (two or three operands, the d pre-activation store): it pays up to 96 tiles
  const long limit = DK == 0 ? slots : slots * 3 / 4;
  if (!(knob && knob[0] == '0') && jobs * 4 <= limit)
    return g3_table_launch_shape<DK, 1, 1>(x, m, din, x_ld, tw, w_ld, trans_w, bias, y, dout, y_ld, act, da, s);
  g3_table_launch_shape<DK, 2, 2>(x, m, din, x_ld, tw, w_ld, trans_w, bias, y, dout, y_ld, act, da, s);
}

// table == nullptr: W is split inside the kernel; else `table` is the fragment table of wtable.hip for (w, trans_w)
int launch_gemm3_fwd(const float* x, long m, int din, long x_ld, const float* w, long w_ld, int trans_w,
                     const float* bias, float* y, int dout, long y_ld, int act, const void* table, hipStream_t s) {
  static thread_local bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm3_fwd_kernel<true, false, 2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm3_fwd_kernel<false, false, 2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm3_fwd_kernel<true, true, 2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm3_fwd_kernel<false, true, 2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    attr_set = true;
  }
  const long ntiles = (m + G3_BM - 1) / G3_BM;
  const bool xvec = (din % 4 == 0) && (x_ld % 4 == 0) && aligned16(x);
  static const char* hknob = dev_knob("KGCN_GEMMH");           // development: the f16 kernels to use, e.g. "fw" (f forward / dX, d dX with act', w weight gradient); "0" = none
  if (table && !(hknob && !strchr(hknob, 'f')) && gemmh_fwd_ok(x, m, din, x_ld, dout))
    return launch_gemmh_fwd(x, m, din, x_ld, static_cast<const char*>(table) + wtable_bf16_bytes(din, dout), bias, y, dout, y_ld,
                            act, s);
  if (table) {
    const size_t lds = 2 * (size_t)G3_XP * 16;
    const float* tw = static_cast<const float*>(table);
    static const char* mw = dev_knob("KGCN_GEMM3_MW");         // development: "2" = the 8-wave workgroup
    if (!(mw && mw[0] == '2')) {
      if (xvec) {
        g3_table_launch<0>(x, m, din, x_ld, tw, w_ld, trans_w, bias, y, dout, y_ld, act, G3Dact{}, s);
      } else {
        const long nt64 = (m + 63) / 64;
        const long cap = 2L * kNumCU;
        const dim3 grid((unsigned)(nt64 < cap ? nt64 : cap), (unsigned)((dout + G3_BN - 1) / G3_BN));
        hipLaunchKernelGGL((gemm3_fwd_kernel<false, true, 1>), grid, dim3(256), lds, s, x, m, din, x_ld, tw, w_ld, trans_w,
                           bias, y, dout, y_ld, act, G3Dact{});
      }
      return check_launch("gemm3_fwd_kernel");
    }
    const dim3 grid((unsigned)(ntiles < kNumCU ? ntiles : kNumCU), (unsigned)((dout + G3_BN - 1) / G3_BN));
    if (xvec)
      hipLaunchKernelGGL((gemm3_fwd_kernel<true, true, 2>), grid, dim3(512), lds, s, x, m, din, x_ld, tw, w_ld, trans_w, bias,
                         y, dout, y_ld, act, G3Dact{});
    else
      hipLaunchKernelGGL((gemm3_fwd_kernel<false, true, 2>), grid, dim3(512), lds, s, x, m, din, x_ld, tw, w_ld, trans_w,
                         bias, y, dout, y_ld, act, G3Dact{});
    return check_launch("gemm3_fwd_kernel");
  }
  const dim3 grid((unsigned)(ntiles < kNumCU ? ntiles : kNumCU), (unsigned)((dout + G3_BN - 1) / G3_BN));
  if (xvec)
    hipLaunchKernelGGL((gemm3_fwd_kernel<true, false, 2>), grid, dim3(512), G3_LDS, s, x, m, din, x_ld, w, w_ld, trans_w,
                       bias, y, dout, y_ld, act, G3Dact{});
  else
    hipLaunchKernelGGL((gemm3_fwd_kernel<false, false, 2>), grid, dim3(512), G3_LDS, s, x, m, din, x_ld, w, w_ld, trans_w,
                       bias, y, dout, y_ld, act, G3Dact{});
  return check_launch("gemm3_fwd_kernel");
}

// dx = (grad (.) act'(act_out)) @ W^T through the table variant, dpre written on the way (see G3Dact).  Returns -1 when the
// shape / alignment is not one the fused form takes (the caller then runs the activation backward on its own).
// pooled_grad != nullptr: gathered gradient (see G3Dact), `grad` may then be nullptr (nothing handed on besides the read-out)
int launch_gemm3_dx_dact(const float* grad, const float* act_out, float* dpre, long m, int k, long ld, const void* table,
                         float* dx, int n, long dx_ld, int dact, hipStream_t s, const float* pooled_grad, int n_nodes, long pooled_ld) {
  if (pooled_ld <= 0) pooled_ld = k;                    // contiguous [graphs, k]
  const bool ok = (k % 4 == 0) && (ld % 4 == 0) && (!grad || aligned16(grad)) && aligned16(act_out) && aligned16(dpre) && table &&
                  dact != KGCN_ACT_NONE && dpre != grad && (grad || pooled_grad) &&
                  (!pooled_grad || (aligned16(pooled_grad) && n_nodes > 0 && pooled_ld % 4 == 0 && pooled_ld >= k));
  if (!ok) return -1;
  static const char* hknob = dev_knob("KGCN_GEMMH");
  if (!(hknob && !strchr(hknob, 'd'))) {
    const int rc = launch_gemmh_dx_dact(grad, act_out, dpre, m, k, ld, static_cast<const char*>(table) + wtable_bf16_bytes(k, n), dx,
                                        n, dx_ld, dact, s, pooled_grad, n_nodes, pooled_ld, nullptr);
    if (rc >= 0) return rc;
  }
  G3Dact da;
  const float* base = grad ? grad : act_out;             // the staging threads address everything relative to their x row
  da.ydiff = act_out - base;
  da.pdiff = dpre - base;
  da.bc = pooled_grad;
  da.bc_ld = pooled_ld;
  da.bc_n = n_nodes > 0 ? n_nodes : 1;
  da.bc_only = grad ? 0 : 1;
  da.c0 = dact == KGCN_ACT_TANH ? 1.f : 0.f;
  da.c1 = dact == KGCN_ACT_SIGMOID ? 1.f : 0.f;
  da.c2 = -1.f;
  const float* tw = static_cast<const float*>(table);
  if (dact == KGCN_ACT_RELU) g3_table_launch<2>(base, m, k, ld, tw, 0L, 1, nullptr, dx, n, dx_ld, KGCN_ACT_NONE, da, s);
  else g3_table_launch<1>(base, m, k, ld, tw, 0L, 1, nullptr, dx, n, dx_ld, KGCN_ACT_NONE, da, s);
  return check_launch("gemm3_fwd_kernel(dact)");
}

// ------------------------------------------------------------------------------------------------
// Weight gradient:  dW[din, dout] = x^T dy,  dbias = colsum(dy).  The batch rows are the K dimension.
// One workgroup owns a [128 x 256] block of dW and a contiguous range of 32-row chunks; both operands are
// staged like the forward's W chunk -- task (column, 8-row group): 8 strided loads, coalesced across the
// columns of consecutive threads -> split -> fragment tables (A: 4 m-tiles of x columns, B: 8 n-tiles of dy
// columns).  The block's partial dW (and the column sums of dy accumulated by the staging threads) go to the
// workspace; reduce_partials_kernel adds the partials in a fixed order.
// ------------------------------------------------------------------------------------------------
constexpr int G3W_BM = 128;   // din columns per workgroup
template <bool DACT> struct G3RawTT { float a[8], b0[8], b1[8]; };
template <> struct G3RawTT<true> { float a[8], b0[8], b1[8], y0[8], y1[8]; };

// DACT: dy is the gradient of an ACTIVATED layer output; the saved output `yact` (same layout as dy) is staged next to it and
// d pre-activation = dy * act'(yact) is formed when the chunk is split -- it never exists in HBM (used when the layer's
// input needs no gradient, so no dX GEMM produces it on the way: the first layer of a model).
template <bool DACT>
__global__ __launch_bounds__(512, 2) void gemm3_wgrad_kernel(
    const float* __restrict__ x, long x_ld, const float* __restrict__ dy, long dy_ld, long m, int din, int dout,
    float* __restrict__ part_dw, float* __restrict__ part_db, long chunks_per_block,
    const float* __restrict__ yact, int act) {
  using G3RawT = G3RawTT<DACT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  u32x4* lds = reinterpret_cast<u32x4*>(dsm);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, hi = lane >> 5;
  const int wr = wave >> 2, wc = wave & 3;
  const int i0 = blockIdx.y * G3W_BM, j0 = blockIdx.z * G3_BN;
  const long nchunks = (m + G3_BK - 1) / G3_BK;
  const long c_begin = (long)blockIdx.x * chunks_per_block;
  long c_end = c_begin + chunks_per_block;
  if (c_end > nchunks) c_end = nchunks;

  // staging tasks.  x: (column tid % 128, q = tid / 128) -- one 8-row group of one column;
  // dy: columns tid % 256 with q = tid / 256 (rows 0-15) and q + 2 (rows 16-31)
  const int ac = tid & 127, aq = tid >> 7;
  const int bc = tid & 255, bq = tid >> 8;
  const bool aok = i0 + ac < din, bok = j0 + bc < dout;
  const int ca = aok ? i0 + ac : din - 1, cb = bok ? j0 + bc : dout - 1;
  auto issue = [&](G3RawT& r, long chunk) __attribute__((always_inline)) {
    const long row0 = chunk * G3_BK;      // clamped rows / columns, masked at split time
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long ra = row0 + 8 * aq + j, rb0 = row0 + 8 * bq + j, rb1 = row0 + 8 * (bq + 2) + j;
      r.a[j] = x[(ra < m ? ra : m - 1) * x_ld + ca];
      r.b0[j] = dy[(rb0 < m ? rb0 : m - 1) * dy_ld + cb];
      r.b1[j] = dy[(rb1 < m ? rb1 : m - 1) * dy_ld + cb];
      if constexpr (DACT) {
        r.y0[j] = yact[(rb0 < m ? rb0 : m - 1) * dy_ld + cb];
        r.y1[j] = yact[(rb1 < m ? rb1 : m - 1) * dy_ld + cb];
      }
    }
  };
  float colsum = 0.f;               // of dy column bc over this thread's row groups
  Frag3 fa, f0, f1;
  auto split_step = [&](auto sc, const G3RawT& r, long chunk) __attribute__((always_inline)) {
    constexpr int STEP = decltype(sc)::value, J = STEP & 3;
    const long row0 = chunk * G3_BK;
    unsigned q1, q2, q3;
    if constexpr (STEP < 4) {
      const long row = row0 + 8 * aq + 2 * J;
      split_pair((aok && row < m) ? r.a[2 * J] : 0.f, (aok && row + 1 < m) ? r.a[2 * J + 1] : 0.f, q1, q2, q3);
      fa.p1[J] = q1; fa.p2[J] = q2; fa.p3[J] = q3;
    } else if constexpr (STEP < 8) {
      const long row = row0 + 8 * bq + 2 * J;
      float v0 = (bok && row < m) ? r.b0[2 * J] : 0.f, v1 = (bok && row + 1 < m) ? r.b0[2 * J + 1] : 0.f;
      if constexpr (DACT) { v0 *= act_dout(r.y0[2 * J], act); v1 *= act_dout(r.y0[2 * J + 1], act); }
      colsum += v0 + v1;
      split_pair(v0, v1, q1, q2, q3);
      f0.p1[J] = q1; f0.p2[J] = q2; f0.p3[J] = q3;
    } else {
      const long row = row0 + 8 * (bq + 2) + 2 * J;
      float v0 = (bok && row < m) ? r.b1[2 * J] : 0.f, v1 = (bok && row + 1 < m) ? r.b1[2 * J + 1] : 0.f;
      if constexpr (DACT) { v0 *= act_dout(r.y1[2 * J], act); v1 *= act_dout(r.y1[2 * J + 1], act); }
      colsum += v0 + v1;
      split_pair(v0, v1, q1, q2, q3);
      f1.p1[J] = q1; f1.p2[J] = q2; f1.p3[J] = q3;
    }
  };
  auto write_tables = [&](u32x4* xp, u32x4* wp, int which) __attribute__((always_inline)) {
    if (which == 0) g3_write(xp, ac >> 5, aq, ac & 31, fa);
    else if (which == 1) g3_write(wp, bc >> 5, bq, bc & 31, f0);
    else g3_write(wp, bc >> 5, bq + 2, bc & 31, f1);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  if (c_begin < c_end) {
    G3RawT ra, rb;
    issue(ra, c_begin);
    static_for<12>([&](auto sc) __attribute__((always_inline)) { split_step(sc, ra, c_begin); });
    write_tables(lds, lds + G3_XP, 0); write_tables(lds, lds + G3_XP, 1); write_tables(lds, lds + G3_XP, 2);
    issue(rb, c_begin + 1 < c_end ? c_begin + 1 : c_begin);
    __syncthreads();
    int buf = 0;
    long c = c_begin;
    auto step = [&](G3RawT& RS, G3RawT& RL) __attribute__((always_inline)) {
      const bool have1 = c + 1 < c_end;
      const long c1 = have1 ? c + 1 : c;                  // chunk being split (garbage copy of c at the very end)
      const long c2 = c + 2 < c_end ? c + 2 : c1;         // chunk being requested
      const u32x4* xp = lds + buf * (G3_XP + G3_WP);
      const u32x4* wp = xp + G3_XP;
      u32x4* xq = lds + (buf ^ 1) * (G3_XP + G3_WP);
      u32x4* wq = xq + G3_XP;
      const float keep = colsum;
      static_for<2>([&](auto ksc) __attribute__((always_inline)) {
        constexpr int ks = decltype(ksc)::value;
        u32x4 A[2][3], B[2][3];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            A[t2][p] = xp[g3_slot(2 * wr + t2, ks, p, li, hi)];
            B[t2][p] = wp[g3_slot(2 * wc + t2, ks, p, li, hi)];
          }
        static_for<24>([&](auto mc) __attribute__((always_inline)) {
          constexpr int mm = decltype(mc)::value, pr = mm >> 2, tl4 = mm & 3, slot = 24 * ks + mm;
          constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};   // smallest terms first
          acc[tl4 >> 1][tl4 & 1] = mfma_bf16(A[tl4 >> 1][PA[pr]], B[tl4 & 1][PB[pr]], acc[tl4 >> 1][tl4 & 1]);
          if constexpr (slot < 12) split_step(std::integral_constant<int, slot>{}, RS, c1);
          else if constexpr (slot == 12) write_tables(xq, wq, 0);
          else if constexpr (slot == 13) write_tables(xq, wq, 1);
          else if constexpr (slot == 14) write_tables(xq, wq, 2);
          else if constexpr (slot == 16) issue(RL, c2);      // (pinning these like the forward's: 1-2 % slower, measured)
        });
      });
      if (!have1) colsum = keep;                           // the re-split of the last chunk must not count twice
      __syncthreads();
      buf ^= 1;
      ++c;
    };
    for (;;) {
      step(rb, ra);
      if (c >= c_end) break;
      step(ra, rb);
      if (c >= c_end) break;
    }
  }

  // ---- partial dW block of this workgroup ------------------------------------------------------------
  float* pw = part_dw + (long)blockIdx.x * din * dout;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int col = j0 + 64 * wc + 32 * nt + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + 64 * wr + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (row < din && col < dout) pw[(long)row * dout + col] = acc[mt][nt][r];
      }
    }
  // ---- column sums of dy: the two row-group threads of a column meet in LDS ----------------------------
  if (part_db && blockIdx.y == 0) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(dsm);
    red[tid] = colsum;
    __syncthreads();
    if (tid < 256 && j0 + tid < dout) part_db[(long)blockIdx.x * dout + j0 + tid] = red[tid] + red[tid + 256];
  }
}

int launch_gemm3_wgrad(const float* x, long x_ld, const float* dy, long dy_ld, long m, int din, int dout,
                       float* part_dw, float* part_db, int nblocks, hipStream_t s, const float* yact, int act) {
  static thread_local bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm3_wgrad_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm3_wgrad_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    attr_set = true;
  }
  const long nchunks = (m + G3_BK - 1) / G3_BK;
  const long cpb = (nchunks + nblocks - 1) / nblocks;
  const dim3 grid((unsigned)nblocks, (unsigned)((din + G3W_BM - 1) / G3W_BM), (unsigned)((dout + G3_BN - 1) / G3_BN));
  if (yact && act != KGCN_ACT_NONE)
    hipLaunchKernelGGL(gemm3_wgrad_kernel<true>, grid, dim3(512), G3_LDS, s, x, x_ld, dy, dy_ld, m, din, dout, part_dw,
                       part_db, cpb, yact, act);
  else
    hipLaunchKernelGGL(gemm3_wgrad_kernel<false>, grid, dim3(512), G3_LDS, s, x, x_ld, dy, dy_ld, m, din, dout, part_dw,
                       part_db, cpb, nullptr, KGCN_ACT_NONE);
  return check_launch("gemm3_wgrad_kernel");
}

}  // namespace kgcn

#ifdef KGCN_PROBE
extern "C" int kgcn_g3_probe_set(void* buf) {
  long long* p = static_cast<long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(kgcn::g3_probe), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif
