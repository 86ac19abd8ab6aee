// W (or W^T) of a wide dense layer as a bf16 fragment table for the bf16-split GEMM of gemm3.hip:
// the exact 3-way split (kgcn_common.h) is done ONCE per call by a small kernel instead of by every workgroup for
// every row tile; the table is written in MFMA B-operand order -- one 1 KiB lane-linear block per
// (16-wide k-step ks, 32-column tile nt, piece p):  entry [((ks * ntiles + nt) * 3 + p) * 64 + lane], 16 bytes each --
// and stays L2 resident (0.4-1.5 MB for the layers of example_model/model_multitask.py:51-57 and sparse.py:30).
// Rows k >= din and columns n >= dout are zero.
// Behind the bf16 section lies the f16 section of the same operand for the two-piece GEMMs of gemmh.hip (layout: gemmh.h):
// W' = W * 2^kc with kc per column (its maximum lands in [2^14, 2^15)), high and low f16 pieces, then the kc themselves.
// Both sections are written by the same launch; every consumer reads the one it was built for.
//
// (Round 2 also tried two GEMMs that share nothing between waves on this table -- one wave per SIMD with a [128 x 64]
// block in registers, and two waves per SIMD with [64 x 64] blocks, no LDS, no barrier.  Both lost to gemm3 + table because
// every wave then re-splits its x rows: measurements and ablations in profiles/r02_gemm_experiments.txt.)
#include "gemmh.h"

namespace kgcn {

constexpr int WT_BN = 64;      // columns are padded to whole 64-column blocks (two 32-column tiles)

__host__ __device__ inline long wtable_entries(int din, int dout) {
  const long ksteps = (din + 15) / 16, ntiles = ((dout + WT_BN - 1) / WT_BN) * (WT_BN / 32);
  return ksteps * ntiles * 3 * 64;
}


// f16 section: FOUR workgroups (256 threads) per 32-column tile, each splitting a quarter of the tile's k-steps (one per wave and
// trip).  Every one of them computes the column maxima over all k itself -- the 32 KB column block comes from L2, and the launch
// is latency: with one workgroup per tile (four sequential k-step trips behind the maxima) it took 17 us of a training step.
constexpr int WTH_SPLIT = 4;
// kw / extra (stacked operand, trans_w == 0): rows k < kw come from w, row kw from extra[n], rows beyond are zero -- the
// operand [W; bias; 0] of an aggregate-first GraphConv (kgcn_amd/layers.py) without a concatenation pass; extra == nullptr: plain
__device__ __forceinline__ void wtableh_tile(const float* __restrict__ w, long w_ld, int trans_w, int din, int dout, int job,
                                             unsigned char* __restrict__ sec, int kw = 0, const float* __restrict__ extra = nullptr) {
  __shared__ float red[8][32];
  const int nt = job / WTH_SPLIT, part_ks = job % WTH_SPLIT;
  const int tid = threadIdx.x, li = tid & 31, part = tid >> 5;
  const int nt32 = gh_nt32(dout), kse = gh_kse(din);
  const int n = 32 * nt + li;
  auto at = [&](int k) __attribute__((always_inline)) {
    if (extra) return (n < dout && k <= kw) ? (k < kw ? w[(long)k * w_ld + n] : extra[n]) : 0.f;
    return (n < dout && k < din) ? (trans_w ? w[(long)n * w_ld + k] : w[(long)k * w_ld + n]) : 0.f;
  };
  // column maxima: 8 independent loads in flight per thread
  float mx = 0.f;
  for (int k0 = part; k0 < din; k0 += 64) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = at(k0 + 8 * u);            // at() returns 0 beyond din
#pragma unroll
    for (int u = 0; u < 8; ++u) mx = fmaxf(mx, fabsf(v[u]));
  }
  red[part][li] = mx;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 8; ++q) mx = fmaxf(mx, red[q][li]);
  const int kc = scale_exp(mx);
  u32x4* tab = reinterpret_cast<u32x4*>(sec);
  int* kctab = reinterpret_cast<int*>(sec + gh_table_blocks(din, dout) * 1024);
  if (tid < 32 && part_ks == 0) kctab[n] = kc;
  const int lane = tid & 63, hi = lane >> 5, wave = tid >> 6;
  for (int ks = 4 * part_ks + wave; ks < kse; ks += 4 * WTH_SPLIT) {
    u32x4 h, l;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = 16 * ks + 8 * hi + 2 * q;
      unsigned hh, ll;
      splith_pair(__builtin_ldexpf(at(k), kc), __builtin_ldexpf(at(k + 1), kc), hh, ll);
      h[q] = hh; l[q] = ll;
    }
    u32x4* d = tab + ((long)(ks * nt32 + nt) * 2) * 64 + lane;
    d[0] = h; d[64] = l;
  }
}

// one thread per (k-step, column, lane half): the 8 k-values of its lane; workgroups nb3 .. of the grid write the f16 section
__global__ __launch_bounds__(256) void wtable_split_kernel(const float* __restrict__ w, long w_ld, int trans_w, int din,
                                                           int dout, u32x4* __restrict__ table, int nb3) {
  if ((int)blockIdx.x >= nb3) {
    wtableh_tile(w, w_ld, trans_w, din, dout, (int)blockIdx.x - nb3,
                 reinterpret_cast<unsigned char*>(table) + wtable_entries(din, dout) * 16);
    return;
  }
  const int ntiles = ((dout + WT_BN - 1) / WT_BN) * (WT_BN / 32);
  const long total = (long)((din + 15) / 16) * ntiles * 64;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)nb3 * 256) {
    const int lane = (int)(i & 63);
    const long blk = i >> 6;
    const int nt = (int)(blk % ntiles), ks = (int)(blk / ntiles);
    const int n = 32 * nt + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + j;
      float x = 0.f;
      if (n < dout && k < din) x = trans_w ? w[(long)n * w_ld + k] : w[(long)k * w_ld + n];
      v[j] = x;
    }
    Frag3 f;
    split8(v, f);
    u32x4* d = table + ((long)(ks * ntiles + nt) * 3) * 64 + lane;
    d[0] = f.p1; d[64] = f.p2; d[128] = f.p3;
  }
}

// the same for up to KGCN_WTABLE_MAX_JOBS operands in one launch (grid.y = operand): a training step splits every weight
// operand of its wide layers -- W for the forward, W^T for d input -- once, at its start, instead of 5 us launches in front
// of every GEMM (7-9 per step of the BASELINE models)
struct WtJobs {
  const float* w[KGCN_WTABLE_MAX_JOBS];
  u32x4* table[KGCN_WTABLE_MAX_JOBS];
  long w_ld[KGCN_WTABLE_MAX_JOBS];
  int trans[KGCN_WTABLE_MAX_JOBS], din[KGCN_WTABLE_MAX_JOBS], dout[KGCN_WTABLE_MAX_JOBS];
  const float* extra[KGCN_WTABLE_MAX_JOBS];
  int kw[KGCN_WTABLE_MAX_JOBS];
};

__global__ __launch_bounds__(256) void wtable_split_multi_kernel(WtJobs jb, int nb3) {
  const int q = blockIdx.y;
  const float* __restrict__ w = jb.w[q];
  const long w_ld = jb.w_ld[q];
  const int trans_w = jb.trans[q], din = jb.din[q], dout = jb.dout[q];
  u32x4* __restrict__ table = jb.table[q];
  if ((int)blockIdx.x >= nb3) {                 // the f16 section: one workgroup per 32-column tile of this job
    const int nt = (int)blockIdx.x - nb3;
    if (nt < WTH_SPLIT * gh_nt32(dout))
      wtableh_tile(w, w_ld, trans_w, din, dout, nt, reinterpret_cast<unsigned char*>(table) + wtable_entries(din, dout) * 16, jb.kw[q],
                   jb.extra[q]);
    return;
  }
  const int ntiles = ((dout + WT_BN - 1) / WT_BN) * (WT_BN / 32);
  const long total = (long)((din + 15) / 16) * ntiles * 64;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)nb3 * 256) {
    const int lane = (int)(i & 63);
    const long blk = i >> 6;
    const int nt = (int)(blk % ntiles), ks = (int)(blk / ntiles);
    const int n = 32 * nt + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + j;
      float x = 0.f;
      if (jb.extra[q]) { if (n < dout && k <= jb.kw[q]) x = k < jb.kw[q] ? w[(long)k * w_ld + n] : jb.extra[q][n]; }
      else if (n < dout && k < din) x = trans_w ? w[(long)n * w_ld + k] : w[(long)k * w_ld + n];
      v[j] = x;
    }
    Frag3 f;
    split8(v, f);
    u32x4* d = table + ((long)(ks * ntiles + nt) * 3) * 64 + lane;
    d[0] = f.p1; d[64] = f.p2; d[128] = f.p3;
  }
}

int64_t wtable_bf16_bytes(int din, int dout) { return wtable_entries(din, dout) * 16; }
int64_t wtable_bytes(int din, int dout) { return wtable_bf16_bytes(din, dout) + gh_table_bytes(din, dout); }

void launch_wtable_split(const float* w, long w_ld, int trans_w, int din, int dout, void* workspace, hipStream_t s) {
  const long threads = wtable_entries(din, dout) / 3;
  const int nb3 = (int)((threads + 255) / 256);
  hipLaunchKernelGGL(wtable_split_kernel, dim3((unsigned)(nb3 + WTH_SPLIT * gh_nt32(dout))), dim3(256), 0, s, w, w_ld, trans_w, din,
                     dout, static_cast<u32x4*>(workspace), nb3);
}

}  // namespace kgcn

using namespace kgcn;

extern "C" int kgcn_wtable_split_multi(const kgcn_wtable_job* jobs, int32_t num_jobs, void* stream) {
  if (num_jobs < 0 || (num_jobs > 0 && !jobs)) return fail("kgcn_wtable_split_multi: bad job list");
  for (int base = 0; base < num_jobs; base += KGCN_WTABLE_MAX_JOBS) {
    const int n = num_jobs - base < KGCN_WTABLE_MAX_JOBS ? num_jobs - base : KGCN_WTABLE_MAX_JOBS;
    WtJobs jb{};
    long most = 0;
    int most_nt = 0;
    for (int q = 0; q < n; ++q) {
      const kgcn_wtable_job& j = jobs[base + q];
      if (!j.w || !j.table || j.k <= 0 || j.n <= 0) return fail("kgcn_wtable_split_multi: job %d: bad operand", base + q);
      if (j.w_ld < (j.trans_w ? j.k : j.n)) return fail("kgcn_wtable_split_multi: job %d: w_ld too small", base + q);
      jb.w[q] = j.w; jb.table[q] = static_cast<u32x4*>(j.table); jb.w_ld[q] = (long)j.w_ld;
      jb.trans[q] = j.trans_w; jb.din[q] = j.k; jb.dout[q] = j.n;
      jb.extra[q] = j.extra_row; jb.kw[q] = j.k_w;
      if (j.extra_row && (j.trans_w || j.k_w < 0 || j.k_w >= j.k))
        return fail("kgcn_wtable_split_multi: job %d: a stacked operand needs trans_w = 0 and 0 <= k_w < k", base + q);
      const long threads = wtable_entries(j.k, j.n) / 3;
      if (threads > most) most = threads;
      if (gh_nt32(j.n) > most_nt) most_nt = gh_nt32(j.n);
    }
    const int nb3 = (int)((most + 255) / 256);
    hipLaunchKernelGGL(wtable_split_multi_kernel, dim3((unsigned)(nb3 + WTH_SPLIT * most_nt), n), dim3(256), 0, as_stream(stream), jb, nb3);
    if (int rc = check_launch("wtable_split_multi_kernel")) return rc;
  }
  return 0;
}
