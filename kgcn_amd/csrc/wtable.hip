// W (or W^T) of a wide dense layer as a bf16 fragment table for the bf16-split GEMM of gemm3.hip:
// the exact 3-way split (kgcn_common.h) is done ONCE per call by a small kernel instead of by every workgroup for
// every row tile; the table is written in MFMA B-operand order -- one 1 KiB lane-linear block per
// (16-wide k-step ks, 32-column tile nt, piece p):  entry [((ks * ntiles + nt) * 3 + p) * 64 + lane], 16 bytes each --
// and stays L2 resident (0.4-1.5 MB for the layers of example_model/model_multitask.py:51-57 and sparse.py:30).
// Rows k >= din and columns n >= dout are zero.
//
// (Round 2 also tried two GEMMs that share nothing between waves on this table -- one wave per SIMD with a [128 x 64]
// block in registers, and two waves per SIMD with [64 x 64] blocks, no LDS, no barrier.  Both lost to gemm3 + table because
// every wave then re-splits its x rows: measurements and ablations in profiles/r02_gemm_experiments.txt.)
#include "kgcn_common.h"

namespace kgcn {

constexpr int WT_BN = 64;      // columns are padded to whole 64-column blocks (two 32-column tiles)

__host__ __device__ inline long wtable_entries(int din, int dout) {
  const long ksteps = (din + 15) / 16, ntiles = ((dout + WT_BN - 1) / WT_BN) * (WT_BN / 32);
  return ksteps * ntiles * 3 * 64;
}

// one thread per (k-step, column, lane half): the 8 k-values of its lane
__global__ __launch_bounds__(256) void wtable_split_kernel(const float* __restrict__ w, long w_ld, int trans_w, int din,
                                                           int dout, u32x4* __restrict__ table) {
  const int ntiles = ((dout + WT_BN - 1) / WT_BN) * (WT_BN / 32);
  const long total = (long)((din + 15) / 16) * ntiles * 64;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
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

int64_t wtable_bytes(int din, int dout) { return wtable_entries(din, dout) * 16; }

void launch_wtable_split(const float* w, long w_ld, int trans_w, int din, int dout, void* workspace, hipStream_t s) {
  const long threads = wtable_entries(din, dout) / 3;
  hipLaunchKernelGGL(wtable_split_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, w, w_ld, trans_w, din,
                     dout, static_cast<u32x4*>(workspace));
}

}  // namespace kgcn
