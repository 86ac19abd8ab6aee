// Memory skeleton of the fused GraphConv backward (development tool, GPU box only): persistent waves that move
// exactly the HBM bytes of graphconv_bwd_full_kernel -- per graph two [32 x 64] fp32 tiles in (x, g), a CSR
// slice in, one [32 x 64] tile out -- with NO contraction / aggregation work, so what it reaches is the
// ceiling the memory system gives this access pattern (2:1 read:write, 8 KiB tiles, T graphs strided over the
// waves).  The store pattern is the variable:
//   0  32 x global_store_dword   (MFMA C layout of dX = dFW W^T: lane = column, 2 x 128 B per instruction)
//   1   8 x global_store_dwordx4 (whole rows: 1 KiB contiguous per instruction)
//   2   8 x global_store_dwordx4 (C layout of dX^T = W dFW^T: lane = row, 32 B segments at 256 B stride)
//   3   as 1, nontemporal
//   4   no stores (read side alone)
// usage: bwd_skeleton [graphs] [waves per CU: 4 or 8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void skel(const float* __restrict__ x, const float* __restrict__ g,
                                            const f4* __restrict__ cv, float* __restrict__ dx, int T) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = gridDim.x * (blockDim.x >> 6);
  const int t0 = blockIdx.x * (blockDim.x >> 6) + wave;
  if (t0 >= T) return;
  const int cnt = (T - 1 - t0) / nw + 1;
  const int li = lane & 31, hi = lane >> 5;
  f4 xr[DEPTH][8], gr[DEPTH][8], cr[DEPTH][2];
  auto issue = [&](int k, int slot) {
    const int t = t0 + (k < cnt ? k : cnt - 1) * nw;
    const f4* xs = reinterpret_cast<const f4*>(x + (long)t * 2048);
    const f4* gs = reinterpret_cast<const f4*>(g + (long)t * 2048);
#pragma unroll
    for (int q = 0; q < 8; ++q) { xr[slot][q] = xs[lane + 64 * q]; gr[slot][q] = gs[lane + 64 * q]; }
    cr[slot][0] = cv[(long)t * 80 + lane];
    cr[slot][1] = lane < 16 ? cv[(long)t * 80 + 64 + lane] : f4{0, 0, 0, 0};
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(d, d);
  for (int i = 0; i < cnt; i += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (i + d >= cnt) break;
      const int t = t0 + (i + d) * nw;
      f4 o[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = xr[d][q] * gr[d][q] + cr[d][q & 1];
      issue(i + d + DEPTH, d);
      float* dst = dx + (long)t * 2048;
      if constexpr (MODE == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
          dst[row * 64 + li] = o[r >> 2][r & 3];
          dst[row * 64 + 32 + li] = o[4 + (r >> 2)][r & 3];
        }
      } else if constexpr (MODE == 1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) reinterpret_cast<f4*>(dst)[lane + 64 * q] = o[q];
      } else if constexpr (MODE == 2) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
          *reinterpret_cast<f4*>(dst + li * 64 + (q >> 2) * 32 + 8 * (q & 3) + 4 * hi) = o[q];
      } else if constexpr (MODE == 3) {
#pragma unroll
        for (int q = 0; q < 8; ++q) __builtin_nontemporal_store(o[q], reinterpret_cast<f4*>(dst) + lane + 64 * q);
      } else {
        f4 s = o[0] + o[1] + o[2] + o[3] + o[4] + o[5] + o[6] + o[7];
        if (s[0] + s[1] + s[2] + s[3] == 123.456f) dst[lane] = 1.f;
      }
    }
  }
}

template <int MODE, int DEPTH>
static float run(const float* x, const float* g, const f4* cv, float* dx, int T, int wpc) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * wpc / 4;
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((skel<MODE, DEPTH>), dim3(blocks), dim3(256), 0, 0, x, g, cv, dx, T);
  hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((skel<MODE, DEPTH>), dim3(blocks), dim3(256), 0, 0, x, g, cv, dx, T);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 20;
}

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 100000;
  float *x, *g, *dx; f4* cv;
  const size_t tb = (size_t)T * 8192;
  hipMalloc(&x, tb); hipMalloc(&g, tb); hipMalloc(&dx, tb); hipMalloc(&cv, (size_t)T * 1280);
  hipMemset(x, 0x11, tb); hipMemset(g, 0x22, tb); hipMemset(cv, 0, (size_t)T * 1280);
  const double rd = (double)T * (8192 * 2 + 1280), wr = (double)T * 8192;
  const char* names[5] = {"32 x dword (C layout)", "8 x dwordx4 rows (1 KiB)", "8 x dwordx4, 32 B segments",
                          "8 x dwordx4 rows, nontemporal", "no stores"};
  for (int wpc : {4, 8}) {
    printf("waves per CU = %d, graphs = %d\n", wpc, T);
    float ms[5] = {run<0, 2>(x, g, cv, dx, T, wpc), run<1, 2>(x, g, cv, dx, T, wpc), run<2, 2>(x, g, cv, dx, T, wpc),
                   run<3, 2>(x, g, cv, dx, T, wpc), run<4, 2>(x, g, cv, dx, T, wpc)};
    for (int m = 0; m < 5; ++m)
      printf("  depth 2  %-34s %.3f ms  %.0f GB/s\n", names[m], ms[m], (rd + (m == 4 ? 0 : wr)) / ms[m] / 1e6);
    float m1[2] = {run<1, 1>(x, g, cv, dx, T, wpc), run<0, 1>(x, g, cv, dx, T, wpc)};
    printf("  depth 1  %-34s %.3f ms  %.0f GB/s\n", names[1], m1[0], (rd + wr) / m1[0] / 1e6);
    printf("  depth 1  %-34s %.3f ms  %.0f GB/s\n", names[0], m1[1], (rd + wr) / m1[1] / 1e6);
  }
  return 0;
}
