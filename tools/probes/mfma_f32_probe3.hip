// development probe 3: what one vector-memory instruction costs next to v_mfma_f32_32x32x2_f32.  Each trip issues 24 MFMAs (3
// accumulator chains) and LOADS dword loads per lane out of an 8 KB window (L1 resident: no HBM traffic), with 1, 2 or 4 waves
// per SIMD; the same with ds_read_b32 out of LDS.  Prints ns per MFMA of the SIMD (HIP events over 20 launches).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int LOADS, bool LDS>
__global__ void probe(const float* __restrict__ src, float* out, int iters) {
  __shared__ float tile[2048];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) tile[i] = 0.001f * (i & 63);
  __syncthreads();
  f32x16 acc[3];
  for (int c = 0; c < 3; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  const float* p = LDS ? tile + (threadIdx.x & 63) : src + (threadIdx.x & 63) + (long)blockIdx.x * 2048;
  float a = threadIdx.x * 0.5f, b = 1.0f;
  for (int it = 0; it < iters; ++it) {
    float v[LOADS > 0 ? LOADS : 1];
#pragma unroll
    for (int u = 0; u < LOADS; ++u) v[u] = p[((it * LOADS + u) * 64) & 1983];
    float extra = 0.f;
#pragma unroll
    for (int u = 0; u < LOADS; ++u) extra += v[u];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + u + extra, b + c, acc[c], 0, 0, 0);
  }
  float s = 0;
  for (int c = 0; c < 3; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int LOADS, bool LDS>
void run(int threads, const float* src) {
  float* out;
  const int blocks = 256, iters = 1000;
  hipMalloc(&out, 4 * threads * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<LOADS, LDS><<<blocks, threads>>>(src, out, iters);
  hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) probe<LOADS, LDS><<<blocks, threads>>>(src, out, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = 24.0 * iters * (threads / 256.0);
  printf("%-6s %2d loads per 24 MFMAs, %d wave(s) per SIMD: %6.1f ns per MFMA of the SIMD\n", LDS ? "LDS" : "global", LOADS,
         threads / 256, ms / 20 * 1e6 / mfma_per_simd);
  hipFree(out);
}

template <bool LDS> void sweep(const float* src) {
  for (int threads : {256, 512, 1024}) {
    run<0, LDS>(threads, src); run<4, LDS>(threads, src); run<8, LDS>(threads, src); run<16, LDS>(threads, src);
    run<32, LDS>(threads, src);
  }
}

int main() {
  float* src; hipMalloc(&src, 256L * 2048 * 4 + 4096); hipMemset(src, 0, 256L * 2048 * 4 + 4096);
  sweep<false>(src);
  sweep<true>(src);
  return 0;
}
