// development probe 2: why do real kernels see ~150 cycles per v_mfma_f32_32x32x2_f32?  Variants: waves per SIMD, number of
// accumulator chains, operands from global memory (fresh registers per MFMA), inline asm volatile barrier between loads.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS, bool GLOBAL, bool RANDOM = false, int WINDOW = 60 * 1024>
__global__ void probe(const float* __restrict__ src, float* out, unsigned long long* cyc, int iters) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  const float* p = src + threadIdx.x + (long)blockIdx.x * 64 * 1024;
  float a = threadIdx.x * 0.5f, b = 1.0f;
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    float av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      av[u] = GLOBAL ? p[(it * 16 + 2 * u) * 64 % WINDOW] : a + u;
      bv[u] = GLOBAL ? p[(it * 16 + 2 * u + 1) * 64 % WINDOW] : b + u;
      if (!GLOBAL && !RANDOM && CHAINS == 4) {         // variant: every operand freshly written by ONE cheap VALU op
        av[u] = a = a * 1.0001f; bv[u] = b = b * 0.9999f;
      }
      if (RANDOM) {                                    // full-entropy significands, magnitudes around 1
        h = h * 1664525u + 1013904223u; av[u] = __uint_as_float(0x3f000000u | (h >> 9)) - 0.75f;
        h = h * 1664525u + 1013904223u; bv[u] = __uint_as_float(0x3f000000u | (h >> 9)) - 0.75f;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[(u + c) & 7], acc[c], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int CHAINS, bool GLOBAL, bool RANDOM = false, int WINDOW = 60 * 1024>
void run(const char* name, int threads, int blocks, const float* src) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4 * threads * blocks); hipMalloc(&cyc, 8);
  const int iters = 1000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<CHAINS, GLOBAL, RANDOM, WINDOW><<<blocks, threads>>>(src, out, cyc, iters);
  hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) probe<CHAINS, GLOBAL, RANDOM, WINDOW><<<blocks, threads>>>(src, out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double per_wave = 8.0 * CHAINS * iters, waves_per_simd = threads / 256.0;
  printf("%-40s threads %3d blocks %3d: %6.1f ticks / MFMA / wave = %6.1f ticks per MFMA of the SIMD, %6.1f ns (events, 20 launches)\n", name,
         threads, blocks, (double)c / per_wave, (double)c / per_wave / waves_per_simd, ms / 20 * 1e6 / per_wave / waves_per_simd);
  hipFree(out); hipFree(cyc);
}

int main() {
  float* src; hipMalloc(&src, 256L * 64 * 1024 * 4 + (1 << 20)); hipMemset(src, 0, 256L * 64 * 1024 * 4 + (1 << 20));
  run<2, false>("2 chains, registers", 256, 256, src);
  run<3, false>("3 chains, registers", 256, 256, src);
  run<3, false>("3 chains, registers, 2 waves/SIMD", 512, 256, src);
  run<3, false, true>("3 chains, registers, RANDOM data", 256, 256, src);
  run<3, false, true>("3 chains, registers, RANDOM, 2 waves/SIMD", 512, 256, src);
  run<4, false>("4 chains, operands rewritten by 1 VALU op", 256, 256, src);
  run<4, false>("4 chains, rewritten, 2 waves/SIMD", 512, 256, src);
  run<3, true, false, 2048>("3 chains, global operands, 8 KB window (L1)", 256, 256, src);
  run<3, true, false, 2048>("3 chains, global, 8 KB window, 2 waves/SIMD", 512, 256, src);
  run<3, true, false, 2048>("3 chains, global, 8 KB window, 4 waves/SIMD", 1024, 256, src);
  run<3, true>("3 chains, global operands", 256, 256, src);
  run<3, true>("3 chains, global operands, 2 waves/SIMD", 512, 256, src);
  run<1, true>("1 chain, global operands, 2 waves/SIMD", 512, 256, src);
  return 0;
}
