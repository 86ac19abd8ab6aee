// Does the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, 8 passes) run concurrently with VALU work on gfx950,
// unlike v_mfma_f32_32x32x2_f32 (tools/mfma_shadow.hip)?  WAVES = waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NV, int NM, bool BF>
__global__ void k(float* out, int iters, long long* cyc) {
  f32x16 acc[4];
  for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
  float a = threadIdx.x, b = 1.0f;
  bf16x8 pa, pb;
  for (int i = 0; i < 8; ++i) { pa[i] = (__bf16)(float)(threadIdx.x + i); pb[i] = (__bf16)1.0f; }
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if constexpr (BF) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, acc[m & 3], 0, 0, 0);
      else acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 15]) : "v"(b), "v"(a));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}
template <int NV, int NM, bool BF>
void run(const char* name, int waves, float* out, long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<NV, NM, BF>), dim3(256), dim3(256 * waves), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  long long hh[32]; hipMemcpy(hh, cyc, 8 * 2 * 4 * waves, hipMemcpyDeviceToHost);
  long long lo = hh[0], hi = hh[1];
  for (int i = 0; i < 4 * waves; ++i) { if (hh[2 * i] < lo) lo = hh[2 * i]; if (hh[2 * i + 1] > hi) hi = hh[2 * i + 1]; }
  long long h = hi - lo;  // first start .. last end over all waves of workgroup 0
  printf("%-28s waves/SIMD %d: %8.1f cycles per MFMA per wave (%6.1f per SIMD)\n", name, waves,
         (double)h / iters / NM, (double)h / iters / NM / waves);
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8 * 64);
  for (int w = 1; w <= 2; ++w) {
    run<0, 4, false>("f32 MFMA only", w, out, cyc);
    run<8, 4, false>("f32 MFMA + 8 v_fma", w, out, cyc);
    run<16, 4, false>("f32 MFMA + 16 v_fma", w, out, cyc);
    run<0, 4, true>("bf16 MFMA only", w, out, cyc);
    run<2, 4, true>("bf16 MFMA + 2 v_fma", w, out, cyc);
    run<4, 4, true>("bf16 MFMA + 4 v_fma", w, out, cyc);
    run<8, 4, true>("bf16 MFMA + 8 v_fma", w, out, cyc);
    run<16, 4, true>("bf16 MFMA + 16 v_fma", w, out, cyc);
    run<32, 4, true>("bf16 MFMA + 32 v_fma", w, out, cyc);
  }
  return 0;
}
