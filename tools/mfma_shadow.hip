// How many filler instructions hide in the shadow of one v_mfma_f32_32x32x2_f32 (one wave per SIMD)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int NL, int NM>   // VALU fillers, LDS-read fillers, MFMAs per iteration
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, long long* cyc) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
  __syncthreads();
  f32x16 acc[4];
  for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
  float a = threadIdx.x, b = 1.0f;
  float ld = 0.f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 15]) : "v"(b), "v"(a));
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        float tmp;
        asm volatile("ds_read_b32 %0, %1" : "=v"(tmp) : "v"((threadIdx.x * 4 + j * 1024) & 16383));
        asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
        ld += 0.f;  // do not consume tmp right away
        v[(j + 7) & 15] = tmp * 0.f + v[(j + 7) & 15];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  long long t1 = __builtin_readcyclecounter();
  float s = ld;
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NV, int NL, int NM>
void run(const char* name, float* out, long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<NV, NL, NM>), dim3(256), dim3(256), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-34s %8.1f cycles per MFMA\n", name, (double)h / iters / NM);
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  run<0, 0, 4>("MFMA only", out, cyc);
  run<2, 0, 4>("MFMA + 2 v_fma", out, cyc);
  run<4, 0, 4>("MFMA + 4 v_fma", out, cyc);
  run<8, 0, 4>("MFMA + 8 v_fma", out, cyc);
  run<12, 0, 4>("MFMA + 12 v_fma", out, cyc);
  run<16, 0, 4>("MFMA + 16 v_fma", out, cyc);
  run<24, 0, 4>("MFMA + 24 v_fma", out, cyc);
  run<0, 1, 4>("MFMA + 1 ds_read_b32(+1 valu)", out, cyc);
  run<0, 2, 4>("MFMA + 2 ds_read_b32(+2 valu)", out, cyc);
  run<0, 4, 4>("MFMA + 4 ds_read_b32(+4 valu)", out, cyc);
  run<4, 2, 4>("MFMA + 4 v_fma + 2 ds_read", out, cyc);
  return 0;
}
