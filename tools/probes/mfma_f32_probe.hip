// development probe: cycles per v_mfma_f32_32x32x2_f32 (s_memtime) for 1 / 2 accumulator chains, 1 or 4 waves per workgroup,
// with and without the ds_read_b64 operand loads of stack_tile.hip in the loop.   hipcc --offload-arch=gfx950 -O3 -o probe ...
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int CHAINS, bool LDS>
__global__ void probe(float* out, unsigned long long* cyc, int iters) {
  __shared__ float tile[65 * 66 * 2];
  for (int i = threadIdx.x; i < 65 * 66 * 2; i += blockDim.x) tile[i] = 0.001f * (i & 63);
  __syncthreads();
  const int li = threadIdx.x & 31, lh = (threadIdx.x >> 5) & 1;
  const float* ap = tile + li * 66 + 2 * lh;
  const float* bp = tile + 65 * 66 + li * 66 + 2 * lh;
  f32x16 acc = {0}, acc2 = {0};
  float a = threadIdx.x * 0.5f, b = 1.0f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    f32x2 a0 = {a, a}, a1 = {a, a}, b0 = {b, b}, b1 = {b, b};
    if (LDS) {
      const int k = (it & 7) * 8;
      a0 = *reinterpret_cast<const f32x2*>(ap + k); a1 = *reinterpret_cast<const f32x2*>(ap + k + 4);
      b0 = *reinterpret_cast<const f32x2*>(bp + k); b1 = *reinterpret_cast<const f32x2*>(bp + k + 4);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[0], b0[0], acc, 0, 0, 0);
    if (CHAINS == 2) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[1], b0[1], acc2, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[1], b0[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0], b1[0], acc, 0, 0, 0);
    if (CHAINS == 2) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[1], b1[1], acc2, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[1], b1[1], acc, 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i] + acc2[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int CHAINS, bool LDS>
void run(const char* name, int threads, int blocks) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4 * threads * blocks); hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<CHAINS, LDS><<<blocks, threads>>>(out, cyc, iters);
  hipEventRecord(e0);
  probe<CHAINS, LDS><<<blocks, threads>>>(out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-34s threads %3d blocks %3d: %7.1f ticks / MFMA, %6.1f ns / MFMA (event)\n", name, threads, blocks, (double)c / (4.0 * iters),
         ms * 1e6 / (4.0 * iters));
  hipFree(out); hipFree(cyc);
}

int main() {
  run<1, false>("1 chain, register operands", 64, 1);
  run<2, false>("2 chains, register operands", 64, 1);
  run<2, false>("2 chains, register operands", 256, 1);
  run<2, false>("2 chains, register operands", 256, 256);
  run<2, true>("2 chains, ds_read_b64 operands", 64, 1);
  run<2, true>("2 chains, ds_read_b64 operands", 256, 1);
  run<2, true>("2 chains, ds_read_b64 operands", 256, 256);
  run<1, true>("1 chain, ds_read_b64 operands", 256, 256);
  return 0;
}
