// Achievable HBM bandwidth on this box: float4 grid-stride copy / read-only / write-only kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_k(const f4* __restrict__ a, f4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void read_k(const f4* __restrict__ a, float* __restrict__ out, size_t n) {
  f4 s = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += a[i];
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void write_k(f4* __restrict__ b, size_t n) {
  f4 v = {1, 2, 3, 4};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = v;
}
int main() {
  const size_t bytes = (size_t)1 << 30;  // 1 GiB per buffer (beyond the 256 MiB Infinity Cache)
  const size_t n = bytes / 16;
  f4 *a, *b; float* o;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
  hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {2048, 4096, 8192, 16384, 65536}) {
    float ms[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) {
      for (int w = 0; w < 3; ++w) {
        if (k == 0) hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, 0, a, b, n);
        if (k == 1) hipLaunchKernelGGL(read_k, dim3(grid), dim3(256), 0, 0, a, o, n);
        if (k == 2) hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, 0, b, n);
      }
      hipEventRecord(e0);
      for (int r = 0; r < 20; ++r) {
        if (k == 0) hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, 0, a, b, n);
        if (k == 1) hipLaunchKernelGGL(read_k, dim3(grid), dim3(256), 0, 0, a, o, n);
        if (k == 2) hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, 0, b, n);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[k], e0, e1); ms[k] /= 20;
    }
    printf("grid %6d: copy %.1f GB/s (r+w)  read %.1f GB/s  write %.1f GB/s\n", grid,
           2.0 * bytes / ms[0] / 1e6, 1.0 * bytes / ms[1] / 1e6, 1.0 * bytes / ms[2] / 1e6);
  }
  return 0;
}
