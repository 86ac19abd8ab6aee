// Do big by-value kernel arguments cost a copy kernel per hipGraph replay?  (cfg1's step shows 4 x __amd_rocclr_copyBuffer, 10 % of
// its 113 us, and the library issues no memcpy: the candidates are the job tables it passes BY VALUE to its multi-tensor kernels.)
// One captured graph per argument size with 8 launches of a trivial kernel; run under `rocprofv3 --kernel-trace --stats` and
// count the copyBuffer dispatches per size (the marker kernel `size_marker<N>` brackets each size).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Blob { int v[N]; };
template <int N> __global__ void take(Blob<N> b, int* out) { if (threadIdx.x == 0 && b.v[N - 1] == 12345) out[0] = 1; }
template <int N> __global__ void size_marker(int* out) { if (out == nullptr) printf("x"); }
template <int N> void run(hipStream_t s, int* out) {
  Blob<N> b{};
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(take<N>, dim3(1), dim3(64), 0, s, b, out);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipLaunchKernelGGL(size_marker<N>, dim3(1), dim3(1), 0, s, out);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) hipGraphLaunch(ge, s);
  hipEventRecord(e0, s);
  for (int r = 0; r < 20; ++r) hipGraphLaunch(ge, s);
  hipEventRecord(e1, s);
  hipStreamSynchronize(s);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("kernarg %5d bytes: %.2f us per replay of 8 launches\n", (int)sizeof(Blob<N>) + 8, ms / 20 * 1e3);
}
int main() {
  hipStream_t s; hipStreamCreate(&s);
  int* out; hipMalloc(&out, 4);
  run<4>(s, out); run<16>(s, out); run<56>(s, out); run<64>(s, out); run<120>(s, out); run<128>(s, out); run<256>(s, out); run<512>(s, out); run<1000>(s, out);
  return 0;
}
