// Which address-to-time pattern streams fastest?  A 2:1 read:write stream (two 1 GiB inputs, one 1 GiB output: the traffic mix of the
// fused GraphConv backward) and a plain copy, float4 per lane, under different assignments of 4 KiB chunks to workgroups.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_pattern.bin hbm_pattern.hip && ./hbm_pattern.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
// MODE 0: grid-stride (chunk c = i * G + b)                       -- a window of G chunks sweeps through memory
// MODE 1: contiguous shard per workgroup (c = b * per + i)        -- G streams spread over the whole buffer
// MODE 2: XCD-contiguous: XCD x = b % 8 owns an eighth of the buffer, its workgroups sweep it grid-stride
// MODE 3: like 0, two chunks per iteration in flight (8 KiB per workgroup step)
template <int MODE, int NIN>
__global__ __launch_bounds__(256) void stream_k(const f4* __restrict__ a, const f4* __restrict__ a2, f4* __restrict__ o, long nchunks) {
  const long G = gridDim.x, b = blockIdx.x;
  const int t = threadIdx.x;
  if (MODE == 0) {
    for (long c = b; c < nchunks; c += G) {
      f4 v = a[c * 256 + t];
      if (NIN == 2) v += a2[c * 256 + t];
      o[c * 256 + t] = v;
    }
  } else if (MODE == 1) {
    const long per = (nchunks + G - 1) / G;
    for (long c = b * per; c < (b + 1) * per && c < nchunks; ++c) {
      f4 v = a[c * 256 + t];
      if (NIN == 2) v += a2[c * 256 + t];
      o[c * 256 + t] = v;
    }
  } else if (MODE == 2) {
    const long x = b % 8, j = b / 8, gx = G / 8, per = nchunks / 8;
    for (long c = j; c < per; c += gx) {
      const long cc = x * per + c;
      f4 v = a[cc * 256 + t];
      if (NIN == 2) v += a2[cc * 256 + t];
      o[cc * 256 + t] = v;
    }
  } else {
    for (long c = 2 * b; c + 1 < nchunks; c += 2 * G) {
      f4 v = a[c * 256 + t], w = a[(c + 1) * 256 + t];
      if (NIN == 2) { v += a2[c * 256 + t]; w += a2[(c + 1) * 256 + t]; }
      o[c * 256 + t] = v;
      o[(c + 1) * 256 + t] = w;
    }
  }
}
// MODE 4: persistent workgroups take the next chunk from an atomic counter (requested one chunk ahead): the chip works on the
// lowest unassigned chunks, like the dispatcher does for one-chunk workgroups, instead of G streams drifting apart
template <int NIN, int Q>
__global__ __launch_bounds__(256) void queue_k(const f4* __restrict__ a, const f4* __restrict__ a2, f4* __restrict__ o, long nchunks,
                                               unsigned long long* counter) {
  __shared__ long slot[2];
  const int t = threadIdx.x;
  if (t == 0) slot[0] = (long)atomicAdd(counter, (unsigned long long)Q);
  __syncthreads();
  long c = slot[0];
  int ph = 0;
  while (c < nchunks) {
    if (t == 0) slot[ph ^ 1] = (long)atomicAdd(counter, (unsigned long long)Q);     // the next Q chunks, while these stream
    for (int q = 0; q < Q && c + q < nchunks; ++q) {
      f4 v = a[(c + q) * 256 + t];
      if (NIN == 2) v += a2[(c + q) * 256 + t];
      o[(c + q) * 256 + t] = v;
    }
    __syncthreads();
    ph ^= 1;
    c = slot[ph];
  }
}
// one chunk per workgroup, grid = nchunks
template <int NIN>
__global__ __launch_bounds__(256) void big_k(const f4* __restrict__ a, const f4* __restrict__ a2, f4* __restrict__ o) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  f4 v = a[i];
  if (NIN == 2) v += a2[i];
  o[i] = v;
}
template <class F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) f();
  hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 20;
}
int main() {
  const size_t bytes = (size_t)1 << 30;
  const long nchunks = bytes / 4096;
  f4 *a, *a2, *o;
  unsigned long long* ctr; hipMalloc(&ctr, 8);
  hipMalloc(&a, bytes); hipMalloc(&a2, bytes); hipMalloc(&o, bytes);
  hipMemset(a, 1, bytes); hipMemset(a2, 1, bytes); hipMemset(o, 0, bytes);
  for (int nin = 1; nin <= 2; ++nin) {
    const double gb = (nin + 1.0) * bytes / 1e6;
    printf("== %d input stream(s) + 1 output stream\n", nin);
    float ms = nin == 1 ? timeit([&] { hipLaunchKernelGGL(big_k<1>, dim3(nchunks), dim3(256), 0, 0, a, a2, o); })
                        : timeit([&] { hipLaunchKernelGGL(big_k<2>, dim3(nchunks), dim3(256), 0, 0, a, a2, o); });
    printf("one chunk per workgroup (grid %ld): %.0f GB/s\n", nchunks, gb / ms);
    for (int G : {512, 1024, 2048, 4096}) {
      float m[4];
#define RUN(MODE) m[MODE] = nin == 1 ? timeit([&] { hipLaunchKernelGGL((stream_k<MODE, 1>), dim3(G), dim3(256), 0, 0, a, a2, o, nchunks); }) \
                                    : timeit([&] { hipLaunchKernelGGL((stream_k<MODE, 2>), dim3(G), dim3(256), 0, 0, a, a2, o, nchunks); });
      RUN(0) RUN(1) RUN(2) RUN(3)
      float mq[3];
#define RUNQ(I, Q) mq[I] = nin == 1 ? timeit([&] { hipMemsetAsync(ctr, 0, 8, 0); hipLaunchKernelGGL((queue_k<1, Q>), dim3(G), dim3(256), 0, 0, a, a2, o, nchunks, ctr); }) \
                                    : timeit([&] { hipMemsetAsync(ctr, 0, 8, 0); hipLaunchKernelGGL((queue_k<2, Q>), dim3(G), dim3(256), 0, 0, a, a2, o, nchunks, ctr); });
      RUNQ(0, 4) RUNQ(1, 16) RUNQ(2, 64)
      printf("grid %5d: grid-stride %.0f  shard per workgroup %.0f  XCD-contiguous %.0f  two chunks in flight %.0f  atomic queue (4 / 16 / 64 chunks per grab) %.0f %.0f %.0f GB/s\n", G,
             gb / m[0], gb / m[1], gb / m[2], gb / m[3], gb / mq[0], gb / mq[1], gb / mq[2]);
    }
  }
  return 0;
}
