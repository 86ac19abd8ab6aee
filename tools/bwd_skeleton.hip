// Memory skeleton of the fused GraphConv backward (development tool, GPU box only): kernels that move exactly the HBM bytes of
// graphconv_bwd_planes_kernel -- per graph two [32 x 64] fp32 tiles in (x, g), a CSR slice in, one [32 x 64] tile out -- with NO
// contraction / aggregation work, so what they reach is the ceiling the memory system gives this byte pattern (2 : 1 read : write,
// 8 KiB tiles).  Round 2 measured the persistent form at 4 and 8 waves per CU (profiles/r02_microbench_bwd_memory_skeleton.txt:
// 0.472-0.484 ms whatever the occupancy); round 6 adds the questions the two-wave-roles kernel raises:
//   P   persistent waves, one graph per wave, prefetch depth 1 / 2, 4 / 8 / 12 waves per CU           (the shipped structure)
//   PX  as P with x loaded in the dW A-fragment layout of the shipped kernel (16 x 8-byte loads, two 256-byte rows each)
//   H   persistent PAIRS: two waves share a graph (wave A: g + CSR in, dX out; wave B: x in), 8 waves per CU   (the two-role form)
//   N   non-persistent, one wave per graph, no prefetch, as many waves per CU as registers allow       (the SpMM's structure)
// usage: bwd_skeleton [graphs]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// MODE 0: x as 8 x dwordx4 rows;  1: x in the fragment layout (16 x dwordx2)
template <int MODE, int DEPTH, int BAR = 0, int NT = 256>
__global__ __launch_bounds__(NT) void skel_p(const float* __restrict__ x, const float* __restrict__ g,
                                              const f4* __restrict__ cv, float* __restrict__ dx, int T) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = gridDim.x * (blockDim.x >> 6);
  const int t0 = blockIdx.x * (blockDim.x >> 6) + wave;
  if (BAR == 0 && t0 >= T) return;
  const int cnt = t0 < T ? (T - 1 - t0) / nw + 1 : 0;
  const int cnt_max = BAR ? (T - 1) / nw + 1 : cnt;
  const int li = lane & 31, hi = lane >> 5;
  f4 xr[DEPTH][8], gr[DEPTH][8], cr[DEPTH][2];
  auto issue = [&](int k, int slot) {
    const int t = cnt > 0 ? t0 + (k < cnt ? k : cnt - 1) * nw : 0;
    const f4* gs = reinterpret_cast<const f4*>(g + (long)t * 2048);
    if constexpr (MODE == 0) {
      const f4* xs = reinterpret_cast<const f4*>(x + (long)t * 2048);
#pragma unroll
      for (int q = 0; q < 8; ++q) xr[slot][q] = xs[lane + 64 * q];
    } else {
      const float* xb = x + (long)t * 2048 + (8 * hi) * 64 + 2 * li;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const f2 v = *reinterpret_cast<const f2*>((q >> 3 ? xb + 16 * 64 : xb) + (q & 7) * 64);
        xr[slot][q >> 1][2 * (q & 1)] = v[0];
        xr[slot][q >> 1][2 * (q & 1) + 1] = v[1];
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) gr[slot][q] = gs[lane + 64 * q];
    cr[slot][0] = cv[(long)t * 80 + lane];
    cr[slot][1] = lane < 16 ? cv[(long)t * 80 + 64 + lane] : f4{0, 0, 0, 0};
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(d, d);
  for (int i = 0; i < cnt_max; i += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (BAR == 0 && i + d >= cnt) break;
      const int t = t0 + (i + d) * nw;
      f4 o[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = xr[d][q] * gr[d][q] + cr[d][q & 1];
      issue(i + d + DEPTH, d);
      if constexpr (BAR != 0) __syncthreads();
      float* dst = dx + (long)t * 2048;
      if (i + d < cnt) {
#pragma unroll
        for (int q = 0; q < 8; ++q)          // the C layout of dX^T = W dFW^T: lane = row, 32-byte segments at 256-byte stride
          *reinterpret_cast<f4*>(dst + li * 64 + (q >> 2) * 32 + 8 * (q & 3) + 4 * hi) = o[q];
      }
    }
  }
}

// pairs: wave 2p moves g + CSR in and dX out, wave 2p+1 moves x in (and hands a checksum over through LDS so that nothing is dead)
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
// LOAD: synthetic arithmetic per wave and graph next to the memory streams (no LDS traffic, no dependence on the loaded data beyond a
// final select): 1 = 48 v_mfma_f32_32x32x16_bf16 (the kernel's 96 per graph over the pair), 2 = those + 400 v_fma_f32, 3 = 400 v_fma_f32 only
template <int DEPTH, int BAR = 1, int LOAD = 0, int NT = 0, int CONTIG = 0, int LATE = 0>
__global__ __launch_bounds__(512) void skel_h(const float* __restrict__ x, const float* __restrict__ g,
                                              const f4* __restrict__ cv, float* __restrict__ dx, int T) {
  __shared__ float hand[8][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pair = wave >> 1, role = wave & 1;
  const int npairs = gridDim.x * 4;
  const int t0 = blockIdx.x * 4 + pair;
  const int cnt_max = (T - 1) / npairs + 1;
  const int cnt = t0 < T ? (T - 1 - t0) / npairs + 1 : 0;
  const int li = lane & 31, hi = lane >> 5;
  f4 r[DEPTH][8], cr[DEPTH][2];
  f16v acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
  bf8 fa, fb;
#pragma unroll
  for (int e = 0; e < 8; ++e) { fa[e] = (__bf16)(0.001f * (lane + e)); fb[e] = (__bf16)(0.002f * (lane - e)); }
  float va[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
  const float vm = 0.999f + 1e-9f * lane, vb = 1e-7f * lane;
  auto issue = [&](int k, int slot) {
    const int kk = k < cnt ? k : (cnt > 0 ? cnt - 1 : 0);
    int t = cnt > 0 ? t0 + kk * npairs : 0;
    if constexpr (CONTIG != 0) {                               // every pair walks a contiguous range of graphs instead of a strided one
      const long tc = (long)t0 * cnt_max + kk;
      t = (int)(tc < T ? tc : T - 1);
    }
    const f4* s = reinterpret_cast<const f4*>((role ? x : g) + (long)t * 2048);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if constexpr ((NT & 1) != 0) r[slot][q] = __builtin_nontemporal_load(s + lane + 64 * q);
      else r[slot][q] = s[lane + 64 * q];
    }
    if (!role) {
      cr[slot][0] = cv[(long)t * 80 + lane];
      cr[slot][1] = lane < 16 ? cv[(long)t * 80 + 64 + lane] : f4{0, 0, 0, 0};
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(d, d);
  for (int i = 0; i < cnt_max; i += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const bool live = i + d < cnt;
      f4 o[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = r[d][q];
      if (role) {
        f4 s = o[0] + o[1] + o[2] + o[3] + o[4] + o[5] + o[6] + o[7];
        hand[wave][lane] = s[0] + s[1] + s[2] + s[3];
      }
      if constexpr (LATE == 0) issue(i + d + DEPTH, d);      // LATE: the next graph is requested BEHIND the arithmetic (as the kernel's roles do)
      if constexpr (LOAD == 1 || LOAD == 2 || LOAD == 5) {
#pragma unroll
        for (int m = 0; m < (LOAD == 5 ? 24 : 12); ++m) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc2, 0, 0, 0);
          acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc3, 0, 0, 0);
        }
      }
      if constexpr (LOAD == 4) {                              // the same 48 MFMAs and 400 v_fma_f32, one MFMA then eight v_fma_f32
#pragma unroll
        for (int m = 0; m < 12; ++m) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (q == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
            if (q == 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
            if (q == 2) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc2, 0, 0, 0);
            if (q == 3) acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc3, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 8; ++e) va[e] = __builtin_fmaf(va[e], vm, vb);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      if constexpr (LOAD == 2 || LOAD == 3) {
#pragma unroll
        for (int m = 0; m < 50; ++m) {
#pragma unroll
          for (int e = 0; e < 8; ++e) va[e] = __builtin_fmaf(va[e], vm, vb);
        }
      }
      if constexpr (LATE != 0) issue(i + d + DEPTH, d);
      if constexpr (BAR != 0) __syncthreads();   // one workgroup barrier per graph, as the two-role kernel would pay
      if (!role && live) {
        const float hv = hand[wave + 1][lane];
        long td = t0 + (long)(i + d) * npairs;
        if constexpr (CONTIG != 0) { td = (long)t0 * cnt_max + (i + d); if (td >= T) td = T - 1; }
        float* dst = dx + td * 2048;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const f4 val = o[q] * hv + cr[d][q & 1];
          f4* dp = reinterpret_cast<f4*>(dst + li * 64 + (q >> 2) * 32 + 8 * (q & 3) + 4 * hi);
          if constexpr ((NT & 2) != 0) __builtin_nontemporal_store(val, dp);
          else *dp = val;
        }
      }
    }
  }
  if constexpr (LOAD != 0) {
    float z = va[0] + va[1] + va[2] + va[3] + va[4] + va[5] + va[6] + va[7];
#pragma unroll
    for (int e = 0; e < 16; ++e) z += acc0[e] + acc1[e] + acc2[e] + acc3[e];
    if (z == 123.456f) dx[lane] = z;
  }
}

// non-persistent: one wave per graph, everything requested at once
template <int WPB>
__global__ __launch_bounds__(64 * WPB) void skel_n(const float* __restrict__ x, const float* __restrict__ g,
                                                   const f4* __restrict__ cv, float* __restrict__ dx, int T) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = blockIdx.x * WPB + wave;
  if (t >= T) return;
  const int li = lane & 31, hi = lane >> 5;
  const f4* xs = reinterpret_cast<const f4*>(x + (long)t * 2048);
  const f4* gs = reinterpret_cast<const f4*>(g + (long)t * 2048);
  f4 xr[8], gr[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { xr[q] = xs[lane + 64 * q]; gr[q] = gs[lane + 64 * q]; }
  const f4 c0 = cv[(long)t * 80 + lane];
  const f4 c1 = lane < 16 ? cv[(long)t * 80 + 64 + lane] : f4{0, 0, 0, 0};
  float* dst = dx + (long)t * 2048;
#pragma unroll
  for (int q = 0; q < 8; ++q)
    *reinterpret_cast<f4*>(dst + li * 64 + (q >> 2) * 32 + 8 * (q & 3) + 4 * hi) = xr[q] * gr[q] + ((q & 1) ? c1 : c0);
}

// forward: x + CSR in, out tile out (17,316 algorithmic bytes per graph).  ROLE 0: every wave reads and writes its graph;
// ROLE 1: pairs -- wave A reads x + CSR, wave B writes the tile (handed over through LDS), one barrier per graph
template <int ROLE, int BAR, int NT>
__global__ __launch_bounds__(NT) void skel_f(const float* __restrict__ x, const f4* __restrict__ cv, float* __restrict__ out, int T) {
  __shared__ f4 tile[NT / 64][8][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int units = ROLE ? NT / 128 : NT / 64, unit = ROLE ? wave >> 1 : wave, role = ROLE ? wave & 1 : 0;
  const int nu = gridDim.x * units;
  const int t0 = blockIdx.x * units + unit;
  const int cnt = t0 < T ? (T - 1 - t0) / nu + 1 : 0;
  const int cnt_max = (T - 1) / nu + 1;
  f4 r[8], c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  auto issue = [&](int k) {
    const int t = cnt > 0 ? t0 + (k < cnt ? k : cnt - 1) * nu : 0;
    if (ROLE == 0 || role == 0) {
      const f4* s = reinterpret_cast<const f4*>(x + (long)t * 2048);
#pragma unroll
      for (int q = 0; q < 8; ++q) r[q] = s[lane + 64 * q];
      c0 = cv[(long)t * 80 + lane];
      c1 = lane < 16 ? cv[(long)t * 80 + 64 + lane] : f4{0, 0, 0, 0};
    }
  };
  issue(0);
  for (int i = 0; i < (BAR || ROLE ? cnt_max : cnt); ++i) {
    f4 o[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) o[q] = r[q] + ((q & 1) ? c1 : c0);
    if constexpr (ROLE != 0) {
      if (role == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) tile[wave][q][lane] = o[q];
      }
    }
    issue(i + 1);
    if constexpr (BAR != 0 || ROLE != 0) __syncthreads();
    if (i < cnt) {
      f4* dst = reinterpret_cast<f4*>(out + (long)(t0 + i * nu) * 2048);
      if constexpr (ROLE != 0) {
        if (role == 1) {
#pragma unroll
          for (int q = 0; q < 8; ++q) dst[lane + 64 * q] = tile[wave - 1][q][lane];
        }
        __syncthreads();
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) dst[lane + 64 * q] = o[q];
      }
    } else if (ROLE != 0) {
      __syncthreads();
    }
  }
}

// forward, persistent, prefetch depth DEPTH (graphs in flight per wave beyond the one being written), NT threads per workgroup
template <int DEPTH, int NT>
__global__ __launch_bounds__(NT) void skel_fd(const float* __restrict__ x, const f4* __restrict__ cv, float* __restrict__ out, int T) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = gridDim.x * (NT / 64);
  const int t0 = blockIdx.x * (NT / 64) + wave;
  if (t0 >= T) return;
  const int cnt = (T - 1 - t0) / nw + 1;
  f4 r[DEPTH][8], c0[DEPTH], c1[DEPTH];
  auto issue = [&](int k, int slot) {
    const int t = t0 + (k < cnt ? k : cnt - 1) * nw;
    const f4* s = reinterpret_cast<const f4*>(x + (long)t * 2048);
#pragma unroll
    for (int q = 0; q < 8; ++q) r[slot][q] = s[lane + 64 * q];
    c0[slot] = cv[(long)t * 80 + lane];
    c1[slot] = lane < 16 ? cv[(long)t * 80 + 64 + lane] : f4{0, 0, 0, 0};
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(d, d);
  for (int i = 0; i < cnt; i += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (i + d >= cnt) break;
      f4 o[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = r[d][q] + ((q & 1) ? c1[d] : c0[d]);
      issue(i + d + DEPTH, d);
      f4* dst = reinterpret_cast<f4*>(out + (long)(t0 + (i + d) * nw) * 2048);
#pragma unroll
      for (int q = 0; q < 8; ++q) dst[lane + 64 * q] = o[q];
    }
  }
}

static float* X; static float* G; static float* DX; static f4* CV; static int T;
template <typename L>
static float timeit(L launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 5; ++w) launch();
  hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 20;
}
static void linef(const char* name, float ms) {
  const double bytes = (double)T * (8192 * 2 + 932);
  printf("  %-64s %.3f ms  %.0f GB/s  %.3f of 8 TB/s\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
  fflush(stdout);
}
static void line(const char* name, float ms) {
  const double bytes = (double)T * (8192 * 3 + 932);           // the kernel's ALGORITHMIC bytes (the skeleton reads 1,280 B of CSR)
  printf("  %-64s %.3f ms  %.0f GB/s  %.3f of 8 TB/s\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
  fflush(stdout);
}

int main(int argc, char** argv) {
  T = argc > 1 ? atoi(argv[1]) : 100000;
  const size_t tb = (size_t)T * 8192;
  hipMalloc(&X, tb); hipMalloc(&G, tb); hipMalloc(&DX, tb); hipMalloc(&CV, (size_t)T * 1280);
  hipMemset(X, 0x11, tb); hipMemset(G, 0x22, tb); hipMemset(CV, 0, (size_t)T * 1280);
  printf("graphs = %d, algorithmic bytes per graph 25,508\n", T);
  char nm[128];
  for (int rep = 0; rep < 2; ++rep) {
    for (int wpc : {4, 8, 12}) {
      const int blocks = 256 * wpc / 4;
      snprintf(nm, sizeof nm, "P  persistent, depth 2, %2d waves/CU", wpc);
      line(nm, timeit([&] { hipLaunchKernelGGL((skel_p<0, 2>), dim3(blocks), dim3(256), 0, 0, X, G, CV, DX, T); }));
      snprintf(nm, sizeof nm, "P  persistent, depth 1, %2d waves/CU", wpc);
      line(nm, timeit([&] { hipLaunchKernelGGL((skel_p<0, 1>), dim3(blocks), dim3(256), 0, 0, X, G, CV, DX, T); }));
      snprintf(nm, sizeof nm, "PX persistent, depth 1, x in fragment layout, %2d waves/CU", wpc);
      line(nm, timeit([&] { hipLaunchKernelGGL((skel_p<1, 1>), dim3(blocks), dim3(256), 0, 0, X, G, CV, DX, T); }));
    }
    line("H  pairs (g+CSR+dX | x), depth 1, 8 waves/CU, barrier per graph",
         timeit([&] { hipLaunchKernelGGL((skel_h<1>), dim3(256), dim3(512), 0, 0, X, G, CV, DX, T); }));
    line("H  pairs (g+CSR+dX | x), depth 2, 8 waves/CU, barrier per graph",
         timeit([&] { hipLaunchKernelGGL((skel_h<2>), dim3(256), dim3(512), 0, 0, X, G, CV, DX, T); }));
    line("H  pairs, depth 2, 16 waves/CU",
         timeit([&] { hipLaunchKernelGGL((skel_h<2>), dim3(512), dim3(512), 0, 0, X, G, CV, DX, T); }));
    line("H  pairs, every pair a CONTIGUOUS range of graphs",
         timeit([&] { hipLaunchKernelGGL((skel_h<1, 1, 0, 0, 1>), dim3(256), dim3(512), 0, 0, X, G, CV, DX, T); }));
    line("H  pairs + MFMAs + FMAs, contiguous ranges",
         timeit([&] { hipLaunchKernelGGL((skel_h<1, 1, 2, 0, 1>), dim3(256), dim3(512), 0, 0, X, G, CV, DX, T); }));
    line("H  pairs + MFMAs + FMAs, the next graph requested BEHIND the arithmetic",
         timeit([&] { hipLaunchKernelGGL((skel_h<1, 1, 2, 0, 0, 1>), dim3(256), dim3(512), 0, 0, X, G, CV, DX, T); }));
    line("H  pairs, the next graph requested just before the barrier (no arithmetic)",
         timeit([&] { hipLaunchKernelGGL((skel_h<1, 1, 0, 0, 0, 1>), dim3(256), dim3(512), 0, 0, X, G, CV, DX, T); }));
    line("H  pairs, nontemporal LOADS",
         timeit([&] { hipLaunchKernelGGL((skel_h<1, 1, 0, 1>), dim3(256), dim3(512), 0, 0, X, G, CV, DX, T); }));
    line("H  pairs, nontemporal STORES",
         timeit([&] { hipLaunchKernelGGL((skel_h<1, 1, 0, 2>), dim3(256), dim3(512), 0, 0, X, G, CV, DX, T); }));
    line("H  pairs, nontemporal loads and stores",
         timeit([&] { hipLaunchKernelGGL((skel_h<1, 1, 0, 3>), dim3(256), dim3(512), 0, 0, X, G, CV, DX, T); }));
    line("H  pairs + MFMAs + FMAs, nontemporal loads",
         timeit([&] { hipLaunchKernelGGL((skel_h<1, 1, 2, 1>), dim3(256), dim3(512), 0, 0, X, G, CV, DX, T); }));
    line("H  pairs + 48 bf16 MFMAs per wave and graph",
         timeit([&] { hipLaunchKernelGGL((skel_h<1, 1, 1>), dim3(256), dim3(512), 0, 0, X, G, CV, DX, T); }));
    line("H  pairs + 48 bf16 MFMAs + 400 v_fma_f32 per wave and graph",
         timeit([&] { hipLaunchKernelGGL((skel_h<1, 1, 2>), dim3(256), dim3(512), 0, 0, X, G, CV, DX, T); }));
    line("H  pairs + 48 MFMAs and 400 v_fma_f32 INTERLEAVED (1 : 8)",
         timeit([&] { hipLaunchKernelGGL((skel_h<1, 1, 4>), dim3(256), dim3(512), 0, 0, X, G, CV, DX, T); }));
    line("H  pairs + 96 bf16 MFMAs per wave and graph (twice the kernel's)",
         timeit([&] { hipLaunchKernelGGL((skel_h<1, 1, 5>), dim3(256), dim3(512), 0, 0, X, G, CV, DX, T); }));
    line("H  pairs + 400 v_fma_f32 per wave and graph",
         timeit([&] { hipLaunchKernelGGL((skel_h<1, 1, 3>), dim3(256), dim3(512), 0, 0, X, G, CV, DX, T); }));
    line("H  pairs, depth 1, 8 waves/CU, NO barrier",
         timeit([&] { hipLaunchKernelGGL((skel_h<1, 0>), dim3(256), dim3(512), 0, 0, X, G, CV, DX, T); }));
    line("PB persistent, depth 1, 4 waves/CU, barrier per graph",
         timeit([&] { hipLaunchKernelGGL((skel_p<0, 1, 1, 256>), dim3(256), dim3(256), 0, 0, X, G, CV, DX, T); }));
    line("PB persistent, depth 1, x in fragment layout, 4 waves/CU, barrier per graph",
         timeit([&] { hipLaunchKernelGGL((skel_p<1, 1, 1, 256>), dim3(256), dim3(256), 0, 0, X, G, CV, DX, T); }));
    line("PB persistent, depth 1, 8 waves/CU as ONE 512-thread workgroup, barrier per graph",
         timeit([&] { hipLaunchKernelGGL((skel_p<0, 1, 1, 512>), dim3(256), dim3(512), 0, 0, X, G, CV, DX, T); }));
    line("P  persistent, depth 1, 8 waves/CU as ONE 512-thread workgroup, no barrier",
         timeit([&] { hipLaunchKernelGGL((skel_p<0, 1, 0, 512>), dim3(256), dim3(512), 0, 0, X, G, CV, DX, T); }));
    linef("F  forward: persistent, 8 waves/CU (512-thread workgroup), no barrier",
          timeit([&] { hipLaunchKernelGGL((skel_f<0, 0, 512>), dim3(256), dim3(512), 0, 0, X, CV, DX, T); }));
    linef("F  forward: persistent, 8 waves/CU (2 x 256-thread workgroups), no barrier",
          timeit([&] { hipLaunchKernelGGL((skel_f<0, 0, 256>), dim3(512), dim3(256), 0, 0, X, CV, DX, T); }));
    linef("FD forward: persistent, depth 1, 8 waves/CU (2 x 256)",
          timeit([&] { hipLaunchKernelGGL((skel_fd<1, 256>), dim3(512), dim3(256), 0, 0, X, CV, DX, T); }));
    linef("FD forward: persistent, depth 2, 8 waves/CU (2 x 256)",
          timeit([&] { hipLaunchKernelGGL((skel_fd<2, 256>), dim3(512), dim3(256), 0, 0, X, CV, DX, T); }));
    linef("FD forward: persistent, depth 3, 8 waves/CU (2 x 256)",
          timeit([&] { hipLaunchKernelGGL((skel_fd<3, 256>), dim3(512), dim3(256), 0, 0, X, CV, DX, T); }));
    linef("FD forward: persistent, depth 1, 16 waves/CU",
          timeit([&] { hipLaunchKernelGGL((skel_fd<1, 256>), dim3(1024), dim3(256), 0, 0, X, CV, DX, T); }));
    linef("FD forward: persistent, depth 1, 32 waves/CU",
          timeit([&] { hipLaunchKernelGGL((skel_fd<1, 256>), dim3(2048), dim3(256), 0, 0, X, CV, DX, T); }));
    linef("FD forward: persistent, depth 2, 16 waves/CU",
          timeit([&] { hipLaunchKernelGGL((skel_fd<2, 256>), dim3(1024), dim3(256), 0, 0, X, CV, DX, T); }));
    linef("FD forward: NON persistent, one wave per graph (64-thread workgroups)",
          timeit([&] { hipLaunchKernelGGL((skel_fd<1, 64>), dim3(T), dim3(64), 0, 0, X, CV, DX, T); }));
    linef("FB forward: persistent, 8 waves/CU (512-thread workgroup), barrier per graph",
          timeit([&] { hipLaunchKernelGGL((skel_f<0, 1, 512>), dim3(256), dim3(512), 0, 0, X, CV, DX, T); }));
    linef("FB forward: persistent, 4 waves/CU, barrier per graph",
          timeit([&] { hipLaunchKernelGGL((skel_f<0, 1, 256>), dim3(256), dim3(256), 0, 0, X, CV, DX, T); }));
    linef("FH forward: pairs (x + CSR in | tile out via LDS), 8 waves/CU",
          timeit([&] { hipLaunchKernelGGL((skel_f<1, 1, 512>), dim3(256), dim3(512), 0, 0, X, CV, DX, T); }));
    linef("FH forward: pairs, 16 waves/CU",
          timeit([&] { hipLaunchKernelGGL((skel_f<1, 1, 512>), dim3(512), dim3(512), 0, 0, X, CV, DX, T); }));
    line("N  one wave per graph, 64-thread workgroups",
         timeit([&] { hipLaunchKernelGGL((skel_n<1>), dim3(T), dim3(64), 0, 0, X, G, CV, DX, T); }));
    line("N  one wave per graph, 256-thread workgroups",
         timeit([&] { hipLaunchKernelGGL((skel_n<4>), dim3((T + 3) / 4), dim3(256), 0, 0, X, G, CV, DX, T); }));
  }
  return 0;
}
