// Price list for a one-wave-per-SIMD persistent kernel on gfx950 (development tool, GPU box only): cycles per
// v_mfma_f32_32x32x16_bf16 slot when N instructions of one kind sit behind every MFMA (4 waves per CU, all 256 CUs
// busy, memory operands L2/LDS resident).  The MFMA alone is 32 cycles; whatever a filler adds beyond that is
// its exposed issue cost next to the matrix pipe.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short i16x4 __attribute__((ext_vector_type(4)));
#define LDSP __attribute__((address_space(3)))

enum Kind { NONE, VFMA, GLD1_V, GLD1_S, GLD2_S, GLD4_S, GST4, GST1, DSR128, DSRTR, DSR32, DSW64, DSW128, DSW32, BUF1,
            LDSDMA4 };

template <int KIND, int N>
__global__ __launch_bounds__(256, 1) void k(const float* __restrict__ src, float* __restrict__ dst, float* out,
                                            int iters, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  bf16x8 pa, pb;
  for (int i = 0; i < 8; ++i) { pa[i] = (__bf16)(float)(lane + i); pb[i] = (__bf16)1.0f; }
  float v[8];
  const float pa_f = lane * 0.5f;
  for (int i = 0; i < 8; ++i) v[i] = lane * 0.001f + i;
  float* lds = reinterpret_cast<float*>(smem) + wave * 4096;          // 16 KiB per wave
  for (int i = lane; i < 4096; i += 64) lds[i] = i;
  const float* gs = src + ((blockIdx.x * 4 + wave) & 255) * 4096;     // 16 KiB per wave, L2 resident
  float* gd = dst + (blockIdx.x * 4 + wave) * 4096;
  f32x4 r4[4] = {};
  f32x2 r2[4] = {};
  float r1[4] = {};
  const f32x4 w4 = {1.f, 2.f, 3.f, 4.f};

  auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)gs, 0, 16384, 0x00020000);
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, acc[m], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const int o = ((m * N + j) & 7) * 64 + (it & 3) * 512;   // floats; varies with `it`: nothing can be hoisted
        if constexpr (KIND == VFMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j & 7]) : "v"(pa_f));
        if constexpr (KIND == GLD1_V) r1[j & 3] = (gs + lane)[o];
        if constexpr (KIND == GLD1_S) r1[j & 3] = gs[(unsigned)(lane + o)];
        if constexpr (KIND == GLD2_S) r2[j & 3] = *reinterpret_cast<const f32x2*>(gs + (unsigned)(2 * lane + 2 * o));
        if constexpr (KIND == GLD4_S) r4[j & 3] = *reinterpret_cast<const f32x4*>(gs + (unsigned)(4 * lane + (o & 0x3ff) * 3));
        if constexpr (KIND == GST4) *reinterpret_cast<f32x4*>(gd + (unsigned)(4 * lane + (o & 0x3ff) * 3)) = w4;
        if constexpr (KIND == GST1) gd[(unsigned)(lane + o)] = w4[0];
        if constexpr (KIND == DSR128) r4[j & 3] = *reinterpret_cast<const f32x4*>(lds + 4 * lane + (o & 0x3ff) * 3);
        if constexpr (KIND == DSR32) r1[j & 3] = lds[lane + o];
        if constexpr (KIND == DSRTR) {
          auto t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDSP i16x4*)(unsigned)(size_t)(LDSP char*)(lds + 2 * lane + o));
          r2[j & 3] = __builtin_bit_cast(f32x2, t);
        }
        if constexpr (KIND == DSW64) *reinterpret_cast<f32x2*>(lds + 2 * lane + o) = r2[j & 3];
        if constexpr (KIND == DSW128) *reinterpret_cast<f32x4*>(lds + 4 * lane + (o & 0x3ff) * 3) = w4;
        if constexpr (KIND == DSW32) lds[lane + o] = w4[1];
        if constexpr (KIND == BUF1) r1[j & 3] = __builtin_amdgcn_raw_buffer_load_b32(rs, (lane + o) * 4, 0, 0);
        if constexpr (KIND == LDSDMA4)
          __builtin_amdgcn_global_load_lds(gs + (unsigned)(4 * lane + (o & 0x3ff) * 3), (LDSP void*)(lds + (o & 0x3ff) * 3), 16, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) s += r1[i] + r2[i][0] + r2[i][1] + r4[i][0] + r4[i][1] + r4[i][2] + r4[i][3];
  for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[lane];
  if (lane == 0 && blockIdx.x == 7) { cyc[2 * wave] = t0; cyc[2 * wave + 1] = t1; }
}

template <int KIND, int N>
void run(const char* name, const float* src, float* dst, float* out, long long* cyc) {
  const int iters = 1000;
  hipFuncSetAttribute((const void*)k<KIND, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL((k<KIND, N>), dim3(256), dim3(256), 65536, 0, src, dst, out, iters, cyc);
  hipDeviceSynchronize();
  long long hh[8];
  hipMemcpy(hh, cyc, sizeof(hh), hipMemcpyDeviceToHost);
  double tot = 0;
  for (int i = 0; i < 4; ++i) tot += (double)(hh[2 * i + 1] - hh[2 * i]);
  const double per = tot / 4 / iters / 4;
  printf("%-44s x%d: %7.1f cycles per MFMA slot  -> %6.1f per filler beyond the bare MFMA\n", name, N, per,
         N ? (per - 32.8) / N : 0.0);
}

int main() {
  float *src, *dst, *out; long long* cyc;
  hipMalloc(&src, 256 * 16384); hipMalloc(&dst, 1024 * 16384); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 64);
  hipMemset(src, 0, 256 * 16384);
#define R(K, N, NAME) run<K, N>(NAME, src, dst, out, cyc);
  R(NONE, 0, "bare MFMA")
  R(VFMA, 4, "v_fma_f32") R(VFMA, 6, "v_fma_f32") R(VFMA, 8, "v_fma_f32") R(VFMA, 12, "v_fma_f32")
  R(GLD1_V, 1, "global_load_dword, 64-bit vaddr") R(GLD1_V, 2, "global_load_dword, 64-bit vaddr")
  R(GLD1_S, 1, "global_load_dword, saddr + voffset") R(GLD1_S, 2, "global_load_dword, saddr + voffset")
  R(GLD2_S, 1, "global_load_dwordx2, saddr") R(GLD2_S, 2, "global_load_dwordx2, saddr")
  R(GLD4_S, 1, "global_load_dwordx4, saddr") R(GLD4_S, 2, "global_load_dwordx4, saddr")
  R(BUF1, 1, "buffer_load_dword") R(BUF1, 2, "buffer_load_dword")
  R(LDSDMA4, 1, "global_load_lds_dwordx4") R(LDSDMA4, 2, "global_load_lds_dwordx4")
  R(GST4, 1, "global_store_dwordx4") R(GST4, 2, "global_store_dwordx4")
  R(GST1, 1, "global_store_dword") R(GST1, 2, "global_store_dword")
  R(DSR128, 1, "ds_read_b128") R(DSR128, 2, "ds_read_b128") R(DSR128, 4, "ds_read_b128")
  R(DSRTR, 2, "ds_read_b64_tr_b16") R(DSRTR, 4, "ds_read_b64_tr_b16")
  R(DSR32, 2, "ds_read_b32") R(DSR32, 4, "ds_read_b32")
  R(DSW64, 1, "ds_write_b64") R(DSW64, 3, "ds_write_b64")
  R(DSW128, 1, "ds_write_b128") R(DSW128, 2, "ds_write_b128")
  R(DSW32, 2, "ds_write_b32") R(DSW32, 4, "ds_write_b32")
  return 0;
}
