// Register-resident tall-skinny fp32 GEMM on the bf16 matrix pipe for the wide layers (BASELINE configs 3-5):
//
//   y[m, dout] = act(x[m, din] @ W + bias)      /      dx = dy @ W^T  (trans_w)
//
// exact 3-way bf16 split of both operands (kgcn_common.h), six v_mfma_f32_32x32x16_bf16 products per k-step.
// gemm3.hip stages both operands through LDS with a workgroup barrier per 32-wide k chunk and reaches ~140 TF; its
// waves spend their time in lockstep.  Here nothing is shared and nothing is synchronised:
//   * W is split ONCE per call by a small kernel into a fragment table in HBM/L2 (three bf16 pieces, already in MFMA
//     operand order: one 1 KiB lane-linear block per (k-step, 32-column tile, piece)); the GEMM's waves read their B
//     fragments straight from L2 into registers, one k-step ahead -- 0.4-1.5 MB for the layers of the path, L2 resident;
//   * one wave per SIMD (512 registers) owns a [128 rows x 128 columns] output block: 16 accumulator tiles = 256
//     registers, so every B fragment feeds four MFMAs and every A fragment four (128 B of L2 traffic per MFMA);
//   * x rows come from HBM directly in A-fragment layout (lane = row, 8 consecutive k = two 16-byte loads), one k-step
//     ahead, and are split in registers behind the MFMAs (half a split_pair per MFMA slot);
//   * no LDS, no barrier: the two waves that work on the same rows (column blocks 0/1 of a 256-wide layer) meet in L1/L2;
//   * bias is the initial accumulator value, the activation is applied in registers; a lane owns one COLUMN, so every
//     store instruction writes two full 128-byte lines (a row-per-lane layout with 16-byte stores was 5x slower: the
//     CU's address path takes one segment per lane).
// Matrix-pipe bound: 96 MFMAs = 3,072 cycles per k-step and wave next to ~230 other instructions.
#include "kgcn_common.h"

namespace kgcn {

constexpr int G4_BM = 128, G4_BN = 64;

#ifdef KGCN_PROBE   // development: per-wave cycle sums per phase (tools/gemm4_probe.py)
__device__ long long* g4_probe = nullptr;
#define G4P_DECL long long pt_[4] = {0, 0, 0, 0}; long long pc_ = __builtin_readcyclecounter();
#define G4P(k) { const long long n_ = __builtin_readcyclecounter(); pt_[k] += n_ - pc_; pc_ = n_; }
#define G4P_FLUSH if (g4_probe && lane == 0) { for (int k_ = 0; k_ < 4; ++k_) g4_probe[((long)blockIdx.x * 4 + wave) * 4 + k_] = pt_[k_]; }
#else
#define G4P_DECL
#define G4P(k)
#define G4P_FLUSH
#endif

// fragment table: [k-step ks][32-column tile nt][piece p][lane] x 16 bytes
__host__ __device__ inline long g4_table_entries(int din, int dout) {
  const long ksteps = (din + 15) / 16, ntiles = ((dout + G4_BN - 1) / G4_BN) * (G4_BN / 32);
  return ksteps * ntiles * 3 * 64;
}

// one thread per (k-step, column): the 2 x 8 k-values of its two lanes (hi = 0, 1)
__global__ __launch_bounds__(256) void g4_split_w_kernel(const float* __restrict__ w, long w_ld, int trans_w, int din,
                                                         int dout, u32x4* __restrict__ table) {
  const int ntiles = ((dout + G4_BN - 1) / G4_BN) * (G4_BN / 32);
  const long total = (long)((din + 15) / 16) * ntiles * 64;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int lane = (int)(i & 63);
    const long blk = i >> 6;
    const int nt = (int)(blk % ntiles), ks = (int)(blk / ntiles);
    const int n = 32 * nt + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + j;
      float x = 0.f;
      if (n < dout && k < din) x = trans_w ? w[(long)n * w_ld + k] : w[(long)k * w_ld + n];
      v[j] = x;
    }
    Frag3 f;
    split8(v, f);
    u32x4* d = table + ((long)(ks * ntiles + nt) * 3) * 64 + lane;
    d[0] = f.p1; d[64] = f.p2; d[128] = f.p3;
  }
}

// One wave: a [128 rows x 64 columns] block per item.  Per k-step 48 MFMAs (6 products x 2 column tiles x 4 row tiles).
// While item i+1 is contracted, the finished block of item i -- dumped to this wave's LDS slab at the item boundary --
// leaves the chip behind the MFMAs as 1 KiB stores of whole row segments.
constexpr int G4_SLD = G4_BN + 4;                       // slab row stride (floats): conflict-free dword writes / row reads
constexpr size_t G4_SLAB_BYTES = (size_t)G4_BM * G4_SLD * 4;

// Preconditions (launch_gemm4_fwd): m % 128 == 0 (a partial last row block goes to gemm3), dout % 64 == 0, y rows 16-byte
// aligned -- so that every deferred store is ONE unconditional instruction: a conditional vector-memory operation in the
// k-step makes the compiler's vmcnt bookkeeping pessimistic (it waited for the loads it had just issued).
// DPK: row groups of the previous block stored per k-step (ceil(32 / k-steps)).
// KMASK: din is not a multiple of 16 (the last k-step is partly beyond din: those values are replaced by 0 -- two selects
// per split pair, which the common widths do not pay).
template <bool XVEC, int DPK, bool KMASK>
__global__ __launch_bounds__(256, 1) void gemm4_fwd_kernel(const float* __restrict__ x, long m, int din, long x_ld,
                                                           const u32x4* __restrict__ table, const float* __restrict__ bias,
                                                           float* __restrict__ y, int dout, long y_ld, int act) {
  extern __shared__ __attribute__((aligned(16))) unsigned char g4_smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 31, hi = lane >> 5;
  float* slab = reinterpret_cast<float*>(g4_smem + (size_t)wave * G4_SLAB_BYTES);
  const int ncb = (dout + G4_BN - 1) / G4_BN;               // column blocks
  const int ntiles = ncb * (G4_BN / 32);
  const long nrb = (m + G4_BM - 1) / G4_BM;                 // row blocks
  const long items = nrb * ncb;
  const int ksteps = (din + 15) / 16;
  const long nwaves = (long)gridDim.x * 4;

  // deferred output of the previous item (its block sits in the slab): rows 4 i + (lane >> 4), 16 bytes per lane --
  // one instruction = four 256-byte row segments.  i is clamped to the last group (a k-step count that does not divide
  // 32 re-stores it: same bytes, harmless).
  float* pdst = y;                       // this lane's address in the previous block: row (lane >> 4), column 4 (lane & 15)
  const float* sl_r = slab + (lane >> 4) * G4_SLD + (lane & 15) * 4;
  auto drain = [&](int i) __attribute__((always_inline)) {
    const int ii = i < 31 ? i : 31;
    const f32x4 v = *reinterpret_cast<const f32x4*>(sl_r + ii * (4 * G4_SLD));
    *reinterpret_cast<f32x4*>(pdst + (long)ii * 4 * y_ld) = v;
  };

  G4P_DECL
  // the four waves of a workgroup take the column blocks of one row block first: they read the same x rows (L1 / L2)
  auto do_item = [&](long item, auto drain_tag) __attribute__((always_inline)) {
    constexpr bool DRAIN = decltype(drain_tag)::value;
    const long rb = item / ncb;
    const int cb = (int)(item - rb * ncb);
    const long row0 = rb * G4_BM;

    f32x16 acc[2][4];                                       // [nt][mt]: lane (li, hi) = column 32 nt + li
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int c = G4_BN * cb + 32 * nt + li;
      const float bv = (bias && c < dout) ? bias[c] : 0.f;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][mt][r] = bv;
    }

    // this lane's four rows (one per 32-row tile), clamped; rows beyond m are never stored
    const float* xrow[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const long r = row0 + 32 * mt + li;
      xrow[mt] = x + (r < m ? r : m - 1) * x_ld + 8 * hi;
    }
    const u32x4* tb = table + ((long)(2 * cb) * 3) * 64 + lane;       // + ks * ntiles * 192

    f32x4 rawA[4][2], rawB[4][2];                           // raw x of the next k-step, alternating sets
    u32x4 A0[4][3], A1[4][3], B0[2][3], B1[2][3];           // fragments, alternating by k-step parity
    auto load_raw = [&](f32x4 (&raw)[4][2], int ks) __attribute__((always_inline)) {
      const int k = 16 * ks + 8 * hi;                       // this lane's 8 k-values
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        if constexpr (XVEC) {
          raw[mt][0] = *reinterpret_cast<const f32x4*>(xrow[mt] + (k < din ? 16 * ks : 0));
          raw[mt][1] = *reinterpret_cast<const f32x4*>(xrow[mt] + (k + 4 < din ? 16 * ks + 4 : 0));
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            raw[mt][0][j] = xrow[mt][k + j < din ? 16 * ks + j : 0];
            raw[mt][1][j] = xrow[mt][k + 4 + j < din ? 16 * ks + 4 + j : 0];
          }
        }
      }
    };
    auto load_b = [&](u32x4 (&Bd)[2][3], int ks) __attribute__((always_inline)) {
      const u32x4* p = tb + (long)ks * ntiles * 192;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) Bd[nt][pc] = p[(nt * 3 + pc) * 64];
    };
    // split pair J (k-values 2J, 2J+1 of this lane) of tile mt, k-step ks, in two halves (one per MFMA slot: a whole
    // split_pair is 11 VALU operations, more than one MFMA covers); values beyond din count as 0
    float hr0 = 0.f, hr1 = 0.f, hs0 = 0.f, hv0 = 0.f, hv1 = 0.f;      // state between the two halves of a pair
    auto split_half = [&](const f32x4 (&raw)[4][2], u32x4 (&Ad)[4][3], int mt, int J, int ks, int half)
        __attribute__((always_inline)) {
      const unsigned msk = 0xffff0000u;
      if (half == 0) {
        float v0 = J < 2 ? raw[mt][0][2 * J] : raw[mt][1][2 * J - 4];
        float v1 = J < 2 ? raw[mt][0][2 * J + 1] : raw[mt][1][2 * J - 3];
        if constexpr (KMASK) {
          const int k = 16 * ks + 8 * hi + 2 * J;
          v0 = k < din ? v0 : 0.f;
          v1 = k + 1 < din ? v1 : 0.f;
        }
        hv0 = v0; hv1 = v1;
        hr0 = v0 - __uint_as_float(__float_as_uint(v0) & msk);
        hr1 = v1 - __uint_as_float(__float_as_uint(v1) & msk);
        hs0 = hr0 - __uint_as_float(__float_as_uint(hr0) & msk);
        asm volatile("" : "+v"(hr0), "+v"(hr1), "+v"(hs0), "+v"(hv0), "+v"(hv1));
      } else {
        const float s1 = hr1 - __uint_as_float(__float_as_uint(hr1) & msk);
        Ad[mt][0][J] = __builtin_amdgcn_perm(__float_as_uint(hv1), __float_as_uint(hv0), 0x07060302u);
        Ad[mt][1][J] = __builtin_amdgcn_perm(__float_as_uint(hr1), __float_as_uint(hr0), 0x07060302u);
        Ad[mt][2][J] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(hs0), 0x07060302u);
        KGCN_PIN3(Ad[mt][0][J], Ad[mt][1][J], Ad[mt][2][J]);
      }
    };
    auto split_into = [&](const f32x4 (&raw)[4][2], u32x4 (&Ad)[4][3], int mt, int J, int ks) __attribute__((always_inline)) {
      split_half(raw, Ad, mt, J, ks, 0);
      split_half(raw, Ad, mt, J, ks, 1);
    };

    // prologue: fragments of k-step 0; raw x of k-steps 1 (set B) and 2 (set A), B fragments of k-step 1 in flight
    // (the loads are issued in the order of the steady state -- raw, B, raw, B -- so that the waits the compiler derives
    // for the loop header from this path are the ones of the back edge, not stricter)
    load_raw(rawA, 0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int J = 0; J < 4; ++J) split_into(rawA, A0, mt, J, 0);
    load_raw(rawB, ksteps > 1 ? 1 : 0);
    load_b(B0, 0);
    load_raw(rawA, ksteps > 2 ? 2 : 0);
    load_b(B1, ksteps > 1 ? 1 : 0);
    G4P(0)

    // one k-step: 48 MFMAs on (Ac, Bc); behind them the fragments of k-step ks + 1 are split out of RAWN into An, RAWN is
    // reloaded with k-step ks + 3, Bc with k-step ks + 2, and four row groups of the previous item's block leave the slab
    auto kstep = [&](u32x4 (&Ac)[4][3], u32x4 (&An)[4][3], u32x4 (&Bc)[2][3], f32x4 (&RAWN)[4][2], int ks)
        __attribute__((always_inline)) {
      const int kn = ks + 1 < ksteps ? ks + 1 : ks;         // clamped: loads stay in bounds, results unused
      static_for<48>([&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value, pr = s >> 3, nt = (s >> 2) & 1, mt = s & 3;
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};   // smallest terms first
        acc[nt][mt] = mfma_bf16(Ac[mt][PA[pr]], Bc[nt][PB[pr]], acc[nt][mt]);
        // behind the MFMAs, at most ~5 other instructions each: slots 0-31 the 32 half split pairs of k-step ks + 1;
        // slots 32-39 RAWN <- raw x of k-step ks + 3 (one 16-byte load per slot: its registers are free once split);
        // slots 40-47: DPK row groups of the previous block leave the slab
        if constexpr (s < 32) {
          constexpr int u = s >> 1;
          split_half(RAWN, An, u >> 2, u & 3, kn, s & 1);
        } else if constexpr (s < 40) {
          constexpr int q = s - 32, lmt = q >> 1, h2 = q & 1;
          const int k3 = ks + 3 < ksteps ? ks + 3 : kn;
          const int k = 16 * k3 + 8 * hi + 4 * h2;
          if constexpr (XVEC) {
            RAWN[lmt][h2] = *reinterpret_cast<const f32x4*>(xrow[lmt] + (k < din ? 16 * k3 + 4 * h2 : 0));
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) RAWN[lmt][h2][j] = xrow[lmt][k + j < din ? 16 * k3 + 4 * h2 + j : 0];
          }
        } else if constexpr (DRAIN && s - 40 < DPK) {
          drain(DPK * ks + s - 40);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      load_b(Bc, ks + 2 < ksteps ? ks + 2 : kn);
    };
    // pairs of k-steps in the loop, an odd last one behind it: an exit in the middle of the body would merge two
    // different sets of loads in flight at the loop latch and make every wait of the first k-step 16 operations stricter
    int ks = 0;
    for (; ks + 1 < ksteps; ks += 2) {
      kstep(A0, A1, B0, rawB, ks);
      kstep(A1, A0, B1, rawA, ks + 1);
    }
    if (ks < ksteps) kstep(A0, A1, B0, rawB, ks);
    G4P(1)

    // ---- this item's block: activation, -> slab (lane = column: conflict-free dword writes) ----------------------
    if (act != KGCN_ACT_NONE) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[nt][mt][r] = act_fwd(acc[nt][mt][r], act);
    }
    __builtin_amdgcn_wave_barrier();
    {
      float* sw = slab + (4 * hi) * G4_SLD + li;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) sw[(32 * mt + (r & 3) + 8 * (r >> 2)) * G4_SLD + 32 * nt] = acc[nt][mt][r];
    }
    __builtin_amdgcn_wave_barrier();
    pdst = y + (row0 + (lane >> 4)) * y_ld + G4_BN * cb + (lane & 15) * 4;
    G4P(2)
  };
  long item = (long)blockIdx.x * 4 + wave;
  if (item < items) {
    do_item(item, std::false_type{});                        // nothing to drain yet
    for (item += nwaves; item < items; item += nwaves) do_item(item, std::true_type{});
    for (int i = 0; i < 32; ++i) drain(i);
  }
  G4P_FLUSH
}

#ifdef KGCN_PROBE
}  // namespace kgcn
extern "C" int kgcn_g4_probe_set(void* buf) {
  long long* p = static_cast<long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(kgcn::g4_probe), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
namespace kgcn {
#endif

int64_t gemm4_workspace_bytes(int din, int dout) { return g4_table_entries(din, dout) * 16; }

// returns -1 when the shape is not one the kernel takes (caller falls back to gemm3); rows_done = rows it computed
int launch_gemm4_fwd(const float* x, long m, int din, long x_ld, const float* w, long w_ld, int trans_w,
                     const float* bias, float* y, int dout, long y_ld, int act, void* workspace, long* rows_done,
                     hipStream_t s) {
  *rows_done = 0;
  const long mfull = (m / G4_BM) * G4_BM;
  if (dout % G4_BN != 0 || y_ld % 4 != 0 || !aligned16(y) || mfull == 0) return -1;
  u32x4* table = static_cast<u32x4*>(workspace);
  const long threads = g4_table_entries(din, dout) / 3;
  hipLaunchKernelGGL(g4_split_w_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, w, w_ld, trans_w, din,
                     dout, table);
  const long items = (mfull / G4_BM) * (dout / G4_BN);
  long blocks = (items + 3) / 4;
  if (blocks > kNumCU) blocks = kNumCU;
  const bool xvec = (din % 4 == 0) && (x_ld % 4 == 0) && aligned16(x);
  const size_t lds = 4 * G4_SLAB_BYTES;
  const int ksteps = (din + 15) / 16;
  const int dpk = (32 + ksteps - 1) / ksteps;
  const bool kmask = din % 16 != 0;
#define G4_LAUNCH(XV, DPKV, KM)                                                                                          \
  {                                                                                                                      \
    static thread_local bool attr_set = false;                                                                           \
    if (!attr_set) {                                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm4_fwd_kernel<XV, DPKV, KM>),                           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);                                 \
      attr_set = true;                                                                                                   \
    }                                                                                                                    \
    hipLaunchKernelGGL((gemm4_fwd_kernel<XV, DPKV, KM>), dim3((unsigned)blocks), dim3(256), lds, s, x, mfull, din, x_ld,    \
                       table, bias, y, dout, y_ld, act);                                                                 \
  }
#define G4_BY_DPK(XV, KM)                                    \
  if (dpk <= 2) G4_LAUNCH(XV, 2, KM)                          \
  else if (dpk <= 4) G4_LAUNCH(XV, 4, KM)                     \
  else if (dpk <= 8) G4_LAUNCH(XV, 8, KM)                     \
  else return -1;
  if (xvec && !kmask) { G4_BY_DPK(true, false) }
  else if (xvec) { G4_BY_DPK(true, true) }
  else { G4_BY_DPK(false, true) }
#undef G4_BY_DPK
#undef G4_LAUNCH
  *rows_done = mfull;
  return check_launch("gemm4_fwd_kernel");
}

}  // namespace kgcn
