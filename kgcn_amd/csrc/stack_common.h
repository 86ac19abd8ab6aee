// Shared by the two routes of the cross-layer kernels: stack.hip (one graph per workgroup trip, plain fp32 FMAs) and
// stack_tile.hip (64-row tiles of whole graphs on the f32 MFMA).
#pragma once
#include "kgcn_common.h"

namespace kgcn {

constexpr int SK_MAXL = KGCN_STACK_MAX_LAYERS;
constexpr int SK_LD = 64;          // activation tile leading dimension (floats)
constexpr int SK_WLD = 65;         // weight leading dimension: conflict-free along k AND along j

constexpr int S2_MAXM = 4;         // matrix layers the tile backward keeps dW accumulators for (16 VGPRs each; five spill)

struct StackArgs {
  int nl, N, gather, max_nnz;
  int kind[SK_MAXL], act[SK_MAXL], din[SK_MAXL], dout[SK_MAXL];
  const float* w[SK_MAXL];
  const float* b[SK_MAXL];
  const float* mean[SK_MAXL];
  const float* var[SK_MAXL];
  float eps[SK_MAXL];
  float* out[SK_MAXL];            // saved layer outputs [T, N, dout]
  int woff[SK_MAXL];              // float offset of the layer's weight block in LDS (kind 2: scale | shift)
  int poff[SK_MAXL];              // float offset of the layer's gradients in the flat parameter-gradient layout
  int goff[SK_MAXL];              // float offset of the layer's accumulator block in LDS (backward)
  int wtotal;                     // floats of all weight blocks
  int ptotal;                     // floats of the flat parameter-gradient layout
  // tile route (stack_tile.hip)
  int woff2[SK_MAXL];             // float offset of the layer's [64 x 66] (+ 64) weight block in LDS
  int wtotal2;
  int G;                          // whole graphs per 64-row tile
  int max_ent;                    // stored entries one tile can hold
  int mslot[SK_MAXL];             // index of the layer's dW accumulator among the matrix layers (kind 2: -1)
  int mlayer[S2_MAXM];            // ... and back: the layer of matrix slot i (-1: unused)
  int abl;                        // development (KGCN_DEV_KNOBS builds, KGCN_S2_ABL): phases to skip when measuring; else 0
};

// ---- tile route ------------------------------------------------------------------------------------------------------
constexpr int S2_LD = 66;          // tile leading dimension: ds_read_b64 of 32 rows at one column pair covers the 64 banks once
constexpr int S2_R = 64;           // rows of a tile
constexpr int S2_TILE = (S2_R + 1) * S2_LD;   // + one row of zeros (target of the ELL padding)
constexpr int S2_MAXL = SK_MAXL;   // layers (one dbias / dgamma / dbeta register pair per thread each)
constexpr int S2_AUX = 68 + 3 * 64 + 4;   // ints: row pointers | first tile row of the row's graph | its valid rows | its index | misc

struct Stack2Plan { bool ok; size_t lds_fwd, lds_bwd; };

Stack2Plan stack2_plan(StackArgs& a);     // fills woff2 / wtotal2 / G / max_ent
int stack2_blocks(long T, int G);
int launch_stack2_fwd(const StackArgs& a, const Stack2Plan& p, const int* rowptr, const int2* cv, const float* x,
                      const int* enabled, long T, float* pooled, hipStream_t s);
int launch_stack2_bwd(const StackArgs& a, const Stack2Plan& p, const int* rowptr_t, const int2* cv_t, const float* x,
                      const int* enabled, long T, const float* dlast, float* dx, float* part, int blocks, hipStream_t s);

}  // namespace kgcn
