// The pieces of a training step that are NOT layers: the models' loss definitions and the optimiser update
// (SURVEY 8f N1).  In the reference these are a dozen TF ops each (tf.nn.sigmoid_cross_entropy_with_logits, reduce_sum,
// reduce_mean, tf.train.AdamOptimizer's per-variable update ops); written as torch ops they are ~20 + ~25 tiny launches
// per step, 0.3-0.4 ms of a 2 ms step (profiles/r03a_cfg4_ragged.txt).  Here:
//
//   masked_sigmoid_ce / masked_softmax_ce   ONE pass over the logits: per-graph cost, the gradient of the summed cost
//                                           with respect to the logits, block partial sums; a second one-block kernel
//                                           adds the partials in a fixed order (cost_sum, cost_mean)
//   adam_tf                                 tf.train.AdamOptimizer (kgcn/core.py:124) over ONE flat parameter buffer:
//                                             lr_t = lr sqrt(1 - b2^t) / (1 - b1^t)      (fp64, once per workgroup)
//                                             m = b1 m + (1 - b1) g ; v = b2 v + (1 - b2) g^2
//                                             p -= lr_t m / (sqrt(v) + eps)              ("epsilon hat" OUTSIDE the root)
//                                           t lives in device memory (a captured hipGraph advances it on replay).
#include "kgcn_common.h"

namespace kgcn {

constexpr int kLossBlock = 256;

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  red[threadIdx.x] = v;
  __syncthreads();
#pragma unroll
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  const float s = red[0];
  __syncthreads();
  return s;
}

// example_model/model_multitask.py:66-76:  cost[b] = mask[b] * sum_t mask_label[b,t] * ce(logits[b,t], labels[b,t])
//   ce  = max(x,0) - x z + log(1 + exp(-|x|))                                   (tf.nn.sigmoid_cross_entropy_with_logits)
//   wce = (1 - z) x + (1 + (q - 1) z) (log(1 + exp(-|x|)) + max(-x, 0))         (tf.nn.weighted_cross_entropy_with_logits)
// dlogits[b,t] = d(sum_b cost[b]) / d logits[b,t]
__global__ __launch_bounds__(kLossBlock) void sigmoid_ce_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                                               const float* __restrict__ mask, const float* __restrict__ mask_label,
                                                               long B, int T, int weighted, float q,
                                                               const float* __restrict__ qvec, float* __restrict__ cost,
                                                               float* __restrict__ dlogits, float* __restrict__ part,
                                                               float* __restrict__ sums_direct) {
  __shared__ float red[kLossBlock];
  const long b = (long)blockIdx.x * kLossBlock + threadIdx.x;
  float c = 0.f;
  if (b < B) {
    const float mk = mask[b];
    for (int t = 0; t < T; ++t) {
      const long i = b * T + t;
      const float x = logits[i], z = labels[i], ml = mask_label ? mask_label[i] : 1.f;
      const float sp = log1pf(__expf(-fabsf(x)));
      const float sg = 1.0f / (1.0f + __expf(-x));
      float ce, dce;
      if (!weighted) {
        ce = fmaxf(x, 0.f) - x * z + sp;
        dce = sg - z;
      } else {
        const float lw = 1.f + ((qvec ? qvec[t] : q) - 1.f) * z;   // info.pos_weight is per TASK (kgcn/data_util.py:563-568)
        ce = (1.f - z) * x + lw * (sp + fmaxf(-x, 0.f));
        dce = (1.f - z) - lw * (1.f - sg);
      }
      c += ml * ce;
      dlogits[i] = mk * ml * dce;
    }
    c *= mk;
    if (cost) cost[b] = c;
  }
  const float s = block_sum_256(c, red);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = s;
    if (sums_direct) {                                 // the batch fits this one workgroup: no finishing launch
      sums_direct[0] = s;
      sums_direct[1] = s / (float)B;
    }
  }
}

// example_model/model.py:56-61:  cost[b] = mask[b] * -(sum_c labels[b,c] log_softmax(logits[b])[c])
// labels: dense [B, C] (model.py's one-hot / soft labels), or NULL with label_idx [B] = the class index per graph
// (tf.nn.sparse_softmax_cross_entropy_with_logits, example_model/sparse.py:112): z[k] = (k == label_idx[b])
__global__ __launch_bounds__(kLossBlock) void softmax_ce_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                                               const long long* __restrict__ label_idx,
                                                               const float* __restrict__ mask, long B, int C,
                                                               float* __restrict__ cost, float* __restrict__ dlogits,
                                                               float* __restrict__ part, float* __restrict__ sums_direct) {
  __shared__ float red[kLossBlock];
  const long b = (long)blockIdx.x * kLossBlock + threadIdx.x;
  float c = 0.f;
  if (b < B) {
    const float* x = logits + b * C;
    const float* z = labels ? labels + b * C : nullptr;
    const int li = label_idx ? (int)label_idx[b] : -1;
    float mx = -INFINITY;
    for (int k = 0; k < C; ++k) mx = fmaxf(mx, x[k]);
    float se = 0.f, zs = 0.f, zx = 0.f;
    for (int k = 0; k < C; ++k) {
      const float zk = z ? z[k] : (k == li ? 1.f : 0.f);
      se += __expf(x[k] - mx);
      zs += zk;
      zx += zk * (x[k] - mx);
    }
    const float lse = __logf(se);
    const float mk = mask ? mask[b] : 1.f;
    c = mk * (zs * lse - zx);                                   // -sum_c z_c (x_c - mx - lse)
    const float inv = 1.0f / se;
    for (int k = 0; k < C; ++k)
      dlogits[b * C + k] = mk * (__expf(x[k] - mx) * inv * zs - (z ? z[k] : (k == li ? 1.f : 0.f)));
    if (cost) cost[b] = c;
  }
  const float s = block_sum_256(c, red);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = s;
    if (sums_direct) {                                 // the batch fits this one workgroup: no finishing launch
      sums_direct[0] = s;
      sums_direct[1] = s / (float)B;
    }
  }
}

// sums[0] = sum of the block partials (fixed order), sums[1] = sums[0] / B  (reduce_sum / reduce_mean over the PADDED batch)
__global__ __launch_bounds__(kLossBlock) void loss_finish_kernel(const float* __restrict__ part, int nparts, long B,
                                                                float* __restrict__ sums) {
  __shared__ float red[kLossBlock];
  float a = 0.f;
  for (int i = threadIdx.x; i < nparts; i += kLossBlock) a += part[i];
  const float s = block_sum_256(a, red);
  if (threadIdx.x == 0) {
    sums[0] = s;
    sums[1] = s / (float)B;
  }
}

// backward of the loss heads: d logits = dlogits (= d cost_sum / d logits) * (g_sum + g_opt / B), the two upstream gradients
// read from device scalars -- one launch instead of autograd's scalar multiply, add and broadcast multiply
__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ dlogits, const float* __restrict__ g_opt,
                                                        const float* __restrict__ g_sum, float inv_batch, long n,
                                                        float* __restrict__ out) {
  const float sc = (g_opt ? *g_opt * inv_batch : 0.f) + (g_sum ? *g_sum : 0.f);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = dlogits[i] * sc;
}

// the update of ONE element, the same instruction sequence in every kernel below (explicit fmas: -ffp-contract must not give the
// vector, scalar and per-segment paths different roundings)
__device__ __forceinline__ void adam_tf_update(float& p, float& m, float& v, float g, float lr_t, float b1, float b2, float c1,
                                               float c2, float eps) {
  m = __builtin_fmaf(m, b1, g * c1);
  v = __builtin_fmaf(v, b2, (g * g) * c2);
  p = __builtin_fmaf(-lr_t, m / (__builtin_sqrtf(v) + eps), p);
}

// (The step counter is advanced by a one-thread launch behind the update.  Advancing it from the update kernel's last workgroup --
// every workgroup adds an arrival to the counter's upper bits, the one that completes the count takes them out and ticks -- was
// tried: 3,584 same-address atomics at ~12 ns each made the update 48 us instead of 8 + 4, DESIGN.md lesson 30 / section 3.)
__device__ __forceinline__ long long adam_step_of(const long long* counter) { return *counter + 1; }

template <bool VEC>
__global__ __launch_bounds__(256) void adam_tf_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                      float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                      const long long* __restrict__ counter) {
  __shared__ float lr_s;
  if (threadIdx.x == 0) {
    const double t = (double)adam_step_of(counter);
    lr_s = (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
  }
  __syncthreads();
  const float lr_t = lr_s;
  const float c1 = 1.0f - b1, c2 = 1.0f - b2;
  if constexpr (VEC) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
      f32x4 pv = reinterpret_cast<f32x4*>(p)[i], mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
      const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float pj = pv[j], mj = mv[j], vj = vv[j];
        adam_tf_update(pj, mj, vj, gv[j], lr_t, b1, b2, c1, c2, eps);
        pv[j] = pj; mv[j] = mj; vv[j] = vj;
      }
      reinterpret_cast<f32x4*>(p)[i] = pv;
      reinterpret_cast<f32x4*>(m)[i] = mv;
      reinterpret_cast<f32x4*>(v)[i] = vv;
    }
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
      float pj = p[i], mj = m[i], vj = v[i];
      adam_tf_update(pj, mj, vj, g[i], lr_t, b1, b2, c1, c2, eps);
      p[i] = pj; m[i] = mj; v[i] = vj;
    }
  }
}

__global__ void adam_tick_kernel(long long* counter) { *counter += 1; }

}  // namespace kgcn

using namespace kgcn;

extern "C" int64_t kgcn_loss_workspace_bytes(int64_t batch) {
  if (batch <= 0) return 4;
  return ((batch + kLossBlock - 1) / kLossBlock) * 4;
}

static int loss_args(const char* who, const void* logits, const void* labels, const void* mask, int64_t batch, int32_t width,
                     const void* dlogits, const void* sums, const void* ws, int64_t wsb) {
  if (batch <= 0 || width <= 0) return fail("%s: bad shape batch=%lld width=%d", who, (long long)batch, width);
  if (!logits || !labels || !mask || !dlogits || !sums) return fail("%s: NULL operand", who);
  if (!ws || wsb < kgcn_loss_workspace_bytes(batch))
    return fail("%s: workspace %lld < %lld bytes", who, (long long)wsb, (long long)kgcn_loss_workspace_bytes(batch));
  if ((batch + kLossBlock - 1) / kLossBlock > 0x7fffffffLL) return fail("%s: batch too large", who);
  return 0;
}

extern "C" int kgcn_masked_sigmoid_ce_f32(const float* logits, const float* labels, const float* mask,
                                          const float* mask_label, int64_t batch, int32_t tasks, int32_t weighted,
                                          float pos_weight, const float* pos_weight_per_task, float* cost, float* dlogits,
                                          float* sums, void* workspace, int64_t workspace_bytes, void* stream) {
  if (int rc = loss_args("kgcn_masked_sigmoid_ce_f32", logits, labels, mask, batch, tasks, dlogits, sums, workspace,
                         workspace_bytes))
    return rc;
  const int nb = (int)((batch + kLossBlock - 1) / kLossBlock);
  float* part = static_cast<float*>(workspace);
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(sigmoid_ce_kernel, dim3(nb), dim3(kLossBlock), 0, s, logits, labels, mask, mask_label, (long)batch, tasks,
                     weighted ? 1 : 0, pos_weight, weighted ? pos_weight_per_task : nullptr, cost, dlogits, part,
                     nb == 1 ? sums : nullptr);
  if (int rc = check_launch("sigmoid_ce_kernel")) return rc;
  if (nb == 1) return 0;
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(kLossBlock), 0, s, part, nb, (long)batch, sums);
  return check_launch("loss_finish_kernel");
}

extern "C" int kgcn_masked_softmax_ce_f32(const float* logits, const float* labels, const float* mask, int64_t batch,
                                          int32_t classes, float* cost, float* dlogits, float* sums, void* workspace,
                                          int64_t workspace_bytes, void* stream) {
  if (int rc = loss_args("kgcn_masked_softmax_ce_f32", logits, labels, mask, batch, classes, dlogits, sums, workspace,
                         workspace_bytes))
    return rc;
  const int nb = (int)((batch + kLossBlock - 1) / kLossBlock);
  float* part = static_cast<float*>(workspace);
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(softmax_ce_kernel, dim3(nb), dim3(kLossBlock), 0, s, logits, labels, nullptr, mask, (long)batch, classes,
                     cost, dlogits, part, nb == 1 ? sums : nullptr);
  if (int rc = check_launch("softmax_ce_kernel")) return rc;
  if (nb == 1) return 0;
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(kLossBlock), 0, s, part, nb, (long)batch, sums);
  return check_launch("loss_finish_kernel");
}

extern "C" int kgcn_sparse_softmax_ce_f32(const float* logits, const int64_t* label_idx, const float* mask, int64_t batch,
                                          int32_t classes, float* cost, float* dlogits, float* sums, void* workspace,
                                          int64_t workspace_bytes, void* stream) {
  if (batch <= 0 || classes <= 0) return fail("kgcn_sparse_softmax_ce_f32: bad shape batch=%lld classes=%d", (long long)batch, classes);
  if (!logits || !label_idx || !dlogits || !sums) return fail("kgcn_sparse_softmax_ce_f32: NULL operand");
  if (!workspace || workspace_bytes < kgcn_loss_workspace_bytes(batch)) return fail("kgcn_sparse_softmax_ce_f32: workspace too small");
  const int nb = (int)((batch + kLossBlock - 1) / kLossBlock);
  float* part = static_cast<float*>(workspace);
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(softmax_ce_kernel, dim3(nb), dim3(kLossBlock), 0, s, logits, nullptr,
                     reinterpret_cast<const long long*>(label_idx), mask, (long)batch, classes, cost, dlogits, part,
                     nb == 1 ? sums : nullptr);
  if (int rc = check_launch("softmax_ce_kernel")) return rc;
  if (nb == 1) return 0;
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(kLossBlock), 0, s, part, nb, (long)batch, sums);
  return check_launch("loss_finish_kernel");
}

extern "C" int kgcn_loss_grad_f32(const float* dlogits, const float* g_opt, const float* g_sum, int64_t batch, int64_t n,
                                  float* out, void* stream) {
  if (n < 0 || batch <= 0) return fail("kgcn_loss_grad_f32: bad shape n=%lld batch=%lld", (long long)n, (long long)batch);
  if (n == 0) return 0;
  if (!dlogits || !out) return fail("kgcn_loss_grad_f32: NULL operand");
  long blocks = (n + 255) / 256;
  if (blocks > 4L * kNumCU) blocks = 4L * kNumCU;
  hipLaunchKernelGGL(loss_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), dlogits, g_opt, g_sum,
                     1.0f / (float)batch, (long)n, out);
  return check_launch("loss_grad_kernel");
}

// The same update with the gradients read where the backward pass left them (one tensor per parameter) instead of a packed
// copy: segment q = `numel[q]` floats at `offset[q]` of the flat parameter / moment buffers, its gradient at grad[q].
struct AdamSegs {
  const float* grad[KGCN_ADAM_MAX_SEGMENTS];
  long offset[KGCN_ADAM_MAX_SEGMENTS], numel[KGCN_ADAM_MAX_SEGMENTS];
};

__global__ __launch_bounds__(256) void adam_tf_multi_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                            AdamSegs sg, float lr, float b1, float b2, float eps,
                                                            const long long* __restrict__ counter) {
  __shared__ float lr_s;
  if (threadIdx.x == 0) {
    const double t = (double)adam_step_of(counter);
    lr_s = (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
  }
  __syncthreads();
  const float lr_t = lr_s;
  const float c1 = 1.0f - b1, c2 = 1.0f - b2;
  const int q = blockIdx.y;
  const float* __restrict__ g = sg.grad[q];
  const long off = sg.offset[q], n = sg.numel[q];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float pj = p[off + i], mj = m[off + i], vj = v[off + i];
    adam_tf_update(pj, mj, vj, g[i], lr_t, b1, b2, c1, c2, eps);
    p[off + i] = pj; m[off + i] = mj; v[off + i] = vj;
  }
}

extern "C" int kgcn_adam_tf_multi_f32(float* params, float* m, float* v, int64_t n, const kgcn_adam_segment* segments,
                                      int32_t num_segments, float lr, float beta1, float beta2, float eps, int64_t* step_counter,
                                      void* stream) {
  if (n < 0 || num_segments < 0) return fail("kgcn_adam_tf_multi_f32: negative size");
  if (!step_counter) return fail("kgcn_adam_tf_multi_f32: step_counter is NULL");
  if (num_segments > 0 && (!params || !m || !v || !segments)) return fail("kgcn_adam_tf_multi_f32: NULL operand");
  hipStream_t s = as_stream(stream);
  for (int base = 0; base < num_segments; base += KGCN_ADAM_MAX_SEGMENTS) {
    const int cnt = num_segments - base < KGCN_ADAM_MAX_SEGMENTS ? num_segments - base : KGCN_ADAM_MAX_SEGMENTS;
    AdamSegs sg{};
    long most = 1;
    for (int q = 0; q < cnt; ++q) {
      const kgcn_adam_segment& e = segments[base + q];
      if (e.offset < 0 || e.numel < 0 || e.offset + e.numel > n || (e.numel > 0 && !e.grad))
        return fail("kgcn_adam_tf_multi_f32: segment %d: offset %lld + %lld floats of %lld", base + q, (long long)e.offset,
                    (long long)e.numel, (long long)n);
      sg.grad[q] = e.grad; sg.offset[q] = (long)e.offset; sg.numel[q] = (long)e.numel;
      if (e.numel > most) most = (long)e.numel;
    }
    long blocks = (most + 255) / 256;
    if (blocks > 2L * kNumCU) blocks = 2L * kNumCU;
    hipLaunchKernelGGL(adam_tf_multi_kernel, dim3((unsigned)blocks, cnt), dim3(256), 0, s, params, m, v, sg, lr, beta1, beta2, eps,
                       reinterpret_cast<const long long*>(step_counter));
    if (int rc = check_launch("adam_tf_multi_kernel")) return rc;
  }
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, s, reinterpret_cast<long long*>(step_counter));
  return check_launch("adam_tick_kernel");
}

extern "C" int kgcn_adam_tf_f32(float* params, const float* grads, float* m, float* v, int64_t n, float lr, float beta1,
                                float beta2, float eps, int64_t* step_counter, void* stream) {
  if (n < 0) return fail("kgcn_adam_tf_f32: n < 0");
  if (!step_counter) return fail("kgcn_adam_tf_f32: step_counter is NULL");
  hipStream_t s = as_stream(stream);
  if (n > 0) {
    if (!params || !grads || !m || !v) return fail("kgcn_adam_tf_f32: NULL operand");
    const bool vec = (n % 4 == 0) && aligned16(params) && aligned16(grads) && aligned16(m) && aligned16(v);
    long blocks = ((vec ? n / 4 : n) + 255) / 256;
    if (blocks > (long)kNumCU * 8) blocks = (long)kNumCU * 8;
    if (blocks < 1) blocks = 1;
    const long long* ctr = reinterpret_cast<const long long*>(step_counter);
    if (vec)
      hipLaunchKernelGGL(adam_tf_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, params, grads, m, v, (long)n, lr,
                         beta1, beta2, eps, ctr);
    else
      hipLaunchKernelGGL(adam_tf_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, params, grads, m, v, (long)n, lr,
                         beta1, beta2, eps, ctr);
    if (int rc = check_launch("adam_tf_kernel")) return rc;
  }
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, s, reinterpret_cast<long long*>(step_counter));
  return check_launch("adam_tick_kernel");
}
