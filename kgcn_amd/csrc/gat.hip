// GAT (kgcn/layers.py:477-542): the reference's attention layer WITHOUT its own weight matrix.  Per graph t
// and adjacency channel, over the stored entries e = (row r_e, col c_e) -- the values are not used:
//   t_e = x[c_e] . wa[0:D] + x[r_e] . wa[D:2D]  = u[c_e] + v[r_e]
//   E_e = exp(leaky_relu(t_e, 0.2));  denom[i] = sum_{e in row i} E_e
//   alpha_e = E_e / (denom[c_e] + 1e-10)          (the denominator is gathered at the COLUMN index, :528-529)
//   out[i] (+)= sigmoid( sum_{e in row i} alpha_e x[c_e] )
// The reference builds this with one-hot matmuls per graph (O(N * nnz * D)); here it is three / six small
// CSR-vector kernels (lane group per row, gathers through L1/L2), all quantities per NODE (u, v, denom,
// d denom, du, dv) in a caller-provided workspace, no atomics: sums grouped by column walk the A^T container.
#include "kgcn_common.h"

namespace kgcn {

int launch_reduce_partials(const float* part, int nparts, long n, float* out, hipStream_t s);

constexpr float kGatSlope = 0.2f;     // tf.nn.leaky_relu default
constexpr float kGatEps = 1.0e-10f;

__device__ __forceinline__ float group_sum(float v, int lpr) {
  for (int o = lpr >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float lrelu(float t) { return t > 0.f ? t : kGatSlope * t; }
__device__ __forceinline__ float lrelu_grad(float t) { return t > 0.f ? 1.f : kGatSlope; }
__device__ __forceinline__ float sigmoidf(float r) { return 1.f / (1.f + __expf(-r)); }

#define GAT_ROW_LOOP(total)                                                                          \
  const int lpr = 1 << lpr_log2;                                                                     \
  const int cl = threadIdx.x & (lpr - 1);                                                            \
  const long nworkers = ((long)gridDim.x * 256) >> lpr_log2;                                         \
  for (long row = ((long)blockIdx.x * 256 + threadIdx.x) >> lpr_log2; row < (total); row += nworkers)

// u[node] = x[node] . wa[0:D],  v[node] = x[node] . wa[D:2D]
__global__ __launch_bounds__(256) void gat_scores_kernel(const float* __restrict__ x, const float* __restrict__ wa,
                                                          float* __restrict__ u, float* __restrict__ v, long nodes,
                                                          int d, int lpr_log2) {
  GAT_ROW_LOOP(nodes) {
    const float* xr = x + row * d;
    float su = 0.f, sv = 0.f;
    for (int c = cl; c < d; c += lpr) {
      const float xv = xr[c];
      su += xv * wa[c];
      sv += xv * wa[d + c];
    }
    su = group_sum(su, lpr);
    sv = group_sum(sv, lpr);
    if (cl == 0) { u[row] = su; v[row] = sv; }
  }
}

// denom[row] = sum over the row's entries of exp(lrelu(u[col] + v[row]))
__global__ __launch_bounds__(256) void gat_denom_kernel(const int* __restrict__ rowptr, const int2* __restrict__ cv,
                                                         const float* __restrict__ u, const float* __restrict__ v,
                                                         float* __restrict__ denom, int M, long total_rows) {
  for (long row = (long)blockIdx.x * 256 + threadIdx.x; row < total_rows; row += (long)gridDim.x * 256) {
    const long nb = (row / M) * M;
    const float vr = v[row];
    float s = 0.f;
    for (int k = rowptr[row]; k < rowptr[row + 1]; ++k) s += __expf(lrelu(u[nb + cv[k].x] + vr));
    denom[row] = s;
  }
}

// MODE 0: out[row] = beta*out + sigmoid(r[row]);  MODE 1: dr[row] = g[row] * sigmoid'(r[row])
template <int MODE>
__global__ __launch_bounds__(256) void gat_aggregate_kernel(
    const int* __restrict__ rowptr, const int2* __restrict__ cv, const float* __restrict__ x,
    const float* __restrict__ u, const float* __restrict__ v, const float* __restrict__ denom,
    const float* __restrict__ g, float* __restrict__ out, int M, long total_rows, int d, int lpr_log2, float beta) {
  GAT_ROW_LOOP(total_rows) {
    const long nb = (row / M) * M;
    const int s = rowptr[row], e = rowptr[row + 1];
    const float vr = v[row];
    for (int c = cl; c < d; c += lpr) {
      float r = 0.f;
      for (int k = s; k < e; ++k) {
        const long j = nb + cv[k].x;
        const float al = __expf(lrelu(u[j] + vr)) / (denom[j] + kGatEps);
        r += al * x[j * d + c];
      }
      const float sg = sigmoidf(r);
      float* o = out + row * d + c;
      if constexpr (MODE == 0) *o = (beta != 0.f ? *o : 0.f) + sg;
      else *o = g[row * d + c] * sg * (1.f - sg);
    }
  }
}

// Column-grouped sums, walking A^T: row j of A^T lists the entries (i, j) of A.
//   dx[j] (+)= sum_i alpha_(i,j) dr[i];  ddenom[j] = -sum_i dalpha E / den_j^2;  du1[j] = sum_i dalpha E l' / den_j
// with dalpha_(i,j) = <dr[i], x[j]>.
__global__ __launch_bounds__(256) void gat_bwd_cols_kernel(
    const int* __restrict__ rowptr_t, const int2* __restrict__ cv_t, const float* __restrict__ x,
    const float* __restrict__ u, const float* __restrict__ v, const float* __restrict__ denom,
    const float* __restrict__ dr, float* __restrict__ dx, float* __restrict__ ddenom, float* __restrict__ du1,
    int K, int M, long total_cols, int d, int lpr_log2, float beta) {
  GAT_ROW_LOOP(total_cols) {
    const long t = row / K;
    const long nbr = t * M;                       // node base of the ORIGINAL rows (M == K: square adjacency)
    const int s = rowptr_t[row], e = rowptr_t[row + 1];
    const float uj = u[row], den = denom[row] + kGatEps;
    const float* xj = x + row * d;
    float sdd = 0.f, sdu = 0.f;
    for (int k = s; k < e; ++k) {
      const long i = nbr + cv_t[k].x;
      const float* dri = dr + i * d;
      float dot = 0.f;
      for (int c = cl; c < d; c += lpr) dot += dri[c] * xj[c];
      dot = group_sum(dot, lpr);
      const float tt = uj + v[i];
      const float E = __expf(lrelu(tt));
      sdd -= dot * E / (den * den);
      sdu += dot * E * lrelu_grad(tt) / den;
    }
    for (int c = cl; c < d; c += lpr) {
      float acc = 0.f;
      for (int k = s; k < e; ++k) {
        const long i = nbr + cv_t[k].x;
        acc += __expf(lrelu(uj + v[i])) / den * dr[i * d + c];
      }
      float* o = dx + row * d + c;
      *o = (beta != 0.f ? *o : 0.f) + acc;
    }
    if (cl == 0) { ddenom[row] = sdd; du1[row] = sdu; }
  }
}

// Row-grouped: dv[i] = sum_{e in row i} (dalpha_e / den_{c_e} + ddenom[i]) E_e l'_e
__global__ __launch_bounds__(256) void gat_bwd_rows_kernel(
    const int* __restrict__ rowptr, const int2* __restrict__ cv, const float* __restrict__ x,
    const float* __restrict__ u, const float* __restrict__ v, const float* __restrict__ denom,
    const float* __restrict__ dr, const float* __restrict__ ddenom, float* __restrict__ dv, int M,
    long total_rows, int d, int lpr_log2) {
  GAT_ROW_LOOP(total_rows) {
    const long nb = (row / M) * M;
    const float vr = v[row], ddi = ddenom[row];
    const float* dri = dr + row * d;
    float s = 0.f;
    for (int k = rowptr[row]; k < rowptr[row + 1]; ++k) {
      const long j = nb + cv[k].x;
      const float* xj = x + j * d;
      float dot = 0.f;
      for (int c = cl; c < d; c += lpr) dot += dri[c] * xj[c];
      dot = group_sum(dot, lpr);
      const float tt = u[j] + vr;
      s += (dot / (denom[j] + kGatEps) + ddi) * __expf(lrelu(tt)) * lrelu_grad(tt);
    }
    if (cl == 0) dv[row] = s;
  }
}

// du[j] = du1[j] + sum_{(i,j)} ddenom[i] E l'   (A^T walk, scalar work)
__global__ __launch_bounds__(256) void gat_bwd_du_kernel(const int* __restrict__ rowptr_t, const int2* __restrict__ cv_t,
                                                          const float* __restrict__ u, const float* __restrict__ v,
                                                          const float* __restrict__ ddenom, const float* __restrict__ du1,
                                                          float* __restrict__ du, int K, int M, long total_cols) {
  for (long col = (long)blockIdx.x * 256 + threadIdx.x; col < total_cols; col += (long)gridDim.x * 256) {
    const long nbr = (col / K) * M;
    const float uj = u[col];
    float s = du1[col];
    for (int k = rowptr_t[col]; k < rowptr_t[col + 1]; ++k) {
      const long i = nbr + cv_t[k].x;
      const float tt = uj + v[i];
      s += ddenom[i] * __expf(lrelu(tt)) * lrelu_grad(tt);
    }
    du[col] = s;
  }
}

// dx[j] += du[j] wa[0:D] + dv[j] wa[D:2D];  partial d wa of this workgroup's node range: [2D] per workgroup
__global__ __launch_bounds__(256) void gat_bwd_dots_kernel(const float* __restrict__ x, const float* __restrict__ wa,
                                                            const float* __restrict__ du, const float* __restrict__ dv,
                                                            float* __restrict__ dx, float* __restrict__ part, long nodes,
                                                            int d, long nodes_per_block) {
  const long n0 = (long)blockIdx.x * nodes_per_block;
  const long n1 = n0 + nodes_per_block < nodes ? n0 + nodes_per_block : nodes;
  for (int c = threadIdx.x; c < d; c += 256) {
    const float wu = wa[c], wv = wa[d + c];
    float su = 0.f, sv = 0.f;
    for (long j = n0; j < n1; ++j) {
      const float a = du[j], b = dv[j], xv = x[j * d + c];
      dx[j * d + c] += a * wu + b * wv;
      su += a * xv;
      sv += b * xv;
    }
    part[(long)blockIdx.x * 2 * d + c] = su;
    part[(long)blockIdx.x * 2 * d + d + c] = sv;
  }
}

static int ilog2c(int v) { int l = 0; while ((1 << l) < v) ++l; return l > 6 ? 6 : l; }
static unsigned row_blocks(long rows, int lpr_log2) {
  long b = ((rows << lpr_log2) + 255) / 256;
  if (b > (long)kNumCU * 64) b = (long)kNumCU * 64;
  return (unsigned)(b < 1 ? 1 : b);
}
constexpr int kGatDotBlocks = 512;

}  // namespace kgcn

using namespace kgcn;

extern "C" int64_t kgcn_gat_workspace_bytes(int32_t num_graphs, int32_t n_nodes, int32_t d) {
  if (num_graphs <= 0 || n_nodes <= 0 || d <= 0) return 0;
  const int64_t nodes = (int64_t)num_graphs * n_nodes;
  // u, v, denom, ddenom, du1, du, dv | dr [nodes x d] | d wa partials
  return (7 * nodes + nodes * d + (int64_t)kGatDotBlocks * 2 * d) * 4;
}

static int gat_check(const kgcn_csr_batch* a, const char* who, int d, const void* ws, int64_t ws_bytes) {
  if (int rc = validate_csr(a, who)) return rc;
  if (a->rows != a->cols) return fail("%s: adjacency must be square", who);
  if (d <= 0) return fail("%s: d=%d", who, d);
  const int64_t need = kgcn_gat_workspace_bytes(a->num_graphs, a->rows, d);
  if (need > 0 && (!ws || ws_bytes < need))
    return fail("%s: workspace %lld < %lld bytes", who, (long long)ws_bytes, (long long)need);
  return 0;
}

extern "C" int kgcn_gat_fwd_f32(const kgcn_csr_batch* a, const float* x, int32_t d, const float* weight_a,
                                float* out, float beta, void* workspace, int64_t workspace_bytes, void* stream) {
  if (int rc = gat_check(a, "kgcn_gat_fwd_f32", d, workspace, workspace_bytes)) return rc;
  if (a->num_graphs == 0 || a->rows == 0) return 0;
  if (!x || !weight_a || !out) return fail("kgcn_gat_fwd_f32: NULL operand");
  if (beta != 0.f && beta != 1.f) return fail("kgcn_gat_fwd_f32: beta must be 0 or 1");
  hipStream_t s = as_stream(stream);
  const long nodes = (long)a->num_graphs * a->rows;
  float* u = static_cast<float*>(workspace);
  float* v = u + nodes;
  float* denom = v + nodes;
  const int2* cv = reinterpret_cast<const int2*>(a->cv);
  const int lg = ilog2c(d);
  hipLaunchKernelGGL(gat_scores_kernel, dim3(row_blocks(nodes, lg)), dim3(256), 0, s, x, weight_a, u, v, nodes, d, lg);
  hipLaunchKernelGGL(gat_denom_kernel, dim3(row_blocks(nodes, 0)), dim3(256), 0, s, a->rowptr, cv, u, v, denom, a->rows,
                     nodes);
  hipLaunchKernelGGL((gat_aggregate_kernel<0>), dim3(row_blocks(nodes, lg)), dim3(256), 0, s, a->rowptr, cv, x, u, v,
                     denom, nullptr, out, a->rows, nodes, d, lg, beta);
  return check_launch("gat_aggregate_kernel");
}

extern "C" int kgcn_gat_bwd_f32(const kgcn_csr_batch* a, const kgcn_csr_batch* at, const float* x, int32_t d,
                                const float* weight_a, const float* dout_grad, float* dx, float beta,
                                float* dweight_a, void* workspace, int64_t workspace_bytes, void* stream) {
  if (int rc = gat_check(a, "kgcn_gat_bwd_f32", d, workspace, workspace_bytes)) return rc;
  if (int rc = validate_csr(at, "kgcn_gat_bwd_f32")) return rc;
  if (at->num_graphs != a->num_graphs || at->rows != a->cols || at->cols != a->rows || at->nnz != a->nnz)
    return fail("kgcn_gat_bwd_f32: `at` is not the transposed batch of `a`");
  if (!dweight_a) return fail("kgcn_gat_bwd_f32: dweight_a is NULL");
  hipStream_t s = as_stream(stream);
  if (a->num_graphs == 0 || a->rows == 0) {
    (void)hipMemsetAsync(dweight_a, 0, (size_t)2 * d * 4, s);
    return 0;
  }
  if (!x || !weight_a || !dout_grad || !dx) return fail("kgcn_gat_bwd_f32: NULL operand");
  if (beta != 0.f && beta != 1.f) return fail("kgcn_gat_bwd_f32: beta must be 0 or 1");
  const long nodes = (long)a->num_graphs * a->rows;
  float* u = static_cast<float*>(workspace);
  float* v = u + nodes;
  float* denom = v + nodes;
  float* ddenom = denom + nodes;
  float* du1 = ddenom + nodes;
  float* du = du1 + nodes;
  float* dv = du + nodes;
  float* dr = dv + nodes;
  float* part = dr + nodes * d;
  const int2* cv = reinterpret_cast<const int2*>(a->cv);
  const int2* cvt = reinterpret_cast<const int2*>(at->cv);
  const int lg = ilog2c(d);
  const dim3 gv(row_blocks(nodes, lg)), gs(row_blocks(nodes, 0)), blk(256);
  hipLaunchKernelGGL(gat_scores_kernel, gv, blk, 0, s, x, weight_a, u, v, nodes, d, lg);
  hipLaunchKernelGGL(gat_denom_kernel, gs, blk, 0, s, a->rowptr, cv, u, v, denom, a->rows, nodes);
  hipLaunchKernelGGL((gat_aggregate_kernel<1>), gv, blk, 0, s, a->rowptr, cv, x, u, v, denom, dout_grad, dr, a->rows,
                     nodes, d, lg, 0.f);
  hipLaunchKernelGGL(gat_bwd_cols_kernel, gv, blk, 0, s, at->rowptr, cvt, x, u, v, denom, dr, dx, ddenom, du1, at->rows,
                     a->rows, nodes, d, lg, beta);
  hipLaunchKernelGGL(gat_bwd_rows_kernel, gv, blk, 0, s, a->rowptr, cv, x, u, v, denom, dr, ddenom, dv, a->rows, nodes, d,
                     lg);
  hipLaunchKernelGGL(gat_bwd_du_kernel, gs, blk, 0, s, at->rowptr, cvt, u, v, ddenom, du1, du, at->rows, a->rows, nodes);
  int nblk = (int)(nodes < kGatDotBlocks ? nodes : kGatDotBlocks);
  const long npb = (nodes + nblk - 1) / nblk;
  nblk = (int)((nodes + npb - 1) / npb);
  hipLaunchKernelGGL(gat_bwd_dots_kernel, dim3(nblk), blk, 0, s, x, weight_a, du, dv, dx, part, nodes, d, npb);
  if (int rc = check_launch("gat_bwd kernels")) return rc;
  return launch_reduce_partials(part, nblk, 2L * d, dweight_a, s);
}
