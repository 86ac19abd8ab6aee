/*
 * kgcn_hip.h -- C ABI of libkgcn_hip.so: the MI355X (gfx950) implementation of kGCN's batched
 * graph-convolution hot path.
 *
 * What this replaces.  The reference loads three TensorFlow custom-op libraries from the
 * working directory -- tf.load_op_library('./bspmm.so') (kgcn/bspmm_call.py:9,19),
 * './bconv.so' (kgcn/bconv_call.py:9,26), './batched.so' (kgcn/batched_call.py:9,19,30) --
 * and calls their ops Bspmm / Bconv / Bspmdt from GraphConv.call and GINAggregate.call
 * (kgcn/layers.py:77,88,102,435,445,458).  Neither sources nor binaries of those libraries are
 * in the reference tree, so there is no ABI to copy: the entry points below are what a binding
 * for this path needs, one per op contract plus the dense contraction, the
 * fused layer and the tiny reductions that close a forward+backward.
 *
 * Conventions
 *  - plain C, no C++/torch types; every pointer marked "device" is a HIP device pointer owned
 *    by the CALLER.  The library never allocates or frees device memory and keeps no global
 *    mutable state besides a thread-local error string; scratch space is passed in and sized
 *    by the *_workspace_bytes queries.
 *  - every launch is asynchronous on the given stream (hipStream_t passed as void*); no
 *    implicit synchronisation.  Entry points are re-entrant.
 *  - return value: 0 = ok, non-zero = error; text via kgcn_last_error().  No exception crosses
 *    the ABI.
 *  - all tensors are IEEE fp32, indices int32; every contraction accumulates in fp32.  Three evaluation routes,
 *    chosen per call from the shapes (kgcn_dense_mfma_products(kind, m, din, dout) reports the route of a dense call:
 *    1, 6 or 3 matrix-pipe products per fp32 product):
 *      (1) v_mfma_f32_32x32x2_f32 -- fp32 operands, exact products (kgcn_dense_* up to 128 output columns, the narrow
 *          50-wide layers, the cross-layer stack kernels);
 *      (2) bf16 x 3 -- an EXACT three-way bf16 split of every fp32 operand (v = p1 + p2 + p3, 8+8+8 significand bits) and
 *          the six products a_i b_j with i + j <= 4; the neglected terms are below 2^-23 |a b|, one fp32 rounding of the
 *          product (the fused GraphConv kernels kgcn_graphconv_*; kgcn_dense_* with more than 128 output columns below
 *          16,384 rows, with 65..96 input columns, and the 256 -> 50 layers);
 *      (3) f16 x 2 -- the wide-layer GEMMs of kgcn_dense_fwd_* / kgcn_dense_dx_dact_* / kgcn_dense_wgrad_* /
 *          kgcn_dense_bwd_* at >= 16,384 rows: each operand is scaled by an exact power of two per ROW (x, gradients) or
 *          per COLUMN (weights; both operands of a weight gradient) so that its largest magnitude lands in [2^14, 2^15),
 *          split into two f16 pieces v' = h + l (22 significand bits) and multiplied as l*H + h*L + h*H.
 *          Error of one dot product of length K:  <= 2^-23 sum_k |x_k w_k|  +  K * 2^-40 * max_k|x_k| * max_k|w_k|
 *          (the second term: elements more than 2^14 below their row's / column's maximum keep fewer than 22 bits --
 *          f16 denormal spacing).  For data within 2^14 of the row / column maximum this is fp32 arithmetic with one
 *          rounding per product; the edge suite (exponent spread, denormals, heavy-tailed rows) is
 *          tests/test_gpu_dense_edges.py, measured errors in profiles/r05_accuracy.json.
 *    Routes (2) and (3) never see reduced-precision INPUTS: the splits are of the caller's fp32 values.
 *
 * Adjacency layout in HBM ("batched CSR", one per adjacency channel).  T graphs, each an
 * [rows x cols] sparse matrix (uniform, = the reference's max_node_num padding,
 * kgcn/data_util.py:30-37, 410).  Row r of graph t owns entries rowptr[t*rows+r] ..
 * rowptr[t*rows+r+1]-1 of `cv`; entry e is the pair cv[2e] = column index LOCAL to the graph
 * (int32 bit pattern), cv[2e+1] = value (fp32 bit pattern).  Entries of a row keep the order
 * they had in the reference's COO (tf.SparseTensor) input; duplicates are simply repeated
 * entries (they accumulate, as in TF); rows without entries produce zeros; graphs without
 * entries (the dummy graphs padding a short batch, kgcn/feed.py:123-126) are legal.
 */
#ifndef KGCN_HIP_H_
#define KGCN_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 5): kgcn_csr_batch grew block_ptr / num_blocks / block_rows_max (round 4), kgcn_dense_dx_dact_gather_f32 gained
 * pooled_ld and kgcn_masked_sigmoid_ce_f32 gained pos_weight_per_task.  A binder built against version 1 must not call a
 * version-2 library: compare kgcn_abi_version() with the KGCN_HIP_ABI_VERSION it was compiled against AND
 * kgcn_csr_batch_size() with its own sizeof(kgcn_csr_batch) before the first call (kgcn_amd/_lib.py and
 * tests/abi_consumer.c do both).
 * Round 6 ADDED entry points only (no signature or layout changed, the version stays 2): kgcn_bconv_fanout_f32,
 * kgcn_copy2d_multi_f32 (+ kgcn_copy2d_job), kgcn_hbm_probe. */
#define KGCN_HIP_ABI_VERSION 2

/* Column index of the padding entries of a row-padded batch (see row_pad): they carry value 0 and
 * gather an all-zero row the fused kernels keep in LDS, so they contribute exactly 0 (never 0*inf). */
#define KGCN_PAD_COL 32

/* Non-finite inputs, per route (see "Conventions").  bf16 x 3 (fused GraphConv kernels, route 2): p1 + p2 + p3 == x bit for
 * bit for finite x; +-inf splits into (inf, NaN, NaN) and NaN into (NaN, NaN, NaN).  f16 x 2 (wide-layer GEMMs, route 3): a
 * row of x / a column of W (a column of either operand in a weight gradient) that holds +-inf or NaN gets a meaningless
 * scale and non-finite pieces, so EVERY output of that row / column is non-finite -- each of them depends on the non-finite
 * input.  In all routes: every output element that depends on a non-finite input comes out non-finite (NaN where fp32
 * arithmetic would give +-inf is possible), every other element is unaffected -- never a silently finite wrong value
 * (tests/test_gpu_parity.py::test_bf16_split_non_finite_inputs, tests/test_gpu_dense_edges.py).  Padding entries of the
 * row-padded layout gather an all-zero row with value 0, so they contribute exactly 0 (never 0 * inf). */

typedef struct kgcn_csr_batch {
  int32_t num_graphs;        /* T */
  int32_t rows;              /* M: rows per graph (padded, uniform) */
  int32_t cols;              /* K: columns per graph = rows of each rhs block */
  int32_t max_nnz_per_graph; /* max over graphs of stored entries (sizes the LDS staging) */
  int32_t row_pad;           /* 0: plain CSR.  4: every row holds a positive multiple of 4 entries,
                                padded with (col = KGCN_PAD_COL, value = 0) -- the layout the fused
                                GraphConv kernels read (mask-free 4-entry gathers); only those
                                kernels accept it */
  int32_t reserved_;
  int64_t nnz;               /* total stored entries (including padding entries) */
  const int32_t* rowptr;     /* device, [T*M + 1], absolute offsets into cv (in entries) */
  const int32_t* cv;         /* device, [2*nnz], interleaved (local col, fp32 value bits) */
  /* row_pad == 4 only (NULL otherwise): */
  const int32_t* slots;      /* device, [T*M]: per graph its M rows ordered by DEcreasing entry count,
                                slot = (offset of the row's first entry inside the graph) |
                                (entry count << 16) | (row index << 24).  A 4-row aggregation pass
                                reads 4 consecutive slots, so rows longer than 4 entries share the
                                first pass(es) and the others never branch */
  const int32_t* graph_ptr;  /* device, [T + 1]: graph_ptr[t] = rowptr[t*M] */
  /* Block structure of a block-diagonal one-graph container (the ragged-compact batches, kgcn_ragged_compact_csr); NULL /
   * 0 everywhere else.  The rows are cut into num_blocks ROW BLOCKS of whole molecules: block k = rows
   * [block_ptr[k], block_ptr[k+1]) holds the molecules whose first row lies in [k*block_rows, (k+1)*block_rows) -- at
   * most block_rows_max = block_rows + n_nodes - 1 rows, and no adjacency entry leaves its block.  The aggregation
   * kernels stage a block's rhs rows in LDS once and gather there (spmm.hip, spmm_block_kernel); an entry that does
   * leave its block (a container built by hand) is gathered from memory instead, so the result never depends on it. */
  const int32_t* block_ptr;  /* device, [num_blocks + 1], non-decreasing, block_ptr[num_blocks] = rows */
  int32_t num_blocks;
  int32_t block_rows_max;
} kgcn_csr_batch;

/* -- library info ------------------------------------------------------------------------ */
int kgcn_abi_version(void);
/* sizeof(kgcn_csr_batch) as THIS library was compiled: a caller whose own sizeof differs was built against another layout
 * of the descriptor (it grew in ABI version 2) and must not pass it in. */
int64_t kgcn_csr_batch_size(void);
/* Thread-local message of the last failing call on this thread ("" if none). */
const char* kgcn_last_error(void);
/* Name of the code-object architecture the kernels were built for ("gfx950"). */
const char* kgcn_build_arch(void);

/* -- Bspmm: batched sparse x dense ------------------------------------------------------- */
/* Replaces op `Bspmm` (kgcn/bspmm_call.py:16; backward use :45, kgcn/bconv_call.py:58,
 * kgcn/batched_call.py:60) and, because rhs/out are addressed as one strided tensor, also
 * `Bspmdt` (kgcn/batched_call.py:27: rhs is ONE [T*K, D] tensor -> rhs_graph_stride = K*rhs_ld).
 *   out[t] = beta * out[t] + A[t] @ rhs[t],   t = 0..T-1
 * rhs[t] = rhs + t*rhs_graph_stride is [K x d] with leading dimension rhs_ld (floats);
 * out[t] likewise [M x d].  beta is 0 (overwrite) or 1 (accumulate, used for the channel
 * add-n).  adjoint_a of the op is served by passing the batched CSR of A^T (the host packer
 * builds it once per batch); adjoint_b is a host-side transpose of rhs. */
int kgcn_bspmm_f32(const kgcn_csr_batch* a, const float* rhs, int64_t rhs_ld,
                   int64_t rhs_graph_stride, int32_t d, float* out, int64_t out_ld,
                   int64_t out_graph_stride, float beta, void* stream);

/* -- Bconv: fused multi-channel SpMM + channel add-n -------------------------------------- */
/* Replaces op `Bconv` (kgcn/bconv_call.py:18-23):  out[t] = sum_c A_c[t] @ rhs_c[t].
 * a_ch is a HOST array of num_channels descriptors; channel c's dense operand is
 * rhs + c*rhs_channel_stride (so the channels may be column slices of one [T*K, C*d] GEMM
 * output: rhs_channel_stride = d, rhs_ld = C*d). */
int kgcn_bconv_f32(const kgcn_csr_batch* a_ch, int32_t num_channels, const float* rhs,
                   int64_t rhs_ld, int64_t rhs_graph_stride, int64_t rhs_channel_stride,
                   int32_t d, float* out, int64_t out_ld, int64_t out_graph_stride,
                   void* stream);

/* -- gradient w.r.t. the sparse values ---------------------------------------------------- */
/* kgcn/bspmm_call.py:50-55:  dval[e] = sum_k grad[t][row_e,k] * rhs[t][col_e,k]  for every
 * stored entry e (same order as `cv`).  dval: device [nnz]. */
int kgcn_spmm_values_grad_f32(const kgcn_csr_batch* a, const float* grad, int64_t grad_ld,
                              int64_t grad_graph_stride, const float* rhs, int64_t rhs_ld,
                              int64_t rhs_graph_stride, int32_t d, float* dval, void* stream);

/* -- dense contraction (GraphDense, and the X.W part of the unfused GraphConv) ------------- */
/* kgcn/layers.py:255-262 (Keras Dense on reshape(X,[-1,Din])) and :99-100 / :112:
 *   y[m, dout] = x[m, din] @ w + bias        (trans_w = 0: w is [din x dout], ld w_ld)
 *   y[m, dout] = x[m, din] @ w^T + bias      (trans_w = 1: w is [dout x din], ld w_ld)
 * bias may be NULL.  fp32 MFMA. */
int kgcn_dense_fwd_f32(const float* x, int64_t m, int32_t din, int64_t x_ld, const float* w,
                       int64_t w_ld, int32_t trans_w, const float* bias, float* y,
                       int32_t dout, int64_t y_ld, void* stream);

/* Weight/bias gradients of the contraction:  dw[din,dout] = x^T @ dy,  dbias[dout] = colsum(dy)
 * (the reduction over all m = B*N rows: TF MatMul/BiasAdd gradients summed by AddN over the
 * graphs).  Deterministic two-stage reduction through `workspace`.
 * dw (ld = dout) and dbias may each be NULL. */
int64_t kgcn_dense_wgrad_workspace_bytes(int64_t m, int32_t din, int32_t dout);
int kgcn_dense_wgrad_f32(const float* x, int64_t x_ld, const float* dy, int64_t dy_ld, int64_t m,
                         int32_t din, int32_t dout, float* dw, float* dbias, void* workspace,
                         int64_t workspace_bytes, void* stream);

/* -- fused GraphConv layer (single adjacency channel) -------------------------------------- */
/* kgcn/layers.py:64-116, all four branches compute
 *   out[t] = A[t] @ (x[t] @ w + bias)            x[t]: [N x din], out[t]: [N x dout]
 * Forward in ONE kernel (x tile -> LDS, fp32 MFMA, aggregation out of LDS; X.W never touches
 * HBM).  Supported fused shapes are reported by kgcn_graphconv_fused_supported(); other
 * shapes and multi-channel layers use kgcn_dense_* + kgcn_bconv_f32. */
/* a / at of the fused entry points must be row-padded batches (row_pad == 4). */
int kgcn_graphconv_fused_supported(int32_t n_nodes, int32_t din, int32_t dout,
                                   int32_t max_nnz_per_graph);
int kgcn_graphconv_fwd_f32(const kgcn_csr_batch* a, const float* x, const float* w,
                           const float* bias, int32_t din, int32_t dout, float* out,
                           void* stream);
/* Backward in ONE kernel + a small deterministic reduction:
 *   dfw[t] = A[t]^T @ dout[t];  dx[t] = dfw[t] @ w^T;  dw = sum_t x[t]^T dfw[t];
 *   dbias = sum_t colsum(dfw[t])          (SURVEY 3.3 / kgcn/bspmm_call.py:45)
 * at = batched CSR of A^T.  dx may be NULL (first layer). */
int64_t kgcn_graphconv_bwd_workspace_bytes(int32_t num_graphs, int32_t din, int32_t dout);
int kgcn_graphconv_bwd_f32(const kgcn_csr_batch* at, const float* x, const float* w,
                           const float* dout_grad, int32_t din, int32_t dout, float* dx,
                           float* dw, float* dbias, void* workspace, int64_t workspace_bytes,
                           void* stream);

/* -- GINAggregate -------------------------------------------------------------------------- */
/* kgcn/layers.py:461-472:  out[t] = sum_c (eps[c] * x[t] + A_c[t] @ x[t]).
 * eps: device [num_channels] or NULL (the accelerated branches :429-460 drop the eps term). */
int kgcn_gin_aggregate_f32(const kgcn_csr_batch* a_ch, int32_t num_channels, const float* x,
                           int32_t d, const float* eps, float* out, void* stream);

/* -- mini-batch assembly on the device ---------------------------------------------------------- */
/* kgcn/feed.py:112-126 without the host: `src` is the container of the WHOLE dataset (resident in HBM),
 * sel[t] (device, int32[num_sel]) the dataset index of batch graph t or -1 for an empty dummy graph.
 * Writes the batch container: dst_rowptr[num_sel*rows + 1], dst_cv[2 * total entries] (capacity in
 * ENTRIES; exact total or worst case num_sel * max_nnz_per_graph), dst_graph_ptr[num_sel + 1]
 * (graph_ptr[num_sel] = total entries) and, for a row-padded source, dst_slots[num_sel*rows] (dummy
 * graphs become rows of 4 padding entries).  sel values must be < src->num_graphs (not checked: device
 * data).  workspace >= kgcn_csr_gather_workspace_bytes(num_sel). */
int64_t kgcn_csr_gather_workspace_bytes(int32_t num_sel);
int kgcn_csr_gather_graphs(const kgcn_csr_batch* src, const int32_t* sel, int32_t num_sel,
                           int32_t* dst_rowptr, int32_t* dst_cv, int64_t dst_cv_capacity,
                           int32_t* dst_slots, int32_t* dst_graph_ptr, void* workspace,
                           int64_t workspace_bytes, void* stream);

/* The whole mini-batch in TWO launches: up to KGCN_ASSEMBLE_MAX_CSR containers of the same dataset (A, A^T, their row-padded
 * copies: each as in kgcn_csr_gather_graphs, dst_graph_ptr[num_sel + 1] included) and up to KGCN_ASSEMBLE_MAX_TABLES per-graph
 * float tables (features [G, N * F], labels, masks ... -- what kgcn/feed.py:112-133 slices per batch on the host):
 * table_out[k][t, :] = table[k][sel[t], :], a row of zeros for sel[t] = -1.  dst_cv_capacity must cover the worst case
 * num_sel * max(max_nnz_per_graph, 4 * rows of a row-padded source): the selection is device data.
 * workspace >= kgcn_batch_assemble_workspace_bytes(num_sel). */
#define KGCN_ASSEMBLE_MAX_CSR 4
#define KGCN_ASSEMBLE_MAX_TABLES 6
typedef struct kgcn_assemble_plan {
  int32_t num_csr, num_tables;
  const kgcn_csr_batch* src[KGCN_ASSEMBLE_MAX_CSR];
  int32_t* dst_rowptr[KGCN_ASSEMBLE_MAX_CSR];
  int32_t* dst_cv[KGCN_ASSEMBLE_MAX_CSR];
  int64_t dst_cv_capacity[KGCN_ASSEMBLE_MAX_CSR];
  int32_t* dst_slots[KGCN_ASSEMBLE_MAX_CSR];
  int32_t* dst_graph_ptr[KGCN_ASSEMBLE_MAX_CSR];
  const float* table[KGCN_ASSEMBLE_MAX_TABLES];
  float* table_out[KGCN_ASSEMBLE_MAX_TABLES];
  int64_t row_floats[KGCN_ASSEMBLE_MAX_TABLES];
} kgcn_assemble_plan;
int64_t kgcn_batch_assemble_workspace_bytes(int32_t num_sel);
int kgcn_batch_assemble(const kgcn_assemble_plan* plan, const int32_t* sel, int32_t num_sel, void* workspace,
                        int64_t workspace_bytes, void* stream);

/* -- GraphMaxPooling ------------------------------------------------------------------------ */
/* kgcn/layers.py:122-150 (one adjacency channel per call; beta = 1 accumulates the channel add-n):
 *   out[t,i,k] = beta*out[t,i,k] + max_j dense(A[t] .* x[t][:,k])[i,j]
 * = the maximum of a_ij * x[t][j,k] over the stored entries of row i, with 0 as a candidate unless
 * the row stores all `cols` entries (absent entries of the densified row are zeros).  Entries of a row must
 * be unique (the reference densifies with tf.sparse_tensor_to_dense, which rejects repeated indices). */
int kgcn_graph_maxpool_fwd_f32(const kgcn_csr_batch* a, const float* x, int32_t d, float* out,
                               float beta, void* stream);
/* Gradient as TF builds it (reduce_max splits the gradient equally among ALL maximal elements of
 * the densified row, implicit zeros included):  dx[t,j,k] = beta*dx + sum_i a_ij * share(i,j,k).
 * at = batched CSR of A^T (same value per entry); workspace >= kgcn_graph_maxpool_bwd_workspace_bytes. */
int64_t kgcn_graph_maxpool_bwd_workspace_bytes(int32_t num_graphs, int32_t rows, int32_t d);
int kgcn_graph_maxpool_bwd_f32(const kgcn_csr_batch* a, const kgcn_csr_batch* at, const float* x,
                               const float* dout_grad, int32_t d, float* dx, float beta,
                               void* workspace, int64_t workspace_bytes, void* stream);

/* -- GAT ------------------------------------------------------------------------------------ */
/* kgcn/layers.py:477-542, one adjacency channel per call (beta = 1 accumulates the channel add-n).  Only
 * the PATTERN of `a` is used (the reference ignores the values, :517-520).  weight_a: device [2*d]
 * (the layer's `weight_a{i}` [2d,1]).  With t_e = x[col_e].wa[0:d] + x[row_e].wa[d:2d], E_e = exp(leaky_relu(
 * t_e, 0.2)), denom[i] = sum over row i of E, alpha_e = E_e / (denom[col_e] + 1e-10) (column-indexed, as the
 * reference gathers it):   out[t,i,:] = beta*out + sigmoid(sum_{e in row i} alpha_e x[t, col_e, :]). */
int64_t kgcn_gat_workspace_bytes(int32_t num_graphs, int32_t n_nodes, int32_t d);
int kgcn_gat_fwd_f32(const kgcn_csr_batch* a, const float* x, int32_t d, const float* weight_a, float* out,
                     float beta, void* workspace, int64_t workspace_bytes, void* stream);
/* dx = beta*dx + d loss / d x,  dweight_a [2*d] overwritten (deterministic two-stage reduction).
 * at = batched CSR of the transposed pattern. */
int kgcn_gat_bwd_f32(const kgcn_csr_batch* a, const kgcn_csr_batch* at, const float* x, int32_t d,
                     const float* weight_a, const float* dout_grad, float* dx, float beta, float* dweight_a,
                     void* workspace, int64_t workspace_bytes, void* stream);

/* -- decoders: per-graph weighted Gram matrix ------------------------------------------------ */
/* GraphDecoderInnerProd (kgcn/layers.py:268-282, w = NULL), GraphDecoderDistMult (:285-305), DistMult.call
 * (:347-354, one relation channel per call):  out[t,i,j] = sum_k w[k] x[t,i,k] x[t,j,k].
 * x [T, N, d], w [d] or NULL, out [T, N, N].  Backward: dx = beta*dx + w * ((g + g^T) x); dw [d] overwritten
 * (NULL to skip; workspace >= kgcn_gram_workspace_bytes(d), deterministic two-stage sum). */
int64_t kgcn_gram_workspace_bytes(int32_t d);
int kgcn_gram_fwd_f32(const float* x, int32_t num_graphs, int32_t n_nodes, int32_t d, const float* w, float* out,
                      void* stream);
int kgcn_gram_bwd_f32(const float* x, int32_t num_graphs, int32_t n_nodes, int32_t d, const float* w,
                      const float* dout_grad, float* dx, float beta, float* dw, void* workspace,
                      int64_t workspace_bytes, void* stream);

/* -- GraphGather ---------------------------------------------------------------------------- */
/* kgcn/layers.py:163-164: out[b, :] = sum_n x[b, n, :] (padding rows included). */
int kgcn_graph_gather_fwd_f32(const float* x, int64_t batch, int32_t n_nodes, int32_t d,
                              float* out, void* stream);
/* the same into a column block of a wider tensor: out[b * out_ld + c] (tf.concat of several read-outs, model_gin.py:61, without
 * the concatenation pass) */
int kgcn_graph_gather_fwd_ld_f32(const float* x, int64_t batch, int32_t n_nodes, int32_t d, float* out, int64_t out_ld,
                                 void* stream);
/* dx[b, n, :] = dout[b, :] */
int kgcn_graph_gather_bwd_f32(const float* dout_grad, int64_t batch, int32_t n_nodes, int32_t d,
                              float* dx, void* stream);

/* dx[b, n, :] = dx_in[b, n, :] + dout[b, :] -- the gradient of a tensor that feeds a GraphGather read-out AND the next layer
 * (example_model/model_gin.py:45-60) in one pass; d % 4 == 0, 16-byte aligned tensors; dx may alias dx_in. */
int kgcn_graph_gather_bwd_add_f32(const float* dout_grad, const float* dx_in, int64_t batch, int32_t n_nodes, int32_t d,
                                  float* dx, void* stream);

/* -- small reductions used by the layer gradients ------------------------------------------- */
/* out[0] = sum_i a[i]*b[i]  (d eps of GINAggregate).  workspace >= kgcn_dot_workspace_bytes(). */
int64_t kgcn_dot_workspace_bytes(int64_t n);
int kgcn_dot_f32(const float* a, const float* b, int64_t n, float* out, void* workspace,
                 int64_t workspace_bytes, void* stream);

/* -- activations fused into the producing kernels -------------------------------------------- */
/* The reference's models wrap the layers in tf.sigmoid / tf.nn.relu / tf.tanh (example_model/model.py:43-53,
 * model_multitask.py:52-60, sparse.py:76, model_gin.py:47-50): one elementwise TF op -- one read and one write of the
 * activation tensor -- per layer and direction.  Here the activation rides in the epilogue of the kernel that produces
 * the tensor, and its derivative (expressed in the layer OUTPUT: sigmoid a(1-a), relu a > 0, tanh 1 - a^2) in the
 * prologue of the adjoint aggregation. */
#define KGCN_ACT_NONE 0
#define KGCN_ACT_SIGMOID 1
#define KGCN_ACT_RELU 2
#define KGCN_ACT_TANH 3
/* kgcn_bconv_f32 followed by act:  out[t] = act( sum_c A_c[t] @ rhs_c[t] ) */
int kgcn_bconv_act_f32(const kgcn_csr_batch* a_ch, int32_t num_channels, const float* rhs, int64_t rhs_ld,
                       int64_t rhs_graph_stride, int64_t rhs_channel_stride, int32_t d, float* out, int64_t out_ld,
                       int64_t out_graph_stride, int32_t act, void* stream);

/* The adjoint of Bconv for ALL channels from one read of the gradient (kgcn/bconv_call.py:45-53: addn_grad fans the incoming
 * gradient out to every channel):   out_c[t] = A_c[t]^T (grad[t] (.) act'(act_out[t]))   for c = 0 .. num_channels - 1,
 * out_c = out + c * out_channel_stride.  at_ch: the TRANSPOSED containers (A_c^T) of one batch shape; act / act_out as in
 * kgcn_bspmm_dact_f32 (act = KGCN_ACT_NONE: act_out may be NULL).  Equivalent to num_channels calls of kgcn_bspmm_dact_f32 with
 * beta = 0 (bit for bit: the same sums in the same order). */
int kgcn_bconv_fanout_f32(const kgcn_csr_batch* at_ch, int32_t num_channels, const float* grad, const float* act_out, int64_t ld,
                          int64_t graph_stride, int32_t d, int32_t act, float* out, int64_t out_ld, int64_t out_graph_stride,
                          int64_t out_channel_stride, void* stream);
/* backward of that layer through one channel (pass the A^T container):
 *   out[t] = beta*out[t] + A[t] @ ( grad[t] (.) act'(act_out[t]) )      grad, act_out: same layout (ld, graph stride) */
int kgcn_bspmm_dact_f32(const kgcn_csr_batch* a, const float* grad, const float* act_out, int64_t ld,
                        int64_t graph_stride, int32_t d, int32_t act, float* out, int64_t out_ld,
                        int64_t out_graph_stride, float beta, void* stream);
/* kgcn_dense_fwd_f32 followed by act:  y = act(x @ w(^T) + bias) */
int kgcn_dense_fwd_act_f32(const float* x, int64_t m, int32_t din, int64_t x_ld, const float* w, int64_t w_ld,
                           int32_t trans_w, const float* bias, float* y, int32_t dout, int64_t y_ld, int32_t act,
                           void* stream);
/* The same contraction with a caller-provided workspace of kgcn_dense_fwd_workspace_bytes(din, dout) bytes (0 for
 * layers that do not use one): for wide layers (dout > 128, din >= 192: the 256-wide layers of example_model/
 * model_multitask.py:51-57 and sparse.py:30) the weight operand is split once per call into a bf16 fragment table in
 * that workspace (csrc/wtable.hip) which the GEMM reads instead of splitting W in every workgroup.
 * workspace == NULL: kgcn_dense_fwd_act_f32. */
int64_t kgcn_dense_fwd_workspace_bytes(int32_t din, int32_t dout);
int kgcn_dense_fwd_ws_f32(const float* x, int64_t m, int32_t din, int64_t x_ld, const float* w, int64_t w_ld,
                          int32_t trans_w, const float* bias, float* y, int32_t dout, int64_t y_ld, int32_t act,
                          void* workspace, int64_t workspace_bytes, void* stream);
/* Fragment tables ahead of time: ONE launch splits every listed operand (a training step: W of each wide layer for the forward,
 * W^T for d input -- 7-9 separate 5 us launches otherwise); job = the operand of a contraction over k with n output columns,
 * i.e. w [k x n] (trans_w = 0) or [n x k] (trans_w = 1), table = kgcn_dense_fwd_workspace_bytes(k, n) bytes.  The *_tab entry
 * points take such a READY table (valid as long as w is unchanged) instead of a workspace; table == NULL: no table. */
#define KGCN_WTABLE_MAX_JOBS 16
typedef struct kgcn_wtable_job {
  const float* w;
  int64_t w_ld;
  int32_t trans_w, k, n;
  int32_t k_w;               /* stacked operand (extra_row != NULL, trans_w = 0): rows 0 .. k_w - 1 come from w */
  void* table;
  const float* extra_row;    /* NULL: plain operand.  Else the operand is [w (k_w rows); extra_row (1 row, n floats); zeros] of k rows:
                                [W; bias; 0] of an aggregate-first GraphConv -- A (X W + 1 b) = (A [X | 1]) [W; b], kgcn/layers.py:112-113 --
                                split without a concatenation pass (ABI version 2) */
} kgcn_wtable_job;
int kgcn_wtable_split_multi(const kgcn_wtable_job* jobs, int32_t num_jobs, void* stream);
int kgcn_dense_fwd_tab_f32(const float* x, int64_t m, int32_t din, int64_t x_ld, const float* w, int64_t w_ld,
                           int32_t trans_w, const float* bias, float* y, int32_t dout, int64_t y_ld, int32_t act,
                           const void* table, int64_t table_bytes, void* stream);
/* Backward of y = act(x @ w + bias) with respect to x (w: [din x dout], ld w_ld):
 *   dpre = grad (.) act'(act_out)   written to dpre (same layout as grad; must not alias it) -- the operand of
 *                                   kgcn_dense_wgrad_f32 for dw / dbias,
 *   dx   = dpre @ w^T               [m x din].
 * For wide layers with a workspace of kgcn_dense_fwd_workspace_bytes(dout, din) bytes both happen in ONE pass of the
 * GEMM (the derivative is applied while the gradient rows are staged); otherwise two launches. */
int kgcn_dense_dx_dact_f32(const float* grad, const float* act_out, int64_t m, int32_t dout, int64_t ld, const float* w,
                           int64_t w_ld, int32_t din, float* dx, int64_t dx_ld, int32_t act, float* dpre,
                           void* workspace, int64_t workspace_bytes, void* stream);
int kgcn_dense_dx_dact_tab_f32(const float* grad, const float* act_out, int64_t m, int32_t dout, int64_t ld, const float* w,
                               int64_t w_ld, int32_t din, float* dx, int64_t dx_ld, int32_t act, float* dpre, const void* table,
                               int64_t table_bytes, void* stream);
/* The same for a layer whose output was read out by GraphGather (kgcn/layers.py:163-164) and possibly handed on as well
 * (example_model/model_gin.py:45-60): the incoming gradient of node row r is  grad[r] + pooled_grad[r / n_nodes]  (grad == NULL: the
 * read-out alone).  The broadcast of d pooled over the node rows is formed while the GEMM stages its rows -- it never exists in
 * HBM (a [rows x dout] tensor written by kgcn_graph_gather_bwd(_add)_f32 and read back otherwise).  Wide layers only:
 * kgcn_dense_dx_dact_gather_supported(m, din, dout); table: kgcn_dense_fwd_workspace_bytes(dout, din) bytes, already split
 * (table_ready != 0, kgcn_wtable_split_multi) or a workspace the call splits into. */
int kgcn_dense_dx_dact_gather_supported(int64_t m, int32_t din, int32_t dout);
/* pooled_ld: row stride of pooled_grad [graphs, dout] in floats (a multiple of 4, >= dout: the gradient of a column block of a
 * wider read-out tensor -- model_gin.py:61 concatenates the read-outs of its blocks -- is used where it lies) */
int kgcn_dense_dx_dact_gather_f32(const float* grad, const float* pooled_grad, int64_t pooled_ld, int32_t n_nodes,
                                  const float* act_out, int64_t m, int32_t dout, int64_t ld, const float* w, int64_t w_ld,
                                  int32_t din, float* dx, int64_t dx_ld, int32_t act, float* dpre, void* table,
                                  int64_t table_bytes, int32_t table_ready, void* stream);
/* ONE-PASS backward of a wide dense layer (round 5, gemmb.hip): the whole backward of y = act(x @ w + bias) -- Keras Dense inside
 * GraphDense (kgcn/layers.py:248,260) and the X.W part of GraphConv (:99-100, :112) at the widths of example_model/model_gin.py:45-54
 * and example_model/model_multitask.py:51-57 -- from ONE sweep over (grad, act_out, x):
 *   dpre = (grad [+ pooled_grad[row / n_nodes]]) (.) act'(act_out)      (act == KGCN_ACT_NONE: dpre = grad, act_out may be NULL)
 *   dx = dpre @ w^T      dw = x^T @ dpre      dbias = colsum(dpre)      (dbias may be NULL)
 * The d pre-activation tensor is never written (kgcn_dense_dx_dact_f32 + kgcn_dense_wgrad_f32 write it and read it back: six
 * passes over [m, 256] tensors instead of four).  grad may be NULL when pooled_grad is given (the layer output was only read out
 * by GraphGather); pooled_grad as in kgcn_dense_dx_dact_gather_f32, with n_nodes >= 8 (smaller graphs: the two-call route).
 * ld: row stride of grad and act_out.
 * kgcn_dense_bwd_supported(m, din, dout): dout == 256, 128 < din <= 256 (a multiple of 4), m >= 16,384.  table / table_ready: the
 * fragment tables of w^T as for kgcn_dense_dx_dact_gather_f32 (kgcn_dense_fwd_workspace_bytes(dout, din) bytes); workspace >=
 * kgcn_dense_wgrad_workspace_bytes(m, din, dout) (128 partials, fixed-order second stage, deferrable: kgcn_reduce_defer).
 * Arithmetic: route 3 of "Conventions" (f16 x 2) with a row scale on dpre and a counter-scaled, per-column online scale on x. */
int kgcn_dense_bwd_supported(int64_t m, int32_t din, int32_t dout);
int kgcn_dense_bwd_f32(const float* grad, const float* pooled_grad, int64_t pooled_ld, int32_t n_nodes, const float* act_out,
                       int32_t act, int64_t ld, const float* x, int64_t x_ld, int64_t m, int32_t din, int32_t dout,
                       const float* w, int64_t w_ld, float* dx, int64_t dx_ld, float* dw, float* dbias, void* table,
                       int64_t table_bytes, int32_t table_ready, void* workspace, int64_t workspace_bytes, void* stream);
/* ... and its form for a layer whose d-input product is only needed for an inner product (kgcn_dense_dx_dact_dot_f32's case: the
 * activated wide GraphDense behind a GINAggregate whose input needs no gradient, example_model/model_gin.py:45-50):
 *   dot_out[0] = < dpre @ w^T , dotx >     dw = x^T @ dpre     dbias = colsum(dpre)        dpre = grad (.) act'(act_out)
 * from one sweep over (grad, act_out, x, dotx); nothing of size [m, .] is written.  Shapes, table and workspace as for
 * kgcn_dense_bwd_f32. */
int kgcn_dense_bwd_dot_f32(const float* grad, const float* act_out, int32_t act, int64_t ld, const float* x, int64_t x_ld, int64_t m,
                           int32_t din, int32_t dout, const float* w, int64_t w_ld, const float* dotx, int64_t dotx_ld, float* dw,
                           float* dbias, float* dot_out, void* table, int64_t table_bytes, int32_t table_ready, void* workspace,
                           int64_t workspace_bytes, void* stream);
/* The same dX contraction where the product is only needed for an inner product (d epsilon of a GINAggregate, kgcn/layers.py:469
 * <d out, x>, in front of an activated wide layer whose input needs no gradient -- the first block of example_model/model_gin.py):
 *   dot_out[0] = < (grad (.) act'(act_out)) @ w^T , dotx >       dotx [m, din] (row stride dotx_ld, 16-byte aligned rows)
 * d pre-activation is written to dpre as in kgcn_dense_dx_dact_f32; the [m, din] product never exists in HBM.
 * kgcn_dense_dx_dact_dot_supported(m, din, dout); workspace >= kgcn_dense_dx_dact_dot_workspace_bytes(m, din) (one partial per
 * workgroup, fixed-order second stage); table as for kgcn_dense_dx_dact_gather_f32. */
int kgcn_dense_dx_dact_dot_supported(int64_t m, int32_t din, int32_t dout);
int64_t kgcn_dense_dx_dact_dot_workspace_bytes(int64_t m, int32_t din);
int kgcn_dense_dx_dact_dot_f32(const float* grad, const float* act_out, int64_t m, int32_t dout, int64_t ld, const float* w,
                               int64_t w_ld, int32_t din, const float* dotx, int64_t dotx_ld, int32_t act, float* dpre,
                               void* table, int64_t table_bytes, int32_t table_ready, float* dot_out, void* workspace,
                               int64_t workspace_bytes, void* stream);
/* stand-alone forms: y = act(x) over n floats; dpre = grad (.) act'(act_out) (dpre may alias grad) */
int kgcn_act_fwd_f32(const float* x, int64_t n, int32_t act, float* y, void* stream);
int kgcn_act_bwd_f32(const float* act_out, const float* grad, int64_t n, int32_t act, float* dpre, void* stream);

/* -- GraphBatchNormalization (kgcn/layers.py:170-220) --------------------------------------- */
/* Keras BatchNormalization over the VALID node rows of a padded batch x [T, N, D]: rows n < enabled[t] of graph t
 * (enabled == NULL: every row; the reference gathers those rows, normalises the stacked [rows, D] matrix per feature and
 * pads the result back with zeros, :196-210 / :211-216).
 *   kgcn_graph_bn_stats_f32   mean[c], var[c] (population variance, two passes like tf.nn.moments) over the valid rows --
 *                             the batch statistics of training mode
 *   kgcn_graph_bn_apply_f32   y = gamma (x - mean) / sqrt(var + eps) + beta on valid rows, 0 on padding rows; with the
 *                             moving statistics this is inference mode (the TF1 default phase)
 *   kgcn_graph_bn_bwd_f32     dgamma = sum g xhat, dbeta = sum g, and dx (may be NULL):
 *                             training != 0: gamma rstd (g - dbeta/n - xhat dgamma/n);  training == 0: gamma rstd g
 * mean / var / gamma / beta / dgamma / dbeta: device [D].  workspace: kgcn_graph_bn_workspace_bytes(D) bytes; reductions
 * are deterministic (per-workgroup partials, fixed-order second stage). */
int64_t kgcn_graph_bn_workspace_bytes(int32_t d);
int kgcn_graph_bn_stats_f32(const float* x, int64_t graphs, int32_t n_nodes, int32_t d, const int32_t* enabled,
                            float* mean, float* var, void* workspace, int64_t workspace_bytes, void* stream);
int kgcn_graph_bn_apply_f32(const float* x, int64_t graphs, int32_t n_nodes, int32_t d, const int32_t* enabled,
                            const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                            float* y, void* stream);
int kgcn_graph_bn_bwd_f32(const float* x, const float* grad, int64_t graphs, int32_t n_nodes, int32_t d,
                          const int32_t* enabled, const float* mean, const float* var, const float* gamma, float eps,
                          int32_t training, float* dx, float* dgamma, float* dbeta, void* workspace,
                          int64_t workspace_bytes, void* stream);

/* act(bn(x)) in one pass (the padding rows become act(0), as tf.sigmoid(bn(...)) makes them) and its backward: the incoming
 * gradient is multiplied by act'(act_out) while it is read (act_out = the output of kgcn_graph_bn_apply_act_f32). */
int kgcn_graph_bn_apply_act_f32(const float* x, int64_t graphs, int32_t n_nodes, int32_t d, const int32_t* enabled,
                                const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                                int32_t act, float* y, void* stream);
int kgcn_graph_bn_bwd_dact_f32(const float* x, const float* grad, const float* act_out, int32_t act, int64_t graphs,
                               int32_t n_nodes, int32_t d, const int32_t* enabled, const float* mean, const float* var,
                               const float* gamma, float eps, int32_t training, float* dx, float* dgamma, float* dbeta,
                               void* workspace, int64_t workspace_bytes, void* stream);

/* -- COO -> batched CSR on the device (the feed side: kgcn/feed.py:112-126 assembles the same triples in Python) ------ */
/* (graph, row, col, val)[nnz]: device arrays in ANY order (val NULL = all ones) -> rowptr_out [T*R + 1], cv_out [nnz]
 * (int2: column, fp32 value bits) with R = rows and the entries of a row in feed order (stable sort: duplicates stay and
 * are summed in feed order, like the host packer), or, transposed != 0, the container of A^T: R = cols, entries of a
 * transposed row in ascending original row.  perm_out (may be NULL): [nnz] input position of every stored entry.
 * stats_out: device int32[2] = {max stored entries per graph, number of out-of-range triples (must be 0)}.
 * workspace >= kgcn_coo_pack_workspace_bytes(nnz, T, rows, cols). */
int64_t kgcn_coo_pack_workspace_bytes(int64_t nnz, int32_t num_graphs, int32_t rows, int32_t cols);
int kgcn_coo_pack_f32(const int32_t* graph, const int32_t* row, const int32_t* col, const float* val, int64_t nnz,
                      int32_t num_graphs, int32_t rows, int32_t cols, int32_t transposed, int32_t* rowptr_out,
                      void* cv_out, int32_t* perm_out, int32_t* stats_out, void* workspace, int64_t workspace_bytes,
                      void* stream);
/* plain container (row_pad = 0, square, rows <= KGCN_PAD_COL) -> the row-padded layout of the fused GraphConv kernels:
 * rowptr4_out [T*M + 1], cv4_out [cv4_capacity entries; nnz + 4*T*M always suffices], slots_out [T*M],
 * graph_ptr_out [T + 1].  stats_out: device int32[3] = {max padded entries per graph, total padded entries,
 * number of violations (row longer than 252 entries, graph longer than 65535, capacity exceeded; must be 0)}.
 * workspace >= kgcn_csr_pad4_workspace_bytes(T, M). */
int64_t kgcn_csr_pad4_workspace_bytes(int32_t num_graphs, int32_t rows);
int kgcn_csr_pad4(const kgcn_csr_batch* a, int32_t* rowptr4_out, void* cv4_out, int64_t cv4_capacity,
                  int32_t* slots_out, int32_t* graph_ptr_out, int32_t* stats_out, void* workspace,
                  int64_t workspace_bytes, void* stream);

/* -- ragged-compact batches: the layer stack on the VALID node rows only ------------------------------------------------ */
/* The reference pads every graph to max_node_num rows (kgcn/data_util.py:30-37, feed.py:127-133); its ragged layers
 * gather the first enabled_node_nums[b] rows of every graph, compute on the stacked [sum n_b, D] matrix and pad back
 * (GraphDense kgcn/layers.py:243-254, GraphBatchNormalization :196-210).  These entry points build that stacked layout
 * ONCE per batch so that every layer runs on it:
 *   rows [graph_ptr[t], graph_ptr[t+1])  the n_t valid rows of batch graph t
 *   rows [R, capacity_rows)              padding representatives: zero features, no adjacency entries; they go through
 *                                        the layers like the padded rows of the reference's layout, so any one of them
 *                                        holds the value ALL padded rows of the padded formulation have (GraphConv -> 0,
 *                                        act(0), un-ragged GraphDense -> one constant row, ragged BN -> 0)
 *   adjacency                            one block-diagonal [capacity_rows x capacity_rows] CSR (num_graphs = 1),
 *                                        accepted by every kgcn_bspmm / kgcn_bconv entry point
 * capacity_rows >= R + 1 is chosen by the caller (a fixed capacity lets a captured hipGraph replay batches of varying R).
 *
 * kgcn_ragged_plan: n_t = clamp(sizes[g], 0, rows) and e_t = stored entries in rows [0, n_t) of graph g = sel[t]
 * (sel NULL: g = t; sel[t] < 0: an empty dummy graph); graph_ptr / entry_ptr [num_sel + 1] = their exclusive scans
 * (graph_ptr[num_sel] = R).  src: square batched CSR of the source graphs (the batch itself or the whole dataset).
 * workspace >= kgcn_ragged_workspace_bytes(num_sel). */
int64_t kgcn_ragged_workspace_bytes(int32_t num_sel);
int kgcn_ragged_plan(const kgcn_csr_batch* src, const int32_t* sizes, const int32_t* sel, int32_t num_sel,
                     int32_t* graph_ptr, int32_t* entry_ptr, void* workspace, int64_t workspace_bytes, void* stream);
/* dst_rowptr [capacity_rows + 1], dst_cv [dst_cv_capacity entries] <- the block-diagonal container of the selected graphs
 * (also for A^T: pass the transposed source container with the SAME plan).  status (may be NULL): device int32, += the
 * number of stored entries that do not fit the compact layout (rows or columns >= n_t, plan mismatch, capacity exceeded);
 * must read 0. */
int kgcn_ragged_compact_csr(const kgcn_csr_batch* src, const int32_t* sel, int32_t num_sel, const int32_t* graph_ptr,
                            const int32_t* entry_ptr, int32_t capacity_rows, int32_t* dst_rowptr, int32_t* dst_cv,
                            int64_t dst_cv_capacity, int32_t* status, void* stream);
/* The same for a channel's A and A^T (one plan, two destination containers) in ONE launch. */
int kgcn_ragged_compact_csr_pair(const kgcn_csr_batch* src, const kgcn_csr_batch* src_t, const int32_t* sel, int32_t num_sel,
                                 const int32_t* graph_ptr, const int32_t* entry_ptr, int32_t capacity_rows, int32_t* dst_rowptr,
                                 int32_t* dst_cv, int64_t dst_cv_capacity, int32_t* dst_t_rowptr, int32_t* dst_t_cv,
                                 int64_t dst_t_cv_capacity, int32_t* status, void* stream);
/* The row blocks of a ragged-compact batch (kgcn_csr_batch.block_ptr): block_ptr [kgcn_ragged_num_blocks(capacity_rows) + 1]
 * from the plan's graph_ptr [num_sel + 1]; blocks past the valid rows cover the padding rows in steps of
 * KGCN_RAGGED_BLOCK_ROWS.  Every block holds at most KGCN_RAGGED_BLOCK_ROWS + n_nodes - 1 rows. */
#define KGCN_RAGGED_BLOCK_ROWS 32
int32_t kgcn_ragged_block_rows(void);                 /* KGCN_RAGGED_BLOCK_ROWS of the library that was loaded */
int32_t kgcn_ragged_num_blocks(int32_t capacity_rows);
int kgcn_ragged_blocks(const int32_t* graph_ptr, int32_t num_sel, int32_t capacity_rows, int32_t* block_ptr, void* stream);
/* dst [capacity_rows, d] <- the valid rows of src [num_source_graphs, n_nodes, d] (feed.py:127-133 layout), zeros on the
 * rows >= R. */
int kgcn_ragged_compact_rows_f32(const float* src, const int32_t* sel, int32_t num_sel, int32_t n_nodes, int32_t d,
                                 const int32_t* graph_ptr, int32_t capacity_rows, float* dst, void* stream);
/* The same rows as [x | 1 | 0 ...]: dst [capacity_rows, dst_ld], dst_ld >= d + 1; column d of EVERY row <- 1, columns d + 1 .. <- 0
 * (= kgcn_ragged_compact_rows_f32 followed by kgcn_augment_ones_f32 in one pass: the operand of the aggregate-first GraphConv,
 * A (X W + 1 b) = (A [X | 1]) [W ; b], kgcn/layers.py:99-113).  n_nodes * dst_ld < 2^22. */
int kgcn_ragged_compact_rows_aug_f32(const float* src, const int32_t* sel, int32_t num_sel, int32_t n_nodes, int32_t d,
                                     const int32_t* graph_ptr, int32_t capacity_rows, float* dst, int32_t dst_ld, void* stream);
/* back to the padded layout: dst [num_graphs, n_nodes, d]; padded rows <- row fill_row of src (the padding representative)
 * or zeros (fill_row < 0). */
int kgcn_ragged_expand_rows_f32(const float* src, int32_t num_graphs, int32_t n_nodes, int32_t d,
                                const int32_t* graph_ptr, int32_t fill_row, float* dst, void* stream);
/* GraphGather (kgcn/layers.py:163-164, padding rows included: quirk Q4) on the compact layout:
 *   out[b, :] = sum_{r in graph b} x[r, :] + (n_nodes - n_b) * x[pad_row, :]
 * backward: dx[r] = dout[b(r)] on valid rows, dx[pad_row] = sum_b (n_nodes - n_b) dout[b] (deterministic two-stage sum;
 * workspace >= kgcn_ragged_gather_bwd_workspace_bytes(d)), 0 on the other rows >= R. */
int kgcn_ragged_gather_fwd_f32(const float* x, const int32_t* graph_ptr, int64_t batch, int32_t n_nodes, int32_t d,
                               int32_t pad_row, float* out, void* stream);
int64_t kgcn_ragged_gather_bwd_workspace_bytes(int32_t d);
int kgcn_ragged_gather_bwd_f32(const float* dout_grad, const int32_t* graph_ptr, int64_t batch, int32_t n_nodes, int32_t d,
                               int32_t pad_row, int32_t capacity_rows, float* dx, void* workspace, int64_t workspace_bytes,
                               void* stream);

/* -- loss definitions and optimiser update of the model files ---------------------------------------------- */
/* example_model/model_multitask.py:66-79:  cost[b] = mask[b] * sum_t mask_label[b,t] * ce(logits[b,t], labels[b,t]) with
 * tf.nn.sigmoid_cross_entropy_with_logits (weighted == 0) or tf.nn.weighted_cross_entropy_with_logits(pos_weight):
 * pos_weight_per_task [tasks] (device; the reference's info.pos_weight is one weight per label column,
 * kgcn/data_util.py:563-568) or, when it is NULL, the scalar pos_weight for every task;
 * sums[0] = reduce_sum(cost) (cost_sum), sums[1] = reduce_mean(cost) over the PADDED batch (cost_opt, quirk Q5);
 * dlogits [batch, tasks] = d sums[0] / d logits.  mask_label may be NULL (all ones), cost [batch] may be NULL.
 * workspace >= kgcn_loss_workspace_bytes(batch); deterministic (block partials added in a fixed order). */
int64_t kgcn_loss_workspace_bytes(int64_t batch);
int kgcn_masked_sigmoid_ce_f32(const float* logits, const float* labels, const float* mask, const float* mask_label,
                               int64_t batch, int32_t tasks, int32_t weighted, float pos_weight,
                               const float* pos_weight_per_task, float* cost, float* dlogits, float* sums, void* workspace,
                               int64_t workspace_bytes, void* stream);
/* example_model/model.py:56-61:  cost[b] = mask[b] * softmax_cross_entropy(labels[b], logits[b]); same outputs. */
int kgcn_masked_softmax_ce_f32(const float* logits, const float* labels, const float* mask, int64_t batch, int32_t classes,
                               float* cost, float* dlogits, float* sums, void* workspace, int64_t workspace_bytes,
                               void* stream);
/* example_model/sparse.py:112-113: tf.nn.sparse_softmax_cross_entropy_with_logits -- label_idx [batch] int64 class indices
 * instead of dense labels; mask may be NULL (all ones); outputs as above (sums[0] = the reduce_sum the model minimises). */
int kgcn_sparse_softmax_ce_f32(const float* logits, const int64_t* label_idx, const float* mask, int64_t batch,
                               int32_t classes, float* cost, float* dlogits, float* sums, void* workspace,
                               int64_t workspace_bytes, void* stream);
/* Backward of those heads: out = dlogits * (g_sum + g_opt / batch), g_opt / g_sum = the upstream gradients of sums[1] (cost_opt)
 * and sums[0] (cost_sum) as DEVICE scalars, either may be NULL (no gradient through that output). */
int kgcn_loss_grad_f32(const float* dlogits, const float* g_opt, const float* g_sum, int64_t batch, int64_t n, float* out,
                       void* stream);
/* tf.train.AdamOptimizer(lr) (kgcn/core.py:124) over one flat buffer of n floats, with t = *step_counter + 1:
 *   lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t) (fp64);  m = beta1 m + (1 - beta1) g;  v = beta2 v + (1 - beta2) g^2;
 *   params -= lr_t m / (sqrt(v) + eps)
 * then *step_counter += 1 (device int64: a captured hipGraph advances it on every replay). */
int kgcn_adam_tf_f32(float* params, const float* grads, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                     float eps, int64_t* step_counter, void* stream);
/* The same update without a packed gradient buffer: segment q covers `numel` floats at `offset` of params / m / v and reads its
 * gradient from `grad` (the tensor the backward pass produced for that parameter); floats outside the segments are untouched. */
#define KGCN_ADAM_MAX_SEGMENTS 32
typedef struct kgcn_adam_segment {
  const float* grad;
  int64_t offset, numel;
} kgcn_adam_segment;
int kgcn_adam_tf_multi_f32(float* params, float* m, float* v, int64_t n, const kgcn_adam_segment* segments, int32_t num_segments,
                           float lr, float beta1, float beta2, float eps, int64_t* step_counter, void* stream);

/* -- aggregate-FIRST GraphConv: A (X W + 1 b) = (A [X | 1]) [W ; b] --------------------------------------------------- */
/* kgcn/layers.py:112-113 computes fw = X W + b and then A fw: a [rows, dout] aggregation.  When din + 1 < dout the other
 * association is cheaper (an aggregation of din + 1 columns, the same contraction, and in the backward no dout-wide adjoint
 * aggregation at all: dW', db = (A [X | 1])^T d pre-activation).  The bias term rowsum(A) (x) b -- NOT b: rows of A sum to
 * anything, empty rows to 0 -- is carried by a column of ones:
 *   out[r, 0:din] = x[r, 0:din], out[r, din] = 1, out[r, din+1 : out_ld] = 0          (out_ld >= din + 1)
 * backward: dx[r, 0:din] = dout[r, 0:din]. */
int kgcn_augment_ones_f32(const float* x, int64_t m, int32_t din, int64_t x_ld, float* out, int64_t out_ld, void* stream);
int kgcn_augment_ones_bwd_f32(const float* dout_grad, int64_t m, int32_t din, int64_t g_ld, float* dx, int64_t dx_ld,
                              void* stream);

/* -- cross-layer kernels for small graphs: the node-level body of example_model/model.py:42-54 in ONE launch ------------ */
/* For N <= 32 nodes per graph and layer widths <= 64 a graph's activations and all weights fit LDS: one forward and one
 * backward launch run   [GraphConv -> act] x k -> [BatchNormalization (moving statistics) -> act] -> [GraphDense -> act]
 * -> GraphGather   for every graph, instead of one launch and one HBM round trip per layer (at the reference's batch sizes
 * -- 30 graphs in example_config/synth.json -- a step is bound by launch latency).  Two routes with the same results up to
 * the summation order: one graph per workgroup trip on plain fp32 FMAs (a few hundred graphs: the latency of one graph), and
 * 64-row tiles of floor(64 / N) whole graphs on v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate) for thousands.
 *   kind 0  H <- act(A (H W + b))   one adjacency channel (kgcn/layers.py:105-116);  w [din, dout], b [dout] or NULL
 *   kind 1  H <- act(H W + b)       GraphDense (:255-262)
 *   kind 2  H <- act(gamma (H - mean) / sqrt(var + eps) + beta) on rows < enabled[t], act(0) on the others
 *                                   GraphBatchNormalization with its moving statistics (:196-216); w = gamma, b = beta
 * layer_out: HOST array of num_layers device pointers, layer l's output [T, N, dout_l] (written by the forward, read by
 * the backward).  pooled (forward) non-NULL: GraphGather over all N rows (:163-164) -> [T, dout_last].
 * backward: dlast = d pooled [T, dout_last] (gather != 0) or d (last layer output) [T, N, dout_last]; dx [T, N, din_0] or
 * NULL; dparams: flat, per layer dW [din, dout] | db [dout] (kind 2: dgamma [d] | dbeta [d]), kgcn_gcn_stack_param_floats()
 * floats in total; deterministic (one partial per workgroup, fixed-order second stage);
 * workspace >= kgcn_gcn_stack_bwd_workspace_bytes(). */
#define KGCN_STACK_MAX_LAYERS 8
typedef struct kgcn_stack_layer {
  int32_t kind, act, din, dout;
  const float* w;
  const float* b;
  const float* mean; /* kind 2 only */
  const float* var;  /* kind 2 only */
  float eps;         /* kind 2 only */
  int32_t route;     /* read from layers[0] only: 0 automatic, 1 one graph per workgroup trip (plain fp32 FMAs), 2 64-row tiles of
                        whole graphs on the f32 MFMA (<= 4 weight matrices); the others: 0 */
} kgcn_stack_layer;
int kgcn_gcn_stack_supported(int32_t n_nodes, int32_t max_nnz_per_graph, const kgcn_stack_layer* layers, int32_t num_layers);
int64_t kgcn_gcn_stack_param_floats(const kgcn_stack_layer* layers, int32_t num_layers);
int kgcn_gcn_stack_fwd_f32(const kgcn_csr_batch* a, const float* x, const int32_t* enabled, const kgcn_stack_layer* layers,
                           int32_t num_layers, float* const* layer_out, float* pooled, void* stream);
int64_t kgcn_gcn_stack_bwd_workspace_bytes(int32_t num_graphs, const kgcn_stack_layer* layers, int32_t num_layers);
int kgcn_gcn_stack_bwd_f32(const kgcn_csr_batch* at, const float* x, const int32_t* enabled, const kgcn_stack_layer* layers,
                           int32_t num_layers, float* const* layer_out, const float* dlast, int32_t gather, float* dx,
                           float* dparams, void* workspace, int64_t workspace_bytes, void* stream);

/* Deferred second stages of PARAMETER gradients.  Every weight-gradient entry point (kgcn_dense_wgrad*_f32, kgcn_graphconv_bwd_f32)
 * ends by adding its per-workgroup partials in a fixed order -- a 5-7 us launch each, 7-9 per training step of the model files.
 * kgcn_reduce_defer(1) (PROCESS-wide -- frameworks run backward nodes on worker threads --; returns the previous setting) makes those calls QUEUE that second stage instead: dw /
 * dbias are then NOT valid, and the workspaces must stay untouched, until kgcn_reduce_flush(stream) has added all queued
 * partials in ONE launch (same order of additions: bit-identical results).  kgcn_reduce_pending(): queued second stages.
 * Deferring entry points: kgcn_dense_wgrad*_f32, kgcn_dense_bwd*_f32, kgcn_graphconv_bwd_f32 and -- with training == 0 only
 * (learning phase 0: d gamma / d beta are not read inside the call) -- kgcn_graph_bn_bwd_f32 / kgcn_graph_bn_bwd_dact_f32, whose
 * dgamma / dbeta buffers must likewise stay allocated until the flush.
 * The reference has no counterpart (TF sums gradients inside its executor, core.py:124); used by kgcn_amd.train. */
int kgcn_reduce_defer(int32_t on);
int kgcn_reduce_pending(void);
int kgcn_reduce_flush(void* stream);

/* Measurement aid (bench.py / tools/abi_roofline.py price a dense call against the pipe it really runs on): the number of
 * matrix-pipe products per fp32 product of the kernel the library routes this call to, operands assumed 16-byte aligned with
 * the W table present -- 3: f16 two-piece kernels (gemmh.hip, 2.5 PF / 3), 6: bf16 three-piece kernels (gemm3 / gemmn /
 * wgradn / wgradx, 2.5 PF / 6), 1: v_mfma_f32_32x32x2_f32 kernels (157.3 TF), 0: no matrix pipe (read-out layers).
 * kind 0: y = act(x W + b) / dx = dy W^T (din = contraction width), 1: dx with the activation derivative, 2: weight gradient. */
int kgcn_dense_mfma_products(int32_t kind, int64_t m, int32_t din, int32_t dout);

/* Several strided 2-D fp32 copies in ONE launch: dst[r * dst_ld + c] = src[r * src_ld + c] for r < rows, c < cols of every job.
 * Used for the operand of a multi-channel GraphConv's single GEMM, [W_0 | W_1 | ...] and [b_0 | b_1 | ...] (kgcn/layers.py:68-78
 * builds one MatMul per channel; one GEMM over the concatenated kernels produces the same FW[b][ch] blocks side by side), and for
 * the split of that operand's gradient into one contiguous tensor per parameter. */
#define KGCN_COPY2D_MAX_JOBS 16
typedef struct kgcn_copy2d_job {
  const float* src;
  float* dst;
  int64_t rows, cols, src_ld, dst_ld;
} kgcn_copy2d_job;
int kgcn_copy2d_multi_f32(const kgcn_copy2d_job* jobs, int32_t num_jobs, void* stream);

/* Measurement aid (bench.py: `roofline.hbm_probe`, SURVEY 8(d) "report both nominal and achievable"): one grid-stride float4
 * stream over `bytes` bytes per operand in the read : write mix of the kernel being priced, so that a bench line carries what
 * THIS box's HBM delivers right after the timed region.  mix 0: b = a (1 : 1, the batched SpMM's mix); 1: b = a + a2 (2 : 1, the
 * fused backward's mix); 2: read a only (b receives at most 16 bytes); 3: write b only.  No reference counterpart. */
int kgcn_hbm_probe(int32_t mix, const void* a, const void* a2, void* b, int64_t bytes, void* stream);

/* dW, dbias of act(x W + b) when the layer INPUT needs no gradient (first layer of a model): d pre-activation = dy * act'(act_out)
 * is formed while the weight-gradient GEMM stages the gradient rows, so it never exists in HBM.  Wide layers only
 * (kgcn_dense_wgrad_dact_supported(din, dout) != 0); otherwise run kgcn_act_bwd_f32 + kgcn_dense_wgrad_f32.  dy and act_out
 * share the leading dimension dy_ld; workspace as for kgcn_dense_wgrad_f32. */
int kgcn_dense_wgrad_dact_supported(int32_t din, int32_t dout);
int kgcn_dense_wgrad_dact_f32(const float* x, int64_t x_ld, const float* dy, const float* act_out, int64_t dy_ld, int32_t act,
                              int64_t m, int32_t din, int32_t dout, float* dw, float* dbias, void* workspace,
                              int64_t workspace_bytes, void* stream);

/* Backward of GINAggregate (kgcn/layers.py:461-472) in one call: dx = sum_c (eps_c g + A_c^T g) -- at_ch = the TRANSPOSED
 * channel containers -- and d eps = <g, x> (one scalar: the same for every channel, :469), accumulated while the gradient
 * tiles are staged for the aggregation (no pass of its own over g and x); deterministic (one partial per graph, fixed-order
 * sum).  dx or deps may be NULL.  workspace >= kgcn_gin_aggregate_bwd_workspace_bytes(T, N, d). */
int64_t kgcn_gin_aggregate_bwd_workspace_bytes(int32_t num_graphs, int32_t n_nodes, int32_t d);
int kgcn_gin_aggregate_bwd_f32(const kgcn_csr_batch* at_ch, int32_t num_channels, const float* grad, int32_t d,
                               const float* eps, const float* x, float* dx, float* deps, void* workspace,
                               int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KGCN_HIP_H_ */
