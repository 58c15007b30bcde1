/*
 * mclstexp_hip.h -- C ABI of libmclstexp_hip.so: hand-written gfx950 (MI355X) kernels for the
 * contrastive training hot path of mclSTExp.
 *
 * The reference (ZhicengShi/mclSTExp) has no native code and no FFI: every entry point below
 * replaces the ATen dispatch behind one call site of /root/reference/model.py or train.py, cited
 * per function.  Conventions:
 *   - plain pointers + sizes only; every buffer is caller-owned DEVICE memory (the Python host
 *     passes torch tensors' data_ptr()); the library never allocates, frees or retains pointers;
 *   - every call only ENQUEUES work on the caller's stream (hipStream_t passed as void*); no host
 *     synchronisation, no global mutable state; safe from one host thread per device;
 *   - return 0 on success, a positive hipError_t if a launch failed, a negative MCL_E* code for a
 *     rejected argument.  No exceptions cross the ABI.
 *   - all matrices are fp32 row-major with explicit leading dimensions (elements, not bytes);
 *     G (genes) need not be a multiple of anything: tails are masked in-kernel.
 */
#ifndef MCLSTEXP_HIP_H
#define MCLSTEXP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCL_ABI_VERSION 9

#define MCL_OK 0
#define MCL_EINVAL (-1)       /* null pointer / non-positive size / inconsistent arguments */
#define MCL_EUNSUPPORTED (-2) /* valid but not implemented (e.g. head_dim != 64) */
#define MCL_EWORKSPACE (-3)   /* workspace too small (see the *_workspace_bytes query) */

typedef void* mcl_stream_t; /* hipStream_t */

int mcl_abi_version(void);
const char* mcl_error_string(int code);

/* ---------------------------------------------------------------- GEMM (K3, K5, K6, K7, K8 contractions)
 * C[b] = epilogue( alpha * A[b] (M x K) * B[b] (K x N) ), fp32 in HBM.
 * Element (m,k) of A is A[b*sAb + m*sAm + k*sAk]; element (k,n) of B is B[b*sBb + k*sBk + n*sBn];
 * exactly one of (sAm,sAk) and one of (sBk,sBn) must be 1.  This single strided form covers
 *   y = x W^T      nn.Linear forward      model.py:23,27,43,45,155,157   (A k-contig, B k-contig)
 *   dx = dy W      nn.Linear backward-data                              (A k-contig, B n-contig)
 *   dW = dy^T x    nn.Linear backward-weight                            (A m-contig, B n-contig)
 *   q k^T, attn v  einsum model.py:53,55 batched over heads (b = head, strided views of qkv)
 *   S = E_spot E_img^T / T                model.py:242
 * Epilogue, in order:  v = alpha*acc ; v += bias[n] ; if pre_out: pre_out[m,n] = v ;
 *   GELU: v = gelu_erf(v) (nn.GELU, model.py:25,156) ; GELU_BWD: v *= gelu_erf'(aux[m,n]) ;
 *   v += resid[m,n] (residual, model.py:67,68,165) ; ACCUM: v += C[m,n] ; C[m,n] = v.
 * compute: MCL_COMPUTE_F32 = v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate);
 *          MCL_COMPUTE_BF16 = operands rounded to bf16 at LDS staging, v_mfma_f32_32x32x16_bf16.
 */
#define MCL_EPI_GELU 1
#define MCL_EPI_GELU_BWD 2
#define MCL_EPI_ACCUM 4
#define MCL_COMPUTE_F32 0
#define MCL_COMPUTE_BF16 1

typedef struct mcl_gemm_args {
  uint32_t struct_size;  /* ABI 8: sizeof(mcl_gemm_args) AS THE CALLER COMPILED / DECLARED IT.  The struct travels by pointer and
                            has grown (ABI 7 appended the flt_* block): mcl_gemm copies min(struct_size, its own sizeof) bytes and
                            reads every field beyond the caller's size as zero / NULL, so a binding written against a shorter
                            layout stays valid; a size below the end of `workspace` (the ABI-6 fields) is MCL_EINVAL.  */
  int32_t M, N, K, batch;
  const float* A; int64_t sAm, sAk, sAb;
  const float* B; int64_t sBk, sBn, sBb;
  float* C; int64_t ldc, sCb;
  float alpha;
  int32_t flags;         /* MCL_EPI_* */
  const float* bias;     /* [N] or NULL */
  const float* resid; int64_t ldr, sRb;  /* NULL or (M,N) per batch */
  float* pre_out; int64_t ldp;           /* NULL or (M,N): pre-activation store (batch must be 1) */
  const float* aux; int64_t ldaux;       /* (M,N) for MCL_EPI_GELU_BWD (batch must be 1) */
  int32_t compute;       /* MCL_COMPUTE_* */
  int32_t ksplit;        /* 0 / 1: one pass.  > 1: K is cut into that many slices whose fp32 partials go to
                            `workspace` and a second launch adds them in fixed order and applies the epilogue
                            (deterministic; for skinny problems such as the spot path's M = batch of 128 spots) */
  float* workspace;      /* >= mcl_gemm_workspace_floats(M, N, batch, ksplit) floats when ksplit > 1, else unused */
  /* ABI 7: threshold FILTER instead of the C store (fused similarity + top-k, /root/reference/evel_her2st.py:74-84).
   * flt_thr != NULL: nothing is written to C (may be NULL); every product alpha * (A B)[i][j] >= flt_thr[i] is appended to row i's
   * candidate list: pos = atomic flt_cnt[i]++ (zero on entry), flt_val[i * flt_cap + pos] = value, flt_idx[...] = j when
   * pos < flt_cap (flt_cnt keeps counting past the capacity: the caller sees the overflow).  batch 1, no split-K, no epilogue. */
  const float* flt_thr; int32_t* flt_cnt; float* flt_val; int32_t* flt_idx; int32_t flt_cap;
  /* ABI 9: split-K in ONE launch.  counters != NULL (and ksplit > 1, 64x64 tiles): ceil(M/64) * ceil(N/64) * batch uint32, ZERO on
   * entry and left zero -- the last K slice to arrive at a tile adds the slices in slice order and applies the epilogue itself
   * (bit-identical to the two-launch form); one counter array per concurrently running call (e.g. per stream).  NULL: two launches. */
  uint32_t* counters;
} mcl_gemm_args;

int mcl_gemm(const mcl_gemm_args* args, mcl_stream_t stream);
/* sizeof(mcl_gemm_args) of THIS library build and the smallest struct_size it accepts (the fields up to `workspace`): a binding
 * in a language without the header (ctypes, cgo, JNI) checks its own struct declaration against these at load time. */
uint32_t mcl_gemm_args_size(void);
uint32_t mcl_gemm_args_min_size(void);
/* Slices that give a problem with few 64x64 output tiles ~256 workgroups (1 = do not split). */
int32_t mcl_gemm_auto_ksplit(int32_t M, int32_t N, int32_t K, int32_t batch);
int64_t mcl_gemm_workspace_floats(int32_t M, int32_t N, int32_t batch, int32_t ksplit);
/* n <= 4 independent problems as ONE launch (a layer's backward: its weight gradients and its data gradient are a few dozen
 * 64x64 tiles each).  `args`: an array of mcl_gemm_args, args[0].struct_size bytes apart, every element with the same
 * struct_size.  Each problem keeps its own layouts and epilogue; results are bit-identical to n separate mcl_gemm calls.
 * Restrictions: compute MCL_COMPUTE_F32, batch 1, ksplit <= 1, no filter epilogue -> MCL_EUNSUPPORTED otherwise.           */
int mcl_gemm_group(const mcl_gemm_args* args, int32_t n, mcl_stream_t stream);

/* ---------------------------------------------------------------- K7 ProjectionHead (model.py:151-168), projection_dim = 256
 * Forward as ONE launch, exact fp32 (csrc/proj_head.hip):
 *     p = x Wp^T + bp ;  a = gelu(p) ;  z = a Wf^T + bf + p ;  e = LayerNorm(z; gamma, beta, eps)      (dropout p = 0)
 * x (M, D) row stride ldx; Wp (256, D) row stride ldwp; Wf (256, 256) row stride ldwf; bp, bf, gamma, beta (256).
 * Outputs, all dense (M, 256) / (M): e, and for the backward p, a, z, mean, rstd.  The first product is cut into `ksplit`
 * K slices over the grid (mcl_proj_head_ksplit(M, D) recommends; 1..16); the last workgroup to arrive at a block of 16 rows adds
 * the slices in slice order (bit-reproducible) and finishes the rows.  ws: >= mcl_proj_head_ws_floats(M, ksplit) floats;
 * counters: ceil(M / 16) uint32, ZERO before the first call and left zero by every call -- one counter array per concurrently
 * running call.  16-byte aligned: e, p, a, z, ws, bp, gamma, beta.
 * Backward, row-local part as ONE launch:
 *     dz = LayerNorm'(de) ;  dp = (dz Wf) * gelu'(p) + dz ;  dgamma = sum_rows de * xhat, dbeta = sum_rows de,
 *     dbf = sum_rows dz, dbp = sum_rows dp     (bit i of accumulate_mask: output i of (dgamma, dbeta, dbf, dbp) is added to --
 *     the parameter's .grad -- instead of overwritten; a NULL output is skipped; sums over row blocks in block order).
 * dz, dp dense (M, 256); ws >= ceil(M / 16) * 1024 floats; counter: ONE zero uint32, as above.  The remaining three products
 * (dWf = dz^T a, dWp = dp^T x, dx = dp Wp) are plain problems for mcl_gemm_group.                                              */
int32_t mcl_proj_head_ksplit(int32_t M, int32_t D);
int64_t mcl_proj_head_ws_floats(int32_t M, int32_t ksplit);
int mcl_proj_head_fwd(const float* x, int64_t ldx, int32_t M, int32_t D, const float* wp, int64_t ldwp, const float* bp,
                      const float* wf, int64_t ldwf, const float* bf, const float* gamma, const float* beta, float eps,
                      float* e, float* p, float* a, float* z, float* mean, float* rstd, float* ws, uint32_t* counters,
                      int32_t ksplit, mcl_stream_t stream);
int mcl_proj_head_bwd_rows(const float* de, int64_t ldde, int32_t M, const float* z, const float* mean, const float* rstd,
                           const float* gamma, const float* p, const float* wf, int64_t ldwf, float* dz, float* dp,
                           float* dgamma, float* dbeta, float* dbf, float* dbp, int32_t accumulate_mask, float* ws,
                           uint32_t* counter, mcl_stream_t stream);

/* ---------------------------------------------------------------- K1 position-embedding add
 * model.py:230-235:  out[b,:] = expr[b,:] + X[(long)pos[b,0],:] + Y[(long)pos[b,1],:]
 * (tables model.py:204-205, 65536 rows).  (long) truncates toward zero.  ix/iy (int32, B) receive
 * the indices for the backward.  Indices outside [0,n_rows) are clamped and *err_flag (device int,
 * may be NULL) is set to 1 -- nn.Embedding would raise.                                        */
int mcl_pos_embed_add_fwd(const float* expr, int64_t ld_expr, const float* pos /*(B,2)*/,
                          const float* x_table, const float* y_table, int64_t ld_table, int32_t n_rows,
                          float* out, int64_t ld_out, int32_t* ix, int32_t* iy, int32_t* err_flag,
                          int32_t B, int32_t G, mcl_stream_t stream);

/* Backward of the gather (autograd of nn.Embedding, dense in the reference: two (65536,G) grads).
 * Row-sparse, deterministic: slot b is "owner" iff no b' < b has idx[b'] == idx[b]; then
 * row_grad[b,:] = sum_{b'>=b, idx[b']==idx[b]} d_out[b',:] (ascending b') and owner_idx[b] = idx[b];
 * otherwise owner_idx[b] = -1 and row_grad[b,:] is untouched.                                   */
int mcl_embed_rowgrad(const float* d_out, int64_t ld_dout, const int32_t* idx, int32_t* owner_idx,
                      float* row_grad, int64_t ld_rg, int32_t B, int32_t G, mcl_stream_t stream);

/* Dense scatter of owner rows: table_grad[owner_idx[b],:] (+)= row_grad[b,:] for owners
 * (accumulate != 0 adds, else overwrites).  For drop-in use with a stock dense optimizer.      */
int mcl_embed_scatter_rows(const int32_t* owner_idx, const float* row_grad, int64_t ld_rg, float* table_grad,
                           int64_t ld_table, int32_t B, int32_t G, int32_t accumulate, mcl_stream_t stream);

/* ---------------------------------------------------------------- K2 LayerNorm (model.py:13,17,158,166)
 * y = (x-mean)*rstd*gamma + beta over the last dim (biased variance, eps inside the sqrt).
 * mean/rstd (rows) are saved for the backward.                                                */
int mcl_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y, int64_t ldy,
                      float* mean, float* rstd, int32_t rows, int32_t cols, float eps, mcl_stream_t stream);
/* dx = rstd*(g - mean(g) - xhat*mean(g*xhat)) (+ dx_add if non-NULL: the residual branch's gradient,
 * may alias dx), g = dy*gamma.  dgamma/dbeta (cols) are OVERWRITTEN with the column sums over rows (accumulate_params != 0:
 * added to).  ABI 9: dgamma == dbeta == NULL computes dx alone (the parameter gradients then belong to a grouped
 * mcl_colred_group launch).                                                                              */
int mcl_layernorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                      const float* mean, const float* rstd, const float* dx_add, int64_t ldadd, float* dx,
                      int64_t lddx, float* dgamma, float* dbeta, int32_t accumulate_params, int32_t rows, int32_t cols,
                      mcl_stream_t stream);

/* Fused attention core of the spot Transformer (/root/reference/model.py:49-57), head dimension 64, fp32 on the matrix cores
 * (v_mfma_f32_32x32x2_f32), no (heads, B, B) tensor in HBM.  qkv: (B, 3*heads*64) fp32, row stride ld, columns q | k | v,
 * head-major inside each; out: (B, heads*64), row stride ldo; lse: (heads, B) row log-sum-exp of the scaled scores (all the
 * backward needs beside qkv and out).  Backward: dqkv (B, 3*heads*64, row stride ldq) from dout (row stride ldo, as out);
 * dvec: (heads, B) fp32 scratch.  One launch forward, two backward (the unfused sequence: 3 + 5).  16-byte aligned bases,
 * strides multiples of 4 floats; dim_head != 64 -> MCL_EUNSUPPORTED (callers keep the GEMM + softmax path for it).      */
int mcl_attention_fwd(const float* qkv, int64_t ld, int32_t B, int32_t heads, int32_t dim_head, float scale, float* out,
                      int64_t ldo, float* lse, mcl_stream_t stream);
int mcl_attention_bwd(const float* qkv, int64_t ld, int32_t B, int32_t heads, int32_t dim_head, float scale, const float* out,
                      const float* dout, int64_t ldo, const float* lse, float* dvec, float* dqkv, int64_t ldq,
                      mcl_stream_t stream);
/* The same core over nseq independent sequences of B tokens each (ABI 7): rows sequence-major, lse / dvec (nseq, heads, B).
 * The fp32 ("reference numerics") ViT image encoder runs it with one sequence per image (/root/reference/model.py:104-116:
 * timm's Attention inside vit_base_patch{16,32}_224, T = 197 / 50 tokens). */
int mcl_attention_batched_fwd(const float* qkv, int64_t ld, int32_t B, int32_t nseq, int32_t heads, int32_t dim_head, float scale,
                              float* out, int64_t ldo, float* lse, mcl_stream_t stream);
int mcl_attention_batched_bwd(const float* qkv, int64_t ld, int32_t B, int32_t nseq, int32_t heads, int32_t dim_head, float scale,
                              const float* out, const float* dout, int64_t ldo, const float* lse, float* dvec, float* dqkv,
                              int64_t ldq, mcl_stream_t stream);
/* ---------------------------------------------------------------- K4 attention softmax (model.py:53-54)
 * In place over n_rows rows of length cols (row stride ld):  p = softmax(scale * s).           */
int mcl_softmax_rows_fwd(float* s, int64_t ld, int32_t n_rows, int32_t cols, float scale, mcl_stream_t stream);
/* In place on dp:  ds = scale * p * (dp - sum_j dp_j p_j).                                     */
int mcl_softmax_rows_bwd(const float* p, float* dp, int64_t ld, int32_t n_rows, int32_t cols, float scale,
                         mcl_stream_t stream);

/* ---------------------------------------------------------------- bias gradient
 * out[n] = sum_m x[m,n]  (nn.Linear bias backward).                                           */
int mcl_colsum(const float* x, int64_t ldx, float* out, int32_t rows, int32_t cols, int32_t accumulate, mcl_stream_t stream);
/* n <= 6 column reductions over the same number of rows (<= 1024) as ONE launch -- a Transformer layer's bias gradients and
 * LayerNorm parameter gradients (model.py:13,17,23,27,45).  Host arrays of n entries.  x[i] == NULL: out0[i][c] (+)= sum_r a[i][r, c]
 * (mcl_colsum).  x[i] != NULL: out0[i][c] (+)= sum_r a[i][r, c] * (x[i][r, c] - mean[i][r]) * rstd[i][r] and out1[i][c] (+)=
 * sum_r a[i][r, c] (the dgamma / dbeta of mcl_layernorm_bwd, which computes dx alone when its dgamma and dbeta are both NULL).
 * accumulate[i] != 0: +=.  Per problem bit-identical to the separate launches.                                              */
int mcl_colred_group(int32_t n, const float* const* a, const int64_t* lda, const float* const* x, const int64_t* ldx,
                     const float* const* mean, const float* const* rstd, float* const* out0, float* const* out1,
                     const int32_t* cols, const int32_t* accumulate, int32_t rows, mcl_stream_t stream);
/* ABI 7: the two column reductions above for MANY rows (the fp32 ViT's 6 400 - 25 216 token rows).  With a workspace of
 * mcl_rowred_workspace_floats(rows, cols) floats and rows > 1024 the rows are reduced in 128-row chunks on a 2-D grid and the
 * chunk partials added in chunk order by a second launch (deterministic); otherwise exactly the calls above. */
int64_t mcl_rowred_workspace_floats(int32_t rows, int32_t cols);
int mcl_colsum_ws(const float* x, int64_t ldx, float* out, int32_t rows, int32_t cols, int32_t accumulate, float* workspace,
                  mcl_stream_t stream);
int mcl_layernorm_bwd_ws(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, const float* mean,
                         const float* rstd, const float* dx_add, int64_t ldadd, float* dx, int64_t lddx, float* dgamma,
                         float* dbeta, int32_t accumulate_params, int32_t rows, int32_t cols, float* workspace, mcl_stream_t stream);

/* ---------------------------------------------------------------- f4: soft-target contrastive loss, elementwise middle (ABI 7)
 * /root/reference/baselines/Bleep/models.py:34-43,66-76,228-234.  Given S = E_s E_i^T / T, the soft targets Tg (row softmax),
 * the row / column log-sum-exp of S (mcl_infonce_lse) and the column sums of Tg (mcl_colsum), ONE pass writes
 * dS = c (softmax_row(S) + softmax_col(S) tcol - 2 Tg), dA = -c (logsoftmax_row(S) + logsoftmax_col(S)) and, per row, the loss
 * partial loss_rows[i] = -c sum_j Tg_ij (logsoftmax_row + logsoftmax_col)_ij (their sum is the loss; c = 1 / (2B)).  All (B, B)
 * row-major fp32.  mcl_symmetrize: out = (a + a^T) / 2 (out != a). */
int mcl_soft_clip_mid(const float* S, const float* Tg, const float* lse_row, const float* lse_col, const float* tcol, int32_t B,
                      float c, float* dS, float* dA, float* loss_rows, mcl_stream_t stream);
int mcl_symmetrize(const float* a, int32_t B, float* out, mcl_stream_t stream);

/* ---------------------------------------------------------------- K8 symmetric InfoNCE (model.py:242-247)
 * Works on a logits strip S (R x C, already divided by T): this rank's rows are global rows
 * [row0,row0+R) and its columns are global columns [col0,col0+C); the target (identity) entry of
 * local (i,j) is at row0+i == col0+j.
 *
 * mcl_infonce_lse: row_lse[i] = LSE_j S_ij (if row_lse != NULL) and col_lse[j] = LSE_i S_ij (if
 * col_lse != NULL), max-subtracted.                                                            */
int mcl_infonce_lse(const float* S, int64_t ldS, int32_t R, int32_t C, float* row_lse, float* col_lse,
                    mcl_stream_t stream);
/* loss_sum[0] (+)= sum_i (row_lse[i] - S_ii)  and  loss_sum[1] (+)= sum_j (col_lse[j] - S_jj) over
 * the target entries present in this strip; n_diag = how many to visit starting at local
 * (di0, dj0).  The reference's loss is (loss_sum[0]+loss_sum[1]) / (2*B_glob).                 */
int mcl_infonce_loss(const float* S, int64_t ldS, const float* row_lse, const float* col_lse,
                     int32_t di0, int32_t dj0, int32_t n_diag, int32_t use_rows, int32_t use_cols,
                     float* loss_sum, mcl_stream_t stream);
/* dS_ij = coef * (exp(S_ij-row_lse[i]) + exp(S_ij-col_lse[j]) - 2*[row0+i == col0+j]), written to
 * dS (may alias S).  coef = 1/(2*B_glob*T) folds the temperature of model.py:242.              */
/* loss_out[0] = (sum_t (row_lse[t] - S[t][t]) + sum_t (col_lse[t] - S[t][t])) / denom over the n diagonal entries: the
 * scalar of /root/reference/model.py:244-247 in ONE launch (mcl_infonce_loss accumulates the two sums for the strip form
 * and leaves the final add / divide to the caller).                                                                    */
int mcl_infonce_loss_mean(const float* S, int64_t ldS, const float* row_lse, const float* col_lse, int32_t n, float denom,
                          float* loss_out, mcl_stream_t stream);
int mcl_infonce_dlogits(const float* S, int64_t ldS, const float* row_lse, const float* col_lse, int32_t R,
                        int32_t C, int32_t row0, int32_t col0, float coef, float* dS, int64_t lddS,
                        mcl_stream_t stream);

/* ---------------------------------------------------------------- K8 fused InfoNCE (model.py:242-247), bf16 MFMA
 * The same loss without ever writing the logits to HBM (flash-attention-shaped; csrc/infonce_fused.hip).
 * a: (R, dim) and b: (C, dim) bf16 with row strides lda / ldb (elements, multiples of 8; bases 16-byte
 * aligned), dim == 256 (projection_dim); S = a b^T * inv_temp.
 * The positive pair of local row r is column r + diag_off (data parallel: rank * B_loc).  Calling with (a, b)
 * = (E_spot, E_img) gives the row direction of the symmetric loss, with (E_img, E_spot) the column direction.
 *
 * mcl_infonce_fused_lse:  lse[r] = log sum_c exp(S[r,c]);  diag[r] = S[r, r+diag_off] (written where that
 *   column exists; diag may be NULL).
 * mcl_infonce_fused_grad: dA[r,:] = coef * sum_c (exp(S[r,c]-lse_a[r]) + exp(S[r,c]-lse_b[c])
 *                                                - 2*[c == r+diag_off]) * b[c,:]      (fp32, (R, dim) contiguous)
 *   i.e. d loss / d a with coef = inv_temp / (2*B_glob): the closed-form backward of model.py:244-247.
 * workspace: mcl_infonce_fused_workspace_bytes(R, C, dim) bytes of device memory, 16-byte aligned.       */
int64_t mcl_infonce_fused_workspace_bytes(int32_t R, int32_t C, int32_t dim);
int mcl_infonce_fused_lse(const void* a, int64_t lda, const void* b, int64_t ldb, int32_t R, int32_t C, int32_t dim,
                          int32_t diag_off, float inv_temp, float* lse, float* diag, void* workspace, int64_t ws_bytes,
                          mcl_stream_t stream);
int mcl_infonce_fused_grad(const void* a, int64_t lda, const void* b, int64_t ldb, int32_t R, int32_t C, int32_t dim,
                           int32_t diag_off, float inv_temp, const float* lse_a, const float* lse_b, float coef, float* dA,
                           void* workspace, int64_t ws_bytes, mcl_stream_t stream);
/* y (bf16, row stride ldy) = round-to-nearest-even(x (fp32, row stride ldx)); cols % 8 == 0.             */
int mcl_cast_f32_to_bf16(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int32_t cols,
                         mcl_stream_t stream);

/* ---- ViT image encoder on bf16 (csrc/gemm_bf16.hip, csrc/vit_ops.hip; /root/reference/model.py:104-116).
 * mcl_gemm_bf16: C[b] = epilogue(alpha * A[b] B[b]) with fp32 accumulation on the bf16 MFMA; per-operand storage flags
 * (bit 0: A stored reduction-major [K][M]; bit 1: B stored reduction-major [K][N]; default: A [M][K], B [N][K]),
 * bit 2: exact-erf GELU (pre_out != NULL also stores the pre-activation; with bit 5: stores gelu'(pre-activation) instead),
 * bit 3: multiply by gelu'(aux), bit 6: multiply by aux itself (aux = the derivative stored through bit 5), bit 4: fp32
 * output.  Two-level batch: problem bi = (bi / batch2, bi % batch2) with strides (s?b, s?b2) -- (image, head) for the
 * attention products.  bias [N] fp32, resid [M][N] bf16 (outer-batch stride sRb, 0 = broadcast).  Leading dimensions multiples of 8 (bf16) /
 * 4 (fp32 C); ragged M, N, K allowed; bf16 C needs ldc >= round_up(N, 8).  ksplit > 1 (fp32 output, batch 1, no
 * epilogue): the K range is split over workgroups, fp32 slabs in `workspace` (mcl_gemm_bf16_workspace_floats) are
 * merged in fixed order into C (accumulate != 0: +=) -- the weight-gradient form, deterministic.                      */
int64_t mcl_gemm_bf16_workspace_floats(int32_t M, int64_t ldc, int32_t ksplit);
int mcl_gemm_bf16(const void* A, int64_t lda, int64_t sAb, const void* B, int64_t ldb, int64_t sBb, void* C, int64_t ldc,
                  int64_t sCb, int32_t M, int32_t N, int32_t K, int32_t batch, int32_t batch2, int64_t sAb2, int64_t sBb2,
                  int64_t sCb2, float alpha, int32_t flags,
                  const float* bias, const void* resid, int64_t ldr, int64_t sRb, const void* aux, int64_t ldaux,
                  void* pre_out, int64_t ldp, int32_t ksplit, float* workspace, int32_t accumulate, mcl_stream_t stream);
/* LayerNorm over the last dimension D (multiple of 8, <= 1024) on bf16 rows, fp32 affine parameters and statistics. */
int mcl_ln_bf16_fwd(const void* x, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy, float* mean,
                    float* rstd, int64_t rows, int32_t D, float eps, mcl_stream_t stream);
/* dx = LayerNorm backward (+ dx_add); dgamma / dbeta (accumulate != 0: +=) merged deterministically.               */
int64_t mcl_colred_workspace_floats(int64_t rows, int32_t D);
int mcl_ln_bf16_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma, const float* mean,
                    const float* rstd, const void* dx_add, int64_t ldadd, void* dx, int64_t lddx, float* workspace,
                    float* dgamma, float* dbeta, int32_t accumulate, int64_t rows, int32_t D, mcl_stream_t stream);
/* out[c] (+)= sum over rows of a bf16 (rows, D) matrix (bias gradients), deterministic; D % 4 == 0, D <= 3072.      */
int mcl_colsum_bf16(const void* x, int64_t ldx, int64_t rows, int32_t D, float* workspace, float* out, int32_t accumulate,
                    mcl_stream_t stream);
/* Row softmax in place on bf16 scores (row length n <= 256, row stride ld; padding columns are zeroed) and its
 * backward in place on dP: dS = P * (dP - sum P dP) * scale.                                                        */
int mcl_softmax_bf16_fwd(void* s, int64_t ld, int64_t rows, int32_t n, mcl_stream_t stream);
int mcl_softmax_bf16_bwd(const void* P, void* dP, int64_t ld, int64_t rows, int32_t n, float scale, mcl_stream_t stream);
/* Fused attention core of the ViT encoder (csrc/vit_attention.hip; replaces the reference's timm Attention.forward core,
 * /root/reference/model.py:104-116 -> timm vision_transformer.Attention: softmax(q k^T scale) v per image and head), bf16,
 * head dimension 64, T <= 224 tokens, no (B heads, T, T) tensor in HBM.  qkv (B, T, 3 heads 64) bf16 as the qkv linear
 * writes it; o / dout (B, T, heads 64) bf16; lse, dsum (B heads, T) fp32 (dsum is scratch written by the backward);
 * dqkv (B, T, 3 heads 64) bf16, every element written.  T > 224 -> MCL_EUNSUPPORTED (callers keep the GEMM + softmax path). */
int mcl_vit_attn_fwd(const void* qkv, void* o, float* lse, int32_t B, int32_t T, int32_t heads, float scale, mcl_stream_t stream);
int mcl_vit_attn_bwd(const void* qkv, const void* o, const void* dout, const float* lse, float* dsum, void* dqkv, int32_t B,
                     int32_t T, int32_t heads, float scale, mcl_stream_t stream);
/* ---- the ViT's token assembly and token mean on own kernels (csrc/glue.hip; timm VisionTransformer.forward_features behind
 * /root/reference/model.py:104-116).  x / dx: (B, T, D) with T = patches + 1, row 0 of an image = the class token; dtype 0 = fp32,
 * 1 = bf16; cls (D), pos (T*D) fp32 parameters.
 *   mcl_vit_patchify_tokens  mcl_vit_patchify with (lead_zero_row) one zero row in front of every image's patches -- the (B, T, K)
 *                            token matrix whose row 0 carries no patch -- and (out_f32) an fp32 result;
 *   mcl_vit_cls_row          x[b][0] = cls + pos[0]                        (the patch rows come from the patch GEMM's epilogue)
 *   mcl_vit_assemble_f32     x[b][0] = cls + pos[0] ; x[b][t] = tok[b*(T-1) + t-1] + pos[t]          (fp32 path)
 *   mcl_vit_tokens_extract   dtok[b*(T-1) + t-1] = dx[b][t], t >= 1
 *   mcl_vit_token_mean_fwd   feat[b] = mean over t >= 1 of x[b][t]  (fp32, token order)              (global_pool = 'avg')
 *   mcl_vit_token_mean_bwd   dx[b][0] = 0 ; dx[b][t >= 1] = dfeat[b] / (T - 1)
 *   mcl_vit_pos_grad         dpos[t] (+)= sum_b dx[b][t] (batch order) ; dcls (+)= its t = 0 row (NULL: skipped);
 *                            accumulate: bit 0 adds into dpos, bit 1 adds into dcls
 *   mcl_vit_zero_cls_rows    dx[b][0] = 0                                   (before the patch-embedding bias gradient)        */
int mcl_vit_patchify_tokens(const float* img, int64_t sb, int64_t sc, int64_t sy, int64_t sx, int32_t B, int32_t H, int32_t W,
                            int32_t p, void* out, int32_t lead_zero_row, int32_t out_f32, mcl_stream_t stream);
int mcl_vit_cls_row(const float* cls, const float* pos, void* x, int32_t B, int32_t T, int32_t D, int32_t dtype,
                    mcl_stream_t stream);
int mcl_vit_assemble_f32(const float* tok, const float* cls, const float* pos, float* x, int32_t B, int32_t T, int32_t D,
                         mcl_stream_t stream);
int mcl_vit_tokens_extract(const void* dx, void* dtok, int32_t B, int32_t T, int32_t D, int32_t dtype, mcl_stream_t stream);
int mcl_vit_token_mean_fwd(const void* x, float* feat, int32_t B, int32_t T, int32_t D, int32_t dtype, mcl_stream_t stream);
int mcl_vit_token_mean_bwd(const float* dfeat, void* dx, int32_t B, int32_t T, int32_t D, int32_t dtype, mcl_stream_t stream);
int mcl_vit_pos_grad(const void* dx, float* dpos, float* dcls, int32_t B, int32_t T, int32_t D, int32_t dtype,
                     int32_t accumulate, mcl_stream_t stream);
int mcl_vit_zero_cls_rows(void* dx, int32_t B, int32_t T, int32_t D, int32_t dtype, mcl_stream_t stream);
/* 4-D strided copy of an fp32 source, element strides on both sides: dst[i . d] (+)= src[i . a], i over (n0, n1, n2, n3); dst fp32
 * (dst_dtype 0) or bf16 (1).  Packs a Conv2d weight of any memory format into the (c, iy, ix) column order of the patch GEMM
 * (with the cast), and adds a contiguous weight gradient into a .grad that has the parameter's own strides.               */
int mcl_strided4_f32(const float* src, int32_t n0, int32_t n1, int32_t n2, int32_t n3, int64_t a0, int64_t a1, int64_t a2,
                     int64_t a3, void* dst, int64_t d0, int64_t d1, int64_t d2, int64_t d3, int32_t dst_dtype, int32_t accumulate,
                     mcl_stream_t stream);
/* 2-D copy with row strides in BYTES (a channel slice of a channels-last buffer -> dense rows, or back); 2-byte granularity. */
int mcl_copy_rows(const void* src, int64_t ld_src_bytes, void* dst, int64_t ld_dst_bytes, int64_t rows, int64_t row_bytes,
                  mcl_stream_t stream);
/* wf[ci][ky][kx][co] = w[co][k-1-ky][k-1-kx][ci]: the weight of the "same" convolution that IS the backward-data pass of a
 * stride-1 convolution (w: (Co, k, k, Ci) storage = a channels-last Conv2d weight); dtype 0 fp32 / 1 bf16.                  */
int mcl_weight_rot180(const void* w, void* wf, int32_t Co, int32_t k, int32_t Ci, int32_t dtype, mcl_stream_t stream);
/* Patch extraction for the patch-embedding GEMM: out[(b, py, px)][c*p*p + iy*p + ix] (bf16) from an fp32 image
 * addressed by element strides (sb, sc, sy, sx): NCHW or channels-last.                                            */
int mcl_vit_patchify(const float* img, int64_t sb, int64_t sc, int64_t sy, int64_t sx, int32_t B, int32_t H, int32_t W,
                     int32_t p, void* out_bf16, mcl_stream_t stream);

/* ---- fp8 (OCP e4m3) similarity contraction for the InfoNCE (csrc/infonce_fp8.hip; BASELINE configs[4]).
 * mcl_quant_e4m3_rows: per row one power-of-two scale 2^e (smallest with max|x| <= 448 * 2^e); q = e4m3_rne(x 2^-e),
 * scale byte = e + 127 (E8M0, what v_mfma_scale_f32_32x32x64_f8f6f4 consumes); optionally the dequantised copy in
 * bf16 (exactly representable).  x (rows, 256) fp32 row stride ldx; q (rows, 256) bytes row stride ldq; scale bytes
 * with byte stride ld_scale; deq (rows, 256) bf16 row stride ldd or NULL.  cols must be 256.                       */
int mcl_quant_e4m3_rows(const float* x, int64_t ldx, int32_t rows, int32_t cols, void* q, int64_t ldq, void* scale,
                        int64_t ld_scale, void* deq_bf16, int64_t ldd, mcl_stream_t stream);
int mcl_dequant_e4m3_rows(const void* q, int64_t ldq, const void* scale, int64_t ld_scale, int32_t rows, int32_t cols,
                          void* deq_bf16, int64_t ldd, mcl_stream_t stream);
/* lse[r] = log sum_c exp(inv_temp * sum_k (a8[r][k] 2^(sa[r]-127)) (b8[c][k] 2^(sb[c]-127))), logits never in HBM,
 * fp8 MFMA with hardware block scales, fp32 accumulate and statistics.  a8 (R, 256), b8 (C, 256) e4m3 bytes, row
 * strides multiples of 16.  workspace: mcl_infonce_fp8_workspace_bytes(R, C).                                      */
int64_t mcl_infonce_fp8_workspace_bytes(int32_t R, int32_t C);
int mcl_infonce_fp8_lse(const void* a8, int64_t lda, const void* scale_a, int64_t ld_sa, const void* b8, int64_t ldb,
                        const void* scale_b, int64_t ld_sb, int32_t R, int32_t C, int32_t dim, float inv_temp, float* lse,
                        void* workspace, int64_t ws_bytes, mcl_stream_t stream);
/* diag[r] = inv_temp * a[r] . b[r + diag_off] on bf16 rows (rows whose partner lies outside [0, C) are left untouched) */
int mcl_infonce_rowdot_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, int32_t R, int32_t C, int32_t dim,
                            int32_t diag_off, float inv_temp, float* diag, mcl_stream_t stream);

/* ---------------------------------------------------------------- K10 DenseNet BatchNorm(+ReLU), channels-last
 * Train-mode nn.BatchNorm2d (+ nn.ReLU) of the torchvision DenseNet-121 feature extractor that
 * model.py:75-76 wraps, on NHWC activations viewed as (S = B*H*W rows) x (C channels) with a row stride
 * ld (elements) -- so a dense layer reads its input as a channel slice of the block's concat buffer in
 * place (no torch.cat), per-channel statistics are computed once per produced feature map, and the
 * data gradient is accumulated in place.  dtype: 0 = fp32, 1 = bf16 activations; statistics and affine
 * parameters are fp32.  C must be a multiple of 4 (fp32) / 8 (bf16), bases 16-byte aligned.
 * workspace: mcl_bn_workspace_floats(S, C, dtype) floats.                                        */
int64_t mcl_bn_workspace_floats(int64_t S, int32_t C, int32_t dtype);
/* mean/var (biased)/rstd = 1/sqrt(var+eps) per channel; if copy_out != NULL also copies x there
 * (used to place a produced feature map into its slice of the concat buffer in the same pass).  */
int mcl_bn_stats(const void* x, int64_t ld, int64_t S, int32_t C, int32_t dtype, void* copy_out, int64_t ld_out,
                 float* workspace, float eps, float* mean, float* var, float* rstd, mcl_stream_t stream);
/* y = relu?( (x-mean)*rstd*gamma + beta ) */
int mcl_bn_act_fwd(const void* x, int64_t ldx, int64_t S, int32_t C, int32_t dtype, const float* gamma,
                   const float* beta, const float* mean, const float* rstd, int32_t relu, void* y, int64_t ldy,
                   mcl_stream_t stream);
/* g = dy*[y>0] (or dy); dgamma (+)= sum g*xhat; dbeta (+)= sum g  (accumulate_params != 0 adds into the
 * given buffers -- the parameters' .grad views of the flat optimizer bucket -- instead of overwriting);
 * dx (+)= gamma*rstd*(g - mean(g) - xhat*mean(g*xhat))   (accumulate != 0: read-modify-write of dx) */
int mcl_bn_act_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, int64_t S, int32_t C, int32_t dtype,
                   const float* gamma, const float* beta, const float* mean, const float* rstd, int32_t relu,
                   float* workspace, float* dgamma, float* dbeta, int32_t accumulate_params, void* dx, int64_t lddx,
                   int32_t accumulate, mcl_stream_t stream);

/* DenseNet bottleneck 1x1 convolution with both BatchNorms folded in (torchvision _DenseLayer norm1/relu1/conv1 +
 * norm2's statistics):   z[s][n] = sum_k relu(x[s][k]*g[k]*rstd[k] + beta[k] - mean[k]*g[k]*rstd[k]) * W[n][k],
 * n < 128;  zmean/zvar (biased)/zrstd = batch statistics of the bf16-rounded z.  x: (S, K) bf16 row stride ldx
 * (a channel slice of the block's concat buffer), W: (128, K) bf16 contiguous, z: (S, 128) bf16 row stride ldz.
 * K % 8 == 0, K <= 1024.  workspace: mcl_dense_conv1x1_workspace_floats(S) floats.  zmean = zvar = zrstd = NULL:
 * no statistics (inference, where mean/rstd are the running statistics of norm1 and norm2 uses its own).       */
int64_t mcl_dense_conv1x1_workspace_floats(int64_t S);
int mcl_dense_conv1x1_fwd(const void* x, int64_t ldx, int64_t S, int32_t K, const float* gamma, const float* beta,
                          const float* mean, const float* rstd, const void* W, void* z, int64_t ldz, float* workspace,
                          float eps, float* zmean, float* zvar, float* zrstd, mcl_stream_t stream);

/* DenseNet growth 3x3 convolution (pad 1, 128 -> 32) with both BatchNorms folded in (torchvision _DenseLayer
 * norm2/relu2/conv2 + the statistics of the new feature map), written straight into the concat buffer:
 *   y[p][co] = sum_{ky,kx,ci} relu(bn2(z))[p + (ky-1)*W + (kx-1)][ci] * W2[co][ky][kx][ci]   (zero outside the image)
 * z: (S, 128) bf16 contiguous NHWC pixels (S = B*H*W), W2: (32, 3, 3, 128) bf16 (a channels-last (32,128,3,3)
 * weight), out: bf16 with row stride ldo (the block buffer's channel slice), ymean/yvar (biased)/yrstd = batch
 * statistics of the bf16-rounded y (all three NULL: none, inference).  workspace:
 * mcl_dense_conv3x3_workspace_floats(S) floats.                                                           */
int64_t mcl_dense_conv3x3_workspace_floats(int64_t S);
int mcl_dense_conv3x3_fwd(const void* z, int64_t S, int32_t H, int32_t W, const float* gamma, const float* beta,
                          const float* mean, const float* rstd, const void* W2, void* out, int64_t ldo,
                          float* workspace, float eps, float* ymean, float* yvar, float* yrstd, mcl_stream_t stream);

/* Weight gradient of that 3x3 convolution: dW2[co][ky][kx][ci] (+)= sum_p dy[p][co] * relu(bn2(z))[p + tap][ci], the
 * normalised input recomputed from z on the fly; dW2: (32, 3, 3, 128) fp32 contiguous (the channels-last parameter's
 * .grad).  dy: (S, 32) bf16 row stride lddy; z: (S, 128) bf16 contiguous.  No atomics: every pixel group stores its fp32
 * partial (32 x 1152) in the workspace and a merge launch adds the partials in fixed order -> dW (accumulate_w != 0: +=).
 * Bit-reproducible.  workspace: mcl_dense_conv3x3_wrw_workspace_floats(S) floats.                                  */
int64_t mcl_dense_conv3x3_wrw_workspace_floats(int64_t S);
int mcl_dense_conv3x3_wrw_det(const void* dy, int64_t lddy, const void* z, int64_t S, int32_t H, int32_t W,
                              const float* gamma, const float* beta, const float* mean, const float* rstd,
                              float* workspace, float* dW, int32_t accumulate_w, mcl_stream_t stream);

/* Backward of a dense layer's head x -> norm1 -> relu1 -> conv1 (1x1) -> z with respect to x, fused (csrc/dense_bwd.hip):
 *   da = dz W1 ; g = da*[bn1(x) > 0] ; dgamma (+)= sum g*xhat ; dbeta (+)= sum g ;
 *   gbuf[s, :C] += gamma*rstd*(g - mean(g) - xhat*mean(g*xhat))
 * without materialising da (both passes recompute dz W1 on the matrix cores).  dz: (S, 128) bf16 contiguous, W1:
 * (128, C) bf16 contiguous, x and gbuf: (S, C) bf16 channel slices with row strides ldx / ldg.  accumulate_params
 * != 0 adds dgamma/dbeta into the given buffers (the parameters' .grad views).  C % 8 == 0.
 * workspace: mcl_dense_bn1_bwd_workspace_floats(S, C) floats.                                              */
int64_t mcl_dense_bn1_bwd_workspace_floats(int64_t S, int32_t C);
int mcl_dense_bn1_bwd(const void* dz, const void* W1, int32_t C, const void* x, int64_t ldx, int64_t S,
                      const float* gamma, const float* beta, const float* mean, const float* rstd, float* workspace,
                      float* dgamma, float* dbeta, int32_t accumulate_params, void* gbuf, int64_t ldg,
                      mcl_stream_t stream);

/* Round 6 -- two consecutive layers' dx passes as ONE pass over the channels both read (the BatchNorm-1 backward of the 56 x 56 /
 * 28 x 28 dense blocks re-reads and re-writes the block's whole growing gradient buffer once per layer: O(L^2) bytes).
 *   mcl_dense_bn1_dx_window  mcl_dense_bn1_dx restricted to the channel window [c0, c0 + nc) of the layer's C input channels
 *                            (c0 % 8 == 0): layer l's term on the 32 channels layer l-1 produced, which layer l-1's 3x3
 *                            backward consumes next;
 *   mcl_dense_bn1_dx_pair    gbuf[:, 0:C] += term_A + term_B for layer A = l (input C + 32 channels: W1A rows are ldwA long) and
 *                            layer B = l - 1 (input C channels), x and gbuf read once, gbuf written once; both dz W1 products
 *                            recomputed on the matrix cores; the two deltas are added in fp32 and rounded to bf16 once.
 * Operands as mcl_dense_bn1_dx (coef = that layer's mcl_dense_bn1_wrw output).                                             */
int mcl_dense_bn1_dx_window(const void* dz, const void* W1, int32_t C, int32_t c0, int32_t nc, const void* x, int64_t ldx,
                            int64_t S, const float* gamma, const float* beta, const float* mean, const float* rstd,
                            const float* coef, void* gbuf, int64_t ldg, mcl_stream_t stream);
int mcl_dense_bn1_dx_pair(const void* dzA, const void* W1A, int32_t ldwA, const float* gammaA, const float* betaA,
                          const float* coefA, const void* dzB, const void* W1B, const float* gammaB, const float* betaB,
                          const float* coefB, int32_t C, const void* x, int64_t ldx, int64_t S, const float* mean,
                          const float* rstd, void* gbuf, int64_t ldg, mcl_stream_t stream);

/* Deterministic replacement of the reduce launch of mcl_dense_bn1_bwd AND of the bottleneck weight gradient
 * (csrc/wrw_fused.hip): one pass over (dz, x) forms the Gram matrices R = dz^T mask and Qx = dz^T (mask*x) and from
 * them  dW1 (+)= dz^T relu(bn1(x))  [accumulate_w != 0: += into the parameter's fp32 .grad (128, C) contiguous],
 * dgamma / dbeta (+)= the BatchNorm-backward sums, and coef_out[2c] = mean(g), coef_out[2c+1] = mean(g*xhat) for
 * mcl_dense_bn1_dx.  Slab partials go through the workspace and are merged in fixed order: no atomics,
 * bit-reproducible.  Same operand layouts as mcl_dense_bn1_bwd.  workspace: mcl_wrw_workspace_floats(S, 128, C). */
int64_t mcl_wrw_workspace_floats(int64_t S, int32_t M, int32_t N);
int mcl_dense_bn1_wrw(const void* dz, const void* W1, int32_t C, const void* x, int64_t ldx, int64_t S,
                      const float* gamma, const float* beta, const float* mean, const float* rstd, float* workspace,
                      float* dW, int32_t accumulate_w, float* dgamma, float* dbeta, int32_t accumulate_params,
                      float* coef_out, mcl_stream_t stream);
/* The dx pass of mcl_dense_bn1_bwd alone, with the two per-channel means supplied (coef: 2*C floats).       */
int mcl_dense_bn1_dx(const void* dz, const void* W1, int32_t C, const void* x, int64_t ldx, int64_t S,
                     const float* gamma, const float* beta, const float* mean, const float* rstd, const float* coef,
                     void* gbuf, int64_t ldg, mcl_stream_t stream);
/* Single-pass form of mcl_dense_bn1_bwd for the latency-bound small maps.  dx = gamma*rstd*(g - mean g - xhat*mean(g*xhat)) is
 * linear in the two means, and the statistics (mean, rstd) of a concat-buffer channel are the same for every layer of a dense
 * block.  ONE kernel adds gamma*rstd*g into gbuf, reduces the two sums (no separate reduce pass over dz and x) and, when
 * have_prev != 0, subtracts on the same elements the mean terms of the PREVIOUS pass  kprev[c] = gamma*rstd*(mean g, mean g*xhat)
 * (2*C_total floats, [c][2]); its finalize adds dgamma / dbeta and overwrites kprev[0 .. C) with this layer's terms.  Every
 * layer's mean terms thus reach the channels the next layer reads one pass late, and mcl_dense_bn1_fix applies them to the
 * channels [c0, c0 + nc) the next pass does not cover (c0, nc multiples of 8):  gbuf[s][c] -= K1[c] + K2[c]*xhat[s][c].
 * workspace: mcl_dense_bn1_bwd_workspace_floats(S, C).  Replaces the same torch sequence as mcl_dense_bn1_bwd
 * (/root/reference/model.py:75-76 via torchvision _DenseLayer).                                                        */
int mcl_dense_bn1_dx_sums(const void* dz, const void* W1, int32_t C, const void* x, int64_t ldx, int64_t S,
                          const float* gamma, const float* beta, const float* mean, const float* rstd, float* workspace,
                          float* dgamma, float* dbeta, int32_t accumulate_params, float* kprev, int32_t have_prev, void* gbuf,
                          int64_t ldg, mcl_stream_t stream);
int mcl_dense_bn1_fix(const void* x, int64_t ldx, void* gbuf, int64_t ldg, int64_t S, int32_t c0, int32_t nc,
                      const float* mean, const float* rstd, const float* kacc, mcl_stream_t stream);
/* Deterministic 1x1 weight gradient dW[M][N] (+)= dz[S][M]^T a'[S][N]: a' = a (gamma..rstd NULL: transition
 * convolutions) or relu(BatchNorm(a)) recomputed in registers (all four given) -- the atomics-free form of
 * mcl_conv1x1_wrw_bf16.  workspace: mcl_wrw_workspace_floats(S, min(M,128), N).                                 */
int mcl_conv1x1_wrw_det(const void* dz, int64_t ldz, const void* a, int64_t lda, const float* gamma, const float* beta,
                        const float* mean, const float* rstd, float* workspace, float* dW, int32_t accumulate_w,
                        int64_t S, int32_t M, int32_t N, mcl_stream_t stream);

/* Backward of a dense layer's tail z -> norm2 -> relu2 -> conv2 (3x3, pad 1, 128 -> 32) with respect to z, fused:
 *   da2 = conv3x3 backward-data of dy ; g2 = da2*[bn2(z) > 0] ; dgamma2 (+)= sum g2*zhat ; dbeta2 (+)= sum g2 ;
 *   dz = gamma2*rstd2*(g2 - mean(g2) - zhat*mean(g2*zhat)).
 * dy: (S, 32) bf16 row stride lddy (the gradient buffer's channel slice, read in place); W2: (32, 3, 3, 128) bf16;
 * z: (S, 128) bf16 contiguous NHWC pixels; g2 (scratch) and dz: (S, 128) bf16 contiguous.
 * workspace: mcl_dense_conv3x3_bwd_workspace_floats(S) floats.                                            */
int64_t mcl_dense_conv3x3_bwd_workspace_floats(int64_t S);
int mcl_dense_conv3x3_bwd(const void* dy, int64_t lddy, int64_t S, int32_t H, int32_t W, const void* W2, const void* z,
                          const float* gamma, const float* beta, const float* mean, const float* rstd, float* workspace,
                          float* dgamma, float* dbeta, int32_t accumulate_params, void* g2, void* dz, mcl_stream_t stream);
/* mcl_dense_conv3x3_bwd with mcl_dense_bn1_fix folded into the dy staging (single-pass BatchNorm-1 backward, maps narrower
 * than 17 pixels): dy' = dy - (K1 + K2*xhat) on the layer's 32 output channels (xfix: those channels of the concat buffer,
 * row stride ldxf; fmean / frstd: their statistics; fk: [32][2] mean terms of the previous pass), used for the data
 * gradient and written to dyc (S x 32 bf16 contiguous) for mcl_dense_conv3x3_wrw_det.  Bit-identical to the two launches. */
int mcl_dense_conv3x3_bwd_fix(const void* dy, int64_t lddy, int64_t S, int32_t H, int32_t W, const void* W2, const void* z,
                              const float* gamma, const float* beta, const float* mean, const float* rstd, float* workspace,
                              float* dgamma, float* dbeta, int32_t accumulate_params, void* g2, void* dz, const void* xfix,
                              int64_t ldxf, const float* fmean, const float* frstd, const float* fk, void* dyc,
                              mcl_stream_t stream);

/* Pooling layers of the DenseNet stem / transitions on channels-last bf16 (C % 8 == 0, 16-byte aligned, dense NHWC).
 * mcl_avgpool2_nhwc_bf16: AvgPool2d(2, 2); backward == 0: x (N,H,W,C) -> y (N,H/2,W/2,C); backward != 0: x is dy
 *   (N,H/2,W/2,C) and y receives dx (N,H,W,C) = dy/4.  H, W even.
 * mcl_maxpool3s2_*: MaxPool2d(3, stride 2, padding 1); the forward also records idx (one byte per output element:
 *   window position of the first maximum in row-major order, ATen's tie rule); the backward GATHERS dy through it:
 *   deterministic, no atomics.                                                                             */
int mcl_avgpool2_nhwc_bf16(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t backward,
                           mcl_stream_t stream);
int mcl_maxpool3s2_nhwc_bf16_fwd(const void* x, void* y, void* idx, int32_t N, int32_t H, int32_t W, int32_t C,
                                 mcl_stream_t stream);
int mcl_maxpool3s2_nhwc_bf16_bwd(const void* idx, const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                                 mcl_stream_t stream);
/* the same with dy addressed through a row stride (elements per pooled pixel): the channel slice [:C] of a wider gradient
 * buffer is read in place (no contiguous copy of the first dense block's input gradient).                             */
int mcl_maxpool3s2_nhwc_bf16_bwd_ld(const void* idx, const void* dy, int64_t lddy, void* dx, int32_t N, int32_t H, int32_t W,
                                    int32_t C, mcl_stream_t stream);

/* DenseNet transition (torchvision _Transition: norm -> relu -> conv1x1 -> AvgPool2d(2,2)) with the pool moved in
 * front of the (linear, per-pixel) convolution:  p = avgpool2x2(relu(bn(x))), bf16 NHWC, x (N*H*W, C) with row stride
 * ldx, p (N*H/2*W/2, C) with row stride ldy; H, W even.  The convolution then runs on p (mcl_dense_conv1x1_fwd with an
 * identity prologue), a quarter of the pixels.
 * Backward: dp = gradient of p (pooled rows, row stride lddp); every pixel receives dp[pooled pixel]/4 through the
 * relu mask and the full train-mode BatchNorm backward: dgamma/dbeta (accumulated into when accumulate_params != 0)
 * and dx (written, row stride lddx).  workspace: mcl_bn_workspace_floats(N*H*W, C, 1) floats.                  */
int mcl_bn_act_avgpool_fwd(const void* x, int64_t ldx, int32_t N, int32_t H, int32_t W, int32_t C, const float* gamma,
                           const float* beta, const float* mean, const float* rstd, void* y, int64_t ldy,
                           mcl_stream_t stream);
int mcl_bn_act_avgpool_bwd(const void* dp, int64_t lddp, const void* x, int64_t ldx, int32_t N, int32_t H, int32_t W,
                           int32_t C, const float* gamma, const float* beta, const float* mean, const float* rstd,
                           float* workspace, float* dgamma, float* dbeta, int32_t accumulate_params, void* dx,
                           int64_t lddx, mcl_stream_t stream);
/* DenseNet stem convolution conv0 (7x7, stride 2, pad 3, 3 -> 64; torchvision densenet121.features.conv0) with the
 * batch statistics of its output for norm0:  x (N,H,W,3) bf16 NHWC contiguous, Wt (64,7,7,3) bf16 (a channels-last
 * (64,3,7,7) weight), y (N,H/2,W/2,64) bf16 NHWC; mean/var (biased)/rstd of the bf16-rounded y (all three NULL:
 * none).  H % 4 == 0, W % 8 == 0, W <= 256.  workspace: mcl_conv0_workspace_floats(N, H, W) floats.            */
int64_t mcl_conv0_workspace_floats(int32_t N, int32_t H, int32_t W);
int mcl_conv0_fwd(const void* x, int32_t N, int32_t H, int32_t W, const void* Wt, void* y, float* workspace, float eps,
                  float* mean, float* var, float* rstd, mcl_stream_t stream);
/* Weight gradient of conv0: dW (64,7,7,3) fp32 (the channels-last parameter's .grad) (+)= sum_p dy[p] (x) patch(x)[p].
 * dy: (N,H/2,W/2,64) bf16 NHWC contiguous.  H % 4 == 0, W % 8 == 0, W <= 256.  workspace:
 * mcl_conv0_wrw_workspace_floats floats -- per-workgroup partials merged in fixed order (deterministic);
 * accumulate_w != 0 adds into dW, else overwrites.                                                              */
int64_t mcl_conv0_wrw_workspace_floats(int32_t N, int32_t H, int32_t W);
int mcl_conv0_wrw(const void* x, int32_t N, int32_t H, int32_t W, const void* dy, float* workspace, float* dW,
                  int32_t accumulate_w, mcl_stream_t stream);

/* DenseNet stem tail norm0 -> relu0 -> pool0 (MaxPool2d(3, 2, 1)) in one pass over the conv0 output x (N,H,W,C) bf16
 * NHWC contiguous: y (N,OH,OW,C) = maxpool(relu(bn(x))), idx = arg-max byte per pooled element (ky*3+kx, first maximum
 * in window order).  Backward: mcl_maxpool3s2_nhwc_bf16_bwd(idx, dy) followed by mcl_bn_act_bwd(relu = 1) on x
 * (fusing the gather into both BatchNorm-backward passes was measured slower: 388 us vs 217 us at 128 x 112^2 x 64). */
int mcl_bn_act_maxpool_fwd(const void* x, int32_t N, int32_t H, int32_t W, int32_t C, const float* gamma,
                           const float* beta, const float* mean, const float* rstd, void* y, void* idx,
                           mcl_stream_t stream);
/* dst[i] += (float)src[i], i < n, in storage order (src_dtype 0 = fp32, 1 = bf16): adds a low-precision
 * weight gradient into the fp32 .grad view of the flat optimizer bucket (both dense, identical strides). */
int mcl_accum_into_f32(float* dst, const void* src, int64_t n, int32_t src_dtype, mcl_stream_t stream);

/* ---------------------------------------------------------------- the last pieces of the step (csrc/step_misc.hip)
 * mcl_bn_gap_fwd: the tail of the DenseNet feature extractor, norm5 -> adaptive_avg_pool2d((1,1)) -> flatten
 *   (/root/reference/model.py:81-85; no ReLU): out[b][c] = gamma*rstd*(mean_hw x[b][hw][c] - mean[c]) + beta, fp32 (B, C);
 *   x = (B, HW, C) bf16 rows of stride ldx; xmean (B, C) fp32 (may be NULL) receives the raw pooled means for the backward.
 * mcl_bn_gap_bwd: from g = dL/dout (B, C) fp32: dgamma / dbeta (accumulate_params != 0: +=), coef (2C floats of scratch:
 *   the two BatchNorm-backward means) and dx (B, HW, C) bf16 = gamma*rstd*(g/HW - mean(dy) - xhat*mean(dy*xhat)) -- the
 *   train-mode BatchNorm backward with dy = g/HW broadcast over the map.  Deterministic (fixed-order sums over B).
 * mcl_bn_running_update: nn.BatchNorm2d's bookkeeping for n layers (HOST arrays of n device pointers / values):
 *   running_mean.lerp_(mean, momentum); running_var.lerp_(var * factor, momentum)  (factor = count / (count - 1): unbiased);
 *   num_batches_tracked += 1 (entries may be NULL).  The pointer table travels by value in the kernel arguments, 64 layers
 *   per launch.
 * mcl_image_to_bf16_nhwc: y (B, H, W, C) bf16 contiguous = x[b*sb + c*sc + y*sy + x*sx] (fp32, element strides): the
 *   ``image.to(bf16).contiguous(channels_last)`` in front of the stem for NCHW and channels-last inputs alike.
 * mcl_fill_zero: bytes (a multiple of 4, p 4-byte aligned) of zeros written by a kernel -- not hipMemsetAsync: a memset
 *   node inside the captured step graph cost the step its two-lane execution (measured).                              */
int mcl_bn_gap_fwd(const void* x, int64_t ldx, int32_t B, int32_t HW, int32_t C, const float* gamma, const float* beta,
                   const float* mean, const float* rstd, float* out, float* xmean, mcl_stream_t stream);
int mcl_bn_gap_bwd(const float* g, const float* xmean, const void* x, int64_t ldx, int32_t B, int32_t HW, int32_t C,
                   const float* gamma, const float* mean, const float* rstd, float* coef, float* dgamma, float* dbeta,
                   int32_t accumulate_params, void* dx, int64_t lddx, mcl_stream_t stream);
/* eval mode (evel_her2st.py:48-50: model.eval()): rstd_out[k][c] = 1 / sqrt(running_var[k][c] + eps[k]) for n BatchNorm
 * layers (HOST arrays of n device pointers / values), 64 layers per launch. */
int mcl_bn_eval_rstd(int32_t n, const float* const* running_var, float* const* rstd_out, const int32_t* C, const float* eps,
                     mcl_stream_t stream);
int mcl_bn_running_update(int32_t n, float* const* running_mean, float* const* running_var, const float* const* mean,
                          const float* const* var, int64_t* const* num_batches_tracked, const int32_t* C,
                          const float* factor, const float* momentum, mcl_stream_t stream);
int mcl_image_to_bf16_nhwc(const float* x, int64_t sb, int64_t sc, int64_t sy, int64_t sx, int32_t B, int32_t C, int32_t H,
                           int32_t W, void* y, mcl_stream_t stream);
int mcl_fill_zero(void* p, int64_t bytes, mcl_stream_t stream);
/* nn.Dropout(p) / nn.GELU() / the residual add of the dropout > 0 path (model.py:25-29,156,164-165; never active in the
 * reference, model.py:217) as own elementwise kernels on contiguous fp32.  mcl_dropout_fwd: mask byte = 1 where kept (a
 * counter-based generator of (seed, element index): reproducible per seed), y = kept ? x / (1 - p) : 0; mcl_dropout_bwd:
 * dx = mask ? dy / (1 - p) : 0.  mcl_gelu_f32: out = dy ? dy * gelu'(x) : gelu(x) (exact erf form).                       */
int mcl_dropout_fwd(const float* x, float* y, void* mask, int64_t n, float p, uint64_t seed, mcl_stream_t stream);
int mcl_dropout_bwd(const float* dy, const void* mask, float* dx, int64_t n, float p, mcl_stream_t stream);
int mcl_gelu_f32(const float* x, const float* dy, float* out, int64_t n, mcl_stream_t stream);
int mcl_add_f32(const float* a, const float* b, float* y, int64_t n, mcl_stream_t stream);
/* debug: buf[idx] (uint64) = the GPU wall clock (100 MHz) when the stream reaches this point (MCL_STAMPS=1 timelines). */
int mcl_stamp(void* buf, int32_t idx, mcl_stream_t stream);
/* ya[i] = a[i] * s[0] (i < na), yb[i] = b[i] * s[0] (i < nb): a loss gradient scaled by autograd's upstream scalar (a
 * DEVICE value: no host read) for both embedding gradients in one launch.                                             */
int mcl_scale2_f32(const float* a, int64_t na, const float* b, int64_t nb, const float* s, float* ya, float* yb,
                   mcl_stream_t stream);

/* ---------------------------------------------------------------- generic convolution lowering + pooling, fp32 AND bf16
 * (csrc/im2col.hip, csrc/pool_generic.hip; dtype 0 = fp32, 1 = bf16).  For the shapes the specialised DenseNet kernels do not
 * cover: fp32 activations (the reference-numerics mode of the backbone, /root/reference/model.py:72-85) and the ResNet
 * encoders (/root/reference/model.py:88-148).  A convolution is im2col + this library's GEMM (mcl_gemm / mcl_gemm_bf16):
 *   mcl_im2col_nhwc: cols[(n,oy,ox)][(ky,kx,c)] = x[n][oy*stride-pad+ky][ox*stride-pad+kx][c] (0 outside); x rows of stride
 *     ldx (a channel slice is read in place); cols (N*OH*OW, KH*KW*C) contiguous; the column order is the storage order of a
 *     channels-last weight (C_out, kh, kw, C_in).
 *   mcl_col2im_nhwc: the transpose (backward-data): dx[n][y][x][c] (+)= the sum of the columns that reference the pixel, as a
 *     gather in fixed tap order (deterministic); dx rows of stride lddx.
 *   mcl_maxpool3s2_nhwc_{fwd,bwd}_any: MaxPool2d(3, 2, 1) with ATen's first-maximum tie rule, idx = one byte per element.
 *   mcl_avgpool2_nhwc_any: AvgPool2d(2, 2) (floor on odd maps) forward / backward.
 *   mcl_gap_nhwc_{fwd,bwd}: adaptive_avg_pool2d((1,1)) + flatten -> (B, C) fp32, and its backward g/HW broadcast.
 *   mcl_add_relu: y = relu(a + b) (backward != 0: y = a * [b > 0] with a = dy, b = the forward output).                 */
int mcl_im2col_nhwc(const void* x, int64_t ldx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t KH, int32_t KW,
                    int32_t stride, int32_t pad, int32_t dtype, void* cols, mcl_stream_t stream);
int mcl_col2im_nhwc(const void* dcols, int32_t N, int32_t H, int32_t W, int32_t C, int32_t KH, int32_t KW, int32_t stride,
                    int32_t pad, int32_t dtype, void* dx, int64_t lddx, int32_t accumulate, mcl_stream_t stream);
int mcl_maxpool3s2_nhwc_fwd_any(const void* x, void* y, void* idx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype,
                                mcl_stream_t stream);
int mcl_maxpool3s2_nhwc_bwd_any(const void* idx, const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                                int32_t dtype, mcl_stream_t stream);
int mcl_avgpool2_nhwc_any(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t backward, int32_t dtype,
                          mcl_stream_t stream);
int mcl_gap_nhwc_fwd(const void* x, int64_t ldx, int32_t B, int32_t HW, int32_t C, int32_t dtype, float* out,
                     mcl_stream_t stream);
int mcl_gap_nhwc_bwd(const float* g, int32_t B, int32_t HW, int32_t C, int32_t dtype, void* dx, mcl_stream_t stream);
/* backward: 0 = y = relu(a + b); 1 = dx = dy*[y > 0] (a = dy, b = forward output); 2 = y = a + b (gradient of a tensor with
 * two consumers: the fork in front of a residual block). */
int mcl_add_relu(const void* a, const void* b, void* y, int64_t n, int32_t backward, int32_t dtype, mcl_stream_t stream);

/* ---------------------------------------------------------------- a whole dense block, forward, 7 x 7 maps (ABI 6)
 * torchvision _DenseBlock (/root/reference/model.py:75-76: densenet121(...).features.denseblock4) in train mode as ONE
 * persistent launch: replaces, per layer, mcl_dense_conv1x1_fwd + mcl_dense_conv3x3_fwd (and their statistics finalize
 * launches) on H = W = 7 maps.  One workgroup per image (B <= compute units, all resident); the batch statistics are
 * exchanged in-launch (two all-to-all seams per layer, write-through stores + flags, fixed-order merges: deterministic).
 *   buf        (B, 7, 7, Ct) bf16 NHWC concat buffer, Ct = C0 + 32*L <= 1024; channels [0, C0) hold the block input on entry,
 *              channels [C0, Ct) are written;
 *   layer_ptrs HOST array of 10*L pointers, per layer: norm1.weight, norm1.bias (fp32 [C_in]), conv1 weight PACKED by
 *              mcl_dense_block_pack_w1 (bf16, 128*C_in elements in the kernel's MFMA-fragment streaming order),
 *              norm2.weight, norm2.bias (fp32 [128]), conv2 weight (bf16 [32][3][3][128]), z out (bf16 (B, 7, 7, 128), saved
 *              for the backward), norm2 batch mean / biased var / rstd out (fp32 [128] each);
 *   mean / var / rstd  fp32 [Ct] batch statistics of the concat channels: [0, C0) given (mean, rstd read), [C0, Ct) written;
 *   workspace  mcl_dense_block_fwd_workspace_bytes(B, L) bytes, 256-byte aligned; err_flag: device int32, set to 1 if a seam
 *              wait timed out (results invalid; never hangs).  The HOST must read it at its next synchronisation and discard
 *              the step (mclstexp_amd.ops.check_device_errors raises SeamTimeoutError);
 *   max_spins  bound of every seam poll loop (0 = the default, 2^19 polls ~ 0.5 s); a test passes 1 to force the timeout path;
 *   stamps     NULL, or a device buffer of B*L*8 uint64 that the launch fills with in-kernel phase stamps (100 MHz wall
 *              clock) per image and layer -- per call: the library keeps no state between calls.
 * Algorithmic bytes: the block input read once, z and the new channels written once (+ the weights).                      */
/* conv1 weights (bf16 [128][C_in_l], C_in_l = C0 + 32*l, k-contiguous rows) of the L layers -> the packed order (same sizes);
 * w1_ptrs / out_ptrs: HOST arrays of L device pointers.  One launch; run it whenever the weights have changed. */
int mcl_dense_block_pack_w1(const void* const* w1_ptrs, void* const* out_ptrs, int32_t L, int32_t C0, mcl_stream_t stream);
int64_t mcl_dense_block_fwd_workspace_bytes(int32_t B, int32_t L);
int mcl_dense_block_fwd(void* buf, int32_t B, int32_t H, int32_t W, int32_t Ct, int32_t C0, int32_t L,
                        const void* const* layer_ptrs, float eps1, float eps2, float* mean, float* var, float* rstd,
                        void* workspace, int32_t* err_flag, uint32_t max_spins, void* stamps, mcl_stream_t stream);

/* The same dense block BACKWARD (7 x 7 maps) as one persistent launch: replaces, per layer, mcl_dense_conv3x3_bwd_fix and
 * mcl_dense_bn1_dx_sums (+ their finalize launches and bn2_dz) with the same arithmetic; the two weight gradients of a layer
 * stay mcl_dense_conv3x3_wrw_det / mcl_conv1x1_wrw_det on the dz / dyc tensors this launch writes.
 *   buf        the forward's concat buffer (B, 7, 7, Ct) bf16;  gbuf: the block's gradient buffer, same shape: read whole,
 *              channels [0, C0) (the gradient of the block input) written back;
 *   layer_ptrs HOST array of 15*L pointers, per layer: norm1.weight, norm1.bias, conv1 weight packed by
 *              mcl_dense_block_pack_bwd (w1t), norm2.weight, norm2.bias, conv2 weight packed (w2t), z (forward, (B,7,7,128)),
 *              norm2 batch mean, norm2 batch rstd, dz out (B,7,7,128), dyc out (B,7,7,32: the gradient of the layer's output as
 *              consumed), then the fp32 gradients of norm1.weight, norm1.bias, norm2.weight, norm2.bias (ACCUMULATED into);
 *   mean / rstd  fp32 [Ct] batch statistics of the concat channels (forward);
 *   workspace  mcl_dense_block_bwd_workspace_bytes(B, L) bytes, 256-byte aligned; err_flag / max_spins / stamps as in the forward. */
int mcl_dense_block_pack_bwd(const void* const* w1_ptrs, const void* const* w2_ptrs, void* const* w1t_ptrs,
                             void* const* w2t_ptrs, int32_t L, int32_t C0, mcl_stream_t stream);
int64_t mcl_dense_block_bwd_workspace_bytes(int32_t B, int32_t L);
int mcl_dense_block_bwd(const void* buf, void* gbuf, int32_t B, int32_t H, int32_t W, int32_t Ct, int32_t C0, int32_t L,
                        const void* const* layer_ptrs, const float* mean, const float* rstd, void* workspace,
                        int32_t* err_flag, uint32_t max_spins, void* stamps, mcl_stream_t stream);

/* ---------------------------------------------------------------- K9 Adam with L2 weight decay
 * torch.optim.Adam(lr, betas, eps, weight_decay) as used by train.py:118-120, one fused pass:
 *   g += wd*p ; m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ;
 *   p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps),   bc1 = 1-b1^t, bc2 = 1-b2^t (host-computed).
 * Hyper-parameters are doubles (Python floats in torch.optim) and rounded to fp32 once, so that
 * 1-beta matches torch's double-computed constant.
 * Flat fp32 buffers of n elements (28 B/element of HBM traffic).                               */
int mcl_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                  double eps, double weight_decay, double bc1, double bc2, mcl_stream_t stream);
/* Same update for an embedding table (n_rows x cols, contiguous) whose data gradient is row-sparse:
 * g = wd*p + (row_slot[r] >= 0 ? row_grad[row_slot[r],:] : 0).  24 B/element: the dense zero
 * gradient of the reference (model.py:204-205 under autograd) is never materialised.
 * row_slot (n_rows int32) maps table row -> slot in row_grad or -1.                            */
int mcl_adam_table_step(float* p, float* m, float* v, int32_t n_rows, int32_t cols, const int32_t* row_slot,
                        const float* row_grad, int64_t ld_rg, double lr, double beta1, double beta2, double eps,
                        double weight_decay, double bc1, double bc2, mcl_stream_t stream);

/* Graph-replayable Adam: the step counter (one int64) and the derived constants (8 floats) live on the device.
 * mcl_adam_consts_update advances the counter and refreshes the constants (one thread, double arithmetic, rounded once
 * as the host-constant entry points do); the _dev kernels read them, so nothing step-dependent is baked into a launch
 * and the whole optimizer step can sit inside a captured HIP graph.  `hyper` = DEVICE pointer to 5 doubles
 * {lr, beta1, beta2, eps, weight_decay}: the host refreshes them with an ordinary copy between replays, so an LR schedule
 * (torch.optim.lr_scheduler / a manual param_groups edit) is honoured by a captured step (ABI 2; ABI 1 took them by value
 * and froze them into the graph).                                                                                     */
int mcl_adam_consts_update(int64_t* step, float* consts, const double* hyper, mcl_stream_t stream);
int mcl_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* consts, mcl_stream_t stream);
int mcl_adam_table_step_dev(float* p, float* m, float* v, int32_t n_rows, int32_t cols, const int32_t* row_slot,
                            const float* row_grad, int64_t ld_rg, const float* consts, mcl_stream_t stream);
/* mcl_adam_step_dev with a bf16 shadow: additionally stores round-to-nearest-even(p) into shadow_bf16[i] (the flat low-
 * precision weight copy the backbone kernels read), so no separate cast pass over the parameters runs after the step. */
int mcl_adam_step_dev_shadow(float* p, const float* g, float* m, float* v, int64_t n, const float* consts,
                             void* shadow_bf16, mcl_stream_t stream);
/* Lazy-exact position-table Adam (ABI 6).  Replaces, for the two (65536, G) tables of /root/reference/model.py:204-205 under
 * /root/reference/train.py:118-120, the dense pass above: with L2 weight decay every row moves on every step, but a row
 * without a data gradient follows a recurrence in its own (p, m, v) and the step constants only.  Each row carries a
 * "valid through step" stamp (row_step, n_rows int32, 0 = initial state); a row is brought up to date by replaying its missed
 * steps in registers -- the same fp32 operation sequence as the dense kernel, with the constants those steps used, kept in a
 * ring `hist` of hist_len (power of two) x 8 floats that mcl_adam_consts_update_hist fills (slot = step & (hist_len - 1)) --
 * so the result is bit-identical to running mcl_adam_table_step_dev on every step.  The caller guarantees that no row is
 * more than hist_len - 1 steps behind (materialise in time).  One entry point, three uses; `step` = device step counter:
 *   catch-up of gathered rows (forward): pos != NULL ((n_owner, 2) fp32 positions; table 0 takes column 0, table 1 column 1);
 *       rows are advanced through *step; duplicates are resolved inside (first spot naming a row owns it);
 *   step with data gradients: owner0/owner1 (n_owner int32: row index or -1, as mcl_embed_rowgrad writes them) and
 *       row_grad0/row_grad1 ((n_owner, cols) fp32, leading dimension ld_rg); *step is the step being applied (already advanced
 *       by mcl_adam_consts_update_hist): rows are replayed through *step - 1, then updated with g = wd*p + row_grad;
 *   materialise: pos, owner*, row_grad* all NULL and n_owner == n_rows: every row is advanced through *step.
 * Table 1 is optional (p1 == NULL).  Algorithmic bytes: 24 B x touched rows x cols (+ the gradient rows).                */
int mcl_adam_consts_update_hist(int64_t* step, float* consts, const double* hyper, float* hist, int32_t hist_len,
                                mcl_stream_t stream);
int mcl_adam_table_lazy(float* p0, float* m0, float* v0, int32_t* row_step0, float* p1, float* m1, float* v1,
                        int32_t* row_step1, int32_t n_rows, int32_t cols, const float* pos, const int32_t* owner0,
                        const int32_t* owner1, int32_t n_owner, const float* row_grad0, const float* row_grad1, int64_t ld_rg,
                        const int64_t* step, const float* hist, int32_t hist_len, mcl_stream_t stream);
/* row_slot maintenance: set row_slot[owner_idx[b]] = b for owners (fill != 0) or back to -1.    */
int mcl_row_slot_update(int32_t* row_slot, const int32_t* owner_idx, int32_t B, int32_t fill, mcl_stream_t stream);

/* ---------------------------------------------------------------- inference-time retrieval
 * Replaces, on the device, the reference's find_matches() + weighting loop:
 *   evel_her2st.py:74-84,174-187 (top 200, L1 distance), evel_cscc.py:74-84,197-215 (top 600, values returned,
 *   L2 distance), evel_visium.py:94-104,193-205 (top 200, L2 distance).
 * mcl_l2_normalize_rows: y[r,:] = x[r,:] / max(||x[r,:]||_2, 1e-12)   (F.normalize(p=2, dim=-1)).
 *   The similarity matrix itself is mcl_gemm(query_n, key_n^T) with MCL_COMPUTE_F32.
 * mcl_topk_rows: torch.topk(sim, k) along each row of sim (rows x n, leading dimension ld): values (rows x k) best
 *   first and int64 indices; exact selection (radix select on the fp32 bit patterns), equal values ordered by
 *   ascending index.  k <= mcl_topk_rows_max_k() (2048), k <= n.
 * mcl_knn_weighted_average: per query i with neighbours idx = indices[i,:] (k of them):
 *     a_j = || spot_key[idx_j,:] - query[i,:] ||_ord   (ord = 1 or 2, UN-normalised embeddings, fp32)
 *     w_j = a_j^-2 / sum_j a_j^-2
 *     emb_pred[i,:]  = sum_j w_j spot_key[idx_j,:]        (n_query x dim,   may be NULL)
 *     expr_pred[i,:] = sum_j w_j expression_key[idx_j,:]  (n_query x genes, may be NULL)
 *   (np.average(..., weights=w)); the weighted sums accumulate in fp64.  An exact match (a_j = 0) yields a NaN row,
 *   as numpy's inf/inf does.                                                                     */
int mcl_l2_normalize_rows(const float* x, int64_t ldx, float* y, int64_t ldy, int32_t rows, int32_t dim,
                          mcl_stream_t stream);
int mcl_topk_rows_max_k(void);
int mcl_topk_rows(const float* sim, int64_t ld, int32_t rows, int32_t n, int32_t k, float* values, int64_t* indices,
                  mcl_stream_t stream);
/* top-k of CANDIDATE LISTS (mcl_gemm's threshold filter): row i holds n (value, original column) pairs in arbitrary order, unused
 * slots -inf.  Returns the original columns, equal values ordered by original column like mcl_topk_rows.  A row whose k-th value is
 * an exact tie that would have to be cut by original column sets tie_flag[i] = 1 (its output is then unspecified: recompute that
 * row with mcl_topk_rows on the materialised similarities). */
int mcl_topk_rows_indexed(const float* cand_val, int64_t ld, const int32_t* cand_idx, int64_t ld_idx, int rows, int n, int k,
                          float* values, int64_t* indices, int32_t* tie_flag, mcl_stream_t stream);
int mcl_knn_weighted_average(const float* spot_key, int64_t ldk, const float* expression_key, int64_t lde,
                             const float* query, int64_t ldq, const int64_t* indices, int32_t n_query, int32_t k,
                             int32_t dim, int32_t genes, int32_t ord, float* emb_pred, float* expr_pred,
                             mcl_stream_t stream);

/* ---------------------------------------------------------------- input pipeline on the GPU
 * mcl_patch_gather: the reference's per-spot patch extraction (dataset.py:226-231 PIL crop + transforms.ToTensor;
 *   dataset.py:330-336 numpy crop of the cv2 image + TenxDataset.transform) for a whole batch: image_u8 (Hs, Ws, 3)
 *   uint8 resident in HBM, centers_rc (N, 2) int32 (row, col), patch side 2r.  Source pixels outside the image read 0.
 *   ops (N bytes, may be NULL): bit 0 horizontal flip, bit 1 vertical flip, bits 2-3 = k: then a counter-clockwise
 *   rotation by k*90 degrees (TF.hflip -> TF.vflip -> TF.rotate(angle), angle = 90 k).  Values are divided by divisor
 *   (255: ToTensor's .div(255); 1: Tenx) and written as fp32 NCHW (N,3,2r,2r) and/or bf16 NHWC (N,2r,2r,3); either may be NULL.
 * mcl_log_library_size_normalize: out = log10(counts / rowsum(counts) * rescale + 1) (scprep.transform.log(
 *   scprep.normalize.library_size_normalize(.)), dataset.py:188-189; rescale 1e4); all-zero rows stay zero.      */
int mcl_patch_gather(const void* image_u8, int32_t Hs, int32_t Ws, const int32_t* centers_rc, int32_t N, int32_t r,
                     const void* ops, float divisor, float* out_nchw_f32, void* out_nhwc_bf16, mcl_stream_t stream);
int mcl_log_library_size_normalize(const float* counts, int64_t ldx, float* out, int64_t ldy, int32_t rows,
                                   int32_t cols, float rescale, mcl_stream_t stream);

/* HER2ST / cSCC training transform (/root/reference/dataset.py:63-68: ColorJitter(0.5,0.5,0.5), RandomHorizontalFlip,
 * RandomRotation(180), ToTensor) for a batch of patches cropped from the resident slide, with the random draws supplied:
 * params = N records of 16 32-bit words {order (2 bits per step: 0 brightness, 1 contrast, 2 saturation), hflip,
 * rot_mode (0 quarter turns / 1 affine), rot_k, brightness, contrast, saturation (float), a[6] = libImaging's 16.16
 * fixed-point affine coefficients, 3 pad}.  Bit-exact with PIL (ImageEnhance / Image.rotate NEAREST).  2r <= 232.   */
int mcl_her2st_train_patches(const void* image_u8, int32_t Hs, int32_t Ws, const int32_t* centers_rc, int32_t N, int32_t r,
                             const void* params, float divisor, float* out_nchw_f32, void* out_nhwc_bf16,
                             mcl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MCLSTEXP_HIP_H */
