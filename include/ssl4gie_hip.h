/* libssl4gie_hip.so — C ABI of the MI355X (gfx950) compute path for the SSL4GIE hot path.
 *
 * The reference (ESandML/SSL4GIE) is pure Python/PyTorch and has no native boundary of its own:
 * its hot path is the ATen ops reached from timm `Block`/`PatchEmbed` (un-vendored) and
 * `Models/mae/models_mae.py`, `Models/models.py`.  Every entry point below names the reference
 * call site(s) whose arithmetic it replaces.  Conventions:
 *   - plain C types only; device pointers + explicit sizes/strides + a HIP stream (`void*`);
 *   - no per-call allocation, nothing freed, no host synchronisation; workspaces are
 *     caller-allocated and sized by the matching `*_workspace_bytes()` query.  Three objects are
 *     created once per device on first use and live as long as the library: the 256-byte zero page
 *     of the implicit convolution, the weight-gradient side stream + its events
 *     (ssl4gie_set_wgrad_stream), and — opt-in — the launch profiler (ssl4gie_prof_*); the direct
 *     all-reduce owns one IPC exchange region per handle (ssl4gie_allreduce_direct_init / _destroy);
 *   - returns 0 on success, SSL4GIE_EARG (1000) for an invalid argument, otherwise a hipError_t;
 *   - callable from any host thread; the only mutable process-wide settings are the execution
 *     options ssl4gie_set_wgrad_stream / ssl4gie_set_compute_cus and the profiler;
 *   - ssl4gie_abi_version() = 7 (6: before ssl4gie_bn_coef_stats / ssl4gie_bn_apply_bits / ssl4gie_bn_bwd_reduce_bits existed — additions only; 5: before SSL4GIE_PROF_KINDS grew from 5 to 7 — the profiler's arrays; 1: before ssl4gie_gemm_desc gained `colsum_a` / `conv`; 2: before
 *     ssl4gie_block_bwd's `accumulate` became a flag word and the grouped / deferred weight-gradient
 *     entry points existed; 3: before the direct transport's error word / time-out / all-gather,
 *     ssl4gie_bn_combine_stats and ssl4gie_debug_nt256_stamps existed — additions only; 4: before
 *     ssl4gie_gemm_desc gained `scale` / `relu` (appended; SSL4GIE_EPI_AFFINE_AUX_RELU and the
 *     statistics-only product with C == NULL) and ssl4gie_bn_bwd_xmask / ssl4gie_bn_coef_partials / ssl4gie_bn_maxpool3x3s2_fwd /
 *     ssl4gie_conv3x3_direct_{fwd,wgrad}_affine / ssl4gie_bn_fwd_partials_bits / ssl4gie_bn_bwd_bits /
 *     ssl4gie_bn_bwd_{reduce,apply}_xmask / ssl4gie_conv3x3_weight_pack_batch existed);
 *   - "lp" tensors are the MFMA operand type: SSL4GIE_BF16 for the production path,
 *     SSL4GIE_F32 for the exact-fp32 parity path (f32 MFMA, bit-level fp32 FMA chains).
 */
#ifndef SSL4GIE_HIP_H
#define SSL4GIE_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SSL4GIE_F32 0
#define SSL4GIE_BF16 1
#define SSL4GIE_EARG 1000
#define SSL4GIE_EPEER 1001 /* direct all-reduce: a peer did not arrive in time (sticky, see _error) */

int ssl4gie_abi_version(void);

/* ---------------------------------------------------------------- LayerNorm (eps=1e-6)
 * replaces nn.LayerNorm at models_mae.py:227 / models.py:384,498 inside timm Block
 * (norm1/norm2) and the final `norm` (models_mae.py:168).  x fp32 [rows, cols]; y in
 * y_dtype; mean/rstd fp32 [rows] saved for backward (may be NULL). */
int ssl4gie_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y,
                          int y_dtype, float* mean, float* rstd, int rows, int cols, float eps,
                          void* stream);
size_t ssl4gie_layernorm_bwd_workspace_bytes(int rows, int cols);
/* dx = (dres ? dres : 0) + LN'(dy); dx_lp (optional) is the same value in lp_dtype (feeds the
 * next backward GEMM); dgamma/dbeta overwritten or accumulated. */
int ssl4gie_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* gamma,
                          const float* mean, const float* rstd, const float* dres, float* dx,
                          void* dx_lp, int lp_dtype, float* dgamma, float* dbeta, int accumulate,
                          float* workspace, int rows, int cols, void* stream);

/* column sums of x[rows, cols] (row stride ld) -> out[cols]: bias gradients of nn.Linear */
size_t ssl4gie_colsum_workspace_bytes(int rows, int cols);
int ssl4gie_colsum(const void* x, int dtype, float* out, int accumulate, float* workspace,
                   int rows, int cols, long long ld, void* stream);

/* ---------------------------------------------------------------- GEMM with fused epilogues
 * replaces nn.Linear fwd/bwd (timm Attention.qkv/proj, Mlp.fc1/fc2; decoder_embed/decoder_pred
 * models_mae.py:47,59), the patch-embed conv lowered to a GEMM (models_mae.py:33), and the
 * batched QK^T / PV products of the fp32 parity attention.
 *   C[b](m,n) = epilogue( alpha * sum_k A[b](m,k) * B[b](k,n) )
 * A(m,k) at A + b1*sAb1 + b2*sAb2 + m*sAm + k*sAk (element strides), likewise B(k,n);
 * C row-major with row stride ldc.  Fast MFMA-bf16 paths: "NT" (sAk==1 && sBk==1, K%64==0)
 * and "TN" (sAm==1 && sBn==1, split-K through the workspace); anything else takes the generic
 * f32-MFMA kernel. */
enum {
    SSL4GIE_EPI_NONE = 0,          /* C = alpha*acc                       */
    SSL4GIE_EPI_BIAS = 1,          /* C = acc + bias[n]                   */
    SSL4GIE_EPI_BIAS_GELU = 2,     /* C = u = acc + bias[n]; out2 = gelu(u) (exact erf) */
    SSL4GIE_EPI_BIAS_RESIDUAL = 3, /* C = acc + bias[n] + residual[m,n] (fp32 residual) */
    SSL4GIE_EPI_DGELU = 4,         /* C = acc * gelu'(aux[m,n])           */
    /* the pair the block executor uses: the forward stores gelu'(u) instead of u, so that the
     * backward epilogue is one multiply (no transcendental work on the data-gradient GEMM) */
    SSL4GIE_EPI_BIAS_GELU_GRAD = 5, /* u = acc + bias[n]; C = gelu'(u); out2 = gelu(u) */
    SSL4GIE_EPI_MUL_AUX = 6,        /* C = acc * aux[m,n]                  */
    SSL4GIE_EPI_RELU_MASK_AUX = 7,  /* C = aux[m,n] > 0 ? acc : 0: gradient through the ReLU in front of a
                                       convolution (aux = the convolution's input); implicit-conv NT
                                       products with bf16 output only */
    SSL4GIE_EPI_ADD_AUX = 8,        /* C = acc + aux[m,n]: a second gradient contribution of the same
                                       tensor (residual branch) joined in the data-gradient GEMM;
                                       256x256 NT kernel, bf16 output only */
    SSL4GIE_EPI_AFFINE_AUX_RELU = 9 /* C = act(acc * scale[n] + bias[n] (+ aux[m,n])), act = ReLU if `relu`:
                                       the training-mode BatchNorm after a 1x1 convolution (scale = rstd gamma,
                                       bias = beta - mean scale, from a statistics-only first product: colstats
                                       with C == NULL), the bottleneck's residual add and ReLU, applied to the
                                       fp32 accumulators — the raw convolution output is never written or re-read
                                       (torchvision Bottleneck conv3 / bn3 / downsample under torch.no_grad();
                                       256x256 NT kernel, bf16 output only; aux may be NULL) */
};
/* Implicit patch-matrix operand of a 3x3 / pad-1 convolution over a channels-last bf16 map
 * x [B, H, W, C] (ssl4gie_gemm_desc::conv).  The patch matrix
 *     P[(b, oy, ox), (dy*3 + dx)*C + c] = act(x[b, oy*s + dy - 1, ox*s + dx - 1, c])   (0 outside)
 * of ssl4gie_im2col3x3 is never materialised: the 256x256 kernels gather its K-tiles straight
 * from the map with per-lane LDS-DMA addresses (out-of-image taps read a zero page), act = ReLU
 * applied to the MFMA fragments when `relu`.  Replaces the cuDNN implicit-GEMM convolutions
 * behind nn.Conv2d(k=3, p=1) in Models/DPT_decoder.py:212-233,397-447,469-478 and torchvision
 * Bottleneck.conv2.
 *   NT (forward / stride-1 data gradient):  A := P, d->A = x, M = B*Ho*Wo, K = 9*C, C % 64 == 0;
 *       sAm / sAk are ignored.
 *   TN (weight gradient dW = dY^T P):       B := P, d->B = x, K = B*Ho*Wo (% 64 == 0), N = 9*C,
 *       C % 8 == 0; sBk / sBn are ignored.
 * Limits: bf16 operands, stride in {1, 2}, map smaller than 2 GiB, Wo >= 2.  A descriptor that
 * carries `conv` and does not meet them is rejected with SSL4GIE_EARG (no fallback inside). */
typedef struct ssl4gie_conv3x3_geom {
    int B, H, W, C;
    int stride;
    int relu;
} ssl4gie_conv3x3_geom;

typedef struct ssl4gie_gemm_desc {
    int M, N, K;
    int batch1, batch2; /* >=1 */
    const void* A;
    long long sAm, sAk, sAb1, sAb2;
    const void* B;
    long long sBk, sBn, sBb1, sBb2;
    void* C;
    long long ldc, sCb1, sCb2;
    int dtype_ab; /* SSL4GIE_F32 | SSL4GIE_BF16 */
    int dtype_c;  /* type of C, out2 and aux */
    float alpha;
    int epilogue;
    const float* bias;     /* [N] */
    const float* residual; /* fp32 [M, N], row stride ldr */
    long long ldr;
    const void* aux; /* [M, N] row stride ldc, dtype_c */
    void* out2;      /* [M, N] row stride ldc, dtype_c */
    int accumulate;  /* C += (EPI_NONE only) */
    /* optional [M] fp32: colsum_a[m] (+)= sum_k A(m,k) with the same `accumulate` flag (alpha not
     * applied).  The bias gradient of nn.Linear riding on its weight-gradient product
     * dW = dY^T X (A = dY^T): fused into the 256x256 TN kernel (one extra MFMA against a ones
     * fragment per A fragment), a separate column-sum pass on the other paths.  Needs sAm == 1. */
    float* colsum_a;
    /* optional: the K-major operand (A of an NT product, B of a TN product) is the implicit 3x3
     * patch matrix of this map instead of a matrix in memory (see ssl4gie_conv3x3_geom) */
    const ssl4gie_conv3x3_geom* conv;
    /* optional fp32 [ceil(M / 128)][2][N]: per 128-row block, the column sums ([0]) and sums of
     * squares ([1]) of the STORED outputs — the batch statistics of the BatchNorm that follows a
     * convolution, produced by the GEMM's epilogue (see ssl4gie_bn_fwd_partials).  NT products with
     * bf16 C, EPI_NONE, no accumulate, N % 8 == 0 only; anything else is SSL4GIE_EARG.  With C == NULL the
     * product only produces these statistics (of the values it WOULD store, bf16-rounded): nothing is written
     * to C — the first half of the BatchNorm-fused 1x1 convolution (SSL4GIE_EPI_AFFINE_AUX_RELU). */
    float* colstats;
    const float* scale; /* [N], SSL4GIE_EPI_AFFINE_AUX_RELU only */
    int relu;           /* SSL4GIE_EPI_AFFINE_AUX_RELU only */
} ssl4gie_gemm_desc;
size_t ssl4gie_gemm_workspace_bytes(const ssl4gie_gemm_desc* d);
int ssl4gie_gemm(const ssl4gie_gemm_desc* d, void* workspace, size_t workspace_bytes,
                 void* stream);
/* Two weight-gradient (TN) products with the same contraction length K in one launch: they share
 * the one-workgroup-per-CU grid, so each needs half the split-K slabs (and half the reduction
 * traffic) it would need alone.  Descriptors as for ssl4gie_gemm; pairs that do not qualify run as
 * two ssl4gie_gemm calls.  Used for (dW_fc2, dW_fc1) and (dW_proj, dW_qkv) of a transformer block. */
size_t ssl4gie_gemm_tn_pair_workspace_bytes(const ssl4gie_gemm_desc* a, const ssl4gie_gemm_desc* b);
int ssl4gie_gemm_tn_pair(const ssl4gie_gemm_desc* a, const ssl4gie_gemm_desc* b, void* workspace,
                         size_t workspace_bytes, void* stream);
/* n (1..16) weight-gradient (TN) products with the same contraction length K in ONE launch.  With
 * enough output tiles in the group (>= 70 % of the CUs) every workgroup runs a whole-K tile and
 * writes C directly: no split-K slabs, no slab reduction — the 8 + 8 products of two encoder blocks
 * (216 tiles) or four decoder blocks (192 tiles) of the MAE step.  Smaller groups split K like the
 * pair.  `descs` is an array of n descriptors (alpha / accumulate / colsum_a per product); products
 * that do not qualify run as n ssl4gie_gemm calls.  Replaces the per-layer autograd weight-gradient
 * GEMMs of nn.Linear (Models/mae/models_mae.py:39-41,53-55 via timm Block). */
size_t ssl4gie_gemm_tn_group_workspace_bytes(const ssl4gie_gemm_desc* descs, int n);
int ssl4gie_gemm_tn_group(const ssl4gie_gemm_desc* descs, int n, void* workspace,
                          size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- attention
 * replaces timm Attention.forward == Models/models.py:195-209 minus windowing:
 * softmax(q k^T * hd^-1/2) v on the packed qkv activation [B, N, 3, H, hd] (row = token).
 * out [B, N, H*hd]; lse fp32 [B, H, N] (saved for backward).  bf16 path is one fused kernel
 * (whole K/V of a head staged in LDS, softmax in registers, no N x N materialisation); the
 * f32 parity path materialises scores in the caller-provided workspace. */
size_t ssl4gie_attn_workspace_bytes(int dtype, int B, int N, int H, int hd);
int ssl4gie_attn_fwd(const void* qkv, void* out, float* lse, int dtype, int B, int N, int H,
                     int hd, void* workspace, void* stream);
int ssl4gie_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                     void* dqkv, int dtype, int B, int N, int H, int hd, void* workspace,
                     void* stream);

/* ---------------------------------------------------------------- casts
 * fp32 master weights -> MFMA operand copies (bf16), plain and transposed ([R,C] -> [C,R]). */
int ssl4gie_cast(const float* src, void* dst, int dst_dtype, long long n, void* stream);
int ssl4gie_cast_transpose(const float* src, void* dst, int dst_dtype, int rows, int cols,
                           void* stream);
/* bf16 transposed copies of S matrices of one fp32 arena in ONE launch (the per-step refresh of the
 * pre-transposed weight operands).  Device tables: mat_off [S] int64 element offsets (the same in src
 * and dst), mat_rows / mat_cols [S], tile_start [S+1] = running count of 64 x 64 tiles per matrix
 * (ceil(rows/64) * ceil(cols/64)), total_tiles = tile_start[S] (passed by value: no device read).
 * dst[off + c * rows + r] = bf16(src[off + r * cols + c]). */
int ssl4gie_cast_transpose_batch(const float* src, void* dst, const long long* mat_off,
                                 const int* mat_rows, const int* mat_cols, const int* tile_start,
                                 int S, int total_tiles, void* stream);
/* out = a + b (b may be NULL), optionally also written as an operand-type copy out_lp:
 * plumbing of the fp32 residual-gradient stream (tap gradients, models.py:450-454). n % 4 == 0 */
int ssl4gie_add_cast(const float* a, const float* b, float* out, void* out_lp, int lp_dtype,
                     long long n, void* stream);

/* ---------------------------------------------------------------- MAE glue
 * random_masking (models_mae.py:123-148): stable argsort of fp32 noise [B, L] ->
 * ids_shuffle / ids_restore (int64, bit-exact) and mask (fp32; 1 = removed). */
int ssl4gie_mask_argsort(const float* noise, long long* ids_shuffle, long long* ids_restore,
                         float* mask, int B, int L, int len_keep, void* stream);
/* im2col of non-overlapping p x p patches (PatchEmbed conv k=s=p as a GEMM, models_mae.py:152):
 * out[b*nsel + j, c*p*p + py*p + px] = img[b, c, gy*p+py, gx*p+px], patch = ids ? ids[b, j] : j.
 * order==1 emits the 'nhwpqc' layout of patchify (models_mae.py:95-107) instead. */
int ssl4gie_patch_gather(const float* img, const long long* ids, void* out, int out_dtype,
                         int B, int C, int H, int W, int p, int nsel, long long ids_stride,
                         int order, void* stream);
/* encoder input assembly (models_mae.py:155-163 / models.py:445-448):
 * x[b,0,:] = cls + pos[0]; x[b,1+j,:] = y[b*nsel+j,:] + pos[1 + (ids ? ids[b,j] : j)] */
int ssl4gie_tokens_assemble(const void* y, int y_dtype, const float* cls, const float* pos,
                            const long long* ids, long long ids_stride, float* x, int B,
                            int nsel, int D, void* stream);
/* backward of the above: dy[b*nsel+j] = dx[b,1+j]; dcls = sum_b dx[b,0] */
int ssl4gie_tokens_assemble_bwd(const float* dx, void* dy, int dy_dtype, float* dcls,
                                int accumulate, int B, int nsel, int D, void* stream);
/* decoder input assembly (models_mae.py:177-183): xd[b,0] = y[b,0] + dpos[0];
 * xd[b,1+i] = (r = ids_restore[b,i]) < nkeep ? y[b,1+r] : mask_token, + dpos[1+i] */
int ssl4gie_decoder_assemble(const void* y, int y_dtype, const float* mask_token,
                             const float* dpos, const long long* ids_restore, float* xd, int B,
                             int L, int nkeep, int D, void* stream);
size_t ssl4gie_decoder_assemble_bwd_workspace_bytes(int B, int L, int D);
/* dy[b,0] = dxd[b,0]; dy[b,1+j] = dxd[b, 1+ids_shuffle[b,j]] (j<nkeep);
 * dmask_token = sum over removed positions of dxd */
int ssl4gie_decoder_assemble_bwd(const float* dxd, const long long* ids_shuffle, void* dy,
                                 int dy_dtype, float* dmask_token, int accumulate,
                                 float* workspace, int B, int L, int nkeep, int D, void* stream);
/* forward_loss (models_mae.py:198-214) and its gradient.  pred fp32 [B, has_cls+L, P] (with
 * has_cls=1 row 0 of each sample is the cls prediction and is ignored: decoder_pred runs on all
 * 1+L tokens, models_mae.py:191-194), img fp32 NCHW, mask [B, L] (1 = removed).
 *   per_patch (optional) [B, L] = mask * mean_k (pred - target)^2   (host sums / mask.sum())
 *   dpred (optional, same shape as pred) = gscale_host * gpp[b,l] * mask * 2 (pred-target) / P, cls rows
 *   zeroed; gpp [B, L] is the upstream gradient of per_patch (NULL = 1). */
int ssl4gie_mae_loss(const float* pred, const float* img, const float* mask, float* per_patch,
                     float* dpred, const float* gpp, float gscale_host, int norm_pix,
                     int has_cls, int B, int C, int H, int W, int p, void* stream);

/* ---------------------------------------------------------------- transformer-block executor
 * One timm Block: x += proj(attn(norm1(x))); x += fc2(gelu(fc1(norm2(x)))).
 * Residual stream fp32; MFMA operands in `dtype`.  `w*` are operand-type copies of the weights
 * ([out,in] row-major) and `w*_t` their transposes ([in,out]) used by the data-gradient GEMMs.
 * Saved activations live in caller-owned buffers (`ssl4gie_block_act`). */
typedef struct ssl4gie_block_weights {
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    const float *bqkv, *bproj, *bfc1, *bfc2;
    const void *wqkv, *wproj, *wfc1, *wfc2;         /* [3D,D] [D,D] [F,D] [D,F] */
    const void *wqkv_t, *wproj_t, *wfc1_t, *wfc2_t; /* transposes (backward only) */
} ssl4gie_block_weights;
typedef struct ssl4gie_block_grads { /* fp32, same shapes as the fp32 master parameters */
    float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    float *bqkv, *bproj, *bfc1, *bfc2;
    float *wqkv, *wproj, *wfc1, *wfc2;
} ssl4gie_block_grads;
typedef struct ssl4gie_block_act { /* per-block saved activations; T = tokens = B*N */
    float *mean1, *rstd1, *mean2, *rstd2; /* [T] */
    void* h1;                             /* [T, D]  norm1 out     */
    void* qkv;                            /* [T, 3D]               */
    void* attn;                           /* [T, D]  attention out */
    float* lse;                           /* [B, H, N]             */
    float* xmid;                          /* [T, D]  fp32          */
    void* h2;                             /* [T, D]  norm2 out     */
    void* u;                              /* [T, F]  gelu'(fc1 pre-activation) */
    void* g;                              /* [T, F]  gelu(u)       */
} ssl4gie_block_act;
typedef struct ssl4gie_block_dims {
    int B, N, D, H, F; /* F = mlp hidden */
    int dtype;
    float eps;
} ssl4gie_block_dims;
size_t ssl4gie_block_workspace_bytes(const ssl4gie_block_dims* d);
/* x_in fp32 [T, D] -> x_out fp32 [T, D] (may alias nothing) */
int ssl4gie_block_fwd(const ssl4gie_block_dims* d, const ssl4gie_block_weights* w,
                      const ssl4gie_block_act* a, const float* x_in, float* x_out,
                      void* workspace, void* stream);
/* dx_out fp32 (+ dx_out_lp, its operand-type copy; may be NULL in f32 mode) -> dx_in fp32 and
 * dx_in_lp; parameter grads overwritten or accumulated.  `accumulate` is a flag word:
 * SSL4GIE_BWD_ACCUMULATE (1) and SSL4GIE_BWD_DEFER_WGRAD (2): with the latter the four weight-gradient
 * products (and the bias gradients riding on them) are NOT launched; `workspace` then holds their
 * dY operands and must stay untouched until the caller has launched them (next two entries). */
#define SSL4GIE_BWD_ACCUMULATE 1
#define SSL4GIE_BWD_DEFER_WGRAD 2
/* SSL4GIE_BWD_NO_JOIN | SSL4GIE_BWD_SLOT(s), s in 0..3: `stream` does NOT wait for the block's weight-gradient
 * products at the end of the call (the default join makes the next block's first data-gradient GEMM wait for
 * this block's last weight gradient: 20 stalls per MAE step).  Instead their completion is recorded under slot s
 * (as ssl4gie_wgrad_group does): `workspace`, dx_out_lp and the gradient targets must stay untouched until
 * ssl4gie_wgrad_wait(s, stream) has been called — the caller alternates two workspaces and slots. */
#define SSL4GIE_BWD_NO_JOIN 4
#define SSL4GIE_BWD_SLOT(s) (((s) & 3) << 4)
int ssl4gie_block_bwd(const ssl4gie_block_dims* d, const ssl4gie_block_weights* w,
                      const ssl4gie_block_act* a, const ssl4gie_block_grads* g,
                      const float* x_in, const float* dx_out, const void* dx_out_lp,
                      float* dx_in, void* dx_in_lp, int accumulate, void* workspace,
                      void* stream);
/* The four weight-gradient products of ssl4gie_block_bwd are enqueued on a library-owned
 * non-blocking side stream (one per device, created on first use) and ordered against `stream`
 * with events only: each waits for its dY producer, and `stream` waits for the last of them
 * before any later work — to the caller the call still behaves as if everything ran on `stream`.
 * on = 0 folds them back onto `stream` (also: environment SSL4GIE_WGRAD_STREAM=0); bench.py does
 * that while it measures per-kernel durations. */
int ssl4gie_set_wgrad_stream(int on);
/* Deferred weight gradients: out4[0..3] = the dW_fc2, dW_fc1, dW_proj, dW_qkv product descriptors
 * of a block whose ssl4gie_block_bwd ran with SSL4GIE_BWD_DEFER_WGRAD on `workspace` (dy = the
 * dx_out_lp it was given; dx_out in f32 mode).  Collect the descriptors of several blocks and run
 * them as one ssl4gie_gemm_tn_group: */
int ssl4gie_block_wgrad_descs(const ssl4gie_block_dims* d, const ssl4gie_block_act* a,
                              const ssl4gie_block_grads* g, const void* dy, void* workspace,
                              int accumulate, ssl4gie_gemm_desc* out4);
/* ssl4gie_gemm_tn_group on the weight-gradient side stream: the side stream first waits for all
 * work enqueued on `stream` so far, and event `slot` (0..3) then marks the group's completion.
 * Nothing waits for the group until ssl4gie_wgrad_wait(slot, s) makes stream `s` do so — the caller
 * must issue that wait before anything reads the gradients or rewrites the operands (block
 * workspaces, activations).  Without a side stream the group simply runs on `stream`. */
int ssl4gie_wgrad_group(const ssl4gie_gemm_desc* descs, int n, void* workspace, size_t workspace_bytes,
                        int slot, void* stream);
int ssl4gie_wgrad_wait(int slot, void* stream);
/* Number of CUs the persistent / one-workgroup-per-CU GEMM grids are sized for (8..256, default 240:
 * the weight-gradient side stream shares the chip; environment SSL4GIE_COMPUTE_CUS).  The 256x256 kernels hold all 160 KiB of a CU's LDS, so an RCCL
 * kernel running beside them needs CUs of its own: ssl4gie_amd.parallel reserves a few in
 * data-parallel runs (and caps RCCL's channel count to match) so that no GEMM workgroup has to
 * wait for a second wave behind a busy CU. */
int ssl4gie_set_compute_cus(int n);

/* ---------------------------------------------------------------- DPT decoder glue (channels-last)
 * Replaces the torch ops around the convolutions of Models/DPT_decoder.py (depth variant): the
 * maps are [B, H, W, C] in the operand type (`dtype`), C % 8 == 0 (bf16) / % 4 == 0 (fp32).  With
 * this layout Conv2d(k=1) and ConvTranspose2d(k=s) are token-major ssl4gie_gemm calls and
 * Conv2d(k=3, pad=1) is ssl4gie_gemm over the patch matrix below.
 *
 * im2col3x3: cols[(b,oy,ox), (dy*3+dx)*C + c] = act(x[b, oy*s+dy-1, ox*s+dx-1, c]) (0 outside),
 *   act = ReLU if `relu` (ResidualConvUnit_custom.forward :225-229 activates the conv INPUT);
 *   row stride ld >= 9C, columns [9C, ld) zero-filled; s in {1, 2} (act_postprocess42.1 :397-403).
 * col2im3x3: gather-form transpose (data gradient of the strided conv). */
/* Operand images of an nn.Conv2d(k = 3) weight [Cout][Cin][3][3] (fp32 parameter) in one launch, cast included:
 * mode 0 out[co][tap Cin + ci] (row stride ld >= 9 Cin: forward / weight-gradient layout), mode 1
 * out[ci][(8 - tap) Cout + co] (ld >= 9 Cout: the flipped kernel of the stride-1 data gradient), mode 2
 * out[tap Cin + ci][co] (ld >= 9 Cin rows: transpose of mode 0); padding is written as zeros.  What cuDNN's
 * filter transforms do behind the reference's nn.Conv2d calls (Models/DPT_decoder.py:212-233,397-447,469-478;
 * torchvision Bottleneck.conv2), once per optimizer step. */
int ssl4gie_conv3x3_weight_pack(const float* w, void* out, int dtype, int Cout, int Cin, int mode, int ld,
                                void* stream);
/* The same for n weights in ONE launch (arrays of n pointers / geometries; all outputs of type `dtype`): every 3x3
 * convolution of a model re-packs its operand images after each optimizer step. */
int ssl4gie_conv3x3_weight_pack_batch(const void* const* w, void* const* out, const int* Cout, const int* Cin,
                                      const int* mode, const int* ld, int n, int dtype, void* stream);
/* ... and back for the weight gradient: dw2 [Cout][ld] fp32 with columns (tap, ci) -> (+)= dW [Cout][Cin][3][3] */
int ssl4gie_conv3x3_wgrad_unpack(const float* dw2, float* dw, int Cout, int Cin, int ld, int accumulate,
                                 void* stream);
int ssl4gie_im2col3x3(const void* x, void* cols, int dtype, int B, int H, int W, int C, int stride,
                      int relu, long long ld, void* stream);
int ssl4gie_col2im3x3(const void* dcols, void* dx, int dtype, int B, int H, int W, int C,
                      int stride, long long ld, void* stream);
/* Direct 3x3 convolution, stride 1, pad 1, bf16, for NARROW channel counts — the DPT output head's
 * nn.Conv2d(128, 32, 3, 1, 1) (output_conv.2, DPT_decoder.py:473-478) and its gradients, where the
 * gathered 256x256 GEMM tiles would be 7/8 padding.  x [B,H,W,Cin], w2 [Cout, 9*Cin] (the layout
 * of the implicit-GEMM path: taps row-major, channels innermost), y [B,H,W,Cout]:
 *     y = conv(relu_in ? relu(x) : x, w2) (+ bias[Cout] fp32)  (then y = relu_mask > 0 ? y : 0, with
 *     relu_mask [B,H,W,Cout] bf16 — the data gradient of a convolution behind a ReLU)
 * colstats (optional, not together with relu_mask): fp32 [ssl4gie_conv3x3_direct_tiles(B,H,W)][2][Cout],
 * per 8 x 32 pixel tile the column sums ([0]) and sums of squares ([1]) of the stored y — the
 * partial statistics ssl4gie_bn_fwd_partials / ssl4gie_bn_stats_partials take (as
 * ssl4gie_gemm_desc.colstats, with one partial per tile instead of per 128 rows).
 * The data gradient is the same call on dy with w2 := weight.flip(2,3) as [Cin, 9*Cout].
 * _ok(): Cin % 32 == 0 and Cout % 8 == 0 (any H, W); otherwise the calls return SSL4GIE_EARG.
 * Meant for Cout <= 128 or Cin == 32 (at 256 -> 256 it merely ties the gathered GEMM). */
int ssl4gie_conv3x3_direct_ok(int B, int H, int W, int Cin, int Cout);
int ssl4gie_conv3x3_direct_tiles(int B, int H, int W);
int ssl4gie_conv3x3_direct_fwd(const void* x, const void* w2, const float* bias,
                               const void* relu_mask, void* y, float* colstats, int B, int H, int W,
                               int Cin, int Cout, int relu_in, void* stream);
/* dW2 [Cout, 9*Cin] fp32 (+)= sum over pixels of dy [B,H,W,Cout] x patch(relu_in ? relu(x) : x);
 * Cout % 32 == 0 (<= 256), Cin % 64 == 0 (<= 512); dbias [Cout] fp32 (+)= sum over pixels of dy, or NULL (it rides on a spare
 * accumulator of the same kernel).  Persistent workgroups write one fp32 partial each into the
 * workspace, a second kernel sums them in a fixed order (deterministic, no atomics). */
int ssl4gie_conv3x3_direct_wgrad_ok(int B, int H, int W, int Cin, int Cout);
size_t ssl4gie_conv3x3_direct_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int ssl4gie_conv3x3_direct_wgrad(const void* dy, const void* x, float* dw2, float* dbias,
                                 void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin,
                                 int Cout, int relu_in, int accumulate, void* stream);
/* The same two kernels with a training-mode BatchNorm (+ ReLU) applied to their input on the way in: the operand
 * is act(x in_coef[0][ci] + in_coef[1][ci]) rounded to bf16 (in_coef [2][Cin] as ssl4gie_bn_coef_partials writes
 * it; act = ReLU if relu_in), zero-padded AFTER the normalisation — torchvision Bottleneck bn1 -> relu -> conv2
 * without a BatchNorm pass or a normalised map in memory (its backward: ssl4gie_bn_bwd_xmask on the data gradient).
 * Values equal ssl4gie_bn_fwd_partials followed by the plain kernels bit for bit. */
int ssl4gie_conv3x3_direct_fwd_affine(const void* x, const float* in_coef, const void* w2, const float* bias,
                                      void* y, float* colstats, int B, int H, int W, int Cin, int Cout,
                                      int relu_in, void* stream);
int ssl4gie_conv3x3_direct_wgrad_affine(const void* dy, const void* x, const float* in_coef, float* dw2,
                                        float* dbias, void* workspace, size_t workspace_bytes, int B, int H,
                                        int W, int Cin, int Cout, int relu_in, int accumulate, void* stream);
/* torchvision ResNet.conv1 = nn.Conv2d(3, 64, 7, stride 2, pad 3, bias=False) (reference
 * Models/models.py:63-69) WITHOUT a patch matrix (ssl4gie_stem_im2col7x7 + GEMM remain for fp32).
 *   pack:  img fp32 [B,3,H,W] -> packed bf16 [B, 2 Ho + 6, 2 Wo + 6, 4] (three channels + a zero,
 *          three pixels / rows of zero padding in front), Ho = (H-1)/2 + 1, Wo likewise; once per
 *          batch, shared by every forward on it and by the weight gradient.
 *   fwd:   y [B,Ho,Wo,64] bf16 = conv(packed, w2s), w2s bf16 [64][8][8][4] = weight[co][c][ky][kx]
 *          at [co][ky][kx][c], zero where ky = 7, kx = 7 or c = 3.  colstats (optional): fp32
 *          [ssl4gie_stem7x7_tiles(B,H,W)][2][64] per-tile BatchNorm partial statistics of y.
 *   wgrad: dw2s fp32 [64][7][8][4] (+)= sum over output pixels of dy [B,Ho,Wo,64] x patch; the
 *          kx = 7 and c = 3 entries are arithmetic by-products, not gradients (their weights do
 *          not exist).  Deterministic: per-workgroup partials + fixed-order reduction. */
size_t ssl4gie_stem7x7_packed_bytes(int B, int H, int W);
int ssl4gie_stem7x7_pack(const float* img, void* packed, int B, int H, int W, void* stream);
int ssl4gie_stem7x7_tiles(int B, int H, int W);
int ssl4gie_stem7x7_fwd(const void* packed, const void* w2s, void* y, float* colstats, int B, int H,
                        int W, void* stream);
size_t ssl4gie_stem7x7_wgrad_workspace_bytes(int B, int H, int W);
int ssl4gie_stem7x7_wgrad(const void* dy, const void* packed, float* dw2s, void* workspace,
                          size_t workspace_bytes, int B, int H, int W, int accumulate, void* stream);
/* F.interpolate(scale_factor=2, mode="bilinear", align_corners=True) (:293-295, Interpolate :69-104)
 * x [B,H,W,C] -> y [B,2H,2W,C]; backward in gather form (no atomics) */
int ssl4gie_bilinear2x_fwd(const void* x, void* y, int dtype, int B, int H, int W, int C,
                           void* stream);
int ssl4gie_bilinear2x_bwd(const void* dy, void* dx, int dtype, int B, int H, int W, int C,
                           void* stream);
/* ConvTranspose2d(k = s) (:345-354, :371-380) = GEMM + scatter: g [B*H*W, k*k*C] with columns
 * ordered (i, j, c) -> y[b, k*y+i, k*x+j, c] = g + bias[c]; unshuffle is its inverse (gradient) */
int ssl4gie_pixel_shuffle(const void* g, const float* bias, void* y, int dtype, int B, int H, int W,
                          int k, int C, void* stream);
int ssl4gie_pixel_unshuffle(const void* dy, void* dg, int dtype, int B, int H, int W, int k, int C,
                            void* stream);
/* Slice(1) + Transpose + Unflatten (:5-11, :449-459): fp32 tap z [B, 1+L, D] -> x [B*L, D] in
 * `dtype` (cls row dropped); gradient: dz[:,0] = 0, dz[:,1:] = dx */
int ssl4gie_tokens_to_map(const float* z, void* x, int dtype, int B, int L, int D, void* stream);
int ssl4gie_map_to_tokens(const void* dx, float* dz, int dtype, int B, int L, int D, void* stream);
/* op 0: out = a + b (skip_add :233, :290);  op 1: out = (a > 0 ? b : 0) + (c ? c : 0): gradient
 * through the ReLU that precedes a conv (a = the activation's input) plus the skip gradient */
int ssl4gie_eltwise(int op, const void* a, const void* b, const void* c, void* out, int dtype,
                    long long n, void* stream);
/* depth head tail ReLU -> Conv2d(32, 1, 1) -> Sigmoid (:479-481): y[m] = sigmoid(sum_c relu(x[m,c])
 * w[c] + bias[0]), y fp32 [M]; backward gives dx and dw [C], db [1] (two-stage reduction) */
int ssl4gie_depth_head_fwd(const void* x, const float* w, const float* bias, float* y, int dtype,
                           long long M, int C, void* stream);
size_t ssl4gie_depth_head_bwd_workspace_bytes(long long M, int C);
int ssl4gie_depth_head_bwd(const void* x, const float* w, const float* y, const float* dy, void* dx,
                           float* dw, float* db, int accumulate, float* workspace, int dtype,
                           long long M, int C, void* stream);

/* ---------------------------------------------------------------- ResNet50 glue (channels-last)
 * Replaces the torch ops around the convolutions of torchvision ResNet(Bottleneck,[3,4,6,3]) as the
 * reference builds it (Models/models.py:63-152; MoCo: Models/moco_v3/main_moco.py:185-187).  1x1
 * convolutions are token-major ssl4gie_gemm calls, 3x3 ones go through ssl4gie_im2col3x3.
 * stem_im2col7x7: fp32 NCHW image -> conv1 (7x7, s2, p3) patch matrix [B*Ho*Wo, ld], K = 147
 *   ordered (dy, dx, c), columns [147, ld) zero.
 * subsample2: rows of a stride-2 1x1 convolution (downsample.0); backward = 1 scatters zeros. */
int ssl4gie_stem_im2col7x7(const float* img, void* cols, int dtype, int B, int H, int W,
                           long long ld, void* stream);
int ssl4gie_subsample2(const void* x, void* y, int dtype, int B, int H, int W, int C, int backward,
                       void* stream);
/* BatchNorm2d / BatchNorm1d in training mode over the rows of x [rows, C] (C % 8 == 0): batch
 * statistics in fp32 (biased variance for the normalisation, unbiased for running_var, momentum as
 * nn.BatchNorm: running = (1-m) running + m batch), y = act(xhat gamma + beta (+ res)) with
 * act = ReLU if `relu`; gamma / beta / res / running_* may be NULL.  mean / rstd [C] are outputs
 * kept for backward; `workspace` (ssl4gie_bn_workspace_bytes) is always required (with training == 0 they are INPUTS: the caller's running statistics, no
 * reduction runs).  Backward: g = relu ? (y > 0 ? dy : 0) : dy; dgamma = sum g xhat, dbeta = sum g,
 * dx = gamma rstd (g - mean(g) - xhat mean(g xhat)); dres (optional) = g. */
size_t ssl4gie_bn_workspace_bytes(long long rows, int C);
int ssl4gie_bn_fwd(const void* x, const float* gamma, const float* beta, const void* res, void* y,
                   float* mean, float* rstd, float* running_mean, float* running_var,
                   float momentum, float eps, int relu, int training, float* workspace, int dtype,
                   long long rows, int C, void* stream);
int ssl4gie_bn_bwd(const void* dy, const void* y, const void* x, const float* gamma,
                   const float* mean, const float* rstd, void* dx, void* dres, float* dgamma,
                   float* dbeta, int accumulate, int relu, float* workspace, int dtype,
                   long long rows, int C, void* stream);
/* The same backward for BatchNorm + ReLU WITHOUT a residual input (bn1 / bn2 of a torchvision Bottleneck, the
 * stem's bn1): the ReLU mask is rebuilt as x a + b > 0 from the forward's own coefficients (a = rstd gamma,
 * b = beta - mean a: gamma / beta must be the forward's) instead of read from the ReLU output — the two passes
 * stream 5 tensors instead of 7.  The results equal ssl4gie_bn_bwd(relu = 1, dres = NULL) exactly unless the
 * forward rounded a positive pre-activation below the operand type's smallest subnormal to zero. */
int ssl4gie_bn_bwd_xmask(const void* dy, const void* x, const float* gamma, const float* beta,
                         const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta,
                         int accumulate, float* workspace, int dtype, long long rows, int C, void* stream);
/* The ReLU mask of a BatchNorm (+ residual) + ReLU as a bit map instead of the ReLU output (bf16 maps): the forward
 * writes bit j of relu_bits[i] = (y[8 i + j] > 0) beside y (rows * C / 8 bytes), the backward's reduction pass reads
 * that byte stream instead of y — 1/16 of the bytes — and writes the masked gradient `dres` (required), which its
 * apply pass reads.  bn3 of a torchvision Bottleneck (its mask depends on the residual input too, so it cannot be
 * rebuilt from the BatchNorm input as ssl4gie_bn_bwd_xmask does).  Results equal ssl4gie_bn_bwd exactly. */
int ssl4gie_bn_fwd_partials_bits(const void* x, const float* partial, int parts, const float* gamma,
                                 const float* beta, const void* res, void* y, unsigned char* relu_bits,
                                 float* mean, float* rstd, float* running_mean, float* running_var,
                                 float momentum, float eps, float* workspace, int dtype, long long rows, int C,
                                 void* stream);
int ssl4gie_bn_bwd_bits(const void* dy, const unsigned char* relu_bits, const void* x, const float* gamma,
                        const float* mean, const float* rstd, void* dx, void* dres, float* dgamma, float* dbeta,
                        int accumulate, float* workspace, int dtype, long long rows, int C, void* stream);
/* SyncBatchNorm (convert_sync_batchnorm: Depth_estimation/train_depth.py:225,
 * Models/moco_v3/main_moco.py:196) = the same kernels with the exchange step between them:
 *   forward : ssl4gie_bn_stats (LOCAL mean / biased var) -> caller combines over ranks ->
 *             ssl4gie_bn_fwd(training = 0) with the global mean / rstd;
 *   backward: ssl4gie_bn_bwd_reduce (LOCAL sum g, sum g xhat -> sums [2, C]; also dres = g) ->
 *             caller all-reduces sums -> ssl4gie_bn_bwd_apply with 1 / global row count. */
int ssl4gie_bn_stats(const void* x, float* mean, float* var, float* workspace, int dtype,
                     long long rows, int C, void* stream);
/* The exchange's arithmetic as ONE launch: gathered [world][2C + 1] = every rank's (mean[C], biased var[C],
 * row count) -> pooled mean / rstd (ranks may hold different row counts), the total row count (device
 * scalar) and, if given, the running statistics (unbiased variance, momentum) — what torch.nn.SyncBatchNorm
 * does between its all_gather and its normalisation kernel. */
int ssl4gie_bn_combine_stats(const float* gathered, int world, int C, float eps, float momentum,
                             float* running_mean, float* running_var, float* mean, float* rstd, float* total,
                             void* stream);
int ssl4gie_bn_bwd_reduce(const void* dy, const void* y, const void* x, const float* mean,
                          const float* rstd, void* dres, float* sums, int relu, float* workspace,
                          int dtype, long long rows, int C, void* stream);
int ssl4gie_bn_bwd_apply(const void* dy, const void* y, const void* x, const float* gamma,
                         const float* mean, const float* rstd, const float* sums, float inv_count,
                         void* dx, int relu, float* workspace, int dtype, long long rows, int C,
                         void* stream);
/* The two SyncBatchNorm backward halves for BatchNorm + ReLU without a residual input, with the ReLU mask rebuilt
 * from x and the forward's coefficients (gamma, beta and the GLOBAL mean / rstd) as ssl4gie_bn_bwd_xmask does:
 * the ReLU output is read by neither pass. */
int ssl4gie_bn_bwd_reduce_xmask(const void* dy, const void* x, const float* gamma, const float* beta,
                                const float* mean, const float* rstd, float* sums, float* workspace, int dtype,
                                long long rows, int C, void* stream);
int ssl4gie_bn_bwd_apply_xmask(const void* dy, const void* x, const float* gamma, const float* beta,
                               const float* mean, const float* rstd, const float* sums, float inv_count, void* dx,
                               float* workspace, int dtype, long long rows, int C, void* stream);
/* SyncBatchNorm with the single-process fusions (ABI 7): the GLOBAL statistics come back from the exchange, so the
 * fused consumers take them instead of computing their own.  coef_stats: (mean, rstd, gamma, beta) -> coef [2][C]
 * (y = x coef[0] + coef[1]) for ssl4gie_bn_maxpool3x3s2_fwd and the SSL4GIE_EPI_AFFINE_AUX_RELU epilogue; apply_bits:
 * y = relu(x coef[0] + coef[1] (+ res)) and the ReLU bit map of ssl4gie_bn_fwd_partials_bits (bf16); bwd_reduce_bits:
 * ssl4gie_bn_bwd_reduce with the mask read from that bit map (dres = the masked gradient, then ssl4gie_bn_bwd_apply
 * on dres with relu = 0).  Reference: torch.nn.SyncBatchNorm (convert_sync_batchnorm, Models/moco_v3/main_moco.py:196,
 * Depth_estimation/train_depth.py:225). */
int ssl4gie_bn_coef_stats(const float* mean, const float* rstd, const float* gamma, const float* beta, float* coef,
                          int C, void* stream);
int ssl4gie_bn_apply_bits(const void* x, const float* coef, const void* res, void* y, unsigned char* relu_bits,
                          int dtype, long long rows, int C, void* stream);
int ssl4gie_bn_bwd_reduce_bits(const void* dy, const unsigned char* relu_bits, const void* x, const float* mean,
                               const float* rstd, void* dres, float* sums, float* workspace, int dtype,
                               long long rows, int C, void* stream);
/* MoCo._update_momentum_encoder (moco/builder.py:57-61): dst = dst m + src (1 - m), fp32, over a
 * whole parameter-arena slice */
int ssl4gie_ema_update(float* dst, const float* src, float m, long long n, void* stream);
/* MaxPool2d(3, stride 2, pad 1) with the argmax window position saved (first maximum in row-major
 * scan order); backward in gather form.  Global average pool -> fp32 [B, C] and its gradient. */
/* The same forward / SyncBatchNorm statistics with the batch statistics taken from per-128-row
 * partial sums `partial` [parts][2][C] written by the producing GEMM (ssl4gie_gemm_desc::colstats)
 * instead of a pass over x.  workspace: ssl4gie_bn_workspace_bytes(rows, C). */
int ssl4gie_bn_fwd_partials(const void* x, const float* partial, int parts, const float* gamma,
                            const float* beta, const void* res, void* y, float* mean, float* rstd,
                            float* running_mean, float* running_var, float momentum, float eps,
                            int relu, float* workspace, int dtype, long long rows, int C,
                            void* stream);
/* The statistics half of ssl4gie_bn_fwd_partials alone: mean / rstd / running statistics and coef [2][C] with
 * y = x coef[0][c] + coef[1][c], for a consumer that applies the normalisation itself — the
 * SSL4GIE_EPI_AFFINE_AUX_RELU epilogue of the 1x1 convolution recomputed after its statistics-only product
 * (torchvision Bottleneck conv3 + bn3 (+ identity, ReLU) and downsample under torch.no_grad(): MoCo's momentum
 * encoder, moco/builder.py:127-135). */
int ssl4gie_bn_coef_partials(const float* partial, int parts, const float* gamma, const float* beta,
                             float* mean, float* rstd, float* running_mean, float* running_var,
                             float momentum, float eps, float* coef, float* workspace, long long rows, int C,
                             void* stream);
int ssl4gie_bn_stats_partials(const float* partial, int parts, float* mean, float* var,
                              float* workspace, long long rows, int C, void* stream);
int ssl4gie_maxpool3x3s2_fwd(const void* x, void* y, unsigned char* arg, int dtype, int B, int H,
                             int W, int C, void* stream);
int ssl4gie_maxpool3x3s2_bwd(const void* dy, const unsigned char* arg, void* dx, int dtype, int B,
                             int H, int W, int C, void* stream);
/* The same pool over act(x coef[0][c] + coef[1][c]), act = ReLU if `relu` (coef [2][C] as ssl4gie_bn_coef_partials
 * writes it): torchvision ResNet's bn1 -> relu -> maxpool behind the stem convolution in ONE pass over the
 * convolution's output — the normalised map is never written (the backward rebuilds its ReLU mask from the
 * convolution output: ssl4gie_bn_bwd_xmask).  Values and argmax equal ssl4gie_bn_fwd_partials followed by
 * ssl4gie_maxpool3x3s2_fwd bit for bit.  C % 8 == 0 (bf16) / C % 4 == 0 (fp32). */
int ssl4gie_bn_maxpool3x3s2_fwd(const void* x, const float* coef, int relu, void* y, unsigned char* arg,
                                int dtype, int B, int H, int W, int C, void* stream);
int ssl4gie_avgpool_fwd(const void* x, float* y, int dtype, int B, int HW, int C, void* stream);
int ssl4gie_avgpool_bwd(const float* dy, void* dx, int dtype, int B, int HW, int C, void* stream);

/* ---------------------------------------------------------------- optimizer steps over the arena
 * One parameter arena = S segments (one per parameter, 64-element aligned; padding stays zero).
 * Device tables: seg_start [S+1] int64 element offsets, seg_lr [S] (< 0: skip the segment: frozen or
 * without a gradient this step), seg_wd [S], seg_mat [S] (LARS: 1 for p.ndim > 1).
 * adamw_arena: torch.optim.AdamW's update (main_pretrain.py:179-180, train_depth.py:280), `step` =
 *   1-based step count for the bias corrections; m / v are the flat moment buffers.
 * lars_arena: Models/moco_v3/moco/optimizer.py:18-43: matrices dp = (g + wd p) trust |p|/|dp|
 *   (1 where a norm is 0), vectors dp = g; mu = momentum mu + dp; p -= lr mu. */
int ssl4gie_adamw_arena(float* p, const float* g, float* m, float* v, const long long* seg_start,
                        const float* seg_lr, const float* seg_wd, int S, float beta1, float beta2,
                        float eps, int step, long long n, void* stream);
/* the same step, also writing the bf16 operand copy of every UPDATED element into lp_bf16 [n] (the
 * flat shadow arena the GEMM operands are views of; skipped segments are left as they are) — saves
 * the separate cast pass over the arena.  lp_bf16 = NULL: as ssl4gie_adamw_arena. */
int ssl4gie_adamw_arena_lp(float* p, const float* g, float* m, float* v, const long long* seg_start,
                           const float* seg_lr, const float* seg_wd, int S, float beta1, float beta2,
                           float eps, int step, long long n, void* lp_bf16, void* stream);
/* the same step restricted to the arena elements [lo, hi) (multiples of 4; whole segments in
 * practice): lets the update of a transformer block's parameters be enqueued on a side stream as
 * soon as that block's backward is, behind the rest of the backward pass (optim.ArenaAdamW,
 * overlap_backward).  The union of the ranges of one step must cover [0, n) exactly once. */
int ssl4gie_adamw_arena_range(float* p, const float* g, float* m, float* v, const long long* seg_start,
                              const float* seg_lr, const float* seg_wd, int S, float beta1, float beta2,
                              float eps, int step, long long lo, long long hi, void* lp_bf16,
                              void* stream);
size_t ssl4gie_lars_workspace_bytes(int S);
int ssl4gie_lars_arena(float* p, const float* g, float* mu, const long long* seg_start,
                       const float* seg_lr, const float* seg_wd, const float* seg_mat, int S,
                       float momentum, float trust, float* workspace, long long n, void* stream);

/* ---------------------------------------------------------------- input pipeline
 * transforms.ToTensor() + transforms.Normalize(mean, std) (Depth_estimation/Data/dataloaders.py:
 * 55-63) on the device: uint8 HWC [B, H, W, 3] -> fp32 NCHW [B, 3, H, W] = (x / 255 - mean) / std.
 * mean / std are HOST arrays of 3 floats; H*W % 4 == 0. */
int ssl4gie_normalize_u8(const unsigned char* img, float* out, const float* mean, const float* std,
                         int B, int H, int W, void* stream);

/* ---------------------------------------------------------------- detection pyramid glue (channels-last)
 * ViTDet_FPN (Models/models.py:213-259) around its GEMM-shaped convolutions:
 * maxpool2x2: nn.MaxPool2d(2) (:218); backward recomputes the window's first maximum from x.
 * gelu_map: nn.GELU (:241), exact erf form; dy != NULL gives dy * gelu'(x).
 * map_layernorm: nn.LayerNorm((C, H, W)) (:220-222 ...): per-image statistics over all M = H*W*C
 *   elements (biased variance, eps inside the sqrt) and a per-ELEMENT affine; w / bias / dw / db
 *   are fp32 [M] in the map's own (channels-last) element order.  mean / rstd [B] are kept for
 *   backward.  M % 8 == 0. */
int ssl4gie_maxpool2x2_fwd(const void* x, void* y, int dtype, int B, int H, int W, int C,
                           void* stream);
int ssl4gie_maxpool2x2_bwd(const void* x, const void* dy, void* dx, int dtype, int B, int H, int W,
                           int C, void* stream);
int ssl4gie_gelu_map(const void* x, const void* dy, void* out, int dtype, long long n, void* stream);
size_t ssl4gie_map_layernorm_workspace_bytes(int B);
int ssl4gie_map_layernorm_fwd(const void* x, const float* w, const float* bias, void* y, float* mean,
                              float* rstd, float eps, float* workspace, int dtype, int B, long long M,
                              void* stream);
int ssl4gie_map_layernorm_bwd(const void* x, const void* dy, const float* w, const float* mean,
                              const float* rstd, void* dx, float* dw, float* db, int accumulate,
                              float* workspace, int dtype, int B, long long M, void* stream);

/* ---------------------------------------------------------------- launch profiler (bench.py)
 * HIP events on the launch stream around every launch of the heavy kernels, used for the
 * `roofline` object of the bench line.  Process-global, not thread-safe, off by default.
 * kinds: 0 bf16 NT GEMM, 1 bf16 TN GEMM (kernel only, not its slab reduction), 2 fused attention
 * fwd, 3 fused attention bwd, 4 generic f32-MFMA GEMM.  flops are algorithmic (2MNK; attention
 * 4 B H N^2 hd forward, 10 B H N^2 hd backward). */
#define SSL4GIE_PROF_KINDS 7 /* 0 NT GEMM, 1 TN GEMM, 2 / 3 attention fwd / bwd, 4 generic GEMM (FLOPs); 5 BatchNorm, 6 LayerNorm (algorithmic BYTES in the `flops` slot) */
int ssl4gie_prof_begin(int max_launches);
int ssl4gie_prof_collect(double* ms, double* flops, long long* launches);
int ssl4gie_prof_end(void);

/* ---------------------------------------------------------------- finetune losses (value + gradient)
 * ScaleAndShiftInvariantLoss(alpha, scales) of the depth finetune step
 * (Depth_estimation/Metrics/losses.py:120-146; used at train_depth.py:43,280): pred, target fp32
 * [B, H, W] (the [B, 1, H, W] maps), valid pixels = target > 0.  Writes the scalar loss and
 * dpred = dloss/dpred (fp32 [B, H, W]) in five launches; deterministic two-stage reductions.
 * SoftDiceLoss(smooth) of the segmentation step (Binary_segmentation/Metrics/losses.py:5-24): logits,
 * target fp32 [B, n]; loss = 1 - mean_b 2 (sum s t + smooth) / (sum s^2 + sum t^2 + smooth),
 * s = sigmoid(logits); writes the loss and dlogits. */
size_t ssl4gie_ssi_loss_workspace_bytes(int B, int H, int W);
int ssl4gie_ssi_loss(const float* pred, const float* target, float* loss, float* dpred, int B, int H,
                     int W, float alpha, int scales, void* workspace, void* stream);
size_t ssl4gie_dice_loss_workspace_bytes(int B);
int ssl4gie_dice_loss(const float* logits, const float* target, float* loss, float* dlogits, int B,
                      long long n, float smooth, void* workspace, void* stream);

/* ---------------------------------------------------------------- direct xGMI gradient all-reduce
 * replaces the NCCL bucket all-reduce of DistributedDataParallel (Models/mae/main_pretrain.py:175,
 * Depth_estimation/train_depth.py:226-229, Models/moco_v3/main_moco.py:208) for ONE node of up to 8
 * fully connected GPUs: every rank pushes chunk p of a bucket straight into peer p's memory
 * (reduce-scatter), the owners push the reduced chunks back (all-gather) — one hop, all 7 links at
 * once, instead of a ring bound by one link.  csrc/allreduce.hip has the protocol.
 *   init     allocates this rank's exchange region (the ONE allocation this interface makes; sized for
 *            buckets of up to max_elems fp32) and writes ssl4gie_allreduce_direct_blob_bytes() bytes
 *            of IPC description to export_blob; the caller exchanges the blobs of all ranks by any
 *            out-of-band means (ssl4gie_amd.parallel: torch.distributed.all_gather_object);
 *   connect  maps the peers' regions from the `world` blobs laid end to end in rank order;
 *   enqueue  grad[0 .. n_elems) <- scale * sum over ranks, in place, on `stream` (5 launches, no host
 *            synchronisation).  Every rank must enqueue the same sequence of sizes; sums run in rank
 *            order on every rank, so all ranks end with bitwise identical values;
 *   error    the handle's sticky error word: 0, or (sequence number << 8 | 1 + peer rank) of the first
 *            bucket in which a waiting kernel gave up on a peer (bounded poll: 10 min by default, SSL4GIE_AR_TIMEOUT_S
 *            or set_timeout).  Such a kernel writes NaN instead of stale sums; from then on enqueue
 *            returns SSL4GIE_EPEER (1001) and ssl4gie_amd.parallel.DataParallel raises.  Read without
 *            synchronising (mapped host memory);
 *   destroy  unmaps / frees (after the streams that used the handle have drained).
 * init fails (hipError_t) when fine-grained device memory is not available: peer stores and in-kernel
 * flag polls are not coherent on coarse-grained memory, so there is no fallback to it. */
/* Diagnostics: the in-kernel time stamps of the 256x256 NT kernel (SSL4GIE_NT256_NOEPI=4; csrc/gemm_nt256.hip):
 * [16 workgroups][16 tiles][5] uint64 ticks of the 100 MHz s_memrealtime counter, copied to host memory. */
int ssl4gie_debug_nt256_stamps(void* dst, size_t bytes);

typedef struct ssl4gie_ar_handle ssl4gie_ar_handle;
size_t ssl4gie_allreduce_direct_blob_bytes(void);
int ssl4gie_allreduce_direct_init(int rank, int world, size_t max_elems, void* export_blob,
                                  ssl4gie_ar_handle** out);
int ssl4gie_allreduce_direct_connect(ssl4gie_ar_handle* h, const void* all_blobs);
int ssl4gie_allreduce_direct_enqueue(ssl4gie_ar_handle* h, float* grad, size_t n_elems, float scale,
                                     void* stream);
/* All-gather of n_elems floats per rank over the same handle (n_elems <= ceil(max_elems / world)):
 * dst[w * n_elems + i] = rank w's src[i], in rank order on every rank; 3 launches, no host synchronisation.
 * Carries nn.SyncBatchNorm's per-layer (mean, var, count) exchange and its backward sums
 * (reference Models/moco_v3/main_moco.py:196, Depth_estimation/train_depth.py:225) without an RCCL launch;
 * use a handle of its own per stream (calls on one handle are ordered by the stream they are enqueued on). */
int ssl4gie_allgather_direct_enqueue(ssl4gie_ar_handle* h, const float* src, size_t n_elems, float* dst,
                                     void* stream);
unsigned ssl4gie_allreduce_direct_error(const ssl4gie_ar_handle* h);
int ssl4gie_allreduce_direct_set_timeout(ssl4gie_ar_handle* h, double seconds);
int ssl4gie_allreduce_direct_destroy(ssl4gie_ar_handle* h);

#ifdef __cplusplus
}
#endif
#endif
