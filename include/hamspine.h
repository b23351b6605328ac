/*
 * hamspine.h -- C ABI of libhamspine_hip.so (MI355X / gfx950 only).
 *
 * This is the drop-in boundary underneath the reference's Python nn.Module surface
 * (reference: model.py:21-345, encoder.py:13-134, modules/fusion_blocks.py, modules/heads.py,
 * mibf_net/model_resnet.py:10-94, mibf_net/attention.py:31-70).  The reference has no FFI of its
 * own: every kernel it runs is an ATen/cuDNN/cuBLAS kernel chosen by torchvision / transformers /
 * torch.nn.  Each entry point below names the reference call site whose arithmetic it replaces.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; the caller owns every buffer (device memory).
 *   - `stream` is a hipStream_t passed as void*.
 *   - every function returns hs_status (0 = ok); hs_last_error() gives the text (thread local).
 *   - element types: HS_F32 (exact mode, f32-in MFMA == fmaf chain) or HS_BF16 (throughput mode,
 *     bf16-in MFMA, f32 accumulate).  Parameters, biases, norm scales, statistics, losses and
 *     parameter gradients are always f32.
 *   - image activations are NHWC ("channels_last") in memory.
 */
#ifndef HAMSPINE_H
#define HAMSPINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int hs_status;
enum { HS_OK = 0, HS_ERR_ARG = 1, HS_ERR_HIP = 2, HS_ERR_UNSUPPORTED = 3 };
enum { HS_F32 = 0, HS_BF16 = 1 };

const char* hs_last_error(void);
int hs_version(void);
/* 1 if a gfx950 device is visible; never throws. */
int hs_device_ok(void);

/* ------------------------------------------------------------------------------------------- */
/* GEMM / implicit-GEMM convolution core                                                        */
/*   D[m][n] = epilogue( alpha * sum_k A(m,k) * B(n,k) )                                        */
/* Replaces: torch.nn.functional.linear / conv2d and their backward (cuBLAS / cuDNN under the   */
/* reference's torchvision ResNet, encoder.py:35-42, and transformers BertLayer, encoder.py:131).*/
/* ------------------------------------------------------------------------------------------- */

/* how operand A supplies A(m,k) */
enum {
    HS_A_KC = 0,     /* A[m*lda + k]                      (k contiguous)                        */
    HS_A_RC = 1,     /* A[k*lda + m]                      (m contiguous; "transposed" operand)  */
    HS_A_CONV = 2,   /* im2col gather from an NHWC image: m=(n,p,q), k=(r,s,c)     (conv fwd)   */
    HS_A_DGRAD = 3   /* gather from dY (NPQK):            m=(n,h,w), k=(r,s,ko)    (conv dgrad) */
};
/* how operand B supplies B(n,k) */
enum {
    HS_B_KC = 0,     /* B[n*ldb + k]                                                            */
    HS_B_RC = 1,     /* B[k*ldb + n]                                                            */
    HS_B_WDGRAD = 2, /* filter KRSC read as n=c, k=(r,s,ko)                       (conv dgrad)  */
    HS_B_CONV = 3    /* im2col gather with n=(r,s,c), k=(n,p,q)                   (conv wgrad)  */
};
enum { HS_ACT_NONE = 0, HS_ACT_RELU = 1, HS_ACT_GELU = 2 };
/* epilogue multiplier modes (backward) */
enum { HS_MUL_NONE = 0, HS_MUL_GELU_GRAD = 1, HS_MUL_RELU_MASK = 2 };

typedef struct hs_conv_geom {
    int32_t N, H, W, C;      /* input image: batch, height, width, channels                     */
    int32_t P, Q, K;         /* output image: height, width, channels                           */
    int32_t R, S;            /* filter                                                          */
    int32_t stride, pad;
    int32_t row_pitch;       /* elements between input rows   (W*C unless pre-padded)           */
    int32_t img_pitch;       /* elements between input images (H*row_pitch)                     */
    int32_t qstep;           /* elements per output-column step (stride*C unless stem packing)  */
    int32_t no_bounds;       /* 1: the input is pre-padded, skip the h/w range test             */
} hs_conv_geom;

struct hs_bn_params;
typedef struct hs_gemm_params {
    int32_t dtype;           /* element type of A and B (HS_F32 | HS_BF16)                      */
    int32_t a_kind, b_kind;
    int32_t M, N, K;
    const void* A;
    const void* B;
    int64_t a_elems, b_elems; /* addressable elements behind A / B (bounds for buffer loads)    */
    int32_t lda, ldb;
    hs_conv_geom g;
    /* batching: blockIdx.z = b; operand offset = (b / batch_inner)*bs0 + (b % batch_inner)*bs1 */
    int32_t batch, batch_inner;
    int64_t a_bs0, a_bs1, b_bs0, b_bs1, d_bs0, d_bs1;
    /* split-K (batch must be 1): partial sums go to `splitk_ws` (f32, hs_gemm_splitk_ws_bytes(p) bytes: one M*N slab per
       slice plus, beyond 8 slices, one per group of 8).  bf16: reduced inside the launch (two-level last-arriver hand-off,
       fixed summation order) before the epilogue; f32: by a second kernel that applies the epilogue.  0/1 = off. */
    int32_t split_k;
    float* splitk_ws;
    /* epilogue */
    void* D;
    int32_t ldd;
    int32_t out_dtype;       /* HS_F32 | HS_BF16                                                */
    float alpha;
    const float* bias;       /* [N] or NULL                                                     */
    int32_t act;
    void* D_preact;          /* optional second output: value before `act` (same type/ld as D)  */
    const void* residual;    /* optional, added last; same type as D                            */
    int32_t ldr;
    float dropout_p;         /* applied after act, before residual                              */
    uint64_t dropout_seed;
    int32_t mul_mode;        /* backward multipliers                                            */
    const void* mul_src;     /* u (GELU input) or y (ReLU output); same type as A, ld = ldm     */
    int32_t ldm;
    int32_t accumulate;      /* 1: D += result (f32 outputs only)                               */
    /* segmented output (optional): rows [i*seg_rows, (i+1)*seg_rows) go to D_seg[i-1] for i = 1, 2 (row index
       rebased), rows below seg_rows to D.  Lets one GEMM write three parameter gradients (fused QKV wgrad). */
    int32_t seg_rows;
    void* D_seg[2];
    /* optional per-column scale applied before the bias: v = alpha * acc * colscale[n] + bias[n] (an eval-mode BatchNorm
       folded into the convolution), and residual_before_act = 1 to add `residual` before the activation
       (y = relu(bn(conv) + identity)) instead of after it */
    const float* colscale;
    int32_t residual_before_act;
    /* optional (bf16 operands, no split-K, no batching): per (row tile, column) statistics of the result for a following
       BatchNorm, written as (count, mean, M2) triples to colstats[(tile_row * N + n) * 3 ...]; hs_gemm_stat_rows(p) gives
       the number of tile rows.  Saves BatchNorm's own pass over the convolution output (hs_bn_params.partial_rows). */
    float* colstats;
    /* optional (bf16, a_kind = HS_A_RC, no split-K, batch 1): rowsum_a[m] = sum over k of A[k][m] in f32 -- with A = dY this
       is the bias gradient of the Linear whose weight gradient the GEMM computes, so db costs one extra MFMA per A
       fragment in the first tile column instead of a column-sum pass over dY.  With seg_rows the sums of segment i go to
       rowsum_seg[i-1] (row index rebased), like D_seg. */
    float* rowsum_a;
    float* rowsum_seg[2];
    /* optional (bf16 result, batch 1, plain epilogue: no bias / activation / residual / multiplier, alpha = 1; data-gradient
       layouts): the GEMM result is the gradient of relu(BatchNorm(c)) -- the epilogue also produces that BatchNorm's backward
       sums per (row tile, channel): bnb_partials[(tile_row * N + n) * 2 + {0, 1}] = sum over the tile's rows of dz and dz * xhat,
       dz = D * (fma(c, scale, shift) > 0), xhat = (c - mean) * invstd; hs_gemm_tile_rows(p) gives the number of tile rows.
       hs_bn_bwd_params.partial_rows then skips BatchNorm's own partial-sum pass (reference: the same sums torch's
       batch_norm_backward takes in its own kernel). */
    const void* bnb_x;          /* c: [M][N], leading dimension ldd, operand dtype */
    const float* bnb_scale;
    const float* bnb_shift;
    const float* bnb_mean;
    const float* bnb_invstd;
    float* bnb_partials;
    /* optional (with colstats, when hs_gemm_bn_finish_rows(p) > 0): the launch also FINISHES the statistics of the BatchNorm
       that follows -- the last workgroup of each column tile merges the tile rows' partials and writes bn_finish->save_mean /
       save_invstd / scale / shift and updates running_mean / running_var (train-mode arithmetic of hs_batchnorm_fwd: biased
       variance for normalisation, unbiased for running_var).  colstats must then hold hs_gemm_bn_finish_rows(p) rows.  The
       caller runs hs_batchnorm_fwd with stats_done = 1 (apply pass only): one launch per BatchNorm less on the stream. */
    const struct hs_bn_params* bn_finish;
    /* optional (with bnb_partials): the result is the gradient of relu(BatchNorm(c) + identity) -- a residual block's OUTPUT --
       so the ReLU mask is read from the saved output bnb_y ([M][N], leading dimension ldd: mask = y > 0) instead of being
       recomputed from c, and the launch may carry `residual` (the identity path's gradient, added BEFORE the sums are taken:
       reference torchvision Bottleneck.forward `out += identity; out = relu(out)`).  bnb_scale / bnb_shift are then unused. */
    const void* bnb_y;
    /* optional (with bnb_partials, when hs_gemm_bnb_finish_rows(p) > 0): the launch also FINISHES that BatchNorm's backward sums --
       the last workgroup of each column tile adds the tile rows' partials and writes bnb_finish->dbeta / dgamma and, behind
       the hs_gemm_bnb_finish_rows(p) rows of bnb_partials, the four per-channel coefficient vectors of the apply pass
       (gamma, M, training are read from *bnb_finish).  The caller then runs hs_batchnorm_bwd with ws = bnb_partials,
       partial_rows = hs_gemm_bnb_finish_rows(p) and sums_done = 1 (apply pass only): one launch per BatchNorm less on the
       stream (reference: torch's batch_norm_backward reduce + elementwise kernels). */
    const struct hs_bn_bwd_params* bnb_finish;
} hs_gemm_params;

hs_status hs_gemm(const hs_gemm_params* p, void* stream);
/* workspace (bytes) hs_gemm needs in p->splitk_ws for the given p (0 when split_k <= 1). */
int64_t hs_gemm_splitk_ws_bytes(const hs_gemm_params* p);
/* number of row tiles hs_gemm will use for p (rows of a colstats buffer); 0 when p cannot produce colstats. */
int32_t hs_gemm_stat_rows(const hs_gemm_params* p);
/* rows the colstats buffer needs when the launch for p (colstats set) can also finish the BatchNorm statistics
   (hs_gemm_params.bn_finish): tile rows + merge rows; 0 when this launch cannot. */
int32_t hs_gemm_bn_finish_rows(const hs_gemm_params* p);
/* rows bnb_partials must hold (tile rows + merge rows) when the launch for p can also finish the BatchNorm-backward sums
   (hs_gemm_params.bnb_finish), else 0 (split-K launches, layouts without the rider). */
int32_t hs_gemm_bnb_finish_rows(const hs_gemm_params* p);
/* tile rows of the launch hs_gemm makes for p (p->split_k as it will be launched): rows of bnb_partials */
int32_t hs_gemm_tile_rows(const hs_gemm_params* p);
/* heuristic split-K factor for a (M,N,K) problem so that the grid fills 256 CUs. */
int32_t hs_gemm_suggest_split(int32_t M, int32_t N, int32_t K, int32_t dtype);
/* Optional timing of every GEMM-core launch with HIP events on its stream (measurement only).
   Classes: 0 bf16 GEMM, 1 bf16 implicit-GEMM conv, 2 f32 GEMM, 3 f32 conv.  hs_prof_collect synchronises the
   device, fills 4 entries of algorithmic flops / elapsed ms / launch counts and clears the records. */
/* measurement only: force a tile configuration (0 128x128, 1 128x64, 2 64x64, -1 auto) and ablation bits
   (16 = plain n-fastest tile order instead of the L2-grouped one; 32 = always the generic epilogue body instead of
   the specialised ones; results stay correct).  cfg also accepts 4 (256x128, 8 waves), 5 (128x128, BK 32), 6 (256x128, BK 32)
   for plain bf16 GEMMs. */
void hs_gemm_debug(int32_t cfg_override, int32_t ablate);
/* measurement only: while device_buffer is not NULL every bf16 hs_gemm launch writes 6 shader-clock stamps per workgroup
   (start, first DMA issued, first K tile landed, K loop done, epilogue done, unused) to it; size it 48 bytes x workgroups. */
void hs_gemm_debug_stamps(void* device_buffer);
void hs_prof_enable(int32_t on);
hs_status hs_prof_collect(double* flops, double* ms, int64_t* launches);
/* measurement only: append one CSV line per recorded launch (class, operand combo, tile cfg, M, N, K, batch, split_k,
   filter R, stride, milliseconds) to `path` and clear the records. */
hs_status hs_prof_dump(const char* path);
/* measurement only: mean elapsed time (us) of `n` empty kernels inside the profiler's event bracket, i.e. what the
   bracket adds to every timed launch. */
hs_status hs_prof_calibrate(void* stream, int32_t n, float* avg_us);

/* ------------------------------------------------------------------------------------------- */
/* BatchNorm2d over NHWC activations viewed as [M = N*H*W][C]                                    */
/* Replaces torch.nn.BatchNorm2d inside torchvision ResNet blocks (reference encoder.py:35-42,  */
/* mibf_net/model_resnet.py:15): train mode = batch statistics (biased var for normalisation,   */
/* unbiased for running_var, momentum 0.1), eval mode = running statistics.                     */
/* ------------------------------------------------------------------------------------------- */
typedef struct hs_bn_params {
    int32_t dtype;
    int32_t C;
    int64_t M;
    int32_t training;        /* 1: batch stats (+ running update), 0: running stats              */
    int32_t relu;            /* fuse ReLU into the apply pass                                    */
    float eps, momentum;
    const void* x;           /* [M][C]                                                           */
    const void* residual;    /* optional [M][C], added before the ReLU                           */
    void* y;                 /* [M][C] (NULL: statistics only)                                   */
    const float* gamma;
    const float* beta;
    float* running_mean;     /* may be NULL in training (no update)                              */
    float* running_var;
    float* save_mean;        /* [C] out                                                          */
    float* save_invstd;      /* [C] out                                                          */
    float* scale;            /* [C] out: gamma*invstd                                            */
    float* shift;            /* [C] out: beta - mean*gamma*invstd                                */
    void* ws;                /* hs_batchnorm_ws_bytes(M, C, dtype)                               */
    int64_t ws_bytes;
    int32_t partial_rows;    /* > 0: ws already holds that many rows of (count, mean, M2) partials per channel
                                (hs_gemm colstats): skip the statistics pass over x                */
    int32_t stats_done;      /* 1: save_mean / save_invstd / scale / shift (and the running statistics) are already
                                written (hs_gemm_params.bn_finish): apply pass only                 */
} hs_bn_params;

typedef struct hs_bn_bwd_params {
    int32_t dtype;
    int32_t C;
    int64_t M;
    int32_t training;
    int32_t relu;            /* forward fused a ReLU: dz = dy * (y > 0)                          */
    const void* dy;
    const void* y;           /* forward output (needed when relu, unless scale / shift are given) */
    const void* x;           /* forward input                                                    */
    const float* gamma;
    const float* save_mean;
    const float* save_invstd;
    void* dx;                /* [M][C] (NULL: parameter gradients only)                          */
    void* dres;              /* optional: gradient of the residual input (= dz)                  */
    float* dgamma;
    float* dbeta;
    void* ws;
    int64_t ws_bytes;
    /* optional, relu without a residual input: with y == NULL the ReLU mask is recomputed as fma(x, scale, shift) > 0 from the
       forward's scale / shift (hs_bn_params.scale / .shift of the same layer): one activation read less in both passes */
    const float* scale;
    const float* shift;
    /* > 0: ws already holds [partial_rows][C][2] backward sums (hs_gemm_params.bnb_partials): no partial pass is made */
    int32_t partial_rows;
    /* 1 (with partial_rows > 0): the producing GEMM also finished the sums (hs_gemm_params.bnb_finish): dgamma / dbeta are
       written and the coefficients sit behind the partial rows -- only the apply pass runs */
    int32_t sums_done;
} hs_bn_bwd_params;

hs_status hs_batchnorm_fwd(const hs_bn_params* p, void* stream);
hs_status hs_batchnorm_bwd(const hs_bn_bwd_params* p, void* stream);
int64_t hs_batchnorm_ws_bytes(int64_t M, int32_t C, int32_t dtype);

/* LayerNorm over the last dim of [M][H]. Replaces torch.nn.LayerNorm (reference                */
/* modules/fusion_blocks.py:17-34, modules/heads.py:36; transformers Bert*Output.LayerNorm).    */
hs_status hs_layernorm_fwd(int32_t dtype, const void* x, const float* gamma, const float* beta, void* y,
                           float* mean, float* rstd, int64_t M, int32_t H, float eps, void* stream);
hs_status hs_layernorm_bwd(int32_t dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                           const float* rstd, void* dx, float* dgamma, float* dbeta, void* ws, int64_t ws_bytes,
                           int64_t M, int32_t H, void* stream);
/* the same, also serving the layer in front of the LayerNorm (BertSelfOutput / BertOutput: y = LN(dropout(dense(h)) + x)):
   dx_dropped = dx * dropout mask (same (seed, element index) mask as the forward; p = 0: a copy; may be NULL) and
   dbias[H] = column sums of dx_dropped = that dense layer's bias gradient.  Saves the stand-alone dropout and column-sum
   passes of transformers BertSelfOutput / BertOutput backward. */
hs_status hs_layernorm_bwd_pre(int32_t dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                               const float* rstd, void* dx, float* dgamma, float* dbeta, void* dx_dropped, float* dbias,
                               float dropout_p, uint64_t seed, void* ws, int64_t ws_bytes, int64_t M, int32_t H, void* stream);
int64_t hs_layernorm_bwd_ws_bytes(int64_t M, int32_t H);

/* ------------------------------------------------------------------------------------------- */
/* pooling, packing, casts, reductions, softmax, embeddings, loss, optimizer (HBM-bound helpers) */
/* ------------------------------------------------------------------------------------------- */
#define HS_CAST_MAX 48
#define HS_ADAM_MAX 32

/* MaxPool2d(k, s, p) on NHWC; idx (uint8, same shape as y) keeps the arg-max tap for the backward.
   Replaces torchvision ResNet.maxpool (reference encoder.py:67). */
hs_status hs_maxpool_fwd(int32_t dtype, const void* x, void* y, void* idx, int32_t N, int32_t H, int32_t W, int32_t C,
                         int32_t ksize, int32_t stride, int32_t pad, void* stream);
hs_status hs_maxpool_bwd(int32_t dtype, const void* dy, const void* idx, void* dx, int32_t N, int32_t H, int32_t W,
                         int32_t C, int32_t ksize, int32_t stride, int32_t pad, void* stream);
/* y[b][:] = mean_t x[b][t][:]  (AdaptiveAvgPool / tokens.mean(dim=1); reference
   modules/fusion_blocks.py:97-98,170-178, model.py:283-290, torchvision ResNet.avgpool). */
hs_status hs_mean_tokens_fwd(int32_t dtype, const void* x, void* y, int32_t B, int32_t Nt, int32_t H, int32_t out_f32,
                             void* stream);
hs_status hs_mean_tokens_bwd(int32_t dtype, const void* dy, void* dx, int32_t B, int32_t Nt, int32_t H, int32_t dy_f32,
                             void* stream);
/* f32 NCHW image -> zero-bordered NHWC4 image of type dtype: [N][Hp][Wp][4], pixel (h,w) at (h+pad, w+pad). */
hs_status hs_pack_image(int32_t dtype, const float* x, void* y, int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t Hp,
                        int32_t Wp, int32_t pad, void* stream);
/* stem filter [K][R][S][Cin] f32 (channels_last storage of (K,Cin,R,S)) <-> packed [K][R][8][4]. */
hs_status hs_pack_stem_weight(int32_t dtype, const float* w, void* out, int32_t K, int32_t R, int32_t S, int32_t Cin,
                              void* stream);
hs_status hs_unpack_stem_wgrad(const float* g, float* dw, int32_t K, int32_t R, int32_t S, int32_t Cin, void* stream);
/* one launch per HS_CAST_MAX tensors: dst[i] (bf16) = src[i] (f32). */
hs_status hs_cast_f32_to_bf16_multi(int32_t count, const float* const* src, void* const* dst, const int64_t* n,
                                    void* stream);
/* dst[c][r] = src[r][c] for a bf16 matrix of R rows x C columns (leading dimensions in elements; R, C, ld % 8 == 0).
   Used by the BertLayer backward to turn its weight gradients (dY^T X, reduction over the row index of both operands)
   into K-contiguous GEMMs; replaces nothing in the reference (torch.nn.Linear's backward is one cuBLAS call there). */
hs_status hs_transpose_bf16(const void* src, void* dst, int32_t R, int32_t C, int64_t ld_src, int64_t ld_dst, void* stream);
/* the same for up to HS_TRANSPOSE_MAX matrices per launch (entries beyond that go to further launches) */
#define HS_TRANSPOSE_MAX 8
hs_status hs_transpose_bf16_multi(int32_t count, const void* const* src, void* const* dst, const int32_t* R, const int32_t* C,
                                  const int64_t* ld_src, const int64_t* ld_dst, void* stream);
/* out = a*x + b*y (y may be NULL); dtypes are HS_F32/HS_BF16 for inputs (shared) and output. */
hs_status hs_axpby(int32_t in_dtype, int32_t out_dtype, const void* x, const void* y, void* out, int64_t n, float a,
                   float b, void* stream);
/* out[i] = x[i] * keep(seed, i) / (1-p): the same call is its own backward. */
hs_status hs_dropout(int32_t dtype, const void* x, void* out, int64_t n, float p, uint64_t seed, void* stream);
hs_status hs_relu_fwd(int32_t dtype, const void* x, void* out, int64_t n, void* stream);
hs_status hs_relu_bwd(int32_t dtype, const void* dy, const void* y, void* dx, int64_t n, void* stream);
/* dx = dy * gelu'(u)  (erf GELU). */
hs_status hs_gelu_bwd(int32_t dtype, const void* dy, const void* u, void* dx, int64_t n, void* stream);
/* out = x * scalar[0] with the scalar on the device (loss backward without a host sync). */
hs_status hs_mul_dev_scalar(const float* x, const float* scalar, float* out, int64_t n, void* stream);
/* out[n] (+)= sum_m x[m*ld + n]  (bias gradients). */
hs_status hs_colsum(int32_t dtype, const void* x, int64_t M, int32_t N, int32_t ld, float* out, void* ws,
                    int64_t ws_bytes, int32_t accumulate, void* stream);
int64_t hs_colsum_ws_bytes(int64_t M, int32_t N);
/* attention probabilities from materialised f32 scores; rows = (b,h,q); mask[b][k] == 0 masks key k.
   P gets the probabilities (padding columns Lk..ldP-1 zeroed), P_drop (optional) the dropped-out copy. */
hs_status hs_softmax_fwd(int32_t dtype, const float* S, const int64_t* mask, void* P, void* P_drop, int64_t rows,
                         int32_t Lk, int32_t ldS, int32_t ldP, int32_t rows_per_batch, float dropout_p, uint64_t seed,
                         void* stream);
hs_status hs_softmax_bwd(int32_t dtype, const float* dP, const void* P, void* dS, int64_t rows, int32_t Lk, int32_t ldG,
                         int32_t ldP, float dropout_p, uint64_t seed, void* stream);
/* BertEmbeddings (token_type_ids == 0): sum_out = word[ids]+pos[l]+type0 ; y = dropout(LN(sum_out)). */
hs_status hs_bert_embed_fwd(int32_t dtype, const int64_t* ids, const float* word, const float* pos, const float* type0,
                            const float* gamma, const float* beta, void* sum_out, void* y, float* mean, float* rstd,
                            int64_t tokens, int32_t L, int32_t H, int32_t V, float eps, float dropout_p, uint64_t seed,
                            void* stream);
/* dword (zero-filled by the caller) += scatter(dsum by ids), skipping ids == pad_id (nn.Embedding
   padding_idx; -1 = none); dpos[l] = sum_b dsum[b][l]. */
hs_status hs_bert_embed_bwd(int32_t dtype, const int64_t* ids, const void* dsum, float* dword, float* dpos, int32_t B,
                            int32_t L, int32_t H, int32_t V, int32_t pad_id, void* stream);
/* mean-reduced CrossEntropyLoss(weight, label_smoothing): writes the scalar loss, d loss / d logits
   (optional) and the per-row unreduced losses (optional). reference scripts/train.py:240,252-254. */
hs_status hs_cross_entropy(const float* logits, const int64_t* labels, const float* weight, float label_smoothing,
                           int32_t B, int32_t C, float* loss, float* dlogits, float* row_loss, void* stream);
/* fused multi-tensor Adam (decoupled=0) / AdamW (decoupled=1) step over f32 tensors. */
hs_status hs_adam_step_multi(int32_t count, float* const* params, const float* const* grads, float* const* exp_avg,
                             float* const* exp_avg_sq, const int64_t* n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int32_t step, int32_t decoupled, float grad_scale, void* stream);
/* the same, additionally writing bf16 copies of the updated parameters (bf16_shadow[i] may be NULL; the whole array may be
   NULL): the copies the composites read as weights once registered with hs_weight_shadow_set */
hs_status hs_adam_step_multi_shadow(int32_t count, float* const* params, const float* const* grads, float* const* exp_avg,
                                    float* const* exp_avg_sq, void* const* bf16_shadow, const int64_t* n, float lr, float beta1,
                                    float beta2, float eps, float weight_decay, int32_t step, int32_t decoupled,
                                    float grad_scale, void* stream);
/* Caller-scoped grouping of weight-gradient GEMMs: between _begin and _end, the K-contiguous ("nt") weight-gradient GEMMs
   that hs_linear_bwd issues on `stream` are collected and then launched as ONE grid (an MLP's two Linear layers fill the chip
   together where each alone does not).  The workspaces passed to the collected hs_linear_bwd calls must stay alive and must
   not overlap until _end returns.  No reference counterpart (cuBLAS launches every GEMM on its own). */
hs_status hs_wgrad_group_begin(void* stream);
hs_status hs_wgrad_group_end(void* stream);
/* bf16 shadow registry: while registered, every composite / tower reads `bf16` (n elements of the weight, same memory order)
   instead of casting `w` in its forward.  The caller keeps it equal to bf16(w).  bf16 = NULL forgets the entry.  No
   reference counterpart (torch autocast re-casts weights per call). */
hs_status hs_weight_shadow_set(const float* w, void* bf16);
void hs_weight_shadow_clear(void);

/* fused multi-tensor SGD, torch.optim.SGD semantics (the reference's fallback optimizer, scripts/train.py:309):
   g += wd*p; buf = first ? g : momentum*buf + g; g = nesterov ? g + momentum*buf : buf; p -= lr*g.  momentum_buf may be
   NULL when momentum == 0. */
hs_status hs_sgd_step_multi(int32_t count, float* const* params, const float* const* grads, float* const* momentum_buf,
                            const int64_t* n, float lr, float momentum, float weight_decay, int32_t nesterov, int32_t first,
                            float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------- */
/* small f32 operators at the fusion / head / loss boundary                                       */
/* ------------------------------------------------------------------------------------------- */
/* out[b][:] = x[b][t][:] as f32 (CLS pooling, reference modules/fusion_blocks.py:170-173). */
hs_status hs_select_token_fwd(int32_t dtype, const void* x, float* out, int32_t B, int32_t Nt, int32_t H, int32_t t,
                              void* stream);
hs_status hs_select_token_bwd(int32_t dtype, const float* dy, void* dx, int32_t B, int32_t Nt, int32_t H, int32_t t,
                              void* stream);
/* out[r] = [a[r] | b[r]] and its inverse (torch.cat(dim=1): fusion_blocks.py:186, model_resnet.py:59). */
hs_status hs_concat2(const float* a, int32_t Ha, const float* b, int32_t Hb, float* out, int64_t rows, void* stream);
hs_status hs_split2(const float* g, float* da, int32_t Ha, float* db, int32_t Hb, int64_t rows, void* stream);
/* the same on token tensors of either element type (global/local token concat, reference model.py:311-313). */
hs_status hs_concat2_t(int32_t dtype, const void* a, int32_t Ha, const void* b, int32_t Hb, void* out, int64_t rows,
                       void* stream);
hs_status hs_split2_t(int32_t dtype, const void* g, void* da, int32_t Ha, void* db, int32_t Hb, int64_t rows, void* stream);
/* out = a*b; b_mode 0: same shape, 1: b is (rows,1), 2: b is a scalar. */
hs_status hs_mul(const float* a, const float* b, float* out, int64_t rows, int32_t cols, int32_t b_mode, void* stream);
hs_status hs_rowdot(const float* a, const float* b, float* out, int32_t rows, int32_t cols, void* stream);
/* out[r] = KL(p[r] || q[r]) on probabilities clamped to [eps, 1] (reference mibf_net/attention.py:25-28).  With w (the
   incoming gradient per row) dp / dq receive the gradients; out may then be NULL. */
hs_status hs_kl_rows(const float* p, const float* q, float* out, const float* w, float* dp, float* dq, int32_t rows,
                     int32_t cols, float eps, void* stream);
/* out[0] = sum a[i]*b[i] (b NULL: sum a). */
hs_status hs_dot(const float* a, const float* b, float* out, int64_t n, void* stream);
hs_status hs_sigmoid_fwd(const float* x, float* out, int64_t n, void* stream);
hs_status hs_sigmoid_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream);
/* ent[r] = -sum softmax(z)*log(softmax(z)+1e-8); with g/dlogits set it also writes the backward
   (reference model.py:276-278). */
hs_status hs_softmax_entropy(const float* logits, const float* g, float* ent, float* dlogits, int32_t rows, int32_t C,
                             void* stream);
/* MIBF MP-Loss: 0.3 CE(img) + 0.6 CE(txt) + 1.1 mean(exp(symKL)) CE(fused), loss + three logit gradients
   (reference mibf_net/model_resnet.py:76-94). scratch: B floats. */
hs_status hs_mp_loss(const float* image_logits, const float* text_logits, const float* fused_logits, const int64_t* labels,
                     int32_t B, int32_t C, float* loss, float* d_image, float* d_text, float* d_fused, float* scratch,
                     void* stream);
/* FocalLoss(gamma, weight), mean reduction (reference scripts/train.py:46-61). */
hs_status hs_focal_loss(const float* logits, const int64_t* labels, const float* weight, float gamma, int32_t B, int32_t C,
                        float* loss, float* dlogits, void* stream);
/* out = bilinear_resize(x[:, :, y0:y0+ch, x0:x0+cw], (H, W)), align_corners=False, f32 NCHW
   (reference model.py:292-301). */
hs_status hs_center_crop_resize(const float* x, float* out, int32_t N, int32_t Cc, int32_t H, int32_t W, int32_t y0,
                                int32_t x0, int32_t ch, int32_t cw, void* stream);

/* LSTM / GRU cells of the slice-sequence encoder (reference modules/sequence_blocks.py:22-34,58-62 -> torch.nn.LSTM /
   nn.GRU), f32.  gx = x W_ih^T + b_ih and gh = h W_hh^T + b_hh come from hs_linear_fwd (row pitches ldx / ldh); gate order
   as torch (LSTM i,f,g,o; GRU r,z,n).  `act` [B][4H] keeps the activated gates for the backward. */
hs_status hs_lstm_cell_fwd(const float* gx, int32_t ldx, const float* gh, int32_t ldh, const float* c_prev, float* h, float* c,
                           float* act, int32_t B, int32_t H, void* stream);
/* dgates [B][4H] is the gradient of (gx + gh); dh / dc may be NULL (= zero). */
hs_status hs_lstm_cell_bwd(const float* dh, const float* dc, const float* act, const float* c_prev, const float* c,
                           float* dgates, float* dc_prev, int32_t B, int32_t H, void* stream);
hs_status hs_gru_cell_fwd(const float* gx, int32_t ldx, const float* gh, int32_t ldh, const float* h_prev, float* h,
                          float* act, int32_t B, int32_t H, void* stream);
/* dgx / dgh [B][3H]: gradients of gx and gh; dh_prev: the direct z*dh path only (the W_hh path comes from dgh). */
hs_status hs_gru_cell_bwd(const float* dh, const float* act, const float* h_prev, float* dgx, float* dgh, float* dh_prev,
                          int32_t B, int32_t H, void* stream);

/* KANLinear.regularization_loss (reference ConNexT/models/block/kan1.py:216-236) on spline_weight viewed as
   [rows = out*in][coeffs]: loss = ra * sum_j l_j + re * entropy(l / sum l), l_j = mean_c |w[j][c]|; dw (optional) is
   d loss / d w.  Single workgroup, deterministic. */
hs_status hs_kan_regularization(const float* w, int64_t rows, int32_t coeffs, float reg_activation, float reg_entropy,
                                float* loss, float* dw, void* stream);

/* SupConLoss(temperature) on (B, D) features, mean over anchors, loss + d/d features
   (reference scripts/train.py:23-44).  ws: hs_supcon_ws_bytes(B, D). */
hs_status hs_supcon_loss(const float* feat, const int64_t* labels, int32_t B, int32_t D, float temperature, float* loss,
                         float* dfeat, float* ws, void* stream);
int64_t hs_supcon_ws_bytes(int32_t B, int32_t D);

/* ------------------------------------------------------------------------------------------- */
/* ConvNeXt operators on NHWC activations (reference ConNexT/models/ourmodel.py:43,78 ->          */
/* transformers ConvNextModel: ConvNextLayer.dwconv / layer_scale_parameter, ConvNextEmbeddings   */
/* .patch_embeddings, ConvNextStage.downsampling_layer).                                          */
/* ------------------------------------------------------------------------------------------- */
/* depthwise ksize x ksize convolution (ksize in {3,5,7}), stride 1, padding ksize/2, C % 4 == 0.
   w: f32 [C][ksize*ksize] (memory of a torch groups=C weight (C,1,k,k)); bias f32 [C] or NULL. */
int64_t hs_dwconv_ws_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t ksize);
hs_status hs_dwconv_fwd(int32_t dtype, const void* x, const float* w, const float* bias, void* y, int32_t N, int32_t H,
                        int32_t W, int32_t C, int32_t ksize, void* ws, int64_t ws_bytes, void* stream);
/* dx (optional) = correlation of dy with the flipped filter; dw [C][k*k] and db [C] (optional) are reduced
   deterministically through ws. */
hs_status hs_dwconv_bwd(int32_t dtype, const void* x, const float* w, const void* dy, void* dx, float* dw, float* db,
                        int32_t N, int32_t H, int32_t W, int32_t C, int32_t ksize, void* ws, int64_t ws_bytes, void* stream);
/* out[m][c] = res[m][c] + gamma[c] * rowscale[m / rows_per_sample] * u[m][c]; rowscale (f32, one value per sample:
   stochastic depth keep/(1-p)) may be NULL. */
hs_status hs_layerscale_fwd(int32_t dtype, const void* u, const float* gamma, const float* rowscale, int32_t rows_per_sample,
                            const void* res, void* out, int64_t M, int32_t C, void* stream);
/* du (optional) = gamma * rowscale * dy ; dgamma[c] = sum_m rowscale * dy * u. */
hs_status hs_layerscale_bwd(int32_t dtype, const void* dy, const void* u, const float* gamma, const float* rowscale,
                            int32_t rows_per_sample, void* du, float* dgamma, void* ws, int64_t ws_bytes, int64_t M, int32_t C,
                            void* stream);
int64_t hs_layerscale_ws_bytes(int64_t M, int32_t C);
/* space-to-depth for stride == kernel convolutions: out[(n,p,q)][(r*k+s)*C + c] = x[n][p*k+r][q*k+s][c], P = H/k,
   Q = W/k (floor), row pitch ldo >= k*k*C (extra columns zeroed); _bwd is the inverse scatter into dx (N,H,W,C). */
hs_status hs_patchify_fwd(int32_t dtype, const void* x, void* out, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k,
                          int32_t ldo, void* stream);
hs_status hs_patchify_bwd(int32_t dtype, const void* dpatch, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k,
                          int32_t ldp, void* stream);

/* ------------------------------------------------------------------------------------------- */
/* KAN layer / MoE gating pieces (reference ConNexT/models/block/kan1.py:77-165, moe.py:171-291).  */
/* KANLinear(x) = [act(x) | b_splines(x)] @ [base_weight | spline_weight*spline_scaler]^T: the      */
/* feature / weight packing below, the contraction on hs_gemm.                                     */
/* ------------------------------------------------------------------------------------------- */
/* feat[b] = [act(x[b,:]) | bases(x[b,0]) .. bases(x[b,in-1])], row length in*(1 + grid_size + order);
   grid: (in, grid_size + 2*order + 1) knots.  base_act = the layer's `base_activation` (kan1.py:17,35): 0 SiLU (the
   reference default), 1 GELU (erf), 2 ReLU, 3 identity -- the KAN classifier head (reference modules/heads.py:108-140,
   `act_mode`) selects it. */
hs_status hs_kan_features_fwd(const float* x, const float* grid, float* feat, int64_t B, int32_t in_f, int32_t grid_size,
                              int32_t order, int32_t base_act, void* stream);
hs_status hs_kan_features_bwd(const float* x, const float* grid, const float* dfeat, float* dx, int64_t B, int32_t in_f,
                              int32_t grid_size, int32_t order, int32_t base_act, void* stream);
/* wcat[o] = [base_w[o,:] | spline_w[o,i,:]*scaler[o,i] ...]  (scaler may be NULL), nb = grid_size + order. */
hs_status hs_kan_pack_weight(const float* base_w, const float* spline_w, const float* scaler, float* wcat, int32_t out_f,
                             int32_t in_f, int32_t nb, void* stream);
hs_status hs_kan_unpack_wgrad(const float* dwcat, const float* spline_w, const float* scaler, float* d_base, float* d_spline,
                              float* d_scaler, int32_t out_f, int32_t in_f, int32_t nb, void* stream);
/* noisy top-k gating on clean = x@w_gate (and raw_noise = x@w_noise when noisy): gates (B,E), load-balancing loss
   coef*(cv^2(importance)+cv^2(load)), plus everything the backward needs (p, top: (B,17) int32, z, sigma,
   loadrow, d_imp[E], d_load[E]).  E <= 16.
   noise: optional (B,E) standard-normal draw to use instead of the counter RNG (NULL = draw from `seed`): the reference's
   torch.randn_like(clean_logits), moe.py:247, recorded by a caller that wants its gating reproduced exactly. */
hs_status hs_moe_gate_fwd(const float* clean, const float* raw_noise, const float* noise, int32_t B, int32_t E, int32_t k, int32_t noisy,
                          float noise_eps, uint64_t seed, float coef, float* gates, float* p, int32_t* top, float* z,
                          float* sigma, float* loadrow, float* loss, float* d_imp, float* d_load, void* stream);
hs_status hs_moe_gate_bwd(const float* clean, const float* raw_noise, const float* p, const int32_t* top, const float* z,
                          const float* sigma, const float* dgates, const float* g_loss, const float* d_imp,
                          const float* d_load, int32_t B, int32_t E, int32_t k, int32_t noisy, float* d_clean, float* d_raw,
                          void* stream);
/* y[b,:] = sum_e gates[b,e]*outs[e][b,:] (dense form of SparseDispatcher.combine, moe.py:86-103) and its backward. */
hs_status hs_moe_combine_fwd(const float* gates, const float* const* outs, float* y, int32_t B, int32_t E, int32_t O,
                             void* stream);
hs_status hs_moe_combine_bwd(const float* gates, const float* const* outs, const float* dy, float* const* douts,
                             float* dgates, int32_t B, int32_t E, int32_t O, void* stream);

/* Streaming 1x1 convolution of the image tower (csrc/pw_stream.hip; reference: torchvision Bottleneck conv1 / conv3 / downsample
   + train-mode BatchNorm, encoder.py:35-58, mibf_net/model_resnet.py:15): y[M][N] = x[M][K] . w[N][K]^T in bf16 with f32
   accumulation, persistent workgroups over 64-row blocks.  stats (optional): hs_pointwise_stat_rows(M, N, K) rows of
   (count, mean, M2) per output channel -- the partials hs_bn_params.partial_rows consumes.  Covered shapes: K % 64 == 0,
   64 <= K <= 256, N % 128 == 0 or N == 64.  A measured alternative to hs_gemm for these layers (not faster end to end: see
   the kernel's header); the composite executors use it only under HAMSPINE_PW_STREAM=1. */
int32_t hs_pointwise_stat_rows(int64_t M, int32_t N, int32_t K);
hs_status hs_pointwise_fwd(const void* x, int64_t M, int32_t K, int32_t ldx, const void* w, int32_t N, void* y, int32_t ldy,
                           float* stats, void* stream);

/* Sparse dispatch (SparseDispatcher, moe.py:48-112: expert e sees only the rows with gates[b][e] > 0, i.e. k/E of the work).
   hs_moe_dispatch_index: idx[e*B + 0 .. count[e]) = those rows, ascending.  hs_rows_gather: dst[i] = src[idx[i]] (rows of D
   floats).  hs_rows_scatter_add: dst[idx[i]] += scale[idx[i]*ld_scale + col] * src[i] (scale may be NULL; idx unique, so the
   update is a plain read-modify-write and deterministic).  hs_rows_scatter_add_bwd: the backward of the weighted scatter for
   one expert column: dsrc[i] = gates[idx[i]][col] * dy[idx[i]], dgates[idx[i]][col] = <dy[idx[i]], src[i]>. */
hs_status hs_moe_dispatch_index(const float* gates, int32_t B, int32_t E, int32_t* idx, int32_t* count, void* stream);
hs_status hs_rows_gather(const float* src, const int32_t* idx, float* dst, int32_t n, int32_t D, void* stream);
hs_status hs_rows_scatter_add(float* dst, const int32_t* idx, const float* scale, int32_t ld_scale, int32_t col, const float* src,
                              int32_t n, int32_t D, void* stream);
hs_status hs_rows_scatter_add_bwd(const float* dy, const int32_t* idx, const float* gates, int32_t E, int32_t col, const float* src,
                                  float* dsrc, float* dgates, int32_t n, int32_t D, void* stream);

/* ------------------------------------------------------------------------------------------- */
/* Composite executors: one call = one nn.Module forward (or backward) of the reference's module  */
/* tree, launched from C++ so the Python host issues O(10) calls per step instead of O(1000).     */
/* Every composite has a *_query that returns the bytes of `saved` (activations kept for the      */
/* backward, also used as forward temporaries) and `ws` (scratch) it needs for a given descriptor. */
/* The caller allocates both; `saved` must be passed unchanged to the matching backward.          */
/* ------------------------------------------------------------------------------------------- */

/* multi-head attention core on projected q/k/v (materialised scores):
     S = scale * Q K^T (+ key mask) ; P = softmax(S) ; O = dropout(P) V
   q/k/v/o element (b, t, h, j) lives at  base + b*bs + t*ld + h*hd + j.
   Replaces the attention inside torch.nn.MultiheadAttention (reference modules/fusion_blocks.py:
   18-31,107-112; modules/heads.py:86) and transformers BertSelfAttention (encoder.py:131). */
typedef struct hs_attn_desc {
    int32_t dtype;
    int32_t B, H, Lq, Lk, hd;      /* batch, heads, query len, key len, head dim                  */
    int64_t q_bs, k_bs, v_bs, o_bs;/* batch strides (elements)                                    */
    int32_t q_ld, k_ld, v_ld, o_ld;/* token strides (elements)                                    */
    float scale;
    float dropout_p;
    uint64_t seed;
    const int64_t* key_mask;       /* [B][Lk], 0 = masked key; NULL = no mask                     */
} hs_attn_desc;
hs_status hs_attention_query(const hs_attn_desc* d, int64_t* saved_bytes, int64_t* ws_bytes);
hs_status hs_attention_fwd(const hs_attn_desc* d, const void* q, const void* k, const void* v, void* o, void* saved,
                           int64_t saved_bytes, void* ws, int64_t ws_bytes, void* stream);
/* dq/dk/dv use the q/k/v strides of the descriptor; do uses the o strides. */
hs_status hs_attention_bwd(const hs_attn_desc* d, const void* q, const void* k, const void* v, const void* d_o, void* dq,
                           void* dk, void* dv, void* saved, int64_t saved_bytes, void* ws, int64_t ws_bytes,
                           void* stream);

/* weight-gradient side stream inside the composites below: 1 on (default; env HAMSPINE_OVERLAP=0 turns it off),
   0 off = every kernel of a composite runs on the caller's stream (used to time kernels in isolation). */
void hs_set_overlap(int32_t on);
/* Gradient milestones for bucketed data-parallel exchange (reference mibf_net/train_resnet.py:134: DDP starts a bucket's
   all-reduce when its last gradient is ready).  The NEXT hs_resnet_bwd / hs_bert_bwd on the calling thread records
   events[i] (hipEvent_t) on its stream right after the kernels that finish the parameter gradient at grad_ptrs[i] (one of the
   dw / db / dgamma / dbeta pointers of its descriptor) have been enqueued; entries it does not recognise are recorded when
   it returns.  n <= 64; n = 0 clears the list. */
hs_status hs_grad_milestones(int32_t n, const void* const* grad_ptrs, void* const* events);
/* 1 when the library was built with -DHS_MEASURE (the HAMSPINE_KNOCKOUT work-skipping switch of tools/knockout.sh is compiled
   in; results may be garbage when that variable is set), 0 for the release build.  bench.py refuses a measurement build. */
int32_t hs_measure_build(void);
/* BertLayer backward (bf16): weight gradients as K-contiguous GEMMs on transposed copies of dY and X (1, default) or straight
   from the row-major operands (0); measurement / A-B testing switch. */
void hs_set_wgrad_nt(int32_t on);

/* conv + BatchNorm pair of a residual block. `w` is the f32 filter stored KRSC (channels_last). */
typedef struct hs_conv_bn {
    int32_t Cin, Cout, R, stride, pad;
    const float* w;
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    float* dw;        /* backward outputs (f32); NULL = frozen parameter, wgrad skipped            */
    float* dgamma;
    float* dbeta;
} hs_conv_bn;

/* torchvision BasicBlock (n_main = 2: 3x3,3x3) / Bottleneck (n_main = 3: 1x1,3x3[stride],1x1) with an
   optional 1x1 downsample branch: y = relu(main(x) + (ds ? bn(conv(x)) : x)).
   Replaces torchvision.models.resnet.{BasicBlock,Bottleneck}.forward under reference
   encoder.py:69-72 and mibf_net/model_resnet.py:15. */
typedef struct hs_resblock_desc {
    int32_t dtype;
    int32_t N, H, W;              /* input image; channels = main[0].Cin                           */
    int32_t training;             /* BatchNorm mode                                                */
    float eps, momentum;
    int32_t n_main;
    hs_conv_bn main[3];
    int32_t has_ds;
    hs_conv_bn ds;
    int32_t inference;            /* 1 (only with training = 0): no backward follows -- every BatchNorm is folded into
                                     its convolution's epilogue (scale, shift, ReLU, identity add) and only the weight
                                     casts use `saved`                                                 */
} hs_resblock_desc;
hs_status hs_resblock_query(const hs_resblock_desc* d, int64_t* saved_bytes, int64_t* ws_bytes);
hs_status hs_resblock_fwd(const hs_resblock_desc* d, const void* x, void* y, void* saved, int64_t saved_bytes, void* ws,
                          int64_t ws_bytes, void* stream);
/* dx may be NULL (no input gradient wanted). */
hs_status hs_resblock_bwd(const hs_resblock_desc* d, const void* x, const void* y, const void* dy, void* dx, void* saved,
                          int64_t saved_bytes, void* ws, int64_t ws_bytes, void* stream);

/* ResNet stem: conv7x7/2 (3->64) + BN + ReLU + maxpool3x3/2 on an f32 NCHW image.
   Replaces reference encoder.py:63-68 (self.stem) / torchvision ResNet conv1..maxpool. */
typedef struct hs_stem_desc {
    int32_t dtype;
    int32_t N, H, W;              /* input image (3 channels, f32 NCHW)                            */
    int32_t training;
    float eps, momentum;
    hs_conv_bn cb;                /* Cin = 3, Cout = 64, R = 7, stride = 2, pad = 3                */
    int32_t inference;            /* as hs_resblock_desc.inference                                 */
} hs_stem_desc;
hs_status hs_stem_query(const hs_stem_desc* d, int64_t* saved_bytes, int64_t* ws_bytes);
hs_status hs_stem_fwd(const hs_stem_desc* d, const float* image, void* y, void* saved, int64_t saved_bytes, void* ws,
                      int64_t ws_bytes, void* stream);
hs_status hs_stem_bwd(const hs_stem_desc* d, const void* y, const void* dy, void* saved, int64_t saved_bytes, void* ws,
                      int64_t ws_bytes, void* stream);

/* Linear layer parameters (f32) and their gradient outputs. */
typedef struct hs_linear {
    int32_t in_f, out_f;
    const float* w;               /* [out][in]                                                     */
    const float* b;               /* [out] or NULL                                                 */
    float* dw;                    /* NULL = frozen                                                 */
    float* db;
} hs_linear;
typedef struct hs_norm {
    const float* gamma;
    const float* beta;
    float* dgamma;
    float* dbeta;
} hs_norm;

/* transformers BertLayer (post-LN, erf-GELU): self-attention + output dense/LN + FFN + output LN.
   Replaces transformers.models.bert.modeling_bert.BertLayer.forward under reference
   encoder.py:131, mibf_net/bert.py:12. */
typedef struct hs_bert_layer_desc {
    int32_t dtype;
    int32_t B, L, hidden, heads, inter;
    float ln_eps;
    float hidden_dropout, attn_dropout;   /* 0 in eval mode                                         */
    uint64_t seed;
    const int64_t* attention_mask;        /* [B][L], 1 = attend                                     */
    hs_linear q, k, v, ao;                /* attention.self.{query,key,value}, attention.output.dense */
    hs_norm ln1;                          /* attention.output.LayerNorm                             */
    hs_linear inter_l, out_l;             /* intermediate.dense, output.dense                       */
    hs_norm ln2;                          /* output.LayerNorm                                       */
} hs_bert_layer_desc;
hs_status hs_bert_layer_query(const hs_bert_layer_desc* d, int64_t* saved_bytes, int64_t* ws_bytes);
hs_status hs_bert_layer_fwd(const hs_bert_layer_desc* d, const void* x, void* y, void* saved, int64_t saved_bytes,
                            void* ws, int64_t ws_bytes, void* stream);
hs_status hs_bert_layer_bwd(const hs_bert_layer_desc* d, const void* x, const void* dy, void* dx, void* saved,
                            int64_t saved_bytes, void* ws, int64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------- */
/* Whole-tower executors: one call runs a complete tower forward or backward (the stem and every residual block /
   the embeddings and every BertLayer), with the same kernels and per-composite layout code as the per-block entry
   points above.  They exist for the host side: the unchanged reference loop calls `model(...)` once per step
   (scripts/train.py:373-385, mibf_net/train_resnet.py:30-32), so the product can hand a whole tower to ONE C call
   instead of ~70 (descriptors are built once per module and reused; nothing is re-planned or re-allocated per block).
   Replaces torchvision ResNet.forward up to layer4 (reference encoder.py:88-100, mibf_net/model_resnet.py:15) and
   transformers BertModel.forward (encoder.py:130-134, mibf_net/bert.py:11-13).                                       */
/* ------------------------------------------------------------------------------------------- */
/* sizeof of ABI struct number `which` as the library was compiled: 0 hs_conv_geom, 1 hs_gemm_params, 2 hs_bn_params,
   3 hs_bn_bwd_params, 4 hs_attn_desc, 5 hs_conv_bn, 6 hs_resblock_desc, 7 hs_stem_desc, 8 hs_linear, 9 hs_norm,
   10 hs_bert_layer_desc, 11 hs_resnet_desc, 12 hs_resnet_plan, 13 hs_bert_desc; -1 for an unknown number.  A binding
   (the ctypes mirror in hamspine/_lib.py, a reference-side stub) checks its own declarations against it. */
int64_t hs_abi_sizeof(int32_t which);

#define HS_RESNET_MAX_BLOCKS 24
#define HS_RESNET_MAX_TAPS 4
typedef struct hs_resnet_desc {
    hs_stem_desc stem;
    int32_t n_blocks;
    hs_resblock_desc blocks[HS_RESNET_MAX_BLOCKS];   /* layer1..layer4 flattened; block i takes block i-1's output    */
    int32_t n_taps;                                  /* block outputs handed back (encoder.py:94-100: layer2/3/4)     */
    int32_t tap_block[HS_RESNET_MAX_TAPS];           /* ascending block indices; the last one is n_blocks - 1          */
} hs_resnet_desc;
typedef struct hs_resnet_plan {
    int64_t saved_bytes, ws_bytes;
    int64_t tap_offset[HS_RESNET_MAX_TAPS];          /* byte offset of tap t inside `saved` (NHWC, compute dtype)      */
    int32_t tap_C[HS_RESNET_MAX_TAPS], tap_H[HS_RESNET_MAX_TAPS], tap_W[HS_RESNET_MAX_TAPS];
} hs_resnet_plan;
hs_status hs_resnet_query(const hs_resnet_desc* d, hs_resnet_plan* plan);
/* image: f32 NCHW.  The tap activations are left inside `saved` at plan.tap_offset. */
hs_status hs_resnet_fwd(const hs_resnet_desc* d, const float* image, void* saved, int64_t saved_bytes, void* ws,
                        int64_t ws_bytes, void* stream);
/* dy_taps[t]: gradient of tap t (NHWC, compute dtype) or NULL; the last tap's gradient is required.  Parameter
   gradients go to the dw / dgamma / dbeta pointers of the descriptor (NULL = frozen). */
hs_status hs_resnet_bwd(const hs_resnet_desc* d, const void* const* dy_taps, void* saved, int64_t saved_bytes, void* ws,
                        int64_t ws_bytes, void* stream);

/* Test / analysis aid (tests/test_decisions_gpu.py): byte offsets, inside the `saved` arena of hs_resnet_fwd, of every
   ReLU output of the tower and of the stem's max-pool arg-max, in forward order: [0] stem BN+ReLU output ([N][P][Q][C],
   compute dtype), [1] max-pool arg-max taps (uint8 [N][P2][Q2][C], tap = 3*dh + dw inside the 3x3 window), then per block
   its inner stage outputs (n_main - 1 of them) and its output.  Returns the number of entries (-1: bad descriptor or
   max_entries too small). */
int32_t hs_resnet_debug_offsets(const hs_resnet_desc* d, int64_t* out, int32_t max_entries);

#define HS_BERT_MAX_LAYERS 24
typedef struct hs_bert_desc {
    int32_t dtype;
    int32_t B, L, hidden, vocab, max_pos, n_types;
    int32_t pad_id;                       /* nn.Embedding padding_idx of the word table (-1: none)                    */
    float ln_eps, embed_dropout;          /* BertEmbeddings LayerNorm eps / dropout (0 in eval mode)                   */
    uint64_t seed;                        /* dropout seed of this call: embeddings use it, layer i uses seed+16(i+1)   */
    const float* word;                    /* [vocab][hidden]                                                           */
    const float* pos;                     /* [max_pos][hidden]                                                         */
    const float* type0;                   /* [n_types][hidden]; row 0 is used (token_type_ids = 0)                     */
    const float* gamma;
    const float* beta;
    float* dword;                         /* gradient outputs, NULL = frozen                                           */
    float* dpos;
    float* dtype0;
    float* dgamma;
    float* dbeta;
    int32_t n_layers;
    hs_bert_layer_desc layers[HS_BERT_MAX_LAYERS];   /* attention_mask / seed fields are overridden per call           */
} hs_bert_desc;
/* out_offset: byte offset of last_hidden_state ([B][L][hidden], compute dtype) inside `saved`. */
hs_status hs_bert_query(const hs_bert_desc* d, int64_t* saved_bytes, int64_t* ws_bytes, int64_t* out_offset);
hs_status hs_bert_fwd(const hs_bert_desc* d, const int64_t* ids, const int64_t* mask, void* saved, int64_t saved_bytes,
                      void* ws, int64_t ws_bytes, void* stream);
hs_status hs_bert_bwd(const hs_bert_desc* d, const int64_t* ids, const int64_t* mask, const void* dy, void* saved,
                      int64_t saved_bytes, void* ws, int64_t ws_bytes, void* stream);

/* y = act(x W^T + b) with optional dropout / residual; generic Linear fwd + bwd on [M][in] rows.
   Replaces torch.nn.Linear call sites of the reference (encoder.py:80-86, modules/ *.py, model.py:195-200,
   mibf_net/attention.py:40-45, mibf_net/model_resnet.py:16,22-34). */
hs_status hs_linear_fwd(int32_t dtype, const void* x, int64_t M, int32_t ldx, const hs_linear* lin, const void* w_lp,
                        void* y, int32_t ldy, int32_t out_dtype, int32_t act, void* preact, const void* residual,
                        int32_t ldr, float dropout_p, uint64_t seed, void* stream);
/* dx = (dy W) [* act'(preact)] ; dw/db per `lin` (skipped when NULL). `dy` must already include the
   activation derivative when act != none (use mul_mode to fold it into this call's dx = ... of the
   PREVIOUS layer instead).  ws: hs_linear_bwd_ws_bytes. */
hs_status hs_linear_bwd(int32_t dtype, const void* x, int64_t M, int32_t ldx, const hs_linear* lin, const void* w_lp,
                        const void* dy, int32_t ldy, void* dx, int32_t lddx, int32_t dx_dtype, int32_t mul_mode,
                        const void* mul_src, int32_t ldm, const void* dx_residual, void* ws, int64_t ws_bytes,
                        void* stream);
int64_t hs_linear_bwd_ws_bytes(int64_t M, int32_t in_f, int32_t out_f, int32_t dtype);

/* ------------------------------------------------------------------------------------------- */
/* callers either side of the path: inference helpers and input staging     */
/* ------------------------------------------------------------------------------------------- */
/* Test-time augmentation as ONE batch (replaces the per-variant model calls of reference scripts/predict.py:33-42,
   63-70): out[v] = op_v(x) over `planes` = B*C (or B*T*C) image planes of H x W f32.  ops[v]: 0 identity, 1 flip(-1)
   ("hflip"), 2 flip(-2) ("vflip"), 3 torch.rot90(k=1, dims=(-2,-1)) (square planes only).  1 <= V <= 8. */
hs_status hs_tta_expand(const float* x, float* out, int64_t planes, int32_t H, int32_t W, const int32_t* ops, int32_t V,
                        void* stream);
/* out[i] = mean over v of x[v*n + i]: torch.stack(logits_list, 0).mean(0) of scripts/predict.py:70. */
hs_status hs_group_mean(const float* x, float* out, int32_t V, int64_t n, void* stream);
/* out[v*n + i] = x[i]: tiles text tokens / ids / masks over the TTA variants (elements of 2, 4 or 8 bytes). */
hs_status hs_repeat(int32_t elem_bytes, const void* x, void* out, int64_t n, int32_t V, void* stream);
/* Grad-CAM map per image from one stage's activation and gradient, both [B][HW][C] (NHWC memory):
   w = mean_hw(grad); cam = relu(sum_c w[c] * act[:, c]); cam /= max(cam) if max(cam) > 0.
   Replaces the numpy loop of reference analysis_tools.py:78-93 (before its cv2.resize). */
hs_status hs_gradcam(int32_t dtype, const void* act, const void* grad, float* cam, int32_t B, int32_t C, int32_t HW,
                     void* stream);
/* Input staging: decoded u8 images [B][H][W][3] -> f32 [B][3][H][W], (v/255 - mean[c]) / std[c] in IEEE f32
   (torchvision ToTensor + Normalize of the reference's data pipeline, data_loader.py transforms). */
hs_status hs_stage_images_u8(const uint8_t* src, float* out, int32_t B, int32_t H, int32_t W, const float* mean,
                             const float* std, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HAMSPINE_H */
