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
    /* split-K (batch must be 1): partial sums go to `splitk_ws` (f32, split*M*N) and are
       reduced by a second kernel that applies the epilogue. 0/1 = off. */
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
} hs_gemm_params;

hs_status hs_gemm(const hs_gemm_params* p, void* stream);
/* workspace (bytes) hs_gemm needs in p->splitk_ws for the given p (0 when split_k <= 1). */
int64_t hs_gemm_splitk_ws_bytes(const hs_gemm_params* p);
/* heuristic split-K factor for a (M,N,K) problem so that the grid fills 256 CUs. */
int32_t hs_gemm_suggest_split(int32_t M, int32_t N, int32_t K, int32_t dtype);

#ifdef __cplusplus
}
#endif
#endif /* HAMSPINE_H */
