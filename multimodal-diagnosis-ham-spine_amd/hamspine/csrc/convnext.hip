// ConvNeXt-specific operators on NHWC activations (gfx950): depthwise KxK convolution (forward, data and filter
// gradients), layer-scale + residual, and the space-to-depth gather that turns the stride==kernel "patchify"
// convolutions into plain GEMMs.  Everything here is HBM/L2 bound vector work; the pointwise (1x1) convolutions and
// the patchify matmuls run on the MFMA GEMM core.
//
// Replaces (reference ConNexT/models/ourmodel.py:43,78 -> transformers ConvNextModel):
//   ConvNextLayer.dwconv            nn.Conv2d(dim, dim, 7, padding=3, groups=dim)        -> hs_dwconv_fwd / hs_dwconv_bwd
//   layer_scale_parameter * x + res (+ optional per-sample stochastic-depth scale)       -> hs_layerscale_fwd / _bwd
//   ConvNextEmbeddings.patch_embeddings (4x4/4) and downsampling_layer[1] (2x2/2)        -> hs_patchify_fwd / _bwd + GEMM
#include <algorithm>
#include "hs_common.h"

namespace hs {

#define CN_DISPATCH_T(dtype, fn, ...) \
    ((dtype) == HS_BF16 ? fn<bf16_t>(__VA_ARGS__) : (dtype) == HS_F32 ? fn<float>(__VA_ARGS__) : (set_error("bad dtype %d", (int)(dtype)), HS_ERR_ARG))

static inline int cn_grid(long long total, int block = 256) { return (int)std::min<long long>((total + block - 1) / block, 1 << 30); }

// 4 consecutive channels <-> 4 floats
__device__ __forceinline__ void load4(const float* p, float* f) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
}
__device__ __forceinline__ void load4(const bf16_t* p, float* f) {
    const u32x2 v = *reinterpret_cast<const u32x2*>(p);
    f[0] = __uint_as_float(v[0] << 16); f[1] = __uint_as_float(v[0] & 0xffff0000u);
    f[2] = __uint_as_float(v[1] << 16); f[3] = __uint_as_float(v[1] & 0xffff0000u);
}
__device__ __forceinline__ void store4(float* p, const float* f) {
    f32x4 v = {f[0], f[1], f[2], f[3]};
    *reinterpret_cast<f32x4*>(p) = v;
}
__device__ __forceinline__ void store4(bf16_t* p, const float* f) {
    bf16x4 v = {(bf16_t)f[0], (bf16_t)f[1], (bf16_t)f[2], (bf16_t)f[3]};
    *reinterpret_cast<bf16x4*>(p) = v;
}

// ------------------------------------------------------------------------------------------------------------
// filter [C][KS*KS] (torch layout of a groups=C conv weight) -> [KS*KS][C], optionally flipped (data gradient)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dw_pack_filter_kernel(const float* __restrict__ w, float* __restrict__ wt, int C,
                                                             int taps, int flip) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C * taps) return;
    const int t = i / C, c = i - t * C;
    wt[i] = w[c * taps + (flip ? taps - 1 - t : t)];
}

// ------------------------------------------------------------------------------------------------------------
// y[n,h,w,c] = b[c] + sum_{r,s} x[n,h+r-pad,w+s-pad,c] * wt[r*KS+s][c]
// One thread: 4 channels x TW consecutive output columns.  Each input vector is loaded once per filter row and
// feeds up to KS outputs from registers, so L1 traffic is ~(TW+KS-1)/TW loads per output per row instead of KS.
// ------------------------------------------------------------------------------------------------------------
// (A branch-free variant of this kernel -- padding through out-of-range buffer offsets, as in dwconv_wgrad_kernel below --
// measured 54.7 vs 35.6 us on the 14 x 14 x 512 stage: here the skipped taps are real work saved and the loads of a row
// already issue together, so the predicated form stays.)
template <typename T, int KS, int TW>
__global__ __launch_bounds__(256) void dwconv_kernel(const T* __restrict__ x, const float* __restrict__ wt,
                                                     const float* __restrict__ bias, T* __restrict__ y, int N, int H,
                                                     int W, int C, int WT, long long total) {
    constexpr int PAD = KS / 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int C4 = C >> 2;
    const int c = (int)(idx % C4) * 4;
    long long rest = idx / C4;
    const int wt_i = (int)(rest % WT);
    rest /= WT;
    const int h = (int)(rest % H);
    const int n = (int)(rest / H);
    const int w0 = wt_i * TW;
    float acc[TW][4];
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) load4(bias + c, bv);
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[t][k] = bv[k];
#pragma unroll
    for (int r = 0; r < KS; ++r) {
        const int ih = h + r - PAD;
        if (ih < 0 || ih >= H) continue;
        float wv[KS][4];
#pragma unroll
        for (int s = 0; s < KS; ++s) load4(wt + (size_t)(r * KS + s) * C + c, wv[s]);
        const T* xr = x + ((size_t)(n * H + ih) * W) * C + c;
#pragma unroll
        for (int j = 0; j < TW + KS - 1; ++j) {
            const int iw = w0 + j - PAD;
            if (iw < 0 || iw >= W) continue;
            float xv[4];
            load4(xr + (size_t)iw * C, xv);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int t = j - s;
                if (t >= 0 && t < TW) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[t][k] = fmaf(xv[k], wv[s][k], acc[t][k]);
                }
            }
        }
    }
    T* yr = y + ((size_t)(n * H + h) * W) * C + c;
#pragma unroll
    for (int t = 0; t < TW; ++t)
        if (w0 + t < W) store4(yr + (size_t)(w0 + t) * C, acc[t]);
}

// ------------------------------------------------------------------------------------------------------------
// filter / bias gradient: dw[c][r][s] = sum_{n,h,w} dy[n,h,w,c] * x[n,h+r-pad,w+s-pad,c];  db[c] = sum dy.
// One wave per (128-channel group, chunk of image rows); a lane owns TWO adjacent channels (4-byte bf16 pair loads).
// A row is walked in segments of SW columns: the dy segment and, per filter row, the SW + KS - 1 inputs under it are
// fetched by INDEPENDENT loads into registers before any of them is used.  (The first version slid a KS-wide window along
// w with one dy load per column consumed in the same iteration: every iteration waited a full memory round trip, 343 of
// them in a row for a 14 x 14 map -- 190 us per ConvNeXt stage-3 block against a 26 MB / 0.3 GFMA problem.)  Chunk
// partials go to ws, a second kernel sums them in a fixed order (deterministic).
// ------------------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void load2(const float* p, float* f) {
    const f32x2 v = *(const f32x2*)p;
    f[0] = v[0];
    f[1] = v[1];
}
__device__ __forceinline__ void load2(const bf16_t* p, float* f) {
    const unsigned v = *(const unsigned*)p;
    f[0] = __uint_as_float(v << 16);
    f[1] = __uint_as_float(v & 0xffff0000u);
}
// two adjacent channels through a buffer descriptor: out-of-range offsets (kOOB) read as zeros, so image borders and
// row tails need no branch and the loads of a batch issue back to back
template <typename T>
__device__ __forceinline__ void bload2(__amdgpu_buffer_rsrc_t rs, unsigned off, float* f) {
    if constexpr (sizeof(T) == 2) {
        const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0);
        f[0] = __uint_as_float(v << 16);
        f[1] = __uint_as_float(v & 0xffff0000u);
    } else {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0);
        f[0] = __uint_as_float(v[0]);
        f[1] = __uint_as_float(v[1]);
    }
}
template <typename T, int KS>
__global__ __launch_bounds__(64) void dwconv_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                          float* __restrict__ ws, int N, int H, int W, int C) {
    constexpr int PAD = KS / 2;
    constexpr int TAPS = KS * KS;
    constexpr int SW = 16;
    const int c = (blockIdx.x * 64 + threadIdx.x) * 2;
    if (c >= C) return;                    // C % 4 == 0: a live lane owns two live channels
    float acc[KS][KS][2];
    float accb[2] = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < KS; ++r)
#pragma unroll
        for (int s = 0; s < KS; ++s) acc[r][s][0] = acc[r][s][1] = 0.f;
    const int rows = N * H;
    const unsigned long long bytes = (unsigned long long)rows * W * C * sizeof(T);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (unsigned)min(bytes, 0x7fffff00ull));
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(dy, (unsigned)min(bytes, 0x7fffff00ull));
    const unsigned pix = (unsigned)C * sizeof(T);                 // bytes per pixel
    for (int row = blockIdx.y; row < rows; row += gridDim.y) {
        const int n = row / H, h = row - n * H;
        const unsigned gbase = (unsigned)row * W * pix + c * sizeof(T);
        for (int w0 = 0; w0 < W; w0 += SW) {
            float g[SW][2];
#pragma unroll
            for (int j = 0; j < SW; ++j) bload2<T>(rg, w0 + j < W ? gbase + (unsigned)(w0 + j) * pix : kOOB, g[j]);
#pragma unroll
            for (int j = 0; j < SW; ++j) {
                accb[0] += g[j][0];
                accb[1] += g[j][1];
            }
#pragma unroll
            for (int r = 0; r < KS; ++r) {
                const int ih = h + r - PAD;
                const bool rok = ih >= 0 && ih < H;               // (wave-uniform; a dead row reads zeros)
                const unsigned xbase = (unsigned)((n * H + ih) * W) * pix + c * sizeof(T);
                float xv[SW + KS - 1][2];
#pragma unroll
                for (int j = 0; j < SW + KS - 1; ++j) {
                    const int iw = w0 - PAD + j;
                    bload2<T>(rx, (rok && iw >= 0 && iw < W) ? xbase + (unsigned)iw * pix : kOOB, xv[j]);
                }
#pragma unroll
                for (int j = 0; j < SW; ++j)
#pragma unroll
                    for (int s_ = 0; s_ < KS; ++s_) {
                        acc[r][s_][0] = fmaf(g[j][0], xv[j + s_][0], acc[r][s_][0]);
                        acc[r][s_][1] = fmaf(g[j][1], xv[j + s_][1], acc[r][s_][1]);
                    }
            }
        }
    }
    float* o = ws + (size_t)blockIdx.y * (TAPS + 1) * C + c;
#pragma unroll
    for (int r = 0; r < KS; ++r)
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) *(f32x2*)(o + (size_t)(r * KS + s_) * C) = f32x2{acc[r][s_][0], acc[r][s_][1]};
    *(f32x2*)(o + (size_t)TAPS * C) = f32x2{accb[0], accb[1]};
}
__global__ __launch_bounds__(256) void dwconv_wgrad_final_kernel(const float* __restrict__ ws, int chunks, int C, int taps,
                                                                 float* __restrict__ dw, float* __restrict__ db) {
    // 64 consecutive (tap, channel) entries x 4 chunk groups per block, summed in a fixed order
    __shared__ float sh[4][64];
    const int il = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + il;   // i = t*C + c over (taps+1)*C
    const int n = (taps + 1) * C;
    float a0 = 0.f, a1 = 0.f;
    if (i < n) {
        const size_t stride = (size_t)n;
        int k = grp;
        for (; k + 4 < chunks; k += 8) {
            a0 += ws[(size_t)k * stride + i];
            a1 += ws[(size_t)(k + 4) * stride + i];
        }
        if (k < chunks) a0 += ws[(size_t)k * stride + i];
    }
    sh[grp][il] = a0 + a1;
    __syncthreads();
    if (grp == 0 && i < n) {
        const float v = (sh[0][il] + sh[1][il]) + (sh[2][il] + sh[3][il]);
        const int t = i / C, c = i - t * C;
        if (t < taps) dw[(size_t)c * taps + t] = v;
        else if (db) db[c] = v;
    }
}

static inline int dw_chunks(int N, int H, int C) {
    const int groups = (C + 63) / 64;
    int ch = (2048 + groups - 1) / groups;
    return std::max(1, std::min(ch, N * H));
}

template <typename T>
static int dwconv_launch(const T* x, const float* wt, const float* bias, T* y, int N, int H, int W, int C, int ks,
                         hipStream_t s) {
    constexpr int TW = 4;
    const int WT = (W + TW - 1) / TW;
    const long long total = (long long)N * H * WT * (C / 4);
    if (ks == 7)
        hipLaunchKernelGGL((dwconv_kernel<T, 7, TW>), dim3(cn_grid(total)), dim3(256), 0, s, x, wt, bias, y, N, H, W, C, WT, total);
    else if (ks == 5)
        hipLaunchKernelGGL((dwconv_kernel<T, 5, TW>), dim3(cn_grid(total)), dim3(256), 0, s, x, wt, bias, y, N, H, W, C, WT, total);
    else
        hipLaunchKernelGGL((dwconv_kernel<T, 3, TW>), dim3(cn_grid(total)), dim3(256), 0, s, x, wt, bias, y, N, H, W, C, WT, total);
    HS_LAUNCH_CHECK();
    return HS_OK;
}

template <typename T>
static int dwconv_fwd_t(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int C, int ks,
                        float* ws, hipStream_t s) {
    const int taps = ks * ks;
    hipLaunchKernelGGL(dw_pack_filter_kernel, dim3(cn_grid((long long)C * taps)), dim3(256), 0, s, w, ws, C, taps, 0);
    HS_LAUNCH_CHECK();
    return dwconv_launch<T>((const T*)x, ws, bias, (T*)y, N, H, W, C, ks, s);
}

template <typename T>
static int dwconv_bwd_t(const void* x, const float* w, const void* dy, void* dx, float* dw, float* db, int N, int H, int W,
                        int C, int ks, float* ws, hipStream_t s) {
    const int taps = ks * ks;
    if (dx) {
        hipLaunchKernelGGL(dw_pack_filter_kernel, dim3(cn_grid((long long)C * taps)), dim3(256), 0, s, w, ws, C, taps, 1);
        HS_LAUNCH_CHECK();
        HS_PROPAGATE(dwconv_launch<T>((const T*)dy, ws, nullptr, (T*)dx, N, H, W, C, ks, s));
    }
    if (dw) {
        float* part = ws + (size_t)C * taps;
        const int chunks = dw_chunks(N, H, C);
        const dim3 grid((C + 127) / 128, chunks);
        if (ks == 7)
            hipLaunchKernelGGL((dwconv_wgrad_kernel<T, 7>), grid, dim3(64), 0, s, (const T*)x, (const T*)dy, part, N, H, W, C);
        else if (ks == 5)
            hipLaunchKernelGGL((dwconv_wgrad_kernel<T, 5>), grid, dim3(64), 0, s, (const T*)x, (const T*)dy, part, N, H, W, C);
        else
            hipLaunchKernelGGL((dwconv_wgrad_kernel<T, 3>), grid, dim3(64), 0, s, (const T*)x, (const T*)dy, part, N, H, W, C);
        HS_LAUNCH_CHECK();
        hipLaunchKernelGGL(dwconv_wgrad_final_kernel, dim3(cn_grid((long long)(taps + 1) * C, 64)), dim3(256), 0, s, part,
                           chunks, C, taps, dw, db);
        HS_LAUNCH_CHECK();
    }
    return HS_OK;
}

// ------------------------------------------------------------------------------------------------------------
// layer scale + residual:  out[m][c] = res[m][c] + gamma[c] * rs[m / rps] * u[m][c]
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void layerscale_fwd_kernel(const T* __restrict__ u, const float* __restrict__ gamma,
                                                             const float* __restrict__ rowscale, int rps,
                                                             const T* __restrict__ res, T* __restrict__ out, long long M,
                                                             int C) {
    const int C4 = C >> 2;
    const long long total = M * C4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long m = i / C4;
        const int c = (int)(i - m * C4) * 4;
        float uv[4], rv[4], gv[4], o[4];
        load4(u + m * C + c, uv);
        load4(res + m * C + c, rv);
        load4(gamma + c, gv);
        const float rs = rowscale ? rowscale[m / rps] : 1.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = fmaf(gv[k] * rs, uv[k], rv[k]);
        store4(out + m * C + c, o);
    }
}
// du = gamma * rs * dy ; partial dgamma[c] = sum_m rs * dy * u.  A thread owns FOUR adjacent channels (8-byte bf16 accesses;
// the first version owned one: 2-byte accesses and, for C = 128, half a block idle -- 245 us on the 200704 x 128 stage); a
// block covers 256 / (C/4) rows per pass, blockIdx.y strides the row groups, and the rows of a block fold through LDS in a
// fixed order (deterministic).
template <typename T>
__global__ __launch_bounds__(256) void layerscale_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ u,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ rowscale, int rps,
                                                             T* __restrict__ du, float* __restrict__ ws, long long M,
                                                             int C) {
    __shared__ float sh[256][4];
    const int C4 = C >> 2;
    const int ncol = min(256, C4 - (int)blockIdx.x * 256);      // 4-channel columns of this block
    const int rpb = 256 / ncol;                                 // rows per pass
    const int tid = threadIdx.x;
    const int rsub = tid / ncol, col = tid - rsub * ncol;
    const int c = ((int)blockIdx.x * 256 + col) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (rsub < rpb) {
        float gv[4];
        load4(gamma + c, gv);
        for (long long m = (long long)blockIdx.y * rpb + rsub; m < M; m += (long long)gridDim.y * rpb) {
            const float rs = rowscale ? rowscale[m / rps] : 1.f;
            float d[4], uv[4];
            load4(dy + m * C + c, d);
            load4(u + m * C + c, uv);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d[k] *= rs;
                acc[k] = fmaf(d[k], uv[k], acc[k]);
            }
            if (du) {
                float o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = d[k] * gv[k];
                store4(du + m * C + c, o);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) sh[tid][k] = acc[k];
    __syncthreads();
    if (rsub == 0) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < rpb; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] += sh[r * ncol + col][k];
        store4(ws + (size_t)blockIdx.y * C + c, t);
    }
}
__global__ __launch_bounds__(256) void partial_colsum_final_kernel(const float* __restrict__ ws, int gy, int C,
                                                                   float* __restrict__ out) {
    __shared__ float sh[4][64];
    const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float a0 = 0.f, a1 = 0.f;
    if (c < C) {
        int k = grp;
        for (; k + 4 < gy; k += 8) {
            a0 += ws[(size_t)k * C + c];
            a1 += ws[(size_t)(k + 4) * C + c];
        }
        if (k < gy) a0 += ws[(size_t)k * C + c];
    }
    sh[grp][cl] = a0 + a1;
    __syncthreads();
    if (grp == 0 && c < C) out[c] = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
}
static inline int ls_gy(long long M) { return (int)std::min<long long>(std::max<long long>(M / 16, 1), 256); }

template <typename T>
static int layerscale_fwd_t(const void* u, const float* gamma, const float* rowscale, int rps, const void* res, void* out,
                            long long M, int C, hipStream_t s) {
    const long long total = M * (C / 4);
    hipLaunchKernelGGL(layerscale_fwd_kernel<T>, dim3(std::min(cn_grid(total), 256 * 32)), dim3(256), 0, s, (const T*)u, gamma,
                       rowscale, rps, (const T*)res, (T*)out, M, C);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
template <typename T>
static int layerscale_bwd_t(const void* dy, const void* u, const float* gamma, const float* rowscale, int rps, void* du,
                            float* dgamma, float* ws, long long M, int C, hipStream_t s) {
    const int gy = ls_gy(M);
    hipLaunchKernelGGL(layerscale_bwd_kernel<T>, dim3((C / 4 + 255) / 256, gy), dim3(256), 0, s, (const T*)dy, (const T*)u, gamma,
                       rowscale, rps, (T*)du, ws, M, C);
    HS_LAUNCH_CHECK();
    hipLaunchKernelGGL(partial_colsum_final_kernel, dim3((C + 63) / 64), dim3(256), 0, s, ws, gy, C, dgamma);
    HS_LAUNCH_CHECK();
    return HS_OK;
}

// ------------------------------------------------------------------------------------------------------------
// space-to-depth:  out[(n,p,q)][(r*k+s)*C + c] = x[n][p*k+r][q*k+s][c]   (columns k*k*C..ldo-1 are zeroed)
// and its inverse (rows/columns of x beyond P*k / Q*k, which the convolution never reads, get zero gradient).
// V = channels per thread (vector width); V == 1 is the 3-channel image case.
// ------------------------------------------------------------------------------------------------------------
template <typename T, int V>
__global__ __launch_bounds__(256) void patchify_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, int N, int H, int W,
                                                           int C, int k, int P, int Q, int ldo, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cols = ldo / V;
    const long long row = idx / cols;
    const int col = (int)(idx - row * cols) * V;
    const int q = (int)(row % Q);
    const long long t = row / Q;
    const int p = (int)(t % P), n = (int)(t / P);
    T* o = out + row * ldo + col;
    if (col >= k * k * C) {
#pragma unroll
        for (int v = 0; v < V; ++v) o[v] = from_f32<T>(0.f);
        return;
    }
    const int rs = col / C, c = col - rs * C;
    const int r = rs / k, s = rs - r * k;
    const T* src = x + (((size_t)n * H + p * k + r) * W + q * k + s) * C + c;
    if (V == 1) o[0] = src[0];
    else if (V * sizeof(T) == 16) *reinterpret_cast<u32x4*>(o) = *reinterpret_cast<const u32x4*>(src);
    else *reinterpret_cast<u32x2*>(o) = *reinterpret_cast<const u32x2*>(src);
}
template <typename T, int V>
__global__ __launch_bounds__(256) void patchify_bwd_kernel(const T* __restrict__ dp, T* __restrict__ dx, int N, int H, int W,
                                                           int C, int k, int P, int Q, int ldp, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int CV = C / V;
    const int c = (int)(idx % CV) * V;
    long long t = idx / CV;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H), n = (int)(t / H);
    T* o = dx + idx * V;
    const int p = h / k, q = w / k;
    if (p >= P || q >= Q) {
#pragma unroll
        for (int v = 0; v < V; ++v) o[v] = from_f32<T>(0.f);
        return;
    }
    const int r = h - p * k, s = w - q * k;
    const T* src = dp + (((size_t)n * P + p) * Q + q) * ldp + (size_t)(r * k + s) * C + c;
    if (V == 1) o[0] = src[0];
    else if (V * sizeof(T) == 16) *reinterpret_cast<u32x4*>(o) = *reinterpret_cast<const u32x4*>(src);
    else *reinterpret_cast<u32x2*>(o) = *reinterpret_cast<const u32x2*>(src);
}

template <typename T>
static int patchify_fwd_t(const void* x, void* out, int N, int H, int W, int C, int k, int ldo, hipStream_t s) {
    const int P = H / k, Q = W / k;
    constexpr int VF = 16 / sizeof(T);
    const long long rows = (long long)N * P * Q;
    if (C % VF == 0 && ldo % VF == 0) {
        const long long total = rows * (ldo / VF);
        hipLaunchKernelGGL((patchify_fwd_kernel<T, VF>), dim3(cn_grid(total)), dim3(256), 0, s, (const T*)x, (T*)out, N, H, W, C,
                           k, P, Q, ldo, total);
    } else {
        const long long total = rows * ldo;
        hipLaunchKernelGGL((patchify_fwd_kernel<T, 1>), dim3(cn_grid(total)), dim3(256), 0, s, (const T*)x, (T*)out, N, H, W, C,
                           k, P, Q, ldo, total);
    }
    HS_LAUNCH_CHECK();
    return HS_OK;
}
template <typename T>
static int patchify_bwd_t(const void* dp, void* dx, int N, int H, int W, int C, int k, int ldp, hipStream_t s) {
    const int P = H / k, Q = W / k;
    constexpr int VF = 16 / sizeof(T);
    if (C % VF == 0 && ldp % VF == 0) {
        const long long total = (long long)N * H * W * (C / VF);
        hipLaunchKernelGGL((patchify_bwd_kernel<T, VF>), dim3(cn_grid(total)), dim3(256), 0, s, (const T*)dp, (T*)dx, N, H, W, C,
                           k, P, Q, ldp, total);
    } else {
        const long long total = (long long)N * H * W * C;
        hipLaunchKernelGGL((patchify_bwd_kernel<T, 1>), dim3(cn_grid(total)), dim3(256), 0, s, (const T*)dp, (T*)dx, N, H, W, C,
                           k, P, Q, ldp, total);
    }
    HS_LAUNCH_CHECK();
    return HS_OK;
}

}  // namespace hs

using namespace hs;

extern "C" {

int64_t hs_dwconv_ws_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t ksize) {
    (void)W;
    const long long taps = (long long)ksize * ksize;
    return (taps * C + (long long)dw_chunks(N, H, C) * (taps + 1) * C) * 4;
}
hs_status hs_dwconv_fwd(int32_t dtype, const void* x, const float* w, const float* bias, void* y, int32_t N, int32_t H,
                        int32_t W, int32_t C, int32_t ksize, void* ws, int64_t ws_bytes, void* stream) {
    HS_REQUIRE(x && w && y && ws, "dwconv: null argument");
    HS_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "dwconv: C=%d must be a positive multiple of 4", C);
    HS_REQUIRE((long long)N * H * W * C * (dtype == HS_BF16 ? 2 : 4) < 0x7fffff00ll, "dwconv: activation larger than 2 GiB");
    HS_REQUIRE(ksize == 3 || ksize == 5 || ksize == 7, "dwconv: kernel size %d not in {3,5,7}", ksize);
    HS_REQUIRE(ws_bytes >= (int64_t)ksize * ksize * C * 4, "dwconv: workspace too small");
    HS_REQUIRE((long long)N * H * W * C < (1ll << 40), "dwconv: image too large");
    return CN_DISPATCH_T(dtype, dwconv_fwd_t, x, w, bias, y, N, H, W, C, ksize, (float*)ws, (hipStream_t)stream);
}
hs_status hs_dwconv_bwd(int32_t dtype, const void* x, const float* w, const void* dy, void* dx, float* dw, float* db,
                        int32_t N, int32_t H, int32_t W, int32_t C, int32_t ksize, void* ws, int64_t ws_bytes, void* stream) {
    HS_REQUIRE(x && w && dy && ws, "dwconv_bwd: null argument");
    HS_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "dwconv_bwd: C=%d must be a positive multiple of 4", C);
    HS_REQUIRE(ksize == 3 || ksize == 5 || ksize == 7, "dwconv_bwd: kernel size %d not in {3,5,7}", ksize);
    HS_REQUIRE(ws_bytes >= hs_dwconv_ws_bytes(N, H, W, C, ksize), "dwconv_bwd: workspace too small");
    HS_REQUIRE(dw || !db, "dwconv_bwd: db needs dw");
    HS_REQUIRE((long long)N * H * W * C * (dtype == HS_BF16 ? 2 : 4) < 0x7fffff00ll, "dwconv_bwd: activation larger than 2 GiB");
    return CN_DISPATCH_T(dtype, dwconv_bwd_t, x, w, dy, dx, dw, db, N, H, W, C, ksize, (float*)ws, (hipStream_t)stream);
}

int64_t hs_layerscale_ws_bytes(int64_t M, int32_t C) { return (int64_t)ls_gy(M) * C * 4; }
hs_status hs_layerscale_fwd(int32_t dtype, const void* u, const float* gamma, const float* rowscale, int32_t rows_per_sample,
                            const void* res, void* out, int64_t M, int32_t C, void* stream) {
    HS_REQUIRE(u && gamma && res && out && M > 0, "layerscale: null argument");
    HS_REQUIRE(C > 0 && C % 4 == 0, "layerscale: C=%d must be a multiple of 4", C);
    HS_REQUIRE(!rowscale || rows_per_sample > 0, "layerscale: rows_per_sample");
    return CN_DISPATCH_T(dtype, layerscale_fwd_t, u, gamma, rowscale, rows_per_sample, res, out, M, C, (hipStream_t)stream);
}
hs_status hs_layerscale_bwd(int32_t dtype, const void* dy, const void* u, const float* gamma, const float* rowscale,
                            int32_t rows_per_sample, void* du, float* dgamma, void* ws, int64_t ws_bytes, int64_t M, int32_t C,
                            void* stream) {
    HS_REQUIRE(dy && u && gamma && dgamma && ws && M > 0 && C > 0, "layerscale_bwd: null argument");
    HS_REQUIRE(ws_bytes >= hs_layerscale_ws_bytes(M, C), "layerscale_bwd: workspace too small");
    HS_REQUIRE(!rowscale || rows_per_sample > 0, "layerscale_bwd: rows_per_sample");
    return CN_DISPATCH_T(dtype, layerscale_bwd_t, dy, u, gamma, rowscale, rows_per_sample, du, dgamma, (float*)ws, M, C,
                         (hipStream_t)stream);
}

hs_status hs_patchify_fwd(int32_t dtype, const void* x, void* out, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k,
                          int32_t ldo, void* stream) {
    HS_REQUIRE(x && out && N > 0 && C > 0 && k > 0 && H >= k && W >= k, "patchify: bad argument");
    HS_REQUIRE(ldo >= k * k * C, "patchify: ldo=%d < k*k*C=%d", ldo, k * k * C);
    return CN_DISPATCH_T(dtype, patchify_fwd_t, x, out, N, H, W, C, k, ldo, (hipStream_t)stream);
}
hs_status hs_patchify_bwd(int32_t dtype, const void* dpatch, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k,
                          int32_t ldp, void* stream) {
    HS_REQUIRE(dpatch && dx && N > 0 && C > 0 && k > 0 && H >= k && W >= k, "patchify_bwd: bad argument");
    HS_REQUIRE(ldp >= k * k * C, "patchify_bwd: ldp=%d < k*k*C=%d", ldp, k * k * C);
    return CN_DISPATCH_T(dtype, patchify_bwd_t, dpatch, dx, N, H, W, C, k, ldp, (hipStream_t)stream);
}

}  // extern "C"
