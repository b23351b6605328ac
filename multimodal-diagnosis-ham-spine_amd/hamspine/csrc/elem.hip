// HBM-bound helpers of the hot path: pooling, packing, casts, column sums, row softmax (attention),
// BERT embeddings, softmax cross-entropy, small elementwise ops, fused AdamW.
// 16-byte accesses wherever the layout allows, wave-shuffle reductions, no atomics except the
// embedding scatter-add (duplicate token ids).
#include <algorithm>
#include "hs_common.h"

namespace hs {

static inline int grid_for(long long n, int cap = 4096) {
    long long b = (n + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// ============================================================================================
// max pool (NHWC), argmax kept as a byte so the backward is a gather (deterministic, no atomics).
// Replaces torch.nn.MaxPool2d(3, 2, 1) of the torchvision stem (reference encoder.py:63-68).
// ============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          unsigned char* __restrict__ idx, int N, int H, int W, int C,
                                                          int P, int Q, int ks, int st, int pad) {
    constexpr int E = Chunk<T>::N;
    const int cg = C / E;
    const long long total = (long long)N * P * Q * cg;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % cg);
        long long t = i / cg;
        const int q = (int)(t % Q);
        t /= Q;
        const int p = (int)(t % P);
        const int n = (int)(t / P);
        float best[E];
        unsigned char bi[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            best[e] = -INFINITY;
            bi[e] = 0;
        }
        for (int r = 0; r < ks; ++r) {
            const int h = p * st - pad + r;
            if ((unsigned)h >= (unsigned)H) continue;
            for (int s = 0; s < ks; ++s) {
                const int w = q * st - pad + s;
                if ((unsigned)w >= (unsigned)W) continue;
                float f[E];
                Chunk<T>::unpack(*(const u32x4*)(x + (((long long)n * H + h) * W + w) * C + c * E), f);
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if (f[e] > best[e] || f[e] != f[e]) {   // first maximum wins (torch CPU scan order)
                        best[e] = f[e];
                        bi[e] = (unsigned char)(r * ks + s);
                    }
            }
        }
        const long long o = (((long long)n * P + p) * Q + q) * C + c * E;
        *(u32x4*)(y + o) = Chunk<T>::pack(best);
#pragma unroll
        for (int e = 0; e < E; ++e) idx[o + e] = bi[e];
    }
}
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                          T* __restrict__ dx, int N, int H, int W, int C, int P, int Q,
                                                          int ks, int st, int pad) {
    constexpr int E = Chunk<T>::N;
    const int cg = C / E;
    const long long total = (long long)N * H * W * cg;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % cg);
        long long t = i / cg;
        const int w = (int)(t % W);
        t /= W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        float acc[E];
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = 0.f;
        const int p_lo = max(0, (h + pad - ks + st) / st), p_hi = min(P - 1, (h + pad) / st);
        const int q_lo = max(0, (w + pad - ks + st) / st), q_hi = min(Q - 1, (w + pad) / st);
        for (int p = p_lo; p <= p_hi; ++p)
            for (int q = q_lo; q <= q_hi; ++q) {
                const int r = h + pad - p * st, s = w + pad - q * st;
                if (r < 0 || r >= ks || s < 0 || s >= ks) continue;
                const unsigned char me = (unsigned char)(r * ks + s);
                const long long o = (((long long)n * P + p) * Q + q) * C + c * E;
                float g[E];
                Chunk<T>::unpack(*(const u32x4*)(dy + o), g);
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if (idx[o + e] == me) acc[e] += g[e];
            }
        *(u32x4*)(dx + (((long long)n * H + h) * W + w) * C + c * E) = Chunk<T>::pack(acc);
    }
}

// ============================================================================================
// mean over the token axis: x [B][Nt][H] -> y [B][H]   (global avg-pool / token pooling)
// Replaces AdaptiveAvgPool2d(1) of torchvision ResNet and tokens.mean(dim=1) /
// AdaptiveAvgPool1d(1) of the fusion modules (reference modules/fusion_blocks.py:97-98,170-178).
// ============================================================================================
template <typename T, typename TO>
__global__ __launch_bounds__(256) void mean_tokens_kernel(const T* __restrict__ x, TO* __restrict__ y, int Nt, int H,
                                                          int tpc, float scale) {
    constexpr int E = Chunk<T>::N;
    const int cg = H / E;
    const int tx = threadIdx.x % tpc, ty = threadIdx.x / tpc, rp = 256 / tpc;
    const int cc = blockIdx.x * tpc + tx;
    const int b = blockIdx.y;
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    if (cc < cg)
        for (int t = ty; t < Nt; t += rp) {
            float f[E];
            Chunk<T>::unpack(*(const u32x4*)(x + ((long long)b * Nt + t) * H + cc * E), f);
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] += f[e];
        }
    __shared__ float sh[256 * E];
#pragma unroll
    for (int e = 0; e < E; ++e) sh[threadIdx.x * E + e] = acc[e];
    __syncthreads();
    if (ty == 0 && cc < cg) {
        for (int j = 1; j < rp; ++j)
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] += sh[(j * tpc + tx) * E + e];
#pragma unroll
        for (int e = 0; e < E; ++e) y[(long long)b * H + cc * E + e] = from_f32<TO>(acc[e] * scale);
    }
}
// dx[b][t][:] = dy[b][:] * scale
template <typename T, typename TI>
__global__ __launch_bounds__(256) void mean_tokens_bwd_kernel(const TI* __restrict__ dy, T* __restrict__ dx, int Nt,
                                                              int H, long long nchunks, float scale) {
    constexpr int E = Chunk<T>::N;
    const int cg = H / E;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % cg);
        const long long b = i / ((long long)cg * Nt);
        float f[E];
#pragma unroll
        for (int e = 0; e < E; ++e) f[e] = to_f32(dy[b * H + c * E + e]) * scale;
        *(u32x4*)(dx + i * E) = Chunk<T>::pack(f);
    }
}

// ============================================================================================
// image packing for the stem: f32 NCHW [N][3][H][W] -> T [N][H+2p][Wp][4] with zero borders, so the
// 7x7/2 conv runs as an implicit GEMM with (R=7, S=1, C=32): one filter row = 8 pixels x 4 channels.
// ============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void pack_image_kernel(const float* __restrict__ x, T* __restrict__ y, int N, int Cin,
                                                         int H, int W, int Hp, int Wp, int pad) {
    const long long total = (long long)N * Hp * Wp;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int wp = (int)(i % Wp);
        long long t = i / Wp;
        const int hp = (int)(t % Hp);
        const int n = (int)(t / Hp);
        const int h = hp - pad, w = wp - pad;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W)
            for (int c = 0; c < Cin && c < 4; ++c) v[c] = x[(((long long)n * Cin + c) * H + h) * W + w];
        T* o = y + i * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = from_f32<T>(v[c]);
    }
}
// stem filter (K,3,7,7) stored channels_last [K][7][7][3] f32 -> T [K][7][8][4] (zero padded)
template <typename T>
__global__ void pack_stem_weight_kernel(const float* __restrict__ w, T* __restrict__ o, int K, int R, int S, int Cin) {
    const int total = K * R * 8 * 4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int c = i & 3, s = (i >> 2) & 7, r = (i >> 5) % R, k = (i >> 5) / R;
        float v = 0.f;
        if (s < S && c < Cin) v = w[((k * R + r) * S + s) * Cin + c];
        o[i] = from_f32<T>(v);
    }
}
// f32 [K][7][8][4] gradient -> f32 [K][7][7][3]
__global__ void unpack_stem_wgrad_kernel(const float* __restrict__ g, float* __restrict__ dw, int K, int R, int S,
                                         int Cin) {
    const int total = K * R * S * Cin;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int c = i % Cin, s = (i / Cin) % S, r = (i / (Cin * S)) % R, k = i / (Cin * S * R);
        dw[i] = g[((k * R + r) * 8 + s) * 4 + c];
    }
}

// ============================================================================================
// multi-tensor f32 -> bf16 cast (weights, once per step in bf16 mode)
// ============================================================================================
struct CastTable {
    const float* src[HS_CAST_MAX];
    void* dst[HS_CAST_MAX];
    long long n[HS_CAST_MAX];
};
__global__ __launch_bounds__(256) void cast_multi_kernel(const CastTable t) {
    const int e = blockIdx.y;
    const float* s = t.src[e];
    bf16_t* d = (bf16_t*)t.dst[e];
    const long long n = t.n[e];
    const long long n8 = n / 8;
    const bool al = ((((uintptr_t)s) & 15) == 0) && ((((uintptr_t)d) & 15) == 0);
    if (al) {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
            const f32x4 a = *(const f32x4*)(s + i * 8), b = *(const f32x4*)(s + i * 8 + 4);
            float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
            *(u32x4*)(d + i * 8) = Chunk<bf16_t>::pack(f);
        }
        for (long long i = n8 * 8 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
            d[i] = (bf16_t)s[i];
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
            d[i] = (bf16_t)s[i];
    }
}

// generic elementwise cast / axpby:  out = a*x + b*y  (y optional), any of f32/bf16 in and out
template <typename TX, typename TO>
__global__ __launch_bounds__(256) void axpby_kernel(const TX* __restrict__ x, const TX* __restrict__ y, TO* __restrict__ o,
                                                    long long n, float a, float b) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float v = a * to_f32(x[i]);
        if (y) v += b * to_f32(y[i]);
        o[i] = from_f32<TO>(v);
    }
}

// Same-type elementwise maps run on 16-byte chunks (8 bf16 / 4 f32 per lane) when every pointer is 16-byte aligned,
// with a scalar tail; `al` is decided on the host.
template <typename T, typename F1, typename FS>
__device__ __forceinline__ void map1(const T* __restrict__ x, T* __restrict__ o, long long n, bool al, F1 fvec, FS fs) {
    constexpr int V = Chunk<T>::N;
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
    const long long nv = al ? n / V : 0;
    for (long long i = tid; i < nv; i += stride) {
        float f[V];
        Chunk<T>::unpack(((const u32x4*)x)[i], f);
        fvec(i * V, f);
        ((u32x4*)o)[i] = Chunk<T>::pack(f);
    }
    for (long long i = nv * V + tid; i < n; i += stride) o[i] = from_f32<T>(fs(i, to_f32(x[i])));
}
template <typename T, typename F2>
__device__ __forceinline__ void map2(const T* __restrict__ x, const T* __restrict__ y, T* __restrict__ o, long long n, bool al,
                                     F2 f2) {
    constexpr int V = Chunk<T>::N;
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
    const long long nv = al ? n / V : 0;
    for (long long i = tid; i < nv; i += stride) {
        float f[V], g[V];
        Chunk<T>::unpack(((const u32x4*)x)[i], f);
        Chunk<T>::unpack(((const u32x4*)y)[i], g);
#pragma unroll
        for (int k = 0; k < V; ++k) f[k] = f2(f[k], g[k]);
        ((u32x4*)o)[i] = Chunk<T>::pack(f);
    }
    for (long long i = nv * V + tid; i < n; i += stride) o[i] = from_f32<T>(f2(to_f32(x[i]), to_f32(y[i])));
}
static inline bool al16(const void* a, const void* b = nullptr, const void* c = nullptr) {
    return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15) == 0;
}

// elementwise dropout (forward and backward use the same (seed, index) mask)
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ o, long long n, int al,
                                                      unsigned thresh, float inv_keep, unsigned long long seed) {
    constexpr int V = Chunk<T>::N;
    map1<T>(x, o, n, al != 0,
            [&](long long i0, float* f) {
#pragma unroll
                for (int h = 0; h < V / 4; ++h) {
                    float sc[4];
                    dropout_scale4(seed, i0 + 4 * h, thresh, inv_keep, sc);
#pragma unroll
                    for (int k = 0; k < 4; ++k) f[4 * h + k] *= sc[k];
                }
            },
            [&](long long i, float v) { return v * dropout_scale(seed, i, thresh, inv_keep); });
}

// relu forward / backward (standalone; the fused forms live in the GEMM and BN epilogues)
template <typename T>
__global__ __launch_bounds__(256) void relu_kernel(const T* __restrict__ x, T* __restrict__ o, long long n, int al) {
    constexpr int V = Chunk<T>::N;
    map1<T>(x, o, n, al != 0,
            [&](long long, float* f) {
#pragma unroll
                for (int k = 0; k < V; ++k) f[k] = fmaxf(f[k], 0.f);
            },
            [&](long long, float v) { return fmaxf(v, 0.f); });
}
template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                       T* __restrict__ o, long long n, int al) {
    map2<T>(dy, y, o, n, al != 0, [](float g, float yy) { return yy > 0.f ? g : 0.f; });
}

// dx = dy * gelu'(u)
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ u,
                                                       T* __restrict__ o, long long n, int al) {
    map2<T>(dy, u, o, n, al != 0, [](float g, float uu) { return g * gelu_grad_t<T>(uu); });
}
// out = x * (*scalar)   (scalar lives on the device: no host sync in loss backward)
__global__ __launch_bounds__(256) void mul_dev_scalar_kernel(const float* __restrict__ x, const float* __restrict__ sc,
                                                             float* __restrict__ o, long long n) {
    const float s = sc[0];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) o[i] = x[i] * s;
}

// ============================================================================================
// column sum: X [M][N] -> out[N] (f32), used for bias gradients.  Deterministic two-stage.
// ============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, long long M, int N, int ld,
                                                             float* __restrict__ ws) {
    // thread per column, blockIdx.y strides rows (generic, works for any N / ld)
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    float acc = 0.f;
    for (long long r = blockIdx.y; r < M; r += gridDim.y) acc += to_f32(x[r * ld + c]);
    ws[(long long)blockIdx.y * N + c] = acc;
}
// 16-byte variant (N, ld multiples of the chunk, base aligned): a block covers min(64, N / chunk) chunk columns and
// 256 / that many rows per pass (narrow matrices -- ConvNeXt's 200704 x 128 -- used to leave 3/4 of a 64-column block idle:
// 155 us); the row lanes merge through LDS in a fixed order
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_vec_kernel(const T* __restrict__ x, long long M, int N, int ld,
                                                                 float* __restrict__ ws) {
    constexpr int E = Chunk<T>::N;
    __shared__ float sh[256][E];
    const int ncol = min(64, N / E - (int)blockIdx.x * 64);     // chunk columns of this block
    const int rpb = 256 / ncol;                                 // rows per pass
    const int tid = threadIdx.x;
    const int ty = tid / ncol, tx = tid - ty * ncol;
    const int cc = blockIdx.x * 64 + tx;
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    if (ty < rpb) {
        for (long long r = (long long)blockIdx.y * rpb + ty; r < M; r += (long long)gridDim.y * rpb) {
            float f[E];
            Chunk<T>::unpack(*(const u32x4*)(x + r * ld + (long long)cc * E), f);
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] += f[e];
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) sh[tid][e] = acc[e];
    __syncthreads();
    if (ty == 0) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            float t = 0.f;
            for (int r = 0; r < rpb; ++r) t += sh[r * ncol + tx][e];
            ws[(long long)blockIdx.y * N + cc * E + e] = t;
        }
    }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ ws, int gy, int N,
                                                           float* __restrict__ out, int accumulate) {
    __shared__ float sh[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float acc = 0.f;
    if (c < N) {
        int i = rg;
        for (; i + 12 < gy; i += 16) {
            const float v0 = ws[(long long)i * N + c], v1 = ws[(long long)(i + 4) * N + c];
            const float v2 = ws[(long long)(i + 8) * N + c], v3 = ws[(long long)(i + 12) * N + c];
            acc += (v0 + v1) + (v2 + v3);
        }
        for (; i < gy; i += 4) acc += ws[(long long)i * N + c];
    }
    sh[rg][cl] = acc;
    __syncthreads();
    if (rg == 0 && c < N) {
        const float t = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
        out[c] = accumulate ? out[c] + t : t;
    }
}

// ============================================================================================
// attention row softmax over materialised scores (f32) with key mask and dropout.
//   P[r][k] = softmax_k(S[r][k] + (mask[b][k] ? 0 : -big));   Pd = P * keep / (1-p)
// Rows are (b, h, q) flattened; b = r / rows_per_batch.
// Replaces the softmax + dropout inside torch.nn.MultiheadAttention (modules/fusion_blocks.py:48,62)
// and transformers BertSelfAttention.
// ============================================================================================
constexpr int SM_MAXV = 16;   // Lk <= 1024
template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ S, const long long* __restrict__ mask,
                                                          T* __restrict__ P, T* __restrict__ Pd, long long rows,
                                                          int Lk, int ldS, int ldP, int rows_per_batch,
                                                          unsigned thresh, float inv_keep, unsigned long long seed) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const long long b = r / rows_per_batch;
    float v[SM_MAXV];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < SM_MAXV; ++i) {
        const int k = lane + 64 * i;
        v[i] = -INFINITY;
        if (k < Lk) {
            float s = S[r * ldS + k];
            if (mask && mask[b * Lk + k] == 0) s = -3.0e38f;
            v[i] = s;
            mx = fmaxf(mx, s);
        }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < SM_MAXV; ++i) {
        const int k = lane + 64 * i;
        if (k < Lk) {
            v[i] = __expf(v[i] - mx);
            sum += v[i];
        }
    }
    const float inv = 1.f / wave_sum(sum);
#pragma unroll
    for (int i = 0; i < SM_MAXV; ++i) {
        const int k = lane + 64 * i;
        if (k < ldP) {
            const float p = k < Lk ? v[i] * inv : 0.f;   // padding columns are written as zeros
            P[r * ldP + k] = from_f32<T>(p);
            if (Pd) {
                const float sc = (thresh && k < Lk) ? dropout_scale(seed, (unsigned long long)r * Lk + k, thresh, inv_keep) : 1.f;
                Pd[r * ldP + k] = from_f32<T>(p * sc);
            }
        }
    }
}
// dS = P * (dPd - sum_k P*dPd),  dPd = dP * keep/(1-p)
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ dP, const T* __restrict__ P,
                                                          T* __restrict__ dS, long long rows, int Lk, int ldG, int ldP,
                                                          unsigned thresh, float inv_keep, unsigned long long seed) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float g[SM_MAXV], p[SM_MAXV];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < SM_MAXV; ++i) {
        const int k = lane + 64 * i;
        g[i] = p[i] = 0.f;
        if (k < Lk) {
            float d = dP[r * ldG + k];
            if (thresh) d *= dropout_scale(seed, (unsigned long long)r * Lk + k, thresh, inv_keep);
            g[i] = d;
            p[i] = to_f32(P[r * ldP + k]);
            dot += d * p[i];
        }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int i = 0; i < SM_MAXV; ++i) {
        const int k = lane + 64 * i;
        if (k < ldP) dS[r * ldP + k] = from_f32<T>(k < Lk ? p[i] * (g[i] - dot) : 0.f);
    }
}

// 16-byte variants for Lk % 4 == 0, Lk <= 256 and 4-aligned pitches: G = pow2 >= Lk/4 lanes own one row (4 consecutive
// keys per lane), so a wave carries 64/G rows, every access is a 16/8-byte vector and one dropout hash serves 4 elements.
template <int G> __device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
template <int G> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <typename T> __device__ __forceinline__ void store4_any(T* p, const float* f);
template <> __device__ __forceinline__ void store4_any<float>(float* p, const float* f) {
    *(f32x4*)p = f32x4{f[0], f[1], f[2], f[3]};
}
template <> __device__ __forceinline__ void store4_any<bf16_t>(bf16_t* p, const float* f) {
    *(bf16x4*)p = bf16x4{(bf16_t)f[0], (bf16_t)f[1], (bf16_t)f[2], (bf16_t)f[3]};
}
template <typename T> __device__ __forceinline__ void load4_any(const T* p, float* f);
template <> __device__ __forceinline__ void load4_any<float>(const float* p, float* f) {
    const f32x4 v = *(const f32x4*)p;
    f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
}
template <> __device__ __forceinline__ void load4_any<bf16_t>(const bf16_t* p, float* f) {
    const u32x2 v = *(const u32x2*)p;
    f[0] = __uint_as_float(v[0] << 16); f[1] = __uint_as_float(v[0] & 0xffff0000u);
    f[2] = __uint_as_float(v[1] << 16); f[3] = __uint_as_float(v[1] & 0xffff0000u);
}

template <typename T, int G>
__global__ __launch_bounds__(256) void softmax_fwd_vec_kernel(const float* __restrict__ S, const long long* __restrict__ mask,
                                                              T* __restrict__ P, T* __restrict__ Pd, long long rows, int Lk,
                                                              int ldS, int ldP, int rows_per_batch, unsigned thresh,
                                                              float inv_keep, unsigned long long seed) {
    constexpr int RW = 64 / G;
    const int lane = threadIdx.x & 63, sub = lane / G, li = lane % G;
    const long long r = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RW + sub;
    if (r >= rows) return;            // whole row groups leave together; the shuffles below stay inside a group
    const int k0 = li * 4;
    const bool act = k0 < Lk;
    float v[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (act) {
        load4_any<float>(S + r * ldS + k0, v);
        if (mask) {
            const long long* mk = mask + (r / rows_per_batch) * Lk + k0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (mk[j] == 0) v[j] = -3.0e38f;
        }
    }
    const float mx = group_max<G>(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
    float sum = 0.f;
    if (act) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = __expf(v[j] - mx);
            sum += v[j];
        }
    }
    const float inv = 1.f / group_sum<G>(sum);
    if (k0 < ldP) {
        float p[4] = {0.f, 0.f, 0.f, 0.f};   // padding columns are written as zeros
        if (act) {
#pragma unroll
            for (int j = 0; j < 4; ++j) p[j] = v[j] * inv;
        }
        store4_any<T>(P + r * ldP + k0, p);
        if (Pd) {
            if (thresh && act) {
                float sc[4];
                dropout_scale4(seed, (unsigned long long)r * Lk + k0, thresh, inv_keep, sc);
#pragma unroll
                for (int j = 0; j < 4; ++j) p[j] *= sc[j];
            }
            store4_any<T>(Pd + r * ldP + k0, p);
        }
    }
}
template <typename T, int G>
__global__ __launch_bounds__(256) void softmax_bwd_vec_kernel(const float* __restrict__ dP, const T* __restrict__ P,
                                                              T* __restrict__ dS, long long rows, int Lk, int ldG, int ldP,
                                                              unsigned thresh, float inv_keep, unsigned long long seed) {
    constexpr int RW = 64 / G;
    const int lane = threadIdx.x & 63, sub = lane / G, li = lane % G;
    const long long r = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RW + sub;
    if (r >= rows) return;
    const int k0 = li * 4;
    const bool act = k0 < Lk;
    float g[4] = {0.f, 0.f, 0.f, 0.f}, p[4] = {0.f, 0.f, 0.f, 0.f};
    float dot = 0.f;
    if (act) {
        load4_any<float>(dP + r * ldG + k0, g);
        load4_any<T>(P + r * ldP + k0, p);
        if (thresh) {
            float sc[4];
            dropout_scale4(seed, (unsigned long long)r * Lk + k0, thresh, inv_keep, sc);
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] *= sc[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) dot += g[j] * p[j];
    }
    dot = group_sum<G>(dot);
    if (k0 < ldP) {
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (act) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = p[j] * (g[j] - dot);
        }
        store4_any<T>(dS + r * ldP + k0, o);
    }
}

// ============================================================================================
// BERT embeddings: word[ids] + pos[l] + type[0]  (pre-LN sum saved), LayerNorm, dropout.
// Replaces transformers BertEmbeddings.forward under reference encoder.py:131 / mibf_net/bert.py:12.
// ============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const long long* __restrict__ ids, const float* __restrict__ word,
                                                        const float* __restrict__ pos, const float* __restrict__ type0,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        T* __restrict__ sum_out, T* __restrict__ y, float* __restrict__ mean_out,
                                                        float* __restrict__ rstd_out, long long tokens, int L, int H, int V,
                                                        float eps, unsigned thresh, float inv_keep, unsigned long long seed) {
    const int lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= tokens) return;
    long long id = ids[t];
    if (id < 0) id = 0;
    if (id >= V) id = V - 1;
    const int l = (int)(t % L);
    const int nch = H / 4;
    float v[4][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            const f32x4 a = *(const f32x4*)(word + id * H + c * 4);
            const f32x4 b = *(const f32x4*)(pos + (long long)l * H + c * 4);
            const f32x4 d = *(const f32x4*)(type0 + c * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // the saved sum is what the backward LN sees; round it to T first so fwd == bwd inputs
                v[i][e] = to_f32(from_f32<T>(a[e] + b[e] + d[e]));
                s += v[i][e];
            }
        }
    }
    const float mean = wave_sum(s) / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c < nch)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[i][e] - mean;
                sq += d * d;
            }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)H + eps);
    if (lane == 0) {
        mean_out[t] = mean;
        rstd_out[t] = rstd;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c < nch)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int h = c * 4 + e;
                sum_out[t * H + h] = from_f32<T>(v[i][e]);
                float o = (v[i][e] - mean) * rstd * gamma[h] + beta[h];
                if (thresh) o *= dropout_scale(seed, (unsigned long long)t * H + h, thresh, inv_keep);
                y[t * H + h] = from_f32<T>(o);
            }
    }
}
// Gradient of the word table, deterministic and atomics-free: one workgroup per token t.  The workgroup lists every token
// that carries t's id, in token order (256 contiguous id chunks, ordered compaction through a block prefix sum); only the
// FIRST occurrence of an id owns its table row, sums the listed dsum rows in that fixed order and stores the row once.
// Duplicate ids (frequent in text) therefore add up in the same order on every run, whatever else shares the chip
// (float atomics here made two runs of one step differ in their last bits).  Rows of ids that do not occur stay as the
// caller zero-filled them; the pad row receives no gradient (nn.Embedding(padding_idx)).
__device__ __forceinline__ long long clamp_id(long long id, int V) { return id < 0 ? 0 : (id >= V ? V - 1 : id); }
template <typename T>
__global__ __launch_bounds__(256) void embed_word_bwd_kernel(const long long* __restrict__ ids, const T* __restrict__ dsum,
                                                             float* __restrict__ dword, int tokens, int H, int V,
                                                             int pad_id) {
    extern __shared__ __attribute__((aligned(16))) int s_mem[];   // [tokens] match list + [8] wave totals
    int* s_list = s_mem;
    int* s_wave = s_mem + tokens;
    const int t = blockIdx.x;
    const long long id = clamp_id(ids[t], V);
    if (id == pad_id) return;                                     // block-uniform
    const int per = (tokens + 255) / 256;
    const int lo = min((int)threadIdx.x * per, tokens), hi = min(lo + per, tokens);
    int c = 0;
    for (int i = lo; i < hi; ++i) c += clamp_id(ids[i], V) == id ? 1 : 0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int incl = c;                                                 // inclusive scan inside the wave
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    int off = incl - c;
    for (int w = 0; w < wv; ++w) off += s_wave[w];
    const int n = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    for (int i = lo; i < hi; ++i)
        if (clamp_id(ids[i], V) == id) s_list[off++] = i;
    __syncthreads();
    if (s_list[0] != t) return;                                   // a smaller token index owns this id (block-uniform)
    for (int h = threadIdx.x; h < H; h += 256) {
        float acc = 0.f;
        for (int k = 0; k < n; ++k) acc += to_f32(dsum[(long long)s_list[k] * H + h]);
        dword[id * H + h] = acc;
    }
}
// the same scatter with float atomics: only for token counts whose match list does not fit the LDS
template <typename T>
__global__ __launch_bounds__(256) void embed_word_bwd_atomic_kernel(const long long* __restrict__ ids,
                                                                    const T* __restrict__ dsum, float* __restrict__ dword,
                                                                    long long tokens, int H, int V, int pad_id) {
    const long long total = tokens * H;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long t = i / H;
        const int h = (int)(i % H);
        const long long id = clamp_id(ids[t], V);
        if (id == pad_id) continue;
        atomicAdd(dword + id * H + h, to_f32(dsum[i]));
    }
}
// dpos[l][h] = sum_b dsum[b][l][h]   (rows l >= L are zeroed by the caller)
template <typename T>
__global__ __launch_bounds__(256) void embed_pos_bwd_kernel(const T* __restrict__ dsum, float* __restrict__ dpos, int B,
                                                            int L, int H) {
    const long long total = (long long)L * H;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += to_f32(dsum[(long long)b * L * H + i]);
        dpos[i] = acc;
    }
}

// ============================================================================================
// bf16 matrix transpose through LDS: dst[c][r] = src[r][c].  Weight gradients dW = dY^T X reduce over the token dimension,
// which is the SLOW index of both row-major operands; staged as [k][row] tiles ("tn") each workgroup walks a 128-byte-wide
// column stripe through its operands and the GEMM core measured 283 TFLOP/s on BERT-base shapes against 480 for
// K-contiguous operands of the same size (tools/gemm_sweep.py).  Transposing dY and X once (HBM-bound, ~40 us for the 100 MB
// of a BertLayer) and running the weight gradients as K-contiguous GEMMs is the faster total.
// 64x64 tiles; LDS image of 32-bit words with a 33-word pitch, the word index XOR-ed with 4 in the lower half of the rows: the
// read phase has rows 8k + q and 8(k + 4) + q in one 32-lane group, which a pitch alone leaves on the same bank (PMC: 40 % of
// the LDS cycles were conflicts); 16-byte global accesses.
// ============================================================================================
__device__ __forceinline__ void transpose_bf16_tile(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst,
                                                    int R, int Cc, long long lds_, long long ldd, int bx, int by,
                                                    unsigned (*tile)[33]) {
    const int r0 = by * 64, c0 = bx * 64;
    for (int i = threadIdx.x; i < 512; i += 256) {
        const int r = i >> 3, ch = i & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r0 + r < R && c0 + ch * 8 < Cc) v = *(const u32x4*)(src + (long long)(r0 + r) * lds_ + c0 + ch * 8);
#pragma unroll
        for (int w = 0; w < 4; ++w) tile[r][(ch * 4 + w) ^ ((r >> 5) << 2)] = v[w];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256) {
        const int j = i >> 3, k = i & 7;             // output row c0 + j, output columns r0 + 8k .. r0 + 8k + 7
        if (c0 + j >= Cc || r0 + k * 8 >= R) continue;
        unsigned e[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const unsigned w = tile[k * 8 + q][(j >> 1) ^ (((k * 8 + q) >> 5) << 2)];
            e[q] = (j & 1) ? (w >> 16) : (w & 0xffffu);
        }
        const u32x4 o = {e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16)};
        *(u32x4*)(dst + (long long)(c0 + j) * ldd + r0 + k * 8) = o;
    }
}
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const unsigned short* __restrict__ src,
                                                             unsigned short* __restrict__ dst, int R, int Cc, long long lds_,
                                                             long long ldd) {
    __shared__ unsigned tile[64][33];
    transpose_bf16_tile(src, dst, R, Cc, lds_, ldd, blockIdx.x, blockIdx.y, tile);
}
// several matrices in one grid (a BertLayer's backward transposes four saved activations and three weight copies before it
// starts: seven 7 us launches become one)
struct TransposeTable {
    const unsigned short* src[HS_TRANSPOSE_MAX];
    unsigned short* dst[HS_TRANSPOSE_MAX];
    int R[HS_TRANSPOSE_MAX], C[HS_TRANSPOSE_MAX];
    long long ld_src[HS_TRANSPOSE_MAX], ld_dst[HS_TRANSPOSE_MAX];
    int first[HS_TRANSPOSE_MAX + 1];      // first workgroup of entry i; first[count] = grid size
    int count;
};
__global__ __launch_bounds__(256) void transpose_bf16_multi_kernel(const TransposeTable t) {
    __shared__ unsigned tile[64][33];
    int e = 0;
    for (int i = 1; i < t.count; ++i) e += (int)blockIdx.x >= t.first[i];
    const int local = blockIdx.x - t.first[e];
    const int tx = (t.C[e] + 63) / 64;
    transpose_bf16_tile(t.src[e], t.dst[e], t.R[e], t.C[e], t.ld_src[e], t.ld_dst[e], local % tx, local / tx, tile);
}

// ============================================================================================
// softmax cross-entropy with class weights and label smoothing (mean reduction), loss + dlogits.
// Replaces nn.CrossEntropyLoss(weight, label_smoothing=0.02) (reference scripts/train.py:240,252-254)
// and F.cross_entropy (mibf_net/model_resnet.py:88-90).  One block, rows strided over waves.
// ============================================================================================
__global__ __launch_bounds__(1024) void ce_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                 const float* __restrict__ weight, float smoothing, int B, int C,
                                                 float* __restrict__ loss_out, float* __restrict__ dlogits,
                                                 float* __restrict__ row_loss) {
    __shared__ float s_num[16], s_den[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;      // 4..16 waves: a wave per 1-2 rows of a batch of 32
    float num = 0.f, den = 0.f;
    for (int r = wv; r < B; r += nwv) {
        const float* z = logits + (long long)r * C;
        float mx = -INFINITY;
        for (int c = lane; c < C; c += 64) mx = fmaxf(mx, z[c]);
        mx = wave_max(mx);
        float se = 0.f;
        for (int c = lane; c < C; c += 64) se += __expf(z[c] - mx);
        const float lse = mx + __logf(wave_sum(se));
        const int y = (int)labels[r];
        const float wy = weight ? weight[y] : 1.f;
        float sm = 0.f;
        for (int c = lane; c < C; c += 64) sm += (weight ? weight[c] : 1.f) * (lse - z[c]);
        sm = wave_sum(sm);
        const float l = (1.f - smoothing) * wy * (lse - z[y]) + smoothing / (float)C * sm;
        if (lane == 0 && row_loss) row_loss[r] = l;
        num += l;
        den += wy;
    }
    if (lane == 0) {
        s_num[wv] = num;
        s_den[wv] = den;
    }
    __syncthreads();
    float tn = 0.f, td = 0.f;
    for (int w = 0; w < nwv; ++w) {
        tn += s_num[w];
        td += s_den[w];
    }
    if (threadIdx.x == 0) loss_out[0] = tn / td;
    if (!dlogits) return;
    for (int r = wv; r < B; r += nwv) {
        const float* z = logits + (long long)r * C;
        float mx = -INFINITY;
        for (int c = lane; c < C; c += 64) mx = fmaxf(mx, z[c]);
        mx = wave_max(mx);
        float se = 0.f;
        for (int c = lane; c < C; c += 64) se += __expf(z[c] - mx);
        const float inv = 1.f / wave_sum(se);
        const int y = (int)labels[r];
        const float wy = weight ? weight[y] : 1.f;
        float wsum = 0.f;
        for (int c = lane; c < C; c += 64) wsum += weight ? weight[c] : 1.f;
        wsum = wave_sum(wsum);
        for (int c = lane; c < C; c += 64) {
            const float p = __expf(z[c] - mx) * inv;
            const float wc = weight ? weight[c] : 1.f;
            // d/dz_c of [(1-s) wy (lse - z_y) + s/C sum_k w_k (lse - z_k)]
            float g = (1.f - smoothing) * wy * (p - (c == y ? 1.f : 0.f)) + smoothing / (float)C * (wsum * p - wc);
            dlogits[(long long)r * C + c] = g / td;
        }
    }
}

// ============================================================================================
// fused multi-tensor AdamW / Adam step (f32 params).  HBM-bound: 16 B read + 12 B written per param.
// Replaces torch.optim.{Adam,AdamW}.step (reference scripts/train.py:257-261, train_resnet.py:139).
// ============================================================================================
struct AdamTable {
    float* p[HS_ADAM_MAX];
    const float* g[HS_ADAM_MAX];
    float* m[HS_ADAM_MAX];
    float* v[HS_ADAM_MAX];
    long long n[HS_ADAM_MAX];
    bf16_t* h[HS_ADAM_MAX];      // optional bf16 shadow of p (the compute-dtype copy the towers read), written with the update
};
__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamTable t, float lr, float beta1, float beta2, float eps,
                                                         float wd, float bc1, float bc2_sqrt, int decoupled,
                                                         float grad_scale) {
    const int e = blockIdx.y;
    float* p = t.p[e];
    const float* g = t.g[e];
    float* m = t.m[e];
    float* v = t.v[e];
    const long long n = t.n[e];
    auto upd = [&](float& pi, float gi, float& mi, float& vi) {
        gi *= grad_scale;
        if (decoupled) pi *= 1.f - lr * wd;
        else gi += wd * pi;
        mi = beta1 * mi + (1.f - beta1) * gi;
        vi = beta2 * vi + (1.f - beta2) * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi = pi - (lr / bc1) * (mi / denom);
    };
    bf16_t* h = t.h[e];
    const bool al = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0 && (((uintptr_t)h) & 7) == 0;
    const long long n4 = al ? n / 4 : 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f32x4 pv = ((f32x4*)p)[i], mv = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
        const f32x4 gv = ((const f32x4*)g)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pi = pv[k], mi = mv[k], vi = vv[k];
            upd(pi, gv[k], mi, vi);
            pv[k] = pi; mv[k] = mi; vv[k] = vi;
        }
        ((f32x4*)p)[i] = pv;
        ((f32x4*)m)[i] = mv;
        ((f32x4*)v)[i] = vv;
        if (h) ((bf16x4*)h)[i] = bf16x4{(bf16_t)pv[0], (bf16_t)pv[1], (bf16_t)pv[2], (bf16_t)pv[3]};
    }
    for (long long i = n4 * 4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float pi = p[i], mi = m[i], vi = v[i];
        upd(pi, g[i], mi, vi);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (h) h[i] = (bf16_t)pi;
    }
}

// fused multi-tensor SGD (torch.optim.SGD semantics: weight decay added to the gradient, optional momentum buffer with
// dampening 0, optional Nesterov); `first` = the momentum buffers are uninitialised (torch seeds them with the gradient)
__global__ __launch_bounds__(256) void sgd_multi_kernel(const AdamTable t, float lr, float momentum, float wd, int nesterov,
                                                        int first, float grad_scale) {
    const int e = blockIdx.y;
    float* p = t.p[e];
    const float* g = t.g[e];
    float* buf = t.m[e];            // NULL when momentum == 0
    const long long n = t.n[e];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float gi = g[i] * grad_scale;
        const float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        if (buf) {
            const float b = first ? gi : momentum * buf[i] + gi;
            buf[i] = b;
            gi = nesterov ? gi + momentum * b : b;
        }
        p[i] = pi - lr * gi;
    }
}

// --------------------------------------------------------------------------------------------
// host wrappers
// --------------------------------------------------------------------------------------------
#define DISPATCH_T(dtype, fn, ...) ((dtype) == HS_BF16 ? fn<bf16_t>(__VA_ARGS__) : fn<float>(__VA_ARGS__))

template <typename T>
static int maxpool_fwd_t(const void* x, void* y, void* idx, int N, int H, int W, int C, int P, int Q, int ks, int st,
                         int pad, hipStream_t s) {
    HS_REQUIRE(C % Chunk<T>::N == 0, "maxpool: C %% %d != 0", Chunk<T>::N);
    const long long total = (long long)N * P * Q * (C / Chunk<T>::N);
    hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, (const T*)x, (T*)y,
                       (unsigned char*)idx, N, H, W, C, P, Q, ks, st, pad);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
template <typename T>
static int maxpool_bwd_t(const void* dy, const void* idx, void* dx, int N, int H, int W, int C, int P, int Q, int ks,
                         int st, int pad, hipStream_t s) {
    HS_REQUIRE(C % Chunk<T>::N == 0, "maxpool_bwd: C %% %d != 0", Chunk<T>::N);
    const long long total = (long long)N * H * W * (C / Chunk<T>::N);
    hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, (const T*)dy,
                       (const unsigned char*)idx, (T*)dx, N, H, W, C, P, Q, ks, st, pad);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
static void mean_geom(int H, int E, int& tpc, int& gx) {
    const int cg = H / E;
    tpc = 1;
    while (tpc < cg && tpc < 64) tpc <<= 1;
    gx = ceil_div(cg, tpc);
}
template <typename T>
static int mean_tokens_t(const void* x, void* y, int B, int Nt, int H, int out_f32, hipStream_t s) {
    constexpr int E = Chunk<T>::N;
    HS_REQUIRE(H % E == 0, "mean_tokens: H %% %d != 0", E);
    int tpc, gx;
    mean_geom(H, E, tpc, gx);
    if (out_f32)
        hipLaunchKernelGGL((mean_tokens_kernel<T, float>), dim3(gx, B), dim3(256), 0, s, (const T*)x, (float*)y, Nt, H, tpc,
                           1.f / (float)Nt);
    else
        hipLaunchKernelGGL((mean_tokens_kernel<T, T>), dim3(gx, B), dim3(256), 0, s, (const T*)x, (T*)y, Nt, H, tpc,
                           1.f / (float)Nt);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
template <typename T>
static int mean_tokens_bwd_t(const void* dy, void* dx, int B, int Nt, int H, int dy_f32, hipStream_t s) {
    constexpr int E = Chunk<T>::N;
    HS_REQUIRE(H % E == 0, "mean_tokens_bwd: H %% %d != 0", E);
    const long long nch = (long long)B * Nt * H / E;
    if (dy_f32)
        hipLaunchKernelGGL((mean_tokens_bwd_kernel<T, float>), dim3(grid_for(nch)), dim3(256), 0, s, (const float*)dy,
                           (T*)dx, Nt, H, nch, 1.f / (float)Nt);
    else
        hipLaunchKernelGGL((mean_tokens_bwd_kernel<T, T>), dim3(grid_for(nch)), dim3(256), 0, s, (const T*)dy, (T*)dx, Nt,
                           H, nch, 1.f / (float)Nt);
    HS_LAUNCH_CHECK();
    return HS_OK;
}

// row groups (grid.y): enough blocks to cover the chip also when the matrix is narrow (one or two blocks along x)
static inline int colsum_gy(long long M, int N) {
    if (M <= 64) return 1;             // a head's batch of rows: one row group, and its partial row IS the result (colsum_t)
    const long long cap = std::max<long long>(64, 1024 / ceil_div(N, 512));
    return (int)std::min<long long>(std::max<long long>(M / 16, 1), cap);
}
template <typename T>
static int colsum_t(const void* x, long long M, int N, int ld, float* out, float* ws, long long ws_bytes, int accumulate,
                    hipStream_t s) {
    constexpr int E = Chunk<T>::N;
    const int gy = colsum_gy(M, N);
    HS_REQUIRE(ws && ws_bytes >= (long long)gy * N * 4, "colsum: workspace too small");
    const bool direct = gy == 1 && !accumulate;        // one row group: its sums go straight to `out`, no second launch
    if (direct) ws = out;
    if (N % E == 0 && ld % E == 0 && ((((uintptr_t)x) & 15) == 0))
        hipLaunchKernelGGL(colsum_partial_vec_kernel<T>, dim3(ceil_div(N / E, 64), gy), dim3(256), 0, s, (const T*)x, M, N, ld, ws);
    else
        hipLaunchKernelGGL(colsum_partial_kernel<T>, dim3(ceil_div(N, 256), gy), dim3(256), 0, s, (const T*)x, M, N, ld, ws);
    HS_LAUNCH_CHECK();
    if (direct) return HS_OK;
    hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(N, 64)), dim3(256), 0, s, ws, gy, N, out, accumulate);
    HS_LAUNCH_CHECK();
    return HS_OK;
}

static inline int softmax_group(int Lk, int ldP) {   // lanes per row of the 16-byte kernels, 0 = not eligible
    if (Lk % 4 || ldP % 4 || ldP > 256) return 0;
    const int need = ldP / 4;
    return need <= 16 ? 16 : (need <= 32 ? 32 : 64);
}
template <typename T>
static int softmax_fwd_t(const float* S, const long long* mask, void* P, void* Pd, long long rows, int Lk, int ldS,
                         int ldP, int rpb, float p, unsigned long long seed, hipStream_t s) {
    HS_REQUIRE(Lk <= 64 * SM_MAXV && ldP <= 64 * SM_MAXV, "softmax: Lk=%d too long (max %d)", Lk, 64 * SM_MAXV);
    const unsigned th = p > 0.f ? dropout_thresh(p) : 0u;
    const float ik = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const int G = (ldS % 4 == 0 && ((((uintptr_t)S) | ((uintptr_t)P) | ((uintptr_t)Pd)) & 15) == 0) ? softmax_group(Lk, ldP) : 0;
#define SM_FWD(GG)                                                                                                       \
    hipLaunchKernelGGL((softmax_fwd_vec_kernel<T, GG>), dim3(ceil_div(rows, 4 * (64 / GG))), dim3(256), 0, s, S, mask, (T*)P, \
                       (T*)Pd, rows, Lk, ldS, ldP, rpb, th, ik, seed)
    if (G == 16) SM_FWD(16);
    else if (G == 32) SM_FWD(32);
    else if (G == 64) SM_FWD(64);
    else
        hipLaunchKernelGGL(softmax_fwd_kernel<T>, dim3(ceil_div(rows, 4)), dim3(256), 0, s, S, mask, (T*)P, (T*)Pd, rows, Lk,
                           ldS, ldP, rpb, th, ik, seed);
#undef SM_FWD
    HS_LAUNCH_CHECK();
    return HS_OK;
}
template <typename T>
static int softmax_bwd_t(const float* dP, const void* P, void* dS, long long rows, int Lk, int ldG, int ldP, float p,
                         unsigned long long seed, hipStream_t s) {
    HS_REQUIRE(Lk <= 64 * SM_MAXV && ldP <= 64 * SM_MAXV, "softmax_bwd: Lk=%d too long", Lk);
    const unsigned th = p > 0.f ? dropout_thresh(p) : 0u;
    const float ik = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const int G = (ldG % 4 == 0 && ((((uintptr_t)dP) | ((uintptr_t)P) | ((uintptr_t)dS)) & 15) == 0) ? softmax_group(Lk, ldP) : 0;
#define SM_BWD(GG)                                                                                                      \
    hipLaunchKernelGGL((softmax_bwd_vec_kernel<T, GG>), dim3(ceil_div(rows, 4 * (64 / GG))), dim3(256), 0, s, dP, (const T*)P, \
                       (T*)dS, rows, Lk, ldG, ldP, th, ik, seed)
    if (G == 16) SM_BWD(16);
    else if (G == 32) SM_BWD(32);
    else if (G == 64) SM_BWD(64);
    else
        hipLaunchKernelGGL(softmax_bwd_kernel<T>, dim3(ceil_div(rows, 4)), dim3(256), 0, s, dP, (const T*)P, (T*)dS, rows, Lk,
                           ldG, ldP, th, ik, seed);
#undef SM_BWD
    HS_LAUNCH_CHECK();
    return HS_OK;
}

}  // namespace hs

using namespace hs;

extern "C" {

hs_status hs_maxpool_fwd(int32_t dtype, const void* x, void* y, void* idx, int32_t N, int32_t H, int32_t W, int32_t C,
                         int32_t ksize, int32_t stride, int32_t pad, void* stream) {
    HS_REQUIRE(x && y && idx, "maxpool: null argument");
    HS_REQUIRE(ksize * ksize <= 255, "maxpool: window too large");
    const int P = (H + 2 * pad - ksize) / stride + 1, Q = (W + 2 * pad - ksize) / stride + 1;
    return DISPATCH_T(dtype, maxpool_fwd_t, x, y, idx, N, H, W, C, P, Q, ksize, stride, pad, (hipStream_t)stream);
}
hs_status hs_maxpool_bwd(int32_t dtype, const void* dy, const void* idx, void* dx, int32_t N, int32_t H, int32_t W,
                         int32_t C, int32_t ksize, int32_t stride, int32_t pad, void* stream) {
    HS_REQUIRE(dy && dx && idx, "maxpool_bwd: null argument");
    const int P = (H + 2 * pad - ksize) / stride + 1, Q = (W + 2 * pad - ksize) / stride + 1;
    return DISPATCH_T(dtype, maxpool_bwd_t, dy, idx, dx, N, H, W, C, P, Q, ksize, stride, pad, (hipStream_t)stream);
}
hs_status hs_mean_tokens_fwd(int32_t dtype, const void* x, void* y, int32_t B, int32_t Nt, int32_t H, int32_t out_f32,
                             void* stream) {
    HS_REQUIRE(x && y && B > 0 && Nt > 0, "mean_tokens: bad argument");
    return DISPATCH_T(dtype, mean_tokens_t, x, y, B, Nt, H, out_f32, (hipStream_t)stream);
}
hs_status hs_mean_tokens_bwd(int32_t dtype, const void* dy, void* dx, int32_t B, int32_t Nt, int32_t H, int32_t dy_f32,
                             void* stream) {
    HS_REQUIRE(dy && dx && B > 0 && Nt > 0, "mean_tokens_bwd: bad argument");
    return DISPATCH_T(dtype, mean_tokens_bwd_t, dy, dx, B, Nt, H, dy_f32, (hipStream_t)stream);
}
hs_status hs_pack_image(int32_t dtype, const float* x, void* y, int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t Hp,
                        int32_t Wp, int32_t pad, void* stream) {
    HS_REQUIRE(x && y && Cin <= 4, "pack_image: bad argument");
    const long long total = (long long)N * Hp * Wp;
    if (dtype == HS_BF16)
        hipLaunchKernelGGL(pack_image_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x,
                           (bf16_t*)y, N, Cin, H, W, Hp, Wp, pad);
    else
        hipLaunchKernelGGL(pack_image_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, (float*)y,
                           N, Cin, H, W, Hp, Wp, pad);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_pack_stem_weight(int32_t dtype, const float* w, void* out, int32_t K, int32_t R, int32_t S, int32_t Cin,
                              void* stream) {
    HS_REQUIRE(w && out && S <= 8 && Cin <= 4, "pack_stem_weight: bad argument");
    const int total = K * R * 32;
    if (dtype == HS_BF16)
        hipLaunchKernelGGL(pack_stem_weight_kernel<bf16_t>, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                           (bf16_t*)out, K, R, S, Cin);
    else
        hipLaunchKernelGGL(pack_stem_weight_kernel<float>, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                           (float*)out, K, R, S, Cin);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_unpack_stem_wgrad(const float* g, float* dw, int32_t K, int32_t R, int32_t S, int32_t Cin, void* stream) {
    HS_REQUIRE(g && dw, "unpack_stem_wgrad: null argument");
    hipLaunchKernelGGL(unpack_stem_wgrad_kernel, dim3(ceil_div(K * R * S * Cin, 256)), dim3(256), 0, (hipStream_t)stream, g,
                       dw, K, R, S, Cin);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_cast_f32_to_bf16_multi(int32_t count, const float* const* src, void* const* dst, const int64_t* n,
                                    void* stream) {
    HS_REQUIRE(count >= 0 && (count == 0 || (src && dst && n)), "cast_multi: bad argument");
    for (int base = 0; base < count; base += HS_CAST_MAX) {
        CastTable t;
        memset(&t, 0, sizeof(t));
        const int cnt = std::min(HS_CAST_MAX, count - base);
        long long mx = 0;
        for (int i = 0; i < cnt; ++i) {
            t.src[i] = src[base + i];
            t.dst[i] = dst[base + i];
            t.n[i] = n[base + i];
            mx = std::max<long long>(mx, t.n[i]);
        }
        const int gx = (int)std::min<long long>(std::max<long long>((mx / 8 + 255) / 256, 1), 256);
        hipLaunchKernelGGL(cast_multi_kernel, dim3(gx, cnt), dim3(256), 0, (hipStream_t)stream, t);
        HS_LAUNCH_CHECK();
    }
    return HS_OK;
}
hs_status hs_axpby(int32_t in_dtype, int32_t out_dtype, const void* x, const void* y, void* out, int64_t n, float a,
                   float b, void* stream) {
    HS_REQUIRE(x && out && n >= 0, "axpby: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const dim3 g(grid_for(n)), bl(256);
    if (in_dtype == HS_BF16 && out_dtype == HS_BF16)
        hipLaunchKernelGGL((axpby_kernel<bf16_t, bf16_t>), g, bl, 0, s, (const bf16_t*)x, (const bf16_t*)y, (bf16_t*)out, n, a, b);
    else if (in_dtype == HS_BF16)
        hipLaunchKernelGGL((axpby_kernel<bf16_t, float>), g, bl, 0, s, (const bf16_t*)x, (const bf16_t*)y, (float*)out, n, a, b);
    else if (out_dtype == HS_BF16)
        hipLaunchKernelGGL((axpby_kernel<float, bf16_t>), g, bl, 0, s, (const float*)x, (const float*)y, (bf16_t*)out, n, a, b);
    else
        hipLaunchKernelGGL((axpby_kernel<float, float>), g, bl, 0, s, (const float*)x, (const float*)y, (float*)out, n, a, b);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_dropout(int32_t dtype, const void* x, void* out, int64_t n, float p, uint64_t seed, void* stream) {
    HS_REQUIRE(x && out && p >= 0.f && p < 1.f, "dropout: bad argument");
    const unsigned th = dropout_thresh(p);
    const float ik = 1.f / (1.f - p);
    if (dtype == HS_BF16)
        hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                           (bf16_t*)out, n, (int)al16(x, out), th, ik, seed);
    else
        hipLaunchKernelGGL(dropout_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                           (float*)out, n, (int)al16(x, out), th, ik, seed);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_relu_fwd(int32_t dtype, const void* x, void* out, int64_t n, void* stream) {
    HS_REQUIRE(x && out, "relu: null argument");
    if (dtype == HS_BF16)
        hipLaunchKernelGGL(relu_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)out, n, (int)al16(x, out));
    else
        hipLaunchKernelGGL(relu_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)out, n, (int)al16(x, out));
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_relu_bwd(int32_t dtype, const void* dy, const void* y, void* dx, int64_t n, void* stream) {
    HS_REQUIRE(dy && y && dx, "relu_bwd: null argument");
    if (dtype == HS_BF16)
        hipLaunchKernelGGL(relu_bwd_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                           (const bf16_t*)y, (bf16_t*)dx, n, (int)al16(dy, y, dx));
    else
        hipLaunchKernelGGL(relu_bwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)dy,
                           (const float*)y, (float*)dx, n, (int)al16(dy, y, dx));
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_gelu_bwd(int32_t dtype, const void* dy, const void* u, void* dx, int64_t n, void* stream) {
    HS_REQUIRE(dy && u && dx, "gelu_bwd: null argument");
    if (dtype == HS_BF16)
        hipLaunchKernelGGL(gelu_bwd_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                           (const bf16_t*)u, (bf16_t*)dx, n, (int)al16(dy, u, dx));
    else
        hipLaunchKernelGGL(gelu_bwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)dy,
                           (const float*)u, (float*)dx, n, (int)al16(dy, u, dx));
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_mul_dev_scalar(const float* x, const float* scalar, float* out, int64_t n, void* stream) {
    HS_REQUIRE(x && scalar && out, "mul_dev_scalar: null argument");
    hipLaunchKernelGGL(mul_dev_scalar_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, scalar, out, n);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_colsum(int32_t dtype, const void* x, int64_t M, int32_t N, int32_t ld, float* out, void* ws,
                    int64_t ws_bytes, int32_t accumulate, void* stream) {
    HS_REQUIRE(x && out && M > 0 && N > 0, "colsum: bad argument");
    return DISPATCH_T(dtype, colsum_t, x, M, N, ld, out, (float*)ws, ws_bytes, accumulate, (hipStream_t)stream);
}
int64_t hs_colsum_ws_bytes(int64_t M, int32_t N) { return (int64_t)colsum_gy(M, N) * N * 4; }
hs_status hs_softmax_fwd(int32_t dtype, const float* S, const int64_t* mask, void* P, void* P_drop, int64_t rows,
                         int32_t Lk, int32_t ldS, int32_t ldP, int32_t rows_per_batch, float dropout_p, uint64_t seed,
                         void* stream) {
    HS_REQUIRE(S && P && rows > 0 && Lk > 0 && rows_per_batch > 0, "softmax: bad argument");
    return DISPATCH_T(dtype, softmax_fwd_t, S, (const long long*)mask, P, P_drop, rows, Lk, ldS, ldP, rows_per_batch,
                      dropout_p, seed, (hipStream_t)stream);
}
hs_status hs_softmax_bwd(int32_t dtype, const float* dP, const void* P, void* dS, int64_t rows, int32_t Lk, int32_t ldG,
                         int32_t ldP, float dropout_p, uint64_t seed, void* stream) {
    HS_REQUIRE(dP && P && dS && rows > 0 && Lk > 0, "softmax_bwd: bad argument");
    return DISPATCH_T(dtype, softmax_bwd_t, dP, P, dS, rows, Lk, ldG, ldP, dropout_p, seed, (hipStream_t)stream);
}
hs_status hs_bert_embed_fwd(int32_t dtype, const int64_t* ids, const float* word, const float* pos, const float* type0,
                            const float* gamma, const float* beta, void* sum_out, void* y, float* mean, float* rstd,
                            int64_t tokens, int32_t L, int32_t H, int32_t V, float eps, float dropout_p, uint64_t seed,
                            void* stream) {
    HS_REQUIRE(ids && word && pos && type0 && gamma && beta && sum_out && y && mean && rstd, "bert_embed: null argument");
    HS_REQUIRE(H % 4 == 0 && H <= 1024, "bert_embed: H=%d unsupported", H);
    const unsigned th = dropout_p > 0.f ? dropout_thresh(dropout_p) : 0u;
    const float ik = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
    const dim3 g(ceil_div(tokens, 4)), b(256);
    if (dtype == HS_BF16)
        hipLaunchKernelGGL(embed_fwd_kernel<bf16_t>, g, b, 0, (hipStream_t)stream, (const long long*)ids, word, pos, type0,
                           gamma, beta, (bf16_t*)sum_out, (bf16_t*)y, mean, rstd, tokens, L, H, V, eps, th, ik, seed);
    else
        hipLaunchKernelGGL(embed_fwd_kernel<float>, g, b, 0, (hipStream_t)stream, (const long long*)ids, word, pos, type0,
                           gamma, beta, (float*)sum_out, (float*)y, mean, rstd, tokens, L, H, V, eps, th, ik, seed);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_bert_embed_bwd(int32_t dtype, const int64_t* ids, const void* dsum, float* dword, float* dpos, int32_t B,
                            int32_t L, int32_t H, int32_t V, int32_t pad_id, void* stream) {
    HS_REQUIRE(ids && dsum, "bert_embed_bwd: null argument");
    hipStream_t s = (hipStream_t)stream;
    const long long tokens = (long long)B * L;
    if (dword) {   // caller zero-fills dword first
        const bool det = tokens <= 15360;                 // match list (4 B per token) + wave totals within 64 KiB of LDS
        const size_t lds = det ? (size_t)(tokens + 8) * sizeof(int) : 0;
        if (det && dtype == HS_BF16)
            hipLaunchKernelGGL(embed_word_bwd_kernel<bf16_t>, dim3((unsigned)tokens), dim3(256), lds, s, (const long long*)ids,
                               (const bf16_t*)dsum, dword, (int)tokens, H, V, pad_id);
        else if (det)
            hipLaunchKernelGGL(embed_word_bwd_kernel<float>, dim3((unsigned)tokens), dim3(256), lds, s, (const long long*)ids,
                               (const float*)dsum, dword, (int)tokens, H, V, pad_id);
        else if (dtype == HS_BF16)
            hipLaunchKernelGGL(embed_word_bwd_atomic_kernel<bf16_t>, dim3(grid_for(tokens * H)), dim3(256), 0, s,
                               (const long long*)ids, (const bf16_t*)dsum, dword, tokens, H, V, pad_id);
        else
            hipLaunchKernelGGL(embed_word_bwd_atomic_kernel<float>, dim3(grid_for(tokens * H)), dim3(256), 0, s,
                               (const long long*)ids, (const float*)dsum, dword, tokens, H, V, pad_id);
        HS_LAUNCH_CHECK();
    }
    if (dpos) {
        if (dtype == HS_BF16)
            hipLaunchKernelGGL(embed_pos_bwd_kernel<bf16_t>, dim3(grid_for((long long)L * H)), dim3(256), 0, s,
                               (const bf16_t*)dsum, dpos, B, L, H);
        else
            hipLaunchKernelGGL(embed_pos_bwd_kernel<float>, dim3(grid_for((long long)L * H)), dim3(256), 0, s,
                               (const float*)dsum, dpos, B, L, H);
        HS_LAUNCH_CHECK();
    }
    return HS_OK;
}
hs_status hs_transpose_bf16(const void* src, void* dst, int32_t R, int32_t Cc, int64_t ld_src, int64_t ld_dst, void* stream) {
    HS_REQUIRE(src && dst && R > 0 && Cc > 0, "transpose_bf16: bad argument");
    HS_REQUIRE(R % 8 == 0 && Cc % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0 && ld_src >= Cc && ld_dst >= R &&
                   (((uintptr_t)src | (uintptr_t)dst) & 15) == 0,
               "transpose_bf16: dims / leading dimensions must be multiples of 8 and bases 16-byte aligned");
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3(ceil_div(Cc, 64), ceil_div(R, 64)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)src, (unsigned short*)dst, R, Cc, (long long)ld_src, (long long)ld_dst);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_transpose_bf16_multi(int32_t count, const void* const* src, void* const* dst, const int32_t* R, const int32_t* Cc,
                                  const int64_t* ld_src, const int64_t* ld_dst, void* stream) {
    HS_REQUIRE(count >= 0 && (count == 0 || (src && dst && R && Cc && ld_src && ld_dst)), "transpose_bf16_multi: bad argument");
    for (int base = 0; base < count; base += HS_TRANSPOSE_MAX) {
        TransposeTable t;
        memset(&t, 0, sizeof(t));
        t.count = std::min(HS_TRANSPOSE_MAX, count - base);
        int blocks = 0;
        for (int i = 0; i < t.count; ++i) {
            const int k = base + i;
            HS_REQUIRE(src[k] && dst[k] && R[k] > 0 && Cc[k] > 0 && R[k] % 8 == 0 && Cc[k] % 8 == 0 && ld_src[k] % 8 == 0 &&
                           ld_dst[k] % 8 == 0 && ld_src[k] >= Cc[k] && ld_dst[k] >= R[k] &&
                           (((uintptr_t)src[k] | (uintptr_t)dst[k]) & 15) == 0,
                       "transpose_bf16_multi: entry %d: dims / leading dimensions must be multiples of 8 and bases 16-byte aligned", k);
            t.src[i] = (const unsigned short*)src[k];
            t.dst[i] = (unsigned short*)dst[k];
            t.R[i] = R[k]; t.C[i] = Cc[k];
            t.ld_src[i] = ld_src[k]; t.ld_dst[i] = ld_dst[k];
            t.first[i] = blocks;
            blocks += ceil_div(Cc[k], 64) * ceil_div(R[k], 64);
        }
        t.first[t.count] = blocks;
        hipLaunchKernelGGL(transpose_bf16_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t);
        HS_LAUNCH_CHECK();
    }
    return HS_OK;
}
hs_status hs_cross_entropy(const float* logits, const int64_t* labels, const float* weight, float label_smoothing,
                           int32_t B, int32_t C, float* loss, float* dlogits, float* row_loss, void* stream) {
    HS_REQUIRE(logits && labels && loss && B > 0 && C > 0, "cross_entropy: bad argument");
    const int waves = B >= 32 ? 16 : B >= 16 ? 8 : 4;
    hipLaunchKernelGGL(ce_kernel, dim3(1), dim3(64 * waves), 0, (hipStream_t)stream, logits, (const long long*)labels, weight,
                       label_smoothing, B, C, loss, dlogits, row_loss);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_sgd_step_multi(int32_t count, float* const* params, const float* const* grads, float* const* momentum_buf,
                            const int64_t* n, float lr, float momentum, float weight_decay, int32_t nesterov, int32_t first,
                            float grad_scale, void* stream) {
    HS_REQUIRE(count >= 0 && (momentum == 0.f || momentum_buf), "sgd: bad argument");
    for (int base = 0; base < count; base += HS_ADAM_MAX) {
        AdamTable t;
        memset(&t, 0, sizeof(t));
        const int cnt = std::min(HS_ADAM_MAX, count - base);
        long long mx = 0;
        for (int i = 0; i < cnt; ++i) {
            t.p[i] = params[base + i];
            t.g[i] = grads[base + i];
            t.m[i] = momentum != 0.f ? momentum_buf[base + i] : nullptr;
            t.n[i] = n[base + i];
            mx = std::max<long long>(mx, t.n[i]);
        }
        const int gx = (int)std::min<long long>(std::max<long long>((mx + 1023) / 1024, 1), 512);
        hipLaunchKernelGGL(sgd_multi_kernel, dim3(gx, cnt), dim3(256), 0, (hipStream_t)stream, t, lr, momentum, weight_decay,
                           nesterov, first, grad_scale);
        HS_LAUNCH_CHECK();
    }
    return HS_OK;
}
hs_status hs_adam_step_multi(int32_t count, float* const* params, const float* const* grads, float* const* exp_avg,
                             float* const* exp_avg_sq, const int64_t* n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int32_t step, int32_t decoupled, float grad_scale, void* stream) {
    return hs_adam_step_multi_shadow(count, params, grads, exp_avg, exp_avg_sq, nullptr, n, lr, beta1, beta2, eps, weight_decay, step,
                                     decoupled, grad_scale, stream);
}
hs_status hs_adam_step_multi_shadow(int32_t count, float* const* params, const float* const* grads, float* const* exp_avg,
                                    float* const* exp_avg_sq, void* const* bf16_shadow, const int64_t* n, float lr, float beta1,
                                    float beta2, float eps, float weight_decay, int32_t step, int32_t decoupled,
                                    float grad_scale, void* stream) {
    HS_REQUIRE(count >= 0 && step >= 1, "adam: bad argument");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    for (int base = 0; base < count; base += HS_ADAM_MAX) {
        AdamTable t;
        memset(&t, 0, sizeof(t));
        const int cnt = std::min(HS_ADAM_MAX, count - base);
        long long mx = 0;
        for (int i = 0; i < cnt; ++i) {
            t.p[i] = params[base + i];
            t.g[i] = grads[base + i];
            t.m[i] = exp_avg[base + i];
            t.v[i] = exp_avg_sq[base + i];
            t.n[i] = n[base + i];
            t.h[i] = bf16_shadow ? (bf16_t*)bf16_shadow[base + i] : nullptr;
            mx = std::max<long long>(mx, t.n[i]);
        }
        const int gx = (int)std::min<long long>(std::max<long long>((mx + 4095) / 4096, 1), 256);
        hipLaunchKernelGGL(adam_multi_kernel, dim3(gx, cnt), dim3(256), 0, (hipStream_t)stream, t, lr, beta1, beta2, eps,
                           weight_decay, bc1, bc2s, decoupled, grad_scale);
        HS_LAUNCH_CHECK();
    }
    return HS_OK;
}
}
