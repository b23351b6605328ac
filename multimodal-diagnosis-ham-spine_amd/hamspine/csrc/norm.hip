// BatchNorm (train/eval, NHWC) and LayerNorm forward/backward -- HBM-bound kernels.
// All 16-byte vector accesses, f32 statistics, deterministic two-stage column reductions.
//
// Replaces torch.nn.BatchNorm2d inside torchvision's ResNet blocks (reference encoder.py:35-42,
// mibf_net/model_resnet.py:15) and torch.nn.LayerNorm (modules/fusion_blocks.py:17-34, heads.py:36,
// transformers BertEmbeddings / BertSelfOutput / BertOutput LayerNorm, eps 1e-12).
#include <algorithm>
#include "hs_common.h"

namespace hs {

// --------------------------------------------------------------------------------------------
// generic column-reduction geometry over a row-major [M][C] matrix (C % chunk == 0)
// --------------------------------------------------------------------------------------------
struct ColGeom {
    int cg;        // chunk columns = C / EPC
    int tpc;       // threads across columns (<=256, power of two divides 256)
    int rows_par;  // 256 / tpc
    int gx, gy;    // grid
};
static ColGeom col_geom(long long M, int C, int epc) {
    ColGeom g;
    g.cg = C / epc;
    int tpc = 1;
    while (tpc < g.cg && tpc < 256) tpc <<= 1;
    g.tpc = tpc;
    g.rows_par = 256 / tpc;
    g.gx = ceil_div(g.cg, tpc);
    long long want = 2048 / g.gx;
    if (want < 1) want = 1;
    long long maxy = ceil_div(M, (long long)g.rows_par * 4);   // >= 4 rows per thread
    if (maxy < 1) maxy = 1;
    g.gy = (int)(want < maxy ? want : maxy);
    return g;
}

// ============================================================================================
// BatchNorm
// ============================================================================================
// block-level column reduction of per-thread partial sums: lanes of a wave that share a column (tx) first fold with
// xor-shuffles (offsets tpc..32), then the four waves fold through LDS; fixed order -> deterministic.  Returns the
// block total in the threads with ty == 0 (valid for every e).
template <int E, int NV>
__device__ __forceinline__ void block_col_reduce(float (&v)[NV][E], int tpc, float* sh /* [4][64*E*NV] or [256*E*NV] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (tpc < 64) {
        for (int o = 32; o >= tpc; o >>= 1) {
#pragma unroll
            for (int k = 0; k < NV; ++k)
#pragma unroll
                for (int e = 0; e < E; ++e) v[k][e] += __shfl_xor(v[k][e], o, 64);
        }
        // lanes 0..tpc-1 of every wave now hold their wave's column sums
        if (lane < tpc) {
#pragma unroll
            for (int k = 0; k < NV; ++k)
#pragma unroll
                for (int e = 0; e < E; ++e) sh[((wave * NV + k) * 64 + lane) * E + e] = v[k][e];
        }
        __syncthreads();
        if (wave == 0 && lane < tpc) {
#pragma unroll
            for (int k = 0; k < NV; ++k)
#pragma unroll
                for (int e = 0; e < E; ++e)
                    v[k][e] = (sh[((0 * NV + k) * 64 + lane) * E + e] + sh[((1 * NV + k) * 64 + lane) * E + e]) +
                              (sh[((2 * NV + k) * 64 + lane) * E + e] + sh[((3 * NV + k) * 64 + lane) * E + e]);
        }
    } else {
        // tpc in {64, 128, 256}: rows_par = 256 / tpc row lanes, one per (group of) wave(s)
        const int rows_par = 256 / tpc, tx = threadIdx.x % tpc, ty = threadIdx.x / tpc;
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int e = 0; e < E; ++e) sh[((ty * NV + k) * tpc + tx) * E + e] = v[k][e];
        __syncthreads();
        if (ty == 0) {
            for (int j = 1; j < rows_par; ++j)
#pragma unroll
                for (int k = 0; k < NV; ++k)
#pragma unroll
                    for (int e = 0; e < E; ++e) v[k][e] += sh[((j * NV + k) * tpc + tx) * E + e];
        }
    }
}

// stage 1: per (row block, channel) partials (count, mean, M2).  Sums are taken relative to a per-block, per-channel
// shift x0 (the block's first row), which keeps sum((x-x0)^2) - sum(x-x0)^2/n well conditioned without Welford's
// per-element division; four independent rows are in flight per thread.
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const T* __restrict__ x, long long M, int C, int tpc,
                                                               float* __restrict__ ws) {
    constexpr int E = Chunk<T>::N;
    __shared__ float sh[256 * E * 2];
    const int cg = C / E;
    const int tx = threadIdx.x % tpc, ty = threadIdx.x / tpc, rows_par = 256 / tpc;
    const int cc = blockIdx.x * tpc + tx;
    const bool live = cc < cg;
    const long long r0 = (long long)blockIdx.y * rows_par;
    float x0[E], acc[2][E];
#pragma unroll
    for (int e = 0; e < E; ++e) x0[e] = acc[0][e] = acc[1][e] = 0.f;
    float cnt = 0.f;
    if (live) {
        Chunk<T>::unpack(*(const u32x4*)(x + r0 * C + (long long)cc * E), x0);   // r0 < M by construction of the grid
        const long long step = (long long)gridDim.y * rows_par;
        long long r = r0 + ty;
        const T* px = x + (long long)cc * E;
        for (; r + 3 * step < M; r += 4 * step) {
            float f[4][E];
#pragma unroll
            for (int u = 0; u < 4; ++u) Chunk<T>::unpack(*(const u32x4*)(px + (r + u * step) * C), f[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const float d = f[u][e] - x0[e];
                    acc[0][e] += d;
                    acc[1][e] = fmaf(d, d, acc[1][e]);
                }
            cnt += 4.f;
        }
        for (; r < M; r += step) {
            float f[E];
            Chunk<T>::unpack(*(const u32x4*)(px + r * C), f);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float d = f[e] - x0[e];
                acc[0][e] += d;
                acc[1][e] = fmaf(d, d, acc[1][e]);
            }
            cnt += 1.f;
        }
    }
    // row count of the block: every column has the same one
    float cv[1][1] = {{cnt}};
    block_col_reduce<1, 1>(cv, tpc, sh);
    __syncthreads();
    block_col_reduce<E, 2>(acc, tpc, sh);
    if (ty == 0 && live) {
        const float n = cv[0][0];
        float* o = ws + ((long long)blockIdx.y * C + (long long)cc * E) * 3;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const float s1 = acc[0][e];
            o[e * 3 + 0] = n;
            o[e * 3 + 1] = x0[e] + s1 / n;
            o[e * 3 + 2] = fmaxf(acc[1][e] - s1 * s1 / n, 0.f);
        }
    }
}

// stage 2: merge row blocks; produce mean / invstd / (scale, shift); update running statistics.
// One wave per channel: lanes merge a strided subset of the partials (Chan's formula), then a butterfly of
// pairwise merges -- the order is fixed, so the result is deterministic.
__device__ __forceinline__ void chan_merge(float& cnt, float& mean, float& m2, float nb, float mb, float m2b) {
    if (nb > 0.f) {
        const float n = cnt + nb;
        const float d = mb - mean;
        const float w = nb / n;
        mean += d * w;
        m2 += m2b + d * d * cnt * w;
        cnt = n;
    }
}
__global__ __launch_bounds__(256) void bn_stats_final_kernel(const float* __restrict__ ws, int gy, int C, long long M,
                                                             float eps, float momentum, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             float* __restrict__ running_mean,
                                                             float* __restrict__ running_var, float* __restrict__ mean_out,
                                                             float* __restrict__ invstd_out, float* __restrict__ scale_out,
                                                             float* __restrict__ shift_out) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    float cnt = 0.f, mean = 0.f, m2 = 0.f;
    for (int b = lane; b < gy; b += 64) {
        const float* p = ws + ((long long)b * C + c) * 3;
        chan_merge(cnt, mean, m2, p[0], p[1], p[2]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float nb = __shfl_xor(cnt, o, 64), mb = __shfl_xor(mean, o, 64), qb = __shfl_xor(m2, o, 64);
        // both partners must compute the same merged value: merge (lower lane) + (upper lane) in that order
        float c0 = cnt, m0 = mean, q0 = m2, c1 = nb, m1 = mb, q1 = qb;
        if (lane & o) {
            c0 = nb; m0 = mb; q0 = qb; c1 = cnt; m1 = mean; q1 = m2;
        }
        chan_merge(c0, m0, q0, c1, m1, q1);
        cnt = c0; mean = m0; m2 = q0;
    }
    if (lane != 0) return;
    const float var = m2 / (float)M;                   // biased, used for normalisation
    const float invstd = rsqrtf(var + eps);
    mean_out[c] = mean;
    invstd_out[c] = invstd;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    scale_out[c] = g * invstd;
    shift_out[c] = b - mean * g * invstd;
    if (running_mean) {
        const float unbiased = M > 1 ? m2 / (float)(M - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

// eval mode: scale/shift from running statistics
__global__ void bn_eval_scale_kernel(int C, float eps, const float* gamma, const float* beta,
                                     const float* running_mean, const float* running_var, float* mean_out,
                                     float* invstd_out, float* scale_out, float* shift_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = rsqrtf(running_var[c] + eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    mean_out[c] = running_mean[c];
    invstd_out[c] = invstd;
    scale_out[c] = g * invstd;
    shift_out[c] = b - running_mean[c] * g * invstd;
}

// y = x*scale + shift (+ residual) (relu)
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const T* __restrict__ res,
                                                       T* __restrict__ y, long long nchunks, int cg, int relu) {
    constexpr int E = Chunk<T>::N;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (long long)gridDim.x * 256) {
        const int c0 = (int)(i % cg) * E;
        float f[E], r[E], sc[E], sh[E];
        Chunk<T>::unpack(*(const u32x4*)(x + i * E), f);
        if (res) Chunk<T>::unpack(*(const u32x4*)(res + i * E), r);
#pragma unroll
        for (int e = 0; e < E; e += 4) {
            const f32x4 a = *(const f32x4*)(scale + c0 + e), b = *(const f32x4*)(shift + c0 + e);
            sc[e] = a[0]; sc[e + 1] = a[1]; sc[e + 2] = a[2]; sc[e + 3] = a[3];
            sh[e] = b[0]; sh[e + 1] = b[1]; sh[e + 2] = b[2]; sh[e + 3] = b[3];
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            float v = fmaf(f[e], sc[e], sh[e]);        // (explicit: the backward recomputes this very value for the ReLU mask)
            if (res) v += r[e];
            if (relu) v = fmaxf(v, 0.f);
            f[e] = v;
        }
        *(u32x4*)(y + i * E) = Chunk<T>::pack(f);
    }
}

// backward stage 1: per (row block, channel) sums of dz and dz*xhat, dz = dy * (y > 0 if relu)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                             const T* __restrict__ x, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, long long M, int C,
                                                             int tpc, int relu, float* __restrict__ ws,
                                                             const float* __restrict__ scale, const float* __restrict__ shift) {
    // y == nullptr with relu: the forward output was relu(fma(x, scale, shift)) with no residual, so its sign is recomputed
    // from x (one activation read less; bf16 rounding never turns a positive f32 into 0)
    constexpr int E = Chunk<T>::N;
    const int cg = C / E;
    const int tx = threadIdx.x % tpc, ty = threadIdx.x / tpc, rows_par = 256 / tpc;
    const int cc = blockIdx.x * tpc + tx;
    const bool maskx = relu && y == nullptr;
    float acc[2][E], mu[E], is[E], sc[E], sf[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[0][e] = acc[1][e] = mu[e] = is[e] = sc[e] = sf[e] = 0.f;
    if (cc < cg) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            mu[e] = mean[cc * E + e];
            is[e] = invstd[cc * E + e];
            if (maskx) {
                sc[e] = scale[cc * E + e];
                sf[e] = shift[cc * E + e];
            }
        }
        const long long step = (long long)gridDim.y * rows_par;
        long long r = (long long)blockIdx.y * rows_par + ty;
        auto row = [&](long long rr, float (&g)[E], float (&xv)[E], float (&yv)[E]) {
            const long long o = rr * C + (long long)cc * E;
            Chunk<T>::unpack(*(const u32x4*)(dy + o), g);
            Chunk<T>::unpack(*(const u32x4*)(x + o), xv);
            if (relu) {
                if (maskx) {
#pragma unroll
                    for (int e = 0; e < E; ++e) yv[e] = fmaf(xv[e], sc[e], sf[e]);
                } else {
                    Chunk<T>::unpack(*(const u32x4*)(y + o), yv);
                }
            }
        };
        auto fold = [&](const float (&g)[E], const float (&xv)[E], const float (&yv)[E]) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float dz = (relu && !(yv[e] > 0.f)) ? 0.f : g[e];
                acc[0][e] += dz;
                acc[1][e] = fmaf(dz, (xv[e] - mu[e]) * is[e], acc[1][e]);
            }
        };
        for (; r + step < M; r += 2 * step) {
            float g0[E], x0[E], y0[E], g1[E], x1[E], y1[E];
            row(r, g0, x0, y0);
            row(r + step, g1, x1, y1);
            fold(g0, x0, y0);
            fold(g1, x1, y1);
        }
        for (; r < M; r += step) {
            float g0[E], x0[E], y0[E];
            row(r, g0, x0, y0);
            fold(g0, x0, y0);
        }
    }
    __shared__ float sh[256 * E * 2];
    block_col_reduce<E, 2>(acc, tpc, sh);
    if (ty == 0 && cc < cg) {
        float* o = ws + ((long long)blockIdx.y * C + (long long)cc * E) * 2;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            o[e * 2 + 0] = acc[0][e];
            o[e * 2 + 1] = acc[1][e];
        }
    }
}
// dx = ca*dz + cb*(x - mean) + cc per channel:  train  ca = g*is, cb = -g*is*is*dgamma/M, cc = -ca*dbeta/M
//                                               eval   ca = g*is, cb = 0, cc = 0
// (x - mean is formed first: folding mean into cc cancels catastrophically when |mean| >> std)
__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const float* __restrict__ ws, int gy, int C,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, float invM, int train,
                                                           float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                           float* __restrict__ coef) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    float a = 0.f, b = 0.f;
    for (int i = lane; i < gy; i += 64) {
        a += ws[((long long)i * C + c) * 2 + 0];
        b += ws[((long long)i * C + c) * 2 + 1];
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane != 0) return;
    dbeta[c] = a;
    dgamma[c] = b;
    const float g = gamma ? gamma[c] : 1.f, is = invstd[c];
    const float ca = g * is;
    const float cb = train ? -ca * is * b * invM : 0.f;
    const float cc = train ? -ca * a * invM : 0.f;
    coef[c] = ca;
    coef[C + c] = cb;
    coef[2 * C + c] = cc;
    coef[3 * C + c] = mean[c];
}
// backward stage 2: dx = ca*dz + cb*x + cc (coefficients from bn_bwd_final_kernel); optional dres = dz
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                           const T* __restrict__ x, const float* __restrict__ coef,
                                                           T* __restrict__ dx, T* __restrict__ dres, long long nchunks,
                                                           int cg, int C, int relu, const float* __restrict__ scale,
                                                           const float* __restrict__ shift) {
    constexpr int E = Chunk<T>::N;
    const bool maskx = relu && y == nullptr;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (long long)gridDim.x * 256) {
        const int c0 = (int)(i % cg) * E;
        float g[E], xv[E], yv[E], o[E], ca[E], cb[E], cc[E], mu[E];
        Chunk<T>::unpack(*(const u32x4*)(dy + i * E), g);
        Chunk<T>::unpack(*(const u32x4*)(x + i * E), xv);
        if (relu) {
            if (maskx) {
#pragma unroll
                for (int e = 0; e < E; ++e) yv[e] = fmaf(xv[e], scale[c0 + e], shift[c0 + e]);
            } else {
                Chunk<T>::unpack(*(const u32x4*)(y + i * E), yv);
            }
        }
#pragma unroll
        for (int e = 0; e < E; e += 4) {
            const f32x4 a = *(const f32x4*)(coef + c0 + e), b = *(const f32x4*)(coef + C + c0 + e),
                        c = *(const f32x4*)(coef + 2 * C + c0 + e), m = *(const f32x4*)(coef + 3 * C + c0 + e);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ca[e + j] = a[j];
                cb[e + j] = b[j];
                cc[e + j] = c[j];
                mu[e + j] = m[j];
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const float dz = (relu && !(yv[e] > 0.f)) ? 0.f : g[e];
            g[e] = dz;
            o[e] = ca[e] * dz + cb[e] * (xv[e] - mu[e]) + cc[e];
        }
        *(u32x4*)(dx + i * E) = Chunk<T>::pack(o);
        if (dres) *(u32x4*)(dres + i * E) = Chunk<T>::pack(g);
    }
}

template <typename T>
static int bn_fwd_t(const hs_bn_params* p, hipStream_t s) {
    constexpr int E = Chunk<T>::N;
    HS_REQUIRE(p->C % E == 0, "bn: C %% %d != 0", E);
    const long long M = p->M;
    const int C = p->C;
    if (p->training && p->stats_done) {
        // the producing GEMM finished the statistics as well (hs_gemm_params.bn_finish): scale / shift are in place
    } else if (p->training) {
        ColGeom g = col_geom(M, C, E);
        int rows = g.gy;
        if (p->partial_rows > 0) {     // the producing GEMM already left (count, mean, M2) per row tile in ws
            rows = p->partial_rows;
            HS_REQUIRE(p->ws && p->ws_bytes >= (long long)rows * C * 3 * 4, "bn: workspace smaller than the given partials");
        } else {
            HS_REQUIRE(p->ws && p->ws_bytes >= (long long)g.gy * C * 3 * 4, "bn: workspace too small");
            hipLaunchKernelGGL(bn_stats_partial_kernel<T>, dim3(g.gx, g.gy), dim3(256), 0, s, (const T*)p->x, M, C, g.tpc,
                               (float*)p->ws);
            HS_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(bn_stats_final_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, s, (const float*)p->ws, rows, C, M,
                           p->eps, p->momentum, p->gamma, p->beta, p->running_mean, p->running_var, p->save_mean,
                           p->save_invstd, p->scale, p->shift);
        HS_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(bn_eval_scale_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, s, C, p->eps, p->gamma, p->beta,
                           p->running_mean, p->running_var, p->save_mean, p->save_invstd, p->scale, p->shift);
        HS_LAUNCH_CHECK();
    }
    if (p->y) {
        const long long nch = M * C / E;
        const int blocks = (int)std::min<long long>((nch + 255) / 256, 4096);
        hipLaunchKernelGGL(bn_apply_kernel<T>, dim3(blocks), dim3(256), 0, s, (const T*)p->x, p->scale, p->shift,
                           (const T*)p->residual, (T*)p->y, nch, C / E, p->relu);
        HS_LAUNCH_CHECK();
    }
    return HS_OK;
}

template <typename T>
static int bn_bwd_t(const hs_bn_bwd_params* p, hipStream_t s) {
    constexpr int E = Chunk<T>::N;
    HS_REQUIRE(p->C % E == 0, "bn_bwd: C %% %d != 0", E);
    const long long M = p->M;
    const int C = p->C;
    ColGeom g = col_geom(M, C, E);
    if (p->partial_rows > 0) g.gy = p->partial_rows;       // the sums came with the data gradient (hs_gemm_params.bnb_partials)
    HS_REQUIRE(p->ws && p->ws_bytes >= (long long)g.gy * C * 2 * 4 + 4ll * C * 4, "bn_bwd: workspace too small");
    float* coef = (float*)p->ws + (long long)g.gy * C * 2;
    if (p->partial_rows <= 0) {
        hipLaunchKernelGGL(bn_bwd_partial_kernel<T>, dim3(g.gx, g.gy), dim3(256), 0, s, (const T*)p->dy, (const T*)p->y,
                           (const T*)p->x, p->save_mean, p->save_invstd, M, C, g.tpc, p->relu, (float*)p->ws, p->scale, p->shift);
        HS_LAUNCH_CHECK();
    }
    if (!(p->partial_rows > 0 && p->sums_done)) {          // (sums_done: the producing GEMM's last workgroups wrote dbeta / dgamma / coef)
        hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, s, (const float*)p->ws, g.gy, C, p->gamma,
                           p->save_mean, p->save_invstd, 1.f / (float)M, p->training, p->dbeta, p->dgamma, coef);
        HS_LAUNCH_CHECK();
    }
    if (p->dx) {
        const long long nch = M * C / E;
        const int blocks = (int)std::min<long long>((nch + 255) / 256, 4096);
        hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(blocks), dim3(256), 0, s, (const T*)p->dy, (const T*)p->y,
                           (const T*)p->x, coef, (T*)p->dx, (T*)p->dres, nch, C / E, C, p->relu, p->scale, p->shift);
        HS_LAUNCH_CHECK();
    }
    return HS_OK;
}

int bn_fwd(const hs_bn_params* p, hipStream_t s) {
    HS_REQUIRE(p && p->x && p->scale && p->shift && p->save_mean && p->save_invstd, "bn: null argument");
    return p->dtype == HS_BF16 ? bn_fwd_t<bf16_t>(p, s) : bn_fwd_t<float>(p, s);
}
int bn_bwd(const hs_bn_bwd_params* p, hipStream_t s) {
    HS_REQUIRE(p && p->dy && p->x && p->dbeta && p->dgamma, "bn_bwd: null argument");
    HS_REQUIRE(!p->relu || p->y || (p->scale && p->shift && !p->dres),
               "bn_bwd: a fused ReLU needs the forward output y, or (no residual) the forward's scale / shift to recompute its sign");
    return p->dtype == HS_BF16 ? bn_bwd_t<bf16_t>(p, s) : bn_bwd_t<float>(p, s);
}
long long bn_ws_bytes(long long M, int C, int dtype) {
    ColGeom g = col_geom(M, C, dtype == HS_BF16 ? 8 : 4);
    // partials (3 floats/channel/row block; up to one row per 64-row GEMM tile when the convolution supplies them) + bwd
    // coefficients
    // (+ the merge rows of a GEMM that also finishes the statistics: one per 32 tile rows, hs_gemm_bn_finish_rows)
    const long long tiles = (M + 63) / 64;
    const long long rows = std::max<long long>(g.gy, tiles + (tiles + 31) / 32);
    return rows * C * 3 * 4 + 4ll * C * 4;
}

// ============================================================================================
// LayerNorm: one wave per row, row cached in registers (H <= 64 * E * LN_MAXV)
// ============================================================================================
constexpr int LN_MAXV = 4;   // chunks per lane: H <= 64*8*4 = 2048 (bf16) / 1024 (f32)

template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     long long M, int H, float eps) {
    constexpr int E = Chunk<T>::N;
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nch = H / E;
    float v[NV][E];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            Chunk<T>::unpack(*(const u32x4*)(x + row * H + (long long)c * E), v[i]);
#pragma unroll
            for (int e = 0; e < E; ++e) sum += v[i][e];
        }
    }
    const float mean = wave_sum(sum) / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float d = v[i][e] - mean;
                sq += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)H + eps);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            float o[E];
#pragma unroll
            for (int e = 0; e < E; ++e) o[e] = (v[i][e] - mean) * rstd * gamma[c * E + e] + beta[c * E + e];
            *(u32x4*)(y + row * H + (long long)c * E) = Chunk<T>::pack(o);
        }
    }
}

// backward: each wave walks rows (stride = total waves), writes dx and keeps per-lane dgamma/dbeta
// partials; the block's four waves merge theirs through LDS into ws[block][2][H]; a second kernel reduces over blocks.
// PRE (compile time): also emit what the layer in front of this LayerNorm needs -- dxd = dx * dropout mask (the forward
// applied dropout to that layer's output before the residual add) and the column sums of dxd (its bias gradient), as
// a third partial row.
template <typename T, int NV, bool PRE>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, T* __restrict__ dx,
                                                     float* __restrict__ ws, long long M, int H, T* __restrict__ dxd,
                                                     unsigned thresh, float inv_keep, unsigned long long seed) {
    constexpr int E = Chunk<T>::N;
    constexpr int NP = PRE ? 3 : 2;          // partial rows per block: dgamma, dbeta (, bias gradient)
    float dbs[PRE ? NV : 1][E];
#pragma unroll
    for (int i = 0; i < (PRE ? NV : 1); ++i)
#pragma unroll
        for (int e = 0; e < E; ++e) dbs[i][e] = 0.f;
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const int nch = H / E;
    float dg[NV][E], db[NV][E], gm[NV][E];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < E; ++e) {
            dg[i][e] = db[i][e] = 0.f;
            const int c = lane + 64 * i;
            gm[i][e] = c < nch ? gamma[c * E + e] : 0.f;
        }
    for (long long row = wid; row < M; row += nw) {
        const float mu = mean[row], rs = rstd[row];
        float g[NV][E], xh[NV][E];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                float xv[E];
                Chunk<T>::unpack(*(const u32x4*)(dy + row * H + (long long)c * E), g[i]);
                Chunk<T>::unpack(*(const u32x4*)(x + row * H + (long long)c * E), xv);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    xh[i][e] = (xv[e] - mu) * rs;
                    dg[i][e] += g[i][e] * xh[i][e];
                    db[i][e] += g[i][e];
                    const float gg = g[i][e] * gm[i][e];
                    s1 += gg;
                    s2 += gg * xh[i][e];
                }
            }
        }
        s1 = wave_sum(s1) / (float)H;
        s2 = wave_sum(s2) / (float)H;
        if (dx || PRE) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + 64 * i;
                if (c < nch) {
                    float o[E];
#pragma unroll
                    for (int e = 0; e < E; ++e) o[e] = rs * (g[i][e] * gm[i][e] - s1 - xh[i][e] * s2);
                    if (dx) *(u32x4*)(dx + row * H + (long long)c * E) = Chunk<T>::pack(o);
                    if constexpr (PRE) {
                        if (thresh) {
#pragma unroll
                            for (int q4 = 0; q4 < E / 4; ++q4) {
                                float sc[4];
                                dropout_scale4(seed, (unsigned long long)row * H + (unsigned long long)c * E + 4 * q4, thresh, inv_keep, sc);
#pragma unroll
                                for (int k = 0; k < 4; ++k) o[4 * q4 + k] *= sc[k];
                            }
                        }
                        if (dxd) *(u32x4*)(dxd + row * H + (long long)c * E) = Chunk<T>::pack(o);
#pragma unroll
                        for (int e = 0; e < E; ++e) dbs[i][e] += o[e];
                    }
                }
            }
        }
    }
    // merge the four waves of the block in a fixed order through LDS, then one partial row per block
    extern __shared__ float ln_sh[];   // [NP][H]
    const int wv = threadIdx.x >> 6;
    for (int turn = 0; turn < 4; ++turn) {
        if (wv == turn) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + 64 * i;
                if (c < nch) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const int k = c * E + e;
                        if (turn == 0) {
                            ln_sh[k] = dg[i][e];
                            ln_sh[H + k] = db[i][e];
                            if constexpr (PRE) ln_sh[2 * H + k] = dbs[i][e];
                        } else {
                            ln_sh[k] += dg[i][e];
                            ln_sh[H + k] += db[i][e];
                            if constexpr (PRE) ln_sh[2 * H + k] += dbs[i][e];
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    for (int k = threadIdx.x; k < NP * H; k += 256) ws[(long long)blockIdx.x * NP * H + k] = ln_sh[k];
}
template <int CW>
__global__ __launch_bounds__(256) void ln_bwd_final_kernel(const float* __restrict__ ws, int nw, int H, int np,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           float* __restrict__ dbias) {
    // block = CW columns x RG = 256/CW row groups; the RG partial sums are merged through LDS in a fixed order.
    // CW = 16 gives H/16 workgroups (48 for BERT-base instead of 12 at CW = 64: the 1.5 MB of partials of a 4096-row
    // call were read by 12 CUs) with 64-byte row segments.
    // partial row w holds np (2 or 3) vectors of H floats: dgamma, dbeta(, bias gradient of the preceding layer)
    constexpr int RG = 256 / CW;
    __shared__ float sh[3][RG][CW];
    const int cl = threadIdx.x % CW, rg = threadIdx.x / CW;
    const int c = blockIdx.x * CW + cl;
    float acc[3] = {0.f, 0.f, 0.f};
    if (c < H) {
        int w = rg;
        for (; w + 3 * RG < nw; w += 4 * RG) {   // 4 independent loads in flight per accumulator
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k < np) {
                    const float v0 = ws[((long long)w * np + k) * H + c], v1 = ws[((long long)(w + RG) * np + k) * H + c];
                    const float v2 = ws[((long long)(w + 2 * RG) * np + k) * H + c];
                    const float v3 = ws[((long long)(w + 3 * RG) * np + k) * H + c];
                    acc[k] += (v0 + v1) + (v2 + v3);
                }
            }
        }
        for (; w < nw; w += RG)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (k < np) acc[k] += ws[((long long)w * np + k) * H + c];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) sh[k][rg][cl] = acc[k];
    __syncthreads();
    if (rg < 3 && rg < np && c < H) {             // row group k merges vector k
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < RG; ++r) t += sh[rg][r][cl];
        if (rg == 0) dgamma[c] = t;
        else if (rg == 1) dbeta[c] = t;
        else if (dbias) dbias[c] = t;
    }
}

static int ln_bwd_blocks(long long M) {
    long long b = (M + 15) / 16;   // >= 4 rows per wave
    if (b > 512) b = 512;          // 2 workgroups per CU
    if (b < 1) b = 1;
    return (int)b;
}

template <typename T>
static int ln_fwd_t(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                    long long M, int H, float eps, hipStream_t s) {
    constexpr int E = Chunk<T>::N;
    HS_REQUIRE(H % E == 0 && H <= 64 * E * LN_MAXV, "layernorm: H=%d unsupported for this dtype", H);
    // NV = 16-byte chunks per lane (registers scale with it: H = 768 in bf16 needs 2, not the maximum of 4)
    const int nv = ceil_div(H / E, 64);
#define LN_FWD(NVV) hipLaunchKernelGGL((ln_fwd_kernel<T, NVV>), dim3(ceil_div(M, 4)), dim3(256), 0, s, (const T*)x, gamma, beta, (T*)y, mean, rstd, M, H, eps)
    if (nv <= 1) LN_FWD(1);
    else if (nv == 2) LN_FWD(2);
    else if (nv == 3) LN_FWD(3);
    else LN_FWD(4);
#undef LN_FWD
    HS_LAUNCH_CHECK();
    return HS_OK;
}
template <typename T>
static int ln_bwd_t(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                    void* dx, float* dgamma, float* dbeta, float* ws, long long ws_bytes, long long M, int H,
                    hipStream_t s, bool pre = false, void* dxd = nullptr, float* dbias = nullptr, float p = 0.f,
                    unsigned long long seed = 0) {
    constexpr int E = Chunk<T>::N;
    HS_REQUIRE(H % E == 0 && H <= 64 * E * LN_MAXV, "layernorm_bwd: H=%d unsupported for this dtype", H);
    const int blocks = ln_bwd_blocks(M);
    const int np = pre ? 3 : 2;
    HS_REQUIRE(ws && ws_bytes >= (long long)blocks * np * H * 4, "layernorm_bwd: workspace too small");
    const int nv = ceil_div(H / E, 64);
    const unsigned th = (pre && p > 0.f) ? dropout_thresh(p) : 0u;
    const float ik = (pre && p > 0.f) ? 1.f / (1.f - p) : 1.f;
#define LN_BWD(NVV, PRE)                                                                                                  \
    hipLaunchKernelGGL((ln_bwd_kernel<T, NVV, PRE>), dim3(blocks), dim3(256), np * H * sizeof(float), s, (const T*)dy,     \
                       (const T*)x, gamma, mean, rstd, (T*)dx, ws, M, H, (T*)dxd, th, ik, seed)
    if (pre) {
        if (nv <= 1) LN_BWD(1, true);
        else if (nv == 2) LN_BWD(2, true);
        else if (nv == 3) LN_BWD(3, true);
        else LN_BWD(4, true);
    } else {
        if (nv <= 1) LN_BWD(1, false);
        else if (nv == 2) LN_BWD(2, false);
        else if (nv == 3) LN_BWD(3, false);
        else LN_BWD(4, false);
    }
#undef LN_BWD
    HS_LAUNCH_CHECK();
    hipLaunchKernelGGL(ln_bwd_final_kernel<16>, dim3(ceil_div(H, 16)), dim3(256), 0, s, ws, blocks, H, np, dgamma, dbeta, dbias);
    HS_LAUNCH_CHECK();
    return HS_OK;
}

int ln_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
           long long M, int H, float eps, hipStream_t s) {
    HS_REQUIRE(x && gamma && beta && y, "layernorm: null argument");
    return dtype == HS_BF16 ? ln_fwd_t<bf16_t>(x, gamma, beta, y, mean, rstd, M, H, eps, s)
                            : ln_fwd_t<float>(x, gamma, beta, y, mean, rstd, M, H, eps, s);
}
int ln_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
           void* dx, float* dgamma, float* dbeta, float* ws, long long ws_bytes, long long M, int H, hipStream_t s) {
    HS_REQUIRE(dy && x && gamma && mean && rstd && dgamma && dbeta, "layernorm_bwd: null argument");
    return dtype == HS_BF16 ? ln_bwd_t<bf16_t>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, ws, ws_bytes, M, H, s)
                            : ln_bwd_t<float>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, ws, ws_bytes, M, H, s);
}
// LayerNorm backward that also serves the layer in front of it: dx_dropped = dx * dropout mask(p, seed, element index)
// (p = 0: dx itself; may be NULL) and dbias = column sums of dx_dropped.
int ln_bwd_pre(int dtype, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
               float* dgamma, float* dbeta, void* dx_dropped, float* dbias, float p, unsigned long long seed, float* ws,
               long long ws_bytes, long long M, int H, hipStream_t s) {
    HS_REQUIRE(dy && x && gamma && mean && rstd && dgamma && dbeta, "layernorm_bwd_pre: null argument");
    HS_REQUIRE(p >= 0.f && p < 1.f, "layernorm_bwd_pre: bad dropout probability");
    return dtype == HS_BF16
               ? ln_bwd_t<bf16_t>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, ws, ws_bytes, M, H, s, true, dx_dropped, dbias, p, seed)
               : ln_bwd_t<float>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, ws, ws_bytes, M, H, s, true, dx_dropped, dbias, p, seed);
}
long long ln_bwd_ws_bytes(long long M, int H) { return (long long)ln_bwd_blocks(M) * 3 * H * 4; }

}  // namespace hs

extern "C" {
hs_status hs_batchnorm_fwd(const hs_bn_params* p, void* stream) { return hs::bn_fwd(p, (hipStream_t)stream); }
hs_status hs_batchnorm_bwd(const hs_bn_bwd_params* p, void* stream) { return hs::bn_bwd(p, (hipStream_t)stream); }
int64_t hs_batchnorm_ws_bytes(int64_t M, int32_t C, int32_t dtype) { return hs::bn_ws_bytes(M, C, dtype); }
hs_status hs_layernorm_fwd(int32_t dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean,
                           float* rstd, int64_t M, int32_t H, float eps, void* stream) {
    return hs::ln_fwd(dtype, x, gamma, beta, y, mean, rstd, M, H, eps, (hipStream_t)stream);
}
hs_status hs_layernorm_bwd(int32_t dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                           const float* rstd, void* dx, float* dgamma, float* dbeta, void* ws, int64_t ws_bytes,
                           int64_t M, int32_t H, void* stream) {
    return hs::ln_bwd(dtype, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, (float*)ws, ws_bytes, M, H,
                      (hipStream_t)stream);
}
hs_status hs_layernorm_bwd_pre(int32_t dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                               const float* rstd, void* dx, float* dgamma, float* dbeta, void* dx_dropped, float* dbias,
                               float dropout_p, uint64_t seed, void* ws, int64_t ws_bytes, int64_t M, int32_t H, void* stream) {
    return hs::ln_bwd_pre(dtype, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, dx_dropped, dbias, dropout_p, seed, (float*)ws,
                          ws_bytes, M, H, (hipStream_t)stream);
}
int64_t hs_layernorm_bwd_ws_bytes(int64_t M, int32_t H) { return hs::ln_bwd_ws_bytes(M, H); }
}
