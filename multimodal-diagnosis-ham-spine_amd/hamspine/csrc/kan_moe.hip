// KAN (B-spline) layer features and noisy-top-k mixture-of-experts gating -- the element-wise / per-row parts
// of reference ConNexT/models/block/kan1.py:77-165 and moe.py:171-291.  The contractions themselves
// ([SiLU(x) | bases] x [base_weight | scaled spline weight]) run on the MFMA GEMM core (hs_gemm, f32).
// Also: supervised-contrastive loss (reference scripts/train.py:23-44).
#include <algorithm>
#include "hs_common.h"

namespace hs {

constexpr int KAN_MAXK = 32;   // knots per feature: grid_size + 2*order + 1 <= 32

static inline int grid_for(long long n, int cap = 2048) {
    long long b = (n + 255) / 256;
    return (int)std::max<long long>(1, std::min<long long>(b, cap));
}

// Cox-de Boor recursion exactly as kan1.py:92-103, carried in forward mode so the derivative equals what
// autograd produces for that formula: B (and dB/dx) of order `order`, nb = G + order values.
__device__ __forceinline__ void bspline_eval(float x, const float* __restrict__ g, int nk, int order, float* B, float* dB) {
    const int n0 = nk - 1;
    for (int j = 0; j < n0; ++j) {
        B[j] = (x >= g[j] && x < g[j + 1]) ? 1.f : 0.f;
        dB[j] = 0.f;
    }
    for (int k = 1; k <= order; ++k) {
        const int n = n0 - k;
        for (int j = 0; j < n; ++j) {
            const float d1 = g[j + k] - g[j], d2 = g[j + k + 1] - g[j + 1];
            const float a = (x - g[j]) / d1, c = (g[j + k + 1] - x) / d2;
            const float nb = a * B[j] + c * B[j + 1];
            const float nd = B[j] / d1 + a * dB[j] - B[j + 1] / d2 + c * dB[j + 1];
            B[j] = nb;
            dB[j] = nd;
        }
    }
}

// base activation of the layer (kan1.py:17 `base_activation`, SiLU by default; the KAN classifier head of
// modules/heads.py:108-140 selects it with `act_mode`): value and derivative
enum { KAN_ACT_SILU = 0, KAN_ACT_GELU = 1, KAN_ACT_RELU = 2, KAN_ACT_IDENTITY = 3 };
__device__ __forceinline__ float kan_act(float x, int act) {
    switch (act) {
        case KAN_ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
        case KAN_ACT_RELU: return x > 0.f ? x : 0.f;
        case KAN_ACT_IDENTITY: return x;
        default: return x / (1.f + __expf(-x));
    }
}
__device__ __forceinline__ float kan_act_grad(float x, int act) {
    switch (act) {
        case KAN_ACT_GELU:
            return 0.5f * (1.f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
        case KAN_ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case KAN_ACT_IDENTITY: return 1.f;
        default: {
            const float sg = 1.f / (1.f + __expf(-x));
            return sg * (1.f + x * (1.f - sg));
        }
    }
}

// feat[b] = [ act(x[b, :]) | bases(x[b, 0]) ... bases(x[b, in-1]) ]   (row length in*(1+nb))
__global__ void kan_features_kernel(const float* __restrict__ x, const float* __restrict__ grid, float* __restrict__ feat,
                                    long long B, int in_f, int nk, int order, int act) {
    const int nb = nk - 1 - order;
    const long long total = B * in_f;
    const int row = in_f * (1 + nb);
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const long long b = t / in_f;
        const int i = (int)(t % in_f);
        const float xv = x[t];
        float Bv[KAN_MAXK], dBv[KAN_MAXK];
        bspline_eval(xv, grid + (long long)i * nk, nk, order, Bv, dBv);
        float* o = feat + b * row;
        o[i] = kan_act(xv, act);
        for (int j = 0; j < nb; ++j) o[in_f + i * nb + j] = Bv[j];
    }
}
// dx[b,i] = dfeat[b,i]*act'(x) + sum_j dfeat[b, in + i*nb + j] * dB_j/dx
__global__ void kan_features_bwd_kernel(const float* __restrict__ x, const float* __restrict__ grid,
                                        const float* __restrict__ dfeat, float* __restrict__ dx, long long B, int in_f,
                                        int nk, int order, int act) {
    const int nb = nk - 1 - order;
    const long long total = B * in_f;
    const int row = in_f * (1 + nb);
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const long long b = t / in_f;
        const int i = (int)(t % in_f);
        const float xv = x[t];
        float Bv[KAN_MAXK], dBv[KAN_MAXK];
        bspline_eval(xv, grid + (long long)i * nk, nk, order, Bv, dBv);
        const float* g = dfeat + b * row;
        float acc = g[i] * kan_act_grad(xv, act);
        for (int j = 0; j < nb; ++j) acc += g[in_f + i * nb + j] * dBv[j];
        dx[t] = acc;
    }
}
// Wcat[o] = [ base_weight[o, :] | spline_weight[o, i, :] * scaler[o, i] ... ]
__global__ void kan_pack_weight_kernel(const float* __restrict__ base_w, const float* __restrict__ spline_w,
                                       const float* __restrict__ scaler, float* __restrict__ wcat, int out_f, int in_f,
                                       int nb) {
    const int row = in_f * (1 + nb);
    const long long total = (long long)out_f * row;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int o = (int)(t / row), c = (int)(t % row);
        float v;
        if (c < in_f) {
            v = base_w[(long long)o * in_f + c];
        } else {
            const int i = (c - in_f) / nb;
            v = spline_w[(long long)o * in_f * nb + (c - in_f)] * (scaler ? scaler[(long long)o * in_f + i] : 1.f);
        }
        wcat[t] = v;
    }
}
// gradients of the three parameter tensors from d(Wcat)
__global__ void kan_unpack_wgrad_kernel(const float* __restrict__ dwcat, const float* __restrict__ spline_w,
                                        const float* __restrict__ scaler, float* __restrict__ d_base,
                                        float* __restrict__ d_spline, float* __restrict__ d_scaler, int out_f, int in_f,
                                        int nb) {
    const int row = in_f * (1 + nb);
    const long long total = (long long)out_f * in_f;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int o = (int)(t / in_f), i = (int)(t % in_f);
        const float* g = dwcat + (long long)o * row;
        if (d_base) d_base[t] = g[i];
        const float sc = scaler ? scaler[t] : 1.f;
        float ds = 0.f;
        for (int j = 0; j < nb; ++j) {
            const float gj = g[in_f + i * nb + j];
            if (d_spline) d_spline[t * nb + j] = gj * sc;
            ds += gj * spline_w[t * nb + j];
        }
        if (d_scaler) d_scaler[t] = ds;
    }
}

// ============================================================================================
// MoE gating (moe.py:231-268), one thread per batch row, E <= 16
// ============================================================================================
constexpr int MOE_MAXE = 16;
__device__ __forceinline__ float normal_cdf(float u) { return 0.5f * (1.f + erff(u * 0.70710678118654752440f)); }
__device__ __forceinline__ float normal_pdf(float u) { return 0.39894228040143267794f * __expf(-0.5f * u * u); }
__device__ __forceinline__ float gauss(unsigned long long seed, unsigned long long idx) {   // Box-Muller on the hash RNG
    const float u1 = (hash_u32(seed, 2 * idx) + 1.f) * (1.f / 4294967296.f);
    const float u2 = hash_u32(seed, 2 * idx + 1) * (1.f / 4294967296.f);
    return sqrtf(-2.f * __logf(u1)) * __cosf(6.283185307179586f * u2);
}

// outputs per row: gates[E], p[E] (softmax), top[k+1] indices, z[E] noise, sigma[E], loadrow[E]
__global__ void moe_gate_fwd_kernel(const float* __restrict__ clean, const float* __restrict__ raw_noise,
                                    const float* __restrict__ noise, int B, int E, int k,
                                    int noisy, float noise_eps, unsigned long long seed, float* __restrict__ gates,
                                    float* __restrict__ p_out, int* __restrict__ top_out, float* __restrict__ z_out,
                                    float* __restrict__ sigma_out, float* __restrict__ loadrow) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float c[MOE_MAXE], h[MOE_MAXE], sg[MOE_MAXE], z[MOE_MAXE], p[MOE_MAXE];
    for (int e = 0; e < E; ++e) {
        c[e] = clean[(long long)b * E + e];
        if (noisy) {
            const float r = raw_noise[(long long)b * E + e];
            sg[e] = (r > 20.f ? r : log1pf(__expf(r))) + noise_eps;   // softplus (torch threshold 20)
            // the standard-normal draw: the caller's tensor when it supplies one (a recorded torch.randn_like draw: the
            // reference's noise, moe.py:247), else the counter RNG
            z[e] = noise ? noise[(long long)b * E + e] : gauss(seed, (unsigned long long)b * E + e);
            h[e] = c[e] + z[e] * sg[e];
        } else {
            sg[e] = 0.f;
            z[e] = 0.f;
            h[e] = c[e];
        }
    }
    float mx = -INFINITY, se = 0.f;
    for (int e = 0; e < E; ++e) mx = fmaxf(mx, h[e]);
    for (int e = 0; e < E; ++e) {
        p[e] = __expf(h[e] - mx);
        se += p[e];
    }
    for (int e = 0; e < E; ++e) p[e] /= se;
    // top-(k+1) by selection (ties: lowest index first, as torch.topk on CPU for distinct values)
    const int m = min(k + 1, E);
    int top[MOE_MAXE];
    bool used[MOE_MAXE];
    for (int e = 0; e < E; ++e) used[e] = false;
    for (int i = 0; i < m; ++i) {
        int best = -1;
        for (int e = 0; e < E; ++e)
            if (!used[e] && (best < 0 || p[e] > p[best])) best = e;
        used[best] = true;
        top[i] = best;
    }
    float S = 0.f;
    for (int i = 0; i < k; ++i) S += p[top[i]];
    float gt[MOE_MAXE];
    for (int e = 0; e < E; ++e) gt[e] = 0.f;
    for (int i = 0; i < k; ++i) gt[top[i]] = p[top[i]] / (S + 1e-6f);
    for (int e = 0; e < E; ++e) {
        gates[(long long)b * E + e] = gt[e];
        p_out[(long long)b * E + e] = p[e];
        z_out[(long long)b * E + e] = z[e];
        sigma_out[(long long)b * E + e] = sg[e];
    }
    for (int i = 0; i < m; ++i) top_out[(long long)b * (MOE_MAXE + 1) + i] = top[i];
    if (noisy && k < E) {
        const float thr_in = p[top[k]], thr_out = p[top[k - 1]];
        for (int e = 0; e < E; ++e) {
            const bool is_in = h[e] > thr_in;
            loadrow[(long long)b * E + e] = normal_cdf((c[e] - (is_in ? thr_in : thr_out)) / sg[e]);
        }
    } else {
        for (int e = 0; e < E; ++e) loadrow[(long long)b * E + e] = gt[e] > 0.f ? 1.f : 0.f;
    }
}
// importance / load sums, cv^2 loss and its derivative w.r.t. importance_e and load_e (single thread; E <= 16)
__device__ __forceinline__ float cv2(const float* x, int E, float* dx) {
    if (E == 1) {
        dx[0] = 0.f;
        return 0.f;
    }
    float mu = 0.f;
    for (int e = 0; e < E; ++e) mu += x[e];
    mu /= E;
    float var = 0.f;
    for (int e = 0; e < E; ++e) var += (x[e] - mu) * (x[e] - mu);
    var /= (E - 1);
    const float den = mu * mu + 1e-10f;
    for (int e = 0; e < E; ++e) dx[e] = 2.f * (x[e] - mu) / (E - 1) / den - var * (2.f * mu / E) / (den * den);
    return var / den;
}
__global__ void moe_aux_kernel(const float* __restrict__ gates, const float* __restrict__ loadrow, int B, int E, float coef,
                               float* __restrict__ loss, float* __restrict__ d_imp, float* __restrict__ d_load) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float imp[MOE_MAXE], ld[MOE_MAXE], di[MOE_MAXE], dl[MOE_MAXE];
    for (int e = 0; e < E; ++e) {
        imp[e] = 0.f;
        ld[e] = 0.f;
    }
    for (int b = 0; b < B; ++b)
        for (int e = 0; e < E; ++e) {
            imp[e] += gates[(long long)b * E + e];
            ld[e] += loadrow[(long long)b * E + e];
        }
    const float l = cv2(imp, E, di) + cv2(ld, E, dl);
    loss[0] = coef * l;
    for (int e = 0; e < E; ++e) {
        d_imp[e] = coef * di[e];
        d_load[e] = coef * dl[e];
    }
}
// backward of the gating for one row: dgates (from the combine) + g_loss * d_imp  ->  d clean, d raw_noise
__global__ void moe_gate_bwd_kernel(const float* __restrict__ clean, const float* __restrict__ raw_noise,
                                    const float* __restrict__ p_in, const int* __restrict__ top_in, const float* __restrict__ z_in,
                                    const float* __restrict__ sigma_in, const float* __restrict__ dgates,
                                    const float* __restrict__ g_loss, const float* __restrict__ d_imp,
                                    const float* __restrict__ d_load, int B, int E, int k, int noisy,
                                    float* __restrict__ d_clean, float* __restrict__ d_raw) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float gl = g_loss ? g_loss[0] : 0.f;
    float p[MOE_MAXE], dp[MOE_MAXE], dc[MOE_MAXE], dsg[MOE_MAXE];
    int top[MOE_MAXE + 1];
    const int m = min(k + 1, E);
    for (int i = 0; i < m; ++i) top[i] = top_in[(long long)b * (MOE_MAXE + 1) + i];
    for (int e = 0; e < E; ++e) {
        p[e] = p_in[(long long)b * E + e];
        dp[e] = 0.f;
        dc[e] = 0.f;
        dsg[e] = 0.f;
    }
    float S = 0.f;
    for (int i = 0; i < k; ++i) S += p[top[i]];
    const float inv = 1.f / (S + 1e-6f);
    float dot = 0.f;
    for (int i = 0; i < k; ++i) {
        const int e = top[i];
        dot += (dgates[(long long)b * E + e] + gl * d_imp[e]) * p[e] * inv;
    }
    for (int i = 0; i < k; ++i) {
        const int e = top[i];
        dp[e] = ((dgates[(long long)b * E + e] + gl * d_imp[e]) - dot) * inv;
    }
    if (noisy && k < E) {   // load = sum_b Phi((c - thr)/sigma), thr = p[top[k]] or p[top[k-1]]
        const float thr_in = p[top[k]], thr_out = p[top[k - 1]];
        for (int e = 0; e < E; ++e) {
            const float c = clean[(long long)b * E + e], sg = sigma_in[(long long)b * E + e];
            const float h = c + z_in[(long long)b * E + e] * sg;
            const bool is_in = h > thr_in;
            const float u = (c - (is_in ? thr_in : thr_out)) / sg;
            const float gq = gl * d_load[e] * normal_pdf(u) / sg;
            dc[e] += gq;
            dsg[e] += -gq * u;
            dp[is_in ? top[k] : top[k - 1]] -= gq;
        }
    }
    // softmax backward: dh = p * (dp - sum p dp)
    float sp = 0.f;
    for (int e = 0; e < E; ++e) sp += p[e] * dp[e];
    for (int e = 0; e < E; ++e) {
        const float dh = p[e] * (dp[e] - sp);
        dc[e] += dh;
        if (noisy) dsg[e] += dh * z_in[(long long)b * E + e];
    }
    for (int e = 0; e < E; ++e) {
        d_clean[(long long)b * E + e] = dc[e];
        if (d_raw) {
            const float r = raw_noise ? raw_noise[(long long)b * E + e] : 0.f;
            d_raw[(long long)b * E + e] = noisy ? dsg[e] / (1.f + __expf(-r)) : 0.f;
        }
    }
}

// y[b,:] = sum_e gates[b,e] * out_e[b,:]   (dense form of SparseDispatcher.combine, moe.py:86-103: rows with a
// zero gate contribute nothing, exactly as if they had not been dispatched)
struct ExpertPtrs {
    const float* p[MOE_MAXE];
};
struct ExpertPtrsMut {
    float* p[MOE_MAXE];
};
__global__ void moe_combine_kernel(const float* __restrict__ gates, const ExpertPtrs outs, float* __restrict__ y, int B, int E,
                                   int O) {
    const long long n = (long long)B * O;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long b = i / O;
        float acc = 0.f;
        for (int e = 0; e < E; ++e) acc += gates[b * E + e] * outs.p[e][i];
        y[i] = acc;
    }
}
// d out_e[b,:] = gates[b,e] * dy[b,:] ;  dgates[b,e] = dy[b,:] . out_e[b,:]   (one wave per (b,e))
__global__ void moe_combine_bwd_kernel(const float* __restrict__ gates, const ExpertPtrs outs, const float* __restrict__ dy,
                                       const ExpertPtrsMut douts, float* __restrict__ dgates, int B, int E, int O) {
    const int lane = threadIdx.x & 63;
    const int be = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (be >= B * E) return;
    const int b = be / E, e = be % E;
    const float g = gates[be];
    float acc = 0.f;
    for (int o = lane; o < O; o += 64) {
        const float d = dy[(long long)b * O + o];
        acc += d * outs.p[e][(long long)b * O + o];
        douts.p[e][(long long)b * O + o] = g * d;
    }
    acc = wave_sum(acc);
    if (lane == 0) dgates[be] = acc;
}

// ---- sparse dispatch (SparseDispatcher, moe.py:48-112): expert e works on the rows whose gate is > 0 only ------------------
// idx[e][0 .. count[e]) = rows b (ascending) with gates[b][e] > 0.  One wave per expert walks the rows 64 at a time.
__global__ void moe_dispatch_index_kernel(const float* __restrict__ gates, int B, int E, int* __restrict__ idx, int* __restrict__ count) {
    const int e = blockIdx.x, lane = threadIdx.x;
    int n = 0;
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int b = b0 + lane;
        const bool on = b < B && gates[(long long)b * E + e] > 0.f;
        const unsigned long long m = __ballot(on);
        if (on) idx[(long long)e * B + n + __popcll(m & ((1ull << lane) - 1ull))] = b;
        n += __popcll(m);
    }
    if (lane == 0) count[e] = n;
}
// dst[i][:] = src[idx[i]][:]
__global__ void rows_gather_kernel(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst, int n, int D) {
    const long long total = (long long)n * D;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / D;
        dst[i] = src[(long long)idx[r] * D + (i - r * D)];
    }
}
// dst[idx[i]][:] += scale[idx[i] * ld_scale + col] * src[i][:]   (idx unique: plain read-modify-write, deterministic)
__global__ void rows_scatter_add_kernel(float* __restrict__ dst, const int* __restrict__ idx, const float* __restrict__ scale,
                                        int ld_scale, int col, const float* __restrict__ src, int n, int D) {
    const long long total = (long long)n * D;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / D;
        const int b = idx[r];
        const float sc = scale ? scale[(long long)b * ld_scale + col] : 1.f;
        dst[(long long)b * D + (i - r * D)] += sc * src[i];
    }
}
// backward of the weighted scatter: d src[i][:] = g * dy[idx[i]][:],  dgates[idx[i]][col] = dy[idx[i]][:] . src[i][:]   (one wave per row)
__global__ void rows_scatter_add_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ idx, const float* __restrict__ gates,
                                            int E, int col, const float* __restrict__ src, float* __restrict__ dsrc,
                                            float* __restrict__ dgates, int n, int D) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const int b = idx[r];
    const float g = gates[(long long)b * E + col];
    float acc = 0.f;
    for (int o = lane; o < D; o += 64) {
        const float d = dy[(long long)b * D + o];
        acc += d * src[(long long)r * D + o];
        dsrc[(long long)r * D + o] = g * d;
    }
    acc = wave_sum(acc);
    if (lane == 0) dgates[(long long)b * E + col] = acc;
}

// ============================================================================================
// SupCon loss (scripts/train.py:23-44), single block, B <= 256, D <= 1024; loss + d(loss)/d(features)
// ============================================================================================
__global__ __launch_bounds__(256) void supcon_kernel(const float* __restrict__ f, const long long* __restrict__ labels, int B,
                                                     int D, float temperature, float* __restrict__ loss_out,
                                                     float* __restrict__ dfeat, float* __restrict__ ws /* B*D + 2*B*B + B */) {
    float* fn = ws;                 // normalised features [B][D]
    float* sim = ws + (long long)B * D;       // logits s_ij = fn_i . fn_j / T
    float* G = sim + (long long)B * B;        // d loss / d s_ij
    float* nrm = G + (long long)B * B;
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = wv; i < B; i += 4) {
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += f[(long long)i * D + d] * f[(long long)i * D + d];
        s = fmaxf(sqrtf(wave_sum(s)), 1e-12f);
        if (lane == 0) nrm[i] = s;
        for (int d = lane; d < D; d += 64) fn[(long long)i * D + d] = f[(long long)i * D + d] / s;
    }
    __syncthreads();
    for (int ij = wv; ij < B * B; ij += 4) {
        const int i = ij / B, j = ij % B;
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += fn[(long long)i * D + d] * fn[(long long)j * D + d];
        s = wave_sum(s);
        if (lane == 0) sim[ij] = s / temperature;
    }
    __syncthreads();
    float acc = 0.f;
    for (int i = tid; i < B; i += 256) {
        float mx = -INFINITY;
        for (int j = 0; j < B; ++j) mx = fmaxf(mx, sim[i * B + j]);
        float Ei = 1e-8f, P = 0.f, msum = 0.f;
        for (int j = 0; j < B; ++j)
            if (j != i) Ei += __expf(sim[i * B + j] - mx);
        const float lE = __logf(Ei);
        for (int j = 0; j < B; ++j)
            if (j != i && labels[j] == labels[i]) {
                P += 1.f;
                msum += sim[i * B + j] - mx - lE;
            }
        acc += -msum / (P + 1e-8f) / (float)B;
        const float w = 1.f / ((float)B * (P + 1e-8f));
        for (int j = 0; j < B; ++j) {
            float g = 0.f;
            if (j != i) {
                const float q = __expf(sim[i * B + j] - mx) / Ei;
                const float mk = labels[j] == labels[i] ? 1.f : 0.f;
                g = -w * (mk - P * q);
            }
            G[i * B + j] = g;
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) red[wv] = acc;
    __syncthreads();
    if (tid == 0) loss_out[0] = red[0] + red[1] + red[2] + red[3];
    if (!dfeat) return;
    // d fn_i = (1/T) sum_k (G_ik + G_ki) fn_k ; then through the normalisation
    for (int i = wv; i < B; i += 4) {
        float dotp = 0.f;
        for (int d = lane; d < D; d += 64) {
            float g = 0.f;
            for (int k = 0; k < B; ++k) g += (G[i * B + k] + G[k * B + i]) * fn[(long long)k * D + d];
            g /= temperature;
            dfeat[(long long)i * D + d] = g;   // temporarily d/d fn
            dotp += g * fn[(long long)i * D + d];
        }
        dotp = wave_sum(dotp);
        for (int d = lane; d < D; d += 64)
            dfeat[(long long)i * D + d] = (dfeat[(long long)i * D + d] - dotp * fn[(long long)i * D + d]) / nrm[i];
    }
}


// ------------------------------------------------------------------------------------------------------------
// KANLinear.regularization_loss (reference ConNexT/models/block/kan1.py:216-236):
//   l_j = mean_c |w[j][c]| over the spline coefficients;  A = sum_j l_j;  p_j = l_j / A;  E = -sum_j p_j log p_j
//   loss = ra * A + re * E;      d loss / d l_k = ra - re * (log p_k + E) / A
// One workgroup, three strided passes (deterministic); rows = out*in, C = coefficients per row.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum256(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ __launch_bounds__(256) void kan_regularization_kernel(const float* __restrict__ w, long long rows, int C, float ra,
                                                                 float re, float* __restrict__ loss, float* __restrict__ dw) {
    __shared__ float sh[4];
    const float invc = 1.f / (float)C;
    float a = 0.f, s = 0.f;
    for (long long j = threadIdx.x; j < rows; j += 256) {
        float l = 0.f;
        for (int c = 0; c < C; ++c) l += fabsf(w[j * C + c]);
        l *= invc;
        a += l;
        s += l > 0.f ? l * __logf(l) : 0.f;
    }
    const float A = block_sum256(a, sh);
    const float S = block_sum256(s, sh);          // sum l log l
    const float E = __logf(A) - S / A;
    if (threadIdx.x == 0 && loss) loss[0] = ra * A + re * E;
    if (!dw) return;
    for (long long j = threadIdx.x; j < rows; j += 256) {
        float l = 0.f;
        for (int c = 0; c < C; ++c) l += fabsf(w[j * C + c]);
        l *= invc;
        const float dl = ra - re * ((l > 0.f ? __logf(l / A) : 0.f) + E) / A;
        for (int c = 0; c < C; ++c) {
            const float v = w[j * C + c];
            dw[j * C + c] = v > 0.f ? dl * invc : (v < 0.f ? -dl * invc : 0.f);
        }
    }
}
}  // namespace hs

using namespace hs;

extern "C" {
hs_status hs_kan_features_fwd(const float* x, const float* grid, float* feat, int64_t B, int32_t in_f, int32_t grid_size,
                              int32_t order, int32_t base_act, void* stream) {
    const int nk = grid_size + 2 * order + 1;
    HS_REQUIRE(x && grid && feat && nk <= KAN_MAXK && order >= 0, "kan_features: bad argument");
    HS_REQUIRE(base_act >= KAN_ACT_SILU && base_act <= KAN_ACT_IDENTITY, "kan_features: unknown base activation %d", base_act);
    hipLaunchKernelGGL(kan_features_kernel, dim3(grid_for(B * in_f)), dim3(256), 0, (hipStream_t)stream, x, grid, feat,
                       (long long)B, in_f, nk, order, base_act);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_kan_features_bwd(const float* x, const float* grid, const float* dfeat, float* dx, int64_t B, int32_t in_f,
                              int32_t grid_size, int32_t order, int32_t base_act, void* stream) {
    const int nk = grid_size + 2 * order + 1;
    HS_REQUIRE(x && grid && dfeat && dx && nk <= KAN_MAXK, "kan_features_bwd: bad argument");
    HS_REQUIRE(base_act >= KAN_ACT_SILU && base_act <= KAN_ACT_IDENTITY, "kan_features_bwd: unknown base activation %d", base_act);
    hipLaunchKernelGGL(kan_features_bwd_kernel, dim3(grid_for(B * in_f)), dim3(256), 0, (hipStream_t)stream, x, grid, dfeat,
                       dx, (long long)B, in_f, nk, order, base_act);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_kan_pack_weight(const float* base_w, const float* spline_w, const float* scaler, float* wcat, int32_t out_f,
                             int32_t in_f, int32_t nb, void* stream) {
    HS_REQUIRE(base_w && spline_w && wcat, "kan_pack_weight: null argument");
    hipLaunchKernelGGL(kan_pack_weight_kernel, dim3(grid_for((long long)out_f * in_f * (1 + nb))), dim3(256), 0,
                       (hipStream_t)stream, base_w, spline_w, scaler, wcat, out_f, in_f, nb);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_kan_unpack_wgrad(const float* dwcat, const float* spline_w, const float* scaler, float* d_base, float* d_spline,
                              float* d_scaler, int32_t out_f, int32_t in_f, int32_t nb, void* stream) {
    HS_REQUIRE(dwcat && spline_w, "kan_unpack_wgrad: null argument");
    hipLaunchKernelGGL(kan_unpack_wgrad_kernel, dim3(grid_for((long long)out_f * in_f)), dim3(256), 0, (hipStream_t)stream,
                       dwcat, spline_w, scaler, d_base, d_spline, d_scaler, out_f, in_f, nb);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_moe_gate_fwd(const float* clean, const float* raw_noise, const float* noise, int32_t B, int32_t E, int32_t k, int32_t noisy,
                          float noise_eps, uint64_t seed, float coef, float* gates, float* p, int32_t* top, float* z,
                          float* sigma, float* loadrow, float* loss, float* d_imp, float* d_load, void* stream) {
    HS_REQUIRE(clean && gates && p && top && z && sigma && loadrow && loss && d_imp && d_load, "moe_gate_fwd: null argument");
    HS_REQUIRE(E >= 1 && E <= MOE_MAXE && k >= 1 && k <= E, "moe_gate_fwd: need 1 <= k <= E <= %d", MOE_MAXE);
    HS_REQUIRE(!noisy || raw_noise, "moe_gate_fwd: noisy gating needs raw_noise");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(moe_gate_fwd_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, s, clean, raw_noise, noise, B, E, k, noisy,
                       noise_eps, (unsigned long long)seed, gates, p, (int*)top, z, sigma, loadrow);
    HS_LAUNCH_CHECK();
    hipLaunchKernelGGL(moe_aux_kernel, dim3(1), dim3(64), 0, s, gates, loadrow, B, E, coef, loss, d_imp, d_load);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_moe_gate_bwd(const float* clean, const float* raw_noise, const float* p, const int32_t* top, const float* z,
                          const float* sigma, const float* dgates, const float* g_loss, const float* d_imp,
                          const float* d_load, int32_t B, int32_t E, int32_t k, int32_t noisy, float* d_clean, float* d_raw,
                          void* stream) {
    HS_REQUIRE(clean && p && top && z && sigma && dgates && d_imp && d_load && d_clean, "moe_gate_bwd: null argument");
    hipLaunchKernelGGL(moe_gate_bwd_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, clean, raw_noise, p,
                       (const int*)top, z, sigma, dgates, g_loss, d_imp, d_load, B, E, k, noisy, d_clean, d_raw);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_moe_combine_fwd(const float* gates, const float* const* outs, float* y, int32_t B, int32_t E, int32_t O,
                             void* stream) {
    HS_REQUIRE(gates && outs && y && E >= 1 && E <= MOE_MAXE, "moe_combine: bad argument");
    ExpertPtrs ep;
    memset(&ep, 0, sizeof(ep));
    for (int e = 0; e < E; ++e) ep.p[e] = outs[e];
    hipLaunchKernelGGL(moe_combine_kernel, dim3(grid_for((long long)B * O)), dim3(256), 0, (hipStream_t)stream, gates, ep, y,
                       B, E, O);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_moe_combine_bwd(const float* gates, const float* const* outs, const float* dy, float* const* douts,
                             float* dgates, int32_t B, int32_t E, int32_t O, void* stream) {
    HS_REQUIRE(gates && outs && dy && douts && dgates && E >= 1 && E <= MOE_MAXE, "moe_combine_bwd: bad argument");
    ExpertPtrs ep;
    ExpertPtrsMut dp;
    memset(&ep, 0, sizeof(ep));
    memset(&dp, 0, sizeof(dp));
    for (int e = 0; e < E; ++e) {
        ep.p[e] = outs[e];
        dp.p[e] = douts[e];
    }
    hipLaunchKernelGGL(moe_combine_bwd_kernel, dim3(ceil_div(B * E, 4)), dim3(256), 0, (hipStream_t)stream, gates, ep, dy, dp,
                       dgates, B, E, O);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_moe_dispatch_index(const float* gates, int32_t B, int32_t E, int32_t* idx, int32_t* count, void* stream) {
    HS_REQUIRE(gates && idx && count && B >= 1 && E >= 1 && E <= MOE_MAXE, "moe_dispatch_index: bad argument");
    hipLaunchKernelGGL(moe_dispatch_index_kernel, dim3(E), dim3(64), 0, (hipStream_t)stream, gates, B, E, (int*)idx, (int*)count);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_rows_gather(const float* src, const int32_t* idx, float* dst, int32_t n, int32_t D, void* stream) {
    if (n == 0) return HS_OK;                       // an expert without rows: nothing to move (the tensors may be empty)
    HS_REQUIRE(src && idx && dst && n > 0 && D >= 1, "rows_gather: bad argument");
    hipLaunchKernelGGL(rows_gather_kernel, dim3(grid_for((long long)n * D)), dim3(256), 0, (hipStream_t)stream, src, (const int*)idx, dst, n, D);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_rows_scatter_add(float* dst, const int32_t* idx, const float* scale, int32_t ld_scale, int32_t col, const float* src,
                              int32_t n, int32_t D, void* stream) {
    if (n == 0) return HS_OK;
    HS_REQUIRE(dst && idx && src && n > 0 && D >= 1, "rows_scatter_add: bad argument");
    hipLaunchKernelGGL(rows_scatter_add_kernel, dim3(grid_for((long long)n * D)), dim3(256), 0, (hipStream_t)stream, dst, (const int*)idx,
                       scale, ld_scale, col, src, n, D);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_rows_scatter_add_bwd(const float* dy, const int32_t* idx, const float* gates, int32_t E, int32_t col, const float* src,
                                  float* dsrc, float* dgates, int32_t n, int32_t D, void* stream) {
    if (n == 0) return HS_OK;
    HS_REQUIRE(dy && idx && gates && src && dsrc && dgates && n > 0 && D >= 1, "rows_scatter_add_bwd: bad argument");
    hipLaunchKernelGGL(rows_scatter_add_bwd_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, (hipStream_t)stream, dy, (const int*)idx, gates, E,
                       col, src, dsrc, dgates, n, D);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_supcon_loss(const float* feat, const int64_t* labels, int32_t B, int32_t D, float temperature, float* loss,
                         float* dfeat, float* ws, void* stream) {
    HS_REQUIRE(feat && labels && loss && ws && B >= 1 && B <= 256 && D >= 1, "supcon: bad argument");
    hipLaunchKernelGGL(supcon_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, feat, (const long long*)labels, B, D,
                       temperature, loss, dfeat, ws);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
int64_t hs_supcon_ws_bytes(int32_t B, int32_t D) { return ((int64_t)B * D + 2ll * B * B + B) * 4; }
hs_status hs_kan_regularization(const float* w, int64_t rows, int32_t coeffs, float reg_activation, float reg_entropy,
                                float* loss, float* dw, void* stream) {
    HS_REQUIRE(w && rows > 0 && coeffs > 0 && (loss || dw), "kan_regularization: bad argument");
    hipLaunchKernelGGL(kan_regularization_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w, rows, coeffs, reg_activation,
                       reg_entropy, loss, dw);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
}
