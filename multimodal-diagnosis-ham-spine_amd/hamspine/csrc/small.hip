// Small f32 operators at the fusion / head / loss boundary (tensors of B x H or B x C elements):
// token select, concat, products, sigmoid gating, softmax entropy, MP-Loss, focal loss, SupCon,
// centre-crop + bilinear resize.  Launch-latency bound; each is one kernel per direction.
#include <algorithm>
#include "hs_common.h"

namespace hs {

static inline int grid_for(long long n, int cap = 1024) {
    long long b = (n + 255) / 256;
    return (int)std::max<long long>(1, std::min<long long>(b, cap));
}

// out[b][:] = x[b][t][:]   (CLS pooling: reference modules/fusion_blocks.py:173)
template <typename T>
__global__ void select_token_kernel(const T* __restrict__ x, float* __restrict__ o, int B, int Nt, int H, int t) {
    const long long n = (long long)B * H;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long b = i / H;
        o[i] = to_f32(x[(b * Nt + t) * H + (i % H)]);
    }
}
template <typename T>
__global__ void select_token_bwd_kernel(const float* __restrict__ dy, T* __restrict__ dx, int B, int Nt, int H, int t) {
    const long long n = (long long)B * Nt * H;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int h = (int)(i % H);
        const long long bt = i / H;
        const int tt = (int)(bt % Nt);
        const long long b = bt / Nt;
        dx[i] = from_f32<T>(tt == t ? dy[b * H + h] : 0.f);
    }
}
// out[r] = [a[r] | b[r]]
template <typename T>
__global__ void concat2_kernel(const T* __restrict__ a, int Ha, const T* __restrict__ b, int Hb,
                               T* __restrict__ o, long long rows) {
    const int H = Ha + Hb;
    const long long n = rows * H;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long r = i / H;
        const int c = (int)(i % H);
        o[i] = c < Ha ? a[r * Ha + c] : b[r * Hb + (c - Ha)];
    }
}
template <typename T>
__global__ void split2_kernel(const T* __restrict__ g, T* __restrict__ da, int Ha, T* __restrict__ db, int Hb,
                              long long rows) {
    const int H = Ha + Hb;
    const long long n = rows * H;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long r = i / H;
        const int c = (int)(i % H);
        if (c < Ha) {
            if (da) da[r * Ha + c] = g[i];
        } else if (db) {
            db[r * Hb + (c - Ha)] = g[i];
        }
    }
}
// out = a * b  (b broadcast over rows when b_rows == 1, or per-row scalar when b_cols == 1)
__global__ void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, long long rows,
                           int cols, int b_mode) {
    const long long n = rows * cols;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float bv;
        if (b_mode == 0) bv = b[i];                 // same shape
        else if (b_mode == 1) bv = b[i / cols];     // (rows, 1)
        else bv = b[0];                             // scalar
        o[i] = a[i] * bv;
    }
}
// out[r] = sum_c a[r][c] * b[r][c]   (one wave per row)
__global__ void rowdot_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int rows,
                              int cols) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float acc = 0.f;
    for (int c = lane; c < cols; c += 64) acc += a[(long long)r * cols + c] * b[(long long)r * cols + c];
    acc = wave_sum(acc);
    if (lane == 0) o[r] = acc;
}
// KL(p || q) per row on probabilities clamped to [eps, 1] (reference mibf_net/attention.py:25-28); one wave per row.
// dp / dq (optional): gradients of sum_r w[r] * kl[r]; the clamp passes no gradient where it is active.
__global__ void kl_rows_kernel(const float* __restrict__ p, const float* __restrict__ q, float* __restrict__ o,
                               const float* __restrict__ w, float* __restrict__ dp, float* __restrict__ dq, int rows,
                               int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float acc = 0.f;
    const float wr = w ? w[r] : 0.f;
    for (int c = lane; c < cols; c += 64) {
        const long long i = (long long)r * cols + c;
        const float pr = p[i], qr = q[i];
        const float pc = fminf(fmaxf(pr, eps), 1.f), qc = fminf(fmaxf(qr, eps), 1.f);
        const float lp = __logf(pc), lq = __logf(qc);
        acc += pc * (lp - lq);
        if (dp) dp[i] = (pr > eps && pr < 1.f) ? wr * (lp - lq + 1.f) : 0.f;
        if (dq) dq[i] = (qr > eps && qr < 1.f) ? -wr * pc / qc : 0.f;
    }
    acc = wave_sum(acc);
    if (lane == 0 && o) o[r] = acc;
}
// out[0] = sum_i a[i]*b[i]  (single block, deterministic)
__global__ void dot_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, long long n) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (long long i = threadIdx.x; i < n; i += 256) acc += a[i] * (b ? b[i] : 1.f);
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) o[0] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void sigmoid_kernel(const float* __restrict__ x, float* __restrict__ o, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        o[i] = 1.f / (1.f + __expf(-x[i]));
}
// dx = dy * y * (1 - y)
__global__ void sigmoid_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ o,
                                   long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        o[i] = dy[i] * y[i] * (1.f - y[i]);
}
// ent[r] = -sum_c p*log(p + 1e-8), p = softmax(z[r])     (reference model.py:276-278)
// backward: dz_c = g[r] * p_c * (t_c - sum_k p_k t_k),  t_c = -(log(p_c+eps) + p_c/(p_c+eps))
__global__ void entropy_kernel(const float* __restrict__ z, const float* __restrict__ g, float* __restrict__ ent,
                               float* __restrict__ dz, int rows, int C) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* zr = z + (long long)r * C;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, zr[c]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += __expf(zr[c] - mx);
    const float inv = 1.f / wave_sum(se);
    float e = 0.f, pt = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float p = __expf(zr[c] - mx) * inv;
        const float lp = __logf(p + 1e-8f);
        e -= p * lp;
        pt += p * (-(lp + p / (p + 1e-8f)));
    }
    e = wave_sum(e);
    pt = wave_sum(pt);
    if (ent && lane == 0) ent[r] = e;
    if (dz) {
        const float gr = g[r];
        for (int c = lane; c < C; c += 64) {
            const float p = __expf(zr[c] - mx) * inv;
            const float t = -(__logf(p + 1e-8f) + p / (p + 1e-8f));
            dz[(long long)r * C + c] = gr * p * (t - pt);
        }
    }
}

// --------------------------------------------------------------------------------------------
// MIBF MP-Loss (reference mibf_net/model_resnet.py:76-94, mibf_net/attention.py:25-28):
//   kl_b = clamp(nan_to_num(0.5*(KL(pi||pt)+KL(pt||pi))), 0, 10), p clamped to [1e-8, 1]
//   loss = 0.3*CE(img) + 0.6*CE(txt) + 1.1*mean_b(exp(kl_b))*CE(fused)
// one block; writes the loss and the three logit gradients.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum4(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(256) void mp_loss_kernel(const float* __restrict__ zi, const float* __restrict__ zt,
                                                      const float* __restrict__ zf, const long long* __restrict__ labels,
                                                      int B, int C, float* __restrict__ loss, float* __restrict__ dzi,
                                                      float* __restrict__ dzt, float* __restrict__ dzf,
                                                      float* __restrict__ scratch /* [B] exp(kl) */) {
    __shared__ float sh[4];
    const float eps = 1e-8f;
    float ce_i = 0.f, ce_t = 0.f, ce_f = 0.f, wsum = 0.f;
    // pass 1: per-row quantities (thread per row; C is tiny)
    for (int b = threadIdx.x; b < B; b += 256) {
        const float* a = zi + (long long)b * C;
        const float* t = zt + (long long)b * C;
        const float* f = zf + (long long)b * C;
        float ma = -INFINITY, mt = -INFINITY, mf = -INFINITY;
        for (int c = 0; c < C; ++c) {
            ma = fmaxf(ma, a[c]);
            mt = fmaxf(mt, t[c]);
            mf = fmaxf(mf, f[c]);
        }
        float sa = 0.f, st = 0.f, sf = 0.f;
        for (int c = 0; c < C; ++c) {
            sa += __expf(a[c] - ma);
            st += __expf(t[c] - mt);
            sf += __expf(f[c] - mf);
        }
        const int y = (int)labels[b];
        ce_i += ma + __logf(sa) - a[y];
        ce_t += mt + __logf(st) - t[y];
        ce_f += mf + __logf(sf) - f[y];
        float kl = 0.f;
        for (int c = 0; c < C; ++c) {
            const float p = fminf(fmaxf(__expf(a[c] - ma) / sa, eps), 1.f);
            const float q = fminf(fmaxf(__expf(t[c] - mt) / st, eps), 1.f);
            kl += 0.5f * (p - q) * (__logf(p) - __logf(q));
        }
        if (kl != kl) kl = 0.f;
        kl = fminf(fmaxf(kl, 0.f), 10.f);
        const float w = __expf(kl);
        scratch[b] = w;
        wsum += w;
    }
    const float invB = 1.f / (float)B;
    ce_i = block_sum4(ce_i, sh) * invB;
    ce_t = block_sum4(ce_t, sh) * invB;
    ce_f = block_sum4(ce_f, sh) * invB;
    const float wmean = block_sum4(wsum, sh) * invB;
    if (threadIdx.x == 0) loss[0] = 0.3f * ce_i + 0.6f * ce_t + 1.1f * wmean * ce_f;
    if (!dzi) return;
    // pass 2: gradients
    for (int b = threadIdx.x; b < B; b += 256) {
        const float* a = zi + (long long)b * C;
        const float* t = zt + (long long)b * C;
        const float* f = zf + (long long)b * C;
        float ma = -INFINITY, mt = -INFINITY, mf = -INFINITY;
        for (int c = 0; c < C; ++c) {
            ma = fmaxf(ma, a[c]);
            mt = fmaxf(mt, t[c]);
            mf = fmaxf(mf, f[c]);
        }
        float sa = 0.f, st = 0.f, sf = 0.f;
        for (int c = 0; c < C; ++c) {
            sa += __expf(a[c] - ma);
            st += __expf(t[c] - mt);
            sf += __expf(f[c] - mf);
        }
        const int y = (int)labels[b];
        // recompute the raw kl to know whether the outer clamp passes gradient
        float kl = 0.f;
        for (int c = 0; c < C; ++c) {
            const float p = fminf(fmaxf(__expf(a[c] - ma) / sa, eps), 1.f);
            const float q = fminf(fmaxf(__expf(t[c] - mt) / st, eps), 1.f);
            kl += 0.5f * (p - q) * (__logf(p) - __logf(q));
        }
        const bool live = (kl == kl) && kl >= 0.f && kl <= 10.f;
        // d loss / d kl_b = 1.1 * ce_f * exp(kl_b) / B
        const float gk = live ? 1.1f * ce_f * scratch[b] * invB : 0.f;
        // softmax backprop needs sum_k p_k g_k over the unclamped entries
        float dot_p = 0.f, dot_q = 0.f;
        for (int c = 0; c < C; ++c) {
            const float pr = __expf(a[c] - ma) / sa, qr = __expf(t[c] - mt) / st;
            const float p = fminf(fmaxf(pr, eps), 1.f), q = fminf(fmaxf(qr, eps), 1.f);
            const float gp = (pr >= eps && pr <= 1.f) ? 0.5f * (__logf(p) - __logf(q) + 1.f - q / p) : 0.f;
            const float gq = (qr >= eps && qr <= 1.f) ? 0.5f * (__logf(q) - __logf(p) + 1.f - p / q) : 0.f;
            dot_p += pr * gp;
            dot_q += qr * gq;
        }
        for (int c = 0; c < C; ++c) {
            const float pr = __expf(a[c] - ma) / sa, qr = __expf(t[c] - mt) / st, fr = __expf(f[c] - mf) / sf;
            const float p = fminf(fmaxf(pr, eps), 1.f), q = fminf(fmaxf(qr, eps), 1.f);
            const float gp = (pr >= eps && pr <= 1.f) ? 0.5f * (__logf(p) - __logf(q) + 1.f - q / p) : 0.f;
            const float gq = (qr >= eps && qr <= 1.f) ? 0.5f * (__logf(q) - __logf(p) + 1.f - p / q) : 0.f;
            const float oh = c == y ? 1.f : 0.f;
            dzi[(long long)b * C + c] = 0.3f * (pr - oh) * invB + gk * pr * (gp - dot_p);
            dzt[(long long)b * C + c] = 0.6f * (qr - oh) * invB + gk * qr * (gq - dot_q);
            dzf[(long long)b * C + c] = 1.1f * wmean * (fr - oh) * invB;
        }
    }
}

// --------------------------------------------------------------------------------------------
// Focal loss (reference scripts/train.py:46-61): ce = weighted CE per row, pt = exp(-ce),
// loss = mean((1-pt)^gamma * ce).   d loss/d ce = ((1-pt)^gamma + gamma*(1-pt)^(gamma-1)*pt*ce)/B
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void focal_kernel(const float* __restrict__ z, const long long* __restrict__ labels,
                                                    const float* __restrict__ weight, float gamma, int B, int C,
                                                    float* __restrict__ loss, float* __restrict__ dz) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float* a = z + (long long)b * C;
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, a[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += __expf(a[c] - m);
        const int y = (int)labels[b];
        const float wy = weight ? weight[y] : 1.f;
        const float ce = wy * (m + __logf(s) - a[y]);
        const float pt = __expf(-ce);
        const float om = 1.f - pt;
        acc += powf(om, gamma) * ce;
        if (dz) {
            const float dce = (powf(om, gamma) + gamma * powf(om, gamma - 1.f) * pt * ce) / (float)B;
            for (int c = 0; c < C; ++c) {
                const float p = __expf(a[c] - m) / s;
                dz[(long long)b * C + c] = dce * wy * (p - (c == y ? 1.f : 0.f));
            }
        }
    }
    acc = block_sum4(acc, sh);
    if (threadIdx.x == 0) loss[0] = acc / (float)B;
}

// --------------------------------------------------------------------------------------------
// centre crop + bilinear resize (align_corners = False) on f32 NCHW; forward only (the image needs
// no gradient).  reference model.py:292-301.
// --------------------------------------------------------------------------------------------
__global__ void crop_resize_kernel(const float* __restrict__ x, float* __restrict__ o, int NC, int H, int W, int y0,
                                   int x0, int ch, int cw, int OH, int OW) {
    const long long n = (long long)NC * OH * OW;
    const float sy = (float)ch / (float)OH, sx = (float)cw / (float)OW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % OW);
        const int oy = (int)((i / OW) % OH);
        const long long nc = i / ((long long)OW * OH);
        float fy = fmaxf((oy + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf((ox + 0.5f) * sx - 0.5f, 0.f);
        const int iy0 = min((int)fy, ch - 1), ix0 = min((int)fx, cw - 1);
        const int iy1 = min(iy0 + 1, ch - 1), ix1 = min(ix0 + 1, cw - 1);
        const float ly = fy - iy0, lx = fx - ix0;
        const float* p = x + nc * H * W;
        const float v00 = p[(y0 + iy0) * W + x0 + ix0], v01 = p[(y0 + iy0) * W + x0 + ix1];
        const float v10 = p[(y0 + iy1) * W + x0 + ix0], v11 = p[(y0 + iy1) * W + x0 + ix1];
        o[i] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    }
}


// ------------------------------------------------------------------------------------------------------------
// LSTM / GRU cells of the slice-sequence encoder (reference modules/sequence_blocks.py:22-34,58-62 -> torch.nn.LSTM /
// nn.GRU).  The two matmuls of a step (x W_ih^T + b_ih for all steps at once, h W_hh^T + b_hh per step) run on the GEMM
// core; these kernels fuse the gate arithmetic.  f32, gate order as torch: LSTM i,f,g,o; GRU r,z,n.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// gx, gh: [B][4H] with row pitches ldx / ldh; c_prev [B][H]; outputs h, c [B][H] and the activated gates [B][4H]
__global__ void lstm_cell_fwd_kernel(const float* __restrict__ gx, int ldx, const float* __restrict__ gh, int ldh,
                                     const float* __restrict__ c_prev, float* __restrict__ h, float* __restrict__ c,
                                     float* __restrict__ act, int B, int H) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, j = i - b * H;
    const float* x = gx + (long long)b * ldx;
    const float* r = gh + (long long)b * ldh;
    const float ig = sigmoidf_(x[j] + r[j]);
    const float fg = sigmoidf_(x[H + j] + r[H + j]);
    const float gg = tanhf(x[2 * H + j] + r[2 * H + j]);
    const float og = sigmoidf_(x[3 * H + j] + r[3 * H + j]);
    const float cn = fg * c_prev[i] + ig * gg;
    c[i] = cn;
    h[i] = og * tanhf(cn);
    float* a = act + (long long)b * 4 * H;
    a[j] = ig; a[H + j] = fg; a[2 * H + j] = gg; a[3 * H + j] = og;
}
// dgates [B][4H] (pre-activation gradient, shared by the x and h paths), dc_prev [B][H]
__global__ void lstm_cell_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ dc_in,
                                     const float* __restrict__ act, const float* __restrict__ c_prev,
                                     const float* __restrict__ c, float* __restrict__ dgates, float* __restrict__ dc_prev,
                                     int B, int H) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, j = i - b * H;
    const float* a = act + (long long)b * 4 * H;
    const float ig = a[j], fg = a[H + j], gg = a[2 * H + j], og = a[3 * H + j];
    const float tc = tanhf(c[i]);
    const float g_h = dh ? dh[i] : 0.f;
    const float dcn = (dc_in ? dc_in[i] : 0.f) + g_h * og * (1.f - tc * tc);
    float* d = dgates + (long long)b * 4 * H;
    d[j] = dcn * gg * ig * (1.f - ig);
    d[H + j] = dcn * c_prev[i] * fg * (1.f - fg);
    d[2 * H + j] = dcn * ig * (1.f - gg * gg);
    d[3 * H + j] = g_h * tc * og * (1.f - og);
    dc_prev[i] = dcn * fg;
}
// GRU: gx, gh [B][3H]; saves r, z, n and hn = (W_hn h + b_hn) in act [B][4H]
__global__ void gru_cell_fwd_kernel(const float* __restrict__ gx, int ldx, const float* __restrict__ gh, int ldh,
                                    const float* __restrict__ h_prev, float* __restrict__ h, float* __restrict__ act, int B,
                                    int H) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, j = i - b * H;
    const float* x = gx + (long long)b * ldx;
    const float* r_ = gh + (long long)b * ldh;
    const float rg = sigmoidf_(x[j] + r_[j]);
    const float zg = sigmoidf_(x[H + j] + r_[H + j]);
    const float hn = r_[2 * H + j];
    const float ng = tanhf(x[2 * H + j] + rg * hn);
    h[i] = (1.f - zg) * ng + zg * h_prev[i];
    float* a = act + (long long)b * 4 * H;
    a[j] = rg; a[H + j] = zg; a[2 * H + j] = ng; a[3 * H + j] = hn;
}
// dgx [B][3H] (gradient of x W_ih^T + b_ih), dgh [B][3H] (gradient of h W_hh^T + b_hh), dh_prev direct part [B][H]
__global__ void gru_cell_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ act,
                                    const float* __restrict__ h_prev, float* __restrict__ dgx, float* __restrict__ dgh,
                                    float* __restrict__ dh_prev, int B, int H) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, j = i - b * H;
    const float* a = act + (long long)b * 4 * H;
    const float rg = a[j], zg = a[H + j], ng = a[2 * H + j], hn = a[3 * H + j];
    const float g = dh[i];
    const float dn = g * (1.f - zg) * (1.f - ng * ng);
    const float dz = g * (h_prev[i] - ng) * zg * (1.f - zg);
    const float dr = dn * hn * rg * (1.f - rg);
    float* dx = dgx + (long long)b * 3 * H;
    float* dhh = dgh + (long long)b * 3 * H;
    dx[j] = dr; dx[H + j] = dz; dx[2 * H + j] = dn;
    dhh[j] = dr; dhh[H + j] = dz; dhh[2 * H + j] = dn * rg;
    dh_prev[i] = g * zg;
}
}  // namespace hs

using namespace hs;

extern "C" {
hs_status hs_select_token_fwd(int32_t dtype, const void* x, float* out, int32_t B, int32_t Nt, int32_t H, int32_t t,
                              void* stream) {
    HS_REQUIRE(x && out && t >= 0 && t < Nt, "select_token: bad argument");
    if (dtype == HS_BF16)
        hipLaunchKernelGGL(select_token_kernel<bf16_t>, dim3(grid_for((long long)B * H)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)x, out, B, Nt, H, t);
    else
        hipLaunchKernelGGL(select_token_kernel<float>, dim3(grid_for((long long)B * H)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, out, B, Nt, H, t);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_select_token_bwd(int32_t dtype, const float* dy, void* dx, int32_t B, int32_t Nt, int32_t H, int32_t t,
                              void* stream) {
    HS_REQUIRE(dy && dx && t >= 0 && t < Nt, "select_token_bwd: bad argument");
    const long long n = (long long)B * Nt * H;
    if (dtype == HS_BF16)
        hipLaunchKernelGGL(select_token_bwd_kernel<bf16_t>, dim3(grid_for(n, 4096)), dim3(256), 0, (hipStream_t)stream, dy,
                           (bf16_t*)dx, B, Nt, H, t);
    else
        hipLaunchKernelGGL(select_token_bwd_kernel<float>, dim3(grid_for(n, 4096)), dim3(256), 0, (hipStream_t)stream, dy,
                           (float*)dx, B, Nt, H, t);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_concat2(const float* a, int32_t Ha, const float* b, int32_t Hb, float* out, int64_t rows, void* stream) {
    HS_REQUIRE(a && b && out, "concat2: null argument");
    hipLaunchKernelGGL(concat2_kernel<float>, dim3(grid_for(rows * (Ha + Hb))), dim3(256), 0, (hipStream_t)stream, a, Ha, b, Hb,
                       out, rows);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_concat2_t(int32_t dtype, const void* a, int32_t Ha, const void* b, int32_t Hb, void* out, int64_t rows,
                       void* stream) {
    HS_REQUIRE(a && b && out, "concat2: null argument");
    if (dtype == HS_BF16)
        hipLaunchKernelGGL(concat2_kernel<bf16_t>, dim3(grid_for(rows * (Ha + Hb))), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)a, Ha, (const bf16_t*)b, Hb, (bf16_t*)out, rows);
    else
        hipLaunchKernelGGL(concat2_kernel<float>, dim3(grid_for(rows * (Ha + Hb))), dim3(256), 0, (hipStream_t)stream,
                           (const float*)a, Ha, (const float*)b, Hb, (float*)out, rows);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_split2_t(int32_t dtype, const void* g, void* da, int32_t Ha, void* db, int32_t Hb, int64_t rows, void* stream) {
    HS_REQUIRE(g, "split2: null argument");
    if (dtype == HS_BF16)
        hipLaunchKernelGGL(split2_kernel<bf16_t>, dim3(grid_for(rows * (Ha + Hb))), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)g, (bf16_t*)da, Ha, (bf16_t*)db, Hb, rows);
    else
        hipLaunchKernelGGL(split2_kernel<float>, dim3(grid_for(rows * (Ha + Hb))), dim3(256), 0, (hipStream_t)stream,
                           (const float*)g, (float*)da, Ha, (float*)db, Hb, rows);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_split2(const float* g, float* da, int32_t Ha, float* db, int32_t Hb, int64_t rows, void* stream) {
    HS_REQUIRE(g, "split2: null argument");
    hipLaunchKernelGGL(split2_kernel<float>, dim3(grid_for(rows * (Ha + Hb))), dim3(256), 0, (hipStream_t)stream, g, da, Ha, db,
                       Hb, rows);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_mul(const float* a, const float* b, float* out, int64_t rows, int32_t cols, int32_t b_mode, void* stream) {
    HS_REQUIRE(a && b && out && b_mode >= 0 && b_mode <= 2, "mul: bad argument");
    hipLaunchKernelGGL(mul_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, (hipStream_t)stream, a, b, out, rows, cols,
                       b_mode);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_rowdot(const float* a, const float* b, float* out, int32_t rows, int32_t cols, void* stream) {
    HS_REQUIRE(a && b && out, "rowdot: null argument");
    hipLaunchKernelGGL(rowdot_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, a, b, out, rows, cols);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_kl_rows(const float* p, const float* q, float* out, const float* w, float* dp, float* dq, int32_t rows,
                     int32_t cols, float eps, void* stream) {
    HS_REQUIRE(p && q && rows > 0 && cols > 0 && (out || dp || dq) && (w || (!dp && !dq)), "kl_rows: bad argument");
    hipLaunchKernelGGL(kl_rows_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, p, q, out, w, dp, dq, rows,
                       cols, eps);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_dot(const float* a, const float* b, float* out, int64_t n, void* stream) {
    HS_REQUIRE(a && out, "dot: null argument");
    hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_sigmoid_fwd(const float* x, float* out, int64_t n, void* stream) {
    HS_REQUIRE(x && out, "sigmoid: null argument");
    hipLaunchKernelGGL(sigmoid_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, out, n);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_sigmoid_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream) {
    HS_REQUIRE(dy && y && dx, "sigmoid_bwd: null argument");
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_softmax_entropy(const float* logits, const float* g, float* ent, float* dlogits, int32_t rows, int32_t C,
                             void* stream) {
    HS_REQUIRE(logits && (ent || (g && dlogits)), "softmax_entropy: bad argument");
    hipLaunchKernelGGL(entropy_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, logits, g, ent, dlogits,
                       rows, C);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_mp_loss(const float* image_logits, const float* text_logits, const float* fused_logits, const int64_t* labels,
                     int32_t B, int32_t C, float* loss, float* d_image, float* d_text, float* d_fused, float* scratch,
                     void* stream) {
    HS_REQUIRE(image_logits && text_logits && fused_logits && labels && loss && scratch, "mp_loss: null argument");
    hipLaunchKernelGGL(mp_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, image_logits, text_logits, fused_logits,
                       (const long long*)labels, B, C, loss, d_image, d_text, d_fused, scratch);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_focal_loss(const float* logits, const int64_t* labels, const float* weight, float gamma, int32_t B, int32_t C,
                        float* loss, float* dlogits, void* stream) {
    HS_REQUIRE(logits && labels && loss, "focal_loss: null argument");
    hipLaunchKernelGGL(focal_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, (const long long*)labels, weight,
                       gamma, B, C, loss, dlogits);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_center_crop_resize(const float* x, float* out, int32_t N, int32_t Cc, int32_t H, int32_t W, int32_t y0,
                                int32_t x0, int32_t ch, int32_t cw, void* stream) {
    HS_REQUIRE(x && out && ch > 0 && cw > 0 && y0 >= 0 && x0 >= 0 && y0 + ch <= H && x0 + cw <= W,
               "center_crop_resize: bad argument");
    const long long n = (long long)N * Cc * H * W;
    hipLaunchKernelGGL(crop_resize_kernel, dim3(grid_for(n, 4096)), dim3(256), 0, (hipStream_t)stream, x, out, N * Cc, H, W,
                       y0, x0, ch, cw, H, W);
    HS_LAUNCH_CHECK();
    return HS_OK;
}

hs_status hs_lstm_cell_fwd(const float* gx, int32_t ldx, const float* gh, int32_t ldh, const float* c_prev, float* h, float* c,
                           float* act, int32_t B, int32_t H, void* stream) {
    HS_REQUIRE(gx && gh && c_prev && h && c && act && B > 0 && H > 0, "lstm_cell_fwd: bad argument");
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(ceil_div(B * H, 256)), dim3(256), 0, (hipStream_t)stream, gx, ldx, gh, ldh,
                       c_prev, h, c, act, B, H);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_lstm_cell_bwd(const float* dh, const float* dc, const float* act, const float* c_prev, const float* c,
                           float* dgates, float* dc_prev, int32_t B, int32_t H, void* stream) {
    HS_REQUIRE(act && c_prev && c && dgates && dc_prev && B > 0 && H > 0, "lstm_cell_bwd: bad argument");
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(ceil_div(B * H, 256)), dim3(256), 0, (hipStream_t)stream, dh, dc, act,
                       c_prev, c, dgates, dc_prev, B, H);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_gru_cell_fwd(const float* gx, int32_t ldx, const float* gh, int32_t ldh, const float* h_prev, float* h,
                          float* act, int32_t B, int32_t H, void* stream) {
    HS_REQUIRE(gx && gh && h_prev && h && act && B > 0 && H > 0, "gru_cell_fwd: bad argument");
    hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3(ceil_div(B * H, 256)), dim3(256), 0, (hipStream_t)stream, gx, ldx, gh, ldh,
                       h_prev, h, act, B, H);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
hs_status hs_gru_cell_bwd(const float* dh, const float* act, const float* h_prev, float* dgx, float* dgh, float* dh_prev,
                          int32_t B, int32_t H, void* stream) {
    HS_REQUIRE(dh && act && h_prev && dgx && dgh && dh_prev && B > 0 && H > 0, "gru_cell_bwd: bad argument");
    hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3(ceil_div(B * H, 256)), dim3(256), 0, (hipStream_t)stream, dh, act, h_prev,
                       dgx, dgh, dh_prev, B, H);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
}
