// Callers either side of the training path: test-time-augmentation batch assembly and logit
// averaging (reference scripts/predict.py:33-42,63-70), Grad-CAM maps from stage-boundary activations / gradients
// (reference analysis_tools.py:75-96), and the input staging kernel that turns decoded u8 HWC images into the
// normalised f32 NCHW batch the stem reads (torchvision ToTensor + Normalize of the reference's data pipeline).
// All HBM-bound byte movers; one launch each.
#include <algorithm>
#include "hs_common.h"

namespace hs {

static inline int igrid(long long n, int cap = 4096) {
    long long b = (n + 255) / 256;
    return (int)std::max<long long>(1, std::min<long long>(b, cap));
}

struct TtaOps {
    int op[8];
};

// out[v][p][y][x] = x[p][src_v(y, x)] over square or (for flips) rectangular planes.
// op 0 identity, 1 flip(-1), 2 flip(-2), 3 rot90(k=1, dims=(-2,-1)): out[i][j] = in[j][W-1-i]  (needs H == W)
__global__ void tta_expand_kernel(const float* __restrict__ x, float* __restrict__ o, long long planes, int H, int W,
                                  int V, TtaOps ops) {
    const long long per = planes * H * W;
    const long long n = per * V;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int v = (int)(i / per);
        const long long r = i - (long long)v * per;
        const int xx = (int)(r % W);
        const long long t = r / W;
        const int yy = (int)(t % H);
        const long long p = t / H;
        int sy = yy, sx = xx;
        const int op = ops.op[v];
        if (op == 1) sx = W - 1 - xx;
        else if (op == 2) sy = H - 1 - yy;
        else if (op == 3) { sy = xx; sx = W - 1 - yy; }
        o[i] = x[(p * H + sy) * W + sx];
    }
}

// out[i] = mean_v x[v][i]   (torch.stack(logits_list).mean(0): sequential f32 sum in variant order, then / V)
__global__ void group_mean_kernel(const float* __restrict__ x, float* __restrict__ o, int V, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float s = x[i];
        for (int v = 1; v < V; ++v) s += x[(long long)v * n + i];
        o[i] = s / (float)V;
    }
}

template <typename T>
__global__ void repeat_kernel(const T* __restrict__ x, T* __restrict__ o, long long n, int V) {
    const long long tot = n * V;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long long)gridDim.x * 256) o[i] = x[i % n];
}

// One workgroup per image.  act / grad are [HW][C] (the NHWC memory of a (C, H, W) stage output).
//   w[c]   = mean_hw grad[hw][c]
//   cam[p] = max(0, sum_c w[c] * act[p][c]);  cam /= max(cam) when that is > 0
template <typename T>
__global__ void __launch_bounds__(256) gradcam_kernel(const T* __restrict__ act, const T* __restrict__ grad,
                                                      float* __restrict__ cam, int C, int HW) {
    extern __shared__ float sm[];          // [C] weights, then [4] wave maxima
    float* w = sm;
    float* red = sm + C;
    const int b = blockIdx.x;
    const T* a = act + (long long)b * HW * C;
    const T* g = grad + (long long)b * HW * C;
    float* out = cam + (long long)b * HW;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int p = 0; p < HW; ++p) s += to_f32(g[(long long)p * C + c]);
        w[c] = s / (float)HW;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float mx = 0.f;
    for (int p = wave; p < HW; p += 4) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += w[c] * to_f32(a[(long long)p * C + c]);
        s = wave_sum(s);
        s = fmaxf(s, 0.f);
        if (lane == 0) out[p] = s;
        mx = fmaxf(mx, s);
    }
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (mx > 0.f)
        for (int p = threadIdx.x; p < HW; p += 256) out[p] = out[p] / mx;
}

struct Norm3 {
    float mean[3], std[3];
};
// out[b][c][y][x] = (src[b][y][x][c] / 255 - mean[c]) / std[c]    (ToTensor then Normalize, both IEEE f32)
__global__ void stage_u8_kernel(const unsigned char* __restrict__ src, float* __restrict__ o, int B, int H, int W, Norm3 nm) {
    const long long hw = (long long)H * W;
    const long long n = (long long)B * hw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long b = i / hw, p = i - b * hw;
        const unsigned char* s = src + i * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = (float)s[c] / 255.f;
            o[(b * 3 + c) * hw + p] = (v - nm.mean[c]) / nm.std[c];
        }
    }
}

}  // namespace hs

using namespace hs;

extern "C" {

hs_status hs_tta_expand(const float* x, float* out, int64_t planes, int32_t H, int32_t W, const int32_t* ops, int32_t V,
                        void* stream) {
    HS_REQUIRE(x && out && ops && planes > 0 && H > 0 && W > 0 && V >= 1 && V <= 8, "tta_expand: bad argument (1..8 variants)");
    TtaOps t{};
    for (int v = 0; v < V; ++v) {
        HS_REQUIRE(ops[v] >= 0 && ops[v] <= 3, "tta_expand: op must be 0 (identity), 1 (hflip), 2 (vflip) or 3 (rot90)");
        HS_REQUIRE(ops[v] != 3 || H == W, "tta_expand: rot90 inside a fused batch needs square images");
        t.op[v] = ops[v];
    }
    hipLaunchKernelGGL(tta_expand_kernel, dim3(igrid(planes * H * W * V, 8192)), dim3(256), 0, (hipStream_t)stream, x, out,
                       (long long)planes, H, W, V, t);
    HS_LAUNCH_CHECK();
    return HS_OK;
}

hs_status hs_group_mean(const float* x, float* out, int32_t V, int64_t n, void* stream) {
    HS_REQUIRE(x && out && V >= 1 && n > 0, "group_mean: bad argument");
    hipLaunchKernelGGL(group_mean_kernel, dim3(igrid(n)), dim3(256), 0, (hipStream_t)stream, x, out, V, (long long)n);
    HS_LAUNCH_CHECK();
    return HS_OK;
}

hs_status hs_repeat(int32_t elem_bytes, const void* x, void* out, int64_t n, int32_t V, void* stream) {
    HS_REQUIRE(x && out && n > 0 && V >= 1, "repeat: bad argument");
    HS_REQUIRE(elem_bytes == 2 || elem_bytes == 4 || elem_bytes == 8, "repeat: element size must be 2, 4 or 8 bytes");
    const dim3 g(igrid(n * V));
    if (elem_bytes == 2)
        hipLaunchKernelGGL(repeat_kernel<unsigned short>, g, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x,
                           (unsigned short*)out, (long long)n, V);
    else if (elem_bytes == 4)
        hipLaunchKernelGGL(repeat_kernel<unsigned>, g, dim3(256), 0, (hipStream_t)stream, (const unsigned*)x, (unsigned*)out,
                           (long long)n, V);
    else
        hipLaunchKernelGGL(repeat_kernel<unsigned long long>, g, dim3(256), 0, (hipStream_t)stream,
                           (const unsigned long long*)x, (unsigned long long*)out, (long long)n, V);
    HS_LAUNCH_CHECK();
    return HS_OK;
}

hs_status hs_gradcam(int32_t dtype, const void* act, const void* grad, float* cam, int32_t B, int32_t C, int32_t HW,
                     void* stream) {
    HS_REQUIRE(act && grad && cam && B > 0 && C > 0 && HW > 0 && C <= 8192, "gradcam: bad argument");
    const size_t lds = (size_t)(C + 4) * sizeof(float);
    if (dtype == HS_BF16)
        hipLaunchKernelGGL(gradcam_kernel<bf16_t>, dim3(B), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)act,
                           (const bf16_t*)grad, cam, C, HW);
    else if (dtype == HS_F32)
        hipLaunchKernelGGL(gradcam_kernel<float>, dim3(B), dim3(256), lds, (hipStream_t)stream, (const float*)act,
                           (const float*)grad, cam, C, HW);
    else
        HS_REQUIRE(false, "gradcam: dtype must be f32 or bf16");
    HS_LAUNCH_CHECK();
    return HS_OK;
}

hs_status hs_stage_images_u8(const uint8_t* src, float* out, int32_t B, int32_t H, int32_t W, const float* mean,
                             const float* std, void* stream) {
    HS_REQUIRE(src && out && mean && std && B > 0 && H > 0 && W > 0, "stage_images_u8: bad argument");
    Norm3 nm;
    for (int c = 0; c < 3; ++c) {
        HS_REQUIRE(std[c] != 0.f, "stage_images_u8: std must be non-zero");
        nm.mean[c] = mean[c];
        nm.std[c] = std[c];
    }
    hipLaunchKernelGGL(stage_u8_kernel, dim3(igrid((long long)B * H * W, 8192)), dim3(256), 0, (hipStream_t)stream, src, out, B,
                       H, W, nm);
    HS_LAUNCH_CHECK();
    return HS_OK;
}

}  // extern "C"
