// hamspine HIP core -- shared device/host helpers (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../../include/hamspine.h"

namespace hs {

// ----------------------------------------------------------------------------------------------
// error plumbing: every C-ABI entry returns hs_status; the text goes to a thread-local buffer.
// ----------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define HS_CHECK_HIP(expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            hs::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return HS_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

#define HS_REQUIRE(cond, ...)                                                                \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            hs::set_error(__VA_ARGS__);                                                      \
            return HS_ERR_ARG;                                                               \
        }                                                                                    \
    } while (0)

#define HS_PROPAGATE(expr)                                                                   \
    do {                                                                                     \
        int _s = (expr);                                                                     \
        if (_s != HS_OK) return _s;                                                          \
    } while (0)

#define HS_LAUNCH_CHECK()                                                                    \
    do {                                                                                     \
        hipError_t _e = hipGetLastError();                                                   \
        if (_e != hipSuccess) {                                                              \
            hs::set_error("%s:%d: kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
            return HS_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

// ----------------------------------------------------------------------------------------------
// element types. Activations are either bf16 (throughput mode) or f32 (exact / parity mode);
// all arithmetic accumulates in f32.
// ----------------------------------------------------------------------------------------------
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
    static constexpr int kChunk = 4;   // elements per 16-byte chunk
    static constexpr int kDtype = HS_F32;
};
template <> struct ElemTraits<bf16_t> {
    static constexpr int kChunk = 8;
    static constexpr int kDtype = HS_BF16;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// unpack a 16-byte chunk into floats / pack back
template <typename T> struct Chunk;
template <> struct Chunk<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void unpack(const u32x4& c, float* f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(c[i]);
    }
    static __device__ __forceinline__ u32x4 pack(const float* f) {
        u32x4 c;
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = __float_as_uint(f[i]);
        return c;
    }
};
template <> struct Chunk<bf16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void unpack(const u32x4& c, float* f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(c[i] << 16);
            f[2 * i + 1] = __uint_as_float(c[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ u32x4 pack(const float* f) {
        u32x4 c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16_t lo = (bf16_t)f[2 * i], hi = (bf16_t)f[2 * i + 1];
            unsigned short ulo = __builtin_bit_cast(unsigned short, lo);
            unsigned short uhi = __builtin_bit_cast(unsigned short, hi);
            c[i] = (unsigned)ulo | ((unsigned)uhi << 16);
        }
        return c;
    }
};

// ----------------------------------------------------------------------------------------------
// raw buffer access. Out-of-range voffsets (>= num_records) read as zero and drop stores: that is
// how tile tails and conv padding are predicated without branches.
// ----------------------------------------------------------------------------------------------
static constexpr unsigned kOOB = 0x7ffffff0u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
}
__device__ __forceinline__ unsigned buf_load4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
}
__device__ __forceinline__ unsigned short buf_load2(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0);
}

// ----------------------------------------------------------------------------------------------
// fast unsigned division by a launch-time constant (valid for n < 2^31).
// ----------------------------------------------------------------------------------------------
struct FastDiv {
    unsigned d, mul, shr;
};
inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    f.d = d;
    if (d <= 1) {
        f.mul = 0;
        f.shr = 0;
        return f;
    }
    unsigned lg = 0;
    while ((1u << lg) < d) ++lg;
    unsigned p = 31 + lg;
    unsigned long long m = ((1ull << p) + d - 1) / d;
    f.mul = (unsigned)m;
    f.shr = p - 32;
    return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) {
    return f.d <= 1 ? n : (__umulhi(n, f.mul) >> f.shr);
}

// ----------------------------------------------------------------------------------------------
// counter-based dropout RNG: keep(i) is a pure function of (seed, element index), so forward and
// backward regenerate the same mask without storing it.
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned hash_u32(unsigned long long seed, unsigned long long idx) {
    unsigned long long z = seed + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (unsigned)(z >> 32);
}
// Dropout masks come in groups of four consecutive elements: one cheap 32-bit mix yields 4 x 16 uniform bits for
// elements 4q..4q+3 (the 64-bit multiplies of hash_u32 cost ~190 issue cycles per element on the quarter-rate
// integer multiplier; this is ~35).  keep(i) = bits16(q = i / 4, i % 4) >= thresh >> 16, i.e. p is honoured to 2^-16.
__device__ __forceinline__ unsigned mix32(unsigned x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ u32x2 dropout_bits4(unsigned long long seed, unsigned long long q) {
    const unsigned s0 = (unsigned)seed, s1 = (unsigned)(seed >> 32);
    const unsigned x = (unsigned)q * 0x9E3779B1u + s0 + (unsigned)(q >> 32) * 0x85EBCA77u;
    u32x2 r;
    r[0] = mix32(x ^ s1);
    r[1] = mix32(r[0] + 0x6C8E9CF5u + s1);
    return r;
}
// scale (0 or 1/(1-p)) of one element; thresh = p * 2^32
__device__ __forceinline__ float dropout_scale(unsigned long long seed, unsigned long long idx,
                                               unsigned thresh, float inv_keep) {
    const u32x2 b = dropout_bits4(seed, idx >> 2);
    const unsigned w = (idx & 2) ? b[1] : b[0];
    const unsigned f = (idx & 1) ? (w >> 16) : (w & 0xffffu);
    return f >= (thresh >> 16) ? inv_keep : 0.f;
}
// scales of elements idx..idx+3, idx % 4 == 0
__device__ __forceinline__ void dropout_scale4(unsigned long long seed, unsigned long long idx, unsigned thresh,
                                               float inv_keep, float* sc) {
    const u32x2 b = dropout_bits4(seed, idx >> 2);
    const unsigned t = thresh >> 16;
    sc[0] = (b[0] & 0xffffu) >= t ? inv_keep : 0.f;
    sc[1] = (b[0] >> 16) >= t ? inv_keep : 0.f;
    sc[2] = (b[1] & 0xffffu) >= t ? inv_keep : 0.f;
    sc[3] = (b[1] >> 16) >= t ? inv_keep : 0.f;
}
inline unsigned dropout_thresh(float p) {
    double t = (double)p * 4294967296.0;
    if (t < 0) t = 0;
    if (t > 4294967295.0) t = 4294967295.0;
    return (unsigned)t;
}

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float kInvSqrt2Pi = 0.39894228040143267794f;
    return 0.5f * (1.f + erff(x * 0.70710678118654752440f)) + x * kInvSqrt2Pi * __expf(-0.5f * x * x);
}
// Throughput-mode (bf16 activations) GELU: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below bf16
// resolution) = one v_rcp, one v_exp and six fma instead of libm's branchy erff; e = exp(-x^2/2) is shared with the
// Gaussian term of the derivative.  The exact-f32 mode keeps erff.
__device__ __forceinline__ float erf_as_from_exp(float ax, float e) {   // erf(ax / sqrt2) for ax >= 0, e = exp(-ax^2/2)
    const float t = __frcp_rn(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    return fmaf(-p * t, e, 1.f);
}
__device__ __forceinline__ float gelu_fast(float x) {
    const float ax = fabsf(x);
    const float e = __expf(-0.5f * ax * ax);
    const float er = copysignf(erf_as_from_exp(ax, e), x);
    return 0.5f * x * (1.f + er);
}
__device__ __forceinline__ float gelu_grad_fast(float x) {
    const float ax = fabsf(x);
    const float e = __expf(-0.5f * ax * ax);
    const float er = copysignf(erf_as_from_exp(ax, e), x);
    return fmaf(x * 0.39894228040143267794f, e, 0.5f * (1.f + er));
}
template <typename T> __device__ __forceinline__ float gelu_fwd_t(float x) {
    if constexpr (sizeof(T) == 2) return gelu_fast(x);
    else return gelu_erf(x);
}
template <typename T> __device__ __forceinline__ float gelu_grad_t(float x) {
    if constexpr (sizeof(T) == 2) return gelu_grad_fast(x);
    else return gelu_erf_grad(x);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace hs
