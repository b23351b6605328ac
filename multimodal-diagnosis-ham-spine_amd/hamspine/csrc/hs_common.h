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
// returns scale (0 or 1/(1-p)); thresh = p * 2^32
__device__ __forceinline__ float dropout_scale(unsigned long long seed, unsigned long long idx,
                                               unsigned thresh, float inv_keep) {
    return hash_u32(seed, idx) >= thresh ? inv_keep : 0.f;
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
