// 3x3 stride-1 pad-1 convolution as a HALO-PATCH implicit GEMM (bf16, NHWC): torchvision Bottleneck.conv2 / BasicBlock.conv1-2
// forward (reference: encoder.py:96-118 -> torchvision resnet) and, with the filter transposed and mirrored by the caller,
// their data gradient.
//
// Why a second convolution body.  The generic implicit GEMM (gemm_core.h) stages an im2col tile per K tile: every input pixel
// travels L2 -> LDS nine times (once per tap), the 64 x 64 tiles that fill the chip on these small outputs (1.6-6.4 M
// elements, K = 576..4608) stage 16 bytes per 256 MACs and the launch sits at 160-200 TFLOP/s on operand bytes in flight.
// Here a workgroup owns R whole image rows (R * W <= 112 output pixels: 2 x 56, 4 x 28, 7 x 14, 7 x 7) x 64 output channels and
// walks the input channels in chunks of 64:
//   * the input patch of a chunk -- (R + 2) x (W + 2) pixels x 64 channels, the padding ring zero-filled by out-of-range
//     buffer offsets -- is staged ONCE and serves all nine taps: a tap is a constant pixel offset in the patch, so the A
//     fragment addresses are per-lane constants computed once per tile (bytes staged per MAC: 1/43 instead of 1/16);
//   * the filter slice of a (chunk, tap) is 64 x 64: each of the four waves owns 16 output channels, stages ITS 16 filter rows
//     (2 KiB) itself and reads nobody else's, so the filter ring needs no barrier at all -- slot = tap, refilled with the next
//     chunk's tap right after the wave's own reads of it, nine taps (one whole chunk, ~2 us) ahead, behind counted vmcnt waits;
//   * one barrier per chunk (the patch is shared), fragment reads of tap t + 1 issued before the MFMAs of tap t.
// Per wave and tap: 14 A + 2 B ds_read_b128 for 14 MFMAs (LDS-read bound at ~1.1 PFLOP/s chip-wide -- four waves read the same
// A fragments -- against the ~0.2 of the generic body).  The epilogue mirrors the generic one: BatchNorm (count, mean, M2)
// partial row per tile and the in-launch hand-off that finishes the statistics (bnf_handoff), rows stored 128 bytes at a time
// through an LDS transpose.
#include "gemm_launch.h"
#include "gemm_p8.h"

namespace hs {

namespace {
constexpr int C3_PBUF = 32 * 1024;                 // one patch buffer: 256 pixels x 128 bytes (8 DMA pieces per wave)
constexpr int C3_WSLOT = 2048;                     // a wave's filter rows of one tap: 16 x 128 bytes
constexpr int C3_WBASE = 2 * C3_PBUF;
constexpr int C3_STG = C3_WBASE + 4 * 9 * C3_WSLOT;  // output staging [rows][64] bf16
}  // namespace

int conv3_lds_bytes(int fm) { return C3_STG + fm * 16 * 128; }

// DMA pieces still allowed in flight when the filter slot of the NEXT tap must have landed (issue order per tap: 2 filter
// pieces, then 2 patch pieces in taps 0-3; see the accounting in DESIGN.md / the comments below)
template <int T, bool FIRST>
__device__ __forceinline__ void c3_wait_next() {
    if constexpr (T == 8) wait_vmcnt<8>();                          // next chunk's patch (last pair: tap 3) and its tap 0
    else if constexpr (FIRST) wait_vmcnt<(T < 4 ? 14 + 2 * T : 22)>();   // filter slots still come from the prologue
    else wait_vmcnt<(T < 4 ? 20 : 22)>();
}

template <int FM, bool BNS>
__global__ __launch_bounds__(256) void conv3_bf16_kernel(const GemmArgs a) {
    typedef __attribute__((address_space(3))) char lds_char;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const unsigned long long t_entry = __builtin_readcyclecounter();

    // ---- which tile: workgroups of one XCD (blockIdx % 8) take consecutive tiles, n fastest, so the column tiles that share
    // a patch meet in one L2
    int tm, tn;
    {
        const int total = a.tiles_m * a.tiles_n, q = total >> 3, r = total & 7;
        const int x = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int lin = x * q + min(x, r) + idx;
        tm = lin / a.tiles_n;
        tn = lin - tm * a.tiles_n;
    }
    const int H = a.g.H, W = a.g.W, C = a.g.C, R = a.c3_rows, PW = W + 2;
    const int img = tm / a.c3_tpi, h0 = (tm - img * a.c3_tpi) * R;
    const int tpv = min(R, H - h0) * W;                    // output pixels of this tile (rows m >= tpv are padding)
    const int npix = (R + 2) * PW;
    const int nch = C >> 6;
    const int n0 = tn * 64;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(a.A, (unsigned)min(a.a_bytes, 0x7fffff00ull));
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(a.B, (unsigned)min(a.b_bytes, 0x7fffff00ull));

    // ---- DMA source offsets.  Patch piece i of this wave fills LDS bytes [(wave * 8 + i) * 1024, + 1024) of a patch buffer:
    // patch pixel q = that / 128, physical 16-byte chunk lane % 8 = logical chunk ^ ((q >> 1) & 7) (swizzle on the source side)
    unsigned poff[8], woff[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int slot = (wave * 8 + i) * 64 + lane;
        const int q = slot >> 3, lc = (slot & 7) ^ ((q >> 1) & 7);
        const int ph = (int)fdiv((unsigned)q, a.div_qw), pw = q - ph * PW;
        const int h = h0 - 1 + ph, x = pw - 1;
        const bool ok = q < npix && h >= 0 && h < H && x >= 0 && x < W;
        poff[i] = ok ? (unsigned)(((img * H + h) * W + x) * C) * 2u + (unsigned)(lc << 4) : kOOB;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int s = j * 64 + lane, row = s >> 3, lc = (s & 7) ^ ((row >> 1) & 7);
        woff[j] = (unsigned)((n0 + wave * 16 + row) * 9 * C) * 2u + (unsigned)(lc << 4);
    }
    auto dma_patch = [&](const int c, const int buf, const int i) {          // piece i of chunk c's patch
        lds_char* dst = (lds_char*)smem + buf * C3_PBUF + (wave * 8 + i) * 1024;
        const unsigned off = c < nch ? poff[i] : kOOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)dst, 16, off, c * 128, 0, 0);
    };
    auto dma_w = [&](const int c, const int t) {                             // this wave's filter rows of (chunk c, tap t)
        lds_char* dst = (lds_char*)smem + C3_WBASE + (wave * 9 + t) * C3_WSLOT;
        const int koff = (t * C + c * 64) * 2;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned off = c < nch ? woff[j] : kOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, off, koff, 0, 0);
        }
    };

    // ---- fragment read addresses: output pixel m = (r, x) of the tile reads patch pixel (r + dr, x + ds) for tap (dr, ds)
    unsigned pa[FM][9];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = i * 16 + l15, mm = m < tpv ? m : 0;
        const int r = (int)fdiv((unsigned)mm, a.div_mw), x = mm - r * W;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int q = (r + t / 3) * PW + x + t % 3;
            pa[i][t] = (unsigned)(q * 128 + ((g ^ ((q >> 1) & 7)) << 4));
        }
    }
    const unsigned wa = (unsigned)(C3_WBASE + wave * 9 * C3_WSLOT + l15 * 128 + ((g ^ ((l15 >> 1) & 7)) << 4));

    bf16x8 af[2][FM][2], bw[2][2];
    f32x4 acc[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto reads = [&](auto setc, auto bufc, auto tc) {
        constexpr int set = decltype(setc)::value, buf = decltype(bufc)::value, t = decltype(tc)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) bw[set][ks] = *(const bf16x8*)(smem + ((wa + t * C3_WSLOT) ^ (unsigned)(ks * 64)));
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[set][i][ks] = *(const bf16x8*)(smem + buf * C3_PBUF + (pa[i][t] ^ (unsigned)(ks * 64)));
    };

    // ---- prologue: patch of chunk 0, then the nine filter slots of chunk 0 in tap order ---------------------------------------
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_patch(0, 0, i);
#pragma unroll
    for (int t = 0; t < 9; ++t) dma_w(0, t);
    HS_STAMP(1);
    if (a.stamps && threadIdx.x == 0) a.stamps[(long long)blockIdx.x * 6] = t_entry;
    wait_vmcnt<16>();                                      // patch and tap 0 have landed (taps 1-8 may be in flight)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    HS_STAMP(2);
    reads(I0{}, I0{}, I0{});

    // One chunk: nine taps.  Fragment set of (chunk parity b, tap t) = (b + t) & 1 (nine taps: the parity flips per chunk).
    auto chunk = [&](auto bc, auto firstc, const int c) {
        constexpr int b = decltype(bc)::value;
        constexpr bool first = decltype(firstc)::value;
        static_for<0, 9>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int cur = (b + t) & 1, nxt = cur ^ 1;
            // 1. what the next tap's reads need has landed (own pieces; for the shared patch the barrier makes it everyone's)
            c3_wait_next<t, first>();
            if constexpr (t == 8) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this chunk's patch reads are complete: its buffer may be refilled
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (c + 1 < nch) reads(std::integral_constant<int, nxt>{}, std::integral_constant<int, b ^ 1>{}, I0{});
            } else {
                reads(std::integral_constant<int, nxt>{}, std::integral_constant<int, b>{}, std::integral_constant<int, (t + 1) % 9>{});
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * FM + 2 > 15 ? 15 : 2 * FM + 2) : "memory");   // tap t's fragments are in registers (4-bit counter)
            }
            __builtin_amdgcn_sched_barrier(0);
            // 2. refill: this tap's filter slot with the next chunk's tap; the other patch buffer with the next chunk's patch
            dma_w(c + 1, t);
            if constexpr (t < 4) {
                dma_patch(c + 1, b ^ 1, 2 * t);
                dma_patch(c + 1, b ^ 1, 2 * t + 1);
            }
            // 3. the tap's MFMAs (operand roles swapped: a lane ends up with 4 consecutive output channels of one pixel)
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < FM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[cur][ks], af[cur][i][ks], acc[i], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        });
    };
    chunk(I0{}, std::true_type{}, 0);
    for (int c = 1; c < nch; c += 2) {
        chunk(I1{}, std::false_type{}, c);
        if (c + 1 < nch) chunk(I0{}, std::false_type{}, c + 1);
    }
    wait_vmcnt<0>();                                       // (the refills issued past the last chunk were out-of-range no-ops)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    HS_STAMP(3);

    // ---- epilogue: lane owns pixel m = i * 16 + l15, channels n0 + 16 wave + 4 g + {0..3} --------------------------------------
    const long long mbase = ((long long)img * H + h0) * W;            // tile rows are consecutive NHWC pixels
    int bnf_drawn = 0;
    if constexpr (!BNS) {
        if (a.colstats) {
            // (count, mean, M2) of this tile's rows per channel: in-lane over the row fragments, xor-shuffles over the 16 row lanes;
            // a wave owns its 16 channels, so no cross-wave merge
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const bool live = i * 16 + l15 < tpv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = live ? acc[i][e] : 0.f;
                    s1[e] += v;
                    s2[e] = fmaf(v, v, s2[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    s1[e] += __shfl_xor(s1[e], o, 64);
                    s2[e] += __shfl_xor(s2[e], o, 64);
                }
            if (l15 == 0) {
                const float cnt = (float)tpv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float* o = a.colstats + ((long long)tm * a.N + n0 + wave * 16 + 4 * g + e) * 3;
                    const float mean = s1[e] / cnt, m2 = fmaxf(s2[e] - s1[e] * s1[e] / cnt, 0.f);
                    __hip_atomic_store(o, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(o + 1, mean, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(o + 2, m2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (a.bnf_tickets) {
                const int ng = stat_groups(a.tiles_m);
                unsigned* ticket = ng ? a.bnf_tickets + a.tiles_n + tn * ng + tm / kStatGroup : a.bnf_tickets + tn;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the partial stores have left
                __syncthreads();
                if (tid == 0) bnf_drawn = (int)__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    HS_STAMP(5);
    // rows through LDS ([row][64] bf16, 16-byte chunks XOR-ed with the row so that the 16 row lanes of a write land on different
    // banks), then 128 contiguous bytes per pixel
    {
        char* stg = smem + C3_STG;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = i * 16 + l15;
            u32x2 v;
            {
                const bf16_t b0 = (bf16_t)acc[i][0], b1 = (bf16_t)acc[i][1], b2 = (bf16_t)acc[i][2], b3 = (bf16_t)acc[i][3];
                v[0] = (unsigned)__builtin_bit_cast(unsigned short, b0) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
                v[1] = (unsigned)__builtin_bit_cast(unsigned short, b2) | ((unsigned)__builtin_bit_cast(unsigned short, b3) << 16);
            }
            *(u32x2*)(stg + m * 128 + (((2 * wave + (g >> 1)) ^ (m & 7)) << 4) + (g & 1) * 8) = v;
        }
        __syncthreads();
        for (int idx = tid; idx < tpv * 8; idx += 256) {
            const int row = idx >> 3, ch = idx & 7;
            const u32x4 v = *(const u32x4*)(stg + row * 128 + ((ch ^ (row & 7)) << 4));
            *(u32x4*)(a.D + ((mbase + row) * a.ldd + n0 + ch * 8) * 2) = v;
        }
    }
    HS_STAMP(4);
    if constexpr (!BNS) {
        if (a.colstats && a.bnf_tickets) {
            __syncthreads();                               // (the hand-off's scratch is the first patch buffer: nobody reads LDS any more)
            bnf_handoff<256, 64>(a, smem, tid, tm, tn, n0, bnf_drawn);
        }
    }
}

int launch_bf16_conv3(const GemmArgs& a, int fm, dim3 grid, hipStream_t s) {
    const int lds = conv3_lds_bytes(fm);
    if (fm == 7) return launch_with_lds(conv3_bf16_kernel<7, false>, lds, lds, a, grid, s);
    if (fm == 4) return launch_with_lds(conv3_bf16_kernel<4, false>, lds, lds, a, grid, s);
    set_error("launch_bf16_conv3: no variant for %d row fragments", fm);
    return HS_ERR_ARG;
}

}  // namespace hs
