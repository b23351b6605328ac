// Streaming 1x1 convolution (pointwise GEMM) for the image tower: D[M][N] = T(A)[M][K] . W[N][K]^T, bf16, f32 accumulation.
//
// reference: every 1x1 convolution of torchvision's Bottleneck (encoder.py:35-58 -> resnet50 conv1 / conv3 / downsample;
// mibf_net/model_resnet.py:15) followed by train-mode BatchNorm (+ ReLU) -- at batch 32 these are M = 1.5 k .. 100 k rows
// against 64 .. 2048 channels: streams of activations against a weight matrix that fits a CU's LDS many times over.
//
// Why not the tiled GEMM bodies.  Cut into 64x64 / 128x64 output tiles such a convolution is thousands of workgroups that each
// live for one to eight K steps: set-up, first-fetch latency, statistics hand-off and epilogue dominate (every 1x1
// launch costs 20-40 us whatever its size, 0.13-0.15 of the HBM roofline).  Here instead:
//   * PERSISTENT workgroups (one or two per CU) walk 64-row blocks of A; a block's [64][K] tile is staged ONCE (A-stationary:
//     every activation byte crosses the CU boundary once) and multiplied against ALL N output channels,
//   * the weights stream through a 3-slot LDS-DMA ring as [BN][64] tiles in one continuous sequence that simply repeats per
//     row block (they are L2-resident: N*K*2 <= 2 MB), prefetched across row-block boundaries behind counted vmcnt waits,
//   * a wave owns whole output COLUMNS (all 64 rows x BN/4 columns of a chunk), so the per-channel sum / sum of squares of
//     the following BatchNorm accumulate in registers over every row block the workgroup processes: ONE (count, mean, M2)
//     partial row per workgroup, no cross-wave or cross-workgroup hand-off, deterministic,
//
// MEASURED (round 3, rocprofv3 on the C2 step, profiles/round3_pw_stream_ab.txt): the nine big-M launches take 23.5 us here
// against 30.8 us in the tiled kernel with its statistics riders -- but the partial rows then need the 6.3 us finaliser as a
// launch of its own, which the tiled kernel runs in its tail: a wash (0.65 vs 0.51 ms per step over the 18 launches it took,
// worse where N > 512).  An input-transform variant (the producing BatchNorm's apply fused into the staging through
// registers) measured 42 us on the same shapes and was removed.  The kernel stays as a tested building block, OFF in the
// executors (HAMSPINE_PW_STREAM=1 switches it on).
#include <algorithm>
#include <mutex>
#include <unordered_set>
#include "gemm_core.h"
#include "pw_stream.h"

namespace hs {

template <int N>
__device__ __forceinline__ void pw_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// s_waitcnt takes an immediate: a wave-uniform run-time count goes through a jump table (counts above 40 wait for 40)
__device__ __forceinline__ void pw_wait_vm_rt(int n) {
#define PW_W(k) case k: pw_wait_vm<k>(); break;
    switch (n) {
        PW_W(0) PW_W(1) PW_W(2) PW_W(3) PW_W(4) PW_W(5) PW_W(6) PW_W(7) PW_W(8) PW_W(9) PW_W(10) PW_W(11) PW_W(12) PW_W(13)
        PW_W(14) PW_W(15) PW_W(16) PW_W(17) PW_W(18) PW_W(19) PW_W(20) PW_W(21) PW_W(22) PW_W(23) PW_W(24) PW_W(25) PW_W(26)
        PW_W(27) PW_W(28) PW_W(29) PW_W(30) PW_W(31) PW_W(32) PW_W(33) PW_W(34) PW_W(35) PW_W(36) PW_W(37) PW_W(38) PW_W(39)
        default: pw_wait_vm<40>(); break;
    }
#undef PW_W
}

// BN: output columns per chunk (128, or 64 for N = 64); SM: statistics mode -- 0 none, 1 accumulated
// in registers over the workgroup's row blocks (N / BN <= 4 chunks; one partial row per WORKGROUP), 2 flushed per row block
// (any N; one partial row per ROW BLOCK).
template <int BN, int SM>
__global__ __launch_bounds__(256) void pw_stream_kernel(const PwArgs a) {
    constexpr int BM = 64, NSW = 3;
    constexpr int FM = 4, FN = BN / 64;                 // wave tile: 64 rows x BN/4 columns
    constexpr int WTILE = BN * 128;                     // bytes of a [BN][64] weight tile
    constexpr int NWP = BN / 32;                        // DMA pieces per wave per weight tile
    constexpr int NCMAX = SM == 1 ? 4 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char lds_char;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int KT = a.K >> 6, NC = a.N / BN;
    const int ABYTES = KT * 8192;                       // one [64][K] tile: KT panels of [64][64]
    const int steps_per_block = NC * KT;
    // prefetch distance of the A tiles in row blocks: the tile a block needs must be OLDER in the memory queue than the weight
    // tiles in flight (see the wait below), i.e. issued at least 3 steps before it is read
    const int PA = steps_per_block >= 3 ? 1 : (steps_per_block == 2 ? 2 : 3);
    const int NAB = PA + 1;
    char* Abuf = smem;
    char* Wring = smem + NAB * ABYTES;

    const int G = gridDim.x;
    const int first = blockIdx.x;
    const int my_blocks = first < a.nblocks ? (a.nblocks - first + G - 1) / G : 0;
    if (my_blocks == 0) return;                         // (whole workgroup)

    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(a.A, (unsigned)min(a.a_bytes, 0x7fffff00ull));
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(a.W, (unsigned)min(a.w_bytes, 0x7fffff00ull));
    const __amdgpu_buffer_rsrc_t rsD = make_rsrc(a.D, (unsigned)min((unsigned long long)((long long)(a.M - 1) * a.ldd + a.N) * 2ull, 0x7fffff00ull));

    // ---- A tile by LDS-DMA: piece pi (1 KiB) of a tile = panel pi / 8, rows 8 (pi % 8) + lane / 8, physical chunk lane % 8 ----
    const int NAP = 2 * KT;                             // pieces per wave
    auto dma_a = [&](int it) {                          // row block `it` of this workgroup -> buffer it % NAB
        const int rb = first + it * G;
        lds_char* dst = (lds_char*)Abuf + (it % NAB) * ABYTES;
        for (int i = 0; i < NAP; ++i) {
            const int pi = wave * NAP + i;
            const int r = 8 * (pi & 7) + (lane >> 3);
            const int m = rb * BM + r;
            const int k = (pi >> 3) * 64 + (((lane & 7) ^ kc_swz<8>(r)) << 3);
            const unsigned off = m < a.M ? (unsigned)(((long long)m * a.lda + k) * 2) : kOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(dst + pi * 1024), 16, off, 0, 0, 0);
        }
    };
    // ---- weight tile j of the stream -> ring slot j % NSW: tile (nc, kt) = rows nc * BN .. of W, k = kt * 64 .. ----
    unsigned woff[NWP];
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
        const int pi = wave * NWP + i;
        const int r = 8 * pi + (lane >> 3);
        woff[i] = (unsigned)(((long long)r * a.K + (((lane & 7) ^ kc_swz<8>(r)) << 3)) * 2);
    }
    const long long total_steps = (long long)my_blocks * steps_per_block;
    auto dma_w = [&](long long j) {
        const int s = (int)(j % steps_per_block);
        const int nc = s / KT, kt = s - nc * KT;
        const int koff = (nc * BN * a.K + kt * 64) * 2;            // scalar offset of the tile's first element
        lds_char* dst = (lds_char*)Wring + (int)(j % NSW) * WTILE + wave * (NWP * 1024);
#pragma unroll
        for (int i = 0; i < NWP; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, (unsigned)woff[i], koff, 0, 0);   // (the cast keeps the call non-type-dependent: woff's array type depends on BN, and a dependent call to a target builtin is silently dropped from the HOST-side instantiation -- no kernel stub, undefined symbol at load)
    };

    // ---- statistics accumulators: this lane's columns (FN fragments x 4) of every chunk, over rows = its l15 slice ----------
    float s1[NCMAX][FN][4], s2[NCMAX][FN][4];
#pragma unroll
    for (int c = 0; c < NCMAX; ++c)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { s1[c][j][e] = 0.f; s2[c][j][e] = 0.f; }
    long long rows_done = 0;
    auto flush_stats = [&](auto ncc, int nc_rt, int row, float count) {        // lanes with l15 == 0 write columns' (count, mean, M2)
        constexpr int c = decltype(ncc)::value;
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float u = s1[c][j][e], w = s2[c][j][e];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    u += __shfl_xor(u, o, 64);
                    w += __shfl_xor(w, o, 64);
                }
                if (l15 == 0) {
                    const int n = nc_rt * BN + wave * (BN / 4) + j * 16 + 4 * g + e;
                    float* o3 = a.stats + ((long long)row * a.N + n) * 3;
                    const float mean = u / count;
                    o3[0] = count;
                    o3[1] = mean;
                    o3[2] = fmaxf(w - u * mean, 0.f);
                }
                s1[c][j][e] = 0.f;
                s2[c][j][e] = 0.f;
            }
    };

    // ---- prologue: the first PA row blocks' A tiles, the first two weight tiles ------------------------------------------------
    for (int p = 0; p < PA && p < my_blocks; ++p) dma_a(p);
    dma_w(0);
    if (total_steps > 1) dma_w(1);

    f32x4 acc[FM][FN];
    long long j = 0;
    int since_a = 1000;                                 // steps since the last A DMA issue (big: none pending among the young ones)
    constexpr int NST = FM;                             // store instructions of one chunk epilogue (per wave)
    int ep1 = 0, ep2 = 0;                               // did step j - 1 / j - 2 end with a chunk epilogue
    for (int it = 0; it < my_blocks; ++it) {
        const int rb = first + it * G;
        const char* At = Abuf + (it % NAB) * ABYTES;
        const int valid = min(BM, a.M - rb * BM);
        for (int nc = 0; nc < NC; ++nc) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int jn = 0; jn < FN; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int kt = 0; kt < KT; ++kt, ++j) {
                // tile j (and, at a block's first step, the block's A tile, which is older in the queue) must have landed.  Younger
                // than W(j): W(j + 1) if it exists, and an A tile issued at step j - 1 or j - 2.
                {
                    // vmcnt retires in issue order and counts the epilogue's STORES too: behind W(j) (issued at step j - 2) sit the
                    // stores of the chunk epilogues of steps j - 2 and j - 1, W(j + 1), and an A tile issued at one of those steps.
                    // Waiting with a smaller count would stall every step on the previous chunk's store acknowledgements
                    // (measured: 3.7 us per step on the 64 -> 256 convolution of layer 1).
                    int allowed = 0;
                    if (j + 1 < total_steps) allowed += NWP;
                    if (since_a <= 1) allowed += NAP;              // an A tile was issued at step j - 1 or j - 2
                    if constexpr (SM != 2) allowed += NST * (ep1 + ep2);   // (SM 2 also stores statistics rows: not counted, waits longer)
                    pw_wait_vm_rt(allowed);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();           // everyone's pieces of tile j are in; everyone has finished step j - 1
                if (j + 2 < total_steps) dma_w(j + 2);  // its slot held tile j - 1
                ++since_a;
                if (nc == 0 && kt == 0 && it + PA < my_blocks) {
                    dma_a(it + PA);                     // buffer (it + PA) % NAB = (it - 1) % NAB: block it - 1 is finished (barrier above)
                    since_a = 0;
                }
                // fragments + MFMA
                const char* Wt = Wring + (int)(j % NSW) * WTILE;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 af[FM], bf[FN];
#pragma unroll
                    for (int i = 0; i < FM; ++i) af[i] = *(const bf16x8*)(At + kt * 8192 + kc_off_bf16<64>(i * 16 + l15, ks * 4 + g));
#pragma unroll
                    for (int jn = 0; jn < FN; ++jn) bf[jn] = *(const bf16x8*)(Wt + kc_off_bf16<64>(wave * (BN / 4) + jn * 16 + l15, ks * 4 + g));
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int jn = 0; jn < FN; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[jn], af[i], acc[i][jn], 0, 0, 0);
                }
                ep2 = ep1;
                ep1 = kt == KT - 1 ? 1 : 0;               // (the epilogue below follows this step)
            }
            // ---- chunk epilogue: statistics of the f32 results, bf16 store (rows past M hold exact zeros) ------------------------
            if constexpr (SM != 0) {
                auto add = [&](auto ncc) {
                    constexpr int c = decltype(ncc)::value;
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int jn = 0; jn < FN; ++jn)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float v = acc[i][jn][e];
                                s1[c][jn][e] += v;
                                s2[c][jn][e] = fmaf(v, v, s2[c][jn][e]);
                            }
                };
                if constexpr (SM == 1) {
                    switch (nc) {
                        case 0: add(std::integral_constant<int, 0>{}); break;
                        case 1: add(std::integral_constant<int, 1>{}); break;
                        case 2: add(std::integral_constant<int, 2>{}); break;
                        default: add(std::integral_constant<int, 3>{}); break;
                    }
                } else {
                    add(std::integral_constant<int, 0>{});
                    if (a.stats) flush_stats(std::integral_constant<int, 0>{}, nc, rb, (float)valid);
                }
            }
            // buffer stores with an out-of-range offset for rows past M: the hardware drops them, but every wave issues exactly
            // NST store instructions per epilogue -- the counted waits above rely on that
            const int n0 = nc * BN + wave * (BN / 4);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = rb * BM + i * 16 + l15;
                if constexpr (FN == 2) {
                    // lanes (g, g ^ 1) exchange a fragment so that every lane stores 8 consecutive columns = 16 bytes
                    const bool odd = g & 1;
                    const bf16x4 b0 = {(bf16_t)acc[i][0][0], (bf16_t)acc[i][0][1], (bf16_t)acc[i][0][2], (bf16_t)acc[i][0][3]};
                    const bf16x4 b1 = {(bf16_t)acc[i][1][0], (bf16_t)acc[i][1][1], (bf16_t)acc[i][1][2], (bf16_t)acc[i][1][3]};
                    const u32x2 a0 = __builtin_bit_cast(u32x2, b0), a1 = __builtin_bit_cast(u32x2, b1);
                    const u32x2 send = odd ? a0 : a1;
                    u32x2 recv;
                    recv[0] = __shfl_xor(send[0], 16, 64);
                    recv[1] = __shfl_xor(send[1], 16, 64);
                    const u32x4 o = odd ? u32x4{recv[0], recv[1], a1[0], a1[1]} : u32x4{a0[0], a0[1], recv[0], recv[1]};
                    const int n = n0 + 4 * g + (odd ? 12 : 0);
                    const unsigned off = m < a.M ? (unsigned)(((long long)m * a.ldd + n) * 2) : kOOB;
                    __builtin_amdgcn_raw_buffer_store_b128(o, rsD, off, 0, 0);
                } else {
                    const bf16x4 b0 = {(bf16_t)acc[i][0][0], (bf16_t)acc[i][0][1], (bf16_t)acc[i][0][2], (bf16_t)acc[i][0][3]};
                    const unsigned off = m < a.M ? (unsigned)(((long long)m * a.ldd + n0 + 4 * g) * 2) : kOOB;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, b0), rsD, off, 0, 0);
                }
            }
        }
        rows_done += valid;
    }
    if constexpr (SM == 1) {
        if (a.stats) {
            auto fl = [&](auto ncc) {
                constexpr int c = decltype(ncc)::value;
                if (c < NC) flush_stats(ncc, c, blockIdx.x, (float)rows_done);
            };
            fl(std::integral_constant<int, 0>{});
            fl(std::integral_constant<int, 1>{});
            fl(std::integral_constant<int, 2>{});
            fl(std::integral_constant<int, 3>{});
        }
    }
    pw_wait_vm<0>();
}

// ----------------------------------------------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------------------------------------------
static int pw_cus() {
    static const int cus = [] { hipDeviceProp_t pr; int d = 0; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess) ? pr.multiProcessorCount : 256; }();
    return cus;
}
PwPlan pw_stream_plan(long long M, int N, int K, bool want_stats) {
    PwPlan p;
    memset(&p, 0, sizeof(p));
    if (M < 64 || M > 0x3fffffff || K % 64 != 0 || K < 64 || K > 256 || N % 64 != 0 || N < 64 || N > 4096) return p;
    if ((long long)M * K * 2 >= 0x7fffff00ll || (long long)M * N * 2 >= 0x7fffff00ll) return p;
    p.bn = N % 128 == 0 ? 128 : 64;
    if (p.bn == 64 && N != 64) return p;               // 64-wide chunks only for N = 64 (statistics registers)
    const int KT = K / 64, NC = N / p.bn;
    const int spb = NC * KT;
    const int PA = spb >= 3 ? 1 : (spb == 2 ? 2 : 3);
    p.lds = (PA + 1) * KT * 8192 + 3 * p.bn * 128;
    if (p.lds > 160 * 1024) return p;
    p.sm = !want_stats ? 0 : (NC <= 4 ? 1 : 2);
    const int nblocks = (int)((M + 63) / 64);
    const int per_cu = p.lds <= 80 * 1024 ? 2 : 1;
    p.grid = std::min(nblocks, pw_cus() * per_cu);
    p.stat_rows = p.sm == 1 ? p.grid : (p.sm == 2 ? nblocks : 0);
    p.ok = 1;
    return p;
}
template <typename Kn>
static int pw_launch(Kn kernel, const PwPlan& pl, const PwArgs& a, hipStream_t s) {
    static std::mutex mu;
    static std::unordered_set<const void*> seen;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (seen.insert((const void*)kernel).second)
            HS_CHECK_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    hipLaunchKernelGGL(kernel, dim3(pl.grid), dim3(256), pl.lds, s, a);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
int pw_stream_run(const PwPlan& pl, const PwArgs& a, hipStream_t s) {
#define PW_CASE(BNv, SMv) \
    if (pl.bn == BNv && pl.sm == SMv) return pw_launch(pw_stream_kernel<BNv, SMv>, pl, a, s)
    PW_CASE(128, 0); PW_CASE(128, 1); PW_CASE(128, 2);
    PW_CASE(64, 0); PW_CASE(64, 1);
#undef PW_CASE
    set_error("pw_stream_run: no kernel for bn %d sm %d", pl.bn, pl.sm);
    return HS_ERR_ARG;
}

}  // namespace hs

using namespace hs;
extern "C" {
int32_t hs_pointwise_stat_rows(int64_t M, int32_t N, int32_t K) {
    const PwPlan pl = pw_stream_plan(M, N, K, true);
    return pl.ok ? pl.stat_rows : 0;
}
hs_status hs_pointwise_fwd(const void* x, int64_t M, int32_t K, int32_t ldx, const void* w, int32_t N, void* y, int32_t ldy,
                           float* stats, void* stream) {
    HS_REQUIRE(x && w && y && M > 0, "pointwise_fwd: null argument");
    const PwPlan pl = pw_stream_plan(M, N, K, stats != nullptr);
    HS_REQUIRE(pl.ok, "pointwise_fwd: shape %lld x %d x %d is not covered (K %% 64, 64 <= K <= 256, N %% 128 or N = 64)", (long long)M, N, K);
    HS_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= K && ldy >= N && ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) == 0),
               "pointwise_fwd: rows must be 16-byte aligned");
    PwArgs a;
    memset(&a, 0, sizeof(a));
    a.A = (const char*)x; a.W = (const char*)w; a.D = (char*)y;
    a.a_bytes = (unsigned long long)((M - 1) * ldx + K) * 2;
    a.w_bytes = (unsigned long long)N * K * 2;
    a.lda = ldx; a.ldd = ldy;
    a.M = (int)M; a.N = N; a.K = K;
    a.nblocks = (int)((M + 63) / 64);
    a.stats = stats;
    return pw_stream_run(pl, a, (hipStream_t)stream);
}
}
