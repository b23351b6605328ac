// Phase-pipelined bf16 GEMM body for the large dense contractions of the path (BERT projections / FFN: reference
// encoder.py:130-134 -> transformers BertLayer; D = epilogue(A . B^T), both operands K-contiguous, "nt").
//
// Why a second body.  The generic body (gemm_core.h) keeps one barrier per K tile and lets the compiler interleave; its K loop
// is bound by operand bytes in flight per CU ("A model that fits every tile measurement").  This body is built
// around the three things the CDNA4 guide's 256x256 template gets its rate from, re-derived for our operand layouts:
//   * a 256 x 256 x 64 tile on 8 waves (2 along M x 4 along N, 128 x 64 per wave): 128 FLOP per operand byte staged,
//   * the two waves of a SIMD run ONE BARRIER APART (waves 4-7 execute one extra s_barrier up front): between two barriers
//     one of them issues its LDS fragment reads + LDS-DMA, the other runs 16 MFMAs -- matrix beside memory on every SIMD,
//   * operand stages stay in flight ACROSS barriers behind counted s_waitcnt vmcnt(N).
// New here: the LDS ring is cut into QUADRANT-ORDERED half tiles.  A K tile is four 16 KiB slots -- A rows of the waves'
// first / second row quadrant (Aq0 / Aq1), B columns of the first / second column quadrant (Bq0 / Bq1) -- and a wave walks
// the four 64 x 32 quadrants of its output in the order (0,0) (0,1) (1,1) (1,0) with the B fragments of both column quadrants
// resident, so the slots die one after the other (Aq0 + Bq0 after phase 0, Bq1 after phase 1, Aq1 after phase 2) and each is
// refilled ONE phase after its last read with the data of tile t + 2.  Every half tile then has 6-7 phases (> 1.5 K tiles)
// to land and 5 of the 8 slots (80 KiB per CU) are in flight at any time -- the template's fixed half tiles leave 3.
//
// Hazards (P = global phase; G0 = waves 0..3, G1 = waves 4..7; interval I_k = between barriers k - 1 and k):
//   G0:  R(0) | M(0) | R(1) | M(1) | ...        G1:  -- | R(0) | M(0) | R(1) | ...      (R(P) in I_2P / I_2P+1)
//   R(P) = fragment reads of phase P, LDS-DMA of the slot freed by phase P - 1, then s_waitcnt vmcnt(N) for the slots phase
//   P + 1 reads and s_waitcnt lgkmcnt(0), THEN the barrier.  RAW: every wave's DMA pieces of a slot are retired by its own
//   counted vmcnt in R(P - 1) or earlier, and at least one barrier lies between that wait and any wave's R(P).  WAR: a slot
//   read in R(P) is overwritten by DMA issued in R(P + 1); G1's R(P) ends (lgkmcnt(0)) before barrier 2P + 1, G0's R(P + 1)
//   starts after it.  Both hold for any interleaving the barriers allow.
#pragma once
#include "gemm_core.h"

namespace hs {

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// BM x BN tile, WGM x WGN waves.  RS: also the row sums of A (bias gradient of a K-contiguous weight gradient, see gemm_core.h).
// STAG: waves NW/2.. run one barrier behind (two waves per SIMD); off for 4-wave workgroups (one wave per SIMD, the CU's
// second workgroup is the partner).
// DIAG (measurement build of the kernel, launched only when hs_gemm_debug_stamps armed a buffer): waves 0 and NW/2 stamp the
// shader clock around every segment of ONE K tile in the middle of the walk: 17 stamps per wave at a.stamps[(bx * 2 + group) * 20].
// TAG: unused by the code.  Two __global__ templates that instantiate the SAME specialization of this body fail on hipcc's host
// pass ("no matching function ... substitution failure", ROCm 7.2); a distinct TAG gives the second kernel its own specialization.
template <int BM, int BN, int WGM, int WGN, bool RS, bool STAG, bool DIAG = false, int TAG = 0>
__device__ __forceinline__ void gemm_bf16_p8_body(const GemmArgs& a, const int bx) {
    typedef bf16_t T;
    constexpr int BK = 64, NW = WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN;            // rows / columns of a wave's output
    constexpr int QM = WM / 2, QN = WN / 2;                // a phase's quadrant
    constexpr int FMQ = QM / 16, FNQ = QN / 16;            // 16 x 16 fragments per quadrant
    constexpr int FM = 2 * FMQ, FN = 2 * FNQ;
    constexpr int A_SLOT = (BM / 2) * BK * 2, B_SLOT = (BN / 2) * BK * 2;
    constexpr int BUF = 2 * A_SLOT + 2 * B_SLOT;           // one K tile
    constexpr int NA = A_SLOT / 1024 / NW, NB = B_SLOT / 1024 / NW;   // DMA instructions per wave per slot
    static_assert(QM % 16 == 0 && QN % 16 == 0 && NA >= 1 && NB >= 1 && A_SLOT % (1024 * NW) == 0 && B_SLOT % (1024 * NW) == 0, "tile / wave shape");
    static_assert(3 * NA + 3 * NB <= 63, "vmcnt field");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char lds_char;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int g = lane >> 4, l15 = lane & 15;
    int tm, tn;
    tile_from_block(a, tm, tn, bx);
    const int m0 = tm * BM, n0 = tn * BN;
    const int ntiles = a.K / BK;
    // (A rotated K walk -- each tile starting at a different K tile so that the workgroups sharing an operand panel through
    // one XCD's L2 do not miss on the same lines at the same moment -- was built and measured: 2.0 instead of 1.39 us per K
    // tile at 4096 x 4096.  Walking in lockstep is what lets the L2 merge the sharers' misses into one fetch.)

    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(a.A, (unsigned)min(a.a_bytes, 0x7fffff00ull));
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(a.B, (unsigned)min(a.b_bytes, 0x7fffff00ull));

    // ---- per-lane source offsets of the DMA pieces (bytes, without the K offset, which travels as the scalar offset) ----
    // piece i of this wave covers LDS bytes [(wave * NI + i) * 1024, + 1024) of a slot: slot row r = 8 * (wave * NI + i) + lane / 8,
    // physical 16-byte chunk lane % 8, which holds LOGICAL chunk (lane % 8) ^ swz(r) (the XOR swizzle sits on the source side
    // and on the fragment reads; the DMA writes linearly).  Slot row r of quadrant q <-> tile row (r / QM) * WM + q * QM + r % QM.
    unsigned offA[2][NA], offB[2][NB];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int r = 8 * (wave * NA + i) + (lane >> 3);
            const int row = m0 + (r / QM) * WM + q * QM + (r % QM);
            offA[q][i] = (unsigned)row * (unsigned)a.lda * 2u + (unsigned)(((lane & 7) ^ kc_swz<8>(r)) << 4);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int r = 8 * (wave * NB + i) + (lane >> 3);
            const int col = n0 + (r / QN) * WN + q * QN + (r % QN);
            offB[q][i] = (unsigned)col * (unsigned)a.ldb * 2u + (unsigned)(((lane & 7) ^ kc_swz<8>(r)) << 4);
        }
    }
    // slot bases inside a buffer: Aq0, Aq1, Bq0, Bq1
    auto dma_a = [&](int t, int q) {
        if (t >= ntiles) return;
        lds_char* dst = (lds_char*)smem + (t & 1) * BUF + q * A_SLOT + wave * (NA * 1024);
        const int koff = t * (BK * 2);
#pragma unroll
        for (int i = 0; i < NA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, offA[q][i], koff, 0, 0);
    };
    auto dma_b = [&](int t, int q) {
        if (t >= ntiles) return;
        lds_char* dst = (lds_char*)smem + (t & 1) * BUF + 2 * A_SLOT + q * B_SLOT + wave * (NB * 1024);
        const int koff = t * (BK * 2);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, offB[q][i], koff, 0, 0);
    };

    // ---- fragment reads ---------------------------------------------------------------------------------------------------
    bf16x8 af[FMQ][2], bq[2][FNQ][2];       // A fragments of the current row quadrant; B fragments of both column quadrants
    auto read_a = [&](int buf, int q) {     // (af is overwritten: q only selects the slot)
        const char* s = smem + buf * BUF + q * A_SLOT;
#pragma unroll
        for (int i = 0; i < FMQ; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[i][ks] = *(const bf16x8*)(s + kc_off_bf16<BK>(wm * QM + i * 16 + l15, ks * 4 + g));
    };
    auto read_b = [&](int buf, auto qc) {
        constexpr int q = decltype(qc)::value;
        const char* s = smem + buf * BUF + 2 * A_SLOT + q * B_SLOT;
#pragma unroll
        for (int j = 0; j < FNQ; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bq[q][j][ks] = *(const bf16x8*)(s + kc_off_bf16<BK>(wn * QN + j * 16 + l15, ks * 4 + g));
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accb[RS ? FM : 1];
    bool do_rowsum = false;
    if constexpr (RS) {
        do_rowsum = a.rowsum[0] != nullptr && tn == 0 && wn == 0;
#pragma unroll
        for (int i = 0; i < FM; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // the quadrant (qm, qn) of the wave's tile, K = 64: FMQ x FNQ x 2 MFMAs (operand roles swapped: a lane ends up with 4
    // consecutive n of one m, see gemm_core.h)
    auto mma = [&](auto qmc, auto qnc, bool with_rowsum) {
        constexpr int qm = decltype(qmc)::value, qn = decltype(qnc)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < FMQ; ++i)
#pragma unroll
                for (int j = 0; j < FNQ; ++j)
                    acc[qm * FMQ + i][qn * FNQ + j] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[qn][j][ks], af[i][ks], acc[qm * FMQ + i][qn * FNQ + j], 0, 0, 0);
        if constexpr (RS) {
            if (with_rowsum && do_rowsum) {
                const s16x4 o4 = {0x3f80, 0x3f80, 0x3f80, 0x3f80};      // bf16 1.0
                const bf16x8 ones = __builtin_bit_cast(bf16x8, __builtin_shufflevector(o4, o4, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < FMQ; ++i)
                        accb[qm * FMQ + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[i][ks], accb[qm * FMQ + i], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };
    // end of a read segment: the slots the NEXT phase reads have landed (this wave's pieces; the barrier makes it everyone's),
    // this wave's fragment reads are complete (the partner group may overwrite the slot after the barrier)
    auto fence_r = [&](auto nc, bool drain) {
        constexpr int N = decltype(nc)::value;
        if (drain) wait_vmcnt<0>();
        else if constexpr (N >= 0) wait_vmcnt<(N >= 0 ? N : 0)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto fence_m = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // DMA pieces that may still be in flight when the slots of the next phase must have landed (issue order per K tile:
    // Aq1 [phase 0, tile t + 1], Aq0, Bq0, Bq1 [phases 1-3, tile t + 2])
    using W_P0 = std::integral_constant<int, 2 * NA + 3 * NB>;       // before phase 0: Aq0, Bq0 of the tile landed
    using W_P1 = std::integral_constant<int, 3 * NA + 2 * NB>;       // before phase 1: Bq1
    using W_P2 = std::integral_constant<int, 3 * NA + 2 * NB>;       // before phase 2: Aq1
    using W_NONE = std::integral_constant<int, -1>;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // ---- prologue: tiles 0 and 1 in the steady-state issue order (the loop's first phase issues Aq1 of tile 1) ----------------
    dma_a(0, 0); dma_b(0, 0); dma_b(0, 1); dma_a(0, 1);
    dma_a(1, 0); dma_b(1, 0); dma_b(1, 1);
    if (ntiles >= 2) wait_vmcnt<W_P0::value>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (STAG) {
        if (wave >= NW / 2) {
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    unsigned long long stamp[DIAG ? 17 : 1];
    int nst = 0;
    auto STAMP = [&](int t) {
        if constexpr (DIAG) {
            if (t == ntiles / 2 && nst < 17) {
                __builtin_amdgcn_sched_barrier(0);
                stamp[nst++] = __builtin_readcyclecounter();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        // DMAs that the steady-state counts assume behind the awaited ones do not exist near the end of the K walk: drain instead
        const bool tail = t + 2 >= ntiles;
        // phase 0: quadrant (0, 0)
        STAMP(t);
        read_a(buf, 0);
        read_b(buf, I0{});
        dma_a(t + 1, 1);                       // slot Aq1 of the other buffer: read in phase 2 of tile t - 1
        if constexpr (DIAG) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); STAMP(t); }
        fence_r(W_P1{}, tail);
        STAMP(t);
        mma(I0{}, I0{}, true);
        STAMP(t);
        fence_m();
        // phase 1: quadrant (0, 1)
        STAMP(t);
        read_b(buf, I1{});
        dma_a(t + 2, 0);                       // slot Aq0 of this buffer: read in phase 0
        if constexpr (DIAG) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); STAMP(t); }
        fence_r(W_P2{}, tail);
        STAMP(t);
        mma(I0{}, I1{}, false);
        STAMP(t);
        fence_m();
        // phase 2: quadrant (1, 1)
        STAMP(t);
        read_a(buf, 1);
        dma_b(t + 2, 0);                       // slot Bq0 of this buffer: read in phase 0
        if constexpr (DIAG) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); STAMP(t); }
        fence_r(W_NONE{}, false);
        STAMP(t);
        mma(I1{}, I1{}, true);
        STAMP(t);
        fence_m();
        // phase 3: quadrant (1, 0), all operands in registers
        STAMP(t);
        dma_b(t + 2, 1);                       // slot Bq1 of this buffer: read in phase 1
        fence_r(W_P0{}, tail);
        STAMP(t);
        mma(I1{}, I0{}, false);
        STAMP(t);
        fence_m();
    }
    if constexpr (DIAG) {
        if (a.stamps && lane == 0 && (wave == 0 || wave == NW / 2)) {
            unsigned long long* o = a.stamps + ((long long)bx * 2 + (wave ? 1 : 0)) * 20;
#pragma unroll
            for (int i = 0; i < 17; ++i) o[i] = stamp[i];
            o[17] = (unsigned long long)nst;
        }
    }
    if constexpr (STAG) {
        if (wave < NW / 2) __builtin_amdgcn_s_barrier();      // the barrier the late group still owes
    }
    wait_vmcnt<0>();                                          // (nothing is in flight; the ring is free for the epilogue)

    // ---- epilogue (gemm_core.h): lane owns m = .. + l15, n = .. + 4g + {0..3} ----------------------------------------------
    const unsigned epi = epi_flags(a);
    if constexpr (RS) {
        if (do_rowsum && g == 0) {
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = m0 + wm * WM + i * 16 + l15;
                const int seg = a.seg_rows > 0 ? m / a.seg_rows : 0;
                a.rowsum[seg][m - seg * (a.seg_rows > 0 ? a.seg_rows : 0)] = accb[i][0];
            }
        }
    }
    // One straight-line body per feature set, instantiated per ROW FRAGMENT (static_for: with 32 fragments per lane a
    // `#pragma unroll` loop over them exceeds the unroller's budget, stays a loop, and the accumulators -- indexed by the loop
    // counter -- go to scratch).  The launcher (p8_epilogue_supported) sends only these feature sets here.
    const unsigned key = epi & ~EPI_VEC16;
#define HS_EPI_CASE(F)                                                                                                        \
    case (F):                                                                                                                 \
        static_for<0, FM>([&](auto ic) {                                                                                      \
            constexpr int i = decltype(ic)::value;                                                                            \
            run_epilogue<T, (F), true, 1, FN, 16, WN>(a, epi, *reinterpret_cast<f32x4(*)[1][FN]>(&acc[i]), m0 + wm * WM + i * 16, n0, 0, \
                                                      wn, l15, g, 0, 0);                                                     \
        });                                                                                                                   \
        break
    switch (key) {
        HS_EPI_CASE(EPI_VEC);
        HS_EPI_CASE(EPI_VEC | EPI_BIAS);
        HS_EPI_CASE(EPI_VEC | EPI_BIAS | EPI_GELU | EPI_PREACT);
        HS_EPI_CASE(EPI_VEC | EPI_BIAS | EPI_RES_POST);
        HS_EPI_CASE(EPI_VEC | EPI_BIAS | EPI_DROP | EPI_RES_POST);
        HS_EPI_CASE(EPI_VEC | EPI_MUL_GELU);
        HS_EPI_CASE(EPI_VEC | EPI_RES_POST);
        HS_EPI_CASE(EPI_VEC | EPI_OUT_F32);
        HS_EPI_CASE(EPI_VEC | EPI_OUT_F32 | EPI_SEG);
        default: break;      // not reachable: see p8_epilogue_supported
    }
#undef HS_EPI_CASE
}

// the epilogue feature sets the body above has code for (bf16 results need 16-byte storable rows)
inline bool p8_epilogue_supported(const GemmArgs& a) {
    unsigned f = 0;
    if (a.bias) f |= EPI_BIAS;
    if (a.colscale) f |= EPI_COLSCALE;
    if (a.mul_mode == HS_MUL_GELU_GRAD) f |= EPI_MUL_GELU;
    else if (a.mul_mode != HS_MUL_NONE) f |= EPI_MUL_RELU;
    if (a.seg_rows > 0) f |= EPI_SEG;
    if (a.D_preact) f |= EPI_PREACT;
    if (a.residual) f |= a.res_pre_act ? EPI_RES_PRE : EPI_RES_POST;
    if (a.act == HS_ACT_RELU) f |= EPI_RELU;
    else if (a.act == HS_ACT_GELU) f |= EPI_GELU;
    if (a.drop_thresh) f |= EPI_DROP;
    if (a.out_f32) f |= EPI_OUT_F32;
    if (a.accumulate) f |= EPI_ACCUM;
    if (a.vec_store) f |= EPI_VEC;
    if (a.parity) f |= EPI_PARITY;
    if (a.epi_generic || a.alpha != 1.f) return false;
    if (!(f & EPI_OUT_F32) && !a.vec16) return false;
    switch (f) {
        case EPI_VEC: case EPI_VEC | EPI_BIAS: case EPI_VEC | EPI_BIAS | EPI_GELU | EPI_PREACT: case EPI_VEC | EPI_BIAS | EPI_RES_POST:
        case EPI_VEC | EPI_BIAS | EPI_DROP | EPI_RES_POST: case EPI_VEC | EPI_MUL_GELU: case EPI_VEC | EPI_RES_POST:
        case EPI_VEC | EPI_OUT_F32: case EPI_VEC | EPI_OUT_F32 | EPI_SEG:
            return true;
    }
    return false;
}

// 256 x 256 tile, 8 waves, 128 KiB of LDS: one workgroup per CU
template <bool RS>
__global__ __launch_bounds__(512) void gemm_bf16_p8_256_kernel(const GemmArgs a) {
    gemm_bf16_p8_body<256, 256, 2, 4, RS, true>(a, blockIdx.x);
}
__global__ __launch_bounds__(512) void gemm_bf16_p8_256_diag_kernel(const GemmArgs a);   // measurement build (gemm_bf16_p8.hip)
// Grouped: the workgroups of up to 64 independent nt GEMMs (256 x 256 tiles) in ONE grid, problem table in device memory --
// the K-contiguous weight gradients of TWO BertLayers (8 GEMMs, K = the 4096 tokens: 216 tiles, one per CU, 64 K tiles each:
// the steady state this body is built for).  first_wg[i] = first workgroup of problem i, first_wg[n] = grid size.
template <bool RS>
__global__ __launch_bounds__(512) void gemm_bf16_p8_grouped_kernel(const GemmArgs* __restrict__ list, const int* __restrict__ first_wg, int n) {
    const int bid = blockIdx.x, lane = threadIdx.x & 63;
    const int lo = lane < n ? first_wg[lane] : 0x7fffffff;
    const int p = __builtin_amdgcn_readfirstlane(__popcll(__ballot(lo <= bid)) - 1);
    const int local = bid - __builtin_amdgcn_readfirstlane(first_wg[p]);
    gemm_bf16_p8_body<256, 256, 2, 4, RS, true, false, 1>(list[p], local);   // (TAG: its own specialization -- see the note at the body)
}
// 256 x 128 tile, 8 waves (4 x 2, 64 x 64 per wave), 96 KiB: the shapes whose N gives too few 256-wide tiles
template <bool RS>
__global__ __launch_bounds__(512) void gemm_bf16_p8_256x128_kernel(const GemmArgs a) {
    gemm_bf16_p8_body<256, 128, 4, 2, RS, true>(a, blockIdx.x);
}
// 128 x 128 tile, 4 waves (2 x 2, 64 x 64 per wave), 64 KiB: two workgroups per CU are each other's partners
template <bool RS>
__global__ __launch_bounds__(256) void gemm_bf16_p8_128_kernel(const GemmArgs a) {
    gemm_bf16_p8_body<128, 128, 2, 2, RS, false>(a, blockIdx.x);
}

}  // namespace hs
