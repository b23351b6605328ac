// MFMA GEMM / implicit-GEMM convolution core for gfx950.
//
//   D[m][n] = epilogue( alpha * sum_k A(m,k) * B(n,k) )
//
// One 256-thread workgroup (4 waves, 2x2) owns a BM x BN output tile.  Operand tiles are staged
// global -> registers -> LDS (double buffered; loads for tile t+1 are issued before the MFMAs of
// tile t and written to LDS after them), so im2col gathers, conv padding and tile tails are all
// predicated in the loader (out-of-range buffer offsets read as zero).
//
// Two MFMA paths share the loaders and the epilogue:
//   bf16 : v_mfma_f32_16x16x32_bf16, K-contiguous operands sit in LDS as [row][k] (XOR swizzled,
//          ds_read_b128), row-contiguous ("transposed") operands sit as [k][row] and are read with
//          ds_read_b64_tr_b16, so wgrad/dgrad need no transposed copies in HBM.
//   f32  : v_mfma_f32_32x32x2_f32 (bit-exact fmaf chain), both operands sit in LDS as [k][row].
//
// MFMA operand roles are swapped (MFMA-A := B tile, MFMA-B := A tile) so that each lane ends up with
// 4 consecutive n for one m: the epilogue then issues 8/16-byte row-contiguous accesses.
#pragma once
#include <type_traits>
#include "hs_common.h"

namespace hs {

struct GemmArgs {
    const char* A;
    const char* B;
    unsigned long long a_bytes, b_bytes;
    int lda, ldb;
    int M, N, K;
    hs_conv_geom g;
    FastDiv div_mhw, div_mw;   // rows m -> (n, y, x): divide by (Y*X), X
    FastDiv div_sc, div_c;     // cols   -> (r, s, c): divide by (S*C), C
    int batch_inner;
    long long a_bs0, a_bs1, b_bs0, b_bs1, d_bs0, d_bs1;   // in elements
    int tiles_m, tiles_n;
    int group_m;               // tile rows per L2 group (tile_from_block)
    // stride-2 dgrad: rows ordered by input-pixel parity class so a tile's structurally-zero filter taps can be skipped
    int parity;                // 1: row m = class * quarter + (n, i, j), pixel (2i + class/2, 2j + class%2)
    int quarter;               // N * (H/2) * (W/2)
    FastDiv div_qhw, div_qw;   // divide by (H/2)*(W/2), W/2
    const float* colscale;     // optional per-column scale before the bias (folded eval-mode BatchNorm)
    int res_pre_act;           // residual is added before the activation
    float* colstats;           // optional per (row tile, column) (count, mean, M2) of the result (fused BatchNorm statistics)
    int persist;               // >0: persistent launch, grid size = tile stride (split_k == 1, batch == 1)
    int lds_stages;            // ring slots actually allocated: min(3, K tiles per workgroup) (bf16 kernel)
    int split_k;               // >1: blockIdx.z is the split index
    int k_per_split;           // multiple of BK
    float* splitk_ws;
    unsigned* tickets;         // split-K (bf16 kernel): one arrival counter per output tile, zero before the launch; the
                               // workgroup that draws split_k - 1 sums the slabs and runs the epilogue, then re-zeroes it
    // epilogue
    char* D;
    int ldd;
    int out_f32;
    float alpha;
    const float* bias;
    int act;
    char* D_preact;
    const char* residual;
    int ldr;
    unsigned drop_thresh;
    float drop_inv_keep;
    unsigned long long drop_seed;
    int mul_mode;
    const char* mul_src;
    int ldm;
    int accumulate;
    int vec_store;             // 1: N%4==0 and all row strides/bases allow 4-wide accesses
    int seg_rows;              // >0: output rows are split over D / D_seg[0] / D_seg[1]
    char* D_seg[2];
    // optional (BNS kernels): BatchNorm-backward sums of the result, see hs_gemm_params.bnb_*
    const char* bnb_x;
    const float *bnb_scale, *bnb_shift, *bnb_mean, *bnb_invstd;
    float* bnb_partials;
    const char* bnb_y;         // optional: the mask comes from this saved block output (y > 0) and a.residual is added before the sums
    // optional (with bnb_partials, split_k == 1): the launch also FINISHES that BatchNorm's backward sums -- the last workgroup of
    // a column tile adds the tile rows' partials and writes dbeta, dgamma and the four coefficient vectors of the apply pass
    // behind the partial rows (hs_gemm_params.bnb_finish; bnb_handoff below).  bnb_tickets: zeroed arrival counters as for bnf.
    const float* bnb_gamma;
    float *bnb_dgamma, *bnb_dbeta;
    float bnb_invm;
    int bnb_train;
    unsigned* bnb_tickets;
    // optional (BNF kernels, with colstats): the launch also FINISHES the following BatchNorm's statistics -- the last workgroup
    // of a column tile merges the row tiles' partials and writes mean / invstd / scale / shift (+ running statistics); see
    // hs_gemm_params.bn_finish.  bnf_tickets: tiles_n * (1 + stat_groups(tiles_m)) zeroed arrival counters.
    const float *bnf_gamma, *bnf_beta;
    float *bnf_rmean, *bnf_rvar, *bnf_mean, *bnf_invstd, *bnf_scale, *bnf_shift;
    float bnf_eps, bnf_momentum;
    unsigned* bnf_tickets;
    float* rowsum[3];          // optional (row-contiguous A): rowsum[seg][m] = sum_k A[k][m], the bias gradient of a wgrad GEMM
    int c3_rows, c3_tpi, c3_fm;   // 3x3 halo kernel (conv3.hip): image rows per tile, tiles per image, 16-row fragments per tile
    int vec16;                    // bf16 result rows allow 16-byte (8-column) stores: N, ldd, batch strides % 8 == 0, bases 16-byte aligned
    int epi_generic;              // measurement only (hs_gemm_debug bit 32): always take the generic epilogue body
    int dbg;                      // measurement only: the hs_gemm_debug ablation bits
    unsigned long long* stamps;   // measurement only (hs_gemm_debug_stamps): 6 shader-clock stamps per workgroup, else NULL
};
// stamp k of this workgroup: 0 start (clock taken at entry, stored together with stamp 1), 1 first DMA issued, 2 first tile landed (barrier passed), 3 K loop done,
// 4 epilogue done (stores issued); slot 5 = XCC/CU id bits of HW_ID
#define HS_STAMP(k)                                                                                              \
    do {                                                                                                         \
        if (a.stamps && threadIdx.x == 0)                                                                        \
            a.stamps[((long long)blockIdx.z * gridDim.x + blockIdx.x) * 6 + (k)] = __builtin_readcyclecounter(); \
    } while (0)

// parity-major row order of a stride-2 dgrad: permuted row m -> (image, input pixel)
__device__ __forceinline__ int parity_class(const GemmArgs& a, int m) {
    return (m >= a.quarter) + (m >= 2 * a.quarter) + (m >= 3 * a.quarter);
}
__device__ __forceinline__ void parity_pixel(const GemmArgs& a, int m, unsigned& n, unsigned& h, unsigned& w) {
    const int cls = parity_class(a, m);
    const unsigned idx = m - cls * a.quarter;
    n = fdiv(idx, a.div_qhw);
    const unsigned ij = idx - n * a.div_qhw.d;
    const unsigned i = fdiv(ij, a.div_qw);
    h = 2 * i + (cls >> 1);
    w = 2 * (ij - i * a.div_qw.d) + (cls & 1);
}
__device__ __forceinline__ long long parity_row(const GemmArgs& a, int m) {   // natural (n, h, w) row index
    unsigned n, h, w;
    parity_pixel(a, m, n, h, w);
    return ((long long)n * a.g.H + h) * a.g.W + w;
}


// ------------------------------------------------------------------------------------------------
// epilogue for 4 consecutive n of one row m.  TIn = element type of mul_src; TOut chosen at run time.
// ------------------------------------------------------------------------------------------------
// Epilogue operands are device-global memory: the address space is stated explicitly so that no access can degrade to a
// flat_ instruction.
#define HS_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ void load4(const char* base, long long idx, bool vec, int nvalid, float* f) {
    const HS_GLOBAL T* p = (const HS_GLOBAL T*)((const T*)base + idx);
    if (vec) {
        if constexpr (sizeof(T) == 4) {
            f32x4 v = *(const HS_GLOBAL f32x4*)p;
            f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
        } else {
            u32x2 v = *(const HS_GLOBAL u32x2*)p;
            f[0] = __uint_as_float(v[0] << 16);
            f[1] = __uint_as_float(v[0] & 0xffff0000u);
            f[2] = __uint_as_float(v[1] << 16);
            f[3] = __uint_as_float(v[1] & 0xffff0000u);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = j < nvalid ? to_f32(p[j]) : 0.f;
    }
}
template <typename T>
__device__ __forceinline__ void store4(char* base, long long idx, bool vec, int nvalid, const float* f) {
    HS_GLOBAL T* p = (HS_GLOBAL T*)((T*)base + idx);
    if (vec) {
        if constexpr (sizeof(T) == 4) {
            f32x4 v = {f[0], f[1], f[2], f[3]};
            *(HS_GLOBAL f32x4*)p = v;
        } else {
            bf16x4 v = {(bf16_t)f[0], (bf16_t)f[1], (bf16_t)f[2], (bf16_t)f[3]};
            *(HS_GLOBAL bf16x4*)p = v;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < nvalid) p[j] = from_f32<T>(f[j]);
    }
}

// Epilogue features of a launch as one bit mask.  The generic epilogue body tests ~12 of them per output fragment (each an
// `s_load` + wait on a kernel argument and a branch; ~650 instructions of code per fragment): tools/gemm_stamps.py measured
// it at 6 us of the 12.4 us a 128x64 workgroup lives.  The hot feature sets therefore get compile-time-specialised bodies.
enum {
    EPI_BIAS = 1, EPI_COLSCALE = 2, EPI_MUL_GELU = 4, EPI_MUL_RELU = 8, EPI_SEG = 16, EPI_PREACT = 32, EPI_RES_PRE = 64,
    EPI_RES_POST = 128, EPI_RELU = 256, EPI_GELU = 512, EPI_DROP = 1024, EPI_OUT_F32 = 2048, EPI_ACCUM = 4096, EPI_VEC = 8192,
    EPI_PARITY = 16384, EPI_VEC16 = 32768
};
// The feature mask of a launch, computed once per workgroup.  (A struct holding copies of all epilogue arguments was tried
// first: it ended up on the stack -- 120 B of scratch per lane, measured as 2.75x the algorithmic HBM write bytes of a GEMM
// -- so the epilogue reads the kernel arguments directly; with a compile-time feature set that is a handful of scalar loads
// in straight-line code.)
__device__ __forceinline__ unsigned epi_flags(const GemmArgs& a) {
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
    if (a.vec16) f |= EPI_VEC16;
    return f;
}

// CF >= 0: the feature set is a compile-time constant (straight-line code for the combinations the training step uses, see
// the dispatch in the kernel); CF < 0: tested at run time.  FULL: the tile lies inside the matrix and rows are 4-wide
// storable, so no lane needs bounds or tail handling.  The fully generic body is ~650 instructions of branches per fragment
// (x8-16 fragments: > 40 KB of code walked sparsely), which is what made the epilogue as long as the K loop.
// DEFER: nothing is stored; v returns the final values and pv the pre-activation copy (EPI_PREACT), for the caller's
// 16-byte paired stores.
template <typename T, int CF = -1, bool FULL = false, bool DEFER = false>
__device__ __forceinline__ void epilogue4(const GemmArgs& a, unsigned rt_flags, long long dbase, int z, int m, int n,
                                          float* v, float* pv = nullptr) {
    if constexpr (!FULL) {
        if (m >= a.M || n >= a.N) return;
    }
    const unsigned fl = CF >= 0 ? (unsigned)CF : rt_flags;
    const long long mr = (fl & EPI_PARITY) ? parity_row(a, m) : (long long)m;   // row of D / residual / multiplier source
    const int nvalid = FULL ? 4 : min(4, a.N - n);
    const bool vec = FULL ? true : ((fl & EPI_VEC) && nvalid == 4);
    const bool out_f32 = fl & EPI_OUT_F32;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] *= a.alpha;
    if (fl & EPI_COLSCALE) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < nvalid) v[j] *= ((const HS_GLOBAL float*)a.colscale)[n + j];
    }
    if (fl & EPI_BIAS) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < nvalid) v[j] += ((const HS_GLOBAL float*)a.bias)[n + j];
    }
    if (fl & (EPI_MUL_GELU | EPI_MUL_RELU)) {
        float u[4];
        load4<T>(a.mul_src, mr * a.ldm + n, vec, nvalid, u);   // mul_src is never batched
        if (fl & EPI_MUL_GELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= gelu_grad_t<T>(u[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = u[j] > 0.f ? v[j] : 0.f;
        }
    }
    char* Dp = a.D;
    long long didx = dbase + mr * a.ldd + n;
    if (fl & EPI_SEG) {
        const int seg = m / a.seg_rows;
        if (seg > 0) Dp = seg == 1 ? a.D_seg[0] : a.D_seg[1];
        didx = (long long)(m - seg * a.seg_rows) * a.ldd + n;
    }
    if (fl & EPI_PREACT) {
        if constexpr (DEFER) {
#pragma unroll
            for (int j = 0; j < 4; ++j) pv[j] = v[j];
        } else {
            if (out_f32) store4<float>(a.D_preact, didx, vec, nvalid, v);
            else store4<T>(a.D_preact, didx, vec, nvalid, v);
        }
    }
    if (fl & EPI_RES_PRE) {
        float r[4];
        const long long ridx = dbase + mr * a.ldr + n;
        if (out_f32) load4<float>(a.residual, ridx, vec, nvalid, r);
        else load4<T>(a.residual, ridx, vec, nvalid, r);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += r[j];
    }
    if (fl & EPI_RELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
    } else if (fl & EPI_GELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = gelu_fwd_t<T>(v[j]);
    }
    if (fl & EPI_DROP) {
        const unsigned long long q = ((unsigned long long)z * a.M + m) * (unsigned long long)a.N + n;
        if ((q & 3) == 0) {
            float sc[4];
            dropout_scale4(a.drop_seed, q, a.drop_thresh, a.drop_inv_keep, sc);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= sc[j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= dropout_scale(a.drop_seed, q + j, a.drop_thresh, a.drop_inv_keep);
        }
    }
    if (fl & EPI_RES_POST) {
        float r[4];
        const long long ridx = dbase + mr * a.ldr + n;   // residual shares D's batch strides
        if (out_f32) load4<float>(a.residual, ridx, vec, nvalid, r);
        else load4<T>(a.residual, ridx, vec, nvalid, r);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += r[j];
    }
    if constexpr (DEFER) return;
    if (out_f32) {
        if (fl & EPI_ACCUM) {
            float o[4];
            load4<float>(Dp, didx, vec, nvalid, o);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += o[j];
        }
        store4<float>(Dp, didx, vec, nvalid, v);
    } else {
        store4<T>(Dp, didx, vec, nvalid, v);
    }
}

// ------------------------------------------------------------------------------------------------
// LDS layouts
// ------------------------------------------------------------------------------------------------
// bf16, K-contiguous tile [rows][BK]: 16-byte chunk `kc` of row `r`.
// XOR swizzle of the chunk index, a function of the row's 256-byte bank row j.  ds_read_b128 serves a wave in four groups of 16
// lanes that are NOT contiguous ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...: micro-architecture guide, LDS table), and a
// fragment read has lane = 16 g + row, chunk = g.  BK = 64 (8 chunks per row): swz = j puts the 16 lanes of every group on 16
// different 16-byte bank quads.  BK = 32 (4 chunks per row): swz = j was a 2-way conflict in every group (rows 0-3 with chunk 0
// and rows 4-7 with chunk 1 ^ 1 met on the same quads; SQ_LDS_BANK_CONFLICT = 48 % of SQ_LDS_IDX_ACTIVE on the 128x128x32
// kernel) -- swz = (-j) mod 4 = {0, 3, 2, 1} separates them: {s0, s3, 1^s1, 1^s2} = {0, 1, 2, 3} and likewise for the other groups.
template <int CPR>
__device__ __forceinline__ int kc_swz(int r) {
    constexpr int RPB = 16 / CPR;          // rows per 256-byte bank row
    const int j = (r / RPB) % CPR;
    return CPR == 4 ? ((CPR - j) & (CPR - 1)) : j;
}
template <int BK>
__device__ __forceinline__ int kc_off_bf16(int r, int kc) {
    constexpr int CPR = BK / 8;            // chunks per row
    return r * (BK * 2) + ((kc ^ kc_swz<CPR>(r)) << 4);
}
// bf16, row-contiguous tile [BK][BR]: byte offset of element (k, col); 32-byte granules are XOR-ed
// with a function of k so that the 8 k-rows one half-wave transposed read touches hit distinct banks.
template <int BR>
__device__ __forceinline__ int rc_off_bf16(int k, int col) {
    constexpr int G = BR / 16;             // 32-byte granules per k-row
    int f;
    if constexpr (G >= 8) f = (k & 3) | (((k >> 3) & 1) << 2);
    else f = ((k >> 1) & 1) | (((k >> 3) & 1) << 1);
    return k * (BR * 2) + ((((col >> 4) ^ f) & (G - 1)) << 5) + ((col & 15) << 1);
}

// ------------------------------------------------------------------------------------------------
// operand address generation.  All offsets are in BYTES into the operand's buffer resource.
// ------------------------------------------------------------------------------------------------
// K-contiguous operand (A only): per-thread row state.
template <int KIND>
struct KcRow {
    int base;      // element offset of the row (plain) / of the image window origin (conv)
    int hb, wb;    // conv: window origin (may be negative); dgrad: h+pad, w+pad
    bool valid;
};

template <int KIND>
__device__ __forceinline__ void kc_row_setup(const GemmArgs& a, int m, KcRow<KIND>& st) {
    st.valid = m < a.M;
    st.hb = st.wb = 0;
    if constexpr (KIND == HS_A_KC) {
        st.base = m * a.lda;
    } else if constexpr (KIND == HS_A_CONV) {
        const unsigned n = fdiv(m, a.div_mhw);
        const unsigned pq = m - n * a.div_mhw.d;
        const unsigned p = fdiv(pq, a.div_mw);
        const unsigned q = pq - p * a.div_mw.d;
        st.hb = (int)p * a.g.stride - a.g.pad;
        st.wb = (int)q * a.g.stride - a.g.pad;
        st.base = (int)n * a.g.img_pitch + st.hb * a.g.row_pitch + (int)q * a.g.qstep - a.g.pad * a.g.C;
    } else {   // HS_A_DGRAD: m = (n, h, w) over the conv INPUT grid (or its parity-major permutation)
        unsigned n, h, w;
        if (a.parity) {
            parity_pixel(a, st.valid ? m : 0, n, h, w);
        } else {
            n = fdiv(m, a.div_mhw);
            const unsigned hw = m - n * a.div_mhw.d;
            h = fdiv(hw, a.div_mw);
            w = hw - h * a.div_mw.d;
        }
        st.hb = (int)h + a.g.pad;
        st.wb = (int)w + a.g.pad;
        st.base = (int)n * a.g.P * a.g.Q * a.g.K;
    }
}

// uniform (per k-tile) decomposition of k0 -> (r, s, c0) with channel count CH
struct KTile {
    int r, s, c0;
};
__device__ __forceinline__ KTile ktile_rsc(int k0, int CH, int S) {
    KTile t;
    const int rs = k0 / CH;
    t.c0 = k0 - rs * CH;
    t.r = rs / S;
    t.s = rs - t.r * S;
    return t;
}

template <int KIND, int ESZ>
__device__ __forceinline__ unsigned kc_chunk_off(const GemmArgs& a, const KcRow<KIND>& st, const KTile& kt,
                                                 int k, int kend) {
    // k = absolute k index of the chunk's first element
    bool ok = st.valid && k < kend;
    int off;
    if constexpr (KIND == HS_A_KC) {
        off = st.base + k;
    } else if constexpr (KIND == HS_A_CONV) {
        const int h = st.hb + kt.r, w = st.wb + kt.s;
        if (!a.g.no_bounds) ok = ok && (unsigned)h < (unsigned)a.g.H && (unsigned)w < (unsigned)a.g.W;
        off = st.base + kt.r * a.g.row_pitch + kt.s * a.g.C + (k - (kt.r * a.g.S + kt.s) * a.g.C);
    } else {
        int hp = st.hb - kt.r, wp = st.wb - kt.s;
        ok = ok && hp >= 0 && wp >= 0;
        if (a.g.stride == 2) {
            ok = ok && !((hp | wp) & 1);
            hp >>= 1;
            wp >>= 1;
        }
        ok = ok && hp < a.g.P && wp < a.g.Q;
        off = st.base + (hp * a.g.Q + wp) * a.g.K + (k - (kt.r * a.g.S + kt.s) * a.g.K);
    }
    return ok ? (unsigned)off * ESZ : kOOB;
}

// Row-contiguous operands (A_RC, B_RC, B_WDGRAD, B_CONV): chunk = EPC consecutive rows/cols at one k.
struct RcCol {
    int base;     // element offset contribution of the column chunk
    int r, s;     // B_CONV: filter tap of the column chunk
    bool valid;
};
template <int KIND, bool IS_A>
__device__ __forceinline__ void rc_col_setup(const GemmArgs& a, int col, RcCol& st) {
    st.r = st.s = 0;
    if constexpr (IS_A) {
        st.valid = col < a.M;
        st.base = col;
    } else if constexpr (KIND == HS_B_RC) {
        st.valid = col < a.N;
        st.base = col;
    } else if constexpr (KIND == HS_B_WDGRAD) {
        st.valid = col < a.N;   // n = c (input channel)
        st.base = col;
    } else {   // HS_B_CONV: col = (r, s, c)
        st.valid = col < a.N;
        const unsigned r = fdiv(col, a.div_sc);
        const unsigned sc = col - r * a.div_sc.d;
        const unsigned s = fdiv(sc, a.div_c);
        const unsigned c = sc - s * a.div_c.d;
        st.r = r;
        st.s = s;
        st.base = (int)r * a.g.row_pitch + ((int)s - a.g.pad) * a.g.C + (int)c;
    }
}
template <int KIND, bool IS_A, int ESZ>
__device__ __forceinline__ unsigned rc_chunk_off(const GemmArgs& a, const RcCol& st, const KTile& kt, int k,
                                                 int kend) {
    bool ok = st.valid && k < kend;
    int off;
    if constexpr (IS_A) {
        off = k * a.lda + st.base;
    } else if constexpr (KIND == HS_B_RC) {
        off = k * a.ldb + st.base;
    } else if constexpr (KIND == HS_B_WDGRAD) {
        // k = (r, s, ko); filter element [ko][r][s][c]
        const int rs = kt.r * a.g.S + kt.s;
        const int ko = k - rs * a.g.K;
        off = ko * (a.g.R * a.g.S * a.g.C) + rs * a.g.C + st.base;
    } else {
        // k = m = (n, p, q)
        const unsigned n = fdiv(k, a.div_mhw);
        const unsigned pq = k - n * a.div_mhw.d;
        const unsigned p = fdiv(pq, a.div_mw);
        const unsigned q = pq - p * a.div_mw.d;
        const int h = (int)p * a.g.stride - a.g.pad + st.r;
        const int w = (int)q * a.g.stride - a.g.pad + st.s;
        if (!a.g.no_bounds) ok = ok && (unsigned)h < (unsigned)a.g.H && (unsigned)w < (unsigned)a.g.W;
        off = (int)n * a.g.img_pitch + ((int)p * a.g.stride - a.g.pad) * a.g.row_pitch + (int)q * a.g.qstep + st.base;
    }
    return ok ? (unsigned)off * ESZ : kOOB;
}

// scalar-element fallback load of one chunk (used when strides/sizes are not chunk aligned):
// element j of a K-contiguous chunk is valid while k+j < kend; of a row-contiguous chunk while col+j < limit.
template <typename T>
__device__ __forceinline__ u32x4 load_chunk(__amdgpu_buffer_rsrc_t rs, unsigned off, bool vec, int nvalid) {
    if (vec) return buf_load16(rs, off);
    u32x4 c = {0, 0, 0, 0};
    if (off == kOOB) return c;
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < nvalid) c[j] = buf_load4(rs, off + 4 * j);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < nvalid) {
                unsigned v = buf_load2(rs, off + 2 * j);
                c[j >> 1] |= (j & 1) ? (v << 16) : v;
            }
    }
    return c;
}

// zero the elements of a K-contiguous chunk that lie at or beyond kend (only taken when K % chunk != 0)
template <typename T>
__device__ __forceinline__ void mask_tail(u32x4& c, int nvalid) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j >= nvalid) c[j] = 0u;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j >= nvalid) c[j >> 1] &= (j & 1) ? 0x0000ffffu : 0xffff0000u;
    }
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// in-launch split-K: slices hand their slabs off in groups of this many (gemm_bf16_kernel); the workspace holds one slab per
// slice plus, for more than one group, one per group
constexpr int kSplitGroup = 8;
constexpr int kStatGroup = 32;     // row tiles whose BatchNorm partials one hand-off merges (BNF kernels)
__host__ __device__ inline int stat_groups(int tiles_m) { return tiles_m <= kStatGroup ? 0 : (tiles_m + kStatGroup - 1) / kStatGroup; }
inline long long splitk_slabs(int split) { return split <= kSplitGroup ? split : split + (split + kSplitGroup - 1) / kSplitGroup; }

constexpr bool a_is_rc(int k) { return k == HS_A_RC; }
constexpr bool b_is_rc(int k) { return k != HS_B_KC; }

__device__ __forceinline__ void tile_from_block(const GemmArgs& a, int& tm, int& tn, int bid) {
    // XCD-aware remap: workgroup ids go round-robin over the 8 XCDs, so XCD x is handed the contiguous run of tile
    // ids [x*nwg/8, (x+1)*nwg/8) (bijective for any grid size).  Tile ids walk the output in groups of `group_m` tile
    // rows, m fastest inside a group: the workgroups an XCD runs at one time then cover group_m x (S/group_m) tiles
    // and share group_m A panels and S/group_m B panels through that XCD's L2 (each XCD has its own L2; with plain
    // n-fastest order one XCD streamed the whole B operand once per tile row).
    const int nwg = a.tiles_m * a.tiles_n;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int gsz = a.group_m * a.tiles_n;
    const int grp = id / gsz;
    const int first_m = grp * a.group_m;
    const int gm = min(a.group_m, a.tiles_m - first_m);
    const int within = id - grp * gsz;
    tn = within / gm;
    tm = first_m + (within - tn * gm);
}

// ================================================================================================
// bf16 kernel
// ================================================================================================
// inverse of the granule XOR of rc_off_bf16 for a physical 16-byte chunk index: which logical chunk lives there
template <int BR>
__device__ __forceinline__ int rc_logical_chunk(int k, int pcc) {
    constexpr int G = BR / 16;
    int f;
    if constexpr (G >= 8) f = (k & 3) | (((k >> 3) & 1) << 2);
    else f = ((k >> 1) & 1) | (((k >> 3) & 1) << 1);
    return ((((pcc >> 1) ^ f) & (G - 1)) << 1) | (pcc & 1);
}

// Staging is LDS-DMA (buffer_load_dwordx4 ... lds): every wave instruction moves 64 x 16 B straight into a
// linear 1 KiB run of the LDS tile, so the XOR swizzles are applied on the per-lane SOURCE address (which
// logical chunk belongs at this LDS slot) and again on the fragment reads.  Out-of-range source offsets
// write zeros (verified on gfx950), which is how M/N/K tails and conv padding are predicated.
// Contract for K-contiguous operands: K % 8 == 0, or the row is zero padded up to the next multiple of 8.
// The epilogue of one workgroup tile for a compile-time feature set CF (or the run-time mask `epi` when CF < 0).
template <typename T, int CF, bool FULL, int FM, int FN, int WM, int WN>
__device__ __forceinline__ void run_epilogue(const GemmArgs& a, const unsigned epi, f32x4 (&acc)[FM][FN], int m0, int n0, int wm,
                                             int wn, int l15, int g, long long d_boff, int z) {
        // bf16 results of full tiles: 8-byte stores of the lane-owned 4 columns reach memory as separate partial-sector
        // writes once the epilogue is fast (measured 2.75x the algorithmic write bytes), so two column fragments are
        // exchanged between lane pairs (g, g^1) and every lane stores 8 consecutive columns = 16 bytes: each row gets whole
        // 64-byte sectors from one instruction.
        if constexpr (FULL && CF >= 0 && !(CF & (EPI_OUT_F32 | EPI_SEG | EPI_PARITY)) && (FN % 2 == 0) && sizeof(T) == 2) {
            const bool odd = g & 1;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = m0 + wm * WM + i * 16 + l15;
#pragma unroll
                for (int j = 0; j < FN; j += 2) {
                    const int n = n0 + wn * WN + j * 16 + 4 * g;
                    float v0[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    float v1[4] = {acc[i][j + 1][0], acc[i][j + 1][1], acc[i][j + 1][2], acc[i][j + 1][3]};
                    float p0[4], p1[4];
                    epilogue4<T, CF, true, true>(a, epi, d_boff, z, m, n, v0, p0);
                    epilogue4<T, CF, true, true>(a, epi, d_boff, z, m, n + 16, v1, p1);
                    const long long didx = d_boff + (long long)m * a.ldd + (odd ? n + 12 : n);
                    auto pair_store = [&](char* base, const float* x0, const float* x1) {
                        const bf16x4 b0 = {(bf16_t)x0[0], (bf16_t)x0[1], (bf16_t)x0[2], (bf16_t)x0[3]};
                        const bf16x4 b1 = {(bf16_t)x1[0], (bf16_t)x1[1], (bf16_t)x1[2], (bf16_t)x1[3]};
                        const u32x2 a0 = __builtin_bit_cast(u32x2, b0), a1 = __builtin_bit_cast(u32x2, b1);
                        const u32x2 send = odd ? a0 : a1;       // even lanes keep fragment j, odd lanes fragment j + 1
                        u32x2 recv;
                        recv[0] = __shfl_xor(send[0], 16, 64);
                        recv[1] = __shfl_xor(send[1], 16, 64);
                        const u32x4 o = odd ? u32x4{recv[0], recv[1], a1[0], a1[1]} : u32x4{a0[0], a0[1], recv[0], recv[1]};
                        *(HS_GLOBAL u32x4*)((HS_GLOBAL bf16_t*)base + didx) = o;
                    };
                    if constexpr (CF & EPI_PREACT) pair_store(a.D_preact, p0, p1);
                    pair_store(a.D, v0, v1);
                    if (i == 0 && j == 0) HS_STAMP(5);
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = m0 + wm * WM + i * 16 + l15;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int n = n0 + wn * WN + j * 16 + 4 * g;
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                epilogue4<T, CF, FULL>(a, epi, d_boff, z, m, n, v);
                if (i == 0 && j == 0) HS_STAMP(5);
            }
        }
}

// WGM: waves along M (x 2 along N): 2 -> 256 threads (the default), 4 -> 512 threads for the 256-row tile.
// RS: the row sums of A (sum over k) are produced too (rowsum[]: the bias gradient of a weight-gradient GEMM whose A operand
// is dY, row-contiguous or -- RS instantiations -- K-contiguous dY^T); row-contiguous A always has the code.
// NS: slots of the LDS operand ring (3 by default).  Long-K launches that cannot fill the chip with workgroups (split-K weight
// gradients of the convolutions: 1-2 workgroups per CU, HBM-bound streams of 64 MB and more) take a deeper ring: with NS - 1
// tiles in flight per workgroup instead of 2, a CU keeps enough bytes in flight to cover the memory latency.
template <int NDMA>
__device__ __forceinline__ void wait_vm_tiles(int tiles) {       // wait until at most `tiles` of this wave's tile DMAs are in flight
    switch (tiles) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NDMA) : "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NDMA) : "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * NDMA) : "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * NDMA) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(7 * NDMA) : "memory"); break;
    }
}
// ---- BatchNorm statistics hand-off of a forward convolution (BNF kernels; also the 3x3 halo kernel, conv3.hip) ------------
// Called by every thread of the workgroup after its (count, mean, M2) partial row of tile row tm has been stored
// write-through and its arrival ticket drawn (bnf_drawn, held by thread 0).  NT threads, BN columns per tile, smem: at
// least (NT * 3 + 1) floats that nobody else uses any more.
template <int NT, int BN>
__device__ __forceinline__ void bnf_handoff(const GemmArgs& a, char* smem, const int tid, const int tm, const int tn, const int en0,
                                            int bnf_drawn) {
    // The statistics of the following BatchNorm are finished inside the launch (the separate bn_stats_final_kernel sat on
    // the ResNet stream's critical path 53 times a step: ~7 us each + a dispatch gap, for microseconds of arithmetic).
    // Same hand-off as the split-K reduction above: write-through partials -> vmcnt(0) -> barrier -> one relaxed
    // agent-scope ticket; the workgroup that arrives last acquires and merges.  Two levels so that the tail stays
    // short: the row tiles of a column tile hand off in groups of kStatGroup (the last of a group merges it into one
    // (count, mean, M2) row behind the tile rows), then the last group merges the group rows and writes the
    // statistics.  Merging is the grouped formula (n = sum n_i, mean = sum n_i mean_i / n,
    // M2 = sum M2_i + n_i (mean_i - mean)^2, evaluated about a shift) in a fixed order: deterministic whichever tile
    // arrives last.
    constexpr int TPC = NT / BN, UB = 8;
    static_assert(NT % BN == 0 && TPC >= 1, "threads per column");
    float* sh = (float*)smem;                                  // [TPC][BN][3]; the ring is free
    int* flag = (int*)(sh + NT * 3);
    const int tiles_m = a.tiles_m, ngroups = stat_groups(tiles_m);
    const int col = tid % BN, sub = tid / BN, n = en0 + col;
    const bool cok = n < a.N;
    const int grp = tm / kStatGroup;
    int first = ngroups ? grp * kStatGroup : 0, count = ngroups ? min(kStatGroup, tiles_m - first) : tiles_m;
    unsigned* ticket = ngroups ? a.bnf_tickets + a.tiles_n + tn * ngroups + grp : a.bnf_tickets + tn;
    const unsigned st_bytes = (unsigned)min((unsigned long long)(tiles_m + ngroups) * a.N * 12ull, 0x7fffff00ull);
    const __amdgpu_buffer_rsrc_t rsS = make_rsrc(a.colstats, st_bytes);
    float cnt = 0.f, mean = 0.f, m2 = 0.f;
    bool early = true;                                         // the first hand-off's ticket was drawn before the epilogue
#pragma unroll 1
    for (int level = ngroups ? 0 : 1; level < 2; ++level) {
        if (!early) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the group row's stores have left
            __syncthreads();
            if (tid == 0) bnf_drawn = (int)__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        early = false;
        if (tid == 0) *flag = bnf_drawn;
        __syncthreads();
        const int drawn = *flag;
        __syncthreads();
        if (drawn != count - 1) return;                        // (workgroup-uniform)
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // for the next launch
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // one pass, every load of a batch in flight together (dependent passes cost a memory round trip each on the
        // launch's tail).  Each thread sums n_i, n_i d_i and M2_i + n_i d_i^2 of its rows with d_i = mean_i - k, k = the
        // mean of the thread's FIRST row (any row's mean is within a tile's standard error of the answer, so the final
        // S2 - S1^2 / S0 cancels nothing; the first row rather than this workgroup's own tile so that the rounding does
        // not depend on which tile arrived last); the TPC threads of a column then merge their triples the same way.
        float k = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int r0 = sub; r0 < count; r0 += TPC * UB) {
            float v[UB][3];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int r = r0 + u * TPC;
                const unsigned off = (cok && r < count) ? (unsigned)(((long long)(first + r) * a.N + n) * 12) : kOOB;
#pragma unroll
                for (int e = 0; e < 3; ++e)
                    v[u][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsS, off + 4 * e, 0, 16 /* sc1: written by other XCDs */));
            }
            if (r0 == sub) k = v[0][1];
#pragma unroll
            for (int u = 0; u < UB; ++u) {                      // rows past the range read as zeros: no contribution
                const float c = v[u][0], d = v[u][1] - k;
                s0 += c;
                s1 = fmaf(c, d, s1);
                s2 += fmaf(c * d, d, v[u][2]);
            }
        }
        {
            const float dm = s0 > 0.f ? s1 / s0 : 0.f;
            sh[(sub * BN + col) * 3] = s0;
            sh[(sub * BN + col) * 3 + 1] = k + dm;
            sh[(sub * BN + col) * 3 + 2] = fmaxf(s2 - s1 * dm, 0.f);
        }
        __syncthreads();
        k = sh[col * 3 + 1];                                   // thread 0 of the column always has a row
        s0 = 0.f; s1 = 0.f; s2 = 0.f;
#pragma unroll
        for (int u = 0; u < TPC; ++u) {
            const float c = sh[(u * BN + col) * 3], d = sh[(u * BN + col) * 3 + 1] - k;
            s0 += c;
            s1 = fmaf(c, d, s1);
            s2 += fmaf(c * d, d, sh[(u * BN + col) * 3 + 2]);
        }
        __syncthreads();
        cnt = s0;
        const float dm = s0 > 0.f ? s1 / s0 : 0.f;
        mean = k + dm;
        m2 = fmaxf(s2 - s1 * dm, 0.f);
        if (level == 0) {
            if (sub == 0 && cok) {
                float* o = a.colstats + ((long long)(tiles_m + grp) * a.N + n) * 3;
                __hip_atomic_store(o, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(o + 1, mean, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(o + 2, m2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            first = tiles_m;
            count = ngroups;
            ticket = a.bnf_tickets + tn;
        }
    }
    if (sub == 0 && cok) {                                     // the launch's last workgroup of this column tile
        const float rows = (float)a.M;
        const float var = m2 / rows;                           // biased, used for normalisation
        const float invstd = rsqrtf(var + a.bnf_eps);
        a.bnf_mean[n] = mean;
        a.bnf_invstd[n] = invstd;
        const float gm = a.bnf_gamma ? a.bnf_gamma[n] : 1.f, bt = a.bnf_beta ? a.bnf_beta[n] : 0.f;
        a.bnf_scale[n] = gm * invstd;
        a.bnf_shift[n] = bt - mean * gm * invstd;
        if (a.bnf_rmean) {
            const float unbiased = a.M > 1 ? m2 / (rows - 1.f) : var;
            a.bnf_rmean[n] = (1.f - a.bnf_momentum) * a.bnf_rmean[n] + a.bnf_momentum * mean;
            a.bnf_rvar[n] = (1.f - a.bnf_momentum) * a.bnf_rvar[n] + a.bnf_momentum * unbiased;
        }
    }
}
// ---- BatchNorm-backward sums hand-off of a data-gradient GEMM (BNS kernels) ----------------------------------------------------
// Same protocol as bnf_handoff with plain sums: partial row tm = (sum dz, sum dz * xhat) per column; the last tile row of a
// group of kStatGroup adds the group into a row behind the tile rows, the last group adds the group rows and writes what
// bn_bwd_final_kernel (norm.hip) writes: dbeta, dgamma and coef = [ca | cb | cc | mean] behind all rows.  Fixed summation
// order whichever workgroup arrives last.
template <int NT, int BN>
__device__ __forceinline__ void bnb_handoff(const GemmArgs& a, char* smem, const int tid, const int tm, const int tn, const int en0,
                                            int drawn_) {
    constexpr int TPC = NT / BN, UB = 8;
    static_assert(NT % BN == 0 && TPC >= 1, "threads per column");
    float* sh = (float*)smem;                                  // [TPC][BN][2]; the ring is free
    int* flag = (int*)(sh + NT * 2);
    const int tiles_m = a.tiles_m, ngroups = stat_groups(tiles_m);
    const int col = tid % BN, sub = tid / BN, n = en0 + col;
    const bool cok = n < a.N;
    const int grp = tm / kStatGroup;
    int first = ngroups ? grp * kStatGroup : 0, count = ngroups ? min(kStatGroup, tiles_m - first) : tiles_m;
    unsigned* ticket = ngroups ? a.bnb_tickets + a.tiles_n + tn * ngroups + grp : a.bnb_tickets + tn;
    const unsigned st_bytes = (unsigned)min((unsigned long long)(tiles_m + ngroups) * a.N * 8ull, 0x7fffff00ull);
    const __amdgpu_buffer_rsrc_t rsS = make_rsrc(a.bnb_partials, st_bytes);
    float t1 = 0.f, t2 = 0.f;
    bool early = true;
#pragma unroll 1
    for (int level = ngroups ? 0 : 1; level < 2; ++level) {
        if (!early) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the group row's stores have left
            __syncthreads();
            if (tid == 0) drawn_ = (int)__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        early = false;
        if (tid == 0) *flag = drawn_;
        __syncthreads();
        const int drawn = *flag;
        __syncthreads();
        if (drawn != count - 1) return;                        // (workgroup-uniform)
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // for the next launch
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        float s1 = 0.f, s2 = 0.f;
        for (int r0 = sub; r0 < count; r0 += TPC * UB) {
            float v[UB][2];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int r = r0 + u * TPC;
                const unsigned off = (cok && r < count) ? (unsigned)(((long long)(first + r) * a.N + n) * 8) : kOOB;
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    v[u][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsS, off + 4 * e, 0, 16 /* sc1: written by other XCDs */));
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {                      // rows past the range read as zeros
                s1 += v[u][0];
                s2 += v[u][1];
            }
        }
        sh[(sub * BN + col) * 2] = s1;
        sh[(sub * BN + col) * 2 + 1] = s2;
        __syncthreads();
        t1 = 0.f; t2 = 0.f;
#pragma unroll
        for (int u = 0; u < TPC; ++u) {
            t1 += sh[(u * BN + col) * 2];
            t2 += sh[(u * BN + col) * 2 + 1];
        }
        __syncthreads();
        if (level == 0) {
            if (sub == 0 && cok) {
                float* o = a.bnb_partials + ((long long)(tiles_m + grp) * a.N + n) * 2;
                __hip_atomic_store(o, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(o + 1, t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            first = tiles_m;
            count = ngroups;
            ticket = a.bnb_tickets + tn;
        }
    }
    if (sub == 0 && cok) {                                     // the launch's last workgroup of this column tile
        float* coef = a.bnb_partials + (long long)(tiles_m + ngroups) * a.N * 2;
        a.bnb_dbeta[n] = t1;
        a.bnb_dgamma[n] = t2;
        const float gm = a.bnb_gamma ? a.bnb_gamma[n] : 1.f, is = a.bnb_invstd[n];
        const float ca = gm * is;
        coef[n] = ca;
        coef[a.N + n] = a.bnb_train ? -ca * is * t2 * a.bnb_invm : 0.f;
        coef[2 * a.N + n] = a.bnb_train ? -ca * t1 * a.bnb_invm : 0.f;
        coef[3 * a.N + n] = a.bnb_mean[n];
    }
}
// SK: the launch may be a split-K one (false: the slab / hand-off code is left out -- its registers count against every launch).
// PS: persistent-capable (the tile loop and the next-tile prefetch are compiled in; costs registers, so it is a variant).
// BNS: the epilogue can also take the BatchNorm-backward sums of the result (a.bnb_partials; data gradients of ResNet blocks).
// BNF: a colstats launch can also finish the BatchNorm statistics (a.bnf_tickets; forward convolutions of ResNet blocks).
template <int BM, int BN, int BK, int AK, int BKIND, bool VEC, int WGM = 2, bool RS = false, int NS = 3, bool SK = true, bool PS = false,
          bool BNS = false, bool BNF = false>
__device__ __forceinline__ void gemm_bf16_body(const GemmArgs& a, const int bx, const int bz) {
    typedef bf16_t T;
    constexpr int ESZ = 2;
    constexpr bool A_RC = a_is_rc(AK), B_RC = b_is_rc(BKIND);
    constexpr bool ROWSUM = A_RC || RS;
    constexpr int NW = 2 * WGM;
    constexpr int WM = BM / WGM, WN = BN / 2, FM = WM / 16, FN = WN / 16;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int A_NI = A_BYTES / 1024 / NW, B_NI = B_BYTES / 1024 / NW;   // DMA instructions per wave per tile
    constexpr int CPR = BK / 8;
    static_assert(A_NI >= 1 && B_NI >= 1, "tile too small for its DMA waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char lds_char;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, l15 = lane & 15;
    int tm, tn;
    const unsigned long long t_entry = __builtin_readcyclecounter();   // stamp 0, written with stamp 1 (no argument load here)
    {   // one batch of scalar loads for everything the set-up reads: left to the scheduler they were 5-7 dependent
        // s_load + s_waitcnt round trips in front of the first DMA (tools/gemm_stamps.py: 1.3-1.45 us of set-up)
        const int q0 = a.tiles_m, q1 = a.tiles_n, q2 = a.group_m, q3 = a.split_k, q4 = a.k_per_split, q5 = a.batch_inner;
        const int q6 = a.M, q7 = a.N, q8 = a.K, q9 = a.lda, q10 = a.ldb;
        const char* pa = a.A;
        const char* pb = a.B;
        const unsigned long long ba = a.a_bytes, bb = a.b_bytes;
        asm volatile("" ::"s"(q0), "s"(q1), "s"(q2), "s"(q3), "s"(q4), "s"(q5), "s"(q6), "s"(q7), "s"(q8), "s"(q9), "s"(q10),
                     "s"(pa), "s"(pb), "s"(ba), "s"(bb));
    }
    int m0, n0;
    const int z = bz;

    long long a_boff = 0, b_boff = 0, d_boff = 0;
    int kbeg = 0, kend = a.K;
    if (a.split_k > 1) {
        kbeg = z * a.k_per_split;
        kend = min(a.K, kbeg + a.k_per_split);
    } else {
        const int zo = z / a.batch_inner, zi = z - zo * a.batch_inner;
        a_boff = zo * a.a_bs0 + zi * a.a_bs1;
        b_boff = zo * a.b_bs0 + zi * a.b_bs1;
        d_boff = zo * a.d_bs0 + zi * a.d_bs1;
    }
    const unsigned long long a_rem = a.a_bytes - (unsigned long long)a_boff * ESZ;
    const unsigned long long b_rem = a.b_bytes - (unsigned long long)b_boff * ESZ;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(a.A + a_boff * ESZ, (unsigned)min(a_rem, 0x7fffff00ull));
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(a.B + b_boff * ESZ, (unsigned)min(b_rem, 0x7fffff00ull));

    // ---- per-thread staging state: one LDS slot per DMA instruction ---------------------------
    // slot s = (wave*NI + i)*64 + lane.  K-contiguous tile: row = s / CPR, physical chunk = s % CPR.
    // Row-contiguous tile [BK][BR]: k = s / (BR/8), physical chunk = s % (BR/8).
    KcRow<A_RC ? HS_A_KC : AK> a_rows[A_RC ? 1 : A_NI];
    KcRow<HS_A_KC> b_rows[B_RC ? 1 : B_NI];
    RcCol a_cols[A_RC ? A_NI : 1], b_cols[B_RC ? B_NI : 1];
    int a_kl[A_NI], b_kl[B_NI];   // KC: logical k-chunk (x8) of the slot; RC: local k row of the slot
    constexpr int NDMA = A_NI + B_NI;
    static_assert(NS >= 3 && NS <= 8 && (NS - 1) * NDMA <= 63, "ring depth: the vmcnt field holds 6 bits");
    constexpr int KS = BK / 32;
    int ntiles = 0;
    // stride-2 dgrad with parity-major rows: when every row of this tile is in one parity class, only the filter taps
    // (r, s) with (class_h + pad - r) and (class_w + pad - s) even can reach a stored output pixel; the K walk keeps just
    // those taps (1, 2, 2 or 4 of 9 for a 3x3 filter; 1 or 0 of 1 for a 1x1).  taps: 4 bits per kept tap index.
    unsigned long long taps = 0;
    int tiles_per_tap = 1;
    bool filtered = false;
    // everything that depends on WHICH output tile this workgroup works on (vb = the tile's position in launch order): a
    // persistent launch (a.persist) calls it once per tile
    auto setup_tile = [&](const int vb) {
    tile_from_block(a, tm, tn, vb);
    m0 = tm * BM;
    n0 = tn * BN;
#pragma unroll
    for (int i = 0; i < A_NI; ++i) {
        const int s = (wave * A_NI + i) * 64 + lane;
        if constexpr (!A_RC) {
            const int r = s / CPR;
            kc_row_setup<AK>(a, m0 + r, a_rows[i]);
            a_kl[i] = ((s % CPR) ^ kc_swz<CPR>(r)) * 8;
        } else {
            const int k = s / (BM / 8);
            a_kl[i] = k;
            rc_col_setup<AK, true>(a, m0 + rc_logical_chunk<BM>(k, s % (BM / 8)) * 8, a_cols[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < B_NI; ++i) {
        const int s = (wave * B_NI + i) * 64 + lane;
        if constexpr (!B_RC) {
            const int r = s / CPR;
            const int n = n0 + r;
            b_rows[i].valid = n < a.N;
            b_rows[i].base = n * a.ldb;
            b_rows[i].hb = b_rows[i].wb = 0;
            b_kl[i] = ((s % CPR) ^ kc_swz<CPR>(r)) * 8;
        } else {
            const int k = s / (BN / 8);
            b_kl[i] = k;
            rc_col_setup<BKIND, false>(a, n0 + rc_logical_chunk<BN>(k, s % (BN / 8)) * 8, b_cols[i]);
        }
    }
    ntiles = (kend - kbeg + BK - 1) / BK;
    taps = 0;
    tiles_per_tap = 1;
    filtered = false;
    if constexpr (AK == HS_A_DGRAD) {
        if (a.parity) {
            const int c_first = parity_class(a, m0), c_last = parity_class(a, min(m0 + BM, a.M) - 1);
            if (c_first == c_last) {
                filtered = true;
                tiles_per_tap = a.g.K / BK;
                int kept = 0;
                for (int r = 0; r < a.g.R; ++r)
                    for (int q = 0; q < a.g.S; ++q)
                        if ((((c_first >> 1) + a.g.pad - r) & 1) == 0 && (((c_first & 1) + a.g.pad - q) & 1) == 0) {
                            taps |= (unsigned long long)(r * a.g.S + q) << (4 * kept);
                            ++kept;
                        }
                ntiles = kept * tiles_per_tap;
            }
        }
    }
    };   // setup_tile
    setup_tile(bx);

    auto stage_dma = [&](int buf, int k0) {
        KTile kta = {0, 0, 0}, ktb = {0, 0, 0};
        if constexpr (AK == HS_A_CONV) kta = ktile_rsc(k0, a.g.C, a.g.S);
        if constexpr (AK == HS_A_DGRAD) kta = ktile_rsc(k0, a.g.K, a.g.S);
        if constexpr (BKIND == HS_B_WDGRAD) ktb = ktile_rsc(k0, a.g.K, a.g.S);
        lds_char* la = (lds_char*)(smem) + buf * STAGE + wave * (A_NI * 1024);
        lds_char* lb = (lds_char*)(smem) + buf * STAGE + A_BYTES + wave * (B_NI * 1024);
#pragma unroll
        for (int i = 0; i < A_NI; ++i) {
            unsigned off;
            if constexpr (!A_RC) off = kc_chunk_off<AK, ESZ>(a, a_rows[i], kta, k0 + a_kl[i], kend);
            else off = rc_chunk_off<AK, true, ESZ>(a, a_cols[i], kta, k0 + a_kl[i], kend);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(la + i * 1024), 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_NI; ++i) {
            unsigned off;
            if constexpr (!B_RC) off = kc_chunk_off<HS_A_KC, ESZ>(a, b_rows[i], ktb, k0 + b_kl[i], kend);
            else off = rc_chunk_off<BKIND, false, ESZ>(a, b_cols[i], ktb, k0 + b_kl[i], kend);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(lb + i * 1024), 16, off, 0, 0, 0);
        }
    };

    f32x4 acc[FM][FN];
    // bias gradient of a weight-gradient GEMM (A = dY stored [k][m]): the column sums of dY are one more output column,
    // dY^T . 1 -- the waves of the first tile column feed their A fragments to one extra MFMA against a fragment of ones.
    f32x4 accb[ROWSUM ? FM : 1];
    bool do_rowsum = false;

    // 3-slot LDS ring, software pipelined at two levels.
    //  * tiles: when the waves meet at the barrier of tile t, tile t+1 has landed, tile t+2 is in flight and the
    //    slot of tile t (whose fragment reads every wave has drained) is refilled with tile t+3.  The wait before
    //    the barrier is COUNTED (all but the newest in-flight tile of this wave; a __syncthreads() would drain
    //    vmcnt to 0).
    //  * fragments: the ds_reads of k-phase p+1 (or of phase 0 of the next tile, right after the barrier) are issued
    //    before the MFMAs of phase p, into the other half of a double-buffered fragment set, so the LDS latency
    //    hides under 8..16 MFMAs instead of stalling every MFMA pair.
    auto k_of = [&](int t) -> int {          // first k of the t-th tile of this workgroup's K walk
        if constexpr (AK == HS_A_DGRAD) {
            if (filtered) {
                const int ti = t / tiles_per_tap;
                return (int)((taps >> (4 * ti)) & 15) * a.g.K + (t - ti * tiles_per_tap) * BK;
            }
        }
        return kbeg + t * BK;
    };

    auto load_frags = [&](int slot, int ks, bf16x8 (&af)[FM], bf16x8 (&bfr)[FN]) {
        const char* sa = smem + slot * STAGE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int r0 = wm * WM + i * 16;
            if constexpr (!A_RC) {
                af[i] = *(const bf16x8*)(sa + kc_off_bf16<BK>(r0 + l15, ks * 4 + g));
            } else {
                const int kb = ks * 32 + 8 * g + (l15 >> 2);
                const int col = r0 + 4 * (lane & 3);
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(sa + rc_off_bf16<BM>(kb, col)));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(sa + rc_off_bf16<BM>(kb + 4, col)));
                af[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int r0 = wn * WN + j * 16;
            if constexpr (!B_RC) {
                bfr[j] = *(const bf16x8*)(sb + kc_off_bf16<BK>(r0 + l15, ks * 4 + g));
            } else {
                const int kb = ks * 32 + 8 * g + (l15 >> 2);
                const int col = r0 + 4 * (lane & 3);
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(sb + rc_off_bf16<BN>(kb, col)));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(sb + rc_off_bf16<BN>(kb + 4, col)));
                bfr[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
        }
    };
    auto mma = [&](const bf16x8 (&af)[FM], const bf16x8 (&bfr)[FN]) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        if constexpr (ROWSUM) {
            if (do_rowsum) {
                const s16x4 o4 = {0x3f80, 0x3f80, 0x3f80, 0x3f80};      // bf16 1.0
                const bf16x8 ones = __builtin_bit_cast(bf16x8, __builtin_shufflevector(o4, o4, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int i = 0; i < FM; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[i], accb[i], 0, 0, 0);
            }
        }
    };

    int cur = 0;   // slot of tile t
    // tile boundary: this wave's reads of slot `cur` are complete and its DMAs of tile t+1 have landed (tiles t+2 ..
    // t+NS-1 may still be in flight); after the barrier so have everyone else's, and slot `cur` takes tile t+NS.
    auto next_tile = [&](int t) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wait_vm_tiles<NDMA>(min(NS - 1, ntiles - 1 - t) - 1);
        __builtin_amdgcn_s_barrier();
        if (t + NS < ntiles) stage_dma(cur, k_of(t + NS));
        cur = cur == NS - 1 ? 0 : cur + 1;
    };

    // the first ring slots of the current tile's K walk
    auto issue_prologue = [&]() {
        if (ntiles > 0) {
            stage_dma(0, k_of(0));
#pragma unroll
            for (int q = 1; q < NS; ++q)
                if (q < ntiles) stage_dma(q, k_of(q));
        }
    };
    // BNS: the rider's operands (the BatchNorm input c; the saved block output and the skip gradient in the block-output
    // form) do not depend on this GEMM: fetched HERE, in front of the K walk (they are the oldest entries of the memory
    // pipe, so the first counted wait covers them), they cost 8 registers per operand on a 64x64 tile instead of one
    // exposed round trip per fragment behind the last MFMA.  Split-K launches keep the loads in the rider: only the
    // workgroup that finishes a tile runs it.
    constexpr int PFM = BNS ? FM : 1, PFN = BNS ? FN : 1;
    u32x2 pf_c[PFM][PFN], pf_y[PFM][PFN], pf_r[PFM][PFN];
    bool pf = false;
    if constexpr (BNS) {
        if (a.bnb_partials && a.split_k == 1) {
            pf = true;
            const bool wy = a.bnb_y != nullptr, wr = wy && a.residual != nullptr;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int m = m0 + wm * WM + i * 16 + l15, n = n0 + wn * WN + j * 16 + 4 * g;
                    const bool live = m < a.M && n < a.N;
                    pf_c[i][j] = pf_y[i][j] = pf_r[i][j] = u32x2{0u, 0u};
                    if (live) {
                        pf_c[i][j] = *(const HS_GLOBAL u32x2*)((const HS_GLOBAL T*)a.bnb_x + (long long)m * a.ldd + n);
                        if (wy) pf_y[i][j] = *(const HS_GLOBAL u32x2*)((const HS_GLOBAL T*)a.bnb_y + (long long)m * a.ldd + n);
                        if (wr) pf_r[i][j] = *(const HS_GLOBAL u32x2*)((const HS_GLOBAL T*)a.residual + (long long)m * a.ldr + n);
                    }
                }
        }
    }
    issue_prologue();
    HS_STAMP(1);
    if (a.stamps && threadIdx.x == 0) a.stamps[((long long)blockIdx.z * gridDim.x + blockIdx.x) * 6] = t_entry;
    // Persistent launches (a.persist = grid size > 0; plain bf16 GEMMs / convolutions with more tiles than the chip holds at
    // once): the workgroup walks tiles vb, vb + grid, ... and issues the NEXT tile's address set-up and first ring slots
    // before the CURRENT tile's epilogue, so the ~1.9 us from tile entry to the first landed operands (tools/gemm_stamps.py)
    // run under the epilogue's stores instead of in front of every K loop.
    const int pstride = PS ? a.persist : 0;
    int vb = bx;
    bool first_tile = true;
    for (;;) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (ROWSUM) {
        do_rowsum = a.rowsum[0] != nullptr && tn == 0 && wn == 0;
#pragma unroll
        for (int i = 0; i < FM; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    cur = 0;
    if (ntiles > 0) {
        // first tile: counted wait (all but the newest in-flight slots); later tiles: the epilogue's loads and stores sit
        // between this K walk's DMAs in the memory pipe, so drain everything (the slots have had the whole epilogue to land)
        if (first_tile) wait_vm_tiles<NDMA>(min(NS, ntiles) - 1);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        HS_STAMP(2);
        bf16x8 a0[FM], b0[FN], a1[FM], b1[FN];
        load_frags(0, 0, a0, b0);
        static_assert(KS == 1 || KS == 2, "BK must be 32 or 64");
        if constexpr (KS == 2) {
            for (int t = 0; t + 1 < ntiles; ++t) {
                load_frags(cur, 1, a1, b1);
                mma(a0, b0);
                next_tile(t);
                load_frags(cur, 0, a0, b0);
                mma(a1, b1);
            }
            load_frags(cur, 1, a1, b1);
            mma(a0, b0);
            mma(a1, b1);
        } else if constexpr (BM * BN >= 128 * 128) {
            // big tiles with BK = 32: one fragment set (the second one costs 32-48 registers, which is the difference between
            // two and three resident workgroups per CU; the other workgroups' MFMAs cover this one's LDS reads)
            for (int t = 0; t + 1 < ntiles; ++t) {
                mma(a0, b0);
                next_tile(t);
                load_frags(cur, 0, a0, b0);
            }
            mma(a0, b0);
        } else {
            for (int t = 0; t + 1 < ntiles; ++t) {
                next_tile(t);
                load_frags(cur, 0, a1, b1);
                mma(a0, b0);
#pragma unroll
                for (int i = 0; i < FM; ++i) a0[i] = a1[i];
#pragma unroll
                for (int j = 0; j < FN; ++j) b0[j] = b1[j];
            }
            mma(a0, b0);
        }
    }

    HS_STAMP(3);
    // ---- optional: column statistics of this tile for a following BatchNorm --------------------------
    // Rows of the tile beyond M hold exact zeros (their operand rows were zero-filled), so plain sums over the whole
    // tile with the true row count give the tile's (count, mean, M2).  Per column: in-lane sum over the FM row
    // fragments, xor-shuffles over the 16 row lanes, then the two row waves fold through LDS (the ring is free now).
    if (a.colstats) {
        float s1[FN][4], s2[FN][4];
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float u = 0.f, w = 0.f;
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const float v = acc[i][j][e] * a.alpha;
                    u += v;
                    w = fmaf(v, v, w);
                }
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    u += __shfl_xor(u, o, 64);
                    w += __shfl_xor(w, o, 64);
                }
                s1[j][e] = u;
                s2[j][e] = w;
            }
        __syncthreads();                       // every wave is done with the last tile's fragments
        float* sh = (float*)smem;              // [WGM row waves][BN][2]
        if (l15 == 0) {
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = wn * WN + j * 16 + 4 * g + e;
                    sh[(wm * BN + c) * 2 + 0] = s1[j][e];
                    sh[(wm * BN + c) * 2 + 1] = s2[j][e];
                }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < a.N) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < WGM; ++w) {
                t1 += sh[(w * BN + tid) * 2];
                t2 += sh[(w * BN + tid) * 2 + 1];
            }
            const float cnt = (float)min(BM, a.M - m0);
            float* o = a.colstats + ((long long)tm * a.N + n0 + tid) * 3;
            const float mean = t1 / cnt, m2 = fmaxf(t2 - t1 * t1 / cnt, 0.f);
            if constexpr (BNF) {               // read by another workgroup of this launch: write-through (see the hand-off below)
                __hip_atomic_store(o, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(o + 1, mean, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(o + 2, m2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                o[0] = cnt;
                o[1] = mean;
                o[2] = m2;
            }
        }
    }
    // (BNF) this tile's arrival at the statistics hand-off (see the end of the body) is drawn here, so that the counter's
    // round trip runs under the epilogue and only the three partial stores above are waited for, not the result's stores
    int bnf_drawn = 0;
    if constexpr (BNF && !PS) {
        if (a.bnf_tickets) {
            const int ng = stat_groups(a.tiles_m);
            unsigned* ticket = ng ? a.bnf_tickets + a.tiles_n + tn * ng + tm / kStatGroup : a.bnf_tickets + tn;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the partial stores have left
            __syncthreads();
            if (tid == 0) bnf_drawn = (int)__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // ---- epilogue: lane owns m = .. + l15, n = .. + 4g + {0..3} ---------------------------------
    const unsigned epi = epi_flags(a);
    const int split_k = SK ? a.split_k : 1, argM = a.M, argN = a.N;
    float* splitk_ws = a.splitk_ws;
    if constexpr (ROWSUM) {
        if (do_rowsum && g == 0) {           // every n of the ones-operand holds the same sum: lanes 0..15 write one row each
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = m0 + wm * WM + i * 16 + l15;
                if (m < a.M) {
                    const int seg = a.seg_rows > 0 ? m / a.seg_rows : 0;
                    a.rowsum[seg][m - seg * (a.seg_rows > 0 ? a.seg_rows : 0)] = accb[i][0];
                }
            }
        }
    }
    if (split_k > 1 && !a.tickets) {                              // slabs for the separate reduce pass
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = m0 + wm * WM + i * 16 + l15;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int n = n0 + wn * WN + j * 16 + 4 * g;
                if (m < argM && n < argN) {
                    HS_GLOBAL float* w = (HS_GLOBAL float*)splitk_ws + ((long long)z * argM + m) * argN + n;
                    if (n + 3 < argN && (argN & 3) == 0)
                        *(HS_GLOBAL f32x4*)w = f32x4{acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    else
                        for (int e = 0; e < 4 && n + e < argN; ++e) w[e] = acc[i][j][e];
                }
            }
        }
        HS_STAMP(4);
        return;
    }
    if (split_k > 1) {
        // In-launch split-K reduction (the separate splitk_reduce pass cost 65 launches / 0.63 ms per C2 step).
        // Protocol of one hand-off (cdna_hip_programming.md, "In-launch split-K reduction" / Guideline 16, counter form):
        //   producers: slab stores WRITE-THROUGH (sc1: the bytes leave this XCD's L2, so no release fence) -> every wave
        //   s_waitcnt vmcnt(0) -> workgroup barrier -> ONE lane: relaxed agent-scope ticket add; the workgroup whose add comes
        //   last is the consumer: one agent-scope acquire (drops this CU's stale L1 lines) -> vmcnt(0) -> barrier -> all
        //   waves read the slabs in a fixed order -> re-zero the ticket for the next launch.
        // Two levels: the consumer of a hand-off is alone on its tile while everyone else has retired, so its slab reads are
        // the tail of the launch, one memory round trip per batch of loads (a 64-way split read slab after slab by one
        // workgroup measured 60+ us of tail on a 30 us stream).  Slices therefore hand off in GROUPS of kSplitGroup (8)
        // consecutive slices; the last slice of a group sums that group (one batch) and publishes the group's slab; the last
        // GROUP of a tile sums the <= 8 group slabs and runs the epilogue.  The tail is two batches deep instead of split_k
        // slabs, and the sum has one fixed association (slices in order inside a group, groups in order): deterministic
        // whichever slice arrives last.  Placement independent: nothing depends on where or in which order slices run.
        constexpr int NF = FM * FN;
        // slabs whose loads are in flight together (12-16 loads per lane): about what fits in the registers the K loop's
        // operand fragments have just vacated -- registers beyond the K loop's own need lower the occupancy of EVERY launch of
        // the kernel (8 / 4 / 2 slabs measured: 202-256 VGPRs instead of 79-176 and the whole family 11 % slower)
        constexpr int U = NF <= 4 ? 3 : NF <= 8 ? 2 : 1;
        static_assert(NF % 4 == 0 && U * NF <= 60, "slab loads in flight: the vmcnt field holds 6 bits");
        const int ngroups = (split_k + kSplitGroup - 1) / kSplitGroup;
        const int nslabs = split_k + (ngroups > 1 ? ngroups : 0);     // slice slabs, then group slabs
        const unsigned ws_bytes = (unsigned)min((unsigned long long)nslabs * argM * argN * 4ull, 0x7fffff00ull);
        const __amdgpu_buffer_rsrc_t rsW = make_rsrc(splitk_ws, ws_bytes);
        const unsigned long long wsa = (unsigned long long)splitk_ws;
        const u32x4 rsw = {(unsigned)wsa, (unsigned)(wsa >> 32) & 0xffffu, ws_bytes, 0x00020000u};
        const bool vecw = (argN & 3) == 0;
        const unsigned slab = (unsigned)((long long)argM * argN * 4);
        unsigned eoff[NF];                                           // byte offset of this lane's fragment f inside a slab
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = m0 + wm * WM + i * 16 + l15;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int n = n0 + wn * WN + j * 16 + 4 * g;
                eoff[i * FN + j] = (m < argM && n < argN) ? (unsigned)(((long long)m * argN + n) * 4) : kOOB;
            }
        }
        int* flag = (int*)smem;                                      // the operand ring is free by now
        auto publish = [&](int slab_idx, unsigned* ticket) -> int {  // store acc as slab `slab_idx`, arrive; returns the ticket drawn
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (eoff[f] == kOOB) continue;
                const unsigned off = (unsigned)slab_idx * slab + eoff[f];
                const f32x4& v4 = acc[f / FN][f % FN];
                if (vecw) {
                    const u32x4 v = {__float_as_uint(v4[0]), __float_as_uint(v4[1]), __float_as_uint(v4[2]), __float_as_uint(v4[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsW, off, 0, 16 /* sc1: write-through */);
                } else {
                    const int n = n0 + wn * WN + (f % FN) * 16 + 4 * g;
                    for (int e = 0; e < 4 && n + e < argN; ++e)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v4[e]), rsW, off + 4 * e, 0, 16);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its own stores
            __syncthreads();
            if (tid == 0) *flag = (int)__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const int drawn = *flag;
            __syncthreads();                                          // the flag word is reused by the next hand-off
            return drawn;
        };
        auto gather = [&](int first, int count) {                    // acc = slab[first] + slab[first + 1] + ... (in order)
            if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (vecw) {
                // The compiler will not keep more than a few of these loads in flight on its own (it serialises them against
                // the dependent adds to save registers), so the loads and their COUNTED waits are spelled out: slab u is
                // added once at most (U - 1 - u) slabs' loads are still outstanding.  Slabs past `count` read as zeros.
                for (int s0 = 0; s0 < count; s0 += U) {
                    u32x4 x[U][NF];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const bool live = s0 + u < count;
#pragma unroll
                        for (int f = 0; f < NF; ++f) {
                            const unsigned off = (live && eoff[f] != kOOB) ? (unsigned)(first + s0 + u) * slab + eoff[f] : kOOB;
                            if constexpr (U == 1) x[u][f] = buf_load16(rsW, off);      // one slab: the compiler's own schedule will do
                            else asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(x[u][f]) : "v"(off), "s"(rsw) : "memory");
                        }
                    }
                    static_for<0, U>([&](auto uc) {
                        constexpr int u = decltype(uc)::value;
                        if constexpr (U > 1) {
#pragma unroll
                            for (int f = 0; f < NF; f += 4)
                                asm volatile("s_waitcnt vmcnt(%4)"
                                             : "+v"(x[u][f]), "+v"(x[u][f + 1]), "+v"(x[u][f + 2]), "+v"(x[u][f + 3])
                                             : "n"((U - 1 - u) * NF));
                        }
#pragma unroll
                        for (int f = 0; f < NF; ++f)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[f / FN][f % FN][e] += __uint_as_float(x[u][f][e]);
                    });
                }
            } else {
                for (int sidx = first; sidx < first + count; ++sidx) {
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        if (eoff[f] == kOOB) continue;
                        const int n = n0 + wn * WN + (f % FN) * 16 + 4 * g;
                        const HS_GLOBAL float* w = (const HS_GLOBAL float*)((const HS_GLOBAL char*)splitk_ws + (long long)sidx * slab + eoff[f]);
                        for (int e = 0; e < 4 && n + e < argN; ++e) acc[f / FN][f % FN][e] += w[e];
                    }
                }
            }
        };
        // level 0: the slices of a group (of the whole tile when there is one group); level 1: the groups of the tile.  One
        // copy of the code, walked once or twice (emitted twice it held both levels' registers: 231 instead of 149 VGPRs).
        const int tile_id = tm * a.tiles_n + tn;
        const int grp = z / kSplitGroup;
        int slab_idx = z, first = ngroups > 1 ? grp * kSplitGroup : 0;
        int count = ngroups > 1 ? min(kSplitGroup, split_k - first) : split_k;
        unsigned* ticket = ngroups > 1 ? a.tickets + a.tiles_m * a.tiles_n + tile_id * ngroups + grp : a.tickets + tile_id;
#pragma unroll 1
        for (int level = ngroups > 1 ? 0 : 1; level < 2; ++level) {
            if (publish(slab_idx, ticket) != count - 1) {
                HS_STAMP(4);
                return;                                               // (workgroup-uniform)
            }
            gather(first, count);
            if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            slab_idx = split_k + grp;
            first = split_k;
            count = ngroups;
            ticket = a.tickets + tile_id;
        }
    }
    // ---- optional: BatchNorm-backward sums of this tile (the result is d relu(bn(c)); plain epilogue, alpha = 1) -----------
    // Block-output form (a.bnb_y): the result + a.residual is the gradient of relu(bn(c) + identity); the residual is added
    // HERE (the epilogue below then runs without it) and the mask is y > 0 on the saved block output.
    bool res_consumed = false;
    int bnb_drawn = 0;
    if constexpr (BNS) {
        if (a.bnb_partials) {
            const bool from_y = a.bnb_y != nullptr;
            res_consumed = from_y && a.residual != nullptr;
            float s1[FN][4], s2[FN][4];
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int n = n0 + wn * WN + j * 16 + 4 * g;
                const bool nok = n < argN;
                f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc, mu = sc, is = sc;
                if (nok) {
                    if (!from_y) {
                        sc = *(const HS_GLOBAL f32x4*)(a.bnb_scale + n);
                        sh = *(const HS_GLOBAL f32x4*)(a.bnb_shift + n);
                    }
                    mu = *(const HS_GLOBAL f32x4*)(a.bnb_mean + n);
                    is = *(const HS_GLOBAL f32x4*)(a.bnb_invstd + n);
                }
                float u[4] = {0.f, 0.f, 0.f, 0.f}, w[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int m = m0 + wm * WM + i * 16 + l15;
                    float cv[4] = {0.f, 0.f, 0.f, 0.f}, yv[4] = {0.f, 0.f, 0.f, 0.f};
                    const bool live = nok && m < argM;
                    auto unpack = [](const u32x2 v, float* f) {
                        f[0] = __uint_as_float(v[0] << 16);
                        f[1] = __uint_as_float(v[0] & 0xffff0000u);
                        f[2] = __uint_as_float(v[1] << 16);
                        f[3] = __uint_as_float(v[1] & 0xffff0000u);
                    };
                    if (pf) {
                        unpack(pf_c[i][j], cv);
                        if (from_y) {
                            unpack(pf_y[i][j], yv);
                            if (res_consumed) {
                                float rv[4];
                                unpack(pf_r[i][j], rv);
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[i][j][e] += rv[e];
                            }
                        }
                    } else {
                    if (live) load4<T>(a.bnb_x, (long long)m * a.ldd + n, true, 4, cv);
                    if (from_y && live) {
                        load4<T>(a.bnb_y, (long long)m * a.ldd + n, true, 4, yv);
                        if (res_consumed) {
                            float rv[4];
                            load4<T>(a.residual, (long long)m * a.ldr + n, true, 4, rv);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[i][j][e] += rv[e];
                        }
                    }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gq = (float)(bf16_t)acc[i][j][e];                      // the value the epilogue stores
                        const bool on = from_y ? yv[e] > 0.f : fmaf(cv[e], sc[e], sh[e]) > 0.f;
                        const float dz = (live && on) ? gq : 0.f;
                        u[e] += dz;
                        w[e] = fmaf(dz, (cv[e] - mu[e]) * is[e], w[e]);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) {
                        u[e] += __shfl_xor(u[e], o, 64);
                        w[e] += __shfl_xor(w[e], o, 64);
                    }
                    s1[j][e] = u[e];
                    s2[j][e] = w[e];
                }
            }
            __syncthreads();                       // every wave is done with the operand ring
            float* sh2 = (float*)smem;             // [WGM row waves][BN][2]
            if (l15 == 0) {
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = wn * WN + j * 16 + 4 * g + e;
                        sh2[(wm * BN + c) * 2 + 0] = s1[j][e];
                        sh2[(wm * BN + c) * 2 + 1] = s2[j][e];
                    }
            }
            __syncthreads();
            if (tid < BN && n0 + tid < argN) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int w = 0; w < WGM; ++w) {
                    t1 += sh2[(w * BN + tid) * 2];
                    t2 += sh2[(w * BN + tid) * 2 + 1];
                }
                float* o = a.bnb_partials + ((long long)tm * argN + n0 + tid) * 2;
                if (a.bnb_tickets) {               // read by another workgroup of this launch: write-through
                    __hip_atomic_store(o, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(o + 1, t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    o[0] = t1;
                    o[1] = t2;
                }
            }
            __syncthreads();                       // (persistent variants reuse the ring right after)
            if (a.bnb_tickets) {                   // arrival drawn here: the counter's round trip runs under the epilogue
                const int ng = stat_groups(a.tiles_m);
                unsigned* ticket = ng ? a.bnb_tickets + a.tiles_n + tn * ng + tm / kStatGroup : a.bnb_tickets + tn;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the partial stores have left
                __syncthreads();
                if (tid == 0) bnb_drawn = (int)__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // ---- next tile's set-up and first operand slots go out before this tile's epilogue (persistent launches) ----------
    const int em0 = m0, en0 = n0;
    const int nvb = vb + pstride;
    const bool more = PS && pstride > 0 && nvb < a.tiles_m * a.tiles_n;
    if (more) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every wave is done reading this tile's last fragments
        __builtin_amdgcn_s_barrier();
        setup_tile(nvb);
        issue_prologue();
    }
    // whole tile inside the matrix and 4-wide storable: one of the feature sets the training / inference steps use gets
    // branch-free code; anything else (edge tiles, rare combinations) takes the generic body
    const unsigned epi_e = res_consumed ? (epi & ~(unsigned)EPI_RES_POST) : epi;     // (the sums' rider already added the residual)
    const bool full = !a.epi_generic && (epi & EPI_VEC) && em0 + BM <= argM && en0 + BN <= argN;
    bool done = false;
    const int ze = split_k > 1 ? 0 : z;                       // batch index seen by the epilogue (a split-K launch has no batch)
    if (full) {
        done = true;
#define HS_EPI_CASE(F) case (F): run_epilogue<T, (F), true, FM, FN, WM, WN>(a, epi_e, acc, em0, en0, wm, wn, l15, g, d_boff, ze); break
        const bool v16 = epi_e & EPI_VEC16;
        const unsigned key = epi_e & ~EPI_VEC16;
        if (!v16 && !(key & EPI_OUT_F32)) done = false;      // bf16 rows that cannot take 16-byte stores: generic body
        else switch (key) {
            HS_EPI_CASE(EPI_VEC);                                                     // plain bf16 result
            HS_EPI_CASE(EPI_VEC | EPI_BIAS);                                          // Linear
            HS_EPI_CASE(EPI_VEC | EPI_BIAS | EPI_GELU | EPI_PREACT);                  // FFN up-projection
            HS_EPI_CASE(EPI_VEC | EPI_BIAS | EPI_RES_POST);                           // dense + residual (eval / p = 0)
            HS_EPI_CASE(EPI_VEC | EPI_BIAS | EPI_DROP | EPI_RES_POST);                // dense + hidden dropout + residual
            HS_EPI_CASE(EPI_VEC | EPI_MUL_GELU);                                      // data gradient through GELU
            HS_EPI_CASE(EPI_VEC | EPI_RES_POST);                                      // data gradient + skip gradient
            HS_EPI_CASE(EPI_VEC | EPI_OUT_F32);                                       // weight gradient (f32)
            HS_EPI_CASE(EPI_VEC | EPI_OUT_F32 | EPI_SEG);                             // fused Q/K/V weight gradient
            HS_EPI_CASE(EPI_VEC | EPI_PARITY);                                        // stride-2 conv data gradient
            HS_EPI_CASE(EPI_VEC | EPI_BIAS | EPI_COLSCALE | EPI_RELU);                // folded conv+BN+ReLU (inference)
            HS_EPI_CASE(EPI_VEC | EPI_BIAS | EPI_COLSCALE);                           // folded conv+BN
            HS_EPI_CASE(EPI_VEC | EPI_BIAS | EPI_COLSCALE | EPI_RELU | EPI_RES_PRE);  // folded conv+BN+add+ReLU
            default: done = false;
        }
#undef HS_EPI_CASE
    }
    if (!done) run_epilogue<T, -1, false, FM, FN, WM, WN>(a, epi_e, acc, em0, en0, wm, wn, l15, g, d_boff, ze);
    HS_STAMP(4);
    if constexpr (BNF && !PS) {
        if (a.bnf_tickets) bnf_handoff<WGM * 128, BN>(a, smem, tid, tm, tn, en0, bnf_drawn);
    }
    if constexpr (BNS && !PS) {
        if (a.bnb_partials && a.bnb_tickets) bnb_handoff<WGM * 128, BN>(a, smem, tid, tm, tn, en0, bnb_drawn);
    }
    if (!more) break;
    vb = nvb;
    first_tile = false;
    }   // tiles of this workgroup
}

template <int BM, int BN, int BK, int AK, int BKIND, bool VEC, int WGM = 2, bool RS = false, int NS = 3>
__global__ __launch_bounds__(WGM * 128) void gemm_bf16_kernel(const GemmArgs a) {
    gemm_bf16_body<BM, BN, BK, AK, BKIND, VEC, WGM, RS, NS>(a, blockIdx.x, blockIdx.z);
}
// data-gradient GEMMs that also take the following BatchNorm's backward sums (a.bnb_partials): 64x64 and 128x64 tiles
template <int BM, int BN, int AK, int BKIND>
__global__ __launch_bounds__(256) void gemm_bf16_bns_kernel(const GemmArgs a) {
    gemm_bf16_body<BM, BN, 64, AK, BKIND, true, 2, false, 3, true, false, true>(a, blockIdx.x, blockIdx.z);
}
// forward convolutions that also finish the following BatchNorm's statistics (a.colstats + a.bnf_tickets)
template <int BM, int BN, int BK, int AK, int BKIND>
__global__ __launch_bounds__(256) void gemm_bf16_bnf_kernel(const GemmArgs a) {
    gemm_bf16_body<BM, BN, BK, AK, BKIND, true, 2, false, 3, false, false, false, true>(a, blockIdx.x, blockIdx.z);
}
// persistent-capable variant (64x64 and 128x64 tiles): launched with a.persist = grid size when a GEMM has more tiles than
// the chip holds at once
template <int BM, int BN, int AK, int BKIND>
__global__ __launch_bounds__(256) void gemm_bf16_persistent_kernel(const GemmArgs a) {
    gemm_bf16_body<BM, BN, 64, AK, BKIND, true, 2, false, 3, false, true>(a, blockIdx.x, 0);
}
// The 128x128 tile with BK = 32 takes 48 KB of LDS: three workgroups fit a CU if their registers do (<= 168 per lane).  A
// 4096 x 3072 output is 768 tiles: one resident round on 256 CUs x 3 instead of one and a half on 256 x 2.
#ifndef HS_W3_RING
#define HS_W3_RING 3
#endif
template <int AK, int BKIND>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void gemm_bf16_kernel_w3(const GemmArgs a) {
    gemm_bf16_body<128, 128, 32, AK, BKIND, true, 2, false, HS_W3_RING, false>(a, blockIdx.x, blockIdx.z);
}

// Grouped launch: the workgroups of up to 64 independent GEMMs of one operand layout (64x64 tiles) in ONE grid.  The weight
// gradients of a ResNet backward are 53 such GEMMs with a long K and a small output: each alone is a short stream plus the
// split-K hand-off tail on a mostly idle chip; together they fill it and there is one tail.  first_wg[i] = first workgroup
// of problem i (ascending, first_wg[0] = 0, first_wg[n] = grid size); list[i] = its arguments, in device memory.
template <int AK, int BKIND>
__global__ __launch_bounds__(256) void gemm_bf16_grouped_kernel(const GemmArgs* __restrict__ list, const int* __restrict__ first_wg, int n) {
    const int bid = blockIdx.x, lane = threadIdx.x & 63;
    const int lo = lane < n ? first_wg[lane] : 0x7fffffff;
    const int p = __builtin_amdgcn_readfirstlane(__popcll(__ballot(lo <= bid)) - 1);
    const int local = bid - __builtin_amdgcn_readfirstlane(first_wg[p]);
    const GemmArgs& a = list[p];
    const int tiles = a.tiles_m * a.tiles_n;
    const int bz = local / tiles;
    gemm_bf16_body<64, 64, 64, AK, BKIND, true>(a, local - bz * tiles, bz);
}

// Grouped K-contiguous weight gradients (the four Linear layers of a BertLayer: dW = dY^T X on transposed operands, K = the
// 4096 tokens): alone each is 72-288 tiles on a chip with 256-768 slots (430 TFLOP/s); together, as 256x128 tiles with the
// bias gradients as row sums and no split-K code, they fill it (one GEMM of their combined size measured 611 TFLOP/s).
template <int AK, int BKIND>
__global__ __launch_bounds__(512) void gemm_bf16_grouped_big_kernel(const GemmArgs* __restrict__ list, const int* __restrict__ first_wg, int n) {
    const int bid = blockIdx.x, lane = threadIdx.x & 63;
    const int lo = lane < n ? first_wg[lane] : 0x7fffffff;
    const int p = __builtin_amdgcn_readfirstlane(__popcll(__ballot(lo <= bid)) - 1);
    const int local = bid - __builtin_amdgcn_readfirstlane(first_wg[p]);
    gemm_bf16_body<256, 128, 64, AK, BKIND, true, 4, true, 3, false>(list[p], local, 0);
}

// the same with up to four problems passed BY VALUE in the kernel arguments (no device-side table to keep current: callers
// whose operand addresses change every step -- autograd-allocated gradients -- group for free)
struct GemmArgsPack4 {
    GemmArgs a[4];
    int first[5];
    int n;
};
template <int AK, int BKIND>
__global__ __launch_bounds__(512) void gemm_bf16_grouped_big4_kernel(const GemmArgsPack4 pk) {
    const int bid = blockIdx.x;
    int p = 0;
    for (int i = 1; i < pk.n; ++i) p += bid >= pk.first[i];
    p = __builtin_amdgcn_readfirstlane(p);
    gemm_bf16_body<256, 128, 64, AK, BKIND, true, 4, true, 3, false>(pk.a[p], bid - pk.first[p], 0);
}

// ================================================================================================
// exact-f32 kernel (v_mfma_f32_32x32x2_f32); BK = 32
// ================================================================================================
template <int BM, int BN, int AK, int BKIND, bool VEC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs a) {
    typedef float T;
    constexpr int BK = 32, EPC = 4, ESZ = 4;
    constexpr bool A_RC = a_is_rc(AK), B_RC = b_is_rc(BKIND);
    constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 32, FN = WN / 32;
    constexpr int A_BYTES = BM * BK * 4, B_BYTES = BN * BK * 4, STAGE = A_BYTES + B_BYTES;
    constexpr int A_NCH = BM * BK / 4 / 256, B_NCH = BN * BK / 4 / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hh = lane >> 5;
    int tm, tn;
    tile_from_block(a, tm, tn, blockIdx.x);
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.z;

    long long a_boff = 0, b_boff = 0, d_boff = 0;
    int kbeg = 0, kend = a.K;
    if (a.split_k > 1) {
        kbeg = z * a.k_per_split;
        kend = min(a.K, kbeg + a.k_per_split);
    } else {
        const int zo = z / a.batch_inner, zi = z - zo * a.batch_inner;
        a_boff = zo * a.a_bs0 + zi * a.a_bs1;
        b_boff = zo * a.b_bs0 + zi * a.b_bs1;
        d_boff = zo * a.d_bs0 + zi * a.d_bs1;
    }
    const unsigned long long a_rem = a.a_bytes - (unsigned long long)a_boff * ESZ;
    const unsigned long long b_rem = a.b_bytes - (unsigned long long)b_boff * ESZ;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(a.A + a_boff * ESZ, (unsigned)min(a_rem, 0x7fffff00ull));
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(a.B + b_boff * ESZ, (unsigned)min(b_rem, 0x7fffff00ull));

    // K-contiguous staging: chunk id -> row = id % ROWS (lane-fastest: conflict-free LDS scatter),
    // kc = id / ROWS.  Row-contiguous staging: id -> k = id / (ROWS/4), cc = id % (ROWS/4).
    KcRow<A_RC ? HS_A_KC : AK> a_row;
    KcRow<HS_A_KC> b_row;
    RcCol a_col, b_col;
    if constexpr (!A_RC) kc_row_setup<AK>(a, m0 + tid % BM, a_row);
    else rc_col_setup<AK, true>(a, m0 + (tid % (BM / 4)) * 4, a_col);
    if constexpr (!B_RC) {
        const int n = n0 + tid % BN;
        b_row.valid = n < a.N;
        b_row.base = n * a.ldb;
        b_row.hb = b_row.wb = 0;
    } else {
        rc_col_setup<BKIND, false>(a, n0 + (tid % (BN / 4)) * 4, b_col);
    }

    u32x4 a_reg[A_NCH], b_reg[B_NCH];
    auto stage_load = [&](int k0) {
        KTile kta = {0, 0, 0}, ktb = {0, 0, 0};
        if constexpr (AK == HS_A_CONV) kta = ktile_rsc(k0, a.g.C, a.g.S);
        if constexpr (AK == HS_A_DGRAD) kta = ktile_rsc(k0, a.g.K, a.g.S);
        if constexpr (BKIND == HS_B_WDGRAD) ktb = ktile_rsc(k0, a.g.K, a.g.S);
#pragma unroll
        for (int i = 0; i < A_NCH; ++i) {
            const int id = tid + 256 * i;
            unsigned off;
            int nvalid = 4;
            if constexpr (!A_RC) {
                const int k = k0 + (id / BM) * 4;
                off = kc_chunk_off<AK, ESZ>(a, a_row, kta, k, kend);
                nvalid = kend - k;
            } else {
                const int k = k0 + id / (BM / 4);
                off = rc_chunk_off<AK, true, ESZ>(a, a_col, kta, k, kend);
                if constexpr (!VEC) nvalid = a.M - a_col.base;
            }
            a_reg[i] = load_chunk<T>(rsA, off, VEC, nvalid);
            if constexpr (!A_RC && VEC) if (nvalid < 4 && nvalid > 0) mask_tail<T>(a_reg[i], nvalid);
        }
#pragma unroll
        for (int i = 0; i < B_NCH; ++i) {
            const int id = tid + 256 * i;
            unsigned off;
            int nvalid = 4;
            if constexpr (!B_RC) {
                const int k = k0 + (id / BN) * 4;
                off = kc_chunk_off<HS_A_KC, ESZ>(a, b_row, ktb, k, kend);
                nvalid = kend - k;
            } else {
                const int k = k0 + id / (BN / 4);
                off = rc_chunk_off<BKIND, false, ESZ>(a, b_col, ktb, k, kend);
                if constexpr (!VEC) nvalid = a.N - (n0 + (tid % (BN / 4)) * 4);
            }
            b_reg[i] = load_chunk<T>(rsB, off, VEC, nvalid);
            if constexpr (!B_RC && VEC) if (nvalid < 4 && nvalid > 0) mask_tail<T>(b_reg[i], nvalid);
        }
    };
    auto stage_write = [&](int buf) {
        float* sa = (float*)(smem + buf * STAGE);
        float* sb = (float*)(smem + buf * STAGE + A_BYTES);
#pragma unroll
        for (int i = 0; i < A_NCH; ++i) {
            const int id = tid + 256 * i;
            if constexpr (!A_RC) {
                const int r = id % BM, kc = id / BM;
#pragma unroll
                for (int j = 0; j < 4; ++j) sa[(kc * 4 + j) * BM + r] = __uint_as_float(a_reg[i][j]);
            } else {
                *(u32x4*)(sa + (id / (BM / 4)) * BM + (id % (BM / 4)) * 4) = a_reg[i];
            }
        }
#pragma unroll
        for (int i = 0; i < B_NCH; ++i) {
            const int id = tid + 256 * i;
            if constexpr (!B_RC) {
                const int r = id % BN, kc = id / BN;
#pragma unroll
                for (int j = 0; j < 4; ++j) sb[(kc * 4 + j) * BN + r] = __uint_as_float(b_reg[i][j]);
            } else {
                *(u32x4*)(sb + (id / (BN / 4)) * BN + (id % (BN / 4)) * 4) = b_reg[i];
            }
        }
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int ntiles = (kend - kbeg + BK - 1) / BK;
    if (ntiles > 0) {
        stage_load(kbeg);
        stage_write(0);
    }
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < ntiles) stage_load(kbeg + (t + 1) * BK);
        const float* sa = (const float*)(smem + cur * STAGE);
        const float* sb = (const float*)(smem + cur * STAGE + A_BYTES);
#pragma unroll 4
        for (int kk = 0; kk < BK; kk += 2) {
            float av[FM], bv[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) av[i] = sa[(kk + hh) * BM + wm * WM + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < FN; ++j) bv[j] = sb[(kk + hh) * BN + wn * WN + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[j], av[i], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < ntiles) stage_write(cur ^ 1);
        __syncthreads();
    }

    // lane owns m = .. + l31; register e -> n = 8*(e>>2) + 4*hh + (e&3)
    const unsigned epi = epi_flags(a);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * WM + i * 32 + l31;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int n = n0 + wn * WN + j * 32 + 8 * qd + 4 * hh;
                float v[4] = {acc[i][j][4 * qd], acc[i][j][4 * qd + 1], acc[i][j][4 * qd + 2], acc[i][j][4 * qd + 3]};
                if (a.split_k > 1) {
                    if (m < a.M && n < a.N) {
                        float* w = a.splitk_ws + ((long long)z * a.M + m) * a.N + n;
                        if (n + 3 < a.N && (a.N & 3) == 0) *(f32x4*)w = f32x4{v[0], v[1], v[2], v[3]};
                        else
                            for (int e = 0; e < 4 && n + e < a.N; ++e) w[e] = v[e];
                    }
                } else {
                    epilogue4<T>(a, epi, d_boff, z, m, n, v);
                }
            }
        }
    }
}

// split-K second pass: sum the slabs in a fixed order (deterministic) and apply the epilogue.
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs a) {
    const long long nq = ((long long)a.N + 3) / 4;
    const long long total = (long long)a.M * nq;
    const unsigned epi = epi_flags(a);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i / nq), n = (int)(i - (long long)m * nq) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const int nvalid = min(4, a.N - n);
        for (int s = 0; s < a.split_k; ++s) {
            const float* w = a.splitk_ws + ((long long)s * a.M + m) * a.N + n;
            if (nvalid == 4 && (a.N & 3) == 0) {
                f32x4 x = *(const f32x4*)w;
                v[0] += x[0]; v[1] += x[1]; v[2] += x[2]; v[3] += x[3];
            } else {
                for (int e = 0; e < nvalid; ++e) v[e] += w[e];
            }
        }
        epilogue4<T>(a, epi, 0, 0, m, n, v);
    }
}

}  // namespace hs
