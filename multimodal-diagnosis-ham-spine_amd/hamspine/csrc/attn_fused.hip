// Fused attention (bf16; head dim 64 or 32; any number of queries in 128-row chunks; Lk <= 128 keys).  Forward, BERT shape
// first: one workgroup per (batch, head[, query chunk]) keeps
// Q, K, V in LDS (48 KiB) and the 128x128 score tile in registers:
//     S = scale * Q K^T (+ key mask)  ->  P = softmax(S)  ->  Pd = dropout(P)  ->  O = Pd V
// P and Pd are written once for the backward (same layout as the unfused path, which still provides it); the f32 score
// matrix never reaches HBM.  Replaces two batched GEMM launches and the softmax pass of transformers BertSelfAttention
// (reference encoder.py:131, mibf_net/bert.py:12).
//
// MFMA operand roles are swapped as in the GEMM core (MFMA-A := the "B" tile), so a lane ends up with 4 consecutive keys
// of one query row: row max / sum are 32 in-lane values plus two xor-shuffles, and P is stored as 8-byte vectors.  For
// O = Pd V the probabilities are used straight from the accumulator registers: within a 32-key block a lane holds keys
// {4g..4g+3} and {16+4g..16+4g+3}; MFMA's k index is only a summation index, so V is fetched in the same permuted key
// order (two ds_read_b64_tr_b16 at rows 4g and 16+4g) and no transpose through LDS is needed.
#include "gemm_core.h"

namespace hs {

// keep-scale of 4 consecutive probabilities starting at flat index idx (row * Lk + key): one hash when the group is aligned,
// element by element otherwise (Lk not a multiple of 4: the 49-token image grids) -- the same values either way
__device__ __forceinline__ void attn_drop4(unsigned long long seed, unsigned long long idx, unsigned thresh, float inv_keep, float* sc) {
    if ((idx & 3) == 0) {
        dropout_scale4(seed, idx, thresh, inv_keep, sc);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) sc[e] = dropout_scale(seed, idx + e, thresh, inv_keep);
    }
}

struct AttnFusedArgs {
    const char* q;
    const char* k;
    const char* v;
    char* o;
    char* P;
    char* Pd;                         // NULL without dropout
    const long long* key_mask;        // [B][Lk] or NULL
    long long q_bs, k_bs, v_bs, o_bs; // batch strides (elements)
    int q_ld, k_ld, v_ld, o_ld;       // token strides (elements)
    unsigned long long q_bytes, k_bytes, v_bytes;
    int H, Lq, Lk, ldP;
    float scale;
    unsigned thresh;
    float inv_keep;
    unsigned long long seed;
};

// HD: head dimension (64: BERT; 32: the fusion modules' nn.MultiheadAttention(256, 8), reference modules/fusion_blocks.py:18-40,
// 107-113).  blockIdx.y = query chunk: rows [128 y, 128 y + 128) of the Lq queries (the 784 x 128 score tile of the layer-2
// CrossAttentionBlock is 7 chunks); every chunk stages all Lk <= 128 keys.
// LK: key capacity of the workgroup (128, or 256: BERT at the MIBF loader's caption padding of 256 tokens -- 80 KiB of LDS, the
// 128 x 256 score tile is 128 accumulator registers per lane)
template <int HD, int LK = 128>
__global__ __launch_bounds__(256) void attn_fwd_fused_kernel(const AttnFusedArgs a) {
    constexpr int LMAX = 128, CPR = HD / 8, NP = LMAX * HD * 2 / 1024 / 4;      // DMA pieces per wave per tile (queries)
    constexpr int NPK = LK * HD * 2 / 1024 / 4;                                 // ... of the key / value tiles
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char lds_char;
    char* Qs = smem;                    // [128 queries][64]  k-contiguous, XOR swizzled (kc_off_bf16<64>)
    char* Ks = smem + LMAX * HD * 2;    // [128 keys][64]     k-contiguous
    char* Vs = smem + (LMAX + LK) * HD * 2;  // [LK keys][64]   key-major ("row-contiguous") for the P V product

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
    const int q0 = blockIdx.y * LMAX;                       // first query row of this chunk

    const long long qo = (long long)b * a.q_bs + (long long)h * HD, ko = (long long)b * a.k_bs + (long long)h * HD;
    const long long vo = (long long)b * a.v_bs + (long long)h * HD;
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(a.q + qo * 2, (unsigned)min(a.q_bytes - (unsigned long long)qo * 2, 0x7fffff00ull));
    const __amdgpu_buffer_rsrc_t rk = make_rsrc(a.k + ko * 2, (unsigned)min(a.k_bytes - (unsigned long long)ko * 2, 0x7fffff00ull));
    const __amdgpu_buffer_rsrc_t rv = make_rsrc(a.v + vo * 2, (unsigned)min(a.v_bytes - (unsigned long long)vo * 2, 0x7fffff00ull));

    // ---- stage Q, K, V: LMAX * HD * 2 / 1024 LDS-DMA pieces (1 KiB) per tile, NP per wave ------------------------
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int s = (wave * NP + i) * 64 + lane;          // 16-byte slot of the query tile
        const int r = s / CPR, pc = s % CPR;
        const int kl = (pc ^ kc_swz<CPR>(r)) * 8;
        const unsigned offq = q0 + r < a.Lq ? (unsigned)(((q0 + r) * a.q_ld + kl) * 2) : kOOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (__attribute__((address_space(3))) void*)((lds_char*)Qs + (wave * NP + i) * 1024), 16, offq, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NPK; ++i) {
        const int s = (wave * NPK + i) * 64 + lane;         // 16-byte slot of the key / value tiles
        {   // K-contiguous tile: row = s / CPR, physical chunk = s % CPR holds logical chunk pc ^ kc_swz(row)
            const int r = s / CPR, pc = s % CPR;
            const int kl = (pc ^ kc_swz<CPR>(r)) * 8;
            const unsigned offk = r < a.Lk ? (unsigned)((r * a.k_ld + kl) * 2) : kOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (__attribute__((address_space(3))) void*)((lds_char*)Ks + (wave * NPK + i) * 1024), 16, offk, 0, 0, 0);
        }
        {   // key-major V tile [key][HD]: key = s / CPR, physical chunk s % CPR holds logical chunk rc_logical_chunk
            const int key = s / (HD / 8), pcc = s % (HD / 8);
            const int col = rc_logical_chunk<HD>(key, pcc) * 8;
            const unsigned offv = key < a.Lk ? (unsigned)((key * a.v_ld + col) * 2) : kOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void*)((lds_char*)Vs + (wave * NPK + i) * 1024), 16, offv, 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- S = Q K^T: this wave owns query rows [32 wave, 32 wave + 32) ---------------------------------------------
    constexpr int FM = 2, FN = LK / 16;
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) {
        bf16x8 af[FM], bf[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = *(const bf16x8*)(Qs + kc_off_bf16<HD>(wave * 32 + i * 16 + l15, ks * 4 + g));
#pragma unroll
        for (int j = 0; j < FN; ++j) bf[j] = *(const bf16x8*)(Ks + kc_off_bf16<HD>(j * 16 + l15, ks * 4 + g));
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
    }

    // ---- softmax over the 128 keys of each row; lane holds keys 16 j + 4 g + e of row 32 wave + 16 i + l15 --------
    const long long* mk = a.key_mask ? a.key_mask + (long long)b * a.Lk : nullptr;
    bool dead[FN][4];
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = j * 16 + 4 * g + e;
            dead[j][e] = key >= a.Lk || (mk && key < a.Lk && mk[key] == 0);
        }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int qrow = q0 + wave * 32 + i * 16 + l15;
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = j * 16 + 4 * g + e;
                float s = acc[i][j][e] * a.scale;
                if (dead[j][e]) s = key < a.Lk ? -3.0e38f : -INFINITY;   // masked keys as the unfused softmax, padding never counts
                acc[i][j][e] = s;
                mx = fmaxf(mx, s);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p = __expf(acc[i][j][e] - mx);
                acc[i][j][e] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        const long long prow = (long long)bh * a.Lq + qrow;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int key0 = j * 16 + 4 * g;
            float p[4], pd[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) p[e] = (key0 + e < a.Lk) ? acc[i][j][e] * inv : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) pd[e] = p[e];
            if (a.thresh && key0 < a.Lk) {
                float sc[4];
                attn_drop4(a.seed, (unsigned long long)prow * a.Lk + key0, a.thresh, a.inv_keep, sc);
#pragma unroll
                for (int e = 0; e < 4; ++e) pd[e] *= sc[e];
            }
            if (qrow < a.Lq && key0 < a.ldP) {
                bf16_t* dst = (bf16_t*)a.P + prow * a.ldP + key0;
                *(bf16x4*)dst = bf16x4{(bf16_t)p[0], (bf16_t)p[1], (bf16_t)p[2], (bf16_t)p[3]};
                if (a.Pd) {
                    bf16_t* dd = (bf16_t*)a.Pd + prow * a.ldP + key0;
                    *(bf16x4*)dd = bf16x4{(bf16_t)pd[0], (bf16_t)pd[1], (bf16_t)pd[2], (bf16_t)pd[3]};
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = pd[e];     // what multiplies V
        }
    }

    // ---- O = Pd V: probabilities from registers, V in the matching permuted key order ----------------------------
    constexpr int FO = HD / 16;
    f32x4 oacc[FM][FO];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int jn = 0; jn < FO; ++jn) oacc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < LK / 32; ++kk) {
        bf16x8 pa[FM], vb[FO];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const f32x4 lo = acc[i][2 * kk], hi = acc[i][2 * kk + 1];
            pa[i] = bf16x8{(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3],
                           (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
        }
#pragma unroll
        for (int jn = 0; jn < FO; ++jn) {
            const int col = jn * 16 + 4 * (lane & 3);
            const int k_lo = kk * 32 + 4 * g + (l15 >> 2), k_hi = k_lo + 16;
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Vs + rc_off_bf16<HD>(k_lo, col)));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Vs + rc_off_bf16<HD>(k_hi, col)));
            vb[jn] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int jn = 0; jn < FO; ++jn)
                oacc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb[jn], pa[i], oacc[i][jn], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int qrow = q0 + wave * 32 + i * 16 + l15;
        if (qrow >= a.Lq) continue;
        bf16_t* orow = (bf16_t*)a.o + (long long)b * a.o_bs + (long long)qrow * a.o_ld + (long long)h * HD;
#pragma unroll
        for (int jn = 0; jn < FO; ++jn) {
            const f32x4 v = oacc[i][jn];
            *(bf16x4*)(orow + jn * 16 + 4 * g) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Fused backward for the same shape.  One workgroup per (batch, head); wave w owns query rows [32w, 32w+32) for the
// row-wise part and key rows [32w, 32w+32) for dV / dK:
//   A. dPd = dO V^T           (dO, V k-contiguous tiles)            -> registers, same layout as the forward scores
//      P from HBM (saved by the forward), mask regenerated:  Pd = P*M,  dS = P * (dPd*M - sum_keys(dPd*M*P))
//   B. dQ  = scale * dS K     (dS from registers, K key-major tile in the permuted key order, as P V in the forward)
//   C. dV  = Pd^T dO          (Pd through a 32 KiB LDS tile, read back transposed; dO key-major tile)
//   D. dK  = scale * dS^T Q   (dS through the same LDS tile; Q key-major tile)
// LDS: V_kc, dO_kc, dO_rc, K_rc, Q_rc (16 KiB each) = 80 KiB; the 128x128 bf16 transpose tile reuses V_kc + dO_kc after phase A.
// ------------------------------------------------------------------------------------------------------------------
struct AttnFusedBwdArgs {
    const char *q, *k, *v, *dO;
    char *dq, *dk, *dv;
    const char* P;
    long long q_bs, k_bs, v_bs, o_bs;
    int q_ld, k_ld, v_ld, o_ld;
    unsigned long long q_bytes, k_bytes, v_bytes, o_bytes;
    int H, Lq, Lk, ldP;
    float scale;
    unsigned thresh;
    float inv_keep;
    unsigned long long seed;
};

__global__ __launch_bounds__(256) void attn_bwd_fused_kernel(const AttnFusedBwdArgs a) {
    constexpr int HD = 64, LMAX = 128, CPR = HD / 8, TILE = LMAX * HD * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char lds_char;
    char* Vkc = smem;                // [key][hd]  k-contiguous swizzle
    char* Okc = smem + TILE;         // dO [q][hd] k-contiguous swizzle
    char* Orc = smem + 2 * TILE;     // dO [q][hd] key-major ("rc") swizzle, contraction over q
    char* Krc = smem + 3 * TILE;     // K  [key][hd] rc swizzle, contraction over keys
    char* Qrc = smem + 4 * TILE;     // Q  [q][hd]  rc swizzle, contraction over q
    char* PS = smem;                 // [q][key] bf16, rc swizzle with 128 columns: Pd, then dS.  ALIASES V_kc + dO_kc, which are
                                     // dead once phase A's fragments are in registers (barrier below): 80 KiB instead of 112,
                                     // so two workgroups share a CU and the 384 (batch, head) workgroups run in one round

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
    const long long qo = (long long)b * a.q_bs + (long long)h * HD, ko = (long long)b * a.k_bs + (long long)h * HD;
    const long long vo = (long long)b * a.v_bs + (long long)h * HD, oo = (long long)b * a.o_bs + (long long)h * HD;
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(a.q + qo * 2, (unsigned)min(a.q_bytes - (unsigned long long)qo * 2, 0x7fffff00ull));
    const __amdgpu_buffer_rsrc_t rk = make_rsrc(a.k + ko * 2, (unsigned)min(a.k_bytes - (unsigned long long)ko * 2, 0x7fffff00ull));
    const __amdgpu_buffer_rsrc_t rv = make_rsrc(a.v + vo * 2, (unsigned)min(a.v_bytes - (unsigned long long)vo * 2, 0x7fffff00ull));
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(a.dO + oo * 2, (unsigned)min(a.o_bytes - (unsigned long long)oo * 2, 0x7fffff00ull));

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = (wave * 4 + i) * 64 + lane;
        const int slot = (wave * 4 + i) * 1024;
        {   // k-contiguous tiles
            const int r = s / CPR, pc = s % CPR;
            const int kl = (pc ^ ((r / 2) % CPR)) * 8;
            const unsigned offv = r < a.Lk ? (unsigned)((r * a.v_ld + kl) * 2) : kOOB;
            const unsigned offo = r < a.Lq ? (unsigned)((r * a.o_ld + kl) * 2) : kOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void*)((lds_char*)Vkc + slot), 16, offv, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ro, (__attribute__((address_space(3))) void*)((lds_char*)Okc + slot), 16, offo, 0, 0, 0);
        }
        {   // key-major tiles
            const int row = s / (HD / 8), pcc = s % (HD / 8);
            const int col = rc_logical_chunk<HD>(row, pcc) * 8;
            const unsigned offo = row < a.Lq ? (unsigned)((row * a.o_ld + col) * 2) : kOOB;
            const unsigned offk = row < a.Lk ? (unsigned)((row * a.k_ld + col) * 2) : kOOB;
            const unsigned offq = row < a.Lq ? (unsigned)((row * a.q_ld + col) * 2) : kOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ro, (__attribute__((address_space(3))) void*)((lds_char*)Orc + slot), 16, offo, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (__attribute__((address_space(3))) void*)((lds_char*)Krc + slot), 16, offk, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (__attribute__((address_space(3))) void*)((lds_char*)Qrc + slot), 16, offq, 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- A. dPd = dO V^T for this wave's 32 query rows ------------------------------------------------------------
    constexpr int FM = 2, FN = 8, FO = HD / 16;
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[FM], bf[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = *(const bf16x8*)(Okc + kc_off_bf16<HD>(wave * 32 + i * 16 + l15, ks * 4 + g));
#pragma unroll
        for (int j = 0; j < FN; ++j) bf[j] = *(const bf16x8*)(Vkc + kc_off_bf16<HD>(j * 16 + l15, ks * 4 + g));
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();                                   // every wave has read its V / dO fragments: their tiles become the transpose tile
    // row-wise softmax backward; Pd goes to the transpose tile, dS stays in acc
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int qrow = wave * 32 + i * 16 + l15;
        const long long prow = (long long)bh * a.Lq + qrow;
        float pv[FN][4];
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int key0 = j * 16 + 4 * g;
            float sc[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) pv[j][e] = 0.f;
            if (qrow < a.Lq && key0 < a.ldP) {
                const u32x2 raw = *(const u32x2*)((const bf16_t*)a.P + prow * a.ldP + key0);
                pv[j][0] = __uint_as_float(raw[0] << 16); pv[j][1] = __uint_as_float(raw[0] & 0xffff0000u);
                pv[j][2] = __uint_as_float(raw[1] << 16); pv[j][3] = __uint_as_float(raw[1] & 0xffff0000u);
            }
            if (a.thresh && key0 < a.Lk) attn_drop4(a.seed, (unsigned long long)prow * a.Lk + key0, a.thresh, a.inv_keep, sc);
            float pd[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dp = acc[i][j][e] * sc[e];       // d loss / d P
                pd[e] = pv[j][e] * sc[e];                    // dropped-out probability (multiplies dO in dV)
                acc[i][j][e] = dp;
                dot = fmaf(dp, pv[j][e], dot);
            }
            *(bf16x4*)(PS + rc_off_bf16<LMAX>(qrow, key0)) = bf16x4{(bf16_t)pd[0], (bf16_t)pd[1], (bf16_t)pd[2], (bf16_t)pd[3]};
        }
        dot += __shfl_xor(dot, 16, 64);
        dot += __shfl_xor(dot, 32, 64);
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = pv[j][e] * (acc[i][j][e] - dot);     // dS
    }

    // ---- B. dQ = scale * dS K ------------------------------------------------------------------------------------
    {
        f32x4 qa[FM][FO];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int jn = 0; jn < FO; ++jn) qa[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < LMAX / 32; ++kk) {
            bf16x8 sa[FM], kb[FO];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const f32x4 lo = acc[i][2 * kk], hi = acc[i][2 * kk + 1];
                sa[i] = bf16x8{(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3],
                               (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
            }
#pragma unroll
            for (int jn = 0; jn < FO; ++jn) {
                const int col = jn * 16 + 4 * (lane & 3);
                const int k_lo = kk * 32 + 4 * g + (l15 >> 2), k_hi = k_lo + 16;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Krc + rc_off_bf16<HD>(k_lo, col)));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Krc + rc_off_bf16<HD>(k_hi, col)));
                kb[jn] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int jn = 0; jn < FO; ++jn) qa[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kb[jn], sa[i], qa[i][jn], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int qrow = wave * 32 + i * 16 + l15;
            if (qrow >= a.Lq) continue;
            bf16_t* row = (bf16_t*)a.dq + (long long)b * a.q_bs + (long long)qrow * a.q_ld + (long long)h * HD;
#pragma unroll
            for (int jn = 0; jn < FO; ++jn) {
                const f32x4 v = qa[i][jn];
                *(bf16x4*)(row + jn * 16 + 4 * g) = bf16x4{(bf16_t)(v[0] * a.scale), (bf16_t)(v[1] * a.scale),
                                                           (bf16_t)(v[2] * a.scale), (bf16_t)(v[3] * a.scale)};
            }
        }
    }

    // ---- C / D. transposed products over all 128 query rows: this wave's 32 keys x 64 head columns ---------------
    auto transposed_product = [&](const char* rhs_rc, char* out, long long out_bs, int out_ld, float alpha) {
        f32x4 ta[FM][FO];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int jn = 0; jn < FO; ++jn) ta[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < LMAX / 32; ++ks) {
            bf16x8 pa[FM], rb[FO];
            const int kb = ks * 32 + 8 * g + (l15 >> 2);                 // query rows 8g .. 8g+7 of this 32-row block
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int col = wave * 32 + i * 16 + 4 * (lane & 3);     // keys
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(PS + rc_off_bf16<LMAX>(kb, col)));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(PS + rc_off_bf16<LMAX>(kb + 4, col)));
                pa[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int jn = 0; jn < FO; ++jn) {
                const int col = jn * 16 + 4 * (lane & 3);                // head columns
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(rhs_rc + rc_off_bf16<HD>(kb, col)));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(rhs_rc + rc_off_bf16<HD>(kb + 4, col)));
                rb[jn] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int jn = 0; jn < FO; ++jn) ta[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rb[jn], pa[i], ta[i][jn], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int key = wave * 32 + i * 16 + l15;
            if (key >= a.Lk) continue;
            bf16_t* row = (bf16_t*)out + (long long)b * out_bs + (long long)key * out_ld + (long long)h * HD;
#pragma unroll
            for (int jn = 0; jn < FO; ++jn) {
                const f32x4 v = ta[i][jn];
                *(bf16x4*)(row + jn * 16 + 4 * g) = bf16x4{(bf16_t)(v[0] * alpha), (bf16_t)(v[1] * alpha), (bf16_t)(v[2] * alpha),
                                                           (bf16_t)(v[3] * alpha)};
            }
        }
    };
    __syncthreads();                                   // every wave's Pd rows are in the transpose tile
    transposed_product(Orc, a.dv, a.v_bs, a.v_ld, 1.f);            // dV = Pd^T dO
    __syncthreads();                                   // all reads of Pd done
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int qrow = wave * 32 + i * 16 + l15;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const f32x4 v = acc[i][j];
            *(bf16x4*)(PS + rc_off_bf16<LMAX>(qrow, j * 16 + 4 * g)) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        }
    }
    __syncthreads();
    transposed_product(Qrc, a.dk, a.k_bs, a.k_ld, a.scale);        // dK = scale * dS^T Q
}

// ------------------------------------------------------------------------------------------------------------------
// General backward: head dim HD (32 / 64), ANY number of query rows (chunks of 128 walked by ONE workgroup per (batch, head),
// dK / dV accumulated in registers across the chunks), Lk <= 128.  Same phases as above per chunk; V and K are staged
// once, dO / Q per chunk; the transpose tile has memory of its own here (V's tile is needed by every chunk).
// ------------------------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_fused_gen_kernel(const AttnFusedBwdArgs a) {
    constexpr int LMAX = 128, CPR = HD / 8, TILE = LMAX * HD * 2, NP = TILE / 1024 / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char lds_char;
    char* Vkc = smem;
    char* Okc = smem + TILE;
    char* Orc = smem + 2 * TILE;
    char* Krc = smem + 3 * TILE;
    char* Qrc = smem + 4 * TILE;
    char* PS = smem + 5 * TILE;      // [q][key] bf16, rc swizzle with 128 columns (32 KiB)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
    const long long qo = (long long)b * a.q_bs + (long long)h * HD, ko = (long long)b * a.k_bs + (long long)h * HD;
    const long long vo = (long long)b * a.v_bs + (long long)h * HD, oo = (long long)b * a.o_bs + (long long)h * HD;
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(a.q + qo * 2, (unsigned)min(a.q_bytes - (unsigned long long)qo * 2, 0x7fffff00ull));
    const __amdgpu_buffer_rsrc_t rk = make_rsrc(a.k + ko * 2, (unsigned)min(a.k_bytes - (unsigned long long)ko * 2, 0x7fffff00ull));
    const __amdgpu_buffer_rsrc_t rv = make_rsrc(a.v + vo * 2, (unsigned)min(a.v_bytes - (unsigned long long)vo * 2, 0x7fffff00ull));
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(a.dO + oo * 2, (unsigned)min(a.o_bytes - (unsigned long long)oo * 2, 0x7fffff00ull));

    // V (k-contiguous) and K (key-major): once
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int s = (wave * NP + i) * 64 + lane;
        const int slot = (wave * NP + i) * 1024;
        {
            const int r = s / CPR, pc = s % CPR;
            const int kl = (pc ^ kc_swz<CPR>(r)) * 8;
            const unsigned offv = r < a.Lk ? (unsigned)((r * a.v_ld + kl) * 2) : kOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void*)((lds_char*)Vkc + slot), 16, offv, 0, 0, 0);
        }
        {
            const int row = s / (HD / 8), pcc = s % (HD / 8);
            const int col = rc_logical_chunk<HD>(row, pcc) * 8;
            const unsigned offk = row < a.Lk ? (unsigned)((row * a.k_ld + col) * 2) : kOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (__attribute__((address_space(3))) void*)((lds_char*)Krc + slot), 16, offk, 0, 0, 0);
        }
    }
    constexpr int FM = 2, FN = 8, FO = HD / 16;
    f32x4 tv[FM][FO], tk[FM][FO];              // dV, dK of this wave's 32 keys x HD head columns, summed over the chunks
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int jn = 0; jn < FO; ++jn) {
            tv[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
            tk[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

    const int nchunks = (a.Lq + LMAX - 1) / LMAX;
    for (int c = 0; c < nchunks; ++c) {
        const int q0 = c * LMAX;
        // dO (both layouts) and Q (key-major) of this chunk
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int s = (wave * NP + i) * 64 + lane;
            const int slot = (wave * NP + i) * 1024;
            {
                const int r = s / CPR, pc = s % CPR;
                const int kl = (pc ^ kc_swz<CPR>(r)) * 8;
                const unsigned offo = q0 + r < a.Lq ? (unsigned)(((q0 + r) * a.o_ld + kl) * 2) : kOOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ro, (__attribute__((address_space(3))) void*)((lds_char*)Okc + slot), 16, offo, 0, 0, 0);
            }
            {
                const int row = s / (HD / 8), pcc = s % (HD / 8);
                const int col = rc_logical_chunk<HD>(row, pcc) * 8;
                const unsigned offo = q0 + row < a.Lq ? (unsigned)(((q0 + row) * a.o_ld + col) * 2) : kOOB;
                const unsigned offq = q0 + row < a.Lq ? (unsigned)(((q0 + row) * a.q_ld + col) * 2) : kOOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ro, (__attribute__((address_space(3))) void*)((lds_char*)Orc + slot), 16, offo, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (__attribute__((address_space(3))) void*)((lds_char*)Qrc + slot), 16, offq, 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        // ---- A. dPd = dO V^T for this wave's 32 query rows of the chunk
        f32x4 acc[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks) {
            bf16x8 af[FM], bf[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = *(const bf16x8*)(Okc + kc_off_bf16<HD>(wave * 32 + i * 16 + l15, ks * 4 + g));
#pragma unroll
            for (int j = 0; j < FN; ++j) bf[j] = *(const bf16x8*)(Vkc + kc_off_bf16<HD>(j * 16 + l15, ks * 4 + g));
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
        // row-wise softmax backward; Pd goes to the transpose tile, dS stays in acc
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int lrow = wave * 32 + i * 16 + l15, qrow = q0 + lrow;
            const long long prow = (long long)bh * a.Lq + qrow;
            float pv[FN][4];
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int key0 = j * 16 + 4 * g;
                float sc[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) pv[j][e] = 0.f;
                if (qrow < a.Lq && key0 < a.ldP) {
                    const u32x2 raw = *(const u32x2*)((const bf16_t*)a.P + prow * a.ldP + key0);
                    pv[j][0] = __uint_as_float(raw[0] << 16); pv[j][1] = __uint_as_float(raw[0] & 0xffff0000u);
                    pv[j][2] = __uint_as_float(raw[1] << 16); pv[j][3] = __uint_as_float(raw[1] & 0xffff0000u);
                }
                if (a.thresh && key0 < a.Lk) attn_drop4(a.seed, (unsigned long long)prow * a.Lk + key0, a.thresh, a.inv_keep, sc);
                float pd[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dp = acc[i][j][e] * sc[e];
                    pd[e] = pv[j][e] * sc[e];
                    acc[i][j][e] = dp;
                    dot = fmaf(dp, pv[j][e], dot);
                }
                *(bf16x4*)(PS + rc_off_bf16<LMAX>(lrow, key0)) = bf16x4{(bf16_t)pd[0], (bf16_t)pd[1], (bf16_t)pd[2], (bf16_t)pd[3]};
            }
            dot += __shfl_xor(dot, 16, 64);
            dot += __shfl_xor(dot, 32, 64);
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = pv[j][e] * (acc[i][j][e] - dot);     // dS
        }
        // ---- B. dQ = scale * dS K for the chunk's rows
        {
            f32x4 qa[FM][FO];
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int jn = 0; jn < FO; ++jn) qa[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < LMAX / 32; ++kk) {
                bf16x8 sa[FM], kb[FO];
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const f32x4 lo = acc[i][2 * kk], hi = acc[i][2 * kk + 1];
                    sa[i] = bf16x8{(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3],
                                   (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
                }
#pragma unroll
                for (int jn = 0; jn < FO; ++jn) {
                    const int col = jn * 16 + 4 * (lane & 3);
                    const int k_lo = kk * 32 + 4 * g + (l15 >> 2), k_hi = k_lo + 16;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Krc + rc_off_bf16<HD>(k_lo, col)));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Krc + rc_off_bf16<HD>(k_hi, col)));
                    kb[jn] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int jn = 0; jn < FO; ++jn) qa[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kb[jn], sa[i], qa[i][jn], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int qrow = q0 + wave * 32 + i * 16 + l15;
                if (qrow >= a.Lq) continue;
                bf16_t* row = (bf16_t*)a.dq + (long long)b * a.q_bs + (long long)qrow * a.q_ld + (long long)h * HD;
#pragma unroll
                for (int jn = 0; jn < FO; ++jn) {
                    const f32x4 v = qa[i][jn];
                    *(bf16x4*)(row + jn * 16 + 4 * g) = bf16x4{(bf16_t)(v[0] * a.scale), (bf16_t)(v[1] * a.scale),
                                                               (bf16_t)(v[2] * a.scale), (bf16_t)(v[3] * a.scale)};
                }
            }
        }
        // ---- C / D. transposed products over the chunk's 128 query rows, accumulated into this wave's keys
        auto transposed_accumulate = [&](const char* rhs_rc, f32x4 (&ta)[FM][FO]) {
#pragma unroll
            for (int ks = 0; ks < LMAX / 32; ++ks) {
                bf16x8 pa[FM], rb[FO];
                const int kb = ks * 32 + 8 * g + (l15 >> 2);
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int col = wave * 32 + i * 16 + 4 * (lane & 3);
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(PS + rc_off_bf16<LMAX>(kb, col)));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(PS + rc_off_bf16<LMAX>(kb + 4, col)));
                    pa[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
                for (int jn = 0; jn < FO; ++jn) {
                    const int col = jn * 16 + 4 * (lane & 3);
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(rhs_rc + rc_off_bf16<HD>(kb, col)));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(rhs_rc + rc_off_bf16<HD>(kb + 4, col)));
                    rb[jn] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int jn = 0; jn < FO; ++jn) ta[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rb[jn], pa[i], ta[i][jn], 0, 0, 0);
            }
        };
        __syncthreads();                                   // every wave's Pd rows are in the transpose tile
        transposed_accumulate(Orc, tv);                    // dV += Pd^T dO
        __syncthreads();                                   // all reads of Pd done
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int lrow = wave * 32 + i * 16 + l15;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const f32x4 v = acc[i][j];
                *(bf16x4*)(PS + rc_off_bf16<LMAX>(lrow, j * 16 + 4 * g)) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            }
        }
        __syncthreads();
        transposed_accumulate(Qrc, tk);                    // dK += dS^T Q   (scaled at the end)
        __syncthreads();                                   // the chunk's tiles may be overwritten
    }
    auto store_keys = [&](const f32x4 (&ta)[FM][FO], char* out, long long out_bs, int out_ld, float alpha) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int key = wave * 32 + i * 16 + l15;
            if (key >= a.Lk) continue;
            bf16_t* row = (bf16_t*)out + (long long)b * out_bs + (long long)key * out_ld + (long long)h * HD;
#pragma unroll
            for (int jn = 0; jn < FO; ++jn) {
                const f32x4 v = ta[i][jn];
                *(bf16x4*)(row + jn * 16 + 4 * g) = bf16x4{(bf16_t)(v[0] * alpha), (bf16_t)(v[1] * alpha), (bf16_t)(v[2] * alpha),
                                                           (bf16_t)(v[3] * alpha)};
            }
        }
    };
    store_keys(tv, a.dv, a.v_bs, a.v_ld, 1.f);
    store_keys(tk, a.dk, a.k_bs, a.k_ld, a.scale);
}

// host side: eligibility + launch.  Returns 1 when the fused kernel ran, 0 when the shape is not covered, < 0 on error.
int attention_fwd_fused(const hs_attn_desc& d, const void* q, const void* k, const void* v, void* o, void* P, void* Pd,
                        int ldP, hipStream_t s) {
    static int enabled = -1;
    if (enabled < 0) {
        const char* e = getenv("HAMSPINE_FUSED_ATTENTION");
        enabled = (e && e[0] == '0') ? 0 : 1;
    }
    // head dim 64 (BERT) or 32 (the fusion modules' 8-head attention over 256 features), any number of queries (128-row
    // chunks on grid.y), at most 128 keys -- 256 at head dim 64, forward only: the backward of such a call takes the unfused
    // kernels on the probabilities this kernel stored
    if (!enabled || d.dtype != HS_BF16 || (d.hd != 64 && d.hd != 32) || d.Lq < 1 || d.Lq > 128 * 65535 || d.Lk < 1 || d.Lk > (d.hd == 64 ? 256 : 128) || ldP % 4 != 0 || ldP < d.Lk) return 0;
    const long long strides[] = {d.q_bs, d.k_bs, d.v_bs, d.o_bs, d.q_ld, d.k_ld, d.v_ld, d.o_ld};
    for (long long x : strides)
        if (x % 8 != 0) return 0;
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) || (((uintptr_t)o | (uintptr_t)P | (uintptr_t)Pd) & 7)) return 0;
    auto span = [&](long long bs, int L, int ld) { return ((long long)(d.B - 1) * bs + (long long)(L - 1) * ld + (long long)d.H * d.hd) * 2; };
    if (span(d.q_bs, d.Lq, d.q_ld) >= 0x7fffff00ll || span(d.k_bs, d.Lk, d.k_ld) >= 0x7fffff00ll ||
        span(d.v_bs, d.Lk, d.v_ld) >= 0x7fffff00ll)
        return 0;
    AttnFusedArgs a;
    memset(&a, 0, sizeof(a));
    a.q = (const char*)q; a.k = (const char*)k; a.v = (const char*)v; a.o = (char*)o;
    a.P = (char*)P; a.Pd = d.dropout_p > 0.f ? (char*)Pd : nullptr;
    a.key_mask = (const long long*)d.key_mask;
    a.q_bs = d.q_bs; a.k_bs = d.k_bs; a.v_bs = d.v_bs; a.o_bs = d.o_bs;
    a.q_ld = d.q_ld; a.k_ld = d.k_ld; a.v_ld = d.v_ld; a.o_ld = d.o_ld;
    a.q_bytes = (unsigned long long)span(d.q_bs, d.Lq, d.q_ld);
    a.k_bytes = (unsigned long long)span(d.k_bs, d.Lk, d.k_ld);
    a.v_bytes = (unsigned long long)span(d.v_bs, d.Lk, d.v_ld);
    a.H = d.H; a.Lq = d.Lq; a.Lk = d.Lk; a.ldP = ldP;
    a.scale = d.scale;
    if (d.dropout_p > 0.f) {
        a.thresh = dropout_thresh(d.dropout_p);
        a.inv_keep = 1.f / (1.f - d.dropout_p);
        a.seed = d.seed;
    }
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)attn_fwd_fused_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 128 * 64 * 2) != hipSuccess) return -1;
        if (hipFuncSetAttribute((const void*)attn_fwd_fused_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 128 * 32 * 2) != hipSuccess) return -1;
        if (hipFuncSetAttribute((const void*)attn_fwd_fused_kernel<64, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (128 + 2 * 256) * 64 * 2) != hipSuccess) return -1;
        attr_set = true;
    }
    const dim3 grid(d.B * d.H, (d.Lq + 127) / 128);
    if (d.hd == 64 && d.Lk > 128) hipLaunchKernelGGL((attn_fwd_fused_kernel<64, 256>), grid, dim3(256), (128 + 2 * 256) * 64 * 2, s, a);
    else if (d.hd == 64) hipLaunchKernelGGL(attn_fwd_fused_kernel<64>, grid, dim3(256), 3 * 128 * 64 * 2, s, a);
    else hipLaunchKernelGGL(attn_fwd_fused_kernel<32>, grid, dim3(256), 3 * 128 * 32 * 2, s, a);
    if (hipGetLastError() != hipSuccess) return -1;
    return 1;
}

int attention_bwd_fused(const hs_attn_desc& d, const void* q, const void* k, const void* v, const void* dO, void* dq, void* dk,
                        void* dv, const void* P, int ldP, hipStream_t s) {
    static int enabled = -1;
    if (enabled < 0) {
        const char* e = getenv("HAMSPINE_FUSED_ATTENTION");
        enabled = (e && e[0] == '0') ? 0 : 1;
    }
    if (!enabled || d.dtype != HS_BF16 || (d.hd != 64 && d.hd != 32) || d.Lq < 1 || d.Lk < 1 || d.Lk > 128 || ldP % 4 != 0 || ldP < d.Lk) return 0;
    const long long strides[] = {d.q_bs, d.k_bs, d.v_bs, d.o_bs, d.q_ld, d.k_ld, d.v_ld, d.o_ld};
    for (long long x : strides)
        if (x % 8 != 0) return 0;
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dO) & 15) ||
        (((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv | (uintptr_t)P) & 7))
        return 0;
    auto span = [&](long long bs, int L, int ld) { return ((long long)(d.B - 1) * bs + (long long)(L - 1) * ld + (long long)d.H * d.hd) * 2; };
    const long long sq = span(d.q_bs, d.Lq, d.q_ld), sk = span(d.k_bs, d.Lk, d.k_ld), sv = span(d.v_bs, d.Lk, d.v_ld),
                    so = span(d.o_bs, d.Lq, d.o_ld);
    if (sq >= 0x7fffff00ll || sk >= 0x7fffff00ll || sv >= 0x7fffff00ll || so >= 0x7fffff00ll) return 0;
    AttnFusedBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.q = (const char*)q; a.k = (const char*)k; a.v = (const char*)v; a.dO = (const char*)dO;
    a.dq = (char*)dq; a.dk = (char*)dk; a.dv = (char*)dv; a.P = (const char*)P;
    a.q_bs = d.q_bs; a.k_bs = d.k_bs; a.v_bs = d.v_bs; a.o_bs = d.o_bs;
    a.q_ld = d.q_ld; a.k_ld = d.k_ld; a.v_ld = d.v_ld; a.o_ld = d.o_ld;
    a.q_bytes = (unsigned long long)sq; a.k_bytes = (unsigned long long)sk; a.v_bytes = (unsigned long long)sv;
    a.o_bytes = (unsigned long long)so;
    a.H = d.H; a.Lq = d.Lq; a.Lk = d.Lk; a.ldP = ldP;
    a.scale = d.scale;
    if (d.dropout_p > 0.f) {
        a.thresh = dropout_thresh(d.dropout_p);
        a.inv_keep = 1.f / (1.f - d.dropout_p);
        a.seed = d.seed;
    }
    static bool attr_set = false;
    const int lds = 5 * 128 * 64 * 2, lds_g64 = 5 * 128 * 64 * 2 + 128 * 128 * 2, lds_g32 = 5 * 128 * 32 * 2 + 128 * 128 * 2;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)attn_bwd_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -1;
        if (hipFuncSetAttribute((const void*)attn_bwd_fused_gen_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_g64) != hipSuccess) return -1;
        if (hipFuncSetAttribute((const void*)attn_bwd_fused_gen_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_g32) != hipSuccess) return -1;
        attr_set = true;
    }
    // the BERT shape keeps its own kernel (transpose tile aliased onto dead operand tiles: two workgroups per CU)
    if (d.hd == 64 && d.Lq <= 128) hipLaunchKernelGGL(attn_bwd_fused_kernel, dim3(d.B * d.H), dim3(256), lds, s, a);
    else if (d.hd == 64) hipLaunchKernelGGL(attn_bwd_fused_gen_kernel<64>, dim3(d.B * d.H), dim3(256), lds_g64, s, a);
    else hipLaunchKernelGGL(attn_bwd_fused_gen_kernel<32>, dim3(d.B * d.H), dim3(256), lds_g32, s, a);
    if (hipGetLastError() != hipSuccess) return -1;
    return 1;
}

}  // namespace hs
