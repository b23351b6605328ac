// host side of the GEMM / implicit-GEMM core: argument checking, tile selection, split-K.
#include <mutex>
#include <unordered_set>
#include <vector>
#include <stdarg.h>
#include <math.h>
#include <map>
#include <tuple>
#include "gemm_launch.h"
#include "gemm_p8.h"

namespace hs {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* last_error() { return g_err; }

bool lds_attr_needed(const void* fn) {
    static std::mutex mu;
    static std::unordered_set<const void*> seen;
    std::lock_guard<std::mutex> lk(mu);
    return seen.insert(fn).second;
}

// ------------------------------------------------------------------------------------------------
// optional per-launch timing of the GEMM kernels (bench.py's roofline leg): HIP events are recorded on
// the launch stream around every main kernel; classes: 0 bf16 GEMM, 1 bf16 conv, 2 f32 GEMM, 3 f32 conv
// ------------------------------------------------------------------------------------------------
struct ProfRec {
    hipEvent_t a, b;
    double flops;
    int cls;
    int M, N, K, combo, cfg, split, batch, conv_r, conv_stride;
    double bytes = 0;     // grouped launches: algorithmic bytes of all their problems (single launches: derived from the shape)
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_pool;

static bool prof_begin(hipStream_t s, double flops, int cls, ProfRec& r) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_on) return false;
    if (!g_prof_pool.empty()) {
        r.a = g_prof_pool.back().first;
        r.b = g_prof_pool.back().second;
        g_prof_pool.pop_back();
    } else if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) {
        return false;
    }
    r.flops = flops;
    r.cls = cls;
    (void)hipEventRecord(r.a, s);
    return true;
}
static void prof_end(hipStream_t s, ProfRec& r) {
    (void)hipEventRecord(r.b, s);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(r);
}

__global__ void null_kernel() {}

// Arrival counters of the in-launch split-K reduction (bf16 kernel): one zeroed pool per (device, stream).  Launches on
// one stream are ordered and the reducing workgroup re-zeroes its counter, so every launch of a stream can use the same
// counters; launches on different streams (the two towers) run concurrently and therefore get different pools.
static constexpr int kTicketPool = 16384;
static unsigned* ticket_pool(hipStream_t s) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, unsigned*> pools;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    auto it = pools.find({dev, s});
    if (it != pools.end()) return it->second;
    unsigned* p = nullptr;
    if (hipMalloc(&p, kTicketPool * sizeof(unsigned)) != hipSuccess) return nullptr;
    if (hipMemsetAsync(p, 0, kTicketPool * sizeof(unsigned), s) != hipSuccess) return nullptr;   // ordered before the first use
    pools[{dev, s}] = p;
    return p;
}

static int g_dbg_cfg = -1, g_dbg_ablate = 0;   // measurement overrides (hs_gemm_debug)
static unsigned long long* g_dbg_stamps = nullptr;   // hs_gemm_debug_stamps: per-workgroup clock stamps of the next launches

static int combo_of(int ak, int bk) {
    if (ak == HS_A_KC && bk == HS_B_KC) return 0;
    if (ak == HS_A_KC && bk == HS_B_RC) return 1;
    if (ak == HS_A_RC && bk == HS_B_RC) return 2;
    if (ak == HS_A_CONV && bk == HS_B_KC) return 3;
    if (ak == HS_A_DGRAD && bk == HS_B_WDGRAD) return 4;
    if (ak == HS_A_RC && bk == HS_B_CONV) return 5;
    return -1;
}

static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

static bool batch_is_one(const hs_gemm_params* p) { return p->batch <= 1; }
// the phase-pipelined body (gemm_p8.h) for the shapes it wins on; HAMSPINE_P8=0 keeps every GEMM on the generic body
static bool p8_enabled() {
    static const bool on = [] { const char* e = getenv("HAMSPINE_P8"); return !(e && e[0] == '0'); }();
    return on;
}
static int p8_cfg(const hs_gemm_params* p) {
    // 256 x 256 tiles where they give the chip enough workgroups (BERT-base at 4096 tokens: N = 2304 / 3072 -> 144 / 192)
    if (p->M % 256 == 0 && p->N % 256 == 0 && (long long)(p->M / 256) * (p->N / 256) >= 128) return CFG_P8_256;
    return -1;
}
// tile choice for everything but the stem (see the comment at the call site)
static int auto_cfg(const hs_gemm_params* p, bool vec) {
    const long long z = (long long)(p->batch > 0 ? p->batch : 1) * (p->split_k > 1 ? p->split_k : 1);
    const long long t12864 = (long long)ceil_div(p->M, 128) * ceil_div(p->N, 64) * z;
    const long long t128 = (long long)ceil_div(p->M, 128) * ceil_div(p->N, 128) * z;
    if (p->dtype == HS_BF16) {
        // plain GEMMs of BERT-base size, cold caches (tools/gemm_sweep.py on MI355X, profiles/round2_gemm_sweep.txt):
        //   wide outputs with a short K (QKV / FFN1 forward, FFN2 data gradient: N >= 2048, K <= 1024): 128x128, BK = 32
        //     (36.8 vs 46.3 us on nn 4096x3072x768; 37.1 vs 40.2 us on nt);
        //   K-contiguous operands with a long K (FFN2 forward 4096x768x3072): 128x64 once it gives >= 256 tiles (40.4 vs 48.8 us).
        // (The QKV weight gradient 2304x768x4096 would gain 18 % as 128x64 tiles split two ways, but a split launch cannot
        // also produce the bias gradient -- rowsum_a -- and the three column-sum passes that replace it cost more.)
        const bool plain = (p->a_kind == HS_A_KC || p->a_kind == HS_A_RC) && (p->b_kind == HS_B_KC || p->b_kind == HS_B_RC);
        if (plain && batch_is_one(p)) {
            if (p->a_kind == HS_A_KC && p->split_k <= 1 && p->N >= 2048 && p->K <= 1024 && p->M >= 2048 && p->K % 32 == 0) return CFG_128x128x32;
            if (p->a_kind == HS_A_KC && p->b_kind == HS_B_KC && p->split_k <= 1 && p->K >= 2048 && t12864 >= 256) return CFG_128x64;
        }
        return t12864 >= 1536 ? CFG_128x64 : CFG_64x64;
    }
    return (vec && t128 >= 200) ? CFG_128x128 : CFG_64x64;
}
// The 3x3 halo-patch kernel (conv3.hip) takes stride-1 pad-1 3x3 forward convolutions in standard NHWC with a plain bf16
// epilogue (optionally the BatchNorm statistics riders): tile = R whole image rows (R * W <= 112 pixels) x 64 channels.
// HAMSPINE_CONV3=0 keeps them on the generic implicit GEMM.
static bool conv3_enabled() {
    static const bool on = [] { const char* e = getenv("HAMSPINE_CONV3"); return !(e && e[0] == '0'); }();
    return on;
}
static bool conv3_plan(const hs_gemm_params* p, int* rows, int* tpi, int* fm) {
    if (!conv3_enabled() || g_dbg_cfg >= 0) return false;
    const hs_conv_geom& g = p->g;
    if (p->dtype != HS_BF16 || p->a_kind != HS_A_CONV || p->b_kind != HS_B_KC || p->split_k > 1 || p->batch > 1) return false;
    if (g.R != 3 || g.S != 3 || g.stride != 1 || g.pad != 1 || g.P != g.H || g.Q != g.W || g.C % 64 != 0 || p->N % 64 != 0) return false;
    if (g.row_pitch != g.W * g.C || g.img_pitch != g.H * g.W * g.C || g.qstep != g.C || g.no_bounds) return false;
    if (p->bias || p->colscale || p->act != HS_ACT_NONE || p->residual || p->D_preact || p->mul_mode != HS_MUL_NONE || p->dropout_p > 0.f ||
        p->accumulate || p->seg_rows > 0 || p->alpha != 1.f || p->out_dtype != HS_BF16 || p->rowsum_a || p->bnb_partials)
        return false;
    if (p->ldd % 8 != 0 || (((uintptr_t)p->D) & 15) != 0 || p->ldb != 9 * g.C || p->K != 9 * g.C || p->N != g.K) return false;
    if (g.W > 112 || g.W < 1) return false;
    const int rmax = std::min(g.H, 112 / g.W);
    const int t = ceil_div(g.H, rmax), r = ceil_div(g.H, t);
    const int tp = r * g.W;
    if (tp < 32 || (r + 2) * (g.W + 2) > 256) return false;
    *rows = r;
    *tpi = t;
    *fm = tp > 64 ? 7 : 4;
    return true;
}
// rows of a colstats buffer = row tiles the bf16 kernel will use for p
int gemm_stat_rows(const hs_gemm_params* p) {
    if (!p || p->dtype != HS_BF16 || p->split_k > 1 || p->batch > 1) return 0;
    {
        int r, t, fm;
        if (conv3_plan(p, &r, &t, &fm)) return p->g.N * t;
    }
    int cfg = (p->a_kind == HS_A_CONV && p->b_kind == HS_B_KC && p->g.C % 64 != 0) ? CFG_STEM : auto_cfg(p, true);
    if (g_dbg_cfg >= 0 && g_dbg_cfg <= CFG_64x64 && cfg != CFG_STEM) cfg = g_dbg_cfg;
    if (g_dbg_cfg >= CFG_256x128 && g_dbg_cfg <= CFG_256x128x32 && !(p->a_kind == HS_A_CONV || p->a_kind == HS_A_DGRAD || p->b_kind == HS_B_CONV)) cfg = g_dbg_cfg;
    return ceil_div(p->M, cfg == CFG_64x64 ? 64 : (cfg == CFG_256x128 || cfg == CFG_256x128x32) ? 256 : 128);
}

// tiles / layouts with a BatchNorm-finishing kernel variant (gemm_bf16_bnf_kernel): forward convolutions and 1x1 convolutions
static bool bn_finish_variant(int cfg, int combo) {
    return (combo == 0 && (cfg == CFG_64x64 || cfg == CFG_128x64)) || (combo == 3 && (cfg == CFG_64x64 || cfg == CFG_128x64 || cfg == CFG_STEM || cfg == CFG_C3));
}

// everything gemm_impl decides before the launch
struct Prepared {
    GemmArgs a;
    int cfg, combo, split, batch;
    bool bf16, conv, vec;
    dim3 grid;
    double flops;
};
static int gemm_prepare(const hs_gemm_params* p, hipStream_t stream, Prepared& q, int force_cfg = -1) {
    HS_REQUIRE(p != nullptr, "hs_gemm: null params");
    HS_REQUIRE(p->dtype == HS_F32 || p->dtype == HS_BF16, "hs_gemm: bad dtype %d", p->dtype);
    HS_REQUIRE(p->M > 0 && p->N > 0 && p->K >= 0, "hs_gemm: bad dims %d %d %d", p->M, p->N, p->K);
    HS_REQUIRE(p->A && p->B && p->D, "hs_gemm: null operand");
    const int combo = combo_of(p->a_kind, p->b_kind);
    HS_REQUIRE(combo >= 0, "hs_gemm: unsupported operand kinds (%d,%d)", p->a_kind, p->b_kind);
    const bool bf16 = p->dtype == HS_BF16;
    const int esz = bf16 ? 2 : 4, epc = bf16 ? 8 : 4;
    const int batch = p->batch > 0 ? p->batch : 1;
    const int split = p->split_k > 1 ? p->split_k : 1;
    HS_REQUIRE(!(batch > 1 && split > 1), "hs_gemm: batch and split_k are exclusive");
    HS_REQUIRE((long long)p->a_elems * esz < 0x7fffff00ll && (long long)p->b_elems * esz < 0x7fffff00ll,
               "hs_gemm: operand larger than 2 GiB");
    HS_REQUIRE(p->out_dtype == HS_F32 || p->out_dtype == p->dtype, "hs_gemm: out_dtype must be f32 or the operand type");
    HS_REQUIRE(!p->accumulate || p->out_dtype == HS_F32, "hs_gemm: accumulate needs an f32 output");

    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = (const char*)p->A;
    a.B = (const char*)p->B;
    a.a_bytes = (unsigned long long)p->a_elems * esz;
    a.b_bytes = (unsigned long long)p->b_elems * esz;
    a.lda = p->lda;
    a.ldb = p->ldb;
    a.M = p->M;
    a.N = p->N;
    a.K = p->K;
    a.g = p->g;
    a.batch_inner = p->batch_inner > 0 ? p->batch_inner : 1;
    a.a_bs0 = p->a_bs0; a.a_bs1 = p->a_bs1;
    a.b_bs0 = p->b_bs0; a.b_bs1 = p->b_bs1;
    a.d_bs0 = p->d_bs0; a.d_bs1 = p->d_bs1;
    a.D = (char*)p->D;
    a.ldd = p->ldd;
    a.out_f32 = p->out_dtype == HS_F32;
    a.alpha = p->alpha;
    a.bias = p->bias;
    a.act = p->act;
    a.D_preact = (char*)p->D_preact;
    a.residual = (const char*)p->residual;
    a.ldr = p->ldr;
    if (p->dropout_p > 0.f) {
        HS_REQUIRE(p->dropout_p < 1.f, "hs_gemm: dropout_p must be < 1");
        a.drop_thresh = dropout_thresh(p->dropout_p);
        a.drop_inv_keep = 1.f / (1.f - p->dropout_p);
        a.drop_seed = p->dropout_seed;
    }
    a.mul_mode = p->mul_mode;
    a.mul_src = (const char*)p->mul_src;
    a.ldm = p->ldm;
    a.accumulate = p->accumulate;
    a.colstats = p->colstats;
    if (p->bnb_partials) {
        HS_REQUIRE(bf16 && p->out_dtype == HS_BF16 && batch == 1 && (combo == 1 || combo == 4) && p->bnb_x && ((p->bnb_scale && p->bnb_shift) || p->bnb_y) &&
                       p->bnb_mean && p->bnb_invstd && !p->bias && !p->colscale && p->act == HS_ACT_NONE && (!p->residual || (p->bnb_y && p->ldr == p->ldd)) && !p->D_preact &&
                       p->mul_mode == HS_MUL_NONE && p->dropout_p == 0.f && !p->accumulate && p->seg_rows <= 0 && p->alpha == 1.f &&
                       p->N % 4 == 0 && p->ldd % 4 == 0 && ((((uintptr_t)p->bnb_x) & 7) == 0),
                   "hs_gemm: bnb_partials needs a bf16 data-gradient GEMM with a plain epilogue (no bias / activation / residual / multiplier)");
        a.bnb_x = (const char*)p->bnb_x;
        a.bnb_scale = p->bnb_scale; a.bnb_shift = p->bnb_shift; a.bnb_mean = p->bnb_mean; a.bnb_invstd = p->bnb_invstd;
        a.bnb_partials = p->bnb_partials;
        a.bnb_y = (const char*)p->bnb_y;
        HS_REQUIRE(!p->bnb_y || ((((uintptr_t)p->bnb_y) & 7) == 0 && (!p->residual || (((uintptr_t)p->residual) & 7) == 0)),
                   "hs_gemm: bnb_y / residual must be 8-byte aligned");
    }
    a.colscale = p->colscale;
    a.res_pre_act = p->residual_before_act;
    a.stamps = g_dbg_stamps;
    a.epi_generic = (g_dbg_ablate & 32) ? 1 : 0;
    a.dbg = g_dbg_ablate;
    a.rowsum[0] = p->rowsum_a;
    a.rowsum[1] = p->rowsum_seg[0];
    a.rowsum[2] = p->rowsum_seg[1];
    if (p->rowsum_a) {
        HS_REQUIRE(p->dtype == HS_BF16 && (p->a_kind == HS_A_RC || (p->a_kind == HS_A_KC && p->b_kind == HS_B_KC)) &&
                       p->split_k <= 1 && batch == 1,
                   "hs_gemm: rowsum_a needs bf16, a row-contiguous A (or K-contiguous A and B), no split-K and no batch");
        HS_REQUIRE(p->seg_rows <= 0 || (p->rowsum_seg[0] && (p->M <= 2 * p->seg_rows || p->rowsum_seg[1])),
                   "hs_gemm: rowsum_a with seg_rows needs rowsum_seg");
    }
    a.seg_rows = p->seg_rows;
    a.D_seg[0] = (char*)p->D_seg[0];
    a.D_seg[1] = (char*)p->D_seg[1];
    if (p->seg_rows > 0) {
        HS_REQUIRE(batch == 1 && !p->D_preact && !p->residual && p->D_seg[0] && (p->M <= 2 * p->seg_rows || p->D_seg[1]) &&
                       p->M <= 3 * p->seg_rows,
                   "hs_gemm: bad segmented output");
    }
    HS_REQUIRE(a.mul_mode == HS_MUL_NONE || a.mul_src, "hs_gemm: mul_mode without mul_src");

    // conv geometry
    const bool conv = combo >= 3;
    const int bk = bf16 ? 64 : 32;
    int cfg = -1;
    if (conv) {
        const hs_conv_geom& g = p->g;
        HS_REQUIRE(g.stride == 1 || g.stride == 2, "hs_gemm: conv stride must be 1 or 2");
        if (combo == 3) {
            HS_REQUIRE(p->K == g.R * g.S * g.C, "conv fwd: K != R*S*C");
            HS_REQUIRE(p->M == g.N * g.P * g.Q, "conv fwd: M != N*P*Q");
            a.div_mhw = make_fastdiv(g.P * g.Q);
            a.div_mw = make_fastdiv(g.Q);
            if (bf16 && g.C % 64 != 0) {
                HS_REQUIRE(g.C % 32 == 0 && p->N <= 64, "conv fwd (bf16): C must be a multiple of 64 (or 32 with N<=64)");
                cfg = CFG_STEM;
            } else {
                HS_REQUIRE(g.C % bk == 0, "conv fwd: C %% %d != 0", bk);
            }
        } else if (combo == 4) {
            HS_REQUIRE(p->K == g.R * g.S * g.K, "conv dgrad: K != R*S*Kout");
            HS_REQUIRE(p->M == g.N * g.H * g.W, "conv dgrad: M != N*H*W");
            HS_REQUIRE(p->N == g.C, "conv dgrad: N != C");
            HS_REQUIRE(g.K % bk == 0, "conv dgrad: Kout %% %d != 0", bk);
            a.div_mhw = make_fastdiv(g.H * g.W);
            a.div_mw = make_fastdiv(g.W);
            // stride 2: parity-major row order lets the kernel skip the 3/4 of the filter taps that cannot reach a given
            // input pixel (HAMSPINE_PARITY_DGRAD=0 keeps the natural order)
            static int parity_on = -1;
            if (parity_on < 0) {
                const char* e = getenv("HAMSPINE_PARITY_DGRAD");
                parity_on = (e && e[0] == '0') ? 0 : 1;
            }
            if (parity_on && bf16 && g.stride == 2 && g.H % 2 == 0 && g.W % 2 == 0 && g.R * g.S <= 15 && split == 1 && batch == 1 &&
                !p->seg_rows && !p->colstats && !p->bnb_partials) {
                a.parity = 1;
                a.quarter = g.N * (g.H / 2) * (g.W / 2);
                a.div_qhw = make_fastdiv((g.H / 2) * (g.W / 2));
                a.div_qw = make_fastdiv(g.W / 2);
            }
        } else {
            HS_REQUIRE(p->K == g.N * g.P * g.Q, "conv wgrad: K != N*P*Q");
            HS_REQUIRE(p->N == g.R * g.S * g.C, "conv wgrad: N != R*S*C");
            HS_REQUIRE(p->M == g.K, "conv wgrad: M != Kout");
            HS_REQUIRE(g.C % epc == 0, "conv wgrad: C %% %d != 0", epc);
            a.div_mhw = make_fastdiv(g.P * g.Q);
            a.div_mw = make_fastdiv(g.Q);
            a.div_sc = make_fastdiv(g.S * g.C);
            a.div_c = make_fastdiv(g.C);
        }
    }

    // vector-path eligibility (16-byte chunks): strides and bases chunk aligned.
    bool vec = aligned16(p->A) && aligned16(p->B);
    if (p->a_kind == HS_A_KC || p->a_kind == HS_A_RC) vec = vec && (p->lda % epc == 0);
    if (p->b_kind == HS_B_KC || p->b_kind == HS_B_RC) vec = vec && (p->ldb % epc == 0);
    if (batch > 1) vec = vec && (p->a_bs0 % epc == 0) && (p->a_bs1 % epc == 0) && (p->b_bs0 % epc == 0) && (p->b_bs1 % epc == 0);
    if (conv) HS_REQUIRE(vec, "hs_gemm: conv operands must be 16-byte aligned");
    if (bf16) {
        HS_REQUIRE(vec, "hs_gemm(bf16): lda/ldb/batch strides must be multiples of 8 and bases 16-byte aligned");
        // LDS-DMA moves whole 16-byte chunks: a K tail inside a chunk must be zero padding in memory
        const int k8 = (p->K + 7) / 8 * 8;
        if (p->a_kind == HS_A_KC) HS_REQUIRE(p->K % 8 == 0 || p->lda >= k8, "hs_gemm(bf16): A rows need K%%8==0 or zero padding to %d", k8);
        if (p->b_kind == HS_B_KC) HS_REQUIRE(p->K % 8 == 0 || p->ldb >= k8, "hs_gemm(bf16): B rows need K%%8==0 or zero padding to %d", k8);
    }
    // 4-wide epilogue accesses
    {
        const int oesz = a.out_f32 ? 4 : esz;
        bool vs = (p->N % 4 == 0) && (p->ldd % 4 == 0) && ((((uintptr_t)p->D) % (4 * oesz)) == 0);
        if (batch > 1) vs = vs && (p->d_bs0 % 4 == 0) && (p->d_bs1 % 4 == 0);
        if (p->D_preact) vs = vs && ((((uintptr_t)p->D_preact) % (4 * oesz)) == 0);
        if (p->residual) vs = vs && (p->ldr % 4 == 0) && ((((uintptr_t)p->residual) % (4 * oesz)) == 0);
        if (p->mul_src) vs = vs && (p->ldm % 4 == 0) && ((((uintptr_t)p->mul_src) % (4 * esz)) == 0);
        a.vec_store = vs;
        bool v16 = bf16 && !a.out_f32 && (p->N % 8 == 0) && (p->ldd % 8 == 0) && ((((uintptr_t)p->D) & 15) == 0);
        if (batch > 1) v16 = v16 && (p->d_bs0 % 8 == 0) && (p->d_bs1 % 8 == 0);
        if (p->D_preact) v16 = v16 && ((((uintptr_t)p->D_preact) & 15) == 0);
        a.vec16 = v16 ? 1 : 0;
    }

    // tile selection.  Measured on MI355X (tools/gemm_bench.py, profiles/): the GEMMs of this workload are
    // 1-20 GFLOP, i.e. a few microseconds of MFMA time, so filling the chip evenly beats per-tile arithmetic
    // intensity: 128x128 never wins, 128x64 wins once it yields >= ~1500 tiles, otherwise 64x64.
    if (cfg < 0) cfg = auto_cfg(p, vec);
    if (p->rowsum_a && p->a_kind == HS_A_KC && cfg != CFG_128x64 && cfg != CFG_64x64) cfg = CFG_128x64;   // instantiated tiles of the RS variant
    if (force_cfg >= 0) cfg = force_cfg;                       // grouped launches: the group's tile
    if (p->bnb_partials && cfg != CFG_64x64 && cfg != CFG_128x64) cfg = CFG_128x64;   // instantiated tiles of the BatchNorm-sum variant
    if (g_dbg_cfg >= 0 && g_dbg_cfg <= CFG_64x64 && cfg != CFG_STEM) cfg = g_dbg_cfg;
    if (g_dbg_cfg >= CFG_256x128 && g_dbg_cfg <= CFG_256x128x32 && bf16 && !conv) cfg = g_dbg_cfg;
    if (cfg == CFG_128x128x32 && split > 1) cfg = CFG_128x128;      // the three-workgroups-per-CU kernel has no split-K code
    // The phase-pipelined body (gemm_p8.h): plain nt GEMMs whose tiles are all full, no split / batch / statistics riders, and
    // an epilogue feature set it has code for.  Which of its tiles (if any) a shape takes: p8_cfg below.
    {
        const bool p8_ok = bf16 && combo == 0 && batch == 1 && split == 1 && !a.colstats && !p->bnb_partials && !p->bn_finish &&
                           p->K % 64 == 0 && p->K >= 128 && p->lda % 8 == 0 && p->ldb % 8 == 0 && force_cfg < 0 &&
                           (long long)p->M * p->lda * 2 < 0x7fffff00ll && (long long)p->N * p->ldb * 2 < 0x7fffff00ll && p8_epilogue_supported(a);
        auto fits = [&](int c) {
            const int bm = c == CFG_P8_128 ? 128 : 256, bn = c == CFG_P8_256 ? 256 : 128;
            return p->M % bm == 0 && p->N % bn == 0 && !(c == CFG_P8_128 && p->rowsum_a);
        };
        int want = -1;
        if (g_dbg_cfg >= CFG_P8_256 && g_dbg_cfg <= CFG_P8_128) want = g_dbg_cfg;
        else if (g_dbg_cfg < 0 && p8_enabled()) want = p8_cfg(p);
        if (p8_ok && want >= 0 && fits(want)) cfg = want;
    }
    if (force_cfg < 0 && combo == 3 && conv3_plan(p, &a.c3_rows, &a.c3_tpi, &a.c3_fm)) cfg = CFG_C3;
    // (A deeper operand ring -- 5 / 8 slots for the long-K split weight gradients of the convolutions -- was built and
    // measured: 1.93 vs 1.92 ms per step on the 1x1 weight gradients, 0.70 vs 0.52 ms on the 3x3 ones, where the larger LDS
    // footprint costs a resident workgroup.  The kernel keeps the ring depth as a template parameter; 3 is what runs.)
    const int ring = cfg == CFG_128x128x32 ? HS_W3_RING : 3;
    int BM = 64, BN = 64;
    if (cfg == CFG_P8_256) { BM = 256; BN = 256; }
    else if (cfg == CFG_P8_256x128) { BM = 256; BN = 128; }
    else if (cfg == CFG_P8_128) { BM = 128; BN = 128; }
    else if (cfg == CFG_128x128 || cfg == CFG_128x128x32) { BM = 128; BN = 128; }
    else if (cfg == CFG_256x128 || cfg == CFG_256x128x32) { BM = 256; BN = 128; }
    else if (cfg == CFG_128x64 || cfg == CFG_STEM) { BM = 128; BN = 64; }
    a.tiles_m = ceil_div(p->M, BM);
    a.tiles_n = ceil_div(p->N, BN);
    if (cfg == CFG_C3) {
        a.tiles_m = p->g.N * a.c3_tpi;
        a.tiles_n = p->N / 64;
        a.div_mw = make_fastdiv(p->g.W);
        a.div_qw = make_fastdiv(p->g.W + 2);
    }
    const int kb_cfg = (cfg == CFG_STEM || cfg == CFG_128x128x32 || cfg == CFG_256x128x32) ? 32 : bk;
    {
        // K tiles one workgroup walks -> LDS ring slots it needs (a single-tile 1x1 convolution allocates one slot, so
        // 5-6 workgroups instead of 2 share a CU and hide each other's load -> MFMA -> store latency chain)
        const int ktiles_all = ceil_div(p->K, kb_cfg);
        const int kt = split > 1 ? ceil_div(ktiles_all, split) : ktiles_all;
        a.lds_stages = std::max(1, std::min(ring, kt));
        // L2 grouping: an XCD runs S workgroups at a time (32 CUs x resident workgroups); the panels they touch are
        // fewest when group_m * BM == (S / group_m) * BN
        const bool p8 = cfg >= CFG_P8_256 && cfg <= CFG_P8_128;
        const int lds = p8 ? 2 * (BM + BN) * 64 * 2 : bf16 ? a.lds_stages * (BM + BN) * kb_cfg * 2 : 2 * (BM + BN) * 32 * 4;
        const int per_cu = std::max(1, std::min(4, (160 * 1024) / lds));
        const double S = 32.0 * per_cu;
        int gm = (int)(sqrt(S * BN / BM) + 0.5);
        gm = std::max(1, std::min(gm, a.tiles_m));
        if (g_dbg_ablate & 16) gm = 1;   // measurement: plain n-fastest order
        a.group_m = gm;
    }

    a.split_k = split;
    if (split > 1) {
        HS_REQUIRE(p->splitk_ws != nullptr, "hs_gemm: split_k needs a workspace");
        const int ktiles = ceil_div(p->K, kb_cfg);
        a.k_per_split = ceil_div(ktiles, split) * kb_cfg;
        a.splitk_ws = p->splitk_ws;
        // bf16: the last-arriving slice of a tile reduces it inside the launch; f32 (parity mode) keeps the second pass
        const long long ngroups = (split + kSplitGroup - 1) / kSplitGroup;
        if (bf16 && (long long)a.tiles_m * a.tiles_n * (1 + ngroups) <= kTicketPool && splitk_slabs(split) * p->M * p->N * 4 < 0x7fffff00ll) {
            a.tickets = ticket_pool(stream);
            HS_REQUIRE(a.tickets != nullptr, "hs_gemm: cannot allocate the split-K arrival counters");
        }
    }
    if (a.colstats) HS_REQUIRE(bf16 && split == 1 && batch == 1, "hs_gemm: colstats needs bf16 operands, no split-K, no batch");
    if (p->bnb_finish) {       // the launch also finishes the BatchNorm-backward sums (bnb_handoff; hs_gemm_bnb_finish_rows said it can)
        const hs_bn_bwd_params* f = p->bnb_finish;
        HS_REQUIRE(a.bnb_partials && split == 1 && (cfg == CFG_64x64 || cfg == CFG_128x64) && a.persist == 0,
                   "hs_gemm: bnb_finish needs bnb_partials and a launch hs_gemm_bnb_finish_rows accepts");
        HS_REQUIRE(f->C == p->N && f->M > 0 && f->dgamma && f->dbeta, "hs_gemm: bnb_finish needs the BatchNorm's C, M, dgamma, dbeta");
        HS_REQUIRE((long long)a.tiles_n * (1 + stat_groups(a.tiles_m)) <= kTicketPool, "hs_gemm: bnb_finish: too many column tiles");
        a.bnb_gamma = f->gamma; a.bnb_dgamma = f->dgamma; a.bnb_dbeta = f->dbeta;
        a.bnb_invm = 1.f / (float)f->M; a.bnb_train = f->training;
        a.bnb_tickets = ticket_pool(stream);
        HS_REQUIRE(a.bnb_tickets != nullptr, "hs_gemm: cannot allocate the arrival counters");
    }
    if (p->bn_finish) {        // the launch also finishes the BatchNorm statistics (BNF kernels; hs_gemm_bn_finish_rows said it can)
        const hs_bn_params* f = p->bn_finish;
        HS_REQUIRE(a.colstats && bn_finish_variant(cfg, combo), "hs_gemm: bn_finish needs colstats and a tile / layout hs_gemm_bn_finish_rows accepts");
        HS_REQUIRE(f->training && f->C == p->N && f->M == p->M && f->save_mean && f->save_invstd && f->scale && f->shift,
                   "hs_gemm: bn_finish needs a training-mode hs_bn_params of the result's shape with save_mean / save_invstd / scale / shift");
        HS_REQUIRE((long long)a.tiles_n * (1 + stat_groups(a.tiles_m)) <= kTicketPool, "hs_gemm: bn_finish: too many column tiles");
        a.bnf_gamma = f->gamma; a.bnf_beta = f->beta;
        a.bnf_rmean = f->running_var ? f->running_mean : nullptr; a.bnf_rvar = f->running_var;
        a.bnf_mean = f->save_mean; a.bnf_invstd = f->save_invstd; a.bnf_scale = f->scale; a.bnf_shift = f->shift;
        a.bnf_eps = f->eps; a.bnf_momentum = f->momentum;
        a.bnf_tickets = ticket_pool(stream);
        HS_REQUIRE(a.bnf_tickets != nullptr, "hs_gemm: cannot allocate the arrival counters");
    }
    {   // measurement aid (HAMSPINE_EPI_HIST=1): which epilogue feature sets the launches use, printed at exit
        static const bool hist = [] { const char* e = getenv("HAMSPINE_EPI_HIST"); return e && e[0] == '1'; }();
        if (hist) {
            static std::mutex mu;
            static std::map<std::pair<unsigned, int>, long long> cnt;
            static const bool reg = [] {
                atexit([] {
                    for (auto& kv : cnt) fprintf(stderr, "[epi] flags 0x%05x bf16 %d : %lld launches\n", kv.first.first, kv.first.second, kv.second);
                });
                return true;
            }();
            (void)reg;
            unsigned f = 0;
            if (p->bias) f |= EPI_BIAS;
            if (p->colscale) f |= EPI_COLSCALE;
            if (p->mul_mode == HS_MUL_GELU_GRAD) f |= EPI_MUL_GELU;
            else if (p->mul_mode != HS_MUL_NONE) f |= EPI_MUL_RELU;
            if (p->seg_rows > 0) f |= EPI_SEG;
            if (p->D_preact) f |= EPI_PREACT;
            if (p->residual) f |= p->residual_before_act ? EPI_RES_PRE : EPI_RES_POST;
            if (p->act == HS_ACT_RELU) f |= EPI_RELU;
            else if (p->act == HS_ACT_GELU) f |= EPI_GELU;
            if (p->dropout_p > 0.f) f |= EPI_DROP;
            if (p->out_dtype == HS_F32) f |= EPI_OUT_F32;
            if (p->accumulate) f |= EPI_ACCUM;
            if (a.vec_store) f |= EPI_VEC;
            if (a.parity) f |= EPI_PARITY;
            if (split > 1) f |= 0x80000;
            std::lock_guard<std::mutex> lk(mu);
            cnt[{f, p->dtype == HS_BF16 ? 1 : 0}]++;
        }
    }
    dim3 grid(a.tiles_m * a.tiles_n, 1, split > 1 ? split : batch);
    {   // persistent launch: more tiles than workgroups the chip holds at once -> one workgroup per resident slot walks
        // several tiles and prefetches the next tile under the current epilogue.  Built, parity-tested and measured: no gain
        // (4096x3072x768 as 128x64 tiles 40.1 vs 39.8 us; the K = 64 ResNet layers 36.0 vs 31.2 us, because the variant's
        // registers leave room for 2 instead of 3-5 workgroups per CU and those layers live on bytes in flight): the
        // hardware already overlaps one workgroup's set-up with its neighbours' epilogues.  OFF unless HAMSPINE_PERSISTENT=1
        // (or hs_gemm_debug ablation bit 64).
        static const bool on = [] { const char* e = getenv("HAMSPINE_PERSISTENT"); return e && e[0] == '1'; }();
        static const int cus = [] { hipDeviceProp_t pr; int d = 0; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess) ? pr.multiProcessorCount : 256; }();
        if ((on || (g_dbg_ablate & 64)) && bf16 && split == 1 && batch == 1 && !a.stamps && !p->rowsum_a && !p->bnb_partials && !p->bn_finish && (cfg == CFG_64x64 || cfg == CFG_128x64) &&
            (combo == 0 || combo == 1 || combo == 3 || combo == 4)) {
            const int lds = a.lds_stages * (BM + BN) * 64 * 2;
            const int per_cu = std::max(1, std::min(cfg == CFG_64x64 ? 3 : 2, (160 * 1024) / lds));   // 168 / 256 VGPRs, LDS
            const int slots = cus * per_cu;
            if ((int)grid.x > slots) {
                a.persist = slots;
                grid.x = slots;
            }
        }
    }
    double flops = 2.0 * p->M * p->N * (double)p->K * batch;
    if (a.parity) {   // strided dgrad: only the (pixel parity, filter tap) pairs that reach an output pixel are real work
        int kept = 0;
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < p->g.R; ++r)
                for (int q = 0; q < p->g.S; ++q)
                    kept += ((((c >> 1) + p->g.pad - r) & 1) == 0 && (((c & 1) + p->g.pad - q) & 1) == 0) ? 1 : 0;
        flops *= (double)kept / (4.0 * p->g.R * p->g.S);
    }
    q.a = a;
    q.cfg = cfg; q.combo = combo; q.split = split; q.batch = batch;
    q.bf16 = bf16; q.conv = conv; q.vec = vec;
    q.grid = grid;
    q.flops = flops;
    return HS_OK;
}

// tile rows of the launch gemm_impl makes for p (a prepare pass without side effects beyond the ticket pool)
int gemm_tile_rows(const hs_gemm_params* p) {
    if (!p) return 0;
    hs_gemm_params c = *p;
    unsigned char dummy[16];
    if (c.split_k > 1 && !c.splitk_ws) c.splitk_ws = (float*)dummy;        // prepare only checks that it is there
    Prepared q;
    if (gemm_prepare(&c, nullptr, q) != HS_OK) return 0;
    return q.a.tiles_m;
}

// rows of the colstats buffer when the launch for p can also finish the BatchNorm statistics (tile rows + group rows), else 0
int gemm_bn_finish_rows(const hs_gemm_params* p) {
    if (!p || !p->colstats) return 0;
    hs_gemm_params c = *p;
    c.bn_finish = nullptr;
    Prepared q;
    if (gemm_prepare(&c, nullptr, q) != HS_OK) return 0;
    if (!q.bf16 || q.split != 1 || q.batch != 1 || q.a.persist > 0 || !bn_finish_variant(q.cfg, q.combo)) return 0;
    if ((long long)q.a.tiles_n * (1 + stat_groups(q.a.tiles_m)) > kTicketPool) return 0;
    return q.a.tiles_m + stat_groups(q.a.tiles_m);
}

// rows of the bnb_partials buffer when the launch for p can also finish the BatchNorm-backward sums (tile rows + group rows), else 0
int gemm_bnb_finish_rows(const hs_gemm_params* p) {
    if (!p || !p->bnb_partials) return 0;
    hs_gemm_params c = *p;
    c.bnb_finish = nullptr;
    unsigned char dummy[16];
    if (c.split_k > 1 && !c.splitk_ws) c.splitk_ws = (float*)dummy;
    Prepared q;
    if (gemm_prepare(&c, nullptr, q) != HS_OK) return 0;
    if (!q.bf16 || q.split != 1 || q.batch != 1 || q.a.persist > 0 || !(q.cfg == CFG_64x64 || q.cfg == CFG_128x64)) return 0;
    if ((long long)q.a.tiles_n * (1 + stat_groups(q.a.tiles_m)) > kTicketPool) return 0;
    return q.a.tiles_m + stat_groups(q.a.tiles_m);
}

int gemm_impl(const hs_gemm_params* p, hipStream_t stream) {
    Prepared q;
    HS_PROPAGATE(gemm_prepare(p, stream, q));
    const GemmArgs& a = q.a;
    const int cfg = q.cfg, combo = q.combo, split = q.split, batch = q.batch;
    const bool bf16 = q.bf16, conv = q.conv, vec = q.vec;
    const dim3 grid = q.grid;
    int st;
    ProfRec rec;
    const bool timed = prof_begin(stream, q.flops, (bf16 ? 0 : 2) + (conv ? 1 : 0), rec);
    if (bf16 && cfg == CFG_C3) st = launch_bf16_conv3(a, a.c3_fm, grid, stream);
    else if (bf16 && cfg >= CFG_P8_256 && cfg <= CFG_P8_128) st = launch_bf16_p8(cfg, a, grid, stream);
    else if (bf16) st = conv ? launch_bf16_conv(cfg, combo, a, grid, stream) : launch_bf16_plain(cfg, combo, a, grid, stream);
    else st = conv ? launch_f32_conv(cfg, combo, a, grid, stream) : launch_f32_plain(cfg, combo, vec, a, grid, stream);
    if (timed) {
        rec.M = p->M; rec.N = p->N; rec.K = p->K; rec.combo = combo; rec.cfg = cfg; rec.split = split; rec.batch = batch;
        rec.conv_r = conv ? p->g.R : 0; rec.conv_stride = conv ? p->g.stride : 0;
        prof_end(stream, rec);
    }
    HS_PROPAGATE(st);
    if (split > 1 && !a.tickets) {
        const long long work = (long long)p->M * ((p->N + 3) / 4);
        const int blocks = (int)std::min<long long>((work + 255) / 256, 2048);
        if (bf16) hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(blocks), dim3(256), 0, stream, a);
        HS_LAUNCH_CHECK();
    }
    return HS_OK;
}


// ------------------------------------------------------------------------------------------------
// grouped launches (gemm_bf16_grouped_kernel): problems are collected with gemm_group_add and launched as one grid by
// gemm_group_flush.  A group object belongs to one (device, stream, slot) and keeps its device-side table between steps:
// a training step adds the same problems with the same pointers every time, so the table is uploaded once.
// ------------------------------------------------------------------------------------------------
struct GemmGroup {
    int combo = -1;       // 2 / 5: 64x64 tiles (gemm_bf16_grouped_kernel); 0: K-contiguous operands, 256x128 tiles, row sums (_big_kernel)
    int cfg = -1;         // combo 0: CFG_256x128 (generic body) or CFG_P8_256 (phase-pipelined body, 256x256 tiles)
    bool rs = false;      // some problem of a CFG_P8_256 group wants row sums
    std::vector<GemmArgs> items;
    std::vector<int> first;
    int total = 0, ticket_off = 0;
    double flops = 0, bytes = 0;
    char* dev = nullptr;
    size_t dev_cap = 0;
    std::vector<char> shadow;      // what the device table holds
};
static constexpr int kGroupMax = 64;
GemmGroup* gemm_group_open(hipStream_t s, long long slot) {
    static std::mutex mu;
    static std::map<std::tuple<int, hipStream_t, long long>, GemmGroup*> groups;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    GemmGroup*& g = groups[std::make_tuple(dev, s, slot)];
    if (!g) g = new GemmGroup();
    g->combo = -1;
    g->cfg = -1;
    g->rs = false;
    g->items.clear();
    g->first.clear();
    g->total = 0;
    g->ticket_off = 0;
    g->flops = g->bytes = 0;
    return g;
}
int gemm_group_flush(GemmGroup* g, hipStream_t s) {
    if (!g || g->items.empty()) return HS_OK;
    const int n = (int)g->items.size();
    constexpr size_t kHead = 512;                                   // first_wg[0..64] and padding
    const size_t bytes = kHead + (size_t)n * sizeof(GemmArgs);
    if (g->combo == 0 && n <= 4 && g->cfg != CFG_P8_256) {          // small groups of the big-tile kind travel in the kernel arguments
        ProfRec rec;
        const bool timed = prof_begin(s, g->flops, 0, rec);
        const int st = launch_bf16_grouped_big4(g->items.data(), g->first.data(), n, g->total, s);
        if (timed) {
            rec.M = n; rec.N = 0; rec.K = 0; rec.combo = 8; rec.cfg = CFG_256x128; rec.split = 0; rec.batch = 1;
            rec.conv_r = 0; rec.conv_stride = 0;
            rec.bytes = g->bytes;
            prof_end(s, rec);
        }
        g->combo = -1;
        g->cfg = -1;
        g->rs = false;
        g->items.clear();
        g->first.clear();
        g->total = 0;
        g->ticket_off = 0;
        g->flops = g->bytes = 0;
        return st;
    }
    std::vector<char> host(bytes, 0);
    int* fw = (int*)host.data();
    for (int i = 0; i < n; ++i) fw[i] = g->first[i];
    fw[n] = g->total;
    memcpy(host.data() + kHead, g->items.data(), (size_t)n * sizeof(GemmArgs));
    if (host != g->shadow) {
        // rare (first step, or an arena moved): the previous launch of this group may still be reading the table
        static const bool dbg = [] { const char* e = getenv("HAMSPINE_GROUP_DEBUG"); return e && e[0] == '1'; }();
        if (dbg) {
            size_t diff = 0;
            for (size_t i = 0; i < std::min(host.size(), g->shadow.size()); ++i) diff += host[i] != g->shadow[i];
            fprintf(stderr, "[group] table upload: %d problems, %zu bytes (%zu before), %zu bytes differ\n", n, bytes, g->shadow.size(), diff);
        }
        HS_CHECK_HIP(hipStreamSynchronize(s));
        if (bytes > g->dev_cap) {
            if (g->dev) (void)hipFree(g->dev);
            g->dev = nullptr;
            g->dev_cap = 0;
            HS_CHECK_HIP(hipMalloc(&g->dev, bytes * 2));
            g->dev_cap = bytes * 2;
        }
        HS_CHECK_HIP(hipMemcpy(g->dev, host.data(), bytes, hipMemcpyHostToDevice));
        g->shadow.swap(host);
    }
    ProfRec rec;
    const bool timed = prof_begin(s, g->flops, g->combo == 5 ? 1 : 0, rec);
    const int st = g->combo == 5   ? launch_bf16_grouped_conv(5, (const GemmArgs*)(g->dev + kHead), (const int*)g->dev, n, g->total, s)
                   : (g->combo == 0 && g->cfg == CFG_P8_256) ? launch_bf16_p8_grouped((const GemmArgs*)(g->dev + kHead), (const int*)g->dev, n, g->total, g->rs, s)
                   : g->combo == 0 ? launch_bf16_grouped_big(0, (const GemmArgs*)(g->dev + kHead), (const int*)g->dev, n, g->total, s)
                                   : launch_bf16_grouped_plain(2, (const GemmArgs*)(g->dev + kHead), (const int*)g->dev, n, g->total, s);
    if (timed) {
        rec.M = n; rec.N = 0; rec.K = 0; rec.combo = g->combo == 5 ? 7 : g->combo == 0 ? 8 : 6;
        rec.cfg = g->combo == 0 ? (g->cfg == CFG_P8_256 ? CFG_P8_256 : CFG_256x128) : CFG_64x64; rec.split = 0; rec.batch = 1;
        rec.conv_r = 0; rec.conv_stride = 0;
        rec.bytes = g->bytes;
        prof_end(s, rec);
    }
    g->combo = -1;
    g->cfg = -1;
    g->rs = false;
    g->items.clear();
    g->first.clear();
    g->total = 0;
    g->ticket_off = 0;
    g->flops = g->bytes = 0;
    return st;
}
// A caller that defers work a grouped GEMM depends on (the BERT tower's dY^T transposes) registers a hook: it runs before any
// problem that gemm_group_add launches on its own instead of queueing.
static thread_local int (*g_group_immediate_hook)(void*) = nullptr;
static thread_local void* g_group_immediate_ctx = nullptr;
void gemm_group_set_immediate_hook(int (*fn)(void*), void* ctx) {
    g_group_immediate_hook = fn;
    g_group_immediate_ctx = ctx;
}
static int launch_ungrouped(const hs_gemm_params* p, hipStream_t s) {
    if (g_group_immediate_hook) HS_PROPAGATE(g_group_immediate_hook(g_group_immediate_ctx));
    return gemm_impl(p, s);
}
// adds p to the group, or launches it on its own when it cannot be grouped (f32, another layout or tile, no in-launch reduce)
int gemm_group_add(GemmGroup* g, const hs_gemm_params* p, hipStream_t s) {
    if (!g) return launch_ungrouped(p, s);
    Prepared q;
    const bool big = p->a_kind == HS_A_KC && p->b_kind == HS_B_KC;       // K-contiguous weight gradients: 256x128 tiles
    // ... or, when every tile is a full 256 x 256 one and the group is a deep-K one (K >= 2048: the K walk has to amortise the
    // one-tile-per-CU body's ramp), the phase-pipelined body.  HAMSPINE_P8_WGRAD=0 keeps the generic 256x128 tiles.
    static const bool p8w = [] { const char* e = getenv("HAMSPINE_P8_WGRAD"); return !(e && e[0] == '0'); }();
    const bool want_p8 = big && p8w && p->dtype == HS_BF16 && p->M % 256 == 0 && p->N % 256 == 0 && p->K % 64 == 0 && p->K >= 2048 &&
                         (p->batch <= 1) && p->split_k <= 1 && g_dbg_cfg < 0 && (g->items.empty() || g->combo != 0 || g->cfg == CFG_P8_256);
    HS_PROPAGATE(gemm_prepare(p, s, q, big ? (want_p8 ? CFG_P8_256 : CFG_256x128) : -1));
    if (want_p8 && !(p8_epilogue_supported(q.a) && !q.a.colstats && !q.a.bnb_partials && (long long)p->M * p->lda * 2 < 0x7fffff00ll &&
                     (long long)p->N * p->ldb * 2 < 0x7fffff00ll))
        HS_PROPAGATE(gemm_prepare(p, s, q, CFG_256x128));
    const bool ok = big ? (q.bf16 && (q.cfg == CFG_256x128 || q.cfg == CFG_P8_256) && q.combo == 0 && q.batch == 1 && q.split == 1 && !q.a.stamps && !q.a.colstats)
                        : (q.bf16 && q.cfg == CFG_64x64 && (q.combo == 2 || q.combo == 5) && q.batch == 1 && !q.a.stamps &&
                           (q.split == 1 || q.a.tickets != nullptr) && !q.a.rowsum[0] && !q.a.colstats);
    if (!ok) return launch_ungrouped(p, s);
    const long long ngroups = q.split > 1 ? (q.split + kSplitGroup - 1) / kSplitGroup : 0;
    const long long need = q.split > 1 ? (long long)q.a.tiles_m * q.a.tiles_n * (1 + (ngroups > 1 ? ngroups : 0)) : 0;
    if (!g->items.empty() && (g->combo != q.combo || (big && g->cfg != q.cfg) || (int)g->items.size() >= kGroupMax || g->ticket_off + need > kTicketPool)) {
        if (g_group_immediate_hook) HS_PROPAGATE(g_group_immediate_hook(g_group_immediate_ctx));
        HS_PROPAGATE(gemm_group_flush(g, s));
    }
    if (need > kTicketPool) return launch_ungrouped(p, s);
    g->combo = q.combo;
    if (big) {
        g->cfg = q.cfg;
        g->rs = g->rs || q.a.rowsum[0] != nullptr;
    }
    if (q.split > 1) q.a.tickets += g->ticket_off;
    g->ticket_off += (int)need;
    g->first.push_back(g->total);
    g->total += (int)(q.grid.x * q.grid.z);
    g->items.push_back(q.a);
    g->flops += q.flops;
    const double R2 = q.conv ? (double)p->g.R * p->g.S : 1.0, st2 = q.conv ? (double)p->g.stride * p->g.stride : 1.0;
    g->bytes += 2.0 * ((double)p->M * p->K + (double)p->N * p->K / R2 * st2) + 4.0 * (double)p->M * p->N;
    return HS_OK;
}

}  // namespace hs

extern "C" {
/* measurement only: device buffer of 6 x (workgroups of the largest launch) uint64, or NULL to switch the stamps off */
void hs_gemm_debug_stamps(void* device_buffer) { hs::g_dbg_stamps = (unsigned long long*)device_buffer; }
void hs_gemm_debug(int32_t cfg_override, int32_t ablate) {
    hs::g_dbg_cfg = cfg_override;
    hs::g_dbg_ablate = ablate;
}
void hs_prof_enable(int32_t on) {
    std::lock_guard<std::mutex> lk(hs::g_prof_mu);
    hs::g_prof_on = on != 0;
}
/* Synchronises the device, then sums flops / milliseconds / launches per class (4 entries each) and clears. */
hs_status hs_prof_collect(double* flops, double* ms, int64_t* launches) {
    HS_CHECK_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(hs::g_prof_mu);
    for (int i = 0; i < 4; ++i) {
        flops[i] = 0;
        ms[i] = 0;
        launches[i] = 0;
    }
    for (auto& r : hs::g_prof) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            flops[r.cls] += r.flops;
            ms[r.cls] += t;
            launches[r.cls] += 1;
        }
        hs::g_prof_pool.emplace_back(r.a, r.b);
    }
    hs::g_prof.clear();
    return HS_OK;
}
/* measurement only: what the event bracket itself costs.  Launches `n` empty kernels on `stream`, each between the same
   two event records the profiler puts around a GEMM launch, and returns the mean elapsed time in microseconds (an empty
   kernel runs for ~2 us; the rest is dispatch latency the bracket adds to every timed launch). */
hs_status hs_prof_calibrate(void* stream, int32_t n, float* avg_us) {
    HS_REQUIRE(n > 0 && avg_us, "hs_prof_calibrate: bad argument");
    hipStream_t s = (hipStream_t)stream;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev(n);
    for (auto& e : ev) {
        HS_CHECK_HIP(hipEventCreate(&e.first));
        HS_CHECK_HIP(hipEventCreate(&e.second));
    }
    for (auto& e : ev) {
        HS_CHECK_HIP(hipEventRecord(e.first, s));
        hipLaunchKernelGGL(hs::null_kernel, dim3(1), dim3(64), 0, s);
        HS_CHECK_HIP(hipEventRecord(e.second, s));
    }
    HS_CHECK_HIP(hipStreamSynchronize(s));
    double tot = 0;
    for (auto& e : ev) {
        float t = 0.f;
        HS_CHECK_HIP(hipEventElapsedTime(&t, e.first, e.second));
        tot += t;
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    *avg_us = (float)(tot / n * 1e3);
    return HS_OK;
}
/* Synchronises, appends one CSV line per recorded launch (cls,combo,cfg,M,N,K,batch,split,R,stride,ms,flops,bytes; a grouped launch: combo 6 / 7 / 8 = grouped tn / conv / nt weight gradients, M = problems in it) to `path`, clears. */
hs_status hs_prof_dump(const char* path) {
    HS_CHECK_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(hs::g_prof_mu);
    FILE* f = fopen(path, "a");
    HS_REQUIRE(f != nullptr, "hs_prof_dump: cannot open %s", path);
    for (auto& r : hs::g_prof) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess)
            fprintf(f, "%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%.6f,%.0f,%.0f\n", r.cls, r.combo, r.cfg, r.M, r.N, r.K, r.batch, r.split, r.conv_r,
                    r.conv_stride, t, r.flops, r.bytes);
        hs::g_prof_pool.emplace_back(r.a, r.b);
    }
    hs::g_prof.clear();
    fclose(f);
    return HS_OK;
}
const char* hs_last_error(void) { return hs::last_error(); }
int hs_version(void) { return 100; }
int hs_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 0;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}
hs_status hs_gemm(const hs_gemm_params* p, void* stream) { return hs::gemm_impl(p, (hipStream_t)stream); }
int32_t hs_gemm_stat_rows(const hs_gemm_params* p) { return hs::gemm_stat_rows(p); }
int32_t hs_gemm_bn_finish_rows(const hs_gemm_params* p) { return hs::gemm_bn_finish_rows(p); }
int32_t hs_gemm_bnb_finish_rows(const hs_gemm_params* p) { return hs::gemm_bnb_finish_rows(p); }
int32_t hs_gemm_tile_rows(const hs_gemm_params* p) { return hs::gemm_tile_rows(p); }
int64_t hs_gemm_splitk_ws_bytes(const hs_gemm_params* p) {
    if (!p || p->split_k <= 1) return 0;
    return (int64_t)hs::splitk_slabs(p->split_k) * p->M * p->N * 4;   // slice slabs + the group slabs of the in-launch reduction
}
int32_t hs_gemm_suggest_split(int32_t M, int32_t N, int32_t K, int32_t dtype) {
    const int bk = dtype == HS_BF16 ? 64 : 32;
    const long long tiles = (long long)hs::ceil_div(M, 64) * hs::ceil_div(N, 64);
    const int ktiles = hs::ceil_div(K, bk);
    if (tiles >= 256 || ktiles < 16) return 1;
    static const long long units = [] {          // measurement: HAMSPINE_SPLIT_UNITS overrides the target number of workgroups
        const char* e = getenv("HAMSPINE_SPLIT_UNITS");
        const long long v = e ? atoll(e) : 0;
        return v > 0 ? v : 512ll;
    }();
    long long s = (units + tiles - 1) / tiles;   // 256/384/768 work units measured the same on C2 with the separate reduce pass
    const long long smax = ktiles / 8 > 0 ? ktiles / 8 : 1;   // keep >= 8 k-tiles per split
    if (s > smax) s = smax;
    static const long long cap = [] {            // measurement: HAMSPINE_SPLIT_MAX overrides the largest split
        const char* e = getenv("HAMSPINE_SPLIT_MAX");
        const long long v = e ? atoll(e) : 0;
        return v > 0 ? v : 64ll;
    }();
    if (s > cap) s = cap;
    return (int32_t)(s < 1 ? 1 : s);
}
}
