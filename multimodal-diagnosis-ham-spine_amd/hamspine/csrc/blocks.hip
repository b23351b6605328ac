// Composite executors: attention core, Linear fwd/bwd, ResNet residual block, ResNet stem and
// BertLayer, each launched from C++ on top of the op-level ABI (hs_gemm, hs_batchnorm_*, ...).
// Buffers come from two caller-provided arenas: `saved` (activations kept for the backward) and `ws`
// (scratch).  The same layout code runs in "plan" mode (no launches) to answer the *_query calls, so
// sizes and pointers cannot drift apart.
#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <vector>
#include <stdlib.h>
#include "hs_common.h"
#include "pw_stream.h"

namespace hs {

// dst = [a | b | c], n floats each (fused QKV bias)
__global__ __launch_bounds__(256) void gather3_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ c, float* __restrict__ dst, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * n) return;
    const int w = i / n, k = i - w * n;
    dst[i] = w == 0 ? a[k] : (w == 1 ? b[k] : c[k]);
}

int gemm_impl(const hs_gemm_params* p, hipStream_t stream);
int gemm_stat_rows(const hs_gemm_params* p);
int gemm_bn_finish_rows(const hs_gemm_params* p);
int gemm_tile_rows(const hs_gemm_params* p);
int gemm_bnb_finish_rows(const hs_gemm_params* p);
void gemm_group_set_immediate_hook(int (*fn)(void*), void* ctx);
struct GemmGroup;
GemmGroup* gemm_group_open(hipStream_t s, long long slot);   // slot: any key that is stable across steps (its device table is cached)
int gemm_group_add(GemmGroup* g, const hs_gemm_params* p, hipStream_t s);
int gemm_group_flush(GemmGroup* g, hipStream_t s);
int attention_bwd_fused(const hs_attn_desc& d, const void* q, const void* k, const void* v, const void* dO, void* dq, void* dk,
                        void* dv, const void* P, int ldP, hipStream_t s);
int attention_fwd_fused(const hs_attn_desc& d, const void* q, const void* k, const void* v, void* o, void* P, void* Pd, int ldP,
                        hipStream_t s);

static inline long long align_up(long long v, long long a) { return (v + a - 1) / a * a; }
static inline int esize(int dt) { return dt == HS_BF16 ? 2 : 4; }

struct Arena {
    char* base = nullptr;
    long long cap = 0, off = 0, peak = 0;
    bool plan = false;
    bool overflow = false;
    void* alloc(long long bytes) {
        bytes = align_up(bytes > 0 ? bytes : 1, 256);
        char* p = base ? base + off : nullptr;
        off += bytes;
        if (off > peak) peak = off;
        if (!plan && off > cap) overflow = true;
        return plan ? nullptr : p;
    }
    long long mark() const { return off; }
    void release(long long m) { off = m; }
};

struct Run {
    hipStream_t s;
    int dt;
    bool plan;
    Arena saved, ws;
    // second stream for work that is off the critical path of a backward (weight / bias gradients): it is
    // forked behind the producers of its inputs and joined before the composite returns
    hipStream_t side = nullptr;
    bool side_used = false;
    // tower backward: weight-gradient GEMMs are collected here and launched as grouped grids when the tower's data-gradient
    // chain is done (their operands -- saved activations and the ws gradients, which are then not released -- stay alive)
    bool defer_wgrad = false;
    GemmGroup *grp_pw = nullptr, *grp_conv = nullptr;
    // BertLayer backward: the layer's four K-contiguous weight-gradient GEMMs are collected and run as ONE grouped grid at the
    // end of the layer (group_nt is set in plan mode too: it decides what scratch the layer needs)
    bool group_nt = false;
    GemmGroup* grp_nt = nullptr;
    // BERT tower backward: the group above belongs to the TOWER and spans several layers (bert_bwd_run opens and flushes it);
    // a layer then neither opens nor flushes, and the tower keeps the layers' transposed operands allocated until the flush
    GemmGroup* tower_grp = nullptr;
    // ... and the dY^T transposes those GEMMs read: collected here and made by ONE launch in front of the grouped grid (the
    // row-major gradients stay allocated until then) instead of one launch each where the gradient appears
    struct PendTr { const void* src; void* dst; int R, C; long long ld; };
    std::vector<PendTr> pend_tr;
};

// one non-blocking side stream and a ring of events per device (events are re-recordable; every composite joins
// before it returns, so a ring of 64 cannot wrap onto a pending event)
static hipStream_t g_side_stream[16] = {};
static hipEvent_t g_events[16][64] = {};
static int g_event_next[16] = {};
static std::mutex g_side_mu;
// Weight-gradient GEMMs of a composite on a side stream beside its data-gradient chain.  OFF by default since the GEMM
// epilogue work: with workgroups that live 7.8 instead of 12.4 us a single GEMM fills the chip well enough that running two
// at once only makes them contend (C2 step 14.5 ms with the side stream, 12.6 ms without; it was 21.5 -> 19.7 ms the other
// way round when it was introduced).  HAMSPINE_OVERLAP=1 / hs_set_overlap(1) switch it on.
static int g_overlap = -1;   // -1: take HAMSPINE_OVERLAP from the environment on first use
static bool overlap_enabled() {
    if (g_overlap < 0) {
        const char* e = getenv("HAMSPINE_OVERLAP");
        g_overlap = (e && e[0] == '1') ? 1 : 0;
    }
    return g_overlap == 1;
}
static int side_setup(Run& r) {
    if (r.plan || !overlap_enabled()) return HS_OK;
    int dev = 0;
    HS_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) return HS_OK;
    std::lock_guard<std::mutex> lk(g_side_mu);
    if (!g_side_stream[dev]) HS_CHECK_HIP(hipStreamCreateWithFlags(&g_side_stream[dev], hipStreamNonBlocking));
    r.side = g_side_stream[dev];
    return HS_OK;
}
static int next_event(hipEvent_t* ev) {
    int dev = 0;
    HS_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_side_mu);
    int& n = g_event_next[dev & 15];
    hipEvent_t& e = g_events[dev & 15][n];
    n = (n + 1) & 63;
    if (!e) HS_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *ev = e;
    return HS_OK;
}
// run `fn` on the side stream, ordered after everything launched so far on the main stream
template <typename F>
static int on_side(Run& r, F fn) {
    if (r.plan || !r.side) return fn();
    hipEvent_t ev;
    HS_PROPAGATE(next_event(&ev));
    HS_CHECK_HIP(hipEventRecord(ev, r.s));
    HS_CHECK_HIP(hipStreamWaitEvent(r.side, ev, 0));
    hipStream_t keep = r.s;
    r.s = r.side;
    const int st = fn();
    r.s = keep;
    r.side_used = true;
    return st;
}
static int side_join(Run& r) {
    if (r.plan || !r.side || !r.side_used) return HS_OK;
    hipEvent_t ev;
    HS_PROPAGATE(next_event(&ev));
    HS_CHECK_HIP(hipEventRecord(ev, r.side));
    HS_CHECK_HIP(hipStreamWaitEvent(r.s, ev, 0));
    r.side_used = false;
    return HS_OK;
}
static void run_init(Run& r, int dt, bool plan, void* saved, long long saved_bytes, void* ws, long long ws_bytes,
                     hipStream_t s) {
    r.s = s;
    r.dt = dt;
    r.plan = plan;
    r.saved.base = (char*)saved;
    r.saved.cap = saved_bytes;
    r.saved.plan = plan;
    r.ws.base = (char*)ws;
    r.ws.cap = ws_bytes;
    r.ws.plan = plan;
}
#define RUN_CHECK_ARENAS(r, what)                                                                          \
    do {                                                                                                   \
        if (!(r).plan && ((r).saved.overflow || (r).ws.overflow)) {                                        \
            set_error("%s: arena too small (saved %lld/%lld, ws %lld/%lld)", what, (r).saved.peak,          \
                      (r).saved.cap, (r).ws.peak, (r).ws.cap);                                             \
            return HS_ERR_ARG;                                                                             \
        }                                                                                                  \
    } while (0)
#define CALL(r, expr)                        \
    do {                                     \
        if (!(r).plan) HS_PROPAGATE(expr);   \
    } while (0)
// Measurement only (HAMSPINE_KNOCKOUT=<bit mask>, off by default): leave a class of launches OUT of the step to read its
// marginal cost in the overlapped schedule off the step time (results are then garbage; tools/knockout.sh).  Bits: 1 ResNet
// weight gradients, 2 BERT weight-gradient GEMMs, 4 transposes, 8 BatchNorm forward, 16 BatchNorm backward, 64 weight casts,
// 128 LayerNorm backward, 256 attention cores, 512 ResNet data-gradient convolutions, 1024 BERT data-gradient GEMMs.
// The switch exists only in measurement builds (make EXTRA=-DHS_MEASURE, what tools/knockout.sh builds): a release library
// ignores the variable, hs_measure_build() reports which one is loaded and bench.py refuses to run on a measurement build.
#ifdef HS_MEASURE
static int knock() {
    static const int v = [] { const char* e = getenv("HAMSPINE_KNOCKOUT"); return e ? atoi(e) : 0; }();
    return v;
}
#else
static constexpr int knock() { return 0; }
#endif
#define CALLK(r, bit, expr)                                        \
    do {                                                           \
        if (!(r).plan && !(knock() & (bit))) HS_PROPAGATE(expr);   \
    } while (0)

// ---- gradient milestones (hs_grad_milestones) ---------------------------------------------------------------------
// A data-parallel caller that exchanges gradients bucket by bucket wants to start a bucket's exchange as soon as the LAST
// gradient of that bucket has been enqueued -- not when the whole tower's backward has.  It hands the next tower backward on
// this thread a list of (parameter-gradient pointer, HIP event): the executor records the event on its stream right after
// the kernels that finish that gradient have been enqueued (BERT: at the end of the layer that owns the pointer, or after
// the embeddings; ResNet: the weight gradients run as grouped grids behind the data-gradient chain, so all at the end).
// Entries left over when the executor returns are recorded there, so the caller can always wait on every event.
constexpr int kMaxMilestones = 64;
struct Milestones {
    int n = 0;
    const void* ptr[kMaxMilestones];
    hipEvent_t ev[kMaxMilestones];
};
static thread_local Milestones g_ms;
static int milestones_hit(Run& r, const void* const* ptrs, int np) {     // record the events whose pointer is among ptrs
    if (r.plan) return HS_OK;
    for (int m = 0; m < g_ms.n;) {
        bool hit = false;
        for (int k = 0; k < np && !hit; ++k) hit = ptrs[k] && ptrs[k] == g_ms.ptr[m];
        if (hit) {
            HS_CHECK_HIP(hipEventRecord(g_ms.ev[m], r.s));
            g_ms.ptr[m] = g_ms.ptr[g_ms.n - 1];
            g_ms.ev[m] = g_ms.ev[g_ms.n - 1];
            --g_ms.n;
        } else {
            ++m;
        }
    }
    return HS_OK;
}
static int milestones_flush(Run& r) {                                    // record every remaining event
    if (r.plan) return HS_OK;
    for (int m = 0; m < g_ms.n; ++m) HS_CHECK_HIP(hipEventRecord(g_ms.ev[m], r.s));
    g_ms.n = 0;
    return HS_OK;
}

static hs_gemm_params gemm_defaults(int dt) {
    hs_gemm_params p;
    memset(&p, 0, sizeof(p));
    p.dtype = dt;
    p.out_dtype = dt;
    p.alpha = 1.f;
    p.batch = 1;
    p.batch_inner = 1;
    return p;
}
// wgrad-style GEMMs reduce over a long K into a small output: pick split-K, borrow slabs from ws.
static int gemm_splitk(Run& r, hs_gemm_params& p, bool may_defer = false) {
    const int split = hs_gemm_suggest_split(p.M, p.N, p.K, p.dtype);
    const long long mk = r.ws.mark();
    p.split_k = split;
    if (split > 1) p.splitk_ws = (float*)r.ws.alloc(hs_gemm_splitk_ws_bytes(&p));
    const bool defer = may_defer && r.defer_wgrad && p.dtype == HS_BF16;
    if (!r.plan) {
        if (r.ws.overflow) {
            set_error("gemm_splitk: workspace too small");
            return HS_ERR_ARG;
        }
        if (may_defer && (knock() & 1)) {
        } else if (defer) HS_PROPAGATE(gemm_group_add(p.b_kind == HS_B_CONV ? r.grp_conv : r.grp_pw, &p, r.s));
        else HS_PROPAGATE(gemm_impl(&p, r.s));
    }
    if (defer) return HS_OK;                 // the slabs stay allocated until the grouped launch has run
    // with a side stream, slabs of consecutive GEMMs may be live concurrently: keep them until the composite ends
    if (r.plan ? !overlap_enabled() : !r.side) r.ws.release(mk);
    return HS_OK;
}

// ============================================================================================
// attention core
// ============================================================================================
struct AttnLayout {
    int ldP;
    void *P, *Pd;        // saved
    float* S;            // ws (forward) / dP (backward)
    void* dS;            // ws (backward)
};
static AttnLayout attn_layout(const hs_attn_desc& d, Run& r, bool backward) {
    AttnLayout l;
    const long long BH = (long long)d.B * d.H;
    l.ldP = (int)align_up(d.Lk, 8);
    l.P = r.saved.alloc(BH * d.Lq * l.ldP * esize(d.dtype));
    l.Pd = d.dropout_p > 0.f ? r.saved.alloc(BH * d.Lq * l.ldP * esize(d.dtype)) : l.P;
    l.S = (float*)r.ws.alloc(BH * d.Lq * d.Lk * 4);
    l.dS = backward ? r.ws.alloc(BH * d.Lq * l.ldP * esize(d.dtype)) : nullptr;
    return l;
}
static int attn_check(const hs_attn_desc& d) {
    HS_REQUIRE(d.B > 0 && d.H > 0 && d.Lq > 0 && d.Lk > 0 && d.hd > 0, "attention: bad dims");
    HS_REQUIRE(d.Lk <= 1024, "attention: Lk=%d > 1024 unsupported", d.Lk);
    return HS_OK;
}
static long long span_elems(int B, long long bs, int L, int ld, int H, int hd) {
    return (long long)(B - 1) * bs + (long long)(L - 1) * ld + (long long)H * hd;
}

static int attention_fwd_run(Run& r, const hs_attn_desc& d, const void* q, const void* k, const void* v, void* o) {
    HS_PROPAGATE(attn_check(d));
    AttnLayout l = attn_layout(d, r, false);
    const int BH = d.B * d.H;
    if (!r.plan && (knock() & 256)) return HS_OK;
    if (!r.plan && !r.saved.overflow && !r.ws.overflow) {
        // BERT shape (bf16, head dim 64, <= 128 tokens): one fused kernel, scores stay in registers (csrc/attn_fused.hip)
        const int fused = attention_fwd_fused(d, q, k, v, o, l.P, l.Pd, l.ldP, r.s);
        if (fused < 0) {
            set_error("attention_fwd: fused kernel launch failed");
            return HS_ERR_HIP;
        }
        if (fused == 1) {
            RUN_CHECK_ARENAS(r, "attention_fwd");
            return HS_OK;
        }
    }
    // S = scale * Q K^T
    hs_gemm_params p = gemm_defaults(d.dtype);
    p.a_kind = HS_A_KC; p.b_kind = HS_B_KC;
    p.M = d.Lq; p.N = d.Lk; p.K = d.hd;
    p.A = q; p.B = k;
    p.a_elems = span_elems(d.B, d.q_bs, d.Lq, d.q_ld, d.H, d.hd);
    p.b_elems = span_elems(d.B, d.k_bs, d.Lk, d.k_ld, d.H, d.hd);
    p.lda = d.q_ld; p.ldb = d.k_ld;
    p.batch = BH; p.batch_inner = d.H;
    p.a_bs0 = d.q_bs; p.a_bs1 = d.hd;
    p.b_bs0 = d.k_bs; p.b_bs1 = d.hd;
    p.d_bs0 = (long long)d.H * d.Lq * d.Lk; p.d_bs1 = (long long)d.Lq * d.Lk;
    p.D = l.S; p.ldd = d.Lk; p.out_dtype = HS_F32;
    p.alpha = d.scale;
    CALL(r, gemm_impl(&p, r.s));
    CALL(r, hs_softmax_fwd(d.dtype, l.S, d.key_mask, l.P, d.dropout_p > 0.f ? l.Pd : nullptr, (long long)BH * d.Lq, d.Lk,
                           d.Lk, l.ldP, d.H * d.Lq, d.dropout_p, d.seed, r.s));
    // O = Pd V
    hs_gemm_params g = gemm_defaults(d.dtype);
    g.a_kind = HS_A_KC; g.b_kind = HS_B_RC;
    g.M = d.Lq; g.N = d.hd; g.K = d.Lk;
    g.A = l.Pd; g.B = v;
    g.a_elems = (long long)BH * d.Lq * l.ldP;
    g.b_elems = span_elems(d.B, d.v_bs, d.Lk, d.v_ld, d.H, d.hd);
    g.lda = l.ldP; g.ldb = d.v_ld;
    g.batch = BH; g.batch_inner = d.H;
    g.a_bs0 = (long long)d.H * d.Lq * l.ldP; g.a_bs1 = (long long)d.Lq * l.ldP;
    g.b_bs0 = d.v_bs; g.b_bs1 = d.hd;
    g.d_bs0 = d.o_bs; g.d_bs1 = d.hd;
    g.D = o; g.ldd = d.o_ld;
    CALL(r, gemm_impl(&g, r.s));
    RUN_CHECK_ARENAS(r, "attention_fwd");
    return HS_OK;
}

static int attention_bwd_run(Run& r, const hs_attn_desc& d, const void* q, const void* k, const void* v, const void* dO,
                             void* dq, void* dk, void* dv) {
    HS_PROPAGATE(attn_check(d));
    AttnLayout l = attn_layout(d, r, true);
    const int BH = d.B * d.H;
    const long long o_el = span_elems(d.B, d.o_bs, d.Lq, d.o_ld, d.H, d.hd);
    const long long p_el = (long long)BH * d.Lq * l.ldP;
    if (!r.plan && (knock() & 256)) return HS_OK;
    if (!r.plan && !r.saved.overflow && !r.ws.overflow) {
        // BERT shape: one fused kernel (csrc/attn_fused.hip) instead of four batched GEMMs and the softmax backward
        const int fused = attention_bwd_fused(d, q, k, v, dO, dq, dk, dv, l.P, l.ldP, r.s);
        if (fused < 0) {
            set_error("attention_bwd: fused kernel launch failed");
            return HS_ERR_HIP;
        }
        if (fused == 1) {
            RUN_CHECK_ARENAS(r, "attention_bwd");
            return HS_OK;
        }
    }
    // dPd = dO V^T  (f32)
    hs_gemm_params p = gemm_defaults(d.dtype);
    p.a_kind = HS_A_KC; p.b_kind = HS_B_KC;
    p.M = d.Lq; p.N = d.Lk; p.K = d.hd;
    p.A = dO; p.B = v;
    p.a_elems = o_el;
    p.b_elems = span_elems(d.B, d.v_bs, d.Lk, d.v_ld, d.H, d.hd);
    p.lda = d.o_ld; p.ldb = d.v_ld;
    p.batch = BH; p.batch_inner = d.H;
    p.a_bs0 = d.o_bs; p.a_bs1 = d.hd;
    p.b_bs0 = d.v_bs; p.b_bs1 = d.hd;
    p.d_bs0 = (long long)d.H * d.Lq * d.Lk; p.d_bs1 = (long long)d.Lq * d.Lk;
    p.D = l.S; p.ldd = d.Lk; p.out_dtype = HS_F32;
    CALL(r, gemm_impl(&p, r.s));
    // dV = Pd^T dO
    hs_gemm_params gv = gemm_defaults(d.dtype);
    gv.a_kind = HS_A_RC; gv.b_kind = HS_B_RC;
    gv.M = d.Lk; gv.N = d.hd; gv.K = d.Lq;
    gv.A = l.Pd; gv.B = dO;
    gv.a_elems = p_el; gv.b_elems = o_el;
    gv.lda = l.ldP; gv.ldb = d.o_ld;
    gv.batch = BH; gv.batch_inner = d.H;
    gv.a_bs0 = (long long)d.H * d.Lq * l.ldP; gv.a_bs1 = (long long)d.Lq * l.ldP;
    gv.b_bs0 = d.o_bs; gv.b_bs1 = d.hd;
    gv.d_bs0 = d.v_bs; gv.d_bs1 = d.hd;
    gv.D = dv; gv.ldd = d.v_ld;
    CALL(r, gemm_impl(&gv, r.s));
    // dS = softmax'(P, dPd)
    CALL(r, hs_softmax_bwd(d.dtype, l.S, l.P, l.dS, (long long)BH * d.Lq, d.Lk, d.Lk, l.ldP, d.dropout_p, d.seed, r.s));
    // dQ = scale * dS K
    hs_gemm_params gq = gemm_defaults(d.dtype);
    gq.a_kind = HS_A_KC; gq.b_kind = HS_B_RC;
    gq.M = d.Lq; gq.N = d.hd; gq.K = d.Lk;
    gq.A = l.dS; gq.B = k;
    gq.a_elems = p_el;
    gq.b_elems = span_elems(d.B, d.k_bs, d.Lk, d.k_ld, d.H, d.hd);
    gq.lda = l.ldP; gq.ldb = d.k_ld;
    gq.batch = BH; gq.batch_inner = d.H;
    gq.a_bs0 = (long long)d.H * d.Lq * l.ldP; gq.a_bs1 = (long long)d.Lq * l.ldP;
    gq.b_bs0 = d.k_bs; gq.b_bs1 = d.hd;
    gq.d_bs0 = d.q_bs; gq.d_bs1 = d.hd;
    gq.D = dq; gq.ldd = d.q_ld;
    gq.alpha = d.scale;
    CALL(r, gemm_impl(&gq, r.s));
    // dK = scale * dS^T Q
    hs_gemm_params gk = gemm_defaults(d.dtype);
    gk.a_kind = HS_A_RC; gk.b_kind = HS_B_RC;
    gk.M = d.Lk; gk.N = d.hd; gk.K = d.Lq;
    gk.A = l.dS; gk.B = q;
    gk.a_elems = p_el;
    gk.b_elems = span_elems(d.B, d.q_bs, d.Lq, d.q_ld, d.H, d.hd);
    gk.lda = l.ldP; gk.ldb = d.q_ld;
    gk.batch = BH; gk.batch_inner = d.H;
    gk.a_bs0 = (long long)d.H * d.Lq * l.ldP; gk.a_bs1 = (long long)d.Lq * l.ldP;
    gk.b_bs0 = d.q_bs; gk.b_bs1 = d.hd;
    gk.d_bs0 = d.k_bs; gk.d_bs1 = d.hd;
    gk.D = dk; gk.ldd = d.k_ld;
    gk.alpha = d.scale;
    CALL(r, gemm_impl(&gk, r.s));
    RUN_CHECK_ARENAS(r, "attention_bwd");
    return HS_OK;
}

// ============================================================================================
// Linear
// ============================================================================================
static int linear_fwd_run(Run& r, const void* x, long long M, int ldx, const hs_linear& lin, const void* w_c, void* y,
                          int ldy, int out_dtype, int act, void* preact, const void* residual, int ldr, float drop_p,
                          unsigned long long seed) {
    hs_gemm_params p = gemm_defaults(r.dt);
    p.a_kind = HS_A_KC; p.b_kind = HS_B_KC;
    p.M = (int)M; p.N = lin.out_f; p.K = lin.in_f;
    p.A = x; p.B = w_c;
    p.a_elems = (M - 1) * ldx + lin.in_f;
    p.b_elems = (long long)lin.out_f * lin.in_f;
    p.lda = ldx; p.ldb = lin.in_f;
    p.D = y; p.ldd = ldy; p.out_dtype = out_dtype;
    p.bias = lin.b;
    p.act = act;
    p.D_preact = preact;
    p.residual = residual; p.ldr = ldr;
    p.dropout_p = drop_p; p.dropout_seed = seed;
    CALL(r, gemm_impl(&p, r.s));
    return HS_OK;
}
// parameter gradients: dw = dy^T x, db = colsum(dy)
static bool fused_bias_grad_enabled() {     // HAMSPINE_FUSED_BIAS_GRAD=0: bias gradients by separate column-sum passes
    static const bool on = [] {
        const char* e = getenv("HAMSPINE_FUSED_BIAS_GRAD");
        return !(e && e[0] == '0');
    }();
    return on;
}
// a weight-gradient GEMM (bf16, unsplit) can produce the bias gradient as the row sums of its A operand
static bool bias_in_wgrad(int dt, int out_f, int in_f, long long M) {
    return dt == HS_BF16 && fused_bias_grad_enabled() && hs_gemm_suggest_split(out_f, in_f, (int)M, dt) <= 1;
}
// Caller-scoped grouping (hs_wgrad_group_begin / _end): between the two calls the K-contiguous weight-gradient GEMMs that
// hs_linear_bwd issues on that stream are collected and run as one grouped grid at _end (an MLP's two Linear layers: two
// under-filled launches become one).  The caller keeps the workspaces of the collected calls alive and distinct until _end.
static std::mutex g_wg_mu;
static std::unordered_map<hipStream_t, GemmGroup*> g_wg_open;
static GemmGroup* open_wgrad_group(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_wg_mu);
    auto it = g_wg_open.find(s);
    return it == g_wg_open.end() ? nullptr : it->second;
}
static bool wgrad_nt_enabled();
static int linear_wgrad_nt_run(Run& r, const void* xT, const void* dyT, long long M, int in_f, int out_f, float* dw, float* db,
                               int seg_rows, float* const* dw_seg, float* const* db_seg);
static int linear_wgrad_run(Run& r, const void* x, long long M, int ldx, const hs_linear& lin, const void* dy, int ldy) {
    bool db_done = false;
    // bf16, compute-bound shapes (in*out/(in+out) FLOP per operand byte above the ridge; ConvNeXt's 512 <-> 2048 MLPs over
    // 12544 tokens: 172 us as a "tn" GEMM with transposed LDS reads): transpose dY and X once (one launch) and run the
    // K-contiguous GEMM, as the BertLayer backward does.  Long-K / small-output shapes stay on the split-K "tn" form (they are
    // HBM-bound and the transposes would double their traffic).
    if (lin.dw && r.dt == HS_BF16 && wgrad_nt_enabled() && M >= 2048 && M % 8 == 0 && lin.in_f % 8 == 0 && lin.out_f % 8 == 0 &&
        ldx % 8 == 0 && ldy % 8 == 0 && ldx >= lin.in_f && ldy >= lin.out_f &&
        (long long)lin.in_f * lin.out_f >= 256ll * (lin.in_f + lin.out_f)) {
        const long long mk = r.ws.mark();
        char* tA = (char*)r.ws.alloc(M * (long long)lin.out_f * 2);
        char* tB = (char*)r.ws.alloc(M * (long long)lin.in_f * 2);
        const void* src[2] = {dy, x};
        void* dst[2] = {tA, tB};
        const int32_t R[2] = {(int32_t)M, (int32_t)M}, Cc[2] = {lin.out_f, lin.in_f};
        const int64_t lds[2] = {ldy, ldx}, ldd[2] = {M, M};
        CALLK(r, 4, hs_transpose_bf16_multi(2, src, dst, R, Cc, lds, ldd, r.s));
        GemmGroup* og = r.plan ? nullptr : open_wgrad_group(r.s);
        const bool keep_flag = r.group_nt;
        GemmGroup* keep_grp = r.grp_nt;
        if (og) {
            r.group_nt = true;
            r.grp_nt = og;
        }
        const bool fused_b = lin.db && (r.group_nt || hs_gemm_suggest_split(lin.out_f, lin.in_f, (int)M, r.dt) <= 1) && fused_bias_grad_enabled();
        const int st_nt = linear_wgrad_nt_run(r, tB, tA, M, lin.in_f, lin.out_f, lin.dw, lin.db, 0, nullptr, nullptr);
        r.group_nt = keep_flag;
        r.grp_nt = keep_grp;
        HS_PROPAGATE(st_nt);
        if (lin.db && !fused_b) {
            const long long wsb = hs_colsum_ws_bytes(M, lin.out_f);
            void* w = r.ws.alloc(wsb);
            CALL(r, hs_colsum(r.dt, dy, M, lin.out_f, ldy, lin.db, w, wsb, 0, r.s));
        }
        // the transposed operands stay allocated while a caller-scoped group may still read them (the plan pass cannot know:
        // it always keeps them, so a plan is never smaller than the run)
        if (!r.plan && !og && !r.side) r.ws.release(mk);
        return HS_OK;
    }
    if (lin.dw) {
        hs_gemm_params p = gemm_defaults(r.dt);
        p.a_kind = HS_A_RC; p.b_kind = HS_B_RC;
        p.M = lin.out_f; p.N = lin.in_f; p.K = (int)M;
        p.A = dy; p.B = x;
        p.a_elems = (M - 1) * ldy + lin.out_f;
        p.b_elems = (M - 1) * ldx + lin.in_f;
        p.lda = ldy; p.ldb = ldx;
        p.D = lin.dw; p.ldd = lin.in_f; p.out_dtype = HS_F32;
        if (lin.db && bias_in_wgrad(r.dt, lin.out_f, lin.in_f, M)) {
            p.rowsum_a = lin.db;
            db_done = true;
        }
        HS_PROPAGATE(gemm_splitk(r, p));
    }
    if (lin.db && !db_done) {
        const long long mk = r.ws.mark();
        const long long wsb = hs_colsum_ws_bytes(M, lin.out_f);
        void* w = r.ws.alloc(wsb);
        CALL(r, hs_colsum(r.dt, dy, M, lin.out_f, ldy, lin.db, w, wsb, 0, r.s));
        if (r.plan ? !overlap_enabled() : !r.side) r.ws.release(mk);
    }
    return HS_OK;
}
// The same gradients from TRANSPOSED operands (bf16): dyT [out][M] and xT [in][M] are both K-contiguous along the token
// dimension, so dW = dyT . xT^T runs on the ds_read_b128 operand path (measured 1.5-1.7x the [k][row] "tn" form on
// BERT-base shapes, tools/gemm_sweep.py).  The bias gradient is the row sums of dyT (one extra MFMA per fragment).
static int g_wgrad_nt = -1;                  // -1: HAMSPINE_WGRAD_NT from the environment (default on); hs_set_wgrad_nt overrides
static bool wgrad_nt_enabled() {             // off: weight gradients from the row-major operands ("tn")
    if (g_wgrad_nt < 0) {
        const char* e = getenv("HAMSPINE_WGRAD_NT");
        g_wgrad_nt = (e && e[0] == '0') ? 0 : 1;
    }
    return g_wgrad_nt == 1;
}
static bool grouped_bert_wgrad_enabled() {   // HAMSPINE_GROUPED_BERT_WGRAD=0: every weight-gradient GEMM of a BertLayer on its own
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("HAMSPINE_GROUPED_BERT_WGRAD");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}
static bool dgrad_nt_enabled() {          // HAMSPINE_DGRAD_NT=0: BERT data gradients read the row-contiguous weights
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("HAMSPINE_DGRAD_NT");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}
static bool deferred_transposes_enabled() {     // HAMSPINE_DEFER_TRANSPOSES=0: one launch per dY^T where the gradient appears
    static const bool on = [] { const char* e = getenv("HAMSPINE_DEFER_TRANSPOSES"); return !(e && e[0] == '0'); }();
    return on;
}
static int transpose_run(Run& r, const void* src, void* dst, long long R, int Cc, int ld_src, bool for_grouped_wgrad = false) {
    if (for_grouped_wgrad && r.tower_grp && r.group_nt && !r.plan && deferred_transposes_enabled()) {       // its reader is the tower's grouped grid: see pend_tr
        r.pend_tr.push_back(Run::PendTr{src, dst, (int)R, Cc, ld_src});
        return HS_OK;
    }
    CALLK(r, 4, hs_transpose_bf16(src, dst, (int)R, Cc, ld_src, R, r.s));
    return HS_OK;
}
static int pending_transposes_flush(Run& r) {
    const int n = (int)r.pend_tr.size();
    if (n == 0) return HS_OK;
    std::vector<const void*> src(n);
    std::vector<void*> dst(n);
    std::vector<int32_t> R(n), Cc(n);
    std::vector<int64_t> lds(n), ldd(n);
    for (int i = 0; i < n; ++i) {
        src[i] = r.pend_tr[i].src; dst[i] = r.pend_tr[i].dst; R[i] = r.pend_tr[i].R; Cc[i] = r.pend_tr[i].C;
        lds[i] = r.pend_tr[i].ld; ldd[i] = r.pend_tr[i].R;
    }
    r.pend_tr.clear();
    CALLK(r, 4, hs_transpose_bf16_multi(n, src.data(), dst.data(), R.data(), Cc.data(), lds.data(), ldd.data(), r.s));
    return HS_OK;
}
// seg: 0 = plain; 3 = fused Q/K/V (lin = the q layer; D_seg / rowsum_seg = k, v)
static int linear_wgrad_nt_run(Run& r, const void* xT, const void* dyT, long long M, int in_f, int out_f, float* dw, float* db,
                               int seg_rows, float* const* dw_seg, float* const* db_seg) {
    hs_gemm_params p = gemm_defaults(r.dt);
    p.a_kind = HS_A_KC; p.b_kind = HS_B_KC;
    p.M = out_f; p.N = in_f; p.K = (int)M;
    p.A = dyT; p.B = xT;
    p.a_elems = (long long)out_f * M;
    p.b_elems = (long long)in_f * M;
    p.lda = (int)M; p.ldb = (int)M;
    p.D = dw; p.ldd = in_f; p.out_dtype = HS_F32;
    if (seg_rows > 0) {
        p.seg_rows = seg_rows;
        p.D_seg[0] = dw_seg[0];
        p.D_seg[1] = dw_seg[1];
    }
    if (db && (r.group_nt || hs_gemm_suggest_split(out_f, in_f, (int)M, r.dt) <= 1) && fused_bias_grad_enabled()) {
        p.rowsum_a = db;
        if (seg_rows > 0) {
            p.rowsum_seg[0] = db_seg[0];
            p.rowsum_seg[1] = db_seg[1];
        }
    }
    if (r.group_nt) {                    // no split-K inside a grouped grid: the group fills the chip
        if (!r.plan && !(knock() & 2)) HS_PROPAGATE(gemm_group_add(r.grp_nt, &p, r.s));
        return HS_OK;
    }
    if (!r.plan && (knock() & 2)) { const int sp = hs_gemm_suggest_split(p.M, p.N, p.K, p.dtype); p.split_k = sp; if (sp > 1) (void)r.ws.alloc(hs_gemm_splitk_ws_bytes(&p)); return HS_OK; }
    return gemm_splitk(r, p);
}
// dx = dy W (* multiplier) (+ residual)
// w_t (optional): W^T [in_f][out_f] in the compute dtype -- both operands K-contiguous (ds_read_b128 fragments instead of the
// transposed ds_read_b64_tr_b16 reads of a row-contiguous W: 36.5 vs 47.5 us on 4096x768x3072)
static int linear_dgrad_run(Run& r, const hs_linear& lin, const void* w_c, const void* dy, long long M, int ldy, void* dx,
                            int lddx, int dx_dtype, int mul_mode, const void* mul_src, int ldm, const void* residual,
                            const void* w_t = nullptr) {
    hs_gemm_params p = gemm_defaults(r.dt);
    p.a_kind = HS_A_KC; p.b_kind = HS_B_RC;
    p.M = (int)M; p.N = lin.in_f; p.K = lin.out_f;
    p.A = dy; p.B = w_c;
    p.a_elems = (M - 1) * ldy + lin.out_f;
    p.b_elems = (long long)lin.out_f * lin.in_f;
    p.lda = ldy; p.ldb = lin.in_f;
    long long mk = -1;
    if (!w_t && r.dt == HS_BF16 && dgrad_nt_enabled() && lin.out_f >= 1536 && M >= 2048 && lin.in_f % 8 == 0 && lin.out_f % 8 == 0) {
        // long K (= out_f) under many rows: one small transpose of the weight copy buys the K-contiguous operand path
        // (ConvNeXt MLP 12544 x 512 x 2048: 73.6 us as "nn", 48.8 us as "nt")
        mk = r.ws.mark();
        char* wt = (char*)r.ws.alloc((long long)lin.in_f * lin.out_f * 2);
        HS_PROPAGATE(transpose_run(r, w_c, wt, lin.out_f, lin.in_f, lin.in_f));
        w_t = wt;
    }
    if (w_t) {
        p.b_kind = HS_B_KC;
        p.B = w_t;
        p.ldb = lin.out_f;
    }
    p.D = dx; p.ldd = lddx; p.out_dtype = dx_dtype;
    p.mul_mode = mul_mode; p.mul_src = mul_src; p.ldm = ldm;
    p.residual = residual; p.ldr = lddx;
    CALLK(r, 1024, gemm_impl(&p, r.s));
    if (mk >= 0 && (r.plan ? !overlap_enabled() : !r.side)) r.ws.release(mk);
    return HS_OK;
}

// ============================================================================================
// convolution helpers (NHWC, filters KRSC in the compute dtype)
// ============================================================================================
struct ConvShape {
    int N, H, W, Cin, Cout, R, stride, pad, P, Q;
};
static ConvShape conv_shape(int N, int H, int W, const hs_conv_bn& c) {
    ConvShape s;
    s.N = N; s.H = H; s.W = W; s.Cin = c.Cin; s.Cout = c.Cout; s.R = c.R; s.stride = c.stride; s.pad = c.pad;
    s.P = (H + 2 * c.pad - c.R) / c.stride + 1;
    s.Q = (W + 2 * c.pad - c.R) / c.stride + 1;
    return s;
}
static hs_conv_geom geom_of(const ConvShape& s) {
    hs_conv_geom g;
    memset(&g, 0, sizeof(g));
    g.N = s.N; g.H = s.H; g.W = s.W; g.C = s.Cin; g.P = s.P; g.Q = s.Q; g.K = s.Cout; g.R = s.R; g.S = s.R;
    g.stride = s.stride; g.pad = s.pad;
    g.row_pitch = s.W * s.Cin;
    g.img_pitch = s.H * g.row_pitch;
    g.qstep = s.stride * s.Cin;
    return g;
}
static bool is_pointwise(const ConvShape& s) { return s.R == 1 && s.stride == 1 && s.pad == 0; }
static bool fused_ln_bwd_enabled() {        // HAMSPINE_FUSED_LN_BWD=0: separate LayerNorm-backward / dropout / column-sum passes
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("HAMSPINE_FUSED_LN_BWD");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}
static bool grouped_wgrad_enabled() {     // HAMSPINE_GROUPED_WGRAD=0: the tower backward launches every weight gradient on its own
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("HAMSPINE_GROUPED_WGRAD");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}
static bool bn_mask_from_x_enabled() {     // HAMSPINE_BN_MASK_FROM_X=1: inner-stage BatchNorm backward recomputes the ReLU mask from the
    static int v = -1;                     // convolution output instead of reading the saved activation (one read less per pass;
    if (v < 0) {                           // measured neutral on C2: 11.68-11.72 vs 11.66-11.70 ms -- off by default)
        const char* e = getenv("HAMSPINE_BN_MASK_FROM_X");
        v = (e && e[0] == '1') ? 1 : 0;
    }
    return v == 1;
}
static bool fused_bn_stats_enabled() {     // HAMSPINE_FUSED_BN_STATS=0: BatchNorm makes its own statistics pass
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("HAMSPINE_FUSED_BN_STATS");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

// stats (optional, bf16 mode): receives the (count, mean, M2) partials of y per row tile for the BatchNorm that follows;
// *stat_rows is set to the number of partial rows (0: not produced, BatchNorm makes its own pass)
struct FoldedBn {          // eval-mode BatchNorm folded into the convolution: y = act(conv * scale + shift (+ identity))
    const float* scale = nullptr;
    const float* shift = nullptr;
    int relu = 0;
    const void* identity = nullptr;
};
static bool bn_finish_enabled() {     // HAMSPINE_BN_FINISH=0: BatchNorm finishes its statistics in its own launch
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("HAMSPINE_BN_FINISH");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}
// bn (optional, with stats): the BatchNorm that follows; when the launch can it finishes the statistics too (bn->stats_done)
static int conv_fwd_run(Run& r, const ConvShape& s, const void* x, const void* w_c, void* y, float* stats = nullptr,
                        int* stat_rows = nullptr, const FoldedBn* fold = nullptr, hs_bn_params* bn = nullptr) {
    // optional: 1x1 convolutions of a training step through the streaming kernel (pw_stream.hip: persistent workgroups,
    // A-stationary row blocks, one partial statistics row per workgroup; the BatchNorm finishes them in its own small launch)
    static const bool pw_on = [] { const char* e = getenv("HAMSPINE_PW_STREAM"); return e && e[0] == '1'; }();   // measured: a wash, off
    if (pw_on && !fold && is_pointwise(s) && r.dt == HS_BF16 && stats && stat_rows && fused_bn_stats_enabled()) {
        const long long Mo = (long long)s.N * s.P * s.Q;
        const PwPlan pl = pw_stream_plan(Mo, s.Cout, s.Cin, true);
        if (pl.ok) {
            *stat_rows = pl.stat_rows;
            if (!r.plan) {
                PwArgs a;
                memset(&a, 0, sizeof(a));
                a.A = (const char*)x; a.W = (const char*)w_c; a.D = (char*)y;
                a.a_bytes = (unsigned long long)Mo * s.Cin * 2;
                a.w_bytes = (unsigned long long)s.Cout * s.Cin * 2;
                a.lda = s.Cin; a.ldd = s.Cout;
                a.M = (int)Mo; a.N = s.Cout; a.K = s.Cin;
                a.nblocks = (int)((Mo + 63) / 64);
                a.stats = stats;
                HS_PROPAGATE(pw_stream_run(pl, a, r.s));
            }
            return HS_OK;
        }
    }
    hs_gemm_params p = gemm_defaults(r.dt);
    if (fold) {
        p.colscale = fold->scale;
        p.bias = fold->shift;
        p.act = fold->relu ? HS_ACT_RELU : HS_ACT_NONE;
        p.residual = fold->identity;
        p.ldr = s.Cout;
        p.residual_before_act = 1;
    }
    const long long Mo = (long long)s.N * s.P * s.Q;
    p.M = (int)Mo; p.N = s.Cout; p.K = s.R * s.R * s.Cin;
    p.A = x; p.B = w_c;
    p.a_elems = (long long)s.N * s.H * s.W * s.Cin;
    p.b_elems = (long long)s.Cout * p.K;
    p.ldb = p.K;
    p.D = y; p.ldd = s.Cout;
    if (is_pointwise(s)) {
        p.a_kind = HS_A_KC; p.lda = s.Cin;
    } else {
        p.a_kind = HS_A_CONV; p.g = geom_of(s);
    }
    p.b_kind = HS_B_KC;
    if (stat_rows) *stat_rows = 0;
    if (stats && stat_rows && fused_bn_stats_enabled()) {
        const int rows = gemm_stat_rows(&p);
        if (rows > 0) {
            p.colstats = stats;
            *stat_rows = rows;
            if (bn && bn->training && bn_finish_enabled()) {
                const int frows = gemm_bn_finish_rows(&p);
                if (frows > 0 && bn->ws == (void*)stats && bn->ws_bytes >= (long long)frows * s.Cout * 3 * 4) {
                    p.bn_finish = bn;
                    bn->stats_done = 1;
                }
            }
        }
    }
    CALL(r, gemm_impl(&p, r.s));
    return HS_OK;
}
// bnb (optional): dx is the gradient of relu(bn(c)) of the PREVIOUS stage -- the GEMM's epilogue also takes that BatchNorm's
// backward sums (hs_gemm_params.bnb_*); *bnb->rows receives the number of partial rows (0: not produced)
struct BnSums {
    const void* c;
    const float *scale, *shift, *mean, *invstd;
    float* partials;
    long long partials_cap;     // bytes available at `partials` (incl. the 4*C coefficient floats behind the sums)
    int* rows;
    const void* y = nullptr;    // block-output form: dx (+ residual) is d relu(bn(c) + identity); the mask is y > 0 (hs_gemm_params.bnb_y)
    // optional: the BatchNorm whose sums these are (C, M, training, gamma, dgamma, dbeta) -- the GEMM's last workgroups then also
    // finish the sums (hs_gemm_params.bnb_finish) and *done = 1: hs_batchnorm_bwd runs its apply pass only
    const hs_bn_bwd_params* finish = nullptr;
    int* done = nullptr;
};
// MEASURED (round 3, rocprofv3, un-overlapped C2 step): with the sums finished inside the data-gradient GEMM the 37 rider launches
// take 1.42 instead of 1.20 ms (+5.9 us each: write-through stores -> counter round trip -> the last workgroup's dependent
// reads) and 32 bn_bwd_final launches (0.19 ms, 5.8 us each) go away: 14.18 vs 14.04 ms of kernels, 11.08 vs 11.08 ms per
// overlapped step.  A per-layer reduction dependency costs ~6 us as a launch or as an in-launch tail.  OFF unless
// HAMSPINE_BNB_FINISH=1.
static bool bnb_finish_enabled() {
    static const bool on = [] { const char* e = getenv("HAMSPINE_BNB_FINISH"); return e && e[0] == '1'; }();
    return on;
}
static bool fused_bn_bwd_enabled() {        // HAMSPINE_FUSED_BN_BWD=0: BatchNorm backward makes its own partial-sum pass
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("HAMSPINE_FUSED_BN_BWD");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}
static int conv_dgrad_run(Run& r, const ConvShape& s, const void* dy, const void* w_c, void* dx, const void* residual,
                          const BnSums* bnb = nullptr) {
    hs_gemm_params p = gemm_defaults(r.dt);
    p.M = (int)((long long)s.N * s.H * s.W); p.N = s.Cin; p.K = s.R * s.R * s.Cout;
    p.A = dy; p.B = w_c;
    p.a_elems = (long long)s.N * s.P * s.Q * s.Cout;
    p.b_elems = (long long)s.Cout * s.R * s.R * s.Cin;
    p.D = dx; p.ldd = s.Cin;
    p.residual = residual; p.ldr = s.Cin;
    if (is_pointwise(s)) {
        p.a_kind = HS_A_KC; p.lda = s.Cout;
        p.b_kind = HS_B_RC; p.ldb = s.Cin;
    } else {
        p.a_kind = HS_A_DGRAD; p.b_kind = HS_B_WDGRAD; p.g = geom_of(s);
    }
    // late layers: few output tiles under a long K (7x7 maps: 200 tiles x 72 K tiles) -- split-K, reduced inside the launch
    // with the full epilogue (the parity-ordered stride-2 walk has its own K schedule and stays whole)
    if (bnb && bnb->rows) *bnb->rows = 0;
    if (!r.plan && (knock() & 512)) return HS_OK;
    if (bnb && (!residual || bnb->y) && r.dt == HS_BF16 && s.stride == 1 && s.Cin % 8 == 0 && fused_bn_bwd_enabled() && !(knock() & 16)) {
        p.split_k = hs_gemm_suggest_split(p.M, p.N, p.K, p.dtype);             // what gemm_splitk will choose
        hs_gemm_params probe = p;
        probe.bnb_x = bnb->c; probe.bnb_scale = bnb->scale; probe.bnb_shift = bnb->shift; probe.bnb_mean = bnb->mean;
        probe.bnb_invstd = bnb->invstd; probe.bnb_partials = bnb->partials; probe.bnb_y = bnb->y;
        const int rows = r.plan ? 0 : gemm_tile_rows(&probe);
        if (bnb->done) *bnb->done = 0;
        if (rows > 0 && ((long long)rows * s.Cin * 2 + 4ll * s.Cin) * 4 <= bnb->partials_cap) {
            p = probe;
            *bnb->rows = rows;
            const int frows = (bnb->finish && bnb->done && bnb_finish_enabled()) ? gemm_bnb_finish_rows(&probe) : 0;
            if (frows > 0 && ((long long)frows * s.Cin * 2 + 4ll * s.Cin) * 4 <= bnb->partials_cap) {
                p.bnb_finish = bnb->finish;
                *bnb->rows = frows;
                *bnb->done = 1;
            }
        }
    }
    if (r.dt == HS_BF16 && s.stride == 1) return gemm_splitk(r, p);
    CALL(r, gemm_impl(&p, r.s));
    return HS_OK;
}
static int conv_wgrad_run(Run& r, const ConvShape& s, const void* dy, const void* x, float* dw) {
    hs_gemm_params p = gemm_defaults(r.dt);
    p.M = s.Cout; p.N = s.R * s.R * s.Cin; p.K = (int)((long long)s.N * s.P * s.Q);
    p.A = dy; p.B = x;
    p.a_elems = (long long)s.N * s.P * s.Q * s.Cout;
    p.b_elems = (long long)s.N * s.H * s.W * s.Cin;
    p.a_kind = HS_A_RC; p.lda = s.Cout;
    p.D = dw; p.ldd = p.N; p.out_dtype = HS_F32;
    if (is_pointwise(s)) {
        p.b_kind = HS_B_RC; p.ldb = s.Cin;
    } else {
        p.b_kind = HS_B_CONV; p.g = geom_of(s);
    }
    return gemm_splitk(r, p, true);
}

// weights in the compute dtype: f32 mode reads the parameters directly, bf16 mode casts them into
// the saved arena once per forward (the backward reuses the copies).
struct CastList {
    const float* src[HS_CAST_MAX];
    void* dst[HS_CAST_MAX];
    int64_t n[HS_CAST_MAX];
    int count = 0;
};
// bf16 weight SHADOWS: a caller that keeps a bf16 copy of a parameter up to date itself (hamspine.rt: the fused optimizer
// writes it together with the f32 update, the towers re-cast it when anything else touched the parameter) registers it here by
// the parameter's address; the composites then read the shadow instead of casting the weight again in every forward.  The
// slot in the saved arena is still reserved, so an arena layout never depends on what is registered.
static std::mutex g_shadow_mu;
static std::unordered_map<const float*, void*> g_shadow;
static void* shadow_of(const float* w) {
    std::lock_guard<std::mutex> lk(g_shadow_mu);
    auto it = g_shadow.find(w);
    return it == g_shadow.end() ? nullptr : it->second;
}
static const void* weight_c(Run& r, CastList& cl, const float* w, long long n) {
    if (r.dt == HS_F32) return w;
    void* d = r.saved.alloc(n * 2);
    if (void* sh = shadow_of(w)) return sh;
    cl.src[cl.count] = w;
    cl.dst[cl.count] = d;
    cl.n[cl.count] = n;
    cl.count++;
    return d;
}
static int run_casts(Run& r, CastList& cl) {
    if (cl.count == 0 || r.plan) return HS_OK;
    if (knock() & 64) return HS_OK;
    return hs_cast_f32_to_bf16_multi(cl.count, cl.src, cl.dst, cl.n, r.s);
}

// ============================================================================================
// residual block
// ============================================================================================
struct StageBuf {
    ConvShape s;
    const void* w_c;
    void* c;          // conv output (pre-BN)
    void* a;          // BN(+ReLU) output (the last main stage writes the block output y instead)
    float *mean, *invstd, *scale, *shift;
};
struct ResLayout {
    StageBuf main[3];
    StageBuf ds;
    void* bn_ws;
    long long bn_ws_bytes;
    CastList casts;
};
static void stage_layout(Run& r, ResLayout& L, StageBuf& b, const hs_conv_bn& cb, int N, int H, int W, bool own_out) {
    b.s = conv_shape(N, H, W, cb);
    b.w_c = weight_c(r, L.casts, cb.w, (long long)cb.Cout * cb.R * cb.R * cb.Cin);
    const long long Mo = (long long)N * b.s.P * b.s.Q;
    b.c = r.saved.alloc(Mo * cb.Cout * esize(r.dt));
    b.a = own_out ? r.saved.alloc(Mo * cb.Cout * esize(r.dt)) : nullptr;
    float* st = (float*)r.saved.alloc((long long)cb.Cout * 4 * 4);
    b.mean = st;
    b.invstd = st ? st + cb.Cout : nullptr;
    b.scale = st ? st + 2 * cb.Cout : nullptr;
    b.shift = st ? st + 3 * cb.Cout : nullptr;
    L.bn_ws_bytes = std::max<long long>(L.bn_ws_bytes, hs_batchnorm_ws_bytes(Mo, cb.Cout, r.dt));
}
static int res_layout(Run& r, const hs_resblock_desc& d, ResLayout& L) {
    HS_REQUIRE(d.n_main == 2 || d.n_main == 3, "resblock: n_main must be 2 or 3");
    L.bn_ws_bytes = 0;
    int H = d.H, W = d.W;
    for (int i = 0; i < d.n_main; ++i) {
        stage_layout(r, L, L.main[i], d.main[i], d.N, H, W, i + 1 < d.n_main);
        H = L.main[i].s.P;
        W = L.main[i].s.Q;
        if (i > 0) HS_REQUIRE(d.main[i].Cin == d.main[i - 1].Cout, "resblock: channel mismatch at stage %d", i);
    }
    if (d.has_ds) {
        stage_layout(r, L, L.ds, d.ds, d.N, d.H, d.W, true);
        HS_REQUIRE(L.ds.s.P == H && L.ds.s.Q == W && d.ds.Cout == d.main[d.n_main - 1].Cout && d.ds.Cin == d.main[0].Cin,
                   "resblock: downsample shape mismatch");
    } else {
        HS_REQUIRE(H == d.H && W == d.W && d.main[0].Cin == d.main[d.n_main - 1].Cout,
                   "resblock: identity shortcut needs matching shapes");
    }
    L.bn_ws = r.ws.alloc(L.bn_ws_bytes);
    return HS_OK;
}
static hs_bn_params bn_params(Run& r, const hs_resblock_desc& d, const hs_conv_bn& cb, const StageBuf& b, long long M) {
    hs_bn_params p;
    memset(&p, 0, sizeof(p));
    p.dtype = r.dt; p.C = cb.Cout; p.M = M;
    p.training = d.training;
    p.eps = d.eps; p.momentum = d.momentum;
    p.x = b.c;
    p.gamma = cb.gamma; p.beta = cb.beta;
    p.running_mean = cb.running_mean; p.running_var = cb.running_var;
    p.save_mean = b.mean; p.save_invstd = b.invstd; p.scale = b.scale; p.shift = b.shift;
    return p;
}

static int resblock_fwd_run(Run& r, const hs_resblock_desc& d, const void* x, void* y) {
    ResLayout L;
    HS_PROPAGATE(res_layout(r, d, L));
    RUN_CHECK_ARENAS(r, "resblock_fwd");
    HS_PROPAGATE(run_casts(r, L.casts));
    const void* identity = x;
    if (d.inference && !d.training) {
        // no backward follows: scale/shift from the running statistics, then ONE GEMM per convolution with BatchNorm,
        // ReLU and the identity add in its epilogue
        auto folded = [&](const hs_conv_bn& cb, StageBuf& b, const void* in, void* out, int relu, const void* ident) -> int {
            const long long Mo = (long long)d.N * b.s.P * b.s.Q;
            hs_bn_params bp = bn_params(r, d, cb, b, Mo);
            bp.y = nullptr;                                   // statistics only: fills b.scale / b.shift
            bp.ws = L.bn_ws; bp.ws_bytes = L.bn_ws_bytes;
            CALLK(r, 8, hs_batchnorm_fwd(&bp, r.s));
            FoldedBn f;
            f.scale = b.scale; f.shift = b.shift; f.relu = relu; f.identity = ident;
            return conv_fwd_run(r, b.s, in, b.w_c, out, nullptr, nullptr, &f);
        };
        if (d.has_ds) {
            HS_PROPAGATE(folded(d.ds, L.ds, x, L.ds.a, 0, nullptr));
            identity = L.ds.a;
        }
        const void* in = x;
        for (int i = 0; i < d.n_main; ++i) {
            const bool last = i + 1 == d.n_main;
            HS_PROPAGATE(folded(d.main[i], L.main[i], in, last ? y : L.main[i].a, 1, last ? identity : nullptr));
            in = L.main[i].a;
        }
        return HS_OK;
    }
    if (d.has_ds) {
        const long long Mo = (long long)d.N * L.ds.s.P * L.ds.s.Q;
        int srows = 0;
        hs_bn_params bp = bn_params(r, d, d.ds, L.ds, Mo);
        bp.y = L.ds.a; bp.relu = 0;
        bp.ws = L.bn_ws; bp.ws_bytes = L.bn_ws_bytes;
        HS_PROPAGATE(conv_fwd_run(r, L.ds.s, x, L.ds.w_c, L.ds.c, d.training ? (float*)L.bn_ws : nullptr, &srows, nullptr, &bp));
        bp.partial_rows = srows;
        CALLK(r, 8, hs_batchnorm_fwd(&bp, r.s));
        identity = L.ds.a;
    }
    const void* in = x;
    for (int i = 0; i < d.n_main; ++i) {
        StageBuf& b = L.main[i];
        const long long Mo = (long long)d.N * b.s.P * b.s.Q;
        int srows = 0;
        hs_bn_params bp = bn_params(r, d, d.main[i], b, Mo);
        const bool last = i + 1 == d.n_main;
        bp.y = last ? y : b.a;
        bp.relu = 1;
        bp.residual = last ? identity : nullptr;
        bp.ws = L.bn_ws; bp.ws_bytes = L.bn_ws_bytes;
        HS_PROPAGATE(conv_fwd_run(r, b.s, in, b.w_c, b.c, d.training ? (float*)L.bn_ws : nullptr, &srows, nullptr, &bp));
        bp.partial_rows = srows;
        CALLK(r, 8, hs_batchnorm_fwd(&bp, r.s));
        in = b.a;
    }
    return HS_OK;
}

// next_bn (optional): dx is the gradient of the PREVIOUS block's output relu(bn(c3) + identity); the stage-0 data-gradient GEMM
// that writes dx also takes that BatchNorm's backward sums (block-output form of BnSums).  own (optional, own_rows > 0): this
// block's own final BatchNorm finds its sums already there (produced by the next block's backward).
struct OwnSums {
    float* partials = nullptr;
    long long bytes = 0;
    int rows = 0;
    int done = 0;      // the producing GEMM also finished the sums
};
static int resblock_bwd_run(Run& r, const hs_resblock_desc& d, const void* x, const void* y, const void* dy, void* dx,
                            const BnSums* next_bn = nullptr, const OwnSums* own = nullptr) {
    ResLayout L;
    HS_PROPAGATE(res_layout(r, d, L));
    const int last = d.n_main - 1;
    const int es = esize(r.dt);
    const long long Mout = (long long)d.N * L.main[last].s.P * L.main[last].s.Q;
    const int Cout = d.main[last].Cout;
    // scratch: gradient wrt the residual branch, stage gradients, BN parameter-gradient fallbacks
    void* dres = r.ws.alloc(Mout * Cout * es);
    float* scratch_pg = (float*)r.ws.alloc(2 * 4096 * 4);
    long long max_act = 0;
    for (int i = 0; i < d.n_main; ++i) {
        const StageBuf& b = L.main[i];
        max_act = std::max<long long>(max_act, (long long)d.N * b.s.P * b.s.Q * b.s.Cout);
        max_act = std::max<long long>(max_act, (long long)d.N * b.s.H * b.s.W * b.s.Cin);
    }
    if (d.has_ds) max_act = std::max<long long>(max_act, Mout * Cout);
    // one gradient buffer per stage (no ping-pong): the weight-gradient GEMMs run on the side stream while the
    // main stream continues with dgrad + BN backward of the next stage
    void* g_conv[3] = {nullptr, nullptr, nullptr};   // d(conv output) of main stage i
    void* g_in[3] = {nullptr, nullptr, nullptr};     // d(input) of main stage i (i >= 1)
    for (int i = 0; i < d.n_main; ++i) {
        const StageBuf& b = L.main[i];
        g_conv[i] = r.ws.alloc((long long)d.N * b.s.P * b.s.Q * b.s.Cout * es);
        if (i > 0) g_in[i] = r.ws.alloc((long long)d.N * b.s.H * b.s.W * b.s.Cin * es);
    }
    void* g_ds = d.has_ds ? r.ws.alloc(Mout * Cout * es) : nullptr;
    void* dx_ds = (d.has_ds && dx) ? r.ws.alloc((long long)d.N * d.H * d.W * d.main[0].Cin * es) : nullptr;
    (void)max_act;
    for (int i = 0; i < d.n_main; ++i) HS_REQUIRE(d.main[i].Cout <= 4096, "resblock: Cout > 4096");
    HS_PROPAGATE(side_setup(r));

    auto bn_backward = [&](const hs_conv_bn& cb, const StageBuf& b, long long M, const void* g_out, const void* y_out,
                           int relu, void* g_in_, void* g_res, int partial_rows = 0, const OwnSums* pre = nullptr,
                           int sums_done = 0) -> int {
        hs_bn_bwd_params q;
        memset(&q, 0, sizeof(q));
        q.partial_rows = partial_rows;
        q.sums_done = partial_rows > 0 ? sums_done : 0;
        q.dtype = r.dt; q.C = cb.Cout; q.M = M;
        q.training = d.training; q.relu = relu;
        q.dy = g_out; q.y = y_out; q.x = b.c;
        q.gamma = cb.gamma; q.save_mean = b.mean; q.save_invstd = b.invstd;
        q.dx = g_in_; q.dres = g_res;
        if (relu && !g_res && bn_mask_from_x_enabled()) {     // inner stages: relu(bn(c)) with no residual -- the sign of the
            q.y = nullptr;                                    // forward value is recomputed from c (one read less per pass)
            q.scale = b.scale;
            q.shift = b.shift;
        }
        q.dgamma = cb.dgamma ? cb.dgamma : scratch_pg;
        q.dbeta = cb.dbeta ? cb.dbeta : (scratch_pg ? scratch_pg + 4096 : nullptr);
        q.ws = L.bn_ws; q.ws_bytes = L.bn_ws_bytes;
        if (pre && pre->rows > 0) {               // the sums came with the gradient (another block's data-gradient GEMM)
            q.partial_rows = pre->rows;
            q.sums_done = pre->done;
            q.ws = pre->partials; q.ws_bytes = pre->bytes;
        }
        CALLK(r, 16, hs_batchnorm_bwd(&q, r.s));
        return HS_OK;
    };

    // last main stage: y = relu(bn(c_last) + identity)
    HS_PROPAGATE(bn_backward(d.main[last], L.main[last], Mout, dy, y, 1, g_conv[last], dres, 0, own));
    const void* dres_for_x = dres;   // identity shortcut: dres flows straight into dx
    if (d.has_ds) {
        // downsample branch: identity = bn(conv(x)), no ReLU
        HS_PROPAGATE(bn_backward(d.ds, L.ds, Mout, dres, nullptr, 0, g_ds, nullptr));
        if (d.ds.dw) HS_PROPAGATE(on_side(r, [&]() { return conv_wgrad_run(r, L.ds.s, g_ds, x, d.ds.dw); }));
        if (dx) HS_PROPAGATE(conv_dgrad_run(r, L.ds.s, g_ds, L.ds.w_c, dx_ds, nullptr));
        dres_for_x = dx_ds;
    }
    // walk the main path backwards
    for (int i = last; i >= 0; --i) {
        const StageBuf& b = L.main[i];
        const void* in = i == 0 ? x : L.main[i - 1].a;
        if (d.main[i].dw) HS_PROPAGATE(on_side(r, [&]() { return conv_wgrad_run(r, b.s, g_conv[i], in, d.main[i].dw); }));
        if (i == 0) {
            if (dx) HS_PROPAGATE(conv_dgrad_run(r, b.s, g_conv[i], b.w_c, dx, dres_for_x, next_bn));
        } else {
            // the data gradient of stage i is d relu(bn(c)) of stage i - 1: its GEMM takes that BatchNorm's backward sums along
            const StageBuf& pb = L.main[i - 1];
            int sum_rows = 0, sum_done = 0;
            const long long Mp = (long long)d.N * pb.s.P * pb.s.Q;
            const hs_conv_bn& pcb = d.main[i - 1];
            hs_bn_bwd_params fin;                 // what the GEMM's finishing workgroups need of that BatchNorm
            memset(&fin, 0, sizeof(fin));
            fin.dtype = r.dt; fin.C = pcb.Cout; fin.M = Mp; fin.training = d.training;
            fin.gamma = pcb.gamma;
            fin.dgamma = pcb.dgamma ? pcb.dgamma : scratch_pg;
            fin.dbeta = pcb.dbeta ? pcb.dbeta : (scratch_pg ? scratch_pg + 4096 : nullptr);
            BnSums bs = {pb.c, pb.scale, pb.shift, pb.mean, pb.invstd, (float*)L.bn_ws, L.bn_ws_bytes, &sum_rows};
            if (fin.dgamma && fin.dbeta) {
                bs.finish = &fin;
                bs.done = &sum_done;
            }
            HS_PROPAGATE(conv_dgrad_run(r, b.s, g_conv[i], b.w_c, g_in[i], nullptr, &bs));
            HS_PROPAGATE(bn_backward(d.main[i - 1], pb, Mp, g_in[i], pb.a, 1, g_conv[i - 1], nullptr, sum_rows, nullptr, sum_done));
        }
    }
    HS_PROPAGATE(side_join(r));
    RUN_CHECK_ARENAS(r, "resblock_bwd");
    return HS_OK;
}

// ============================================================================================
// stem
// ============================================================================================
struct StemLayout {
    int Hp, Wp, P, Q, P2, Q2;
    void* packed;      // [N][Hp][Wp][4]
    void* w_packed;    // [64][7][8][4] compute dtype
    void* c;           // conv out [N][P][Q][64]
    void* a;           // bn+relu out
    void* idx;         // maxpool argmax
    float *mean, *invstd, *scale, *shift;
    void* bn_ws;
    long long bn_ws_bytes;
    hs_conv_geom g;
};
static int stem_layout(Run& r, const hs_stem_desc& d, StemLayout& L) {
    const hs_conv_bn& cb = d.cb;
    HS_REQUIRE(cb.Cin == 3 && cb.R == 7 && cb.stride == 2 && cb.pad == 3 && cb.Cout % 8 == 0 && cb.Cout <= 64,
               "stem: expects conv 7x7/2 pad 3 with 3 input channels and <= 64 output channels");
    L.P = (d.H + 6 - 7) / 2 + 1;
    L.Q = (d.W + 6 - 7) / 2 + 1;
    L.Hp = std::max(d.H + 6, 2 * L.P + 5);
    L.Wp = (int)align_up(std::max(d.W + 6, 2 * L.Q + 6), 2);
    L.P2 = (L.P + 2 - 3) / 2 + 1;
    L.Q2 = (L.Q + 2 - 3) / 2 + 1;
    const int es = esize(r.dt);
    L.packed = r.saved.alloc((long long)d.N * L.Hp * L.Wp * 4 * es);
    L.w_packed = r.saved.alloc((long long)cb.Cout * 7 * 32 * es);
    const long long Mo = (long long)d.N * L.P * L.Q;
    L.c = r.saved.alloc(Mo * cb.Cout * es);
    L.a = r.saved.alloc(Mo * cb.Cout * es);
    L.idx = r.saved.alloc((long long)d.N * L.P2 * L.Q2 * cb.Cout);
    float* st = (float*)r.saved.alloc((long long)cb.Cout * 4 * 4);
    L.mean = st;
    L.invstd = st ? st + cb.Cout : nullptr;
    L.scale = st ? st + 2 * cb.Cout : nullptr;
    L.shift = st ? st + 3 * cb.Cout : nullptr;
    L.bn_ws_bytes = hs_batchnorm_ws_bytes(Mo, cb.Cout, r.dt);
    L.bn_ws = r.ws.alloc(L.bn_ws_bytes);
    hs_conv_geom& g = L.g;
    memset(&g, 0, sizeof(g));
    g.N = d.N; g.H = L.Hp; g.W = L.Wp; g.C = 32; g.P = L.P; g.Q = L.Q; g.K = cb.Cout; g.R = 7; g.S = 1;
    g.stride = 2; g.pad = 0;
    g.row_pitch = L.Wp * 4;
    g.img_pitch = L.Hp * L.Wp * 4;
    g.qstep = 8;
    g.no_bounds = 1;
    return HS_OK;
}
static int stem_fwd_run(Run& r, const hs_stem_desc& d, const float* image, void* y) {
    StemLayout L;
    HS_PROPAGATE(stem_layout(r, d, L));
    RUN_CHECK_ARENAS(r, "stem_fwd");
    const hs_conv_bn& cb = d.cb;
    CALL(r, hs_pack_image(r.dt, image, L.packed, d.N, 3, d.H, d.W, L.Hp, L.Wp, 3, r.s));
    CALL(r, hs_pack_stem_weight(r.dt, cb.w, L.w_packed, cb.Cout, 7, 7, 3, r.s));
    const long long Mo = (long long)d.N * L.P * L.Q;
    hs_gemm_params p = gemm_defaults(r.dt);
    p.a_kind = HS_A_CONV; p.b_kind = HS_B_KC;
    p.M = (int)Mo; p.N = cb.Cout; p.K = 7 * 32;
    p.A = L.packed; p.B = L.w_packed;
    p.a_elems = (long long)d.N * L.Hp * L.Wp * 4;
    p.b_elems = (long long)cb.Cout * 224;
    p.ldb = 224;
    p.g = L.g;
    p.D = L.c; p.ldd = cb.Cout;
    if (d.inference && !d.training) {
        hs_bn_params be;
        memset(&be, 0, sizeof(be));
        be.dtype = r.dt; be.C = cb.Cout; be.M = Mo; be.training = 0; be.eps = d.eps; be.momentum = d.momentum;
        be.x = L.c; be.y = nullptr;
        be.gamma = cb.gamma; be.beta = cb.beta; be.running_mean = cb.running_mean; be.running_var = cb.running_var;
        be.save_mean = L.mean; be.save_invstd = L.invstd; be.scale = L.scale; be.shift = L.shift;
        be.ws = L.bn_ws; be.ws_bytes = L.bn_ws_bytes;
        CALLK(r, 8, hs_batchnorm_fwd(&be, r.s));
        p.colscale = L.scale; p.bias = L.shift; p.act = HS_ACT_RELU;
        p.D = L.a;                                            // BatchNorm + ReLU folded: the GEMM writes the pooled input
        CALL(r, gemm_impl(&p, r.s));
        CALL(r, hs_maxpool_fwd(r.dt, L.a, y, L.idx, d.N, L.P, L.Q, cb.Cout, 3, 2, 1, r.s));
        return HS_OK;
    }
    int srows = 0;
    if (d.training && fused_bn_stats_enabled() && (srows = gemm_stat_rows(&p)) > 0) p.colstats = (float*)L.bn_ws;
    hs_bn_params bp;
    memset(&bp, 0, sizeof(bp));
    bp.dtype = r.dt; bp.C = cb.Cout; bp.M = Mo;
    bp.partial_rows = srows;
    bp.training = d.training; bp.relu = 1;
    bp.eps = d.eps; bp.momentum = d.momentum;
    bp.x = L.c; bp.y = L.a;
    bp.gamma = cb.gamma; bp.beta = cb.beta;
    bp.running_mean = cb.running_mean; bp.running_var = cb.running_var;
    bp.save_mean = L.mean; bp.save_invstd = L.invstd; bp.scale = L.scale; bp.shift = L.shift;
    bp.ws = L.bn_ws; bp.ws_bytes = L.bn_ws_bytes;
    if (p.colstats && bn_finish_enabled()) {
        const int frows = gemm_bn_finish_rows(&p);
        if (frows > 0 && L.bn_ws_bytes >= (long long)frows * cb.Cout * 3 * 4) {
            p.bn_finish = &bp;
            bp.stats_done = 1;
        }
    }
    CALL(r, gemm_impl(&p, r.s));
    CALLK(r, 8, hs_batchnorm_fwd(&bp, r.s));
    CALL(r, hs_maxpool_fwd(r.dt, L.a, y, L.idx, d.N, L.P, L.Q, cb.Cout, 3, 2, 1, r.s));
    return HS_OK;
}
static int stem_bwd_run(Run& r, const hs_stem_desc& d, const void* y, const void* dy) {
    StemLayout L;
    HS_PROPAGATE(stem_layout(r, d, L));
    const hs_conv_bn& cb = d.cb;
    const int es = esize(r.dt);
    const long long Mo = (long long)d.N * L.P * L.Q;
    void* da = r.ws.alloc(Mo * cb.Cout * es);
    void* dc = r.ws.alloc(Mo * cb.Cout * es);
    float* scratch_pg = (float*)r.ws.alloc(2 * 64 * 4);
    float* dw_packed = (float*)r.ws.alloc((long long)cb.Cout * 224 * 4);
    RUN_CHECK_ARENAS(r, "stem_bwd");
    (void)y;
    CALL(r, hs_maxpool_bwd(r.dt, dy, L.idx, da, d.N, L.P, L.Q, cb.Cout, 3, 2, 1, r.s));
    hs_bn_bwd_params q;
    memset(&q, 0, sizeof(q));
    q.dtype = r.dt; q.C = cb.Cout; q.M = Mo;
    q.training = d.training; q.relu = 1;
    q.dy = da; q.y = L.a; q.x = L.c;
    q.gamma = cb.gamma; q.save_mean = L.mean; q.save_invstd = L.invstd;
    q.dx = cb.dw ? dc : nullptr;
    q.dgamma = cb.dgamma ? cb.dgamma : scratch_pg;
    q.dbeta = cb.dbeta ? cb.dbeta : (scratch_pg ? scratch_pg + 64 : nullptr);
    q.ws = L.bn_ws; q.ws_bytes = L.bn_ws_bytes;
    CALLK(r, 16, hs_batchnorm_bwd(&q, r.s));
    if (cb.dw) {
        hs_gemm_params p = gemm_defaults(r.dt);
        p.a_kind = HS_A_RC; p.b_kind = HS_B_CONV;
        p.M = cb.Cout; p.N = 224; p.K = (int)Mo;
        p.A = dc; p.B = L.packed;
        p.a_elems = Mo * cb.Cout;
        p.b_elems = (long long)d.N * L.Hp * L.Wp * 4;
        p.lda = cb.Cout;
        p.g = L.g;
        p.D = dw_packed; p.ldd = 224; p.out_dtype = HS_F32;
        HS_PROPAGATE(gemm_splitk(r, p));
        CALL(r, hs_unpack_stem_wgrad(dw_packed, cb.dw, cb.Cout, 7, 7, 3, r.s));
    }
    return HS_OK;
}

// ============================================================================================
// BertLayer
// ============================================================================================
struct BertLayout {
    const void *wqkv, *wao, *wi, *wo;   // compute-dtype weights ([3*hidden][hidden] fused QKV)
    float* bqkv;                        // fused f32 bias [3*hidden]
    void *qkv, *ctx, *h1, *x1, *u, *g, *h2;
    float *mean1, *rstd1, *mean2, *rstd2;
    CastList casts;
};
static int bert_layout(Run& r, const hs_bert_layer_desc& d, BertLayout& L) {
    const int Hd = d.hidden, I = d.inter;
    const long long M = (long long)d.B * d.L;
    const int es = esize(r.dt);
    HS_REQUIRE(Hd % d.heads == 0 && d.q.in_f == Hd && d.q.out_f == Hd && d.inter_l.out_f == I && d.out_l.in_f == I,
               "bert_layer: inconsistent dims");
    // fused QKV weight: always a private copy (cast in bf16 mode, plain copy in f32 mode)
    void* wq = r.saved.alloc(3ll * Hd * Hd * es);
    L.wqkv = wq;
    L.bqkv = (float*)r.saved.alloc(3ll * Hd * 4);
    char* shq = r.dt == HS_BF16 ? (char*)shadow_of(d.q.w) : nullptr;
    if (shq && shadow_of(d.k.w) == shq + (long long)Hd * Hd * 2 && shadow_of(d.v.w) == shq + 2ll * Hd * Hd * 2) {
        L.wqkv = shq;                          // the three shadows are the segments of one [3H][H] buffer (hamspine.tower)
    } else if (r.dt == HS_BF16) {
        const float* ws3[3] = {d.q.w, d.k.w, d.v.w};
        for (int i = 0; i < 3; ++i) {
            L.casts.src[L.casts.count] = ws3[i];
            L.casts.dst[L.casts.count] = wq ? (char*)wq + (long long)i * Hd * Hd * 2 : nullptr;
            L.casts.n[L.casts.count] = (long long)Hd * Hd;
            L.casts.count++;
        }
    }
    L.wao = weight_c(r, L.casts, d.ao.w, (long long)Hd * Hd);
    L.wi = weight_c(r, L.casts, d.inter_l.w, (long long)I * Hd);
    L.wo = weight_c(r, L.casts, d.out_l.w, (long long)Hd * I);
    L.qkv = r.saved.alloc(M * 3 * Hd * es);
    L.ctx = r.saved.alloc(M * Hd * es);
    L.h1 = r.saved.alloc(M * Hd * es);
    L.x1 = r.saved.alloc(M * Hd * es);
    L.u = r.saved.alloc(M * I * es);
    L.g = r.saved.alloc(M * I * es);
    L.h2 = r.saved.alloc(M * Hd * es);
    float* st = (float*)r.saved.alloc(M * 4 * 4);
    L.mean1 = st;
    L.rstd1 = st ? st + M : nullptr;
    L.mean2 = st ? st + 2 * M : nullptr;
    L.rstd2 = st ? st + 3 * M : nullptr;
    return HS_OK;
}
static hs_attn_desc bert_attn_desc(const hs_bert_layer_desc& d, int dt) {
    hs_attn_desc a;
    memset(&a, 0, sizeof(a));
    const int Hd = d.hidden;
    a.dtype = dt;
    a.B = d.B; a.H = d.heads; a.Lq = d.L; a.Lk = d.L; a.hd = Hd / d.heads;
    a.q_bs = a.k_bs = a.v_bs = (long long)d.L * 3 * Hd;
    a.q_ld = a.k_ld = a.v_ld = 3 * Hd;
    a.o_bs = (long long)d.L * Hd;
    a.o_ld = Hd;
    a.scale = 1.f / sqrtf((float)a.hd);
    a.dropout_p = d.attn_dropout;
    a.seed = d.seed * 8 + 1;
    a.key_mask = d.attention_mask;
    return a;
}
static int bert_layer_fwd_run(Run& r, const hs_bert_layer_desc& d, const void* x, void* y) {
    BertLayout L;
    HS_PROPAGATE(bert_layout(r, d, L));
    const int Hd = d.hidden, I = d.inter;
    const long long M = (long long)d.B * d.L;
    const int es = esize(r.dt);
    RUN_CHECK_ARENAS(r, "bert_layer_fwd(layout)");
    HS_PROPAGATE(run_casts(r, L.casts));
    if (!r.plan) {
        if (r.dt == HS_F32) {
            const float* ws3[3] = {d.q.w, d.k.w, d.v.w};
            for (int i = 0; i < 3; ++i)
                HS_CHECK_HIP(hipMemcpyAsync((char*)L.wqkv + (long long)i * Hd * Hd * 4, ws3[i], (long long)Hd * Hd * 4,
                                            hipMemcpyDeviceToDevice, r.s));
        }
        hipLaunchKernelGGL(gather3_kernel, dim3(ceil_div(3 * Hd, 256)), dim3(256), 0, r.s, d.q.b, d.k.b, d.v.b, L.bqkv, Hd);
        HS_LAUNCH_CHECK();
    }
    // 1. fused QKV projection
    hs_linear qkv_lin;
    memset(&qkv_lin, 0, sizeof(qkv_lin));
    qkv_lin.in_f = Hd; qkv_lin.out_f = 3 * Hd; qkv_lin.b = L.bqkv;
    HS_PROPAGATE(linear_fwd_run(r, x, M, Hd, qkv_lin, L.wqkv, L.qkv, 3 * Hd, r.dt, HS_ACT_NONE, nullptr, nullptr, 0, 0.f, 0));
    // 2. attention
    hs_attn_desc a = bert_attn_desc(d, r.dt);
    const char* qkv = (const char*)L.qkv;
    HS_PROPAGATE(attention_fwd_run(r, a, qkv, qkv ? qkv + (long long)Hd * es : nullptr,
                                   qkv ? qkv + 2ll * Hd * es : nullptr, L.ctx));
    // 3. attention output: h1 = dropout(ctx Wo^T + b) + x ; x1 = LN(h1)
    HS_PROPAGATE(linear_fwd_run(r, L.ctx, M, Hd, d.ao, L.wao, L.h1, Hd, r.dt, HS_ACT_NONE, nullptr, x, Hd, d.hidden_dropout,
                                d.seed * 8 + 2));
    CALL(r, hs_layernorm_fwd(r.dt, L.h1, d.ln1.gamma, d.ln1.beta, L.x1, L.mean1, L.rstd1, M, Hd, d.ln_eps, r.s));
    // 4. FFN
    HS_PROPAGATE(linear_fwd_run(r, L.x1, M, Hd, d.inter_l, L.wi, L.g, I, r.dt, HS_ACT_GELU, L.u, nullptr, 0, 0.f, 0));
    HS_PROPAGATE(linear_fwd_run(r, L.g, M, I, d.out_l, L.wo, L.h2, Hd, r.dt, HS_ACT_NONE, nullptr, L.x1, Hd,
                                d.hidden_dropout, d.seed * 8 + 3));
    CALL(r, hs_layernorm_fwd(r.dt, L.h2, d.ln2.gamma, d.ln2.beta, y, L.mean2, L.rstd2, M, Hd, d.ln_eps, r.s));
    RUN_CHECK_ARENAS(r, "bert_layer_fwd");
    return HS_OK;
}
static int bert_layer_bwd_run(Run& r, const hs_bert_layer_desc& d, const void* x, const void* dy, void* dx) {
    BertLayout L;
    HS_PROPAGATE(bert_layout(r, d, L));
    const int Hd = d.hidden, I = d.inter;
    const long long M = (long long)d.B * d.L;
    const int es = esize(r.dt);
    hs_attn_desc a = bert_attn_desc(d, r.dt);
    // the attention arenas must be laid out exactly as in the forward: saved first
    // (attention_bwd_run below re-derives P/Pd from r.saved in the same order as attention_fwd_run)
    // separate buffers for the two halves (no reuse): the weight / bias gradient GEMMs run on the side stream
    // and may still be reading a buffer while the main stream moves on with the dgrad chain
    void* dh2 = r.ws.alloc(M * Hd * es);     // gradient wrt h2 (output-LN input)
    void* dd2 = r.ws.alloc(M * Hd * es);     // ... after the dropout mask
    void* dh1 = r.ws.alloc(M * Hd * es);     // gradient wrt h1 (attention-output-LN input)
    void* dd1 = r.ws.alloc(M * Hd * es);
    void* du = r.ws.alloc(M * I * es);       // gradient wrt u
    void* dx1 = r.ws.alloc(M * Hd * es);
    void* dctx = r.ws.alloc(M * Hd * es);
    void* dqkv = r.ws.alloc(M * 3 * Hd * es);
    const long long ln_ws_bytes = hs_layernorm_bwd_ws_bytes(M, Hd);
    float* ln_ws = (float*)r.ws.alloc(ln_ws_bytes);
    float* scratch = (float*)r.ws.alloc(2ll * Hd * 4);
    HS_PROPAGATE(side_setup(r));

    // ---- output LN + FFN ----
    // one pass: LayerNorm backward, the hidden-dropout mask of the dense output and that dense layer's bias gradient
    const bool fuse_ln = fused_ln_bwd_enabled();
    const void* g2 = d.hidden_dropout > 0.f ? dd2 : dh2;
    hs_linear out_l_w = d.out_l;
    if (fuse_ln) {
        CALL(r, hs_layernorm_bwd_pre(r.dt, dy, L.h2, d.ln2.gamma, L.mean2, L.rstd2, dh2, d.ln2.dgamma ? d.ln2.dgamma : scratch,
                                     d.ln2.dbeta ? d.ln2.dbeta : (scratch ? scratch + Hd : nullptr),
                                     d.hidden_dropout > 0.f ? dd2 : nullptr, d.out_l.db, d.hidden_dropout, d.seed * 8 + 3,
                                     ln_ws, ln_ws_bytes, M, Hd, r.s));
        out_l_w.db = nullptr;                  // bias gradient already produced above
    } else {
        CALLK(r, 128, hs_layernorm_bwd(r.dt, dy, L.h2, d.ln2.gamma, L.mean2, L.rstd2, dh2, d.ln2.dgamma ? d.ln2.dgamma : scratch,
                                 d.ln2.dbeta ? d.ln2.dbeta : (scratch ? scratch + Hd : nullptr), ln_ws, ln_ws_bytes, M, Hd, r.s));
        if (d.hidden_dropout > 0.f) CALL(r, hs_dropout(r.dt, dh2, dd2, M * Hd, d.hidden_dropout, d.seed * 8 + 3, r.s));
    }
    // bf16: weight gradients from transposed operands (see linear_wgrad_nt_run); tbuf holds the two transposes of one layer
    const bool nt = r.dt == HS_BF16 && wgrad_nt_enabled() && M % 8 == 0 && Hd % 8 == 0 && I % 8 == 0;
    // grouped: the four GEMMs run together at the end of the layer, so each keeps its own dY^T
    // (only when the four outputs give the chip enough 256x128 tiles: BERT-base 216; a narrow model keeps separate launches)
    const long long big_tiles = (long long)ceil_div(3 * Hd, 256) * ceil_div(Hd, 128) + (long long)ceil_div(Hd, 256) * ceil_div(Hd, 128) +
                                (long long)ceil_div(I, 256) * ceil_div(Hd, 128) + (long long)ceil_div(Hd, 256) * ceil_div(I, 128);
    const bool grouped = nt && grouped_bert_wgrad_enabled() && !overlap_enabled() && big_tiles >= 128;
    char* tA = nt ? (char*)r.ws.alloc(M * (long long)std::max(3 * Hd, I) * 2) : nullptr;   // dY^T (one at a time; grouped: FFN1's)
    char* tA_ffn2 = grouped ? (char*)r.ws.alloc(M * (long long)Hd * 2) : tA;
    char* tA_ao = grouped ? (char*)r.ws.alloc(M * (long long)Hd * 2) : tA;
    char* tA_qkv = grouped ? (char*)r.ws.alloc(M * 3ll * Hd * 2) : tA;
    r.group_nt = grouped;
    if (grouped && !r.plan) {
        // one cached problem table per layer, keyed by the layer's WEIGHT pointer: stable across steps (a gradient pointer is
        // not -- models that allocate fresh gradient memory per backward would add a table per address -- and is NULL for a
        // frozen weight, which would collide with the image tower's slots 0 / 1); + 16 keeps it clear of slots -1, 0, 1
        r.grp_nt = r.tower_grp ? r.tower_grp : gemm_group_open(r.s, (long long)(uintptr_t)d.q.w + 16);
        HS_REQUIRE(r.grp_nt != nullptr, "bert_layer_bwd: cannot set up the grouped weight-gradient launch");
    }
    // X^T of the four saved activations the weight gradients read: all known when the layer's backward starts, so they are
    // transposed by ONE launch (together with the weight copies below) instead of one launch in front of each GEMM
    char* tX_g = nt ? (char*)r.ws.alloc(M * (long long)I * 2) : nullptr;
    char* tX_x1 = nt ? (char*)r.ws.alloc(M * (long long)Hd * 2) : nullptr;
    char* tX_ctx = nt ? (char*)r.ws.alloc(M * (long long)Hd * 2) : nullptr;
    char* tX_x = nt ? (char*)r.ws.alloc(M * (long long)Hd * 2) : nullptr;
    // bias gradient of a layer whose weight gradient is split (no fused row sums): column sums of the row-major dy
    auto bias_by_colsum = [&](const hs_linear& lin, const void* dy_rm, int ldy) -> int {
        hs_linear only_b = lin;
        only_b.dw = nullptr;
        return linear_wgrad_run(r, nullptr, M, 0, only_b, dy_rm, ldy);
    };
    auto wgrad = [&](const void* x_rm, const void* xT, int in_f, const hs_linear& lin, const void* dy_rm, int ldy, char* tAk) -> int {
        if (!nt || !lin.dw) {
            const bool keep = r.group_nt;
            r.group_nt = false;                                  // (bias-only / f32 paths are not grouped)
            const int st = linear_wgrad_run(r, x_rm, M, in_f, lin, dy_rm, ldy);
            r.group_nt = keep;
            return st;
        }
        HS_PROPAGATE(transpose_run(r, dy_rm, tAk, M, lin.out_f, ldy, true));
        const bool fused_b = lin.db && (r.group_nt || hs_gemm_suggest_split(lin.out_f, in_f, (int)M, r.dt) <= 1) && fused_bias_grad_enabled();
        HS_PROPAGATE(linear_wgrad_nt_run(r, xT, tAk, M, in_f, lin.out_f, lin.dw, lin.db, 0, nullptr, nullptr));
        if (lin.db && !fused_b) HS_PROPAGATE(bias_by_colsum(lin, dy_rm, ldy));
        return HS_OK;
    };
    // bf16: data gradients of the three wide Linears on transposed weight copies (K-contiguous B operand)
    const bool dnt = r.dt == HS_BF16 && dgrad_nt_enabled() && Hd % 8 == 0 && I % 8 == 0;
    char* wo_t = dnt ? (char*)r.ws.alloc((long long)Hd * I * 2) : nullptr;
    char* wi_t = dnt ? (char*)r.ws.alloc((long long)Hd * I * 2) : nullptr;
    char* wqkv_t = dnt ? (char*)r.ws.alloc(3ll * Hd * Hd * 2) : nullptr;
    {
        const void* src[7];
        void* dst[7];
        int32_t R[7], Cc[7];
        int64_t lds[7], ldd[7];
        int n = 0;
        auto add = [&](const void* s_, void* d_, long long rows, int cols) {
            src[n] = s_; dst[n] = d_; R[n] = (int32_t)rows; Cc[n] = cols; lds[n] = cols; ldd[n] = rows;
            ++n;
        };
        if (nt) {
            add(L.g, tX_g, M, I);
            add(L.x1, tX_x1, M, Hd);
            add(L.ctx, tX_ctx, M, Hd);
            add(x, tX_x, M, Hd);
        }
        if (dnt) {
            add(L.wo, wo_t, Hd, I);              // W [out = Hd][in = I] -> [I][Hd]
            add(L.wi, wi_t, I, Hd);
            add(L.wqkv, wqkv_t, 3 * Hd, Hd);
        }
        if (n > 0) CALLK(r, 4, hs_transpose_bf16_multi(n, src, dst, R, Cc, lds, ldd, r.s));
    }
    HS_PROPAGATE(on_side(r, [&]() { return wgrad(L.g, tX_g, I, out_l_w, g2, Hd, tA_ffn2); }));
    HS_PROPAGATE(linear_dgrad_run(r, d.out_l, L.wo, g2, M, Hd, du, I, r.dt, HS_MUL_GELU_GRAD, L.u, I, nullptr, wo_t));
    HS_PROPAGATE(on_side(r, [&]() { return wgrad(L.x1, tX_x1, Hd, d.inter_l, du, I, tA); }));
    // dx1 = du Wi + dh2 (residual into x1)
    HS_PROPAGATE(linear_dgrad_run(r, d.inter_l, L.wi, du, M, I, dx1, Hd, r.dt, HS_MUL_NONE, nullptr, 0, dh2, wi_t));
    // ---- attention output LN + dense ----
    const void* g1 = d.hidden_dropout > 0.f ? dd1 : dh1;
    hs_linear ao_w = d.ao;
    if (fuse_ln) {
        CALL(r, hs_layernorm_bwd_pre(r.dt, dx1, L.h1, d.ln1.gamma, L.mean1, L.rstd1, dh1, d.ln1.dgamma ? d.ln1.dgamma : scratch,
                                     d.ln1.dbeta ? d.ln1.dbeta : (scratch ? scratch + Hd : nullptr),
                                     d.hidden_dropout > 0.f ? dd1 : nullptr, d.ao.db, d.hidden_dropout, d.seed * 8 + 2,
                                     ln_ws, ln_ws_bytes, M, Hd, r.s));
        ao_w.db = nullptr;
    } else {
        CALLK(r, 128, hs_layernorm_bwd(r.dt, dx1, L.h1, d.ln1.gamma, L.mean1, L.rstd1, dh1, d.ln1.dgamma ? d.ln1.dgamma : scratch,
                                 d.ln1.dbeta ? d.ln1.dbeta : (scratch ? scratch + Hd : nullptr), ln_ws, ln_ws_bytes, M, Hd, r.s));
        if (d.hidden_dropout > 0.f) CALL(r, hs_dropout(r.dt, dh1, dd1, M * Hd, d.hidden_dropout, d.seed * 8 + 2, r.s));
    }
    HS_PROPAGATE(on_side(r, [&]() { return wgrad(L.ctx, tX_ctx, Hd, ao_w, g1, Hd, tA_ao); }));
    HS_PROPAGATE(linear_dgrad_run(r, d.ao, L.wao, g1, M, Hd, dctx, Hd, r.dt, HS_MUL_NONE, nullptr, 0, nullptr));
    // ---- attention core ----
    const char* qkv = (const char*)L.qkv;
    char* dq = (char*)dqkv;
    HS_PROPAGATE(attention_bwd_run(r, a, qkv, qkv ? qkv + (long long)Hd * es : nullptr, qkv ? qkv + 2ll * Hd * es : nullptr,
                                   dctx, dq, dq ? dq + (long long)Hd * es : nullptr, dq ? dq + 2ll * Hd * es : nullptr));
    // ---- QKV projection ----
    HS_PROPAGATE(on_side(r, [&]() -> int {
        // one GEMM for the three weight gradients: [dWq; dWk; dWv] = dqkv^T x, rows routed to the three tensors
        const bool fused = d.q.dw && d.k.dw && d.v.dw;
        if (fused && nt) {
            HS_PROPAGATE(transpose_run(r, dqkv, tA_qkv, M, 3 * Hd, 3 * Hd, true));
            const bool bias_too = d.q.db && d.k.db && d.v.db && (r.group_nt || hs_gemm_suggest_split(3 * Hd, Hd, (int)M, r.dt) <= 1) &&
                                  fused_bias_grad_enabled();
            float* dws[2] = {d.k.dw, d.v.dw};
            float* dbs[2] = {d.k.db, d.v.db};
            HS_PROPAGATE(linear_wgrad_nt_run(r, tX_x, tA_qkv, M, Hd, 3 * Hd, d.q.dw, bias_too ? d.q.db : nullptr, Hd, dws, dbs));
            if (bias_too) return HS_OK;
            const hs_linear* lins[3] = {&d.q, &d.k, &d.v};
            for (int i = 0; i < 3; ++i) HS_PROPAGATE(bias_by_colsum(*lins[i], dq ? dq + (long long)i * Hd * es : nullptr, 3 * Hd));
            return HS_OK;
        }
        if (fused) {
            hs_gemm_params p = gemm_defaults(r.dt);
            p.a_kind = HS_A_RC; p.b_kind = HS_B_RC;
            p.M = 3 * Hd; p.N = Hd; p.K = (int)M;
            p.A = dqkv; p.B = x;
            p.a_elems = M * 3 * Hd;
            p.b_elems = M * Hd;
            p.lda = 3 * Hd; p.ldb = Hd;
            p.D = d.q.dw; p.ldd = Hd; p.out_dtype = HS_F32;
            p.seg_rows = Hd;
            p.D_seg[0] = d.k.dw;
            p.D_seg[1] = d.v.dw;
            const bool bias_too = d.q.db && d.k.db && d.v.db && bias_in_wgrad(r.dt, 3 * Hd, Hd, M);
            if (bias_too) {
                p.rowsum_a = d.q.db;
                p.rowsum_seg[0] = d.k.db;
                p.rowsum_seg[1] = d.v.db;
            }
            HS_PROPAGATE(gemm_splitk(r, p));
            if (bias_too) return HS_OK;
        }
        const hs_linear* lins[3] = {&d.q, &d.k, &d.v};
        for (int i = 0; i < 3; ++i) {
            hs_linear li = *lins[i];
            if (fused) li.dw = nullptr;   // bias gradients only
            HS_PROPAGATE(linear_wgrad_run(r, x, M, Hd, li, dq ? dq + (long long)i * Hd * es : nullptr, 3 * Hd));
        }
        return HS_OK;
    }));
    if (dx) {
        hs_linear qkv_lin;
        memset(&qkv_lin, 0, sizeof(qkv_lin));
        qkv_lin.in_f = Hd; qkv_lin.out_f = 3 * Hd;
        // dx = dqkv Wqkv + dh1 (residual of the attention-output LN input)
        HS_PROPAGATE(linear_dgrad_run(r, qkv_lin, L.wqkv, dqkv, M, 3 * Hd, dx, Hd, r.dt, HS_MUL_NONE, nullptr, 0, dh1, wqkv_t));
    }
    if (r.group_nt && !r.plan && !r.tower_grp) HS_PROPAGATE(gemm_group_flush(r.grp_nt, r.s));
    r.group_nt = false;
    HS_PROPAGATE(side_join(r));
    RUN_CHECK_ARENAS(r, "bert_layer_bwd");
    return HS_OK;
}


// ============================================================================================
// whole-tower executors: ONE call runs the stem and every residual block (or the embeddings and every BertLayer) of a
// tower, forward or backward.  Same kernels and the same per-composite layout code as the per-block entry points above;
// what goes away is the host cost between them (a C2 step made ~70 Python -> C calls with ~3 layout passes, a descriptor
// rebuild and several tensor allocations each: 11 ms of host time for 12.6 ms of GPU time).
// Saved arena of a tower = for every composite, in forward order: [its output activation][its own saved layout].
// ============================================================================================
static long long out_bytes_of(const hs_resblock_desc& d, int dt, int* Cout, int* Ho, int* Wo) {
    int H = d.H, W = d.W;
    for (int i = 0; i < d.n_main; ++i) {
        const hs_conv_bn& c = d.main[i];
        H = (H + 2 * c.pad - c.R) / c.stride + 1;
        W = (W + 2 * c.pad - c.R) / c.stride + 1;
    }
    const int C = d.main[d.n_main - 1].Cout;
    if (Cout) *Cout = C;
    if (Ho) *Ho = H;
    if (Wo) *Wo = W;
    return (long long)d.N * H * W * C * esize(dt);
}
struct ResnetLayout {
    long long y_off[HS_RESNET_MAX_BLOCKS + 1];       // offset of the output of the stem (0) / of block i (i + 1) in `saved`
    long long lay_off[HS_RESNET_MAX_BLOCKS + 1];     // offset at which that composite's own layout starts
    long long y_bytes[HS_RESNET_MAX_BLOCKS + 1];
    long long max_act = 0;
};
static int resnet_check(const hs_resnet_desc& d) {
    HS_REQUIRE(d.n_blocks >= 1 && d.n_blocks <= HS_RESNET_MAX_BLOCKS, "resnet: n_blocks %d out of range", d.n_blocks);
    HS_REQUIRE(d.n_taps >= 1 && d.n_taps <= HS_RESNET_MAX_TAPS, "resnet: n_taps %d out of range", d.n_taps);
    for (int t = 0; t < d.n_taps; ++t)
        HS_REQUIRE(d.tap_block[t] >= 0 && d.tap_block[t] < d.n_blocks && (t == 0 || d.tap_block[t] > d.tap_block[t - 1]),
                   "resnet: tap blocks must be ascending block indices");
    HS_REQUIRE(d.tap_block[d.n_taps - 1] == d.n_blocks - 1, "resnet: the last tap must be the last block");
    for (int i = 0; i < d.n_blocks; ++i)
        HS_REQUIRE(d.blocks[i].dtype == d.stem.dtype && d.blocks[i].N == d.stem.N, "resnet: block %d dtype / batch mismatch", i);
    return HS_OK;
}
// forward pass over the tower; in plan mode it only lays the arenas out (and fills `lo`)
static int resnet_fwd_run(Run& r, const hs_resnet_desc& d, const float* image, ResnetLayout& lo) {
    const int dt = d.stem.dtype;
    const int es = esize(dt);
    const int P = (d.stem.H + 6 - 7) / 2 + 1, Q = (d.stem.W + 6 - 7) / 2 + 1;
    const int P2 = (P + 2 - 3) / 2 + 1, Q2 = (Q + 2 - 3) / 2 + 1;
    lo.y_off[0] = r.saved.mark();
    lo.y_bytes[0] = (long long)d.stem.N * P2 * Q2 * d.stem.cb.Cout * es;
    lo.max_act = lo.y_bytes[0];
    char* x = (char*)r.saved.alloc(lo.y_bytes[0]);
    lo.lay_off[0] = r.saved.mark();
    long long wm = r.ws.mark();
    HS_PROPAGATE(stem_fwd_run(r, d.stem, image, x));
    r.ws.release(wm);
    HS_REQUIRE(d.blocks[0].H == P2 && d.blocks[0].W == Q2 && d.blocks[0].main[0].Cin == d.stem.cb.Cout,
               "resnet: block 0 does not take the stem's output shape");
    int C = d.stem.cb.Cout, H = P2, W = Q2;
    for (int i = 0; i < d.n_blocks; ++i) {
        const hs_resblock_desc& b = d.blocks[i];
        HS_REQUIRE(b.H == H && b.W == W && b.main[0].Cin == C, "resnet: block %d input shape mismatch", i);
        lo.y_off[i + 1] = r.saved.mark();
        lo.y_bytes[i + 1] = out_bytes_of(b, dt, &C, &H, &W);
        lo.max_act = std::max(lo.max_act, lo.y_bytes[i + 1]);
        char* y = (char*)r.saved.alloc(lo.y_bytes[i + 1]);
        lo.lay_off[i + 1] = r.saved.mark();
        wm = r.ws.mark();
        HS_PROPAGATE(resblock_fwd_run(r, b, x, y));
        r.ws.release(wm);
        x = y;
    }
    return HS_OK;
}
static int resnet_bwd_run(Run& r, const hs_resnet_desc& d, const void* const* dy_taps, const ResnetLayout& lo) {
    const int dt = d.stem.dtype;
    // two gradient buffers walk the chain: block i reads the gradient of its output from one and writes the gradient
    // of its input to the other
    char* gbuf[2] = {(char*)r.ws.alloc(lo.max_act), (char*)r.ws.alloc(lo.max_act)};
    int cur = 0;
    const void* dy = nullptr;
    int tap = d.n_taps - 1;
    // bf16: the blocks' weight-gradient GEMMs are not launched where they occur but collected and run as two grouped grids
    // (1x1 / spatial filters) behind the data-gradient chain; the blocks' scratch is then kept to the end (HAMSPINE_GROUPED_WGRAD=0:
    // one launch per convolution, scratch released per block)
    r.defer_wgrad = dt == HS_BF16 && grouped_wgrad_enabled() && !overlap_enabled();
    static const bool staged_on = [] { const char* e = getenv("HAMSPINE_STAGED_WGRAD"); return !(e && e[0] == '0'); }();
    const bool staged = r.defer_wgrad && !r.plan && g_ms.n > 0 && staged_on;
    int seg = 0;
    std::vector<const void*> seg_ptrs;
    if (r.defer_wgrad && !r.plan) {
        r.grp_pw = gemm_group_open(r.s, staged ? 8 : 0);
        r.grp_conv = gemm_group_open(r.s, staged ? 9 : 1);
        HS_REQUIRE(r.grp_pw && r.grp_conv, "resnet_bwd: cannot set up the grouped weight-gradient launches");
    }
    // Block-final BatchNorm sums across the block boundary: block i + 1's stage-0 data-gradient GEMM writes the gradient of
    // block i's output, so it also takes the sums of block i's last BatchNorm (two buffers, alternating; HAMSPINE_FUSED_BN_BWD=0
    // or an intermediate tap whose gradient joins afterwards switch it off for that block).
    long long xs_bytes = 0;
    for (int i = 0; i < d.n_blocks; ++i) {
        int C, H, W;
        const long long ob = out_bytes_of(d.blocks[i], dt, &C, &H, &W);
        const long long rows = (ob / esize(dt) / C + 63) / 64;
        xs_bytes = std::max<long long>(xs_bytes, ((rows + rows / 32 + 2) * C * 2 + 4ll * C) * 4);   // tile rows + merge rows + coefficients
    }
    float* xs_buf[2] = {(float*)r.ws.alloc(xs_bytes), (float*)r.ws.alloc(xs_bytes)};
    OwnSums own_next;                 // sums of the block about to be processed, if the previous iteration produced them
    const char* xb_env = getenv("HAMSPINE_XBLOCK_BN");       // read per call: tests compare both settings in one process
    const int xblock = xb_env ? atoi(xb_env) : 1;
    int xs_rows = 0, xs_done = 0;
    hs_bn_bwd_params xs_fin;
    for (int i = d.n_blocks - 1; i >= 0; --i) {
        const bool is_tap = tap >= 0 && d.tap_block[tap] == i;
        const void* ext = is_tap && dy_taps ? dy_taps[tap] : nullptr;
        if (is_tap) --tap;
        if (i == d.n_blocks - 1) {
            HS_REQUIRE(r.plan || ext, "resnet_bwd: the gradient of the last block's output is missing");
            dy = ext;
        } else if (ext) {      // an intermediate tap (multi-scale tokens): its gradient joins the one flowing down the chain
            CALL(r, hs_axpby(dt, dt, dy, ext, gbuf[cur], lo.y_bytes[i + 1] / esize(dt), 1.f, 1.f, r.s));
            dy = gbuf[cur];
        }
        const char* x = r.saved.base ? r.saved.base + lo.y_off[i] : nullptr;
        const char* y = r.saved.base ? r.saved.base + lo.y_off[i + 1] : nullptr;
        char* dx = gbuf[cur ^ 1];
        // sums for block i - 1's final BatchNorm ride on this block's stage-0 data gradient when that gradient IS the whole
        // gradient of block i - 1's output (no tap gradient joins it later) and block i - 1 is a three-stage block
        BnSums next_bn;
        const BnSums* nb = nullptr;
        const bool prev_tap = i > 0 && tap >= 0 && d.tap_block[tap] == i - 1 && dy_taps && dy_taps[tap];
        // MEASURED (round 3, rocprofv3, un-overlapped C2 step, rider operands fetched in front of the K walk): the rider reads
        // three more tensors (c3, y, identity gradient).  Where the block input is small (64x64-tile launches, M <= 6272 at
        // C2) the launch takes 29 us against 18 + 23 us for GEMM + bn_bwd_partial; on the big early layers (128x64 tiles,
        // M >= 25088) it is bandwidth-bound inside a tiled GEMM (205 MB at 3.4 TB/s: 61 us against 30 + 23) -> size gate.
        // HAMSPINE_XBLOCK_BN=0 turns it off, =2 takes it on every eligible block.
        const bool xb_size = xblock >= 2 || lo.y_bytes[i] <= (16ll << 20);
        if (xblock > 0 && xb_size && i > 0 && dt == HS_BF16 && !prev_tap && !r.plan && d.blocks[i].main[0].stride == 1 && fused_bn_bwd_enabled()) {
            const hs_resblock_desc& pb = d.blocks[i - 1];
            const long long keep_saved = r.saved.off, keep_ws = r.ws.mark();
            r.saved.off = lo.lay_off[i];
            ResLayout PL;
            HS_PROPAGATE(res_layout(r, pb, PL));
            const StageBuf& lb = PL.main[pb.n_main - 1];
            r.saved.off = keep_saved;
            r.ws.release(keep_ws);
            xs_rows = 0;
            xs_done = 0;
            next_bn = BnSums{lb.c, lb.scale, lb.shift, lb.mean, lb.invstd, xs_buf[i & 1], xs_bytes, &xs_rows, x};
            const hs_conv_bn& pcb = pb.main[pb.n_main - 1];
            memset(&xs_fin, 0, sizeof(xs_fin));
            xs_fin.dtype = dt; xs_fin.C = pcb.Cout; xs_fin.M = (long long)pb.N * lb.s.P * lb.s.Q; xs_fin.training = pb.training;
            xs_fin.gamma = pcb.gamma; xs_fin.dgamma = pcb.dgamma; xs_fin.dbeta = pcb.dbeta;
            if (pcb.dgamma && pcb.dbeta) {
                next_bn.finish = &xs_fin;
                next_bn.done = &xs_done;
            }
            nb = &next_bn;
        }
        r.saved.off = lo.lay_off[i + 1];
        const long long wm = r.ws.mark();
        HS_PROPAGATE(resblock_bwd_run(r, d.blocks[i], x, y, dy, dx, nb, own_next.rows > 0 ? &own_next : nullptr));
        own_next = OwnSums{};
        if (nb && xs_rows > 0) own_next = OwnSums{xs_buf[i & 1], xs_bytes, xs_rows, xs_done};
        if (!r.defer_wgrad) r.ws.release(wm);
        // Under a data-parallel wrapper (gradient milestones registered) the deferred weight gradients leave per STAGE: when the
        // first block of a stage (the one with the downsample branch) is done, the stage's two grouped grids run and the
        // events of the buckets that end in this stage are recorded -- layer4 holds 60 % of a ResNet50's parameters, and its
        // exchange then has the rest of the backward to hide in instead of starting behind the tower's last kernel.
        // Without a wrapper (N = 1) the two grids at the end stay as they are.
        if (staged) {
            const hs_resblock_desc& bd = d.blocks[i];
            for (int k = 0; k < bd.n_main; ++k) {
                seg_ptrs.push_back(bd.main[k].dw); seg_ptrs.push_back(bd.main[k].dgamma); seg_ptrs.push_back(bd.main[k].dbeta);
            }
            if (bd.has_ds) {
                seg_ptrs.push_back(bd.ds.dw); seg_ptrs.push_back(bd.ds.dgamma); seg_ptrs.push_back(bd.ds.dbeta);
            }
            if (bd.has_ds && i > 0) {
                HS_PROPAGATE(gemm_group_flush(r.grp_pw, r.s));
                HS_PROPAGATE(gemm_group_flush(r.grp_conv, r.s));
                HS_PROPAGATE(milestones_hit(r, seg_ptrs.data(), (int)seg_ptrs.size()));
                seg_ptrs.clear();
                ++seg;
                r.grp_pw = gemm_group_open(r.s, 8 + 2 * seg);
                r.grp_conv = gemm_group_open(r.s, 9 + 2 * seg);
                HS_REQUIRE(r.grp_pw && r.grp_conv, "resnet_bwd: cannot set up the grouped weight-gradient launches");
            }
        }
        dy = dx;
        cur ^= 1;
    }
    if (r.defer_wgrad && !r.plan) {
        HS_PROPAGATE(gemm_group_flush(r.grp_pw, r.s));
        HS_PROPAGATE(gemm_group_flush(r.grp_conv, r.s));
    }
    r.defer_wgrad = false;
    const hs_conv_bn& cb = d.stem.cb;
    if (cb.dw || cb.dgamma || cb.dbeta) {
        r.saved.off = lo.lay_off[0];
        const long long wm = r.ws.mark();
        HS_PROPAGATE(stem_bwd_run(r, d.stem, r.saved.base ? r.saved.base + lo.y_off[0] : nullptr, dy));
        r.ws.release(wm);
    }
    return milestones_flush(r);
}

// ---- BERT ----------------------------------------------------------------------------------
struct BertTowerLayout {
    long long ssum_off, stats_off;
    long long y_off[HS_BERT_MAX_LAYERS + 1], lay_off[HS_BERT_MAX_LAYERS + 1];
    long long act_bytes;
};
static int bert_check(const hs_bert_desc& d) {
    HS_REQUIRE(d.n_layers >= 0 && d.n_layers <= HS_BERT_MAX_LAYERS, "bert: n_layers %d out of range", d.n_layers);
    HS_REQUIRE(d.B > 0 && d.L > 0 && d.hidden > 0 && d.vocab > 0, "bert: bad dims");
    HS_REQUIRE(d.word && d.pos && d.type0 && d.gamma && d.beta, "bert: null embedding parameter");
    for (int i = 0; i < d.n_layers; ++i)
        HS_REQUIRE(d.layers[i].dtype == d.dtype && d.layers[i].B == d.B && d.layers[i].L == d.L && d.layers[i].hidden == d.hidden,
                   "bert: layer %d shape mismatch", i);
    return HS_OK;
}
static hs_bert_layer_desc bert_layer_of(const hs_bert_desc& d, int i, const int64_t* mask) {
    hs_bert_layer_desc l = d.layers[i];
    l.attention_mask = mask;
    l.seed = d.seed + 16ull * (unsigned)(i + 1);
    return l;
}
static int bert_fwd_run(Run& r, const hs_bert_desc& d, const int64_t* ids, const int64_t* mask, BertTowerLayout& lo) {
    const long long M = (long long)d.B * d.L;
    const int es = esize(d.dtype);
    lo.act_bytes = M * d.hidden * es;
    lo.ssum_off = r.saved.mark();
    void* ssum = r.saved.alloc(lo.act_bytes);
    lo.stats_off = r.saved.mark();
    float* stats = (float*)r.saved.alloc(2 * M * 4);
    lo.y_off[0] = r.saved.mark();
    char* x = (char*)r.saved.alloc(lo.act_bytes);
    lo.lay_off[0] = r.saved.mark();
    CALL(r, hs_bert_embed_fwd(d.dtype, ids, d.word, d.pos, d.type0, d.gamma, d.beta, ssum, x, stats, stats + M, M, d.L, d.hidden,
                              d.vocab, d.ln_eps, d.embed_dropout, d.seed, r.s));
    for (int i = 0; i < d.n_layers; ++i) {
        lo.y_off[i + 1] = r.saved.mark();
        char* y = (char*)r.saved.alloc(lo.act_bytes);
        lo.lay_off[i + 1] = r.saved.mark();
        const hs_bert_layer_desc l = bert_layer_of(d, i, mask);
        const long long wm = r.ws.mark();
        HS_PROPAGATE(bert_layer_fwd_run(r, l, x, y));
        r.ws.release(wm);
        x = y;
    }
    return HS_OK;
}
static int bert_bwd_run(Run& r, const hs_bert_desc& d, const int64_t* ids, const int64_t* mask, const void* dy_in,
                        const BertTowerLayout& lo) {
    const long long M = (long long)d.B * d.L;
    const int Hd = d.hidden;
    char* gbuf[2] = {(char*)r.ws.alloc(lo.act_bytes), (char*)r.ws.alloc(lo.act_bytes)};
    int cur = 0;
    const void* dy = dy_in;
    char* base = r.saved.base;
    // The K-contiguous weight-gradient GEMMs of `span` consecutive layers run as ONE grouped grid (BERT-base, two layers: 8
    // GEMMs = 216 tiles of 256 x 256, one per CU, K = 4096: the phase-pipelined body's steady state -- alone a layer is 108
    // such tiles, or 216 of 256 x 128 on the generic body at half the rate).  The layers' transposed operands stay allocated
    // until the flush; HAMSPINE_WGRAD_LAYERS=1 flushes per layer (round 2 behaviour).
    static const int span = [] { const char* e = getenv("HAMSPINE_WGRAD_LAYERS"); const int v = e ? atoi(e) : 2; return v >= 1 && v <= 8 ? v : 2; }();
    int pending = 0;
    long long wm = 0;
    const void* gps[8][16];
    struct HookGuard {          // a weight-gradient GEMM that gemm_group_add launches on its own must find its dY^T made
        explicit HookGuard(Run* rr) { gemm_group_set_immediate_hook([](void* c) -> int { return pending_transposes_flush(*(Run*)c); }, rr); }
        ~HookGuard() { gemm_group_set_immediate_hook(nullptr, nullptr); }
    } hook_guard(&r);
    for (int i = d.n_layers - 1; i >= 0; --i) {
        const hs_bert_layer_desc l = bert_layer_of(d, i, mask);
        char* dx = gbuf[cur];
        r.saved.off = lo.lay_off[i + 1];
        if (pending == 0) {
            wm = r.ws.mark();
            r.tower_grp = (span > 1 && !r.plan) ? gemm_group_open(r.s, (long long)(uintptr_t)l.q.w + 24) : nullptr;
        }
        HS_PROPAGATE(bert_layer_bwd_run(r, l, base ? base + lo.y_off[i] : nullptr, dy, dx));
        {
            const void* gp[16] = {l.q.dw, l.q.db, l.k.dw, l.k.db, l.v.dw, l.v.db, l.ao.dw, l.ao.db, l.ln1.dgamma, l.ln1.dbeta,
                                  l.inter_l.dw, l.inter_l.db, l.out_l.dw, l.out_l.db, l.ln2.dgamma, l.ln2.dbeta};
            memcpy(gps[pending], gp, sizeof(gp));
        }
        ++pending;
        if (pending == span || i == 0) {
            HS_PROPAGATE(pending_transposes_flush(r));
            if (r.tower_grp) HS_PROPAGATE(gemm_group_flush(r.tower_grp, r.s));
            r.tower_grp = nullptr;
            r.ws.release(wm);
            for (int k = 0; k < pending; ++k) HS_PROPAGATE(milestones_hit(r, gps[k], 16));
            pending = 0;
        }
        dy = dx;
        cur ^= 1;
    }
    // embeddings: dropout mask, LayerNorm backward, scatter into the tables (BertEmbeddings with token_type_ids = 0)
    const bool any = d.dword || d.dpos || d.dtype0 || d.dgamma || d.dbeta;
    if (!any) return milestones_flush(r);
    const void* g = dy;
    if (d.embed_dropout > 0.f) {
        char* gd = gbuf[cur];
        CALL(r, hs_dropout(d.dtype, dy, gd, M * Hd, d.embed_dropout, d.seed, r.s));
        g = gd;
        cur ^= 1;
    }
    char* dsum = gbuf[cur];
    const long long lnb = hs_layernorm_bwd_ws_bytes(M, Hd);
    const long long csb = hs_colsum_ws_bytes(M, Hd);
    void* lnws = r.ws.alloc(lnb);
    void* csws = r.ws.alloc(csb);
    float* scratch = (float*)r.ws.alloc(2ll * Hd * 4);
    const float* stats = base ? (const float*)(base + lo.stats_off) : nullptr;
    CALLK(r, 128, hs_layernorm_bwd(d.dtype, g, base ? base + lo.ssum_off : nullptr, d.gamma, stats, stats ? stats + M : nullptr, dsum,
                             d.dgamma ? d.dgamma : scratch, d.dbeta ? d.dbeta : (scratch ? scratch + Hd : nullptr), lnws, lnb, M,
                             Hd, r.s));
    if (!r.plan) {
        if (d.dword) HS_CHECK_HIP(hipMemsetAsync(d.dword, 0, (size_t)d.vocab * Hd * 4, r.s));
        if (d.dpos) HS_CHECK_HIP(hipMemsetAsync(d.dpos, 0, (size_t)d.max_pos * Hd * 4, r.s));
    }
    if (d.dword || d.dpos) CALL(r, hs_bert_embed_bwd(d.dtype, ids, dsum, d.dword, d.dpos, d.B, d.L, Hd, d.vocab, d.pad_id, r.s));
    if (d.dtype0) {
        if (!r.plan && d.n_types > 1) HS_CHECK_HIP(hipMemsetAsync(d.dtype0 + Hd, 0, (size_t)(d.n_types - 1) * Hd * 4, r.s));
        CALL(r, hs_colsum(d.dtype, dsum, M, Hd, Hd, d.dtype0, csws, csb, 0, r.s));
    }
    return milestones_flush(r);
}

}  // namespace hs

using namespace hs;

// run the layout in plan mode first and refuse to launch anything when an arena is too small
#define PLAN_CHECK(what, dt, svb, wsb, RUNCALL)                                                        \
    do {                                                                                               \
        Run pl;                                                                                        \
        run_init(pl, dt, true, nullptr, 0, nullptr, 0, nullptr);                                       \
        Run& r = pl;                                                                                   \
        HS_PROPAGATE(RUNCALL);                                                                         \
        HS_REQUIRE(pl.saved.peak <= (svb) && pl.ws.peak <= (wsb),                                      \
                   "%s: arena too small (saved %lld needed / %lld given, ws %lld needed / %lld given)", \
                   what, pl.saved.peak, (long long)(svb), pl.ws.peak, (long long)(wsb));               \
    } while (0)

#define QUERY_BODY(DESC_DT, RUNCALL)                                   \
    Run r;                                                             \
    run_init(r, DESC_DT, true, nullptr, 0, nullptr, 0, nullptr);       \
    int st = RUNCALL;                                                  \
    if (st != HS_OK) return st;                                        \
    if (saved_bytes) *saved_bytes = r.saved.peak;                      \
    if (ws_bytes) *ws_bytes = r.ws.peak;                               \
    return HS_OK;

extern "C" {

hs_status hs_attention_query(const hs_attn_desc* d, int64_t* saved_bytes, int64_t* ws_bytes) {
    HS_REQUIRE(d, "attention_query: null desc");
    // the backward needs the larger workspace; report max(fwd, bwd)
    Run r;
    run_init(r, d->dtype, true, nullptr, 0, nullptr, 0, nullptr);
    HS_PROPAGATE(attention_bwd_run(r, *d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
    if (saved_bytes) *saved_bytes = r.saved.peak;
    if (ws_bytes) *ws_bytes = r.ws.peak;
    return HS_OK;
}
hs_status hs_attention_fwd(const hs_attn_desc* d, const void* q, const void* k, const void* v, void* o, void* saved,
                           int64_t saved_bytes, void* ws, int64_t ws_bytes, void* stream) {
    HS_REQUIRE(d && q && k && v && o && saved && ws, "attention_fwd: null argument");
    PLAN_CHECK("attention_fwd", d->dtype, saved_bytes, ws_bytes, attention_fwd_run(r, *d, q, k, v, o));
    Run r;
    run_init(r, d->dtype, false, saved, saved_bytes, ws, ws_bytes, (hipStream_t)stream);
    return attention_fwd_run(r, *d, q, k, v, o);
}
hs_status hs_attention_bwd(const hs_attn_desc* d, const void* q, const void* k, const void* v, const void* d_o, void* dq,
                           void* dk, void* dv, void* saved, int64_t saved_bytes, void* ws, int64_t ws_bytes,
                           void* stream) {
    HS_REQUIRE(d && q && k && v && d_o && dq && dk && dv && saved && ws, "attention_bwd: null argument");
    PLAN_CHECK("attention_bwd", d->dtype, saved_bytes, ws_bytes, attention_bwd_run(r, *d, q, k, v, d_o, dq, dk, dv));
    Run r;
    run_init(r, d->dtype, false, saved, saved_bytes, ws, ws_bytes, (hipStream_t)stream);
    return attention_bwd_run(r, *d, q, k, v, d_o, dq, dk, dv);
}

hs_status hs_resblock_query(const hs_resblock_desc* d, int64_t* saved_bytes, int64_t* ws_bytes) {
    HS_REQUIRE(d, "resblock_query: null desc");
    Run r;
    run_init(r, d->dtype, true, nullptr, 0, nullptr, 0, nullptr);
    HS_PROPAGATE(resblock_fwd_run(r, *d, nullptr, nullptr));
    const long long sv = r.saved.peak, w1 = r.ws.peak;
    Run b;
    run_init(b, d->dtype, true, nullptr, 0, nullptr, 0, nullptr);
    int dummy;
    HS_PROPAGATE(resblock_bwd_run(b, *d, nullptr, nullptr, nullptr, &dummy));
    if (saved_bytes) *saved_bytes = std::max(sv, b.saved.peak);
    if (ws_bytes) *ws_bytes = std::max(w1, b.ws.peak);
    return HS_OK;
}
hs_status hs_resblock_fwd(const hs_resblock_desc* d, const void* x, void* y, void* saved, int64_t saved_bytes, void* ws,
                          int64_t ws_bytes, void* stream) {
    HS_REQUIRE(d && x && y && saved && ws, "resblock_fwd: null argument");
    PLAN_CHECK("resblock_fwd", d->dtype, saved_bytes, ws_bytes, resblock_fwd_run(r, *d, x, y));
    Run r;
    run_init(r, d->dtype, false, saved, saved_bytes, ws, ws_bytes, (hipStream_t)stream);
    return resblock_fwd_run(r, *d, x, y);
}
hs_status hs_resblock_bwd(const hs_resblock_desc* d, const void* x, const void* y, const void* dy, void* dx, void* saved,
                          int64_t saved_bytes, void* ws, int64_t ws_bytes, void* stream) {
    HS_REQUIRE(d && x && y && dy && saved && ws, "resblock_bwd: null argument");
    PLAN_CHECK("resblock_bwd", d->dtype, saved_bytes, ws_bytes, resblock_bwd_run(r, *d, x, y, dy, dx));
    Run r;
    run_init(r, d->dtype, false, saved, saved_bytes, ws, ws_bytes, (hipStream_t)stream);
    return resblock_bwd_run(r, *d, x, y, dy, dx);
}

hs_status hs_stem_query(const hs_stem_desc* d, int64_t* saved_bytes, int64_t* ws_bytes) {
    HS_REQUIRE(d, "stem_query: null desc");
    Run r;
    run_init(r, d->dtype, true, nullptr, 0, nullptr, 0, nullptr);
    HS_PROPAGATE(stem_fwd_run(r, *d, nullptr, nullptr));
    const long long sv = r.saved.peak, w1 = r.ws.peak;
    Run b;
    run_init(b, d->dtype, true, nullptr, 0, nullptr, 0, nullptr);
    hs_stem_desc dd = *d;
    float dummy;
    if (!dd.cb.dw) dd.cb.dw = &dummy;   // size the workspace for the trainable case
    HS_PROPAGATE(stem_bwd_run(b, dd, nullptr, nullptr));
    if (saved_bytes) *saved_bytes = std::max(sv, b.saved.peak);
    if (ws_bytes) *ws_bytes = std::max(w1, b.ws.peak);
    return HS_OK;
}
hs_status hs_stem_fwd(const hs_stem_desc* d, const float* image, void* y, void* saved, int64_t saved_bytes, void* ws,
                      int64_t ws_bytes, void* stream) {
    HS_REQUIRE(d && image && y && saved && ws, "stem_fwd: null argument");
    PLAN_CHECK("stem_fwd", d->dtype, saved_bytes, ws_bytes, stem_fwd_run(r, *d, image, y));
    Run r;
    run_init(r, d->dtype, false, saved, saved_bytes, ws, ws_bytes, (hipStream_t)stream);
    return stem_fwd_run(r, *d, image, y);
}
hs_status hs_stem_bwd(const hs_stem_desc* d, const void* y, const void* dy, void* saved, int64_t saved_bytes, void* ws,
                      int64_t ws_bytes, void* stream) {
    HS_REQUIRE(d && dy && saved && ws, "stem_bwd: null argument");
    PLAN_CHECK("stem_bwd", d->dtype, saved_bytes, ws_bytes, stem_bwd_run(r, *d, y, dy));
    Run r;
    run_init(r, d->dtype, false, saved, saved_bytes, ws, ws_bytes, (hipStream_t)stream);
    return stem_bwd_run(r, *d, y, dy);
}

hs_status hs_bert_layer_query(const hs_bert_layer_desc* d, int64_t* saved_bytes, int64_t* ws_bytes) {
    HS_REQUIRE(d, "bert_layer_query: null desc");
    Run r;
    run_init(r, d->dtype, true, nullptr, 0, nullptr, 0, nullptr);
    HS_PROPAGATE(bert_layer_fwd_run(r, *d, nullptr, nullptr));
    const long long sv = r.saved.peak, w1 = r.ws.peak;
    Run b;
    run_init(b, d->dtype, true, nullptr, 0, nullptr, 0, nullptr);
    hs_bert_layer_desc dd = *d;
    float dummy;
    hs_linear* ls[6] = {&dd.q, &dd.k, &dd.v, &dd.ao, &dd.inter_l, &dd.out_l};
    for (auto* l : ls) {   // size the workspace for the trainable case
        if (!l->dw) l->dw = &dummy;
        if (!l->db) l->db = &dummy;
    }
    int dmy;
    HS_PROPAGATE(bert_layer_bwd_run(b, dd, nullptr, nullptr, &dmy));
    if (saved_bytes) *saved_bytes = std::max(sv, b.saved.peak);
    if (ws_bytes) *ws_bytes = std::max(w1, b.ws.peak);
    return HS_OK;
}
hs_status hs_bert_layer_fwd(const hs_bert_layer_desc* d, const void* x, void* y, void* saved, int64_t saved_bytes,
                            void* ws, int64_t ws_bytes, void* stream) {
    HS_REQUIRE(d && x && y && saved && ws, "bert_layer_fwd: null argument");
    PLAN_CHECK("bert_layer_fwd", d->dtype, saved_bytes, ws_bytes, bert_layer_fwd_run(r, *d, x, y));
    Run r;
    run_init(r, d->dtype, false, saved, saved_bytes, ws, ws_bytes, (hipStream_t)stream);
    return bert_layer_fwd_run(r, *d, x, y);
}
hs_status hs_bert_layer_bwd(const hs_bert_layer_desc* d, const void* x, const void* dy, void* dx, void* saved,
                            int64_t saved_bytes, void* ws, int64_t ws_bytes, void* stream) {
    HS_REQUIRE(d && x && dy && saved && ws, "bert_layer_bwd: null argument");
    PLAN_CHECK("bert_layer_bwd", d->dtype, saved_bytes, ws_bytes, bert_layer_bwd_run(r, *d, x, dy, dx));
    Run r;
    run_init(r, d->dtype, false, saved, saved_bytes, ws, ws_bytes, (hipStream_t)stream);
    return bert_layer_bwd_run(r, *d, x, dy, dx);
}

hs_status hs_linear_fwd(int32_t dtype, const void* x, int64_t M, int32_t ldx, const hs_linear* lin, const void* w_lp,
                        void* y, int32_t ldy, int32_t out_dtype, int32_t act, void* preact, const void* residual,
                        int32_t ldr, float dropout_p, uint64_t seed, void* stream) {
    HS_REQUIRE(x && lin && y && lin->w, "linear_fwd: null argument");
    HS_REQUIRE(dtype == HS_F32 || w_lp, "linear_fwd: bf16 mode needs the bf16 weight copy");
    Run r;
    run_init(r, dtype, false, nullptr, 0, nullptr, 0, (hipStream_t)stream);
    return linear_fwd_run(r, x, M, ldx, *lin, dtype == HS_F32 ? (const void*)lin->w : w_lp, y, ldy, out_dtype, act, preact,
                          residual, ldr, dropout_p, seed);
}
hs_status hs_linear_bwd(int32_t dtype, const void* x, int64_t M, int32_t ldx, const hs_linear* lin, const void* w_lp,
                        const void* dy, int32_t ldy, void* dx, int32_t lddx, int32_t dx_dtype, int32_t mul_mode,
                        const void* mul_src, int32_t ldm, const void* dx_residual, void* ws, int64_t ws_bytes,
                        void* stream) {
    HS_REQUIRE(x && lin && dy && lin->w, "linear_bwd: null argument");
    HS_REQUIRE(dtype == HS_F32 || w_lp || !dx, "linear_bwd: bf16 mode needs the bf16 weight copy");
    auto both = [&](Run& r) -> int {
        HS_PROPAGATE(linear_wgrad_run(r, x, M, ldx, *lin, dy, ldy));
        if (dx)
            HS_PROPAGATE(linear_dgrad_run(r, *lin, dtype == HS_F32 ? (const void*)lin->w : w_lp, dy, M, ldy, dx, lddx, dx_dtype,
                                          mul_mode, mul_src, ldm, dx_residual));
        return HS_OK;
    };
    PLAN_CHECK("linear_bwd", dtype, 0, ws_bytes, both(r));
    Run r;
    run_init(r, dtype, false, nullptr, 0, ws, ws_bytes, (hipStream_t)stream);
    HS_PROPAGATE(both(r));
    RUN_CHECK_ARENAS(r, "linear_bwd");
    return HS_OK;
}
int64_t hs_linear_bwd_ws_bytes(int64_t M, int32_t in_f, int32_t out_f, int32_t dtype) {
    // a plan-mode pass of the code hs_linear_bwd runs (split-K slabs incl. group slabs, column-sum partials, the transposed
    // operand copies of the K-contiguous form), for dense rows (ldx = in_f, ldy = out_f) and both gradients wanted
    Run r;
    run_init(r, dtype, true, nullptr, 0, nullptr, 0, nullptr);
    hs_linear lin;
    memset(&lin, 0, sizeof(lin));
    lin.in_f = in_f;
    lin.out_f = out_f;
    lin.dw = (float*)(uintptr_t)256;      // "wanted" markers: a plan-mode pass dereferences nothing
    lin.db = (float*)(uintptr_t)256;
    if (linear_wgrad_run(r, nullptr, M, in_f, lin, nullptr, out_f) != HS_OK) return -1;
    if (linear_dgrad_run(r, lin, nullptr, nullptr, M, out_f, nullptr, in_f, dtype, HS_MUL_NONE, nullptr, 0, nullptr) != HS_OK) return -1;
    return r.ws.peak + 1024;
}

/* see open_wgrad_group: collect the K-contiguous weight-gradient GEMMs of the hs_linear_bwd calls on `stream` ... */
hs_status hs_wgrad_group_begin(void* stream) {
    hipStream_t s = (hipStream_t)stream;
    GemmGroup* g = gemm_group_open(s, -1);
    HS_REQUIRE(g != nullptr, "wgrad_group_begin: cannot set up the group");
    std::lock_guard<std::mutex> lk(g_wg_mu);
    g_wg_open[s] = g;
    return HS_OK;
}
/* ... and launch them as one grouped grid (no-op when nothing was collected) */
hs_status hs_wgrad_group_end(void* stream) {
    hipStream_t s = (hipStream_t)stream;
    GemmGroup* g = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_wg_mu);
        auto it = g_wg_open.find(s);
        if (it != g_wg_open.end()) {
            g = it->second;
            g_wg_open.erase(it);
        }
    }
    return g ? gemm_group_flush(g, s) : HS_OK;
}
/* bf16 shadow of an f32 weight (see weight_c): `bf16` = NULL forgets the entry */
hs_status hs_weight_shadow_set(const float* w, void* bf16) {
    HS_REQUIRE(w != nullptr, "weight_shadow_set: null weight");
    std::lock_guard<std::mutex> lk(g_shadow_mu);
    if (bf16) g_shadow[w] = bf16;
    else g_shadow.erase(w);
    return HS_OK;
}
void hs_weight_shadow_clear(void) {
    std::lock_guard<std::mutex> lk(g_shadow_mu);
    g_shadow.clear();
}

/* sizeof of the ABI structs as this library was compiled (a binding checks its mirror declarations against it) */
int64_t hs_abi_sizeof(int32_t which) {
    switch (which) {
        case 0: return sizeof(hs_conv_geom);
        case 1: return sizeof(hs_gemm_params);
        case 2: return sizeof(hs_bn_params);
        case 3: return sizeof(hs_bn_bwd_params);
        case 4: return sizeof(hs_attn_desc);
        case 5: return sizeof(hs_conv_bn);
        case 6: return sizeof(hs_resblock_desc);
        case 7: return sizeof(hs_stem_desc);
        case 8: return sizeof(hs_linear);
        case 9: return sizeof(hs_norm);
        case 10: return sizeof(hs_bert_layer_desc);
        case 11: return sizeof(hs_resnet_desc);
        case 12: return sizeof(hs_resnet_plan);
        case 13: return sizeof(hs_bert_desc);
    }
    return -1;
}
/* ---- whole-tower executors ------------------------------------------------------------------ */
hs_status hs_resnet_query(const hs_resnet_desc* d, hs_resnet_plan* plan) {
    HS_REQUIRE(d && plan, "resnet_query: null argument");
    HS_PROPAGATE(resnet_check(*d));
    Run r;
    run_init(r, d->stem.dtype, true, nullptr, 0, nullptr, 0, nullptr);
    ResnetLayout lo;
    HS_PROPAGATE(resnet_fwd_run(r, *d, nullptr, lo));
    Run b;
    run_init(b, d->stem.dtype, true, nullptr, 0, nullptr, 0, nullptr);
    hs_resnet_desc dd = *d;           // size the workspace for the trainable case
    float dummy;
    auto need = [&](hs_conv_bn& c) { if (!c.dw) c.dw = &dummy; if (!c.dgamma) c.dgamma = &dummy; if (!c.dbeta) c.dbeta = &dummy; };
    need(dd.stem.cb);
    for (int i = 0; i < dd.n_blocks; ++i) {
        for (int j = 0; j < dd.blocks[i].n_main; ++j) need(dd.blocks[i].main[j]);
        if (dd.blocks[i].has_ds) need(dd.blocks[i].ds);
    }
    HS_PROPAGATE(resnet_bwd_run(b, dd, nullptr, lo));
    memset(plan, 0, sizeof(*plan));
    plan->saved_bytes = r.saved.peak;
    plan->ws_bytes = std::max(r.ws.peak, b.ws.peak);
    for (int t = 0; t < d->n_taps; ++t) {
        const int i = d->tap_block[t];
        int C, H, W;
        out_bytes_of(d->blocks[i], d->stem.dtype, &C, &H, &W);
        plan->tap_offset[t] = lo.y_off[i + 1];
        plan->tap_C[t] = C; plan->tap_H[t] = H; plan->tap_W[t] = W;
    }
    return HS_OK;
}
hs_status hs_resnet_fwd(const hs_resnet_desc* d, const float* image, void* saved, int64_t saved_bytes, void* ws,
                        int64_t ws_bytes, void* stream) {
    HS_REQUIRE(d && image && saved && ws, "resnet_fwd: null argument");
    HS_PROPAGATE(resnet_check(*d));
    ResnetLayout lo;
    {
        Run pl;
        run_init(pl, d->stem.dtype, true, nullptr, 0, nullptr, 0, nullptr);
        HS_PROPAGATE(resnet_fwd_run(pl, *d, nullptr, lo));
        HS_REQUIRE(pl.saved.peak <= saved_bytes && pl.ws.peak <= ws_bytes,
                   "resnet_fwd: arena too small (saved %lld needed / %lld given, ws %lld needed / %lld given)", pl.saved.peak,
                   (long long)saved_bytes, pl.ws.peak, (long long)ws_bytes);
    }
    Run r;
    run_init(r, d->stem.dtype, false, saved, saved_bytes, ws, ws_bytes, (hipStream_t)stream);
    HS_PROPAGATE(resnet_fwd_run(r, *d, image, lo));
    RUN_CHECK_ARENAS(r, "resnet_fwd");
    return HS_OK;
}
hs_status hs_resnet_bwd(const hs_resnet_desc* d, const void* const* dy_taps, void* saved, int64_t saved_bytes, void* ws,
                        int64_t ws_bytes, void* stream) {
    HS_REQUIRE(d && dy_taps && saved && ws, "resnet_bwd: null argument");
    HS_PROPAGATE(resnet_check(*d));
    ResnetLayout lo;
    {
        Run pl;
        run_init(pl, d->stem.dtype, true, nullptr, 0, nullptr, 0, nullptr);
        HS_PROPAGATE(resnet_fwd_run(pl, *d, nullptr, lo));
        Run pb;
        run_init(pb, d->stem.dtype, true, nullptr, 0, nullptr, 0, nullptr);
        HS_PROPAGATE(resnet_bwd_run(pb, *d, nullptr, lo));
        HS_REQUIRE(pl.saved.peak <= saved_bytes && pb.ws.peak <= ws_bytes,
                   "resnet_bwd: arena too small (saved %lld needed / %lld given, ws %lld needed / %lld given)", pl.saved.peak,
                   (long long)saved_bytes, pb.ws.peak, (long long)ws_bytes);
    }
    Run r;
    run_init(r, d->stem.dtype, false, saved, saved_bytes, ws, ws_bytes, (hipStream_t)stream);
    HS_PROPAGATE(resnet_bwd_run(r, *d, dy_taps, lo));
    RUN_CHECK_ARENAS(r, "resnet_bwd");
    return HS_OK;
}

/* test / analysis aid: where the post-activation buffers of the image tower live inside `saved` (every ReLU output and the
   max-pool arg-max), in forward order: out[0] = stem BN+ReLU output [N][P][Q][C], out[1] = stem max-pool arg-max taps
   (uint8 [N][P2][Q2][C], tap = 3*dh + dw of the 3x3 window), then per block its inner stage outputs (n_main - 1 entries)
   followed by the block output.  Returns the number of entries written (<= max_entries), or -1. */
int32_t hs_resnet_debug_offsets(const hs_resnet_desc* d, int64_t* out, int32_t max_entries) {
    if (!d || !out || resnet_check(*d) != HS_OK) return -1;
    char* const fake = (char*)(uintptr_t)0x100000;
    Run r;
    run_init(r, d->stem.dtype, true, nullptr, 0, nullptr, 0, nullptr);
    ResnetLayout lo;
    if (resnet_fwd_run(r, *d, nullptr, lo) != HS_OK) return -1;
    int n = 0;
    auto put = [&](long long off) { if (n < max_entries) out[n] = off; ++n; };
    {   // stem: replay its layout at its offset with a non-null base so that the pointers carry the offsets
        Run q;
        run_init(q, d->stem.dtype, false, fake, 1ll << 50, fake, 1ll << 50, nullptr);
        q.saved.off = lo.lay_off[0];
        StemLayout L;
        if (stem_layout(q, d->stem, L) != HS_OK) return -1;
        put((char*)L.a - fake);
        put((char*)L.idx - fake);
    }
    for (int i = 0; i < d->n_blocks; ++i) {
        Run q;
        run_init(q, d->stem.dtype, false, fake, 1ll << 50, fake, 1ll << 50, nullptr);
        q.saved.off = lo.lay_off[i + 1];
        ResLayout L;
        if (res_layout(q, d->blocks[i], L) != HS_OK) return -1;
        for (int j = 0; j + 1 < d->blocks[i].n_main; ++j) put((char*)L.main[j].a - fake);
        put(lo.y_off[i + 1]);
    }
    return n <= max_entries ? n : -1;
}

hs_status hs_bert_query(const hs_bert_desc* d, int64_t* saved_bytes, int64_t* ws_bytes, int64_t* out_offset) {
    HS_REQUIRE(d, "bert_query: null desc");
    HS_PROPAGATE(bert_check(*d));
    Run r;
    run_init(r, d->dtype, true, nullptr, 0, nullptr, 0, nullptr);
    BertTowerLayout lo;
    HS_PROPAGATE(bert_fwd_run(r, *d, nullptr, nullptr, lo));
    Run b;
    run_init(b, d->dtype, true, nullptr, 0, nullptr, 0, nullptr);
    hs_bert_desc dd = *d;
    float dummy;
    for (int i = 0; i < dd.n_layers; ++i) {
        hs_linear* ls[6] = {&dd.layers[i].q, &dd.layers[i].k, &dd.layers[i].v, &dd.layers[i].ao, &dd.layers[i].inter_l, &dd.layers[i].out_l};
        for (auto* l : ls) {
            if (!l->dw) l->dw = &dummy;
            if (!l->db) l->db = &dummy;
        }
    }
    if (!dd.dword) dd.dword = &dummy;
    if (!dd.dtype0) dd.dtype0 = &dummy;
    HS_PROPAGATE(bert_bwd_run(b, dd, nullptr, nullptr, nullptr, lo));
    if (saved_bytes) *saved_bytes = r.saved.peak;
    if (ws_bytes) *ws_bytes = std::max(r.ws.peak, b.ws.peak);
    if (out_offset) *out_offset = lo.y_off[d->n_layers];
    return HS_OK;
}
hs_status hs_bert_fwd(const hs_bert_desc* d, const int64_t* ids, const int64_t* mask, void* saved, int64_t saved_bytes,
                      void* ws, int64_t ws_bytes, void* stream) {
    HS_REQUIRE(d && ids && saved && ws, "bert_fwd: null argument");
    HS_PROPAGATE(bert_check(*d));
    BertTowerLayout lo;
    {
        Run pl;
        run_init(pl, d->dtype, true, nullptr, 0, nullptr, 0, nullptr);
        HS_PROPAGATE(bert_fwd_run(pl, *d, nullptr, nullptr, lo));
        HS_REQUIRE(pl.saved.peak <= saved_bytes && pl.ws.peak <= ws_bytes,
                   "bert_fwd: arena too small (saved %lld needed / %lld given, ws %lld needed / %lld given)", pl.saved.peak,
                   (long long)saved_bytes, pl.ws.peak, (long long)ws_bytes);
    }
    Run r;
    run_init(r, d->dtype, false, saved, saved_bytes, ws, ws_bytes, (hipStream_t)stream);
    HS_PROPAGATE(bert_fwd_run(r, *d, ids, mask, lo));
    RUN_CHECK_ARENAS(r, "bert_fwd");
    return HS_OK;
}
hs_status hs_bert_bwd(const hs_bert_desc* d, const int64_t* ids, const int64_t* mask, const void* dy, void* saved,
                      int64_t saved_bytes, void* ws, int64_t ws_bytes, void* stream) {
    HS_REQUIRE(d && ids && dy && saved && ws, "bert_bwd: null argument");
    HS_PROPAGATE(bert_check(*d));
    BertTowerLayout lo;
    {
        Run pl;
        run_init(pl, d->dtype, true, nullptr, 0, nullptr, 0, nullptr);
        HS_PROPAGATE(bert_fwd_run(pl, *d, nullptr, nullptr, lo));
        Run pb;
        run_init(pb, d->dtype, true, nullptr, 0, nullptr, 0, nullptr);
        HS_PROPAGATE(bert_bwd_run(pb, *d, nullptr, nullptr, nullptr, lo));
        HS_REQUIRE(pl.saved.peak <= saved_bytes && pb.ws.peak <= ws_bytes,
                   "bert_bwd: arena too small (saved %lld needed / %lld given, ws %lld needed / %lld given)", pl.saved.peak,
                   (long long)saved_bytes, pb.ws.peak, (long long)ws_bytes);
    }
    Run r;
    run_init(r, d->dtype, false, saved, saved_bytes, ws, ws_bytes, (hipStream_t)stream);
    HS_PROPAGATE(bert_bwd_run(r, *d, ids, mask, dy, lo));
    RUN_CHECK_ARENAS(r, "bert_bwd");
    return HS_OK;
}
/* weight-gradient side stream inside the composites: 1 on (default, or HAMSPINE_OVERLAP), 0 off (every kernel of a
   composite on the caller's stream, e.g. to time kernels in isolation). */
void hs_set_overlap(int32_t on) { hs::g_overlap = on ? 1 : 0; }
hs_status hs_grad_milestones(int32_t n, const void* const* grad_ptrs, void* const* events) {
    HS_REQUIRE(n >= 0 && n <= kMaxMilestones, "grad_milestones: %d entries (max %d)", n, kMaxMilestones);
    HS_REQUIRE(n == 0 || (grad_ptrs && events), "grad_milestones: null argument");
    g_ms.n = n;
    for (int i = 0; i < n; ++i) {
        HS_REQUIRE(grad_ptrs[i] && events[i], "grad_milestones: null entry %d", i);
        g_ms.ptr[i] = grad_ptrs[i];
        g_ms.ev[i] = (hipEvent_t)events[i];
    }
    return HS_OK;
}
int32_t hs_measure_build(void) {
#ifdef HS_MEASURE
    return 1;
#else
    return 0;
#endif
}
/* BertLayer weight gradients from transposed (K-contiguous) operands: 1 on (default), 0 = the row-major "tn" form. */
void hs_set_wgrad_nt(int32_t on) { hs::g_wgrad_nt = on ? 1 : 0; }
}
