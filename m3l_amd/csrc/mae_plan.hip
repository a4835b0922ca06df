// Host-side runtime of the MI355X-native MAE step: the C ABI of include/m3l_amd.h.
// Each entry point lays its activations out in the caller's workspace (bump arena, 256-byte aligned) and issues the
// kernel sequence of one module (patch embed / transformer stack / un-shuffle / heads+loss), forward or backward, on
// the caller's HIP stream.  No allocation, no synchronisation, no host<->device copies: safe to capture in a hipGraph.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/m3l_amd.h"
#include "common.cuh"
#include "kernels.h"


namespace {

constexpr float LN_EPS = 1e-5f;

// Side stream for work that is off the critical path of the backward (the grouped weight-gradient GEMMs): created once
// per process and per device, joined to the caller's stream with events (fork / join — legal inside stream capture too).
struct SideStream {
    hipStream_t s = nullptr;
    std::vector<hipEvent_t> ev;
    int next = 0;
};
// one side stream + event ring per DEVICE (created on first use, never destroyed), guarded by a mutex: two host threads that drive
// two devices get their own; two threads on ONE device share the stream and take events under the lock (the contract of the header
// — one host thread per device drives a step — is what keeps the event ring from being recycled under a waiter)
std::mutex g_side_mu;
std::map<int, SideStream> g_sides;
thread_local SideStream* t_side = nullptr;
#define g_side (*t_side)
int side_init() {
    int dev = 0;
    M3L_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_side_mu);
    SideStream& ss = g_sides[dev];
    t_side = &ss;
    if (ss.s) return 0;
    // LOWEST stream priority, for two reasons.  (1) The work is off the critical path (weight gradients, next layers' weight
    // casts).  (2) HIP multiplexes streams of one priority class onto a small pool of hardware queues; once a process holds
    // many streams (torch's stream pool after torch.distributed initialises RCCL) a normal-priority side stream lands on the
    // SAME hardware queue as the caller's stream and the two serialise (measured: -12 % step rate).  Each priority class has
    // its own queue pool and nothing else in a torch process uses the lowest one.
    int prio_least = 0, prio_greatest = 0;
    M3L_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    M3L_HIP(hipStreamCreateWithPriority(&ss.s, hipStreamNonBlocking, prio_least));
    ss.ev.resize(64);
    for (auto& e : ss.ev) M3L_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return 0;
}
// Deferred join (m3l_set_defer_join): the weight gradients of the LAST layers of a backward range have nothing left to hide behind
// inside the range; with the join deferred they overlap whatever the caller enqueues next (the glue / encoder backward, the patch
// embed backward) and the caller orders them with m3l_side_join before it consumes the gradients (optimizer step / all-reduce).
// The caller must also keep the workspaces alive until then (the side stream still reads them).
// (Process-wide and mutex-guarded, NOT thread-local: torch runs backward on an autograd worker thread, the join comes from the thread
// that called backward().)
int g_defer_join = 0;
std::vector<hipEvent_t> g_pending;
std::map<const void*, std::vector<hipEvent_t>> g_set_done;   // operand-set events of a backward in progress, by workspace (see bwd_range)
std::vector<hipEvent_t> g_tail_ring;
int g_tail_next = 0;
hipEvent_t tail_event() {
    std::lock_guard<std::mutex> lock(g_side_mu);
    if (g_tail_ring.empty()) {
        g_tail_ring.resize(16);
        for (auto& e : g_tail_ring) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    }
    hipEvent_t e = g_tail_ring[g_tail_next];
    g_tail_next = (g_tail_next + 1) % (int)g_tail_ring.size();
    return e;
}
hipEvent_t side_event() {
    std::lock_guard<std::mutex> lock(g_side_mu);
    hipEvent_t e = g_side.ev[g_side.next];
    g_side.next = (g_side.next + 1) % (int)g_side.ev.size();
    return e;
}

// A section of off-critical-path work of a backward call (small weight-gradient GEMMs, bias column sums).  In deferred-join mode
// fork(st) orders the library's side stream behind everything queued on `st` and returns it, and end() leaves ONE pending tail event
// for m3l_side_join; otherwise the work stays on the caller's stream.
// M3L_WGRAD_INLINE=1: every weight gradient on the caller's stream (no overlap: per-kernel PMC counters attribute cleanly) — diagnostic
int g_wgrad_inline = -1;      // -1 = environment not read yet
bool wgrad_inline() {
    if (g_wgrad_inline < 0) g_wgrad_inline = getenv("M3L_WGRAD_INLINE") != nullptr && atoi(getenv("M3L_WGRAD_INLINE")) > 0 ? 1 : 0;
    return g_wgrad_inline > 0;
}
struct SideSection {
    bool on = false;
    int fork(hipStream_t st, hipStream_t* s) {
        *s = st;
        if (!g_defer_join || wgrad_inline()) return 0;
        if (side_init()) return 2;
        hipEvent_t ready = side_event();
        M3L_HIP(hipEventRecord(ready, st));
        M3L_HIP(hipStreamWaitEvent(g_side.s, ready, 0));
        *s = g_side.s;
        on = true;
        return 0;
    }
    int end() {
        if (!on) return 0;
        hipEvent_t tail = tail_event();
        M3L_HIP(hipEventRecord(tail, g_side.s));
        std::lock_guard<std::mutex> lock(g_side_mu);
        g_pending.push_back(tail);
        on = false;
        return 0;
    }
};

struct Arena {
    char* base;
    size_t off = 0;
    explicit Arena(void* b) : base(reinterpret_cast<char*>(b)) {}
    void* take(size_t bytes) {
        off = (off + 255) & ~size_t(255);
        void* p = base ? base + off : nullptr;
        off += bytes;
        return p;
    }
    template <typename T> T* take_n(size_t n) { return reinterpret_cast<T*>(take(n * sizeof(T))); }
};

inline int pad8(int x) { return (x + 7) & ~7; }
inline int pad64(int x) { return (x + 63) & ~63; }
// LayerNorm forward / backward fused into the epilogue of the neighbouring GEMM (gemm_rowln.hip).  Correct (parity-tested) but
// measured 2-3 % SLOWER than the separate kernels at cfg 2 on MI355X (a full-row tile caps the kernel at 2 workgroups per CU,
// the stand-alone LayerNorm kernels run at 8 waves per SIMD), so it is off unless requested (M3L_ROWLN=1 / m3l_set_rowln).
int g_rowln = -1;
inline bool use_rowln() {
    if (g_rowln < 0) g_rowln = getenv("M3L_ROWLN") != nullptr ? 1 : 0;
    return g_rowln == 1;
}
inline size_t esz(int dtype) { return dtype ? 2 : 4; }
// Hidden activation h = GELU(u) of the fused MLP kernels: 1 = not written by the forward — the weight-gradient kernel of fc2 stages the
// pre-activation u instead and applies the GELU in its LDS stage (wgrad.hip), bit-identical gradients, 0.7 GB less HBM traffic per cfg-2
// step — and a 40 % longer weight-gradient kernel (140 -> 196 us: the GELU pass is NOT hidden behind its memory waits), which the side
// stream absorbs: same step time.  0 (default) = saved.  env M3L_DROP_H / m3l_set_drop_h.
int g_drop_h = -1;
inline bool drop_h() {
    if (g_drop_h < 0) g_drop_h = getenv("M3L_DROP_H") ? (atoi(getenv("M3L_DROP_H")) > 0 ? 1 : 0) : 0;
    return g_drop_h == 1;
}
// bf16 residual stream (round 4; default ON, env M3L_RES_BF16=0 / m3l_set_residual_bf16(0) restore fp32): x, x1, xout of every layer of a stack and its running
// residual gradient travel through HBM as bf16 instead of fp32 (fp32 in registers: the arithmetic is unchanged, each half layer's result is
// rounded once).  The tails of the fused kernels move mostly residual-type data at ~5.5 TB/s, so halving it is direct time; the residual
// gradient needs no separate copy at all (the compute-type copy the GEMMs read IS it).  A stack runs in this mode when every layer takes
// kernels that support it; its input and its input gradient stay fp32 at the interface (one cast each).
int g_res_bf16 = -1;
inline bool res_bf16_mode() {
    if (g_res_bf16 < 0) g_res_bf16 = getenv("M3L_RES_BF16") ? (atoi(getenv("M3L_RES_BF16")) > 0 ? 1 : 0) : 1;
    return g_res_bf16 == 1;
}
struct RbScope {          // residual mode of the launches issued inside the scope (read by the launchers: m3l_call_rb)
    explicit RbScope(bool on) { m3l_set_call_rb(on ? 1 : 0); }
    ~RbScope() { m3l_set_call_rb(0); }
};
bool tf_rb(const m3l_tf_cfg* c, int B, int n, bool fuse) {
    if (!res_bf16_mode() || c->dtype != 1 || fuse || !c->project_out || c->depth < 1) return false;
    // every kernel a bf16 stack can take has a bf16-residual form — the per-sample blocks, the row tiles, and (since the EPI_RES16 epilogue
    // of the NT GEMM and the typed dres of ln_bwd) the per-op chain — except the one-launch encoder backward; the residual epilogue of the
    // out-proj / fc2 GEMMs needs K = heads * 64 / mlp_dim to be a multiple of the 64-element K tile
    if (c->mlp_dim % 64 != 0) return false;
    if ((m3l_enc_mega_enabled() & 2) && m3l_attn_block_supported(1, c->dim, c->heads, n, c->project_out)) return false;
    (void)B;
    return true;
}
// which fused kernel (if any) computes the feed-forward half of a layer in the FORWARD — the same decisions as m3l_transformer_fwd below:
// 0 = per-op GEMMs (h is the operand of the fc2 GEMM: always saved), 1 = per-sample block kernel, 2 = row-tiled kernel
// (both evaluate the fitted GELU, as every bf16 kernel does).  The backward reads it to know whether h exists and which GELU reproduces it.
int fused_mlp_kind(const m3l_tf_cfg* c, int B, int n, bool fuse) {
    const int M = B * n, D = c->dim, mlp = c->mlp_dim, dt = c->dtype;
    if (fuse || dt != 1) return 0;
    const bool block = m3l_attn_block_supported(dt, D, c->heads, n, c->project_out);
    const bool t192 = m3l_mlp_t192_supported(dt, D, mlp, M);
    if (block && !(m3l_mlp_t192_short() && t192) && m3l_mlp_block_supported(dt, D, mlp, n)) return 1;
    return t192 ? 2 : 0;
}

// scratch big enough for every reduction / split-K slab of one module
size_t scratch_bytes(int M, const std::vector<std::pair<int, int>>& wshapes, int maxcols) {
    size_t s = (size_t)2048 * 4 * (size_t)maxcols * sizeof(float);
    for (auto& nk : wshapes) s = std::max(s, m3l_gemm_tn_ws_bytes(M, nk.first, nk.second, nullptr));
    return s;
}

GemmEpi epi0(int ldc) {
    GemmEpi e;
    memset(&e, 0, sizeof(e));
    e.ldc = ldc;
    e.alpha = 1.f;
    return e;
}

// ---------------------------------------------------------------------------------------------------------------
// geometry helpers
struct Geo {
    int n_img, n_tac, k;              // active counts (0 when the modality is absent from the call)
    int pd_img, pd_tac, pdp_img, pdp_tac;
};
Geo geo_of(const m3l_geom* g) {
    Geo o;
    o.n_img = g->use_vision ? (g->image_h / g->image_patch) * (g->image_w / g->image_patch) : 0;
    o.k = (g->use_tactile && g->num_tactiles > 0) ? g->num_tactiles : 0;
    o.n_tac = o.k ? (g->tactile_h / g->tactile_patch) * (g->tactile_w / g->tactile_patch) : 0;
    o.pd_img = g->image_channels * g->image_patch * g->image_patch;
    o.pd_tac = g->tactile_channels * g->tactile_patch * g->tactile_patch;
    // patch vectors are the K of the patch-embed GEMM and of the heads' dgrad: padded (zero columns) to the 64-element K tile of the
    // LDS-DMA GEMM — K = 48 (cfg 2's tactile patches) and K = 2352 (cfg 5) took the generic kernel before
    o.pdp_img = pad64(o.pd_img);
    o.pdp_tac = pad64(o.pd_tac);
    return o;
}
PatchGroup group_img(const m3l_geom* g, const Geo& ge, const float* image) {
    PatchGroup pg;
    memset(&pg, 0, sizeof(pg));
    pg.src[0] = image;
    pg.nsrc = 1;
    pg.C = g->image_channels; pg.H = g->image_h; pg.W = g->image_w; pg.P = g->image_patch;
    pg.npatch = ge.n_img > 0 ? ge.n_img : 1;
    pg.base = 0;
    return pg;
}
PatchGroup group_tac(const m3l_geom* g, const Geo& ge, const float* const* tactiles) {
    PatchGroup pg;
    memset(&pg, 0, sizeof(pg));
    for (int i = 0; i < ge.k; ++i) pg.src[i] = tactiles[i];
    pg.nsrc = ge.k > 0 ? ge.k : 1;
    pg.C = g->tactile_channels; pg.H = g->tactile_h; pg.W = g->tactile_w; pg.P = g->tactile_patch;
    pg.npatch = ge.n_tac > 0 ? ge.n_tac : 1;
    pg.base = ge.n_img;
    return pg;
}
int check_geom(const m3l_geom* g) {
    M3L_CHECK(g->num_tactiles >= 0 && g->num_tactiles <= M3L_MAX_TACTILES, "num_tactiles=%d out of range", g->num_tactiles);
    M3L_CHECK(g->image_h % g->image_patch == 0 && g->image_w % g->image_patch == 0, "Image dimensions must be divisible by the patch size.");
    M3L_CHECK(g->num_tactiles == 0 || (g->tactile_h % g->tactile_patch == 0 && g->tactile_w % g->tactile_patch == 0),
              "Tactile dimensions must be divisible by the patch size.");
    M3L_CHECK(g->use_vision || (g->use_tactile && g->num_tactiles > 0), "neither vision nor tactile tokens present");
    return 0;
}
void mask_counts(const Geo& ge, double ratio, int* c) {
    // pretrain_models.py:223-227 — Python double arithmetic with int() truncation
    const int N = ge.n_img + ge.k * ge.n_tac;
    const int num_masked = (int)(ratio * (double)N);
    const double image_perc = (double)ge.n_img / (double)N;
    const int nm_img = (int)((double)num_masked * image_perc);
    const int nm_tac = ge.k > 0 ? (num_masked - nm_img) / ge.k : 0;
    const int total_masked = nm_img + ge.k * nm_tac;
    c[0] = total_masked; c[1] = N - total_masked; c[2] = nm_img; c[3] = nm_tac; c[4] = ge.n_img; c[5] = ge.n_tac;
}

// ---------------------------------------------------------------------------------------------------------------
// transformer stack
struct TfLayer {
    void *wqkv, *wqkvT, *wo, *woT, *w1, *w1T, *w2, *w2T;
    void *xn1, *qkv, *o, *xn2, *u, *h;
    float *lse, *x1, *xout;
};
struct TfWs {
    std::vector<TfLayer> L;
    float* dx;
    // operands of the weight-gradient GEMMs, a ring of `nset` per-layer sets (set of layer l = l % nset): the wgrads of `wg_batch`
    // layers go out as ONE grouped launch on the side stream, beside the dgrad chains of the layers below
    int nset, wg_batch;
    std::vector<void*> dx_t, dx1_t, du, dqkv;
    std::vector<float*> scratch2;   // [cdiv(M,128) or B][mlp] column-sum partials of the fused dgrad epilogue (fc1 bias gradient), per set
    void *dxn, *d_o;
    float* scratch_tn;
    float* dsum;
    float* scratch;
    float* ln_part;       // [2 depth + 1][ln_part_stride]: dgamma / dbeta / bias-grad partials of every LayerNorm backward, reduced in one launch
    size_t ln_part_stride;
    size_t scratch_b, scratch_tn_b;
    size_t total;
};
// bf16 residual mode: the fp32 running-gradient buffer w.dx is not used; its memory holds the stack's input as bf16 (kept for the backward
// of layer 0) and the bf16 gradient of layer 0's input (cast to the caller's fp32 dx_in)
inline void* rb_xin(const TfWs& w) { return w.dx; }
inline void* rb_out0(const TfWs& w, size_t M, size_t D) { return reinterpret_cast<char*>(w.dx) + M * D * 2; }

void layer_wgrad_problems(TnProblem* pr, int D, int HD, int mlp, bool project_out, int* np_out) {
    memset(pr, 0, 4 * sizeof(TnProblem));
    int np = 0;
    pr[np].N = D; pr[np++].K = mlp;
    pr[np].N = mlp; pr[np++].K = D;
    pr[np].N = 3 * HD; pr[np++].K = D;
    if (project_out) { pr[np].N = D; pr[np++].K = HD; }
    *np_out = np;
}
TfWs tf_layout(const m3l_tf_cfg* c, int B, int n, void* ws) {
    Arena a(ws);
    const size_t M = (size_t)B * n, D = c->dim, HD = (size_t)c->heads * 64, mlp = c->mlp_dim, e = esz(c->dtype);
    TfWs w;
    w.L.resize(c->depth);
    for (auto& l : w.L) {
        l.wqkv = a.take(3 * HD * D * e); l.wqkvT = a.take(D * 3 * HD * e);
        l.wo = a.take(D * HD * e);       l.woT = a.take(HD * D * e);
        l.w1 = a.take(mlp * D * e);      l.w1T = a.take(D * mlp * e);
        l.w2 = a.take(D * mlp * e);      l.w2T = a.take(mlp * D * e);
        l.xn1 = a.take(M * D * e);
        l.qkv = a.take(M * 3 * HD * e);
        l.o = a.take(M * HD * e);
        l.lse = a.take_n<float>((size_t)B * c->heads * n);
        l.x1 = a.take_n<float>(M * D);
        l.xn2 = a.take(M * D * e);
        l.u = a.take(M * mlp * e);
        l.h = a.take(M * mlp * e);
        l.xout = a.take_n<float>(M * D);
    }
    w.dx = a.take_n<float>(M * D);
    // layers per weight-gradient launch: enough tiles to give every CU a workgroup with >= ~2048 rows per split (wgrad.hip)
    TnProblem pr[4 * 4];
    int np = 0;
    layer_wgrad_problems(pr, (int)D, (int)HD, (int)mlp, c->project_out != 0, &np);
    w.wg_batch = c->dtype == 1 ? std::max(1, std::min(std::min(4, std::max(1, c->depth)), m3l_wgrad_layers_per_launch((int)M, pr, np))) : 1;
    {   // experiment switch: layers per grouped weight-gradient launch (<= 4: M3L_TN_MAX_PROBLEMS / 4 weights per layer)
        static const int forced = getenv("M3L_WGRAD_LAYERS") ? atoi(getenv("M3L_WGRAD_LAYERS")) : 0;
        if (forced > 0 && c->dtype == 1) w.wg_batch = std::max(1, std::min(std::min(4, std::max(1, c->depth)), forced));
    }
    w.nset = std::max(1, std::min(std::max(1, c->depth), 2 * w.wg_batch));
    w.dx_t.resize(w.nset); w.dx1_t.resize(w.nset); w.du.resize(w.nset); w.dqkv.resize(w.nset); w.scratch2.resize(w.nset);
    for (int i = 0; i < w.nset; ++i) {
        w.dx_t[i] = a.take(M * D * e);
        w.dx1_t[i] = a.take(M * D * e);
        w.du[i] = a.take(M * mlp * e);
        w.dqkv[i] = a.take(M * 3 * HD * e);
    }
    w.dxn = a.take(M * D * e);
    w.d_o = a.take(M * HD * e);
    w.dsum = a.take_n<float>((size_t)B * c->heads * n);
    std::vector<std::pair<int, int>> shapes = {{(int)(3 * HD), (int)D}, {(int)D, (int)HD}, {(int)mlp, (int)D}, {(int)D, (int)mlp}};
    w.scratch_b = scratch_bytes((int)M, shapes, (int)std::max(std::max(mlp, 3 * HD), D));
    for (int j = 1; j < w.wg_batch; ++j) memcpy(pr + j * np, pr, np * sizeof(TnProblem));
    // a group may hold fewer layers than wg_batch (the last group of a stack, a chunked backward's range ends), and fewer tiles mean more
    // splits per tile: the slab workspace is sized for the worst group size
    w.scratch_tn_b = 0;
    for (int j = 1; j <= w.wg_batch; ++j) w.scratch_tn_b = std::max(w.scratch_tn_b, m3l_gemm_tn_grouped_ws_bytes((int)M, pr, np * j));
    w.scratch = reinterpret_cast<float*>(a.take(w.scratch_b));
    w.scratch_tn = reinterpret_cast<float*>(a.take(w.scratch_tn_b));
    // partial rows of the row-tiled kernels: sized for EVERY tile shape they may pick (48- / 96- / 128- / 192-row tiles; one fc1-bias row per
    // wave at most), so that the layout does not depend on the tuning switches in force at the time of the call
    const int t192_rows_max = (int)((M + 15) / 16) + 16, t192_tiles_max = (int)((M + 47) / 48) + 1;
    for (int i = 0; i < w.nset; ++i) w.scratch2[i] = a.take_n<float>((size_t)std::max(std::max(m3l_gemm_nt_colsum_rows((int)M, (int)mlp), B), t192_rows_max) * mlp);   // B: one partial row per sample (block kernels)
    w.ln_part_stride = (size_t)std::max(std::max(m3l_ln_bwd_blocks((int)M), B), t192_tiles_max) * 3 * D;
    w.ln_part = a.take_n<float>((size_t)(2 * c->depth + 1) * w.ln_part_stride);
    w.total = a.off + 256;
    return w;
}
int check_tf(const m3l_tf_cfg* c, int B, int n) {
    M3L_CHECK(c->dtype == 0 || c->dtype == 1, "transformer: bad dtype %d", c->dtype);
    M3L_CHECK(c->dim > 0 && c->dim % 8 == 0 && c->dim <= 1024, "transformer: dim=%d must be a multiple of 8, <= 1024", c->dim);
    M3L_CHECK(c->mlp_dim > 0 && c->mlp_dim % 8 == 0, "transformer: mlp_dim=%d must be a multiple of 8", c->mlp_dim);
    M3L_CHECK(c->heads > 0 && c->depth >= 0, "transformer: heads=%d depth=%d", c->heads, c->depth);
    M3L_CHECK(c->project_out || c->heads * 64 == c->dim, "transformer: identity to_out needs heads*64 == dim");
    M3L_CHECK(c->dim % 64 == 0, "transformer: dim=%d must be a multiple of 64 (LDS-DMA GEMM K-tile)", c->dim);
    M3L_CHECK(B > 0 && n > 0, "transformer: empty input B=%d n=%d", B, n);
    return 0;
}

}  // namespace

// =================================================================================================================
extern "C" {

int m3l_version(void) { return 400; }

// diagnostic switch (bench.py's stand-alone roofline figure, PMC passes): 1 = every weight gradient on the caller's stream
int m3l_set_wgrad_inline(int on) {
    const int old = wgrad_inline() ? 1 : 0;
    g_wgrad_inline = on ? 1 : 0;
    return old;
}

// the library's side stream of the current device (created on first use): callers that want to order their own off-critical-path work
// behind the weight gradients (m3l_amd.parallel issues the gradient all-reduce's bookkeeping there) use it as an external stream
void* m3l_side_stream(void) {
    if (side_init()) return nullptr;
    return (void*)g_side.s;
}

// building blocks for work that other translation units put on the side stream (comm.hip): fork = the side stream waits for
// everything queued on `after_stream` so far and is returned; mark_pending = an event behind the side stream's last work joins the
// pending list that m3l_side_join consumes
int m3l_side_fork(void* after_stream, void** side_out) {
    if (side_init()) return 2;
    hipEvent_t ready = side_event();
    M3L_HIP(hipEventRecord(ready, (hipStream_t)after_stream));
    M3L_HIP(hipStreamWaitEvent(g_side.s, ready, 0));
    *side_out = (void*)g_side.s;
    return 0;
}
int m3l_side_mark_pending(void) {
    if (side_init()) return 2;
    hipEvent_t tail = tail_event();
    M3L_HIP(hipEventRecord(tail, g_side.s));
    std::lock_guard<std::mutex> lock(g_side_mu);
    g_pending.push_back(tail);
    return 0;
}

int m3l_set_defer_join(int on) {
    const int old = g_defer_join;
    g_defer_join = on ? 1 : 0;
    return old;
}

int m3l_side_join(void* stream) {
    std::vector<hipEvent_t> evs;
    {
        std::lock_guard<std::mutex> lock(g_side_mu);
        evs.swap(g_pending);
    }
    for (hipEvent_t ev : evs) M3L_HIP(hipStreamWaitEvent((hipStream_t)stream, ev, 0));
    return (int)0;
}
int m3l_side_pending(void) {
    std::lock_guard<std::mutex> lock(g_side_mu);
    return (int)g_pending.size();
}

// 1 when this stack runs the bf16 residual stream at (B, n) (mae_step.hip: whether its neighbours exchange bf16 with it)
int m3l_transformer_rb(const m3l_tf_cfg* c, int B, int n) {
    const int D = c->dim, HD = c->heads * 64, mlp = c->mlp_dim;
    return tf_rb(c, B, n, use_rowln() && m3l_gemm_nt_rowln_supported(c->dtype, D, HD) && m3l_gemm_nt_rowln_supported(c->dtype, D, mlp)) ? 1 : 0;
}

int m3l_set_residual_bf16(int on) {
    const int old = res_bf16_mode() ? 1 : 0;
    g_res_bf16 = on ? 1 : 0;
    return old;
}

int m3l_set_drop_h(int on) {
    const int old = drop_h() ? 1 : 0;
    g_drop_h = on ? 1 : 0;
    return old;
}

int m3l_set_rowln(int enable) {
    const int old = use_rowln() ? 1 : 0;
    g_rowln = enable ? 1 : 0;
    return old;
}

int m3l_mask_counts(const m3l_geom* g, double ratio, int* counts_host) {
    if (check_geom(g)) return 1;
    M3L_CHECK(ratio > 0.0 && ratio < 1.0, "masking ratio must be kept between 0 and 1");
    mask_counts(geo_of(g), ratio, counts_host);
    return 0;
}

int m3l_mask_sample(const m3l_geom* g, double ratio, int B, const float* const* noise, int64_t* masked, int64_t* unmasked,
                    void* stream) {
    if (check_geom(g)) return 1;
    M3L_CHECK(ratio > 0.0 && ratio < 1.0, "masking ratio must be kept between 0 and 1");
    int c[6];
    mask_counts(geo_of(g), ratio, c);
    return m3l_mask_sample_counts(g, c[2], c[3], B, noise, masked, unmasked, stream);
}

// the same with explicit per-modality masked counts (VTMAE.reconstruct uses its own rule: int(r*n_img), int(r*n_tac_total/k),
// pretrain_models.py:425,433)
int m3l_mask_sample_counts(const m3l_geom* g, int nm_img, int nm_tac, int B, const float* const* noise, int64_t* masked,
                           int64_t* unmasked, void* stream) {
    if (check_geom(g)) return 1;
    hipStream_t st = (hipStream_t)stream;
    const Geo ge = geo_of(g);
    M3L_CHECK(nm_img >= 0 && nm_img <= ge.n_img && nm_tac >= 0 && nm_tac <= ge.n_tac, "mask_sample: bad counts %d %d", nm_img, nm_tac);
    const int nmask = nm_img + ge.k * nm_tac, nvis = ge.n_img + ge.k * ge.n_tac - nmask;
    // the image and every tactile sensor in ONE launch (blockIdx.y = modality group)
    MaskRankArgs a;
    memset(&a, 0, sizeof(a));
    int ni = 0;
    if (ge.n_img > 0) a.g[a.count++] = MaskRankGroup{noise[ni++], ge.n_img, nm_img, 0, 0, 0};
    for (int s = 0; s < ge.k; ++s)
        a.g[a.count++] = MaskRankGroup{noise[ni++], ge.n_tac, nm_tac, ge.n_img + s * ge.n_tac, nm_img + s * nm_tac,
                                       (ge.n_img - nm_img) + s * (ge.n_tac - nm_tac)};
    if (a.count == 0) return 0;
    return m3l_mask_rank_groups(&a, B, masked, nmask, unmasked, nvis, st);
}

// ---------------------------------------------------------------------------------------------------------------
// patch embed
namespace {
struct EmbGroupWs {
    void *w, *wT, *xn, *dE, *dxn;
    float* E;
};
struct EmbWs {
    EmbGroupWs g[2];
    float* scratch;
    float* scratch_side;     // partial sums of the weight-gradient work when it runs on the side stream (deferred join)
    size_t scratch_b, total;
};
EmbWs emb_layout(const Geo& ge, int D, int dtype, int B, void* ws) {
    Arena a(ws);
    EmbWs w;
    const size_t e = esz(dtype);
    const int pdp[2] = {ge.pdp_img, ge.pdp_tac};
    const int maxrows[2] = {B * ge.n_img, B * ge.k * ge.n_tac};   // capacity = all patches; a visible list uses a prefix
    for (int i = 0; i < 2; ++i) {
        w.g[i].w = a.take((size_t)D * pdp[i] * e);
        w.g[i].wT = a.take((size_t)pdp[i] * D * e);
        w.g[i].xn = a.take((size_t)maxrows[i] * pdp[i] * e);
        w.g[i].E = a.take_n<float>((size_t)maxrows[i] * D);
        w.g[i].dE = a.take((size_t)maxrows[i] * D * e);
        w.g[i].dxn = a.take((size_t)maxrows[i] * pdp[i] * e);
    }
    const int Mmax = std::max(maxrows[0], maxrows[1]);
    std::vector<std::pair<int, int>> shapes = {{D, ge.pdp_img}, {D, ge.pdp_tac}};
    w.scratch_b = scratch_bytes(std::max(Mmax, 1), shapes, std::max(D * (2 + M3L_MAX_TACTILES), std::max(ge.pdp_img, ge.pdp_tac)));
    w.scratch = reinterpret_cast<float*>(a.take(w.scratch_b));
    w.scratch_side = reinterpret_cast<float*>(a.take(w.scratch_b));
    w.total = a.off + 256;
    return w;
}
}  // namespace

size_t m3l_embed_ws_bytes(const m3l_geom* g, int D, int dtype, int B, int L) {
    if (check_geom(g)) return 0;
    const Geo ge = geo_of(g);
    (void)L;
    return emb_layout(ge, D, dtype, B, nullptr).total;
}

namespace {
int embed_run(bool backward, const m3l_geom* g, int D, int dtype, int B, int L, int cnt_img, const int64_t* idx, const float* image,
              const float* const* tactiles, const void* const* tensors, void* ws, float* tokens, const float* dtokens,
              float* const* grads, hipStream_t st) {
    if (check_geom(g)) return 1;
    M3L_CHECK(D % 8 == 0 && D <= 1024, "embed: dim=%d must be a multiple of 8, <= 1024", D);
    const Geo ge = geo_of(g);
    EmbWs w = emb_layout(ge, D, dtype, B, ws);
    M3L_CHECK(idx || L == ge.n_img + ge.k * ge.n_tac, "embed: L=%d must equal the number of patches when idx is NULL", L);
    const int cnt[2] = {cnt_img, L - cnt_img};
    const int j0[2] = {0, cnt_img};
    M3L_CHECK(cnt[0] >= 0 && cnt[1] >= 0 && cnt[0] <= ge.n_img && cnt[1] <= ge.k * ge.n_tac, "embed: inconsistent list split %d/%d", cnt[0], cnt[1]);
    const PatchGroup pgs[2] = {group_img(g, ge, image), group_tac(g, ge, tactiles)};
    const int pd[2] = {ge.pd_img, ge.pd_tac}, pdp[2] = {ge.pdp_img, ge.pdp_tac};
    const float* mod = (const float*)tensors[12];
    const float* pos[2] = {(const float*)tensors[13], (const float*)tensors[14]};
    const int mod0[2] = {0, 1};
    SideSection side;
    if (!backward) {
        // compute-type weight copies of both groups in ONE launch (the zero padding of K is written by the same kernel)
        WeightPack pk;
        memset(&pk, 0, sizeof(pk));
        for (int i = 0; i < 2; ++i) {
            if (cnt[i] == 0) continue;
            pk.d[pk.count++] = WeightDesc{(const float*)tensors[6 * i + 2], w.g[i].w, w.g[i].wT, D, pd[i], pdp[i], D};
        }
        if (m3l_prep_weights(dtype, &pk, st)) return 1;
        if (m3l_prep_mode() == 1) return 0;        // collect pass of a call chain (mae_step.hip): the copies are recorded, nothing else runs
    }
    for (int i = 0; i < 2; ++i) {
        if (cnt[i] == 0) continue;
        const int rows = B * cnt[i];
        const void* const* t = tensors + 6 * i;
        const float *ln1_w = (const float*)t[0], *ln1_b = (const float*)t[1], *W = (const float*)t[2], *bias = (const float*)t[3];
        const float *ln2_w = (const float*)t[4], *ln2_b = (const float*)t[5];
        if (!backward) {
            if (m3l_patch_ln(dtype, &pgs[i], idx, L, j0[i], cnt[i], B, ln1_w, ln1_b, LN_EPS, w.g[i].xn, pdp[i], st)) return 1;
            GemmEpi e = epi0(D);
            e.bias = bias;
            e.out_f32 = w.g[i].E;
            if (m3l_gemm_nt(dtype, w.g[i].xn, pdp[i], w.g[i].w, pdp[i], rows, D, pdp[i], &e, st)) return 1;
            if (m3l_embed_finalize(w.g[i].E, D, &pgs[i], idx, L, j0[i], cnt[i], B, ln2_w, ln2_b, LN_EPS, mod, mod0[i], pos[i], tokens, L, st))
                return 1;
        } else {
            float* const* gr = grads + 6 * i;
            if (m3l_embed_finalize_bwd(dtype, dtokens, L, w.g[i].E, D, &pgs[i], idx, L, j0[i], cnt[i], B, ln2_w, LN_EPS, w.g[i].dE,
                                       w.scratch, gr[4], gr[5], grads[12], mod0[i], 0, st))
                return 1;
            // projection weight / bias gradients: nothing downstream in this call reads them (side stream in deferred-join mode)
            hipStream_t s2;
            if (side.fork(st, &s2)) return 2;
            float* sc = side.on ? w.scratch_side : w.scratch;
            if (m3l_gemm_tn(dtype, w.g[i].dE, D, w.g[i].xn, pdp[i], rows, D, pdp[i], sc, w.scratch_b, gr[2], pd[i], D, pd[i], 0, s2)) return 1;
            if (m3l_colsum(dtype, w.g[i].dE, rows, D, D, sc, gr[3], 0, s2)) return 1;
            GemmEpi e = epi0(pdp[i]);
            e.out_t = w.g[i].dxn;
            if (m3l_gemm_nt(dtype, w.g[i].dE, D, w.g[i].wT, D, rows, pdp[i], D, &e, st)) return 1;
            if (m3l_patch_ln_bwd(dtype, &pgs[i], idx, L, j0[i], cnt[i], B, LN_EPS, w.g[i].dxn, pdp[i], w.scratch, gr[0], gr[1], 0, st))
                return 1;
        }
    }
    return side.end();
}
}  // namespace

int m3l_embed_fwd(const m3l_geom* g, int D, int dtype, int B, int L, int cnt_img, const int64_t* idx, const float* image,
                  const float* const* tactiles, const void* const* tensors, void* ws, float* tokens, void* stream) {
    return embed_run(false, g, D, dtype, B, L, cnt_img, idx, image, tactiles, tensors, ws, tokens, nullptr, nullptr, (hipStream_t)stream);
}
int m3l_embed_bwd(const m3l_geom* g, int D, int dtype, int B, int L, int cnt_img, const int64_t* idx, const float* image,
                  const float* const* tactiles, const void* const* tensors, void* ws, const float* dtokens, float* const* grads,
                  void* stream) {
    return embed_run(true, g, D, dtype, B, L, cnt_img, idx, image, tactiles, tensors, ws, nullptr, dtokens, grads, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// transformer
size_t m3l_transformer_ws_bytes(const m3l_tf_cfg* c, int B, int n) {
    if (check_tf(c, B, n)) return 0;
    return tf_layout(c, B, n, nullptr).total;
}

int m3l_transformer_fwd(const m3l_tf_cfg* c, int B, int n, const float* x_in, const void* const* tensors, void* ws, void* y_t,
                        float* y32, void* stream) {
    if (check_tf(c, B, n)) return 1;
    hipStream_t st = (hipStream_t)stream;
    TfWs w = tf_layout(c, B, n, ws);
    const int M = B * n, D = c->dim, HD = c->heads * 64, mlp = c->mlp_dim, dt = c->dtype;
    const float* x = x_in;
    // compute-type weight copies (+ transposes for the dgrads) of every layer: one launch per 16 layers, on the caller's stream
    {
        WeightPack pk;
        memset(&pk, 0, sizeof(pk));
        for (int l = 0; l < c->depth; ++l) {
            TfLayer& L = w.L[l];
            const void* const* t = tensors + 11 * l;
            pk.d[pk.count++] = WeightDesc{(const float*)t[2], L.wqkv, L.wqkvT, 3 * HD, D, D, 3 * HD};
            pk.d[pk.count++] = WeightDesc{(const float*)t[7], L.w1, L.w1T, mlp, D, D, mlp};
            pk.d[pk.count++] = WeightDesc{(const float*)t[9], L.w2, L.w2T, D, mlp, mlp, D};
            if (c->project_out) pk.d[pk.count++] = WeightDesc{(const float*)t[3], L.wo, L.woT, D, HD, HD, D};
            if (pk.count + 4 > M3L_WPACK || l + 1 == c->depth) {
                if (m3l_prep_weights(dt, &pk, st)) return 1;
                pk.count = 0;
            }
        }
    }
    if (m3l_prep_mode() == 1) return 0;            // collect pass of a call chain (mae_step.hip)
    // LayerNorms fused into the epilogue of the GEMM that produces their input (gemm_rowln.hip) when the row width allows it:
    // out-proj + LN2 of the layer, fc2 + LN1 of the next layer (or the final norm).  Only the very first LN1 is a kernel.
    const bool fuse = use_rowln() && m3l_gemm_nt_rowln_supported(dt, D, HD) && m3l_gemm_nt_rowln_supported(dt, D, mlp);
    const void* const* tfin = tensors + 11 * c->depth;
    bool final_done = false;
    // bf16 residual stream: the stack's input as bf16 (kept for the backward of layer 0 in the otherwise unused w.dx buffer); from here on
    // every `float*` of the residual stream points at bf16 data and the launchers are told so
    const bool rb = tf_rb(c, B, n, fuse);
    if (rb && !m3l_call_io()) {          // (m3l_call_io: the caller hands x_in over as bf16 already)
        if (m3l_cast_f32(1, x_in, (long)M * D, rb_xin(w), st)) return 1;
        x = reinterpret_cast<const float*>(rb_xin(w));
    }
    RbScope rb_scope(rb);
    // short sequences whose every half layer takes a block kernel: the whole stack in ONE launch (enc_mega.hip)
    int l_begin = 0;
    if ((m3l_enc_mega_enabled() & 1) && !fuse && c->depth >= 1 && c->depth <= M3L_MEGA_MAX_LAYERS && c->project_out &&
        m3l_attn_block_supported(dt, D, c->heads, n, c->project_out) && m3l_mlp_block_supported(dt, D, mlp, n) &&
        !(m3l_mlp_t192_short() && m3l_mlp_t192_supported(dt, D, mlp, M))) {
        const void* lay[M3L_MEGA_MAX_LAYERS][20];
        for (int l = 0; l < c->depth; ++l) {
            TfLayer& L = w.L[l];
            const void* const* t = tensors + 11 * l;
            const void* v[20] = {t[0], t[1], L.wqkv, L.wo, t[4], t[5], t[6], L.xn1, L.qkv, L.o, L.lse, L.x1, L.xn2,
                                 L.w1, t[8], L.w2, t[10], L.u, drop_h() ? nullptr : L.h, L.xout};
            memcpy(lay[l], v, sizeof(v));
        }
        if (m3l_enc_fwd_mega(D, mlp, B, n, x, &lay[0][0], c->depth, LN_EPS, st)) return 1;
        x = w.L[c->depth - 1].xout;
        l_begin = c->depth;
    }
    void* const h_null = nullptr;
    for (int l = l_begin; l < c->depth; ++l) {
        TfLayer& L = w.L[l];
        void* const h_fused = drop_h() ? h_null : L.h;        // fused MLP kernels only: the per-op fc2 GEMM reads h from memory
        const void* const* t = tensors + 11 * l;
        const float *ln1_w = (const float*)t[0], *ln1_b = (const float*)t[1], *out_b = (const float*)t[4], *ln2_w = (const float*)t[5],
                    *ln2_b = (const float*)t[6], *fc1_b = (const float*)t[8], *fc2_b = (const float*)t[10];

        const bool block = !fuse && m3l_attn_block_supported(dt, D, c->heads, n, c->project_out);
        if (block) {
            // the whole attention half of the layer in one launch (short sequences: the MAE encoder)
            if (m3l_attn_block_fwd(D, B, n, x, ln1_w, ln1_b, L.wqkv, L.wo, out_b, ln2_w, ln2_b, LN_EPS, L.xn1, L.qkv, L.o, L.lse, L.x1,
                                   L.xn2, st))
                return 1;
        }
        const bool attn_t = !block && !fuse && c->project_out && m3l_attn_t192_fwd_supported(dt, D, c->heads, n, B);
        if (attn_t) {
            // long sequences: LN1 + QKV + attention of a sample in one launch (the out-proj + LN2 continue in the feed-forward launch)
            if (m3l_attn_t192_fwd(D, B, n, x, ln1_w, ln1_b, L.wqkv, LN_EPS, L.xn1, L.qkv, L.o, L.lse, st)) return 1;
        }
        if (!block && !attn_t && (!fuse || l == 0)) {
            if (m3l_ln_fwd(dt, x, M, D, ln1_w, ln1_b, LN_EPS, L.xn1, nullptr, st)) return 1;
        }
        GemmEpi e = epi0(3 * HD);
        e.out_t = L.qkv;
        if (!block) {
        if (!attn_t) {
        if (m3l_gemm_nt(dt, L.xn1, D, L.wqkv, D, M, 3 * HD, D, &e, st)) return 1;
        if (m3l_attn_fwd(dt, L.qkv, L.o, L.lse, B, n, c->heads, st)) return 1;
        }
        if (c->project_out && !fuse && m3l_attn_tail_mlp_t192_supported(dt, D, HD, mlp, M)) {
            // long sequences: out-proj + residual + LN2 + fc1 + GELU + fc2 + residual in ONE launch per 192-row tile
            if (m3l_attn_tail_mlp_t192_fwd(D, M, mlp, L.o, x, L.wo, out_b, ln2_w, ln2_b, LN_EPS, L.x1, L.xn2, L.w1, fc1_b, L.w2, fc2_b, L.u, h_fused,
                                           L.xout, st))
                return 1;
            x = L.xout;
            continue;
        }
        if (c->project_out && fuse) {
            RowLnEpi r;
            memset(&r, 0, sizeof(r));
            r.bias = out_b; r.res = x; r.x_out = L.x1; r.gamma = ln2_w; r.beta = ln2_b; r.out_t = L.xn2; r.eps = LN_EPS;
            if (m3l_gemm_nt_rowln(dt, ROWLN_FWD, L.o, HD, L.wo, HD, M, D, HD, &r, st)) return 1;
        } else {
            if (c->project_out) {
                e = epi0(D);
                e.bias = out_b;
                if (rb) { e.res_t = x; e.out_t = L.x1; } else { e.res = x; e.out_f32 = L.x1; }
                if (m3l_gemm_nt(dt, L.o, HD, L.wo, HD, M, D, HD, &e, st)) return 1;
            } else {
                if (m3l_axpy_t(dt, x, L.o, (long)M * D, L.x1, st)) return 1;
            }
            if (m3l_ln_fwd(dt, L.x1, M, D, ln2_w, ln2_b, LN_EPS, L.xn2, nullptr, st)) return 1;
        }
        }
        if (block && !(m3l_mlp_t192_short() && m3l_mlp_t192_supported(dt, D, mlp, M)) && m3l_mlp_block_supported(dt, D, mlp, n)) {
            // the feed-forward half in one launch as well
            if (m3l_mlp_block_fwd(D, mlp, B, n, L.xn2, L.x1, L.w1, fc1_b, L.w2, fc2_b, L.u, h_fused, L.xout, st)) return 1;
            x = L.xout;
            continue;
        }
        if (!fuse && m3l_mlp_t192_supported(dt, D, mlp, M)) {
            // long sequences: fc1 + GELU + fc2 + residual per 192-row tile, the hidden activation never leaves the CU between the GEMMs
            if (m3l_mlp_t192_fwd(D, M, mlp, L.xn2, L.x1, L.w1, fc1_b, L.w2, fc2_b, L.u, h_fused, L.xout, st)) return 1;
            x = L.xout;
            continue;
        }
        e = epi0(mlp);
        e.bias = fc1_b; e.act = 1; e.out_pre = L.u; e.out_t = L.h;
        if (m3l_gemm_nt(dt, L.xn2, D, L.w1, D, M, mlp, D, &e, st)) return 1;
        if (fuse) {
            RowLnEpi r;
            memset(&r, 0, sizeof(r));
            r.bias = fc2_b; r.res = L.x1; r.x_out = L.xout; r.eps = LN_EPS;
            if (l + 1 < c->depth) {
                const void* const* tn = tensors + 11 * (l + 1);
                r.gamma = (const float*)tn[0]; r.beta = (const float*)tn[1]; r.out_t = w.L[l + 1].xn1;
            } else {
                r.gamma = (const float*)tfin[0]; r.beta = (const float*)tfin[1];
                r.out_t = dt ? y_t : (y_t ? y_t : nullptr);
                r.out_f32 = y32;
                if (dt == 0 && y_t && y32 == nullptr) { r.out_f32 = (float*)y_t; r.out_t = nullptr; }
                final_done = true;
            }
            if (m3l_gemm_nt_rowln(dt, ROWLN_FWD, L.h, mlp, L.w2, mlp, M, D, mlp, &r, st)) return 1;
        } else {
            e = epi0(D);
            e.bias = fc2_b;
            if (rb) { e.res_t = L.x1; e.out_t = L.xout; } else { e.res = L.x1; e.out_f32 = L.xout; }
            if (m3l_gemm_nt(dt, L.h, mlp, L.w2, mlp, M, D, mlp, &e, st)) return 1;
        }
        x = L.xout;
    }
    if (final_done) return 0;
    return m3l_ln_fwd(dt, x, M, D, (const float*)tfin[0], (const float*)tfin[1], LN_EPS, y_t, y32, st);
}

int m3l_transformer_bwd(const m3l_tf_cfg* c, int B, int n, const float* x_in, const void* const* tensors, void* ws, const void* dy,
                        int dy_dtype, float* dx_in, float* const* grads, void* stream) {
    return m3l_transformer_bwd_range(c, B, n, x_in, tensors, ws, dy, dy_dtype, dx_in, grads, c->depth, 0, stream);
}

// Backward of layers [layer_lo, layer_hi), descending.  layer_hi == depth also runs the final-norm backward (consumes dy);
// layer_lo == 0 writes dx_in.  The running residual gradient lives in the workspace between calls, so a caller can split the
// backward into chunks and start the gradient all-reduce of a chunk while the next one computes.  On return every gradient of
// the layers in the range (and fc2.bias of layer_lo - 1) is ordered on `stream`.
int m3l_transformer_bwd_range(const m3l_tf_cfg* c, int B, int n, const float* x_in, const void* const* tensors, void* ws, const void* dy,
                              int dy_dtype, float* dx_in, float* const* grads, int layer_hi, int layer_lo, void* stream) {
    if (check_tf(c, B, n)) return 1;
    M3L_CHECK(0 <= layer_lo && layer_lo <= layer_hi && layer_hi <= c->depth, "transformer_bwd: bad layer range [%d, %d)", layer_lo, layer_hi);
    M3L_CHECK(dy_dtype == 0 || dy_dtype == c->dtype, "transformer_bwd: dy dtype %d incompatible with compute dtype %d", dy_dtype, c->dtype);
    hipStream_t st = (hipStream_t)stream;
    TfWs w = tf_layout(c, B, n, ws);
    const int M = B * n, D = c->dim, HD = c->heads * 64, mlp = c->mlp_dim, dt = c->dtype;
    // every LayerNorm backward leaves its [G][3 D] partials (dgamma | dbeta | column sums = bias gradient of the Linear that fed the
    // residual branch) in its own slot; ONE batched reduce at the end of the range, on the side stream, replaces 2 small launches per
    // layer on the critical path
    std::vector<ReduceBatch> batches(1);
    batches[0].count = 0;
    batches[0].D = D;
    const int lnG = m3l_ln_bwd_blocks(M);
    auto ln_slot = [&](int id, float* dgamma, float* dbeta, float* dbias, int G = 0) -> float* {
        if (batches.back().count == M3L_REDUCE_BATCH_MAX) {
            batches.emplace_back();
            batches.back().count = 0;
            batches.back().D = D;
        }
        ReduceBatch& rb = batches.back();
        float* slot = w.ln_part + (size_t)id * w.ln_part_stride;
        rb.it[rb.count++] = ReduceBatchItem{slot, {dgamma, dbeta, dbias}, G > 0 ? G : lnG};
        return slot;
    };
    const float* x_last = c->depth ? w.L[c->depth - 1].xout : x_in;
    const bool rb = tf_rb(c, B, n, use_rowln() && m3l_gemm_nt_rowln_supported(dt, D, mlp) && m3l_gemm_nt_rowln_supported(dt, D, 3 * HD));
    RbScope rb_scope(rb);     // bf16 residual stream: x / x1 / xout hold bf16, the running residual gradient lives in the dx_t sets only
    const void* const* tf = tensors + 11 * c->depth;
    float* const* gf = grads + 11 * c->depth;
    // every ln_bwd also emits its result in the compute type (operand of the next GEMMs) and the column sums of it
    // (= bias gradient of the Linear that produced the residual branch): no separate cast / colsum passes.
    const int NS = w.nset;
    if (layer_hi == c->depth) {
        float* db_last = c->depth ? grads[11 * (c->depth - 1) + 10] : nullptr;   // fc2 bias of the last layer
        const int top = c->depth ? (c->depth - 1) % NS : 0;
        if (m3l_ln_bwd(dy_dtype, dy, x_last, M, D, (const float*)tf[0], LN_EPS, nullptr, rb ? nullptr : w.dx, c->depth ? w.dx_t[top] : nullptr, dt,
                       ln_slot(2 * c->depth, gf[0], gf[1], db_last), nullptr, nullptr, nullptr, 0, st))
            return 1;
    }
    if (side_init()) return 2;
    const bool fuse = use_rowln() && m3l_gemm_nt_rowln_supported(dt, D, mlp) && m3l_gemm_nt_rowln_supported(dt, D, 3 * HD);
    const bool wg_inline = wgrad_inline();
    hipStream_t s2 = wg_inline ? st : g_side.s;
    // completion of the wgrad launch that last read operand set i.  A backward split into several range calls (chunks, for the
    // overlap with the gradient all-reduce) carries these across the calls: in deferred-join mode the previous call's weight
    // gradients may still be reading a set when this call reaches it again.
    std::vector<hipEvent_t> set_done(NS, nullptr);
    {
        std::lock_guard<std::mutex> lock(g_side_mu);
        auto it = g_set_done.find(ws);
        if (layer_hi < c->depth && it != g_set_done.end() && (int)it->second.size() == NS) set_done = it->second;
        if (it != g_set_done.end()) g_set_done.erase(it);
    }
    std::vector<hipEvent_t> launched;                 // every wgrad launch of this call (joined at the end)
    std::vector<TnProblem> pend;                      // weight-gradient problems of the layers waiting for the next grouped launch
    std::vector<TnExtra> pend_ex;
    std::vector<int> pend_sets;
    auto flush_wgrads = [&]() -> int {
        if (pend.empty()) return 0;
        if (wg_inline) {
            if (m3l_gemm_tn_grouped(dt, pend.data(), (int)pend.size(), M, w.scratch_tn, w.scratch_tn_b, 0, st, pend_ex.data(), (int)pend_ex.size()))
                return 1;
            pend.clear(); pend_ex.clear(); pend_sets.clear();
            return 0;
        }
        hipEvent_t ready = side_event();
        M3L_HIP(hipEventRecord(ready, st));
        M3L_HIP(hipStreamWaitEvent(s2, ready, 0));
        if (m3l_gemm_tn_grouped(dt, pend.data(), (int)pend.size(), M, w.scratch_tn, w.scratch_tn_b, 0, s2, pend_ex.data(), (int)pend_ex.size()))
            return 1;
        hipEvent_t done = side_event();
        M3L_HIP(hipEventRecord(done, s2));
        for (int sidx : pend_sets) set_done[sidx] = done;
        launched.push_back(done);
        pend.clear(); pend_ex.clear(); pend_sets.clear();
        return 0;
    };
    // before a kernel overwrites operand set i: the weight-gradient launch that read its previous contents must be done
    auto claim_set = [&](int i) -> int {
        if (set_done[i]) {
            M3L_HIP(hipStreamWaitEvent(st, set_done[i], 0));
            set_done[i] = nullptr;
        }
        return 0;
    };
    const int csrows = m3l_gemm_nt_colsum_rows(M, mlp);
    // short sequences whose both backward halves take a block kernel: one launch per weight-gradient group of layers (enc_mega.hip)
    const bool mega_bwd = (m3l_enc_mega_enabled() & 2) && !fuse && c->project_out && m3l_mlp_block_bwd_supported(dt, D, mlp, n) &&
                          !(m3l_mlp_t192_supported(dt, D, mlp, M) && m3l_mlp_t192_short()) && m3l_attn_block_bwd_enabled() &&
                          m3l_attn_block_supported(dt, D, c->heads, n, c->project_out) && w.wg_batch <= M3L_MEGA_BWD_MAX_LAYERS;
    for (int l = layer_hi - 1; l >= layer_lo; --l) {
        if (mega_bwd && pend_sets.empty()) {
            const int g_lo = std::max(layer_lo, l - w.wg_batch + 1), cnt = l - g_lo + 1;
            const void* lay[M3L_MEGA_BWD_MAX_LAYERS][21];
            for (int ll = l; ll >= g_lo; --ll) {
                TfLayer& LL = w.L[ll];
                const int cur = ll % NS, nxt = ll ? (ll - 1) % NS : 0;
                const void* const* t = tensors + 11 * ll;
                float* const* g = grads + 11 * ll;
                if (ll && claim_set(nxt)) return 2;                 // the attention body of layer ll writes dx_t[nxt]
                float* p2 = ln_slot(2 * ll + 1, g[5], g[6], g[4], B);
                float* p1 = ln_slot(2 * ll, g[0], g[1], ll ? grads[11 * (ll - 1) + 10] : nullptr, B);
                const void* v[21] = {w.dx_t[cur], LL.x1, t[5], LL.u, LL.w2T, LL.w1T, w.du[cur], w.dx1_t[cur], w.scratch2[cur], p2,
                                     ll ? w.L[ll - 1].xout : x_in, t[0], LL.qkv, LL.o, LL.lse, LL.woT, LL.wqkvT, w.dqkv[cur],
                                     (ll == 0 && dx_in) ? dx_in : w.dx, ll ? w.dx_t[nxt] : nullptr, p1};
                memcpy(lay[l - ll], v, sizeof(v));
            }
            if (m3l_enc_bwd_mega(D, mlp, B, n, w.dx, &lay[0][0], cnt, LN_EPS, st)) return 1;
            for (int ll = l; ll >= g_lo; --ll) {
                TfLayer& LL = w.L[ll];
                const int cur = ll % NS;
                float* const* g = grads + 11 * ll;
                const int hk = (drop_h() && fused_mlp_kind(c, B, n, fuse)) ? 2 : 0;
                pend.push_back(TnProblem{w.dx_t[cur], hk ? LL.u : LL.h, D, mlp, D, mlp, g[9], mlp, D, mlp, 0, 0, hk});
                pend.push_back(TnProblem{w.du[cur], LL.xn2, mlp, D, mlp, D, g[7], D, mlp, D, 0, 0});
                pend.push_back(TnProblem{w.dqkv[cur], LL.xn1, 3 * HD, D, 3 * HD, D, g[2], D, 3 * HD, D, 0, 0});
                pend.push_back(TnProblem{w.dx1_t[cur], LL.o, D, HD, D, HD, g[3], HD, D, HD, 0, 0});
                pend_ex.push_back(TnExtra{w.scratch2[cur], g[8], B, mlp});
                pend_sets.push_back(cur);
            }
            if (int rc = flush_wgrads()) return rc;
            l = g_lo;
            continue;
        }
        TfLayer& L = w.L[l];
        const int cur = l % NS, nxt = l ? (l - 1) % NS : 0;
        const float* xl = l ? w.L[l - 1].xout : ((rb && !m3l_call_io()) ? reinterpret_cast<const float*>(rb_xin(w)) : x_in);
        const void* const* t = tensors + 11 * l;
        float* const* g = grads + 11 * l;
        // set `cur` (dx_t[cur] was written by layer l+1's last kernel, which claimed the set) receives du / dx1_t / dqkv of this layer
        // ---- feed-forward: x_out = x1 + fc2(gelu(fc1(LN2(x1))))
        GemmEpi e = epi0(mlp);
        const bool mlp_t192 = !fuse && m3l_mlp_t192_supported(dt, D, mlp, M) && (m3l_mlp_t192_short() || !m3l_mlp_block_bwd_supported(dt, D, mlp, n));
        const bool mlp_block = !fuse && !mlp_t192 && m3l_mlp_block_bwd_supported(dt, D, mlp, n);
        int cs_rows = csrows;
        if (mlp_block) {
            // short sequences: du, its column sums, dxn2 and the LN2 backward in one launch; one partial row per sample
            cs_rows = B;
            if (m3l_mlp_block_bwd(D, mlp, B, n, w.dx_t[cur], rb ? nullptr : w.dx, L.x1, (const float*)t[5], L.u, L.w2T, L.w1T, LN_EPS, w.du[cur], w.dx1_t[cur],
                                  w.scratch2[cur], ln_slot(2 * l + 1, g[5], g[6], c->project_out ? g[4] : nullptr, B), st))
                return 1;
        } else if (mlp_t192) {
            // long sequences: the same chain per 192-row tile; one partial row per tile
            cs_rows = m3l_mlp_t192_cs_rows(D, M);
            if (m3l_mlp_t192_bwd(D, M, mlp, w.dx_t[cur], rb ? nullptr : w.dx, L.x1, (const float*)t[5], L.u, L.w2T, L.w1T, LN_EPS, w.du[cur], w.dx1_t[cur],
                                 w.scratch2[cur], ln_slot(2 * l + 1, g[5], g[6], c->project_out ? g[4] : nullptr, m3l_mlp_t192_tiles(D, M)), st))
                return 1;
        } else {
        e.out_t = w.du[cur]; e.gelu_u = L.u; e.colsum_part = w.scratch2[cur];
        if (m3l_gemm_nt(dt, w.dx_t[cur], D, L.w2T, D, M, mlp, D, &e, st)) return 1;                   // du = (dx W2) * gelu'(u)
        // (the per-row-block column sums in scratch2[cur] = fc1 bias gradient partials are reduced by the wgrad group's reduce)
        }
        if (mlp_block || mlp_t192) {
        } else if (fuse) {
            // dxn2 = du W1 and the LN2 backward in one kernel: dx1 = dx + dLN(dxn2) (in place) + compute-type copy + partials
            RowLnEpi r;
            memset(&r, 0, sizeof(r));
            r.x = L.x1; r.gamma = (const float*)t[5]; r.res = w.dx; r.x_out = w.dx; r.out_t = w.dx1_t[cur]; r.part = w.scratch; r.eps = LN_EPS;
            if (m3l_gemm_nt_rowln(dt, ROWLN_BWD, w.du[cur], mlp, L.w1T, mlp, M, D, mlp, &r, st)) return 1;
            if (m3l_reduce_rows_seg3(w.scratch, cdiv(M, 64), D, g[5], g[6], c->project_out ? g[4] : nullptr, 0, st)) return 1;
        } else {
            e = epi0(D);
            e.out_t = w.dxn;
            if (m3l_gemm_nt(dt, w.du[cur], mlp, L.w1T, mlp, M, D, mlp, &e, st)) return 1;             // dxn2 = du W1
            // dx1 = dx + LN2-backward (in place), + compute-type copy, + out-proj bias grad
            // (bf16 residual stream: the residual gradient is dx_t[cur] itself, the result exists in the compute type only)
            if (m3l_ln_bwd(dt, w.dxn, L.x1, M, D, (const float*)t[5], LN_EPS, rb ? reinterpret_cast<const float*>(w.dx_t[cur]) : w.dx,
                           rb ? nullptr : w.dx, w.dx1_t[cur], dt, ln_slot(2 * l + 1, g[5], g[6], c->project_out ? g[4] : nullptr), nullptr,
                           nullptr, nullptr, 0, st))
                return 1;
        }
        // ---- attention: x1 = x + to_out(attn(LN1(x)))
        const bool attn_block = mlp_block && m3l_attn_block_bwd_enabled() && m3l_attn_block_supported(dt, D, c->heads, n, c->project_out);
        float* dx_dst = (l == 0 && dx_in) ? dx_in : w.dx;
        float* db_prev = l ? grads[11 * (l - 1) + 10] : nullptr;                                      // fc2 bias of layer l-1
        if (attn_block) {
            // short sequences: dO, the attention backward, dxn1 and the LN1 backward in one launch.  It writes dx_t[nxt] (the next
            // layer's operand set)
            if (l && claim_set(nxt)) return 2;
            if (rb) {      // the residual gradient is dx1_t itself; the result goes to the next layer's dx_t, or — layer 0 — through d_o to dx_in
                const bool io = m3l_call_io() && dx_in;
                if (m3l_attn_block_bwd(D, B, n, w.dx1_t[cur], nullptr, xl, (const float*)t[0], L.qkv, L.o, L.lse, L.woT, L.wqkvT, LN_EPS, w.dqkv[cur],
                                       nullptr, l ? w.dx_t[nxt] : (io ? (void*)dx_in : rb_out0(w, M, D)), ln_slot(2 * l, g[0], g[1], db_prev, B), st))
                    return 1;
                if (l == 0 && dx_in && !io && m3l_cast_bf16_f32(rb_out0(w, M, D), (long)M * D, dx_in, st)) return 1;
            } else if (m3l_attn_block_bwd(D, B, n, w.dx1_t[cur], w.dx, xl, (const float*)t[0], L.qkv, L.o, L.lse, L.woT, L.wqkvT, LN_EPS, w.dqkv[cur],
                                          dx_dst, l ? w.dx_t[nxt] : nullptr, ln_slot(2 * l, g[0], g[1], db_prev, B), st))
                return 1;
        }
        const void* d_o = w.dx1_t[cur];
        const bool attn_t = !attn_block && !fuse && c->project_out && m3l_attn_t192_fwd_supported(dt, D, c->heads, n, B);
        if (attn_t) {
            // long sequences: dO + both attention-backward passes of a sample in one launch
            if (m3l_attn_t192_bwd(D, B, n, w.dx1_t[cur], L.qkv, L.o, L.lse, L.woT, w.dqkv[cur], st)) return 1;
        }
        if (!attn_block && !attn_t) {
        if (c->project_out) {
            e = epi0(HD);
            e.out_t = w.d_o;
            if (m3l_gemm_nt(dt, w.dx1_t[cur], D, L.woT, D, M, HD, D, &e, st)) return 1;               // do = dx1 Wo
            d_o = w.d_o;
        }
        if (m3l_attn_bwd(dt, L.qkv, L.o, d_o, L.lse, w.dsum, w.dqkv[cur], B, n, c->heads, st)) return 1;
        }
        // ---- the weight gradients of the layer join the pending group; every wg_batch layers (and at the end of the range) the group
        // goes out as ONE grouped TN launch + one reduce on the side stream, overlapping the dgrad chains of the layers below
        {
            // dW2 = dx^T h; when the forward did not save h the kernel stages u and applies the GELU of the kernel that made it
            const int hk = (drop_h() && fused_mlp_kind(c, B, n, fuse)) ? 2 : 0;     // every bf16 kernel evaluates the fitted GELU (form 2)
            pend.push_back(TnProblem{w.dx_t[cur], hk ? L.u : L.h, D, mlp, D, mlp, g[9], mlp, D, mlp, 0, 0, hk});
            pend.push_back(TnProblem{w.du[cur], L.xn2, mlp, D, mlp, D, g[7], D, mlp, D, 0, 0});             // dW1 = du^T xn2
            pend.push_back(TnProblem{w.dqkv[cur], L.xn1, 3 * HD, D, 3 * HD, D, g[2], D, 3 * HD, D, 0, 0}); // dWqkv = dqkv^T xn1
            if (c->project_out) pend.push_back(TnProblem{w.dx1_t[cur], L.o, D, HD, D, HD, g[3], HD, D, HD, 0, 0});   // dWo = dx1^T o
            pend_ex.push_back(TnExtra{w.scratch2[cur], g[8], cs_rows, mlp});                                // fc1 bias gradient
            pend_sets.push_back(cur);
            if ((int)pend_sets.size() >= w.wg_batch || l == layer_lo) {
                if (int rc = flush_wgrads()) return rc;
            }
        }
        // the last kernel of this layer writes dx_t[nxt] (and the next layer then du / dx1_t / dqkv[nxt])
        if (attn_block) continue;
        if (fuse) {
            if (l && claim_set(nxt)) return 2;
            RowLnEpi r;
            memset(&r, 0, sizeof(r));
            r.x = xl; r.gamma = (const float*)t[0]; r.res = w.dx; r.x_out = dx_dst; r.out_t = l ? w.dx_t[nxt] : nullptr;
            r.part = w.scratch; r.eps = LN_EPS;
            if (m3l_gemm_nt_rowln(dt, ROWLN_BWD, w.dqkv[cur], 3 * HD, L.wqkvT, 3 * HD, M, D, 3 * HD, &r, st)) return 1;   // dxn1 + LN1 backward
            if (m3l_reduce_rows_seg3(w.scratch, cdiv(M, 64), D, g[0], g[1], db_prev, 0, st)) return 1;
        } else if (m3l_qkv_bwd_t192_supported(dt, D, 3 * HD, M)) {
            // long sequences: dxn1 and the LN1 backward per 192-row tile, dxn1 never leaves the registers
            if (l && claim_set(nxt)) return 2;
            const int tiles = m3l_qkv_bwd_t192_tiles(D, M);
            if (rb) {
                // dres = dx1_t of this layer (bf16), result only in the compute type: the next layer's dx_t, or — layer 0 — a bf16 scratch
                // (the upper half of the unused w.dx) that is cast to the fp32 gradient of the stack's input
                const bool io = m3l_call_io() && dx_in;           // the caller takes the input gradient as bf16: no cast
                void* out_t = l ? w.dx_t[nxt] : (io ? (void*)dx_in : rb_out0(w, M, D));
                if (m3l_qkv_bwd_t192(D, M, 3 * HD, w.dqkv[cur], xl, (const float*)t[0], L.wqkvT, reinterpret_cast<const float*>(w.dx1_t[cur]), LN_EPS,
                                     nullptr, out_t, ln_slot(2 * l, g[0], g[1], db_prev, tiles), st))
                    return 1;
                if (l == 0 && dx_in && !io && m3l_cast_bf16_f32(rb_out0(w, M, D), (long)M * D, dx_in, st)) return 1;
            } else if (m3l_qkv_bwd_t192(D, M, 3 * HD, w.dqkv[cur], xl, (const float*)t[0], L.wqkvT, w.dx, LN_EPS, dx_dst, l ? w.dx_t[nxt] : nullptr,
                                        ln_slot(2 * l, g[0], g[1], db_prev, tiles), st))
                return 1;
        } else {
            e = epi0(D);
            e.out_t = w.dxn;
            if (m3l_gemm_nt(dt, w.dqkv[cur], 3 * HD, L.wqkvT, 3 * HD, M, D, 3 * HD, &e, st)) return 1;    // dxn1 = dqkv Wqkv
            if (l && claim_set(nxt)) return 2;
            if (rb) {
                const bool io = m3l_call_io() && dx_in;
                void* out_t = l ? w.dx_t[nxt] : (io ? (void*)dx_in : rb_out0(w, M, D));
                if (m3l_ln_bwd(dt, w.dxn, xl, M, D, (const float*)t[0], LN_EPS, reinterpret_cast<const float*>(w.dx1_t[cur]), nullptr, out_t, dt,
                               ln_slot(2 * l, g[0], g[1], db_prev), nullptr, nullptr, nullptr, 0, st))
                    return 1;
                if (l == 0 && dx_in && !io && m3l_cast_bf16_f32(rb_out0(w, M, D), (long)M * D, dx_in, st)) return 1;
            } else if (m3l_ln_bwd(dt, w.dxn, xl, M, D, (const float*)t[0], LN_EPS, w.dx, dx_dst, l ? w.dx_t[nxt] : nullptr, dt,
                                  ln_slot(2 * l, g[0], g[1], db_prev), nullptr, nullptr, nullptr, 0, st))
                return 1;
        }
    }
    // the batched LayerNorm-parameter reduce: after the last ln_bwd of the range, on the side stream behind the weight gradients
    bool side_work = !launched.empty();
    if (batches[0].count > 0 && wg_inline) {
        for (const ReduceBatch& rb : batches)
            if (m3l_reduce_rows_batch(&rb, 0, st)) return 1;
    } else if (batches[0].count > 0) {
        hipEvent_t ln_ready = side_event();
        M3L_HIP(hipEventRecord(ln_ready, st));
        M3L_HIP(hipStreamWaitEvent(g_side.s, ln_ready, 0));
        for (const ReduceBatch& rb : batches)
            if (m3l_reduce_rows_batch(&rb, 0, g_side.s)) return 1;
        side_work = true;
    }
    // join: every gradient of the range is complete before anything later on the caller's stream — now, or (deferred mode) when the
    // caller asks for it.  The side stream is in order, so one event behind its last kernel covers all of them.
    if (side_work) {
        hipEvent_t tail = g_defer_join ? tail_event() : side_event();
        M3L_HIP(hipEventRecord(tail, g_side.s));
        if (g_defer_join) {
            std::lock_guard<std::mutex> lock(g_side_mu);
            g_pending.push_back(tail);
        } else {
            M3L_HIP(hipStreamWaitEvent(st, tail, 0));
        }
    }
    if (layer_lo > 0 && g_defer_join) {                // more ranges of this backward follow: they inherit the set events
        std::lock_guard<std::mutex> lock(g_side_mu);
        g_set_done[ws] = set_done;
    }
    if (c->depth == 0 && dx_in && layer_hi == 0) M3L_HIP(hipMemcpyAsync(dx_in, w.dx, (size_t)M * D * sizeof(float), hipMemcpyDeviceToDevice, st));
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// frozen ViT (inference only): the DINOv2-S/14-reg feature extractor of the cfg-5 fusion head
// (models/pretrain_models_dino_cat_mae.py:886 `self.dino_model(obs_viso)`; train_dino_cat_mae.py:29 torch.hub dinov2_vits14_reg).
// Same kernels as the trainable transformer, no saved activations, weights handed over already in the compute type (they never
// change), qkv WITH bias, caller-chosen LayerNorm eps (1e-6), LayerScale folded into the out-proj / fc2 weights by the caller.
namespace {
struct FvWs {
    void *xn, *qkv, *o, *h;
    float *xa, *xb, *lse;
    size_t total;
};
FvWs fv_layout(const m3l_tf_cfg* c, int B, int n, void* ws) {
    Arena a(ws);
    FvWs w;
    const size_t M = (size_t)B * n, e = esz(c->dtype), HD = (size_t)c->heads * 64;
    w.xn = a.take(M * c->dim * e);
    w.qkv = a.take(M * 3 * HD * e);
    w.o = a.take(M * HD * e);
    w.h = a.take(M * c->mlp_dim * e);
    w.xa = a.take_n<float>(M * c->dim);
    w.xb = a.take_n<float>(M * c->dim);
    w.lse = a.take_n<float>((size_t)B * c->heads * n);
    w.total = a.off + 256;
    return w;
}
}  // namespace

size_t m3l_frozen_vit_ws_bytes(const m3l_tf_cfg* c, int B, int n) {
    if (check_tf(c, B, n)) return 0;
    return fv_layout(c, B, n, nullptr).total;
}

int m3l_frozen_vit_fwd(const m3l_tf_cfg* c, float ln_eps, int B, int n, const float* x_in, const void* const* tensors, void* ws,
                       float* y32, void* stream) {
    if (check_tf(c, B, n)) return 1;
    M3L_CHECK(c->project_out, "frozen_vit: the block always has an output projection");
    M3L_CHECK(ln_eps > 0.f && x_in && tensors && ws && y32, "frozen_vit: null argument / eps");
    hipStream_t st = (hipStream_t)stream;
    FvWs w = fv_layout(c, B, n, ws);
    const int M = B * n, D = c->dim, HD = c->heads * 64, mlp = c->mlp_dim, dt = c->dtype;
    const float* x = x_in;
    for (int l = 0; l < c->depth; ++l) {
        const void* const* t = tensors + 12 * l;
        float* x1 = w.xa;                          // x -> x1 (attention branch) -> x2 (MLP branch); x is x_in or xb
        float* x2 = w.xb;
        if (m3l_ln_fwd(dt, x, M, D, (const float*)t[0], (const float*)t[1], ln_eps, w.xn, nullptr, st)) return 1;
        GemmEpi e = epi0(3 * HD);
        e.bias = (const float*)t[3]; e.out_t = w.qkv;
        if (m3l_gemm_nt(dt, w.xn, D, t[2], D, M, 3 * HD, D, &e, st)) return 1;
        if (m3l_attn_fwd(dt, w.qkv, w.o, w.lse, B, n, c->heads, st)) return 1;
        e = epi0(D);
        e.bias = (const float*)t[5]; e.res = x; e.out_f32 = x1;
        if (m3l_gemm_nt(dt, w.o, HD, t[4], HD, M, D, HD, &e, st)) return 1;
        if (m3l_ln_fwd(dt, x1, M, D, (const float*)t[6], (const float*)t[7], ln_eps, w.xn, nullptr, st)) return 1;
        e = epi0(mlp);
        e.bias = (const float*)t[9]; e.act = 1; e.out_t = w.h;
        if (m3l_gemm_nt(dt, w.xn, D, t[8], D, M, mlp, D, &e, st)) return 1;
        e = epi0(D);
        e.bias = (const float*)t[11]; e.res = x1; e.out_f32 = x2;
        if (m3l_gemm_nt(dt, w.h, mlp, t[10], mlp, M, D, mlp, &e, st)) return 1;
        x = x2;
    }
    const void* const* tf = tensors + 12 * c->depth;
    return m3l_ln_fwd(0, x, M, D, (const float*)tf[0], (const float*)tf[1], ln_eps, nullptr, y32, st);
}

// ---------------------------------------------------------------------------------------------------------------
// encoder -> decoder glue
namespace {
struct UnWs {
    void *w, *wT, *dsrc_t;
    float *proj, *dsrc;
    float* scratch;
    size_t scratch_b, total;
};
UnWs un_layout(int D, int dd, int dtype, int B, int nvis, void* ws) {
    Arena a(ws);
    UnWs w;
    const size_t e = esz(dtype), Mv = (size_t)B * nvis;
    w.w = a.take((size_t)dd * D * e);
    w.wT = a.take((size_t)D * dd * e);
    w.proj = a.take_n<float>(Mv * dd);
    w.dsrc = a.take_n<float>(Mv * dd);
    w.dsrc_t = a.take(Mv * dd * e);
    std::vector<std::pair<int, int>> shapes = {{dd, D}};
    w.scratch_b = scratch_bytes((int)Mv, shapes, std::max(D, dd * (2 + M3L_MAX_TACTILES)));
    w.scratch = reinterpret_cast<float*>(a.take(w.scratch_b));
    w.total = a.off + 256;
    return w;
}
}  // namespace

size_t m3l_unshuffle_ws_bytes(const m3l_geom* g, int D, int dd, int dtype, int B, int nvis, int nmask) {
    (void)g; (void)nmask;
    return un_layout(D, dd, dtype, B, nvis, nullptr).total;
}

int m3l_unshuffle_fwd(const m3l_geom* g, int D, int dd, int dtype, int B, int nvis, int nmask, const int64_t* unmasked,
                      const int64_t* masked, const float* enc32, const void* enc_t, const void* const* tensors, void* ws,
                      float* dec_in, void* stream) {
    if (check_geom(g)) return 1;
    M3L_CHECK(dd % 8 == 0 && D % 8 == 0, "unshuffle: dims must be multiples of 8");
    hipStream_t st = (hipStream_t)stream;
    const Geo ge = geo_of(g);
    M3L_CHECK(nvis + nmask == ge.n_img + ge.k * ge.n_tac, "unshuffle: nvis + nmask = %d != number of patches", nvis + nmask);
    UnWs w = un_layout(D, dd, dtype, B, nvis, ws);
    const float* e2d_w = (const float*)tensors[0];
    const float* src = enc32;
    if (e2d_w) {
        WeightPack pk;
        memset(&pk, 0, sizeof(pk));
        pk.d[0] = WeightDesc{e2d_w, w.w, w.wT, dd, D, D, dd};
        pk.count = 1;
        if (m3l_prep_weights(dtype, &pk, st)) return 1;
        if (m3l_prep_mode() == 1) return 0;        // collect pass of a call chain (mae_step.hip)
        GemmEpi e = epi0(dd);
        e.bias = (const float*)tensors[1];
        e.out_f32 = w.proj;
        if (m3l_gemm_nt(dtype, enc_t, D, w.w, D, B * nvis, dd, D, &e, st)) return 1;
        src = w.proj;
    } else {
        M3L_CHECK(D == dd, "unshuffle: enc_to_dec weight missing but encoder dim %d != decoder dim %d", D, dd);
        if (m3l_prep_mode() == 1) return 0;
    }
    return k_unshuffle_fwd(src, (const float*)tensors[2], unmasked, nvis, masked, nmask, B, dd, ge.n_img, ge.n_tac,
                             (const float*)tensors[3], (const float*)tensors[4], (const float*)tensors[5], dec_in, st);
}

int m3l_unshuffle_bwd(const m3l_geom* g, int D, int dd, int dtype, int B, int nvis, int nmask, const int64_t* unmasked,
                      const int64_t* masked, const void* enc_t, const void* const* tensors, void* ws, const float* d_dec_in,
                      void* d_enc, int* d_enc_dtype, float* const* grads, void* stream) {
    if (check_geom(g)) return 1;
    hipStream_t st = (hipStream_t)stream;
    const Geo ge = geo_of(g);
    UnWs w = un_layout(D, dd, dtype, B, nvis, ws);
    const bool proj = tensors[0] != nullptr;
    float* dsrc = proj ? w.dsrc : (float*)d_enc;
    if (k_unshuffle_bwd(d_dec_in, unmasked, nvis, masked, nmask, B, dd, ge.n_img, ge.n_tac, 1 + ge.k, dsrc, w.scratch, grads[2],
                          grads[3], 0, st))
        return 1;
    if (!proj) {
        if (d_enc_dtype) *d_enc_dtype = 0;
        return 0;
    }
    const int Mv = B * nvis;
    if (m3l_cast_f32(dtype, w.dsrc, (long)Mv * dd, w.dsrc_t, st)) return 1;
    if (m3l_gemm_tn(dtype, w.dsrc_t, dd, enc_t, D, Mv, dd, D, w.scratch, w.scratch_b, grads[0], D, dd, D, 0, st)) return 1;
    if (m3l_colsum(0, w.dsrc, Mv, dd, dd, w.scratch, grads[1], 0, st)) return 1;
    GemmEpi e = epi0(D);
    e.out_t = d_enc;
    if (m3l_gemm_nt(dtype, w.dsrc_t, dd, w.wT, dd, Mv, D, dd, &e, st)) return 1;
    if (d_enc_dtype) *d_enc_dtype = dtype;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// heads + masked MSE
namespace {
struct HeadGroupWs {
    void *w, *wT, *dg, *dpred, *dpred_s, *ddg;
    float* pred;
};
struct HeadWs {
    HeadGroupWs g[2];
    float* loss_part;
    float* scratch;
    float* scratch_side;     // as EmbWs::scratch_side
    size_t scratch_b, total;
};
HeadWs head_layout(const Geo& ge, int dd, int dtype, int B, int nmask, void* ws) {
    Arena a(ws);
    HeadWs w;
    const size_t e = esz(dtype);
    const int pdp[2] = {ge.pdp_img, ge.pdp_tac};
    const size_t rows = (size_t)B * nmask;      // capacity per group (each group uses a share)
    for (int i = 0; i < 2; ++i) {
        w.g[i].w = a.take((size_t)pdp[i] * dd * e);
        w.g[i].wT = a.take((size_t)dd * pdp[i] * e);
        w.g[i].dg = a.take(rows * dd * e);
        w.g[i].pred = a.take_n<float>(rows * pdp[i]);
        w.g[i].dpred = a.take(rows * pdp[i] * e);
        w.g[i].dpred_s = a.take(rows * pdp[i] * e);
        w.g[i].ddg = a.take(rows * dd * e);
    }
    w.loss_part = a.take_n<float>(2 * M3L_MAX_PARTIAL_BLOCKS);
    std::vector<std::pair<int, int>> shapes = {{ge.pdp_img, dd}, {ge.pdp_tac, dd}};
    w.scratch_b = scratch_bytes((int)std::max(rows, (size_t)1), shapes, std::max(dd, std::max(ge.pdp_img, ge.pdp_tac)));
    w.scratch = reinterpret_cast<float*>(a.take(w.scratch_b));
    w.scratch_side = reinterpret_cast<float*>(a.take(w.scratch_b));
    w.total = a.off + 256;
    return w;
}
}  // namespace

size_t m3l_heads_ws_bytes(const m3l_geom* g, int dd, int dtype, int B, int nmask) {
    if (check_geom(g)) return 0;
    return head_layout(geo_of(g), dd, dtype, B, nmask, nullptr).total;
}

int m3l_heads_loss_fwd(const m3l_geom* g, int dd, int dtype, int B, int N, int nmask, int nm_img, const int64_t* masked,
                       const float* image, const float* const* tactiles, const void* dec_t, const void* const* tensors, void* ws,
                       float* loss, float* pred_img, float* tgt_img, float* pred_tac, float* tgt_tac, void* stream) {
    return m3l_heads_loss_fwd2(g, dd, dtype, B, N, nmask, nm_img, masked, image, tactiles, dec_t, tensors, ws, loss, nullptr, pred_img,
                               tgt_img, pred_tac, tgt_tac, stream);
}

// loss_parts (optional, f32[2]): the two WEIGHTED terms of the loss separately: mse(image), 10 * mse(tactile)
int m3l_heads_loss_fwd2(const m3l_geom* g, int dd, int dtype, int B, int N, int nmask, int nm_img, const int64_t* masked,
                        const float* image, const float* const* tactiles, const void* dec_t, const void* const* tensors, void* ws,
                        float* loss, float* loss_parts, float* pred_img, float* tgt_img, float* pred_tac, float* tgt_tac, void* stream) {
    if (check_geom(g)) return 1;
    hipStream_t st = (hipStream_t)stream;
    const Geo ge = geo_of(g);
    HeadWs w = head_layout(ge, dd, dtype, B, nmask, ws);
    const int cnt[2] = {nm_img, nmask - nm_img}, j0[2] = {0, nm_img};
    const int pd[2] = {ge.pd_img, ge.pd_tac}, pdp[2] = {ge.pdp_img, ge.pdp_tac};
    const float weight[2] = {1.f, 10.f};
    const PatchGroup pgs[2] = {group_img(g, ge, image), group_tac(g, ge, tactiles)};
    float* tgt[2] = {tgt_img, tgt_tac};
    float* pred_out[2] = {pred_img, pred_tac};
    int nparts = 0;
    int part_beg[2] = {0, 0}, part_cnt[2] = {0, 0};
    {
        WeightPack pk;                       // both heads' compute-type weight copies in one launch
        memset(&pk, 0, sizeof(pk));
        for (int i = 0; i < 2; ++i) {
            if (cnt[i] == 0) continue;
            pk.d[pk.count++] = WeightDesc{(const float*)tensors[2 * i], w.g[i].w, w.g[i].wT, pd[i], dd, dd, pdp[i]};
        }
        if (m3l_prep_weights(dtype, &pk, st)) return 1;
        if (m3l_prep_mode() == 1) return 0;        // collect pass of a call chain (mae_step.hip)
    }
    if (m3l_gather_rows2(dtype, dec_t, N, dd, masked, nmask, cnt[0], cnt[1], B, w.g[0].dg, w.g[1].dg, st)) return 1;   // both groups' rows
    for (int i = 0; i < 2; ++i) {
        if (cnt[i] == 0) continue;
        const int rows = B * cnt[i];
        GemmEpi e = epi0(pdp[i]);
        e.bias = (const float*)tensors[2 * i + 1];
        e.n_bias = pd[i];
        e.out_f32 = w.g[i].pred;
        if (m3l_gemm_nt(dtype, w.g[i].dg, dd, w.g[i].w, dd, rows, pdp[i], dd, &e, st)) return 1;
        int nb = 0;
        if (m3l_mse(dtype, w.g[i].pred, pdp[i], &pgs[i], masked, nmask, j0[i], cnt[i], B, weight[i], w.loss_part + nparts, &nb,
                    w.g[i].dpred, tgt[i], st))
            return 1;
        part_beg[i] = nparts; part_cnt[i] = nb;
        nparts += nb;
        if (pred_out[i])
            M3L_HIP(hipMemcpy2DAsync(pred_out[i], (size_t)pd[i] * 4, w.g[i].pred, (size_t)pdp[i] * 4, (size_t)pd[i] * 4, rows,
                                     hipMemcpyDeviceToDevice, st));
    }
    if (loss_parts) {
        M3L_HIP(hipMemsetAsync(loss_parts, 0, 2 * sizeof(float), st));
        for (int i = 0; i < 2; ++i)
            if (part_cnt[i] && m3l_reduce_rows(w.loss_part + part_beg[i], part_cnt[i], 1, 1, loss_parts + i, 0, st)) return 1;
    }
    return m3l_reduce_rows(w.loss_part, nparts, 1, 1, loss, 0, st);
}

int m3l_heads_loss_bwd(const m3l_geom* g, int dd, int dtype, int B, int N, int nmask, int nm_img, const int64_t* masked,
                       const void* const* tensors, void* ws, const float* dloss, void* d_dec, float* const* grads, void* stream) {
    if (check_geom(g)) return 1;
    (void)tensors;
    hipStream_t st = (hipStream_t)stream;
    const Geo ge = geo_of(g);
    HeadWs w = head_layout(ge, dd, dtype, B, nmask, ws);
    const int cnt[2] = {nm_img, nmask - nm_img}, j0[2] = {0, nm_img};
    const int pd[2] = {ge.pd_img, ge.pd_tac}, pdp[2] = {ge.pdp_img, ge.pdp_tac};
    M3L_HIP(hipMemsetAsync(d_dec, 0, (size_t)B * N * dd * esz(dtype), st));
    SideSection side;
    for (int i = 0; i < 2; ++i) {
        if (cnt[i] == 0) continue;
        const int rows = B * cnt[i];
        // d loss / d pred for dloss = 1 was written by the forward.  The upstream gradient dloss (a device scalar) scales everything
        // linearly: on the critical path it is folded into the scatter below (the dgrad GEMM reads the unscaled dpred); the weight /
        // bias gradients take a scaled copy, made where they run (side stream in deferred-join mode)
        hipStream_t s2;
        if (side.fork(st, &s2)) return 2;
        const void* dpred = w.g[i].dpred;
        if (dloss) {
            if (m3l_scale_by_dev(dtype, w.g[i].dpred, (long)rows * pdp[i], dloss, w.g[i].dpred_s, s2)) return 1;
            dpred = w.g[i].dpred_s;
        }
        float* sc = side.on ? w.scratch_side : w.scratch;
        if (m3l_gemm_tn(dtype, dpred, pdp[i], w.g[i].dg, dd, rows, pdp[i], dd, sc, w.scratch_b, grads[2 * i], dd, pd[i], dd, 0, s2)) return 1;
        if (m3l_colsum(dtype, dpred, rows, pd[i], pdp[i], sc, grads[2 * i + 1], 0, s2)) return 1;
        GemmEpi e = epi0(dd);
        e.out_t = w.g[i].ddg;
        if (m3l_gemm_nt(dtype, w.g[i].dpred, pdp[i], w.g[i].wT, pdp[i], rows, dd, pdp[i], &e, st)) return 1;
    }
    if (m3l_scatter_rows2(dtype, w.g[0].ddg, w.g[1].ddg, N, dd, masked, nmask, cnt[0], cnt[1], B, dloss, d_dec, st)) return 1;
    return side.end();
}

// ---------------------------------------------------------------------------------------------------------------
// EarlyCNN stem
namespace {
struct ConvL { int Ci, Co, KH, S, P, H, W, OH, OW, K, Kpad; long M; };
struct CnnWs {
    ConvL L[4];
    void *col[3], *act[3], *w[4], *wT[4];
    void *dout_t, *dpreA, *dpreB, *dcol;
    void *wf[3], *wd[3];     // direct convolutions (conv.hip): forward / weight-gradient and input-gradient weight copies
    float* slab;             // ... and the weight-gradient partial sums
    bool direct;
    float* scratch;
    size_t scratch_b, total;
};
// direct (implicit-GEMM) convolutions for the three strided / 3x3 layers of the stem instead of im2col + GEMM: bf16, the stems of
// dim 64 / 128 / 256; env M3L_DIRECT_CONV=0 keeps the im2col path (A/B, and the reference point of the parity tests)
int g_direct_conv = -1;
bool cnn_direct(const m3l_cnn_cfg* c) {
    if (g_direct_conv < 0) g_direct_conv = getenv("M3L_DIRECT_CONV") ? (atoi(getenv("M3L_DIRECT_CONV")) > 0 ? 1 : 0) : 1;
    if (!g_direct_conv || c->dtype != 1) return false;
    const int D = c->dim, co[3] = {D / 8, D / 4, D / 2};
    int ci = c->in_channels;
    for (int l = 0; l < 3; ++l) {
        const bool k3 = l == 2 && c->tactile;
        if (!m3l_conv_direct_supported(c->dtype, ci, co[l], k3 ? 3 : 4, k3 ? 1 : 2, 1, l == 0)) return false;
        ci = co[l];
    }
    return true;
}
CnnWs cnn_layout(const m3l_cnn_cfg* c, int Btot, void* ws) {
    Arena a(ws);
    CnnWs w;
    const int D = c->dim;
    const int co[4] = {D / 8, D / 4, D / 2, D};
    int ci = c->in_channels, H = c->height, W = c->width;
    for (int l = 0; l < 4; ++l) {
        ConvL& L = w.L[l];
        L.Ci = ci; L.Co = co[l]; L.H = H; L.W = W;
        if (l == 3) { L.KH = 1; L.S = 1; L.P = 0; }
        else if (l == 2 && c->tactile) { L.KH = 3; L.S = 1; L.P = 1; }
        else { L.KH = 4; L.S = 2; L.P = 1; }
        L.OH = (H + 2 * L.P - L.KH) / L.S + 1;
        L.OW = (W + 2 * L.P - L.KH) / L.S + 1;
        L.K = L.Ci * L.KH * L.KH;
        L.Kpad = pad8(L.K);
        L.M = (long)Btot * L.OH * L.OW;
        ci = L.Co; H = L.OH; W = L.OW;
    }
    const size_t e = esz(c->dtype);
    size_t max_pre = 0, max_col = 0, max_slab = 0;
    w.direct = cnn_direct(c);
    for (int l = 0; l < 4; ++l) {
        const ConvL& L = w.L[l];
        w.w[l] = a.take((size_t)L.Co * L.Kpad * e);
        w.wT[l] = a.take((size_t)L.Kpad * L.Co * e);
        if (l < 3) {
            w.col[l] = w.direct ? nullptr : a.take((size_t)L.M * L.Kpad * e);       // the column matrices exist only on the im2col path
            w.act[l] = a.take((size_t)L.M * L.Co * e);
            max_col = std::max(max_col, (size_t)L.M * L.Kpad);
            w.wf[l] = w.wd[l] = nullptr;
            if (w.direct) {
                w.wf[l] = a.take(m3l_conv_wf_elems(L.Ci, L.Co, L.KH) * 2);
                if (l) w.wd[l] = a.take(m3l_conv_wd_elems(L.Ci, L.Co, L.KH, L.S) * 2);
                max_slab = std::max(max_slab, m3l_conv_wgrad_slab_elems(Btot, L.Ci, L.H, L.W, L.Co, L.KH, L.S, L.P));
            }
        }
        max_pre = std::max(max_pre, (size_t)L.M * L.Co);
    }
    w.dout_t = a.take((size_t)w.L[3].M * D * e);
    w.dpreA = a.take(max_pre * e);
    w.dpreB = a.take(max_pre * e);
    w.dcol = w.direct ? nullptr : a.take(max_col * e);
    w.slab = w.direct ? reinterpret_cast<float*>(a.take(max_slab * sizeof(float))) : nullptr;
    std::vector<std::pair<int, int>> shapes;
    long Mmax = 1;
    for (int l = 0; l < 4; ++l) { shapes.push_back({w.L[l].Co, w.L[l].Kpad}); Mmax = std::max(Mmax, w.L[l].M); }
    w.scratch_b = 0;
    for (int l = 0; l < 4; ++l) w.scratch_b = std::max(w.scratch_b, m3l_gemm_tn_ws_bytes((int)w.L[l].M, w.L[l].Co, w.L[l].Kpad, nullptr));
    w.scratch_b = std::max(w.scratch_b, (size_t)M3L_MAX_PARTIAL_BLOCKS * D * sizeof(float));
    w.scratch = reinterpret_cast<float*>(a.take(w.scratch_b));
    w.total = a.off + 256;
    return w;
}
int check_cnn(const m3l_cnn_cfg* c, int B, int nsrc) {
    M3L_CHECK(c->dtype == 0 || c->dtype == 1, "earlycnn: bad dtype");
    M3L_CHECK(c->dim % 64 == 0, "earlycnn: dim=%d must be a multiple of 64", c->dim);
    M3L_CHECK(B > 0 && nsrc >= 1 && nsrc <= M3L_MAX_SENSORS, "earlycnn: B=%d nsrc=%d", B, nsrc);
    M3L_CHECK(c->height % 8 == 0 && c->width % 8 == 0 || (c->tactile && c->height % 4 == 0 && c->width % 4 == 0), "earlycnn: image size");
    return 0;
}
}  // namespace

int m3l_set_direct_conv(int on) {
    if (g_direct_conv < 0) g_direct_conv = getenv("M3L_DIRECT_CONV") ? (atoi(getenv("M3L_DIRECT_CONV")) > 0 ? 1 : 0) : 1;
    const int old = g_direct_conv;
    g_direct_conv = on ? 1 : 0;
    return old;
}

size_t m3l_earlycnn_ws_bytes(const m3l_cnn_cfg* c, int B, int nsrc) {
    if (check_cnn(c, B, nsrc)) return 0;
    return cnn_layout(c, B * nsrc, nullptr).total;
}

int m3l_earlycnn_fwd(const m3l_cnn_cfg* c, int B, int nsrc, const float* const* srcs, const void* const* tensors, void* ws, float* out,
                     void* stream) {
    if (check_cnn(c, B, nsrc)) return 1;
    hipStream_t st = (hipStream_t)stream;
    const int Btot = B * nsrc, dt = c->dtype;
    CnnWs w = cnn_layout(c, Btot, ws);
    for (int l = 0; l < 4; ++l) {
        const ConvL& L = w.L[l];
        WeightPack pk;
        memset(&pk, 0, sizeof(pk));
        pk.d[0] = WeightDesc{(const float*)tensors[2 * l], w.w[l], w.wT[l], L.Co, L.K, L.Kpad, L.Co};
        pk.count = 1;
        if (m3l_prep_weights(dt, &pk, st)) return 1;
    }
    if (m3l_prep_mode() == 1) return 0;            // collect pass of a call chain (mae_step.hip)
    ConvSrc cs;
    memset(&cs, 0, sizeof(cs));
    for (int i = 0; i < nsrc; ++i) cs.src[i] = srcs[i];
    cs.nsrc = nsrc; cs.nchw = 1;
    for (int l = 0; l < 3; ++l) {
        const ConvL& L = w.L[l];
        if (w.direct) {
            // direct convolution (conv.hip): the input patch of an 8 x 8 output tile staged once in LDS, no column matrix
            if (m3l_conv_prep((const float*)tensors[2 * l], L.Ci, L.Co, L.KH, L.S, w.wf[l], w.wd[l], st)) return 1;
            if (m3l_conv_fwd(&cs, B, Btot, L.Ci, L.H, L.W, L.Co, L.KH, L.S, L.P, w.wf[l], (const float*)tensors[2 * l + 1], w.act[l], st)) return 1;
            memset(&cs, 0, sizeof(cs));
            cs.src[0] = w.act[l]; cs.nsrc = 1; cs.nchw = 0;
            continue;
        }
        if (m3l_im2col(dt, &cs, l == 0 ? B : Btot, L.Ci, L.H, L.W, L.KH, L.S, L.P, L.OH, L.OW, L.Kpad, w.col[l], st)) return 1;
        GemmEpi e = epi0(L.Co);
        e.bias = (const float*)tensors[2 * l + 1];
        e.act = 2;
        e.out_t = w.act[l];
        if (m3l_gemm_nt(dt, w.col[l], L.Kpad, w.w[l], L.Kpad, (int)L.M, L.Co, L.Kpad, &e, st)) return 1;
        memset(&cs, 0, sizeof(cs));
        cs.src[0] = w.act[l]; cs.nsrc = 1; cs.nchw = 0;
    }
    const ConvL& L3 = w.L[3];
    GemmEpi e = epi0(L3.Co);
    e.bias = (const float*)tensors[7];
    e.out_f32 = out;
    return m3l_gemm_nt(dt, w.act[2], L3.Kpad, w.w[3], L3.Kpad, (int)L3.M, L3.Co, L3.Kpad, &e, st);
}

int m3l_earlycnn_bwd(const m3l_cnn_cfg* c, int B, int nsrc, const float* const* srcs, const void* const* tensors, void* ws, const float* dout,
                     float* const* grads, void* stream) {
    if (check_cnn(c, B, nsrc)) return 1;
    (void)tensors;
    hipStream_t st = (hipStream_t)stream;
    const int Btot = B * nsrc, dt = c->dtype, D = c->dim;
    CnnWs w = cnn_layout(c, Btot, ws);
    const ConvL& L3 = w.L[3];
    // conv4 (1x1): dW4 = dout^T a2, db4 = colsum(dout), d(pre3) = (dout W4) masked by a2 > 0
    if (m3l_cast_f32(dt, dout, L3.M * D, w.dout_t, st)) return 1;
    if (m3l_gemm_tn(dt, w.dout_t, D, w.act[2], L3.Kpad, (int)L3.M, D, L3.Kpad, w.scratch, w.scratch_b, grads[6], L3.K, D, L3.K, 0, st)) return 1;
    if (m3l_colsum(0, dout, (int)L3.M, D, D, w.scratch, grads[7], 0, st)) return 1;
    void* dpre = w.dpreA;
    void* dnext = w.dpreB;
    {
        GemmEpi e = epi0(L3.Kpad);
        e.out_t = dpre; e.relu_ref = w.act[2];
        if (m3l_gemm_nt(dt, w.dout_t, D, w.wT[3], D, (int)L3.M, L3.Kpad, D, &e, st)) return 1;   // [M, Co_2] (Kpad_3 == Co_2)
    }
    for (int l = 2; l >= 0; --l) {
        const ConvL& L = w.L[l];
        if (w.direct) {
            M3L_CHECK(srcs != nullptr, "earlycnn_bwd: the direct convolutions read the stem's input frames again (srcs)");
            ConvSrc cs;
            memset(&cs, 0, sizeof(cs));
            if (l == 0) {
                for (int i = 0; i < nsrc; ++i) cs.src[i] = srcs[i];
                cs.nsrc = nsrc; cs.nchw = 1;
            } else {
                cs.src[0] = w.act[l - 1]; cs.nsrc = 1; cs.nchw = 0;
            }
            if (m3l_conv_wgrad(&cs, B, Btot, L.Ci, L.H, L.W, L.Co, L.KH, L.S, L.P, dpre, w.slab, grads[2 * l], st)) return 1;
            if (m3l_colsum(dt, dpre, (int)L.M, L.Co, L.Co, w.scratch, grads[2 * l + 1], 0, st)) return 1;
            if (l == 0) break;
            if (m3l_conv_dgrad(dpre, Btot, L.Ci, L.H, L.W, L.Co, L.KH, L.S, L.P, w.wd[l], w.act[l - 1], dnext, st)) return 1;
            std::swap(dpre, dnext);
            continue;
        }
        if (m3l_gemm_tn(dt, dpre, L.Co, w.col[l], L.Kpad, (int)L.M, L.Co, L.Kpad, w.scratch, w.scratch_b, grads[2 * l], L.K, L.Co, L.K, 0, st))
            return 1;
        if (m3l_colsum(dt, dpre, (int)L.M, L.Co, L.Co, w.scratch, grads[2 * l + 1], 0, st)) return 1;
        if (l == 0) break;
        GemmEpi e = epi0(L.Kpad);
        e.out_t = w.dcol;
        if (m3l_gemm_nt(dt, dpre, L.Co, w.wT[l], L.Co, (int)L.M, L.Kpad, L.Co, &e, st)) return 1;  // dcol = dpre W_l
        if (m3l_col2im_relu(dt, w.dcol, Btot, L.Ci, L.H, L.W, L.KH, L.S, L.P, L.OH, L.OW, L.Kpad, w.act[l - 1], dnext, st)) return 1;
        std::swap(dpre, dnext);
    }
    return 0;
}

size_t m3l_tokens_assemble_ws_bytes(const m3l_geom* g, int D) {
    return (size_t)M3L_MAX_PARTIAL_BLOCKS * (1 + (g->num_tactiles > 0 ? g->num_tactiles : 0)) * D * sizeof(float) + 256;
}
int m3l_tokens_assemble_fwd(const m3l_geom* g, int D, int B, const float* img_tok, const float* tac_tok, const void* const* tensors,
                            float* tokens, void* stream) {
    if (check_geom(g)) return 1;
    const Geo ge = geo_of(g);
    return k_tokens_assemble(img_tok, tac_tok, B, D, ge.n_img, ge.n_tac, ge.k, (const float*)tensors[0], (const float*)tensors[1],
                             (const float*)tensors[2], tokens, (hipStream_t)stream);
}
int m3l_tokens_assemble_bwd(const m3l_geom* g, int D, int B, const float* dtokens, float* d_img, float* d_tac, void* ws, float* dmod,
                            void* stream) {
    if (check_geom(g)) return 1;
    const Geo ge = geo_of(g);
    // slots of absent modalities get exact zeros from the reduction (their partial sums are zero)
    float* part = (float*)ws;
    // dmod has (1 + num_tactiles) rows in the model; this call fills rows [0, 1 + k) and the caller pre-zeroes the rest
    return k_tokens_assemble_bwd(dtokens, B, D, ge.n_img, ge.n_tac, ge.k, d_img, d_tac, part, dmod, 0, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// stand-alone ops
size_t m3l_layernorm_ws_bytes(int D) { return (size_t)2048 * 3 * D * sizeof(float) + 256; }

int m3l_layernorm_fwd(int out_dtype, const float* x, int M, int D, const float* gamma, const float* beta, float eps, void* y,
                      float* y32, void* stream) {
    return m3l_ln_fwd(out_dtype, x, M, D, gamma, beta, eps, y, y32, (hipStream_t)stream);
}
int m3l_layernorm_bwd(int dy_dtype, const void* dy, const float* x, int M, int D, const float* gamma, float eps, const float* dres,
                      float* dx, void* ws, float* dgamma, float* dbeta, void* stream) {
    return m3l_ln_bwd(dy_dtype, dy, x, M, D, gamma, eps, dres, dx, nullptr, dy_dtype, (float*)ws, dgamma, dbeta, nullptr, 0, (hipStream_t)stream);
}
int m3l_gather_tokens(const float* src, int B, int N, int D, const int64_t* idx, int K, float* dst, void* stream) {
    return m3l_gather_rows(0, src, N, D, idx, K, 0, K, B, dst, (hipStream_t)stream);
}
int m3l_scatter_tokens(const float* src, int B, int N, int D, const int64_t* idx, int K, float* dst_zeroed, void* stream) {
    return m3l_scatter_rows(0, src, N, D, idx, K, 0, K, B, dst_zeroed, (hipStream_t)stream);
}
int m3l_vt_load(const float* image_nhwc, int B, int H, int W, int C, float* image_nchw, const float* tactile, int th, int tw,
                int n_sensors, int frame_stack, float* const* tactile_out, void* stream) {
    return m3l_vt_load_launch(image_nhwc, 0, B, H, W, C, 0.f, 1.f, image_nchw, tactile, 0, th, tw, n_sensors, frame_stack, -1.f, 2.f, tactile_out,
                              (hipStream_t)stream);
}
int m3l_vt_load2(const void* image_nhwc, int image_u8, int B, int H, int W, int C, double img_lo, double img_hi, float* image_nchw,
                 const void* tactile, int tactile_u8, int th, int tw, int n_sensors, int frame_stack, double tac_lo, double tac_hi,
                 float* const* tactile_out, void* stream) {
    // utils/pretrain_utils.py:28-30,47-49: `tensor - lo` rounds the Python double lo to fp32; `/ (hi - lo)` forms the span in double
    // and rounds it to fp32 once
    return m3l_vt_load_launch(image_nhwc, image_u8, B, H, W, C, (float)img_lo, (float)(img_hi - img_lo), image_nchw, tactile, tactile_u8, th, tw,
                              n_sensors, frame_stack, (float)tac_lo, (float)(tac_hi - tac_lo), tactile_out, (hipStream_t)stream);
}

int m3l_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int step, void* stream) {
    return m3l_adam_flat(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, 1.0f, (hipStream_t)stream);
}
// the same with the gradients read as grad_scale * grads[i] (data parallel: the 1 / world of a SUM all-reduce folded into the update —
// no separate pass over the gradient buffer)
int m3l_adam_step_scaled(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1, float beta2,
                         float eps, float weight_decay, int step, float grad_scale, void* stream) {
    return m3l_adam_flat(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, (hipStream_t)stream);
}

int m3l_adamw_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, float grad_scale, float max_grad_norm, float* norm_ws, int scale_grads, void* stream) {
    return m3l_adamw_flat(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, max_grad_norm, norm_ws,
                          scale_grads, (hipStream_t)stream);
}

int m3l_adam_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                      float beta2, float eps, float weight_decay, int* step_dev, float* bias_corr_dev, void* stream) {
    return m3l_adam_flat_dev(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step_dev, bias_corr_dev,
                             (hipStream_t)stream);
}

int m3l_op_gemm_nt(int dtype, const void* A, int lda, const void* W, int ldw, int M, int N, int K, const float* bias,
                   const float* res, float* out_f32, void* out_t, void* out_pre, const void* gelu_u, int act, int ldc, void* stream) {
    GemmEpi e = epi0(ldc);
    e.bias = bias; e.res = res; e.out_f32 = out_f32; e.out_t = out_t; e.out_pre = out_pre; e.gelu_u = gelu_u; e.act = act;
    return m3l_gemm_nt(dtype, A, lda, W, ldw, M, N, K, &e, (hipStream_t)stream);
}
size_t m3l_op_gemm_tn_ws_bytes(int M, int N, int K) { return m3l_gemm_tn_ws_bytes(M, N, K, nullptr); }
int m3l_op_gemm_tn(int dtype, const void* Y, int ldy, const void* X, int ldx, int M, int N, int K, void* ws, size_t ws_bytes,
                   float* out, int ldo, void* stream) {
    return m3l_gemm_tn(dtype, Y, ldy, X, ldx, M, N, K, (float*)ws, ws_bytes, out, ldo, N, K, 0, (hipStream_t)stream);
}
size_t m3l_op_gemm_tn_grouped_ws_bytes(int dtype, int count, int M, const int* N, const int* K) {
    (void)dtype;
    if (count < 1 || count > M3L_TN_MAX_PROBLEMS || M <= 0 || !N || !K) return 0;
    TnProblem pr[M3L_TN_MAX_PROBLEMS];
    memset(pr, 0, sizeof(pr));
    for (int i = 0; i < count; ++i) { pr[i].N = N[i]; pr[i].K = K[i]; }
    return m3l_gemm_tn_grouped_ws_bytes(M, pr, count);
}
int m3l_op_gemm_tn_grouped(int dtype, int count, int M, const void* const* Y, const int* ldy, const void* const* X, const int* ldx, const int* N,
                           const int* K, float* const* out, void* ws, size_t ws_bytes, void* stream) {
    M3L_CHECK(count >= 1 && count <= M3L_TN_MAX_PROBLEMS && Y && X && ldy && ldx && N && K && out && ws, "gemm_tn_grouped: bad arguments (count %d)", count);
    TnProblem pr[M3L_TN_MAX_PROBLEMS];
    memset(pr, 0, sizeof(pr));
    for (int i = 0; i < count; ++i) {
        pr[i].Y = Y[i]; pr[i].X = X[i]; pr[i].ldy = ldy[i]; pr[i].ldx = ldx[i]; pr[i].N = N[i]; pr[i].K = K[i];
        pr[i].out = out[i]; pr[i].ldo = K[i]; pr[i].nvalid = N[i]; pr[i].kvalid = K[i];
    }
    return m3l_gemm_tn_grouped(dtype, pr, count, M, (float*)ws, ws_bytes, 0, (hipStream_t)stream);
}
// ---- building blocks of the cfg-5 fusion head's trainable MLP (models/pretrain_models_dino_cat_mae.py:828-836,899-903) ----
int m3l_op_colsum(int dtype, const void* Y, int M, int N, int ld, void* ws, float* out, void* stream) {
    return m3l_colsum(dtype, Y, M, N, ld, (float*)ws, out, 0, (hipStream_t)stream);
}
size_t m3l_op_colsum_ws_bytes(int N) { return (size_t)M3L_MAX_PARTIAL_BLOCKS * N * sizeof(float) + 256; }
int m3l_op_prep_weight(int dtype, const float* src, int rows, int cols, void* dst, void* dstT, void* stream) {
    WeightPack pk;
    memset(&pk, 0, sizeof(pk));
    pk.d[0] = WeightDesc{src, dst, dstT, rows, cols, cols, rows};
    pk.count = 1;
    return m3l_prep_weights(dtype, &pk, (hipStream_t)stream);
}
int m3l_op_mask_scale(int mode, const float* src, const void* ref, float scale, long count, float* out, void* stream) {
    return m3l_mask_scale(mode, src, ref, scale, count, out, (hipStream_t)stream);
}
int m3l_op_concat2(float* a, int na, float* b, int nb, int rows, float* cat, int split, void* stream) {
    return m3l_concat2(a, na, b, nb, rows, cat, split, (hipStream_t)stream);
}
// patch extraction of a P x P / stride-P patch-embed convolution (the frozen DINOv2's `patch_embed.proj`): x NCHW f32 -> col [B gh gw, Kpad]
// in the Conv2d weight's K order (c, kh, kw), compute type, pad columns zero; and the encoder's token assembly
int m3l_op_patch_cols(int dtype, const float* x, int B, int C, int H, int W, int P, int Kpad, void* col, void* stream) {
    M3L_CHECK(x && col && B > 0 && P > 0 && H % P == 0 && W % P == 0 && Kpad >= C * P * P, "patch_cols: bad arguments");
    ConvSrc cs;
    memset(&cs, 0, sizeof(cs));
    cs.src[0] = x; cs.nsrc = 1; cs.nchw = 1;
    return m3l_im2col(dtype, &cs, B, C, H, W, P, P, 0, H / P, W / P, Kpad, col, (hipStream_t)stream);
}
int m3l_op_vit_tokens(const float* emb, const float* cls, const float* regs, const float* pos, int B, int npatch, int R, int D, float* tok,
                      void* stream) {
    return m3l_vit_tokens(emb, cls, regs, pos, B, npatch, R, D, tok, (hipStream_t)stream);
}
int m3l_op_attn_fwd(int dtype, const void* qkv, void* o, float* lse, int B, int n, int H, void* stream) {
    return m3l_attn_fwd(dtype, qkv, o, lse, B, n, H, (hipStream_t)stream);
}
int m3l_op_attn_bwd(int dtype, const void* qkv, const void* o, const void* dO, const float* lse, float* dsum, void* dqkv, int B,
                    int n, int H, void* stream) {
    return m3l_attn_bwd(dtype, qkv, o, dO, lse, dsum, dqkv, B, n, H, (hipStream_t)stream);
}

}  // extern "C"
