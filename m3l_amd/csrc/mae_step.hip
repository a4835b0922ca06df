// The whole MAE training step as TWO host calls (forward, backward): the module plans of mae_plan.hip chained in C.
//
// Reference call site: `loss = self.mae(x); loss.backward()` (models/ppo_mae.py:262-263, models/sac_mae.py:284-291) over
// VTMAE.forward (models/pretrain_models.py:146-342).  The per-module entry points (m3l_embed_* / m3l_transformer_* / m3l_unshuffle_* /
// m3l_heads_loss_*) are each driven by one torch.autograd.Function; at cfg 2 the ten Python hops, their argument marshalling and
// workspace allocations cost 1.8-2.7 ms of host time per 4 ms step.  Here Python makes two ctypes calls per step: every intermediate
// activation (tokens, encoder / decoder outputs, their gradients) lives in ONE caller-provided workspace, the backward issues the
// chunked transformer backwards itself and — data parallel — hands each finished prefix of the flat gradient buffer to
// m3l_comm_allreduce (comm.hip) with the same bucket rule m3l_amd.parallel.GradSync applies.
// Same kernels, same launch order, same workspaces per module as the per-module path: results are bit-identical to it.
#include <string.h>

#include <algorithm>

#include "../../include/m3l_amd.h"
#include "common.cuh"
#include "kernels.h"

namespace {

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
};

struct StepDims {
    int n_img, n_tac, k, N, nmask, nvis, nm_img, nm_tac, nvis_img;
    int D, dd, dt;
};

struct StepWs {
    float *tokens, *enc32, *dec_in, *dec32, *d_dec_in, *dtokens;
    void *enc_t, *dec_t, *d_dec, *d_enc;
    void *ws_embed, *ws_enc, *ws_glue, *ws_dec, *ws_heads;
    size_t total;
};

int step_dims(const m3l_mae_cfg* c, StepDims* d) {
    int cnt[6];
    if (m3l_mask_counts(&c->geom, c->masking_ratio, cnt)) return 1;
    const m3l_geom& g = c->geom;
    d->n_img = g.use_vision ? cnt[4] : 0;
    d->k = (g.use_tactile && g.num_tactiles > 0) ? g.num_tactiles : 0;
    d->n_tac = d->k ? cnt[5] : 0;
    d->nmask = cnt[0];
    d->nvis = cnt[1];
    d->N = cnt[0] + cnt[1];
    d->nm_img = cnt[2];
    d->nm_tac = cnt[3];
    d->nvis_img = d->n_img - d->nm_img;
    d->D = c->enc.dim;
    d->dd = c->dec.dim;
    d->dt = c->enc.dtype;
    M3L_CHECK(c->enc.dtype == c->dec.dtype, "mae_step: encoder / decoder compute types differ (%d / %d)", c->enc.dtype, c->dec.dtype);
    M3L_CHECK(d->nvis > 0 && d->nmask > 0, "mae_step: masking leaves %d visible / %d masked tokens", d->nvis, d->nmask);
    return 0;
}

StepWs step_layout(const m3l_mae_cfg* c, const StepDims& d, int B, void* ws) {
    Arena a(ws);
    StepWs w;
    const size_t e = d.dt ? 2 : 4, Mv = (size_t)B * d.nvis, Ma = (size_t)B * d.N;
    w.tokens = (float*)a.take(Mv * d.D * 4);
    w.enc32 = (float*)a.take(Mv * d.D * 4);
    w.enc_t = d.dt ? a.take(Mv * d.D * e) : nullptr;
    w.dec_in = (float*)a.take(Ma * d.dd * 4);
    w.dec32 = d.dt ? nullptr : (float*)a.take(Ma * d.dd * 4);
    w.dec_t = d.dt ? a.take(Ma * d.dd * e) : nullptr;
    w.d_dec = a.take(Ma * d.dd * e);
    w.d_dec_in = (float*)a.take(Ma * d.dd * 4);
    w.d_enc = a.take(Mv * d.D * 4);
    w.dtokens = (float*)a.take(Mv * d.D * 4);
    w.ws_embed = a.take(m3l_embed_ws_bytes(&c->geom, d.D, d.dt, B, d.nvis));
    w.ws_enc = a.take(m3l_transformer_ws_bytes(&c->enc, B, d.nvis));
    w.ws_glue = a.take(m3l_unshuffle_ws_bytes(&c->geom, d.D, d.dd, d.dt, B, d.nvis, d.nmask));
    w.ws_dec = a.take(m3l_transformer_ws_bytes(&c->dec, B, d.N));
    w.ws_heads = a.take(m3l_heads_ws_bytes(&c->geom, d.dd, d.dt, B, d.nmask));
    w.total = a.off + 256;
    return w;
}

// tensor groups inside the flat `tensors` / `grads` arrays
struct Groups { int embed, enc, glue, dec, heads, total; };
Groups groups_of(const m3l_mae_cfg* c) {
    Groups g;
    g.embed = 0;
    g.enc = 15;
    g.glue = g.enc + 11 * c->enc.depth + 2;
    g.dec = g.glue + 6;
    g.heads = g.dec + 11 * c->dec.depth + 2;
    g.total = g.heads + 4;
    return g;
}

}  // namespace

extern "C" {

int m3l_mae_step_num_tensors(const m3l_mae_cfg* c) { return groups_of(c).total; }

size_t m3l_mae_step_ws_bytes(const m3l_mae_cfg* c, int B) {
    StepDims d;
    if (B <= 0 || step_dims(c, &d)) return 0;
    return step_layout(c, d, B, nullptr).total;
}

int m3l_mae_step_fwd(const m3l_mae_cfg* c, int B, const float* image, const float* const* tactiles, const float* const* noise,
                     const void* const* tensors, void* ws, float* loss, int64_t* masked, int64_t* unmasked, void* stream) {
    StepDims d;
    M3L_CHECK(B > 0 && tensors && ws && loss && noise && masked && unmasked, "mae_step_fwd: null argument / B=%d", B);
    if (step_dims(c, &d)) return 1;
    hipStream_t st = (hipStream_t)stream;
    const StepWs w = step_layout(c, d, B, ws);
    const Groups g = groups_of(c);
    // mask sampling (pretrain_models.py:223-248) -> the caller's index lists (the backward reads them again)
    if (m3l_mask_sample_counts(&c->geom, d.nm_img, d.nm_tac, B, noise, masked, unmasked, st)) return 1;
    // patch embed of the visible tokens (:157-216,255-256)
    if (m3l_embed_fwd(&c->geom, d.D, d.dt, B, d.nvis, d.nvis_img, unmasked, image, tactiles, tensors + g.embed, w.ws_embed, w.tokens, st)) return 1;
    // encoder (:266)
    if (m3l_transformer_fwd(&c->enc, B, d.nvis, w.tokens, tensors + g.enc, w.ws_enc, w.enc_t, w.enc32, st)) return 1;
    // enc_to_dec + un-shuffle + decoder positions (:270-307); f32 compute: the "compute-type" encoder output is the f32 one
    if (m3l_unshuffle_fwd(&c->geom, d.D, d.dd, d.dt, B, d.nvis, d.nmask, unmasked, masked, w.enc32, d.dt ? w.enc_t : (void*)w.enc32,
                          tensors + g.glue, w.ws_glue, w.dec_in, st))
        return 1;
    // decoder (:309)
    if (m3l_transformer_fwd(&c->dec, B, d.N, w.dec_in, tensors + g.dec, w.ws_dec, w.dec_t, w.dec32, st)) return 1;
    // heads + masked MSE (:260-262,327-340)
    return m3l_heads_loss_fwd2(&c->geom, d.dd, d.dt, B, d.N, d.nmask, d.nm_img, masked, image, tactiles, d.dt ? w.dec_t : (void*)w.dec32,
                               tensors + g.heads, w.ws_heads, loss, nullptr, nullptr, nullptr, nullptr, nullptr, st);
}

// Backward of the step.  grads: one f32 pointer per tensor (same order; NULL where the tensor has no gradient).  dloss: device scalar
// or NULL (= 1).  comm (may be NULL = no communication): the flat gradient buffer and, per finished stage, the end of the prefix of it
// that is final — heads, each decoder chunk (top-down), glue, each encoder chunk, embed — as GradSync lays it out.
int m3l_mae_step_bwd(const m3l_mae_cfg* c, int B, const float* image, const float* const* tactiles, const int64_t* masked,
                     const int64_t* unmasked, const void* const* tensors, void* ws, const float* dloss, float* const* grads,
                     const m3l_comm_plan* comm, void* stream) {
    StepDims d;
    M3L_CHECK(B > 0 && tensors && ws && grads && masked && unmasked, "mae_step_bwd: null argument / B=%d", B);
    if (step_dims(c, &d)) return 1;
    hipStream_t st = (hipStream_t)stream;
    const StepWs w = step_layout(c, d, B, ws);
    const Groups g = groups_of(c);
    const int chunk = (comm && comm->layers_per_chunk > 0) ? comm->layers_per_chunk : 0;
    long sent = 0;
    int stage = 0;
    // a stage's gradients are final: the finished prefix travels once it holds min_bucket elements (few, large collectives: xGMI rings
    // are latency-bound below a few MB), or at the very end
    auto stage_done = [&]() -> int {
        if (!comm) return 0;
        M3L_CHECK(stage < comm->n_stages, "mae_step_bwd: comm plan has %d stages, the backward reached stage %d", comm->n_stages, stage);
        const long end = comm->stage_end[stage++];
        M3L_CHECK(end >= sent && end <= comm->total, "mae_step_bwd: comm plan stage end %ld outside [%ld, %ld]", end, sent, comm->total);
        if (end > sent && (end - sent >= comm->min_bucket || end == comm->total)) {
            if (m3l_comm_allreduce(comm->flat + sent, (size_t)(end - sent), stream)) return 1;
            sent = end;
        }
        return 0;
    };
    auto tf_bwd = [&](const m3l_tf_cfg* cfg, int n, const float* x_in, const void* const* t, void* tws, const void* dy, int dy_code,
                      float* dx, float* const* gr) -> int {
        if (!chunk || chunk >= cfg->depth) {
            if (m3l_transformer_bwd(cfg, B, n, x_in, t, tws, dy, dy_code, dx, gr, stream)) return 1;
            return stage_done();
        }
        for (int hi = cfg->depth; hi > 0;) {
            const int lo = std::max(0, hi - chunk);
            if (m3l_transformer_bwd_range(cfg, B, n, x_in, t, tws, dy, dy_code, dx, gr, hi, lo, stream)) return 1;
            if (stage_done()) return 1;
            hi = lo;
        }
        return 0;
    };
    if (m3l_heads_loss_bwd(&c->geom, d.dd, d.dt, B, d.N, d.nmask, d.nm_img, masked, tensors + g.heads, w.ws_heads, dloss, w.d_dec,
                           grads + g.heads, st))
        return 1;
    if (stage_done()) return 1;
    if (tf_bwd(&c->dec, d.N, w.dec_in, tensors + g.dec, w.ws_dec, w.d_dec, d.dt, w.d_dec_in, grads + g.dec)) return 1;
    int enc_code = 0;
    if (m3l_unshuffle_bwd(&c->geom, d.D, d.dd, d.dt, B, d.nvis, d.nmask, unmasked, masked, d.dt ? w.enc_t : (void*)w.enc32,
                          tensors + g.glue, w.ws_glue, w.d_dec_in, w.d_enc, &enc_code, grads + g.glue, st))
        return 1;
    if (stage_done()) return 1;
    if (tf_bwd(&c->enc, d.nvis, w.tokens, tensors + g.enc, w.ws_enc, w.d_enc, enc_code, w.dtokens, grads + g.enc)) return 1;
    if (m3l_embed_bwd(&c->geom, d.D, d.dt, B, d.nvis, d.nvis_img, unmasked, image, tactiles, tensors + g.embed, w.ws_embed, w.dtokens,
                      grads + g.embed, st))
        return 1;
    if (stage_done()) return 1;
    if (comm) {
        M3L_CHECK(stage == comm->n_stages, "mae_step_bwd: comm plan has %d stages, the backward ran %d", comm->n_stages, stage);
        if (comm->sent_out) *comm->sent_out = sent;
    }
    return 0;
}

}  // extern "C"
