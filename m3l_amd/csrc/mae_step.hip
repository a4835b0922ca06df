// The whole MAE training step as TWO host calls (forward, backward), and the policy-side consumer of the MAE as two more: the module
// plans of mae_plan.hip chained in C.
//
// Reference call sites: `loss = self.mae(x); loss.backward()` (models/ppo_mae.py:262-263, models/sac_mae.py:284-291) over
// VTMAE.forward (models/pretrain_models.py:146-342), and `MAEExtractor.forward` (models/pretrain_models.py:819-841: get_embeddings ->
// 1-layer Transformer -> mean over tokens; every environment step at B = number of envs, every PPO minibatch with grad,
// models/ppo_mae.py:280).  The per-module entry points (m3l_embed_* / m3l_earlycnn_* / m3l_transformer_* / m3l_unshuffle_* /
// m3l_heads_loss_*) are each driven by one torch.autograd.Function; at cfg 2 the ten Python hops, their argument marshalling and
// workspace allocations cost 1.8-2.7 ms of host time per 4 ms step.  Here Python makes two ctypes calls per step: every intermediate
// activation (tokens, encoder / decoder outputs, their gradients) lives in ONE caller-provided workspace, the backward issues the
// chunked transformer backwards itself and — data parallel — hands each finished prefix of the flat gradient buffer to
// m3l_comm_allreduce (comm.hip) with the same bucket rule m3l_amd.parallel.GradSync applies.
// Same kernels, same launch order, same workspaces per module as the per-module path: results are bit-identical to it.
//
// Round 4: both front ends (the patch embed, and the EarlyCNN stems of early_conv_masking=True — the reference's default flag,
// train.py:62 — with their loss over ALL patches, pretrain_models.py:180-191,311-322) and both position modes (fixed sincos buffers, or
// learned tables whose gradient is the batch sum of the token gradients, :218-219,280-287).
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

// identity index rows (the early-conv loss runs the masked-patch kernels over every patch) and the token mean of the extractor
__global__ void iota_rows_kernel(int64_t* out, long total, int N) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) out[i] = i % N;
}
// out[b][d] = mean over the n tokens of y[b][:, d]   (torch.mean(dim=1), pretrain_models.py:838); one block per sample
__global__ void mean_tokens_kernel(const float* __restrict__ y, int n, int D, float* __restrict__ out) {
    const int b = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float s = 0.f;
        for (int t = 0; t < n; ++t) s += y[((long)b * n + t) * D + d];
        out[(long)b * D + d] = s / (float)n;
    }
}
__global__ void mean_tokens_bwd_kernel(const float* __restrict__ dout, int n, int D, float* __restrict__ dy) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i < (long)n * D) dy[(long)b * n * D + i] = dout[(long)b * D + (i % D)] / (float)n;
}

struct IoScope {          // m3l_call_io for the calls inside the scope
    explicit IoScope(bool on) { m3l_set_call_io(on ? 1 : 0); }
    ~IoScope() { m3l_set_call_io(0); }
};

struct StepDims {
    int n_img, n_tac, k, N, nmask, nvis, nm_img, nm_tac, nvis_img;
    int D, dd, dt;
};

// front end of the encoder: patch embed (15 tensors) or EarlyCNN stems (8 + 8 + 3 tensors)
struct FrontWs {
    void* ws_embed;                          // patch embed
    float *img_tok, *tac_tok, *d_img, *d_tac, *tok_all, *d_all;     // stems: per-modality stem tokens, all N tokens, their gradients
    void *ws_cnn_img, *ws_cnn_tac, *ws_asm;
};
struct StepWs {
    float *tokens, *enc32, *dec_in, *dec32, *d_dec_in, *dtokens;
    void *enc_t, *dec_t, *d_dec, *d_enc;
    FrontWs f;
    void *ws_enc, *ws_glue, *ws_dec, *ws_heads;
    int64_t* all_rows;                       // early conv: identity index rows (B, N)
    size_t total;
};

int front_tensors(const m3l_mae_cfg* c) { return c->early_conv ? 19 : 15; }
m3l_cnn_cfg cnn_cfg(const m3l_mae_cfg* c, bool tactile) {
    const m3l_geom& g = c->geom;
    m3l_cnn_cfg k;
    k.in_channels = tactile ? g.tactile_channels : g.image_channels;
    k.height = tactile ? g.tactile_h : g.image_h;
    k.width = tactile ? g.tactile_w : g.image_w;
    k.dim = c->enc.dim;
    k.tactile = tactile ? 1 : 0;
    k.dtype = c->enc.dtype;
    return k;
}

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
    if (c->early_conv) {
        // the stems of the reference downsample by 8 (image) / 4 (tactile): one stem token per patch position (pretrain_models.py:37-56)
        M3L_CHECK(!g.use_vision || (g.image_patch == 8 && g.image_h % 8 == 0 && g.image_w % 8 == 0), "mae_step: the image EarlyCNN stem needs patch size 8");
        M3L_CHECK(d->k == 0 || (g.tactile_patch == 4 && g.tactile_h % 4 == 0 && g.tactile_w % 4 == 0), "mae_step: the tactile EarlyCNN stem needs patch size 4");
    }
    return 0;
}

// front-end buffers for L tokens per sample out of N (L == N: no gather)
void front_layout(Arena& a, const m3l_mae_cfg* c, const StepDims& d, int B, int L, bool learned_scatter, FrontWs* f) {
    memset(f, 0, sizeof(*f));
    const size_t Ma = (size_t)B * d.N;
    if (c->early_conv) {
        const m3l_cnn_cfg ci = cnn_cfg(c, false), ct = cnn_cfg(c, true);
        if (d.n_img) {
            f->img_tok = (float*)a.take((size_t)B * d.n_img * d.D * 4);
            f->d_img = (float*)a.take((size_t)B * d.n_img * d.D * 4);
            f->ws_cnn_img = a.take(m3l_earlycnn_ws_bytes(&ci, B, 1));
        }
        if (d.k) {
            f->tac_tok = (float*)a.take((size_t)B * d.k * d.n_tac * d.D * 4);
            f->d_tac = (float*)a.take((size_t)B * d.k * d.n_tac * d.D * 4);
            f->ws_cnn_tac = a.take(m3l_earlycnn_ws_bytes(&ct, B, d.k));
        }
        f->ws_asm = a.take(m3l_tokens_assemble_ws_bytes(&c->geom, d.D));
        if (L != d.N) {
            f->tok_all = (float*)a.take(Ma * d.D * 4);
            f->d_all = (float*)a.take(Ma * d.D * 4);
        }
    } else {
        f->ws_embed = a.take(m3l_embed_ws_bytes(&c->geom, d.D, d.dt, B, L));
        if (learned_scatter && L != d.N) f->d_all = (float*)a.take(Ma * d.D * 4);     // learned positions: token gradients back at their positions
    }
}

StepWs step_layout(const m3l_mae_cfg* c, const StepDims& d, int B, void* ws) {
    Arena a(ws);
    StepWs w;
    memset(&w, 0, sizeof(w));
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
    front_layout(a, c, d, B, d.nvis, c->learned_pos != 0, &w.f);
    w.ws_enc = a.take(m3l_transformer_ws_bytes(&c->enc, B, d.nvis));
    w.ws_glue = a.take(m3l_unshuffle_ws_bytes(&c->geom, d.D, d.dd, d.dt, B, d.nvis, d.nmask));
    w.ws_dec = a.take(m3l_transformer_ws_bytes(&c->dec, B, d.N));
    const int nrows = c->early_conv ? d.N : d.nmask;       // early conv: loss over all patches
    w.ws_heads = a.take(m3l_heads_ws_bytes(&c->geom, d.dd, d.dt, B, nrows));
    if (c->early_conv) w.all_rows = (int64_t*)a.take(Ma * sizeof(int64_t));
    w.total = a.off + 256;
    return w;
}

// tensor groups inside the flat `tensors` / `grads` arrays
struct Groups { int embed, enc, glue, dec, heads, total; };
Groups groups_of(const m3l_mae_cfg* c) {
    Groups g;
    g.embed = 0;
    g.enc = front_tensors(c);
    g.glue = g.enc + 11 * c->enc.depth + 2;
    g.dec = g.glue + 6;
    g.heads = g.dec + 11 * c->dec.depth + 2;
    g.total = g.heads + 4;
    return g;
}

// weight copies of a whole chain in one launch (elementwise.hip: m3l_prep_set_mode); env M3L_PREP_BATCH=0: every module issues its own
static int prep_first_pass() {
    static const int on = getenv("M3L_PREP_BATCH") ? (atoi(getenv("M3L_PREP_BATCH")) > 0 ? 1 : 0) : 1;
    return on ? 1 : 2;
}
static int prep_mode_of(int pass) { return prep_first_pass() == 1 ? pass : 0; }
// ---- front end forward: -> tokens (B, L, D); idx (B, L) positions of the tokens, or NULL with L == N (all of them, in order)
int front_fwd(const m3l_mae_cfg* c, const StepDims& d, const FrontWs& f, int B, int L, int cnt_img, const int64_t* idx, const float* image,
              const float* const* tactiles, const void* const* t, float* tokens, hipStream_t st) {
    if (!c->early_conv) return m3l_embed_fwd(&c->geom, d.D, d.dt, B, L, cnt_img, idx, image, tactiles, t, f.ws_embed, tokens, st);
    const m3l_cnn_cfg ci = cnn_cfg(c, false), ct = cnn_cfg(c, true);
    if (d.n_img && m3l_earlycnn_fwd(&ci, B, 1, &image, t, f.ws_cnn_img, f.img_tok, st)) return 1;
    if (d.k && m3l_earlycnn_fwd(&ct, B, d.k, tactiles, t + 8, f.ws_cnn_tac, f.tac_tok, st)) return 1;
    if (m3l_prep_mode() == 1) return 0;            // collect pass: the stems have recorded their weight copies
    float* all = idx ? f.tok_all : tokens;
    if (m3l_tokens_assemble_fwd(&c->geom, d.D, B, f.img_tok, f.tac_tok, t + 16, all, st)) return 1;
    if (idx) return m3l_gather_tokens(all, B, d.N, d.D, idx, L, tokens, st);
    return 0;
}
// ---- front end backward: dtokens (B, L, D) -> parameter gradients (grads[i] NULL: not wanted).  Learned positions: grads of the two
// position tables = sum over the batch of the token gradients at each position.
int front_bwd(const m3l_mae_cfg* c, const StepDims& d, const FrontWs& f, int B, int L, int cnt_img, const int64_t* idx, const float* image,
              const float* const* tactiles, const void* const* t, const float* dtokens, float* const* grads, hipStream_t st) {
    const int ipos = c->early_conv ? 17 : 13;                 // index of pos_img in the group (pos_tac follows)
    const bool want_pos = c->learned_pos && (grads[ipos] || grads[ipos + 1]);
    const float* dense = dtokens;                              // (B, N, D) token gradients at their positions
    if (idx && (c->early_conv || want_pos)) {
        M3L_HIP(hipMemsetAsync(f.d_all, 0, (size_t)B * d.N * d.D * 4, st));
        if (m3l_scatter_tokens(dtokens, B, d.N, d.D, idx, L, f.d_all, st)) return 1;
        dense = f.d_all;
    }
    if (want_pos) {
        const int stride = d.N * d.D;
        if (grads[ipos] && d.n_img && m3l_reduce_rows(dense, B, stride, d.n_img * d.D, grads[ipos], 0, st)) return 1;
        if (grads[ipos + 1] && d.k && m3l_reduce_rows(dense + (size_t)d.n_img * d.D, B, stride, d.k * d.n_tac * d.D, grads[ipos + 1], 0, st)) return 1;
    }
    if (!c->early_conv) return m3l_embed_bwd(&c->geom, d.D, d.dt, B, L, cnt_img, idx, image, tactiles, t, f.ws_embed, dtokens, grads, st);
    const m3l_cnn_cfg ci = cnn_cfg(c, false), ct = cnn_cfg(c, true);
    // rows of the modality table that belong to sensors absent from this call keep a zero gradient
    if (grads[16]) M3L_HIP(hipMemsetAsync(grads[16], 0, (size_t)(1 + c->geom.num_tactiles) * d.D * 4, st));
    if (m3l_tokens_assemble_bwd(&c->geom, d.D, B, dense, f.d_img, f.d_tac, f.ws_asm, grads[16], st)) return 1;
    if (d.k && m3l_earlycnn_bwd(&ct, B, d.k, tactiles, t + 8, f.ws_cnn_tac, f.d_tac, grads + 8, st)) return 1;
    if (d.n_img && m3l_earlycnn_bwd(&ci, B, 1, &image, t, f.ws_cnn_img, f.d_img, grads, st)) return 1;
    return 0;
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
    // every module's compute-type weight copies in one launch per 64 matrices: a collect pass over the chain (each module records its
    // matrices and returns), the flush, then the chain itself with the copies in place (elementwise.hip: m3l_prep_set_mode)
    struct PrepReset { ~PrepReset() { m3l_prep_set_mode(0); } } prep_reset;
    for (int pass = prep_first_pass(); pass <= 2; ++pass) {
    m3l_prep_set_mode(prep_mode_of(pass));
    if (pass == 2 && m3l_prep_flush(st)) return 1;
    // patch embed of the visible tokens (:157-216,255-256) / EarlyCNN stems over the frames, then the visible gather (:180-191)
    if (front_fwd(c, d, w.f, B, d.nvis, d.nvis_img, unmasked, image, tactiles, tensors + g.embed, w.tokens, st)) return 1;
    // encoder (:266)
    if (m3l_transformer_fwd(&c->enc, B, d.nvis, w.tokens, tensors + g.enc, w.ws_enc, w.enc_t, w.enc32, st)) return 1;
    // enc_to_dec + un-shuffle + decoder positions (:270-307); f32 compute: the "compute-type" encoder output is the f32 one.  When the
    // decoder runs the bf16 residual stream, its input (and, in the backward, the gradient of it) is exchanged as bf16: no boundary casts
    const IoScope dec_io(m3l_transformer_rb(&c->dec, B, d.N) && !c->learned_pos);      // (the same rule as the backward's)
    if (m3l_unshuffle_fwd(&c->geom, d.D, d.dd, d.dt, B, d.nvis, d.nmask, unmasked, masked, w.enc32, d.dt ? w.enc_t : (void*)w.enc32,
                          tensors + g.glue, w.ws_glue, w.dec_in, st))
        return 1;
    // decoder (:309)
    if (m3l_transformer_fwd(&c->dec, B, d.N, w.dec_in, tensors + g.dec, w.ws_dec, w.dec_t, w.dec32, st)) return 1;
    m3l_set_call_io(0);
    if (pass == 1) {       // the heads' copies (their other arguments play no part in the collect pass)
        if (m3l_heads_loss_fwd2(&c->geom, d.dd, d.dt, B, d.N, c->early_conv ? d.N : d.nmask, c->early_conv ? d.n_img : d.nm_img, masked, image, tactiles,
                                d.dt ? w.dec_t : (void*)w.dec32, tensors + g.heads, w.ws_heads, loss, nullptr, nullptr, nullptr, nullptr, nullptr, st))
            return 1;
    }
    }
    m3l_prep_set_mode(prep_mode_of(2));      // (the heads below run with their copies in place)
    // heads + masked MSE (:260-262,327-340); early conv: every patch is predicted and scored (:311-322)
    const int64_t* rows = masked;
    int nrows = d.nmask, nrows_img = d.nm_img;
    if (c->early_conv) {
        const long total = (long)B * d.N;
        iota_rows_kernel<<<cdiv(total, 256), 256, 0, st>>>(w.all_rows, total, d.N);
        M3L_LAUNCH_CHECK();
        rows = w.all_rows; nrows = d.N; nrows_img = d.n_img;
    }
    return m3l_heads_loss_fwd2(&c->geom, d.dd, d.dt, B, d.N, nrows, nrows_img, rows, image, tactiles, d.dt ? w.dec_t : (void*)w.dec32,
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
        if (end > sent && (end - sent >= comm->min_bucket || end == comm->total || stage == comm->n_stages)) {
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
    const int64_t* rows = c->early_conv ? w.all_rows : masked;
    const int nrows = c->early_conv ? d.N : d.nmask, nrows_img = c->early_conv ? d.n_img : d.nm_img;
    if (m3l_heads_loss_bwd(&c->geom, d.dd, d.dt, B, d.N, nrows, nrows_img, rows, tensors + g.heads, w.ws_heads, dloss, w.d_dec,
                           grads + g.heads, st))
        return 1;
    if (stage_done()) return 1;
    const bool dec_io = m3l_transformer_rb(&c->dec, B, d.N) && !c->learned_pos;     // (the batch sums of learned positions read fp32 d_dec_in)
    int enc_code = 0;
    {
        const IoScope io(dec_io);
        if (tf_bwd(&c->dec, d.N, w.dec_in, tensors + g.dec, w.ws_dec, w.d_dec, d.dt, w.d_dec_in, grads + g.dec)) return 1;
        if (m3l_unshuffle_bwd(&c->geom, d.D, d.dd, d.dt, B, d.nvis, d.nmask, unmasked, masked, d.dt ? w.enc_t : (void*)w.enc32,
                              tensors + g.glue, w.ws_glue, w.d_dec_in, w.d_enc, &enc_code, grads + g.glue, st))
            return 1;
    }
    if (c->learned_pos) {        // decoder_pos_emb rows (:280-287): every position is present in the decoder input -> plain batch sums
        float* const* gg = grads + g.glue;
        const int stride = d.N * d.dd;
        if (gg[4] && d.n_img && m3l_reduce_rows(w.d_dec_in, B, stride, d.n_img * d.dd, gg[4], 0, st)) return 1;
        if (gg[5] && d.k && m3l_reduce_rows(w.d_dec_in + (size_t)d.n_img * d.dd, B, stride, d.k * d.n_tac * d.dd, gg[5], 0, st)) return 1;
    }
    if (stage_done()) return 1;
    if (tf_bwd(&c->enc, d.nvis, w.tokens, tensors + g.enc, w.ws_enc, w.d_enc, enc_code, w.dtokens, grads + g.enc)) return 1;
    if (front_bwd(c, d, w.f, B, d.nvis, d.nvis_img, unmasked, image, tactiles, tensors + g.embed, w.dtokens, grads + g.embed, st)) return 1;
    if (stage_done()) return 1;
    if (comm) {
        M3L_CHECK(stage == comm->n_stages, "mae_step_bwd: comm plan has %d stages, the backward ran %d", comm->n_stages, stage);
        if (comm->sent_out) *comm->sent_out = sent;
    }
    return 0;
}

// =================================================================================================================================
// Policy-side consumer: obs -> get_embeddings (encoder over ALL tokens, no masking, pretrain_models.py:588-668) -> `head` (the extractor's
// own 1-layer Transformer, :807-817) -> mean over tokens (:838) -> (B, D).  c->geom carries use_vision / use_tactile of the call
// (vision_only_control), c->dec and c->masking_ratio are unused.  tensors = front group | encoder group | head group (11 depth + 2).
namespace {
struct ExtWs {
    float *tokens, *enc32, *head32, *d_head, *d_enc, *dtokens;
    FrontWs f;
    void *ws_enc, *ws_head;
    size_t total;
};
int ext_dims(const m3l_mae_cfg* c, const m3l_tf_cfg* head, StepDims* d) {
    m3l_mae_cfg tmp = *c;
    tmp.masking_ratio = 0.5;                                   // any valid ratio: only the token counts are used
    tmp.dec = tmp.enc;
    if (step_dims(&tmp, d)) return 1;
    M3L_CHECK(head->dim == c->enc.dim && head->dtype == c->enc.dtype, "extractor: head transformer dim / dtype (%d / %d) differ from the encoder's (%d / %d)",
              head->dim, head->dtype, c->enc.dim, c->enc.dtype);
    return 0;
}
ExtWs ext_layout(const m3l_mae_cfg* c, const m3l_tf_cfg* head, const StepDims& d, int B, void* ws) {
    Arena a(ws);
    ExtWs w;
    memset(&w, 0, sizeof(w));
    const size_t Ma = (size_t)B * d.N;
    w.tokens = (float*)a.take(Ma * d.D * 4);
    w.enc32 = (float*)a.take(Ma * d.D * 4);
    w.head32 = (float*)a.take(Ma * d.D * 4);
    w.d_head = (float*)a.take(Ma * d.D * 4);
    w.d_enc = (float*)a.take(Ma * d.D * 4);
    w.dtokens = (float*)a.take(Ma * d.D * 4);
    front_layout(a, c, d, B, d.N, false, &w.f);
    w.ws_enc = a.take(m3l_transformer_ws_bytes(&c->enc, B, d.N));
    w.ws_head = a.take(m3l_transformer_ws_bytes(head, B, d.N));
    w.total = a.off + 256;
    return w;
}
}  // namespace

int m3l_extractor_num_tensors(const m3l_mae_cfg* c, const m3l_tf_cfg* head) {
    return front_tensors(c) + 11 * c->enc.depth + 2 + 11 * head->depth + 2;
}
size_t m3l_extractor_ws_bytes(const m3l_mae_cfg* c, const m3l_tf_cfg* head, int B) {
    StepDims d;
    if (B <= 0 || ext_dims(c, head, &d)) return 0;
    return ext_layout(c, head, d, B, nullptr).total;
}
int m3l_extractor_fwd(const m3l_mae_cfg* c, const m3l_tf_cfg* head, int B, const float* image, const float* const* tactiles,
                      const void* const* tensors, void* ws, float* out, void* stream) {
    StepDims d;
    M3L_CHECK(B > 0 && tensors && ws && out, "extractor_fwd: null argument / B=%d", B);
    if (ext_dims(c, head, &d)) return 1;
    hipStream_t st = (hipStream_t)stream;
    const ExtWs w = ext_layout(c, head, d, B, ws);
    const int g_enc = front_tensors(c), g_head = g_enc + 11 * c->enc.depth + 2;
    struct PrepReset { ~PrepReset() { m3l_prep_set_mode(0); } } prep_reset;
    for (int pass = prep_first_pass(); pass <= 2; ++pass) {        // collect the chain's weight copies, flush them as one launch, run the chain
        m3l_prep_set_mode(prep_mode_of(pass));
        if (pass == 2 && m3l_prep_flush(st)) return 1;
        if (front_fwd(c, d, w.f, B, d.N, d.n_img, nullptr, image, tactiles, tensors, w.tokens, st)) return 1;
        if (m3l_transformer_fwd(&c->enc, B, d.N, w.tokens, tensors + g_enc, w.ws_enc, nullptr, w.enc32, st)) return 1;
        if (m3l_transformer_fwd(head, B, d.N, w.enc32, tensors + g_head, w.ws_head, nullptr, w.head32, st)) return 1;
    }
    m3l_prep_set_mode(0);
    mean_tokens_kernel<<<B, 256, 0, st>>>(w.head32, d.N, d.D, out);
    M3L_LAUNCH_CHECK();
    return 0;
}
// dout (B, D) f32 -> parameter gradients (same order as tensors; NULL = not wanted)
int m3l_extractor_bwd(const m3l_mae_cfg* c, const m3l_tf_cfg* head, int B, const float* image, const float* const* tactiles,
                      const void* const* tensors, void* ws, const float* dout, float* const* grads, void* stream) {
    StepDims d;
    M3L_CHECK(B > 0 && tensors && ws && dout && grads, "extractor_bwd: null argument / B=%d", B);
    if (ext_dims(c, head, &d)) return 1;
    hipStream_t st = (hipStream_t)stream;
    const ExtWs w = ext_layout(c, head, d, B, ws);
    const int g_enc = front_tensors(c), g_head = g_enc + 11 * c->enc.depth + 2;
    mean_tokens_bwd_kernel<<<dim3(cdiv((long)d.N * d.D, 256), B), 256, 0, st>>>(dout, d.N, d.D, w.d_head);
    M3L_LAUNCH_CHECK();
    if (m3l_transformer_bwd(head, B, d.N, w.enc32, tensors + g_head, w.ws_head, w.d_head, 0, w.d_enc, grads + g_head, stream)) return 1;
    if (m3l_transformer_bwd(&c->enc, B, d.N, w.tokens, tensors + g_enc, w.ws_enc, w.d_enc, 0, w.dtokens, grads + g_enc, stream)) return 1;
    return front_bwd(c, d, w.f, B, d.N, d.n_img, nullptr, image, tactiles, tensors, w.dtokens, grads, st);
}

}  // extern "C"
