/* m3l_amd — C ABI of the MI355X-native masked multimodal auto-encoder (MAE) training step.
 *
 * The reference (Leonhard111/M3L) is pure Python: its "FFI" for this path is the nn.Module object graph
 * (models/pretrain_models.py:59-786).  This header is the boundary a binding would load instead: plain pointers
 * and sizes, no torch types.  Every entry point
 *   - is asynchronous on the given hipStream_t (pass torch.cuda.current_stream().cuda_stream),
 *   - allocates nothing (the caller passes workspaces sized by the *_ws_bytes functions),
 *   - returns 0 on success, non-zero on error (text via m3l_last_error; nothing is thrown across the ABI),
 *   - takes device pointers unless a parameter says "host".
 * dtype codes: 0 = f32 compute (parity path), 1 = bf16 compute (fp32 master weights, fp32 accumulation,
 * fp32 residual stream).  dim_head is 64 everywhere (all reference configs).
 *
 * Reference interface each entry replaces:
 *   m3l_mask_sample        torch.rand(B,n).argsort(-1) per modality + slicing      pretrain_models.py:223-248
 *   m3l_embed_fwd/bwd      Rearrange + LayerNorm/Linear/LayerNorm + modality + sincos + visible gather
 *                                                                                   pretrain_models.py:157-216,255-256,768-778
 *   m3l_transformer_*      vit_pytorch.vit.Transformer.forward (encoder :266, decoder :309, extractor :836)
 *   m3l_unshuffle_*        enc_to_dec + zeros/scatter/mask_token + decoder modality/sincos   :270-307
 *   m3l_heads_loss_*       masked-row gather + to_pixels/to_tactiles + F.mse_loss (x1 / x10)   :260-262,327-340
 *   m3l_earlycnn_*         EarlyCNN.forward (early_conv_masking=True)                                   :37-56,180-191
 *   m3l_layernorm_*        nn.LayerNorm (models/VTT.py:354 final norm of the DINO-style encoder)
 *   m3l_vt_load            utils/pretrain_utils.py:7-57 (NHWC -> NCHW, per-sensor channel pick, [-1,1] -> [0,1])
 */
/* Threading / device contract.  One process per GPU (the torch.distributed model) with ONE host thread driving a device's steps:
 * entry points are asynchronous on the caller's HIP stream, allocate nothing, and keep no state except (a) the thread-local error
 * string, (b) a per-device low-priority side stream + event ring (weight gradients, weight casts), created on first use under a
 * mutex, and (c) process-wide tuning switches (m3l_set_attn_block / m3l_set_t192 / m3l_set_rowln) and one-time kernel-attribute
 * initialisation for the first device used.  A second device in the same process, or two threads stepping one device concurrently,
 * are outside the contract. */
#ifndef M3L_AMD_H
#define M3L_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M3L_MAX_TACTILES 8

typedef struct m3l_geom {
    int image_h, image_w, image_patch, image_channels;
    int tactile_h, tactile_w, tactile_patch, tactile_channels;
    int num_tactiles;        /* sensors in the model */
    int use_vision;          /* 0 -> image tokens absent from this call */
    int use_tactile;         /* 0 -> tactile tokens absent from this call */
} m3l_geom;

typedef struct m3l_tf_cfg {
    int dim, depth, heads, mlp_dim;
    int project_out;         /* vit_pytorch: to_out is Identity when heads == 1 and dim_head == dim */
    int dtype;
} m3l_tf_cfg;

int m3l_version(void);
int m3l_last_error(char* buf, size_t n);
/* experimental: run the LayerNorms inside the epilogue of the neighbouring GEMM (returns the previous setting; default off) */
int m3l_set_rowln(int enable);
/* fused per-sample block kernels for short sequences (bf16, dim 128 / 192, heads = dim / 64, n <= 48): mode 0 = off, 1 =
 * forward attention block (LN1 + QKV + attention + out-proj + residual + LN2) + forward / backward MLP blocks, 3 (default) = also
 * the attention backward block; env M3L_ATTN_BLOCK sets the initial mode.  Returns the
 * previous mode.  The fused kernels read and write exactly the activations of the unfused ones. */
int m3l_set_attn_block(int mode);
/* Short-sequence stacks whose every half layer takes a block kernel, a bit mask: 1 (default) = the whole forward of the stack in ONE launch
 * (the same kernel bodies, layer after layer inside one workgroup per sample); 2 = also its backward as one launch per weight-gradient
 * group of layers (opt-in: measured slower beside the side stream's weight gradients); 0 = one launch per half layer; env M3L_ENC_MEGA.
 * Returns the previous setting.  Bit-identical results either way. */
int m3l_set_enc_mega(int mode);
/* Profiling hook of the forward attention block (tools/attn_phase_probe.py): dev_buf = device buffer of >= 8 * B uint64 that every later
 * launch fills, per sample, with the shader-clock stamps [start, LN1 done, QKV done, attention done, out-proj done, end]; NULL (the
 * default) switches it off. */
void m3l_set_attn_phase_buffer(void* dev_buf);
/* Deferred join of the side stream.  By default every backward entry point returns with all its gradients ordered on the caller's
 * stream.  With m3l_set_defer_join(1) the weight gradients that are still running on the library's side stream when a backward
 * entry point returns are NOT joined: they overlap what the caller enqueues next, and the caller calls m3l_side_join(stream) before
 * anything on `stream` consumes parameter gradients (optimizer step, all-reduce) — and keeps every workspace it passed to a backward
 * alive until then.  m3l_amd.parallel.GradSync does both when it is not communicating.  Returns the previous setting / 0. */
int m3l_set_defer_join(int on);
/* Diagnostic: 1 = run every weight-gradient kernel on the caller's stream instead of the side stream (no overlap: per-kernel counters and
 * stand-alone durations are attributable); env M3L_WGRAD_INLINE sets the initial value.  Returns the previous setting. */
int m3l_set_wgrad_inline(int on);
int m3l_side_join(void* stream);
/* Gradient all-reduce by RCCL called directly, on the library's side stream (comm.hip; RCCL is dlopen'ed at run time — every entry returns
 * non-zero with m3l_last_error set when it is not available, and the caller falls back to torch.distributed).  One communicator per
 * process: m3l_comm_unique_id on rank 0 -> 128 bytes to every rank (any channel) -> m3l_comm_init(id, rank, world) on every rank
 * (collective).  m3l_comm_allreduce(buf, count, after_stream): buf[0..count) fp32 <- sum over ranks, in place, on the side stream behind
 * everything queued on after_stream and behind the side stream's earlier work; the result is ordered for a consumer by
 * m3l_side_join(stream).  Every rank issues the same sequence of calls. */
int m3l_comm_available(void);     /* local probe, no communication: 0 = RCCL loads in this process (agree on it across ranks BEFORE m3l_comm_init) */
int m3l_comm_unique_id(void* out128);
/* a repeated call with the same (rank, world) keeps the live communicator; a different (rank, world) is refused until m3l_comm_destroy */
int m3l_comm_init(const void* id128, int rank, int world);
int m3l_comm_world(void);
int m3l_comm_allreduce(float* buf, size_t count, void* after_stream);
int m3l_comm_destroy(void);
/* building blocks of the above (ordering only): the side stream waits for `after_stream` and is returned / a pending tail is left */
int m3l_side_fork(void* after_stream, void** side_out);
int m3l_side_mark_pending(void);
/* The library's side stream of the current device (hipStream_t, lowest priority, created on first use), or NULL on error. */
void* m3l_side_stream(void);
int m3l_side_pending(void);      /* number of un-joined deferred tails (diagnostic / tests) */
/* row-tiled fused half layers for long sequences (bf16, dim 192, n > 48: the MAE decoder, models/pretrain_models.py:309): 192 token
 * rows per workgroup (t192.hip).  Bit mask: 1 (default) = sequences longer than 48 tokens, 2 = also the MLP halves of short
 * sequences (48-row tiles, two chunk parities; correct, measured equal to the per-sample block kernels); env M3L_T192 sets the
 * initial mode, 0 falls back to the per-op kernels.  Returns the previous mode. */
/* Hidden activation h = GELU(u) of the fused feed-forward kernels: 1 = not written by the forward (one store stream less, 1.5 KB per token row
 * and layer less HBM traffic); the weight-gradient kernel of fc2 then stages u and applies the GELU itself (bit-identical gradients; that
 * kernel gets 40 % longer, the step time does not change).  Default 0.
 * Set it BEFORE a forward and keep it until its backward has run.  env M3L_DROP_H.  Returns the previous setting. */
int m3l_set_drop_h(int on);
int m3l_set_t192(int on);
/* height of the tall row tiles at width 192: 12 token tiles = 192 rows, one workgroup per CU, or 6 = 96 rows, 6 + 2 waves
 * and a 3-stage ring so that two workgroups share a CU and overlap each other's memory-only phases; env M3L_T192_TT.  Returns the previous value. */
int m3l_set_t192_tt(int tt);
/* 1: bf16 residual stream (opt-in, round 4; env M3L_RES_BF16): inside a transformer stack whose every layer runs on the fused kernels, the
 * layer inputs / outputs x, x1, xout and the running residual gradient travel through HBM as bf16 instead of fp32 (registers stay fp32;
 * the stack's input and input gradient stay fp32 at this interface).  Set it before a forward and keep it until its backward has run.
 * Loss stays within the 1e-2 of bf16 compute; gradients carry one more bf16 rounding per half layer.  Returns the previous setting. */
int m3l_set_residual_bf16(int on);
/* experiment switch of the row-tiled feed-forward backward (EXPERIMENTS 5.2): phase-offset wave groups / static wave priority; 0, 0 = off */
int m3l_set_t192_stagger(int lead_mask, int prio_mask);

/* ---- mask sampling (INT path, bit-exact): noise[i] is (B, n_i) f32, RNG order image, tactile1..k.
 * Stable ascending argsort; outputs int64 (B, num_masked) / (B, num_unmasked) in the reference's concat order.
 * counts_host (out, host): [num_masked, num_unmasked, nm_img, nm_tac, n_img, n_tac] */
int m3l_mask_counts(const m3l_geom* g, double ratio, int* counts_host);
int m3l_mask_sample(const m3l_geom* g, double ratio, int B, const float* const* noise, int64_t* masked, int64_t* unmasked,
                    void* stream);

/* explicit per-modality masked counts (VTMAE.reconstruct: int(r*n_img), int(r*n_tac_total/k); pretrain_models.py:425,433) */
int m3l_mask_sample_counts(const m3l_geom* g, int nm_img, int nm_tac, int B, const float* const* noise, int64_t* masked,
                           int64_t* unmasked, void* stream);

/* ---- patch embed.  idx == NULL: all patches (get_embeddings); else idx = unmasked (B, L) and only those rows are embedded.
 * cnt_img = number of image entries at the head of each list row (n_img when idx == NULL, n_img - nm_img for the visible list).
 * tensors: image {ln1_w, ln1_b, W[D,pd], b, ln2_w, ln2_b}, tactile {same 6}, mod_emb[(1+k),D], pos_img[n_img,D], pos_tac[k*n_tac,D]
 * tokens out: f32 (B, L, D).  grads: same order as tensors (entries 12..14: mod_emb grad, NULL, NULL). */
size_t m3l_embed_ws_bytes(const m3l_geom* g, int D, int dtype, int B, int L);
int m3l_embed_fwd(const m3l_geom* g, int D, int dtype, int B, int L, int cnt_img, const int64_t* idx, const float* image,
                  const float* const* tactiles, const void* const* tensors, void* ws, float* tokens, void* stream);
int m3l_embed_bwd(const m3l_geom* g, int D, int dtype, int B, int L, int cnt_img, const int64_t* idx, const float* image,
                  const float* const* tactiles, const void* const* tensors, void* ws, const float* dtokens, float* const* grads,
                  void* stream);

/* ---- transformer stack.  x_in f32 (B, n, D).  tensors: per layer {ln1_w, ln1_b, qkv_w[3HD,D], out_w[D,HD], out_b, ln2_w, ln2_b,
 * fc1_w[mlp,D], fc1_b, fc2_w[D,mlp], fc2_b} x depth, then {norm_w, norm_b}.  Outputs: y_t (compute type, may be NULL) and
 * y32 (f32, may be NULL) = final LayerNorm output.  ws keeps the activations for the backward. */
size_t m3l_transformer_ws_bytes(const m3l_tf_cfg* c, int B, int n);
int m3l_transformer_fwd(const m3l_tf_cfg* c, int B, int n, const float* x_in, const void* const* tensors, void* ws, void* y_t,
                        float* y32, void* stream);
/* dy_dtype: 0 -> dy is f32, 1 -> dy is bf16 (must equal c->dtype when 1).  dx_in out: f32 (B, n, D) or NULL. */
int m3l_transformer_bwd(const m3l_tf_cfg* c, int B, int n, const float* x_in, const void* const* tensors, void* ws, const void* dy,
                        int dy_dtype, float* dx_in, float* const* grads, void* stream);

/* the same, restricted to layers [layer_lo, layer_hi) (descending): lets the caller interleave the gradient all-reduce of finished
 * layers with the backward of the remaining ones.  Call with layer_hi == depth first (consumes dy), down to layer_lo == 0. */
int m3l_transformer_bwd_range(const m3l_tf_cfg* c, int B, int n, const float* x_in, const void* const* tensors, void* ws,
                              const void* dy, int dy_dtype, float* dx_in, float* const* grads, int layer_hi, int layer_lo, void* stream);

/* ---- frozen ViT forward (inference only): the DINOv2-S/14-reg image feature of the cfg-5 fusion head
 * (models/pretrain_models_dino_cat_mae.py:886 `self.dino_model(obs_viso)`; train_dino_cat_mae.py:29).  x: tokens (B, n, dim) f32
 * (cls + registers + patches, positions already added); y32: final-norm output (B, n, dim) f32.
 * tensors: per layer 12 pointers {norm1.w, norm1.b, qkv.W[3*heads*64, dim], qkv.b, proj.W[dim, heads*64], proj.b, norm2.w, norm2.b,
 * fc1.W[mlp, dim], fc1.b, fc2.W[dim, mlp], fc2.b}, then {norm.w, norm.b}.  The four W are in the COMPUTE type (c->dtype; they
 * never change, so the caller casts once), everything else f32; LayerScale is folded into proj / fc2 by the caller. */
size_t m3l_frozen_vit_ws_bytes(const m3l_tf_cfg* c, int B, int n);
int m3l_frozen_vit_fwd(const m3l_tf_cfg* c, float ln_eps, int B, int n, const float* x, const void* const* tensors, void* ws,
                       float* y32, void* stream);

/* ---- encoder -> decoder glue.  enc32 / enc_t: final-norm output of the encoder (B, nvis, D) in f32 / compute type.
 * tensors: {e2d_w[dd,D] or NULL, e2d_b or NULL, mask_token[dd], dec_mod[(1+k),dd], pos_img_dec[n_img,dd], pos_tac_dec[k*n_tac,dd]} */
size_t m3l_unshuffle_ws_bytes(const m3l_geom* g, int D, int dd, int dtype, int B, int nvis, int nmask);
int m3l_unshuffle_fwd(const m3l_geom* g, int D, int dd, int dtype, int B, int nvis, int nmask, const int64_t* unmasked,
                      const int64_t* masked, const float* enc32, const void* enc_t, const void* const* tensors, void* ws,
                      float* dec_in, void* stream);
/* d_enc out: gradient w.r.t. the encoder output; f32 when e2d is NULL (then *d_enc_dtype = 0) else compute type (= dtype). */
int m3l_unshuffle_bwd(const m3l_geom* g, int D, int dd, int dtype, int B, int nvis, int nmask, const int64_t* unmasked,
                      const int64_t* masked, const void* enc_t, const void* const* tensors, void* ws, const float* d_dec_in,
                      void* d_enc, int* d_enc_dtype, float* const* grads, void* stream);

/* ---- heads + masked MSE.  dec_t: decoder output (B, N, dd) compute type.  tensors: {pix_w[pd_i,dd], pix_b, tac_w[pd_t,dd], tac_b}.
 * loss out: f32 scalar = mse(img) + 10 mse(tac).  Optional dumps (f32, may be NULL): pred/target of each head. */
size_t m3l_heads_ws_bytes(const m3l_geom* g, int dd, int dtype, int B, int nmask);
int m3l_heads_loss_fwd(const m3l_geom* g, int dd, int dtype, int B, int N, int nmask, int nm_img, const int64_t* masked,
                       const float* image, const float* const* tactiles, const void* dec_t, const void* const* tensors, void* ws,
                       float* loss, float* pred_img, float* tgt_img, float* pred_tac, float* tgt_tac, void* stream);
/* the same + loss_parts (f32[2], may be NULL): mse(image) and 10*mse(tactile) separately (VTMAE.reconstruct logging) */
int m3l_heads_loss_fwd2(const m3l_geom* g, int dd, int dtype, int B, int N, int nmask, int nm_img, const int64_t* masked,
                        const float* image, const float* const* tactiles, const void* dec_t, const void* const* tensors, void* ws,
                        float* loss, float* loss_parts, float* pred_img, float* tgt_img, float* pred_tac, float* tgt_tac, void* stream);
/* d_dec out: compute type (B, N, dd), zero on visible rows.  dloss: device f32 scalar (upstream gradient) or NULL for 1. */
int m3l_heads_loss_bwd(const m3l_geom* g, int dd, int dtype, int B, int N, int nmask, int nm_img, const int64_t* masked,
                       const void* const* tensors, void* ws, const float* dloss, void* d_dec, float* const* grads, void* stream);

/* ---- the whole MAE step in two calls: VTMAE.forward + loss.backward() (models/pretrain_models.py:146-342 as called at
 * models/ppo_mae.py:262-263, models/sac_mae.py:284-291), for both front ends (patch embed, or the EarlyCNN stems of early_conv_masking=True
 * with patch sizes 8 / 4: cfg.early_conv) and both position encodings (cfg.learned_pos).  The module entry
 * points above chained in C: same kernels, launch order and results (bit-identical), two host calls instead of ten autograd hops.
 * tensors / grads: ONE array, the groups in this order — embed (15, as m3l_embed_fwd) | encoder transformer (11 * depth + 2) |
 * glue (6, as m3l_unshuffle_fwd) | decoder transformer (11 * depth + 2) | heads (4): m3l_mae_step_num_tensors() entries.
 * noise: one (B, n_i) f32 array per present modality, RNG order image, tactile1..k (m3l_mask_sample).  ws: m3l_mae_step_ws_bytes, holds
 * every activation between the two calls.  masked / unmasked (out of the forward, in of the backward; caller-owned, unchanged in
 * between): int64 (B, num_masked) / (B, num_unmasked) index lists in the reference's order. */
typedef struct m3l_mae_cfg {
    m3l_geom geom;
    m3l_tf_cfg enc, dec;
    double masking_ratio;
    int early_conv;      /* early_conv_masking=True (train.py:62, the reference's default): EarlyCNN stems over the frames + visible gather
                          * instead of the patch embed, loss over ALL patches (pretrain_models.py:180-191,311-322).  The embed group is then
                          * 19 tensors: image stem {conv1.w, conv1.b, ..., conv4.w, conv4.b}, tactile stem (same 8), modality table, pos_img, pos_tac */
    int learned_pos;     /* use_sincosmod_encodings=False (:218-219,280-287): the position tables are trained; their gradient slots (embed group
                          * [13..14] / [17..18], glue group [4..5]) receive the batch sums of the token gradients */
} m3l_mae_cfg;
/* Data-parallel plan of m3l_mae_step_bwd (NULL = one rank): `flat` is the flat fp32 gradient buffer of `total` elements that the
 * grads pointers point into, laid out in backward order; stage_end[i] = end of the prefix of it that is final after stage i — stages:
 * heads, decoder chunks (top-down, layers_per_chunk layers each; one stage when layers_per_chunk is 0 or >= depth), glue, encoder
 * chunks, embed.  Once the finished-but-unsent prefix holds >= min_bucket elements (or reaches total) it is summed over the ranks by
 * m3l_comm_allreduce on the side stream.  *sent_out (optional) = elements sent; the caller sends any rest and joins (m3l_side_join). */
typedef struct m3l_comm_plan {
    float* flat;
    long total, min_bucket;
    int layers_per_chunk, n_stages;
    const long* stage_end;
    long* sent_out;
} m3l_comm_plan;
int m3l_mae_step_num_tensors(const m3l_mae_cfg* c);
size_t m3l_mae_step_ws_bytes(const m3l_mae_cfg* c, int B);
int m3l_mae_step_fwd(const m3l_mae_cfg* c, int B, const float* image, const float* const* tactiles, const float* const* noise,
                     const void* const* tensors, void* ws, float* loss, int64_t* masked, int64_t* unmasked, void* stream);
/* dloss: device f32 scalar (upstream gradient) or NULL for 1.  grads[i]: f32, shape of tensors[i], NULL = not wanted / no gradient. */
int m3l_mae_step_bwd(const m3l_mae_cfg* c, int B, const float* image, const float* const* tactiles, const int64_t* masked,
                     const int64_t* unmasked, const void* const* tensors, void* ws, const float* dloss, float* const* grads,
                     const m3l_comm_plan* comm, void* stream);

/* ---- the policy-side consumer of the MAE in two calls: MAEExtractor.forward (models/pretrain_models.py:819-841; run on every environment
 * step at B = number of envs, and with grad on every PPO minibatch, models/ppo_mae.py:280):
 *     tokens = get_embeddings(x)  (encoder over ALL tokens, no masking, :588-668)  ->  `head` (the extractor's own 1-layer Transformer,
 *     :807-817)  ->  mean over the tokens (:838)  ->  out (B, D) f32.
 * c: geom (use_vision / use_tactile of the call = vision_only_control), enc, early_conv, learned_pos; dec and masking_ratio are unused.
 * tensors / grads: front group (15, or 19 with early_conv) | encoder group (11 depth + 2) | head group (11 head->depth + 2).
 * The workspace holds every activation between the two calls; m3l_extractor_fwd alone is the no-grad rollout path. */
int m3l_extractor_num_tensors(const m3l_mae_cfg* c, const m3l_tf_cfg* head);
size_t m3l_extractor_ws_bytes(const m3l_mae_cfg* c, const m3l_tf_cfg* head, int B);
int m3l_extractor_fwd(const m3l_mae_cfg* c, const m3l_tf_cfg* head, int B, const float* image, const float* const* tactiles,
                      const void* const* tensors, void* ws, float* out, void* stream);
int m3l_extractor_bwd(const m3l_mae_cfg* c, const m3l_tf_cfg* head, int B, const float* image, const float* const* tactiles,
                      const void* const* tensors, void* ws, const float* dout, float* const* grads, void* stream);

/* ---- EarlyCNN stem (early_conv_masking=True, the reference's default flag; pretrain_models.py:37-56,180-191): three
 * Conv2d+ReLU and a 1x1 Conv2d as im2col + MFMA GEMM.  srcs: nsrc NCHW f32 inputs of B samples each (the tactile sensors share
 * one stem; they are processed as one batch of nsrc*B).  tensors / grads: {conv1.w, conv1.b, ..., conv4.w, conv4.b}.
 * out: f32 (nsrc*B, h*w, dim) stem tokens (source-major). */
typedef struct m3l_cnn_cfg {
    int in_channels, height, width, dim;
    int tactile;             /* 0: image stem (conv3 4/2/1), 1: tactile stem (conv3 3/1/1) */
    int dtype;
} m3l_cnn_cfg;
/* 1 (default): the bf16 stems of dim 64 / 128 / 256 run their three Conv2d + ReLU as direct (implicit-GEMM) convolutions (conv.hip: input
 * patch of an 8 x 8 output tile staged once in LDS, no column matrix); 0: im2col + GEMM everywhere.  env M3L_DIRECT_CONV.  Set it before a
 * forward and keep it until its backward has run (the workspace layout depends on it).  Returns the previous setting. */
int m3l_set_direct_conv(int on);
size_t m3l_earlycnn_ws_bytes(const m3l_cnn_cfg* c, int B, int nsrc);
int m3l_earlycnn_fwd(const m3l_cnn_cfg* c, int B, int nsrc, const float* const* srcs, const void* const* tensors, void* ws, float* out,
                     void* stream);
/* srcs: the same input frames as the forward (the direct convolutions of the bf16 path keep no column matrix and read them again) */
int m3l_earlycnn_bwd(const m3l_cnn_cfg* c, int B, int nsrc, const float* const* srcs, const void* const* tensors, void* ws, const float* dout,
                     float* const* grads, void* stream);
/* stem tokens -> encoder tokens: + modality embedding + sincos position (pretrain_models.py:202-216); tensors {mod_emb, pos_img, pos_tac} */
size_t m3l_tokens_assemble_ws_bytes(const m3l_geom* g, int D);
int m3l_tokens_assemble_fwd(const m3l_geom* g, int D, int B, const float* img_tok, const float* tac_tok, const void* const* tensors,
                            float* tokens, void* stream);
int m3l_tokens_assemble_bwd(const m3l_geom* g, int D, int B, const float* dtokens, float* d_img, float* d_tac, void* ws, float* dmod,
                            void* stream);

/* ---- stand-alone ops (also used by the DINO-style VTT front end and by the unit tests) */
size_t m3l_layernorm_ws_bytes(int D);
int m3l_layernorm_fwd(int out_dtype, const float* x, int M, int D, const float* gamma, const float* beta, float eps, void* y,
                      float* y32, void* stream);
int m3l_layernorm_bwd(int dy_dtype, const void* dy, const float* x, int M, int D, const float* gamma, float eps, const float* dres,
                      float* dx, void* ws, float* dgamma, float* dbeta, void* stream);
/* rows of src (B, N, D) f32 picked by idx (B, K) -> dst (B, K, D)  [tactile_ssl.utils.apply_masks] and its scatter-add-free adjoint */
int m3l_gather_tokens(const float* src, int B, int N, int D, const int64_t* idx, int K, float* dst, void* stream);
int m3l_scatter_tokens(const float* src, int B, int N, int D, const int64_t* idx, int K, float* dst_zeroed, void* stream);
/* obs (B, H, W, 3*fs) f32 NHWC -> image (B, 3*fs, H, W); tactile (B, 3*S*fs, h, w) -> per-sensor (B, 3*fs, h, w), (x+1)/2 */
int m3l_vt_load(const float* image_nhwc, int B, int H, int W, int C, float* image_nchw, const float* tactile, int th, int tw,
                int n_sensors, int frame_stack, float* const* tactile_out, void* stream);
/* the same with the observation dtype (0 = f32, 1 = uint8: raw camera / sensor frames, no host-side cast) and the reference's
 * normalisation arguments: out = (x - lo) / (hi - lo), fp32 IEEE arithmetic as utils/pretrain_utils.py:28-30,47-49 */
/* lo / hi are the reference's Python numbers, i.e. doubles: the kernel subtracts (float)lo and divides by (float)(hi - lo), the span
 * formed in double and rounded ONCE, exactly as `tensor / (hi - lo)` does (bit-exact for ranges like [0.1, 0.3] too) */
int m3l_vt_load2(const void* image_nhwc, int image_u8, int B, int H, int W, int C, double img_lo, double img_hi, float* image_nchw,
                 const void* tactile, int tactile_u8, int th, int tw, int n_sensors, int frame_stack, double tac_lo, double tac_hi,
                 float* const* tactile_out, void* stream);

/* ---- torch.optim.Adam semantics (models/ppo_mae.py:182-183) over one flat fp32 buffer of n elements; step counts from 1 */
int m3l_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int step, void* stream);
/* gradients read as grad_scale * grads[i]: the 1 / world of a SUM all-reduce folded into the update (grad_scale = 1 is bit-identical
 * to m3l_adam_step) */
int m3l_adam_step_scaled(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1, float beta2,
                         float eps, float weight_decay, int step, float grad_scale, void* stream);
/* torch.optim.AdamW semantics (VTMAE.initialize_training, pretrain_models.py:675: decoupled weight decay) with an optional fused
 * torch.nn.utils.clip_grad_norm_(params, max_grad_norm) (pretrain_models.py:710) over the same flat gradient buffer: max_grad_norm > 0 ->
 * norm_ws (>= 1026 floats, device) receives partial sums, then norm_ws[1024] = the clip coefficient min(1, max / (||grad_scale g|| + 1e-6))
 * and norm_ws[1025] = the norm; the coefficient is applied inside the update, and with scale_grads != 0 the scaled gradient is also left in
 * `grads` (what clip_grad_norm_ leaves behind).  max_grad_norm <= 0: no clipping, norm_ws may be NULL.  Three launches, fixed summation order. */
int m3l_adamw_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, float grad_scale, float max_grad_norm, float* norm_ws, int scale_grads, void* stream);
/* graph-capturable form: the step counter lives on the device (int, incremented by the call) together with the two bias
 * corrections (float[2] scratch), so a captured launch stays correct on every replay (torch's Adam(capturable=True)) */
int m3l_adam_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                      float beta2, float eps, float weight_decay, int* step_dev, float* bias_corr_dev, void* stream);

/* ---- in-library HIP-event timing of kernel classes (bench.py roofline).  filter: substring of "kind[AxBxC]" or NULL = all;
 * stride: bracket every stride-th matching launch (sampling keeps the perturbation of the timed region small). */
void m3l_prof_begin(const char* filter, int stride);
void m3l_prof_end(void);
double m3l_prof_event_overhead_us(void* stream, int n);   /* median cost of an EMPTY event bracket (no kernel between) */
int m3l_prof_count(void);
int m3l_prof_get(int i, char* name, size_t n, double* ms_total, long* launches, double* flops_total, double* bytes_total);

/* ---- raw kernels, exported for the per-kernel parity tests */
int m3l_op_gemm_nt(int dtype, const void* A, int lda, const void* W, int ldw, int M, int N, int K, const float* bias,
                   const float* res, float* out_f32, void* out_t, void* out_pre, const void* gelu_u, int act, int ldc, void* stream);
size_t m3l_op_gemm_tn_ws_bytes(int M, int N, int K);
int m3l_op_gemm_tn(int dtype, const void* Y, int ldy, const void* X, int ldx, int M, int N, int K, void* ws, size_t ws_bytes,
                   float* out, int ldo, void* stream);
/* `count` weight gradients that share M in ONE grouped launch (what a transformer layer group's backward issues: dW_i [N_i, K_i] = Y_i^T X_i,
 * every nn.Linear.weight.grad of vit_pytorch Attention / FeedForward via loss.backward(), models/ppo_mae.py:263); bf16 operands take the
 * 256 x 192 split-M kernel of wgrad.hip.  Y / X / out: host arrays of device pointers; ld* / N / K: host int arrays. */
size_t m3l_op_gemm_tn_grouped_ws_bytes(int dtype, int count, int M, const int* N, const int* K);
int m3l_op_gemm_tn_grouped(int dtype, int count, int M, const void* const* Y, const int* ldy, const void* const* X, const int* ldx, const int* N,
                           const int* K, float* const* out, void* ws, size_t ws_bytes, void* stream);
/* building blocks of the trainable fusion MLP of the cfg-5 extractor (models/pretrain_models_dino_cat_mae.py:828-836,899-903: Linear + ReLU +
 * Dropout x 2, Linear, over cat(pooled MAE tokens, DINOv2 feature)): column sums (bias gradients), compute-type + transposed weight copy
 * (dst / dstT may be NULL), keep-mask / ReLU-mask scaling (mode 0: ref = uint8 mask; mode 1: ref = f32, on where ref > 0), and the
 * (rows, na) | (rows, nb) concat (split = 0) / its adjoint (split = 1: a, b <- columns of cat; b may be NULL) */
size_t m3l_op_colsum_ws_bytes(int N);
int m3l_op_colsum(int dtype, const void* Y, int M, int N, int ld, void* ws, float* out, void* stream);
int m3l_op_prep_weight(int dtype, const float* src, int rows, int cols, void* dst, void* dstT, void* stream);
int m3l_op_mask_scale(int mode, const float* src, const void* ref, float scale, long count, float* out, void* stream);
int m3l_op_concat2(float* a, int na, float* b, int nb, int rows, float* cat, int split, void* stream);
/* front end of the frozen image encoder (DINOv2 `patch_embed` + `prepare_tokens_with_masks`; models/pretrain_models_dino_cat_mae.py:886): patches
 * of x (B, C, H, W) f32 as GEMM rows [B (H/P)(W/P), Kpad] in the Conv2d weight's K order; tokens = [cls + pos0 | registers | patch + pos] */
int m3l_op_patch_cols(int dtype, const float* x, int B, int C, int H, int W, int P, int Kpad, void* col, void* stream);
int m3l_op_vit_tokens(const float* emb, const float* cls, const float* regs, const float* pos, int B, int npatch, int R, int D, float* tok,
                      void* stream);
int m3l_op_attn_fwd(int dtype, const void* qkv, void* o, float* lse, int B, int n, int H, void* stream);
int m3l_op_attn_bwd(int dtype, const void* qkv, const void* o, const void* dO, const float* lse, float* dsum, void* dqkv, int B,
                    int n, int H, void* stream);

#ifdef __cplusplus
}
#endif
#endif
