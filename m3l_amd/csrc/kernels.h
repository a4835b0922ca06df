// Internal launcher interface shared by the kernel translation units and the MAE step plan (mae_plan.hip).
// Every launcher is asynchronous on the given stream, allocates nothing, returns 0 on success and sets the
// thread-local error string otherwise (m3l_last_error).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define M3L_MAX_SENSORS 8
#define M3L_MAX_PARTIAL_BLOCKS 2048      // capacity of the partial-row workspaces; the launch grids use m3l_part_blocks() <= this

// dtype codes: 0 = f32, 1 = bf16

// RAII HIP-event bracket around one launch (prof.hip); a no-op unless m3l_prof_begin() is active.
struct ProfScope {
    int idx;
    hipStream_t stream;
    // work = algorithmic FLOPs of the launch, bytes = algorithmic HBM bytes (each operand / result once)
    ProfScope(const char* kind, long a, long b, long c, double work, hipStream_t st, double bytes = 0.0);
    ~ProfScope();
};

struct GemmEpi {
    const float* bias;     // [N] or null
    const float* res;      // f32 [M, ldc] residual added last, or null
    float* out_f32;        // f32 [M, ldc] or null
    void* out_t;           // compute-type [M, ldc] or null (post-activation)
    void* out_pre;         // compute-type [M, ldc] pre-activation copy or null
    const void* gelu_u;    // compute-type [M, ldc]: multiply by gelu'(u) (dgrad through GELU) or null
    int act;               // 0 none, 1 GELU(erf), 2 ReLU
    const void* relu_ref;  // compute-type [M, ldc]: zero the result where relu_ref <= 0 (dgrad through ReLU) or null
    int ldc;
    float alpha;
    int n_bias;            // bias has n_bias valid entries (columns beyond read as 0); 0 -> N
    float* colsum_part;    // [m3l_gemm_nt_colsum_rows(M, N), N] per-row-block column sums of out_t (bias gradient fused into a dgrad) or null
    const void* res_t;     // bf16 [M, ldc] residual added last (bf16 residual stream: the result goes to out_t), or null; not with res / gelu_u
};

struct WeightDesc {
    const float* src;      // [rows, cols] fp32 master
    void* dst;             // [rows, ld_dst]  compute-type copy (or null)
    void* dstT;            // [cols, ld_dstT] transposed compute-type copy (or null)
    int rows, cols, ld_dst, ld_dstT;
};
#define M3L_WPACK 64
struct WeightPack {
    WeightDesc d[M3L_WPACK];
    int tile0[M3L_WPACK + 1];   // filled by m3l_prep_weights: first 64x64 tile of each matrix in the launch's tile list
    int count;
};

struct PatchGroup {
    const float* src[M3L_MAX_SENSORS];   // NCHW f32 inputs of the sensors of this modality group
    int nsrc, C, H, W, P;
    int npatch;                          // patches per sensor
    int base;                            // token index of the group's first patch
};

extern "C" int m3l_side_fork(void* after_stream, void** side_out);
extern "C" int m3l_side_mark_pending(void);
int m3l_part_blocks(void);   // workgroups of the partial-sum kernels (default 1024; env M3L_PART_BLOCKS)
int m3l_gemm_nt(int dtype, const void* A, int lda, const void* W, int ldw, int M, int N, int K, const GemmEpi* epi, hipStream_t st);
size_t m3l_gemm_tn_ws_bytes(int M, int N, int K, int* splits_out);
int m3l_gemm_tn(int dtype, const void* Y, int ldy, const void* X, int ldx, int M, int N, int K, float* partial_ws, size_t ws_bytes,
                float* out, int ldo, int nvalid, int kvalid, int accumulate, hipStream_t st);
// NT GEMM with the LayerNorm forward / backward fused on the full output row (gemm_rowln.hip); N must be 64*{1,2,3,4,6}
enum { ROWLN_FWD = 0, ROWLN_BWD = 1 };
struct RowLnEpi {
    const float* bias;     // FWD: Linear bias [N] or null
    const float* res;      // FWD: fp32 residual added before the norm; BWD: fp32 residual gradient added after it (may alias x_out)
    const float* x;        // BWD: the LayerNorm input of the forward pass (fp32 [M, N])
    const float* gamma;    // LayerNorm weight
    const float* beta;     // FWD: LayerNorm bias
    float* x_out;          // FWD: pre-norm sum (new residual stream); BWD: dx (fp32) — or null
    void* out_t;           // FWD: normalised row in the compute type; BWD: dx in the compute type — or null
    float* out_f32;        // FWD: normalised row in fp32 — or null
    float* part;           // BWD: [cdiv(M,64)][3*N] partials: dgamma | dbeta | colsum(dx)
    float eps;
};
bool m3l_gemm_nt_rowln_supported(int dtype, int N, int K);
int m3l_gemm_nt_rowln(int dtype, int mode, const void* A, int lda, const void* W, int ldw, int M, int N, int K, const RowLnEpi* ep,
                      hipStream_t st);
// out_i[j] = sum_g part[g*3*D + i*D + j], i = 0..2 (null outputs skipped): reduction of the LayerNorm-backward partial slabs
// many [G][3 D] partial slabs -> their three destinations, one launch
constexpr int M3L_REDUCE_BATCH_MAX = 56;
struct ReduceBatchItem {
    const float* part;
    float* out[3];
    int G;
};
struct ReduceBatch {
    ReduceBatchItem it[M3L_REDUCE_BATCH_MAX];
    int count, D;
};
int m3l_reduce_rows_batch(const ReduceBatch* b, int accumulate, hipStream_t st);
int m3l_ln_bwd_blocks(int M);
int m3l_reduce_rows_seg3(const float* part, int G, int D, float* out0, float* out1, float* out2, int accumulate, hipStream_t st);

// grouped weight-gradient GEMMs: up to 4 problems dW_i[N_i,K_i] = Y_i^T X_i that share M (one launch + one reduce)
struct TnProblem {
    const void* Y;
    const void* X;
    int ldy, ldx, N, K;
    float* out;            // [nvalid, ldo] fp32
    int ldo, nvalid, kvalid;
    int tile0;             // filled by the launcher
    long part_off;         // filled by the launcher (float offset of this problem's slabs)
    int gelu_x;            // bf16 kernel only: X holds the PRE-activation u and the product is over GELU(u) (the hidden activation h was not saved):
                           // 1 = erf form (gelu_f, the per-sample block kernels), 2 = fitted form (gelu_fast, the row-tiled kernels); 0 = X as it is
};
struct TnGroup {
    TnProblem p[4];
    int count, tiles_total;
    // optional extra column reduction carried by the group's reduce launch: out[j] = sum_g part[g*width + j]
    // (the fc1 bias gradient: per-row-block column sums written by the fused dgrad epilogue)
    const float* extra_part;
    float* extra_out;
    int extra_G, extra_width;
};
// bf16 problems take the 256 x 64 TBT kernel of wgrad.hip (up to M3L_TN_MAX_PROBLEMS per launch); f32 keeps the 128 x 128 kernel of
// gemm.hip (launched in chunks of 4).  Extras: column reductions out[j] = sum_g part[g * width + j] carried by the reduce launch
// (the fc1 bias gradients: per-row-block column sums written by the fused dgrad epilogues).
#define M3L_TN_MAX_PROBLEMS 16
#define M3L_TN_MAX_EXTRAS 4
struct TnExtra {
    const float* part;
    float* out;
    int G, width;
};
size_t m3l_gemm_tn_grouped_ws_bytes(int M, const TnProblem* probs, int count);
int m3l_gemm_tn_grouped(int dtype, TnProblem* probs, int count, int M, float* partial_ws, size_t ws_bytes, int accumulate,
                        hipStream_t st, const TnExtra* extras = nullptr, int extra_count = 0);
size_t m3l_wgrad_ws_bytes(int M, const TnProblem* probs, int count);
// how many layers (each with the given `count` problems) one grouped launch should carry to give every CU a workgroup
int m3l_wgrad_layers_per_launch(int M, const TnProblem* layer_probs, int count);
int m3l_wgrad_bf16(TnProblem* probs, int count, int M, float* ws, size_t ws_bytes, int accumulate, hipStream_t st, const TnExtra* extras,
                   int extra_count);
int m3l_gemm_init();
int m3l_gemm_nt_colsum_rows(int M, int N);   // number of partial rows written through GemmEpi::colsum_part

// fused LN1 + QKV + attention + out-proj + residual + LN2 for short sequences (attn_block.hip)
int m3l_attn_block_supported(int dtype, int D, int heads, int n, int project_out);
int m3l_attn_block_bwd_enabled(void);
int m3l_attn_block_fwd(int D, int B, int n, const float* x, const float* ln1_w, const float* ln1_b, const void* wqkv, const void* wo,
                       const float* bo, const float* ln2_w, const float* ln2_b, float eps, void* xn1, void* qkv, void* o, float* lse,
                       float* x1, void* xn2, hipStream_t st);
// fused fc1 + GELU + fc2 + residual for short sequences (mlp_block.hip), companion of the attention block
int m3l_mlp_block_supported(int dtype, int D, int mlp, int n);
int m3l_mlp_block_fwd(int D, int mlp, int B, int n, const void* xn2, const float* x1, const void* w1, const float* b1, const void* w2,
                      const float* b2, void* u, void* h, float* xout, hipStream_t st);
int m3l_mlp_block_bwd_supported(int dtype, int D, int mlp, int n);
// dgrad chain of the feed-forward half + LN2 backward; cs_part [B][mlp] and ln_part [B][3 D] are per-sample partial rows
int m3l_mlp_block_bwd(int D, int mlp, int B, int n, const void* dxt, float* dx, const float* x1, const float* ln2_w, const void* u,
                      const void* w2T, const void* w1T, float eps, void* du, void* dx1t, float* cs_part, float* ln_part, hipStream_t st);
// dgrad chain of the attention half + LN1 backward (attn_block.hip); ln_part [B][3 D] per-sample partial rows; dxt_out may be null
int m3l_attn_block_bwd(int D, int B, int n, const void* dx1t, const float* dres, const float* x, const float* ln1_w, const void* qkv,
                       const void* o, const float* lse, const void* woT, const void* wqkvT, float eps, void* dqkv, float* dx_out, void* dxt_out,
                       float* ln_part, hipStream_t st);
// whole-stack launches for short sequences (enc_mega.hip): the bodies of the block kernels above, layer after layer in one launch
#define M3L_MEGA_MAX_LAYERS 16
#define M3L_MEGA_BWD_MAX_LAYERS 4        // = the largest weight-gradient group (TfWs::wg_batch)
int m3l_enc_mega_enabled(void);      // bit 1: forward, bit 2: backward groups (env M3L_ENC_MEGA, default 1)
int m3l_enc_fwd_mega(int D, int mlp, int B, int n, const float* x0, const void* const* layers, int count, float eps, hipStream_t st);
int m3l_enc_bwd_mega(int D, int mlp, int B, int n, float* dx, const void* const* layers, int count, float eps, hipStream_t st);
unsigned long long* m3l_attn_phase_buffer(void);
// row-tiled fused half layers (t192.hip), bf16, model width D = 192 / 256 / 384: a workgroup owns 192 / 128 / 96 token rows, any M
int m3l_mlp_t192_supported(int dtype, int D, int mlp, int M);
int m3l_mlp_t192_fwd(int D, int M, int mlp, const void* xn2, const float* x1, const void* w1, const float* b1, const void* w2, const float* b2,
                     void* u, void* h, float* xout, hipStream_t st);
// LN1 + QKV + attention per sample for 48 < n <= 192 (D = 192 / 3 heads, D = 256 / 4 heads): xn1, qkv, o, lse as the per-op kernels write them
int m3l_attn_t192_fwd_supported(int dtype, int D, int heads, int n, int B);
int m3l_attn_t192_fwd(int D, int B, int n, const float* x, const float* ln_w, const float* ln_b, const void* wqkv, float eps, void* xn1, void* qkv,
                      void* o, float* lse, hipStream_t st);
// dO = dx1_t Wo + the whole attention backward of a sample -> dqkv (same support as the forward)
int m3l_attn_t192_bwd(int D, int B, int n, const void* dx1t, const void* qkv, const void* o, const float* lse, const void* woT, void* dqkv,
                      hipStream_t st);
int m3l_attn_tail_mlp_t192_supported(int dtype, int D, int HD, int mlp, int M);
int m3l_attn_tail_mlp_t192_fwd(int D, int M, int mlp, const void* o, const float* x, const void* wo, const float* bo, const float* ln2_w,
                               const float* ln2_b, float eps, float* x1, void* xn2, const void* w1, const float* b1, const void* w2,
                               const float* b2, void* u, void* h, float* xout, hipStream_t st);
int m3l_mlp_t192_short(void);        // 1: use the row-tiled MLP kernels for short sequences too
int m3l_mlp_t192_tiles(int D, int M);     // partial rows written to ln_part [tiles][3 D]
int m3l_mlp_t192_cs_rows(int D, int M);   // partial rows written to cs_part [rows][mlp]: tiles (D = 192) or tiles x waves
int m3l_mlp_t192_bwd(int D, int M, int mlp, const void* dxt, float* dx, const float* x1, const float* ln2_w, const void* u, const void* w2T,
                     const void* w1T, float eps, void* du, void* dx1t, float* cs_part, float* ln_part, hipStream_t st);
// dxn1 = dqkv Wqkv + LN1 backward per row tile; ln_part [tiles][3 D]
int m3l_qkv_bwd_t192_supported(int dtype, int D, int K, int M);
int m3l_qkv_bwd_t192_tiles(int D, int M);
int m3l_qkv_bwd_t192(int D, int M, int K, const void* dqkv, const float* x, const float* ln1_w, const void* wqkvT, const float* dres, float eps,
                     float* dx_out, void* dxt_out, float* ln_part, hipStream_t st);
int m3l_attn_fwd(int dtype, const void* qkv, void* o, float* lse, int B, int n, int H, hipStream_t st);
int m3l_attn_bwd(int dtype, const void* qkv, const void* o, const void* dO, const float* lse, float* dsum, void* dqkv, int B, int n,
                 int H, hipStream_t st);

int m3l_ln_fwd(int out_dtype, const float* x, int M, int D, const float* gamma, const float* beta, float eps, void* y, float* y32,
               hipStream_t st);
// dx_out (f32) and/or dx_t_out (compute type ct_dtype) = dres + dLN/dx; dbias (optional) = column sums of that result
int m3l_ln_bwd(int dy_dtype, const void* dy, const float* x, int M, int D, const float* gamma, float eps, const float* dres,
               float* dx_out, void* dx_t_out, int ct_dtype, float* part_ws, float* dgamma, float* dbeta, float* dbias, int accumulate,
               hipStream_t st);
int m3l_reduce_rows(const float* part, int G, int stride, int count, float* out, int accumulate, hipStream_t st);
int m3l_colsum(int dtype, const void* Y, int M, int N, int ld, float* part_ws, float* out, int accumulate, hipStream_t st);
int m3l_prep_weights(int dtype, const WeightPack* pack_host, hipStream_t st);
// batching of the modules' weight copies over one call chain: 0 = off, 1 = collect (modules return after recording), 2 = skip (done)
int m3l_prep_mode(void);
void m3l_prep_set_mode(int mode);
int m3l_prep_flush(hipStream_t st);
int m3l_axpy_t(int dtype, const float* x, const void* o, long count, float* out, hipStream_t st);          // out = x + (float)o
int m3l_cast_f32(int dtype, const float* x, long count, void* out, hipStream_t st);                        // out = (T)x
struct m3l_tf_cfg;
extern "C" int m3l_transformer_rb(const m3l_tf_cfg* c, int B, int n);       // 1: the stack runs the bf16 residual stream at this shape
int m3l_cast_bf16_f32(const void* x, long count, float* out, hipStream_t st);                               // out = (float)x
int m3l_scale_by_dev(int dtype, const void* x, long count, const float* scale_dev, void* out, hipStream_t st);  // out = x * *scale
int m3l_mask_scale(int mode, const float* src, const void* ref, float scale, long count, float* out, hipStream_t st);
int m3l_vit_tokens(const float* emb, const float* cls, const float* regs, const float* pos, int B, int npatch, int R, int D, float* tok,
                   hipStream_t st);
int m3l_concat2(float* a, int na, float* b2, int nb, int rows, float* cat, int split, hipStream_t st);
int m3l_vt_load_launch(const void* image_nhwc, int image_u8, int B, int H, int W, int C, float img_lo, float img_span, float* image_nchw,
                       const void* tactile, int tactile_u8, int th, int tw, int n_sensors, int frame_stack, float tac_lo, float tac_span,
                       float* const* tactile_out, hipStream_t st);
struct ConvSrc {
    const void* src[M3L_MAX_SENSORS];   // nchw: f32 [B, Ci, H, W] per source (sources concatenated on batch); else one NHWC compute-type [Btot*H*W, Ci]
    int nsrc, nchw;
};
int m3l_im2col(int dtype, const ConvSrc* src, int Bsrc, int Ci, int H, int W, int KH, int S, int P, int OH, int OW, int Kpad, void* col,
               hipStream_t st);
// direct (implicit-GEMM) convolutions of the EarlyCNN stem (conv.hip), bf16
int m3l_conv_direct_supported(int dtype, int Ci, int Co, int KH, int S, int P, int first_layer);
size_t m3l_conv_wf_elems(int Ci, int Co, int KH);
size_t m3l_conv_wd_elems(int Ci, int Co, int KH, int S);
size_t m3l_conv_wgrad_slab_elems(int B, int Ci, int H, int W, int Co, int KH, int S, int P);
int m3l_conv_prep(const float* W, int Ci, int Co, int KH, int S, void* Wf, void* Wd, hipStream_t st);
int m3l_conv_fwd(const ConvSrc* src, int Bsrc, int B, int Ci, int H, int W, int Co, int KH, int S, int P, const void* Wf, const float* bias,
                 void* out, hipStream_t st);
int m3l_conv_wgrad(const ConvSrc* src, int Bsrc, int B, int Ci, int H, int W, int Co, int KH, int S, int P, const void* dY, float* slab,
                   float* dW, hipStream_t st);
int m3l_conv_dgrad(const void* dY, int B, int Ci, int H, int W, int Co, int KH, int S, int P, const void* Wd, const void* act, void* dX,
                   hipStream_t st);
int m3l_col2im_relu(int dtype, const void* dcol, int Btot, int Ci, int H, int W, int KH, int S, int P, int OH, int OW, int Kpad,
                    const void* act, void* dX, hipStream_t st);
int k_tokens_assemble(const float* img_tok, const float* tac_tok, int B, int D, int n_img, int n_tac, int k, const float* mod,
                        const float* pos_img, const float* pos_tac, float* tokens, hipStream_t st);
int k_tokens_assemble_bwd(const float* dtok, int B, int D, int n_img, int n_tac, int k, float* d_img, float* d_tac, float* part_ws,
                            float* dmod, int accumulate, hipStream_t st);
int m3l_adam_flat(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd, int step,
                  float gscale, hipStream_t st);
int m3l_adamw_flat(float* p, float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd, int step, float gscale,
                   float max_norm, float* norm_ws, int scale_grads, hipStream_t st);
int m3l_adam_flat_dev(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd,
                      int* step_dev, float* bc_dev, hipStream_t st);
struct MaskRankGroup { const float* noise; int n, nm, token_offset, masked_off, unmasked_off; };
struct MaskRankArgs { MaskRankGroup g[M3L_MAX_SENSORS + 1]; int count; };
int m3l_mask_rank_groups(const MaskRankArgs* args, int B, int64_t* masked, int masked_ld, int64_t* unmasked, int unmasked_ld, hipStream_t st);
int m3l_mask_rank(const float* noise, int B, int n, int nm, int token_offset, int64_t* masked, int masked_ld, int masked_off,
                  int64_t* unmasked, int unmasked_ld, int unmasked_off, hipStream_t st);
int m3l_patch_ln(int dtype, const PatchGroup* pg, const int64_t* idx, int idx_ld, int j0, int cnt, int B, const float* gamma,
                 const float* beta, float eps, void* xn, int pdpad, hipStream_t st);
int m3l_patch_ln_bwd(int dtype, const PatchGroup* pg, const int64_t* idx, int idx_ld, int j0, int cnt, int B, float eps,
                     const void* dxn, int pdpad, float* part_ws, float* dgamma, float* dbeta, int accumulate, hipStream_t st);
int m3l_embed_finalize(const float* E, int D, const PatchGroup* pg, const int64_t* idx, int idx_ld, int j0, int cnt, int B,
                       const float* gamma, const float* beta, float eps, const float* mod, int mod0, const float* pos, float* tokens,
                       int L, hipStream_t st);
int m3l_embed_finalize_bwd(int dtype, const float* dtok, int L, const float* E, int D, const PatchGroup* pg, const int64_t* idx,
                           int idx_ld, int j0, int cnt, int B, const float* gamma, float eps, void* dE, float* part_ws, float* dgamma,
                           float* dbeta, float* dmod, int mod0, int accumulate, hipStream_t st);
int k_unshuffle_fwd(const float* src, const float* mask_token, const int64_t* unmasked, int nvis, const int64_t* masked, int nmask,
                      int B, int dd, int n_img, int n_tac, const float* dmod, const float* pos_img, const float* pos_tac, float* dec_in,
                      hipStream_t st);
int k_unshuffle_bwd(const float* dY, const int64_t* unmasked, int nvis, const int64_t* masked, int nmask, int B, int dd, int n_img,
                      int n_tac, int nmod, float* dsrc, float* part_ws, float* dmask_token, float* ddmod, int accumulate, hipStream_t st);
int m3l_gather_rows(int dtype, const void* src, int N, int D, const int64_t* idx, int idx_ld, int j0, int cnt, int B, void* dst,
                    hipStream_t st);
int m3l_gather_rows2(int dtype, const void* src, int N, int D, const int64_t* idx, int idx_ld, int cnt0, int cnt1, int B, void* dst0, void* dst1,
                     hipStream_t st);
int m3l_scatter_rows2(int dtype, const void* src0, const void* src1, int N, int D, const int64_t* idx, int idx_ld, int cnt0, int cnt1, int B,
                      const float* scale_dev, void* dst, hipStream_t st);
int m3l_scatter_rows(int dtype, const void* src, int N, int D, const int64_t* idx, int idx_ld, int j0, int cnt, int B, void* dst,
                     hipStream_t st);
int m3l_mse(int dtype, const float* pred, int pdpad, const PatchGroup* pg, const int64_t* idx, int idx_ld, int j0, int cnt, int B,
            float weight, float* part_ws, int* nblocks_out, void* dpred, float* target_out, hipStream_t st);
