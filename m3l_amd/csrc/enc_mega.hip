// Whole-stack launch for short sequences (the MAE encoder, n <= 48): every layer of a transformer forward in ONE launch, by calling the
// bodies of the per-half-layer block kernels (attn_block.hip, mlp_block.hip) back to back inside one workgroup per sample.  Nothing about
// the arithmetic or the saved activations changes — each body still reads its inputs from and writes its outputs to global memory exactly
// as its own kernel does (the next body finds them in this CU's L2) — what goes away is one launch per half layer: 24 launches -> 1 for
// the forward of the ViT-Tiny encoder.  Measured: 63.1 k vs 63.2 k samples/s with / without (the workgroup ramp of a launch is ~1 us of a
// 17-25 us body; the bodies are chains of dependent steps inside), 0.1 ms less host enqueue time per step.  The backward keeps its per-
// half-layer launches: the side stream's grouped weight-gradient GEMMs are ordered against them per layer group.
//
// Between two bodies: __syncthreads() (workgroup-scope release / acquire + barrier).  All 13 waves of the workgroup run on one CU and
// share its vector L1; the global tensors exchanged between bodies are written once and read afterwards within the launch, so the
// workgroup-scope ordering is all that is needed.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#define M3L_BLOCK_BODIES_ONLY
// Every body derives its lane / wave indices and all its addresses from the thread id.  Inlined into the layer loop those are loop
// invariants: the compiler hoists them all in front of the loop and spills them (74-170 VGPRs, -3 % end to end).  An opaque copy of the
// thread id per body call keeps each body's address arithmetic inside the body.
static __device__ __forceinline__ int m3l_body_tid() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
#define M3L_BODY_TID() m3l_body_tid()
#include "attn_block.hip"
#include "mlp_block.hip"
#undef M3L_BLOCK_BODIES_ONLY

namespace {

static_assert(AB_THREADS == MB_THREADS, "the attention and feed-forward bodies share one workgroup shape");

struct MegaFwdLayer {
    const float *ln1_w, *ln1_b;
    const bf16 *wqkv, *wo;
    const float *bo, *ln2_w, *ln2_b;
    bf16 *xn1, *qkv, *o;
    float *lse, *x1;
    bf16* xn2;
    const bf16* w1;
    const float* b1;
    const bf16* w2;
    const float* b2;
    bf16 *u, *h;
    float* xout;
};
struct MegaFwdPack {
    MegaFwdLayer L[M3L_MEGA_MAX_LAYERS];
    int count;
};

// R: storage type of the residual stream (the float* x / x1 / xout of the pack then point at bf16 data: m3l_set_residual_bf16)
template <int KT, typename R>
__global__ __launch_bounds__(AB_THREADS) void enc_fwd_mega_kernel(const float* __restrict__ x0, MegaFwdPack P, float eps, int n, int mlp,
                                                                    unsigned long long* __restrict__ phase_ts) {
    // (Odd workgroups started 8 - 40 K cycles late, so that the phases of neighbouring CUs interleave instead of running chip-wide in
    // lockstep: 508 - 510 us with every delay, as without — the bodies are chains of dependent steps per sample, not bandwidth-bound phases.)
    // one body per loop iteration (even steps: attention half, odd steps: feed-forward half): with both bodies inlined one after the
    // other in the loop body the register allocator spills 74-118 VGPRs; as two branches of one loop their live ranges stay apart
    const float* x = x0;
    for (int step = 0; step < 2 * P.count; ++step) {
        const MegaFwdLayer& L = P.L[step >> 1];
        if ((step & 1) == 0) {
            attn_block_fwd_body<KT, R>(reinterpret_cast<const R*>(x), L.ln1_w, L.ln1_b, L.wqkv, L.wo, L.bo, L.ln2_w, L.ln2_b, eps, n, L.xn1, L.qkv, L.o, L.lse,
                                       reinterpret_cast<R*>(L.x1), L.xn2, step == 0 ? phase_ts : nullptr);
        } else {
            mlp_block_fwd_body<KT, R>(L.xn2, reinterpret_cast<const R*>(L.x1), L.w1, L.b1, L.w2, L.b2, n, mlp, L.u, L.h, reinterpret_cast<R*>(L.xout));
            x = L.xout;
        }
        __syncthreads();
    }
}

template <int KT> constexpr int mega_fwd_lds(int mlp) {     // the feed-forward body keeps b1 [mlp] behind its fixed layout
    const int a = AbLayout<KT>::TOTAL, b = MbLayout<KT>::TOTAL + mlp * (int)sizeof(float);
    return a > b ? a : b;
}

// Backward of a GROUP of layers (the layers whose weight gradients go out as one grouped launch on the side stream) in one launch: per
// layer the feed-forward backward body, then the attention backward body, top layer first.  Every tensor a body hands to the next one (the
// running residual gradient dx, updated in place; dx1_t; dx_t of the layer below) is per-sample rows that this workgroup wrote itself.
// The idea was that the backward runs beside the side stream's weight-gradient workgroups, and at every kernel boundary of the compute
// stream the CUs that drain first are handed to those (each then holds its CU for its whole duration: no two of these workgroups fit one
// CU) — fewer boundaries, fewer hand-overs.  Measured (round 3, A/B in one job): 64.3-64.4 k samples/s per half layer, 63.5-63.6 k with the
// groups fused.  The weight gradients need their CU-time wherever they get it; the boundary tails are where they get it without displacing
// anything.  Bit-identical (test_whole_stack_forward_launch_is_bit_identical), opt-in: m3l_set_enc_mega(3) / M3L_ENC_MEGA=3.
struct MegaBwdLayer {
    const bf16* dxt;                // feed-forward half: dx_t of this layer, saved x1, gamma2, u, W2^T, W1^T -> du, dx1_t, partial rows
    const float *x1, *ln2_w;
    const bf16 *u, *w2T, *w1T;
    bf16 *du, *dx1t;
    float *cs_part, *ln2_part;
    const float *x, *ln1_w;         // attention half: the layer's input, gamma1, saved qkv / o / lse, Wo^T, Wqkv^T -> dqkv, dx, dx_t below
    const bf16 *qkv, *o;
    const float* lse;
    const bf16 *woT, *wqkvT;
    bf16* dqkv;
    float* dx_out;
    bf16* dxt_out;
    float* ln1_part;
};
struct MegaBwdPack {
    MegaBwdLayer L[M3L_MEGA_BWD_MAX_LAYERS];
    int count;
};

template <int KT, typename R>
__global__ __launch_bounds__(AB_THREADS) void enc_bwd_mega_kernel(float* __restrict__ dx, MegaBwdPack P, float eps, int n, int mlp) {
    for (int step = 0; step < 2 * P.count; ++step) {
        const MegaBwdLayer& L = P.L[step >> 1];
        if ((step & 1) == 0)
            mlp_block_bwd_body<KT, R>(L.dxt, reinterpret_cast<R*>(dx), reinterpret_cast<const R*>(L.x1), L.ln2_w, L.u, L.w2T, L.w1T, eps, n, mlp, L.du, L.dx1t,
                                      L.cs_part, L.ln2_part);
        else
            attn_block_bwd_body<KT, R>(L.dx1t, reinterpret_cast<const R*>(dx), reinterpret_cast<const R*>(L.x), L.ln1_w, L.qkv, L.o, L.lse, L.woT, L.wqkvT, eps, n,
                                       L.dqkv, reinterpret_cast<R*>(L.dx_out), L.dxt_out, L.ln1_part);
        __syncthreads();
    }
}

template <int KT> constexpr int mega_bwd_lds(int mlp) {
    const int a = AbBwdLayout<KT>::TOTAL, b = MbLayout<KT>::TOTAL + 3 * mlp * (int)sizeof(float);
    return a > b ? a : b;
}

}  // namespace

static int g_enc_mega = -1;          // -1 = environment not read yet
int m3l_enc_mega_enabled(void) {
    if (g_enc_mega < 0) g_enc_mega = getenv("M3L_ENC_MEGA") ? atoi(getenv("M3L_ENC_MEGA")) : 1;      // bit 1: forward, bit 2: backward groups (opt-in: measured slower)
    return g_enc_mega;
}
// bit 1 = one launch for the whole forward of a short-sequence stack (default), bit 2 = one launch per weight-gradient group of layers
// in its backward (opt-in), 0 = one launch per half layer; returns the previous setting
extern "C" int m3l_set_enc_mega(int mode) {
    const int old = m3l_enc_mega_enabled();
    g_enc_mega = mode > 0 ? mode : 0;
    return old;
}

// x0 [B, n, D] fp32; layers[i] = the 20 pointers of MegaFwdLayer in declaration order
int m3l_enc_fwd_mega(int D, int mlp, int B, int n, const float* x0, const void* const* layers, int count, float eps, hipStream_t st) {
    static int inited_mlp = 0;
    if (inited_mlp != mlp) {
        M3L_HIP(hipFuncSetAttribute((const void*)enc_fwd_mega_kernel<2, float>, hipFuncAttributeMaxDynamicSharedMemorySize, mega_fwd_lds<2>(mlp)));
        M3L_HIP(hipFuncSetAttribute((const void*)enc_fwd_mega_kernel<3, float>, hipFuncAttributeMaxDynamicSharedMemorySize, mega_fwd_lds<3>(mlp)));
        M3L_HIP(hipFuncSetAttribute((const void*)enc_fwd_mega_kernel<2, bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, mega_fwd_lds<2>(mlp)));
        M3L_HIP(hipFuncSetAttribute((const void*)enc_fwd_mega_kernel<3, bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, mega_fwd_lds<3>(mlp)));
        inited_mlp = mlp;
    }
    M3L_CHECK(D == 128 || D == 192, "enc_fwd_mega: D=%d unsupported", D);
    M3L_CHECK(count >= 1 && count <= M3L_MEGA_MAX_LAYERS, "enc_fwd_mega: %d layers (max %d)", count, M3L_MEGA_MAX_LAYERS);
    static_assert(sizeof(MegaFwdLayer) == 20 * sizeof(void*), "MegaFwdLayer is 20 pointers");
    MegaFwdPack P;
    memcpy(P.L, layers, (size_t)count * sizeof(MegaFwdLayer));
    P.count = count;
    const double rsz = m3l_call_rb() ? 2.0 : 4.0;
    ProfScope prof("enc_fwd_mega", B, n, count, (double)count * (2.0 * B * n * (4.0 * D * D) + 4.0 * B * (D / 64) * (double)n * n * 64 + 4.0 * B * n * (double)D * mlp), st,
                   (double)count * B * n * (D * (12.0 + 3.0 * rsz) + mlp * 4.0));        // per layer: xn1, qkv, o, xn2 | x, x1, xout | u, h (each once)
    if (m3l_call_rb()) {
        if (D == 128) enc_fwd_mega_kernel<2, bf16><<<B, AB_THREADS, mega_fwd_lds<2>(mlp), st>>>(x0, P, eps, n, mlp, m3l_attn_phase_buffer());
        else enc_fwd_mega_kernel<3, bf16><<<B, AB_THREADS, mega_fwd_lds<3>(mlp), st>>>(x0, P, eps, n, mlp, m3l_attn_phase_buffer());
    } else if (D == 128)
        enc_fwd_mega_kernel<2, float><<<B, AB_THREADS, mega_fwd_lds<2>(mlp), st>>>(x0, P, eps, n, mlp, m3l_attn_phase_buffer());
    else
        enc_fwd_mega_kernel<3, float><<<B, AB_THREADS, mega_fwd_lds<3>(mlp), st>>>(x0, P, eps, n, mlp, m3l_attn_phase_buffer());
    M3L_LAUNCH_CHECK();
    return 0;
}

// Backward of `count` consecutive layers (top first).  layers[i] = the 21 pointers of MegaBwdLayer in declaration order; dx [B, n, D] fp32
// is the running residual gradient (read and updated in place by every body).
int m3l_enc_bwd_mega(int D, int mlp, int B, int n, float* dx, const void* const* layers, int count, float eps, hipStream_t st) {
    static int inited_mlp = 0;
    M3L_CHECK(D == 128 || D == 192, "enc_bwd_mega: D=%d unsupported", D);
    M3L_CHECK(count >= 1 && count <= M3L_MEGA_BWD_MAX_LAYERS, "enc_bwd_mega: %d layers (max %d)", count, M3L_MEGA_BWD_MAX_LAYERS);
    const int lds = D == 128 ? mega_bwd_lds<2>(mlp) : mega_bwd_lds<3>(mlp);
    M3L_CHECK(lds <= 160 * 1024, "enc_bwd_mega: mlp=%d needs %d bytes of LDS", mlp, lds);
    M3L_CHECK(!m3l_call_rb(), "enc_bwd_mega: the grouped backward launch does not run the bf16 residual mode (per-half-layer launches do)");
    if (inited_mlp != mlp) {
        M3L_HIP(hipFuncSetAttribute((const void*)enc_bwd_mega_kernel<2, float>, hipFuncAttributeMaxDynamicSharedMemorySize, mega_bwd_lds<2>(mlp)));
        M3L_HIP(hipFuncSetAttribute((const void*)enc_bwd_mega_kernel<3, float>, hipFuncAttributeMaxDynamicSharedMemorySize, mega_bwd_lds<3>(mlp)));
        inited_mlp = mlp;
    }
    static_assert(sizeof(MegaBwdLayer) == 21 * sizeof(void*), "MegaBwdLayer is 21 pointers");
    MegaBwdPack P;
    memcpy(P.L, layers, (size_t)count * sizeof(MegaBwdLayer));
    P.count = count;
    ProfScope prof("enc_bwd_mega", B, n, count, (double)count * (2.0 * B * n * (4.0 * D * D) + 10.0 * B * (D / 64) * (double)n * n * 64 + 4.0 * B * n * (double)D * mlp), st);
    if (D == 128)
        enc_bwd_mega_kernel<2, float><<<B, AB_THREADS, lds, st>>>(dx, P, eps, n, mlp);
    else
        enc_bwd_mega_kernel<3, float><<<B, AB_THREADS, lds, st>>>(dx, P, eps, n, mlp);
    M3L_LAUNCH_CHECK();
    return 0;
}
