// One launch for the attention half of a pre-norm transformer layer on a SHORT sequence (the MAE encoder: n = 48 visible tokens):
//
//     xn1 = LN1(x);  qkv = xn1 Wqkv^T;  o = softmax(q k^T / 8) v  per head;  x1 = x + o Wo^T + bo;  xn2 = LN2(x1)
//
// (vit_pytorch Attention.forward + the residual + the FeedForward's LayerNorm, models/pretrain_models.py:266).  The unfused path is
// five dependent launches of 6-10 us each for 2-3 us worth of traffic; at M = B * 48 rows every one of them is launch / latency
// bound.  Here one workgroup owns one sample: the sample's tokens never leave the CU between the five stages, the weights
// (Wqkv, Wo: L2-resident) stream through a 2-stage LDS ring filled by a DMA-only wave, and every tensor the backward needs
// (xn1, qkv, o, lse, x1, xn2) is written exactly as the unfused kernels write it, so the backward is unchanged.
//
// Workgroup = 13 waves: 12 compute waves (3 per SIMD: with one workgroup per CU, that is all the latency hiding there is), wave 12
// only issues LDS-DMA (a wave that also reads LDS or stores gets its vmcnt waits
// drained by the compiler / by its own stores — see EXPERIMENTS.md 2).  bf16 operands, fp32 accumulation and statistics.
// Supported: D = 64 KT (KT = 2, 3: LDS), heads = D / 64 (so to_out is a real projection with K = D), n <= 48 (3 row tiles).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdlib.h>

#include "common.cuh"
#include "kernels.h"

#ifndef M3L_BODY_INLINE
#define M3L_BODY_INLINE __forceinline__
#endif
#ifndef M3L_BODY_TID
#define M3L_BODY_TID() ((int)threadIdx.x)     // enc_mega.hip launders it per body call (keeps loop-invariant code motion out of the layer loop)
#endif

namespace {

constexpr int AB_CW = 12;                 // compute waves: (column tile 0..3) x (row tile 0..2) in the projections, (head, query tile) in attention
constexpr int AB_THREADS = 64 * (AB_CW + 1);   // + the DMA wave
constexpr float AB_SCALE = 0.125f;        // dim_head ** -0.5, dim_head = 64

// (row16_sum — common.cuh — gives each LayerNorm ROW to one 16-lane group: a wave works on 4 rows at a time, and their two reductions per
// row cost 8 DPP adds instead of 12 dependent ds_bpermute shuffles — with one wave per SIMD nothing else hides that latency.)

template <int KT> struct AbLayout {
    static constexpr int D = 64 * KT;
    static constexpr int XN_PITCH = D * 2 + 16;          // bytes; (pitch / 16) odd -> conflict-free ds_read_b128 over 16 rows
    static constexpr int QKV_PITCH = 3 * D * 2 + 16;
    static constexpr int Y_PITCH = D * 4 + 16;
    static constexpr int ROWS = 48;                       // n <= 48: three 16-row tiles
    static constexpr int XN_BYTES = ROWS * XN_PITCH;      // xn1, later o
    static constexpr int QKV_BYTES = ROWS * QKV_PITCH;    // q | k | v, later y (f32, needs ROWS * Y_PITCH <= QKV_BYTES)
    static constexpr int WBLK = KT * 64 * 128;            // one 64-row weight block: KT sub-tiles [64 rows][128 B], chunk ^ (row & 7)
    static constexpr int NSTAGE = 3;                      // weight ring: two blocks in flight while one is multiplied
    static constexpr int TOTAL = XN_BYTES + QKV_BYTES + NSTAGE * WBLK;
    static_assert(ROWS * Y_PITCH <= QKV_BYTES, "y must fit where qkv was");
    static_assert(KT == 2 || KT == 3, "vmcnt immediates in the DMA wave are 16 / 24");
};

// (the body is a device function: attn_block_fwd_kernel wraps it, and enc_mega.hip calls it for every layer of a stack in one launch)
template <int KT, typename R>
__device__ M3L_BODY_INLINE void attn_block_fwd_body(
    const R* __restrict__ x, const float* __restrict__ ln1_w, const float* __restrict__ ln1_b, const bf16* __restrict__ Wqkv,
    const bf16* __restrict__ Wo, const float* __restrict__ bo, const float* __restrict__ ln2_w, const float* __restrict__ ln2_b,
    float eps, int n, bf16* __restrict__ xn1_out, bf16* __restrict__ qkv_out, bf16* __restrict__ o_out, float* __restrict__ lse_out,
    R* __restrict__ x1_out, bf16* __restrict__ xn2_out, unsigned long long* __restrict__ phase_ts) {
    using Ly = AbLayout<KT>;
    constexpr int D = Ly::D, H = KT, KSTEPS = 2 * KT;
    constexpr int NB_QKV = 3 * KT, NB = 4 * KT;          // 64-row weight blocks: Wqkv then Wo
    // profiling hook (m3l_set_attn_phase_buffer, null in production): shader-clock stamps of thread 0 at the phase boundaries
    // [start, LN1 done, QKV done, attention done, out-proj done, end] -> phase_ts[block][8]
    auto stamp = [&](int i) {
        if (phase_ts && threadIdx.x == 0) phase_ts[(long)blockIdx.x * 8 + i] = __builtin_readcyclecounter();
    };
    constexpr int NDMA = 8 * KT;                          // LDS-DMA instructions per block (1 KiB each)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* XN = smem;                                      // [48][XN_PITCH]  bf16 xn1, later o
    char* QKV = smem + Ly::XN_BYTES;                      // [48][QKV_PITCH] bf16 q|k|v, later y f32 [48][Y_PITCH]
    char* WR = QKV + Ly::QKV_BYTES;                       // NSTAGE x WBLK
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef __attribute__((address_space(1))) const void* gl_vp;

    const int tid = M3L_BODY_TID(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int b = blockIdx.x;
    const long row0 = (long)b * n;
    const int RT = (n + 15) >> 4;                         // 16-row tiles that hold real rows

    stamp(0);
    if (wave == AB_CW) {
        // ------------------------------------------------------------------ DMA wave
        const int srow = lane >> 3, spc = lane & 7;
        auto issue = [&](int blk) {
#ifdef AB_ABL
            if (AB_ABL & 1) return;                       // diagnostic build (tools/t192_ablate.sh): no weight stream
#endif
            const bf16* Wsrc = blk < NB_QKV ? Wqkv + (long)blk * 64 * D : Wo + (long)(blk - NB_QKV) * 64 * D;
            char* dst = WR + (blk % Ly::NSTAGE) * Ly::WBLK;
#pragma unroll
            for (int rg = 0; rg < 8; ++rg) {
                const bf16* src = Wsrc + (long)(8 * rg + srow) * D + ((spc ^ srow) << 3);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
                    __builtin_amdgcn_global_load_lds((gl_vp)(src + kt * 64), (lds_vp)(dst + kt * 8192 + rg * 1024), 16, 0, 0);
            }
        };
        issue(0);
        issue(1);
        __builtin_amdgcn_s_barrier();                                     // B0 (xn1 ready; nothing to do here)
        for (int blk = 0; blk < NB; ++blk) {
            if (blk == NB_QKV) {
                __builtin_amdgcn_s_barrier();                             // B1 (qkv complete)
                __builtin_amdgcn_s_barrier();                             // B2 (o ready)
            }
            if (blk + 1 < NB) {                                           // loads retire in order: only block blk+1 may remain
                if (NDMA == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                                 // R_blk: block landed; every compute wave is done with block blk-1
            if (blk + 2 < NB) issue(blk + 2);                             // into the stage block blk-1 occupied
        }
        __builtin_amdgcn_s_barrier();                                     // B3 (y complete)
        return;
    }

    // ---------------------------------------------------------------------- compute waves
    // ---- LN1: wave w owns rows 4 w .. 4 w + 3 (row = 4 w + g): the row's 16 lanes hold KT float4 chunks (columns 4 (li + 16 c)).
    // The same thread finishes the same row chunks at the end (residual + LN2): x stays in registers and bo / gamma2 / beta2 are
    // requested before the out-projection, so that the tail has no memory round trip left in it.
    f32x4 x0[KT], bor[KT], g2r[KT], b2r[KT];
    {
        const int r = 4 * wave + g;
        f32x4 xr[KT];
#pragma unroll
        for (int c = 0; c < KT; ++c) {
            xr[c] = (r < n) ? ld_res4<R>(x + (row0 + r) * D + 4 * (li + 16 * c)) : f32x4{0.f, 0.f, 0.f, 0.f};
            x0[c] = xr[c];
        }
        float s1 = 0.f;
#pragma unroll
        for (int c = 0; c < KT; ++c) s1 += (xr[c][0] + xr[c][1]) + (xr[c][2] + xr[c][3]);
        const float mean = row16_sum(s1) / D;
        float s2 = 0.f;
#pragma unroll
        for (int c = 0; c < KT; ++c) {
            const f32x4 d = xr[c] - mean;
            xr[c] = d;
            s2 += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
        const float rstd = rsqrtf(row16_sum(s2) / D + eps);
#pragma unroll
        for (int c = 0; c < KT; ++c) {
            const f32x4 y = xr[c] * rstd * *reinterpret_cast<const f32x4*>(ln1_w + 4 * (li + 16 * c)) + *reinterpret_cast<const f32x4*>(ln1_b + 4 * (li + 16 * c));
            bf16x4 pk;
            pk[0] = (bf16)y[0]; pk[1] = (bf16)y[1]; pk[2] = (bf16)y[2]; pk[3] = (bf16)y[3];
            if (r >= n) pk[0] = pk[1] = pk[2] = pk[3] = (bf16)0.f;       // padding rows: zeros (their products are never stored)
            *reinterpret_cast<bf16x4*>(XN + r * Ly::XN_PITCH + 8 * (li + 16 * c)) = pk;
            if (r < n) *reinterpret_cast<bf16x4*>(xn1_out + (row0 + r) * D + 4 * (li + 16 * c)) = pk;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                         // B0: xn1 in LDS
    stamp(1);

    // ---- a 64-column block of A[48 x D] (LDS, row pitch PITCH) times W block^T: wave (ct, rt) owns one 16 x 16 tile
    // The A fragments (this wave's 16 rows, all k) are the same for every weight block of a projection: read once (load_a).  Per block
    // the KSTEPS weight fragments are requested together and feed two accumulator chains — one ds_read + MFMA pair after the other on
    // a single accumulator left the LDS latency and the MFMA dependency exposed KSTEPS times per block (22 % MFMA rate in this phase).
    const int ct = wave & 3, rt = wave >> 2;
    Frag<bf16> fa_c[KSTEPS];
    auto load_a = [&](const char* Abase, int pitch) {
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) fa_c[ks].v = *reinterpret_cast<const bf16x8*>(Abase + (16 * rt + li) * pitch + (ks * 32 + 8 * g) * 2);
    };
    auto block_mma = [&](const char* Wb) {
        const int wrow = 16 * ct + li;
        Frag<bf16> fw[KSTEPS];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
            fw[ks].v = *reinterpret_cast<const bf16x8*>(Wb + (ks >> 1) * 8192 + wrow * 128 + ((((ks & 1) * 4 + g) ^ (wrow & 7)) << 4));
        f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks += 2) {
            a0 = mma16(fa_c[ks], fw[ks], a0);
            a1 = mma16(fa_c[ks + 1], fw[ks + 1], a1);
        }
        return a0 + a1;
    };
    if (rt < RT) load_a(XN, Ly::XN_PITCH);

    // ---- QKV projection: 3 KT blocks; results (C layout: row 4g + r, col li) -> QKV LDS as bf16
    for (int blk = 0; blk < NB_QKV; ++blk) {
        __builtin_amdgcn_s_barrier();                                     // R_blk
        if (rt < RT) {
            const f32x4 acc = block_mma(WR + (blk % Ly::NSTAGE) * Ly::WBLK);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<bf16*>(QKV + (16 * rt + 4 * g + r) * Ly::QKV_PITCH + (64 * blk + 16 * ct + li) * 2) = (bf16)acc[r];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                         // B1: q | k | v complete in LDS
    stamp(2);

    // ---- qkv -> global (what the unfused to_qkv GEMM writes): 16-byte row segments, all compute threads
    {
        constexpr int CPR = 3 * D / 8;                                    // 16-byte chunks per row
        for (int id = tid; id < n * CPR; id += 64 * AB_CW) {
            const int r = id / CPR, c = id % CPR;
            *reinterpret_cast<uint4*>(qkv_out + (row0 + r) * 3 * D + c * 8) = *reinterpret_cast<const uint4*>(QKV + r * Ly::QKV_PITCH + c * 16);
        }
    }

    // ---- attention: wave w = head w (H <= 4).  S^T = K Q^T per 16-query tile, online softmax over 32-key tiles, O^T = V^T P^T.
    if (wave < H * RT) {
        const int hh = wave / RT;
        const char* Qb = QKV + hh * 128;                                  // column offset of q_h (64 bf16 = 128 B)
        const char* Kb = QKV + (D + hh * 64) * 2;
        const bf16* Vb = reinterpret_cast<const bf16*>(QKV + (2 * D + hh * 64) * 2);
        constexpr int VLD = Ly::QKV_PITCH / 2;                            // row pitch of the LDS tile in elements
        const int ntile = (n + 31) >> 5;
        {
            const int qt = wave % RT;
            const int q = 16 * qt + li;
            Frag<bf16> fq[2];
            fq[0].v = *reinterpret_cast<const bf16x8*>(Qb + q * Ly::QKV_PITCH + (8 * g) * 2);
            fq[1].v = *reinterpret_cast<const bf16x8*>(Qb + q * Ly::QKV_PITCH + (32 + 8 * g) * 2);
            f32x4 oacc[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) oacc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
            float m = -INFINITY, lsum = 0.f;
            for (int kt = 0; kt < ntile; ++kt) {
                f32x4 s[2];
                const bool hi_ok = 32 * kt + 16 < 16 * RT;            // the second 16 keys of the tile exist in LDS (rows < 16 RT)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (t == 0 || hi_ok) {
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            Frag<bf16> fk;
                            fk.v = *reinterpret_cast<const bf16x8*>(Kb + (32 * kt + 16 * t + li) * Ly::QKV_PITCH + (ks * 32 + 8 * g) * 2);
                            s[t] = mma16(fk, fq[ks], s[t]);
                        }
                    }
                }
                float mx = -INFINITY;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kt * 32 + 16 * t + 4 * g + r;
                        const float v = (key < n) ? s[t][r] * AB_SCALE : -INFINITY;
                        s[t][r] = v;
                        mx = fmaxf(mx, v);
                    }
                mx = xor16_max(mx);
                mx = xor32_max(mx);
                const float mn = fmaxf(m, mx);
                const float alpha = __expf(m - mn);
                float ps = 0.f;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = __expf(s[t][r] - mn);
                        s[t][r] = p;
                        ps += p;
                    }
                lsum = lsum * alpha + ps;
                m = mn;
#pragma unroll
                for (int d = 0; d < 4; ++d) oacc[d] *= alpha;
                const Frag<bf16> fp = acc_to_frag<bf16>(s[0], s[1]);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    // V^T fragment (k = key, KMAP_ACC): keys 32 kt + 4 g .. +3 and, if present, + 16
                    const int q4 = li >> 2, p4 = li & 3;
                    typedef __attribute__((address_space(3))) bf16x4* lds_p;
                    const bf16* vp = Vb + (32 * kt + 4 * g + q4) * VLD + 16 * d + 4 * p4;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)vp);
                    bf16x4 hi;
                    hi[0] = hi[1] = hi[2] = hi[3] = (bf16)0.f;
                    if (hi_ok) hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(vp + 16 * VLD));
                    Frag<bf16> fv;
                    fv.v[0] = lo[0]; fv.v[1] = lo[1]; fv.v[2] = lo[2]; fv.v[3] = lo[3];
                    fv.v[4] = hi[0]; fv.v[5] = hi[1]; fv.v[6] = hi[2]; fv.v[7] = hi[3];
                    oacc[d] = mma16(fv, fp, oacc[d]);
                }
            }
            lsum = xor16_sum(lsum);
            lsum = xor32_sum(lsum);
            const float inv = 1.0f / lsum;
            // o (transposed accumulator: row = query li, columns 16 d + 4 g + r) -> LDS (XN region) and global
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bf16x4 pk;
                pk[0] = (bf16)(oacc[d][0] * inv); pk[1] = (bf16)(oacc[d][1] * inv);
                pk[2] = (bf16)(oacc[d][2] * inv); pk[3] = (bf16)(oacc[d][3] * inv);
                if (q >= n) pk[0] = pk[1] = pk[2] = pk[3] = (bf16)0.f;
                *reinterpret_cast<bf16x4*>(XN + q * Ly::XN_PITCH + (hh * 64 + 16 * d + 4 * g) * 2) = pk;
                if (q < n) *reinterpret_cast<bf16x4*>(o_out + (row0 + q) * D + hh * 64 + 16 * d + 4 * g) = pk;
            }
            if (q < n && g == 0) lse_out[((long)b * H + hh) * n + q] = m + __logf(lsum);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                         // B2: o in LDS, qkv no longer needed
    stamp(3);

    // ---- out-projection: KT blocks of Wo; y (f32) -> LDS where qkv was
    float* Y = reinterpret_cast<float*>(QKV);
    constexpr int YLD = Ly::Y_PITCH / 4;
    if (rt < RT) load_a(XN, Ly::XN_PITCH);                                // o rows of this wave's tile
#pragma unroll
    for (int c = 0; c < KT; ++c) {                                        // tail operands: on their way during the out-projection
        bor[c] = *reinterpret_cast<const f32x4*>(bo + 4 * (li + 16 * c));
        g2r[c] = *reinterpret_cast<const f32x4*>(ln2_w + 4 * (li + 16 * c));
        b2r[c] = *reinterpret_cast<const f32x4*>(ln2_b + 4 * (li + 16 * c));
    }
    for (int blk = NB_QKV; blk < NB; ++blk) {
        __builtin_amdgcn_s_barrier();                                     // R_blk
        if (rt < RT) {
            const f32x4 acc = block_mma(WR + (blk % Ly::NSTAGE) * Ly::WBLK);
#pragma unroll
            for (int r = 0; r < 4; ++r) Y[(16 * rt + 4 * g + r) * YLD + 64 * (blk - NB_QKV) + 16 * ct + li] = acc[r];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                         // B3: y complete in LDS
    stamp(4);

    // ---- x1 = x + y + bo (fp32 residual stream), xn2 = LN2(x1): same row -> 16-lane mapping as LN1
    {
        const int r = 4 * wave + g;
        const bool ok = r < n;
        f32x4 v[KT];
        float s1 = 0.f;
#pragma unroll
        for (int c = 0; c < KT; ++c) {
            v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok) {
                v[c] = round_res<R>(*reinterpret_cast<const f32x4*>(Y + r * YLD + 4 * (li + 16 * c)) + bor[c] + x0[c]);     // x1 as the backward reads it
                st_res4<R>(x1_out + (row0 + r) * D + 4 * (li + 16 * c), v[c]);
            }
            s1 += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
        }
        const float mean = row16_sum(s1) / D;
        float s2 = 0.f;
#pragma unroll
        for (int c = 0; c < KT; ++c) {
            const f32x4 d = v[c] - mean;
            v[c] = d;
            s2 += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
        const float rstd = rsqrtf(row16_sum(s2) / D + eps);
        if (ok) {
#pragma unroll
            for (int c = 0; c < KT; ++c) {
                const f32x4 y = v[c] * rstd * g2r[c] + b2r[c];
                bf16x4 pk;
                pk[0] = (bf16)y[0]; pk[1] = (bf16)y[1]; pk[2] = (bf16)y[2]; pk[3] = (bf16)y[3];
                *reinterpret_cast<bf16x4*>(xn2_out + (row0 + r) * D + 4 * (li + 16 * c)) = pk;
            }
        }
    }
    stamp(5);
}

#ifndef M3L_BLOCK_BODIES_ONLY
template <int KT, typename R>
__global__ __launch_bounds__(AB_THREADS) void attn_block_fwd_kernel(
    const R* __restrict__ x, const float* __restrict__ ln1_w, const float* __restrict__ ln1_b, const bf16* __restrict__ Wqkv,
    const bf16* __restrict__ Wo, const float* __restrict__ bo, const float* __restrict__ ln2_w, const float* __restrict__ ln2_b,
    float eps, int n, bf16* __restrict__ xn1_out, bf16* __restrict__ qkv_out, bf16* __restrict__ o_out, float* __restrict__ lse_out,
    R* __restrict__ x1_out, bf16* __restrict__ xn2_out, unsigned long long* __restrict__ phase_ts) {
    attn_block_fwd_body<KT, R>(x, ln1_w, ln1_b, Wqkv, Wo, bo, ln2_w, ln2_b, eps, n, xn1_out, qkv_out, o_out, lse_out, x1_out, xn2_out, phase_ts);
}

int g_ab_state = 0;   // 0 = unknown, 1 = on, -1 = off
unsigned long long* g_attn_phase_ts = nullptr;   // profiling hook: device buffer [B][8] of phase stamps, or null
#endif

// ---------------------------------------------------------------------------------------------------------------
// Backward of the attention half of a short-sequence layer, dgrad chain only (weight gradients stay with the grouped TN GEMM on the
// side stream, which reads dqkv written here together with the saved xn1, o and the incoming dx1_t):
//     dO   = dx1_t Wo                                         (per head: one 64-row block of Wo^T)
//     dq, dk, dv = attention backward (q, k, v, o, dO, lse)   per head, 3 + 3 waves (dQ by query tile, dK/dV by key tile)
//     dxn1 = dqkv Wqkv                                         accumulated over the 3 H column blocks of Wqkv^T, fp32, stays on chip
//     dx   = dx1 + LN1-backward(dxn1; x, gamma1)               + compute-type copy + [3 D] partials (dgamma1 | dbeta1 | colsum dx)
// One head at a time: its q / k / v / o / dO tiles (48 x 64 each) are the only attention state in LDS; the weight ring carries, per
// head, the Wo^T block and the three Wqkv^T blocks of that head's q, k and v columns.
template <int KT> struct AbBwdLayout {
    static constexpr int D = 64 * KT;
    static constexpr int A_PITCH = D * 2 + 16;
    static constexpr int T_PITCH = 64 * 2 + 16;           // 144 B: k-contiguous b128 reads and transpose reads are both conflict-free
    static constexpr int Y_PITCH = D * 4 + 16;
    static constexpr int A_BYTES = 48 * A_PITCH;          // dx1_t
    static constexpr int T_BYTES = 48 * T_PITCH;          // one 48 x 64 tile
    static constexpr int NT = 8;                          // q, k, v, o, dO, dq, dk, dv
    static constexpr int WBLK = KT * 64 * 128;
    static constexpr int NSTAGE = 3;
    static constexpr int STATS = 2 * 48 * 4;              // lse, Dsum
    static constexpr int TOTAL = A_BYTES + NT * T_BYTES + STATS + NSTAGE * WBLK;
    static_assert(48 * Y_PITCH <= NSTAGE * WBLK, "dxn1 staging reuses the ring");
    static_assert(AB_CW * 3 * D * 4 <= A_BYTES + NT * T_BYTES, "LayerNorm partials reuse the tile region");
};

// R = storage type of the residual stream (x, dres, dx_out).  bf16: dres is not read — the incoming residual gradient IS dx1t — and dx_out is
// not written (only the compute-type result dxt_out, which must then be given)
template <int KT, typename R>
__device__ M3L_BODY_INLINE void attn_block_bwd_body(
    const bf16* __restrict__ dx1t, const R* __restrict__ dres, const R* __restrict__ x, const float* __restrict__ ln1_w,
    const bf16* __restrict__ qkv, const bf16* __restrict__ o, const float* __restrict__ lse, const bf16* __restrict__ WoT,
    const bf16* __restrict__ WqkvT, float eps, int n, bf16* __restrict__ dqkv_out, R* __restrict__ dx_out, bf16* __restrict__ dxt_out,
    float* __restrict__ ln_part) {
    using Ly = AbBwdLayout<KT>;
    constexpr int D = Ly::D, H = KT, KSTEPS = 2 * KT, NB = 4 * KT, NDMA = 8 * KT, TP = Ly::T_PITCH, TLD = TP / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* A0 = smem;
    char* TQ = A0 + Ly::A_BYTES;
    char* TK = TQ + Ly::T_BYTES;
    char* TV = TK + Ly::T_BYTES;
    char* TO = TV + Ly::T_BYTES;
    char* TDO = TO + Ly::T_BYTES;
    char* TDQ = TDO + Ly::T_BYTES;
    char* TDK = TDQ + Ly::T_BYTES;
    char* TDV = TDK + Ly::T_BYTES;
    float* LS = reinterpret_cast<float*>(TDV + Ly::T_BYTES);
    float* DSUM = LS + 48;
    char* WR = reinterpret_cast<char*>(DSUM + 48);
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef __attribute__((address_space(1))) const void* gl_vp;
    typedef __attribute__((address_space(3))) bf16x4* lds_p4;

    const int tid = M3L_BODY_TID(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int b = blockIdx.x;
    const long row0 = (long)b * n;
    const int RT = (n + 15) >> 4;

    if (wave == AB_CW) {
        // ------------------------------------------------------------------ DMA wave: per head Wo^T rows 64h.., then the Wqkv^T column
        // blocks of q_h, k_h, v_h (columns 64h, D + 64h, 2D + 64h of the [D][3D] matrix)
        const int srow = lane >> 3, spc = lane & 7;
        auto issue = [&](int blk) {
            char* dst = WR + (blk % Ly::NSTAGE) * Ly::WBLK;
            const int h = blk >> 2, kind = blk & 3;
            if (kind == 0) {
#pragma unroll
                for (int rg = 0; rg < 8; ++rg) {
                    const bf16* src = WoT + (long)(64 * h + 8 * rg + srow) * D + ((spc ^ srow) << 3);
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt)
                        __builtin_amdgcn_global_load_lds((gl_vp)(src + kt * 64), (lds_vp)(dst + kt * 8192 + rg * 1024), 16, 0, 0);
                }
            } else {
                const int col0 = (kind - 1) * D + 64 * h;
#pragma unroll
                for (int rg = 0; rg < 8; ++rg) {
#pragma unroll
                    for (int j = 0; j < KT; ++j) {
                        const bf16* src = WqkvT + (long)(64 * j + 8 * rg + srow) * 3 * D + col0 + ((spc ^ srow) << 3);
                        __builtin_amdgcn_global_load_lds((gl_vp)src, (lds_vp)(dst + j * 8192 + rg * 1024), 16, 0, 0);
                    }
                }
            }
        };
        issue(0);
        issue(1);
        __builtin_amdgcn_s_barrier();                                     // B0
        for (int blk = 0; blk < NB; ++blk) {
            if ((blk & 3) == 1) {                                         // X1, X2, X3 of this head (between its first and second block)
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_barrier();
            }
            if (blk + 1 < NB) {
                if (NDMA == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                                 // R_blk
            if (blk + 2 < NB) issue(blk + 2);
        }
        __builtin_amdgcn_s_barrier();                                     // BE1
        __builtin_amdgcn_s_barrier();                                     // BE2
        __builtin_amdgcn_s_barrier();                                     // BE3
        return;
    }

    // ---------------------------------------------------------------------- compute waves
    // tile loader: 4 tiles (q, k, v, o) x 48 rows x 8 pieces of 16 bytes = 1536 pieces, 2 per thread
    const int pr0 = tid >> 3, pc0 = tid & 7;                              // piece (row, 16-byte column) for tid < 384; tiles 2 * (tid / 384) + {0, 1}
    const int lrow = pr0 % 48, lt0 = (pr0 / 48) * 2;                      // lt0 in {0, 2}: this thread loads tiles lt0 and lt0 + 1
    auto tile_src = [&](int t, int h) -> const bf16* {                    // t: 0 q, 1 k, 2 v, 3 o
        return t < 3 ? qkv + (row0 + lrow) * 3 * D + t * D + 64 * h + pc0 * 8 : o + (row0 + lrow) * D + 64 * h + pc0 * 8;
    };
    uint4 nx0 = uint4{0u, 0u, 0u, 0u}, nx1 = nx0;
    if (lrow < n) {
        nx0 = *reinterpret_cast<const uint4*>(tile_src(lt0, 0));
        nx1 = *reinterpret_cast<const uint4*>(tile_src(lt0 + 1, 0));
    }
    {
        constexpr int CPR = D / 8;
        for (int id = tid; id < 48 * CPR; id += 64 * AB_CW) {
            const int r = id / CPR, c = id % CPR;
            uint4 v = uint4{0u, 0u, 0u, 0u};
            if (r < n) v = *reinterpret_cast<const uint4*>(dx1t + (row0 + r) * D + c * 8);
            *reinterpret_cast<uint4*>(A0 + r * Ly::A_PITCH + c * 16) = v;
        }
    }
    *reinterpret_cast<uint4*>(TQ + lt0 * Ly::T_BYTES + lrow * TP + pc0 * 16) = nx0;
    *reinterpret_cast<uint4*>(TQ + (lt0 + 1) * Ly::T_BYTES + lrow * TP + pc0 * 16) = nx1;
    if (tid < 48) LS[tid] = tid < n ? lse[((long)b * H + 0) * n + tid] : INFINITY;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                         // B0

    const int ct = wave & 3, rt = wave >> 2;
    f32x4 yacc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) yacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntile = (n + 31) >> 5;
    // operands of the LayerNorm tail (this thread's x / dres row chunks, gamma1): requested during the LAST head's projection blocks,
    // once the attention registers are free (the head loop is peeled for that), so that the tail starts without a memory round trip
    f32x4 xr_t[KT], dr_t[KT], gm_t[KT];

    auto head = [&](int h, auto last_c) {
        constexpr bool LAST = decltype(last_c)::value;
        float ls_next = INFINITY;
        if (h + 1 < H) {                                                  // next head's tiles and lse on their way (registers)
            if (lrow < n) {
                nx0 = *reinterpret_cast<const uint4*>(tile_src(lt0, h + 1));
                nx1 = *reinterpret_cast<const uint4*>(tile_src(lt0 + 1, h + 1));
            }
            if (tid < n) ls_next = lse[((long)b * H + h + 1) * n + tid];
        }
        // ---- dO_h = dx1_t Wo[:, head h]: block 4h
        __builtin_amdgcn_s_barrier();                                     // R(4h)
        if (rt < RT) {
            const char* Wb = WR + ((4 * h) % Ly::NSTAGE) * Ly::WBLK;
            const int wrow = 16 * ct + li;
            Frag<bf16> fw[KSTEPS], fa[KSTEPS];                            // all fragments requested together, two accumulator chains
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                fw[ks].v = *reinterpret_cast<const bf16x8*>(Wb + (ks >> 1) * 8192 + wrow * 128 + ((((ks & 1) * 4 + g) ^ (wrow & 7)) << 4));
                fa[ks].v = *reinterpret_cast<const bf16x8*>(A0 + (16 * rt + li) * Ly::A_PITCH + (ks * 32 + 8 * g) * 2);
            }
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ks += 2) {
                acc = mma16(fa[ks], fw[ks], acc);
                acc1 = mma16(fa[ks + 1], fw[ks + 1], acc1);
            }
            acc = acc + acc1;
#pragma unroll
            for (int r = 0; r < 4; ++r) *reinterpret_cast<bf16*>(TDO + (16 * rt + 4 * g + r) * TP + (16 * ct + li) * 2) = (bf16)acc[r];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                     // X1: dO_h visible
        // ---- Dsum[q] = sum_d dO[q, d] o[q, d]: 8 threads per query
        if (tid < 384) {
            const int q = tid >> 3, part = tid & 7;
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(TDO + q * TP + part * 16);
            const bf16x8 c = *reinterpret_cast<const bf16x8*>(TO + q * TP + part * 16);
            float sdo = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) sdo += (float)a[j] * (float)c[j];
            sdo += __shfl_xor(sdo, 1, 64);
            sdo += __shfl_xor(sdo, 2, 64);
            sdo += __shfl_xor(sdo, 4, 64);
            if (part == 0) DSUM[q] = q < n ? sdo : 0.f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                     // X2: Dsum visible
        if (wave < RT) {
            // ---- dQ for query tile `wave` (lane li = query): S^T = K Q^T, dP^T = V dO^T, dS^T -> operand, dQ^T += K^T dS^T
            const int q = 16 * wave + li;
            Frag<bf16> fq[2], fdo[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                fq[ks].v = *reinterpret_cast<const bf16x8*>(TQ + q * TP + (ks * 32 + 8 * g) * 2);
                fdo[ks].v = *reinterpret_cast<const bf16x8*>(TDO + q * TP + (ks * 32 + 8 * g) * 2);
            }
            const float Dq = DSUM[q], lq = LS[q];
            f32x4 dq[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int kt = 0; kt < ntile; ++kt) {
                const bool hi_ok = 32 * kt + 16 < 16 * RT;
                f32x4 ds[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (t == 0 || hi_ok) {
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            Frag<bf16> fk, fv;
                            fk.v = *reinterpret_cast<const bf16x8*>(TK + (32 * kt + 16 * t + li) * TP + (ks * 32 + 8 * g) * 2);
                            fv.v = *reinterpret_cast<const bf16x8*>(TV + (32 * kt + 16 * t + li) * TP + (ks * 32 + 8 * g) * 2);
                            sc = mma16(fk, fq[ks], sc);
                            dp = mma16(fv, fdo[ks], dp);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kt * 32 + 16 * t + 4 * g + r;
                        const float p = (key < n) ? __expf(sc[r] * AB_SCALE - lq) : 0.f;
                        ds[t][r] = p * (dp[r] - Dq) * AB_SCALE;
                    }
                }
                const Frag<bf16> fds = acc_to_frag<bf16>(ds[0], ds[1]);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int q4 = li >> 2, p4 = li & 3;
                    const bf16* kp = reinterpret_cast<const bf16*>(TK) + (32 * kt + 4 * g + q4) * TLD + 16 * d + 4 * p4;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p4)kp);
                    bf16x4 hi;
                    hi[0] = hi[1] = hi[2] = hi[3] = (bf16)0.f;
                    if (hi_ok) hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p4)(kp + 16 * TLD));
                    Frag<bf16> fkT;
                    fkT.v[0] = lo[0]; fkT.v[1] = lo[1]; fkT.v[2] = lo[2]; fkT.v[3] = lo[3];
                    fkT.v[4] = hi[0]; fkT.v[5] = hi[1]; fkT.v[6] = hi[2]; fkT.v[7] = hi[3];
                    dq[d] = mma16(fkT, fds, dq[d]);
                }
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bf16x4 pk;
                pk[0] = (bf16)dq[d][0]; pk[1] = (bf16)dq[d][1]; pk[2] = (bf16)dq[d][2]; pk[3] = (bf16)dq[d][3];
                if (q >= n) pk[0] = pk[1] = pk[2] = pk[3] = (bf16)0.f;
                *reinterpret_cast<bf16x4*>(TDQ + q * TP + (16 * d + 4 * g) * 2) = pk;
            }
        } else if (wave < 2 * RT) {
            // ---- dK / dV for key tile `wave - RT` (lane li = key): S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS
            const int key = 16 * (wave - RT) + li;
            Frag<bf16> fk[2], fv[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                fk[ks].v = *reinterpret_cast<const bf16x8*>(TK + key * TP + (ks * 32 + 8 * g) * 2);
                fv[ks].v = *reinterpret_cast<const bf16x8*>(TV + key * TP + (ks * 32 + 8 * g) * 2);
            }
            f32x4 dk[4], dv[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                dk[d] = f32x4{0.f, 0.f, 0.f, 0.f};
                dv[d] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            for (int qt = 0; qt < ntile; ++qt) {
                const bool hi_ok = 32 * qt + 16 < 16 * RT;
                f32x4 pp[2], ds[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (t == 0 || hi_ok) {
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            Frag<bf16> fa, fg;
                            fa.v = *reinterpret_cast<const bf16x8*>(TQ + (32 * qt + 16 * t + li) * TP + (ks * 32 + 8 * g) * 2);
                            fg.v = *reinterpret_cast<const bf16x8*>(TDO + (32 * qt + 16 * t + li) * TP + (ks * 32 + 8 * g) * 2);
                            sc = mma16(fa, fk[ks], sc);
                            dp = mma16(fg, fv[ks], dp);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ql = 32 * qt + 16 * t + 4 * g + r;
                        const bool live = key < n && ql < n;
                        const float pv = live ? __expf(sc[r] * AB_SCALE - LS[live ? ql : 0]) : 0.f;
                        pp[t][r] = pv;
                        ds[t][r] = live ? pv * (dp[r] - DSUM[ql]) * AB_SCALE : 0.f;
                    }
                }
                const Frag<bf16> fp = acc_to_frag<bf16>(pp[0], pp[1]);
                const Frag<bf16> fds = acc_to_frag<bf16>(ds[0], ds[1]);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int q4 = li >> 2, p4 = li & 3;
                    const int roff = (32 * qt + 4 * g + q4) * TLD + 16 * d + 4 * p4;
                    const bf16* gp = reinterpret_cast<const bf16*>(TDO) + roff;
                    const bf16* qp = reinterpret_cast<const bf16*>(TQ) + roff;
                    const bf16x4 glo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p4)gp);
                    const bf16x4 qlo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p4)qp);
                    bf16x4 ghi, qhi;
                    ghi[0] = ghi[1] = ghi[2] = ghi[3] = (bf16)0.f;
                    qhi = ghi;
                    if (hi_ok) {
                        ghi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p4)(gp + 16 * TLD));
                        qhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p4)(qp + 16 * TLD));
                    }
                    Frag<bf16> fgT, fqT;
                    fgT.v[0] = glo[0]; fgT.v[1] = glo[1]; fgT.v[2] = glo[2]; fgT.v[3] = glo[3];
                    fgT.v[4] = ghi[0]; fgT.v[5] = ghi[1]; fgT.v[6] = ghi[2]; fgT.v[7] = ghi[3];
                    fqT.v[0] = qlo[0]; fqT.v[1] = qlo[1]; fqT.v[2] = qlo[2]; fqT.v[3] = qlo[3];
                    fqT.v[4] = qhi[0]; fqT.v[5] = qhi[1]; fqT.v[6] = qhi[2]; fqT.v[7] = qhi[3];
                    dv[d] = mma16(fgT, fp, dv[d]);
                    dk[d] = mma16(fqT, fds, dk[d]);
                }
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bf16x4 kk, vv;
                kk[0] = (bf16)dk[d][0]; kk[1] = (bf16)dk[d][1]; kk[2] = (bf16)dk[d][2]; kk[3] = (bf16)dk[d][3];
                vv[0] = (bf16)dv[d][0]; vv[1] = (bf16)dv[d][1]; vv[2] = (bf16)dv[d][2]; vv[3] = (bf16)dv[d][3];
                if (key >= n) { kk[0] = kk[1] = kk[2] = kk[3] = (bf16)0.f; vv = kk; }
                *reinterpret_cast<bf16x4*>(TDK + key * TP + (16 * d + 4 * g) * 2) = kk;
                *reinterpret_cast<bf16x4*>(TDV + key * TP + (16 * d + 4 * g) * 2) = vv;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                     // X3: dq / dk / dv of this head visible; q, k, v, o, lse free
        // next head's tiles into place; this head's dq | dk | dv out to global (128-byte row segments)
        if (h + 1 < H) {
            *reinterpret_cast<uint4*>(TQ + lt0 * Ly::T_BYTES + lrow * TP + pc0 * 16) = nx0;
            *reinterpret_cast<uint4*>(TQ + (lt0 + 1) * Ly::T_BYTES + lrow * TP + pc0 * 16) = nx1;
            if (tid < 48) LS[tid] = ls_next;
        }
        for (int id = tid; id < n * 24; id += 64 * AB_CW) {
            const int t = id / (n * 8), rc = id % (n * 8), r = rc >> 3, cc = rc & 7;
            *reinterpret_cast<uint4*>(dqkv_out + (row0 + r) * 3 * D + t * D + 64 * h + cc * 8) =
                *reinterpret_cast<const uint4*>(TDQ + t * Ly::T_BYTES + r * TP + cc * 16);
        }
        if (LAST) {
            const int r = 4 * wave + g;
#pragma unroll
            for (int c = 0; c < KT; ++c) {
                const int col = 4 * (li + 16 * c);
                gm_t[c] = *reinterpret_cast<const f32x4*>(ln1_w + col);
                xr_t[c] = r < n ? ld_res4<R>(x + (row0 + r) * D + col) : f32x4{0.f, 0.f, 0.f, 0.f};
                if (sizeof(R) == 2) dr_t[c] = r < n ? ld_res4<bf16>(dx1t + (row0 + r) * D + col) : f32x4{0.f, 0.f, 0.f, 0.f};
                else dr_t[c] = r < n ? ld_res4<R>(dres + (row0 + r) * D + col) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        // ---- dxn1 += dq_h Wqkv[q_h cols] + dk_h Wqkv[k_h cols] + dv_h Wqkv[v_h cols]: blocks 4h + 1 .. 4h + 3
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            __builtin_amdgcn_s_barrier();                                 // R(4h + 1 + t)
            if (rt < RT) {
                const char* Wb = WR + ((4 * h + 1 + t) % Ly::NSTAGE) * Ly::WBLK;
                const char* Ts = TDQ + t * Ly::T_BYTES;
                Frag<bf16> fa[2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) fa[ks].v = *reinterpret_cast<const bf16x8*>(Ts + (16 * rt + li) * TP + (ks * 32 + 8 * g) * 2);
                Frag<bf16> fw2[KT][2];
#pragma unroll
                for (int j = 0; j < KT; ++j) {
                    const int rw = 16 * (ct + 4 * j) + li;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        fw2[j][ks].v = *reinterpret_cast<const bf16x8*>(Wb + (rw >> 6) * 8192 + (rw & 63) * 128 + (((ks * 4 + g) ^ (rw & 7)) << 4));
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int j = 0; j < KT; ++j) yacc[j] = mma16(fa[ks], fw2[j][ks], yacc[j]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    for (int h = 0; h + 1 < H; ++h) head(h, std::false_type{});
    head(H - 1, std::true_type{});
    __builtin_amdgcn_s_barrier();                                         // BE1: ring free
    float* Y = reinterpret_cast<float*>(WR);
    constexpr int YLD = Ly::Y_PITCH / 4;
    if (rt < RT) {
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Y[(16 * rt + 4 * g + r) * YLD + 16 * (ct + 4 * j) + li] = yacc[j][r];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                         // BE2: dxn1 complete
    // ---- LN1 backward, row = 4 wave + g on 16 lanes
    float* LP = reinterpret_cast<float*>(smem);
    {
        const int r = 4 * wave + g;
        const bool ok = r < n;
        f32x4 xh[KT], dy[KT], gd[KT];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < KT; ++c) {
            xh[c] = xr_t[c];
            s += (xh[c][0] + xh[c][1]) + (xh[c][2] + xh[c][3]);
        }
        const float mean = row16_sum(s) / D;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < KT; ++c) {
            xh[c] = xh[c] - mean;
            q += (xh[c][0] * xh[c][0] + xh[c][1] * xh[c][1]) + (xh[c][2] * xh[c][2] + xh[c][3] * xh[c][3]);
        }
        const float rstd = rsqrtf(row16_sum(q) / D + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < KT; ++c) {
            xh[c] = xh[c] * rstd;
            dy[c] = ok ? *reinterpret_cast<const f32x4*>(Y + r * YLD + 4 * (li + 16 * c)) : f32x4{0.f, 0.f, 0.f, 0.f};
            gd[c] = dy[c] * gm_t[c];
            s1 += (gd[c][0] + gd[c][1]) + (gd[c][2] + gd[c][3]);
            const f32x4 t = gd[c] * xh[c];
            s2 += (t[0] + t[1]) + (t[2] + t[3]);
        }
        s1 = row16_sum(s1) / D;
        s2 = row16_sum(s2) / D;
#pragma unroll
        for (int c = 0; c < KT; ++c) {
            const int col = 4 * (li + 16 * c);
            f32x4 rr = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok) {
                rr = (gd[c] - s1 - xh[c] * s2) * rstd + dr_t[c];
                if (sizeof(R) == 4) st_res4<R>(dx_out + (row0 + r) * D + col, rr);
                if (dxt_out) {
                    bf16x4 pk;
                    pk[0] = (bf16)rr[0]; pk[1] = (bf16)rr[1]; pk[2] = (bf16)rr[2]; pk[3] = (bf16)rr[3];
                    *reinterpret_cast<bf16x4*>(dxt_out + (row0 + r) * D + col) = pk;
                }
            }
            f32x4 pg = dy[c] * xh[c], pb = dy[c], pc = rr;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pg[e] = xor16_sum(pg[e]); pg[e] = xor32_sum(pg[e]);
                pb[e] = xor16_sum(pb[e]); pb[e] = xor32_sum(pb[e]);
                pc[e] = xor16_sum(pc[e]); pc[e] = xor32_sum(pc[e]);
            }
            if (g == 0) {
                *reinterpret_cast<f32x4*>(LP + (wave * 3 + 0) * D + col) = pg;
                *reinterpret_cast<f32x4*>(LP + (wave * 3 + 1) * D + col) = pb;
                *reinterpret_cast<f32x4*>(LP + (wave * 3 + 2) * D + col) = pc;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                         // BE3
    for (int id = tid; id < 3 * D; id += 64 * AB_CW) {
        float sacc = 0.f;
#pragma unroll
        for (int w = 0; w < AB_CW; ++w) sacc += LP[w * 3 * D + id];
        ln_part[(long)b * 3 * D + id] = sacc;
    }
}

#ifndef M3L_BLOCK_BODIES_ONLY
template <int KT, typename R>
__global__ __launch_bounds__(AB_THREADS) void attn_block_bwd_kernel(
    const bf16* __restrict__ dx1t, const R* __restrict__ dres, const R* __restrict__ x, const float* __restrict__ ln1_w,
    const bf16* __restrict__ qkv, const bf16* __restrict__ o, const float* __restrict__ lse, const bf16* __restrict__ WoT,
    const bf16* __restrict__ WqkvT, float eps, int n, bf16* __restrict__ dqkv_out, R* __restrict__ dx_out, bf16* __restrict__ dxt_out,
    float* __restrict__ ln_part) {
    attn_block_bwd_body<KT, R>(dx1t, dres, x, ln1_w, qkv, o, lse, WoT, WqkvT, eps, n, dqkv_out, dx_out, dxt_out, ln_part);
}
#endif

}  // namespace

#ifndef M3L_BLOCK_BODIES_ONLY

// g_ab_state: 0 = unread, -1 = off, otherwise a bit mask: 1 = forward blocks + MLP backward block, 2 = attention backward block too
static int ab_state() {
    if (!g_ab_state) {
        const char* e = getenv("M3L_ATTN_BLOCK");
        g_ab_state = e ? (atoi(e) > 0 ? atoi(e) : -1) : 3;
    }
    return g_ab_state;
}
// Sets the block-kernel mode (0 = off) and returns the mode that was in effect (environment default resolved), so that
// `m3l_set_attn_block(m3l_set_attn_block(x))` restores the previous state exactly.
extern "C" int m3l_set_attn_block(int mode) {
    const int old = ab_state() < 0 ? 0 : g_ab_state;
    g_ab_state = mode > 0 ? mode : -1;
    return old;
}

// Profiling hook (tools/attn_phase_probe.py): a device buffer of at least 8 * B uint64 that every later attention-block forward launch
// fills with the shader-clock stamps of its phase boundaries; NULL switches it off (the default: the kernel then writes nothing).
extern "C" void m3l_set_attn_phase_buffer(void* dev_buf) { g_attn_phase_ts = reinterpret_cast<unsigned long long*>(dev_buf); }
unsigned long long* m3l_attn_phase_buffer(void) { return g_attn_phase_ts; }

// 1 when the fused attention-block kernel takes this problem
int m3l_attn_block_supported(int dtype, int D, int heads, int n, int project_out) {
    if (ab_state() < 1) return 0;
    return dtype == 1 && project_out && (D == 128 || D == 192) && heads * 64 == D && n >= 1 && n <= 48;
}

// The attention BACKWARD block (mode bit 2).  Round 1 kept it opt-in: the weight-gradient side stream (2.5 ms per step) then bounded
// the encoder backward and fusing more of the compute stream only exposed it (47.8k -> 42.8k samples/s).  With the round-2 weight
// gradient kernel (wgrad.hip, 1.1 ms per step) it pays: 49.4k -> 50.7k, so mode 3 is the default.
int m3l_attn_block_bwd_enabled(void) { return ab_state() > 0 && (g_ab_state & 2); }

int m3l_attn_block_fwd(int D, int B, int n, const float* x, const float* ln1_w, const float* ln1_b, const void* wqkv, const void* wo,
                       const float* bo, const float* ln2_w, const float* ln2_b, float eps, void* xn1, void* qkv, void* o, float* lse,
                       float* x1, void* xn2, hipStream_t st) {
    static int inited = 0;
    if (!inited) {
        M3L_HIP(hipFuncSetAttribute((const void*)attn_block_fwd_kernel<2, float>, hipFuncAttributeMaxDynamicSharedMemorySize, AbLayout<2>::TOTAL));
        M3L_HIP(hipFuncSetAttribute((const void*)attn_block_fwd_kernel<3, float>, hipFuncAttributeMaxDynamicSharedMemorySize, AbLayout<3>::TOTAL));
        M3L_HIP(hipFuncSetAttribute((const void*)attn_block_fwd_kernel<2, bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, AbLayout<2>::TOTAL));
        M3L_HIP(hipFuncSetAttribute((const void*)attn_block_fwd_kernel<3, bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, AbLayout<3>::TOTAL));
        inited = 1;
    }
    ProfScope prof("attn_block_fwd", B, n, D, 2.0 * B * n * (4.0 * D * D) + 4.0 * B * (D / 64) * (double)n * n * 64, st,
                   (double)B * n * D * (12.0 + 2.0 * (m3l_call_rb() ? 2.0 : 4.0)));       // xn1, qkv, o, xn2 | x, x1
    // (m3l_call_rb(): the residual stream x / x1 of this launch is bf16)
#define AB_LAUNCH(KT, R)                                                                                                              \
    attn_block_fwd_kernel<KT, R><<<B, AB_THREADS, AbLayout<KT>::TOTAL, st>>>((const R*)x, ln1_w, ln1_b, (const bf16*)wqkv, (const bf16*)wo, bo, ln2_w, \
                                                                            ln2_b, eps, n, (bf16*)xn1, (bf16*)qkv, (bf16*)o, lse, (R*)x1, (bf16*)xn2, \
                                                                            g_attn_phase_ts)
    M3L_CHECK(D == 128 || D == 192, "attn_block: D=%d unsupported", D);
    if (m3l_call_rb()) { if (D == 128) AB_LAUNCH(2, bf16); else AB_LAUNCH(3, bf16); }
    else if (D == 128) AB_LAUNCH(2, float);
    else AB_LAUNCH(3, float);
#undef AB_LAUNCH
    M3L_LAUNCH_CHECK();
    return 0;
}

int m3l_attn_block_bwd(int D, int B, int n, const void* dx1t, const float* dres, const float* x, const float* ln1_w, const void* qkv,
                       const void* o, const float* lse, const void* woT, const void* wqkvT, float eps, void* dqkv, float* dx_out, void* dxt_out,
                       float* ln_part, hipStream_t st) {
    static int inited = 0;
    if (!inited) {
        M3L_HIP(hipFuncSetAttribute((const void*)attn_block_bwd_kernel<2, float>, hipFuncAttributeMaxDynamicSharedMemorySize, AbBwdLayout<2>::TOTAL));
        M3L_HIP(hipFuncSetAttribute((const void*)attn_block_bwd_kernel<3, float>, hipFuncAttributeMaxDynamicSharedMemorySize, AbBwdLayout<3>::TOTAL));
        M3L_HIP(hipFuncSetAttribute((const void*)attn_block_bwd_kernel<2, bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, AbBwdLayout<2>::TOTAL));
        M3L_HIP(hipFuncSetAttribute((const void*)attn_block_bwd_kernel<3, bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, AbBwdLayout<3>::TOTAL));
        inited = 1;
    }
    M3L_CHECK(D == 128 || D == 192, "attn_block_bwd: D=%d unsupported", D);
    M3L_CHECK(!m3l_call_rb() || dxt_out, "attn_block_bwd: the bf16 residual mode writes its result to dxt_out only");
    ProfScope prof("attn_block_bwd", B, n, D, 2.0 * B * n * (4.0 * D * D) + 10.0 * B * (D / 64) * (double)n * n * 64, st,
                   (double)B * n * D * (18.0 + (m3l_call_rb() ? 2.0 : 12.0)));            // dx1t, qkv, o, dqkv, dx_t | x (+ fp32: dres, dx)
#define ABB_LAUNCH(KT, R)                                                                                                               \
    attn_block_bwd_kernel<KT, R><<<B, AB_THREADS, AbBwdLayout<KT>::TOTAL, st>>>((const bf16*)dx1t, (const R*)dres, (const R*)x, ln1_w, (const bf16*)qkv, \
                                                                               (const bf16*)o, lse, (const bf16*)woT, (const bf16*)wqkvT, eps, n,     \
                                                                               (bf16*)dqkv, (R*)dx_out, (bf16*)dxt_out, ln_part)
    if (m3l_call_rb()) { if (D == 128) ABB_LAUNCH(2, bf16); else ABB_LAUNCH(3, bf16); }
    else if (D == 128) ABB_LAUNCH(2, float);
    else ABB_LAUNCH(3, float);
#undef ABB_LAUNCH
    M3L_LAUNCH_CHECK();
    return 0;
}
#endif  // M3L_BLOCK_BODIES_ONLY
