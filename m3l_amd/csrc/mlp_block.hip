// One launch for the feed-forward half of a pre-norm transformer layer on a SHORT sequence (the MAE encoder, n <= 48 tokens):
//
//     u = xn2 W1^T + b1;   h = GELU(u);   xout = x1 + h W2^T + b2
//
// (vit_pytorch FeedForward.forward + the residual, models/pretrain_models.py:266; xn2 = LN2(x1) comes from the attention block).
// Companion of attn_block.hip, same structure: one workgroup per sample, 12 compute waves + one DMA-only wave that streams the
// weights through a 4-stage LDS ring — here alternating a 64-row block of W1 (64 hidden units x D) and the matching 64-column
// block of W2 (D outputs x 64 hidden units), so the hidden activation is produced and consumed 64 units at a time and never
// exists as a whole in LDS.  u and h (what the backward needs) leave through a small LDS staging tile as 128-byte row segments.
// Supported: bf16, D = 128 / 192, mlp_dim % 64 == 0, n <= 48.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "common.cuh"
#include "kernels.h"

namespace {

constexpr int MB_CW = 12;
constexpr int MB_THREADS = 64 * (MB_CW + 1);

template <int KT> struct MbLayout {
    static constexpr int D = 64 * KT;
    static constexpr int XN_PITCH = D * 2 + 16;          // (pitch / 16) odd: conflict-free ds_read_b128 over 16 rows
    static constexpr int HC_PITCH = 64 * 2 + 16;          // one 64-wide chunk of u or h
    static constexpr int Y_PITCH = D * 4 + 16;
    static constexpr int ROWS = 48;
    static constexpr int XN_BYTES = ROWS * XN_PITCH;
    static constexpr int HC_BYTES = ROWS * HC_PITCH;      // per chunk buffer; 2 (double buffer) x 2 (u, h)
    static constexpr int WBLK = KT * 64 * 128;
    static constexpr int NSTAGE = 4;                      // three blocks in flight while one is multiplied (the stream is L2-latency bound)
    static constexpr int TOTAL = XN_BYTES + 4 * HC_BYTES + NSTAGE * WBLK;
    static_assert(ROWS * Y_PITCH <= NSTAGE * WBLK, "the fp32 output staging tile reuses the weight ring");
    static_assert(KT == 2 || KT == 3, "vmcnt immediates in the DMA wave are 16 / 24");
};

template <int KT>
__global__ __launch_bounds__(MB_THREADS) void mlp_block_fwd_kernel(const bf16* __restrict__ xn2, const float* __restrict__ x1,
                                                                     const bf16* __restrict__ W1, const float* __restrict__ b1,
                                                                     const bf16* __restrict__ W2, const float* __restrict__ b2, int n,
                                                                     int mlp, bf16* __restrict__ u_out, bf16* __restrict__ h_out,
                                                                     float* __restrict__ xout) {
    using Ly = MbLayout<KT>;
    constexpr int D = Ly::D, KSTEPS = 2 * KT;
    constexpr int NDMA = 8 * KT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* XN = smem;                                      // [48][XN_PITCH] bf16 xn2
    char* HC = smem + Ly::XN_BYTES;                       // [2][u | h][48][HC_PITCH]
    char* WR = HC + 4 * Ly::HC_BYTES;                     // NSTAGE x WBLK, later y f32 [48][Y_PITCH]
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef __attribute__((address_space(1))) const void* gl_vp;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const long row0 = (long)blockIdx.x * n;
    const int RT = (n + 15) >> 4;
    const int NC = mlp >> 6;                              // 64-wide hidden chunks
    const int NB = 2 * NC;                                // weight blocks: W1 chunk c = block 2c, W2 chunk c = block 2c + 1

    if (wave == MB_CW) {
        // ------------------------------------------------------------------ DMA wave
        const int srow = lane >> 3, spc = lane & 7;
        auto issue = [&](int blk) {
            char* dst = WR + (blk % Ly::NSTAGE) * Ly::WBLK;
            const int c = blk >> 1;
            if ((blk & 1) == 0) {                         // W1 rows 64c .. 64c+63, all D columns: KT sub-tiles along k
#pragma unroll
                for (int rg = 0; rg < 8; ++rg) {
                    const bf16* src = W1 + (long)(64 * c + 8 * rg + srow) * D + ((spc ^ srow) << 3);
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt)
                        __builtin_amdgcn_global_load_lds((gl_vp)(src + kt * 64), (lds_vp)(dst + kt * 8192 + rg * 1024), 16, 0, 0);
                }
            } else {                                      // W2 columns 64c .. 64c+63 of all D rows: KT sub-tiles along the output rows
#pragma unroll
                for (int rg = 0; rg < 8; ++rg) {
#pragma unroll
                    for (int j = 0; j < KT; ++j) {
                        const bf16* src = W2 + (long)(64 * j + 8 * rg + srow) * mlp + 64 * c + ((spc ^ srow) << 3);
                        __builtin_amdgcn_global_load_lds((gl_vp)src, (lds_vp)(dst + j * 8192 + rg * 1024), 16, 0, 0);
                    }
                }
            }
        };
        issue(0);
        issue(1);
        if (NB > 2) issue(2);
        __builtin_amdgcn_s_barrier();                                     // B0 (xn2 in LDS)
        for (int blk = 0; blk < NB; ++blk) {
            // loads retire in order: blocks blk+1 and blk+2 (NDMA instructions each) may remain in flight
            if (blk + 2 < NB) {
                if (NDMA == 16) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
            } else if (blk + 1 < NB) {
                if (NDMA == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                                 // block landed; every compute wave is done with block blk-1
            if (blk + 3 < NB) issue(blk + 3);                             // into the stage block blk-1 occupied
        }
        __builtin_amdgcn_s_barrier();                                     // BE1: every wave is done with the last W2 block (ring -> y)
        __builtin_amdgcn_s_barrier();                                     // BE2: y complete
        return;
    }

    // ---------------------------------------------------------------------- compute waves
    // xn2 -> LDS: 16-byte chunks, all compute threads
    {
        constexpr int CPR = D / 8;
        for (int id = tid; id < 48 * CPR; id += 64 * MB_CW) {
            const int r = id / CPR, c = id % CPR;
            uint4 v = uint4{0u, 0u, 0u, 0u};
            if (r < n) v = *reinterpret_cast<const uint4*>(xn2 + (row0 + r) * D + c * 8);
            *reinterpret_cast<uint4*>(XN + r * Ly::XN_PITCH + c * 16) = v;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                         // B0

    const int ct = wave & 3, rt = wave >> 2;
    f32x4 yacc[3];                                                        // output tiles (rt, ct + 4 j), j = 0..2 (D / 64 = KT of them used)
#pragma unroll
    for (int j = 0; j < 3; ++j) yacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float bias_next = b1[16 * ct + li];                                  // b1 of chunk c is fetched during chunk c - 1 (a global load per
    for (int c = 0; c < NC; ++c) {                                        // chunk with its latency exposed cost 2 000 cycles per chunk)
        const float bias = bias_next;
        if (c + 1 < NC) bias_next = b1[64 * (c + 1) + 16 * ct + li];
        char* US = HC + (c & 1) * 2 * Ly::HC_BYTES;
        char* HS = US + Ly::HC_BYTES;
        __builtin_amdgcn_s_barrier();                                     // W1 chunk c landed (block 2c)
        if (rt < RT) {
            const char* Wb = WR + ((2 * c) % Ly::NSTAGE) * Ly::WBLK;
            const int wrow = 16 * ct + li;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                Frag<bf16> fw, fa;
                fw.v = *reinterpret_cast<const bf16x8*>(Wb + (ks >> 1) * 8192 + wrow * 128 + ((((ks & 1) * 4 + g) ^ (wrow & 7)) << 4));
                fa.v = *reinterpret_cast<const bf16x8*>(XN + (16 * rt + li) * Ly::XN_PITCH + (ks * 32 + 8 * g) * 2);
                acc = mma16(fa, fw, acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bf16 ub = (bf16)(acc[r] + bias);                    // pre-activation as the backward will read it
                const bf16 hb = (bf16)gelu_f((float)ub);
                const int off = (16 * rt + 4 * g + r) * Ly::HC_PITCH + (16 * ct + li) * 2;
                *reinterpret_cast<bf16*>(US + off) = ub;
                *reinterpret_cast<bf16*>(HS + off) = hb;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                     // u / h chunk visible; W2 chunk c landed (block 2c + 1)
        // u, h chunk -> global: 128-byte row segments
        for (int id = tid; id < n * 16; id += 64 * MB_CW) {
            const int which = id & 1, rc = id >> 1, r = rc >> 3, cc = rc & 7;
            const uint4 v = *reinterpret_cast<const uint4*>((which ? HS : US) + r * Ly::HC_PITCH + cc * 16);
            *reinterpret_cast<uint4*>((which ? h_out : u_out) + (row0 + r) * mlp + 64 * c + cc * 8) = v;
        }
        if (rt < RT) {
            const char* Wb = WR + ((2 * c + 1) % Ly::NSTAGE) * Ly::WBLK;
            Frag<bf16> fa[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[ks].v = *reinterpret_cast<const bf16x8*>(HS + (16 * rt + li) * Ly::HC_PITCH + (ks * 32 + 8 * g) * 2);
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                const int rw = 16 * (ct + 4 * j) + li;                    // output column (= W2 row) of this lane's B fragment
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    Frag<bf16> fw;
                    fw.v = *reinterpret_cast<const bf16x8*>(Wb + (rw >> 6) * 8192 + (rw & 63) * 128 + (((ks * 4 + g) ^ (rw & 7)) << 4));
                    yacc[j] = mma16(fa[ks], fw, yacc[j]);
                }
            }
        }
    }
    __builtin_amdgcn_s_barrier();                                         // BE1: the ring is free
    float* Y = reinterpret_cast<float*>(WR);
    constexpr int YLD = Ly::Y_PITCH / 4;
    if (rt < RT) {
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Y[(16 * rt + 4 * g + r) * YLD + 16 * (ct + 4 * j) + li] = yacc[j][r];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                         // BE2: y complete
    // xout = x1 + y + b2: row = 4 wave + g, the row's 16 lanes hold KT float4 chunks
    {
        const int r = 4 * wave + g;
        if (r < n) {
#pragma unroll
            for (int c = 0; c < KT; ++c) {
                const int col = 4 * (li + 16 * c);
                const f32x4 v = *reinterpret_cast<const f32x4*>(Y + r * YLD + col) + *reinterpret_cast<const f32x4*>(b2 + col) +
                                *reinterpret_cast<const f32x4*>(x1 + (row0 + r) * D + col);
                *reinterpret_cast<f32x4*>(xout + (row0 + r) * D + col) = v;
            }
        }
    }
}

}  // namespace

int m3l_mlp_block_supported(int dtype, int D, int mlp, int n) {
    return m3l_attn_block_supported(dtype, D, D / 64, n, 1) && mlp % 64 == 0 && mlp >= 64;
}

int m3l_mlp_block_fwd(int D, int mlp, int B, int n, const void* xn2, const float* x1, const void* w1, const float* b1, const void* w2,
                      const float* b2, void* u, void* h, float* xout, hipStream_t st) {
    static int inited = 0;
    if (!inited) {
        M3L_HIP(hipFuncSetAttribute((const void*)mlp_block_fwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, MbLayout<2>::TOTAL));
        M3L_HIP(hipFuncSetAttribute((const void*)mlp_block_fwd_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, MbLayout<3>::TOTAL));
        inited = 1;
    }
    M3L_CHECK(D == 128 || D == 192, "mlp_block: D=%d unsupported", D);
    ProfScope prof("mlp_block_fwd", B, n, mlp, 4.0 * B * n * (double)D * mlp, st);
#define MB_LAUNCH(KT)                                                                                                                  \
    mlp_block_fwd_kernel<KT><<<B, MB_THREADS, MbLayout<KT>::TOTAL, st>>>((const bf16*)xn2, x1, (const bf16*)w1, b1, (const bf16*)w2, b2, n, \
                                                                        mlp, (bf16*)u, (bf16*)h, xout)
    if (D == 128) MB_LAUNCH(2);
    else MB_LAUNCH(3);
#undef MB_LAUNCH
    M3L_LAUNCH_CHECK();
    return 0;
}
