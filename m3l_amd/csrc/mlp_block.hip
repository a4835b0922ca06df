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

#ifndef M3L_BODY_INLINE
#define M3L_BODY_INLINE __forceinline__
#endif
#ifndef M3L_BODY_TID
#define M3L_BODY_TID() ((int)threadIdx.x)     // enc_mega.hip launders it per body call (keeps loop-invariant code motion out of the layer loop)
#endif

namespace {

#ifndef MB_ABL
#define MB_ABL 0            // diagnostic builds only (tools/t192_ablate.sh): bits switch pieces of the forward kernel off
#endif
#ifndef MB_STAMP
#define MB_STAMP 0          // diagnostic builds only: per-wave shader-clock stamps inside the forward's chunk loop (tools/mlp_phase_probe.py)
#endif
#if MB_STAMP
__device__ unsigned long long* d_mb_stamps;          // [blocks][12 waves][16] accumulated cycles per loop section
#define MB_T(i)                                                        \
    {                                                                  \
        const unsigned long long t_ = __builtin_readcyclecounter();    \
        tacc[i] += t_ - tlast;                                         \
        tlast = t_;                                                    \
    }
#else
#define MB_T(i)
#endif
constexpr int MB_CW = 12;
constexpr int MB_THREADS = 64 * (MB_CW + 1);

__device__ __forceinline__ float row16_sum_mb(float v) {   // sum over the 16 lanes of a DPP row (see attn_block.hip)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
    return v;
}

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

template <int KT, typename R>
__device__ M3L_BODY_INLINE void mlp_block_fwd_body(const bf16* __restrict__ xn2, const R* __restrict__ x1,
                                                                     const bf16* __restrict__ W1, const float* __restrict__ b1,
                                                                     const bf16* __restrict__ W2, const float* __restrict__ b2, int n,
                                                                     int mlp, bf16* __restrict__ u_out, bf16* __restrict__ h_out,
                                                                     R* __restrict__ xout) {
    using Ly = MbLayout<KT>;
    constexpr int D = Ly::D, KSTEPS = 2 * KT;
    constexpr int NDMA = 8 * KT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* XN = smem;                                      // [48][XN_PITCH] bf16 xn2
    char* HC = smem + Ly::XN_BYTES;                       // [2][u | h][48][HC_PITCH]
    char* WR = HC + 4 * Ly::HC_BYTES;                     // NSTAGE x WBLK, later y f32 [48][Y_PITCH]
    float* B1S = reinterpret_cast<float*>(WR + Ly::NSTAGE * Ly::WBLK);    // [mlp] fc1 bias (see the chunk loop)
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef __attribute__((address_space(1))) const void* gl_vp;

    const int tid = M3L_BODY_TID(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const long row0 = (long)blockIdx.x * n;
    const int RT = (n + 15) >> 4;
    const int NC = mlp >> 6;                              // 64-wide hidden chunks
    const int NB = 2 * NC;                                // weight blocks: W1 chunk c = block 2c, W2 chunk c = block 2c + 1

    if (wave == MB_CW) {
        // ------------------------------------------------------------------ DMA wave
        const int srow = lane >> 3, spc = lane & 7;
        auto issue = [&](int blk) {
            if (MB_ABL & 1) return;
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
        for (int id = tid; id < mlp; id += 64 * MB_CW) B1S[id] = b1[id];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                         // B0

    // the epilogue's operands (this thread's residual row chunks, b2) are requested now: loaded at the end they were one more exposed
    // memory round trip of a kernel that is a chain of dependent steps
    f32x4 x1r[KT], b2r[KT];
    {
        const int r = 4 * wave + g;
#pragma unroll
        for (int c = 0; c < KT; ++c) {
            const int col = 4 * (li + 16 * c);
            b2r[c] = *reinterpret_cast<const f32x4*>(b2 + col);
            x1r[c] = r < n ? ld_res4<R>(x1 + (row0 + r) * D + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const int ct = wave & 3, rt = wave >> 2;
    // this wave's 16 rows of the A operand (all k) are the same for every hidden chunk: read once; per chunk the weight fragments are
    // requested together and feed two accumulator chains (see attn_block.hip block_mma)
    Frag<bf16> fa_c[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) fa_c[ks].v = *reinterpret_cast<const bf16x8*>(XN + (16 * rt + li) * Ly::XN_PITCH + (ks * 32 + 8 * g) * 2);
    f32x4 yacc[3];                                                        // output tiles (rt, ct + 4 j), j = 0..2 (D / 64 = KT of them used)
#pragma unroll
    for (int j = 0; j < 3; ++j) yacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // The chunk loop issues NO vector-memory load: a wave's vmcnt counts its loads and stores together and they may complete out of
    // order, so whenever a wave that has stores in flight needs a load's result the compiler must wait for vmcnt(0) — with b1 fetched
    // from global memory once per chunk (even a chunk ahead) that was the HBM round trip of the chunk's u / h stores, 12 times per
    // kernel.  b1 comes from LDS; the stores are never waited for inside the loop.
#if MB_STAMP
    unsigned long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
    const unsigned long long tstart = tlast;
#endif
    for (int c = 0; c < NC; ++c) {
        const float bias = B1S[64 * c + 16 * ct + li];
        char* US = HC + (c & 1) * 2 * Ly::HC_BYTES;
        char* HS = US + Ly::HC_BYTES;
        MB_T(0)                                                           // loop overhead + bias read
        __builtin_amdgcn_s_barrier();                                     // W1 chunk c landed (block 2c)
        MB_T(1)                                                           // barrier 1 (waits for the W1 block and for the slowest wave)
        if (rt < RT) {
            const char* Wb = WR + ((2 * c) % Ly::NSTAGE) * Ly::WBLK;
            const int wrow = 16 * ct + li;
            Frag<bf16> fw[KSTEPS];
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks)
                fw[ks].v = *reinterpret_cast<const bf16x8*>(Wb + (ks >> 1) * 8192 + wrow * 128 + ((((ks & 1) * 4 + g) ^ (wrow & 7)) << 4));
#if MB_STAMP
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) asm volatile("" : "+v"(fw[ks].v));
#endif
            MB_T(2)                                                       // fc1 weight fragments: 6 ds_read_b128 issued and landed
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!(MB_ABL & 4)) {
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ks += 2) {
                    acc = mma16(fa_c[ks], fw[ks], acc);
                    acc1 = mma16(fa_c[ks + 1], fw[ks + 1], acc1);
                }
            }
            acc = acc + acc1;
#if MB_STAMP
            asm volatile("" : "+v"(acc));
#endif
            MB_T(3)                                                       // fc1: 6 MFMAs in two chains
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bf16 ub = (bf16)(acc[r] + bias);                    // pre-activation as the backward will read it
                const bf16 hb = (bf16)gelu_fast((float)ub);
                const int off = (16 * rt + 4 * g + r) * Ly::HC_PITCH + (16 * ct + li) * 2;
                *reinterpret_cast<bf16*>(US + off) = ub;
                *reinterpret_cast<bf16*>(HS + off) = hb;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        MB_T(4)                                                           // bias + GELU + 8 ds_write_b16, drained
        __builtin_amdgcn_s_barrier();                                     // u / h chunk visible; W2 chunk c landed (block 2c + 1)
        MB_T(5)                                                           // barrier 2
        // u, h chunk -> global: 128-byte row segments
        for (int id = tid; id < n * 16 && !(MB_ABL & 2); id += 64 * MB_CW) {
            const int which = id & 1, rc = id >> 1, r = rc >> 3, cc = rc & 7;
            if (which && !h_out) continue;                                // h not saved: the weight-gradient kernel recomputes GELU(u)
            const uint4 v = *reinterpret_cast<const uint4*>((which ? HS : US) + r * Ly::HC_PITCH + cc * 16);
            *reinterpret_cast<uint4*>((which ? h_out : u_out) + (row0 + r) * mlp + 64 * c + cc * 8) = v;
        }
        MB_T(6)                                                           // u / h chunk: LDS -> registers -> global stores issued
        if (rt < RT) {
            const char* Wb = WR + ((2 * c + 1) % Ly::NSTAGE) * Ly::WBLK;
            Frag<bf16> fa[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[ks].v = *reinterpret_cast<const bf16x8*>(HS + (16 * rt + li) * Ly::HC_PITCH + (ks * 32 + 8 * g) * 2);
            Frag<bf16> fw2[KT][2];
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                const int rw = 16 * (ct + 4 * j) + li;                    // output column (= W2 row) of this lane's B fragment
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    fw2[j][ks].v = *reinterpret_cast<const bf16x8*>(Wb + (rw >> 6) * 8192 + (rw & 63) * 128 + (((ks * 4 + g) ^ (rw & 7)) << 4));
            }
#if MB_STAMP
#pragma unroll
            for (int j = 0; j < KT; ++j) { asm volatile("" : "+v"(fw2[j][0].v)); asm volatile("" : "+v"(fw2[j][1].v)); }
#endif
            MB_T(7)                                                       // fc2 fragments: 8 ds_read_b128 issued and landed
            if (!(MB_ABL & 4)) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int j = 0; j < KT; ++j) yacc[j] = mma16(fa[ks], fw2[j][ks], yacc[j]);
            }
#if MB_STAMP
#pragma unroll
            for (int j = 0; j < KT; ++j) asm volatile("" : "+v"(yacc[j]));
#endif
            MB_T(8)                                                       // fc2: 6 MFMAs in three chains
        }
    }
#if MB_STAMP
    if (d_mb_stamps && lane == 0) {
        unsigned long long* o = d_mb_stamps + ((long)blockIdx.x * MB_CW + wave) * 16;
#pragma unroll
        for (int i = 0; i < 10; ++i) o[i] = tacc[i];
        o[10] = tlast - tstart;
    }
#endif
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
                const f32x4 v = *reinterpret_cast<const f32x4*>(Y + r * YLD + col) + b2r[c] + x1r[c];
                st_res4<R>(xout + (row0 + r) * D + col, v);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of the same half layer, dgrad chain only (the weight gradients stay with the grouped TN GEMM on the side stream, which
// reads du and dx1_t written here):
//     du   = (dx_t W2) * gelu'(u)                    [n, mlp]   (+ its column sums = fc1 bias gradient, one partial row per sample)
//     dxn2 = du W1                                    [n, D]     (kept in fp32, never leaves the CU)
//     dx1  = dx + LN2-backward(dxn2; x1, gamma2)      in place, + compute-type copy, + [3 D] partials (dgamma2 | dbeta2 | colsum dx1)
// Same ring / barrier structure as the forward with (W2^T, W1^T) in place of (W1, W2): block 2c = rows 64c.. of W2^T [mlp][D],
// block 2c + 1 = columns 64c.. of W1^T [D][mlp].  The DMA wave brings chunk c of u with block 2c.
// R = storage type of the residual stream (dx, x1).  bf16: the incoming residual gradient is read from dxt (it IS that gradient) and only
// the compute-type result dx1t_out is written
template <int KT, typename R>
__device__ M3L_BODY_INLINE void mlp_block_bwd_body(const bf16* __restrict__ dxt, R* __restrict__ dx,
                                                                     const R* __restrict__ x1, const float* __restrict__ ln2_w,
                                                                     const bf16* __restrict__ u, const bf16* __restrict__ W2T,
                                                                     const bf16* __restrict__ W1T, float eps, int n, int mlp,
                                                                     bf16* __restrict__ du_out, bf16* __restrict__ dx1t_out,
                                                                     float* __restrict__ cs_part, float* __restrict__ ln_part) {
    using Ly = MbLayout<KT>;
    constexpr int D = Ly::D, KSTEPS = 2 * KT;
    constexpr int NDMA = 8 * KT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* XN = smem;                                      // [48][XN_PITCH] bf16 dx_t; later (with HC) the LayerNorm partials
    char* HC = smem + Ly::XN_BYTES;                       // [2][u | du][48][HC_PITCH]
    char* WR = HC + 4 * Ly::HC_BYTES;                     // NSTAGE x WBLK, later dxn2 f32 [48][Y_PITCH]
    float* CS = reinterpret_cast<float*>(WR + Ly::NSTAGE * Ly::WBLK);     // [3][mlp] column sums of du per row tile
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef __attribute__((address_space(1))) const void* gl_vp;

    const int tid = M3L_BODY_TID(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int b = blockIdx.x;
    const long row0 = (long)b * n;
    const int RT = (n + 15) >> 4;
    const int NC = mlp >> 6, NB = 2 * NC;

    if (wave == MB_CW) {
        const int srow = lane >> 3, spc = lane & 7;
        auto issue = [&](int blk) {
            char* dst = WR + (blk % Ly::NSTAGE) * Ly::WBLK;
            const int c = blk >> 1;
            if ((blk & 1) == 0) {
#pragma unroll
                for (int rg = 0; rg < 8; ++rg) {
                    const bf16* src = W2T + (long)(64 * c + 8 * rg + srow) * D + ((spc ^ srow) << 3);
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt)
                        __builtin_amdgcn_global_load_lds((gl_vp)(src + kt * 64), (lds_vp)(dst + kt * 8192 + rg * 1024), 16, 0, 0);
                }
                // ... and the chunk's u [48][64] as 128-byte rows into US[c & 1] (rows past n: the last row again, never used).  The compute
                // waves store du every chunk: a load of theirs would make them wait for those stores' round trip (see the forward).
                char* us = HC + (c & 1) * 2 * Ly::HC_BYTES;
#pragma unroll
                for (int pc = 0; pc < 6; ++pc) {
                    const int r = min(8 * pc + srow, n - 1);
                    __builtin_amdgcn_global_load_lds((gl_vp)(u + (row0 + r) * mlp + 64 * c + spc * 8), (lds_vp)(us + pc * 1024), 16, 0, 0);
                }
            } else {
#pragma unroll
                for (int rg = 0; rg < 8; ++rg) {
#pragma unroll
                    for (int j = 0; j < KT; ++j) {
                        const bf16* src = W1T + (long)(64 * j + 8 * rg + srow) * mlp + 64 * c + ((spc ^ srow) << 3);
                        __builtin_amdgcn_global_load_lds((gl_vp)src, (lds_vp)(dst + j * 8192 + rg * 1024), 16, 0, 0);
                    }
                }
            }
        };
        issue(0);
        issue(1);
        if (NB > 2) issue(2);
        __builtin_amdgcn_s_barrier();                                     // B0
        for (int blk = 0; blk < NB; ++blk) {
            // blocks blk + 1 and blk + 2 may remain in flight: one of the two is even and carries its chunk's 6 u pieces
            if (blk + 2 < NB) {
                if (NDMA == 16) asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(54)" ::: "memory");
            } else if (blk + 1 < NB) {                                    // blk + 1 = NB - 1 is odd: no u pieces
                if (NDMA == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (blk + 3 < NB) issue(blk + 3);
        }
        __builtin_amdgcn_s_barrier();                                     // BE1
        __builtin_amdgcn_s_barrier();                                     // BE2
        __builtin_amdgcn_s_barrier();                                     // BE3
        return;
    }

    // ---- dx_t and the first u chunk -> LDS; column-sum slab cleared
    {
        constexpr int CPR = D / 8;
        for (int id = tid; id < 48 * CPR; id += 64 * MB_CW) {
            const int r = id / CPR, c = id % CPR;
            uint4 v = uint4{0u, 0u, 0u, 0u};
            if (r < n) v = *reinterpret_cast<const uint4*>(dxt + (row0 + r) * D + c * 8);
            *reinterpret_cast<uint4*>(XN + r * Ly::XN_PITCH + c * 16) = v;
        }
        for (int id = tid; id < 3 * mlp; id += 64 * MB_CW) CS[id] = 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                         // B0

    // the LayerNorm tail's operands (this thread's x1 / dx row chunks, gamma) are requested now (see the forward)
    f32x4 x1r[KT], dxr[KT], gmr[KT];
    {
        const int r = 4 * wave + g;
#pragma unroll
        for (int c = 0; c < KT; ++c) {
            const int col = 4 * (li + 16 * c);
            gmr[c] = *reinterpret_cast<const f32x4*>(ln2_w + col);
            x1r[c] = r < n ? ld_res4<R>(x1 + (row0 + r) * D + col) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (sizeof(R) == 2) dxr[c] = r < n ? ld_res4<bf16>(dxt + (row0 + r) * D + col) : f32x4{0.f, 0.f, 0.f, 0.f};
            else dxr[c] = r < n ? ld_res4<R>(dx + (row0 + r) * D + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const int ct = wave & 3, rt = wave >> 2;
    // this wave's 16 rows of the A operand (all k) are the same for every hidden chunk: read once; per chunk the weight fragments are
    // requested together and feed two accumulator chains (see attn_block.hip block_mma)
    Frag<bf16> fa_c[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) fa_c[ks].v = *reinterpret_cast<const bf16x8*>(XN + (16 * rt + li) * Ly::XN_PITCH + (ks * 32 + 8 * g) * 2);
    f32x4 yacc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) yacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < NC; ++c) {
        char* US = HC + (c & 1) * 2 * Ly::HC_BYTES;
        char* DS = US + Ly::HC_BYTES;
        __builtin_amdgcn_s_barrier();                                     // W2^T chunk c and u chunk c landed (block 2c)
        if (rt < RT) {
            const char* Wb = WR + ((2 * c) % Ly::NSTAGE) * Ly::WBLK;
            const int wrow = 16 * ct + li;
            Frag<bf16> fw[KSTEPS];
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks)
                fw[ks].v = *reinterpret_cast<const bf16x8*>(Wb + (ks >> 1) * 8192 + wrow * 128 + ((((ks & 1) * 4 + g) ^ (wrow & 7)) << 4));
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ks += 2) {
                acc = mma16(fa_c[ks], fw[ks], acc);
                acc1 = mma16(fa_c[ks + 1], fw[ks + 1], acc1);
            }
            acc = acc + acc1;
            float csum = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int off = (16 * rt + 4 * g + r) * Ly::HC_PITCH + (16 * ct + li) * 2;
                const float uv = (float)*reinterpret_cast<const bf16*>(US + (16 * rt + 4 * g + r) * 128 + (16 * ct + li) * 2);   // DMA image: 128-byte rows
                const bf16 db_ = (bf16)(acc[r] * gelu_grad_fast(uv));
                *reinterpret_cast<bf16*>(DS + off) = db_;
                csum += (float)db_;
            }
            csum = xor16_sum(csum);
            csum = xor32_sum(csum);
            if (g == 0) CS[rt * mlp + 64 * c + 16 * ct + li] = csum;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                     // du chunk visible; W1^T chunk c landed (block 2c + 1)
        for (int id = tid; id < n * 8; id += 64 * MB_CW) {                // du chunk -> global, 128-byte row segments
            const int r = id >> 3, cc = id & 7;
            *reinterpret_cast<uint4*>(du_out + (row0 + r) * mlp + 64 * c + cc * 8) = *reinterpret_cast<const uint4*>(DS + r * Ly::HC_PITCH + cc * 16);
        }
        if (rt < RT) {
            const char* Wb = WR + ((2 * c + 1) % Ly::NSTAGE) * Ly::WBLK;
            Frag<bf16> fa[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[ks].v = *reinterpret_cast<const bf16x8*>(DS + (16 * rt + li) * Ly::HC_PITCH + (ks * 32 + 8 * g) * 2);
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
    __builtin_amdgcn_s_barrier();                                         // BE1: ring free, all column sums written
    float* Y = reinterpret_cast<float*>(WR);
    constexpr int YLD = Ly::Y_PITCH / 4;
    if (rt < RT) {
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Y[(16 * rt + 4 * g + r) * YLD + 16 * (ct + 4 * j) + li] = yacc[j][r];
    }
    for (int id = tid; id < mlp; id += 64 * MB_CW) cs_part[(long)b * mlp + id] = (CS[id] + CS[mlp + id]) + CS[2 * mlp + id];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                         // BE2: dxn2 complete
    // ---- LN2 backward, row = 4 wave + g on 16 lanes; partial sums over the wave's 4 rows -> LP[wave][3][D]
    float* LP = reinterpret_cast<float*>(smem);                           // 12 x 3 x D floats (XN + HC are free now)
    {
        const int r = 4 * wave + g;
        const bool ok = r < n;
        f32x4 xh[KT], dy[KT];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < KT; ++c) {
            xh[c] = x1r[c];
            s += (xh[c][0] + xh[c][1]) + (xh[c][2] + xh[c][3]);
        }
        const float mean = row16_sum_mb(s) / D;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < KT; ++c) {
            xh[c] = xh[c] - mean;
            q += (xh[c][0] * xh[c][0] + xh[c][1] * xh[c][1]) + (xh[c][2] * xh[c][2] + xh[c][3] * xh[c][3]);
        }
        const float rstd = rsqrtf(row16_sum_mb(q) / D + eps);
        float s1 = 0.f, s2 = 0.f;
        f32x4 gd[KT];
#pragma unroll
        for (int c = 0; c < KT; ++c) {
            xh[c] = xh[c] * rstd;
            dy[c] = ok ? *reinterpret_cast<const f32x4*>(Y + r * YLD + 4 * (li + 16 * c)) : f32x4{0.f, 0.f, 0.f, 0.f};
            gd[c] = dy[c] * gmr[c];
            s1 += (gd[c][0] + gd[c][1]) + (gd[c][2] + gd[c][3]);
            const f32x4 t = gd[c] * xh[c];
            s2 += (t[0] + t[1]) + (t[2] + t[3]);
        }
        s1 = row16_sum_mb(s1) / D;
        s2 = row16_sum_mb(s2) / D;
#pragma unroll
        for (int c = 0; c < KT; ++c) {
            const int col = 4 * (li + 16 * c);
            f32x4 rr = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok) {
                rr = (gd[c] - s1 - xh[c] * s2) * rstd + dxr[c];
                if (sizeof(R) == 4) st_res4<R>(dx + (row0 + r) * D + col, rr);
                bf16x4 pk;
                pk[0] = (bf16)rr[0]; pk[1] = (bf16)rr[1]; pk[2] = (bf16)rr[2]; pk[3] = (bf16)rr[3];
                *reinterpret_cast<bf16x4*>(dx1t_out + (row0 + r) * D + col) = pk;
            }
            f32x4 pg = dy[c] * xh[c], pb = dy[c], pc = rr;                // sums over the wave's 4 rows (lanes that differ in g)
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
    for (int id = tid; id < 3 * D; id += 64 * MB_CW) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < MB_CW; ++w) s += LP[w * 3 * D + id];
        ln_part[(long)b * 3 * D + id] = s;
    }
}

#ifndef M3L_BLOCK_BODIES_ONLY
template <int KT, typename R>
__global__ __launch_bounds__(MB_THREADS) void mlp_block_fwd_kernel(const bf16* __restrict__ xn2, const R* __restrict__ x1,
                                                                     const bf16* __restrict__ W1, const float* __restrict__ b1,
                                                                     const bf16* __restrict__ W2, const float* __restrict__ b2, int n,
                                                                     int mlp, bf16* __restrict__ u_out, bf16* __restrict__ h_out,
                                                                     R* __restrict__ xout) {
    mlp_block_fwd_body<KT, R>(xn2, x1, W1, b1, W2, b2, n, mlp, u_out, h_out, xout);
}
template <int KT, typename R>
__global__ __launch_bounds__(MB_THREADS) void mlp_block_bwd_kernel(const bf16* __restrict__ dxt, R* __restrict__ dx,
                                                                     const R* __restrict__ x1, const float* __restrict__ ln2_w,
                                                                     const bf16* __restrict__ u, const bf16* __restrict__ W2T,
                                                                     const bf16* __restrict__ W1T, float eps, int n, int mlp,
                                                                     bf16* __restrict__ du_out, bf16* __restrict__ dx1t_out,
                                                                     float* __restrict__ cs_part, float* __restrict__ ln_part) {
    mlp_block_bwd_body<KT, R>(dxt, dx, x1, ln2_w, u, W2T, W1T, eps, n, mlp, du_out, dx1t_out, cs_part, ln_part);
}
#endif

}  // namespace

#ifndef M3L_BLOCK_BODIES_ONLY

static size_t mb_fwd_lds(int kt, int mlp);
int m3l_mlp_block_supported(int dtype, int D, int mlp, int n) {
    return m3l_attn_block_supported(dtype, D, D / 64, n, 1) && mlp % 64 == 0 && mlp >= 64 && mb_fwd_lds(D / 64, mlp) <= 160 * 1024;
}

// LDS of the forward: the layout + [mlp] floats (b1); of the backward: the layout + [3][mlp] floats of column sums
static size_t mb_fwd_lds(int kt, int mlp) { return (kt == 2 ? MbLayout<2>::TOTAL : MbLayout<3>::TOTAL) + (size_t)mlp * sizeof(float); }
static size_t mb_bwd_lds(int kt, int mlp) { return (kt == 2 ? MbLayout<2>::TOTAL : MbLayout<3>::TOTAL) + (size_t)3 * mlp * sizeof(float); }

int m3l_mlp_block_bwd_supported(int dtype, int D, int mlp, int n) {
    return m3l_mlp_block_supported(dtype, D, mlp, n) && mb_bwd_lds(D / 64, mlp) <= 160 * 1024;
}

int m3l_mlp_block_bwd(int D, int mlp, int B, int n, const void* dxt, float* dx, const float* x1, const float* ln2_w, const void* u,
                      const void* w2T, const void* w1T, float eps, void* du, void* dx1t, float* cs_part, float* ln_part, hipStream_t st) {
    static int inited_mlp = 0;
    M3L_CHECK(D == 128 || D == 192, "mlp_block_bwd: D=%d unsupported", D);
    const size_t lds = mb_bwd_lds(D / 64, mlp);
    if (inited_mlp != mlp) {
        M3L_HIP(hipFuncSetAttribute((const void*)mlp_block_bwd_kernel<2, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mb_bwd_lds(2, mlp)));
        M3L_HIP(hipFuncSetAttribute((const void*)mlp_block_bwd_kernel<3, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mb_bwd_lds(3, mlp)));
        M3L_HIP(hipFuncSetAttribute((const void*)mlp_block_bwd_kernel<2, bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mb_bwd_lds(2, mlp)));
        M3L_HIP(hipFuncSetAttribute((const void*)mlp_block_bwd_kernel<3, bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mb_bwd_lds(3, mlp)));
        inited_mlp = mlp;
    }
    ProfScope prof("mlp_block_bwd", B, n, mlp, 4.0 * B * n * (double)D * mlp, st,
                   (double)B * n * (D * 4.0 + (m3l_call_rb() ? 2.0 : 12.0) * D + mlp * 4.0));
    // (m3l_call_rb(): the residual stream of this launch — x1; dx is then unused — is bf16)
#define MBB_LAUNCH(KT, R)                                                                                                                  \
    mlp_block_bwd_kernel<KT, R><<<B, MB_THREADS, lds, st>>>((const bf16*)dxt, (R*)dx, (const R*)x1, ln2_w, (const bf16*)u, (const bf16*)w2T,     \
                                                           (const bf16*)w1T, eps, n, mlp, (bf16*)du, (bf16*)dx1t, cs_part, ln_part)
    if (m3l_call_rb()) { if (D == 128) MBB_LAUNCH(2, bf16); else MBB_LAUNCH(3, bf16); }
    else if (D == 128) MBB_LAUNCH(2, float);
    else MBB_LAUNCH(3, float);
#undef MBB_LAUNCH
    M3L_LAUNCH_CHECK();
    return 0;
}

#if MB_STAMP
extern "C" int m3l_mb_set_stamps(void* p) { return hipMemcpyToSymbol(HIP_SYMBOL(d_mb_stamps), &p, sizeof(p)) == hipSuccess ? 0 : 1; }
#endif
int m3l_mlp_block_fwd(int D, int mlp, int B, int n, const void* xn2, const float* x1, const void* w1, const float* b1, const void* w2,
                      const float* b2, void* u, void* h, float* xout, hipStream_t st) {
    static int inited_mlp = 0;
    if (inited_mlp != mlp) {
        M3L_HIP(hipFuncSetAttribute((const void*)mlp_block_fwd_kernel<2, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mb_fwd_lds(2, mlp)));
        M3L_HIP(hipFuncSetAttribute((const void*)mlp_block_fwd_kernel<3, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mb_fwd_lds(3, mlp)));
        M3L_HIP(hipFuncSetAttribute((const void*)mlp_block_fwd_kernel<2, bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mb_fwd_lds(2, mlp)));
        M3L_HIP(hipFuncSetAttribute((const void*)mlp_block_fwd_kernel<3, bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mb_fwd_lds(3, mlp)));
        inited_mlp = mlp;
    }
    M3L_CHECK(D == 128 || D == 192, "mlp_block: D=%d unsupported", D);
    ProfScope prof("mlp_block_fwd", B, n, mlp, 4.0 * B * n * (double)D * mlp, st,
                   (double)B * n * (D * 2.0 + 2.0 * (m3l_call_rb() ? 2.0 : 4.0) * D + mlp * (h ? 4.0 : 2.0)));
#define MB_LAUNCH(KT, R)                                                                                                                  \
    mlp_block_fwd_kernel<KT, R><<<B, MB_THREADS, mb_fwd_lds(KT, mlp), st>>>((const bf16*)xn2, (const R*)x1, (const bf16*)w1, b1, (const bf16*)w2, b2, n, \
                                                                           mlp, (bf16*)u, (bf16*)h, (R*)xout)
    if (m3l_call_rb()) { if (D == 128) MB_LAUNCH(2, bf16); else MB_LAUNCH(3, bf16); }
    else if (D == 128) MB_LAUNCH(2, float);
    else MB_LAUNCH(3, float);
#undef MB_LAUNCH
    M3L_LAUNCH_CHECK();
    return 0;
}
#endif  // M3L_BLOCK_BODIES_ONLY
