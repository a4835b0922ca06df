// MFMA GEMMs of the M3L MAE path for gfx950 (wave64, v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x4_f32).
//
//   gemm_nt : C[M,N] = A[M,K] * W[N,K]^T  (+bias, GELU, gelu'(u) multiply, f32 residual) — every nn.Linear
//             forward of the path (reference: vit_pytorch Attention.to_qkv / to_out, FeedForward.net[1],[4];
//             models/pretrain_models.py:111,115,116,770,777) and, with the pre-transposed weight copy, every dgrad.
//   gemm_tn : dW[N,K] = sum_m Y[m,N]^T X[m,K]   (weight gradients), split over m with deterministic f32 slabs.
//
// Tile: 128 x 128 per 256-thread workgroup (2x2 waves of 64x64 = 4x4 MFMA tiles), register-staged double-buffered
// LDS (global_load_dwordx4 issued before the MFMA block, ds_write after it: guide T14), epilogue staged through LDS
// so that every global store is a 16-byte row segment.
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "common.cuh"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128;

template <typename T> struct NtCfg;
template <> struct NtCfg<bf16> { static constexpr int BK = 64, KSTEPS = 2, ROW = 72; };    // 144-byte padded LDS rows
template <> struct NtCfg<float> { static constexpr int BK = 32, KSTEPS = 1, ROW = 36; };

constexpr int NT_STAGE_BYTES = 128 * 144;           // one operand tile
constexpr int NT_LDS_BYTES = 4 * NT_STAGE_BYTES;    // A,B x 2 stages = 73,728 B ; epilogue tile 128*132*4 = 67,584 B fits
constexpr int CS_LD = 132;

template <typename T>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const T* __restrict__ A, int lda, const T* __restrict__ W, int ldw,
                                                        int M, int N, int K, GemmEpi epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using C = NtCfg<T>;
    constexpr int EPC = Chunk<T>::N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, g = lane >> 4, li = lane & 15;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    auto As = [&](int st) { return reinterpret_cast<T*>(smem + (2 * st) * NT_STAGE_BYTES); };
    auto Bs = [&](int st) { return reinterpret_cast<T*>(smem + (2 * st + 1) * NT_STAGE_BYTES); };

    // register staging (named registers + macros: lambdas capturing the arrays pushed them to scratch)
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    const uint4 zero4 = {0u, 0u, 0u, 0u};
#define NT_GLOAD1(i, RA, RB, k0)                                                                              \
    {                                                                                                         \
        const int c = tid + 256 * (i), row = c >> 3, cc = c & 7;                                             \
        const int k = (k0) + cc * EPC;                                                                        \
        RA = (m0 + row < M && k < K) ? *reinterpret_cast<const uint4*>(A + (long)(m0 + row) * lda + k) : zero4; \
        RB = (n0 + row < N && k < K) ? *reinterpret_cast<const uint4*>(W + (long)(n0 + row) * ldw + k) : zero4; \
    }
#define NT_GLOAD(k0) NT_GLOAD1(0, ra0, rb0, k0) NT_GLOAD1(1, ra1, rb1, k0) NT_GLOAD1(2, ra2, rb2, k0) NT_GLOAD1(3, ra3, rb3, k0)
#define NT_SWRITE1(i, RA, RB, st)                                                                             \
    {                                                                                                         \
        const int c = tid + 256 * (i), row = c >> 3, cc = c & 7;                                             \
        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(As(st)) + row * 144 + cc * 16) = RA;               \
        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(Bs(st)) + row * 144 + cc * 16) = RB;               \
    }
#define NT_SWRITE(st) NT_SWRITE1(0, ra0, rb0, st) NT_SWRITE1(1, ra1, rb1, st) NT_SWRITE1(2, ra2, rb2, st) NT_SWRITE1(3, ra3, rb3, st)

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (K + C::BK - 1) / C::BK;
    NT_GLOAD(0)
    NT_SWRITE(0)
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        if (t + 1 < nk) { NT_GLOAD((t + 1) * C::BK) }
#pragma unroll
        for (int ks = 0; ks < C::KSTEPS; ++ks) {
            Frag<T> fa[4], fb[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                fa[x] = load_kc(As(cur) + (wr * 64 + x * 16 + li) * C::ROW + ks * 32 + g * 8);
                fb[x] = load_kc(Bs(cur) + (wc * 64 + x * 16 + li) * C::ROW + ks * 32 + g * 8);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = mma16(fa[a], fb[b], acc[a][b]);
        }
        if (t + 1 < nk) { NT_SWRITE(cur ^ 1) }
        __syncthreads();
    }
#undef NT_GLOAD
#undef NT_GLOAD1
#undef NT_SWRITE
#undef NT_SWRITE1

    // ---- epilogue: accumulators -> LDS (f32) -> 16-byte row segments ---------------------------------------
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Cs[(wr * 64 + a * 16 + 4 * g + r) * CS_LD + wc * 64 + b * 16 + li] = acc[a][b][r];
    __syncthreads();

    const int cchunk = tid & 15, r0 = tid >> 4;
    const int col = n0 + cchunk * 8;
    if (col >= N) return;
    float bias[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bias[j] = (epi.bias && col + j < epi.n_bias) ? epi.bias[col + j] : 0.f;
    T* out_t = reinterpret_cast<T*>(epi.out_t);
    T* out_pre = reinterpret_cast<T*>(epi.out_pre);
    const T* gelu_u = reinterpret_cast<const T*>(epi.gelu_u);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int rl = r0 + 16 * i, row = m0 + rl;
        if (row >= M) continue;
        float v[8];
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(Cs + rl * CS_LD + cchunk * 8);
        const f32x4 p1 = *reinterpret_cast<const f32x4*>(Cs + rl * CS_LD + cchunk * 8 + 4);
        v[0] = p0[0]; v[1] = p0[1]; v[2] = p0[2]; v[3] = p0[3];
        v[4] = p1[0]; v[5] = p1[1]; v[6] = p1[2]; v[7] = p1[3];
        const long o = (long)row * epi.ldc + col;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * epi.alpha + bias[j];
        if (gelu_u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= gelu_grad_t<T>(to_f32(gelu_u[o + j]));
        }
        if (out_pre) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const T u = from_f32<T>(v[j]);
                out_pre[o + j] = u;
                v[j] = to_f32(u);
            }
        }
        if (epi.act == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = gelu_t<T>(v[j]);
        } else if (epi.act == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if (epi.relu_ref) {
            const T* rr = reinterpret_cast<const T*>(epi.relu_ref);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = to_f32(rr[o + j]) > 0.f ? v[j] : 0.f;
        }
        if (epi.res) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += epi.res[o + j];
        }
        if (epi.out_f32) {
            *reinterpret_cast<f32x4*>(epi.out_f32 + o) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(epi.out_f32 + o + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
        if (out_t) {
#pragma unroll
            for (int j = 0; j < 8; ++j) out_t[o + j] = from_f32<T>(v[j]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// gemm_nt, main path: LDS-DMA staging (global_load_lds_dwordx4, no VGPR round trip), unpadded 128-byte LDS rows with the
// 16-byte chunk index XOR-ed by (row & 7) — applied on the per-lane SOURCE address (the DMA destination is lane-linear)
// and again on the fragment read (guide rule 21) -> conflict-free ds_read_b128.  Two stages; the next K-tile's DMA is
// issued before the MFMA block of the current one.  Needs K % (128 / sizeof(T)) == 0; rows past M / N are clamped (their
// products are never stored).
// EPI: epilogue specialisation resolved at compile time (0 = generic, every GemmEpi flag tested at run time; the four hot
// combinations of the transformer get branch-free code the compiler can schedule across the unrolled row segments):
//   1 FC1    bias + pre-activation copy + GELU -> out_t          2 DGELU  * gelu'(u) -> out_t (+ column-sum partials)
//   3 PLAIN  (bias) -> out_t                                      4 RES32  bias + fp32 residual -> out_f32
enum { EPI_GENERIC = 0, EPI_FC1 = 1, EPI_DGELU = 2, EPI_PLAIN = 3, EPI_RES32 = 4, EPI_RES16 = 5 };
template <typename T, int BN, int R, int BM = 128, int EPI = 0>
__global__ __launch_bounds__(256) void gemm_nt_glds_kernel(const T* __restrict__ A, int lda, const T* __restrict__ W, int ldw,
                                                             int M, int N, int K, GemmEpi epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int EPC = Chunk<T>::N;             // elements per 16-byte chunk
    constexpr int BKE = 8 * EPC;                 // elements per 128-byte LDS row (K-tile depth): 64 bf16 / 32 f32
    constexpr int KSTEPS = BKE / 32;             // MFMA macro steps per K-tile
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int WM = (BN == 128) ? BM / 2 : BM / 4;   // rows per wave: 2x2 waves (BN = 128) or 4x1 waves (BN = 64), 64 columns each
    constexpr int ASEG = BM / 32;                // 1-KiB A segments (8 rows) per wave and K-tile
    constexpr int TM = WM / 16;
    constexpr int CSLD = BN + 4;
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef __attribute__((address_space(1))) const void* gl_vp;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int wr = (BN == 128) ? (wave >> 1) : wave, wc = (BN == 128) ? (wave & 1) : 0;
    // XCD-aware tile order (guide T1): workgroup ids are dealt round-robin over the 8 XCDs, so ids that differ by a
    // multiple of 8 share an L2.  All column tiles of one 128-row block get ids b, b+8, b+16, ... -> the A rows they
    // share are fetched from HBM once per XCD-L2 instead of once per column tile.  Placement only affects speed.
    const int gx = (N + BN - 1) / BN, gy = (M + BM - 1) / BM;
    const int bid = blockIdx.x, rest = bid >> 3;
    const int bx = rest % gx, by = (rest / gx) * 8 + (bid & 7);
    if (by >= gy) return;
    const int m0 = by * BM, n0 = bx * BN;

    // per-lane source pointers of this wave's DMA segments (1 KiB = 8 rows x 8 chunks each)
    const int srow = lane >> 3, spc = lane & 7;
    const T* asrc[ASEG];
#pragma unroll
    for (int i = 0; i < ASEG; ++i) {
        const int row = (wave * ASEG + i) * 8 + srow;
        asrc[i] = A + (long)min(m0 + row, M - 1) * lda + (spc ^ (row & 7)) * EPC;
    }
    constexpr int BSEG = BN / 32;                // B segments per wave: 4 (BN=128) or 2 (BN=64)
    const T* bsrc[BSEG];
#pragma unroll
    for (int i = 0; i < BSEG; ++i) {
        const int row = (wave * BSEG + i) * 8 + srow;
        bsrc[i] = W + (long)min(n0 + row, N - 1) * ldw + (spc ^ (row & 7)) * EPC;
    }
    auto stage = [&](int st, int k0) {
        char* base = smem + st * STAGE;
#pragma unroll
        for (int i = 0; i < ASEG; ++i)
            __builtin_amdgcn_global_load_lds((gl_vp)(asrc[i] + k0), (lds_vp)(base + (wave * ASEG + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < BSEG; ++i)
            __builtin_amdgcn_global_load_lds((gl_vp)(bsrc[i] + k0), (lds_vp)(base + A_BYTES + (wave * BSEG + i) * 1024), 16, 0, 0);
    };
    // fragment of macro step ks for LDS row `row` of a tile at `base`
    auto frag = [&](const char* base, int row, int ks) {
        Frag<T> f;
        if constexpr (sizeof(T) == 2) {
            f.v = *reinterpret_cast<const bf16x8*>(base + row * 128 + (((ks * 4 + g) ^ (row & 7)) << 4));
        } else {
            const f32x4 a = *reinterpret_cast<const f32x4*>(base + row * 128 + (((2 * g) ^ (row & 7)) << 4));
            const f32x4 b = *reinterpret_cast<const f32x4*>(base + row * 128 + (((2 * g + 1) ^ (row & 7)) << 4));
            f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
            f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
        }
        return f;
    };

    f32x4 acc[TM][4];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // R-deep ring of K-tiles.  R == 2: classic double buffer (the barrier's vmcnt(0) drains the prefetch, so only one DMA
    // latency is hidden per tile).  R == 3: two tiles stay in flight across the barrier — counted s_waitcnt vmcnt(N) + raw
    // s_barrier (guide "Pipelining across barriers"): with K = 192 (3 tiles) the whole panel is requested up front and the
    // workgroup pays ONE DMA latency instead of three.
    constexpr int DMA_PER_STAGE = ASEG + BSEG;     // LDS-DMA instructions per wave and K-tile
    static_assert(R != 3 || DMA_PER_STAGE == 6 || DMA_PER_STAGE == 8, "counted vmcnt immediates exist for 6 / 8 DMA per stage");
    const int nk = K / BKE;
    if (R != 1) stage(0, 0);
    if (R == 3 && nk > 1) stage(1, BKE);
    if (R == 2) __syncthreads();                   // vmcnt(0) + barrier: tile 0 landed for every wave
    for (int t = 0; t < nk; ++t) {
        int cur;
        if (R == 1) {                              // no ring: latency is hidden by co-resident workgroups only
            cur = 0;
            stage(0, t * BKE);
            __syncthreads();
        } else if (R == 2) {
            cur = t & 1;
            if (t + 1 < nk) stage(cur ^ 1, (t + 1) * BKE);
        } else {
            cur = t % 3;
            if (t + 1 < nk) {
                if (DMA_PER_STAGE == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();          // tile t landed for every wave; every wave is done reading tile t-1
            if (t + 2 < nk) stage((t + 2) % 3, (t + 2) * BKE);
        }
        const char* As = smem + cur * STAGE;
        const char* Bs = As + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            Frag<T> fa[TM], fb[4];
#pragma unroll
            for (int x = 0; x < TM; ++x) fa[x] = frag(As, wr * WM + x * 16 + li, ks);
#pragma unroll
            for (int y = 0; y < 4; ++y) fb[y] = frag(Bs, wc * 64 + y * 16 + li, ks);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = mma16(fa[a], fb[b], acc[a][b]);
        }
        if (R == 2 || R == 1) __syncthreads();     // next tile landed (vmcnt(0)) and this one fully read
    }
    if (R == 3) __syncthreads();                   // all tiles consumed before the stages are reused by the epilogue

    // ---- epilogue: accumulators -> LDS (f32) -> 16-byte row segments -----------------------------------------
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(wr * WM + a * 16 + 4 * g + r) * CSLD + wc * 64 + b * 16 + li] = acc[a][b][r];
    constexpr int CPR = BN / 8, RSTEP = 256 / CPR, ITERS = BM / RSTEP;
    const int cchunk = tid % CPR, r0 = tid / CPR;
    const int col = n0 + cchunk * 8;
    const bool col_ok = col < N;
    float bias[8], csum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bias[j] = (epi.bias && col_ok && col + j < epi.n_bias) ? epi.bias[col + j] : 0.f;
        csum[j] = 0.f;
    }
    T* out_t = reinterpret_cast<T*>(epi.out_t);
    T* out_pre = reinterpret_cast<T*>(epi.out_pre);
    const T* gelu_u = reinterpret_cast<const T*>(epi.gelu_u);
    constexpr bool GEN = (EPI == EPI_GENERIC);
    const bool f_gelu_u = GEN ? (gelu_u != nullptr) : (EPI == EPI_DGELU);
    const bool f_out_pre = GEN ? (out_pre != nullptr) : (EPI == EPI_FC1);
    const int f_act = GEN ? epi.act : (EPI == EPI_FC1 ? 1 : 0);
    const bool f_relu_ref = GEN ? (epi.relu_ref != nullptr) : false;
    const bool f_res = GEN ? (epi.res != nullptr) : (EPI == EPI_RES32);
    const bool f_out32 = GEN ? (epi.out_f32 != nullptr) : (EPI == EPI_RES32);
    const bool f_out_t = GEN ? (out_t != nullptr) : (EPI != EPI_RES32);
    const bool f_colsum = GEN ? (epi.colsum_part != nullptr) : (EPI == EPI_DGELU && epi.colsum_part != nullptr);
    const bool f_res16 = sizeof(T) == 2 && (GEN ? (epi.res_t != nullptr) : (EPI == EPI_RES16));     // bf16 residual (shares ruu with gelu_u)
    // The epilogue's row operands (fp32 residual, pre-activation u) of ALL this thread's row segments are requested here, before the
    // accumulators' trip through LDS: loaded inside the store loop below, segment i's loads came behind segment i - 1's stores, and a
    // wave's vmcnt covers both kinds — every segment waited for the previous segment's HBM write as well as for its own read (four
    // dependent round trips per 128-row tile).  Rows past M read row M - 1 (dropped).
    f32x4 rq0[ITERS], rq1[ITERS];
    Frag<T> ruu[ITERS];
    if (col_ok && (f_res || f_gelu_u || f_res16)) {
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const long o = (long)min(m0 + r0 + RSTEP * i, M - 1) * epi.ldc + col;
            if (f_res) {
                rq0[i] = *reinterpret_cast<const f32x4*>(epi.res + o);
                rq1[i] = *reinterpret_cast<const f32x4*>(epi.res + o + 4);
            }
            if (f_gelu_u) {
                if constexpr (sizeof(T) == 2) {
                    ruu[i].v = *reinterpret_cast<const bf16x8*>(gelu_u + o);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) ruu[i].v[j] = gelu_u[o + j];
                }
            }
            if constexpr (sizeof(T) == 2) {
                if (f_res16) ruu[i].v = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const T*>(epi.res_t) + o);
            }
        }
    }
    __syncthreads();

    // gelu'(u) of every segment before the first store, in straight-line code: the segments below are basic blocks of their own (the row
    // guard), and at each block entry the compiler must assume the loads above still pending next to the previous block's store -> vmcnt(0)
    float gfac[GEN || EPI == EPI_DGELU ? ITERS : 1][8];
    if (col_ok && f_gelu_u) {
#pragma unroll
        for (int i = 0; i < ITERS; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f = gelu_grad_t<T>(to_f32(ruu[i].v[j]));
                asm volatile("" : "+v"(f));                    // evaluated HERE (left alone the compiler sinks it into the segment that uses it)
                gfac[GEN || EPI == EPI_DGELU ? i : 0][j] = f;
            }
    }
    if (col_ok) {
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int rl = r0 + RSTEP * i, row = m0 + rl;
            if (row >= M) break;
            float v[8];
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(Cs + rl * CSLD + cchunk * 8);
            const f32x4 p1 = *reinterpret_cast<const f32x4*>(Cs + rl * CSLD + cchunk * 8 + 4);
            v[0] = p0[0]; v[1] = p0[1]; v[2] = p0[2]; v[3] = p0[3];
            v[4] = p1[0]; v[5] = p1[1]; v[6] = p1[2]; v[7] = p1[3];
            const long o = (long)row * epi.ldc + col;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] * epi.alpha + bias[j];
            if (f_gelu_u) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= gfac[GEN || EPI == EPI_DGELU ? i : 0][j];
            }
            if (f_out_pre) {
                Frag<T> pk;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    pk.v[j] = from_f32<T>(v[j]);
                    v[j] = to_f32(pk.v[j]);
                }
                if constexpr (sizeof(T) == 2) {
                    *reinterpret_cast<bf16x8*>(out_pre + o) = pk.v;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) out_pre[o + j] = pk.v[j];
                }
            }
            if (f_act == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = gelu_t<T>(v[j]);
            } else if (f_act == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            if (f_relu_ref) {
                const T* rr = reinterpret_cast<const T*>(epi.relu_ref);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = to_f32(rr[o + j]) > 0.f ? v[j] : 0.f;
            }
            if (f_res) {
                const f32x4 q0 = rq0[i], q1 = rq1[i];
                v[0] += q0[0]; v[1] += q0[1]; v[2] += q0[2]; v[3] += q0[3];
                v[4] += q1[0]; v[5] += q1[1]; v[6] += q1[2]; v[7] += q1[3];
            }
            if (f_res16) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += to_f32(ruu[i].v[j]);
            }
            if (f_out32) {
                *reinterpret_cast<f32x4*>(epi.out_f32 + o) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(epi.out_f32 + o + 4) = f32x4{v[4], v[5], v[6], v[7]};
            }
            if (f_out_t) {
                Frag<T> pk;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    pk.v[j] = from_f32<T>(v[j]);
                    csum[j] += to_f32(pk.v[j]);
                }
                if constexpr (sizeof(T) == 2) {
                    *reinterpret_cast<bf16x8*>(out_t + o) = pk.v;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) out_t[o + j] = pk.v[j];
                }
            }
        }
    }
    if (f_colsum) {
        // column sums of the compute-type output over this workgroup's rows -> part[blockIdx.y][N] (fixed order)
        float* red = reinterpret_cast<float*>(smem) + BM * CSLD;      // [RSTEP][BN]
#pragma unroll
        for (int j = 0; j < 8; ++j) red[r0 * BN + cchunk * 8 + j] = csum[j];
        __syncthreads();
        if (tid < BN && n0 + tid < N) {
            float s = 0.f;
            for (int r = 0; r < RSTEP; ++r) s += red[r * BN + tid];
            epi.colsum_part[(long)by * N + n0 + tid] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// gemm_tn, main path: both operands arrive by LDS-DMA with hardware bounds checking (buffer_load ... lds: rows past M read
// as zero, so ragged M needs no tail code), land in unpadded row-major [m][128] tiles whose 16-byte chunk index is XOR-ed
// with a row-dependent mask on the SOURCE side, and are consumed k-strided: bf16 through ds_read_b64_tr_b16 (the 8 rows a
// half-wave touches fall on 8 distinct 32-byte slots -> conflict-free), f32 through scalar reads.
template <typename T> struct TnSwz;
template <> struct TnSwz<bf16> {
    static constexpr int ROWB = 256, CPR = 16, RPI = 4, BMT = 64;
    static __device__ __forceinline__ int swz(int row) { return 2 * (row & 7); }
};
template <> struct TnSwz<float> {
    static constexpr int ROWB = 512, CPR = 32, RPI = 2, BMT = 32;
    static __device__ __forceinline__ int swz(int row) { return 4 * ((row >> 2) & 1); }
};
__device__ __forceinline__ Frag<bf16> load_ks_swz(const char* tile, int k0, int c0, int lane, bf16) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int r0 = k0 + 4 * g + q;
    const int chunk = (c0 >> 3) + (p >> 1);
    const int x = ((chunk ^ (2 * (r0 & 7))) << 4) + (p & 1) * 8;
    typedef __attribute__((address_space(3))) bf16x4* lds_p;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(tile + r0 * 256 + x));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(tile + (r0 + 16) * 256 + x));
    Frag<bf16> f;
    f.v[0] = lo[0]; f.v[1] = lo[1]; f.v[2] = lo[2]; f.v[3] = lo[3];
    f.v[4] = hi[0]; f.v[5] = hi[1]; f.v[6] = hi[2]; f.v[7] = hi[3];
    return f;
}
__device__ __forceinline__ Frag<float> load_ks_swz(const char* tile, int k0, int c0, int lane, float) {
    const int g = lane >> 4, i = lane & 15;
    const int col = c0 + i, chunk = col >> 2, off = (col & 3) * 4;
    Frag<float> f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = k0 + 4 * g + (j & 3) + 16 * (j >> 2);
        f.v[j] = *reinterpret_cast<const float*>(tile + r * 512 + ((chunk ^ (4 * ((r >> 2) & 1))) << 4) + off);
    }
    return f;
}

template <typename T>
__global__ __launch_bounds__(256) void gemm_tn_glds_kernel(TnGroup grp, int M, int m_per_split, int nsplit,
                                                             float* __restrict__ partial_base) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using S = TnSwz<T>;
    constexpr int ES = (int)sizeof(T), EPC = Chunk<T>::N;
    constexpr int TILE_B = S::BMT * S::ROWB;          // 16 KiB per operand tile
    constexpr int STAGE = 2 * TILE_B;
    constexpr int MS = S::BMT / 32;
    typedef __attribute__((address_space(3))) void* lds_vp;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, g = lane >> 4, li = lane & 15;
    // XCD-aware order: the (n, k) tiles of one m-split read the same rows of Y and X -> give them ids that differ by
    // multiples of 8 (same XCD L2).
    const int bid = blockIdx.x, rest = bid >> 3;
    const int gtile = rest % grp.tiles_total, sp = (rest / grp.tiles_total) * 8 + (bid & 7);
    if (sp >= nsplit) return;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < grp.count && gtile >= grp.p[i].tile0) pi = i;
    const TnProblem& pb = grp.p[pi];
    const T* Y = reinterpret_cast<const T*>(pb.Y);
    const T* X = reinterpret_cast<const T*>(pb.X);
    const int ldy = pb.ldy, ldx = pb.ldx, N = pb.N, K = pb.K;
    float* partial = partial_base + pb.part_off;
    const int gn = (N + 127) / 128;
    const int tile = gtile - pb.tile0;
    const int n0 = (tile % gn) * 128, k0 = (tile / gn) * 128;
    const int m_beg = sp * m_per_split;
    const int m_end = min(M, m_beg + m_per_split);

    const auto rsY = __builtin_amdgcn_make_buffer_rsrc((void*)Y, 0, (int)min((long)M * ldy * ES, 2147483647L), 0x00020000);
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)min((long)M * ldx * ES, 2147483647L), 0x00020000);
    unsigned voy[4], vox[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * S::RPI + lane / S::CPR;
        const int cc = (lane % S::CPR) ^ S::swz(row);
        const int cy = min(n0 + cc * EPC, N - EPC), cx = min(k0 + cc * EPC, K - EPC);     // columns past N / K: any valid chunk
        voy[i] = (unsigned)(((long)(m_beg + row) * ldy + cy) * ES);
        vox[i] = (unsigned)(((long)(m_beg + row) * ldx + cx) * ES);
    }
    auto stage = [&](int st, int t) {
        char* base = smem + st * STAGE;
        const unsigned ady = (unsigned)((long)t * S::BMT * ldy * ES), adx = (unsigned)((long)t * S::BMT * ldx * ES);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (lds_vp)(base + (wave * 4 + i) * 1024), 16, voy[i] + ady, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_vp)(base + TILE_B + (wave * 4 + i) * 1024), 16, vox[i] + adx, 0, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = (m_end - m_beg + S::BMT - 1) / S::BMT;     // m_per_split is a multiple of BMT: only the global tail is ragged
    if (nt > 0) stage(0, 0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) stage(cur ^ 1, t + 1);
        const char* Ys = smem + cur * STAGE;
        const char* Xs = Ys + TILE_B;
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
            Frag<T> fa[4], fb[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                fa[x] = load_ks_swz(Ys, ms * 32, wr * 64 + x * 16, lane, T());
                fb[x] = load_ks_swz(Xs, ms * 32, wc * 64 + x * 16, lane, T());
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = mma16(fa[a], fb[b], acc[a][b]);
        }
        __syncthreads();
    }
    float* P = partial + (long)sp * N * K;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wr * 64 + a * 16 + 4 * g + r, k = k0 + wc * 64 + b * 16 + li;
                if (n < N && k < K) P[(long)n * K + k] = acc[a][b][r];
            }
}

__global__ void reduce_splits_kernel(const float* __restrict__ partial, int S, int N, int K, float* __restrict__ out, int ldo,
                                     int nvalid, int kvalid, int accumulate) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * K) return;
    const int n = (int)(idx / K), k = (int)(idx % K);
    if (k >= kvalid || n >= nvalid) return;
    float s = 0.f;
    for (int i = 0; i < S; ++i) s += partial[(long)i * N * K + idx];
    float* o = out + (long)n * ldo + k;
    *o = accumulate ? (*o + s) : s;
}

__global__ void reduce_splits_grouped_kernel(TnGroup grp, const float* __restrict__ partial_base, int S, int accumulate) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if ((int)blockIdx.y == grp.count) {          // the extra column reduction (bias gradient partials)
        if (idx >= grp.extra_width) return;
        float s0 = 0.f, s1 = 0.f;
        int g = 0;
        for (; g + 1 < grp.extra_G; g += 2) {
            s0 += grp.extra_part[(long)g * grp.extra_width + idx];
            s1 += grp.extra_part[(long)(g + 1) * grp.extra_width + idx];
        }
        if (g < grp.extra_G) s0 += grp.extra_part[(long)g * grp.extra_width + idx];
        grp.extra_out[idx] = accumulate ? grp.extra_out[idx] + (s0 + s1) : (s0 + s1);
        return;
    }
    const TnProblem& pb = grp.p[blockIdx.y];
    const long cnt = (long)pb.N * pb.K;
    if (idx >= cnt) return;
    const int n = (int)(idx / pb.K), k = (int)(idx % pb.K);
    if (k >= pb.kvalid || n >= pb.nvalid) return;
    const float* P = partial_base + pb.part_off + idx;
    float s0 = 0.f, s1 = 0.f;
    int i = 0;
    for (; i + 1 < S; i += 2) {
        s0 += P[(long)i * cnt];
        s1 += P[(long)(i + 1) * cnt];
    }
    if (i < S) s0 += P[(long)i * cnt];
    float* o = pb.out + (long)n * pb.ldo + k;
    *o = accumulate ? (*o + (s0 + s1)) : (s0 + s1);
}

}  // namespace

static constexpr int glds_lds_bytes(int bn, int r = 2, int bm = 128) {
    const int stages = r * (bm * 128 + bn * 128), cs = bm * (bn + 4) * 4 + (256 / (bn / 8)) * bn * 4;
    return stages > cs ? stages : cs;
}
// row-tile height of the LDS-DMA NT GEMM for a problem: 64 rows while 128-row tiles would give fewer than ~4 workgroups per CU
static int nt_tile_rows(int M, int N) { return ((long)cdiv(M, 128) * cdiv(N, 64) < 1024) ? 64 : 128; }
static int g_gemm_inited = 0;
int m3l_gemm_init() {
    if (g_gemm_inited) return 0;
    M3L_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS_BYTES));
    M3L_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS_BYTES));
#define NT_ATTR(T, BMv, E) M3L_HIP(hipFuncSetAttribute((const void*)gemm_nt_glds_kernel<T, 64, 2, BMv, E>, hipFuncAttributeMaxDynamicSharedMemorySize, glds_lds_bytes(64, 2, BMv)))
#define NT_ATTR5(T, BMv) NT_ATTR(T, BMv, 0); NT_ATTR(T, BMv, 1); NT_ATTR(T, BMv, 2); NT_ATTR(T, BMv, 3); NT_ATTR(T, BMv, 4); NT_ATTR(T, BMv, 5)
    NT_ATTR5(bf16, 128); NT_ATTR5(float, 128); NT_ATTR5(bf16, 64); NT_ATTR5(float, 64);
#undef NT_ATTR5
#undef NT_ATTR
    M3L_HIP(hipFuncSetAttribute((const void*)gemm_tn_glds_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    g_gemm_inited = 1;
    return 0;
}

int m3l_gemm_nt(int dtype, const void* A, int lda, const void* W, int ldw, int M, int N, int K, const GemmEpi* epi_in,
                hipStream_t st) {
    if (m3l_gemm_init()) return 2;
    GemmEpi e = *epi_in;
    if (e.n_bias <= 0) e.n_bias = N;
    const GemmEpi* epi = &e;
    M3L_CHECK(dtype == 0 || dtype == 1, "gemm_nt: bad dtype %d", dtype);
    M3L_CHECK(M > 0 && N > 0 && K > 0, "gemm_nt: empty problem %d %d %d", M, N, K);
    M3L_CHECK(K % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldw % 8 == 0 && epi->ldc % 8 == 0,
              "gemm_nt: K,N,lda,ldw,ldc must be multiples of 8 (got K=%d N=%d lda=%d ldw=%d ldc=%d)", K, N, lda, ldw, epi->ldc);
    const int bke = dtype == 1 ? 64 : 32;
    M3L_CHECK(!epi->res_t || (dtype == 1 && K % bke == 0 && epi->out_t && !epi->res && !epi->gelu_u),
              "gemm_nt: a bf16 residual needs bf16 operands, K %% 64 == 0, a compute-type output, and neither an fp32 residual nor gelu_u");
    const double es = dtype ? 2.0 : 4.0;
    // algorithmic HBM bytes: A and W once, every epilogue operand / result once
    const double bytes = (double)M * K * es + (double)N * K * es +
                         (double)M * N * ((epi->res ? 4.0 : 0.0) + (epi->res_t ? 2.0 : 0.0) + (epi->out_f32 ? 4.0 : 0.0) + (epi->out_t ? es : 0.0) +
                                          (epi->out_pre ? es : 0.0) + (epi->gelu_u ? es : 0.0));
    const char* kind = (K % bke != 0) ? "gemm_nt_generic" : "gemm_nt_glds64";
    ProfScope prof(kind, M, N, K, 2.0 * M * N * K, st, bytes);
    if (K % bke == 0) {
        // 64-wide column tiles, 2-stage ring (48 KiB of LDS -> 3 workgroups per CU: measured faster than deeper rings or wider
        // tiles at 1-2 workgroups per CU); 64-row tiles when 128-row tiles would leave the chip under-filled
        const bool plain_alpha = epi->alpha == 1.0f && !epi->relu_ref;
        int mode = EPI_GENERIC;
        if (plain_alpha && epi->out_t && epi->out_pre && epi->act == 1 && !epi->gelu_u && !epi->res && !epi->res_t && !epi->out_f32 && !epi->colsum_part)
            mode = EPI_FC1;
        else if (plain_alpha && epi->out_t && epi->gelu_u && !epi->out_pre && epi->act == 0 && !epi->res && !epi->res_t && !epi->out_f32)
            mode = EPI_DGELU;
        else if (plain_alpha && epi->out_t && !epi->gelu_u && !epi->out_pre && epi->act == 0 && !epi->res && !epi->res_t && !epi->out_f32 && !epi->colsum_part)
            mode = EPI_PLAIN;
        else if (plain_alpha && !epi->out_t && !epi->gelu_u && !epi->out_pre && epi->act == 0 && epi->res && epi->out_f32 && !epi->colsum_part)
            mode = EPI_RES32;
        else if (plain_alpha && epi->out_t && epi->res_t && !epi->gelu_u && !epi->out_pre && epi->act == 0 && !epi->res && !epi->out_f32 && !epi->colsum_part)
            mode = EPI_RES16;
        const bool small = nt_tile_rows(M, N) == 64;
        const dim3 grid(8 * cdiv(N, 64) * cdiv(cdiv(M, small ? 64 : 128), 8));
#define NT_LAUNCH(T, BMv, E) gemm_nt_glds_kernel<T, 64, 2, BMv, E><<<grid, 256, glds_lds_bytes(64, 2, BMv), st>>>((const T*)A, lda, (const T*)W, ldw, M, N, K, *epi)
#define NT_MODES(T, BMv)                                       \
    switch (mode) {                                            \
        case EPI_FC1: NT_LAUNCH(T, BMv, EPI_FC1); break;       \
        case EPI_DGELU: NT_LAUNCH(T, BMv, EPI_DGELU); break;   \
        case EPI_PLAIN: NT_LAUNCH(T, BMv, EPI_PLAIN); break;   \
        case EPI_RES32: NT_LAUNCH(T, BMv, EPI_RES32); break;   \
        case EPI_RES16: NT_LAUNCH(T, BMv, EPI_RES16); break;   \
        default: NT_LAUNCH(T, BMv, EPI_GENERIC); break;        \
    }
        if (dtype == 1) {
            if (small) { NT_MODES(bf16, 64) } else { NT_MODES(bf16, 128) }
        } else {
            if (small) { NT_MODES(float, 64) } else { NT_MODES(float, 128) }
        }
#undef NT_MODES
#undef NT_LAUNCH
    } else {
        M3L_CHECK(epi->colsum_part == nullptr, "gemm_nt: colsum epilogue needs K %% %d == 0", bke);
        dim3 grid(cdiv(N, BN), cdiv(M, BM));
        if (dtype == 1)
            gemm_nt_kernel<bf16><<<grid, 256, NT_LDS_BYTES, st>>>((const bf16*)A, lda, (const bf16*)W, ldw, M, N, K, *epi);
        else
            gemm_nt_kernel<float><<<grid, 256, NT_LDS_BYTES, st>>>((const float*)A, lda, (const float*)W, ldw, M, N, K, *epi);
    }
    M3L_LAUNCH_CHECK();
    return 0;
}

int m3l_gemm_nt_colsum_rows(int M, int N) { return cdiv(M, nt_tile_rows(M, N)); }

static int tn_splits(int M, int tiles) {
    int S = cdiv(512, tiles);                 // ~2 workgroups per CU over 256 CUs
    const int max_s = cdiv(M, 256);           // at least 256 rows per split
    if (S > max_s) S = max_s;
    if (S < 1) S = 1;
    return S;
}

size_t m3l_gemm_tn_ws_bytes(int M, int N, int K, int* splits_out) {
    const int S = tn_splits(M, cdiv(N, 128) * cdiv(K, 128));
    if (splits_out) *splits_out = S;
    TnProblem p;
    memset(&p, 0, sizeof(p));
    p.N = N; p.K = K;
    return std::max((size_t)S * N * K * sizeof(float), m3l_wgrad_ws_bytes(M, &p, 1));
}

// workspace that serves either compute type (the f32 path launches chunks of <= 4 problems that reuse the same slabs)
size_t m3l_gemm_tn_grouped_ws_bytes(int M, const TnProblem* probs, int count) {
    size_t f32 = 0;
    for (int c0 = 0; c0 < count; c0 += 4) {
        int tiles = 0;
        size_t elems = 0;
        for (int i = c0; i < std::min(count, c0 + 4); ++i) {
            tiles += cdiv(probs[i].N, 128) * cdiv(probs[i].K, 128);
            elems += (size_t)probs[i].N * probs[i].K;
        }
        f32 = std::max(f32, (size_t)tn_splits(M, tiles > 0 ? tiles : 1) * elems * sizeof(float));
    }
    return std::max(f32, m3l_wgrad_ws_bytes(M, probs, count));
}

static int tn_grouped_f32(TnProblem* probs, int count, int M, float* partial_ws, size_t ws_bytes, int accumulate, hipStream_t st,
                          const float* extra_part, int extra_G, int extra_width, float* extra_out) {
    const int dtype = 0;
    TnGroup grp;
    memset(&grp, 0, sizeof(grp));
    int tiles = 0, maxnk = 0;
    long off = 0;
    double flops = 0, bytes = 0;
    for (int i = 0; i < count; ++i) {
        TnProblem& p = probs[i];
        M3L_CHECK(p.N > 0 && p.K > 0 && p.K % 8 == 0 && p.N % 8 == 0 && p.ldy % 8 == 0 && p.ldx % 8 == 0,
                  "gemm_tn: N,K,ldy,ldx must be positive multiples of 8");
        M3L_CHECK((long)M * p.ldy * 4 < 2147483647L && (long)M * p.ldx * 4 < 2147483647L, "gemm_tn: operand larger than 2 GiB");
        p.tile0 = tiles;
        tiles += cdiv(p.N, 128) * cdiv(p.K, 128);
        if (p.N * p.K > maxnk) maxnk = p.N * p.K;
        flops += 2.0 * M * p.N * p.K;
        bytes += (double)M * (p.N + p.K) * 4;
    }
    const int unit = 32;
    int S = tn_splits(M, tiles);
    const int mps = cdiv(cdiv(M, S), unit) * unit;
    S = cdiv(M, mps);
    for (int i = 0; i < count; ++i) {
        probs[i].part_off = off;
        off += (long)S * probs[i].N * probs[i].K;
        grp.p[i] = probs[i];
    }
    grp.count = count;
    grp.tiles_total = tiles;
    const bool extra = extra_part && extra_out && extra_G > 0 && extra_width > 0;
    if (extra) {
        grp.extra_part = extra_part; grp.extra_out = extra_out; grp.extra_G = extra_G; grp.extra_width = extra_width;
        if (extra_width > maxnk) maxnk = extra_width;
    }
    M3L_CHECK(ws_bytes >= (size_t)off * sizeof(float), "gemm_tn: workspace too small (%zu < %zu)", ws_bytes, (size_t)off * sizeof(float));
    {
        ProfScope prof("gemm_tn", M, tiles, S, flops, st, bytes + (double)off * 4.0);
        dim3 g1(8 * tiles * cdiv(S, 8));
        gemm_tn_glds_kernel<float><<<g1, 256, 65536, st>>>(grp, M, mps, S, partial_ws);
    }
    M3L_LAUNCH_CHECK();
    ProfScope prof2("gemm_tn_reduce", S, tiles, count, 0.0, st, (double)off * 4.0);
    reduce_splits_grouped_kernel<<<dim3(cdiv(maxnk, 256), count + (extra ? 1 : 0)), 256, 0, st>>>(grp, partial_ws, S, accumulate);
    M3L_LAUNCH_CHECK();
    (void)dtype;
    return 0;
}

int m3l_gemm_tn_grouped(int dtype, TnProblem* probs, int count, int M, float* partial_ws, size_t ws_bytes, int accumulate,
                        hipStream_t st, const TnExtra* extras, int extra_count) {
    if (m3l_gemm_init()) return 2;
    M3L_CHECK(dtype == 0 || dtype == 1, "gemm_tn: bad dtype %d", dtype);
    M3L_CHECK(count >= 1 && count <= M3L_TN_MAX_PROBLEMS && M > 0, "gemm_tn_grouped: count=%d M=%d", count, M);
    if (dtype == 1) return m3l_wgrad_bf16(probs, count, M, partial_ws, ws_bytes, accumulate, st, extras, extra_count);
    // f32: the 128 x 128 kernel takes 4 problems and one extra reduction per launch; chunks run back to back on the same slabs
    int e = 0;
    for (int c0 = 0; c0 < count; c0 += 4) {
        const int n = std::min(4, count - c0);
        const bool last = c0 + 4 >= count;
        const TnExtra* x = e < extra_count ? &extras[e++] : nullptr;
        if (tn_grouped_f32(probs + c0, n, M, partial_ws, ws_bytes, accumulate, st, x ? x->part : nullptr, x ? x->G : 0, x ? x->width : 0,
                           x ? x->out : nullptr))
            return 1;
        if (last) {
            while (e < extra_count) {                // more extras than chunks: plain row reductions
                if (m3l_reduce_rows(extras[e].part, extras[e].G, extras[e].width, extras[e].width, extras[e].out, accumulate, st)) return 1;
                ++e;
            }
        }
    }
    return 0;
}

int m3l_gemm_tn(int dtype, const void* Y, int ldy, const void* X, int ldx, int M, int N, int K, float* partial_ws,
                size_t ws_bytes, float* out, int ldo, int nvalid, int kvalid, int accumulate, hipStream_t st) {
    M3L_CHECK(M > 0 && N > 0 && K > 0, "gemm_tn: empty problem %d %d %d", M, N, K);
    TnProblem p;
    memset(&p, 0, sizeof(p));
    p.Y = Y; p.X = X; p.ldy = ldy; p.ldx = ldx; p.N = N; p.K = K; p.out = out; p.ldo = ldo; p.nvalid = nvalid; p.kvalid = kvalid;
    return m3l_gemm_tn_grouped(dtype, &p, 1, M, partial_ws, ws_bytes, accumulate, st);
}
