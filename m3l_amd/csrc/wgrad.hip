// Weight gradients dW[N, K] = Y^T X (sum over the M token rows) for bf16 operands on gfx950 — every nn.Linear.weight.grad of the
// MAE step (reference: autograd through vit_pytorch Attention.to_qkv / to_out, FeedForward.net[1] / net[4] and the patch / head
// Linears, models/pretrain_models.py:111,115,116,770,777 via loss.backward(), models/ppo_mae.py:263).
//
// Grouped: up to 16 problems that share M (the four weights of up to four transformer layers) in ONE launch + one reduce.
//   tile      256 (the wider of N / K) x 64 TBT (the narrower; TBT = 2, 3, 4 picked per launch), 8 waves, wave tile 64 x 32 TBT
//             -> 110 MAC per staged element at 256 x 192 (the 128 x 128 tile of round 1: 64), so the L2 -> LDS stream no longer
//             bounds the MFMA rate;
//   staging   both operands by bounds-checked LDS-DMA (buffer_load ... lds, rows >= M read as zero) issued from INLINE ASM: the
//             compiler does not see the LDS write, so it does not drain vmcnt(0) in front of the next ds_read (it does for the
//             builtin form: round 1's kernel waited for every prefetch before computing).  Four 32-row stages, three requested
//             ahead (~96 KiB in flight per CU: one stage ahead left every step waiting ~2 us for HBM); counted s_waitcnt vmcnt
//             (4 pieces per wave and stage) + one barrier per stage;
//   LDS image 64-column panels of [32 rows][128 B]; 16-byte chunk index XOR 2 ((row >> 1) & 3), applied on the per-lane SOURCE
//             address (the DMA destination is lane-linear) and on the read: the 8 rows a half-wave's ds_read_b64_tr_b16 touches
//             fall on 8 distinct 32-byte bank slots (conflict-free k-strided transpose reads);
//   split-M   S = #CU / tiles row ranges (>= 1 workgroup per CU, <= 1 round), each writing its fp32 tile in FRAGMENT order (one
//             1-KiB contiguous store per accumulator tile); the reduce kernel sums the S slabs in fixed order (deterministic,
//             float4 coalesced reads) and scatters to dW with the problem's strides and valid ranges.
#include <string.h>

#include <algorithm>

#include "common.cuh"
#include "kernels.h"

namespace {

typedef int int4v __attribute__((ext_vector_type(4)));

struct WgProb {
    const bf16* A;      // wide operand   [M][lda], a columns
    const bf16* B;      // narrow operand [M][ldb], b columns
    float* out;         // out[ia * so_a + ib * so_b]
    int lda, ldb, a, b, so_a, so_b, avalid, bvalid;
    int tile0, tiles_b;
    int gelu_a, gelu_b;  // GELU form (TnProblem::gelu_x) to apply to the staged wide / narrow operand (0 = none)
};
struct WgExtra {
    const float* part;  // [G][width] partial rows
    float* out;         // [width]
    int G, width;
};
struct WgGroup {
    WgProb p[M3L_TN_MAX_PROBLEMS];
    WgExtra ex[M3L_TN_MAX_EXTRAS];
    int count, tiles_total, extra_count;
};

// LDS-DMA of one 1-KiB piece (64 lanes x 16 B, lane-linear at lds_dst) through a buffer resource; M0 carries the LDS base and is
// restored (the compiler reserves it).  Not counted by the compiler: pair with an explicit s_waitcnt vmcnt + barrier.
template <bool NT = false>
__device__ __forceinline__ void dma16(int4v rsrc, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(rsrc), "s"(lds_dst)
                     : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(rsrc), "s"(lds_dst)
                     : "memory");
}
__device__ __forceinline__ int4v make_rsrc(const void* p, long bytes) {
    const unsigned long a = (unsigned long)p;
    int4v r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
    r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)(bytes > 2147483647L ? 2147483647L : bytes));
    r[3] = __builtin_amdgcn_readfirstlane(0x00020000);
    return r;
}

// fragment of a [64 k-rows][64 cols] panel (128-byte rows, swizzled): lane (i, g) <- element j = panel[(k0 + kacc(g, j))][c0 + i]
__device__ __forceinline__ Frag<bf16> ldtr(const char* panel, int k0, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int r0 = k0 + 4 * g + q;
    const int chunk = (c0 >> 3) + (p >> 1);
    const int x = ((chunk ^ (2 * ((r0 >> 1) & 3))) << 4) + (p & 1) * 8;
    typedef __attribute__((address_space(3))) bf16x4* lds_p;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(panel + r0 * 128 + x));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(panel + (r0 + 16) * 128 + x));
    Frag<bf16> f;
    f.v[0] = lo[0]; f.v[1] = lo[1]; f.v[2] = lo[2]; f.v[3] = lo[3];
    f.v[4] = hi[0]; f.v[5] = hi[1]; f.v[6] = hi[2]; f.v[7] = hi[3];
    return f;
}

#ifndef WG_ABL
#define WG_ABL 0            // diagnostic builds only (tools/t192_ablate.sh "..." wgrad WG_ABL): 1 no MFMAs, 2 no fragment reads either, 4 no DMA, 8 no slab stores
#endif
constexpr int WG_NST = 4;                    // ring stages of 32 rows: three stages requested ahead

// wait until at most n of this wave's vector-memory operations remain outstanding (n is wave-uniform, 0 .. 8: up to two stages of up to
// four pieces each may stay in flight)
__device__ __forceinline__ void wait_vm(int n) {
    switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
}

// Tile = 64 NA (wide operand) x 64 TBT (narrow operand) outputs, 2 NA waves of 64 x 32 TBT each.  <4, *>: 256-wide, 8 waves (2 per SIMD,
// up to 183 VGPRs).  <6, 3>: 384 x 192, 12 waves (3 per SIMD, 149 VGPRs): the ViT-Tiny layer (every weight has a 192 dimension) as
// 2 + 2 + 2 + 1 tiles instead of 3 + 3 + 3 + 1 — the narrow operand is staged once per TWO thirds of the wide one, 3648 instead of 4480
// staged columns per token row against 3072 unique ones — and panels of the wide operand that lie outside the problem (the second
// tile of the 576-wide qkv gradient holds 192 columns, the out-projection's 192 of 384) are neither staged nor multiplied: their waves
// only keep the barriers.
template <int NA, int TBT, bool NTL>
__global__ __launch_bounds__(128 * NA) void wgrad_kernel(WgGroup grp, int M, int rows_per_split, int nsplit, float* __restrict__ slabs, int per_xcd) {
    constexpr int TA = 64 * NA, TB = 64 * TBT, NB = TB / 32, NT = 4 * NB, NW = 2 * NA;
    constexpr int SLOT = (NA + TBT) * 4096, P = 4 * (NA + TBT), JMAX = (P + NW - 1) / NW;
    static_assert(JMAX <= 4, "at most four pieces per wave and stage (wait_vm)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    // Work list in split-major order (w = split * tiles + tile: neighbours read the same token rows), cut into 8 contiguous chunks, chunk x
    // handed to the blocks with id = x (mod 8), which share an XCD (and its L2) under the round-robin placement: balanced to one workgroup
    // for ANY split count (the earlier map gave split sp to XCD sp mod 8: 80 workgroups on one XCD and 60 on the others at 25 splits)
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int W = grp.tiles_total * nsplit, w = xcd * per_xcd + idx;
    if (w >= W || w >= (xcd + 1) * per_xcd) return;
    const int gtile = w % grp.tiles_total, sp = w / grp.tiles_total;
    int pi = 0;
    for (int i = 1; i < grp.count; ++i)
        if (gtile >= grp.p[i].tile0) pi = i;
    const WgProb& pb = grp.p[pi];
    const int tile = gtile - pb.tile0;
    const int a0 = (tile / pb.tiles_b) * TA, b0 = (tile % pb.tiles_b) * TB;
    const int m_beg = sp * rows_per_split;
    const int m_end = min(M, m_beg + rows_per_split);
    const int nst = (m_end - m_beg + 31) >> 5;               // 32-row stages

    const int4v rsA = make_rsrc(pb.A, (long)M * pb.lda * 2), rsB = make_rsrc(pb.B, (long)M * pb.ldb * 2);
    // A stage image = NA + TBT panels of [32 rows][128 B] (panels 0 .. NA - 1 = A, then B).  Piece q = (panel q >> 2, 8-row group q & 3);
    // wave v stages the pieces q = v + NW j that exist and whose panel lies inside the problem: `mine` of them per stage (wave-uniform),
    // the unit of its counted waits.
    unsigned voff[JMAX], pdst[JMAX];
    bool pa[JMAX], pv[JMAX];
    int mine = 0;
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const int q = wave + NW * j, panel = q >> 2, rg = q & 3;
        const bool is_a = panel < NA;
        const int c0 = is_a ? a0 + 64 * panel : b0 + 64 * (panel - NA), width = is_a ? pb.a : pb.b, ld = is_a ? pb.lda : pb.ldb;
        pa[j] = is_a;
        pv[j] = q < P && c0 < width;                          // wave-uniform
        mine += pv[j] ? 1 : 0;
        const int row = rg * 8 + (lane >> 3);
        const int csrc = (lane & 7) ^ (2 * ((row >> 1) & 3));
        const int col = min(c0 + csrc * 8, width - 8);        // chunks past the operand inside a partly valid panel: any valid chunk (never stored)
        voff[j] = (unsigned)(((long)(m_beg + row) * ld + col) * 2);
        pdst[j] = (unsigned)(panel * 4096 + rg * 1024);
    }
    const unsigned incA = (unsigned)(32 * pb.lda * 2), incB = (unsigned)(32 * pb.ldb * 2);
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
    auto issue = [&](int t) {
        if (WG_ABL & 4) return;
        const unsigned base = lds0 + (t % WG_NST) * SLOT;
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            if (!pv[j]) continue;
            if (pa[j]) dma16<NTL>(rsA, voff[j] + (unsigned)t * incA, __builtin_amdgcn_readfirstlane(base + pdst[j]));
            else dma16<NTL>(rsB, voff[j] + (unsigned)t * incB, __builtin_amdgcn_readfirstlane(base + pdst[j]));
        }
    };
    const bool work = a0 + 64 * wr < pb.a;                     // this wave's 64 rows of the output tile exist (wave-uniform)

    f32x4 acc[4][NB];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < NB; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < WG_NST - 1 && t < nst; ++t) issue(t);
    for (int t = 0; t < nst; ++t) {
        // this wave's pieces of stage t have landed once at most the pieces of the (up to two) younger stages remain (in-order retire)
        wait_vm(min(WG_NST - 2, nst - 1 - t) * mine);
        __builtin_amdgcn_s_barrier();                         // ... everyone's have; everyone is done reading stage t - 1
        asm volatile("" ::: "memory");
        if (t + WG_NST - 1 < nst) issue(t + WG_NST - 1);      // into the slot stage t - 1 occupied
        if (pb.gelu_a | pb.gelu_b) {                          // workgroup-uniform
            // the staged operand is the pre-activation u: h = GELU(u) in place, rounded as the forward rounded it, 8 values per thread
            // and pass (rows past M were staged as zeros: GELU(0) = 0).  The kernel waits on HBM most of its time: the pass hides there.
            char* base = smem + (t % WG_NST) * SLOT + (pb.gelu_a ? 0 : NA * 4096);
            const int bytes = pb.gelu_a ? NA * 4096 : TBT * 4096, form = pb.gelu_a | pb.gelu_b;
            for (int off = tid * 16; off < bytes; off += 64 * NW * 16) {
                bf16x8 v = *reinterpret_cast<bf16x8*>(base + off);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (bf16)(form == 1 ? gelu_f((float)v[j]) : gelu_fast((float)v[j]));
                *reinterpret_cast<bf16x8*>(base + off) = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if ((WG_ABL & 2) || !work) continue;
        const char* As = smem + (t % WG_NST) * SLOT + wr * 4096;
        const char* Bs = smem + (t % WG_NST) * SLOT + NA * 4096;
        Frag<bf16> fa[4], fb[NB];
#pragma unroll
        for (int x = 0; x < 4; ++x) fa[x] = ldtr(As, 0, x * 16, lane);
#pragma unroll
        for (int y = 0; y < NB; ++y) {
            const int col = wc * (TB / 2) + y * 16;
            fb[y] = ldtr(Bs + (col >> 6) * 4096, 0, col & 63, lane);
        }
        if (WG_ABL & 1) {
#pragma unroll
            for (int x = 0; x < 4; ++x) asm volatile("" ::"v"(fa[x].v));
#pragma unroll
            for (int y = 0; y < NB; ++y) asm volatile("" ::"v"(fb[y].v));
            continue;
        }
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < NB; ++y) acc[x][y] = mma16(fa[x], fb[y], acc[x][y]);
    }
    if (WG_ABL & 8) return;
    // fragment-order slab: [wave][tile x * NB + y][lane] float4 (rows 4 g .. 4 g + 3 of the 16 x 16 tile, column li).  16 x 16 tiles that lie
    // entirely outside the problem (a 192 x 32 conv weight in a 256 x 128 tile: 104 of its 128 tiles) are neither written nor read back
    float* S = slabs + (long)w * (long)(TA * TB);
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        if (a0 + wr * 64 + x * 16 >= pb.avalid) continue;       // wave-uniform
#pragma unroll
        for (int y = 0; y < NB; ++y) {
            if (b0 + wc * (TB / 2) + y * 16 >= pb.bvalid) continue;
            *reinterpret_cast<f32x4*>(S + (((wave * NT + x * NB + y) * 64 + lane) << 2)) = acc[x][y];
        }
    }
}

template <int NA, int TBT>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgGroup grp, const float* __restrict__ slabs, int S, int accumulate) {
    constexpr int TA = 64 * NA, TB = 64 * TBT, NB = TB / 32, NT = 4 * NB;
    const int f = blockIdx.x * 256 + threadIdx.x;
    const int gt = blockIdx.y;
    if (gt >= grp.tiles_total) {
        // extra column reductions (bias-gradient partial rows [G][width]): 32 columns x 8 row slices per block, 4 independent
        // accumulators per thread, slices added in a fixed order through LDS.  (One thread per column looping over G = 256 rows was a
        // 128-deep chain of dependent L2 round trips: it alone set this kernel's duration, ~50 us for 12 us worth of slab traffic.)
        __shared__ float red[8][33];
        const WgExtra& e = grp.ex[gt - grp.tiles_total];
        const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
        const int j = blockIdx.x * 32 + cl;
        float acc = 0.f;
        if (j < e.width) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int g = sl;
            for (; g + 24 < e.G; g += 32) {
                s0 += e.part[(long)g * e.width + j];
                s1 += e.part[(long)(g + 8) * e.width + j];
                s2 += e.part[(long)(g + 16) * e.width + j];
                s3 += e.part[(long)(g + 24) * e.width + j];
            }
            for (; g < e.G; g += 8) s0 += e.part[(long)g * e.width + j];
            acc = (s0 + s1) + (s2 + s3);
        }
        red[sl][cl] = acc;
        __syncthreads();
        if (sl == 0 && j < e.width) {
            float t = red[0][cl];
#pragma unroll
            for (int i = 1; i < 8; ++i) t += red[i][cl];
            e.out[j] = accumulate ? e.out[j] + t : t;
        }
        return;
    }
    if (f >= TA / 4 * TB) return;
    int pi = 0;
    for (int i = 1; i < grp.count; ++i)
        if (gt >= grp.p[i].tile0) pi = i;
    const WgProb& pb = grp.p[pi];
    const int tile = gt - pb.tile0;
    const int a0 = (tile / pb.tiles_b) * TA, b0 = (tile % pb.tiles_b) * TB;
    const long stride = (long)grp.tiles_total * (TA * TB);
    const float* P = slabs + (long)gt * (TA * TB) + ((long)f << 2);
    {   // this thread's 16 x 16 tile lies outside the problem: nothing was written, nothing to add (checked BEFORE the S slab reads)
        const int wt0 = f >> 6, t0 = wt0 % NT, w0 = wt0 / NT;
        if (a0 + (w0 >> 1) * 64 + (t0 / NB) * 16 >= pb.avalid || b0 + (w0 & 1) * (TB / 2) + (t0 % NB) * 16 >= pb.bvalid) return;
    }
    // 8 independent 16-byte loads in flight per thread (the loop over a run-time S with two accumulators issued them in pairs: 13
    // dependent round trips for S = 25); the sum order is fixed: ((s0 + s1) + (s2 + s3)) + ... over i mod 8, then the tail
    f32x4 a8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a8[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    int i = 0;
    for (; i + 8 <= S; i += 8) {
        // requested together, then added: written as `a8[k] += load` the compiler emitted load -> s_waitcnt vmcnt(0) -> add eight times in a
        // row (tools/asm_roundtrips.py: 23 dependent round trips per thread for S = 23), i.e. this kernel WAS the chain its comment describes
        f32x4 t8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t8[k] = *reinterpret_cast<const f32x4*>(P + (long)(i + k) * stride);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < 8; ++k) a8[k] += t8[k];
    }
    // the tail in one batch as well: a slab index past the end re-reads the last slab and adds nothing (each guarded load was a basic
    // block with its own wait: up to seven dependent round trips for S = 23)
    {
        f32x4 t8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t8[k] = *reinterpret_cast<const f32x4*>(P + (long)min(i + k, S - 1) * stride);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (i + k < S) a8[k] += t8[k];
    }
    f32x4 s0 = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    const int lane = f & 63, wt = f >> 6, t = wt % NT, w = wt / NT, x = t / NB, y = t % NB;
    const int ia = a0 + (w >> 1) * 64 + x * 16 + 4 * (lane >> 4);
    const int ib = b0 + (w & 1) * (TB / 2) + y * 16 + (lane & 15);
    if (ib >= pb.bvalid) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (ia + r >= pb.avalid) break;
        float* o = pb.out + (long)(ia + r) * pb.so_a + (long)ib * pb.so_b;
        *o = accumulate ? (*o + s0[r]) : s0[r];
    }
}

int g_ncu = 0;
int cu_count() {
    if (!g_ncu) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        g_ncu = n;
    }
    return g_ncu;
}

struct WgPlanHost {
    int na, tbt, tiles_total, S, rows_per_split;
    size_t ws_bytes;
};
// cost of a tiling per token row: the columns its LDS-DMA stream stages (panels outside a problem are not staged) + its MFMA area in
// units of 256 multiply-accumulates (rows of the wide operand outside a problem are not multiplied, columns of the narrow one are)
long staged_cols(const TnProblem* probs, int count, int na, int tbt) {
    long cost = 0;
    const int TA = 64 * na, TB = 64 * tbt;
    for (int i = 0; i < count; ++i) {
        const int a = std::max(probs[i].N, probs[i].K), b = std::min(probs[i].N, probs[i].K);
        const int ta = cdiv(a, TA), tb = cdiv(b, TB);
        cost += (long)tb * cdiv(a, 64) * 64 + (long)ta * cdiv(b, 64) * 64;       // every tile stages its valid A panels and its valid B panels
        cost += (long)cdiv(a, 64) * 64 * tb * TB / 256;
    }
    return cost;
}
// wide / narrow orientation and tiling of a problem list (the same decisions for the size query and the launch)
WgPlanHost plan_of(int M, const TnProblem* probs, int count, WgGroup* grp) {
    WgPlanHost pl;
    // tile shape: the lowest cost; ties -> the smaller tile (fewer idle waves).  <6, 4> does not exist (176 VGPRs at 12 waves).
    static const int force_na = getenv("M3L_WGRAD_NA") ? atoi(getenv("M3L_WGRAD_NA")) : 0;      // 4 / 6: A/B of the tile heights
    static const int force_tbt = getenv("M3L_WGRAD_TBT") ? atoi(getenv("M3L_WGRAD_TBT")) : 0;
    long best = -1;
    pl.na = 4;
    pl.tbt = 3;
    for (int na = 4; na <= 6; na += 2) {
        if (force_na && na != force_na) continue;
        for (int tbt = 2; tbt <= (na == 6 ? 3 : 4); ++tbt) {
            if (force_tbt && tbt != force_tbt && !(na == 6 && force_tbt == 4)) continue;
            const long cost = staged_cols(probs, count, na, tbt);
            if (best < 0 || cost < best) {
                best = cost;
                pl.na = na;
                pl.tbt = tbt;
            }
        }
    }
    const int TA = 64 * pl.na, TB = 64 * pl.tbt;
    int tiles = 0;
    for (int i = 0; i < count; ++i) {
        const TnProblem& p = probs[i];
        const bool y_wide = p.N >= p.K;
        const int a = y_wide ? p.N : p.K, b = y_wide ? p.K : p.N;
        const int tb = cdiv(b, TB);
        if (grp) {
            WgProb& w = grp->p[i];
            w.A = (const bf16*)(y_wide ? p.Y : p.X); w.B = (const bf16*)(y_wide ? p.X : p.Y);
            w.lda = y_wide ? p.ldy : p.ldx; w.ldb = y_wide ? p.ldx : p.ldy;
            w.a = a; w.b = b;
            w.out = p.out;
            w.so_a = y_wide ? p.ldo : 1; w.so_b = y_wide ? 1 : p.ldo;
            w.avalid = y_wide ? p.nvalid : p.kvalid; w.bvalid = y_wide ? p.kvalid : p.nvalid;
            w.tile0 = tiles; w.tiles_b = tb;
            w.gelu_a = y_wide ? 0 : p.gelu_x;        // X is the wide operand when K > N
            w.gelu_b = y_wide ? p.gelu_x : 0;
        }
        tiles += cdiv(a, TA) * tb;
    }
    pl.tiles_total = tiles;
    // splits over M: M3L_WGRAD_WAVES workgroups per CU in all.  1 (the default since round 4) = one wave of workgroups, each as long as the
    // launch: half the partial slabs of 2.  (Round 2-3 ran 2: with the split -> XCD map of that kernel one wave left XCDs 0-3 with twice
    // the workgroups of XCDs 4-7; with the balanced work list one wave is 126 us per grouped launch in-step against 160 for two, 105
    // against 113 stand-alone, and the step gains 1 %.)
    static const int waves = getenv("M3L_WGRAD_WAVES") ? std::max(1, atoi(getenv("M3L_WGRAD_WAVES"))) : 1;
    int S = std::max(1, waves * cu_count() / std::max(1, tiles));
    S = std::min(S, std::max(1, M / 256));                     // at least 256 rows per split
    pl.rows_per_split = cdiv(cdiv(M, S), 64) * 64;
    pl.S = cdiv(M, pl.rows_per_split);
    pl.ws_bytes = (size_t)pl.S * tiles * TA * TB * sizeof(float);
    return pl;
}

template <int NA, int TBT>
int launch_wgrad(const WgGroup& grp, const WgPlanHost& pl, int M, float* ws, int accumulate, int maxw, int extra_count, hipStream_t st, int count,
                 double flops, double bytes, int nt) {
    constexpr int TA = 64 * NA, TB = 64 * TBT;
    constexpr size_t lds = (size_t)WG_NST * (NA + TBT) * 4096;
    static bool inited = false;
    if (!inited) {
        M3L_HIP(hipFuncSetAttribute((const void*)wgrad_kernel<NA, TBT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        M3L_HIP(hipFuncSetAttribute((const void*)wgrad_kernel<NA, TBT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        inited = true;
    }
    {
        ProfScope prof(count >= 3 ? "wgrad" : "wgrad_small", M, pl.tiles_total, pl.S, flops, st, bytes);   // grouped layer launches vs single Linears
        const int per_xcd = cdiv((long)pl.tiles_total * pl.S, 8);
        const dim3 g1(8 * per_xcd);
        if (nt) wgrad_kernel<NA, TBT, true><<<g1, 128 * NA, lds, st>>>(grp, M, pl.rows_per_split, pl.S, ws, per_xcd);
        else wgrad_kernel<NA, TBT, false><<<g1, 128 * NA, lds, st>>>(grp, M, pl.rows_per_split, pl.S, ws, per_xcd);
    }
    M3L_LAUNCH_CHECK();
    ProfScope prof2("wgrad_reduce", pl.S, pl.tiles_total, count, 0.0, st, (double)pl.ws_bytes);
    const dim3 g2(std::max(cdiv(TA / 4 * TB, 256), cdiv(maxw, 32)), pl.tiles_total + extra_count);
    wgrad_reduce_kernel<NA, TBT><<<g2, 256, 0, st>>>(grp, ws, pl.S, accumulate);
    M3L_LAUNCH_CHECK();
    return 0;
}

}  // namespace

size_t m3l_wgrad_ws_bytes(int M, const TnProblem* probs, int count) { return plan_of(M, probs, count, nullptr).ws_bytes; }

// layers per grouped launch: as many as the problem table holds (4).  Every launch writes and re-reads one slab per workgroup (one wave of
// workgroups = a fixed ~75 MB at 384 x 192 tiles) whatever it multiplies, so more layers per launch means less slab traffic per layer;
// measured at cfg 2 (decoder: 2 -> 4 layers per launch): same step time, grouped launch 0.44 -> 0.50 of the HBM roof in-step, -150 MB per
// step; cfg 4 / M3L default +1-2 %.  (More layers per launch only lengthen a split's row range; the caller bounds it by the stack's depth.)
int m3l_wgrad_layers_per_launch(int M, const TnProblem* layer_probs, int count) {
    (void)M; (void)layer_probs; (void)count;
    return M3L_TN_MAX_PROBLEMS / 4;
}

int m3l_wgrad_bf16(TnProblem* probs, int count, int M, float* ws, size_t ws_bytes, int accumulate, hipStream_t st, const TnExtra* extras,
                   int extra_count) {
    M3L_CHECK(count >= 1 && count <= M3L_TN_MAX_PROBLEMS && M > 0, "wgrad: count=%d M=%d", count, M);
    M3L_CHECK(extra_count >= 0 && extra_count <= M3L_TN_MAX_EXTRAS, "wgrad: %d extra reductions", extra_count);
    WgGroup grp;
    memset(&grp, 0, sizeof(grp));
    double flops = 0, bytes = 0;
    for (int i = 0; i < count; ++i) {
        const TnProblem& p = probs[i];
        M3L_CHECK(p.N >= 8 && p.K >= 8 && p.K % 8 == 0 && p.N % 8 == 0 && p.ldy % 8 == 0 && p.ldx % 8 == 0,
                  "wgrad: N, K, ldy, ldx must be positive multiples of 8 (N=%d K=%d ldy=%d ldx=%d)", p.N, p.K, p.ldy, p.ldx);
        M3L_CHECK((long)M * p.ldy * 2 < 2147483647L && (long)M * p.ldx * 2 < 2147483647L, "wgrad: operand larger than 2 GiB");
        flops += 2.0 * M * p.N * p.K;
        bytes += 2.0 * M * (p.N + p.K) + 4.0 * p.N * p.K;          // algorithmic: both operands once, dW once
    }
    const WgPlanHost pl = plan_of(M, probs, count, &grp);
    grp.count = count;
    grp.tiles_total = pl.tiles_total;
    grp.extra_count = extra_count;
    int maxw = 0;
    for (int i = 0; i < extra_count; ++i) {
        grp.ex[i] = WgExtra{extras[i].part, extras[i].out, extras[i].G, extras[i].width};
        maxw = std::max(maxw, extras[i].width);
    }
    M3L_CHECK(ws_bytes >= pl.ws_bytes, "wgrad: workspace too small (%zu < %zu)", ws_bytes, pl.ws_bytes);
    static const int wg_nt = getenv("M3L_WGRAD_NT") ? atoi(getenv("M3L_WGRAD_NT")) : 0;     // opt-in: non-temporal operand loads (stand-alone 136 -> 126 us, nothing end to end)
    if (pl.na == 6) {
        if (pl.tbt == 2) return launch_wgrad<6, 2>(grp, pl, M, ws, accumulate, maxw, extra_count, st, count, flops, bytes, wg_nt);
        return launch_wgrad<6, 3>(grp, pl, M, ws, accumulate, maxw, extra_count, st, count, flops, bytes, wg_nt);
    }
    if (pl.tbt == 2) return launch_wgrad<4, 2>(grp, pl, M, ws, accumulate, maxw, extra_count, st, count, flops, bytes, wg_nt);
    if (pl.tbt == 3) return launch_wgrad<4, 3>(grp, pl, M, ws, accumulate, maxw, extra_count, st, count, flops, bytes, wg_nt);
    return launch_wgrad<4, 4>(grp, pl, M, ws, accumulate, maxw, extra_count, st, count, flops, bytes, wg_nt);
}
