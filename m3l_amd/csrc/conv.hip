// Direct (implicit-GEMM) convolutions for the EarlyCNN stem of the MAE (reference: models/pretrain_models.py:37-56 — three
// Conv2d + ReLU, 4x4 / stride 2 / pad 1 (the tactile stem's third one 3x3 / stride 1 / pad 1), called at :180-191 when
// early_conv_masking=True, the reference's default flag) — bf16 operands, fp32 accumulation.
//
// Round 1-2 ran every convolution as im2col + GEMM: the column matrix is 4x the layer input (201 MB for conv1 at the reference's
// default B = 512, frame_stack 4), written by one kernel, read by the forward GEMM, read again by the weight-gradient GEMM, and the
// input gradient went through a second column matrix + col2im: 27 % of that architecture's kernel time moved bytes that exist only
// because of the lowering.  Here a workgroup owns an 8 x 8 tile of output pixels of one image, stages the input patch it touches
// ((8 S + KH - S)^2 pixels x Ci channels, NHWC, bf16) in LDS once, and every MFMA operand is a 16-byte read of 8 consecutive channels
// of one tap — the "column matrix" never exists:
//   forward       out^T[co][pixel] = sum_k Wf[co][k] * patch[pixel @ tap(k)][ci(k)],  k = (tap, ci);  + bias, ReLU, NHWC bf16 out
//   weight grad   dW[co][tap][ci]  = sum_pixel dY[pixel][co] * patch[pixel @ tap][ci]  (both operands k-strided: ds_read_b64_tr_b16);
//                 partial sums per group of tiles -> slabs in the Conv2d weight layout -> fixed-order reduce (deterministic)
//   input grad    dX^T[ci][pixel]  = sum_(tap, co) Wd[ci][(tap, co)] * dYpatch[pixel @ tap][co], the taps that reach an input pixel
//                 (stride 2: 2 x 2 of the 4 x 4, by the pixel's parity class — one class per wave); * [act > 0], NHWC bf16 out
// Products are computed transposed (weights = A operand) so that a lane ends with 4 consecutive channels of ONE pixel: 8-byte stores.
// Shapes: Ci padded to 16 in LDS (12 input channels at frame_stack 4), Co in {8 .. 128}; other widths keep the im2col path.
#include <hip/hip_runtime.h>
#include <string.h>

#include <algorithm>

#include "common.cuh"
#include "kernels.h"

namespace {

constexpr int CT = 8;                         // output (forward / weight grad) or input (stride-1 input grad) tile edge, pixels

struct ConvGeo {
    int B, Ci, Cip, H, W, Co, OH, OW, T, Kp;  // T = KH * KH taps, Kp = T * Cip rounded up to 32
    int tiles_y, tiles_x;
};

__device__ __forceinline__ int pix_bytes(int Cip) { return Cip * 2 + 16; }      // padded pixel stride in the LDS patch

// stage the input patch of output tile (ty, tx) of image b: rows ih0 .. ih0 + PH - 1, cols iw0 .. (zeros outside the image and in the
// channel padding).  Source: NCHW f32 frames (the first layer; `Bsrc` samples per source, sources concatenated on batch) or the NHWC
// bf16 activation of the previous layer.
template <int PH>
__device__ __forceinline__ void load_patch(char* patch, const ConvSrc& cs, int Bsrc, int b, int Ci, int Cip, int H, int W, int ih0, int iw0,
                                           int tid, int nthreads) {
    const int PIX = pix_bytes(Cip);
    if (cs.nchw) {
        const float* src = reinterpret_cast<const float*>(cs.src[b / Bsrc]) + (long)(b % Bsrc) * Ci * H * W;
        for (int id = tid; id < Cip * PH * PH; id += nthreads) {
            const int c = id % PH, r = (id / PH) % PH, ci = id / (PH * PH);
            const int ih = ih0 + r, iw = iw0 + c;
            float v = 0.f;
            if (ci < Ci && ih >= 0 && ih < H && iw >= 0 && iw < W) v = src[((long)ci * H + ih) * W + iw];
            *reinterpret_cast<bf16*>(patch + (r * PH + c) * PIX + ci * 2) = (bf16)v;
        }
    } else {
        const bf16* src = reinterpret_cast<const bf16*>(cs.src[0]) + (long)b * H * W * Ci;
        const int chunks = Cip >> 3;
        for (int id = tid; id < PH * PH * chunks; id += nthreads) {
            const int cc = id % chunks, c = (id / chunks) % PH, r = id / (chunks * PH);
            const int ih = ih0 + r, iw = iw0 + c;
            uint4 v = uint4{0u, 0u, 0u, 0u};
            if (8 * cc < Ci && ih >= 0 && ih < H && iw >= 0 && iw < W) v = *reinterpret_cast<const uint4*>(src + ((long)ih * W + iw) * Ci + 8 * cc);
            *reinterpret_cast<uint4*>(patch + (r * PH + c) * PIX + cc * 16) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// forward: out[b, oh, ow, co] = relu(bias[co] + sum_(kh, kw, ci) in[b, oh S - P + kh, ow S - P + kw, ci] W[co, ci, kh, kw])
// Wf: [16 MT][Kp] bf16, k = tap * Cip + ci (zero rows / columns in the padding).  One wave per 16-pixel row pair of the tile.
template <int KH, int S, int MT>
__global__ __launch_bounds__(256) void conv_fwd_kernel(ConvSrc cs, int Bsrc, ConvGeo g, int P, const bf16* __restrict__ Wf,
                                                       const float* __restrict__ bias, bf16* __restrict__ out) {
    constexpr int PH = CT * S + KH - S;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, gq = lane >> 4;
    const int tile = blockIdx.x % (g.tiles_y * g.tiles_x), b = blockIdx.x / (g.tiles_y * g.tiles_x);
    const int oh0 = (tile / g.tiles_x) * CT, ow0 = (tile % g.tiles_x) * CT;
    load_patch<PH>(smem, cs, Bsrc, b, g.Ci, g.Cip, g.H, g.W, oh0 * S - P, ow0 * S - P, tid, 256);
    __syncthreads();
    const int PIX = pix_bytes(g.Cip);
    const int py = 2 * wave + (li >> 3), px = li & 7;         // this lane's output pixel (column of every accumulator tile)
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < (g.Kp >> 5); ++ks) {
        const int k0 = 32 * ks + 8 * gq, tap = k0 / g.Cip, ci0 = k0 - tap * g.Cip;
        Frag<bf16> fb;
        uint4 z = uint4{0u, 0u, 0u, 0u};
        if (tap < g.T) z = *reinterpret_cast<const uint4*>(smem + ((py * S + tap / KH) * PH + px * S + tap % KH) * PIX + ci0 * 2);
        fb.v = __builtin_bit_cast(bf16x8, z);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            Frag<bf16> fa;
            fa.v = *reinterpret_cast<const bf16x8*>(Wf + (long)(16 * m + li) * g.Kp + k0);
            acc[m] = mma16(fa, fb, acc[m]);
        }
    }
    const int oh = oh0 + py, ow = ow0 + px;
    if (oh < g.OH && ow < g.OW) {
        bf16* o = out + (((long)b * g.OH + oh) * g.OW + ow) * g.Co;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int co = 16 * m + 4 * gq;
            if (co < g.Co) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + co);
                bf16x4 pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[r] = (bf16)fmaxf(acc[m][r] + bv[r], 0.f);
                *reinterpret_cast<bf16x4*>(o + co) = pk;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// weight gradient: slab[group][co][ci][tap] (the Conv2d weight layout) = sum over the group's tiles of dY^T X_tap.
// Workgroup (group, tap group): wave w owns taps tg TPG + w, + 4, ... (TW of them); MT x NT accumulator tiles (co x ci) per tap.
template <int KH, int S, int MT, int NT, int TW>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(ConvSrc cs, int Bsrc, ConvGeo g, int P, const bf16* __restrict__ dY, int tiles_per_wg,
                                                         int TPG, float* __restrict__ slab) {
    constexpr int PH = CT * S + KH - S;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int PIX = pix_bytes(g.Cip);
    char* patch = smem;
    const int DYLD = 16 * MT + 8;                             // dY tile [64 pixels][16 MT + 8] bf16
    bf16* dys = reinterpret_cast<bf16*>(smem + ((PH * PH * PIX + 15) & ~15));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, gq = lane >> 4;
    const int tpi = g.tiles_y * g.tiles_x, total = g.B * tpi;
    const int t_beg = blockIdx.x * tiles_per_wg, t_end = min(total, t_beg + tiles_per_wg);
    const int tap_base = blockIdx.y * TPG;
    f32x4 acc[TW][MT][NT];
#pragma unroll
    for (int a = 0; a < TW; ++a)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[a][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t = t_beg; t < t_end; ++t) {
        const int b = t / tpi, tile = t % tpi;
        const int oh0 = (tile / g.tiles_x) * CT, ow0 = (tile % g.tiles_x) * CT;
        __syncthreads();                                      // everyone is done with the previous tile's images
        load_patch<PH>(patch, cs, Bsrc, b, g.Ci, g.Cip, g.H, g.W, oh0 * S - P, ow0 * S - P, tid, 256);
        for (int id = tid; id < 64 * 2 * MT; id += 256) {     // dY rows of the tile's 64 pixels, 8 channels per piece (zeros outside)
            const int cc = id % (2 * MT), p = id / (2 * MT);
            const int oh = oh0 + (p >> 3), ow = ow0 + (p & 7);
            uint4 v = uint4{0u, 0u, 0u, 0u};
            if (oh < g.OH && ow < g.OW && 8 * cc < g.Co) v = *reinterpret_cast<const uint4*>(dY + (((long)b * g.OH + oh) * g.OW + ow) * g.Co + 8 * cc);
            *reinterpret_cast<uint4*>(dys + p * DYLD + 8 * cc) = v;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {                      // k = pixel: 2 steps of 32 pixels (4 tile rows each)
            Frag<bf16> fa[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) fa[m] = load_ks<KMAP_NAT>(dys, DYLD, 32 * ks, 16 * m, lane);       // A[co][pixel]
#pragma unroll
            for (int a = 0; a < TW; ++a) {
                const int tap = tap_base + wave + 4 * a;      // wave-uniform
                if (4 * a + wave >= TPG || tap >= g.T) continue;
                const int kh = tap / KH, kw = tap % KH;
                // B[pixel][ci]: pixel 32 ks + 8 g + j sits in tile row 4 ks + g, column j -> patch row (4 ks + g) S + kh, column j S + kw
                const int q = li >> 2, pp = li & 3;
                const char* rowp = patch + (((4 * ks + gq) * S + kh) * PH + kw) * PIX;
                typedef __attribute__((address_space(3))) bf16x4* lds_p;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(rowp + (q * S) * PIX + (16 * n + 4 * pp) * 2));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(rowp + ((q + 4) * S) * PIX + (16 * n + 4 * pp) * 2));
                    Frag<bf16> fb;
                    fb.v[0] = lo[0]; fb.v[1] = lo[1]; fb.v[2] = lo[2]; fb.v[3] = lo[3];
                    fb.v[4] = hi[0]; fb.v[5] = hi[1]; fb.v[6] = hi[2]; fb.v[7] = hi[3];
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[a][m][n] = mma16(fa[m], fb, acc[a][m][n]);
                }
            }
        }
    }
    // slab of this group in the Conv2d weight layout [co][ci][tap]: lane holds co = 16 m + 4 g + r, ci = 16 n + li
    float* S0 = slab + (long)blockIdx.x * g.Co * g.Ci * g.T;
#pragma unroll
    for (int a = 0; a < TW; ++a) {
        const int tap = tap_base + wave + 4 * a;
        if (4 * a + wave >= TPG || tap >= g.T) continue;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int ci = 16 * n + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = 16 * m + 4 * gq + r;
                    if (co < g.Co && ci < g.Ci) S0[((long)co * g.Ci + ci) * g.T + tap] = acc[a][m][n][r];
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// input gradient: dX[b, ih, iw, ci] = [act > 0] * sum over the taps (kh, kw) with (ih + P - kh) % S == 0 (same for w) and the output pixel
// inside the layer's output, sum_co dY[b, (ih + P - kh) / S, (iw + P - kw) / S, co] W[co, ci, kh, kw].
// Wd: [class][16 NTI][KC] bf16, KC = taps-per-class * Co, k = (tap index within the class, co).
//   S == 2 (4x4, pad 1): input tile 16 x 16; class (a, c) = (ih & 1, iw & 1) has the 2 x 2 taps kh in {1 - a, 3 - a}, kw in {1 - c, 3 - c};
//                        wave = class, 4 pixel tiles of 16 (two rows of 8 class pixels each).
//   S == 1 (3x3, pad 1): input tile 8 x 8, one class with all 9 taps; wave = pixel tile.
template <int KH, int S, int NTI>
__global__ __launch_bounds__(256) void conv_dgrad_kernel(const bf16* __restrict__ dY, ConvGeo g, const bf16* __restrict__ Wd, int KC,
                                                         const bf16* __restrict__ act, bf16* __restrict__ dX) {
    constexpr int IT = S == 2 ? 16 : 8;                       // input tile edge
    constexpr int PD = S == 2 ? 10 : 10;                      // dY patch edge: rows oh_base .. oh_base + 9
    constexpr int NPT = S == 2 ? 4 : 1;                       // pixel tiles per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int PIX = pix_bytes(g.Co);                          // patch pixels hold Co channels here (Co % 8 == 0)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, gq = lane >> 4;
    const int tx_n = (g.W + IT - 1) / IT, ty_n = (g.H + IT - 1) / IT;
    const int tile = blockIdx.x % (tx_n * ty_n), b = blockIdx.x / (tx_n * ty_n);
    const int ih0 = (tile / tx_n) * IT, iw0 = (tile % tx_n) * IT;
    // dY rows oh_b .. oh_b + 9: S == 2: oh_b = ih0 / 2 - 1;  S == 1: oh_b = ih0 - 1
    const int oh_b = ih0 / S - 1, ow_b = iw0 / S - 1;
    {
        const bf16* src = dY + (long)b * g.OH * g.OW * g.Co;
        const int chunks = g.Co >> 3;
        for (int id = tid; id < PD * PD * chunks; id += 256) {
            const int cc = id % chunks, c = (id / chunks) % PD, r = id / (chunks * PD);
            const int oh = oh_b + r, ow = ow_b + c;
            uint4 v = uint4{0u, 0u, 0u, 0u};
            if (oh >= 0 && oh < g.OH && ow >= 0 && ow < g.OW) v = *reinterpret_cast<const uint4*>(src + ((long)oh * g.OW + ow) * g.Co + 8 * cc);
            *reinterpret_cast<uint4*>(smem + (r * PD + c) * PIX + cc * 16) = v;
        }
    }
    __syncthreads();
    const int cls = S == 2 ? wave : 0;
    const int ca = cls >> 1, cc2 = cls & 1;                   // parity of (ih, iw) in this class (S == 2)
    const bf16* W = Wd + (long)cls * 16 * NTI * KC;
    f32x4 acc[NTI][NPT];
#pragma unroll
    for (int n = 0; n < NTI; ++n)
#pragma unroll
        for (int p = 0; p < NPT; ++p) acc[n][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < (KC >> 5); ++ks) {
        const int k0 = 32 * ks + 8 * gq, ti = k0 / g.Co, co0 = k0 - ti * g.Co;
        // tap ti of the class: S == 2: (dh, dw) = (ti >> 1, ti & 1): kh = (1 - a) + 2 dh  ->  oh = y + a - dh  (y = class-pixel row), patch row = oh - oh_b
        //                      S == 1: kh = ti / 3, kw = ti % 3 -> oh = ih + 1 - kh
        Frag<bf16> fa[NTI];
#pragma unroll
        for (int n = 0; n < NTI; ++n) fa[n].v = *reinterpret_cast<const bf16x8*>(W + (long)(16 * n + li) * KC + k0);
#pragma unroll
        for (int p = 0; p < NPT; ++p) {
            int pr, pc;                                        // patch row / column of this lane's pixel for this tap
            if (S == 2) {
                const int y = 2 * p + (li >> 3), x = li & 7;  // class pixel (y, x): ih = ih0 + 2 y + a
                pr = y + ca - (ti >> 1) + 1;                   // oh - oh_b with oh_b = ih0 / 2 - 1
                pc = x + cc2 - (ti & 1) + 1;
            } else {
                const int y = 2 * wave + (li >> 3), x = li & 7;
                pr = y + 1 - ti / 3 + 1;                       // oh - oh_b = (ih0 + y + 1 - kh) - (ih0 - 1)
                pc = x + 1 - ti % 3 + 1;
            }
            Frag<bf16> fb;
            fb.v = *reinterpret_cast<const bf16x8*>(smem + (pr * PD + pc) * PIX + co0 * 2);
#pragma unroll
            for (int n = 0; n < NTI; ++n) acc[n][p] = mma16(fa[n], fb, acc[n][p]);
        }
    }
#pragma unroll
    for (int p = 0; p < NPT; ++p) {
        int ih, iw;
        if (S == 2) { ih = ih0 + 2 * (2 * p + (li >> 3)) + ca; iw = iw0 + 2 * (li & 7) + cc2; }
        else { ih = ih0 + 2 * wave + (li >> 3); iw = iw0 + (li & 7); }
        if (ih >= g.H || iw >= g.W) continue;
        const long o = (((long)b * g.H + ih) * g.W + iw) * g.Ci;
#pragma unroll
        for (int n = 0; n < NTI; ++n) {
            const int ci = 16 * n + 4 * gq;
            if (ci >= g.Ci) continue;
            const bf16x4 av = *reinterpret_cast<const bf16x4*>(act + o + ci);
            bf16x4 pk;
#pragma unroll
            for (int r = 0; r < 4; ++r) pk[r] = (float)av[r] > 0.f ? (bf16)acc[n][p][r] : (bf16)0.f;
            *reinterpret_cast<bf16x4*>(dX + o + ci) = pk;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// weight copies: W fp32 [Co][Ci][KH][KH] -> Wf bf16 [16 MT][Kp] (k = tap Cip + ci) and, when wanted, Wd bf16 [classes][16 NTI][KC]
__global__ void conv_prep_kernel(const float* __restrict__ W, int Co, int Ci, int Cip, int KH, int S, int Kp, int MT, bf16* __restrict__ Wf,
                                 int NTI, int KC, bf16* __restrict__ Wd) {
    const int T = KH * KH;
    const long nf = (long)16 * MT * Kp;
    const int classes = S == 2 ? 4 : 1;
    const long nd = Wd ? (long)classes * 16 * NTI * KC : 0;
    for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < nf + nd; id += (long)gridDim.x * blockDim.x) {
        if (id < nf) {
            const int k = (int)(id % Kp), co = (int)(id / Kp);
            const int tap = k / Cip, ci = k - tap * Cip;
            float v = 0.f;
            if (co < Co && tap < T && ci < Ci) v = W[((long)co * Ci + ci) * T + tap];
            Wf[id] = (bf16)v;
        } else {
            const long j = id - nf;
            const int k = (int)(j % KC), ci = (int)((j / KC) % (16 * NTI)), cls = (int)(j / ((long)KC * 16 * NTI));
            const int ti = k / Co, co = k - ti * Co;
            int kh, kw;
            if (S == 2) { kh = (1 - (cls >> 1)) + 2 * (ti >> 1); kw = (1 - (cls & 1)) + 2 * (ti & 1); }
            else { kh = ti / 3; kw = ti % 3; }
            float v = 0.f;
            if (ci < Ci) v = W[((long)co * Ci + ci) * T + kh * KH + kw];
            Wd[j] = (bf16)v;
        }
    }
}

inline int pad16(int x) { return (x + 15) & ~15; }
inline int pad32(int x) { return (x + 31) & ~31; }

ConvGeo geo_of(int B, int Ci, int H, int W, int Co, int KH, int S, int P) {
    ConvGeo g;
    g.B = B; g.Ci = Ci; g.Cip = pad16(Ci); g.H = H; g.W = W; g.Co = Co;
    g.OH = (H + 2 * P - KH) / S + 1;
    g.OW = (W + 2 * P - KH) / S + 1;
    g.T = KH * KH;
    g.Kp = pad32(g.T * g.Cip);
    g.tiles_y = cdiv(g.OH, CT);
    g.tiles_x = cdiv(g.OW, CT);
    return g;
}
inline int mt_of(int Co) { return Co <= 16 ? 1 : Co <= 32 ? 2 : Co <= 64 ? 4 : 8; }
inline int kind_of(int KH, int S, int P) { return (KH == 4 && S == 2 && P == 1) ? 0 : (KH == 3 && S == 1 && P == 1) ? 1 : -1; }

}  // namespace

// supported: 4x4 / 2 / 1 and 3x3 / 1 / 1, Ci <= 64, Co a multiple of 8 up to 128 with (ceil(Co / 16), Cip / 16) in
// {(1,1), (2,1), (4,2), (8,4)}: the EarlyCNN stems of dim 64 / 128 / 256 (input channels 3 .. 16)
int m3l_conv_direct_supported(int dtype, int Ci, int Co, int KH, int S, int P, int first_layer) {
    if (dtype != 1 || kind_of(KH, S, P) < 0 || Co % 8 || Co < 8 || Co > 128 || Ci < 1 || Ci > 64) return 0;
    if (!first_layer && Ci % 8) return 0;
    const int MT = mt_of(Co), NT = pad16(Ci) / 16;
    return (MT == 1 && NT == 1) || (MT == 2 && NT == 1) || (MT == 4 && NT == 2) || (MT == 8 && NT == 4);
}

size_t m3l_conv_wf_elems(int Ci, int Co, int KH) { return (size_t)16 * mt_of(Co) * pad32(KH * KH * pad16(Ci)); }
size_t m3l_conv_wd_elems(int Ci, int Co, int KH, int S) {
    const int classes = S == 2 ? 4 : 1, tpc = S == 2 ? 4 : KH * KH;
    return (size_t)classes * pad16(Ci) * tpc * Co;
}
// groups of tiles (= partial-sum slabs) of a weight-gradient launch: enough workgroups to fill the chip several times over while the slabs
// stay small — 16 M floats in all (conv1 of the reference stem: 6144 floats per slab -> 2730 groups; conv3: 131 072 -> 128 groups)
int m3l_conv_wgrad_groups(int B, int Ci, int H, int W, int Co, int KH, int S, int P) {
    const ConvGeo g = geo_of(B, Ci, H, W, Co, KH, S, P);
    const long total = (long)B * g.tiles_y * g.tiles_x, per = (long)Co * Ci * KH * KH;
    return (int)std::max(1L, std::min(total, std::max(128L, (16L << 20) / per)));
}
size_t m3l_conv_wgrad_slab_elems(int B, int Ci, int H, int W, int Co, int KH, int S, int P) {
    return (size_t)m3l_conv_wgrad_groups(B, Ci, H, W, Co, KH, S, P) * Co * Ci * KH * KH;
}

int m3l_conv_prep(const float* W, int Ci, int Co, int KH, int S, void* Wf, void* Wd, hipStream_t st) {
    const int Cip = pad16(Ci), Kp = pad32(KH * KH * Cip), MT = mt_of(Co);
    const int tpc = S == 2 ? 4 : KH * KH;
    conv_prep_kernel<<<128, 256, 0, st>>>(W, Co, Ci, Cip, KH, S, Kp, MT, (bf16*)Wf, Cip / 16, tpc * Co, (bf16*)Wd);
    M3L_LAUNCH_CHECK();
    return 0;
}

#define CONV_KIND(K, ...)                                                          \
    if ((K) == 0) { constexpr int KH = 4, S = 2; __VA_ARGS__; }                    \
    else { constexpr int KH = 3, S = 1; __VA_ARGS__; }
#define CONV_MT(MTV, ...)                                                          \
    if ((MTV) == 1) { constexpr int MT = 1; __VA_ARGS__; }                         \
    else if ((MTV) == 2) { constexpr int MT = 2; __VA_ARGS__; }                    \
    else if ((MTV) == 4) { constexpr int MT = 4; __VA_ARGS__; }                    \
    else { constexpr int MT = 8; __VA_ARGS__; }

int m3l_conv_fwd(const ConvSrc* src, int Bsrc, int B, int Ci, int H, int W, int Co, int KH, int S, int P, const void* Wf, const float* bias,
                 void* out, hipStream_t st) {
    const int kind = kind_of(KH, S, P);
    M3L_CHECK(kind >= 0 && m3l_conv_direct_supported(1, Ci, Co, KH, S, P, src->nchw), "conv_fwd: unsupported shape Ci=%d Co=%d k=%d s=%d", Ci, Co, KH, S);
    const ConvGeo g = geo_of(B, Ci, H, W, Co, KH, S, P);
    const int PH = CT * S + KH - S;
    const size_t lds = (size_t)PH * PH * (g.Cip * 2 + 16);
    ProfScope prof("conv_fwd", (long)B * g.OH * g.OW, Co, g.T * Ci, 2.0 * B * g.OH * g.OW * (double)Co * g.T * Ci, st);
    const int grid = B * g.tiles_y * g.tiles_x;
    CONV_KIND(kind, CONV_MT(mt_of(Co), (conv_fwd_kernel<KH, S, MT><<<grid, 256, lds, st>>>(*src, Bsrc, g, P, (const bf16*)Wf, bias, (bf16*)out))));
    M3L_LAUNCH_CHECK();
    return 0;
}

int m3l_conv_wgrad(const ConvSrc* src, int Bsrc, int B, int Ci, int H, int W, int Co, int KH, int S, int P, const void* dY, float* slab,
                   float* dW, hipStream_t st) {
    const int kind = kind_of(KH, S, P);
    M3L_CHECK(kind >= 0 && m3l_conv_direct_supported(1, Ci, Co, KH, S, P, src->nchw), "conv_wgrad: unsupported shape Ci=%d Co=%d k=%d s=%d", Ci, Co, KH, S);
    const ConvGeo g = geo_of(B, Ci, H, W, Co, KH, S, P);
    const int MT = mt_of(Co), NT = g.Cip / 16;
    const int total = B * g.tiles_y * g.tiles_x, G = m3l_conv_wgrad_groups(B, Ci, H, W, Co, KH, S, P), tpw = cdiv(total, G);
    const int TW = std::min(4, 32 / (MT * NT)), TPG = std::min(g.T, 4 * TW), ngrp = cdiv(g.T, TPG);
    const int PH = CT * S + KH - S;
    const size_t lds = (((size_t)PH * PH * (g.Cip * 2 + 16) + 15) & ~(size_t)15) + (size_t)64 * (16 * MT + 8) * 2;
    ProfScope prof("conv_wgrad", (long)B * g.OH * g.OW, Co, g.T * Ci, 2.0 * B * g.OH * g.OW * (double)Co * g.T * Ci, st);
    const dim3 grid(cdiv(total, tpw), ngrp);
#define WG_LAUNCH(MTv, NTv, TWv) conv_wgrad_kernel<KH, S, MTv, NTv, TWv><<<grid, 256, lds, st>>>(*src, Bsrc, g, P, (const bf16*)dY, tpw, TPG, slab)
    CONV_KIND(kind, {
        if (MT == 1) WG_LAUNCH(1, 1, 4);
        else if (MT == 2) WG_LAUNCH(2, 1, 4);
        else if (MT == 4) WG_LAUNCH(4, 2, 4);
        else WG_LAUNCH(8, 4, 1);
    });
#undef WG_LAUNCH
    M3L_LAUNCH_CHECK();
    // fixed-order sum of the groups' slabs -> dW in the Conv2d layout (deterministic)
    const int cnt = Co * Ci * g.T;
    return m3l_reduce_rows(slab, (int)grid.x, cnt, cnt, dW, 0, st);
}

int m3l_conv_dgrad(const void* dY, int B, int Ci, int H, int W, int Co, int KH, int S, int P, const void* Wd, const void* act, void* dX,
                   hipStream_t st) {
    const int kind = kind_of(KH, S, P);
    M3L_CHECK(kind >= 0 && Ci % 8 == 0 && Ci <= 64 && Co % 8 == 0, "conv_dgrad: unsupported shape Ci=%d Co=%d k=%d s=%d", Ci, Co, KH, S);
    ConvGeo g = geo_of(B, Ci, H, W, Co, KH, S, P);
    const int NTI = g.Cip / 16, tpc = S == 2 ? 4 : KH * KH, KC = tpc * Co;
    M3L_CHECK(KC % 32 == 0 && (NTI == 1 || NTI == 2 || NTI == 4), "conv_dgrad: K = %d taps x %d channels must be a multiple of 32", tpc, Co);
    const int IT = S == 2 ? 16 : 8;
    const int grid = B * cdiv(H, IT) * cdiv(W, IT);
    const size_t lds = (size_t)10 * 10 * (Co * 2 + 16);
    ProfScope prof("conv_dgrad", (long)B * H * W, Ci, tpc * Co, 2.0 * B * H * W * (double)Ci * tpc * Co, st);
#define DG_LAUNCH(NTv) conv_dgrad_kernel<KH, S, NTv><<<grid, 256, lds, st>>>((const bf16*)dY, g, (const bf16*)Wd, KC, (const bf16*)act, (bf16*)dX)
    CONV_KIND(kind, {
        if (NTI == 1) DG_LAUNCH(1);
        else if (NTI == 2) DG_LAUNCH(2);
        else DG_LAUNCH(4);
    });
#undef DG_LAUNCH
    M3L_LAUNCH_CHECK();
    return 0;
}
