// HBM-bound kernels of the M3L MAE path (gfx950): LayerNorm fwd/bwd, patchify + gather + LayerNorm (patch embed
// front end), embed finalisation (+modality +sincos), random-mask rank sort, decoder un-shuffle scatter, masked-row
// gather/scatter, masked-patch MSE, column sums for bias gradients, weight down-cast/transposition.
// One wavefront (64 lanes) owns one row; all cross-lane reductions are wave reductions, all parameter-gradient
// reductions over rows go through per-workgroup partial slabs + a fixed-order reduce (bitwise reproducible).
#include "common.cuh"
#include "kernels.h"
#include <cstring>
#include <vector>

namespace {

constexpr int MAXV = 16;          // elements per lane: rows up to 1024 wide
constexpr int WPB = 4;            // waves (rows) per 256-thread block

__device__ __forceinline__ void store_val(float* p, float v) { *p = v; }
__device__ __forceinline__ void store_val(bf16* p, float v) { *p = (bf16)v; }

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm forward: y = (x - mean) * rstd * gamma + beta    (nn.LayerNorm, biased variance).  One wave per row, one
// 16-byte chunk (4 floats) per lane and pass: D = 192 keeps 48 lanes busy with a single global_load_dwordx4.
// MAXC = float4 chunks per lane (template parameter): 1 for D <= 256, 2 for D <= 512, 4 for D <= 1024 (D % 4 == 0).
// Specialising keeps a D = 192 row in 4 registers per array instead of 16 -> more waves per SIMD, more loads in flight.

template <typename TO> __device__ __forceinline__ void store4(TO* p, f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, f32x4 v) {
    bf16x4 o;
    o[0] = (bf16)v[0]; o[1] = (bf16)v[1]; o[2] = (bf16)v[2]; o[3] = (bf16)v[3];
    *reinterpret_cast<bf16x4*>(p) = o;
}
__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4(const bf16* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

// row statistics of x[row] held in registers v[] (chunk i of this lane = elements 4*(lane+64 i) .. +3)
template <int MAXC, typename TX>
__device__ __forceinline__ void row_stats(const TX* xr, int D, int lane, f32x4 (&v)[MAXC], float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int e = 4 * (lane + 64 * i);
        v[i] = load4(xr + min(e, D - 4));          // clamped, not `if (e < D)`: every chunk is requested before the first is used
        if (e >= D) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int e = 4 * (lane + 64 * i);
        if (e < D) {
            const f32x4 d = v[i] - mean;
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
    rstd = rsqrtf(wave_sum(q) / D + eps);
}

// TX: storage type of x (float; bf16 = the residual stream of a stack in m3l_set_residual_bf16 mode)
template <typename TO, int MAXC, typename TX>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TX* __restrict__ x, int M, int D, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, TO* __restrict__ y,
                                                       float* __restrict__ y32) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (row >= M) return;
    f32x4 v[MAXC], gm[MAXC], bt[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {               // gamma / beta travel with the row, not after its statistics
        const int ec = min(4 * (lane + 64 * i), D - 4);
        gm[i] = load4(gamma + ec);
        bt[i] = load4(beta + ec);
    }
    float mean, rstd;
    row_stats<MAXC>(x + (long)row * D, D, lane, v, eps, mean, rstd);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int e = 4 * (lane + 64 * i);
        if (e < D) {
            const f32x4 r = (v[i] - mean) * rstd * gm[i] + bt[i];
            if (y) store4<TO>(y + (long)row * D + e, r);
            if (y32) store4<float>(y32 + (long)row * D + e, r);
        }
    }
}

// LayerNorm backward.  dx_out = (dres ? dres : 0) + dLN/dx, optionally also stored in the compute type (dx_t_out: the
// operand of the next dgrad / wgrad GEMM).  Per-block partials -> part[G][3*D]: dgamma | dbeta | column sums of dx_out
// (= the bias gradient of the Linear whose output gradient dx_out is).
template <typename TD, typename TC, int MAXC, typename TX>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TD* __restrict__ dy, const TX* __restrict__ x, int M, int D,
                                                       const float* __restrict__ gamma, float eps, const float* __restrict__ dres,
                                                       float* __restrict__ dx_out, TC* __restrict__ dx_t_out, float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float red[WPB][3][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 dg[MAXC], db[MAXC], dc[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) dg[i] = db[i] = dc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 gm[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) gm[i] = load4(gamma + min(4 * (lane + 64 * i), D - 4));
    for (int row = blockIdx.x * WPB + wave; row < M; row += gridDim.x * WPB) {
        // every operand of the row (x, dy, the residual gradient) is requested before the statistics: one memory round trip per row
        // instead of three (loads under `if (e < D)` sit in their own basic blocks, each with its wait)
        f32x4 v[MAXC], gdy[MAXC], dyv[MAXC], rs[MAXC];
        const TD* dyr = dy + (long)row * D;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ec = min(4 * (lane + 64 * i), D - 4);
            dyv[i] = load4(dyr + ec);
            // (bf16 residual stream — TX is bf16 —: the residual gradient arrives in the compute type too)
            rs[i] = dres ? load4(reinterpret_cast<const TX*>(dres) + (long)row * D + ec) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float mean, rstd;
        row_stats<MAXC>(x + (long)row * D, D, lane, v, eps, mean, rstd);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int e = 4 * (lane + 64 * i);
            if (e < D) {
                const f32x4 xh = (v[i] - mean) * rstd;
                const f32x4 d = dyv[i];
                dg[i] += d * xh;
                db[i] += d;
                const f32x4 gd = d * gm[i];
                gdy[i] = gd;
                v[i] = xh;
                s1 += (gd[0] + gd[1]) + (gd[2] + gd[3]);
                const f32x4 t = gd * xh;
                s2 += (t[0] + t[1]) + (t[2] + t[3]);
            }
        }
        s1 = wave_sum(s1) / D;
        s2 = wave_sum(s2) / D;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int e = 4 * (lane + 64 * i);
            if (e < D) {
                f32x4 r = (gdy[i] - s1 - v[i] * s2) * rstd;
                if (dres) r += rs[i];
                dc[i] += r;
                if (dx_out) store4<float>(dx_out + (long)row * D + e, r);
                if (dx_t_out) store4<TC>(dx_t_out + (long)row * D + e, r);
            }
        }
    }
    // reduce the 4 waves' register partials through LDS in a fixed order, 256 columns at a time
    float* out = part + (long)blockIdx.x * 3 * D;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        if (256 * i >= D) break;                 // uniform
        *reinterpret_cast<f32x4*>(&red[wave][0][4 * lane]) = dg[i];
        *reinterpret_cast<f32x4*>(&red[wave][1][4 * lane]) = db[i];
        *reinterpret_cast<f32x4*>(&red[wave][2][4 * lane]) = dc[i];
        __syncthreads();
        const int e = 256 * i + threadIdx.x;
        if (e < D) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                out[k * D + e] = (red[0][k][threadIdx.x] + red[1][k][threadIdx.x]) + (red[2][k][threadIdx.x] + red[3][k][threadIdx.x]);
        }
        __syncthreads();
    }
}

// out[j] = (accumulate ? out[j] : 0) + sum_g part[g*stride + j],  j < count.
// 32 columns x 32 row-slices per 1024-thread block; each slice sums its rows in a fixed order with 8 independent accumulators
// (G = 512 partial rows cost 2 dependent load rounds), then the 32 slices are added in a fixed order through LDS -> bitwise
// reproducible.  These reductions follow every partial-sum kernel of the step and are pure latency: the loads in flight per thread
// set their duration.
__device__ __forceinline__ void reduce_cols_32x32(const float* __restrict__ p, int G, int stride, int j, bool valid,
                                                  float* __restrict__ out, int accumulate) {
    __shared__ float red[32][33];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    float s = 0.f;
    if (valid) {
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int g = sl;
        for (; g + 224 < G; g += 256) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += p[(long)(g + 32 * u) * stride + j];
        }
        for (; g < G; g += 32) a[0] += p[(long)g * stride + j];
        s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    red[sl][cl] = s;
    __syncthreads();
    if (sl == 0 && valid) {
        float t = red[0][cl];
#pragma unroll
        for (int i = 1; i < 32; ++i) t += red[i][cl];
        out[j] = accumulate ? out[j] + t : t;
    }
}

__global__ __launch_bounds__(1024) void reduce_rows_kernel(const float* __restrict__ part, int G, int stride, int count,
                                                             float* __restrict__ out, int accumulate) {
    const int j = blockIdx.x * 32 + (threadIdx.x & 31);
    reduce_cols_32x32(part, G, stride, j, j < count, out, accumulate);
}

// out[0] = (accumulate ? out[0] : 0) + sum_i v[i]: one workgroup, coalesced strided partial sums, fixed-order tree (the loss partials)
__global__ __launch_bounds__(1024) void reduce_vec_kernel(const float* __restrict__ v, int G, float* __restrict__ out, int accumulate) {
    __shared__ float red[16];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int i = threadIdx.x;
    for (; i + 3072 < G; i += 4096) { a0 += v[i]; a1 += v[i + 1024]; a2 += v[i + 2048]; a3 += v[i + 3072]; }
    for (; i < G; i += 1024) a0 += v[i];
    float s = wave_sum((a0 + a1) + (a2 + a3));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w];
        out[0] = accumulate ? out[0] + t : t;
    }
}

// up to 4 equal-width column segments of one partial slab, each to its own destination (blockIdx.y = segment)
struct ReduceSegs { float* out[4]; };
__global__ __launch_bounds__(1024) void reduce_rows_seg_kernel(const float* __restrict__ part, int G, int stride, int width,
                                                                 ReduceSegs segs, int accumulate) {
    float* out = segs.out[blockIdx.y];
    if (!out) return;                            // uniform per block
    const int j = blockIdx.x * 32 + (threadIdx.x & 31);
    reduce_cols_32x32(part + (long)blockIdx.y * width, G, stride, j, j < width, out, accumulate);
}

// the same reduction for MANY partial slabs in one launch (blockIdx.z = slab): the dgamma / dbeta / bias-gradient partials of every
// LayerNorm backward of a transformer stack are reduced once, off the critical path, instead of by one small launch per LayerNorm
__global__ __launch_bounds__(1024) void reduce_rows_batch_kernel(ReduceBatch b, int accumulate) {
    const ReduceBatchItem it = b.it[blockIdx.z];
    float* out = it.out[blockIdx.y];
    if (!out) return;                            // uniform per block
    const int j = blockIdx.x * 32 + (threadIdx.x & 31);
    reduce_cols_32x32(it.part + (long)blockIdx.y * b.D, it.G, 3 * b.D, j, j < b.D, out, accumulate);
}

// column sums of Y[M, N] (ld) -> part[G][N]; block = 256 threads, a block owns rows_per_block consecutive rows.  A thread owns 4 consecutive
// columns (one 8- / 16-byte load per row instead of a 2-byte one: at the 2352-wide patches of cfg 5 the scalar version ran 45 us for
// 36 MB) and keeps four independent row chains.
template <typename T> struct Vec4;
template <> struct Vec4<bf16> { typedef bf16x4 type; };
template <> struct Vec4<float> { typedef f32x4 type; };
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ Y, int M, int N, int ld, int rows_per_block,
                                                       float* __restrict__ part, int vec_ok) {
    typedef typename Vec4<T>::type V4;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    for (int c = threadIdx.x * 4; c < N; c += 1024) {
        if (vec_ok && c + 4 <= N) {
            f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
            auto ld4 = [&](int r) {
                const V4 v = *reinterpret_cast<const V4*>(Y + (long)r * ld + c);
                return f32x4{to_f32(v[0]), to_f32(v[1]), to_f32(v[2]), to_f32(v[3])};
            };
            int r = r0;
            for (; r + 3 < r1; r += 4) {
                s0 += ld4(r);
                s1 += ld4(r + 1);
                s2 += ld4(r + 2);
                s3 += ld4(r + 3);
            }
            for (; r < r1; ++r) s0 += ld4(r);
            *reinterpret_cast<f32x4*>(part + (long)blockIdx.x * N + c) = (s0 + s1) + (s2 + s3);
        } else {
            for (int cc = c; cc < min(N, c + 4); ++cc) {
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;      // four independent load -> add chains per thread
                int r = r0;
                for (; r + 3 < r1; r += 4) {
                    s0 += to_f32(Y[(long)r * ld + cc]);
                    s1 += to_f32(Y[(long)(r + 1) * ld + cc]);
                    s2 += to_f32(Y[(long)(r + 2) * ld + cc]);
                    s3 += to_f32(Y[(long)(r + 3) * ld + cc]);
                }
                for (; r < r1; ++r) s0 += to_f32(Y[(long)r * ld + cc]);
                part[(long)blockIdx.x * N + cc] = (s0 + s1) + (s2 + s3);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// weights: fp32 master -> compute-type copy (zero-padded leading dim) + transposed copy (for dgrad as an NT GEMM)
// One 64x64 tile per workgroup, all matrices of the pack in one launch: coalesced fp32 rows in, coalesced compute-type rows out for both
// copies (the transposed one through LDS).
template <typename T>
__global__ __launch_bounds__(256) void prep_weights_kernel(WeightPack pack) {
    __shared__ float tile[64][65];
    int m = 0;
    while (m + 1 < pack.count && (int)blockIdx.x >= pack.tile0[m + 1]) ++m;
    const WeightDesc d = pack.d[m];
    const int tcols = (d.cols + 63) >> 6;
    const int t = blockIdx.x - pack.tile0[m];
    const int r0 = (t / tcols) << 6, c0 = (t % tcols) << 6;
    const int tid = threadIdx.x, q = tid & 15, p = tid >> 4;
    T* dst = reinterpret_cast<T*>(d.dst);
    T* dstT = reinterpret_cast<T*>(d.dstT);
    const bool vec_in = (d.cols & 3) == 0 && (reinterpret_cast<uintptr_t>(d.src) & 15) == 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + p + 16 * k, c = c0 + 4 * q;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < d.rows) {
            if (vec_in && c + 3 < d.cols) {
                const float4 f = *reinterpret_cast<const float4*>(d.src + (long)r * d.cols + c);
                v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < d.cols) v[j] = d.src[(long)r * d.cols + c + j];
            }
        }
        // the copy is [ld_dstT rows][ld_dst cols]: the padding (zero-padded K of the patch / head / conv GEMMs; < 64 either way, so the
        // tiles of the matrix cover it) is written here as zeros — no memset launches in front
        if (dst && r < d.ld_dstT) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c + j < d.ld_dst) dst[(long)r * d.ld_dst + c + j] = from_f32<T>(v[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[p + 16 * k][4 * q + j] = v[j];
    }
    if (!dstT) return;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + p + 16 * k, r = r0 + 4 * q;
        if (c >= d.ld_dst) continue;                 // transposed copy: [ld_dst rows][ld_dstT cols], padding zero
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (r + j < d.ld_dstT) dstT[(long)c * d.ld_dstT + r + j] = from_f32<T>(tile[4 * q + j][p + 16 * k]);
    }
}

template <typename T>
__global__ void axpy_t_kernel(const float* __restrict__ x, const T* __restrict__ o, long count, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = x[i] + to_f32(o[i]);
}
template <typename T>
__global__ void cast_kernel(const float* __restrict__ x, long count, T* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = from_f32<T>(x[i]);
}

template <typename T>
__global__ void scale_by_dev_kernel(const T* __restrict__ x, long count, const float* __restrict__ s, T* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = from_f32<T>(to_f32(x[i]) * s[0]);
}

// out = src * (ref "on" ? scale : 0): MODE 0 = ref is a uint8 keep-mask (nn.Dropout forward: scale = 1 / (1 - p)); MODE 1 = ref is the f32
// OUTPUT of ReLU (+ Dropout) whose input gradient is wanted: on where ref > 0 (the fusion MLP of the cfg-5 extractor,
// models/pretrain_models_dino_cat_mae.py:828-836)
template <int MODE>
__global__ void mask_scale_kernel(const float* __restrict__ src, const void* __restrict__ ref, float scale, long count, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const bool on = MODE == 0 ? reinterpret_cast<const uint8_t*>(ref)[i] != 0 : reinterpret_cast<const float*>(ref)[i] > 0.f;
    out[i] = on ? src[i] * scale : 0.f;
}
// ViT token assembly of the frozen image encoder (DINOv2 prepare_tokens_with_masks; consumer: models/pretrain_models_dino_cat_mae.py:886):
// tok[b, 0] = cls + pos[0];  tok[b, 1 .. R] = registers (no position);  tok[b, 1 + R + i] = emb[b, i] + pos[1 + i]
__global__ void vit_tokens_kernel(const float* __restrict__ emb, const float* __restrict__ cls, const float* __restrict__ regs,
                                  const float* __restrict__ pos, int B, int npatch, int R, int D, float* __restrict__ tok) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = 1 + R + npatch;
    if (i >= (long)B * n * D) return;
    const int d = (int)(i % D), t = (int)((i / D) % n), b = (int)(i / ((long)D * n));
    float v;
    if (t == 0) v = cls[d] + pos[d];
    else if (t <= R) v = regs[(long)(t - 1) * D + d];
    else v = emb[((long)b * npatch + (t - 1 - R)) * D + d] + pos[(long)(t - R) * D + d];
    tok[i] = v;
}
// dst[b, :] = a[b, 0 .. na) | b2[b, 0 .. nb)   (torch.cat((pooled, dino), dim=-1), :899) and its adjoint split
__global__ void concat2_kernel(const float* __restrict__ a, int na, const float* __restrict__ b2, int nb, int rows, float* __restrict__ dst, int split) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int w = na + nb;
    if (i >= (long)rows * w) return;
    const int r = (int)(i / w), c = (int)(i % w);
    if (!split) dst[i] = c < na ? a[(long)r * na + c] : b2[(long)r * nb + c - na];
    else if (c < na) const_cast<float*>(a)[(long)r * na + c] = dst[i];
    else if (b2) const_cast<float*>(b2)[(long)r * nb + c - na] = dst[i];
}

// ---------------------------------------------------------------------------------------------------------------
// EarlyCNN stem (pretrain_models.py:37-56) as im2col + MFMA GEMM.  K order = (ci, kh, kw) = the Conv2d weight layout, so
// the weight / weight-gradient matrices need no permutation.  Activations between the convolutions are NHWC
// ([B*H*W, C], compute type) = exactly the [tokens, channels] matrix the next GEMM and the transformer consume.
template <typename T>
__global__ void im2col_kernel(ConvSrc cs, int Bsrc, int Ci, int H, int W, int KH, int S, int P, int OH, int OW, int Kpad,
                              long total, T* __restrict__ col) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int k = (int)(idx % Kpad);
    const long r = idx / Kpad;
    const int ow = (int)(r % OW), oh = (int)((r / OW) % OH);
    const int b = (int)(r / ((long)OW * OH));
    float v = 0.f;
    if (k < Ci * KH * KH) {
        const int kw = k % KH, kh = (k / KH) % KH, ci = k / (KH * KH);
        const int ih = oh * S + kh - P, iw = ow * S + kw - P;
        if (ih >= 0 && ih < H && iw >= 0 && iw < W) {
            if (cs.nchw) {
                const float* src = reinterpret_cast<const float*>(cs.src[b / Bsrc]);
                v = src[(((long)(b % Bsrc) * Ci + ci) * H + ih) * W + iw];
            } else {
                v = to_f32(reinterpret_cast<const T*>(cs.src[0])[(((long)b * H + ih) * W + iw) * Ci + ci]);
            }
        }
    }
    col[idx] = from_f32<T>(v);
}

// 4 x 4 kernels in bf16 (the three strided convolutions of the stem): one thread per (output pixel, input channel) writes that channel's
// 16 consecutive k = 32 contiguous bytes of the column row; lanes run over ci, so an NHWC source is read in coalesced 2-byte runs and
// an NCHW fp32 source in 16-byte rows.  (Kpad == Ci * 16: no pad columns.)
__global__ __launch_bounds__(256) void im2col4_bf16_kernel(ConvSrc cs, int Bsrc, int Ci, int H, int W, int S, int P, int OH, int OW, long total,
                                                           bf16* __restrict__ col) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;       // (row r, ci), ci fastest
    if (idx >= total) return;
    const int ci = (int)(idx % Ci);
    const long r = idx / Ci;
    const int ow = (int)(r % OW), oh = (int)((r / OW) % OH);
    const int b = (int)(r / ((long)OW * OH));
    bf16x8 out[2];
#pragma unroll
    for (int kh = 0; kh < 4; ++kh) {
        const int ih = oh * S + kh - P;
        const bool rok = ih >= 0 && ih < H;
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
            const int iw = ow * S + kw - P;
            float v = 0.f;
            if (rok && iw >= 0 && iw < W) {
                if (cs.nchw) v = reinterpret_cast<const float*>(cs.src[b / Bsrc])[(((long)(b % Bsrc) * Ci + ci) * H + ih) * W + iw];
                else v = (float)reinterpret_cast<const bf16*>(cs.src[0])[(((long)b * H + ih) * W + iw) * Ci + ci];
            }
            out[kh >> 1][(kh & 1) * 4 + kw] = (bf16)v;
        }
    }
    bf16x8* dst = reinterpret_cast<bf16x8*>(col + (r * Ci + ci) * 16);
    dst[0] = out[0];
    dst[1] = out[1];
}

// adjoint of im2col in gather form (no atomics) + ReLU mask of the layer input: dX[b,ih,iw,ci] = act > 0 ? sum : 0
template <typename T>
__global__ void col2im_relu_kernel(const T* __restrict__ dcol, int Btot, int Ci, int H, int W, int KH, int S, int P, int OH, int OW,
                                   int Kpad, const T* __restrict__ act, T* __restrict__ dX) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)Btot * H * W * Ci) return;
    const int ci = (int)(idx % Ci);
    const long pix = idx / Ci;
    const int iw = (int)(pix % W), ih = (int)((pix / W) % H);
    const int b = (int)(pix / ((long)W * H));
    float s = 0.f;
    if (to_f32(act[idx]) > 0.f) {
        for (int kh = 0; kh < KH; ++kh) {
            const int t = ih + P - kh;
            if (t < 0 || t % S) continue;
            const int oh = t / S;
            if (oh >= OH) continue;
            for (int kw = 0; kw < KH; ++kw) {
                const int u = iw + P - kw;
                if (u < 0 || u % S) continue;
                const int ow = u / S;
                if (ow >= OW) continue;
                s += to_f32(dcol[(((long)b * OH + oh) * OW + ow) * Kpad + (ci * KH + kh) * KH + kw]);
            }
        }
    }
    dX[idx] = from_f32<T>(s);
}

// tokens[b, pos] = stem_token + modality[m(pos)] + sincos[pos]   (pretrain_models.py:202-216 on the EarlyCNN outputs;
// tactile stem tokens arrive sensor-major: [k*B, n_tac, D])
__global__ __launch_bounds__(256) void tokens_assemble_kernel(const float* __restrict__ img_tok, const float* __restrict__ tac_tok, int B,
                                                                int D, int n_img, int n_tac, int k, const float* __restrict__ mod,
                                                                const float* __restrict__ pos_img, const float* __restrict__ pos_tac,
                                                                float* __restrict__ tokens) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * WPB + (threadIdx.x >> 6);
    const int N = n_img + k * n_tac;
    if (r >= (long)B * N) return;
    const int b = (int)(r / N), pos = (int)(r % N);
    const float *src, *prow;
    int m;
    if (pos < n_img) {
        src = img_tok + ((long)b * n_img + pos) * D;
        prow = pos_img + (long)pos * D;
        m = 0;
    } else {
        const int local = pos - n_img, s = local / n_tac, p = local % n_tac;
        src = tac_tok + (((long)s * B + b) * n_tac + p) * D;
        prow = pos_tac + (long)local * D;
        m = 1 + s;
    }
    float* out = tokens + r * D;
    for (int e = lane; e < D; e += 64) out[e] = src[e] + mod[(long)m * D + e] + prow[e];
}
__global__ __launch_bounds__(256) void tokens_assemble_bwd_kernel(const float* __restrict__ dtok, int B, int D, int n_img, int n_tac, int k,
                                                                    float* __restrict__ d_img, float* __restrict__ d_tac,
                                                                    float* __restrict__ part) {
    extern __shared__ float sm[];   // [WPB][(1 + k) * D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int N = n_img + k * n_tac, PL = (1 + k) * D;
    for (int i = threadIdx.x; i < WPB * PL; i += 256) sm[i] = 0.f;
    __syncthreads();
    float* my = sm + wave * PL;
    for (long r = (long)blockIdx.x * WPB + wave; r < (long)B * N; r += (long)gridDim.x * WPB) {
        const int b = (int)(r / N), pos = (int)(r % N);
        float* dst;
        int m;
        if (pos < n_img) {
            dst = d_img + ((long)b * n_img + pos) * D;
            m = 0;
        } else {
            const int local = pos - n_img, s = local / n_tac, p = local % n_tac;
            dst = d_tac + (((long)s * B + b) * n_tac + p) * D;
            m = 1 + s;
        }
        const float* g = dtok + r * D;
        for (int e = lane; e < D; e += 64) {
            const float d = g[e];
            dst[e] = d;
            my[m * D + e] += d;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PL; i += 256)
        part[(long)blockIdx.x * PL + i] = sm[i] + sm[PL + i] + sm[2 * PL + i] + sm[3 * PL + i];
}

// Adam over ONE flat parameter / gradient / moment buffer (reference optimizer: torch.optim.Adam(mae.parameters(), lr=1e-4),
// models/ppo_mae.py:182-183): same update rule and operation order as torch's, one launch for all 7.3 M parameters.
// DEC = 1: torch.optim.AdamW (the optimizer of VTMAE.initialize_training, pretrain_models.py:675): decoupled weight decay, p *= 1 - lr wd
// before the Adam update, the gradient itself unchanged.  clip_dev (may be null): a device-side factor on the gradient — the
// clip_grad_norm_ coefficient of pretrain_models.py:710, computed by gradnorm_* below — applied inside the update; with scale_g the
// clipped gradient is also written back, as clip_grad_norm_ leaves it.
template <int DEC>
__global__ void adam_flat_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                 long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                 const float* __restrict__ bc_dev, float gscale, const float* __restrict__ clip_dev, int scale_g) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    if (bc_dev) {            // graph-capturable form: bias corrections of the device-side step counter
        bc1 = bc_dev[0];
        bc2_sqrt = bc_dev[1];
    }
    if (clip_dev) gscale *= clip_dev[0];
    if (i + 4 <= n) {
        f32x4 pp = *reinterpret_cast<f32x4*>(p + i), gg = *reinterpret_cast<const f32x4*>(g + i);
        f32x4 mm = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            gg[j] = gg[j] * gscale;                                     // gscale: 1 / world of a SUM all-reduce (1.0f is exact)
            if (DEC) pp[j] = pp[j] * wd;                                // DEC: wd carries the factor 1 - lr * weight_decay
            const float gr = DEC ? gg[j] : gg[j] + wd * pp[j];
            mm[j] = mm[j] + (gr - mm[j]) * (1.0f - b1);                 // lerp, as torch: exp_avg.lerp_(grad, 1 - beta1)
            vv[j] = vv[j] * b2 + (1.0f - b2) * gr * gr;
            const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
            pp[j] = pp[j] - (lr / bc1) * (mm[j] / denom);
        }
        *reinterpret_cast<f32x4*>(p + i) = pp;
        *reinterpret_cast<f32x4*>(m + i) = mm;
        *reinterpret_cast<f32x4*>(v + i) = vv;
        if (scale_g) *reinterpret_cast<f32x4*>(g + i) = gg;
    } else {
        for (long k = i; k < n; ++k) {
            const float gs = g[k] * gscale;
            if (DEC) p[k] = p[k] * wd;
            const float gr = DEC ? gs : gs + wd * p[k];
            m[k] = m[k] + (gr - m[k]) * (1.0f - b1);
            v[k] = v[k] * b2 + (1.0f - b2) * gr * gr;
            p[k] = p[k] - (lr / bc1) * (m[k] / (sqrtf(v[k]) / bc2_sqrt + eps));
            if (scale_g) g[k] = gs;
        }
    }
}

// torch.nn.utils.clip_grad_norm_(params, max_norm) over the flat gradient buffer (pretrain_models.py:710): out[0] = the clip coefficient
// min(1, max_norm / (||gscale g||_2 + 1e-6)), out[1] = the norm.  Two launches, fixed summation order (bit-reproducible).
__global__ __launch_bounds__(256) void gradnorm_part_kernel(const float* __restrict__ g, long n, float* __restrict__ part) {
    __shared__ float red[4];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(g + i);
            a0 += v[0] * v[0]; a1 += v[1] * v[1]; a2 += v[2] * v[2]; a3 += v[3] * v[3];
        } else {
            for (long k = i; k < n; ++k) a0 += g[k] * g[k];
        }
    }
    const float w = wave_sum((a0 + a1) + (a2 + a3));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(64) void gradnorm_final_kernel(const float* __restrict__ part, int G, float gscale, float max_norm, float* __restrict__ out) {
    float a = 0.f;
    for (int i = threadIdx.x; i < G; i += 64) a += part[i];
    a = wave_sum(a);
    if (threadIdx.x == 0) {
        const float norm = sqrtf(a) * fabsf(gscale);
        const float c = max_norm / (norm + 1e-6f);
        out[0] = c < 1.0f ? c : 1.0f;
        out[1] = norm;
    }
}

// step += 1; bias corrections 1 - beta^step in double (as torch's Python-float arithmetic), for the capturable Adam
__global__ void adam_bias_kernel(int* __restrict__ step, float b1, float b2, float* __restrict__ bc) {
    const int t = *step + 1;
    *step = t;
    bc[0] = (float)(1.0 - pow((double)b1, (double)t));
    bc[1] = (float)sqrt(1.0 - pow((double)b2, (double)t));
}

// vt_load (utils/pretrain_utils.py:7-57): image NHWC -> NCHW; tactile (B, 3*S*fs, h, w): sensor s takes channels {f*3S + 3s + c};
// both normalised as the reference writes it, (x - lo) / (hi - lo) in fp32 with an IEEE division (defaults [0, 1] = identity and
// [-1, 1] = (x + 1) / 2, both exact).  TI = float or uint8_t (torch.Tensor(uint8 array) converts exactly).
template <typename TI>
__global__ void vt_image_kernel(const TI* __restrict__ in, int B, int H, int W, int C, float lo, float span, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * C * H * W) return;
    const int w = (int)(i % W), h = (int)((i / W) % H), c = (int)((i / ((long)W * H)) % C), b = (int)(i / ((long)W * H * C));
    out[i] = __fdiv_rn((float)in[(((long)b * H + h) * W + w) * C + c] - lo, span);
}
struct VtTactileOut { float* p[M3L_MAX_SENSORS]; };
template <typename TI>
__global__ void vt_tactile_kernel(const TI* __restrict__ in, int B, int th, int tw, int S, int fs, float lo, float span, VtTactileOut o) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int CH = 3 * S * fs;
    if (i >= (long)B * CH * th * tw) return;
    const int px = (int)(i % ((long)th * tw)), ch = (int)((i / ((long)th * tw)) % CH), b = (int)(i / ((long)th * tw * CH));
    const int f = ch / (3 * S), r = ch % (3 * S), s = r / 3, c = r % 3;
    o.p[s][(((long)b * 3 * fs) + f * 3 + c) * th * tw + px] = __fdiv_rn((float)in[i] - lo, span);
}

// ---------------------------------------------------------------------------------------------------------------
// random-mask sampling: per (sample, modality) row of noise -> STABLE ascending rank -> masked / unmasked lists
// (reference: torch.rand(B, n).argsort(-1), models/pretrain_models.py:229-248; tie-break = ascending index)
// (one launch for all modality groups of a sample: blockIdx.y picks the group's noise / sizes / offsets)
__global__ void mask_rank_kernel(MaskRankArgs a, int64_t* __restrict__ masked, int masked_ld, int64_t* __restrict__ unmasked,
                                 int unmasked_ld) {
    extern __shared__ float keys[];
    const int b = blockIdx.x;
    const MaskRankGroup gr = a.g[blockIdx.y];
    const float* __restrict__ noise = gr.noise;
    const int n = gr.n, nm = gr.nm, token_offset = gr.token_offset, masked_off = gr.masked_off, unmasked_off = gr.unmasked_off;
    for (int i = threadIdx.x; i < n; i += blockDim.x) keys[i] = noise[(long)b * n + i];
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float k = keys[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float o = keys[j];
            rank += (o < k) || (o == k && j < i);
        }
        const int64_t tok = (int64_t)(i + token_offset);
        if (rank < nm)
            masked[(long)b * masked_ld + masked_off + rank] = tok;
        else
            unmasked[(long)b * unmasked_ld + unmasked_off + rank - nm] = tok;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// patch front end.  Row r = b*cnt + jj of a modality group; its token is idx[b*idx_ld + j0 + jj] (or base + jj when
// idx == nullptr: all patches).  local = token - base, sensor s = local / npatch, patch = local % npatch.
// patch element e = (p1*P + p2)*C + c  <->  src[s][b, c, ph*P + p1, pw*P + p2]   (einops 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)')
__device__ __forceinline__ void patch_locate(const PatchGroup& pg, const int64_t* idx, int idx_ld, int j0, int cnt, int row,
                                             int& b, int& s, int& ph, int& pw, int& local) {
    b = row / cnt;
    const int jj = row % cnt;
    const int tok = idx ? (int)idx[(long)b * idx_ld + j0 + jj] : pg.base + jj;
    local = tok - pg.base;
    s = local / pg.npatch;
    const int pidx = local % pg.npatch;
    const int gw = pg.W / pg.P;
    ph = pidx / gw;
    pw = pidx % gw;
}
// first pixel of channel 0 of the patch (the source pointer is picked ONCE per row: pg.src[s] with a run-time s is a load from a
// private copy of the argument struct, and per element it was a dependent round trip in front of every gather)
__device__ __forceinline__ const float* patch_base(const PatchGroup& pg, int b, int s, int ph, int pw) {
    const float* src = pg.src[0];
#pragma unroll
    for (int i = 1; i < M3L_MAX_SENSORS; ++i) src = (s == i) ? pg.src[i] : src;
    return src + ((long)b * pg.C * pg.H + ph * pg.P) * pg.W + pw * pg.P;
}
// The t-th element of a patch in IMAGE order (c, p1, p2 with p2 fastest) and its patch-vector index e.  Kernels that walk a
// patch with consecutive lanes use this order: P consecutive lanes read P consecutive pixels of one image row (one 4 P-byte
// segment) instead of 64 lanes touching 64 different (channel, row) places; sums over the patch do not care about the order.
// t / d by one multiply: m = floor(2^32 / d) + 1 makes (t * m) >> 32 exact while t * d < 2^32 (here t < 2560, d <= 2560).  A patch
// element otherwise costs two ~35-instruction integer divisions — at the 2352-wide patches of cfg 5 that arithmetic, not the memory,
// set the time of the patch kernels (mse 104 us for 120 MB).
// The lanes always walk a patch in IMAGE order; what has to be read or written in PATCH-VECTOR order beside it (xn, dxn, pred, dpred,
// gamma, beta) goes through an LDS copy of the patch in patch-vector order: the gather scatters into it (ds_write_b32, stride C), and a
// second pass reads it back as 16-byte pieces next to fully coalesced 16-byte global accesses.  (Walking the patch vector with stride C
// straight in global memory made every wave store of cfg 5's 2352-wide bf16 rows touch 12+ cache lines: patch_ln 34 us for 1920 rows.)
struct PatchDiv {
    unsigned m_pp, m_p;
    int pp;
};
__device__ __forceinline__ PatchDiv patch_div(const PatchGroup& pg) {
    PatchDiv d;
    d.pp = pg.P * pg.P;
    d.m_pp = 0xFFFFFFFFu / (unsigned)d.pp + 1u;
    d.m_p = 0xFFFFFFFFu / (unsigned)pg.P + 1u;
    return d;
}
// offset (from patch_base) of the t-th element of the walk and its patch-vector index e
__device__ __forceinline__ int patch_off_t(const PatchGroup& pg, const PatchDiv& dv, int t, int& e) {
    const int c = (int)__umulhi((unsigned)t, dv.m_pp);
    const int rem = t - c * dv.pp;
    const int p1 = (int)__umulhi((unsigned)rem, dv.m_p);
    const int p2 = rem - p1 * pg.P;
    e = rem * pg.C + c;
    return (c * pg.H + p1) * pg.W + p2;
}
// gather the patch of one row into `my` (patch-vector order) and return its LayerNorm statistics.  All NV gathers of a lane are requested
// before the first is used (indices past the patch clamp to its last element: a load under `if (t < pd)` sits in its own basic block with
// its wait, and the 40 gathers of a 2352-wide patch were 40 dependent round trips: 36 us for 1920 rows).
template <int NV>
__device__ __forceinline__ void patch_stage_stats(const PatchGroup& pg, const PatchDiv& dv, const float* __restrict__ base, int lane, int pd,
                                                  float eps, float* __restrict__ my, float& mean, float& rstd) {
    float v[NV];
    int ei[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = base[patch_off_t(pg, dv, min(lane + 64 * i, pd - 1), ei[i])];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (lane + 64 * i < pd) my[ei[i]] = v[i];
        else v[i] = 0.f;
        sum += v[i];
    }
    mean = wave_sum(sum) / pd;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float d = (lane + 64 * i < pd) ? v[i] - mean : 0.f;
        q += d * d;
    }
    rstd = rsqrtf(wave_sum(q) / pd + eps);
}

// gather + LayerNorm(pd) -> xn [rows, pdpad] (compute type; pad columns zeroed)
// NV = elements per lane: 4 for patch dims <= 256 (cfg 2: 48), 16 for <= 1024, 40 for <= 2560 (cfg 5: 14x14 patches of 4 stacked RGB
// frames = 2352); every lane requests all NV (clamped), so NV follows the patch dim
// vec: pd, pdpad multiples of 4 and xn / gamma / beta aligned for 4-element accesses (host-checked)
template <typename T, int NV>
__global__ __launch_bounds__(256) void patch_ln_kernel(PatchGroup pg, const int64_t* __restrict__ idx, int idx_ld, int j0, int cnt,
                                                         int rows, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float eps, T* __restrict__ xn, int pdpad, int vec) {
    __shared__ __attribute__((aligned(16))) float buf[WPB][NV * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = blockIdx.x * WPB + wave;
    const int pd = pg.C * pg.P * pg.P;
    const PatchDiv dv = patch_div(pg);
    float* my = buf[wave];
    const bool live = row < rows;
    float mean = 0.f, rstd = 0.f;
    if (live) {
        int b, s, ph, pw, local;
        patch_locate(pg, idx, idx_ld, j0, cnt, row, b, s, ph, pw, local);
        patch_stage_stats<NV>(pg, dv, patch_base(pg, b, s, ph, pw), lane, pd, eps, my, mean, rstd);
    }
    __syncthreads();
    if (!live) return;
    T* out = xn + (long)row * pdpad;
    if (vec) {
        constexpr int NC = NV / 4;                 // pieces per lane: gamma / beta of all of them are requested up front
        f32x4 gm[NC], bt[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int ec = min(4 * (lane + 64 * k), pd - 4);
            gm[k] = load4(gamma + ec);
            bt[k] = load4(beta + ec);
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int e = 4 * (lane + 64 * k);
            if (e < pdpad) {
                f32x4 r = {0.f, 0.f, 0.f, 0.f};
                if (e < pd) r = (*reinterpret_cast<const f32x4*>(my + e) - mean) * rstd * gm[k] + bt[k];
                store4<T>(out + e, r);
            }
        }
    } else {
        for (int e = lane; e < pdpad; e += 64) out[e] = from_f32<T>(e < pd ? (my[e] - mean) * rstd * gamma[e] + beta[e] : 0.f);
    }
}

// backward of the first patch LayerNorm w.r.t. its affine parameters (the input is data): part[G][2*pd].
// A lane owns the same 4-element pieces of the patch vector for every row of its wave (NV / 4 pieces: 20 accumulator registers per
// 256 elements); the four waves' sums meet in the patch buffers in a fixed order.
template <typename T, int NV>
__global__ __launch_bounds__(256) void patch_ln_bwd_kernel(PatchGroup pg, const int64_t* __restrict__ idx, int idx_ld, int j0, int cnt,
                                                             int rows, float eps, const T* __restrict__ dxn, int pdpad,
                                                             float* __restrict__ part, int vec) {
    __shared__ __attribute__((aligned(16))) float buf[WPB][NV * 64];
    constexpr int NC = NV / 4;
    static_assert(NV % 4 == 0, "4-element pieces");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pd = pg.C * pg.P * pg.P;
    const PatchDiv dv = patch_div(pg);
    float* my = buf[wave];
    f32x4 dg[NC], db[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) dg[k] = db[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int r0 = blockIdx.x * WPB; r0 < rows; r0 += gridDim.x * WPB) {       // block-uniform trip count (barriers inside)
        const int row = r0 + wave;
        const bool live = row < rows;
        float mean = 0.f, rstd = 0.f;
        if (live) {
            int b, s, ph, pw, local;
            patch_locate(pg, idx, idx_ld, j0, cnt, row, b, s, ph, pw, local);
            patch_stage_stats<NV>(pg, dv, patch_base(pg, b, s, ph, pw), lane, pd, eps, my, mean, rstd);
        }
        __syncthreads();
        if (live) {
            const T* dr = dxn + (long)row * pdpad;
            f32x4 d[NC];
#pragma unroll
            for (int k = 0; k < NC; ++k) {                                     // every piece of the row requested before the first is used
                const int e = 4 * (lane + 64 * k);
                if (vec) {
                    d[k] = load4(dr + min(e, pd - 4));
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) d[k][q] = to_f32(dr[min(e + q, pd - 1)]);
                }
            }
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int e = 4 * (lane + 64 * k);
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
                if (vec) {
                    if (e < pd) a = *reinterpret_cast<const f32x4*>(my + e);
                    else d[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (e + q < pd) a[q] = my[e + q];
                        else d[k][q] = 0.f;
                    }
                }
                dg[k] += d[k] * (a - mean) * rstd;
                db[k] += d[k];
            }
        }
        __syncthreads();                                                       // the buffers are refilled by the next row
    }
    float* out = part + (long)blockIdx.x * 2 * pd;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int k = 0; k < NC; ++k) *reinterpret_cast<f32x4*>(my + 4 * (lane + 64 * k)) = pass ? db[k] : dg[k];
        __syncthreads();
        for (int e = threadIdx.x; e < pd; e += 256) out[pass * pd + e] = buf[0][e] + buf[1][e] + buf[2][e] + buf[3][e];
        __syncthreads();
    }
}

// second patch LayerNorm + modality embedding + fixed sincos position -> tokens[b, j0 + jj, :]  (f32 residual stream)
__global__ __launch_bounds__(256) void embed_finalize_kernel(const float* __restrict__ E, int rows, int D, PatchGroup pg,
                                                               const int64_t* __restrict__ idx, int idx_ld, int j0, int cnt,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float eps, const float* __restrict__ mod, int mod0,
                                                               const float* __restrict__ pos, float* __restrict__ tokens, int L) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (row >= rows) return;
    int b, s, ph, pw, local;
    patch_locate(pg, idx, idx_ld, j0, cnt, row, b, s, ph, pw, local);
    const float* xr = E + (long)row * D;
    float v[MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int e = lane + 64 * i;
        v[i] = (e < D) ? xr[e] : 0.f;
        sum += v[i];
    }
    const float mean = wave_sum(sum) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int e = lane + 64 * i;
        const float d = (e < D) ? v[i] - mean : 0.f;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
    const int jj = row % cnt;
    float* out = tokens + ((long)b * L + j0 + jj) * D;
    const float* mrow = mod + (long)(mod0 + s) * D;
    const float* prow = pos + (long)local * D;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int e = lane + 64 * i;
        if (e < D) out[e] = (v[i] - mean) * rstd * gamma[e] + beta[e] + mrow[e] + prow[e];
    }
}

// backward of embed_finalize: dE (compute type) + partials [G][(2 + nslot) * D]: dgamma | dbeta | dmod[slot]
template <typename T>
__global__ __launch_bounds__(256) void embed_finalize_bwd_kernel(const float* __restrict__ dtok, int L, const float* __restrict__ E,
                                                                   int rows, int D, PatchGroup pg, const int64_t* __restrict__ idx,
                                                                   int idx_ld, int j0, int cnt, const float* __restrict__ gamma,
                                                                   float eps, T* __restrict__ dE, float* __restrict__ part, int nslot) {
    extern __shared__ float sm[];   // [WPB][(2 + nslot) * D]: one private slab per wave (no atomics -> fixed order)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int PL = (2 + nslot) * D;
    for (int i = threadIdx.x; i < WPB * PL; i += 256) sm[i] = 0.f;
    __syncthreads();
    float* my = sm + wave * PL;
    for (int row = blockIdx.x * WPB + wave; row < rows; row += gridDim.x * WPB) {
        int b, s, ph, pw, local;
        patch_locate(pg, idx, idx_ld, j0, cnt, row, b, s, ph, pw, local);
        const int jj = row % cnt;
        const float* xr = E + (long)row * D;
        const float* dyr = dtok + ((long)b * L + j0 + jj) * D;
        float v[MAXV], gdy[MAXV];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int e = lane + 64 * i;
            v[i] = (e < D) ? xr[e] : 0.f;
            sum += v[i];
        }
        const float mean = wave_sum(sum) / D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int e = lane + 64 * i;
            const float d = (e < D) ? v[i] - mean : 0.f;
            q += d * d;
        }
        const float rstd = rsqrtf(wave_sum(q) / D + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int e = lane + 64 * i;
            if (e < D) {
                const float xh = (v[i] - mean) * rstd;
                const float d = dyr[e];
                my[e] += d * xh;                           // element e is always owned by lane e % 64 of this wave
                my[D + e] += d;
                my[(2 + s) * D + e] += d;
                const float gd = d * gamma[e];
                gdy[i] = gd;
                v[i] = xh;
                s1 += gd;
                s2 += gd * xh;
            } else {
                gdy[i] = 0.f;
                v[i] = 0.f;
            }
        }
        s1 = wave_sum(s1) / D;
        s2 = wave_sum(s2) / D;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int e = lane + 64 * i;
            if (e < D) dE[(long)row * D + e] = from_f32<T>(rstd * (gdy[i] - s1 - v[i] * s2));
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PL; i += 256)
        part[(long)blockIdx.x * PL + i] = sm[i] + sm[PL + i] + sm[2 * PL + i] + sm[3 * PL + i];
}

// ---------------------------------------------------------------------------------------------------------------
// decoder un-shuffle (pretrain_models.py:279-307): dec_in[b, pos] = (visible ? src[b, j] : mask_token) + dmod[m(pos)] + dpos[pos]
template <typename TO>
__global__ __launch_bounds__(256) void unshuffle_fwd_kernel(const float* __restrict__ src, const float* __restrict__ mask_token,
                                                              const int64_t* __restrict__ unmasked, int nvis,
                                                              const int64_t* __restrict__ masked, int nmask, int B, int dd,
                                                              int n_img, int n_tac, const float* __restrict__ dmod,
                                                              const float* __restrict__ pos_img, const float* __restrict__ pos_tac,
                                                              TO* __restrict__ dec_in) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * WPB + (threadIdx.x >> 6);
    const int N = nvis + nmask;
    if (r >= (long)B * N) return;
    const int b = (int)(r / N), j = (int)(r % N);
    const bool vis = j < nvis;
    const int pos = (int)(vis ? unmasked[(long)b * nvis + j] : masked[(long)b * nmask + j - nvis]);
    const float* s = vis ? src + ((long)b * nvis + j) * dd : mask_token;
    const int m = pos < n_img ? 0 : 1 + (pos - n_img) / n_tac;
    const float* prow = pos < n_img ? pos_img + (long)pos * dd : pos_tac + (long)(pos - n_img) * dd;
    TO* out = dec_in + ((long)b * N + pos) * dd;
    const float* mrow = dmod + (long)m * dd;
    for (int e = 4 * lane; e < dd; e += 256) store4<TO>(out + e, load4(s + e) + load4(mrow + e) + load4(prow + e));      // (dd % 4 == 0)
}

// backward of the decoder-input assembly: dsrc[b, j] = dY[b, unmasked[b, j]] ; partials [G][(1 + nmod) * dd]: dmask_token | ddmod[m].
// Every position of a sample is either visible or masked exactly once, so no index is needed for the sums:
//   ddmod[m]    = sum of dY over ALL rows whose position belongs to modality m (contiguous position ranges),
//   dmask_token = (sum over all rows) - (sum over the visible rows).
// A wave walks a contiguous run of rows with 16-byte loads and keeps running column sums in registers, flushed into its private LDS
// slab when the modality changes (at most nmod + 1 times per sample); a second loop gathers its share of the visible rows.
template <int MAXC, typename TI>
__global__ __launch_bounds__(256) void unshuffle_bwd_kernel(const TI* __restrict__ dY, const int64_t* __restrict__ unmasked, int nvis,
                                                              int B, int N, int dd, int n_img, int n_tac, int nmod, int rows_per_wave,
                                                              int vis_per_wave, float* __restrict__ dsrc, float* __restrict__ part) {
    extern __shared__ float sm[];   // [WPB][(1 + nmod) * dd]: one private slab per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int PL = (1 + nmod) * dd;
    for (int i = threadIdx.x; i < WPB * PL; i += 256) sm[i] = 0.f;
    __syncthreads();
    float* my = sm + wave * PL;
    const long gw = (long)blockIdx.x * WPB + wave;
    f32x4 acc[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#define UNSH_FLUSH(slot, sign)                                                      \
    _Pragma("unroll") for (int c = 0; c < MAXC; ++c) {                              \
        const int e4 = (lane + 64 * c) * 4;                                         \
        if (e4 < dd) {                                                              \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                         \
                my[e4 + q] += (sign) * acc[c][q];                                   \
                if ((slot) >= 0) my[(1 + (slot)) * dd + e4 + q] += acc[c][q];       \
            }                                                                       \
        }                                                                           \
        acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};                                         \
    }
    // phase 1: every row of this wave's run, one contiguous same-modality segment at a time, four rows in flight
    const long total = (long)B * N;
    const long r0 = gw * rows_per_wave, r1 = r0 + rows_per_wave < total ? r0 + rows_per_wave : total;
    for (long r = r0; r < r1;) {
        const int pos = (int)(r % N);
        const int m = pos < n_img ? 0 : 1 + (pos - n_img) / n_tac;
        const int pos_end = pos < n_img ? n_img : min(N, n_img + m * n_tac);
        const long seg_end = min(r1, r - pos + pos_end);
        for (; r + 4 <= seg_end; r += 4) {
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const int e4 = (lane + 64 * c) * 4;
                if (e4 < dd) {
                    const TI* g = dY + r * dd + e4;
                    const f32x4 d0 = load4(g), d1 = load4(g + dd), d2 = load4(g + 2 * dd), d3 = load4(g + 3 * dd);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[c][q] += (d0[q] + d1[q]) + (d2[q] + d3[q]);
                }
            }
        }
        for (; r < seg_end; ++r) {
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const int e4 = (lane + 64 * c) * 4;
                if (e4 < dd) {
                    const f32x4 d = load4(dY + r * dd + e4);
                    acc[c][0] += d[0]; acc[c][1] += d[1]; acc[c][2] += d[2]; acc[c][3] += d[3];
                }
            }
        }
        { UNSH_FLUSH(m, 1.f) }
    }
    // phase 2: this wave's share of the visible rows: copy out, and subtract their sum from the mask-token slot (four rows in flight)
    const long vtotal = (long)B * nvis;
    const long v0 = gw * vis_per_wave, v1 = v0 + vis_per_wave < vtotal ? v0 + vis_per_wave : vtotal;
    long v = v0;
    for (; v + 4 <= v1; v += 4) {
        const TI* g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) g[u] = dY + ((long)((v + u) / nvis) * N + (int)unmasked[v + u]) * dd;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int e4 = (lane + 64 * c) * 4;
            if (e4 < dd) {
                f32x4 d[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) d[u] = load4(g[u] + e4);
#pragma unroll
                for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(dsrc + (v + u) * dd + e4) = d[u];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[c][q] += (d[0][q] + d[1][q]) + (d[2][q] + d[3][q]);
            }
        }
    }
    for (; v < v1; ++v) {
        const int b = (int)(v / nvis);
        const int pos = (int)unmasked[v];
        const TI* g = dY + ((long)b * N + pos) * dd;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int e4 = (lane + 64 * c) * 4;
            if (e4 < dd) {
                const f32x4 d = load4(g + e4);
                *reinterpret_cast<f32x4*>(dsrc + v * dd + e4) = d;
                acc[c][0] += d[0]; acc[c][1] += d[1]; acc[c][2] += d[2]; acc[c][3] += d[3];
            }
        }
    }
    { UNSH_FLUSH(-1, -1.f) }
#undef UNSH_FLUSH
    __syncthreads();
    for (int i = threadIdx.x; i < PL; i += 256)
        part[(long)blockIdx.x * PL + i] = sm[i] + sm[PL + i] + sm[2 * PL + i] + sm[3 * PL + i];
}

// rows of src [B, N, D] selected by idx[b, j0 + jj] -> dst [B*cnt, D]
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ src, int N, int D, const int64_t* __restrict__ idx,
                                                            int idx_ld, int j0, int cnt, int rows, T* __restrict__ dst) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = row / cnt, jj = row % cnt;
    const long pos = idx[(long)b * idx_ld + j0 + jj];
    const T* s = src + ((long)b * N + pos) * D;
    T* d = dst + (long)row * D;
    for (int e = lane; e < D; e += 64) d[e] = s[e];
}
template <typename T>
__global__ __launch_bounds__(256) void scatter_rows_kernel(const T* __restrict__ src, int N, int D, const int64_t* __restrict__ idx,
                                                             int idx_ld, int j0, int cnt, int rows, T* __restrict__ dst) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = row / cnt, jj = row % cnt;
    const long pos = idx[(long)b * idx_ld + j0 + jj];
    const T* s = src + (long)row * D;
    T* d = dst + ((long)b * N + pos) * D;
    for (int e = lane; e < D; e += 64) d[e] = s[e];
}

// both modality groups of the heads in one launch: row (b, j), j < cnt0 -> group 0 (row b*cnt0 + j), else group 1 (row b*cnt1 + j - cnt0);
// the token of (b, j) is idx[b, j]; rows move as 16-byte pieces (D * sizeof(T) % 16 == 0)
template <typename T>
__global__ __launch_bounds__(256) void gather_rows2_kernel(const T* __restrict__ src, int N, int D, const int64_t* __restrict__ idx, int idx_ld,
                                                             int cnt0, int cnt1, int rows, T* __restrict__ dst0, T* __restrict__ dst1) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = row / (cnt0 + cnt1), j = row % (cnt0 + cnt1);
    const long pos = idx[(long)b * idx_ld + j];
    const uint4* s = reinterpret_cast<const uint4*>(src + ((long)b * N + pos) * D);
    uint4* d = reinterpret_cast<uint4*>(j < cnt0 ? dst0 + ((long)b * cnt0 + j) * D : dst1 + ((long)b * cnt1 + (j - cnt0)) * D);
    const int pieces = D * (int)sizeof(T) / 16;
    for (int p = lane; p < pieces; p += 64) d[p] = s[p];
}
// ... and the adjoint, times the device scalar *scale (null = 1): the upstream gradient of the loss folded into the scatter
template <typename T>
__global__ __launch_bounds__(256) void scatter_rows2_kernel(const T* __restrict__ src0, const T* __restrict__ src1, int N, int D,
                                                              const int64_t* __restrict__ idx, int idx_ld, int cnt0, int cnt1, int rows,
                                                              const float* __restrict__ scale, T* __restrict__ dst) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = row / (cnt0 + cnt1), j = row % (cnt0 + cnt1);
    const long pos = idx[(long)b * idx_ld + j];
    const T* s = j < cnt0 ? src0 + ((long)b * cnt0 + j) * D : src1 + ((long)b * cnt1 + (j - cnt0)) * D;
    T* d = dst + ((long)b * N + pos) * D;
    const float sc = scale ? *scale : 1.f;
    constexpr int E = 16 / (int)sizeof(T);
    const int pieces = D / E;
    for (int p = lane; p < pieces; p += 64) {
        uint4 v = reinterpret_cast<const uint4*>(s)[p];
        if (scale) {
            T* e = reinterpret_cast<T*>(&v);
#pragma unroll
            for (int k = 0; k < E; ++k) e[k] = from_f32<T>(to_f32(e[k]) * sc);
        }
        reinterpret_cast<uint4*>(d)[p] = v;
    }
}

// masked-patch MSE (pretrain_models.py:260-262,327-340): pred [rows, pdpad] f32 vs raw target patches gathered by the
// masked indices; part[G] = w * sum (pred - tgt)^2 ; dpred = 2 w (pred - tgt) in compute type (pad columns zero).
// The target patch is gathered in image order into an LDS copy in patch-vector order (dynamic LDS: WPB x pdpad floats); pred, dpred and
// target_out are then walked as 16-byte pieces (cfg 5, 7680 rows x 2352: 119 us one element at a time -> 104 -> 53 with the patch-order
// walk of round 3 -> this form).
template <typename T>
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ pred, int pdpad, PatchGroup pg,
                                                    const int64_t* __restrict__ idx, int idx_ld, int j0, int cnt, int rows, float w,
                                                    float* __restrict__ part, T* __restrict__ dpred, float* __restrict__ target_out, int vec) {
    extern __shared__ __attribute__((aligned(16))) float mbuf[];
    __shared__ float red[WPB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pd = pg.C * pg.P * pg.P;
    const PatchDiv dv = patch_div(pg);
    float* my = mbuf + (long)wave * pdpad;
    float acc = 0.f;
    for (int r0 = blockIdx.x * WPB; r0 < rows; r0 += gridDim.x * WPB) {       // block-uniform trip count (barriers inside)
        const int row = r0 + wave;
        const bool live = row < rows;
        if (live) {
            int b, s, ph, pw, local;
            patch_locate(pg, idx, idx_ld, j0, cnt, row, b, s, ph, pw, local);
            const float* base = patch_base(pg, b, s, ph, pw);
            for (int tb = 0; tb < pd; tb += 512) {                            // eight gathers in flight per lane
                const int t0 = tb + lane;
                float tv[8];
                int e[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (tb + 64 * u >= pd) break;                              // wave-uniform
                    const int t = t0 + 64 * u;
                    tv[u] = base[patch_off_t(pg, dv, min(t, pd - 1), e[u])];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (tb + 64 * u >= pd) break;
                    if (t0 + 64 * u < pd) my[e[u]] = tv[u];
                }
            }
        }
        __syncthreads();
        if (live) {
            const float* pr = pred + (long)row * pdpad;
            T* dp = dpred + (long)row * pdpad;
            if (vec) {
#pragma unroll 4
                for (int e = 4 * lane; e < pdpad; e += 256) {
                    f32x4 d = {0.f, 0.f, 0.f, 0.f};
                    if (e < pd) {
                        const f32x4 t = *reinterpret_cast<const f32x4*>(my + e);
                        d = load4(pr + e) - t;
                        if (target_out) store4<float>(target_out + (long)row * pd + e, t);
                    }
                    acc += d[0] * d[0];
                    acc += d[1] * d[1];
                    acc += d[2] * d[2];
                    acc += d[3] * d[3];
                    store4<T>(dp + e, d * (2.f * w));
                }
            } else {
                for (int e = lane; e < pdpad; e += 64) {
                    const float d = e < pd ? pr[e] - my[e] : 0.f;
                    if (e < pd && target_out) target_out[(long)row * pd + e] = my[e];
                    acc += d * d;
                    dp[e] = from_f32<T>(2.f * w * d);
                }
            }
        }
        __syncthreads();
    }
    acc = wave_sum(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = w * (red[0] + red[1] + red[2] + red[3]);
}

}  // namespace

// =================================================================================================================
static int ln_grid(int M) { return cdiv(M, WPB); }
static bool aligned_to(const void* p, size_t a) { return ((uintptr_t)p & (a - 1)) == 0; }
int m3l_part_blocks(void) {
    static const int nb = [] {
        const char* e = getenv("M3L_PART_BLOCKS");
        const int v = e ? atoi(e) : 1024;
        return v < 64 ? 64 : (v > M3L_MAX_PARTIAL_BLOCKS ? M3L_MAX_PARTIAL_BLOCKS : v);
    }();
    return nb;
}
static int part_grid(long rows) {
    long g = (rows + WPB - 1) / WPB;
    if (g > m3l_part_blocks()) g = m3l_part_blocks();
    return (int)(g < 1 ? 1 : g);
}

int m3l_ln_fwd(int out_dtype, const float* x, int M, int D, const float* gamma, const float* beta, float eps, void* y, float* y32,
               hipStream_t st) {
    M3L_CHECK(M > 0 && D > 0 && D % 4 == 0 && D <= 1024, "ln_fwd: bad shape M=%d D=%d (D must be a multiple of 4, <= 1024)", M, D);
    ProfScope prof("ln_fwd", M, D, out_dtype, (double)M * D * (4.0 + (y ? (out_dtype ? 2 : 4) : 0) + (y32 ? 4 : 0)), st);
#define LN_FWD(TO, C, TX) ln_fwd_kernel<TO, C, TX><<<ln_grid(M), 256, 0, st>>>((const TX*)x, M, D, gamma, beta, eps, (TO*)y, y32)
    if (m3l_call_rb()) {          // the input is a bf16 residual stream (final LayerNorm of a stack)
        if (out_dtype == 1) { if (D <= 256) LN_FWD(bf16, 1, bf16); else if (D <= 512) LN_FWD(bf16, 2, bf16); else LN_FWD(bf16, 4, bf16); }
        else { if (D <= 256) LN_FWD(float, 1, bf16); else if (D <= 512) LN_FWD(float, 2, bf16); else LN_FWD(float, 4, bf16); }
    } else if (out_dtype == 1) {
        if (D <= 256) LN_FWD(bf16, 1, float); else if (D <= 512) LN_FWD(bf16, 2, float); else LN_FWD(bf16, 4, float);
    } else {
        if (D <= 256) LN_FWD(float, 1, float); else if (D <= 512) LN_FWD(float, 2, float); else LN_FWD(float, 4, float);
    }
#undef LN_FWD
    M3L_LAUNCH_CHECK();
    return 0;
}

int m3l_reduce_rows(const float* part, int G, int stride, int count, float* out, int accumulate, hipStream_t st) {
    if (count == 1 && stride == 1)
        reduce_vec_kernel<<<1, 1024, 0, st>>>(part, G, out, accumulate);
    else
        reduce_rows_kernel<<<cdiv(count, 32), 1024, 0, st>>>(part, G, stride, count, out, accumulate);
    M3L_LAUNCH_CHECK();
    return 0;
}

// number of partial rows [G][3 D] one ln_bwd launch writes
int m3l_ln_bwd_blocks(int M) {
    // rows per wave: 4 for large M (fewer partial rows to reduce), down to 1 for the short stacks of cfg 4 / cfg 5 (M = 2 k .. 10 k rows: a
    // wave's rows are a chain of dependent load -> reduce -> load -> reduce -> store steps of ~3 us each; env M3L_LN_ROWS_PER_WAVE forces it)
    static const int forced = getenv("M3L_LN_ROWS_PER_WAVE") ? atoi(getenv("M3L_LN_ROWS_PER_WAVE")) : 0;
    const int rpw = forced > 0 ? forced : (M >= 12288 ? 4 : std::max(1, M / 4096));
    int G = cdiv(M, rpw * WPB);                  // up to 8 workgroups per CU
    if (G > 2048) G = 2048;
    return G < 1 ? 1 : G;
}

// dgamma == dbeta == dbias == nullptr: leave the partials in part_ws for a later m3l_reduce_rows_batch
int m3l_ln_bwd(int dy_dtype, const void* dy, const float* x, int M, int D, const float* gamma, float eps, const float* dres,
               float* dx_out, void* dx_t_out, int ct_dtype, float* part_ws, float* dgamma, float* dbeta, float* dbias,
               int accumulate, hipStream_t st) {
    M3L_CHECK(M > 0 && D > 0 && D % 4 == 0 && D <= 1024, "ln_bwd: bad shape M=%d D=%d (D must be a multiple of 4, <= 1024)", M, D);
    const int G = m3l_ln_bwd_blocks(M);
    {
        ProfScope prof("ln_bwd", M, D, dy_dtype,
                       (double)M * D * ((m3l_call_rb() ? 2.0 : 4.0) + (dy_dtype ? 2 : 4) + (dres ? (m3l_call_rb() ? 2 : 4) : 0) + (dx_out ? 4 : 0) + (dx_t_out ? (ct_dtype ? 2 : 4) : 0)), st);
#define LN_BWD(TD, TC, C, TX) ln_bwd_kernel<TD, TC, C, TX><<<G, 256, 0, st>>>((const TD*)dy, (const TX*)x, M, D, gamma, eps, dres, dx_out, (TC*)dx_t_out, part_ws)
#define LN_BWD_C(TD, TC, TX) { if (D <= 256) LN_BWD(TD, TC, 1, TX); else if (D <= 512) LN_BWD(TD, TC, 2, TX); else LN_BWD(TD, TC, 4, TX); }
        if (m3l_call_rb()) {       // x is a bf16 residual stream: dres (if any) is bf16 as well, only dx_t_out is written
            M3L_CHECK(!dx_out && dx_t_out && ct_dtype == 1, "ln_bwd: bf16 residual mode writes the compute-type gradient only");
            if (dy_dtype == 1) LN_BWD_C(bf16, bf16, bf16) else LN_BWD_C(float, bf16, bf16)
        }
        else if (dy_dtype == 1) LN_BWD_C(bf16, bf16, float)
        else if (ct_dtype == 1) LN_BWD_C(float, bf16, float)
        else LN_BWD_C(float, float, float)
#undef LN_BWD_C
#undef LN_BWD
    }
    M3L_LAUNCH_CHECK();
    if (!dgamma && !dbeta && !dbias) return 0;
    ReduceSegs segs = {{dgamma, dbeta, dbias, nullptr}};
    reduce_rows_seg_kernel<<<dim3(cdiv(D, 32), 3), 1024, 0, st>>>(part_ws, G, 3 * D, D, segs, accumulate);
    M3L_LAUNCH_CHECK();
    return 0;
}

int m3l_reduce_rows_batch(const ReduceBatch* b, int accumulate, hipStream_t st) {
    if (b->count <= 0) return 0;
    M3L_CHECK(b->count <= M3L_REDUCE_BATCH_MAX && b->D > 0, "reduce_rows_batch: count=%d D=%d", b->count, b->D);
    ProfScope prof("reduce_batch", b->count, b->D, 0, 0.0, st);
    reduce_rows_batch_kernel<<<dim3(cdiv(b->D, 32), 3, b->count), 1024, 0, st>>>(*b, accumulate);
    M3L_LAUNCH_CHECK();
    return 0;
}

int m3l_reduce_rows_seg3(const float* part, int G, int D, float* out0, float* out1, float* out2, int accumulate, hipStream_t st) {
    ReduceSegs segs = {{out0, out1, out2, nullptr}};
    reduce_rows_seg_kernel<<<dim3(cdiv(D, 32), 3), 1024, 0, st>>>(part, G, 3 * D, D, segs, accumulate);
    M3L_LAUNCH_CHECK();
    return 0;
}

int m3l_colsum(int dtype, const void* Y, int M, int N, int ld, float* part_ws, float* out, int accumulate, hipStream_t st) {
    M3L_CHECK(M > 0 && N > 0, "colsum: bad shape");
    // rows per block: 64 for narrow matrices (one pass of the 256 threads covers 1024 columns), 16 for wide ones so that the launch still
    // has a few hundred blocks (7680 x 2352 at cfg 5: 120 blocks of 64 rows left half the CUs idle)
    int G = cdiv(M, N > 512 ? 16 : 64);
    if (G > m3l_part_blocks()) G = m3l_part_blocks();
    const int rpb = cdiv(M, G);
    G = cdiv(M, rpb);
    ProfScope prof("colsum", M, N, dtype, (double)M * N * (dtype ? 2 : 4), st);
    // vector path: 4-column pieces aligned in Y (ld and base) and in the partial rows (N)
    const int vec_ok = (ld % 4 == 0) && (N % 4 == 0) && (((unsigned long)Y) % 16 == 0) && (((unsigned long)part_ws) % 16 == 0);
    if (dtype == 1)
        colsum_kernel<bf16><<<G, 256, 0, st>>>((const bf16*)Y, M, N, ld, rpb, part_ws, vec_ok);
    else
        colsum_kernel<float><<<G, 256, 0, st>>>((const float*)Y, M, N, ld, rpb, part_ws, vec_ok);
    M3L_LAUNCH_CHECK();
    return m3l_reduce_rows(part_ws, G, N, N, out, accumulate, st);
}

// Weight-copy batching across the modules of one library call chain (the fused step / extractor chain, mae_step.hip): in COLLECT mode a
// module's m3l_prep_weights only records its matrices (and the module returns right after it: m3l_prep_mode() == 1); m3l_prep_flush
// issues them as ONE launch per 64 matrices; in SKIP mode the modules then run with their copies in place.  Thread-local, like the
// other per-call flags (err.hip).  Four launches of 7-16 us spread over the forward become one or two at its start.
static thread_local int g_prep_mode = 0;
static thread_local int g_prep_dtype = -1;
static thread_local std::vector<WeightDesc>* g_prep_list = nullptr;
int m3l_prep_mode(void) { return g_prep_mode; }
void m3l_prep_set_mode(int mode) {
    g_prep_mode = mode;
    if (mode == 1) {
        if (!g_prep_list) g_prep_list = new std::vector<WeightDesc>();
        g_prep_list->clear();
        g_prep_dtype = -1;
    }
}
int m3l_prep_flush(hipStream_t st) {
    const int saved = g_prep_mode;
    g_prep_mode = 0;
    int rc = 0;
    if (g_prep_list && !g_prep_list->empty()) {
        for (size_t i = 0; i < g_prep_list->size() && !rc; i += M3L_WPACK) {
            WeightPack pk;
            memset(&pk, 0, sizeof(pk));
            for (size_t k = i; k < g_prep_list->size() && k < i + M3L_WPACK; ++k) pk.d[pk.count++] = (*g_prep_list)[k];
            rc = m3l_prep_weights(g_prep_dtype, &pk, st);
        }
        g_prep_list->clear();
    }
    g_prep_mode = saved;
    return rc;
}

int m3l_prep_weights(int dtype, const WeightPack* pack_in, hipStream_t st) {
    if (pack_in->count <= 0) return 0;
    M3L_CHECK(pack_in->count <= M3L_WPACK, "prep_weights: %d matrices in one pack (max %d)", pack_in->count, M3L_WPACK);
    if (g_prep_mode == 2) return 0;                 // copies already issued by m3l_prep_flush
    if (g_prep_mode == 1) {
        M3L_CHECK(g_prep_dtype < 0 || g_prep_dtype == dtype, "prep_weights: modules of one chain disagree on the compute type");
        g_prep_dtype = dtype;
        for (int i = 0; i < pack_in->count; ++i) g_prep_list->push_back(pack_in->d[i]);
        return 0;
    }
    WeightPack pack = *pack_in;
    int tiles = 0;
    for (int i = 0; i < pack.count; ++i) {
        const WeightDesc& d = pack.d[i];
        M3L_CHECK(d.ld_dst >= d.cols && d.ld_dstT >= d.rows && d.ld_dst <= ((d.cols + 63) / 64) * 64 && d.ld_dstT <= ((d.rows + 63) / 64) * 64,
                  "prep_weights: padding of matrix %d ([%d, %d] -> [%d, %d]) must stay inside its last 64 x 64 tiles", i, d.rows, d.cols, d.ld_dstT, d.ld_dst);
        pack.tile0[i] = tiles;
        tiles += ((d.rows + 63) / 64) * ((d.cols + 63) / 64);
    }
    pack.tile0[pack.count] = tiles;
    if (tiles == 0) return 0;
    ProfScope prof("prep_weights", pack.count, dtype, 0, 0.0, st);
    if (dtype == 1)
        prep_weights_kernel<bf16><<<tiles, 256, 0, st>>>(pack);
    else
        prep_weights_kernel<float><<<tiles, 256, 0, st>>>(pack);
    M3L_LAUNCH_CHECK();
    return 0;
}

int m3l_axpy_t(int dtype, const float* x, const void* o, long count, float* out, hipStream_t st) {
    if (dtype == 1)
        axpy_t_kernel<bf16><<<cdiv(count, 256), 256, 0, st>>>(x, (const bf16*)o, count, out);
    else
        axpy_t_kernel<float><<<cdiv(count, 256), 256, 0, st>>>(x, (const float*)o, count, out);
    M3L_LAUNCH_CHECK();
    return 0;
}

// 8 elements per thread (16-byte bf16 side): the boundary casts of the bf16 residual mode (count % 8 == 0, 16-byte aligned)
__global__ void cast8_f32_bf16_kernel(const float* __restrict__ x, long n8, bf16* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + 8 * i), b = *reinterpret_cast<const f32x4*>(x + 8 * i + 4);
    bf16x8 o;
    o[0] = (bf16)a[0]; o[1] = (bf16)a[1]; o[2] = (bf16)a[2]; o[3] = (bf16)a[3];
    o[4] = (bf16)b[0]; o[5] = (bf16)b[1]; o[6] = (bf16)b[2]; o[7] = (bf16)b[3];
    *reinterpret_cast<bf16x8*>(out + 8 * i) = o;
}
__global__ void cast8_bf16_f32_kernel(const bf16* __restrict__ x, long n8, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + 8 * i);
    *reinterpret_cast<f32x4*>(out + 8 * i) = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    *reinterpret_cast<f32x4*>(out + 8 * i + 4) = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
}
int m3l_cast_bf16_f32(const void* x, long count, float* out, hipStream_t st) {
    M3L_CHECK(count % 8 == 0, "cast_bf16_f32: count %ld must be a multiple of 8", count);
    cast8_bf16_f32_kernel<<<cdiv(count / 8, 256), 256, 0, st>>>((const bf16*)x, count / 8, out);
    M3L_LAUNCH_CHECK();
    return 0;
}

int m3l_cast_f32(int dtype, const float* x, long count, void* out, hipStream_t st) {
    ProfScope prof("cast", count, dtype, 0, (double)count * (4.0 + (dtype ? 2 : 4)), st);
    if (dtype == 1 && count % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0)
        cast8_f32_bf16_kernel<<<cdiv(count / 8, 256), 256, 0, st>>>(x, count / 8, (bf16*)out);
    else if (dtype == 1)
        cast_kernel<bf16><<<cdiv(count, 256), 256, 0, st>>>(x, count, (bf16*)out);
    else
        cast_kernel<float><<<cdiv(count, 256), 256, 0, st>>>(x, count, (float*)out);
    M3L_LAUNCH_CHECK();
    return 0;
}

int m3l_im2col(int dtype, const ConvSrc* src, int Bsrc, int Ci, int H, int W, int KH, int S, int P, int OH, int OW, int Kpad, void* col,
               hipStream_t st) {
    const int Btot = src->nchw ? Bsrc * src->nsrc : Bsrc;
    const long total = (long)Btot * OH * OW * Kpad;
    if (dtype == 1 && KH == 4 && Kpad == Ci * 16) {
        const long pairs = (long)Btot * OH * OW * Ci;
        im2col4_bf16_kernel<<<cdiv(pairs, 256), 256, 0, st>>>(*src, Bsrc, Ci, H, W, S, P, OH, OW, pairs, (bf16*)col);
        M3L_LAUNCH_CHECK();
        return 0;
    }
    if (dtype == 1)
        im2col_kernel<bf16><<<cdiv(total, 256), 256, 0, st>>>(*src, Bsrc, Ci, H, W, KH, S, P, OH, OW, Kpad, total, (bf16*)col);
    else
        im2col_kernel<float><<<cdiv(total, 256), 256, 0, st>>>(*src, Bsrc, Ci, H, W, KH, S, P, OH, OW, Kpad, total, (float*)col);
    M3L_LAUNCH_CHECK();
    return 0;
}

int m3l_col2im_relu(int dtype, const void* dcol, int Btot, int Ci, int H, int W, int KH, int S, int P, int OH, int OW, int Kpad,
                    const void* act, void* dX, hipStream_t st) {
    const long total = (long)Btot * H * W * Ci;
    if (dtype == 1)
        col2im_relu_kernel<bf16><<<cdiv(total, 256), 256, 0, st>>>((const bf16*)dcol, Btot, Ci, H, W, KH, S, P, OH, OW, Kpad, (const bf16*)act, (bf16*)dX);
    else
        col2im_relu_kernel<float><<<cdiv(total, 256), 256, 0, st>>>((const float*)dcol, Btot, Ci, H, W, KH, S, P, OH, OW, Kpad, (const float*)act, (float*)dX);
    M3L_LAUNCH_CHECK();
    return 0;
}

int k_tokens_assemble(const float* img_tok, const float* tac_tok, int B, int D, int n_img, int n_tac, int k, const float* mod,
                        const float* pos_img, const float* pos_tac, float* tokens, hipStream_t st) {
    const long rows = (long)B * (n_img + k * n_tac);
    tokens_assemble_kernel<<<cdiv(rows, WPB), 256, 0, st>>>(img_tok, tac_tok, B, D, n_img, n_tac > 0 ? n_tac : 1, k, mod, pos_img, pos_tac, tokens);
    M3L_LAUNCH_CHECK();
    return 0;
}

int k_tokens_assemble_bwd(const float* dtok, int B, int D, int n_img, int n_tac, int k, float* d_img, float* d_tac, float* part_ws,
                            float* dmod, int accumulate, hipStream_t st) {
    const long rows = (long)B * (n_img + k * n_tac);
    const int PL = (1 + k) * D;
    const int G = part_grid(rows);
    tokens_assemble_bwd_kernel<<<G, 256, WPB * PL * sizeof(float), st>>>(dtok, B, D, n_img, n_tac > 0 ? n_tac : 1, k, d_img, d_tac, part_ws);
    M3L_LAUNCH_CHECK();
    reduce_rows_kernel<<<cdiv(PL, 32), 1024, 0, st>>>(part_ws, G, PL, PL, dmod, accumulate);
    M3L_LAUNCH_CHECK();
    return 0;
}

int m3l_adam_flat(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd, int step,
                  float gscale, hipStream_t st) {
    M3L_CHECK(n > 0 && step >= 1, "adam: n=%ld step=%d", n, step);
    const float bc1 = 1.0f - powf(b1, (float)step);
    const float bc2_sqrt = sqrtf(1.0f - powf(b2, (float)step));
    adam_flat_kernel<0><<<cdiv(cdiv(n, 4), 256), 256, 0, st>>>(p, const_cast<float*>(g), m, v, n, lr, b1, b2, eps, wd, bc1, bc2_sqrt, nullptr, gscale, nullptr, 0);
    M3L_LAUNCH_CHECK();
    return 0;
}

// AdamW (decoupled weight decay) with an optional fused clip_grad_norm_: max_norm > 0 -> norm_ws (>= 1026 floats) receives the partial
// sums, then [1024] = the clip coefficient and [1025] = the gradient norm; scale_grads: also leave the clipped gradient in g
int m3l_adamw_flat(float* p, float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd, int step, float gscale,
                   float max_norm, float* norm_ws, int scale_grads, hipStream_t st) {
    M3L_CHECK(n > 0 && step >= 1, "adamw: n=%ld step=%d", n, step);
    M3L_CHECK(max_norm <= 0.f || norm_ws, "adamw: clipping needs the norm workspace");
    const float bc1 = 1.0f - powf(b1, (float)step);
    const float bc2_sqrt = sqrtf(1.0f - powf(b2, (float)step));
    const float* clip = nullptr;
    if (max_norm > 0.f) {
        const int G = (int)std::min<long>(1024, cdiv(cdiv(n, 4), 256));
        gradnorm_part_kernel<<<G, 256, 0, st>>>(g, n, norm_ws);
        M3L_LAUNCH_CHECK();
        gradnorm_final_kernel<<<1, 64, 0, st>>>(norm_ws, G, gscale, max_norm, norm_ws + 1024);
        M3L_LAUNCH_CHECK();
        clip = norm_ws + 1024;
    }
    const float decay = (float)(1.0 - (double)lr * (double)wd);      // as torch: param.mul_(1 - lr * weight_decay), the factor formed in double
    adam_flat_kernel<1><<<cdiv(cdiv(n, 4), 256), 256, 0, st>>>(p, g, m, v, n, lr, b1, b2, eps, decay, bc1, bc2_sqrt, nullptr, gscale, clip,
                                                              scale_grads && (clip || gscale != 1.0f));
    M3L_LAUNCH_CHECK();
    return 0;
}

int m3l_adam_flat_dev(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd,
                      int* step_dev, float* bc_dev, hipStream_t st) {
    M3L_CHECK(n > 0 && step_dev && bc_dev, "adam (device step): n=%ld", n);
    adam_bias_kernel<<<1, 1, 0, st>>>(step_dev, b1, b2, bc_dev);
    M3L_LAUNCH_CHECK();
    adam_flat_kernel<0><<<cdiv(cdiv(n, 4), 256), 256, 0, st>>>(p, const_cast<float*>(g), m, v, n, lr, b1, b2, eps, wd, 1.0f, 1.0f, bc_dev, 1.0f, nullptr, 0);
    M3L_LAUNCH_CHECK();
    return 0;
}

int m3l_scale_by_dev(int dtype, const void* x, long count, const float* scale_dev, void* out, hipStream_t st) {
    if (dtype == 1)
        scale_by_dev_kernel<bf16><<<cdiv(count, 256), 256, 0, st>>>((const bf16*)x, count, scale_dev, (bf16*)out);
    else
        scale_by_dev_kernel<float><<<cdiv(count, 256), 256, 0, st>>>((const float*)x, count, scale_dev, (float*)out);
    M3L_LAUNCH_CHECK();
    return 0;
}

int m3l_mask_scale(int mode, const float* src, const void* ref, float scale, long count, float* out, hipStream_t st) {
    M3L_CHECK((mode == 0 || mode == 1) && count > 0 && src && ref && out, "mask_scale: bad arguments");
    if (mode == 0) mask_scale_kernel<0><<<cdiv(count, 256), 256, 0, st>>>(src, ref, scale, count, out);
    else mask_scale_kernel<1><<<cdiv(count, 256), 256, 0, st>>>(src, ref, scale, count, out);
    M3L_LAUNCH_CHECK();
    return 0;
}
int m3l_vit_tokens(const float* emb, const float* cls, const float* regs, const float* pos, int B, int npatch, int R, int D, float* tok,
                   hipStream_t st) {
    M3L_CHECK(B > 0 && npatch > 0 && R >= 0 && D > 0 && emb && cls && pos && tok && (R == 0 || regs), "vit_tokens: bad arguments");
    vit_tokens_kernel<<<cdiv((long)B * (1 + R + npatch) * D, 256), 256, 0, st>>>(emb, cls, regs, pos, B, npatch, R, D, tok);
    M3L_LAUNCH_CHECK();
    return 0;
}
int m3l_concat2(float* a, int na, float* b2, int nb, int rows, float* cat, int split, hipStream_t st) {
    M3L_CHECK(rows > 0 && na > 0 && nb > 0 && a && cat && (split || b2), "concat2: bad arguments");
    concat2_kernel<<<cdiv((long)rows * (na + nb), 256), 256, 0, st>>>(a, na, b2, nb, rows, cat, split);
    M3L_LAUNCH_CHECK();
    return 0;
}

int m3l_vt_load_launch(const void* image_nhwc, int image_u8, int B, int H, int W, int C, float img_lo, float img_span, float* image_nchw,
                       const void* tactile, int tactile_u8, int th, int tw, int n_sensors, int frame_stack, float tac_lo, float tac_span,
                       float* const* tactile_out, hipStream_t st) {
    if (image_nhwc) {
        M3L_CHECK(img_span != 0.f, "vt_load: empty image normalisation range");
        const long total = (long)B * C * H * W;
        if (image_u8)
            vt_image_kernel<uint8_t><<<cdiv(total, 256), 256, 0, st>>>((const uint8_t*)image_nhwc, B, H, W, C, img_lo, img_span, image_nchw);
        else
            vt_image_kernel<float><<<cdiv(total, 256), 256, 0, st>>>((const float*)image_nhwc, B, H, W, C, img_lo, img_span, image_nchw);
        M3L_LAUNCH_CHECK();
    }
    if (tactile) {
        M3L_CHECK(n_sensors >= 1 && n_sensors <= M3L_MAX_SENSORS, "vt_load: n_sensors=%d", n_sensors);
        M3L_CHECK(tac_span != 0.f, "vt_load: empty tactile normalisation range");
        VtTactileOut o;
        for (int s = 0; s < n_sensors; ++s) o.p[s] = tactile_out[s];
        const long total = (long)B * 3 * n_sensors * frame_stack * th * tw;
        if (tactile_u8)
            vt_tactile_kernel<uint8_t><<<cdiv(total, 256), 256, 0, st>>>((const uint8_t*)tactile, B, th, tw, n_sensors, frame_stack, tac_lo,
                                                                         tac_span, o);
        else
            vt_tactile_kernel<float><<<cdiv(total, 256), 256, 0, st>>>((const float*)tactile, B, th, tw, n_sensors, frame_stack, tac_lo,
                                                                       tac_span, o);
        M3L_LAUNCH_CHECK();
    }
    return 0;
}

int m3l_mask_rank_groups(const MaskRankArgs* args, int B, int64_t* masked, int masked_ld, int64_t* unmasked, int unmasked_ld, hipStream_t st) {
    M3L_CHECK(B > 0 && args->count >= 1 && args->count <= M3L_MAX_SENSORS + 1, "mask_rank: B=%d groups=%d", B, args->count);
    int nmax = 0;
    for (int i = 0; i < args->count; ++i) {
        const MaskRankGroup& g = args->g[i];
        M3L_CHECK(g.n > 0 && g.nm >= 0 && g.nm <= g.n, "mask_rank: bad shape n=%d nm=%d", g.n, g.nm);
        M3L_CHECK(g.n <= 8192, "mask_rank: n=%d too large", g.n);
        nmax = std::max(nmax, g.n);
    }
    const int threads = nmax <= 64 ? 64 : (nmax <= 128 ? 128 : 256);
    mask_rank_kernel<<<dim3(B, args->count), threads, nmax * sizeof(float), st>>>(*args, masked, masked_ld, unmasked, unmasked_ld);
    M3L_LAUNCH_CHECK();
    return 0;
}
int m3l_mask_rank(const float* noise, int B, int n, int nm, int token_offset, int64_t* masked, int masked_ld, int masked_off,
                  int64_t* unmasked, int unmasked_ld, int unmasked_off, hipStream_t st) {
    MaskRankArgs a;
    memset(&a, 0, sizeof(a));
    a.count = 1;
    a.g[0] = MaskRankGroup{noise, n, nm, token_offset, masked_off, unmasked_off};
    return m3l_mask_rank_groups(&a, B, masked, masked_ld, unmasked, unmasked_ld, st);
}

constexpr int PATCH_NV_MAX = 40;
static int check_pg(const PatchGroup& pg) {
    const int pd = pg.C * pg.P * pg.P;
    M3L_CHECK(pg.nsrc >= 1 && pg.nsrc <= M3L_MAX_SENSORS, "patch group: nsrc=%d", pg.nsrc);
    M3L_CHECK(pd <= 64 * PATCH_NV_MAX, "patch dim %d > %d unsupported", pd, 64 * PATCH_NV_MAX);
    M3L_CHECK(pg.H % pg.P == 0 && pg.W % pg.P == 0, "image dims must be divisible by the patch size");
    return 0;
}

int m3l_patch_ln(int dtype, const PatchGroup* pg, const int64_t* idx, int idx_ld, int j0, int cnt, int B, const float* gamma,
                 const float* beta, float eps, void* xn, int pdpad, hipStream_t st) {
    if (check_pg(*pg)) return 1;
    const int rows = B * cnt;
    if (rows == 0) return 0;
    const int pd_ = pg->C * pg->P * pg->P, esz = dtype == 1 ? 2 : 4;
    const bool wide = pd_ > 64 * MAXV, tiny = pd_ <= 256;
    const int vec = pd_ % 4 == 0 && pdpad % 4 == 0 && aligned_to(gamma, 16) && aligned_to(beta, 16) && aligned_to(xn, 4 * esz);
#define PATCH_LN(T, NV) patch_ln_kernel<T, NV><<<ln_grid(rows), 256, 0, st>>>(*pg, idx, idx_ld, j0, cnt, rows, gamma, beta, eps, (T*)xn, pdpad, vec)
    if (dtype == 1) {
        if (wide) PATCH_LN(bf16, PATCH_NV_MAX); else if (tiny) PATCH_LN(bf16, 4); else PATCH_LN(bf16, MAXV);
    } else {
        if (wide) PATCH_LN(float, PATCH_NV_MAX); else if (tiny) PATCH_LN(float, 4); else PATCH_LN(float, MAXV);
    }
#undef PATCH_LN
    M3L_LAUNCH_CHECK();
    return 0;
}

int m3l_patch_ln_bwd(int dtype, const PatchGroup* pg, const int64_t* idx, int idx_ld, int j0, int cnt, int B, float eps,
                     const void* dxn, int pdpad, float* part_ws, float* dgamma, float* dbeta, int accumulate, hipStream_t st) {
    if (check_pg(*pg)) return 1;
    const int rows = B * cnt;
    if (rows == 0) return 0;
    const int pd = pg->C * pg->P * pg->P;
    const int G = part_grid(rows);
    const bool wide = pd > 64 * MAXV, tiny = pd <= 256;
    const int vec = pd % 4 == 0 && pdpad % 4 == 0 && aligned_to(dxn, 4 * (dtype == 1 ? 2 : 4));
#define PATCH_LN_BWD(T, NV) patch_ln_bwd_kernel<T, NV><<<G, 256, 0, st>>>(*pg, idx, idx_ld, j0, cnt, rows, eps, (const T*)dxn, pdpad, part_ws, vec)
    if (dtype == 1) {
        if (wide) PATCH_LN_BWD(bf16, PATCH_NV_MAX); else if (tiny) PATCH_LN_BWD(bf16, 4); else PATCH_LN_BWD(bf16, MAXV);
    } else {
        if (wide) PATCH_LN_BWD(float, PATCH_NV_MAX); else if (tiny) PATCH_LN_BWD(float, 4); else PATCH_LN_BWD(float, MAXV);
    }
#undef PATCH_LN_BWD
    M3L_LAUNCH_CHECK();
    ReduceSegs segs = {{dgamma, dbeta, nullptr, nullptr}};      // both halves of the [G][2 pd] slab in one launch
    reduce_rows_seg_kernel<<<dim3(cdiv(pd, 32), 2), 1024, 0, st>>>(part_ws, G, 2 * pd, pd, segs, accumulate);
    M3L_LAUNCH_CHECK();
    return 0;
}

int m3l_embed_finalize(const float* E, int D, const PatchGroup* pg, const int64_t* idx, int idx_ld, int j0, int cnt, int B,
                       const float* gamma, const float* beta, float eps, const float* mod, int mod0, const float* pos,
                       float* tokens, int L, hipStream_t st) {
    const int rows = B * cnt;
    if (rows == 0) return 0;
    M3L_CHECK(D <= 64 * MAXV, "embed_finalize: D=%d too large", D);
    embed_finalize_kernel<<<ln_grid(rows), 256, 0, st>>>(E, rows, D, *pg, idx, idx_ld, j0, cnt, gamma, beta, eps, mod, mod0, pos, tokens, L);
    M3L_LAUNCH_CHECK();
    return 0;
}

int m3l_embed_finalize_bwd(int dtype, const float* dtok, int L, const float* E, int D, const PatchGroup* pg, const int64_t* idx,
                           int idx_ld, int j0, int cnt, int B, const float* gamma, float eps, void* dE, float* part_ws,
                           float* dgamma, float* dbeta, float* dmod, int mod0, int accumulate, hipStream_t st) {
    const int rows = B * cnt;
    if (rows == 0) return 0;
    const int nslot = pg->nsrc;
    const int PL = (2 + nslot) * D;
    const int G = part_grid(rows);
    if (dtype == 1)
        embed_finalize_bwd_kernel<bf16><<<G, 256, WPB * PL * sizeof(float), st>>>(dtok, L, E, rows, D, *pg, idx, idx_ld, j0, cnt, gamma, eps, (bf16*)dE, part_ws, nslot);
    else
        embed_finalize_bwd_kernel<float><<<G, 256, WPB * PL * sizeof(float), st>>>(dtok, L, E, rows, D, *pg, idx, idx_ld, j0, cnt, gamma, eps, (float*)dE, part_ws, nslot);
    M3L_LAUNCH_CHECK();
    if (nslot <= 2) {                                           // dgamma | dbeta | dmod rows: equal-width segments, one launch
        float* dm = dmod + (long)mod0 * D;
        ReduceSegs segs = {{dgamma, dbeta, dm, nslot > 1 ? dm + D : nullptr}};
        reduce_rows_seg_kernel<<<dim3(cdiv(D, 32), 2 + nslot), 1024, 0, st>>>(part_ws, G, PL, D, segs, accumulate);
    } else {
        reduce_rows_kernel<<<cdiv(D, 32), 1024, 0, st>>>(part_ws, G, PL, D, dgamma, accumulate);
        reduce_rows_kernel<<<cdiv(D, 32), 1024, 0, st>>>(part_ws + D, G, PL, D, dbeta, accumulate);
        reduce_rows_kernel<<<cdiv(nslot * D, 32), 1024, 0, st>>>(part_ws + 2 * D, G, PL, nslot * D, dmod + (long)mod0 * D, accumulate);
    }
    M3L_LAUNCH_CHECK();
    return 0;
}

int k_unshuffle_fwd(const float* src, const float* mask_token, const int64_t* unmasked, int nvis, const int64_t* masked,
                      int nmask, int B, int dd, int n_img, int n_tac, const float* dmod, const float* pos_img, const float* pos_tac,
                      float* dec_in, hipStream_t st) {
    const long rows = (long)B * (nvis + nmask);
    M3L_CHECK(dd % 4 == 0, "unshuffle_fwd: dd=%d must be a multiple of 4", dd);
    if (m3l_call_io())      // the decoder runs the bf16 residual stream and takes its input as bf16 (fused step)
        unshuffle_fwd_kernel<bf16><<<cdiv(rows, WPB), 256, 0, st>>>(src, mask_token, unmasked, nvis, masked, nmask, B, dd, n_img, n_tac > 0 ? n_tac : 1, dmod,
                                                                     pos_img, pos_tac, reinterpret_cast<bf16*>(dec_in));
    else
        unshuffle_fwd_kernel<float><<<cdiv(rows, WPB), 256, 0, st>>>(src, mask_token, unmasked, nvis, masked, nmask, B, dd, n_img, n_tac > 0 ? n_tac : 1, dmod,
                                                                      pos_img, pos_tac, dec_in);
    M3L_LAUNCH_CHECK();
    return 0;
}

int k_unshuffle_bwd(const float* dY, const int64_t* unmasked, int nvis, const int64_t* masked, int nmask, int B, int dd, int n_img,
                      int n_tac, int nmod, float* dsrc, float* part_ws, float* dmask_token, float* ddmod, int accumulate,
                      hipStream_t st) {
    const long rows = (long)B * (nvis + nmask);
    const int PL = (1 + nmod) * dd;
    M3L_CHECK(dd % 4 == 0 && dd <= 1024, "unshuffle_bwd: dd=%d must be a multiple of 4, <= 1024", dd);
    (void)masked;
    int rpw = 16;                                       // rows per wave; fewer waves than M3L_MAX_PARTIAL_BLOCKS workgroups
    while (cdiv(cdiv(rows, rpw), WPB) > m3l_part_blocks()) rpw *= 2;
    const int G = cdiv(cdiv(rows, rpw), WPB);
    const int vpw = cdiv((long)B * nvis, (long)G * WPB);
    const int N = nvis + nmask, nt = n_tac > 0 ? n_tac : 1;
    const size_t lds = WPB * PL * sizeof(float);
#define UNSH_BWD(C, TI) unshuffle_bwd_kernel<C, TI><<<G, 256, lds, st>>>((const TI*)dY, unmasked, nvis, B, N, dd, n_img, nt, nmod, rpw, vpw, dsrc, part_ws)
    if (m3l_call_io()) {      // the gradient of the decoder's input arrives as bf16 (fused step, bf16 residual stream)
        if (dd <= 256) UNSH_BWD(1, bf16); else if (dd <= 512) UNSH_BWD(2, bf16); else UNSH_BWD(4, bf16);
    } else {
        if (dd <= 256) UNSH_BWD(1, float); else if (dd <= 512) UNSH_BWD(2, float); else UNSH_BWD(4, float);
    }
#undef UNSH_BWD
    M3L_LAUNCH_CHECK();
    if (nmod <= 3) {                                            // mask token + every modality row: equal-width segments, one launch
        ReduceSegs segs = {{dmask_token, ddmod, nmod > 1 ? ddmod + dd : nullptr, nmod > 2 ? ddmod + 2 * dd : nullptr}};
        reduce_rows_seg_kernel<<<dim3(cdiv(dd, 32), 1 + nmod), 1024, 0, st>>>(part_ws, G, PL, dd, segs, accumulate);
    } else {
        reduce_rows_kernel<<<cdiv(dd, 32), 1024, 0, st>>>(part_ws, G, PL, dd, dmask_token, accumulate);
        reduce_rows_kernel<<<cdiv(nmod * dd, 32), 1024, 0, st>>>(part_ws + dd, G, PL, nmod * dd, ddmod, accumulate);
    }
    M3L_LAUNCH_CHECK();
    return 0;
}

int m3l_gather_rows(int dtype, const void* src, int N, int D, const int64_t* idx, int idx_ld, int j0, int cnt, int B, void* dst,
                    hipStream_t st) {
    const int rows = B * cnt;
    if (rows == 0) return 0;
    if (dtype == 1)
        gather_rows_kernel<bf16><<<ln_grid(rows), 256, 0, st>>>((const bf16*)src, N, D, idx, idx_ld, j0, cnt, rows, (bf16*)dst);
    else
        gather_rows_kernel<float><<<ln_grid(rows), 256, 0, st>>>((const float*)src, N, D, idx, idx_ld, j0, cnt, rows, (float*)dst);
    M3L_LAUNCH_CHECK();
    return 0;
}

int m3l_gather_rows2(int dtype, const void* src, int N, int D, const int64_t* idx, int idx_ld, int cnt0, int cnt1, int B, void* dst0, void* dst1,
                     hipStream_t st) {
    const int rows = B * (cnt0 + cnt1);
    if (rows == 0) return 0;
    M3L_CHECK((D * (dtype ? 2 : 4)) % 16 == 0, "gather_rows2: rows of %d elements are not 16-byte pieces", D);
    if (dtype == 1)
        gather_rows2_kernel<bf16><<<ln_grid(rows), 256, 0, st>>>((const bf16*)src, N, D, idx, idx_ld, cnt0, cnt1, rows, (bf16*)dst0, (bf16*)dst1);
    else
        gather_rows2_kernel<float><<<ln_grid(rows), 256, 0, st>>>((const float*)src, N, D, idx, idx_ld, cnt0, cnt1, rows, (float*)dst0, (float*)dst1);
    M3L_LAUNCH_CHECK();
    return 0;
}
int m3l_scatter_rows2(int dtype, const void* src0, const void* src1, int N, int D, const int64_t* idx, int idx_ld, int cnt0, int cnt1, int B,
                      const float* scale_dev, void* dst, hipStream_t st) {
    const int rows = B * (cnt0 + cnt1);
    if (rows == 0) return 0;
    M3L_CHECK((D * (dtype ? 2 : 4)) % 16 == 0, "scatter_rows2: rows of %d elements are not 16-byte pieces", D);
    if (dtype == 1)
        scatter_rows2_kernel<bf16><<<ln_grid(rows), 256, 0, st>>>((const bf16*)src0, (const bf16*)src1, N, D, idx, idx_ld, cnt0, cnt1, rows, scale_dev,
                                                                  (bf16*)dst);
    else
        scatter_rows2_kernel<float><<<ln_grid(rows), 256, 0, st>>>((const float*)src0, (const float*)src1, N, D, idx, idx_ld, cnt0, cnt1, rows,
                                                                   scale_dev, (float*)dst);
    M3L_LAUNCH_CHECK();
    return 0;
}

int m3l_scatter_rows(int dtype, const void* src, int N, int D, const int64_t* idx, int idx_ld, int j0, int cnt, int B, void* dst,
                     hipStream_t st) {
    const int rows = B * cnt;
    if (rows == 0) return 0;
    if (dtype == 1)
        scatter_rows_kernel<bf16><<<ln_grid(rows), 256, 0, st>>>((const bf16*)src, N, D, idx, idx_ld, j0, cnt, rows, (bf16*)dst);
    else
        scatter_rows_kernel<float><<<ln_grid(rows), 256, 0, st>>>((const float*)src, N, D, idx, idx_ld, j0, cnt, rows, (float*)dst);
    M3L_LAUNCH_CHECK();
    return 0;
}

int m3l_mse(int dtype, const float* pred, int pdpad, const PatchGroup* pg, const int64_t* idx, int idx_ld, int j0, int cnt, int B,
            float weight, float* part_ws, int* nblocks_out, void* dpred, float* target_out, hipStream_t st) {
    if (check_pg(*pg)) return 1;
    const int rows = B * cnt;
    if (nblocks_out) *nblocks_out = 0;
    if (rows == 0) return 0;
    const int pd = pg->C * pg->P * pg->P;
    const float w = weight / ((float)rows * (float)pd);
    const int G = part_grid(rows);
    M3L_CHECK(pdpad >= pd, "mse: pdpad %d < patch dim %d", pdpad, pd);
    const int vec = pd % 4 == 0 && pdpad % 4 == 0 && aligned_to(pred, 16) && aligned_to(dpred, 4 * (dtype == 1 ? 2 : 4)) &&
                    (!target_out || aligned_to(target_out, 16));
    const size_t lds = (size_t)WPB * pdpad * sizeof(float);
    M3L_CHECK(lds <= 64 * 1024, "mse: padded patch dim %d too wide", pdpad);
    if (dtype == 1)
        mse_kernel<bf16><<<G, 256, lds, st>>>(pred, pdpad, *pg, idx, idx_ld, j0, cnt, rows, w, part_ws, (bf16*)dpred, target_out, vec);
    else
        mse_kernel<float><<<G, 256, lds, st>>>(pred, pdpad, *pg, idx, idx_ld, j0, cnt, rows, w, part_ws, (float*)dpred, target_out, vec);
    M3L_LAUNCH_CHECK();
    if (nblocks_out) *nblocks_out = G;
    return 0;
}
