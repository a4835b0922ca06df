// Multi-head attention of the ViT encoder/decoder (vit_pytorch Attention.forward: softmax(Q K^T d_h^-0.5) V, d_h = 64)
// for gfx950.  QK^T, PV and the five backward products all run on MFMA 16x16 tiles; K/V (fwd, dQ pass) or Q/dO
// (dK/dV pass) are staged in LDS, the softmax is computed inside one wavefront.
//
// Orientation.  Forward and the dQ pass compute S^T = K Q^T, so a lane owns ONE query column and 8 keys of a
// 32-key tile: the row max/sum is a register reduction + two cross-lane steps (xor 16, 32), the online-softmax
// rescale of O is lane-local, and P^T is already the B operand of O^T = V^T P^T (accumulator-as-operand, KMAP_ACC).
// V^T / K^T operands come from the row-major LDS tile through the hardware transpose read (load_ks).
// The dK/dV pass computes S = Q K^T with the KEY on the lane for the same reason: dV^T = dO^T P, dK^T = Q^T dS.
//
// Layout: qkv [B*n, 3*H*64] as written by the to_qkv GEMM (q | k | v, head-major inside each), o / dO [B*n, H*64].
#include "common.cuh"
#include "kernels.h"

namespace {

template <typename T> struct AtCfg;
template <> struct AtCfg<bf16> { static constexpr int ROW = 80; };   // 160-byte rows: conflict-free tr reads (rows 0..7 -> 8 slots)
template <> struct AtCfg<float> { static constexpr int ROW = 68; };  // 272-byte rows

// stage a [32 x 64] tile (rows r0..r0+31 of a [n x ld] matrix, zero-filled past n) into LDS
template <typename T>
__device__ __forceinline__ void stage_tile(T* dst, const T* src, long ld, int r0, int n, int tid) {
    constexpr int EPC = Chunk<T>::N, CPR = 64 / EPC, TOT = 32 * CPR;
#pragma unroll
    for (int c = tid; c < TOT; c += 256) {
        const int row = c / CPR, cc = c % CPR;
        uint4 v = {0u, 0u, 0u, 0u};
        if (r0 + row < n) v = *reinterpret_cast<const uint4*>(src + (long)(r0 + row) * ld + cc * EPC);
        *reinterpret_cast<uint4*>(dst + row * AtCfg<T>::ROW + cc * EPC) = v;
    }
}

// 4 consecutive columns of one output row as ONE store (8 bytes of bf16 / 16 bytes of f32).  The transposed accumulator layout puts
// consecutive lanes on consecutive ROWS, so every store instruction scatters over 16 rows; four scalar 2-byte stores per
// (lane, 16-column tile) quadruple the number of write requests for nothing.
__device__ __forceinline__ void store4(bf16* p, f32x4 v) {
    bf16x4 pk;
    pk[0] = (bf16)v[0]; pk[1] = (bf16)v[1]; pk[2] = (bf16)v[2]; pk[3] = (bf16)v[3];
    *reinterpret_cast<bf16x4*>(p) = pk;
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ o, float* __restrict__ lse,
                                                         int n, int H, float scale) {
    constexpr int ROW = AtCfg<T>::ROW;
    __shared__ __attribute__((aligned(16))) T Ks[32 * ROW];
    __shared__ __attribute__((aligned(16))) T Vs[32 * ROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    const int b = blockIdx.y / H, hh = blockIdx.y % H;
    const long ld = 3L * H * 64, ldo = (long)H * 64;
    const T* Q = qkv + (long)b * n * ld + hh * 64;
    const T* K = Q + H * 64;
    const T* V = K + H * 64;
    const int q = blockIdx.x * 64 + wave * 16 + li;
    const int qc = min(q, n - 1);
    Frag<T> fq[2];
    fq[0] = load_kc(Q + (long)qc * ld + 8 * g);
    fq[1] = load_kc(Q + (long)qc * ld + 32 + 8 * g);

    f32x4 oacc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) oacc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, lsum = 0.f;

    const int ntile = (n + 31) / 32;
    for (int kt = 0; kt < ntile; ++kt) {
        stage_tile<T>(Ks, K, ld, kt * 32, n, tid);
        stage_tile<T>(Vs, V, ld, kt * 32, n, tid);
        __syncthreads();
        f32x4 s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const Frag<T> fk = load_kc(Ks + (16 * t + li) * ROW + ks * 32 + 8 * g);
                s[t] = mma16(fk, fq[ks], s[t]);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 32 + 16 * t + 4 * g + r;
                const float v = (key < n) ? s[t][r] * scale : -INFINITY;
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
        const Frag<T> fp = acc_to_frag<T>(s[0], s[1]);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const Frag<T> fv = load_ks<KMAP_ACC>(Vs, ROW, 0, 16 * d, lane);
            oacc[d] = mma16(fv, fp, oacc[d]);
        }
        __syncthreads();
    }
    lsum = xor16_sum(lsum);
    lsum = xor32_sum(lsum);
    if (q < n) {
        const float inv = 1.0f / lsum;
        T* orow = o + ((long)b * n + q) * ldo + hh * 64;
#pragma unroll
        for (int d = 0; d < 4; ++d) store4(orow + 16 * d + 4 * g, oacc[d] * inv);
        if (g == 0) lse[((long)b * H + hh) * n + q] = m + __logf(lsum);
    }
}

// dQ pass (also produces Dsum[b,h,q] = sum_d dO*O for the dK/dV pass)
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ o,
                                                            const T* __restrict__ dO, const float* __restrict__ lse,
                                                            float* __restrict__ dsum, T* __restrict__ dqkv, int n, int H,
                                                            float scale) {
    constexpr int ROW = AtCfg<T>::ROW;
    __shared__ __attribute__((aligned(16))) T Ks[32 * ROW];
    __shared__ __attribute__((aligned(16))) T Vs[32 * ROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    const int b = blockIdx.y / H, hh = blockIdx.y % H;
    const long ld = 3L * H * 64, ldo = (long)H * 64;
    const T* Q = qkv + (long)b * n * ld + hh * 64;
    const T* K = Q + H * 64;
    const T* V = K + H * 64;
    const int q = blockIdx.x * 64 + wave * 16 + li;
    const int qc = min(q, n - 1);
    const T* dOr = dO + ((long)b * n + qc) * ldo + hh * 64;
    const T* Or = o + ((long)b * n + qc) * ldo + hh * 64;
    Frag<T> fq[2], fdo[2];
    float dpart = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        fq[ks] = load_kc(Q + (long)qc * ld + ks * 32 + 8 * g);
        fdo[ks] = load_kc(dOr + ks * 32 + 8 * g);
        const Frag<T> fo = load_kc(Or + ks * 32 + 8 * g);
#pragma unroll
        for (int j = 0; j < 8; ++j) dpart += to_f32(fdo[ks].v[j]) * to_f32(fo.v[j]);
    }
    dpart = xor16_sum(dpart);
    dpart = xor32_sum(dpart);
    const float Dq = dpart;
    const float lq = lse[((long)b * H + hh) * n + qc];
    if (q < n && g == 0) dsum[((long)b * H + hh) * n + q] = Dq;

    f32x4 dq[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntile = (n + 31) / 32;
    for (int kt = 0; kt < ntile; ++kt) {
        stage_tile<T>(Ks, K, ld, kt * 32, n, tid);
        stage_tile<T>(Vs, V, ld, kt * 32, n, tid);
        __syncthreads();
        f32x4 ds[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const Frag<T> fk = load_kc(Ks + (16 * t + li) * ROW + ks * 32 + 8 * g);
                const Frag<T> fv = load_kc(Vs + (16 * t + li) * ROW + ks * 32 + 8 * g);
                s = mma16(fk, fq[ks], s);
                dp = mma16(fv, fdo[ks], dp);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 32 + 16 * t + 4 * g + r;
                const float p = (key < n) ? __expf(s[r] * scale - lq) : 0.f;
                ds[t][r] = p * (dp[r] - Dq) * scale;
            }
        }
        const Frag<T> fds = acc_to_frag<T>(ds[0], ds[1]);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const Frag<T> fkT = load_ks<KMAP_ACC>(Ks, ROW, 0, 16 * d, lane);
            dq[d] = mma16(fkT, fds, dq[d]);
        }
        __syncthreads();
    }
    if (q < n) {
        T* row = dqkv + ((long)b * n + q) * ld + hh * 64;
#pragma unroll
        for (int d = 0; d < 4; ++d) store4(row + 16 * d + 4 * g, dq[d]);
    }
}

// dK / dV pass: a wave owns 16 keys and sweeps all queries.
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ dO,
                                                             const float* __restrict__ lse, const float* __restrict__ dsum,
                                                             T* __restrict__ dqkv, int n, int H, float scale) {
    constexpr int ROW = AtCfg<T>::ROW;
    __shared__ __attribute__((aligned(16))) T Qs[32 * ROW];
    __shared__ __attribute__((aligned(16))) T Gs[32 * ROW];   // dO tile
    __shared__ float Ls[32], Ds[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    const int b = blockIdx.y / H, hh = blockIdx.y % H;
    const long ld = 3L * H * 64, ldo = (long)H * 64;
    const T* Q = qkv + (long)b * n * ld + hh * 64;
    const T* K = Q + H * 64;
    const T* V = K + H * 64;
    const T* dOb = dO + (long)b * n * ldo + hh * 64;
    const int key = blockIdx.x * 64 + wave * 16 + li;
    const int kc = min(key, n - 1);
    Frag<T> fk[2], fv[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        fk[ks] = load_kc(K + (long)kc * ld + ks * 32 + 8 * g);
        fv[ks] = load_kc(V + (long)kc * ld + ks * 32 + 8 * g);
    }
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        dk[d] = f32x4{0.f, 0.f, 0.f, 0.f};
        dv[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int ntile = (n + 31) / 32;
    for (int qt = 0; qt < ntile; ++qt) {
        stage_tile<T>(Qs, Q, ld, qt * 32, n, tid);
        stage_tile<T>(Gs, dOb, ldo, qt * 32, n, tid);
        if (tid < 32) {
            const int qq = qt * 32 + tid;
            Ls[tid] = (qq < n) ? lse[((long)b * H + hh) * n + qq] : INFINITY;
            Ds[tid] = (qq < n) ? dsum[((long)b * H + hh) * n + qq] : 0.f;
        }
        __syncthreads();
        f32x4 p[2], ds[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const Frag<T> fa = load_kc(Qs + (16 * t + li) * ROW + ks * 32 + 8 * g);
                const Frag<T> fg = load_kc(Gs + (16 * t + li) * ROW + ks * 32 + 8 * g);
                s = mma16(fa, fk[ks], s);
                dp = mma16(fg, fv[ks], dp);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ql = 16 * t + 4 * g + r;
                const float pv = (key < n) ? __expf(s[r] * scale - Ls[ql]) : 0.f;
                p[t][r] = pv;
                ds[t][r] = pv * (dp[r] - Ds[ql]) * scale;
            }
        }
        const Frag<T> fp = acc_to_frag<T>(p[0], p[1]);
        const Frag<T> fds = acc_to_frag<T>(ds[0], ds[1]);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const Frag<T> fgT = load_ks<KMAP_ACC>(Gs, ROW, 0, 16 * d, lane);
            const Frag<T> fqT = load_ks<KMAP_ACC>(Qs, ROW, 0, 16 * d, lane);
            dv[d] = mma16(fgT, fp, dv[d]);
            dk[d] = mma16(fqT, fds, dk[d]);
        }
        __syncthreads();
    }
    if (key < n) {
        T* krow = dqkv + ((long)b * n + key) * ld + H * 64 + hh * 64;
        T* vrow = krow + H * 64;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            store4(krow + 16 * d + 4 * g, dk[d]);
            store4(vrow + 16 * d + 4 * g, dv[d]);
        }
    }
}

}  // namespace

int m3l_attn_fwd(int dtype, const void* qkv, void* o, float* lse, int B, int n, int H, hipStream_t st) {
    M3L_CHECK(dtype == 0 || dtype == 1, "attn_fwd: bad dtype %d", dtype);
    M3L_CHECK(B > 0 && n > 0 && H > 0, "attn_fwd: empty problem B=%d n=%d H=%d", B, n, H);
    dim3 grid(cdiv(n, 64), B * H);
    const float scale = 0.125f;   // dim_head ** -0.5, dim_head = 64
    ProfScope prof("attn_fwd", B, n, H, 4.0 * B * H * (double)n * n * 64, st);
    if (dtype == 1)
        attn_fwd_kernel<bf16><<<grid, 256, 0, st>>>((const bf16*)qkv, (bf16*)o, lse, n, H, scale);
    else
        attn_fwd_kernel<float><<<grid, 256, 0, st>>>((const float*)qkv, (float*)o, lse, n, H, scale);
    M3L_LAUNCH_CHECK();
    return 0;
}

int m3l_attn_bwd(int dtype, const void* qkv, const void* o, const void* dO, const float* lse, float* dsum, void* dqkv, int B,
                 int n, int H, hipStream_t st) {
    M3L_CHECK(dtype == 0 || dtype == 1, "attn_bwd: bad dtype %d", dtype);
    M3L_CHECK(B > 0 && n > 0 && H > 0, "attn_bwd: empty problem B=%d n=%d H=%d", B, n, H);
    dim3 grid(cdiv(n, 64), B * H);
    const float scale = 0.125f;
    ProfScope prof("attn_bwd", B, n, H, 10.0 * B * H * (double)n * n * 64, st);
    if (dtype == 1) {
        attn_bwd_dq_kernel<bf16><<<grid, 256, 0, st>>>((const bf16*)qkv, (const bf16*)o, (const bf16*)dO, lse, dsum, (bf16*)dqkv, n, H, scale);
        attn_bwd_dkv_kernel<bf16><<<grid, 256, 0, st>>>((const bf16*)qkv, (const bf16*)dO, lse, dsum, (bf16*)dqkv, n, H, scale);
    } else {
        attn_bwd_dq_kernel<float><<<grid, 256, 0, st>>>((const float*)qkv, (const float*)o, (const float*)dO, lse, dsum, (float*)dqkv, n, H, scale);
        attn_bwd_dkv_kernel<float><<<grid, 256, 0, st>>>((const float*)qkv, (const float*)dO, lse, dsum, (float*)dqkv, n, H, scale);
    }
    M3L_LAUNCH_CHECK();
    return 0;
}
