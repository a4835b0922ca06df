// NT GEMM whose column tile spans the WHOLE output row (N == D == 64 * NG), so that the LayerNorm that follows (forward) or
// the LayerNorm backward that consumes the result (backward) runs in the epilogue, on the row while it is still on chip:
//
//   ROWLN_FWD : v = A W^T + bias + res          -> x_out (fp32 residual stream)
//               LN(v) * gamma + beta            -> xn (compute type) [+ optional fp32 copy]
//               (out-proj + LN2 of the same layer; fc2 + LN1 of the NEXT layer / the final norm)
//   ROWLN_BWD : dy = A W^T                       (dgrad through the Linear that follows a LayerNorm: dxn)
//               dx = dres + dLN/dx(dy; x, gamma) -> dx (fp32, may alias dres) and dx_t (compute type)
//               per-workgroup partials [3*D]: dgamma | dbeta | column sums of dx   -> reduced by reduce_rows_seg
//
// This removes the separate ln_fwd / ln_bwd launches and the dxn / pre-norm round trips through HBM
// (reference arithmetic: vit_pytorch Attention / FeedForward pre-norm blocks, see gemm.hip / elementwise.hip).
// Structure: 64-row tiles, 4 waves x (16 rows x D columns), LDS-DMA staging with the same XOR swizzle as gemm_nt_glds.
#include <string.h>

#include "common.cuh"
#include "kernels.h"

namespace {

template <typename TO> __device__ __forceinline__ void st4(TO* p, f32x4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void st4<bf16>(bf16* p, f32x4 v) {
    bf16x4 o;
    o[0] = (bf16)v[0]; o[1] = (bf16)v[1]; o[2] = (bf16)v[2]; o[3] = (bf16)v[3];
    *reinterpret_cast<bf16x4*>(p) = o;
}
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// (row16_sum: common.cuh)
// sum over the 4 lanes that hold the same row (same lane & 15)
__device__ __forceinline__ float col4_sum(float v) {
    v = xor16_sum(v);
    v = xor32_sum(v);
    return v;
}

// The product is accumulated TRANSPOSED (A operand = W rows, B operand = X rows), so that lane (li = lane & 15, g = lane >> 4)
// of wave w ends up with row m = w*16 + li of the output and, in accumulator tile b, its 4 consecutive columns 16 b + 4 g ..+3:
// a whole output row lives in 4 lanes' registers.  The LayerNorm statistics are a register sum + two cross-lane steps, every
// global access of the epilogue is a 16-byte (fp32) / 8-byte (bf16) row segment, and no LDS staging or barrier is needed after
// the K loop (2-stage K ring, 64 KiB of LDS at D = 192).
template <typename T, int NG, int MODE>
__global__ __launch_bounds__(256) void gemm_nt_rowln_kernel(const T* __restrict__ A, int lda, const T* __restrict__ W, int ldw, int M,
                                                              int K, RowLnEpi ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 64, BN = 64 * NG, D = BN, NB = 4 * NG;
    constexpr int EPC = Chunk<T>::N, BKE = 8 * EPC, KSTEPS = BKE / 32;
    constexpr int A_BYTES = BM * 128;
    constexpr int ASEG = 2, BSEG = 2 * NG;          // 1-KiB DMA segments per wave and K-tile
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef __attribute__((address_space(1))) const void* gl_vp;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int m0 = blockIdx.x * BM;

    const int srow = lane >> 3, spc = lane & 7;
    const T* asrc[ASEG];
#pragma unroll
    for (int i = 0; i < ASEG; ++i) {
        const int row = (wave * ASEG + i) * 8 + srow;
        asrc[i] = A + (long)min(m0 + row, M - 1) * lda + (spc ^ (row & 7)) * EPC;
    }
    const T* bsrc[BSEG];
#pragma unroll
    for (int i = 0; i < BSEG; ++i) {
        const int row = (wave * BSEG + i) * 8 + srow;       // < BN == N: every W row exists
        bsrc[i] = W + (long)row * ldw + (spc ^ (row & 7)) * EPC;
    }
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

    f32x4 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BKE;
    constexpr int STAGE = A_BYTES + BN * 128;
    auto stage = [&](int st, int k0) {
        char* base = smem + st * STAGE;
#pragma unroll
        for (int i = 0; i < ASEG; ++i)
            __builtin_amdgcn_global_load_lds((gl_vp)(asrc[i] + k0), (lds_vp)(base + (wave * ASEG + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < BSEG; ++i)
            __builtin_amdgcn_global_load_lds((gl_vp)(bsrc[i] + k0), (lds_vp)(base + A_BYTES + (wave * BSEG + i) * 1024), 16, 0, 0);
    };
    stage(0, 0);
    __syncthreads();                                   // vmcnt(0) + barrier: tile 0 landed
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        if (t + 1 < nk) stage(cur ^ 1, (t + 1) * BKE);
        const char* As = smem + cur * STAGE;
        const char* Bs = As + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const Frag<T> fx = frag(As, wave * 16 + li, ks);          // B operand: columns = output rows m
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const Frag<T> fw = frag(Bs, b * 16 + li, ks);         // A operand: rows = output columns n
                acc[b] = mma16(fw, fx, acc[b]);
            }
        }
        __syncthreads();                               // next tile landed, this one fully read
    }

    // ---- row epilogue straight from the accumulators -------------------------------------------------------
    const long row = (long)m0 + wave * 16 + li;
    const bool ok = row < M;
    const long rowc = ok ? row : (long)M - 1;
    const int c0 = 4 * g;                              // this lane's columns in tile b: 16 b + c0 .. + 3
    if constexpr (MODE == ROWLN_FWD) {
        T* xn = reinterpret_cast<T*>(ep.out_t);
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int n = 16 * b + c0;
            if (ep.bias) acc[b] += ld4(ep.bias + n);
            if (ep.res) acc[b] += ld4(ep.res + rowc * D + n);
            s += (acc[b][0] + acc[b][1]) + (acc[b][2] + acc[b][3]);
        }
        const float mean = col4_sum(s) / D;
        float q = 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const f32x4 d = acc[b] - mean;
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
        const float rstd = rsqrtf(col4_sum(q) / D + ep.eps);
        if (ok) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int n = 16 * b + c0;
                if (ep.x_out) st4<float>(ep.x_out + row * D + n, acc[b]);
                const f32x4 r = (acc[b] - mean) * rstd * ld4(ep.gamma + n) + ld4(ep.beta + n);
                if (xn) st4<T>(xn + row * D + n, r);
                if (ep.out_f32) st4<float>(ep.out_f32 + row * D + n, r);
            }
        }
    } else {
        T* dx_t = reinterpret_cast<T*>(ep.out_t);
        f32x4 xh[NB];
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            xh[b] = ld4(ep.x + rowc * D + 16 * b + c0);
            s += (xh[b][0] + xh[b][1]) + (xh[b][2] + xh[b][3]);
        }
        const float mean = col4_sum(s) / D;
        float q = 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            xh[b] = xh[b] - mean;
            q += (xh[b][0] * xh[b][0] + xh[b][1] * xh[b][1]) + (xh[b][2] * xh[b][2] + xh[b][3] * xh[b][3]);
        }
        const float rstd = rsqrtf(col4_sum(q) / D + ep.eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            xh[b] = xh[b] * rstd;
            // the unfused path rounds dxn to the compute type before the LayerNorm backward: keep the same numerics
            f32x4 d = acc[b];
            d[0] = to_f32(from_f32<T>(d[0])); d[1] = to_f32(from_f32<T>(d[1]));
            d[2] = to_f32(from_f32<T>(d[2])); d[3] = to_f32(from_f32<T>(d[3]));
            if (!ok) d = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[b] = d;                                   // dy (kept for the parameter-gradient partials)
            const f32x4 gd = d * ld4(ep.gamma + 16 * b + c0);
            s1 += (gd[0] + gd[1]) + (gd[2] + gd[3]);
            const f32x4 tt = gd * xh[b];
            s2 += (tt[0] + tt[1]) + (tt[2] + tt[3]);
        }
        s1 = col4_sum(s1) / D;
        s2 = col4_sum(s2) / D;
        // partial sums over this wave's 16 rows (DPP row reduction), then over the 4 waves through LDS (fixed order)
        float* red = reinterpret_cast<float*>(smem);      // [4 waves][3][D]  (the K tiles are dead: last barrier passed)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int n = 16 * b + c0;
            const f32x4 gd = acc[b] * ld4(ep.gamma + n);
            f32x4 r = (gd - s1 - xh[b] * s2) * rstd;
            if (ep.res) r += ld4(ep.res + rowc * D + n);
            if (!ok) r = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok) {
                if (ep.x_out) st4<float>(ep.x_out + row * D + n, r);
                if (dx_t) st4<T>(dx_t + row * D + n, r);
            }
            f32x4 pg = acc[b] * xh[b], pb = acc[b], pc = r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pg[j] = row16_sum(pg[j]);
                pb[j] = row16_sum(pb[j]);
                pc[j] = row16_sum(pc[j]);
            }
            if (li == 0) {
                *reinterpret_cast<f32x4*>(red + (wave * 3 + 0) * D + n) = pg;
                *reinterpret_cast<f32x4*>(red + (wave * 3 + 1) * D + n) = pb;
                *reinterpret_cast<f32x4*>(red + (wave * 3 + 2) * D + n) = pc;
            }
        }
        __syncthreads();
        float* out = ep.part + (long)blockIdx.x * 3 * D;
        for (int j = tid; j < 3 * D; j += 256)
            out[j] = (red[j] + red[3 * D + j]) + (red[6 * D + j] + red[9 * D + j]);
    }
}

constexpr int rowln_lds_bytes(int ng) {
    const int bn = 64 * ng;
    const int stages = 2 * (64 * 128 + bn * 128);
    const int red = 4 * 3 * bn * 4;
    return stages > red ? stages : red;
}

template <typename T, int NG, int MODE>
int launch_rowln(const void* A, int lda, const void* W, int ldw, int M, int K, const RowLnEpi& ep, hipStream_t st) {
    static bool inited = false;
    if (!inited) {
        M3L_HIP(hipFuncSetAttribute((const void*)gemm_nt_rowln_kernel<T, NG, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    rowln_lds_bytes(NG)));
        inited = true;
    }
    gemm_nt_rowln_kernel<T, NG, MODE><<<cdiv(M, 64), 256, rowln_lds_bytes(NG), st>>>((const T*)A, lda, (const T*)W, ldw, M, K, ep);
    M3L_LAUNCH_CHECK();
    return 0;
}

template <typename T, int MODE>
int dispatch_ng(int ng, const void* A, int lda, const void* W, int ldw, int M, int K, const RowLnEpi& ep, hipStream_t st) {
    switch (ng) {
        case 1: return launch_rowln<T, 1, MODE>(A, lda, W, ldw, M, K, ep, st);
        case 2: return launch_rowln<T, 2, MODE>(A, lda, W, ldw, M, K, ep, st);
        case 3: return launch_rowln<T, 3, MODE>(A, lda, W, ldw, M, K, ep, st);
        case 4: return launch_rowln<T, 4, MODE>(A, lda, W, ldw, M, K, ep, st);
    }
    m3l_set_error("gemm_nt_rowln: unsupported row width %d", 64 * ng);
    return 1;
}

}  // namespace

bool m3l_gemm_nt_rowln_supported(int dtype, int N, int K) {
    const int ng = N / 64;
    return N % 64 == 0 && ng >= 1 && ng <= 4 && K % (dtype ? 64 : 32) == 0;      // D <= 256: a row fits 4 lanes' registers
}

int m3l_gemm_nt_rowln(int dtype, int mode, const void* A, int lda, const void* W, int ldw, int M, int N, int K, const RowLnEpi* ep,
                      hipStream_t st) {
    M3L_CHECK(m3l_gemm_nt_rowln_supported(dtype, N, K), "gemm_nt_rowln: unsupported shape N=%d K=%d", N, K);
    M3L_CHECK(M > 0 && lda % 8 == 0 && ldw % 8 == 0, "gemm_nt_rowln: bad M / leading dimensions");
    const double es = dtype ? 2.0 : 4.0;
    const double bytes = (double)M * K * es + (double)N * K * es +
                         (double)M * N * ((ep->res ? 4.0 : 0.0) + (ep->x_out ? 4.0 : 0.0) + (ep->out_t ? es : 0.0) + (ep->out_f32 ? 4.0 : 0.0) +
                                          (mode == ROWLN_BWD ? 4.0 : 0.0));
    ProfScope prof(mode == ROWLN_FWD ? "gemm_nt_lnfwd" : "gemm_nt_lnbwd", M, N, K, 2.0 * M * N * K, st, bytes);
    const int ng = N / 64;
    if (dtype == 1)
        return mode == ROWLN_FWD ? dispatch_ng<bf16, ROWLN_FWD>(ng, A, lda, W, ldw, M, K, *ep, st)
                                 : dispatch_ng<bf16, ROWLN_BWD>(ng, A, lda, W, ldw, M, K, *ep, st);
    return mode == ROWLN_FWD ? dispatch_ng<float, ROWLN_FWD>(ng, A, lda, W, ldw, M, K, *ep, st)
                             : dispatch_ng<float, ROWLN_BWD>(ng, A, lda, W, ldw, M, K, *ep, st);
}
