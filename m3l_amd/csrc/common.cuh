// Shared device-side building blocks for the gfx950 (MI355X / CDNA4) kernels of m3l_amd.
//
// One abstraction carries every matrix product in this library, for both compute types:
//
//   Frag<T>   = the 8 k-elements one lane feeds to a 16x16 MFMA "macro step" of depth 32
//               T = bf16 : one v_mfma_f32_16x16x32_bf16   (lane (i = lane&15, g = lane>>4) holds k = kmap(g, j))
//               T = float: eight v_mfma_f32_16x16x4_f32   (sub-step j uses element j of every lane; exact f32)
//   A k-map says which k index element j of lane-group g stands for.  Both operands of a product must use
//   the same map; the sum over k does not care about the order:
//     KMAP_NAT : k = 8 g + j                      (what a 16-byte k-contiguous read delivers)
//     KMAP_ACC : k = 4 g + (j & 3) + 16 (j >> 2)  (what two stacked 16x16 accumulator tiles hold per lane,
//                                                  MI355X guide "An accumulator tile as the next MFMA's operand")
//   Accumulator (C/D) layout of a 16x16 tile: col = lane & 15, row = 4 (lane >> 4) + reg.
//
//   load_kc : operand whose k index is CONTIGUOUS in LDS (row = i, 8 consecutive k)  -> ds_read_b128
//   load_ks : operand whose k index is the ROW index of a row-major LDS tile (k-strided)
//             bf16 -> two ds_read_b64_tr_b16 (hardware transpose read), f32 -> eight ds_read_b32
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define M3L_WAVE 64

enum { KMAP_NAT = 0, KMAP_ACC = 1 };

template <typename T> struct Frag;
template <> struct Frag<bf16> { bf16x8 v; };
template <> struct Frag<float> { float v[8]; };

template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float x) { return (bf16)x; }
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16 x) { return (float)x; }

// D = A * B + C for one 16x16 tile over a macro step of 32 k.
__device__ __forceinline__ f32x4 mma16(const Frag<bf16>& a, const Frag<bf16>& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma16(const Frag<float>& a, const Frag<float>& b, f32x4 c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
    return c;
}

// k-contiguous operand: p -> 8 consecutive k elements of this lane's row (16-byte aligned).
__device__ __forceinline__ Frag<bf16> load_kc(const bf16* p) {
    Frag<bf16> f;
    f.v = *reinterpret_cast<const bf16x8*>(p);
    return f;
}
__device__ __forceinline__ Frag<float> load_kc(const float* p) {
    Frag<float> f;
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
    f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
    return f;
}

// k-strided operand from a row-major LDS tile `t` (row = k, leading dimension ld elements).
// Delivers, for lane (i = lane & 15, g = lane >> 4): element j = t[(k0 + kmap(g, j)) * ld + c0 + i].
// bf16: ds_read_b64_tr_b16 — within each 16-lane group, lane 4q+p supplies the address of block row q,
// columns 4p..4p+3, and lane i receives column i of the 4 rows (guide T10).  EXEC must be all ones.
template <int KMAP>
__device__ __forceinline__ Frag<bf16> load_ks(const bf16* t, int ld, int k0, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int r0 = (KMAP == KMAP_NAT) ? (k0 + 8 * g + q) : (k0 + 4 * g + q);
    const int r1 = (KMAP == KMAP_NAT) ? (r0 + 4) : (r0 + 16);
    typedef __attribute__((address_space(3))) bf16x4* lds_p;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(t + r0 * ld + c0 + 4 * p));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(t + r1 * ld + c0 + 4 * p));
    Frag<bf16> f;
    f.v[0] = lo[0]; f.v[1] = lo[1]; f.v[2] = lo[2]; f.v[3] = lo[3];
    f.v[4] = hi[0]; f.v[5] = hi[1]; f.v[6] = hi[2]; f.v[7] = hi[3];
    return f;
}
template <int KMAP>
__device__ __forceinline__ Frag<float> load_ks(const float* t, int ld, int k0, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    Frag<float> f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = (KMAP == KMAP_NAT) ? (8 * g + j) : (4 * g + (j & 3) + 16 * (j >> 2));
        f.v[j] = t[(k0 + k) * ld + c0 + i];
    }
    return f;
}

// Two stacked accumulator tiles (rows 0..15 and 16..31 of a 32-deep k range) -> operand fragment (KMAP_ACC).
template <typename T> __device__ __forceinline__ Frag<T> acc_to_frag(f32x4 t0, f32x4 t1);
template <> __device__ __forceinline__ Frag<bf16> acc_to_frag<bf16>(f32x4 t0, f32x4 t1) {
    Frag<bf16> f;
    f.v[0] = (bf16)t0[0]; f.v[1] = (bf16)t0[1]; f.v[2] = (bf16)t0[2]; f.v[3] = (bf16)t0[3];
    f.v[4] = (bf16)t1[0]; f.v[5] = (bf16)t1[1]; f.v[6] = (bf16)t1[2]; f.v[7] = (bf16)t1[3];
    return f;
}
template <> __device__ __forceinline__ Frag<float> acc_to_frag<float>(f32x4 t0, f32x4 t1) {
    Frag<float> f;
    f.v[0] = t0[0]; f.v[1] = t0[1]; f.v[2] = t0[2]; f.v[3] = t0[3];
    f.v[4] = t1[0]; f.v[5] = t1[1]; f.v[6] = t1[2]; f.v[7] = t1[3];
    return f;
}

// 16-byte global <-> register chunk of T (8 bf16 or 4 floats)
template <typename T> struct Chunk;
template <> struct Chunk<bf16> { static constexpr int N = 8; };
template <> struct Chunk<float> { static constexpr int N = 4; };

// sum over the 16 lanes of a DPP row (lanes that share lane >> 4), every lane of the row gets the total: four VALU-rate DPP steps, no
// LDS crossbar
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));   // row_mirror
    return v;
}
// Exchange between the lane groups of a wave (lanes i and i ^ 16, i and i ^ 32) by v_permlane16_swap / v_permlane32_swap (gfx950): two
// VALU passes each instead of a ds_bpermute round trip through the LDS crossbar (~100+ cycles, and __shfl_xor(v, 16 / 32) compiles to
// exactly that).  With both operands equal to v the swap leaves {v[i], v[i ^ 16]} (resp. 32) in the two registers, in an order that
// depends on the lane: enough for commutative combinations, which is all the kernels need (LayerNorm / softmax sums and maxima over the
// four 16-lane groups; every lane must be active).  Inline asm: the __builtin_amdgcn_permlane16_swap form is miscompiled by ROCm 7.2's
// hipcc (both results come back as one register: tools/probes/permlane_swap_test.hip); s_nop 1 = the wait states a VALU write of the
// operands needs in front of the swap.
__device__ __forceinline__ float xor16_sum(float v) {
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float xor32_sum(float v) {
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float xor16_max(float v) {
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
__device__ __forceinline__ float xor32_max(float v) {
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
// sum over the whole wave (call with every lane active), every lane gets the total: the four row totals meet through v_readlane (scalar
// registers) instead of a chain of six ds_bpermute round trips through the LDS crossbar (~500 cycles; the one-row-per-wave LayerNorm kernels
// do two to four of these in a row and nothing hides them)
__device__ __forceinline__ float wave_sum(float v) {
    const int r = __builtin_bit_cast(int, row16_sum(v));
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 0)), b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 16));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 32)), d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 48));
    return (a + b) + (c + d);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// GELU (erf form, as torch.nn.GELU()) and its derivative from ONE shared evaluation.  erf by Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7, below fp32 GELU round-off for the parity tolerance): with a = |x|/sqrt(2), t = 1/(1 + p a),
//   erf(a) = 1 - (a1 t + ... + a5 t^5) exp(-a^2),   and exp(-a^2) = exp(-x^2/2) is also the Gaussian pdf of the derivative.
// Cost per element: one v_exp_f32 and one v_rcp_f32 (quarter-rate) + ~10 full-rate ops — the GEMM epilogues evaluate this
// 10^8 times per step and are VALU-bound on it, so no IEEE division (v_div_scale/fmas/fixup = 10 more ops) and no second exp.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& e) {
    const float ax = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    e = __expf(-ax * ax);
    const float y = 1.0f - p * t * e;                      // erf(|x| / sqrt 2)
    cdf = fmaf(0.5f, copysignf(y, x), 0.5f);
}
__device__ __forceinline__ float erf_fast(float x) {
    float cdf, e;
    gelu_parts(x * 1.41421356237309504880f, cdf, e);
    return 2.0f * cdf - 1.0f;
}
__device__ __forceinline__ float gelu_f(float x) {
    float cdf, e;
    gelu_parts(x, cdf, e);
    return x * cdf;
}
__device__ __forceinline__ float gelu_grad_f(float x) {
    float cdf, e;
    gelu_parts(x, cdf, e);
    return fmaf(x * 0.39894228040143267794f, e, cdf);
}


// GELU for the bf16 block kernels: x * sigmoid(a x + b x^3 + c x^5) with (a, b, c) fitted to the erf form (max |error| 2.5e-5 on
// gelu, 1.1e-4 on its derivative: 1/80 and 1/18 of a bf16 half-ulp at 1) — 11 VALU issue slots per value instead of 18, and the
// fused kernels are VALU-bound on it.  The polynomial turns over past |x| = 11, so x is clamped to [-8, 8] inside it (sigmoid
// saturates to 0 / 1 in fp32 well before).  Coefficients below are pre-multiplied by -log2(e) for v_exp_f32 (= exp2).
#define M3L_GF_A 1.59501577f
#define M3L_GF_B 7.40112921e-02f
#define M3L_GF_C (-7.03033579e-04f)
__device__ __forceinline__ float gelu_fast_sig(float x, float& x2) {
    const float xc = __builtin_amdgcn_fmed3f(x, -8.f, 8.f);
    x2 = xc * xc;
    float p = fmaf(-1.4426950408889634f * M3L_GF_C, x2, -1.4426950408889634f * M3L_GF_B);
    p = fmaf(p, x2, -1.4426950408889634f * M3L_GF_A);
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(p * xc));
}
__device__ __forceinline__ float gelu_fast(float x) {
    float x2;
    return x * gelu_fast_sig(x, x2);
}
__device__ __forceinline__ float gelu_grad_fast(float x) {
    float x2;
    const float s = gelu_fast_sig(x, x2);
    float zd = fmaf(5.0f * M3L_GF_C, x2, 3.0f * M3L_GF_B);
    zd = fmaf(zd, x2, M3L_GF_A);
    return fmaf((x * s) * (1.0f - s), zd, s);
}
// the GELU of a compute type: erf form in fp32, the fitted form in bf16 (every bf16 kernel, GEMM epilogues included: at M of a few
// thousand rows the fc1 / d_fc2 epilogues are VALU-bound on it)
template <typename T> __device__ __forceinline__ float gelu_t(float x) { return sizeof(T) == 2 ? gelu_fast(x) : gelu_f(x); }
template <typename T> __device__ __forceinline__ float gelu_grad_t(float x) { return sizeof(T) == 2 ? gelu_grad_fast(x) : gelu_grad_f(x); }

// ---- the residual stream's storage type ------------------------------------------------------------
// float (default) or bf16 (m3l_set_residual_bf16, round 4): x, x1, xout of every layer and the running residual gradient.  The
// arithmetic on them stays fp32 in registers; only what travels through HBM is rounded (once per half layer).
template <typename R> __device__ __forceinline__ f32x4 ld_res4(const R* p);
template <> __device__ __forceinline__ f32x4 ld_res4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 ld_res4<bf16>(const bf16* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ bf16x4 pack_bf16x4(f32x4 v) {
    bf16x4 pk;
    pk[0] = (bf16)v[0]; pk[1] = (bf16)v[1]; pk[2] = (bf16)v[2]; pk[3] = (bf16)v[3];
    return pk;
}
template <typename R> __device__ __forceinline__ void st_res4(R* p, f32x4 v);
template <> __device__ __forceinline__ void st_res4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void st_res4<bf16>(bf16* p, f32x4 v) { *reinterpret_cast<bf16x4*>(p) = pack_bf16x4(v); }
// the value as it will be read back (a layer's LayerNorm must see the x1 that the backward will see)
template <typename R> __device__ __forceinline__ f32x4 round_res(f32x4 v) {
    if (sizeof(R) == 4) return v;
    const bf16x4 pk = pack_bf16x4(v);
    return f32x4{(float)pk[0], (float)pk[1], (float)pk[2], (float)pk[3]};
}

// ---- host side -------------------------------------------------------------------------------------
// residual-stream mode of the launch being issued (set by the transformer plan around its kernels; 0 = fp32, 1 = bf16)
int m3l_call_rb(void);
void m3l_set_call_rb(int rb);
// 1: the stack input / input gradient handed to m3l_transformer_fwd / _bwd, the un-shuffle's output and its backward's input are bf16 too
// (the fused step sets it around the decoder when that stack runs the bf16 residual stream: no boundary casts)
int m3l_call_io(void);
void m3l_set_call_io(int on);
void m3l_set_error(const char* fmt, ...);
#define M3L_CHECK(cond, ...)                 \
    do {                                     \
        if (!(cond)) {                       \
            m3l_set_error(__VA_ARGS__);      \
            return 1;                        \
        }                                    \
    } while (0)
#define M3L_HIP(expr)                                                              \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            m3l_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 2;                                                              \
        }                                                                          \
    } while (0)
#define M3L_LAUNCH_CHECK() M3L_HIP(hipGetLastError())

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
