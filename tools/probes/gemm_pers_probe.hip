// Prototype of a persistent, software-pipelined bf16 NT GEMM  C[M, N] = A[M, K] W[N, K]^T  for the mid-size shapes of the per-op path
// (cfg 4 / cfg 5: M of 2-10 k rows, K = 384 ... 1536), where the tiled kernel of gemm.hip spends memory time + MFMA time instead of the
// larger of the two (one or two rounds of workgroups in lockstep: load, compute, store).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/gemm_pers_probe tools/probes/gemm_pers_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void* lds_vp;
typedef __attribute__((address_space(1))) const void* gl_vp;

constexpr int PG_CW = 8, PG_DW = 2, PG_THREADS = 64 * (PG_CW + PG_DW);
constexpr int PG_BM = 128, PG_BN = 128, PG_STAGE = (PG_BM + PG_BN) * 128;      // 32 KiB: a 64-deep K tile of both operands

template <int R>
__global__ __launch_bounds__(PG_THREADS) void gemm_nt_pers_kernel(const bf16* __restrict__ A, int lda, const bf16* __restrict__ W, int ldw, int M,
                                                                  int N, int K, bf16* __restrict__ C, int ldc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gx = (N + PG_BN - 1) / PG_BN, gy = (M + PG_BM - 1) / PG_BM, nk = K / 64;
    const int nvirt = 8 * gx * ((gy + 7) / 8), G = gridDim.x, w = blockIdx.x;
    // this workgroup's tiles: virtual blocks w, w + G, ... (XCD-aware: ids congruent mod 8 share an L2; G % 8 == 0)
    auto tile_of = [&](int vb, int& m0, int& n0) -> bool {
        const int rest = vb >> 3, bx = rest % gx, by = (rest / gx) * 8 + (vb & 7);
        m0 = by * PG_BM; n0 = bx * PG_BN;
        return by < gy;
    };
    int ntiles = 0;
    for (int vb = w; vb < nvirt; vb += G) {
        int a, b;
        ntiles += tile_of(vb, a, b) ? 1 : 0;
    }
    const int total = ntiles * nk;              // flattened (tile, k-tile) iterations of this workgroup
    if (total == 0) return;

    if (wave >= PG_CW) {
        // ---- DMA waves: wave 8 stages the A tile, wave 9 the W tile; 16 pieces of 1 KiB (8 rows x 128 B) per stage each
        const bool isA = wave == PG_CW;
        const bf16* base = isA ? A : W;
        const int ld = isA ? lda : ldw, lim = isA ? M : N;
        const int srow = lane >> 3, spc = lane & 7;
        int vb = w, m0 = 0, n0 = 0;
        while (!tile_of(vb, m0, n0)) vb += G;
        int kt = 0;
        const bf16* src[16];
        auto set_tile = [&]() {
            const int r0 = isA ? m0 : n0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = i * 8 + srow;
                src[i] = base + (long)min(r0 + row, lim - 1) * ld + (spc ^ (row & 7)) * 8;
            }
        };
        set_tile();
        auto issue = [&](int it) {
            char* dst = smem + (it % R) * PG_STAGE + (isA ? 0 : PG_BM * 128);
#pragma unroll
            for (int i = 0; i < 16; ++i) __builtin_amdgcn_global_load_lds((gl_vp)(src[i] + kt * 64), (lds_vp)(dst + i * 1024), 16, 0, 0);
            if (++kt == nk) {
                kt = 0;
                vb += G;
                while (vb < nvirt && !tile_of(vb, m0, n0)) vb += G;
                if (vb < nvirt) set_tile();
            }
        };
        for (int it = 0; it < R - 1 && it < total; ++it) issue(it);
        for (int it = 0; it < total; ++it) {
            const int ahead = min(R - 2, total - 1 - it);
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // stage `it` landed; every compute wave is done with stage it - 1
            if (it + R - 1 < total) issue(it + R - 1);
        }
        return;
    }
    // ---- compute waves: 2 (n halves of 64) x 4 (m quarters of 32); transposed product C^T = W A^T: an accumulator tile holds 4 consecutive
    // n of one m per lane, and with the W rows of a 32-row block read in the order 8 (i >> 2) + 4 t + (i & 3) two stacked tiles give a lane
    // 8 consecutive n: one 16-byte store
    const int g = lane >> 4, li = lane & 15;
    const int wn = wave & 1, wm = wave >> 1;
    f32x4 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    int wrow[4], arow[2];
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) wrow[tn] = wn * 64 + (tn >> 1) * 32 + 8 * (li >> 2) + 4 * (tn & 1) + (li & 3);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) arow[tm] = wm * 32 + tm * 16 + li;
    int vb = w, m0 = 0, n0 = 0;
    while (!tile_of(vb, m0, n0)) vb += G;
    int kt = 0;
    for (int it = 0; it < total; ++it) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* As = smem + (it % R) * PG_STAGE;
        const char* Ws = As + PG_BM * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fw[4], fa[2];
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) fw[tn] = *reinterpret_cast<const bf16x8*>(Ws + wrow[tn] * 128 + (((ks * 4 + g) ^ (wrow[tn] & 7)) << 4));
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) fa[tm] = *reinterpret_cast<const bf16x8*>(As + arow[tm] * 128 + (((ks * 4 + g) ^ (arow[tm] & 7)) << 4));
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[tn], fa[tm], acc[tn][tm], 0, 0, 0);
        }
        if (++kt == nk) {
            // epilogue straight from the accumulators (the DMA waves are already staging the next tile)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    const int m = m0 + wm * 32 + tm * 16 + li, n = n0 + wn * 64 + p * 32 + 8 * g;
                    if (m < M && n < N) {
                        bf16x8 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            o[r] = (bf16)acc[2 * p][tm][r];
                            o[4 + r] = (bf16)acc[2 * p + 1][tm][r];
                        }
                        *reinterpret_cast<bf16x8*>(C + (long)m * ldc + n) = o;
                    }
                }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            kt = 0;
            vb += G;
            while (vb < nvirt && !tile_of(vb, m0, n0)) vb += G;
        }
    }
}

__global__ void ref_kernel(const bf16* A, const bf16* W, int M, int N, int K, float* C) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)A[(long)m * K + k] * (float)W[(long)n * K + k];
    C[(long)m * N + n] = s;
}

int main(int argc, char** argv) {
    const int shapes[][3] = {{7232, 1152, 384}, {7232, 384, 384}, {7232, 1536, 384}, {7232, 384, 1536}, {7232, 384, 1152},
                             {9600, 1536, 384}, {9600, 384, 1536}, {5120, 768, 256}, {1920, 1152, 384}, {28928, 576, 192}, {200, 136, 64}};
    int ncu = 256;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    constexpr int R = 4;
    CK(hipFuncSetAttribute((const void*)gemm_nt_pers_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize, R * PG_STAGE));
    for (auto& s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        std::vector<bf16> hA((size_t)M * K), hW((size_t)N * K);
        srand(1);
        for (auto& v : hA) v = (bf16)((rand() % 2001 - 1000) / 1000.f);
        for (auto& v : hW) v = (bf16)((rand() % 2001 - 1000) / 1000.f);
        bf16 *dA, *dW, *dC;
        float* dR;
        CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&dC, (size_t)M * N * 2)); CK(hipMalloc(&dR, (size_t)M * N * 4));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemset(dC, 0, (size_t)M * N * 2));
        const int grid = (ncu / 8) * 8;
        auto launch = [&]() { gemm_nt_pers_kernel<R><<<grid, PG_THREADS, R * PG_STAGE>>>(dA, K, dW, K, M, N, K, dC, N); };
        launch();
        CK(hipDeviceSynchronize());
        ref_kernel<<<dim3((N + 63) / 64, M), 64>>>(dA, dW, M, N, K, dR);
        CK(hipDeviceSynchronize());
        std::vector<bf16> hC((size_t)M * N);
        std::vector<float> hR((size_t)M * N);
        CK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hR.data(), dR, hR.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (size_t i = 0; i < hC.size(); ++i) {
            const double d = fabs((double)(float)hC[i] - hR[i]) / (fabs(hR[i]) + 1.0);
            if (d > worst) worst = d;
        }
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 10; ++i) launch();
        CK(hipEventRecord(e0));
        for (int i = 0; i < 100; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 10.0;
        printf("(%6d,%5d,%5d)  %7.2f us  %7.1f TFLOP/s   worst rel err %.2e\n", M, N, K, us, 2.0 * M * N * K / us * 1e-6, worst);
        CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(dC)); CK(hipFree(dR));
    }
    return 0;
}
