// Per-CU fill rate of an L2-resident weight block: LDS-DMA vs loads into registers, one workgroup per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/l2_fill_probe tools/probes/l2_fill_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void* lds_vp;
typedef __attribute__((address_space(1))) const void* gl_vp;
typedef int v4i __attribute__((ext_vector_type(4)));

// MODE 0: LDS-DMA, DEPTH 1-KiB pieces in flight per wave.  MODE 1: global_load_dwordx4 into registers, DEPTH in flight per wave.
template <int MODE, int DEPTH>
__global__ __launch_bounds__(1024) void fill_kernel(const char* __restrict__ w, long bytes, int reps, int* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long pieces = bytes / 1024;                 // 1 KiB per wave instruction
    v4i acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        for (long p = wave; p < pieces; p += (long)nw * DEPTH) {
            if (MODE == 0) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    const long q = p + (long)d * nw;
                    if (q < pieces)
                        __builtin_amdgcn_global_load_lds((gl_vp)(w + q * 1024 + lane * 16), (lds_vp)(smem + ((wave * DEPTH + d) & 63) * 1024), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                v4i v[DEPTH];
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    const long q = p + (long)d * nw;
                    v[d] = (q < pieces) ? *reinterpret_cast<const v4i*>(w + q * 1024 + lane * 16) : v4i{0, 0, 0, 0};
                }
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
            }
        }
    }
    if (MODE == 0) acc[0] = *reinterpret_cast<int*>(smem + threadIdx.x * 4);
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678) sink[0] = 1;
}

template <int MODE, int DEPTH>
static void run(const char* w, long bytes, int waves, int wgs, int* sink, int lds_bytes) {
    const int reps = 40;
    CK(hipFuncSetAttribute((const void*)fill_kernel<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    fill_kernel<MODE, DEPTH><<<wgs, waves * 64, lds_bytes>>>(w, bytes, 2, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    fill_kernel<MODE, DEPTH><<<wgs, waves * 64, lds_bytes>>>(w, bytes, reps, sink);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    const double per_cu = (double)bytes * reps / (ms * 1e-3) / 1e9;
    printf("%-8s depth %2d waves %2d wgs %4d lds %3d KB: %7.1f us/pass  %6.1f GB/s per WG  %6.2f TB/s total\n", MODE ? "regs" : "lds-dma", DEPTH, waves, wgs,
           lds_bytes >> 10, ms * 1e3 / reps, per_cu, per_cu * wgs / 1e3);
}

int main(int argc, char** argv) {
    const long bytes = argc > 1 ? atol(argv[1]) : 885 * 1024;
    char* w; int* sink;
    CK(hipMalloc(&w, bytes)); CK(hipMemset(w, 1, bytes)); CK(hipMalloc(&sink, 4));
    const int big = 100 << 10;      // > 80 KB: one workgroup per CU
    for (int wgs : {256, 512}) {
        const int lds = wgs == 256 ? big : (70 << 10);
        for (int waves : {4, 8, 16}) {
            run<0, 4>(w, bytes, waves, wgs, sink, lds);
            run<0, 8>(w, bytes, waves, wgs, sink, lds);
            run<1, 4>(w, bytes, waves, wgs, sink, lds);
            run<1, 8>(w, bytes, waves, wgs, sink, lds);
            run<1, 16>(w, bytes, waves, wgs, sink, lds);
        }
    }
    run<0, 8>(w, bytes, 16, 32, sink, big);     // one XCD's worth of workgroups... (placement is round-robin, so 4 per XCD)
    run<1, 8>(w, bytes, 16, 32, sink, big);
    return 0;
}
