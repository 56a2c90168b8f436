// Micro-benchmark: does a wave64 f64 VALU instruction take less time when whole 16-lane quarters of EXEC are off?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, unsigned long long mask, int iters) {
    const int lane = threadIdx.x & 63;
    double a = 1.0 + lane * 1e-3, b = 0.999, c = 1e-9;
    unsigned long long t0 = __builtin_readcyclecounter();
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int j = 0; j < 32; j++) a = __builtin_fma(a, b, c);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[gridDim.x * blockDim.x] = (double)(t1 - t0);
}
int main() {
    double* d; hipMalloc(&d, (1 << 20) * 8 + 8);
    struct { const char* name; unsigned long long m; } cases[] = {
        {"all 64", ~0ull}, {"low 32", 0xffffffffull}, {"low 16", 0xffffull}, {"low 8", 0xffull}, {"lane 0", 1ull},
        {"high 32", 0xffffffff00000000ull}, {"every 4th", 0x1111111111111111ull}, {"every 2nd", 0x5555555555555555ull},
        {"quarters 0,2", 0x0000ffff0000ffffull}, {"17 lanes", 0x1ffffull}};
    for (auto& cs : cases) {
        for (int waves_per_simd : {1, 4}) {
            int blocks = 256 * waves_per_simd;   // 256-thread blocks: 4 waves, one per SIMD
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            k<<<blocks, 256>>>(d, cs.m, 10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            k<<<blocks, 256>>>(d, cs.m, 2000);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double cyc; hipMemcpy(&cyc, d + blocks * 256, 8, hipMemcpyDeviceToHost);
            printf("%-14s waves/SIMD %d: %.3f ms, %.1f cycles per fma (wave 0)\n", cs.name, waves_per_simd, ms, cyc / (2000.0 * 32));
        }
    }
    return 0;
}
