// Micro-benchmark: achievable wave64 f64 FMA issue rate on gfx950 vs waves per SIMD and independent chains per lane
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__global__ void k(double* out, int iters) {
    double a[CH];
    for (int c = 0; c < CH; c++) a[c] = 1.0 + (threadIdx.x + c) * 1e-3;
    const double b = 0.999, d = 1e-9;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++)
#pragma unroll
            for (int c = 0; c < CH; c++) a[c] = __builtin_fma(a[c], b, d);
    }
    double s = 0; for (int c = 0; c < CH; c++) s += a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH> void run(double* d, int wps) {
    int blocks = 256 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<CH><<<blocks, 256>>>(d, 10); hipDeviceSynchronize();
    const int iters = 4000;
    hipEventRecord(e0); k<CH><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fmas = (double)blocks * 4 * iters * 16 * CH;           // wave-level FMA instructions
    double per_simd_per_s = fmas / 1024 / (ms * 1e-3);
    printf("chains %d waves/SIMD %d: %.3f ms  %.3g wave-FMA/s/SIMD = %.2f cycles/FMA @2.4GHz, %.1f TFLOP/s\n", CH, wps, ms,
           per_simd_per_s, 2.4e9 / per_simd_per_s, fmas * 128 / (ms * 1e-3) / 1e12);
}
int main() {
    double* d; hipMalloc(&d, (1 << 22) * 8);
    for (int wps : {1, 2, 4, 8}) { run<1>(d, wps); run<2>(d, wps); run<4>(d, wps); }
    return 0;
}
