// Sustained fp32-MFMA rate of the chip under its power limit: every SIMD runs `waves` waves that issue
// independent v_mfma_f32_32x32x2_f32 chains on register operands (no memory traffic).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip ; run: ./mfma_peak [waves_per_simd] [zero|rand]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 * (threadIdx.x % 7 + 1), b = b0 * (threadIdx.x % 5 + 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
}
int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2;
    const bool zero = argc > 2 && !strcmp(argv[2], "zero");
    float* out; hipMalloc(&out, 4);
    const int wgs = 256 * wps, iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, out, iters, zero ? 0.f : 1.0001f, zero ? 0.f : 0.9999f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)wgs * 4 /*waves*/ * iters * 32 /*mfma*/ * 32 * 32 * 2 * 2;
        printf("waves/SIMD=%d %s: %.2f ms  %.1f TFLOP/s\n", wps, zero ? "zeros" : "non-zero", ms, flop / ms / 1e9);
    }
    return 0;
}
