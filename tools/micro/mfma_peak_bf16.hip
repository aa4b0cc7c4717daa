// Sustained bf16-MFMA rate of the chip under its power limit: every SIMD runs `waves` waves that issue independent
// v_mfma_f32_32x32x16_bf16 chains on REGISTER operands (no memory traffic, no other instructions): the ceiling a kernel that does nothing
// but matrix work could reach.  Operands: zeros, or eight different pseudo-random bf16 fragments per side rotated from MFMA to MFMA
// (what a real kernel's operand buses see).  2.5 PFLOP/s = 1024 SIMDs x 1024 FLOP/clk x 2.4 GHz: the printed rate / 2.5 PF x 2.4 GHz is the
// clock the chip sustains.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak_bf16 mfma_peak_bf16.hip ; run: ./mfma_peak_bf16 [waves_per_simd] [zero|rand]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned seed) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 A[8], B[8];
    unsigned s = seed * (threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u);
    for (int u = 0; u < 8; ++u) {
        u32x4 a, b;
        for (int e = 0; e < 4; ++e) {
            // two bf16 per dword with exponents near 1.0 (0x3f80 +- a few steps) and random mantissas: finite, no denormals
            s = s * 1664525u + 1013904223u; a[e] = seed ? ((s & 0x007f007fu) | 0x3f003f00u | ((s >> 9) & 0x00800080u)) : 0u;
            s = s * 1664525u + 1013904223u; b[e] = seed ? ((s & 0x007f007fu) | 0x3f003f00u | ((s >> 9) & 0x80008000u)) : 0u;
        }
        A[u] = __builtin_bit_cast(bf16x8, a);
        B[u] = __builtin_bit_cast(bf16x8, b);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[u], B[(u + 3 * i) & 7], acc[i], 0, 0, 0);
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r];
    if (t == 123.456f) out[0] = t;
}
int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2;
    const bool zero = argc > 2 && !strcmp(argv[2], "zero");
    float* out; hipMalloc(&out, 4);
    const int wgs = 256 * wps, iters = 40000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, out, iters, zero ? 0u : 12345u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)wgs * 4 /*waves*/ * iters * 32 /*mfma*/ * 32 * 32 * 16 * 2;
        printf("waves/SIMD=%d %s: %.2f ms  %.0f TFLOP/s = %.2f GHz-equivalent of the 2.5 PF peak\n", wps, zero ? "zeros" : "random", ms, flop / ms / 1e9,
               flop / ms / 1e9 / 2500.0 * 2.4);
    }
    return 0;
}
