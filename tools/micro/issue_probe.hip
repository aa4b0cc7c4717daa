// How many plain VALU instructions fit beside back-to-back v_mfma_f32_32x32x16_bf16 on one SIMD, with ONE and with TWO waves per SIMD
// (round 6: what bounds the split-3 Winograd kernels -- 8-12 side instructions per MFMA -- once memory is taken away).
//   mode 0  every wave: [1 MFMA + NV independent v_fma_f32] x iters                    (the kernels' own shape)
//   mode 1  two waves per SIMD, even waves MFMA only, odd waves VALU only (NV per MFMA-equivalent): do separate waves overlap fully?
// Prints cycles per MFMA per SIMD (wall time x the clock GRBM would report is not available here: cycles = s_memtime deltas of wave 0).
// Build: hipcc --offload-arch=gfx950 -O3 -o issue_probe_bin issue_probe.hip ; run: ./issue_probe_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NV, int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, unsigned seed) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 a, b;
    unsigned s = seed * (threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u);
    for (int e = 0; e < 4; ++e) {
        s = s * 1664525u + 1013904223u; a[e] = (s & 0x007f007fu) | 0x3f003f00u;
        s = s * 1664525u + 1013904223u; b[e] = (s & 0x007f007fu) | 0x3f003f00u;
    }
    const bf16x8 A = __builtin_bit_cast(bf16x8, a), B = __builtin_bit_cast(bf16x8, b);
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = (float)(threadIdx.x + i);
    const float c1 = 1.0001f, c2 = 0.5f;
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = MODE == 0 || (wave & 4) == 0;       // mode 1: waves 0-3 (first on each SIMD) MFMA, waves 4-7 VALU
    const bool do_valu = MODE == 0 || (wave & 4) != 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (do_mfma) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[u & 3], 0, 0, 0);
            if (do_valu) {
#pragma unroll
                for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(u * NV + j) & 15]) : "v"(c1), "v"(c2));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float t = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r];
    for (int i = 0; i < 16; ++i) t += v[i];
    if (t == 123.456f) out[0] = t;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NV, int MODE>
void run(int threads, float* out, unsigned long long* cyc) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f; unsigned long long c = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NV, MODE>), dim3(256), dim3(threads), 0, 0, out, cyc, iters, 12345u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); }
    }
    const int wps = threads / 256;
    const double mfma_per_simd = (double)iters * 8 * (MODE == 0 ? wps : 1);
    printf("mode %d  waves/SIMD %d  VALU per MFMA %2d : %7.3f ms   %6.1f ns per MFMA per SIMD   %6.1f wave-0 cycles per own MFMA step (s_memtime @100 MHz x24 -> see note)\n",
           MODE, wps, NV, best, best * 1e6 / mfma_per_simd, (double)c / (iters * 8));
}

int main() {
    float* out; unsigned long long* cyc; hipMalloc(&out, 4); hipMalloc(&cyc, 8);
    run<0, 0>(256, out, cyc); run<2, 0>(256, out, cyc); run<4, 0>(256, out, cyc); run<6, 0>(256, out, cyc); run<8, 0>(256, out, cyc);
    run<10, 0>(256, out, cyc); run<12, 0>(256, out, cyc); run<16, 0>(256, out, cyc);
    run<0, 0>(512, out, cyc); run<2, 0>(512, out, cyc); run<4, 0>(512, out, cyc); run<6, 0>(512, out, cyc); run<8, 0>(512, out, cyc);
    run<10, 0>(512, out, cyc); run<12, 0>(512, out, cyc); run<16, 0>(512, out, cyc);
    run<4, 1>(512, out, cyc); run<8, 1>(512, out, cyc); run<12, 1>(512, out, cyc); run<16, 1>(512, out, cyc); run<24, 1>(512, out, cyc);
    return 0;
}
