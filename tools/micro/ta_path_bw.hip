// How many bytes per clock a CU can pull from L2 into registers (the vector-memory / texture path): every wave issues 16-byte-per-lane
// buffer loads (1 KB per wave instruction, fully coalesced) from a buffer that stays L2-resident, eight independent loads in flight per wave.
// DESIGN.md prices the split-3 Winograd kernels against 64 B/clk/CU (MI355X_MICROARCH.md); this measures it on the box.
// Build: hipcc --offload-arch=gfx950 -O3 -o ta_path_bw ta_path_bw.hip ; run: ./ta_path_bw [waves_per_cu] [footprint_KB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ src, unsigned mask16, int iters, unsigned* out) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(src), 0, (mask16 + 1) * 16u, 0x00020000);
    unsigned idx = (blockIdx.x * 256u + threadIdx.x) & mask16;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(r, ((idx + u * 4096u) & mask16) * 16u, 0, 0);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
        idx = (idx + 8u * 4096u + 64u) & mask16;
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}
int main(int argc, char** argv) {
    const int wpc = argc > 1 ? atoi(argv[1]) : 8, kb = argc > 2 ? atoi(argv[2]) : 2048;
    const unsigned n16 = (unsigned)kb * 64u;                 // 16-byte elements; a power of two
    u32x4* src; unsigned* out;
    hipMalloc(&src, (size_t)n16 * 16); hipMalloc(&out, 4);
    hipMemset(src, 1, (size_t)n16 * 16);
    const int wgs = 256 * wpc / 4, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, src, n16 - 1, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)wgs * 256 * iters * 8 * 16;
        printf("waves/CU=%d footprint=%d KB: %.2f ms  %.1f TB/s = %.1f B/clk/CU at 2.4 GHz (%.1f at 2.0 GHz)\n", wpc, kb, ms, bytes / ms / 1e9,
               bytes / (ms * 1e-3) / 256 / 2.4e9, bytes / (ms * 1e-3) / 256 / 2.0e9);
    }
    return 0;
}
