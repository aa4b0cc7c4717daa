// HBM streaming rates on MI355X: linear reads / writes at 4, 8 and 16 bytes per lane, and the GEMM-epilogue pattern
// (a wave writes 2 rows x 128 B per instruction, row pitch P) -- what bounds the bf16 Linear epilogues (csrc/linear_bf16.hip).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/hbm_bw.hip -o tools/micro/hbm_bw && tools/micro/hbm_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <typename T> __global__ void k_write(T* p, size_t n, T v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
template <typename T> __global__ void k_read(const T* p, size_t n, T* sink) {
    T acc = {};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        T v = p[i];
        acc.x += v.x;
    }
    if (acc.x == 123.456f) *sink = acc;
}
template <typename T> __global__ void k_copy(const T* a, T* b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// epilogue pattern: workgroup (256 thr) owns 128 rows; per "chunk" each wave w writes rows r, r+4 (lane>>5) x 128 B at column 512 ch + 128 w
__global__ void k_epi(unsigned* out, int rows_total, int pitch_dw, int nchunks) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t m0 = (size_t)blockIdx.x * 128;
    for (int ch = 0; ch < nchunks; ++ch)
        for (int a = 0; a < 4; ++a)
            for (int r = 0; r < 16; ++r) {
                const int ml = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                out[(m0 + ml) * pitch_dw + ch * 128 + w * 32 + (lane & 31)] = 0x3f803f80u;
            }
}
// same bytes, 16 B per lane: a wave writes 2 rows x 512 B per instruction (the whole chunk width of the workgroup)
__global__ void k_epi16(uint4* out, int rows_total, int pitch_q, int nchunks) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t m0 = (size_t)blockIdx.x * 128;
    for (int ch = 0; ch < nchunks; ++ch)
        for (int k = 0; k < 16; ++k) {
            const int ml = w * 32 + k * 2 + (lane >> 5);
            out[(m0 + ml) * pitch_q + ch * 32 + (lane & 31)] = make_uint4(1, 2, 3, 4);
        }
}

// quad-transposed MFMA tile: a wave writes 8 rows x 128 B per instruction (lane quad q -> row, 8 quads -> 128 B of the wave's 64 columns)
__global__ void k_epi_q(uint4* out, int rows_total, int pitch_q, int nchunks) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5, q = li & 3;
    const size_t m0 = (size_t)blockIdx.x * 128;
    for (int ch = 0; ch < nchunks; ++ch)
        for (int a = 0; a < 4; ++a)
            for (int g = 0; g < 4; ++g) {
                const int ml = a * 32 + 8 * g + 4 * lh + q;
                out[(m0 + ml) * pitch_q + ch * 32 + w * 8 + (li >> 2)] = make_uint4(1, 2, 3, 4);
            }
}

// the same stores under the occupancy of the GEMM kernels: dynamic LDS limits the workgroups per CU
__global__ void k_epi_q_lds(uint4* out, int rows_total, int pitch_q, int nchunks) {
    extern __shared__ unsigned char sm[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5, q = li & 3;
    if (rows_total < 0) sm[threadIdx.x] = 1;
    const size_t m0 = (size_t)blockIdx.x * 128;
    for (int ch = 0; ch < nchunks; ++ch)
        for (int a = 0; a < 4; ++a)
            for (int g = 0; g < 4; ++g) {
                const int ml = a * 32 + 8 * g + 4 * lh + q;
                out[(m0 + ml) * pitch_q + ch * 32 + w * 8 + (li >> 2)] = make_uint4(1, 2, 3, 4);
            }
}

// the same through a range-checked buffer resource per workgroup (as csrc/linear_bf16.hip stores), data = a hash of the address
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k_epi_q_buf(unsigned char* out, int rows_total, int pitch_bytes, int nchunks, int nt) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5, q = li & 3;
    const size_t m0 = (size_t)blockIdx.x * 128;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out + m0 * pitch_bytes, 0, (unsigned)(128 * pitch_bytes), 0x00020000);
    for (int ch = 0; ch < nchunks; ++ch)
        for (int a = 0; a < 4; ++a)
            for (int g = 0; g < 4; ++g) {
                const int ml = a * 32 + 8 * g + 4 * lh + q;
                const unsigned off = (unsigned)(ml * pitch_bytes + (ch * 32 + w * 8 + (li >> 2)) * 16);
                const unsigned h = off * 2654435761u + blockIdx.x * 40503u;
                const u32x4 v = {h, h ^ 0x9e3779b9u, h * 3u, h + 12345u};
                if (nt) __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 2);
                else __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0);
            }
}

template <typename F> static float timeit(F f, int it = 10) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < it; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / it * 1e3f;
}

int main() {
    const size_t bytes = (size_t)281600 * 2304 * 2;   // the QKV output: 1.3 GB
    void *A, *B;
    hipMalloc(&A, bytes); hipMalloc(&B, bytes);
    hipMemset(A, 0, bytes); hipMemset(B, 0, bytes);
    float4* sink; hipMalloc(&sink, 64);
    const int G = 256 * 16;
    float us;
    us = timeit([&] { k_write<float4><<<G, 256>>>((float4*)A, bytes / 16, make_float4(1, 2, 3, 4)); });
    printf("write 16 B/lane linear     %8.1f us  %.2f TB/s\n", us, bytes / us / 1e6);
    us = timeit([&] { k_write<float2><<<G, 256>>>((float2*)A, bytes / 8, make_float2(1, 2)); });
    printf("write  8 B/lane linear     %8.1f us  %.2f TB/s\n", us, bytes / us / 1e6);
    us = timeit([&] { k_write<float1><<<G, 256>>>((float1*)A, bytes / 4, make_float1(1)); });
    printf("write  4 B/lane linear     %8.1f us  %.2f TB/s\n", us, bytes / us / 1e6);
    us = timeit([&] { k_read<float4><<<G, 256>>>((const float4*)A, bytes / 16, sink); });
    printf("read  16 B/lane linear     %8.1f us  %.2f TB/s\n", us, bytes / us / 1e6);
    us = timeit([&] { k_copy<float4><<<G, 256>>>((const float4*)A, (float4*)B, bytes / 16); });
    printf("copy  16 B/lane linear     %8.1f us  %.2f TB/s (read + write)\n", us, 2.0 * bytes / us / 1e6);
    us = timeit([&] { k_epi<<<2200, 256>>>((unsigned*)A, 281600, 2304 / 2, 9); });
    printf("epilogue 4 B/lane, 2 rows x 128 B per instr, pitch 4608 B, 9 chunks   %8.1f us  %.2f TB/s\n", us, bytes / us / 1e6);
    us = timeit([&] { k_epi16<<<2200, 256>>>((uint4*)A, 281600, 2304 / 8, 9); });
    printf("epilogue 16 B/lane, 2 rows x 512 B per instr, pitch 4608 B, 9 chunks  %8.1f us  %.2f TB/s\n", us, bytes / us / 1e6);
    us = timeit([&] { k_epi_q<<<2200, 256>>>((uint4*)A, 281600, 2304 / 8, 9); });
    printf("epilogue 16 B/lane, 8 rows x 128 B per instr, pitch 4608 B, 9 chunks  %8.1f us  %.2f TB/s\n", us, bytes / us / 1e6);
    us = timeit([&] { k_epi_q<<<2200, 256>>>((uint4*)A, 281600, 256 / 8, 1); });
    printf("epilogue 16 B/lane, 8 rows x 128 B per instr, pitch 512 B, 1 chunk    %8.1f us  %.2f TB/s\n", us, 281600.0 * 512 / us / 1e6);
    for (int nt : {0, 1}) {
        us = timeit([&] { k_epi_q_buf<<<2200, 256>>>((unsigned char*)A, 281600, 4608, 9, nt); });
        printf("  same, buffer stores, hashed data, nt=%d                               %8.1f us  %.2f TB/s\n", nt, us, bytes / us / 1e6);
    }
    for (int lds : {0, 33 * 1024, 66 * 1024, 130 * 1024}) {
        hipFuncSetAttribute((const void*)k_epi_q_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        us = timeit([&] { k_epi_q_lds<<<2200, 256, lds>>>((uint4*)A, 281600, 2304 / 8, 9); });
        printf("  same, %3d KB of LDS per workgroup (occupancy)                        %8.1f us  %.2f TB/s\n", lds / 1024, us, bytes / us / 1e6);
    }
    us = timeit([&] { k_epi<<<2200, 256>>>((unsigned*)A, 281600, 256 / 2, 1); });
    printf("epilogue 4 B/lane, pitch 512 B (256-column layer), 1 chunk            %8.1f us  %.2f TB/s\n", us, 281600.0 * 512 / us / 1e6);
    us = timeit([&] { k_epi16<<<2200, 256>>>((uint4*)A, 281600, 256 / 8, 1); });
    printf("epilogue 16 B/lane, pitch 512 B, 1 chunk                              %8.1f us  %.2f TB/s\n", us, 281600.0 * 512 / us / 1e6);
    return 0;
}
