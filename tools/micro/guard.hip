// Debug probe: a victim kernel that holds known patterns in LDS and in VGPRs while other kernels run on the same CUs, then checks them.
// Built by hand: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/micro/libablate_guard.so tools/micro/guard.hip
#include <hip/hip_runtime.h>
#include <cstdint>
__global__ __launch_bounds__(256) void guard_kernel(unsigned* err, int lds_words, int spins) {
    extern __shared__ unsigned lds[];
    const unsigned tid = threadIdx.x, b = blockIdx.x;
    for (int i = tid; i < lds_words; i += 256) lds[i] = 0x9e3779b9u * (i + 1) ^ b;
    unsigned r[48];
#pragma unroll
    for (int k = 0; k < 48; ++k) r[k] = 0x85ebca6bu * (k + 1) ^ (tid * 2654435761u) ^ b;
    __syncthreads();
    for (int s = 0; s < spins; ++s) {
#pragma unroll
        for (int k = 0; k < 48; ++k) asm volatile("" : "+v"(r[k]));
        __builtin_amdgcn_s_sleep(64);
    }
    unsigned bad_l = 0, bad_r = 0;
    for (int i = tid; i < lds_words; i += 256) bad_l += lds[i] != (0x9e3779b9u * (i + 1) ^ b);
#pragma unroll
    for (int k = 0; k < 48; ++k) bad_r += r[k] != (0x85ebca6bu * (k + 1) ^ (tid * 2654435761u) ^ b);
    if (bad_l) atomicAdd(&err[0], bad_l);
    if (bad_r) atomicAdd(&err[1], bad_r);
}
extern "C" int guard_launch(unsigned* err, int blocks, int lds_bytes, int spins, void* stream) {
    hipLaunchKernelGGL(guard_kernel, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, err, lds_bytes / 4, spins);
    return (int)hipGetLastError();
}
