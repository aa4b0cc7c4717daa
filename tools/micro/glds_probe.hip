#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* in, unsigned in_bytes, float* out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, in_bytes, 0x00020000);
    // lane l reads 16 B at offset (63-l)*16 (reversed); lanes 5 and 9 are out of range
    unsigned voff = (unsigned)((63 - lane) * 16);
    if (lane == 5 || lane == 9) voff = 0x80000000u;
    for (int i = 0; i < 512; ++i) smem[threadIdx.x * 8 + i % 8] = -1.f;
    __syncthreads();
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + 256), 16, voff, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 8; ++i) out[lane * 8 + i] = smem[lane * 8 + i];
}
int main() {
    std::vector<float> h(256); for (int i = 0; i < 256; ++i) h[i] = i;
    float *d, *o; hipMalloc(&d, 1024); hipMalloc(&o, 2048); hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, 1024u, o);
    std::vector<float> r(512); hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
    printf("err %s\n", hipGetErrorString(hipGetLastError()));
    for (int i = 248; i < 320; ++i) printf("%g ", r[i]); printf("\n");
    return 0;
}
