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
    unsigned ra[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) ra[k] = 0xc2b2ae35u * (k + 1) ^ (tid * 40503u) ^ b;
    __syncthreads();
    for (int s = 0; s < spins; ++s) {
#pragma unroll
        for (int k = 0; k < 48; ++k) asm volatile("" : "+v"(r[k]));
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("" : "+a"(ra[k]));      // these live in AGPRs across the spin
        __builtin_amdgcn_s_sleep(64);
    }
    unsigned bad_a = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) bad_a += ra[k] != (0xc2b2ae35u * (k + 1) ^ (tid * 40503u) ^ b);
    if (bad_a) atomicAdd(&err[2], bad_a);
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

// ---- instruction-class victims: each wave runs a fixed chain and writes a checksum; compare with the solo run
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ __launch_bounds__(256) void chain_kernel(float* out, int iters, int spins) {
    const int tid = threadIdx.x, gid = blockIdx.x * 256 + tid;
    float x = 0.001f * (float)((gid * 2654435761u) >> 20), y = 0.37f + 0.0001f * (float)(tid & 63);
    float res = 0.f;
    for (int s = 0; s < spins; ++s) {
        if (KIND == 0) {            // v_mfma_f32_16x16x4_f32 dependent chain
            v4f a = {0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < iters; ++i) a = __builtin_amdgcn_mfma_f32_16x16x4f32(x + 0.01f * i, y, a, 0, 0, 0);
            res += a[0] + a[1] + a[2] + a[3];
        } else if (KIND == 1) {     // v_mfma_f32_32x32x2_f32
            v16f a = {};
            for (int i = 0; i < iters; ++i) a = __builtin_amdgcn_mfma_f32_32x32x2f32(x + 0.01f * i, y, a, 0, 0, 0);
            for (int k = 0; k < 16; ++k) res += a[k];
        } else if (KIND == 2) {     // exp2 + cross-lane shuffles (ds_bpermute / dpp)
            float m = x;
            for (int i = 0; i < iters; ++i) {
                m = __builtin_amdgcn_exp2f(m * 0.5f - 1.0f) + y;
                m += __shfl_xor(m, 16);
                m = fmaxf(m, __shfl_xor(m, 32)) * 0.25f;
            }
            res += m;
        } else {                    // plain VALU fma chain
            float m = x;
            for (int i = 0; i < iters; ++i) m = fmaf(m, 0.999f, y);
            res += m;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    out[gid] = res;
}
extern "C" int chain_launch(int kind, float* out, int blocks, int iters, int spins, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL(chain_kernel<0>, dim3(blocks), dim3(256), 0, st, out, iters, spins);
    else if (kind == 1) hipLaunchKernelGGL(chain_kernel<1>, dim3(blocks), dim3(256), 0, st, out, iters, spins);
    else if (kind == 2) hipLaunchKernelGGL(chain_kernel<2>, dim3(blocks), dim3(256), 0, st, out, iters, spins);
    else hipLaunchKernelGGL(chain_kernel<3>, dim3(blocks), dim3(256), 0, st, out, iters, spins);
    return (int)hipGetLastError();
}
