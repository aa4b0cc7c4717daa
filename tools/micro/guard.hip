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
        } else if (KIND == 4) {     // v_pk_add_f32 with op_sel (the high / low halves of the second operand swapped), what the wave attention
                                    // kernel's bias add compiles to
            typedef float v2f __attribute__((ext_vector_type(2)));
            v2f a = {x, y}, b = {y * 3.0f, x * 5.0f + 1.0f};
            for (int i = 0; i < iters; ++i) {
                asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(a) : "v"(b));
                a *= 0.5f;
            }
            res += a.x + 2.0f * a.y;
        } else if (KIND >= 6 && KIND <= 12) {   // other modifier patterns the compiler emits (all on the same data)
            typedef float v2f __attribute__((ext_vector_type(2)));
            v2f a = {x, y}, b = {y * 3.0f, x * 5.0f + 1.0f}, c = {0.25f, 0.75f};
            for (int i = 0; i < iters; ++i) {
                if (KIND == 6) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a) : "v"(b));                       // broadcast of the low half
                if (KIND == 7) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(a) : "v"(b));                          // both results take b's HIGH half... (hi default 1)
                if (KIND == 8) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(a) : "v"(c));
                if (KIND == 9) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(a) : "v"(c), "v"(b));
                if (KIND == 10) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(a) : "v"(c), "v"(b));
                if (KIND == 11) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a) : "v"(b));
                if (KIND == 12) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[1,0] op_sel_hi:[0,1]" : "+v"(a) : "v"(b));      // swap of the FIRST operand
                a *= 0.5f;
            }
            res += a.x + 2.0f * a.y;
        } else if (KIND == 13 || KIND == 14) {   // 16-bit packed ops with the second source's OP_SEL bit: are they disturbed too?
            unsigned a = 0x3c003800u ^ (gid & 0xff), b = 0x34003000u;      // two f16 / u16 halves
            for (int i = 0; i < iters; ++i) {
                if (KIND == 13) asm volatile("v_pk_add_f16 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(a) : "v"(b));
                if (KIND == 14) asm volatile("v_pk_add_u16 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(a) : "v"(b));
                a = (a & 0x83ff83ffu) | 0x38003800u;
            }
            res += (float)(a & 0xffff) + 3.0f * (float)(a >> 16);
        } else if (KIND == 5) {     // the same without op_sel
            typedef float v2f __attribute__((ext_vector_type(2)));
            v2f a = {x, y}, b = {y * 3.0f, x * 5.0f + 1.0f};
            for (int i = 0; i < iters; ++i) {
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
                a *= 0.5f;
            }
            res += a.x + 2.0f * a.y;
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
    else if (kind == 4) hipLaunchKernelGGL(chain_kernel<4>, dim3(blocks), dim3(256), 0, st, out, iters, spins);
    else if (kind == 5) hipLaunchKernelGGL(chain_kernel<5>, dim3(blocks), dim3(256), 0, st, out, iters, spins);
    else if (kind == 6) hipLaunchKernelGGL(chain_kernel<6>, dim3(blocks), dim3(256), 0, st, out, iters, spins);
    else if (kind == 7) hipLaunchKernelGGL(chain_kernel<7>, dim3(blocks), dim3(256), 0, st, out, iters, spins);
    else if (kind == 8) hipLaunchKernelGGL(chain_kernel<8>, dim3(blocks), dim3(256), 0, st, out, iters, spins);
    else if (kind == 9) hipLaunchKernelGGL(chain_kernel<9>, dim3(blocks), dim3(256), 0, st, out, iters, spins);
    else if (kind == 10) hipLaunchKernelGGL(chain_kernel<10>, dim3(blocks), dim3(256), 0, st, out, iters, spins);
    else if (kind == 11) hipLaunchKernelGGL(chain_kernel<11>, dim3(blocks), dim3(256), 0, st, out, iters, spins);
    else if (kind == 12) hipLaunchKernelGGL(chain_kernel<12>, dim3(blocks), dim3(256), 0, st, out, iters, spins);
    else if (kind == 13) hipLaunchKernelGGL(chain_kernel<13>, dim3(blocks), dim3(256), 0, st, out, iters, spins);
    else if (kind == 14) hipLaunchKernelGGL(chain_kernel<14>, dim3(blocks), dim3(256), 0, st, out, iters, spins);
    else hipLaunchKernelGGL(chain_kernel<3>, dim3(blocks), dim3(256), 0, st, out, iters, spins);
    return (int)hipGetLastError();
}

// ---- micro aggressors: which instruction of the split-3 kernels disturbs a co-resident wave's op_sel'd packed-fp32 add?
typedef __bf16 g_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 g_bf16x2 __attribute__((ext_vector_type(2)));
typedef float g_f32x16 __attribute__((ext_vector_type(16)));
template <int KIND>
__global__ __launch_bounds__(256) void aggressor_kernel(float* out, int iters) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    float x = 1.0f + 0.001f * (float)(gid & 1023), y = 0.5f;
    float res = 0.f;
    if (KIND == 0) {                 // v_cvt_pk_bf16_f32 only
        for (int i = 0; i < iters; ++i) {
            unsigned r;
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
            x = __uint_as_float(r << 16) * 1.0001f + 0.25f;
            y = __uint_as_float(r & 0xffff0000u) * 0.9999f + 0.125f;
        }
        res = x + y;
    } else if (KIND == 1) {          // v_mfma_f32_32x32x16_bf16 only
        g_bf16x8 a, b;
        for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(x + k); b[k] = (__bf16)(y + k); }
        g_f32x16 acc = {};
        for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        for (int k = 0; k < 16; ++k) res += acc[k];
    } else if (KIND == 2) {          // v_mfma_f32_32x32x2_f32 (the fp32-input pipe: the kernels that never disturb)
        g_f32x16 acc = {};
        for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);
        for (int k = 0; k < 16; ++k) res += acc[k];
    } else {                         // packed bf16 VALU conversions through the compiler's own lowering (fp32 -> bf16 vector convert)
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 v = {x, y};
        for (int i = 0; i < iters; ++i) {
            g_bf16x2 h = __builtin_convertvector(v, g_bf16x2);
            v = __builtin_convertvector(h, f2) * 1.0001f + 0.25f;
        }
        res = v.x + v.y;
    }
    out[gid] = res;
}
extern "C" int aggressor_launch(int kind, float* out, int blocks, int iters, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL(aggressor_kernel<0>, dim3(blocks), dim3(256), 0, st, out, iters);
    else if (kind == 1) hipLaunchKernelGGL(aggressor_kernel<1>, dim3(blocks), dim3(256), 0, st, out, iters);
    else if (kind == 2) hipLaunchKernelGGL(aggressor_kernel<2>, dim3(blocks), dim3(256), 0, st, out, iters);
    else hipLaunchKernelGGL(aggressor_kernel<3>, dim3(blocks), dim3(256), 0, st, out, iters);
    return (int)hipGetLastError();
}
