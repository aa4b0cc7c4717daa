// Shared pieces of the split-3 ("x3") Winograd kernels (conv_wino_x3.hip: F(2x2,3x3); conv_wino4_x3.hip: F(4x4,3x3)): vector types, the
// order of the six partial products, the hi / mid / lo split of an A fragment spread over the MFMA steps of a group, compile-time loops.
#pragma once
#include <type_traits>

#include "conv_common.hpp"

namespace {


typedef float x3_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 x3_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 x3_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned x3_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void x3_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// A and B plane of the p-th partial product (0 = hi, 1 = mid, 2 = lo), smallest first
__device__ constexpr int x3_ap(int p) { return p == 0 ? 2 : p == 1 ? 0 : p == 2 ? 1 : p == 3 ? 1 : 0; }
__device__ constexpr int x3_bp(int p) { return p == 0 ? 0 : p == 1 ? 2 : p == 2 ? 1 : p == 3 ? 0 : p == 4 ? 1 : 0; }

// Channel-pair add / subtract, component by component: beside MFMAs (one wave per SIMD) a packed fp32 instruction costs far more than
// the two scalar ones it replaces (v_pk_add_f32: ~+13 cycles each, MI355X_MICROARCH.md "price of one filler"), so these are written --
// and, with -fno-slp-vectorize for the translation units that include this header (build.py), stay -- scalar.
#ifndef AV2X_X3_MB1_OCC
#define AV2X_X3_MB1_OCC 1
#endif
__device__ __forceinline__ x3_f32x2 x3_pk_add(x3_f32x2 a, x3_f32x2 b) {
    x3_f32x2 r;
    r.x = a.x + b.x; r.y = a.y + b.y;
    return r;
}
__device__ __forceinline__ x3_f32x2 x3_pk_sub(x3_f32x2 a, x3_f32x2 b) {
    x3_f32x2 r;
    r.x = a.x - b.x; r.y = a.y - b.y;
    return r;
}

// hi / mid / lo split of the eight values of an A fragment (four pairs q = 0..3), 36 VALU instructions spread over the 12 MFMA
// steps of a group so that no instruction uses the result of the one before it.  Units u = (pair, phase): A(u) = v_cvt_pk_bf16_f32
// + the two words of the converted pair back as fp32 (shift / and), P(u) = the exact remainder (v_pk_add_f32, negated operand),
// F(q) = the last convert.  Step j runs A(x3_sa[j]), P(x3_sp[j]), F(x3_sf[j])  (-1 = none); unit u = 4 (q >> 1) + 2 phase + (q & 1).
__device__ constexpr int x3_sa(int j) { return j == 0 ? 0 : j == 1 ? 1 : j == 2 ? 2 : j == 3 ? 3 : j == 5 ? 4 : j == 6 ? 5 : j == 7 ? 6 : j == 8 ? 7 : -1; }
__device__ constexpr int x3_sp(int j) { return j == 1 ? 0 : j == 2 ? 1 : j == 3 ? 2 : j == 4 ? 3 : j == 6 ? 4 : j == 7 ? 5 : j == 8 ? 6 : j == 9 ? 7 : -1; }
__device__ constexpr int x3_sf(int j) { return j == 4 ? 0 : j == 5 ? 1 : j == 9 ? 2 : j == 10 ? 3 : -1; }
struct X3Split { x3_f32x2 hf[2], r[2]; };
template <int J>
__device__ __forceinline__ void x3_split_step(const x3_f32x2 (&raw)[4], unsigned (&pl)[3][4], X3Split& t) {
    constexpr int ua = x3_sa(J), up = x3_sp(J), qf = x3_sf(J);
    unsigned w = 0;
    if constexpr (ua >= 0) {
        constexpr int q = 2 * (ua >> 2) + (ua & 1), ph = (ua >> 1) & 1;
        const x3_f32x2 src = ph == 0 ? raw[q] : t.r[q & 1];
        w = __builtin_bit_cast(unsigned, __builtin_convertvector(src, x3_bf16x2));
        pl[ph][q] = w;
    }
    if constexpr (up >= 0) {
        constexpr int q = 2 * (up >> 2) + (up & 1), ph = (up >> 1) & 1;
        t.r[q & 1] = x3_pk_sub(ph == 0 ? raw[q] : t.r[q & 1], t.hf[up & 1]);
    }
    if constexpr (ua >= 0) {
        t.hf[ua & 1].x = __builtin_bit_cast(float, w << 16);
        t.hf[ua & 1].y = __builtin_bit_cast(float, w & 0xffff0000u);
    }
    if constexpr (qf >= 0) pl[2][qf] = __builtin_bit_cast(unsigned, __builtin_convertvector(t.r[qf & 1], x3_bf16x2));
}

struct x3_pair2 { x3_f32x2 a, b; };

__device__ __forceinline__ x3_bf16x8 x3_frag(const unsigned (&w)[4]) {
    x3_u32x4 v;
    v[0] = w[0]; v[1] = w[1]; v[2] = w[2]; v[3] = w[3];
    return __builtin_bit_cast(x3_bf16x8, v);
}

template <int I, int N, typename F>
__device__ __forceinline__ void x3_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        x3_static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ unsigned short x3_bf16_rne(float f) {
    const unsigned u = __builtin_bit_cast(unsigned, f);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);   // finite inputs only (weights)
}

}  // namespace
