// SplitAttn's radix softmax and weighted sum (split_attn.py:55-61) for four channels of one token: the ONE definition used by
// split_combine_kernel (v2xvit.hip) and by the panel load of linear_bf16_occ_kernel<SRC_LNC, .> (linear_bf16.hip), so that the fused
// launch writes the bits of the separate one.
#pragma once
#include <hip/hip_runtime.h>

namespace av2x {

// w[b][e] = softmax over the three branches of logits[b * C + c + e]        (logits of one agent viewed (radix = 3, C))
__device__ __forceinline__ void split_attn_weights(const float* __restrict__ lg, int c, int C, float (&w)[3][4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a0 = lg[c + e], a1 = lg[C + c + e], a2 = lg[2 * C + c + e];
        const float mx = fmaxf(a0, fmaxf(a1, a2));
        const float e0 = expf(a0 - mx), e1 = expf(a1 - mx), e2 = expf(a2 - mx);
        const float inv = 1.0f / ((e0 + e1) + e2);
        w[0][e] = e0 * inv; w[1][e] = e1 * inv; w[2][e] = e2 * inv;
    }
}

// y = ((x0 w0 + x1 w1) + x2 w2) + r
__device__ __forceinline__ float4 split_attn_combine4(const float4 x0, const float4 x1, const float4 x2, const float (&w)[3][4], const float4 r) {
    float4 y;
    y.x = ((x0.x * w[0][0] + x1.x * w[1][0]) + x2.x * w[2][0]) + r.x;
    y.y = ((x0.y * w[0][1] + x1.y * w[1][1]) + x2.y * w[2][1]) + r.y;
    y.z = ((x0.z * w[0][2] + x1.z * w[1][2]) + x2.z * w[2][2]) + r.z;
    y.w = ((x0.w * w[0][3] + x1.w * w[1][3]) + x2.w * w[2][3]) + r.w;
    return y;
}

}  // namespace av2x
