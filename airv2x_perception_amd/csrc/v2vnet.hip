// V2VNet fusion pieces (models/v2vnet_modules/v2v_fuse.py:110-180): the masked mean / max over the neighbours' messages.
// The two 3x3 convolutions either side of it (msg_cnn split by linearity, the ConvGRU gate / candidate convolutions with the
// sigmoid and gated-tanh epilogues) run on conv_igemm; the warps on warp_affine_kernel (v2xvit.hip).  HBM-bound:
// (n + 1) * h * w * c * 4 bytes read, h * w * c * 4 written.
#include "av2x_common.hpp"

namespace {

__device__ __forceinline__ float lin_m1_1_(int i, int n) { return n > 1 ? (2.f * (float)i) / (float)(n - 1) - 1.f : 0.f; }

// sum of the in-image bilinear weights of output pixel (i, j) under F.affine_grid(theta, align_corners=False) +
// F.grid_sample(ones, bilinear, zeros): the value warp_affine_simple gives on an all-ones map (torch_transformation_utils.py:327-334)
__device__ __forceinline__ float roi_weight(const float* th, int i, int j, int H, int W) {
    const float xn = (lin_m1_1_(j, W) * (float)(W - 1)) / (float)W, yn = (lin_m1_1_(i, H) * (float)(H - 1)) / (float)H;
    const float gx = th[0] * xn + th[1] * yn + th[2];
    const float gy = th[3] * xn + th[4] * yn + th[5];
    const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - x0f, wx0 = (x0f + 1.f) - ix, wy1 = iy - y0f, wy0 = (y0f + 1.f) - iy;
    const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)x1 < (unsigned)W;
    const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)y1 < (unsigned)H;
    float m = 0.f;
    if (vy0 && vx0) m += wx0 * wy0;
    if (vy0 && vx1) m += wx1 * wy0;
    if (vy1 && vx0) m += wx0 * wy1;
    if (vy1 && vx1) m += wx1 * wy1;
    return m;
}

// 16 lanes per pixel, float4 per lane and chunk of 64 channels
template <bool MAX>
__global__ __launch_bounds__(256) void v2v_aggregate_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            const float* __restrict__ theta, int n, int H, int W, int C,
                                                            float* __restrict__ out) {
    const int t = threadIdx.x & 15;
    const int pix = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (pix >= H * W) return;
    const int i = pix / W, j = pix - i * W;
    float mask[32];
    for (int k = 0; k < n; ++k) mask[k] = roi_weight(theta + 6 * k, i, j, H, W);
    const size_t hwc = (size_t)H * W * C;
    const float inv = 1.0f / (float)n;
    for (int c0 = 4 * t; c0 < C; c0 += 64) {
        const float4 bv = *reinterpret_cast<const float4*>(b + (size_t)pix * C + c0);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < n; ++k) {
            const float4 av = *reinterpret_cast<const float4*>(a + k * hwc + (size_t)pix * C + c0);
            const float m = mask[k];
            const float4 v = make_float4((av.x + bv.x) * m, (av.y + bv.y) * m, (av.z + bv.z) * m, (av.w + bv.w) * m);
            if (MAX) {
                acc = k == 0 ? v : make_float4(fmaxf(acc.x, v.x), fmaxf(acc.y, v.y), fmaxf(acc.z, v.z), fmaxf(acc.w, v.w));
            } else {
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        if (!MAX) { acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv; }
        *reinterpret_cast<float4*>(out + (size_t)pix * C + c0) = acc;
    }
}

}  // namespace

extern "C" int av2x_v2v_aggregate(const float* msg_a, const float* ego_b, const float* theta, int32_t n, int32_t h, int32_t w,
                                  int32_t c, int32_t op, float* out, av2x_stream_t stream) {
    if (!msg_a || !ego_b || !theta || !out) return av2x::fail("av2x_v2v_aggregate: null argument");
    if (n < 1 || n > 32 || h <= 0 || w <= 0 || c <= 0 || c % 4) return av2x::fail("av2x_v2v_aggregate: bad sizes (n=%d c=%d)", n, c);
    if (op != 0 && op != 1) return av2x::fail("av2x_v2v_aggregate: op %d (0 avg, 1 max)", op);
    const dim3 grid((h * w + 15) / 16), block(256);
    hipStream_t st = av2x::as_stream(stream);
    if (op == 1) hipLaunchKernelGGL(v2v_aggregate_kernel<true>, grid, block, 0, st, msg_a, ego_b, theta, n, h, w, c, out);
    else hipLaunchKernelGGL(v2v_aggregate_kernel<false>, grid, block, 0, st, msg_a, ego_b, theta, n, h, w, c, out);
    return av2x::check_launch("v2v_aggregate_kernel");
}
