// Training-side kernels of the camera branch (CamEncode / BevEncode, models/sub_modules/lss_submodule.py:22-189, 312-350, and the
// efficientnet_pytorch trunk CamEncode walks): what the reference gets from torch autograd for the pieces that are not convolutions.
// All HBM-bound; every reduction runs in one fixed order (slabs of pixels, then the slabs in ascending order): run-to-run identical.
//
//   unary_kernel / unary_grad_kernel        swish (x sigmoid(x)) and sigmoid, forward and derivative from the PRE-activation value
//   add_act_kernel                          y = act(a + b): the skip connections (MBConv: no activation; BasicBlock: ReLU after the add)
//   gap_partial / gap_finish / gap_bwd      squeeze of squeeze-and-excite: per-(image, channel) mean over the pixels and its broadcast adjoint
//   scale_kernel / scale_bwd (+ partials)   excite: y = x * gate[image][channel]; dx = dy * gate, dgate = sum over pixels of dy * x
//   resize_bwd_kernel                       adjoint of nn.Upsample(bilinear, align_corners=True) (the Up blocks), scattered with the
//                                           2^-32 fixed-point atomics of the warp adjoint (order-independent integer sums)
#include <cstdint>

#include "av2x_common.hpp"
#include "airv2x_hip.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr float kFix = 4294967296.0f;   // 2^32
constexpr int kSlab = 256;              // pixels per slab of the per-(image, channel) reductions

__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }

// act: 3 sigmoid, 6 swish (the activation codes of av2x_conv_desc.relu)
template <int ACT>
__device__ __forceinline__ float act_f(float v) { return ACT == 3 ? sigm(v) : v * sigm(v); }
template <int ACT>
__device__ __forceinline__ float act_d(float v) {
    const float s = sigm(v);
    return ACT == 3 ? s * (1.f - s) : s * (1.f + v * (1.f - s));
}

template <int ACT>
__global__ __launch_bounds__(256) void unary_kernel(const f4* __restrict__ x, f4* __restrict__ y, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f4 v = x[i];
        y[i] = f4{act_f<ACT>(v.x), act_f<ACT>(v.y), act_f<ACT>(v.z), act_f<ACT>(v.w)};
    }
}

template <int ACT>
__global__ __launch_bounds__(256) void unary_grad_kernel(const f4* __restrict__ x, const f4* __restrict__ dy, f4* __restrict__ dx, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f4 v = x[i], d = dy[i];
        dx[i] = f4{d.x * act_d<ACT>(v.x), d.y * act_d<ACT>(v.y), d.z * act_d<ACT>(v.z), d.w * act_d<ACT>(v.w)};
    }
}

__global__ __launch_bounds__(256) void add_act_kernel(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ y, size_t n4, int relu) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f4 u = a[i], v = b[i];
        f4 o = f4{u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w};
        if (relu) o = f4{fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f)};
        y[i] = o;
    }
}

// partial[n][s][c] = sum over the pixels of slab s of x[n][p][c] (* w[n][p][c] when w != nullptr); grid (ceil(c / 256), slabs, n)
__global__ __launch_bounds__(256) void gap_partial_kernel(const float* __restrict__ x, const float* __restrict__ w, int hw, int c, int slabs,
                                                          float* __restrict__ partial) {
    const int ch = blockIdx.x * 256 + threadIdx.x;
    if (ch >= c) return;
    const int s = blockIdx.y, n = blockIdx.z;
    const int p0 = s * kSlab, p1 = min(hw, p0 + kSlab);
    const size_t base = (size_t)n * hw * c + ch;
    float a = 0.f;
    if (w) for (int p = p0; p < p1; ++p) a = fmaf(x[base + (size_t)p * c], w[base + (size_t)p * c], a);
    else for (int p = p0; p < p1; ++p) a += x[base + (size_t)p * c];
    partial[((size_t)n * slabs + s) * c + ch] = a;
}

__global__ __launch_bounds__(256) void gap_finish_kernel(const float* __restrict__ partial, int n, int c, int slabs, float scale, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n * c) return;
    const int img = (int)(i / c), ch = (int)(i % c);
    float a = 0.f;
    for (int s = 0; s < slabs; ++s) a += partial[((size_t)img * slabs + s) * c + ch];
    out[i] = a * scale;
}

// dx[n][p][c] = g[n][c] * scale (* dy[n][p][c] when dy != nullptr)
__global__ __launch_bounds__(256) void bcast_kernel(const float* __restrict__ g, const float* __restrict__ dy, int hw, int c, float scale,
                                                    float* __restrict__ dx, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ch = (int)(i % c);
        const size_t img = i / ((size_t)hw * c);
        const float v = g[img * c + ch] * scale;
        dx[i] = dy ? dy[i] * v : v;
    }
}

// adjoint of the align_corners = True bilinear enlargement (h, w) -> (h2, w2): every enlarged pixel hands its gradient to its four sources
__global__ __launch_bounds__(256) void resize_bwd_kernel(const float* __restrict__ dy, int h, int w, int C, int H2, int W2, float sy, float sx,
                                                         unsigned long long* __restrict__ acc, long long total) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int CQ = C >> 2;
    const int cq = (int)(gid % CQ);
    const long long pix = gid / CQ;
    const int x2 = (int)(pix % W2);
    const long long r = pix / W2;
    const int y2 = (int)(r % H2);
    const long long n = r / H2;
    const float fy = sy * (float)y2, fx = sx * (float)x2;       // as resize_bilinear_kernel (camera.hip)
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.0f - ly, hx = 1.0f - lx;
    const f4 d = *reinterpret_cast<const f4*>(dy + (size_t)pix * C + cq * 4);
    unsigned long long* base = acc + (size_t)n * h * w * C + cq * 4;
    auto add = [&](int yy, int xx, float wgt) {
        if (wgt == 0.f) return;
        unsigned long long* a = base + ((size_t)yy * w + xx) * C;
        atomicAdd(a + 0, (unsigned long long)__float2ll_rn(d.x * wgt * kFix));
        atomicAdd(a + 1, (unsigned long long)__float2ll_rn(d.y * wgt * kFix));
        atomicAdd(a + 2, (unsigned long long)__float2ll_rn(d.z * wgt * kFix));
        atomicAdd(a + 3, (unsigned long long)__float2ll_rn(d.w * wgt * kFix));
    };
    add(y0, x0, hy * hx); add(y0, x1, hy * lx); add(y1, x0, ly * hx); add(y1, x1, ly * lx);
}

__global__ __launch_bounds__(256) void fixed_to_float_kernel(const long long* __restrict__ acc, float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = (float)((double)acc[i] * (1.0 / 4294967296.0));
}

unsigned blocks_for(size_t n) {
    size_t b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : b > 8192 ? 8192 : b);
}

}  // namespace

extern "C" int av2x_unary_forward(const float* x, uint64_t n_elems, int32_t act, float* y, av2x_stream_t stream) {
    if (n_elems == 0) return 0;
    if (!x || !y) return av2x::fail("av2x_unary_forward: null argument");
    if (n_elems % 4) return av2x::fail("av2x_unary_forward: element count must be a multiple of 4");
    if (act != 3 && act != 6) return av2x::fail("av2x_unary_forward: activation %d (3 sigmoid, 6 swish)", act);
    const size_t n4 = n_elems / 4;
    hipStream_t st = av2x::as_stream(stream);
    if (act == 3) hipLaunchKernelGGL(unary_kernel<3>, dim3(blocks_for(n4)), dim3(256), 0, st, reinterpret_cast<const f4*>(x), reinterpret_cast<f4*>(y), n4);
    else hipLaunchKernelGGL(unary_kernel<6>, dim3(blocks_for(n4)), dim3(256), 0, st, reinterpret_cast<const f4*>(x), reinterpret_cast<f4*>(y), n4);
    return av2x::check_launch("unary_kernel");
}

extern "C" int av2x_unary_backward(const float* x, const float* dy, uint64_t n_elems, int32_t act, float* dx, av2x_stream_t stream) {
    if (n_elems == 0) return 0;
    if (!x || !dy || !dx) return av2x::fail("av2x_unary_backward: null argument");
    if (n_elems % 4) return av2x::fail("av2x_unary_backward: element count must be a multiple of 4");
    if (act != 3 && act != 6) return av2x::fail("av2x_unary_backward: activation %d (3 sigmoid, 6 swish)", act);
    const size_t n4 = n_elems / 4;
    hipStream_t st = av2x::as_stream(stream);
    if (act == 3) hipLaunchKernelGGL(unary_grad_kernel<3>, dim3(blocks_for(n4)), dim3(256), 0, st, reinterpret_cast<const f4*>(x), reinterpret_cast<const f4*>(dy), reinterpret_cast<f4*>(dx), n4);
    else hipLaunchKernelGGL(unary_grad_kernel<6>, dim3(blocks_for(n4)), dim3(256), 0, st, reinterpret_cast<const f4*>(x), reinterpret_cast<const f4*>(dy), reinterpret_cast<f4*>(dx), n4);
    return av2x::check_launch("unary_grad_kernel");
}

extern "C" int av2x_add_act(const float* a, const float* b, uint64_t n_elems, int32_t relu, float* y, av2x_stream_t stream) {
    if (n_elems == 0) return 0;
    if (!a || !b || !y) return av2x::fail("av2x_add_act: null argument");
    if (n_elems % 4) return av2x::fail("av2x_add_act: element count must be a multiple of 4");
    const size_t n4 = n_elems / 4;
    hipLaunchKernelGGL(add_act_kernel, dim3(blocks_for(n4)), dim3(256), 0, av2x::as_stream(stream), reinterpret_cast<const f4*>(a), reinterpret_cast<const f4*>(b),
                       reinterpret_cast<f4*>(y), n4, relu ? 1 : 0);
    return av2x::check_launch("add_act_kernel");
}

extern "C" uint64_t av2x_gap_workspace_bytes(int32_t n, int32_t hw, int32_t c) {
    if (n <= 0 || hw <= 0 || c <= 0) return 0;
    return (uint64_t)n * ((hw + kSlab - 1) / kSlab) * c * sizeof(float);
}

// out (n, c) = scale * sum over the hw pixels of x (n, hw, c) [* w (n, hw, c)]: scale = 1 / hw is the squeeze, w = dy gives dgate of the excite
extern "C" int av2x_gap(const float* x, const float* w, int32_t n, int32_t hw, int32_t c, float scale, float* workspace, float* out, av2x_stream_t stream) {
    if (n == 0) return 0;
    if (!x || !workspace || !out) return av2x::fail("av2x_gap: null argument");
    if (n < 0 || hw <= 0 || c <= 0) return av2x::fail("av2x_gap: bad sizes");
    const int slabs = (hw + kSlab - 1) / kSlab;
    hipStream_t st = av2x::as_stream(stream);
    hipLaunchKernelGGL(gap_partial_kernel, dim3((c + 255) / 256, slabs, n), dim3(256), 0, st, x, w, hw, c, slabs, workspace);
    hipLaunchKernelGGL(gap_finish_kernel, dim3((unsigned)(((size_t)n * c + 255) / 256)), dim3(256), 0, st, workspace, n, c, slabs, scale, out);
    return av2x::check_launch("gap_kernel");
}

// out (n, hw, c) = g (n, c) * scale [* y (n, hw, c)]: the adjoint of the squeeze (y NULL, scale 1 / hw), the excite's forward (y = x, scale 1)
// and the excite's data gradient (y = dy, scale 1)
extern "C" int av2x_channel_broadcast(const float* g, const float* y, int32_t n, int32_t hw, int32_t c, float scale, float* out, av2x_stream_t stream) {
    if (n == 0) return 0;
    if (!g || !out) return av2x::fail("av2x_channel_broadcast: null argument");
    if (n < 0 || hw <= 0 || c <= 0) return av2x::fail("av2x_channel_broadcast: bad sizes");
    const size_t total = (size_t)n * hw * c;
    hipLaunchKernelGGL(bcast_kernel, dim3(blocks_for(total)), dim3(256), 0, av2x::as_stream(stream), g, y, hw, c, scale, out, total);
    return av2x::check_launch("bcast_kernel");
}

extern "C" uint64_t av2x_resize_bilinear_backward_workspace_bytes(int32_t n, int32_t h, int32_t w, int32_t c) { return (uint64_t)n * h * w * c * 8ull; }

// dy (n, h2, w2, c) = gradient of the (h, w) -> (h2, w2) align_corners = True enlargement; dx (n, h, w, c)
extern "C" int av2x_resize_bilinear_backward(const float* dy, int32_t n, int32_t h, int32_t w, int32_t c, int32_t h2, int32_t w2, void* workspace,
                                             float* dx, av2x_stream_t stream) {
    if (n == 0) return 0;
    if (!dy || !workspace || !dx) return av2x::fail("av2x_resize_bilinear_backward: null argument");
    if (n < 0 || h <= 0 || w <= 0 || h2 <= 0 || w2 <= 0 || c <= 0 || c % 4) return av2x::fail("av2x_resize_bilinear_backward: bad sizes (c %% 4 == 0)");
    hipStream_t st = av2x::as_stream(stream);
    const size_t total_in = (size_t)n * h * w * c;
    hipError_t e = hipMemsetAsync(workspace, 0, total_in * 8ull, st);
    if (e != hipSuccess) return av2x::fail("av2x_resize_bilinear_backward: memset: %s", hipGetErrorString(e));
    const float sy = h2 > 1 ? (float)(h - 1) / (float)(h2 - 1) : 0.f;
    const float sx = w2 > 1 ? (float)(w - 1) / (float)(w2 - 1) : 0.f;
    const long long total = (long long)n * h2 * w2 * (c / 4);
    hipLaunchKernelGGL(resize_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dy, h, w, c, h2, w2, sy, sx,
                       reinterpret_cast<unsigned long long*>(workspace), total);
    hipLaunchKernelGGL(fixed_to_float_kernel, dim3(blocks_for(total_in)), dim3(256), 0, st, reinterpret_cast<const long long*>(workspace), dx, total_in);
    return av2x::check_launch("resize_bwd_kernel");
}
