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

// the same partial sums for c % 4 == 0: 64 channel quads x 4 pixel lanes per workgroup (grid (ceil(c / 256), slabs, n)), 16-byte loads, four
// independent loads in flight per lane; pixel lane l sums pixels p0 + l, p0 + l + 4, ... in ascending order, the four lane sums are added in
// lane order: a fixed association.  The thread-per-channel kernel above walks its slab with one dependent 4-byte load per pixel: 79 us per
// launch at the EfficientNet maps against ~15 us for the traffic.
__global__ __launch_bounds__(256) void gap_partial4_kernel(const float4* __restrict__ x, const float4* __restrict__ w, int hw, int c4, int slabs,
                                                           float4* __restrict__ partial) {
    __shared__ float4 red[4][64];
    const int ql = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + ql;
    const int s = blockIdx.y, n = blockIdx.z;
    const int p0 = s * kSlab, p1 = min(hw, p0 + kSlab);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < c4) {
        const size_t base = (size_t)n * hw * c4 + q;
        int p = p0 + pl;
        for (; p + 12 < p1; p += 16) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = x[base + (size_t)(p + 4 * u) * c4];
            if (w) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 m = w[base + (size_t)(p + 4 * u) * c4];
                    a.x = fmaf(v[u].x, m.x, a.x); a.y = fmaf(v[u].y, m.y, a.y); a.z = fmaf(v[u].z, m.z, a.z); a.w = fmaf(v[u].w, m.w, a.w);
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
            }
        }
        for (; p < p1; p += 4) {
            const float4 v = x[base + (size_t)p * c4];
            if (w) {
                const float4 m = w[base + (size_t)p * c4];
                a.x = fmaf(v.x, m.x, a.x); a.y = fmaf(v.y, m.y, a.y); a.z = fmaf(v.z, m.z, a.z); a.w = fmaf(v.w, m.w, a.w);
            } else {
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
        }
    }
    red[pl][ql] = a;
    __syncthreads();
    if (pl == 0 && q < c4) {
        const float4 b = red[1][ql], c = red[2][ql], d = red[3][ql];
        partial[((size_t)n * slabs + s) * c4 + q] = make_float4(((a.x + b.x) + c.x) + d.x, ((a.y + b.y) + c.y) + d.y, ((a.z + b.z) + c.z) + d.z,
                                                                ((a.w + b.w) + c.w) + d.w);
    }
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

// adjoint of the align_corners = True bilinear enlargement (h, w) -> (h2, w2), as a GATHER: a thread owns one source pixel (x channel quad)
// and walks the enlarged rows / columns whose taps can land on it (source coordinate within one pixel), re-deriving every tap with the
// forward's own float arithmetic (resize_bilinear_kernel, camera.hip) -- so exactly the forward's (pixel, weight) pairs are summed, in
// ascending (y2, x2) order, with no atomics.
__device__ __forceinline__ float resize_tap_weight(int src, int o, float scale, int extent) {
    const float f = scale * (float)o;
    const int i0 = (int)f;
    const int i1 = i0 + (i0 < extent - 1 ? 1 : 0);
    const float l = f - (float)i0;
    float wgt = 0.f;
    if (i0 == src) wgt += 1.0f - l;
    if (i1 == src) wgt += l;
    return wgt;
}

__global__ __launch_bounds__(256) void resize_bwd_kernel(const float* __restrict__ dy, int h, int w, int C, int H2, int W2, float sy, float sx,
                                                         float isy, float isx, float* __restrict__ dx, long long total) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int CQ = C >> 2;
    const int cq = (int)(gid % CQ);
    const long long pix = gid / CQ;
    const int x = (int)(pix % w);
    const long long r = pix / w;
    const int y = (int)(r % h);
    const long long n = r / h;
    // candidates: source coordinate in (y - 1, y + 1), one enlarged row of slack either side for the float rounding of sy * y2
    const int ya = max(0, (int)floorf((float)(y - 1) * isy) - 1), yb = min(H2 - 1, (int)ceilf((float)(y + 1) * isy) + 1);
    const int xa = max(0, (int)floorf((float)(x - 1) * isx) - 1), xb = min(W2 - 1, (int)ceilf((float)(x + 1) * isx) + 1);
    f4 a = f4{0.f, 0.f, 0.f, 0.f};
    for (int y2 = ya; y2 <= yb; ++y2) {
        const float wy = resize_tap_weight(y, y2, sy, h);
        if (wy == 0.f) continue;
        const float* row = dy + (((size_t)n * H2 + y2) * W2) * C + cq * 4;
        for (int x2 = xa; x2 <= xb; ++x2) {
            const float wx = resize_tap_weight(x, x2, sx, w);
            if (wx == 0.f) continue;
            const f4 d = *reinterpret_cast<const f4*>(row + (size_t)x2 * C);
            // the forward's products: (hy * hx), (hy * lx), (ly * hx), (ly * lx) -- when both taps of an axis fall on this pixel (the last
            // row / column) their weights were already added above
            const float wgt = wy * wx;
            a.x = fmaf(d.x, wgt, a.x); a.y = fmaf(d.y, wgt, a.y); a.z = fmaf(d.z, wgt, a.z); a.w = fmaf(d.w, wgt, a.w);
        }
    }
    *reinterpret_cast<f4*>(dx + (size_t)pix * C + cq * 4) = a;
}

// ------------------------------------------------------------------------------------------ depthwise conv: weight gradient
// dw[tap][c] = sum over (n, ho, wo) of dy[n][ho][wo][c] * x[n][ho * stride - pad + kh][wo * stride - pad + kw][c] (zero outside the image).
// grid (ceil(C / 64), slabs); block 256 = 16 channel quads x 16 pixel lanes; a thread keeps all K * K taps of its channel quad in registers
// over its pixels of the slab (stride 16), the 16 lanes are summed in a fixed order (2 shuffles inside the wave, the 4 waves through LDS) and
// the slabs by gap_finish_kernel -- bit-reproducible.  x is read K * K / stride^2 times, all but the first from L1 / L2.
template <int K>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, int H, int W, int C, int stride,
                                                       int pad, int Ho, int Wo, long long npix, int slab, float* __restrict__ partial) {
    __shared__ f4 red[4][K * K][16];
    const int ql = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int cq = blockIdx.x * 16 + ql;
    const long long p0 = (long long)blockIdx.y * slab, p1 = min(npix, p0 + (long long)slab);
    f4 acc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};
    if (cq * 4 < C) {
        for (long long p = p0 + lane; p < p1; p += 16) {
            const int wo = (int)(p % Wo);
            const long long r = p / Wo;
            const int ho = (int)(r % Ho);
            const long long n = r / Ho;
            const f4 d = *reinterpret_cast<const f4*>(dy + (size_t)p * C + cq * 4);
#pragma unroll
            for (int kh = 0; kh < K; ++kh) {
                const int hi = ho * stride - pad + kh;
                if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
                for (int kw = 0; kw < K; ++kw) {
                    const int wi = wo * stride - pad + kw;
                    if ((unsigned)wi >= (unsigned)W) continue;
                    const f4 v = *reinterpret_cast<const f4*>(x + (((size_t)n * H + hi) * W + wi) * C + cq * 4);
                    f4& a = acc[kh * K + kw];
                    a.x = fmaf(v.x, d.x, a.x); a.y = fmaf(v.y, d.y, a.y); a.z = fmaf(v.z, d.z, a.z); a.w = fmaf(v.w, d.w, a.w);
                }
            }
        }
    }
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
        f4 a = acc[t];
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            a.x += __shfl_xor(a.x, off); a.y += __shfl_xor(a.y, off); a.z += __shfl_xor(a.z, off); a.w += __shfl_xor(a.w, off);
        }
        if ((threadIdx.x & 48) == 0) red[wave][t][ql] = a;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < K * K * 16; e += 256) {
        const int t = e >> 4, q = e & 15;
        const int c4 = (blockIdx.x * 16 + q) * 4;
        if (c4 >= C) continue;
        f4 a = red[0][t][q];
        const f4 b1 = red[1][t][q], b2 = red[2][t][q], b3 = red[3][t][q];
        a.x = ((a.x + b1.x) + b2.x) + b3.x; a.y = ((a.y + b1.y) + b2.y) + b3.y;
        a.z = ((a.z + b1.z) + b2.z) + b3.z; a.w = ((a.w + b1.w) + b2.w) + b3.w;
        *reinterpret_cast<f4*>(partial + ((size_t)blockIdx.y * K * K + t) * C + c4) = a;
    }
}

constexpr int kDwSlabMin = 256, kDwSlabMax = 4096;
static int dw_slab(long long npix, int c) {      // >= ~1024 workgroups where the map allows it, slabs of a multiple of 16 pixels
    const long long groups = (c + 63) / 64;
    long long slab = npix * groups / 1024;
    slab = (slab + 15) / 16 * 16;
    return (int)(slab < kDwSlabMin ? kDwSlabMin : slab > kDwSlabMax ? kDwSlabMax : slab);
}

// ---- nn.MaxPool2d backward as a gather: an input pixel collects dy from every window that contains it and whose FIRST maximum (scan order
// kh, kw: torch's forward keeps the first element that is greater) sits on it -- no atomics, no saved indices
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int H, int W, int C, int ks,
                                                          int stride, int pad, int Ho, int Wo, long long total, float* __restrict__ dx) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int c = (int)(gid % C);
    const long long pix = gid / C;
    const int wi = (int)(pix % W);
    const long long r = pix / W;
    const int hi = (int)(r % H);
    const long long n = r / H;
    const float* xi = x + (size_t)n * H * W * C + c;
    float g = 0.f;
    // windows (ho, wo) with ho * stride - pad <= hi < ho * stride - pad + ks
    const int ho1 = min(Ho - 1, (hi + pad) / stride), wo1 = min(Wo - 1, (wi + pad) / stride);
    const int ho0 = max(0, (hi + pad - ks + stride) / stride), wo0 = max(0, (wi + pad - ks + stride) / stride);
    for (int ho = ho0; ho <= ho1; ++ho)
        for (int wo = wo0; wo <= wo1; ++wo) {
            float m = -INFINITY;
            int mh = -1, mw = -1;
            for (int kh = 0; kh < ks; ++kh) {
                const int h2 = ho * stride - pad + kh;
                if ((unsigned)h2 >= (unsigned)H) continue;
                for (int kw = 0; kw < ks; ++kw) {
                    const int w2 = wo * stride - pad + kw;
                    if ((unsigned)w2 >= (unsigned)W) continue;
                    const float v = xi[((size_t)h2 * W + w2) * C];
                    if (v > m || mh < 0) { m = v; mh = h2; mw = w2; }
                }
            }
            if (mh == hi && mw == wi) g += dy[(((size_t)n * Ho + ho) * Wo + wo) * C + c];
        }
    dx[gid] = g;
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
    const bool al16 = reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(workspace) % 16 == 0 && (!w || reinterpret_cast<uintptr_t>(w) % 16 == 0);
    if (c % 4 == 0 && al16)
        hipLaunchKernelGGL(gap_partial4_kernel, dim3((c / 4 + 63) / 64, slabs, n), dim3(256), 0, st, reinterpret_cast<const float4*>(x),
                           reinterpret_cast<const float4*>(w), hw, c / 4, slabs, reinterpret_cast<float4*>(workspace));
    else
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

extern "C" uint64_t av2x_resize_bilinear_backward_workspace_bytes(int32_t, int32_t, int32_t, int32_t) { return 0; }

// dy (n, h2, w2, c) = gradient of the (h, w) -> (h2, w2) align_corners = True enlargement; dx (n, h, w, c).  `workspace` is unused (may be NULL).
extern "C" int av2x_resize_bilinear_backward(const float* dy, int32_t n, int32_t h, int32_t w, int32_t c, int32_t h2, int32_t w2, void* workspace,
                                             float* dx, av2x_stream_t stream) {
    (void)workspace;
    if (n == 0) return 0;
    if (!dy || !dx) return av2x::fail("av2x_resize_bilinear_backward: null argument");
    if (n < 0 || h <= 0 || w <= 0 || h2 <= 0 || w2 <= 0 || c <= 0 || c % 4) return av2x::fail("av2x_resize_bilinear_backward: bad sizes (c %% 4 == 0)");
    hipStream_t st = av2x::as_stream(stream);
    const float sy = h2 > 1 ? (float)(h - 1) / (float)(h2 - 1) : 0.f;
    const float sx = w2 > 1 ? (float)(w - 1) / (float)(w2 - 1) : 0.f;
    // inverse scales for the candidate window; a degenerate axis (one source row: every enlarged row reads it) scans the whole axis
    const float isy = sy > 0.f ? 1.0f / sy : (float)h2, isx = sx > 0.f ? 1.0f / sx : (float)w2;
    const long long total = (long long)n * h * w * (c / 4);
    hipLaunchKernelGGL(resize_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dy, h, w, c, h2, w2, sy, sx, isy, isx, dx, total);
    return av2x::check_launch("resize_bwd_kernel");
}

// dx (n, h, w, c) of av2x_maxpool2d: x its input, dy (n, ho, wo, c)
extern "C" int av2x_maxpool2d_backward(const float* x, const float* dy, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ks, int32_t stride,
                                       int32_t pad, int32_t ho, int32_t wo, float* dx, av2x_stream_t stream) {
    if (n == 0) return 0;
    if (!x || !dy || !dx) return av2x::fail("av2x_maxpool2d_backward: null argument");
    if (n < 0 || h <= 0 || w <= 0 || c <= 0 || ks <= 0 || stride <= 0 || pad < 0 || 2 * pad > ks) return av2x::fail("av2x_maxpool2d_backward: bad sizes");
    if (ho != (h + 2 * pad - ks) / stride + 1 || wo != (w + 2 * pad - ks) / stride + 1) return av2x::fail("av2x_maxpool2d_backward: output size mismatch");
    const long long total = (long long)n * h * w * c;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, av2x::as_stream(stream), x, dy, h, w, c, ks, stride, pad,
                       ho, wo, total, dx);
    return av2x::check_launch("maxpool_bwd_kernel");
}

extern "C" uint64_t av2x_dwconv2d_wgrad_workspace_bytes(int32_t n, int32_t ho, int32_t wo, int32_t c, int32_t ks) {
    if (n <= 0 || ho <= 0 || wo <= 0 || c <= 0 || ks <= 0) return 0;
    const long long npix = (long long)n * ho * wo;
    const int slab = dw_slab(npix, c);
    return (uint64_t)((npix + slab - 1) / slab) * ks * ks * c * sizeof(float);
}

// dw (ks*ks, c) of the depthwise conv av2x_dwconv2d ran with (stride, pad_t = pad_l = pad): x (n, h, w, c), dy (n, ho, wo, c)
extern "C" int av2x_dwconv2d_wgrad(const float* x, const float* dy, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ks, int32_t stride, int32_t pad,
                                   int32_t ho, int32_t wo, void* workspace, float* dw, av2x_stream_t stream) {
    if (!x || !dy || !workspace || !dw) return av2x::fail("av2x_dwconv2d_wgrad: null argument");
    if (n <= 0 || h <= 0 || w <= 0 || ho <= 0 || wo <= 0 || c <= 0 || c % 4) return av2x::fail("av2x_dwconv2d_wgrad: bad sizes (c %% 4 == 0)");
    if ((ks != 3 && ks != 5) || (stride != 1 && stride != 2) || pad < 0) return av2x::fail("av2x_dwconv2d_wgrad: ks %d (3 | 5), stride %d (1 | 2)", ks, stride);
    const long long npix = (long long)n * ho * wo;
    const int slab = dw_slab(npix, c);
    const int slabs = (int)((npix + slab - 1) / slab);
    hipStream_t st = av2x::as_stream(stream);
    float* part = static_cast<float*>(workspace);
    const dim3 grid((c + 63) / 64, slabs);
    if (ks == 3) hipLaunchKernelGGL(dw_wgrad_kernel<3>, grid, dim3(256), 0, st, x, dy, h, w, c, stride, pad, ho, wo, npix, slab, part);
    else hipLaunchKernelGGL(dw_wgrad_kernel<5>, grid, dim3(256), 0, st, x, dy, h, w, c, stride, pad, ho, wo, npix, slab, part);
    const int tc = ks * ks * c;
    hipLaunchKernelGGL(gap_finish_kernel, dim3((unsigned)((tc + 255) / 256)), dim3(256), 0, st, part, 1, tc, slabs, 1.0f, dw);
    return av2x::check_launch("dw_wgrad_kernel");
}
