// Camera branch of the multimodal frame (BASELINE configs[4]): the HBM-bound pieces around the convolutions of
// CamEncode / BevEncode (models/sub_modules/lss_submodule.py:22-189, 312-350).
//
//   cam_stem_kernel        EfficientNet stem: Conv3x3 / stride 2 with the package's static "same" padding, folded BN, swish, reading
//                          the NCHW image planes the dataset hands over and writing NHWC (what every later kernel reads)
//   dwconv_kernel<K>       MBConv depthwise K x K conv (3 / 5, stride 1 / 2, asymmetric zero padding) + folded BN + swish
//   se_gap_kernel          squeeze: per-(image, channel) sums in a FIXED order (slabs of pixels, 16 lanes per channel quad, LDS tree)
//   se_fc_kernel           excite: mean -> 1x1 reduce + swish -> 1x1 expand + sigmoid = one gate per (image, channel)
//   channel_scale_kernel   x *= gate[image][channel]
//   resize_bilinear_kernel nn.Upsample(bilinear, align_corners=True) + F.pad + the channel slice of torch.cat in one pass
//   softmax_channels_kernel depth distribution: softmax over the D logits of a pixel
//   lift_pool_gt / lift_pool_prob   depth (one-hot of the binned ground-truth depth | softmax) (x) image features, lifted along the
//                          camera rays and summed into the BEV grid with the 64-bit fixed-point atomics of lss.hip: the
//                          (B, N, D, fH, fW, C) volume of CamEncode.forward (:176-186) is never written.
//
// All of these are HBM / L2 bound (a 360 x 640 image's whole EfficientNet-B0 activation set is ~150 MB); the FLOPs of the
// branch sit in the pointwise / 3x3 / 7x7 convolutions, which run on conv_igemm / conv_wino.
#include "av2x_common.hpp"

namespace {

__device__ __forceinline__ float swishf(float v) { return v / (1.0f + expf(-v)); }

// ---------------------------------------------------------------------------------------------------------------- stem
// thread -> (output pixel, quad of the 32 output channels); weights [27][32] (tap-major: (kh*3+kw)*3+ci) in LDS
__global__ __launch_bounds__(256) void cam_stem_kernel(const float* __restrict__ img, int planes, int H, int W, const float* __restrict__ w,
                                                       const float* __restrict__ scale, const float* __restrict__ shift, int pad_t, int pad_l,
                                                       int Ho, int Wo, long long npix, float* __restrict__ out) {
    __shared__ float ws[27 * 32];
    __shared__ float sc[32], sh[32];
    for (int i = threadIdx.x; i < 27 * 32; i += 256) ws[i] = w[i];
    if (threadIdx.x < 32) { sc[threadIdx.x] = scale[threadIdx.x]; sh[threadIdx.x] = shift[threadIdx.x]; }
    __syncthreads();
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long pix = gid >> 3;
    const int q = (int)(gid & 7);
    if (pix >= npix) return;
    const int wo = (int)(pix % Wo);
    const long long r = pix / Wo;
    const int ho = (int)(r % Ho);
    const long long n = r / Ho;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int hi = ho * 2 - pad_t + kh;
        if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int wi = wo * 2 - pad_l + kw;
            if ((unsigned)wi >= (unsigned)W) continue;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const float v = img[((n * planes + ci) * H + hi) * (long long)W + wi];
                const float* wp = ws + ((kh * 3 + kw) * 3 + ci) * 32 + q * 4;
                a0 = fmaf(v, wp[0], a0); a1 = fmaf(v, wp[1], a1); a2 = fmaf(v, wp[2], a2); a3 = fmaf(v, wp[3], a3);
            }
        }
    }
    float4 o;
    o.x = swishf(a0 * sc[q * 4 + 0] + sh[q * 4 + 0]);
    o.y = swishf(a1 * sc[q * 4 + 1] + sh[q * 4 + 1]);
    o.z = swishf(a2 * sc[q * 4 + 2] + sh[q * 4 + 2]);
    o.w = swishf(a3 * sc[q * 4 + 3] + sh[q * 4 + 3]);
    *reinterpret_cast<float4*>(out + pix * 32 + q * 4) = o;
}

// ------------------------------------------------------------------------------------------------------------ depthwise
// x (n,H,W,C) NHWC -> out (n,Ho,Wo,C); w [K*K][C]; thread -> (output pixel, channel quad): a wave reads 1 KiB runs of channels
template <int K>
__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ x, int H, int W, int C, const float* __restrict__ w,
                                                     const float* __restrict__ scale, const float* __restrict__ shift, int stride, int pad_t,
                                                     int pad_l, int Ho, int Wo, long long total, int act, float* __restrict__ out) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int CQ = C >> 2;
    const int cq = (int)(gid % CQ);
    const long long pix = gid / CQ;
    const int wo = (int)(pix % Wo);
    const long long r = pix / Wo;
    const int ho = (int)(r % Ho);
    const long long n = r / Ho;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int kh = 0; kh < K; ++kh) {
        const int hi = ho * stride - pad_t + kh;
        if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
            const int wi = wo * stride - pad_l + kw;
            if ((unsigned)wi >= (unsigned)W) continue;
            const float4 v = *reinterpret_cast<const float4*>(x + ((n * H + hi) * (long long)W + wi) * C + cq * 4);
            const float4 k4 = *reinterpret_cast<const float4*>(w + (kh * K + kw) * C + cq * 4);
            a.x = fmaf(v.x, k4.x, a.x); a.y = fmaf(v.y, k4.y, a.y); a.z = fmaf(v.z, k4.z, a.z); a.w = fmaf(v.w, k4.w, a.w);
        }
    }
    const float4 s4 = *reinterpret_cast<const float4*>(scale + cq * 4);
    const float4 b4 = *reinterpret_cast<const float4*>(shift + cq * 4);
    float4 o = make_float4(a.x * s4.x + b4.x, a.y * s4.y + b4.y, a.z * s4.z + b4.z, a.w * s4.w + b4.w);
    if (act == 6) { o.x = swishf(o.x); o.y = swishf(o.y); o.z = swishf(o.z); o.w = swishf(o.w); }
    else if (act == 1) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    *reinterpret_cast<float4*>(out + pix * C + cq * 4) = o;
}

// ------------------------------------------------------------------------------------------------------- squeeze-excite
// grid (C/64 ceil, S, n); block 256 = 16 channel quads x 16 pixel lanes; partial[n][s][c] = sum over the slab's pixels
__global__ __launch_bounds__(256) void se_gap_kernel(const float* __restrict__ x, int HW, int C, int S, float* __restrict__ partial) {
    __shared__ float4 red[256];
    const int ql = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int cq = blockIdx.x * 16 + ql;
    const int s = blockIdx.y, n = blockIdx.z;
    const int per = (HW + S - 1) / S;
    const int p0 = s * per, p1 = min(HW, p0 + per);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cq * 4 < C) {
        const float* base = x + (size_t)n * HW * C + cq * 4;
        for (int p = p0 + lane; p < p1; p += 16) {
            const float4 v = *reinterpret_cast<const float4*>(base + (size_t)p * C);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    red[threadIdx.x] = a;
    __syncthreads();
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
        if (lane < off) {
            const float4 b = red[threadIdx.x + off * 16];
            float4 c = red[threadIdx.x];
            c.x += b.x; c.y += b.y; c.z += b.z; c.w += b.w;
            red[threadIdx.x] = c;
        }
        __syncthreads();
    }
    if (lane == 0 && cq * 4 < C) *reinterpret_cast<float4*>(partial + ((size_t)n * S + s) * C + cq * 4) = red[ql];
}

// one workgroup per image: gate[n][c] = sigmoid(be[c] + sum_j we[c][j] * swish(br[j] + sum_c' wr[j][c'] * mean[c']))
__global__ __launch_bounds__(256) void se_fc_kernel(const float* __restrict__ partial, int S, int HW, int C, int Cse,
                                                    const float* __restrict__ wr, const float* __restrict__ br,
                                                    const float* __restrict__ we, const float* __restrict__ be, float* __restrict__ gate) {
    extern __shared__ float lds[];          // mean[C] | r[Cse]
    float* mean = lds;
    float* rr = lds + C;
    const int n = blockIdx.x;
    const float inv = 1.0f / (float)HW;
    for (int c = threadIdx.x; c < C; c += 256) {
        // the slab partials in ascending order (fixed association), loads issued eight at a time instead of one per dependent add
        const float* pp = partial + (size_t)n * S * C + c;
        float a = 0.f;
        int s = 0;
        for (; s + 16 <= S; s += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = pp[(size_t)(s + u) * C];
#pragma unroll
            for (int u = 0; u < 16; ++u) a += v[u];
        }
        for (; s + 8 <= S; s += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = pp[(size_t)(s + u) * C];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; s < S; ++s) a += pp[(size_t)s * C];
        mean[c] = a * inv;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, ln = threadIdx.x & 63;
    // Memory-level parallelism is what this kernel is about (26 workgroups, every load a cold miss): four reduce rows x four channel
    // groups = 16 loads in flight per lane and pass; per row the sum order is unchanged (ascending c per lane, then the butterfly).
    for (int j0 = wave; j0 < Cse; j0 += 16) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c0 = ln; c0 < C; c0 += 256) {
            float wv[4][4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + 4 * u, c = c0 + 64 * g;
                    wv[g][u] = (j < Cse && c < C) ? wr[(size_t)j * C + c] : 0.f;
                }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = c0 + 64 * g;
                if (c < C) {
                    const float mc = mean[c];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (j0 + 4 * u < Cse) a[u] = fmaf(wv[g][u], mc, a[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v = a[u];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
            const int j = j0 + 4 * u;
            if (ln == 0 && j < Cse) rr[j] = swishf(v + br[j]);
        }
    }
    __syncthreads();
    // we_t [Cse][C] (transposed on the host): neighbouring lanes read neighbouring floats; sixteen loads in flight per thread
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = be[c];
        int j = 0;
        for (; j + 16 <= Cse; j += 16) {
            float wv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) wv[u] = we[(size_t)(j + u) * C + c];
#pragma unroll
            for (int u = 0; u < 16; ++u) a = fmaf(wv[u], rr[j + u], a);
        }
        for (; j < Cse; ++j) a = fmaf(we[(size_t)j * C + c], rr[j], a);
        gate[(size_t)n * C + c] = 1.0f / (1.0f + expf(-a));
    }
}

__global__ __launch_bounds__(256) void channel_scale_kernel(float* __restrict__ x, const float* __restrict__ gate, int HW, int C,
                                                            long long total4) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total4) return;
    const int CQ = C >> 2;
    const int cq = (int)(gid % CQ);
    const long long n = (gid / CQ) / HW;
    float4 v = *reinterpret_cast<float4*>(x + gid * 4);
    const float4 g = *reinterpret_cast<const float4*>(gate + n * C + cq * 4);
    v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
    *reinterpret_cast<float4*>(x + gid * 4) = v;
}

// ------------------------------------------------------------------------------------------------------------- resize
// in (n,h,w) with channel stride in_ctot / offset in_coff, C channels -> out (n,Hout,Wout) channel slice [out_coff, out_coff+C):
// the (H2,W2) bilinear enlargement (align_corners=True) placed at (pad_t, pad_l), zeros around it (F.pad of Up.forward :41-45)
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ in, int h, int w, int C, int in_ctot, int in_coff,
                                                              int H2, int W2, float sy, float sx, int pad_t, int pad_l, int Hout, int Wout,
                                                              float* __restrict__ out, int out_ctot, int out_coff, long long total) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int CQ = C >> 2;
    const int cq = (int)(gid % CQ);
    const long long pix = gid / CQ;
    const int xo = (int)(pix % Wout);
    const long long r = pix / Wout;
    const int yo = (int)(r % Hout);
    const long long n = r / Hout;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    const int y2 = yo - pad_t, x2 = xo - pad_l;
    if ((unsigned)y2 < (unsigned)H2 && (unsigned)x2 < (unsigned)W2) {
        const float fy = sy * (float)y2, fx = sx * (float)x2;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float hy = 1.0f - ly, hx = 1.0f - lx;
        const float* b = in + (size_t)n * h * w * in_ctot + in_coff + cq * 4;
        const float4 v00 = *reinterpret_cast<const float4*>(b + ((size_t)y0 * w + x0) * in_ctot);
        const float4 v01 = *reinterpret_cast<const float4*>(b + ((size_t)y0 * w + x1) * in_ctot);
        const float4 v10 = *reinterpret_cast<const float4*>(b + ((size_t)y1 * w + x0) * in_ctot);
        const float4 v11 = *reinterpret_cast<const float4*>(b + ((size_t)y1 * w + x1) * in_ctot);
        o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
        o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
        o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
        o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
    }
    *reinterpret_cast<float4*>(out + pix * out_ctot + out_coff + cq * 4) = o;
}

// ------------------------------------------------------------------------------------------------------------ max pool
// nn.MaxPool2d(ks, stride, pad) on NHWC maps (the 3x3 / 2 / 1 pool of a torchvision ResNet stem): thread -> (output pixel, channel quad);
// padding cells do not take part (torch pads with -inf)
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ x, int H, int W, int C, int ks, int stride, int pad, int Ho, int Wo,
                                                      long long total, float* __restrict__ out) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int CQ = C >> 2;
    const int cq = (int)(gid % CQ);
    const long long pix = gid / CQ;
    const int wo = (int)(pix % Wo);
    const long long r = pix / Wo;
    const int ho = (int)(r % Ho);
    const long long n = r / Ho;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int kh = 0; kh < ks; ++kh) {
        const int hi = ho * stride - pad + kh;
        if ((unsigned)hi >= (unsigned)H) continue;
        for (int kw = 0; kw < ks; ++kw) {
            const int wi = wo * stride - pad + kw;
            if ((unsigned)wi >= (unsigned)W) continue;
            const float4 v = *reinterpret_cast<const float4*>(x + ((n * H + hi) * (long long)W + wi) * C + cq * 4);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    }
    *reinterpret_cast<float4*>(out + pix * C + cq * 4) = m;
}

// ------------------------------------------------------------------------------------------------------------ softmax
// one wave per pixel: softmax over the first D of `stride` channels, written to out (rows, D)
__global__ __launch_bounds__(256) void softmax_channels_kernel(const float* __restrict__ x, long long rows, int D, int stride,
                                                               float* __restrict__ out) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int ln = threadIdx.x & 63;
    if (row >= rows) return;
    const float* p = x + row * stride;
    float m = -INFINITY;
    for (int c = ln; c < D; c += 64) m = fmaxf(m, p[c]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    float s = 0.f;
    for (int c = ln; c < D; c += 64) s += expf(p[c] - m);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    for (int c = ln; c < D; c += 64) out[row * D + c] = expf(p[c] - m) / s;
}

// --------------------------------------------------------------------------------------------------------- lift + pool
struct LiftCam { float ipr[9], pt[3], comb[9], tr[3]; };     // as LssCam of lss.hip
struct LiftGrid { float lo[3], dx[3]; int nx[3]; };
constexpr float kFix = 4294967296.0f;                          // 2^32

// frustum point -> voxel (the fp32 operation order of get_geometry, airv2x_encoder.py:147-166, and of `.long()` :227)
__device__ __forceinline__ bool lift_voxel(const LiftCam& c, const LiftGrid& g, float px, float py, float pd, int& ix, int& iy, int& iz) {
    const float fx = __fsub_rn(px, c.pt[0]), fy = __fsub_rn(py, c.pt[1]), fz = __fsub_rn(pd, c.pt[2]);
    auto dot3 = [](const float* m, float a, float b, float d) {
        return __fadd_rn(__fadd_rn(__fmul_rn(m[0], a), __fmul_rn(m[1], b)), __fmul_rn(m[2], d));
    };
    const float ux = dot3(c.ipr + 0, fx, fy, fz), uy = dot3(c.ipr + 3, fx, fy, fz), uz = dot3(c.ipr + 6, fx, fy, fz);
    const float vx = __fmul_rn(ux, uz), vy = __fmul_rn(uy, uz), vz = uz;
    const float gx = __fadd_rn(dot3(c.comb + 0, vx, vy, vz), c.tr[0]);
    const float gy = __fadd_rn(dot3(c.comb + 3, vx, vy, vz), c.tr[1]);
    const float gz = __fadd_rn(dot3(c.comb + 6, vx, vy, vz), c.tr[2]);
    const long long lx = (long long)__fdiv_rn(__fsub_rn(gx, g.lo[0]), g.dx[0]);
    const long long ly = (long long)__fdiv_rn(__fsub_rn(gy, g.lo[1]), g.dx[1]);
    const long long lz = (long long)__fdiv_rn(__fsub_rn(gz, g.lo[2]), g.dx[2]);
    if (lx < 0 || lx >= g.nx[0] || ly < 0 || ly >= g.nx[1] || lz < 0 || lz >= g.nx[2]) return false;
    ix = (int)lx; iy = (int)ly; iz = (int)lz;
    return true;
}

__device__ __forceinline__ void lift_add(unsigned long long* a, float4 v) {
    if (v.x != 0.f) atomicAdd(a + 0, (unsigned long long)__float2ll_rn(v.x * kFix));
    if (v.y != 0.f) atomicAdd(a + 1, (unsigned long long)__float2ll_rn(v.y * kFix));
    if (v.z != 0.f) atomicAdd(a + 2, (unsigned long long)__float2ll_rn(v.z * kFix));
    if (v.w != 0.f) atomicAdd(a + 3, (unsigned long long)__float2ll_rn(v.w * kFix));
}

struct DepthBins { float dmin, dmax, bin; int nbins, mode, target; };   // mode 0 UD, 1 LID (utils/camera_utils.py:247-298)

// ground-truth depth (use_depth_gt: CamEncode.get_gt_depth_dist :93-116): the pixel at the centre of every downsample x downsample
// cell carries ONE depth bin; its C features go to that bin's voxel.  LPP = C / 4 lanes per feature pixel.
template <int LPP>
__global__ __launch_bounds__(256) void lift_pool_gt_kernel(const float* __restrict__ feat, const float* __restrict__ imgs, int planes, int H,
                                                           int W, int ds, DepthBins db, const float* __restrict__ frustum,
                                                           const LiftCam* __restrict__ cams, LiftGrid g, int fH, int fW, int cams_per_batch,
                                                           long long npix, int C, unsigned long long* __restrict__ acc) {
    const int t = threadIdx.x % LPP;
    const long long p = (long long)blockIdx.x * (256 / LPP) + threadIdx.x / LPP;
    if (p >= npix) return;
    const int fw = (int)(p % fW);
    const long long r = p / fW;
    const int fh = (int)(r % fH);
    const int cam = (int)(r / fH);
    float d = imgs[(((size_t)cam * planes + 3) * H + (ds / 2 + fh * ds)) * W + (ds / 2 + fw * ds)];
    d = fminf(d, db.dmax);                                                 // torch.clamp_max_ (:103); NaN stays NaN
    float idx;
    if (db.mode == 0) idx = __fdiv_rn(__fsub_rn(d, db.dmin), db.bin);
    else idx = __fadd_rn(-0.5f, __fmul_rn(0.5f, __fsqrt_rn(__fadd_rn(1.0f, __fdiv_rn(__fmul_rn(8.0f, __fsub_rn(d, db.dmin)), db.bin)))));
    const bool bad = (idx < 0.f) || (idx >= (float)db.nbins) || !isfinite(idx);
    if (bad && !db.target) return;                                         // eval: out-of-range depths are masked out (:110-115)
    int bin;
    if (!isfinite(idx)) bin = db.nbins - 1;
    else if (idx < 0.f) bin = 0;
    else if (idx >= (float)db.nbins) bin = db.nbins - 1;
    else bin = (int)idx;
    const int fp = (bin * fH + fh) * fW + fw;
    int ix, iy, iz;
    if (!lift_voxel(cams[cam], g, frustum[3 * fp + 0], frustum[3 * fp + 1], frustum[3 * fp + 2], ix, iy, iz)) return;
    const int b = cam / cams_per_batch;
    const size_t cell = (((size_t)b * g.nx[2] + iz) * g.nx[1] + iy) * g.nx[0] + ix;
    lift_add(acc + cell * C + 4 * t, *reinterpret_cast<const float4*>(feat + (size_t)p * C + 4 * t));
}

// predicted depth: prob (BN, fH, fW, D) softmax rows; every (pixel, bin) adds prob * features (one fp32 product, as :181)
template <int LPP>
__global__ __launch_bounds__(256) void lift_pool_prob_kernel(const float* __restrict__ feat, const float* __restrict__ prob, int D,
                                                             const float* __restrict__ frustum, const LiftCam* __restrict__ cams, LiftGrid g,
                                                             int fH, int fW, int cams_per_batch, long long npts, int C,
                                                             unsigned long long* __restrict__ acc) {
    const int t = threadIdx.x % LPP;
    const long long q = (long long)blockIdx.x * (256 / LPP) + threadIdx.x / LPP;     // (pixel, bin), bin fastest
    if (q >= npts) return;
    const int bin = (int)(q % D);
    const long long p = q / D;
    const int fw = (int)(p % fW);
    const long long r = p / fW;
    const int fh = (int)(r % fH);
    const int cam = (int)(r / fH);
    const int fp = (bin * fH + fh) * fW + fw;
    int ix, iy, iz;
    if (!lift_voxel(cams[cam], g, frustum[3 * fp + 0], frustum[3 * fp + 1], frustum[3 * fp + 2], ix, iy, iz)) return;
    const float pr = prob[p * D + bin];
    float4 v = *reinterpret_cast<const float4*>(feat + (size_t)p * C + 4 * t);
    v.x = __fmul_rn(pr, v.x); v.y = __fmul_rn(pr, v.y); v.z = __fmul_rn(pr, v.z); v.w = __fmul_rn(pr, v.w);
    const int b = cam / cams_per_batch;
    const size_t cell = (((size_t)b * g.nx[2] + iz) * g.nx[1] + iy) * g.nx[0] + ix;
    lift_add(acc + cell * C + 4 * t, v);
}

__global__ __launch_bounds__(256) void lift_finish_kernel(const long long* __restrict__ acc, int B, int nz, int ny, int nx, int C,
                                                          float* __restrict__ out) {
    const size_t n = (size_t)B * nz * ny * nx * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int xx = (int)(r % nx); r /= nx;
        const int yy = (int)(r % ny); r /= ny;
        const int zz = (int)(r % nz);
        const int b = (int)(r / nz);
        out[(((size_t)b * ny + yy) * nx + xx) * ((size_t)nz * C) + (size_t)zz * C + c] = (float)((double)acc[i] * (1.0 / 4294967296.0));
    }
}

// Airv2xBase.fuse_bev (airv2x_base_model.py:167-177) for two modality maps: torch.mean(torch.stack([a, b]), 0) = (a + b) / 2; b == nullptr copies
__global__ __launch_bounds__(256) void mean2_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = a[i];
        if (b) {
            const float4 u = b[i];
            v.x = (v.x + u.x) * 0.5f; v.y = (v.y + u.y) * 0.5f; v.z = (v.z + u.z) * 0.5f; v.w = (v.w + u.w) * 0.5f;
        }
        out[i] = v;
    }
}

inline unsigned blocks_for(long long total) { return (unsigned)((total + 255) / 256); }

}  // namespace

extern "C" int av2x_cam_stem(const float* imgs, int32_t n, int32_t planes, int32_t h, int32_t w, const float* weight, const float* scale,
                             const float* shift, int32_t pad_t, int32_t pad_l, int32_t ho, int32_t wo, float* out, av2x_stream_t stream) {
    if (!imgs || !weight || !scale || !shift || !out) return av2x::fail("av2x_cam_stem: null argument");
    if (n <= 0 || planes < 3 || h <= 0 || w <= 0 || ho <= 0 || wo <= 0) return av2x::fail("av2x_cam_stem: bad sizes");
    const long long npix = (long long)n * ho * wo;
    hipLaunchKernelGGL(cam_stem_kernel, dim3(blocks_for(npix * 8)), dim3(256), 0, av2x::as_stream(stream), imgs, planes, h, w, weight, scale,
                       shift, pad_t, pad_l, ho, wo, npix, out);
    return av2x::check_launch("cam_stem_kernel");
}

extern "C" int av2x_dwconv2d(const float* x, int32_t n, int32_t h, int32_t w, int32_t c, const float* weight, const float* scale,
                             const float* shift, int32_t ks, int32_t stride, int32_t pad_t, int32_t pad_l, int32_t ho, int32_t wo,
                             int32_t act, float* out, av2x_stream_t stream) {
    if (!x || !weight || !scale || !shift || !out) return av2x::fail("av2x_dwconv2d: null argument");
    if (c % 4 || c <= 0 || n <= 0) return av2x::fail("av2x_dwconv2d: c=%d must be a positive multiple of 4", c);
    if (ks != 3 && ks != 5) return av2x::fail("av2x_dwconv2d: ks=%d (3 or 5)", ks);
    if (act != 0 && act != 1 && act != 6) return av2x::fail("av2x_dwconv2d: activation %d (0 none, 1 ReLU, 6 swish)", act);
    if ((ho - 1) * stride - pad_t >= h || (wo - 1) * stride - pad_l >= w) return av2x::fail("av2x_dwconv2d: output larger than the padded input");
    const long long total = (long long)n * ho * wo * (c / 4);
    hipStream_t st = av2x::as_stream(stream);
    if (ks == 3)
        hipLaunchKernelGGL(dwconv_kernel<3>, dim3(blocks_for(total)), dim3(256), 0, st, x, h, w, c, weight, scale, shift, stride, pad_t, pad_l, ho,
                           wo, total, act, out);
    else
        hipLaunchKernelGGL(dwconv_kernel<5>, dim3(blocks_for(total)), dim3(256), 0, st, x, h, w, c, weight, scale, shift, stride, pad_t, pad_l, ho,
                           wo, total, act, out);
    return av2x::check_launch("dwconv_kernel");
}

extern "C" int32_t av2x_se_slabs(int32_t hw) { return hw >= 8192 ? 32 : (hw >= 1024 ? 8 : 1); }

extern "C" int av2x_squeeze_excite(float* x, int32_t n, int32_t hw, int32_t c, const float* w_reduce, const float* b_reduce, int32_t c_se,
                                   const float* w_expand, const float* b_expand, float* workspace, int32_t apply, av2x_stream_t stream) {
    if (!x || !w_reduce || !b_reduce || !w_expand || !b_expand || !workspace) return av2x::fail("av2x_squeeze_excite: null argument");
    if (c % 4 || c <= 0 || c > 8192 || c_se <= 0 || c_se > 256 || n <= 0 || hw <= 0) return av2x::fail("av2x_squeeze_excite: bad sizes");
    hipStream_t st = av2x::as_stream(stream);
    const int S = av2x_se_slabs(hw);
    float* partial = workspace;                       // [n][S][c]
    float* gate = workspace + (size_t)n * S * c;      // [n][c]
    hipLaunchKernelGGL(se_gap_kernel, dim3((c + 63) / 64, S, n), dim3(256), 0, st, x, hw, c, S, partial);
    hipLaunchKernelGGL(se_fc_kernel, dim3(n), dim3(256), (size_t)(c + c_se) * sizeof(float), st, partial, S, hw, c, c_se, w_reduce, b_reduce,
                       w_expand, b_expand, gate);
    if (apply) {
        const long long total4 = (long long)n * hw * (c / 4);
        hipLaunchKernelGGL(channel_scale_kernel, dim3(blocks_for(total4)), dim3(256), 0, st, x, gate, hw, c, total4);
    }
    return av2x::check_launch("squeeze_excite");
}

extern "C" uint64_t av2x_squeeze_excite_workspace_bytes(int32_t n, int32_t hw, int32_t c) {
    return (uint64_t)n * c * (av2x_se_slabs(hw) + 1) * sizeof(float);
}

extern "C" int av2x_resize_bilinear(const float* in, int32_t n, int32_t h, int32_t w, int32_t c, int32_t in_ctot, int32_t in_coff,
                                    int32_t h2, int32_t w2, int32_t pad_t, int32_t pad_l, int32_t hout, int32_t wout, float* out,
                                    int32_t out_ctot, int32_t out_coff, av2x_stream_t stream) {
    if (!in || !out) return av2x::fail("av2x_resize_bilinear: null argument");
    if (c % 4 || in_ctot % 4 || in_coff % 4 || out_ctot % 4 || out_coff % 4 || c <= 0) return av2x::fail("av2x_resize_bilinear: channel counts / offsets must be multiples of 4");
    if (n <= 0 || h <= 0 || w <= 0 || h2 <= 0 || w2 <= 0 || pad_t < 0 || pad_l < 0 || pad_t + h2 > hout || pad_l + w2 > wout)
        return av2x::fail("av2x_resize_bilinear: bad sizes");
    // align_corners=True source scale, as ATen's area_pixel_compute_scale: (in - 1) / (out - 1), 0 for a single output row
    const float sy = h2 > 1 ? (float)(h - 1) / (float)(h2 - 1) : 0.f;
    const float sx = w2 > 1 ? (float)(w - 1) / (float)(w2 - 1) : 0.f;
    const long long total = (long long)n * hout * wout * (c / 4);
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(blocks_for(total)), dim3(256), 0, av2x::as_stream(stream), in, h, w, c, in_ctot, in_coff, h2,
                       w2, sy, sx, pad_t, pad_l, hout, wout, out, out_ctot, out_coff, total);
    return av2x::check_launch("resize_bilinear_kernel");
}

extern "C" int av2x_maxpool2d(const float* x, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ks, int32_t stride, int32_t pad, int32_t ho,
                              int32_t wo, float* out, av2x_stream_t stream) {
    if (n == 0) return 0;
    if (!x || !out) return av2x::fail("av2x_maxpool2d: null argument");
    if (n < 0 || h <= 0 || w <= 0 || c <= 0 || c % 4 || ks <= 0 || stride <= 0 || pad < 0 || 2 * pad > ks)
        return av2x::fail("av2x_maxpool2d: bad sizes (c %% 4 == 0, pad <= ks / 2)");
    if (ho != (h + 2 * pad - ks) / stride + 1 || wo != (w + 2 * pad - ks) / stride + 1) return av2x::fail("av2x_maxpool2d: output size mismatch");
    const long long total = (long long)n * ho * wo * (c / 4);
    hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, av2x::as_stream(stream), x, h, w, c, ks, stride, pad, ho,
                       wo, total, out);
    return av2x::check_launch("maxpool_kernel");
}

extern "C" int av2x_softmax_channels(const float* x, int64_t rows, int32_t d, int32_t stride, float* out, av2x_stream_t stream) {
    if (!x || !out || rows <= 0 || d <= 0 || stride < d) return av2x::fail("av2x_softmax_channels: bad argument");
    hipLaunchKernelGGL(softmax_channels_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, av2x::as_stream(stream), x, (long long)rows, d,
                       stride, out);
    return av2x::check_launch("softmax_channels_kernel");
}

extern "C" int av2x_lss_lift_pool(const float* feat, const float* prob, const float* imgs, int32_t planes, int32_t img_h, int32_t img_w,
                                  int32_t downsample, const float* depth3, int32_t nbins, int32_t depth_mode, int32_t target,
                                  const float* frustum, const float* cam_params, int32_t b, int32_t n_cams, int32_t fh, int32_t fw,
                                  int32_t c, const float* lo3, const float* dx3, const int32_t* nx3, void* workspace, float* out,
                                  av2x_stream_t stream) {
    if (!feat || !frustum || !cam_params || !lo3 || !dx3 || !nx3 || !workspace || !out) return av2x::fail("av2x_lss_lift_pool: null argument");
    if ((prob == nullptr) == (imgs == nullptr)) return av2x::fail("av2x_lss_lift_pool: exactly one of prob (predicted depth) / imgs (ground-truth depth plane)");
    if (b <= 0 || n_cams <= 0 || fh <= 0 || fw <= 0 || nbins <= 0) return av2x::fail("av2x_lss_lift_pool: bad sizes");
    if (c != 32 && c != 64 && c != 128) return av2x::fail("av2x_lss_lift_pool: c=%d (32, 64 or 128 feature channels)", c);
    LiftGrid g;
    for (int i = 0; i < 3; ++i) { g.lo[i] = lo3[i]; g.dx[i] = dx3[i]; g.nx[i] = nx3[i]; }
    hipStream_t st = av2x::as_stream(stream);
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(workspace);
    const size_t cells = (size_t)b * g.nx[0] * g.nx[1] * g.nx[2];
    hipError_t e = hipMemsetAsync(acc, 0, cells * c * 8ull, st);
    if (e != hipSuccess) return av2x::fail("av2x_lss_lift_pool: memset: %s", hipGetErrorString(e));
    const LiftCam* cams = reinterpret_cast<const LiftCam*>(cam_params);
    const long long npix = (long long)b * n_cams * fh * fw;
    if (imgs) {
        if (planes < 4 || !depth3) return av2x::fail("av2x_lss_lift_pool: the ground-truth depth is plane 3 of a (>= 4)-plane image");
        if (downsample <= 0 || downsample / 2 + (fh - 1) * downsample >= img_h || downsample / 2 + (fw - 1) * downsample >= img_w)
            return av2x::fail("av2x_lss_lift_pool: feature map %dx%d x downsample %d exceeds the image %dx%d", fh, fw, downsample, img_h, img_w);
        if (depth_mode != 0 && depth_mode != 1) return av2x::fail("av2x_lss_lift_pool: depth_mode %d (0 UD, 1 LID)", depth_mode);
        DepthBins db{depth3[0], depth3[1], depth3[2], nbins, depth_mode, target};
#define AV2X_LIFT_GT(LPP)                                                                                                              \
        hipLaunchKernelGGL(lift_pool_gt_kernel<LPP>, dim3((unsigned)((npix + (256 / LPP) - 1) / (256 / LPP))), dim3(256), 0, st, feat, imgs, \
                           planes, img_h, img_w, downsample, db, frustum, cams, g, fh, fw, n_cams, npix, c, acc)
        if (c == 32) AV2X_LIFT_GT(8); else if (c == 64) AV2X_LIFT_GT(16); else AV2X_LIFT_GT(32);
#undef AV2X_LIFT_GT
    } else {
        const long long npts = npix * nbins;
#define AV2X_LIFT_PR(LPP)                                                                                                              \
        hipLaunchKernelGGL(lift_pool_prob_kernel<LPP>, dim3((unsigned)((npts + (256 / LPP) - 1) / (256 / LPP))), dim3(256), 0, st, feat, prob, \
                           nbins, frustum, cams, g, fh, fw, n_cams, npts, c, acc)
        if (c == 32) AV2X_LIFT_PR(8); else if (c == 64) AV2X_LIFT_PR(16); else AV2X_LIFT_PR(32);
#undef AV2X_LIFT_PR
    }
    hipLaunchKernelGGL(lift_finish_kernel, dim3(2048), dim3(256), 0, st, reinterpret_cast<const long long*>(acc), b, g.nx[2], g.nx[1], g.nx[0], c, out);
    return av2x::check_launch("lift_pool");
}

// ---- training: the adjoint of the ground-truth-depth lift (what torch autograd gives the reference for voxel_pooling's index_put /
// cumsum trick, airv2x_encoder.py:208-275, and the one-hot product of CamEncode.forward :170-175): every feature pixel went to ONE voxel
// (or none), so its gradient is a gather from that voxel -- no atomics, run-to-run identical.
template <int LPP>
__global__ __launch_bounds__(256) void lift_pool_gt_backward_kernel(const float* __restrict__ dout, const float* __restrict__ imgs, int planes, int H,
                                                                    int W, int ds, DepthBins db, const float* __restrict__ frustum,
                                                                    const LiftCam* __restrict__ cams, LiftGrid g, int fH, int fW, int cams_per_batch,
                                                                    long long npix, int C, float* __restrict__ dfeat) {
    const int t = threadIdx.x % LPP;
    const long long p = (long long)blockIdx.x * (256 / LPP) + threadIdx.x / LPP;
    if (p >= npix) return;
    const int fw = (int)(p % fW);
    const long long r = p / fW;
    const int fh = (int)(r % fH);
    const int cam = (int)(r / fH);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    float d = imgs[(((size_t)cam * planes + 3) * H + (ds / 2 + fh * ds)) * W + (ds / 2 + fw * ds)];
    d = fminf(d, db.dmax);
    float idx;
    if (db.mode == 0) idx = __fdiv_rn(__fsub_rn(d, db.dmin), db.bin);
    else idx = __fadd_rn(-0.5f, __fmul_rn(0.5f, __fsqrt_rn(__fadd_rn(1.0f, __fdiv_rn(__fmul_rn(8.0f, __fsub_rn(d, db.dmin)), db.bin)))));
    const bool bad = (idx < 0.f) || (idx >= (float)db.nbins) || !isfinite(idx);
    if (!(bad && !db.target)) {
        int bin;
        if (!isfinite(idx)) bin = db.nbins - 1;
        else if (idx < 0.f) bin = 0;
        else if (idx >= (float)db.nbins) bin = db.nbins - 1;
        else bin = (int)idx;
        const int fp = (bin * fH + fh) * fW + fw;
        int ix, iy, iz;
        if (lift_voxel(cams[cam], g, frustum[3 * fp + 0], frustum[3 * fp + 1], frustum[3 * fp + 2], ix, iy, iz)) {
            const int b = cam / cams_per_batch;
            const size_t cell = (((size_t)b * g.nx[2] + iz) * g.nx[1] + iy) * g.nx[0] + ix;
            o = *reinterpret_cast<const float4*>(dout + cell * C + 4 * t);
        }
    }
    *reinterpret_cast<float4*>(dfeat + (size_t)p * C + 4 * t) = o;
}

extern "C" int av2x_lss_lift_pool_backward(const float* dout, const float* imgs, int32_t planes, int32_t img_h, int32_t img_w, int32_t downsample,
                                           const float* depth3, int32_t nbins, int32_t depth_mode, int32_t target, const float* frustum,
                                           const float* cam_params, int32_t b, int32_t n_cams, int32_t fh, int32_t fw, int32_t c, const float* lo3,
                                           const float* dx3, const int32_t* nx3, float* dfeat, av2x_stream_t stream) {
    if (!dout || !imgs || !depth3 || !frustum || !cam_params || !lo3 || !dx3 || !nx3 || !dfeat) return av2x::fail("av2x_lss_lift_pool_backward: null argument");
    if (b <= 0 || n_cams <= 0 || fh <= 0 || fw <= 0 || nbins <= 0) return av2x::fail("av2x_lss_lift_pool_backward: bad sizes");
    if (c != 32 && c != 64 && c != 128) return av2x::fail("av2x_lss_lift_pool_backward: c=%d (32, 64 or 128 feature channels)", c);
    if (planes < 4) return av2x::fail("av2x_lss_lift_pool_backward: the ground-truth depth is plane 3 of a (>= 4)-plane image");
    if (downsample <= 0 || downsample / 2 + (fh - 1) * downsample >= img_h || downsample / 2 + (fw - 1) * downsample >= img_w)
        return av2x::fail("av2x_lss_lift_pool_backward: feature map %dx%d x downsample %d exceeds the image %dx%d", fh, fw, downsample, img_h, img_w);
    if (depth_mode != 0 && depth_mode != 1) return av2x::fail("av2x_lss_lift_pool_backward: depth_mode %d (0 UD, 1 LID)", depth_mode);
    LiftGrid g;
    for (int i = 0; i < 3; ++i) { g.lo[i] = lo3[i]; g.dx[i] = dx3[i]; g.nx[i] = nx3[i]; }
    hipStream_t st = av2x::as_stream(stream);
    const LiftCam* cams = reinterpret_cast<const LiftCam*>(cam_params);
    const long long npix = (long long)b * n_cams * fh * fw;
    DepthBins db{depth3[0], depth3[1], depth3[2], nbins, depth_mode, target};
#define AV2X_LIFT_GTB(LPP)                                                                                                                       \
    hipLaunchKernelGGL(lift_pool_gt_backward_kernel<LPP>, dim3((unsigned)((npix + (256 / LPP) - 1) / (256 / LPP))), dim3(256), 0, st, dout, imgs, \
                       planes, img_h, img_w, downsample, db, frustum, cams, g, fh, fw, n_cams, npix, c, dfeat)
    if (c == 32) AV2X_LIFT_GTB(8); else if (c == 64) AV2X_LIFT_GTB(16); else AV2X_LIFT_GTB(32);
#undef AV2X_LIFT_GTB
    return av2x::check_launch("lift_pool_gt_backward_kernel");
}

// ---- training: the adjoint of the predicted-depth lift (CamEncode.forward :176-186 + voxel_pooling): out[voxel(p, d)] += prob[p][d] * feat[p],
// so dfeat[p] = sum_d prob[p][d] * dout[voxel(p, d)] and dprob[p][d] = <feat[p], dout[voxel(p, d)]> -- gathers, no atomics.  One group of
// LPP lanes per feature pixel walks its D bins; the dot product is reduced over the group's lanes with xor shuffles (fixed order).
template <int LPP>
__global__ __launch_bounds__(256) void lift_pool_prob_backward_kernel(const float* __restrict__ dout, const float* __restrict__ feat,
                                                                      const float* __restrict__ prob, int D, const float* __restrict__ frustum,
                                                                      const LiftCam* __restrict__ cams, LiftGrid g, int fH, int fW, int cams_per_batch,
                                                                      long long npix, int C, float* __restrict__ dfeat, float* __restrict__ dprob) {
    const int t = threadIdx.x % LPP;
    const long long p = (long long)blockIdx.x * (256 / LPP) + threadIdx.x / LPP;
    if (p >= npix) return;                               // whole groups leave together (256 % LPP == 0): the shuffles below stay inside a group
    const int fw = (int)(p % fW);
    const long long r = p / fW;
    const int fh = (int)(r % fH);
    const int cam = (int)(r / fH);
    const int b = cam / cams_per_batch;
    const float4 f = *reinterpret_cast<const float4*>(feat + (size_t)p * C + 4 * t);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int bin = 0; bin < D; ++bin) {
        const int fp = (bin * fH + fh) * fW + fw;
        int ix, iy, iz;
        float s = 0.f;
        if (lift_voxel(cams[cam], g, frustum[3 * fp + 0], frustum[3 * fp + 1], frustum[3 * fp + 2], ix, iy, iz)) {
            const size_t cell = (((size_t)b * g.nx[2] + iz) * g.nx[1] + iy) * g.nx[0] + ix;
            const float4 d4 = *reinterpret_cast<const float4*>(dout + cell * C + 4 * t);
            const float pr = prob[p * D + bin];
            o.x = fmaf(pr, d4.x, o.x); o.y = fmaf(pr, d4.y, o.y); o.z = fmaf(pr, d4.z, o.z); o.w = fmaf(pr, d4.w, o.w);
            s = fmaf(f.x, d4.x, fmaf(f.y, d4.y, fmaf(f.z, d4.z, f.w * d4.w)));
        }
#pragma unroll
        for (int off = LPP / 2; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        if (t == 0) dprob[p * D + bin] = s;
    }
    *reinterpret_cast<float4*>(dfeat + (size_t)p * C + 4 * t) = o;
}

// dlogit = prob * (dprob - <prob, dprob>): one wave per pixel (rows, D)
__global__ __launch_bounds__(256) void softmax_channels_backward_kernel(const float* __restrict__ prob, const float* __restrict__ dprob, long long rows,
                                                                        int D, int stride, float* __restrict__ dlogit) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int ln = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.f;
    for (int c = ln; c < D; c += 64) s = fmaf(prob[row * D + c], dprob[row * D + c], s);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    for (int c = ln; c < stride; c += 64) dlogit[row * stride + c] = c < D ? prob[row * D + c] * (dprob[row * D + c] - s) : 0.f;
}

extern "C" int av2x_lss_lift_pool_prob_backward(const float* dout, const float* feat, const float* prob, int32_t nbins, const float* frustum,
                                                const float* cam_params, int32_t b, int32_t n_cams, int32_t fh, int32_t fw, int32_t c,
                                                const float* lo3, const float* dx3, const int32_t* nx3, float* dfeat, float* dprob,
                                                av2x_stream_t stream) {
    if (!dout || !feat || !prob || !frustum || !cam_params || !lo3 || !dx3 || !nx3 || !dfeat || !dprob)
        return av2x::fail("av2x_lss_lift_pool_prob_backward: null argument");
    if (b <= 0 || n_cams <= 0 || fh <= 0 || fw <= 0 || nbins <= 0) return av2x::fail("av2x_lss_lift_pool_prob_backward: bad sizes");
    if (c != 32 && c != 64 && c != 128) return av2x::fail("av2x_lss_lift_pool_prob_backward: c=%d (32, 64 or 128 feature channels)", c);
    LiftGrid g;
    for (int i = 0; i < 3; ++i) { g.lo[i] = lo3[i]; g.dx[i] = dx3[i]; g.nx[i] = nx3[i]; }
    hipStream_t st = av2x::as_stream(stream);
    const LiftCam* cams = reinterpret_cast<const LiftCam*>(cam_params);
    const long long npix = (long long)b * n_cams * fh * fw;
#define AV2X_LIFT_PRB(LPP)                                                                                                                       \
    hipLaunchKernelGGL(lift_pool_prob_backward_kernel<LPP>, dim3((unsigned)((npix + (256 / LPP) - 1) / (256 / LPP))), dim3(256), 0, st, dout, feat, \
                       prob, nbins, frustum, cams, g, fh, fw, n_cams, npix, c, dfeat, dprob)
    if (c == 32) AV2X_LIFT_PRB(8); else if (c == 64) AV2X_LIFT_PRB(16); else AV2X_LIFT_PRB(32);
#undef AV2X_LIFT_PRB
    return av2x::check_launch("lift_pool_prob_backward_kernel");
}

// prob, dprob (rows, d) -> dlogit (rows, stride): the gradient of av2x_softmax_channels with respect to its first d input channels (the rest: 0)
extern "C" int av2x_softmax_channels_backward(const float* prob, const float* dprob, int64_t rows, int32_t d, int32_t stride, float* dlogit,
                                              av2x_stream_t stream) {
    if (!prob || !dprob || !dlogit || rows <= 0 || d <= 0 || stride < d) return av2x::fail("av2x_softmax_channels_backward: bad argument");
    hipLaunchKernelGGL(softmax_channels_backward_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, av2x::as_stream(stream), prob, dprob,
                       (long long)rows, d, stride, dlogit);
    return av2x::check_launch("softmax_channels_backward_kernel");
}

extern "C" int av2x_mean2(const float* a, const float* b, float* out, uint64_t n, av2x_stream_t stream) {
    if (!a || !out || n % 4) return av2x::fail("av2x_mean2: null argument or n %% 4 != 0");
    if (n == 0) return 0;
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(mean2_kernel, dim3((unsigned)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192)), dim3(256), 0, av2x::as_stream(stream),
                       reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b), reinterpret_cast<float4*>(out), n4);
    return av2x::check_launch("mean2_kernel");
}
