// Training-mode pieces of the Where2Comm path: what torch autograd does for the reference in
// tools/train.py:220-247, as explicit forward / backward kernels.  All HBM-bound, all deterministic (fixed-order fp64
// partial sums, no floating-point atomics):
//   av2x_bn_stats             BatchNorm2d / 1d in train mode: per-channel batch mean and biased variance of an NHWC map
//                             (base_bev_backbone.py:52,65,83 -- eps 1e-3, momentum 0.01)
//   av2x_affine_act           y = act(z * scale[c] + shift[c])  (the normalisation itself, scale = gamma * rstd)
//   av2x_bn_backward          d gamma, d beta and dz of y = relu(gamma * (z - mean) * rstd + beta) in two passes over z
//   av2x_pixel_attn_backward  gradient of AttentionFusion (where2comm_fuse.py:152-164, ego row) w.r.t. every agent's map
//   av2x_pillar_moments       S = sum of the augmented 10-vectors, F = sum of their outer products over ALL 32 rows of all
//                             pillars: the batch statistics of PFNLayer's BatchNorm1d follow from them exactly
//                             (mean = W S / N, E[lin^2] = diag(W F W^T) / N; airv2x_pillar_vfe.py:27-49)
//   av2x_pillar_vfe_backward  gathers the canvas gradient at every pillar (PointPillarScatter backward), routes it to the
//                             arg-max point of every channel (max over 32, ReLU) and accumulates
//                             G[c][k] = sum g feat_k, d beta, d gamma -- the BatchNorm1d / Linear gradients follow on the
//                             host side from G, S, F (opencood_iface/train_ops.py)
#include "av2x_common.hpp"

namespace {

constexpr int kSlabRows = 128;

// ---------------------------------------------------------------------------------------------------- per-channel moments
// part layout: [slab][2][c] doubles.  MODE 0: (sum z, sum z^2).  MODE 1: (sum g, sum g * xhat) with
// g = dy * [act ? z * scale + shift > 0 : 1], xhat = (z - mean) * rstd.
// HBM-bound: one pass over z (and dy).  c = 4 * 2^k: a thread owns 4 channels (16-byte loads) of every (256 / (c/4))-th row of
// its 128-row slab -- <= 32 fp32 terms per partial, then fp64 across the row lanes, the slabs (stage 2) in a fixed order.
template <int MODE>
__global__ __launch_bounds__(256) void moments_stage1(const float* __restrict__ z, const float* __restrict__ dy, size_t rows,
                                                      int c, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                      double* __restrict__ part) {
    __shared__ float4 red[2][256];
    const size_t r0 = (size_t)blockIdx.x * kSlabRows, r1 = r0 + kSlabRows < rows ? r0 + kSlabRows : rows;
    const int tid = threadIdx.x;
    double* out = part + (size_t)blockIdx.x * 2 * c;
    const int c4 = c >> 2;
    if ((c & 3) == 0 && c4 <= 256 && 256 % c4 == 0) {
        const int q = tid % c4, rl = tid / c4, rstep = 256 / c4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        float4 mu = a, rs = a, sc = a, sh = a;
        if (MODE == 1) {
            mu = reinterpret_cast<const float4*>(mean)[q]; rs = reinterpret_cast<const float4*>(rstd)[q];
            sc = reinterpret_cast<const float4*>(scale)[q]; sh = reinterpret_cast<const float4*>(shift)[q];
        }
        const float4* z4 = reinterpret_cast<const float4*>(z);
        const float4* d4 = reinterpret_cast<const float4*>(dy);
#pragma unroll 4
        for (size_t r = r0 + rl; r < r1; r += rstep) {
            const float4 v = z4[r * c4 + q];
            if (MODE == 0) {
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
                b.x = fmaf(v.x, v.x, b.x); b.y = fmaf(v.y, v.y, b.y); b.z = fmaf(v.z, v.z, b.z); b.w = fmaf(v.w, v.w, b.w);
            } else {
                float4 g = d4[r * c4 + q];
                if (act) {
                    if (!(fmaf(v.x, sc.x, sh.x) > 0.f)) g.x = 0.f;
                    if (!(fmaf(v.y, sc.y, sh.y) > 0.f)) g.y = 0.f;
                    if (!(fmaf(v.z, sc.z, sh.z) > 0.f)) g.z = 0.f;
                    if (!(fmaf(v.w, sc.w, sh.w) > 0.f)) g.w = 0.f;
                }
                a.x += g.x; a.y += g.y; a.z += g.z; a.w += g.w;
                b.x = fmaf(g.x, (v.x - mu.x) * rs.x, b.x); b.y = fmaf(g.y, (v.y - mu.y) * rs.y, b.y);
                b.z = fmaf(g.z, (v.z - mu.z) * rs.z, b.z); b.w = fmaf(g.w, (v.w - mu.w) * rs.w, b.w);
            }
        }
        red[0][tid] = a;
        red[1][tid] = b;
        __syncthreads();
        if (tid < c4) {
            double sa[4] = {0.0, 0.0, 0.0, 0.0}, sb[4] = {0.0, 0.0, 0.0, 0.0};
            for (int k = 0; k < rstep; ++k) {
                const float4 x = red[0][k * c4 + tid], y = red[1][k * c4 + tid];
                sa[0] += x.x; sa[1] += x.y; sa[2] += x.z; sa[3] += x.w;
                sb[0] += y.x; sb[1] += y.y; sb[2] += y.z; sb[3] += y.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { out[4 * tid + e] = sa[e]; out[c + 4 * tid + e] = sb[e]; }
        }
    } else {
        for (int ch = tid; ch < c; ch += 256) {
            double a = 0.0, b = 0.0;
            float mu = 0.f, rs = 0.f, sc = 0.f, sh = 0.f;
            if (MODE == 1) { mu = mean[ch]; rs = rstd[ch]; sc = scale[ch]; sh = shift[ch]; }
            for (size_t r = r0; r < r1; ++r) {
                const float v = z[r * c + ch];
                if (MODE == 0) {
                    a += (double)v;
                    b += (double)v * (double)v;
                } else {
                    float g = dy[r * c + ch];
                    if (act && !(fmaf(v, sc, sh) > 0.f)) g = 0.f;
                    a += (double)g;
                    b += (double)(g * ((v - mu) * rs));
                }
            }
            out[ch] = a;
            out[c + ch] = b;
        }
    }
}

// Sums the slab partials: a workgroup owns 2 channels, 128 lanes walk the slabs (lane l takes slabs l, l + 128, ...), then the
// lane sums are added in a fixed two-level order (16 groups of 8 lanes, then the 16 groups) -- fp64, bit-reproducible.
// MODE 0: out0 = mean, out1 = biased variance.  MODE 1: out0 = sum g (d beta), out1 = sum g xhat (d gamma).  MODE 2: out0 = the
// first sum only.
template <int MODE>
__global__ __launch_bounds__(256) void moments_stage2(const double* __restrict__ part, int nslabs, int c, double n,
                                                      float* __restrict__ out0, float* __restrict__ out1) {
    __shared__ double red[2][128][2];
    __shared__ double red2[2][16][2];
    const int j = threadIdx.x & 1, sl = threadIdx.x >> 1;
    const int ch = blockIdx.x * 2 + j;
    double a = 0.0, b = 0.0;
    if (ch < c) {
        for (int k = sl; k < nslabs; k += 128) {
            a += part[(size_t)k * 2 * c + ch];
            b += part[(size_t)k * 2 * c + c + ch];
        }
    }
    red[0][sl][j] = a;
    red[1][sl][j] = b;
    __syncthreads();
    if (sl < 16) {
        a = 0.0; b = 0.0;
        for (int k = 0; k < 8; ++k) { a += red[0][sl * 8 + k][j]; b += red[1][sl * 8 + k][j]; }
        red2[0][sl][j] = a;
        red2[1][sl][j] = b;
    }
    __syncthreads();
    if (sl == 0 && ch < c) {
        a = 0.0; b = 0.0;
        for (int k = 0; k < 16; ++k) { a += red2[0][k][j]; b += red2[1][k][j]; }
        if (MODE == 0) {
            const double m = a / n;
            double v = b / n - m * m;
            if (v < 0.0) v = 0.0;
            out0[ch] = (float)m;
            out1[ch] = (float)v;
        } else if (MODE == 1) {
            out0[ch] = (float)a;
            out1[ch] = (float)b;
        } else {
            out0[ch] = (float)a;      // MODE 2: the plain column sum
        }
    }
}

// One launch per BatchNorm layer for everything torch would do with ~12 tiny kernels: rstd = 1 / sqrt(var + eps),
// scale = gamma * rstd, shift = beta - mean * scale, and nn.BatchNorm's train-mode side effect applied `times` times with
// the same batch statistics: running = (1 - momentum) * running + momentum * (mean | var * n / (n - 1)).
__global__ void bn_finalize_kernel(const float* __restrict__ mean, const float* __restrict__ var, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, int c, float eps, double count, float momentum, int times,
                                   float* __restrict__ rstd, float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, long long* __restrict__ nbt) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch == 0 && nbt) nbt[0] += times;
    if (ch >= c) return;
    const float m = mean[ch], v = var[ch];
    const float r = 1.0f / sqrtf(v + eps);
    const float sc = gamma[ch] * r;
    rstd[ch] = r;
    scale[ch] = sc;
    shift[ch] = beta[ch] - m * sc;
    if (running_mean && running_var) {
        const float unb = (float)((double)v * (count / (count > 1.0 ? count - 1.0 : 1.0)));
        float rm = running_mean[ch], rv = running_var[ch];
        for (int t = 0; t < times; ++t) {
            rm = rm * (1.0f - momentum) + momentum * m;
            rv = rv * (1.0f - momentum) + momentum * unb;
        }
        running_mean[ch] = rm;
        running_var[ch] = rv;
    }
}

// moments_stage2<0> + bn_finalize_kernel in one launch (av2x_bn_train_forward): the same two-level fp64 sums, mean / variance rounded to
// fp32 and the fold computed FROM the rounded values -- bit-identical to the two kernels run one after the other.
__global__ __launch_bounds__(256) void moments_finalize_kernel(const double* __restrict__ part, int nslabs, int c, double n,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                               float momentum, int times, float* __restrict__ stats5,
                                                               float* __restrict__ running_mean, float* __restrict__ running_var,
                                                               long long* __restrict__ nbt) {
    __shared__ double red[2][128][2];
    __shared__ double red2[2][16][2];
    const int j = threadIdx.x & 1, sl = threadIdx.x >> 1;
    const int ch = blockIdx.x * 2 + j;
    double a = 0.0, b = 0.0;
    if (ch < c) {
        for (int k = sl; k < nslabs; k += 128) {
            a += part[(size_t)k * 2 * c + ch];
            b += part[(size_t)k * 2 * c + c + ch];
        }
    }
    red[0][sl][j] = a;
    red[1][sl][j] = b;
    __syncthreads();
    if (sl < 16) {
        a = 0.0; b = 0.0;
        for (int k = 0; k < 8; ++k) { a += red[0][sl * 8 + k][j]; b += red[1][sl * 8 + k][j]; }
        red2[0][sl][j] = a;
        red2[1][sl][j] = b;
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) nbt[0] += times;
    if (sl == 0 && ch < c) {
        a = 0.0; b = 0.0;
        for (int k = 0; k < 16; ++k) { a += red2[0][k][j]; b += red2[1][k][j]; }
        const double md = a / n;
        double vd = b / n - md * md;
        if (vd < 0.0) vd = 0.0;
        const float m = (float)md, v = (float)vd;
        const float r = 1.0f / sqrtf(v + eps);
        const float sc = gamma[ch] * r;
        stats5[ch] = m;
        stats5[c + ch] = v;
        stats5[2 * (size_t)c + ch] = r;
        stats5[3 * (size_t)c + ch] = sc;
        stats5[4 * (size_t)c + ch] = beta[ch] - m * sc;
        if (running_mean && running_var) {
            const float unb = (float)((double)v * (n / (n > 1.0 ? n - 1.0 : 1.0)));
            float rm = running_mean[ch], rv = running_var[ch];
            for (int t = 0; t < times; ++t) {
                rm = rm * (1.0f - momentum) + momentum * m;
                rv = rv * (1.0f - momentum) + momentum * unb;
            }
            running_mean[ch] = rm;
            running_var[ch] = rv;
        }
    }
}

__global__ __launch_bounds__(256) void affine_act_kernel(const float4* __restrict__ z, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, size_t n4, int c4, int act,
                                                         float4* __restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % c4);
        const float4 sc = scale ? reinterpret_cast<const float4*>(scale)[q] : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 sh = shift ? reinterpret_cast<const float4*>(shift)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 v = z[i];
        float4 r;
        r.x = fmaf(v.x, sc.x, sh.x); r.y = fmaf(v.y, sc.y, sh.y); r.z = fmaf(v.z, sc.z, sh.z); r.w = fmaf(v.w, sc.w, sh.w);
        if (act) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
        y[i] = r;
    }
}

// dz = scale * (g - d beta / N - xhat * d gamma / N)
__global__ __launch_bounds__(256) void bn_backward_apply_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                const float* __restrict__ dbeta, const float* __restrict__ dgamma,
                                                                size_t total, int c, float inv_n, int act, float* __restrict__ dz) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % c);
        const float v = z[i];
        float g = dy[i];
        if (act && !(fmaf(v, scale[ch], shift[ch]) > 0.f)) g = 0.f;
        const float xh = (v - mean[ch]) * rstd[ch];
        dz[i] = scale[ch] * (g - dbeta[ch] * inv_n - xh * (dgamma[ch] * inv_n));
    }
}

// ---------------------------------------------------------------------------------------------------- pixel attention
constexpr int kMaxAgentsT = 32;
struct AgentPtrsT { const float* p[kMaxAgentsT]; };
struct AgentOutPtrsT { float* p[kMaxAgentsT]; };

// out = sum_j p_j x_j, p = softmax_j(x0 . xj / sqrt(C)).  16 lanes per pixel (as the forward kernel).
//   d x_j = p_j dout + ds_j x_0,   ds_j = p_j (dout . x_j - sum_k p_k dout . x_k) / sqrt(C),   d x_0 += sum_j ds_j x_j
__global__ __launch_bounds__(256) void pixel_attn_backward_kernel(const AgentPtrsT ap, const AgentOutPtrsT dp, int n_agents, int hw,
                                                                  int c, float sqrt_c, const float* __restrict__ dout) {
    __shared__ float prob[16][kMaxAgentsT];
    __shared__ float dsc[16][kMaxAgentsT];
    const int t = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int pix = blockIdx.x * 16 + g;
    if (pix >= hw) return;   // whole 16-lane groups leave together; LDS rows are private to a group (no workgroup barrier)
    const size_t base = (size_t)pix * c;
    float mx = -INFINITY;
    for (int j = 0; j < n_agents; ++j) {
        float dot = 0.f, dd = 0.f;
        for (int ch = 4 * t; ch < c; ch += 64) {
            const float4 q = *reinterpret_cast<const float4*>(ap.p[0] + base + ch);
            const float4 x = *reinterpret_cast<const float4*>(ap.p[j] + base + ch);
            const float4 d = *reinterpret_cast<const float4*>(dout + base + ch);
            dot = fmaf(q.x, x.x, dot); dot = fmaf(q.y, x.y, dot); dot = fmaf(q.z, x.z, dot); dot = fmaf(q.w, x.w, dot);
            dd = fmaf(d.x, x.x, dd); dd = fmaf(d.y, x.y, dd); dd = fmaf(d.z, x.z, dd); dd = fmaf(d.w, x.w, dd);
        }
#pragma unroll
        for (int s = 8; s >= 1; s >>= 1) { dot += __shfl_xor(dot, s, 16); dd += __shfl_xor(dd, s, 16); }
        const float sc = dot / sqrt_c;
        mx = fmaxf(mx, sc);
        if (t == 0) { prob[g][j] = sc; dsc[g][j] = dd; }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    float l = 0.f;
    for (int j = 0; j < n_agents; ++j) l += expf(prob[g][j] - mx);
    float dbar = 0.f;
    for (int j = 0; j < n_agents; ++j) dbar = fmaf(expf(prob[g][j] - mx) / l, dsc[g][j], dbar);
    __builtin_amdgcn_wave_barrier();
    if (t == 0) {
        for (int j = 0; j < n_agents; ++j) {
            const float pj = expf(prob[g][j] - mx) / l;
            const float ds = pj * (dsc[g][j] - dbar) / sqrt_c;
            prob[g][j] = pj;
            dsc[g][j] = ds;
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    for (int ch = 4 * t; ch < c; ch += 64) {
        const float4 q = *reinterpret_cast<const float4*>(ap.p[0] + base + ch);
        const float4 d = *reinterpret_cast<const float4*>(dout + base + ch);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = n_agents - 1; j >= 0; --j) {
            const float4 x = (j == 0) ? q : *reinterpret_cast<const float4*>(ap.p[j] + base + ch);
            const float pj = prob[g][j], ds = dsc[g][j];
            acc.x = fmaf(ds, x.x, acc.x); acc.y = fmaf(ds, x.y, acc.y); acc.z = fmaf(ds, x.z, acc.z); acc.w = fmaf(ds, x.w, acc.w);
            float4 r;
            r.x = fmaf(pj, d.x, ds * q.x); r.y = fmaf(pj, d.y, ds * q.y); r.z = fmaf(pj, d.z, ds * q.z); r.w = fmaf(pj, d.w, ds * q.w);
            if (j == 0) { r.x += acc.x; r.y += acc.y; r.z += acc.z; r.w += acc.w; }
            *reinterpret_cast<float4*>(dp.p[j] + base + ch) = r;
        }
    }
}

// ---------------------------------------------------------------------------------------------------- pillar feature net
constexpr int kPts = 32, kFeat = 10, kLdF = 12;
constexpr int kMom = kFeat + kFeat * kFeat;   // S (10) then F (10 x 10, row major)

// the augmented 10-vector of point `lane` of pillar `pil` (pillar.hip, airv2x_pillar_vfe.py:121-155), zero for padded rows
__device__ __forceinline__ void pillar_features(const float4* __restrict__ vox, const int4 c, int num, int npts_raw, int pil, int lane,
                                                float vx, float vy, float vz, float xoff, float yoff, float zoff, float* f) {
    float4 pt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < kPts) pt = vox[(size_t)pil * kPts + lane];
    float sx = pt.x, sy = pt.y, sz = pt.z;
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) {
        sx += __shfl_xor(sx, o);
        sy += __shfl_xor(sy, o);
        sz += __shfl_xor(sz, o);
    }
    const float fn = (float)npts_raw;
    const float mx = sx / fn, my = sy / fn, mz = sz / fn;
    const bool valid = lane < num;
    const float cx = __fadd_rn(__fmul_rn((float)c.w, vx), xoff);
    const float cy = __fadd_rn(__fmul_rn((float)c.z, vy), yoff);
    const float cz = __fadd_rn(__fmul_rn((float)c.y, vz), zoff);
    f[0] = pt.x; f[1] = pt.y; f[2] = pt.z; f[3] = pt.w;
    f[4] = pt.x - mx; f[5] = pt.y - my; f[6] = pt.z - mz;
    f[7] = pt.x - cx; f[8] = pt.y - cy; f[9] = pt.z - cz;
#pragma unroll
    for (int j = 0; j < kFeat; ++j) f[j] = (valid && lane < kPts) ? f[j] : 0.f;
}

__global__ __launch_bounds__(256) void pillar_moments_kernel(const float4* __restrict__ vox, const int4* __restrict__ coords,
                                                             const int* __restrict__ npts, int n_pillars, float vx, float vy, float vz,
                                                             float xoff, float yoff, float zoff, double* __restrict__ part) {
    __shared__ double red[4][kMom];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc[kMom];
#pragma unroll
    for (int k = 0; k < kMom; ++k) acc[k] = 0.0;
    for (int pil = blockIdx.x * 4 + wave; pil < n_pillars; pil += gridDim.x * 4) {
        const int4 c = coords[pil];
        const int raw = npts[pil];
        const int num = raw < 0 ? 0 : (raw > kPts ? kPts : raw);
        float f[kFeat];
        pillar_features(vox, c, num, raw, pil, lane, vx, vy, vz, xoff, yoff, zoff, f);
#pragma unroll
        for (int j = 0; j < kFeat; ++j) {
            acc[j] += (double)f[j];
#pragma unroll
            for (int k = 0; k < kFeat; ++k) acc[kFeat + j * kFeat + k] += (double)f[j] * (double)f[k];
        }
    }
#pragma unroll
    for (int k = 0; k < kMom; ++k) {
        double v = acc[k];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < kMom; k += 256) part[(size_t)blockIdx.x * kMom + k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
}

// out[k] = sum over the parts of part[b][k]: 8 columns per workgroup, 32 lanes walk the parts, lane sums added in lane order
__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ part, int nparts, int width, double* __restrict__ out) {
    __shared__ double red[32][8];
    const int j = threadIdx.x & 7, sl = threadIdx.x >> 3;
    const int k = blockIdx.x * 8 + j;
    double s = 0.0;
    if (k < width)
        for (int b = sl; b < nparts; b += 32) s += part[(size_t)b * width + k];
    red[sl][j] = s;
    __syncthreads();
    if (sl == 0 && k < width) {
        s = 0.0;
        for (int b = 0; b < 32; ++b) s += red[b][j];
        out[k] = s;
    }
}

constexpr int kBwdW = kFeat + 2;   // per channel: G[0..9], d beta, d gamma

// lane = output channel.  part: [workgroup][64][12] doubles.  ROWS: ``dcanvas`` is the (n_pillars, 64) gradient of the pillar
// features themselves (the stand-alone PillarVFE module) instead of the canvas it was scattered into.
template <bool ROWS>
__global__ __launch_bounds__(256) void pillar_backward_kernel(const float4* __restrict__ vox, const int4* __restrict__ coords,
                                                              const int* __restrict__ npts, int n_pillars,
                                                              const float* __restrict__ pfn_w, const float* __restrict__ bn_scale,
                                                              const float* __restrict__ bn_shift, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, float vx, float vy, float vz, float xoff,
                                                              float yoff, float zoff, const float* __restrict__ dcanvas, int agent0,
                                                              const int* __restrict__ slot_map, int n_agents, int ny, int nx,
                                                              double* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float feats[4][kPts][kLdF];
    __shared__ double red[4][64][kBwdW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float w[kFeat];
#pragma unroll
    for (int j = 0; j < kFeat; ++j) w[j] = pfn_w[lane * kFeat + j];
    const float sc = bn_scale[lane], sh = bn_shift[lane], mu = mean[lane], rs = rstd[lane];
    double acc[kBwdW];
#pragma unroll
    for (int k = 0; k < kBwdW; ++k) acc[k] = 0.0;
    for (int pil = blockIdx.x * 4 + wave; pil < n_pillars; pil += gridDim.x * 4) {
        const int4 c = coords[pil];
        const int raw = npts[pil];
        const int num = raw < 0 ? 0 : (raw > kPts ? kPts : raw);
        float f[kFeat];
        pillar_features(vox, c, num, raw, pil, lane, vx, vy, vz, xoff, yoff, zoff, f);
        if (lane < kPts) {
#pragma unroll
            for (int j = 0; j < kFeat; ++j) feats[wave][lane][j] = f[j];
            feats[wave][lane][10] = 0.f; feats[wave][lane][11] = 0.f;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // arg-max over the 32 rows exactly as the forward kernel walks them; a padded row is the all-zero feature vector
        float best = 0.f, best_lin = 0.f;
        int best_q = -1;                       // -1: no active row (ReLU output 0 everywhere -> no gradient)
        for (int q = 0; q < num; ++q) {
            const float4 a = *reinterpret_cast<const float4*>(&feats[wave][q][0]);
            const float4 b = *reinterpret_cast<const float4*>(&feats[wave][q][4]);
            const float4 d = *reinterpret_cast<const float4*>(&feats[wave][q][8]);
            float lin = a.x * w[0];
            lin = fmaf(a.y, w[1], lin); lin = fmaf(a.z, w[2], lin); lin = fmaf(a.w, w[3], lin);
            lin = fmaf(b.x, w[4], lin); lin = fmaf(b.y, w[5], lin); lin = fmaf(b.z, w[6], lin);
            lin = fmaf(b.w, w[7], lin); lin = fmaf(d.x, w[8], lin); lin = fmaf(d.y, w[9], lin);
            const float y = fmaf(lin, sc, sh);
            if (y > best) { best = y; best_lin = lin; best_q = q; }
        }
        if (num < kPts && sh > best) { best = sh; best_lin = 0.f; best_q = kPts; }   // a padded row wins: features are zero
        float g = 0.f;
        if (ROWS) {
            if (best_q >= 0) g = dcanvas[(size_t)pil * 64 + lane];
        } else if (best_q >= 0 && c.x >= 0 && c.x < n_agents && (unsigned)c.z < (unsigned)ny && (unsigned)c.w < (unsigned)nx) {
            const int agent = slot_map ? slot_map[c.x] : agent0 + c.x;
            const size_t pixel = ((size_t)agent * ny + c.z) * nx + c.w + c.y;
            g = dcanvas[pixel * 64 + lane];
        }
        if (g != 0.f) {
            if (best_q < kPts) {
#pragma unroll
                for (int k = 0; k < kFeat; ++k) acc[k] += (double)(g * feats[wave][best_q][k]);
            }
            acc[kFeat] += (double)g;
            acc[kFeat + 1] += (double)(g * ((best_lin - mu) * rs));
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int k = 0; k < kBwdW; ++k) red[wave][lane][k] = acc[k];
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * kBwdW; i += 256) {
        const int ch = i / kBwdW, k = i % kBwdW;
        part[(size_t)blockIdx.x * 64 * kBwdW + i] = red[0][ch][k] + red[1][ch][k] + red[2][ch][k] + red[3][ch][k];
    }
}

inline int pillar_blocks(int n_pillars) {
    int blocks = (n_pillars + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    return blocks;
}

// pillar_moments_kernel ends with a 110-value fp64 wave reduction per wave (1 320 cross-lane moves) and a workgroup's pillars are few: one
// round of workgroups (a CU each) instead of four amortises that epilogue (113 -> see profiles/r04l) and quarters the partials
inline int pillar_moment_blocks(int n_pillars) {
    int blocks = (n_pillars + 3) / 4;
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    return blocks;
}

inline int ew_blocks(size_t n) {
    size_t b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" uint64_t av2x_bn_workspace_bytes(int64_t rows, int32_t c) {
    if (rows <= 0 || c <= 0) return 0;
    const uint64_t slabs = ((uint64_t)rows + kSlabRows - 1) / kSlabRows;
    return slabs * 2ull * (uint64_t)c * sizeof(double);
}

extern "C" int av2x_bn_stats(const float* z, int64_t rows, int32_t c, void* workspace, float* mean, float* var,
                             av2x_stream_t stream) {
    if (!z || !workspace || !mean || !var) return av2x::fail("av2x_bn_stats: null argument");
    if (rows <= 0 || c <= 0) return av2x::fail("av2x_bn_stats: bad sizes");
    const int slabs = (int)((rows + kSlabRows - 1) / kSlabRows);
    hipStream_t st = av2x::as_stream(stream);
    double* part = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(moments_stage1<0>, dim3(slabs), dim3(256), 0, st, z, (const float*)nullptr, (size_t)rows, c, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, part);
    hipLaunchKernelGGL(moments_stage2<0>, dim3((c + 1) / 2), dim3(256), 0, st, part, slabs, c, (double)rows, mean, var);
    return av2x::check_launch("bn_stats kernels");
}

// d bias[c] = sum over rows of x[r][c] (conv_backward's bias gradient): the MODE 0 slab pass, first moment only
extern "C" uint64_t av2x_channel_sum_workspace_bytes(int64_t rows, int32_t c) { return av2x_bn_workspace_bytes(rows, c); }

extern "C" int av2x_channel_sum(const float* x, int64_t rows, int32_t c, void* workspace, float* out, av2x_stream_t stream) {
    if (!x || !workspace || !out || rows <= 0 || c <= 0) return av2x::fail("av2x_channel_sum: bad argument");
    const int slabs = (int)((rows + kSlabRows - 1) / kSlabRows);
    hipStream_t st = av2x::as_stream(stream);
    double* part = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(moments_stage1<0>, dim3(slabs), dim3(256), 0, st, x, (const float*)nullptr, (size_t)rows, c, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, part);
    hipLaunchKernelGGL(moments_stage2<2>, dim3((c + 1) / 2), dim3(256), 0, st, part, slabs, c, (double)rows, out, (float*)nullptr);
    return av2x::check_launch("channel_sum kernels");
}

extern "C" int av2x_bn_finalize(const float* mean, const float* var, const float* gamma, const float* beta, int32_t c, float eps,
                                int64_t count, float momentum, int32_t times, float* rstd, float* scale, float* shift,
                                float* running_mean, float* running_var, int64_t* num_batches_tracked, av2x_stream_t stream) {
    if (!mean || !var || !gamma || !beta || !rstd || !scale || !shift) return av2x::fail("av2x_bn_finalize: null argument");
    if (c <= 0 || count <= 0 || times < 0) return av2x::fail("av2x_bn_finalize: bad sizes");
    if ((running_mean == nullptr) != (running_var == nullptr)) return av2x::fail("av2x_bn_finalize: running_mean and running_var go together");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((c + 255) / 256), dim3(256), 0, av2x::as_stream(stream), mean, var, gamma, beta, c, eps,
                       (double)count, momentum, times, rstd, scale, shift, running_mean, running_var,
                       reinterpret_cast<long long*>(num_batches_tracked));
    return av2x::check_launch("bn_finalize_kernel");
}

extern "C" int av2x_affine_act(const float* z, int64_t rows, int32_t c, const float* scale, const float* shift, int32_t act,
                               float* y, av2x_stream_t stream) {
    if (!z || !y) return av2x::fail("av2x_affine_act: null argument");
    if (rows <= 0 || c <= 0 || c % 4) return av2x::fail("av2x_affine_act: bad sizes (c must be a multiple of 4)");
    if (act != 0 && act != 1) return av2x::fail("av2x_affine_act: act %d (0 identity, 1 ReLU)", act);
    const size_t n4 = (size_t)rows * c / 4;
    hipLaunchKernelGGL(affine_act_kernel, dim3(ew_blocks(n4)), dim3(256), 0, av2x::as_stream(stream),
                       reinterpret_cast<const float4*>(z), scale, shift, n4, c / 4, act, reinterpret_cast<float4*>(y));
    return av2x::check_launch("affine_act_kernel");
}

// stats + finalize + normalise in ONE call (three launches): what a train-mode BatchNorm forward is made of
extern "C" int av2x_bn_train_forward(const float* z, int64_t rows, int32_t c, const float* gamma, const float* beta, float eps,
                                     float momentum, int32_t times, int32_t act, void* workspace, float* stats5, float* y,
                                     float* running_mean, float* running_var, int64_t* num_batches_tracked, av2x_stream_t stream) {
    if (!stats5 || !z || !workspace || !gamma || !beta) return av2x::fail("av2x_bn_train_forward: null argument");
    if (rows <= 0 || c <= 0 || times < 0) return av2x::fail("av2x_bn_train_forward: bad sizes");
    if ((running_mean == nullptr) != (running_var == nullptr)) return av2x::fail("av2x_bn_train_forward: running_mean and running_var go together");
    float* scale = stats5 + 3 * (size_t)c, *shift = stats5 + 4 * (size_t)c;
    const int slabs = (int)((rows + kSlabRows - 1) / kSlabRows);
    hipStream_t st = av2x::as_stream(stream);
    double* part = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(moments_stage1<0>, dim3(slabs), dim3(256), 0, st, z, (const float*)nullptr, (size_t)rows, c, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, part);
    hipLaunchKernelGGL(moments_finalize_kernel, dim3((c + 1) / 2), dim3(256), 0, st, part, slabs, c, (double)rows, gamma, beta, eps, momentum, times,
                       stats5, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked));
    if (int e = av2x::check_launch("bn_train_forward statistics")) return e;
    return av2x_affine_act(z, rows, c, scale, shift, act, y, stream);
}

extern "C" int av2x_bn_backward(const float* dy, const float* z, int64_t rows, int32_t c, const float* mean, const float* rstd,
                                const float* scale, const float* shift, int32_t act, void* workspace, float* dgamma, float* dbeta,
                                float* dz, av2x_stream_t stream) {
    if (!dy || !z || !mean || !rstd || !scale || !shift || !workspace || !dgamma || !dbeta || !dz)
        return av2x::fail("av2x_bn_backward: null argument");
    if (rows <= 0 || c <= 0) return av2x::fail("av2x_bn_backward: bad sizes");
    if (act != 0 && act != 1) return av2x::fail("av2x_bn_backward: act %d (0 identity, 1 ReLU)", act);
    const int slabs = (int)((rows + kSlabRows - 1) / kSlabRows);
    hipStream_t st = av2x::as_stream(stream);
    double* part = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(moments_stage1<1>, dim3(slabs), dim3(256), 0, st, z, dy, (size_t)rows, c, mean, rstd, scale, shift, act, part);
    hipLaunchKernelGGL(moments_stage2<1>, dim3((c + 1) / 2), dim3(256), 0, st, part, slabs, c, (double)rows, dbeta, dgamma);
    const size_t total = (size_t)rows * c;
    hipLaunchKernelGGL(bn_backward_apply_kernel, dim3(ew_blocks(total)), dim3(256), 0, st, dy, z, mean, rstd, scale, shift, dbeta, dgamma,
                       total, c, (float)(1.0 / (double)rows), act, dz);
    return av2x::check_launch("bn_backward kernels");
}

extern "C" int av2x_pixel_attn_backward(const float* const* agents, int32_t n_agents, int32_t hw, int32_t c, const float* dout,
                                        float* const* dagents, av2x_stream_t stream) {
    if (!agents || !dagents || !dout) return av2x::fail("av2x_pixel_attn_backward: null argument");
    if (n_agents < 1 || n_agents > kMaxAgentsT) return av2x::fail("av2x_pixel_attn_backward: 1 <= n_agents <= %d", kMaxAgentsT);
    if (hw <= 0 || c <= 0 || c % 4) return av2x::fail("av2x_pixel_attn_backward: bad sizes (c must be a multiple of 4)");
    AgentPtrsT ap;
    AgentOutPtrsT dp;
    for (int j = 0; j < kMaxAgentsT; ++j) {
        ap.p[j] = j < n_agents ? agents[j] : nullptr;
        dp.p[j] = j < n_agents ? dagents[j] : nullptr;
        if (j < n_agents && (!ap.p[j] || !dp.p[j])) return av2x::fail("av2x_pixel_attn_backward: null agent map %d", j);
    }
    hipLaunchKernelGGL(pixel_attn_backward_kernel, dim3((hw + 15) / 16), dim3(256), 0, av2x::as_stream(stream), ap, dp, n_agents, hw, c,
                       sqrtf((float)c), dout);
    return av2x::check_launch("pixel_attn_backward_kernel");
}

extern "C" uint64_t av2x_pillar_train_workspace_bytes(int32_t n_pillars) {
    const uint64_t blocks = (uint64_t)pillar_blocks(n_pillars < 0 ? 0 : n_pillars);
    const uint64_t a = blocks * kMom * sizeof(double), b = blocks * 64 * kBwdW * sizeof(double);
    return a > b ? a : b;
}

extern "C" int av2x_pillar_moments(const float* voxel_features, const int32_t* voxel_coords, const int32_t* voxel_num_points,
                                   int32_t n_pillars, const float* geom, void* workspace, double* moments, av2x_stream_t stream) {
    if (!voxel_features || !voxel_coords || !voxel_num_points || !geom || !workspace || !moments)
        return av2x::fail("av2x_pillar_moments: null argument");
    if (n_pillars <= 0) return av2x::fail("av2x_pillar_moments: no pillars");
    const int blocks = pillar_moment_blocks(n_pillars);
    hipStream_t st = av2x::as_stream(stream);
    double* part = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(pillar_moments_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const float4*>(voxel_features),
                       reinterpret_cast<const int4*>(voxel_coords), voxel_num_points, n_pillars, geom[0], geom[1], geom[2], geom[3],
                       geom[4], geom[5], part);
    hipLaunchKernelGGL(sum_partials_kernel, dim3((kMom + 7) / 8), dim3(256), 0, st, part, blocks, kMom, moments);
    return av2x::check_launch("pillar_moments kernels");
}

extern "C" int av2x_pillar_vfe_backward(const float* voxel_features, const int32_t* voxel_coords, const int32_t* voxel_num_points,
                                        int32_t n_pillars, const float* pfn_w, const float* bn_scale, const float* bn_shift,
                                        const float* mean, const float* rstd, const float* geom, const float* dcanvas,
                                        int32_t canvas_agent0, const int32_t* slot_map, int32_t n_agents_type, int32_t ny, int32_t nx,
                                        void* workspace, double* out, av2x_stream_t stream) {
    if (!voxel_features || !voxel_coords || !voxel_num_points || !pfn_w || !bn_scale || !bn_shift || !mean || !rstd || !geom ||
        !dcanvas || !workspace || !out)
        return av2x::fail("av2x_pillar_vfe_backward: null argument");
    if (n_pillars <= 0 || ny <= 0 || nx <= 0) return av2x::fail("av2x_pillar_vfe_backward: bad sizes");
    const int blocks = pillar_blocks(n_pillars);
    hipStream_t st = av2x::as_stream(stream);
    double* part = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(pillar_backward_kernel<false>, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const float4*>(voxel_features),
                       reinterpret_cast<const int4*>(voxel_coords), voxel_num_points, n_pillars, pfn_w, bn_scale, bn_shift, mean, rstd,
                       geom[0], geom[1], geom[2], geom[3], geom[4], geom[5], dcanvas, canvas_agent0, slot_map, n_agents_type, ny, nx, part);
    hipLaunchKernelGGL(sum_partials_kernel, dim3((64 * kBwdW + 7) / 8), dim3(256), 0, st, part, blocks, 64 * kBwdW, out);
    return av2x::check_launch("pillar_vfe_backward kernels");
}

extern "C" int av2x_pillar_vfe_backward_rows(const float* voxel_features, const int32_t* voxel_coords, const int32_t* voxel_num_points,
                                             int32_t n_pillars, const float* pfn_w, const float* bn_scale, const float* bn_shift,
                                             const float* mean, const float* rstd, const float* geom, const float* dfeatures,
                                             void* workspace, double* out, av2x_stream_t stream) {
    if (!voxel_features || !voxel_coords || !voxel_num_points || !pfn_w || !bn_scale || !bn_shift || !mean || !rstd || !geom ||
        !dfeatures || !workspace || !out)
        return av2x::fail("av2x_pillar_vfe_backward_rows: null argument");
    if (n_pillars <= 0) return av2x::fail("av2x_pillar_vfe_backward_rows: bad sizes");
    const int blocks = pillar_blocks(n_pillars);
    hipStream_t st = av2x::as_stream(stream);
    double* part = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(pillar_backward_kernel<true>, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const float4*>(voxel_features),
                       reinterpret_cast<const int4*>(voxel_coords), voxel_num_points, n_pillars, pfn_w, bn_scale, bn_shift, mean, rstd,
                       geom[0], geom[1], geom[2], geom[3], geom[4], geom[5], dfeatures, 0, (const int*)nullptr, 0, 0, 0, part);
    hipLaunchKernelGGL(sum_partials_kernel, dim3((64 * kBwdW + 7) / 8), dim3(256), 0, st, part, blocks, 64 * kBwdW, out);
    return av2x::check_launch("pillar_vfe_backward_rows kernels");
}

// PointPillarScatter backward (point_pillar_scatter.py:59-68): d pillar_features[pil, :] = d canvas[agent, y, x, :]
__global__ void pillar_gather_kernel(const float4* __restrict__ dcanvas, const int4* __restrict__ coords, int n_pillars, int c4,
                                     float4* __restrict__ dfeat, int n_agents, int ny, int nx) {
    const size_t total = (size_t)n_pillars * c4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int pil = (int)(i / c4), q = (int)(i % c4);
        const int4 c = coords[pil];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c.x >= 0 && c.x < n_agents && (unsigned)c.z < (unsigned)ny && (unsigned)c.w < (unsigned)nx)
            v = dcanvas[(((size_t)c.x * ny + c.z) * nx + c.w + c.y) * c4 + q];
        dfeat[i] = v;
    }
}

extern "C" int av2x_pillar_gather(const float* dcanvas, const int32_t* voxel_coords, int32_t n_pillars, int32_t channels,
                                  float* dfeatures, int32_t n_agents, int32_t ny, int32_t nx, av2x_stream_t stream) {
    if (n_pillars == 0) return 0;
    if (!dcanvas || !voxel_coords || !dfeatures) return av2x::fail("av2x_pillar_gather: null argument");
    if (n_pillars < 0 || channels <= 0 || channels % 4 || n_agents <= 0 || ny <= 0 || nx <= 0)
        return av2x::fail("av2x_pillar_gather: bad sizes (channels must be a multiple of 4)");
    const size_t total = (size_t)n_pillars * (channels / 4);
    hipLaunchKernelGGL(pillar_gather_kernel, dim3(ew_blocks(total)), dim3(256), 0, av2x::as_stream(stream),
                       reinterpret_cast<const float4*>(dcanvas), reinterpret_cast<const int4*>(voxel_coords), n_pillars, channels / 4,
                       reinterpret_cast<float4*>(dfeatures), n_agents, ny, nx);
    return av2x::check_launch("pillar_gather_kernel");
}
