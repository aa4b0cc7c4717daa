// OPV2V-style Where2comm (models/where2comm_modules/where2comm_attn.py + where2comm.py), all HBM-bound:
//   warp_fuse_kernel            warp_affine_simple of every agent into the ego frame FUSED with the per-pixel fusion
//                               operator (AttenFusion :55-67 / MaxFusion :70-75), so the warped maps never exist in HBM:
//                               N x (<= 4 taps) reads + ONE write per pixel, instead of N reads + N writes (warp) followed by
//                               N reads + one write (fusion)
//   agent_max_kernel            MaxFusion on maps that are already aligned
//   count_nonzero_where_kernel  `(batch_x[b] * communication_mask).count_nonzero()` of where2comm.py:93-95 without
//                               materialising the product
#include "av2x_common.hpp"

namespace {

constexpr int kMaxAgents = 32;

struct AgentPtrs {
    const float* p[kMaxAgents];
};

struct AgentThetas {
    float t[kMaxAgents][6];
};

__device__ __forceinline__ float lin_m1_1(int i, int n) {   // torch.linspace(-1, 1, n)[i]: ATen fills the two halves from either end
    if (n <= 1) return -1.0f;
    const float step = 2.0f / (float)(n - 1);
    return (i < n / 2) ? (-1.0f + step * (float)i) : (1.0f - step * (float)(n - 1 - i));
}

// The tap arithmetic is warp_affine_kernel<.., false>'s (v2xvit.hip; pinned by the When2com goldens), the attention
// pixel_attn_kernel's (where2comm.hip).  The ego (agent 0) goes through the sampler with its own matrix like every
// other agent (where2comm_attn.py:364-366 warps all N maps with t_matrix[0, :N]), and the warped ego is the query.
// C = 64 * CK; 16 lanes own a pixel, lane t holds channels {64k + 4t .. +3}.
template <int CK, int MODE>   // MODE 0: ATTEN, 1: MAX
__global__ __launch_bounds__(256) void warp_fuse_kernel(const AgentPtrs ap, const AgentThetas th, int n_agents, int H, int W,
                                                        float sqrt_c, float* __restrict__ out) {
    constexpr int C = 64 * CK;
    const int t = threadIdx.x & 15;
    const int pix = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (pix >= H * W) return;
    const int i = pix / W, jx = pix - i * W;
    const float xn = (lin_m1_1(jx, W) * (float)(W - 1)) / (float)W;   // F.affine_grid, align_corners=False
    const float yn = (lin_m1_1(i, H) * (float)(H - 1)) / (float)H;
    float4 q[CK], o[CK];
#pragma unroll
    for (int k = 0; k < CK; ++k) o[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    float mrun = -INFINITY, lrun = 0.f;
    for (int j = 0; j < n_agents; ++j) {
        const float gx = th.t[j][0] * xn + th.t[j][1] * yn + th.t[j][2];
        const float gy = th.t[j][3] * xn + th.t[j][4] * yn + th.t[j][5];
        const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f;        // grid_sampler_unnormalize, align_corners=False
        const float iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
        const float x0f = floorf(ix), y0f = floorf(iy);
        const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
        const float wx1 = ix - x0f, wx0 = (x0f + 1.f) - ix, wy1 = iy - y0f, wy0 = (y0f + 1.f) - iy;
        const float w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;   // nw, ne, sw, se
        const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)x1 < (unsigned)W;
        const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)y1 < (unsigned)H;
        const float* base = ap.p[j] + 4 * t;
        float4 x[CK];
#pragma unroll
        for (int k = 0; k < CK; ++k) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            auto add = [&](bool ok, int yy, int xx, float wgt) {
                if (ok) {
                    const float4 v = *reinterpret_cast<const float4*>(base + ((size_t)yy * W + xx) * C + 64 * k);
                    acc.x += v.x * wgt; acc.y += v.y * wgt; acc.z += v.z * wgt; acc.w += v.w * wgt;
                }
            };
            add(vy0 && vx0, y0, x0, w00);
            add(vy0 && vx1, y0, x1, w01);
            add(vy1 && vx0, y1, x0, w10);
            add(vy1 && vx1, y1, x1, w11);
            x[k] = acc;
        }
        if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < CK; ++k) {
                if (j == 0) {
                    o[k] = x[k];
                } else {
                    o[k].x = fmaxf(o[k].x, x[k].x); o[k].y = fmaxf(o[k].y, x[k].y);
                    o[k].z = fmaxf(o[k].z, x[k].z); o[k].w = fmaxf(o[k].w, x[k].w);
                }
            }
            continue;
        }
        if (j == 0) {
#pragma unroll
            for (int k = 0; k < CK; ++k) q[k] = x[k];
        }
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < CK; ++k) {
            dot = fmaf(q[k].x, x[k].x, dot);
            dot = fmaf(q[k].y, x[k].y, dot);
            dot = fmaf(q[k].z, x[k].z, dot);
            dot = fmaf(q[k].w, x[k].w, dot);
        }
#pragma unroll
        for (int s = 8; s >= 1; s >>= 1) dot += __shfl_xor(dot, s, 16);
        const float sc = dot / sqrt_c;             // score / np.sqrt(dim) (where2comm_attn.py:49)
        const float mnew = fmaxf(mrun, sc);
        const float alpha = expf(mrun - mnew);     // exp(-inf) = 0 on the first agent
        const float pj = expf(sc - mnew);
        lrun = lrun * alpha + pj;
#pragma unroll
        for (int k = 0; k < CK; ++k) {
            o[k].x = fmaf(pj, x[k].x, o[k].x * alpha);
            o[k].y = fmaf(pj, x[k].y, o[k].y * alpha);
            o[k].z = fmaf(pj, x[k].z, o[k].z * alpha);
            o[k].w = fmaf(pj, x[k].w, o[k].w * alpha);
        }
        mrun = mnew;
    }
    const float inv = (MODE == 1) ? 1.0f : 1.0f / lrun;
    float* dst = out + (size_t)pix * C + 4 * t;
#pragma unroll
    for (int k = 0; k < CK; ++k) {
        float4 r = o[k];
        if (MODE == 0) { r.x *= inv; r.y *= inv; r.z *= inv; r.w *= inv; }
        *reinterpret_cast<float4*>(dst + 64 * k) = r;
    }
}

__global__ void agent_max_kernel(const AgentPtrs ap, int n_agents, size_t n4, float4* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 m = reinterpret_cast<const float4*>(ap.p[0])[i];
        for (int j = 1; j < n_agents; ++j) {
            const float4 v = reinterpret_cast<const float4*>(ap.p[j])[i];
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
        out[i] = m;
    }
}

__global__ __launch_bounds__(256) void count_nonzero_where_kernel(const float4* __restrict__ x, const float* __restrict__ gate,
                                                                  float thr, size_t n4, int c4,
                                                                  unsigned long long* __restrict__ result) {
    unsigned int local = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        if (gate[i / c4] > thr) {
            const float4 v = x[i];
            local += (v.x != 0.f) + (v.y != 0.f) + (v.z != 0.f) + (v.w != 0.f);
        }
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) local += __shfl_xor(local, s, 64);
    __shared__ unsigned int wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long tot = (unsigned long long)wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (tot) atomicAdd(result, tot);   // integer: the order of the adds does not matter
    }
}

}  // namespace

extern "C" int av2x_warp_fuse(const float* const* agents, const float* theta_host, int32_t n_agents, int32_t h, int32_t w,
                              int32_t c, int32_t mode, float* out, av2x_stream_t stream) {
    if (!agents || !theta_host || !out) return av2x::fail("av2x_warp_fuse: null argument");
    if (n_agents < 1 || n_agents > kMaxAgents) return av2x::fail("av2x_warp_fuse: n_agents=%d outside [1,%d]", n_agents, kMaxAgents);
    if (mode != 0 && mode != 1) return av2x::fail("av2x_warp_fuse: mode=%d (0 = ATTEN, 1 = MAX)", mode);
    if (h <= 0 || w <= 0) return av2x::fail("av2x_warp_fuse: bad sizes");
    AgentPtrs ap;
    AgentThetas th;
    for (int i = 0; i < kMaxAgents; ++i) {
        if (i < n_agents && !agents[i]) return av2x::fail("av2x_warp_fuse: agent %d is null", i);
        ap.p[i] = i < n_agents ? agents[i] : nullptr;
        for (int k = 0; k < 6; ++k) th.t[i][k] = i < n_agents ? theta_host[i * 6 + k] : 0.f;
    }
    const float sq = (float)sqrt((double)c);
    const dim3 grid((h * w + 15) / 16), block(256);
    hipStream_t st = av2x::as_stream(stream);
#define AV2X_WF(CK)                                                                                                      \
    if (mode == 0) hipLaunchKernelGGL((warp_fuse_kernel<CK, 0>), grid, block, 0, st, ap, th, n_agents, h, w, sq, out);   \
    else hipLaunchKernelGGL((warp_fuse_kernel<CK, 1>), grid, block, 0, st, ap, th, n_agents, h, w, sq, out)
    switch (c) {
        case 64: AV2X_WF(1); break;
        case 128: AV2X_WF(2); break;
        case 256: AV2X_WF(4); break;
        default: return av2x::fail("av2x_warp_fuse: c=%d unsupported (64/128/256)", c);
    }
#undef AV2X_WF
    return av2x::check_launch("warp_fuse_kernel");
}

extern "C" int av2x_agent_max(const float* const* agents, int32_t n_agents, uint64_t elems_per_agent, float* out,
                              av2x_stream_t stream) {
    if (!agents || !out) return av2x::fail("av2x_agent_max: null argument");
    if (n_agents < 1 || n_agents > kMaxAgents) return av2x::fail("av2x_agent_max: n_agents=%d outside [1,%d]", n_agents, kMaxAgents);
    if (elems_per_agent % 4) return av2x::fail("av2x_agent_max: elems_per_agent must be a multiple of 4");
    if (elems_per_agent == 0) return 0;
    AgentPtrs ap;
    for (int i = 0; i < kMaxAgents; ++i) ap.p[i] = i < n_agents ? agents[i] : nullptr;
    const size_t n4 = elems_per_agent / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(agent_max_kernel, dim3((unsigned)blocks), dim3(256), 0, av2x::as_stream(stream), ap, n_agents, n4,
                       reinterpret_cast<float4*>(out));
    return av2x::check_launch("agent_max_kernel");
}

extern "C" int av2x_count_nonzero_where(const float* x, const float* gate, float thr, uint64_t n_pixels, int32_t c,
                                        unsigned long long* result, av2x_stream_t stream) {
    if (n_pixels == 0) return 0;
    if (!x || !gate || !result) return av2x::fail("av2x_count_nonzero_where: null argument");
    if (c <= 0 || c % 4) return av2x::fail("av2x_count_nonzero_where: c=%d must be a positive multiple of 4", c);
    const size_t n4 = (size_t)n_pixels * (c / 4);
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(count_nonzero_where_kernel, dim3((unsigned)blocks), dim3(256), 0, av2x::as_stream(stream),
                       reinterpret_cast<const float4*>(x), gate, thr, n4, c / 4, result);
    return av2x::check_launch("count_nonzero_where_kernel");
}
