// Where2Comm fusion kernels (all HBM-bound):
//   comm_conf_kernel   sigmoid(max_c psm)                       1.97 MB read / agent
//   comm_mask_kernel   k x k smoothing conv + bias, > threshold, ego override, exact popcount
//   apply_mask_kernel  x[n,h,w,:] *= mask[n,h,w]
//   pixel_attn_kernel  per pixel softmax(x0 . xj / sqrt(C)) weighted sum over the agents
//                      (ego row only), online softmax, 16 lanes per pixel, 16-byte loads
#include "av2x_common.hpp"
#include "block_scan.hpp"

namespace {

constexpr int kMaxAgents = 32;

__global__ void comm_conf_kernel(const float* __restrict__ psm, int npix, int ctot, int c, float* __restrict__ conf) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const float* row = psm + (size_t)p * ctot;
    float m = row[0];
    for (int j = 1; j < c; ++j) m = fmaxf(m, row[j]);
    // sigmoid is monotonic: max_c sigmoid(x_c) == sigmoid(max_c x_c) bit for bit
    conf[p] = 1.0f / (1.0f + expf(-m));
}

// one thread per output pixel; the conf map (140 KB / agent) is L2 resident
__global__ void comm_mask_kernel(const float* __restrict__ conf, int n, int h, int w, const float* __restrict__ gw,
                                 const float* __restrict__ gb, int k, float threshold,
                                 const int* __restrict__ sample_of_agent, const int* __restrict__ is_ego,
                                 float* __restrict__ smooth, float* __restrict__ mask, int* __restrict__ count) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;      // blockDim = (64, 4): one wave per row, four rows per workgroup
    const int a = blockIdx.z;
    int one = 0;
    if (x < w && y < h) {
        const int r = (k - 1) / 2;
        const float* img = conf + (size_t)a * h * w;
        float acc = 0.f;
        for (int dy = 0; dy < k; ++dy) {
            const int yy = y + dy - r;
            if ((unsigned)yy >= (unsigned)h) continue;
            for (int dx = 0; dx < k; ++dx) {
                const int xx = x + dx - r;
                if ((unsigned)xx >= (unsigned)w) continue;
                acc = fmaf(img[yy * w + xx], gw[dy * k + dx], acc);
            }
        }
        acc += gb[0];
        const size_t o = ((size_t)a * h + y) * w + x;
        smooth[o] = acc;
        one = (threshold > 0.f) ? (acc > threshold) : 1;
        mask[o] = (one || is_ego[a]) ? 1.f : 0.f;
    }
    // exact integer count of transmitted cells BEFORE the ego override (where2comm_fuse.py:137); ONE atomic per workgroup:
    // same-address atomics serialise in L2 (~10 ns each), 2 400 of them used to be most of this kernel's 33 us
    __shared__ int wsum[4];
    const unsigned long long bal = __ballot(one);
    if (threadIdx.x == 0) wsum[threadIdx.y] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (tot) atomicAdd(&count[sample_of_agent[a]], tot);
    }
}

__global__ void apply_mask_kernel(float4* __restrict__ x, const float* __restrict__ mask, size_t n4, int c4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float m = mask[i / c4];
        float4 v = x[i];
        v.x *= m; v.y *= m; v.z *= m; v.w *= m;
        x[i] = v;
    }
}

struct AgentPtrs {
    const float* p[kMaxAgents];
};

// C = 64 * CK channels; a pixel is owned by 16 lanes, lane t holds channels {64*k + 4*t .. +3}
template <int CK>
__global__ __launch_bounds__(256) void pixel_attn_kernel(const AgentPtrs ap, int n_agents, int hw, float sqrt_c,
                                                         float* __restrict__ out) {
    const int t = threadIdx.x & 15;
    const int pix = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (pix >= hw) return;  // whole 16-lane groups exit together
    const size_t base = (size_t)pix * (64 * CK) + 4 * t;
    float4 q[CK], o[CK];
#pragma unroll
    for (int k = 0; k < CK; ++k) {
        q[k] = *reinterpret_cast<const float4*>(ap.p[0] + base + 64 * k);
        o[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float mrun = -INFINITY, lrun = 0.f;
    for (int j = 0; j < n_agents; ++j) {
        float4 x[CK];
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < CK; ++k) {
            x[k] = (j == 0) ? q[k] : *reinterpret_cast<const float4*>(ap.p[j] + base + 64 * k);
            dot = fmaf(q[k].x, x[k].x, dot);
            dot = fmaf(q[k].y, x[k].y, dot);
            dot = fmaf(q[k].z, x[k].z, dot);
            dot = fmaf(q[k].w, x[k].w, dot);
        }
#pragma unroll
        for (int s = 8; s >= 1; s >>= 1) dot += __shfl_xor(dot, s, 16);
        const float sc = dot / sqrt_c;  // score / np.sqrt(dim) (where2comm_fuse.py:42)
        const float mnew = fmaxf(mrun, sc);
        const float alpha = expf(mrun - mnew);  // exp(-inf) = 0 on the first agent
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
    const float inv = 1.0f / lrun;
#pragma unroll
    for (int k = 0; k < CK; ++k) {
        float4 r = o[k];
        r.x *= inv; r.y *= inv; r.z *= inv; r.w *= inv;
        *reinterpret_cast<float4*>(out + base + 64 * k) = r;
    }
}

// Any other channel count (multiple of 4): scores first, then the weighted sum (second read of the agents' pixels
// comes from L2).  Only the reduced test / sub-module configurations take this path.
__global__ __launch_bounds__(256) void pixel_attn_generic_kernel(const AgentPtrs ap, int n_agents, int hw, int c, float sqrt_c,
                                                                 float* __restrict__ out) {
    const int t = threadIdx.x & 15;
    const int pix = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (pix >= hw) return;
    const size_t base = (size_t)pix * c;
    float wgt[kMaxAgents];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < kMaxAgents; ++j) {
        wgt[j] = 0.f;
        if (j < n_agents) {
            float dot = 0.f;
            for (int ch = 4 * t; ch < c; ch += 64) {
                const float4 q = *reinterpret_cast<const float4*>(ap.p[0] + base + ch);
                const float4 x = *reinterpret_cast<const float4*>(ap.p[j] + base + ch);
                dot = fmaf(q.x, x.x, dot); dot = fmaf(q.y, x.y, dot); dot = fmaf(q.z, x.z, dot); dot = fmaf(q.w, x.w, dot);
            }
#pragma unroll
            for (int s = 8; s >= 1; s >>= 1) dot += __shfl_xor(dot, s, 16);
            wgt[j] = dot / sqrt_c;
            mx = fmaxf(mx, wgt[j]);
        }
    }
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxAgents; ++j)
        if (j < n_agents) { wgt[j] = expf(wgt[j] - mx); l += wgt[j]; }
    const float inv = 1.0f / l;
    for (int ch = 4 * t; ch < c; ch += 64) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < kMaxAgents; ++j)
            if (j < n_agents) {
                const float4 x = *reinterpret_cast<const float4*>(ap.p[j] + base + ch);
                o.x = fmaf(wgt[j], x.x, o.x); o.y = fmaf(wgt[j], x.y, o.y); o.z = fmaf(wgt[j], x.z, o.z); o.w = fmaf(wgt[j], x.w, o.w);
            }
        o.x *= inv; o.y *= inv; o.z *= inv; o.w *= inv;
        *reinterpret_cast<float4*>(out + base + ch) = o;
    }
}

}  // namespace

extern "C" int av2x_comm_mask(const float* psm, int32_t n, int32_t h, int32_t w, int32_t ctot, int32_t c,
                              const float* gauss_w, const float* gauss_b, int32_t k, float threshold,
                              const int32_t* sample_of_agent, const int32_t* is_ego, float* conf, float* smooth,
                              float* mask, int32_t* count, av2x_stream_t stream) {
    if (n == 0) return 0;
    if (!psm || !gauss_w || !gauss_b || !sample_of_agent || !is_ego || !conf || !smooth || !mask || !count)
        return av2x::fail("av2x_comm_mask: null argument");
    if (n < 0 || h <= 0 || w <= 0 || c <= 0 || c > ctot || k <= 0 || (k & 1) == 0)
        return av2x::fail("av2x_comm_mask: bad sizes (n=%d h=%d w=%d c=%d ctot=%d k=%d)", n, h, w, c, ctot, k);
    hipStream_t st = av2x::as_stream(stream);
    const int npix = n * h * w;
    hipLaunchKernelGGL(comm_conf_kernel, dim3((npix + 255) / 256), dim3(256), 0, st, psm, npix, ctot, c, conf);
    if (int e = av2x::check_launch("comm_conf_kernel")) return e;
    hipLaunchKernelGGL(comm_mask_kernel, dim3((w + 63) / 64, (h + 3) / 4, n), dim3(64, 4), 0, st, conf, n, h, w, gauss_w, gauss_b, k,
                       threshold, sample_of_agent, is_ego, smooth, mask, count);
    return av2x::check_launch("comm_mask_kernel");
}

namespace {
// Training proxy objective of Communication.forward (where2comm_fuse.py:104-121): every agent transmits the K cells with
// the largest smoothed confidence (K drawn by the caller: int(H * W * random.uniform(0, 1)), :106).  One workgroup per
// agent: the K-th largest value by bisection over an order-preserving key of the fp32 bits (exact, 32 passes over the
// L2-resident map), then mask = value above it, plus the lowest-indexed cells equal to it until K are set (torch.topk's
// choice among exactly tied values is unspecified; tie-free maps give the same set).  count[sample] += cells set, before
// the ego override (:137-143).
__device__ __forceinline__ unsigned order_key(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(1024) void comm_topk_kernel(const float* __restrict__ smooth, int hw, const int* __restrict__ k_of_agent,
                                                         const int* __restrict__ sample_of, const int* __restrict__ is_ego,
                                                         float* __restrict__ mask, int* __restrict__ count) {
    const int a = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* v = smooth + (size_t)a * hw;
    float* m = mask + (size_t)a * hw;
    int K = k_of_agent[a];
    K = K < 0 ? 0 : (K > hw ? hw : K);
    __shared__ int red[16];
    __shared__ int tot;
    // the lane's cells as order keys in registers when the map fits (36 x 1024 cells: the 100 x 352 map does): the 33 bisection passes
    // compare registers instead of re-reading the map from L2 (171 -> ~40 us per launch)
    constexpr int KPT = 36;
    const bool cached = hw <= KPT * 1024;
    unsigned keys[KPT];
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
        const int i = tid + u * 1024;
        keys[u] = (cached && i < hw) ? order_key(v[i]) : 0u;
    }
    const int mine = cached ? (hw > tid ? (hw - tid + 1023) / 1024 : 0) : 0;      // cells of this lane
    auto count_ge = [&](unsigned key) {
        int c = 0;
        if (cached) {
#pragma unroll
            for (int u = 0; u < KPT; ++u) c += (u < mine && keys[u] >= key) ? 1 : 0;
        } else {
            for (int i = tid; i < hw; i += 1024) c += order_key(v[i]) >= key;
        }
        for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o);
        __syncthreads();
        if (lane == 0) red[wave] = c;
        __syncthreads();
        int t = 0;
        for (int k = 0; k < 16; ++k) t += red[k];
        return t;
    };
    unsigned cut = 0xffffffffu;
    int c_gt = 0;
    if (K > 0 && K < hw) {
        unsigned lo = 0u, hi = 0xffffffffu;     // |{key >= lo}| = hw >= K ; |{key >= hi}| < K unless K values are +NaN-max (not for finite maps)
        while (hi - lo > 1u) {
            const unsigned mid = lo + ((hi - lo) >> 1);
            if (count_ge(mid) >= K) lo = mid; else hi = mid;
        }
        cut = lo;
        c_gt = cut == 0xffffffffu ? 0 : count_ge(cut + 1u);
    }
    __syncthreads();
    if (K == 0 || K == hw) {
        for (int i = tid; i < hw; i += 1024) m[i] = K ? 1.f : 0.f;
    } else {
        const int need = K - c_gt;     // cells equal to the cut that still get a one, lowest index first
        av2x::block_scan<16>(
            hw, [&](int i) { return order_key(v[i]) == cut ? 1 : 0; },
            [&](int i, int ex) {
                const unsigned key = order_key(v[i]);
                m[i] = (key > cut || (key == cut && ex < need)) ? 1.f : 0.f;
            },
            &tot);
    }
    __syncthreads();
    if (tid == 0) atomicAdd(count + sample_of[a], K);
    if (is_ego[a]) {
        __syncthreads();
        for (int i = tid; i < hw; i += 1024) m[i] = 1.f;
    }
}

}  // namespace

extern "C" int av2x_comm_mask_topk(const float* smooth, int32_t n, int32_t hw, const int32_t* k_of_agent,
                                   const int32_t* sample_of_agent, const int32_t* is_ego, float* mask, int32_t* count,
                                   av2x_stream_t stream) {
    if (n == 0) return 0;
    if (!smooth || !k_of_agent || !sample_of_agent || !is_ego || !mask || !count) return av2x::fail("av2x_comm_mask_topk: null argument");
    if (n < 0 || hw <= 0) return av2x::fail("av2x_comm_mask_topk: bad sizes");
    hipLaunchKernelGGL(comm_topk_kernel, dim3(n), dim3(1024), 0, av2x::as_stream(stream), smooth, hw, k_of_agent, sample_of_agent,
                       is_ego, mask, count);
    return av2x::check_launch("comm_topk_kernel");
}

namespace {
// com = mean over samples of count[b] / (agents[b] * hw)   (where2comm_fuse.py:137, :147), fp32 like the reference
__global__ void comm_rate_kernel(const int* __restrict__ count, const float* __restrict__ agents, int B, float hw,
                                 float* __restrict__ com) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += (float)count[b] / (agents[b] * hw);
        com[0] = s / (float)B;
    }
}

}  // namespace

extern "C" int av2x_comm_rate(const int32_t* count, const float* agents_per_sample, int32_t n_samples, int32_t hw,
                              float* com, av2x_stream_t stream) {
    if (!count || !agents_per_sample || !com || n_samples <= 0) return av2x::fail("av2x_comm_rate: bad argument");
    hipLaunchKernelGGL(comm_rate_kernel, dim3(1), dim3(64), 0, av2x::as_stream(stream), count, agents_per_sample, n_samples,
                       (float)hw, com);
    return av2x::check_launch("comm_rate_kernel");
}

extern "C" int av2x_apply_mask(float* x, const float* mask, int32_t n, int32_t hw, int32_t c, av2x_stream_t stream) {
    if (n == 0) return 0;
    if (!x || !mask) return av2x::fail("av2x_apply_mask: null argument");
    if (c % 4) return av2x::fail("av2x_apply_mask: c=%d must be a multiple of 4", c);
    const size_t n4 = (size_t)n * hw * (c / 4);
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(apply_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, av2x::as_stream(stream),
                       reinterpret_cast<float4*>(x), mask, n4, c / 4);
    return av2x::check_launch("apply_mask_kernel");
}

namespace {
// F.interpolate(mask, size=(ho, wo), mode="bilinear", align_corners=False) of single-channel maps (where2comm_fuse.py:229-235: the
// communication mask when the first block's map and the confidence map differ in size).  ATen's arithmetic: scale = in / out,
// src = scale (dst + 0.5) - 0.5 clamped at 0, the two weights of each axis, rows combined after columns.
__global__ void mask_resize_kernel(const float* __restrict__ in, int n, int hi, int wi, int ho, int wo, float sh, float sw,
                                   float* __restrict__ out) {
    const size_t total = (size_t)n * ho * wo;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(e % wo), y = (int)((e / wo) % ho), img = (int)(e / ((size_t)wo * ho));
        float fy = sh * ((float)y + 0.5f) - 0.5f, fx = sw * ((float)x + 0.5f) - 0.5f;
        fy = fy < 0.f ? 0.f : fy;
        fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < hi - 1 ? 1 : 0), x1 = x0 + (x0 < wi - 1 ? 1 : 0);
        const float ly1 = fy - (float)y0, lx1 = fx - (float)x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const float* p = in + (size_t)img * hi * wi;
        out[e] = ly0 * (lx0 * p[(size_t)y0 * wi + x0] + lx1 * p[(size_t)y0 * wi + x1])
               + ly1 * (lx0 * p[(size_t)y1 * wi + x0] + lx1 * p[(size_t)y1 * wi + x1]);
    }
}
}  // namespace

extern "C" int av2x_mask_resize_bilinear(const float* in, int32_t n, int32_t hi, int32_t wi, int32_t ho, int32_t wo, float* out,
                                         av2x_stream_t stream) {
    if (n == 0) return 0;
    if (!in || !out) return av2x::fail("av2x_mask_resize_bilinear: null argument");
    if (hi <= 0 || wi <= 0 || ho <= 0 || wo <= 0) return av2x::fail("av2x_mask_resize_bilinear: empty map");
    const size_t total = (size_t)n * ho * wo;
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mask_resize_kernel, dim3((unsigned)blocks), dim3(256), 0, av2x::as_stream(stream), in, n, hi, wi, ho, wo,
                       (float)hi / (float)ho, (float)wi / (float)wo, out);
    return av2x::check_launch("mask_resize_kernel");
}

extern "C" int av2x_pixel_attn_fuse(const float* const* agents, int32_t n_agents, int32_t hw, int32_t c, float* out,
                                    av2x_stream_t stream) {
    if (!agents || !out) return av2x::fail("av2x_pixel_attn_fuse: null argument");
    if (n_agents < 1 || n_agents > kMaxAgents) return av2x::fail("av2x_pixel_attn_fuse: n_agents=%d outside [1,%d]", n_agents, kMaxAgents);
    if (hw <= 0) return 0;
    AgentPtrs ap;
    for (int i = 0; i < kMaxAgents; ++i) ap.p[i] = i < n_agents ? agents[i] : nullptr;
    const float inv = (float)sqrt((double)c);
    const dim3 grid((hw + 15) / 16), block(256);
    hipStream_t st = av2x::as_stream(stream);
    switch (c) {
        case 64: hipLaunchKernelGGL(pixel_attn_kernel<1>, grid, block, 0, st, ap, n_agents, hw, inv, out); break;
        case 128: hipLaunchKernelGGL(pixel_attn_kernel<2>, grid, block, 0, st, ap, n_agents, hw, inv, out); break;
        case 256: hipLaunchKernelGGL(pixel_attn_kernel<4>, grid, block, 0, st, ap, n_agents, hw, inv, out); break;
        default:
            if (c <= 0 || c % 4) return av2x::fail("av2x_pixel_attn_fuse: c=%d must be a positive multiple of 4", c);
            hipLaunchKernelGGL(pixel_attn_generic_kernel, grid, block, 0, st, ap, n_agents, hw, c, inv, out);
            break;
    }
    return av2x::check_launch("pixel_attn_kernel");
}
