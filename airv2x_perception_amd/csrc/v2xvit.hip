// Kernels of the V2X-ViT fusion that are not GEMMs / LayerNorms:
//
//   warp_affine_kernel     F.affine_grid + F.grid_sample(bilinear, zeros, align_corners=True) of
//                          torch_transformation_utils.warp_affine :337-381 on NHWC maps       HBM-bound
//   roi_mask_kernel        the same sampling with mode="nearest" of an all-ones image (:96-113),
//                          times the cav mask -> com_mask (:15-53)
//   add_agent_vector       x[l,:,:,c] += v[l,c]   (RTE, v2xvit_basic.py:58-80)
//   hgt_attention_kernel   HGTCavAttention per pixel (hmsa.py:133-151) on the FOLDED projections
//                          (relation_att / relation_msg multiplied into the q / v Linear weights on
//                          the host), masked softmax over agents, one wave per pixel
//   window_attn_kernel     BaseWindowAttention (mswin.py:52-96): one thread per (token, head)
//   gap3_kernel            mean over H,W of (sw + mw + bw)             (split_attn.py:48-50)
//   split_combine_kernel   radix softmax over the 3 branches + weighted sum + residual (:55-61)
#include <type_traits>

#include "av2x_common.hpp"
#include "split_attn_rows.hpp"

namespace {

// ---- element access for fp32 and bf16 (AMP mode: bf16 activations in HBM, arithmetic in fp32) buffers
__device__ __forceinline__ float bf16_bits_to_float(unsigned hi16) { return __builtin_bit_cast(float, hi16); }
template <typename T> __device__ __forceinline__ float4 ld4(const T* p);
template <> __device__ __forceinline__ float4 ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 ld4<__bf16>(const __bf16* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(bf16_bits_to_float(u.x << 16), bf16_bits_to_float(u.x & 0xffff0000u), bf16_bits_to_float(u.y << 16),
                       bf16_bits_to_float(u.y & 0xffff0000u));
}
template <typename T> __device__ __forceinline__ float ld1(const T* p) { return (float)*p; }
typedef __bf16 v2x_bf16x4 __attribute__((ext_vector_type(4)));
typedef float v2x_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(__bf16* p, float4 v) {   // round-to-nearest-even
    const v2x_f32x4 f = {v.x, v.y, v.z, v.w};
    *reinterpret_cast<v2x_bf16x4*>(p) = __builtin_convertvector(f, v2x_bf16x4);
}
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(__bf16* p, float v) { *p = (__bf16)v; }

// torch.linspace(-1, 1, n)[i] as ATen computes it (symmetric halves)
__device__ __forceinline__ float lin_m1_1(int i, int n) {
    if (n <= 1) return -1.0f;
    const float step = 2.0f / (float)(n - 1);
    return (i < n / 2) ? (-1.0f + step * (float)i) : (1.0f - step * (float)(n - 1 - i));
}

// AC = align_corners of F.affine_grid / F.grid_sample: true for warp_affine (:337-381), false for
// warp_affine_simple (:327-334): base grid linspace(-1,1,n)*(n-1)/n and unnormalise ((g+1)*n - 1)/2.
template <int CK, bool AC>  // C = 64 * CK: 16 lanes x float4 per pixel chunk
__global__ __launch_bounds__(256) void warp_affine_kernel(const float* __restrict__ src, const float* __restrict__ theta,
                                                          float* __restrict__ dst, int H, int W, const float* __restrict__ addv) {
    const int t = threadIdx.x & 15;
    const int pix = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int n = blockIdx.y;
    if (pix >= H * W) return;
    const int i = pix / W, j = pix - i * W;
    const float* th = theta + n * 6;
    float xn = lin_m1_1(j, W), yn = lin_m1_1(i, H);
    if (!AC) { xn = (xn * (float)(W - 1)) / (float)W; yn = (yn * (float)(H - 1)) / (float)H; }
    const float gx = th[0] * xn + th[1] * yn + th[2];
    const float gy = th[3] * xn + th[4] * yn + th[5];
    // grid_sampler_unnormalize
    const float ix = AC ? ((gx + 1.f) * 0.5f) * (float)(W - 1) : ((gx + 1.f) * (float)W - 1.f) * 0.5f;
    const float iy = AC ? ((gy + 1.f) * 0.5f) * (float)(H - 1) : ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - x0f, wx0 = (x0f + 1.f) - ix, wy1 = iy - y0f, wy0 = (y0f + 1.f) - iy;
    const float w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;  // nw, ne, sw, se
    const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)x1 < (unsigned)W;
    const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)y1 < (unsigned)H;
    constexpr int C = 64 * CK;
    const float* base = src + (size_t)n * H * W * C + 4 * t;
    float* out = dst + ((size_t)n * H * W + pix) * C + 4 * t;
#pragma unroll
    for (int k = 0; k < CK; ++k) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // addv: the source map is src + addv[n] (a per-agent channel vector, e.g. the RTE embedding) inside the map, zero outside --
        // the bits of adding the vector to the stored map first and sampling that
        const float4 b = addv ? *reinterpret_cast<const float4*>(addv + (size_t)n * C + 4 * t + 64 * k) : make_float4(0.f, 0.f, 0.f, 0.f);
        auto add = [&](bool ok, int yy, int xx, float w) {
            if (ok) {
                float4 v = *reinterpret_cast<const float4*>(base + ((size_t)yy * W + xx) * C + 64 * k);
                if (addv) { v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
                acc.x += v.x * w; acc.y += v.y * w; acc.z += v.z * w; acc.w += v.w * w;
            }
        };
        add(vy0 && vx0, y0, x0, w00);
        add(vy0 && vx1, y0, x1, w01);
        add(vy1 && vx0, y1, x0, w10);
        add(vy1 && vx1, y1, x1, w11);
        *reinterpret_cast<float4*>(out + 64 * k) = acc;
    }
}

__global__ void roi_mask_kernel(const float* __restrict__ theta, const int* __restrict__ cav_mask, float* __restrict__ mask,
                                int n, int H, int W) {
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    const int a = blockIdx.y;
    if (pix >= H * W) return;
    const int i = pix / W, j = pix - i * W;
    const float* th = theta + a * 6;
    const float xn = lin_m1_1(j, W), yn = lin_m1_1(i, H);
    const float gx = th[0] * xn + th[1] * yn + th[2];
    const float gy = th[3] * xn + th[4] * yn + th[5];
    const float ix = nearbyintf(((gx + 1.f) * 0.5f) * (float)(W - 1));   // mode="nearest": round half to even
    const float iy = nearbyintf(((gy + 1.f) * 0.5f) * (float)(H - 1));
    const bool in = ix >= 0.f && ix <= (float)(W - 1) && iy >= 0.f && iy <= (float)(H - 1);
    mask[(size_t)a * H * W + pix] = (in && cav_mask[a]) ? 1.f : 0.f;
}

__global__ void add_agent_vector_kernel(const float4* x, const float4* __restrict__ v, float4* out, size_t n4_per_agent, int c4) {
    const int a = blockIdx.y;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4_per_agent; i += (size_t)gridDim.x * blockDim.x) {
        float4 r = x[(size_t)a * n4_per_agent + i];
        const float4 b = v[(size_t)a * c4 + (i % c4)];
        r.x += b.x; r.y += b.y; r.z += b.z; r.w += b.w;
        out[(size_t)a * n4_per_agent + i] = r;
    }
}

// proj row layout (1280 floats): [q'(.->type0) | q'(.->type1) | k | v'(type0<-.) | v'(type1<-.)], 8 heads x 32 each
template <typename T>
struct HgtParams {
    const T* proj;       // (n, HW, 1280)
    const float* mask;   // (n, HW) com_mask of the KEY agent at the pixel
    T* out;              // (n, HW, 256)
    int n, hw;
    int nq;              // query agents 0 .. nq-1 are computed (nq = n, or 1 when only the ego's output is consumed)
    int types[32];
    int need_v[2];       // whether any query agent is of type 0 / 1 (which v' blocks are read)
    float scale;
};

template <typename T>
__global__ __launch_bounds__(256) void hgt_attention_kernel(const HgtParams<T> p) {
    const int lane = threadIdx.x & 63;
    const int pix = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= p.hw) return;
    const int col = lane * 4;   // head = lane / 8, 4 of its 32 dims
    constexpr int PC = 1280;
    for (int i = 0; i < p.nq; ++i) {
        const int ti = p.types[i];
        const T* qi = p.proj + ((size_t)i * p.hw + pix) * PC;
        const float4 q0 = ld4(qi + col);         // keys of type 0
        const float4 q1 = ld4(qi + 256 + col);   // keys of type 1
        float m = -INFINITY, l = 0.f;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < p.n; ++j) {
            if (p.mask[(size_t)j * p.hw + pix] == 0.f) continue;   // masked_fill(mask == 0, -inf): exp(-inf) = 0
            const T* kj = p.proj + ((size_t)j * p.hw + pix) * PC;
            const float4 k = ld4(kj + 512 + col);
            const float4 q = p.types[j] ? q1 : q0;
            float s = q.x * k.x + q.y * k.y + q.z * k.z + q.w * k.w;
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);   // 8 lanes = one head
            s *= p.scale;
            const float4 v = ld4(kj + 768 + 256 * ti + col);
            const float mn = fmaxf(m, s);
            const float alpha = expf(m - mn), pj = expf(s - mn);
            l = l * alpha + pj;
            o.x = fmaf(pj, v.x, o.x * alpha); o.y = fmaf(pj, v.y, o.y * alpha);
            o.z = fmaf(pj, v.z, o.z * alpha); o.w = fmaf(pj, v.w, o.w * alpha);
            m = mn;
        }
        const float inv = 1.0f / l;
        st4(p.out + ((size_t)i * p.hw + pix) * 256 + col, make_float4(o.x * inv, o.y * inv, o.z * inv, o.w * inv));
    }
}

// Same arithmetic in the same order, with the keys and both value projections of all n <= NMAX agents of the pixel loaded ONCE into
// registers (the kernel above re-reads k_j and v_j for every query agent: n times the pixel's 2.5 n KB through the L1).  One wave
// per pixel; a masked key agent is skipped exactly as above, so the results are bit-identical.
template <typename T, int NMAX>
__global__ __launch_bounds__(256) void hgt_attention_reg_kernel(const HgtParams<T> p) {
    const int lane = threadIdx.x & 63;
    const int pix = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= p.hw) return;
    const int col = lane * 4;
    constexpr int PC = 1280;
    float4 k[NMAX], v0[NMAX], v1[NMAX];
    bool vis[NMAX];
    const bool need0 = p.need_v[0], need1 = p.need_v[1];
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
        vis[j] = j < p.n && p.mask[(size_t)j * p.hw + pix] != 0.f;
        if (vis[j]) {
            const T* kj = p.proj + ((size_t)j * p.hw + pix) * PC;
            k[j] = ld4(kj + 512 + col);
            if (need0) v0[j] = ld4(kj + 768 + col);
            if (need1) v1[j] = ld4(kj + 1024 + col);
        }
    }
    for (int i = 0; i < p.nq; ++i) {
        const int ti = p.types[i];
        const T* qi = p.proj + ((size_t)i * p.hw + pix) * PC;
        const float4 q0 = ld4(qi + col);
        const float4 q1 = ld4(qi + 256 + col);
        float m = -INFINITY, l = 0.f;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < NMAX; ++j) {
            if (!vis[j]) continue;
            const float4 q = p.types[j] ? q1 : q0;
            float s = q.x * k[j].x + q.y * k[j].y + q.z * k[j].z + q.w * k[j].w;
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
            s *= p.scale;
            const float4 v = ti ? v1[j] : v0[j];
            const float mn = fmaxf(m, s);
            const float alpha = expf(m - mn), pj = expf(s - mn);
            l = l * alpha + pj;
            o.x = fmaf(pj, v.x, o.x * alpha); o.y = fmaf(pj, v.y, o.y * alpha);
            o.z = fmaf(pj, v.z, o.z * alpha); o.w = fmaf(pj, v.w, o.w * alpha);
            m = mn;
        }
        const float inv = 1.0f / l;
        st4(p.out + ((size_t)i * p.hw + pix) * 256 + col, make_float4(o.x * inv, o.y * inv, o.z * inv, o.w * inv));
    }
}

// bf16 storage, 16-byte loads: a lane owns 8 of the 256 columns (head = lane32 / 4), a wave two pixels.  Keys and values stay packed
// (bf16 pairs) in registers and are widened on use.  The additions happen in the order of the kernels above -- the two 4-dim partial
// dot products of a lane are exactly what lanes 2 j and 2 j + 1 hold there, then the same butterfly over the head -- so on the same
// inputs the result equals theirs bit for bit (tests/test_gpu_bf16_activations.py).
__device__ __forceinline__ void bf16x8_to_f32(const uint4 u, float4& a, float4& b) {
    a = make_float4(bf16_bits_to_float(u.x << 16), bf16_bits_to_float(u.x & 0xffff0000u), bf16_bits_to_float(u.y << 16), bf16_bits_to_float(u.y & 0xffff0000u));
    b = make_float4(bf16_bits_to_float(u.z << 16), bf16_bits_to_float(u.z & 0xffff0000u), bf16_bits_to_float(u.w << 16), bf16_bits_to_float(u.w & 0xffff0000u));
}

template <int NMAX>
__global__ __launch_bounds__(256) void hgt_attention_bf16x8_kernel(const HgtParams<__bf16> p) {
    const int lane = threadIdx.x & 63, l32 = lane & 31;
    const int pix = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);
    if (pix >= p.hw) return;
    const int col = l32 * 8;
    constexpr int PC = 1280;
    uint4 k[NMAX], v0[NMAX], v1[NMAX];
    bool vis[NMAX];
    const bool need0 = p.need_v[0], need1 = p.need_v[1];
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
        vis[j] = j < p.n && p.mask[(size_t)j * p.hw + pix] != 0.f;
        if (vis[j]) {
            const __bf16* kj = p.proj + ((size_t)j * p.hw + pix) * PC;
            k[j] = *reinterpret_cast<const uint4*>(kj + 512 + col);
            if (need0) v0[j] = *reinterpret_cast<const uint4*>(kj + 768 + col);
            if (need1) v1[j] = *reinterpret_cast<const uint4*>(kj + 1024 + col);
        }
    }
    for (int i = 0; i < p.nq; ++i) {
        const int ti = p.types[i];
        const __bf16* qi = p.proj + ((size_t)i * p.hw + pix) * PC;
        float4 q0a, q0b, q1a, q1b;
        bf16x8_to_f32(*reinterpret_cast<const uint4*>(qi + col), q0a, q0b);
        bf16x8_to_f32(*reinterpret_cast<const uint4*>(qi + 256 + col), q1a, q1b);
        float m = -INFINITY, l = 0.f;
        float4 oa = make_float4(0.f, 0.f, 0.f, 0.f), ob = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < NMAX; ++j) {
            if (!vis[j]) continue;
            float4 ka, kb, va, vb;
            bf16x8_to_f32(k[j], ka, kb);
            const float4 qa = p.types[j] ? q1a : q0a, qb = p.types[j] ? q1b : q0b;
            const float sa = qa.x * ka.x + qa.y * ka.y + qa.z * ka.z + qa.w * ka.w;
            const float sb = qb.x * kb.x + qb.y * kb.y + qb.z * kb.z + qb.w * kb.w;
            float s = sa + sb;
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);   // 4 lanes = one head
            s *= p.scale;
            bf16x8_to_f32(ti ? v1[j] : v0[j], va, vb);
            const float mn = fmaxf(m, s);
            const float alpha = expf(m - mn), pj = expf(s - mn);
            l = l * alpha + pj;
            oa.x = fmaf(pj, va.x, oa.x * alpha); oa.y = fmaf(pj, va.y, oa.y * alpha);
            oa.z = fmaf(pj, va.z, oa.z * alpha); oa.w = fmaf(pj, va.w, oa.w * alpha);
            ob.x = fmaf(pj, vb.x, ob.x * alpha); ob.y = fmaf(pj, vb.y, ob.y * alpha);
            ob.z = fmaf(pj, vb.z, ob.z * alpha); ob.w = fmaf(pj, vb.w, ob.w * alpha);
            m = mn;
        }
        const float inv = 1.0f / l;
        __bf16* dst = p.out + ((size_t)i * p.hw + pix) * 256 + col;
        st4(dst, make_float4(oa.x * inv, oa.y * inv, oa.z * inv, oa.w * inv));
        st4(dst + 4, make_float4(ob.x * inv, ob.y * inv, ob.z * inv, ob.w * inv));
    }
}

// one thread per (token, head); qkv row = [q | k | v], each heads*DHD wide, at column offset `coff` of a `ctot` row
template <int DHD, int WS, typename T>
__global__ __launch_bounds__(256) void window_attn_kernel(const T* __restrict__ qkv, int ctot, int coff,
                                                          const float* __restrict__ pos, T* __restrict__ out,
                                                          int n, int H, int W, int heads, float scale) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n * H * W * heads;
    if (gid >= total) return;
    const int head = (int)(gid % heads);
    const size_t tok = gid / heads;
    const int pix = (int)(tok % ((size_t)H * W));
    const int a = (int)(tok / ((size_t)H * W));
    const int y = pix / W, x = pix - y * W;
    const int wy0 = (y / WS) * WS, wx0 = (x / WS) * WS, iy = y - wy0, ixx = x - wx0;
    const int inner = heads * DHD;
    const T* row = qkv + tok * ctot + coff + head * DHD;
    float q[DHD], o[DHD];
#pragma unroll
    for (int d = 0; d < DHD; d += 4) {
        const float4 v = ld4(row + d);
        q[d] = v.x; q[d + 1] = v.y; q[d + 2] = v.z; q[d + 3] = v.w;
    }
    float s[WS * WS];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < WS * WS; ++j) {
        const int jy = j / WS, jx = j % WS;
        const T* kr = qkv + ((size_t)a * H * W + (size_t)(wy0 + jy) * W + (wx0 + jx)) * ctot + coff + inner + head * DHD;
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < DHD; d += 4) {
            const float4 k = ld4(kr + d);
            acc = fmaf(q[d], k.x, acc); acc = fmaf(q[d + 1], k.y, acc); acc = fmaf(q[d + 2], k.z, acc); acc = fmaf(q[d + 3], k.w, acc);
        }
        // dots * scale + pos_embedding[rel_y][rel_x], rel = idx[j] - idx[i] + ws - 1   (mswin.py:13-18, :77-80)
        acc = acc * scale + pos[(jy - iy + WS - 1) * (2 * WS - 1) + (jx - ixx + WS - 1)];
        s[j] = acc;
        m = fmaxf(m, acc);
    }
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < WS * WS; ++j) { s[j] = expf(s[j] - m); l += s[j]; }
    const float inv = 1.0f / l;
#pragma unroll
    for (int d = 0; d < DHD; ++d) o[d] = 0.f;
#pragma unroll
    for (int j = 0; j < WS * WS; ++j) {
        const int jy = j / WS, jx = j % WS;
        const T* vr = qkv + ((size_t)a * H * W + (size_t)(wy0 + jy) * W + (wx0 + jx)) * ctot + coff + 2 * inner + head * DHD;
        const float pj = s[j] * inv;
#pragma unroll
        for (int d = 0; d < DHD; d += 4) {
            const float4 v = ld4(vr + d);
            o[d] = fmaf(pj, v.x, o[d]); o[d + 1] = fmaf(pj, v.y, o[d + 1]); o[d + 2] = fmaf(pj, v.z, o[d + 2]); o[d + 3] = fmaf(pj, v.w, o[d + 3]);
        }
    }
    T* dst = out + tok * inner + head * DHD;
#pragma unroll
    for (int d = 0; d < DHD; d += 4) st4(dst + d, make_float4(o[d], o[d + 1], o[d + 2], o[d + 3]));
}

// MFMA form of the 4x4-window attention (T = 16 tokens): one wave per (window, head) task.
// v_mfma_f32_16x16x4_f32: A[row = lane%16][k = lane/16], B[k = lane/16][col = lane%16],
// D reg r of lane l = D[row = 4*(l/16) + r][col = l%16].
//   S^T = K Q^T  (row = key j, col = query i): lane (t = l%16, h = l/16) loads float4 K[t][4(h+4g)..] and
//         Q[t][4(h+4g)..] -- four MFMAs per float4 pair, any K-permutation is valid as A and B use the same one;
//         D reg r of lane l = score(query i = l%16, key j = 4h + r), i.e. key (jy = h, jx = r): the softmax over
//         keys is 4 in-lane values x the lanes l, l^16, l^32, l^48, and P is ALREADY in the A-operand layout
//         of the second product (row = i = l%16, k-slot = h), no transpose through LDS.
//   O   = P V    : MFMA r uses A = P reg r, B = V[key (h, r)][n0 + l%16]; D reg r' of lane l = O[query (l/16, r')][n0 + l%16].
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int DH, typename T>
__global__ __launch_bounds__(256) void window_attn_mfma_kernel(const T* __restrict__ qkv, int ctot, int coff,
                                                               const float* __restrict__ pos, T* __restrict__ out,
                                                               int n, int H, int W, int heads, float scale) {
    constexpr int WS = 4, NB = DH / 16;
    const int lane = threadIdx.x & 63;
    const int t = lane & 15, h = lane >> 4;
    const int wx_n = W / WS, wy_n = H / WS;
    const long long tasks = (long long)n * wy_n * wx_n * heads;
    const int inner = heads * DH;
    // relative-position bias of (query i = t, key (h, r)): pos[(jy - iy + 3) * 7 + (jx - ix + 3)]
    const int iy = t >> 2, ix = t & 3;
    float bias[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[r] = pos[(h - iy + WS - 1) * (2 * WS - 1) + (r - ix + WS - 1)];
    for (long long task = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); task < tasks; task += (long long)gridDim.x * 4) {
        const int head = (int)(task % heads);
        long long wdw = task / heads;
        const int wx = (int)(wdw % wx_n); wdw /= wx_n;
        const int wy = (int)(wdw % wy_n);
        const int a = (int)(wdw / wy_n);
        const size_t pix0 = ((size_t)a * H + (size_t)wy * WS) * W + (size_t)wx * WS;
        // token t of the window: pixel (wy*4 + t/4, wx*4 + t%4)
        const T* rowt = qkv + (pix0 + (size_t)(t >> 2) * W + (t & 3)) * ctot + coff + head * DH;
        f32x4_t st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            const float4 qq = ld4(rowt + 4 * (h + 4 * g));
            const float4 kk = ld4(rowt + inner + 4 * (h + 4 * g));
            st = __builtin_amdgcn_mfma_f32_16x16x4f32(kk.x, qq.x, st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_16x16x4f32(kk.y, qq.y, st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_16x16x4f32(kk.z, qq.z, st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_16x16x4f32(kk.w, qq.w, st, 0, 0, 0);
        }
        float sc[4], m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) { sc[r] = st[r] * scale + bias[r]; m = fmaxf(m, sc[r]); }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float l = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { sc[r] = expf(sc[r] - m); l += sc[r]; }
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / l;
        if constexpr (std::is_same<T, __bf16>::value) {
            // bf16 storage: the window's V tile (16 keys x DH) comes in and its O tile goes out as 16-byte pieces through a per-wave
            // LDS tile (the element-wise form below moves 2 bytes per lane and instruction: 2.6 TB/s at DH = 64)
            __shared__ __attribute__((aligned(16))) __bf16 vtile[4][16 * DH], otile[4][16 * DH];
            const int wv = threadIdx.x >> 6;
            constexpr int PPK = DH / 8;                      // 16-byte pieces per key / query row
#pragma unroll
            for (int e = lane; e < 16 * PPK; e += 64) {
                const int key = e / PPK, part = e % PPK;
                *reinterpret_cast<uint4*>(&vtile[wv][key * DH + part * 8]) = *reinterpret_cast<const uint4*>(
                    qkv + (pix0 + (size_t)(key >> 2) * W + (key & 3)) * ctot + coff + 2 * inner + head * DH + part * 8);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                f32x4_t o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    o = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[r] * inv, (float)vtile[wv][(4 * h + r) * DH + nb * 16 + t], o, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) otile[wv][(4 * h + r) * DH + nb * 16 + t] = (__bf16)o[r];
            }
#pragma unroll
            for (int e = lane; e < 16 * PPK; e += 64) {
                const int qi = e / PPK, part = e % PPK;
                *reinterpret_cast<uint4*>(out + (pix0 + (size_t)(qi >> 2) * W + (qi & 3)) * inner + head * DH + part * 8) =
                    *reinterpret_cast<const uint4*>(&otile[wv][qi * DH + part * 8]);
            }
        } else {
            // V rows of the keys (jy = h, jx = r)
            const T* vrow = qkv + (pix0 + (size_t)h * W) * ctot + coff + 2 * inner + head * DH + t;
            T* orow = out + (pix0 + (size_t)h * W) * inner + head * DH + t;   // query (iy = h, ix = r')
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                f32x4_t o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    o = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[r] * inv, ld1(vrow + (size_t)r * ctot + nb * 16), o, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) st1(orow + (size_t)r * inner + nb * 16, o[r]);
            }
        }
    }
}

// gap[a][c] = mean over pixels of (s0 + s1 + s2), deterministic two-stage reduction:
//   stage 1: grid (C/64, n, GAP_CHUNKS), block 256 = 4 pixel groups x 64 channels -> part[a][chunk][c]
//   stage 2: grid (C/64, n): sum of the chunks in fixed order, / hw
constexpr int GAP_CHUNKS = 128;

template <typename T>
__global__ __launch_bounds__(256) void gap3_stage1(const T* __restrict__ s0, const T* __restrict__ s1,
                                                   const T* __restrict__ s2, float* __restrict__ partial, int hw, int C) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6, a = blockIdx.y, ch = blockIdx.z;
    const int per = (hw + GAP_CHUNKS - 1) / GAP_CHUNKS;
    const int p0 = ch * per, p1 = min(hw, p0 + per);
    float acc = 0.f;
    for (int p = p0 + g; p < p1; p += 4) {
        const size_t o = ((size_t)a * hw + p) * C + c;
        acc += (ld1(s0 + o) + ld1(s1 + o)) + ld1(s2 + o);
    }
    part[g][threadIdx.x & 63] = acc;
    __syncthreads();
    if (g == 0)
        partial[((size_t)a * GAP_CHUNKS + ch) * C + c] =
            ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
}

__global__ void gap3_stage2(const float* __restrict__ partial, float* __restrict__ gap, int hw, int C) {
    const int c = blockIdx.x * 64 + threadIdx.x, a = blockIdx.y;
    float acc = 0.f;
    for (int ch = 0; ch < GAP_CHUNKS; ++ch) acc += partial[((size_t)a * GAP_CHUNKS + ch) * C + c];
    gap[(size_t)a * C + c] = acc / (float)hw;
}

template <typename T>
__global__ void split_combine_kernel(const T* __restrict__ s0, const T* __restrict__ s1, const T* __restrict__ s2,
                                     const float* __restrict__ logits, const float4* __restrict__ res,
                                     const __bf16* __restrict__ delta, float4* __restrict__ out, size_t n4_per_agent, int C) {
    const int a = blockIdx.y;
    const int c4 = C / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4_per_agent; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4) * 4;
        float w[3][4];
        av2x::split_attn_weights(logits + (size_t)a * 3 * C, c, C, w);
        const size_t o = (size_t)a * n4_per_agent + i;
        const float4 x0 = ld4(s0 + 4 * o), x1 = ld4(s1 + 4 * o), x2 = ld4(s2 + 4 * o);
        float4 r = res[o];
        if (delta) {     // a residual add still pending on the stream (the bf16 output of the previous sub-layer): r = res + delta first
            const float4 dl = ld4(delta + 4 * o);
            r.x += dl.x; r.y += dl.y; r.z += dl.z; r.w += dl.w;
        }
        const float4 y = av2x::split_attn_combine4(x0, x1, x2, w, r);
        out[o] = y;
    }
}

}  // namespace

template <bool AC>
static int warp_launch(const float* src, const float* theta, float* dst, int32_t n, int32_t h, int32_t w, int32_t c,
                       av2x_stream_t stream, const float* addv = nullptr) {
    if (n == 0) return 0;
    if (!src || !theta || !dst) return av2x::fail("av2x_warp_affine: null argument");
    if (n < 0 || h <= 0 || w <= 0) return av2x::fail("av2x_warp_affine: bad sizes");
    const dim3 grid((h * w + 15) / 16, n), block(256);
    hipStream_t st = av2x::as_stream(stream);
    switch (c) {
        case 64: hipLaunchKernelGGL((warp_affine_kernel<1, AC>), grid, block, 0, st, src, theta, dst, h, w, addv); break;
        case 128: hipLaunchKernelGGL((warp_affine_kernel<2, AC>), grid, block, 0, st, src, theta, dst, h, w, addv); break;
        case 256: hipLaunchKernelGGL((warp_affine_kernel<4, AC>), grid, block, 0, st, src, theta, dst, h, w, addv); break;
        default: return av2x::fail("av2x_warp_affine: c=%d unsupported (64/128/256)", c);
    }
    return av2x::check_launch("warp_affine_kernel");
}

extern "C" int av2x_warp_affine(const float* src, const float* theta, float* dst, int32_t n, int32_t h, int32_t w, int32_t c,
                                av2x_stream_t stream) {
    return warp_launch<true>(src, theta, dst, n, h, w, c, stream);
}

extern "C" int av2x_warp_affine_add(const float* src, const float* theta, const float* addv, float* dst, int32_t n, int32_t h, int32_t w,
                                    int32_t c, av2x_stream_t stream) {
    if (n && !addv) return av2x::fail("av2x_warp_affine_add: null argument");
    return warp_launch<true>(src, theta, dst, n, h, w, c, stream, addv);
}

extern "C" int av2x_warp_affine_simple(const float* src, const float* theta, float* dst, int32_t n, int32_t h, int32_t w,
                                       int32_t c, av2x_stream_t stream) {
    return warp_launch<false>(src, theta, dst, n, h, w, c, stream);
}

extern "C" int av2x_roi_mask(const float* theta, const int32_t* cav_mask, float* mask, int32_t n, int32_t h, int32_t w,
                             av2x_stream_t stream) {
    if (n == 0) return 0;
    if (!theta || !cav_mask || !mask) return av2x::fail("av2x_roi_mask: null argument");
    hipLaunchKernelGGL(roi_mask_kernel, dim3((h * w + 255) / 256, n), dim3(256), 0, av2x::as_stream(stream), theta, cav_mask,
                       mask, n, h, w);
    return av2x::check_launch("roi_mask_kernel");
}

extern "C" int av2x_add_agent_vector_to(const float* x, const float* v, float* out, int32_t n, int64_t elems_per_agent, int32_t c,
                                        av2x_stream_t stream);

extern "C" int av2x_add_agent_vector(float* x, const float* v, int32_t n, int64_t elems_per_agent, int32_t c, av2x_stream_t stream) {
    return av2x_add_agent_vector_to(x, v, x, n, elems_per_agent, c, stream);
}

extern "C" int av2x_add_agent_vector_to(const float* x, const float* v, float* out, int32_t n, int64_t elems_per_agent, int32_t c,
                                        av2x_stream_t stream) {
    if (n == 0) return 0;
    if (!x || !v || !out) return av2x::fail("av2x_add_agent_vector: null argument");
    if (c % 4 || elems_per_agent % c) return av2x::fail("av2x_add_agent_vector: bad sizes");
    const size_t n4 = (size_t)elems_per_agent / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(add_agent_vector_kernel, dim3((unsigned)blocks, n), dim3(256), 0, av2x::as_stream(stream),
                       reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(v), reinterpret_cast<float4*>(out), n4, c / 4);
    return av2x::check_launch("add_agent_vector_kernel");
}

template <typename T>
static int hgt_launch(const T* proj, const float* mask, const int32_t* types_host, T* out, int32_t n, int32_t n_query, int32_t hw,
                      int32_t heads, int32_t dim_head, av2x_stream_t stream) {
    if (n_query < 1 || n_query > n) return av2x::fail("av2x_hgt_attention: n_query=%d outside 1..%d", n_query, n);
    if (!proj || !mask || !types_host || !out) return av2x::fail("av2x_hgt_attention: null argument");
    if (heads != 8 || dim_head != 32) return av2x::fail("av2x_hgt_attention: heads=%d dim_head=%d unsupported (8 x 32)", heads, dim_head);
    if (n < 1 || n > 32 || hw <= 0) return av2x::fail("av2x_hgt_attention: bad sizes");
    HgtParams<T> p;
    p.proj = proj; p.mask = mask; p.out = out; p.n = n; p.hw = hw; p.nq = n_query;
    for (int i = 0; i < 32; ++i) p.types[i] = i < n ? (types_host[i] != 0) : 0;
    p.scale = 1.0f / sqrtf((float)dim_head);
    p.need_v[0] = p.need_v[1] = 0;
    for (int i = 0; i < n_query; ++i) p.need_v[p.types[i]] = 1;
    if constexpr (std::is_same<T, __bf16>::value) {
        if (n <= 8) {
            hipLaunchKernelGGL((hgt_attention_bf16x8_kernel<8>), dim3((hw + 7) / 8), dim3(256), 0, av2x::as_stream(stream), p);
            return av2x::check_launch("hgt_attention_bf16x8_kernel");
        }
    }
    if (n <= 8) hipLaunchKernelGGL((hgt_attention_reg_kernel<T, 8>), dim3((hw + 3) / 4), dim3(256), 0, av2x::as_stream(stream), p);
    else hipLaunchKernelGGL(hgt_attention_kernel<T>, dim3((hw + 3) / 4), dim3(256), 0, av2x::as_stream(stream), p);
    return av2x::check_launch("hgt_attention_kernel");
}

extern "C" int av2x_hgt_attention_q(const float* proj, const float* mask, const int32_t* types_host, float* out, int32_t n,
                                    int32_t n_query, int32_t hw, int32_t heads, int32_t dim_head, av2x_stream_t stream) {
    return hgt_launch<float>(proj, mask, types_host, out, n, n_query, hw, heads, dim_head, stream);
}

extern "C" int av2x_hgt_attention_bf16(const uint16_t* proj, const float* mask, const int32_t* types_host, uint16_t* out, int32_t n,
                                       int32_t n_query, int32_t hw, int32_t heads, int32_t dim_head, av2x_stream_t stream) {
    return hgt_launch<__bf16>(reinterpret_cast<const __bf16*>(proj), mask, types_host, reinterpret_cast<__bf16*>(out), n, n_query, hw, heads,
                              dim_head, stream);
}

extern "C" int av2x_hgt_attention(const float* proj, const float* mask, const int32_t* types_host, float* out, int32_t n,
                                  int32_t hw, int32_t heads, int32_t dim_head, av2x_stream_t stream) {
    return av2x_hgt_attention_q(proj, mask, types_host, out, n, n, hw, heads, dim_head, stream);
}

template <typename T>
static int window_launch(const T* qkv, int32_t ctot, int32_t coff, const float* pos_embedding, T* out, int32_t n, int32_t h, int32_t w,
                         int32_t heads, int32_t dim_head, int32_t window, av2x_stream_t stream) {
    if (!qkv || !pos_embedding || !out) return av2x::fail("av2x_window_attention: null argument");
    if (h % (window & 0xff) || w % (window & 0xff))
        return av2x::fail("av2x_window_attention: map %dx%d not divisible by window %d", h, w, window & 0xff);
    if (ctot % 4 || coff % 4) return av2x::fail("av2x_window_attention: ctot / coff must be multiples of 4");
    const bool force_valu = (window & 0x100) != 0;   // test hook: the scalar reference kernel
    window &= 0xff;
    const size_t total = (size_t)n * h * w * heads;
    if (total == 0) return 0;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    const float scale = 1.0f / sqrtf((float)dim_head);
    hipStream_t st = av2x::as_stream(stream);
    if (window == 4 && !force_valu && (dim_head == 32 || dim_head == 64)) {
        const long long tasks = (long long)n * (h / 4) * (w / 4) * heads;
        const unsigned wgs = (unsigned)((tasks + 3) / 4 < 256 * 16 ? (tasks + 3) / 4 : 256 * 16);
        if (dim_head == 32) hipLaunchKernelGGL((window_attn_mfma_kernel<32, T>), dim3(wgs), block, 0, st, qkv, ctot, coff, pos_embedding, out, n, h, w, heads, scale);
        else hipLaunchKernelGGL((window_attn_mfma_kernel<64, T>), dim3(wgs), block, 0, st, qkv, ctot, coff, pos_embedding, out, n, h, w, heads, scale);
        return av2x::check_launch("window_attn_mfma_kernel");
    }
    if (dim_head == 16 && window == 2) hipLaunchKernelGGL((window_attn_kernel<16, 2, T>), grid, block, 0, st, qkv, ctot, coff, pos_embedding, out, n, h, w, heads, scale);
    else if (dim_head == 32 && window == 4) hipLaunchKernelGGL((window_attn_kernel<32, 4, T>), grid, block, 0, st, qkv, ctot, coff, pos_embedding, out, n, h, w, heads, scale);
    else if (dim_head == 64 && window == 4) hipLaunchKernelGGL((window_attn_kernel<64, 4, T>), grid, block, 0, st, qkv, ctot, coff, pos_embedding, out, n, h, w, heads, scale);
    else return av2x::fail("av2x_window_attention: (dim_head %d, window %d) unsupported: (16,2) (32,4) (64,4)", dim_head, window);
    return av2x::check_launch("window_attn_kernel");
}

extern "C" int av2x_window_attention(const float* qkv, int32_t ctot, int32_t coff, const float* pos_embedding, float* out,
                                     int32_t n, int32_t h, int32_t w, int32_t heads, int32_t dim_head, int32_t window,
                                     av2x_stream_t stream) {
    return window_launch<float>(qkv, ctot, coff, pos_embedding, out, n, h, w, heads, dim_head, window, stream);
}

extern "C" int av2x_window_attention_bf16(const uint16_t* qkv, int32_t ctot, int32_t coff, const float* pos_embedding, uint16_t* out,
                                          int32_t n, int32_t h, int32_t w, int32_t heads, int32_t dim_head, int32_t window,
                                          av2x_stream_t stream) {
    return window_launch<__bf16>(reinterpret_cast<const __bf16*>(qkv), ctot, coff, pos_embedding, reinterpret_cast<__bf16*>(out), n, h, w, heads,
                                 dim_head, window, stream);
}

template <typename T>
static int gap_launch(const T* s0, const T* s1, const T* s2, float* gap, float* scratch, int32_t n, int32_t hw, int32_t c,
                      av2x_stream_t stream) {
    if (n == 0) return 0;
    if (!s0 || !s1 || !s2 || !gap || !scratch) return av2x::fail("av2x_split_attn_gap: null argument");
    if (c % 64) return av2x::fail("av2x_split_attn_gap: c must be a multiple of 64");
    hipStream_t st = av2x::as_stream(stream);
    hipLaunchKernelGGL(gap3_stage1<T>, dim3(c / 64, n, GAP_CHUNKS), dim3(256), 0, st, s0, s1, s2, scratch, hw, c);
    hipLaunchKernelGGL(gap3_stage2, dim3(c / 64, n), dim3(64), 0, st, scratch, gap, hw, c);
    return av2x::check_launch("gap3");
}

extern "C" int av2x_split_attn_gap(const float* s0, const float* s1, const float* s2, float* gap, float* scratch, int32_t n,
                                   int32_t hw, int32_t c, av2x_stream_t stream) {
    return gap_launch<float>(s0, s1, s2, gap, scratch, n, hw, c, stream);
}

extern "C" int av2x_split_attn_gap_bf16(const uint16_t* s0, const uint16_t* s1, const uint16_t* s2, float* gap, float* scratch,
                                        int32_t n, int32_t hw, int32_t c, av2x_stream_t stream) {
    return gap_launch<__bf16>(reinterpret_cast<const __bf16*>(s0), reinterpret_cast<const __bf16*>(s1), reinterpret_cast<const __bf16*>(s2), gap,
                              scratch, n, hw, c, stream);
}

template <typename T>
static int combine_launch(const T* s0, const T* s1, const T* s2, const float* logits, const float* residual, const __bf16* delta,
                          float* out, int32_t n, int32_t hw, int32_t c, av2x_stream_t stream) {
    if (n == 0) return 0;
    if (!s0 || !s1 || !s2 || !logits || !residual || !out) return av2x::fail("av2x_split_attn_combine: null argument");
    if (c % 4) return av2x::fail("av2x_split_attn_combine: c must be a multiple of 4");
    const size_t n4 = (size_t)hw * c / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(split_combine_kernel<T>, dim3((unsigned)blocks, n), dim3(256), 0, av2x::as_stream(stream), s0, s1, s2, logits,
                       reinterpret_cast<const float4*>(residual), delta, reinterpret_cast<float4*>(out), n4, c);
    return av2x::check_launch("split_combine_kernel");
}

extern "C" int av2x_split_attn_combine(const float* s0, const float* s1, const float* s2, const float* logits,
                                       const float* residual, float* out, int32_t n, int32_t hw, int32_t c, av2x_stream_t stream) {
    return combine_launch<float>(s0, s1, s2, logits, residual, nullptr, out, n, hw, c, stream);
}

extern "C" int av2x_split_attn_combine_bf16(const uint16_t* s0, const uint16_t* s1, const uint16_t* s2, const float* logits,
                                            const float* residual, float* out, int32_t n, int32_t hw, int32_t c, av2x_stream_t stream) {
    return combine_launch<__bf16>(reinterpret_cast<const __bf16*>(s0), reinterpret_cast<const __bf16*>(s1), reinterpret_cast<const __bf16*>(s2),
                                  logits, residual, nullptr, out, n, hw, c, stream);
}

extern "C" int av2x_split_attn_combine_delta_bf16(const uint16_t* s0, const uint16_t* s1, const uint16_t* s2, const float* logits,
                                                  const float* residual, const uint16_t* delta, float* out, int32_t n, int32_t hw, int32_t c,
                                                  av2x_stream_t stream) {
    return combine_launch<__bf16>(reinterpret_cast<const __bf16*>(s0), reinterpret_cast<const __bf16*>(s1), reinterpret_cast<const __bf16*>(s2),
                                  logits, residual, reinterpret_cast<const __bf16*>(delta), out, n, hw, c, stream);
}
