// Backward kernels of the V2X-ViT fusion (`.train()` of Airv2xV2XVit; the reference differentiates
// models/v2xvit_modules/{hmsa.py, mswin.py, split_attn.py, v2xvit_basic.py} and torch_transformation_utils.warp_affine with torch autograd).
//
//   hgt_attention_backward_kernel   HGTCavAttention (hmsa.py:133-151) on the FOLDED projections [q'(->t0) | q'(->t1) | k | v'(t0<-) | v'(t1<-)]:
//                                   one wave per pixel (lane = head x 4 of its 32 dims, as the forward kernel); scores recomputed; every
//                                   gradient element is owned by one lane (read-modify-write of that lane's own dproj entries): no atomics
//   window_attn_backward_q / _kv    BaseWindowAttention (mswin.py:52-96): a thread per (token, head) like the forward kernel -- pass q: row
//                                   statistics, D = dO . O, dq, the pos_embedding gradient (2^-32 fixed point: LDS, then global atomics);
//                                   pass kv: dk, dv of the token as a key
//   split_sums_kernel               sum over pixels of dout * s_r per (agent, branch, channel) (gradient of the radix weights)
//   split_backward_kernel           ds_r = a_r * dout + dgap / hw
//   warp_affine_backward_gather_kernel   adjoint of the bilinear sampling of warp_affine (:337-381) as a GATHER over the few output pixels whose
//                                   footprint holds the source pixel (the warp is affine: their bounding box follows from theta); degenerate
//                                   thetas fall back, per image, to warp_affine_backward_kernel: scattered with 2^-32 fixed-point atomics
#include <cstdlib>

#include "av2x_common.hpp"

namespace {

constexpr float kFixF = 4294967296.0f;

// ------------------------------------------------------------------------------------------------------------ HGT attention
struct HgtBwdParams {
    const float* proj;    // (n, HW, 1280)
    const float* mask;    // (n, HW)
    const float* dout;    // (n, HW, 256)
    float* dproj;         // (n, HW, 1280), pre-zeroed
    int n, hw;
    int types[32];
    float scale;
};

__global__ __launch_bounds__(256) void hgt_attention_backward_kernel(const HgtBwdParams p) {
    const int lane = threadIdx.x & 63;
    const int pix = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= p.hw) return;
    const int col = lane * 4;
    constexpr int PC = 1280;
    auto head_sum = [](float s) { s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); return s; };
    for (int i = 0; i < p.n; ++i) {
        const int ti = p.types[i];
        const float* qi = p.proj + ((size_t)i * p.hw + pix) * PC;
        const float4 q0 = *reinterpret_cast<const float4*>(qi + col);
        const float4 q1 = *reinterpret_cast<const float4*>(qi + 256 + col);
        const float4 g = *reinterpret_cast<const float4*>(p.dout + ((size_t)i * p.hw + pix) * 256 + col);
        // pass 1: row max / sum and the output row (for D = dO . O)
        float m = -INFINITY, l = 0.f;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < p.n; ++j) {
            if (p.mask[(size_t)j * p.hw + pix] == 0.f) continue;
            const float* kj = p.proj + ((size_t)j * p.hw + pix) * PC;
            const float4 k = *reinterpret_cast<const float4*>(kj + 512 + col);
            const float4 q = p.types[j] ? q1 : q0;
            const float s = head_sum(q.x * k.x + q.y * k.y + q.z * k.z + q.w * k.w) * p.scale;
            const float4 v = *reinterpret_cast<const float4*>(kj + 768 + 256 * ti + col);
            const float mn = fmaxf(m, s);
            const float alpha = expf(m - mn), pj = expf(s - mn);
            l = l * alpha + pj;
            o.x = fmaf(pj, v.x, o.x * alpha); o.y = fmaf(pj, v.y, o.y * alpha);
            o.z = fmaf(pj, v.z, o.z * alpha); o.w = fmaf(pj, v.w, o.w * alpha);
            m = mn;
        }
        const float inv = 1.0f / l;
        const float Di = head_sum(g.x * o.x + g.y * o.y + g.z * o.z + g.w * o.w) * inv;
        // pass 2: gradients
        float4 dq0 = make_float4(0.f, 0.f, 0.f, 0.f), dq1 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < p.n; ++j) {
            if (p.mask[(size_t)j * p.hw + pix] == 0.f) continue;
            const size_t rowj = ((size_t)j * p.hw + pix) * PC;
            const float* kj = p.proj + rowj;
            const float4 k = *reinterpret_cast<const float4*>(kj + 512 + col);
            const int tj = p.types[j];
            const float4 q = tj ? q1 : q0;
            const float s = head_sum(q.x * k.x + q.y * k.y + q.z * k.z + q.w * k.w) * p.scale;
            const float4 v = *reinterpret_cast<const float4*>(kj + 768 + 256 * ti + col);
            const float pij = expf(s - m) * inv;
            const float dp = head_sum(g.x * v.x + g.y * v.y + g.z * v.z + g.w * v.w);
            const float ds = pij * (dp - Di) * p.scale;
            if (tj) { dq1.x = fmaf(ds, k.x, dq1.x); dq1.y = fmaf(ds, k.y, dq1.y); dq1.z = fmaf(ds, k.z, dq1.z); dq1.w = fmaf(ds, k.w, dq1.w); }
            else { dq0.x = fmaf(ds, k.x, dq0.x); dq0.y = fmaf(ds, k.y, dq0.y); dq0.z = fmaf(ds, k.z, dq0.z); dq0.w = fmaf(ds, k.w, dq0.w); }
            // this lane owns (pixel, agent j, its 4 columns) of dk and dv'(ti<-): plain read-modify-write, queries i in order
            float4* dk = reinterpret_cast<float4*>(p.dproj + rowj + 512 + col);
            float4 a = *dk;
            a.x = fmaf(ds, q.x, a.x); a.y = fmaf(ds, q.y, a.y); a.z = fmaf(ds, q.z, a.z); a.w = fmaf(ds, q.w, a.w);
            *dk = a;
            float4* dv = reinterpret_cast<float4*>(p.dproj + rowj + 768 + 256 * ti + col);
            float4 b = *dv;
            b.x = fmaf(pij, g.x, b.x); b.y = fmaf(pij, g.y, b.y); b.z = fmaf(pij, g.z, b.z); b.w = fmaf(pij, g.w, b.w);
            *dv = b;
        }
        float* dqi = p.dproj + ((size_t)i * p.hw + pix) * PC;
        *reinterpret_cast<float4*>(dqi + col) = dq0;
        *reinterpret_cast<float4*>(dqi + 256 + col) = dq1;
    }
}

// The same for n <= NMAX agents with every key / value row of the pixel and every dk / dv accumulator in registers: proj is read once and
// dproj written once (no read-modify-write per (query, key) pair, no pre-zeroing pass over the 1280-column gradient); the sums run over the
// queries in the same order as above -- bit-identical to it.
template <int NMAX>
__global__ __launch_bounds__(256) void hgt_attention_backward_reg_kernel(const HgtBwdParams p) {
    const int lane = threadIdx.x & 63;
    const int pix = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= p.hw) return;
    const int col = lane * 4;
    constexpr int PC = 1280;
    auto head_sum = [](float s) { s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); return s; };
    float4 k[NMAX], v0[NMAX], v1[NMAX], dk[NMAX], dv0[NMAX], dv1[NMAX];
    bool on[NMAX];
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
        on[j] = false;
        dk[j] = dv0[j] = dv1[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < p.n) {
            on[j] = p.mask[(size_t)j * p.hw + pix] != 0.f;
            const float* kj = p.proj + ((size_t)j * p.hw + pix) * PC;
            k[j] = *reinterpret_cast<const float4*>(kj + 512 + col);
            v0[j] = *reinterpret_cast<const float4*>(kj + 768 + col);
            v1[j] = *reinterpret_cast<const float4*>(kj + 1024 + col);
        }
    }
    for (int i = 0; i < p.n; ++i) {
        const int ti = p.types[i];
        const float* qi = p.proj + ((size_t)i * p.hw + pix) * PC;
        const float4 q0 = *reinterpret_cast<const float4*>(qi + col);
        const float4 q1 = *reinterpret_cast<const float4*>(qi + 256 + col);
        const float4 g = *reinterpret_cast<const float4*>(p.dout + ((size_t)i * p.hw + pix) * 256 + col);
        float m = -INFINITY, l = 0.f;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < NMAX; ++j) {
            if (j < p.n && on[j]) {
                const float4 q = p.types[j] ? q1 : q0;
                const float s = head_sum(q.x * k[j].x + q.y * k[j].y + q.z * k[j].z + q.w * k[j].w) * p.scale;
                const float4 v = ti ? v1[j] : v0[j];
                const float mn = fmaxf(m, s);
                const float alpha = expf(m - mn), pj = expf(s - mn);
                l = l * alpha + pj;
                o.x = fmaf(pj, v.x, o.x * alpha); o.y = fmaf(pj, v.y, o.y * alpha);
                o.z = fmaf(pj, v.z, o.z * alpha); o.w = fmaf(pj, v.w, o.w * alpha);
                m = mn;
            }
        }
        const float inv = 1.0f / l;
        const float Di = head_sum(g.x * o.x + g.y * o.y + g.z * o.z + g.w * o.w) * inv;
        float4 dq0 = make_float4(0.f, 0.f, 0.f, 0.f), dq1 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < NMAX; ++j) {
            if (j < p.n && on[j]) {
                const int tj = p.types[j];
                const float4 q = tj ? q1 : q0;
                const float s = head_sum(q.x * k[j].x + q.y * k[j].y + q.z * k[j].z + q.w * k[j].w) * p.scale;
                const float4 v = ti ? v1[j] : v0[j];
                const float pij = expf(s - m) * inv;
                const float dp = head_sum(g.x * v.x + g.y * v.y + g.z * v.z + g.w * v.w);
                const float ds = pij * (dp - Di) * p.scale;
                if (tj) { dq1.x = fmaf(ds, k[j].x, dq1.x); dq1.y = fmaf(ds, k[j].y, dq1.y); dq1.z = fmaf(ds, k[j].z, dq1.z); dq1.w = fmaf(ds, k[j].w, dq1.w); }
                else { dq0.x = fmaf(ds, k[j].x, dq0.x); dq0.y = fmaf(ds, k[j].y, dq0.y); dq0.z = fmaf(ds, k[j].z, dq0.z); dq0.w = fmaf(ds, k[j].w, dq0.w); }
                dk[j].x = fmaf(ds, q.x, dk[j].x); dk[j].y = fmaf(ds, q.y, dk[j].y); dk[j].z = fmaf(ds, q.z, dk[j].z); dk[j].w = fmaf(ds, q.w, dk[j].w);
                if (ti) { dv1[j].x = fmaf(pij, g.x, dv1[j].x); dv1[j].y = fmaf(pij, g.y, dv1[j].y); dv1[j].z = fmaf(pij, g.z, dv1[j].z); dv1[j].w = fmaf(pij, g.w, dv1[j].w); }
                else { dv0[j].x = fmaf(pij, g.x, dv0[j].x); dv0[j].y = fmaf(pij, g.y, dv0[j].y); dv0[j].z = fmaf(pij, g.z, dv0[j].z); dv0[j].w = fmaf(pij, g.w, dv0[j].w); }
            }
        }
        float* dqi = p.dproj + ((size_t)i * p.hw + pix) * PC;
        *reinterpret_cast<float4*>(dqi + col) = dq0;
        *reinterpret_cast<float4*>(dqi + 256 + col) = dq1;
    }
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
        if (j < p.n) {
            float* dj = p.dproj + ((size_t)j * p.hw + pix) * PC;
            *reinterpret_cast<float4*>(dj + 512 + col) = dk[j];
            *reinterpret_cast<float4*>(dj + 768 + col) = dv0[j];
            *reinterpret_cast<float4*>(dj + 1024 + col) = dv1[j];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------ window attention
template <int DHD, int WS>
__global__ __launch_bounds__(256) void window_attn_backward_q(const float* __restrict__ qkv, int ctot, int coff, const float* __restrict__ pos,
                                                              const float* __restrict__ out, const float* __restrict__ dout,
                                                              float* __restrict__ dqkv, float* __restrict__ stats,
                                                              unsigned long long* __restrict__ dpos, int n, int H, int W, int heads, float scale) {
    constexpr int NP = (2 * WS - 1) * (2 * WS - 1);
    __shared__ unsigned long long lpos[NP];
    for (int i = threadIdx.x; i < NP; i += 256) lpos[i] = 0ull;
    __syncthreads();
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n * H * W * heads;
    if (gid < total) {
        const int head = (int)(gid % heads);
        const size_t tok = gid / heads;
        const int pix = (int)(tok % ((size_t)H * W));
        const int a = (int)(tok / ((size_t)H * W));
        const int y = pix / W, x = pix - y * W;
        const int wy0 = (y / WS) * WS, wx0 = (x / WS) * WS, iy = y - wy0, ixx = x - wx0;
        const int inner = heads * DHD;
        const float* row = qkv + tok * ctot + coff + head * DHD;
        const float* orow = out + tok * inner + head * DHD;
        const float* grow = dout + tok * inner + head * DHD;
        float q[DHD], g[DHD], dq[DHD];
        float Di = 0.f;
#pragma unroll
        for (int d = 0; d < DHD; ++d) { q[d] = row[d]; g[d] = grow[d]; dq[d] = 0.f; Di = fmaf(g[d], orow[d], Di); }
        float s[WS * WS];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < WS * WS; ++j) {
            const int jy = j / WS, jx = j % WS;
            const float* kr = qkv + ((size_t)a * H * W + (size_t)(wy0 + jy) * W + (wx0 + jx)) * ctot + coff + inner + head * DHD;
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < DHD; ++d) acc = fmaf(q[d], kr[d], acc);
            acc = acc * scale + pos[(jy - iy + WS - 1) * (2 * WS - 1) + (jx - ixx + WS - 1)];
            s[j] = acc;
            m = fmaxf(m, acc);
        }
        float l = 0.f;
#pragma unroll
        for (int j = 0; j < WS * WS; ++j) l += expf(s[j] - m);
        const float inv = 1.0f / l;
#pragma unroll
        for (int j = 0; j < WS * WS; ++j) {
            const int jy = j / WS, jx = j % WS;
            const size_t krow = ((size_t)a * H * W + (size_t)(wy0 + jy) * W + (wx0 + jx)) * ctot + coff + head * DHD;
            const float* kr = qkv + krow + inner;
            const float* vr = qkv + krow + 2 * inner;
            float dp = 0.f;
#pragma unroll
            for (int d = 0; d < DHD; ++d) dp = fmaf(g[d], vr[d], dp);
            const float pij = expf(s[j] - m) * inv;
            const float ds = pij * (dp - Di);
#pragma unroll
            for (int d = 0; d < DHD; ++d) dq[d] = fmaf(ds, kr[d], dq[d]);
            if (ds != 0.f) atomicAdd(&lpos[(jy - iy + WS - 1) * (2 * WS - 1) + (jx - ixx + WS - 1)], (unsigned long long)__float2ll_rn(ds * kFixF));
        }
        float* st = stats + gid * 3;
        st[0] = m; st[1] = inv; st[2] = Di;
        float* dqr = dqkv + tok * ctot + coff + head * DHD;
#pragma unroll
        for (int d = 0; d < DHD; ++d) dqr[d] = dq[d] * scale;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NP; i += 256)
        if (lpos[i]) atomicAdd(dpos + i, lpos[i]);
}

template <int DHD, int WS>
__global__ __launch_bounds__(256) void window_attn_backward_kv(const float* __restrict__ qkv, int ctot, int coff, const float* __restrict__ pos,
                                                               const float* __restrict__ dout, const float* __restrict__ stats,
                                                               float* __restrict__ dqkv, int n, int H, int W, int heads, float scale) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n * H * W * heads;
    if (gid >= total) return;
    const int head = (int)(gid % heads);
    const size_t tok = gid / heads;
    const int pix = (int)(tok % ((size_t)H * W));
    const int a = (int)(tok / ((size_t)H * W));
    const int y = pix / W, x = pix - y * W;
    const int wy0 = (y / WS) * WS, wx0 = (x / WS) * WS, jy = y - wy0, jx = x - wx0;
    const int inner = heads * DHD;
    const float* row = qkv + tok * ctot + coff + head * DHD;
    float k[DHD], v[DHD], dk[DHD], dv[DHD];
#pragma unroll
    for (int d = 0; d < DHD; ++d) { k[d] = row[inner + d]; v[d] = row[2 * inner + d]; dk[d] = 0.f; dv[d] = 0.f; }
#pragma unroll
    for (int i = 0; i < WS * WS; ++i) {
        const int iy = i / WS, ixx = i % WS;
        const size_t itok = (size_t)a * H * W + (size_t)(wy0 + iy) * W + (wx0 + ixx);
        const float* qr = qkv + itok * ctot + coff + head * DHD;
        const float* gr = dout + itok * inner + head * DHD;
        const float* st = stats + (itok * heads + head) * 3;
        float sc = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < DHD; ++d) { sc = fmaf(qr[d], k[d], sc); dp = fmaf(gr[d], v[d], dp); }
        sc = sc * scale + pos[(jy - iy + WS - 1) * (2 * WS - 1) + (jx - ixx + WS - 1)];
        const float pij = expf(sc - st[0]) * st[1];
        const float ds = pij * (dp - st[2]) * scale;
#pragma unroll
        for (int d = 0; d < DHD; ++d) { dk[d] = fmaf(ds, qr[d], dk[d]); dv[d] = fmaf(pij, gr[d], dv[d]); }
    }
    float* drow = dqkv + tok * ctot + coff + head * DHD;
#pragma unroll
    for (int d = 0; d < DHD; ++d) { drow[inner + d] = dk[d]; drow[2 * inner + d] = dv[d]; }
}

// window = 4: the 16 tokens of a window are one v_mfma_f32_16x16x4_f32 tile -- the register tiling of fax_attention_backward_wave_kernel
// (train_fusion.hip) with one query / key "agent": a wave per (agent, window, head); lane (t = lane & 15, h = lane >> 4) holds
// X[token t][d = 16c + 4h .. + 3] of Q, K, V, dO (A operand with row = t, B operand with column = t alike);
//   S^T, P^T, dP^T, dS^T with row = key 4h + r, column = query t  ->  pos_embedding gradient (fixed point, LDS), dQ = dS K (dS^T IS the A operand)
//   S, P, dP, dS with row = query 4h + r, column = key t (operands swapped; statistics of query 4h + r by ds_bpermute)  ->  dK = dS^T Q, dV = P^T dO
// No stats buffer, no second pass over the keys, one launch.  The thread-per-(token, head) kernels above read every K / V row of the window
// from global memory per thread: 1.1 + 0.84 ms (dim_head 32) and 1.7 + 1.15 ms (64) per launch at the BASELINE grid.
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int DHD>
__global__ __launch_bounds__(256) void window_attn_backward_wave_kernel(const float* __restrict__ qkv, int ctot, int coff, const float* __restrict__ pos,
                                                                        const float* __restrict__ out, const float* __restrict__ dout,
                                                                        float* __restrict__ dqkv, unsigned long long* __restrict__ dpos, int n, int H, int W,
                                                                        int heads, float scale) {
    constexpr int NC = DHD / 16;
    __shared__ float lpos[64];
    __shared__ unsigned long long ldpos[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = lane & 15, h = lane >> 4;
    if (threadIdx.x < 64) lpos[threadIdx.x] = threadIdx.x < 49 ? pos[threadIdx.x] : 0.f;
    ldpos[wave][lane] = 0ull;
    __syncthreads();
    const int X = H / 4, Y = W / 4;
    const long long items = (long long)n * X * Y * heads;
    const int inner = heads * DHD;
    const int w1q = t >> 2, w2q = t & 3;
    const float kLog2e = 1.4426950408889634f;
    for (long long it = (long long)blockIdx.x * 4 + wave; it < items; it += (long long)gridDim.x * 4) {
        const int head = (int)(it % heads);
        const long long wi = it / heads;
        const int win = (int)(wi % (X * Y)), a = (int)(wi / (X * Y));
        const int wx = win / Y, wy = win - wx * Y;
        const size_t tok_t = ((size_t)a * H + wx * 4 + w1q) * W + wy * 4 + w2q;
        size_t tok_hr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) tok_hr[r] = ((size_t)a * H + wx * 4 + h) * W + wy * 4 + r;
        const float* base = qkv + coff + head * DHD;
        f32x4v q[NC], kf[NC], vf[NC], g[NC];
        float Dq = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float* row = base + tok_t * ctot + 16 * c + 4 * h;
            q[c] = *reinterpret_cast<const f32x4v*>(row);
            q[c] *= scale;
            kf[c] = *reinterpret_cast<const f32x4v*>(row + inner);
            vf[c] = *reinterpret_cast<const f32x4v*>(row + 2 * inner);
            g[c] = *reinterpret_cast<const f32x4v*>(dout + tok_t * inner + head * DHD + 16 * c + 4 * h);
            const f32x4v o = *reinterpret_cast<const f32x4v*>(out + tok_t * inner + head * DHD + 16 * c + 4 * h);
            Dq = fmaf(g[c].x, o.x, Dq); Dq = fmaf(g[c].y, o.y, Dq); Dq = fmaf(g[c].z, o.z, Dq); Dq = fmaf(g[c].w, o.w, Dq);
        }
        Dq += __shfl_xor(Dq, 16);
        Dq += __shfl_xor(Dq, 32);
        // ---- row = key (h, r), column = query t
        f32x4v st = {0.f, 0.f, 0.f, 0.f}, dpT = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[c].x, q[c].x, st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[c].y, q[c].y, st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[c].z, q[c].z, st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[c].w, q[c].w, st, 0, 0, 0);
            dpT = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[c].x, g[c].x, dpT, 0, 0, 0);
            dpT = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[c].y, g[c].y, dpT, 0, 0, 0);
            dpT = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[c].z, g[c].z, dpT, 0, 0, 0);
            dpT = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[c].w, g[c].w, dpT, 0, 0, 0);
        }
        const int iT = (h - w1q + 3) * 7 + (3 - w2q);            // + r: pos index of key (h, r) against query (w1q, w2q)
        float m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) { st[r] += lpos[iT + r]; m = fmaxf(m, st[r]); }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        const float mb = m * kLog2e;
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { st[r] = __builtin_amdgcn_exp2f(fmaf(st[r], kLog2e, -mb)); sum += st[r]; }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        f32x4v dq[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) dq[c] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ds = st[r] * inv * (dpT[r] - Dq);
            if (ds != 0.f) atomicAdd(&ldpos[wave][iT + r], (unsigned long long)__float2ll_rn(ds * kFixF));
            const float* kb = base + tok_hr[r] * ctot + inner + t;
#pragma unroll
            for (int c = 0; c < NC; ++c) dq[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds, kb[16 * c], dq[c], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* dst = dqkv + tok_hr[r] * ctot + coff + head * DHD + t;
#pragma unroll
            for (int c = 0; c < NC; ++c) dst[16 * c] = dq[c][r] * scale;
        }
        // ---- row = query (h, r), column = key t
        f32x4v s2 = {0.f, 0.f, 0.f, 0.f}, d2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            s2 = __builtin_amdgcn_mfma_f32_16x16x4f32(q[c].x, kf[c].x, s2, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_f32_16x16x4f32(q[c].y, kf[c].y, s2, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_f32_16x16x4f32(q[c].z, kf[c].z, s2, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_f32_16x16x4f32(q[c].w, kf[c].w, s2, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(g[c].x, vf[c].x, d2, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(g[c].y, vf[c].y, d2, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(g[c].z, vf[c].z, d2, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(g[c].w, vf[c].w, d2, 0, 0, 0);
        }
        const int i2 = (w1q - h + 3) * 7 + (w2q + 3);              // - r: pos index of key (w1q, w2q) against query (h, r)
        f32x4v dk[NC], dv[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) { dk[c] = (f32x4v){0.f, 0.f, 0.f, 0.f}; dv[c] = (f32x4v){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float m2 = __shfl(mb, 4 * h + r), inv2 = __shfl(inv, 4 * h + r), D2 = __shfl(Dq, 4 * h + r);
            const float pr = __builtin_amdgcn_exp2f(fmaf(s2[r] + lpos[i2 - r], kLog2e, -m2)) * inv2;
            const float ds = pr * (d2[r] - D2);
            const float* qq = base + tok_hr[r] * ctot + t;
            const float* gg = dout + tok_hr[r] * inner + head * DHD + t;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                dk[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds, qq[16 * c] * scale, dk[c], 0, 0, 0);
                dv[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pr, gg[16 * c], dv[c], 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* dst = dqkv + tok_hr[r] * ctot + coff + head * DHD + t;
#pragma unroll
            for (int c = 0; c < NC; ++c) { dst[inner + 16 * c] = dk[c][r]; dst[2 * inner + 16 * c] = dv[c][r]; }
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    if (lane < 49 && ldpos[wave][lane]) atomicAdd(dpos + lane, ldpos[wave][lane]);
}

__global__ __launch_bounds__(256) void fixed_to_float_kernel(const long long* __restrict__ acc, float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (float)((double)acc[i] * (1.0 / 4294967296.0));
}

// ------------------------------------------------------------------------------------------------------------ split attention
constexpr int SPLIT_CHUNKS = 128;

// partial[a][chunk][r][c] = sum over the chunk's pixels of dout * s_r; grid (C / 64, n, SPLIT_CHUNKS), block 256 = 64 channels x 4 pixel lanes
__global__ __launch_bounds__(256) void split_sums_kernel(const float* __restrict__ s0, const float* __restrict__ s1, const float* __restrict__ s2,
                                                         const float* __restrict__ dout, float* __restrict__ partial, int hw, int C) {
    __shared__ float part[3][4][64];
    const int cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, g = threadIdx.x >> 6, a = blockIdx.y, ch = blockIdx.z;
    const int per = (hw + SPLIT_CHUNKS - 1) / SPLIT_CHUNKS;
    const int p0 = ch * per, p1 = min(hw, p0 + per);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int p = p0 + g; p < p1; p += 4) {
        const size_t o = ((size_t)a * hw + p) * C + c;
        const float d = dout[o];
        a0 = fmaf(d, s0[o], a0); a1 = fmaf(d, s1[o], a1); a2 = fmaf(d, s2[o], a2);
    }
    part[0][g][cl] = a0; part[1][g][cl] = a1; part[2][g][cl] = a2;
    __syncthreads();
    if (g < 3)
        partial[(((size_t)a * SPLIT_CHUNKS + ch) * 3 + g) * C + c] = ((part[g][0][cl] + part[g][1][cl]) + part[g][2][cl]) + part[g][3][cl];
}

__global__ void split_sums_finish(const float* __restrict__ partial, float* __restrict__ da, int C) {   // da (n, 3, C)
    const int c = blockIdx.x * 64 + threadIdx.x, r = blockIdx.y, a = blockIdx.z;
    float acc = 0.f;
    for (int ch = 0; ch < SPLIT_CHUNKS; ++ch) acc += partial[(((size_t)a * SPLIT_CHUNKS + ch) * 3 + r) * C + c];
    da[((size_t)a * 3 + r) * C + c] = acc;
}

// ds_r = w[a][r][c] * dout + dgap[a][c] * inv_hw
__global__ void split_backward_kernel(const float4* __restrict__ dout, const float* __restrict__ w, const float* __restrict__ dgap,
                                      float4* __restrict__ d0, float4* __restrict__ d1, float4* __restrict__ d2, size_t n4_per_agent, int C,
                                      float inv_hw) {
    const int a = blockIdx.y;
    const int c4 = C / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4_per_agent; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4) * 4;
        const size_t o = (size_t)a * n4_per_agent + i;
        const float4 d = dout[o];
        const float* wa = w + (size_t)a * 3 * C;
        const float* ga = dgap + (size_t)a * C + c;
        const float4 gg = make_float4(ga[0] * inv_hw, ga[1] * inv_hw, ga[2] * inv_hw, ga[3] * inv_hw);
        d0[o] = make_float4(fmaf(wa[c], d.x, gg.x), fmaf(wa[c + 1], d.y, gg.y), fmaf(wa[c + 2], d.z, gg.z), fmaf(wa[c + 3], d.w, gg.w));
        d1[o] = make_float4(fmaf(wa[C + c], d.x, gg.x), fmaf(wa[C + c + 1], d.y, gg.y), fmaf(wa[C + c + 2], d.z, gg.z), fmaf(wa[C + c + 3], d.w, gg.w));
        d2[o] = make_float4(fmaf(wa[2 * C + c], d.x, gg.x), fmaf(wa[2 * C + c + 1], d.y, gg.y), fmaf(wa[2 * C + c + 2], d.z, gg.z),
                            fmaf(wa[2 * C + c + 3], d.w, gg.w));
    }
}

// ------------------------------------------------------------------------------------------------------------ warp, adjoint
__device__ __forceinline__ float lin_m1_1(int i, int n) {     // as v2xvit.hip
    if (n <= 1) return -1.0f;
    const float step = 2.0f / (float)(n - 1);
    return (i < n / 2) ? (-1.0f + step * (float)i) : (1.0f - step * (float)(n - 1 - i));
}

// The warp is affine in the pixel indices: (ix, iy)(i, j) = (Ax j + Bx i + Cx, Ay j + By i + Cy).  A source pixel (sy, sx) receives from the
// output pixels with |ix - sx| < 1 and |iy - sy| < 1: a parallelogram around M^-1 (s - C) whose bounding box has half-widths
// (|By| + |Bx|) / |det| columns and (|Ay| + |Ax|) / |det| rows.  `small`: that box (with a safety margin) holds at most 64 candidates -- every
// rigid transform of a V2X scene; the adjoint is then a GATHER (no atomics, no 8-byte workspace traffic).  Anything else (a degenerate
// theta) keeps the fixed-point scatter below.  The predicate is a pure function of theta, evaluated identically by the three kernels.
struct WarpBox { float Ax, Bx, Ay, By, Cx, Cy, inv_det, ej, ei; bool small; };

template <bool AC>
__device__ __forceinline__ void warp_src(const float* __restrict__ th, int i, int j, int H, int W, float& ix, float& iy) {
    float xn = lin_m1_1(j, W), yn = lin_m1_1(i, H);
    if (!AC) { xn = (xn * (float)(W - 1)) / (float)W; yn = (yn * (float)(H - 1)) / (float)H; }
    const float gx = th[0] * xn + th[1] * yn + th[2];
    const float gy = th[3] * xn + th[4] * yn + th[5];
    ix = AC ? ((gx + 1.f) * 0.5f) * (float)(W - 1) : ((gx + 1.f) * (float)W - 1.f) * 0.5f;
    iy = AC ? ((gy + 1.f) * 0.5f) * (float)(H - 1) : ((gy + 1.f) * (float)H - 1.f) * 0.5f;
}

template <bool AC>
__device__ __forceinline__ WarpBox warp_box(const float* __restrict__ th, int H, int W) {
    WarpBox b;
    const float sx = AC ? (float)(W - 1) : (float)W, sy = AC ? (float)(H - 1) : (float)H;
    b.Ax = th[0]; b.Bx = th[1] * sx / sy; b.Ay = th[3] * sy / sx; b.By = th[4];
    warp_src<AC>(th, 0, 0, H, W, b.Cx, b.Cy);
    const float det = b.Ax * b.By - b.Bx * b.Ay;
    b.inv_det = 1.0f / det;
    const float m = 0.05f + 1e-5f * (float)(H + W);          // rounding of the centre over a map of this size, generously
    b.ej = (fabsf(b.By) + fabsf(b.Bx)) * fabsf(b.inv_det) + m;
    b.ei = (fabsf(b.Ay) + fabsf(b.Ax)) * fabsf(b.inv_det) + m;
    const float cand = (2.f * b.ej + 2.f) * (2.f * b.ei + 2.f);
    b.small = (H > 1 && W > 1) && fabsf(det) > 1e-6f && cand <= 64.f && fabsf(b.Cx) < 1e6f && fabsf(b.Cy) < 1e6f;   // false for NaN / inf too
    return b;
}

// gather: thread -> (source pixel, channel quad), as the scatter kernel's (output pixel, channel quad).  Candidates in row-major order, the
// sampling position and the weights of each recomputed with the forward's expressions: float sums in a fixed order -- bit-reproducible.
template <int CK, bool AC>
__global__ __launch_bounds__(256) void warp_affine_backward_gather_kernel(const float* __restrict__ ddst, const float* __restrict__ theta,
                                                                          float* __restrict__ dsrc, unsigned long long* __restrict__ acc, int H, int W) {
    const int t = threadIdx.x & 15;
    const int pix = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int n = blockIdx.y;
    if (pix >= H * W) return;
    const float* th = theta + n * 6;
    const WarpBox b = warp_box<AC>(th, H, W);
    constexpr int C = 64 * CK;
    if (!b.small) {        // this image goes through the scatter kernel: zero the pixel's fixed-point accumulators
        unsigned long long* a = acc + ((size_t)n * H * W + pix) * C + 4 * t;
#pragma unroll
        for (int k = 0; k < CK; ++k) { a[64 * k] = 0ull; a[64 * k + 1] = 0ull; a[64 * k + 2] = 0ull; a[64 * k + 3] = 0ull; }
        return;
    }
    const int sy = pix / W, sx = pix - sy * W;
    const float ux = (float)sx - b.Cx, uy = (float)sy - b.Cy;
    const float jc = (b.By * ux - b.Bx * uy) * b.inv_det, ic = (b.Ax * uy - b.Ay * ux) * b.inv_det;
    const int j0 = max(0, (int)ceilf(jc - b.ej)), j1 = min(W - 1, (int)floorf(jc + b.ej));
    const int i0 = max(0, (int)ceilf(ic - b.ei)), i1 = min(H - 1, (int)floorf(ic + b.ei));
    float4 s[CK];
#pragma unroll
    for (int k = 0; k < CK; ++k) s[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = i0; i <= i1; ++i)
        for (int j = j0; j <= j1; ++j) {
            float ix, iy;
            warp_src<AC>(th, i, j, H, W, ix, iy);
            const float x0f = floorf(ix), y0f = floorf(iy);
            const int x0 = (int)x0f, y0 = (int)y0f;
            const float wx = x0 == sx ? (x0f + 1.f) - ix : (x0 + 1 == sx ? ix - x0f : 0.f);
            const float wy = y0 == sy ? (y0f + 1.f) - iy : (y0 + 1 == sy ? iy - y0f : 0.f);
            const float w = wx * wy;
            if (w != 0.f) {
                const float* g = ddst + (((size_t)n * H + i) * W + j) * C + 4 * t;
#pragma unroll
                for (int k = 0; k < CK; ++k) {
                    const float4 d = *reinterpret_cast<const float4*>(g + 64 * k);
                    s[k].x = fmaf(d.x, w, s[k].x); s[k].y = fmaf(d.y, w, s[k].y); s[k].z = fmaf(d.z, w, s[k].z); s[k].w = fmaf(d.w, w, s[k].w);
                }
            }
        }
    float* o = dsrc + ((size_t)n * H * W + pix) * C + 4 * t;
#pragma unroll
    for (int k = 0; k < CK; ++k) *reinterpret_cast<float4*>(o + 64 * k) = s[k];
}

// the images the gather kernel left to the scatter: fixed point -> float
template <bool AC>
__global__ __launch_bounds__(256) void warp_fixed_to_float_kernel(const long long* __restrict__ acc, const float* __restrict__ theta, float* __restrict__ out,
                                                                  size_t per_image, int H, int W) {
    const int n = blockIdx.y;
    if (warp_box<AC>(theta + n * 6, H, W).small) return;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < per_image; i += (size_t)gridDim.x * 256)
        out[(size_t)n * per_image + i] = (float)((double)acc[(size_t)n * per_image + i] * (1.0 / 4294967296.0));
}

template <int CK, bool AC>   // C = 64 * CK; AC = align_corners: true for warp_affine, false for warp_affine_simple (as v2xvit.hip)
__global__ __launch_bounds__(256) void warp_affine_backward_kernel(const float* __restrict__ ddst, const float* __restrict__ theta,
                                                                   unsigned long long* __restrict__ acc, int H, int W) {
    const int t = threadIdx.x & 15;
    const int pix = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int n = blockIdx.y;
    if (pix >= H * W) return;
    const int i = pix / W, j = pix - i * W;
    const float* th = theta + n * 6;
    if (warp_box<AC>(th, H, W).small) return;          // done by warp_affine_backward_gather_kernel
    float ix, iy;
    warp_src<AC>(th, i, j, H, W, ix, iy);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - x0f, wx0 = (x0f + 1.f) - ix, wy1 = iy - y0f, wy0 = (y0f + 1.f) - iy;
    const float w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;
    const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)x1 < (unsigned)W;
    const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)y1 < (unsigned)H;
    constexpr int C = 64 * CK;
    const float* g = ddst + ((size_t)n * H * W + pix) * C + 4 * t;
    unsigned long long* base = acc + (size_t)n * H * W * C + 4 * t;
#pragma unroll
    for (int k = 0; k < CK; ++k) {
        const float4 d = *reinterpret_cast<const float4*>(g + 64 * k);
        auto add = [&](bool ok, int yy, int xx, float w) {
            if (ok && w != 0.f) {
                unsigned long long* a = base + ((size_t)yy * W + xx) * C + 64 * k;
                atomicAdd(a + 0, (unsigned long long)__float2ll_rn(d.x * w * kFixF));
                atomicAdd(a + 1, (unsigned long long)__float2ll_rn(d.y * w * kFixF));
                atomicAdd(a + 2, (unsigned long long)__float2ll_rn(d.z * w * kFixF));
                atomicAdd(a + 3, (unsigned long long)__float2ll_rn(d.w * w * kFixF));
            }
        };
        add(vy0 && vx0, y0, x0, w00);
        add(vy0 && vx1, y0, x1, w01);
        add(vy1 && vx0, y1, x0, w10);
        add(vy1 && vx1, y1, x1, w11);
    }
}

}  // namespace

extern "C" int av2x_hgt_attention_backward(const float* proj, const float* mask, const int32_t* types_host, const float* dout, float* dproj,
                                           int32_t n, int32_t hw, int32_t heads, int32_t dim_head, av2x_stream_t stream) {
    if (!proj || !mask || !types_host || !dout || !dproj) return av2x::fail("av2x_hgt_attention_backward: null argument");
    if (heads != 8 || dim_head != 32) return av2x::fail("av2x_hgt_attention_backward: heads=%d dim_head=%d (8 x 32)", heads, dim_head);
    if (n < 1 || n > 32 || hw < 1) return av2x::fail("av2x_hgt_attention_backward: bad sizes");
    HgtBwdParams p;
    p.proj = proj; p.mask = mask; p.dout = dout; p.dproj = dproj; p.n = n; p.hw = hw;
    for (int i = 0; i < n; ++i) p.types[i] = types_host[i];
    p.scale = 1.0f / sqrtf((float)dim_head);
    hipStream_t st = av2x::as_stream(stream);
    static const bool no_reg = [] { const char* e = getenv("AV2X_HGT_BWD_NO_REG"); return e && e[0] == '1'; }();
    if (n <= 8 && !no_reg) {          // every column of dproj is written: no pre-zeroing
        if (n <= 4) hipLaunchKernelGGL(hgt_attention_backward_reg_kernel<4>, dim3((hw + 3) / 4), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(hgt_attention_backward_reg_kernel<8>, dim3((hw + 3) / 4), dim3(256), 0, st, p);
        return av2x::check_launch("hgt_attention_backward_reg_kernel");
    }
    hipError_t e = hipMemsetAsync(dproj, 0, (size_t)n * hw * 1280 * sizeof(float), st);
    if (e != hipSuccess) return av2x::fail("av2x_hgt_attention_backward: memset: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(hgt_attention_backward_kernel, dim3((hw + 3) / 4), dim3(256), 0, st, p);
    return av2x::check_launch("hgt_attention_backward_kernel");
}

extern "C" uint64_t av2x_window_attention_backward_workspace_bytes(int32_t n, int32_t h, int32_t w, int32_t heads, int32_t window) {
    return (uint64_t)n * h * w * heads * 3 * sizeof(float) + (uint64_t)(2 * window - 1) * (2 * window - 1) * 8ull + 64;
}

extern "C" int av2x_window_attention_backward(const float* qkv, int32_t ctot, int32_t coff, const float* pos_embedding, const float* out,
                                              const float* dout, float* dqkv, float* dpos, void* workspace, int32_t n, int32_t h, int32_t w,
                                              int32_t heads, int32_t dim_head, int32_t window, av2x_stream_t stream) {
    if (!qkv || !pos_embedding || !out || !dout || !dqkv || !dpos || !workspace) return av2x::fail("av2x_window_attention_backward: null argument");
    if (h % window || w % window) return av2x::fail("av2x_window_attention_backward: map %dx%d not divisible by window %d", h, w, window);
    const size_t total = (size_t)n * h * w * heads;
    if (total == 0) return 0;
    const int np = (2 * window - 1) * (2 * window - 1);
    unsigned long long* dposfix = reinterpret_cast<unsigned long long*>(workspace);
    float* stats = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + (((size_t)np * 8 + 63) & ~(size_t)63));
    hipStream_t st = av2x::as_stream(stream);
    hipError_t e = hipMemsetAsync(dposfix, 0, (size_t)np * 8, st);
    if (e != hipSuccess) return av2x::fail("av2x_window_attention_backward: memset: %s", hipGetErrorString(e));
    const float scale = 1.0f / sqrtf((float)dim_head);
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
#define AV2X_WAB(DHD, WS)                                                                                                              \
    if (dim_head == DHD && window == WS) {                                                                                             \
        hipLaunchKernelGGL((window_attn_backward_q<DHD, WS>), grid, block, 0, st, qkv, ctot, coff, pos_embedding, out, dout, dqkv, stats, \
                           dposfix, n, h, w, heads, scale);                                                                            \
        hipLaunchKernelGGL((window_attn_backward_kv<DHD, WS>), grid, block, 0, st, qkv, ctot, coff, pos_embedding, dout, stats, dqkv, n, \
                           h, w, heads, scale);                                                                                        \
        hipLaunchKernelGGL(fixed_to_float_kernel, dim3(1), dim3(256), 0, st, reinterpret_cast<const long long*>(dposfix), dpos, (size_t)np); \
        return av2x::check_launch("window_attn_backward");                                                                             \
    }
    static const bool no_wave = [] { const char* e = getenv("AV2X_WIN_BWD_NO_WAVE"); return e && e[0] == '1'; }();
    if (window == 4 && (dim_head == 32 || dim_head == 64) && h % 4 == 0 && w % 4 == 0 && ctot % 4 == 0 && coff % 4 == 0 && !no_wave) {
        const long long items = (long long)n * (h / 4) * (w / 4) * heads;
        const long long wgs = (items + 3) / 4;
        const dim3 g2((unsigned)(wgs < 2048 ? wgs : 2048));
        if (dim_head == 32)
            hipLaunchKernelGGL((window_attn_backward_wave_kernel<32>), g2, block, 0, st, qkv, ctot, coff, pos_embedding, out, dout, dqkv, dposfix, n, h, w, heads, scale);
        else
            hipLaunchKernelGGL((window_attn_backward_wave_kernel<64>), g2, block, 0, st, qkv, ctot, coff, pos_embedding, out, dout, dqkv, dposfix, n, h, w, heads, scale);
        hipLaunchKernelGGL(fixed_to_float_kernel, dim3(1), dim3(256), 0, st, reinterpret_cast<const long long*>(dposfix), dpos, (size_t)np);
        return av2x::check_launch("window_attn_backward_wave_kernel");
    }
    AV2X_WAB(16, 2)
    AV2X_WAB(32, 4)
    AV2X_WAB(64, 4)
#undef AV2X_WAB
    return av2x::fail("av2x_window_attention_backward: (dim_head, window) = (%d, %d) unsupported: (16,2) (32,4) (64,4)", dim_head, window);
}

extern "C" uint64_t av2x_split_attn_backward_workspace_bytes(int32_t n, int32_t c) { return (uint64_t)n * SPLIT_CHUNKS * 3 * c * sizeof(float); }

extern "C" int av2x_split_attn_sums(const float* s0, const float* s1, const float* s2, const float* dout, float* da, float* workspace,
                                    int32_t n, int32_t hw, int32_t c, av2x_stream_t stream) {
    if (!s0 || !s1 || !s2 || !dout || !da || !workspace) return av2x::fail("av2x_split_attn_sums: null argument");
    if (c % 64 || n < 1) return av2x::fail("av2x_split_attn_sums: c must be a multiple of 64");
    hipStream_t st = av2x::as_stream(stream);
    hipLaunchKernelGGL(split_sums_kernel, dim3(c / 64, n, SPLIT_CHUNKS), dim3(256), 0, st, s0, s1, s2, dout, workspace, hw, c);
    hipLaunchKernelGGL(split_sums_finish, dim3(c / 64, 3, n), dim3(64), 0, st, workspace, da, c);
    return av2x::check_launch("split_sums");
}

extern "C" int av2x_split_attn_backward(const float* dout, const float* weights, const float* dgap, float* ds0, float* ds1, float* ds2,
                                        int32_t n, int32_t hw, int32_t c, av2x_stream_t stream) {
    if (!dout || !weights || !dgap || !ds0 || !ds1 || !ds2) return av2x::fail("av2x_split_attn_backward: null argument");
    if (c % 4 || n < 1) return av2x::fail("av2x_split_attn_backward: c must be a multiple of 4");
    const size_t n4 = (size_t)hw * c / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(split_backward_kernel, dim3((unsigned)blocks, n), dim3(256), 0, av2x::as_stream(stream), reinterpret_cast<const float4*>(dout),
                       weights, dgap, reinterpret_cast<float4*>(ds0), reinterpret_cast<float4*>(ds1), reinterpret_cast<float4*>(ds2), n4, c,
                       1.0f / (float)hw);
    return av2x::check_launch("split_backward_kernel");
}

extern "C" uint64_t av2x_warp_affine_backward_workspace_bytes(int32_t n, int32_t h, int32_t w, int32_t c) { return (uint64_t)n * h * w * c * 8ull; }

template <bool AC>
static int warp_affine_backward_launch(const char* who, const float* ddst, const float* theta, float* dsrc, void* workspace, int32_t n, int32_t h, int32_t w,
                                       int32_t c, av2x_stream_t stream) {
    if (!ddst || !theta || !dsrc || !workspace) return av2x::fail("%s: null argument", who);
    if (c != 64 && c != 128 && c != 256) return av2x::fail("%s: c=%d (64, 128, 256)", who, c);
    if (n < 1) return 0;
    hipStream_t st = av2x::as_stream(stream);
    const size_t total = (size_t)n * h * w * c;
    // per image: the gather (every well-conditioned theta), or -- the gather kernel zeroes the image's accumulators instead -- the fixed-point
    // scatter + conversion, which return at once for the gathered images
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(workspace);
    const dim3 grid((h * w + 15) / 16, n), block(256);
    if (c == 64) {
        hipLaunchKernelGGL((warp_affine_backward_gather_kernel<1, AC>), grid, block, 0, st, ddst, theta, dsrc, acc, h, w);
        hipLaunchKernelGGL((warp_affine_backward_kernel<1, AC>), grid, block, 0, st, ddst, theta, acc, h, w);
    } else if (c == 128) {
        hipLaunchKernelGGL((warp_affine_backward_gather_kernel<2, AC>), grid, block, 0, st, ddst, theta, dsrc, acc, h, w);
        hipLaunchKernelGGL((warp_affine_backward_kernel<2, AC>), grid, block, 0, st, ddst, theta, acc, h, w);
    } else {
        hipLaunchKernelGGL((warp_affine_backward_gather_kernel<4, AC>), grid, block, 0, st, ddst, theta, dsrc, acc, h, w);
        hipLaunchKernelGGL((warp_affine_backward_kernel<4, AC>), grid, block, 0, st, ddst, theta, acc, h, w);
    }
    const size_t per_image = (size_t)h * w * c;
    const size_t cb = (per_image + 255) / 256;
    hipLaunchKernelGGL((warp_fixed_to_float_kernel<AC>), dim3((unsigned)(cb < 1024 ? cb : 1024), n), dim3(256), 0, st, reinterpret_cast<const long long*>(acc), theta,
                       dsrc, per_image, h, w);
    (void)total;
    return av2x::check_launch("warp_affine_backward kernels");
}

extern "C" int av2x_warp_affine_backward(const float* ddst, const float* theta, float* dsrc, void* workspace, int32_t n, int32_t h, int32_t w,
                                         int32_t c, av2x_stream_t stream) {
    return warp_affine_backward_launch<true>("av2x_warp_affine_backward", ddst, theta, dsrc, workspace, n, h, w, c, stream);
}

extern "C" int av2x_warp_affine_simple_backward(const float* ddst, const float* theta, float* dsrc, void* workspace, int32_t n, int32_t h,
                                                int32_t w, int32_t c, av2x_stream_t stream) {
    return warp_affine_backward_launch<false>("av2x_warp_affine_simple_backward", ddst, theta, dsrc, workspace, n, h, w, c, stream);
}
