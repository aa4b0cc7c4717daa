// Backward kernels of the transformer-style fusion heads (`.train()` of Airv2xCoBEVT; the reference trains them through
// torch autograd, tools/train.py:220-247, over models/cobevt_modules/swap_fusion_modules.py:78-195 and base_transformer.py:6-38).
//
//   layernorm_backward_kernel   nn.LayerNorm: dx per token (one wave per token, statistics recomputed from x), dgamma / dbeta as
//                               per-workgroup partial rows reduced in a fixed order by av2x_channel_sum            HBM-bound
//   gelu_kernel / gelu_backward_kernel   exact GELU (nn.GELU()) and its derivative on the stored pre-activation     HBM-bound
//   fax_attention_backward_kernel        Attention.forward :78-127 differentiated: per (window, head) the scores are RECOMPUTED
//                               from q, k (nothing but the forward's output is kept), phase A (a lane per query token) gives the
//                               row statistics, D_i = dO_i . O_i and dq_i; phase B (a lane per key token) gives dk_j, dv_j; the
//                               relative-position-bias gradient is summed as 2^-32 fixed point (LDS, then one 64-bit global atomic
//                               per table entry and wave): integer sums, bit-reproducible whatever the order
//   scale_broadcast_kernel      backward of the mean over the agent axis (mlp_head's Reduce, :270)
//   dropout_kernel              x * mask * scale (forward and backward of nn.Dropout with a caller-supplied Bernoulli mask)
#include <cstdlib>

#include "av2x_common.hpp"

namespace {

constexpr int DH = 32;
constexpr float kFixF = 4294967296.0f;   // 2^32

// ------------------------------------------------------------------------------------------------------------ LayerNorm
// block = 4 waves; every wave walks tokens blockIdx.x * TPB + wave, + 4, ...; C = 256 * CK
template <int CK>
__global__ __launch_bounds__(256) void layernorm_backward_kernel(const float4* __restrict__ x, const float4* __restrict__ gamma,
                                                                 const float4* __restrict__ dy, float4* __restrict__ dx,
                                                                 float* __restrict__ partial, long long n_tokens, int tpb, float eps) {
    constexpr int C = 256 * CK;
    __shared__ float4 red[2][4][64 * CK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 g[CK], dg[CK], db[CK];
#pragma unroll
    for (int k = 0; k < CK; ++k) {
        g[k] = gamma[k * 64 + lane];
        dg[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        db[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const long long t0 = (long long)blockIdx.x * tpb;
    for (int i = wave; i < tpb; i += 4) {
        const long long tok = t0 + i;
        if (tok >= n_tokens) break;
        float4 v[CK], d[CK];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < CK; ++k) {
            v[k] = x[(size_t)tok * (C / 4) + k * 64 + lane];
            d[k] = dy[(size_t)tok * (C / 4) + k * 64 + lane];
            s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < CK; ++k) {
            v[k].x -= mean; v[k].y -= mean; v[k].z -= mean; v[k].w -= mean;
            q += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = 1.0f / sqrtf(q / (float)C + eps);
        float sa = 0.f, sb = 0.f;     // sum(a), sum(a * xhat), a = dy * gamma
#pragma unroll
        for (int k = 0; k < CK; ++k) {
            v[k].x *= rstd; v[k].y *= rstd; v[k].z *= rstd; v[k].w *= rstd;          // xhat
            dg[k].x += d[k].x * v[k].x; dg[k].y += d[k].y * v[k].y; dg[k].z += d[k].z * v[k].z; dg[k].w += d[k].w * v[k].w;
            db[k].x += d[k].x; db[k].y += d[k].y; db[k].z += d[k].z; db[k].w += d[k].w;
            d[k].x *= g[k].x; d[k].y *= g[k].y; d[k].z *= g[k].z; d[k].w *= g[k].w;  // a
            sa += (d[k].x + d[k].y) + (d[k].z + d[k].w);
            sb += (d[k].x * v[k].x + d[k].y * v[k].y) + (d[k].z * v[k].z + d[k].w * v[k].w);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o); }
        const float ma = sa / (float)C, mb = sb / (float)C;
#pragma unroll
        for (int k = 0; k < CK; ++k) {
            float4 r;
            r.x = rstd * (d[k].x - ma - v[k].x * mb);
            r.y = rstd * (d[k].y - ma - v[k].y * mb);
            r.z = rstd * (d[k].z - ma - v[k].z * mb);
            r.w = rstd * (d[k].w - ma - v[k].w * mb);
            dx[(size_t)tok * (C / 4) + k * 64 + lane] = r;
        }
    }
#pragma unroll
    for (int k = 0; k < CK; ++k) { red[0][wave][k * 64 + lane] = dg[k]; red[1][wave][k * 64 + lane] = db[k]; }
    __syncthreads();
    // waves 0 / 1 add the four wave partials of dgamma / dbeta in wave order and write this workgroup's partial row
    if (wave < 2) {
#pragma unroll
        for (int k = 0; k < CK; ++k) {
            float4 a = red[wave][0][k * 64 + lane];
#pragma unroll
            for (int u = 1; u < 4; ++u) {
                const float4 b = red[wave][u][k * 64 + lane];
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
            *reinterpret_cast<float4*>(partial + ((size_t)(wave * gridDim.x + blockIdx.x)) * C + (k * 64 + lane) * 4) = a;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------ GELU
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_df(float v) {
    return 0.5f * (1.0f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * expf(-0.5f * v * v);
}

__global__ __launch_bounds__(256) void gelu_kernel(const float4* __restrict__ z, const float4* __restrict__ dy, float4* __restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = z[i];
        float4 r;
        if (dy) {
            const float4 d = dy[i];
            r = make_float4(d.x * gelu_df(v.x), d.y * gelu_df(v.y), d.z * gelu_df(v.z), d.w * gelu_df(v.w));
        } else {
            r = make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
        }
        out[i] = r;
    }
}

__global__ __launch_bounds__(256) void scale_broadcast_kernel(const float4* __restrict__ dy, float4* __restrict__ dx, int n_agents, size_t per4, float scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < per4; i += (size_t)gridDim.x * 256) {
        float4 v = dy[i];
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        for (int l = 0; l < n_agents; ++l) dx[(size_t)l * per4 + i] = v;
    }
}

__global__ __launch_bounds__(256) void dropout_kernel(const float4* __restrict__ x, const unsigned* __restrict__ mask4, float4* __restrict__ y, size_t n4, float scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = x[i];
        const unsigned m = mask4[i];       // four 0 / 1 bytes
        y[i] = make_float4((m & 0xffu) ? v.x * scale : 0.f, (m & 0xff00u) ? v.y * scale : 0.f, (m & 0xff0000u) ? v.z * scale : 0.f,
                           (m & 0xff000000u) ? v.w * scale : 0.f);
    }
}

// nn.Dropout with the Bernoulli draw in the kernel: Philox4x32-10 keyed by a 64-bit seed the caller draws from torch's generator, counter =
// the float4's index -- forward and backward regenerate the same keep-mask from the seed, so no mask tensor exists (three torch launches and
// 1 + 4 + 1 bytes per element of HBM traffic per site before).  Element e of quad i is kept iff (r_e >> 8) * 2^-24 >= p.
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

__global__ __launch_bounds__(256) void dropout_seeded_kernel(const float4* __restrict__ x, const float4* __restrict__ res, float4* __restrict__ y, size_t n4,
                                                             float p, float scale, unsigned k0, unsigned k1) {
    const float k24 = 1.0f / 16777216.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const uint4 r = philox4x32_10(make_uint4((unsigned)i, (unsigned)(i >> 32), 0u, 0u), k0, k1);
        const float4 v = x[i];
        float4 o = make_float4((float)(r.x >> 8) * k24 >= p ? v.x * scale : 0.f, (float)(r.y >> 8) * k24 >= p ? v.y * scale : 0.f,
                               (float)(r.z >> 8) * k24 >= p ? v.z * scale : 0.f, (float)(r.w >> 8) * k24 >= p ? v.w * scale : 0.f);
        if (res) { const float4 a = res[i]; o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w; }      // the residual the block adds after its dropout
        y[i] = o;
    }
}

// ------------------------------------------------------------------------------------------- fused axial attention, backward
struct FaxBwdParams {
    const float* qkv;    // (L*H*W, 3C)
    const float* table;  // (tab_n, heads)
    const float* out;    // (L*H*W, C) forward output (heads merged)
    const float* dout;   // (L*H*W, C)
    float* dqkv;         // (L*H*W, 3C)
    long long* dtable;   // (tab_n, heads) 2^-32 fixed point, pre-zeroed
    int L, n_valid, H, W, ws, heads, grid;
    float scale;
};

// workgroup = one window, 2 waves; wave w handles heads w, w + 2, ...  LDS per wave: K | V | Q (pre-scaled) | dO, each [T][32], then
// m | l | D [T] each, then the head's bias column [tab_n] and its fixed-point gradient [tab_n] (64-bit)
__global__ __launch_bounds__(128) void fax_attention_backward_kernel(const FaxBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ws = p.ws, ws2 = ws * ws;
    const int T = p.L * ws2, Tk = p.n_valid * ws2;
    const int X = p.H / ws, Y = p.W / ws;
    const int wx = blockIdx.x / Y, wy = blockIdx.x % Y;
    const int C = p.heads * DH, C3 = 3 * C;
    const int s1 = 2 * ws - 1;
    const int tab_n = (2 * p.L - 1) * s1 * s1;
    const int tab_p = (tab_n + 3) & ~3;
    const size_t per_wave = (size_t)4 * T * DH + 3 * T + tab_p + 2 * (size_t)tab_p;
    float* Kl = lds + wave * per_wave;
    float* Vl = Kl + T * DH;
    float* Ql = Vl + T * DH;
    float* Dl = Ql + T * DH;                 // dO
    float* ml = Dl + T * DH;
    float* ll = ml + T;
    float* dl = ll + T;                      // D_i
    float* tab = dl + T;
    unsigned long long* dtab = reinterpret_cast<unsigned long long*>(tab + tab_p);

    auto token_row = [&](int t) -> int {
        const int l = t / ws2, r = t - l * ws2, w1 = r / ws, w2 = r - w1 * ws;
        const int ph = p.grid ? (w1 * X + wx) : (wx * ws + w1);
        const int pw = p.grid ? (w2 * Y + wy) : (wy * ws + w2);
        return (l * p.H + ph) * p.W + pw;
    };
    auto rel_i = [&](int t) -> int {   // the query-side part of the relative-position index (:53-75)
        const int li = t / ws2, ri = t - li * ws2, hi = ri / ws, wi = ri - hi * ws;
        return ((li + p.L - 1) * s1 + (hi + ws - 1)) * s1 + (wi + ws - 1);
    };
    auto rel_j = [&](int t) -> int {
        const int lj = t / ws2, rj = t - lj * ws2, hj = rj / ws, wj = rj - hj * ws;
        return (lj * s1 + hj) * s1 + wj;
    };

    for (int h = wave; h < p.heads; h += 2) {
        // ---- stage K, V (valid agents), Q * scale, dO (all agents), the head's bias column; zero its gradient accumulator
        for (int j0 = 0; j0 < T; j0 += 8) {
            const int j = j0 + (lane >> 3), d4 = lane & 7;
            if (j < T) {
                const int row = token_row(j);
                const float* src = p.qkv + (size_t)row * C3 + h * DH + d4 * 4;
                float4 q = *reinterpret_cast<const float4*>(src);
                q.x *= p.scale; q.y *= p.scale; q.z *= p.scale; q.w *= p.scale;
                *reinterpret_cast<float4*>(Ql + j * DH + d4 * 4) = q;
                *reinterpret_cast<float4*>(Dl + j * DH + d4 * 4) = *reinterpret_cast<const float4*>(p.dout + (size_t)row * C + h * DH + d4 * 4);
                if (j < Tk) {
                    *reinterpret_cast<float4*>(Kl + j * DH + d4 * 4) = *reinterpret_cast<const float4*>(src + C);
                    *reinterpret_cast<float4*>(Vl + j * DH + d4 * 4) = *reinterpret_cast<const float4*>(src + 2 * C);
                }
            }
        }
        for (int i = lane; i < tab_n; i += 64) { tab[i] = p.table[(size_t)i * p.heads + h]; dtab[i] = 0ull; }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);

        // ---- phase A: a lane per query token i: row max / sum, D_i = dO_i . O_i, dq_i, and the bias-table gradient
        for (int r0 = 0; r0 < T; r0 += 64) {
            const int t = r0 + lane;
            const bool act = t < T;
            const int tt = act ? t : 0;
            const int row = token_row(tt);
            float q[DH], g[DH], dq[DH];
            float Di = 0.f;
            {
                const float* osrc = p.out + (size_t)row * C + h * DH;
#pragma unroll
                for (int d = 0; d < DH; ++d) {
                    q[d] = Ql[tt * DH + d];
                    g[d] = Dl[tt * DH + d];
                    dq[d] = 0.f;
                    Di = fmaf(g[d], osrc[d], Di);
                }
            }
            const int base_i = rel_i(tt);
            float m = -INFINITY, lsum = 0.f;
            for (int j = 0; j < Tk; ++j) {
                const float* kj = Kl + j * DH;
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < DH; ++d) s = fmaf(q[d], kj[d], s);
                s += tab[base_i - rel_j(j)];
                const float mn = fmaxf(m, s);
                lsum = lsum * expf(m - mn) + expf(s - mn);
                m = mn;
            }
            const float inv = 1.0f / lsum;
            for (int j = 0; j < Tk; ++j) {
                const float* kj = Kl + j * DH;
                const float* vj = Vl + j * DH;
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < DH; ++d) { s = fmaf(q[d], kj[d], s); dp = fmaf(g[d], vj[d], dp); }
                const int ti = base_i - rel_j(j);
                s += tab[ti];
                const float pij = expf(s - m) * inv;
                const float ds = pij * (dp - Di);
#pragma unroll
                for (int d = 0; d < DH; ++d) dq[d] = fmaf(ds, kj[d], dq[d]);
                if (act && ds != 0.f) atomicAdd(dtab + ti, (unsigned long long)__float2ll_rn(ds * kFixF));
            }
            if (act) {
                ml[t] = m; ll[t] = inv; dl[t] = Di;
                float* dst = p.dqkv + (size_t)row * C3 + h * DH;
#pragma unroll
                for (int d = 0; d < DH; d += 4)
                    *reinterpret_cast<float4*>(dst + d) = make_float4(dq[d] * p.scale, dq[d + 1] * p.scale, dq[d + 2] * p.scale, dq[d + 3] * p.scale);
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);

        // ---- phase B: a lane per key token j: dk_j = sum_i ds_ij q_i (q pre-scaled), dv_j = sum_i p_ij dO_i
        for (int r0 = 0; r0 < T; r0 += 64) {
            const int j = r0 + lane;
            if (j >= T) break;
            const int row = token_row(j);
            float* dst = p.dqkv + (size_t)row * C3 + h * DH;
            if (j >= Tk) {   // padded agents are never keys (:103-108): zero gradient for their k, v
#pragma unroll
                for (int d = 0; d < DH; d += 4) {
                    *reinterpret_cast<float4*>(dst + C + d) = make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(dst + 2 * C + d) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                continue;
            }
            float k[DH], v[DH], dk[DH], dv[DH];
#pragma unroll
            for (int d = 0; d < DH; ++d) { k[d] = Kl[j * DH + d]; v[d] = Vl[j * DH + d]; dk[d] = 0.f; dv[d] = 0.f; }
            const int off_j = rel_j(j);
            for (int i = 0; i < T; ++i) {
                const float* qi = Ql + i * DH;
                const float* gi = Dl + i * DH;
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < DH; ++d) { s = fmaf(qi[d], k[d], s); dp = fmaf(gi[d], v[d], dp); }
                s += tab[rel_i(i) - off_j];
                const float pij = expf(s - ml[i]) * ll[i];
                const float ds = pij * (dp - dl[i]);
#pragma unroll
                for (int d = 0; d < DH; ++d) { dk[d] = fmaf(ds, qi[d], dk[d]); dv[d] = fmaf(pij, gi[d], dv[d]); }
            }
#pragma unroll
            for (int d = 0; d < DH; d += 4) {
                *reinterpret_cast<float4*>(dst + C + d) = make_float4(dk[d], dk[d + 1], dk[d + 2], dk[d + 3]);
                *reinterpret_cast<float4*>(dst + 2 * C + d) = make_float4(dv[d], dv[d + 1], dv[d + 2], dv[d + 3]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // ---- this (window, head)'s bias-table gradient into the global fixed-point table
        for (int i = lane; i < tab_n; i += 64) {
            const unsigned long long v = dtab[i];
            if (v) atomicAdd(reinterpret_cast<unsigned long long*>(p.dtable) + (size_t)i * p.heads + h, v);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------- ws = 4, one WAVE per (window, head), everything in registers
// The backward of fax_attention_wave_kernel (transformer.hip) in the same register tiling: v_mfma_f32_16x16x4_f32 tiles = one agent's 16
// tokens; lane (t = lane & 15, h = lane >> 4) holds X[token t][d = 4h .. 4h+3, 16+4h .. 16+4h+3] of K, V, Q, dO -- one register set that
// serves as the A operand (row = t) and as the B operand (column = t) alike.  Per query agent qt:
//   S^T[key 4h+r][query t] = K (Q scale)^T + bias, softmax over (kt, r) and the lanes xor 16 / 32       -> P^T     (as the forward)
//   dP^T = V dO^T, D = dO . O,  dS^T = P^T (dP^T - D)   -> the bias-table gradient (2^-32 fixed point, LDS atomics: integer sums)
//   dQ = dS K: dS^T in the accumulator layout IS the A operand (row = query t, k = key (h, r)); B = K rows of token (h, r)
//   the same products with the operands swapped give S, dP with row = query 4h+r, column = key t (statistics of query 4h+r by ds_bpermute):
//   dK += dS^T(A) Q(B), dV += P^T(A) dO(B) accumulate over qt in registers.
// Nothing is staged in LDS but the bias column and its gradient, which a wave keeps across all the windows it walks (persistent grid): one
// 64-bit global atomic per table entry, head and WAVE at the end instead of per window.  The 128-thread kernel above (a lane per token,
// K / V / Q / dO broadcast from LDS, one wave per SIMD) took 12.4 ms per launch at the BASELINE grid; this one is matrix-core work.
typedef float f32x4w __attribute__((ext_vector_type(4)));

template <int NV>
__global__ __launch_bounds__(256) void fax_attention_backward_wave_kernel(const FaxBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = lane & 15, h = lane >> 4;
    const int X = p.H / 4, Y = p.W / 4, nwin = X * Y;
    const int C = p.heads * DH, C3 = 3 * C;
    const int L = p.L, nv = p.n_valid;
    const int tab_n = (2 * L - 1) * 49, tab_s = (tab_n + 63) & ~63;
    const int hpw = (p.heads + 3) >> 2;                     // heads per wave
    float* wl = lds + (size_t)wave * hpw * tab_s * 3;       // per head: bias column [tab_s] | its gradient [tab_s] x 64 bit
    for (int k = 0; k < hpw; ++k) {
        const int head = wave + 4 * k;
        if (head < p.heads) {
            float* tab = wl + (size_t)k * tab_s * 3;
            unsigned long long* dtab = reinterpret_cast<unsigned long long*>(tab + tab_s);
            for (int i = lane; i < tab_s; i += 64) { tab[i] = i < tab_n ? p.table[(size_t)i * p.heads + head] : 0.f; dtab[i] = 0ull; }
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    const int w1q = t >> 2, w2q = t & 3;
    const size_t HW = (size_t)p.H * p.W;
    const float kLog2e = 1.4426950408889634f;
    for (int win = blockIdx.x; win < nwin; win += gridDim.x) {
        const int wx = win / Y, wy = win - wx * Y;
        const int ph_q = p.grid ? (w1q * X + wx) : (wx * 4 + w1q), pw_q = p.grid ? (w2q * Y + wy) : (wy * 4 + w2q);
        const size_t pix_t = (size_t)ph_q * p.W + pw_q;      // token t = (w1q, w2q): A rows / B columns
        const int ph_h = p.grid ? (h * X + wx) : (wx * 4 + h);
        size_t pix_hr[4];                                     // token (w1 = h, w2 = r): B rows of the second products, output rows
#pragma unroll
        for (int r = 0; r < 4; ++r) pix_hr[r] = (size_t)ph_h * p.W + (p.grid ? (r * Y + wy) : (wy * 4 + r));
        for (int k = 0; k < hpw; ++k) {
            const int head = wave + 4 * k;
            if (head >= p.heads) break;
            const float* tab = wl + (size_t)k * tab_s * 3;
            unsigned long long* dtab = reinterpret_cast<unsigned long long*>(const_cast<float*>(tab) + tab_s);
            const float* base = p.qkv + head * DH;
            f32x4w kf[NV][2], vf[NV][2], dk[NV][2], dv[NV][2];
#pragma unroll
            for (int kt = 0; kt < NV; ++kt) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    dk[kt][c] = (f32x4w){0.f, 0.f, 0.f, 0.f};
                    dv[kt][c] = (f32x4w){0.f, 0.f, 0.f, 0.f};
                }
                if (kt < nv) {
                    const float* kr = base + ((size_t)kt * HW + pix_t) * C3 + C + 4 * h;
                    kf[kt][0] = *reinterpret_cast<const f32x4w*>(kr);
                    kf[kt][1] = *reinterpret_cast<const f32x4w*>(kr + 16);
                    vf[kt][0] = *reinterpret_cast<const f32x4w*>(kr + C);
                    vf[kt][1] = *reinterpret_cast<const f32x4w*>(kr + C + 16);
                }
            }
            for (int qt = 0; qt < L; ++qt) {
                const size_t row_t = (size_t)qt * HW + pix_t;
                const float* qr = base + row_t * C3 + 4 * h;
                f32x4w q[2] = {*reinterpret_cast<const f32x4w*>(qr), *reinterpret_cast<const f32x4w*>(qr + 16)};
                q[0] *= p.scale; q[1] *= p.scale;
                const float* gr = p.dout + row_t * C + head * DH + 4 * h;
                const f32x4w g[2] = {*reinterpret_cast<const f32x4w*>(gr), *reinterpret_cast<const f32x4w*>(gr + 16)};
                float Dq;
                {
                    const float* orow = p.out + row_t * C + head * DH + 4 * h;
                    const f32x4w o0 = *reinterpret_cast<const f32x4w*>(orow), o1 = *reinterpret_cast<const f32x4w*>(orow + 16);
                    float a = g[0].x * o0.x;
                    a = fmaf(g[0].y, o0.y, a); a = fmaf(g[0].z, o0.z, a); a = fmaf(g[0].w, o0.w, a);
                    a = fmaf(g[1].x, o1.x, a); a = fmaf(g[1].y, o1.y, a); a = fmaf(g[1].z, o1.z, a); a = fmaf(g[1].w, o1.w, a);
                    a += __shfl_xor(a, 16);
                    a += __shfl_xor(a, 32);
                    Dq = a;                                  // D of query t, in its four h lanes
                }
                // ---- S^T, P^T (row = key 4h + r, column = query t): exactly the forward
                const int cq = ((qt + L - 1) * 7 + (w1q - h + 3)) * 7 + (w2q + 3);
                f32x4w st[NV];
                float m = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < NV; ++kt) {
                    if (kt < nv) {
                        f32x4w a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][c].x, q[c].x, a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][c].y, q[c].y, a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][c].z, q[c].z, a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][c].w, q[c].w, a, 0, 0, 0);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            a[r] += tab[cq - kt * 49 - r];
                            m = fmaxf(m, a[r]);
                        }
                        st[kt] = a;
                    }
                }
                m = fmaxf(m, __shfl_xor(m, 16));
                m = fmaxf(m, __shfl_xor(m, 32));
                const float mb = m * kLog2e;
                float sum = 0.f;
#pragma unroll
                for (int kt = 0; kt < NV; ++kt) {
                    if (kt < nv) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float e = __builtin_amdgcn_exp2f(fmaf(st[kt][r], kLog2e, -mb));
                            st[kt][r] = e;
                            sum += e;
                        }
                    }
                }
                sum += __shfl_xor(sum, 16);
                sum += __shfl_xor(sum, 32);
                const float inv = 1.0f / sum;
                // ---- dP^T = V dO^T, dS^T = P^T (dP^T - D): bias-table gradient, dQ = dS K
                f32x4w dq0 = {0.f, 0.f, 0.f, 0.f}, dq1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < NV; ++kt) {
                    if (kt < nv) {
                        f32x4w a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            a = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kt][c].x, g[c].x, a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kt][c].y, g[c].y, a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kt][c].z, g[c].z, a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kt][c].w, g[c].w, a, 0, 0, 0);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float ds = st[kt][r] * inv * (a[r] - Dq);
                            if (ds != 0.f) atomicAdd(dtab + (cq - kt * 49 - r), (unsigned long long)__float2ll_rn(ds * kFixF));
                            const float* kb = base + ((size_t)kt * HW + pix_hr[r]) * C3 + C + t;     // K of token (h, r): d = t, t + 16
                            dq0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ds, kb[0], dq0, 0, 0, 0);
                            dq1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ds, kb[16], dq1, 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {                // dq rows = query 4h + r = token (h, r), columns d = t, t + 16
                    float* dst = p.dqkv + ((size_t)qt * HW + pix_hr[r]) * C3 + head * DH + t;
                    dst[0] = dq0[r] * p.scale;
                    dst[16] = dq1[r] * p.scale;
                }
                // ---- the other orientation (row = query 4h + r, column = key t): dK += dS^T Q, dV += P^T dO
                float m2[4], inv2[4], D2[4], qb[4][2], gb[4][2];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    m2[r] = __shfl(mb, 4 * h + r);
                    inv2[r] = __shfl(inv, 4 * h + r);
                    D2[r] = __shfl(Dq, 4 * h + r);
                    const size_t row = (size_t)qt * HW + pix_hr[r];
                    const float* qq = base + row * C3 + t;
                    qb[r][0] = qq[0] * p.scale;
                    qb[r][1] = qq[16] * p.scale;
                    const float* gg = p.dout + row * C + head * DH + t;
                    gb[r][0] = gg[0];
                    gb[r][1] = gg[16];
                }
                const int c2 = ((qt + L - 1) * 7 + (h - w1q + 3)) * 7 + (3 - w2q);
#pragma unroll
                for (int kt = 0; kt < NV; ++kt) {
                    if (kt < nv) {
                        f32x4w s2 = {0.f, 0.f, 0.f, 0.f}, d2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            s2 = __builtin_amdgcn_mfma_f32_16x16x4f32(q[c].x, kf[kt][c].x, s2, 0, 0, 0);
                            s2 = __builtin_amdgcn_mfma_f32_16x16x4f32(q[c].y, kf[kt][c].y, s2, 0, 0, 0);
                            s2 = __builtin_amdgcn_mfma_f32_16x16x4f32(q[c].z, kf[kt][c].z, s2, 0, 0, 0);
                            s2 = __builtin_amdgcn_mfma_f32_16x16x4f32(q[c].w, kf[kt][c].w, s2, 0, 0, 0);
                            d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(g[c].x, vf[kt][c].x, d2, 0, 0, 0);
                            d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(g[c].y, vf[kt][c].y, d2, 0, 0, 0);
                            d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(g[c].z, vf[kt][c].z, d2, 0, 0, 0);
                            d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(g[c].w, vf[kt][c].w, d2, 0, 0, 0);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float sv = s2[r] + tab[c2 - kt * 49 + r];
                            const float pr = __builtin_amdgcn_exp2f(fmaf(sv, kLog2e, -m2[r])) * inv2[r];
                            const float ds = pr * (d2[r] - D2[r]);
                            dk[kt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds, qb[r][0], dk[kt][0], 0, 0, 0);
                            dk[kt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds, qb[r][1], dk[kt][1], 0, 0, 0);
                            dv[kt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pr, gb[r][0], dv[kt][0], 0, 0, 0);
                            dv[kt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pr, gb[r][1], dv[kt][1], 0, 0, 0);
                        }
                    }
                }
            }
            // ---- dk, dv rows = key 4h + r = token (h, r) of agent kt; padded agents are never keys (:103-108): zeros
#pragma unroll
            for (int kt = 0; kt < NV; ++kt) {
                if (kt < nv) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* dst = p.dqkv + ((size_t)kt * HW + pix_hr[r]) * C3 + head * DH + t;
                        dst[C] = dk[kt][0][r];
                        dst[C + 16] = dk[kt][1][r];
                        dst[2 * C] = dv[kt][0][r];
                        dst[2 * C + 16] = dv[kt][1][r];
                    }
                }
            }
            for (int kt = nv; kt < L; ++kt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* dst = p.dqkv + ((size_t)kt * HW + pix_hr[r]) * C3 + head * DH + t;
                    dst[C] = 0.f; dst[C + 16] = 0.f; dst[2 * C] = 0.f; dst[2 * C + 16] = 0.f;
                }
            }
        }
    }
    // ---- this wave's share of the bias-table gradient into the global fixed-point table
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    for (int k = 0; k < hpw; ++k) {
        const int head = wave + 4 * k;
        if (head >= p.heads) break;
        const unsigned long long* dtab = reinterpret_cast<const unsigned long long*>(wl + (size_t)k * tab_s * 3 + tab_s);
        for (int i = lane; i < tab_n; i += 64) {
            const unsigned long long v = dtab[i];
            if (v) atomicAdd(reinterpret_cast<unsigned long long*>(p.dtable) + (size_t)i * p.heads + head, v);
        }
    }
}

__global__ __launch_bounds__(256) void fixed_to_float_kernel(const long long* __restrict__ acc, float* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (float)((double)acc[i] * (1.0 / 4294967296.0));
}

inline unsigned grid_for(size_t n, unsigned cap = 8192) { const size_t b = (n + 255) / 256; return (unsigned)(b < cap ? (b ? b : 1) : cap); }

}  // namespace

extern "C" int32_t av2x_layernorm_backward_rows(int64_t n_tokens) {
    const int tpb = 64;
    return (int32_t)((n_tokens + tpb - 1) / tpb);
}

extern "C" int av2x_layernorm_backward(const float* x, const float* gamma, const float* dy, int64_t n_tokens, int32_t c, float eps,
                                       float* dx, float* partial, av2x_stream_t stream) {
    if (!x || !gamma || !dy || !dx || !partial) return av2x::fail("av2x_layernorm_backward: null argument");
    if (n_tokens <= 0) return 0;
    const int tpb = 64;
    const unsigned nb = (unsigned)((n_tokens + tpb - 1) / tpb);
    hipStream_t st = av2x::as_stream(stream);
    auto X = reinterpret_cast<const float4*>(x);
    auto G = reinterpret_cast<const float4*>(gamma);
    auto D = reinterpret_cast<const float4*>(dy);
    auto O = reinterpret_cast<float4*>(dx);
    switch (c) {
        case 256: hipLaunchKernelGGL(layernorm_backward_kernel<1>, dim3(nb), dim3(256), 0, st, X, G, D, O, partial, (long long)n_tokens, tpb, eps); break;
        case 512: hipLaunchKernelGGL(layernorm_backward_kernel<2>, dim3(nb), dim3(256), 0, st, X, G, D, O, partial, (long long)n_tokens, tpb, eps); break;
        default: return av2x::fail("av2x_layernorm_backward: c=%d unsupported (256/512)", c);
    }
    return av2x::check_launch("layernorm_backward_kernel");
}

extern "C" int av2x_gelu(const float* z, const float* dy, float* out, uint64_t n, av2x_stream_t stream) {
    if (!z || !out || n % 4) return av2x::fail("av2x_gelu: null argument or n %% 4 != 0");
    if (n == 0) return 0;
    hipLaunchKernelGGL(gelu_kernel, dim3(grid_for(n / 4)), dim3(256), 0, av2x::as_stream(stream), reinterpret_cast<const float4*>(z),
                       reinterpret_cast<const float4*>(dy), reinterpret_cast<float4*>(out), (size_t)(n / 4));
    return av2x::check_launch("gelu_kernel");
}

extern "C" int av2x_scale_broadcast(const float* dy, float* dx, int32_t n_agents, uint64_t elems_per_agent, float scale, av2x_stream_t stream) {
    if (!dy || !dx || elems_per_agent % 4 || n_agents < 1) return av2x::fail("av2x_scale_broadcast: bad argument");
    hipLaunchKernelGGL(scale_broadcast_kernel, dim3(grid_for(elems_per_agent / 4)), dim3(256), 0, av2x::as_stream(stream),
                       reinterpret_cast<const float4*>(dy), reinterpret_cast<float4*>(dx), n_agents, (size_t)(elems_per_agent / 4), scale);
    return av2x::check_launch("scale_broadcast_kernel");
}

extern "C" int av2x_dropout(const float* x, const uint8_t* mask, float* y, uint64_t n, float scale, av2x_stream_t stream) {
    if (!x || !mask || !y || n % 4) return av2x::fail("av2x_dropout: null argument or n %% 4 != 0");
    if (n == 0) return 0;
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n / 4)), dim3(256), 0, av2x::as_stream(stream), reinterpret_cast<const float4*>(x),
                       reinterpret_cast<const unsigned*>(mask), reinterpret_cast<float4*>(y), (size_t)(n / 4), scale);
    return av2x::check_launch("dropout_kernel");
}

extern "C" int av2x_dropout_seeded(const float* x, const float* residual, float* y, uint64_t n, float p, uint64_t seed, av2x_stream_t stream) {
    if (!x || !y || n % 4) return av2x::fail("av2x_dropout_seeded: null argument or n %% 4 != 0");
    if (!(p >= 0.f && p < 1.f)) return av2x::fail("av2x_dropout_seeded: p = %g (0 <= p < 1)", (double)p);
    if (n == 0) return 0;
    hipLaunchKernelGGL(dropout_seeded_kernel, dim3(grid_for(n / 4)), dim3(256), 0, av2x::as_stream(stream), reinterpret_cast<const float4*>(x),
                       reinterpret_cast<const float4*>(residual), reinterpret_cast<float4*>(y), (size_t)(n / 4), p, 1.0f / (1.0f - p), (unsigned)seed,
                       (unsigned)(seed >> 32));
    return av2x::check_launch("dropout_seeded_kernel");
}

extern "C" uint64_t av2x_fax_attention_backward_workspace_bytes(int32_t n_agents_padded, int32_t window, int32_t heads) {
    return (uint64_t)(2 * n_agents_padded - 1) * (2 * window - 1) * (2 * window - 1) * heads * 8ull;
}

extern "C" int av2x_fax_attention_backward(const float* qkv, const float* bias_table, const float* out, const float* dout,
                                           int32_t n_agents_padded, int32_t n_valid, int32_t h, int32_t w, int32_t window, int32_t heads,
                                           int32_t dim_head, int32_t grid_partition, float* dqkv, float* dbias_table, void* workspace,
                                           av2x_stream_t stream) {
    if (!qkv || !bias_table || !out || !dout || !dqkv || !dbias_table || !workspace) return av2x::fail("av2x_fax_attention_backward: null argument");
    if (dim_head != DH) return av2x::fail("av2x_fax_attention_backward: dim_head=%d unsupported (32)", dim_head);
    if (n_valid < 1 || n_valid > n_agents_padded || window < 1 || h % window || w % window || heads < 1)
        return av2x::fail("av2x_fax_attention_backward: bad sizes");
    const int T = n_agents_padded * window * window;
    const int tab_n = (2 * n_agents_padded - 1) * (2 * window - 1) * (2 * window - 1);
    const int tab_p = (tab_n + 3) & ~3;
    const size_t lds = 2 * ((size_t)4 * T * DH + 3 * T + tab_p + 2 * (size_t)tab_p) * sizeof(float);
    FaxBwdParams p;
    p.qkv = qkv; p.table = bias_table; p.out = out; p.dout = dout; p.dqkv = dqkv; p.dtable = reinterpret_cast<long long*>(workspace);
    p.L = n_agents_padded; p.n_valid = n_valid; p.H = h; p.W = w; p.ws = window; p.heads = heads; p.grid = grid_partition & 1;
    p.scale = 1.0f / sqrtf((float)dim_head);
    hipStream_t st = av2x::as_stream(stream);
    hipError_t e = hipMemsetAsync(workspace, 0, (size_t)tab_n * heads * 8ull, st);
    if (e != hipSuccess) return av2x::fail("av2x_fax_attention_backward: memset: %s", hipGetErrorString(e));
    static const bool no_wave = [] { const char* e = getenv("AV2X_FAX_BWD_NO_WAVE"); return e && e[0] == '1'; }();
    const int tab_s = (tab_n + 63) & ~63;
    const size_t lds_w = (size_t)4 * ((heads + 3) / 4) * tab_s * 3 * sizeof(float);
    if (window == 4 && n_valid <= 8 && lds_w <= 160 * 1024 && !no_wave) {      // the register-tile kernel: one wave per (window, head)
        static av2x::LdsLimit lw4, lw8;
        const int nwin = (h / 4) * (w / 4);
        const dim3 grid(nwin < 512 ? nwin : 512);
        if (n_valid <= 4) {
            lw4.ensure(reinterpret_cast<const void*>(&fax_attention_backward_wave_kernel<4>), lds_w);
            hipLaunchKernelGGL(fax_attention_backward_wave_kernel<4>, grid, dim3(256), lds_w, st, p);
        } else {
            lw8.ensure(reinterpret_cast<const void*>(&fax_attention_backward_wave_kernel<8>), lds_w);
            hipLaunchKernelGGL(fax_attention_backward_wave_kernel<8>, grid, dim3(256), lds_w, st, p);
        }
    } else {
        if (lds > 160 * 1024) return av2x::fail("av2x_fax_attention_backward: %d tokens per window need %zu bytes of LDS (max 160 KB)", T, lds);
        static av2x::LdsLimit lim;
        lim.ensure(reinterpret_cast<const void*>(&fax_attention_backward_kernel), lds);
        hipLaunchKernelGGL(fax_attention_backward_kernel, dim3((h / window) * (w / window)), dim3(128), lds, st, p);
    }
    hipLaunchKernelGGL(fixed_to_float_kernel, dim3((tab_n * heads + 255) / 256), dim3(256), 0, st, reinterpret_cast<const long long*>(workspace),
                       dbias_table, tab_n * heads);
    return av2x::check_launch("fax_attention_backward_kernel");
}
