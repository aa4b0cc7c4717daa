// Kernels of the CoBEVT fused-axial-attention fusion that are not GEMMs
// (the Linear layers run on conv_igemm_f32 as 1x1 convolutions over tokens):
//
//   layernorm_kernel     nn.LayerNorm over C (one wave per token, 16-byte loads)           HBM-bound
//   fax_attention_kernel window / grid attention of swap_fusion_modules.py:78-127: per (window, head)
//                        softmax(q*scale . k^T + relative-position bias, padded agents masked) . v
//                        with the 'b m d (x w1) (y w2)' / 'b m d (w1 x) (w2 y)' partitions done as
//                        index arithmetic on the NHWC token buffer (no rearrange copies)
//   agent_mean_kernel    mean over the agent axis (mlp_head Reduce, :270)                   HBM-bound
#include <cstdlib>

#include "av2x_common.hpp"

namespace {

// ---------------------------------------------------------------- LayerNorm
template <int CK>  // C = 256 * CK / ... : each lane holds CK float4 (C = 256*CK)
__global__ __launch_bounds__(256) void layernorm_kernel(const float4* __restrict__ x, const float4* __restrict__ gamma,
                                                        const float4* __restrict__ beta, float4* __restrict__ y,
                                                        int n_tokens, float eps, int relu) {
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= n_tokens) return;
    constexpr int C = 256 * CK;
    float4 v[CK];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < CK; ++k) {
        v[k] = x[(size_t)tok * (C / 4) + k * 64 + lane];
        s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < CK; ++k) {
        const float a = v[k].x - mean, b = v[k].y - mean, c = v[k].z - mean, d = v[k].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);  // biased variance, as torch.nn.functional.layer_norm
#pragma unroll
    for (int k = 0; k < CK; ++k) {
        const float4 g = gamma[k * 64 + lane], b = beta[k * 64 + lane];
        float4 r;
        r.x = (v[k].x - mean) * rstd * g.x + b.x;
        r.y = (v[k].y - mean) * rstd * g.y + b.y;
        r.z = (v[k].z - mean) * rstd * g.z + b.z;
        r.w = (v[k].w - mean) * rstd * g.w + b.w;
        if (relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
        y[(size_t)tok * (C / 4) + k * 64 + lane] = r;
    }
}

// The statistics half of layernorm_kernel<1> alone: stats[tok] = (mean, rstd) with the SAME arithmetic (same sums, same order), for a
// consumer that normalises while it loads (conv_igemm_x3p's LN form): (x - mean) * rstd * gamma + beta there has the bits of this file's
// kernel, and the normalised tensor (288 MB written + read per LayerNorm at 8 agents) never exists.
__global__ __launch_bounds__(256) void layernorm_stats_kernel(const float4* __restrict__ x, float2* __restrict__ stats, int n_tokens, float eps) {
    constexpr int C = 256, TPW = 4;                           // four tokens per wave: their 1-KB rows are requested together
    const int lane = threadIdx.x & 63;
    const int tok0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * TPW;
    if (tok0 >= n_tokens) return;
    float4 v[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tok = min(tok0 + t, n_tokens - 1);
        v[t] = x[(size_t)tok * (C / 4) + lane];
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        float s = (v[t].x + v[t].y) + (v[t].z + v[t].w);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / (float)C;
        const float a = v[t].x - mean, b = v[t].y - mean, c = v[t].z - mean, d = v[t].w - mean;
        float q = (a * a + b * b) + (c * c + d * d);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = 1.0f / sqrtf(q / (float)C + eps);
        if (lane == 0 && tok0 + t < n_tokens) stats[tok0 + t] = make_float2(mean, rstd);
    }
}

// ---------------------------------------------------------------- fused axial attention
struct FaxParams {
    const float* qkv;   // (L*H*W, 3*C) rows = tokens in NHWC order (agent-major)
    const float* table; // ((2L-1)*(2ws-1)^2, heads)
    float* out;         // (L*H*W, C)
    int L, n_valid, H, W, ws, heads, grid;
    float scale;
};

constexpr int DH = 32;  // dim_head (fax_fusion.dim_head, cobevt yaml)

__global__ __launch_bounds__(256) void fax_attention_kernel(const FaxParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ws = p.ws, ws2 = ws * ws;
    const int T = p.L * ws2, Tk = p.n_valid * ws2;
    const int X = p.H / ws, Y = p.W / ws;
    const int wx = blockIdx.x / Y, wy = blockIdx.x % Y;
    const int C = p.heads * DH, C3 = 3 * C;
    const int tab_n = (2 * p.L - 1) * (2 * ws - 1) * (2 * ws - 1);
    // per-wave LDS: K[Tk][32], V[Tk][32], bias table of the current head
    float* Kl = lds + (size_t)wave * (2 * Tk * DH + tab_n);
    float* Vl = Kl + Tk * DH;
    float* tab = Vl + Tk * DH;

    auto token_row = [&](int t) -> int {  // token (l, w1, w2) of this window -> row of the token buffer
        const int l = t / ws2, r = t - l * ws2, w1 = r / ws, w2 = r - w1 * ws;
        const int ph = p.grid ? (w1 * X + wx) : (wx * ws + w1);
        const int pw = p.grid ? (w2 * Y + wy) : (wy * ws + w2);
        return (l * p.H + ph) * p.W + pw;
    };

    const int nwaves = blockDim.x >> 6;   // 4, or fewer when K / V of that many heads do not fit the 160 KB of LDS at once
    for (int h = wave; h < p.heads; h += nwaves) {
        // stage K, V of the valid (un-padded) agents and this head's bias column
        for (int j0 = 0; j0 < Tk; j0 += 8) {
            const int j = j0 + (lane >> 3), d4 = lane & 7;
            if (j < Tk) {
                const float* src = p.qkv + (size_t)token_row(j) * C3 + h * DH + d4 * 4;
                *reinterpret_cast<float4*>(Kl + j * DH + d4 * 4) = *reinterpret_cast<const float4*>(src + C);
                *reinterpret_cast<float4*>(Vl + j * DH + d4 * 4) = *reinterpret_cast<const float4*>(src + 2 * C);
            }
        }
        for (int i = lane; i < tab_n; i += 64) tab[i] = p.table[(size_t)i * p.heads + h];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);

        for (int r0 = 0; r0 < T; r0 += 64) {
            const int t = r0 + lane;
            const bool act = t < T;
            const int tt = act ? t : 0;
            const int row = token_row(tt);
            float q[DH], o[DH];
            {
                const float* src = p.qkv + (size_t)row * C3 + h * DH;
#pragma unroll
                for (int d = 0; d < DH; d += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(src + d);
                    q[d] = v.x * p.scale; q[d + 1] = v.y * p.scale; q[d + 2] = v.z * p.scale; q[d + 3] = v.w * p.scale;
                }
            }
#pragma unroll
            for (int d = 0; d < DH; ++d) o[d] = 0.f;
            // relative-position index (swap_fusion_modules.py:53-75): (dl + L-1)*(2ws-1)^2 + (dh + ws-1)*(2ws-1) + (dw + ws-1)
            const int li = tt / ws2, ri = tt - li * ws2, hi = ri / ws, wi = ri - hi * ws;
            const int s1 = 2 * ws - 1;
            const int base_i = ((li + p.L - 1) * s1 + (hi + ws - 1)) * s1 + (wi + ws - 1);
            float m = -INFINITY, lsum = 0.f;
            for (int j = 0; j < Tk; ++j) {
                const int lj = j / ws2, rj = j - lj * ws2, hj = rj / ws, wj = rj - hj * ws;
                const float bias = tab[base_i - ((lj * s1 + hj) * s1 + wj)];
                const float* kj = Kl + j * DH;
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < DH; d += 4) {
                    const float4 kv = *reinterpret_cast<const float4*>(kj + d);
                    s = fmaf(q[d], kv.x, s); s = fmaf(q[d + 1], kv.y, s); s = fmaf(q[d + 2], kv.z, s); s = fmaf(q[d + 3], kv.w, s);
                }
                s += bias;
                const float mn = fmaxf(m, s);
                const float alpha = expf(m - mn), pj = expf(s - mn);
                lsum = lsum * alpha + pj;
                const float* vj = Vl + j * DH;
#pragma unroll
                for (int d = 0; d < DH; d += 4) {
                    const float4 vv = *reinterpret_cast<const float4*>(vj + d);
                    o[d] = fmaf(pj, vv.x, o[d] * alpha); o[d + 1] = fmaf(pj, vv.y, o[d + 1] * alpha);
                    o[d + 2] = fmaf(pj, vv.z, o[d + 2] * alpha); o[d + 3] = fmaf(pj, vv.w, o[d + 3] * alpha);
                }
                m = mn;
            }
            if (act) {
                const float inv = 1.0f / lsum;
                float* dst = p.out + (size_t)row * C + h * DH;
#pragma unroll
                for (int d = 0; d < DH; d += 4)
                    *reinterpret_cast<float4*>(dst + d) = make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------- fused axial attention on MFMA
// One workgroup (4 wave64) per window; heads are processed one after the other, the four waves
// take the four 32-query-row strips of the head (T = L*ws*ws <= 128 tokens), so K / V / the bias
// column of the head are staged in LDS once and shared.  Per strip:
//   S = (Q*scale) K^T      v_mfma_f32_32x32x2_f32, A = Q rows straight from global (each element is
//                          used by exactly one lane), B = K rows from LDS ([j][36] padded, b128)
//   S += bias, pad-mask    gathered from the staged table with base[row] - sub[col] (:53-75)
//   row max / exp / sum    the C layout keeps a row in one (register, half-wave): 5 DPP/shuffle steps
//   O = P V                P goes through LDS (C layout -> A layout), V is staged k-quad major
//                          ([j/4][d][4]) so that a lane's four k values are one b128; O is divided
//                          by the row sum held in the same (register, half) slot
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int KLD = 36;   // K row stride in LDS (floats)
constexpr int PLD = 68;   // P half-strip row stride (64 keys + 4)

__global__ __launch_bounds__(256) void fax_attention_mfma_kernel(const FaxParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li32 = lane & 31, lh = lane >> 5;
    const int ws = p.ws, ws2 = ws * ws, s1 = 2 * ws - 1;
    const int T = p.L * ws2, Tk = p.n_valid * ws2;
    const int TkP = (Tk + 31) & ~31;                      // keys padded to whole 32-column tiles
    const int X = p.H / ws, Y = p.W / ws;
    const int wx = blockIdx.x / Y, wy = blockIdx.x % Y;
    const int C = p.heads * DH, C3 = 3 * C;
    const int tab_n = (2 * p.L - 1) * s1 * s1;
    float* Ks = lds;                                       // [TkP][KLD]
    float* Vs = Ks + TkP * KLD;                            // [TkP/4][32][4]
    float* tab = Vs + TkP * DH;                            // [tab_n]
    int* subj = reinterpret_cast<int*>(tab + ((tab_n + 3) & ~3));   // [TkP]
    int* rowtok = subj + TkP;                              // [128] token row of every query of the window
    int* basei = rowtok + 128;                             // [128]
    float* Ps = reinterpret_cast<float*>(basei + 128) + wave * (32 * PLD);  // per wave [32][PLD]

    auto token_row = [&](int t) -> int {
        const int l = t / ws2, r = t - l * ws2, w1 = r / ws, w2 = r - w1 * ws;
        const int ph = p.grid ? (w1 * X + wx) : (wx * ws + w1);
        const int pw = p.grid ? (w2 * Y + wy) : (wy * ws + w2);
        return (l * p.H + ph) * p.W + pw;
    };
    if (tid < 128) {
        const int t = tid < T ? tid : T - 1;               // padding rows re-read the last token (never stored)
        const int l = t / ws2, r = t - l * ws2, w1 = r / ws, w2 = r - w1 * ws;
        rowtok[tid] = token_row(t);
        basei[tid] = ((l + p.L - 1) * s1 + (w1 + ws - 1)) * s1 + (w2 + ws - 1);
    }
    for (int j = tid; j < TkP; j += 256) {
        const int l = j / ws2, r = j - l * ws2, w1 = r / ws, w2 = r - w1 * ws;
        subj[j] = (l * s1 + w1) * s1 + w2;
    }
    const int strip = wave;                                // 32 query rows [32*strip, 32*strip+32)
    const bool strip_on = strip * 32 < T;

    for (int h = 0; h < p.heads; ++h) {
        __syncthreads();                                   // previous head fully consumed (and the index arrays visible)
        for (int idx = tid; idx < TkP * 8; idx += 256) {
            const int j = idx >> 3, d4 = idx & 7;
            f32x4v kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (j < Tk) {
                const float* src = p.qkv + (size_t)rowtok[j] * C3 + h * DH + d4 * 4;   // valid keys are tokens 0..Tk-1
                kv = *reinterpret_cast<const f32x4v*>(src + C);
                vv = *reinterpret_cast<const f32x4v*>(src + 2 * C);
            }
            *reinterpret_cast<f32x4v*>(Ks + j * KLD + d4 * 4) = kv;
            float* vd = Vs + ((j >> 2) * DH + d4 * 4) * 4 + (j & 3);
            vd[0] = vv.x; vd[4] = vv.y; vd[8] = vv.z; vd[12] = vv.w;
        }
        for (int i = tid; i < tab_n; i += 256) tab[i] = p.table[(size_t)i * p.heads + h];
        __syncthreads();
        if (!strip_on) continue;

        // ---- S strip = (Q * scale) K^T : A fragments straight from global
        f32x4v qa[4];
        {
            const float* qsrc = p.qkv + (size_t)rowtok[strip * 32 + li32] * C3 + h * DH + lh * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                qa[g] = *reinterpret_cast<const f32x4v*>(qsrc + g * 8);
                qa[g] *= p.scale;
            }
        }
        f32x16 sacc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[nt][r] = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            if (nt * 32 < TkP) {
                const float* kb = Ks + (nt * 32 + li32) * KLD + lh * 4;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4v kf = *reinterpret_cast<const f32x4v*>(kb + g * 8);
                    sacc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[g].x, kf.x, sacc[nt], 0, 0, 0);
                    sacc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[g].y, kf.y, sacc[nt], 0, 0, 0);
                    sacc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[g].z, kf.z, sacc[nt], 0, 0, 0);
                    sacc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[g].w, kf.w, sacc[nt], 0, 0, 0);
                }
            }
        }
        // ---- bias + padding mask, row max, exp, row sum (row of register r: (r&3) + 8*(r>>2) + 4*lh)
        float rmax[16], rsum[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int bi = basei[strip * 32 + row];
            float m = -INFINITY;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                if (nt * 32 < TkP) {
                    const int col = nt * 32 + li32;
                    float s = sacc[nt][r] + tab[bi - subj[col]];
                    s = col < Tk ? s : -INFINITY;
                    sacc[nt][r] = s;
                    m = fmaxf(m, s);
                }
            }
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 32));
            rmax[r] = m;
            float sum = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                if (nt * 32 < TkP) {
                    const float e = expf(sacc[nt][r] - m);
                    sacc[nt][r] = e;
                    sum += e;
                }
            }
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 32);
            rsum[r] = sum;
        }
        // ---- O = P V, 64 keys at a time through the per-wave LDS transpose buffer
        f32x16 oacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            if (hh * 64 < TkP) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int nt = hh * 2 + tt;
                    if (nt * 32 < TkP) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            Ps[((r & 3) + 8 * (r >> 2) + 4 * lh) * PLD + tt * 32 + li32] = sacc[nt][r];
                    }
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xc07f);
                const int kcols = (TkP - hh * 64) < 64 ? (TkP - hh * 64) : 64;   // 32 or 64 keys in this half
                const float* pb = Ps + li32 * PLD + lh * 4;
                const float* vb = Vs + ((hh * 16 + lh) * DH + li32) * 4;
                for (int g = 0; g < kcols / 8; ++g) {
                    const f32x4v pf = *reinterpret_cast<const f32x4v*>(pb + g * 8);
                    const f32x4v vf = *reinterpret_cast<const f32x4v*>(vb + g * 2 * DH * 4);
                    oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(pf.x, vf.x, oacc, 0, 0, 0);
                    oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(pf.y, vf.y, oacc, 0, 0, 0);
                    oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(pf.z, vf.z, oacc, 0, 0, 0);
                    oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(pf.w, vf.w, oacc, 0, 0, 0);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        // ---- normalise and store: row (r, lh) of the strip, column d = lane & 31
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = strip * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row < T) p.out[(size_t)rowtok[row] * C + h * DH + li32] = oacc[r] / rsum[r];
        }
    }
}

// ---------------------------------------------------------------- ws = 4 specialisation, P kept in registers
// Same work split as fax_attention_mfma_kernel (one workgroup per window, heads in sequence, wave = 32-query strip),
// but the first product is computed TRANSPOSED: S^T = K (Q*scale)^T, i.e. MFMA row = key, col = query.  In the
// 32x32 C/D layout (col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)) every lane then owns ONE query and its
// 16 registers per key tile are 16 different keys, so
//   * the softmax over keys is an in-lane reduction plus ONE cross-lane step (xor 32) instead of 10 shuffles per row,
//   * P[query i][key j] is already the A operand of O = P V (A[row = lane&31][k = lane>>5]): MFMA number r of a
//     key tile uses A = P reg r and B = V[key j(r, half)][n = lane&31] -- no transpose through LDS,
//   * the relative-position bias index base[i] - sub[j] has a compile-time sub[j] up to the lane half:
//     key j = 32*tile + (r&3) + 8*(r>>2) + 4*half  ->  agent 2*tile + (r>>3), w1 = 2*((r>>2)&1) + half, w2 = r&3.
// V is staged row-major [key][40] (conflict-free ds_read_b32 for two rows 4 apart), K as before [key][36].
constexpr int VLD4 = 40;

__global__ __launch_bounds__(256) void fax_attention_mfma4_kernel(const FaxParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int ws = 4, ws2 = 16, s1 = 7;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li32 = lane & 31, lh = lane >> 5;
    const int T = p.L * ws2, Tk = p.n_valid * ws2;
    const int TkP = (Tk + 31) & ~31;
    const int ktiles = TkP >> 5;                           // 1..4 key tiles of 32
    const int X = p.H / ws, Y = p.W / ws;
    const int wx = blockIdx.x / Y, wy = blockIdx.x % Y;
    const int C = p.heads * DH, C3 = 3 * C;
    const int tab_n = (2 * p.L - 1) * s1 * s1;
    float* Ks = lds;                                       // [TkP][KLD]
    float* Vs = Ks + TkP * KLD;                            // [TkP][VLD4]
    float* tab = Vs + TkP * VLD4;                          // [tab_n]
    int* rowtok = reinterpret_cast<int*>(tab + ((tab_n + 3) & ~3));   // [128]

    if (tid < 128) {
        const int t = tid < T ? tid : T - 1;               // padding queries re-read the last token (never stored)
        const int l = t >> 4, w1 = (t >> 2) & 3, w2 = t & 3;
        const int ph = p.grid ? (w1 * X + wx) : (wx * ws + w1);
        const int pw = p.grid ? (w2 * Y + wy) : (wy * ws + w2);
        rowtok[tid] = (l * p.H + ph) * p.W + pw;
    }
    const int strip = wave;
    const bool strip_on = strip * 32 < T;
    // this lane's query i = 32*strip + li32: bias base index, minus the half-dependent part of sub[j]
    int bih;
    {
        const int t = min(strip * 32 + li32, T - 1);
        const int l = t >> 4, w1 = (t >> 2) & 3, w2 = t & 3;
        bih = ((l + p.L - 1) * s1 + (w1 + ws - 1)) * s1 + (w2 + ws - 1) - lh * s1;
    }
    const float kLog2e = 1.4426950408889634f;

    for (int h = 0; h < p.heads; ++h) {
        __syncthreads();                                   // previous head fully consumed (and rowtok visible)
        for (int idx = tid; idx < TkP * 8; idx += 256) {
            const int j = idx >> 3, d4 = idx & 7;
            f32x4v kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (j < Tk) {
                const float* src = p.qkv + (size_t)rowtok[j] * C3 + h * DH + d4 * 4;
                kv = *reinterpret_cast<const f32x4v*>(src + C);
                vv = *reinterpret_cast<const f32x4v*>(src + 2 * C);
            }
            *reinterpret_cast<f32x4v*>(Ks + j * KLD + d4 * 4) = kv;
            *reinterpret_cast<f32x4v*>(Vs + j * VLD4 + d4 * 4) = vv;
        }
        for (int i = tid; i < tab_n; i += 256) tab[i] = p.table[(size_t)i * p.heads + h];
        __syncthreads();
        if (!strip_on) continue;

        // ---- S^T tiles: A = K rows from LDS, B = (Q * scale) rows straight from global
        f32x4v qa[4];
        {
            const float* qsrc = p.qkv + (size_t)rowtok[strip * 32 + li32] * C3 + h * DH + lh * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                qa[g] = *reinterpret_cast<const f32x4v*>(qsrc + g * 8);
                qa[g] *= p.scale;
            }
        }
        f32x16 sacc[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
            if (kt < ktiles) {
                const float* kb = Ks + (kt * 32 + li32) * KLD + lh * 4;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4v kf = *reinterpret_cast<const f32x4v*>(kb + g * 8);
                    sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qa[g].x, sacc[kt], 0, 0, 0);
                    sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qa[g].y, sacc[kt], 0, 0, 0);
                    sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qa[g].z, sacc[kt], 0, 0, 0);
                    sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qa[g].w, sacc[kt], 0, 0, 0);
                }
            }
        }
        // ---- bias, padding mask, max over keys (in-lane + xor 32)
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt < ktiles) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int agent = 2 * kt + (r >> 3);                       // wave-uniform
                    const int sub0 = (agent * s1 + 2 * ((r >> 2) & 1)) * s1 + (r & 3);   // sub[j] without the half term
                    float sv = sacc[kt][r] + tab[bih - sub0];
                    sv = agent < p.n_valid ? sv : -INFINITY;
                    sacc[kt][r] = sv;
                    m = fmaxf(m, sv);
                }
            }
        }
        m = fmaxf(m, __shfl_xor(m, 32));
        const float mb = m * kLog2e;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt < ktiles) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(sacc[kt][r], kLog2e, -mb));   // exp(s - m)
                    sacc[kt][r] = e;
                    sum += e;
                }
            }
        }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        // ---- O = P V : A = P registers (normalised), B = V rows from LDS
        f32x16 oacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt < ktiles) {
                const float* vb = Vs + (kt * 32 + 4 * lh) * VLD4 + li32;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(sacc[kt][r] * inv, vb[((r & 3) + 8 * (r >> 2)) * VLD4], oacc, 0, 0, 0);
            }
        }
        // ---- store: row (r, lh) of the strip = query, column d = lane & 31
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = strip * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row < T) p.out[(size_t)rowtok[row] * C + h * DH + li32] = oacc[r];
        }
    }
}

// ---------------------------------------------------------------- ws = 4, split-3 operands on the bf16 matrix cores (round 5)
// fax_attention_mfma4_kernel's work split and S^T trick (one workgroup per window, heads in sequence, wave = 32-query strip, every lane owns
// ONE query so P stays in registers), with both contractions on v_mfma_f32_32x32x16_bf16: every fp32 operand -- K, V, the scaled Q and the
// normalised probabilities P -- enters as hi + mid + lo bf16 terms (the value to 2^-24), six partial products accumulated in fp32, smallest
// first, exactly as the convolutions of the x3 mode (conv_x3p.hip).  gfx950's fp32-input MFMA runs at 1/16 of this pipe: 96 bf16 MFMAs of 32
// cycles replace 128 fp32 MFMAs of 64 cycles per (strip, head).  K and V are split ONCE per (window, head) while they are staged into LDS
// (amortised over the four strips): K as three [key][32 d] bf16 planes (80-byte rows: conflict-free 16-byte A fragments), V TRANSPOSED as
// three [d][key slot] planes so that a B fragment (8 consecutive k of one column) is one 16-byte read; the key slots follow the register
// order of P -- lane half hh, element e of k16-step s of a 32-key tile holds key 16 s + 8 (e >> 2) + 4 hh + (e & 3) -- so P's registers
// 8 s .. 8 s + 7 ARE the A fragment of step s.
typedef __bf16 fx_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned fx_u32x4 __attribute__((ext_vector_type(4)));
constexpr int KROW3 = 40;                                   // bf16 per K row (32 + 8 pad)

typedef float fx_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 fx_bf16x2 __attribute__((ext_vector_type(2)));
// (a, b) = hi + mid + lo per component (each subtraction exact): three packed bf16 pairs (a in the low half) -- one v_cvt_pk_bf16_f32 per
// plane, the converted pair back as fp32 with a shift and a mask: 11 VALU per pair, as x3_split_step of the convolutions
__device__ __forceinline__ void fx_split2(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(fx_f32x2{a, b}, fx_bf16x2));
    const float ra = a - __builtin_bit_cast(float, hi << 16), rb = b - __builtin_bit_cast(float, hi & 0xffff0000u);
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector(fx_f32x2{ra, rb}, fx_bf16x2));
    const float sa = ra - __builtin_bit_cast(float, mid << 16), sb = rb - __builtin_bit_cast(float, mid & 0xffff0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(fx_f32x2{sa, sb}, fx_bf16x2));
}
// eight values -> three A / B fragments (planes hi, mid, lo), element e in bits 16 (e & 1) of dword e >> 1
__device__ __forceinline__ void fx_split8(const float (&v)[8], fx_u32x4 (&pl)[3]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned h, m, l;
        fx_split2(v[2 * q], v[2 * q + 1], h, m, l);
        pl[0][q] = h; pl[1][q] = m; pl[2][q] = l;
    }
}
// the six partial products >= 2^-16 of the full product, smallest first (A plane, B plane): lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi
template <bool FIRST = false>      // FIRST: the accumulator starts at zero (an inline-constant C operand: no 16 register moves)
__device__ __forceinline__ f32x16 fx_mma6(const fx_u32x4 (&a)[3], const fx_u32x4 (&b)[3], f32x16 acc) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    if constexpr (FIRST) {
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(fx_bf16x8, a[PA[0]]), __builtin_bit_cast(fx_bf16x8, b[PB[0]]), z, 0, 0, 0);
    }
#pragma unroll
    for (int i = FIRST ? 1 : 0; i < 6; ++i)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(fx_bf16x8, a[PA[i]]), __builtin_bit_cast(fx_bf16x8, b[PB[i]]), acc, 0, 0, 0);
    return acc;
}

__global__ __launch_bounds__(256, 2) void fax_attention_x3_kernel(const FaxParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
    constexpr int ws = 4, ws2 = 16, s1 = 7;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li32 = lane & 31, lh = lane >> 5;
    const int T = p.L * ws2, Tk = p.n_valid * ws2;
    const int TkP = (Tk + 31) & ~31;
    const int ktiles = TkP >> 5;                           // 1..4 key tiles of 32
    const int X = p.H / ws, Y = p.W / ws;
    const int wx = blockIdx.x / Y, wy = blockIdx.x % Y;
    const int C = p.heads * DH, C3 = 3 * C;
    const int tab_n = (2 * p.L - 1) * s1 * s1;
    const int VROW = TkP + 8;                              // bf16 per V^T row (key slots + pad: conflict-free 16-byte B fragments)
    const int KPL = TkP * KROW3 * 2, VPL = DH * VROW * 2;  // bytes per plane
    unsigned char* Kp = lds3;                              // [3][TkP][KROW3] bf16
    unsigned char* Vt = Kp + 3 * KPL;                      // [3][32 d][VROW] bf16
    float* tab = reinterpret_cast<float*>(Vt + 3 * VPL);   // [tab_n]
    int* rowtok = reinterpret_cast<int*>(tab + ((tab_n + 3) & ~3));   // [128]

    if (tid < 128) {
        const int t = tid < T ? tid : T - 1;               // padding queries re-read the last token (never stored)
        const int l = t >> 4, w1 = (t >> 2) & 3, w2 = t & 3;
        const int ph = p.grid ? (w1 * X + wx) : (wx * ws + w1);
        const int pw = p.grid ? (w2 * Y + wy) : (wy * ws + w2);
        rowtok[tid] = (l * p.H + ph) * p.W + pw;
    }
    const int strip = wave;
    const bool strip_on = strip * 32 < T;
    int bih;                                               // this lane's query i = 32 strip + li32: bias base index (see fax_attention_mfma4_kernel)
    {
        const int t = min(strip * 32 + li32, T - 1);
        const int l = t >> 4, w1 = (t >> 2) & 3, w2 = t & 3;
        bih = ((l + p.L - 1) * s1 + (w1 + ws - 1)) * s1 + (w2 + ws - 1) - lh * s1;
    }
    const float kLog2e = 1.4426950408889634f;

    // K / V / Q of head h + 1 are fetched into registers while head h is computed (two workgroups of 8 waves per CU hide little of a
    // global round trip per head otherwise): kreg / vreg / qreg hold the raw fp32 rows, split and stored at the top of the next iteration
    f32x4v kreg[4], vreg[2][2], qreg[4];
    float treg[4];                                         // this thread's entries of the head's bias-table column (tab_n <= 1024)
    auto fetch = [&](int h) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = tid + 256 * it;
            treg[it] = i < tab_n ? p.table[(size_t)i * p.heads + h] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + 256 * it, j = idx >> 3, d4 = idx & 7;
            kreg[it] = f32x4v{0.f, 0.f, 0.f, 0.f};
            if (idx < TkP * 8 && j < Tk) kreg[it] = *reinterpret_cast<const f32x4v*>(p.qkv + (size_t)rowtok[j] * C3 + C + h * DH + d4 * 4);
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + 256 * it, j = 2 * (idx >> 3), d4 = idx & 7;
            vreg[it][0] = vreg[it][1] = f32x4v{0.f, 0.f, 0.f, 0.f};
            if (idx < TkP * 4) {
                if (j < Tk) vreg[it][0] = *reinterpret_cast<const f32x4v*>(p.qkv + (size_t)rowtok[j] * C3 + 2 * C + h * DH + d4 * 4);
                if (j + 1 < Tk) vreg[it][1] = *reinterpret_cast<const f32x4v*>(p.qkv + (size_t)rowtok[j + 1] * C3 + 2 * C + h * DH + d4 * 4);
            }
        }
        if (strip_on) {
            const float* qsrc = p.qkv + (size_t)rowtok[strip * 32 + li32] * C3 + h * DH + lh * 8;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                qreg[2 * s] = *reinterpret_cast<const f32x4v*>(qsrc + 16 * s);
                qreg[2 * s + 1] = *reinterpret_cast<const f32x4v*>(qsrc + 16 * s + 4);
            }
        }
    };
    __syncthreads();                                       // rowtok visible
    fetch(0);
    for (int h = 0; h < p.heads; ++h) {
        __syncthreads();                                   // previous head fully consumed
        // ---- K: (key j, d quad) items, split, 8 bytes per plane
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + 256 * it, j = idx >> 3, d4 = idx & 7;
            if (idx < TkP * 8) {
                unsigned h0, m0, l0, h1, m1, l1;
                fx_split2(kreg[it][0], kreg[it][1], h0, m0, l0);
                fx_split2(kreg[it][2], kreg[it][3], h1, m1, l1);
                typedef unsigned fx_u32x2 __attribute__((ext_vector_type(2)));
                unsigned char* dst = Kp + (j * KROW3 + d4 * 4) * 2;
                *reinterpret_cast<fx_u32x2*>(dst) = fx_u32x2{h0, h1};
                *reinterpret_cast<fx_u32x2*>(dst + KPL) = fx_u32x2{m0, m1};
                *reinterpret_cast<fx_u32x2*>(dst + 2 * KPL) = fx_u32x2{l0, l1};
            }
        }
        // ---- V transposed: (key pair, d quad) items; keys 2 jp, 2 jp + 1 share a dword of every d row
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + 256 * it, j = 2 * (idx >> 3), d4 = idx & 7;
            if (idx < TkP * 4) {
                const int kk = j & 31, rem = kk & 15;
                const int slot = (j & ~31) + 16 * (kk >> 4) + 8 * ((rem >> 2) & 1) + 4 * (rem >> 3) + (rem & 3);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned hv, mv, lv;
                    fx_split2(vreg[it][0][e], vreg[it][1][e], hv, mv, lv);
                    unsigned char* dst = Vt + ((d4 * 4 + e) * VROW + slot) * 2;
                    *reinterpret_cast<unsigned*>(dst) = hv;
                    *reinterpret_cast<unsigned*>(dst + VPL) = mv;
                    *reinterpret_cast<unsigned*>(dst + 2 * VPL) = lv;
                }
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it)
            if (tid + 256 * it < tab_n) tab[tid + 256 * it] = treg[it];
        // ---- Q * scale of this lane's query, d = 16 s + 8 lh .. + 7, split in registers
        fx_u32x4 qpl[2][3];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f32x4v a = qreg[2 * s], b = qreg[2 * s + 1];
            const float v[8] = {a.x * p.scale, a.y * p.scale, a.z * p.scale, a.w * p.scale, b.x * p.scale, b.y * p.scale, b.z * p.scale, b.w * p.scale};
            fx_split8(v, qpl[s]);
        }
        if (h + 1 < p.heads) fetch(h + 1);                 // in flight during this head's products and softmax
        __syncthreads();
        if (!strip_on) continue;

        // ---- S^T tiles: A = K planes from LDS (row = key), B = Q planes (column = query)
        f32x16 sacc[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt < ktiles) {             // (tiles beyond ktiles are never read: every use below is under the same condition)
                const unsigned char* kb = Kp + ((kt * 32 + li32) * KROW3 + lh * 8) * 2;
                fx_u32x4 ka[2][3];
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) ka[s][pl] = *reinterpret_cast<const fx_u32x4*>(kb + pl * KPL + s * 32);
                sacc[kt] = fx_mma6<true>(ka[0], qpl[0], sacc[kt]);
                sacc[kt] = fx_mma6(ka[1], qpl[1], sacc[kt]);
            }
        }
        // ---- bias, padding mask, max over keys (in-lane + xor 32): as fax_attention_mfma4_kernel
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt < ktiles) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int agent = 2 * kt + (r >> 3);                       // wave-uniform
                    const int sub0 = (agent * s1 + 2 * ((r >> 2) & 1)) * s1 + (r & 3);   // sub[j] without the half term
                    float sv = sacc[kt][r] + tab[bih - sub0];
                    sv = agent < p.n_valid ? sv : -INFINITY;
                    sacc[kt][r] = sv;
                    m = fmaxf(m, sv);
                }
            }
        }
        m = fmaxf(m, __shfl_xor(m, 32));
        const float mb = m * kLog2e;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt < ktiles) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(sacc[kt][r], kLog2e, -mb));   // exp(s - m)
                    sacc[kt][r] = e;
                    sum += e;
                }
            }
        }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        // ---- O = P V: A = P registers 8 s .. 8 s + 7 (normalised, split), B = V^T planes from LDS
        f32x16 oacc;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt < ktiles) {
                const unsigned char* vb = Vt + (li32 * VROW + kt * 32 + lh * 8) * 2;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float pv[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) pv[e] = sacc[kt][8 * s + e] * inv;
                    fx_u32x4 pa[3], vv[3];
                    fx_split8(pv, pa);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) vv[pl] = *reinterpret_cast<const fx_u32x4*>(vb + pl * VPL + s * 32);
                    if (kt == 0 && s == 0) oacc = fx_mma6<true>(pa, vv, oacc);      // ktiles >= 1: the first product starts the accumulator
                    else oacc = fx_mma6(pa, vv, oacc);
                }
            }
        }
        // ---- store: row (r, lh) of the strip = query, column d = lane & 31
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = strip * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row < T) p.out[(size_t)rowtok[row] * C + h * DH + li32] = oacc[r];
        }
    }
}

// ---------------------------------------------------------------- ws = 4, one WAVE per (window, head): no barriers
// The whole attention of a (window, head) lives in one wave's registers (v_mfma_f32_16x16x4_f32 tiles = one agent's 16
// tokens): K and V fragments of the valid agents are loaded once, then for every query agent qt
//   S^T[key tile kt][query] = K_kt (Q_qt*scale)^T   (row = key 4h+r -> (w1 = h, w2 = r), col = query lane&15)
//   + bias tab[((qt-kt+L-1)*7 + (w1q-h+3))*7 + (w2q-r+3)], softmax over (kt, r) in-lane + lanes xor 16 / 32,
//   O = P V with P already the A operand (see window_attn_mfma_kernel in v2xvit.hip), stored 64 B per row.
// Wave w of a workgroup serves heads w, w+4, ... of the workgroup's windows, so its bias tables are staged once.
typedef float f32x4w __attribute__((ext_vector_type(4)));

template <int NV>   // compile-time bound on the valid agents: 4 or 8 key tiles held in registers
__global__ __launch_bounds__(256) void fax_attention_wave_kernel(const FaxParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = lane & 15, h = lane >> 4;
    const int X = p.H / 4, Y = p.W / 4, nwin = X * Y;
    const int C = p.heads * DH, C3 = 3 * C;
    const int L = p.L, nv = p.n_valid;
    const int tab_n = (2 * L - 1) * 49, tab_s = (tab_n + 63) & ~63;
    const int hpw = (p.heads + 3) >> 2;                     // heads per wave
    float* tabs = lds + (size_t)wave * hpw * tab_s;
    for (int k = 0; k < hpw; ++k) {
        const int head = wave + 4 * k;
        if (head < p.heads)
            for (int i = lane; i < tab_n; i += 64) tabs[k * tab_s + i] = p.table[(size_t)i * p.heads + head];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    const int w1q = t >> 2, w2q = t & 3;
    const size_t HW = (size_t)p.H * p.W;
    const float kLog2e = 1.4426950408889634f;
    for (int win = blockIdx.x; win < nwin; win += gridDim.x) {
        const int wx = win / Y, wy = win - wx * Y;
        // pixel of window token (w1, w2)
        const int ph_q = p.grid ? (w1q * X + wx) : (wx * 4 + w1q), pw_q = p.grid ? (w2q * Y + wy) : (wy * 4 + w2q);
        const size_t pix_t = (size_t)ph_q * p.W + pw_q;      // this lane's token as key row / query column
        const int ph_h = p.grid ? (h * X + wx) : (wx * 4 + h);
        size_t pix_hr[4];                                     // token (w1 = h, w2 = r): V rows and output rows
#pragma unroll
        for (int r = 0; r < 4; ++r) pix_hr[r] = (size_t)ph_h * p.W + (p.grid ? (r * Y + wy) : (wy * 4 + r));
        for (int k = 0; k < hpw; ++k) {
            const int head = wave + 4 * k;
            if (head >= p.heads) break;
            const float* tab = tabs + k * tab_s;
            const float* base = p.qkv + head * DH;
            f32x4w kf[NV][2];
            float vf[NV][4][2];
#pragma unroll
            for (int kt = 0; kt < NV; ++kt) {
                if (kt < nv) {
                    const float* kr = base + ((size_t)kt * HW + pix_t) * C3 + C + 4 * h;
                    kf[kt][0] = *reinterpret_cast<const f32x4w*>(kr);
                    kf[kt][1] = *reinterpret_cast<const f32x4w*>(kr + 16);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float* vr = base + ((size_t)kt * HW + pix_hr[r]) * C3 + 2 * C + t;
                        vf[kt][r][0] = vr[0];
                        vf[kt][r][1] = vr[16];
                    }
                }
            }
            for (int qt = 0; qt < L; ++qt) {
                const float* qr = base + ((size_t)qt * HW + pix_t) * C3 + 4 * h;
                f32x4w q0 = *reinterpret_cast<const f32x4w*>(qr), q1 = *reinterpret_cast<const f32x4w*>(qr + 16);
                q0 *= p.scale; q1 *= p.scale;
                const int cq = ((qt + L - 1) * 7 + (w1q - h + 3)) * 7 + (w2q + 3);
                f32x4w st[NV];
                float m = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < NV; ++kt) {
                    if (kt < nv) {
                        f32x4w a = {0.f, 0.f, 0.f, 0.f};
                        a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][0].x, q0.x, a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][0].y, q0.y, a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][0].z, q0.z, a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][0].w, q0.w, a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][1].x, q1.x, a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][1].y, q1.y, a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][1].z, q1.z, a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][1].w, q1.w, a, 0, 0, 0);
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
                f32x4w o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < NV; ++kt) {
                    if (kt < nv) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pr = st[kt][r] * inv;
                            o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pr, vf[kt][r][0], o0, 0, 0, 0);
                            o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pr, vf[kt][r][1], o1, 0, 0, 0);
                        }
                    }
                }
                float* ob = p.out + head * DH + t;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* o = ob + ((size_t)qt * HW + pix_hr[r]) * C;
                    o[0] = o0[r];
                    o[16] = o1[r];
                }
            }
        }
    }
}

__global__ void agent_mean_kernel(const float4* __restrict__ x, float4* __restrict__ y, size_t n4, int L) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 s = x[i];
        for (int l = 1; l < L; ++l) {
            const float4 v = x[i + (size_t)l * n4];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const float inv = (float)L;
        y[i] = make_float4(s.x / inv, s.y / inv, s.z / inv, s.w / inv);
    }
}

}  // namespace

extern "C" int av2x_layernorm_act(const float* x, const float* gamma, const float* beta, float* y, int64_t n_tokens,
                                  int32_t c, float eps, int32_t relu, av2x_stream_t stream);

extern "C" int av2x_layernorm_stats(const float* x, float* stats, int64_t n_tokens, int32_t c, float eps, av2x_stream_t stream) {
    if (n_tokens == 0) return 0;
    if (!x || !stats) return av2x::fail("av2x_layernorm_stats: null argument");
    if (c != 256) return av2x::fail("av2x_layernorm_stats: c=%d unsupported (256)", c);
    if (n_tokens < 0 || n_tokens > (1ll << 31) - 8) return av2x::fail("av2x_layernorm_stats: bad token count");
    hipLaunchKernelGGL(layernorm_stats_kernel, dim3((unsigned)((n_tokens + 15) / 16)), dim3(256), 0, av2x::as_stream(stream),
                       reinterpret_cast<const float4*>(x), reinterpret_cast<float2*>(stats), (int)n_tokens, eps);
    return av2x::check_launch("layernorm_stats_kernel");
}

extern "C" int av2x_layernorm(const float* x, const float* gamma, const float* beta, float* y, int64_t n_tokens, int32_t c,
                              float eps, av2x_stream_t stream) {
    return av2x_layernorm_act(x, gamma, beta, y, n_tokens, c, eps, 0, stream);
}

extern "C" int av2x_layernorm_act(const float* x, const float* gamma, const float* beta, float* y, int64_t n_tokens,
                                  int32_t c, float eps, int32_t relu, av2x_stream_t stream) {
    if (n_tokens == 0) return 0;
    if (!x || !gamma || !beta || !y) return av2x::fail("av2x_layernorm: null argument");
    if (n_tokens < 0 || n_tokens > (1ll << 31) - 8) return av2x::fail("av2x_layernorm: bad token count");
    const dim3 grid((unsigned)((n_tokens + 3) / 4)), block(256);
    hipStream_t st = av2x::as_stream(stream);
    auto X = reinterpret_cast<const float4*>(x);
    auto G = reinterpret_cast<const float4*>(gamma);
    auto Bt = reinterpret_cast<const float4*>(beta);
    auto Yp = reinterpret_cast<float4*>(y);
    switch (c) {
        case 256: hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, st, X, G, Bt, Yp, (int)n_tokens, eps, relu); break;
        case 512: hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, st, X, G, Bt, Yp, (int)n_tokens, eps, relu); break;
        default: return av2x::fail("av2x_layernorm: c=%d unsupported (256/512)", c);
    }
    return av2x::check_launch("layernorm_kernel");
}

extern "C" int av2x_fax_attention(const float* qkv, const float* bias_table, float* out, int32_t n_agents_padded,
                                  int32_t n_valid, int32_t h, int32_t w, int32_t window, int32_t heads, int32_t dim_head,
                                  int32_t grid_partition, av2x_stream_t stream) {
    if (!qkv || !bias_table || !out) return av2x::fail("av2x_fax_attention: null argument");
    if (dim_head != DH) return av2x::fail("av2x_fax_attention: dim_head=%d unsupported (32)", dim_head);
    if (n_valid < 1 || n_valid > n_agents_padded || window < 1 || h % window || w % window || heads < 1)
        return av2x::fail("av2x_fax_attention: bad sizes (L=%d valid=%d h=%d w=%d ws=%d)", n_agents_padded, n_valid, h, w, window);
    FaxParams p;
    p.qkv = qkv; p.table = bias_table; p.out = out;
    p.L = n_agents_padded; p.n_valid = n_valid; p.H = h; p.W = w; p.ws = window; p.heads = heads; p.grid = grid_partition & 1;
    p.scale = 1.0f / sqrtf((float)dim_head);
    const int Tk = n_valid * window * window;
    const int tab_n = (2 * n_agents_padded - 1) * (2 * window - 1) * (2 * window - 1);
    const int T = n_agents_padded * window * window;
    // ws = 4: one wave per (window, head), everything in registers.  With more than 4 valid agents the 8-key-tile variant needs
    // the whole register file (1 wave / SIMD) and only ties the workgroup-per-window kernel: it is used up to 4 valid agents
    // (test hook: grid_partition bit 4 forces it for any count).
    static const bool no_wave = getenv("AV2X_FAX_NO_WAVE") != nullptr;     // probe: always the workgroup-per-window kernel
    if (T <= 128 && window == 4 && !(grid_partition & 14) && (n_valid <= 4 || (grid_partition & 16)) && !(no_wave && !(grid_partition & 16))) {
        const int tab_s = (tab_n + 63) & ~63;
        const size_t lds_w0 = (size_t)4 * ((heads + 3) / 4) * tab_s * sizeof(float);
        if (lds_w0 <= 48 * 1024) {
            const int nwin = (h / 4) * (w / 4);
            const dim3 grid(nwin < 2048 ? nwin : 2048);
            static const bool force8 = getenv("AV2X_FAX_NV8") != nullptr;      // debug probe (tools/micro/pipe_t32.py)
            // Round 4: this kernel returned wrong rows while its waves shared a SIMD with v_mfma_f32_32x32x16_bf16 waves of another stream.
            // Cause (tools/micro/coreside.py, guard.hip): its bias add compiled to v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0],
            // and a packed-fp32 op with the second source's OP_SEL bit set is disturbed by bf16 MFMAs of a co-resident wave on gfx950.
            // This file is now compiled without packed-fp32 instructions and build.py's lint_isa() rejects the pattern library-wide.
            // AV2X_FAX_OWN_CU=1 (probe) makes the kernel ask for 124 KB of LDS so that no split-3 workgroup can join it on a CU.
            static const bool own = getenv("AV2X_FAX_OWN_CU") != nullptr;
            const size_t lds_w = own ? (size_t)124 * 1024 : lds_w0;
            static av2x::LdsLimit lim_w4, lim_w8;
            lim_w4.ensure(reinterpret_cast<const void*>(&fax_attention_wave_kernel<4>), lds_w);
            lim_w8.ensure(reinterpret_cast<const void*>(&fax_attention_wave_kernel<8>), lds_w);
            if (n_valid <= 4 && !force8) hipLaunchKernelGGL(fax_attention_wave_kernel<4>, grid, dim3(256), lds_w, av2x::as_stream(stream), p);
            else hipLaunchKernelGGL(fax_attention_wave_kernel<8>, grid, dim3(256), lds_w, av2x::as_stream(stream), p);
            return av2x::check_launch("fax_attention_wave_kernel");
        }
    }
    if (T <= 128 && window == 4 && !(grid_partition & 6) && (grid_partition & 32) && tab_n <= 1024) {
        // bit 5: split-3 operands on the bf16 matrix cores (the engine's x3 mode; fp32-accurate products, not the bits of the fp32-input MFMA kernel)
        const int TkP = (Tk + 31) & ~31;
        const size_t lds_3 = (size_t)3 * TkP * KROW3 * 2 + (size_t)3 * DH * (TkP + 8) * 2 + ((size_t)((tab_n + 3) & ~3) + 128) * sizeof(float);
        static av2x::LdsLimit lim_attr_3;
        lim_attr_3.ensure(reinterpret_cast<const void*>(&fax_attention_x3_kernel), lds_3);
        hipLaunchKernelGGL(fax_attention_x3_kernel, dim3((h / window) * (w / window)), dim3(256), lds_3, av2x::as_stream(stream), p);
        return av2x::check_launch("fax_attention_x3_kernel");
    }
    if (T <= 128 && window == 4 && !(grid_partition & 6)) {   // ws = 4: S^T form, P stays in registers (test hook: bit 3)
        const int TkP = (Tk + 31) & ~31;
        const size_t lds_4 = ((size_t)TkP * KLD + (size_t)TkP * VLD4 + ((tab_n + 3) & ~3) + 128) * sizeof(float);
        static av2x::LdsLimit lim_attr_4;
        lim_attr_4.ensure(reinterpret_cast<const void*>(&fax_attention_mfma4_kernel), lds_4);
        hipLaunchKernelGGL(fax_attention_mfma4_kernel, dim3((h / window) * (w / window)), dim3(256), lds_4, av2x::as_stream(stream), p);
        return av2x::check_launch("fax_attention_mfma4_kernel");
    }
    if (T <= 128 && !(grid_partition & 2)) {  // generic-window MFMA path (bit 1 forces the VALU reference kernel, bit 2 this one: tests)
        const int TkP = (Tk + 31) & ~31;
        const size_t lds_m = ((size_t)TkP * KLD + (size_t)TkP * DH + ((tab_n + 3) & ~3) + TkP + 256 + 4 * 32 * PLD) * sizeof(float);
        static av2x::LdsLimit lim_attr_m;
        lim_attr_m.ensure(reinterpret_cast<const void*>(&fax_attention_mfma_kernel), lds_m);
        hipLaunchKernelGGL(fax_attention_mfma_kernel, dim3((h / window) * (w / window)), dim3(256), lds_m, av2x::as_stream(stream), p);
        return av2x::check_launch("fax_attention_mfma_kernel");
    }
    p.grid = grid_partition & 1;
    int nw = 4;                                  // waves per workgroup = heads whose K / V are resident at once
    while (nw > 1 && (size_t)nw * (2 * Tk * DH + tab_n) * sizeof(float) > 160 * 1024) nw >>= 1;
    const size_t lds = (size_t)nw * (2 * Tk * DH + tab_n) * sizeof(float);
    if (lds > 160 * 1024) return av2x::fail("av2x_fax_attention: %zu B of LDS needed (> 160 KiB): too many valid agents", lds);
    static av2x::LdsLimit lim_attr;
    lim_attr.ensure(reinterpret_cast<const void*>(&fax_attention_kernel), lds);
    hipLaunchKernelGGL(fax_attention_kernel, dim3((h / window) * (w / window)), dim3(64 * nw), lds, av2x::as_stream(stream), p);
    return av2x::check_launch("fax_attention_kernel");
}

extern "C" int av2x_agent_mean(const float* x, float* y, int32_t n_agents, int64_t elems_per_agent, av2x_stream_t stream) {
    if (!x || !y) return av2x::fail("av2x_agent_mean: null argument");
    if (n_agents < 1 || elems_per_agent <= 0 || elems_per_agent % 4) return av2x::fail("av2x_agent_mean: bad sizes");
    const size_t n4 = (size_t)elems_per_agent / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(agent_mean_kernel, dim3((unsigned)blocks), dim3(256), 0, av2x::as_stream(stream),
                       reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), n4, n_agents);
    return av2x::check_launch("agent_mean_kernel");
}

