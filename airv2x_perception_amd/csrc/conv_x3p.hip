// Pipelined split-3 implicit GEMM (tile flag 0x0400 | 0x1000): fp32-accurate products on the bf16 matrix cores for the 1x1 /
// strided / transposed convolutions and the token Linears -- every layer the Winograd kernels do not take.
//
// Same arithmetic as conv_igemm_bf16x3 (conv_igemm_bf16.inc): every fp32 operand is hi + mid + lo bf16 terms (weights split at
// pack time, opencood_iface/packing.py to_bf16x3_koct; the A tile between its staging registers and LDS), six partial products per
// 16 k on v_mfma_f32_32x32x16_bf16 in the same order (lo.hi', hi.lo', mid.mid', mid.hi', hi.mid', hi.hi'), K walked in the same
// order: BIT-IDENTICAL results to that kernel (tests/test_gpu_x3p.py).  What changes is the pipeline around the MFMAs, built to
// the budget the Winograd split-3 kernel taught (profiles/r04_pmc_sq_conv_wino_x3_*.txt: one wave per SIMD issues one instruction
// per four cycles, the texture path moves 64 B/clk/CU):
//   * K step = 16 channels of one tap; two LDS stages of 24 KB (A: 3 planes x 128 rows x 16 k bf16, k-oct major so that a fragment
//     read is 16 contiguous bytes per row; B: 3 planes x 16 k x BN) -> three workgroups of 128 x 128 fit a CU's LDS, two its
//     registers: two waves per SIMD, one barrier per step;
//   * the weights' planes go L2 -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, no VGPRs, no VALU), three 1 KB pieces per wave
//     and step; the A tile is prefetched two steps ahead into registers (two 16-byte loads per thread), split with nine VALU
//     instructions per value pair and stored with three ds_write_b64 -- 36 VALU per wave and step beside 24 MFMAs;
//   * a wave owns 64 x 64 (BN = 128) or 64 x 32 (BN = 64) of the tile: per 16 k it reads 12 (9) fragments of 1 KB for 24 (12) MFMAs.
// Per CU and step that is 40 KB through the texture path and 96 KB of LDS reads per 1536 matrix cycles: the matrix cores bound it.
#include <cstdlib>
#include <cstring>

#include "conv_common.hpp"

namespace {

typedef __bf16 p3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 p3_bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void p3_glds16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
#endif
}

#ifdef AV2X_X3P_NOBARRIER       // timing experiment only (races): the per-step workgroup barrier left out
#define P3_STEP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define P3_STEP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

// LayerNorm of the A rows applied while they are loaded (LN = true: token Linears whose input is nn.LayerNorm(x) over the 256 input channels --
// 1x1, stride 1, whole rows): stats = (mean, rstd) per row from layernorm_stats_kernel, gamma / beta in LDS; (x - mean) * rstd * gamma + beta is
// the expression of layernorm_kernel, so the operand the split sees has the bits of the materialised LayerNorm output.
struct X3pLn {
    const float* stats;
    const float* gamma;
    const float* beta;
};

template <int BN, bool LN>
__global__ __launch_bounds__(256, 2) void conv_igemm_x3p(const ConvParams p, const X3pLn ln) {
    constexpr int BM = 128, MT = 2, NT = BN / 64;          // 4 waves as 2 x 2; wave tile 64 x (BN / 2)
    constexpr int WN = BN / 2;
    constexpr int AOS = BM * 16 + 64;                       // bytes per (plane, k-oct) of A: [row][8 bf16] + pad (store conflicts)
    constexpr int A_PL = 2 * AOS, A_BYTES = 3 * A_PL;
    constexpr int BOS = BN * 16;                            // bytes per (plane, k-oct) of B: [col][8 bf16]
    constexpr int B_PL = 2 * BOS, B_BYTES = 3 * B_PL;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int NB_INST = 6 * BN / 64;                    // 1 KB LDS-DMA pieces of B per step
    extern __shared__ __attribute__((aligned(64))) unsigned char smem_p3[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = gridDim.x, b = blockIdx.x;
    const int q8 = nb >> 3, r8 = nb & 7, xcd = b & 7;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * WN;
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const int li = lane & 31, lh = lane >> 5;

    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // ---- A side: thread -> rows (tid >> 2) and (tid >> 2) + 64, k quad tid & 3 (four lanes read the 64 contiguous bytes of a row)
    const int qA = tid & 3;
    int hi0[2], wi0[2], pix0[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + (tid >> 2) + 64 * i;
        if (m < p.M) {
            const int img = m / p.HoWo, rem = m - img * p.HoWo;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            hi0[i] = ho * p.stride - p.pad;
            wi0[i] = wo * p.stride - p.pad;
            pix0[i] = img * p.H * p.W;
        } else {
            hi0[i] = -(1 << 20);
            wi0[i] = 0;
            pix0[i] = 0;
        }
    }
    float lmean[2] = {0.f, 0.f}, lrstd[2] = {0.f, 0.f};
    float* lgb = reinterpret_cast<float*>(smem_p3 + 2 * STAGE);      // LN: gamma[Cin] | beta[Cin] behind the two stages
    if constexpr (LN) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + (tid >> 2) + 64 * i;
            if (m < p.M) {
                lmean[i] = ln.stats[2 * (size_t)m];
                lrstd[i] = ln.stats[2 * (size_t)m + 1];
            }
        }
        for (int k = tid; k < p.Cin; k += 256) {
            lgb[k] = ln.gamma[k];
            lgb[p.Cin + k] = ln.beta[k];
        }
        // (visible to every wave after the prologue's first barrier, before the first lstore_a that reads them ... see below)
    }
    unsigned voffA[2];
    auto tap_offsets = [&](int tp) {
        const int kh = tp / p.ks, kw = tp - kh * p.ks;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hi = hi0[i] + kh, wi = wi0[i] + kw;
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            voffA[i] = ok ? (unsigned)(((pix0[i] + hi * p.W + wi) * p.in_ctot + p.in_coff + qA * 4) * 4) : OOB;
        }
    };
    // ---- B side: DMA piece j of a step = (plane, k-oct, 64-column run); this wave issues pieces wave, wave + 4, ...
    const unsigned plane_bytes = p.w_bytes / 3;
    const int c16 = p.Cin >> 4;                             // 16-channel steps per tap
    const int nst = p.ks * p.ks * c16;
    int tap = 0, cc = 0, issued = 0;
    tap_offsets(0);
    auto advance = [&]() {
        if (issued + 1 < nst) {
            ++issued;
            if (++cc == c16) {
                cc = 0;
                ++tap;
                tap_offsets(tap);
            }
        }
    };
    auto issue_b = [&](int buf) {   // the (tap, cc) step -> LDS stage buf
        unsigned char* st = smem_p3 + buf * STAGE + A_BYTES;
        const unsigned sb = (unsigned)((tap * (p.Cin >> 3) + cc * 2) * p.CoutP * 16);
#pragma unroll
        for (int j0 = 0; j0 < NB_INST; j0 += 4) {
            const int j = j0 + wave;
            if (NB_INST % 4 == 0 || j < NB_INST) {
                const int pl = j / (NB_INST / 3), r = j % (NB_INST / 3), oct = r / (BN / 64), run = r % (BN / 64);
                p3_glds16(rwt, st + pl * B_PL + oct * BOS + run * 1024,
                          (unsigned)((oct * p.CoutP + n0 + run * 64 + lane) * 16), sb + pl * plane_bytes);
            }
        }
    };
    f32x4 ra[2][2];
    int rk[2] = {0, 0};                                                // LN: first channel of this thread's quad in the step held by ra[.]
    auto gload_a = [&](f32x4 (&r)[2], int& k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) r[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, voffA[i], cc * 64, 0));
        k0 = cc * 16 + qA * 4;
    };
    const unsigned wA = (unsigned)((qA >> 1) * AOS + (tid >> 2) * 16 + (qA & 1) * 8);
    auto lstore_a = [&](const f32x4 (&r)[2], int buf, int k0) {
        unsigned char* st = smem_p3 + buf * STAGE + wA;
        f32x4 g4 = {1.f, 1.f, 1.f, 1.f}, b4 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (LN) {
            g4 = *reinterpret_cast<const f32x4*>(lgb + k0);
            b4 = *reinterpret_cast<const f32x4*>(lgb + p.Cin + k0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x4 v = r[i];
            if constexpr (LN) {
                v[0] = (v[0] - lmean[i]) * lrstd[i] * g4[0] + b4[0];
                v[1] = (v[1] - lmean[i]) * lrstd[i] * g4[1] + b4[1];
                v[2] = (v[2] - lmean[i]) * lrstd[i] * g4[2] + b4[2];
                v[3] = (v[3] - lmean[i]) * lrstd[i] * g4[3] + b4[3];
            }
#ifdef AV2X_X3P_NOSPLIT      // timing experiment only (wrong results): what the kernel costs without the hi / mid / lo split
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 raw = {__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y)};
            const p3_bf16x4 h0 = __builtin_bit_cast(p3_bf16x4, raw), h1 = h0, h2 = h0;
#else
            const p3_bf16x4 h0 = __builtin_convertvector(v, p3_bf16x4);
            const f32x4 r1 = v - __builtin_convertvector(h0, f32x4);
            const p3_bf16x4 h1 = __builtin_convertvector(r1, p3_bf16x4);
            const f32x4 r2 = r1 - __builtin_convertvector(h1, f32x4);
            const p3_bf16x4 h2 = __builtin_convertvector(r2, p3_bf16x4);
#endif
            *reinterpret_cast<p3_bf16x4*>(st + i * 1024) = h0;
            *reinterpret_cast<p3_bf16x4*>(st + i * 1024 + A_PL) = h1;
            *reinterpret_cast<p3_bf16x4*>(st + i * 1024 + 2 * A_PL) = h2;
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    const unsigned rA = (unsigned)(lh * AOS + (wm0 + li) * 16);
    const unsigned rB = (unsigned)(A_BYTES + lh * BOS + (wn0 + li) * 16);
    auto compute = [&](int buf) {
#ifdef AV2X_X3P_NOLDS        // timing experiment only (wrong results): every step reads the fragments of stage 0's first rows -- same instruction
        (void)buf;               // count, but the loads hit the same few LDS lines (no bank / bandwidth pressure)
        const unsigned char* st = smem_p3 - rA + (threadIdx.x & 63) * 16;
#else
        const unsigned char* st = smem_p3 + buf * STAGE;
#endif
        p3_bf16x8 fa[3][MT], fb[3][NT];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int a = 0; a < MT; ++a) fa[pl][a] = *reinterpret_cast<const p3_bf16x8*>(st + rA + pl * A_PL + a * 512);
#pragma unroll
            for (int c = 0; c < NT; ++c) fb[pl][c] = *reinterpret_cast<const p3_bf16x8*>(st + rB + pl * B_PL + c * 512);
        }
#define P3_MM(PA, PB)                                                                                     \
    _Pragma("unroll") for (int a = 0; a < MT; ++a) _Pragma("unroll") for (int c = 0; c < NT; ++c)         \
        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA][a], fb[PB][c], acc[a][c], 0, 0, 0);
        P3_MM(2, 0) P3_MM(0, 2) P3_MM(1, 1) P3_MM(1, 0) P3_MM(0, 1) P3_MM(0, 0)
#undef P3_MM
    };

    // prologue: step 0 -> stage 0; the A tile of step 1 in flight
    issue_b(0);
    gload_a(ra[0], rk[0]);
    advance();
    gload_a(ra[1], rk[1]);
    if constexpr (LN) __syncthreads();                         // gamma / beta are in LDS
    lstore_a(ra[0], 0, rk[0]);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");          // the DMA of step 0 has landed (the two loads of step 1 may still fly)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    for (int s = 0; s < nst; s += 2) {
        // even step s: stage 0 holds it; ra[1] holds step s + 1
        issue_b(1);                                           // step s + 1 (its (tap, cc) is the current one)
        advance();
        const int k1 = rk[1];
        gload_a(ra[0], rk[0]);                                // step s + 2
        compute(0);
        lstore_a(ra[1], 1, k1);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        P3_STEP_BARRIER();
        if (s + 1 < nst) {
            issue_b(0);                                       // step s + 2
            advance();
            const int k0 = rk[0];
            gload_a(ra[1], rk[1]);                            // step s + 3
            compute(1);
            lstore_a(ra[0], 0, k0);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            P3_STEP_BARRIER();
        }
    }
#ifdef AV2X_X3P_NOEPI           // timing experiment only: one store per lane instead of the 64-value epilogue
    {
        float keep = 0.f;       // every accumulator tile stays live (no MFMA may be optimised away)
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int c = 0; c < NT; ++c) keep += acc[a][c][(a * NT + c) & 15];
        if (keep == 123456.f) p.out[threadIdx.x] = keep;
    }
#else
    conv_epilogue<MT, NT>(p, acc, m0 + wm0, n0 + wn0, lane);
#endif
}

template <int BN>
int launch_x3p(const ConvParams& p, const X3pLn& ln, hipStream_t st) {
    constexpr int BM = 128;
    const int tiles_m = (p.M + BM - 1) / BM;
    ConvParams q = p;
    q.tiles_n = p.CoutP / BN;
    const size_t lds = 2ull * (3 * 2 * (BM * 16 + 64) + 3 * 2 * BN * 16);
    if (ln.stats) {
        hipLaunchKernelGGL((conv_igemm_x3p<BN, true>), dim3(tiles_m * q.tiles_n), dim3(256), lds + 2ull * p.Cin * sizeof(float), st, q, ln);
        return av2x::check_launch("conv_igemm_x3p<LN>");
    }
    hipLaunchKernelGGL((conv_igemm_x3p<BN, false>), dim3(tiles_m * q.tiles_n), dim3(256), lds, st, q, ln);
    return av2x::check_launch("conv_igemm_x3p");
}

}  // namespace

namespace av2x {

// called by av2x_conv2d* (conv_igemm.hip) for tile flag 0x0400 | 0x1000 with the validated ConvParams of that translation unit
// (same header, same layout; passed as bytes because the type lives in an anonymous namespace)
int x3p_dispatch(const void* conv_params, size_t bytes, int bm, int bn, hipStream_t st, const float* ln_stats, const float* ln_gamma,
                 const float* ln_beta) {
    ConvParams p;
    if (bytes != sizeof(ConvParams)) return fail("av2x_conv2d: internal parameter block mismatch");
    std::memcpy(&p, conv_params, sizeof(p));
    if (p.Cin % 16) return fail("av2x_conv2d: the pipelined split-3 tiles need cin %% 16 == 0 (cin=%d)", p.Cin);
    if (bm != 128 || (bn != 128 && bn != 64)) return fail("av2x_conv2d: the pipelined split-3 tiles are 128x128 and 128x64 (tile %dx%d)", bm, bn);
    if (p.CoutP % bn) return fail("av2x_conv2d: tile BN=%d does not divide coutp=%d", bn, p.CoutP);
    const X3pLn ln = {ln_stats, ln_gamma, ln_beta};
    if (ln_stats) {
        if (!ln_gamma || !ln_beta) return fail("av2x_conv2d_ln: null gamma / beta");
        if (p.ks != 1 || p.stride != 1 || p.pad != 0 || p.mode != AV2X_CONV || p.in_ctot != p.Cin || p.in_coff != 0 || p.Cin > 1024)
            return fail("av2x_conv2d_ln: LayerNorm in the operand load needs a 1x1 / stride-1 layer over whole rows (in_ctot == cin, in_coff == 0)");
    }
    return bn == 128 ? launch_x3p<128>(p, ln, st) : launch_x3p<64>(p, ln, st);
}

}  // namespace av2x
