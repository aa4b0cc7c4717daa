// Shared by conv_igemm.hip (fp32-input MFMA) and conv_igemm_bf16.hip (bf16 MFMA, "AMP" mode): kernel parameters
// and the fused epilogue.  Everything lives in an anonymous namespace (one copy per translation unit).
#pragma once
#include <type_traits>

#include "av2x_common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvParams {
    const float* in;
    const float* w;
    const float* scale;
    const float* shift;
    const float* res;  // optional residual, same layout as out (AV2X_CONV mode only)
    float* out;
    int H, W, Cin, in_ctot, in_coff;
    int Ho, Wo, HoWo;
    int Cout, CoutP, out_ctot, out_coff;
    int ks, stride, pad, relu, mode, up;
    int M, tiles_n, cchunks, steps;
    unsigned in_bytes, w_bytes;
    unsigned long long out_bytes;   // whole output tensor
    int in16, out16;                // conv_igemm_bf16 only: input / output activations stored as bf16
    // stream-K (SK kernels only): workgroup g first computes the whole tiles g, g + G, ... < sk_dp (data-parallel part:
    // sk_dp = the largest multiple of the grid size G that fits), then its share of the REMAINDER tiles sk_dp .. tiles-1,
    // whose (tile, K-step) iteration space of sk_total iterations is cut into contiguous ranges of sk_per; partial
    // accumulators of cut tiles go to ws (see conv_fixup_f32).  Fewer tiles than workgroups: sk_dp = 0, pure stream-K.
    int sk_per, sk_total, sk_dp;
    float* ws;
};

constexpr int BK = 32;
constexpr int LDA = 36;

// Epilogue shared by the GEMM kernels and the stream-K fix-up: folded BN / bias, activation, residual, and the NHWC-slice /
// deconv-scatter / NCHW addressing.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
//
// In all three modes the element offset is SEPARABLE, off(m, n) = rowoff(m) + coloff(n):
//   AV2X_CONV    m out_ctot                                        +  out_coff + co
//   AV2X_DECONV  ((img Ho up + ho up) Wo up + wo up) out_ctot      +  (di Wo up + dj) out_ctot + out_coff + co     (n = (di up + dj) Cout + co)
//   AV2X_NCHW    img Cout Ho Wo + ho Wo + wo                       +  co Ho Wo
// so the integer divisions are done once per accumulator ROW and once per COLUMN of the lane (they used to sit in the innermost loop:
// a 128 x 64 eight-wave tile spent more instructions there than in its K loop for the short-K layers), and the stores are buffer
// stores with 32-bit byte offsets whose invalid elements (row >= M, column >= cout) get an out-of-range offset the hardware drops.
// The epilogue is straight-line code executed once per tile from a cold instruction cache: size is cost (see conv_wino4.inc).
// Values are formed exactly as before: bit-identical results.
template <int MT, int NT, bool OUT16 = false>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, const f32x16 (&acc)[MT][NT], int mw, int nw, int lane) {
    const int li = lane & 31, lh = lane >> 5;
    // the descriptor is based at the wave's first row (CONV) / first image (DECONV, NCHW) so that the 32-bit offsets stay small
    // whatever the size of the whole tensor (the 2304-wide QKV of 8 agents is 2.6 GB); its range ends with the tensor
    const int mwu = __builtin_amdgcn_readfirstlane(mw);
    const int img0 = p.mode == AV2X_CONV ? 0 : mwu / p.HoWo;
    const size_t base_elem = p.mode == AV2X_CONV ? (size_t)mwu * p.out_ctot
                             : p.mode == AV2X_DECONV ? (size_t)img0 * p.HoWo * p.up * p.up * p.out_ctot : (size_t)img0 * p.Cout * p.HoWo;
    // OUT16: the output tensor holds bf16 (AMP mode with bf16 activations, conv_igemm_bf16 only): same element offsets, 2-byte stores
    constexpr unsigned ESZ = OUT16 ? 2u : 4u;
    const unsigned long long left = p.out_bytes > base_elem * ESZ ? p.out_bytes - base_elem * ESZ : 0ull;
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(p.out) + base_elem * ESZ, 0,
                                                                          (unsigned)(left < 0x7fffffffull ? left : 0x7fffffffull), 0x00020000);
    constexpr unsigned BAD = 0x80000000u;
    // ---- columns
    unsigned coloff[NT];
    float sc[NT], sh[NT];
    int cco[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const int n = nw + c * 32 + li;  // GEMM column
        int co = n, ij = 0;
        if (p.mode == AV2X_DECONV) { ij = n / p.Cout; co = n - ij * p.Cout; }
        const bool nok = (p.mode == AV2X_DECONV) ? (n < p.CoutP) : (n < p.Cout);
        sc[c] = (nok && p.scale) ? p.scale[co] : 1.f;
        sh[c] = nok ? p.shift[co] : 0.f;
        cco[c] = co;
        unsigned o;
        if (p.mode == AV2X_CONV) o = (unsigned)(p.out_coff + co);
        else if (p.mode == AV2X_DECONV) {
            const int di = ij / p.up, dj = ij - di * p.up;
            o = (unsigned)((di * (p.Wo * p.up) + dj) * p.out_ctot + p.out_coff + co);
        } else o = (unsigned)(co * p.HoWo);
        coloff[c] = nok ? o * ESZ : BAD;
    }
    const bool relu1 = p.relu == 1;
    // ---- rows: the 16 rows of a lane in a row tile are m0 + {0,1,2,3, 8,..,11, 16,.., 24,..} (m0 = mw + 32 a + 4 lh).
    // Two copies of the loop nest: the common case (no residual, ReLU or nothing) without any of the other activations' code
    auto rows = [&](auto simple_c) {
        constexpr bool SIMPLE = decltype(simple_c)::value;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            int m = mw + a * 32 + 4 * lh;
            int img = 0, ho = 0, wo = 0;
            if (p.mode != AV2X_CONV) {
                img = m / p.HoWo;
                const int rem = m - img * p.HoWo;
                ho = rem / p.Wo;
                wo = rem - ho * p.Wo;
            }
            // Residual operand (AV2X_CONV only): all 16 x NT values of this row tile are fetched BEFORE the first store.  The in-place
            // residual layers (out == res: `x = f(x) + x` of the transformer blocks) used to serialise -- a load could not be moved above
            // the preceding store to a pointer that may alias it, so every element paid a full memory latency (64 in a row per lane: a
            // 256 -> 256 token Linear with a residual took 390 us, its GEMM 190).  A lane reads exactly the elements it writes, so
            // reading them first is the same program.  Same values, same operations: bit-identical results.
            float rv[16][NT];
            if constexpr (!SIMPLE) {
                if (p.mode == AV2X_CONV && p.res) {
                    const bool gate = p.relu == 4;
                    const unsigned rstride = gate ? (unsigned)p.Cout : (unsigned)p.out_ctot;
                    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<float*>(p.res) + (size_t)mwu * rstride, 0, 0x7ffffffcu, 0x00020000);
                    int mr = m;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const bool mok = mr < p.M;
#pragma unroll
                        for (int c = 0; c < NT; ++c) {
                            const unsigned ro_ = (unsigned)(mr - mwu) * rstride * 4u + (gate ? (unsigned)cco[c] * 4u : coloff[c]);
                            rv[r][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, (mok && coloff[c] != BAD) ? ro_ : BAD, 0, 0));
                        }
                        mr += (r & 3) == 3 ? 5 : 1;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                unsigned ro;
                if (p.mode == AV2X_CONV) ro = (unsigned)(m - mwu) * (unsigned)p.out_ctot;
                else if (p.mode == AV2X_DECONV) ro = (unsigned)((((img - img0) * p.Ho * p.up + ho * p.up) * (p.Wo * p.up) + wo * p.up) * p.out_ctot);
                else ro = (unsigned)((img - img0) * p.Cout * p.HoWo + ho * p.Wo + wo);
                const bool mok = m < p.M;
                const unsigned rowoff = ro * ESZ;
#pragma unroll
                for (int c = 0; c < NT; ++c) {
                    float v = acc[a][c][r] * sc[c] + sh[c];
                    const unsigned off = (mok && coloff[c] != BAD) ? rowoff + coloff[c] : BAD;
                    if constexpr (SIMPLE) {
                        v = relu1 ? fmaxf(v, 0.f) : v;
                    } else {
                        if (p.relu == 1) v = fmaxf(v, 0.f);
                        else if (p.relu == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));  // exact GELU (nn.GELU())
                        else if (p.relu == 3) v = 1.0f / (1.0f + expf(-v));                              // sigmoid (ConvGRU gates)
                        else if (p.relu == 4) v = tanhf(v);   // tanh; with a residual pointer the result is GATED by it (x res, not + res)
                        else if (p.relu == 6) v = v / (1.0f + expf(-v));                                 // swish (EfficientNet MBConv)
                        if (p.mode == AV2X_CONV) {
                            if (p.res && off != BAD) v = (p.relu == 4) ? v * rv[r][c] : v + rv[r][c];
                            if (p.relu == 5) v = fmaxf(v, 0.f);   // ReLU AFTER the residual add (ResNet BasicBlock)
                        }
                    }
                    if constexpr (OUT16) __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (__bf16)v), rout, off, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, off, 0, 0);
                }
                // next row of this lane: +1, or +5 after every fourth
                const int step = (r & 3) == 3 ? 5 : 1;
                m += step;
                if (p.mode != AV2X_CONV) {
                    wo += step;
                    while (wo >= p.Wo) {
                        wo -= p.Wo;
                        if (++ho >= p.Ho) { ho = 0; ++img; }
                    }
                }
            }
        }
    };
    if (!p.res && (p.relu == 0 || p.relu == 1)) rows(std::true_type{});
    else rows(std::false_type{});
}

}  // namespace
