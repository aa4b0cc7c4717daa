// Shared by conv_igemm.hip (fp32-input MFMA) and conv_igemm_bf16.hip (bf16 MFMA, "AMP" mode): kernel parameters
// and the fused epilogue.  Everything lives in an anonymous namespace (one copy per translation unit).
#pragma once
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
    // stream-K (SK kernels only): workgroup g first computes the whole tiles g, g + G, ... < sk_dp (data-parallel part:
    // sk_dp = the largest multiple of the grid size G that fits), then its share of the REMAINDER tiles sk_dp .. tiles-1,
    // whose (tile, K-step) iteration space of sk_total iterations is cut into contiguous ranges of sk_per; partial
    // accumulators of cut tiles go to ws (see conv_fixup_f32).  Fewer tiles than workgroups: sk_dp = 0, pure stream-K.
    int sk_per, sk_total, sk_dp;
    float* ws;
};

constexpr int BK = 32;
constexpr int LDA = 36;

// Epilogue shared by the GEMM kernel and the stream-K fix-up: folded BN / bias, activation, residual,
// and the NHWC-slice / deconv-scatter / NCHW addressing.  C/D map of the 32x32 MFMA:
// col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
template <int MT, int NT>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, const f32x16 (&acc)[MT][NT], int mw, int nw, int lane) {
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const int n = nw + c * 32 + li;  // GEMM column
        int co = n, ij = 0;
        if (p.mode == AV2X_DECONV) { ij = n / p.Cout; co = n - ij * p.Cout; }
        const bool nok = (p.mode == AV2X_DECONV) ? (n < p.CoutP) : (n < p.Cout);
        const float sc = (nok && p.scale) ? p.scale[co] : 1.f;
        const float sh = nok ? p.shift[co] : 0.f;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mw + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (!nok || m >= p.M) continue;
                float v = acc[a][c][r] * sc + sh;
                if (p.relu == 1) v = fmaxf(v, 0.f);
                else if (p.relu == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));  // exact GELU (nn.GELU())
                else if (p.relu == 3) v = 1.0f / (1.0f + expf(-v));                              // sigmoid (ConvGRU gates)
                else if (p.relu == 4) v = tanhf(v);   // tanh; with a residual pointer the result is GATED by it (x res, not + res)
                else if (p.relu == 6) v = v / (1.0f + expf(-v));                                 // swish (EfficientNet MBConv)
                size_t off;
                if (p.mode == AV2X_CONV) {
                    off = (size_t)m * p.out_ctot + p.out_coff + co;
                    if (p.res) v = (p.relu == 4) ? v * p.res[(size_t)m * p.Cout + co] : v + p.res[off];
                    if (p.relu == 5) v = fmaxf(v, 0.f);   // ReLU AFTER the residual add (ResNet BasicBlock)
                } else {
                    const int img = m / p.HoWo, rem = m - img * p.HoWo;
                    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                    if (p.mode == AV2X_DECONV) {
                        const int di = ij / p.up, dj = ij - di * p.up;
                        off = ((size_t)(img * p.Ho * p.up + ho * p.up + di) * (p.Wo * p.up) + wo * p.up + dj) * p.out_ctot +
                              p.out_coff + co;
                    } else {  // NCHW
                        off = ((size_t)(img * p.Cout + co) * p.Ho + ho) * p.Wo + wo;
                    }
                }
                p.out[off] = v;
            }
        }
    }
}

}  // namespace
