// The first convolution of the BEV backbone on the SCATTERED pillar canvas (base_bev_backbone.py:30-48: ZeroPad2d(1) + Conv2d(64, 64, 3,
// stride 2) + BatchNorm + ReLU of block 0) as a sparse gather.
//
// At the BASELINE grid the canvas of an agent has 140 800 cells and ~8 000 pillars: 94 % of the input pixels are exact zeros, and the dense
// split-3 GEMM spends 83 us per 4-agent frame (1 100 full workgroups of matrix work) multiplying them.  A zero tap contributes exactly 0 to
// every sum, so an output pixel only needs the taps whose input cell holds a pillar -- told by the OCCUPANCY bytes the counting scatter writes
// (av2x_pillar_vfe_scatter_count, csrc/pillar.hip).  Per output pixel: out[n] = act(scale[n] * sum over the occupied taps (ky, kx) in row-major
// order, channels ascending, of in[2 oy + ky - 1][2 ox + kx - 1][c] * W[ky][kx][c][n] + shift[n]) in fp32 FMAs -- exact products, one rounding
// per accumulate; an output pixel without an occupied tap is act(shift).
//
//   wave = up to 64 consecutive output pixels (as many as one pass of 256 CUs x 16 waves needs): lane l looks up the nine occupancy bytes of pixel o0 + l, the wave writes act(shift) rows for the
//   batch's empty pixels and then walks the occupied ones; lane = output channel, an occupied input pixel is one 256-byte row load (prefetched one output
//   pixel ahead) broadcast through an LDS slot of the wave, the 3 x 3 x 64 x 64 weights (147 KB, the engine's (tap, cin / 4, cout, 4) packing) live in LDS.
//   HBM-side traffic: the occupancy map (0.56 MB), the occupied input pixels (~2 MB per agent) and the output (9 MB per agent) instead of
//   the 36 MB canvas of every agent.
#include "av2x_common.hpp"

namespace {

constexpr int SC_C = 64;            // input and output channels
constexpr int SC_WAVES = 16;

__global__ __launch_bounds__(64 * SC_WAVES, 1) void conv3x3s2_sparse_kernel(const float* __restrict__ in, const unsigned char* __restrict__ occ,
                                                                            const float4* __restrict__ wpk, const float* __restrict__ scale,
                                                                            const float* __restrict__ shift, int relu, float* __restrict__ out,
                                                                            int n, int H, int W, int Ho, int Wo, int BS) {
    extern __shared__ __attribute__((aligned(16))) float4 sc_w[];          // [tap][cin / 4][cout]  (x, y, z, w = the four channels of the quad)
    float* sc_x = reinterpret_cast<float*>(sc_w + 9 * 16 * SC_C);          // [wave][64]: the input row being multiplied
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 9 * 16 * SC_C; i += 64 * SC_WAVES) sc_w[i] = wpk[i];
    const float sc = scale ? scale[lane] : 1.0f, sh = shift[lane];
    const float idle = relu ? fmaxf(sh, 0.f) : sh;                           // an output pixel whose nine taps are all empty
    __syncthreads();

    const int total = n * Ho * Wo;                                          // < 2^31 (checked by the launcher)
    const int batches = (total + BS - 1) / BS;
    for (int bi = blockIdx.x * SC_WAVES + wave; bi < batches; bi += gridDim.x * SC_WAVES) {
        const int o0 = bi * BS;
        const int rows = min(BS, total - o0);
        // ---- lane l < rows: the occupied taps of output pixel o0 + l (bit t = 3 ky + kx) and the offset of its input window
        unsigned mask = 0;
        int ibase = 0;                                                      // element offset of input pixel (2 oy - 1, 2 ox - 1) of its image
        if (lane < rows) {
            const int o = o0 + lane;
            const int ox = o % Wo, r = o / Wo;
            const int oy = r % Ho, img = r / Ho;
            const unsigned char* ob = occ + (long long)img * H * W;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = 2 * oy + ky - 1;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = 2 * ox + kx - 1;
                    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && ob[iy * W + ix]) mask |= 1u << (3 * ky + kx);
                }
            }
            ibase = ((img * H + 2 * oy - 1) * W + 2 * ox - 1) * SC_C;     // < 2^31 elements (checked by the launcher)
        }
        unsigned long long act = __ballot(mask != 0);
        // ---- the rows of the batch's empty pixels: the empty-pixel value (4 rows of 256 B per store instruction of the wave)
        {
            float* ob = out + (long long)o0 * SC_C;
            const float4 v4 = {__shfl(idle, (lane & 15) * 4 + 0), __shfl(idle, (lane & 15) * 4 + 1), __shfl(idle, (lane & 15) * 4 + 2),
                               __shfl(idle, (lane & 15) * 4 + 3)};
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int row = 4 * j + (lane >> 4);
                if (row < rows && !((act >> row) & 1ull)) *reinterpret_cast<float4*>(ob + row * SC_C + (lane & 15) * 4) = v4;
            }
        }
        // ---- the occupied pixels of the batch, one after the other; lane = output channel.  The (up to nine) occupied input pixels of an
        // output pixel are fetched as one 256-byte row each (lane = input channel), those of the NEXT occupied output pixel while the current
        // one is computed; a row reaches the FMAs through a 256-byte LDS slot of the wave (broadcast reads of four channels).
        float* xs = sc_x + wave * SC_C;
        float cur[9], nxt[9];
        auto fetch = [&](float (&dst)[9], int l) __attribute__((always_inline)) {
            const unsigned m = (unsigned)__builtin_amdgcn_readlane((int)mask, l);
            const float* ib = in + __builtin_amdgcn_readlane(ibase, l) + lane;
#pragma unroll
            for (int t = 0; t < 9; ++t) dst[t] = ((m >> t) & 1u) ? ib[((t / 3) * W + t % 3) * SC_C] : 0.f;
        };
        if (act) fetch(cur, __builtin_ctzll(act));
        while (act) {
            const int l = __builtin_ctzll(act);
            act &= act - 1;
            if (act) fetch(nxt, __builtin_ctzll(act));
            const unsigned m = (unsigned)__builtin_amdgcn_readlane((int)mask, l);
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (!((m >> t) & 1u)) continue;                               // wave-uniform
                xs[lane] = cur[t];
                __builtin_amdgcn_wave_barrier();
                const float4* wt = sc_w + t * 16 * SC_C + lane;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float4 x4 = *reinterpret_cast<const float4*>(xs + 4 * q);
                    const float4 w4 = wt[q * SC_C];
                    acc = fmaf(x4.x, w4.x, acc);
                    acc = fmaf(x4.y, w4.y, acc);
                    acc = fmaf(x4.z, w4.z, acc);
                    acc = fmaf(x4.w, w4.w, acc);
                }
                __builtin_amdgcn_wave_barrier();
            }
            float v = acc * sc + sh;
            if (relu) v = fmaxf(v, 0.f);
            out[(long long)(o0 + l) * SC_C + lane] = v;
#pragma unroll
            for (int t = 0; t < 9; ++t) cur[t] = nxt[t];
        }
    }
}

}  // namespace

extern "C" int av2x_conv3x3s2_sparse(const float* in, const uint8_t* occupancy, const float* w_packed, const float* scale, const float* shift,
                                     int32_t relu, float* out, int32_t n, int32_t h, int32_t w, int32_t cin, int32_t cout, av2x_stream_t stream) {
    if (n == 0) return 0;
    if (!in || !occupancy || !w_packed || !shift || !out) return av2x::fail("av2x_conv3x3s2_sparse: null argument");
    if (cin != SC_C || cout != SC_C) return av2x::fail("av2x_conv3x3s2_sparse: built for 64 -> 64 channels (got %d -> %d)", cin, cout);
    if (n < 0 || h <= 0 || w <= 0) return av2x::fail("av2x_conv3x3s2_sparse: bad sizes");
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    if ((long long)n * h * w * SC_C >= (1ll << 31)) return av2x::fail("av2x_conv3x3s2_sparse: input of more than 2^31 elements");
    // one pass of 256 CUs x 16 waves where the map allows it: a wave takes up to 64 consecutive output pixels
    const int total = n * ho * wo;
    int bs = (total + 256 * SC_WAVES - 1) / (256 * SC_WAVES);
    bs = bs < 8 ? 8 : bs > 64 ? 64 : bs;
    const int batches = (total + bs - 1) / bs;
    int blocks = (batches + SC_WAVES - 1) / SC_WAVES;
    if (blocks > 256) blocks = 256;
    const size_t lds = (size_t)9 * 16 * SC_C * sizeof(float4) + (size_t)SC_WAVES * SC_C * sizeof(float);
    static av2x::LdsLimit lim;
    lim.ensure(reinterpret_cast<const void*>(&conv3x3s2_sparse_kernel), lds);
    hipLaunchKernelGGL(conv3x3s2_sparse_kernel, dim3((unsigned)blocks), dim3(64 * SC_WAVES), lds, av2x::as_stream(stream), in, occupancy,
                       reinterpret_cast<const float4*>(w_packed), scale, shift, relu, out, n, h, w, ho, wo, bs);
    return av2x::check_launch("conv3x3s2_sparse_kernel");
}
