// Backward pieces of the BEV convolutions (first slice of training on the MI355X path):
//   av2x_conv2d_wgrad   dW[co][ci][kh][kw] = sum over output pixels of dY[p][co] * X[p shifted by the tap][ci]
//   av2x_act_backward   dZ = dY * act'(Y) * scale[c]      (ReLU / identity, folded-BN or unit scale)
//   (av2x_channel_sum, db[c] = sum over pixels of dZ[p][c], lives in train.hip with the other per-channel reductions)
// The data gradient needs no kernel of its own: for stride 1 it IS a convolution of dZ with the 180-degree-rotated,
// channel-transposed weights (av2x_conv2d on weights packed that way); for stride 2 the same on the zero-upsampled dZ
// (opencood_iface/autograd.py).
//
// wgrad as a GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32): per tap C[co][ci] = A^T B with A = dY (pixels x
// cout) and B = X shifted (pixels x cin) -- the REDUCTION runs over pixels, which are the slow axis of both NHWC operands,
// so a K-step is 32 pixel ROWS of each (fully coalesced 4*C-byte runs).  Both stages go L2 -> LDS by LDS-DMA in their
// natural [pixel][channel] image; an MFMA lane (i = lane & 31, h = lane >> 5) needs A[k = 2s + h][row i]: one ds_read_b64
// fetches channels 2i, 2i+1 of pixel 2s + h, i.e. the operands of TWO accumulator tiles whose rows are interleaved
// (tile a holds output channels 2i + a) -- a free permutation of the output rows that the store undoes.
// The pixel axis is cut into chunks (one workgroup per (tile, tap, chunk)); the per-chunk partial dW slabs are summed
// in ascending chunk order by a second kernel: deterministic, no atomics.
#include <cstdlib>

#include "av2x_common.hpp"

namespace {

typedef float f32x16b __attribute__((ext_vector_type(16)));
typedef float f32x2b __attribute__((ext_vector_type(2)));

struct WgradParams {
    const float* x;
    const float* dy;
    float* part;      // [chunk][tap][Cout][Cin]
    int H, W, Cin, in_ctot, in_coff, Ho, Wo, HoWo, Cout, dy_ctot, dy_coff, ks, stride, pad;
    int M, chunk, nchunks, tiles_ci;
    unsigned x_bytes, dy_bytes;
};

__device__ __forceinline__ void glds16b(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
#endif
}

// workgroup = 4 waves, output tile TC couts x TC cins (TC = 128: each wave 64 x 64; TC = 64: each wave 32 x 32)
template <int TC>
__global__ __launch_bounds__(256, 1) void conv_wgrad_kernel(const WgradParams p) {
    constexpr int WT = TC / 2;                 // wave tile edge
    constexpr int MT = WT / 32;                // accumulator tiles per wave edge (2 or 1)
    constexpr int ROWB = TC * 4;               // bytes of one pixel row of a stage
    constexpr int STAGE = 32 * ROWB;           // one operand, 32 pixels
    constexpr int PPI = 1024 / ROWB;           // pixels per 1 KiB DMA instruction (2 or 4)
    constexpr int INST = 32 / PPI;             // instructions per operand stage (16 or 8)
    constexpr int LD = INST / 4;               // per wave
    extern __shared__ __attribute__((aligned(1024))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = p.tiles_ci * ((p.Cout + TC - 1) / TC);
    int b = blockIdx.x;
    const int tile = b % tiles; b /= tiles;
    const int tap = b % (p.ks * p.ks);
    const int chunk = b / (p.ks * p.ks);
    const int co0 = (tile / p.tiles_ci) * TC, ci0 = (tile % p.tiles_ci) * TC;
    const int kh = tap / p.ks, kw = tap - kh * p.ks;
    const int p0 = chunk * p.chunk, p1 = min(p0 + p.chunk, p.M);
    const int nst = (p1 - p0 + 31) / 32;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dy_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int lpp = ROWB / 16;                 // lanes per pixel row (32 or 16)
    const int lp = lane / lpp, lq = lane % lpp;

    auto issue = [&](int step, int buf) {
        unsigned char* sa = sm + buf * 2 * STAGE;
        unsigned char* sb = sa + STAGE;
#pragma unroll
        for (int i = 0; i < LD; ++i) {
            const int inst = wave * LD + i;
            const int pix = p0 + step * 32 + inst * PPI + lp;
            unsigned va = OOB, vb = OOB;
            if (pix < p1) {
                const int img = pix / p.HoWo, rem = pix - img * p.HoWo;
                const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                if (co0 + lq * 4 < p.Cout) va = (unsigned)(((size_t)pix * p.dy_ctot + p.dy_coff + co0 + lq * 4) * 4);
                const int hi = ho * p.stride - p.pad + kh, wi = wo * p.stride - p.pad + kw;
                if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W && ci0 + lq * 4 < p.Cin)
                    vb = (unsigned)((((size_t)img * p.H + hi) * p.W + wi) * p.in_ctot + p.in_coff + ci0 + lq * 4) * 4u;
            }
            glds16b(rdy, sa + inst * 1024, va, 0);
            glds16b(rx, sb + inst * 1024, vb, 0);
        }
    };

    f32x16b acc[MT][MT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int c = 0; c < MT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    const int li = lane & 31, lh = lane >> 5;
    const int wr = (wave >> 1) * WT, wc = (wave & 1) * WT;     // wave's corner inside the tile (cout, cin)

    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int s = 0; s < nst; ++s) {
        const int buf = s & 1;
        if (s + 1 < nst) issue(s + 1, buf ^ 1);
        const unsigned char* sa = sm + buf * 2 * STAGE;
        const unsigned char* sb = sa + STAGE;
#pragma unroll 4
        for (int k2 = 0; k2 < 16; ++k2) {                       // 16 MFMA k-pairs = 32 pixels
            const int row = 2 * k2 + lh;
            float av[MT], bv[MT];
            if constexpr (MT == 2) {
                const f32x2b ta = *reinterpret_cast<const f32x2b*>(sa + row * ROWB + (wr + 2 * li) * 4);
                const f32x2b tb = *reinterpret_cast<const f32x2b*>(sb + row * ROWB + (wc + 2 * li) * 4);
                av[0] = ta.x; av[1] = ta.y; bv[0] = tb.x; bv[1] = tb.y;
            } else {
                av[0] = *reinterpret_cast<const float*>(sa + row * ROWB + (wr + li) * 4);
                bv[0] = *reinterpret_cast<const float*>(sb + row * ROWB + (wc + li) * 4);
            }
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int c = 0; c < MT; ++c)
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[c], acc[a][c], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); tile (a, c) row i holds
    // cout wr + MT * i + a, column j holds cin wc + MT * j + c
    float* out = p.part + ((size_t)chunk * p.ks * p.ks + tap) * p.Cout * p.Cin;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int c = 0; c < MT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int co = co0 + wr + MT * i + a, ci = ci0 + wc + MT * li + c;
                if (co < p.Cout && ci < p.Cin) out[(size_t)co * p.Cin + ci] = acc[a][c][r];
            }
}

// 3x3 / stride 1 / pad 1 layers: ONE workgroup accumulates the three kw taps of a tap row kh.  A K-step is a 32-pixel segment
// of ONE output row, so the three taps read the same staged input pixels shifted by one: per step 32 x 128 dY + 34 x 128 X
// (33 KiB) feed 3 x 64 MFMAs per wave -- three times the flops per staged byte of the per-tap kernel, whose L2 -> LDS
// stream (4 TB/s at the matrix cores' rate) was its limit.  Segments past the row end and the halo columns are zero-filled
// by the buffer bounds check.  Tile 128 couts x 128 cins, 192 accumulator registers per lane.
struct Wgrad3Params {
    const float* x;
    const float* dy;
    float* part;      // [chunk][tap = kh * 3 + kw][Cout][Cin]
    int H, W, Cin, in_ctot, in_coff, Cout, dy_ctot, dy_coff;
    int segs, steps_total, chunk_steps, nchunks, tiles_ci;
    unsigned x_bytes, dy_bytes;
};

__global__ __launch_bounds__(256, 2) void conv_wgrad3_kernel(const Wgrad3Params p) {
    constexpr int TC = 128, ROWB = TC * 4;
    constexpr int STAGE_A = 32 * ROWB, STAGE_B = 34 * ROWB, STAGE = STAGE_A + STAGE_B;
    extern __shared__ __attribute__((aligned(1024))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = p.tiles_ci * ((p.Cout + TC - 1) / TC);
    int b = blockIdx.x;
    const int tile = b % tiles; b /= tiles;
    const int kh = b % 3;
    const int chunk = b / 3;
    const int co0 = (tile / p.tiles_ci) * TC, ci0 = (tile % p.tiles_ci) * TC;
    const int t0 = chunk * p.chunk_steps, t1 = min(t0 + p.chunk_steps, p.steps_total);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dy_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int lp = lane >> 5, lq = lane & 31;          // a 1 KiB DMA instruction = 2 pixel rows of 128 channels

    auto issue = [&](int t, int buf) {
        unsigned char* sa = sm + buf * STAGE;
        unsigned char* sb = sa + STAGE_A;
        const int seg = t % p.segs, rowid = t / p.segs;
        const int ho = rowid % p.H, img = rowid / p.H;          // stride 1, pad 1: Ho == H, Wo == W
        const int wo0 = seg * 32;
        const int hi = ho + kh - 1;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int j = wave + 4 * i;                          // 16 dY instructions, then 17 X instructions
            if (j < 16) {
                const int q = 2 * j + lp;
                unsigned va = OOB;
                if (wo0 + q < p.W && co0 + lq * 4 < p.Cout)
                    va = (unsigned)((((size_t)(img * p.H + ho) * p.W + wo0 + q) * p.dy_ctot + p.dy_coff + co0 + lq * 4) * 4);
                glds16b(rdy, sa + j * 1024, va, 0);
            } else if (j < 33) {
                const int e = 2 * (j - 16) + lp;
                const int wi = wo0 - 1 + e;
                unsigned vb = OOB;
                if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W && ci0 + lq * 4 < p.Cin)
                    vb = (unsigned)((((size_t)(img * p.H + hi) * p.W + wi) * p.in_ctot + p.in_coff + ci0 + lq * 4) * 4);
                glds16b(rx, sb + (j - 16) * 1024, vb, 0);
            }
        }
    };

    f32x16b acc[3][2][2];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[k][a][c][r] = 0.f;
    const int li = lane & 31, lh = lane >> 5;
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;

    if (t0 < t1) {
        issue(t0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    for (int t = t0; t < t1; ++t) {
        const int buf = (t - t0) & 1;
        if (t + 1 < t1) issue(t + 1, buf ^ 1);
        const unsigned char* sa = sm + buf * STAGE;
        const unsigned char* sb = sa + STAGE_A;
#pragma unroll 2
        for (int k2 = 0; k2 < 16; ++k2) {
            const int row = 2 * k2 + lh;
            const f32x2b ta = *reinterpret_cast<const f32x2b*>(sa + row * ROWB + (wr + 2 * li) * 4);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const f32x2b tb = *reinterpret_cast<const f32x2b*>(sb + (row + k) * ROWB + (wc + 2 * li) * 4);
                acc[k][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta.x, tb.x, acc[k][0][0], 0, 0, 0);
                acc[k][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta.x, tb.y, acc[k][0][1], 0, 0, 0);
                acc[k][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta.y, tb.x, acc[k][1][0], 0, 0, 0);
                acc[k][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta.y, tb.y, acc[k][1][1], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float* out = p.part + ((size_t)chunk * 9 + kh * 3 + k) * p.Cout * p.Cin;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int co = co0 + wr + 2 * i + a, ci = ci0 + wc + 2 * li;       // columns ci, ci + 1: one 8-byte store
                if (co < p.Cout && ci + 1 < p.Cin) {
                    f32x2b v2;
                    v2.x = acc[k][a][0][r];
                    v2.y = acc[k][a][1][r];
                    *reinterpret_cast<f32x2b*>(out + (size_t)co * p.Cin + ci) = v2;
                } else if (co < p.Cout && ci < p.Cin) {
                    out[(size_t)co * p.Cin + ci] = acc[k][a][0][r];
                }
            }
    }
}

// dw[co][ci][tap] = sum over chunks (ascending) of part[chunk][tap][co][ci].  The launch reads nchunks x taps x cout x cin floats (50 MB for
// every layer of the three-tap kernel: one slab per workgroup) and is HBM / L2 bound: four channels per lane as one 16-byte load, eight
// chunks' loads in flight; the additions stay in ascending chunk order (bit-identical to a scalar loop).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float4* __restrict__ part, int nchunks, int taps, int cout, int cin,
                                                           float* __restrict__ dw) {
    const size_t n4 = (size_t)taps * cout * cin / 4;
    const int cin4 = cin / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4* p = part + i;
        int c = 0;
        for (; c + 8 <= nchunks; c += 8) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = p[(size_t)(c + j) * n4];
#pragma unroll
            for (int j = 0; j < 8; ++j) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
        }
        for (; c < nchunks; ++c) {
            const float4 v = p[(size_t)c * n4];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const int ci = (int)(i % cin4) * 4;
        const size_t r = i / cin4;
        const int co = (int)(r % cout), tap = (int)(r / cout);
        float* o = dw + ((size_t)co * cin + ci) * taps + tap;
        o[0] = s.x; o[taps] = s.y; o[2 * (size_t)taps] = s.z; o[3 * (size_t)taps] = s.w;
    }
}

// dz = dy * act'(y) * scale[c]   (act 0: identity, 1: ReLU -- y is the layer's OUTPUT, y > 0 <=> pre-activation > 0)
__global__ __launch_bounds__(256) void act_backward_kernel(const float4* __restrict__ y, const float4* __restrict__ dy,
                                                           const float* __restrict__ scale, size_t n4, int c4, int act,
                                                           float4* __restrict__ dz) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 g = dy[i];
        if (act == 1) {
            const float4 v = y[i];
            g.x = v.x > 0.f ? g.x : 0.f; g.y = v.y > 0.f ? g.y : 0.f; g.z = v.z > 0.f ? g.z : 0.f; g.w = v.w > 0.f ? g.w : 0.f;
        }
        if (scale) {
            const int c = (int)(i % c4) * 4;
            g.x *= scale[c]; g.y *= scale[c + 1]; g.z *= scale[c + 2]; g.w *= scale[c + 3];
        }
        dz[i] = g;
    }
}

// wp[tap][ci / 4][co (padded to coutp)][ci % 4] of a convolution with `cout` outputs and `cin` inputs, from the nn.Conv2d parameter:
// flipped == 0: w is (cout, cin, k, k); flipped == 1: w is (cin, cout, k, k) and the taps are rotated by 180 degrees -- the data
// gradient's weights.  One float4 of the packing per lane; the weights change every optimiser step, so this runs per layer and step.
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, int cout, int cin, int coutp, int ks, int flipped,
                                                          float4* __restrict__ wp) {
    const int taps = ks * ks, q4 = cin / 4;
    const size_t n = (size_t)taps * q4 * coutp;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int co = (int)(i % coutp);
        const size_t r = i / coutp;
        const int q = (int)(r % q4), t = (int)(r / q4);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (co < cout) {
            const int ts = flipped ? taps - 1 - t : t;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ci = 4 * q + e;
                v[e] = flipped ? w[((size_t)ci * cout + co) * taps + ts] : w[((size_t)co * cin + ci) * taps + ts];
            }
        }
        wp[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// the split-3 bf16 planes conv_igemm_x3p reads, (3, taps, cin / 8, coutp, 8): hi = bf16(w), mid = bf16(w - hi), lo = bf16(w - hi - mid) (both
// subtractions exact in fp32; round to nearest even as torch's conversion), from the fp32 k-quad packing (taps, cin / 4, coutp, 4) --
// packing.to_bf16x3_koct in one launch, for weights that change every optimiser step
__device__ __forceinline__ unsigned bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

__global__ __launch_bounds__(256) void split3_koct_kernel(const float4* __restrict__ wp, int taps, int q4, int coutp, uint2* __restrict__ out) {
    const size_t n = (size_t)taps * q4 * coutp;
    const size_t plane = n;                                   // uint2 (four bf16) per float4, per plane
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int co = (int)(i % coutp);
        const size_t r = i / coutp;
        const int qq = (int)(r % q4), t = (int)(r / q4);
        const float4 v = wp[i];
        const float w[4] = {v.x, v.y, v.z, v.w};
        unsigned hi[4], mi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi[e] = bf16_rne(w[e]);
            const float r1 = w[e] - __uint_as_float(hi[e] << 16);
            mi[e] = bf16_rne(r1);
            lo[e] = bf16_rne(r1 - __uint_as_float(mi[e] << 16));
        }
        const size_t o = (((size_t)t * (q4 / 2) + (qq >> 1)) * coutp + co) * 2 + (qq & 1);
        out[o] = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
        out[plane + o] = make_uint2(mi[0] | (mi[1] << 16), mi[2] | (mi[3] << 16));
        out[2 * plane + o] = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
    }
}

constexpr int kWgradChunk = 2048;   // output pixels per workgroup (64 K-steps of 32)

}  // namespace

// the three-taps-per-workgroup kernel: 3x3 / stride 1 / pad 1 with more than 64 channels on either side
static bool wgrad3_applies(const av2x_conv_desc* d) {
    static const bool off = [] { const char* e = getenv("AV2X_WGRAD3"); return e && e[0] == '0'; }();
    return !off && d->ks == 3 && d->stride == 1 && d->pad == 1 && d->ho == d->h && d->wo == d->w && (d->cin > 64 || d->cout > 64);
}

// K-steps (32-pixel row segments) per workgroup: enough workgroups to fill the chip twice over, at least 8 steps each
static void wgrad3_plan(const av2x_conv_desc* d, int* segs, int* steps_total, int* chunk_steps, int* nchunks) {
    *segs = (d->w + 31) / 32;
    *steps_total = d->n * d->h * *segs;
    const int tiles = ((d->cin + 127) / 128) * ((d->cout + 127) / 128);
    // one round of workgroups (<= one per CU; two extra workgroups would double the launch's duration): as many pixel chunks as
    // fit, at least 8 K-steps each
    static const int target = [] { const char* e = getenv("AV2X_WGRAD3_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();
    int max_chunks = target / (3 * tiles);
    if (max_chunks < 1) max_chunks = 1;
    long long cs = ((long long)*steps_total + max_chunks - 1) / max_chunks;
    if (cs < 8) cs = 8;
    *chunk_steps = (int)cs;
    *nchunks = (*steps_total + *chunk_steps - 1) / *chunk_steps;
}

// per-tap kernel: output pixels per workgroup (a multiple of 32, 256 .. 2048) -- short enough that (tiles x taps x chunks) fills
// the chip when the map is small or the layer is a 1x1 / transposed convolution with few taps
static int wgrad_chunk(const av2x_conv_desc* d, long long M) {
    const int tc = (d->cin > 64 || d->cout > 64) ? 128 : 64;
    const long long tiles = (long long)((d->cin + tc - 1) / tc) * ((d->cout + tc - 1) / tc) * d->ks * d->ks;
    long long chunk = (M * tiles / 384 + 31) / 32 * 32;
    if (chunk < 256) chunk = 256;
    if (chunk > kWgradChunk) chunk = kWgradChunk;
    return (int)chunk;
}

static unsigned wgrad_reduce_blocks(int taps, const av2x_conv_desc* d) {
    const size_t b = ((size_t)taps * d->cout * d->cin / 4 + 255) / 256;
    return (unsigned)(b < 1 ? 1 : b > 2048 ? 2048 : b);
}

extern "C" uint64_t av2x_conv2d_wgrad_workspace_bytes(const av2x_conv_desc* d) {
    if (!d) return 0;
    if (wgrad3_applies(d)) {
        int segs, st, cs, nch;
        wgrad3_plan(d, &segs, &st, &cs, &nch);
        return (uint64_t)nch * 9ull * d->cout * d->cin * 4ull;
    }
    const long long M = (long long)d->n * d->ho * d->wo;
    const int chunk = wgrad_chunk(d, M);
    const long long nch = (M + chunk - 1) / chunk;
    return (uint64_t)nch * d->ks * d->ks * d->cout * d->cin * 4ull;
}

extern "C" int av2x_conv2d_wgrad(const av2x_conv_desc* d, const float* x, const float* dy, void* workspace, float* dw,
                                 av2x_stream_t stream) {
    if (!d || !x || !dy || !workspace || !dw) return av2x::fail("av2x_conv2d_wgrad: null argument");
    if (d->mode != AV2X_CONV) return av2x::fail("av2x_conv2d_wgrad: plain convolutions only (mode AV2X_CONV)");
    if (d->ks != 1 && d->ks != 3 && d->ks != 5 && d->ks != 7) return av2x::fail("av2x_conv2d_wgrad: ks=%d unsupported (1, 3, 5, 7)", d->ks);
    if (d->cin % 4 || d->cout % 4 || d->in_ctot % 4 || d->in_coff % 4 || d->out_ctot % 4 || d->out_coff % 4)
        return av2x::fail("av2x_conv2d_wgrad: channel counts / offsets must be multiples of 4");
    if (d->ho != (d->h + 2 * d->pad - d->ks) / d->stride + 1 || d->wo != (d->w + 2 * d->pad - d->ks) / d->stride + 1)
        return av2x::fail("av2x_conv2d_wgrad: output dims inconsistent with input/stride/pad");
    const long long M = (long long)d->n * d->ho * d->wo;
    if (M <= 0) return av2x::fail("av2x_conv2d_wgrad: empty tensor");
    if (reinterpret_cast<uintptr_t>(workspace) % 16) return av2x::fail("av2x_conv2d_wgrad: the workspace must be 16-byte aligned");
    const unsigned long long xb = (unsigned long long)d->n * d->h * d->w * d->in_ctot * 4ull;
    const unsigned long long yb = (unsigned long long)M * d->out_ctot * 4ull;
    if (xb >= (1ull << 31) || yb >= (1ull << 31)) return av2x::fail("av2x_conv2d_wgrad: tensor exceeds the 2 GiB buffer-descriptor window");
    hipStream_t st3 = av2x::as_stream(stream);
    if (wgrad3_applies(d)) {
        Wgrad3Params q;
        q.x = x; q.dy = dy; q.part = reinterpret_cast<float*>(workspace);
        q.H = d->h; q.W = d->w; q.Cin = d->cin; q.in_ctot = d->in_ctot; q.in_coff = d->in_coff;
        q.Cout = d->cout; q.dy_ctot = d->out_ctot; q.dy_coff = d->out_coff;
        wgrad3_plan(d, &q.segs, &q.steps_total, &q.chunk_steps, &q.nchunks);
        q.tiles_ci = (d->cin + 127) / 128;
        q.x_bytes = (unsigned)xb; q.dy_bytes = (unsigned)yb;
        const int tiles = q.tiles_ci * ((d->cout + 127) / 128);
        const size_t lds = 2 * (32 + 34) * 512;
        static av2x::LdsLimit lim3;
        lim3.ensure(reinterpret_cast<const void*>(&conv_wgrad3_kernel), lds);
        hipLaunchKernelGGL(conv_wgrad3_kernel, dim3(tiles * 3 * q.nchunks), dim3(256), lds, st3, q);
        if (int e = av2x::check_launch("conv_wgrad3_kernel")) return e;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(wgrad_reduce_blocks(9, d)), dim3(256), 0, st3, reinterpret_cast<const float4*>(q.part), q.nchunks, 9,
                           d->cout, d->cin, dw);
        return av2x::check_launch("wgrad_reduce_kernel");
    }
    WgradParams p;
    p.x = x; p.dy = dy; p.part = reinterpret_cast<float*>(workspace);
    p.H = d->h; p.W = d->w; p.Cin = d->cin; p.in_ctot = d->in_ctot; p.in_coff = d->in_coff;
    p.Ho = d->ho; p.Wo = d->wo; p.HoWo = d->ho * d->wo; p.Cout = d->cout; p.dy_ctot = d->out_ctot; p.dy_coff = d->out_coff;
    p.ks = d->ks; p.stride = d->stride; p.pad = d->pad;
    p.M = (int)M; p.chunk = wgrad_chunk(d, M); p.nchunks = (int)((M + p.chunk - 1) / p.chunk);
    p.x_bytes = (unsigned)xb; p.dy_bytes = (unsigned)yb;
    hipStream_t st = av2x::as_stream(stream);
    const int taps = d->ks * d->ks;
    static av2x::LdsLimit lim128, lim64;
    if (d->cin > 64 || d->cout > 64) {
        p.tiles_ci = (d->cin + 127) / 128;
        const int tiles = p.tiles_ci * ((d->cout + 127) / 128);
        const size_t lds = 2 * 2 * 32 * 512;
        lim128.ensure(reinterpret_cast<const void*>(&conv_wgrad_kernel<128>), lds);
        hipLaunchKernelGGL(conv_wgrad_kernel<128>, dim3(tiles * taps * p.nchunks), dim3(256), lds, st, p);
    } else {
        p.tiles_ci = 1;
        const size_t lds = 2 * 2 * 32 * 256;
        lim64.ensure(reinterpret_cast<const void*>(&conv_wgrad_kernel<64>), lds);
        hipLaunchKernelGGL(conv_wgrad_kernel<64>, dim3(taps * p.nchunks), dim3(256), lds, st, p);
    }
    if (int e = av2x::check_launch("conv_wgrad_kernel")) return e;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(wgrad_reduce_blocks(taps, d)), dim3(256), 0, st, reinterpret_cast<const float4*>(p.part), p.nchunks, taps,
                       d->cout, d->cin, dw);
    return av2x::check_launch("wgrad_reduce_kernel");
}

extern "C" int av2x_pack_conv_weight(const float* w, int32_t cout, int32_t cin, int32_t ks, int32_t flipped, float* wp, av2x_stream_t stream) {
    if (!w || !wp) return av2x::fail("av2x_pack_conv_weight: null argument");
    if (cout <= 0 || cin <= 0 || cin % 4 || ks <= 0 || ks > 7) return av2x::fail("av2x_pack_conv_weight: bad sizes (cin %% 4 == 0, ks <= 7)");
    if (reinterpret_cast<uintptr_t>(wp) % 16) return av2x::fail("av2x_pack_conv_weight: wp must be 16-byte aligned");
    const int coutp = (cout + 31) / 32 * 32;
    const size_t n = (size_t)ks * ks * (cin / 4) * coutp;
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, av2x::as_stream(stream), w, cout, cin, coutp, ks, flipped != 0,
                       reinterpret_cast<float4*>(wp));
    return av2x::check_launch("pack_weight_kernel");
}

extern "C" int av2x_split3_koct(const float* packed, int32_t taps, int32_t cin, int32_t coutp, void* planes, av2x_stream_t stream) {
    if (!packed || !planes) return av2x::fail("av2x_split3_koct: null argument");
    if (taps <= 0 || cin <= 0 || cin % 8 || coutp <= 0) return av2x::fail("av2x_split3_koct: bad sizes (cin %% 8 == 0)");
    if (reinterpret_cast<uintptr_t>(packed) % 16 || reinterpret_cast<uintptr_t>(planes) % 8) return av2x::fail("av2x_split3_koct: misaligned buffer");
    const size_t n = (size_t)taps * (cin / 4) * coutp;
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(split3_koct_kernel, dim3((unsigned)blocks), dim3(256), 0, av2x::as_stream(stream), reinterpret_cast<const float4*>(packed), taps, cin / 4,
                       coutp, reinterpret_cast<uint2*>(planes));
    return av2x::check_launch("split3_koct_kernel");
}

extern "C" int av2x_act_backward(const float* y, const float* dy, const float* scale, int64_t rows, int32_t c, int32_t act,
                                 float* dz, av2x_stream_t stream) {
    if (!dy || !dz || (act == 1 && !y)) return av2x::fail("av2x_act_backward: null argument");
    if (act != 0 && act != 1) return av2x::fail("av2x_act_backward: act %d (0 identity, 1 ReLU)", act);
    if (rows <= 0 || c <= 0 || c % 4) return av2x::fail("av2x_act_backward: bad sizes");
    const size_t n4 = (size_t)rows * c / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(act_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, av2x::as_stream(stream),
                       reinterpret_cast<const float4*>(y), reinterpret_cast<const float4*>(dy), scale, n4, c / 4, act,
                       reinterpret_cast<float4*>(dz));
    return av2x::check_launch("act_backward_kernel");
}
