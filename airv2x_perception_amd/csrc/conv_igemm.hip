// Implicit-GEMM 2-D convolution for gfx950 on the fp32-input matrix cores
// (v_mfma_f32_32x32x2_f32: exact f32 FMA chains at the 157 TFLOP/s fp32 matrix rate).
//
// GEMM view:  M = n*ho*wo output pixels, N = output channels, K = taps*cin.
//   A[m][k]  gathered on the fly from the NHWC input (zero outside the image),
//   B[k][n]  pre-packed weights [tap][cin/4][coutp][4] (see include/airv2x_hip.h).
// One 256-thread workgroup (4 wave64) owns a BM x BN tile; every wave owns WM x WN of it as
// (WM/32) x (WN/32) accumulator tiles of 32x32.  K advances 32 input channels of one tap per
// step, double-buffered through LDS with register staging:
//   As[BM][36]   (row padded 32->36 floats: ds_read_b128 of 16 rows hits 16 distinct 16-B
//                 slots because 9 is odd mod 16 -> conflict-free, writes are 128 B runs)
//   Bs[8][BN][4] (k-quad major: a lane's four k values are one ds_read_b128, lanes contiguous)
// Lane (i = lane&31, h = lane>>5) feeds the MFMA with A[i][k] / B[k][i] for k = 4*(2g+h)+j,
// j = 0..3, so both operands of four consecutive MFMAs come from ONE 16-byte LDS read each.
// The workgroup->tile map is XCD-aware (block b runs on XCD b%8): every XCD walks a
// contiguous range of tiles so that halo rows / weights are re-used out of its private L2.
#include "av2x_common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvParams {
    const float* in;
    const float* w;
    const float* scale;
    const float* shift;
    float* out;
    int H, W, Cin, in_ctot, in_coff;
    int Ho, Wo, HoWo;
    int Cout, CoutP, out_ctot, out_coff;
    int ks, stride, pad, relu, mode, up;
    int M, tiles_n, cchunks, steps;
};

constexpr int BK = 32;
constexpr int LDA = 36;

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN), 1) void conv_igemm_f32(const ConvParams p) {
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int NTHR = 64 * (BM / WM) * (BN / WN);  // 256 (4 waves) or 512 (8 waves)
    constexpr int A_ROWS = NTHR / 8;                   // A rows staged per pass (8 threads x 16 B per row)
    constexpr int A_LD = BM / A_ROWS, B_LD = 8 * BN / NTHR;
    static_assert(NTHR == 256 || NTHR == 512, "4 or 8 waves per workgroup");
    static_assert(A_LD >= 1 && B_LD >= 1 && BM % A_ROWS == 0 && (8 * BN) % NTHR == 0, "tile/threads mismatch");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                   // [2][BM*LDA]
    float* Bs = smem + 2 * BM * LDA;    // [2][8*BN*4]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware, bijective remap of the 1-D grid
    const int nb = gridDim.x, b = blockIdx.x;
    const int q8 = nb >> 3, r8 = nb & 7, xcd = b & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
    const int tile_n = swz % p.tiles_n, tile_m = swz / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;

    // ---- per-thread gather bookkeeping for the A tile: row (tid>>3)+32*i, k-quad tid&7
    const int qA = tid & 7;
    int hi0[A_LD], wi0[A_LD], pix0[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int m = m0 + (tid >> 3) + A_ROWS * i;
        if (m < p.M) {
            const int img = m / p.HoWo, rem = m - img * p.HoWo;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            hi0[i] = ho * p.stride - p.pad;
            wi0[i] = wo * p.stride - p.pad;
            pix0[i] = img * p.H * p.W;
        } else {
            hi0[i] = -(1 << 20);  // never inside the image
            wi0[i] = 0;
            pix0[i] = 0;
        }
    }

    // Register staging of the NEXT K-step (branch-free: out-of-image / out-of-range rows read a
    // valid dummy address and are zeroed by a select, so every load is issued unconditionally and
    // stays in flight behind the MFMAs of the current step).
    f32x4 ra[A_LD], rb[B_LD];
    unsigned okmask = 0;  // bit i: row i of this thread's A loads is inside the image (applied at the LDS store)
    const float* __restrict__ gin = p.in + p.in_coff + qA * 4;
    const float* __restrict__ gw = p.w + (size_t)n0 * 4;
    // B tile: float4 #idx of the [8][BN] k-quad-major tile, idx = tid + 256*i -> row idx/BN, col idx%BN
    constexpr int ROWS_PER_PASS = NTHR / BN > 0 ? NTHR / BN : 1;
    const int bk0 = tid / BN, bn0 = (tid % BN) * 4;

#define AV2X_GLOAD(TAP, CC)                                                                             \
    {                                                                                                   \
        const int kh_ = (TAP) / p.ks, kw_ = (TAP) - kh_ * p.ks;                                         \
        okmask = 0;                                                                                     \
        _Pragma("unroll") for (int i = 0; i < A_LD; ++i) {                                              \
            const int hi = hi0[i] + kh_, wi = wi0[i] + kw_;                                             \
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;               \
            const int pix = ok ? (pix0[i] + hi * p.W + wi) : 0;                                         \
            okmask |= (ok ? 1u : 0u) << i;                                                              \
            ra[i] = *reinterpret_cast<const f32x4*>(gin + (size_t)pix * p.in_ctot + (CC)*BK);           \
        }                                                                                               \
        const size_t wrow_ = (size_t)((TAP) * (p.Cin >> 2) + (CC)*8) * p.CoutP * 4;                     \
        _Pragma("unroll") for (int i = 0; i < B_LD; ++i) {                                              \
            rb[i] = *reinterpret_cast<const f32x4*>(gw + wrow_ +                                        \
                                                    (size_t)(bk0 + i * ROWS_PER_PASS) * p.CoutP * 4 + bn0); \
        }                                                                                               \
    }
#define AV2X_LSTORE(BUF)                                                                                \
    {                                                                                                   \
        float* a_ = As + (BUF) * (BM * LDA);                                                            \
        float* b_ = Bs + (BUF) * (8 * BN * 4);                                                          \
        _Pragma("unroll") for (int i = 0; i < A_LD; ++i) {                                              \
            const f32x4 z_ = {0.f, 0.f, 0.f, 0.f};                                                      \
            *reinterpret_cast<f32x4*>(a_ + ((tid >> 3) + A_ROWS * i) * LDA + qA * 4) =                  \
                ((okmask >> i) & 1u) ? ra[i] : z_;                                                      \
        }                                                                                               \
        _Pragma("unroll") for (int i = 0; i < B_LD; ++i)                                                \
            *reinterpret_cast<f32x4*>(b_ + (tid + NTHR * i) * 4) = rb[i];                               \
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    int tap = 0, cc = 0;
    AV2X_GLOAD(tap, cc);
    AV2X_LSTORE(0);
    __syncthreads();

    const int li = lane & 31, lh = lane >> 5;
    for (int s = 0; s < p.steps; ++s) {
        const int buf = s & 1;
        // prefetch step s+1 (the last iteration re-fetches the final tile: harmless, keeps the loop branch-free)
        if (s + 1 < p.steps) {
            if (++cc == p.cchunks) { cc = 0; ++tap; }
        }
        AV2X_GLOAD(tap, cc);
        // pin the loads here: without this fence hipcc sinks them below the MFMAs (next to the
        // ds_write that consumes them) and the whole L2/HBM latency is exposed every K-step
        __builtin_amdgcn_sched_barrier(0);
        const float* Ab = As + buf * (BM * LDA) + (wm0 + li) * LDA + lh * 4;
        const float* Bb = Bs + buf * (8 * BN * 4) + (lh * BN + wn0 + li) * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 fa[MT], fb[NT];
#pragma unroll
            for (int a = 0; a < MT; ++a) fa[a] = *reinterpret_cast<const f32x4*>(Ab + a * 32 * LDA + g * 8);
#pragma unroll
            for (int c = 0; c < NT; ++c) fb[c] = *reinterpret_cast<const f32x4*>(Bb + (g * 2 * BN + c * 32) * 4);
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int c = 0; c < NT; ++c) {
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].x, fb[c].x, acc[a][c], 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].y, fb[c].y, acc[a][c], 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].z, fb[c].z, acc[a][c], 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].w, fb[c].w, acc[a][c], 0, 0, 0);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        AV2X_LSTORE(buf ^ 1);
        __syncthreads();
    }
#undef AV2X_GLOAD
#undef AV2X_LSTORE

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const int n = n0 + wn0 + c * 32 + li;  // GEMM column
        int co = n, ij = 0;
        if (p.mode == AV2X_DECONV) { ij = n / p.Cout; co = n - ij * p.Cout; }
        const bool nok = (p.mode == AV2X_DECONV) ? (n < p.CoutP) : (n < p.Cout);
        const float sc = (nok && p.scale) ? p.scale[co] : 1.f;
        const float sh = nok ? p.shift[co] : 0.f;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (!nok || m >= p.M) continue;
                float v = acc[a][c][r] * sc + sh;
                if (p.relu) v = fmaxf(v, 0.f);
                size_t off;
                if (p.mode == AV2X_CONV) {
                    off = (size_t)m * p.out_ctot + p.out_coff + co;
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

template <int BM, int BN, int WM, int WN>
int launch(const ConvParams& p, hipStream_t st) {
    const int tiles_m = (p.M + BM - 1) / BM;
    ConvParams q = p;
    q.tiles_n = p.CoutP / BN;
    const size_t lds = (size_t)(2 * BM * LDA + 2 * 8 * BN * 4) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f32<BM, BN, WM, WN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_igemm_f32<BM, BN, WM, WN>), dim3(tiles_m * q.tiles_n), dim3(64 * (BM / WM) * (BN / WN)), lds,
                       st, q);
    return av2x::check_launch("conv_igemm_f32");
}

}  // namespace

extern "C" int av2x_conv2d(const av2x_conv_desc* d, const float* in, const float* w, const float* scale,
                           const float* shift, float* out, av2x_stream_t stream) {
    if (!d || !in || !w || !shift || !out) return av2x::fail("av2x_conv2d: null argument");
    if (d->cin % BK != 0) return av2x::fail("av2x_conv2d: cin=%d must be a multiple of %d", d->cin, BK);
    if (d->coutp % 32 != 0 || d->coutp <= 0) return av2x::fail("av2x_conv2d: coutp=%d must be a positive multiple of 32", d->coutp);
    if (d->mode < 0 || d->mode > 2) return av2x::fail("av2x_conv2d: bad mode %d", d->mode);
    if (d->in_coff % 4 || d->in_ctot % 4) return av2x::fail("av2x_conv2d: input channel offset/stride must be multiples of 4");
    ConvParams p;
    p.in = in; p.w = w; p.scale = scale; p.shift = shift; p.out = out;
    p.H = d->h; p.W = d->w; p.Cin = d->cin; p.in_ctot = d->in_ctot; p.in_coff = d->in_coff;
    p.relu = d->relu; p.mode = d->mode; p.up = d->up;
    p.Cout = d->cout; p.CoutP = d->coutp; p.out_ctot = d->out_ctot; p.out_coff = d->out_coff;
    if (d->mode == AV2X_DECONV) {
        if (d->up < 1 || d->cout % 32 != 0 || d->coutp != d->up * d->up * d->cout)
            return av2x::fail("av2x_conv2d: deconv needs cout %% 32 == 0 and coutp == up*up*cout");
        p.ks = 1; p.stride = 1; p.pad = 0; p.Ho = d->h; p.Wo = d->w;
    } else {
        if (d->ks != 1 && d->ks != 3) return av2x::fail("av2x_conv2d: ks=%d unsupported", d->ks);
        p.ks = d->ks; p.stride = d->stride; p.pad = d->pad; p.Ho = d->ho; p.Wo = d->wo;
        if (p.Ho != (d->h + 2 * d->pad - d->ks) / d->stride + 1 || p.Wo != (d->w + 2 * d->pad - d->ks) / d->stride + 1)
            return av2x::fail("av2x_conv2d: output dims %dx%d inconsistent with input/stride/pad", p.Ho, p.Wo);
        if (d->cout > d->coutp) return av2x::fail("av2x_conv2d: cout > coutp");
    }
    p.HoWo = p.Ho * p.Wo;
    const long long M = (long long)d->n * p.HoWo;
    if (M <= 0) return 0;
    if (M > (1ll << 30)) return av2x::fail("av2x_conv2d: too many output pixels");
    p.M = (int)M;
    p.cchunks = p.Cin / BK;
    p.steps = p.ks * p.ks * p.cchunks;
    p.tiles_n = 0;
    hipStream_t st = av2x::as_stream(stream);

    int bm = (d->tile >> 16) & 0x7fff, bn = d->tile & 0x7fff;
    const bool w8 = (d->tile & 0x8000) != 0;  // 8-wave (512-thread) variant of the same tile
    if (d->tile == 0) {
        // Heuristic: the largest tile that still yields >= ~2 workgroups per CU (256 CUs).
        bn = (p.CoutP % 128 == 0) ? 128 : (p.CoutP % 64 == 0 ? 64 : 32);
        bm = 128;
        auto wgs = [&](int BMv, int BNv) { return ((M + BMv - 1) / BMv) * (p.CoutP / BNv); };
        if (bn == 128 && wgs(128, 128) < 512) { bn = 64; }
        if (bn == 64 && wgs(128, 64) < 512) { bm = 64; }
    }
    if (p.CoutP % bn != 0) return av2x::fail("av2x_conv2d: tile BN=%d does not divide coutp=%d", bn, p.CoutP);
    if (w8) {
        if (bm == 128 && bn == 128) return launch<128, 128, 64, 32>(p, st);
        if (bm == 128 && bn == 64) return launch<128, 64, 32, 32>(p, st);
        if (bm == 256 && bn == 128) return launch<256, 128, 64, 64>(p, st);
        return av2x::fail("av2x_conv2d: unsupported 8-wave tile %dx%d", bm, bn);
    }
    if (bm == 128 && bn == 128) return launch<128, 128, 64, 64>(p, st);
    if (bm == 128 && bn == 64) return launch<128, 64, 64, 32>(p, st);
    if (bm == 64 && bn == 64) return launch<64, 64, 32, 32>(p, st);
    if (bm == 128 && bn == 32) return launch<128, 32, 32, 32>(p, st);
    if (bm == 64 && bn == 128) return launch<64, 128, 32, 64>(p, st);
    return av2x::fail("av2x_conv2d: unsupported tile %dx%d", bm, bn);
}
