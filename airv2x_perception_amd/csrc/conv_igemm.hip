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
#include "conv_common.hpp"

namespace av2x {
int wino_x3_dispatch(const av2x_conv_desc* d, const float* in, const void* u, const float* scale, const float* shift,
                     const float* residual, float* out, hipStream_t st);   // conv_wino_x3.hip
int wino4_x3_dispatch(const av2x_conv_desc* d, const float* in, const void* u, const float* scale, const float* shift,
                      const float* residual, float* out, hipStream_t st);   // conv_wino4_x3.hip
int x3p_dispatch(const void* conv_params, size_t bytes, int bm, int bn, hipStream_t st, const float* ln_stats = nullptr,
                 const float* ln_gamma = nullptr, const float* ln_beta = nullptr);   // conv_x3p.hip
}

// Test-only: -DAV2X_ABLATE=<bits> (tools/micro/ablate.sh) removes pieces of the prefetch-2 main loop to see what each costs
// (1 global loads, 2 LDS stores, 4 barriers, 8 LDS fragment reads after the first step).  Results are garbage, only the
// timing means something; the product library is never built with it.
#ifndef AV2X_ABLATE
#define AV2X_ABLATE 0
#endif

namespace {

// Stream-K fix-up: tile t was cut between workgroups g_lo..g_hi of the SK kernel; their raw accumulators
// (same per-thread layout as the kernel: [slot][(a*NT+c)*16+r][tid]) are summed in ascending K order
// (deterministic) and sent through the normal epilogue.  Slot 2g holds the partial of g's FIRST
// segment (the tile its range starts in), slot 2g+1 the partial of its last one.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN)) void conv_fixup_f32(const ConvParams p) {
    constexpr int MT = WM / 32, NT = WN / 32, WAVES_N = BN / WN, NTHR = 64 * (BM / WM) * (BN / WN);
    const int rt = blockIdx.x, tid = threadIdx.x;   // rt: index among the remainder (stream-K) tiles
    const int tile = p.sk_dp + rt;
    const int it0 = rt * p.steps, it1 = it0 + p.steps - 1;
    const int g_lo = it0 / p.sk_per, g_hi = it1 / p.sk_per;
    if (g_lo == g_hi) return;  // computed whole by one workgroup, already written
    f32x16 acc[MT][NT];
    for (int g = g_lo; g <= g_hi; ++g) {
        const int slot = ((g * p.sk_per) / p.steps == rt) ? 2 * g : 2 * g + 1;
        const float* wsp = p.ws + (size_t)slot * (BM * BN) + tid;
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int c = 0; c < NT; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = wsp[((a * NT + c) * 16 + r) * NTHR];
                    acc[a][c][r] = (g == g_lo) ? v : acc[a][c][r] + v;
                }
    }
    const int wave = tid >> 6;
    const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
    conv_epilogue<MT, NT>(p, acc, tile_m * BM + (wave / WAVES_N) * WM, tile_n * BN + (wave % WAVES_N) * WN, tid & 63);
}

// MODE 0: data-parallel (one workgroup = one output tile)
// MODE 1: stream-K (equal ranges of tiles x K-steps per persistent workgroup + conv_fixup_f32)
// MODE 2: persistent, whole tiles: every workgroup walks a contiguous range of tiles as ONE flat stream of
//         K-steps -- the loads of the next tile's first steps are already in flight while the current tile's
//         epilogue (an HBM write burst) runs, and there is no per-tile launch / prologue bubble.  This is the
//         schedule for short-K GEMMs (1x1 convs, transformer Linears: 8-12 K-steps per tile); results are
//         bit-identical to MODE 0.
template <int BM, int BN, int WM, int WN, bool DEEP, int MODE>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN), 1) void conv_igemm_f32(const ConvParams p) {
    constexpr bool SK = MODE == 1, PERSIST = MODE == 2;
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
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
    // iteration range of this workgroup: one whole tile, or (stream-K) sk_per iterations that may
    // cover the tail of one tile, whole tiles, and the head of another
    int it = SK ? swz * p.sk_per : swz * p.steps;
    const int it_end = SK ? min(it + p.sk_per, p.sk_total) : it + p.steps;
    bool first_seg = true;
    int dp_t = swz;   // SK: next whole tile of the data-parallel part (dp_t < p.sk_dp), before the stream-K range
    if (SK && !(dp_t < p.sk_dp || it < it_end)) return;
    // PERSIST: tiles [pt0, pt1) of this workgroup (sk_total = number of tiles)
    const int pq = PERSIST ? p.sk_total / nb : 0, pr = PERSIST ? p.sk_total - pq * nb : 0;
    const int pt0 = swz * pq + min(swz, pr), pt1 = pt0 + pq + (swz < pr ? 1 : 0);
    if (PERSIST && pt0 >= pt1) return;
    do {
    const bool dp = SK && dp_t < p.sk_dp;
    const int rt = SK ? it / p.steps : 0;   // remainder-tile index of the stream-K range
    const int tile = PERSIST ? pt0 : (SK ? (dp ? dp_t : p.sk_dp + rt) : swz);
    const int ks0 = (SK && !dp) ? it - rt * p.steps : 0;
    // number of K-steps of this pass: a tile, a stream-K segment, or (PERSIST) the whole flat stream
    const int nst = PERSIST ? (pt1 - pt0) * p.steps : ((SK && !dp) ? min(p.steps - ks0, it_end - it) : p.steps);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;

    // ---- per-thread gather bookkeeping for the A tile: row (tid>>3)+32*i, k-quad tid&7
    const int qA = tid & 7;
    int hi0[A_LD], wi0[A_LD], pix0[A_LD];
    unsigned voffA[A_LD], voffB[B_LD];
    // B tile: float4 #idx of the [8][BN] k-quad-major tile, idx = tid + NTHR*i -> row idx/BN, col idx%BN
    constexpr int ROWS_PER_PASS = NTHR / BN > 0 ? NTHR / BN : 1;
#define AV2X_TILE_SETUP(TILE)                                                                           \
    {                                                                                                   \
        const int tm0_ = ((TILE) / p.tiles_n) * BM, tn0_ = ((TILE) % p.tiles_n) * BN;                   \
        _Pragma("unroll") for (int i = 0; i < A_LD; ++i) {                                              \
            const int m = tm0_ + (tid >> 3) + A_ROWS * i;                                               \
            if (m < p.M) {                                                                              \
                const int img = m / p.HoWo, rem = m - img * p.HoWo;                                     \
                const int ho = rem / p.Wo, wo = rem - ho * p.Wo;                                        \
                hi0[i] = ho * p.stride - p.pad;                                                         \
                wi0[i] = wo * p.stride - p.pad;                                                         \
                pix0[i] = img * p.H * p.W;                                                              \
            } else {                                                                                    \
                hi0[i] = -(1 << 20); /* never inside the image */                                       \
                wi0[i] = 0;                                                                             \
                pix0[i] = 0;                                                                            \
            }                                                                                           \
        }                                                                                               \
        _Pragma("unroll") for (int i = 0; i < B_LD; ++i)                                                \
            voffB[i] = (unsigned)(((tid / BN + i * ROWS_PER_PASS) * p.CoutP + tn0_ + (tid % BN)) * 16); \
    }
    AV2X_TILE_SETUP(tile)

    // Register staging of the NEXT K-step through BUFFER loads (T8): the descriptor is built from
    // kernel arguments (provably wave-uniform -> no waterfall loops), the per-row byte offset lives in
    // a VGPR that only changes when the tap changes, the channel-chunk / weight-row offset is a scalar
    // (soffset).  Rows outside the image (or beyond M) get voffset = 2^31 >= num_records, for which the
    // hardware returns zeros: no address select, no branch, no zero-fill select at the LDS store, and
    // the per-step VALU work of the gather is gone.
    // Two register sets: with DEEP the tile of step s+2 is requested while step s computes (prefetch
    // distance 2), else only set 0 is used (distance 1).
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    f32x4 ra0[A_LD], rb0[B_LD], ra1[A_LD], rb1[B_LD];
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

#define AV2X_TAPOFF(TAP)                                                                                \
    {                                                                                                   \
        const int kh_ = (TAP) / p.ks, kw_ = (TAP) - kh_ * p.ks;                                         \
        _Pragma("unroll") for (int i = 0; i < A_LD; ++i) {                                              \
            const int hi = hi0[i] + kh_, wi = wi0[i] + kw_;                                             \
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;               \
            voffA[i] = ok ? (unsigned)(((pix0[i] + hi * p.W + wi) * p.in_ctot + p.in_coff + qA * 4) * 4) : OOB; \
        }                                                                                               \
    }
#define AV2X_GLOAD(ra, rb, TAP, CC)                                                                     \
    {                                                                                                   \
        if ((CC) == 0) AV2X_TAPOFF(TAP) /* new tap: refresh the per-row offsets (uniform branch) */     \
        const unsigned sa_ = (unsigned)((CC)*BK * 4);                                                   \
        _Pragma("unroll") for (int i = 0; i < A_LD; ++i) ra[i] =                                        \
            __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, voffA[i], sa_, 0));    \
        const unsigned sb_ = (unsigned)(((TAP) * (p.Cin >> 2) + (CC)*8) * p.CoutP * 16);                \
        _Pragma("unroll") for (int i = 0; i < B_LD; ++i) rb[i] =                                        \
            __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rwt, voffB[i], sb_, 0));    \
    }
#define AV2X_LSTORE(ra, rb, BUF)                                                                        \
    {                                                                                                   \
        float* a_ = As + (BUF) * (BM * LDA);                                                            \
        float* b_ = Bs + (BUF) * (8 * BN * 4);                                                          \
        _Pragma("unroll") for (int i = 0; i < A_LD; ++i)                                                \
            *reinterpret_cast<f32x4*>(a_ + ((tid >> 3) + A_ROWS * i) * LDA + qA * 4) = ra[i];           \
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

    // (tap, channel chunk) of the NEXT K-step to request; clamps at the last step of the segment
    int tap = SK ? ks0 / p.cchunks : 0, cc = SK ? ks0 - tap * p.cchunks : 0;
    int issued = 0;
    if (SK && cc != 0) AV2X_TAPOFF(tap)
    int ltile = tile;            // PERSIST: tile the loader is fetching; (ctile, cstep): tile / K-step being computed
    int ctile = tile, cstep = 0;
    const int taps = p.ks * p.ks;
#define AV2X_ADVANCE()                                                                  \
    {                                                                                   \
        if (issued + 1 < nst) {                                                         \
            ++issued;                                                                   \
            if (++cc == p.cchunks) {                                                    \
                cc = 0;                                                                 \
                ++tap;                                                                  \
                if (PERSIST && tap == taps) { /* the loader crosses into the next tile */ \
                    tap = 0;                                                            \
                    ++ltile;                                                            \
                    AV2X_TILE_SETUP(ltile)                                              \
                }                                                                       \
            }                                                                           \
        }                                                                               \
    }
    // PERSIST: after the MFMAs of a tile's last K-step, store it and start the next one with zero accumulators
#define AV2X_TILE_END()                                                                                   \
    {                                                                                                     \
        if (PERSIST && ++cstep == p.steps) {                                                              \
            conv_epilogue<MT, NT>(p, acc, (ctile / p.tiles_n) * BM + wm0, (ctile % p.tiles_n) * BN + wn0, lane); \
            _Pragma("unroll") for (int a = 0; a < MT; ++a) _Pragma("unroll") for (int c = 0; c < NT; ++c)  \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;                         \
            cstep = 0;                                                                                    \
            ++ctile;                                                                                      \
        }                                                                                                 \
    }
    const int li = lane & 31, lh = lane >> 5;
#if AV2X_ABLATE & 8
    f32x4 fa0[MT], fb0[NT], fa1[MT], fb1[NT];   // hoisted: only the first two K-steps read their fragments
#define AV2X_FRAG_DECL
#define AV2X_FRAG_ON (s < 2)
#else
#define AV2X_FRAG_DECL f32x4 fa0[MT], fb0[NT], fa1[MT], fb1[NT];
#define AV2X_FRAG_ON true
#endif
#define AV2X_FRAGS(FA, FB, G)                                                                                       \
    if (AV2X_FRAG_ON) {                                                                                             \
        _Pragma("unroll") for (int a = 0; a < MT; ++a) FA[a] =                                                      \
            *reinterpret_cast<const f32x4*>(Ab + a * 32 * LDA + (G)*8);                                             \
        _Pragma("unroll") for (int c = 0; c < NT; ++c) FB[c] =                                                      \
            *reinterpret_cast<const f32x4*>(Bb + ((G)*2 * BN + c * 32) * 4);                                        \
    }
#define AV2X_MFMAS(FA, FB)                                                                                          \
    {                                                                                                               \
        _Pragma("unroll") for (int a = 0; a < MT; ++a) _Pragma("unroll") for (int c = 0; c < NT; ++c) {             \
            acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[a].x, FB[c].x, acc[a][c], 0, 0, 0);                 \
            acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[a].y, FB[c].y, acc[a][c], 0, 0, 0);                 \
            acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[a].z, FB[c].z, acc[a][c], 0, 0, 0);                 \
            acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[a].w, FB[c].w, acc[a][c], 0, 0, 0);                 \
        }                                                                                                           \
    }
    // fragments of k-group g+1 are requested from LDS before the MFMAs of group g are issued (two
    // fragment sets), so the ds_read latency overlaps the 64-cycle MFMA chain of the previous group
#define AV2X_COMPUTE(BUF)                                                                                           \
    {                                                                                                               \
        const float* Ab = As + (BUF) * (BM * LDA) + (wm0 + li) * LDA + lh * 4;                                      \
        const float* Bb = Bs + (BUF) * (8 * BN * 4) + (lh * BN + wn0 + li) * 4;                                     \
        AV2X_FRAG_DECL                                                                                              \
        AV2X_FRAGS(fa0, fb0, 0);                                                                                    \
        AV2X_FRAGS(fa1, fb1, 1);                                                                                    \
        AV2X_MFMAS(fa0, fb0);                                                                                       \
        AV2X_FRAGS(fa0, fb0, 2);                                                                                    \
        AV2X_MFMAS(fa1, fb1);                                                                                       \
        AV2X_FRAGS(fa1, fb1, 3);                                                                                    \
        AV2X_MFMAS(fa0, fb0);                                                                                       \
        AV2X_MFMAS(fa1, fb1);                                                                                       \
    }

    AV2X_GLOAD(ra0, rb0, tap, cc);
    AV2X_LSTORE(ra0, rb0, 0);
    __syncthreads();
    if constexpr (!DEEP) {
        for (int s = 0; s < nst; ++s) {
            const int buf = s & 1;
            AV2X_ADVANCE();  // the last iteration re-fetches the final tile: harmless, keeps the loop branch-free
            AV2X_GLOAD(ra0, rb0, tap, cc);
            // pin the loads here: without this fence hipcc sinks them below the MFMAs (next to the
            // ds_write that consumes them) and the whole L2/HBM latency is exposed every K-step
            __builtin_amdgcn_sched_barrier(0);
            AV2X_COMPUTE(buf);
            __builtin_amdgcn_sched_barrier(0);
            AV2X_LSTORE(ra0, rb0, buf ^ 1);
            AV2X_TILE_END();
            __syncthreads();
        }
    } else {
        // invariant at the top of an even step s: tile s is in LDS buffer 0, tile s+1 is in flight in set 0
#define ABL_GLOAD(ra, rb) { if (!(AV2X_ABLATE & 1)) AV2X_GLOAD(ra, rb, tap, cc) }
#define ABL_LSTORE(ra, rb, BUF) { if (!(AV2X_ABLATE & 2)) AV2X_LSTORE(ra, rb, BUF) }
#define ABL_SYNC() { if (!(AV2X_ABLATE & 4)) __syncthreads(); }
        AV2X_ADVANCE();
        AV2X_GLOAD(ra0, rb0, tap, cc);
        for (int s = 0; s < nst; s += 2) {
            AV2X_ADVANCE();
            ABL_GLOAD(ra1, rb1);  // tile s+2
            __builtin_amdgcn_sched_barrier(0);
            AV2X_COMPUTE(0);
            __builtin_amdgcn_sched_barrier(0);
            ABL_LSTORE(ra0, rb0, 1);  // tile s+1 -> buffer 1
            AV2X_TILE_END();
            ABL_SYNC();
            if (s + 1 < nst) {
                AV2X_ADVANCE();
                ABL_GLOAD(ra0, rb0);  // tile s+3
                __builtin_amdgcn_sched_barrier(0);
                AV2X_COMPUTE(1);
                __builtin_amdgcn_sched_barrier(0);
                ABL_LSTORE(ra1, rb1, 0);  // tile s+2 -> buffer 0
                AV2X_TILE_END();
                ABL_SYNC();
            }
        }
    }
#undef ABL_GLOAD
#undef ABL_LSTORE
#undef ABL_SYNC
#undef AV2X_COMPUTE
#undef AV2X_FRAGS
#undef AV2X_FRAG_DECL
#undef AV2X_FRAG_ON
#undef AV2X_MFMAS
#undef AV2X_ADVANCE
#undef AV2X_TILE_END
#undef AV2X_TILE_SETUP
#undef AV2X_GLOAD
#undef AV2X_TAPOFF
#undef AV2X_LSTORE

    if (PERSIST) {
        // every tile was stored by AV2X_TILE_END
    } else if (!SK || nst == p.steps) {
        conv_epilogue<MT, NT>(p, acc, m0 + wm0, n0 + wn0, lane);
    } else {
        float* wsp = p.ws + (size_t)(2 * swz + (first_seg ? 0 : 1)) * (BM * BN) + tid;
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int c = 0; c < NT; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) wsp[((a * NT + c) * 16 + r) * NTHR] = acc[a][c][r];
    }
    if (dp) dp_t += nb;
    else { it += nst; first_seg = false; }
    if (SK) __syncthreads();  // the next segment re-fills LDS buffer 0
    } while (SK && (dp_t < p.sk_dp || it < it_end));
}

template <int BM, int BN, int WM, int WN, bool DEEP = false>
int launch(const ConvParams& p, hipStream_t st) {
    const int tiles_m = (p.M + BM - 1) / BM;
    ConvParams q = p;
    q.tiles_n = p.CoutP / BN;
    const size_t lds = (size_t)(2 * BM * LDA + 2 * 8 * BN * 4) * sizeof(float);
    static av2x::LdsLimit lds_limit;
    lds_limit.ensure(reinterpret_cast<const void*>(&conv_igemm_f32<BM, BN, WM, WN, DEEP, 0>), lds);
    hipLaunchKernelGGL((conv_igemm_f32<BM, BN, WM, WN, DEEP, 0>), dim3(tiles_m * q.tiles_n),
                       dim3(64 * (BM / WM) * (BN / WN)), lds, st, q);
    return av2x::check_launch("conv_igemm_f32");
}

// Stream-K schedule of `wgs` persistent workgroups over `tiles` tiles (see ConvParams): fills sk_dp / sk_per / sk_total,
// returns the grid size; *fix_tiles = remainder tiles the fix-up kernel has to visit (0 = none is cut).
inline int sk_schedule(ConvParams& q, int tiles, int wgs, int* fix_tiles) {
    const long long all = (long long)tiles * q.steps;
    if (wgs > all) wgs = (int)all;
    q.sk_dp = (tiles / wgs) * wgs;
    const int rem = tiles - q.sk_dp;
    const long long total = (long long)rem * q.steps;
    q.sk_total = (int)total;
    q.sk_per = 1;
    *fix_tiles = 0;
    int grid = wgs;
    if (total > 0) {
        long long per = (total + wgs - 1) / wgs;
        const long long floor_len = q.steps < 4 ? q.steps : 4;   // no segments shorter than 4 K-steps
        if (per < floor_len) per = floor_len;
        q.sk_per = (int)per;
        const int active = (int)((total + per - 1) / per);
        if (q.sk_dp == 0) grid = active;
        if (per % q.steps != 0) *fix_tiles = rem;
    }
    return grid;
}

// Stream-K launch, then the fix-up of the cut tiles.
template <int BM, int BN, int WM, int WN>
int launch_sk(const ConvParams& p, int wgs, float* ws, unsigned long long ws_bytes, hipStream_t st) {
    const int tiles_m = (p.M + BM - 1) / BM;
    ConvParams q = p;
    q.tiles_n = p.CoutP / BN;
    const int tiles = tiles_m * q.tiles_n;
    if ((long long)tiles * p.steps >= (1ll << 31)) return av2x::fail("av2x_conv2d: stream-K iteration space too large");
    int fix_tiles = 0;
    wgs = sk_schedule(q, tiles, wgs, &fix_tiles);
    q.ws = ws;
    if (!ws || ws_bytes < 2ull * wgs * BM * BN * sizeof(float))
        return av2x::fail("av2x_conv2d: stream-K workspace too small (%llu B, need %llu B)", ws_bytes,
                          2ull * wgs * BM * BN * sizeof(float));
    const size_t lds = (size_t)(2 * BM * LDA + 2 * 8 * BN * 4) * sizeof(float);
    static av2x::LdsLimit lds_limit;
    lds_limit.ensure(reinterpret_cast<const void*>(&conv_igemm_f32<BM, BN, WM, WN, true, 1>), lds);
    constexpr int NTHR = 64 * (BM / WM) * (BN / WN);
    hipLaunchKernelGGL((conv_igemm_f32<BM, BN, WM, WN, true, 1>), dim3(wgs), dim3(NTHR), lds, st, q);
    if (fix_tiles > 0)  // some tile is cut
        hipLaunchKernelGGL((conv_fixup_f32<BM, BN, WM, WN>), dim3(fix_tiles), dim3(NTHR), 0, st, q);
    return av2x::check_launch("conv_igemm_f32 (stream-K)");
}

#include "conv_igemm_bf16.inc"
#include "conv_igemm_glds.inc"
#include "conv_wino.inc"
#include "conv_wino4.inc"
#include "conv_halo_bf16.inc"

}  // namespace

// Persistent launch (MODE 2): `wgs` workgroups, each a contiguous range of whole tiles.
template <int BM, int BN, int WM, int WN>
int launch_persist(const ConvParams& p, int wgs, hipStream_t st) {
    const int tiles_m = (p.M + BM - 1) / BM;
    ConvParams q = p;
    q.tiles_n = p.CoutP / BN;
    const long long tiles = (long long)tiles_m * q.tiles_n;
    if (tiles * p.steps >= (1ll << 31)) return av2x::fail("av2x_conv2d: persistent iteration space too large");
    if (wgs > tiles) wgs = (int)tiles;
    q.sk_total = (int)tiles;
    const size_t lds = (size_t)(2 * BM * LDA + 2 * 8 * BN * 4) * sizeof(float);
    static av2x::LdsLimit lds_limit;
    lds_limit.ensure(reinterpret_cast<const void*>(&conv_igemm_f32<BM, BN, WM, WN, true, 2>), lds);
    hipLaunchKernelGGL((conv_igemm_f32<BM, BN, WM, WN, true, 2>), dim3(wgs), dim3(64 * (BM / WM) * (BN / WN)), lds, st, q);
    return av2x::check_launch("conv_igemm_f32 (persistent)");
}

extern "C" int av2x_conv2d_sk(const av2x_conv_desc* d, const float* in, const float* w, const float* scale,
                              const float* shift, const float* residual, float* out, float* workspace,
                              uint64_t workspace_bytes, av2x_stream_t stream);

extern "C" int av2x_conv2d_res(const av2x_conv_desc* d, const float* in, const float* w, const float* scale,
                               const float* shift, const float* residual, float* out, av2x_stream_t stream) {
    return av2x_conv2d_sk(d, in, w, scale, shift, residual, out, nullptr, 0, stream);
}

extern "C" uint64_t av2x_conv2d_sk_workspace_bytes(int32_t tile, int32_t sk_wgs) {
    const int bm = (tile >> 16) & 0x7fff, bn = tile & 0x01ff;
    return (tile & 0x2000) ? 2ull * (unsigned)sk_wgs * bm * bn * sizeof(float) : 0ull;
}

extern "C" int av2x_conv2d(const av2x_conv_desc* d, const float* in, const float* w, const float* scale,
                           const float* shift, float* out, av2x_stream_t stream) {
    return av2x_conv2d_res(d, in, w, scale, shift, nullptr, out, stream);
}

static int conv2d_impl(const av2x_conv_desc* d, const float* in, const float* w, const float* scale, const float* shift, const float* residual,
                       float* out, float* workspace, uint64_t workspace_bytes, av2x_stream_t stream, const float* ln_stats,
                       const float* ln_gamma, const float* ln_beta);

extern "C" int av2x_conv2d_sk(const av2x_conv_desc* d, const float* in, const float* w, const float* scale,
                              const float* shift, const float* residual, float* out, float* workspace,
                              uint64_t workspace_bytes, av2x_stream_t stream) {
    return conv2d_impl(d, in, w, scale, shift, residual, out, workspace, workspace_bytes, stream, nullptr, nullptr, nullptr);
}

extern "C" int av2x_conv2d_ln(const av2x_conv_desc* d, const float* in, const float* ln_stats, const float* ln_gamma, const float* ln_beta,
                              const float* w, const float* scale, const float* shift, const float* residual, float* out,
                              av2x_stream_t stream) {
    if (!ln_stats || !ln_gamma || !ln_beta) return av2x::fail("av2x_conv2d_ln: null LayerNorm argument");
    if (!d || (d->tile & 0x1400) != 0x1400 || (d->tile & 0x40000000))
        return av2x::fail("av2x_conv2d_ln: only the pipelined split-3 tiles (flag 0x0400 | 0x1000) normalise while they load");
    return conv2d_impl(d, in, w, scale, shift, residual, out, nullptr, 0, stream, ln_stats, ln_gamma, ln_beta);
}

static int conv2d_impl(const av2x_conv_desc* d, const float* in, const float* w, const float* scale, const float* shift, const float* residual,
                       float* out, float* workspace, uint64_t workspace_bytes, av2x_stream_t stream, const float* ln_stats,
                       const float* ln_gamma, const float* ln_beta) {
    if (!d || !in || !w || !shift || !out) return av2x::fail("av2x_conv2d: null argument");
    if (residual && d->mode != AV2X_CONV) return av2x::fail("av2x_conv2d: residual only with mode AV2X_CONV");
    if (d->relu < 0 || d->relu > 6)
        return av2x::fail("av2x_conv2d: relu/activation code %d (0 none, 1 ReLU, 2 GELU, 3 sigmoid, 4 tanh [x residual], 5 ReLU after the residual, 6 swish)", d->relu);
    if (d->relu == 5 && d->mode != AV2X_CONV) return av2x::fail("av2x_conv2d: activation 5 (ReLU after the residual) needs mode AV2X_CONV");
    if ((d->tile & 0x60000400) == 0x60000400)   // Winograd F(4x4,3x3) with split-3 bf16 operands (conv_wino4_x3.hip): `w` = av2x_wino4_x3_pack_weights
        return av2x::wino4_x3_dispatch(d, in, w, scale, shift, residual, out, av2x::as_stream(stream));
    if ((d->tile & 0x60000000) == 0x60000000)   // Winograd F(4x4,3x3): `w` is the transformed packing of av2x_wino4_pack_weights
        return wino4_dispatch(d, in, w, scale, shift, residual, out, av2x::as_stream(stream));
    if ((d->tile & 0x40000400) == 0x40000400)   // Winograd F(2x2,3x3) with split-3 bf16 operands (conv_wino_x3.hip): `w` = av2x_wino_x3_pack_weights
        return av2x::wino_x3_dispatch(d, in, w, scale, shift, residual, out, av2x::as_stream(stream));
    if (d->tile & 0x40000000)   // Winograd F(2x2,3x3): `w` is the transformed packing of av2x_wino_pack_weights
        return wino_dispatch(d, in, w, scale, shift, residual, out, av2x::as_stream(stream));
    const bool x3p = (d->tile & 0x1400) == 0x1400 && !(d->tile & 0x40000000);   // pipelined split-3 tiles step 16 channels
    if (d->cin % BK != 0 && !(x3p && d->cin % 16 == 0)) return av2x::fail("av2x_conv2d: cin=%d must be a multiple of %d", d->cin, BK);
    if (d->coutp % 32 != 0 || d->coutp <= 0) return av2x::fail("av2x_conv2d: coutp=%d must be a positive multiple of 32", d->coutp);
    if (d->mode < 0 || d->mode > 2) return av2x::fail("av2x_conv2d: bad mode %d", d->mode);
    if (d->in_coff % 4 || d->in_ctot % 4) return av2x::fail("av2x_conv2d: input channel offset/stride must be multiples of 4");
    ConvParams p;
    p.in = in; p.w = w; p.scale = scale; p.shift = shift; p.res = residual; p.out = out;
    p.H = d->h; p.W = d->w; p.Cin = d->cin; p.in_ctot = d->in_ctot; p.in_coff = d->in_coff;
    p.relu = d->relu; p.mode = d->mode; p.up = d->up;
    p.Cout = d->cout; p.CoutP = d->coutp; p.out_ctot = d->out_ctot; p.out_coff = d->out_coff;
    if (d->mode == AV2X_DECONV) {
        if (d->up < 1 || d->cout % 32 != 0 || d->coutp != d->up * d->up * d->cout)
            return av2x::fail("av2x_conv2d: deconv needs cout %% 32 == 0 and coutp == up*up*cout");
        p.ks = 1; p.stride = 1; p.pad = 0; p.Ho = d->h; p.Wo = d->w;
    } else {
        if (d->ks < 1 || d->ks > 7) return av2x::fail("av2x_conv2d: ks=%d unsupported (1 .. 7)", d->ks);
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
    p.sk_per = 0; p.sk_total = 0; p.sk_dp = 0; p.ws = nullptr;
    p.in16 = d->act16 & 1; p.out16 = (d->act16 >> 1) & 1;
    if (d->act16 & ~3) return av2x::fail("av2x_conv2d: act16=%d (bit 0: bf16 input activations, bit 1: bf16 output)", d->act16);
    if (d->act16 && !(d->tile & 0x0800)) return av2x::fail("av2x_conv2d: bf16 activations need the bf16 matrix-core tiles (tile flag 0x0800)");
    if (p.out16 && residual) return av2x::fail("av2x_conv2d: no residual operand with a bf16 output");
    const unsigned long long in_bytes = (unsigned long long)d->n * d->h * d->w * d->in_ctot * (p.in16 ? 2ull : 4ull);
    const unsigned long long w_bytes = (unsigned long long)p.ks * p.ks * p.Cin * p.CoutP * 4ull;
    if (in_bytes >= (1ull << 31) || w_bytes >= (1ull << 31))
        return av2x::fail("av2x_conv2d: input (%llu B) or weights (%llu B) exceed the 2 GiB buffer-descriptor window", in_bytes, w_bytes);
    p.in_bytes = (unsigned)in_bytes;
    p.w_bytes = (unsigned)w_bytes;
    // the epilogue addresses the output with 32-bit byte offsets through a buffer descriptor
    const unsigned long long oesz = p.out16 ? 2ull : 4ull;
    const unsigned long long out_bytes = d->mode == AV2X_DECONV ? (unsigned long long)d->n * d->h * d->up * d->w * d->up * d->out_ctot * oesz
                                        : d->mode == AV2X_CONV ? (unsigned long long)M * d->out_ctot * oesz
                                                               : (unsigned long long)M * d->cout * oesz;
    if (d->mode != AV2X_CONV && out_bytes / (unsigned long long)d->n >= (1ull << 30))
        return av2x::fail("av2x_conv2d: one output image (%llu B) exceeds the 1 GiB window of the epilogue's 32-bit offsets", out_bytes / d->n);
    if ((unsigned long long)d->out_ctot * 4ull * 256ull >= (1ull << 30)) return av2x::fail("av2x_conv2d: out_ctot too large");
    p.out_bytes = out_bytes;
    hipStream_t st = av2x::as_stream(stream);

    int bm = (d->tile >> 16) & 0x7fff, bn = d->tile & 0x01ff;
    if (d->tile & 0x0400) {   // split-3: fp32-accurate products from three bf16 terms per operand; w = [3] bf16 planes
        p.w_bytes = (unsigned)(w_bytes / 2 * 3);
        if (d->tile & 0x1000) return av2x::x3p_dispatch(&p, sizeof(p), bm, bn, st, ln_stats, ln_gamma, ln_beta);   // pipelined form (conv_x3p.hip), same bits
        if (p.CoutP % bn != 0) return av2x::fail("av2x_conv2d: tile BN=%d does not divide coutp=%d", bn, p.CoutP);
        const bool w8b = (d->tile & 0x8000) != 0, db3 = (d->tile & 0x4000) != 0;   // 0x4000: second LDS buffer set
#define AV2X_X3(W8, BMv, BNv, WMv, WNv)                                                                   \
        if (w8b == W8 && bm == BMv && bn == BNv)                                                          \
            return db3 ? launch_bf16x3<BMv, BNv, WMv, WNv, true>(p, st) : launch_bf16x3<BMv, BNv, WMv, WNv, false>(p, st);
        AV2X_X3(true, 128, 128, 64, 32)
        AV2X_X3(true, 128, 64, 32, 32)
        AV2X_X3(false, 128, 128, 64, 64)
        AV2X_X3(false, 128, 64, 64, 32)
        AV2X_X3(false, 64, 64, 32, 32)
        AV2X_X3(false, 128, 32, 32, 32)
#undef AV2X_X3
        return av2x::fail("av2x_conv2d: unsupported split-3 tile %dx%d", bm, bn);
    }
    if (d->tile & 0x0800) {   // bf16 matrix-core operands ("AMP" mode): w is the bf16 packing [tap][cin/8][coutp][8]
        p.w_bytes = (unsigned)(w_bytes / 2);
        if (d->tile & 0x10000000) return halo_bf16_dispatch(d, p, residual, st);   // halo-tile direct convolution on bf16 activations
        if (p.CoutP % bn != 0) return av2x::fail("av2x_conv2d: tile BN=%d does not divide coutp=%d", bn, p.CoutP);
        const bool w8b = (d->tile & 0x8000) != 0;
        if (w8b && bm == 128 && bn == 128) return launch_bf16<128, 128, 64, 32, true>(p, st);
        if (w8b && bm == 128 && bn == 64) return launch_bf16<128, 64, 32, 32, true>(p, st);
        if (!w8b && bm == 128 && bn == 128) return launch_bf16<128, 128, 64, 64>(p, st);
        if (!w8b && bm == 128 && bn == 64) return launch_bf16<128, 64, 64, 32>(p, st);
        if (!w8b && bm == 64 && bn == 64) return launch_bf16<64, 64, 32, 32, true>(p, st);
        if (!w8b && bm == 128 && bn == 32) return launch_bf16<128, 32, 32, 32>(p, st);
        return av2x::fail("av2x_conv2d: unsupported bf16 tile %dx%d", bm, bn);
    }
    if (d->tile & 0x0200) {  // LDS-DMA operand path (conv_igemm_glds.inc): 2 LDS stages, or 3 with the prefetch-2 flag
        if (p.CoutP % bn != 0) return av2x::fail("av2x_conv2d: tile BN=%d does not divide coutp=%d", bn, p.CoutP);
        const bool w8g = (d->tile & 0x8000) != 0, s3 = (d->tile & 0x4000) != 0, skg = (d->tile & 0x2000) != 0;
        if (skg && d->sk_wgs <= 0) return av2x::fail("av2x_conv2d: stream-K tile needs sk_wgs > 0");
#define AV2X_GLDS_CASE(W8, BMv, BNv, WMv, WNv)                                                                              \
        if (w8g == W8 && bm == BMv && bn == BNv) {                                                                          \
            if (skg) return s3 ? launch_glds_sk<BMv, BNv, WMv, WNv, 3>(p, d->sk_wgs, workspace, workspace_bytes, st)       \
                               : launch_glds_sk<BMv, BNv, WMv, WNv, 2>(p, d->sk_wgs, workspace, workspace_bytes, st);      \
            return s3 ? launch_glds<BMv, BNv, WMv, WNv, 3>(p, st) : launch_glds<BMv, BNv, WMv, WNv, 2>(p, st);             \
        }
        AV2X_GLDS_CASE(true, 128, 128, 64, 32)
        AV2X_GLDS_CASE(true, 128, 64, 32, 32)
        AV2X_GLDS_CASE(false, 128, 128, 64, 64)
        AV2X_GLDS_CASE(false, 128, 64, 64, 32)
        AV2X_GLDS_CASE(false, 64, 64, 32, 32)
#undef AV2X_GLDS_CASE
        return av2x::fail("av2x_conv2d: unsupported LDS-DMA tile %dx%d", bm, bn);
    }
    if (d->tile & 0x1000) {  // persistent whole-tile schedule: sk_wgs workgroups (prefetch-2 pipeline)
        if (d->sk_wgs <= 0) return av2x::fail("av2x_conv2d: persistent tile needs sk_wgs > 0");
        if (p.CoutP % bn != 0) return av2x::fail("av2x_conv2d: tile BN=%d does not divide coutp=%d", bn, p.CoutP);
        const bool w8p = (d->tile & 0x8000) != 0;
        if (w8p && bm == 128 && bn == 128) return launch_persist<128, 128, 64, 32>(p, d->sk_wgs, st);
        if (w8p && bm == 128 && bn == 64) return launch_persist<128, 64, 32, 32>(p, d->sk_wgs, st);
        if (!w8p && bm == 128 && bn == 128) return launch_persist<128, 128, 64, 64>(p, d->sk_wgs, st);
        if (!w8p && bm == 128 && bn == 64) return launch_persist<128, 64, 64, 32>(p, d->sk_wgs, st);
        if (!w8p && bm == 64 && bn == 64) return launch_persist<64, 64, 32, 32>(p, d->sk_wgs, st);
        if (!w8p && bm == 64 && bn == 128) return launch_persist<64, 128, 32, 64>(p, d->sk_wgs, st);
        return av2x::fail("av2x_conv2d: unsupported persistent tile %dx%d", bm, bn);
    }
    if (d->tile & 0x2000) {  // stream-K: sk_wgs persistent workgroups (always the prefetch-2 pipeline)
        if (d->sk_wgs <= 0) return av2x::fail("av2x_conv2d: stream-K tile needs sk_wgs > 0");
        if (p.CoutP % bn != 0) return av2x::fail("av2x_conv2d: tile BN=%d does not divide coutp=%d", bn, p.CoutP);
        const bool w8s = (d->tile & 0x8000) != 0;
        if (w8s && bm == 128 && bn == 128) return launch_sk<128, 128, 64, 32>(p, d->sk_wgs, workspace, workspace_bytes, st);
        if (w8s && bm == 128 && bn == 64) return launch_sk<128, 64, 32, 32>(p, d->sk_wgs, workspace, workspace_bytes, st);
        if (!w8s && bm == 128 && bn == 128) return launch_sk<128, 128, 64, 64>(p, d->sk_wgs, workspace, workspace_bytes, st);
        if (!w8s && bm == 128 && bn == 64) return launch_sk<128, 64, 64, 32>(p, d->sk_wgs, workspace, workspace_bytes, st);
        if (!w8s && bm == 64 && bn == 64) return launch_sk<64, 64, 32, 32>(p, d->sk_wgs, workspace, workspace_bytes, st);
        if (!w8s && bm == 64 && bn == 128) return launch_sk<64, 128, 32, 64>(p, d->sk_wgs, workspace, workspace_bytes, st);
        return av2x::fail("av2x_conv2d: unsupported stream-K tile %dx%d", bm, bn);
    }
    const bool w8 = (d->tile & 0x8000) != 0;    // 8-wave (512-thread) variant of the same tile
    const bool deep = (d->tile & 0x4000) != 0;  // prefetch distance 2 (two register sets)
    if (d->tile == 0) {
        // Heuristic: the largest tile that still yields >= ~2 workgroups per CU (256 CUs).
        bn = (p.CoutP % 128 == 0) ? 128 : (p.CoutP % 64 == 0 ? 64 : 32);
        bm = 128;
        auto wgs = [&](int BMv, int BNv) { return ((M + BMv - 1) / BMv) * (p.CoutP / BNv); };
        if (bn == 128 && wgs(128, 128) < 512) { bn = 64; }
        if (bn == 64 && wgs(128, 64) < 512) { bm = 64; }
    }
    if (p.CoutP % bn != 0) return av2x::fail("av2x_conv2d: tile BN=%d does not divide coutp=%d", bn, p.CoutP);
    if (deep) {
        if (w8) {
            if (bm == 128 && bn == 128) return launch<128, 128, 64, 32, true>(p, st);
            if (bm == 128 && bn == 64) return launch<128, 64, 32, 32, true>(p, st);
            return av2x::fail("av2x_conv2d: unsupported deep 8-wave tile %dx%d", bm, bn);
        }
        if (bm == 128 && bn == 128) return launch<128, 128, 64, 64, true>(p, st);
        if (bm == 128 && bn == 64) return launch<128, 64, 64, 32, true>(p, st);
        if (bm == 64 && bn == 64) return launch<64, 64, 32, 32, true>(p, st);
        if (bm == 64 && bn == 128) return launch<64, 128, 32, 64, true>(p, st);
        return av2x::fail("av2x_conv2d: unsupported deep tile %dx%d", bm, bn);
    }
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
