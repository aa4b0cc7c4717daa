// Winograd F(4x4, 3x3) convolution with fp32-accurate products on the bf16 matrix cores (tile flag 0x60000000 | 0x0400):
// the 36-position algorithm of conv_wino4.inc (Y = A^T [ (G g G^T) . (B^T d B) ] A, 2.25 multiplies per output) with the operand
// handling of conv_wino_x3.hip: every fp32 value of V = B^T d B and U = G g G^T enters the matrix core as three bf16 terms
// (hi + mid + lo = the value to 2^-24) and the six partial products >= 2^-16 are accumulated in fp32 on v_mfma_f32_32x32x16_bf16,
// smallest first.  The transforms stay fp32 (packed v_pk_*_f32 on channel pairs).  Against fp64 the error is that of the fp32
// F(4x4) kernel or below (tests/test_gpu_wino4_x3.py): the products are as exact as fp32 products, the transforms are the same.
//
// Replaces the 3x3 / stride-1 layers of the reference's DownsampleConv (common_modules/downsample_conv.py:8-54) and BaseBEVBackbone
// (base_bev_backbone.py:6-154) that the engine's wino4 rule selects (the 256 -> 256 shrink convolutions at 100 x 352).
//
// Mapping.  A workgroup = 4 waves = 32 tiles (4x4 outputs each) x 64 couts x all 36 positions.  Wave w owns positions 9 w .. 9 w + 8
// for all tiles and both 32-cout blocks: 18 accumulator tiles of 32 x 32 = 288 registers (16 tiles in the accumulation registers,
// two in ordinary VGPRs), one wave per SIMD, one workgroup per CU.  Every V element is read by exactly ONE wave, so the hi / mid /
// lo split is done on the A fragment in registers, once per element (x3_split_step, 36 VALU instructions per fragment spread over
// the 12 MFMAs that use the previous one).
//   V (input side)   thread (tile tl = tid / 8, channel pair cg = tid % 8) gathers its 6x6 patch with 8-byte buffer loads (the eight
//                    lanes of a tile read the 64 contiguous bytes of a pixel), transforms it and writes fp32 V to LDS as
//                    [stage 2][pos 36][k quad 4][tile 32][4]  (2 x 76.5 KB): a lane's A fragment half is one ds_read_b128.
//                    Patch offsets are NOT kept per element (36 registers): six per-column offsets + the row offset as the scalar
//                    offset of the load; rows that can fall outside the image (all but row 1) cost one v_cndmask per load.
//   U (weight side)  split at pack time (av2x_wino4_x3_pack_weights, fp64 inside) into [pos 36][cin/16][plane 3][k half 2][coutp][8]
//                    bf16; a wave's B fragments of a position (2 cout blocks x 3 planes x 16 bytes per lane) come straight from L2
//                    into registers one position ahead (two register sets).
//   K loop           a chunk = 16 input channels = 108 steps of ONE MFMA + a few side instructions, pinned by scheduling barriers:
//                    steps 0..23 B^T d of the next chunk's patch, 24..83 (B^T d) B row by row with its six LDS stores, ONE barrier at
//                    step 83, the patch of the chunk after next re-gathered row by row as soon as a row's registers are stored.
//                    9 positions per wave is odd, so the register double buffers alternate per chunk: the loop body is two chunks.
//   output           the 36 M tiles go through LDS (the V stages, free by then) one 32-cout block at a time; wave w finalises
//                    accumulator rows 4 w .. 4 w + 3: Y = A^T M A in one fixed order (run-to-run identical), scale / shift /
//                    activation / residual as conv_epilogue, 128-byte stores per half wave.
#include <cstdlib>

#include "x3_common.hpp"

// timing experiments only (tools/micro/w4x3_ablate.hip): pieces of the K loop removed -- 1 B loads, 2 split, 4 transforms, 8 gathers,
// 16 LDS stores, 32 MFMAs, 64 A-fragment reads, 128 the whole output stage (exchange, A^T M A, stores).  0 in the library.
#ifndef AV2X_W4X3_ABLATE
#define AV2X_W4X3_ABLATE 0
#endif
// 1: the transforms, LDS stores and gathers of a chunk run as ONE block without MFMAs at the head of the chunk and the MFMA gaps carry
// only the split, the B loads and the A reads (<= 5 side instructions per gap); 0: everything spread over the 108 gaps.
#ifndef AV2X_W4X3_BULK
#define AV2X_W4X3_BULK 0
#endif
// the six steps (of a group's twelve) that issue a B-fragment load, as a bit mask; results do not depend on it
#ifndef AV2X_W4X3_BMASK
#define AV2X_W4X3_BMASK 0x1CE
#endif

// timing experiments on the ping-pong form (tools/micro/w4x3_ablate.hip -DAV2X_W4PP_ABLATE=..): 1 consecutive MFMAs go to ALTERNATING
// accumulators (wrong sums: is the chain of six dependent MFMAs per position the M phase's length?), 2 no T-phase work (gathers,
// transforms, LDS stores), 4 no MFMAs, 8 no split.  0 in the library.
#ifndef AV2X_W4PP_ABLATE
#define AV2X_W4PP_ABLATE 0
#endif

namespace {

struct Wino4X3Params {
    const float* in;
    const void* u;       // bf16 [36 pos][cin/16][3 planes][2 k halves][coutp][8]
    const float* scale;
    const float* shift;
    const float* res;
    float* out;
    int H, W, Cin, in_ctot, in_coff;
    int Cout, CoutP, out_ctot, out_coff, relu;
    int TH, TW, tiles_per_img, T;
    int nblocks, chunks;
    int xcd_w;           // 1: the weight-stream form of the workgroup -> (tile block, cout block) map (see the kernel)
    unsigned in_bytes, u_bytes, out_bytes;
};

// B^T applied to six values of ONE channel in three pieces of four scalar operations (12 per transform), in place.  Scalar on purpose:
// beside MFMAs a packed fp32 instruction costs far more than the two scalar ones it replaces (MI355X_MICROARCH.md, "price of one filler");
// the translation unit is compiled with -fno-slp-vectorize so that the compiler does not re-pack them.
//   t1 = i4 - 4 i2, t2 = i3 - 4 i1, u1 = i4 - i2, s = i3 - i1
//   row 0: 4 i0 + (i4 - 5 i2)     row 1: t1 + t2     row 2: t1 - t2     row 3: u1 + 2 s     row 4: u1 - 2 s     row 5: 4 i1 + (i5 - 5 i3)
struct W4Tmp { float t1, t2, u1, s; };
template <int PIECE>
__device__ __forceinline__ void w4_bt6(float& v0, float& v1, float& v2, float& v3, float& v4, float& v5, W4Tmp& m) {
    if constexpr (PIECE == 0) {
        m.t1 = fmaf(-4.f, v2, v4); m.t2 = fmaf(-4.f, v1, v3); m.u1 = v4 - v2; m.s = v3 - v1;
    } else if constexpr (PIECE == 1) {
        v0 = fmaf(4.f, v0, fmaf(-5.f, v2, v4));
        v5 = fmaf(4.f, v1, fmaf(-5.f, v3, v5));
    } else {
        v1 = m.t1 + m.t2; v2 = m.t1 - m.t2; v3 = fmaf(2.f, m.s, m.u1); v4 = fmaf(-2.f, m.s, m.u1);
    }
}

template <bool GENERAL>
__global__ __launch_bounds__(256, 1) void conv_wino4_x3(const Wino4X3Params p) {
    constexpr int TB = 32, NP = 36, NG = 9, STEPS = 12 * NG;
    constexpr int KQS = TB * 16 + 32;      // bytes per k quad: [tile 32][4 floats] + 32 (pad: the gather threads' stores stay conflict-free)
    constexpr int POSB = 4 * KQS;          // bytes per position of a V stage
    constexpr int VSTAGE = NP * POSB;      // 78 336
    constexpr unsigned OOB = 0xC0000000u;  // stays >= 2^31 after the row-0 subtraction (row stride < 2^30, checked by the host)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // V[2][36][4][32][4] fp32; re-used for the exchange

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbk = gridDim.x, b = blockIdx.x;
    const int q8 = nbk >> 3, r8 = nbk & 7, xcd = b & 7;
    int mblock, nblock;
    if (p.xcd_w) {
        // Weight-stream map (workgroup b runs on XCD b % 8; speed and traffic only, same per-workgroup arithmetic): XCD x takes cout block
        // x % nblocks of every (8 / nblocks)-th tile block, so an XCD's L2 streams 1 / nblocks of the layer's split planes -- 3.5 of the
        // 14.2 MB of a 256 -> 256 layer, which every XCD fetched whole before (8 x 14.2 = 113 of the launch's 135 MB at 4 x 25 x 88) -- at the
        // price of the input map being fetched by nblocks XCDs instead of one.  The host picks it where that trade wins (weights x (8 - 8 /
        // nblocks) > input x (nblocks - 1)).  The hardware's workgroups-per-XCD counts, q8 + (x < r8), are exactly what this assignment needs.
        const int G = 8 / p.nblocks;
        nblock = xcd % p.nblocks;
        mblock = xcd / p.nblocks + (b >> 3) * G;
    } else {
        // activation map: an XCD takes a contiguous range of (tile block, cout block) items -- every cout block of its tile blocks
        const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
        mblock = swz / p.nblocks;
        nblock = swz - mblock * p.nblocks;
    }
    const int t0 = mblock * TB;
    const int n0 = nblock * 64;

    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.u), 0, p.u_bytes, 0x00020000);

    // ---- weight side: plane (pos, kb, pl) is [2 k halves][coutp][8 bf16]; lane (j = lane & 31, kh = lane >> 5) reads 16 bytes
    const unsigned voffU = (unsigned)(((lane >> 5) * p.CoutP + n0 + (lane & 31)) * 16);
    const int plane_stride = 32 * p.CoutP;                           // bytes
    const int kb_stride = 3 * plane_stride, pos_stride = (p.Cin >> 4) * kb_stride;
    const int ubase = 9 * wv * pos_stride;
    x3_u32x4 bs[3][2][3];                                           // [set = position % 3][32-cout block][plane]
    auto load_b = [&](auto set_c, auto i_c, int lp, int kb) {       // i = 3 nb + plane
        constexpr int set = decltype(set_c)::value, i = decltype(i_c)::value;
        bs[set][i / 3][i % 3] = __builtin_bit_cast(x3_u32x4, __builtin_amdgcn_raw_buffer_load_b128(
            ru, voffU + (i / 3) * 512, ubase + lp * pos_stride + kb * kb_stride + (i % 3) * plane_stride, 0));
    };
    x3_static_for<0, 6>([&](auto i) { load_b(std::integral_constant<int, 0>{}, i, 0, 0); });
    x3_static_for<0, 6>([&](auto i) { load_b(std::integral_constant<int, 1>{}, i, 1, 0); });

    f32x16 acc[18];                                                  // tile 2 lp + nb
#pragma unroll
    for (int i = 0; i < 18; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // ---- input side: this thread's tile and channel pair.  The anchor pixel (4 ty, 4 tx) = patch element (1, 1) always lies inside
    // the image; colE[e] = its byte offset moved to patch column e (OOB when that column is outside, or the tile is beyond the last one)
    const int tl = tid >> 3, cg = tid & 7;
    const int RSb = p.W * p.in_ctot * 4;                             // bytes per image row
    unsigned colE[6];
    unsigned long long rmask[6];                                     // lanes whose patch row a lies inside the image (wave-wide masks: SGPRs)
    {
        const int t = t0 + tl;
        const bool tok = t < p.T;
        const int tt = tok ? t : 0;
        const int img = tt / p.tiles_per_img, r = tt - img * p.tiles_per_img;
        const int ty = r / p.TW, tx = r - ty * p.TW;
        const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
        const int base = (((img * p.H + y0 + 1) * p.W + x0 + 1) * p.in_ctot + p.in_coff + cg * 2) * 4;
#pragma unroll
        for (int e = 0; e < 6; ++e)
            colE[e] = (tok && (unsigned)(x0 + e) < (unsigned)p.W) ? (unsigned)(base + (e - 1) * p.in_ctot * 4) : OOB;
#pragma unroll
        for (int a = 0; a < 6; ++a) rmask[a] = __builtin_amdgcn_ballot_w64((unsigned)(y0 + a) < (unsigned)p.H);
    }
    const unsigned oobv = OOB;
    float d[36][2];                                                  // the patch, [6 a + e][channel of the pair]
    auto gather = [&](auto k_c, int c) {
        constexpr int k = decltype(k_c)::value, a = k / 6, e = k % 6;
        // written out: as plain C++ the 30 selects are loop invariants, and the compiler keeps them in 30 registers this kernel does not have
        unsigned off = colE[e];
        if constexpr (a == 0)
            asm volatile("v_sub_u32 %0, %1, %2\n\tv_cndmask_b32 %0, %3, %0, %4" : "=&v"(off) : "v"(colE[e]), "s"(RSb), "v"(oobv), "s"(rmask[0]));
        else if constexpr (a >= 2)
            asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(off) : "v"(oobv), "v"(colE[e]), "s"(rmask[a]));
        const x3_f32x2 v = __builtin_bit_cast(x3_f32x2, __builtin_amdgcn_raw_buffer_load_b64(rin, off, (a >= 2 ? (a - 1) * RSb : 0) + c * 64, 0));
        d[k][0] = v.x; d[k][1] = v.y;
    };
    W4Tmp tm[2];                                                     // one set of temporaries per channel of the pair
    // B^T d: unit u = 6 e + 2 piece + channel, column e of the patch, in place (four scalar operations per unit)
    auto rows = [&](auto u_c) {
        constexpr int u = decltype(u_c)::value, e = u / 6, pc = (u % 6) >> 1, ch = u & 1;
        if constexpr (u % 6 == 0)   // the wait for the patch: HERE
            asm volatile("" : "+v"(d[e][0]), "+v"(d[6 + e][0]), "+v"(d[12 + e][0]), "+v"(d[18 + e][0]), "+v"(d[24 + e][0]), "+v"(d[30 + e][0]),
                              "+v"(d[e][1]), "+v"(d[6 + e][1]), "+v"(d[12 + e][1]), "+v"(d[18 + e][1]), "+v"(d[24 + e][1]), "+v"(d[30 + e][1]));
        w4_bt6<pc>(d[e][ch], d[6 + e][ch], d[12 + e][ch], d[18 + e][ch], d[24 + e][ch], d[30 + e][ch], tm[ch]);
    };
    // (B^T d) B: unit u = 6 a + 2 piece + channel, row a, in place; then element (a, j) -> V[pos 6 a + j][k quad][tile][.]
    auto colsx = [&](auto u_c) {
        constexpr int u = decltype(u_c)::value, a = u / 6, pc = (u % 6) >> 1, ch = u & 1;
        w4_bt6<pc>(d[6 * a][ch], d[6 * a + 1][ch], d[6 * a + 2][ch], d[6 * a + 3][ch], d[6 * a + 4][ch], d[6 * a + 5][ch], tm[ch]);
    };
    const unsigned wbase = (unsigned)(((cg * 2) >> 2) * KQS + tl * 16 + ((cg * 2) & 3) * 4);
    auto vstore = [&](auto o_c, int stage) {
        constexpr int o = decltype(o_c)::value;
        x3_f32x2 v;
        v.x = d[o][0]; v.y = d[o][1];
        *reinterpret_cast<x3_f32x2*>(smem + stage * VSTAGE + o * POSB + wbase) = v;
    };

    // A fragment of local position lp: lane (i = lane & 31, kh = lane >> 5) reads k quads 2 kh, 2 kh + 1 of tile i
    const unsigned rbase = (unsigned)((2 * (lane >> 5)) * KQS + (lane & 31) * 16) + 9 * wv * POSB;
    x3_f32x2 raw[4];      // ONE raw fragment: half h of the next one is read as soon as the split has consumed half h of this one
    unsigned pl[2][3][4];
    X3Split sp;
    auto read_a = [&](auto half_c, int lp, int stage) {
        constexpr int half = decltype(half_c)::value;
        const x3_pair2 v = __builtin_bit_cast(x3_pair2, *reinterpret_cast<const f32x4*>(smem + stage * VSTAGE + rbase + lp * POSB + half * KQS));
        raw[2 * half] = v.a; raw[2 * half + 1] = v.b;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // prologue: chunk 0 -> stage 0, the patch of chunk 1 in flight, the first two A fragments read and the first one split
    x3_static_for<0, 36>([&](auto k) { gather(k, 0); });
    x3_static_for<0, 36>([&](auto u) { rows(u); });
    x3_static_for<0, 36>([&](auto u) { colsx(u); });
    x3_static_for<0, 36>([&](auto o) { vstore(o, 0); });
    {
        const int c1 = min(1, p.chunks - 1);
        x3_static_for<0, 36>([&](auto k) { gather(k, c1); });
    }
    __syncthreads();
    read_a(I0{}, 0, 0);
    read_a(I1{}, 0, 0);
    x3_static_for<0, 12>([&](auto jj) { x3_split_step<decltype(jj)::value>(raw, pl[0], sp); });
    read_a(I0{}, 1, 0);
    read_a(I1{}, 1, 0);

    // One chunk (parity PC = c & 1 = LDS stage of V(c); split planes of group g: pl[(g + PC) & 1], B fragments: bs[g % 3], loaded TWO
    // groups ahead -- an L2 round trip under load is ~600 cycles, a group's MFMAs 384):
    //   steps 0 .. 35        B^T d of the patch of chunk c + 1 (gathered during chunk c - 1), one unit of four operations per step
    //   steps 36 + 8 a ..    row a of (B^T d) B: six units, one per step; its six LDS stores into the other stage in steps 4 .. 7
    //                        of the row (elements 0 and 5 are complete after unit 3, elements 1 .. 4 after unit 5)
    //   step  83             the ONE barrier: V(c + 1) complete, nobody reads stage PC any more (last read: group 6, for group 8)
    //   groups 7, 8          read the first two A fragments of chunk c + 1; group 8 splits the first one -> no bubble between chunks
    //   steps 44 + 8 a ..    row a of the patch of chunk c + 2, one load per step (its registers are free once row a is stored)
    auto chunk = [&](auto pc_c, int c) {
        constexpr int PC = decltype(pc_c)::value;
        const int c1 = min(c + 1, p.chunks - 1), c2 = min(c + 2, p.chunks - 1);
        constexpr int AB0 = AV2X_W4X3_ABLATE;
        constexpr bool BULK = AV2X_W4X3_BULK != 0;
        if constexpr (BULK) {
            // One wave per SIMD hides at most ~5 side instructions behind an MFMA; every further one costs far more than its issue slot
            // (measured: 108 gaps with ~8.7 side instructions ran 96 cycles per MFMA, the MFMAs alone 35, the side stream alone 50).  So
            // the ~400 instructions of the patch (both transform passes, 36 LDS stores, 36 gathers + their row selects) run here, back
            // to back at ~5 cycles each with the matrix pipe idle, and the gaps below stay at <= 5.
            if constexpr (!(AB0 & 4)) {
                x3_static_for<0, 36>([&](auto u) { rows(u); });
                x3_static_for<0, 36>([&](auto u) { colsx(u); });
            }
            if constexpr (!(AB0 & 16)) x3_static_for<0, 36>([&](auto o) { vstore(o, PC ^ 1); });
            if constexpr (!(AB0 & 8)) x3_static_for<0, 36>([&](auto k) { gather(k, c2); });
            __builtin_amdgcn_sched_barrier(0);
        }
        x3_static_for<0, STEPS>([&](auto ss) {
            constexpr int s = decltype(ss)::value;
            constexpr int g = s / 12, j = s % 12, pr = j >> 1, nb = j & 1, T = 2 * g + nb;
            constexpr int cur = (g + PC) & 1, nxt = cur ^ 1;
            constexpr int AB = AV2X_W4X3_ABLATE;
            if constexpr (AB & 32) {
            } else if constexpr (T < 16) {
                acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3_frag(pl[cur][x3_ap(pr)]), __builtin_bit_cast(x3_bf16x8, bs[g % 3][nb][x3_bp(pr)]),
                                                               acc[T], 0, 0, 0);
            } else {   // the two tiles beyond the 256 accumulation registers live in ordinary VGPRs: fixed register class (conv_wino4.inc)
                x3_u32x4 af;
                af[0] = pl[cur][x3_ap(pr)][0]; af[1] = pl[cur][x3_ap(pr)][1]; af[2] = pl[cur][x3_ap(pr)][2]; af[3] = pl[cur][x3_ap(pr)][3];
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[T]) : "v"(af), "v"(bs[g % 3][nb][x3_bp(pr)]));
            }
            // split of the next group's A fragment (the last group: the first fragment of the next chunk)
            if constexpr (!(AB & 2)) x3_split_step<j>(raw, pl[nxt], sp);
            // raw A fragment of group g + 2 (of the next chunk from group 7 on): each half right after the split's last read of it
            if constexpr ((j == 4 || j == 9) && !(AB & 64)) {
                using HALF = std::integral_constant<int, j == 4 ? 0 : 1>;
                if constexpr (g + 2 < NG) read_a(HALF{}, g + 2, PC);
                else read_a(HALF{}, g + 2 - NG, PC ^ 1);
            }
            // B fragments of position g + 2 into the register set position g - 1 has released (BULK: in the gaps the split leaves light)
            // which six of a group's twelve steps carry a B load: AV2X_W4X3_BMASK (bit j = step j; 0x1CE = steps 1,2,3,6,7,8)
            constexpr int BM_ = BULK ? 0xE31 : AV2X_W4X3_BMASK;
            constexpr bool bstep = (BM_ >> j) & 1;
            if constexpr (bstep && !(AB & 1)) {
                using IB = std::integral_constant<int, __builtin_popcount(BM_ & ((1 << j) - 1))>;
                if constexpr (g + 2 < NG) load_b(std::integral_constant<int, (g + 2) % 3>{}, IB{}, g + 2, c);
                else load_b(std::integral_constant<int, (g + 2) % 3>{}, IB{}, g + 2 - NG, c1);
            }
            if constexpr (s < 36 && !(AB & 4) && !BULK) rows(std::integral_constant<int, s>{});
            if constexpr (s >= 36 && s < 84 && !BULK) {
                constexpr int a = (s - 36) / 8, q = (s - 36) % 8;
                if constexpr (q < 6 && !(AB & 4)) colsx(std::integral_constant<int, 6 * a + q>{});
                if constexpr (!(AB & 16)) {
                    if constexpr (q == 4) vstore(std::integral_constant<int, 6 * a>{}, PC ^ 1);
                    if constexpr (q == 5) vstore(std::integral_constant<int, 6 * a + 5>{}, PC ^ 1);
                    if constexpr (q == 6) { vstore(std::integral_constant<int, 6 * a + 1>{}, PC ^ 1); vstore(std::integral_constant<int, 6 * a + 2>{}, PC ^ 1); }
                    if constexpr (q == 7) { vstore(std::integral_constant<int, 6 * a + 3>{}, PC ^ 1); vstore(std::integral_constant<int, 6 * a + 4>{}, PC ^ 1); }
                }
            }
            if constexpr (s >= 44 && s < 92 && (s - 44) % 8 < 6 && !(AB & 8) && !BULK) gather(std::integral_constant<int, 6 * ((s - 44) / 8) + (s - 44) % 8>{}, c2);
            if constexpr (s == 83) x3_lds_barrier();   // LDS traffic only: the register prefetches stay in flight
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    for (int c = 0; c < p.chunks; c += 2) {
        chunk(I0{}, c);
        chunk(I1{}, c + 1);
    }

    // ---- output transform through LDS, one 32-cout block per pass: X[pos][row][lane]; wave w finalises rows 4 w .. 4 w + 3 of the block:
    // Z[a][nu] = sum_xi A^T[a][xi] M[xi][nu], Y[a][e] = sum_nu Z[a][nu] A^T[e][nu]
    if constexpr ((AV2X_W4X3_ABLATE & 128) != 0) {   // timing only: no exchange, no output transform, one store per lane
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 18; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) t += acc[i][r];
        p.out[(size_t)b * 256 + tid] = t;
        return;
    }
    float* X = reinterpret_cast<float*>(smem);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    const int opix = p.out_ctot * 4;
    const bool relu1 = p.relu == 1;
    x3_static_for<0, 2>([&](auto nb_c) {
        constexpr int nb = decltype(nb_c)::value;
        __syncthreads();   // the V stages (first pass) / the previous pass's exchange have been read
        x3_static_for<0, 9>([&](auto lp_c) {
            constexpr int lp = decltype(lp_c)::value;
            float* xp = X + ((9 * wv + lp) * 16) * 64 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) xp[r * 64] = acc[2 * lp + nb][r];
        });
        __syncthreads();
        // finalising side, COUT-MAJOR (as conv_wino_x3): four consecutive lanes of an X row are four consecutive couts of one tile, so thread
        // (quad q = tid & 7, accumulator row = tid >> 4, half-wave kh = (tid >> 3) & 1) takes ONE (tile, cout quad) item of the pass: 36
        // ds_read_b128, the two transforms on four couts at once, 16 pixel stores of 16 bytes (16 consecutive threads write the 128 bytes of
        // a pixel's 32 couts).  Round 4: one dword per lane and pixel -- 64 store instructions per lane and pass.
        const int q = tid & 7, kh = (tid >> 3) & 1, row = tid >> 4;
        const int n = n0 + nb * 32 + q * 4;
        const bool nok = n < p.Cout;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (nok) {
            if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n);
            sh = *reinterpret_cast<const f32x4*>(p.shift + n);
        }
        const unsigned ocol = (unsigned)((p.out_coff + n) * 4);
        // row r of the accumulator tile is tile (r & 3) + 8 (r >> 2) + 4 kh of the block
        const int t = t0 + (row & 3) + 8 * (row >> 2) + 4 * kh;
        const int img = t / p.tiles_per_img;
        const int ty = (t - img * p.tiles_per_img) / p.TW;
        const int tx = t - img * p.tiles_per_img - ty * p.TW;
        const float* xq = X + row * 64 + kh * 32 + q * 4;
        f32x4 z[4][6];
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) {
            f32x4 m[6];
#pragma unroll
            for (int xi = 0; xi < 6; ++xi) m[xi] = *reinterpret_cast<const f32x4*>(xq + ((xi * 6 + nu) * 16) * 64);
            const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
            z[0][nu] = (m[0] + s12) + s34;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                z[1][nu][e] = fmaf(2.f, d34[e], d12[e]);
                z[2][nu][e] = fmaf(4.f, s34[e], s12[e]);
                z[3][nu][e] = fmaf(8.f, d34[e], d12[e]) + m[5][e];
            }
        }
        const bool tvalid = nok && t < p.T;
        const int py = 4 * ty, px = 4 * tx;
        const unsigned pix = (unsigned)(((img * p.H + py) * p.W + px) * opix) + ocol;    // < 2^31 (checked by the host)
        bool colok[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) colok[e] = tvalid && px + e < p.W;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f32x4 s12 = z[a][1] + z[a][2], d12 = z[a][1] - z[a][2], s34 = z[a][3] + z[a][4], d34 = z[a][3] - z[a][4];
            f32x4 y[4];
            y[0] = (z[a][0] + s12) + s34;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                y[1][c] = fmaf(2.f, d34[c], d12[c]);
                y[2][c] = fmaf(4.f, s34[c], s12[c]);
                y[3][c] = fmaf(8.f, d34[c], d12[c]) + z[a][5][c];
            }
            const bool rowok = py + a < p.H;
            const unsigned rowoff = pix + (unsigned)(a * p.W * opix);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned off = (rowok && colok[e]) ? rowoff + (unsigned)(e * opix) : 0x80000000u;   // invalid: outside the range
                f32x4 v;
                f32x4 rs = {0.f, 0.f, 0.f, 0.f};
                if (GENERAL) {
                    if (p.res && off < 0x80000000u) {
                        const size_t m_ = (size_t)(off - ocol) / (size_t)opix;
                        rs = *reinterpret_cast<const f32x4*>(p.relu == 4 ? p.res + m_ * p.Cout + n : p.res + m_ * p.out_ctot + p.out_coff + n);
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float u = fmaf(y[e][c], sc[c], sh[c]);
                    if (GENERAL) {
                        if (p.relu == 1) u = fmaxf(u, 0.f);
                        else if (p.relu == 3) u = 1.0f / (1.0f + expf(-u));
                        else if (p.relu == 4) u = tanhf(u);
                        if (p.res && off < 0x80000000u) u = (p.relu == 4) ? u * rs[c] : u + rs[c];
                        if (p.relu == 5) u = fmaxf(u, 0.f);
                    } else {
                        u = relu1 ? fmaxf(u, 0.f) : u;
                    }
                    v[c] = u;
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(x3_u32x4, v), rout, off, 0, 0);
            }
        }
    });
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv_wino4_x3_pp: the same arithmetic (bit for bit: every accumulator receives the same six products per chunk in the same order, the
// transforms and the finalising side are the same expressions) on EIGHT waves -- two per SIMD -- in a ping-pong schedule (round 6).
//
// Why.  The four-wave kernel above keeps ONE in-order wave per SIMD: its K loop is a sum, not a maximum, of its matrix time (3 456 cycles
// per 16-channel chunk) and its vector-memory time (216 KB of B fragments + 74 KB of patch pixels through the CU's 64 B/clk path = 4 530
// cycles): 7 100 cycles per chunk measured, zero overlap (profiles/r05w_wino4_x3_small_launches.txt), because a wave that blocks on a full
// memory queue or on the issue latency of its ~11 instructions per MFMA (4.7 cycles each: SQ_ACTIVE_INST_ANY / instructions) leaves the
// matrix pipe of its SIMD idle.  Two waves per SIMD halve the register budget (256), and 36 positions x 32 tiles x 64 couts of fp32
// accumulators are 144 registers per wave of eight -- so the two waves of a SIMD must never hold their other big register sets at the same
// program point.  Hence two ROLES that alternate:
//   M phase  (matrix)  wave (pg, nb) multiplies its nine positions 9 pg .. 9 pg + 8 of chunk c for its 32-cout block nb: 54 MFMAs, the
//                      hi / mid / lo split of its A fragments, B fragments from L2 three positions ahead (four register sets);
//   T phase  (memory)  its threads gather the 6x6 patches of 16 tiles x 16 channels of chunk c + 1 (one channel per thread), apply B^T d B
//                      and store fp32 V to the other LDS stage; the first three positions' B fragments of the NEXT M phase are requested
//                      first, so they arrive while the patch is transformed.
// The halves H0 = waves 0-3 (nb = 0, tiles 0-15 as producers) and H1 = waves 4-7 (nb = 1, tiles 16-31) run the two roles in opposite
// phases, a workgroup barrier after each:   phase 2c: H0 M(c) | H1 T(c+1)     phase 2c+1: H0 T(c+1) | H1 M(c).
// Every SIMD then always holds one wave that issues MFMAs and one that issues loads and VALU: matrix beside memory, the pairing that pays
// (MI355X_MICROARCH.md, "Two waves per SIMD", item 5).  V(c+1) is complete after phase 2c+1; its stage was last read in phase 2c-1.
// Same V layout, same LDS budget (2 x 76.5 KB), same B and patch bytes per MFMA as the four-wave kernel; the split runs once per (position,
// cout block) instead of once per position (6 instead of 3 VALU per MFMA), which the second wave's issue slots absorb.
template <bool GENERAL>
__global__ __launch_bounds__(512) void conv_wino4_x3_pp(const Wino4X3Params p) {
    constexpr int TB = 32, NP = 36, NG = 9;
    constexpr int KQS = TB * 16 + 32;
    constexpr int POSB = 4 * KQS;
    constexpr int VSTAGE = NP * POSB;
    constexpr unsigned OOB = 0xC0000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wv & 3, half = wv >> 2;
    const int nbk = gridDim.x, b = blockIdx.x;
    const int q8 = nbk >> 3, r8 = nbk & 7, xcd = b & 7;
    int mblock, nblock;
    if (p.xcd_w) {
        const int G = 8 / p.nblocks;
        nblock = xcd % p.nblocks;
        mblock = xcd / p.nblocks + (b >> 3) * G;
    } else {
        const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
        mblock = swz / p.nblocks;
        nblock = swz - mblock * p.nblocks;
    }
    const int t0 = mblock * TB;
    const int n0 = nblock * 64;

    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.u), 0, p.u_bytes, 0x00020000);

    // ---- weight side (M role): this wave's 32-cout block is n0 + 32 half
    const unsigned voffU = (unsigned)(((lane >> 5) * p.CoutP + n0 + 32 * half + (lane & 31)) * 16);
    const int plane_stride = 32 * p.CoutP;
    const int kb_stride = 3 * plane_stride, pos_stride = (p.Cin >> 4) * kb_stride;
    const int ubase = 9 * pg * pos_stride;
    // [set = position % 3][plane].  A plane's registers are re-loaded with position g + 3's fragment right after the last MFMA of
    // position g that reads them (lo: step 1, mid: step 4, hi: step 5; the data returns hundreds of cycles after that MFMA has read its
    // operands): a three-deep ring with a prefetch distance of 13-17 MFMAs (~500-650 cycles, an L2 round trip under load)
    x3_u32x4 bs[3][3];
    auto load_b = [&](auto set_c, auto pl_c, int lp, int kb) {
        constexpr int set = decltype(set_c)::value, pln = decltype(pl_c)::value;
        bs[set][pln] = __builtin_bit_cast(x3_u32x4, __builtin_amdgcn_raw_buffer_load_b128(
            ru, voffU, ubase + lp * pos_stride + kb * kb_stride + pln * plane_stride, 0));
    };
    // B fragments of positions 0 .. 2 of chunk kb: requested at the head of the T phase that precedes the M phase of that chunk
    auto bprefetch = [&](int kb) {
        x3_static_for<0, 9>([&](auto i) {
            constexpr int ii = decltype(i)::value;
            load_b(std::integral_constant<int, ii / 3>{}, std::integral_constant<int, ii % 3>{}, ii / 3, kb);
        });
    };

    f32x16 acc[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // ---- input side (T role): thread t of the half takes tile 16 half + (t >> 4), channel t & 15 of every chunk
    const int tq = tid & 255;
    const int tl = 16 * half + (tq >> 4), ch = tq & 15;
    const int RSb = p.W * p.in_ctot * 4;
    unsigned colE[6];
    unsigned long long rmask[6];
    {
        const int t = t0 + tl;
        const bool tok = t < p.T;
        const int tt = tok ? t : 0;
        const int img = tt / p.tiles_per_img, r = tt - img * p.tiles_per_img;
        const int ty = r / p.TW, tx = r - ty * p.TW;
        const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
        const int base = (((img * p.H + y0 + 1) * p.W + x0 + 1) * p.in_ctot + p.in_coff + ch) * 4;
#pragma unroll
        for (int e = 0; e < 6; ++e)
            colE[e] = (tok && (unsigned)(x0 + e) < (unsigned)p.W) ? (unsigned)(base + (e - 1) * p.in_ctot * 4) : OOB;
#pragma unroll
        for (int a = 0; a < 6; ++a) rmask[a] = __builtin_amdgcn_ballot_w64((unsigned)(y0 + a) < (unsigned)p.H);
    }
    const unsigned oobv = OOB;
    float d[36];
    auto gather = [&](auto k_c, int c) {
        constexpr int k = decltype(k_c)::value, a = k / 6, e = k % 6;
        unsigned off = colE[e];
        if constexpr (a == 0)
            asm volatile("v_sub_u32 %0, %1, %2\n\tv_cndmask_b32 %0, %3, %0, %4" : "=&v"(off) : "v"(colE[e]), "s"(RSb), "v"(oobv), "s"(rmask[0]));
        else if constexpr (a >= 2)
            asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(off) : "v"(oobv), "v"(colE[e]), "s"(rmask[a]));
        d[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rin, off, (a >= 2 ? (a - 1) * RSb : 0) + c * 64, 0));
    };
    W4Tmp tm;
    auto rows = [&](auto u_c) {      // B^T d: unit u = 3 e + piece, column e of the patch, in place
        constexpr int u = decltype(u_c)::value, e = u / 3, pc = u % 3;
        if constexpr (pc == 0)       // the wait for the patch column: HERE
            asm volatile("" : "+v"(d[e]), "+v"(d[6 + e]), "+v"(d[12 + e]), "+v"(d[18 + e]), "+v"(d[24 + e]), "+v"(d[30 + e]));
        w4_bt6<pc>(d[e], d[6 + e], d[12 + e], d[18 + e], d[24 + e], d[30 + e], tm);
    };
    auto colsx = [&](auto u_c) {     // (B^T d) B: unit u = 3 a + piece, row a, in place
        constexpr int u = decltype(u_c)::value, a = u / 3, pc = u % 3;
        w4_bt6<pc>(d[6 * a], d[6 * a + 1], d[6 * a + 2], d[6 * a + 3], d[6 * a + 4], d[6 * a + 5], tm);
    };
    const unsigned wbase = (unsigned)((ch >> 2) * KQS + tl * 16 + (ch & 3) * 4);
    auto vstore = [&](auto o_c, int stage) {
        constexpr int o = decltype(o_c)::value;
        *reinterpret_cast<float*>(smem + stage * VSTAGE + o * POSB + wbase) = d[o];
    };
    // T phase body: patch of chunk c -> V stage `stage`
    auto tproduce = [&](int c, int stage) {
        if constexpr ((AV2X_W4PP_ABLATE & 2) != 0) return;
        x3_static_for<0, 36>([&](auto k) { gather(k, c); });
        __builtin_amdgcn_sched_barrier(0);
        x3_static_for<0, 18>([&](auto u) { rows(u); });
        x3_static_for<0, 6>([&](auto a_c) {
            constexpr int a = decltype(a_c)::value;
            colsx(std::integral_constant<int, 3 * a>{});
            colsx(std::integral_constant<int, 3 * a + 1>{});
            vstore(std::integral_constant<int, 6 * a>{}, stage);          // elements 0 and 5 of the row are complete after piece 1
            vstore(std::integral_constant<int, 6 * a + 5>{}, stage);
            colsx(std::integral_constant<int, 3 * a + 2>{});
            vstore(std::integral_constant<int, 6 * a + 1>{}, stage);
            vstore(std::integral_constant<int, 6 * a + 2>{}, stage);
            vstore(std::integral_constant<int, 6 * a + 3>{}, stage);
            vstore(std::integral_constant<int, 6 * a + 4>{}, stage);
        });
    };

    // ---- A fragments (M role): lane (i = lane & 31, kh = lane >> 5) reads k quads 2 kh, 2 kh + 1 of tile i of its positions
    const unsigned rbase = (unsigned)((2 * (lane >> 5)) * KQS + (lane & 31) * 16) + 9 * pg * POSB;
    x3_f32x2 raw[4];
    // split planes of the current / next position: hi and mid double-buffered; lo is read by the FIRST product of a position only
    // (x3_ap(0) == 2), and the split writes the next position's lo plane from its step 4 on (MFMA step 2): one buffer
    unsigned plh[2][4], plm[2][4], pll[4];
    X3Split sp;
    auto split = [&](auto j_c, auto nxt_c) {      // x3_split_step<J> with the three planes held separately
        constexpr int J = decltype(j_c)::value, NX = decltype(nxt_c)::value;
        constexpr int ua = x3_sa(J), up = x3_sp(J), qf = x3_sf(J);
        unsigned w = 0;
        if constexpr (ua >= 0) {
            constexpr int q = 2 * (ua >> 2) + (ua & 1), ph = (ua >> 1) & 1;
            const x3_f32x2 src = ph == 0 ? raw[q] : sp.r[q & 1];
            w = __builtin_bit_cast(unsigned, __builtin_convertvector(src, x3_bf16x2));
            if constexpr (ph == 0) plh[NX][q] = w;
            else plm[NX][q] = w;
        }
        if constexpr (up >= 0) {
            constexpr int q = 2 * (up >> 2) + (up & 1), ph = (up >> 1) & 1;
            sp.r[q & 1] = x3_pk_sub(ph == 0 ? raw[q] : sp.r[q & 1], sp.hf[up & 1]);
        }
        if constexpr (ua >= 0) {
            sp.hf[ua & 1].x = __builtin_bit_cast(float, w << 16);
            sp.hf[ua & 1].y = __builtin_bit_cast(float, w & 0xffff0000u);
        }
        if constexpr (qf >= 0) pll[qf] = __builtin_bit_cast(unsigned, __builtin_convertvector(sp.r[qf & 1], x3_bf16x2));
    };
    auto read_a = [&](auto half_c, int lp, int stage) {
        constexpr int hf = decltype(half_c)::value;
        const x3_pair2 v = __builtin_bit_cast(x3_pair2, *reinterpret_cast<const f32x4*>(smem + stage * VSTAGE + rbase + lp * POSB + hf * KQS));
        raw[2 * hf] = v.a; raw[2 * hf + 1] = v.b;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // head of an M phase: fragment 0 split, fragment 1 read
    auto mhead = [&](int stage) {
        read_a(I0{}, 0, stage);
        read_a(I1{}, 0, stage);
        x3_static_for<0, 12>([&](auto jj) { split(jj, I0{}); });
        read_a(I0{}, 1, stage);
        read_a(I1{}, 1, stage);
    };
    // M phase body: 9 groups (positions) x 6 MFMAs on chunk c; B fragments of positions 0 .. 2 are in sets 0 .. 2 (bprefetch)
    auto mphase = [&](int c, int stage) {
        x3_static_for<0, 6 * NG>([&](auto ss) {
            constexpr int s = decltype(ss)::value;
            constexpr int g = s / 6, j = s % 6, cur = g & 1, nxt = cur ^ 1;
            constexpr int ap = x3_ap(j), bp = x3_bp(j);
            constexpr int AG = (AV2X_W4PP_ABLATE & 1) ? (g + (j & 1)) % NG : g;
            if constexpr ((AV2X_W4PP_ABLATE & 4) != 0) {
            } else if constexpr (ap == 0) acc[AG] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3_frag(plh[cur]), __builtin_bit_cast(x3_bf16x8, bs[g % 3][bp]), acc[AG], 0, 0, 0);
            else if constexpr (ap == 1) acc[AG] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3_frag(plm[cur]), __builtin_bit_cast(x3_bf16x8, bs[g % 3][bp]), acc[AG], 0, 0, 0);
            else acc[AG] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3_frag(pll), __builtin_bit_cast(x3_bf16x8, bs[g % 3][bp]), acc[AG], 0, 0, 0);
            if constexpr (g + 1 < NG && !(AV2X_W4PP_ABLATE & 8)) {      // split of the next position's A fragment: two of its twelve steps per MFMA
                split(std::integral_constant<int, 2 * j>{}, std::integral_constant<int, nxt>{});
                split(std::integral_constant<int, 2 * j + 1>{}, std::integral_constant<int, nxt>{});
            }
            if constexpr (g + 2 < NG && j == 2) read_a(I0{}, g + 2, stage);   // each half right after the split's last read of it
            if constexpr (g + 2 < NG && j == 4) read_a(I1{}, g + 2, stage);
            // B fragments three positions ahead into the plane this step has read for the last time (x3_bp: lo at 1, mid at 4, hi at 5)
            if constexpr (g + 3 < NG && (j == 1 || j == 4 || j == 5))
                load_b(std::integral_constant<int, g % 3>{}, std::integral_constant<int, (j == 1 ? 2 : j == 4 ? 1 : 0)>{}, g + 3, c);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // ---- prologue: V(0) by both halves; then the two halves run their roles in opposite phases
    if (half == 0) bprefetch(0);
    tproduce(0, 0);
    x3_lds_barrier();
    if (half == 0) {
        for (int c = 0; c < p.chunks; ++c) {
            mhead(c & 1);
            mphase(c, c & 1);
            x3_lds_barrier();
            if (c + 1 < p.chunks) {
                bprefetch(c + 1);
                tproduce(c + 1, (c + 1) & 1);
            }
            x3_lds_barrier();
        }
    } else {
        for (int c = 0; c < p.chunks; ++c) {
            bprefetch(c);
            if (c + 1 < p.chunks) tproduce(c + 1, (c + 1) & 1);
            mhead(c & 1);                 // V(c) has been complete since the previous barrier: no wait at the head of this half's M phase
            x3_lds_barrier();
            mphase(c, c & 1);
            x3_lds_barrier();
        }
    }

    // ---- output transform through LDS in two passes of 16 TILES each (accumulator registers 8 pass .. 8 pass + 7 of every M tile = tiles
    // 16 pass .. 16 pass + 15, both cout blocks): every wave dumps half of its accumulators per pass, so at most 72 of them are live while
    // the finalising side runs, and all 512 threads finalise in both passes: thread (cout quad q8, half-wave kh, register rr, cout block nbq)
    // of half h takes output rows 2 h, 2 h + 1 of its (tile, cout quad) item.  Y = A^T M A in the four-wave kernel's order: same bits.
    float* X = reinterpret_cast<float*>(smem);     // [pos 36][rr 8][cout block 2][lane 64]
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    const int opix = p.out_ctot * 4;
    const bool relu1 = p.relu == 1;
    const int cq = tq & 7, kh = (tq >> 3) & 1, rr = (tq >> 4) & 7, nbq = tq >> 7;
    const int n = n0 + nbq * 32 + cq * 4;
    const bool nok = n < p.Cout;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (nok) {
        if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n);
        sh = *reinterpret_cast<const f32x4*>(p.shift + n);
    }
    const unsigned ocol = (unsigned)((p.out_coff + n) * 4);
    const float* xq = X + (rr * 2 + nbq) * 64 + kh * 32 + cq * 4;
    x3_static_for<0, 2>([&](auto ps_c) {
        constexpr int ps = decltype(ps_c)::value;
        if constexpr (ps == 1) __syncthreads();      // the previous pass's exchange has been read
        x3_static_for<0, 9>([&](auto lp_c) {
            constexpr int lp = decltype(lp_c)::value;
            float* xp = X + (((9 * pg + lp) * 8) * 2 + half) * 64 + lane;
#pragma unroll
            for (int r = 0; r < 8; ++r) xp[r * 128] = acc[lp][8 * ps + r];
        });
        __syncthreads();
        // register 8 ps + rr of an accumulator tile is tile (rr & 3) + 8 ((8 ps + rr) >> 2) + 4 kh of the block
        const int t = t0 + (rr & 3) + 8 * (2 * ps + (rr >> 2)) + 4 * kh;
        const int img = t / p.tiles_per_img;
        const int ty = (t - img * p.tiles_per_img) / p.TW;
        const int tx = t - img * p.tiles_per_img - ty * p.TW;
        const bool tvalid = nok && t < p.T;
        const int py = 4 * ty, px = 4 * tx;
        const unsigned pix = (unsigned)(((img * p.H + py) * p.W + px) * opix) + ocol;
        bool colok[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) colok[e] = tvalid && px + e < p.W;
        auto fin = [&](auto ah_c) {
            constexpr int AH = decltype(ah_c)::value;
            f32x4 z[2][6];
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) {
                f32x4 m[6];
#pragma unroll
                for (int xi = 0; xi < 6; ++xi) m[xi] = *reinterpret_cast<const f32x4*>(xq + (xi * 6 + nu) * 1024);
                const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
                if constexpr (AH == 0) {
                    z[0][nu] = (m[0] + s12) + s34;
#pragma unroll
                    for (int e = 0; e < 4; ++e) z[1][nu][e] = fmaf(2.f, d34[e], d12[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        z[0][nu][e] = fmaf(4.f, s34[e], s12[e]);
                        z[1][nu][e] = fmaf(8.f, d34[e], d12[e]) + m[5][e];
                    }
                }
            }
#pragma unroll
            for (int al = 0; al < 2; ++al) {
                const int a = 2 * AH + al;
                const f32x4 s12 = z[al][1] + z[al][2], d12 = z[al][1] - z[al][2], s34 = z[al][3] + z[al][4], d34 = z[al][3] - z[al][4];
                f32x4 y[4];
                y[0] = (z[al][0] + s12) + s34;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    y[1][c] = fmaf(2.f, d34[c], d12[c]);
                    y[2][c] = fmaf(4.f, s34[c], s12[c]);
                    y[3][c] = fmaf(8.f, d34[c], d12[c]) + z[al][5][c];
                }
                const bool rowok = py + a < p.H;
                const unsigned rowoff = pix + (unsigned)(a * p.W * opix);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned off = (rowok && colok[e]) ? rowoff + (unsigned)(e * opix) : 0x80000000u;
                    f32x4 v;
                    f32x4 rs = {0.f, 0.f, 0.f, 0.f};
                    if (GENERAL) {
                        if (p.res && off < 0x80000000u) {
                            const size_t m_ = (size_t)(off - ocol) / (size_t)opix;
                            rs = *reinterpret_cast<const f32x4*>(p.relu == 4 ? p.res + m_ * p.Cout + n : p.res + m_ * p.out_ctot + p.out_coff + n);
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float u = fmaf(y[e][c], sc[c], sh[c]);
                        if (GENERAL) {
                            if (p.relu == 1) u = fmaxf(u, 0.f);
                            else if (p.relu == 3) u = 1.0f / (1.0f + expf(-u));
                            else if (p.relu == 4) u = tanhf(u);
                            if (p.res && off < 0x80000000u) u = (p.relu == 4) ? u * rs[c] : u + rs[c];
                            if (p.relu == 5) u = fmaxf(u, 0.f);
                        } else {
                            u = relu1 ? fmaxf(u, 0.f) : u;
                        }
                        v[c] = u;
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(x3_u32x4, v), rout, off, 0, 0);
                }
            }
        };
        if (half == 0) fin(I0{});
        else fin(I1{});
    });
}

// U = G g G^T per (cin, cout) in fp64, split into hi / mid / lo bf16 (round to nearest even at every step):
// w packed [tap][cin/4][coutp][4] fp32  ->  u [pos 36][cin/16][plane][k half][coutp][8] bf16
__global__ void wino4_x3_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ u, int cin, int coutp) {
    const size_t plane = (size_t)cin * coutp;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;    // index into [cin/4][coutp][4]
    if (e >= plane) return;
    const int k4 = (int)(e & 3), co = (int)((e >> 2) % coutp), q = (int)((e >> 2) / coutp);
    const int k = 4 * q + k4, kb = k >> 4, kh = (k >> 3) & 1, k8 = k & 7;
    const double G[6][3] = {{0.25, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    double g[3][3], t[6][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) g[i][j] = (double)w[(size_t)(i * 3 + j) * plane + e];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[0][j] + G[i][1] * g[1][j] + G[i][2] * g[2][j];
    const size_t kbs = (size_t)(cin >> 4);
    const size_t pstride = (size_t)2 * coutp * 8;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            const double U = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
            const float h = __builtin_bit_cast(float, (unsigned)x3_bf16_rne((float)U) << 16);
            const double r1 = U - (double)h;
            const float m = __builtin_bit_cast(float, (unsigned)x3_bf16_rne((float)r1) << 16);
            const double r2 = r1 - (double)m;
            const size_t base = ((((size_t)(i * 6 + j) * kbs + kb) * 3) * 2 + kh) * (size_t)coutp * 8 + (size_t)co * 8 + k8;
            u[base] = (unsigned short)(__builtin_bit_cast(unsigned, h) >> 16);
            u[base + pstride] = (unsigned short)(__builtin_bit_cast(unsigned, m) >> 16);
            u[base + 2 * pstride] = x3_bf16_rne((float)r2);
        }
}

}  // namespace

namespace av2x {

// called by av2x_conv2d* for tile flag 0x60000000 | 0x0400 (conv_igemm.hip); `u` is the packing of av2x_wino4_x3_pack_weights
int wino4_x3_dispatch(const av2x_conv_desc* d, const float* in, const void* u, const float* scale, const float* shift,
                      const float* residual, float* out, hipStream_t st) {
    if (d->mode != AV2X_CONV || d->ks != 3 || d->stride != 1 || d->pad != 1 || d->ho != d->h || d->wo != d->w)
        return fail("av2x_conv2d: the Winograd tiles take 3x3 / stride 1 / pad 1 NHWC convolutions only");
    if (d->relu != 0 && d->relu != 1 && d->relu != 3 && d->relu != 4 && d->relu != 5)
        return fail("av2x_conv2d: the Winograd tiles fuse no activation, ReLU (before / after the residual), sigmoid or tanh (relu=%d)", d->relu);
    if (d->cin % 32 || d->cout % 64 || d->coutp % 64 || d->cout > d->coutp)
        return fail("av2x_conv2d: split-3 Winograd F(4x4,3x3) needs cin %% 32 == 0 and cout %% 64 == 0 (cin=%d cout=%d)", d->cin, d->cout);
    if (d->in_coff % 4 || d->in_ctot % 4) return fail("av2x_conv2d: input channel offset/stride must be multiples of 4");
    if (d->out_coff % 4 || d->out_ctot % 4) return fail("av2x_conv2d: split-3 Winograd stores 16-byte cout quads: output channel offset/stride must be multiples of 4");
    if (((d->tile >> 16) & 0x1fff) != 32 || (d->tile & 0x01ff) != 64)
        return fail("av2x_conv2d: the split-3 Winograd F(4x4,3x3) tile is 32 tiles x 64 couts");
    Wino4X3Params p;
    p.in = in; p.u = u; p.scale = scale; p.shift = shift; p.res = residual; p.out = out;
    p.H = d->h; p.W = d->w; p.Cin = d->cin; p.in_ctot = d->in_ctot; p.in_coff = d->in_coff;
    p.Cout = d->cout; p.CoutP = d->coutp; p.out_ctot = d->out_ctot; p.out_coff = d->out_coff; p.relu = d->relu;
    p.TH = (d->h + 3) / 4; p.TW = (d->w + 3) / 4; p.tiles_per_img = p.TH * p.TW;
    const long long T = (long long)d->n * p.tiles_per_img;
    if (T <= 0) return 0;
    if (T > (1ll << 28)) return fail("av2x_conv2d: too many Winograd tiles");
    p.T = (int)T;
    p.chunks = d->cin / 16;
    p.nblocks = d->cout / 64;
    const unsigned long long in_bytes = (unsigned long long)d->n * d->h * d->w * d->in_ctot * 4ull;
    const unsigned long long u_bytes = 36ull * d->cin * d->coutp * 6ull;
    if (in_bytes >= (1ull << 31) || u_bytes >= (1ull << 31))
        return fail("av2x_conv2d: input (%llu B) or transformed weights (%llu B) exceed the 2 GiB buffer-descriptor window", in_bytes, u_bytes);
    if ((unsigned long long)d->w * d->in_ctot * 4ull * 5ull >= (1ull << 30)) return fail("av2x_conv2d: image row too long for the split-3 F(4x4,3x3) tile");
    p.in_bytes = (unsigned)in_bytes;
    p.u_bytes = (unsigned)u_bytes;
    const unsigned long long out_bytes = (unsigned long long)d->n * d->h * d->w * d->out_ctot * 4ull;
    if (out_bytes >= (1ull << 31)) return fail("av2x_conv2d: output (%llu B) exceeds the 2 GiB buffer-descriptor window", out_bytes);
    p.out_bytes = (unsigned)out_bytes;
    const int mblocks = (p.T + 31) / 32;
    // which XCD map (see the kernel): the weight-stream form where the split planes, fetched once per XCD, outweigh the input map
    static const bool xcd_off = [] { const char* e = getenv("AV2X_W4X3_XCDMAP"); return e && e[0] == '0'; }();
    p.xcd_w = (!xcd_off && p.nblocks > 1 && 8 % p.nblocks == 0 && u_bytes * (unsigned long long)(8 - 8 / p.nblocks) > in_bytes * (unsigned long long)(p.nblocks - 1)) ? 1 : 0;
    const size_t lds = 2ull * 36 * 4 * (32 * 16 + 32);
    const bool general = p.res || (p.relu != 0 && p.relu != 1);
    // round 6: the eight-wave ping-pong form, OPT-IN (AV2X_W4X3_PP=1): same bits, measured SLOWER than the four-wave kernel on every launch
    // class (profiles/r06_wino4_pp.txt: 82.9 vs 64.5 us at 4 x 25 x 88, 470.7 vs 426.8 us at 4 x 100 x 352, headline 474 vs 531 frames/s)
    // (read per launch, ~0.1 us: tests/test_gpu_wino4_x3.py flips it inside one process to compare the two kernels bit for bit)
    const char* ppe = getenv("AV2X_W4X3_PP");
    const bool pp = ppe && ppe[0] == '1';
    if (pp) {
        static LdsLimit lim_ps, lim_pg;
        if (general) {
            lim_pg.ensure(reinterpret_cast<const void*>(&conv_wino4_x3_pp<true>), lds);
            hipLaunchKernelGGL((conv_wino4_x3_pp<true>), dim3(mblocks * p.nblocks), dim3(512), lds, st, p);
        } else {
            lim_ps.ensure(reinterpret_cast<const void*>(&conv_wino4_x3_pp<false>), lds);
            hipLaunchKernelGGL((conv_wino4_x3_pp<false>), dim3(mblocks * p.nblocks), dim3(512), lds, st, p);
        }
        return check_launch("conv_wino4_x3_pp");
    }
    static LdsLimit lim_s, lim_g;
    if (general) {
        lim_g.ensure(reinterpret_cast<const void*>(&conv_wino4_x3<true>), lds);
        hipLaunchKernelGGL((conv_wino4_x3<true>), dim3(mblocks * p.nblocks), dim3(256), lds, st, p);
    } else {
        lim_s.ensure(reinterpret_cast<const void*>(&conv_wino4_x3<false>), lds);
        hipLaunchKernelGGL((conv_wino4_x3<false>), dim3(mblocks * p.nblocks), dim3(256), lds, st, p);
    }
    return check_launch("conv_wino4_x3");
}

}  // namespace av2x

extern "C" uint64_t av2x_wino4_x3_weight_bytes(int32_t cin, int32_t coutp) { return 36ull * (uint64_t)cin * (uint64_t)coutp * 6ull; }

extern "C" int av2x_wino4_x3_pack_weights(const float* w_packed, int32_t cin, int32_t coutp, void* u, av2x_stream_t stream) {
    if (!w_packed || !u) return av2x::fail("av2x_wino4_x3_pack_weights: null argument");
    if (cin <= 0 || coutp <= 0 || cin % 32 || coutp % 64) return av2x::fail("av2x_wino4_x3_pack_weights: cin %% 32 / coutp %% 64");
    const size_t plane = (size_t)cin * coutp;
    hipLaunchKernelGGL(wino4_x3_pack_kernel, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, av2x::as_stream(stream), w_packed,
                       reinterpret_cast<unsigned short*>(u), cin, coutp);
    return av2x::check_launch("wino4_x3_pack_kernel");
}
