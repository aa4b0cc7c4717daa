// Winograd F(2x2, 3x3) convolution with fp32-accurate products on the bf16 matrix cores ("split-3"; tile flag
// 0x40000000 | 0x0400).  Same algebra as conv_wino.inc -- Y = A^T [ (G g G^T) . (B^T d B) ] A, 16 positions, i.e. 16 GEMMs
//     M_pos[tile][cout] = sum_cin V_pos[tile][cin] * U_pos[cin][cout],      pos = 4 xi + nu
// -- but every fp32 operand enters the matrix core as three bf16 terms, x = hi + mid + lo (hi = bf16(x), mid = bf16(x - hi),
// lo = bf16(x - hi - mid): the subtractions are exact, the remainder is <= 2^-24 |x|), and the six partial products of
// relative size >= 2^-16 are accumulated in fp32 on v_mfma_f32_32x32x16_bf16 (lo.hi' + hi.lo' + mid.mid' + mid.hi' + hi.mid'
// + hi.hi'; the dropped ones are <= 2^-23 of the product): 6 bf16 MFMAs of 32 cycles replace the 8 fp32 MFMAs of 64 cycles
// that 16 input channels cost on v_mfma_f32_32x32x2_f32.  The transforms themselves (B^T d B, A^T M A) stay fp32 adds.
//
// gfx950's fp32 matrix pipe runs at 1/16 of the bf16 rate, so with 2.67x fewer matrix cycles everything AROUND the MFMAs
// decides the speed.  The layout is chosen so that no byte is moved or split twice:
//   * a workgroup is 4 waves = 32 MB tiles x 64 couts x all 16 positions; wave nu owns the position COLUMN {4 xi + nu} for
//     ALL of the workgroup's tiles and couts: MB x 2 accumulator tiles of 32 x 32 per position = 128 MB accumulation
//     registers (MB = 2: one wave per SIMD, 256 AGPRs; MB = 1: two workgroups per CU);
//   * V = B^T d B is written to LDS as FP32, [pos][k quad][tile][4] (a thread gathers the 4x4 patch of one tile for 2 MB
//     channels with 16-byte / 8-byte buffer loads, zero outside the image; the lanes that share a tile read the 64 contiguous
//     bytes of a pixel);  every V element is read by exactly ONE wave (the
//     owner of its position column), so the hi / mid / lo split is done on the A fragment in registers, once per element --
//     LDS carries 4 bytes per element each way instead of 6, and a K-16 stage of 64 tiles is 64 KB (double buffered: 128 KB);
//     nine VALU instructions split two values (v_cvt_pk_bf16_f32, shift, and, v_pk_add_f32, twice, and a last convert);
//   * U = G g G^T is split at pack time (av2x_wino_x3_pack_weights, fp64 inside) into [pos][cin/16][plane][k half][coutp][8]
//     bf16: a wave's B fragment for 32 couts is two 512-byte runs, read straight from L2 into registers one position ahead
//     (two register sets), and re-used from registers for the MB tile blocks;
//   * the output transform is separable as in conv_wino_f32_q: A^T over xi lane-local, the Z[a][nu] of the four waves go
//     through LDS (the V buffers, free by then) and every wave finalises a quarter of the workgroup's outputs.
// A chunk of 16 input channels is 48 MB steps of ONE MFMA plus a few side instructions each (split of the next A fragment,
// LDS reads, B loads, the next chunk's gather / transform / LDS stores), pinned by scheduling barriers.
#include <cstdlib>

#include "x3_common.hpp"

// timing experiments only (tools/micro/wx3_ablate.hip builds its OWN binary with this set; the library never defines it): bit 0 no
// hi / mid / lo split, 1 no B fragment loads, 2 no patch gathers, 3 no input transform + LDS stores, 4 no A fragment reads, 5 no barrier
#ifndef AV2X_WX3_ABLATE
#define AV2X_WX3_ABLATE 0
#endif
#ifndef AV2X_WX3_SCHED
#define AV2X_WX3_SCHED 1
#endif
// XCD-aware weight map (round 5 experiment, OFF): an XCD takes ONE cout block and streams 1 / nblocks of the layer's split planes.  Measured
// (profiles/r05_pmc_hbm.json vs r04_pmc_hbm.json): -1.5 us on the in-frame 25 x 88 launches, but the input map is then fetched by 8 / nblocks
// XCDs instead of one and the L2-side traffic goes UP (144 workgroups 63 -> 75 MB, 208: 29 -> 61, 276: 48 -> 88 MB per launch).  Not kept.
#ifndef AV2X_WX3_XCDMAP
#define AV2X_WX3_XCDMAP 0
#endif

namespace {

struct WinoX3Params {
    const float* in;
    const void* u;       // bf16 [16 pos][cin/16][3 planes][2 k halves][coutp][8]
    const float* scale;
    const float* shift;
    const float* res;    // optional: added after the activation, or (relu == 4) multiplied as a gate -- as conv_epilogue does
    float* out;
    int H, W, Cin, in_ctot, in_coff;
    int Cout, CoutP, out_ctot, out_coff, relu;
    int TH, TW, tiles_per_img, T;
    int nblocks, chunks;
    unsigned in_bytes, u_bytes, out_bytes;
};

// MB = 32-tile blocks, NBK = 32-cout blocks per workgroup (MB x NBK x 4 positions = the accumulator tiles of a wave):
//   (2, 2)  64 tiles x  64 couts: 256 accumulation registers, one wave per SIMD; an A fragment (split once) feeds 12 MFMAs
//   (1, 2)  32 tiles x  64 couts: 128 registers, two workgroups per CU
//   (1, 4)  32 tiles x 128 couts: 256 registers; an A fragment feeds 24 MFMAs -- HALF the split / transform instructions per MFMA
//           (2.5 instead of 5: the (2, 2) form is issue-bound at one wave per SIMD), at the price of one B fragment per MFMA pair
template <int MB, int NBK, bool GENERAL>
__global__ __launch_bounds__(256, (MB == 1 && NBK == 2) ? AV2X_X3_MB1_OCC : 1) void conv_wino_x3(const WinoX3Params p) {
    constexpr int TB = 32 * MB;            // tiles per workgroup
    constexpr int CH = 2 * MB;             // channels gathered per thread (256 threads = TB tiles x 16 / CH channel groups)
    constexpr int NG = 4 * MB;             // (xi, tile block) groups per chunk and wave; GSZ MFMAs each
    constexpr int GSZ = 6 * NBK;           // six partial products x NBK cout blocks per A fragment
    constexpr int SPL = NBK / 2;           // the 12 split steps of the next fragment go to every SPL-th step of a group
    constexpr int STEPS = GSZ * NG;
    constexpr bool SPREAD = MB == 2 && NBK == 2 && AV2X_WX3_SCHED == 1;
    constexpr int NCG = 16 / CH;           // channel groups per tile and chunk
    constexpr int KQS = TB * 16 + 32;      // bytes per k quad: [tile TB][4 floats] + 32 (the pad keeps the gather threads' stores conflict-free)
    constexpr int POSB = 4 * KQS;          // bytes per position in a V stage: [k quad 4][tile TB][4 floats]
    constexpr int VSTAGE = 16 * POSB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // V[2][16][4][TB][4] fp32; re-used for the exchange

    const int tid = threadIdx.x, lane = tid & 63;
    const int nu = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave = column of the 4 x 4 position grid
    const int nbk = gridDim.x, b = blockIdx.x;
    const int q8 = nbk >> 3, r8 = nbk & 7, xcd = b & 7;
    int mblock, nblock;
    if (AV2X_WX3_XCDMAP && (8 % p.nblocks) == 0) {
        // XCD-aware map for the WEIGHT stream (workgroup b runs on XCD b % 8, observed; speed only): XCD x takes cout block x % nblocks for
        // every (8 / nblocks)-th tile block, so an XCD's L2 streams 1 / nblocks of the layer's split planes (1.6 MB of the 6.3 MB of a
        // 256 -> 256 layer: it fits the 4 MB L2, and the cold fetch from HBM / Infinity Cache at the head of every in-frame launch drops
        // from 8 x 6.3 to 8 x 1.6 MB) at the price of 1 / (8 / nblocks) instead of 1 / 8 of the -- just written, cache-warm -- input map.
        // The hardware's workgroups-per-XCD counts (q8 + (x < r8)) are exactly the counts this assignment needs.
        const int G = 8 / p.nblocks;
        nblock = xcd % p.nblocks;
        mblock = xcd / p.nblocks + (b >> 3) * G;
    } else {
        const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
        mblock = swz / p.nblocks;
        nblock = swz - mblock * p.nblocks;
    }
    const int t0 = mblock * TB;
    const int n0 = nblock * (32 * NBK);

    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.u), 0, p.u_bytes, 0x00020000);

    // ---- weight side: plane (pos, kb, pl) is [2 k halves][coutp][8 bf16]; lane (j = lane & 31, kh = lane >> 5) reads 16 bytes
    const unsigned voffU = (unsigned)(((lane >> 5) * p.CoutP + n0 + (lane & 31)) * 16);
    const int plane_stride = 32 * p.CoutP;                           // bytes
    const int kb_stride = 3 * plane_stride, pos_stride = (p.Cin >> 4) * kb_stride;
    x3_u32x4 bs[2][NBK][3];                                         // [set][32-cout block][plane]
    auto load_b = [&](auto set_c, auto i_c, int xi, int kb) {       // i = 3 nb + plane
        constexpr int set = decltype(set_c)::value, i = decltype(i_c)::value;
        bs[set][i / 3][i % 3] = __builtin_bit_cast(x3_u32x4, __builtin_amdgcn_raw_buffer_load_b128(
            ru, voffU + (i / 3) * 512, (4 * xi + nu) * pos_stride + kb * kb_stride + (i % 3) * plane_stride, 0));
    };
    x3_static_for<0, 3 * NBK>([&](auto i) { load_b(std::integral_constant<int, 0>{}, i, 0, 0); });

    f32x16 acc[4][MB][NBK];
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[xi][mb][nb][r] = 0.f;

    // ---- input side: this thread's tile and channel group; the 16 byte offsets of its 4x4 patch are chunk-invariant
    // (0x80000000 = outside the image or beyond the last tile: the buffer unit returns zero)
    // NCG consecutive lanes read the 64 contiguous bytes of one pixel (a wave-load touches 64 / NCG cache lines, not 64)
    const int tl = tid / NCG, cg = tid % NCG;                        // channels cg CH .. cg CH + CH - 1 of the chunk
    unsigned voffI[16];
    {
        const int t = t0 + tl;
        const bool tok = t < p.T;
        const int tt = tok ? t : 0;
        const int img = tt / p.tiles_per_img, r = tt - img * p.tiles_per_img;
        const int ty = r / p.TW, tx = r - ty * p.TW;
        const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
        const int base = (((img * p.H + y0) * p.W + x0) * p.in_ctot + p.in_coff + cg * CH) * 4;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = tok && (unsigned)(y0 + a) < (unsigned)p.H && (unsigned)(x0 + e) < (unsigned)p.W;
                voffI[a * 4 + e] = ok ? (unsigned)(base + (a * p.W + e) * p.in_ctot * 4) : 0x80000000u;
            }
    }
    x3_f32x2 d[16][MB];                                              // channel pairs: v_pk_add_f32 operands
    auto gather = [&](auto k_c, int c) {
        constexpr int k = decltype(k_c)::value;
        if constexpr (MB == 2) {
            const x3_pair2 v = __builtin_bit_cast(x3_pair2, __builtin_amdgcn_raw_buffer_load_b128(rin, voffI[k], c * 64, 0));
            d[k][0] = v.a; d[k][1] = v.b;
        } else {
            d[k][0] = __builtin_bit_cast(x3_f32x2, __builtin_amdgcn_raw_buffer_load_b64(rin, voffI[k], c * 64, 0));
        }
    };
    // B^T d, column e of the patch and channel pair h, in place: rows 0..3 <- d0 - d2, d1 + d2, d2 - d1, d1 - d3.  Unit u = 2 (MB e + h) + half:
    // half 0 makes rows 0 and 3, half 1 rows 1 and 2 (two v_pk_add_f32 each)
    auto rows = [&](auto u_c) {
        constexpr int u = decltype(u_c)::value, half = u & 1, h = (u >> 1) % MB, e = (u >> 1) / MB;
        if constexpr (half == 0) {
            asm volatile("" : "+v"(d[e][h]), "+v"(d[4 + e][h]), "+v"(d[8 + e][h]), "+v"(d[12 + e][h]));   // keeps the wait for the patch HERE
            d[e][h] = x3_pk_sub(d[e][h], d[8 + e][h]);
            d[12 + e][h] = x3_pk_sub(d[4 + e][h], d[12 + e][h]);
        } else {
            const x3_f32x2 r1 = x3_pk_add(d[4 + e][h], d[8 + e][h]);
            d[8 + e][h] = x3_pk_sub(d[8 + e][h], d[4 + e][h]);
            d[4 + e][h] = r1;
        }
    };
    // element (a, j) of (B^T d) B -> V[pos = 4 a + j][k quad][tile][.]
    const unsigned wbase = (unsigned)(((cg * CH) >> 2) * KQS + tl * 16 + ((cg * CH) & 3) * 4);
    auto cols = [&](auto o_c, int stage) {
        constexpr int o = decltype(o_c)::value, a = o >> 2, j = o & 3;
        x3_f32x2 v[MB];
#pragma unroll
        for (int h = 0; h < MB; ++h)
            v[h] = j == 0 ? x3_pk_sub(d[a * 4 + 0][h], d[a * 4 + 2][h])
                 : j == 1 ? x3_pk_add(d[a * 4 + 1][h], d[a * 4 + 2][h])
                 : j == 2 ? x3_pk_sub(d[a * 4 + 2][h], d[a * 4 + 1][h])
                          : x3_pk_sub(d[a * 4 + 1][h], d[a * 4 + 3][h]);
        if constexpr (MB == 2) {
            x3_pair2 w; w.a = v[0]; w.b = v[1];
            *reinterpret_cast<f32x4*>(smem + stage * VSTAGE + o * POSB + wbase) = __builtin_bit_cast(f32x4, w);
        } else {
            *reinterpret_cast<x3_f32x2*>(smem + stage * VSTAGE + o * POSB + wbase) = v[0];
        }
    };

    // A fragment of group g = (xi, mb): lane (i = lane & 31, kh = lane >> 5) reads k quads 2 kh, 2 kh + 1 of tile 32 mb + i
    const unsigned rbase = (unsigned)((2 * (lane >> 5)) * KQS + (lane & 31) * 16) + nu * POSB;
    x3_f32x2 raw[2][4];
    unsigned pl[2][3][4];
    X3Split sp;
    auto read_a = [&](auto buf_c, auto half_c, int g, int stage) {
        constexpr int buf = decltype(buf_c)::value, half = decltype(half_c)::value;
        const int xi = g / MB, mb = g % MB;
        const x3_pair2 v = __builtin_bit_cast(x3_pair2, *reinterpret_cast<const f32x4*>(smem + stage * VSTAGE + rbase + xi * 4 * POSB + mb * 512 + half * KQS));
        raw[buf][2 * half] = v.a; raw[buf][2 * half + 1] = v.b;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // prologue: chunk 0 -> stage 0, the patch of chunk 1 in flight, the first two A fragments read and the first one split
    x3_static_for<0, 16>([&](auto k) { gather(k, 0); });
    x3_static_for<0, 8 * MB>([&](auto u) { rows(u); });
    x3_static_for<0, 16>([&](auto o) { cols(o, 0); });
    {
        const int c1 = min(1, p.chunks - 1);
        x3_static_for<0, 16>([&](auto k) { gather(k, c1); });
    }
    __syncthreads();
    read_a(I0{}, I0{}, 0, 0);
    read_a(I0{}, I1{}, 0, 0);
    read_a(I1{}, I0{}, 1, 0);
    read_a(I1{}, I1{}, 1, 0);
    x3_static_for<0, 12>([&](auto jj) { x3_split_step<decltype(jj)::value>(raw[0], pl[0], sp); });

    // Steady state, per chunk c (stage st = c & 1 holds V(c)):
    //   steps RS ..        B^T d of the patch of chunk c + 1 (gathered during chunk c - 1), one unit per step
    //   steps WS0 ..       (B^T d) B and the LDS stores of V(c + 1) into the other stage
    //   step  BST          the ONE barrier of the chunk: V(c + 1) complete, and nobody reads stage st any more (the last A
    //                      fragment of chunk c was read in group NG - 3)
    //   groups NG-2, NG-1  read the first two A fragments of chunk c + 1; group NG-1 splits the first one -> no bubble between chunks
    //   steps GS ..        the patch of chunk c + 2 (its registers are free once V(c + 1) is stored)
    // (1, 4): the 12 B fragments of the next position go out in the first half of a group and the patch of chunk c + 2 in the second half
    // of group 2 (+ 4 steps): the wait for the B fragments at the next group's first MFMA then leaves the younger patch loads in flight.
    constexpr int BST = GSZ * (NG - 2) - 1;
    constexpr int RS = MB == 2 ? 12 : 0, WS0 = RS + 8 * MB + (NBK == 4 ? 1 : 0), WST = (MB == 2 || NBK == 4) ? 2 : 1;
    constexpr int GS = NBK == 4 ? 60 : MB == 2 ? 60 : 24;
    static_assert(WS0 + 15 * WST <= BST && GS + 16 <= STEPS && GS >= WS0 + 15 * WST, "schedule");
    for (int c = 0; c < p.chunks; ++c) {
        // branch-free body: V(c + 1) is always written (the last pass stores the re-fetched last chunk into the idle stage) and
        // the patch / weights of chunk min(c + 2, chunks - 1) / min(c + 1, chunks - 1) are always fetched: exact wait counters
        const int c1 = min(c + 1, p.chunks - 1), c2 = min(c + 2, p.chunks - 1);
        const int st = c & 1;
        x3_static_for<0, STEPS>([&](auto ss) {
            constexpr int s = decltype(ss)::value;
            constexpr int g = s / GSZ, j = s % GSZ, xi = g / MB, mb = g % MB, pr = j / NBK, nb = j % NBK;
            acc[xi][mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                x3_frag(pl[g & 1][x3_ap(pr)]), __builtin_bit_cast(x3_bf16x8, bs[xi & 1][nb][x3_bp(pr)]),
                acc[xi][mb][nb], 0, 0, 0);
            constexpr int AB = AV2X_WX3_ABLATE;
            // split of the next group's A fragment (the last group: the first fragment of the next chunk)
            if constexpr (j % SPL == 0 && !(AB & 1)) x3_split_step<j / SPL>(raw[(g + 1) & 1], pl[(g + 1) & 1], sp);
            if constexpr (j == 0 && (AB & 1)) asm volatile("" :: "v"(raw[(g + 1) & 1][0]), "v"(raw[(g + 1) & 1][1]), "v"(raw[(g + 1) & 1][2]), "v"(raw[(g + 1) & 1][3]));
            // raw A fragment of group g + 2 (of the next chunk from group NG - 2 on) into the buffer the split of group g has released
            if constexpr ((j == 4 * SPL || j == 9 * SPL) && !(AB & 16)) {
                using HALF = std::integral_constant<int, j == 4 * SPL ? 0 : 1>;
                if constexpr (g + 2 < NG) read_a(std::integral_constant<int, g & 1>{}, HALF{}, g + 2, st);
                else read_a(std::integral_constant<int, g & 1>{}, HALF{}, g + 2 - NG, st ^ 1);
            }
            // B fragments of the NEXT position into the other register set (its last reader was position xi - 1)
            if constexpr (SPREAD) {
                // (2, 2), spread form: the vector-memory path (64 B/clk per CU: a 1-KB wave-load is 16 cycles of it, 160 KB per chunk = 2 560
                // of the chunk's 3 072 matrix cycles) is what the four in-order waves block on, so no two memory instructions share a step and
                // no burst fills the queue: B fragments on the odd steps 1..11 of a position, patch loads on its odd steps 13..23 -- AFTER the
                // B loads, so that the wait for the B fragments at the next position's first MFMA leaves them in flight --, LDS stores on even
                // steps.  Patch row a = k >> 2 of chunk c + 2 is loaded as soon as (B^T d) B of that row of chunk c + 1 has been stored.
                constexpr int sl = s % (2 * GSZ);
                if constexpr ((sl & 1) && sl < 12 && !(AB & 2)) {
                    using IB = std::integral_constant<int, sl / 2>;
                    if constexpr (xi < 3) load_b(std::integral_constant<int, (xi + 1) & 1>{}, IB{}, xi + 1, c);
                    else load_b(I0{}, IB{}, 0, c1);
                }
                constexpr int gk = xi == 1 ? (sl - 13) / 2 : xi == 2 ? 4 + (sl - 13) / 2 : xi == 3 ? 10 + (sl - 13) / 2 : -1;
                if constexpr ((sl & 1) && sl >= 13 && gk >= 0 && gk < (xi == 1 ? 4 : xi == 2 ? 10 : 16) && !(AB & 4))
                    gather(std::integral_constant<int, gk>{}, c2);
            } else {
                constexpr bool bld = NBK == 4 ? (j < 12) : (j == 1 || j == 2 || j == 3 || j == 6 || j == 7 || j == 8);
                if constexpr (mb == 0 && bld && !(AB & 2)) {
                    using IB = std::integral_constant<int, NBK == 4 ? j : (j < 4 ? j - 1 : j - 3)>;
                    if constexpr (xi < 3) load_b(std::integral_constant<int, (xi + 1) & 1>{}, IB{}, xi + 1, c);
                    else load_b(I0{}, IB{}, 0, c1);
                }
                if constexpr (s >= GS && s < GS + 16 && !(AB & 4)) gather(std::integral_constant<int, s - GS>{}, c2);
            }
            if constexpr (s >= RS && s < RS + 8 * MB && !(AB & 8)) rows(std::integral_constant<int, s - RS>{});
            if constexpr (s >= WS0 && s <= WS0 + 15 * WST && (s - WS0) % WST == 0 && !(AB & 8)) cols(std::integral_constant<int, (s - WS0) / WST>{}, st ^ 1);
            if constexpr (s == BST && !(AB & 32)) x3_lds_barrier();   // LDS traffic only: the register prefetches stay in flight
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    __syncthreads();
    if constexpr ((AV2X_WX3_ABLATE & 64) != 0) {   // timing only: no exchange, no output transform, one store per lane
        float t = 0.f;
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) t += acc[xi][mb][nb][r];
        p.out[(size_t)b * 256 + tid] = t;
        return;
    }

    // ---- Z[a] = A^T over xi (lane-local): Z[0] = (M0 + M1) + M2, Z[1] = (M1 - M2) - M3.  Every wave publishes its (Z[0], Z[1]) pairs
    // through LDS as X[nu][slot = (mb NBK + nb) 16 + r][lane][a] (one ds_write_b64 per accumulator row).  The finalising side reads them
    // COUT-MAJOR: four consecutive lanes of a slot are four consecutive couts of one tile, so a thread takes (tile, cout quad) items --
    // two ds_read_b128 per position column -- computes y[a][0] = (Z0 + Z1) + Z2, y[a][1] = (Z1 - Z2) - Z3 over the four waves' Z for
    // its four couts and stores each of the 2 x 2 output pixels as ONE 16-byte store; 16 (32) consecutive lanes write the 256 (512)
    // contiguous bytes of a pixel.  (Round 4 stored one dword per lane and pixel: 64 store instructions per lane, 6.9 of a 49-us launch.)
    float* X = reinterpret_cast<float*>(smem);
    constexpr int NS = MB * NBK * 16;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                x3_f32x2 zz;
                zz.x = acc[0][mb][nb][r] + acc[1][mb][nb][r] + acc[2][mb][nb][r];
                zz.y = acc[1][mb][nb][r] - acc[2][mb][nb][r] - acc[3][mb][nb][r];
                *reinterpret_cast<x3_f32x2*>(X + ((nu * NS + (mb * NBK + nb) * 16 + r) * 64 + lane) * 2) = zz;
            }
    __syncthreads();
    constexpr int CQ = 8 * NBK;                    // cout quads of the workgroup
    constexpr int RW = 4 * MB;                     // accumulator rows (of the 16 MB row slots) a wave finalises
    constexpr int ITS = RW * 2 * CQ / 64;          // items per lane
    const int q = lane % CQ, fnb = q >> 3, j4 = (q & 7) * 4;
    const int n = n0 + fnb * 32 + j4;
    const bool nok = n < p.Cout;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (nok) {
        if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n);
        sh = *reinterpret_cast<const f32x4*>(p.shift + n);
    }
    const bool relu1 = p.relu == 1;
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    const unsigned ocol = (unsigned)((p.out_coff + n) * 4);
    const int opix = p.out_ctot * 4;
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        const int idx = lane / CQ + (64 / CQ) * it;        // (row of the wave's RW, half-wave of the accumulator tile)
        const int gr = nu * RW + (idx >> 1), kh = idx & 1;
        const int fmb = gr >> 4, r = gr & 15;
        // row r of a 32 x 32 accumulator tile is tile (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the block
        const int t = t0 + 32 * fmb + (r & 3) + 8 * (r >> 2) + 4 * kh;
        const int img = t / p.tiles_per_img, rem = t - img * p.tiles_per_img;
        const int ty = rem / p.TW, tx = rem - ty * p.TW;
        f32x4 z[4][2];                                      // [position column v][4 couts x (Z0, Z1)]
        const float* xs = X + (((fmb * NBK + fnb) * 16 + r) * 64 + kh * 32 + j4) * 2;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            z[v][0] = *reinterpret_cast<const f32x4*>(xs + v * NS * 128);
            z[v][1] = *reinterpret_cast<const f32x4*>(xs + v * NS * 128 + 4);
        }
        f32x4 y[2][2];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int w = cc >> 1, e = (cc & 1) * 2 + a;
                y[a][0][cc] = (z[0][w][e] + z[1][w][e]) + z[2][w][e];
                y[a][1][cc] = (z[1][w][e] - z[2][w][e]) - z[3][w][e];
            }
        const bool tvalid = nok && t < p.T;
        const int py = 2 * ty, px = 2 * tx;
        const unsigned pix = (unsigned)(((img * p.H + py) * p.W + px) * opix) + ocol;    // < 2^31 (checked by the host)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const bool ok = tvalid && py + a < p.H && px + e < p.W;
                const unsigned off = ok ? pix + (unsigned)((a * p.W + e) * opix) : 0x80000000u;
                f32x4 v = y[a][e] * sc + sh;
                if (GENERAL) {
                    f32x4 rs = {0.f, 0.f, 0.f, 0.f};
                    if (p.res && ok) {
                        const size_t m = (size_t)(off - ocol) / (size_t)opix;
                        rs = *reinterpret_cast<const f32x4*>(p.relu == 4 ? p.res + m * p.Cout + n : p.res + m * p.out_ctot + p.out_coff + n);
                    }
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        float u = v[cc];
                        if (p.relu == 1) u = fmaxf(u, 0.f);
                        else if (p.relu == 3) u = 1.0f / (1.0f + expf(-u));
                        else if (p.relu == 4) u = tanhf(u);
                        if (p.res && ok) u = (p.relu == 4) ? u * rs[cc] : u + rs[cc];
                        if (p.relu == 5) u = fmaxf(u, 0.f);                  // ReLU after the residual add (ResNet BasicBlock)
                        v[cc] = u;
                    }
                } else {
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) v[cc] = relu1 ? fmaxf(v[cc], 0.f) : v[cc];
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(x3_u32x4, v), rout, off, 0, 0);
            }
    }
}

// U = G g G^T per (cin, cout) in fp64, split into hi / mid / lo bf16 (round to nearest even at every step):
// w packed [tap][cin/4][coutp][4] fp32  ->  u [pos][cin/16][plane][k half][coutp][8] bf16

__global__ void wino_x3_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ u, int cin, int coutp) {
    const size_t plane = (size_t)cin * coutp;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;    // index into [cin/4][coutp][4]
    if (e >= plane) return;
    const int k4 = (int)(e & 3), co = (int)((e >> 2) % coutp), q = (int)((e >> 2) / coutp);
    const int k = 4 * q + k4, kb = k >> 4, kh = (k >> 3) & 1, k8 = k & 7;
    double g[3][3], t[4][3], U[16];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) g[i][j] = (double)w[(size_t)(i * 3 + j) * plane + e];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        t[0][j] = g[0][j];
        t[1][j] = 0.5 * (g[0][j] + g[1][j] + g[2][j]);
        t[2][j] = 0.5 * (g[0][j] - g[1][j] + g[2][j]);
        t[3][j] = g[2][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        U[i * 4 + 0] = t[i][0];
        U[i * 4 + 1] = 0.5 * (t[i][0] + t[i][1] + t[i][2]);
        U[i * 4 + 2] = 0.5 * (t[i][0] - t[i][1] + t[i][2]);
        U[i * 4 + 3] = t[i][2];
    }
    const size_t kbs = (size_t)(cin >> 4);
#pragma unroll
    for (int pos = 0; pos < 16; ++pos) {
        // the fp32 value the fp32 kernels would use is (float)U; splitting the fp64 value keeps a few more bits in mid / lo
        const float h = __builtin_bit_cast(float, (unsigned)x3_bf16_rne((float)U[pos]) << 16);
        const double r1 = U[pos] - (double)h;
        const float m = __builtin_bit_cast(float, (unsigned)x3_bf16_rne((float)r1) << 16);
        const double r2 = r1 - (double)m;
        const unsigned short hs = (unsigned short)(__builtin_bit_cast(unsigned, h) >> 16);
        const unsigned short ms = (unsigned short)(__builtin_bit_cast(unsigned, m) >> 16);
        const unsigned short ls = x3_bf16_rne((float)r2);
        const size_t base = ((((size_t)pos * kbs + kb) * 3) * 2 + kh) * (size_t)coutp * 8 + (size_t)co * 8 + k8;
        const size_t pstride = (size_t)2 * coutp * 8;
        u[base] = hs;
        u[base + pstride] = ms;
        u[base + 2 * pstride] = ls;
    }
}

template <int MB, int NBK>
static int launch_wino_x3(const WinoX3Params& p0, hipStream_t st) {
    WinoX3Params p = p0;
    constexpr int TB = 32 * MB;
    p.nblocks = p.Cout / (32 * NBK);
    const int mblocks = (p.T + TB - 1) / TB;
    size_t lds = 2ull * 16 * 4 * (TB * 16 + 32);
    if (lds < 4ull * (MB * NBK * 32) * 64 * 4) lds = 4ull * (MB * NBK * 32) * 64 * 4;    // the output exchange: [wave][slot][lane] fp32
    if (const char* pad = getenv("AV2X_X3_LDS_PAD")) lds += (size_t)atoi(pad);   // debug probe (tools/debug/dbg_attn.py)
    const bool general = p.res || (p.relu != 0 && p.relu != 1);
    static av2x::LdsLimit lim_s, lim_g;
    if (general) {
        lim_g.ensure(reinterpret_cast<const void*>(&conv_wino_x3<MB, NBK, true>), lds);
        hipLaunchKernelGGL((conv_wino_x3<MB, NBK, true>), dim3(mblocks * p.nblocks), dim3(256), lds, st, p);
    } else {
        lim_s.ensure(reinterpret_cast<const void*>(&conv_wino_x3<MB, NBK, false>), lds);
        hipLaunchKernelGGL((conv_wino_x3<MB, NBK, false>), dim3(mblocks * p.nblocks), dim3(256), lds, st, p);
    }
    return av2x::check_launch("conv_wino_x3");
}

}  // namespace

namespace av2x {

// called by av2x_conv2d* for tile flag 0x40000000 | 0x0400 (conv_igemm.hip); `u` is the packing of av2x_wino_x3_pack_weights
int wino_x3_dispatch(const av2x_conv_desc* d, const float* in, const void* u, const float* scale, const float* shift,
                     const float* residual, float* out, hipStream_t st) {
    if (d->mode != AV2X_CONV || d->ks != 3 || d->stride != 1 || d->pad != 1 || d->ho != d->h || d->wo != d->w)
        return fail("av2x_conv2d: the Winograd tiles take 3x3 / stride 1 / pad 1 NHWC convolutions only");
    if (d->relu != 0 && d->relu != 1 && d->relu != 3 && d->relu != 4 && d->relu != 5)
        return fail("av2x_conv2d: the Winograd tiles fuse no activation, ReLU (before / after the residual), sigmoid or tanh (relu=%d)", d->relu);
    if (d->cin % 16 || d->cout % 64 || d->coutp % 64 || d->cout > d->coutp)
        return fail("av2x_conv2d: split-3 Winograd needs cin %% 16 == 0 and cout %% 64 == 0 (cin=%d cout=%d)", d->cin, d->cout);
    if (d->in_coff % 4 || d->in_ctot % 4) return fail("av2x_conv2d: input channel offset/stride must be multiples of 4");
    if (d->out_coff % 4 || d->out_ctot % 4) return fail("av2x_conv2d: split-3 Winograd stores 16-byte cout quads: output channel offset/stride must be multiples of 4");
    WinoX3Params p;
    p.in = in; p.u = u; p.scale = scale; p.shift = shift; p.res = residual; p.out = out;
    p.H = d->h; p.W = d->w; p.Cin = d->cin; p.in_ctot = d->in_ctot; p.in_coff = d->in_coff;
    p.Cout = d->cout; p.CoutP = d->coutp; p.out_ctot = d->out_ctot; p.out_coff = d->out_coff; p.relu = d->relu;
    p.TH = (d->h + 1) / 2; p.TW = (d->w + 1) / 2; p.tiles_per_img = p.TH * p.TW;
    const long long T = (long long)d->n * p.tiles_per_img;
    if (T <= 0) return 0;
    if (T > (1ll << 28)) return fail("av2x_conv2d: too many Winograd tiles");
    p.T = (int)T;
    p.chunks = d->cin / 16;
    p.nblocks = 0;
    const unsigned long long in_bytes = (unsigned long long)d->n * d->h * d->w * d->in_ctot * 4ull;
    const unsigned long long u_bytes = 16ull * d->cin * d->coutp * 6ull;
    if (in_bytes >= (1ull << 31) || u_bytes >= (1ull << 31))
        return fail("av2x_conv2d: input (%llu B) or transformed weights (%llu B) exceed the 2 GiB buffer-descriptor window", in_bytes, u_bytes);
    p.in_bytes = (unsigned)in_bytes;
    p.u_bytes = (unsigned)u_bytes;
    const unsigned long long out_bytes = (unsigned long long)d->n * d->h * d->w * d->out_ctot * 4ull;
    if (out_bytes >= (1ull << 31)) return fail("av2x_conv2d: output (%llu B) exceeds the 2 GiB buffer-descriptor window", out_bytes);
    p.out_bytes = (unsigned)out_bytes;
    const int tb = (d->tile >> 16) & 0x3fff, cb = d->tile & 0x01ff;
    if (tb == 64 && cb == 64) return launch_wino_x3<2, 2>(p, st);
    if (tb == 32 && cb == 64) return launch_wino_x3<1, 2>(p, st);
    if (tb == 32 && cb == 128 && d->cout % 128 == 0) return launch_wino_x3<1, 4>(p, st);
    return fail("av2x_conv2d: the split-3 Winograd tiles are 64 x 64, 32 x 64 and 32 x 128 (tile %dx%d, cout %d)", tb, cb, d->cout);
}

}  // namespace av2x

extern "C" uint64_t av2x_wino_x3_weight_bytes(int32_t cin, int32_t coutp) { return 16ull * (uint64_t)cin * (uint64_t)coutp * 6ull; }

extern "C" int av2x_wino_x3_pack_weights(const float* w_packed, int32_t cin, int32_t coutp, void* u, av2x_stream_t stream) {
    if (!w_packed || !u) return av2x::fail("av2x_wino_x3_pack_weights: null argument");
    if (cin <= 0 || coutp <= 0 || cin % 16 || coutp % 64) return av2x::fail("av2x_wino_x3_pack_weights: cin %% 16 / coutp %% 64");
    const size_t plane = (size_t)cin * coutp;
    hipLaunchKernelGGL(wino_x3_pack_kernel, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, av2x::as_stream(stream), w_packed,
                       reinterpret_cast<unsigned short*>(u), cin, coutp);
    return av2x::check_launch("wino_x3_pack_kernel");
}
