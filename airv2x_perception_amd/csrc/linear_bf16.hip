// Token Linear layers of the transformer fusion heads with bf16 ACTIVATIONS in HBM (the storage torch.autocast gives the
// outputs of nn.Linear / matmul; reference tools/train.py:118, inference_utils under autocast): SURVEY 8a a15 in AMP mode.
//
//   linear_bf16_kernel   out[m][n] = act(sum_k A[m][k] W[k][n] + bias[n]) (+ residual[m][n])     K = 256
//       A   (M, 256) bf16 row-major tokens (NHWC buffer viewed as rows)
//       W   bf16 k-oct packing [K/8][CoutP][8] of packing.to_bf16_koct with the COLUMNS of every 64-group interleaved
//           (packing.interleave2_columns): packed column 32 c + i of a group holds logical column 2 i + c, so that an MFMA
//           lane owns TWO CONSECUTIVE output columns of a row (one 4-byte bf16 / 8-byte fp32 store, 128 / 256 contiguous
//           bytes per row and half-wave) while its B fragments stay fully coalesced 16-byte loads
//       out bf16 (next Linear / attention input) or fp32 (+ fp32 residual: the residual stream x stays fp32, as under autocast
//           where LayerNorm outputs and `x + fn(x)` are fp32)
//
// These layers are HBM-bound on MI355X (K = 256: 64-128 flop per byte moved at 2.5 PFLOP/s bf16), so the kernels are built
// around bytes, not MFMA issue: a workgroup owns a PANEL of tokens, pulls it into LDS once with all its loads in flight
// together, then walks over the output columns in chunks of 256 reading the weights straight from L2 into registers
// (<= 1.2 MB per layer: L2-resident), so A is read from HBM exactly once whatever the output width (1280 for the HGT
// projections, 2304 for the three window-attention QKVs).
//   v_mfma_f32_32x32x16_bf16: lane (i = lane & 31, h = lane >> 5) supplies k = 8 h + 0..7 of row / column i
//   C/D map: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
//
//   layernorm_bf16_kernel   nn.LayerNorm over C = 256 of the fp32 residual stream, bf16 out (the A operand above)
#include <cstdlib>
#include <type_traits>

#include "av2x_common.hpp"
#include "split_attn_rows.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct LinParams {
    const __bf16* a;
    const __bf16* w;
    const float* bias;
    const float* res;
    void* out;
    long long M;
    int Cout, CoutP, out_ctot, out_coff, res_ctot, res_coff, act;
    unsigned w_bytes;
    // LayerNorm fused into the panel load (linear_bf16_occ_kernel<true, .>): the fp32 residual stream, the pending bf16 residual of the
    // first add_rows rows, the affine parameters
    float* x;
    const __bf16* delta;
    long long add_rows;
    int write_back;       // 0: x + delta feeds the LayerNorm but x is left as it is (the add stays pending for a later kernel)
    const float* gamma;
    const float* beta;
    float eps;
    // second Linear 256 -> 256 on the LDS-resident hidden panel (linear_bf16_occ_kernel<., true>: the FeedForward pair)
    const __bf16* w2;
    const float* bias2;
    int act2;
    // SplitAttn's combine as the producer of x (linear_bf16_occ_kernel<SRC_LNC, .>): x = x (+ delta) + sum_b softmax_b(logits) br_b,
    // written back, then LayerNorm -> Linear(s): the three bf16 branch maps, the (n, 3, 256) logits, tokens per agent
    const __bf16* br[3];
    const float* logits;
    long long hw;
    // window attention as the panel source (linear_bf16_occ_kernel<2 | 3, .>): the bf16 [q | k | v] rows, the relative-position table, the map
    const __bf16* qkv;
    const float* pos;
    int qkv_ctot, qkv_coff, H, W, heads;
    float scale;
};

__device__ __forceinline__ float4 ld_bf16x4(const __bf16* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u), __builtin_bit_cast(float, u.y << 16),
                       __builtin_bit_cast(float, u.y & 0xffff0000u));
}

// nn.LayerNorm over C = 256 of one token held as four channels per lane of a wave (the ONE definition both the stand-alone
// LayerNorm kernel and the fused panel load use: their outputs are the same bits)
__device__ __forceinline__ f32x4 layernorm_row_256(const float4 v, const float4 g, const float4 bt, const float eps) {
    constexpr int C = 256;
    float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
    float q = (a * a + b * b) + (c * c + d * d);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    const f32x4 r = {a * rstd * g.x + bt.x, b * rstd * g.y + bt.y, c * rstd * g.z + bt.z, d * rstd * g.w + bt.w};
    return r;
}

__device__ __forceinline__ float4 add_bf16x4(float4 v, const uint2 u) {
    v.x += __builtin_bit_cast(float, u.x << 16); v.y += __builtin_bit_cast(float, u.x & 0xffff0000u);
    v.z += __builtin_bit_cast(float, u.y << 16); v.w += __builtin_bit_cast(float, u.y & 0xffff0000u);
    return v;
}

// cache policy of the streamed operands (A panel loads, output stores): non-temporal, so that 1.3 GB of output passing through
// the 4 MB L2 of an XCD does not evict the 1.2 MB of weights every workgroup re-reads
constexpr int LIN_NT = 2;
constexpr int LK = 256, LBM = 128, LROW = LK + 8;   // LDS row stride in bf16 elements: 528 B, conflict-free ds_read_b128

// erf to ~1.5e-7 absolute (Abramowitz & Stegun 7.1.26): the GELU outputs of this kernel are rounded to bf16 (2^-9 relative), and
// erff() costs more VALU time per chunk than the chunk's MFMAs
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float y = 1.0f - poly * t * __expf(-ax * ax);
    return copysignf(y, x);
}

// ---- bf16-out form (what the V2X-ViT engine calls): panels of 64 tokens (33 KB of LDS), 64 accumulator registers per wave ->
// four workgroups = 16 waves per CU; a wave's W fragments come straight from L2 (two 16-byte loads per K-step, four steps of
// prefetch), the epilogue rounds once to bf16, transposes 4 x 4 inside every lane quad (DPP) and stores 16 bytes per lane
// (8 rows x 128 B per instruction: 5.3 TB/s on this pattern against 3.9 TB/s for the 4-byte stores of the raw MFMA layout,
// tools/micro/hbm_bw.hip).
//   wave w of 4: all 64 rows (two MFMA row tiles), columns 64 w + [0, 64) of the chunk (two tiles, interleaved pairs)
// Measured on MI355X, 281 600 tokens (tools/lin16_bench.py): 256 -> 2304 (1.44 GB moved) 555 us, 256 -> 1280 350 us, 256 -> 256 80 us
// = 2.5-3.7 TB/s of algorithmic bytes.  The time is close to (W-fragment phase) + (store phase): with the stores compiled out
// the 2304-column layer takes 254 us (2.7 GB of W fragments from L2 = 10.8 TB/s of L2->CU traffic, MFMA pipe 60 % busy), the
// stores alone take 245 us in tools/micro/hbm_bw.hip, and neither 128-token panels with two workgroups per CU, nor issuing the
// stores of chunk i between the MFMAs of chunk i + 1 (the vector-memory counter is in order over loads AND stores: the next
// W fragment cannot be consumed before the older stores are acknowledged), nor non-temporal hints changed the sum.  Next
// step (not done): W tiles shared through LDS by an 8-wave workgroup to cut the L2->CU traffic four-fold.
//
// LN = true: the panel is LayerNorm(x (+ delta on the first add_rows rows)) of the fp32 residual stream, computed while it is loaded
// (wave w: rows 16 w .. 16 w + 15, four channels per lane, layernorm_row_256) and x is written back where delta was added: the
// normalised bf16 tensor never exists in HBM (288 MB written + read per LayerNorm at 8 agents).
// FFN = true (Cout = CoutP = 256): the activated bf16 output panel goes back into the SAME LDS panel (all waves are past their K
// loop) and a second Linear 256 -> 256 (w2, bias2, act2) runs on it: FeedForward's hidden tensor never exists in HBM either.
// Both produce exactly the bits of the separate launches (same MFMA sequence per output, same roundings).
// SRC = 2 / 3: the panel is the OUTPUT of the window attention (mswin.py:52-96 BaseWindowAttention) of a 4 x 16-pixel block of the map
// -- four 4 x 4 windows (SRC 2: MFMA form, DH = 32 / 64) or sixteen 2 x 2 windows (SRC 3: one thread per (token, head), DH = 16) --
// computed from the bf16 [q | k | v] rows exactly as window_attn_mfma_kernel / window_attn_kernel of v2xvit.hip do, rounded to bf16
// into LDS, and the Linear is the branch's output projection: the attention output never exists in HBM (144 MB written + read per
// branch at 8 agents) and a launch disappears.  Panel row r = block pixel (r >> 4, r & 15).
constexpr int SRC_ROWS = 0, SRC_LN = 1, SRC_WIN4 = 2, SRC_WIN2 = 3, SRC_LNC = 4;
typedef float lin_f32x4 __attribute__((ext_vector_type(4)));

template <int SRC, bool FFN, int DH = 0>
__global__ __launch_bounds__(256, 4) void linear_bf16_occ_kernel(const LinParams p) {
    constexpr int BMO = 64;
    constexpr bool LN = SRC == SRC_LN, LNC = SRC == SRC_LNC, WIN = SRC == SRC_WIN4 || SRC == SRC_WIN2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lin_smem[];
    __bf16* As = reinterpret_cast<__bf16*>(lin_smem);        // [64][264]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    long long m0 = (long long)blockIdx.x * BMO;
    int rows = (int)((p.M - m0) < BMO ? (p.M - m0) : BMO);
    if constexpr (WIN) {     // first pixel of the 4 x 16 block; the output "panel" spans 3 map rows + 16 pixels
        const int bx = p.W >> 4, by = p.H >> 2;
        const int tx = blockIdx.x % bx, ty = (blockIdx.x / bx) % by, ag = blockIdx.x / (bx * by);
        m0 = ((long long)ag * p.H + 4 * ty) * p.W + 16 * tx;
        rows = 3 * p.W + 16;
    }
    const int nchunks = FFN ? 1 : (p.CoutP >> 8);

    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.w), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(FFN ? p.w2 : p.w), 0, p.w_bytes, 0x00020000);
    const unsigned voffB = (unsigned)((lh * p.CoutP + wave * 64 + li) * 16);
    const unsigned step_stride = (unsigned)(2 * p.CoutP * 16);
    u32x4 bf[4][2];
    auto loadB = [&](u32x4 (&dst)[2], const __amdgpu_buffer_rsrc_t r, int ch, int s) {
        const unsigned so = (unsigned)s * step_stride + (unsigned)ch * (256 * 16);
        dst[0] = __builtin_amdgcn_raw_buffer_load_b128(r, voffB, so, 0);
        dst[1] = __builtin_amdgcn_raw_buffer_load_b128(r, voffB + 32 * 16, so, 0);
    };
    loadB(bf[0], rw, 0, 0);
    loadB(bf[1], rw, 0, 1);
    loadB(bf[2], rw, 0, 2);
    loadB(bf[3], rw, 0, 3);
    if constexpr (LN) {
        // rows beyond M read zeros (buffer range) and their outputs are dropped by the output range below
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(p.x + m0 * LK, 0, (unsigned)rows * (LK * 4), 0x00020000);
        long long addl = p.add_rows - m0;
        const int addr = (int)(addl < 0 ? 0 : (addl > rows ? rows : addl));     // rows [0, addr) of the panel receive delta
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.delta ? p.delta + m0 * LK : nullptr), 0,
                                                                            p.delta ? (unsigned)addr * (LK * 2) : 0u, 0x00020000);
        const float4 g = reinterpret_cast<const float4*>(p.gamma)[lane], bt = reinterpret_cast<const float4*>(p.beta)[lane];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 xv[8];
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            u32x2 dv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = wave * 16 + half * 8 + j;
                xv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (unsigned)(r * (LK * 4) + lane * 16), 0, LIN_NT));
                dv[j] = __builtin_amdgcn_raw_buffer_load_b64(rd, (unsigned)(r * (LK * 2) + lane * 8), 0, LIN_NT);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = wave * 16 + half * 8 + j;
                float4 v = make_float4(xv[j][0], xv[j][1], xv[j][2], xv[j][3]);
                if (r < addr) {      // wave-uniform
                    v = add_bf16x4(v, make_uint2(dv[j][0], dv[j][1]));
                    const f32x4 vv = {v.x, v.y, v.z, v.w};
                    if (p.write_back)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vv), rx, (unsigned)(r * (LK * 4) + lane * 16), 0, LIN_NT);
                }
                const f32x4 y = layernorm_row_256(v, g, bt, p.eps);
                *reinterpret_cast<bf16x4*>(As + r * LROW + 4 * lane) = __builtin_convertvector(y, bf16x4);
            }
        }
    } else if constexpr (LNC) {
        // ---- x = SplitAttn combine (+ pending delta) written back, panel = LayerNorm(x): split_combine_kernel's arithmetic
        // (split_attn_rows.hpp) and layernorm_row_256 on the rows of ONE agent (hw % 64 == 0: a panel never straddles two agents)
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(p.x + m0 * LK, 0, (unsigned)rows * (LK * 4), 0x00020000);
        long long addl = p.add_rows - m0;
        const int addr = (int)(addl < 0 ? 0 : (addl > rows ? rows : addl));
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.delta ? p.delta + m0 * LK : nullptr), 0,
                                                                            p.delta ? (unsigned)addr * (LK * 2) : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.br[0] + m0 * LK), 0, (unsigned)rows * (LK * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rb1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.br[1] + m0 * LK), 0, (unsigned)rows * (LK * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rb2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.br[2] + m0 * LK), 0, (unsigned)rows * (LK * 2), 0x00020000);
        const float4 g = reinterpret_cast<const float4*>(p.gamma)[lane], bt = reinterpret_cast<const float4*>(p.beta)[lane];
        float w[3][4];
        av2x::split_attn_weights(p.logits + (m0 / p.hw) * (3 * LK), 4 * lane, LK, w);
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        auto cvt = [](const u32x2 u) {
            return make_float4(__builtin_bit_cast(float, u[0] << 16), __builtin_bit_cast(float, u[0] & 0xffff0000u),
                               __builtin_bit_cast(float, u[1] << 16), __builtin_bit_cast(float, u[1] & 0xffff0000u));
        };
#pragma unroll
        for (int quarter = 0; quarter < 4; ++quarter) {
            f32x4 xv[4];
            u32x2 dv[4], b0[4], b1[4], b2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = wave * 16 + quarter * 4 + j;
                const unsigned o16 = (unsigned)(r * (LK * 2) + lane * 8);
                xv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (unsigned)(r * (LK * 4) + lane * 16), 0, LIN_NT));
                dv[j] = __builtin_amdgcn_raw_buffer_load_b64(rd, o16, 0, LIN_NT);
                b0[j] = __builtin_amdgcn_raw_buffer_load_b64(rb0, o16, 0, LIN_NT);
                b1[j] = __builtin_amdgcn_raw_buffer_load_b64(rb1, o16, 0, LIN_NT);
                b2[j] = __builtin_amdgcn_raw_buffer_load_b64(rb2, o16, 0, LIN_NT);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = wave * 16 + quarter * 4 + j;
                float4 v = make_float4(xv[j][0], xv[j][1], xv[j][2], xv[j][3]);
                if (r < addr) {      // wave-uniform: the pending residual first (r = res + delta of split_combine_kernel)
                    const float4 dl = cvt(dv[j]);
                    v.x += dl.x; v.y += dl.y; v.z += dl.z; v.w += dl.w;
                }
                const float4 y4 = av2x::split_attn_combine4(cvt(b0[j]), cvt(b1[j]), cvt(b2[j]), w, v);
                const f32x4 yv = {y4.x, y4.y, y4.z, y4.w};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, yv), rx, (unsigned)(r * (LK * 4) + lane * 16), 0, LIN_NT);
                const f32x4 y = layernorm_row_256(y4, g, bt, p.eps);
                *reinterpret_cast<bf16x4*>(As + r * LROW + 4 * lane) = __builtin_convertvector(y, bf16x4);
            }
        }
    } else if constexpr (SRC == SRC_WIN4) {
        // ---- window_attn_mfma_kernel of v2xvit.hip, one (window, head) task per wave at a time (same instruction sequence per task)
        constexpr int WS = 4, NB = DH / 16;
        const int t = lane & 15, h = lane >> 4;
        const int inner = p.heads * DH;
        const int iy = t >> 2, ix = t & 3;
        float wbias[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) wbias[r] = p.pos[(h - iy + WS - 1) * (2 * WS - 1) + (r - ix + WS - 1)];
        __bf16* vt = reinterpret_cast<__bf16*>(lin_smem + (size_t)BMO * LROW * 2) + wave * (16 * DH);
        const int ntask = 4 * p.heads;
        for (int tk = wave; tk < ntask; tk += 4) {
            const int head = tk % p.heads, wdw = tk / p.heads;
            const size_t pix0 = (size_t)m0 + 4 * wdw;
            const __bf16* rowt = p.qkv + (pix0 + (size_t)(t >> 2) * p.W + (t & 3)) * p.qkv_ctot + p.qkv_coff + head * DH;
            lin_f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < NB; ++g) {
                const float4 qq = ld_bf16x4(rowt + 4 * (h + 4 * g));
                const float4 kk = ld_bf16x4(rowt + inner + 4 * (h + 4 * g));
                st = __builtin_amdgcn_mfma_f32_16x16x4f32(kk.x, qq.x, st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_16x16x4f32(kk.y, qq.y, st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_16x16x4f32(kk.z, qq.z, st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_16x16x4f32(kk.w, qq.w, st, 0, 0, 0);
            }
            float sc[4], mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) { sc[r] = st[r] * p.scale + wbias[r]; mx = fmaxf(mx, sc[r]); }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float l = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { sc[r] = expf(sc[r] - mx); l += sc[r]; }
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
            const float inv = 1.0f / l;
            constexpr int PPK = DH / 8;
#pragma unroll
            for (int e = lane; e < 16 * PPK; e += 64) {
                const int key = e / PPK, part = e % PPK;
                *reinterpret_cast<uint4*>(&vt[key * DH + part * 8]) = *reinterpret_cast<const uint4*>(
                    p.qkv + (pix0 + (size_t)(key >> 2) * p.W + (key & 3)) * p.qkv_ctot + p.qkv_coff + 2 * inner + head * DH + part * 8);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                lin_f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    o = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[r] * inv, (float)vt[(4 * h + r) * DH + nb * 16 + t], o, 0, 0, 0);
                // query (iy = h, ix = r) of window wdw = panel row 16 h + 4 wdw + r
#pragma unroll
                for (int r = 0; r < 4; ++r) As[(16 * h + 4 * wdw + r) * LROW + head * DH + nb * 16 + t] = (__bf16)o[r];
            }
        }
    } else if constexpr (SRC == SRC_WIN2) {
        // ---- window_attn_kernel<16, 2> of v2xvit.hip: one (token, head) pair per thread and pass, 64 tokens x 16 heads = 4 passes
        constexpr int WS = 2, DHD = 16;
        const int inner = p.heads * DHD;
        for (int e = tid; e < BMO * p.heads; e += 256) {
            const int head = e % p.heads, tok = e / p.heads;
            const int py = tok >> 4, px = tok & 15;
            const int wy0 = py & ~1, wx0 = px & ~1, iy = py - wy0, ixx = px - wx0;
            const __bf16* row = p.qkv + ((size_t)m0 + (size_t)py * p.W + px) * p.qkv_ctot + p.qkv_coff + head * DHD;
            float q[DHD], o[DHD];
#pragma unroll
            for (int d = 0; d < DHD; d += 4) {
                const float4 v = ld_bf16x4(row + d);
                q[d] = v.x; q[d + 1] = v.y; q[d + 2] = v.z; q[d + 3] = v.w;
            }
            float sj[WS * WS];
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < WS * WS; ++j) {
                const int jy = j / WS, jx = j % WS;
                const __bf16* kr = p.qkv + ((size_t)m0 + (size_t)(wy0 + jy) * p.W + (wx0 + jx)) * p.qkv_ctot + p.qkv_coff + inner + head * DHD;
                float acc = 0.f;
#pragma unroll
                for (int d = 0; d < DHD; d += 4) {
                    const float4 k = ld_bf16x4(kr + d);
                    acc = fmaf(q[d], k.x, acc); acc = fmaf(q[d + 1], k.y, acc); acc = fmaf(q[d + 2], k.z, acc); acc = fmaf(q[d + 3], k.w, acc);
                }
                acc = acc * p.scale + p.pos[(jy - iy + WS - 1) * (2 * WS - 1) + (jx - ixx + WS - 1)];
                sj[j] = acc;
                mx = fmaxf(mx, acc);
            }
            float l = 0.f;
#pragma unroll
            for (int j = 0; j < WS * WS; ++j) { sj[j] = expf(sj[j] - mx); l += sj[j]; }
            const float inv = 1.0f / l;
#pragma unroll
            for (int d = 0; d < DHD; ++d) o[d] = 0.f;
#pragma unroll
            for (int j = 0; j < WS * WS; ++j) {
                const int jy = j / WS, jx = j % WS;
                const __bf16* vr = p.qkv + ((size_t)m0 + (size_t)(wy0 + jy) * p.W + (wx0 + jx)) * p.qkv_ctot + p.qkv_coff + 2 * inner + head * DHD;
                const float pj = sj[j] * inv;
#pragma unroll
                for (int d = 0; d < DHD; d += 4) {
                    const float4 v = ld_bf16x4(vr + d);
                    o[d] = fmaf(pj, v.x, o[d]); o[d + 1] = fmaf(pj, v.y, o[d + 1]); o[d + 2] = fmaf(pj, v.z, o[d + 2]); o[d + 3] = fmaf(pj, v.w, o[d + 3]);
                }
            }
#pragma unroll
            for (int d = 0; d < DHD; d += 4) {
                const f32x4 f = {o[d], o[d + 1], o[d + 2], o[d + 3]};
                *reinterpret_cast<bf16x4*>(As + tok * LROW + head * DHD + d) = __builtin_convertvector(f, bf16x4);
            }
        }
    } else {
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.a + m0 * LK), 0,
                                                                            (unsigned)rows * (LK * 2), 0x00020000);
        u32x4 va[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) va[j] = __builtin_amdgcn_raw_buffer_load_b128(ra, (unsigned)((tid + 256 * j) * 16), 0, LIN_NT);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = tid + 256 * j;
            *reinterpret_cast<u32x4*>(As + (e >> 5) * LROW + (e & 31) * 8) = va[j];
        }
    }
    __syncthreads();
    const size_t obytes = (size_t)p.out_ctot * 2;
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(p.out) + (size_t)m0 * obytes, 0,
                                                                          (unsigned)((size_t)rows * obytes), 0x00020000);
    const __bf16* Ab = As + li * LROW + lh * 8;
    const bool odd = li & 1, hi = li & 2;
    f32x16 acc[2][2];
    // K loop of one 256-column chunk `ch` of weights `rcur`, prefetching the first four steps of chunk `chn` of `rnext`
    auto kloop = [&](const __amdgpu_buffer_rsrc_t rcur, int ch, const __amdgpu_buffer_rsrc_t rnext, int chn) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const bf16x8 fa0 = *reinterpret_cast<const bf16x8*>(Ab + s * 16);
            const bf16x8 fa1 = *reinterpret_cast<const bf16x8*>(Ab + 32 * LROW + s * 16);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bf16x8 fb = __builtin_bit_cast(bf16x8, bf[s & 3][c]);
                acc[0][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb, acc[0][c], 0, 0, 0);
                acc[1][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb, acc[1][c], 0, 0, 0);
            }
            if (s + 4 < 16) loadB(bf[s & 3], rcur, ch, s + 4);
            else loadB(bf[s & 3], rnext, chn, s + 4 - 16);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // epilogue: bias, activation, one rounding, 4 x 4 quad transposes, 16-byte stores (8 rows x 128 B per instruction) to the
    // output rows -- or (TO_LDS) into the panel, as the A operand of the second Linear
    auto epilogue = [&](int ch, const float* bias, int act, auto to_lds) {
        constexpr bool TO_LDS = decltype(to_lds)::value;
        const int n = ch * 256 + wave * 64 + 2 * li;
        float b0 = 0.f, b1 = 0.f;
        if (bias && n < p.Cout) { b0 = bias[n]; b1 = bias[n + 1]; }
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        int ostride = p.out_ctot * 2;
        asm volatile("" : "+s"(ostride));
        const int nq = ch * 256 + wave * 64 + 2 * (li & ~3);
        const unsigned off0 = nq < p.Cout ? (unsigned)((4 * lh + (li & 3)) * ostride + (p.out_coff + nq) * 2) : 0x80000000u;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned R[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * g + j;
                    f32x2 v = {acc[a][0][r] + b0, acc[a][1][r] + b1};
                    if (act == 1) {
                        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f);
                    } else if (act == 2) {
                        v[0] = 0.5f * v[0] * (1.0f + erf_as(v[0] * 0.70710678118654752f));
                        v[1] = 0.5f * v[1] * (1.0f + erf_as(v[1] * 0.70710678118654752f));
                    }
                    R[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const unsigned send = odd ? R[2 * m] : R[2 * m + 1];
                    const unsigned got = (unsigned)__builtin_amdgcn_mov_dpp((int)send, 0xB1, 0xf, 0xf, true);
                    R[2 * m] = odd ? got : R[2 * m];
                    R[2 * m + 1] = odd ? R[2 * m + 1] : got;
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const unsigned send = hi ? R[m] : R[m + 2];
                    const unsigned got = (unsigned)__builtin_amdgcn_mov_dpp((int)send, 0x4E, 0xf, 0xf, true);
                    R[m] = hi ? got : R[m];
                    R[m + 2] = hi ? R[m + 2] : got;
                }
                const u32x4 v4 = {R[0], R[1], R[2], R[3]};
                if constexpr (TO_LDS)
                    *reinterpret_cast<u32x4*>(As + (a * 32 + 8 * g + 4 * lh + (li & 3)) * LROW + wave * 64 + 2 * (li & ~3)) = v4;
                else if constexpr (WIN)    // panel row 32 a + 8 g + .. = block pixel (2 a + (g >> 1), 8 (g & 1) + ..)
                    __builtin_amdgcn_raw_buffer_store_b128(v4, rout, off0 + (unsigned)(((2 * a + (g >> 1)) * p.W + 8 * (g & 1)) * ostride), 0, LIN_NT);
                else
                    __builtin_amdgcn_raw_buffer_store_b128(v4, rout, off0 + (unsigned)((a * 32 + 8 * g) * ostride), 0, LIN_NT);
            }
    };
    if constexpr (FFN) {
        kloop(rw, 0, rw2, 0);
        __syncthreads();                       // every wave is past its reads of the input panel
        epilogue(0, p.bias, p.act, std::true_type{});
        __syncthreads();
        kloop(rw2, 0, rw2, 0);
        epilogue(0, p.bias2, p.act2, std::false_type{});
    } else {
        for (int ch = 0; ch < nchunks; ++ch) {
            const int chn = ch + 1 < nchunks ? ch + 1 : ch;
            kloop(rw, ch, rw, chn);
            epilogue(ch, p.bias, p.act, std::false_type{});
        }
    }
}

// ---- LayerNorm -> three window-attention QKVs -> window attention -> to_out of a 4 x 16-pixel block in ONE workgroup (the
// PreNormResidual(PyramidWindowAttention) body of V2XFusionBlock, v2xvit_basic.py:137-159, mswin.py:99-145): the 2304-wide QKV
// tensor (1.3 GB per layer at 8 agents, written and read back by the separate launches) never exists in HBM.
//   LDS: four [64][264] bf16 panels = 132 KB -> one workgroup per CU: P0 = LayerNorm(x (+ delta)), P1 / P2 / P3 = q / k / v of the
//   current branch (chunk 3 b + c of the packed 256 -> 2304 weights), the attention output overwrites q in place (a (window, head)
//   task reads exactly the q region it writes), P1 is then the A panel of to_out (chunk b of the packed 256 -> 768 weights).
//   Every GEMM is the K loop of linear_bf16_occ_kernel (same MFMA sequence per output), the q / k / v panels are rounded to bf16 as
//   the stored tensor was, the attention is the instruction sequence of v2xvit.hip's kernels: the outputs are the bits of
//   av2x_ln_linear_bf16 + 3 x av2x_window_attention_linear_bf16.  Eight K-steps of weight fragments in flight (one workgroup per
//   CU: nobody else hides the L2 latency).
#ifndef AV2X_QW_ABLATE      // timing experiments of tools/micro/qw_ablate.sh (wrong results): 1 no weight stream after the first fragments,
#define AV2X_QW_ABLATE 0    // 2 no attention phases, 4 no output stores, (slab kernel) 8 no q / k / v panel stores, 16 no QKV K loops; the library never defines it
#endif
#ifndef AV2X_QW_DEPTH
#define AV2X_QW_DEPTH 4
#endif
struct QwParams {
    const float* x;
    const __bf16* delta;      // pending residual of every row (or nullptr): feeds the LayerNorm, x is not rewritten
    const float* gamma;
    const float* beta;
    float eps;
    const __bf16* w;          // 256 -> 2304: [q | k | v] of the three branches
    const float* bias;        // or nullptr
    const __bf16* w2;         // 256 -> 768: the three to_out
    const float* bias2;
    const float* pos[3];
    __bf16* out[3];
    int heads[3], dh[3];
    int H, W;
};

// 8 waves: waves 0-3 own the GEMMs (64 output columns each), all 8 share the LayerNorm rows and the attention tasks -- the phases that are
// chains of dependent LDS / VALU work run two waves per SIMD
__global__ __launch_bounds__(512, 1) void ln_qkv_window_out_bf16_kernel(const QwParams p) {
    constexpr int PSZ = 64 * LROW, DEPTH = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char lin_smem[];
    __bf16* P0 = reinterpret_cast<__bf16*>(lin_smem);
    __bf16* P1 = P0 + PSZ;
    __bf16* P2 = P1 + PSZ;
    __bf16* P3 = P2 + PSZ;
    // one workgroup per CU: a global load in front of an epilogue or an attention phase is a fully exposed round trip, so the biases
    // (2304 + 768 floats) and the three relative-position tables (<= 49 floats each) are copied into LDS once, under the panel load
    float* biasl = reinterpret_cast<float*>(P3 + PSZ);             // [3072]
    float* posl = biasl + 3072;                                     // [3][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int bx = p.W >> 4, by = p.H >> 2;
    const int tx = blockIdx.x % bx, ty = (blockIdx.x / bx) % by, ag = blockIdx.x / (bx * by);
    const long long m0 = ((long long)ag * p.H + 4 * ty) * p.W + 16 * tx;
    const int span = 3 * p.W + 16;                                  // tokens from the block's first pixel to its last
    const bool gemm = wave < 4;
    if (tid < 256) {
        float4 bv[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int i4 = tid + 256 * j;                           // float4 index: [0, 576) qkv bias, [576, 768) to_out bias
            bv[j] = i4 < 576 ? (p.bias ? reinterpret_cast<const float4*>(p.bias)[i4] : make_float4(0.f, 0.f, 0.f, 0.f))
                             : reinterpret_cast<const float4*>(p.bias2)[i4 - 576];
        }
        float pv = 0.f;
        if (tid < 192) {
            const int b = tid >> 6, i = tid & 63;
            const int side = p.dh[b] == 16 ? 3 : 7;
            if (i < side * side) pv = p.pos[b][i];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) reinterpret_cast<float4*>(biasl)[tid + 256 * j] = bv[j];
        if (tid < 192) posl[tid] = pv;
    }

    constexpr int CP1 = 2304, CP2 = 768;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.w), 0, (unsigned)((LK / 8) * CP1 * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.w2), 0, (unsigned)((LK / 8) * CP2 * 16), 0x00020000);
    const unsigned voff1 = (unsigned)((lh * CP1 + wave * 64 + li) * 16), voff2 = (unsigned)((lh * CP2 + wave * 64 + li) * 16);
    u32x4 bf[DEPTH][2];
    bool qw_first = true;
    auto loadB = [&](u32x4 (&dst)[2], bool second, int ch, int s) {
        if constexpr ((AV2X_QW_ABLATE & 1) != 0) { if (!qw_first) return; }
        if (second) {
            const unsigned so = (unsigned)s * (2 * CP2 * 16) + (unsigned)ch * (256 * 16);
            dst[0] = __builtin_amdgcn_raw_buffer_load_b128(rw2, voff2, so, 0);
            dst[1] = __builtin_amdgcn_raw_buffer_load_b128(rw2, voff2 + 32 * 16, so, 0);
        } else {
            const unsigned so = (unsigned)s * (2 * CP1 * 16) + (unsigned)ch * (256 * 16);
            dst[0] = __builtin_amdgcn_raw_buffer_load_b128(rw, voff1, so, 0);
            dst[1] = __builtin_amdgcn_raw_buffer_load_b128(rw, voff1 + 32 * 16, so, 0);
        }
    };
    if (gemm) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) loadB(bf[s], false, 0, s);
    }
    qw_first = false;

    // ---- P0 = LayerNorm(x (+ delta)): wave w = block row w, 16 pixels, four channels per lane (layernorm_row_256)
    {
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x) + m0 * LK, 0, (unsigned)span * (LK * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.delta ? p.delta + m0 * LK : nullptr), 0,
                                                                            p.delta ? (unsigned)span * (LK * 2) : 0u, 0x00020000);
        const float4 g = reinterpret_cast<const float4*>(p.gamma)[lane], bt = reinterpret_cast<const float4*>(p.beta)[lane];
        const bool add = p.delta != nullptr;
        {   // wave w: rows 8 w .. 8 w + 7 = block row w >> 1, pixels 8 (w & 1) ..
            f32x4 xv[8];
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            u32x2 dv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int tok = (wave >> 1) * p.W + 8 * (wave & 1) + j;
                xv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (unsigned)(tok * (LK * 4) + lane * 16), 0, LIN_NT));
                dv[j] = __builtin_amdgcn_raw_buffer_load_b64(rd, (unsigned)(tok * (LK * 2) + lane * 8), 0, LIN_NT);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float4 v = make_float4(xv[j][0], xv[j][1], xv[j][2], xv[j][3]);
                if (add) v = add_bf16x4(v, make_uint2(dv[j][0], dv[j][1]));
                const f32x4 y = layernorm_row_256(v, g, bt, p.eps);
                *reinterpret_cast<bf16x4*>(P0 + (wave * 8 + j) * LROW + 4 * lane) = __builtin_convertvector(y, bf16x4);
            }
        }
    }
    __syncthreads();

    const bool odd = li & 1, hi = li & 2;
    f32x16 acc[2][2];
    // q / k / v (and the attention output in q's place) are stored with the columns of row r rotated by 16 (r >> 4) elements: the attention
    // reads rows 16 apart at the same column (four 4 x 4-window rows), which would otherwise fall into the same banks
    auto swz = [](int row, int col) { return (col + 16 * (row >> 4)) & 255; };
    auto kloop = [&](const __bf16* A, bool second, int ch, bool nsecond, int chn, bool rotated = false) {
        const __bf16* Ab = A + li * LROW;
        const int rot0 = rotated ? 16 * (li >> 4) : 0, rot1 = rotated ? 16 * (2 + (li >> 4)) : 0;
        auto acol = [&](int s, int rot) { return (s * 16 + lh * 8 + rot) & 255; };
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
        // one wave per SIMD: the A fragments of step s + 1 are requested before the MFMAs of step s (nobody else hides the LDS latency)
        bf16x8 fa0 = *reinterpret_cast<const bf16x8*>(Ab + acol(0, rot0));
        bf16x8 fa1 = *reinterpret_cast<const bf16x8*>(Ab + 32 * LROW + acol(0, rot1));
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            bf16x8 fn0 = fa0, fn1 = fa1;
            if (s + 1 < 16) {
                fn0 = *reinterpret_cast<const bf16x8*>(Ab + acol(s + 1, rot0));
                fn1 = *reinterpret_cast<const bf16x8*>(Ab + 32 * LROW + acol(s + 1, rot1));
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bf16x8 fb = __builtin_bit_cast(bf16x8, bf[s & (DEPTH - 1)][c]);
                acc[0][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb, acc[0][c], 0, 0, 0);
                acc[1][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb, acc[1][c], 0, 0, 0);
            }
            if (s + DEPTH < 16) loadB(bf[s & (DEPTH - 1)], second, ch, s + DEPTH);
            else loadB(bf[s & (DEPTH - 1)], nsecond, chn, s + DEPTH - 16);
            fa0 = fn0; fa1 = fn1;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // bias, one rounding, quad transposes (as linear_bf16_occ_kernel's epilogue); dst: an LDS panel or the output rows of the block
    auto epilogue = [&](int ch, const float* bias, __bf16* panel, const __amdgpu_buffer_rsrc_t rout, auto to_lds) {
        constexpr bool TO_LDS = decltype(to_lds)::value;
        const int n = ch * 256 + wave * 64 + 2 * li;
        const float b0 = bias[n], b1 = bias[n + 1];                // LDS copy
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        const unsigned off0 = (unsigned)((4 * lh + (li & 3)) * 512 + (wave * 64 + 2 * (li & ~3)) * 2);
        if constexpr (TO_LDS) {
            // into an LDS panel the raw MFMA layout is as good as any: the lane's column pair of every row as one 4-byte store (conflict-free:
            // 32 consecutive banks per half-wave), no transposes
            __bf16* dst = panel + (4 * lh) * LROW;
            const int col = wave * 64 + 2 * li;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const f32x2 v = {acc[a][0][r] + b0, acc[a][1][r] + b1};
                    // row 32 a + 8 (r >> 2) + 4 lh + (r & 3): row >> 4 = 2 a + (r >> 3)
                    *reinterpret_cast<unsigned*>(dst + (a * 32 + (r & 3) + 8 * (r >> 2)) * LROW + ((col + 16 * (2 * a + (r >> 3))) & 255)) =
                        __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
                }
            return;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned R[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * g + j;
                    const f32x2 v = {acc[a][0][r] + b0, acc[a][1][r] + b1};
                    R[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const unsigned send = odd ? R[2 * m] : R[2 * m + 1];
                    const unsigned got = (unsigned)__builtin_amdgcn_mov_dpp((int)send, 0xB1, 0xf, 0xf, true);
                    R[2 * m] = odd ? got : R[2 * m];
                    R[2 * m + 1] = odd ? R[2 * m + 1] : got;
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const unsigned send = hi ? R[m] : R[m + 2];
                    const unsigned got = (unsigned)__builtin_amdgcn_mov_dpp((int)send, 0x4E, 0xf, 0xf, true);
                    R[m] = hi ? got : R[m];
                    R[m + 2] = hi ? R[m + 2] : got;
                }
                const u32x4 v4 = {R[0], R[1], R[2], R[3]};
                if constexpr (TO_LDS)
                    *reinterpret_cast<u32x4*>(panel + (a * 32 + 8 * g + 4 * lh + (li & 3)) * LROW + wave * 64 + 2 * (li & ~3)) = v4;
                else
                    __builtin_amdgcn_raw_buffer_store_b128(v4, rout, off0 + (unsigned)(((2 * a + (g >> 1)) * p.W + 8 * (g & 1)) * 512), 0, LIN_NT);
            }
    };
    // window_attn_mfma_kernel on the LDS panels: (window, head) tasks of the block's four 4 x 4 windows
    auto win4 = [&](auto dhc, int heads, const float* pos) {
        constexpr int DH = decltype(dhc)::value, WS = 4, NB = DH / 16;
        const int t = lane & 15, h = lane >> 4;
        const int iy = t >> 2, ix = t & 3;
        const float scale = 1.0f / sqrtf((float)DH);
        float wbias[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) wbias[r] = pos[(h - iy + WS - 1) * (2 * WS - 1) + (r - ix + WS - 1)];
        const int ntask = 4 * heads;             // a multiple of 16: two independent tasks per wave and pass (one wave per SIMD: the second
        for (int tk0 = wave; tk0 < ntask; tk0 += 16) {   // task's LDS reads and MFMAs fill the first one's latencies); per task the
            int head[2], wdw[2], rowt[2];               // instruction sequence of window_attn_mfma_kernel
            lin_f32x4 st[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int tk = tk0 + 8 * u;
                head[u] = tk % heads; wdw[u] = tk / heads;
                rowt[u] = ((t >> 2) * 16 + 4 * wdw[u] + (t & 3)) * LROW;
                st[u] = lin_f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int g = 0; g < NB; ++g) {
                float4 qq[2], kk[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int cq = swz((t >> 2) * 16, head[u] * DH + 4 * (h + 4 * g));
                    qq[u] = ld_bf16x4(P1 + rowt[u] + cq);
                    kk[u] = ld_bf16x4(P2 + rowt[u] + cq);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) st[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(kk[u].x, qq[u].x, st[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; ++u) st[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(kk[u].y, qq[u].y, st[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; ++u) st[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(kk[u].z, qq[u].z, st[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; ++u) st[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(kk[u].w, qq[u].w, st[u], 0, 0, 0);
            }
            float sc[2][4], inv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float mx = -INFINITY;
#pragma unroll
                for (int r = 0; r < 4; ++r) { sc[u][r] = st[u][r] * scale + wbias[r]; mx = fmaxf(mx, sc[u][r]); }
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                float l = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) { sc[u][r] = expf(sc[u][r] - mx); l += sc[u][r]; }
                l += __shfl_xor(l, 16);
                l += __shfl_xor(l, 32);
                inv[u] = 1.0f / l;
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                float vv[2][4];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) vv[u][r] = (float)P3[(16 * h + 4 * wdw[u] + r) * LROW + swz(16 * h, head[u] * DH + nb * 16 + t)];
                lin_f32x4 o[2] = {lin_f32x4{0.f, 0.f, 0.f, 0.f}, lin_f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int u = 0; u < 2; ++u) o[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[u][r] * inv[u], vv[u][r], o[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) P1[(16 * h + 4 * wdw[u] + r) * LROW + swz(16 * h, head[u] * DH + nb * 16 + t)] = (__bf16)o[u][r];
            }
        }
    };
    // window_attn_kernel<16, 2> on the LDS panels: one (token, head) pair per thread and pass
    auto win2 = [&](int heads, const float* pos) {
        constexpr int WS = 2, DHD = 16;
        const float scale = 1.0f / sqrtf((float)DHD);
        for (int e = tid; e < 64 * heads; e += 512) {
            const int head = e % heads, tok = e / heads;
            const int py = tok >> 4, px = tok & 15;
            const int wy0 = py & ~1, wx0 = px & ~1, iy = py - wy0, ixx = px - wx0;
            float q[DHD], o[DHD];
#pragma unroll
            for (int d = 0; d < DHD; d += 4) {
                const float4 v = ld_bf16x4(P1 + tok * LROW + swz(tok, head * DHD + d));
                q[d] = v.x; q[d + 1] = v.y; q[d + 2] = v.z; q[d + 3] = v.w;
            }
            float sj[WS * WS];
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < WS * WS; ++j) {
                const int jy = j / WS, jx = j % WS;
                const int krow = (wy0 + jy) * 16 + wx0 + jx;
                const __bf16* kr = P2 + krow * LROW + swz(krow, head * DHD);      // 16 columns of a head: no wrap inside (16-aligned)
                float a_ = 0.f;
#pragma unroll
                for (int d = 0; d < DHD; d += 4) {
                    const float4 k = ld_bf16x4(kr + d);
                    a_ = fmaf(q[d], k.x, a_); a_ = fmaf(q[d + 1], k.y, a_); a_ = fmaf(q[d + 2], k.z, a_); a_ = fmaf(q[d + 3], k.w, a_);
                }
                a_ = a_ * scale + pos[(jy - iy + WS - 1) * (2 * WS - 1) + (jx - ixx + WS - 1)];
                sj[j] = a_;
                mx = fmaxf(mx, a_);
            }
            float l = 0.f;
#pragma unroll
            for (int j = 0; j < WS * WS; ++j) { sj[j] = expf(sj[j] - mx); l += sj[j]; }
            const float inv = 1.0f / l;
#pragma unroll
            for (int d = 0; d < DHD; ++d) o[d] = 0.f;
#pragma unroll
            for (int j = 0; j < WS * WS; ++j) {
                const int jy = j / WS, jx = j % WS;
                const int vrow = (wy0 + jy) * 16 + wx0 + jx;
                const __bf16* vr = P3 + vrow * LROW + swz(vrow, head * DHD);
                const float pj = sj[j] * inv;
#pragma unroll
                for (int d = 0; d < DHD; d += 4) {
                    const float4 v = ld_bf16x4(vr + d);
                    o[d] = fmaf(pj, v.x, o[d]); o[d + 1] = fmaf(pj, v.y, o[d + 1]); o[d + 2] = fmaf(pj, v.z, o[d + 2]); o[d + 3] = fmaf(pj, v.w, o[d + 3]);
                }
            }
#pragma unroll
            for (int d = 0; d < DHD; d += 4) {
                const f32x4 f = {o[d], o[d + 1], o[d + 2], o[d + 3]};
                *reinterpret_cast<bf16x4*>(P1 + tok * LROW + swz(tok, head * DHD + d)) = __builtin_convertvector(f, bf16x4);
            }
        }
    };

#pragma unroll 1
    for (int b = 0; b < 3; ++b) {
#pragma unroll 1
        for (int c = 0; c < 3; ++c) {
            const int ch = 3 * b + c;
            if (gemm) kloop(P0, false, ch, c == 2, c == 2 ? b : ch + 1);
            if (c == 0) __syncthreads();          // every wave is past the previous branch's to_out K loop (reads of P1)
            if (gemm) epilogue(ch, biasl, P1 + c * PSZ, rw, std::true_type{});
        }
        __syncthreads();                          // q, k, v of the branch are complete
        if constexpr ((AV2X_QW_ABLATE & 2) == 0) {
            if (p.dh[b] == 16) win2(p.heads[b], posl + 64 * b);
            else if (p.dh[b] == 32) win4(std::integral_constant<int, 32>{}, p.heads[b], posl + 64 * b);
            else win4(std::integral_constant<int, 64>{}, p.heads[b], posl + 64 * b);
        }
        __syncthreads();                          // the attention output (in P1) is complete
        if (gemm) {
            kloop(P1, true, b, false, b < 2 ? 3 * (b + 1) : 0, true);
            const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(p.out[b] + m0 * LK, 0, (unsigned)span * (LK * 2), 0x00020000);
            if constexpr ((AV2X_QW_ABLATE & 4) == 0) epilogue(b, biasl + 2304, nullptr, rout, std::false_type{});
            else if (acc[0][0][0] + acc[1][1][5] == 123.f) epilogue(b, biasl + 2304, nullptr, rout, std::false_type{});
        }
    }
}

// ---- The same fused layer with TWO workgroups per CU (round 5).  ln_qkv_window_out_bf16_kernel above holds four 33-KB panels (132 KB of
// LDS): one workgroup per CU, and its phases -- panel load + LayerNorm, twelve K loops on four of the eight waves, the attention tasks,
// the output stores -- run one after the other with nothing to overlap them (timing-only builds, tools/micro/qw_ablate.sh: without the
// weight stream 852 -> 780 us per 8-agent launch, without the attention phases 665, without both and the stores 519 = 2.5 x the MFMA time of
// its K loops; the weight stream the round-4 notes blamed is NOT what bounds it).  Here a workgroup is four waves and works through the
// 2304 QKV columns in HEAD SLABS of 64 columns: q, k, v of one slab (three [64][72] bf16 panels) -> the attention of the slab's heads ->
// its [64][72] output panel is the K = 64 slice 4 j .. 4 j + 3 of to_out's K loop, accumulated in registers across the four slabs of a
// branch.  LDS: the LayerNorm panel 33 KB + four slab panels 36 KB + the position tables = 71.4 KB -> two workgroups per CU, one in its
// attention / epilogue / store phases while the other one's MFMAs run.  Per slab and workgroup: wave 0 = q (64 x 64), wave 1 = k, waves 2 / 3
// = the even / odd columns of v (64 x 32 each: a B fragment of an interleaved-pair group holds one parity), every wave one 64-column group
// of to_out.  Every output sees the MFMA sequence of the kernel above (same fragments, same K order: to_out's K steps 0..15 in order over
// the slabs), the same bf16 roundings of q / k / v / attention output, and the attention instruction sequence per (window, head): the bits
// of av2x_ln_linear_bf16 + 3 x av2x_window_attention_linear_bf16 (tests/test_gpu_bf16_activations.py).
constexpr int SROW = 64 + 8;    // slab panel row stride in bf16 elements (144 B: conflict-free ds_read_b128 over 16 rows)

__global__ __launch_bounds__(256, 2) void ln_qkv_window_out_slab_kernel(const QwParams p) {
    constexpr int PSZ = 64 * LROW, SSZ = 64 * SROW, DEPTH = AV2X_QW_DEPTH;
    extern __shared__ __attribute__((aligned(16))) unsigned char lin_smem[];
    __bf16* P0 = reinterpret_cast<__bf16*>(lin_smem);
    __bf16* Qs = P0 + PSZ;
    __bf16* Ks = Qs + SSZ;
    __bf16* Vs = Ks + SSZ;
    __bf16* Os = Vs + SSZ;
    float* posl = reinterpret_cast<float*>(Os + SSZ);              // [3][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int bx = p.W >> 4, by = p.H >> 2;
    const int tx = blockIdx.x % bx, ty = (blockIdx.x / bx) % by, ag = blockIdx.x / (bx * by);
    const long long m0 = ((long long)ag * p.H + 4 * ty) * p.W + 16 * tx;
    const int span = 3 * p.W + 16;                                  // tokens from the block's first pixel to its last
    if (tid < 192) {
        const int b = tid >> 6, i = tid & 63;
        const int side = p.dh[b] == 16 ? 3 : 7;
        posl[tid] = i < side * side ? p.pos[b][i] : 0.f;
    }

    constexpr int CP1 = 2304, CP2 = 768;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.w), 0, (unsigned)((LK / 8) * CP1 * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.w2), 0, (unsigned)((LK / 8) * CP2 * 16), 0x00020000);
    // role of the wave in a slab's QKV product: which of q / k / v (the chunk of the branch), and for v which column parity (B tile)
    // (the q / k waves issue twice the MFMAs of the v waves: the two workgroups of a CU should not both put them on SIMDs 0 and 1 -- workgroups
    // b and b + 256 are the likely first pair of a CU (round-robin over 8 XCDs x 32 CUs), so bit 8 of the index swaps the wave pairs)
    const int rwv = wave ^ (2 * ((blockIdx.x >> 8) & 1));
    const int role = rwv < 2 ? rwv : 2;                             // 0 q, 1 k, 2 v
    const int cv = rwv < 2 ? 0 : rwv - 2;                           // v: B tile (= parity of the logical columns) of this wave
    const bool pair = rwv < 2;                                      // q / k waves own both tiles of the slab's group
    u32x4 bq[DEPTH][2];                                             // QKV ring: DEPTH K-steps x (up to) two B tiles
    u32x4 bt[4][2];                                                 // to_out: the four K-steps of a slab x two B tiles
    bool qw_first = true;
    auto load_q = [&](u32x4 (&dst)[2], int b, int j, int s) __attribute__((always_inline)) {       // step s of slab j of branch b, this wave's columns
        if constexpr ((AV2X_QW_ABLATE & 1) != 0) { if (!qw_first) return; }
        const unsigned vo = (unsigned)((lh * CP1 + j * 64 + li) * 16);
        const unsigned so = (unsigned)s * (2 * CP1 * 16) + (unsigned)(3 * b + role) * (256 * 16);
        if (pair) {
            dst[0] = __builtin_amdgcn_raw_buffer_load_b128(rw, vo, so, 0);
            dst[1] = __builtin_amdgcn_raw_buffer_load_b128(rw, vo + 32 * 16, so, 0);
        } else {
            dst[0] = __builtin_amdgcn_raw_buffer_load_b128(rw, vo + (unsigned)cv * (32 * 16), so, 0);
        }
    };
    auto load_t = [&](int b, int j) __attribute__((always_inline)) {                               // to_out of branch b: K-steps 4 j .. 4 j + 3, columns 64 wave ..
        if constexpr ((AV2X_QW_ABLATE & 1) != 0) { if (b + j > 0) return; }
        const unsigned vo = (unsigned)((lh * CP2 + wave * 64 + li) * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const unsigned so = (unsigned)(4 * j + t) * (2 * CP2 * 16) + (unsigned)b * (256 * 16);
            bt[t][0] = __builtin_amdgcn_raw_buffer_load_b128(rw2, vo, so, 0);
            bt[t][1] = __builtin_amdgcn_raw_buffer_load_b128(rw2, vo + 32 * 16, so, 0);
        }
    };
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) load_q(bq[s], 0, 0, s);
    qw_first = false;

    // ---- P0 = LayerNorm(x (+ delta)): wave w = block row w (16 pixels, two passes of eight), four channels per lane (layernorm_row_256)
    {
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x) + m0 * LK, 0, (unsigned)span * (LK * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.delta ? p.delta + m0 * LK : nullptr), 0,
                                                                            p.delta ? (unsigned)span * (LK * 2) : 0u, 0x00020000);
        const float4 g = reinterpret_cast<const float4*>(p.gamma)[lane], bta = reinterpret_cast<const float4*>(p.beta)[lane];
        const bool add = p.delta != nullptr;
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        f32x4 xv[16];
        u32x2 dv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int tok = wave * p.W + j;
            xv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (unsigned)(tok * (LK * 4) + lane * 16), 0, LIN_NT));
            dv[j] = __builtin_amdgcn_raw_buffer_load_b64(rd, (unsigned)(tok * (LK * 2) + lane * 8), 0, LIN_NT);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float4 v = make_float4(xv[j][0], xv[j][1], xv[j][2], xv[j][3]);
            if (add) v = add_bf16x4(v, make_uint2(dv[j][0], dv[j][1]));
            const f32x4 y = layernorm_row_256(v, g, bta, p.eps);
            *reinterpret_cast<bf16x4*>(P0 + (wave * 16 + j) * LROW + 4 * lane) = __builtin_convertvector(y, bf16x4);
        }
    }
    __syncthreads();

    const bool odd = li & 1, hi = li & 2;
    f32x16 acc[2][2], acct[2][2];
    // slab panels: the columns of row r are rotated by 16 (r >> 4) elements (mod 64): the attention reads rows 16 apart at the same column
    auto swz = [](int row, int col) __attribute__((always_inline)) { return (col + 16 * (row >> 4)) & 63; };

    // QKV product of a slab: 64 rows x this wave's columns, K = 256 from the LayerNorm panel; the ring holds steps 0 .. DEPTH - 1 on entry
    auto kloop_qkv = [&](int b, int j) __attribute__((always_inline)) {
        const __bf16* Ab = P0 + li * LROW + lh * 8;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
        bf16x8 fa0 = *reinterpret_cast<const bf16x8*>(Ab);
        bf16x8 fa1 = *reinterpret_cast<const bf16x8*>(Ab + 32 * LROW);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            bf16x8 fn0 = fa0, fn1 = fa1;
            if (s + 1 < 16) {
                fn0 = *reinterpret_cast<const bf16x8*>(Ab + (s + 1) * 16);
                fn1 = *reinterpret_cast<const bf16x8*>(Ab + 32 * LROW + (s + 1) * 16);
            }
            {
                const bf16x8 fb = __builtin_bit_cast(bf16x8, bq[s & (DEPTH - 1)][0]);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb, acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb, acc[1][0], 0, 0, 0);
            }
            if (pair) {
                const bf16x8 fb = __builtin_bit_cast(bf16x8, bq[s & (DEPTH - 1)][1]);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb, acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb, acc[1][1], 0, 0, 0);
            }
            if (s + DEPTH < 16) load_q(bq[s & (DEPTH - 1)], b, j, s + DEPTH);
            fa0 = fn0; fa1 = fn1;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // bias, one rounding to bf16, into the wave's slab panel in the raw MFMA layout (rows rotated as swz): q / k waves store the lane's column
    // pair (4 bytes), the v waves their parity's column (2 bytes)
    auto epilogue_qkv = [&](int b, int j) __attribute__((always_inline)) {
        __bf16* panel = Qs + role * SSZ + (4 * lh) * SROW;          // Qs, Ks, Vs are consecutive
        const int n = (3 * b + role) * 256 + j * 64 + 2 * li;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        if (pair) {
            const float b0 = p.bias ? p.bias[n] : 0.f, b1 = p.bias ? p.bias[n + 1] : 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const f32x2 v = {acc[a][0][r] + b0, acc[a][1][r] + b1};
                    *reinterpret_cast<unsigned*>(panel + (a * 32 + (r & 3) + 8 * (r >> 2)) * SROW + ((2 * li + 16 * (2 * a + (r >> 3))) & 63)) =
                        __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
                }
        } else {
            const float b0 = p.bias ? p.bias[n + cv] : 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    panel[(a * 32 + (r & 3) + 8 * (r >> 2)) * SROW + ((2 * li + cv + 16 * (2 * a + (r >> 3))) & 63)] = (__bf16)(acc[a][0][r] + b0);
        }
    };
    // K-steps 4 j .. 4 j + 3 of to_out on the slab's attention output (rotated rows), this wave's 64 columns
    auto kloop_out = [&]() __attribute__((always_inline)) {
        const __bf16* Ab = Os + li * SROW;
        const int rot0 = 16 * (li >> 4), rot1 = 16 * (2 + (li >> 4));
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16x8 fa0 = *reinterpret_cast<const bf16x8*>(Ab + ((t * 16 + lh * 8 + rot0) & 63));
            const bf16x8 fa1 = *reinterpret_cast<const bf16x8*>(Ab + 32 * SROW + ((t * 16 + lh * 8 + rot1) & 63));
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bf16x8 fb = __builtin_bit_cast(bf16x8, bt[t][c]);
                acct[0][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb, acct[0][c], 0, 0, 0);
                acct[1][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb, acct[1][c], 0, 0, 0);
            }
        }
    };
    // to_out's epilogue of branch b (as the kernel above): bias, one rounding, quad transposes, 16-byte stores into the block's output rows
    auto epilogue_out = [&](int b) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(p.out[b] + m0 * LK, 0, (unsigned)span * (LK * 2), 0x00020000);
        const int n = b * 256 + wave * 64 + 2 * li;
        const float b0 = p.bias2[n], b1 = p.bias2[n + 1];
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        const unsigned off0 = (unsigned)((4 * lh + (li & 3)) * 512 + (wave * 64 + 2 * (li & ~3)) * 2);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned R[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int r = 4 * g + jj;
                    const f32x2 v = {acct[a][0][r] + b0, acct[a][1][r] + b1};
                    R[jj] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const unsigned send = odd ? R[2 * m] : R[2 * m + 1];
                    const unsigned got = (unsigned)__builtin_amdgcn_mov_dpp((int)send, 0xB1, 0xf, 0xf, true);
                    R[2 * m] = odd ? got : R[2 * m];
                    R[2 * m + 1] = odd ? R[2 * m + 1] : got;
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const unsigned send = hi ? R[m] : R[m + 2];
                    const unsigned got = (unsigned)__builtin_amdgcn_mov_dpp((int)send, 0x4E, 0xf, 0xf, true);
                    R[m] = hi ? got : R[m];
                    R[m + 2] = hi ? R[m + 2] : got;
                }
                const u32x4 v4 = {R[0], R[1], R[2], R[3]};
                __builtin_amdgcn_raw_buffer_store_b128(v4, rout, off0 + (unsigned)(((2 * a + (g >> 1)) * p.W + 8 * (g & 1)) * 512), 0, LIN_NT);
            }
    };
    // window_attn_mfma_kernel on the slab panels: (window, head-of-the-slab) tasks of the block's four 4 x 4 windows, output -> Os
    auto win4 = [&](auto dhc, const float* pos) __attribute__((always_inline)) {
        constexpr int DH = decltype(dhc)::value, WS = 4, NB = DH / 16, HS = 64 / DH, U = HS;   // 4 HS tasks: U per wave
        const int t = lane & 15, h = lane >> 4;
        const int iy = t >> 2, ix = t & 3;
        const float scale = 1.0f / sqrtf((float)DH);
        float wbias[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) wbias[r] = pos[(h - iy + WS - 1) * (2 * WS - 1) + (r - ix + WS - 1)];
        int head[U], wdw[U], rowt[U];
        lin_f32x4 st[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int tk = wave + 4 * u;
            head[u] = tk % HS; wdw[u] = tk / HS;
            rowt[u] = ((t >> 2) * 16 + 4 * wdw[u] + (t & 3)) * SROW;
            st[u] = lin_f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            float4 qq[U], kk[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int cq = swz((t >> 2) * 16, head[u] * DH + 4 * (h + 4 * g));
                qq[u] = ld_bf16x4(Qs + rowt[u] + cq);
                kk[u] = ld_bf16x4(Ks + rowt[u] + cq);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) st[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(kk[u].x, qq[u].x, st[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < U; ++u) st[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(kk[u].y, qq[u].y, st[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < U; ++u) st[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(kk[u].z, qq[u].z, st[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < U; ++u) st[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(kk[u].w, qq[u].w, st[u], 0, 0, 0);
        }
        float sc[U][4], inv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) { sc[u][r] = st[u][r] * scale + wbias[r]; mx = fmaxf(mx, sc[u][r]); }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float l = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { sc[u][r] = expf(sc[u][r] - mx); l += sc[u][r]; }
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
            inv[u] = 1.0f / l;
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float vv[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) vv[u][r] = (float)Vs[(16 * h + 4 * wdw[u] + r) * SROW + swz(16 * h, head[u] * DH + nb * 16 + t)];
            lin_f32x4 o[U];
#pragma unroll
            for (int u = 0; u < U; ++u) o[u] = lin_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int u = 0; u < U; ++u) o[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[u][r] * inv[u], vv[u][r], o[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) Os[(16 * h + 4 * wdw[u] + r) * SROW + swz(16 * h, head[u] * DH + nb * 16 + t)] = (__bf16)o[u][r];
        }
    };
    // window_attn_kernel<16, 2> on the slab panels: one (token, head-of-the-slab) pair per thread
    auto win2 = [&](const float* pos) __attribute__((always_inline)) {
        constexpr int WS = 2, DHD = 16, HS = 4;
        const float scale = 1.0f / sqrtf((float)DHD);
        const int head = tid % HS, tok = tid / HS;
        const int py = tok >> 4, px = tok & 15;
        const int wy0 = py & ~1, wx0 = px & ~1, iy = py - wy0, ixx = px - wx0;
        float q[DHD], o[DHD];
#pragma unroll
        for (int d = 0; d < DHD; d += 4) {
            const float4 v = ld_bf16x4(Qs + tok * SROW + swz(tok, head * DHD + d));
            q[d] = v.x; q[d + 1] = v.y; q[d + 2] = v.z; q[d + 3] = v.w;
        }
        float sj[WS * WS];
        float mx = -INFINITY;
#pragma unroll
        for (int jj = 0; jj < WS * WS; ++jj) {
            const int jy = jj / WS, jx = jj % WS;
            const int krow = (wy0 + jy) * 16 + wx0 + jx;
            const __bf16* kr = Ks + krow * SROW + swz(krow, head * DHD);      // 16 columns of a head: no wrap inside (16-aligned)
            float a_ = 0.f;
#pragma unroll
            for (int d = 0; d < DHD; d += 4) {
                const float4 k = ld_bf16x4(kr + d);
                a_ = fmaf(q[d], k.x, a_); a_ = fmaf(q[d + 1], k.y, a_); a_ = fmaf(q[d + 2], k.z, a_); a_ = fmaf(q[d + 3], k.w, a_);
            }
            a_ = a_ * scale + pos[(jy - iy + WS - 1) * (2 * WS - 1) + (jx - ixx + WS - 1)];
            sj[jj] = a_;
            mx = fmaxf(mx, a_);
        }
        float l = 0.f;
#pragma unroll
        for (int jj = 0; jj < WS * WS; ++jj) { sj[jj] = expf(sj[jj] - mx); l += sj[jj]; }
        const float inv = 1.0f / l;
#pragma unroll
        for (int d = 0; d < DHD; ++d) o[d] = 0.f;
#pragma unroll
        for (int jj = 0; jj < WS * WS; ++jj) {
            const int jy = jj / WS, jx = jj % WS;
            const int vrow = (wy0 + jy) * 16 + wx0 + jx;
            const __bf16* vr = Vs + vrow * SROW + swz(vrow, head * DHD);
            const float pj = sj[jj] * inv;
#pragma unroll
            for (int d = 0; d < DHD; d += 4) {
                const float4 v = ld_bf16x4(vr + d);
                o[d] = fmaf(pj, v.x, o[d]); o[d + 1] = fmaf(pj, v.y, o[d + 1]); o[d + 2] = fmaf(pj, v.z, o[d + 2]); o[d + 3] = fmaf(pj, v.w, o[d + 3]);
            }
        }
#pragma unroll
        for (int d = 0; d < DHD; d += 4) {
            const f32x4 f = {o[d], o[d + 1], o[d + 2], o[d + 3]};
            *reinterpret_cast<bf16x4*>(Os + tok * SROW + swz(tok, head * DHD + d)) = __builtin_convertvector(f, bf16x4);
        }
    };

    // ---- the twelve slabs.  Phase A of slab g: to_out's K slice of slab g - 1 (its attention output is complete: barrier 2 of g - 1), the
    // finished branch's output stores, the QKV product of slab g and its epilogue into the slab panels (the attention of g - 1 no longer
    // reads them).  Barrier 1.  Phase B: the attention of slab g -> Os (every wave is past its to_out reads of Os).  Barrier 2.
    auto out_slice = [&](int gp) __attribute__((always_inline)) {    // to_out's K slice of slab gp (and the branch's stores after its last slab)
        const int bp = gp >> 2, jp = gp & 3;
        if (jp == 0) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acct[a][c][r] = 0.f;
        }
        kloop_out();
        if constexpr ((AV2X_QW_ABLATE & 4) == 0) { if (jp == 3) epilogue_out(bp); }
        else if (jp == 3 && acct[0][0][0] + acct[1][1][5] == 123.f) epilogue_out(bp);
    };
#pragma unroll 1
    for (int g = 0; g < 12; ++g) {
        const int b = g >> 2, j = g & 3;
        if (g > 0) out_slice(g - 1);
        if constexpr ((AV2X_QW_ABLATE & 16) == 0) kloop_qkv(b, j);
        if (g + 1 < 12) {
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) load_q(bq[s], (g + 1) >> 2, (g + 1) & 3, s);
        }
        if constexpr ((AV2X_QW_ABLATE & 8) == 0) epilogue_qkv(b, j);
        else if (acc[0][0][0] + acc[1][0][5] == 123.f) epilogue_qkv(b, j);
        __syncthreads();
        load_t(b, j);
        if constexpr ((AV2X_QW_ABLATE & 2) == 0) {
            if (p.dh[b] == 16) win2(posl + 64 * b);
            else if (p.dh[b] == 32) win4(std::integral_constant<int, 32>{}, posl + 64 * b);
            else win4(std::integral_constant<int, 64>{}, posl + 64 * b);
        }
        __syncthreads();
    }
    out_slice(11);
}

template <bool OUT16>
__global__ __launch_bounds__(256, 2) void linear_bf16_kernel(const LinParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lin_smem[];
    __bf16* As = reinterpret_cast<__bf16*>(lin_smem);        // [128][264]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const long long m0 = (long long)blockIdx.x * LBM;
    const int rows = (int)((p.M - m0) < LBM ? (p.M - m0) : LBM);
    const int nchunks = p.CoutP >> 8;

    // ---- weights: packed column of tile c of this wave in chunk ch = 256 ch + 64 wave + 32 c + li; k-oct 2 s + lh
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.w), 0, p.w_bytes, 0x00020000);
    const unsigned voffB = (unsigned)((lh * p.CoutP + wave * 64 + li) * 16);
    const unsigned step_stride = (unsigned)(2 * p.CoutP * 16);
    u32x4 bf[4][2];   // four K-steps in flight (16 steps per chunk: the rotation phase is the same in every chunk)
    auto loadB = [&](u32x4 (&dst)[2], int ch, int s) {
        const unsigned so = (unsigned)s * step_stride + (unsigned)ch * (256 * 16);
        dst[0] = __builtin_amdgcn_raw_buffer_load_b128(rw, voffB, so, 0);
        dst[1] = __builtin_amdgcn_raw_buffer_load_b128(rw, voffB + 32 * 16, so, 0);
    };
    loadB(bf[0], 0, 0);
    loadB(bf[1], 0, 1);
    loadB(bf[2], 0, 2);
    loadB(bf[3], 0, 3);

    // ---- A panel: 128 rows x 512 B = 4096 16-byte pieces, 16 per thread, all in flight together
    {
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.a + m0 * LK), 0,
                                                                            (unsigned)rows * (LK * 2), 0x00020000);
        u32x4 va[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int e = tid + 256 * j;
            va[j] = __builtin_amdgcn_raw_buffer_load_b128(ra, (unsigned)(e * 16), 0, 0);   // rows beyond M: zero
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int e = tid + 256 * j;
            *reinterpret_cast<u32x4*>(As + (e >> 5) * LROW + (e & 31) * 8) = va[j];
        }
    }
    __syncthreads();

    // output / residual rows of this panel: [0, rows) x row stride bytes
    const size_t obytes = (size_t)p.out_ctot * (OUT16 ? 2 : 4);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(p.out) + (size_t)m0 * obytes, 0,
                                                                          (unsigned)((size_t)rows * obytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.res ? p.res + (size_t)m0 * p.res_ctot : nullptr), 0, p.res ? (unsigned)((size_t)rows * p.res_ctot * 4) : 0u, 0x00020000);
    const __bf16* Ab = As + li * LROW + lh * 8;
    for (int ch = 0; ch < nchunks; ++ch) {
        f32x16 acc[4][2];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
        const int chn = ch + 1 < nchunks ? ch + 1 : ch;   // prefetch target beyond the last chunk: a harmless re-read
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            bf16x8 fa[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) fa[a] = *reinterpret_cast<const bf16x8*>(Ab + a * 32 * LROW + s * 16);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bf16x8 fb = __builtin_bit_cast(bf16x8, bf[s & 3][c]);
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb, acc[a][c], 0, 0, 0);
            }
            if (s + 4 < 16) loadB(bf[s & 3], ch, s + 4);
            else loadB(bf[s & 3], chn, s + 4 - 16);
            __builtin_amdgcn_sched_barrier(0);   // keeps the A fragments / B prefetches of later steps from being hoisted (registers)
        }
        // ---- epilogue of the chunk: bias, activation, residual, two consecutive columns per lane.  Row addressing is 32-bit
        // (buffer resources based at the panel's first row) with strides made opaque per chunk: otherwise the row offsets
        // of output and residual are hoisted out of the chunk loop as 64-bit pointers and spilled
        const int n = ch * 256 + wave * 64 + 2 * li;
        if (n < p.Cout) {
            float b0 = 0.f, b1 = 0.f;
            if (p.bias) { b0 = p.bias[n]; b1 = p.bias[n + 1]; }
            int ostride = p.out_ctot * (OUT16 ? 2 : 4), rstride = p.res_ctot * 4;
            asm volatile("" : "+s"(ostride), "+s"(rstride));
            const int row0 = 4 * lh;
            const unsigned ocol = (unsigned)((p.out_coff + n) * (OUT16 ? 2 : 4)), rcol = (unsigned)((p.res_coff + n) * 4);
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = row0 + a * 32 + (r & 3) + 8 * (r >> 2);
                    f32x2 v = {acc[a][0][r] + b0, acc[a][1][r] + b1};
                    if (p.act == 1) {
                        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f);
                    } else if (p.act == 2) {   // GELU (nn.GELU(): 0.5 x (1 + erf(x / sqrt 2)))
                        if (OUT16) {
                            v[0] = 0.5f * v[0] * (1.0f + erf_as(v[0] * 0.70710678118654752f));
                            v[1] = 0.5f * v[1] * (1.0f + erf_as(v[1] * 0.70710678118654752f));
                        } else {
                            v[0] = 0.5f * v[0] * (1.0f + erff(v[0] * 0.70710678118654752f));
                            v[1] = 0.5f * v[1] * (1.0f + erff(v[1] * 0.70710678118654752f));
                        }
                    }
                    // rows beyond M: the buffer range ends at the last valid row, loads return zero and stores are dropped
                    if (p.res) v += __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rres, (unsigned)(ml * rstride) + rcol, 0, 0));
                    if (OUT16)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)), rout,
                                                              (unsigned)(ml * ostride) + ocol, 0, 0);
                    else
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rout, (unsigned)(ml * ostride) + ocol, 0, 0);
                    if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // at most four rows of residual loads in flight (registers)
                }
        }
    }
}

// one wave per token, C = 256: the lane's four channels
// ADD: x += delta first (delta = the bf16 output of the preceding Linear: `x + fn(x)` of PreNormResidual under autocast adds a
// 16-bit Linear output to the fp32 stream), x written back; y == nullptr: only the add
template <bool ADD>
__global__ __launch_bounds__(256) void layernorm_bf16_kernel(float4* __restrict__ x, const __bf16* __restrict__ delta,
                                                             const float4* __restrict__ gamma,
                                                             const float4* __restrict__ beta, __bf16* __restrict__ y,
                                                             long long n_tokens, float eps) {
    const int lane = threadIdx.x & 63;
    const long long tok = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= n_tokens) return;
    constexpr int C = 256;
    float4 v = x[(size_t)tok * (C / 4) + lane];
    if (ADD) {
        v = add_bf16x4(v, *reinterpret_cast<const uint2*>(delta + (size_t)tok * C + 4 * lane));
        x[(size_t)tok * (C / 4) + lane] = v;
        if (!y) return;
    }
    const f32x4 r = layernorm_row_256(v, gamma[lane], beta[lane], eps);
    *reinterpret_cast<bf16x4*>(y + (size_t)tok * C + 4 * lane) = __builtin_convertvector(r, bf16x4);
}

}  // namespace

extern "C" int av2x_linear_bf16(const uint16_t* a, const uint16_t* w_packed, const float* bias, const float* residual, void* out,
                                int64_t m, int32_t k, int32_t cout, int32_t coutp, int32_t out_is_bf16, int32_t out_ctot,
                                int32_t out_coff, int32_t res_ctot, int32_t res_coff, int32_t act, av2x_stream_t stream) {
    if (m == 0) return 0;
    if (!a || !w_packed || !out) return av2x::fail("av2x_linear_bf16: null argument");
    if (k != LK) return av2x::fail("av2x_linear_bf16: k=%d unsupported (256)", k);
    if (m < 0 || coutp <= 0 || coutp % 256 || cout <= 0 || cout > coutp || cout % 2)
        return av2x::fail("av2x_linear_bf16: bad sizes (m=%lld cout=%d coutp=%d: coutp must be a multiple of 256, cout of 2)", (long long)m, cout, coutp);
    if (out_ctot % 2 || out_coff % 2 || out_coff + cout > out_ctot) return av2x::fail("av2x_linear_bf16: bad output slice");
    if (residual && (res_ctot % 2 || res_coff % 2 || res_coff + cout > res_ctot)) return av2x::fail("av2x_linear_bf16: bad residual slice");
    if (act < 0 || act > 2) return av2x::fail("av2x_linear_bf16: act=%d unsupported (0 none, 1 ReLU, 2 GELU)", act);
    if (residual && out_is_bf16) return av2x::fail("av2x_linear_bf16: the residual stream is fp32");
    if (out_is_bf16 && (cout % 8 || out_ctot % 8 || out_coff % 8))
        return av2x::fail("av2x_linear_bf16: bf16 output needs cout, out_ctot and out_coff to be multiples of 8 (16-byte row stores)");
    LinParams p;
    p.a = reinterpret_cast<const __bf16*>(a);
    p.w = reinterpret_cast<const __bf16*>(w_packed);
    p.bias = bias; p.res = residual; p.out = out; p.M = m;
    p.Cout = cout; p.CoutP = coutp; p.out_ctot = out_ctot; p.out_coff = out_coff; p.res_ctot = res_ctot; p.res_coff = res_coff; p.act = act;
    p.w_bytes = (unsigned)((size_t)(LK / 8) * coutp * 16);
    const size_t lds = (size_t)LBM * LROW * 2;
    const unsigned grid = (unsigned)((m + LBM - 1) / LBM);
    hipStream_t st = av2x::as_stream(stream);
    if (out_is_bf16) {
        hipLaunchKernelGGL((linear_bf16_occ_kernel<SRC_ROWS, false>), dim3((unsigned)((m + 63) / 64)), dim3(256), (size_t)64 * LROW * 2, st, p);
    } else {
        static av2x::LdsLimit lim;
        lim.ensure(reinterpret_cast<const void*>(&linear_bf16_kernel<false>), lds);
        hipLaunchKernelGGL(linear_bf16_kernel<false>, dim3(grid), dim3(256), lds, st, p);
    }
    return av2x::check_launch("linear_bf16_kernel");
}

extern "C" int av2x_ln_linear_bf16(float* x, const uint16_t* delta, int64_t add_rows, int32_t write_back_x, const float* gamma, const float* beta, float eps,
                                   const uint16_t* w_packed, const float* bias, int32_t act, int32_t cout, int32_t coutp,
                                   const uint16_t* w2_packed, const float* bias2, int32_t act2, uint16_t* out, int32_t out_ctot,
                                   int32_t out_coff, int64_t m, av2x_stream_t stream) {
    if (m == 0) return 0;
    if (!x || !gamma || !beta || !w_packed || !out) return av2x::fail("av2x_ln_linear_bf16: null argument");
    if (m < 0 || add_rows < 0 || add_rows > m || (add_rows && !delta))
        return av2x::fail("av2x_ln_linear_bf16: bad row counts (m=%lld add_rows=%lld)", (long long)m, (long long)add_rows);
    if (coutp <= 0 || coutp % 256 || cout <= 0 || cout > coutp || cout % 8 || out_ctot % 8 || out_coff % 8)
        return av2x::fail("av2x_ln_linear_bf16: bad sizes (cout=%d coutp=%d out_ctot=%d out_coff=%d)", cout, coutp, out_ctot, out_coff);
    if (act < 0 || act > 2 || act2 < 0 || act2 > 2) return av2x::fail("av2x_ln_linear_bf16: act unsupported (0 none, 1 ReLU, 2 GELU)");
    if (w2_packed && (cout != 256 || coutp != 256)) return av2x::fail("av2x_ln_linear_bf16: the fused second Linear needs a hidden width of 256");
    if (out_coff + (w2_packed ? 256 : cout) > out_ctot) return av2x::fail("av2x_ln_linear_bf16: bad output slice");
    LinParams p = {};
    p.a = nullptr;
    p.w = reinterpret_cast<const __bf16*>(w_packed);
    p.bias = bias; p.res = nullptr; p.out = out; p.M = m;
    p.Cout = cout; p.CoutP = coutp; p.out_ctot = out_ctot; p.out_coff = out_coff; p.res_ctot = 0; p.res_coff = 0; p.act = act;
    p.w_bytes = (unsigned)((size_t)(LK / 8) * coutp * 16);
    p.x = x; p.delta = reinterpret_cast<const __bf16*>(delta); p.add_rows = add_rows; p.write_back = write_back_x; p.gamma = gamma; p.beta = beta; p.eps = eps;
    p.w2 = reinterpret_cast<const __bf16*>(w2_packed); p.bias2 = bias2; p.act2 = act2;
    hipStream_t st = av2x::as_stream(stream);
    const dim3 grid((unsigned)((m + 63) / 64)), block(256);
    const size_t lds = (size_t)64 * LROW * 2;
    if (w2_packed) hipLaunchKernelGGL((linear_bf16_occ_kernel<SRC_LN, true>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((linear_bf16_occ_kernel<SRC_LN, false>), grid, block, lds, st, p);
    return av2x::check_launch("linear_bf16_occ_kernel<LN>");
}

extern "C" int av2x_combine_ln_linear_bf16(float* x, const uint16_t* delta, int64_t add_rows, const uint16_t* s0, const uint16_t* s1,
                                          const uint16_t* s2, const float* logits, int64_t hw, const float* gamma, const float* beta, float eps,
                                          const uint16_t* w_packed, const float* bias, int32_t act, int32_t cout, int32_t coutp,
                                          const uint16_t* w2_packed, const float* bias2, int32_t act2, uint16_t* out, int32_t out_ctot,
                                          int32_t out_coff, int64_t m, av2x_stream_t stream) {
    if (m == 0) return 0;
    if (!x || !s0 || !s1 || !s2 || !logits || !gamma || !beta || !w_packed || !out) return av2x::fail("av2x_combine_ln_linear_bf16: null argument");
    if (m < 0 || add_rows < 0 || add_rows > m || (add_rows && !delta))
        return av2x::fail("av2x_combine_ln_linear_bf16: bad row counts (m=%lld add_rows=%lld)", (long long)m, (long long)add_rows);
    if (hw <= 0 || hw % 64 || m % hw) return av2x::fail("av2x_combine_ln_linear_bf16: hw=%lld must be a multiple of 64 and divide m=%lld", (long long)hw, (long long)m);
    if (coutp <= 0 || coutp % 256 || cout <= 0 || cout > coutp || cout % 8 || out_ctot % 8 || out_coff % 8)
        return av2x::fail("av2x_combine_ln_linear_bf16: bad sizes (cout=%d coutp=%d out_ctot=%d out_coff=%d)", cout, coutp, out_ctot, out_coff);
    if (act < 0 || act > 2 || act2 < 0 || act2 > 2) return av2x::fail("av2x_combine_ln_linear_bf16: act unsupported (0 none, 1 ReLU, 2 GELU)");
    if (w2_packed && (cout != 256 || coutp != 256)) return av2x::fail("av2x_combine_ln_linear_bf16: the fused second Linear needs a hidden width of 256");
    if (out_coff + (w2_packed ? 256 : cout) > out_ctot) return av2x::fail("av2x_combine_ln_linear_bf16: bad output slice");
    LinParams p = {};
    p.w = reinterpret_cast<const __bf16*>(w_packed);
    p.bias = bias; p.out = out; p.M = m;
    p.Cout = cout; p.CoutP = coutp; p.out_ctot = out_ctot; p.out_coff = out_coff; p.act = act;
    p.w_bytes = (unsigned)((size_t)(LK / 8) * coutp * 16);
    p.x = x; p.delta = reinterpret_cast<const __bf16*>(delta); p.add_rows = add_rows; p.write_back = 1; p.gamma = gamma; p.beta = beta; p.eps = eps;
    p.w2 = reinterpret_cast<const __bf16*>(w2_packed); p.bias2 = bias2; p.act2 = act2;
    p.br[0] = reinterpret_cast<const __bf16*>(s0); p.br[1] = reinterpret_cast<const __bf16*>(s1); p.br[2] = reinterpret_cast<const __bf16*>(s2);
    p.logits = logits; p.hw = hw;
    hipStream_t st = av2x::as_stream(stream);
    const dim3 grid((unsigned)((m + 63) / 64)), block(256);
    const size_t lds = (size_t)64 * LROW * 2;
    if (w2_packed) hipLaunchKernelGGL((linear_bf16_occ_kernel<SRC_LNC, true>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((linear_bf16_occ_kernel<SRC_LNC, false>), grid, block, lds, st, p);
    return av2x::check_launch("linear_bf16_occ_kernel<LNC>");
}

extern "C" int av2x_window_attention_linear_bf16(const uint16_t* qkv, int32_t ctot, int32_t coff, const float* pos_embedding,
                                                const uint16_t* w_packed, const float* bias, uint16_t* out, int32_t out_ctot,
                                                int32_t out_coff, int32_t n, int32_t h, int32_t w, int32_t heads, int32_t dim_head,
                                                int32_t window, av2x_stream_t stream) {
    if (n == 0) return 0;
    if (!qkv || !pos_embedding || !w_packed || !out) return av2x::fail("av2x_window_attention_linear_bf16: null argument");
    if (n < 0 || h <= 0 || w <= 0 || h % 4 || w % 16)
        return av2x::fail("av2x_window_attention_linear_bf16: the map (%d x %d) must split into 4 x 16-pixel blocks", h, w);
    if (heads * dim_head != 256) return av2x::fail("av2x_window_attention_linear_bf16: heads x dim_head must be 256");
    if (!((window == 4 && (dim_head == 32 || dim_head == 64)) || (window == 2 && dim_head == 16)))
        return av2x::fail("av2x_window_attention_linear_bf16: (dim_head %d, window %d) unsupported: (16,2) (32,4) (64,4)", dim_head, window);
    if (ctot % 8 || coff % 8 || coff + 768 > ctot || out_ctot % 8 || out_coff % 8 || out_coff + 256 > out_ctot)
        return av2x::fail("av2x_window_attention_linear_bf16: bad slices");
    if ((unsigned long long)(3 * (unsigned long long)w + 16) * out_ctot * 2ull >= (1ull << 31)) return av2x::fail("av2x_window_attention_linear_bf16: map too wide");
    LinParams p = {};
    p.w = reinterpret_cast<const __bf16*>(w_packed);
    p.bias = bias; p.out = out; p.M = (long long)n * h * w;
    p.Cout = 256; p.CoutP = 256; p.out_ctot = out_ctot; p.out_coff = out_coff; p.act = 0;
    p.w_bytes = (unsigned)((size_t)(LK / 8) * 256 * 16);
    p.qkv = reinterpret_cast<const __bf16*>(qkv); p.pos = pos_embedding; p.qkv_ctot = ctot; p.qkv_coff = coff; p.H = h; p.W = w; p.heads = heads;
    p.scale = 1.0f / sqrtf((float)dim_head);
    hipStream_t st = av2x::as_stream(stream);
    const long long blocks = (long long)n * (h / 4) * (w / 16);
    if (blocks > (1ll << 30)) return av2x::fail("av2x_window_attention_linear_bf16: too many workgroups");
    const dim3 grid((unsigned)blocks), block(256);
    const size_t lds = (size_t)64 * LROW * 2;
    if (window == 2) hipLaunchKernelGGL((linear_bf16_occ_kernel<SRC_WIN2, false, 16>), grid, block, lds, st, p);
    else if (dim_head == 32) hipLaunchKernelGGL((linear_bf16_occ_kernel<SRC_WIN4, false, 32>), grid, block, lds + 4 * 16 * 32 * 2, st, p);
    else hipLaunchKernelGGL((linear_bf16_occ_kernel<SRC_WIN4, false, 64>), grid, block, lds + 4 * 16 * 64 * 2, st, p);
    return av2x::check_launch("linear_bf16_occ_kernel<WIN>");
}

extern "C" int av2x_ln_qkv_window_attention_bf16(const float* x, const uint16_t* delta, const float* gamma, const float* beta, float eps,
                                                const uint16_t* wqkv_packed, const float* bias_qkv, const uint16_t* wout3_packed,
                                                const float* bias_out3, const float* const* pos_embeddings, uint16_t* const* outs,
                                                const int32_t* heads, const int32_t* dim_heads, const int32_t* windows, int32_t n,
                                                int32_t h, int32_t w, av2x_stream_t stream) {
    if (n == 0) return 0;
    if (!x || !gamma || !beta || !wqkv_packed || !wout3_packed || !bias_out3 || !pos_embeddings || !outs || !heads || !dim_heads || !windows)
        return av2x::fail("av2x_ln_qkv_window_attention_bf16: null argument");
    if (n < 0 || h <= 0 || w <= 0 || h % 4 || w % 16)
        return av2x::fail("av2x_ln_qkv_window_attention_bf16: the map (%d x %d) must split into 4 x 16-pixel blocks", h, w);
    if ((unsigned long long)(3 * (unsigned long long)w + 16) * 1024ull >= (1ull << 32)) return av2x::fail("av2x_ln_qkv_window_attention_bf16: map too wide");
    QwParams p = {};
    p.x = x; p.delta = reinterpret_cast<const __bf16*>(delta); p.gamma = gamma; p.beta = beta; p.eps = eps;
    p.w = reinterpret_cast<const __bf16*>(wqkv_packed); p.bias = bias_qkv;
    p.w2 = reinterpret_cast<const __bf16*>(wout3_packed); p.bias2 = bias_out3;
    for (int b = 0; b < 3; ++b) {
        if (!pos_embeddings[b] || !outs[b]) return av2x::fail("av2x_ln_qkv_window_attention_bf16: null argument");
        if (heads[b] * dim_heads[b] != 256 ||
            !((windows[b] == 4 && (dim_heads[b] == 32 || dim_heads[b] == 64)) || (windows[b] == 2 && dim_heads[b] == 16)))
            return av2x::fail("av2x_ln_qkv_window_attention_bf16: branch %d (heads %d, dim_head %d, window %d) unsupported: heads x dim_head = 256, "
                              "(dim_head, window) in (16,2) (32,4) (64,4)", b, heads[b], dim_heads[b], windows[b]);
        p.pos[b] = pos_embeddings[b]; p.out[b] = reinterpret_cast<__bf16*>(outs[b]); p.heads[b] = heads[b]; p.dh[b] = dim_heads[b];
    }
    p.H = h; p.W = w;
    const long long blocks = (long long)n * (h / 4) * (w / 16);
    if (blocks > (1ll << 30)) return av2x::fail("av2x_ln_qkv_window_attention_bf16: too many workgroups");
    // AV2X_QW_SLAB=0: the one-workgroup-per-CU form (four full-width panels); default: head slabs, two workgroups per CU (same bits)
    static const bool slab = [] { const char* e = getenv("AV2X_QW_SLAB"); return !(e && e[0] == '0'); }();
    if (slab) {
        size_t lds = (size_t)(64 * LROW + 4 * 64 * SROW) * 2 + 192 * 4;
        if (const char* pad = getenv("AV2X_QW_LDS_PAD")) lds += (size_t)atoi(pad);   // occupancy probe (tools/qw_bench.py): > 10 KB leaves one workgroup per CU
        static av2x::LdsLimit lim2;
        lim2.ensure(reinterpret_cast<const void*>(&ln_qkv_window_out_slab_kernel), lds);
        hipLaunchKernelGGL(ln_qkv_window_out_slab_kernel, dim3((unsigned)blocks), dim3(256), lds, av2x::as_stream(stream), p);
        return av2x::check_launch("ln_qkv_window_out_slab_kernel");
    }
    const size_t lds = (size_t)4 * 64 * LROW * 2 + (3072 + 192) * 4;
    static av2x::LdsLimit lim;
    lim.ensure(reinterpret_cast<const void*>(&ln_qkv_window_out_bf16_kernel), lds);
    hipLaunchKernelGGL(ln_qkv_window_out_bf16_kernel, dim3((unsigned)blocks), dim3(512), lds, av2x::as_stream(stream), p);
    return av2x::check_launch("ln_qkv_window_out_bf16_kernel");
}

extern "C" int av2x_add_layernorm_bf16(float* x, const uint16_t* delta, const float* gamma, const float* beta, uint16_t* y,
                                       int64_t n_tokens, int32_t c, float eps, av2x_stream_t stream) {
    if (n_tokens == 0) return 0;
    if (!x || (!delta && !y)) return av2x::fail("av2x_add_layernorm_bf16: null argument");
    if (y && (!gamma || !beta)) return av2x::fail("av2x_add_layernorm_bf16: gamma / beta are required with y");
    if (c != 256) return av2x::fail("av2x_add_layernorm_bf16: c=%d unsupported (256)", c);
    if (n_tokens < 0) return av2x::fail("av2x_add_layernorm_bf16: bad token count");
    const dim3 grid((unsigned)((n_tokens + 3) / 4)), block(256);
    hipStream_t st = av2x::as_stream(stream);
    auto X = reinterpret_cast<float4*>(x);
    auto G = reinterpret_cast<const float4*>(gamma);
    auto Bt = reinterpret_cast<const float4*>(beta);
    auto Y = reinterpret_cast<__bf16*>(y);
    if (delta) hipLaunchKernelGGL(layernorm_bf16_kernel<true>, grid, block, 0, st, X, reinterpret_cast<const __bf16*>(delta), G, Bt, Y, (long long)n_tokens, eps);
    else hipLaunchKernelGGL(layernorm_bf16_kernel<false>, grid, block, 0, st, X, (const __bf16*)nullptr, G, Bt, Y, (long long)n_tokens, eps);
    return av2x::check_launch("layernorm_bf16_kernel");
}

extern "C" int av2x_layernorm_bf16(const float* x, const float* gamma, const float* beta, uint16_t* y, int64_t n_tokens, int32_t c,
                                   float eps, av2x_stream_t stream) {
    if (n_tokens && !y) return av2x::fail("av2x_layernorm_bf16: null argument");
    return av2x_add_layernorm_bf16(const_cast<float*>(x), nullptr, gamma, beta, y, n_tokens, c, eps, stream);
}

// ---- bf16 -> fp32 widening (exact): the feature-sharing message of the autocast frame is the shrink header's bf16 output (what
// torch.autocast stores for that Conv2d; 18.0 MB per agent at the default grid); the fusion's residual stream is fp32.
// HBM-bound: 16-byte loads of 8 values, two 16-byte stores.
namespace {
__global__ __launch_bounds__(256) void widen_bf16_kernel(const uint4* __restrict__ src, float4* __restrict__ dst, size_t n8,
                                                         const uint16_t* __restrict__ tail_src, float* __restrict__ tail_dst, int ntail) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const uint4 v = src[i];
        float4 a, b;
        a.x = __builtin_bit_cast(float, v.x << 16); a.y = __builtin_bit_cast(float, v.x & 0xffff0000u);
        a.z = __builtin_bit_cast(float, v.y << 16); a.w = __builtin_bit_cast(float, v.y & 0xffff0000u);
        b.x = __builtin_bit_cast(float, v.z << 16); b.y = __builtin_bit_cast(float, v.z & 0xffff0000u);
        b.z = __builtin_bit_cast(float, v.w << 16); b.w = __builtin_bit_cast(float, v.w & 0xffff0000u);
        dst[2 * i] = a;
        dst[2 * i + 1] = b;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail)
        tail_dst[threadIdx.x] = __builtin_bit_cast(float, (unsigned)tail_src[threadIdx.x] << 16);
}
}  // namespace

extern "C" int av2x_bf16_to_f32(const uint16_t* src, float* dst, uint64_t n_elems, av2x_stream_t stream) {
    if (n_elems == 0) return 0;
    if (!src || !dst) return av2x::fail("av2x_bf16_to_f32: null argument");
    if (reinterpret_cast<uintptr_t>(src) % 16 || reinterpret_cast<uintptr_t>(dst) % 16)
        return av2x::fail("av2x_bf16_to_f32: pointers must be 16-byte aligned");
    const size_t n8 = n_elems / 8;
    const int ntail = (int)(n_elems - n8 * 8);
    const size_t want = (n8 + 255) / 256;
    const unsigned blocks = (unsigned)(want < 1 ? 1 : want > 256 * 16 ? 256 * 16 : want);
    hipLaunchKernelGGL(widen_bf16_kernel, dim3(blocks), dim3(256), 0, av2x::as_stream(stream), reinterpret_cast<const uint4*>(src),
                       reinterpret_cast<float4*>(dst), n8, src + n8 * 8, dst + n8 * 8, ntail);
    return av2x::check_launch("widen_bf16_kernel");
}
