/*
 * airv2x_hip.h — C-ABI of the MI355X (gfx950) hot-path library `libairv2x_hip.so`.
 *
 * The reference (taco-group/AirV2X-Perception, an OpenCOOD fork) has NO C/FFI plugin
 * boundary: its hot path is Python calling ATen ops.  This header is therefore the
 * build-side definition of the boundary: each entry point replaces the
 * ATen call sequence of one reference function, cited as `file:line` relative to
 * /root/reference/opencood.  The Python host mirror (airv2x_perception_amd/opencood_iface)
 * binds these with ctypes; INTEGRATION.md shows the stub a reference maintainer adds.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HIP) unless the comment says HOST; tensors are dense
 *     fp32 / int32;
 *   - BEV activations are NHWC ("channels-last": index ((n*H + h)*W + w)*ctot + c);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on
 *     it, nothing allocates, nothing synchronises, every function is re-entrant;
 *   - return value: 0 = ok, non-zero = error (message via av2x_last_error(), thread-local).
 */
#ifndef AIRV2X_HIP_H
#define AIRV2X_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AV2X_ABI_VERSION 1

typedef void* av2x_stream_t;

int av2x_version(void);
const char* av2x_last_error(void);

/* ------------------------------------------------------------------------------------
 * Pillar feature net + BEV scatter.
 * Replaces PillarVFE.forward + PFNLayer.forward
 *   (models/common_modules/airv2x_pillar_vfe.py:105-160, :27-45) and
 * PointPillarScatter.forward (models/common_modules/point_pillar_scatter.py:43-80).
 *   voxel_features (M,32,4) f32 zero padded; voxel_coords (M,4) i32 [agent,z,y,x];
 *   voxel_num_points (M,) i32; pfn_w (64,10) row-major (linear.weight);
 *   bn_scale/bn_shift (64,): eval BatchNorm1d folded to y = x*scale + shift;
 *   geom: HOST array of 6 floats {voxel_x, voxel_y, voxel_z, x_offset, y_offset, z_offset} of the agent TYPE
 *   (airv2x_pillar_vfe.py:84-89);
 *   canvas (n_agents, ny, nx, 64) NHWC — MUST be zero-filled by the caller beforehand
 *   (av2x_fill_zero); agent k of this type (voxel_coords[:,0] == k) is written to canvas slot
 *   slot_map[k] (device, n_agents_type entries) or, if slot_map is NULL, canvas_agent0 + k.
 * ------------------------------------------------------------------------------------ */
int av2x_pillar_vfe_scatter(const float* voxel_features, const int32_t* voxel_coords,
                            const int32_t* voxel_num_points, int32_t n_pillars,
                            const float* pfn_w, const float* bn_scale, const float* bn_shift,
                            const float* geom, float* canvas, int32_t canvas_agent0,
                            const int32_t* slot_map, int32_t n_agents_type, int32_t ny, int32_t nx,
                            av2x_stream_t stream);

/* The two halves on their own, for callers that use the reference's sub-modules separately:
 * av2x_pillar_vfe     = PillarVFE.forward (airv2x_pillar_vfe.py:105-160): pillar_features (M,64) f32.
 * av2x_pillar_scatter = PointPillarScatter.forward (point_pillar_scatter.py:39-80): pillar_features (M,channels),
 *   voxel_coords (M,4) [agent,z,y,x] -> canvas (n_agents, ny, nx, channels) NHWC, zero-filled by the caller;
 *   channels % 4 == 0; pillars whose agent / y / x fall outside the canvas are skipped. */
int av2x_pillar_vfe(const float* voxel_features, const int32_t* voxel_coords, const int32_t* voxel_num_points,
                    int32_t n_pillars, const float* pfn_w, const float* bn_scale, const float* bn_shift,
                    const float* geom, float* pillar_features, av2x_stream_t stream);
/* av2x_pillar_vfe_scatter_dev = av2x_pillar_vfe_scatter fed straight from the voxelizer's outputs of ONE agent:
 *   voxel_coords3 (capacity,3) z,y,x (no agent column; the agent's canvas slot is canvas_slot), pillar count read from
 *   DEVICE memory (n_pillars_dev, clamped to capacity) -- the frame needs no host read-back between voxelizer and network. */
int av2x_pillar_vfe_scatter_dev(const float* voxel_features, const int32_t* voxel_coords3, const int32_t* voxel_num_points,
                                const int32_t* n_pillars_dev, int32_t capacity, const float* pfn_w, const float* bn_scale,
                                const float* bn_shift, const float* geom, float* canvas, int32_t canvas_slot, int32_t ny,
                                int32_t nx, av2x_stream_t stream);
int av2x_pillar_scatter(const float* pillar_features, const int32_t* voxel_coords, int32_t n_pillars, int32_t channels,
                        float* canvas, int32_t n_agents, int32_t ny, int32_t nx, av2x_stream_t stream);
/* The two scatter entries that also COUNT what they write: nonzero (device; NULL = no count) is an array of AV2X_NZ_SLOTS u64 counters
 * that live AV2X_NZ_STRIDE u64 (128 bytes) apart -- AV2X_NZ_SLOTS * AV2X_NZ_STRIDE * 8 = 4096 bytes, zeroed by the caller; workgroup b
 * adds the number of non-zero values it put on the canvas to counter b % AV2X_NZ_SLOTS (one atomic per workgroup, spread over cache
 * lines: thousands of same-address atomics serialise in L2), and the count is the SUM of the array.  With a canvas the caller
 * zero-filled and coordinates that are unique per agent (the voxelizer's are) that is torch.count_nonzero of the scattered canvas -- the `comm_rate` of the LiDAR-only frame (airv2x_where2com.py:122,
 * batch_spatial_features.count_nonzero()) -- without the 144 MB read-back pass of av2x_count_nonzero at the BASELINE grid. */
#define AV2X_NZ_SLOTS 32
#define AV2X_NZ_STRIDE 16
/* result[0] = the sum of the AV2X_NZ_SLOTS counters (a store, not an add) */
int av2x_nonzero_slots_sum(const unsigned long long* slots, unsigned long long* result, av2x_stream_t stream);
int av2x_pillar_vfe_scatter_count(const float* voxel_features, const int32_t* voxel_coords,
                                  const int32_t* voxel_num_points, int32_t n_pillars,
                                  const float* pfn_w, const float* bn_scale, const float* bn_shift,
                                  const float* geom, float* canvas, int32_t canvas_agent0,
                                  const int32_t* slot_map, int32_t n_agents_type, int32_t ny, int32_t nx,
                                  unsigned long long* nonzero, uint8_t* occupancy, av2x_stream_t stream);
int av2x_pillar_vfe_scatter_dev_count(const float* voxel_features, const int32_t* voxel_coords3, const int32_t* voxel_num_points,
                                      const int32_t* n_pillars_dev, int32_t capacity, const float* pfn_w, const float* bn_scale,
                                      const float* bn_shift, const float* geom, float* canvas, int32_t canvas_slot, int32_t ny,
                                      int32_t nx, unsigned long long* nonzero, uint8_t* occupancy, av2x_stream_t stream);
/* occupancy (device; NULL = none): one byte per canvas cell, (canvas slots, ny, nx), zero-filled by the caller; the scatter sets the byte
 * of every cell it writes.  Reader: av2x_conv3x3s2_sparse -- the first convolution of the backbone (base_bev_backbone.py:30-48: ZeroPad2d(1),
 * Conv2d(64, 64, 3, stride 2), BatchNorm folded to scale / shift, ReLU) as a gather over the occupied taps only:
 *   out[n] = act(scale[n] * sum over occupied taps (row-major), channels ascending, of in[2 oy + ky - 1][2 ox + kx - 1][c] W[ky][kx][c][n] + shift[n])
 * in fp32 FMAs; a zero tap contributes exactly 0, so this IS the dense convolution of the scattered canvas (up to the order of fp32
 * roundings).  in (n, h, w, 64) NHWC, w_packed = the (9, 16, 64, 4) packing of the dense entries, scale may be NULL (= 1), out (n, ho, wo, 64). */
int av2x_conv3x3s2_sparse(const float* in, const uint8_t* occupancy, const float* w_packed, const float* scale, const float* shift,
                          int32_t relu, float* out, int32_t n, int32_t h, int32_t w, int32_t cin, int32_t cout, av2x_stream_t stream);

int av2x_fill_zero(void* ptr, uint64_t bytes, av2x_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32), fused per-channel
 * affine (+bias / folded BatchNorm2d) and optional ReLU.
 * Replaces every Conv2d/ConvTranspose2d(+BatchNorm2d)(+ReLU) on the path:
 *   BaseBEVBackbone blocks/deblocks (models/common_modules/base_bev_backbone.py:41-105),
 *   DoubleConv (models/common_modules/downsample_conv.py:17-31),
 *   NaiveCompressor (models/common_modules/naive_compress.py:12-36),
 *   cls/reg/obj heads (models/airv2x_where2com.py:60-69).
 *
 * mode AV2X_CONV      : out[n,ho,wo, out_coff+co] NHWC, square kernels ks = 1 .. 7, any stride/pad.
 * mode AV2X_DECONV    : ConvTranspose2d with kernel == stride == up (no overlap):
 *                       out[n, ho*up+i, wo*up+j, out_coff+co]; (h,w) here are the INPUT dims.
 * mode AV2X_CONV_NCHW : as AV2X_CONV but the result is stored NCHW (out[n,co,ho,wo]); used
 *                       for the detection heads whose consumers expect NCHW.
 *
 * Weight layout `w` (packed once on the host, see opencood_iface/packing.py):
 *   [tap = kh*ks+kw][cin/4][coutp][4]  where element [t][q][n][j] = W[cout=n][cin=4q+j][kh][kw]
 *   (for AV2X_DECONV the GEMM column n = (i*up+j)*cout + co and W = weight[cin][co][i][j]);
 *   coutp = GEMM columns padded with zeros to a multiple of 32; cin % 32 == 0.
 * scale/shift: (cout,) applied as y = acc*scale + shift (scale may be NULL = 1).
 * AMP mode (tile flag 0x0800, any entry point): `w` points to the bf16 k-oct packing [tap][cin/8][coutp][8]
 *   (packing.to_bf16_koct) and the input tile is rounded to bf16 (nearest-even) on its way into LDS; products run on
 *   v_mfma_f32_32x32x16_bf16 with fp32 accumulation, activations stay fp32 in HBM.  The result equals an fp32
 *   convolution of the bf16-rounded operands up to summation order = what torch.autocast computes for Conv2d /
 *   Linear (the reference's validation pass, tools/train.py:212-220).  Tiles 128x128, 128x64, 64x64, 128x32.
 * Split-3 mode (tile flag 0x0400): `w` points to THREE bf16 planes [3][tap][cin/8][coutp][8] = hi, mid, lo with
 *   hi = bf16(w), mid = bf16(w - hi), lo = bf16(w - hi - mid) (packing.to_bf16x3_koct); the input tile is split the
 *   same way on its way into LDS and six partial products per MAC (lo.hi' + hi.lo' + mid.mid' + mid.hi' + hi.mid' +
 *   hi.hi') are accumulated in fp32 on v_mfma_f32_32x32x16_bf16.  Per-product error <= ~4 x 2^-24: fp32-accurate
 *   (its error against fp64 is at or below that of the default fp32-MFMA kernel), not bit-identical to it.
 * ------------------------------------------------------------------------------------ */
enum { AV2X_CONV = 0, AV2X_DECONV = 1, AV2X_CONV_NCHW = 2 };

typedef struct av2x_conv_desc {
    int32_t n, h, w, cin;      /* input tensor (NHWC) and channels consumed             */
    int32_t in_ctot, in_coff;  /* channel stride of an input pixel, first channel used   */
    int32_t ho, wo, cout;      /* output spatial dims (DECONV: unused, = h*up, w*up)     */
    int32_t coutp;             /* padded GEMM column count of `w`                        */
    int32_t out_ctot, out_coff;/* channel stride / offset of the output (concat support) */
    int32_t ks, stride, pad;   /* square kernel                                           */
    int32_t relu;              /* activation after the affine: 0 none, 1 ReLU, 2 exact GELU, 3 sigmoid, 4 tanh -- with a
                                * residual pointer code 4 multiplies by residual[m*cout + c] (a ConvGRU gate) instead of adding;
                                * 5 = ReLU applied AFTER the residual add (torchvision BasicBlock), 6 = swish x*sigmoid(x) */
    int32_t mode;              /* AV2X_CONV / AV2X_DECONV / AV2X_CONV_NCHW                */
    int32_t up;                /* DECONV: kernel == stride                                */
    int32_t tile;              /* 0 = auto; else BM<<16 | BN | 0x8000 (8 waves) | 0x4000 (prefetch distance 2)
                                  | 0x2000 (stream-K, av2x_conv2d_sk only) | 0x1000 (persistent whole tiles)
                                  | 0x0800 (bf16 matrix-core operands: `w` is the bf16 packing, see below)
                                  | 0x0400 (split-3: fp32-accurate products from three bf16 terms, see below) */
    int32_t sk_wgs;            /* stream-K / persistent: number of workgroups launched (else ignored) */
    int32_t act16;             /* bf16 tiles (0x0800) only: bit 0 = the INPUT activations are bf16 in memory, bit 1 = the OUTPUT is written as
                                * bf16 (AMP mode with bf16 activations: what autocast stores between two Conv2d); in_ctot / out_ctot / offsets stay
                                * in elements; served by the 8-wave 128x128 / 128x64 and the 64x64 tiles; no residual with a bf16 output; 0 = fp32 */
} av2x_conv_desc;

int av2x_conv2d(const av2x_conv_desc* d, const float* in, const float* w, const float* scale,
                const float* shift, float* out, av2x_stream_t stream);
/* same, plus `residual` (NULL or a tensor laid out like `out`, AV2X_CONV mode): out = act(affine(acc)) + residual.
 * Used for the transformer Linear layers (a Linear over tokens is a 1x1 convolution):
 * PreNormResidual / FeedForward, models/cobevt_modules/base_transformer.py:6-38. */
int av2x_conv2d_res(const av2x_conv_desc* d, const float* in, const float* w, const float* scale,
                    const float* shift, const float* residual, float* out, av2x_stream_t stream);
/* same, with a stream-K schedule when d->tile has 0x2000 set: d->sk_wgs persistent workgroups split the
 * (output tiles x K-steps) iteration space evenly, so a layer whose tile count does not fill the 256 CUs
 * a whole number of times has no idle tail.  Tiles cut between workgroups are finished by a second
 * (fix-up) launch that adds the partial accumulators from `workspace` in ascending K order, so results are
 * deterministic; they differ from the non-stream-K schedule by fp32 summation order only.
 * workspace: device scratch of av2x_conv2d_sk_workspace_bytes(tile, sk_wgs) bytes (unused without 0x2000).
 * Tile flag 0x1000 (any of the three entry points): d->sk_wgs persistent workgroups each walk a contiguous range
 * of WHOLE tiles as one software-pipelined stream of K-steps (the next tile's loads are in flight during the
 * current tile's epilogue); no K split, so results are bit-identical to the default schedule.  For short-K GEMMs. */
int av2x_conv2d_sk(const av2x_conv_desc* d, const float* in, const float* w, const float* scale,
                   const float* shift, const float* residual, float* out, float* workspace,
                   uint64_t workspace_bytes, av2x_stream_t stream);
uint64_t av2x_conv2d_sk_workspace_bytes(int32_t tile, int32_t sk_wgs);

/* Winograd F(2x2,3x3) form of a 3x3 / stride 1 / pad 1 AV2X_CONV layer (tile flag 0x40000000 | TB << 16 | CB on any of
 * the three entry points above; TB x CB = tiles x couts per workgroup: 32x128, 64x64, 32x64, and 32x64 | 0x8000 = the
 * half-position variant (8 of the 16 positions per wave, two workgroups per CU) and 32x32 | 0x8000 = the quarter-position
 * variant (4 per wave, up to four workgroups per CU); all tilings give the same bits): 2.25x fewer
 * matrix-core multiplies, still fp32 operands and fp32 accumulation on v_mfma_f32_32x32x2_f32; results agree with the
 * direct kernel to fp32 rounding (not bit for bit).  `w` must then point to the transformed weights
 *   u [pos = 4*xi + nu][cin/4][coutp][4] = (G g G^T)[xi][nu] in the k-quad packing of `w`,
 * made once per layer by av2x_wino_pack_weights from the ordinary packing (av2x_wino_weight_bytes(cin, coutp) bytes).
 * cin % 8 == 0, cout % CB == 0, activation codes 0, 1, 3, 4 (no GELU); residual as in av2x_conv2d_res. */
uint64_t av2x_wino_weight_bytes(int32_t cin, int32_t coutp);
int av2x_wino_pack_weights(const float* w_packed, int32_t cin, int32_t coutp, float* u, av2x_stream_t stream);
/* Winograd F(2x2,3x3) with SPLIT-3 operands (tile flag 0x40000000 | 0x0400 | TB << 16 | 64, TB = 64 or 32; csrc/conv_wino_x3.hip):
 * the same 16-position algorithm, but every fp32 operand enters the matrix core as three bf16 terms (hi + mid + lo = the fp32
 * value to 2^-24) and the six partial products >= 2^-16 are accumulated in fp32 on v_mfma_f32_32x32x16_bf16 -- fp32-accurate
 * products at 6 x 32 instead of 8 x 64 matrix cycles per 16 input channels.  Replaces the same reference layers as the tiles
 * above (common_modules/base_bev_backbone.py:6-154, downsample_conv.py:8-54); NOT bit-identical to them (error against an fp64
 * convolution at or below theirs, tests/test_gpu_kernels.py).  `w` = u3 [pos][cin/16][plane hi,mid,lo][k half][coutp][8] bf16
 * from av2x_wino_x3_pack_weights (G g G^T in fp64, split there); cin % 16 == 0, cout % 64 == 0, activations 0, 1, 3, 4, 5. */
uint64_t av2x_wino_x3_weight_bytes(int32_t cin, int32_t coutp);
int av2x_wino_x3_pack_weights(const float* w_packed, int32_t cin, int32_t coutp, void* u3, av2x_stream_t stream);

/* Winograd F(4x4,3x3) with split-3 operands (tile = 0x60000000 | 0x0400 | (32 << 16) | 64; csrc/conv_wino4_x3.hip): the 36-position
 * algorithm of av2x_wino4_pack_weights' kernel (2.25 multiplies per output) with every fp32 operand entering the bf16 matrix core as
 * three terms, six partial products accumulated in fp32 -- fp32-accurate products, fp32 transforms: the error against an fp64
 * convolution is that of the fp32 F(4x4) kernel or below (tests/test_gpu_wino4_x3.py).  Same layers as that kernel (the 3x3 / stride 1
 * convolutions of downsample_conv.py:8-54 / base_bev_backbone.py:6-154 the engine's wino4 rule selects).  `w` = u3
 * [pos 36][cin/16][plane hi,mid,lo][k half][coutp][8] bf16 from av2x_wino4_x3_pack_weights; cin % 32 == 0, cout % 64 == 0,
 * activations 0, 1, 3, 4, 5. */
uint64_t av2x_wino4_x3_weight_bytes(int32_t cin, int32_t coutp);
int av2x_wino4_x3_pack_weights(const float* w_packed, int32_t cin, int32_t coutp, void* u3, av2x_stream_t stream);
/* Winograd F(4x4,3x3) (tile flag 0x60000000 | 32 << 16 | 64): 36 products per 4x4 output tile = 2.25 multiplies per output (F(2x2,3x3): 4,
 * direct: 9), fp32 operands and accumulation; cin % 8 == 0, cout % 64 == 0 and cout == coutp; `w` = the transformed packing
 * [36][cin/4][coutp][4] made by av2x_wino4_pack_weights (av2x_wino4_weight_bytes bytes).  Results agree with the other kernels to
 * fp32 rounding (larger transform constants: ~4x the error of F(2x2,3x3) against fp64), not bit for bit. */
uint64_t av2x_wino4_weight_bytes(int32_t cin, int32_t coutp);
int av2x_wino4_pack_weights(const float* w_packed, int32_t cin, int32_t coutp, float* u, av2x_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Where2Comm communication mask.  Replaces Communication.forward, eval branch
 * (models/where2comm_modules/where2comm_fuse.py:83-149).
 *   psm (n,h,w,ctot) NHWC, first `c` channels = anchor*class logits of each agent;
 *   conf (n,h,w) scratch: sigmoid(max_c psm);  gauss_w (k,k), gauss_b (1,): the
 *   `gaussian_filter` Conv2d parameters (state_dict, where2comm_fuse.py:58-62);
 *   sample_of_agent (n,) i32: sample index b of every agent, is_ego (n,) i32: 1 for the
 *   first agent of each sample (mask forced to 1, :141);
 *   mask (n,h,w) f32 in {0,1};  count (B,) i32 += number of ones BEFORE the ego override
 *   (caller zero-fills; rate_b = count_b / (L_b*h*w), :137).
 *   threshold <= 0 means "no threshold": mask = 1 everywhere (:132-135).
 * ------------------------------------------------------------------------------------ */
int av2x_comm_mask(const float* psm, int32_t n, int32_t h, int32_t w, int32_t ctot, int32_t c,
                   const float* gauss_w, const float* gauss_b, int32_t k, float threshold,
                   const int32_t* sample_of_agent, const int32_t* is_ego,
                   float* conf, float* smooth, float* mask, int32_t* count, av2x_stream_t stream);

/* x[n,h,w,c] *= mask[n,h,w]  (where2comm_fuse.py:237) — in place, NHWC. */
int av2x_apply_mask(float* x, const float* mask, int32_t n, int32_t hw, int32_t c,
                    av2x_stream_t stream);

/* out[n,ho,wo] = F.interpolate(in[n,hi,wi], size=(ho,wo), mode="bilinear", align_corners=False): the communication mask brought to
 * the first block's resolution when the backbone's deblocks change it (where2comm_fuse.py:229-235). */
int av2x_mask_resize_bilinear(const float* in, int32_t n, int32_t hi, int32_t wi, int32_t ho, int32_t wo, float* out,
                              av2x_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Per-pixel scaled-dot-product attention across the agents of one sample, ego row only.
 * Replaces AttentionFusion.forward + ScaledDotProductAttention.forward
 * (where2comm_fuse.py:152-164, :41-45):  out[p,:] = sum_j softmax_j(x0.xj/sqrt(C)) xj.
 *   agents: HOST array of n_agents DEVICE pointers, each to an (hw, c) NHWC map (ego first);
 *   out (hw, c).  c % 4 == 0 (c = 64 / 128 / 256 take the register-resident kernels), 1 <= n_agents <= 32.
 * ------------------------------------------------------------------------------------ */
int av2x_pixel_attn_fuse(const float* const* agents, int32_t n_agents, int32_t hw, int32_t c,
                         float* out, av2x_stream_t stream);

/* com = mean_b count[b] / (agents_per_sample[b] * hw): the `communication_rates` scalar of where2comm_fuse.py:137,147
 * from the popcounts av2x_comm_mask accumulated; count (n_samples,) i32, agents_per_sample (n_samples,) f32, com (1,) f32. */
/* Training branch of Communication.forward (where2comm_fuse.py:104-121): mask = the k_of_agent[a] cells of agent a with
 * the largest SMOOTHED confidence (`smooth` as written by av2x_comm_mask), count[sample] += that many (before the ego
 * override), ego agents all ones.  K = int(H * W * random.uniform(0, 1)) is drawn by the caller (one draw per sample). */
int av2x_comm_mask_topk(const float* smooth, int32_t n, int32_t hw, const int32_t* k_of_agent,
                        const int32_t* sample_of_agent, const int32_t* is_ego, float* mask, int32_t* count,
                        av2x_stream_t stream);
int av2x_comm_rate(const int32_t* count, const float* agents_per_sample, int32_t n_samples, int32_t hw, float* com,
                   av2x_stream_t stream);

/* count_nonzero over a dense fp32 buffer (airv2x_where2com.py:122); result (1,) u64 += */
int av2x_count_nonzero(const float* x, uint64_t n_elems, unsigned long long* result,
                       av2x_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Point preparation ahead of the voxelizer.  Replaces, for one agent's cloud, the numpy / torch sequence of
 * intermediate_fusion_dataset.py:591-603: shuffle_points (pcd_utils.py:193-197) -> mask_ego_points (:168-190) ->
 * project_points_by_matrix_torch (box_utils.py:1038-1067, `proj_first: true`) -> mask_points_by_range (:136-165).
 *   points (n_points,4) f32 device; perm (n_points,) i32 device or NULL: the shuffle permutation (the
 *   reference draws it from numpy's global RNG; results downstream only depend on it through over-full pillars);
 *   transform16: HOST row-major 4x4 (NULL = no projection); x' = T[j][0]*x (+fma) T[j][1]*y (+fma) T[j][2]*z (+fma)
 *   T[j][3], the fp32 evaluation order of torch's einsum on the reference's CPU path (bit-exact);
 *   range6: HOST {xmin,ymin,zmin,xmax,ymax,zmax}, strict inequalities; mask_ego != 0: drop the closed box
 *   x in [-1.95, 2.95], y in [-1.1, 1.1] BEFORE the projection;
 *   workspace: av2x_prepare_points_workspace_bytes(n_points) bytes; out (n_points,4) f32: the surviving points,
 *   order preserved, compacted to the front; count (1,) i32 device: how many.
 * ------------------------------------------------------------------------------------ */
uint64_t av2x_prepare_points_workspace_bytes(int32_t n_points);
int av2x_prepare_points(const float* points, const int32_t* perm, int32_t n_points, const float* transform16,
                        const float* range6, int32_t mask_ego, void* workspace, float* out, int32_t* count,
                        av2x_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Pillar voxelizer (points -> voxels), deterministic.  Replaces the spconv call behind
 * SpVoxelPreprocessor.preprocess (data_utils/pre_processor/sp_voxel_preprocessor.py:93-110,
 * third-party `spconv.utils.Point2VoxelCPU3d`): c = floor((p - range_min)/voxel) in fp32, voxels
 * numbered by first appearance (at most max_voxels), first max_points points per voxel kept in
 * input order, zero padded.
 *   points (n_points,4) f32 device; range6 {xmin,ymin,zmin,xmax,ymax,zmax} and voxel3 are HOST
 *   arrays; workspace: av2x_voxelize_workspace_bytes() bytes of device scratch;
 *   outputs are CAPACITY sized, cap = min(n_points, max_voxels):
 *     voxels (cap,max_points,4) f32, coords (cap,3) i32 in z,y,x order, num_points (cap,) i32,
 *     n_voxels (1,) i32 device scalar = M, the number of valid leading rows.
 * ------------------------------------------------------------------------------------ */
uint64_t av2x_voxelize_workspace_bytes(int32_t n_points, int32_t nx, int32_t ny, int32_t nz);
int av2x_voxelize(const float* points, int32_t n_points, const float* range6, const float* voxel3,
                  int32_t max_points, int32_t max_voxels, void* workspace, float* voxels,
                  int32_t* coords, int32_t* num_points, int32_t* n_voxels, av2x_stream_t stream);
/* av2x_prepare_points + av2x_voxelize in ONE pass: the point at position i is prepare(points[perm[i]]) (ego-box mask,
 * projection by transform16, strict crop to crop_range6 -- same argument meaning as av2x_prepare_points); dropped
 * points get no cell, so voxel order / in-voxel order are those of the compacted cloud and the voxels hold the
 * PROJECTED coordinates.  No intermediate cloud, no host read-back between the two stages. */
int av2x_prepare_voxelize(const float* points, const int32_t* perm, int32_t n_points, const float* transform16,
                          const float* crop_range6, int32_t mask_ego, const float* grid_range6, const float* voxel3,
                          int32_t max_points, int32_t max_voxels, void* workspace, float* voxels, int32_t* coords,
                          int32_t* num_points, int32_t* n_voxels, av2x_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Detection post-processing on the device.  Replaces VoxelPostprocessor.post_process_airv2x
 * (data_utils/post_processor/voxel_postprocessor.py:666-839: sigmoid(obj) > obj_threshold,
 * class argmax over classes 1..C-1 of psm viewed (C,A,H,W), delta_to_boxes3d :585-634,
 * boxes_to_corners_3d / project_box3d / remove_large_pred_bbx / remove_bbx_abnormal_z
 * utils/box_utils.py:195-258,332-366,981-1035) and the host NMS loop box_utils.nms_rotated
 * :823-868 (shapely polygon IoU -> fp64 convex-quad clipping; top-`top` by score, greedy
 * removal of iou > nms_threshold) plus the final range mask :399-430.
 *   psm (1,A*C,H,W), rm (1,A*7,H,W), obj (1,A,H,W) f32 NCHW (the heads' output layout);
 *   anchors (H*W*A,7) f32 [x,y,z,h,w,l,yaw]; transform16 (4x4 row-major) and range6 are HOST;
 *   workspace: av2x_postprocess_workspace_bytes(h,w,a,top) bytes;
 *   outputs CAPACITY `top`: out_corners (top,8,3), out_scores (top,), out_labels (top,) i32,
 *   out_boxes (top,7), out_index (top,) i32 = anchor index (h*W+w)*A+a of each result;
 *   counts (5,) i32 device: {obj candidates, after size/z filters, NMS input, NMS picks, final}.
 * Results are in NMS pick order (descending score), exactly the reference's output order.
 * ------------------------------------------------------------------------------------ */
uint64_t av2x_postprocess_workspace_bytes(int32_t h, int32_t w, int32_t a, int32_t top);
int av2x_postprocess(const float* psm, const float* rm, const float* obj, const float* anchors,
                     int32_t h, int32_t w, int32_t a, int32_t c, const float* transform16,
                     const float* range6, float obj_threshold, float nms_threshold,
                     int32_t order_hwl, int32_t top, void* workspace, float* out_corners,
                     float* out_scores, int32_t* out_labels, float* out_boxes, int32_t* out_index,
                     int32_t* counts, av2x_stream_t stream);
/* Same, with the 4x4 ego transform (data_dict["ego"]["transformation_matrix"], voxel_postprocessor.py:701) read from
 * DEVICE memory (16 f32, row-major): the reference's inference flow keeps the batch on the GPU, and reading the matrix
 * on the host would drain the frame's stream every frame. */
int av2x_postprocess_devt(const float* psm, const float* rm, const float* obj, const float* anchors,
                          int32_t h, int32_t w, int32_t a, int32_t c, const float* transform16_dev,
                          const float* range6, float obj_threshold, float nms_threshold,
                          int32_t order_hwl, int32_t top, void* workspace, float* out_corners,
                          float* out_scores, int32_t* out_labels, float* out_boxes, int32_t* out_index,
                          int32_t* counts, av2x_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Backward of the BEV convolutions (first slice; the reference trains through torch autograd,
 * tools/train.py:220-247).  For y = act(scale * conv(x, w) + shift):
 *   av2x_act_backward   dz = dy * act'(y) * scale[c]   (act 0 identity / 1 ReLU on the stored OUTPUT y; scale may be NULL)
 *   av2x_channel_sum    d shift[c] = sum over pixels of dy * act'(y)    (rows x c, two-stage deterministic sum)
 *   av2x_conv2d_wgrad   dw (cout, cin, ks, ks) = correlation of x with dz (desc as for the forward launch: out_ctot /
 *                       out_coff describe dz's channel stride / offset); fp32 MFMA, pixel axis chunked, partial slabs
 *                       in `workspace` (av2x_conv2d_wgrad_workspace_bytes; 16-byte aligned) summed in a fixed order -- bit-reproducible
 *   data gradient       av2x_conv2d of dz with the 180-degree-rotated, channel-transposed weights (stride 2: on the
 *                       zero-upsampled dz) -- opencood_iface/autograd.py
 * ------------------------------------------------------------------------------------ */
uint64_t av2x_conv2d_wgrad_workspace_bytes(const av2x_conv_desc* d);
/* the kernels' weight packing (ks*ks, cin/4, coutp = cout rounded up to 32, 4) of an nn.Conv2d parameter, on the device (the weights change
 * every optimiser step): flipped 0: w is (cout, cin, ks, ks); flipped 1: w is (cin, cout, ks, ks) and the taps are rotated by 180 degrees --
 * the weights of the data gradient (autograd of F.conv2d, base_bev_backbone.py:41-60).  wp: 16-byte aligned, padded columns are zeroed. */
int av2x_pack_conv_weight(const float* w, int32_t cout, int32_t cin, int32_t ks, int32_t flipped, float* wp, av2x_stream_t stream);
/* the split-3 bf16 planes of a packed weight, (3, taps, cin/8, coutp, 8) = hi | mid | lo with hi + mid + lo = w to 2^-24: what the
 * pipelined split-3 GEMM (tile flag 0x1400) reads; from the fp32 packing (taps, cin/4, coutp, 4) in one launch.  cin % 8 == 0. */
int av2x_split3_koct(const float* packed, int32_t taps, int32_t cin, int32_t coutp, void* planes, av2x_stream_t stream);
int av2x_conv2d_wgrad(const av2x_conv_desc* d, const float* x, const float* dz, void* workspace, float* dw,
                      av2x_stream_t stream);
int av2x_act_backward(const float* y, const float* dy, const float* scale, int64_t rows, int32_t c, int32_t act,
                      float* dz, av2x_stream_t stream);
uint64_t av2x_channel_sum_workspace_bytes(int64_t rows, int32_t c);
int av2x_channel_sum(const float* x, int64_t rows, int32_t c, void* workspace, float* out, av2x_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Train-mode forward / backward pieces of the Where2Comm path (the reference trains through torch autograd,
 * tools/train.py:220-247).  Deterministic: fixed-order fp64 partial sums in `workspace`, no floating-point atomics.
 *   av2x_bn_stats            batch mean / BIASED variance per channel of an NHWC map z (rows x c): nn.BatchNorm2d / 1d in
 *                            train mode (base_bev_backbone.py:52,65,83; airv2x_pillar_vfe.py:21).  workspace:
 *                            av2x_bn_workspace_bytes(rows, c)
 *   av2x_affine_act          y = act(z * scale[c] + shift[c]) (act 0 identity / 1 ReLU; scale / shift may be NULL); y may be z
 *   av2x_bn_backward         for y = act(z * scale + shift), scale = gamma * rstd, shift = beta - mean * scale:
 *                            dgamma[c] = sum g xhat, dbeta[c] = sum g, dz = scale (g - dbeta / rows - xhat dgamma / rows)
 *                            with g = dy * act'(.), xhat = (z - mean) rstd.  dz may alias dy
 *   av2x_pixel_attn_backward gradient of av2x_pixel_attn_fuse (AttentionFusion, where2comm_fuse.py:152-164) w.r.t. every
 *                            agent map: agents / dagents are HOST arrays of n_agents DEVICE pointers to (hw, c) maps
 *   av2x_pillar_moments      moments[0..9] = sum of the augmented 10-vectors over all 32 rows of all pillars (padded rows
 *                            are zero), moments[10..109] = sum of their outer products (fp64): BatchNorm1d's batch
 *                            statistics of PFNLayer (airv2x_pillar_vfe.py:27-49) follow as mean = W S / N,
 *                            E[lin^2] = diag(W F W^T) / N with N = 32 n_pillars
 *   av2x_pillar_vfe_backward PointPillarScatter + max-over-points + ReLU backward: gathers dcanvas (n, ny, nx, 64) at every
 *                            pillar, routes it to the arg-max row of every channel; out[c][0..9] = sum g feat,
 *                            out[c][10] = sum g (d beta), out[c][11] = sum g xhat (d gamma), fp64 (64 x 12)
 *                            workspace for both pillar calls: av2x_pillar_train_workspace_bytes(n_pillars)
 * ------------------------------------------------------------------------------------ */
uint64_t av2x_bn_workspace_bytes(int64_t rows, int32_t c);
int av2x_bn_stats(const float* z, int64_t rows, int32_t c, void* workspace, float* mean, float* var, av2x_stream_t stream);
/* rstd = 1 / sqrt(var + eps), scale = gamma * rstd, shift = beta - mean * scale, and -- running_mean / running_var non-NULL --
 * nn.BatchNorm's train-mode update applied `times` times with these batch statistics (momentum; unbiased variance
 * var * count / (count - 1)); num_batches_tracked (int64, may be NULL) += times. */
int av2x_bn_finalize(const float* mean, const float* var, const float* gamma, const float* beta, int32_t c, float eps,
                     int64_t count, float momentum, int32_t times, float* rstd, float* scale, float* shift,
                     float* running_mean, float* running_var, int64_t* num_batches_tracked, av2x_stream_t stream);
int av2x_affine_act(const float* z, int64_t rows, int32_t c, const float* scale, const float* shift, int32_t act, float* y,
                    av2x_stream_t stream);
/* av2x_bn_stats + av2x_bn_finalize + av2x_affine_act in one call; stats5 (5, c) receives mean, biased var, rstd, scale, shift. */
int av2x_bn_train_forward(const float* z, int64_t rows, int32_t c, const float* gamma, const float* beta, float eps,
                          float momentum, int32_t times, int32_t act, void* workspace, float* stats5, float* y,
                          float* running_mean, float* running_var, int64_t* num_batches_tracked, av2x_stream_t stream);
int av2x_bn_backward(const float* dy, const float* z, int64_t rows, int32_t c, const float* mean, const float* rstd,
                     const float* scale, const float* shift, int32_t act, void* workspace, float* dgamma, float* dbeta,
                     float* dz, av2x_stream_t stream);
int av2x_pixel_attn_backward(const float* const* agents, int32_t n_agents, int32_t hw, int32_t c, const float* dout,
                             float* const* dagents, av2x_stream_t stream);
uint64_t av2x_pillar_train_workspace_bytes(int32_t n_pillars);
int av2x_pillar_moments(const float* voxel_features, const int32_t* voxel_coords, const int32_t* voxel_num_points,
                        int32_t n_pillars, const float* geom, void* workspace, double* moments, av2x_stream_t stream);
int av2x_pillar_vfe_backward(const float* voxel_features, const int32_t* voxel_coords, const int32_t* voxel_num_points,
                             int32_t n_pillars, const float* pfn_w, const float* bn_scale, const float* bn_shift,
                             const float* mean, const float* rstd, const float* geom, const float* dcanvas,
                             int32_t canvas_agent0, const int32_t* slot_map, int32_t n_agents_type, int32_t ny, int32_t nx,
                             void* workspace, double* out, av2x_stream_t stream);
/* The stand-alone halves (PillarVFE / PointPillarScatter sub-modules in train mode): av2x_pillar_vfe_backward with the gradient
 * of the (n_pillars, 64) pillar features instead of the canvas, and the scatter's backward,
 * dfeatures[pil, :] = dcanvas[agent, y, x, :] (point_pillar_scatter.py:59-68). */
int av2x_pillar_vfe_backward_rows(const float* voxel_features, const int32_t* voxel_coords, const int32_t* voxel_num_points,
                                  int32_t n_pillars, const float* pfn_w, const float* bn_scale, const float* bn_shift,
                                  const float* mean, const float* rstd, const float* geom, const float* dfeatures,
                                  void* workspace, double* out, av2x_stream_t stream);
int av2x_pillar_gather(const float* dcanvas, const int32_t* voxel_coords, int32_t n_pillars, int32_t channels, float* dfeatures,
                       int32_t n_agents, int32_t ny, int32_t nx, av2x_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Training labels: VoxelPostprocessor.generate_label_airv2x (voxel_postprocessor.py:217-354) with
 * bbox_overlaps (utils/box_overlaps.pyx:17-57) -- the anchor <-> ground-truth assignment, without the IoU matrix.
 *   anchor_standup (n_anchors,4) / gt_standup (n_gt,4) f32: [xmin,ymin,xmax,ymax] of corner2d_to_standup_box (:266-270);
 *   anchors7 (n_anchors,7) / gt7 (n_gt,7) f64 in the 'hwl' order [x,y,z,h,w,l,yaw]; class_ids (n_gt,) i32;
 *   workspace: 8 * max(n_gt, 1) bytes;
 *   pos_equal_one / neg_equal_one (n_anchors,) f64, targets (n_anchors,7) f64, cls_labels (n_anchors,) i64 -- the flat
 *   (H, W, A[, 7]) arrays of the reference's label_dict.  n_gt = 0: no positives, every anchor negative (the sum over an
 *   empty axis of :290-294 equals iou.shape[1] = 0 everywhere).
 * ------------------------------------------------------------------------------------ */
int av2x_generate_label(const float* anchor_standup, const float* gt_standup, const double* anchors7, const double* gt7,
                        const int32_t* class_ids, int32_t n_anchors, int32_t n_gt, float pos_threshold,
                        float neg_threshold, void* workspace, double* pos_equal_one, double* neg_equal_one,
                        double* targets, int64_t* cls_labels, av2x_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Camera lift-splat: LiftSplatShootEncoder.get_geometry + voxel_pooling
 * (models/common_modules/airv2x_encoder.py:133-167, 208-275; QuickCumsum utils/camera_utils.py:341-365) fused.
 *   x           (b*n_cams*pts_per_cam, c) f32: the lifted features of every frustum point, point-major (b, n, d, fh, fw)
 *               -- the reference's x.permute(0,1,3,4,5,2) (:185); may be NULL together with out (geometry only);
 *   frustum     (pts_per_cam, 3) f32 = create_frustum() (:94-131) flattened (d, fh, fw);
 *   cam_params  (b*n_cams, 24) f32 per camera: inverse(post_rots) (9, row-major), post_trans (3),
 *               rots @ inverse(intrins) (9), trans (3) -- the per-frame host matrices of get_geometry (:147-166);
 *   lo3 = bx - dx/2, dx3, nx3: the voxel grid of gen_dx_bx (utils/camera_utils.py:238-245);
 *   workspace   av2x_lss_pool_workspace_bytes(b, nx, ny, nz, c) bytes (64-bit fixed-point accumulators);
 *   out         (b, ny, nx, nz*c) f32 NHWC, channel z*c + k = the reference's torch.cat(final.unbind(2), 1) (:272);
 *   geom_out    optional (b*n_cams*pts_per_cam, 3) f32: the ego-frame points of get_geometry (tests).
 * Sums are accumulated as 2^-32 fixed point with integer atomics: bit-reproducible, and each voxel's sum is exact to
 * 2^-32 per addend (the reference's cumsum-difference carries the rounding of a running sum over ALL points).
 * ------------------------------------------------------------------------------------ */
uint64_t av2x_lss_pool_workspace_bytes(int32_t b, int32_t nx, int32_t ny, int32_t nz, int32_t c);
int av2x_lss_voxel_pool(const float* x, const float* frustum, const float* cam_params, int32_t b, int32_t n_cams,
                        int32_t pts_per_cam, int32_t c, const float* lo3, const float* dx3, const int32_t* nx3,
                        void* workspace, float* out, float* geom_out, av2x_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Training of the transformer-style fusion heads (Airv2xCoBEVT.train()): backward kernels of
 * models/cobevt_modules/swap_fusion_modules.py:78-195 and base_transformer.py:6-38 (the reference differentiates them with
 * torch autograd, tools/train.py:220-247).  Linear layers differentiate through av2x_conv2d / av2x_conv2d_wgrad (1x1 convolutions).
 *
 * av2x_layernorm_backward  dx (n_tokens, c) of nn.LayerNorm (statistics recomputed from x); `partial`: (2, rows, c) floats with
 *                          rows = av2x_layernorm_backward_rows(n_tokens): per-workgroup partial sums of dgamma (first `rows` rows)
 *                          and dbeta (the next `rows`), to be reduced in a fixed order with av2x_channel_sum.  c in {256, 512}.
 * av2x_gelu                dy == NULL: out = gelu(z) (exact, nn.GELU()); else out = dy * gelu'(z).  n % 4 == 0.
 * av2x_scale_broadcast     dx[l] = scale * dy for l < n_agents: backward of the mean over the agent axis (:270).
 * av2x_dropout             y = x * mask * scale, mask = n bytes of 0 / 1 drawn by the caller (nn.Dropout forward and backward).
 * av2x_fax_attention_backward   Attention.forward :78-127 differentiated for all windows of one sample: qkv / bias_table / window
 *                          layout as av2x_fax_attention, `out` = that call's output, dout its gradient; writes dqkv (same layout as
 *                          qkv; zero for the k | v of padded agents) and dbias_table (tab_n, heads) -- summed over windows as 2^-32
 *                          fixed point in `workspace` (av2x_fax_attention_backward_workspace_bytes), so bit-reproducible.
 *                          L * window^2 <= 128 tokens per window, dim_head 32.
 * ------------------------------------------------------------------------------------ */
int32_t av2x_layernorm_backward_rows(int64_t n_tokens);
int av2x_layernorm_backward(const float* x, const float* gamma, const float* dy, int64_t n_tokens, int32_t c, float eps,
                            float* dx, float* partial, av2x_stream_t stream);
int av2x_gelu(const float* z, const float* dy, float* out, uint64_t n, av2x_stream_t stream);
int av2x_scale_broadcast(const float* dy, float* dx, int32_t n_agents, uint64_t elems_per_agent, float scale, av2x_stream_t stream);
int av2x_dropout(const float* x, const uint8_t* mask, float* y, uint64_t n, float scale, av2x_stream_t stream);
/* nn.Dropout with the draw in the kernel: y = x * keep / (1 - p) (+ residual, may be NULL: the skip connection a block adds after its
 * dropout), keep regenerated from `seed` (Philox4x32-10, counter = index / 4) -- the backward is the same call on dy with the forward's seed
 * and no residual; no mask tensor.  n % 4 == 0, 0 <= p < 1. */
int av2x_dropout_seeded(const float* x, const float* residual, float* y, uint64_t n, float p, uint64_t seed, av2x_stream_t stream);
uint64_t av2x_fax_attention_backward_workspace_bytes(int32_t n_agents_padded, int32_t window, int32_t heads);
int av2x_fax_attention_backward(const float* qkv, const float* bias_table, const float* out, const float* dout,
                                int32_t n_agents_padded, int32_t n_valid, int32_t h, int32_t w, int32_t window, int32_t heads,
                                int32_t dim_head, int32_t grid_partition, float* dqkv, float* dbias_table, void* workspace,
                                av2x_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Training of the V2X-ViT fusion (Airv2xV2XVit.train()): backward kernels of models/v2xvit_modules/hmsa.py:133-151,
 * mswin.py:52-96, split_attn.py:48-61 and of warp_affine (torch_transformation_utils.py:337-381).  Layouts as the forward entry points.
 *
 * av2x_hgt_attention_backward      dproj (n, hw, 1280) of av2x_hgt_attention: the gradient of the FOLDED projections
 *                                  [q'(->t0) | q'(->t1) | k | v'(t0<-) | v'(t1<-)]; the relation matrices receive theirs through the fold,
 *                                  which the host side expresses in differentiable tensor algebra on the (small) weights.
 * av2x_window_attention_backward   dqkv (same (n*h*w, ctot) buffer layout, block at `coff`) and dpos (2w-1, 2w-1) of
 *                                  av2x_window_attention; `out` = that call's output.  The pos_embedding gradient is summed as 2^-32 fixed
 *                                  point (bit-reproducible).  workspace: av2x_window_attention_backward_workspace_bytes.
 * av2x_split_attn_sums             da (n, 3, c) = sum over pixels of dout * s_r: the gradient of the radix weights of SplitAttn.
 * av2x_split_attn_backward         ds_r = weights[a][r][c] * dout + dgap[a][c] / hw  (weights = the radix softmax, (n, 3, c)).
 * av2x_warp_affine_backward        dsrc of av2x_warp_affine (align_corners = True): the adjoint of the bilinear sampling, scattered with
 *                                  2^-32 fixed-point atomics into `workspace` (8 bytes per element) -- bit-reproducible.
 * ------------------------------------------------------------------------------------ */
int av2x_hgt_attention_backward(const float* proj, const float* mask, const int32_t* types_host, const float* dout, float* dproj,
                                int32_t n, int32_t hw, int32_t heads, int32_t dim_head, av2x_stream_t stream);
uint64_t av2x_window_attention_backward_workspace_bytes(int32_t n, int32_t h, int32_t w, int32_t heads, int32_t window);
int av2x_window_attention_backward(const float* qkv, int32_t ctot, int32_t coff, const float* pos_embedding, const float* out,
                                   const float* dout, float* dqkv, float* dpos, void* workspace, int32_t n, int32_t h, int32_t w,
                                   int32_t heads, int32_t dim_head, int32_t window, av2x_stream_t stream);
uint64_t av2x_split_attn_backward_workspace_bytes(int32_t n, int32_t c);
int av2x_split_attn_sums(const float* s0, const float* s1, const float* s2, const float* dout, float* da, float* workspace,
                         int32_t n, int32_t hw, int32_t c, av2x_stream_t stream);
int av2x_split_attn_backward(const float* dout, const float* weights, const float* dgap, float* ds0, float* ds1, float* ds2,
                             int32_t n, int32_t hw, int32_t c, av2x_stream_t stream);
uint64_t av2x_warp_affine_backward_workspace_bytes(int32_t n, int32_t h, int32_t w, int32_t c);
int av2x_warp_affine_backward(const float* ddst, const float* theta, float* dsrc, void* workspace, int32_t n, int32_t h, int32_t w,
                              int32_t c, av2x_stream_t stream);
/* the same for av2x_warp_affine_simple (align_corners = False; the warp of the When2com / V2VNet / OPV2V-style Where2comm fusions) */
int av2x_warp_affine_simple_backward(const float* ddst, const float* theta, float* dsrc, void* workspace, int32_t n, int32_t h, int32_t w,
                                     int32_t c, av2x_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Camera encoder (BASELINE configs[4]): the non-GEMM kernels of CamEncode / BevEncode
 * (models/sub_modules/lss_submodule.py:22-189, 312-350) and of the efficientnet_pytorch trunk CamEncode walks (:118-146).
 * All maps NHWC fp32; the pointwise / 3x3 / 7x7 convolutions run on av2x_conv2d.
 *
 * av2x_cam_stem        EfficientNet stem (trunk._conv_stem + _bn0 + swish, :124-126): imgs (n, planes >= 3, h, w) NCHW as the dataset
 *                      stacks them (intermediate_fusion_dataset.py:561,573; plane 3 = depth, not read here); weight (27, 32) with row
 *                      (kh*3+kw)*3+ci; 3x3 / stride 2, zero padding pad_t / pad_l before and whatever (ho, wo) needs after (the package's
 *                      static "same" padding: 0 before, 1 after); out (n, ho, wo, 32).
 * av2x_dwconv2d        depthwise ks x ks (3 | 5) conv + affine (folded BN) + activation (0 none, 1 ReLU, 6 swish): weight (ks*ks, c);
 *                      asymmetric zero padding as above.  MBConv `_depthwise_conv` + `_bn1` + swish.
 * av2x_squeeze_excite  MBConv squeeze-excite on x (n, hw, c) in place: gate = sigmoid(W_e swish(W_r mean_hw(x) + b_r) + b_e), x *= gate
 *                      (apply = 0: only the gate, left at workspace + n * av2x_se_slabs(hw) * c floats).  w_reduce (c_se, c),
 *                      w_expand TRANSPOSED to (c_se, c) (coalesced reads).  Sums run in a fixed order (bit-reproducible).  workspace: av2x_squeeze_excite_workspace_bytes.
 * av2x_resize_bilinear nn.Upsample(bilinear, align_corners=True) of in (n, h, w, c of in_ctot from in_coff) to (h2, w2), placed at
 *                      (pad_t, pad_l) of the (hout, wout) output with zeros around it (F.pad in Up.forward :41-45), written to the
 *                      channel slice [out_coff, out_coff + c) of out (n, hout, wout, out_ctot) (torch.cat :46).  h2 == h copies.
 * av2x_softmax_channels CamEncode.get_depth_dist (:89-92): softmax over the first d of `stride` channels of every row.
 * av2x_lss_lift_pool   CamEncode's depth (x) feature outer product (:176-186) + LiftSplatShootEncoder.get_geometry / voxel_pooling
 *                      (airv2x_encoder.py:133-275) without the (B, N, D, fH, fW, C) volume: feat (b*n_cams, fh, fw, c) image features;
 *                      EITHER prob (b*n_cams, fh, fw, nbins) predicted depth distribution, OR imgs (b*n_cams, planes, img_h, img_w) whose
 *                      plane 3 is the ground-truth depth: binned as utils/camera_utils.py:247-298 (depth3 = {d_min, d_max, bin size},
 *                      depth_mode 0 UD / 1 LID, target = CamEncode.training: clamp instead of masking), sampled at the centre of every
 *                      downsample x downsample cell (:104-108).  frustum / cam_params / lo3 / dx3 / nx3 / workspace / out as
 *                      av2x_lss_voxel_pool.
 * ------------------------------------------------------------------------------------ */
/* Airv2xBase.fuse_bev (models/common_modules/airv2x_base_model.py:167-177) for the two modality maps of an agent type:
 * out = (a + b) / 2 elementwise (n floats, n % 4 == 0); b == NULL copies a. */
int av2x_mean2(const float* a, const float* b, float* out, uint64_t n, av2x_stream_t stream);
int av2x_cam_stem(const float* imgs, int32_t n, int32_t planes, int32_t h, int32_t w, const float* weight, const float* scale,
                  const float* shift, int32_t pad_t, int32_t pad_l, int32_t ho, int32_t wo, float* out, av2x_stream_t stream);
int av2x_dwconv2d(const float* x, int32_t n, int32_t h, int32_t w, int32_t c, const float* weight, const float* scale,
                  const float* shift, int32_t ks, int32_t stride, int32_t pad_t, int32_t pad_l, int32_t ho, int32_t wo,
                  int32_t act, float* out, av2x_stream_t stream);
int32_t av2x_se_slabs(int32_t hw);
uint64_t av2x_squeeze_excite_workspace_bytes(int32_t n, int32_t hw, int32_t c);
int av2x_squeeze_excite(float* x, int32_t n, int32_t hw, int32_t c, const float* w_reduce, const float* b_reduce, int32_t c_se,
                        const float* w_expand, const float* b_expand, float* workspace, int32_t apply, av2x_stream_t stream);
int av2x_resize_bilinear(const float* in, int32_t n, int32_t h, int32_t w, int32_t c, int32_t in_ctot, int32_t in_coff,
                         int32_t h2, int32_t w2, int32_t pad_t, int32_t pad_l, int32_t hout, int32_t wout, float* out,
                         int32_t out_ctot, int32_t out_coff, av2x_stream_t stream);
/* av2x_maxpool2d: nn.MaxPool2d(ks, stride, pad) on an NHWC map (the 3x3 / 2 / 1 pool of CamEncode_Resnet101's stem, lss_submodule.py:262-266);
 * padding cells are ignored as torch's -inf padding is. */
int av2x_maxpool2d(const float* x, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ks, int32_t stride, int32_t pad, int32_t ho, int32_t wo,
                   float* out, av2x_stream_t stream);
int av2x_softmax_channels(const float* x, int64_t rows, int32_t d, int32_t stride, float* out, av2x_stream_t stream);
int av2x_lss_lift_pool(const float* feat, const float* prob, const float* imgs, int32_t planes, int32_t img_h, int32_t img_w,
                       int32_t downsample, const float* depth3, int32_t nbins, int32_t depth_mode, int32_t target,
                       const float* frustum, const float* cam_params, int32_t b, int32_t n_cams, int32_t fh, int32_t fw,
                       int32_t c, const float* lo3, const float* dx3, const int32_t* nx3, void* workspace, float* out,
                       av2x_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Training of the camera branch (what torch autograd gives the reference for the non-convolution pieces of CamEncode / BevEncode,
 * lss_submodule.py:22-189, 312-350, the efficientnet_pytorch MBConv blocks, and voxel_pooling airv2x_encoder.py:208-275).
 *   av2x_unary_forward / _backward      y = act(x), dx = dy * act'(x) from the PRE-activation x; act 3 sigmoid, 6 swish (x * sigmoid(x))
 *   av2x_add_act                        y = a + b, or relu(a + b) (BasicBlock's ReLU after the residual add; its gradient: av2x_act_backward on y)
 *   av2x_gap                            out (n, c) = scale * sum_p x (n, hw, c) [* w (n, hw, c)]: the squeeze (scale 1 / hw) and -- w = dy -- the
 *                                       gate gradient of the excite; workspace: av2x_gap_workspace_bytes(n, hw, c); fixed summation order
 *   av2x_channel_broadcast              out (n, hw, c) = g (n, c) * scale [* y (n, hw, c)]: the excite (y = x), its data gradient (y = dy) and the
 *                                       adjoint of the squeeze (y NULL, scale 1 / hw)
 *   av2x_resize_bilinear_backward       adjoint of the (h, w) -> (h2, w2) align_corners = True enlargement of av2x_resize_bilinear, gathered per
 *                                       source pixel in a fixed order (no atomics; `workspace` unused, _workspace_bytes returns 0)
 *   av2x_dwconv2d_wgrad                 dw (ks*ks, c) of av2x_dwconv2d (ks 3 | 5, stride 1 | 2, pad_t = pad_l = pad): per-slab partial sums in
 *                                       `workspace` (av2x_dwconv2d_wgrad_workspace_bytes), summed in slab order -- bit-reproducible
 *   av2x_lss_lift_pool_prob_backward    adjoint of av2x_lss_lift_pool in its predicted-depth form: dfeat (b * n_cams, fh, fw, c) = sum over the bins of
 *                                       prob * dout[voxel], dprob (b * n_cams, fh, fw, nbins) = <feat, dout[voxel]> (gathers; 0 outside the grid)
 *   av2x_softmax_channels_backward      dlogit (rows, stride) = prob * (dprob - <prob, dprob>) for the first d channels, 0 for the padding ones
 *   av2x_lss_lift_pool_backward         adjoint of av2x_lss_lift_pool in its ground-truth-depth form: dfeat (b * n_cams, fh, fw, c) gathered from
 *                                       dout (b, nz, ny, nx, c) at the voxel each feature pixel was lifted to (zeros where it left the grid)
 * ------------------------------------------------------------------------------------ */
int av2x_unary_forward(const float* x, uint64_t n_elems, int32_t act, float* y, av2x_stream_t stream);
int av2x_unary_backward(const float* x, const float* dy, uint64_t n_elems, int32_t act, float* dx, av2x_stream_t stream);
int av2x_add_act(const float* a, const float* b, uint64_t n_elems, int32_t relu, float* y, av2x_stream_t stream);
uint64_t av2x_gap_workspace_bytes(int32_t n, int32_t hw, int32_t c);
int av2x_gap(const float* x, const float* w, int32_t n, int32_t hw, int32_t c, float scale, float* workspace, float* out, av2x_stream_t stream);
int av2x_channel_broadcast(const float* g, const float* y, int32_t n, int32_t hw, int32_t c, float scale, float* out, av2x_stream_t stream);
uint64_t av2x_resize_bilinear_backward_workspace_bytes(int32_t n, int32_t h, int32_t w, int32_t c);
int av2x_resize_bilinear_backward(const float* dy, int32_t n, int32_t h, int32_t w, int32_t c, int32_t h2, int32_t w2, void* workspace, float* dx,
                                  av2x_stream_t stream);
int av2x_maxpool2d_backward(const float* x, const float* dy, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ks, int32_t stride, int32_t pad,
                            int32_t ho, int32_t wo, float* dx, av2x_stream_t stream);   /* adjoint of av2x_maxpool2d (first maximum of a window) */
uint64_t av2x_dwconv2d_wgrad_workspace_bytes(int32_t n, int32_t ho, int32_t wo, int32_t c, int32_t ks);
int av2x_dwconv2d_wgrad(const float* x, const float* dy, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ks, int32_t stride, int32_t pad,
                        int32_t ho, int32_t wo, void* workspace, float* dw, av2x_stream_t stream);
int av2x_lss_lift_pool_prob_backward(const float* dout, const float* feat, const float* prob, int32_t nbins, const float* frustum,
                                     const float* cam_params, int32_t b, int32_t n_cams, int32_t fh, int32_t fw, int32_t c, const float* lo3,
                                     const float* dx3, const int32_t* nx3, float* dfeat, float* dprob, av2x_stream_t stream);
int av2x_softmax_channels_backward(const float* prob, const float* dprob, int64_t rows, int32_t d, int32_t stride, float* dlogit,
                                   av2x_stream_t stream);
int av2x_lss_lift_pool_backward(const float* dout, const float* imgs, int32_t planes, int32_t img_h, int32_t img_w, int32_t downsample,
                                const float* depth3, int32_t nbins, int32_t depth_mode, int32_t target, const float* frustum,
                                const float* cam_params, int32_t b, int32_t n_cams, int32_t fh, int32_t fw, int32_t c, const float* lo3,
                                const float* dx3, const int32_t* nx3, float* dfeat, av2x_stream_t stream);

/* ------------------------------------------------------------------------------------
 * V2VNet message aggregation (models/v2vnet_modules/v2v_fuse.py:137-165) for ONE receiving agent i:
 *   message_j = (msg_cnn([warp_j(x_j) | x_i]) ) * roi_mask_ij      (:150-158)
 *   agg       = mean_j / max_j message_j                              (:161-164; op 0 = "avg", 1 = "max")
 * msg_cnn is linear in its concatenated input, so the caller hands over its two halves:
 *   msg_a (n,h,w,c) = conv3x3(warped neighbours; W[:, :c]) without bias, ego_b (h,w,c) = conv3x3(x_i; W[:, c:]) + bias.
 * roi_mask_ij (:110-118) = warp_affine_simple of an all-ones map with theta (n,2,3) (align_corners = False, bilinear,
 * zero padding), i.e. the sum of the in-image bilinear weights of every output pixel: computed in the kernel.
 *   out (h,w,c).  c % 4 == 0, n <= 32.
 * ------------------------------------------------------------------------------------ */
int av2x_v2v_aggregate(const float* msg_a, const float* ego_b, const float* theta, int32_t n, int32_t h, int32_t w,
                       int32_t c, int32_t op, float* out, av2x_stream_t stream);

/* ------------------------------------------------------------------------------------
 * OPV2V-style Where2comm (models/where2comm_modules/where2comm_attn.py:355-370, the multi-scale loop; :391-398 single scale):
 *   neighbor_feature = warp_affine_simple(node_features, t_matrix[0, :N], (H, W));  x_fuse = fuse_module(neighbor_feature)
 * in ONE kernel: the warped maps are never written.  agents: HOST array of n_agents DEVICE pointers to (h,w,c) NHWC maps
 * (agent 0 = ego; it is sampled through its own matrix like the others); theta_host: HOST (n_agents,2,3) fp32, the
 * normalised affine of F.affine_grid(align_corners=False) (where2comm_attn.py:293-307 prepares it);
 * mode 0 = AttenFusion (:55-67, ScaledDotProductAttention :46-52, row 0 only), 1 = MaxFusion (:70-75).
 *   out (h,w,c).  c in {64, 128, 256}, 1 <= n_agents <= 32.
 * ------------------------------------------------------------------------------------ */
int av2x_warp_fuse(const float* const* agents, const float* theta_host, int32_t n_agents, int32_t h, int32_t w,
                   int32_t c, int32_t mode, float* out, av2x_stream_t stream);

/* MaxFusion.forward (where2comm_attn.py:70-75) on already aligned maps: out[i] = max_j agents[j][i].
 * agents: HOST array of DEVICE pointers; elems_per_agent % 4 == 0, 16-byte aligned. */
int av2x_agent_max(const float* const* agents, int32_t n_agents, uint64_t elems_per_agent, float* out,
                   av2x_stream_t stream);

/* `(batch_x[b] * communication_mask).count_nonzero()` of Communication.forward (where2comm.py:93-95) without the
 * product: result (1,) u64 += #{(p, ch): x[p, ch] != 0 and gate[p] > thr}; x (n_pixels, c) NHWC, gate (n_pixels,) = the
 * smoothed confidence av2x_comm_mask wrote, thr = `thre`. */
int av2x_count_nonzero_where(const float* x, const float* gate, float thr, uint64_t n_pixels, int32_t c,
                             unsigned long long* result, av2x_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Detection loss and its gradient with respect to the head maps.  Replaces PointPillarLossMultiClass.forward
 * (loss/point_pillar_loss_multiclass.py:96-179: focal classification :183-214, sin-difference + WeightedSmoothL1Loss(beta 1/9)
 * :279-293, :13-76, objectness BCE :163-168) and the autograd pass through it.
 *   psm (b, a*c, h, w), rm (b, a*7, h, w), obj (b, a, h, w): the head outputs, NCHW fp32;
 *   targets (b, h, w, a*7) f32, pos_equal_one (b, h, w, a) f32, class_ids (b, h, w, a) i32: the label dictionary
 *   (voxel_postprocessor.py:217-354 / av2x_generate_label);
 *   out4 (4,) f32 = total_loss (= reg + conf + obj), reg_loss (x reg_coe), conf_loss (x cls_weight), obj_loss;
 *   dpsm / drm / dobj: NULL, or buffers shaped like psm / rm / obj that receive d total_loss / d input;
 *   workspace: av2x_pp_loss_workspace_bytes(b, h, w) bytes.  Deterministic (fixed-order fp64 partial sums).
 * As written in the reference: every non-positive anchor is a negative with weight 1 (neg_equal_one is not read), the
 * classification sum is divided by the batch size twice, NaN regression targets (codes 0-5) contribute nothing.
 * a <= 8, b <= 64.
 * ------------------------------------------------------------------------------------ */
uint64_t av2x_pp_loss_workspace_bytes(int32_t b, int32_t h, int32_t w);
int av2x_pp_loss(const float* psm, const float* rm, const float* obj, const float* targets, const float* pos_equal_one,
                 const int32_t* class_ids, int32_t b, int32_t h, int32_t w, int32_t a, int32_t c, float cls_weight,
                 float reg_coe, void* workspace, float* out4, float* dpsm, float* drm, float* dobj, av2x_stream_t stream);

/* ------------------------------------------------------------------------------------
 * AP evaluation: true/false positives of one frame.  Replaces the shapely loop of caluclate_tp_fp
 * (utils/eval_utils_opv2v.py:41-97; IoU = common_utils.compute_iou :150-171 on convert_format :174-191 polygons).
 *   det_corners (n_det,8,3) / gt_corners (n_gt,8,3) f32: the first four corners' (x,y) are the BEV quad;
 *   order (n_det,) i32: detections in descending score order (np.argsort(-score), :69);
 *   iou_ws (n_det*n_gt,) f32 scratch, on return iou[d][g] of detection order[d] vs gt g;
 *   tp (n_det,) i32: 1 = matched (fp = 1 - tp), in score order; matched_gt (n_det,) i32: the ORIGINAL index
 *   of the ground-truth box removed by the match, or -1.
 * A detection matches when max IoU over the still unmatched GT boxes is >= iou_thresh; ties take the lowest
 * GT index (np.argmax).  IoU is computed in fp64 (convex clipping) and rounded to fp32 before the compare.
 * ------------------------------------------------------------------------------------ */
int av2x_eval_tp_fp(const float* det_corners, const int32_t* order, int32_t n_det, const float* gt_corners,
                    int32_t n_gt, float iou_thresh, float* iou_ws, int32_t* tp, int32_t* matched_gt,
                    av2x_stream_t stream);

/* ------------------------------------------------------------------------------------
 * CoBEVT fused-axial-attention pieces (models/cobevt_modules/swap_fusion_modules.py).
 * av2x_layernorm: nn.LayerNorm over the last dim of (n_tokens, c) (base_transformer.py:9; eps 1e-5).
 * av2x_fax_attention: Attention.forward :78-127 for ALL windows of one sample.
 *   qkv (L*h*w, 3*heads*dim_head): to_qkv output, token rows in agent-major NHWC order;
 *   bias_table ((2L-1)*(2*window-1)^2, heads): relative_position_bias_table.weight;
 *   tokens of a window are ordered (agent, w1, w2); grid_partition 0 = 'b m d (x w1) (y w2)' (:167),
 *   1 = 'b m d (w1 x) (w2 y)' (:185); keys of agents >= n_valid are masked (-inf, :103-108);
 *   out (L*h*w, heads*dim_head) in the same token order (heads merged, before to_out).
 *   Kernels: window 4 and L*16 <= 128 tokens -> one wave per (window, head) on v_mfma_f32_16x16x4_f32 with the score
 *   tile computed transposed (softmax in-lane, P / K / V stay in registers, no barrier); other windows -> generic MFMA
 *   kernel; > 128 tokens -> scalar kernel.  Test hooks in grid_partition: bit 1 forces the scalar kernel, bit 2 the
 *   generic MFMA kernel, bit 3 the workgroup-per-window transposed-score kernel.  Bit 5 (32): with more than 4 valid agents (window 4)
 *   both contractions run on the bf16 matrix cores with split-3 operands (hi + mid + lo bf16 terms of K, V, the scaled Q and the
 *   probabilities; fp32 accumulation; fax_attention_x3_kernel) -- fp32-accurate, not the bits of the fp32-input MFMA kernel; what the
 *   engines' x3 mode requests.
 * av2x_agent_mean: y = mean over the agent axis of x (n_agents, elems_per_agent)  (:270).
 * ------------------------------------------------------------------------------------ */
/* LayerNorm folded into the consuming token Linear (round 5): av2x_layernorm_stats writes (mean, rstd) per token -- the statistics half of
 * av2x_layernorm, same arithmetic -- and av2x_conv2d_ln is av2x_conv2d_res on nn.LayerNorm(in): the pipelined split-3 tiles (tile flag
 * 0x0400 | 0x1000; 1x1, stride 1, in_ctot == cin <= 1024) apply (x - mean) * rstd * gamma + beta to the operand rows while they load them.
 * Same bits as av2x_layernorm followed by av2x_conv2d_res; the normalised tensor never exists in HBM (base_transformer.py:9-20 PreNorm,
 * swap_fusion_modules.py:78-195). */
int av2x_layernorm_stats(const float* x, float* stats /* (n_tokens, 2) */, int64_t n_tokens, int32_t c, float eps, av2x_stream_t stream);
int av2x_conv2d_ln(const av2x_conv_desc* d, const float* in, const float* ln_stats, const float* ln_gamma, const float* ln_beta,
                   const float* w, const float* scale, const float* shift, const float* residual, float* out, av2x_stream_t stream);
int av2x_layernorm(const float* x, const float* gamma, const float* beta, float* y, int64_t n_tokens,
                   int32_t c, float eps, av2x_stream_t stream);
int av2x_fax_attention(const float* qkv, const float* bias_table, float* out, int32_t n_agents_padded,
                       int32_t n_valid, int32_t h, int32_t w, int32_t window, int32_t heads,
                       int32_t dim_head, int32_t grid_partition, av2x_stream_t stream);
int av2x_agent_mean(const float* x, float* y, int32_t n_agents, int64_t elems_per_agent,
                    av2x_stream_t stream);

/* LayerNorm followed by an optional ReLU (SplitAttn: act1(bn1(fc1(.))), split_attn.py:51). */
int av2x_layernorm_act(const float* x, const float* gamma, const float* beta, float* y, int64_t n_tokens,
                       int32_t c, float eps, int32_t relu, av2x_stream_t stream);

/* ------------------------------------------------------------------------------------
 * V2X-ViT fusion pieces (models/v2xvit_modules/, common_modules/torch_transformation_utils.py).
 * av2x_warp_affine: F.affine_grid + F.grid_sample(bilinear, zeros, align_corners=True) as called by
 *   warp_affine :337-381.  src/dst (n,h,w,c) NHWC; theta (n,2,3) DEVICE fp32 = the matrix handed to
 *   affine_grid (the 3x3 normalise/invert chain is host-side, opencood_iface/warp.py).
 * av2x_roi_mask: get_roi_and_cav_mask :15-53 — nearest-mode sampling of an all-ones image times the
 *   per-agent cav mask (device i32 (n,)) -> mask (n,h,w) in {0,1}.
 * av2x_add_agent_vector: x[a,:,:,:] += v[a,:]  (RTE, v2xvit_basic.py:58-80); x (n, elems_per_agent), v (n,c).
 * av2x_hgt_attention: HGTCavAttention hmsa.py:133-151 per pixel on FOLDED projections, proj (n,hw,1280) =
 *   [q'(->type0) | q'(->type1) | k | v'(type0<-) | v'(type1<-)] with relation_att / relation_msg multiplied
 *   into the Linear weights on the host; mask (n,hw): key agent visible at the pixel; types_host: HOST
 *   i32 (n,) node type (0/1) of every agent; out (n,hw,256) heads merged (before a_linears).
 * av2x_window_attention: BaseWindowAttention mswin.py:52-96 for n agent maps; the [q|k|v] block of the
 *   branch sits at column `coff` of the (n*h*w, ctot) token buffer; pos_embedding (2w-1,2w-1);
 *   out (n*h*w, heads*dim_head).  Supported (dim_head, window): (16,2) (32,4) (64,4).  The 4x4 windows run on
 *   v_mfma_f32_16x16x4_f32 (one wave per window x head); window | 0x100 selects the scalar kernel (tests).
 * av2x_split_attn_gap / _combine: SplitAttn split_attn.py:48-61 — mean over pixels of the branch sum,
 *   then radix-3 softmax of logits (n,3c) + weighted branch sum + residual.
 * ------------------------------------------------------------------------------------ */
int av2x_warp_affine(const float* src, const float* theta, float* dst, int32_t n, int32_t h, int32_t w,
                     int32_t c, av2x_stream_t stream);
/* warp_affine_simple :327-334: the same sampling with align_corners=False; theta is the caller's normalised 2x3
 * (the warp used by the OPV2V-style Where2Comm / When2Com / V2VNet fusion variants, where2comm_attn.py:293-307). */
int av2x_warp_affine_simple(const float* src, const float* theta, float* dst, int32_t n, int32_t h, int32_t w,
                            int32_t c, av2x_stream_t stream);
int av2x_roi_mask(const float* theta, const int32_t* cav_mask, float* mask, int32_t n, int32_t h, int32_t w,
                  av2x_stream_t stream);
/* av2x_warp_affine on (src + addv[agent]) -- addv (n, c), e.g. the RTE embedding of v2xvit_basic.py:58-80 -- without storing the sum:
 * the bits of av2x_add_agent_vector followed by av2x_warp_affine.  av2x_add_agent_vector_to: the add into another buffer. */
int av2x_warp_affine_add(const float* src, const float* theta, const float* addv, float* dst, int32_t n, int32_t h, int32_t w,
                         int32_t c, av2x_stream_t stream);
int av2x_add_agent_vector_to(const float* x, const float* v, float* out, int32_t n, int64_t elems_per_agent, int32_t c,
                             av2x_stream_t stream);
int av2x_add_agent_vector(float* x, const float* v, int32_t n, int64_t elems_per_agent, int32_t c,
                          av2x_stream_t stream);
int av2x_hgt_attention(const float* proj, const float* mask, const int32_t* types_host, float* out, int32_t n,
                       int32_t hw, int32_t heads, int32_t dim_head, av2x_stream_t stream);
/* same, but only the query agents 0 .. n_query-1 are computed (keys / values: all n agents).  V2XTransformer returns the
 * ego's feature only (v2xvit_basic.py: output[:, 0]), so in the LAST encoder layer n_query = 1 and the q columns of the
 * other agents' rows of `proj` are never read. */
int av2x_hgt_attention_q(const float* proj, const float* mask, const int32_t* types_host, float* out, int32_t n,
                         int32_t n_query, int32_t hw, int32_t heads, int32_t dim_head, av2x_stream_t stream);
int av2x_window_attention(const float* qkv, int32_t ctot, int32_t coff, const float* pos_embedding, float* out,
                          int32_t n, int32_t h, int32_t w, int32_t heads, int32_t dim_head, int32_t window,
                          av2x_stream_t stream);
int av2x_split_attn_gap(const float* s0, const float* s1, const float* s2, float* gap, float* scratch /* n*128*c floats */,
                        int32_t n, int32_t hw, int32_t c, av2x_stream_t stream);
int av2x_split_attn_combine(const float* s0, const float* s1, const float* s2, const float* logits,
                            const float* residual, float* out, int32_t n, int32_t hw, int32_t c,
                            av2x_stream_t stream);

/* ------------------------------------------------------------------------------------
 * V2X-ViT fusion with bf16 ACTIVATIONS in HBM (AMP mode of BASELINE configs[3]: what torch.autocast stores for the outputs
 * of nn.Linear / matmul -- reference tools/train.py:118, tools/inference.py under autocast; LayerNorm, softmax statistics,
 * the accumulators and the residual stream x stay fp32).  bf16 buffers are passed as uint16_t* (the upper 16 bits of the
 * fp32 pattern, round-to-nearest-even).  Same semantics as the fp32 entry points above unless stated.
 * av2x_layernorm_bf16: nn.LayerNorm over c = 256 of fp32 x -> bf16 y.
 * av2x_linear_bf16: out (m, cout) = act(a (m, k) . W + bias) (+ residual): a bf16 row-major, k = 256; w_packed = the bf16
 *   k-oct packing [k/8][coutp][8] (coutp % 256 == 0, zero columns beyond cout) whose columns are interleaved inside every
 *   group of 64: packed column 32 c + i holds logical column 2 i + c (opencood_iface/packing.py: interleave2_columns);
 *   bias (cout,) fp32 or NULL; out: bf16 if out_is_bf16 else fp32, slice [out_coff, out_coff + cout) of rows of out_ctot
 *   elements; residual: fp32 slice (fp32 output only) or NULL; act 0 none / 1 ReLU / 2 GELU (erf form); cout, slices: multiples of 8 (bf16 out: 16-byte row stores) / 2 (fp32 out).
 *   One workgroup per panel of 128 tokens: A is read from HBM once whatever cout is.
 * av2x_hgt_attention_bf16 / av2x_window_attention_bf16: proj / qkv and out are bf16; mask, pos_embedding fp32.
 * av2x_split_attn_gap_bf16 / _combine_bf16: the three branch maps are bf16; gap, logits, residual and out fp32.
 * ------------------------------------------------------------------------------------ */
/* dst[i] = (float)src[i] (exact widening) -- the bf16 feature-sharing message of the autocast frame (the shrink header's / compressor's
 * bf16 output: what torch.autocast stores for that Conv2d, 18.0 MB per agent at the default grid; replaces the in-process tensor hand-over
 * of airv2x_base_model.py:250-283 + fuse_utils.py:13-63 at half the bytes per xGMI link) back to the fp32 residual stream of the fusion.
 * Pointers 16-byte aligned. */
int av2x_bf16_to_f32(const uint16_t* src, float* dst, uint64_t n_elems, av2x_stream_t stream);
int av2x_layernorm_bf16(const float* x, const float* gamma, const float* beta, uint16_t* y, int64_t n_tokens, int32_t c,
                        float eps, av2x_stream_t stream);
/* x += delta (bf16: the output of the preceding Linear -- `x + fn(x)` of PreNormResidual, base_transformer.py:12, under autocast adds
 * a 16-bit Linear output to the fp32 stream), x written back, then y = LayerNorm(x) as above.  delta NULL: no add; y NULL: only the add. */
int av2x_add_layernorm_bf16(float* x, const uint16_t* delta, const float* gamma, const float* beta, uint16_t* y, int64_t n_tokens,
                            int32_t c, float eps, av2x_stream_t stream);
int av2x_linear_bf16(const uint16_t* a, const uint16_t* w_packed, const float* bias, const float* residual, void* out,
                     int64_t m, int32_t k, int32_t cout, int32_t coutp, int32_t out_is_bf16, int32_t out_ctot,
                     int32_t out_coff, int32_t res_ctot, int32_t res_coff, int32_t act, av2x_stream_t stream);
/* PreNormResidual(LayerNorm -> Linear [-> Linear]) in ONE pass over the fp32 stream (base_transformer.py:12-21 PreNormResidual,
 * :24-37 FeedForward; v2xvit_basic.py:137-159 the order of the residual adds): for the m rows of x
 *   x[r] += delta[r] for r < add_rows (delta = the bf16 output of the PREVIOUS sub-layer whose residual add is pending; written back
 *   to x if write_back_x, otherwise x is left untouched and the add stays pending for a later kernel, e.g.
 *   av2x_split_attn_combine_delta_bf16: one 16-bit read there instead of an fp32 write here),
 *   h = act(LayerNorm(x) . W + bias) rounded to bf16, and out = h (w2_packed NULL: slice [out_coff, out_coff + cout) of rows of
 *   out_ctot) or out = act2(h . W2 + bias2) (cout = coutp = 256: FeedForward's hidden tensor stays in LDS; 256 output columns).
 * Bit-identical to av2x_add_layernorm_bf16 followed by av2x_linear_bf16 (twice); the normalised tensor (and the hidden one) never
 * exist in HBM.  Packing and constraints of W / W2: as av2x_linear_bf16; k = 256. */
int av2x_ln_linear_bf16(float* x, const uint16_t* delta, int64_t add_rows, int32_t write_back_x, const float* gamma, const float* beta, float eps,
                        const uint16_t* w_packed, const float* bias, int32_t act, int32_t cout, int32_t coutp,
                        const uint16_t* w2_packed, const float* bias2, int32_t act2, uint16_t* out, int32_t out_ctot,
                        int32_t out_coff, int64_t m, av2x_stream_t stream);
int av2x_hgt_attention_bf16(const uint16_t* proj, const float* mask, const int32_t* types_host, uint16_t* out, int32_t n,
                            int32_t n_query, int32_t hw, int32_t heads, int32_t dim_head, av2x_stream_t stream);
int av2x_window_attention_bf16(const uint16_t* qkv, int32_t ctot, int32_t coff, const float* pos_embedding, uint16_t* out,
                               int32_t n, int32_t h, int32_t w, int32_t heads, int32_t dim_head, int32_t window,
                               av2x_stream_t stream);
int av2x_split_attn_gap_bf16(const uint16_t* s0, const uint16_t* s1, const uint16_t* s2, float* gap,
                             float* scratch /* n*128*c floats */, int32_t n, int32_t hw, int32_t c, av2x_stream_t stream);
int av2x_split_attn_combine_bf16(const uint16_t* s0, const uint16_t* s1, const uint16_t* s2, const float* logits,
                                 const float* residual, float* out, int32_t n, int32_t hw, int32_t c, av2x_stream_t stream);
/* SplitAttn's combine (split_attn.py:55-61) as the producer of the stream in front of PreNormResidual(FeedForward)
 * (v2xvit_basic.py:137-159): for the m rows (m % hw == 0, hw % 64 == 0: tokens per agent)
 *   x[r] = (x[r] (+ delta[r], r < add_rows)) + sum_b softmax_b(logits[agent(r)]) s_b[r]   -- av2x_split_attn_combine(_delta)_bf16's bits,
 *   written back; then exactly av2x_ln_linear_bf16 on the new x (LayerNorm -> Linear [-> Linear]).  One pass over x instead of the
 * combine's read + write and the LayerNorm's read. */
int av2x_combine_ln_linear_bf16(float* x, const uint16_t* delta, int64_t add_rows, const uint16_t* s0, const uint16_t* s1,
                                const uint16_t* s2, const float* logits, int64_t hw, const float* gamma, const float* beta, float eps,
                                const uint16_t* w_packed, const float* bias, int32_t act, int32_t cout, int32_t coutp,
                                const uint16_t* w2_packed, const float* bias2, int32_t act2, uint16_t* out, int32_t out_ctot,
                                int32_t out_coff, int64_t m, av2x_stream_t stream);
/* One branch of the pyramid window attention with its output projection (mswin.py:52-96 BaseWindowAttention: to_out Linear of the
 * attention output): out[slice] = bf16(bf16(WindowAttention(qkv slice)) . W + bias), bit-identical to av2x_window_attention_bf16
 * followed by av2x_linear_bf16; the attention output stays in LDS.  h % 4 == 0, w % 16 == 0 (4 x 16-pixel blocks);
 * heads x dim_head = 256; (dim_head, window) in (16,2) (32,4) (64,4); w_packed as av2x_linear_bf16 (256 -> 256). */
int av2x_window_attention_linear_bf16(const uint16_t* qkv, int32_t ctot, int32_t coff, const float* pos_embedding,
                                      const uint16_t* w_packed, const float* bias, uint16_t* out, int32_t out_ctot, int32_t out_coff,
                                      int32_t n, int32_t h, int32_t w, int32_t heads, int32_t dim_head, int32_t window,
                                      av2x_stream_t stream);
/* PreNormResidual(PyramidWindowAttention) up to the three branch outputs in ONE launch (v2xvit_basic.py:137-159, mswin.py:99-145):
 * LayerNorm(x (+ delta, bf16, NULL = none; x itself is not rewritten)) -> the three [q | k | v] projections (wqkv_packed: 256 -> 2304,
 * branch b at columns [768 b, 768 b + 768)) -> window attention of branch b (heads[b] x dim_heads[b] = 256, windows[b]; pos_embeddings[b]) ->
 * to_out of branch b (wout3_packed: 256 -> 768, branch b at columns [256 b, 256 b + 256); bias_out3 (768,)) -> outs[b] (n, h, w, 256) bf16.
 * One workgroup per 4 x 16-pixel block (h % 4 == 0, w % 16 == 0) holds the normalised panel and q, k, v of the current branch in LDS:
 * neither the normalised tensor nor the 2304-wide QKV tensor nor the attention output exist in HBM.  Bit-identical to
 * av2x_ln_linear_bf16 (write_back_x = 0) + 3 x av2x_window_attention_linear_bf16.  pos_embeddings / outs / heads / dim_heads / windows: host
 * arrays of 3. */
int av2x_ln_qkv_window_attention_bf16(const float* x, const uint16_t* delta, const float* gamma, const float* beta, float eps,
                                      const uint16_t* wqkv_packed, const float* bias_qkv, const uint16_t* wout3_packed,
                                      const float* bias_out3, const float* const* pos_embeddings, uint16_t* const* outs,
                                      const int32_t* heads, const int32_t* dim_heads, const int32_t* windows, int32_t n, int32_t h,
                                      int32_t w, av2x_stream_t stream);
/* as av2x_split_attn_combine_bf16 with residual + delta (bf16, same shape; the pending add of av2x_ln_linear_bf16) as the residual */
int av2x_split_attn_combine_delta_bf16(const uint16_t* s0, const uint16_t* s1, const uint16_t* s2, const float* logits,
                                       const float* residual, const uint16_t* delta, float* out, int32_t n, int32_t hw, int32_t c,
                                       av2x_stream_t stream);

/* The reference replaces an EMPTY cloud by two dummy points before voxelising (sp_voxel_preprocessor.py:80-90).
 * av2x_voxelize_dummy_if_empty does the same on the device after av2x_voxelize / av2x_prepare_voxelize: if *n_voxels == 0
 * the two dummy points are voxelised into rows 0.. of the (zeroed) outputs and *n_voxels is updated; otherwise nothing
 * happens.  capacity (pillars the output buffers hold) must be >= 2. */
int av2x_voxelize_dummy_if_empty(const float* range6, const float* voxel3, int32_t max_points, int32_t max_voxels,
                                 int32_t capacity, float* voxels, int32_t* coords, int32_t* num_points, int32_t* n_voxels,
                                 av2x_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * When2com fusion (models/when2com_modules/when2com.py).
 * av2x_linear_rows: y (m,n) = act(x (m,k) . w (n,k)^T + bias) for a few rows m (one per agent): the km_generator
 *   MLPs (:283-297), whose first layer streams 256*H/4*W/4 inputs per output feature (HBM-bound).  w is the
 *   nn.Linear weight as stored (row-major (n,k)); k % 4 == 0; x, w 16-byte aligned; act 0 none / 1 ReLU; bias may be
 *   NULL.  workspace: av2x_linear_rows_workspace_bytes(m, n, k) bytes of device scratch (split-K partials, reduced
 *   in a fixed order: results are run-to-run identical).
 * av2x_when2com_fuse: p = softmax_j(keys[j,:] . query) over the n_agents keys (MIMOGeneralDotProductAttention
 *   :320-348, softmax over the key axis, `query` = attention_net.linear(query_net(ego))), out[e] = sum_j p_j *
 *   agents[j][e] (:340-347).  agents: HOST array of n_agents device pointers, each elems_per_agent floats, 16-byte
 *   aligned (the maps may live in different buffers, e.g. slices of an all-gather result); elems_per_agent % 4 == 0,
 *   n_agents <= 32; coef (n_agents,) receives p (may be NULL).
 * ------------------------------------------------------------------------------------ */
/* Training (what torch autograd of when2com.py gives the reference):
 * av2x_linear_rows_backward: with dz = dy where the forward's ReLU passed (act 1: y > 0; act 0: everywhere) --
 *   dx (m,k) = dz . w, dw (n,k) = dz^T . x, db (n) = column sums of dz; any of the three may be NULL.  w is streamed once (dx) and dw
 *   written once per group of 8 rows; fixed summation orders.  y (m,n) = the forward output (needed for act 1).
 * av2x_when2com_fuse_backward: gradients of av2x_when2com_fuse given dout (elems_per_agent): dagents[j] = p_j dout (HOST array of device
 *   pointers, may be NULL), dkeys (n_agents, key_size) and dquery (key_size) through the softmax over the keys (either may be NULL);
 *   coef = the p the forward returned; workspace: av2x_when2com_fuse_backward_workspace_bytes(n_agents).
 * av2x_warp_affine_simple_backward (declared with av2x_warp_affine_backward): the adjoint of av2x_warp_affine_simple. */
int av2x_linear_rows_backward(const float* x, const float* w, const float* y, const float* dy, int32_t m, int32_t n, int32_t k,
                              int32_t act, float* dx, float* dw, float* db, av2x_stream_t stream);
/* V2VNet training pieces (v2vnet_modules/v2v_fuse.py:110-170, convgru.py:52-73; forward: av2x_v2v_aggregate):
 * av2x_gru_gate: the one-step ConvGRU with a zero hidden state, out = sigmoid(beta) * tanh(cnm) over n_elems (% 4 == 0) elements
 *   (beta = the update-gate half of conv_gates, cnm = conv_can; the reset gate multiplies the zero state), and its gradient.
 * av2x_agent_argmax: out[e] = max_j x[j][e] with index[e] = the first maximising agent (torch.max over dim 0), and the gradient routed to it. */
int av2x_gru_gate(const float* beta, const float* cnm, uint64_t n_elems, float* out, av2x_stream_t stream);
int av2x_gru_gate_backward(const float* beta, const float* cnm, const float* dout, uint64_t n_elems, float* dbeta, float* dcnm,
                           av2x_stream_t stream);
int av2x_agent_argmax(const float* x, int32_t n_agents, uint64_t elems_per_agent, float* out, uint8_t* index, av2x_stream_t stream);
int av2x_agent_argmax_backward(const float* dout, const uint8_t* index, int32_t n_agents, uint64_t elems_per_agent, float* dx,
                            av2x_stream_t stream);
uint64_t av2x_when2com_fuse_backward_workspace_bytes(int32_t n_agents);
int av2x_when2com_fuse_backward(const float* keys, const float* query, const float* coef, int32_t n_agents, int32_t key_size,
                                const float* const* agents, uint64_t elems_per_agent, const float* dout, float* const* dagents,
                                float* dkeys, float* dquery, void* workspace, av2x_stream_t stream);
uint64_t av2x_linear_rows_workspace_bytes(int32_t m, int32_t n, int32_t k);
int av2x_linear_rows(const float* x, const float* w, const float* bias, int32_t m, int32_t n, int32_t k, int32_t act,
                     float* y, void* workspace, uint64_t workspace_bytes, av2x_stream_t stream);
int av2x_when2com_fuse(const float* keys, const float* query, int32_t n_agents, int32_t key_size,
                       const float* const* agents, uint64_t elems_per_agent, float* out, float* coef, av2x_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Import-time native modules of the reference's callers on this path: without them the reference's unmodified
 * build_dataset / build_postprocessor raise ImportError on a ROCm box.  Python shims with the reference's module names and signatures:
 * airv2x_perception_amd/opencood_iface/roiaware_pool3d_cuda.py, box_overlaps.py (install_import_shims() registers them in sys.modules).
 *
 * roiaware_pool3d_cuda (pcdet_utils/roiaware_pool3d/src/roiaware_pool3d.cpp:27-183, roiaware_pool3d_kernel.cu:1-359;
 * boxes are [x, y, z, dx, dy, dz, heading] with (x, y, z) the centre):
 * av2x_points_in_boxes_cpu: HOST arrays; pts_indices (n_boxes, n_pts) = 1 where the point lies in the box (margin 1e-2 in x / y, :128).
 * av2x_points_in_boxes_gpu: boxes (batch, n_boxes, 7), pts (batch, n_pts, 3) on the device; box_idx_of_points (batch, n_pts) receives the index
 *   of the FIRST box holding the point and is left untouched otherwise (the caller pre-fills -1, roiaware_pool3d_utils.py:61-63); margin 1e-5.
 * av2x_roiaware_pool3d_forward: rois (boxes_num, 7), pts (pts_num, 3), pts_feature (pts_num, channels) -> pts_idx_of_voxels
 *   (boxes_num, out_x, out_y, out_z, max_pts_each_voxel) int32 [slot 0 = count, then the point indices in increasing order, at most
 *   max_pts_each_voxel - 1], pooled_features (boxes_num, out_x, out_y, out_z, channels) and, for pool_method 0 (max), argmax (same shape,
 *   -1 = empty voxel).  pts_idx_of_voxels and pooled_features must be ZERO on entry (the reference's new_zeros); pool_method 1 = average.
 *   workspace: av2x_roiaware_pool3d_workspace_bytes(boxes_num, pts_num) bytes of device scratch (the (box, point) voxel codes).
 * av2x_roiaware_pool3d_backward: grad_in (pts_num, channels) += the pooled gradients (atomic adds; zero it first).
 *
 * box_overlaps (utils/box_overlaps.pyx:17-143; HOST float32 arrays, boxes are [x1, y1, x2, y2] with the "+1" pixel convention):
 * av2x_bbox_overlaps: out (n, k) = IoU of boxes[i] and query[j]; intersections_only != 0: intersection / area(query[j]) (bbox_intersections).
 * av2x_box_vote: dets are rows of `cols` >= 5 floats (x1, y1, x2, y2, score, ...): out[i] = score-weighted mean of the dets_all boxes with
 *   IoU >= 0.5 to dets_nms[i] (nan where none), out[i][4] = the original score, further columns 0.
 * ------------------------------------------------------------------------------------ */
int av2x_points_in_boxes_cpu(const float* boxes, const float* pts, int32_t n_boxes, int32_t n_pts, int32_t* pts_indices);
int av2x_points_in_boxes_gpu(const float* boxes, const float* pts, int32_t batch, int32_t n_boxes, int32_t n_pts,
                             int32_t* box_idx_of_points, av2x_stream_t stream);
uint64_t av2x_roiaware_pool3d_workspace_bytes(int32_t boxes_num, int32_t pts_num);
int av2x_roiaware_pool3d_forward(const float* rois, const float* pts, const float* pts_feature, int32_t boxes_num, int32_t pts_num,
                                 int32_t channels, int32_t max_pts_each_voxel, int32_t out_x, int32_t out_y, int32_t out_z,
                                 int32_t* argmax, int32_t* pts_idx_of_voxels, float* pooled_features, int32_t pool_method,
                                 void* workspace, av2x_stream_t stream);
int av2x_roiaware_pool3d_backward(const int32_t* pts_idx_of_voxels, const int32_t* argmax, const float* grad_out, int32_t boxes_num,
                                  int32_t out_x, int32_t out_y, int32_t out_z, int32_t channels, int32_t max_pts_each_voxel,
                                  float* grad_in, int32_t pool_method, av2x_stream_t stream);
int av2x_bbox_overlaps(const float* boxes, const float* query, int32_t n, int32_t k, float* out, int32_t intersections_only);
int av2x_box_vote(const float* dets_nms, const float* dets_all, int32_t n, int32_t m, int32_t cols, float* out);

#ifdef __cplusplus
}
#endif
#endif /* AIRV2X_HIP_H */
