// Pillar feature net (10 -> 64 linear + folded BatchNorm1d + ReLU + max over the pillar's points)
// fused with the BEV scatter.  HBM-bound: 512 B of points + 16 B coords + 4 B count in, 256 B out
// per pillar.  One wave64 per pillar: lanes 0..31 each load one point as a
// single 16-byte load (a pillar = one 512-byte coalesced run), the augmented 10-vector of every
// point is staged in LDS, then lane c produces output channel c and the wave writes one
// contiguous 256-byte NHWC pixel of the canvas.
#include "av2x_common.hpp"

namespace {

constexpr int kPts = 32;   // max_points_per_voxel (where2com yaml:115)
constexpr int kFeat = 10;  // x y z i | cluster xyz | centre xyz (airv2x_pillar_vfe.py:140-148)
constexpr int kOut = 64;
constexpr int kLdF = 12;   // padded feature row (three 16-byte reads)

// MODE 0: scatter into the NHWC canvas (the fused product path).  MODE 1: ``canvas`` is the (n_pillars, 64)
// ``pillar_features`` array of the stand-alone PillarVFE module (airv2x_pillar_vfe.py:156-158).  MODE 2: scatter, with the
// voxelizer's own outputs as input: coordinates (M,3) z,y,x without the agent column (the agent is ``agent0``) and
// the pillar count read from DEVICE memory (``n_dev``; n_pillars is the buffers' capacity) -- no host round trip
// between the voxelizer and the network.
template <int MODE>
__global__ __launch_bounds__(256) void pillar_vfe_scatter_kernel(
    const float4* __restrict__ vox, const int4* __restrict__ coords, const int* __restrict__ npts, int n_pillars,
    const int* __restrict__ n_dev,
    const float* __restrict__ pfn_w, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
    float vx, float vy, float vz, float xoff, float yoff, float zoff, float* __restrict__ canvas, int agent0,
    const int* __restrict__ slot_map, int n_agents, int ny, int nx, unsigned long long* __restrict__ nonzero,
    unsigned char* __restrict__ occupancy) {
    __shared__ __attribute__((aligned(16))) float feats[4][kPts][kLdF];
    unsigned written_nz = 0;    // non-zero values this wave has put on the canvas (wave-uniform)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves_total = gridDim.x * 4;

    // this lane's output channel: weights stay in registers for the whole grid-stride loop
    float w[kFeat];
#pragma unroll
    for (int j = 0; j < kFeat; ++j) w[j] = pfn_w[lane * kFeat + j];
    const float sc = bn_scale[lane], sh = bn_shift[lane];

    if (MODE == 2) n_pillars = min(n_pillars, n_dev[0]);
    for (int pil = blockIdx.x * 4 + wave; pil < n_pillars; pil += waves_total) {
        int4 c;  // agent, z, y, x
        if (MODE == 2) {
            const int* c3 = reinterpret_cast<const int*>(coords) + 3 * (size_t)pil;
            c = make_int4(0, c3[0], c3[1], c3[2]);
        } else {
            c = coords[pil];
        }
        int num = npts[pil];
        num = num < 0 ? 0 : (num > kPts ? kPts : num);
        float4 pt = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < kPts) pt = vox[(size_t)pil * kPts + lane];
        // mean over the stored rows (zero padded rows add 0) / num_points   (:121-123)
        float sx = pt.x, sy = pt.y, sz = pt.z;
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) {
            sx += __shfl_xor(sx, o);
            sy += __shfl_xor(sy, o);
            sz += __shfl_xor(sz, o);
        }
        const float fn = (float)npts[pil];
        const float mx = sx / fn, my = sy / fn, mz = sz / fn;
        if (lane < kPts) {
            float f[kLdF];
            const bool valid = lane < num;  // get_paddings_indicator (:95-103, :151-153)
            // pillar centre: coord * voxel + offset, no FMA contraction (matches the two-op reference)
            const float cx = __fadd_rn(__fmul_rn((float)c.w, vx), xoff);
            const float cy = __fadd_rn(__fmul_rn((float)c.z, vy), yoff);
            const float cz = __fadd_rn(__fmul_rn((float)c.y, vz), zoff);
            f[0] = pt.x; f[1] = pt.y; f[2] = pt.z; f[3] = pt.w;
            f[4] = pt.x - mx; f[5] = pt.y - my; f[6] = pt.z - mz;
            f[7] = pt.x - cx; f[8] = pt.y - cy; f[9] = pt.z - cz;
            f[10] = 0.f; f[11] = 0.f;
#pragma unroll
            for (int j = 0; j < kLdF; ++j) feats[wave][lane][j] = valid ? f[j] : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS writes of this wave are visible
        // rows >= num are all-zero: linear(0) = 0 -> BN -> relu(shift); they take part in the max
        // exactly as in the reference (features *= mask happens BEFORE the PFN, :153-155)
        float best = (num < kPts) ? fmaxf(sh, 0.f) : 0.f;
        for (int q = 0; q < num; ++q) {
            const float4 a = *reinterpret_cast<const float4*>(&feats[wave][q][0]);
            const float4 b = *reinterpret_cast<const float4*>(&feats[wave][q][4]);
            const float4 d = *reinterpret_cast<const float4*>(&feats[wave][q][8]);
            float acc = a.x * w[0];
            acc = fmaf(a.y, w[1], acc); acc = fmaf(a.z, w[2], acc); acc = fmaf(a.w, w[3], acc);
            acc = fmaf(b.x, w[4], acc); acc = fmaf(b.y, w[5], acc); acc = fmaf(b.z, w[6], acc);
            acc = fmaf(b.w, w[7], acc); acc = fmaf(d.x, w[8], acc); acc = fmaf(d.y, w[9], acc);
            best = fmaxf(best, fmaxf(fmaf(acc, sc, sh), 0.f));
        }
        if (MODE == 1) {
            canvas[(size_t)pil * kOut + lane] = best;
        } else if (c.x >= 0 && c.x < n_agents && (unsigned)c.z < (unsigned)ny && (unsigned)c.w < (unsigned)nx) {
            const int agent = slot_map ? slot_map[c.x] : agent0 + c.x;
            // idx = z + y*nx + x with nz == 1 (point_pillar_scatter.py:59-61)
            const size_t pix = ((size_t)agent * ny + c.z) * nx + c.w + c.y;
            canvas[pix * kOut + lane] = best;
            if (occupancy != nullptr && lane == 0) occupancy[pix] = 1;
            written_nz += (unsigned)__popcll(__ballot(best != 0.f));
        }
        __builtin_amdgcn_wave_barrier();
    }
    // count_nonzero of the (zeroed, then scattered) canvas without reading it back: every cell is written by at most one pillar (the
    // voxelizer's coordinates are unique per agent), so the canvas's non-zero count is the sum of the written non-zeros
    // Same-address atomics serialise in L2 (~10 ns each: one per wave of a 2 048-workgroup launch was 80 us of tail), hence ONE per
    // workgroup, spread over AV2X_NZ_SLOTS counters that live 128 bytes apart (different lines, different channels); the count is their sum.
    if (MODE != 1 && nonzero != nullptr) {
        __shared__ unsigned wave_nz[4];
        if (lane == 0) wave_nz[wave] = written_nz;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned t = wave_nz[0] + wave_nz[1] + wave_nz[2] + wave_nz[3];
            if (t) atomicAdd(nonzero + (size_t)(blockIdx.x % AV2X_NZ_SLOTS) * AV2X_NZ_STRIDE, (unsigned long long)t);
        }
    }
}

// PointPillarScatter alone (point_pillar_scatter.py:39-68): canvas[agent, y, x, :] = pillar_features[pil, :].
// One thread per (pillar, 4 channels); the canvas is NHWC so a pillar is one contiguous C*4-byte run.
__global__ void pillar_scatter_kernel(const float4* __restrict__ feats, const int4* __restrict__ coords, int n_pillars,
                                      int c4, float4* __restrict__ canvas, int n_agents, int ny, int nx) {
    const size_t total = (size_t)n_pillars * c4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int pil = (int)(i / c4), q = (int)(i % c4);
        const int4 c = coords[pil];
        if (c.x < 0 || c.x >= n_agents || (unsigned)c.z >= (unsigned)ny || (unsigned)c.w >= (unsigned)nx) continue;
        const size_t pix = ((size_t)c.x * ny + c.z) * nx + c.w + c.y;
        canvas[pix * c4 + q] = feats[i];
    }
}

__global__ void count_nonzero_kernel(const float4* __restrict__ x, size_t n4, const float* __restrict__ tail, int ntail,
                                     unsigned long long* result) {
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = x[i];
        c += (v.x != 0.f) + (v.y != 0.f) + (v.z != 0.f) + (v.w != 0.f);
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) c += tail[threadIdx.x] != 0.f;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o);
    // 1024-thread workgroups, 512 of them: ONE same-address atomic per workgroup (they serialise in L2 at ~10 ns each)
    __shared__ unsigned long long part[16];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += part[k];
        if (t) atomicAdd(result, t);
    }
}

// result[0] = sum of the AV2X_NZ_SLOTS strided counters of the counting scatter (one wave)
__global__ void nonzero_slots_sum_kernel(const unsigned long long* __restrict__ slots, unsigned long long* __restrict__ result) {
    unsigned long long c = threadIdx.x < AV2X_NZ_SLOTS ? slots[(size_t)threadIdx.x * AV2X_NZ_STRIDE] : 0ull;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o);
    if (threadIdx.x == 0) result[0] = c;
}

}  // namespace

extern "C" int av2x_nonzero_slots_sum(const unsigned long long* slots, unsigned long long* result, av2x_stream_t stream) {
    if (!slots || !result) return av2x::fail("av2x_nonzero_slots_sum: null argument");
    hipLaunchKernelGGL(nonzero_slots_sum_kernel, dim3(1), dim3(64), 0, av2x::as_stream(stream), slots, result);
    return av2x::check_launch("nonzero_slots_sum_kernel");
}

extern "C" int av2x_pillar_vfe_scatter_count(const float* voxel_features, const int32_t* voxel_coords,
                                       const int32_t* voxel_num_points, int32_t n_pillars, const float* pfn_w,
                                       const float* bn_scale, const float* bn_shift, const float* geom, float* canvas,
                                       int32_t canvas_agent0, const int32_t* slot_map, int32_t n_agents_type, int32_t ny,
                                       int32_t nx, unsigned long long* nonzero, uint8_t* occupancy, av2x_stream_t stream) {
    if (n_pillars == 0) return 0;
    if (!voxel_features || !voxel_coords || !voxel_num_points || !pfn_w || !bn_scale || !bn_shift || !geom || !canvas)
        return av2x::fail("av2x_pillar_vfe_scatter: null argument");
    if (n_pillars < 0 || ny <= 0 || nx <= 0) return av2x::fail("av2x_pillar_vfe_scatter: bad sizes");
    int blocks = (n_pillars + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(pillar_vfe_scatter_kernel<0>, dim3(blocks), dim3(256), 0, av2x::as_stream(stream),
                       reinterpret_cast<const float4*>(voxel_features), reinterpret_cast<const int4*>(voxel_coords),
                       voxel_num_points, n_pillars, (const int*)nullptr, pfn_w, bn_scale, bn_shift, geom[0], geom[1], geom[2], geom[3],
                       geom[4], geom[5], canvas, canvas_agent0, slot_map, n_agents_type, ny, nx, nonzero, occupancy);
    return av2x::check_launch("pillar_vfe_scatter_kernel");
}

extern "C" int av2x_pillar_vfe_scatter(const float* voxel_features, const int32_t* voxel_coords,
                                       const int32_t* voxel_num_points, int32_t n_pillars, const float* pfn_w,
                                       const float* bn_scale, const float* bn_shift, const float* geom, float* canvas,
                                       int32_t canvas_agent0, const int32_t* slot_map, int32_t n_agents_type, int32_t ny,
                                       int32_t nx, av2x_stream_t stream) {
    return av2x_pillar_vfe_scatter_count(voxel_features, voxel_coords, voxel_num_points, n_pillars, pfn_w, bn_scale, bn_shift, geom, canvas,
                                         canvas_agent0, slot_map, n_agents_type, ny, nx, nullptr, nullptr, stream);
}

extern "C" int av2x_pillar_vfe(const float* voxel_features, const int32_t* voxel_coords, const int32_t* voxel_num_points,
                               int32_t n_pillars, const float* pfn_w, const float* bn_scale, const float* bn_shift,
                               const float* geom, float* pillar_features, av2x_stream_t stream) {
    if (n_pillars == 0) return 0;
    if (!voxel_features || !voxel_coords || !voxel_num_points || !pfn_w || !bn_scale || !bn_shift || !geom || !pillar_features)
        return av2x::fail("av2x_pillar_vfe: null argument");
    if (n_pillars < 0) return av2x::fail("av2x_pillar_vfe: bad sizes");
    int blocks = (n_pillars + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(pillar_vfe_scatter_kernel<1>, dim3(blocks), dim3(256), 0, av2x::as_stream(stream),
                       reinterpret_cast<const float4*>(voxel_features), reinterpret_cast<const int4*>(voxel_coords),
                       voxel_num_points, n_pillars, (const int*)nullptr, pfn_w, bn_scale, bn_shift, geom[0], geom[1], geom[2], geom[3],
                       geom[4], geom[5], pillar_features, 0, (const int*)nullptr, 0, 0, 0, (unsigned long long*)nullptr, (unsigned char*)nullptr);
    return av2x::check_launch("pillar_vfe_kernel");
}

extern "C" int av2x_pillar_vfe_scatter_dev_count(const float* voxel_features, const int32_t* voxel_coords3,
                                           const int32_t* voxel_num_points, const int32_t* n_pillars_dev, int32_t capacity,
                                           const float* pfn_w, const float* bn_scale, const float* bn_shift, const float* geom,
                                           float* canvas, int32_t canvas_slot, int32_t ny, int32_t nx, unsigned long long* nonzero,
                                                 uint8_t* occupancy, av2x_stream_t stream) {
    if (capacity == 0) return 0;
    if (!voxel_features || !voxel_coords3 || !voxel_num_points || !n_pillars_dev || !pfn_w || !bn_scale || !bn_shift || !geom || !canvas)
        return av2x::fail("av2x_pillar_vfe_scatter_dev: null argument");
    if (capacity < 0 || ny <= 0 || nx <= 0 || canvas_slot < 0) return av2x::fail("av2x_pillar_vfe_scatter_dev: bad sizes");
    int blocks = (capacity + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(pillar_vfe_scatter_kernel<2>, dim3(blocks), dim3(256), 0, av2x::as_stream(stream),
                       reinterpret_cast<const float4*>(voxel_features), reinterpret_cast<const int4*>(voxel_coords3),
                       voxel_num_points, capacity, n_pillars_dev, pfn_w, bn_scale, bn_shift, geom[0], geom[1], geom[2], geom[3],
                       geom[4], geom[5], canvas, canvas_slot, (const int*)nullptr, 1, ny, nx, nonzero, occupancy);
    return av2x::check_launch("pillar_vfe_scatter_kernel<2>");
}

extern "C" int av2x_pillar_vfe_scatter_dev(const float* voxel_features, const int32_t* voxel_coords3,
                                           const int32_t* voxel_num_points, const int32_t* n_pillars_dev, int32_t capacity,
                                           const float* pfn_w, const float* bn_scale, const float* bn_shift, const float* geom,
                                           float* canvas, int32_t canvas_slot, int32_t ny, int32_t nx, av2x_stream_t stream) {
    return av2x_pillar_vfe_scatter_dev_count(voxel_features, voxel_coords3, voxel_num_points, n_pillars_dev, capacity, pfn_w, bn_scale, bn_shift,
                                             geom, canvas, canvas_slot, ny, nx, nullptr, nullptr, stream);
}

extern "C" int av2x_pillar_scatter(const float* pillar_features, const int32_t* voxel_coords, int32_t n_pillars,
                                   int32_t channels, float* canvas, int32_t n_agents, int32_t ny, int32_t nx,
                                   av2x_stream_t stream) {
    if (n_pillars == 0) return 0;
    if (!pillar_features || !voxel_coords || !canvas) return av2x::fail("av2x_pillar_scatter: null argument");
    if (n_pillars < 0 || channels <= 0 || channels % 4 || n_agents <= 0 || ny <= 0 || nx <= 0)
        return av2x::fail("av2x_pillar_scatter: bad sizes (channels must be a multiple of 4)");
    const size_t total = (size_t)n_pillars * (channels / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(pillar_scatter_kernel, dim3(blocks), dim3(256), 0, av2x::as_stream(stream),
                       reinterpret_cast<const float4*>(pillar_features), reinterpret_cast<const int4*>(voxel_coords),
                       n_pillars, channels / 4, reinterpret_cast<float4*>(canvas), n_agents, ny, nx);
    return av2x::check_launch("pillar_scatter_kernel");
}

extern "C" int av2x_fill_zero(void* ptr, uint64_t bytes, av2x_stream_t stream) {
    if (bytes == 0) return 0;
    if (!ptr) return av2x::fail("av2x_fill_zero: null pointer");
    hipError_t e = hipMemsetAsync(ptr, 0, bytes, av2x::as_stream(stream));
    if (e != hipSuccess) return av2x::fail("av2x_fill_zero: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int av2x_count_nonzero(const float* x, uint64_t n_elems, unsigned long long* result, av2x_stream_t stream) {
    if (n_elems == 0) return 0;
    if (!x || !result) return av2x::fail("av2x_count_nonzero: null argument");
    if (reinterpret_cast<uintptr_t>(x) % 16) return av2x::fail("av2x_count_nonzero: pointer must be 16-byte aligned");
    const size_t n4 = n_elems / 4;
    const int ntail = (int)(n_elems - n4 * 4);
    hipLaunchKernelGGL(count_nonzero_kernel, dim3(512), dim3(1024), 0, av2x::as_stream(stream),
                       reinterpret_cast<const float4*>(x), n4, x + n4 * 4, ntail, result);
    return av2x::check_launch("count_nonzero_kernel");
}
