// Deterministic GPU pillar voxelizer with the observable semantics of spconv's
// Point2VoxelCPU3d (the third-party call behind SpVoxelPreprocessor.preprocess,
// data_utils/pre_processor/sp_voxel_preprocessor.py:93-110):
//   * c = floor((p - range_min) / voxel_size) in fp32, point dropped when outside the grid;
//   * voxels are numbered in order of first appearance in the input; only the first
//     `max_voxels` of them exist;
//   * a voxel keeps its first `max_points` points in input order; rows beyond are zero.
// No sort: a counting sort by grid cell (atomics only for counts / unordered slot fill) followed
// by an in-cell rank count restores input order exactly, so the result is bit-reproducible.
//
//   k_cell   per point : cell id, cnt[cell]++, first[cell] = min(index)
//   k_flagsum + k_rank  grid-wide exclusive scans (a) over "is first point of its cell" flags in point
//            order -> voxel rank + M, (b) over cnt of the first points, same order -> list offsets
//   k_fill   per point : unordered append of the point index to its cell's list
//   k_emit   per point : pos = #{j in list : j < i}; write voxels / coords / num_points
#include "av2x_common.hpp"

namespace {

struct VoxGeom {
    float rmin[3];
    float vs[3];
    int grid[3];  // x, y, z
};

// Optional point preparation folded into the voxelizer (av2x_prepare_voxelize): the point the voxelizer sees at
// position i is prepare(points[perm[i]]); dropped points simply get no cell, which leaves first-appearance voxel order
// and in-voxel point order exactly as if the cloud had been compacted first (av2x_prepare_points).
struct PrepInline {
    int enabled, project, mask_ego;
    const int* perm;
    float T[16];
    float r[6];
};

__device__ __forceinline__ bool prep_point(const float4* __restrict__ pts, int i, const PrepInline& P, float4* out) {
    float4 q = pts[(P.enabled && P.perm) ? P.perm[i] : i];
    bool keep = true;
    if (P.enabled) {
        if (P.mask_ego) keep = !(q.x >= -1.95f && q.x <= 2.95f && q.y >= -1.1f && q.y <= 1.1f);
        if (P.project) {
            float o[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float acc = __fmul_rn(q.x, P.T[j * 4 + 0]);
                acc = __fmaf_rn(q.y, P.T[j * 4 + 1], acc);
                acc = __fmaf_rn(q.z, P.T[j * 4 + 2], acc);
                acc = __fmaf_rn(1.0f, P.T[j * 4 + 3], acc);
                o[j] = acc;
            }
            q.x = o[0]; q.y = o[1]; q.z = o[2];
        }
        keep = keep && q.x > P.r[0] && q.x < P.r[3] && q.y > P.r[1] && q.y < P.r[4] && q.z > P.r[2] && q.z < P.r[5];
    }
    *out = q;
    return keep;
}

__global__ void k_cell(const float4* __restrict__ pts, int n, VoxGeom g, PrepInline P, int* __restrict__ cell,
                       int* __restrict__ cnt, int* __restrict__ first) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p;
    if (!prep_point(pts, i, P, &p)) { cell[i] = -1; return; }
    // IEEE fp32 subtract and divide, as the CPU implementation (no reciprocal, no contraction)
    const float fx = floorf(__fdiv_rn(__fsub_rn(p.x, g.rmin[0]), g.vs[0]));
    const float fy = floorf(__fdiv_rn(__fsub_rn(p.y, g.rmin[1]), g.vs[1]));
    const float fz = floorf(__fdiv_rn(__fsub_rn(p.z, g.rmin[2]), g.vs[2]));
    int c = -1;
    if (fx >= 0.f && fx < (float)g.grid[0] && fy >= 0.f && fy < (float)g.grid[1] && fz >= 0.f && fz < (float)g.grid[2]) {
        c = ((int)fz * g.grid[1] + (int)fy) * g.grid[0] + (int)fx;
        atomicAdd(&cnt[c], 1);
        atomicMin(&first[c], i);
    }
    cell[i] = c;
}

// The two exclusive scans over the points in input order -- (a) voxel rank = number of earlier "first points" (-> M),
// (b) list offset = points of the cells that appeared earlier (a scan over the points, not over the 140 800 grid cells:
// only occupied cells need a list) -- as a grid-wide two-launch scan: k_flagsum leaves every 256-point workgroup's
// totals, k_rank adds the totals of the workgroups before it (a few hundred pairs, summed by the workgroup itself) to
// its in-workgroup prefix.  A 108 k-point cloud no longer walks through one workgroup 27 times.
__device__ __forceinline__ void first_point_pair(const int* __restrict__ cell, const int* __restrict__ cnt,
                                                 const int* __restrict__ first, int i, int n, int* c_out, int* a, int* b) {
    *a = 0; *b = 0; *c_out = -1;
    if (i < n) {
        const int c = cell[i];
        if (c >= 0 && first[c] == i) { *a = 1; *b = cnt[c]; *c_out = c; }
    }
}

__global__ __launch_bounds__(256) void k_flagsum(const int* __restrict__ cell, int n, const int* __restrict__ cnt,
                                                 const int* __restrict__ first, int2* __restrict__ wg_sum) {
    __shared__ int sa[4], sb[4];
    int c, a, b;
    first_point_pair(cell, cnt, first, blockIdx.x * 256 + threadIdx.x, n, &c, &a, &b);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; sb[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) wg_sum[blockIdx.x] = make_int2(sa[0] + sa[1] + sa[2] + sa[3], sb[0] + sb[1] + sb[2] + sb[3]);
}

__global__ __launch_bounds__(256) void k_rank(const int* __restrict__ cell, int n, const int* __restrict__ cnt,
                                              const int* __restrict__ first, const int2* __restrict__ wg_sum,
                                              int* __restrict__ vrank, int* __restrict__ offs, int* __restrict__ m_out,
                                              int max_voxels) {
    __shared__ int pa[4], pb[4], wa[4], wb[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int oa = 0, ob = 0;
    for (int g = threadIdx.x; g < (int)blockIdx.x; g += 256) { const int2 v = wg_sum[g]; oa += v.x; ob += v.y; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { oa += __shfl_xor(oa, o); ob += __shfl_xor(ob, o); }
    if (lane == 0) { pa[wave] = oa; pb[wave] = ob; }
    int c, a, b;
    first_point_pair(cell, cnt, first, blockIdx.x * 256 + threadIdx.x, n, &c, &a, &b);
    int sa = a, sb = b;                                  // inclusive wave scans
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int xa = __shfl_up(sa, o), xb = __shfl_up(sb, o);
        if (lane >= o) { sa += xa; sb += xb; }
    }
    if (lane == 63) { wa[wave] = sa; wb[wave] = sb; }
    __syncthreads();
    int ba = pa[0] + pa[1] + pa[2] + pa[3], bb = pb[0] + pb[1] + pb[2] + pb[3];
    for (int k = 0; k < wave; ++k) { ba += wa[k]; bb += wb[k]; }
    if (a) { vrank[c] = ba + sa - a; offs[c] = bb + sb - b; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) {
        const int tot = ba + sa;
        m_out[0] = tot < max_voxels ? tot : max_voxels;
    }
}

__global__ void k_fill(const int* __restrict__ cell, int n, const int* __restrict__ offs, int* __restrict__ fill,
                       int* __restrict__ list) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = cell[i];
    if (c < 0) return;
    list[offs[c] + atomicAdd(&fill[c], 1)] = i;
}

__global__ void k_emit(const float4* __restrict__ pts, const int* __restrict__ cell, int n, const int* __restrict__ cnt,
                       const int* __restrict__ first, const int* __restrict__ vrank, const int* __restrict__ offs,
                       const int* __restrict__ list, VoxGeom g, PrepInline P, int max_points, int max_voxels,
                       float4* __restrict__ voxels, int* __restrict__ coords, int* __restrict__ num) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = cell[i];
    if (c < 0) return;
    const int v = vrank[c];
    if (v >= max_voxels) return;
    const int k = cnt[c];
    int pos = 0;
    const int* l = list + offs[c];
    for (int j = 0; j < k; ++j) pos += l[j] < i;
    if (pos < max_points) {
        float4 q;
        prep_point(pts, i, P, &q);   // the prepared (projected) point; kept by construction (it has a cell)
        voxels[(size_t)v * max_points + pos] = q;
    }
    if (first[c] == i) {
        const int x = c % g.grid[0], yz = c / g.grid[0];
        coords[3 * v + 0] = yz / g.grid[1];
        coords[3 * v + 1] = yz % g.grid[1];
        coords[3 * v + 2] = x;
        num[v] = k < max_points ? k : max_points;
    }
}

__global__ void k_init(int* __restrict__ cnt, int* __restrict__ first, int* __restrict__ fill, int ncell) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    cnt[c] = 0;
    fill[c] = 0;
    first[c] = 0x7fffffff;
}

// The reference replaces an empty cloud by two dummy points before voxelising (sp_voxel_preprocessor.py:80-90).  With the
// pillar count staying on the device the same substitution happens here: if the voxelizer found no pillar, voxelise the
// two dummy points (one thread; same cell arithmetic as k_cell).
__global__ void k_dummy_if_empty(VoxGeom g, int max_points, int max_voxels, float4* __restrict__ voxels, int* __restrict__ coords,
                                 int* __restrict__ num, int* __restrict__ n_voxels) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || n_voxels[0] != 0) return;
    const float4 d[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(-0.218277f, -11.13425732f, -80.05884552f, 1.230595649e-38f)};
    int m = 0, cell0 = -1;
    for (int i = 0; i < 2; ++i) {
        const float fx = floorf(__fdiv_rn(__fsub_rn(d[i].x, g.rmin[0]), g.vs[0]));
        const float fy = floorf(__fdiv_rn(__fsub_rn(d[i].y, g.rmin[1]), g.vs[1]));
        const float fz = floorf(__fdiv_rn(__fsub_rn(d[i].z, g.rmin[2]), g.vs[2]));
        if (!(fx >= 0.f && fx < (float)g.grid[0] && fy >= 0.f && fy < (float)g.grid[1] && fz >= 0.f && fz < (float)g.grid[2])) continue;
        const int c = ((int)fz * g.grid[1] + (int)fy) * g.grid[0] + (int)fx;
        if (c == cell0) {   // second point of the same pillar
            if (num[0] < max_points) { voxels[num[0]] = d[i]; num[0] += 1; }
            continue;
        }
        if (m >= max_voxels) continue;
        if (m == 0) cell0 = c;
        for (int r = 0; r < max_points; ++r) voxels[(size_t)m * max_points + r] = make_float4(0.f, 0.f, 0.f, 0.f);
        voxels[(size_t)m * max_points] = d[i];
        coords[3 * m + 0] = (int)fz; coords[3 * m + 1] = (int)fy; coords[3 * m + 2] = (int)fx;
        num[m] = 1;
        ++m;
    }
    n_voxels[0] = m;
}

// ---- point preparation -------------------------------------------------------
struct PrepParams {
    float T[16];
    float r[6];
    int n, project, mask_ego;
};

// q = points[perm[i]] ; drop ego-box returns (pcd_utils.py:168-190, closed box, sensor frame) ; project with
// the 4x4 (box_utils.py:1038-1067; torch's fp32 einsum = one multiply then an FMA chain over k, reproduced
// exactly) ; keep points strictly inside the range (pcd_utils.py:136-165).
__global__ __launch_bounds__(256) void prep_flag(const float4* __restrict__ pts, const int* __restrict__ perm, PrepParams p,
                                                 float4* __restrict__ tmp, int* __restrict__ flag, int* __restrict__ wg_count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool keep = false;
    if (i < p.n) {
        float4 q = pts[perm ? perm[i] : i];
        keep = true;
        if (p.mask_ego) keep = !(q.x >= -1.95f && q.x <= 2.95f && q.y >= -1.1f && q.y <= 1.1f);
        if (p.project) {
            float o[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float acc = __fmul_rn(q.x, p.T[j * 4 + 0]);
                acc = __fmaf_rn(q.y, p.T[j * 4 + 1], acc);
                acc = __fmaf_rn(q.z, p.T[j * 4 + 2], acc);
                acc = __fmaf_rn(1.0f, p.T[j * 4 + 3], acc);
                o[j] = acc;
            }
            q.x = o[0]; q.y = o[1]; q.z = o[2];
        }
        keep = keep && q.x > p.r[0] && q.x < p.r[3] && q.y > p.r[1] && q.y < p.r[4] && q.z > p.r[2] && q.z < p.r[5];
        tmp[i] = q;
        flag[i] = keep ? 1 : 0;
    }
    __shared__ int wc[4];
    const unsigned long long bal = __ballot(keep);
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) wg_count[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}

// order-preserving compaction over the whole grid: offset of a workgroup = sum of the counts before it
__global__ __launch_bounds__(256) void prep_emit(const float4* __restrict__ tmp, const int* __restrict__ flag,
                                                 const int* __restrict__ wg_count, int n, float4* __restrict__ out,
                                                 int* __restrict__ count) {
    __shared__ int part[4], wc[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int s = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) s += wg_count[b];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) part[wave] = s;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool f = i < n && flag[i] != 0;
    const unsigned long long bal = __ballot(f);
    if (lane == 0) wc[wave] = __popcll(bal);
    __syncthreads();
    int off = part[0] + part[1] + part[2] + part[3];
    for (int k = 0; k < wave; ++k) off += wc[k];
    if (f) out[off + __popcll(bal & ((1ull << lane) - 1ull))] = tmp[i];
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
        count[0] = part[0] + part[1] + part[2] + part[3] + wc[0] + wc[1] + wc[2] + wc[3];
}

}  // namespace

extern "C" uint64_t av2x_prepare_points_workspace_bytes(int32_t n_points) {
    return (uint64_t)(n_points > 0 ? n_points : 0) * (sizeof(float4) + 2 * sizeof(int)) + 16;
}

extern "C" int av2x_prepare_points(const float* points, const int32_t* perm, int32_t n_points, const float* transform16,
                                   const float* range6, int32_t mask_ego, void* workspace, float* out, int32_t* count,
                                   av2x_stream_t stream) {
    if (!range6 || !count) return av2x::fail("av2x_prepare_points: null argument");
    if (n_points < 0) return av2x::fail("av2x_prepare_points: negative point count");
    hipStream_t st = av2x::as_stream(stream);
    if (n_points == 0) {
        hipError_t e = hipMemsetAsync(count, 0, sizeof(int), st);
        return e == hipSuccess ? 0 : av2x::fail("av2x_prepare_points: memset: %s", hipGetErrorString(e));
    }
    if (!points || !workspace || !out) return av2x::fail("av2x_prepare_points: null argument");
    PrepParams p;
    p.n = n_points; p.project = transform16 ? 1 : 0; p.mask_ego = mask_ego ? 1 : 0;
    for (int i = 0; i < 16; ++i) p.T[i] = transform16 ? transform16[i] : 0.f;
    for (int i = 0; i < 6; ++i) p.r[i] = range6[i];
    float4* tmp = reinterpret_cast<float4*>(workspace);
    int* flag = reinterpret_cast<int*>(tmp + n_points);
    int* wg_count = flag + n_points;          // (n_points + 255) / 256 entries of the second int array
    const dim3 g((n_points + 255) / 256), b(256);
    hipLaunchKernelGGL(prep_flag, g, b, 0, st, reinterpret_cast<const float4*>(points), perm, p, tmp, flag, wg_count);
    hipLaunchKernelGGL(prep_emit, g, b, 0, st, tmp, flag, wg_count, n_points, reinterpret_cast<float4*>(out), count);
    return av2x::check_launch("av2x_prepare_points");
}

extern "C" uint64_t av2x_voxelize_workspace_bytes(int32_t n_points, int32_t nx, int32_t ny, int32_t nz) {
    const uint64_t ncell = (uint64_t)nx * ny * nz;
    return (5 * ncell + 2 * (uint64_t)n_points + 2 * ((uint64_t)n_points / 256 + 2) + 4) * sizeof(int);
}

static int voxelize_impl(const float* points, int32_t n_points, const PrepInline& P, const float* range6, const float* voxel3,
                         int32_t max_points, int32_t max_voxels, void* workspace, float* voxels, int32_t* coords,
                         int32_t* num_points, int32_t* n_voxels, av2x_stream_t stream) {
    if (!range6 || !voxel3 || !workspace || !voxels || !coords || !num_points || !n_voxels)
        return av2x::fail("av2x_voxelize: null argument");
    if (n_points < 0 || max_points <= 0 || max_voxels <= 0) return av2x::fail("av2x_voxelize: bad sizes");
    VoxGeom g;
    for (int j = 0; j < 3; ++j) {
        g.rmin[j] = range6[j];
        g.vs[j] = voxel3[j];
        g.grid[j] = (int)llround(((double)range6[3 + j] - (double)range6[j]) / (double)voxel3[j]);
        if (g.grid[j] <= 0) return av2x::fail("av2x_voxelize: empty grid");
    }
    const long long ncell_ll = (long long)g.grid[0] * g.grid[1] * g.grid[2];
    if (ncell_ll > (1ll << 28)) return av2x::fail("av2x_voxelize: grid too large");
    const int ncell = (int)ncell_ll;
    hipStream_t st = av2x::as_stream(stream);
    int* w = reinterpret_cast<int*>(workspace);
    int *cnt = w, *first = w + ncell, *fill = w + 2 * (size_t)ncell, *vrank = w + 3 * (size_t)ncell,
        *offs = w + 4 * (size_t)ncell;
    int* cell = w + 5 * (size_t)ncell;
    int* list = cell + n_points;
    int2* wg_sum = reinterpret_cast<int2*>(list + n_points + ((reinterpret_cast<uintptr_t>(list + n_points) & 4) ? 1 : 0));   // 8-byte aligned
    hipLaunchKernelGGL(k_init, dim3((ncell + 255) / 256), dim3(256), 0, st, cnt, first, fill, ncell);
    // outputs are capacity-sized by the caller: voxels (cap, max_points, 4) must start zeroed
    const long long cap = n_points < max_voxels ? n_points : max_voxels;
    hipError_t e = hipMemsetAsync(voxels, 0, (size_t)cap * max_points * 4 * sizeof(float), st);
    if (e != hipSuccess) return av2x::fail("av2x_voxelize: memset: %s", hipGetErrorString(e));
    if (n_points == 0) {
        e = hipMemsetAsync(n_voxels, 0, sizeof(int), st);
        return e == hipSuccess ? 0 : av2x::fail("av2x_voxelize: memset: %s", hipGetErrorString(e));
    }
    if (!points) return av2x::fail("av2x_voxelize: null points");
    const dim3 gp((n_points + 255) / 256), bp(256);
    const float4* p4 = reinterpret_cast<const float4*>(points);
    hipLaunchKernelGGL(k_cell, gp, bp, 0, st, p4, n_points, g, P, cell, cnt, first);
    hipLaunchKernelGGL(k_flagsum, gp, bp, 0, st, cell, n_points, cnt, first, wg_sum);
    hipLaunchKernelGGL(k_rank, gp, bp, 0, st, cell, n_points, cnt, first, wg_sum, vrank, offs, n_voxels, max_voxels);
    hipLaunchKernelGGL(k_fill, gp, bp, 0, st, cell, n_points, offs, fill, list);
    hipLaunchKernelGGL(k_emit, gp, bp, 0, st, p4, cell, n_points, cnt, first, vrank, offs, list, g, P, max_points,
                       max_voxels, reinterpret_cast<float4*>(voxels), coords, num_points);
    return av2x::check_launch("av2x_voxelize");
}

extern "C" int av2x_voxelize(const float* points, int32_t n_points, const float* range6, const float* voxel3,
                             int32_t max_points, int32_t max_voxels, void* workspace, float* voxels, int32_t* coords,
                             int32_t* num_points, int32_t* n_voxels, av2x_stream_t stream) {
    PrepInline P;
    P.enabled = 0; P.project = 0; P.mask_ego = 0; P.perm = nullptr;
    return voxelize_impl(points, n_points, P, range6, voxel3, max_points, max_voxels, workspace, voxels, coords, num_points,
                         n_voxels, stream);
}

extern "C" int av2x_prepare_voxelize(const float* points, const int32_t* perm, int32_t n_points, const float* transform16,
                                     const float* crop_range6, int32_t mask_ego, const float* grid_range6, const float* voxel3,
                                     int32_t max_points, int32_t max_voxels, void* workspace, float* voxels, int32_t* coords,
                                     int32_t* num_points, int32_t* n_voxels, av2x_stream_t stream) {
    if (!crop_range6) return av2x::fail("av2x_prepare_voxelize: null crop range");
    PrepInline P;
    P.enabled = 1; P.project = transform16 ? 1 : 0; P.mask_ego = mask_ego ? 1 : 0; P.perm = perm;
    for (int i = 0; i < 16; ++i) P.T[i] = transform16 ? transform16[i] : 0.f;
    for (int i = 0; i < 6; ++i) P.r[i] = crop_range6[i];
    return voxelize_impl(points, n_points, P, grid_range6, voxel3, max_points, max_voxels, workspace, voxels, coords, num_points,
                         n_voxels, stream);
}

extern "C" int av2x_voxelize_dummy_if_empty(const float* range6, const float* voxel3, int32_t max_points, int32_t max_voxels,
                                            int32_t capacity, float* voxels, int32_t* coords, int32_t* num_points,
                                            int32_t* n_voxels, av2x_stream_t stream) {
    if (!range6 || !voxel3 || !voxels || !coords || !num_points || !n_voxels)
        return av2x::fail("av2x_voxelize_dummy_if_empty: null argument");
    if (max_points <= 0 || max_voxels <= 0 || capacity < 2) return av2x::fail("av2x_voxelize_dummy_if_empty: capacity must be >= 2 pillars");
    VoxGeom g;
    for (int j = 0; j < 3; ++j) {
        g.rmin[j] = range6[j];
        g.vs[j] = voxel3[j];
        g.grid[j] = (int)llround(((double)range6[3 + j] - (double)range6[j]) / (double)voxel3[j]);
        if (g.grid[j] <= 0) return av2x::fail("av2x_voxelize_dummy_if_empty: empty grid");
    }
    hipLaunchKernelGGL(k_dummy_if_empty, dim3(1), dim3(64), 0, av2x::as_stream(stream), g, max_points, max_voxels,
                       reinterpret_cast<float4*>(voxels), coords, num_points, n_voxels);
    return av2x::check_launch("k_dummy_if_empty");
}
