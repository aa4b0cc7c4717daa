// Camera lift-splat: LiftSplatShootEncoder.get_geometry + voxel_pooling
// (models/common_modules/airv2x_encoder.py:133-167, 208-275) as ONE pass over the frustum points.
//
// The reference materialises the (B,N,D,fH,fW,3) geometry, turns it into voxel indices, sorts ALL points by voxel
// (argsort of ~0.7 M keys per agent), forms a running fp32 sum over every point and takes differences at the voxel
// boundaries (QuickCumsum, utils/camera_utils.py:341-365) -- whose rounding error grows with the running sum, i.e. with the
// number of points -- and scatters the sums into a zeroed BEV volume.  Here every point computes its own voxel in
// registers (same fp32 operation order as the reference: subtract post_trans, 3x3, scale x,y by depth, 3x3, add trans;
// `.long()` truncation toward zero, so a point up to one voxel below the lower bound lands in cell 0 exactly as there) and
// adds its C features to that voxel with 64-bit FIXED-POINT atomics (2^-32 resolution): integer addition is associative,
// so the result is bit-reproducible whatever order the hardware takes, and each voxel's sum is exact to 2^-32 per addend
// instead of carrying the cumsum's cancellation error.  A second kernel converts to fp32 in the NHWC layout the
// convolutions read.  HBM-bound: P * C * 4 bytes of features read once.
#include "av2x_common.hpp"

namespace {

struct LssCam {          // per (b, n): inverse(post_rots) row-major, post_trans, rots @ inverse(intrins) row-major, trans
    float ipr[9], pt[3], comb[9], tr[3];
};

struct LssGrid {
    float lo[3];         // bx - dx / 2
    float dx[3];
    int nx[3];
};

constexpr float kFix = 4294967296.0f;   // 2^32

// one wave per group of points: lane -> (point within group, float4 chunk of the C channels)
template <int CK>   // C = 4 * CK lanes-per-point... C / 4 lanes per point
__global__ __launch_bounds__(256) void lss_pool_kernel(const float* __restrict__ x, const float* __restrict__ frustum,
                                                       const LssCam* __restrict__ cams, LssGrid g, int pts_per_cam,
                                                       int cams_per_batch, long long npts, int C,
                                                       unsigned long long* __restrict__ acc, float* __restrict__ geom_out) {
    constexpr int LPP = CK;                         // lanes per point (C / 4)
    const int t = threadIdx.x % LPP;
    const long long p = (long long)blockIdx.x * (256 / LPP) + threadIdx.x / LPP;
    if (p >= npts) return;
    const int cam = (int)(p / pts_per_cam);
    const int fp = (int)(p - (long long)cam * pts_per_cam);
    const LssCam& c = cams[cam];
    // points = frustum - post_trans ; points = inverse(post_rots) @ points            (:147-152)
    const float fx = __fsub_rn(frustum[3 * fp + 0], c.pt[0]);
    const float fy = __fsub_rn(frustum[3 * fp + 1], c.pt[1]);
    const float fz = __fsub_rn(frustum[3 * fp + 2], c.pt[2]);
    auto dot3 = [](const float* m, float a, float b, float d) {
        return __fadd_rn(__fadd_rn(__fmul_rn(m[0], a), __fmul_rn(m[1], b)), __fmul_rn(m[2], d));
    };
    const float ux = dot3(c.ipr + 0, fx, fy, fz), uy = dot3(c.ipr + 3, fx, fy, fz), uz = dot3(c.ipr + 6, fx, fy, fz);
    // (x * z, y * z, z)                                                                  (:155-161)
    const float vx = __fmul_rn(ux, uz), vy = __fmul_rn(uy, uz), vz = uz;
    // combine @ points + trans                                                          (:164-166)
    const float gx = __fadd_rn(dot3(c.comb + 0, vx, vy, vz), c.tr[0]);
    const float gy = __fadd_rn(dot3(c.comb + 3, vx, vy, vz), c.tr[1]);
    const float gz = __fadd_rn(dot3(c.comb + 6, vx, vy, vz), c.tr[2]);
    if (geom_out && t == 0) { geom_out[3 * p + 0] = gx; geom_out[3 * p + 1] = gy; geom_out[3 * p + 2] = gz; }
    // ((geom - (bx - dx / 2)) / dx).long()                                              (:227)
    const long long ix = (long long)__fdiv_rn(__fsub_rn(gx, g.lo[0]), g.dx[0]);
    const long long iy = (long long)__fdiv_rn(__fsub_rn(gy, g.lo[1]), g.dx[1]);
    const long long iz = (long long)__fdiv_rn(__fsub_rn(gz, g.lo[2]), g.dx[2]);
    if (ix < 0 || ix >= g.nx[0] || iy < 0 || iy >= g.nx[1] || iz < 0 || iz >= g.nx[2]) return;   // `kept` (:239-246)
    if (!acc) return;
    const int b = cam / cams_per_batch;
    const size_t cell = (((size_t)b * g.nx[2] + iz) * g.nx[1] + iy) * g.nx[0] + ix;
    const float4 v = *reinterpret_cast<const float4*>(x + (size_t)p * C + 4 * t);
    unsigned long long* a = acc + cell * C + 4 * t;
    if (v.x != 0.f) atomicAdd(a + 0, (unsigned long long)__float2ll_rn(v.x * kFix));
    if (v.y != 0.f) atomicAdd(a + 1, (unsigned long long)__float2ll_rn(v.y * kFix));
    if (v.z != 0.f) atomicAdd(a + 2, (unsigned long long)__float2ll_rn(v.z * kFix));
    if (v.w != 0.f) atomicAdd(a + 3, (unsigned long long)__float2ll_rn(v.w * kFix));
}

// acc (B, nz, ny, nx, C) fixed point -> out (B, ny, nx, nz * C) fp32: channel index z * C + c, the order of
// torch.cat(final.unbind(dim=2), 1) (:272) with the channels innermost (NHWC)
__global__ __launch_bounds__(256) void lss_finish_kernel(const long long* __restrict__ acc, int B, int nz, int ny, int nx, int C,
                                                         float* __restrict__ out) {
    const size_t n = (size_t)B * nz * ny * nx * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int xx = (int)(r % nx); r /= nx;
        const int yy = (int)(r % ny); r /= ny;
        const int zz = (int)(r % nz);
        const int b = (int)(r / nz);
        out[(((size_t)b * ny + yy) * nx + xx) * ((size_t)nz * C) + (size_t)zz * C + c] = (float)((double)acc[i] * (1.0 / 4294967296.0));
    }
}

}  // namespace

extern "C" uint64_t av2x_lss_pool_workspace_bytes(int32_t b, int32_t nx, int32_t ny, int32_t nz, int32_t c) {
    return (uint64_t)b * nx * ny * nz * c * 8ull;
}

extern "C" int av2x_lss_voxel_pool(const float* x, const float* frustum, const float* cam_params, int32_t b, int32_t n_cams,
                                   int32_t pts_per_cam, int32_t c, const float* lo3, const float* dx3, const int32_t* nx3,
                                   void* workspace, float* out, float* geom_out, av2x_stream_t stream) {
    if (!frustum || !cam_params || !lo3 || !dx3 || !nx3) return av2x::fail("av2x_lss_voxel_pool: null argument");
    if ((x == nullptr) != (out == nullptr) || (out && !workspace)) return av2x::fail("av2x_lss_voxel_pool: x / out / workspace must come together");
    if (b <= 0 || n_cams <= 0 || pts_per_cam <= 0) return av2x::fail("av2x_lss_voxel_pool: bad sizes");
    if (out && c != 64 && c != 128 && c != 32) return av2x::fail("av2x_lss_voxel_pool: c=%d (32, 64 or 128 feature channels)", c);
    LssGrid g;
    for (int i = 0; i < 3; ++i) { g.lo[i] = lo3[i]; g.dx[i] = dx3[i]; g.nx[i] = nx3[i]; }
    hipStream_t st = av2x::as_stream(stream);
    const long long npts = (long long)b * n_cams * pts_per_cam;
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(workspace);
    const size_t cells = (size_t)b * g.nx[0] * g.nx[1] * g.nx[2];
    if (out) {
        hipError_t e = hipMemsetAsync(acc, 0, cells * c * 8ull, st);
        if (e != hipSuccess) return av2x::fail("av2x_lss_voxel_pool: memset: %s", hipGetErrorString(e));
    }
    const LssCam* cams = reinterpret_cast<const LssCam*>(cam_params);
    const int cc = out ? c : 64;
#define AV2X_LSS_LAUNCH(CK)                                                                                                \
    hipLaunchKernelGGL(lss_pool_kernel<CK>, dim3((unsigned)((npts + (256 / CK) - 1) / (256 / CK))), dim3(256), 0, st, x, frustum, \
                       cams, g, pts_per_cam, n_cams, npts, cc, out ? acc : nullptr, geom_out)
    if (cc == 32) AV2X_LSS_LAUNCH(8);
    else if (cc == 64) AV2X_LSS_LAUNCH(16);
    else AV2X_LSS_LAUNCH(32);
#undef AV2X_LSS_LAUNCH
    if (out)
        hipLaunchKernelGGL(lss_finish_kernel, dim3(2048), dim3(256), 0, st, reinterpret_cast<const long long*>(acc), b, g.nx[2],
                           g.nx[1], g.nx[0], c, out);
    return av2x::check_launch("lss_pool_kernel");
}
