// The two import-time native modules of the reference's callers on this path, behind the C-ABI:
//   * roiaware_pool3d_cuda  (pcdet_utils/roiaware_pool3d/src/roiaware_pool3d.cpp:27-183 + roiaware_pool3d_kernel.cu:1-359):
//       points_in_boxes_cpu (host), points_in_boxes_gpu, forward (max / avg RoI-aware pooling), backward;
//   * opencood.utils.box_overlaps (utils/box_overlaps.pyx:17-143): bbox_overlaps, bbox_intersections, box_vote (host, "+1" pixel convention).
// Device side: the point -> voxel mask is one thread per (box, point) as in the reference; the per-box collection of the inside points
// (the reference: ONE thread per box walking all points, roiaware_pool3d_kernel.cu:78-113) is one 64-lane wave per box: 64 points per
// step, the lanes that hold an inside point append in lane order (ballot + lowest-lane loop), so each voxel lists its points in
// increasing point index and stops at max_pts_each_voxel - 1 -- the same lists, without a serial walk over the outside points.
#include <cmath>

#include "av2x_common.hpp"

namespace {

// lidar_to_local_coords + check_pt_in_box3d (roiaware_pool3d_kernel.cu:16-37; roiaware_pool3d.cpp:117-140): float rotation,
// the half-extent comparisons in double (dz / 2.0, dx / 2.0 + MARGIN are double expressions there)
template <bool HOST>
__host__ __device__ inline int pt_in_box3d(const float* pt, const float* box, float margin, float& lx, float& ly) {
    const float x = pt[0], y = pt[1], z = pt[2];
    const float cx = box[0], cy = box[1], cz = box[2], dx = box[3], dy = box[4], dz = box[5], rz = box[6];
    if ((double)fabsf(z - cz) > (double)dz / 2.0) return 0;
    const float cosa = cosf(-rz), sina = sinf(-rz);
    const float sx = x - cx, sy = y - cy;
    lx = sx * cosa + sy * (-sina);
    ly = sx * sina + sy * cosa;
    return ((double)fabsf(lx) < (double)dx / 2.0 + (double)margin) & ((double)fabsf(ly) < (double)dy / 2.0 + (double)margin);
}

__global__ void pib_kernel(int batch, int nb, int np, const float* __restrict__ boxes, const float* __restrict__ pts, int* __restrict__ out) {
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch || i >= np) return;
    const float* bx = boxes + (size_t)b * nb * 7;
    const float* p = pts + ((size_t)b * np + i) * 3;
    float lx = 0.f, ly = 0.f;
    for (int k = 0; k < nb; ++k)
        if (pt_in_box3d<false>(p, bx + k * 7, 1e-5f, lx, ly)) {   // the FIRST box that holds the point (:318-324)
            out[(size_t)b * np + i] = k;
            break;
        }
}

__global__ void roi_mask_kernel(int nb, int np, int ox, int oy, int oz, const float* __restrict__ rois, const float* __restrict__ pts,
                                int* __restrict__ mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= np || b >= nb) return;
    const float* p = pts + (size_t)i * 3;
    const float* r = rois + (size_t)b * 7;
    float lx = 0.f, ly = 0.f;
    int code = -1;
    if (pt_in_box3d<false>(p, r, 1e-5f, lx, ly)) {
        const float lz = p[2] - r[2];
        const float dx = r[3], dy = r[4], dz = r[5];
        const float xr = dx / ox, yr = dy / oy, zr = dz / oz;
        // :62-68: the index is an int converted to unsigned, then clamped with min(max(., 0), out - 1) on unsigned values
        unsigned xi = (unsigned)(int)((lx + dx / 2) / xr), yi = (unsigned)(int)((ly + dy / 2) / yr), zi = (unsigned)(int)((lz + dz / 2) / zr);
        xi = min(max(xi, 0u), (unsigned)(ox - 1));
        yi = min(max(yi, 0u), (unsigned)(oy - 1));
        zi = min(max(zi, 0u), (unsigned)(oz - 1));
        code = (int)((xi << 16) + (yi << 8) + zi);
    }
    mask[(size_t)b * np + i] = code;
}

__global__ __launch_bounds__(64) void roi_collect_kernel(int nb, int np, int maxp, int ox, int oy, int oz, const int* __restrict__ mask,
                                                         int* __restrict__ vox) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= nb) return;
    int* v = vox + (size_t)b * ox * oy * oz * maxp;
    const int* m = mask + (size_t)b * np;
    for (int p0 = 0; p0 < np; p0 += 64) {
        const int k = p0 + lane;
        const int code = k < np ? m[k] : -1;
        unsigned long long live = __ballot(code != -1);
        while (live) {                                   // inside points of this step, in increasing point index
            const int src = __ffsll((long long)live) - 1;
            live &= live - 1;
            if (lane == src) {
                const unsigned c = (unsigned)code;
                const unsigned base = ((c >> 16) & 0xFF) * oy * oz * maxp + ((c >> 8) & 0xFF) * oz * maxp + (c & 0xFF) * maxp;
                const int cnt = v[base];
                if (cnt < maxp - 1) {                    // slot 0 is the counter (:86)
                    v[base + cnt + 1] = k;
                    v[base] = cnt + 1;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // the next lane of this wave sees the counter
        }
    }
}

template <bool AVG>
__global__ void roi_pool_kernel(int nb, int ch, int maxp, int nvox, const float* __restrict__ feat, const int* __restrict__ vox,
                                float* __restrict__ pooled, int* __restrict__ argmax) {
    const int b = blockIdx.z, c = blockIdx.y, vflat = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb || c >= ch || vflat >= nvox) return;
    const int* v = vox + ((size_t)b * nvox + vflat) * maxp;
    const size_t o = ((size_t)b * nvox + vflat) * ch + c;
    const int total = v[0];
    if (AVG) {
        float s = 0.f;
        for (int k = 1; k <= total; ++k) s += feat[(size_t)v[k] * ch + c];
        if (total > 0) pooled[o] = s / total;            // :183-185 (empty voxels keep the caller's zeros)
    } else {
        int am = -1;
        float mx = -INFINITY;                            // float max_val = -1e50 (:135): -inf as a float
        for (int k = 1; k <= total; ++k) {
            const float f = feat[(size_t)v[k] * ch + c];
            if (f > mx) { mx = f; am = v[k]; }           // strictly greater: the first maximum
        }
        if (am != -1) pooled[o] = mx;
        argmax[o] = am;
    }
}

template <bool AVG>
__global__ void roi_pool_backward_kernel(int nb, int ch, int maxp, int nvox, const int* __restrict__ vox, const int* __restrict__ argmax,
                                         const float* __restrict__ gout, float* __restrict__ gin) {
    const int b = blockIdx.z, c = blockIdx.y, vflat = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb || c >= ch || vflat >= nvox) return;
    const size_t o = ((size_t)b * nvox + vflat) * ch + c;
    if (AVG) {
        const int* v = vox + ((size_t)b * nvox + vflat) * maxp;
        const int total = v[0];
        const float g = 1 / fmaxf((float)total, 1.0f);
        for (int k = 1; k <= total; ++k) atomicAdd(gin + (size_t)v[k] * ch + c, gout[o] * g);
    } else {
        if (argmax[o] == -1) return;
        atomicAdd(gin + (size_t)argmax[o] * ch + c, gout[o] * 1);
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------- roiaware_pool3d_cuda
extern "C" int av2x_points_in_boxes_cpu(const float* boxes, const float* pts, int32_t n_boxes, int32_t n_pts, int32_t* pts_indices) {
    if (n_boxes < 0 || n_pts < 0) return av2x::fail("av2x_points_in_boxes_cpu: negative size");
    if ((n_boxes && !boxes) || (n_pts && !pts) || (n_boxes && n_pts && !pts_indices)) return av2x::fail("av2x_points_in_boxes_cpu: null argument");
    float lx = 0.f, ly = 0.f;
    for (int i = 0; i < n_boxes; ++i)
        for (int j = 0; j < n_pts; ++j)                  // MARGIN 1e-2 on the host (roiaware_pool3d.cpp:128), 1e-5 on the device
            pts_indices[(size_t)i * n_pts + j] = pt_in_box3d<true>(pts + (size_t)j * 3, boxes + (size_t)i * 7, 1e-2f, lx, ly);
    return 0;
}

extern "C" int av2x_points_in_boxes_gpu(const float* boxes, const float* pts, int32_t batch, int32_t n_boxes, int32_t n_pts,
                                        int32_t* box_idx_of_points, av2x_stream_t stream) {
    if (batch < 0 || n_boxes < 0 || n_pts < 0) return av2x::fail("av2x_points_in_boxes_gpu: negative size");
    if (batch == 0 || n_pts == 0) return 0;
    if (!pts || !box_idx_of_points || (n_boxes && !boxes)) return av2x::fail("av2x_points_in_boxes_gpu: null argument");
    if (batch > 65535) return av2x::fail("av2x_points_in_boxes_gpu: batch > 65535");
    hipLaunchKernelGGL(pib_kernel, dim3((n_pts + 255) / 256, batch), dim3(256), 0, av2x::as_stream(stream), batch, n_boxes, n_pts, boxes, pts,
                       box_idx_of_points);
    return av2x::check_launch("points_in_boxes_kernel");
}

extern "C" uint64_t av2x_roiaware_pool3d_workspace_bytes(int32_t boxes_num, int32_t pts_num) {
    return (uint64_t)(boxes_num > 0 ? boxes_num : 0) * (uint64_t)(pts_num > 0 ? pts_num : 0) * 4ull;
}

extern "C" int av2x_roiaware_pool3d_forward(const float* rois, const float* pts, const float* pts_feature, int32_t boxes_num, int32_t pts_num,
                                            int32_t channels, int32_t max_pts_each_voxel, int32_t out_x, int32_t out_y, int32_t out_z,
                                            int32_t* argmax, int32_t* pts_idx_of_voxels, float* pooled_features, int32_t pool_method,
                                            void* workspace, av2x_stream_t stream) {
    if (boxes_num < 0 || pts_num < 0 || channels < 0) return av2x::fail("av2x_roiaware_pool3d_forward: negative size");
    if (out_x <= 0 || out_y <= 0 || out_z <= 0 || out_x >= 256 || out_y >= 256 || out_z >= 256)
        return av2x::fail("av2x_roiaware_pool3d_forward: out sizes must be in 1..255 (the voxel index is packed in 8 bits, roiaware_pool3d.cpp:51)");
    if (max_pts_each_voxel < 1) return av2x::fail("av2x_roiaware_pool3d_forward: max_pts_each_voxel < 1");
    if (pool_method != 0 && pool_method != 1) return av2x::fail("av2x_roiaware_pool3d_forward: pool_method 0 (max) or 1 (avg)");
    if (boxes_num == 0) return 0;
    if (boxes_num > 65535 || channels > 65535) return av2x::fail("av2x_roiaware_pool3d_forward: more than 65535 boxes / channels");
    if (!rois || !pts_idx_of_voxels || !pooled_features || (pool_method == 0 && !argmax) || (pts_num && (!pts || !pts_feature || !workspace)))
        return av2x::fail("av2x_roiaware_pool3d_forward: null argument");
    hipStream_t st = av2x::as_stream(stream);
    const int nvox = out_x * out_y * out_z;
    if (pts_num > 0) {
        int* mask = reinterpret_cast<int*>(workspace);
        hipLaunchKernelGGL(roi_mask_kernel, dim3((pts_num + 255) / 256, boxes_num), dim3(256), 0, st, boxes_num, pts_num, out_x, out_y, out_z, rois,
                           pts, mask);
        hipLaunchKernelGGL(roi_collect_kernel, dim3(boxes_num), dim3(64), 0, st, boxes_num, pts_num, max_pts_each_voxel, out_x, out_y, out_z, mask,
                           pts_idx_of_voxels);
    }
    if (channels > 0) {
        const dim3 grid((nvox + 255) / 256, channels, boxes_num);
        if (pool_method == 0)
            hipLaunchKernelGGL(roi_pool_kernel<false>, grid, dim3(256), 0, st, boxes_num, channels, max_pts_each_voxel, nvox, pts_feature,
                               pts_idx_of_voxels, pooled_features, argmax);
        else
            hipLaunchKernelGGL(roi_pool_kernel<true>, grid, dim3(256), 0, st, boxes_num, channels, max_pts_each_voxel, nvox, pts_feature,
                               pts_idx_of_voxels, pooled_features, argmax);
    }
    return av2x::check_launch("roiaware_pool3d");
}

extern "C" int av2x_roiaware_pool3d_backward(const int32_t* pts_idx_of_voxels, const int32_t* argmax, const float* grad_out, int32_t boxes_num,
                                             int32_t out_x, int32_t out_y, int32_t out_z, int32_t channels, int32_t max_pts_each_voxel,
                                             float* grad_in, int32_t pool_method, av2x_stream_t stream) {
    if (boxes_num < 0 || channels < 0 || out_x <= 0 || out_y <= 0 || out_z <= 0) return av2x::fail("av2x_roiaware_pool3d_backward: bad size");
    if (pool_method != 0 && pool_method != 1) return av2x::fail("av2x_roiaware_pool3d_backward: pool_method 0 (max) or 1 (avg)");
    if (boxes_num == 0 || channels == 0) return 0;
    if (boxes_num > 65535 || channels > 65535) return av2x::fail("av2x_roiaware_pool3d_backward: more than 65535 boxes / channels");
    if (!grad_out || !grad_in || (pool_method == 0 ? !argmax : !pts_idx_of_voxels)) return av2x::fail("av2x_roiaware_pool3d_backward: null argument");
    const int nvox = out_x * out_y * out_z;
    const dim3 grid((nvox + 255) / 256, channels, boxes_num);
    if (pool_method == 0)
        hipLaunchKernelGGL(roi_pool_backward_kernel<false>, grid, dim3(256), 0, av2x::as_stream(stream), boxes_num, channels, max_pts_each_voxel, nvox,
                           pts_idx_of_voxels, argmax, grad_out, grad_in);
    else
        hipLaunchKernelGGL(roi_pool_backward_kernel<true>, grid, dim3(256), 0, av2x::as_stream(stream), boxes_num, channels, max_pts_each_voxel, nvox,
                           pts_idx_of_voxels, argmax, grad_out, grad_in);
    return av2x::check_launch("roiaware_pool3d_backward");
}

// ---------------------------------------------------------------------------------------------------------------- box_overlaps (host)
// Arithmetic as Cython compiles the .pyx (checked against the reference's own compiled module, tests/golden/box_overlaps_pin.npz): every
// "+ 1" is "+ 1.0" in DOUBLE on a float difference, typed variables (box_area, iw, ih, ua: float) round at the assignment, the products
// of two such double sums are formed in double, iw * ih is a float product.
static inline double p1(float a, float b) { return (double)(a - b) + 1.0; }

extern "C" int av2x_bbox_overlaps(const float* boxes, const float* query, int32_t n, int32_t k, float* out, int32_t intersections_only) {
    if (n < 0 || k < 0) return av2x::fail("av2x_bbox_overlaps: negative size");
    if ((n && !boxes) || (k && !query) || (n && k && !out)) return av2x::fail("av2x_bbox_overlaps: null argument");
    for (size_t i = 0; i < (size_t)n * k; ++i) out[i] = 0.f;
    for (int q = 0; q < k; ++q) {
        const float* qb = query + (size_t)q * 4;
        const float box_area = (float)(p1(qb[2], qb[0]) * p1(qb[3], qb[1]));
        for (int i = 0; i < n; ++i) {
            const float* b = boxes + (size_t)i * 4;
            const float iw = (float)p1(fminf(b[2], qb[2]), fmaxf(b[0], qb[0]));
            if (iw > 0) {
                const float ih = (float)p1(fminf(b[3], qb[3]), fmaxf(b[1], qb[1]));
                if (ih > 0) {
                    if (intersections_only) {
                        out[(size_t)i * k + q] = iw * ih / box_area;              // box_overlaps.pyx:97
                    } else {
                        const float ua = (float)(p1(b[2], b[0]) * p1(b[3], b[1]) + (double)box_area - (double)(iw * ih));   // :50-55
                        out[(size_t)i * k + q] = iw * ih / ua;
                    }
                }
            }
        }
    }
    return 0;
}

extern "C" int av2x_box_vote(const float* dets_nms, const float* dets_all, int32_t n, int32_t m, int32_t cols, float* out) {
    if (n < 0 || m < 0 || cols < 5) return av2x::fail("av2x_box_vote: detections are rows of >= 5 floats (x1, y1, x2, y2, score)");
    if ((n && (!dets_nms || !out)) || (m && !dets_all)) return av2x::fail("av2x_box_vote: null argument");
    const float thresh = 0.5f;
    for (size_t i = 0; i < (size_t)n * cols; ++i) out[i] = 0.f;
    for (int i = 0; i < n; ++i) {
        const float* det = dets_nms + (size_t)i * cols;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        float acc_score = 0.f;
        for (int j = 0; j < m; ++j) {
            const float* d2 = dets_all + (size_t)j * cols;
            const float bi0 = fmaxf(det[0], d2[0]), bi1 = fmaxf(det[1], d2[1]), bi2 = fminf(det[2], d2[2]), bi3 = fminf(det[3], d2[3]);
            // bi2 is an untyped Python float in the .pyx (:108 declares "bit2"): bi2 - bi0 + 1 is double arithmetic; bi3 - bi1 is a float difference
            const float iw = (float)((double)bi2 - (double)bi0 + 1.0), ih = (float)p1(bi3, bi1);
            if (!(iw > 0 && ih > 0)) continue;
            const float ua = (float)(p1(det[2], det[0]) * p1(det[3], det[1]) + p1(d2[2], d2[0]) * p1(d2[3], d2[1]) - (double)(iw * ih));
            const float ov = iw * ih / ua;
            if (ov < thresh) continue;
            for (int c = 0; c < 4; ++c) acc[c] += d2[4] * d2[c];
            acc_score += d2[4];
        }
        for (int c = 0; c < 4; ++c) out[(size_t)i * cols + c] = acc[c] / acc_score;   // 0 / 0 = nan when nothing overlaps, as in the .pyx
        out[(size_t)i * cols + 4] = det[4];
    }
    return 0;
}
