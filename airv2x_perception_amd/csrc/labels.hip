// Anchor-target assignment of the training labels: VoxelPostprocessor.generate_label_airv2x
// (data_utils/post_processor/voxel_postprocessor.py:217-354) with bbox_overlaps (utils/box_overlaps.pyx:17-57).
// The reference builds the (H*W*A, n_gt) IoU matrix on the host (Cython loop) and runs five numpy passes over it
// (argmax, two np.where, np.unique, fancy-index writes) in a DataLoader worker; here the matrix is never stored:
//   label_highest_kernel  one workgroup per ground-truth box: arg-max IoU over all anchors (first index on ties, np.argmax)
//   label_assign_kernel   one thread per anchor: its IoUs against the n_gt boxes in registers -> positive (first box above
//                         pos_threshold, else the first box it is the best anchor of: the np.unique(return_index) rule of
//                         :296-299), negative (all IoUs below neg_threshold and not a best anchor, :290-294,335-339), class
//                         label and the seven regression targets in float64 (:310-330).
// IoU arithmetic = the .pyx operation by operation, in the mixed float / double form Cython emits (see standup_iou).
#include "av2x_common.hpp"

namespace {

// box_overlaps.pyx:39-57 as Cython 3 compiles it: float operands, but the literal 1 is the double 1.0, so every "+ 1" and
// the union inside float(...) are double expressions rounded to float on assignment (iw, ih, box_area, ua are floats);
// the coordinate differences and iw * ih are float operations.
__device__ __forceinline__ float standup_iou(const float4 b, const float4 q) {
    const float box_area = (float)__dmul_rn(__dadd_rn((double)__fsub_rn(q.z, q.x), 1.0), __dadd_rn((double)__fsub_rn(q.w, q.y), 1.0));
    const float iw = (float)__dadd_rn((double)__fsub_rn(fminf(b.z, q.z), fmaxf(b.x, q.x)), 1.0);
    if (!(iw > 0.f)) return 0.f;
    const float ih = (float)__dadd_rn((double)__fsub_rn(fminf(b.w, q.w), fmaxf(b.y, q.y)), 1.0);
    if (!(ih > 0.f)) return 0.f;
    const float inter = __fmul_rn(iw, ih);
    const double barea = __dmul_rn(__dadd_rn((double)__fsub_rn(b.z, b.x), 1.0), __dadd_rn((double)__fsub_rn(b.w, b.y), 1.0));
    const float ua = (float)__dsub_rn(__dadd_rn(barea, (double)box_area), (double)inter);
    return __fdiv_rn(inter, ua);
}

__global__ __launch_bounds__(1024) void label_highest_kernel(const float4* __restrict__ anc, const float4* __restrict__ gt, int NA,
                                                             int* __restrict__ id_highest, float* __restrict__ iou_highest) {
    const int k = blockIdx.x;
    const float4 q = gt[k];
    float best = -1.f;
    int bi = 0x7fffffff;
    for (int a = threadIdx.x; a < NA; a += 1024) {
        const float v = standup_iou(anc[a], q);
        if (v > best) { best = v; bi = a; }      // ascending a per thread: the first maximum stays
    }
    __shared__ float sv[1024];
    __shared__ int si[1024];
    sv[threadIdx.x] = best;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float ov = sv[threadIdx.x + s];
            const int oi = si[threadIdx.x + s];
            if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { id_highest[k] = si[0]; iou_highest[k] = sv[0]; }
}

__global__ __launch_bounds__(256) void label_assign_kernel(const float4* __restrict__ anc, const float4* __restrict__ gt,
                                                           const double* __restrict__ anchors7, const double* __restrict__ gt7,
                                                           const int* __restrict__ cls, int NA, int n, float pos_thr, float neg_thr,
                                                           const int* __restrict__ id_highest, const float* __restrict__ iou_highest,
                                                           double* __restrict__ pos, double* __restrict__ neg,
                                                           double* __restrict__ targets, long long* __restrict__ cls_labels) {
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= NA) return;
    const float4 b = anc[a];
    int first_pos = -1, first_high = -1;
    bool all_below = true, is_high = false;
    for (int k = 0; k < n; ++k) {
        const float v = standup_iou(b, gt[k]);
        if (v > pos_thr && first_pos < 0) first_pos = k;
        if (!(v < neg_thr)) all_below = false;
        if (id_highest[k] == a && iou_highest[k] > 0.f) {        // :283-285: best anchor of box k, with a positive IoU
            is_high = true;
            if (first_high < 0) first_high = k;
        }
    }
    const int g = first_pos >= 0 ? first_pos : first_high;        // np.where pairs come before the id_highest pairs (:294-299)
    pos[a] = g >= 0 ? 1.0 : 0.0;
    neg[a] = (all_below && !is_high) ? 1.0 : 0.0;              // no ground truth: every anchor is a negative, as :290-294 gives
    cls_labels[a] = g >= 0 ? (long long)cls[g] : 0ll;
    double t[7] = {0, 0, 0, 0, 0, 0, 0};
    if (g >= 0) {
        const double* A = anchors7 + (size_t)a * 7;
        const double* G = gt7 + (size_t)g * 7;
        const double d = sqrt(A[4] * A[4] + A[5] * A[5]);
        t[0] = (G[0] - A[0]) / d;
        t[1] = (G[1] - A[1]) / d;
        t[2] = (G[2] - A[2]) / A[3];
        t[3] = log(G[3] / A[3]);
        t[4] = log(G[4] / A[4]);
        t[5] = log(G[5] / A[5]);
        t[6] = G[6] - A[6];
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) targets[(size_t)a * 7 + j] = t[j];
}

}  // namespace

extern "C" int av2x_generate_label(const float* anchor_standup, const float* gt_standup, const double* anchors7, const double* gt7,
                                   const int32_t* class_ids, int32_t n_anchors, int32_t n_gt, float pos_threshold,
                                   float neg_threshold, void* workspace, double* pos_equal_one, double* neg_equal_one,
                                   double* targets, int64_t* cls_labels, av2x_stream_t stream) {
    if (!anchor_standup || !anchors7 || !pos_equal_one || !neg_equal_one || !targets || !cls_labels)
        return av2x::fail("av2x_generate_label: null argument");
    if (n_anchors <= 0 || n_gt < 0) return av2x::fail("av2x_generate_label: bad sizes");
    if (n_gt > 0 && (!gt_standup || !gt7 || !class_ids || !workspace)) return av2x::fail("av2x_generate_label: null ground truth");
    hipStream_t st = av2x::as_stream(stream);
    int* idh = reinterpret_cast<int*>(workspace);
    float* ioh = reinterpret_cast<float*>(idh + (n_gt > 0 ? n_gt : 1));
    if (n_gt > 0)
        hipLaunchKernelGGL(label_highest_kernel, dim3(n_gt), dim3(1024), 0, st, reinterpret_cast<const float4*>(anchor_standup),
                           reinterpret_cast<const float4*>(gt_standup), n_anchors, idh, ioh);
    hipLaunchKernelGGL(label_assign_kernel, dim3((n_anchors + 255) / 256), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(anchor_standup), reinterpret_cast<const float4*>(gt_standup), anchors7, gt7,
                       class_ids, n_anchors, n_gt, pos_threshold, neg_threshold, idh, ioh, pos_equal_one, neg_equal_one, targets,
                       reinterpret_cast<long long*>(cls_labels));
    return av2x::check_launch("av2x_generate_label");
}
