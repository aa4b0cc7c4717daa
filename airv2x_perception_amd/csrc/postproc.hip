// On-device detection post-processing, replacing the torch + shapely host loop
// of VoxelPostprocessor.post_process_airv2x (data_utils/post_processor/voxel_postprocessor.py:666-839)
// and box_utils.nms_rotated (utils/box_utils.py:823-868):
//
//   pp_flag      per anchor : objectness = sigmoid(obj), flag = objectness > obj_threshold
//   pp_compact   order-preserving compaction of flagged anchors (masked_select order) over the whole grid
//   pp_decode    per candidate: class label, delta->box (delta_to_boxes3d :585-634), 8 corners
//                (boxes_to_corners_3d), projection by T, size / z-range keep flag
//   pp_scan2     compaction of kept candidates
//   pp_rank      rank by score (descending, ties: higher index first = argsort()[::-1]), top-K order
//   pp_iou_mask  rotated-quad IoU > threshold bit matrix over the ordered top-K (fp64 Sutherland-Hodgman)
//   pp_greedy    1 workgroup: the sequential greedy pick over the bit matrix (rows staged in LDS)
//   pp_final     range filter (all 8 corners inside [xmin,xmax]x[ymin,ymax]) + compaction, gather outputs
//
// Every count stays on the device; the host reads back one integer (n_out) at the end.
#include "av2x_common.hpp"
#include "block_scan.hpp"

namespace {

struct PPParams {
    int H, W, A, C;
    float obj_thr, nms_thr;
    float zmin, zmax, xmin, xmax, ymin, ymax;
    float T[16];
    const float* Tdev;  // optional: the 4x4 transform in device memory (read instead of T; no host read of a device tensor)
    int order_hwl, top;
};

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// 256 threads per workgroup; also counts the flags of the workgroup (wg_count) for the order-preserving compaction
__global__ __launch_bounds__(256) void pp_flag(const float* __restrict__ obj, PPParams p, float* __restrict__ score,
                                               int* __restrict__ flag, int* __restrict__ wg_count) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int NA = p.H * p.W * p.A;
    bool f = false;
    if (n < NA) {
        const int a = n % p.A, hw = n / p.A;
        const float s = sigmoidf(obj[(size_t)a * p.H * p.W + hw]);  // obj is (1,A,H,W); n = (h*W+w)*A + a
        score[n] = s;
        f = s > p.obj_thr;
        flag[n] = f;
    }
    __shared__ int wc[4];
    const unsigned long long bal = __ballot(f);
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) wg_count[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}

// Order-preserving compaction of the flagged anchors over the whole grid: a workgroup's offset is the sum of the counts of
// the workgroups before it (<= a few hundred ints, summed by the workgroup itself), positions inside come from ballots.
__global__ __launch_bounds__(256) void pp_compact(const int* __restrict__ flag, int n, const int* __restrict__ wg_count,
                                                  int* __restrict__ idx_out, int* __restrict__ count) {
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
    if (f) idx_out[off + __popcll(bal & ((1ull << lane) - 1ull))] = i;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
        count[0] = part[0] + part[1] + part[2] + part[3] + wc[0] + wc[1] + wc[2] + wc[3];
}

__global__ void pp_decode(const float* __restrict__ psm, const float* __restrict__ rm, const float* __restrict__ anchors,
                          const float* __restrict__ score, const int* __restrict__ cand, const int* __restrict__ ncand,
                          PPParams p, float* __restrict__ boxes, float* __restrict__ corners, float* __restrict__ cscore,
                          int* __restrict__ label, int* __restrict__ keep) {
    const int K = ncand[0];
    const int HW = p.H * p.W;
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = p.Tdev ? p.Tdev[i] : p.T[i];
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < K; k += gridDim.x * blockDim.x) {
        const int n = cand[k];
        const int a = n % p.A, hw = n / p.A;
        // class label: psm viewed (C, A, H, W) -> channel c*A + a; classes 1..C-1, first maximum wins
        int best = 1;
        float bv = sigmoidf(psm[(size_t)(1 * p.A + a) * HW + hw]);
        for (int c = 2; c < p.C; ++c) {
            const float v = sigmoidf(psm[(size_t)(c * p.A + a) * HW + hw]);
            if (v > bv) { bv = v; best = c; }
        }
        label[k] = best;
        cscore[k] = score[n];
        float d[7], an[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            d[j] = rm[(size_t)(a * 7 + j) * HW + hw];  // rm (1, A*7, H, W)
            an[j] = anchors[(size_t)n * 7 + j];
        }
        const float diag = sqrtf(__fadd_rn(__fmul_rn(an[4], an[4]), __fmul_rn(an[5], an[5])));
        float b[7];
        b[0] = __fadd_rn(__fmul_rn(d[0], diag), an[0]);
        b[1] = __fadd_rn(__fmul_rn(d[1], diag), an[1]);
        b[2] = __fadd_rn(__fmul_rn(d[2], an[3]), an[2]);
        b[3] = __fmul_rn(expf(d[3]), an[3]);
        b[4] = __fmul_rn(expf(d[4]), an[4]);
        b[5] = __fmul_rn(expf(d[5]), an[5]);
        b[6] = __fadd_rn(d[6], an[6]);
#pragma unroll
        for (int j = 0; j < 7; ++j) boxes[(size_t)k * 7 + j] = b[j];
        // corners: dims (l, w, h) = hwl reordered (box_utils.py:239-240), template / 2, rotate about z
        const float l = p.order_hwl ? b[5] : b[3], w = b[4], h = p.order_hwl ? b[3] : b[5];
        const float ca = cosf(b[6]), sa = sinf(b[6]);
        const float tx[8] = {1, 1, -1, -1, 1, 1, -1, -1}, ty[8] = {-1, 1, 1, -1, -1, 1, 1, -1}, tz[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
        float xmn = INFINITY, xmx = -INFINITY, ymn = INFINITY, ymx = -INFINITY, zmn = INFINITY, zmx = -INFINITY;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float px = __fmul_rn(l, tx[c] * 0.5f), py = __fmul_rn(w, ty[c] * 0.5f), pz = __fmul_rn(h, tz[c] * 0.5f);
            // points @ [[cos, sin, 0], [-sin, cos, 0], [0, 0, 1]]
            const float rx = __fadd_rn(__fmul_rn(px, ca), __fmul_rn(py, -sa)) + b[0];
            const float ry = __fadd_rn(__fmul_rn(px, sa), __fmul_rn(py, ca)) + b[1];
            const float rz = pz + b[2];
            // T @ [x y z 1]^T
            const float qx = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[0], rx), __fmul_rn(T[1], ry)), __fmul_rn(T[2], rz)), T[3]);
            const float qy = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[4], rx), __fmul_rn(T[5], ry)), __fmul_rn(T[6], rz)), T[7]);
            const float qz = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[8], rx), __fmul_rn(T[9], ry)), __fmul_rn(T[10], rz)), T[11]);
            corners[(size_t)k * 24 + c * 3 + 0] = qx;
            corners[(size_t)k * 24 + c * 3 + 1] = qy;
            corners[(size_t)k * 24 + c * 3 + 2] = qz;
            xmn = fminf(xmn, qx); xmx = fmaxf(xmx, qx);
            ymn = fminf(ymn, qy); ymx = fmaxf(ymx, qy);
            zmn = fminf(zmn, qz); zmx = fmaxf(zmx, qz);
        }
        // remove_large_pred_bbx (z_len used as a truth value, box_utils.py:1011-1012) & remove_bbx_abnormal_z
        const bool k1 = (xmx - xmn) <= 6.f && (ymx - ymn) <= 6.f && (zmx - zmn) != 0.f;
        const bool k2 = zmn >= p.zmin && zmx <= p.zmax;
        keep[k] = (k1 && k2) ? 1 : 0;
    }
}

__global__ __launch_bounds__(1024) void pp_scan2(const int* __restrict__ keep, const int* __restrict__ ncand,
                                                 const float* __restrict__ cscore, int* __restrict__ kept,
                                                 float* __restrict__ kscore, int* __restrict__ nkept) {
    __shared__ int tot;
    const int K = ncand[0];
    av2x::block_scan(
        K, [&](int i) { return keep[i]; },
        [&](int i, int ex) { if (keep[i]) { kept[ex] = i; kscore[ex] = cscore[i]; } }, &tot);
    __syncthreads();
    if (threadIdx.x == 0) nkept[0] = tot;
}

// Above PP_RANK_DIRECT kept candidates (an untrained / early-epoch model or a low obj_threshold can put all H*W*A anchors
// above the threshold) the O(K2^2) ranking below would take seconds, so pp_select first finds the exact score of rank
// `top` -- a bisection over the fp32 bit patterns (scores are sigmoids > 0: their unsigned bit patterns sort like the
// values): the largest key `cut` with |{key >= cut}| >= top, hence |{key > cut}| < top -- and places the candidates whose
// score EQUALS the cut itself: rank = |{key > cut}| + (ties at a higher position), from one scan.  pp_rank then ranks only
// the < top candidates above the cut, each against all K2.  Result identical to the full ranking
// (scores.argsort(descending) of box_utils.py:846-849 with ties -> higher position first).
constexpr int PP_RANK_DIRECT = 4096;

__global__ __launch_bounds__(1024) void pp_select(const float* __restrict__ kscore, const int* __restrict__ nkept, int top,
                                                  int* __restrict__ order, unsigned* __restrict__ cutinfo) {
    const int K2 = nkept[0];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (K2 <= PP_RANK_DIRECT) {  // every candidate is ranked directly (positive scores: key > 0)
        if (tid == 0) { cutinfo[0] = 0u; cutinfo[1] = 0u; }
        return;
    }
    __shared__ int red[16];
    __shared__ int tot;
    auto count_ge = [&](unsigned key) {
        int c = 0;
        for (int i = tid; i < K2; i += 1024) c += __float_as_uint(kscore[i]) >= key;
        for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o);
        __syncthreads();  // red[] of the previous round has been read by everyone
        if (lane == 0) red[wave] = c;
        __syncthreads();
        int t = 0;
        for (int k = 0; k < 16; ++k) t += red[k];
        return t;
    };
    unsigned lo = 0u, hi = 0x7f800000u;  // |{key >= lo}| >= top (K2 > top), |{key >= +inf}| = 0 < top
    while (hi - lo > 1u) {
        const unsigned mid = lo + ((hi - lo) >> 1);
        if (count_ge(mid) >= top) lo = mid; else hi = mid;
    }
    const unsigned cut = lo;
    const int c_gt = count_ge(cut + 1u), c_ge = count_ge(cut);
    const int c_eq = c_ge - c_gt;
    __syncthreads();
    av2x::block_scan<16>(
        K2, [&](int i) { return __float_as_uint(kscore[i]) == cut ? 1 : 0; },
        [&](int i, int ex) {
            if (__float_as_uint(kscore[i]) == cut) {
                const int r = c_gt + (c_eq - 1 - ex);  // ties: the higher position ranks first
                if (r < top) order[r] = i;
            }
        },
        &tot);
    if (tid == 0) { cutinfo[0] = cut; cutinfo[1] = (unsigned)c_gt; }
}

// order[r] = position (in the kept list) of the r-th best score; ties -> higher position first
// kscore: scores of the kept candidates, compacted (every lane reads the same kscore[j]: one broadcast load)
__global__ void pp_rank(const float* __restrict__ kscore, const int* __restrict__ nkept, int top, int* __restrict__ order,
                        int* __restrict__ ntop, const unsigned* __restrict__ cutinfo) {
    const int K2 = nkept[0];
    const unsigned cut = cutinfo[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) ntop[0] = K2 < top ? K2 : top;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < K2; i += gridDim.x * blockDim.x) {
        const float si = kscore[i];
        if (__float_as_uint(si) <= cut) continue;  // at or below the rank-`top` score: placed by pp_select or cut off
        int r = 0;
        for (int j = 0; j < K2; ++j) {
            const float sj = kscore[j];
            r += (sj > si) || (sj == si && j > i);
        }
        if (r < top) order[r] = i;
    }
}

struct P2 { double x, y; };

__device__ int clip_edge(const P2* in, int n, P2 a, P2 b, P2* out) {
    int m = 0;
    const double ex = b.x - a.x, ey = b.y - a.y;
    for (int i = 0; i < n; ++i) {
        const P2 cur = in[i], nxt = in[(i + 1 == n) ? 0 : i + 1];
        const double dc = ex * (cur.y - a.y) - ey * (cur.x - a.x);
        const double dn = ex * (nxt.y - a.y) - ey * (nxt.x - a.x);
        if (dc >= 0.0) out[m++] = cur;
        if ((dc >= 0.0) != (dn >= 0.0)) {
            const double t = dc / (dc - dn);
            out[m].x = cur.x + t * (nxt.x - cur.x);
            out[m].y = cur.y + t * (nxt.y - cur.y);
            ++m;
        }
    }
    return m;
}

__device__ double poly_area(const P2* p, int n) {
    double a = 0.0;
    for (int i = 0; i < n; ++i) {
        const P2 u = p[i], v = p[(i + 1 == n) ? 0 : i + 1];
        a += u.x * v.y - v.x * u.y;
    }
    return 0.5 * a;
}

// Exact early-out: quads whose axis-aligned bounding boxes are strictly separated cannot intersect (IoU = 0); the
// comparison is on the fp32 corners themselves, so it never changes a result.  Most of the K^2 pairs of a frame end here.
__device__ __forceinline__ bool quads_separated(const float* ca, const float* cb) {
    float ax0 = ca[0], ax1 = ca[0], ay0 = ca[1], ay1 = ca[1], bx0 = cb[0], bx1 = cb[0], by0 = cb[1], by1 = cb[1];
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        ax0 = fminf(ax0, ca[i * 3]); ax1 = fmaxf(ax1, ca[i * 3]); ay0 = fminf(ay0, ca[i * 3 + 1]); ay1 = fmaxf(ay1, ca[i * 3 + 1]);
        bx0 = fminf(bx0, cb[i * 3]); bx1 = fmaxf(bx1, cb[i * 3]); by0 = fminf(by0, cb[i * 3 + 1]); by1 = fmaxf(by1, cb[i * 3 + 1]);
    }
    return ax1 < bx0 || bx1 < ax0 || ay1 < by0 || by1 < ay0;
}

__device__ double quad_iou(const float* ca, const float* cb) {
    if (quads_separated(ca, cb)) return 0.0;
    P2 a[4], b[4], b1[12], b2[12];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i].x = ca[i * 3]; a[i].y = ca[i * 3 + 1];
        b[i].x = cb[i * 3]; b[i].y = cb[i * 3 + 1];
    }
    double aa = poly_area(a, 4), ab = poly_area(b, 4);
    if (aa < 0) { const P2 t = a[1]; a[1] = a[3]; a[3] = t; aa = -aa; }
    if (ab < 0) { const P2 t = b[1]; b[1] = b[3]; b[3] = t; ab = -ab; }
    int n = 4;
    P2* src = b1; P2* dst = b2;
    for (int i = 0; i < 4; ++i) src[i] = a[i];
    for (int e = 0; e < 4 && n > 0; ++e) {
        n = clip_edge(src, n, b[e], b[(e + 1) & 3], dst);
        P2* t = src; src = dst; dst = t;
    }
    const double inter = n >= 3 ? fabs(poly_area(src, n)) : 0.0;
    const double uni = aa + ab - inter;
    return uni > 0.0 ? inter / uni : 0.0;
}

// bit (i, j) for j > i : iou(order[i], order[j]) > thr (rounded to fp32 first, common_utils.py:174)
__global__ void pp_iou_mask(const float* __restrict__ corners, const int* __restrict__ kept, const int* __restrict__ order,
                            const int* __restrict__ ntop, float thr, int words, unsigned long long* __restrict__ mask) {
    const int m = ntop[0];
    const int i = blockIdx.y;
    const int jw = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (row, column): 64 threads = one word
    const int j = jw;
    if (i >= m) return;
    bool bit = false;
    if (j < m && j > i) {
        const float* ca = corners + (size_t)kept[order[i]] * 24;
        const float* cb = corners + (size_t)kept[order[j]] * 24;
        bit = (float)quad_iou(ca, cb) > thr;
    }
    const unsigned long long bal = __ballot(bit);
    if ((threadIdx.x & 63) == 0 && (j >> 6) < words) mask[(size_t)i * words + (j >> 6)] = bal;
}

// Greedy suppression over the precomputed bit matrix: box i is picked iff no earlier pick overlaps it; a pick ORs its row
// into `removed`.  Inherently sequential, so ONE wave walks the boxes -- but in blocks of 64 so that the serial chain never
// touches memory: lane j loads the DIAGONAL word of box 64b+j (its overlaps inside the block), the block is resolved by a
// scalar loop over the surviving boxes (v_readlane of the diagonal word + bit ops, ~20 cycles per pick), the picks of the
// block are written in order by their own lanes (rank = popcount of the kept bits below), and only then the full rows of
// the kept boxes are ORed into `removed` (lane w holds word w) with independent, pipelined reads.  Identical to the
// box-by-box walk: inside a block the order is preserved, across blocks `removed` is complete before the next block starts.
// Rows come from LDS when they fit (STAGED; copied in by the whole 1024-thread workgroup).
template <bool STAGED>
__global__ __launch_bounds__(1024) void pp_greedy(const unsigned long long* __restrict__ mask, const int* __restrict__ ntop,
                                                  int words, int* __restrict__ pick, int* __restrict__ npick) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long rows[];  // [m][words] when STAGED
    const int m = ntop[0];
    const int lane = threadIdx.x;
    if (STAGED) {
        const int total = m * words;
        for (int i0 = threadIdx.x; i0 < total; i0 += 4 * (int)blockDim.x) {       // four loads in flight per thread
            unsigned long long v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = (i0 + u * (int)blockDim.x) < total ? mask[i0 + u * (int)blockDim.x] : 0ull;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if ((i0 + u * (int)blockDim.x) < total) rows[i0 + u * (int)blockDim.x] = v[u];
        }
        __syncthreads();
    }
    if (threadIdx.x >= 64) return;
    auto word_of = [&](int box, int w) -> unsigned long long {
        return STAGED ? rows[(size_t)box * words + w] : mask[(size_t)box * words + w];
    };
    auto lane64 = [&](unsigned long long v, int l) -> unsigned long long {
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, l);
        const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), l);
        return ((unsigned long long)hi << 32) | lo;
    };
    unsigned long long removed = 0ull;   // word `lane` of the removed set
    int np = 0;
    for (int b = 0; b * 64 < m; ++b) {
        const int base = b * 64;
        const int cnt = m - base < 64 ? m - base : 64;
        const unsigned long long diag = (lane < cnt) ? word_of(base + lane, b) : 0ull;
        unsigned long long alive = ~lane64(removed, b);                     // uniform from here on
        if (cnt < 64) alive &= (1ull << cnt) - 1ull;
        unsigned long long kept = 0ull;
        while (alive) {
            const int j = __builtin_ctzll(alive);
            const unsigned long long bit = 1ull << j;
            kept |= bit;
            alive &= ~(lane64(diag, j) | bit);
        }
        if (lane < cnt && ((kept >> lane) & 1ull)) pick[np + __builtin_popcountll(kept & ((1ull << lane) - 1ull))] = base + lane;
        np += __builtin_popcountll(kept);
        if (lane < words) {
            unsigned long long k = kept;
            while (k) {                                   // four independent reads in flight per trip
                unsigned long long acc[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (k) { acc[u] = word_of(base + __builtin_ctzll(k), lane); k &= k - 1ull; }
                removed |= (acc[0] | acc[1]) | (acc[2] | acc[3]);
            }
        }
    }
    if (lane == 0) npick[0] = np;
}

__global__ __launch_bounds__(1024) void pp_final(const float* __restrict__ boxes, const float* __restrict__ corners,
                                                 const float* __restrict__ cscore, const int* __restrict__ label,
                                                 const int* __restrict__ kept, const int* __restrict__ order,
                                                 const int* __restrict__ pick, const int* __restrict__ npick, PPParams p,
                                                 int* __restrict__ inr, float* __restrict__ out_corners,
                                                 float* __restrict__ out_scores, int* __restrict__ out_labels,
                                                 float* __restrict__ out_boxes, int* __restrict__ out_index,
                                                 const int* __restrict__ cand, int* __restrict__ nout) {
    __shared__ int tot;
    const int np = npick[0];
    for (int q = threadIdx.x; q < np; q += blockDim.x) {
        const float* c = corners + (size_t)kept[order[pick[q]]] * 24;
        bool ok = true;
        for (int v = 0; v < 8; ++v)
            ok = ok && c[v * 3] >= p.xmin && c[v * 3] <= p.xmax && c[v * 3 + 1] >= p.ymin && c[v * 3 + 1] <= p.ymax;
        inr[q] = ok;
    }
    __syncthreads();
    av2x::block_scan(
        np, [&](int q) { return inr[q]; },
        [&](int q, int ex) {
            if (!inr[q]) return;
            const int k = kept[order[pick[q]]];
            for (int v = 0; v < 24; ++v) out_corners[(size_t)ex * 24 + v] = corners[(size_t)k * 24 + v];
            for (int v = 0; v < 7; ++v) out_boxes[(size_t)ex * 7 + v] = boxes[(size_t)k * 7 + v];
            out_scores[ex] = cscore[k];
            out_labels[ex] = label[k];
            out_index[ex] = cand[k];
        },
        &tot);
    __syncthreads();
    if (threadIdx.x == 0) nout[0] = tot;
}

// ---- AP evaluation (utils/eval_utils_opv2v.py:41-97) --------------------------------------------
// iou[d][g] = fp32(IoU(det[order[d]], gt[g])) (common_utils.compute_iou :150-171 returns float32)
__global__ void eval_iou_kernel(const float* __restrict__ det, const int* __restrict__ order, int D,
                                const float* __restrict__ gt, int G, float* __restrict__ iou) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D * G) return;
    const int d = i / G, g = i - d * G;
    iou[i] = (float)quad_iou(det + (size_t)order[d] * 24, gt + (size_t)g * 24);
}

// Greedy matching in score order: a detection is a TP when its best IoU over the GT boxes still
// unmatched is >= thr (`np.max(ious) < iou_thresh` -> FP, :76-79); the matched GT is the first arg-max
// (np.argmax, :84) and leaves the list (:85).  One workgroup; the loop over detections is sequential.
__global__ __launch_bounds__(256) void eval_match_kernel(const float* __restrict__ iou, int D, int G, float thr,
                                                         int* __restrict__ tp, int* __restrict__ matched_gt) {
    extern __shared__ unsigned char alive[];  // [G]
    __shared__ float bv[256];
    __shared__ int bi[256];
    for (int g = threadIdx.x; g < G; g += 256) alive[g] = 1;
    __syncthreads();
    for (int d = 0; d < D; ++d) {
        float best = -1.f;
        int arg = 0x7fffffff;
        for (int g = threadIdx.x; g < G; g += 256) {
            if (!alive[g]) continue;
            const float v = iou[(size_t)d * G + g];
            if (v > best) { best = v; arg = g; }  // ascending g inside a thread: first maximum wins
        }
        bv[threadIdx.x] = best; bi[threadIdx.x] = arg;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) {
                const float ov = bv[threadIdx.x + s];
                const int oi = bi[threadIdx.x + s];
                if (ov > bv[threadIdx.x] || (ov == bv[threadIdx.x] && oi < bi[threadIdx.x])) { bv[threadIdx.x] = ov; bi[threadIdx.x] = oi; }
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const bool hit = bi[0] != 0x7fffffff && !(bv[0] < thr);
            tp[d] = hit ? 1 : 0;
            matched_gt[d] = hit ? bi[0] : -1;
            if (hit) alive[bi[0]] = 0;
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int av2x_eval_tp_fp(const float* det_corners, const int32_t* order, int32_t n_det, const float* gt_corners,
                               int32_t n_gt, float iou_thresh, float* iou_ws, int32_t* tp, int32_t* matched_gt,
                               av2x_stream_t stream) {
    if (n_det < 0 || n_gt < 0) return av2x::fail("av2x_eval_tp_fp: negative count");
    if (n_det == 0) return 0;
    if (!det_corners || !order || !tp || !matched_gt || (n_gt > 0 && (!gt_corners || !iou_ws)))
        return av2x::fail("av2x_eval_tp_fp: null argument");
    if (n_gt > 60000) return av2x::fail("av2x_eval_tp_fp: more than 60000 ground-truth boxes");
    hipStream_t st = av2x::as_stream(stream);
    const long long cells = (long long)n_det * n_gt;
    if (cells >= (1ll << 31)) return av2x::fail("av2x_eval_tp_fp: IoU matrix too large");
    if (n_gt > 0)
        hipLaunchKernelGGL(eval_iou_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, det_corners, order, n_det,
                           gt_corners, n_gt, iou_ws);
    hipLaunchKernelGGL(eval_match_kernel, dim3(1), dim3(256), (size_t)(n_gt > 0 ? n_gt : 1), st, iou_ws, n_det, n_gt,
                       iou_thresh, tp, matched_gt);
    return av2x::check_launch("av2x_eval_tp_fp");
}

extern "C" uint64_t av2x_postprocess_workspace_bytes(int32_t h, int32_t w, int32_t a, int32_t top) {
    const uint64_t na = (uint64_t)h * w * a;
    const uint64_t words = ((uint64_t)top + 63) / 64;
    // score, flag, cand, boxes(7), corners(24), cscore, label, keep, kept | order, pick, inr (top each) | per-workgroup
    // flag counts | mask | 8 counters
    return (na * (1 + 1 + 1 + 7 + 24 + 1 + 1 + 1 + 1 + 1) + 3 * (uint64_t)top + (na / 256 + 2) + 16) * 4 + (uint64_t)top * words * 8 + 64;
}

static int postprocess_impl(const float* psm, const float* rm, const float* obj, const float* anchors, int32_t h,
                            int32_t w, int32_t a, int32_t c, const float* transform16, const float* transform16_dev,
                            const float* range6, float obj_threshold, float nms_threshold, int32_t order_hwl, int32_t top,
                            void* workspace, float* out_corners, float* out_scores, int32_t* out_labels, float* out_boxes,
                            int32_t* out_index, int32_t* counts, av2x_stream_t stream) {
    if (!psm || !rm || !obj || !anchors || (!transform16 && !transform16_dev) || !range6 || !workspace || !out_corners || !out_scores ||
        !out_labels || !out_boxes || !out_index || !counts)
        return av2x::fail("av2x_postprocess: null argument");
    if (h <= 0 || w <= 0 || a <= 0 || c < 2 || top <= 0 || top > 4096) return av2x::fail("av2x_postprocess: bad sizes");
    PPParams p;
    p.H = h; p.W = w; p.A = a; p.C = c;
    p.obj_thr = obj_threshold; p.nms_thr = nms_threshold;
    p.xmin = range6[0]; p.ymin = range6[1]; p.zmin = range6[2]; p.xmax = range6[3]; p.ymax = range6[4]; p.zmax = range6[5];
    for (int i = 0; i < 16; ++i) p.T[i] = transform16 ? transform16[i] : 0.f;
    p.Tdev = transform16_dev;
    p.order_hwl = order_hwl; p.top = top;
    const int NA = h * w * a;
    const int words = (top + 63) / 64;
    hipStream_t st = av2x::as_stream(stream);
    float* f = reinterpret_cast<float*>(workspace);
    float* score = f; f += NA;
    int* flag = reinterpret_cast<int*>(f); f += NA;
    int* cand = reinterpret_cast<int*>(f); f += NA;
    float* boxes = f; f += (size_t)NA * 7;
    float* corners = f; f += (size_t)NA * 24;
    float* cscore = f; f += NA;
    int* label = reinterpret_cast<int*>(f); f += NA;
    int* keep = reinterpret_cast<int*>(f); f += NA;
    int* kept = reinterpret_cast<int*>(f); f += NA;
    float* kscore = f; f += NA;
    int* order = reinterpret_cast<int*>(f); f += top;
    int* pick = reinterpret_cast<int*>(f); f += top;
    int* inr = reinterpret_cast<int*>(f); f += top;
    int* wg_count = reinterpret_cast<int*>(f); f += NA / 256 + 2;
    unsigned* cutinfo = reinterpret_cast<unsigned*>(f); f += 16;
    unsigned long long* mask = reinterpret_cast<unsigned long long*>((reinterpret_cast<uintptr_t>(f) + 15) & ~uintptr_t(15));
    // counts[0..4] = candidates, after size/z filters, NMS input (top), NMS picks, final
    int *ncand = counts, *nkept = counts + 1, *ntop = counts + 2, *npick = counts + 3, *nout = counts + 4;

    hipLaunchKernelGGL(pp_flag, dim3((NA + 255) / 256), dim3(256), 0, st, obj, p, score, flag, wg_count);
    hipLaunchKernelGGL(pp_compact, dim3((NA + 255) / 256), dim3(256), 0, st, flag, NA, wg_count, cand, ncand);
    hipLaunchKernelGGL(pp_decode, dim3(256), dim3(256), 0, st, psm, rm, anchors, score, cand, ncand, p, boxes, corners,
                       cscore, label, keep);
    hipLaunchKernelGGL(pp_scan2, dim3(1), dim3(1024), 0, st, keep, ncand, cscore, kept, kscore, nkept);
    hipLaunchKernelGGL(pp_select, dim3(1), dim3(1024), 0, st, kscore, nkept, top, order, cutinfo);
    hipLaunchKernelGGL(pp_rank, dim3(256), dim3(256), 0, st, kscore, nkept, top, order, ntop, cutinfo);
    hipLaunchKernelGGL(pp_iou_mask, dim3((top + 63) / 64, top), dim3(64), 0, st, corners, kept, order, ntop, nms_threshold,
                       words, mask);
    const size_t lds = (size_t)top * words * 8 <= 128 * 1024 ? (size_t)top * words * 8 : 0;
    static av2x::LdsLimit lds_limit;
    lds_limit.ensure(reinterpret_cast<const void*>(&pp_greedy<true>), 128 * 1024);
    if (lds > 0) hipLaunchKernelGGL(pp_greedy<true>, dim3(1), dim3(1024), lds, st, mask, ntop, words, pick, npick);
    else hipLaunchKernelGGL(pp_greedy<false>, dim3(1), dim3(1024), 0, st, mask, ntop, words, pick, npick);
    hipLaunchKernelGGL(pp_final, dim3(1), dim3(1024), 0, st, boxes, corners, cscore, label, kept, order, pick, npick, p, inr,
                       out_corners, out_scores, out_labels, out_boxes, out_index, cand, nout);
    return av2x::check_launch("av2x_postprocess");
}

extern "C" int av2x_postprocess(const float* psm, const float* rm, const float* obj, const float* anchors, int32_t h,
                                int32_t w, int32_t a, int32_t c, const float* transform16, const float* range6,
                                float obj_threshold, float nms_threshold, int32_t order_hwl, int32_t top, void* workspace,
                                float* out_corners, float* out_scores, int32_t* out_labels, float* out_boxes,
                                int32_t* out_index, int32_t* counts, av2x_stream_t stream) {
    return postprocess_impl(psm, rm, obj, anchors, h, w, a, c, transform16, nullptr, range6, obj_threshold, nms_threshold,
                            order_hwl, top, workspace, out_corners, out_scores, out_labels, out_boxes, out_index, counts, stream);
}

extern "C" int av2x_postprocess_devt(const float* psm, const float* rm, const float* obj, const float* anchors, int32_t h,
                                     int32_t w, int32_t a, int32_t c, const float* transform16_dev, const float* range6,
                                     float obj_threshold, float nms_threshold, int32_t order_hwl, int32_t top, void* workspace,
                                     float* out_corners, float* out_scores, int32_t* out_labels, float* out_boxes,
                                     int32_t* out_index, int32_t* counts, av2x_stream_t stream) {
    return postprocess_impl(psm, rm, obj, anchors, h, w, a, c, nullptr, transform16_dev, range6, obj_threshold, nms_threshold,
                            order_hwl, top, workspace, out_corners, out_scores, out_labels, out_boxes, out_index, counts, stream);
}
