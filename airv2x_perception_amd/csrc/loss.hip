// PointPillarLossMultiClass.forward (loss/point_pillar_loss_multiclass.py:96-179) and its gradient with respect to the
// three head maps, in one pass over the anchors (HBM-bound: every input is read once, every gradient written once):
//   conf = cls_weight / B^2 * sum_{b,h,w,a,c} focal(x, onehot(class_id)) / max(npos_b, 1)      (:196-214, :153-156 -- the
//          reference divides by B inside cls_loss_func AND again outside; every anchor carries weight 1: :118-121)
//   reg  = reg_coe / B * sum over positive anchors, 7 codes, of smoothL1_{1/9}(pred' - tgt') / max(npos_b, 1), with the
//          sin-difference encoding of code 6 (:279-293): pred' - tgt' = sin(r) cos(t) - cos(r) sin(t); NaN targets ignored
//   obj  = mean over (b,h,w,a) of -(pos log(sigmoid(o) + 1e-6) + (1 - pos) log(1 - sigmoid(o) + 1e-6))       (:163-168)
// Layouts as the reference hands them over: psm (B, A*C, H, W), rm (B, A*7, H, W), obj (B, A, H, W) NCHW;
// targets (B, H, W, A*7), pos_equal_one (B, H, W, A), class_ids (B, H, W, A).
// Reductions are deterministic: per-workgroup fp64 partial sums, then one workgroup adds them in index order.
#include "av2x_common.hpp"

namespace {

constexpr int kMaxA = 8;

__global__ __launch_bounds__(256) void loss_count_kernel(const float* __restrict__ pos, int hwa, int* __restrict__ npos) {
    const int b = blockIdx.y;
    int local = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hwa; i += gridDim.x * blockDim.x) local += pos[(size_t)b * hwa + i] > 0.f;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) local += __shfl_xor(local, s, 64);
    __shared__ int ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = ws[0] + ws[1] + ws[2] + ws[3];
        if (t) atomicAdd(&npos[b], t);   // integer: order does not matter
    }
}

struct LossParams {
    const float* psm; const float* rm; const float* obj;
    const float* targets; const float* pos; const int* cls;
    const int* npos;
    double* partial;        // [B * blocks_per_sample][3]
    float* dpsm; float* drm; float* dobj;   // nullable
    int B, H, W, A, C;
    float cls_weight, reg_coe;
};

__device__ __forceinline__ float sl1(float d, float beta) {
    const float n = fabsf(d);
    return n < beta ? 0.5f * n * n / beta : n - 0.5f * beta;
}
__device__ __forceinline__ float sl1_grad(float d, float beta) {
    const float n = fabsf(d);
    return n < beta ? d / beta : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
}

__global__ __launch_bounds__(256) void loss_main_kernel(const LossParams p) {
    const int HW = p.H * p.W;
    const int b = blockIdx.y;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    double conf = 0.0, reg = 0.0, objl = 0.0;
    if (pix < HW) {
        const float inv_np = 1.0f / fmaxf((float)p.npos[b], 1.0f);
        const float gconf = p.cls_weight / ((float)p.B * (float)p.B) * inv_np;
        const float greg = p.reg_coe / (float)p.B * inv_np;
        const float gobj = 1.0f / ((float)p.B * (float)HW * (float)p.A);
        const float beta = 1.0f / 9.0f;
        for (int a = 0; a < p.A; ++a) {
            const size_t anc = ((size_t)b * HW + pix) * p.A + a;
            const float pm = p.pos[anc];
            const int cid = p.cls[anc];
            const float wa = (pm > 0.f || pm == 0.f) ? inv_np : 0.f;   // positives + negatives (:118-121); anything else: weight 0
            // ---- focal classification over the C classes of this anchor
            for (int c = 0; c < p.C; ++c) {
                const size_t o = ((size_t)(b * p.A * p.C + a * p.C + c)) * HW + pix;
                const float x = p.psm[o];
                const float t = (c == cid) ? 1.f : 0.f;
                const float s = 1.0f / (1.0f + expf(-x));
                const float aw = t * 0.25f + (1.f - t) * 0.75f;
                const float pt = t * (1.f - s) + (1.f - t) * s;
                const float fw = aw * pt * pt;
                const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
                conf += (double)(fw * bce * wa);
                if (p.dpsm) {
                    const float dpt = (1.f - 2.f * t) * s * (1.f - s);
                    p.dpsm[o] = (wa > 0.f ? gconf : 0.f) * (aw * 2.f * pt * dpt * bce + fw * (s - t));
                }
            }
            // ---- regression (positives only) with the sin-difference encoding of the heading
            const bool is_pos = pm > 0.f;
            for (int k = 0; k < 7; ++k) {
                const size_t o = ((size_t)(b * p.A * 7 + a * 7 + k)) * HW + pix;
                float g = 0.f;
                if (is_pos) {
                    const float r = p.rm[o];
                    float t = p.targets[anc * 7 + k];
                    if (k == 6) {
                        // the reference tests isnan on the ENCODED target cos(r) sin(t): NaN iff t (or r) is NaN
                        const float pe = sinf(r) * cosf(t), te = cosf(r) * sinf(t);
                        if (!isnan(te)) {
                            const float d = pe - te;
                            reg += (double)(sl1(d, beta) * inv_np);
                            g = greg * sl1_grad(d, beta) * (cosf(r) * cosf(t) + sinf(r) * sinf(t));
                        }
                    } else if (!isnan(t)) {
                        const float d = r - t;
                        reg += (double)(sl1(d, beta) * inv_np);
                        g = greg * sl1_grad(d, beta);
                    }
                }
                if (p.drm) p.drm[o] = g;
            }
            // ---- objectness BCE against pos_equal_one
            {
                const size_t o = ((size_t)(b * p.A + a)) * HW + pix;
                const float x = p.obj[o];
                const float s = 1.0f / (1.0f + expf(-x));
                objl += (double)(-(pm * logf(s + 1e-6f) + (1.f - pm) * logf(1.f - s + 1e-6f)));
                if (p.dobj) {
                    const float ds = s * (1.f - s);
                    p.dobj[o] = gobj * -(pm * ds / (s + 1e-6f) - (1.f - pm) * ds / (1.f - s + 1e-6f));
                }
            }
        }
    }
    // deterministic workgroup reduction (fp64), one partial triple per workgroup
    __shared__ double red[3][4];
    double v[3] = {conf, reg, objl};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) v[q] += __shfl_xor(v[q], s, 64);
        if ((threadIdx.x & 63) == 0) red[q][threadIdx.x >> 6] = v[q];
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int q = threadIdx.x;
        p.partial[((size_t)b * gridDim.x + blockIdx.x) * 3 + q] = red[q][0] + red[q][1] + red[q][2] + red[q][3];
    }
}

__global__ void loss_finish_kernel(const double* __restrict__ partial, int n_partials, int B, int HW, int A, float cls_weight,
                                   float reg_coe, float* __restrict__ out4) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double c = 0.0, r = 0.0, o = 0.0;
    for (int i = 0; i < n_partials; ++i) { c += partial[i * 3 + 0]; r += partial[i * 3 + 1]; o += partial[i * 3 + 2]; }
    const float conf = (float)(c / ((double)B * (double)B)) * cls_weight;
    const float reg = (float)(r / (double)B) * reg_coe;
    const float obj = (float)(o / ((double)B * (double)HW * (double)A));
    out4[0] = reg + conf + obj;   // total_loss = reg_loss + conf_loss + obj_loss (:170)
    out4[1] = reg;
    out4[2] = conf;
    out4[3] = obj;
}

}  // namespace

extern "C" uint64_t av2x_pp_loss_workspace_bytes(int32_t b, int32_t h, int32_t w) {
    const uint64_t blocks = ((uint64_t)h * w + 255) / 256;
    return 256 + (uint64_t)b * blocks * 3 * sizeof(double);   // [B] int counters (padded to 256 B) + partial sums
}

extern "C" int av2x_pp_loss(const float* psm, const float* rm, const float* obj, const float* targets, const float* pos_equal_one,
                            const int32_t* class_ids, int32_t b, int32_t h, int32_t w, int32_t a, int32_t c, float cls_weight,
                            float reg_coe, void* workspace, float* out4, float* dpsm, float* drm, float* dobj,
                            av2x_stream_t stream) {
    if (!psm || !rm || !obj || !targets || !pos_equal_one || !class_ids || !workspace || !out4)
        return av2x::fail("av2x_pp_loss: null argument");
    if (b <= 0 || h <= 0 || w <= 0 || a <= 0 || a > kMaxA || c <= 0) return av2x::fail("av2x_pp_loss: bad sizes (A <= %d)", kMaxA);
    if (b > 64) return av2x::fail("av2x_pp_loss: batch %d > 64", b);
    hipStream_t st = av2x::as_stream(stream);
    int* npos = reinterpret_cast<int*>(workspace);
    double* partial = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + 256);
    if (hipMemsetAsync(npos, 0, 256, st) != hipSuccess) return av2x::fail("av2x_pp_loss: memset failed");
    const int hw = h * w, hwa = hw * a;
    int cb = (hwa + 255) / 256;
    if (cb > 1024) cb = 1024;
    hipLaunchKernelGGL(loss_count_kernel, dim3(cb, b), dim3(256), 0, st, pos_equal_one, hwa, npos);
    LossParams p;
    p.psm = psm; p.rm = rm; p.obj = obj; p.targets = targets; p.pos = pos_equal_one; p.cls = class_ids; p.npos = npos;
    p.partial = partial; p.dpsm = dpsm; p.drm = drm; p.dobj = dobj;
    p.B = b; p.H = h; p.W = w; p.A = a; p.C = c; p.cls_weight = cls_weight; p.reg_coe = reg_coe;
    const int blocks = (hw + 255) / 256;
    hipLaunchKernelGGL(loss_main_kernel, dim3(blocks, b), dim3(256), 0, st, p);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, st, partial, b * blocks, b, hw, a, cls_weight, reg_coe, out4);
    return av2x::check_launch("pp_loss kernels");
}
