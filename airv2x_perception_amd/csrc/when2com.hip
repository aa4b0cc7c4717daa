// When2com fusion (models/when2com_modules/when2com.py) pieces that are not convolutions:
//
//   linear_rows_kernel   y (M,N) = act(x (M,K) . w (N,K)^T + b) for a HANDFUL of rows M (one per agent): the
//                        km_generator MLPs (:283-297).  Their first layer has K = 256 * H/4 * W/4 = 563 200 inputs:
//                        577 MB of fp32 weights streamed once per frame -> purely HBM-bound (0.5 flop / byte at
//                        M = 1, 4 at M = 8).  A workgroup owns 8 output rows x one K-slice: every weight element
//                        is read exactly once, as 16-byte loads that are contiguous along K; the x slice is
//                        re-read by the 32 row groups out of L2.  Split-K partials go to a scratch buffer and are
//                        reduced in fixed order by linear_rows_finish (deterministic, no atomics).
//   when2com_fuse_kernel softmax over the agents of key_j . q  (MIMOGeneralDotProductAttention :320-348, softmax
//                        over the KEY axis) and out = sum_j p_j * warped_j (:340-347), 16-byte loads.
#include <cstdlib>

#include "av2x_common.hpp"

namespace {

constexpr int kRows = 8;     // output features per workgroup
constexpr int kMaxRowsX = 8; // x rows per pass (the kernel is instantiated for 1 / 2 / 4 / 8)
constexpr int kMaxAgents = 32;

template <int kMaxM>
__global__ __launch_bounds__(256) void linear_rows_kernel(const float* __restrict__ x, const float* __restrict__ w, int M,
                                                          int N, int K, int kslice, float* __restrict__ part) {
    const int n0 = blockIdx.x * kRows;
    const int k0 = blockIdx.y * kslice;
    const int k1 = min(K, k0 + kslice);
    float acc[kRows][kMaxM];
#pragma unroll
    for (int r = 0; r < kRows; ++r)
#pragma unroll
        for (int m = 0; m < kMaxM; ++m) acc[r][m] = 0.f;
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int k = k0 + 4 * (int)threadIdx.x; k < k1; k += 1024) {
        // all loads of the iteration are issued before the first use: 8 weight rows (HBM, read once -> non-temporal)
        // and the x rows (L2 resident, shared by the N/8 row groups)
        f4 wv[kRows];
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            const int row = n0 + r < N ? n0 + r : N - 1;
            wv[r] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(w + (size_t)row * K + k));
        }
        float4 xv[kMaxM];
#pragma unroll
        for (int m = 0; m < kMaxM; ++m)
            xv[m] = m < M ? *reinterpret_cast<const float4*>(x + (size_t)m * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
#pragma unroll
            for (int m = 0; m < kMaxM; ++m) {
                acc[r][m] = fmaf(wv[r].x, xv[m].x, acc[r][m]);
                acc[r][m] = fmaf(wv[r].y, xv[m].y, acc[r][m]);
                acc[r][m] = fmaf(wv[r].z, xv[m].z, acc[r][m]);
                acc[r][m] = fmaf(wv[r].w, xv[m].w, acc[r][m]);
            }
        }
    }
    __shared__ float red[4][kRows * kMaxM];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < kRows; ++r)
#pragma unroll
        for (int m = 0; m < kMaxM; ++m) {
            float v = acc[r][m];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0) red[wave][r * kMaxM + m] = v;
        }
    __syncthreads();
    if (threadIdx.x < kRows * kMaxM) {
        const int r = threadIdx.x / kMaxM, m = threadIdx.x % kMaxM;
        if (n0 + r < N && m < M) {
            const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
            part[((size_t)blockIdx.y * M + m) * N + n0 + r] = v;
        }
    }
}

__global__ void linear_rows_finish(const float* __restrict__ part, int nsplit, int M, int N, const float* __restrict__ bias,
                                   int act, float* __restrict__ y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * N) return;
    float v = 0.f;
    for (int s = 0; s < nsplit; ++s) v += part[(size_t)s * M * N + i];
    if (bias) v += bias[i % N];
    if (act == 1) v = fmaxf(v, 0.f);
    y[i] = v;
}

struct AgentMaps {
    const float4* p[kMaxAgents];
};

__global__ __launch_bounds__(256) void when2com_fuse_kernel(const float* __restrict__ keys, const float* __restrict__ q,
                                                            int n, int ks, const AgentMaps agents, size_t hwc4,
                                                            float4* __restrict__ out, float* __restrict__ coef_out) {
    __shared__ float coef[kMaxAgents];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = wave; j < n; j += 4) {           // logits: key_j . q
        float d = 0.f;
        for (int c = lane; c < ks; c += 64) d = fmaf(keys[(size_t)j * ks + c], q[c], d);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) d += __shfl_xor(d, o);
        if (lane == 0) coef[j] = d;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float mx = -INFINITY, l = 0.f;
        for (int j = 0; j < n; ++j) mx = fmaxf(mx, coef[j]);
        for (int j = 0; j < n; ++j) { coef[j] = expf(coef[j] - mx); l += coef[j]; }
        for (int j = 0; j < n; ++j) coef[j] /= l;
    }
    __syncthreads();
    if (blockIdx.x == 0 && (int)threadIdx.x < n && coef_out) coef_out[threadIdx.x] = coef[threadIdx.x];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < hwc4; i += (size_t)gridDim.x * blockDim.x) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < n; ++j) {
            const float4 v = agents.p[j][i];
            const float p = coef[j];
            o.x = fmaf(p, v.x, o.x); o.y = fmaf(p, v.y, o.y); o.z = fmaf(p, v.z, o.z); o.w = fmaf(p, v.w, o.w);
        }
        out[i] = o;
    }
}

int nsplit_for(int N, int K) {
    // 4 workgroups per CU (measured optimum on the 577 MB When2com layer: tools/linrows_bench.py), slices of >= 4 KiB per row
    const int groups = (N + kRows - 1) / kRows;
    static const int target = [] { const char* e = getenv("AV2X_LINROWS_WGS"); return e ? atoi(e) : 1024; }();
    int ns = (target + groups - 1) / groups;
    const int max_ns = (K + 1023) / 1024;
    if (ns > max_ns) ns = max_ns;
    return ns < 1 ? 1 : ns;
}

}  // namespace

extern "C" uint64_t av2x_linear_rows_workspace_bytes(int32_t m, int32_t n, int32_t k) {
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    const int mm = m < kMaxRowsX ? m : kMaxRowsX;
    return (uint64_t)nsplit_for(n, k) * mm * n * sizeof(float);
}

extern "C" int av2x_linear_rows(const float* x, const float* w, const float* bias, int32_t m, int32_t n, int32_t k,
                                int32_t act, float* y, void* workspace, uint64_t workspace_bytes, av2x_stream_t stream) {
    if (m == 0) return 0;
    if (!x || !w || !y || !workspace) return av2x::fail("av2x_linear_rows: null argument");
    if (m < 0 || n <= 0 || k <= 0 || k % 4) return av2x::fail("av2x_linear_rows: bad sizes (m=%d n=%d k=%d; k %% 4 == 0)", m, n, k);
    if (act < 0 || act > 1) return av2x::fail("av2x_linear_rows: activation code %d (0 none, 1 ReLU)", act);
    if (workspace_bytes < av2x_linear_rows_workspace_bytes(m, n, k))
        return av2x::fail("av2x_linear_rows: workspace too small (%llu B)", (unsigned long long)workspace_bytes);
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) % 16)
        return av2x::fail("av2x_linear_rows: x and w must be 16-byte aligned");
    hipStream_t st = av2x::as_stream(stream);
    const int ns = nsplit_for(n, k);
    int kslice = (k + ns - 1) / ns;
    kslice = (kslice + 3) & ~3;
    float* part = static_cast<float*>(workspace);
    for (int m0 = 0; m0 < m; m0 += kMaxRowsX) {   // more than 8 rows: the weights are streamed once per group of 8
        const int mm = (m - m0) < kMaxRowsX ? (m - m0) : kMaxRowsX;
        const dim3 grid((n + kRows - 1) / kRows, ns);
        const float* xs = x + (size_t)m0 * k;
        if (mm == 1) hipLaunchKernelGGL(linear_rows_kernel<1>, grid, dim3(256), 0, st, xs, w, mm, n, k, kslice, part);
        else if (mm == 2) hipLaunchKernelGGL(linear_rows_kernel<2>, grid, dim3(256), 0, st, xs, w, mm, n, k, kslice, part);
        else if (mm <= 4) hipLaunchKernelGGL(linear_rows_kernel<4>, grid, dim3(256), 0, st, xs, w, mm, n, k, kslice, part);
        else hipLaunchKernelGGL(linear_rows_kernel<8>, grid, dim3(256), 0, st, xs, w, mm, n, k, kslice, part);
        hipLaunchKernelGGL(linear_rows_finish, dim3((mm * n + 255) / 256), dim3(256), 0, st, part, ns, mm, n, bias, act,
                           y + (size_t)m0 * n);
    }
    return av2x::check_launch("linear_rows_kernel");
}

extern "C" int av2x_when2com_fuse(const float* keys, const float* query, int32_t n_agents, int32_t key_size,
                                  const float* const* agents, uint64_t elems_per_agent, float* out, float* coef,
                                  av2x_stream_t stream) {
    if (!keys || !query || !agents || !out) return av2x::fail("av2x_when2com_fuse: null argument");
    if (n_agents < 1 || n_agents > kMaxAgents) return av2x::fail("av2x_when2com_fuse: n_agents=%d outside [1,%d]", n_agents, kMaxAgents);
    if (key_size <= 0 || elems_per_agent == 0 || elems_per_agent % 4)
        return av2x::fail("av2x_when2com_fuse: bad sizes (elements per agent must be a multiple of 4)");
    const size_t hwc4 = elems_per_agent / 4;
    size_t blocks = (hwc4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    AgentMaps am;
    for (int j = 0; j < kMaxAgents; ++j) {
        am.p[j] = j < n_agents ? reinterpret_cast<const float4*>(agents[j]) : nullptr;
        if (j < n_agents && (!agents[j] || reinterpret_cast<uintptr_t>(agents[j]) % 16))
            return av2x::fail("av2x_when2com_fuse: agent map %d is null or not 16-byte aligned", j);
    }
    hipLaunchKernelGGL(when2com_fuse_kernel, dim3((unsigned)blocks), dim3(256), 0, av2x::as_stream(stream), keys, query, n_agents,
                       key_size, am, hwc4, reinterpret_cast<float4*>(out), coef);
    return av2x::check_launch("when2com_fuse_kernel");
}
