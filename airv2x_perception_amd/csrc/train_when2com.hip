// Backward kernels of the When2com fusion head (training; forward kernels: when2com.hip).  What the reference gets from torch autograd of
// models/when2com_modules/when2com.py: km_generator's Linear stack (:283-297, the first layer streams 256 * H/4 * W/4 inputs per output
// feature: 577 MB of weights at the default grid) and MIMOGeneralDotProductAttention (:320-348, softmax over the keys, weighted sum of the
// warped maps).  All of it is HBM-bound streaming; every reduction runs in one fixed order (run-to-run identical gradients).
//
//   linear_rows_dx_kernel   dx (m, k) = dz (m, n) . W (n, k): W is read ONCE, a thread owns four consecutive k for all m <= 8 rows
//   linear_rows_dw_kernel   dW (n, k) = dz^T x: a thread keeps its four k of all m rows of x in registers and writes 32 rows of dW
//   (dz = dy where the forward's ReLU passed, else 0; db = column sums of dz -- both formed from the (m, n) operands in LDS)
//   fuse_dot_kernel / fuse_small_kernel / fuse_scale_kernel
//                           dcoef_j = <dout, map_j> (two-stage), softmax backward + dkeys / dquery (one workgroup), dmap_j = coef_j dout
#include <cstdint>

#include "av2x_common.hpp"
#include "airv2x_hip.h"

namespace {

constexpr int kMaxM = 8;         // x rows per pass
constexpr int kMaxAgents = 32;   // as when2com.hip
constexpr int kDotChunks = 512;

typedef float f4 __attribute__((ext_vector_type(4)));

// dz[i][j] = (act && y[i][j] <= 0) ? 0 : dy[i][j] into LDS, [i][n] row-major
__device__ __forceinline__ void load_dz(float* sdz, const float* __restrict__ dy, const float* __restrict__ y, int m, int n, int act) {
    for (int e = threadIdx.x; e < m * n; e += blockDim.x) sdz[e] = (act && y[e] <= 0.f) ? 0.f : dy[e];
    __syncthreads();
}

template <int M>
__global__ __launch_bounds__(256) void linear_rows_dx_kernel(const float* __restrict__ w, const float* __restrict__ dy, const float* __restrict__ y,
                                                             int m, int n, int k, int act, float* __restrict__ dx, float* __restrict__ db,
                                                             int db_acc) {
    extern __shared__ float sdz[];
    load_dz(sdz, dy, y, m, n, act);
    if (db && blockIdx.x == 0) {      // bias gradient: column sums of dz, rows in order (db_acc: on top of the earlier row groups')
        for (int j = threadIdx.x; j < n; j += blockDim.x) {
            float s = db_acc ? db[j] : 0.f;
            for (int i = 0; i < m; ++i) s += sdz[i * n + j];
            db[j] = s;
        }
    }
    const size_t k4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k4 * 4 >= (size_t)k || !dx) return;
    f4 acc[M];
#pragma unroll
    for (int i = 0; i < M; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    const f4* wp = reinterpret_cast<const f4*>(w) + k4;
    const size_t rs = (size_t)k / 4;
    for (int j = 0; j < n; ++j) {
        const f4 wv = __builtin_nontemporal_load(wp + (size_t)j * rs);
#pragma unroll
        for (int i = 0; i < M; ++i)
            if (i < m) {
                const float d = sdz[i * n + j];
                acc[i].x = fmaf(d, wv.x, acc[i].x); acc[i].y = fmaf(d, wv.y, acc[i].y);
                acc[i].z = fmaf(d, wv.z, acc[i].z); acc[i].w = fmaf(d, wv.w, acc[i].w);
            }
    }
#pragma unroll
    for (int i = 0; i < M; ++i)
        if (i < m) reinterpret_cast<f4*>(dx)[(size_t)i * rs + k4] = acc[i];
}

// The same product for SMALL k (a squeeze-excite layer: k = 48 .. 1 152, n = 1 152 .. 48, W <= 221 KB): the streaming kernel above would
// run 12 .. 288 threads through one dependent chain of n loads.  Here a workgroup = KQ k-quads x JL j-lanes (KQ * JL = 256): the rows of W
// are split over the j-lanes (four loads in flight each), the JL partial sums meet in LDS and are added in lane order.
template <int KQ, int JL>
__global__ __launch_bounds__(256) void linear_small_dx_kernel(const float* __restrict__ w, const float* __restrict__ dy, const float* __restrict__ y,
                                                              int m, int n, int k, int act, float* __restrict__ dx, float* __restrict__ db,
                                                              int db_acc) {
    extern __shared__ float sdz[];                    // [m][n] dz, then [JL][kMaxM][KQ] quads
    f4* red = reinterpret_cast<f4*>(sdz + ((m * n + 3) & ~3));
    load_dz(sdz, dy, y, m, n, act);
    if (db && blockIdx.x == 0) {
        for (int j = threadIdx.x; j < n; j += blockDim.x) {
            float s = db_acc ? db[j] : 0.f;
            for (int i = 0; i < m; ++i) s += sdz[i * n + j];
            db[j] = s;
        }
    }
    if (!dx) return;
    const int ql = threadIdx.x % KQ, jl = threadIdx.x / KQ;
    const int kq = blockIdx.x * KQ + ql;
    const size_t rs = (size_t)k / 4;
    f4 acc[kMaxM];
#pragma unroll
    for (int i = 0; i < kMaxM; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    if ((size_t)kq < rs) {
        const f4* wp = reinterpret_cast<const f4*>(w) + kq;
        int j = jl;
        for (; j + 3 * JL < n; j += 4 * JL) {
            f4 wv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) wv[u] = wp[(size_t)(j + u * JL) * rs];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < kMaxM; ++i)
                    if (i < m) {
                        const float d = sdz[i * n + j + u * JL];
                        acc[i].x = fmaf(d, wv[u].x, acc[i].x); acc[i].y = fmaf(d, wv[u].y, acc[i].y);
                        acc[i].z = fmaf(d, wv[u].z, acc[i].z); acc[i].w = fmaf(d, wv[u].w, acc[i].w);
                    }
        }
        for (; j < n; j += JL) {
            const f4 wv = wp[(size_t)j * rs];
#pragma unroll
            for (int i = 0; i < kMaxM; ++i)
                if (i < m) {
                    const float d = sdz[i * n + j];
                    acc[i].x = fmaf(d, wv.x, acc[i].x); acc[i].y = fmaf(d, wv.y, acc[i].y);
                    acc[i].z = fmaf(d, wv.z, acc[i].z); acc[i].w = fmaf(d, wv.w, acc[i].w);
                }
        }
    }
#pragma unroll
    for (int i = 0; i < kMaxM; ++i)
        if (i < m) red[(jl * kMaxM + i) * KQ + ql] = acc[i];
    __syncthreads();
    for (int e = threadIdx.x; e < m * KQ; e += 256) {
        const int i = e / KQ, q = e % KQ;
        const size_t kk = (size_t)blockIdx.x * KQ + q;
        if (kk >= rs) continue;
        f4 a = red[i * KQ + q];
        for (int l = 1; l < JL; ++l) {
            const f4 b = red[(l * kMaxM + i) * KQ + q];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        reinterpret_cast<f4*>(dx)[(size_t)i * rs + kk] = a;
    }
}

template <int M>
__global__ __launch_bounds__(256) void linear_rows_dw_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ y,
                                                             int m, int n, int k, int act, int accumulate, float* __restrict__ dw) {
    extern __shared__ float sdz[];
    load_dz(sdz, dy, y, m, n, act);
    const size_t k4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k4 * 4 >= (size_t)k) return;
    const size_t rs = (size_t)k / 4;
    f4 xv[M];
#pragma unroll
    for (int i = 0; i < M; ++i) xv[i] = i < m ? reinterpret_cast<const f4*>(x)[(size_t)i * rs + k4] : f4{0.f, 0.f, 0.f, 0.f};
    const int j0 = blockIdx.y * 32, j1 = min(n, j0 + 32);
    for (int j = j0; j < j1; ++j) {
        f4 o = accumulate ? reinterpret_cast<const f4*>(dw)[(size_t)j * rs + k4] : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < M; ++i)
            if (i < m) {
                const float d = sdz[i * n + j];
                o.x = fmaf(d, xv[i].x, o.x); o.y = fmaf(d, xv[i].y, o.y); o.z = fmaf(d, xv[i].z, o.z); o.w = fmaf(d, xv[i].w, o.w);
            }
        __builtin_nontemporal_store(o, reinterpret_cast<f4*>(dw) + (size_t)j * rs + k4);
    }
}

struct AgentPtrs { const f4* p[kMaxAgents]; };
struct AgentOut { f4* p[kMaxAgents]; };

__global__ __launch_bounds__(256) void fuse_dot_kernel(const f4* __restrict__ dout, const AgentPtrs agents, size_t hwc4, float* __restrict__ part) {
    const int j = blockIdx.y;
    const f4* a = agents.p[j];
    double s = 0.0;     // fixed assignment of elements to threads and a fixed tree: run-to-run identical
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < hwc4; i += (size_t)gridDim.x * blockDim.x) {
        const f4 d = dout[i], v = a[i];
        s += (double)(d.x * v.x + d.y * v.y) + (double)(d.z * v.z + d.w * v.w);
    }
    __shared__ double red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[(size_t)j * gridDim.x + blockIdx.x] = (float)red[0];
}

__global__ __launch_bounds__(256) void fuse_small_kernel(const float* __restrict__ keys, const float* __restrict__ q, const float* __restrict__ coef,
                                                         const float* __restrict__ part, int chunks, int n, int ks,
                                                         float* __restrict__ dkeys, float* __restrict__ dq) {
    __shared__ float dlog[kMaxAgents];
    __shared__ double dco[kMaxAgents];
    if ((int)threadIdx.x < n) {
        double s = 0.0;
        for (int c = 0; c < chunks; ++c) s += (double)part[(size_t)threadIdx.x * chunks + c];
        dco[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {     // softmax backward over the keys: dlogit_j = p_j (dp_j - sum_i p_i dp_i)
        double t = 0.0;
        for (int j = 0; j < n; ++j) t += (double)coef[j] * dco[j];
        for (int j = 0; j < n; ++j) dlog[j] = (float)((double)coef[j] * (dco[j] - t));
    }
    __syncthreads();
    for (int c = threadIdx.x; c < ks; c += blockDim.x) {
        float s = 0.f;
        for (int j = 0; j < n; ++j) {
            if (dkeys) dkeys[(size_t)j * ks + c] = dlog[j] * q[c];
            s = fmaf(dlog[j], keys[(size_t)j * ks + c], s);
        }
        if (dq) dq[c] = s;
    }
}

__global__ __launch_bounds__(256) void fuse_scale_kernel(const f4* __restrict__ dout, const float* __restrict__ coef, const AgentOut dagents, size_t hwc4) {
    const int j = blockIdx.y;
    const float p = coef[j];
    f4* o = dagents.p[j];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < hwc4; i += (size_t)gridDim.x * blockDim.x) {
        const f4 d = dout[i];
        o[i] = f4{p * d.x, p * d.y, p * d.z, p * d.w};
    }
}

// ---- V2VNet (models/v2vnet_modules/v2v_fuse.py:110-170, convgru.py:52-73): the one-step ConvGRU with a zero hidden state reduces to
// out = sigmoid(update gate) * tanh(candidate); the "max" aggregation over the neighbours' messages routes the gradient to the first maximum.
__global__ __launch_bounds__(256) void gru_gate_kernel(const f4* __restrict__ beta, const f4* __restrict__ cnm, f4* __restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f4 b = beta[i], c = cnm[i];
        out[i] = f4{tanhf(c.x) / (1.f + expf(-b.x)), tanhf(c.y) / (1.f + expf(-b.y)), tanhf(c.z) / (1.f + expf(-b.z)), tanhf(c.w) / (1.f + expf(-b.w))};
    }
}

struct GateGrad { float db, dc; };
__device__ __forceinline__ GateGrad gru_gate_grad(float b, float c, float d) {
    const float u = 1.f / (1.f + expf(-b)), t = tanhf(c);
    return GateGrad{d * t * u * (1.f - u), d * u * (1.f - t * t)};
}

__global__ __launch_bounds__(256) void gru_gate_backward_kernel(const f4* __restrict__ beta, const f4* __restrict__ cnm, const f4* __restrict__ dout,
                                                                f4* __restrict__ dbeta, f4* __restrict__ dcnm, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f4 b = beta[i], c = cnm[i], d = dout[i];
        const GateGrad g0 = gru_gate_grad(b.x, c.x, d.x), g1 = gru_gate_grad(b.y, c.y, d.y), g2 = gru_gate_grad(b.z, c.z, d.z),
                       g3 = gru_gate_grad(b.w, c.w, d.w);
        dbeta[i] = f4{g0.db, g1.db, g2.db, g3.db};
        dcnm[i] = f4{g0.dc, g1.dc, g2.dc, g3.dc};
    }
}

__global__ __launch_bounds__(256) void agent_max_kernel(const float* __restrict__ x, int n, size_t elems, float* __restrict__ out, uint8_t* __restrict__ idx) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < elems; i += (size_t)gridDim.x * blockDim.x) {
        float m = x[i];
        int k = 0;
        for (int j = 1; j < n; ++j) {
            const float v = x[(size_t)j * elems + i];
            if (v > m) { m = v; k = j; }        // the FIRST maximum keeps the gradient (torch.max over a dim on ties)
        }
        out[i] = m;
        idx[i] = (uint8_t)k;
    }
}

__global__ __launch_bounds__(256) void agent_max_backward_kernel(const float* __restrict__ dout, const uint8_t* __restrict__ idx, int n, size_t elems,
                                                                 float* __restrict__ dx) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < elems; i += (size_t)gridDim.x * blockDim.x) {
        const float d = dout[i];
        const int k = idx[i];
        for (int j = 0; j < n; ++j) dx[(size_t)j * elems + i] = j == k ? d : 0.f;
    }
}

}  // namespace

extern "C" int av2x_gru_gate(const float* beta, const float* cnm, uint64_t n_elems, float* out, av2x_stream_t stream) {
    if (n_elems == 0) return 0;
    if (!beta || !cnm || !out) return av2x::fail("av2x_gru_gate: null argument");
    if (n_elems % 4) return av2x::fail("av2x_gru_gate: element count must be a multiple of 4");
    size_t blocks = (n_elems / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gru_gate_kernel, dim3((unsigned)blocks), dim3(256), 0, av2x::as_stream(stream), reinterpret_cast<const f4*>(beta),
                       reinterpret_cast<const f4*>(cnm), reinterpret_cast<f4*>(out), (size_t)(n_elems / 4));
    return av2x::check_launch("gru_gate_kernel");
}

extern "C" int av2x_gru_gate_backward(const float* beta, const float* cnm, const float* dout, uint64_t n_elems, float* dbeta, float* dcnm,
                                      av2x_stream_t stream) {
    if (n_elems == 0) return 0;
    if (!beta || !cnm || !dout || !dbeta || !dcnm) return av2x::fail("av2x_gru_gate_backward: null argument");
    if (n_elems % 4) return av2x::fail("av2x_gru_gate_backward: element count must be a multiple of 4");
    size_t blocks = (n_elems / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gru_gate_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, av2x::as_stream(stream), reinterpret_cast<const f4*>(beta),
                       reinterpret_cast<const f4*>(cnm), reinterpret_cast<const f4*>(dout), reinterpret_cast<f4*>(dbeta), reinterpret_cast<f4*>(dcnm),
                       (size_t)(n_elems / 4));
    return av2x::check_launch("gru_gate_backward_kernel");
}

extern "C" int av2x_agent_argmax(const float* x, int32_t n_agents, uint64_t elems_per_agent, float* out, uint8_t* index, av2x_stream_t stream) {
    if (elems_per_agent == 0) return 0;
    if (!x || !out || !index) return av2x::fail("av2x_agent_argmax: null argument");
    if (n_agents < 1 || n_agents > 255) return av2x::fail("av2x_agent_argmax: n_agents=%d outside [1,255]", n_agents);
    size_t blocks = (elems_per_agent + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(agent_max_kernel, dim3((unsigned)blocks), dim3(256), 0, av2x::as_stream(stream), x, n_agents, (size_t)elems_per_agent, out, index);
    return av2x::check_launch("agent_max_kernel");
}

extern "C" int av2x_agent_argmax_backward(const float* dout, const uint8_t* index, int32_t n_agents, uint64_t elems_per_agent, float* dx,
                                       av2x_stream_t stream) {
    if (elems_per_agent == 0) return 0;
    if (!dout || !index || !dx) return av2x::fail("av2x_agent_argmax_backward: null argument");
    if (n_agents < 1 || n_agents > 255) return av2x::fail("av2x_agent_argmax_backward: n_agents=%d outside [1,255]", n_agents);
    size_t blocks = (elems_per_agent + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(agent_max_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, av2x::as_stream(stream), dout, index, n_agents,
                       (size_t)elems_per_agent, dx);
    return av2x::check_launch("agent_max_backward_kernel");
}

extern "C" int av2x_linear_rows_backward(const float* x, const float* w, const float* y, const float* dy, int32_t m, int32_t n, int32_t k,
                                         int32_t act, float* dx, float* dw, float* db, av2x_stream_t stream) {
    if (m == 0) return 0;
    if (!x || !w || !dy || (act && !y)) return av2x::fail("av2x_linear_rows_backward: null argument");
    if (m < 0 || n <= 0 || k <= 0 || k % 4) return av2x::fail("av2x_linear_rows_backward: bad sizes (m=%d n=%d k=%d; k %% 4 == 0)", m, n, k);
    if (act < 0 || act > 1) return av2x::fail("av2x_linear_rows_backward: activation code %d (0 none, 1 ReLU)", act);
    if ((size_t)kMaxM * n * sizeof(float) > 64 * 1024) return av2x::fail("av2x_linear_rows_backward: n=%d too wide (8 x n floats of LDS)", n);
    for (const void* q : {(const void*)x, (const void*)w, (const void*)dx, (const void*)dw})
        if (reinterpret_cast<uintptr_t>(q) % 16) return av2x::fail("av2x_linear_rows_backward: x, w, dx, dw must be 16-byte aligned");
    hipStream_t st = av2x::as_stream(stream);
    const unsigned kb = (unsigned)((k / 4 + 255) / 256);
    for (int m0 = 0; m0 < m; m0 += kMaxM) {    // more than 8 rows: W is streamed once per group of 8 (dW accumulates over the groups)
        const int mm = (m - m0) < kMaxM ? (m - m0) : kMaxM;
        const size_t lds = (size_t)mm * n * sizeof(float);
        const float* ys = y ? y + (size_t)m0 * n : nullptr;
        if ((dx || db) && k <= 4096) {      // small k: rows of W split over j-lanes (see linear_small_dx_kernel)
            const size_t lds2 = (((size_t)mm * n + 3) & ~(size_t)3) * sizeof(float) + (size_t)256 * kMaxM * 16;
            float* dxs = dx ? dx + (size_t)m0 * k : nullptr;
            if (k >= 256) {
                static av2x::LdsLimit lim;
                lim.ensure(reinterpret_cast<const void*>(&linear_small_dx_kernel<64, 4>), lds2);
                hipLaunchKernelGGL((linear_small_dx_kernel<64, 4>), dim3((unsigned)((k / 4 + 63) / 64)), dim3(256), lds2, st, w, dy + (size_t)m0 * n, ys, mm,
                                   n, k, act, dxs, db, m0 > 0 ? 1 : 0);
            } else {
                static av2x::LdsLimit lim;
                lim.ensure(reinterpret_cast<const void*>(&linear_small_dx_kernel<16, 16>), lds2);
                hipLaunchKernelGGL((linear_small_dx_kernel<16, 16>), dim3((unsigned)((k / 4 + 15) / 16)), dim3(256), lds2, st, w, dy + (size_t)m0 * n, ys, mm,
                                   n, k, act, dxs, db, m0 > 0 ? 1 : 0);
            }
        } else if (dx || db) {
            if (mm <= 4) hipLaunchKernelGGL(linear_rows_dx_kernel<4>, dim3(kb), dim3(256), lds, st, w, dy + (size_t)m0 * n, ys, mm, n, k, act,
                                            dx ? dx + (size_t)m0 * k : nullptr, db, m0 > 0 ? 1 : 0);
            else hipLaunchKernelGGL(linear_rows_dx_kernel<8>, dim3(kb), dim3(256), lds, st, w, dy + (size_t)m0 * n, ys, mm, n, k, act,
                                    dx ? dx + (size_t)m0 * k : nullptr, db, m0 > 0 ? 1 : 0);
        }
        if (dw) {
            const dim3 grid(kb, (unsigned)((n + 31) / 32));
            if (mm <= 4) hipLaunchKernelGGL(linear_rows_dw_kernel<4>, grid, dim3(256), lds, st, x + (size_t)m0 * k, dy + (size_t)m0 * n, ys, mm, n, k,
                                            act, m0 > 0 ? 1 : 0, dw);
            else hipLaunchKernelGGL(linear_rows_dw_kernel<8>, grid, dim3(256), lds, st, x + (size_t)m0 * k, dy + (size_t)m0 * n, ys, mm, n, k, act,
                                    m0 > 0 ? 1 : 0, dw);
        }
    }
    return av2x::check_launch("linear_rows_backward");
}

extern "C" uint64_t av2x_when2com_fuse_backward_workspace_bytes(int32_t n_agents) { return (uint64_t)(n_agents > 0 ? n_agents : 0) * kDotChunks * sizeof(float); }

extern "C" int av2x_when2com_fuse_backward(const float* keys, const float* query, const float* coef, int32_t n_agents, int32_t key_size,
                                           const float* const* agents, uint64_t elems_per_agent, const float* dout, float* const* dagents,
                                           float* dkeys, float* dquery, void* workspace, av2x_stream_t stream) {
    if (!keys || !query || !coef || !agents || !dout || !workspace) return av2x::fail("av2x_when2com_fuse_backward: null argument");
    if (n_agents < 1 || n_agents > kMaxAgents) return av2x::fail("av2x_when2com_fuse_backward: n_agents=%d outside [1,%d]", n_agents, kMaxAgents);
    if (key_size <= 0 || elems_per_agent == 0 || elems_per_agent % 4)
        return av2x::fail("av2x_when2com_fuse_backward: bad sizes (elements per agent must be a multiple of 4)");
    const size_t hwc4 = elems_per_agent / 4;
    AgentPtrs am;
    AgentOut ao;
    for (int j = 0; j < kMaxAgents; ++j) {
        am.p[j] = j < n_agents ? reinterpret_cast<const f4*>(agents[j]) : nullptr;
        ao.p[j] = (j < n_agents && dagents) ? reinterpret_cast<f4*>(dagents[j]) : nullptr;
        if (j < n_agents && (!agents[j] || reinterpret_cast<uintptr_t>(agents[j]) % 16 || (dagents && (!dagents[j] || reinterpret_cast<uintptr_t>(dagents[j]) % 16))))
            return av2x::fail("av2x_when2com_fuse_backward: agent map %d is null or not 16-byte aligned", j);
    }
    hipStream_t st = av2x::as_stream(stream);
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(fuse_dot_kernel, dim3(kDotChunks, n_agents), dim3(256), 0, st, reinterpret_cast<const f4*>(dout), am, hwc4, part);
    hipLaunchKernelGGL(fuse_small_kernel, dim3(1), dim3(256), 0, st, keys, query, coef, part, kDotChunks, n_agents, key_size, dkeys, dquery);
    if (dagents) {
        size_t blocks = (hwc4 + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(fuse_scale_kernel, dim3((unsigned)blocks, n_agents), dim3(256), 0, st, reinterpret_cast<const f4*>(dout), coef, ao, hwc4);
    }
    return av2x::check_launch("when2com_fuse_backward");
}
