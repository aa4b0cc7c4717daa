// Single-workgroup (1024 threads) exclusive scan over a functor, shared by the voxelizer, the point
// preparation and the post-processor compaction steps.  Order-preserving compaction = scan of 0/1 flags.
#pragma once
#include <hip/hip_runtime.h>

namespace av2x {

// exclusive scan of `n` ints produced by functor f(i), single workgroup of 1024 threads.
// Every thread owns ITEMS consecutive elements per pass (in-thread prefix, then a wave scan of the
// thread totals, then the 16 wave totals), so one pass covers 1024 * ITEMS elements with two barriers
// (ITEMS = 4 by default; 16 for the 70 400-anchor compaction of the post-processor).
template <int ITEMS = 4, class F, class G>
__device__ inline void block_scan(int n, F f, G store, int* total) {
    constexpr int PASS = 1024 * ITEMS;
    __shared__ int wsum[16];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < n; base += PASS) {
        const int i0 = base + threadIdx.x * ITEMS;
        int v[ITEMS], tsum = 0;
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            v[k] = (i0 + k) < n ? f(i0 + k) : 0;
            tsum += v[k];
        }
        int s = tsum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(s, o);
            if (lane >= o) s += t;
        }
        if (lane == 63) wsum[wave] = s;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < wave; ++k) woff += wsum[k];
        int excl = carry + woff + s - tsum;
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            if (i0 + k < n) store(i0 + k, excl);
            excl += v[k];
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry = excl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

}  // namespace av2x
