// Single-workgroup (1024 threads) exclusive scan over a functor, shared by the voxelizer and the
// post-processor compaction steps.  Order-preserving compaction = scan of 0/1 flags.
#pragma once
#include <hip/hip_runtime.h>

namespace av2x {

// exclusive scan of `n` ints produced by functor f(i), single workgroup of 1024 threads
template <class F, class G>
__device__ inline void block_scan(int n, F f, G store, int* total) {
    __shared__ int wsum[16];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < n ? f(i) : 0;
        int s = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(s, o);
            if (lane >= o) s += t;
        }
        if (lane == 63) wsum[wave] = s;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < wave; ++k) woff += wsum[k];
        const int excl = carry + woff + s - v;
        if (i < n) store(i, excl);
        __syncthreads();
        if (threadIdx.x == 1023) carry = excl + v;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}


}  // namespace av2x
