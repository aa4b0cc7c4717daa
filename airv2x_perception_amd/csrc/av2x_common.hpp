// Shared host-side helpers for the C-ABI translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>

#include "airv2x_hip.h"

namespace av2x {

char* error_buffer();  // thread-local, 512 bytes (capi.hip)

inline int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return 1;
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("%s: %s", what, hipGetErrorString(e));
    return 0;
}

inline hipStream_t as_stream(av2x_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Raises a kernel's dynamic-LDS limit (needed above 64 KiB) once per (kernel, device).  One instance per kernel as a
// function-local static; safe with several host threads and with several devices in one process.
struct LdsLimit {
    std::atomic<size_t> bytes[16] = {};
    void ensure(const void* fn, size_t need) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::atomic<size_t>& b = bytes[dev & 15];
        if (b.load(std::memory_order_acquire) >= need) return;
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
        size_t cur = b.load(std::memory_order_relaxed);
        while (cur < need && !b.compare_exchange_weak(cur, need, std::memory_order_release)) {}
    }
};

constexpr int kWave = 64;  // CDNA4 wavefront

}  // namespace av2x
