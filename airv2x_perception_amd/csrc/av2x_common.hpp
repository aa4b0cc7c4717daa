// Shared host-side helpers for the C-ABI translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

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

constexpr int kWave = 64;  // CDNA4 wavefront

}  // namespace av2x
