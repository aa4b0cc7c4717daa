// Library-level C-ABI entry points: ABI version and the thread-local error string.
#include "av2x_common.hpp"

namespace av2x {
char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace av2x

extern "C" int av2x_version(void) { return AV2X_ABI_VERSION; }

extern "C" const char* av2x_last_error(void) { return av2x::error_buffer(); }
