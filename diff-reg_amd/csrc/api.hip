// api.hip -- version / error reporting of the C ABI.
#include "common.h"
#include <stdio.h>
#include <string.h>

namespace dr {
static thread_local char g_hip_err[512] = "";
void set_hip_error(hipError_t e, const char* where) {
    snprintf(g_hip_err, sizeof(g_hip_err), "%s: %s (%d)", where, hipGetErrorString(e), (int)e);
}
}  // namespace dr

extern "C" {
int dr_version(void) { return 100; /* 0.1.0 */ }

const char* dr_strerror(int code) {
    switch (code) {
        case DR_OK: return "ok";
        case DR_EINVAL: return "invalid argument";
        case DR_ELAUNCH: return "HIP call failed";
        case DR_ENOSUP: return "shape not supported by this build";
        case DR_EWORKSPACE: return "workspace missing or too small";
        default: return "unknown error";
    }
}

const char* dr_last_hip_error(void) { return dr::g_hip_err; }
}
