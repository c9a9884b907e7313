// gs_error.hip -- error reporting for the C ABI (no exceptions cross the boundary).
#include <stdarg.h>
#include <stdio.h>

#include "gs_common.h"

namespace gs {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return GS_EHIP;
    }
    return GS_OK;
}

}  // namespace gs

extern "C" {
const char* gs_last_error(void) { return gs::g_err; }
int gs_abi_version(void) { return 8; }
}
