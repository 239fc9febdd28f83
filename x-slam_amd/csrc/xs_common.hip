// xs_common.hip — error plumbing of the C ABI (include/xslam_amd.h).
#include "xs_device.h"
#include "../../include/xslam_amd.h"
#include <stdio.h>

static thread_local char g_err[512] = "";

extern "C" int xs_set_error(hipError_t e, const char *what) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what ? what : "", hipGetErrorString(e));
    return (int)e;
}
extern "C" const char *xs_last_error(void) { return g_err; }
extern "C" int xs_abi_version(void) { return 2; }   // 2 (round 6): the per-thread setters are gone (options structs only), the Gauss-Newton loop protocol (xs_gn_*), the integrate workspace layout of round 5
