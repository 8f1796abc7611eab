// Error plumbing shared by every entry point of the C ABI (include/runet_hip.h).
#include "runet_common.h"
#include "../../include/runet_hip.h"
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" const char* runet_last_error(void) { return g_err; }
extern "C" void runet_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" int runet_abi_version(void) { return 1; }
