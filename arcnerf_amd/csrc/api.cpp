// Error string + version of the C ABI (host only).
#include <string.h>

#include "../../include/arcnerf_hip.h"

namespace arcn {
static thread_local char g_err[256] = "";
void set_error(const char *msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
}  // namespace arcn

extern "C" __attribute__((visibility("default"))) const char *arcn_last_error(void) { return arcn::g_err; }
extern "C" __attribute__((visibility("default"))) int arcn_version(void) { return 100; }
