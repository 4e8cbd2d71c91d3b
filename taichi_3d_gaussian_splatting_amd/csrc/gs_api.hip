// gs_api.hip -- error channel and version of the C ABI (include/gsplat_hip.h).
#include <stdarg.h>

#include "gs_common.h"

static thread_local char g_error[512] = "";

void gs_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
}

extern "C" const char *gs_last_error(void) { return g_error; }
// A tuning build (measurement arms compiled in: tools/build_variants.sh ... -DGS_TUNING_BUILD=1) reports another version, so
// the product loader (_lib.load) refuses it.
#ifdef GS_TUNING_BUILD
extern "C" int gs_abi_version(void) { return GS_ABI_VERSION + GS_ABI_TUNING_OFFSET; }
#else
extern "C" int gs_abi_version(void) { return GS_ABI_VERSION; }
#endif
