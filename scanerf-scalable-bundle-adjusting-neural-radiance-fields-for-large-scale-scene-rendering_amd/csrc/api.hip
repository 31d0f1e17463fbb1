// api.hip -- error plumbing and ABI version of libscanerf_hip.
#include <stdarg.h>

#include "common.h"

namespace scanerf {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace scanerf

SCANERF_API const char *scanerf_last_error(void) { return scanerf::g_err; }
SCANERF_API int scanerf_abi_version(void) { return 9; }
// 1 = built with -DSCANERF_EXPERIMENTS (make EXP=1): the tuning switches of csrc/common.h (tune_int / tune_set) read the
// environment; 0 = the product build: every switch compiled to its default, no getenv anywhere in the library
SCANERF_API int scanerf_experiments_enabled(void)
{
#ifdef SCANERF_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}
