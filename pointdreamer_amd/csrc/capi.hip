// libpdhip: version + thread-local error message (never throws across the C ABI).
#include "common.h"
#include <string.h>

namespace pdhip {
static thread_local char g_err[512] = "";
int g_lab_units = 0;
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace pdhip

extern "C" int pdhip_version(void) { return 207; }
// number of translation units of THIS library that were compiled with a wrong-result PD_LAB_* switch (common.h); 0 for a product build
extern "C" int pdhip_lab_build(void) { return pdhip::g_lab_units; }
extern "C" const char* pdhip_last_error(void) { return pdhip::g_err; }
