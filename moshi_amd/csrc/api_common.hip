// Version / error plumbing of the C ABI (include/moshi_mi.h).
#include "mmi_common.h"

static thread_local std::string g_last_error;

void mmi_set_error(const std::string& msg) { g_last_error = msg; }
int mmi_fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

extern "C" int mmi_version(void) { return MMI_ABI_VERSION; }
extern "C" const char* mmi_last_error(void) { return g_last_error.c_str(); }
