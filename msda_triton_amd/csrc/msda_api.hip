// msda_api.hip — library-level pieces of the C ABI: version, last-error text, A/B options.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

#include "../../include/msda_hip.h"

namespace msda {

static std::atomic<int> g_xcd_map{1};
static thread_local char g_err[256] = "";

int option_xcd_map() { return g_xcd_map.load(std::memory_order_relaxed); }

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace msda

extern "C" int msda_abi_version(void) { return MSDA_ABI_VERSION; }

extern "C" const char *msda_last_error(void) { return msda::g_err; }

extern "C" int msda_set_option(const char *key, int value)
{
    if (key && strcmp(key, "xcd_map") == 0) {
        msda::g_xcd_map.store(value ? 1 : 0, std::memory_order_relaxed);
        return 0;
    }
    msda::set_error("unknown option '%s'", key ? key : "(null)");
    return MSDA_ERR_BAD_ARG;
}

extern "C" int msda_get_option(const char *key)
{
    if (key && strcmp(key, "xcd_map") == 0) return msda::option_xcd_map();
    msda::set_error("unknown option '%s'", key ? key : "(null)");
    return MSDA_ERR_BAD_ARG;
}
