// Error plumbing and ABI version for libvlb.
#include <stdarg.h>

#include "common.hpp"

static thread_local char g_err[512] = "";

void vlb_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vlb_last_error(void) { return g_err; }
extern "C" int vlb_abi_version(void) { return VLB_ABI_VERSION; }
