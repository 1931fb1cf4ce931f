// Error reporting + device query for libgpnerf_hip.so.
#include "common.h"

#include <cstdarg>
#include <cstdio>

static thread_local char g_err[512] = "";

int gpn_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int gpn_num_cus() {
  static thread_local int cached_dev = -1, cached = 256;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (dev != cached_dev) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cached = n;
    cached_dev = dev;
  }
  return cached;
}

extern "C" int gpn_abi_version(void) { return GPN_ABI_VERSION; }
extern "C" const char* gpn_last_error(void) { return g_err; }
