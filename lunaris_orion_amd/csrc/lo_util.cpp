#include "lo_common.h"
#include <stdarg.h>
#include <stdio.h>
static thread_local char g_err[512] = "";
void lo_set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
const char* lo_get_error() { return g_err; }
int lo_check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return LO_OK;
  lo_set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
  return LO_ERR_HIP;
}
