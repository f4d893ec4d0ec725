// Thread-local last-error string for the C-ABI (include/vidtok_amd.h).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/vidtok_amd.h"

static thread_local char g_err[512] = "";

void vt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vt_last_error(void) { return g_err; }
extern "C" int vt_version(void) { return 103; }
extern "C" int vt_conv_desc_size(void) { return (int)sizeof(vt_conv_desc); }
