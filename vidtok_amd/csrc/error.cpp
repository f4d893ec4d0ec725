// Thread-local last-error string for the C-ABI (include/vidtok_amd.h).
#include <hip/hip_runtime.h>
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
extern "C" int vt_version(void) { return 104; }
extern "C" int vt_conv_desc_size(void) { return (int)sizeof(vt_conv_desc); }

// Replay of a captured launch sequence: hipGraphLaunch of an instantiated graph on `stream` and nothing else.  The host's graph
// cache (vidtok_amd/graphs.py) captures with its framework's allocator-aware capture, but replays through this entry point: the
// framework's own replay also refreshes the state of its device random generators with two fill kernels per replay, and the
// captured sequences draw no random numbers on the device.
extern "C" int vt_graph_launch(void* graph_exec, vt_stream stream) {
  if (!graph_exec) {
    vt_set_error("vt_graph_launch: null graph");
    return VT_ERR_ARG;
  }
  const hipError_t e = hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(graph_exec), reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) {
    vt_set_error("hipGraphLaunch failed: %s", hipGetErrorString(e));
    return VT_ERR_HIP;
  }
  return VT_OK;
}
