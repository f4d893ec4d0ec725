// vt_set_option / vt_get_option / vt_reset_options (include/vidtok_amd.h): the process-wide switch table of options.h.
#include "options.h"

#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "../../include/vidtok_amd.h"

void vt_set_error(const char* fmt, ...);

namespace {
struct OptDef {
  const char* name;   // vt_set_option name; the environment variable is "VT_" + upper-case name
  int dflt;
};
// order = enum VtOpt
const OptDef kDefs[OPT_COUNT] = {
    {"conv_buf", 1},        {"conv_tinner", 1},     {"conv_ldsepi", 1},      {"conv_ws", 2},
    {"conv_narrow", 1},     {"conv_tile", 0},       {"conv_tile_min", 128},  {"conv_fuse_ln", 1}, {"conv_fuse_ln256", 1},
    {"tblock_fused", 1},    {"tblock_prof_mode", 0},
    {"conv_deep", 1},       {"ws_prof_mode", 0},    {"attn_flash", 1},      {"conv_splitk", 1},
    {"conv_tskip", 1},      {"conv_in8", 1},        {"conv_tup_ln", 1},     {"conv_nt_mb", 64},
};
std::atomic<int> g_val[OPT_COUNT];
std::once_flag g_once;

int env_default(int id) {
  char var[64] = "VT_";
  size_t n = 3;
  for (const char* s = kDefs[id].name; *s && n + 1 < sizeof(var); ++s) var[n++] = (*s >= 'a' && *s <= 'z') ? (char)(*s - 32) : *s;
  var[n] = 0;
  const char* e = getenv(var);
  return e ? atoi(e) : kDefs[id].dflt;
}
void init_all() {
  for (int i = 0; i < OPT_COUNT; ++i) g_val[i].store(env_default(i), std::memory_order_relaxed);
}
int find(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < OPT_COUNT; ++i)
    if (strcmp(name, kDefs[i].name) == 0) return i;
  return -1;
}
}  // namespace

int vt_opt(int id) {
  std::call_once(g_once, init_all);
  return g_val[id].load(std::memory_order_relaxed);
}

extern "C" int vt_set_option(const char* name, int32_t value) {
  const int id = find(name);
  if (id < 0) {
    vt_set_error("vt_set_option: unknown option '%s'", name ? name : "(null)");
    return VT_ERR_ARG;
  }
  std::call_once(g_once, init_all);
  g_val[id].store(value, std::memory_order_relaxed);
  return VT_OK;
}

extern "C" int vt_get_option(const char* name, int32_t* value) {
  const int id = find(name);
  if (id < 0 || !value) {
    vt_set_error("vt_get_option: unknown option '%s' or null output", name ? name : "(null)");
    return VT_ERR_ARG;
  }
  *value = vt_opt(id);
  return VT_OK;
}

extern "C" int vt_reset_options(void) {
  std::call_once(g_once, init_all);
  init_all();
  return VT_OK;
}

extern "C" int vt_option_count(void) { return OPT_COUNT; }
extern "C" const char* vt_option_name(int32_t i) { return (i >= 0 && i < OPT_COUNT) ? kDefs[i].name : nullptr; }
