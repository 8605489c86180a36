#include "options.hpp"
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <cctype>
#include <mutex>
#include <string>
#include "common.hpp"

namespace {
struct OptDef {
  const char* name;
  int dflt, lo, hi;   // accepted range (inclusive): a value outside it is rejected, never silently mapped to some kernel
};
// values inside [lo, hi] that still select nothing
bool in_set(int id, int v) {
  if (id == VJ_OPT_GEMM_EPI_PRE) return v == 0 || v == 4;
  return true;
}
const OptDef kDefs[VJ_OPT_COUNT] = {
    {"gemm_fwd_flags", 0, 0, 0x1ff}, {"gemm_dgrad_flags", 0, 0, 0x1ff}, {"gemm_4w", 0, 0, 1},     {"gemm_persist", 2, 0, 3},
    {"wgrad_tn", 1, 0, 1},           {"wgrad_group", 1, 0, 1},          {"gemm_dbg", 0, 0, 7},    {"attn_softmax", 2, 1, 2},
    {"bias_fuse", 1, 0, 1},          {"gemm_raster", 260, 0, 511},      {"ws_guard", 0, 0, 1},    {"gemm_epi_pre", 4, 0, 4},
};
std::atomic<int> g_val[VJ_OPT_COUNT];
std::once_flag g_once;

void init_once() {
  std::call_once(g_once, [] {
    for (int i = 0; i < VJ_OPT_COUNT; i++) {
      std::string env = "VJ_";
      for (const char* c = kDefs[i].name; *c; c++) env.push_back((char)toupper((unsigned char)*c));
      const char* e = getenv(env.c_str());
      int v = e ? atoi(e) : kDefs[i].dflt;
      if (v < kDefs[i].lo || v > kDefs[i].hi || !in_set(i, v)) {
        fprintf(stderr, "libvjepa_hip: %s=%d outside [%d, %d], using the default %d\n", env.c_str(), v, kDefs[i].lo,
                kDefs[i].hi, kDefs[i].dflt);
        v = kDefs[i].dflt;
      }
      g_val[i].store(v, std::memory_order_relaxed);
    }
  });
}
int find(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < VJ_OPT_COUNT; i++)
    if (strcmp(name, kDefs[i].name) == 0) return i;
  return -1;
}
}  // namespace

int vj_opt(int id) {
  init_once();
  if (id < 0 || id >= VJ_OPT_COUNT) return 0;
  return g_val[id].load(std::memory_order_relaxed);
}

extern "C" int vj_set_option(const char* name, int value) {
  init_once();
  const int i = find(name);
  VJ_CHECK_ARG(i >= 0, "vj_set_option: unknown option '%s'", name ? name : "(null)");
  VJ_CHECK_ARG(value >= kDefs[i].lo && value <= kDefs[i].hi && in_set(i, value),
               "vj_set_option: %s=%d outside [%d, %d] or not a value the option defines", name, value, kDefs[i].lo, kDefs[i].hi);
  g_val[i].store(value, std::memory_order_relaxed);
  return 0;
}

extern "C" int vj_get_option(const char* name, int* value) {
  init_once();
  const int i = find(name);
  VJ_CHECK_ARG(i >= 0 && value != nullptr, "vj_get_option: unknown option '%s'", name ? name : "(null)");
  *value = g_val[i].load(std::memory_order_relaxed);
  return 0;
}
