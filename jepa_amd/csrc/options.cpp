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
  int dflt;
};
const OptDef kDefs[VJ_OPT_COUNT] = {
    {"gemm_fwd_flags", 0}, {"gemm_dgrad_flags", 0}, {"gemm_4w", 0},         {"gemm_persist", 1},
    {"wgrad_tn", 1},       {"wgrad_group", 1},      {"wgrad_slow_issue", 0}, {"attn_dkdv_kt", 0},
    {"gemm_dbg", 0},
};
std::atomic<int> g_val[VJ_OPT_COUNT];
std::once_flag g_once;

void init_once() {
  std::call_once(g_once, [] {
    for (int i = 0; i < VJ_OPT_COUNT; i++) {
      std::string env = "VJ_";
      for (const char* c = kDefs[i].name; *c; c++) env.push_back((char)toupper((unsigned char)*c));
      const char* e = getenv(env.c_str());
      g_val[i].store(e ? atoi(e) : kDefs[i].dflt, std::memory_order_relaxed);
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
  return g_val[id].load(std::memory_order_relaxed);
}

extern "C" int vj_set_option(const char* name, int value) {
  init_once();
  const int i = find(name);
  VJ_CHECK_ARG(i >= 0, "vj_set_option: unknown option '%s'", name ? name : "(null)");
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
