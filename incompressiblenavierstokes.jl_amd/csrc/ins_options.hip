// Run-time switches of the library (A/B paths, tile shapes).  Every switch has a name, is initialised from the environment
// variable of the same name the first time it is read, and can be changed afterwards through the C ABI (ins_set_option), so a
// test can run the fused and the generic path of one entry point in one process on one allocation.  Internal code reads a
// switch by its enum id (O(1), no string compare on the launch path).
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "ins_internal.h"

namespace {

struct Opt {
  const char* name;
  std::atomic<long long> value;
  std::atomic<int> state;  // 0 = not read yet (environment decides), 1 = set
};

#define INS_OPT_ROW(id) {#id, {0}, {0}},
Opt g_opts[INS_OPT_COUNT] = {INS_OPT_LIST(INS_OPT_ROW)};
#undef INS_OPT_ROW

std::atomic<long long> g_epoch{0};

long long from_env(const char* name) {
  const char* v = getenv(name);
  if (!v) return 0;
  if (!*v) return 1;  // "set" counts as on
  char* end = nullptr;
  const long long x = strtoll(v, &end, 10);
  if (end == v) {  // not a number: a few switches take words (INS_SLAB_ZSOLVE=fft lives on the Python side); anything else = on
    return 1;
  }
  return x;
}

}  // namespace

long long ins_opt(int id) {
  Opt& o = g_opts[id];
  if (o.state.load(std::memory_order_acquire) == 0) {
    o.value.store(from_env(o.name), std::memory_order_relaxed);
    o.state.store(1, std::memory_order_release);
  }
  return o.value.load(std::memory_order_relaxed);
}

long long ins_opt_epoch() { return g_epoch.load(std::memory_order_relaxed); }

static int find(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < INS_OPT_COUNT; ++i)
    if (!strcmp(name, g_opts[i].name)) return i;
  return -1;
}

extern "C" int ins_set_option(const char* name, int64_t value) {
  const int id = find(name);
  if (id < 0) {
    ins_set_error("ins_set_option: unknown option '%s'", name ? name : "(null)");
    return INS_ERR_INVALID;
  }
  g_opts[id].value.store(value, std::memory_order_relaxed);
  g_opts[id].state.store(1, std::memory_order_release);
  g_epoch.fetch_add(1, std::memory_order_relaxed);
  return INS_OK;
}

extern "C" int ins_get_option(const char* name, int64_t* value) {
  const int id = find(name);
  if (id < 0 || !value) {
    ins_set_error("ins_get_option: unknown option '%s'", name ? name : "(null)");
    return INS_ERR_INVALID;
  }
  *value = ins_opt(id);
  return INS_OK;
}

extern "C" int ins_option_count(void) { return INS_OPT_COUNT; }

extern "C" const char* ins_option_name(int i) { return (i >= 0 && i < INS_OPT_COUNT) ? g_opts[i].name : nullptr; }
