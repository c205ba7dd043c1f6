// Process-wide A/B and test switches of libwca.so (development aids; every default is the shipped, measured-best choice and the parity /
// bit-identity claims hold for the defaults). Until round 4 these were environment variables read on every launch (a stray variable
// silently changed kernel selection in production); now they are set only through the test entry point wca_test_set_switch, and the
// environment is consulted ONCE, at the first use of a switch, as its initial value (so `WCA_GEMM_SUPERTILE=4 python tools/...` still works).
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "kernels.h"

namespace wca {

namespace {
struct Sw {
  const char* name;
  const char* env;
  std::atomic<int> value;
  std::atomic<int> init;
};
Sw g_sw[DBG_SWITCH_COUNT] = {
    {"attn_split_variant", "WCA_ATTN_SPLIT_VARIANT", {0}, {0}},     // 1: the pair attention on the 16x16x32 kernel everywhere
    {"attn_variant", "WCA_ATTN_VARIANT", {0}, {0}},                 // f16 attention: 1 the 16x16x32 kernel, 3 row sums on the vector ALU
    {"head_stats_general", "WCA_HEAD_STATS_GENERAL", {0}, {0}},     // 1: the general head-statistics kernel
    {"gemm_supertile", "WCA_GEMM_SUPERTILE", {0}, {0}},             // > 0: m-panels per supertile of the persistent GEMM's tile order
    {"ln_pair_v4", "WCA_LN_PAIR_V4", {0}, {0}},                     // 1: the four-wide pair LayerNorm
    {"fail_precision_alloc", "WCA_TEST_FAIL_PRECISION_ALLOC", {0}, {0}},  // 1: inject an allocation failure into wca_set_precision_sites
    {"attn_split_drop", nullptr, {0}, {0}},                         // pass mask of the encoder's pair attention (wca_test_set_attn_split_drop)
    {"gemm_ring", "WCA_GEMM_RING", {0}, {0}},                       // 1: the pair GEMM on round 4's two-slot rings (default: three A slots + one W slot)
};
}  // namespace

int debug_switch(int id) {
  if (id < 0 || id >= DBG_SWITCH_COUNT) return 0;
  Sw& s = g_sw[id];
  if (!s.init.load(std::memory_order_acquire)) {
    int expected = 0;
    if (s.init.compare_exchange_strong(expected, 1)) {
      const char* ev = s.env ? std::getenv(s.env) : nullptr;
      if (ev) s.value.store(atoi(ev), std::memory_order_relaxed);
      s.init.store(2, std::memory_order_release);
    }
  }
  return s.value.load(std::memory_order_relaxed);
}

int set_debug_switch(const char* name, int value) {
  for (int i = 0; i < DBG_SWITCH_COUNT; ++i)
    if (std::strcmp(g_sw[i].name, name) == 0) {
      g_sw[i].init.store(2, std::memory_order_release);
      g_sw[i].value.store(value, std::memory_order_relaxed);
      return 0;
    }
  return -1;
}

}  // namespace wca
