// Profiling probes: hipEvent pairs recorded on the launch stream around selected launches.
#include <vector>

#include "dvt_common.h"

unsigned g_dvt_prof_mask = 0;

namespace {
struct Sample {
  hipEvent_t a, b;
};
struct Probe {
  std::vector<Sample> pool;  // created lazily, reused
  size_t used = 0;
  double work = 0.0;
  hipEvent_t pending = nullptr;
};
Probe g_probes[DVT_N_PROBES];
constexpr size_t MAX_SAMPLES = 200000;
}  // namespace

void dvt_prof_begin(int probe, hipStream_t s) {
  Probe& p = g_probes[probe];
  if (p.used >= MAX_SAMPLES) return;
  if (p.used == p.pool.size()) {
    Sample sm;
    if (hipEventCreate(&sm.a) != hipSuccess || hipEventCreate(&sm.b) != hipSuccess) return;
    p.pool.push_back(sm);
  }
  (void)hipEventRecord(p.pool[p.used].a, s);
  p.pending = p.pool[p.used].a;
}

void dvt_prof_end(int probe, hipStream_t s, double work) {
  Probe& p = g_probes[probe];
  if (p.pending == nullptr) return;
  (void)hipEventRecord(p.pool[p.used].b, s);
  p.pending = nullptr;
  p.used++;
  p.work += work;
}

extern "C" int dvt_prof_enable(unsigned mask) {
  g_dvt_prof_mask = mask & ((1u << DVT_N_PROBES) - 1u);
  for (auto& p : g_probes) {
    p.used = 0;
    p.work = 0.0;
    p.pending = nullptr;
  }
  return 0;
}

extern "C" int dvt_prof_collect(int probe, double* total_ms, int64_t* count, double* work) {
  if (probe < 0 || probe >= DVT_N_PROBES || !total_ms || !count || !work) return DVT_E_BADARG;
  Probe& p = g_probes[probe];
  double tot = 0.0;
  for (size_t i = 0; i < p.used; ++i) {
    hipError_t e = hipEventSynchronize(p.pool[i].b);
    if (e != hipSuccess) return (int)e;
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, p.pool[i].a, p.pool[i].b);
    if (e != hipSuccess) return (int)e;
    tot += ms;
  }
  *total_ms = tot;
  *count = (int64_t)p.used;
  *work = p.work;
  p.used = 0;
  p.work = 0.0;
  return 0;
}
