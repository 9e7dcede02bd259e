// Profiling probes: hipEvent pairs recorded on the launch stream around selected launches.
// Thread-safe: fit_many (dvt_amd/fit.py) drives dvt_fit_run_batched from up to four host threads with the probes enabled, so
// a probe's sample pool is guarded by a mutex and every scope carries the (generation, index) of ITS sample (an a/b pair is
// always recorded by the one scope that drew it, whatever the interleaving of the threads).  The generation counts the resets
// (dvt_prof_enable / dvt_prof_collect): a scope that was open across a reset finds a different generation at its end and
// drops it -- its old index may already belong to another scope's sample again.
#include <mutex>
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
};
Probe g_probes[DVT_N_PROBES];
std::mutex g_prof_mu;
unsigned long long g_prof_gen = 1;  // bumped by every reset of the sample pools
constexpr size_t MAX_SAMPLES = 200000;
}  // namespace

// returns the sample index of this scope (-1 if none could be drawn) and the generation it belongs to
long dvt_prof_begin(int probe, hipStream_t s, unsigned long long* gen) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  Probe& p = g_probes[probe];
  if (p.used >= MAX_SAMPLES) return -1;
  if (p.used == p.pool.size()) {
    Sample sm;
    if (hipEventCreate(&sm.a) != hipSuccess) return -1;
    if (hipEventCreate(&sm.b) != hipSuccess) {
      (void)hipEventDestroy(sm.a);
      return -1;
    }
    p.pool.push_back(sm);
  }
  const long idx = (long)p.used++;
  *gen = g_prof_gen;
  // both events are recorded here and now: a sample whose scope never ends (or whose end loses the race with a
  // dvt_prof_enable reset) still reads as a valid, zero-length pair
  (void)hipEventRecord(p.pool[idx].a, s);
  (void)hipEventRecord(p.pool[idx].b, s);
  return idx;
}

void dvt_prof_end(int probe, long idx, unsigned long long gen, hipStream_t s, double work) {
  if (idx < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  Probe& p = g_probes[probe];
  if (gen != g_prof_gen || (size_t)idx >= p.used) return;  // the probes were reset while this scope was open
  (void)hipEventRecord(p.pool[idx].b, s);
  p.work += work;
}

extern "C" int dvt_prof_enable(unsigned mask) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_dvt_prof_mask = mask & ((1u << DVT_N_PROBES) - 1u);
  ++g_prof_gen;
  for (auto& p : g_probes) {
    p.used = 0;
    p.work = 0.0;
  }
  return 0;
}

extern "C" int dvt_prof_collect(int probe, double* total_ms, int64_t* count, double* work) {
  if (probe < 0 || probe >= DVT_N_PROBES || !total_ms || !count || !work) return DVT_E_BADARG;
  std::lock_guard<std::mutex> lk(g_prof_mu);
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
  // this probe's pool is recycled: scopes of ANY probe that are open right now are dropped at their end (a zero-length,
  // valid pair each) rather than tracked per probe -- collecting while launches are in flight is not a supported measurement
  ++g_prof_gen;
  p.used = 0;
  p.work = 0.0;
  return 0;
}
