// core.hip — version + thread-local error text for the C ABI.
#include "common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void dgtd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int dgtd_version(void) { return 100; }
extern "C" const char* dgtd_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------ opt-in call profiler
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct ProfRec { std::string key; hipEvent_t a, b; int bound; double amount; };
std::atomic<int> g_prof_on{0};
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof_recs;
}  // namespace

DgtdProfScope::DgtdProfScope(hipStream_t st_, int bound_, double amount_, const char* fmt, ...) : on(false), st(st_), bound(bound_), amount(amount_) {
  if (!g_prof_on.load(std::memory_order_relaxed)) return;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(key, sizeof(key), fmt, ap);
  va_end(ap);
  if (hipEventCreate(&a) != hipSuccess) return;
  if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return; }
  (void)hipEventRecord(a, st);
  on = true;
}

DgtdProfScope::~DgtdProfScope() {
  if (!on) return;
  (void)hipEventRecord(b, st);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_recs.push_back(ProfRec{key, a, b, bound, amount});
}

extern "C" int dgtd_profile_enable(int on) {
  if (on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof_recs.clear();
  }
  g_prof_on.store(on ? 1 : 0);
  return 0;
}

// An EMPTY bracket: the event pair with nothing between its two records.  What it measures (the record-to-record latency of two
// back-to-back events on a busy stream) is the floor every profiled call carries on top of its kernels' own duration; bench.py
// subtracts the average of a few of these from every per-call figure and says so.
extern "C" int dgtd_profile_empty(dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, 0.0, "dgtd_profile_empty");
  return 0;
}

// one line per recorded call: "key\tbound\tamount\tmilliseconds\n" (bound: hbm | mfma).  Returns the number of bytes the full dump
// needs (call with buf = NULL / cap = 0 to size it); waits for the recorded events; the records stay until the next enable(1).
extern "C" int64_t dgtd_profile_dump(char* buf, int64_t cap) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  int64_t need = 0;
  for (auto& r : g_prof_recs) {
    float ms = -1.f;
    if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) ms = -1.f;
    char line[192];
    const int n = snprintf(line, sizeof(line), "%s\t%s\t%.6g\t%.6f\n", r.key.c_str(), r.bound == DGTD_MFMA ? "mfma" : "hbm", r.amount, ms);
    if (buf && need + n < cap) memcpy(buf + need, line, n);
    need += n;
  }
  if (buf && cap > 0) buf[need < cap ? need : cap - 1] = 0;
  return need + 1;
}
