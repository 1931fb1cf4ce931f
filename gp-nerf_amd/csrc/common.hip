// Error reporting + device query for libgpnerf_hip.so.
#include "common.h"

#include <cstdarg>
#include <atomic>
#include <cstdio>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

int gpn_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int gpn_num_cus() {
  static thread_local int cached_dev = -1, cached = 256;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (dev != cached_dev) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cached = n;
    cached_dev = dev;
  }
  return cached;
}

extern "C" int gpn_abi_version(void) { return GPN_ABI_VERSION; }
extern "C" const char* gpn_last_error(void) { return g_err; }

// ---------------------------------------------------------------- per-stage event timing
namespace {
struct TimingRec {
  hipEvent_t a, b;
  int stage;
};
// The pool is the library's only mutable process-wide state; g_mu guards it.  g_session counts the
// sessions opened so far: a timer that outlives its session (gpn_timing_end on another thread between
// its constructor and destructor) must not touch the next session's events.
std::mutex g_mu;
std::vector<TimingRec> g_pool;
int g_used = 0;
std::atomic<unsigned> g_mask{0};
unsigned g_session = 0;
}  // namespace

GpnStageTimer::GpnStageTimer(int stage, hipStream_t stream) : slot_(-1), session_(0), stream_(stream) {
  if (g_mask.load(std::memory_order_relaxed) == 0) return;  // no session open (the common case): no lock taken on the launch path
  std::lock_guard<std::mutex> lk(g_mu);
  if (((g_mask.load() >> stage) & 1u) && g_used < (int)g_pool.size()) {
    slot_ = g_used++;
    session_ = g_session;
    g_pool[slot_].stage = stage;
    (void)hipEventRecord(g_pool[slot_].a, stream_);
  }
}
GpnStageTimer::~GpnStageTimer() {
  if (slot_ < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (session_ == g_session && slot_ < (int)g_pool.size()) (void)hipEventRecord(g_pool[slot_].b, stream_);
}

extern "C" int gpn_timing_begin(unsigned stage_mask, int max_records) {
  std::lock_guard<std::mutex> lk(g_mu);
  GPN_REQUIRE(g_pool.empty(), "gpn_timing_begin: a timing session is already open");
  GPN_REQUIRE(max_records > 0 && max_records <= (1 << 20), "gpn_timing_begin: bad max_records");
  g_pool.resize(max_records);
  for (size_t i = 0; i < g_pool.size(); ++i) {
    TimingRec& r = g_pool[i];
    r.a = r.b = nullptr;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) {
      for (size_t k = 0; k <= i; ++k) {  // release what was created: the session did not open
        if (g_pool[k].a) (void)hipEventDestroy(g_pool[k].a);
        if (g_pool[k].b) (void)hipEventDestroy(g_pool[k].b);
      }
      g_pool.clear();
      return gpn_set_error(GPN_E_LAUNCH, "gpn_timing_begin: hipEventCreate failed");
    }
  }
  g_used = 0;
  ++g_session;
  g_mask = stage_mask;
  return GPN_OK;
}

extern "C" int gpn_timing_end(double* ms_per_stage, long long* launches_per_stage) {
  GPN_REQUIRE(ms_per_stage && launches_per_stage, "gpn_timing_end: null pointer");
  std::lock_guard<std::mutex> lk(g_mu);
  g_mask = 0;
  ++g_session;  // timers still alive belong to the session that ends here
  for (int i = 0; i < GPN_NSTAGES; ++i) ms_per_stage[i] = 0.0, launches_per_stage[i] = 0;
  int rc = GPN_OK;
  for (int i = 0; i < g_used; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(g_pool[i].b) != hipSuccess ||
        hipEventElapsedTime(&ms, g_pool[i].a, g_pool[i].b) != hipSuccess) {
      rc = gpn_set_error(GPN_E_LAUNCH, "gpn_timing_end: event query failed");
      break;
    }
    ms_per_stage[g_pool[i].stage] += ms;
    launches_per_stage[g_pool[i].stage] += 1;
  }
  for (auto& r : g_pool) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_pool.clear();
  g_used = 0;
  return rc;
}
