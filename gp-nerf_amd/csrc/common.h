// Shared host-side helpers for libgpnerf_hip.so (error text, argument checks, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/gpnerf_hip.h"

int gpn_set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define GPN_REQUIRE(cond, ...)                                  \
  do {                                                          \
    if (!(cond)) return gpn_set_error(GPN_E_ARG, __VA_ARGS__);  \
  } while (0)

#define GPN_LAUNCH_CHECK(name)                                                          \
  do {                                                                                  \
    hipError_t e_ = hipGetLastError();                                                  \
    if (e_ != hipSuccess)                                                               \
      return gpn_set_error(GPN_E_LAUNCH, "%s: %s", (name), hipGetErrorString(e_));      \
  } while (0)

static inline int gpn_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// Number of compute units of the current device (cached); grids of persistent kernels scale with it.
int gpn_num_cus();

// Optional per-stage event timing (see gpn_timing_begin in the header).  Construct right before a
// kernel launch; the destructor records the closing event after it.
class GpnStageTimer {
 public:
  GpnStageTimer(int stage, hipStream_t stream);
  ~GpnStageTimer();
  GpnStageTimer(const GpnStageTimer&) = delete;
  GpnStageTimer& operator=(const GpnStageTimer&) = delete;

 private:
  int slot_;
  unsigned session_;
  hipStream_t stream_;
};

// N = R*S points and N*V pairs are indexed with 32-bit ints in the kernels and the C-ABI.
#define GPN_REQUIRE_INT_RANGE(R, S, V, name)                                                     \
  GPN_REQUIRE((long long)(R) * (S) * ((V) > 0 ? (V) : 1) <= 2147483647LL,                        \
              "%s: R*S*V = %lld does not fit a 32-bit int (render in smaller chunks)", (name),   \
              (long long)(R) * (S) * ((V) > 0 ? (V) : 1))
#define GPN_REQUIRE_PREC(p, name)                                                                \
  GPN_REQUIRE((p) == GPN_PREC_FP32 || (p) == GPN_PREC_BF16X3 || (p) == GPN_PREC_BF16, "%s: unknown precision %d", \
              (name), (int)(p))
