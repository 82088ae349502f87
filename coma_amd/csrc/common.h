// Shared host-side helpers for libcoma_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/coma_hip.h"

namespace coma {

char* err_buf();  // thread-local, defined in abi.hip

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(COMA_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return COMA_OK;
}

constexpr int kWave = 64;  // CDNA wavefront

// Opt a kernel into more than 64 KB of dynamic LDS.  The attribute belongs to the function object of the CURRENT DEVICE, so the
// "already done" state is kept per device (a process that drives a second GPU must opt that device in as well) and in atomics
// (two threads may race to set it: the call is idempotent).  One LdsOptIn per call site / kernel instantiation.
struct LdsOptIn {
  static constexpr int kMaxDevices = 64;
  std::atomic<size_t> bytes[kMaxDevices] = {};
};

inline int opt_in_lds(LdsOptIn& slot, const void* fn, size_t bytes, const char* who) {
  int dev = -1;
  const bool tracked = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < LdsOptIn::kMaxDevices;
  if (tracked && slot.bytes[dev].load(std::memory_order_acquire) >= bytes) return COMA_OK;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
    return fail(COMA_E_LAUNCH, "%s: cannot reserve %zu bytes of LDS", who, bytes);
  if (tracked) {
    size_t cur = slot.bytes[dev].load(std::memory_order_relaxed);
    while (cur < bytes && !slot.bytes[dev].compare_exchange_weak(cur, bytes, std::memory_order_release)) {}
  }
  return COMA_OK;
}

}  // namespace coma
