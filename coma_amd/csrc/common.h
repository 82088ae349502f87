// Shared host-side helpers for libcoma_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
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

}  // namespace coma
