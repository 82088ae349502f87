// libcoma_hip.so: ABI version + thread-local error text.
#include "common.h"

namespace coma {
char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace coma

extern "C" int coma_abi_version(void) { return COMA_ABI_VERSION; }
extern "C" const char* coma_last_error(void) { return coma::err_buf(); }
