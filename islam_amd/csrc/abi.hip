// libislam_hip.so: error buffer and ABI version.
#include "common.h"

namespace islam {
char* err_buf() {
    static thread_local char buf[512] = "";
    return buf;
}
}  // namespace islam

extern "C" const char* islam_last_error(void) { return islam::err_buf(); }
extern "C" int islam_abi_version(void) { return 1; }
