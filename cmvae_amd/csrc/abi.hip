// Error reporting and version for libcmvae_hip.so.
#include "common.h"
#include <string>

namespace cmvae {
static thread_local std::string g_last_error;

int set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return -1;
}
}  // namespace cmvae

extern "C" const char* cmvae_last_error(void) { return cmvae::g_last_error.c_str(); }
extern "C" int cmvae_abi_version(void) {
    cmvae::clear_stale_error(); return 1; }
