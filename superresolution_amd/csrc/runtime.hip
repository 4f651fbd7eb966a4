// Error channel, per-device immutable state (CU count) and version of libsrhip.so.
#include <mutex>

#include "srhip_common.h"

static thread_local char g_err[512] = "";

void srhip_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int srhip_check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        srhip_set_error("%s: %s", what, hipGetErrorString(e));
        return SRHIP_ERR_LAUNCH;
    }
    return SRHIP_OK;
}

namespace {
constexpr int kMaxDevices = kSrhipMaxDevices;
std::mutex g_mu;
int g_cus[kMaxDevices] = {};
}  // namespace

int srhip_current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 0;
    return dev;
}

int srhip_num_cus()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        g_cus[dev] = n;
    }
    return g_cus[dev];
}

extern "C" {
SRHIP_API int srhip_version(void) { return 1; }
SRHIP_API const char* srhip_last_error(void) { return g_err; }
}
