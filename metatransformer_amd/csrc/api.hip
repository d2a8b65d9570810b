// api.hip -- error plumbing and device queries of the C ABI (include/metaenc.h).
#include "common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void me_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int me_abi_version(void) { return ME_ABI_VERSION; }
extern "C" const char* me_last_error(void) { return g_err; }
extern "C" const char* me_build_arch(void) { return "gfx950"; }

extern "C" int me_device_info(int dev, int* num_cus, int* lds_bytes, int* clock_mhz, char* name, int name_len) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) {
        me_set_error("me_device_info: %s", hipGetErrorString(e));
        return ME_ERR_HIP;
    }
    if (num_cus) *num_cus = prop.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (clock_mhz) *clock_mhz = prop.clockRate / 1000;
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    return ME_OK;
}
