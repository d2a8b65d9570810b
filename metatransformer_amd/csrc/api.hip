// api.hip -- error plumbing and device queries of the C ABI (include/metaenc.h).
#include "common.h"
#include <stdarg.h>
#include <string.h>
#include <mutex>
#include <vector>

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

// ---- optional per-launch timing
namespace {
struct ProfEntry { int op, dt; int64_t M, N, K; hipEvent_t e0, e1; int plan; };
std::mutex g_prof_mu;
std::vector<ProfEntry> g_prof;
bool g_prof_on = false;
}  // namespace

ProfScope::ProfScope(int op_, int dt_, int64_t M_, int64_t N_, int64_t K_, hipStream_t s) : stream(s), op(op_), dt(dt_), M(M_), N(N_), K(K_) {
    if (!g_prof_on) return;
    if (hipEventCreate(&e0) != hipSuccess) { e0 = nullptr; return; }
    (void)hipEventRecord(e0, stream);
}
ProfScope::~ProfScope() {
    if (!e0) return;
    hipEvent_t e1 = nullptr;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return; }
    (void)hipEventRecord(e1, stream);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(ProfEntry{op, dt, M, N, K, e0, e1, plan});
}

extern "C" int me_gemm_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& e : g_prof) { (void)hipEventDestroy(e.e0); (void)hipEventDestroy(e.e1); }
    g_prof.clear();
    g_prof_on = on != 0;
    return ME_OK;
}

extern "C" int me_gemm_profile_read(me_gemm_profile_rec* out, int max) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = 0;
    for (auto& e : g_prof) {
        float ms = 0.f;
        (void)hipEventSynchronize(e.e1);
        (void)hipEventElapsedTime(&ms, e.e0, e.e1);
        if (out && n < max) out[n] = me_gemm_profile_rec{e.op, e.dt, e.M, e.N, e.K, ms, e.plan};
        ++n;
        (void)hipEventDestroy(e.e0);
        (void)hipEventDestroy(e.e1);
    }
    g_prof.clear();
    return n;
}

// ---- dev build only: kernel-family / debug switches for A/B runs (tools/gemm_dev).  Not part of the C ABI: the symbol
// does not exist in the shipped library.
#ifdef ME_DEV
#include "gemm_common.h"
GemmDev g_gemm_dev = {-1, 0, 0, 1, 1};
extern "C" int me_dev_set(const char* key, int value) {
    if (!strcmp(key, "family")) g_gemm_dev.family = value;
    else if (!strcmp(key, "bn")) g_gemm_dev.bn = value;
    else if (!strcmp(key, "debug")) g_gemm_dev.debug = value;
    else if (!strcmp(key, "tail_split")) g_gemm_dev.tail_split = value;
    else if (!strcmp(key, "g3_persistent")) g_gemm_dev.g3_persistent = value;
    else return ME_ERR_ARG;
    return ME_OK;
}
void* g_gemm_dev_trace = nullptr;
extern "C" int me_dev_set_trace(void* buf) { g_gemm_dev_trace = buf; return ME_OK; }
#endif
