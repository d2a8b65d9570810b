// energy_probe -- marginal energy of the chip's basic activities, from socket power under sustained single-activity kernels (dev tool, round 6).
// DESIGN section 4.2 says config 2 is bound by joules per step; this puts numbers on where a joule goes:
//   idle                         socket power with nothing running                               -> P_idle
//   mfma   v_mfma_f32_16x16x32_bf16 on registers only, 8 waves per CU (what the GEMM K-loop issues) -> pJ / flop, clock held
//   lds    ds_read_b128 streams, no MFMA (the K-loop's fragment reads)                              -> pJ / byte
//   l2     buffer_load_dwordx4 over 64 KiB per CU (L2-resident)                                     -> pJ / byte
//   hbm_r  the same over 1 GiB (past L2 and the Infinity Cache)                                     -> pJ / byte
//   hbm_w  buffer_store_dwordx4 over 1 GiB                                                          -> pJ / byte
// Each activity loops for ~1.5 s of back-to-back launches while a host thread samples hwmon power1_average every 10 ms (the first 0.3 s are
// not sampled: DVFS settling).  Marginal energy = (P - P_idle) / rate -- an upper bound on the activity's own cost (it includes the clock
// tree / scheduler power any busy CU draws).
//   hipcc -O2 --offload-arch=gfx950 tools/energy_probe.hip -o tools/_build/energy_probe -lpthread && tools/_build/energy_probe
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <glob.h>
#include <string>
#include <thread>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Power {
    std::vector<std::string> pw;
    std::atomic<bool> stop{false};
    std::thread th;
    double sum = 0, mx = 0; int n = 0;
    Power() {
        glob_t g;
        for (const char* pat : {"/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"}) {
            if (glob(pat, 0, nullptr, &g) == 0) for (size_t i = 0; i < g.gl_pathc; ++i) pw.push_back(g.gl_pathv[i]);
            globfree(&g);
            if (!pw.empty()) break;
        }
    }
    double read() {
        double w = 0;
        for (auto& f : pw) { FILE* fp = fopen(f.c_str(), "r"); long long v; if (fp) { if (fscanf(fp, "%lld", &v) == 1 && v * 1e-6 > w) w = v * 1e-6; fclose(fp); } }
        return w;
    }
    void start() { stop = false; sum = mx = 0; n = 0; th = std::thread([this] { while (!stop) { double w = read(); if (w > 0) { sum += w; ++n; if (w > mx) mx = w; } std::this_thread::sleep_for(std::chrono::milliseconds(10)); } }); }
    void finish() { stop = true; if (th.joinable()) th.join(); }
    double mean() const { return n ? sum / n : 0; }
};

__global__ __launch_bounds__(512) void mfma_kernel(int iters, float* out, unsigned long long* clk) {
    f32x4 acc[16];
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(1.0f - 0.002f * (threadIdx.x * 3 + i)); }
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (s == 1.2345e-30f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

__global__ __launch_bounds__(512) void lds_kernel(int iters, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<unsigned*>(smem)[i] = i * 2654435761u;
    __syncthreads();
    u32x4 acc = {0u, 0u, 0u, 0u};
    const unsigned base = (unsigned)(size_t)smem + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
    for (int it = 0; it < iters; ++it) {
        u32x4 v0, v1, v2, v3;
        asm volatile("ds_read_b128 %0, %4 offset:0\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                     : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(base) : "memory");
        acc ^= v0 ^ v1 ^ v2 ^ v3;
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[0] = 1.f;
}

template <bool STORE>
__global__ __launch_bounds__(512) void mem_kernel(char* base, size_t per_wg, int iters, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* mine = base + (size_t)blockIdx.x * per_wg;
    const size_t chunks = per_wg >> 10;
    u32x4 acc = {0u, 0u, 0u, 0u};
    size_t c = wave;
    for (int it = 0; it < iters; it += 4) {
        u32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            u32x4* p = reinterpret_cast<u32x4*>(mine + ((c % chunks) << 10) + lane * 16);
            if (STORE) { const u32x4 w = {(unsigned)it, (unsigned)lane, 3u, 4u}; __builtin_nontemporal_store(w, p); }
            else v[k] = __builtin_nontemporal_load(p);
            c += 8;
        }
        if (!STORE) acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (!STORE && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[0] = 1.f;
}

template <typename F> static void sustain(const char* name, Power& pw, double p_idle, double units_per_launch, const char* unit, double scale, F launch, unsigned long long* clk_dev = nullptr) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    // calibrate one launch
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms1 = 0; (void)hipEventElapsedTime(&ms1, e0, e1);
    const int n_warm = (int)(300.0 / ms1) + 1, n_loop = (int)(1500.0 / ms1) + 1;
    for (int i = 0; i < n_warm; ++i) launch();
    (void)hipDeviceSynchronize();
    pw.start();
    (void)hipEventRecord(e0);
    for (int i = 0; i < n_loop; ++i) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    pw.finish();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double rate = units_per_launch * n_loop / (ms * 1e-3);          // units per second
    const double w = pw.mean();
    double ghz = 0;
    if (clk_dev) { unsigned long long h[2]; (void)hipMemcpy(h, clk_dev, 16, hipMemcpyDeviceToHost); if (h[1]) ghz = (double)h[0] / ((double)h[1] * 10.0); }
    printf("%-8s %8.1f W (max %6.1f)  %10.2f %s  marginal %7.3f pJ per %s%s", name, w, pw.mx, rate * scale, unit, (w - p_idle) / rate * 1e12, unit[0] == 'T' && unit[1] == 'F' ? "flop" : "byte",
           ghz > 0 ? "" : "\n");
    if (ghz > 0) printf("  clock held %.2f GHz\n", ghz);
    fflush(stdout);
}

int main() {
    int cus = 256, dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    Power pw;
    if (pw.pw.empty()) { printf("no hwmon power file readable\n"); return 1; }
    float* out; unsigned long long* clk; char* big;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&clk, 64); (void)hipMemset(clk, 0, 64);
    const size_t GIB = (size_t)1 << 30;
    (void)hipMalloc(&big, GIB); (void)hipMemset(big, 1, GIB);
    (void)hipDeviceSynchronize();
    std::this_thread::sleep_for(std::chrono::milliseconds(500));
    pw.start(); std::this_thread::sleep_for(std::chrono::milliseconds(1500)); pw.finish();
    const double p_idle = pw.mean();
    printf("energy_probe: %d CUs; socket power by hwmon, 10 ms samples over ~1.5 s of back-to-back launches per activity\n", cus);
    printf("idle     %8.1f W\n", p_idle);
    const int mi = 20000;
    sustain("mfma", pw, p_idle, (double)cus * 8 * mi * 16 * 16384.0, "TF/s", 1e-12, [&] { mfma_kernel<<<cus, 512>>>(mi, out, clk); }, clk);
    (void)hipFuncSetAttribute((const void*)lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int li = 20000;
    sustain("lds", pw, p_idle, (double)cus * 512 * li * 64.0, "TB/s", 1e-12, [&] { lds_kernel<<<cus, 512, 65536>>>(li, out); });
    const int l2i = 4096;
    sustain("l2", pw, p_idle, (double)cus * 8 * l2i * 1024.0, "TB/s", 1e-12, [&] { mem_kernel<false><<<cus, 512>>>(big, 65536, l2i, out); });
    const size_t per = GIB / cus;
    const int hi = (int)(per >> 10) / 8 * 2;        // two passes over the workgroup's 4 MiB
    sustain("hbm_r", pw, p_idle, (double)cus * 8 * hi * 1024.0, "TB/s", 1e-12, [&] { mem_kernel<false><<<cus, 512>>>(big, per, hi, out); });
    sustain("hbm_w", pw, p_idle, (double)cus * 8 * hi * 1024.0, "TB/s", 1e-12, [&] { mem_kernel<true><<<cus, 512>>>(big, per, hi, out); });
    return 0;
}
