// gemm_dev -- torch-free A/B bench + full-tensor check of me_gemm kernel families on the encoder's shapes (dev tool).
//
//   python -m metatransformer_amd.build --dev      # builds tools/_build/libmetaenc_dev.so and tools/_build/gemm_dev
//   tools/_build/gemm_dev [--iters N] [--check] case [case ...]
//   case  = family:M:N:K:epi[:debug]      family in {auto, g128, g2b, g2w, g3, g3x, g3p, g3s, g3t, g3f}; epi 0 bias, 1 gelu(+preact), 2 residual,
//           3 gelu'(aux), 6 * aux (ME_GEMM_AUX_IS_FACTOR), 7 gelu + saved gelu' (ME_GEMM_SAVE_GELU_GRAD), 9 fp32 residual -> fp32 output, 8 residual + row statistics
//           (me_gemm_desc.row_stats); debug = GemmDev::debug bits
//           tn-family:M:N:K               wgrad form C[M, N] = A[K, M]^T B[K, N] (fp32 output), family in {auto, g2b, g3}
//
// Every case is checked (all M x N outputs) against a straightforward fp32 kernel on the same bf16 operands, then timed
// over rotating operand / output sets (3 copies: the 256 MiB Infinity Cache must not hold them -- as inside the model).
// Operands are random full-range values (never zeros: MI355X clocks ~20 % higher on zero-filled operands).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <glob.h>
#include <string>
#include <thread>
#include <vector>
#include "../include/metaenc.h"

extern "C" int me_dev_set(const char* key, int value);
extern "C" int me_dev_set_trace(void* buf);

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
            exit(2);                                                                            \
        }                                                                                       \
    } while (0)

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

__global__ void fill_kernel(uint16_t* p, size_t n, uint64_t seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = seed + i * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        // sum of two uniforms in [-1, 1): full-range signs and exponents
        const float a = (float)(uint32_t)(z >> 40) * (1.0f / 8388608.0f) - 1.0f;
        const float b = (float)(uint32_t)((z >> 16) & 0xffffff) * (1.0f / 8388608.0f) - 1.0f;
        p[i] = f2bf(scale * (a + b));
    }
}
__global__ void fill_f32_kernel(float* p, size_t n, uint64_t seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = seed + i * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z ^= z >> 29;
        p[i] = scale * ((float)(uint32_t)(z >> 40) * (1.0f / 8388608.0f) - 1.0f);
    }
}

// reference: one thread per output, fp32 accumulate in k order; the epilogue in exact libm arithmetic
__global__ void ref_kernel(const uint16_t* A, const uint16_t* B, const float* bias, const uint16_t* rowop, int epi, int64_t M,
                           int64_t N, int64_t K, float* out, float* out_pre) {
    const int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    const uint16_t* a = A + m * K;
    const uint16_t* b = B + n * K;
    float acc = 0.f;
    for (int64_t k = 0; k < K; ++k) acc = fmaf(bf2f(a[k]), bf2f(b[k]), acc);
    float v = acc + (bias ? bias[n] : 0.f);
    if (epi == 1 || epi == 7) {
        out_pre[m * N + n] = epi == 7 ? 0.5f * (1.0f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * expf(-0.5f * v * v) : v;
        v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    } else if (epi == 6) {
        v *= bf2f(rowop[m * N + n]);
    } else if (epi == 2 || epi == 8 || epi == 9) {
        v += bf2f(rowop[m * N + n]);
    } else if (epi == 3) {
        const float x = bf2f(rowop[m * N + n]);
        v *= 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
    }
    out[m * N + n] = v;
}

__global__ void expand_f32_kernel(const uint16_t* src, float* dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = bf2f(src[i]);
}

__global__ void ref_tn_kernel(const uint16_t* A, const uint16_t* B, int64_t M, int64_t N, int64_t K, float* out) {
    const int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float acc = 0.f;
    for (int64_t k = 0; k < K; ++k) acc = fmaf(bf2f(A[k * M + m]), bf2f(B[k * N + n]), acc);
    out[m * N + n] = acc;
}
__global__ void cmp_f32_kernel(const float* got, const float* ref, size_t n, float atol, float rtol, unsigned long long* nbad, float* maxerr) {
    float me = 0.f;
    unsigned long long bad = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float e = fabsf(got[i] - ref[i]);
        if (!(e <= atol + rtol * fabsf(ref[i]))) ++bad;
        me = fmaxf(me, e / (1.0f + fabsf(ref[i])));
    }
    if (bad) atomicAdd(nbad, bad);
    atomicMax(reinterpret_cast<unsigned int*>(maxerr), __float_as_uint(me));
}

__global__ void cmp_kernel(const uint16_t* got, const float* ref, size_t n, float atol, float rtol, unsigned long long* nbad,
                           float* maxerr) {
    float me = 0.f;
    unsigned long long bad = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float g = bf2f(got[i]), r = ref[i];
        const float e = fabsf(g - r);
        if (!(e <= atol + rtol * fabsf(r))) ++bad;
        me = fmaxf(me, e / (1.0f + fabsf(r)));
    }
    if (bad) atomicAdd(nbad, bad);
    atomicMax(reinterpret_cast<unsigned int*>(maxerr), __float_as_uint(me));
}

// ---- socket power / shader clock next to a timed loop (--power SECONDS): the amdgpu hwmon files of the visible device, sampled
// every 20 ms by a host thread while the launches run ("the power-bound claim is half-instrumented").  Energy per
// flop = mean power x time / flops of the loop.  Files that do not exist on a box simply report nothing.
struct PowerProbe {
    std::vector<std::string> pw, fq;
    std::atomic<bool> stop{false};
    std::thread th;
    double sum_w = 0, sum_mhz = 0, max_w = 0;
    int n_w = 0, n_f = 0;
    static std::vector<std::string> find(const char* pat) {
        std::vector<std::string> out;
        glob_t g;
        if (glob(pat, 0, nullptr, &g) == 0) for (size_t i = 0; i < g.gl_pathc; ++i) out.push_back(g.gl_pathv[i]);
        globfree(&g);
        return out;
    }
    static bool read_ll(const std::string& f, long long& v) {
        FILE* fp = fopen(f.c_str(), "r");
        if (!fp) return false;
        const bool ok = fscanf(fp, "%lld", &v) == 1;
        fclose(fp);
        return ok;
    }
    PowerProbe() {
        pw = find("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average");
        if (pw.empty()) pw = find("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input");
        fq = find("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input");
    }
    void start() {
        stop = false; sum_w = sum_mhz = max_w = 0; n_w = n_f = 0;
        th = std::thread([this] {
            while (!stop) {
                long long v;
                double w = 0; bool any = false;
                for (auto& f : pw) if (read_ll(f, v) && v > 0) { w = v * 1e-6 > w ? v * 1e-6 : w; any = true; }    // the busiest visible socket
                if (any) { sum_w += w; ++n_w; max_w = w > max_w ? w : max_w; }
                double mhz = 0; any = false;
                for (auto& f : fq) if (read_ll(f, v) && v > 0) { mhz = v * 1e-6 > mhz ? v * 1e-6 : mhz; any = true; }
                if (any) { sum_mhz += mhz; ++n_f; }
                std::this_thread::sleep_for(std::chrono::milliseconds(20));
            }
        });
    }
    void finish() { stop = true; if (th.joinable()) th.join(); }
};

static int family_code(const std::string& f) {
    if (f == "auto") return -1;
    if (f == "g128") return 0;
    if (f == "g2b" || f == "g2bn" || f == "g2bw" || f == "g2bs") return 2;      // g2bn / g2bw: 128 x 128 / 128 x 256 tiles, NO split-K (small-M planning A/B)
    if (f == "g2w") return 3;
    if (f == "g3" || f == "g3x" || f == "g3p" || f == "g3t" || f == "g3s" || f == "g3f") return 4;      // g3f: resident, whole tiles only (no 128-row items in the last round)      // g3: shipped form (resident); g3t: one tile per workgroup; g3x: without the tail split; g3p: persistent stream-K
    fprintf(stderr, "unknown family %s\n", f.c_str());
    exit(2);
}

int main(int argc, char** argv) {
    int iters = 20;
    bool check = false;
    double power_secs = 0;
    std::vector<std::string> cases;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--iters")) iters = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--check")) check = true;
        else if (!strcmp(argv[i], "--power")) power_secs = atof(argv[++i]);
        else cases.push_back(argv[i]);
    }
    constexpr int NSET = 3;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (const std::string& c0 : cases) {
        if (c0.rfind("tn-", 0) == 0) {
            // ---- wgrad-shaped case
            const std::string c = c0.substr(3);
            char fam[32];
            long long M, N, K;
            int debug = 0;
            if (sscanf(c.c_str(), "%31[^:]:%lld:%lld:%lld:%d", fam, &M, &N, &K, &debug) < 4) { fprintf(stderr, "bad case %s\n", c0.c_str()); return 2; }
            me_dev_set("family", family_code(fam));
            me_dev_set("debug", debug);
            uint16_t *A[NSET], *B[NSET];
            float* C[NSET];
            for (int s = 0; s < NSET; ++s) {
                CK(hipMalloc(&A[s], (size_t)K * M * 2)); CK(hipMalloc(&B[s], (size_t)K * N * 2)); CK(hipMalloc(&C[s], (size_t)M * N * 4));
                fill_kernel<<<2048, 256, 0, st>>>(A[s], (size_t)K * M, 31, 0.05f);
                fill_kernel<<<2048, 256, 0, st>>>(B[s], (size_t)K * N, 77, 1.0f);
                CK(hipMemsetAsync(C[s], 0xff, (size_t)M * N * 4, st));
            }
            me_gemm_desc d;
            memset(&d, 0, sizeof(d));
            d.op = ME_GEMM_TN; d.ab_dtype = ME_BF16; d.M = M; d.N = N; d.K = K;
            d.lda = M; d.ldb = N; d.ldc = N; d.c_dtype = ME_F32; d.alpha = 1.0f; d.beta = 0.0f;
            d.A = A[0]; d.B = B[0]; d.C = C[0];
            float* csum = nullptr;
            CK(hipMalloc(&csum, (size_t)M * 4));
            if (me_gemm_fuses_colsum(&d)) d.colsum_a = csum;
            const size_t wsb = me_gemm_workspace_bytes(&d);
            void* ws = nullptr;
            if (wsb) CK(hipMalloc(&ws, wsb));
            d.workspace = ws; d.workspace_bytes = (int64_t)wsb;
            auto run = [&](int s) {
                d.A = A[s]; d.B = B[s]; d.C = C[s];
                const int rc = me_gemm(&d, st);
                if (rc) { fprintf(stderr, "me_gemm failed (%d): %s\n", rc, me_last_error()); exit(3); }
            };
            for (int s = 0; s < NSET; ++s) run(s);
            CK(hipStreamSynchronize(st));
            std::string verdict = "unchecked";
            if (check && !(debug & 1)) {
                float* ref;
                CK(hipMalloc(&ref, (size_t)M * N * 4));
                ref_tn_kernel<<<dim3((unsigned)((N + 255) / 256), (unsigned)M), 256, 0, st>>>(A[0], B[0], M, N, K, ref);
                unsigned long long* nbad; float* maxerr;
                CK(hipMalloc(&nbad, 8)); CK(hipMalloc(&maxerr, 4));
                verdict.clear();
                for (int s = 0; s < NSET; s += NSET - 1) {
                    CK(hipMemsetAsync(nbad, 0, 8, st)); CK(hipMemsetAsync(maxerr, 0, 4, st));
                    // fp32 accumulation over K in a different order: |err| ~ 1e-6 * sqrt(K) * |terms|
                    cmp_f32_kernel<<<1024, 256, 0, st>>>(C[s], ref, (size_t)M * N, 2e-2f, 2e-3f, nbad, maxerr);
                    unsigned long long hb; float hm;
                    CK(hipMemcpyAsync(&hb, nbad, 8, hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(&hm, maxerr, 4, hipMemcpyDeviceToHost, st));
                    CK(hipStreamSynchronize(st));
                    char buf[128];
                    snprintf(buf, sizeof(buf), "%s[set%d bad=%llu maxerr=%.3g]%s", verdict.empty() ? "" : " ", s, hb, hm, hb ? " MISMATCH" : "");
                    verdict += buf;
                }
                CK(hipFree(ref)); CK(hipFree(nbad)); CK(hipFree(maxerr));
                if (d.colsum_a) {           // the fused bias gradient: column sums of A, against a host sum of set 0
                    run(0);
                    std::vector<uint16_t> ha((size_t)K * M);
                    std::vector<float> hc(M);
                    CK(hipMemcpyAsync(ha.data(), A[0], ha.size() * 2, hipMemcpyDeviceToHost, st));
                    CK(hipMemcpyAsync(hc.data(), csum, (size_t)M * 4, hipMemcpyDeviceToHost, st));
                    CK(hipStreamSynchronize(st));
                    double worst = 0;
                    for (long long m = 0; m < M; ++m) {
                        double acc = 0;
                        for (long long k = 0; k < K; ++k) { uint32_t u = (uint32_t)ha[(size_t)k * M + m] << 16; float f; memcpy(&f, &u, 4); acc += f; }
                        worst = fmax(worst, fabs(acc - hc[m]) / (1.0 + fabs(acc)));
                    }
                    char buf[64];
                    snprintf(buf, sizeof(buf), " [colsum maxerr=%.3g%s]", worst, worst < 2e-3 ? "" : " MISMATCH");
                    verdict += buf;
                }
            }
            for (int i = 0; i < 3; ++i) run(i % NSET);
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) run(i % NSET);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = 1e3 * ms / iters, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
            CK(hipFree(csum));
            printf("%-34s %8.1f us  %7.1f TF/s  %s\n", c0.c_str(), us, tf, verdict.c_str());
            fflush(stdout);
            for (int s = 0; s < NSET; ++s) { CK(hipFree(A[s])); CK(hipFree(B[s])); CK(hipFree(C[s])); }
            if (ws) CK(hipFree(ws));
            continue;
        }
        const std::string& c = c0;
        char fam[32];
        long long M, N, K;
        int epi = 0, debug = 0;
        if (sscanf(c.c_str(), "%31[^:]:%lld:%lld:%lld:%d:%d", fam, &M, &N, &K, &epi, &debug) < 4) {
            fprintf(stderr, "bad case %s\n", c.c_str());
            return 2;
        }
        me_dev_set("family", family_code(fam));
        me_dev_set("bn", strcmp(fam, "g2bn") == 0 ? 128 : strcmp(fam, "g2bw") == 0 ? 256 : strcmp(fam, "g2bs") == 0 ? 64 : 0);      // g2bs: 64 x 128 tiles
        me_dev_set("g3_persistent", strcmp(fam, "g3p") == 0 ? 2 : strcmp(fam, "g3t") == 0 ? 0 : 1);
        me_dev_set("tail_split", strcmp(fam, "g3s") == 0 ? 2 : strcmp(fam, "g3f") == 0 ? 3 : (strcmp(fam, "g3x") != 0 && strcmp(fam, "g2bn") != 0 && strcmp(fam, "g2bw") != 0 && strcmp(fam, "g2bs") != 0));      // g3s: resident, static schedule; g3f: no half items
        me_dev_set("debug", debug);
        uint16_t *A[NSET], *C[NSET], *P[NSET], *Bw, *rowop = nullptr;
        float *bias, *ref = nullptr, *ref_pre = nullptr;
        for (int s = 0; s < NSET; ++s) {
            CK(hipMalloc(&A[s], (size_t)M * K * 2));
            CK(hipMalloc(&C[s], (size_t)M * N * (epi == 9 ? 4 : 2)));
            P[s] = nullptr;
            if (epi == 1 || epi == 7) CK(hipMalloc(&P[s], (size_t)M * N * 2));
            fill_kernel<<<2048, 256, 0, st>>>(A[s], (size_t)M * K, 17, 1.0f);     // same content in every set
            CK(hipMemsetAsync(C[s], 0xff, (size_t)M * N * (epi == 9 ? 4 : 2), st));
        }
        CK(hipMalloc(&Bw, (size_t)N * K * 2));
        CK(hipMalloc(&bias, (size_t)N * 4));
        fill_kernel<<<2048, 256, 0, st>>>(Bw, (size_t)N * K, 99, 0.05f);
        fill_f32_kernel<<<64, 256, 0, st>>>(bias, (size_t)N, 5, 0.5f);
        float* rowop32 = nullptr;
        if (epi == 2 || epi == 3 || epi == 6 || epi == 8 || epi == 9) {
            CK(hipMalloc(&rowop, (size_t)M * N * 2));
            fill_kernel<<<2048, 256, 0, st>>>(rowop, (size_t)M * N, 1234, 1.0f);
        }
        if (epi == 9) {           // bias + fp32 residual -> fp32 (an fp32 token stream: autocast recipes, ME_BF16X3 blocks)
            CK(hipMalloc(&rowop32, (size_t)M * N * 4));
            expand_f32_kernel<<<2048, 256, 0, st>>>(rowop, rowop32, (size_t)M * N);
        }
        me_gemm_desc d;
        memset(&d, 0, sizeof(d));
        d.op = ME_GEMM_NT; d.ab_dtype = ME_BF16; d.M = M; d.N = N; d.K = K;
        d.lda = K; d.B = Bw; d.ldb = K; d.ldc = N; d.c_dtype = ME_BF16;
        d.alpha = 1.0f; d.beta = 0.0f; d.bias = bias;
        if (epi == 1 || epi == 7) { d.act = ME_ACT_GELU; d.ldpre = N; d.preact_dtype = ME_BF16; }
        if (epi == 7) d.flags = ME_GEMM_SAVE_GELU_GRAD;
        if (epi == 6) { d.aux = rowop; d.ldaux = N; d.aux_dtype = ME_BF16; d.flags = ME_GEMM_AUX_IS_FACTOR; }
        if (epi == 2 || epi == 8) { d.residual = rowop; d.ldres = N; d.res_dtype = ME_BF16; }
        if (epi == 9) { d.residual = rowop32; d.ldres = N; d.res_dtype = ME_F32; d.c_dtype = ME_F32; }
        float* rstats = nullptr;
        if (epi == 3) { d.aux = rowop; d.ldaux = N; d.aux_dtype = ME_BF16; }
        d.A = A[0]; d.C = C[0]; d.preact = P[0];
        if (epi == 8) {           // residual + per-row statistics of the output on the side (me_gemm_desc.row_stats)
            if (!me_gemm_emits_row_stats(&d)) { fprintf(stderr, "case %s: this shape does not emit row statistics: %s\n", c.c_str(), me_last_error()); return 2; }
            CK(hipMalloc(&rstats, me_row_stats_partial_bytes(M, (int)N)));
            d.row_stats = rstats;
        }
        const size_t wsb = me_gemm_workspace_bytes(&d);
        void* ws = nullptr;
        if (wsb) CK(hipMalloc(&ws, wsb));
        d.workspace = ws; d.workspace_bytes = (int64_t)wsb;
        auto run = [&](int s) {
            d.A = A[s]; d.C = C[s]; d.preact = P[s];
            const int rc = me_gemm(&d, st);
            if (rc) { fprintf(stderr, "me_gemm failed (%d): %s\n", rc, me_last_error()); exit(3); }
        };
        for (int s = 0; s < NSET; ++s) run(s);
        CK(hipStreamSynchronize(st));
        std::string verdict = "unchecked";
        if (check && !(debug & 1)) {
            CK(hipMalloc(&ref, (size_t)M * N * 4));
            if (epi == 1 || epi == 7) CK(hipMalloc(&ref_pre, (size_t)M * N * 4));
            ref_kernel<<<dim3((unsigned)((N + 255) / 256), (unsigned)M), 256, 0, st>>>(A[0], Bw, bias, rowop, epi, M, N, K, ref, ref_pre);
            unsigned long long* nbad;
            float* maxerr;
            CK(hipMalloc(&nbad, 8));
            CK(hipMalloc(&maxerr, 4));
            char buf[256];
            verdict.clear();
            for (int s = 0; s < NSET; s += NSET - 1) {          // first and last set
                for (int which = 0; which < ((epi == 1 || epi == 7) ? 2 : 1); ++which) {
                    CK(hipMemsetAsync(nbad, 0, 8, st));
                    CK(hipMemsetAsync(maxerr, 0, 4, st));
                    // bf16 output rounding (2^-9 relative) + accumulation-order noise
                    if (epi == 9) cmp_f32_kernel<<<1024, 256, 0, st>>>(reinterpret_cast<const float*>(C[s]), ref, (size_t)M * N, 2e-3f, 2e-4f, nbad, maxerr);
                    else cmp_kernel<<<1024, 256, 0, st>>>(which ? P[s] : C[s], which ? ref_pre : ref, (size_t)M * N, 2e-2f, 8e-3f, nbad, maxerr);
                    unsigned long long hb;
                    float hm;
                    CK(hipMemcpyAsync(&hb, nbad, 8, hipMemcpyDeviceToHost, st));
                    CK(hipMemcpyAsync(&hm, maxerr, 4, hipMemcpyDeviceToHost, st));
                    CK(hipStreamSynchronize(st));
                    snprintf(buf, sizeof(buf), "%s[set%d%s bad=%llu maxerr=%.3g]", verdict.empty() ? "" : " ", s, which ? " preact" : "", hb, hm);
                    verdict += buf;
                    if (hb) verdict += " MISMATCH";
                }
            }
            CK(hipFree(ref));
            if (ref_pre) CK(hipFree(ref_pre));
            CK(hipFree(nbad));
            CK(hipFree(maxerr));
        }
        for (int i = 0; i < 3; ++i) run(i % NSET);
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) run(i % NSET);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = 1e3 * ms / iters, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
        printf("%-34s %8.1f us  %7.1f TF/s  %s\n", c.c_str(), us, tf, verdict.c_str());
        fflush(stdout);
        if (power_secs > 0) {
            // a sustained loop (the DVFS loop settles within a few hundred ms; the first 0.3 s are not sampled)
            const int n_warm = (int)(0.3e6 / us) + 1, n_loop = (int)(power_secs * 1e6 / us) + 1;
            for (int i = 0; i < n_warm; ++i) run(i % NSET);
            CK(hipStreamSynchronize(st));
            PowerProbe pp;
            pp.start();
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < n_loop; ++i) run(i % NSET);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            pp.finish();
            float ms2 = 0.f;
            CK(hipEventElapsedTime(&ms2, e0, e1));
            const double us2 = 1e3 * ms2 / n_loop, flop = 2.0 * M * N * K;
            if (pp.n_w) {
                const double w = pp.sum_w / pp.n_w;
                printf("  power: %6.1f us/launch over %.1f s  %7.1f TF/s  mean %6.1f W (max %6.1f, %d samples)  %.3f pJ/flop", us2, ms2 * 1e-3,
                       flop / (us2 * 1e-6) / 1e12, w, pp.max_w, pp.n_w, w * us2 * 1e-6 / flop * 1e12);
            } else {
                printf("  power: %6.1f us/launch over %.1f s  %7.1f TF/s  (no hwmon power file readable)", us2, ms2 * 1e-3, flop / (us2 * 1e-6) / 1e12);
            }
            if (pp.n_f) printf("  hwmon sclk %.0f MHz", pp.sum_mhz / pp.n_f);
            printf("\n");
            fflush(stdout);
        }
        if (debug & 8) {
            // time stamps of the resident kernel (debug bit 8): [256 workgroups][2 wave rows][16 items][4] shader clocks
            const size_t n = 256 * 2 * 16 * 8;
            unsigned long long* tb;
            CK(hipMalloc(&tb, n * 8));
            CK(hipMemsetAsync(tb, 0, n * 8, st));
            me_dev_set_trace(tb);
            run(0);
            CK(hipStreamSynchronize(st));
            me_dev_set_trace(nullptr);
            std::vector<unsigned long long> h(n);
            CK(hipMemcpy(h.data(), tb, n * 8, hipMemcpyDeviceToHost));
            CK(hipFree(tb));
            for (int wrow = 0; wrow < 2; ++wrow) {
                printf("  wave row %d: mean clocks [K-tile 1 | K-tile 2 | rest of K-loop | realign | epilogue | to next item]  (min..max item total)\n", wrow);
                for (int it = 0; it < 16; ++it) {
                    double d[6] = {0, 0, 0, 0, 0, 0}; int cnt = 0, gc = 0; double tmin = 1e30, tmax = 0;
                    for (int w = 0; w < 256; ++w) {
                        const unsigned long long* t = &h[(((size_t)w * 2 + wrow) * 16 + it) * 8];
                        if (!t[0] || !t[5]) continue;
                        for (int k = 0; k < 5; ++k) d[k] += (double)(t[k + 1] - t[k]);
                        ++cnt;
                        const double tot = (double)(t[5] - t[0]);
                        tmin = tot < tmin ? tot : tmin; tmax = tot > tmax ? tot : tmax;
                        if (it + 1 < 16 && t[8]) { d[5] += (double)(t[8] - t[5]); ++gc; }
                    }
                    if (!cnt) break;
                    printf("    item %2d (%3d wgs): %6.0f | %6.0f | %7.0f | %6.0f | %6.0f | %5.0f   (%.0f..%.0f)\n", it, cnt, d[0] / cnt, d[1] / cnt, d[2] / cnt,
                           d[3] / cnt, d[4] / cnt, gc ? d[5] / gc : 0.0, tmin, tmax);
                }
            }
            for (int wrow = 0; wrow < 2; ++wrow) {
                double d[6] = {0, 0, 0, 0, 0, 0}; int cnt = 0;
                for (int w = 0; w < 8; ++w)
                    for (int it = 1; it < 5; ++it) {
                        const unsigned long long* t = &h[(((size_t)w * 2 + wrow) * 16 + it) * 8];
                        if (!t[0] || !t[5]) continue;
                        for (int k = 0; k < 5; ++k) d[k] += (double)(t[k + 1] - t[k]);
                        ++cnt;
                    }
                if (cnt) printf("  workgroups 0..7, items 1..4, wave row %d: %6.0f | %6.0f | %7.0f | %6.0f | %6.0f\n", wrow, d[0] / cnt, d[1] / cnt, d[2] / cnt, d[3] / cnt, d[4] / cnt);
            }
            {   // per-workgroup busy time (first stamp to last stamp): how unevenly do the CUs finish?
                double mn = 1e30, mx = 0, sum = 0; int cnt = 0;
                double xs[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int xc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int w = 0; w < 256; ++w) {
                    const unsigned long long* t = &h[((size_t)w * 2) * 16 * 8];
                    if (!t[0]) continue;
                    unsigned long long last = 0;
                    for (int it = 0; it < 16; ++it) if (t[it * 8 + 5]) last = t[it * 8 + 5];
                    const double d = (double)(last - t[0]);
                    mn = d < mn ? d : mn; mx = d > mx ? d : mx; sum += d; ++cnt;
                    xs[w & 7] += d; ++xc[w & 7];
                }
                {   // effective shader clock: s_memtime ticks per s_memrealtime tick (constant 100 MHz) over whole items
                    double st_ = 0, rt_ = 0;
                    for (int w = 0; w < 256; ++w)
                        for (int it = 0; it < 16; ++it) {
                            const unsigned long long* q = &h[(((size_t)w * 2) * 16 + it) * 8];
                            if (!q[0] || !q[5] || !q[6] || !q[7]) continue;
                            st_ += (double)(q[5] - q[0]); rt_ += (double)(q[7] - q[6]);
                        }
                    if (rt_ > 0) printf("  shader clock under this kernel: %.3f GHz (s_memtime ticks / s_memrealtime ticks x 100 MHz, summed over items)\n", st_ / rt_ * 0.1);
                }
                printf("  per-workgroup busy clocks: min %.0f  mean %.0f  max %.0f   per XCD mean:", mn, sum / cnt, mx);
                for (int x = 0; x < 8; ++x) printf(" %.0f", xc[x] ? xs[x] / xc[x] : 0.0);
                printf("\n");
            }
            // spread of the start stamps of item 1 across workgroups (how far the CUs drift apart)
            {
                unsigned long long lo = ~0ull, hi = 0;
                for (int w = 0; w < 256; ++w) { const unsigned long long t = h[(((size_t)w * 2) * 16 + 1) * 8]; if (t) { lo = t < lo ? t : lo; hi = t > hi ? t : hi; } }
                printf("  start of item 1 across workgroups: spread %llu clocks\n", hi - lo);
            }
        }
        for (int s = 0; s < NSET; ++s) {
            CK(hipFree(A[s])); CK(hipFree(C[s]));
            if (P[s]) CK(hipFree(P[s]));
        }
        CK(hipFree(Bw)); CK(hipFree(bias));
        if (rowop) CK(hipFree(rowop));
        if (rowop32) CK(hipFree(rowop32));
        if (rstats) CK(hipFree(rstats));
        if (ws) CK(hipFree(ws));
    }
    return 0;
}
