// Developer tool: phase timeline of the small-sequence attention kernels (s_memtime stamps per workgroup).
//   hipcc --offload-arch=gfx950 -O3 -o tools/attn_trace tools/attn_trace.hip metatransformer_amd/csrc/api.hip
#define ME_ATTN_TRACE 1
#include "../metatransformer_amd/csrc/attention.hip"
#include <vector>
#include <algorithm>

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 197, B = argc > 2 ? atoi(argv[2]) : 256, H = argc > 3 ? atoi(argv[3]) : 12, hd = 64, C = H * hd;
    const size_t rows = (size_t)B * N;
    std::vector<uint16_t> hq(rows * 3 * C), hdo(rows * C);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; float f = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; union { float f; uint32_t u; } cv; cv.f = f; return (uint16_t)(cv.u >> 16); };
    for (auto& v : hq) v = rnd();
    for (auto& v : hdo) v = rnd();
    void *qkv, *out, *dout, *dqkv; float *lse, *delta;
    hipMalloc(&qkv, hq.size() * 2); hipMalloc(&out, hdo.size() * 2); hipMalloc(&dout, hdo.size() * 2); hipMalloc(&dqkv, hq.size() * 2);
    hipMalloc(&lse, (size_t)B * H * N * 4); hipMalloc(&delta, (size_t)B * H * N * 4);
    hipMemcpy(qkv, hq.data(), hq.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dout, hdo.data(), hdo.size() * 2, hipMemcpyHostToDevice);
    const int nslots = 4096;
    std::vector<long long> tr((size_t)nslots * 128);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pass = 0; pass < 2; ++pass) {
        float ms = 0.f;
        for (int rep = 0; rep < 6; ++rep) {
            if (rep == 1) hipEventRecord(e0, nullptr);
            int rc = pass == 0 ? me_attention_fwd(qkv, 3 * C, out, C, lse, B, N, H, hd, 0.125f, ME_BF16, 0.f, 0, nullptr)
                               : me_attention_bwd(qkv, 3 * C, out, C, dout, C, lse, delta, dqkv, 3 * C, B, N, H, hd, 0.125f, ME_BF16, 0.f, 0, nullptr);
            if (rc) { printf("error: %s\n", me_last_error()); return 1; }
        }
        hipEventRecord(e1, nullptr); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1);
        hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_trace), tr.size() * 8);
        const int nst = 8;
        const int items = std::min(B * H, nslots);
        printf("%s: %.1f us per launch (stamps compiled in); per-wave timeline, mean shader clocks since the item's first stamp of wave 0, over %d items\n",
               pass == 0 ? "fwd" : "bwd", ms * 1000.f / 5, items);
        for (int w = 0; w < 16; ++w) {
            printf("  wave %d:", w);
            for (int k = 0; k < nst; ++k) {
                double sum = 0; int cnt = 0;
                for (int i = 0; i < items; ++i) { long long v = tr[((size_t)i * 16 + w) * 8 + k], b = tr[(size_t)i * 128]; if (v) { sum += (double)(v - b); ++cnt; } }
                if (cnt) printf(" %9.0f", sum / cnt); else printf("         -");
            }
            printf("\n");
        }
        std::fill(tr.begin(), tr.end(), 0);
        hipMemcpyToSymbol(HIP_SYMBOL(g_trace), tr.data(), tr.size() * 8);
    }
    return 0;
}
