// Developer tool: phase timeline of the small-sequence attention kernels (s_memtime stamps per workgroup).
//   hipcc --offload-arch=gfx950 -O3 -o tools/attn_trace tools/attn_trace.hip metatransformer_amd/csrc/api.hip
#define ME_ATTN_TRACE 1
#include "../metatransformer_amd/csrc/attention.hip"
#include <vector>
#include <algorithm>

int main(int argc, char** argv) {
    const int B = 256, N = argc > 1 ? atoi(argv[1]) : 197, H = 12, hd = 64, C = H * hd;
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
    std::vector<long long> tr(nslots * 16);
    for (int pass = 0; pass < 2; ++pass) {
        for (int rep = 0; rep < 3; ++rep) {
            int rc = pass == 0 ? me_attention_fwd(qkv, 3 * C, out, C, lse, B, N, H, hd, 0.125f, ME_BF16, 0.f, 0, nullptr)
                               : me_attention_bwd(qkv, 3 * C, out, C, dout, C, lse, delta, dqkv, 3 * C, B, N, H, hd, 0.125f, ME_BF16, 0.f, 0, nullptr);
            if (rc) { printf("error: %s\n", me_last_error()); return 1; }
            hipDeviceSynchronize();
        }
        hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_trace), tr.size() * 8);
        const int nst = pass == 0 ? 6 : 7;
        const int items = std::min(B * H, nslots);
        long long t0 = tr[0];
        for (int i = 0; i < items; ++i) t0 = std::min(t0, tr[i * 16]);
        printf("%s: per-workgroup/item phase durations (cycles of the 100 MHz-class s_memtime counter), mean over %d items\n", pass == 0 ? "fwd" : "bwd", items);
        for (int k = 1; k < nst; ++k) {
            double sum = 0; for (int i = 0; i < items; ++i) sum += (double)(tr[i * 16 + k] - tr[i * 16 + k - 1]);
            printf("  stamp %d -> %d : %10.1f\n", k - 1, k, sum / items);
        }
        double tot = 0; for (int i = 0; i < items; ++i) tot += (double)(tr[i * 16 + nst - 1] - tr[i * 16]);
        long long tend = 0; for (int i = 0; i < items; ++i) tend = std::max(tend, tr[i * 16 + nst - 1]);
        printf("  total per item %10.1f ; kernel span %lld\n", tot / items, tend - t0);
    }
    return 0;
}
