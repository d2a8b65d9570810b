// vmem_probe -- what does ONE CU's vector-memory path carry, in bytes per shader clock, with NO matrix work beside it?
// (dev tool; VERDICT r5 weak #6: DESIGN's "the epilogue cannot be hidden" argument rests on a ~32 B/clk/CU figure for LOADS that
// had only been inferred from a K-loop that also issues MFMAs, and a store probe.)  One 512-thread workgroup per CU, the shape of
// the g3 GEMM (8 waves), each wave streaming 16 bytes per lane and instruction (1 KiB per wave-instruction):
//   mode 0  buffer_load ... lds  (LDS-DMA, 16 B / lane: what g3_issue does)           -> LDS, never read
//   mode 1  buffer_load_dwordx4 to VGPRs (what the epilogue's row-operand loads do)   -> consumed by an xor chain
//   mode 2  buffer_store_dwordx4 (the epilogue's stores), whole 128-byte lines per 8 lanes
// Footprint per CU (bytes walked round-robin by its 8 waves): 64 KiB x 32 CUs of an XCD = 2 MiB  -> L2-hot after the first pass;
// 1 MiB x 32 = 32 MiB per XCD, 256 MiB in all -> past the L2s (Infinity Cache / HBM).  `depth` = wave-instructions a wave keeps in flight
// (s_waitcnt vmcnt(depth - 1) after every issue).  Clocks: s_memtime (shader clock) around the loop of wave 0 of every CU, averaged;
// s_memrealtime (100 MHz) beside it gives the clock the chip held.
//   hipcc -O2 --offload-arch=gfx950 tools/vmem_probe.hip -o tools/_build/vmem_probe && tools/_build/vmem_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void3;

template <int MODE, int DEPTH>
__global__ __launch_bounds__(1024) void probe_kernel(char* base, size_t per_cu, int iters, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    // blockIdx b runs on XCD b % 8: give the CUs of one XCD neighbouring regions
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    char* mine = base + ((size_t)xcd * (gridDim.x >> 3) + slot) * per_cu;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(mine, 0, (int)per_cu, 0x00020000);
    const unsigned chunks = (unsigned)(per_cu >> 10);            // 1 KiB wave-instructions in the region
    unsigned c = wave;
    char* dst = smem + wave * 2048;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        const int off = (int)((c % chunks) << 10) + lane * 16;
        if (MODE == 0) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void3*)(dst + (i & 1) * 1024), 16, off, 0, 0, 0);
        } else {
            const u32x4 v = {(unsigned)i, (unsigned)lane, 3u, 4u};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
        c += nw;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
}

// mode 1: an explicit ring of DEPTH registers -- the value consumed is the OLDEST one, behind a counted wait (consuming the newest,
// hipcc would wait for it with vmcnt(0))
template <int DEPTH>
__global__ __launch_bounds__(1024) void probe_vgpr_kernel(char* base, size_t per_cu, int iters, unsigned long long* out) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    char* mine = base + ((size_t)xcd * (gridDim.x >> 3) + slot) * per_cu;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(mine, 0, (int)per_cu, 0x00020000);
    const unsigned chunks = (unsigned)(per_cu >> 10);
    unsigned c = wave;
    u32x4 ring[DEPTH];
    u32x4 acc = {0u, 0u, 0u, 0u};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        const int off = (int)((c % chunks) << 10) + lane * 16;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ring[d]) : "v"(off), "s"(rs) : "memory");
        c += nw;
    }
    for (int i = 0; i < iters; i += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
            asm volatile("" : "+v"(ring[d]));
            acc ^= ring[d];
            const int off = (int)((c % chunks) << 10) + lane * 16;
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ring[d]) : "v"(off), "s"(rs) : "memory");
            c += nw;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc ^= ring[d];
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[4096] = 1;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE, int DEPTH>
static void run(const char* name, char* buf, size_t per_cu, int waves, int G, unsigned long long* dout) {
    const int iters = 4096;                                     // wave-instructions per wave (4 MiB per wave)
    std::vector<unsigned long long> h(2 * G);
    double best_clk = 1e30, best_real = 0, best_ms = 0;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        if (MODE == 1) probe_vgpr_kernel<DEPTH><<<G, waves * 64>>>(buf, per_cu, iters, dout);
        else probe_kernel<MODE, DEPTH><<<G, waves * 64, 65536>>>(buf, per_cu, iters, dout);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h.data(), dout, sizeof(unsigned long long) * 2 * G, hipMemcpyDeviceToHost);
        double clk = 0, real = 0;
        for (int b = 0; b < G; ++b) { clk += (double)h[2 * b]; real += (double)h[2 * b + 1]; }
        clk /= G; real /= G;
        if (rep > 0 && clk < best_clk) { best_clk = clk; best_real = real; best_ms = ms; }
    }
    const double bytes_cu = (double)iters * waves * 1024.0;
    printf("%-34s G=%3d waves=%2d depth=%2d footprint/CU=%5zu KiB : %6.1f B/clk/CU  (%.2f GHz held, %6.1f us, %5.2f TB/s chip)\n", name, G, waves, DEPTH,
           per_cu >> 10, bytes_cu / best_clk, best_clk / (best_real * 10.0) , best_ms * 1e3, bytes_cu * G / (best_ms * 1e-3) / 1e12);
}

int main() {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int G = cus;
    char* buf; unsigned long long* dout;
    const size_t total = (size_t)G << 20;                       // 1 MiB per CU
    (void)hipMalloc(&buf, total);
    (void)hipMemset(buf, 1, total);
    (void)hipMalloc(&dout, sizeof(unsigned long long) * 8192);
    (void)hipFuncSetAttribute((const void*)probe_kernel<0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)probe_kernel<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)probe_kernel<0, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)probe_kernel<2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)probe_kernel<2, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    printf("vmem_probe: one workgroup per CU (%d CUs), 16 B per lane and instruction, no MFMA, no LDS reads\n", G);
    for (size_t per_cu : {(size_t)65536, (size_t)1 << 20}) {
        const char* where = per_cu == 65536 ? "L2-hot" : "past L2";
        printf("-- footprint %s\n", where);
        for (int waves : {8, 16}) {
            run<0, 4>("LDS-DMA  buffer_load..lds b128", buf, per_cu, waves, G, dout);
            run<0, 8>("LDS-DMA  buffer_load..lds b128", buf, per_cu, waves, G, dout);
            run<0, 16>("LDS-DMA  buffer_load..lds b128", buf, per_cu, waves, G, dout);
            run<1, 4>("VGPR     buffer_load_dwordx4", buf, per_cu, waves, G, dout);
            run<1, 8>("VGPR     buffer_load_dwordx4", buf, per_cu, waves, G, dout);
            run<2, 8>("store    buffer_store_dwordx4", buf, per_cu, waves, G, dout);
            run<2, 16>("store    buffer_store_dwordx4", buf, per_cu, waves, G, dout);
        }
    }
    // one CU alone (no contention for the XCD's L2 / fabric): is the limit per CU or shared?
    printf("-- one workgroup per XCD, 8 in all (L2-hot): is the limit the CU's own path or shared?\n");
    run<0, 8>("LDS-DMA  buffer_load..lds b128", buf, 65536, 8, 8, dout);
    run<1, 8>("VGPR     buffer_load_dwordx4", buf, 65536, 8, 8, dout);
    run<2, 8>("store    buffer_store_dwordx4", buf, 65536, 8, 8, dout);
    return 0;
}
