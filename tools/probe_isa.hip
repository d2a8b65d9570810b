// probe_isa.hip -- one-off hardware semantics probes used while designing the kernels (not part of the library).
//   hipcc --offload-arch=gfx950 -O2 tools/probe_isa.hip -o /tmp/probe_isa && /tmp/probe_isa
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// (1) ds_read_b64_tr_b16: LDS holds u16 value == its element index; lane l passes byte address addr[l].
__global__ void tr_probe(const int* addr, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t a = (uint32_t)(uintptr_t)lds + addr[threadIdx.x];
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = r[0] & 0xffff;
    out[threadIdx.x * 4 + 1] = r[0] >> 16;
    out[threadIdx.x * 4 + 2] = r[1] & 0xffff;
    out[threadIdx.x * 4 + 3] = r[1] >> 16;
}

// (2) global_load_lds dwordx4: every lane gives its own global pointer; where do the 16 bytes land in LDS?
__global__ void glds_probe(const uint32_t* src, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    // lane l reads 16 B from src + 4*perm(l) with perm(l) = (l * 5) % 64 (a non-identity permutation)
    const uint32_t* g = src + 4 * ((threadIdx.x * 5) % 64);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}

int main() {
    // ---- tr probe, two address patterns
    int* d_addr; uint16_t* d_out;
    hipMalloc(&d_addr, 64 * sizeof(int)); hipMalloc(&d_out, 256 * sizeof(uint16_t));
    for (int pat = 0; pat < 2; ++pat) {
        std::vector<int> addr(64);
        for (int l = 0; l < 64; ++l) addr[l] = pat == 0 ? l * 8 : ((l & 15) * 64 + (l >> 4) * 8);   // bytes
        hipMemcpy(d_addr, addr.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        std::vector<uint16_t> out(256);
        hipMemcpy(out.data(), d_out, 256 * sizeof(uint16_t), hipMemcpyDeviceToHost);
        printf("TR pattern %d (lane: addr_elems -> 4 values)\n", pat);
        for (int l = 0; l < 64; ++l)
            printf("  l%02d a=%4d -> %4d %4d %4d %4d\n", l, addr[l] / 2, out[l * 4], out[l * 4 + 1], out[l * 4 + 2], out[l * 4 + 3]);
    }
    // ---- glds probe
    uint32_t *d_src, *d_o2;
    hipMalloc(&d_src, 256 * sizeof(uint32_t)); hipMalloc(&d_o2, 512 * sizeof(uint32_t));
    std::vector<uint32_t> src(256);
    for (int i = 0; i < 256; ++i) src[i] = i;
    hipMemcpy(d_src, src.data(), 256 * sizeof(uint32_t), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(glds_probe, dim3(1), dim3(64), 0, 0, d_src, d_o2);
    std::vector<uint32_t> o2(512);
    hipMemcpy(o2.data(), d_o2, 512 * sizeof(uint32_t), hipMemcpyDeviceToHost);
    printf("GLDS: lds dword index -> value (src dword index); expect lds[4*l + j] = 4*perm(l) + j if dest is lane-linear\n");
    int ok = 1;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j)
            if (o2[4 * l + j] != (uint32_t)(4 * ((l * 5) % 64) + j)) ok = 0;
    printf("  lane-linear dest: %s ; first 16 dwords:", ok ? "YES" : "NO");
    for (int i = 0; i < 16; ++i) printf(" %u", o2[i]);
    printf("\n  dwords 256..259: %x %x %x %x\n", o2[256], o2[257], o2[258], o2[259]);
    hipError_t e = hipDeviceSynchronize();
    printf("status: %s\n", hipGetErrorString(e));
    return 0;
}
