// store_probe -- how fast can ONE workgroup per CU store a 256 x 256 bf16 tile, by access pattern?  (dev tool: the g3 GEMM
// epilogue is store-bound per CU; this isolates the pattern dependence.)  Each of G = 256 workgroups (512 threads) writes
// T tiles of a [M, N] bf16 matrix the way a GEMM epilogue would (tile t of workgroup b = tile index b + t * G, row-major
// tiles over N / 256 columns); per wave 16 store instructions of 16 bytes per lane per tile.
//   pattern 0: 16 rows x 64 B per instruction  (lane l: row l & 15, 16-byte group l >> 4; two instructions cover a 128-B line)
//   pattern 1:  8 rows x 128 B per instruction (lane l: row l >> 3, group l & 7)
//   pattern 2:  4 rows x 256 B
//   pattern 3:  2 rows x 512 B                 (a wave owns full 512-B rows of the tile)
//   pattern 4: 16 rows x 64 B, nontemporal
//   pattern 5:  8 rows x 128 B, nontemporal
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(512) void store_kernel(char* C, int N, int tiles_n, int T, int ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const u32x4 v = {(unsigned)lane, (unsigned)wave, 3u, 4u};
    for (int t = 0; t < T; ++t) {
        const int tile = (blockIdx.x + t * gridDim.x) % ntiles;
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        char* base = C + ((size_t)tm * 256 * N + (size_t)tn * 256) * 2;
        // the wave's 128 x 64 sub-tile (128 B per row), 16 instructions
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int row, colb;
            if (PAT == 0 || PAT == 4) { row = (i >> 1) * 16 + (lane & 15); colb = (i & 1) * 64 + (lane >> 4) * 16; }
            else if (PAT == 1 || PAT == 5) { row = i * 8 + (lane >> 3); colb = (lane & 7) * 16; }
            else if (PAT == 2) { row = i * 8 + (lane >> 4) + 4 * 0; colb = (lane & 15) * 16; }     // 256-B rows: the wave pair (wc, wc^1)'s columns
            else { row = i * 8 + (lane >> 5); colb = (lane & 31) * 16; }
            size_t off;
            if (PAT <= 1 || PAT >= 4) off = (size_t)(wr * 128 + row) * N * 2 + wc * 128 + colb;
            else if (PAT == 2) off = (size_t)(wr * 128 + (wc & 1) * 64 + (row >> 1) + 0) * N * 2 + (wc >> 1) * 256 + colb;   // 4 rows / instr, 64 rows per wave
            else off = (size_t)(wave * 32 + (row >> 2)) * N * 2 + colb;                                                       // 2 rows / instr, 32 rows per wave
            u32x4* p = reinterpret_cast<u32x4*>(base + off);
            if (PAT >= 4) __builtin_nontemporal_store(v, p); else *p = v;
        }
    }
}

int main(int argc, char** argv) {
    const int M = 50432, N = 2304, T = 7;
    char* C;
    (void)hipMalloc(&C, (size_t)M * 3072 * 2 * 2);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int tiles_n = N / 256, ntiles = (M / 256) * tiles_n;
    for (int G : {256, 128, 64, 32, 8}) {
        for (int pat = 0; pat < 2; ++pat) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                (void)hipEventRecord(e0);
                if (pat == 0) store_kernel<0><<<G, 512>>>(C, N, tiles_n, T, ntiles);
                else store_kernel<1><<<G, 512>>>(C, N, tiles_n, T, ntiles);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double bytes = (double)T * G * 256 * 256 * 2;
            printf("G=%3d pattern %d: %7.1f us  %6.2f us/tile/CU  %6.2f TB/s\n", G, pat, best * 1e3, best * 1e3 / T, bytes / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
