// gemm_stage.h -- operand staging shared by the LDS-DMA GEMM families (gemm2b.hip): K-step size, the
// global->LDS DMA wrapper, the swizzled NT / TN stagers and the matching fragment reads.
#pragma once
#include "gemm_common.h"

namespace {

constexpr int KS2 = 32;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ void glds16(const void* gsrc, lds_char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gbl_void*)gsrc, (lds_void*)lds_wave_base, 16, 0, 0);
}

// BM = 128: 4 waves (1 x 4), 3 stages, two workgroups per CU ("g2b").
// BM = 256: 8 waves (2 x 4), 4 stages (DMA three steps ahead), one workgroup per CU with the minimum L2 traffic per
//           flop a CU can have (256 x 256 accumulators = half the register file) ("g2w").
// BM = 64:  4 waves (1 x 4) with half the rows each (two 32-row m-subtiles instead of four): the small-M form -- M = 3 072 rows
//           (the reference's B = 32 x 96-token batches) are 288 tiles at N = 768 instead of 144, without a K split.
template <int BM, int BN> struct G2 {
    static constexpr int NW = BM == 64 ? 4 : BM / 32; // waves: 4 / 4 / 8
    static constexpr int MI = BM == 64 ? 2 : 4;       // 32-row m-subtiles per wave
    static constexpr int NTH = NW * 64;
    static constexpr int NSTAGE = BM == 128 ? 3 : 4;
    static constexpr int A_BYTES = BM * KS2 * 2;      // 8 / 16 KiB
    static constexpr int B_BYTES = BN * KS2 * 2;      // 16 / 8 KiB
    static constexpr int STAGE = A_BYTES + B_BYTES;
    static constexpr int NI = BN / 128;
    static constexpr int NDMA = (A_BYTES + B_BYTES) / 1024 / NW;   // DMA instructions per wave per step
};

// ---- NT: tile [ROWS][32 k] = 64-byte rows; one DMA instruction = 16 rows; lane -> (row 16q + lane/4, slot lane%4)
template <int ROWS, int NW>
struct NtStager2 {
    static constexpr int NINS = ROWS / 16 / NW;       // DMA instructions per wave
    const bf16_t* src[NINS];
    __device__ __forceinline__ void init(const bf16_t* S, int64_t ld, int64_t nrows, int64_t r0, int wave, int lane) {
#pragma unroll
        for (int j = 0; j < NINS; ++j) {
            const int q = wave * NINS + j;
            const int row = q * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ ((row >> 2) & 3);
            int64_t gr = r0 + row;
            gr = gr < nrows ? gr : nrows - 1;
            src[j] = S + gr * ld + chunk * 8;
        }
    }
    __device__ __forceinline__ void issue(lds_char* tile, int wave, int64_t k0) const {
#pragma unroll
        for (int j = 0; j < NINS; ++j) glds16(src[j] + k0, tile + (wave * NINS + j) * 1024);
    }
};
__device__ __forceinline__ bf16x8 nt_frag2(const char* tile, int row, int chunk) {
    return *reinterpret_cast<const bf16x8*>(tile + row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4));
}

// ---- TN: tile [32 t][COLS]; CPR 16-byte chunks per row; one DMA instruction = 64 chunk slots
template <int COLS, int NW>
struct TnStager2 {
    static constexpr int CPR = COLS / 8;
    static constexpr int NINS = 32 * CPR / 64 / NW;   // DMA instructions per wave
    const bf16_t* src[NINS];
    __device__ __forceinline__ void init(const bf16_t* S, int64_t ld, int64_t ncols, int64_t c0, int wave, int lane) {
#pragma unroll
        for (int j = 0; j < NINS; ++j) {
            const int s = (wave * NINS + j) * 64 + lane;
            const int t = s / CPR, slot = s % CPR;
            const int chunk = slot ^ (4 * (t & 3));
            int64_t col = c0 + chunk * 8;
            col = col <= ncols - 8 ? col : ncols - 8;
            src[j] = S + (int64_t)t * ld + col;
        }
    }
    __device__ __forceinline__ void issue(lds_char* tile, int wave, int64_t t0, int64_t ld) const {
#pragma unroll
        for (int j = 0; j < NINS; ++j) glds16(src[j] + t0 * ld, tile + (wave * NINS + j) * 1024);
    }
};
// TN fragment gather; kk in {0, 1}.  The MFMA wants, per lane, 8 reduction-consecutive values of ONE operand column; the tile
// holds reduction rows.  ds_read_b64_tr_b16 hands lane (g, p) of a 16-lane group column p & 3 .. of a 4 x 16 block
// transposed, so two reads (r = 0, 1: reduction rows 4r .. 4r+3 of the lane's 8) build the fragment.
template <int COLS>
__device__ __forceinline__ bf16x8 tn_frag2(const lds_char* tile, int cb, int kk, int lane) {
    const int g = lane >> 4, p = lane & 15;
    const int col = cb + 16 * (g & 1) + 4 * (p & 3);
    const int chunk = col >> 3;
    union { bf16x4 q[2]; bf16x8 v; } u;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int t = 16 * kk + 8 * (g >> 1) + 4 * r + (p >> 2);
        const int off = t * (COLS * 2) + ((chunk ^ (4 * (t & 3))) << 4) + ((col & 7) << 1);
        u.q[r] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + off));
    }
    return u.v;
}

}  // namespace
