// gemm256.hip -- the performance GEMM family "g256" (bf16 MFMA, fp32 accumulate) for the encoder's Linear layers.
//
// 256 x BN output tile (BN = 256 or 128), 512 threads = 8 waves as 2 (M) x 4 (N); a wave owns 128 x BN/4 outputs as
// 4 x (BN/128) accumulators of v_mfma_f32_32x32x16_bf16.  K advances 64 per step.  Both operand tiles are copied
// global -> LDS by the LDS-DMA path (global_load_lds_dwordx4: 16 B per lane, 1 KiB per wave instruction, no VGPR
// round trip), two stages, one barrier per K-step: the DMA of step t+1 is issued right after the barrier that
// retires step t-1's readers and flies while step t computes.
//
// LDS-DMA writes are lane-linear (wave base + 16*lane), so bank-conflict avoidance cannot pad or scatter the
// destination; instead the *source* chunk each lane fetches is permuted and the fragment reads apply the same
// involution (cdna guide rule 21: linear dest + swizzled source + swizzled read):
//   NT  (C = A B^T, both K-contiguous): LDS rows are 128 B (64 k); 16-byte chunk c of row r lives at chunk slot
//       c ^ ((r >> 1) & 7); fragments are ds_read_b128 with row = lane -> conflict-free.
//   TN  (C = A^T B, wgrad; reduction index is the global ROW): tiles are staged as they lie in memory, [64 t][cols],
//       512-byte (BN=256) rows, fully coalesced; the transposition happens in the LDS read:
//       ds_read_b64_tr_b16 hands lane (col) four consecutive reduction rows.  Chunk c of row t lives at slot
//       c ^ (4 * (t & 3)), which spreads the four rows a 16-lane group touches over four 32-byte bank groups.
// The MFMA is issued with the weight-side tile as A and the activation-side tile as B, so a lane owns ONE output row
// and quads of 4 consecutive output columns (common.h accumulator layout) -> 8/16-byte epilogue accesses.
//
// Split-K (wgrad: tiny outputs, 50k-long reduction): grid.y slices the K-steps; slices write fp32 slabs and a second
// kernel (gemm.hip) folds them deterministically and applies the epilogue.
#include "gemm_common.h"

namespace {

constexpr int BM = 256;
constexpr int KS = 64;          // K per step (bf16 elements)
constexpr int NTH = 512;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ void glds16(const void* gsrc, __attribute__((address_space(3))) char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gbl_void*)gsrc, (lds_void*)lds_wave_base, 16, 0, 0);
}

template <int BN, bool TN> struct G256 {
    static constexpr int A_BYTES = BM * KS * 2;      // 32 KiB
    static constexpr int B_BYTES = BN * KS * 2;      // 32 / 16 KiB
    static constexpr int STAGE = A_BYTES + B_BYTES;
    static constexpr int NI = BN / 128;              // 32-wide n blocks per wave
    static constexpr int A_INSTR = A_BYTES / 1024 / 8;   // DMA instructions per wave per step (4)
    static constexpr int B_INSTR = B_BYTES / 1024 / 8;   // 4 / 2
};

// ------------------------------------------------------------------------------------------------ NT staging
// tile [ROWS][64 k]: DMA instruction q covers rows 8q..8q+7; lane -> (row 8q + lane/8, slot lane%8).
template <int ROWS>
struct NtStager {
    const bf16_t* src[ROWS / 64];     // per-lane source pointer of each of this wave's instructions (k0 = 0)
    __device__ __forceinline__ void init(const bf16_t* S, int64_t ld, int64_t nrows, int64_t r0, int wave, int lane) {
#pragma unroll
        for (int j = 0; j < ROWS / 64; ++j) {
            const int q = wave * (ROWS / 64) + j;
            const int row = q * 8 + (lane >> 3);
            const int slot = lane & 7;
            const int chunk = slot ^ ((row >> 1) & 7);
            int64_t gr = r0 + row;
            gr = gr < nrows ? gr : nrows - 1;
            src[j] = S + gr * ld + chunk * 8;
        }
    }
    __device__ __forceinline__ void issue(__attribute__((address_space(3))) char* tile, int wave, int64_t k0) const {
#pragma unroll
        for (int j = 0; j < ROWS / 64; ++j) glds16(src[j] + k0, tile + (wave * (ROWS / 64) + j) * 1024);
    }
};
__device__ __forceinline__ bf16x8 nt_frag(const char* tile, int row, int chunk) {
    return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

// ------------------------------------------------------------------------------------------------ TN staging
// tile [64 t][COLS]: a row is COLS*2 bytes = CPR chunks; DMA instruction q covers 64 consecutive chunk slots.
template <int COLS>
struct TnStager {
    static constexpr int CPR = COLS / 8;              // chunks per row: 32 / 16
    static constexpr int NINS = 64 * CPR / 64 / 8;    // instructions per wave per step: 4 / 2
    const bf16_t* src[NINS];
    __device__ __forceinline__ void init(const bf16_t* S, int64_t ld, int64_t ncols, int64_t c0, int wave, int lane) {
#pragma unroll
        for (int j = 0; j < NINS; ++j) {
            const int q = wave * NINS + j;
            const int s = q * 64 + lane;               // chunk slot index in the tile
            const int t = s / CPR, slot = s % CPR;
            const int chunk = slot ^ (4 * (t & 3));
            int64_t col = c0 + chunk * 8;
            col = col <= ncols - 8 ? col : ncols - 8;
            src[j] = S + (int64_t)t * ld + col;
        }
    }
    __device__ __forceinline__ void issue(__attribute__((address_space(3))) char* tile, int wave, int64_t t0, int64_t ld) const {
#pragma unroll
        for (int j = 0; j < NINS; ++j) glds16(src[j] + t0 * ld, tile + (wave * NINS + j) * 1024);
    }
};
// fragment for the 32-column block starting at cb, reduction rows 16*kk .. 16*kk+15 (8 per half):
// lane l: group g = l>>4 -> half h = g>>1, columns cb + 16*(g&1) + (l&15); read r (0,1) covers rows 16kk + 8h + 4r + 0..3.
// Address lane p = l&15 supplies: row +(p>>2), columns +4*(p&3)  (probe_isa: out(lane i, elem j) = mem[addr(lane 4j + i/4)][i%4]).
template <int COLS>
__device__ __forceinline__ bf16x8 tn_frag(__attribute__((address_space(3))) const char* tile, int cb, int kk, int lane) {
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

template <int BN, bool TN>
__global__ __launch_bounds__(NTH) void gemm_g256_kernel(const GemmParams p) {
    typedef G256<BN, TN> G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __attribute__((address_space(3))) char* lds = (__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l31 = lane & 31, h = lane >> 5;

    // XCD-aware bijective tile remap (see gemm.hip)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tm = wgid / p.tiles_n, tn = wgid % p.tiles_n;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
    const int nk_total = (int)(p.K / KS);
    const int ks_begin = blockIdx.y * p.ksteps_per_split;
    int ks_end = ks_begin + p.ksteps_per_split;
    ks_end = ks_end < nk_total ? ks_end : nk_total;
    const int nk = ks_end - ks_begin;

    f32x16 acc[4][G::NI];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < G::NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    NtStager<BM> nta;
    NtStager<BN> ntb;
    TnStager<BM> tna;
    TnStager<BN> tnb;
    if (TN) {
        tna.init(A, p.lda, p.M, m0, wave, lane);
        tnb.init(B, p.ldb, p.N, n0, wave, lane);
    } else {
        nta.init(A, p.lda, p.M, m0, wave, lane);
        ntb.init(B, p.ldb, p.N, n0, wave, lane);
    }
    auto issue = [&](int stage, int kstep) {
        __attribute__((address_space(3))) char* sa = lds + stage * G::STAGE;
        __attribute__((address_space(3))) char* sb = sa + G::A_BYTES;
        const int64_t k0 = (int64_t)kstep * KS;
        if (TN) {
            tna.issue(sa, wave, k0, p.lda);
            tnb.issue(sb, wave, k0, p.ldb);
        } else {
            nta.issue(sa, wave, k0);
            ntb.issue(sb, wave, k0);
        }
    };

    if (nk > 0) issue(0, ks_begin);
    for (int t = 0; t < nk; ++t) {
        const int stage = t & 1;
        // step t landed (this wave's part), then everybody's part + all readers of the other stage are done
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 1 < nk) issue(stage ^ 1, ks_begin + t + 1);
        const char* sa = smem + stage * G::STAGE;
        const char* sb = sa + G::A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 xb[4], wa[G::NI];
            if (TN) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    xb[mi] = tn_frag<BM>((__attribute__((address_space(3))) const char*)sa, wr * 128 + mi * 32, kk, lane);
#pragma unroll
                for (int ni = 0; ni < G::NI; ++ni)
                    wa[ni] = tn_frag<BN>((__attribute__((address_space(3))) const char*)sb, wc * (BN / 4) + ni * 32, kk, lane);
            } else {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) xb[mi] = nt_frag(sa, wr * 128 + mi * 32 + l31, 2 * kk + h);
#pragma unroll
                for (int ni = 0; ni < G::NI; ++ni) wa[ni] = nt_frag(sb, wc * (BN / 4) + ni * 32 + l31, 2 * kk + h);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < G::NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ni], xb[mi], acc[mi][ni], 0, 0, 0);
        }
    }

    // ---- epilogue: lane owns output row m, quads of 4 consecutive columns
    float* slab = p.split_k > 1 ? reinterpret_cast<float*>(p.C) + (int64_t)blockIdx.y * p.M * p.N : nullptr;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int64_t m = m0 + wr * 128 + mi * 32 + l31;
        if (m >= p.M) continue;
#pragma unroll
        for (int ni = 0; ni < G::NI; ++ni) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t n = n0 + wc * (BN / 4) + ni * 32 + 8 * g + 4 * h;
                if (n >= p.N) continue;
                f32x4 v = {acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
                if (slab) *reinterpret_cast<f32x4*>(slab + m * p.N + n) = v;
                else epilogue_quad(p, m, n, v);
            }
        }
    }
}

template <int BN, bool TN>
int launch(const GemmParams& p, hipStream_t stream) {
    typedef G256<BN, TN> G;
    const size_t lds = 2 * G::STAGE;
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_g256_kernel<BN, TN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        once = true;
    }
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)(p.split_k > 1 ? p.split_k : 1));
    hipLaunchKernelGGL((gemm_g256_kernel<BN, TN>), grid, dim3(NTH), lds, stream, p);
    ME_CHECK_LAUNCH("me_gemm(g256)");
    return ME_OK;
}

}  // namespace

bool g256_supported(const GemmParams& p, int op) {
    if (p.K % KS != 0) return false;
    if (op == ME_GEMM_TN) return p.M % 8 == 0 && p.N % 8 == 0 && p.M >= 8 && p.N >= 8;
    return true;
}

int launch_g256(const GemmParams& p, int op, int bn, hipStream_t stream) {
    if (op == ME_GEMM_TN) return bn == 256 ? launch<256, true>(p, stream) : launch<128, true>(p, stream);
    return bn == 256 ? launch<256, false>(p, stream) : launch<128, false>(p, stream);
}
