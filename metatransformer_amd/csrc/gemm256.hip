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
#include <stdlib.h>

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
    __device__ __forceinline__ void issue_one(__attribute__((address_space(3))) char* tile, int wave, int64_t k0, int j) const {
        if (j < ROWS / 64) glds16(src[j] + k0, tile + (wave * (ROWS / 64) + j) * 1024);
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
    __device__ __forceinline__ void issue_one(__attribute__((address_space(3))) char* tile, int wave, int64_t t0, int64_t ld, int j) const {
        if (j < NINS) glds16(src[j] + t0 * ld, tile + (wave * NINS + j) * 1024);
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

// VARIANT bits: 1 = spread the DMA issue over the four k-substeps (instead of a burst after the barrier),
//               2 = s_setprio(1) around each MFMA group, 4 = register double-buffering of the fragments.
template <int BN, bool TN, int VARIANT>
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

    if (p.debug & 4) {                                    // dev: de-synchronise the CUs (break write/read lockstep)
        const int k = (bid >> 3) & 3;
        for (int i = 0; i < k * 64; ++i) __builtin_amdgcn_s_sleep(127);      // ~k * 4 us
    }
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

    auto issue_part = [&](int stage, int kstep, int j) {
        __attribute__((address_space(3))) char* sa = lds + stage * G::STAGE;
        __attribute__((address_space(3))) char* sb = sa + G::A_BYTES;
        const int64_t k0 = (int64_t)kstep * KS;
        if (TN) {
            tna.issue_one(sa, wave, k0, p.lda, j);
            tnb.issue_one(sb, wave, k0, p.ldb, j);
        } else {
            nta.issue_one(sa, wave, k0, j);
            ntb.issue_one(sb, wave, k0, j);
        }
    };
    auto load_frags = [&](const char* sa, const char* sb, int kk, bf16x8 (&xb)[4], bf16x8 (&wa)[G::NI]) {
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
    };
    auto mma = [&](const bf16x8 (&xb)[4], const bf16x8 (&wa)[G::NI]) {
        if (VARIANT & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < G::NI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ni], xb[mi], acc[mi][ni], 0, 0, 0);
        if (VARIANT & 2) __builtin_amdgcn_s_setprio(0);
    };

    if (nk > 0) issue(0, ks_begin);
    for (int t = 0; t < nk; ++t) {
        const int stage = t & 1;
        // step t landed (this wave's part), then everybody's part + all readers of the other stage are done
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool more = t + 1 < nk;
        if (!(VARIANT & 1) && more) issue(stage ^ 1, ks_begin + t + 1);
        const char* sa = smem + stage * G::STAGE;
        const char* sb = sa + G::A_BYTES;
        if (VARIANT & 4) {
            bf16x8 xb0[4], wa0[G::NI], xb1[4], wa1[G::NI];
            load_frags(sa, sb, 0, xb0, wa0);
            if ((VARIANT & 1) && more) issue_part(stage ^ 1, ks_begin + t + 1, 0);
            load_frags(sa, sb, 1, xb1, wa1);
            mma(xb0, wa0);
            if ((VARIANT & 1) && more) issue_part(stage ^ 1, ks_begin + t + 1, 1);
            load_frags(sa, sb, 2, xb0, wa0);
            mma(xb1, wa1);
            if ((VARIANT & 1) && more) issue_part(stage ^ 1, ks_begin + t + 1, 2);
            load_frags(sa, sb, 3, xb1, wa1);
            mma(xb0, wa0);
            if ((VARIANT & 1) && more) issue_part(stage ^ 1, ks_begin + t + 1, 3);
            mma(xb1, wa1);
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                bf16x8 xb[4], wa[G::NI];
                if ((VARIANT & 1) && more) issue_part(stage ^ 1, ks_begin + t + 1, kk);
                load_frags(sa, sb, kk, xb, wa);
                mma(xb, wa);
                if (VARIANT & 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- epilogue.  The accumulator layout gives a lane one output ROW and scattered 4-column quads: stored
    // directly that is 32 partial-line (16 B) segments per wave instruction and the store tail costs more than
    // the K-loop at K = 768.  Instead each wave transposes its accumulators through its own 8.5 KiB LDS patch
    // (32 rows x (BN/4) cols fp32, row pitch padded by 16 B -> conflict-free 16-byte writes), re-reads them
    // row-contiguous (8 columns per lane) and applies the fused epilogue with 16-byte coalesced loads / stores:
    // full 128-byte lines per row for BN = 256.
    if (p.debug & 1) {                                    // dev: measure prologue + K-loop only
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < G::NI; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) keep += acc[i][j][e];
        if (keep == 1.2345e-30f) reinterpret_cast<float*>(p.C)[0] = keep;
        return;
    }
    constexpr int WCOLS = BN / 4;                         // columns per wave: 64 / 32
    constexpr int PITCH = WCOLS * 4 + 16;                 // bytes
    constexpr int LPR = WCOLS / 8;                        // lanes per row in the read phase: 8 / 4
    constexpr int RPI = 64 / LPR;                         // rows per read instruction: 8 / 16
    constexpr int NIT = 32 / RPI;                         // read iterations per 32-row pass: 4 / 2
    __builtin_amdgcn_s_barrier();                         // every wave is done reading the operand stages
    char* patch = smem + wave * (32 * PITCH);
    float* slab = p.split_k > 1 ? reinterpret_cast<float*>(p.C) + (int64_t)blockIdx.y * p.M * p.N : nullptr;

    // This lane's 8 output columns are the same for every row it handles: per-column operands are loaded ONCE.
    const int c0 = 8 * (lane % LPR);
    const int64_t n = n0 + wc * WCOLS + c0;
    const bool n_ok = n + 8 <= p.N;                       // (N % 8 == 4 tails take the quad path below)
    f32x4 bias0 = {0.f, 0.f, 0.f, 0.f}, bias1 = bias0, cs0 = {1.f, 1.f, 1.f, 1.f}, cs1 = cs0;
    if (!slab && n_ok) {
        if (p.bias) { bias0 = *reinterpret_cast<const f32x4*>(p.bias + n); bias1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4); }
        if (p.colscale) { cs0 = *reinterpret_cast<const f32x4*>(p.colscale + n); cs1 = *reinterpret_cast<const f32x4*>(p.colscale + n + 4); }
    }
    // A row-dependent operand (residual | aux | beta*C) is fetched one pass AHEAD of the stores that precede its use:
    // vmcnt retires in order and counts stores, so a load issued after a store cannot be waited for without also
    // draining that store -- the loads of pass i+1 therefore go out before the stores of pass i.  One such operand is
    // pipelined (every call site of the encoder has at most one); combinations take the generic per-row path.
    const int n_rowops = (p.residual ? 1 : 0) + (p.aux ? 1 : 0) + (p.beta != 0.0f ? 1 : 0);
    const bool generic = !slab && n_rowops > 1;
    const bool piped = !slab && n_rowops == 1;
    const void* rop = p.residual ? p.residual : (p.aux ? p.aux : p.C);
    const int rop_dt = p.residual ? p.res_dtype : (p.aux ? p.aux_dtype : p.c_dtype);
    const int64_t rop_ld = p.residual ? p.ldres : (p.aux ? p.ldaux : p.ldc);
    struct RowOp { f32x4 v[NIT][2]; };
    auto row_m = [&](int mi, int i) -> int64_t { return m0 + wr * 128 + mi * 32 + (lane / LPR) + RPI * i; };
    auto out_row = [&](int64_t m) -> int64_t {
        return p.out_group_rows ? (m / p.out_group_rows) * p.out_group_stride + (m % p.out_group_rows) + p.out_row_offset : m;
    };
    auto fetch = [&](int mi, RowOp& ro) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int64_t m = row_m(mi, i);
            if (m < p.M && n_ok) {
                int64_t rr = m;
                if (p.residual) rr = p.res_row_mod ? (m % p.res_row_mod) : m;
                else if (!p.aux) rr = out_row(m);
                load8_as_f32(rop, rop_dt, rr * rop_ld + n, ro.v[i][0], ro.v[i][1]);
            }
        }
    };
    RowOp cur, nxt;
    if (piped) fetch(0, cur);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int ni = 0; ni < G::NI; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
                *reinterpret_cast<f32x4*>(patch + l31 * PITCH + (ni * 32 + 8 * g + 4 * h) * 4) = v;
            }
        if (piped && mi + 1 < 4) fetch(mi + 1, nxt);
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int row = (lane / LPR) + RPI * i;
            f32x4 v0 = *reinterpret_cast<const f32x4*>(patch + row * PITCH + c0 * 4);
            f32x4 v1 = *reinterpret_cast<const f32x4*>(patch + row * PITCH + c0 * 4 + 16);
            const int64_t m = row_m(mi, i);
            if (m >= p.M || n >= p.N) continue;
            if (!n_ok) {                                   // N % 8 == 4 tail: generic quad path
                if (slab) *reinterpret_cast<f32x4*>(slab + m * p.N + n) = v0;
                else epilogue_quad(p, m, n, v0);
                continue;
            }
            if (slab) {
                *reinterpret_cast<f32x4*>(slab + m * p.N + n) = v0;
                *reinterpret_cast<f32x4*>(slab + m * p.N + n + 4) = v1;
                continue;
            }
            if (generic) { epilogue_oct(p, m, n, v0, v1); continue; }
            v0 = v0 * p.alpha + bias0;
            v1 = v1 * p.alpha + bias1;
            if (p.preact) store8_from_f32(p.preact, p.preact_dtype, m * p.ldpre + n, v0, v1);
            if (p.act == ME_ACT_GELU) {
                v0 = gelu_erf4(v0);
                v1 = gelu_erf4(v1);
            }
            if (p.aux) {
                v0 *= gelu_erf_grad4(cur.v[i][0]);
                v1 *= gelu_erf_grad4(cur.v[i][1]);
            }
            v0 *= cs0; v1 *= cs1;
            if (p.residual) { v0 += cur.v[i][0]; v1 += cur.v[i][1]; }
            else if (!p.aux && p.beta != 0.0f) { v0 += p.beta * cur.v[i][0]; v1 += p.beta * cur.v[i][1]; }
            if (p.debug & 2) { if (v0[0] + v1[3] == 1.2345e-30f) store8_from_f32(p.C, p.c_dtype, out_row(m) * p.ldc + n, v0, v1); }
            else store8_from_f32(p.C, p.c_dtype, out_row(m) * p.ldc + n, v0, v1);
        }
        if (piped && mi + 1 < 4) cur = nxt;
    }
}

// VARIANT bits: see the kernel.
template <int BN, bool TN, int VARIANT>
int launch(const GemmParams& p, hipStream_t stream) {
    typedef G256<BN, TN> G;
    const size_t lds = 2 * G::STAGE;
    static OncePerDevice once;
    if (once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_g256_kernel<BN, TN, VARIANT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)(p.split_k > 1 ? p.split_k : 1));
    hipLaunchKernelGGL((gemm_g256_kernel<BN, TN, VARIANT>), grid, dim3(NTH), lds, stream, p);
    ME_CHECK_LAUNCH("me_gemm(g256)");
    return ME_OK;
}

int variant() {     // ME_G256_VARIANT = 0..7 (dev A/B switch); default = the measured best
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ME_G256_VARIANT");
        v = e ? atoi(e) & 7 : 0;
    }
    return v;
}

template <int BN, bool TN>
int launch_v(const GemmParams& p, hipStream_t stream) {
    switch (variant()) {
        case 1: return launch<BN, TN, 1>(p, stream);
        case 2: return launch<BN, TN, 2>(p, stream);
        case 3: return launch<BN, TN, 3>(p, stream);
        case 4: return launch<BN, TN, 4>(p, stream);
        case 5: return launch<BN, TN, 5>(p, stream);
        case 6: return launch<BN, TN, 6>(p, stream);
        case 7: return launch<BN, TN, 7>(p, stream);
        default: return launch<BN, TN, 0>(p, stream);
    }
}

}  // namespace

bool g256_supported(const GemmParams& p, int op) {
    if (p.K % KS != 0) return false;
    if (op == ME_GEMM_TN) return p.M % 8 == 0 && p.N % 8 == 0 && p.M >= 8 && p.N >= 8;
    return true;
}

int launch_g256(const GemmParams& p, int op, int bn, hipStream_t stream) {
    if (op == ME_GEMM_TN) return bn == 256 ? launch_v<256, true>(p, stream) : launch_v<128, true>(p, stream);
    return bn == 256 ? launch_v<256, false>(p, stream) : launch_v<128, false>(p, stream);
}
