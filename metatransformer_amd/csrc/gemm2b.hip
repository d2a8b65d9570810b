// gemm2b.hip -- GEMM family "g2b": TWO co-resident workgroups per CU (bf16 MFMA, fp32 accumulate).
//
// Why: the round-1 PMC profile of the one-workgroup-per-CU family (gemm256.hip) shows waves parked at s_waitcnt /
// s_barrier 55 % of the time and MFMA busy 27 %: with a single 8-wave workgroup per CU every L2 miss, every barrier and
// the whole epilogue (13 us against a 24 us K-loop at K = 768) stall the entire CU, and all CUs run the same phase at
// the same time (read burst, then write burst).  Here a workgroup is 4 waves with a 128 x BN tile (BN = 256 / 128),
// K-step 32, THREE LDS stages (72 / 48 KiB) so two (three) workgroups fit a CU: their phases drift apart, one's
// epilogue, DMA waits and GELU VALU work hide under the other's MFMAs, and the DMA runs two K-steps ahead.
//
//   per workgroup : 4 waves as 1 (M) x 4 (N); wave tile 128 x BN/4 = 4 x (BN/128) accumulators of 32x32x16 bf16 MFMA
//   per K-step    : 8 x NI MFMA + (4 + NI) fragment reads per 16-k substep, 2 substeps; one barrier
//   staging       : LDS-DMA (global_load_lds_dwordx4), lane-linear destination, swizzle on the source address:
//        NT rows are 64 B (32 k) -> 4 chunk slots per row; chunk c of row r sits in slot c ^ ((r >> 2) & 3)
//           (four 64-B rows share a 256-B bank row; the 16 lanes a ds_read_b128 services together then hit 16 slots)
//        TN tiles stay [32 t][cols] as in memory, read with ds_read_b64_tr_b16, slot = chunk ^ (4 * (t & 3))
//   pipeline      : wait(step t landed) -> barrier -> issue DMA(t+2) -> compute(t)      (counted vmcnt, never 0 in-loop)
#include "gemm_common.h"
#include <stdlib.h>

namespace {

constexpr int BM2 = 128;
constexpr int KS2 = 32;
constexpr int NTH2 = 256;
constexpr int NSTAGE = 3;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ void glds16(const void* gsrc, lds_char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gbl_void*)gsrc, (lds_void*)lds_wave_base, 16, 0, 0);
}

template <int BN> struct G2 {
    static constexpr int A_BYTES = BM2 * KS2 * 2;     // 8 KiB
    static constexpr int B_BYTES = BN * KS2 * 2;      // 16 / 8 KiB
    static constexpr int STAGE = A_BYTES + B_BYTES;   // 24 / 16 KiB
    static constexpr int NI = BN / 128;
    static constexpr int NDMA = (A_BYTES + B_BYTES) / 1024 / 4;   // DMA instructions per wave per step: 6 / 4
};

// ---- NT: tile [ROWS][32 k] = 64-byte rows; one DMA instruction = 16 rows; lane -> (row 16q + lane/4, slot lane%4)
template <int ROWS>
struct NtStager2 {
    static constexpr int NINS = ROWS / 16 / 4;        // per wave: 2 (128 rows) / 4 (256 rows)
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
template <int COLS>
struct TnStager2 {
    static constexpr int CPR = COLS / 8;
    static constexpr int NINS = 32 * CPR / 64 / 4;    // per wave: 2 (128 cols) / 4 (256 cols)
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
// same fragment gather as gemm256.hip's tn_frag (see there for the lane algebra); kk in {0, 1}
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

template <int BN, bool TN>
__global__ __launch_bounds__(NTH2) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_g2_kernel(const GemmParams p) {
    typedef G2<BN> G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    lds_char* lds = (lds_char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wc = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave = column group
    const int l31 = lane & 31, h = lane >> 5;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tm = wgid / p.tiles_n, tn = wgid % p.tiles_n;
    const int64_t m0 = (int64_t)tm * BM2, n0 = (int64_t)tn * BN;

    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
    const int nk_total = (int)(p.K / KS2);
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

    NtStager2<BM2> nta;
    NtStager2<BN> ntb;
    TnStager2<BM2> tna;
    TnStager2<BN> tnb;
    if (TN) {
        tna.init(A, p.lda, p.M, m0, wc, lane);
        tnb.init(B, p.ldb, p.N, n0, wc, lane);
    } else {
        nta.init(A, p.lda, p.M, m0, wc, lane);
        ntb.init(B, p.ldb, p.N, n0, wc, lane);
    }
    auto issue = [&](int stage, int kstep) {
        lds_char* sa = lds + stage * G::STAGE;
        lds_char* sb = sa + G::A_BYTES;
        const int64_t k0 = (int64_t)kstep * KS2;
        if (TN) {
            tna.issue(sa, wc, k0, p.lda);
            tnb.issue(sb, wc, k0, p.ldb);
        } else {
            nta.issue(sa, wc, k0);
            ntb.issue(sb, wc, k0);
        }
    };

    if (nk > 0) issue(0, ks_begin);
    if (nk > 1) issue(1, ks_begin + 1);
    int s_cur = 0, s_nxt2 = 2;       // stage of step t, stage of step t+2
    for (int t = 0; t < nk; ++t) {
        // my part of step t has landed (the DMA of step t+1, issued later, may stay in flight) ...
        if (t + 1 < nk) {
            if (G::NDMA == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // ... everybody's part has, and every reader of step t-1's stage is done with it
        __builtin_amdgcn_s_barrier();
        if (t + 2 < nk) issue(s_nxt2, ks_begin + t + 2);
        const char* sa = smem + s_cur * G::STAGE;
        const char* sb = sa + G::A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 xb[4], wa[G::NI];
            if (TN) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) xb[mi] = tn_frag2<BM2>((const lds_char*)sa, mi * 32, kk, lane);
#pragma unroll
                for (int ni = 0; ni < G::NI; ++ni) wa[ni] = tn_frag2<BN>((const lds_char*)sb, wc * (BN / 4) + ni * 32, kk, lane);
            } else {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) xb[mi] = nt_frag2(sa, mi * 32 + l31, 2 * kk + h);
#pragma unroll
                for (int ni = 0; ni < G::NI; ++ni) wa[ni] = nt_frag2(sb, wc * (BN / 4) + ni * 32 + l31, 2 * kk + h);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < G::NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ni], xb[mi], acc[mi][ni], 0, 0, 0);
        }
        s_cur = s_cur == NSTAGE - 1 ? 0 : s_cur + 1;
        s_nxt2 = s_nxt2 == NSTAGE - 1 ? 0 : s_nxt2 + 1;
    }

    // ---- epilogue (same scheme as gemm256.hip): accumulators -> per-wave LDS patch -> row-contiguous 16-byte accesses
    constexpr int WCOLS = BN / 4;
    constexpr int PITCH = WCOLS * 4 + 16;
    constexpr int LPR = WCOLS / 8;
    constexpr int RPI = 64 / LPR;
    constexpr int NIT = 32 / RPI;
    __builtin_amdgcn_s_barrier();
    char* patch = smem + wc * (32 * PITCH);
    float* slab = p.split_k > 1 ? reinterpret_cast<float*>(p.C) + (int64_t)blockIdx.y * p.M * p.N : nullptr;
    const int c0 = 8 * (lane % LPR);
    const int64_t n = n0 + wc * WCOLS + c0;
    const bool n_ok = n + 8 <= p.N;
    f32x4 bias0 = {0.f, 0.f, 0.f, 0.f}, bias1 = bias0, cs0 = {1.f, 1.f, 1.f, 1.f}, cs1 = cs0;
    if (!slab && n_ok) {
        if (p.bias) { bias0 = *reinterpret_cast<const f32x4*>(p.bias + n); bias1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4); }
        if (p.colscale) { cs0 = *reinterpret_cast<const f32x4*>(p.colscale + n); cs1 = *reinterpret_cast<const f32x4*>(p.colscale + n + 4); }
    }
    const int n_rowops = (p.residual ? 1 : 0) + (p.aux ? 1 : 0) + (p.beta != 0.0f ? 1 : 0);
    const bool generic = !slab && (n_rowops > 1 || p.out_group_rows != 0 || p.res_row_mod != 0);
    const bool piped = !slab && !generic && n_rowops == 1;
    const void* rop = p.residual ? p.residual : (p.aux ? p.aux : p.C);
    const int rop_dt = p.residual ? p.res_dtype : (p.aux ? p.aux_dtype : p.c_dtype);
    const int64_t rop_ld = p.residual ? p.ldres : (p.aux ? p.ldaux : p.ldc);
    struct RowOp { f32x4 v[NIT][2]; };
    const int64_t mrow0 = m0 + (lane / LPR);                    // + mi*32 + RPI*i
    auto fetch = [&](int mi, RowOp& ro) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int64_t m = mrow0 + mi * 32 + RPI * i;
            if (m < p.M && n_ok) load8_as_f32(rop, rop_dt, m * rop_ld + n, ro.v[i][0], ro.v[i][1]);
        }
    };
    RowOp cur;
    // one 32-row pass; called four times with a compile-time accumulator reference (a runtime-indexed acc[] would be
    // demoted to scratch memory)
    auto pass = [&](const int mi, const f32x16 (&am)[G::NI]) {
        // row operand of this pass: issued before the LDS transposition so its latency hides under it (the in-order
        // vmcnt makes it also wait for the previous pass's stores -- the co-resident workgroup covers that stall)
        if (piped) fetch(mi, cur);
#pragma unroll
        for (int ni = 0; ni < G::NI; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {am[ni][4 * g], am[ni][4 * g + 1], am[ni][4 * g + 2], am[ni][4 * g + 3]};
                *reinterpret_cast<f32x4*>(patch + l31 * PITCH + (ni * 32 + 8 * g + 4 * h) * 4) = v;
            }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int row = (lane / LPR) + RPI * i;
            f32x4 v0 = *reinterpret_cast<const f32x4*>(patch + row * PITCH + c0 * 4);
            f32x4 v1 = *reinterpret_cast<const f32x4*>(patch + row * PITCH + c0 * 4 + 16);
            const int64_t m = mrow0 + mi * 32 + RPI * i;
            if (m >= p.M || n >= p.N) continue;
            if (!n_ok) {
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
#pragma unroll
                for (int e = 0; e < 4; ++e) { v0[e] = gelu_erf(v0[e]); v1[e] = gelu_erf(v1[e]); }
            }
            if (p.aux) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v0[e] *= gelu_erf_grad(cur.v[i][0][e]); v1[e] *= gelu_erf_grad(cur.v[i][1][e]); }
            }
            v0 *= cs0; v1 *= cs1;
            if (p.residual) { v0 += cur.v[i][0]; v1 += cur.v[i][1]; }
            else if (!p.aux && p.beta != 0.0f) { v0 += p.beta * cur.v[i][0]; v1 += p.beta * cur.v[i][1]; }
            store8_from_f32(p.C, p.c_dtype, m * p.ldc + n, v0, v1);
        }
    };
    pass(0, acc[0]);
    pass(1, acc[1]);
    pass(2, acc[2]);
    pass(3, acc[3]);
}

template <int BN, bool TN>
int launch2(const GemmParams& p, hipStream_t stream) {
    typedef G2<BN> G;
    const size_t lds = NSTAGE * G::STAGE;
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_g2_kernel<BN, TN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        once = true;
    }
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)(p.split_k > 1 ? p.split_k : 1));
    hipLaunchKernelGGL((gemm_g2_kernel<BN, TN>), grid, dim3(NTH2), lds, stream, p);
    ME_CHECK_LAUNCH("me_gemm(g2b)");
    return ME_OK;
}

}  // namespace

bool g2b_supported(const GemmParams& p, int op) {
    if (p.K % KS2 != 0) return false;
    if (op == ME_GEMM_TN) return p.M % 8 == 0 && p.N % 8 == 0 && p.M >= 8 && p.N >= 8;
    return true;
}

int launch_g2b(const GemmParams& p, int op, int bn, hipStream_t stream) {
    if (op == ME_GEMM_TN) return bn == 256 ? launch2<256, true>(p, stream) : launch2<128, true>(p, stream);
    return bn == 256 ? launch2<256, false>(p, stream) : launch2<128, false>(p, stream);
}
