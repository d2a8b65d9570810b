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
#include "gemm_stage.h"

namespace {

// K-step schedule (measured best of three, tools/gemm_bench.py): read the substep-0 fragments, issue the DMA while they
// fly, prefetch the substep-1 fragments, then all 16 MFMAs.  (s_setprio around the MFMAs measured -3 %.)
//
// EPI selects a specialised epilogue so that the common cases are small straight-line code:
//   0 alpha*acc + bias (* colscale)            (QKV forward, dgrad, wgrad / split-K slabs)
//   1 ... -> (store pre-activation) -> GELU     (fc1 forward)
//   2 ... + residual                            (proj / fc2 forward)          } one row operand, fetched one pass ahead
//   3 ... * gelu'(aux)                          (fc2 dgrad -> dH)             }
//   4 generic: anything include/metaenc.h allows (beta, row remap, pos-embed modulo, combinations)
//   5 split-K slab: raw fp32 partial sums (folded by splitk_reduce_kernel, which applies the real epilogue)
//   6 ... * aux (a saved factor: fc2 dgrad of the training MLP)     7 ... -> save gelu'(h) -> GELU (fc1 forward in training)
constexpr int SCHED = 1;
template <int BM, int BN, bool TN, int EPI>
__global__ __launch_bounds__((BM == 64 ? 256 : BM * 2)) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_g2_kernel(const GemmParams p) {
    typedef G2<BM, BN> G;
    constexpr int NSTAGE = G::NSTAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    lds_char* lds = (lds_char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;                        // wave row (BM = 256 only) / column group
    const int l31 = lane & 31, h = lane >> 5;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tm = wgid / p.tiles_n, tn = wgid % p.tiles_n;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
    const int nk_total = (int)(p.K / KS2);
    const int ks_begin = blockIdx.y * p.ksteps_per_split;
    int ks_end = ks_begin + p.ksteps_per_split;
    ks_end = ks_end < nk_total ? ks_end : nk_total;
    const int nk = ks_end - ks_begin;

    constexpr int MI = G::MI;
    f32x16 acc[MI][G::NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < G::NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    NtStager2<BM, G::NW> nta;
    NtStager2<BN, G::NW> ntb;
    TnStager2<BM, G::NW> tna;
    TnStager2<BN, G::NW> tnb;
    if (TN) {
        tna.init(A, p.lda, p.M, m0, wave, lane);
        tnb.init(B, p.ldb, p.N, n0, wave, lane);
    } else {
        nta.init(A, p.lda, p.M, m0, wave, lane);
        ntb.init(B, p.ldb, p.N, n0, wave, lane);
    }
    auto issue = [&](int stage, int kstep) {
        lds_char* sa = lds + stage * G::STAGE;
        lds_char* sb = sa + G::A_BYTES;
        const int64_t k0 = (int64_t)kstep * KS2;
        if (TN) {
            tna.issue(sa, wave, k0, p.lda);
            tnb.issue(sb, wave, k0, p.ldb);
        } else {
            nta.issue(sa, wave, k0);
            ntb.issue(sb, wave, k0);
        }
    };
    // counted wait: leave `ahead` younger steps (NDMA instructions each) in flight
    auto wait_ahead = [&](int ahead) {
        static_assert(G::NDMA == 6 || G::NDMA == 4 || G::NDMA == 3, "vmcnt immediates below");
        if (ahead <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (ahead == 1) {
            if (G::NDMA == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (G::NDMA == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else {
            if (G::NDMA == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (G::NDMA == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        }
    };

    constexpr int LOOK = NSTAGE - 1;          // steps in flight ahead of the one being computed: 2 / 3
#pragma unroll
    for (int i = 0; i < LOOK; ++i)
        if (i < nk) issue(i, ks_begin + i);

    struct Frags { bf16x8 xb[2][MI]; bf16x8 wa[2][G::NI]; };
    // my part of step t has landed (younger steps' DMA may stay in flight); then everybody's part has, and every reader
    // of step t-1's stage is done with it
    auto step_sync = [&](int t) {
        int ahead = nk - 1 - t;
        ahead = ahead < LOOK - 1 ? ahead : LOOK - 1;
        wait_ahead(ahead);
        __builtin_amdgcn_s_barrier();
    };
    auto load_sub = [&](int stage, int kk, Frags& f) {
        const char* sa = smem + stage * G::STAGE;
        const char* sb = sa + G::A_BYTES;
        if (TN) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) f.xb[kk][mi] = tn_frag2<BM>((const lds_char*)sa, wr * 128 + mi * 32, kk, lane);
#pragma unroll
            for (int ni = 0; ni < G::NI; ++ni) f.wa[kk][ni] = tn_frag2<BN>((const lds_char*)sb, wc * (BN / 4) + ni * 32, kk, lane);
        } else {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) f.xb[kk][mi] = nt_frag2(sa, wr * 128 + mi * 32 + l31, 2 * kk + h);
#pragma unroll
            for (int ni = 0; ni < G::NI; ++ni) f.wa[kk][ni] = nt_frag2(sb, wc * (BN / 4) + ni * 32 + l31, 2 * kk + h);
        }
    };
    // wgrad bias gradient: column sums of A (= dY) over this workgroup's reduction range, on the matrix pipe from the
    // fragments already in registers: D = ones x A-fragment puts sum_k A[k, m] in every row of column m.  The four waves
    // take one 32-row m-subtile each, and the N-tiles of the same (M-tile, split) -- which all stage the same A rows --
    // share the steps round-robin, so the extra work is 2/tiles_n MFMAs per wave and step, evenly spread.
    const bool do_cs = TN && BM == 128 && p.colsum_ws != nullptr;
    f32x16 cs;
#pragma unroll
    for (int e = 0; e < 16; ++e) cs[e] = 0.0f;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
    int cs_turn = do_cs ? (tn - ks_begin % p.tiles_n + p.tiles_n) % p.tiles_n : -1;      // steps until my next turn
    auto mma_step = [&](const Frags& f) {
        if constexpr (TN && BM == 128) if (do_cs) {
            if (cs_turn == 0) {
                // wc is wave-uniform: a scalar switch keeps the fragment index static (a runtime index would demote
                // the fragment array to scratch)
#define ME_CS_CASE(W)                                                                                          \
    case W:                                                                                                    \
        cs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, f.xb[0][W], cs, 0, 0, 0);                           \
        cs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, f.xb[1][W], cs, 0, 0, 0);                           \
        break;
                switch (wc) { ME_CS_CASE(0) ME_CS_CASE(1) ME_CS_CASE(2) default: ME_CS_CASE(3) }
#undef ME_CS_CASE
                cs_turn = p.tiles_n;
            }
            --cs_turn;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < G::NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.wa[kk][ni], f.xb[kk][mi], acc[mi][ni], 0, 0, 0);
    };
    // fragments of substep 0, then the DMA issue while they fly, then substep 1 (measured best order)
    auto load_and_issue = [&](int t, int stage, int stage_ahead, Frags& f) {
        load_sub(stage, 0, f);
        __builtin_amdgcn_sched_barrier(0);
        if (t + LOOK < nk) issue(stage_ahead, ks_begin + t + LOOK);
        __builtin_amdgcn_sched_barrier(0);
        load_sub(stage, 1, f);
    };
    auto next_stage = [&](int s) { return s == NSTAGE - 1 ? 0 : s + 1; };

    // The 8-wave workgroup (BM = 256) runs its two wave rows STAGGERED by the MFMA phase: each SIMD hosts one wave of
    // row 0 and one of row 1; after barrier t row 0 reads its fragments / issues DMA while row 1 still executes the
    // MFMAs of step t-1 on fragments it kept in registers, then they swap -- LDS/DMA phases of one wave sit under the
    // MFMA phase of its SIMD partner instead of all eight waves doing the same thing at the same time.
    const bool trailing = (BM == 256) && wr == 1;
    // static priority for the later-dispatched wave row (it loses VALU arbitration to the older row otherwise; measured
    // +2..5 %; per-MFMA-group s_setprio flips measured neutral) -- wave-uniform condition, scalar branch
    if (trailing) __builtin_amdgcn_s_setprio(1);
    int s_cur = 0, s_ahead = LOOK;
    if (!trailing) {
        for (int t = 0; t < nk; ++t) {
            step_sync(t);
            Frags f;
            load_and_issue(t, s_cur, s_ahead, f);
            mma_step(f);
            s_cur = next_stage(s_cur);
            s_ahead = next_stage(s_ahead);
        }
    } else if (nk > 0) {
        // one fragment set is enough: once the 16 MFMAs of step t-1 have ISSUED their operands are read, and the
        // reads of step t may overwrite the registers while the matrix pipe is still busy
        Frags f;
        step_sync(0);
        load_and_issue(0, s_cur, s_ahead, f);
        for (int t = 1; t < nk; ++t) {
            s_cur = next_stage(s_cur);
            s_ahead = next_stage(s_ahead);
            step_sync(t);
            mma_step(f);
            load_and_issue(t, s_cur, s_ahead, f);
        }
        mma_step(f);
    }

    if (do_cs && h == 0) {      // every accumulator row holds the same sums: register 0 of the lower half-wave
        const int64_t m = m0 + wc * 32 + l31;
        if (m < p.M) p.colsum_ws[((int64_t)blockIdx.y * p.tiles_n + tn) * p.M + m] = cs[0];
    }
    // ---- epilogue: accumulators -> per-wave LDS patch -> row-contiguous 16-byte accesses
#ifdef ME_DEV
    if (p.debug & 1) {                                    // dev: K-loop only
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < G::NI; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) keep += acc[i][j][e];
        if (keep == 1.2345e-30f) reinterpret_cast<float*>(p.C)[0] = keep;
        return;
    }
#endif
    // ---- epilogue.  A lane owns one output ROW and scattered 4-column quads (MFMA layout); stored directly that is
    // 16-byte fragments of 32 different lines per instruction.  Each wave therefore transposes 32-row slabs of its
    // accumulators through its own LDS patch (pitch padded by 16 B: conflict-free 16-byte accesses) and re-reads them
    // row-contiguous, 8 columns per lane -> full 128-byte lines per row, 16-byte coalesced loads/stores.
    //
    // The LDS accesses are INLINE ASM on purpose: vmcnt retires in order and also counts stores, and because LDS-DMA
    // is "a pending LDS write on the vm counter" hipcc guards every compiler-visible LDS access that follows a DMA with
    // s_waitcnt vmcnt(0) -- which here would drain the previous slab's global stores four times per tile (measured:
    // the epilogue then costs 18 us per tile, more than the K-loop at K = 768).  Hidden from the compiler, the slabs'
    // stores stay in flight; a wave's DS operations execute in order, so write -> read needs no wait in between.
    constexpr int WCOLS = BN / 4;
    constexpr int PITCH = WCOLS * 4 + 16;
    constexpr int LPR = WCOLS / 8;                        // lanes per row in the read phase: 8 / 4
    constexpr int RPI = 64 / LPR;                         // rows per read instruction: 8 / 16
    constexpr int NIT = 32 / RPI;                         // read iterations per 32-row slab: 4 / 2
    __builtin_amdgcn_s_barrier();                         // every wave is done reading the operand stages
    const uint32_t patch = (uint32_t)(uintptr_t)(lds + wave * (32 * PITCH));
    const uint32_t waddr = patch + l31 * PITCH + 16 * h;              // + (ni*32 + 8g) * 4
    const uint32_t raddr = patch + (lane / LPR) * PITCH + 32 * (lane % LPR);   // + RPI*i*PITCH (+16)
    float* slab = p.split_k > 1 ? reinterpret_cast<float*>(p.C) + (int64_t)blockIdx.y * p.M * p.N : nullptr;
    const int c0 = 8 * (lane % LPR);
    const int64_t n = n0 + wc * WCOLS + c0;
    const bool n_ok = n + 8 <= p.N;                       // N % 8 == 4 tails take the quad path
    const int64_t mrow0 = m0 + wr * 128 + (lane / LPR);   // + mi*32 + RPI*i
    f32x4 bias0 = {0.f, 0.f, 0.f, 0.f}, bias1 = bias0, cs0 = {1.f, 1.f, 1.f, 1.f}, cs1 = cs0;
    if (EPI != 4 && EPI != 5 && n_ok) {
        if (p.bias) { bias0 = *reinterpret_cast<const f32x4*>(p.bias + n); bias1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4); }
        if (p.colscale) { cs0 = *reinterpret_cast<const f32x4*>(p.colscale + n); cs1 = *reinterpret_cast<const f32x4*>(p.colscale + n + 4); }
    }
    // pin the per-column operands in registers NOW (straight-line code): otherwise hipcc waits for them with vmcnt(0)
    // inside every guarded store block, which drains the stores of the previous rows each time
    asm volatile("" ::"v"(bias0), "v"(bias1), "v"(cs0), "v"(cs1));
    // row operand (EPI 2 / 3): bf16 only on the fast path (fp32 row operands take the generic epilogue); loads are
    // unconditional with clamped coordinates so that no load -- hence no wait -- sits inside a branch
    constexpr bool ROWOP = EPI == 2 || EPI == 3 || EPI == 6;       // one bf16 row operand, fetched one pass ahead
    const uint16_t* rop = reinterpret_cast<const uint16_t*>(EPI == 2 ? p.residual : p.aux);
    const int64_t rop_ld = EPI == 2 ? p.ldres : p.ldaux;
    const int64_t n_cl = n_ok ? n : 0;
    struct RowOp { u32x4 raw[NIT]; };
    auto fetch = [&](int mi, RowOp& ro) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            int64_t m = mrow0 + mi * 32 + RPI * i;
            m = m < p.M ? m : p.M - 1;
            ro.raw[i] = *reinterpret_cast<const u32x4*>(rop + m * rop_ld + n_cl);
        }
    };
    auto unpack = [](const u32x4& r, f32x4& a, f32x4& b) {
        a[0] = __uint_as_float(r[0] << 16); a[1] = __uint_as_float(r[0] & 0xffff0000u);
        a[2] = __uint_as_float(r[1] << 16); a[3] = __uint_as_float(r[1] & 0xffff0000u);
        b[0] = __uint_as_float(r[2] << 16); b[1] = __uint_as_float(r[2] & 0xffff0000u);
        b[2] = __uint_as_float(r[3] << 16); b[3] = __uint_as_float(r[3] & 0xffff0000u);
    };
    RowOp cur, nxt;
    if (ROWOP) fetch(0, cur);
    auto slab_pass = [&](const int mi, const f32x16 (&am)[G::NI]) {
#pragma unroll
        for (int ni = 0; ni < G::NI; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {am[ni][4 * g], am[ni][4 * g + 1], am[ni][4 * g + 2], am[ni][4 * g + 3]};
                asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(waddr), "v"(v), "i"((ni * 32 + 8 * g) * 4) : "memory");
            }
        // the next slab's row operand goes out BEFORE this slab's stores (in-order vmcnt: see above)
        if (ROWOP && mi + 1 < MI) fetch(mi + 1, nxt);
        f32x4 r0[NIT], r1[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r0[i]) : "v"(raddr), "i"(RPI * i * PITCH) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r1[i]) : "v"(raddr), "i"(RPI * i * PITCH + 16) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            f32x4 v0 = r0[i], v1 = r1[i];
            const int64_t m = mrow0 + mi * 32 + RPI * i;
            const bool ok = m < p.M && n_ok;              // (N % 8 == 0 is a precondition of this family)
            if (EPI == 5) {                               // split-K slab: raw fp32 partial sums
                if (ok) {
                    *reinterpret_cast<f32x4*>(slab + m * p.N + n) = v0;
                    *reinterpret_cast<f32x4*>(slab + m * p.N + n + 4) = v1;
                }
                continue;
            }
            if (EPI == 4) {
                if (ok) epilogue_oct(p, m, n, v0, v1);
                continue;
            }
            // the arithmetic is unconditional (out-of-range rows compute garbage that is never stored): only the
            // store itself is guarded, so no wait lands inside a branch
            v0 = v0 * p.alpha + bias0;
            v1 = v1 * p.alpha + bias1;
            if (EPI == 1) {
                if (p.preact && ok) store8_from_f32(p.preact, p.preact_dtype, m * p.ldpre + n, v0, v1);
                v0 = gelu_for4(v0, p.c_dtype);
                v1 = gelu_for4(v1, p.c_dtype);
            }
            if (EPI == 7) {                               // the saved tensor is gelu'(h); the erf pair, as everywhere this flag is served
                f32x4 d0, d1;
                gelu_erf_pair4(v0, v0, d0);
                gelu_erf_pair4(v1, v1, d1);
                if (ok) store8_from_f32(p.preact, p.preact_dtype, m * p.ldpre + n, d0, d1);
            }
            f32x4 ra, rb;
            if (ROWOP) unpack(cur.raw[i], ra, rb);
            if (EPI == 3) {
                v0 *= gelu_grad_for4(ra, p.c_dtype);
                v1 *= gelu_grad_for4(rb, p.c_dtype);
            }
            if (EPI == 6) { v0 *= ra; v1 *= rb; }
            v0 *= cs0; v1 *= cs1;
            if (EPI == 2) { v0 += ra; v1 += rb; }
            if (ok) store8_from_f32(p.C, p.c_dtype, m * p.ldc + n, v0, v1);
        }
        if (ROWOP && mi + 1 < MI) cur = nxt;
    };
    slab_pass(0, acc[0]);
    slab_pass(1, acc[1]);
    if constexpr (MI > 2) {
        slab_pass(2, acc[2]);
        slab_pass(3, acc[3]);
    }
}

template <int BM, int BN, bool TN, int EPI>
int launch2e(const GemmParams& p, hipStream_t stream) {
    typedef G2<BM, BN> G;
    const size_t lds = G::NSTAGE * G::STAGE;
    static OncePerDevice once;
    if (once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_g2_kernel<BM, BN, TN, EPI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)(p.split_k > 1 ? p.split_k : 1));
    hipLaunchKernelGGL((gemm_g2_kernel<BM, BN, TN, EPI>), grid, dim3(G::NTH), lds, stream, p);
    ME_CHECK_LAUNCH("me_gemm(g2)");
    return ME_OK;
}

template <int BM, int BN, bool TN>
int launch2(const GemmParams& p, hipStream_t stream) {
    const int epi = pick_epi_ex(p);
    if (TN) {
        if (epi == 5) return launch2e<BM, BN, TN, 5>(p, stream);
        return epi == 0 ? launch2e<BM, BN, TN, 0>(p, stream) : launch2e<BM, BN, TN, 4>(p, stream);
    }
    switch (epi) {
        case 0: return launch2e<BM, BN, TN, 0>(p, stream);
        case 1: return launch2e<BM, BN, TN, 1>(p, stream);
        case 2: return launch2e<BM, BN, TN, 2>(p, stream);
        case 3: return launch2e<BM, BN, TN, 3>(p, stream);
        case 5: return launch2e<BM, BN, TN, 5>(p, stream);
        case 6: return launch2e<BM, BN, TN, 6>(p, stream);
        case 7: return launch2e<BM, BN, TN, 7>(p, stream);
        default: return launch2e<BM, BN, TN, 4>(p, stream);
    }
}

}  // namespace

bool g2b_supported(const GemmParams& p, int op) {
    if (p.K % KS2 != 0 || p.N % 8 != 0) return false;
    if (op == ME_GEMM_TN) return p.M % 8 == 0 && p.N % 8 == 0 && p.M >= 8 && p.N >= 8;
    return true;
}

// bm = 128: two 4-wave workgroups per CU; bm = 256: one 8-wave workgroup, 4 stages
int launch_g2b(const GemmParams& p, int op, int bm, int bn, hipStream_t stream) {
    if (bm == 256) {
        if (op == ME_GEMM_TN) return launch2<256, 256, true>(p, stream);
        return launch2<256, 256, false>(p, stream);
    }
    if (bm == 64 && op == ME_GEMM_NT) return launch2<64, 128, false>(p, stream);
    if (op == ME_GEMM_TN) return bn == 256 ? launch2<128, 256, true>(p, stream) : launch2<128, 128, true>(p, stream);
    return bn == 256 ? launch2<128, 256, false>(p, stream) : launch2<128, 128, false>(p, stream);
}
