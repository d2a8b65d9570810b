// gemm3.hip -- GEMM family "g3": bf16 NT (both operands reduction-contiguous), 256 x 256 tile, K-tile 64, 8 waves,
// ping-pong K-loop in 4 phases per K-tile with the LDS-DMA stream running 7 half-tiles ahead.
//
// Why a third family: the PMC profile of "g2w" (gemm2b.hip, K-step 32, one barrier per step) shows the waves
// issue-stalled half the time with MFMA busy 0.28-0.43.  Three structural changes address that:
//   * K-tile 64: an operand row is one full 128-byte line per tile (K-step 32 fetched every line twice, as two 64-byte
//     halves in different steps), and the barrier / wait overhead per MFMA halves.
//   * four phases per K-tile, 16 MFMAs (v_mfma_f32_16x16x32_bf16) each = one 64 x 32 quadrant of the wave's 128 x 64
//     output x the whole K-tile.  The two wave rows (waves 0-3 / 4-7: one of each per SIMD) run ONE BARRIER out of
//     phase: while one row issues its 16 MFMAs under s_setprio 1, its SIMD partner reads the next phase's fragments
//     from LDS and issues its share of the DMA, then they swap.  The matrix pipe of every SIMD always has a wave in
//     an MFMA-only segment; LDS reads and DMA issue never sit in front of an MFMA of the same wave.
//   * the DMA stream is issued one half-tile (128 rows x 64 k = 16 KiB, two instructions per wave) per phase, seven
//     half-tiles ahead of the phase that reads it, and waited for with ONE counted vmcnt per K-tile (never 0 in the
//     loop): every half-tile has >= 5 phases to land.
//
// LDS: 2 buffers x 4 half-tiles x 16 KiB = 128 KiB.  Half-tiles are ordered by first use:
//   j = 0  B-X  weight rows  wc*64 +  0..31  (all four wave columns)       read in phase 0
//   j = 1  A-X  token rows   wr*128 +  0..63 (both wave rows)              read in phase 0
//   j = 2  B-Y  weight rows  wc*64 + 32..63                                read in phase 1
//   j = 3  A-Y  token rows   wr*128 + 64..127                              read in phase 2
//   phase 0: X x X quadrant   phase 1: A-X x B-Y   phase 2: A-Y x B-Y   phase 3: A-Y x B-X (no reads)
// A half-tile is 128 rows of 128 bytes (64 k); a DMA instruction (1 KiB, lane-linear destination) is 8 rows; chunk c
// (16 bytes = 8 k) of local row r sits in slot c ^ ((r >> 1) & 7) of its row -- two rows share a 256-byte bank row,
// so the 16 lanes a ds_read_b128 services together (MI355X_MICROARCH.md, LDS) hit 16 different 16-byte slots.  The
// permutation is applied on the SOURCE address of the DMA and again on the fragment read.
//
// Hazards (phase index P = 4 t + p counts over the whole K-loop; wave row 1 runs one barrier behind wave row 0):
//   RAW  half-tile i is issued in phase i - 7 and read in phase >= 4 (i/4); every wave waits "all of K-tile t+1 has
//        landed" (vmcnt(6): three younger half-tiles may stay in flight) BEFORE the first barrier of phase 3 of
//        K-tile t, so both wave rows have passed that wait before either reads K-tile t+1.
//   WAR  slot reuse: half-tile i+8 is issued in phase i+1.  A-X, B-Y, A-Y were last read in phase i-1 (two phases and
//        >= 2 barriers earlier for both wave rows).  B-X is read in phase i itself: its four reads are issued first
//        and retired with lgkmcnt(8) before that phase's first barrier, which the issuing wave row passes later.
#include "gemm_common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void3;
typedef const __attribute__((address_space(1))) void gbl_void3;

constexpr int G3_BM = 256, G3_BN = 256, G3_BK = 64;
constexpr int G3_HALF = 128 * 128;              // bytes in a half-tile
constexpr int G3_BUF = 4 * G3_HALF;             // 64 KiB
constexpr int G3_LDS = 2 * G3_BUF;              // 128 KiB

__device__ __forceinline__ void g3_dma16(const char* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gbl_void3*)gsrc, (lds_void3*)lds_wave_base, 16, 0, 0);
}

// Everything the K-loop keeps in registers.  All arrays are indexed with compile-time constants only.
struct G3State {
    f32x4 acc[8][4];            // [m-tile of 16 rows][n-tile of 16 cols] of the wave's 128 x 64 output (transposed MFMA:
                                //  lane l holds row (l & 15), cols 4*(l >> 4) .. +3 of the 16 x 16 tile)
    bf16x8 bx[2][2], by[2][2];  // weight fragments [n-tile][k-sub]
    bf16x8 ax[4][2], ay[4][2];  // token fragments  [m-tile][k-sub]
    uint32_t src[4][2];         // DMA source byte offsets from the tile's A / B row base: [half-tile type][instruction]
    const char* a_base;         // A + m0 * lda (bytes), wave-uniform
    const char* b_base;         // B + n0 * ldb
    char* smem;
    uint32_t ra[2], rb[2];      // fragment read byte offsets inside a half-tile for k-sub 0 / 1 (wave + lane part)
    int wave;
};

template <int J> __device__ __forceinline__ void g3_issue(const G3State& s, int buf, int kt) {
    // half-tile type J of K-tile kt into buffer buf
    const char* base = ((J & 1) ? s.a_base : s.b_base) + (int64_t)kt * (G3_BK * 2);
    char* dst = s.smem + buf * G3_BUF + J * G3_HALF + s.wave * 2048;
    g3_dma16(base + s.src[J][0], dst);
    g3_dma16(base + s.src[J][1], dst + 1024);
}

__device__ __forceinline__ bf16x8 g3_frag(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }

#define G3_MMA(MT, NT, AF, BF)                                                                                   \
    s.acc[MT][NT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[(NT) & 1][0], AF[(MT) & 3][0], s.acc[MT][NT], 0, 0, 0); \
    s.acc[MT][NT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[(NT) & 1][1], AF[(MT) & 3][1], s.acc[MT][NT], 0, 0, 0);

// MODE 0: steady state; 1: second-to-last K-tile (only phase 0 still issues, the wait drains); 2: last K-tile
template <int BUF, int P, int MODE> __device__ __forceinline__ void g3_phase(G3State& s, int kt) {
    const char* buf = s.smem + BUF * G3_BUF;
    if (P == 0) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int k = 0; k < 2; ++k) s.bx[nt][k] = g3_frag(buf + 0 * G3_HALF + nt * 2048 + s.rb[k]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int k = 0; k < 2; ++k) s.ax[mt][k] = g3_frag(buf + 1 * G3_HALF + mt * 2048 + s.ra[k]);
    } else if (P == 1) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int k = 0; k < 2; ++k) s.by[nt][k] = g3_frag(buf + 2 * G3_HALF + nt * 2048 + s.rb[k]);
    } else if (P == 2) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int k = 0; k < 2; ++k) s.ay[mt][k] = g3_frag(buf + 3 * G3_HALF + mt * 2048 + s.ra[k]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 0 || (MODE == 1 && P == 0)) {
        if (P == 0) g3_issue<3>(s, BUF ^ 1, kt + 1);
        if (P == 1) g3_issue<0>(s, BUF, kt + 2);
        if (P == 2) g3_issue<1>(s, BUF, kt + 2);
        if (P == 3) g3_issue<2>(s, BUF, kt + 2);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (P == 0) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    if (P == 3 && MODE == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if (P == 3 && MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if (P == 0) {
        G3_MMA(0, 0, s.ax, s.bx) G3_MMA(0, 1, s.ax, s.bx) G3_MMA(1, 0, s.ax, s.bx) G3_MMA(1, 1, s.ax, s.bx)
        G3_MMA(2, 0, s.ax, s.bx) G3_MMA(2, 1, s.ax, s.bx) G3_MMA(3, 0, s.ax, s.bx) G3_MMA(3, 1, s.ax, s.bx)
    } else if (P == 1) {
        G3_MMA(0, 2, s.ax, s.by) G3_MMA(0, 3, s.ax, s.by) G3_MMA(1, 2, s.ax, s.by) G3_MMA(1, 3, s.ax, s.by)
        G3_MMA(2, 2, s.ax, s.by) G3_MMA(2, 3, s.ax, s.by) G3_MMA(3, 2, s.ax, s.by) G3_MMA(3, 3, s.ax, s.by)
    } else if (P == 2) {
        G3_MMA(4, 2, s.ay, s.by) G3_MMA(4, 3, s.ay, s.by) G3_MMA(5, 2, s.ay, s.by) G3_MMA(5, 3, s.ay, s.by)
        G3_MMA(6, 2, s.ay, s.by) G3_MMA(6, 3, s.ay, s.by) G3_MMA(7, 2, s.ay, s.by) G3_MMA(7, 3, s.ay, s.by)
    } else {
        G3_MMA(4, 0, s.ay, s.bx) G3_MMA(4, 1, s.ay, s.bx) G3_MMA(5, 0, s.ay, s.bx) G3_MMA(5, 1, s.ay, s.bx)
        G3_MMA(6, 0, s.ay, s.bx) G3_MMA(6, 1, s.ay, s.bx) G3_MMA(7, 0, s.ay, s.bx) G3_MMA(7, 1, s.ay, s.bx)
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
}

template <int BUF, int MODE> __device__ __forceinline__ void g3_ktile(G3State& s, int kt) {
    g3_phase<BUF, 0, MODE>(s, kt);
    g3_phase<BUF, 1, MODE>(s, kt);
    g3_phase<BUF, 2, MODE>(s, kt);
    g3_phase<BUF, 3, MODE>(s, kt);
}

// EPI as in gemm2b.hip: 0 bias (* colscale), 1 + GELU (+ pre-activation save), 2 + residual row operand,
// 3 * gelu'(aux row operand), 4 generic (epilogue_oct), 5 raw fp32 slab (split-K partial sums)
template <int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_g3_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tm = wgid / p.tiles_n, tn = wgid % p.tiles_n;
    const int64_t m0 = (int64_t)tm * G3_BM, n0 = (int64_t)tn * G3_BN;

    G3State s;
    s.smem = smem;
    s.wave = wave;
    s.a_base = reinterpret_cast<const char*>(p.A) + m0 * p.lda * 2;
    s.b_base = reinterpret_cast<const char*>(p.B) + n0 * p.ldb * 2;
    {
        // DMA sources: instruction i of this wave covers local rows 16*wave + 8*i + (lane >> 3), slot lane & 7
        const int64_t a_last = p.M - 1 - m0, b_last = p.N - 1 - n0;         // clamp: rows past the edge re-read the last row
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rl = 16 * wave + 8 * i + (lane >> 3);
            const int c = (lane & 7) ^ ((rl >> 1) & 7);
            int64_t ax_row = (rl >> 6) * 128 + (rl & 63), bx_row = (rl >> 5) * 64 + (rl & 31);
            int64_t ay_row = ax_row + 64, by_row = bx_row + 32;
            ax_row = ax_row < a_last ? ax_row : a_last; ay_row = ay_row < a_last ? ay_row : a_last;
            bx_row = bx_row < b_last ? bx_row : b_last; by_row = by_row < b_last ? by_row : b_last;
            s.src[0][i] = (uint32_t)(bx_row * p.ldb * 2 + c * 16);
            s.src[1][i] = (uint32_t)(ax_row * p.lda * 2 + c * 16);
            s.src[2][i] = (uint32_t)(by_row * p.ldb * 2 + c * 16);
            s.src[3][i] = (uint32_t)(ay_row * p.lda * 2 + c * 16);
        }
        // fragment reads: local row = (wave part) + 16 * tile + (lane & 15), chunk = 4 * ksub + (lane >> 4)
        const int l15 = lane & 15;
        const uint32_t lp = (l15 >> 3) * 1024 + (lane & 7) * 128 + ((((lane >> 4) ^ (l15 >> 1)) & 7) << 4);
        s.ra[0] = wr * 8192 + lp; s.ra[1] = s.ra[0] ^ 64;
        s.rb[0] = wc * 4096 + lp; s.rb[1] = s.rb[0] ^ 64;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s.acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkt = (int)(p.K / G3_BK);          // even, >= 2 (g3_supported)
    // prologue: half-tiles 0..6 of the stream (K-tile 0 complete, K-tile 1 without A-Y)
    g3_issue<0>(s, 0, 0); g3_issue<1>(s, 0, 0); g3_issue<2>(s, 0, 0); g3_issue<3>(s, 0, 0);
    g3_issue<0>(s, 1, 1); g3_issue<1>(s, 1, 1); g3_issue<2>(s, 1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();   // wave row 1 runs one barrier behind (wave-uniform scalar branch)

    int kt = 0;
    for (; kt < nkt - 2; kt += 2) {
        g3_ktile<0, 0>(s, kt);
        g3_ktile<1, 0>(s, kt + 1);
    }
    g3_ktile<0, 1>(s, kt);
    g3_ktile<1, 2>(s, kt + 1);
    if (wr == 0) __builtin_amdgcn_s_barrier();   // realign the two wave rows: every operand read is complete

#ifdef ME_DEV
    if (p.debug & 1) {                           // dev: K-loop only
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) keep += s.acc[i][j][0] + s.acc[i][j][1] + s.acc[i][j][2] + s.acc[i][j][3];
        if (keep == 1.2345e-30f) reinterpret_cast<float*>(p.C)[0] = keep;
        return;
    }
#endif

    // ---- epilogue: same scheme as gemm2b.hip.  Each wave transposes 32-row slabs of its accumulators through its own
    // LDS patch (inline-asm DS ops so hipcc adds no vmcnt(0) that would drain the previous slab's stores) and re-reads
    // them row-contiguous, 8 columns per lane: whole 128-byte lines per row, 16-byte coalesced loads / stores.
    constexpr int PITCH = 64 * 4 + 16;
    constexpr int LPR = 8, RPI = 8, NIT = 4;
    const int l15 = lane & 15, g4 = lane >> 4;
    const uint32_t patch = (uint32_t)(uintptr_t)(smem + wave * (32 * PITCH));
    const uint32_t waddr = patch + l15 * PITCH + 16 * g4;                      // + (16*ml) * PITCH + (16*nt) * 4
    const uint32_t raddr = patch + (lane / LPR) * PITCH + 32 * (lane % LPR);    // + RPI*i*PITCH (+16)
    float* slab = p.split_k > 1 ? reinterpret_cast<float*>(p.C) + (int64_t)blockIdx.y * p.M * p.N : nullptr;
    const int64_t n = n0 + wc * 64 + 8 * (lane % LPR);
    const bool n_ok = n + 8 <= p.N;
    const int64_t mrow0 = m0 + wr * 128 + (lane / LPR);
    f32x4 bias0 = {0.f, 0.f, 0.f, 0.f}, bias1 = bias0, cs0 = {1.f, 1.f, 1.f, 1.f}, cs1 = cs0;
    if (EPI != 4 && EPI != 5 && n_ok) {
        if (p.bias) { bias0 = *reinterpret_cast<const f32x4*>(p.bias + n); bias1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4); }
        if (p.colscale) { cs0 = *reinterpret_cast<const f32x4*>(p.colscale + n); cs1 = *reinterpret_cast<const f32x4*>(p.colscale + n + 4); }
    }
    asm volatile("" ::"v"(bias0), "v"(bias1), "v"(cs0), "v"(cs1));
    const uint16_t* rop = reinterpret_cast<const uint16_t*>(EPI == 2 ? p.residual : p.aux);
    const int64_t rop_ld = EPI == 2 ? p.ldres : p.ldaux;
    const int64_t n_cl = n_ok ? n : 0;
    struct RowOp { u32x4 raw[NIT]; };
    auto fetch = [&](int sl, RowOp& ro) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            int64_t m = mrow0 + sl * 32 + RPI * i;
            m = m < p.M ? m : p.M - 1;
            ro.raw[i] = *reinterpret_cast<const u32x4*>(rop + m * rop_ld + n_cl);
        }
    };
    auto unpack = [](const u32x4& rw, f32x4& a, f32x4& b) {
        a[0] = __uint_as_float(rw[0] << 16); a[1] = __uint_as_float(rw[0] & 0xffff0000u);
        a[2] = __uint_as_float(rw[1] << 16); a[3] = __uint_as_float(rw[1] & 0xffff0000u);
        b[0] = __uint_as_float(rw[2] << 16); b[1] = __uint_as_float(rw[2] & 0xffff0000u);
        b[2] = __uint_as_float(rw[3] << 16); b[3] = __uint_as_float(rw[3] & 0xffff0000u);
    };
    RowOp cur, nxt;
    if (EPI == 2 || EPI == 3) fetch(0, cur);
    auto slab_pass = [&](const int sl, const f32x4 (&a0)[4], const f32x4 (&a1)[4]) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(waddr), "v"(a0[nt]), "i"(nt * 64) : "memory");
            asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(waddr), "v"(a1[nt]), "i"(16 * PITCH + nt * 64) : "memory");
        }
        if ((EPI == 2 || EPI == 3) && sl + 1 < 4) fetch(sl + 1, nxt);
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
            const int64_t m = mrow0 + sl * 32 + RPI * i;
            const bool ok = m < p.M && n_ok;
            if (EPI == 5) {
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
            v0 = v0 * p.alpha + bias0;
            v1 = v1 * p.alpha + bias1;
            if (EPI == 1) {
                if (p.preact && ok) store8_from_f32(p.preact, p.preact_dtype, m * p.ldpre + n, v0, v1);
                v0 = gelu_erf4(v0);
                v1 = gelu_erf4(v1);
            }
            f32x4 qa, qb;
            if (EPI == 2 || EPI == 3) unpack(cur.raw[i], qa, qb);
            if (EPI == 3) {
                v0 *= gelu_erf_grad4(qa);
                v1 *= gelu_erf_grad4(qb);
            }
            v0 *= cs0; v1 *= cs1;
            if (EPI == 2) { v0 += qa; v1 += qb; }
            if (ok) store8_from_f32(p.C, p.c_dtype, m * p.ldc + n, v0, v1);
        }
        if ((EPI == 2 || EPI == 3) && sl + 1 < 4) cur = nxt;
    };
    slab_pass(0, s.acc[0], s.acc[1]);
    slab_pass(1, s.acc[2], s.acc[3]);
    slab_pass(2, s.acc[4], s.acc[5]);
    slab_pass(3, s.acc[6], s.acc[7]);
}

template <int EPI> int launch3e(const GemmParams& p, hipStream_t stream) {
    static OncePerDevice once;
    if (once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_g3_kernel<EPI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS);
    }
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n), 1);
    hipLaunchKernelGGL((gemm_g3_kernel<EPI>), grid, dim3(512), G3_LDS, stream, p);
    ME_CHECK_LAUNCH("me_gemm(g3)");
    return ME_OK;
}

}  // namespace

bool g3_supported(const GemmParams& p, int op) {
    if (op != ME_GEMM_NT) return false;
    if (p.K % (2 * G3_BK) != 0 || p.N % 8 != 0) return false;
    // DMA source offsets are 32-bit from the tile's first row
    if (256 * p.lda * 2 >= (1ll << 31) || 256 * p.ldb * 2 >= (1ll << 31)) return false;
    return true;
}

int launch_g3(const GemmParams& p, int epi, hipStream_t stream) {
    switch (epi) {
        case 0: return launch3e<0>(p, stream);
        case 1: return launch3e<1>(p, stream);
        case 2: return launch3e<2>(p, stream);
        case 3: return launch3e<3>(p, stream);
        default: return launch3e<4>(p, stream);
    }
}
