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
#include <algorithm>
#include <map>
#include <mutex>
#include <utility>

namespace {

typedef __attribute__((address_space(3))) void lds_void3;

constexpr int G3_BM = 256, G3_BN = 256, G3_BK = 64;
constexpr int G3_HALF = 128 * 128;              // bytes in a half-tile
constexpr int G3_BUF = 4 * G3_HALF;             // 64 KiB
constexpr int G3_LDS = 2 * G3_BUF;              // 128 KiB
constexpr int G3_SLAB_FLOATS = G3_BM * G3_BN;   // one fp32 partial tile per workgroup (stream-K fix-up)

// Everything the K-loop keeps in registers.  All arrays are indexed with compile-time constants only.
struct G3State {
    f32x4 acc[8][4];            // [m-tile of 16 rows][n-tile of 16 cols] of the wave's 128 x 64 output (transposed MFMA:
                                //  lane l holds row (l & 15), cols 4*(l >> 4) .. +3 of the 16 x 16 tile)
    bf16x8 bx[2][2], by[2][2];  // weight fragments [n-tile][k-sub]
    bf16x8 ax[4][2], ay[4][2];  // token fragments  [m-tile][k-sub]
    uint32_t src[4][2];         // DMA source byte offsets from the tile's first A / B row: [half-tile type][instruction]
    char* smem;
    uint32_t ra[2][2], rb[2][2];// NT: fragment read LDS addresses [buffer][k-sub]: buffer + wave / lane part inside a half-tile
    uint32_t ta[4], tb[2];      // TN: transposing-read byte offsets per m-tile / n-tile of a quadrant (wave + lane part)
    int kstep_a, kstep_b;       // source bytes per K-tile: NT 128 (along the row); TN 64 rows = 128 * ld
    f32x4 binit[4];             // resident NT kernel: what the accumulators of n-tile 0..3 START at (the columns' bias, or zero) --
                                // the C operand of the first MFMAs behind an epilogue (g3_phase<.., SEAM>); dead in between
    float cs[2];                // TN: running column sums of A (the bias gradient) for m-tiles wc and 4 + wc of this wave row
    int wave;
};

// Wave-uniform source of one output tile's operand rows: buffer descriptors over rows [m0, m0+256) of A and [n0, n0+256)
// of B (clipped at the matrix edge: rows past the edge read as zeros through the descriptor's bounds check, so edge
// tiles need no clamping and every lane keeps ONE set of offsets for the whole kernel).
struct G3Src {
    __amdgpu_buffer_rsrc_t a, b;
};
__device__ __forceinline__ G3Src g3_make_src(const GemmParams& p, int tm, int tn) {
    const int64_t m0 = (int64_t)tm * G3_BM, n0 = (int64_t)tn * G3_BN;
    int64_t ra = p.M - m0, rb = p.N - n0;
    ra = ra < G3_BM ? ra : G3_BM;
    rb = rb < G3_BN ? rb : G3_BN;
    G3Src s;
    s.a = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.A)) + m0 * p.lda * 2, 0,
                                            (int)((ra - 1) * p.lda * 2 + p.K * 2), 0x00020000);
    s.b = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.B)) + n0 * p.ldb * 2, 0,
                                            (int)((rb - 1) * p.ldb * 2 + p.K * 2), 0x00020000);
    return s;
}
__device__ __forceinline__ G3Src g3_null_src(const GemmParams& p) {
    G3Src s;
    s.a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, 0, 0x00020000);
    s.b = s.a;
    return s;
}

// half-tile type J of K-tile kt (of the source's own numbering) into buffer buf
template <int J> __device__ __forceinline__ void g3_issue(const G3State& s, const G3Src& src, int buf, int kt) {
    char* dst = s.smem + buf * G3_BUF + J * G3_HALF + s.wave * 2048;
    const __amdgpu_buffer_rsrc_t r = (J & 1) ? src.a : src.b;
    const int koff = kt * ((J & 1) ? s.kstep_a : s.kstep_b);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void3*)dst, 16, (int)s.src[J][0], koff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void3*)(dst + 1024), 16, (int)s.src[J][1], koff, 0, 0);
}

// NT fragment: one 16-byte read at (lane part + buffer) + an IMMEDIATE (half-tile slot, tile).  Inline asm for the same
// reason as the transposing reads below, and so that the address stays "one register + constant": left to itself hipcc
// materialises a separate address register for most of the 24 (slot, tile) combinations of the second buffer (its
// offsets exceed the 16-bit immediate when counted from the start of LDS), which the resident kernel cannot afford.
template <int OFF> __device__ __forceinline__ bf16x8 g3_frag(uint32_t addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF) : "memory");
    return __builtin_bit_cast(bf16x8, v);
}

// TN operand tiles lie in LDS as in memory, [64 k][128 columns] (256-byte rows); the MFMA wants 8 consecutive k of ONE
// column per lane.  ds_read_b64_tr_b16 transposes a 4 (k) x 16 (columns) block per 16-lane group: lane (g = l >> 4,
// p = l & 15) ADDRESSES 4 columns (4 (p & 3)..) of k-row (p >> 2) and RECEIVES column p's four k-values; two reads
// (k-rows 4r + 0..3, r = 0, 1) make the fragment of k-group g.  `base` carries everything lane- and tile-dependent
// (ta / tb); the k-sub (x 32 rows) and r (x 4 rows) parts are immediates.
// The reads are inline asm: with LDS-DMA in flight hipcc guards every compiler-visible transposing read with
// s_waitcnt vmcnt(0) (it cannot prove the intrinsic does not alias the DMA's destination), which would drain the stream
// four times per K-tile.  Their results are consumed only behind the phase's own `s_waitcnt lgkmcnt(0)` + sched_barrier.
template <int OFF> __device__ __forceinline__ bf16x8 g3_frag_tn(uint32_t addr) {
    u32x2 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "i"(OFF) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "i"(OFF + 1024) : "memory");
    const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, v);
}

// sum of a fragment's eight bf16 values (one column of A, eight consecutive k) in fp32
__device__ __forceinline__ float g3_frag_sum(bf16x8 f) {
    // v_dot2_f32_bf16 with a vector of ones: two elements per instruction, fp32 accumulation
    const bf16x2 one = {(bf16_t)1.0f, (bf16_t)1.0f};
    float a = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) a = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{f[2 * e], f[2 * e + 1]}, one, a, false);
    return a;
}
// bias gradient on the side (wgrad): the column sums of A over this workgroup's K-range come from the A fragments the
// wave already holds -- a few VALU additions in the LOAD part of a phase, no extra pass over dY and no extra MFMA.  The
// four wave columns of a wave row hold the same A fragments: wave column wc takes m-tiles wc (A-X) and 4 + wc (A-Y).
// The index is wave-uniform; a scalar if-chain on the state's own arrays keeps every fragment index static (an array
// passed by reference, or indexed at run time, is demoted to scratch).
#define G3_MMA(MT, NT, AF, BF)                                                                                   \
    s.acc[MT][NT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[(NT) & 1][0], AF[(MT) & 3][0], SEAM ? s.binit[NT] : s.acc[MT][NT], 0, 0, 0); \
    s.acc[MT][NT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[(NT) & 1][1], AF[(MT) & 3][1], s.acc[MT][NT], 0, 0, 0);

// One phase of K-tile `BUF`.  s0 / k0: where the NEXT K-tile of the stream comes from (phase 0 issues its A-Y);
// s1 / k1: the K-tile after that (phases 1..3 issue its B-X, A-X, B-Y).  There is ONE code path: past the end of a
// workgroup's stream the source is a null descriptor (zero records: the DMA writes zeros into a buffer nobody reads any
// more and touches no memory), so the issue pattern, and with it the counted wait, never changes -- and the 128
// accumulators never meet a control-flow join inside the K-loop.
// SEAM > 0: the first K-tile after an epilogue of the resident kernel (gemm_g3r_kernel): the A-Y half-tile phase 0 would
// issue went out BEFORE the epilogue, and the counted wait of phase 3 lets the epilogue's SEAM memory operations (which
// sit between that half-tile and this K-tile's own three in the in-order queue) stay in flight.  It is also the first K-tile
// of an output tile: its MFMAs take s.binit as their C operand, so the accumulators need no initialisation pass.
template <int BUF, int P, bool TN = false, int SEAM = 0>
__device__ __forceinline__ void g3_phase(G3State& s, const G3Src& s0, int k0, const G3Src& s1, int k1, bool cs_on = false) {
    if (TN && (P == 1 || P == 3) && cs_on) {      // (wave-uniform) fragments read one / two phases ago, waited for in that phase
        const int wcol = s.wave & 3;
        if (wcol == 0) s.cs[P >> 1] += g3_frag_sum((P == 1 ? s.ax : s.ay)[0][0]) + g3_frag_sum((P == 1 ? s.ax : s.ay)[0][1]);
        else if (wcol == 1) s.cs[P >> 1] += g3_frag_sum((P == 1 ? s.ax : s.ay)[1][0]) + g3_frag_sum((P == 1 ? s.ax : s.ay)[1][1]);
        else if (wcol == 2) s.cs[P >> 1] += g3_frag_sum((P == 1 ? s.ax : s.ay)[2][0]) + g3_frag_sum((P == 1 ? s.ax : s.ay)[2][1]);
        else s.cs[P >> 1] += g3_frag_sum((P == 1 ? s.ax : s.ay)[3][0]) + g3_frag_sum((P == 1 ? s.ax : s.ay)[3][1]);
    }
    // fragment (tile t, k-sub k) of half-tile slot SL: NT one 16-byte read, TN two transposing 8-byte reads
    const uint32_t lbuf = (uint32_t)(uintptr_t)s.smem + BUF * G3_BUF;      // (TN) 32-bit LDS address of this buffer
#define G3_RD_B(SL, t, k) (TN ? g3_frag_tn<(SL) * G3_HALF + (k) * 8192>(lbuf + s.tb[t]) : g3_frag<(SL) * G3_HALF + (t) * 2048>(s.rb[BUF][k]))
#define G3_RD_A(SL, t, k) (TN ? g3_frag_tn<(SL) * G3_HALF + (k) * 8192>(lbuf + s.ta[t]) : g3_frag<(SL) * G3_HALF + (t) * 2048>(s.ra[BUF][k]))
    if (P == 0) {
        s.bx[0][0] = G3_RD_B(0, 0, 0); s.bx[0][1] = G3_RD_B(0, 0, 1); s.bx[1][0] = G3_RD_B(0, 1, 0); s.bx[1][1] = G3_RD_B(0, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        s.ax[0][0] = G3_RD_A(1, 0, 0); s.ax[0][1] = G3_RD_A(1, 0, 1); s.ax[1][0] = G3_RD_A(1, 1, 0); s.ax[1][1] = G3_RD_A(1, 1, 1);
        s.ax[2][0] = G3_RD_A(1, 2, 0); s.ax[2][1] = G3_RD_A(1, 2, 1); s.ax[3][0] = G3_RD_A(1, 3, 0); s.ax[3][1] = G3_RD_A(1, 3, 1);
    } else if (P == 1) {
        s.by[0][0] = G3_RD_B(2, 0, 0); s.by[0][1] = G3_RD_B(2, 0, 1); s.by[1][0] = G3_RD_B(2, 1, 0); s.by[1][1] = G3_RD_B(2, 1, 1);
    } else if (P == 2) {
        s.ay[0][0] = G3_RD_A(3, 0, 0); s.ay[0][1] = G3_RD_A(3, 0, 1); s.ay[1][0] = G3_RD_A(3, 1, 0); s.ay[1][1] = G3_RD_A(3, 1, 1);
        s.ay[2][0] = G3_RD_A(3, 2, 0); s.ay[2][1] = G3_RD_A(3, 2, 1); s.ay[3][0] = G3_RD_A(3, 3, 0); s.ay[3][1] = G3_RD_A(3, 3, 1);
    }
#undef G3_RD_A
#undef G3_RD_B
    __builtin_amdgcn_sched_barrier(0);
    if (P == 0 && SEAM == 0) g3_issue<3>(s, s0, BUF ^ 1, k0);
    if (P == 1) g3_issue<0>(s, s1, BUF, k1);
    if (P == 2) g3_issue<1>(s, s1, BUF, k1);
    if (P == 3) g3_issue<2>(s, s1, BUF, k1);
    __builtin_amdgcn_sched_barrier(0);
    // (the B-X reads are issued first: NT 4 of 12, TN 8 of 24 DS operations -- retire exactly those before the barrier)
    if (P == 0) { if (TN) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); }
    if (P == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + SEAM) : "memory");
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

template <int BUF, bool TN = false, int SEAM = 0>
__device__ __forceinline__ void g3_ktile(G3State& s, const G3Src& s0, int k0, const G3Src& s1, int k1, bool cs_on = false) {
    g3_phase<BUF, 0, TN, SEAM>(s, s0, k0, s1, k1, cs_on);
    g3_phase<BUF, 1, TN, SEAM>(s, s0, k0, s1, k1, cs_on);
    g3_phase<BUF, 2, TN, SEAM>(s, s0, k0, s1, k1, cs_on);
    g3_phase<BUF, 3, TN, SEAM>(s, s0, k0, s1, k1, cs_on);
}

__device__ __forceinline__ void g3_init_lane(G3State& s, const GemmParams& p, char* smem, int wave, int lane) {
    s.smem = smem;
    s.wave = wave;
    const int wr = wave >> 2, wc = wave & 3;
    // DMA sources: instruction i of this wave covers local rows 16*wave + 8*i + (lane >> 3) of a half-tile, slot lane & 7
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rl = 16 * wave + 8 * i + (lane >> 3);
        const int c = (lane & 7) ^ ((rl >> 1) & 7);
        const int ax_row = (rl >> 6) * 128 + (rl & 63), bx_row = (rl >> 5) * 64 + (rl & 31);
        s.src[0][i] = (uint32_t)(bx_row * p.ldb * 2 + c * 16);
        s.src[1][i] = (uint32_t)(ax_row * p.lda * 2 + c * 16);
        s.src[2][i] = (uint32_t)((bx_row + 32) * p.ldb * 2 + c * 16);
        s.src[3][i] = (uint32_t)((ax_row + 64) * p.lda * 2 + c * 16);
    }
    // fragment reads: local row = (wave part) + 16 * tile + (lane & 15), chunk = 4 * ksub + (lane >> 4)
    const int l15 = lane & 15;
    const uint32_t lp = (l15 >> 3) * 1024 + (lane & 7) * 128 + ((((lane >> 4) ^ (l15 >> 1)) & 7) << 4);
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        s.ra[b][0] = lds0 + b * G3_BUF + wr * 8192 + lp; s.ra[b][1] = s.ra[b][0] ^ 64;
        s.rb[b][0] = lds0 + b * G3_BUF + wc * 4096 + lp; s.rb[b][1] = s.rb[b][0] ^ 64;
        // (opaque: eight registers, not two plus arithmetic in front of every read)
        asm volatile("" : "+v"(s.ra[b][0]), "+v"(s.ra[b][1]), "+v"(s.rb[b][0]), "+v"(s.rb[b][1]));
    }
    s.kstep_a = s.kstep_b = G3_BK * 2;
}

// ---- TN (wgrad: C[M, N] = A[K, M]^T B[K, N], reduction index = the ROW of both operands).
// A half-tile is 64 k-rows x 128 columns (256-byte rows): A-X = the columns wave rows 0 / 1 need for their quadrant row 0
// (tile columns 0..63 and 128..191 -> chunks 0..7 / 8..15), A-Y the other 64 + 64; B-X = the four wave columns' first 32
// (tile columns wc*64 + 0..31 -> chunks 4 wc .. 4 wc + 3), B-Y the second 32.  16-byte chunk c of k-row t sits in slot
// c ^ 4 (t & 3) ^ 2 ((t >> 3) & 1): the 32 lanes a transposing read services together touch k-rows (p >> 2) + 8 (g & 1),
// which the permutation spreads over all eight 32-byte sections of the 256-byte bank row.
__device__ __forceinline__ int g3_tn_swz(int t) { return (4 * (t & 3)) ^ (2 * ((t >> 3) & 1)); }
__device__ __forceinline__ void g3_init_lane_tn(G3State& s, const GemmParams& p, char* smem, int wave, int lane) {
    s.smem = smem;
    s.wave = wave;
    const int wr = wave >> 2, wc = wave & 3;
    // DMA sources: instruction i of this wave covers k-rows 8*wave + 4*i + (lane >> 4) of a half-tile, slot lane & 15
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int t = 8 * wave + 4 * i + (lane >> 4);
        const int c = (lane & 15) ^ g3_tn_swz(t);
        const int a_col = (c >> 3) * 128 + (c & 7) * 8, b_col = (c >> 2) * 64 + (c & 3) * 8;
        s.src[0][i] = (uint32_t)(t * p.ldb * 2 + b_col * 2);
        s.src[1][i] = (uint32_t)(t * p.lda * 2 + a_col * 2);
        s.src[2][i] = (uint32_t)(t * p.ldb * 2 + (b_col + 32) * 2);
        s.src[3][i] = (uint32_t)(t * p.lda * 2 + (a_col + 64) * 2);
    }
    // transposing reads: lane (g, pp) addresses k-row 8 g + (pp >> 2) (+ 4 r + 32 ksub as immediates), columns cb + 4 (pp & 3)
    const int g = lane >> 4, pp = lane & 15;
    const int trow = 8 * g + (pp >> 2);
    auto base = [&](int cb) {
        const int col = cb + 4 * (pp & 3);
        return (uint32_t)(trow * 256 + ((((col >> 3) ^ g3_tn_swz(trow)) & 15) << 4) + ((col & 7) << 1));
    };
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) s.ta[mt] = base(wr * 64 + mt * 16);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) s.tb[nt] = base(wc * 32 + nt * 16);
    s.kstep_a = (int)(G3_BK * p.lda * 2);
    s.kstep_b = (int)(G3_BK * p.ldb * 2);
}
// operand columns [m0, ..) of A and [n0, ..) of B, all K rows: rows past K read as zeros (bounds check on the end of the
// matrix); columns past the edge of an edge tile read the next row's data -- they only feed outputs that are never stored
__device__ __forceinline__ G3Src g3_make_src_tn(const GemmParams& p, int tm, int tn) {
    const int64_t m0 = (int64_t)tm * G3_BM, n0 = (int64_t)tn * G3_BN;
    G3Src s;
    s.a = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.A)) + m0 * 2, 0,
                                            (int)(p.K * p.lda * 2 - m0 * 2), 0x00020000);
    s.b = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.B)) + n0 * 2, 0,
                                            (int)(p.K * p.ldb * 2 - n0 * 2), 0x00020000);
    return s;
}
__device__ __forceinline__ void g3_zero(G3State& s) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s.acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// ---- epilogue without LDS.  Two neighbouring 16 x 16 accumulator tiles (n-tiles 2q, 2q+1) are re-dealt inside the wave
// with v_permlane16_swap (rows of 16 lanes: odd rows of the first operand <-> even rows of the second), after which lane
// (r = l & 15, g = l >> 4) holds EIGHT consecutive output columns of row r: n-tile 2q + (g & 1), columns 8 (g >> 1) ..
// +7 -- one 16-byte bf16 store / row-operand load per lane, 64 contiguous bytes per row and instruction.  The operand
// buffers in LDS are not touched, so the DMA stream of the next tile keeps running under the epilogue.
// EPI: 0 bias, 1 + GELU (+ pre-activation save), 2 + residual row operand, 3 * gelu'(aux row operand), 4 generic
// (epilogue_oct: colscale, beta, row remaps, fp32 row operands ...)
// EPI 5: raw fp32 partial sums into a split-K slab (row-major [rows][N], first row = slab_row0)
template <int EPI>
__device__ __forceinline__ void g3_epilogue(const GemmParams& p, G3State& s, int64_t m0, int64_t n0, int lane,
                                            float* slab = nullptr, int64_t slab_row0 = 0) {
    // everything lane-dependent below is derived HERE: an address hoisted out of the persistent loop would sit in
    // registers across the K-loops (which have none to spare) and come back from scratch
    asm volatile("" : "+v"(lane));
    const int wr = s.wave >> 2, wc = s.wave & 3;
    const int r = lane & 15, g = lane >> 4;
    const f32x4 alpha4 = {p.alpha, p.alpha, p.alpha, p.alpha};
    int64_t n[2];
    bool n_ok[2];
    f32x4 bias[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        n[q] = n0 + wc * 64 + (2 * q + (g & 1)) * 16 + 8 * (g >> 1);
        n_ok[q] = n[q] + 8 <= p.N;
        const int64_t nc = n_ok[q] ? n[q] : 0;
        bias[q][0] = bias[q][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (EPI != 4 && EPI != 5 && p.bias) {
            bias[q][0] = *reinterpret_cast<const f32x4*>(p.bias + nc);
            bias[q][1] = *reinterpret_cast<const f32x4*>(p.bias + nc + 4);
        }
    }
    // pin the per-column operands in registers NOW (straight-line code): otherwise hipcc waits for them with vmcnt(0)
    // inside every guarded store block, which drains the stores of the previous rows each time
    asm volatile("" ::"v"(bias[0][0]), "v"(bias[0][1]), "v"(bias[1][0]), "v"(bias[1][1]));
    const uint16_t* rop = reinterpret_cast<const uint16_t*>(EPI == 2 ? p.residual : p.aux);
    const int64_t rop_ld = EPI == 2 ? p.ldres : p.ldaux;
    const int64_t mrow = m0 + wr * 128 + r;
    auto fetch = [&](const int mt, u32x4 (&raw)[2]) {
        int64_t m = mrow + mt * 16;
        m = m < p.M ? m : p.M - 1;                       // unconditional loads with clamped coordinates (no wait in a branch)
#pragma unroll
        for (int q = 0; q < 2; ++q) raw[q] = *reinterpret_cast<const u32x4*>(rop + m * rop_ld + (n_ok[q] ? n[q] : 0));
    };
    auto unpack = [](const u32x4& rw, f32x4& a, f32x4& b) {
        a[0] = __uint_as_float(rw[0] << 16); a[1] = __uint_as_float(rw[0] & 0xffff0000u);
        a[2] = __uint_as_float(rw[1] << 16); a[3] = __uint_as_float(rw[1] & 0xffff0000u);
        b[0] = __uint_as_float(rw[2] << 16); b[1] = __uint_as_float(rw[2] & 0xffff0000u);
        b[2] = __uint_as_float(rw[3] << 16); b[3] = __uint_as_float(rw[3] & 0xffff0000u);
    };
    // row operand (residual / gelu' input): twelve of the tile's sixteen 16-byte loads go out at once, the last four as
    // soon as the first slabs have freed their registers and BEFORE those slabs' stores (vmcnt retires in order) -- one
    // memory latency per tile instead of one per 16-row slab (the K-loop's fragment registers are free here)
    constexpr int AHEAD = 6;
    u32x4 rowop[8][2];
    if (EPI == 2 || EPI == 3) {
#pragma unroll
        for (int mt = 0; mt < AHEAD; ++mt) fetch(mt, rowop[mt]);
    }
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        f32x4 ro[2][2];
        if (EPI == 2 || EPI == 3) {
            unpack(rowop[mt][0], ro[0][0], ro[0][1]);
            unpack(rowop[mt][1], ro[1][0], ro[1][1]);
            if (mt + AHEAD < 8) fetch(mt + AHEAD, rowop[mt + AHEAD]);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int64_t m = mrow + mt * 16;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            f32x4 v0 = s.acc[mt][2 * q], v1 = s.acc[mt][2 * q + 1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v0[e]), __float_as_uint(v1[e]), false, false);
                v0[e] = __uint_as_float(sw[0]);
                v1[e] = __uint_as_float(sw[1]);
            }
            const bool ok = m < p.M && n_ok[q];
            if (EPI == 5) {
                if (ok) {
                    float* d = slab + (m - slab_row0) * p.N + n[q];
                    *reinterpret_cast<f32x4*>(d) = v0;
                    *reinterpret_cast<f32x4*>(d + 4) = v1;
                }
                continue;
            }
            if (EPI == 4) {
                if (ok) epilogue_oct(p, m, n[q], v0, v1);
                continue;
            }
            v0 = v0 * alpha4 + bias[q][0];
            v1 = v1 * alpha4 + bias[q][1];
            if (EPI == 1) {
                if (p.preact && ok) store8_from_f32(p.preact, p.preact_dtype, m * p.ldpre + n[q], v0, v1);
                v0 = gelu_for4(v0, p.c_dtype);
                v1 = gelu_for4(v1, p.c_dtype);
            }
            const f32x4 qa = ro[q][0], qb = ro[q][1];
            if (EPI == 3) {
                v0 *= gelu_grad_for4(qa, p.c_dtype);
                v1 *= gelu_grad_for4(qb, p.c_dtype);
            }
            if (EPI == 2) { v0 += qa; v1 += qb; }
#ifdef ME_DEV
            if (p.debug & 4) {                         // dev: epilogue arithmetic without the stores
                asm volatile("" ::"v"(v0), "v"(v1));
                continue;
            }
#endif
            if (ok) store8_from_f32(p.C, p.c_dtype, m * p.ldc + n[q], v0, v1);
        }
    }
}

#ifdef ME_DEV
// (dev build only: correct and tested with tools/gemm_dev, but not faster than one tile per workgroup yet -- see DESIGN.md)
// ---- persistent kernel: data-parallel rounds + a stream-K remainder.
// Work unit = a PAIR of K-tiles (buffer 0 / buffer 1).  With G workgroups (one per CU) and T tiles of hk pairs each:
//   * R = T / G (rounded down, minus one when the rest would be less than a tile per workgroup) rounds are plain
//     data-parallel: in round j workgroup v owns tile j G + v, so at any time the 32 CUs of an XCD work on 32
//     neighbouring tiles and share their operand panels through the XCD's L2 (v is the XCD-chunked id);
//   * the remaining tiles' pairs are dealt out in contiguous ranges [v q + min(v, r), ...), 1 .. 2 tiles' worth each:
//     every CU gets the same amount of MFMA work whatever T is (N = 768: 591 tiles on 256 CUs used to be 3 rounds for
//     2.31 rounds of work).  A range generally starts inside a tile; that leading fragment is computed FIRST, then the
//     data-parallel rounds, then the rest of the range.  The leading fragments have every length between nothing and
//     a whole tile, so the CUs reach their epilogues at different times for the rest of the launch: the output bursts
//     (all 256 CUs storing at once, then all computing) become a steady stream that overlaps the other CUs' MFMAs.
// A tile split between workgroups is finished by the workgroup that holds its FIRST K-tiles (the end of that
// workgroup's stream); the others (v+1, ...: the very start of theirs) hand over raw fp32 accumulators through `slabs`
// [G][8 waves][32 regs][64 lanes] x 16 B with the release / acquire protocol of cdna_hip_programming.md, Guideline 16
// (flags zeroed by a memset node ahead of every launch).  A workgroup writes its only partial before it ever waits, and
// it waits only for workgroups with a higher id: no cycles.
struct G3Plan {
    int hk;          // K-tile pairs per tile
    int rounds;      // data-parallel rounds R
    int rem_q, rem_r;// remainder pairs per workgroup: total = G rem_q + rem_r
};

// Walks one workgroup's stream of (tile, K-tile pair) on the scalar unit.
struct G3Walk {
    int stage;       // 0 leading fragment, 1 data-parallel rounds, 2 rest of the remainder range, 3 done
    int j;           // round (stage 1)
    int rr;          // position in the remainder pair space (stages 0 / 2): next pair to visit
    int tile, kp, seg_begin, seg_end;       // current tile, current pair in it, this workgroup's share [seg_begin, seg_end)
};
__device__ __forceinline__ void g3_walk_segment(G3Walk& w, const G3Plan& pl, int v, int G, int r1) {
    // enter the next segment; w.stage / w.j / w.rr say where we are
    if (w.stage == 1 && w.j < pl.rounds) {
        w.tile = w.j * G + v; w.kp = 0; w.seg_begin = 0; w.seg_end = pl.hk;
        ++w.j;
        return;
    }
    if (w.stage <= 1) w.stage = 2;
    if (w.rr >= r1) { w.stage = 3; w.kp = 0; w.seg_begin = 0; w.seg_end = 0; return; }
    const int t = __builtin_amdgcn_readfirstlane(w.rr / pl.hk);
    const int kb = w.rr - t * pl.hk;
    int ke = kb + (r1 - w.rr);
    ke = ke < pl.hk ? ke : pl.hk;
    w.tile = pl.rounds * G + t; w.kp = kb; w.seg_begin = kb; w.seg_end = ke;
    w.rr += ke - kb;
}
__device__ __forceinline__ void g3_walk_init(G3Walk& w, const G3Plan& pl, int v, int G, int r0, int r1, bool lead_first) {
    w.j = 0; w.rr = r0;
    const int t = __builtin_amdgcn_readfirstlane(r0 / pl.hk);
    if (lead_first && r0 < r1 && r0 - t * pl.hk != 0) {           // the range starts inside a tile: that fragment goes first
        w.stage = 0;
        g3_walk_segment(w, pl, v, G, r1);
        w.stage = 0;
    } else {
        w.stage = 1;
        g3_walk_segment(w, pl, v, G, r1);
    }
}
// one pair forward; returns true when that moved to another tile
__device__ __forceinline__ bool g3_walk_next(G3Walk& w, const G3Plan& pl, int v, int G, int r1) {
    if (++w.kp < w.seg_end) return false;
    if (w.stage == 0) w.stage = 1;
    g3_walk_segment(w, pl, v, G, r1);
    return true;
}

template <int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_g3p_kernel(const GemmParams p, const G3Plan pl, float* __restrict__ slabs, unsigned* flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2;
    const int G = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, gq = G >> 3, gr = G & 7;
    const int v = (xcd < gr ? xcd * (gq + 1) : gr * (gq + 1) + (xcd - gr) * gq) + (bid >> 3);
    auto range_begin = [&](int w) { return w * pl.rem_q + (w < pl.rem_r ? w : pl.rem_r); };
    const int r0 = range_begin(v), r1 = range_begin(v + 1);

    G3State s;
    g3_init_lane(s, p, smem, wave, lane);
    g3_zero(s);

    G3Walk wc, wn;                                 // compute cursor / the pair after it (DMA source)
#ifdef ME_DEV
    g3_walk_init(wc, pl, v, G, r0, r1, (p.debug & 2) != 0);
#else
    g3_walk_init(wc, pl, v, G, r0, r1, false);
#endif
    if (wc.stage == 3) return;                     // nothing to do (more workgroups than work)
    wn = wc;
    auto src_of = [&](const G3Walk& w) {
        const int tm = __builtin_amdgcn_readfirstlane(w.tile / p.tiles_n);
        return g3_make_src(p, tm, w.tile - tm * p.tiles_n);
    };
    G3Src cur = src_of(wc);
    G3Src nxt = cur;
    auto advance_next = [&]() {
        if (wn.stage == 3) return;
        if (g3_walk_next(wn, pl, v, G, r1)) {
            if (wn.stage == 3) nxt = g3_null_src(p);
            else nxt = src_of(wn);
        }
    };
    // prologue: half-tiles 0..6 of the stream (first K-tile complete, second without A-Y)
    {
        const int k = wc.kp * 2;
        g3_issue<0>(s, cur, 0, k); g3_issue<1>(s, cur, 0, k); g3_issue<2>(s, cur, 0, k); g3_issue<3>(s, cur, 0, k);
        g3_issue<0>(s, cur, 1, k + 1); g3_issue<1>(s, cur, 1, k + 1); g3_issue<2>(s, cur, 1, k + 1);
    }
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();    // wave row 1 runs one barrier behind (wave-uniform scalar branch)
    advance_next();                               // nxt / wn = the second pair of the stream

    while (wc.stage != 3) {
        const int kc = wc.kp * 2, kn = wn.kp * 2;
        g3_ktile<0>(s, cur, kc + 1, nxt, kn);
        g3_ktile<1>(s, nxt, kn, nxt, kn + 1);
        if (wc.kp + 1 == wc.seg_end) {
            // ---- seam: this workgroup's share [seg_begin, seg_end) of tile wc.tile is accumulated
            const int tm = __builtin_amdgcn_readfirstlane(wc.tile / p.tiles_n), tn = wc.tile - tm * p.tiles_n;
            const int64_t m0 = (int64_t)tm * G3_BM, n0 = (int64_t)tn * G3_BN;
            bool skip = false;
#ifdef ME_DEV
            if (p.debug & 1) {                    // dev: K-loops only
                float keep = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) keep += s.acc[i][j][0] + s.acc[i][j][1] + s.acc[i][j][2] + s.acc[i][j][3];
                if (keep == 1.2345e-30f) reinterpret_cast<float*>(p.C)[0] = keep;
                skip = true;
            }
#endif
#ifdef ME_DEV
            if ((p.debug & 8) && (wc.seg_begin != 0 || wc.seg_end < pl.hk)) skip = true;     // dev: no hand-over at all
#endif
            if (skip) {
            } else if (wc.seg_begin != 0) {
                // hand my partial sums to the workgroup that owns the tile's first K-tiles
                int le = lane;
                asm volatile("" : "+v"(le));      // (derive the address here, not ahead of the loop)
                // write-through (sc1) 16-byte stores: visible at agent scope once this wave's vmcnt drains, without the
                // release fence's write-back of the whole L2 (Guideline 16, form R1 / "publish-large")
                const __amdgpu_buffer_rsrc_t slab = __builtin_amdgcn_make_buffer_rsrc(
                    slabs + (int64_t)v * G3_SLAB_FLOATS, 0, G3_SLAB_FLOATS * 4, 0x00020000);
                const int voff = ((wave * 32) * 64 + le) * 16;
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, s.acc[i][j]), slab, voff + (i * 4 + j) * 1024, 0, /*sc1*/ 16);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // EVERY storing wave drains
                if (wr == 0) __builtin_amdgcn_s_barrier();          // realign the wave rows for a true workgroup barrier
                __syncthreads();
                if (tid == 0) __hip_atomic_store(flags + v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (wr == 1) __builtin_amdgcn_s_barrier();          // and stagger them again
            } else {
                if (wc.seg_end < pl.hk) {
                    // I hold the first K-tiles: collect the partial sums of the workgroups after me that cover the rest
                    const int tile_end = (wc.tile - pl.rounds * G + 1) * pl.hk;      // in the remainder pair space
                    int le = lane;
                    asm volatile("" : "+v"(le));
                    if (wr == 0) __builtin_amdgcn_s_barrier();
                    for (int w = v + 1; w < G && range_begin(w) < tile_end; ++w) {
                        if (range_begin(w) >= range_begin(w + 1)) continue;
                        if (tid == 0) {
                            unsigned spins = 0;
                            while (__hip_atomic_load(flags + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                                __builtin_amdgcn_s_sleep(8);
                                if (++spins > (1u << 26)) { flags[G] = 1u + (unsigned)w; break; }   // give up: error word
                            }
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        }
                        __syncthreads();
                        const f32x4* slab = reinterpret_cast<const f32x4*>(slabs + (int64_t)w * G3_SLAB_FLOATS) + (wave * 32) * 64 + le;
#pragma unroll
                        for (int i0 = 0; i0 < 8; i0 += 4) {          // 16 loads in flight
                            f32x4 t[4][4];
#pragma unroll
                            for (int i = 0; i < 4; ++i)
#pragma unroll
                                for (int j = 0; j < 4; ++j) t[i][j] = slab[((i0 + i) * 4 + j) * 64];
#pragma unroll
                            for (int i = 0; i < 4; ++i)
#pragma unroll
                                for (int j = 0; j < 4; ++j) s.acc[i0 + i][j] += t[i][j];
                        }
                    }
                    if (wr == 1) __builtin_amdgcn_s_barrier();
                }
                g3_epilogue<EPI>(p, s, m0, n0, lane);
            }
            g3_zero(s);
        }
        wc = wn;
        cur = nxt;
        advance_next();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing null DMAs must land before the LDS is released
    if (wr == 0) __builtin_amdgcn_s_barrier();   // match wave row 1's final barrier
}
#endif  // ME_DEV

// ---- one tile per workgroup
template <int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_g3_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2;

    // Work ids: F = p.g3_full_tiles whole tiles, then the tiles of the last, mostly empty round as S = p.g3_split parts
    // each (tile quantisation: the encoder's N = 768 outputs are 591 tiles = 2.31 rounds on 256 CUs; as whole tiles that is
    // 3 rounds of time, as 510 tiles + 81 x 3 thirds it is 2.31).  Block b runs on XCD b % 8: every XCD gets a contiguous
    // range of the whole tiles (neighbours share operand panels through its L2) AND its share of the parts, so all XCDs
    // carry the same amount of work.
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
#ifdef ME_DEV
    if ((p.debug >> 4) && bid < 256) {           // dev: stagger the first round (output bursts of the CUs spread out)
        const int n = (slot & 7) * (p.debug >> 4);
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(4);
    }
#endif
    const int F = p.g3_full_tiles;
    const int nf = (F >> 3) + (xcd < (F & 7) ? 1 : 0);
    const int base_f = xcd * (F >> 3) + (xcd < (F & 7) ? xcd : (F & 7));
    int tile, part = -1;
    if (slot < nf) {
        tile = base_f + slot;
    } else {
        const int base_all = xcd * (nwg >> 3) + (xcd < (nwg & 7) ? xcd : (nwg & 7));
        const int tid_ = base_all - base_f + (slot - nf);                 // index among the parts
        const int tq = __builtin_amdgcn_readfirstlane(tid_ / p.g3_split);
        tile = F + tq;
        part = tid_ - tq * p.g3_split;
    }
    const int tm = __builtin_amdgcn_readfirstlane(tile / p.tiles_n), tn = tile - tm * p.tiles_n;
    const int64_t m0 = (int64_t)tm * G3_BM, n0 = (int64_t)tn * G3_BN;

    G3State s;
    g3_init_lane(s, p, smem, wave, lane);
    g3_zero(s);
    const G3Src src = g3_make_src(p, tm, tn);

    const int nkt = (int)(p.K / G3_BK);          // even, >= 2 (g3_supported)
    int kt0 = 0, kt1 = nkt;
    if (part >= 0) {
        kt0 = part * p.g3_ktp;
        kt1 = kt0 + p.g3_ktp < nkt ? kt0 + p.g3_ktp : nkt;
    }
    g3_issue<0>(s, src, 0, kt0); g3_issue<1>(s, src, 0, kt0); g3_issue<2>(s, src, 0, kt0); g3_issue<3>(s, src, 0, kt0);
    g3_issue<0>(s, src, 1, kt0 + 1); g3_issue<1>(s, src, 1, kt0 + 1); g3_issue<2>(s, src, 1, kt0 + 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();

    // past the end of the K-range the source is a null descriptor (see g3_phase)
    const G3Src null = g3_null_src(p);
    for (int kt = kt0; kt < kt1 - 2; kt += 2) {
        g3_ktile<0>(s, src, kt + 1, src, kt + 2);
        g3_ktile<1>(s, src, kt + 2, src, kt + 3);
    }
    g3_ktile<0>(s, src, kt1 - 1, null, 0);
    g3_ktile<1>(s, null, 0, null, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wr == 0) __builtin_amdgcn_s_barrier();
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
    if (part >= 0) {
        // a part of a split tile: raw partial sums; the fold that follows the launch applies the real epilogue
        const int64_t row0 = (int64_t)(F / p.tiles_n) * G3_BM;
        g3_epilogue<5>(p, s, m0, n0, lane, p.g3_slabs + (int64_t)part * (p.M - row0) * p.N, row0);
        return;
    }
    g3_epilogue<EPI>(p, s, m0, n0, lane);
}

// ---- resident form: one workgroup per CU walks its work items (same ids and XCD mapping as gemm_g3_kernel: item
// (xcd, slot) for slot = c, c + G/8, ...), and the operand stream never stops at a tile boundary:
//   * the last two K-tiles of an item already fetch the first two of the next one (the K-loop's sources are just
//     (descriptor, K-tile) pairs), the one half-tile the loop would issue right AFTER the boundary goes out right before
//     the epilogue, and the epilogue touches no LDS -- so the whole prologue latency of the next tile, and its workgroup
//     launch, hide under the epilogue of this one;
//   * vmcnt retires in order and counts stores, so a K-loop that waits for "everything but my last 6 DMAs" right after an
//     epilogue would first drain the epilogue's stores.  The epilogue therefore issues an EXACT number of memory
//     operations (buffer stores / loads whose edge handling is the descriptor's bounds check, never a branch), and the
//     first K-tile after it waits with that many more operations allowed in flight (g3_phase<.., SEAM>): the stores
//     drain under the next tile's first two K-tiles.
// bf16 outputs / row operands only (launch3r checks).  Parts of split tiles (EPI 5 slabs) are the last items of a
// workgroup; should another item follow one, the queue is drained and re-primed.
// The bias is not added here: the accumulators START at the bias of their columns (g3r_bias / s.binit; alpha = 1),
// loaded for the NEXT tile at the top of this epilogue, ahead of its stores -- a load issued behind the stores could only be
// waited for by draining them.
// this lane's index, recomputed from the hardware where it is needed (two VALU operations): a lane id kept in a register
// across the item loop of the resident kernel is the first thing hipcc spills
__device__ __forceinline__ int g3_lane_now() {
    unsigned z = 0;
    asm volatile("" : "+v"(z));          // (opaque: not common-subexpression'd with, or hoisted to, an earlier copy)
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
}
// One ticket from a work counter, drawn by lane 0 alone.  Inline asm with the exec mask narrowed by hand: written as
// `if (lane == 0) atomic` hipcc waits for the result with vmcnt(0) at the join of the branch -- a drain of every store and DMA
// in flight.  The instruction is invisible to the compiler's own wait counting (it is OLDER than everything the callers
// wait for afterwards, which only makes their waits stricter); the caller retires it with a counted s_waitcnt.
__device__ __forceinline__ unsigned g3r_draw(unsigned* ctr) {
    unsigned old, zero = 0, one = 1;
    unsigned long long saved;
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, %4 sc0\n\ts_mov_b64 exec, %1"
                 : "=&v"(old), "=&s"(saved)
                 : "v"(zero), "v"(one), "s"(ctr)
                 : "memory");
    return old;      // valid in lane 0 once the operation has retired
}
// wave 0 hands the ticket to the other waves through the LDS word behind the operand buffers; whoever draws the last ticket
// of the launch (nx - 1) puts the counter back to zero
__device__ __forceinline__ void g3r_publish(unsigned drawn, unsigned* ctr, int nx, uint32_t lds_tick) {
    const unsigned t = __builtin_amdgcn_readfirstlane(drawn);
    if (t == (unsigned)(nx - 1)) {
        unsigned zero = 0;
        unsigned long long saved;
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tglobal_store_dword %1, %1, %2\n\ts_mov_b64 exec, %0"
                     : "=&s"(saved) : "v"(zero), "s"(ctr) : "memory");
    }
    asm volatile("ds_write_b32 %0, %1" ::"v"(lds_tick), "v"(t) : "memory");
}
struct G3Bias { f32x4 v[4]; };       // this lane's bias for n-tiles 0..3 of its wave column (accumulator layout)
__device__ __forceinline__ G3Bias g3r_bias(const __amdgpu_buffer_rsrc_t brs, int tn, int wave, int lane) {
    G3Bias b;
    (void)lane;
    lane = g3_lane_now();                // (derive the offset here: hoisted out of the item loop it would be spilled)
    const int col = tn * G3_BN + (wave & 3) * 64 + 4 * (lane >> 4);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)      // (no bias, or columns past N: zero records / out of range -> zeros)
        b.v[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, (col + nt * 16) * 4, 0, 0));
    return b;
}
__device__ __forceinline__ void g3r_set_binit(G3State& s, const G3Bias& b, bool zero) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) s.binit[j] = zero ? z : b.v[j];
    // (used HERE: the wait for the loads lands in the caller's straight-line code, with the number of younger stores known;
    // at the first real use, behind the item loop's joins, it would be a drain)
    asm volatile("" ::"v"(s.binit[0]), "v"(s.binit[1]), "v"(s.binit[2]), "v"(s.binit[3]));
}

// Store pattern: a store instruction that covers 16 rows x 64 bytes (what the permlane16 re-deal alone gives) costs a CU
// 4.6 us per 256 x 256 bf16 tile, one that covers 8 rows x 128 bytes -- whole cache lines -- 1.7 us (tools/store_probe).
// So the two 64-byte halves (q = 0 / 1) of the wave's 128-byte row segment are re-dealt once more, between the lanes of
// rows r and r + 8 (DPP row_ror:8 under a bank mask): afterwards half A holds rows 0..7 and half B rows 8..15 of the
// 16-row slab, lane (r = l & 15, g = l >> 4) owning the 16-byte chunk (r >> 3) * 4 + 2 (g & 1) + (g >> 1) of row r & 7.
__device__ __forceinline__ void g3r_rows8(f32x4& x0, f32x4& x1, f32x4& y0, f32x4& y1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned a0 = __float_as_uint(x0[e]), a1 = __float_as_uint(x1[e]), b0 = __float_as_uint(y0[e]), b1 = __float_as_uint(y1[e]);
        // lanes 8..15 of every row of 16 take the q = 1 value of the lane 8 below; lanes 0..7 the q = 0 value of the lane 8 above
        x0[e] = __uint_as_float(__builtin_amdgcn_update_dpp(a0, b0, 0x128, 0xf, 0xc, false));
        x1[e] = __uint_as_float(__builtin_amdgcn_update_dpp(a1, b1, 0x128, 0xf, 0xc, false));
        y0[e] = __uint_as_float(__builtin_amdgcn_update_dpp(b0, a0, 0x128, 0xf, 0x3, false));
        y1[e] = __uint_as_float(__builtin_amdgcn_update_dpp(b1, a1, 0x128, 0xf, 0x3, false));
    }
}

__device__ __forceinline__ void g3r_rows8_packed(u32x4& x, u32x4& y) {      // the same re-deal on packed bf16 pairs
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned a = x[e], b = y[e];
        x[e] = __builtin_amdgcn_update_dpp(a, b, 0x128, 0xf, 0xc, false);
        y[e] = __builtin_amdgcn_update_dpp(b, a, 0x128, 0xf, 0x3, false);
    }
}

// Which LANES hold a cache line matters as much as which lines an instruction covers: with the lanes of one 128-byte row
// segment scattered over the wave (r, r + 8, r + 16 ..: what the re-deals above leave) a store instruction costs the CU
// ~77 clocks, with eight CONSECUTIVE lanes per line ~31 (time stamps in the kernel, debug bit 8; tools/store_probe).  So the
// packed 16-byte chunks take one more trip through the lane crossbar (ds_bpermute_b32: no LDS memory involved, the operand
// buffers stay untouched) into the order lane t = row (t >> 3), chunk (t & 7); row operands are loaded in that order and
// taken the opposite way.
__device__ __forceinline__ u32x4 g3r_lanes(const u32x4& v, int addr) {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)v[e]);
    return o;
}

// EPI: 0 bias, 1 GELU (PRE: 0 nothing saved, 1 pre-activation saved, 2 gelu'(pre-activation) saved), 2 + residual row
// operand, 3 * gelu'(row operand), 6 * row operand.
// PRE 3 (EPI 0 / 1): a LayerNorm folded into this Linear (GemmParams::row_affine / col_shift): the accumulators start at zero
// and the epilogue applies v = rstd_m * acc + (-rstd_m mean_m) * s[n] + c[n] in the accumulator layout (one row per lane and
// 16-row slab, four consecutive columns per register quad), ahead of the activation; nothing is saved.
template <int EPI, int PRE>
__device__ __forceinline__ void g3_epilogue_r(const GemmParams& p, G3State& s, int tm, int tn, int lane, const G3Src& nxt, int nk,
                                              const __amdgpu_buffer_rsrc_t brs, int ntn, bool next_zero, unsigned* ctr, int nx,
                                              uint32_t lds_tick) {
    constexpr bool SAVE = PRE == 1 || PRE == 2, LNF = PRE == 3;
    // (claimed schedule: wave 0 draws the ticket for the item after next FIRST, ahead of every store of this epilogue)
    unsigned drawn = 0;
    if (ctr && s.wave == 0) drawn = g3r_draw(ctr);
    (void)lane;
    lane = g3_lane_now();
    const int wr = s.wave >> 2, wc = s.wave & 3;
    const int r = lane & 15, g = lane >> 4;
    const int64_t m0 = (int64_t)tm * G3_BM, n0 = (int64_t)tn * G3_BN;
    int64_t rows = p.M - m0, cols = p.N - n0;
    rows = rows < G3_BM ? rows : G3_BM;
    cols = cols < G3_BN ? cols : G3_BN;
    auto tile_rsrc = [&](const void* base, int64_t ld) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base)) + (m0 * ld + n0) * 2, 0,
                                                 (int)(((rows - 1) * ld + cols) * 2), 0x00020000);
    };
#ifdef ME_DEV
    // dev (debug bit 4): the stores go nowhere (zero-record descriptor), everything else unchanged
    // (debug bit 2 with bit 4: only the first workgroup of every XCD keeps its stores)
    const bool drop = (p.debug & 4) && !((p.debug & 2) && (blockIdx.x >> 3) == 0);
    const __amdgpu_buffer_rsrc_t crs = drop ? __builtin_amdgcn_make_buffer_rsrc(p.C, 0, 0, 0x00020000) : tile_rsrc(p.C, p.ldc);
#else
    const __amdgpu_buffer_rsrc_t crs = tile_rsrc(p.C, p.ldc);
#endif
    const __amdgpu_buffer_rsrc_t prs = SAVE ? tile_rsrc(p.preact, p.ldpre) : crs;
    const __amdgpu_buffer_rsrc_t rrs = EPI == 2 ? tile_rsrc(p.residual, p.ldres) : (EPI == 3 || EPI == 6) ? tile_rsrc(p.aux, p.ldaux) : crs;
    const int rop_ld = (int)(EPI == 2 ? p.ldres : p.ldaux);
    // memory side: lane t = row t >> 3 of an 8-row half slab, 16-byte chunk t & 7 of the wave's 128-byte row segment.  A
    // chunk past the column edge gets an offset no descriptor admits; rows past the row edge fall behind the descriptor's end.
    const int colb = wc * 128 + (lane & 7) * 16;
    const bool ok = (colb >> 1) + 8 <= (int)cols;
    const int row = wr * 128 + (lane >> 3);
    const uint32_t coff = ok ? (uint32_t)(row * (int)p.ldc * 2 + colb) : 0x80000000u;
    const uint32_t poff = ok && SAVE ? (uint32_t)(row * (int)p.ldpre * 2 + colb) : 0x80000000u;
    const uint32_t roff = ok ? (uint32_t)(row * rop_ld * 2 + colb) : 0x80000000u;
    const int cstep = (int)p.ldc * 16, pstep = (int)p.ldpre * 16, rstep = rop_ld * 16;       // 8 rows, bytes
    // register side (after g3r_rows8): lane (r, g) = row r & 7, chunk 4 (r >> 3) + 2 (g & 1) + (g >> 1).  to_mem: the lane
    // that holds memory lane t's chunk; to_reg: the memory lane that holds this lane's chunk (x 4: bpermute byte addresses)
    const int to_mem = 4 * ((lane >> 3) + 8 * ((lane >> 2) & 1) + 16 * (((lane >> 1) & 1) | ((lane & 1) << 1)));
    const int to_reg = 4 * (8 * (r & 7) + 4 * (r >> 3) + 2 * (g & 1) + (g >> 1));
    auto fetch = [&](const int mt, u32x4 (&raw)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) raw[h] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)(roff + (2 * mt + h) * rstep), 0, 0);
    };
    auto unpack = [](const u32x4& rw, f32x4& a, f32x4& b) {
        a[0] = __uint_as_float(rw[0] << 16); a[1] = __uint_as_float(rw[0] & 0xffff0000u);
        a[2] = __uint_as_float(rw[1] << 16); a[3] = __uint_as_float(rw[1] & 0xffff0000u);
        b[0] = __uint_as_float(rw[2] << 16); b[1] = __uint_as_float(rw[2] & 0xffff0000u);
        b[2] = __uint_as_float(rw[3] << 16); b[3] = __uint_as_float(rw[3] & 0xffff0000u);
    };
    auto pack = [](const f32x4& a, const f32x4& b) {
        bf16x8 o;
        o[0] = (bf16_t)a[0]; o[1] = (bf16_t)a[1]; o[2] = (bf16_t)a[2]; o[3] = (bf16_t)a[3];
        o[4] = (bf16_t)b[0]; o[5] = (bf16_t)b[1]; o[6] = (bf16_t)b[2]; o[7] = (bf16_t)b[3];
        return __builtin_bit_cast(u32x4, o);
    };
    constexpr int AHEAD = EPI == 3 ? 4 : 6;      // row-operand slabs in flight ahead of their use (more spills: into the K-loop for gelu', onto the ticket register otherwise)
    u32x4 rowop[8][2];
    if (EPI == 2 || EPI == 3 || EPI == 6) {
#pragma unroll
        for (int mt = 0; mt < AHEAD; ++mt) fetch(mt, rowop[mt]);
    }
    // folded LayerNorm: this tile's per-row pairs (rows wr*128 + 16 mt + r) and per-column s / c, all ahead of the DMA below
    f32x2 lnf_row[8];
    G3Bias lnf_s, lnf_c;
    if (LNF) {
        const __amdgpu_buffer_rsrc_t rars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.row_affine) + m0 * 2, 0, (int)(rows * 8), 0x00020000);
        const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.col_shift), 0, (int)(p.N * 4), 0x00020000);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)      // (rows past the edge: out of range -> zeros; their outputs are never stored)
            lnf_row[mt] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rars, (wr * 128 + mt * 16 + r) * 8, 0, 0));
        lnf_s = g3r_bias(srs, tn, s.wave, lane);
        lnf_c = g3r_bias(brs, tn, s.wave, lane);
    }
    // the half-tile phase 0 of the next K-tile would issue (see SEAM), behind the first row-operand loads so that their
    // wait does not include it; then the next tile's bias
    g3_issue<3>(s, nxt, 1, nk);
    const G3Bias nb = g3r_bias(brs, ntn, s.wave, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        f32x4 ro[2][2];
        if (EPI == 2 || EPI == 3 || EPI == 6) {
            unpack(g3r_lanes(rowop[mt][0], to_reg), ro[0][0], ro[0][1]);
            unpack(g3r_lanes(rowop[mt][1], to_reg), ro[1][0], ro[1][1]);
            if (mt + AHEAD < 8) fetch(mt + AHEAD, rowop[mt + AHEAD]);
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 v[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            v[q][0] = s.acc[mt][2 * q]; v[q][1] = s.acc[mt][2 * q + 1];
            if (LNF) {
                v[q][0] = v[q][0] * lnf_row[mt][0] + (lnf_s.v[2 * q] * lnf_row[mt][1] + lnf_c.v[2 * q]);
                v[q][1] = v[q][1] * lnf_row[mt][0] + (lnf_s.v[2 * q + 1] * lnf_row[mt][1] + lnf_c.v[2 * q + 1]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[q][0][e]), __float_as_uint(v[q][1][e]), false, false);
                v[q][0][e] = __uint_as_float(sw[0]);
                v[q][1][e] = __uint_as_float(sw[1]);
            }
        }
        if ((EPI == 0 || EPI == 1) && !SAVE) {
            // no row operand, one output: the arithmetic runs in the old layout and the PACKED result is re-dealt (half the
            // DPP moves)
            if (EPI == 1) {
                v[0][0] = gelu_bf16_4(v[0][0]); v[0][1] = gelu_bf16_4(v[0][1]);
                v[1][0] = gelu_bf16_4(v[1][0]); v[1][1] = gelu_bf16_4(v[1][1]);
            }
            u32x4 o0 = pack(v[0][0], v[0][1]), o1 = pack(v[1][0], v[1][1]);
            g3r_rows8_packed(o0, o1);
            __builtin_amdgcn_raw_buffer_store_b128(g3r_lanes(o0, to_mem), crs, (int)(coff + (2 * mt) * cstep), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(g3r_lanes(o1, to_mem), crs, (int)(coff + (2 * mt + 1) * cstep), 0, 0);
            continue;
        }
        g3r_rows8(v[0][0], v[0][1], v[1][0], v[1][1]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {           // half A: rows 0..7 of the slab, half B: rows 8..15
            f32x4 v0 = v[h][0], v1 = v[h][1];
            if (EPI == 1) {
                if (PRE == 2) {
                    // gelu and gelu' from the same Phi / Gaussian parts: the backward GEMM multiplies by the saved factor.  (The
                    // erf form stays here: with BOTH outputs wanted it shares one exponential between them, and the two
                    // polynomial chains of the bf16-mode forms measured no faster -- 344 against 339 us per fc1 launch.)
                    f32x4 ph0, ga0, ph1, ga1;
                    phi_parts4(v0, ph0, ga0);
                    phi_parts4(v1, ph1, ga1);
                    const f32x4 d0 = ph0 + v0 * ga0 * 0.3989422804014327f, d1 = ph1 + v1 * ga1 * 0.3989422804014327f;
                    __builtin_amdgcn_raw_buffer_store_b128(g3r_lanes(pack(d0, d1), to_mem), prs, (int)(poff + (2 * mt + h) * pstep), 0, 0);
                    v0 *= ph0;
                    v1 *= ph1;
                } else {
                    if (SAVE) __builtin_amdgcn_raw_buffer_store_b128(g3r_lanes(pack(v0, v1), to_mem), prs, (int)(poff + (2 * mt + h) * pstep), 0, 0);
                    v0 = gelu_bf16_4(v0);
                    v1 = gelu_bf16_4(v1);
                }
            }
            if (EPI == 6) { v0 *= ro[h][0]; v1 *= ro[h][1]; }
            if (EPI == 3) {
                v0 *= gelu_bf16_grad4(ro[h][0]);
                v1 *= gelu_bf16_grad4(ro[h][1]);
            }
            if (EPI == 2) { v0 += ro[h][0]; v1 += ro[h][1]; }
            __builtin_amdgcn_raw_buffer_store_b128(g3r_lanes(pack(v0, v1), to_mem), crs, (int)(coff + (2 * mt + h) * cstep), 0, 0);
        }
    }
    // the next tile's accumulators start at its bias (s.binit)
    __builtin_amdgcn_sched_barrier(0);
    g3r_set_binit(s, nb, next_zero);
    if (ctr && s.wave == 0) {
        // everything this epilogue issued behind the draw may stay in flight: the A-Y half-tile (2), the bias (4), the
        // stores (16 / 32) and the row operands (16)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 + 4 + 16 + (SAVE ? 16 : 0) + ((EPI == 2 || EPI == 3 || EPI == 6) ? 16 : 0)) : "memory");
        g3r_publish(drawn, ctr, nx, lds_tick);
    }
}

// memory operations one g3_epilogue_r issues per wave behind the next tile's A-Y half-tile: >= the stores (+ the later
// row-operand loads); an under-count only makes the wait stricter
template <int EPI, int PRE> constexpr int g3r_seam() { return (PRE == 1 || PRE == 2) ? 32 : EPI >= 2 ? 20 : 16; }

template <int EPI, int PRE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_g3r_kernel(const GemmParams p) {
    constexpr int SEAM = g3r_seam<EPI, PRE>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2;

    const int bid = blockIdx.x, xcd = bid & 7, c = bid >> 3, G8 = gridDim.x >> 3;
    const int tiles = p.tiles_m * p.tiles_n, F = p.g3_full_tiles;
    const int nwork = F + (tiles - F) * p.g3_split;
    const int nf = (F >> 3) + (xcd < (F & 7) ? 1 : 0);                 // whole tiles / all items of this XCD
    const int nx = (nwork >> 3) + (xcd < (nwork & 7) ? 1 : 0);
    const int base_f = xcd * (F >> 3) + (xcd < (F & 7) ? xcd : (F & 7));
    const int base_all = xcd * (nwork >> 3) + (xcd < (nwork & 7) ? xcd : (nwork & 7));
    const int nkt = (int)(p.K / G3_BK);
    auto decode = [&](int slot, int& tile, int& part, int& kt0, int& kt1) {
        part = -1; kt0 = 0; kt1 = nkt;
        if (slot < nf) {
            tile = base_f + slot;
        } else {
            const int pi = base_all - base_f + (slot - nf);
            const int tq = __builtin_amdgcn_readfirstlane(pi / p.g3_split);
            tile = F + tq;
            part = pi - tq * p.g3_split;
            kt0 = part * p.g3_ktp;
            kt1 = kt0 + p.g3_ktp < nkt ? kt0 + p.g3_ktp : nkt;
        }
    };
    auto src_of = [&](int tile, int& tm, int& tn) {
        tm = __builtin_amdgcn_readfirstlane(tile / p.tiles_n);
        tn = tile - tm * p.tiles_n;
        return g3_make_src(p, tm, tn);
    };
    int slot = c;
    if (slot >= nx) return;
#ifdef ME_DEV
    if (p.debug >> 4) {                          // dev: stagger the CUs of an XCD (their output bursts spread out)
        const int n = c * (p.debug >> 4);
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(4);
    }
#endif

    G3State s;
    g3_init_lane(s, p, smem, wave, lane);
    const G3Src null = g3_null_src(p);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? (int)(p.N * 4) : 0, 0x00020000);
    // SEAM stores nothing admits: the in-order queue looks the same ahead of the first item as behind an epilogue
    auto prime = [&]() {
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < SEAM; ++i) __builtin_amdgcn_raw_buffer_store_b128(z, null.a, 0, 0, 0);
    };

    const bool dyn = p.g3_tickets != nullptr;
    unsigned* const ctr = dyn ? p.g3_tickets + xcd * 16 : nullptr;
    const uint32_t lds_tick = (uint32_t)(uintptr_t)smem + G3_LDS;
    unsigned drawn0 = 0;
    if (dyn && wave == 0) drawn0 = g3r_draw(ctr);          // item 1 (item 0 is this workgroup's own slot)

    int tile, part, kt0, kt1, tm, tn;
    decode(slot, tile, part, kt0, kt1);
    G3Src cur = src_of(tile, tm, tn);
    g3_issue<0>(s, cur, 0, kt0); g3_issue<1>(s, cur, 0, kt0); g3_issue<2>(s, cur, 0, kt0); g3_issue<3>(s, cur, 0, kt0);
    g3_issue<0>(s, cur, 1, kt0 + 1); g3_issue<1>(s, cur, 1, kt0 + 1); g3_issue<2>(s, cur, 1, kt0 + 1); g3_issue<3>(s, cur, 1, kt0 + 1);
    {
        const G3Bias b0 = g3r_bias(brs, tn, wave, lane);
        g3r_set_binit(s, b0, part >= 0 || PRE == 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (dyn && wave == 0) g3r_publish(drawn0, ctr, nx, lds_tick);
    prime();
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();

#ifdef ME_DEV
    // dev: time stamps (s_memtime) of waves 0 and 4: [workgroup][wave row][item][8] = item start, first K-tile done,
    // second K-tile done, K-loop done, rows realigned, epilogue done
    unsigned long long* trace = reinterpret_cast<unsigned long long*>(p.colsum_ws);
    int item = 0;
#define G3R_STAMP(i)                                                                                            \
    if (trace && (wave & 3) == 0 && item < 16) {                                                                \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                             \
        if (lane == 0) trace[(((size_t)bid * 2 + wr) * 16 + item) * 8 + (i)] = t_;                              \
    }
#else
#define G3R_STAMP(i)
#endif
    // Which item comes next: static (slot + G8: every workgroup owns a fixed list) or, with p.g3_tickets, CLAIMED from the
    // XCD's counter -- a CU that is slow, or that could not take its workgroup for a while because a communication kernel
    // sat on it, then simply ends up with fewer tiles instead of holding the whole launch back (the first item stays static:
    // no round trip before the first DMA).  Wave 0 draws one item ahead: in the prologue for item 1, at the top of the
    // epilogue of item i (ahead of its stores, so that retiring the draw does not drain them) for item i + 2, and leaves the
    // ticket in one LDS word; every wave picks it up behind the first K-tile pair of the following item (>= 2 pairs per item
    // and whole tiles only in this mode: launch3r).  Tickets 0 .. nx - 1 are drawn per XCD and launch (nx - G8 hits, one miss
    // per workgroup): whoever draws the last one zeroes the counter for the next launch on this stream.
    while (true) {
        G3R_STAMP(0)
        int nslot = 0, ntile = tile, npart = -1, nkt0 = 0, nkt1 = 2, ntm = 0, ntn = 0;
        bool has_next = false;
        G3Src nxt = null;
        auto resolve_next = [&](int ns) {
            nslot = ns;
            has_next = ns < nx;
            if (has_next) {
                decode(nslot, ntile, npart, nkt0, nkt1);
                nxt = src_of(ntile, ntm, ntn);
            }
        };
        if (!dyn) resolve_next(slot + G8);
        const int np = (kt1 - kt0) >> 1;
        {
            G3Src sb = cur;
            int kb = kt0 + 2, kc = kt0 + 3;
            if (np == 1) { sb = nxt; kb = nkt0; kc = nkt0 + 1; }
            g3_ktile<0, false, SEAM>(s, cur, 0, sb, kb);
            G3R_STAMP(1)
            g3_ktile<1>(s, sb, kb, sb, kc);
        }
        G3R_STAMP(2)
        if (dyn) {
            unsigned t;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(lds_tick) : "memory");
            resolve_next(G8 + (int)__builtin_amdgcn_readfirstlane(t));
        }
        for (int i = 1; i < np; ++i) {
            const int k = kt0 + 2 * i;
            G3Src sb = cur;
            int kb = k + 2, kc = k + 3;
            if (i == np - 1) { sb = nxt; kb = nkt0; kc = nkt0 + 1; }
            g3_ktile<0>(s, cur, k + 1, sb, kb);
            g3_ktile<1>(s, sb, kb, sb, kc);
        }
        // the two wave rows run their epilogues SIDE BY SIDE: left one barrier apart, row 1 could not start its epilogue
        // before row 0 had finished its own and reached the next K-tile's first barrier, and row 0 would then wait out
        // row 1's (measured with the time stamps below: 4.6 k of 40 k clocks per tile).  Row 0 gives up its one-barrier
        // lead here and row 1 re-opens it behind the epilogue.
        G3R_STAMP(3)
        if (wr == 0) __builtin_amdgcn_s_barrier();
        G3R_STAMP(4)
        if (part >= 0) {
            g3_issue<3>(s, nxt, 1, nkt0 + 1);
            const int64_t row0 = (int64_t)(F / p.tiles_n) * G3_BM;
            g3_epilogue<5>(p, s, (int64_t)tm * G3_BM, (int64_t)tn * G3_BN, g3_lane_now(), p.g3_slabs + (int64_t)part * (p.M - row0) * p.N, row0);
            const G3Bias nb = g3r_bias(brs, ntn, wave, 0);
            g3r_set_binit(s, nb, npart >= 0 || PRE == 3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            prime();
        } else {
            g3_epilogue_r<EPI, PRE>(p, s, tm, tn, 0, nxt, nkt0 + 1, brs, ntn, npart >= 0 || PRE == 3, has_next ? ctr : nullptr, nx, lds_tick);
        }
        G3R_STAMP(5)
#ifdef ME_DEV
        ++item;
#endif
        if (!has_next) break;
        if (wr == 1) __builtin_amdgcn_s_barrier();
        slot = nslot; tile = ntile; part = npart; kt0 = nkt0; kt1 = nkt1; tm = ntm; tn = ntn;
        cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing null DMAs must land before the LDS is released
}

// ---- wgrad: one (output tile, K-range) per workgroup; raw fp32 partial sums into slab blockIdx.y ... the deterministic
// fold (splitk_reduce_kernel, gemm.hip) sums the slabs and applies the epilogue.  Work ids are split-major (id = split *
// tiles + tile): the workgroups an XCD runs together read the SAME rows of dY and X at the same time, so every operand
// row comes out of HBM once and is shared through that XCD's L2.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_g3tn_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tiles = p.tiles_m * p.tiles_n;
    const int split = __builtin_amdgcn_readfirstlane(wgid / tiles);
    const int tile = wgid - split * tiles;
    const int tm = __builtin_amdgcn_readfirstlane(tile / p.tiles_n), tn = tile - tm * p.tiles_n;
    const int64_t m0 = (int64_t)tm * G3_BM, n0 = (int64_t)tn * G3_BN;

    G3State s;
    g3_init_lane_tn(s, p, smem, wave, lane);
    g3_zero(s);
    const G3Src src = g3_make_src_tn(p, tm, tn);
    const int kt0 = split * p.ksteps_per_split, kt1 = kt0 + p.ksteps_per_split;       // (an even count; tiles past K read zeros)

    g3_issue<0>(s, src, 0, kt0); g3_issue<1>(s, src, 0, kt0); g3_issue<2>(s, src, 0, kt0); g3_issue<3>(s, src, 0, kt0);
    g3_issue<0>(s, src, 1, kt0 + 1); g3_issue<1>(s, src, 1, kt0 + 1); g3_issue<2>(s, src, 1, kt0 + 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();
    const G3Src null = g3_null_src(p);
    // bias gradient: the N-tiles of one (M-tile, split) stage the same A rows -- they share the column sums pair by pair of
    // K-tiles, round-robin (cs_turn = pairs until this workgroup's next turn)
    const bool do_cs = p.colsum_ws != nullptr;
    s.cs[0] = s.cs[1] = 0.f;
    int cs_turn = do_cs ? tn : -1;
    auto my_turn = [&]() {
        if (!do_cs) return false;
        const bool mine = cs_turn == 0;
        cs_turn = mine ? p.tiles_n - 1 : cs_turn - 1;
        return mine;
    };
    for (int kt = kt0; kt < kt1 - 2; kt += 2) {
        const bool c = my_turn();
        g3_ktile<0, true>(s, src, kt + 1, src, kt + 2, c);
        g3_ktile<1, true>(s, src, kt + 2, src, kt + 3, c);
    }
    {
        const bool c = my_turn();
        g3_ktile<0, true>(s, src, kt1 - 1, null, 0, c);
        g3_ktile<1, true>(s, null, 0, null, 0, c);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wr == 0) __builtin_amdgcn_s_barrier();
    if (do_cs) {
        // lanes l, l+16, l+32, l+48 hold the four k-groups of column l & 15: fold, then one row of partial sums per
        // (split, N-tile): [split * tiles_n + tn][M]
        float c0 = s.cs[0], c1 = s.cs[1];
        c0 += __shfl_xor(c0, 16, 64); c0 += __shfl_xor(c0, 32, 64);
        c1 += __shfl_xor(c1, 16, 64); c1 += __shfl_xor(c1, 32, 64);
        if (lane < 16) {
            float* row = p.colsum_ws + ((int64_t)split * p.tiles_n + tn) * p.M;
            const int wcol = wave & 3;
            const int64_t ma = m0 + wr * 128 + wcol * 16 + lane, mb = ma + 64;
            if (ma < p.M) row[ma] = c0;
            if (mb < p.M) row[mb] = c1;
        }
    }
#ifdef ME_DEV
    if (p.debug & 1) {
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) keep += s.acc[i][j][0] + s.acc[i][j][1] + s.acc[i][j][2] + s.acc[i][j][3];
        if (keep == 1.2345e-30f) reinterpret_cast<float*>(p.C)[0] = keep;
        return;
    }
#endif
    g3_epilogue<5>(p, s, m0, n0, lane, reinterpret_cast<float*>(p.C) + (int64_t)split * p.slab_stride, 0);
}

int g3_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    }
    return n;
}

// The work counters of the resident kernel: every stream gets its own set (launches on a stream are serialised, and the
// kernel leaves its counters at zero), handed out on the host from a per-device pool (keyed by the STREAM's device) that is
// allocated and zeroed once, at the first resident launch on the device; a set is zeroed again, on its own stream, when it is
// first handed to a stream.  A launch that is being CAPTURED into a hipGraph always runs the static schedule: a kernel node
// keeps the counter set of its capture stream, but replays run on whatever stream launches the graph (torch.cuda.graph
// captures every graph on one shared stream), so replays on different streams -- or a replay next to eager work on the capture
// stream -- would draw from ONE set and break the "exactly nx draws, the last one resets" invariant.  On a free GPU the two
// schedules time the same (tools/gemm_dev g3 / g3s); claiming matters next to a communication kernel, i.e. in eager training.
unsigned* g3r_tickets(hipStream_t stream) {
    constexpr int SETS = 64, SET_WORDS = 8 * 16;
    struct Pool {
        unsigned* base = nullptr;
        bool failed = false;
        int next = 0;
        std::map<hipStream_t, int> idx;
    };
    static std::mutex mu;
    static std::map<int, Pool> pools;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;      // static schedule
    int dev = 0;
    if (hipStreamGetDevice(stream, &dev) != hipSuccess) (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    Pool& P = pools[dev];
    if (!P.base) {
        if (P.failed) return nullptr;
        unsigned* buf = nullptr;
        if (hipMalloc(&buf, (size_t)SETS * SET_WORDS * sizeof(unsigned)) != hipSuccess ||
            hipMemset(buf, 0, (size_t)SETS * SET_WORDS * sizeof(unsigned)) != hipSuccess) {
            P.failed = true;
            return nullptr;
        }
        P.base = buf;
    }
    auto it = P.idx.find(stream);
    if (it == P.idx.end()) {
        if (P.next >= SETS) return nullptr;
        unsigned* set = P.base + (size_t)P.next * SET_WORDS;
        if (hipMemsetAsync(set, 0, SET_WORDS * sizeof(unsigned), stream) != hipSuccess) return nullptr;
        it = P.idx.emplace(stream, P.next++).first;
    }
    return P.base + (size_t)it->second * SET_WORDS;
}

template <int EPI, int PRE> int launch3r(const GemmParams& q0, int G, hipStream_t stream) {
    static OncePerDevice once;
    if (once.need())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_g3r_kernel<EPI, PRE>), hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS + 64);
    GemmParams q = q0;
    // claimed items need >= 2 K-tile pairs per item and whole tiles only (see the kernel)
    q.g3_tickets = (q.K >= 4 * G3_BK && q.g3_split <= 1 && gemm_dev().g3_persistent == 1) ? g3r_tickets(stream) : nullptr;
#ifdef ME_DEV
    if (gemm_dev().tail_split == 2) q.g3_tickets = nullptr;          // dev: "g3s" = static schedule
    q.colsum_ws = (q.debug & 8) ? reinterpret_cast<float*>(g_gemm_dev_trace) : nullptr;
#endif
    hipLaunchKernelGGL((gemm_g3r_kernel<EPI, PRE>), dim3((unsigned)G), dim3(512), G3_LDS + 64, stream, q);
    ME_CHECK_LAUNCH("me_gemm(g3 resident)");
    return ME_OK;
}

int launch3r_any(int epi, int pre, const GemmParams& q, int G, hipStream_t stream) {
    switch (epi) {
        case 0: return pre == 3 ? launch3r<0, 3>(q, G, stream) : launch3r<0, 0>(q, G, stream);
        case 1: return pre == 3 ? launch3r<1, 3>(q, G, stream) : pre == 2 ? launch3r<1, 2>(q, G, stream) : pre ? launch3r<1, 1>(q, G, stream) : launch3r<1, 0>(q, G, stream);
        case 2: return launch3r<2, 0>(q, G, stream);
        case 3: return launch3r<3, 0>(q, G, stream);
        default: return launch3r<6, 0>(q, G, stream);
    }
}

template <int EPI> int launch3e(const GemmParams& p, void* ws, hipStream_t stream) {
    static OncePerDevice once;
    if (once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_g3_kernel<EPI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS);
#ifdef ME_DEV
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_g3p_kernel<EPI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS);
#endif
    }
    const int tiles = p.tiles_m * p.tiles_n;
#ifdef ME_DEV
    if (ws) {
        G3Plan pl;
        pl.hk = (int)(p.K / (2 * G3_BK));
        int G = g3_cus();
        G = G < tiles ? G : tiles;
        pl.rounds = tiles / G;
        // keep at least one tile's worth of remainder per workgroup (that is what de-phases the epilogues) when there
        // are rounds to take it from
        if (pl.rounds > 0 && (int64_t)(tiles - pl.rounds * G) * pl.hk < (int64_t)G * pl.hk) --pl.rounds;
        const int64_t rem_pairs = (int64_t)(tiles - pl.rounds * G) * pl.hk;
        pl.rem_q = (int)(rem_pairs / G);
        pl.rem_r = (int)(rem_pairs % G);
        float* slabs = reinterpret_cast<float*>(ws);
        unsigned* flags = reinterpret_cast<unsigned*>(slabs + (size_t)G * G3_SLAB_FLOATS);
        if (hipMemsetAsync(flags, 0, (size_t)(G + 1) * sizeof(unsigned), stream) != hipSuccess) {
            me_set_error("me_gemm(g3): flag reset failed");
            return ME_ERR_HIP;
        }
        hipLaunchKernelGGL((gemm_g3p_kernel<EPI>), dim3((unsigned)G), dim3(512), G3_LDS, stream, p, pl, slabs, flags);
        ME_CHECK_LAUNCH("me_gemm(g3p)");
        return ME_OK;
    }
#endif
    (void)ws;
    GemmParams q = p;
    if (q.g3_split <= 1 || q.g3_slabs == nullptr) { q.g3_full_tiles = tiles; q.g3_split = 1; q.g3_ktp = 0; q.g3_slabs = nullptr; }
    const int nwg = q.g3_full_tiles + (tiles - q.g3_full_tiles) * q.g3_split;
    // the resident form (one workgroup per CU, operand stream running through the epilogues) whenever every CU gets work
    // and the output / row operands are bf16 with tile-local 32-bit offsets
    if (gemm_dev().g3_persistent == 1) {
        int repi = EPI <= 3 ? EPI : -1, pre = (EPI == 1 && p.preact) ? 1 : 0;
        const bool plain = p.beta == 0.0f && p.out_group_rows == 0 && p.res_row_mod == 0 && !p.colscale && !p.residual;
        if (EPI == 4 && p.row_affine && !p.flags && plain && !p.preact && !p.aux) {       // folded LayerNorm (bias / GELU forms)
            repi = p.act == ME_ACT_GELU ? 1 : 0;
            pre = 3;
        }
        if (EPI == 4 && p.flags && !p.row_affine) {
            // the two halves of the "save gelu'" pair (pick_epi sends flagged descriptors to the generic epilogue)
            if (plain && p.flags == ME_GEMM_SAVE_GELU_GRAD && p.act == ME_ACT_GELU && p.preact && !p.aux) { repi = 1; pre = 2; }
            if (plain && p.flags == ME_GEMM_AUX_IS_FACTOR && p.act == ME_ACT_NONE && p.aux && p.aux_dtype == ME_BF16 && !p.preact) repi = 6;
        }
        const int G = g3_cus() & ~7;
        const int64_t ldmax = std::max(std::max(p.ldc, p.preact ? p.ldpre : 0), std::max(p.residual ? p.ldres : 0, p.aux ? p.ldaux : 0));
        if (repi >= 0 && G >= 8 && nwg >= G && p.alpha == 1.0f && p.c_dtype == ME_BF16 && (!(pre == 1 || pre == 2) || p.preact_dtype == ME_BF16) &&
            256 * ldmax * 2 < (1ll << 31))
            return launch3r_any(repi, pre, q, G, stream);
    }
    hipLaunchKernelGGL((gemm_g3_kernel<EPI>), dim3((unsigned)nwg), dim3(512), G3_LDS, stream, q);
    ME_CHECK_LAUNCH("me_gemm(g3)");
    return ME_OK;
}

}  // namespace

// TN: p.split_k slabs of p.ksteps_per_split K-tiles (of 64) each into p.C = [split_k][M][N] fp32
int launch_g3_tn(const GemmParams& p, hipStream_t stream) {
    static OncePerDevice once;
    if (once.need())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_g3tn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS);
    const int nwg = p.tiles_m * p.tiles_n * p.split_k;
    hipLaunchKernelGGL(gemm_g3tn_kernel, dim3((unsigned)nwg), dim3(512), G3_LDS, stream, p);
    ME_CHECK_LAUNCH("me_gemm(g3 tn)");
    return ME_OK;
}

bool g3_tn_supported(const GemmParams& p) {
    // 16-byte chunks of 8 columns; lane offsets and per-K-tile steps are 32-bit; the bounds check spans the whole matrix
    return p.M % 8 == 0 && p.N % 8 == 0 && p.lda % 8 == 0 && p.ldb % 8 == 0 && p.K * p.lda * 2 < (1ll << 31) &&
           p.K * p.ldb * 2 < (1ll << 31);
}

bool g3_supported(const GemmParams& p, int op) {
    if (op != ME_GEMM_NT) return false;
    if (p.K % (2 * G3_BK) != 0 || p.N % 8 != 0) return false;
    // per-tile buffer descriptors and lane offsets are 32-bit
    if (256 * p.lda * 2 >= (1ll << 31) || 256 * p.ldb * 2 >= (1ll << 31)) return false;
    return true;
}

// scratch of the persistent stream-K form (dev build): one fp32 partial tile per workgroup + the hand-over flags (+ 1
// error word)
size_t g3_workspace_bytes() {
    const size_t G = (size_t)g3_cus();
    return G * G3_SLAB_FLOATS * sizeof(float) + (G + 1) * sizeof(unsigned);
}

// one tile per workgroup; in the dev build ws != nullptr (>= g3_workspace_bytes()) selects the persistent stream-K form
int launch_g3(const GemmParams& p, int epi, void* ws, hipStream_t stream) {
    if (p.colscale) epi = 4;                       // (the specialised epilogues of this family carry no column scale)
    switch (epi) {
        case 0: return launch3e<0>(p, ws, stream);
        case 1: return launch3e<1>(p, ws, stream);
        case 2: return launch3e<2>(p, ws, stream);
        case 3: return launch3e<3>(p, ws, stream);
        default: return launch3e<4>(p, ws, stream);
    }
}
