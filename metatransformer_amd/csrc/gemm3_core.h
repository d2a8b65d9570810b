// gemm3_core.h -- device core of GEMM family "g3": bf16 NT (both operands reduction-contiguous), 256 x 256 tile, K-tile 64, 8 waves,
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
// (shared by gemm3.hip and, in the dev build only, gemm3_dev.hip: state, DMA issue, the 4-phase K-tile, the LDS-free epilogue)
#pragma once
#include "gemm_common.h"
#include <algorithm>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

namespace {

typedef __attribute__((address_space(3))) void lds_void3;

constexpr int G3_BM = 256, G3_BN = 256, G3_BK = 64;
// Cache policy (the `aux` operand of the buffer instructions: 1 = sc0, 2 = nt, 16 = sc1) of the resident kernel's five streams.
// Compile-time so that an A/B is one more build of the library (tools/r4_policy_builds.sh), never a branch around a
// memory operation: A = token rows (DMA), B = weight rows (DMA), C = output stores, R = row-operand loads, P = saved-tensor stores.
// Defaults (round 4, profiles/r04_policy_ab.txt): the outputs, the saved tensor and the row operands are NON-TEMPORAL -- a
// launch writes 77 .. 620 MB that nothing re-reads before it has left the 4 MB L2s anyway, and kept out of them it stops evicting
// the operand panels the other CUs of the XCD are about to re-read.  Kernel level (sustained loops, rotating buffers): qkv 186 ->
// 154 us, fc1 + GELU 284 -> 247 us, fc1 + saved gelu' 295 -> 274 us, proj 80 -> 76 us at the same 1.39 kW socket power; whole
// step, same box: train 33.2 -> 32.0 ms, forward 9.64 -> 9.51 ms.  sc1 (write-through) stores: no gain; nt on the token rows (A)
// gives most of it back (the 9 .. 12 column tiles of a tile row share them through the L2).
#ifndef G3_POL_A
#define G3_POL_A 0
#endif
#ifndef G3_POL_B
#define G3_POL_B 0
#endif
#ifndef G3_POL_C
#define G3_POL_C 2
#endif
#ifndef G3_POL_R
#define G3_POL_R 2
#endif
#ifndef G3_POL_C_RES
#define G3_POL_C_RES 0             // C policy of the residual epilogue: the 77 MB token stream is what the NEXT kernel reads straight away
#endif                             // (LayerNorm / statistics / the folded GEMM's A operand) -- kept cacheable: forward 9.59 -> 9.43 ms same box
#ifndef G3_POL_P
#define G3_POL_P 2
#endif
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
    int wrap_kt;                // (WRAP instantiations: GemmParams::a_wrap_kt) K-tiles of A from this one on re-read A from K-tile 0
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
// A 128 x 256 item (rows [mh, mh + 128) of A, a whole column tile of B) for the wave row `wr` of the calling wave.  Only the A-X
// half-tile carries data: a wave's DMA lanes address rows 128 wr' + (0..63) for A-X and 128 wr' + 64 + (0..63) for A-Y (wr' = the
// wave row the issuing wave belongs to: waves 0-3 fill the first 64 local rows of a half-tile, waves 4-7 the second 64), so the
// descriptor of wave row 1 starts 64 rows EARLIER (its A-X lanes then land on rows mh + 64 ..) and both descriptors end right
// behind the A-X rows: every A-Y lane is out of range -- zeros into a half-tile nobody reads, no memory traffic.
__device__ __forceinline__ G3Src g3_make_src_half(const GemmParams& p, int tm, int tn, int hsel, int wr) {
    const int64_t mh = (int64_t)tm * G3_BM + hsel * 128, n0 = (int64_t)tn * G3_BN;
    int64_t ra = p.M - mh, rb = p.N - n0;
    ra = ra < 128 ? ra : 128;                       // rows of this item that exist
    ra = ra < 0 ? 0 : ra;
    rb = rb < G3_BN ? rb : G3_BN;
    // wave row 0 sees rows [mh, mh + min(64, ra)); wave row 1 sees relative rows [128, 128 + min(64, ra - 64)) from mh - 64
    const int64_t mine = wr ? ra - 64 : ra;
    const int64_t nrow = mine < 0 ? 0 : (mine < 64 ? mine : 64);
    const int64_t first = wr ? 128 : 0;
    G3Src s;
    const char* base = reinterpret_cast<const char*>(p.A) + (mh - 64 * wr) * p.lda * 2;
    s.a = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, nrow > 0 ? (int)((first + nrow - 1) * p.lda * 2 + p.K * 2) : 0, 0x00020000);
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
// WRAP (one-tile NT kernel only; me_gemm_desc.a_wrap_k): the A operand is an fp32 matrix held as TWO bf16 planes [hi | lo] while the reduction
// walks three segments (hi, lo, hi) against weights [hi | hi | lo] -- from K-tile s.wrap_kt on, A's offset starts over at K-tile 0 (a scalar
// select per issue; the third plane of the ME_BF16X3 layout is not stored at all)
template <int J, bool WRAP = false> __device__ __forceinline__ void g3_issue(const G3State& s, const G3Src& src, int buf, int kt) {
    char* dst = s.smem + buf * G3_BUF + J * G3_HALF + s.wave * 2048;
    const __amdgpu_buffer_rsrc_t r = (J & 1) ? src.a : src.b;
    if (WRAP && (J & 1)) kt = kt >= s.wrap_kt ? kt - s.wrap_kt : kt;
    const int koff = kt * ((J & 1) ? s.kstep_a : s.kstep_b);
    constexpr int pol = (J & 1) ? G3_POL_A : G3_POL_B;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void3*)dst, 16, (int)s.src[J][0], koff, 0, pol);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void3*)(dst + 1024), 16, (int)s.src[J][1], koff, 0, pol);
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
// HALF (resident NT kernel, a 128 x 256 item: the rows of A-X only).  The item's two quadrants (A-X x B-X, A-X x B-Y) ARE phases 0
// and 1; its K-tile has only these two phases (four barriers: both wave rows of a workgroup run the same item, so their one-barrier
// stagger is untouched) and a three-half-tile stream with its own schedule -- keeping phases 2 / 3 as empty shells measured 1 680
// clocks per K-tile against 2 270 for a whole tile (each shell waits out the partner row's MFMA block):
//   phase 0 of K-tile t:  reads B-X(t), A-X(t);  issues B-Y(t+1) -> the other buffer   [skipped when SEAM: it went out ahead of the epilogue]
//   phase 1 of K-tile t:  reads B-Y(t);          issues B-X(t+2), A-X(t+2) -> this buffer
//   both: every fragment read of the phase RETIRES before the phase's first barrier (lgkmcnt(0) ahead of it), then
//         s_waitcnt vmcnt(6 [+ SLACK, see below]) -- phase 1: B-X / A-X of t+1 have landed (behind them in the queue: B-Y(t+1) and the four
//         instructions just issued); phase 0: B-Y(t) has landed (behind it: B-X / A-X(t+1) and the two just issued).
//   WAR: a slot is overwritten one phase after its last read at the earliest, and that read retired before a barrier which the
//        issuing wave row -- and, one barrier later, the other row -- has passed by then.
//   RAW: every wave waits for its own share before the phase's first barrier; the data is read behind the phase's second one.
// Seams: the K-tile ahead of an item's first and second one were issued by the previous item's last K-tiles / epilogue in either
// item's pattern (a whole tile's tail issues A-Y too: out of range for a 128-row item, see g3_make_src_half); the first two K-tiles of an
// item wait with SLACK = the epilogue's operation count more (they sit between those half-tiles and the K-tile's own in the queue).
template <int BUF, int P, bool TN = false, int SEAM = 0, bool HALF = false, int SLACK = SEAM, bool WRAP = false>
__device__ __forceinline__ void g3_phase(G3State& s, const G3Src& s0, int k0, const G3Src& s1, int k1, bool cs_on = false) {
    if (HALF) {
        static_assert(!HALF || (!TN && P < 2), "128-row items: NT, phases 0 and 1");
        if (P == 0) {
            s.bx[0][0] = g3_frag<0 * G3_HALF + 0 * 2048>(s.rb[BUF][0]); s.bx[0][1] = g3_frag<0 * G3_HALF + 0 * 2048>(s.rb[BUF][1]);
            s.bx[1][0] = g3_frag<0 * G3_HALF + 1 * 2048>(s.rb[BUF][0]); s.bx[1][1] = g3_frag<0 * G3_HALF + 1 * 2048>(s.rb[BUF][1]);
            s.ax[0][0] = g3_frag<1 * G3_HALF + 0 * 2048>(s.ra[BUF][0]); s.ax[0][1] = g3_frag<1 * G3_HALF + 0 * 2048>(s.ra[BUF][1]);
            s.ax[1][0] = g3_frag<1 * G3_HALF + 1 * 2048>(s.ra[BUF][0]); s.ax[1][1] = g3_frag<1 * G3_HALF + 1 * 2048>(s.ra[BUF][1]);
            s.ax[2][0] = g3_frag<1 * G3_HALF + 2 * 2048>(s.ra[BUF][0]); s.ax[2][1] = g3_frag<1 * G3_HALF + 2 * 2048>(s.ra[BUF][1]);
            s.ax[3][0] = g3_frag<1 * G3_HALF + 3 * 2048>(s.ra[BUF][0]); s.ax[3][1] = g3_frag<1 * G3_HALF + 3 * 2048>(s.ra[BUF][1]);
        } else {
            s.by[0][0] = g3_frag<2 * G3_HALF + 0 * 2048>(s.rb[BUF][0]); s.by[0][1] = g3_frag<2 * G3_HALF + 0 * 2048>(s.rb[BUF][1]);
            s.by[1][0] = g3_frag<2 * G3_HALF + 1 * 2048>(s.rb[BUF][0]); s.by[1][1] = g3_frag<2 * G3_HALF + 1 * 2048>(s.rb[BUF][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (P == 0 && SEAM == 0) g3_issue<2>(s, s0, BUF ^ 1, k0);
        if (P == 1) { g3_issue<0>(s, s1, BUF, k1); g3_issue<1>(s, s1, BUF, k1); }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // (the slack covers half-tiles that were issued AHEAD of the epilogue: everything the item's first K-tile waits for, and the
        // B-Y its second K-tile's phase 0 waits for -- not the B-X / A-X that phase 1 of the second K-tile waits for: those went
        // out in phase 1 of the first K-tile, behind the epilogue's stores)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + ((P == 0 || SEAM) ? SLACK : 0)) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if (P == 0) {
            G3_MMA(0, 0, s.ax, s.bx) G3_MMA(0, 1, s.ax, s.bx) G3_MMA(1, 0, s.ax, s.bx) G3_MMA(1, 1, s.ax, s.bx)
            G3_MMA(2, 0, s.ax, s.bx) G3_MMA(2, 1, s.ax, s.bx) G3_MMA(3, 0, s.ax, s.bx) G3_MMA(3, 1, s.ax, s.bx)
        } else {
            G3_MMA(0, 2, s.ax, s.by) G3_MMA(0, 3, s.ax, s.by) G3_MMA(1, 2, s.ax, s.by) G3_MMA(1, 3, s.ax, s.by)
            G3_MMA(2, 2, s.ax, s.by) G3_MMA(2, 3, s.ax, s.by) G3_MMA(3, 2, s.ax, s.by) G3_MMA(3, 3, s.ax, s.by)
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        return;
    }
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
    if (P == 0 && SEAM == 0) g3_issue<3, WRAP>(s, s0, BUF ^ 1, k0);
    if (P == 1) g3_issue<0, WRAP>(s, s1, BUF, k1);
    if (P == 2) g3_issue<1, WRAP>(s, s1, BUF, k1);
    if (P == 3) g3_issue<2, WRAP>(s, s1, BUF, k1);
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

template <int BUF, bool TN = false, int SEAM = 0, bool HALF = false, int SLACK = SEAM, bool WRAP = false>
__device__ __forceinline__ void g3_ktile(G3State& s, const G3Src& s0, int k0, const G3Src& s1, int k1, bool cs_on = false) {
    g3_phase<BUF, 0, TN, SEAM, HALF, SLACK, WRAP>(s, s0, k0, s1, k1, cs_on);
    g3_phase<BUF, 1, TN, SEAM, HALF, SLACK, WRAP>(s, s0, k0, s1, k1, cs_on);
    if (HALF) return;
    g3_phase<BUF, 2, TN, SEAM, false, SLACK, WRAP>(s, s0, k0, s1, k1, cs_on);
    g3_phase<BUF, 3, TN, SEAM, false, SLACK, WRAP>(s, s0, k0, s1, k1, cs_on);
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
// EPI: 0 bias, 1 + GELU (+ pre-activation save), 2 + residual row operand, 3 * gelu'(aux row operand), 6 * aux row operand,
// 7 + GELU with gelu'(h) saved as the pre-activation, 4 generic
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
    // per-column scale (layer-scale gamma of the detection / Video Blocks: vit.py:313-316), carried by the residual forms 2 and 8 -- the
    // launches that have one: x + gamma * branch(x).  Ones when absent (a wave-uniform choice ahead of the straight-line part).
    f32x4 cscale[2][2];
    constexpr bool CS = EPI == 2 || EPI == 8;
    if (CS) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            cscale[q][0] = cscale[q][1] = f32x4{1.f, 1.f, 1.f, 1.f};
            if (p.colscale) {
                cscale[q][0] = *reinterpret_cast<const f32x4*>(p.colscale + (n_ok[q] ? n[q] : 0));
                cscale[q][1] = *reinterpret_cast<const f32x4*>(p.colscale + (n_ok[q] ? n[q] : 0) + 4);
            }
        }
        asm volatile("" ::"v"(cscale[0][0]), "v"(cscale[0][1]), "v"(cscale[1][0]), "v"(cscale[1][1]));
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
    if constexpr (EPI == 8 || EPI == 9 || EPI == 10) {
        //   EPI 8   bias + fp32 row operand (the residual stream) -> fp32: proj / fc2 on an fp32 token stream -- the reference's autocast
        //           recipes (fp32 tokens, bf16 compute: Video/engine_for_finetuning.py:92-99) and the ME_BF16X3 Blocks
        // ---- fp32 result as the three bf16 planes of ME_BF16X3 ([hi | lo | hi], ldc >= 3 N) -- the MLP of an ME_BF16X3 Block:
        //   EPI 9   bias -> (p.preact, fp32: gelu'(h) saved for backward) -> GELU (erf form: fp32 accuracy) -> planes     (fc1)
        //   EPI 10  acc * fp32 row operand (the saved gelu') -> planes                                                    (fc2 dgrad)
        // Straight-line, as the other specialised forms: clamped unconditional loads, only the stores are guarded -- the generic
        // epilogue these launches used to take waits for every bias / factor load inside its `if (ok)` (draining the stores before
        // it each time): 869 us per fc1 launch at [50 432, 3 x 768] x 3 072 against 640 us for the same GEMM with a bias epilogue.
        const float* fac = reinterpret_cast<const float*>(EPI == 8 ? p.residual : p.aux);
        const int64_t ldf = EPI == 8 ? p.ldres : p.ldaux;
        uint16_t* Cp = reinterpret_cast<uint16_t*>(p.C);
        f32x4 fa[2][2][2];                                       // [slab parity][q][half]: factor rows one slab ahead
        auto fetchf = [&](const int mt, f32x4 (&dst)[2][2]) {
            int64_t m = mrow + mt * 16;
            m = m < p.M ? m : p.M - 1;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float* fp = fac + m * ldf + (n_ok[q] ? n[q] : 0);
                dst[q][0] = *reinterpret_cast<const f32x4*>(fp);
                dst[q][1] = *reinterpret_cast<const f32x4*>(fp + 4);
            }
        };
        if (EPI != 9) fetchf(0, fa[0]);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            if (EPI != 9 && mt + 1 < 8) fetchf(mt + 1, fa[(mt + 1) & 1]);
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
                v0 = v0 * alpha4 + bias[q][0];
                v1 = v1 * alpha4 + bias[q][1];
                if (EPI == 9) {
                    // Phi and the Gaussian ONCE for GELU and its derivative (round 6: as two calls behind / in front of the branch the
                    // compiler formed them twice -- 256 v_exp_f32 + 256 v_rcp_f32 per thread and tile instead of 128 + 128)
                    f32x4 d0, d1;
                    gelu_erf_pair4(v0, v0, d0);
                    gelu_erf_pair4(v1, v1, d1);
                    if (p.preact && ok) {
                        float* pp = reinterpret_cast<float*>(p.preact) + m * p.ldpre + n[q];
                        *reinterpret_cast<f32x4*>(pp) = d0;
                        *reinterpret_cast<f32x4*>(pp + 4) = d1;
                    }
                } else if (EPI == 10) {
                    v0 *= fa[mt & 1][q][0];
                    v1 *= fa[mt & 1][q][1];
                } else {
                    v0 = v0 * cscale[q][0] + fa[mt & 1][q][0];
                    v1 = v1 * cscale[q][1] + fa[mt & 1][q][1];
                    if (ok) {
                        float* row = reinterpret_cast<float*>(p.C) + m * p.ldc + n[q];
                        *reinterpret_cast<f32x4*>(row) = v0;
                        *reinterpret_cast<f32x4*>(row + 4) = v1;
                    }
                    continue;
                }
                // eight consecutive columns -> one 16-byte store per plane
                bf16x8 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hi[e] = (bf16_t)v0[e];     lo[e] = (bf16_t)(v0[e] - (float)hi[e]);
                    hi[4 + e] = (bf16_t)v1[e]; lo[4 + e] = (bf16_t)(v1[e] - (float)hi[4 + e]);
                }
                if (ok) {
                    uint16_t* row = Cp + m * p.ldc + n[q];
                    *reinterpret_cast<bf16x8*>(row) = hi;
                    *reinterpret_cast<bf16x8*>(row + p.N) = lo;
                    if (p.c_dtype == ME_BF16X3) *reinterpret_cast<bf16x8*>(row + 2 * p.N) = hi;      // (ME_BF16X2: [hi | lo] only -- a third fewer bytes)
                }
            }
        }
        return;
    }
    constexpr int AHEAD = 6;
    constexpr bool ROWOP = EPI == 2 || EPI == 3 || EPI == 6;
    u32x4 rowop[8][2];
    if (ROWOP) {
#pragma unroll
        for (int mt = 0; mt < AHEAD; ++mt) fetch(mt, rowop[mt]);
    }
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        f32x4 ro[2][2];
        if (ROWOP) {
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
                    ME_NT_STORE(ME_POL_SLAB, v0, reinterpret_cast<f32x4*>(d));
                    ME_NT_STORE(ME_POL_SLAB, v1, reinterpret_cast<f32x4*>(d + 4));
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
            if (EPI == 7) {                               // saved: gelu'(h) (ME_GEMM_SAVE_GELU_GRAD), the erf pair as the resident kernel
                f32x4 d0, d1;
                gelu_erf_pair4(v0, v0, d0);
                gelu_erf_pair4(v1, v1, d1);
                if (ok) store8_from_f32(p.preact, p.preact_dtype, m * p.ldpre + n[q], d0, d1);
            }
            const f32x4 qa = ro[q][0], qb = ro[q][1];
            if (EPI == 3) {
                v0 *= gelu_grad_for4(qa, p.c_dtype);
                v1 *= gelu_grad_for4(qb, p.c_dtype);
            }
            if (EPI == 6) { v0 *= qa; v1 *= qb; }
            if (EPI == 2) { v0 = v0 * cscale[q][0] + qa; v1 = v1 * cscale[q][1] + qb; }
            if (kMeDev && (p.debug & 4)) {             // dev: epilogue arithmetic without the stores
                asm volatile("" ::"v"(v0), "v"(v1));
                continue;
            }
            if (ok) store8_from_f32(p.C, p.c_dtype, m * p.ldc + n[q], v0, v1);
        }
    }
}

// CUs of the current device (the resident / persistent forms launch one workgroup per CU)
inline int g3_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    }
    return n;
}

}  // namespace
