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
// The shared device code -- state, DMA issue, the 4-phase K-tile, the LDS-free epilogue -- lives in gemm3_core.h; this file holds the
// kernels the library ships (one tile per workgroup, resident, wgrad) and their launchers.  Dev-build hooks are `if (kMeDev && ...)`
// (a compile-time false in libmetaenc.so); the dev-only stream-K kernel is in gemm3_dev.hip.
#include "gemm3_core.h"

#ifndef G3_ROWOP_AHEAD_HALF
#define G3_ROWOP_AHEAD_HALF 4                  // ... in a 128-row item's epilogue (two: proj 80.5 -> 84.2 us, fc2 203 -> 206 us)
#endif
#ifndef G3_ROWOP_AHEAD_STATS
#define G3_ROWOP_AHEAD_STATS 4                 // ... in the statistics / gelu'(row operand) epilogues
#endif
#ifndef G3_TRAIN_GELU_POLY
#define G3_TRAIN_GELU_POLY 0                   // (A/B arm: the two bf16-mode polynomials instead of the shared-exponential erf pair where gelu AND gelu' are stored)
#endif
#ifndef G3_COLGROUPS
#define G3_COLGROUPS 0                         // (A/B arm, OFF: a column-grouped tile walk of the resident kernel -- measured +0.8 % on the forward and +0.2 % on the
                                               //  train step, profiles/r06_colgroups_ab.txt: the weight re-reads it removes are Infinity-Cache hits, the token re-reads
                                               //  it adds are not all.  Compiled out when 0: its scalars would live across the item loop)
#endif
#ifndef G3_ROWOP_AHEAD
#define G3_ROWOP_AHEAD 4                       // row-operand slabs in flight ahead of their use in a whole tile's epilogue (6 until round 4: proj 84.6 -> 80.6 us,
                                               // fc2 205.7 -> 201.7 us sustained, gpurun_out r4s / profiles/r04_rowop_ahead.txt: fewer spills at the seam)
#endif
namespace {

// ---- one tile per workgroup
template <int EPI, bool WRAP = false>
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
    if (kMeDev && (p.debug >> 4) && bid < 256) { // dev: stagger the first round (output bursts of the CUs spread out)
        const int n = (slot & 7) * (p.debug >> 4);
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(4);
    }
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
    s.wrap_kt = WRAP ? p.a_wrap_kt : 0;
    g3_issue<0, WRAP>(s, src, 0, kt0); g3_issue<1, WRAP>(s, src, 0, kt0); g3_issue<2, WRAP>(s, src, 0, kt0); g3_issue<3, WRAP>(s, src, 0, kt0);
    g3_issue<0, WRAP>(s, src, 1, kt0 + 1); g3_issue<1, WRAP>(s, src, 1, kt0 + 1); g3_issue<2, WRAP>(s, src, 1, kt0 + 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();

    // past the end of the K-range the source is a null descriptor (see g3_phase)
    const G3Src null = g3_null_src(p);
    for (int kt = kt0; kt < kt1 - 2; kt += 2) {
        g3_ktile<0, false, 0, false, 0, WRAP>(s, src, kt + 1, src, kt + 2);
        g3_ktile<1, false, 0, false, 0, WRAP>(s, src, kt + 2, src, kt + 3);
    }
    g3_ktile<0, false, 0, false, 0, WRAP>(s, src, kt1 - 1, null, 0);
    g3_ktile<1, false, 0, false, 0, WRAP>(s, null, 0, null, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wr == 0) __builtin_amdgcn_s_barrier();
    if (kMeDev && (p.debug & 1)) {               // dev: K-loop only
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) keep += s.acc[i][j][0] + s.acc[i][j][1] + s.acc[i][j][2] + s.acc[i][j][3];
        if (keep == 1.2345e-30f) reinterpret_cast<float*>(p.C)[0] = keep;
        return;
    }
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

// ---- PRE 5: the folded LayerNorm's row pairs from the 256-column partials of the launch in front (see g3_epilogue_r).  Thread
// t = 64 wave + lane of the workgroup owns rows 2 (t >> 2), + 1 of an item (m0 = its first row, rows = how many exist) and part
// t & 3 (< row_nparts <= 4): ONE 16-byte load per thread -- 8 wave-level memory instructions per tile (a first version with 64 of them,
// one 8-byte load per 64-column partial, cost the CU's memory path ~2 k clocks per tile in front of the stores).
struct G3Parts { f32x4 v; };
__device__ __forceinline__ G3Parts g3r_parts_load(const GemmParams& p, int64_t m0, int rows, int wave) {
    G3Parts o;
    const int lane = g3_lane_now();
    const int t = wave * 64 + lane, rp = t >> 2, j = t & 3;
    // [part][M] pairs seen from row m0: the last part ends `rows` rows in (rows past the item / the matrix: out of range -> zeros;
    // an odd last row: the second pair of its 16 bytes is out of range -> zeros, never read back)
    const __amdgpu_buffer_rsrc_t pars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.row_affine) + (rows > 0 ? m0 * 2 : 0), 0,
                                                                           rows > 0 ? (int)((((int64_t)p.row_nparts - 1) * p.M + rows) * 8) : 0, 0x00020000);
    const uint32_t off = (j < p.row_nparts && 2 * rp < rows) ? (uint32_t)(((int64_t)j * p.M + 2 * rp) * 8) : 0x80000000u;
    o.v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pars, (int)off, 0, 0));
    return o;
}
// ... folded into (rstd, -rstd * mean) of LayerNorm(256 row_nparts, row_eps) for both rows and written to LDS `lds_pairs`[row]: the
// arithmetic of row_stats_combine_kernel (mean of the part means, M2 = sum M2_j + 256 sum (mean_j - mean)^2) with the sums over a
// quad of lanes (two DPP swaps; the order (a + b) + (c + d) is the same in all four lanes: they hold the same bits)
__device__ __forceinline__ void g3r_parts_fold(const GemmParams& p, const G3Parts& q, int wave, uint32_t lds_pairs) {
    const int lane = g3_lane_now();
    const int t = wave * 64 + lane, rp = t >> 2, j = t & 3;
    auto quad = [](float x) {
        x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0xB1, 0xf, 0xf, true));      // quad_perm [1, 0, 3, 2]
        x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x4E, 0xf, 0xf, true));      // quad_perm [2, 3, 0, 1]
        return x;
    };
    const float w = j < p.row_nparts ? 1.0f : 0.0f;
    const float inv_np = __builtin_amdgcn_rcpf((float)p.row_nparts);
    const float inv_c = inv_np * (1.0f / 256.0f);
    u32x4 out;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const float mean = quad(q.v[2 * r]) * inv_np;                           // (parts past row_nparts / rows past the item are zeros)
        const float m2 = quad(q.v[2 * r + 1]);
        const float d = (q.v[2 * r] - mean) * w;
        const float dev = quad(d * d);
        const float rstd = __builtin_amdgcn_rsqf((m2 + 256.0f * dev) * inv_c + p.row_eps);
        out[2 * r] = __float_as_uint(rstd);
        out[2 * r + 1] = __float_as_uint(-rstd * mean);
    }
    // the four lanes of a row pair hold the same 16 bytes and write them to the same place: no exec-mask games
    asm volatile("ds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(lds_pairs + (uint32_t)rp * 16), "v"(out) : "memory");
}

// EPI: 0 bias, 1 GELU (PRE: 0 nothing saved, 1 pre-activation saved, 2 gelu'(pre-activation) saved), 2 + residual row
// operand, 3 * gelu'(row operand), 6 * row operand.
// PRE 3 (EPI 0 / 1): a LayerNorm folded into this Linear (GemmParams::row_affine / col_shift): the accumulators start at zero
// and the epilogue applies v = rstd_m * acc + (-rstd_m mean_m) * s[n] + c[n] in the accumulator layout (one row per lane and
// 16-row slab, four consecutive columns per register quad), ahead of the activation; nothing is saved.
// PRE 5 (EPI 0 / 1): the same folded LayerNorm, its row pairs formed HERE from the 64-column partials (mean_i, M2_i) the residual
// launch in front of this one left behind (PRE 4; GemmParams::row_affine = [row_nparts][M] pairs over 256 columns each, row_nparts <= 4) -- no
// row_stats_combine launch between the two GEMMs (24 per Base forward).  Software-pipelined over the items of a workgroup so that
// neither the memory latency nor the arithmetic sits in front of an epilogue: the epilogue of item i (a) reads item i's 256 pairs
// from 2 KiB of LDS BESIDE the operand buffers (buffer i & 1), (b) issues the load of item i + 1's partials at its top -- thread t
// of the workgroup takes rows 2 (t >> 2), + 1 of that item and part t & 3: one 16-byte load -- and (c) folds them BEHIND its last store, where the
// CU's memory path is busy draining the stores and the first K-tile of item i + 1 would only wait for it (measured: K-tile 1 takes
// 4.3 .. 5.1 k clocks against 2.4 k in steady state): the arithmetic of row_stats_combine_kernel (mean of the part means, M2 = sum
// M2_j + 256 sum (mean_j - mean)^2; the parts of a row meet through two DPP swaps), pairs into buffer (i + 1) & 1.  A workgroup's
// first item is folded in the prologue.  A first version that loaded, folded and exchanged the pairs at the TOP of each epilogue
// (one workgroup barrier) cost 1 us per tile, and one 8-byte load per 64-column partial (64 wave-level loads per tile) 2 k clocks of
// the CU's memory path in front of the stores -- each as much as the launches it removed (profiles/r06_row_parts_ab.txt).
// All LDS traffic is inline asm: a compiler-visible LDS access would be guarded with vmcnt(0) while the next tile's DMA is in flight.
// PRE 4 (EPI 2): the per-row statistics of the OUTPUT rows on the side (GemmParams::row_stats) -- the LayerNorm that reads this
// residual stream next then needs no pass of its own over it (me_row_stats_combine folds the partials).  After the row re-deal
// eight lanes hold one row's 64 columns of this wave: per half slab every lane forms (S, Q) = sum (v - P), sum (v - P)^2 of its
// eight values against a pivot P = the row's first value in this wave column (shifted sums: no cancellation however far the row
// mean is from zero; P reaches the eight lanes through one DPP move and two lane-row swaps), and the sixteen (half slab) pairs of
// a wave are summed over the eight lanes as a REDUCE-SCATTER -- DPP row_ror:8, v_permlane16_swap, v_permlane32_swap, each step
// halving the number of live values -- so that every lane ends up with two finished rows: (mean, M2) over n = 64 columns
// (a 128-row item: four slabs, one row per lane).  Round 6: these per-wave-column pairs no longer go to memory -- the four wave columns
// of a row meet in 8 KiB of LDS behind the last store and ONE pair per row over the tile's 256 columns is stored, [N / 256][M] (see
// the block behind the slab loop).  The statistics are those of the values AS STORED (rounded to
// bf16 and converted back: eight more operations per half slab) -- for a row whose mean dwarfs its spread the rounding IS the
// spread, and the reference's LayerNorm sees the rounded stream too.
// HALF: a 128 x 256 item (g3_make_src_half): accumulator slabs 0..3 only, this wave row's rows are m0 + 64 wr + ..; the
// epilogue is padded with stores no descriptor admits up to the whole tile's operation count, so that the counted waits behind it
// (g3_phase<.., SEAM>, the ticket wait) are the same for both item kinds.
template <int EPI, int PRE, bool HALF = false>
__device__ __forceinline__ void g3_epilogue_r(const GemmParams& p, G3State& s, int64_t m0, int tn, int lane, const G3Src& nxt, int nk,
                                              const __amdgpu_buffer_rsrc_t brs, int ntn, bool next_zero, unsigned* ctr, int nx,
                                              uint32_t lds_tick, int64_t next_m0 = 0, int next_rows = 0, int par = 0) {
    // PRE 6 (EPI 1 / EPI 6): gelu'(h) travels as EIGHT BITS (ME_GG8: q = rint((g + 0.13) x 255 / 1.26), g in [-0.129, 1.129]) -- EPI 1 saves it,
    // EPI 6 multiplies by it.  Half the bytes of the bf16 factor (310 MB less written and 310 MB less read per layer at config 2: at ~120 pJ
    // per HBM byte the largest single item a train step can shed, profiles/r06_energy_probe.txt); the same number of memory operations,
    // each half as wide (8 bytes per lane, 8 consecutive lanes = the wave's 64-byte row segment), so every counted wait stays what it is.
    // Step 0.0049: |error| <= 0.0025 absolute, about what bf16 leaves at g ~ 1 and more than bf16 leaves near 0.
    constexpr bool G8 = PRE == 6, G8S = G8 && EPI == 1, G8L = G8 && EPI == 6;
    static_assert(!G8 || EPI == 1 || EPI == 6, "8-bit gelu': the fc1 forward that saves it, the fc2 dgrad that reads it");
    constexpr bool SAVE = PRE == 1 || PRE == 2 || G8S, LNF = PRE == 3 || PRE == 5, LNP = PRE == 5, STATS = PRE == 4;
    static_assert(!STATS || EPI == 2, "row statistics: the residual epilogue");
    constexpr int NMT = HALF ? 4 : 8, WROWS = HALF ? 64 : 128;      // 16-row slabs per wave, rows per wave row
    // (claimed schedule: wave 0 draws the ticket for the item after next FIRST, ahead of every store of this epilogue)
    unsigned drawn = 0;
    if (ctr && s.wave == 0) drawn = g3r_draw(ctr);
    (void)lane;
    lane = g3_lane_now();
    const int wr = s.wave >> 2, wc = s.wave & 3;
    const int r = lane & 15, g = lane >> 4;
    const int64_t n0 = (int64_t)tn * G3_BN;
    int64_t rows = p.M - m0, cols = p.N - n0;
    rows = rows < 2 * WROWS ? rows : 2 * WROWS;
    rows = rows < 1 ? 1 : rows;                 // (an item past the last row: every access below is then out of range)
    cols = cols < G3_BN ? cols : G3_BN;
    const bool item_ok = p.M > m0;
    auto tile_rsrc = [&](const void* base, int64_t ld, int esz = 2) {
        if (!item_ok) return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0, 0x00020000);
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base)) + (m0 * ld + n0) * esz, 0,
                                                 (int)(((rows - 1) * ld + cols) * esz), 0x00020000);
    };
    // dev (debug bit 4): the stores go nowhere (zero-record descriptor), everything else unchanged
    // (debug bit 2 with bit 4: only the first workgroup of every XCD keeps its stores)
    const bool drop = kMeDev && (p.debug & 4) && !((p.debug & 2) && (blockIdx.x >> 3) == 0);
    const __amdgpu_buffer_rsrc_t crs = drop ? __builtin_amdgcn_make_buffer_rsrc(p.C, 0, 0, 0x00020000) : tile_rsrc(p.C, p.ldc);
    const __amdgpu_buffer_rsrc_t prs = SAVE ? tile_rsrc(p.preact, p.ldpre, G8S ? 1 : 2) : crs;
    const __amdgpu_buffer_rsrc_t rrs = EPI == 2 ? tile_rsrc(p.residual, p.ldres) : (EPI == 3 || EPI == 6) ? tile_rsrc(p.aux, p.ldaux, G8L ? 1 : 2) : crs;
    const int rop_ld = (int)(EPI == 2 ? p.ldres : p.ldaux);
    // memory side: lane t = row t >> 3 of an 8-row half slab, 16-byte chunk t & 7 of the wave's 128-byte row segment.  A
    // chunk past the column edge gets an offset no descriptor admits; rows past the row edge fall behind the descriptor's end.
    const int colb = wc * 128 + (lane & 7) * 16;
    const bool ok = (colb >> 1) + 8 <= (int)cols;
    const int row = wr * WROWS + (lane >> 3);
    const uint32_t coff = ok ? (uint32_t)(row * (int)p.ldc * 2 + colb) : 0x80000000u;
    const uint32_t poff = ok && SAVE ? (G8S ? (uint32_t)(row * (int)p.ldpre + (colb >> 1)) : (uint32_t)(row * (int)p.ldpre * 2 + colb)) : 0x80000000u;
    const uint32_t roff = ok ? (G8L ? (uint32_t)(row * rop_ld + (colb >> 1)) : (uint32_t)(row * rop_ld * 2 + colb)) : 0x80000000u;
    const int cstep = (int)p.ldc * 16, pstep = (int)p.ldpre * (G8S ? 8 : 16), rstep = rop_ld * (G8L ? 8 : 16);       // 8 rows, bytes
    // register side (after g3r_rows8): lane (r, g) = row r & 7, chunk 4 (r >> 3) + 2 (g & 1) + (g >> 1).  to_mem: the lane
    // that holds memory lane t's chunk; to_reg: the memory lane that holds this lane's chunk (x 4: bpermute byte addresses)
    const int to_mem = 4 * ((lane >> 3) + 8 * ((lane >> 2) & 1) + 16 * (((lane >> 1) & 1) | ((lane & 1) << 1)));
    const int to_reg = 4 * (8 * (r & 7) + 4 * (r >> 3) + 2 * (g & 1) + (g >> 1));
    auto fetch = [&](const int mt, u32x4 (&raw)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (G8L) {
                const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rrs, (int)(roff + (2 * mt + h) * rstep), 0, G3_POL_R);
                raw[h] = u32x4{t[0], t[1], 0u, 0u};
            } else {
                raw[h] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)(roff + (2 * mt + h) * rstep), 0, G3_POL_R);
            }
        }
    };
    // 8-bit gelu' (ME_GG8): eight bytes <-> eight factors
    auto unpack8 = [](const u32x4& rw, f32x4& a, f32x4& b) {
        const f32x4 qa = {(float)(rw[0] & 0xffu), (float)((rw[0] >> 8) & 0xffu), (float)((rw[0] >> 16) & 0xffu), (float)(rw[0] >> 24)};
        const f32x4 qb = {(float)(rw[1] & 0xffu), (float)((rw[1] >> 8) & 0xffu), (float)((rw[1] >> 16) & 0xffu), (float)(rw[1] >> 24)};
        a = qa * ME_GG8_STEP + ME_GG8_LO;
        b = qb * ME_GG8_STEP + ME_GG8_LO;
    };
    auto pack8 = [](const f32x4& a, const f32x4& b) {
        // v_cvt_pk_u8_f32 rounds to nearest even and saturates to 0 .. 255 (NaN -> 0): tools/cvt_pk_u8_probe.hip -- one fma and one pack per factor
        unsigned w0 = 0u, w1 = 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            w0 = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(a[e], 1.0f / ME_GG8_STEP, -ME_GG8_LO / ME_GG8_STEP), e, w0);
            w1 = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(b[e], 1.0f / ME_GG8_STEP, -ME_GG8_LO / ME_GG8_STEP), e, w1);
        }
        return u32x4{w0, w1, 0u, 0u};
    };
    auto lanes2 = [](const u32x4& v, int addr) {        // g3r_lanes on the two live words
        u32x4 o = {0u, 0u, 0u, 0u};
        o[0] = (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)v[0]);
        o[1] = (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)v[1]);
        return o;
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
    // (STATS) destination of this TILE's partials: [part = tn][M] pairs over its 256 columns; a part past the last whole column tile, or an
    // item past the last row, gets a descriptor that admits nothing (the operation count stays what it is).  The four wave columns'
    // 64-column pairs meet in 8 KiB of LDS beside the operand buffers ([row][wave column] pairs, behind the ticket word)
    float st_a[2][3], st_b[2][3];
    __amdgpu_buffer_rsrc_t srs = crs;
    const uint32_t lds_stats = lds_tick + 64;
    if (STATS) {
        const bool part_ok = item_ok && ((int64_t)tn + 1) * 256 <= p.N;
        srs = __builtin_amdgcn_make_buffer_rsrc(p.row_stats + (part_ok ? ((int64_t)tn * p.M + m0) * 2 : 0), 0, part_ok ? (int)(rows * 8) : 0, 0x00020000);
    }
    constexpr int AHEAD = HALF ? G3_ROWOP_AHEAD_HALF : (EPI == 3 || STATS) ? G3_ROWOP_AHEAD_STATS : G3_ROWOP_AHEAD;      // row-operand slabs in flight ahead of their use (more spills: into the K-loop for gelu', onto the ticket register otherwise)
    u32x4 rowop[8][2];
    if (EPI == 2 || EPI == 3 || EPI == 6) {
#pragma unroll
        for (int mt = 0; mt < AHEAD; ++mt) fetch(mt, rowop[mt]);
    }
    // folded LayerNorm: this tile's per-row pairs (rows wr*128 + 16 mt + r) and per-column s / c, all ahead of the DMA below
    f32x2 lnf_row[8];
    G3Bias lnf_s, lnf_c;
    G3Parts lnp;                                // (LNP) this thread's share of the NEXT item's partials (folded behind the stores)
    u32x2 lnq[8];                               // (LNP) this item's pairs on their way in from LDS
    if (LNF) {
        const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.col_shift), 0, (int)(p.N * 4), 0x00020000);
        if (LNP) {
            const uint32_t pa = lds_tick + 64 + (uint32_t)par * 2048 + (uint32_t)(wr * WROWS + r) * 8;
#define G3R_PAIR(i) if ((i) < NMT) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(lnq[i]) : "v"(pa), "i"((i) * 128) : "memory");
            G3R_PAIR(0) G3R_PAIR(1) G3R_PAIR(2) G3R_PAIR(3) G3R_PAIR(4) G3R_PAIR(5) G3R_PAIR(6) G3R_PAIR(7)
#undef G3R_PAIR
            lnp = g3r_parts_load(p, next_m0, next_rows, s.wave);
        } else {
            const __amdgpu_buffer_rsrc_t rars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.row_affine) + m0 * 2, 0, item_ok ? (int)(rows * 8) : 0, 0x00020000);
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)    // (rows past the edge: out of range -> zeros; their outputs are never stored)
                lnf_row[mt] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rars, (wr * WROWS + mt * 16 + r) * 8, 0, 0));
        }
        lnf_s = g3r_bias(srs, tn, s.wave, lane);
        lnf_c = g3r_bias(brs, tn, s.wave, lane);
    }
    // the half-tile phase 0 of the next K-tile would issue (see SEAM), behind the first row-operand loads so that their
    // wait does not include it; then the next tile's bias
    if (HALF) g3_issue<2>(s, nxt, 1, nk);       // (a 128-row item's stream: phase 0 of a K-tile issues the NEXT one's B-Y)
    else g3_issue<3>(s, nxt, 1, nk);
    const G3Bias nb = g3r_bias(brs, ntn, s.wave, lane);
    __builtin_amdgcn_sched_barrier(0);
    if (LNP) {
        if (NMT == 8) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lnq[0]), "+v"(lnq[1]), "+v"(lnq[2]), "+v"(lnq[3]), "+v"(lnq[4]), "+v"(lnq[5]), "+v"(lnq[6]), "+v"(lnq[7])::"memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lnq[0]), "+v"(lnq[1]), "+v"(lnq[2]), "+v"(lnq[3])::"memory");
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) lnf_row[mt] = __builtin_bit_cast(f32x2, lnq[mt]);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
        f32x4 ro[2][2];
        if (EPI == 2 || EPI == 3 || EPI == 6) {
            if (G8L) {
                unpack8(lanes2(rowop[mt][0], to_reg), ro[0][0], ro[0][1]);
                unpack8(lanes2(rowop[mt][1], to_reg), ro[1][0], ro[1][1]);
            } else {
                unpack(g3r_lanes(rowop[mt][0], to_reg), ro[0][0], ro[0][1]);
                unpack(g3r_lanes(rowop[mt][1], to_reg), ro[1][0], ro[1][1]);
            }
            if (mt + AHEAD < NMT) fetch(mt + AHEAD, rowop[mt + AHEAD]);
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
            __builtin_amdgcn_raw_buffer_store_b128(g3r_lanes(o0, to_mem), crs, (int)(coff + (2 * mt) * cstep), 0, G3_POL_C);
            __builtin_amdgcn_raw_buffer_store_b128(g3r_lanes(o1, to_mem), crs, (int)(coff + (2 * mt + 1) * cstep), 0, G3_POL_C);
            continue;
        }
        g3r_rows8(v[0][0], v[0][1], v[1][0], v[1][1]);
        float st_s[2], st_q[2], st_p[2];        // (STATS) this slab's two half slabs
#pragma unroll
        for (int h = 0; h < 2; ++h) {           // half A: rows 0..7 of the slab, half B: rows 8..15
            f32x4 v0 = v[h][0], v1 = v[h][1];
            if (EPI == 1) {
                if (PRE == 2 || G8S) {
                    // gelu and gelu' from the same Phi / Gaussian parts: the backward GEMM multiplies by the saved factor.  (The
                    // erf form stays here: with BOTH outputs wanted it shares one exponential between them, and the two
                    // polynomial chains of the bf16-mode forms measured no faster -- 344 against 339 us per fc1 launch.)
#if G3_TRAIN_GELU_POLY
                    const f32x4 t0 = gelu_clamp4(v0), t1 = gelu_clamp4(v1);
                    const f32x4 d0 = gelu_bf16_grad_from_t4(v0, t0), d1 = gelu_bf16_grad_from_t4(v1, t1);
                    __builtin_amdgcn_raw_buffer_store_b128(g3r_lanes(pack(d0, d1), to_mem), prs, (int)(poff + (2 * mt + h) * pstep), 0, G3_POL_P);
                    v0 = gelu_bf16_from_t4(v0, t0);
                    v1 = gelu_bf16_from_t4(v1, t1);
#else
                    f32x4 ph0, ga0, ph1, ga1;
                    phi_parts4(v0, ph0, ga0);
                    phi_parts4(v1, ph1, ga1);
                    const f32x4 d0 = ph0 + v0 * ga0 * 0.3989422804014327f, d1 = ph1 + v1 * ga1 * 0.3989422804014327f;
                    if (G8S) {
                        const u32x4 q8 = lanes2(pack8(d0, d1), to_mem);
                        __builtin_amdgcn_raw_buffer_store_b64(u32x2{q8[0], q8[1]}, prs, (int)(poff + (2 * mt + h) * pstep), 0, G3_POL_P);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b128(g3r_lanes(pack(d0, d1), to_mem), prs, (int)(poff + (2 * mt + h) * pstep), 0, G3_POL_P);
                    }
                    v0 *= ph0;
                    v1 *= ph1;
#endif
                } else {
                    if (SAVE) __builtin_amdgcn_raw_buffer_store_b128(g3r_lanes(pack(v0, v1), to_mem), prs, (int)(poff + (2 * mt + h) * pstep), 0, G3_POL_P);
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
            const u32x4 pk = pack(v0, v1);
            if (STATS) {
                // the values AS STORED (rounded to bf16): what the LayerNorm behind this launch reads
                unpack(pk, v0, v1);
                // pivot: chunk 0 of the row lives in lane (r & 7) of lane row 0
                unsigned pu = __float_as_uint(v0[0]);
                pu = __builtin_amdgcn_update_dpp(pu, pu, 0x128, 0xf, 0xc, false);                    // lanes 8..15 <- lanes 0..7
                pu = __builtin_amdgcn_permlane16_swap(pu, pu, false, false)[0];                       // lane rows 1, 3 <- rows 0, 2
                pu = __builtin_amdgcn_permlane32_swap(pu, pu, false, false)[0];                       // lane rows 2, 3 <- rows 0, 1
                const float P = __uint_as_float(pu);
                const f32x4 d0 = v0 - P, d1 = v1 - P;
                const f32x4 sd = d0 + d1, qd = d0 * d0 + d1 * d1;
                st_s[h] = (sd[0] + sd[1]) + (sd[2] + sd[3]);
                st_q[h] = (qd[0] + qd[1]) + (qd[2] + qd[3]);
                st_p[h] = P;
            }
            __builtin_amdgcn_raw_buffer_store_b128(g3r_lanes(pk, to_mem), crs, (int)(coff + (2 * mt + h) * cstep), 0, EPI == 2 ? G3_POL_C_RES : G3_POL_C);
        }
        if (STATS) {
            // reduce-scatter over the eight lanes of a row: lane bit 3 (DPP) picks the half slab, bit 4 (permlane16) the slab of
            // a pair, bit 5 (permlane32) the pair of a quadruple; after slab 3 / 7 one finished value per lane and quantity
            auto ror8 = [](float x) { return x + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x128, 0xf, 0xf, true)); };
            const bool b3 = (lane >> 3) & 1, b4 = (lane >> 4) & 1, b5 = (lane >> 5) & 1;
            const float a_s0 = ror8(st_s[0]), a_s1 = ror8(st_s[1]), a_q0 = ror8(st_q[0]), a_q1 = ror8(st_q[1]);
            st_a[mt & 1][0] = b3 ? a_s1 : a_s0;
            st_a[mt & 1][1] = b3 ? a_q1 : a_q0;
            st_a[mt & 1][2] = b3 ? st_p[1] : st_p[0];
            if (mt & 1) {
                auto fold16 = [](float x, float y) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
                    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
                };
                const int k = (mt >> 1) & 1;
                st_b[k][0] = fold16(st_a[0][0], st_a[1][0]);
                st_b[k][1] = fold16(st_a[0][1], st_a[1][1]);
                st_b[k][2] = b4 ? st_a[1][2] : st_a[0][2];
                if ((mt & 3) == 3) {
                    auto fold32 = [](float x, float y) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
                        return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
                    };
                    const float S = fold32(st_b[0][0], st_b[1][0]), Q = fold32(st_b[0][1], st_b[1][1]);
                    const float P = b5 ? st_b[1][2] : st_b[0][2];
                    // this lane's row of slabs mt - 3 .. mt: slab 4 (mt >> 2) + 2 b5 + b4, half b3, row r & 7
                    const float ds = S * (1.0f / 64.0f);
                    const u32x2 out = {__float_as_uint(P + ds), __float_as_uint(Q - S * ds)};
                    const int srow = wr * WROWS + 16 * (4 * (mt >> 2) + 2 * (int)b5 + (int)b4) + 8 * (int)b3 + (lane & 7);
                    asm volatile("ds_write_b64 %0, %1" ::"v"(lds_stats + (uint32_t)(srow * 32 + wc * 8)), "v"(out) : "memory");
                }
            }
        }
    }
    if (STATS) {
        // behind the last store: the four wave columns' (mean, M2) over 64 columns each -> ONE pair per row over the tile's 256 columns
        // (Chan's combination for equal counts, pairwise; thread t of the workgroup: row t >> 1 of the item, wave columns 2 (t & 1) + {0, 1},
        // the two halves of a row meet through one DPP swap), one 8-byte store per row -- a quarter of the partials, and of the
        // statistics stores, of the per-wave-column form; what the consumer (PRE 5) or me_row_stats_combine folds is 3 pairs per row at
        // C = 768 instead of 12.  Both wave rows stand side by side here (see gemm_g3r_kernel), so the barrier costs the skew of one epilogue.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int t = s.wave * 64 + lane, lr = t >> 1, h = t & 1;
        u32x4 raw;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(raw) : "v"(lds_stats + (uint32_t)(lr * 32 + h * 16)) : "memory");
        const float ma = __uint_as_float(raw[0]), qa = __uint_as_float(raw[1]), mb = __uint_as_float(raw[2]), qb = __uint_as_float(raw[3]);
        const float d1 = mb - ma;
        const float m1 = (ma + mb) * 0.5f, q1 = (qa + qb) + d1 * d1 * 32.0f;                 // n = 64 + 64
        auto swap1 = [](float x) { return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0xB1, 0xf, 0xf, true)); };      // quad_perm [1, 0, 3, 2]
        const float mo = swap1(m1), qo = swap1(q1);
        const float d2 = mo - m1;
        const u32x2 tot = {__float_as_uint((m1 + mo) * 0.5f), __float_as_uint((q1 + qo) + d2 * d2 * 64.0f)};      // n = 128 + 128
        __builtin_amdgcn_raw_buffer_store_b64(tot, srs, h == 0 ? lr * 8 : (int)0x80000000u, 0, 0);
    }
    if (LNP) {
        // the next item's pairs, behind this item's last store (the other pair buffer: slower waves may still be reading this item's)
        __builtin_amdgcn_sched_barrier(0);
        g3r_parts_fold(p, lnp, s.wave, lds_tick + 64 + (uint32_t)(par ^ 1) * 2048);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (HALF) {
        // as many memory operations as a whole tile's epilogue issues behind the A-Y half-tile (stores: half of them went out
        // above; late row-operand loads: a whole tile issues 2 (8 - AHEAD) of them, this one none)
        constexpr int PAD = (SAVE ? 16 : 8) + ((EPI == 2 || EPI == 6) ? 4 : EPI == 3 ? 8 : 0) ;      // (the statistics store: one per lane for either item kind)
        const u32x4 z = {0u, 0u, 0u, 0u};
        const __amdgpu_buffer_rsrc_t none = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < PAD; ++i) __builtin_amdgcn_raw_buffer_store_b128(z, none, 0, 0, 0);
    }
    // the next tile's accumulators start at its bias (s.binit)
    __builtin_amdgcn_sched_barrier(0);
    g3r_set_binit(s, nb, next_zero);
    if (ctr && s.wave == 0) {
        // everything this epilogue issued behind the draw may stay in flight: the A-Y half-tile (2), the bias (4), the
        // stores (16 / 32, padding included) and the row operands (16; a 128-row item 8 + 4 or 8 of the padding)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 + 4 + 16 + (SAVE ? 16 : 0) + ((EPI == 2 || EPI == 3 || EPI == 6) ? (HALF ? 8 : 16) : 0)) : "memory");
        g3r_publish(drawn, ctr, nx, lds_tick);
    }
}

// memory operations one g3_epilogue_r issues per wave behind the next tile's A-Y half-tile: >= the stores (+ the later
// row-operand loads); an under-count only makes the wait stricter
template <int EPI, int PRE> constexpr int g3r_seam() { return (PRE == 1 || PRE == 2 || (PRE == 6 && EPI == 1)) ? 32 : EPI >= 2 ? 20 : 16; }

// HI: this instantiation also carries the 128-row item form (run_item<HALF>): a second K-loop and epilogue in the kernel.  Only the
// bias-only forms are built with it (launch3r): around the row-operand epilogues the extra scalar state pushed 16-27 VGPRs of
// per-item spills into the seams (fc2 dgrad 244 -> 280 us in the training step), for items that buy next to nothing there.
template <int EPI, int PRE, bool HI = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_g3r_kernel(const GemmParams p) {
    constexpr int SEAM = g3r_seam<EPI, PRE>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2;

    const int bid = blockIdx.x, xcd = bid & 7, c = bid >> 3, G8 = gridDim.x >> 3;
    const int tiles = p.tiles_m * p.tiles_n, F = p.g3_full_tiles;
    const int nwork = F + (tiles - F) * 2;                             // (F == tiles unless p.g3_half)
    int nf = (F >> 3) + (xcd < (F & 7) ? 1 : 0);                       // whole tiles / all items of this XCD
    int nx = (nwork >> 3) + (xcd < (nwork & 7) ? 1 : 0);
    const int base_f = xcd * (F >> 3) + (xcd < (F & 7) ? xcd : (F & 7));
    const int base_all = xcd * (nwork >> 3) + (xcd < (nwork & 7) ? xcd : (nwork & 7));
    const int nkt = (int)(p.K / G3_BK);
    // Column groups (p.g3_colgroups = Gc > 1, round 6): a weight matrix whose column panels do not fit the XCD's 4 MiB L2 together (fc1:
    // 12 panels x 393 KB = 4.7 MB) is re-read from the Infinity Cache by every round of 32 concurrent tiles -- +343 MB per fc1 launch, at
    // ~64 pJ per byte (profiles/r06_pmc_fwd.json, r06_energy_probe.txt).  The XCDs then split the COLUMN tiles too: XCD x walks the
    // sub-grid (row group x / Gc of 8 / Gc, column group x % Gc of Gc) row-major, so its tiles_n / Gc weight panels stay resident; every
    // token panel is read by Gc XCDs instead of one (+ (Gc - 1) x the A bytes, the smaller side of the trade).  Whole tiles only.
    const int Gc = G3_COLGROUPS ? p.g3_colgroups : 1;
    int cg_c0 = 0, cg_cn = p.tiles_n, cg_r0 = 0;
    if (Gc > 1) {
        const int Rg = 8 / Gc, rg = xcd / Gc;
        cg_cn = p.tiles_n / Gc;
        cg_c0 = (xcd % Gc) * cg_cn;
        cg_r0 = rg * p.tiles_m / Rg;
        nf = nx = ((rg + 1) * p.tiles_m / Rg - cg_r0) * cg_cn;
    }
    auto decode = [&](int slot, int& tile, int& part, int& kt0, int& kt1) {
        part = -1; kt0 = 0; kt1 = nkt;
        if (Gc > 1) {
            const int jr = __builtin_amdgcn_readfirstlane(slot / cg_cn);
            tile = (cg_r0 + jr) * p.tiles_n + cg_c0 + (slot - jr * cg_cn);
        } else if (slot < nf) {
            tile = base_f + slot;
        } else {
            // a tile of the last, mostly empty round: two 128-row items (part = which half), the whole reduction each
            const int pi = base_all - base_f + (slot - nf);
            tile = F + (pi >> 1);
            part = pi & 1;
        }
    };
    auto src_of = [&](int tile, int part, int& tm, int& tn) {
        tm = __builtin_amdgcn_readfirstlane(tile / p.tiles_n);
        tn = tile - tm * p.tiles_n;
        return part >= 0 ? g3_make_src_half(p, tm, tn, part, wr) : g3_make_src(p, tm, tn);
    };
    const bool dyn = p.g3_tickets != nullptr;
    unsigned* const ctr = dyn ? p.g3_tickets + xcd * 16 : nullptr;
    const uint32_t lds_tick = (uint32_t)(uintptr_t)smem + G3_LDS;
    int slot = c;
    if (slot >= nx) return;
    if (kMeDev && (p.debug >> 4)) {              // dev: stagger the CUs of an XCD (their output bursts spread out)
        const int n = c * (p.debug >> 4);
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(4);
    }

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

    unsigned drawn0 = 0;
    if (dyn && wave == 0) drawn0 = g3r_draw(ctr);          // item 1 (item 0 is this workgroup's own slot)

    int tile, part, kt0, kt1, tm, tn;
    decode(slot, tile, part, kt0, kt1);
    G3Src cur = src_of(tile, part, tm, tn);
    g3_issue<0>(s, cur, 0, kt0); g3_issue<1>(s, cur, 0, kt0); g3_issue<2>(s, cur, 0, kt0); g3_issue<3>(s, cur, 0, kt0);
    g3_issue<0>(s, cur, 1, kt0 + 1); g3_issue<1>(s, cur, 1, kt0 + 1); g3_issue<2>(s, cur, 1, kt0 + 1); g3_issue<3>(s, cur, 1, kt0 + 1);
    {
        const G3Bias b0 = g3r_bias(brs, tn, wave, lane);
        g3r_set_binit(s, b0, PRE == 3 || PRE == 5);
    }
    if (PRE == 5) {      // the first item's LayerNorm pairs (every later item's are folded behind the epilogue in front of it)
        const int64_t m00 = (int64_t)tm * G3_BM + (part >= 0 ? part * 128 : 0);
        int64_t r0 = p.M - m00;
        r0 = r0 < (part >= 0 ? 128 : 256) ? r0 : (part >= 0 ? 128 : 256);
        const G3Parts q0 = g3r_parts_load(p, m00, r0 > 0 ? (int)r0 : 0, wave);
        g3r_parts_fold(p, q0, wave, lds_tick + 64);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (dyn && wave == 0) g3r_publish(drawn0, ctr, nx, lds_tick);
    prime();
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();

    // dev: time stamps (s_memtime) of waves 0 and 4: [workgroup][wave row][item][8] = item start, first K-tile done,
    // second K-tile done, K-loop done, rows realigned, epilogue done (tools/gemm_dev, debug bit 8; dead code in the product build)
    unsigned long long* trace = kMeDev ? reinterpret_cast<unsigned long long*>(p.colsum_ws) : nullptr;
    int item = 0;
    int item_par = 0;                    // (PRE 5) which LDS pair buffer holds the current item's LayerNorm pairs
#define G3R_STAMP(i)                                                                                            \
    if (kMeDev && trace && (wave & 3) == 0 && item < 16) {                                                      \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                             \
        if (lane == 0) trace[(((size_t)bid * 2 + wr) * 16 + item) * 8 + (i)] = t_;                              \
        if ((i) == 0 || (i) == 5) {       /* the constant-rate counter next to the shader clock: effective clock under load */ \
            const unsigned long long r_ = __builtin_amdgcn_s_memrealtime();                                     \
            if (lane == 0) trace[(((size_t)bid * 2 + wr) * 16 + item) * 8 + ((i) == 0 ? 6 : 7)] = r_;           \
        }                                                                                                        \
    }
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
                nxt = src_of(ntile, npart, ntm, ntn);
            }
        };
        if (!dyn) resolve_next(slot + G8);
        const int np = (kt1 - kt0) >> 1;
        // One item: the K-loop and the epilogue, instantiated for whole tiles and for 128-row items (HALF).  The two forms meet
        // only behind their epilogues, where no accumulator is live.
        auto run_item = [&](auto half_tag) {
            constexpr bool HALF = decltype(half_tag)::value;
            {
                G3Src sb = cur;
                int kb = kt0 + 2, kc = kt0 + 3;
                if (np == 1) { sb = nxt; kb = nkt0; kc = nkt0 + 1; }
                g3_ktile<0, false, SEAM, HALF>(s, cur, 0, sb, kb);
                G3R_STAMP(1)
                g3_ktile<1, false, 0, HALF, HALF ? SEAM : 0>(s, sb, kb, sb, kc);
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
                g3_ktile<0, false, 0, HALF>(s, cur, k + 1, sb, kb);
                g3_ktile<1, false, 0, HALF>(s, sb, kb, sb, kc);
            }
            // the two wave rows run their epilogues SIDE BY SIDE: left one barrier apart, row 1 could not start its epilogue
            // before row 0 had finished its own and reached the next K-tile's first barrier, and row 0 would then wait out
            // row 1's (measured with the time stamps below: 4.6 k of 40 k clocks per tile).  Row 0 gives up its one-barrier
            // lead here and row 1 re-opens it behind the epilogue.
            G3R_STAMP(3)
            if (wr == 0) __builtin_amdgcn_s_barrier();
            G3R_STAMP(4)
            int64_t nm0 = 0;
            int nrows = 0;
            if (PRE == 5 && has_next) {
                nm0 = (int64_t)ntm * G3_BM + (npart >= 0 ? npart * 128 : 0);
                const int64_t left = p.M - nm0, cap = npart >= 0 ? 128 : 256;
                nrows = (int)(left < cap ? (left > 0 ? left : 0) : cap);
            }
            g3_epilogue_r<EPI, PRE, HALF>(p, s, (int64_t)tm * G3_BM + (HALF ? part * 128 : 0), tn, 0, nxt, nkt0 + 1, brs, ntn, PRE == 3 || PRE == 5,
                                          has_next ? ctr : nullptr, nx, lds_tick, nm0, nrows, item_par);
        };
        if constexpr (HI) {
            if (part >= 0) run_item(std::true_type{});
            else run_item(std::false_type{});
        } else {
            run_item(std::false_type{});
        }
        G3R_STAMP(5)
        if (kMeDev) ++item;
        if (!has_next) break;
        if (wr == 1) __builtin_amdgcn_s_barrier();
        item_par ^= 1;
        slot = nslot; tile = ntile; part = npart; kt0 = nkt0; kt1 = nkt1; tm = ntm; tn = ntn;
        cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing null DMAs must land before the LDS is released
}

// ---- wgrad: one (output tile, K-range) per workgroup; raw fp32 partial sums into slab blockIdx.y ... the deterministic
// fold (splitk_reduce_kernel, gemm.hip) sums the slabs and applies the epilogue.  Work ids are split-major (id = split *
// tiles + tile): the workgroups an XCD runs together read the SAME rows of dY and X at the same time, so every operand
// row comes out of HBM once and is shared through that XCD's L2.
// FOLD: the split-K fold runs INSIDE the launch (every workgroup is resident: tiles x splits <= CUs).  The S workgroups of an output
// tile reduce-scatter their partial tiles: the tile is cut into 32 units (16 rows x the 32-column halves of the four wave columns),
// unit u belongs to split (u S) >> 5; a workgroup writes the units it does not own into its slab (write-through stores: no
// release fence), announces itself on the tile ROW's counter, waits for the S x tiles_n workgroups of that tile row, and then sums its
// own units -- own registers first, the other splits in ascending order: deterministic -- and applies the real epilogue (alpha,
// beta C, output dtype).  One workgroup of the tile row also folds the partial column sums (the bias gradient).  Replaces the
// separate fold launch (48 per training step: ~25 us each + a kernel boundary) by ~1/S of its traffic per workgroup.  Hand-off
// protocol: cdna_hip_programming.md, Guideline 16 (sc1 payload, every storing wave drains, one lane publishes / polls relaxed,
// one agent-scope acquire, then plain loads); the counters are zeroed by a memset node ahead of every launch.
template <bool FOLD>
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
    if (kMeDev && (p.debug & 1)) {
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) keep += s.acc[i][j][0] + s.acc[i][j][1] + s.acc[i][j][2] + s.acc[i][j][3];
        if (keep == 1.2345e-30f) reinterpret_cast<float*>(p.C)[0] = keep;
        return;
    }
    if (!FOLD) {
        g3_epilogue<5>(p, s, m0, n0, lane, reinterpret_cast<float*>(p.C) + (int64_t)split * p.slab_stride, 0);
        return;
    }
    // ---- in-kernel fold
    const int S = p.split_k;
    const int wc = wave & 3;
    const int lr = lane & 15, lg = lane >> 4;
    const int64_t mrow = m0 + wr * 128 + lr;
    int64_t ncol[2];
    bool n_ok[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        ncol[q] = n0 + wc * 64 + (2 * q + (lg & 1)) * 16 + 8 * (lg >> 1);
        n_ok[q] = ncol[q] + 8 <= p.N;
    }
    const __amdgpu_buffer_rsrc_t mine = __builtin_amdgcn_make_buffer_rsrc(p.g3_slabs + (int64_t)split * p.slab_stride, 0, (int)(p.M * p.N * 4), 0x00020000);
    // phase 1: re-deal every 16 x 16 pair into 8 consecutive columns per lane (kept in the accumulators), park the foreign units
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            f32x4 v0 = s.acc[mt][2 * q], v1 = s.acc[mt][2 * q + 1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v0[e]), __float_as_uint(v1[e]), false, false);
                v0[e] = __uint_as_float(sw[0]);
                v1[e] = __uint_as_float(sw[1]);
            }
            s.acc[mt][2 * q] = v0; s.acc[mt][2 * q + 1] = v1;
            const int u = ((wr * 8 + mt) << 1) + q;
            if (((u * S) >> 5) != split) {               // (wave-uniform)
                const int64_t m = mrow + mt * 16;
                const uint32_t off = (m < p.M && n_ok[q]) ? (uint32_t)((m * p.N + ncol[q]) * 4) : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v0), mine, (int)off, 0, 16);        // sc1: write-through
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v1), mine, (int)(off + 16), 0, 16);
            }
        }
    // publish / wait: every workgroup of this tile ROW (all N-tiles, all splits) -- the column sums need all of them anyway
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        unsigned* ctr = p.g3_tickets + tm;
        const unsigned want = (unsigned)(S * p.tiles_n);
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1u << 26)) __builtin_trap();      // (seconds: a workgroup of the row never ran -- the launch was not resident)
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    // phase 2: own units = own registers + the other splits' parked values, ascending; four splits' loads in flight at a time
    const float* slab0 = p.g3_slabs;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int u = ((wr * 8 + mt) << 1) + q;
            if (((u * S) >> 5) != split) continue;       // (wave-uniform)
            const int64_t m = mrow + mt * 16;
            const bool ok = m < p.M && n_ok[q];
            f32x4 v0 = s.acc[mt][2 * q], v1 = s.acc[mt][2 * q + 1];
            const float* src = slab0 + (ok ? m * p.N + ncol[q] : 0);
            for (int sb = 0; sb < S; sb += 4) {
                f32x4 t0[4], t1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int sj = sb + j < S ? sb + j : S - 1;
                    t0[j] = *reinterpret_cast<const f32x4*>(src + (int64_t)sj * p.slab_stride);
                    t1[j] = *reinterpret_cast<const f32x4*>(src + (int64_t)sj * p.slab_stride + 4);
                }
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool use = sb + j < S && sb + j != split;
                    v0 += use ? t0[j] : z;
                    v1 += use ? t1[j] : z;
                }
            }
            if (ok) {
                epilogue_quad_lin(p, m, ncol[q], v0);
                epilogue_quad_lin(p, m, ncol[q] + 4, v1);
            }
        }
    // the bias gradient: partial rows [split * tiles_n + tn][M] of this tile row, folded in row order by one workgroup
    if (do_cs && p.tn_colsum_out && split == 0 && tn == 0 && tid < 256) {
        const int64_t m = m0 + tid;
        if (m < p.M) {
            const int n_part = S * p.tiles_n;
            float t = 0.f;
            for (int j = 0; j < n_part; j += 4) {
                float c[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) c[e] = p.colsum_ws[(int64_t)(j + e < n_part ? j + e : n_part - 1) * p.M + m];
#pragma unroll
                for (int e = 0; e < 4; ++e) t += j + e < n_part ? c[e] : 0.f;
            }
            p.tn_colsum_out[m] = p.beta != 0.0f ? t + p.beta * p.tn_colsum_out[m] : t;
        }
    }
}


// ---- wgrad on a workgroup count that is NOT a multiple of the tile count (a communication kernel holds R CUs: me_gemm_reserve_cus).  The
// (tile, split) grid above gives every workgroup one whole item and wants all 256 CUs at once; with ANY CU held its last workgroups run as a
// SECOND ROUND (+100 % per launch, +3.5 % per train step for R = 8 .. 32: profiles/r05_contention.txt), and a uniform split for 256 - R slots
// quantises badly (36 tiles on 240 slots: 6 parts of 132 K-tiles instead of 7 of 114 = +16 %).  Here p.sk_wgs workgroups carry the same load:
//   * workgroups [0, S T): S = p.sk_levels whole split levels of L1 = p.sk_l1 K-tile pairs per tile, split-major exactly as above -- the
//     workgroups an XCD runs together read the SAME rows of dY and X (a first version that cut the whole tile-major unit list evenly lost
//     that sharing: every workgroup streamed private rows, ~5x the operand traffic, +22 % on the weight gradients: profiles/r06_contention.txt);
//   * the E = sk_wgs - S T workgroups left over share the LEFTOVER U - S L1 pairs of every tile as one tile-major list, evenly, across tile
//     boundaries: such a workgroup may end one tile and begin the next (a segment and a slab each).
// Parts stay static -- a part is a fixed K range and a fixed slab, the fold sums a tile's slabs in K order (levels, then leftover parts) -- so
// the result is bit-reproducible per (shape, workgroups).  Bias gradient: the N-tiles of a tile row share the column sums round-robin pair by
// pair ([level x tiles_n + tn][M] partial rows for the levels, [S tiles_n + leftover part x tiles_n + tn][M] for the leftover segments).
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_g3tn_sk_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);       // neighbours in the id list share an XCD
    const int T = p.tiles_m * p.tiles_n, S = p.sk_levels, L1 = p.sk_l1, E = p.sk_wgs - S * T, Ul = p.sk_upt - S * L1;
    const int64_t TUl = (int64_t)T * Ul;
    const bool regular = wgid < S * T;
    // leftover workgroups: units [u0, u1) of the tile-major leftover list; a regular workgroup is one segment (u0 = 0, u1 = 1: one pass)
    int64_t u0 = 0, u1 = 1;
    if (!regular) {
        const int e = wgid - S * T;
        u0 = __builtin_amdgcn_readfirstlane((int)((int64_t)e * TUl / E));
        u1 = __builtin_amdgcn_readfirstlane((int)((int64_t)(e + 1) * TUl / E));
    }
    G3State s;
    g3_init_lane_tn(s, p, smem, wave, lane);
    const G3Src null = g3_null_src(p);
    const bool do_cs = p.colsum_ws != nullptr;
    while (u0 < u1) {
        int tile, kt0, kt1, part, cs_row;
        int64_t ue;
        if (regular) {
            const int level = __builtin_amdgcn_readfirstlane(wgid / T);
            tile = wgid - level * T;
            kt0 = level * L1 * 2; kt1 = kt0 + L1 * 2;
            part = level;
            ue = u1;
        } else {
            tile = __builtin_amdgcn_readfirstlane((int)(u0 / Ul));
            const int64_t tu0 = (int64_t)tile * Ul;
            ue = u1 < tu0 + Ul ? u1 : tu0 + Ul;
            kt0 = (S * L1 + (int)(u0 - tu0)) * 2; kt1 = (S * L1 + (int)(ue - tu0)) * 2;
            part = S + (wgid - S * T) - sk_first(tile, E, Ul, TUl);
        }
        const int tm = __builtin_amdgcn_readfirstlane(tile / p.tiles_n), tn = tile - tm * p.tiles_n;
        cs_row = regular ? part * p.tiles_n + tn : S * p.tiles_n + (part - S) * p.tiles_n + tn;
        const int64_t m0 = (int64_t)tm * G3_BM, n0 = (int64_t)tn * G3_BN;
        const G3Src src = g3_make_src_tn(p, tm, tn);
        g3_zero(s);
        g3_issue<0>(s, src, 0, kt0); g3_issue<1>(s, src, 0, kt0); g3_issue<2>(s, src, 0, kt0); g3_issue<3>(s, src, 0, kt0);
        g3_issue<0>(s, src, 1, kt0 + 1); g3_issue<1>(s, src, 1, kt0 + 1); g3_issue<2>(s, src, 1, kt0 + 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();
        // column sums: the N-tiles of a tile row stage the same dY rows and share them pair by pair, round-robin -- inside a level all N-tiles walk
        // the same K range; in the leftover the segments of different N-tiles are ragged, so the turn is tied to the ABSOLUTE pair index (pair k
        // belongs to N-tile k % tiles_n: whichever of that tile's segments covers it takes it).  (A first version let the tn = 0 segments sum
        // every pair: those workgroups became the launch's tail, +33 % -- profiles/r06_contention.txt.)
        const bool cs_any = do_cs;
        s.cs[0] = s.cs[1] = 0.f;
        int cs_turn = regular ? tn : (tn + p.tiles_n - ((kt0 >> 1) % p.tiles_n)) % p.tiles_n;
        auto my_turn = [&]() {
            if (!cs_any) return false;
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
        if (wr == 0) __builtin_amdgcn_s_barrier();          // (both wave rows are past their last fragment read: the next segment may refill the LDS)
        if (cs_any) {
            float c0 = s.cs[0], c1 = s.cs[1];
            c0 += __shfl_xor(c0, 16, 64); c0 += __shfl_xor(c0, 32, 64);
            c1 += __shfl_xor(c1, 16, 64); c1 += __shfl_xor(c1, 32, 64);
            if (lane < 16) {
                float* row = p.colsum_ws + (int64_t)cs_row * p.M;
                const int wcol = wave & 3;
                const int64_t ma = m0 + wr * 128 + wcol * 16 + lane, mb = ma + 64;
                if (ma < p.M) row[ma] = c0;
                if (mb < p.M) row[mb] = c1;
            }
        }
        g3_epilogue<5>(p, s, m0, n0, g3_lane_now(), reinterpret_cast<float*>(p.C) + (int64_t)part * p.slab_stride, 0);
        u0 = ue;
    }
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
#ifndef G3_HI_EPI2
#define G3_HI_EPI2 1                           // (A/B arm: 128-row items in the plain residual kernel too)
#endif
#ifndef G3_HI_EPI1
#define G3_HI_EPI1 1                           // (A/B arm: ... and in the GELU kernels -- fc1's 2 364 tiles leave a last round of 60)
#endif
#ifndef G3_HI_EPI6
#define G3_HI_EPI6 0                           // (A/B arm: ... and in the x-row-operand kernel, fc2 dgrad, the same 2 364 tiles: 243 -> 251 us
                                               //  sustained with four row-operand slabs in flight, 263 with two; train +0.4 ms with six -- off)
#endif
    constexpr bool HI = EPI == 0 || (G3_HI_EPI2 && EPI == 2) || (G3_HI_EPI1 && EPI == 1) || (G3_HI_EPI6 && EPI == 6);      // which forms carry the 128-row items (see the kernel)
    constexpr int LDS_BYTES = G3_LDS + 64 + (PRE == 5 ? 4096 : PRE == 4 ? 8192 : 0);      // operand buffers + the ticket word (+ PRE 5: two buffers of 256 LayerNorm row pairs; PRE 4: 256 rows x 4 wave columns of partial pairs)
    static OncePerDevice once;
    if (once.need())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_g3r_kernel<EPI, PRE, HI>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    GemmParams q = q0;
    // Tile quantisation: T tiles on G resident workgroups take ceil(T / G) rounds, and the encoder's N = 768 outputs are 591 tiles =
    // 2.31 rounds (N = 3072: 9.23).  When at most half the CUs would work in the last round, its tiles run as two 128-row items
    // each -- same kernel, same epilogue, no slabs (g3_phase<.., HALF>): the last round then costs a little over half a round.
    const int tiles = q.tiles_m * q.tiles_n, rem = tiles % G;
    q.g3_full_tiles = tiles; q.g3_split = 1; q.g3_half = 0;
    // column groups (see the kernel): the fewest groups that make an XCD's share of the weight panels fit its L2 next to the token panels in
    // flight (<= 2.5 MB), when the column tiles divide evenly and every XCD still gets a round of tiles
    q.g3_colgroups = 1;
    if (G3_COLGROUPS && G == 256 && q.tiles_n >= 4) {
        const int64_t panel = 256 * q.K * 2;
        for (int gc = 1; gc <= 4; gc *= 2) {
            if (q.tiles_n % gc) break;
            if ((q.tiles_n / gc) * panel <= (5ll << 19)) {
                // worth it when the weight re-reads it removes (one pass over B per round of 32 tiles and XCD) outweigh the token re-reads it adds
                const int64_t b_reread = q.N * q.K * 2 * ((int64_t)tiles / 32), a_extra = (int64_t)(gc - 1) * q.M * q.K * 2;
                if (gc > 1 && (int64_t)(q.tiles_m / (8 / gc)) * (q.tiles_n / gc) >= 64 && b_reread > 2 * a_extra) q.g3_colgroups = gc;
                break;
            }
        }
    }
    if (q.g3_colgroups > 1) {
        // (whole tiles only: the 128-row items' bookkeeping assumes the row-major tile list)
    } else if (HI && tiles >= G && rem > 0 && 2 * rem <= G && q.K >= 4 * G3_BK && gemm_dev().tail_split != 3) {
        q.g3_full_tiles = tiles - rem;
        q.g3_half = 1;
    }
    // claimed items need >= 2 K-tile pairs per item (see the kernel)
    q.g3_tickets = (q.K >= 4 * G3_BK && gemm_dev().g3_persistent == 1) ? g3r_tickets(stream) : nullptr;
    if (kMeDev && gemm_dev().tail_split == 2) q.g3_tickets = nullptr;          // dev: "g3s" = static schedule
    ME_DEV_ONLY(q.colsum_ws = (q.debug & 8) ? reinterpret_cast<float*>(g_gemm_dev_trace) : nullptr;)
    hipLaunchKernelGGL((gemm_g3r_kernel<EPI, PRE, HI>), dim3((unsigned)G), dim3(512), LDS_BYTES, stream, q);
    ME_CHECK_LAUNCH("me_gemm(g3 resident)");
    return ME_OK;
}

int launch3r_any(int epi, int pre, const GemmParams& q, int G, hipStream_t stream) {
    switch (epi) {
        case 0: return pre == 5 ? launch3r<0, 5>(q, G, stream) : pre == 3 ? launch3r<0, 3>(q, G, stream) : launch3r<0, 0>(q, G, stream);
        case 1: return pre == 6 ? launch3r<1, 6>(q, G, stream) : pre == 5 ? launch3r<1, 5>(q, G, stream) : pre == 3 ? launch3r<1, 3>(q, G, stream) : pre == 2 ? launch3r<1, 2>(q, G, stream) : pre ? launch3r<1, 1>(q, G, stream) : launch3r<1, 0>(q, G, stream);
        case 2: return pre == 4 ? launch3r<2, 4>(q, G, stream) : launch3r<2, 0>(q, G, stream);
        case 3: return launch3r<3, 0>(q, G, stream);
        default: return pre == 6 ? launch3r<6, 6>(q, G, stream) : launch3r<6, 0>(q, G, stream);
    }
}

template <int EPI> int launch3e(const GemmParams& p, void* ws, hipStream_t stream) {
    static OncePerDevice once;
    if (once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_g3_kernel<EPI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS);
    }
    const int tiles = p.tiles_m * p.tiles_n;
    if (kMeDev && ws) return launch_g3p(p, EPI, ws, stream);       // dev build: the persistent stream-K form (gemm3_dev.hip)
    (void)ws;
    GemmParams q = p;
    if (q.g3_split <= 1 || q.g3_slabs == nullptr) { q.g3_full_tiles = tiles; q.g3_split = 1; q.g3_ktp = 0; q.g3_slabs = nullptr; }
    const int nwg = q.g3_full_tiles + (tiles - q.g3_full_tiles) * q.g3_split;
    // the resident form (one workgroup per CU, operand stream running through the epilogues) whenever every CU gets work
    // and the output / row operands are bf16 with tile-local 32-bit offsets
    if (p.a_wrap_kt) {
        // A held as two planes [hi | lo], three reduction segments (me_gemm_desc.a_wrap_k): the one-tile kernel's WRAP instantiations only
        if constexpr (EPI == 0 || EPI == 4 || EPI == 8) {
            static OncePerDevice once_w;
            if (once_w.need())
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_g3_kernel<EPI, true>), hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS);
            hipLaunchKernelGGL((gemm_g3_kernel<EPI, true>), dim3((unsigned)nwg), dim3(512), G3_LDS, stream, q);
            ME_CHECK_LAUNCH("me_gemm(g3, wrapped A)");
            return ME_OK;
        } else {
            me_set_error("me_gemm: a_wrap_k is served for plain / fp32-residual epilogues only (see me_gemm_takes_a_wrap)");
            return ME_ERR_UNSUPPORTED;
        }
    }
    if (gemm_dev().g3_persistent == 1) {
        int repi = EPI <= 3 ? EPI : -1, pre = (EPI == 1 && p.preact) ? 1 : 0;
        const bool plain = p.beta == 0.0f && p.out_group_rows == 0 && p.res_row_mod == 0 && !p.colscale && !p.residual;
        if (EPI == 4 && p.row_affine && !p.flags && plain && !p.preact && !p.aux) {       // folded LayerNorm (bias / GELU forms)
            repi = p.act == ME_ACT_GELU ? 1 : 0;
            pre = p.row_nparts ? 5 : 3;       // (5: row_affine holds the 64-column partials of the launch in front, see g3_epilogue_r)
        }
        if (EPI == 4 && p.flags && !p.row_affine) {
            // the two halves of the "save gelu'" pair (pick_epi sends flagged descriptors to the generic epilogue)
            if (plain && p.flags == ME_GEMM_SAVE_GELU_GRAD && p.act == ME_ACT_GELU && p.preact && !p.aux) { repi = 1; pre = 2; }
            if (plain && p.flags == ME_GEMM_AUX_IS_FACTOR && p.act == ME_ACT_NONE && p.aux && (p.aux_dtype == ME_BF16 || p.aux_dtype == ME_GG8) && !p.preact) repi = 6;
        }
        if (EPI == 6) repi = 6;                                   // (pick_epi_ex has checked the same conditions)
        if (EPI == 7) { repi = 1; pre = 2; }
        if (repi == 1 && pre == 2 && p.preact_dtype == ME_GG8) pre = 6;       // gelu' in eight bits, both halves of the pair
        if (repi == 6 && p.aux_dtype == ME_GG8) pre = 6;
        if (EPI == 2 && p.row_stats) pre = 4;
        const int G = g3_cus() & ~7;
        const int64_t ldmax = std::max(std::max(p.ldc, p.preact ? p.ldpre : 0), std::max(p.residual ? p.ldres : 0, p.aux ? p.ldaux : 0));
        if (repi >= 0 && !p.colscale && G >= 8 && nwg >= G && q.g3_split <= 1 && p.alpha == 1.0f && p.c_dtype == ME_BF16 && (!(pre == 1 || pre == 2) || p.preact_dtype == ME_BF16) && (pre != 6 || !p.preact || p.ldpre % 8 == 0) &&
            256 * ldmax * 2 < (1ll << 31))
            return launch3r_any(repi, pre, q, G, stream);
    }
    if (p.row_stats) {
        me_set_error("me_gemm: row_stats needs the resident residual kernel (see me_gemm_emits_row_stats)");
        return ME_ERR_UNSUPPORTED;
    }
    if (p.row_nparts) {
        me_set_error("me_gemm: row_parts needs the resident kernel's folded-LayerNorm epilogue (see me_gemm_takes_row_parts)");
        return ME_ERR_UNSUPPORTED;
    }
    if ((p.preact && p.preact_dtype == ME_GG8) || (p.aux && p.aux_dtype == ME_GG8)) {
        me_set_error("me_gemm: ME_GG8 needs the resident kernel (see me_gemm_takes_gg8)");
        return ME_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL((gemm_g3_kernel<EPI>), dim3((unsigned)nwg), dim3(512), G3_LDS, stream, q);
    ME_CHECK_LAUNCH("me_gemm(g3)");
    return ME_OK;
}

}  // namespace

// the same per-stream counter sets for the other persistent kernels of the library (attention): words 1..15 of a set are free (the
// resident GEMM uses words 0, 16, 32, ...); null while the stream is capturing -> static schedule
unsigned* me_work_counters(hipStream_t stream) { return g3r_tickets(stream); }

// TN: p.split_k slabs of p.ksteps_per_split K-tiles (of 64) each into p.C = [split_k][M][N] fp32
int launch_g3_tn(const GemmParams& p, hipStream_t stream) {
    static OncePerDevice once;
    if (once.need())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_g3tn_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS);
    const int nwg = p.tiles_m * p.tiles_n * p.split_k;
    hipLaunchKernelGGL(gemm_g3tn_kernel<false>, dim3((unsigned)nwg), dim3(512), G3_LDS, stream, p);
    ME_CHECK_LAUNCH("me_gemm(g3 tn)");
    return ME_OK;
}

int launch_g3_tn_sk(const GemmParams& p, hipStream_t stream) {
    static OncePerDevice once;
    if (once.need())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_g3tn_sk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS);
    hipLaunchKernelGGL(gemm_g3tn_sk_kernel, dim3((unsigned)p.sk_wgs), dim3(512), G3_LDS, stream, p);
    ME_CHECK_LAUNCH("me_gemm(g3 tn, balanced partition)");
    return ME_OK;
}

#if G3_TN_FOLD
// every workgroup of the launch must be resident at the same time (they wait for each other): one per CU, at most as many as CUs;
// slab offsets are 32-bit; at most 32 splits (the ownership map has 32 units per tile)
bool g3_tn_fold_ok(const GemmParams& p, int split_k) {
    const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n * split_k;
    return nwg <= g3_cus() && split_k <= 32 && p.M * p.N * 4 < (1ll << 31) && p.out_group_rows == 0 && p.res_row_mod == 0;
}

int launch_g3_tn_fold(const GemmParams& p, hipStream_t stream) {
    static OncePerDevice once;
    if (once.need())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_g3tn_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS);
    const int nwg = p.tiles_m * p.tiles_n * p.split_k;
    if (hipMemsetAsync(p.g3_tickets, 0, (size_t)p.tiles_m * sizeof(unsigned), stream) != hipSuccess) {
        me_set_error("me_gemm(g3 tn fold): hipMemsetAsync of the tile-row counters failed");
        return ME_ERR_HIP;
    }
    hipLaunchKernelGGL(gemm_g3tn_kernel<true>, dim3((unsigned)nwg), dim3(512), G3_LDS, stream, p);
    ME_CHECK_LAUNCH("me_gemm(g3 tn fold)");
    return ME_OK;
}
#endif

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

// Will launch_g3(p, 2, ..) run the resident residual kernel, i.e. can p.row_stats be served?  (the conditions of launch3e, restated
// for a descriptor that has not been planned yet: whole tiles, every CU gets one, bf16 output and residual, plain epilogue)
bool g3_emits_row_stats(const GemmParams& p) {
    if (!g3_supported(p, ME_GEMM_NT) || gemm_dev().g3_persistent != 1 || pick_epi(p) != 2 || p.colscale) return false;
    const int G = g3_cus() & ~7;
    const int64_t tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    const int64_t ldmax = std::max(p.ldc, p.ldres);
    return G >= 8 && tiles >= G && p.alpha == 1.0f && p.c_dtype == ME_BF16 && p.res_dtype == ME_BF16 && p.N % 256 == 0 &&
           256 * ldmax * 2 < (1ll << 31) && p.M * 8 < (1ll << 31);
}

// Will launch_g3 run the resident kernel's folded-LayerNorm epilogue on partials (PRE 5), i.e. can me_gemm_desc.row_parts be served?
// (the conditions of launch3e restated, as above; p.row_affine / row_nparts / col_shift already set by fill_params)
#ifndef ME_NO_ROW_PARTS
#define ME_NO_ROW_PARTS 0                      // (A/B arm: 1 = never; the Blocks then run me_row_stats_combine between the GEMMs, as round 5 did)
#endif
bool g3_takes_row_parts(const GemmParams& p) {
    if (ME_NO_ROW_PARTS) return false;
    if (!g3_supported(p, ME_GEMM_NT) || gemm_dev().g3_persistent != 1 || !p.row_affine || !p.col_shift) return false;
    if (p.row_nparts < 1 || p.row_nparts > 4 || (int64_t)p.row_nparts * 256 != p.K) return false;
    if (p.flags || p.preact || p.aux || p.residual || p.colscale || p.beta != 0.0f || p.out_group_rows != 0 || p.res_row_mod != 0) return false;
    const int G = g3_cus() & ~7;
    const int64_t tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    return G >= 8 && tiles >= G && p.alpha == 1.0f && p.c_dtype == ME_BF16 && 256 * p.ldc * 2 < (1ll << 31) && p.M * 8 * 16 < (1ll << 31);
}

// ME_GG8 (gelu' in eight bits): the two flagged descriptors of the training MLP, when launch3e sends them to the resident kernel
bool g3_takes_gg8(const GemmParams& p) {
#ifdef ME_NO_GG8
    return false;                               // (A/B arm)
#endif
    if (!g3_supported(p, ME_GEMM_NT) || gemm_dev().g3_persistent != 1) return false;
    if (p.row_affine || p.residual || p.colscale || p.beta != 0.0f || p.out_group_rows != 0 || p.res_row_mod != 0 || p.a_wrap_kt) return false;
    const bool save = p.flags == ME_GEMM_SAVE_GELU_GRAD && p.act == ME_ACT_GELU && p.preact && p.preact_dtype == ME_GG8 && !p.aux;
    const bool load = p.flags == ME_GEMM_AUX_IS_FACTOR && p.act == ME_ACT_NONE && p.aux && p.aux_dtype == ME_GG8 && !p.preact;
    if (!save && !load) return false;
    const int G = g3_cus() & ~7;
    const int64_t tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    const int64_t ldmax = std::max(p.ldc, save ? p.ldpre : p.ldaux);
    return G >= 8 && tiles >= G && p.alpha == 1.0f && p.c_dtype == ME_BF16 && 256 * ldmax * 2 < (1ll << 31);
}

// scratch of the persistent stream-K form (dev build): one fp32 partial tile per workgroup + the hand-over flags (+ 1
// error word)
size_t g3_workspace_bytes() {
    const size_t G = (size_t)g3_cus();
    return G * G3_SLAB_FLOATS * sizeof(float) + (G + 1) * sizeof(unsigned);
}

// one tile per workgroup; in the dev build ws != nullptr (>= g3_workspace_bytes()) selects the persistent stream-K form
int launch_g3(const GemmParams& p, int epi, void* ws, hipStream_t stream) {
    if (p.colscale && epi != 2 && epi != 8) epi = 4;      // (the residual forms carry the column scale -- layer-scale Blocks; the others do not)
    switch (epi) {
        case 0: return launch3e<0>(p, ws, stream);
        case 1: return launch3e<1>(p, ws, stream);
        case 2: return launch3e<2>(p, ws, stream);
        case 3: return launch3e<3>(p, ws, stream);
        case 6: return launch3e<6>(p, ws, stream);
        case 7: return launch3e<7>(p, ws, stream);
        case 8: return launch3e<8>(p, ws, stream);
        case 9: return launch3e<9>(p, ws, stream);
        case 10: return launch3e<10>(p, ws, stream);
        default: return launch3e<4>(p, ws, stream);
    }
}
