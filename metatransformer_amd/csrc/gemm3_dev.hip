// gemm3_dev.hip -- DEV BUILD ONLY (python -m metatransformer_amd.build --dev, -DME_DEV; empty in libmetaenc.so): the
// persistent stream-K form of the g3 family that tools/gemm_dev measures against the shipped kernels ("g3p").  Correct and
// tested there, not faster than the resident kernel, so it is not part of the product.
#ifdef ME_DEV
#include "gemm3_core.h"

namespace {
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
    g3_walk_init(wc, pl, v, G, r0, r1, (p.debug & 2) != 0);
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
            if (p.debug & 1) {                    // dev: K-loops only
                float keep = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) keep += s.acc[i][j][0] + s.acc[i][j][1] + s.acc[i][j][2] + s.acc[i][j][3];
                if (keep == 1.2345e-30f) reinterpret_cast<float*>(p.C)[0] = keep;
                skip = true;
            }
            if ((p.debug & 8) && (wc.seg_begin != 0 || wc.seg_end < pl.hk)) skip = true;     // dev: no hand-over at all
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

template <int EPI> int launch3p(const GemmParams& p, void* ws, hipStream_t stream) {
    static OncePerDevice once;
    if (once.need())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_g3p_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS);
    const int tiles = p.tiles_m * p.tiles_n;
    {
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
    return ME_OK;
}

}  // namespace

int launch_g3p(const GemmParams& p, int epi, void* ws, hipStream_t stream) {
    switch (epi) {
        case 0: return launch3p<0>(p, ws, stream);
        case 1: return launch3p<1>(p, ws, stream);
        case 2: return launch3p<2>(p, ws, stream);
        case 3: return launch3p<3>(p, ws, stream);
        default: return launch3p<4>(p, ws, stream);
    }
}
#endif  // ME_DEV
