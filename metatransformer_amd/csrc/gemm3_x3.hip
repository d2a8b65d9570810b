// gemm3_x3.hip -- the weight gradient of an ME_BF16X3 (fp32-accurate) Linear as ONE launch of the split-K wgrad kernel.
//
// dW = dOut^T In with both operands held as bf16 planes per row -- dOut3 [tokens, 3 n_out] = [hi | lo | hi], In3 [tokens, 3 n_in] =
// [hi | lo | hi] -- is the sum of three ordinary TN products, (hi, hi) + (lo, hi) + (hi, lo) (block.hip: block_bwd_x3).  As three me_gemm
// calls that is three launches and three fp32 slab folds per weight (144 + 144 per training step of a Base encoder).  Here the three
// products are three SEGMENTS of one reduction: a workgroup walks its token range once per segment with the operand descriptors moved
// to the segment's planes (A columns + n_out for the lo plane of dOut, B columns + n_in for the lo plane of In), the accumulators and
// the LDS-DMA stream running straight through the seams (g3_ktile takes the next two K-tiles' sources).  One launch, one fold; the
// sums are the same fp32 sums in a different order.  The bias gradient colsum(dOut) = colsum(hi) + colsum(lo) comes from segments 0
// and 1 of the same launch (the fused column sums of the A operand).  Planning, slabs, fold, alpha / beta: me_gemm's
// (gemm_tn_with_launcher, gemm.hip).
#include "gemm3_core.h"

int gemm_tn_with_launcher(const me_gemm_desc* d, hipStream_t stream, int (*launch)(const GemmParams&, hipStream_t, const void*), const void* ctx);

namespace {

// operand columns [m0 + pa, ..) of A and [n0 + pb, ..) of B, all K rows (g3_make_src_tn with a plane offset)
__device__ __forceinline__ G3Src x3_seg_src(const GemmParams& p, int64_t m0, int64_t n0, int seg) {
    const int64_t ca = m0 + (seg == 1 ? p.M : 0), cb = n0 + (seg == 2 ? p.N : 0);
    G3Src s;
    s.a = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.A)) + ca * 2, 0, (int)(p.K * p.lda * 2 - ca * 2), 0x00020000);
    s.b = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.B)) + cb * 2, 0, (int)(p.K * p.ldb * 2 - cb * 2), 0x00020000);
    return s;
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_g3tn_x3_kernel(const GemmParams p) {
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
    const int n = p.ksteps_per_split;                       // K-tiles per segment (an even count; tiles past K read zeros)
    const int kt0 = split * n;
    const G3Src null = g3_null_src(p);
    G3Src cur = x3_seg_src(p, m0, n0, 0);

    g3_issue<0>(s, cur, 0, kt0); g3_issue<1>(s, cur, 0, kt0); g3_issue<2>(s, cur, 0, kt0); g3_issue<3>(s, cur, 0, kt0);
    g3_issue<0>(s, cur, 1, kt0 + 1); g3_issue<1>(s, cur, 1, kt0 + 1); g3_issue<2>(s, cur, 1, kt0 + 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();
    // bias gradient: the N-tiles of one (M-tile, split) stage the same dOut rows -- they share the column sums pair by pair of K-tiles,
    // round-robin; segment 2 stages the hi plane of dOut a second time and is left out
    const bool do_cs = p.colsum_ws != nullptr;
    s.cs[0] = s.cs[1] = 0.f;
    int cs_turn = do_cs ? tn : -1;
    auto my_turn = [&](int seg) {
        if (!do_cs || seg == 2) return false;
        const bool mine = cs_turn == 0;
        cs_turn = mine ? p.tiles_n - 1 : cs_turn - 1;
        return mine;
    };
    // pairs of K-tiles: (seg, k) is the pair being multiplied, the pair after it may lie in the next segment
    for (int seg = 0; seg < 3; ++seg) {
        for (int k = 0; k < n; k += 2) {
            const bool last_of_seg = k + 2 >= n;
            const int nseg = last_of_seg ? seg + 1 : seg;
            const int nk = last_of_seg ? 0 : k + 2;
            const G3Src nxt = nseg == seg ? cur : (nseg < 3 ? x3_seg_src(p, m0, n0, nseg) : null);
            const bool c = my_turn(seg);
            g3_ktile<0, true>(s, cur, kt0 + k + 1, nxt, kt0 + nk, c);
            g3_ktile<1, true>(s, nxt, kt0 + nk, nxt, kt0 + nk + 1, c);
            cur = nxt;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wr == 0) __builtin_amdgcn_s_barrier();
    if (do_cs) {
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
    g3_epilogue<5>(p, s, m0, n0, lane, reinterpret_cast<float*>(p.C) + (int64_t)split * p.slab_stride, 0);
}

int launch_g3tn_x3(const GemmParams& p, hipStream_t stream, const void*) {
    static OncePerDevice once;
    if (once.need())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_g3tn_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS);
    const int nwg = p.tiles_m * p.tiles_n * p.split_k;
    hipLaunchKernelGGL(gemm_g3tn_x3_kernel, dim3((unsigned)nwg), dim3(512), G3_LDS, stream, p);
    ME_CHECK_LAUNCH("me_gemm(g3 tn, three planes)");
    return ME_OK;
}

}  // namespace

// dW[n_out, n_in] (beta * dW +) = dOut^T In from the planes of both, dbias (optional; follows beta) = colsum(dOut): one launch + one
// fold.  d: an ME_GEMM_TN descriptor over the PLANE matrices -- A = dOut3 (lda = 3 M), B = In3 (ldb = 3 N), K = tokens, colsum_a = dbias.
// ME_ERR_UNSUPPORTED when the planner does not give the problem to the g3 wgrad family (the caller then runs the three products).
int gemm_tn_x3_planes(const me_gemm_desc* d, hipStream_t stream) {
    // (only the hi and lo planes of either operand are read -- offsets 0 and M / N -- so ME_BF16X2 rows, ld >= 2 M / 2 N, serve as well as ME_BF16X3 ones)
    if (!d || d->op != ME_GEMM_TN || d->ab_dtype != ME_BF16 || d->lda < 2 * d->M || d->ldb < 2 * d->N) return ME_ERR_UNSUPPORTED;
    // (the lo planes end 2 M / 2 N columns further right than the planner's bounds check assumes)
    if (d->K * d->lda * 2 >= (1ll << 31) || d->K * d->ldb * 2 >= (1ll << 31)) return ME_ERR_UNSUPPORTED;
    return gemm_tn_with_launcher(d, stream, launch_g3tn_x3, nullptr);
}
