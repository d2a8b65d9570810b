// patch_embed.hip -- the convolutional patch embeds of Data2Seq as ONE kernel: the patch gather ("im2col") happens in the operand
// stager of the bf16 NT GEMM, so the gathered matrix [tokens, Cin*kt*kh*kw] is never written or re-read.
//
//   Image    Data2Seq/Image.py:19-28                       Conv2d(3, C, k16, s16), flatten(2).transpose(1, 2)
//   Video    Video/models/modeling_finetune.py:263-297     Conv3d(3, C, k = s = (2, 16, 16))
//   Acoustic Data2Seq/Acoustic.py:16-22                    Conv2d(1, C, k16, stride 10): rows of x are only 4-byte aligned -> two passes
//
// How the gather fits the "g3" K-loop (gemm3_core.h) without touching it.  The loop streams 64-element K-tiles of the A operand with
// LDS-DMA: per lane ONE fixed byte offset (G3State::src) + a wave-uniform offset kt * kstep_a, through a buffer descriptor.  A patch
// row (c, dt, dy, 0..kw) is kw contiguous pixels of x; with kh * kw == 256 a (c, dt) PLANE of the patch is exactly four K-tiles of
// 64 / kw patch rows each, and inside a plane the K-tile offset IS affine: kt_local * (64 / kw) * W pixels.  So the K-loop runs plane
// by plane with one descriptor pair per plane (A: x shifted to the plane's (c, dt) image; B: the weight rows shifted by 256 columns),
// local K-tile numbers 0..3, and the lane offset = the patch origin of the lane's token + the lane's (dy, dx) inside the first K-tile.
// The stream runs across plane seams exactly as it runs across tile seams in the resident kernel: g3_ktile takes the source of the
// next and next-but-one K-tile as arguments.  Rows past the last token get an offset beyond every descriptor's range and read zeros.
//
// Fused when: x and W are bf16, kh * kw == 256 with kw in {8, 16, 32, 64}, 16-byte aligned patch rows (W, sw, H * W multiples of 8
// pixels), x smaller than 2 GiB, Cout % 8 == 0.  Anything else (fp32 pixels, the spectrogram's stride 10) goes through me_patchify +
// me_gemm inside the same entry point, with the caller's workspace for the gathered matrix.
#include "gemm3_core.h"

// gemm.hip: validation + parameter block of a descriptor; me_gemm for a g3 wgrad problem with the kernel launch replaced (planning,
// split-K slabs, the deterministic fold with alpha / beta / column sums stay me_gemm's); would the planner take that route
int gemm_fill_params(const me_gemm_desc* d, GemmParams& p);
int gemm_tn_with_launcher(const me_gemm_desc* d, hipStream_t stream, int (*launch)(const GemmParams&, hipStream_t, const void*), const void* ctx);
int gemm_tn_is_g3(const me_gemm_desc* d);

namespace {

// n / d for n < 2^31 as a multiplication (Granlund & Montgomery): q = mulhi(n, ceil(2^(31 + l) / d)) >> (l - 1), l = ceil(log2 d) >= 1
// (d == 1: mul = 0 and the quotient comes through `one`, so that the K-loop of the wgrad kernel stays free of branches)
struct FastDiv {
    uint32_t mul, shift, one;
};
FastDiv make_fastdiv(uint32_t d) {
    FastDiv f{0, 0, 0xffffffffu};
    if (d <= 1) return f;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    f.mul = (uint32_t)(((1ull << (31 + l)) + d - 1) / d);
    f.shift = l - 1;
    f.one = 0;
    return f;
}
__device__ __forceinline__ uint32_t fastdiv(uint32_t n, const FastDiv& f) { return (__umulhi(n, f.mul) >> f.shift) + (n & f.one); }

struct PeGeom {
    int Cin, T, H, W, kt, kw, st, sh, sw;
    int gh, gw;              // patches per column / row of a frame
    int tps, tpf;            // tokens per sample, per frame row of tubelets (gh * gw)
    int planes;              // Cin * kt
    int rpt;                 // patch rows per K-tile = 64 / kw
    int64_t plane_bytes;     // H * W * 2
    int64_t x_bytes;         // B * Cin * T * H * W * 2
    // the wgrad kernel recomputes patch origins every K-tile: divisions as multiplications, the four strides in bytes
    FastDiv d_tps, d_tpf, d_gw;
    uint32_t sb, st_b, sh_b, sw_b;      // bytes per sample / per st frames / per sh rows / per sw pixels
};

// byte offset of the first pixel of token row m's patch in plane (c = 0, dt = 0), or an offset out of every descriptor's range
__device__ __forceinline__ uint32_t pe_origin(const PeGeom& g, int64_t m, int64_t M) {
    if (m >= M) return 0x80000000u;
    const uint32_t mm = (uint32_t)m;
    const uint32_t b = mm / (uint32_t)g.tps, r = mm - b * (uint32_t)g.tps;
    const uint32_t pt = r / (uint32_t)g.tpf, r2 = r - pt * (uint32_t)g.tpf;
    const uint32_t py = r2 / (uint32_t)g.gw, px = r2 - py * (uint32_t)g.gw;
    const uint32_t frame = (b * (uint32_t)g.Cin) * (uint32_t)g.T + pt * (uint32_t)g.st;
    return ((frame * (uint32_t)g.H + py * (uint32_t)g.sh) * (uint32_t)g.W + px * (uint32_t)g.sw) * 2u;
}

// descriptors of plane number pl = c * kt + dt for the tile's weight rows [n0, n0 + 256)
__device__ __forceinline__ G3Src pe_plane_src(const GemmParams& p, const PeGeom& g, int c, int dt, int pl, int64_t n0) {
    G3Src s;
    const int64_t aoff = ((int64_t)c * g.T + dt) * g.plane_bytes;
    s.a = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.A)) + aoff, 0, (int)(g.x_bytes - aoff), 0x00020000);
    int64_t rb = p.N - n0;
    rb = rb < G3_BN ? rb : G3_BN;
    const int64_t boff = (int64_t)pl * 512;
    s.b = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.B)) + n0 * p.ldb * 2 + boff, 0,
                                            (int)((rb - 1) * p.ldb * 2 + p.K * 2 - boff), 0x00020000);
    return s;
}

// One 256 x 256 output tile per workgroup (the schedule of gemm_g3_kernel, gemm3.hip, without its K-split parts).
template <int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void patch_embed_g3_kernel(const GemmParams p, const PeGeom g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2;

    // block b runs on XCD b % 8: every XCD gets a contiguous range of tiles (neighbours share operand panels through its L2)
    const int F = p.tiles_m * p.tiles_n, bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int tile = xcd * (F >> 3) + (xcd < (F & 7) ? xcd : (F & 7)) + slot;
    const int tm = __builtin_amdgcn_readfirstlane(tile / p.tiles_n), tn = tile - tm * p.tiles_n;
    const int64_t m0 = (int64_t)tm * G3_BM, n0 = (int64_t)tn * G3_BN;

    G3State s;
    g3_init_lane(s, p, smem, wave, lane);
    // the A half-tiles' lane offsets: same (local row, slot) as g3_init_lane, the row's bytes come from the image
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rl = 16 * wave + 8 * i + (lane >> 3);
        const int c = (lane & 7) ^ ((rl >> 1) & 7);
        const int ax_row = (rl >> 6) * 128 + (rl & 63);
        const int e = 8 * c;                                         // first of the chunk's 8 elements inside the K-tile
        const uint32_t in_tile = (uint32_t)(((e / g.kw) * g.W + (e % g.kw)) * 2);
        s.src[1][i] = pe_origin(g, m0 + ax_row, p.M) + in_tile;
        s.src[3][i] = pe_origin(g, m0 + ax_row + 64, p.M) + in_tile;
    }
    s.kstep_a = g.rpt * g.W * 2;
    g3_zero(s);

    int c = 0, dt = 0;
    auto advance = [&]() { if (++dt == g.kt) { dt = 0; ++c; } };
    const G3Src null = g3_null_src(p);
    G3Src cur = pe_plane_src(p, g, 0, 0, 0, n0);
    advance();
    G3Src nxt = g.planes > 1 ? pe_plane_src(p, g, c, dt, 1, n0) : null;

    g3_issue<0>(s, cur, 0, 0); g3_issue<1>(s, cur, 0, 0); g3_issue<2>(s, cur, 0, 0); g3_issue<3>(s, cur, 0, 0);
    g3_issue<0>(s, cur, 1, 1); g3_issue<1>(s, cur, 1, 1); g3_issue<2>(s, cur, 1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();

    for (int pl = 0; pl < g.planes; ++pl) {
        g3_ktile<0>(s, cur, 1, cur, 2);
        g3_ktile<1>(s, cur, 2, cur, 3);
        g3_ktile<0>(s, cur, 3, nxt, 0);
        g3_ktile<1>(s, nxt, 0, nxt, 1);
        cur = nxt;
        advance();
        nxt = pl + 2 < g.planes ? pe_plane_src(p, g, c, dt, pl + 2, n0) : null;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wr == 0) __builtin_amdgcn_s_barrier();
    g3_epilogue<EPI>(p, s, m0, n0, lane);
}

template <int EPI> int launch_pe(const GemmParams& p, const PeGeom& g, hipStream_t stream) {
    static OncePerDevice once;
    if (once.need())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_embed_g3_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS);
    hipLaunchKernelGGL((patch_embed_g3_kernel<EPI>), dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(512), G3_LDS, stream, p, g);
    ME_CHECK_LAUNCH("me_patch_embed");
    return ME_OK;
}

// ---- weight gradient dW[Cout, Kp] = dY[tokens, Cout]^T . patches[tokens, Kp]: the split-K wgrad kernel (gemm_g3tn_kernel, gemm3.hip) with
// its B operand gathered.  The reduction runs over TOKENS here, so a lane's patch origin changes every K-tile (64 tokens further on) while
// its feature columns (8 consecutive pixels of one patch row of one plane) stay: lane offset = origin(token) + feature part, the
// wave-uniform K-tile offset of the B stream is zero.  Both B half-tiles a g3_ktile call issues belong to the same K-tile (its s1 / k1),
// so the origins are refreshed once per call.  Tokens past the end read zeros (offset out of the descriptor's range; x < 1 GiB).
constexpr uint32_t PE_OOB = 0x40000000u;
__device__ __forceinline__ uint32_t pe_origin_fast(const PeGeom& g, uint32_t tok, uint32_t ntok) {
    const uint32_t b = fastdiv(tok, g.d_tps), r = tok - b * (uint32_t)g.tps;
    const uint32_t pt = fastdiv(r, g.d_tpf), r2 = r - pt * (uint32_t)g.tpf;
    const uint32_t py = fastdiv(r2, g.d_gw), px = r2 - py * (uint32_t)g.gw;
    const uint32_t o = b * g.sb + pt * g.st_b + py * g.sh_b + px * g.sw_b;
    return tok < ntok ? o : PE_OOB;
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void patch_embed_wgrad_g3_kernel(const GemmParams p, const PeGeom g) {
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
    G3Src src = g3_make_src_tn(p, tm, tn);
    src.b = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, (int)g.x_bytes, 0x00020000);
    s.kstep_b = 0;
    // feature part of the lane's two B-X / B-Y chunks (same k-row t and slot as g3_init_lane_tn) and the k-rows themselves
    uint32_t feat[2][2], trow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int t = 8 * wave + 4 * i + (lane >> 4);
        const int c = (lane & 15) ^ g3_tn_swz(t);
        const int b_col = (c >> 2) * 64 + (c & 3) * 8;
        trow[i] = (uint32_t)t;
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int f = (int)n0 + b_col + 32 * y;                  // < Kp: Kp is a multiple of 256
            const int pl = f >> 8, in = f & 255;
            const int c_ = pl / g.kt, dt = pl - c_ * g.kt;
            feat[i][y] = (uint32_t)((((int64_t)c_ * g.T + dt) * g.plane_bytes) + ((in / g.kw) * g.W + (in % g.kw)) * 2);
        }
    }
    const uint32_t ntok = (uint32_t)p.K;
    auto set_b = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t o = pe_origin_fast(g, (uint32_t)kt * 64u + trow[i], ntok);
            s.src[0][i] = o + feat[i][0];
            s.src[2][i] = o + feat[i][1];
        }
    };
    const int kt0 = split * p.ksteps_per_split, kt1 = kt0 + p.ksteps_per_split;       // (an even count; tiles past K read zeros)

    set_b(kt0);
    g3_issue<0>(s, src, 0, kt0); g3_issue<1>(s, src, 0, kt0); g3_issue<2>(s, src, 0, kt0); g3_issue<3>(s, src, 0, kt0);
    set_b(kt0 + 1);
    g3_issue<0>(s, src, 1, kt0 + 1); g3_issue<1>(s, src, 1, kt0 + 1); g3_issue<2>(s, src, 1, kt0 + 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();
    const G3Src null = g3_null_src(p);
    // bias gradient: the N-tiles of one (M-tile, split) stage the same dY rows -- they share the column sums pair by pair of K-tiles
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
        set_b(kt + 2);
        g3_ktile<0, true>(s, src, kt + 1, src, kt + 2, c);
        set_b(kt + 3);
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

int launch_pe_wgrad(const GemmParams& p, hipStream_t stream, const void* ctx) {
    static OncePerDevice once;
    if (once.need())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_embed_wgrad_g3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS);
    const int nwg = p.tiles_m * p.tiles_n * p.split_k;
    hipLaunchKernelGGL(patch_embed_wgrad_g3_kernel, dim3((unsigned)nwg), dim3(512), G3_LDS, stream, p, *reinterpret_cast<const PeGeom*>(ctx));
    ME_CHECK_LAUNCH("me_patch_embed_wgrad");
    return ME_OK;
}

struct PeShape {
    int gt, gh, gw;
    int64_t tokens, M, K;
};

int pe_shape(const me_patch_embed_desc* d, PeShape& s) {
    ME_CHECK_ARG(d != nullptr, "me_patch_embed: null descriptor");
    ME_CHECK_ARG(d->B > 0 && d->Cin > 0 && d->T > 0 && d->H > 0 && d->W > 0 && d->kt > 0 && d->kh > 0 && d->kw > 0 && d->st > 0 && d->sh > 0 &&
                     d->sw > 0 && d->Cout > 0,
                 "me_patch_embed: bad geometry");
    ME_CHECK_ARG(d->T >= d->kt && d->H >= d->kh && d->W >= d->kw, "me_patch_embed: kernel larger than input");
    ME_CHECK_ARG(d->prefix_rows >= 0, "me_patch_embed: prefix_rows < 0");
    s.gt = (d->T - d->kt) / d->st + 1;
    s.gh = (d->H - d->kh) / d->sh + 1;
    s.gw = (d->W - d->kw) / d->sw + 1;
    s.tokens = (int64_t)s.gt * s.gh * s.gw;
    s.M = (int64_t)d->B * s.tokens;
    s.K = (int64_t)d->Cin * d->kt * d->kh * d->kw;
    return ME_OK;
}

bool pe_fusable(const me_patch_embed_desc* d, const PeShape& s) {
    if (d->x_dtype != ME_BF16 || d->w_dtype != ME_BF16) return false;
    if (d->kh * d->kw != 256 || !(d->kw == 8 || d->kw == 16 || d->kw == 32 || d->kw == 64)) return false;
    // patch rows are read as 16-byte chunks by LDS-DMA; the SOURCE only has to be dword-aligned (round 6: the spectrogram tokenizer's stride 10 --
    // Data2Seq/Acoustic.py:5-23, ast_models.py:86 -- puts a patch row at byte 20 tx: 4-byte aligned; chunks that straddle a 128-byte line cost a
    // second request, nothing else)
    if (d->W % 2 || d->sw % 2 || ((int64_t)d->H * d->W) % 2 || (uintptr_t)d->x % 16) return false;
    const int64_t x_bytes = (int64_t)d->B * d->Cin * d->T * d->H * d->W * 2;
    if (x_bytes >= (1ll << 31) || s.M >= (1ll << 31)) return false;
    if (d->Cout % 8 || 256 * s.K * 2 >= (1ll << 31)) return false;
    return true;
}

PeGeom pe_geom(const me_patch_embed_desc* d, const PeShape& s) {
    PeGeom g;
    g.Cin = d->Cin; g.T = d->T; g.H = d->H; g.W = d->W; g.kt = d->kt; g.kw = d->kw; g.st = d->st; g.sh = d->sh; g.sw = d->sw;
    g.gh = s.gh; g.gw = s.gw;
    g.tpf = s.gh * s.gw;
    g.tps = (int)s.tokens;
    g.planes = d->Cin * d->kt;
    g.rpt = 64 / d->kw;
    g.plane_bytes = (int64_t)d->H * d->W * 2;
    g.x_bytes = (int64_t)d->B * d->Cin * d->T * g.plane_bytes;
    g.d_tps = make_fastdiv((uint32_t)g.tps);
    g.d_tpf = make_fastdiv((uint32_t)g.tpf);
    g.d_gw = make_fastdiv((uint32_t)g.gw);
    g.sb = (uint32_t)((int64_t)d->Cin * d->T * g.plane_bytes);
    g.st_b = (uint32_t)((int64_t)d->st * g.plane_bytes);
    g.sh_b = (uint32_t)(d->sh * d->W * 2);
    g.sw_b = (uint32_t)(d->sw * 2);
    return g;
}

// the projection as a GEMM descriptor: A = the gathered matrix (or, fused, the image itself)
me_gemm_desc pe_gemm_desc(const me_patch_embed_desc* d, const PeShape& s, const void* a) {
    me_gemm_desc g{};
    g.op = ME_GEMM_NT;
    g.ab_dtype = d->w_dtype;
    g.M = s.M; g.N = d->Cout; g.K = s.K;
    g.A = a; g.lda = s.K;
    g.B = d->weight; g.ldb = s.K;
    g.C = d->out; g.ldc = d->ld_out; g.c_dtype = d->out_dtype;
    g.act = ME_ACT_NONE;
    g.alpha = 1.0f; g.beta = 0.0f;
    g.bias = d->bias;
    if (d->pos) {
        g.residual = d->pos; g.ldres = d->ld_pos; g.res_dtype = d->pos_dtype;
        g.res_row_mod = s.tokens;
    }
    if (d->prefix_rows) {
        g.out_group_rows = s.tokens;
        g.out_group_stride = s.tokens + d->prefix_rows;
        g.out_row_offset = d->prefix_rows;
    }
    return g;
}

}  // namespace

extern "C" int me_patch_embed_fused(const me_patch_embed_desc* d) {
    PeShape s;
    if (pe_shape(d, s) != ME_OK) return 0;
    return pe_fusable(d, s) ? 1 : 0;
}

extern "C" size_t me_patch_embed_workspace_bytes(const me_patch_embed_desc* d) {
    PeShape s;
    if (pe_shape(d, s) != ME_OK || pe_fusable(d, s)) return 0;
    const size_t cols = ((size_t)s.M * (size_t)s.K * (size_t)me_dtype_size(d->w_dtype) + 255) & ~(size_t)255;
    me_gemm_desc g = pe_gemm_desc(d, s, d->x);
    return cols + me_gemm_workspace_bytes(&g);
}

extern "C" int me_patch_embed(const me_patch_embed_desc* d, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    PeShape s;
    int rc = pe_shape(d, s);
    if (rc) return rc;
    ME_CHECK_ARG(d->x && d->weight && d->out, "me_patch_embed: null pointer");
    ME_CHECK_ARG(me_dtype_ok(d->x_dtype) && me_dtype_ok(d->w_dtype) && me_dtype_ok(d->out_dtype), "me_patch_embed: bad dtype");
    if (!pe_fusable(d, s)) {
        // two passes: gather into the workspace (converted to the compute dtype on the way), then the ordinary GEMM
        const size_t cols_bytes = ((size_t)s.M * (size_t)s.K * (size_t)me_dtype_size(d->w_dtype) + 255) & ~(size_t)255;
        ME_CHECK_ARG(d->workspace && (size_t)d->workspace_bytes >= cols_bytes,
                     "me_patch_embed: this geometry / dtype needs a workspace of me_patch_embed_workspace_bytes()");
        rc = me_patchify(d->x, d->x_dtype, d->workspace, d->w_dtype, d->B, d->Cin, d->T, d->H, d->W, d->kt, d->kh, d->kw, d->st, d->sh, d->sw, stream_);
        if (rc) return rc;
        me_gemm_desc g = pe_gemm_desc(d, s, d->workspace);
        g.workspace = reinterpret_cast<char*>(d->workspace) + cols_bytes;
        g.workspace_bytes = d->workspace_bytes - (int64_t)cols_bytes;
        return me_gemm(&g, stream_);
    }
    ProfScope prof(ME_GEMM_NT, ME_BF16, s.M, d->Cout, s.K, stream);
    prof.plan = 4;
    me_gemm_desc gd = pe_gemm_desc(d, s, d->x);
    GemmParams p;
    rc = gemm_fill_params(&gd, p);
    if (rc) return rc;
    p.tiles_m = (int)((s.M + 255) / 256);
    p.tiles_n = (int)((d->Cout + 255) / 256);
    const PeGeom g = pe_geom(d, s);
    return pick_epi(p) == 0 ? launch_pe<0>(p, g, stream) : launch_pe<4>(p, g, stream);
}

// ---- parameter gradients
namespace {
// dW = dY^T . patches as a TN descriptor: A = dY [tokens, Cout], B = the gathered matrix (or, fused, the image)
me_gemm_desc pe_wgrad_desc(const me_patch_embed_desc* d, const PeShape& s, const void* dy, int64_t ld_dy, const void* b, void* dw, int dw_dtype,
                           float* dbias, float beta) {
    me_gemm_desc g{};
    g.op = ME_GEMM_TN;
    g.ab_dtype = d->w_dtype;
    g.M = d->Cout; g.N = s.K; g.K = s.M;
    g.A = dy; g.lda = ld_dy;
    g.B = b; g.ldb = s.K;
    g.C = dw; g.ldc = s.K; g.c_dtype = dw_dtype;
    g.act = ME_ACT_NONE;
    g.alpha = 1.0f; g.beta = beta;
    g.colsum_a = dbias;
    return g;
}
bool pe_wgrad_fusable(const me_patch_embed_desc* d, const PeShape& s, const me_gemm_desc& g) {
    if (!pe_fusable(d, s)) return false;
    if ((int64_t)d->B * d->Cin * d->T * d->H * d->W * 2 >= (1ll << 30)) return false;       // (PE_OOB must lie outside x)
    return gemm_tn_is_g3(&g) != 0;
}
size_t pe_cols_bytes(const me_patch_embed_desc* d, const PeShape& s) {
    return ((size_t)s.M * (size_t)s.K * (size_t)me_dtype_size(d->w_dtype) + 255) & ~(size_t)255;
}
}  // namespace

extern "C" int me_patch_embed_wgrad_fused(const me_patch_embed_desc* d, int dw_dtype) {
    PeShape s;
    if (pe_shape(d, s) != ME_OK || !d->x) return 0;
    me_gemm_desc g = pe_wgrad_desc(d, s, d->x, d->Cout, d->x, const_cast<void*>(d->x), dw_dtype, nullptr, 0.0f);
    return pe_wgrad_fusable(d, s, g) ? 1 : 0;
}

extern "C" size_t me_patch_embed_wgrad_workspace_bytes(const me_patch_embed_desc* d, int dw_dtype, int with_bias) {
    PeShape s;
    if (pe_shape(d, s) != ME_OK) return 0;
    float dummy_b = 0.f;
    // (pointers only have to be non-null and aligned for planning; the column sums have room in the GEMM's workspace when asked for)
    me_gemm_desc g = pe_wgrad_desc(d, s, d->x, d->Cout, d->x, const_cast<void*>(d->x), dw_dtype, with_bias ? &dummy_b : nullptr, 0.0f);
    const bool fused = pe_wgrad_fusable(d, s, g);
    size_t n = fused ? 0 : pe_cols_bytes(d, s);
    n += (me_gemm_workspace_bytes(&g) + 255) & ~(size_t)255;
    if (with_bias) n += me_colsum_workspace(d->Cout);          // (used only when the wgrad kernel cannot carry the column sums)
    return n;
}

extern "C" int me_patch_embed_wgrad(const me_patch_embed_desc* d, const void* dy, int64_t ld_dy, void* dw, int dw_dtype, float* dbias,
                                    float beta, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    PeShape s;
    int rc = pe_shape(d, s);
    if (rc) return rc;
    ME_CHECK_ARG(d->x && dy && dw, "me_patch_embed_wgrad: null pointer");
    ME_CHECK_ARG(me_dtype_ok(d->x_dtype) && me_dtype_ok(d->w_dtype) && me_dtype_ok(dw_dtype), "me_patch_embed_wgrad: bad dtype");
    ME_CHECK_ARG(d->workspace && (size_t)d->workspace_bytes >= me_patch_embed_wgrad_workspace_bytes(d, dw_dtype, dbias != nullptr),
                 "me_patch_embed_wgrad: workspace of me_patch_embed_wgrad_workspace_bytes() required");
    //  the column sums follow C's beta only for beta = 0 / 1 (me_colsum accumulates or overwrites)
    ME_CHECK_ARG(beta == 0.0f || beta == 1.0f || !dbias, "me_patch_embed_wgrad: beta must be 0 or 1 when dbias is wanted");
    char* ws = reinterpret_cast<char*>(d->workspace);
    me_gemm_desc g = pe_wgrad_desc(d, s, dy, ld_dy, d->x, dw, dw_dtype, dbias, beta);
    const bool fused = pe_wgrad_fusable(d, s, g);
    {   //  the workspace query plans with dense placeholder operands; a strided / misaligned dy can send THIS call down the
        // two-pass route, whose gathered columns the query did not count -- size the real route and refuse instead of writing past the end
        size_t need = (fused ? 0 : pe_cols_bytes(d, s)) + ((me_gemm_workspace_bytes(&g) + 255) & ~(size_t)255);
        if (dbias) need += me_colsum_workspace(d->Cout);
        if ((size_t)d->workspace_bytes < need) {
            me_set_error("me_patch_embed_wgrad: workspace too small for this dy (strided / misaligned dy takes the two-pass route: %zu bytes needed, %lld given)",
                         need, (long long)d->workspace_bytes);
            return ME_ERR_WORKSPACE;
        }
    }
    if (!fused) {
        rc = me_patchify(d->x, d->x_dtype, ws, d->w_dtype, d->B, d->Cin, d->T, d->H, d->W, d->kt, d->kh, d->kw, d->st, d->sh, d->sw, stream_);
        if (rc) return rc;
        g.B = ws;
        ws += pe_cols_bytes(d, s);
    }
    g.workspace = ws;
    g.workspace_bytes = (int64_t)((me_gemm_workspace_bytes(&g) + 255) & ~(size_t)255);      // (sized with the column sums in)
    ws += g.workspace_bytes;
    const bool cs_fused = dbias && me_gemm_fuses_colsum(&g);
    if (!cs_fused) g.colsum_a = nullptr;
    if (fused) {
        const PeGeom geom = pe_geom(d, s);
        rc = gemm_tn_with_launcher(&g, stream, launch_pe_wgrad, &geom);
    } else {
        rc = me_gemm(&g, stream_);
    }
    if (rc) return rc;
    if (dbias && !cs_fused) return me_colsum(dy, d->w_dtype, ld_dy, s.M, d->Cout, dbias, beta != 0.0f ? 1 : 0, ws, stream_);
    return ME_OK;
}
