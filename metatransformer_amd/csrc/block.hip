// block.hip -- one encoder Block (forward / backward) composed on the C side from the library's own entry points.
// Restates Block.forward, PointCloud/openpoints/models/layers/attention.py:55-58 (and what autograd derives from it):
// no new arithmetic here, only the launch sequence, the activation stash and the scratch carving, so that a host makes
// ONE call per block and direction and the GPU never waits for the host between the ~10 (forward) / ~20 (backward) kernels.
#include "common.h"
#include <string.h>
#include <stdlib.h>
#include <atomic>
#include <map>
#include <mutex>

int gemm_tn_x3_planes(const me_gemm_desc* d, hipStream_t stream);      // gemm3_x3.hip

namespace {

// ---- me_block_bwd: the weight-gradient GEMMs (and their folds) on a SIDE stream.  Nothing downstream of a Block's backward needs
// dW before the call returns, while the chain dY -> dX is serial and a third of its kernels are not matrix-bound (two LayerNorm
// backward passes and the attention backward per block: ~300 of ~2 500 us): the side stream's wgrad workgroups take the CUs those
// kernels and the tails of the resident launches leave idle (train step -2.2 % same box; bit-identical gradients).  One side stream + fork / join events per main stream
// and device, created at first use; the call joins before it returns (the caller sees ordinary stream semantics, and the shared
// backward workspace may be reused by the next call).  Off: me_block_bwd_overlap(0); always off while the
// main stream is being captured into a hipGraph.
struct SideCtx {
    hipStream_t side = nullptr;
    hipEvent_t fork[4] = {nullptr, nullptr, nullptr, nullptr}, join = nullptr;
    bool failed = false;
};
std::atomic<int> g_wgrad_overlap{1};       // me_block_bwd_overlap(); the library itself reads no environment (the Python package maps
                                           // ME_WGRAD_OVERLAP=0 onto that call at import, for the profiling scripts)
bool wgrad_overlap_on() { return g_wgrad_overlap.load() != 0; }
constexpr size_t kMaxSideCtx = 64;         // (device, caller stream) pairs that get a side stream; later ones run in serial order
SideCtx* side_ctx(hipStream_t main) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, SideCtx> ctxs;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(main, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;
    int dev = 0;
    if (hipStreamGetDevice(main, &dev) != hipSuccess) (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    // one side stream + five events per (device, caller stream), kept for the life of the process: bounded, so that a host that
    // creates streams without end does not leak through here
    if (ctxs.size() >= kMaxSideCtx && ctxs.find(std::make_pair(dev, main)) == ctxs.end()) return nullptr;
    SideCtx& c = ctxs[std::make_pair(dev, main)];
    if (c.failed) return nullptr;
    if (!c.side) {
        // streams and events belong to the device that is current when they are created: make that the caller stream's device
        int cur = dev;
        (void)hipGetDevice(&cur);
        if (cur != dev && hipSetDevice(dev) != hipSuccess) { c.failed = true; return nullptr; }
        bool ok = hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking) == hipSuccess;      // (a higher / lower stream priority measured no different)
        for (int i = 0; ok && i < 4; ++i) ok = hipEventCreateWithFlags(&c.fork[i], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&c.join, hipEventDisableTiming) == hipSuccess;
        if (cur != dev) (void)hipSetDevice(cur);
        if (!ok) { c.failed = true; c.side = nullptr; return nullptr; }
    }
    return &c;
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Dims {
    int64_t M;
    int C, C3, Hd, hd;
    size_t esz, rsz;
};
int get_dims(const me_block_desc* d, Dims& s, const char* fn) {
    ME_CHECK_ARG(d != nullptr, "%s: null descriptor", fn);
    ME_CHECK_ARG((me_dtype_ok(d->dtype) || d->dtype == ME_BF16X3) && me_dtype_ok(d->res_dtype), "%s: bad dtype", fn);
    ME_CHECK_ARG(d->B > 0 && d->N > 0 && d->C > 0 && d->heads > 0 && d->hidden > 0 && d->C % d->heads == 0,
                 "%s: bad shape B=%d N=%d C=%d heads=%d hidden=%d", fn, d->B, d->N, d->C, d->heads, d->hidden);
    s.M = (int64_t)d->B * d->N;
    s.C = d->C; s.C3 = 3 * d->C; s.Hd = d->hidden; s.hd = d->C / d->heads;
    s.esz = d->dtype == ME_BF16X3 ? 6 : me_dtype_size(d->dtype); s.rsz = me_dtype_size(d->res_dtype);
    return ME_OK;
}

// ---- activation stash layout
struct Saved {
    char *xn1, *qkv, *o, *x1, *xn2, *hpre, *a;
    float *mean1, *rstd1, *mean2, *rstd2, *lse;
    size_t bytes;
};
Saved carve_saved(const me_block_desc* d, const Dims& s, void* base) {
    Saved v;
    size_t off = 0;
    char* b = reinterpret_cast<char*>(base);
    auto take = [&](size_t n) { char* p = b + off; off += align256(n); return p; };
    v.xn1 = take(s.M * s.C * s.esz);
    v.qkv = take(s.M * s.C3 * s.esz);
    v.o = take(s.M * s.C * s.esz);
    v.x1 = take(s.M * s.C * s.rsz);
    v.xn2 = take(s.M * s.C * s.esz);
    v.hpre = take(s.M * s.Hd * s.esz);
    v.a = take(s.M * s.Hd * s.esz);
    v.mean1 = reinterpret_cast<float*>(take(s.M * 4));
    v.rstd1 = reinterpret_cast<float*>(take(s.M * 4));
    v.mean2 = reinterpret_cast<float*>(take(s.M * 4));
    v.rstd2 = reinterpret_cast<float*>(take(s.M * 4));
    v.lse = reinterpret_cast<float*>(take((size_t)d->B * d->heads * d->N * 4));
    v.bytes = off;
    return v;
}

void gemm_desc(me_gemm_desc& g, int op, int dt, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
               int64_t ldb, void* C, int64_t ldc, int cdt) {
    memset(&g, 0, sizeof(g));
    g.op = op; g.ab_dtype = dt; g.M = M; g.N = N; g.K = K;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.c_dtype = cdt;
    g.alpha = 1.0f;
}

// the largest GEMM scratch any of the block's GEMMs asks for (tail split forward, split-K wgrad backward)
size_t gemm_scratch(const me_block_desc* d, const Dims& s, bool backward) {
    size_t w = 0;
    me_gemm_desc g;
    auto probe = [&](int op, int64_t M, int64_t N, int64_t K, int cdt, bool cs) {
        static char dummy_mem[64] __attribute__((aligned(64)));
        gemm_desc(g, op, d->dtype, M, N, K, dummy_mem, op == ME_GEMM_NT ? K : M, dummy_mem, op == ME_GEMM_NT ? K : N, dummy_mem, N, cdt);
        if (cs) g.colsum_a = reinterpret_cast<float*>(dummy_mem);
        const size_t b = me_gemm_workspace_bytes(&g);
        if (b > w) w = b;
    };
    probe(ME_GEMM_NT, s.M, s.C3, s.C, d->dtype, false);
    probe(ME_GEMM_NT, s.M, s.C, s.C, d->res_dtype, false);
    probe(ME_GEMM_NT, s.M, s.Hd, s.C, d->dtype, false);
    probe(ME_GEMM_NT, s.M, s.C, s.Hd, d->res_dtype, false);
    if (backward) {
        probe(ME_GEMM_NT, s.M, s.C, s.C3, d->dtype, false);
        probe(ME_GEMM_NT, s.M, s.C, s.Hd, d->dtype, false);
        probe(ME_GEMM_TN, s.C3, s.C, s.M, ME_F32, true);
        probe(ME_GEMM_TN, s.C, s.C, s.M, ME_F32, true);
        probe(ME_GEMM_TN, s.Hd, s.C, s.M, ME_F32, true);
        probe(ME_GEMM_TN, s.C, s.Hd, s.M, ME_F32, true);
    }
    return align256(w);
}

size_t aux_scratch(const Dims& s) {      // LayerNorm-backward and column-sum scratch (used one after the other)
    size_t a = me_layernorm_bwd_workspace(s.C), b = me_colsum_workspace(s.Hd > s.C3 ? s.Hd : s.C3);
    return align256(a > b ? a : b);
}

// ---- folded inference: LayerNorm statistics taken from the residual GEMMs' epilogues (me_gemm_desc.row_stats)
bool wants_fold(const me_block_desc* d) {
    return d->dtype != ME_BF16X3 && d->qkv_wf && d->fc1_wf && d->qkv_s && d->qkv_c && d->fc1_s && d->fc1_c && d->res_dtype == d->dtype;
}
// scratch behind the inference intermediates: the [C / 64][M] partials (shared by proj and fc2) + two [M][2] pair buffers that
// me_encoder_fwd hands from block to block
// (round 6: + a second partials buffer, the one me_encoder_fwd hands from block to block as y_parts / x_parts; layout
//  [partials proj -> fc1][partials fc2 -> next qkv][pairs][pairs])
size_t stats_scratch(const Dims& s) { return 2 * align256(me_row_stats_partial_bytes(s.M, s.C)) + 2 * align256((size_t)s.M * 8); }
// can the proj / fc2 launches of this block emit statistics?  (both have the same M, N = C; K differs -- ask for each)
bool emits_stats(const me_block_desc* d, const Dims& s) {
    if (!wants_fold(d) || d->gamma1 || d->gamma2 || s.C % ME_STATS_GROUP) return false;
    static char dummy_mem[64] __attribute__((aligned(64)));
    me_gemm_desc g;
    for (int64_t K : {(int64_t)s.C, (int64_t)s.Hd}) {
        gemm_desc(g, ME_GEMM_NT, d->dtype, s.M, s.C, K, dummy_mem, K, dummy_mem, K, dummy_mem, s.C, d->res_dtype);
        g.bias = reinterpret_cast<const float*>(dummy_mem);
        g.residual = dummy_mem; g.ldres = s.C; g.res_dtype = d->res_dtype;
        g.workspace = dummy_mem; g.workspace_bytes = (int64_t)1 << 40;      // (as me_block_fwd calls it: with a workspace)
        if (!me_gemm_emits_row_stats(&g)) return false;
    }
    return true;
}

// can the qkv / fc1 launches of this (folded) block take the partials directly (me_gemm_desc.row_parts: no combine launch)?
bool takes_parts(const me_block_desc* d, const Dims& s) {
    if (!wants_fold(d) || s.C % ME_STATS_GROUP || s.C > 4 * ME_STATS_GROUP) return false;
    static char dummy_mem[64] __attribute__((aligned(64)));
    me_gemm_desc g;
    for (int64_t N : {(int64_t)s.C3, (int64_t)s.Hd}) {
        gemm_desc(g, ME_GEMM_NT, d->dtype, s.M, N, s.C, dummy_mem, s.C, dummy_mem, s.C, dummy_mem, N, d->dtype);
        g.bias = reinterpret_cast<const float*>(dummy_mem);
        g.col_shift = reinterpret_cast<const float*>(dummy_mem);
        g.row_parts = reinterpret_cast<const float*>(dummy_mem); g.row_nparts = (int32_t)(s.C / ME_STATS_GROUP); g.row_eps = d->eps;
        if (N == s.Hd) g.act = ME_ACT_GELU;
        g.workspace = dummy_mem; g.workspace_bytes = (int64_t)1 << 40;
        if (!me_gemm_takes_row_parts(&g)) return false;
    }
    return true;
}

// does a training bf16 Block keep gelu'(h) in eight bits (ME_GG8)?  When BOTH launches that touch it -- the fc1 forward that saves it, the fc2
// dgrad that multiplies by it -- run on the resident kernel.  Same rule going forward and back (the stash slot keeps its bf16 size).
// OFF by default (-DME_BLOCK_GG8=1 turns it on): same-box A/B at config 2, profiles/r06_gelu_grad_8bit_ab.txt -- the step does not get
// faster for the 620 MB less it moves per layer (fc1 forward 289 -> 296 us with the extra pack arithmetic, fc2 dgrad 244.6 -> 243.9 us),
// so the bf16 factor's precision stays.  me_gemm serves ME_GG8 either way.
#ifndef ME_BLOCK_GG8
#define ME_BLOCK_GG8 0
#endif
bool gelu_grad_gg8(const me_block_desc* d, const Dims& s) {
    if (!ME_BLOCK_GG8 || d->dtype != ME_BF16 || s.Hd % 8) return false;
    static char dummy_mem[64] __attribute__((aligned(64)));
    me_gemm_desc g;
    gemm_desc(g, ME_GEMM_NT, ME_BF16, s.M, s.Hd, s.C, dummy_mem, s.C, dummy_mem, s.C, dummy_mem, s.Hd, ME_BF16);
    g.bias = reinterpret_cast<const float*>(dummy_mem); g.act = ME_ACT_GELU;
    g.preact = dummy_mem; g.ldpre = s.Hd; g.preact_dtype = ME_GG8; g.flags = ME_GEMM_SAVE_GELU_GRAD;
    g.workspace = dummy_mem; g.workspace_bytes = (int64_t)1 << 40;
    if (!me_gemm_takes_gg8(&g)) return false;
    gemm_desc(g, ME_GEMM_NT, ME_BF16, s.M, s.Hd, s.C, dummy_mem, s.C, dummy_mem, s.C, dummy_mem, s.Hd, ME_BF16);
    g.aux = dummy_mem; g.ldaux = s.Hd; g.aux_dtype = ME_GG8; g.flags = ME_GEMM_AUX_IS_FACTOR;
    g.workspace = dummy_mem; g.workspace_bytes = (int64_t)1 << 40;
    return me_gemm_takes_gg8(&g) != 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// ME_BF16X3 blocks: the reference's DEFAULT arithmetic (fp32 tokens, fp32 weights: README.md:113-150) at matrix-pipe speed.  Every
// Linear runs as ONE bf16 NT GEMM over three-plane operands (include/metaenc.h, ME_BF16X3: A = [hi | lo | hi], W = [hi | hi | lo],
// reduction length 3 K), i.e. A_hi W_hi + A_lo W_hi + A_hi W_lo with fp32 accumulation: ~2^-17 relative, against 2^-9 for plain
// bf16 operands and at 3x the bf16 flops instead of the 16x of the exact-fp32 MFMA.  The producers write the three planes
// themselves (LayerNorm, the fc1 epilogue) or one split pass follows them (attention output, incoming gradients); LayerNorm,
// softmax, GELU (erf form), residual stream and every accumulator stay fp32; attention runs on the exact-fp32 kernels.
// Weight gradients: dW = dY^T X as three TN launches on the planes (hi,hi) + (lo,hi) + (hi,lo), accumulated by beta = 1.
// The two [tokens, hidden] tensors of the MLP -- gelu(fc1) going forward, dL/dh going back -- as ME_BF16X2 planes [hi | lo] instead of
// ME_BF16X3's [hi | lo | hi] (round 6): a third fewer bytes written by the two launches that are bound by their output (EPI 9 / 10), and
// saved for backward.  Needs the NT GEMMs that READ them as their A operand to wrap A's reduction index (me_gemm_desc.a_wrap_k): the
// one-tile 256 x 256 family -- i.e. from ~128 tiles on; below that the three-plane form stays.
bool x3_two_planes(const Dims& s) {
    static char dummy_mem[64] __attribute__((aligned(64)));
    me_gemm_desc g;
    for (int res = 0; res < 2; ++res) {      // fc2 forward (fp32 residual epilogue) and the fc1 dgrad (plain fp32 output)
        gemm_desc(g, ME_GEMM_NT, ME_BF16, s.M, s.C, 3 * s.Hd, dummy_mem, 2 * s.Hd, dummy_mem, 3 * s.Hd, dummy_mem, s.C, ME_F32);
        g.a_wrap_k = 2 * s.Hd;
        if (res) { g.bias = reinterpret_cast<const float*>(dummy_mem); g.residual = dummy_mem; g.ldres = s.C; g.res_dtype = ME_F32; }
        g.workspace = dummy_mem; g.workspace_bytes = (int64_t)1 << 40;
        if (!me_gemm_takes_a_wrap(&g)) return false;
    }
    return true;
}

struct SavedX3 {
    char *xn1, *qkv, *o, *o3, *x1, *xn2, *hpre, *a;      // xn1 / o3 / xn2 [M, 3C] bf16, a [M, 3 Hd] bf16; qkv / o / x1 / hpre (= gelu') fp32
    float *mean1, *rstd1, *mean2, *rstd2, *lse;
    size_t bytes;
};
SavedX3 carve_saved_x3(const me_block_desc* d, const Dims& s, void* base) {
    SavedX3 v;
    size_t off = 0;
    char* b = reinterpret_cast<char*>(base);
    auto take = [&](size_t n) { char* p = b + off; off += align256(n); return p; };
    v.xn1 = take(s.M * s.C * 6);
    v.qkv = take(s.M * s.C3 * 4);
    v.o = take(s.M * s.C * 4);
    v.o3 = take(s.M * s.C * 6);
    v.x1 = take(s.M * s.C * 4);
    v.xn2 = take(s.M * s.C * 6);
    v.hpre = take(s.M * s.Hd * 4);
    v.a = take(s.M * s.Hd * (x3_two_planes(s) ? 4 : 6));
    v.mean1 = reinterpret_cast<float*>(take(s.M * 4));
    v.rstd1 = reinterpret_cast<float*>(take(s.M * 4));
    v.mean2 = reinterpret_cast<float*>(take(s.M * 4));
    v.rstd2 = reinterpret_cast<float*>(take(s.M * 4));
    v.lse = reinterpret_cast<float*>(take((size_t)d->B * d->heads * d->N * 4));
    v.bytes = off;
    return v;
}
size_t gemm_scratch_x3(const Dims& s, bool backward) {
    size_t w = 0;
    me_gemm_desc g;
    static char dummy_mem[64] __attribute__((aligned(64)));
    auto probe = [&](int op, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int cdt) {
        gemm_desc(g, op, ME_BF16, M, N, K, dummy_mem, lda, dummy_mem, ldb, dummy_mem, cdt == ME_BF16X3 ? 3 * N : N, cdt);
        if (op == ME_GEMM_TN) g.colsum_a = reinterpret_cast<float*>(dummy_mem);      // (the fused bias gradient's partial rows)
        const size_t b = me_gemm_workspace_bytes(&g);
        if (b > w) w = b;
    };
    probe(ME_GEMM_NT, s.M, s.C3, 3 * s.C, 3 * s.C, 3 * s.C, ME_F32);
    probe(ME_GEMM_NT, s.M, s.C, 3 * s.C, 3 * s.C, 3 * s.C, ME_F32);
    probe(ME_GEMM_NT, s.M, s.Hd, 3 * s.C, 3 * s.C, 3 * s.C, ME_BF16X3);
    probe(ME_GEMM_NT, s.M, s.C, 3 * s.Hd, 3 * s.Hd, 3 * s.Hd, ME_F32);
    if (backward) {
        probe(ME_GEMM_NT, s.M, s.Hd, 3 * s.C, 3 * s.C, 3 * s.C, ME_BF16X3);       // dh (x gelu') -> three planes
        probe(ME_GEMM_NT, s.M, s.C, 3 * s.C3, 3 * s.C3, 3 * s.C3, ME_F32);        // dxn1
        probe(ME_GEMM_TN, s.C3, s.C, s.M, 3 * s.C3, 3 * s.C, ME_F32);
        probe(ME_GEMM_TN, s.C, s.C, s.M, 3 * s.C, 3 * s.C, ME_F32);
        probe(ME_GEMM_TN, s.Hd, s.C, s.M, 3 * s.Hd, 3 * s.C, ME_F32);
        probe(ME_GEMM_TN, s.C, s.Hd, s.M, 3 * s.C, 3 * s.Hd, ME_F32);
    }
    return align256(w);
}
size_t bwd_scratch_x3(const me_block_desc* d, const Dims& s) {
    // dy3, dh3 (two or three planes), dxn (fp32), dx1 (fp32), dx1_3, dout (fp32), dqkv (fp32), dqkv3, delta, LayerNorm partials
    return align256(s.M * s.C * 6) * 2 + align256(s.M * s.Hd * 6) + align256(s.M * s.C * 4) * 3 + align256(s.M * s.C3 * 4) +
           align256(s.M * s.C3 * 6) + align256((size_t)d->B * d->heads * d->N * 4) + align256(me_layernorm_bwd_workspace(s.C));
}

int block_fwd_x3(const me_block_desc* d, const Dims& s, const void* x, void* y, void* saved, void* workspace, void* stream) {
    ME_CHECK_ARG(d->res_dtype == ME_F32, "me_block_fwd: ME_BF16X3 blocks run on an fp32 token stream (res_dtype = ME_F32)");
    ME_CHECK_ARG(s.C % 256 == 0 && s.Hd % 256 == 0, "me_block_fwd: ME_BF16X3 needs C and hidden to be multiples of 256");
    char* ws = reinterpret_cast<char*>(workspace);
    const size_t gsz = gemm_scratch_x3(s, false);
    const bool keep = saved != nullptr;
    SavedX3 v = carve_saved_x3(d, s, keep ? saved : ws + gsz);
    me_gemm_desc g;
    int rc;
    auto run = [&](me_gemm_desc& q) { q.workspace = ws; q.workspace_bytes = (int64_t)gsz; return me_gemm(&q, stream); };
    const int64_t C = s.C, Hd = s.Hd;
    if ((rc = me_layernorm_fwd(x, ME_F32, d->ln1_g, d->ln1_b, v.xn1, ME_BF16X3, keep ? v.mean1 : nullptr, keep ? v.rstd1 : nullptr, s.M, s.C, d->eps, stream))) return rc;
    gemm_desc(g, ME_GEMM_NT, ME_BF16, s.M, s.C3, 3 * C, v.xn1, 3 * C, d->qkv_w, 3 * C, v.qkv, s.C3, ME_F32);
    g.bias = d->qkv_b;
    if ((rc = run(g))) return rc;
    if (s.hd == 64 && d->N > 64) {       // (N <= 64: the exact-fp32 one-workgroup-per-head kernels, attention_tiny.hip, are faster and exact)
        // three-product attention on the bf16 MFMA (attention_x3.hip): writes the proj Linear's planes itself; the fp32 copy only
        // when backward will want it (me_attention_bwd reads o and lse)
        rc = me_attention_fwd_x3(reinterpret_cast<const float*>(v.qkv), s.C3, keep ? reinterpret_cast<float*>(v.o) : nullptr, s.C, v.o3,
                                 keep ? v.lse : nullptr, d->B, d->N, d->heads, s.hd, d->scale, stream);
        if (rc) return rc;
    } else {
        if ((rc = me_attention_fwd(v.qkv, s.C3, v.o, s.C, keep ? v.lse : nullptr, d->B, d->N, d->heads, s.hd, d->scale, ME_F32, 0.f, 0, stream))) return rc;
        if ((rc = me_split3(reinterpret_cast<const float*>(v.o), C, v.o3, s.M, C, 0, stream))) return rc;
    }
    gemm_desc(g, ME_GEMM_NT, ME_BF16, s.M, C, 3 * C, v.o3, 3 * C, d->proj_w, 3 * C, v.x1, C, ME_F32);
    g.bias = d->proj_b; g.colscale = d->gamma1; g.residual = x; g.ldres = C; g.res_dtype = ME_F32;
    if ((rc = run(g))) return rc;
    if ((rc = me_layernorm_fwd(v.x1, ME_F32, d->ln2_g, d->ln2_b, v.xn2, ME_BF16X3, keep ? v.mean2 : nullptr, keep ? v.rstd2 : nullptr, s.M, s.C, d->eps, stream))) return rc;
    const bool two = x3_two_planes(s);
    const int64_t lda_a = two ? 2 * Hd : 3 * Hd;
    gemm_desc(g, ME_GEMM_NT, ME_BF16, s.M, Hd, 3 * C, v.xn2, 3 * C, d->fc1_w, 3 * C, v.a, lda_a, two ? ME_BF16X2 : ME_BF16X3);
    g.bias = d->fc1_b; g.act = ME_ACT_GELU;
    if (keep) { g.preact = v.hpre; g.ldpre = Hd; g.preact_dtype = ME_F32; g.flags = ME_GEMM_SAVE_GELU_GRAD; }
    if ((rc = run(g))) return rc;
    gemm_desc(g, ME_GEMM_NT, ME_BF16, s.M, C, 3 * Hd, v.a, lda_a, d->fc2_w, 3 * Hd, y, C, ME_F32);
    if (two) g.a_wrap_k = 2 * Hd;
    g.bias = d->fc2_b; g.colscale = d->gamma2; g.residual = v.x1; g.ldres = C; g.res_dtype = ME_F32;
    return run(g);
}

int block_bwd_x3(const me_block_desc* d, const Dims& s, const void* x, const void* dy, const void* saved, void* dx,
                 const me_block_grads* gr, void* workspace, void* stream) {
    ME_CHECK_ARG(d->res_dtype == ME_F32, "me_block_bwd: ME_BF16X3 blocks run on an fp32 token stream");
    const SavedX3 v = carve_saved_x3(d, s, const_cast<void*>(saved));
    char* ws = reinterpret_cast<char*>(workspace);
    size_t off = 0;
    auto take = [&](size_t n) { char* p = ws + off; off += align256(n); return p; };
    const size_t gsz = gemm_scratch_x3(s, true);
    void* gws = take(gsz);
    void* aws = take(aux_scratch(s));
    const int64_t C = s.C, C3 = s.C3, Hd = s.Hd;
    uint16_t* dy3 = reinterpret_cast<uint16_t*>(take(s.M * C * 6));
    uint16_t* dh3 = reinterpret_cast<uint16_t*>(take(s.M * Hd * 6));
    float* dxn = reinterpret_cast<float*>(take(s.M * C * 4));
    float* dx1 = reinterpret_cast<float*>(take(s.M * C * 4));
    uint16_t* dx1_3 = reinterpret_cast<uint16_t*>(take(s.M * C * 6));
    float* dout = reinterpret_cast<float*>(take(s.M * C * 4));
    float* dqkv = reinterpret_cast<float*>(take(s.M * C3 * 4));
    uint16_t* dqkv3 = reinterpret_cast<uint16_t*>(take(s.M * C3 * 6));
    float* delta = reinterpret_cast<float*>(take((size_t)d->B * d->heads * d->N * 4));
    void* ln2_ws = take(me_layernorm_bwd_workspace(s.C));
    me_ln_fold_set folds[2];
    me_gemm_desc g;
    int rc;
    // dX-side GEMM: C[M, N] = A3[M, 3K] W3t[N, 3K]^T
    // (a2: the A operand is ME_BF16X2 planes [hi | lo] -- lda = 2 K, the reduction index wraps; cdt ME_BF16X2 / ME_BF16X3: the output planes)
    auto nt = [&](const void* A3, int64_t K, const void* Wt3, void* Cout, int64_t N, int cdt, const void* factor, bool a2 = false) -> int {
        gemm_desc(g, ME_GEMM_NT, ME_BF16, s.M, N, 3 * K, A3, a2 ? 2 * K : 3 * K, Wt3, 3 * K, Cout, cdt == ME_BF16X3 ? 3 * N : cdt == ME_BF16X2 ? 2 * N : N, cdt);
        if (a2) g.a_wrap_k = 2 * K;
        if (factor) { g.aux = factor; g.ldaux = N; g.aux_dtype = ME_F32; g.flags = ME_GEMM_AUX_IS_FACTOR; }
        g.workspace = gws; g.workspace_bytes = (int64_t)gsz;
        return me_gemm(&g, stream);
    };
    // dW[n_out, n_in] = dOut^T In from the planes of both: (hi, hi) + (lo, hi) + (hi, lo)
    // The bias gradient db = colsum(dOut) rides on the first two launches where the kernel can fuse it (me_gemm_desc.colsum_a: the column
    // sums of the A operand from the fragments the kernel stages anyway, accumulated with C's beta): term 0 leaves colsum(hi), term 1
    // adds colsum(lo) -- 2^-17 relative, as the planes themselves; otherwise two plane passes of me_colsum.
    // (ld_out / ld_in: row lengths of the plane matrices -- 3 n for ME_BF16X3 rows, 2 n for ME_BF16X2 ones; only the hi and lo planes are read)
    auto wgrad = [&](const uint16_t* dOut3, int64_t n_out, const void* In3_, int64_t n_in, void* dW, float* dB, int64_t ld_out = 0, int64_t ld_in = 0) -> int {
        const uint16_t* In3 = reinterpret_cast<const uint16_t*>(In3_);
        if (!ld_out) ld_out = 3 * n_out;
        if (!ld_in) ld_in = 3 * n_in;
        const int64_t pa[3] = {0, n_out, 0}, pb[3] = {0, 0, n_in};
        bool db_done = dB == nullptr;
        if (dW) {
            // one launch + one fold: the three products as three segments of the wgrad kernel's reduction (gemm3_x3.hip); the bias gradient
            // rides on it.  Not for problems the planner keeps off the g3 wgrad family (few tokens): the three me_gemm calls below.
            gemm_desc(g, ME_GEMM_TN, ME_BF16, n_out, n_in, s.M, dOut3, ld_out, In3, ld_in, dW, n_in, gr->w_dtype);
            g.beta = gr->accumulate ? 1.0f : 0.0f;
            g.workspace = gws; g.workspace_bytes = (int64_t)gsz;
            g.colsum_a = dB;
            const int r1 = gemm_tn_x3_planes(&g, reinterpret_cast<hipStream_t>(stream));
            if (r1 == ME_OK) return ME_OK;
            if (r1 != ME_ERR_UNSUPPORTED) return r1;
            for (int t = 0; t < 3; ++t) {
                gemm_desc(g, ME_GEMM_TN, ME_BF16, n_out, n_in, s.M, dOut3 + pa[t], ld_out, In3 + pb[t], ld_in, dW, n_in, gr->w_dtype);
                g.beta = (t == 0 && !gr->accumulate) ? 0.0f : 1.0f;
                g.workspace = gws; g.workspace_bytes = (int64_t)gsz;
                if (dB && t < 2 && (t == 1 ? db_done : me_gemm_fuses_colsum(&g))) {      // (both or neither: term 1 adds onto term 0's sums)
                    g.colsum_a = dB;
                    db_done = true;
                }
                const int r = me_gemm(&g, stream);
                if (r) return r;
            }
        }
        if (!db_done) {
            const int r = me_colsum(dOut3, ME_BF16, ld_out, s.M, n_out, dB, gr->accumulate, aws, stream);
            return r ? r : me_colsum(dOut3 + n_out, ME_BF16, ld_out, s.M, n_out, dB, 1, aws, stream);
        }
        return ME_OK;
    };
    // ---- MLP branch
    if ((rc = me_split3(reinterpret_cast<const float*>(dy), C, dy3, s.M, C, 0, stream))) return rc;
    const bool two = x3_two_planes(s);                    // (the layout of v.a is the forward's: same rule, same answer)
    const int64_t ldh = two ? 2 * Hd : 3 * Hd;
    if ((rc = wgrad(dy3, C, v.a, Hd, gr->fc2_w, gr->fc2_b, 0, ldh))) return rc;
    if ((rc = nt(dy3, C, d->fc2_wt, dh3, Hd, two ? ME_BF16X2 : ME_BF16X3, v.hpre))) return rc;      // dA * gelu'(h) as planes
    if ((rc = wgrad(dh3, Hd, v.xn2, C, gr->fc1_w, gr->fc1_b, ldh, 0))) return rc;
    if ((rc = nt(dh3, Hd, d->fc1_wt, dxn, C, ME_F32, nullptr, two))) return rc;
    rc = me_ln_bwd_deferred(dxn, ME_F32, v.x1, ME_F32, v.mean2, v.rstd2, d->ln2_g, dy, ME_F32, dx1, ME_F32, gr->ln2_g, gr->ln2_b, gr->accumulate,
                            s.M, s.C, ln2_ws, stream, &folds[0]);
    if (rc) return rc;
    // ---- attention branch
    if ((rc = me_split3(dx1, C, dx1_3, s.M, C, 0, stream))) return rc;
    if ((rc = wgrad(dx1_3, C, v.o3, C, gr->proj_w, gr->proj_b))) return rc;
    if ((rc = nt(dx1_3, C, d->proj_wt, dout, C, ME_F32, nullptr))) return rc;
    if (s.hd == 64 && d->N > 64) {
        // three-product attention backward on the bf16 MFMA (attention_x3.hip): writes the planes the two qkv GEMMs read by itself --
        // no fp32 dqkv, no split pass
        rc = me_attention_bwd_x3(reinterpret_cast<const float*>(v.qkv), C3, reinterpret_cast<const float*>(v.o), C, dout, C, v.lse, delta, nullptr, C3,
                                 dqkv3, d->B, d->N, d->heads, s.hd, d->scale, stream);
        if (rc) return rc;
    } else {             // other head sizes: the exact-fp32 kernel + one split pass
        rc = me_attention_bwd(v.qkv, C3, v.o, C, dout, C, v.lse, delta, dqkv, C3, d->B, d->N, d->heads, s.hd, d->scale, ME_F32, 0.f, 0, stream);
        if (rc) return rc;
        if ((rc = me_split3(dqkv, C3, dqkv3, s.M, C3, 0, stream))) return rc;
    }
    if ((rc = wgrad(dqkv3, C3, v.xn1, C, gr->qkv_w, gr->qkv_b))) return rc;
    if ((rc = nt(dqkv3, C3, d->qkv_wt, dxn, C, ME_F32, nullptr))) return rc;
    rc = me_ln_bwd_deferred(dxn, ME_F32, x, ME_F32, v.mean1, v.rstd1, d->ln1_g, dx1, ME_F32, dx, ME_F32, gr->ln1_g, gr->ln1_b, gr->accumulate,
                            s.M, s.C, aws, stream, &folds[1]);
    if (rc) return rc;
    return me_ln_bwd_fold_sets(folds, 2, s.C, stream);
}

}  // namespace

extern "C" size_t me_block_saved_bytes(const me_block_desc* d) {
    Dims s;
    if (get_dims(d, s, "me_block_saved_bytes") != ME_OK) return 0;
    if (d->dtype == ME_BF16X3) return carve_saved_x3(d, s, nullptr).bytes;
    return carve_saved(d, s, nullptr).bytes;
}

extern "C" int me_block_emits_stats(const me_block_desc* d) {
    Dims s;
    if (get_dims(d, s, "me_block_emits_stats") != ME_OK || d->dtype == ME_BF16X3) return 0;
    return emits_stats(d, s) ? 1 : 0;
}

extern "C" size_t me_block_workspace_bytes(const me_block_desc* d, int backward) {
    Dims s;
    if (get_dims(d, s, "me_block_workspace_bytes") != ME_OK) return 0;
    if (d->dtype == ME_BF16X3)
        return backward ? gemm_scratch_x3(s, true) + aux_scratch(s) + bwd_scratch_x3(d, s)
                        : gemm_scratch_x3(s, false) + carve_saved_x3(d, s, nullptr).bytes;
    size_t w = gemm_scratch(d, s, backward != 0) + aux_scratch(s);
    if (!backward) return w + carve_saved(d, s, nullptr).bytes + stats_scratch(s);      // inference keeps the intermediates here
    // dy_c, dh, dxn (shared by dxn2 / dxn1), dx1, dx1_c, do, dqkv, delta
    w += align256(s.M * s.C * s.esz) * 4 + align256(s.M * s.Hd * s.esz) + align256(s.M * s.C * s.rsz) +
         align256(s.M * s.C3 * s.esz) + align256((size_t)d->B * d->heads * d->N * 4);
    w += align256(me_layernorm_bwd_workspace(s.C));            // norm2's dgamma / dbeta partials, folded with norm1's at the end
    return w + gemm_scratch(d, s, true) + aux_scratch(s);      // the side stream's own GEMM / column-sum scratch
}

extern "C" int me_block_fwd(const me_block_desc* d, const void* x, void* y, void* saved, void* workspace, size_t workspace_bytes,
                            void* stream) {
    Dims s;
    int rc = get_dims(d, s, "me_block_fwd");
    if (rc) return rc;
    ME_CHECK_ARG(x && y && workspace, "me_block_fwd: null pointer");
    ME_CHECK_ARG(d->qkv_w && d->proj_w && d->fc1_w && d->fc2_w && d->ln1_g && d->ln1_b && d->ln2_g && d->ln2_b,
                 "me_block_fwd: missing parameter");
    ME_CHECK_ARG(s.M * 2 * 4 <= 2 * (int64_t)align256(s.M * 4), "me_block_fwd: stash layout");
    ME_CHECK_ARG(workspace_bytes >= me_block_workspace_bytes(d, 0), "me_block_fwd: workspace too small");
    if (d->dtype == ME_BF16X3) return block_fwd_x3(d, s, x, y, saved, workspace, stream);
    char* ws = reinterpret_cast<char*>(workspace);
    const size_t gsz = gemm_scratch(d, s, false);
    void* gws = ws;
    const bool keep = saved != nullptr;
    Saved v = carve_saved(d, s, keep ? saved : ws + gsz + aux_scratch(s));
    const int dt = d->dtype, rdt = d->res_dtype;
    me_gemm_desc g;

    // inference with both LayerNorms folded into the Linear behind them (me_block_desc.qkv_wf ...): the token stream itself is
    // the GEMM's A operand (it must already be in the compute dtype), the statistics come from one read-only pass, and the
    // epilogue applies rstd * acc - rstd * mean * s + c -- the normalised tokens are neither written nor read back.  The pair
    // buffer [M][2] takes the place of the (adjacent) mean / rstd arrays of the activation stash.
    const bool fold = !keep && wants_fold(d);
    // ... and, where the residual GEMMs can emit them, the statistics come out of the proj / fc2 epilogues (one tiny combine pass
    // each) instead of a pass over the token stream: norm2's from proj, the NEXT block's norm1's from fc2 (d->y_stats)
    const bool stats = fold && emits_stats(d, s);
    // ... and where the qkv / fc1 launches can form the pairs from those partials themselves (row_parts), not even that
    const bool parts = fold && takes_parts(d, s);
    float* partials = reinterpret_cast<float*>(ws + gsz + aux_scratch(s) + v.bytes);
    if (fold) {
        gemm_desc(g, ME_GEMM_NT, dt, s.M, s.C3, s.C, x, s.C, d->qkv_wf, s.C, v.qkv, s.C3, dt);
        g.bias = d->qkv_c; g.col_shift = d->qkv_s;
        if (d->x_parts && parts) {
            g.row_parts = d->x_parts; g.row_nparts = (int32_t)(s.C / ME_STATS_GROUP); g.row_eps = d->eps;
        } else {
            const float* st1 = d->x_stats;
            if (d->x_parts) {
                if ((rc = me_row_stats_combine(d->x_parts, s.M, s.C, d->eps, v.mean1, stream))) return rc;
                st1 = v.mean1;
            } else if (!st1) {
                if ((rc = me_row_stats(x, rdt, v.mean1, s.M, s.C, d->eps, stream))) return rc;
                st1 = v.mean1;
            }
            g.row_affine = st1;
        }
    } else {
        rc = me_layernorm_fwd(x, rdt, d->ln1_g, d->ln1_b, v.xn1, dt, keep ? v.mean1 : nullptr, keep ? v.rstd1 : nullptr, s.M, s.C, d->eps, stream);
        if (rc) return rc;
        gemm_desc(g, ME_GEMM_NT, dt, s.M, s.C3, s.C, v.xn1, s.C, d->qkv_w, s.C, v.qkv, s.C3, dt);
        g.bias = d->qkv_b;
    }
    g.workspace = gws; g.workspace_bytes = (int64_t)gsz;
    if ((rc = me_gemm(&g, stream))) return rc;
    rc = me_attention_fwd(v.qkv, s.C3, v.o, s.C, keep ? v.lse : nullptr, d->B, d->N, d->heads, s.hd, d->scale, dt, 0.f, 0, stream);
    if (rc) return rc;
    gemm_desc(g, ME_GEMM_NT, dt, s.M, s.C, s.C, v.o, s.C, d->proj_w, s.C, v.x1, s.C, rdt);
    g.bias = d->proj_b; g.colscale = d->gamma1; g.residual = x; g.ldres = s.C; g.res_dtype = rdt;
    g.workspace = gws; g.workspace_bytes = (int64_t)gsz;
    if (stats) g.row_stats = partials;
    if ((rc = me_gemm(&g, stream))) return rc;
    if (fold) {
        gemm_desc(g, ME_GEMM_NT, dt, s.M, s.Hd, s.C, v.x1, s.C, d->fc1_wf, s.C, v.a, s.Hd, dt);
        g.bias = d->fc1_c; g.col_shift = d->fc1_s;
        if (stats && parts) {
            g.row_parts = partials; g.row_nparts = (int32_t)(s.C / ME_STATS_GROUP); g.row_eps = d->eps;
        } else {
            if (stats) rc = me_row_stats_combine(partials, s.M, s.C, d->eps, v.mean2, stream);
            else rc = me_row_stats(v.x1, rdt, v.mean2, s.M, s.C, d->eps, stream);
            if (rc) return rc;
            g.row_affine = v.mean2;
        }
    } else {
        rc = me_layernorm_fwd(v.x1, rdt, d->ln2_g, d->ln2_b, v.xn2, dt, keep ? v.mean2 : nullptr, keep ? v.rstd2 : nullptr, s.M, s.C, d->eps, stream);
        if (rc) return rc;
        gemm_desc(g, ME_GEMM_NT, dt, s.M, s.Hd, s.C, v.xn2, s.C, d->fc1_w, s.C, v.a, s.Hd, dt);
        g.bias = d->fc1_b;
    }
    g.act = ME_ACT_GELU;
    // (saved for backward: gelu'(h), not h -- the fc2 dgrad epilogue then multiplies by a stored factor)
    if (keep) { g.preact = v.hpre; g.ldpre = s.Hd; g.preact_dtype = gelu_grad_gg8(d, s) ? ME_GG8 : dt; g.flags = ME_GEMM_SAVE_GELU_GRAD; }
    g.workspace = gws; g.workspace_bytes = (int64_t)gsz;
    if ((rc = me_gemm(&g, stream))) return rc;
    gemm_desc(g, ME_GEMM_NT, dt, s.M, s.C, s.Hd, v.a, s.Hd, d->fc2_w, s.Hd, y, s.C, rdt);
    g.bias = d->fc2_b; g.colscale = d->gamma2; g.residual = v.x1; g.ldres = s.C; g.res_dtype = rdt;
    g.workspace = gws; g.workspace_bytes = (int64_t)gsz;
    float* yp = d->y_parts ? d->y_parts : partials;
    if (stats && (d->y_stats || d->y_parts)) g.row_stats = yp;
    if ((rc = me_gemm(&g, stream))) return rc;
    if (stats && d->y_stats) return me_row_stats_combine(yp, s.M, s.C, d->eps, d->y_stats, stream);
    return ME_OK;
}

extern "C" int me_block_bwd(const me_block_desc* d, const void* x, const void* dy, const void* saved, void* dx,
                            const me_block_grads* gr, void* workspace, size_t workspace_bytes, void* stream) {
    Dims s;
    int rc = get_dims(d, s, "me_block_bwd");
    if (rc) return rc;
    ME_CHECK_ARG(x && dy && saved && dx && gr && workspace, "me_block_bwd: null pointer");
    // (the fc2 weight gradient reads dy on the side stream while the last LayerNorm backward writes dx on `stream`)
    ME_CHECK_ARG(dx != dy && dx != x, "me_block_bwd: dx must not alias dy or x");
    ME_CHECK_ARG(d->qkv_wt && d->proj_wt && d->fc1_wt && d->fc2_wt && d->ln1_g && d->ln2_g, "me_block_bwd: missing parameter");
    ME_CHECK_ARG(!d->gamma1 && !d->gamma2, "me_block_bwd: layer-scale backward is composed by the host (needs the unscaled branch outputs)");
    ME_CHECK_ARG(me_dtype_ok(gr->w_dtype), "me_block_bwd: bad gradient dtype");
    ME_CHECK_ARG(workspace_bytes >= me_block_workspace_bytes(d, 1), "me_block_bwd: workspace too small");
    if (d->dtype == ME_BF16X3) return block_bwd_x3(d, s, x, dy, saved, dx, gr, workspace, stream);
    const Saved v = carve_saved(d, s, const_cast<void*>(saved));
    const int dt = d->dtype, rdt = d->res_dtype;
    char* ws = reinterpret_cast<char*>(workspace);
    size_t off = 0;
    auto take = [&](size_t n) { char* p = ws + off; off += align256(n); return p; };
    const size_t gsz = gemm_scratch(d, s, true);
    void* gws = take(gsz);
    void* aws = take(aux_scratch(s));
    char* dy_c = take(s.M * s.C * s.esz);
    char* dh = take(s.M * s.Hd * s.esz);
    char* dxn = take(s.M * s.C * s.esz);
    char* dx1 = take(s.M * s.C * s.rsz);
    char* dx1_c = take(s.M * s.C * s.esz);
    char* dout = take(s.M * s.C * s.esz);
    char* dqkv = take(s.M * s.C3 * s.esz);
    float* delta = reinterpret_cast<float*>(take((size_t)d->B * d->heads * d->N * 4));
    void* ln2_ws = take(me_layernorm_bwd_workspace(s.C));
    void* gws2 = take(gsz);
    void* aws2 = take(aux_scratch(s));
    me_ln_fold_set folds[2];
    const float beta = gr->accumulate ? 1.0f : 0.0f;
    me_gemm_desc g;
    // weight gradients on the side stream (see SideCtx); sc == null: everything on `stream`, in program order
    hipStream_t mstream = reinterpret_cast<hipStream_t>(stream);
    SideCtx* sc = wgrad_overlap_on() ? side_ctx(mstream) : nullptr;
    void* wstream = sc ? reinterpret_cast<void*>(sc->side) : stream;
    void* wgws = sc ? gws2 : gws;
    void* waws = sc ? aws2 : aws;
    bool forked = false;
    auto fork = [&](int i) -> int {           // the side stream may start on what `stream` has been given so far
        if (!sc) return ME_OK;
        if (hipEventRecord(sc->fork[i], mstream) != hipSuccess || hipStreamWaitEvent(sc->side, sc->fork[i], 0) != hipSuccess) {
            me_set_error("me_block_bwd: event fork failed");
            return ME_ERR_HIP;
        }
        forked = true;
        return ME_OK;
    };
    auto join = [&](int rc_in) -> int {       // `stream` continues behind everything the side stream was given
        if (!sc || !forked) return rc_in;
        if (hipEventRecord(sc->join, sc->side) != hipSuccess || hipStreamWaitEvent(mstream, sc->join, 0) != hipSuccess) {
            if (rc_in == ME_OK) { me_set_error("me_block_bwd: event join failed"); return ME_ERR_HIP; }
        }
        return rc_in;
    };

    const int gdt = gelu_grad_gg8(d, s) ? ME_GG8 : dt;        // (how the forward stored gelu': same rule, same answer)
    auto nt = [&](const void* A, int64_t K, const void* Wt, void* C, int64_t N, const void* aux) -> int {
        gemm_desc(g, ME_GEMM_NT, dt, s.M, N, K, A, K, Wt, K, C, N, dt);
        if (aux) { g.aux = aux; g.ldaux = N; g.aux_dtype = gdt; g.flags = ME_GEMM_AUX_IS_FACTOR; }
        g.workspace = gws; g.workspace_bytes = (int64_t)gsz;
        return me_gemm(&g, stream);
    };
    // dW[out, in] = dOut^T In (+ the bias gradient from the same kernel when it can fuse it)
    auto wgrad = [&](int fk, const void* dOut, int64_t n_out, const void* In, int64_t n_in, void* dW, float* dB) -> int {
        if (!dW && !dB) return ME_OK;             // (frozen parameter)
        int r = fork(fk);
        if (r) return r;
        if (dW) {
            gemm_desc(g, ME_GEMM_TN, dt, n_out, n_in, s.M, dOut, n_out, In, n_in, dW, n_in, gr->w_dtype);
            g.beta = beta; g.workspace = wgws; g.workspace_bytes = (int64_t)gsz;
            const bool fuse = dB && me_gemm_fuses_colsum(&g);
            if (fuse) g.colsum_a = dB;
            r = me_gemm(&g, wstream);
            if (r) return r;
            if (fuse) return ME_OK;
        }
        if (dB) return me_colsum(dOut, dt, n_out, s.M, n_out, dB, gr->accumulate, waws, wstream);
        return ME_OK;
    };

    // ---- MLP branch: y = x1 + fc2(gelu(fc1(LN2(x1))))
    const void* dyc = dy;
    if (rdt != dt) {
        if ((rc = me_cast(dy, rdt, dy_c, dt, s.M * s.C, stream))) return rc;
        dyc = dy_c;
    }
    if ((rc = wgrad(0, dyc, s.C, v.a, s.Hd, gr->fc2_w, gr->fc2_b))) return join(rc);
    if ((rc = nt(dyc, s.C, d->fc2_wt, dh, s.Hd, v.hpre))) return join(rc);                  // dA * gelu'(h)
    if ((rc = wgrad(1, dh, s.Hd, v.xn2, s.C, gr->fc1_w, gr->fc1_b))) return join(rc);
    if ((rc = nt(dh, s.Hd, d->fc1_wt, dxn, s.C, nullptr))) return join(rc);
    rc = me_ln_bwd_deferred(dxn, dt, v.x1, rdt, v.mean2, v.rstd2, d->ln2_g, dy, rdt, dx1, rdt, gr->ln2_g, gr->ln2_b, gr->accumulate, s.M,
                            s.C, ln2_ws, stream, &folds[0]);
    if (rc) return join(rc);
    // ---- attention branch: x1 = x + proj(attn(qkv(LN1(x))))
    const void* dx1c = dx1;
    if (rdt != dt) {
        if ((rc = me_cast(dx1, rdt, dx1_c, dt, s.M * s.C, stream))) return join(rc);
        dx1c = dx1_c;
    }
    if ((rc = wgrad(2, dx1c, s.C, v.o, s.C, gr->proj_w, gr->proj_b))) return join(rc);
    if ((rc = nt(dx1c, s.C, d->proj_wt, dout, s.C, nullptr))) return join(rc);
    rc = me_attention_bwd(v.qkv, s.C3, v.o, s.C, dout, s.C, v.lse, delta, dqkv, s.C3, d->B, d->N, d->heads, s.hd, d->scale, dt, 0.f, 0, stream);
    if (rc) return join(rc);
    if ((rc = wgrad(3, dqkv, s.C3, v.xn1, s.C, gr->qkv_w, gr->qkv_b))) return join(rc);
    if ((rc = nt(dqkv, s.C3, d->qkv_wt, dxn, s.C, nullptr))) return join(rc);
    rc = me_ln_bwd_deferred(dxn, dt, x, rdt, v.mean1, v.rstd1, d->ln1_g, dx1, rdt, dx, rdt, gr->ln1_g, gr->ln1_b, gr->accumulate, s.M, s.C,
                            aws, stream, &folds[1]);
    if (rc) return join(rc);
    return join(me_ln_bwd_fold_sets(folds, 2, s.C, stream));      // dgamma / dbeta of both LayerNorms: one launch
}

// process-wide switch for the side stream of me_block_bwd (default on);
// returns the previous setting
extern "C" int me_block_bwd_overlap(int enable) {
    const int prev = wgrad_overlap_on() ? 1 : 0;
    g_wgrad_overlap.store(enable ? 1 : 0);
    return prev;
}

extern "C" int me_encoder_fwd(const me_block_desc* blocks, int n_blocks, const void* x, void* y, void* pingpong, void* workspace,
                              size_t workspace_bytes, void* stream) {
    ME_CHECK_ARG(blocks && n_blocks > 0 && x && y && workspace, "me_encoder_fwd: bad args");
    ME_CHECK_ARG(n_blocks == 1 || pingpong, "me_encoder_fwd: more than one block needs the ping-pong token buffer");
    // block i writes y when (n_blocks - 1 - i) is even, the ping-pong buffer otherwise: the last block lands in y and no
    // block reads the buffer it writes
    // LayerNorm statistics travel with the tokens: block i leaves the pairs of its output in one of the two pair buffers at the
    // end of the workspace, block i + 1 (same width, same eps) starts from them instead of reading its input once more
    const void* in = x;
    const float* st_in = nullptr;
    const float* parts_in = nullptr;
    for (int i = 0; i < n_blocks; ++i) {
        void* out = ((n_blocks - 1 - i) % 2 == 0) ? y : pingpong;
        me_block_desc d = blocks[i];
        Dims s;
        int rc = get_dims(&d, s, "me_encoder_fwd");
        if (rc) return rc;
        d.x_stats = st_in;
        d.y_stats = nullptr;
        d.x_parts = parts_in;
        d.y_parts = nullptr;
        // (the pair buffers sit at the end of THIS block's workspace layout: the next block may read them only if its own layout
        //  is the same one -- same widths, heads and dtypes -- or its larger intermediates would overlap them)
        const me_block_desc& nx = blocks[i + 1 < n_blocks ? i + 1 : i];
        const bool chain = i + 1 < n_blocks && emits_stats(&d, s) && wants_fold(&nx) && nx.C == d.C && nx.B == d.B && nx.N == d.N &&
                           nx.eps == d.eps && nx.hidden == d.hidden && nx.heads == d.heads && nx.dtype == d.dtype &&
                           nx.res_dtype == d.res_dtype && me_block_workspace_bytes(&nx, 0) == me_block_workspace_bytes(&d, 0) &&
                           workspace_bytes >= me_block_workspace_bytes(&d, 0);
        if (chain) {
            char* tail = reinterpret_cast<char*>(workspace) + me_block_workspace_bytes(&d, 0) - 2 * align256((size_t)s.M * 8);
            Dims sn;
            if (get_dims(&nx, sn, "me_encoder_fwd") == ME_OK && takes_parts(&nx, sn))
                // the partials themselves travel (second partials buffer, right in front of the pair buffers): no combine launch.  One
                // buffer is enough: block i + 1's qkv has read it before its fc2 writes it again (same stream)
                d.y_parts = reinterpret_cast<float*>(tail - align256(me_row_stats_partial_bytes(s.M, s.C)));
            else
                d.y_stats = reinterpret_cast<float*>(tail + (i & 1) * align256((size_t)s.M * 8));
        }
        rc = me_block_fwd(&d, in, out, nullptr, workspace, workspace_bytes, stream);
        if (rc) return rc;
        st_in = d.y_stats;
        parts_in = d.y_parts;
        in = out;
    }
    return ME_OK;
}
