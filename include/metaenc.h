/*
 * metaenc.h -- C ABI of libmetaenc.so: the MI355X (gfx950 / CDNA4) implementation of the
 * Meta-Transformer modality-shared encoder hot path and the Data2Seq tokenizers that feed it.
 *
 * This is the drop-in boundary.  The reference has no native code on this path
 * -- its Block runs as ~15 stock ATen launches per layer -- so each entry point cites the
 * reference *Python* interface it replaces (paths relative to the reference checkout).  The
 * calling convention follows the reference's own native-op precedent
 * (PointCloud/openpoints/cpp/pointnet2_batch/src/sampling.cpp:16-48): plain device pointers +
 * explicit integer dims, caller pre-allocates every output, default-or-given stream -- except
 * that errors are returned (never exit(-1) as sampling_gpu.cu:253-257 does).
 *
 * Conventions
 *   - extern "C", no C++ / torch types.  All pointers are DEVICE pointers owned by the caller
 *     (torch tensors on the Python side); the library never allocates or frees persistent memory.
 *   - Tensors are row-major, last dim contiguous: tokens are [B*N, C] ("rows" = tokens).
 *   - dtype codes: ME_F32 / ME_BF16.  Statistics, biases, LayerNorm affine and accumulators are fp32.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream).  Launches are asynchronous
 *     on that stream; no implicit device synchronisation.
 *   - Return value: 0 = OK, negative = error; message via me_last_error() (thread-local).
 *   - Thread-safe for distinct streams.  Process-wide state is limited to: the optional launch-timing records of
 *     me_gemm_profile_* (mutex-protected, off by default), the lazily resolved RCCL entry points (me_comm_*), explicit
 *     handles (me_comm), per-device kernel attributes / work counters, the me_block_bwd_overlap() switch and -- for
 *     me_block_bwd -- one library-owned side stream + five events per (device, caller stream) pair, created at first use
 *     and kept for the life of the process (at most 64 pairs; later caller streams run the serial order).
 *     The library reads no environment variables.
 */
#ifndef METAENC_H
#define METAENC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ME_ABI_VERSION 1

enum { ME_F32 = 0, ME_BF16 = 1,
       ME_F16 = 2 /* storage only: accepted by me_cast / me_transpose_cast, which convert fp16 tensors at the boundary */,
       ME_BF16X3 = 3 /* an fp32 matrix [rows, cols] held as THREE bf16 planes side by side in one row of 3 * cols bf16 values:
                      *   left-operand order  (activations)  [ hi | lo | hi ]      hi = bf16(v), lo = bf16(v - hi)
                      *   right-operand order (weights)      [ hi | hi | lo ]
                      * so that an ordinary bf16 NT GEMM over the 3 * K long rows computes A_hi B_hi + A_lo B_hi + A_hi B_lo = A B to
                      * ~2^-17 relative (the dropped lo x lo term is 2^-18) on the bf16 matrix pipe: fp32-accurate arithmetic at a
                      * third of the bf16 rate instead of the 1/16 of the exact-fp32 MFMA.  Written by me_split3, by me_layernorm_fwd
                      * (y_dtype) and by me_gemm (c_dtype, left-operand order: the next Linear's A operand); read by me_gemm as plain
                      * ME_BF16 operands with K = 3 * cols.  me_block_desc.dtype = ME_BF16X3 selects this arithmetic for a whole Block
                      * (fp32 tokens in and out).  Reference arithmetic it stands in for: the fp32 default of README.md:113-150. */,
       ME_BF16X2 = 4 /* the same WITHOUT the repeated plane: [ hi | lo ] in one row of 2 * cols bf16 values -- me_gemm output only (c_dtype,
                      * ldc >= 2 N).  As the A operand of the next GEMM it needs me_gemm_desc.a_wrap_k = 2 * cols (the kernel re-reads the hi
                      * plane for the third K segment); as an operand of a plane weight gradient only hi and lo are read anyway.  Two thirds of
                      * the ME_BF16X3 bytes for the two [tokens, hidden] tensors of an ME_BF16X3 Block's MLP (round 6). */,
       ME_GG8 = 5 /* gelu'(h) in EIGHT BITS: one byte q per element, gelu' = -0.13 + q * (1.26 / 255)  (gelu' lies in [-0.129, 1.129]; step
                   * 0.0049, |error| <= 0.0025).  Only as me_gemm_desc.preact_dtype with ME_GEMM_SAVE_GELU_GRAD (the fc1 forward of a training
                   * Block writes it) and as aux_dtype with ME_GEMM_AUX_IS_FACTOR (the fc2 dgrad multiplies by it); ldpre / ldaux in BYTES =
                   * elements, multiples of 8.  Served by the resident kernel only: ask me_gemm_takes_gg8.  Half the HBM bytes of the bf16
                   * factor -- at config 2 a train step is bound by energy and an HBM byte costs what 300 bf16 flops cost. */ };

enum { ME_OK = 0, ME_ERR_ARG = -1, ME_ERR_UNSUPPORTED = -2, ME_ERR_HIP = -3, ME_ERR_WORKSPACE = -4 };

int me_abi_version(void);
const char* me_last_error(void);
/* name of the gfx arch the kernels were compiled for ("gfx950") */
const char* me_build_arch(void);
/* device facts used by bench.py: number of CUs, LDS bytes per CU, clock MHz of device `dev` */
int me_device_info(int dev, int* num_cus, int* lds_bytes, int* clock_mhz, char* name, int name_len);

/* ------------------------------------------------------------------ LayerNorm
 * Replaces nn.LayerNorm(C, eps) on [rows, C] tokens: Block.norm1 / Block.norm2
 * (PointCloud/openpoints/models/layers/attention.py:46,50 via norm.py:65).
 * mean/rstd ([rows], fp32) may be NULL in inference; they are what backward needs. */
/* me_row_stats: per row of [rows, cols] the pair (rstd, -rstd * mean) -> out [rows][2] fp32: the statistics of the same
 * LayerNorm for a Linear that has the normalisation folded in (me_gemm_desc.row_affine).  One read of x, no write of x. */
int me_row_stats(const void* x, int x_dtype, float* out, int64_t rows, int cols, float eps, void* stream);
/* The same pairs from per-row PARTIAL statistics that a producing kernel left behind (me_gemm_desc.row_stats: [cols / 256][rows]
 * pairs (mean, M2) over 256-column groups): Chan's parallel combination -- mean = average of the group means, M2 = sum of the
 * group M2 + 256 * sum (group mean - mean)^2 -- then rstd = (M2 / cols + eps)^-1/2.  cols % 256 == 0.  (A folded GEMM that takes
 * the partials as me_gemm_desc.row_parts does this in its own epilogue; this call serves the shapes that cannot.) */
size_t me_row_stats_partial_bytes(int64_t rows, int cols);
int me_row_stats_combine(const float* partials, int64_t rows, int cols, float eps, float* out, void* stream);
int me_layernorm_fwd(const void* x, int x_dtype, const float* gamma, const float* beta,
                     void* y, int y_dtype, float* mean, float* rstd,
                     int64_t rows, int cols, float eps, void* stream);

/* dx = LN-backward(dy) [+ dres]   ;  dgamma/dbeta ([C], fp32) are OVERWRITTEN (or accumulated when
 * accumulate_affine != 0); pass NULL for both to skip them (frozen encoder).  `dres` (optional) is the
 * gradient arriving through the residual connection, added into dx (Block.forward, attention.py:56-57).
 * workspace: me_layernorm_bwd_workspace(cols) bytes. */
size_t me_layernorm_bwd_workspace(int cols);
int me_layernorm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype,
                     const float* mean, const float* rstd, const float* gamma,
                     const void* dres, int dres_dtype, void* dx, int dx_dtype,
                     float* dgamma, float* dbeta, int accumulate_affine,
                     int64_t rows, int cols, void* workspace, void* stream);

/* ------------------------------------------------------------------ Linear / GEMM with fused epilogue
 * Replaces nn.Linear forward (Attention.qkv / Attention.proj attention.py:21-23,28,36; Mlp.fc1/fc2
 * mlp.py:23-27,30-34), the GELU between fc1 and fc2 (mlp.py:31), the residual adds of Block.forward
 * (attention.py:56-57), and -- for backward -- the dgrad / wgrad contractions autograd derives from them.
 *
 *   ME_GEMM_NT : C[M,N] = A[M,K] * B[N,K]^T          (A, B both K-contiguous; forward, dgrad with W^T)
 *   ME_GEMM_TN : C[M,N] = A[K,M]^T * B[K,N]          (A, B both K-major rows; wgrad: dW = dY^T X)
 *
 * Epilogue, applied in this order on the fp32 accumulator v:
 *   v = alpha * v ; v += bias[n] ; (store preact: v, or gelu'(v) with ME_GEMM_SAVE_GELU_GRAD) ; v = act(v) ;
 *   v *= gelu'(aux[m,n])  (v *= aux[m,n] with ME_GEMM_AUX_IS_FACTOR) ;
 *   v *= colscale[n] ; v += residual[res_row(m), n] ; v += beta * C_old[m,n] ; C[out_row(m), n] = v
 * with res_row(m) = res_row_mod ? m % res_row_mod : m   (pos-embed broadcast over the batch) and
 * out_row(m) = out_group_rows ? (m / out_group_rows) * out_group_stride + m % out_group_rows + out_row_offset : m
 * (patch tokens written behind a cls token).  Requirements: K % (16/sizeof(A elt)) == 0, N % 4 == 0,
 * 16-byte aligned rows. */
enum { ME_GEMM_NT = 0, ME_GEMM_TN = 1 };
enum { ME_ACT_NONE = 0, ME_ACT_GELU = 1 };
/* me_gemm_desc.flags.  The GELU of Mlp.forward (mlp.py:31) needs gelu'(h) in backward; a forward that saves gelu'(h)
 * itself (ME_GEMM_SAVE_GELU_GRAD: one exponential serves GELU and its derivative) turns the backward epilogue into a plain
 * multiplication by the saved factor (ME_GEMM_AUX_IS_FACTOR) -- no transcendental arithmetic in the dgrad GEMM. */
enum { ME_GEMM_SAVE_GELU_GRAD = 1, ME_GEMM_AUX_IS_FACTOR = 2 };

typedef struct me_gemm_desc {
    int32_t op;            /* ME_GEMM_NT / ME_GEMM_TN */
    int32_t ab_dtype;      /* dtype of A and B (ME_F32 -> exact fp32 MFMA, ME_BF16 -> bf16 MFMA, fp32 accumulate) */
    int64_t M, N, K;
    const void* A; int64_t lda;
    const void* B; int64_t ldb;
    void* C; int64_t ldc; int32_t c_dtype;
    int32_t act;           /* ME_ACT_* */
    float alpha, beta;
    const float* bias;     /* [N] or NULL */
    const float* colscale; /* [N] or NULL (layer-scale gamma, Image/.../base/vit.py:313-316) */
    void* preact; int64_t ldpre; int32_t preact_dtype;      /* optional: pre-activation saved for backward */
    int32_t aux_dtype;
    const void* aux; int64_t ldaux;                          /* optional: multiply by gelu'(aux) (GELU backward) */
    const void* residual; int64_t ldres; int32_t res_dtype;  /* optional */
    int32_t flags;         /* ME_GEMM_* bits (0 = none) */
    int64_t res_row_mod;
    int64_t out_group_rows, out_group_stride, out_row_offset;
    void* workspace; int64_t workspace_bytes;               /* optional scratch (split-K slabs), see below */
    float* colsum_a;       /* optional, ME_GEMM_TN only: receives sum_k A[k, m] for m in [0, M) -- the bias gradient
                            * that goes with a weight gradient dW = dY^T X (A = dY), computed on the matrix pipe from
                            * the operand tiles the kernel stages anyway instead of a second pass over dY; accumulated like C
                            * (colsum_a = beta * colsum_a + sums).  Only when
                            * me_gemm_fuses_colsum(d) != 0; otherwise me_gemm rejects the descriptor (use me_colsum). */
    /* optional, ME_GEMM_NT: a LayerNorm folded into this Linear (inference).  With W' = gamma o W as the B operand,
     *   Linear(LayerNorm(x)) = rstd_m * (x W'^T)[m, n] - rstd_m * mean_m * s[n] + c[n],   s[n] = sum_k W'[n, k],
     *   c[n] = sum_k beta[k] W[n, k] + b[n]  (passed as `bias`),
     * so the normalised activations are never written or re-read: row_affine = [M][2] fp32 pairs (rstd, -rstd * mean) from
     * me_row_stats, col_shift = s [N] fp32.  Applied ahead of bias / activation: v = ra[m][0] * acc + ra[m][1] * s[n]. */
    const float* row_affine;
    const float* col_shift;
    /* optional, ME_GEMM_NT with a bf16 residual epilogue (the proj / fc2 launches of a block): the per-row statistics of the
     * OUTPUT rows on the side, so that the LayerNorm reading this residual stream next (norm2 behind proj, the next block's
     * norm1 behind fc2) needs no pass of its own over it.  row_stats receives [N / 256][M] fp32 pairs (mean, M2 = sum of squared
     * deviations from that mean) over 256-column groups (one per output tile) -- me_row_stats_partial_bytes(M, N) bytes -- which
     * the next GEMM takes as its row_parts, or me_row_stats_combine folds into the (rstd, -rstd * mean) pairs of me_row_stats.  Only
     * when me_gemm_emits_row_stats(d) != 0 (whole 256 x 256 tiles on every CU, N % 256 == 0, plain residual epilogue); otherwise
     * me_gemm rejects it. */
    float* row_stats;
    /* optional, ME_GEMM_NT, with col_shift and INSTEAD of row_affine: the folded LayerNorm's statistics handed over as the partials
     * themselves -- row_parts = the [row_nparts][M] (mean, M2) pairs a previous me_gemm left in ITS row_stats (row_nparts = that
     * launch's N / 256 = this launch's K / 256); the kernel forms the (rstd, -rstd * mean) pairs of LayerNorm(K, row_eps) in its own
     * epilogue, so no me_row_stats_combine launch sits between the two GEMMs.  Only when me_gemm_takes_row_parts(d) != 0 (the resident
     * 256 x 256 kernel takes the problem; K = 256 .. 1024; plain bias / GELU epilogue); otherwise me_gemm rejects it. */
    const float* row_parts;
    int32_t row_nparts;
    float row_eps;
    /* optional, ME_GEMM_NT with ME_BF16 operands: the A operand's reduction index WRAPS -- column k >= a_wrap_k of A is read at k - a_wrap_k.  For an
     * A held as ME_BF16X2 planes [hi | lo] (lda >= 2 * cols) against ME_BF16X3 weights [hi | hi | lo]: K = 3 * cols, a_wrap_k = 2 * cols.  A
     * multiple of 128; only on the one-tile 256 x 256 family (me_gemm rejects it where the planner picks another: at least 128 tiles). 0 = off. */
    int64_t a_wrap_k;
} me_gemm_desc;

/* Scratch the kernel selected for this problem can use (0 = none).  wgrad-shaped problems (tiny output, very long
 * reduction) split the reduction over workgroups and fold fp32 slabs deterministically; without a workspace of this
 * size they still run, unsplit and slower. */
size_t me_gemm_workspace_bytes(const me_gemm_desc* d);
/* 1 if me_gemm(d) with d->colsum_a set (and a workspace of me_gemm_workspace_bytes(d)) will produce the column sums */
int me_gemm_fuses_colsum(const me_gemm_desc* d);
/* 1 if me_gemm(d) can serve d->row_stats (evaluated as if it were set) */
int me_gemm_emits_row_stats(const me_gemm_desc* d);
/* CUs that a communication library's kernels hold while gradient buckets are being reduced (RCCL: about one CU per channel).  Process-wide,
 * 0 = none (the default, and what me_comm_init leaves); returns the previous value.  It changes only how weight-gradient GEMMs (ME_GEMM_TN on the
 * 256 x 256 family) are PLANNED: their split-K grid asks for exactly one workgroup per CU; with a reservation the reduction runs on 256 - cus
 * workgroups instead -- whole split levels plus a leftover shared across tile boundaries, every part a fixed K range and slab, so results are
 * bit-reproducible per (shape, cus); on a free GPU it times like the default grid.  OPT-IN: in the one-GPU rehearsal (a kernel holding 16 CUs
 * beside backward) it did not remove the second-round penalty it was built for (profiles/r06_contention.txt) -- measure on the real node
 * (bench.py --gpus N prints per-rank weight-gradient times) before turning it on.  The workspace query covers either plan. */
int me_gemm_reserve_cus(int cus);
/* 1 if me_gemm(d) can serve d's ME_GG8 preact / aux (which must be set) */
int me_gemm_takes_gg8(const me_gemm_desc* d);
/* 1 if me_gemm(d) can serve d->a_wrap_k (which must be set) */
int me_gemm_takes_a_wrap(const me_gemm_desc* d);
/* 1 if me_gemm(d) can serve d->row_parts (which must be set) */
int me_gemm_takes_row_parts(const me_gemm_desc* d);
int me_gemm(const me_gemm_desc* d, void* stream);

/* Per-launch timing for roofline accounting (bench.py): while enabled, every me_gemm call -- including those made from
 * me_block_fwd / me_block_bwd -- and every LayerNorm / attention call is bracketed by HIP events on its stream.  Records
 * carry op = ME_GEMM_NT / ME_GEMM_TN with (M, N, K), or one of the codes below with (M, N, K) = (rows, cols, 0) for
 * LayerNorm / me_row_stats ((rows, cols, 1) for me_row_stats_combine) and (B * heads, N, head_dim) for attention.  me_gemm_profile_read synchronises
 * on the recorded events, fills up to `max` records in call order and returns how many there are (and clears them).
 * Off by default; costs two event records per GEMM when on. */
enum { ME_PROF_LN_FWD = 16, ME_PROF_LN_BWD = 17, ME_PROF_ATTN_FWD = 18, ME_PROF_ATTN_BWD = 19, ME_PROF_ROW_STATS = 20 };
typedef struct me_gemm_profile_rec {
    int32_t op, ab_dtype;
    int64_t M, N, K;
    float ms;          /* start of the (first) GEMM kernel to end of its last kernel (split-K fold included) */
    int32_t plan;      /* GEMM records: which kernel plan ran -- bits 0-3 the family (0 = exact-fp32 / generic 128 x 128 "g128", 2 = "g2b"
                        * 128 x 256 two workgroups per CU, 3 = "g2w" 256 x 256 K-step 32, 4 = "g3" 256 x 256 K-tile 64, resident when every
                        * CU gets a tile), bit 4 = a split-K tail / whole-problem split ran with a fold, bit 5 = the balanced static partition of a weight
                        * gradient (me_gemm_reserve_cus), bits 8-15 = split-K parts (bit 5: the most a tile gets); else 0 */
} me_gemm_profile_rec;
int me_gemm_profile_enable(int on);
int me_gemm_profile_read(me_gemm_profile_rec* out, int max);

/* column sums of a [rows, cols] matrix -> out[cols] fp32 (bias gradients).  accumulate != 0 adds into out.
 * workspace: me_colsum_workspace(cols) bytes. */
size_t me_colsum_workspace(int64_t cols);
int me_colsum(const void* x, int x_dtype, int64_t ldx, int64_t rows, int64_t cols,
              float* out, int accumulate, void* workspace, void* stream);
/* out[c] = sum_r x[r,c] * y[r,c] -- gradient of a per-channel scale: d gamma1/gamma2 of the layer-scale Block variant
 * (Image/detection/mmdet_custom/models/backbones/base/vit.py:313-316) = colsum(dy * branch_output).  Same workspace. */
int me_colsum_mul(const void* x, int x_dtype, int64_t ldx, const void* y, int y_dtype, int64_t ldy,
                  int64_t rows, int64_t cols, float* out, int accumulate, void* workspace, void* stream);

/* ------------------------------------------------------------------ Multi-head self-attention core
 * Replaces attention.py:28-35: the reshape/permute of qkv into heads, (q @ k^T) * scale, softmax(-1),
 * attn @ v and the transpose back to [B,N,C] -- as one fused kernel that never materialises [B,H,N,N].
 * qkv is the Linear output as the reference lays it out: row = token, columns [0,C)=Q, [C,2C)=K,
 * [2C,3C)=V, head h = columns [h*hd,(h+1)*hd) inside each third (ld_qkv = row stride in elements,
 * normally 3C).  out is [B*N, C] head-major.  lse ([B,H,N] fp32, may be NULL) = log-sum-exp of the scaled
 * scores, needed by backward.  scale is applied to the scores in fp32 AFTER QK^T (attention.py:31). */
int me_attention_fwd(const void* qkv, int64_t ld_qkv, void* out, int64_t ld_out, float* lse,
                     int B, int N, int H, int head_dim, float scale, int dtype, float p_drop, uint64_t seed, void* stream);
/* p_drop > 0: attn_drop of attention.py:33 in training mode (the Graph call site, tokengt_graph_encoder.py:191-205, 0.1):
 * each softmax probability is kept with probability 1 - p_drop and scaled by 1/(1 - p_drop) before P V; the mask is a
 * counter-based hash of (seed, b, h, q, k), so me_attention_bwd with the same p_drop / seed regenerates it. */

/* Backward of the above.  dqkv has the layout of qkv.  delta ([B,H,N] fp32) is scratch the caller provides. */
int me_attention_bwd(const void* qkv, int64_t ld_qkv, const void* out, int64_t ld_out,
                     const void* dout, int64_t ld_dout, const float* lse, float* delta,
                     void* dqkv, int64_t ld_dqkv,
                     int B, int N, int H, int head_dim, float scale, int dtype, float p_drop, uint64_t seed, void* stream);

/* fp8 (OCP e4m3) forward for long sequences -- BASELINE config 5 (Large video tokens [32, 1568, 1024]).  Same contract as
 * me_attention_fwd (math of Video/models/modeling_finetune.py:172-195) for bf16 qkv / out and head_dim 64, with Q, K, V and
 * the softmax probabilities quantised to e4m3 (per-tensor scales from an absmax pre-pass; P scaled by 2^7) and both
 * products on V_MFMA_SCALE_F32_32X32X64_F8F6F4 with unit block scales (the fp8 matrix instruction that runs at twice the
 * bf16 rate); running max / sum, the output accumulator and lse are fp32.  The reference has no fp8 path: tests/ state the
 * accuracy twice -- against fp64 attention (e4m3's own error: <= 7 % relative rms, <= 15 % of max|out|) and against the kernel's
 * arithmetic restated in torch with the same e4m3 roundings (<= 1e-2 of max|out|).  workspace: me_attention_fp8_workspace(...) bytes
 * (the quantised, re-laid-out Q / K / V^T).  Returns ME_ERR_UNSUPPORTED for head_dim != 64.  Backward: me_attention_bwd on
 * the bf16 qkv with this call's lse. */
size_t me_attention_fp8_workspace(int B, int N, int H, int head_dim);
int me_attention_fwd_fp8(const void* qkv, int64_t ld_qkv, void* out, int64_t ld_out, float* lse, int B, int N, int H,
                         int head_dim, float scale, void* workspace, size_t workspace_bytes, void* stream);

/* fp32-ACCURATE forward on the bf16 matrix pipe -- the attention of an ME_BF16X3 Block (fp32 qkv in, fp32 out).  Same contract and
 * math as me_attention_fwd with dtype ME_F32 (attention.py:28-35), but Q, K, V and the softmax probabilities are split into bf16
 * hi / lo parts inside the kernel and every product is formed three times (hi hi + lo hi + hi lo) on v_mfma_f32_16x16x32_bf16 with
 * fp32 accumulation: ~1e-5 relative where bf16 operands give ~3e-3, at ~5x the speed of the exact-fp32 MFMA kernel.  out (fp32,
 * [B*N, ld_out]) and out3 (the ME_BF16X3 planes [hi | lo | hi] of the same values, dense [B*N, 3 * H * head_dim]: the A operand of
 * the proj Linear) are both optional, at least one must be given; lse as me_attention_fwd.  head_dim 64 only (ME_ERR_UNSUPPORTED
 * otherwise: use me_attention_fwd).  Backward: me_attention_bwd (ME_F32) with this call's out and lse. */
int me_attention_fwd_x3(const float* qkv, int64_t ld_qkv, float* out, int64_t ld_out, void* out3, float* lse, int B, int N, int H,
                        int head_dim, float scale, void* stream);
/* Backward of the above (what autograd derives from attention.py:28-35), same three-product arithmetic: P recomputed from the
 * forward's lse, delta = dO . O in fp32 (written to `delta`, [B, H, N] scratch), dV = P^T dO, dP = dO V^T, dS = P o (dP - delta),
 * dQ = scale dS K, dK = scale dS^T Q -- every product three bf16 MFMAs on hi / lo split operands with fp32 accumulation.  Inputs
 * fp32.  The gradient goes out as fp32 `dqkv` (layout of qkv, fully written) and / or as `dqkv3`, the ME_BF16X3 planes of the same
 * [B*N, 3C] matrix (dense [B*N, 9C] bf16, [hi | lo | hi]: what the qkv weight-gradient and dgrad GEMMs of an ME_BF16X3 Block read) --
 * either may be NULL, not both.  head_dim 64 only (ME_ERR_UNSUPPORTED otherwise: me_attention_bwd, ME_F32). */
int me_attention_bwd_x3(const float* qkv, int64_t ld_qkv, const float* out, int64_t ld_out, const float* dout, int64_t ld_dout,
                        const float* lse, float* delta, float* dqkv, int64_t ld_dqkv, void* dqkv3, int B, int N, int H, int head_dim,
                        float scale, void* stream);

/* ------------------------------------------------------------------ One encoder Block, composed on the C side
 * Block.forward / its autograd (PointCloud/openpoints/models/layers/attention.py:55-58) as ONE call each: the same
 * kernels as the entry points above, launched back to back on `stream` without returning to the host in between --
 * the form a non-Python host binds, and what metatransformer_amd.Block uses for the plain (non-stochastic,
 * non-windowed) path.
 *   y = x1 + gamma2 * fc2(gelu(fc1(LN2(x1)))),   x1 = x + gamma1 * proj(attn(qkv(LN1(x))))
 * Weights are in the COMPUTE dtype, checkpoint layout [out, in]; LayerNorm affine, biases and gamma are fp32 (NULL bias /
 * gamma = absent).  The *_wt pointers are the transposed [in, out] copies backward's dgrad GEMMs read (NULL is fine for
 * forward-only use).  x / y / dx / dy are [B*N, C] in res_dtype (the residual stream). */
typedef struct me_block_desc {
    int32_t dtype;        /* compute dtype: ME_BF16 (bf16 MFMA), ME_F32 (exact fp32 MFMA) or ME_BF16X3 (fp32-accurate on the bf16 MFMA:
                           * res_dtype must be ME_F32, every weight pointer is the ME_BF16X3 right-operand form of the fp32 matrix --
                           * [out, 3 * in], and [in, 3 * out] for the *_wt copies; the stash and scratch layouts are the library's own (the two [tokens,
                           * hidden] tensors of the MLP are ME_BF16X2 where the GEMMs that read them can wrap their A operand); attention runs as three-product bf16 MFMA too
                           * (me_attention_fwd_x3 / _bwd_x3, ~1e-5) for head_dim 64 and N > 64, on the exact-fp32 kernels otherwise) */
    int32_t res_dtype;    /* dtype of x, y, dx, dy */
    int32_t B, N, C, heads, hidden;
    float eps, scale;     /* LayerNorm eps; attention scale (head_dim^-0.5 unless qk_scale was given) */
    const void *qkv_w, *proj_w, *fc1_w, *fc2_w;
    const void *qkv_wt, *proj_wt, *fc1_wt, *fc2_wt;
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    const float *qkv_b, *proj_b, *fc1_b, *fc2_b;
    const float *gamma1, *gamma2;      /* layer-scale (forward only in this entry point) */
    /* optional, inference only (me_block_fwd with saved == NULL, me_encoder_fwd): both LayerNorms folded into the Linear
     * behind them (me_gemm_desc.row_affine): qkv_wf / fc1_wf = (gamma o W) in the compute dtype, *_s[n] = sum_k W'[n, k]
     * of THOSE rounded values, *_c[n] = sum_k beta[k] W[n, k] + bias[n].  All six set -> the block runs me_row_stats + folded
     * GEMMs instead of me_layernorm_fwd + GEMMs (the normalised tokens are neither written nor re-read); any NULL -> as before. */
    const void *qkv_wf, *fc1_wf;
    const float *qkv_s, *qkv_c, *fc1_s, *fc1_c;
    /* optional, folded inference only: LayerNorm statistics handed from block to block.  x_stats = the [B*N][2] pairs of
     * me_row_stats(x, eps) for THIS block's norm1 (saves its pass over x); y_stats = where to leave the same pairs for y (taken
     * from the fc2 epilogue: me_gemm_desc.row_stats + me_row_stats_combine), for the next block's x_stats.  y_stats is written
     * only when the block's shapes let the GEMMs emit statistics -- me_block_emits_stats(d) -- and is left untouched otherwise.
     * me_encoder_fwd chains them by itself. */
    const float* x_stats;
    float* y_stats;
    /* The same hand-over WITHOUT the combine launch (what me_encoder_fwd and metatransformer_amd.Block use): y_parts = where the fc2
     * epilogue leaves the 256-column partials of y themselves ([C / 256][B*N] (mean, M2) pairs, me_row_stats_partial_bytes(B*N, C)
     * bytes; written when me_block_emits_stats(d) != 0), x_parts = the previous block's y_parts on this block's input -- its qkv GEMM
     * forms the pairs in its own epilogue (me_gemm_desc.row_parts) where me_gemm_takes_row_parts allows, else one combine launch
     * runs first.  x_parts takes precedence over x_stats; y_stats may be asked for beside y_parts (one combine launch). */
    const float* x_parts;
    float* y_parts;
} me_block_desc;

/* Gradient destinations of me_block_bwd; any pointer may be NULL (that gradient is skipped -- frozen encoder).
 * Weight gradients are [out, in] in w_dtype; bias / LayerNorm gradients fp32.  accumulate != 0 adds into the
 * destinations (gradient accumulation into a flat buffer) instead of overwriting. */
typedef struct me_block_grads {
    void *qkv_w, *proj_w, *fc1_w, *fc2_w;
    float *qkv_b, *proj_b, *fc1_b, *fc2_b;
    float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    int32_t w_dtype;
    int32_t accumulate;
} me_block_grads;

/* bytes of the activation stash forward writes for backward (xn1, qkv, o, x1, xn2, fc1 pre-activation, gelu output,
 * LayerNorm statistics, attention LSE), and of the scratch either call needs (intermediates, split-K slabs) */
size_t me_block_saved_bytes(const me_block_desc* d);
/* 1 if me_block_fwd(d, ..., saved = NULL) takes the folded route AND its residual GEMMs emit row statistics (y_stats is served) */
int me_block_emits_stats(const me_block_desc* d);
size_t me_block_workspace_bytes(const me_block_desc* d, int backward);
/* saved == NULL: inference (nothing kept; intermediates live in the workspace) */
int me_block_fwd(const me_block_desc* d, const void* x, void* y, void* saved, void* workspace, size_t workspace_bytes,
                 void* stream);
/* dx may be NULL (input does not need a gradient only if nothing upstream does -- rarely useful; kept for symmetry) */
int me_block_bwd(const me_block_desc* d, const void* x, const void* dy, const void* saved, void* dx,
                 const me_block_grads* g, void* workspace, size_t workspace_bytes, void* stream);
/* me_block_bwd issues the weight-gradient GEMMs (and their folds) on a library-owned SIDE stream, forked from / joined to `stream`
 * by events inside the call (the caller sees plain stream semantics): they fill the CUs the LayerNorm / attention backward kernels
 * and the tails of the dY -> dX chain leave idle.  On by default; off while `stream` is being captured into a hipGraph or after
 * me_block_bwd_overlap(0).  Returns the previous setting.  dx must not alias dy or x (the side stream still reads dy while the
 * last LayerNorm backward writes dx). */
int me_block_bwd_overlap(int enable);

/* The whole encoder, inference: y = Block_{n-1}(... Block_0(x)) -- nn.Sequential(*[Block] * L)(x) of README.md:124-149 as
 * one call for serving hosts.  All blocks share B, N, C and dtypes.  `pingpong` is one token buffer [B*N, C] in res_dtype
 * (unused when n_blocks == 1; x itself is never written); workspace >= me_block_workspace_bytes(&blocks[i], 0) for every i. */
int me_encoder_fwd(const me_block_desc* blocks, int n_blocks, const void* x, void* y, void* pingpong, void* workspace,
                   size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ element-wise helpers */
/* dst = (dst_dtype) src, n elements */
int me_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);
/* fp32 [rows, cols] (row stride ld_src elements) -> ME_BF16X3 [rows, 3 * cols] bf16 (dense): right_operand = 0 writes [hi | lo | hi]
 * (activations / gradients: the A operand of an NT GEMM, and -- plane by plane -- both operands of a TN weight-gradient GEMM),
 * right_operand = 1 writes [hi | hi | lo] (weights: the B operand).  cols % 4 == 0. */
int me_split3(const float* src, int64_t ld_src, void* dst, int64_t rows, int64_t cols, int right_operand, void* stream);
/* dst[c, r] = (dst_dtype) src[r, c]  (weight repack for dgrad: W[out,in] -> W^T[in,out]) */
int me_transpose_cast(const void* src, int src_dtype, void* dst, int dst_dtype,
                      int64_t rows, int64_t cols, void* stream);
/* The same for up to ME_TC_BATCH matrices in ONE launch (all dgrad copies of the encoder weights after an optimizer step:
 * 48 launches of ~12 us each otherwise). */
#define ME_TC_BATCH 48
typedef struct me_tc_batch {
    int32_t n, src_dtype, dst_dtype, reserved;
    struct { const void* src; void* dst; int64_t rows, cols; } item[ME_TC_BATCH];
} me_tc_batch;
int me_transpose_cast_batched(const me_tc_batch* b, void* stream);
/* me_split3 for up to ME_TC_BATCH fp32 matrices in one launch (src_dtype = ME_F32, dst_dtype = ME_BF16X3; rows, cols multiples of 4):
 * transposed = 0: item dst [rows, 3 cols] = planes of the matrix; transposed = 1: dst [cols, 3 rows] = planes of its TRANSPOSE (the
 * dgrad GEMMs' B operand) -- every ME_BF16X3 compute copy of an encoder's weights after an optimizer step in two launches. */
int me_split3_batched(const me_tc_batch* b, int transposed, int right_operand, void* stream);
/* y = x + pos (pos broadcast over batch: row m uses pos row m % pos_rows); PointCloud re-injects pos before
 * every block (PointCloud/openpoints/models/backbone/metatransformer.py:161-163). */
int me_add_rows(const void* x, int x_dtype, const void* pos, int pos_dtype, void* y, int y_dtype,
                int64_t rows, int64_t pos_rows, int cols, void* stream);

/* Window partition (merge = 0) / merge (merge = 1) of token rows for windowed attention
 * (Image/detection/mmdet_custom/models/backbones/base/vit.py:160-190: qkv is zero-padded to multiples of the window AFTER
 * the Linear, unfolded into ws x ws windows, attended per window, folded back and cropped).
 *   partition: src [B*H*W, cols] -> dst [B*gh*gw*ws*ws, cols], gh = ceil(H/ws), gw = ceil(W/ws); window (wy, wx) row-major,
 *              inside a window (iy, ix) row-major; positions outside the H x W grid are zero rows
 *   merge:     src [B*gh*gw*ws*ws, cols] -> dst [B*H*W, cols]  (padded rows dropped)
 * Same dtype both sides, row size a multiple of 16 bytes.  Integer index arithmetic only: bit-exact. */
int me_window_rows(const void* src, void* dst, int dtype, int B, int H, int W, int window, int cols, int merge,
                   void* stream);

/* Stochastic ops of the Block in training mode (identity in eval / p = 0, where they are never called):
 *   out[m,c] = (res ? res[m,c] : 0) + path(m / rows_per_sample) * keep(m,c) * v[m,c]
 * keep = Bernoulli(1-p_drop)/(1-p_drop) per element -- nn.Dropout of Attention.proj_drop / Mlp.drop
 * (PointCloud/openpoints/models/layers/attention.py:37, mlp.py:32,35); path = Bernoulli(1-p_path)/(1-p_path) per SAMPLE --
 * DropPath (PointCloud/openpoints/models/layers/drop.py:135-152).  The masks are a counter-based hash of (seed, index):
 * calling again with the same seed on the incoming gradient (res = NULL) is the backward.  The RNG stream differs
 * from torch's; distribution and scaling are the reference's.  colscale ([cols] fp32 or NULL) multiplies the masked value
 * per channel (layer-scale gamma applied outside the GEMM when its un-scaled branch output must be kept for d gamma). */
int me_dropout_add(const void* v, int v_dtype, const void* res, int res_dtype, void* out, int out_dtype,
                   int64_t rows, int cols, int64_t rows_per_sample, float p_drop, float p_path, uint64_t seed,
                   const float* colscale, void* stream);

/* ------------------------------------------------------------------ Data2Seq tokenizers
 * Patch gather ("im2col") for the convolutional patch-embeds; the projection itself is me_gemm (NT) on
 * the gathered matrix with the conv weight viewed as [Cout, Cin*kt*kh*kw].
 *   Image    Data2Seq/Image.py:16,26          Conv2d(3,C,k16,s16)            -> kt=1,kh=kw=16,st=1,sh=sw=16
 *   Acoustic Data2Seq/Acoustic.py:16,22       Conv2d(1,C,k16,stride(10,10))  -> overlapping patches
 *   Video    Video/models/modeling_finetune.py:283-297  Conv3d(3,C,k=s=(2,16,16))
 * x: [B,Cin,T,H,W] (T=1 for 2-D), cols out: [B*gt*gh*gw, Cin*kt*kh*kw] with feature order (c,dt,dy,dx),
 * token order (t,h,w) row-major == conv_out.flatten(2).transpose(1,2).  The index arithmetic is integer
 * and bit-exact. */
int me_patchify(const void* x, int x_dtype, void* cols, int cols_dtype,
                int B, int Cin, int T, int H, int W, int kt, int kh, int kw, int st, int sh, int sw,
                void* stream);
/* backward of me_patchify when patches do not overlap is the same gather reversed (scatter); for
 * overlapping patches gradients are accumulated.  dx must be zero-initialised by the caller. */
int me_unpatchify_add(const void* dcols, int dcols_dtype, float* dx,
                      int B, int Cin, int T, int H, int W, int kt, int kh, int kw, int st, int sh, int sw,
                      void* stream);

/* The whole patch embed -- Conv2d / Conv3d(Cin, Cout, kernel (kt, kh, kw), stride (st, sh, sw)) + flatten(2).transpose(1, 2)
 * (Data2Seq/Image.py:19-28, Video/models/modeling_finetune.py:283-297, Data2Seq/Acoustic.py:16-22) -- as one call:
 *   out[b * (prefix_rows + tokens) + prefix_rows + t, :] = patch(b, t) . weight^T + bias (+ pos[t, :])
 * with the token / feature order of me_patchify.  weight = the conv weight viewed as [Cout, Cin*kt*kh*kw] in the compute dtype
 * (ME_BF16: bf16 MFMA, fp32 accumulation; ME_F32: exact).  When x and weight are bf16, kh * kw == 256 with kw in {8, 16, 32, 64}
 * and the patch rows of x are 16-byte aligned (W, sw, H*W multiples of 8; the reference's image and tubelet embeds), the gather
 * runs INSIDE the GEMM's operand stager (csrc/patch_embed.hip): the gathered matrix is never materialised, no workspace is
 * needed and me_patch_embed_fused() returns 1.  Every other case (fp32 pixels, the spectrogram's stride-10 patches) runs
 * me_patchify into the caller's workspace followed by me_gemm -- same results as the two calls, one entry point. */
typedef struct me_patch_embed_desc {
    const void* x; int32_t x_dtype;         /* [B, Cin, T, H, W] contiguous (T = 1 for images) */
    int32_t B, Cin, T, H, W, kt, kh, kw, st, sh, sw;
    const void* weight; int32_t w_dtype;    /* [Cout, Cin*kt*kh*kw] */
    int32_t Cout;
    const float* bias;                      /* [Cout] or NULL */
    const void* pos; int32_t pos_dtype; int64_t ld_pos;    /* optional [tokens, Cout], added to every sample's tokens */
    int32_t prefix_rows;                    /* rows left untouched in front of every sample's tokens (cls / register tokens) */
    void* out; int32_t out_dtype; int64_t ld_out;          /* [B * (prefix_rows + tokens), Cout] */
    void* workspace; int64_t workspace_bytes;
} me_patch_embed_desc;
int me_patch_embed_fused(const me_patch_embed_desc* d);
size_t me_patch_embed_workspace_bytes(const me_patch_embed_desc* d);     /* 0 when fused */
int me_patch_embed(const me_patch_embed_desc* d, void* stream);
/* Gradients of the patch embed's parameters (what autograd derives from the Conv2d / Conv3d above; the tokenizer is the trainable part
 * of the frozen-encoder recipes):  dW[Cout, Cin*kt*kh*kw] = beta * dW + dY^T . patches(x),   dbias[Cout] = beta * dbias + colsum(dY)
 * (dbias may be NULL).  dy: [B * tokens, Cout] in the compute dtype d->w_dtype (prefix rows already dropped), row stride ld_dy.  Of *d
 * the input (x, x_dtype, geometry), w_dtype, Cout and the workspace are read.  In the fused case of me_patch_embed (and x < 1 GiB, a
 * problem the split-K wgrad kernel takes) the patches are gathered in THAT kernel's operand stager; otherwise me_patchify runs into the
 * workspace first.  workspace: me_patch_embed_wgrad_workspace_bytes(). */
int me_patch_embed_wgrad_fused(const me_patch_embed_desc* d, int dw_dtype);      /* 1: gathered inside the wgrad kernel */
size_t me_patch_embed_wgrad_workspace_bytes(const me_patch_embed_desc* d, int dw_dtype, int with_bias);
int me_patch_embed_wgrad(const me_patch_embed_desc* d, const void* dy, int64_t ld_dy, void* dw, int dw_dtype, float* dbias,
                         float beta, void* stream);

/* Time-series DataEmbedding (Data2Seq/Time_Series.py:109-126), eval-mode:
 *   out[b,l,:] = sum_{j<3} Wc[:, :, j] x[b,(l-1+j) mod L,:]  (Conv1d k3 circular, no bias, :29-42)
 *              + sum_f table_f[mark[b,l,f]]                     (TemporalEmbedding gathers, :82-93)
 *              + pe[l]                                          (PositionalEmbedding, :25-26)
 * x [B,L,cin] fp32, conv_w [C,cin,3] fp32, marks [B,L,n_mark] int32 or NULL, tables: n_mark device
 * pointers to [size_f, C] fp32 tables given in the order of the mark columns, table_rows[f] their sizes
 * (indices are bounds-checked: an out-of-range index returns ME_ERR_ARG through *err_flag, device int). */
int me_timeseries_embed(const float* x, const float* conv_w, const int32_t* marks, int n_mark,
                        const float* const* tables, const int32_t* table_rows, const float* pe,
                        void* out, int out_dtype, int B, int L, int cin, int C, int32_t* err_flag,
                        void* stream);

/* Backward helper of the circular Conv1d(k=3) token embedding (Data2Seq/Time_Series.py:29-42): the unfolded input
 *   out[(b,l), i*3 + k] = x[b, (l + k - 1) mod L, i]      ([B*L, ncols] fp32, ncols >= 3*cin, extra columns zero)
 * so that the weight gradient is one me_gemm (TN): dW[C, 3*cin] = dY[B*L, C]^T out.  Integer index arithmetic. */
int me_timeseries_unfold(const float* x, float* out, int B, int L, int cin, int ncols, void* stream);

/* ------------------------------------------------------------------ optimizer (fine-tune paths)
 * Fused AdamW step on a flat fp32 parameter / gradient bucket (torch.optim.AdamW semantics, decoupled
 * weight decay), grad_scale multiplies the gradient first (1/world_size after an all-reduce sum). */
int me_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                  float grad_scale, void* bf16_mirror, void* stream);
/* bf16_mirror (optional, n bf16 elements, same indexing as param): receives the updated parameters rounded to bf16 --
 * the forward-layout compute copies of the weights come out of the optimizer pass instead of one cast launch per weight. */

/* The fine-tune recipe around AdamW, fused and decided on the device (no host synchronisation between backward and the step):
 *   * parameter groups: layer-wise lr decay by block index and the no-decay groups of Video/optim_factory.py:28-41, 56-95 and
 *     Image/segmentation/mmcv_custom/layer_decay_optimizer_constructor.py:17-41 -> a table of contiguous segments of the flat
 *     bucket, each with its lr scale and weight decay (me_adamw_segment, device memory, sorted, the last end == n);
 *   * NativeScalerWithGradNormCount (Video/utils.py:376-404) = GradScaler.unscale_ + clip_grad_norm_ + "skip the step when
 *     a gradient is non-finite": me_grad_stats reduces the bucket to {L2 norm of the finite values (squares summed in double: a loss-scaled gradient whose
 *     square leaves the fp32 range is NOT reported as non-finite), number of non-finite values} (device),
 *     me_adamw_prepare turns that into the control block -- gradient multiplier = grad_scale / *loss_scale * min(1, max_norm /
 *     (norm + 1e-6)), skip flag, step counter (advanced only when the step is taken) and its bias corrections -- and
 *     me_adamw_step_segments applies (or skips) the step.  total_norm / found_inf stay readable in the control block (the
 *     caller's dynamic loss scale reads found_inf the same way torch's _amp_update_scale_ does).
 * stats / loss_scale may be NULL (no clipping statistics / no loss scaling); max_norm <= 0 disables clipping.  The control block
 * is caller-owned device memory, zeroed once before the first step. */
typedef struct me_adamw_segment { int64_t end; float lr_scale, weight_decay; } me_adamw_segment;      /* covers [previous end, end) */
typedef struct me_adamw_ctl { float grad_mul, skip, bc1, bc2_sqrt, total_norm, found_inf; int32_t step, reserved; } me_adamw_ctl;
size_t me_grad_stats_workspace(void);
int me_grad_stats(const float* grad, int64_t n, float* stats /* [2], device */, void* workspace, void* stream);
int me_adamw_prepare(me_adamw_ctl* ctl, const float* stats, const float* loss_scale, float grad_scale, float max_norm,
                     float beta1, float beta2, void* stream);
int me_adamw_step_segments(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                           const me_adamw_segment* segments, int n_segments, float lr, float beta1, float beta2, float eps,
                           const me_adamw_ctl* ctl, void* bf16_mirror, void* stream);

/* ------------------------------------------------------------------ token pooling for the task heads
 * [B, N, C] tokens -> [B, C] fp32 features: x.mean(1) ahead of fc_norm / x[:, 0] (Video/models/modeling_finetune.py:445-454),
 * and the 'max' / 'avg' global features of the PointCloud ClsHead (openpoints/models/classification/cls_base.py:126-133).
 * ME_POOL_MAX also writes the arg-max token per (b, c) (first index on ties; may be NULL in inference).
 * me_pool_tokens_bwd scatters dy [B, C] back to dx [B, N, C] (every element written). */
enum { ME_POOL_MEAN = 0, ME_POOL_MAX = 1, ME_POOL_FIRST = 2 };
int me_pool_tokens(const void* x, int x_dtype, float* out, int32_t* argmax, int B, int N, int C, int mode, void* stream);
int me_pool_tokens_bwd(const float* dy, const int32_t* argmax, void* dx, int dx_dtype, int B, int N, int C, int mode,
                       void* stream);

/* ------------------------------------------------------------------ point-cloud tokenizer front end
 * The index-producing part of PointPatchEmbed (PointCloud/openpoints/models/layers/group_embed.py:138-172); the per-point
 * MLP behind it is me_gemm + me_pool_tokens(ME_POOL_MAX).
 * me_fps: farthest point sampling with the selection rule AND tie order of furthest_point_sampling_kernel
 *   (PointCloud/openpoints/cpp/pointnet2_batch/src/sampling_gpu.cu:101-210; its thread-strided scan + shared-memory tree
 *   amount to: largest min-distance, then smallest bit-reversed thread id, then smallest index), one workgroup per cloud with
 *   the points and running distances held in registers for all m rounds: points [B, n, 3] fp32 -> idx [B, m] int32
 *   (idx[:, 0] = 0).  temp = [B, n] fp32 scratch, touched only by the memory-resident form (n > 24 576).
 * me_knn: for every query [B, m, 3] the k nearest of support [B, n, 3] (squared distance, ascending, ties -> lower
 *   index): what KNN.forward's cdist + topk(largest=False) selects (openpoints/models/layers/group.py:12-28). n <= 10240.
 * me_group_relative: rows[(b, s, j), 0:3] = points[b, idx[b, s, j]] - centers[b, s] (grouping_operation + relative_xyz,
 *   group.py:310-313), columns 3..cols-1 zero (cols = the GEMM's padded reduction length). */
int me_fps(const float* points, int32_t* idx, float* temp, int B, int n, int m, void* stream);
int me_knn(const float* support, const float* query, int32_t* idx, int B, int n, int m, int k, void* stream);
int me_group_relative(const float* points, const float* centers, const int32_t* idx, float* rows, int B, int n, int m,
                      int k, int cols, void* stream);

/* ------------------------------------------------------------------ position-embedding table resize
 * Replaces TIMMVisionTransformer.resize_pos_embed (Image/detection/mmdet_custom/models/backbones/base/vit.py:459-486,
 * also vit_adapter.py:73-78): the [h*w, cols] grid part of a pos-embed table resampled to [H*W, cols] with
 * F.interpolate(mode, align_corners=False) semantics, on the token-major layout (rows = grid positions).  The caller keeps
 * the cls row(s) as the reference does (cat).  fp32 arithmetic; modes: */
enum { ME_RESIZE_BILINEAR = 0, ME_RESIZE_BICUBIC = 1 };
int me_resize_rows(const void* src, int src_dtype, void* dst, int dst_dtype, int h, int w, int H, int W, int cols,
                   int mode, void* stream);

/* ------------------------------------------------------------------ data-parallel gradient exchange
 * The ONE exchange step of batch-sharded data parallelism: a sum all-reduce per flat gradient bucket over RCCL (xGMI
 * inside a node).  Replaces _allreduce_coalesced (Image/segmentation/mmseg_custom/core/utils/dist_utils.py:14-35:
 * bucket -> flatten -> dist.all_reduce -> div_(world_size) -> copy back) and DistributedDataParallel's bucketed
 * all-reduce (Video/run_class_finetuning.py:739-742) for the encoder's gradients.  There is no flatten / copy back
 * here (gradients already live in one flat buffer) and the 1/world factor is me_adamw_step's grad_scale.
 *
 * One process per GPU.  Rank 0 calls me_comm_unique_id and hands the ME_COMM_ID_BYTES opaque bytes to the other ranks
 * by any host channel (torch.distributed store, MPI, a file); every rank then calls me_comm_init.  RCCL is resolved at
 * run time by these two calls (dlopen: no link-time dependency; ME_ERR_UNSUPPORTED if it is not installed).
 *
 * me_allreduce_bucket(comm, buf, count, dtype, producer_stream): in-place sum over all ranks of buf[0..count)
 * (ME_F32 or ME_BF16), enqueued on the communicator's OWN stream after everything enqueued so far on
 * producer_stream (the backward stream) -- so the reduction of a finished bucket overlaps the rest of backward.
 * Asynchronous; every rank must call it with the same (count, dtype) sequence.
 * me_comm_join(comm, consumer_stream): consumer_stream (the optimizer's) waits for every reduction enqueued so far. */
#define ME_COMM_ID_BYTES 128
typedef struct me_comm me_comm;
int me_comm_unique_id(void* id_out);
int me_comm_init(me_comm** comm, const void* unique_id, int rank, int world, int device);
int me_comm_destroy(me_comm* comm);
int me_comm_info(const me_comm* comm, int* rank, int* world, int64_t* buckets_reduced);
int me_allreduce_bucket(me_comm* comm, void* buf, int64_t count, int dtype, void* producer_stream);
int me_comm_join(me_comm* comm, void* consumer_stream);

#ifdef __cplusplus
}
#endif
#endif /* METAENC_H */
