// attention_x3.hip -- fp32-ACCURATE attention forward on the bf16 matrix pipe (the attention of an ME_BF16X3 Block).
//
// Replaces, for fp32 tokens, attention.py:28-35 (q @ k^T * scale -> softmax -> @ v) of the reference Block
// (PointCloud/openpoints/models/layers/attention.py; same math as me_attention_fwd with dtype ME_F32) -- but where the exact
// kernel runs on v_mfma_f32_32x32x2_f32 (1/16 of the bf16 rate: 500 us per layer at [256,197,768]), this one splits every
// operand into bf16 hi / lo parts ON THE FLY and forms each product three times on v_mfma_f32_16x16x32_bf16:
//     S^T = K Q^T   = K_hi Q_hi^T + K_lo Q_hi^T + K_hi Q_lo^T          (fp32 accumulate; the dropped lo x lo term is 2^-18 relative)
//     O^T = V^T P^T = V_hi^T P_hi^T + V_lo^T P_hi^T + V_hi^T P_lo^T    (P = exp(scale * S - m) in fp32, split the same way)
// Softmax statistics (running max m, sum l), the rescaling, the output accumulator and lse stay fp32, exactly as in the
// reference's fp32 arithmetic; the scale is applied to the fp32 scores AFTER Q K^T (attention.py:31).
//
// Work: one workgroup = 4 waves x 32 queries (two 16-query tiles per wave: every K / V fragment read from LDS feeds both) = 128 queries
// of one (batch, head); keys stream through LDS in chunks of 64:
//   * every thread loads its share of the chunk's K and V rows as fp32 (prefetched into registers one chunk ahead), splits them
//     and writes   K_hi, K_lo  [64 keys][64 hd]  (row pitch 144 B: conflict-free 16-byte fragment reads)  and
//                  V_hi^T, V_lo^T [64 hd][64 keys] with the keys of a 32-key block in the order the score MFMA leaves them in a
//     lane: key 16 t + 4 g + r of the block sits at column 8 g + 4 t + r -- so that P^T feeds the second MFMA straight from the
//     accumulators (lane (g, q) holds keys 4 g + r of each 16-key score tile = columns 8 g .. 8 g + 7 of V^T) with no shuffle;
//   * per chunk and wave: 2 x (24 + 24) MFMAs, online softmax on 2 x 16 scores per lane; the max / sum over the four lane groups of a
//     query go through v_permlane16_swap / v_permlane32_swap (VALU, no trip through the LDS crossbar).
// head_dim 64 (Base / Large); any N.  lse = m + log l of the scaled scores (what me_attention_bwd reads).
// Outputs: fp32 [B*N, ld_out] and / or the ME_BF16X3 planes [hi | lo | hi] the proj Linear of an ME_BF16X3 Block reads.
#include "common.h"

namespace {

constexpr int X3_HD = 64, X3_QT = 2, X3_QB = 64 * X3_QT, X3_KC = 64;       // head dim, 16-query tiles per wave, queries per workgroup, keys per chunk
// LDS rows are 128 bytes, unpadded; the 16-byte chunk c of row r sits in slot c ^ f(r), f(r) = ((r >> 1) ^ (r >> 4)) & 7.  With that,
// every access of the kernel is bank-conflict free (brute-forced over the hardware's lane groups -- ds_read_b128 services lanes
// {0-3, 12-15, 20-27}, ... together -- for the fragment reads, the K row stores and the transposed V stores; PMC before: 62 % of the
// LDS cycles were conflicts with 144-byte padded rows).
constexpr int X3_PITCH = 128;
constexpr int X3_TILE = 64 * X3_PITCH;                   // one [64][64] bf16 array
constexpr int X3_STAGE = 4 * 16 * X3_QT * 272;           // epilogue staging: 4 waves x 32 rows x 272 B
constexpr int X3_LDS = 4 * X3_TILE > X3_STAGE ? 4 * X3_TILE : X3_STAGE;      // K_hi, K_lo, V_hi^T, V_lo^T
__device__ __forceinline__ int x3_f(int row) { return ((row >> 1) ^ (row >> 4)) & 7; }

__device__ __forceinline__ void split2(float v, bf16_t& h, bf16_t& l) {
    h = (bf16_t)v;
    l = (bf16_t)(v - (float)h);
}

// value of the same lane in the other row of a row pair (lane ^ 16) / in the other half wave (lane ^ 32), combined: VALU only
__device__ __forceinline__ float x3_max4(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float x3_sum4(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

__global__ __launch_bounds__(256) void attn_fwd_x3_kernel(const float* __restrict__ qkv, int64_t ld_qkv, float* __restrict__ out, int64_t ld_out,
                                                          uint16_t* __restrict__ out3, float* __restrict__ lse, int N, int H, float scale) {
    __shared__ __attribute__((aligned(16))) char lds[X3_LDS];
    char* Kh = lds;
    char* Kl = lds + X3_TILE;
    char* Vh = lds + 2 * X3_TILE;
    char* Vl = lds + 3 * X3_TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, q = lane & 15;
    const int nqb = (N + X3_QB - 1) / X3_QB;
    const int qb = blockIdx.x % nqb;
    const int bh = blockIdx.x / nqb;
    const int b = bh / H, h = bh % H;
    const int C = H * X3_HD;
    const float* base = qkv + (int64_t)b * N * ld_qkv + h * X3_HD;      // Q of (b, h): + row * ld; K: + C; V: + 2 C
    const int q0 = qb * X3_QB + wave * 16 * X3_QT;                      // this wave's first query (tile u: q0 + 16 u + q)

    // ---- Q fragments (B operand of S^T = K Q^T): lane (g, q) holds Q[q0 + 16 u + q][32 ks + 8 g .. + 7], hi and lo
    bf16x8 qh[X3_QT][2], ql[X3_QT][2];
#pragma unroll
    for (int u = 0; u < X3_QT; ++u) {
        const int row = q0 + 16 * u + q < N ? q0 + 16 * u + q : N - 1;
        const float* qp = base + (int64_t)row * ld_qkv;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(qp + 32 * ks + 8 * g);
            const f32x4 c = *reinterpret_cast<const f32x4*>(qp + 32 * ks + 8 * g + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bf16_t hh, ll;
                split2(a[e], hh, ll); qh[u][ks][e] = hh; ql[u][ks][e] = ll;
                split2(c[e], hh, ll); qh[u][ks][4 + e] = hh; ql[u][ks][4 + e] = ll;
            }
        }
    }

    // ---- loader roles.  K: thread -> (key = tid >> 2, 16 hd values at 16 (tid & 3)): four 16-byte loads, two 16-byte LDS stores per part.
    //                     V: thread -> (score tile t = tid >> 6, lane group vg = (tid >> 4) & 3, hd quad vq = tid & 15): keys 16 t + 4 vg + r
    const int kkey = tid >> 2, kcol = 16 * (tid & 3);
    const int vt = tid >> 6, vg = (tid >> 4) & 3, vq = tid & 15;
    f32x4 kreg[4], vreg[4];
    // (row offsets in 32-bit arithmetic from the wave-uniform `base`: the launcher checks N * ld_qkv < 2^31)
    const int ldq = (int)ld_qkv;
    auto gload = [&](int chunk) {
        const int k0 = chunk * X3_KC;
        {
            int key = k0 + kkey;
            key = key < N ? key : N - 1;                        // (clamped re-read: masked below by the score mask)
            const float* kp = base + (key * ldq + C + kcol);
#pragma unroll
            for (int i = 0; i < 4; ++i) kreg[i] = *reinterpret_cast<const f32x4*>(kp + 4 * i);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int key = k0 + 16 * vt + 4 * vg + r;
            key = key < N ? key : N - 1;
            // (raw: nothing here may USE the loaded value -- a select right behind the load makes hipcc wait for it on the spot, and the
            //  whole point of the prefetch is that it lands under this chunk's MFMAs; rows past N are zeroed in lstore)
            vreg[r] = *reinterpret_cast<const f32x4*>(base + (key * ldq + 2 * C + 4 * vq));
        }
    };
    auto lstore = [&](int chunk) {
        // K rows: [key][hd], hi and lo
        bf16x8 h8[2], l8[2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bf16_t hh, ll;
                split2(kreg[i][e], hh, ll);
                h8[i >> 1][4 * (i & 1) + e] = hh;
                l8[i >> 1][4 * (i & 1) + e] = ll;
            }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int off = kkey * X3_PITCH + ((((kcol >> 3) + j) ^ x3_f(kkey)) << 4);
            *reinterpret_cast<bf16x8*>(Kh + off) = h8[j];
            *reinterpret_cast<bf16x8*>(Kl + off) = l8[j];
        }
        // V^T rows: [hd][key column]; the key (tile t, group vg, r) of the chunk sits at column 32 (t >> 1) + 8 vg + 4 (t & 1) + r, i.e. in
        // 16-byte chunk 4 (t >> 1) + vg, half (t & 1).  A lane owns four hd rows 4 vq + e; instruction k stores row e = (k + vq) & 3 (the
        // rotation puts rows of both parities into one instruction: with the slot permutation that makes the sixteen lanes conflict-free)
        u32x2 h4[4], l4[4];                                     // (packed pairs: the rotation below selects whole dwords)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bf16x4 hv, lv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bf16_t hh, ll;
                // (keys past N: P is zero there, but 0 x NaN of a stale re-read row is not -- zero the value itself)
                split2(chunk * X3_KC + 16 * vt + 4 * vg + r < N ? vreg[r][e] : 0.0f, hh, ll);
                hv[r] = hh; lv[r] = ll;
            }
            h4[e] = __builtin_bit_cast(u32x2, hv);
            l4[e] = __builtin_bit_cast(u32x2, lv);
        }
        // rotate the four row registers by vq & 3 (two select stages: by 1, by 2) so that instruction k finds row (k + vq) & 3 in slot k
        const bool b0 = vq & 1, b1 = vq & 2;
        u32x2 hs[4], ls[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                hs[k][w] = b0 ? h4[(k + 1) & 3][w] : h4[k][w];
                ls[k][w] = b0 ? l4[(k + 1) & 3][w] : l4[k][w];
            }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                h4[k][w] = b1 ? hs[(k + 2) & 3][w] : hs[k][w];
                l4[k][w] = b1 ? ls[(k + 2) & 3][w] : ls[k][w];
            }
        const int vchunk = 4 * (vt >> 1) + vg;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = 4 * vq + ((k + vq) & 3);
            const int off = row * X3_PITCH + ((vchunk ^ x3_f(row)) << 4) + 8 * (vt & 1);
            *reinterpret_cast<u32x2*>(Vh + off) = h4[k];
            *reinterpret_cast<u32x2*>(Vl + off) = l4[k];
        }
    };

    // ---- state per query tile u: O^T[hd = 16 t + 4 g + r][q] for t = 0..3 (16 registers, all of query q), running max / sum of query q
    f32x4 acc[X3_QT][4];
    float m_run[X3_QT], l_run[X3_QT];
#pragma unroll
    for (int u = 0; u < X3_QT; ++u) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        m_run[u] = -INFINITY;
        l_run[u] = 0.f;
    }
    const float sl2 = scale * 1.4426950408889634f;              // scores in log2 units: exp(x) = exp2(x log2 e)

    const int nchunk = (N + X3_KC - 1) / X3_KC;
    gload(0);
    for (int c = 0; c < nchunk; ++c) {
        __syncthreads();                                        // every wave is done reading the previous chunk
        lstore(c);
        __syncthreads();
        if (c + 1 < nchunk) gload(c + 1);                       // next chunk's rows fly under this chunk's MFMAs

        // S^T tiles: rows = keys 16 T + 4 g + r, column = query q of tile u; a K fragment read serves both query tiles.  The fragment
        // reads run ONE STEP AHEAD of the MFMAs that use them (sched_barrier pins the order): left to itself hipcc emits read -> wait ->
        // six MFMAs per step, i.e. one exposed LDS latency per 96 matrix-pipe clocks.
        const int nvalid = q0 < N ? N - c * X3_KC : 0;          // valid keys of this chunk (0: this wave owns no query -- last row block)
        f32x4 s[X3_QT][4];
#pragma unroll
        for (int u = 0; u < X3_QT; ++u)
#pragma unroll
            for (int T = 0; T < 4; ++T) s[u][T] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            // row 16 T + q, chunk 4 ks + g -> slot (4 ks + g) ^ f(16 T + q) = cg ^ 4 ks ^ T with cg = g ^ ((q >> 1) & 7)
            const char* kbase_h = Kh + q * X3_PITCH;
            const char* kbase_l = Kl + q * X3_PITCH;
            const int cg = g ^ ((q >> 1) & 7);
            bf16x8 fh = *reinterpret_cast<const bf16x8*>(kbase_h + (cg << 4)), fl = *reinterpret_cast<const bf16x8*>(kbase_l + (cg << 4));
#pragma unroll
            for (int st = 0; st < 8; ++st) {                    // step = (T, ks) = (st >> 1, st & 1)
                const int T = st >> 1, ks = st & 1;
                bf16x8 nh = fh, nl = fl;
                if (st + 1 < 8) {
                    const int off = (16 * ((st + 1) >> 1)) * X3_PITCH + ((cg ^ (4 * ((st + 1) & 1)) ^ ((st + 1) >> 1)) << 4);
                    nh = *reinterpret_cast<const bf16x8*>(kbase_h + off);
                    nl = *reinterpret_cast<const bf16x8*>(kbase_l + off);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (16 * T < nvalid) {                          // (round 6: key tiles wholly past N -- masked to P = 0 below -- skip their products)
#pragma unroll
                    for (int u = 0; u < X3_QT; ++u) {
                        s[u][T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, qh[u][ks], s[u][T], 0, 0, 0);
                        s[u][T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl, qh[u][ks], s[u][T], 0, 0, 0);
                        s[u][T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, ql[u][ks], s[u][T], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                fh = nh; fl = nl;
            }
        }
        bf16x8 ph[X3_QT][2], pl[X3_QT][2];                      // P^T B-operand fragments of the two 32-key blocks
        if (nvalid > 0)                                         // (a wave without a query of its own has nothing to normalise)
#pragma unroll
        for (int u = 0; u < X3_QT; ++u) {
            // mask keys past N, scale into log2 units, chunk maximum of query q (in-lane over 16, then over the four lane groups)
            float mx = -INFINITY;
            if (c * X3_KC + X3_KC > N) {                        // (wave-uniform: only a ragged LAST chunk has keys to mask)
#pragma unroll
                for (int T = 0; T < 4; ++T) {
                    if (16 * T >= nvalid) continue;             // (a tile wholly past N: no arithmetic at all, P = 0 below)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = c * X3_KC + 16 * T + 4 * g + r;
                        const float v = key < N ? s[u][T][r] * sl2 : -INFINITY;
                        s[u][T][r] = v;
                        mx = fmaxf(mx, v);
                    }
                }
            } else {
#pragma unroll
                for (int T = 0; T < 4; ++T) {
                    s[u][T] *= sl2;
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[u][T][r]);
                }
            }
            mx = x3_max4(mx);
            const float m_new = fmaxf(m_run[u], mx);            // (finite: every chunk holds at least one key < N)
            const float alpha = __builtin_amdgcn_exp2f(m_run[u] - m_new);
            float sum = 0.f;
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                if (16 * T >= nvalid) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { ph[u][T >> 1][4 * (T & 1) + r] = (bf16_t)0.0f; pl[u][T >> 1][4 * (T & 1) + r] = (bf16_t)0.0f; }
                    continue;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(s[u][T][r] - m_new);
                    sum += p;
                    bf16_t hh, ll;
                    split2(p, hh, ll);
                    ph[u][T >> 1][4 * (T & 1) + r] = hh;
                    pl[u][T >> 1][4 * (T & 1) + r] = ll;
                }
            }
            sum = x3_sum4(sum);
            l_run[u] = l_run[u] * alpha + sum;
            m_run[u] = m_new;
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[u][t] *= alpha;
        }
        // O^T += V^T P^T: A operand = V^T rows hd 16 t + i (i = q), columns (keys) 32 blk + 8 g ..; a V fragment read serves both tiles;
        // reads one step ahead of the MFMAs, as above
        {
            const char* vbase_h = Vh + q * X3_PITCH;
            const char* vbase_l = Vl + q * X3_PITCH;
            const int cg = g ^ ((q >> 1) & 7);                  // row 16 t + q, chunk 4 blk + g -> slot cg ^ 4 blk ^ t
            bf16x8 fh = *reinterpret_cast<const bf16x8*>(vbase_h + (cg << 4)), fl = *reinterpret_cast<const bf16x8*>(vbase_l + (cg << 4));
#pragma unroll
            for (int st = 0; st < 8; ++st) {                    // step = (t, blk) = (st >> 1, st & 1)
                const int t = st >> 1, blk = st & 1;
                bf16x8 nh = fh, nl = fl;
                if (st + 1 < 8) {
                    const int off = (16 * ((st + 1) >> 1)) * X3_PITCH + ((cg ^ (4 * ((st + 1) & 1)) ^ ((st + 1) >> 1)) << 4);
                    nh = *reinterpret_cast<const bf16x8*>(vbase_h + off);
                    nl = *reinterpret_cast<const bf16x8*>(vbase_l + off);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (32 * blk < nvalid) {                        // (a 32-key block wholly past N holds P = 0)
#pragma unroll
                    for (int u = 0; u < X3_QT; ++u) {
                        acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, ph[u][blk], acc[u][t], 0, 0, 0);
                        acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl, ph[u][blk], acc[u][t], 0, 0, 0);
                        acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, pl[u][blk], acc[u][t], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                fh = nh; fl = nl;
            }
        }
    }

    // ---- epilogue: O[q][hd] = O^T / l.  Lane (g, q) holds hd = 16 t + 4 g + r of query q: stored from here a wave instruction would
    // touch sixteen rows x 64 bytes (and the planes sixteen rows x 32 bytes).  Each wave therefore turns its 32 x 64 fp32 block around
    // in its own slice of the (now idle) operand LDS and stores whole rows: sixteen consecutive lanes = one 256-byte fp32 row, or one
    // 128-byte row of each bf16 plane.
    __syncthreads();                                            // every wave has finished reading K / V of the last chunk
    constexpr int OP = 272;                                     // bytes per staged row (64 fp32 + 16 B: conflict-free 16-byte accesses)
    char* stage = lds + wave * (16 * X3_QT * OP);
#pragma unroll
    for (int u = 0; u < X3_QT; ++u) {
        const float inv = 1.0f / l_run[u];
        const int row = q0 + 16 * u + q;
        if (lse && g == 0 && row < N) lse[((int64_t)b * H + h) * N + row] = (m_run[u] + __builtin_amdgcn_logf(l_run[u])) * 0.6931471805599453f;   // log2 -> ln
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(stage + (16 * u + q) * OP + (16 * t + 4 * g) * 4) = acc[u][t] * inv;
    }
    // (a wave's DS operations execute in order: its own reads below see its own writes above; no other wave touches this slice)
    const int rl = lane >> 4, ch = lane & 15;                   // row inside a group of four, 16-byte chunk of the row
#pragma unroll
    for (int i = 0; i < 4 * X3_QT; ++i) {
        const int lr = 4 * i + rl;                              // local row 0 .. 31
        const int row = q0 + lr;
        const f32x4 o = *reinterpret_cast<const f32x4*>(stage + lr * OP + ch * 16);
        if (row < N) {
            const int64_t orow = (int64_t)b * N + row;
            const int col = h * X3_HD + 4 * ch;
            if (out) *reinterpret_cast<f32x4*>(out + orow * ld_out + col) = o;
            if (out3) store4_split3(out3 + orow * 3 * C, C, col, o);
        }
    }
}

// =====================================================================================================================================
// Backward of the same attention (attention.py:28-35 under autograd), three bf16 products per operand pair as well.  With
// P = exp(scale S - lse) recomputed from the forward's lse (fp32), delta_i = dO_i . O_i (fp32):
//     dV = P^T dO        dP = dO V^T        dS = P o (dP - delta)        dQ = scale dS K        dK = scale dS^T Q
// as two kernels in the shape of the forward: the dQ kernel owns 128 queries of a (batch, head) and streams {K, V} in 64-key chunks;
// the dK / dV kernel owns 128 keys and streams {Q, dO} (+ lse, delta) in 64-query chunks.  Every streamed chunk is loaded ONCE per
// thread as 4 rows x 4 head-dim values and written to LDS both as rows (the A operand of the score-shaped products) and transposed
// with the forward's 32-block permutation (the A operand of the products that reduce over the streamed index, fed from the accumulators).
// ---- shared pieces
struct X3Lane { int g, q, cg; };
// A-operand fragment of a [64 rows][64 cols] bf16 array (128-byte rows, chunks permuted by x3_f): rows 16 T + q, columns 32 ks + 8 g ..
__device__ __forceinline__ bf16x8 x3_frag(const char* arr, const X3Lane& L, int T, int ks) {
    return *reinterpret_cast<const bf16x8*>(arr + (16 * T + L.q) * X3_PITCH + ((L.cg ^ (4 * ks) ^ T) << 4));
}
// a . b with both split: a_hi b_hi + a_lo b_hi + a_hi b_lo into acc (a = A operand pair, b = B operand pair)
__device__ __forceinline__ f32x4 x3_mma3(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
}
// B-operand fragments of an OWNED 16-row tile straight from memory: lane (g, n) holds row[n][32 ks + 8 g .. + 7], hi and lo
__device__ __forceinline__ void x3_own_frags(const float* rowp, int g, bf16x8 (&fh)[2], bf16x8 (&fl)[2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(rowp + 32 * ks + 8 * g);
        const f32x4 c = *reinterpret_cast<const f32x4*>(rowp + 32 * ks + 8 * g + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bf16_t hh, ll;
            split2(a[e], hh, ll); fh[ks][e] = hh; fl[ks][e] = ll;
            split2(c[e], hh, ll); fh[ks][4 + e] = hh; fl[ks][4 + e] = ll;
        }
    }
}
// One streamed [64 rows][64 hd] fp32 chunk, as this thread loaded it (4 rows 16 vt + 4 vg + r x 4 values at 4 vq): rows past `nrows`
// become zeros; written as ROWS (rh / rl; null = skip) and TRANSPOSED with the 32-block permutation (th / tl; null = skip).
struct X3Role { int vt, vg, vq; };
__device__ __forceinline__ void x3_stage(const f32x4 (&reg)[4], int row0, int nrows, const X3Role& R, char* rh, char* rl, char* th, char* tl) {
    u32x2 h4[4], l4[4];                                         // [r]: four head-dim values of row r (row layout)
    u32x2 ht[4], lt[4];                                         // [e]: head-dim value e of the four rows (transposed layout)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        bf16x4 hv, lv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bf16_t hh, ll;
            split2(row0 + 16 * R.vt + 4 * R.vg + r < nrows ? reg[r][e] : 0.0f, hh, ll);
            hv[e] = hh; lv[e] = ll;
        }
        h4[r] = __builtin_bit_cast(u32x2, hv);
        l4[r] = __builtin_bit_cast(u32x2, lv);
    }
    if (rh) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * R.vt + 4 * R.vg + r;
            const int off = row * X3_PITCH + (((R.vq >> 1) ^ x3_f(row)) << 4) + 8 * (R.vq & 1);
            *reinterpret_cast<u32x2*>(rh + off) = h4[r];
            *reinterpret_cast<u32x2*>(rl + off) = l4[r];
        }
    }
    if (th) {
        // 4 x 4 transpose of 16-bit values inside the thread: ht[e] = (h4[0][e], h4[1][e], h4[2][e], h4[3][e])
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int w = e >> 1, sh = 16 * (e & 1);
            auto pick = [&](const u32x2 (&src)[4]) {
                u32x2 o;
                o[0] = ((src[0][w] >> sh) & 0xffffu) | (((src[1][w] >> sh) & 0xffffu) << 16);
                o[1] = ((src[2][w] >> sh) & 0xffffu) | (((src[3][w] >> sh) & 0xffffu) << 16);
                return o;
            };
            ht[e] = pick(h4);
            lt[e] = pick(l4);
        }
        // instruction k stores head-dim row e = (k + vq) & 3 (rotation by vq & 3: conflict-free with the slot permutation, see forward)
        const bool b0 = R.vq & 1, b1 = R.vq & 2;
        u32x2 hs[4], ls[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                hs[k][w] = b0 ? ht[(k + 1) & 3][w] : ht[k][w];
                ls[k][w] = b0 ? lt[(k + 1) & 3][w] : lt[k][w];
            }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                ht[k][w] = b1 ? hs[(k + 2) & 3][w] : hs[k][w];
                lt[k][w] = b1 ? ls[(k + 2) & 3][w] : ls[k][w];
            }
        const int chunk = 4 * (R.vt >> 1) + R.vg;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = 4 * R.vq + ((k + R.vq) & 3);
            const int off = row * X3_PITCH + ((chunk ^ x3_f(row)) << 4) + 8 * (R.vt & 1);
            *reinterpret_cast<u32x2*>(th + off) = ht[k];
            *reinterpret_cast<u32x2*>(tl + off) = lt[k];
        }
    }
}
// the wave's 32 x 64 fp32 block (lane (g, q) holds hd = 16 t + 4 g + r of local row 16 u + q) through its LDS slice to whole-row stores
// (dst: fp32 rows, may be null; planes: the ME_BF16X3 form of a `pcols` wide row whose columns pcol0 .. pcol0 + 63 these are, may be null)
__device__ __forceinline__ void x3_store_rows(char* stage, const f32x4 (&acc)[X3_QT][4], float mul, int lane, float* dst, int64_t ld, int row0,
                                              int nrows, uint16_t* planes = nullptr, int64_t pcols = 0, int pcol0 = 0) {
    constexpr int OP = 272;
    const int g = lane >> 4, q = lane & 15;
#pragma unroll
    for (int u = 0; u < X3_QT; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(stage + (16 * u + q) * OP + (16 * t + 4 * g) * 4) = acc[u][t] * mul;
    const int rl = lane >> 4, ch = lane & 15;
#pragma unroll
    for (int i = 0; i < 4 * X3_QT; ++i) {
        const int lr = 4 * i + rl;
        const f32x4 o = *reinterpret_cast<const f32x4*>(stage + lr * OP + ch * 16);
        if (row0 + lr < nrows) {
            if (dst) *reinterpret_cast<f32x4*>(dst + (int64_t)(row0 + lr) * ld + 4 * ch) = o;
            if (planes) store4_split3(planes + (int64_t)(row0 + lr) * 3 * pcols, pcols, pcol0 + 4 * ch, o);
        }
    }
}

constexpr int X3_LDS_DQ = 6 * X3_TILE;                     // K rows, V rows, K^T (hi, lo each)
constexpr int X3_LDS_DKV = 8 * X3_TILE + 512;              // Q rows, dO rows, Q^T, dO^T (hi, lo each) + lse, delta of the chunk

// ---- dQ (and delta): a workgroup owns 128 queries, streams {K, V}
__global__ __launch_bounds__(256) void attn_bwd_dq_x3_kernel(const float* __restrict__ qkv, int64_t ld_qkv, const float* __restrict__ o, int64_t ld_o,
                                                             const float* __restrict__ dout, int64_t ld_do, const float* __restrict__ lse,
                                                             float* __restrict__ delta, float* __restrict__ dqkv, int64_t ld_dqkv,
                                                             uint16_t* __restrict__ dqkv3, int N, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* Krh = lds;                 char* Krl = lds + X3_TILE;
    char* Vrh = lds + 2 * X3_TILE;   char* Vrl = lds + 3 * X3_TILE;
    char* Kth = lds + 4 * X3_TILE;   char* Ktl = lds + 5 * X3_TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    X3Lane L;
    L.g = lane >> 4; L.q = lane & 15; L.cg = L.g ^ ((L.q >> 1) & 7);
    const X3Role R = {tid >> 6, (tid >> 4) & 3, tid & 15};
    const int nqb = (N + X3_QB - 1) / X3_QB;
    const int qb = blockIdx.x % nqb, bh = blockIdx.x / nqb;
    const int b = bh / H, h = bh % H;
    const int C = H * X3_HD;
    const float* base = qkv + (int64_t)b * N * ld_qkv + h * X3_HD;
    const int q0 = qb * X3_QB + wave * 16 * X3_QT;
    const int ldq = (int)ld_qkv;
    const bool own = q0 < N;                                    // (wave-uniform) this wave owns at least one query

    // owned: Q and dO fragments (B operands), lse and delta of this lane's query
    bf16x8 qh[X3_QT][2], ql[X3_QT][2], doh[X3_QT][2], dol[X3_QT][2];
    float lse2[X3_QT], dl[X3_QT];
#pragma unroll
    for (int u = 0; u < X3_QT; ++u) {
        const int row = q0 + 16 * u + L.q < N ? q0 + 16 * u + L.q : N - 1;
        x3_own_frags(base + (int64_t)row * ld_qkv, L.g, qh[u], ql[u]);
        const float* dop = dout + ((int64_t)b * N + row) * ld_do + h * X3_HD;
        const float* op = o + ((int64_t)b * N + row) * ld_o + h * X3_HD;
        x3_own_frags(dop, L.g, doh[u], dol[u]);
        float part = 0.f;                                       // delta = dO . O in fp32, this lane's 16 of the 64 head-dim values
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(dop + 32 * ks + 8 * L.g + 4 * j);
                const f32x4 c = *reinterpret_cast<const f32x4*>(op + 32 * ks + 8 * L.g + 4 * j);
                part += a[0] * c[0] + a[1] * c[1] + a[2] * c[2] + a[3] * c[3];
            }
        dl[u] = x3_sum4(part);
        lse2[u] = lse[((int64_t)b * H + h) * N + row] * 1.4426950408889634f;
        if (L.g == 0 && q0 + 16 * u + L.q < N) delta[((int64_t)b * H + h) * N + row] = dl[u];
    }

    f32x4 kreg[4], vreg[4];
    auto gload = [&](int chunk) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int key = chunk * X3_KC + 16 * R.vt + 4 * R.vg + r;
            key = key < N ? key : N - 1;
            kreg[r] = *reinterpret_cast<const f32x4*>(base + (key * ldq + C + 4 * R.vq));
            vreg[r] = *reinterpret_cast<const f32x4*>(base + (key * ldq + 2 * C + 4 * R.vq));
        }
    };
    f32x4 dq[X3_QT][4];
#pragma unroll
    for (int u = 0; u < X3_QT; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) dq[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float sl2 = scale * 1.4426950408889634f;
    const int nchunk = (N + X3_KC - 1) / X3_KC;
    gload(0);
    for (int c = 0; c < nchunk; ++c) {
        __syncthreads();
        x3_stage(kreg, c * X3_KC, N, R, Krh, Krl, Kth, Ktl);
        x3_stage(vreg, c * X3_KC, N, R, Vrh, Vrl, nullptr, nullptr);
        __syncthreads();
        if (c + 1 < nchunk) gload(c + 1);
        bf16x8 dsh[X3_QT][2], dsl[X3_QT][2];                    // dS^T as B operand of the two 32-key blocks
        // round 6: 16-key tiles that lie wholly past N (the ragged last chunk: 3 of its 4 tiles at N = 197, all but one key at 257) and waves
        // whose own rows do (the last row block) skip their products -- wave-uniform branches; their dS stays zero
        const int nvalid = own ? N - c * X3_KC : 0;             // valid keys of this chunk for this wave (0: nothing to do)
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            if (16 * T >= nvalid) {
#pragma unroll
                for (int u = 0; u < X3_QT; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { dsh[u][T >> 1][4 * (T & 1) + r] = (bf16_t)0.0f; dsl[u][T >> 1][4 * (T & 1) + r] = (bf16_t)0.0f; }
                continue;
            }
            f32x4 s[X3_QT], dp[X3_QT];
#pragma unroll
            for (int u = 0; u < X3_QT; ++u) { s[u] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[u] = s[u]; }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 kh = x3_frag(Krh, L, T, ks), kl = x3_frag(Krl, L, T, ks);
                const bf16x8 vh = x3_frag(Vrh, L, T, ks), vl = x3_frag(Vrl, L, T, ks);
#pragma unroll
                for (int u = 0; u < X3_QT; ++u) {
                    s[u] = x3_mma3(kh, kl, qh[u][ks], ql[u][ks], s[u]);             // S^T[key][q]
                    dp[u] = x3_mma3(vh, vl, doh[u][ks], dol[u][ks], dp[u]);         // dP^T[key][q] = V dO^T
                }
            }
#pragma unroll
            for (int u = 0; u < X3_QT; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = c * X3_KC + 16 * T + 4 * L.g + r;
                    const float p = key < N ? __builtin_amdgcn_exp2f(s[u][r] * sl2 - lse2[u]) : 0.0f;
                    const float ds = p * (dp[u][r] - dl[u]);
                    bf16_t hh, ll;
                    split2(ds, hh, ll);
                    dsh[u][T >> 1][4 * (T & 1) + r] = hh;
                    dsl[u][T >> 1][4 * (T & 1) + r] = ll;
                }
        }
        // dQ^T[hd][q] += K^T[hd][keys] dS^T[keys][q]
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            if (32 * blk >= nvalid) continue;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x8 kh = x3_frag(Kth, L, t, blk), kl = x3_frag(Ktl, L, t, blk);
#pragma unroll
                for (int u = 0; u < X3_QT; ++u) dq[u][t] = x3_mma3(kh, kl, dsh[u][blk], dsl[u][blk], dq[u][t]);
            }
        }
    }
    __syncthreads();
    x3_store_rows(lds + wave * (16 * X3_QT * 272), dq, scale, lane, dqkv ? dqkv + (int64_t)b * N * ld_dqkv + h * X3_HD : nullptr, ld_dqkv, q0, N,
                  dqkv3 ? dqkv3 + (int64_t)b * N * 9 * C : nullptr, 3 * C, h * X3_HD);
}

// ---- dK, dV: a workgroup owns 128 keys, streams {Q, dO, lse, delta}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_dkdv_x3_kernel(const float* __restrict__ qkv, int64_t ld_qkv, const float* __restrict__ dout,
                                                               int64_t ld_do, const float* __restrict__ lse, const float* __restrict__ delta,
                                                               float* __restrict__ dqkv, int64_t ld_dqkv, uint16_t* __restrict__ dqkv3, int N, int H,
                                                               float scale) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* Qrh = lds;                 char* Qrl = lds + X3_TILE;
    char* Drh = lds + 2 * X3_TILE;   char* Drl = lds + 3 * X3_TILE;
    char* Qth = lds + 4 * X3_TILE;   char* Qtl = lds + 5 * X3_TILE;
    char* Dth = lds + 6 * X3_TILE;   char* Dtl = lds + 7 * X3_TILE;
    float* Ls = reinterpret_cast<float*>(lds + 8 * X3_TILE);            // lse (log2 units) of the chunk's 64 queries
    float* Dl = Ls + 64;                                                // delta of the chunk's 64 queries
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    X3Lane L;
    L.g = lane >> 4; L.q = lane & 15; L.cg = L.g ^ ((L.q >> 1) & 7);
    const X3Role R = {tid >> 6, (tid >> 4) & 3, tid & 15};
    const int nkb = (N + X3_QB - 1) / X3_QB;
    const int kb = blockIdx.x % nkb, bh = blockIdx.x / nkb;
    const int b = bh / H, h = bh % H;
    const int C = H * X3_HD;
    const float* base = qkv + (int64_t)b * N * ld_qkv + h * X3_HD;
    const float* dbase = dout + (int64_t)b * N * ld_do + h * X3_HD;
    const float* lrow = lse + ((int64_t)b * H + h) * N;
    const float* drow = delta + ((int64_t)b * H + h) * N;
    const int k0 = kb * X3_QB + wave * 16 * X3_QT;                      // this wave's first key (tile u: k0 + 16 u + n)
    const int ldq = (int)ld_qkv, ldd = (int)ld_do;
    const bool own = k0 < N;                                            // (wave-uniform) this wave owns at least one key

    bf16x8 kh[X3_QT][2], kl[X3_QT][2], vh[X3_QT][2], vl[X3_QT][2];      // owned K, V fragments (B operands)
#pragma unroll
    for (int u = 0; u < X3_QT; ++u) {
        const int row = k0 + 16 * u + L.q < N ? k0 + 16 * u + L.q : N - 1;
        x3_own_frags(base + (int64_t)row * ld_qkv + C, L.g, kh[u], kl[u]);
        x3_own_frags(base + (int64_t)row * ld_qkv + 2 * C, L.g, vh[u], vl[u]);
    }
    f32x4 qreg[4], dreg[4];
    float sreg = 0.f;                                                   // thread < 64: lse, 64 .. 127: delta of chunk query (tid & 63)
    auto gload = [&](int chunk) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int qi = chunk * X3_KC + 16 * R.vt + 4 * R.vg + r;
            qi = qi < N ? qi : N - 1;
            qreg[r] = *reinterpret_cast<const f32x4*>(base + (qi * ldq + 4 * R.vq));
            dreg[r] = *reinterpret_cast<const f32x4*>(dbase + (qi * ldd + 4 * R.vq));
        }
        if (tid < 128) {
            int qi = chunk * X3_KC + (tid & 63);
            qi = qi < N ? qi : N - 1;
            sreg = tid < 64 ? lrow[qi] : drow[qi];
        }
    };
    f32x4 dk[X3_QT][4], dv[X3_QT][4];
#pragma unroll
    for (int u = 0; u < X3_QT; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) { dk[u][t] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[u][t] = dk[u][t]; }
    const float sl2 = scale * 1.4426950408889634f;
    const int nchunk = (N + X3_KC - 1) / X3_KC;
    gload(0);
    for (int c = 0; c < nchunk; ++c) {
        __syncthreads();
        x3_stage(qreg, c * X3_KC, N, R, Qrh, Qrl, Qth, Qtl);
        x3_stage(dreg, c * X3_KC, N, R, Drh, Drl, Dth, Dtl);
        if (tid < 64) Ls[tid] = sreg * 1.4426950408889634f;
        else if (tid < 128) Dl[tid - 64] = sreg;
        __syncthreads();
        if (c + 1 < nchunk) gload(c + 1);
        // (round 6: query tiles wholly past N and waves without a key of their own skip their products, as in the dQ kernel)
        const int nvalid = own ? N - c * X3_KC : 0;                     // valid queries of this chunk for this wave
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            if (32 * blk >= nvalid) continue;
            bf16x8 ph[X3_QT], pl[X3_QT], dsh[X3_QT], dsl[X3_QT];        // P and dS of this 32-query block as B operands
#pragma unroll
            for (int Tl = 0; Tl < 2; ++Tl) {
                const int T = 2 * blk + Tl;
                if (16 * T >= nvalid) {
#pragma unroll
                    for (int u = 0; u < X3_QT; ++u)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            ph[u][4 * Tl + r] = (bf16_t)0.0f; pl[u][4 * Tl + r] = (bf16_t)0.0f;
                            dsh[u][4 * Tl + r] = (bf16_t)0.0f; dsl[u][4 * Tl + r] = (bf16_t)0.0f;
                        }
                    continue;
                }
                f32x4 s[X3_QT], dp[X3_QT];
#pragma unroll
                for (int u = 0; u < X3_QT; ++u) { s[u] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[u] = s[u]; }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 qh_ = x3_frag(Qrh, L, T, ks), ql_ = x3_frag(Qrl, L, T, ks);
                    const bf16x8 dh_ = x3_frag(Drh, L, T, ks), dl_ = x3_frag(Drl, L, T, ks);
#pragma unroll
                    for (int u = 0; u < X3_QT; ++u) {
                        s[u] = x3_mma3(qh_, ql_, kh[u][ks], kl[u][ks], s[u]);       // S[q][key]
                        dp[u] = x3_mma3(dh_, dl_, vh[u][ks], vl[u][ks], dp[u]);     // dP[q][key] = dO V^T
                    }
                }
                const f32x4 ls4 = *reinterpret_cast<const f32x4*>(Ls + 16 * T + 4 * L.g);
                const f32x4 dl4 = *reinterpret_cast<const f32x4*>(Dl + 16 * T + 4 * L.g);
#pragma unroll
                for (int u = 0; u < X3_QT; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int qi = c * X3_KC + 16 * T + 4 * L.g + r;
                        const float p = qi < N ? __builtin_amdgcn_exp2f(s[u][r] * sl2 - ls4[r]) : 0.0f;
                        const float ds = p * (dp[u][r] - dl4[r]);
                        bf16_t hh, ll;
                        split2(p, hh, ll);
                        ph[u][4 * Tl + r] = hh; pl[u][4 * Tl + r] = ll;
                        split2(ds, hh, ll);
                        dsh[u][4 * Tl + r] = hh; dsl[u][4 * Tl + r] = ll;
                    }
            }
            // dV^T[hd][key] += dO^T[hd][q] P[q][key];   dK^T[hd][key] += Q^T[hd][q] dS[q][key]
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x8 dh_ = x3_frag(Dth, L, t, blk), dl_ = x3_frag(Dtl, L, t, blk);
                const bf16x8 qh_ = x3_frag(Qth, L, t, blk), ql_ = x3_frag(Qtl, L, t, blk);
#pragma unroll
                for (int u = 0; u < X3_QT; ++u) {
                    dv[u][t] = x3_mma3(dh_, dl_, ph[u], pl[u], dv[u][t]);
                    dk[u][t] = x3_mma3(qh_, ql_, dsh[u], dsl[u], dk[u][t]);
                }
            }
        }
    }
    __syncthreads();
    char* stage = lds + wave * (16 * X3_QT * 272);
    float* dst = dqkv ? dqkv + (int64_t)b * N * ld_dqkv + h * X3_HD : nullptr;
    uint16_t* pl3 = dqkv3 ? dqkv3 + (int64_t)b * N * 9 * C : nullptr;
    x3_store_rows(stage, dk, scale, lane, dst ? dst + C : nullptr, ld_dqkv, k0, N, pl3, 3 * C, C + h * X3_HD);
    x3_store_rows(stage, dv, 1.0f, lane, dst ? dst + 2 * C : nullptr, ld_dqkv, k0, N, pl3, 3 * C, 2 * C + h * X3_HD);
}

}  // namespace

extern "C" int me_attention_fwd_x3(const float* qkv, int64_t ld_qkv, float* out, int64_t ld_out, void* out3, float* lse, int B, int N, int H,
                                   int head_dim, float scale, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ProfScope prof(ME_PROF_ATTN_FWD, ME_BF16X3, (int64_t)B * H, N, head_dim, stream);
    ME_CHECK_ARG(qkv && (out || out3), "me_attention_fwd_x3: null pointer");
    ME_CHECK_ARG(B > 0 && N > 0 && H > 0, "me_attention_fwd_x3: bad shape B=%d N=%d H=%d", B, N, H);
    if (head_dim != X3_HD) {
        me_set_error("me_attention_fwd_x3: head_dim %d (64 only; use me_attention_fwd with ME_F32)", head_dim);
        return ME_ERR_UNSUPPORTED;
    }
    ME_CHECK_ARG(ld_qkv % 4 == 0 && ld_qkv >= 3 * H * head_dim && (!out || (ld_out % 4 == 0 && ld_out >= H * head_dim)), "me_attention_fwd_x3: bad strides");
    ME_CHECK_ARG(((uintptr_t)qkv | (uintptr_t)out) % 16 == 0 && (uintptr_t)out3 % 8 == 0, "me_attention_fwd_x3: alignment");
    const int64_t nwg = (int64_t)B * H * ((N + X3_QB - 1) / X3_QB);
    ME_CHECK_ARG(nwg < (int64_t)0x7fffffff, "me_attention_fwd_x3: too many workgroups");
    ME_CHECK_ARG((int64_t)N * ld_qkv < (int64_t)0x7fffffff, "me_attention_fwd_x3: N * ld_qkv must fit 31 bits (row offsets are 32-bit)");
    hipLaunchKernelGGL(attn_fwd_x3_kernel, dim3((unsigned)nwg), dim3(256), 0, stream, qkv, ld_qkv, out, ld_out, reinterpret_cast<uint16_t*>(out3), lse, N,
                       H, scale);
    ME_CHECK_LAUNCH("me_attention_fwd_x3");
    return ME_OK;
}

extern "C" int me_attention_bwd_x3(const float* qkv, int64_t ld_qkv, const float* out, int64_t ld_out, const float* dout, int64_t ld_dout,
                                   const float* lse, float* delta, float* dqkv, int64_t ld_dqkv, void* dqkv3, int B, int N, int H, int head_dim,
                                   float scale, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ProfScope prof(ME_PROF_ATTN_BWD, ME_BF16X3, (int64_t)B * H, N, head_dim, stream);
    ME_CHECK_ARG(qkv && out && dout && lse && delta && (dqkv || dqkv3), "me_attention_bwd_x3: null pointer");
    ME_CHECK_ARG(B > 0 && N > 0 && H > 0, "me_attention_bwd_x3: bad shape B=%d N=%d H=%d", B, N, H);
    if (head_dim != X3_HD) {
        me_set_error("me_attention_bwd_x3: head_dim %d (64 only; use me_attention_bwd with ME_F32)", head_dim);
        return ME_ERR_UNSUPPORTED;
    }
    const int C = H * head_dim;
    ME_CHECK_ARG(ld_qkv % 4 == 0 && ld_qkv >= 3 * C && (!dqkv || (ld_dqkv % 4 == 0 && ld_dqkv >= 3 * C)) && ld_out % 4 == 0 && ld_out >= C && ld_dout % 4 == 0 &&
                     ld_dout >= C, "me_attention_bwd_x3: bad strides");
    ME_CHECK_ARG(((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)dout | (uintptr_t)dqkv) % 16 == 0 && (uintptr_t)dqkv3 % 8 == 0, "me_attention_bwd_x3: alignment");
    ME_CHECK_ARG((int64_t)N * ld_qkv < (int64_t)0x7fffffff && (int64_t)N * ld_dout < (int64_t)0x7fffffff, "me_attention_bwd_x3: N * ld must fit 31 bits");
    const int64_t nwg = (int64_t)B * H * ((N + X3_QB - 1) / X3_QB);
    ME_CHECK_ARG(nwg < (int64_t)0x7fffffff, "me_attention_bwd_x3: too many workgroups");
    static OncePerDevice once;
    if (once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS_DQ);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS_DKV);
    }
    hipLaunchKernelGGL(attn_bwd_dq_x3_kernel, dim3((unsigned)nwg), dim3(256), X3_LDS_DQ, stream, qkv, ld_qkv, out, ld_out, dout, ld_dout, lse, delta, dqkv,
                       ld_dqkv, reinterpret_cast<uint16_t*>(dqkv3), N, H, scale);
    ME_CHECK_LAUNCH("me_attention_bwd_x3(dq)");
    hipLaunchKernelGGL(attn_bwd_dkdv_x3_kernel, dim3((unsigned)nwg), dim3(256), X3_LDS_DKV, stream, qkv, ld_qkv, dout, ld_dout, lse, delta, dqkv, ld_dqkv,
                       reinterpret_cast<uint16_t*>(dqkv3), N, H, scale);
    ME_CHECK_LAUNCH("me_attention_bwd_x3(dkdv)");
    return ME_OK;
}
