// attention_tiny.hip -- self-attention for very short sequences (N <= 64 tokens, head_dim <= 64, bf16): one small workgroup per
// (batch, head) -- ONE WAVE PER 16 TOKENS (1, 2 or 4 waves) -- with the whole head resident in LDS.
//
// Who runs this: the reference's Tabular recipe drives the encoder with one token per column (N ~ 14 .. 20, batch 256:
// Tabular/run_experiments/adult/adult_meta-transformer.py:103-161), the Graph recipe with N ~ 30 .. 70 node / edge tokens and 32 heads
// of 24 channels (Graph/scripts/pcqv2-metatransformer_fixed.sh, tokengt_graph_encoder.py:191-206).  The tiled kernels of attention.hip
// give every (batch, head) a 128-query x 64-key workgroup tile that is 1 .. 25 % full at these sizes and three launches per backward:
// 23 us forward / 93 .. 151 us backward per layer, a quarter of those recipes' step.  Math: attention.py:28-35 (S = scale Q K^T, softmax,
// O = P V) and what autograd derives from it.
//
// A workgroup owns a head: Q, K, V (and dO) are staged once into LDS, row-major [token][d] and -- where a product contracts over
// tokens -- transposed [d][token]; wave w owns token tile w (its 16 queries in the score phase, its 16 tokens' gradient rows at the end);
// every MFMA operand is then one 16-byte row read (lane (r, g) = (l & 15, l >> 4): row r, elements
// 8 g .. 8 g + 7 of a 32-long k-step).  All products are arranged so that the result tile D[m][n] has n = the token a lane owns (r) and
// m = four consecutive indices per lane (4 g + i):
//   S^T[key][q] = K Q^T   -> softmax statistics of query r are in-register sums over (tile, i) + two cross-lane steps (g);
//                            P[q][4 consecutive keys] goes to LDS as one 8-byte write
//   O^T[d][q]   = V^T P^T -> a lane holds 4 consecutive channels of its query: 8-byte global stores
//   backward:  dP^T = V dO^T (same layout as S^T), dS^T = P^T o (dP^T - delta_q);
//              dV^T = dO^T P, dK^T = Q^T dS, dQ^T = K^T dS^T-transposed  -- each [d][token], 8-byte stores.
// Two or three workgroup barriers, no atomics, deterministic.  LDS per head 5 .. 65 KB by (N, head_dim) class, i.e. 2 .. 16 heads in
// flight per CU; the arithmetic is a few dozen MFMAs per wave -- the kernel is one memory round trip long.  (One wave per HEAD was
// measured first: it loses to the tiled forward at N > 32, where a single wave's staging + 16-tile score chain is too long.)
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

template <int HD, int NMAX> struct TinyCfg {
    static constexpr int NQ = NMAX / 16;                 // token tiles
    static constexpr int ND = HD / 16;                   // channel tiles
    static constexpr int KD = HD / 32;                   // k-steps when the contraction runs over channels
    static constexpr int KN = (NMAX + 31) / 32;          // ... over tokens (NMAX = 16: half a step, lanes g >= 2 supply zeros)
    static constexpr int CPR = HD / 8;                   // 16-byte chunks per token row
    static constexpr int PR = HD * 2 + 16;               // row pitch in bytes of [token][d] arrays
    static constexpr int PT = NMAX * 2 + 16;             // row pitch of [d][token] and [token][token] arrays
    static constexpr int ROWMAJ = NMAX * PR;             // one [token][d] array
    static constexpr int TRANS = HD * PT;                // one [d][token] array
    static constexpr int SQUARE = NMAX * PT;             // one [token][token] array
    static constexpr int FWD_LDS = 2 * ROWMAJ + TRANS + SQUARE;
    static constexpr int BWD_A = 4 * ROWMAJ > 3 * SQUARE ? 4 * ROWMAJ : 3 * SQUARE;      // {Q, K, V, dO} rows, later {P^T, dS^T, dS}
    static constexpr int BWD_LDS = BWD_A + 3 * TRANS + 2 * NMAX * 4;
};

__device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// fragment of a k-step over CHANNELS: row `row` of a [token][d] array
__device__ __forceinline__ bf16x8 frag_d(const char* base, int pitch, int row, int ks, int g) {
    return *reinterpret_cast<const bf16x8*>(base + row * pitch + (ks * 32 + 8 * g) * 2);
}
// fragment of a k-step over TOKENS: row `row` of a [..][token] array; with 16 tokens the upper half of the step is zeros
template <int NMAX> __device__ __forceinline__ bf16x8 frag_t(const char* base, int pitch, int row, int ks, int g) {
    if (NMAX == 16 && g >= 2) {
        const u32x4 z = {0u, 0u, 0u, 0u};
        return __builtin_bit_cast(bf16x8, z);
    }
    return *reinterpret_cast<const bf16x8*>(base + row * pitch + (ks * 32 + 8 * g) * 2);
}

__device__ __forceinline__ uint32_t pack2(float a, float b) {
    const bf16_t x = (bf16_t)a, y = (bf16_t)b;
    return (uint32_t)__builtin_bit_cast(uint16_t, x) | ((uint32_t)__builtin_bit_cast(uint16_t, y) << 16);
}

// 16-byte chunk c8 of token row `row` of one of the head's operands, zeros past the sequence / the head's channels
__device__ __forceinline__ u32x4 load_chunk(const bf16_t* base, int64_t ld, int row, int c8, int N, int hd) {
    const u32x4 z = {0u, 0u, 0u, 0u};
    if (row >= N || c8 * 8 >= hd) return z;
    return *reinterpret_cast<const u32x4*>(base + (int64_t)row * ld + c8 * 8);
}
__device__ __forceinline__ void put_row(char* arr, int pitch, int row, int c8, u32x4 v) { *reinterpret_cast<u32x4*>(arr + row * pitch + c8 * 16) = v; }
// the same chunk into the transposed array: element e of the chunk is channel 8 c8 + e of token `row`
__device__ __forceinline__ void put_trans(char* arr, int pitch, int row, int c8, u32x4 v) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint16_t x = (uint16_t)(v[e >> 1] >> (16 * (e & 1)));
        *reinterpret_cast<uint16_t*>(arr + (8 * c8 + e) * pitch + row * 2) = x;
    }
}

template <int HD, int NMAX>
__global__ __launch_bounds__(NMAX * 4) void attn_tiny_fwd_kernel(const bf16_t* __restrict__ qkv, int64_t ld, bf16_t* __restrict__ out, int64_t ldo,
                                                           float* __restrict__ lse, int N, int H, int hd, float scale) {
    typedef TinyCfg<HD, NMAX> C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem;
    char* Ks = Qs + C::ROWMAJ;
    char* Vt = Ks + C::ROWMAJ;
    char* Ps = Vt + C::TRANS;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
    const int tq = __builtin_amdgcn_readfirstlane(tid >> 6);          // this wave's token tile
    const int64_t bh = blockIdx.x;
    const int b = (int)(bh / H), head = (int)(bh - (int64_t)b * H);
    const int Cdim = H * hd;
    const bf16_t* qp = qkv + (int64_t)b * N * ld + head * hd;
#pragma unroll
    for (int c = tid; c < NMAX * C::CPR; c += NMAX * 4) {
        const int row = c / C::CPR, c8 = c % C::CPR;
        put_row(Qs, C::PR, row, c8, load_chunk(qp, ld, row, c8, N, hd));
        put_row(Ks, C::PR, row, c8, load_chunk(qp + Cdim, ld, row, c8, N, hd));
        put_trans(Vt, C::PT, row, c8, load_chunk(qp + 2 * Cdim, ld, row, c8, N, hd));
    }
    __syncthreads();
    const float sl = scale * LOG2E;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    {
        bf16x8 qf[C::KD];
#pragma unroll
        for (int ks = 0; ks < C::KD; ++ks) qf[ks] = frag_d(Qs, C::PR, 16 * tq + r, ks, g);
        f32x4 st[C::NQ];
        float m = -INFINITY;
#pragma unroll
        for (int tk = 0; tk < C::NQ; ++tk) {
            f32x4 a = zero;
#pragma unroll
            for (int ks = 0; ks < C::KD; ++ks) a = mma(frag_d(Ks, C::PR, 16 * tk + r, ks, g), qf[ks], a);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = (16 * tk + 4 * g + i < N) ? a[i] * sl : -INFINITY;
                m = fmaxf(m, a[i]);
            }
            st[tk] = a;
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int tk = 0; tk < C::NQ; ++tk)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                st[tk][i] = __builtin_amdgcn_exp2f(st[tk][i] - m);
                sum += st[tk][i];
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        const int q = 16 * tq + r;
#pragma unroll
        for (int tk = 0; tk < C::NQ; ++tk) {
            const u32x2 w = {pack2(st[tk][0] * inv, st[tk][1] * inv), pack2(st[tk][2] * inv, st[tk][3] * inv)};
            *reinterpret_cast<u32x2*>(Ps + q * C::PT + (16 * tk + 4 * g) * 2) = w;
        }
        if (lse && g == 0 && q < N) lse[bh * N + q] = (m + __builtin_amdgcn_logf(sum)) * LN2;
    }
    __syncthreads();
    bf16_t* op = out + (int64_t)b * N * ldo + head * hd;
    {
        bf16x8 pf[C::KN];
#pragma unroll
        for (int ks = 0; ks < C::KN; ++ks) pf[ks] = frag_t<NMAX>(Ps, C::PT, 16 * tq + r, ks, g);
        const int q = 16 * tq + r;
#pragma unroll
        for (int td = 0; td < C::ND; ++td) {
            f32x4 a = zero;
#pragma unroll
            for (int ks = 0; ks < C::KN; ++ks) a = mma(frag_t<NMAX>(Vt, C::PT, 16 * td + r, ks, g), pf[ks], a);
            const int d0 = 16 * td + 4 * g;
            if (q < N && d0 < hd) {
                const u32x2 w = {pack2(a[0], a[1]), pack2(a[2], a[3])};
                *reinterpret_cast<u32x2*>(op + (int64_t)q * ldo + d0) = w;
            }
        }
    }
}

template <int HD, int NMAX>
__global__ __launch_bounds__(NMAX * 4) void attn_tiny_bwd_kernel(const bf16_t* __restrict__ qkv, int64_t ld, const bf16_t* __restrict__ out, int64_t ldo,
                                                           const bf16_t* __restrict__ dout, int64_t lddo, const float* __restrict__ lse,
                                                           float* __restrict__ delta, bf16_t* __restrict__ dqkv, int64_t lddq, int N, int H, int hd,
                                                           float scale) {
    typedef TinyCfg<HD, NMAX> C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem;                       // phase 1: rows of Q, K, V, dO
    char* Ks = Qs + C::ROWMAJ;
    char* Vs = Ks + C::ROWMAJ;
    char* Ds = Vs + C::ROWMAJ;
    char* Pt = smem;                       // phase 2 (same space): P^T [key][q], dS^T [key][q], dS [q][key]
    char* dSt = Pt + C::SQUARE;
    char* dSs = dSt + C::SQUARE;
    char* Qt = smem + C::BWD_A;            // both phases: Q^T, K^T, dO^T [d][token]
    char* Kt = Qt + C::TRANS;
    char* Dt = Kt + C::TRANS;
    float* lse_s = reinterpret_cast<float*>(Dt + C::TRANS);
    float* del_s = lse_s + NMAX;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
    const int tq = __builtin_amdgcn_readfirstlane(tid >> 6);          // this wave's token tile
    const int64_t bh = blockIdx.x;
    const int b = (int)(bh / H), head = (int)(bh - (int64_t)b * H);
    const int Cdim = H * hd;
    const bf16_t* qp = qkv + (int64_t)b * N * ld + head * hd;
    const bf16_t* dop = dout + (int64_t)b * N * lddo + head * hd;
    const bf16_t* outp = out + (int64_t)b * N * ldo + head * hd;
#pragma unroll
    for (int c = tid; c < NMAX * C::CPR; c += NMAX * 4) {
        const int row = c / C::CPR, c8 = c % C::CPR;
        const u32x4 q = load_chunk(qp, ld, row, c8, N, hd), k = load_chunk(qp + Cdim, ld, row, c8, N, hd);
        const u32x4 v = load_chunk(qp + 2 * Cdim, ld, row, c8, N, hd), d = load_chunk(dop, lddo, row, c8, N, hd);
        const u32x4 o = load_chunk(outp, ldo, row, c8, N, hd);
        put_row(Qs, C::PR, row, c8, q); put_row(Ks, C::PR, row, c8, k); put_row(Vs, C::PR, row, c8, v); put_row(Ds, C::PR, row, c8, d);
        put_trans(Qt, C::PT, row, c8, q); put_trans(Kt, C::PT, row, c8, k); put_trans(Dt, C::PT, row, c8, d);
        // delta = dO . O: the row's CPR chunks sit in CPR neighbouring lanes
        float part = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            part += __uint_as_float(o[e] << 16) * __uint_as_float(d[e] << 16) + __uint_as_float(o[e] & 0xffff0000u) * __uint_as_float(d[e] & 0xffff0000u);
#pragma unroll
        for (int s = 1; s < C::CPR; s <<= 1) part += __shfl_xor(part, s, 64);
        if (c8 == 0) {
            del_s[row] = part;
            if (delta && row < N) delta[bh * N + row] = part;
        }
    }
    if (tid < NMAX) lse_s[tid] = tid < N ? lse[bh * N + tid] * LOG2E : INFINITY;
    __syncthreads();
    const float sl = scale * LOG2E;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 p[C::NQ], ds[C::NQ];          // [key tile]: lane holds keys 16 tk + 4 g + i of query 16 tq + r
    {
        bf16x8 qf[C::KD], df[C::KD];
#pragma unroll
        for (int ks = 0; ks < C::KD; ++ks) {
            qf[ks] = frag_d(Qs, C::PR, 16 * tq + r, ks, g);
            df[ks] = frag_d(Ds, C::PR, 16 * tq + r, ks, g);
        }
        const float lq = lse_s[16 * tq + r], dq = del_s[16 * tq + r];
#pragma unroll
        for (int tk = 0; tk < C::NQ; ++tk) {
            f32x4 s = zero, dp = zero;
#pragma unroll
            for (int ks = 0; ks < C::KD; ++ks) {
                s = mma(frag_d(Ks, C::PR, 16 * tk + r, ks, g), qf[ks], s);
                dp = mma(frag_d(Vs, C::PR, 16 * tk + r, ks, g), df[ks], dp);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float pe = (16 * tk + 4 * g + i < N) ? __builtin_amdgcn_exp2f(s[i] * sl - lq) : 0.f;      // (padded queries: lq = +inf -> 0)
                p[tk][i] = pe;
                ds[tk][i] = pe * (dp[i] - dq);
            }
        }
    }
    __syncthreads();          // the row arrays are dead: their space takes P^T, dS^T, dS
#pragma unroll
    for (int tk = 0; tk < C::NQ; ++tk) {
        const int q = 16 * tq + r, k0 = 16 * tk + 4 * g;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16_t pv = (bf16_t)p[tk][i], dv = (bf16_t)ds[tk][i];
            *reinterpret_cast<bf16_t*>(Pt + (k0 + i) * C::PT + q * 2) = pv;
            *reinterpret_cast<bf16_t*>(dSt + (k0 + i) * C::PT + q * 2) = dv;
        }
        const u32x2 w = {pack2(ds[tk][0], ds[tk][1]), pack2(ds[tk][2], ds[tk][3])};
        *reinterpret_cast<u32x2*>(dSs + q * C::PT + k0 * 2) = w;
    }
    __syncthreads();
    bf16_t* gp = dqkv + (int64_t)b * N * lddq + head * hd;
    // token tile tt = this wave's: dQ^T[d][q] = K^T dS-rows, dK^T[d][key] = Q^T dS^T-rows, dV^T[d][key] = dO^T P^T-rows
    {
        const int tt = tq;
        bf16x8 sq[C::KN], sk[C::KN], pk[C::KN];
#pragma unroll
        for (int ks = 0; ks < C::KN; ++ks) {
            sq[ks] = frag_t<NMAX>(dSs, C::PT, 16 * tt + r, ks, g);
            sk[ks] = frag_t<NMAX>(dSt, C::PT, 16 * tt + r, ks, g);
            pk[ks] = frag_t<NMAX>(Pt, C::PT, 16 * tt + r, ks, g);
        }
        const int tok = 16 * tt + r;
#pragma unroll
        for (int td = 0; td < C::ND; ++td) {
            f32x4 aq = zero, ak = zero, av = zero;
#pragma unroll
            for (int ks = 0; ks < C::KN; ++ks) {
                aq = mma(frag_t<NMAX>(Kt, C::PT, 16 * td + r, ks, g), sq[ks], aq);
                ak = mma(frag_t<NMAX>(Qt, C::PT, 16 * td + r, ks, g), sk[ks], ak);
                av = mma(frag_t<NMAX>(Dt, C::PT, 16 * td + r, ks, g), pk[ks], av);
            }
            const int d0 = 16 * td + 4 * g;
            if (tok < N && d0 < hd) {
                bf16_t* row = gp + (int64_t)tok * lddq + d0;
                const u32x2 wq = {pack2(aq[0] * scale, aq[1] * scale), pack2(aq[2] * scale, aq[3] * scale)};
                const u32x2 wk = {pack2(ak[0] * scale, ak[1] * scale), pack2(ak[2] * scale, ak[3] * scale)};
                const u32x2 wv = {pack2(av[0], av[1]), pack2(av[2], av[3])};
                *reinterpret_cast<u32x2*>(row) = wq;
                *reinterpret_cast<u32x2*>(row + Cdim) = wk;
                *reinterpret_cast<u32x2*>(row + 2 * Cdim) = wv;
            }
        }
    }
}

template <int HD, int NMAX>
int launch_fwd(const void* qkv, int64_t ld, void* out, int64_t ldo, float* lse, int B, int N, int H, int hd, float scale, hipStream_t stream) {
    typedef TinyCfg<HD, NMAX> C;
    static OncePerDevice once;
    if (once.need()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_tiny_fwd_kernel<HD, NMAX>), hipFuncAttributeMaxDynamicSharedMemorySize, C::FWD_LDS);
    hipLaunchKernelGGL((attn_tiny_fwd_kernel<HD, NMAX>), dim3((unsigned)((int64_t)B * H)), dim3(NMAX * 4), C::FWD_LDS, stream, reinterpret_cast<const bf16_t*>(qkv), ld,
                       reinterpret_cast<bf16_t*>(out), ldo, lse, N, H, hd, scale);
    ME_CHECK_LAUNCH("me_attention_fwd(tiny)");
    return ME_OK;
}
template <int HD, int NMAX>
int launch_bwd(const void* qkv, int64_t ld, const void* out, int64_t ldo, const void* dout, int64_t lddo, const float* lse, float* delta, void* dqkv,
               int64_t lddq, int B, int N, int H, int hd, float scale, hipStream_t stream) {
    typedef TinyCfg<HD, NMAX> C;
    static OncePerDevice once;
    if (once.need()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_tiny_bwd_kernel<HD, NMAX>), hipFuncAttributeMaxDynamicSharedMemorySize, C::BWD_LDS);
    hipLaunchKernelGGL((attn_tiny_bwd_kernel<HD, NMAX>), dim3((unsigned)((int64_t)B * H)), dim3(NMAX * 4), C::BWD_LDS, stream, reinterpret_cast<const bf16_t*>(qkv), ld,
                       reinterpret_cast<const bf16_t*>(out), ldo, reinterpret_cast<const bf16_t*>(dout), lddo, lse, delta, reinterpret_cast<bf16_t*>(dqkv), lddq,
                       N, H, hd, scale);
    ME_CHECK_LAUNCH("me_attention_bwd(tiny)");
    return ME_OK;
}

}  // namespace

// bf16, no dropout, N <= 64, head_dim <= 64 and a multiple of 8, 16-byte aligned head rows (ld, ld_out, ld_dout, ld_dqkv and head_dim multiples of 8)
bool attn_tiny_ok(int64_t ld_qkv, int64_t ld_out, int B, int N, int H, int hd) {
    return N >= 1 && N <= 64 && hd >= 8 && hd <= 64 && hd % 8 == 0 && ld_qkv % 8 == 0 && ld_out % 8 == 0 && (int64_t)B * H < (1ll << 31);
}

int launch_attn_tiny_fwd(const void* qkv, int64_t ld, void* out, int64_t ldo, float* lse, int B, int N, int H, int hd, float scale, hipStream_t stream) {
#define TINY_FWD(HD_, NM_) return launch_fwd<HD_, NM_>(qkv, ld, out, ldo, lse, B, N, H, hd, scale, stream)
    if (hd <= 32) {
        if (N <= 16) TINY_FWD(32, 16);
        if (N <= 32) TINY_FWD(32, 32);
        TINY_FWD(32, 64);
    }
    if (N <= 16) TINY_FWD(64, 16);
    if (N <= 32) TINY_FWD(64, 32);
    TINY_FWD(64, 64);
#undef TINY_FWD
}

int launch_attn_tiny_bwd(const void* qkv, int64_t ld, const void* out, int64_t ldo, const void* dout, int64_t lddo, const float* lse, float* delta,
                         void* dqkv, int64_t lddq, int B, int N, int H, int hd, float scale, hipStream_t stream) {
#define TINY_BWD(HD_, NM_) return launch_bwd<HD_, NM_>(qkv, ld, out, ldo, dout, lddo, lse, delta, dqkv, lddq, B, N, H, hd, scale, stream)
    if (hd <= 32) {
        if (N <= 16) TINY_BWD(32, 16);
        if (N <= 32) TINY_BWD(32, 32);
        TINY_BWD(32, 64);
    }
    if (N <= 16) TINY_BWD(64, 16);
    if (N <= 32) TINY_BWD(64, 32);
    TINY_BWD(64, 64);
#undef TINY_BWD
}
