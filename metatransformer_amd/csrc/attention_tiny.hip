// attention_tiny.hip -- self-attention for very short sequences (N <= 64 tokens, head_dim <= 64; bf16, and fp32 on the exact-fp32 MFMA):
// one small workgroup per (batch, head) -- ONE WAVE PER 16 TOKENS (1, 2 or 4 waves) -- with the whole head resident in LDS.
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
//
// fp32 (the reference's default arithmetic; Tabular runs it): the same kernels on v_mfma_f32_16x16x4_f32 -- exact products, fp32 P and dS.
// A lane's 16-byte fragment is then 4 floats = its share of FOUR k-steps of 4 (the contraction order inside a 16-long block is permuted
// identically on both operands), so the LDS addressing is byte for byte the bf16 one: row * pitch + 64 * kstep + 16 * g.
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// element traits: T = bf16_t (v_mfma_f32_16x16x32_bf16, 8 elements per 16-byte fragment) or float (v_mfma_f32_16x16x4_f32 x 4)
template <typename T> struct Tr;
template <> struct Tr<bf16_t> {
    static constexpr int ES = 2;
    static __device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ void put1(char* p, float v) { *reinterpret_cast<bf16_t*>(p) = (bf16_t)v; }
    static __device__ __forceinline__ void put4(void* p, f32x4 v) {          // four consecutive elements (8 bytes)
        const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        *reinterpret_cast<bf16x4*>(p) = o;
    }
    static __device__ __forceinline__ float dot(u32x4 a, u32x4 b) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            s += __uint_as_float(a[e] << 16) * __uint_as_float(b[e] << 16) + __uint_as_float(a[e] & 0xffff0000u) * __uint_as_float(b[e] & 0xffff0000u);
        return s;
    }
};
template <> struct Tr<float> {
    static constexpr int ES = 4;
    static __device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[j]), __uint_as_float(b[j]), c, 0, 0, 0);
        return c;
    }
    static __device__ __forceinline__ void put1(char* p, float v) { *reinterpret_cast<float*>(p) = v; }
    static __device__ __forceinline__ void put4(void* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
    static __device__ __forceinline__ float dot(u32x4 a, u32x4 b) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) s += __uint_as_float(a[e]) * __uint_as_float(b[e]);
        return s;
    }
};

template <typename T, int HD, int NMAX> struct TinyCfg {
    static constexpr int ES = Tr<T>::ES;
    static constexpr int NQ = NMAX / 16;                 // token tiles
    static constexpr int ND = HD / 16;                   // channel tiles
    static constexpr int KD = HD * ES / 64;              // 64-byte k-steps when the contraction runs over channels
    static constexpr int KN = (NMAX * ES + 63) / 64;     // ... over tokens (bf16, NMAX = 16: half a step, lanes g >= 2 supply zeros)
    static constexpr bool HALF = NMAX * ES < 64;
    static constexpr int CPR = HD * ES / 16;             // 16-byte chunks per token row
    static constexpr int EPC = 16 / ES;                  // elements per chunk
    static constexpr int PR = HD * ES + 16;              // row pitch in bytes of [token][d] arrays
    static constexpr int PT = NMAX * ES + 16;            // row pitch of [d][token] and [token][token] arrays
    static constexpr int ROWMAJ = NMAX * PR;             // one [token][d] array
    static constexpr int TRANS = HD * PT;                // one [d][token] array
    static constexpr int SQUARE = NMAX * PT;             // one [token][token] array
    static constexpr int FWD_LDS = 2 * ROWMAJ + TRANS + SQUARE;
    static constexpr int BWD_A = 4 * ROWMAJ > 3 * SQUARE ? 4 * ROWMAJ : 3 * SQUARE;      // {Q, K, V, dO} rows, later {P^T, dS^T, dS}
    static constexpr int BWD_LDS = BWD_A + 3 * TRANS + 2 * NMAX * 4;
};

// the 16-byte fragment of lane group g for k-step ks of row `row` (either element type: 64 bytes per k-step, 16 per lane group)
__device__ __forceinline__ u32x4 frag(const char* base, int pitch, int row, int ks, int g) {
    return *reinterpret_cast<const u32x4*>(base + row * pitch + ks * 64 + 16 * g);
}
// ... of a contraction over TOKENS: with 16 bf16 tokens a row is half a k-step, lane groups 2 and 3 supply zeros
template <bool HALF> __device__ __forceinline__ u32x4 frag_t(const char* base, int pitch, int row, int ks, int g) {
    if (HALF && g >= 2) return u32x4{0u, 0u, 0u, 0u};
    return *reinterpret_cast<const u32x4*>(base + row * pitch + ks * 64 + 16 * g);
}

// 16-byte chunk c of token row `row` of one of the head's operands, zeros past the sequence / the head's channels
template <typename T> __device__ __forceinline__ u32x4 load_chunk(const T* base, int64_t ld, int row, int c, int N, int hd) {
    constexpr int EPC = 16 / Tr<T>::ES;
    if (row >= N || c * EPC >= hd) return u32x4{0u, 0u, 0u, 0u};
    return *reinterpret_cast<const u32x4*>(base + (int64_t)row * ld + c * EPC);
}
__device__ __forceinline__ void put_row(char* arr, int pitch, int row, int c, u32x4 v) { *reinterpret_cast<u32x4*>(arr + row * pitch + c * 16) = v; }
// the same chunk into the transposed array: element e of the chunk is channel EPC c + e of token `row`
template <typename T> __device__ __forceinline__ void put_trans(char* arr, int pitch, int row, int c, u32x4 v) {
    if (Tr<T>::ES == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) *reinterpret_cast<uint16_t*>(arr + (8 * c + e) * pitch + row * 2) = (uint16_t)(v[e >> 1] >> (16 * (e & 1)));
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) *reinterpret_cast<uint32_t*>(arr + (4 * c + e) * pitch + row * 4) = v[e];
    }
}

template <typename T, int HD, int NMAX>
__global__ __launch_bounds__(NMAX * 4) void attn_tiny_fwd_kernel(const T* __restrict__ qkv, int64_t ld, T* __restrict__ out, int64_t ldo,
                                                           float* __restrict__ lse, int N, int H, int hd, float scale) {
    typedef TinyCfg<T, HD, NMAX> C;
    typedef Tr<T> X;
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
    const T* qp = qkv + (int64_t)b * N * ld + head * hd;
#pragma unroll
    for (int c = tid; c < NMAX * C::CPR; c += NMAX * 4) {
        const int row = c / C::CPR, c8 = c % C::CPR;
        put_row(Qs, C::PR, row, c8, load_chunk<T>(qp, ld, row, c8, N, hd));
        put_row(Ks, C::PR, row, c8, load_chunk<T>(qp + Cdim, ld, row, c8, N, hd));
        put_trans<T>(Vt, C::PT, row, c8, load_chunk<T>(qp + 2 * Cdim, ld, row, c8, N, hd));
    }
    __syncthreads();
    const float sl = scale * LOG2E;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    {
        u32x4 qf[C::KD];
#pragma unroll
        for (int ks = 0; ks < C::KD; ++ks) qf[ks] = frag(Qs, C::PR, 16 * tq + r, ks, g);
        f32x4 st[C::NQ];
        float m = -INFINITY;
#pragma unroll
        for (int tk = 0; tk < C::NQ; ++tk) {
            f32x4 a = zero;
#pragma unroll
            for (int ks = 0; ks < C::KD; ++ks) a = X::mma(frag(Ks, C::PR, 16 * tk + r, ks, g), qf[ks], a);
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
        for (int tk = 0; tk < C::NQ; ++tk) X::put4(Ps + q * C::PT + (16 * tk + 4 * g) * C::ES, st[tk] * inv);
        if (lse && g == 0 && q < N) lse[bh * N + q] = (m + __builtin_amdgcn_logf(sum)) * LN2;
    }
    __syncthreads();
    T* op = out + (int64_t)b * N * ldo + head * hd;
    {
        u32x4 pf[C::KN];
#pragma unroll
        for (int ks = 0; ks < C::KN; ++ks) pf[ks] = frag_t<C::HALF>(Ps, C::PT, 16 * tq + r, ks, g);
        const int q = 16 * tq + r;
#pragma unroll
        for (int td = 0; td < C::ND; ++td) {
            f32x4 a = zero;
#pragma unroll
            for (int ks = 0; ks < C::KN; ++ks) a = X::mma(frag_t<C::HALF>(Vt, C::PT, 16 * td + r, ks, g), pf[ks], a);
            const int d0 = 16 * td + 4 * g;
            if (q < N && d0 < hd) X::put4(op + (int64_t)q * ldo + d0, a);
        }
    }
}

template <typename T, int HD, int NMAX>
__global__ __launch_bounds__(NMAX * 4) void attn_tiny_bwd_kernel(const T* __restrict__ qkv, int64_t ld, const T* __restrict__ out, int64_t ldo,
                                                           const T* __restrict__ dout, int64_t lddo, const float* __restrict__ lse,
                                                           float* __restrict__ delta, T* __restrict__ dqkv, int64_t lddq, int N, int H, int hd,
                                                           float scale) {
    typedef TinyCfg<T, HD, NMAX> C;
    typedef Tr<T> X;
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
    const T* qp = qkv + (int64_t)b * N * ld + head * hd;
    const T* dop = dout + (int64_t)b * N * lddo + head * hd;
    const T* outp = out + (int64_t)b * N * ldo + head * hd;
#pragma unroll
    for (int c = tid; c < NMAX * C::CPR; c += NMAX * 4) {
        const int row = c / C::CPR, c8 = c % C::CPR;
        const u32x4 q = load_chunk<T>(qp, ld, row, c8, N, hd), k = load_chunk<T>(qp + Cdim, ld, row, c8, N, hd);
        const u32x4 v = load_chunk<T>(qp + 2 * Cdim, ld, row, c8, N, hd), d = load_chunk<T>(dop, lddo, row, c8, N, hd);
        const u32x4 o = load_chunk<T>(outp, ldo, row, c8, N, hd);
        put_row(Qs, C::PR, row, c8, q); put_row(Ks, C::PR, row, c8, k); put_row(Vs, C::PR, row, c8, v); put_row(Ds, C::PR, row, c8, d);
        put_trans<T>(Qt, C::PT, row, c8, q); put_trans<T>(Kt, C::PT, row, c8, k); put_trans<T>(Dt, C::PT, row, c8, d);
        // delta = dO . O: the row's CPR chunks sit in CPR neighbouring lanes
        float part = X::dot(o, d);
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
        u32x4 qf[C::KD], df[C::KD];
#pragma unroll
        for (int ks = 0; ks < C::KD; ++ks) {
            qf[ks] = frag(Qs, C::PR, 16 * tq + r, ks, g);
            df[ks] = frag(Ds, C::PR, 16 * tq + r, ks, g);
        }
        const float lq = lse_s[16 * tq + r], dq = del_s[16 * tq + r];
#pragma unroll
        for (int tk = 0; tk < C::NQ; ++tk) {
            f32x4 s = zero, dp = zero;
#pragma unroll
            for (int ks = 0; ks < C::KD; ++ks) {
                s = X::mma(frag(Ks, C::PR, 16 * tk + r, ks, g), qf[ks], s);
                dp = X::mma(frag(Vs, C::PR, 16 * tk + r, ks, g), df[ks], dp);
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
            X::put1(Pt + (k0 + i) * C::PT + q * C::ES, p[tk][i]);
            X::put1(dSt + (k0 + i) * C::PT + q * C::ES, ds[tk][i]);
        }
        X::put4(dSs + q * C::PT + k0 * C::ES, ds[tk]);
    }
    __syncthreads();
    T* gp = dqkv + (int64_t)b * N * lddq + head * hd;
    // token tile tt = this wave's: dQ^T[d][q] = K^T dS-rows, dK^T[d][key] = Q^T dS^T-rows, dV^T[d][key] = dO^T P^T-rows
    {
        const int tt = tq;
        u32x4 sq[C::KN], sk[C::KN], pk[C::KN];
#pragma unroll
        for (int ks = 0; ks < C::KN; ++ks) {
            sq[ks] = frag_t<C::HALF>(dSs, C::PT, 16 * tt + r, ks, g);
            sk[ks] = frag_t<C::HALF>(dSt, C::PT, 16 * tt + r, ks, g);
            pk[ks] = frag_t<C::HALF>(Pt, C::PT, 16 * tt + r, ks, g);
        }
        const int tok = 16 * tt + r;
#pragma unroll
        for (int td = 0; td < C::ND; ++td) {
            f32x4 aq = zero, ak = zero, av = zero;
#pragma unroll
            for (int ks = 0; ks < C::KN; ++ks) {
                aq = X::mma(frag_t<C::HALF>(Kt, C::PT, 16 * td + r, ks, g), sq[ks], aq);
                ak = X::mma(frag_t<C::HALF>(Qt, C::PT, 16 * td + r, ks, g), sk[ks], ak);
                av = X::mma(frag_t<C::HALF>(Dt, C::PT, 16 * td + r, ks, g), pk[ks], av);
            }
            const int d0 = 16 * td + 4 * g;
            if (tok < N && d0 < hd) {
                T* row = gp + (int64_t)tok * lddq + d0;
                X::put4(row, aq * scale);
                X::put4(row + Cdim, ak * scale);
                X::put4(row + 2 * Cdim, av);
            }
        }
    }
}

template <typename T, int HD, int NMAX>
int launch_fwd(const void* qkv, int64_t ld, void* out, int64_t ldo, float* lse, int B, int N, int H, int hd, float scale, hipStream_t stream) {
    typedef TinyCfg<T, HD, NMAX> C;
    static OncePerDevice once;
    if (once.need()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_tiny_fwd_kernel<T, HD, NMAX>), hipFuncAttributeMaxDynamicSharedMemorySize, C::FWD_LDS);
    hipLaunchKernelGGL((attn_tiny_fwd_kernel<T, HD, NMAX>), dim3((unsigned)((int64_t)B * H)), dim3(NMAX * 4), C::FWD_LDS, stream, reinterpret_cast<const T*>(qkv), ld,
                       reinterpret_cast<T*>(out), ldo, lse, N, H, hd, scale);
    ME_CHECK_LAUNCH("me_attention_fwd(tiny)");
    return ME_OK;
}
template <typename T, int HD, int NMAX>
int launch_bwd(const void* qkv, int64_t ld, const void* out, int64_t ldo, const void* dout, int64_t lddo, const float* lse, float* delta, void* dqkv,
               int64_t lddq, int B, int N, int H, int hd, float scale, hipStream_t stream) {
    typedef TinyCfg<T, HD, NMAX> C;
    static OncePerDevice once;
    if (once.need()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_tiny_bwd_kernel<T, HD, NMAX>), hipFuncAttributeMaxDynamicSharedMemorySize, C::BWD_LDS);
    hipLaunchKernelGGL((attn_tiny_bwd_kernel<T, HD, NMAX>), dim3((unsigned)((int64_t)B * H)), dim3(NMAX * 4), C::BWD_LDS, stream, reinterpret_cast<const T*>(qkv), ld,
                       reinterpret_cast<const T*>(out), ldo, reinterpret_cast<const T*>(dout), lddo, lse, delta, reinterpret_cast<T*>(dqkv), lddq,
                       N, H, hd, scale);
    ME_CHECK_LAUNCH("me_attention_bwd(tiny)");
    return ME_OK;
}

}  // namespace

// no dropout, N <= 64, head_dim <= 64, 16-byte aligned head rows (row strides and head_dim multiples of 8 bf16 / 4 fp32 elements)
bool attn_tiny_ok(int dtype, int64_t ld_qkv, int64_t ld_out, int B, int N, int H, int hd) {
    const int E = dtype == ME_BF16 ? 8 : 4;
    return (dtype == ME_BF16 || dtype == ME_F32) && N >= 1 && N <= 64 && hd >= E && hd <= 64 && hd % E == 0 && ld_qkv % E == 0 && ld_out % E == 0 &&
           (int64_t)B * H < (1ll << 31);
}

int launch_attn_tiny_fwd(int dtype, const void* qkv, int64_t ld, void* out, int64_t ldo, float* lse, int B, int N, int H, int hd, float scale,
                         hipStream_t stream) {
#define TINY_FWD(T_, HD_, NM_) return launch_fwd<T_, HD_, NM_>(qkv, ld, out, ldo, lse, B, N, H, hd, scale, stream)
#define TINY_FWD_T(T_)                      \
    do {                                    \
        if (hd <= 32) {                     \
            if (N <= 16) TINY_FWD(T_, 32, 16); \
            if (N <= 32) TINY_FWD(T_, 32, 32); \
            TINY_FWD(T_, 32, 64);           \
        }                                   \
        if (N <= 16) TINY_FWD(T_, 64, 16);  \
        if (N <= 32) TINY_FWD(T_, 64, 32);  \
        TINY_FWD(T_, 64, 64);               \
    } while (0)
    if (dtype == ME_BF16) TINY_FWD_T(bf16_t);
    TINY_FWD_T(float);
#undef TINY_FWD_T
#undef TINY_FWD
}

int launch_attn_tiny_bwd(int dtype, const void* qkv, int64_t ld, const void* out, int64_t ldo, const void* dout, int64_t lddo, const float* lse,
                         float* delta, void* dqkv, int64_t lddq, int B, int N, int H, int hd, float scale, hipStream_t stream) {
#define TINY_BWD(T_, HD_, NM_) return launch_bwd<T_, HD_, NM_>(qkv, ld, out, ldo, dout, lddo, lse, delta, dqkv, lddq, B, N, H, hd, scale, stream)
#define TINY_BWD_T(T_)                      \
    do {                                    \
        if (hd <= 32) {                     \
            if (N <= 16) TINY_BWD(T_, 32, 16); \
            if (N <= 32) TINY_BWD(T_, 32, 32); \
            TINY_BWD(T_, 32, 64);           \
        }                                   \
        if (N <= 16) TINY_BWD(T_, 64, 16);  \
        if (N <= 32) TINY_BWD(T_, 64, 32);  \
        TINY_BWD(T_, 64, 64);               \
    } while (0)
    if (dtype == ME_BF16) TINY_BWD_T(bf16_t);
    TINY_BWD_T(float);
#undef TINY_BWD_T
#undef TINY_BWD
}
