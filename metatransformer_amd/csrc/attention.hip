// attention.hip -- fused multi-head self-attention core (forward + backward), flash-style.
//
// Replaces PointCloud/openpoints/models/layers/attention.py:28-35: the qkv reshape/permute, (q @ k^T) * scale,
// softmax(-1), attn @ v and the transpose back to [B,N,C].  The [B,H,N,N] score matrix is never written: K/V
// tiles of 64 keys are staged in LDS, the softmax runs online in registers, heads are addressed in place in
// the [tokens, 3C] Linear output (no permute copy) and O is written head-major straight into [tokens, C].
//
// MFMA formulation (32x32 shapes, see common.h): every product is issued "transposed" so that the softmax
// row (one query) lives in ONE lane (plus its partner lane^32):
//   S^T[kv][q] = K Q^T           A = K rows (from LDS), B = Q rows (registers)      -> lane q, regs kv
//   O^T[d][q] += V^T P^T         A = V^T rows d (LDS, transposed while staging), B = P (registers, in place)
// The row max / row sum are 16-register reductions plus one xor-32 shuffle; the running rescale of O^T is a
// per-lane scalar.  The reduction-index permutation the accumulator layout imposes on P is absorbed by reading
// V^T with the same permutation (common.h: any assignment works if A and B agree).
// fp32 runs the same code on the exact-fp32 MFMA (parity mode); bf16 is the performance mode.
#include "common.h"
#include <math.h>
#include <stdlib.h>

// attention_tiny.hip: N <= 64, one wave per (batch, head)   (-DME_TINY_ATTN=0: A/B arm without it, tools/runs/r5_run_tiny.sh)
#ifndef ME_TINY_ATTN
#define ME_TINY_ATTN 1
#endif
bool attn_tiny_ok(int dtype, int64_t ld_qkv, int64_t ld_out, int B, int N, int H, int hd);
int launch_attn_tiny_fwd(int dtype, const void* qkv, int64_t ld, void* out, int64_t ldo, float* lse, int B, int N, int H, int hd, float scale,
                         hipStream_t stream);
int launch_attn_tiny_bwd(int dtype, const void* qkv, int64_t ld, const void* out, int64_t ldo, const void* dout, int64_t lddo, const float* lse,
                         float* delta, void* dqkv, int64_t lddq, int B, int N, int H, int hd, float scale, hipStream_t stream);

namespace {

constexpr int AT_THREADS = 256;
constexpr int KVT = 64;      // keys per LDS tile
constexpr int QPB = 128;     // queries per block (32 per wave)

template <typename T, int HD> struct Cfg {
    static constexpr int E = 16 / sizeof(T);
    static constexpr int CPR = HD / E;                       // 16-byte chunks per row along d
    static constexpr int NKK = HD / (2 * E);                 // chunk pairs along d
    static constexpr int NDB = HD / 32;                      // 32-wide d blocks
    static constexpr int RROW = HD * sizeof(T) + 16;         // row-major tile row stride (bytes), conflict-free pad
    static constexpr int TROW = KVT * sizeof(T) + (sizeof(T) == 2 ? 8 : 16);   // transposed tile row stride
    static constexpr int R_BYTES = KVT * RROW;               // [64][HD] row-major tile
    static constexpr int T_BYTES = HD * TROW;                // [HD][64] transposed tile
    static constexpr int NPC = KVT / (2 * E);                // P chunks per 64-key tile (4 bf16 / 8 fp32)
    static constexpr int R_ITEMS = (KVT * CPR + AT_THREADS - 1) / AT_THREADS;
    static constexpr int T_ITEMS = sizeof(T) == 2 ? ((KVT / 2) * CPR + AT_THREADS - 1) / AT_THREADS
                                                  : (KVT * CPR + AT_THREADS - 1) / AT_THREADS;
    static constexpr int T_REGS = sizeof(T) == 2 ? 2 * T_ITEMS : T_ITEMS;
};

__device__ __forceinline__ u32x4 zero4() { return u32x4{0u, 0u, 0u, 0u}; }

// ---- row-major [64][HD] tile: global -> regs -> LDS
template <typename T, int HD> struct RowStage {
    typedef Cfg<T, HD> C;
    u32x4 v[C::R_ITEMS];
    __device__ __forceinline__ void load(const T* __restrict__ base, int64_t ld, int r0, int nrows, int hd, int tid) {
#pragma unroll
        for (int i = 0; i < C::R_ITEMS; ++i) {
            const int it = tid + AT_THREADS * i;
            const int chunk = it % C::CPR, row = it / C::CPR;
            const bool ok = (row < KVT) && (r0 + row < nrows) && (chunk * C::E < hd);
            v[i] = ok ? *reinterpret_cast<const u32x4*>(base + (int64_t)(r0 + row) * ld + chunk * C::E) : zero4();
        }
    }
    __device__ __forceinline__ void store(char* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < C::R_ITEMS; ++i) {
            const int it = tid + AT_THREADS * i;
            const int chunk = it % C::CPR, row = it / C::CPR;
            if (row < KVT) *reinterpret_cast<u32x4*>(lds + row * C::RROW + chunk * 16) = v[i];
        }
    }
};

// ---- transposed [HD][64] tile (rows = d, 64 keys/queries contiguous): global -> regs -> LDS
template <typename T, int HD> struct TransStage;
template <int HD> struct TransStage<bf16_t, HD> {
    typedef Cfg<bf16_t, HD> C;
    u32x4 v[C::T_REGS];
    __device__ __forceinline__ void load(const bf16_t* __restrict__ base, int64_t ld, int r0, int nrows, int hd, int tid) {
#pragma unroll
        for (int i = 0; i < C::T_ITEMS; ++i) {
            const int it = tid + AT_THREADS * i;
            const int chunk = it % C::CPR, p = it / C::CPR;
            const int row = r0 + 2 * p;
            const bool okc = (p < KVT / 2) && (chunk * 8 < hd);
            v[2 * i] = (okc && row < nrows) ? *reinterpret_cast<const u32x4*>(base + (int64_t)row * ld + chunk * 8) : zero4();
            v[2 * i + 1] = (okc && row + 1 < nrows) ? *reinterpret_cast<const u32x4*>(base + (int64_t)(row + 1) * ld + chunk * 8) : zero4();
        }
    }
    __device__ __forceinline__ void store(char* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < C::T_ITEMS; ++i) {
            const int it = tid + AT_THREADS * i;
            const int chunk = it % C::CPR, p = it / C::CPR;
            if (p < KVT / 2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t w0 = v[2 * i][e >> 1], w1 = v[2 * i + 1][e >> 1];
                    const uint32_t lo = (e & 1) ? (w0 >> 16) : (w0 & 0xffffu);
                    const uint32_t hi = (e & 1) ? (w1 & 0xffff0000u) : (w1 << 16);
                    *reinterpret_cast<uint32_t*>(lds + (chunk * 8 + e) * C::TROW + p * 4) = lo | hi;
                }
            }
        }
    }
};
template <int HD> struct TransStage<float, HD> {
    typedef Cfg<float, HD> C;
    u32x4 v[C::T_REGS];
    __device__ __forceinline__ void load(const float* __restrict__ base, int64_t ld, int r0, int nrows, int hd, int tid) {
#pragma unroll
        for (int i = 0; i < C::T_ITEMS; ++i) {
            const int it = tid + AT_THREADS * i;
            const int chunk = it % C::CPR, row = it / C::CPR;
            const bool ok = (row < KVT) && (r0 + row < nrows) && (chunk * 4 < hd);
            v[i] = ok ? *reinterpret_cast<const u32x4*>(base + (int64_t)(r0 + row) * ld + chunk * 4) : zero4();
        }
    }
    __device__ __forceinline__ void store(char* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < C::T_ITEMS; ++i) {
            const int it = tid + AT_THREADS * i;
            const int chunk = it % C::CPR, row = it / C::CPR;
            if (row < KVT) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    *reinterpret_cast<uint32_t*>(lds + (chunk * 4 + e) * C::TROW + row * 4) = v[i][e];
            }
        }
    }
};

// A-operand chunk from a row-major tile: row, d-chunk index (2*kk + h)
template <typename T, int HD>
__device__ __forceinline__ typename Chunk<T>::type rtile_chunk(const char* lds, int row, int chunk) {
    return *reinterpret_cast<const typename Chunk<T>::type*>(lds + row * Cfg<T, HD>::RROW + chunk * 16);
}
// A-operand chunk from a transposed tile: row d, P-chunk c (0..NPC-1) of the 64-wide tile, half h.
// Element e of the chunk is reduction index (within the 64-tile):
//   bf16: 16c + 4h + (e&3) + 8(e>>2)        fp32: 8c + 4h + e
// which is exactly the index that accumulator register (E*(c % (NPC/2)) + e) of 32-subtile u = c / (NPC/2)
// holds in half h (acc_row), so S/P registers feed the next MFMA without any cross-lane movement.
template <int HD>
__device__ __forceinline__ bf16x8 ttile_chunk(const bf16_t*, const char* lds, int d, int c, int h) {
    const char* base = lds + d * Cfg<bf16_t, HD>::TROW;
    union { u32x2 w[2]; bf16x8 b; } u;
    u.w[0] = *reinterpret_cast<const u32x2*>(base + (16 * c + 4 * h) * 2);
    u.w[1] = *reinterpret_cast<const u32x2*>(base + (16 * c + 8 + 4 * h) * 2);
    return u.b;
}
template <int HD>
__device__ __forceinline__ f32x4 ttile_chunk(const float*, const char* lds, int d, int c, int h) {
    return *reinterpret_cast<const f32x4*>(lds + d * Cfg<float, HD>::TROW + (8 * c + 4 * h) * 4);
}
// bf16: the same operand straight from a ROW-MAJOR [64][HD] tile with the LDS transpose read (ds_read_b64_tr_b16),
// no transposed staging at all.  A 16-lane group reads four reduction rows x 16 columns; lane p of the group supplies
// the address of row +(p>>2), columns +4*(p&3) and receives column +p of the four rows (tools/probe_isa.hip).  Rows are
// picked so that element e = 4r + j of half h is reduction index 16c + 8r + 4h + j == the accumulator-register mapping
// above, so P / dS registers still feed the MFMA unchanged.
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;
template <int HD>
__device__ __forceinline__ bf16x8 tr_chunk(const char* tile, int dbase, int c, int lane) {
    const int g = lane >> 4, p = lane & 15, h = g >> 1;
    const int col = dbase + 16 * (g & 1) + 4 * (p & 3);
    const uint32_t base = (uint32_t)(uintptr_t)tile + col * 2;
    union { bf16x4 q[2]; bf16x8 v; } u;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = 16 * c + 8 * r + 4 * h + (p >> 2);
        u.q[r] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(base + row * Cfg<bf16_t, HD>::RROW));
    }
    return u.v;
}
// "operand with the reduction index along the tile rows": fp32 -> transposed tile + 16-byte reads, bf16 -> row tile + tr reads
template <typename T, int HD> struct TRead;
template <int HD> struct TRead<float, HD> {
    static constexpr bool kNeedsTransposedTile = true;
    static __device__ __forceinline__ f32x4 chunk(const char* ttile, const char*, int db, int c, int lane) {
        return ttile_chunk<HD>((const float*)nullptr, ttile, 32 * db + (lane & 31), c, lane >> 5);
    }
};
template <int HD> struct TRead<bf16_t, HD> {
    static constexpr bool kNeedsTransposedTile = false;
    static __device__ __forceinline__ bf16x8 chunk(const char*, const char* rtile, int db, int c, int lane) {
        return tr_chunk<HD>(rtile, 32 * db, c, lane);
    }
};

// B-operand chunk c from the two 32x32 accumulators of a 64-wide tile
__device__ __forceinline__ bf16x8 pack_chunk(const bf16_t*, const f32x16 (&s)[2], int c) {
    const int u = c >> 1, o = (c & 1) * 8;
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (bf16_t)s[u][o + e];
    return r;
}
__device__ __forceinline__ f32x4 pack_chunk(const float*, const f32x16 (&s)[2], int c) {
    const int u = c >> 2, o = (c & 3) * 4;
    return f32x4{s[u][o], s[u][o + 1], s[u][o + 2], s[u][o + 3]};
}

template <typename T> __device__ __forceinline__ void store_quad(T* p, f32x4 v);
template <> __device__ __forceinline__ void store_quad<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void store_quad<bf16_t>(bf16_t* p, f32x4 v) {
    bf16x4 o;
    o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
    *reinterpret_cast<bf16x4*>(p) = o;
}

// =====================================================================================================
// forward
// =====================================================================================================
template <typename T, int HD>
__global__ __launch_bounds__(AT_THREADS) void attn_fwd_kernel(const T* __restrict__ qkv, int64_t ld,
                                                              T* __restrict__ out, int64_t ldo,
                                                              float* __restrict__ lse, int N, int H, int hd,
                                                              float scale, float p_drop, uint64_t seed) {
    typedef Cfg<T, HD> C;
    typedef typename Chunk<T>::type chunk_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool TT = TRead<T, HD>::kNeedsTransposedTile;
    char* Ks = smem;
    char* Vs = smem + C::R_BYTES;              // fp32: transposed [HD][64]; bf16: row-major [64][HD] (read with tr)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int qbase = blockIdx.x * QPB + wave * 32;
    const int Cdim = H * hd;
    const T* qptr = qkv + (int64_t)b * N * ld + head * hd;
    const T* kptr = qptr + Cdim;
    const T* vptr = qptr + 2 * Cdim;

    chunk_t qf[C::NKK];
    {
        const int qrow = (qbase + l31 < N) ? qbase + l31 : N - 1;
#pragma unroll
        for (int kk = 0; kk < C::NKK; ++kk) {
            const int d = (2 * kk + h) * C::E;
            u32x4 raw = (d < hd) ? *reinterpret_cast<const u32x4*>(qptr + (int64_t)qrow * ld + d) : zero4();
            qf[kk] = *reinterpret_cast<chunk_t*>(&raw);
        }
    }
    f32x16 o[C::NDB];
#pragma unroll
    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const bool active = qbase < N;      // wave-uniform
    const int ntiles = (N + KVT - 1) / KVT;

    RowStage<T, HD> ks, vrs;
    TransStage<T, HD> vts;
    ks.load(kptr, ld, 0, N, hd, tid);
    if (TT) vts.load(vptr, ld, 0, N, hd, tid); else vrs.load(vptr, ld, 0, N, hd, tid);
    for (int j = 0; j < ntiles; ++j) {
        __syncthreads();
        ks.store(Ks, tid);
        if (TT) vts.store(Vs, tid); else vrs.store(Vs, tid);
        __syncthreads();
        if (j + 1 < ntiles) {
            ks.load(kptr, ld, (j + 1) * KVT, N, hd, tid);
            if (TT) vts.load(vptr, ld, (j + 1) * KVT, N, hd, tid); else vrs.load(vptr, ld, (j + 1) * KVT, N, hd, tid);
        }
        if (!active) continue;
        f32x16 s[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[u][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < C::NKK; ++kk)
                s[u] = mma_chunk(rtile_chunk<T, HD>(Ks, 32 * u + l31, 2 * kk + h), qf[kk], s[u]);
        }
        const int kv0 = j * KVT;
        float mt = -INFINITY;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kv = kv0 + 32 * u + acc_row(r, h);
                const float v = (kv < N) ? s[u][r] * scale : -INFINITY;
                s[u][r] = v;
                mt = fmaxf(mt, v);
            }
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);      // finite: every tile holds at least one valid key
        const float alpha = __expf(m_run - m_new);
        float ps = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __expf(s[u][r] - m_new);
                s[u][r] = p;
                ps += p;
            }
        l_run = l_run * alpha + ps;
        m_run = m_new;
        if (p_drop > 0.f) {      // attn_drop (attention.py:33): mask the NORMALISED probabilities -- the row sum stays unmasked
            const float keep = 1.0f / (1.0f - p_drop);
            const uint64_t rowbase = (((uint64_t)b * H + head) * N + (uint64_t)(qbase + l31)) * (uint64_t)N;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    s[u][r] = u01_hash(seed, rowbase + (uint64_t)(kv0 + 32 * u + acc_row(r, h))) >= p_drop ? s[u][r] * keep : 0.f;
        }
#pragma unroll
        for (int db = 0; db < C::NDB; ++db) o[db] *= alpha;
#pragma unroll
        for (int c = 0; c < C::NPC; ++c) {
            const chunk_t pb = pack_chunk((const T*)nullptr, s, c);
#pragma unroll
            for (int db = 0; db < C::NDB; ++db)
                o[db] = mma_chunk(TRead<T, HD>::chunk(Vs, Vs, db, c, lane), pb, o[db]);
        }
    }
    if (!active) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q = qbase + l31;
    if (q < N) {
        T* orow = out + ((int64_t)b * N + q) * ldo + head * hd;
#pragma unroll
        for (int db = 0; db < C::NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = 32 * db + 8 * g + 4 * h;
                if (d < hd)
                    store_quad<T>(orow + d, f32x4{o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv,
                                                  o[db][4 * g + 3] * inv});
            }
        if (lse && h == 0) lse[((int64_t)b * H + head) * N + q] = m_run + __logf(l_tot);
    }
}

// =====================================================================================================
// backward
//   delta[q]   = sum_d dO[q,d] O[q,d]
//   P          = exp(S*scale - lse[q]) ; dP = dO V^T ; dS = P * (dP - delta[q]) * scale
//   dV = P^T dO ; dK = dS^T Q ; dQ = dS K
// Two kernels, no atomics: (1) dK/dV: a wave owns 32 keys, walks query tiles; (2) dQ: a wave owns 32 queries,
// walks key tiles (recomputing S and dP).
// =====================================================================================================
__global__ __launch_bounds__(256) void attn_delta_kernel(const void* __restrict__ o, int64_t ldo,
                                                         const void* __restrict__ dout, int64_t lddo, int dt,
                                                         float* __restrict__ delta, int N, int H, int hd, int64_t rows) {
    // one wave per (token row, head)
    const int lane = threadIdx.x & 63;
    const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= rows * H) return;
    const int64_t row = gw / H;
    const int head = (int)(gw % H);
    float s = 0.f;
    for (int d = lane; d < hd; d += 64)
        s += load1_as_f32(o, dt, row * ldo + head * hd + d) * load1_as_f32(dout, dt, row * lddo + head * hd + d);
    s = wave_sum(s);
    if (lane == 0) {
        const int64_t b = row / N, n = row % N;
        delta[(b * H + head) * N + n] = s;
    }
}

// vector path for bf16 with head_dim a power-of-two multiple of 8 (<= 512): one wave per token row, a lane owns 8
// consecutive channels (16-byte loads of O and dO), the head_dim/8 lanes of a head fold with xor-shuffles.
__global__ __launch_bounds__(256) void attn_delta_vec_kernel(const bf16_t* __restrict__ o, int64_t ldo,
                                                             const bf16_t* __restrict__ dout, int64_t lddo,
                                                             float* __restrict__ delta, int N, int H, int hd, int64_t rows) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int C8 = (H * hd) / 8, lph = hd / 8;      // chunks per row, lanes per head
    const int64_t b = row / N, n = row % N;
    for (int c = lane; c < ((C8 + 63) / 64) * 64; c += 64) {
        float s = 0.f;
        if (c < C8) {
            const u32x4 ro = *reinterpret_cast<const u32x4*>(o + row * ldo + c * 8);
            const u32x4 rd = *reinterpret_cast<const u32x4*>(dout + row * lddo + c * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                s += __uint_as_float(ro[e] << 16) * __uint_as_float(rd[e] << 16) +
                     __uint_as_float(ro[e] & 0xffff0000u) * __uint_as_float(rd[e] & 0xffff0000u);
        }
        for (int off = 1; off < lph; off <<= 1) s += __shfl_xor(s, off, 64);
        if (c < C8 && (c % lph) == 0) delta[(b * H + c / lph) * N + n] = s;
    }
}

// ---- dK / dV
template <typename T, int HD>
__global__ __launch_bounds__(AT_THREADS) void attn_bwd_dkdv_kernel(const T* __restrict__ qkv, int64_t ld,
                                                                   const T* __restrict__ dout, int64_t lddo,
                                                                   const float* __restrict__ lse,
                                                                   const float* __restrict__ delta,
                                                                   T* __restrict__ dqkv, int64_t lddq, int N, int H,
                                                                   int hd, float scale, float p_drop, uint64_t seed) {
    typedef Cfg<T, HD> C;
    typedef typename Chunk<T>::type chunk_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TB = TRead<T, HD>::kNeedsTransposedTile ? C::T_BYTES : 0;   // bf16 reads the row tiles transposed (tr)
    char* Qs = smem;                           // [64 q][HD]
    char* dOs = Qs + C::R_BYTES;               // [64 q][HD]
    char* QTs = dOs + C::R_BYTES;              // [HD][64 q]   (fp32 only)
    char* dOTs = QTs + TB;                     // [HD][64 q]   (fp32 only)
    float* lse_s = reinterpret_cast<float*>(dOTs + TB);           // [64]
    float* del_s = lse_s + KVT;                                   // [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int kvbase = blockIdx.x * QPB + wave * 32;
    const int Cdim = H * hd;
    const T* qptr = qkv + (int64_t)b * N * ld + head * hd;
    const T* kptr = qptr + Cdim;
    const T* vptr = qptr + 2 * Cdim;
    const T* doptr = dout + (int64_t)b * N * lddo + head * hd;
    const float* lse_bh = lse + ((int64_t)b * H + head) * N;
    const float* del_bh = delta + ((int64_t)b * H + head) * N;

    chunk_t kf[C::NKK], vf[C::NKK];
    {
        const int kvrow = (kvbase + l31 < N) ? kvbase + l31 : N - 1;
#pragma unroll
        for (int kk = 0; kk < C::NKK; ++kk) {
            const int d = (2 * kk + h) * C::E;
            u32x4 rk = (d < hd) ? *reinterpret_cast<const u32x4*>(kptr + (int64_t)kvrow * ld + d) : zero4();
            u32x4 rv = (d < hd) ? *reinterpret_cast<const u32x4*>(vptr + (int64_t)kvrow * ld + d) : zero4();
            kf[kk] = *reinterpret_cast<chunk_t*>(&rk);
            vf[kk] = *reinterpret_cast<chunk_t*>(&rv);
        }
    }
    f32x16 dk[C::NDB], dv[C::NDB];
#pragma unroll
    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
    const bool active = kvbase < N;
    const bool kv_ok = kvbase + l31 < N;
    const int ntiles = (N + KVT - 1) / KVT;

    constexpr bool TT = TRead<T, HD>::kNeedsTransposedTile;
    RowStage<T, HD> qs, dos;
    TransStage<T, HD> qts, dots;
    float lse_r = INFINITY, del_r = 0.f;
    auto gload = [&](int q0) {
        qs.load(qptr, ld, q0, N, hd, tid);
        dos.load(doptr, lddo, q0, N, hd, tid);
        if (TT) {
            qts.load(qptr, ld, q0, N, hd, tid);
            dots.load(doptr, lddo, q0, N, hd, tid);
        }
        if (tid < KVT) {
            const int q = q0 + tid;
            lse_r = (q < N) ? lse_bh[q] : INFINITY;          // exp(s - inf) = 0 masks padded queries
            del_r = (q < N) ? del_bh[q] : 0.f;
        }
    };
    gload(0);
    for (int j = 0; j < ntiles; ++j) {
        __syncthreads();
        qs.store(Qs, tid);
        dos.store(dOs, tid);
        if (TT) {
            qts.store(QTs, tid);
            dots.store(dOTs, tid);
        }
        if (tid < KVT) {
            lse_s[tid] = lse_r;
            del_s[tid] = del_r;
        }
        __syncthreads();
        if (j + 1 < ntiles) gload((j + 1) * KVT);            // next tile's global loads fly during this tile's MFMAs
        if (!active) continue;
        f32x16 s[2], dp[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[u][r] = 0.f; dp[u][r] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < C::NKK; ++kk) {
                s[u] = mma_chunk(rtile_chunk<T, HD>(Qs, 32 * u + l31, 2 * kk + h), kf[kk], s[u]);      // S[q][kv]
                dp[u] = mma_chunk(rtile_chunk<T, HD>(dOs, 32 * u + l31, 2 * kk + h), vf[kk], dp[u]);   // dP[q][kv]
            }
        }
        // lane: kv = l31 (fixed), regs: q = 32u + acc_row(r,h)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 L = *reinterpret_cast<const f32x4*>(lse_s + 32 * u + 8 * g + 4 * h);
                const f32x4 D = *reinterpret_cast<const f32x4*>(del_s + 32 * u + 8 * g + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const float p = kv_ok ? __expf(s[u][r] * scale - L[e]) : 0.f;
                    float pm = p, dpm = dp[u][r];
                    if (p_drop > 0.f) {      // same mask as forward: dV sees the dropped P, dS the dropped dP
                        const uint64_t q = (uint64_t)(j * KVT + 32 * u + acc_row(r, h));
                        const uint64_t idx = (((uint64_t)b * H + head) * N + q) * (uint64_t)N + (uint64_t)(kvbase + l31);
                        const float k = u01_hash(seed, idx) >= p_drop ? 1.0f / (1.0f - p_drop) : 0.f;
                        pm *= k; dpm *= k;
                    }
                    s[u][r] = pm;                                  // (dropped) P
                    dp[u][r] = p * (dpm - D[e]) * scale;           // dS
                }
            }
#pragma unroll
        for (int c = 0; c < C::NPC; ++c) {
            const chunk_t pb = pack_chunk((const T*)nullptr, s, c);
            const chunk_t dsb = pack_chunk((const T*)nullptr, dp, c);
#pragma unroll
            for (int db = 0; db < C::NDB; ++db) {
                dv[db] = mma_chunk(TRead<T, HD>::chunk(dOTs, dOs, db, c, lane), pb, dv[db]);   // dV^T[d][kv]
                dk[db] = mma_chunk(TRead<T, HD>::chunk(QTs, Qs, db, c, lane), dsb, dk[db]);  // dK^T[d][kv]
            }
        }
    }
    if (!active || !kv_ok) return;
    T* dkrow = dqkv + ((int64_t)b * N + kvbase + l31) * lddq + Cdim + head * hd;
    T* dvrow = dkrow + Cdim;
#pragma unroll
    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 32 * db + 8 * g + 4 * h;
            if (d < hd) {
                store_quad<T>(dkrow + d, f32x4{dk[db][4 * g], dk[db][4 * g + 1], dk[db][4 * g + 2], dk[db][4 * g + 3]});
                store_quad<T>(dvrow + d, f32x4{dv[db][4 * g], dv[db][4 * g + 1], dv[db][4 * g + 2], dv[db][4 * g + 3]});
            }
        }
}

// ---- dQ
template <typename T, int HD>
__global__ __launch_bounds__(AT_THREADS) void attn_bwd_dq_kernel(const T* __restrict__ qkv, int64_t ld,
                                                                 const T* __restrict__ dout, int64_t lddo,
                                                                 const float* __restrict__ lse,
                                                                 const float* __restrict__ delta,
                                                                 T* __restrict__ dqkv, int64_t lddq, int N, int H,
                                                                 int hd, float scale, float p_drop, uint64_t seed) {
    typedef Cfg<T, HD> C;
    typedef typename Chunk<T>::type chunk_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;                    // [64 kv][HD]
    char* Vs = Ks + C::R_BYTES;         // [64 kv][HD]
    char* KTs = Vs + C::R_BYTES;        // [HD][64 kv]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int qbase = blockIdx.x * QPB + wave * 32;
    const int Cdim = H * hd;
    const T* qptr = qkv + (int64_t)b * N * ld + head * hd;
    const T* kptr = qptr + Cdim;
    const T* vptr = qptr + 2 * Cdim;
    const T* doptr = dout + (int64_t)b * N * lddo + head * hd;

    const bool q_ok = qbase + l31 < N;
    const int qrow = q_ok ? qbase + l31 : N - 1;
    chunk_t qf[C::NKK], dof[C::NKK];
#pragma unroll
    for (int kk = 0; kk < C::NKK; ++kk) {
        const int d = (2 * kk + h) * C::E;
        u32x4 rq = (d < hd) ? *reinterpret_cast<const u32x4*>(qptr + (int64_t)qrow * ld + d) : zero4();
        u32x4 rd = (d < hd) ? *reinterpret_cast<const u32x4*>(doptr + (int64_t)qrow * lddo + d) : zero4();
        qf[kk] = *reinterpret_cast<chunk_t*>(&rq);
        dof[kk] = *reinterpret_cast<chunk_t*>(&rd);
    }
    const float lse_q = lse[((int64_t)b * H + head) * N + qrow];
    const float del_q = delta[((int64_t)b * H + head) * N + qrow];
    f32x16 dq[C::NDB];
#pragma unroll
    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
    const bool active = qbase < N;
    const int ntiles = (N + KVT - 1) / KVT;

    constexpr bool TT = TRead<T, HD>::kNeedsTransposedTile;
    RowStage<T, HD> ks, vs;
    TransStage<T, HD> kts;
    auto gload = [&](int kv0) {
        ks.load(kptr, ld, kv0, N, hd, tid);
        vs.load(vptr, ld, kv0, N, hd, tid);
        if (TT) kts.load(kptr, ld, kv0, N, hd, tid);
    };
    gload(0);
    for (int j = 0; j < ntiles; ++j) {
        const int kv0 = j * KVT;
        __syncthreads();
        ks.store(Ks, tid);
        vs.store(Vs, tid);
        if (TT) kts.store(KTs, tid);
        __syncthreads();
        if (j + 1 < ntiles) gload((j + 1) * KVT);
        if (!active) continue;
        f32x16 s[2], dp[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[u][r] = 0.f; dp[u][r] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < C::NKK; ++kk) {
                s[u] = mma_chunk(rtile_chunk<T, HD>(Ks, 32 * u + l31, 2 * kk + h), qf[kk], s[u]);      // S^T[kv][q]
                dp[u] = mma_chunk(rtile_chunk<T, HD>(Vs, 32 * u + l31, 2 * kk + h), dof[kk], dp[u]);   // dP^T[kv][q]
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kv = kv0 + 32 * u + acc_row(r, h);
                const float p = (kv < N) ? __expf(s[u][r] * scale - lse_q) : 0.f;
                float dpm = dp[u][r];
                if (p_drop > 0.f) {
                    const uint64_t idx = (((uint64_t)b * H + head) * N + (uint64_t)(qbase + l31)) * (uint64_t)N + (uint64_t)kv;
                    dpm = u01_hash(seed, idx) >= p_drop ? dpm * (1.0f / (1.0f - p_drop)) : 0.f;
                }
                dp[u][r] = p * (dpm - del_q) * scale;          // dS^T
            }
#pragma unroll
        for (int c = 0; c < C::NPC; ++c) {
            const chunk_t dsb = pack_chunk((const T*)nullptr, dp, c);
#pragma unroll
            for (int db = 0; db < C::NDB; ++db)
                dq[db] = mma_chunk(TRead<T, HD>::chunk(KTs, Ks, db, c, lane), dsb, dq[db]);   // dQ^T[d][q]
        }
    }
    if (!active || !q_ok) return;
    T* dqrow = dqkv + ((int64_t)b * N + qbase + l31) * lddq + head * hd;
#pragma unroll
    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 32 * db + 8 * g + 4 * h;
            if (d < hd)
                store_quad<T>(dqrow + d, f32x4{dq[db][4 * g], dq[db][4 * g + 1], dq[db][4 * g + 2], dq[db][4 * g + 3]});
        }
}


// =====================================================================================================
// small-sequence kernels (bf16, head_dim <= 64, N <= 256): ONE workgroup of 8 waves per (batch, head), the whole
// sequence resident in LDS.  At the encoder's N = 197 the tiled kernels above are bound by staging, not math: two
// query blocks per head re-fetch K/V through different XCD L2s, every 64-key tile costs two barriers, and the
// 128-query / 64-key tiling pads 197 to 256 both ways.  Here each array is fetched exactly once with all loads in
// flight together, rows are padded only to a multiple of 32, and the per-wave loops over key (query) sub-tiles run
// without any barrier.  The backward is ONE kernel: phase A (a wave owns 32 queries) computes delta = rowsum(dO*O)
// and dQ, phase B (a wave owns 32 keys) computes dK and dV from the same resident tiles -- no delta kernel, no second
// pass over HBM.  Both phases recompute S and dP (7 MFMA products instead of the minimal 5); results are
// deterministic (no atomics).
// =====================================================================================================
// developer instrumentation (tools/attn_trace.hip defines ME_ATTN_TRACE and includes this file): per-workgroup
// s_memtime stamps at the phase boundaries; compiled out of libmetaenc.so
#ifdef ME_ATTN_TRACE
__device__ long long g_trace[1 << 19];      // [item slot < 4096][wave < 16][stamp < 8]
#define TRACE_STAMP(slot, k) do { if ((threadIdx.x & 63) == 0 && (slot) < (1 << 12)) g_trace[((slot) * 16 + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define TRACE_STAMP(slot, k) do { } while (0)
#endif

constexpr int SM_THREADS = 512;
constexpr int SM_MAXN = 256;
constexpr int SM_MINN = 64;      // at or below one 64-key tile the tiled kernels (more workgroups per CU) win
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

template <int HD, int NARR, int MAXN = SM_MAXN, int NTHR = SM_THREADS> struct SmallStage {
    static constexpr int CPR = HD / 8;
    static constexpr int ITEMS = (MAXN * CPR) / NTHR;
    u32x4 v[NARR][ITEMS];
    // branch-free: out-of-range rows / chunks read a clamped (valid) address; they are zeroed when the values are WRITTEN to
    // LDS (store), not here -- a select on a loaded value right behind the load makes the wave wait for the data, which
    // is exactly what a prefetch must not do (measured: ~4 k of 15.7 k clocks per item in the persistent forward)
    __device__ __forceinline__ void load(const bf16_t* const (&base)[NARR], const int64_t (&ld)[NARR], int N, int hd, int tid) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int it = tid + NTHR * i;
            const int chunk = it % CPR, row = it / CPR;
            const bool ok = (row < N) && (chunk * 8 < hd);
            const int rc = ok ? row : 0, cc = ok ? chunk : 0;
#pragma unroll
            for (int a = 0; a < NARR; ++a) v[a][i] = *reinterpret_cast<const u32x4*>(base[a] + (int64_t)rc * ld[a] + cc * 8);
        }
    }
    // N / hd: the bounds the matching load() was given (rows >= N and chunks past hd are written as zeros)
    __device__ __forceinline__ void store(char* const (&lds)[NARR], int NR, int tid, int N, int hd) const {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int it = tid + NTHR * i;
            const int chunk = it % CPR, row = it / CPR;
            const bool ok = (row < N) && (chunk * 8 < hd);
            if (row < NR) {
#pragma unroll
                for (int a = 0; a < NARR; ++a)
                    *reinterpret_cast<u32x4*>(lds[a] + row * Cfg<bf16_t, HD>::RROW + chunk * 16) = ok ? v[a][i] : zero4();
            }
        }
    }
};

// Write one wave's 32 x HD result (accumulators in the transposed layout: lane = row l31, registers = d) as bf16 rows.
// Straight from registers this is a row-per-lane scatter of 8-byte pieces (64 rows touched per store instruction), which
// is store-issue bound; instead the wave transposes through a private [32][HD] LDS scratch and stores whole 128-byte
// rows: 16 bytes per lane, 64/CPR rows per instruction.
template <int HD>
__device__ __forceinline__ void store_rows_via_lds(char* scr, const f32x16 (&acc)[Cfg<bf16_t, HD>::NDB], float mul,
                                                   bf16_t* __restrict__ grow0, int64_t ldg, int rows_valid, int hd,
                                                   int lane) {
    typedef Cfg<bf16_t, HD> C;
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(acc[db][4 * g + e] * mul);
            *reinterpret_cast<bf16x4*>(scr + l31 * C::RROW + (32 * db + 8 * g + 4 * h) * 2) = o;
        }
    __builtin_amdgcn_wave_barrier();
    constexpr int RPI = 64 / C::CPR;      // rows per instruction
    const int chunk = lane % C::CPR, r0 = lane / C::CPR;
#pragma unroll
    for (int k = 0; k < 32 / RPI; ++k) {
        const int r = r0 + RPI * k;
        const u32x4 v = *reinterpret_cast<const u32x4*>(scr + r * C::RROW + chunk * 16);
        if (r < rows_valid && chunk * 8 < hd) *reinterpret_cast<u32x4*>(grow0 + (int64_t)r * ldg + chunk * 8) = v;
    }
    __builtin_amdgcn_wave_barrier();
}

// one step of the forward over NU 32-key sub-tiles starting at key kv0 (Kt / Vt point at that row of the resident tiles)
template <int HD, int NU, bool MASK>
__device__ __forceinline__ void fwd_small_step(const char* Kt, const char* Vt, int kv0, int N,
                                               const bf16x8 (&qf)[Cfg<bf16_t, HD>::NKK], float sl, float& m_run,
                                               float& l_run, f32x16 (&o)[Cfg<bf16_t, HD>::NDB], int lane) {
    typedef Cfg<bf16_t, HD> C;
    const int l31 = lane & 31, h = lane >> 5;
    f32x16 s[2];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[u][r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < C::NKK; ++kk)
            s[u] = mma_chunk(rtile_chunk<bf16_t, HD>(Kt, 32 * u + l31, 2 * kk + h), qf[kk], s[u]);
    }
    float mt = -INFINITY;
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (MASK) s[u][r] = (kv0 + 32 * u + acc_row(r, h) < N) ? s[u][r] : -INFINITY;
            mt = fmaxf(mt, s[u][r]);
        }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64)) * sl;          // log2 domain; every step holds at least one valid key
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float ps = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(s[u][r] * sl - m_new);
            s[u][r] = p;
            ps += p;
        }
    l_run = l_run * alpha + ps;
    m_run = m_new;
#pragma unroll
    for (int db = 0; db < C::NDB; ++db) o[db] *= alpha;
#pragma unroll
    for (int c = 0; c < 2 * NU; ++c) {
        const bf16x8 pb = pack_chunk((const bf16_t*)nullptr, s, c);
#pragma unroll
        for (int db = 0; db < C::NDB; ++db) o[db] = mma_chunk(tr_chunk<HD>(Vt, 32 * db, c, lane), pb, o[db]);
    }
}

// Persistent: a workgroup walks (batch, head) items with stride gridDim.x.  While item i is computed, the K/V rows and
// Q fragments of item i+1 are already in flight into registers; they are parked in LDS once every wave is done with
// item i -- HBM latency and the bursty all-CUs-load-at-once phase hide behind the math.
template <int HD>
__global__ __launch_bounds__(SM_THREADS) void attn_fwd_small_kernel(const bf16_t* __restrict__ qkv, int64_t ld,
                                                                    bf16_t* __restrict__ out, int64_t ldo,
                                                                    float* __restrict__ lse, int N, int H, int hd,
                                                                    float scale, int items) {
    typedef Cfg<bf16_t, HD> C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NS = (N + 31) >> 5, NR = NS * 32;
    const int tile_bytes = NR * C::RROW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    char* Ks = smem;
    char* Vs = smem + tile_bytes;
    char* scr = smem + 2 * tile_bytes + wave * 32 * C::RROW;      // per-wave output transposition scratch
    const int Cdim = H * hd;
    const int q = 32 * wave + l31;
    const int qrow = (q < N) ? q : N - 1;
    const bool active = 32 * wave < N;      // wave-uniform
    const float sl = scale * LOG2E;
    const int64_t ldv[2] = {ld, ld};

    SmallStage<HD, 2> st;
    bf16x8 qf[C::NKK], qn[C::NKK];
    auto fresh_tid = [&]() { int t = tid; asm volatile("" : "+v"(t)); return t; };
    auto issue = [&](int it, bf16x8 (&qdst)[C::NKK]) {
        const bf16_t* qptr = qkv + (int64_t)(it / H) * N * ld + (it % H) * hd;
        const bf16_t* const bases[2] = {qptr + Cdim, qptr + 2 * Cdim};
        st.load(bases, ldv, N, hd, fresh_tid());
#pragma unroll
        for (int kk = 0; kk < C::NKK; ++kk) {
            const int d = (2 * kk + h) * 8;
            const bool ok = d < hd;
            const u32x4 raw = *reinterpret_cast<const u32x4*>(qptr + (int64_t)qrow * ld + (ok ? d : 0));
            qdst[kk] = *reinterpret_cast<const bf16x8*>(&raw);      // (chunks past hd are zeroed by q_take, when the value is first needed)
        }
    };
    auto q_take = [&](const bf16x8 (&src)[C::NKK]) {
#pragma unroll
        for (int kk = 0; kk < C::NKK; ++kk) {
            const bool ok = (2 * kk + h) * 8 < hd;
            u32x4 raw = *reinterpret_cast<const u32x4*>(&src[kk]);
            raw = ok ? raw : zero4();
            qf[kk] = *reinterpret_cast<bf16x8*>(&raw);
        }
    };
    auto park = [&]() {
        char* const tiles[2] = {Ks, Vs};
        st.store(tiles, NR, fresh_tid(), N, hd);
    };
    int it = blockIdx.x;
    issue(it, qn);
    park();
    q_take(qn);
    __syncthreads();
    for (; it < items; it += gridDim.x) {
        const int nxt = (it + (int)gridDim.x < items) ? it + (int)gridDim.x : it;
        TRACE_STAMP(it, 0);
        issue(nxt, qn);
        if (active) {
            f32x16 o[C::NDB];
#pragma unroll
            for (int db = 0; db < C::NDB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
            float m_run = -INFINITY, l_run = 0.f;
            int t = 0;
            for (; 32 * (t + 2) <= N; t += 2)      // full 64-key steps
                fwd_small_step<HD, 2, false>(Ks + 32 * t * C::RROW, Vs + 32 * t * C::RROW, 32 * t, N, qf, sl, m_run, l_run, o, lane);
            if (t + 2 <= NS) {
                fwd_small_step<HD, 2, true>(Ks + 32 * t * C::RROW, Vs + 32 * t * C::RROW, 32 * t, N, qf, sl, m_run, l_run, o, lane);
            } else if (t < NS) {
                fwd_small_step<HD, 1, true>(Ks + 32 * t * C::RROW, Vs + 32 * t * C::RROW, 32 * t, N, qf, sl, m_run, l_run, o, lane);
            }
            TRACE_STAMP(it, 1);
            const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
            const float inv = 1.0f / l_tot;
            const int b = it / H, head = it % H;
            store_rows_via_lds<HD>(scr, o, inv, out + ((int64_t)b * N + 32 * wave) * ldo + head * hd, ldo, N - 32 * wave, hd, lane);
            if (lse && h == 0 && q < N) lse[((int64_t)b * H + head) * N + q] = (m_run + __builtin_amdgcn_logf(l_tot)) * LN2;
        }
        TRACE_STAMP(it, 2);
        __syncthreads();      // every wave is done with the K/V tiles of this item
        TRACE_STAMP(it, 3);
        park();
        TRACE_STAMP(it, 4);
        q_take(qn);
        __syncthreads();
        TRACE_STAMP(it, 5);
    }
}

// -----------------------------------------------------------------------------------------------------
// Resident-sequence kernels, ring form (SM_MINN < N <= RS_MAXN = 224: every Base / Large image, window and point-cloud shape).
// What per-wave time stamps of the kernels above showed (tools/attn_trace, N = 197, forward 15.3 k clocks per item):
// ~4 k clocks in which every wave sits in the ISSUE of its prefetch loads (the CU's vector-memory path takes ~32 B/clk: a
// whole item's 100 KB asked for at once backs the queue up and an in-order wave cannot get past a load it cannot issue),
// ~1 k parking the prefetched rows in LDS between two barriers, a step loop hipcc schedules as read -> wait -> MFMA one
// fragment at a time, and two 250-register waves per SIMD that cannot keep the VALU fed (tools/probe_valu).  The ring form:
//   * the NEXT item's arrays go global -> LDS by DMA (buffer_load ... lds) from dedicated LOADER waves, paced, into the half
//     of LDS the current phase does not read: nothing queues in front of the compute waves, no staging registers, no park
//     phase, one barrier per phase.  Rows are unpadded; the 16-byte chunks are XOR-swizzled on the SOURCE side (a DMA lane
//     chooses which global chunk lands in its fixed LDS slot) so that both the 16-byte row reads and the transposing
//     ds_read_b64_tr_b16 reads stay bank-conflict free.  Rows >= N and chunks >= hd arrive as zeros through the buffer
//     descriptor's bounds check.
//   * 16 waves of 16 rows (16 x 16 x 32 MFMA), <= 128 registers, four waves per SIMD.
//   * forward: two-pass softmax -- N <= 224 keeps the whole S^T row block in registers, so there is no running maximum, no
//     rescale of O and no per-step shuffle; MFMA passes software-pipelined by hand (sched_barrier).
typedef __attribute__((address_space(3))) void lds_dma_t;
constexpr int RS_MAXN = 224;
template <int HD> struct RCfg {
    static constexpr int RB = HD * 2;                          // bytes per (unpadded) row
    static constexpr int CPR = HD / 8;                         // 16-byte chunks per row
    static constexpr int RPI = 1024 / RB;                      // rows one DMA instruction fills (64 lanes x 16 B)
    static constexpr int SCR = 32 * Cfg<bf16_t, HD>::RROW;     // per-wave output transposition scratch (padded rows)
};
// -----------------------------------------------------------------------------------------------------
// Ring form with 16-query waves.  tools/probe_valu: ONE wave issues a VALU instruction every ~5 clocks whatever it is, but a SIMD
// retires one per ~1.9 (add / fma), ~2.8 (max3, cvt_pk, packed) or ~5.3 (exp) clocks once three or four waves feed it -- the
// softmax arithmetic of this kernel is bound by how many waves issue it, not by the VALU itself, and two 250-register waves per
// SIMD leave more than half of it idle.  So: 16 waves per workgroup (four per SIMD, <= 128 registers each), a wave owns 16
// queries and uses the 16 x 16 x 32 MFMA:
//   S^T tile (16 keys x 16 queries) = K rows (A: lane = key l & 15, d = 8 (l >> 4) ..) x Q^T (B: lane = query l & 15, same d)
//       -> lane (query l & 15, g = l >> 4) holds keys 16 t + 4 g + j, j = 0..3;
//   O^T tile (16 d x 16 queries) += V^T (A, transposing read) x P^T (B): a lane's 8 reduction slots of a 32-key step are
//       exactly the 2 x 4 accumulator registers it already holds (keys 32 kk + 4 g + j and 32 kk + 16 + 4 g + j), and the
//       transposing read fetches V^T in the same order -- no cross-lane movement between the two products.
// Waves 14 and 15 are the loaders (paced LDS-DMA of the next item's K and V into the other ring half), other waves >= ceil(N / 16)
// only keep the barrier.
constexpr int R16_THREADS = 1024;
#ifndef ME_R16_SLEEP
#define ME_R16_SLEEP 1      // loader pacing: s_sleep units (64 clocks) behind every DMA piece
#endif
// slot of chunk c in row r: c ^ r16_swz(r).  HD = 64: the b128 operand read has lanes {0-3, 12-15} on rows r, chunk c and lanes
// {20-27} on rows 4-11, chunk c + 1 in one LDS cycle, the transposing read 8 consecutive rows x one aligned chunk pair: row bits
// 1..2 -> slot bits 1..2 separates both.  HD = 32 (four rows per 256-byte bank row): per 4-row block the values 0, 2, 3, 1.
template <int CPR> __device__ __forceinline__ int r16_swz(int r) {
    return CPR == 8 ? (((r >> 1) & 3) << 1) : ((0x78 >> (2 * ((r >> 2) & 3))) & 3);
}
__device__ __forceinline__ f32x4 mma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

template <int HD, int NS>
__global__ __launch_bounds__(R16_THREADS) void attn_fwd_ring16_kernel(const bf16_t* __restrict__ qkv, int64_t ld,
                                                                      bf16_t* __restrict__ out, int64_t ldo,
                                                                      float* __restrict__ lse, int N, int H, int hd,
                                                                      float scale, int items) {
    typedef Cfg<bf16_t, HD> C;
    typedef RCfg<HD> R;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NR = NS * 32;
    constexpr int arr_bytes = NR * R::RB;             // one array (K or V) of one item
    constexpr int NRI = NR / R::RPI;                  // DMA instructions per array
    constexpr int NT = 2 * NS;                        // 16-key tiles
    constexpr int NKS = HD / 32;                      // 32-wide d steps of QK^T
    constexpr int NDT = HD / 16;                      // 16-wide d tiles of O^T
    constexpr int SCR = 16 * C::RROW;                 // per-wave output transposition scratch
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int Cdim = H * hd;
    const int G = (int)gridDim.x;
    const bool active = 16 * wave < N;                // wave-uniform

    if (!active) {
        if (wave < 14) {                              // spare waves: the barrier count only
            __syncthreads();
            for (int it = blockIdx.x; it < items; it += G) __syncthreads();
            return;
        }
        // ---- the two loader waves (14: K, 15: V): lane l fills LDS bytes [16 l, 16 l + 16) of an instruction's 1 KiB = slot `pos` of row `rg` of the
        // row group, with the chunk that belongs there; rows >= N / chunks >= hd are out of the descriptor's range -> zeros
        const int rg = (lane * 16) / R::RB, pos = ((lane * 16) % R::RB) / 16;
        const int csrc = pos ^ r16_swz<R::CPR>(rg);
        const int dma_voff = (csrc * 8 < hd) ? rg * (int)ld * 2 + csrc * 16 : 0x7f000000;
        const int dma_gstep = R::RPI * (int)ld * 2;
        const int rec_bytes = (int)(((int64_t)(N - 1) * ld + hd) * 2);
        // PACED in the steady state (one piece per ~150 clocks and loader): asked for all at once, an item's 56 KiB backs up the
        // CU's vector-memory path and the compute waves' own loads and stores queue behind it
        const int which = wave - 14;                  // 0: K, 1: V
        auto fill = [&](int it, char* half, bool paced) {
            const bf16_t* base = qkv + (int64_t)(it / H) * N * ld + (it % H) * hd + (1 + which) * Cdim;
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base), 0, rec_bytes, 0x00020000);
#pragma unroll
            for (int i = 0; i < NRI; ++i) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_dma_t*)(half + which * arr_bytes + i * 1024), 16, dma_voff + i * dma_gstep, 0, 0, 0);
                if (paced && ME_R16_SLEEP) __builtin_amdgcn_s_sleep(ME_R16_SLEEP);
            }
        };
        int it = blockIdx.x, cur = 0;
        fill(it, smem, false);
        __syncthreads();
        for (; it < items; it += G) {
            if (it + G < items) fill(it + G, smem + (cur ^ 1) * 2 * arr_bytes, true);
            __syncthreads();                          // (vmcnt(0) first: the fill has landed)
            cur ^= 1;
        }
        return;
    }

    // ---- compute waves
    char* scr = smem + 4 * arr_bytes + wave * SCR;
    const int q = 16 * wave + l15;
    const int qrow = (q < N) ? q : N - 1;
    const float sl = scale * LOG2E;
    int koff[NKS];                                    // QK^T operand: row 16 t + l15, chunk 4 ks + g
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) koff[ks] = l15 * R::RB + 16 * ((4 * ks + g) ^ r16_swz<R::CPR>(l15));
    int voff[NDT];                                    // PV operand: rows 32 kk + 16 r + 4 g + (l15 >> 2), 8 bytes at d = 16 dt + 4 (l15 & 3)
    {
        const int rr = 4 * g + (l15 >> 2);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
            voff[dt] = rr * R::RB + 16 * ((2 * dt + ((l15 & 3) >> 1)) ^ r16_swz<R::CPR>(rr)) + 8 * (l15 & 1);
    }

    bf16x8 qf[NKS], qn[NKS];
    auto q_issue = [&](int it) {
        const bf16_t* qptr = qkv + (int64_t)(it / H) * N * ld + (it % H) * hd;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d = 32 * ks + 8 * g;
            const u32x4 raw = *reinterpret_cast<const u32x4*>(qptr + (int64_t)qrow * ld + (d < hd ? d : 0));
            qn[ks] = *reinterpret_cast<const bf16x8*>(&raw);
        }
    };
    auto q_take = [&]() {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const bool ok = 32 * ks + 8 * g < hd;
            u32x4 raw = *reinterpret_cast<const u32x4*>(&qn[ks]);
            raw = ok ? raw : zero4();
            qf[ks] = *reinterpret_cast<bf16x8*>(&raw);
        }
    };

    f32x4 o[NDT];
    float m2 = 0.f, l_tot = 1.f;
    int it_out = -1;
    auto flush = [&]() {
        const float inv = __builtin_amdgcn_rcpf(l_tot);      // (1 ulp; the result is rounded to bf16)
        const int b = it_out / H, head = it_out % H;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            bf16x4 v4;
#pragma unroll
            for (int e = 0; e < 4; ++e) v4[e] = (bf16_t)(o[dt][e] * inv);
            *reinterpret_cast<bf16x4*>(scr + l15 * C::RROW + (16 * dt + 4 * g) * 2) = v4;
        }
        __builtin_amdgcn_wave_barrier();
        bf16_t* grow0 = out + ((int64_t)b * N + 16 * wave) * ldo + head * hd;
        constexpr int RPI = 64 / C::CPR;              // rows per store instruction
        const int chunk = lane % C::CPR, r0 = lane / C::CPR;
#pragma unroll
        for (int k = 0; k < 16 / RPI; ++k) {
            const int r = r0 + RPI * k;
            const u32x4 v = *reinterpret_cast<const u32x4*>(scr + r * C::RROW + chunk * 16);
            if (16 * wave + r < N && chunk * 8 < hd) *reinterpret_cast<u32x4*>(grow0 + (int64_t)r * ldo + chunk * 8) = v;
        }
        __builtin_amdgcn_wave_barrier();
        if (lse && g == 0 && q < N) lse[((int64_t)b * H + head) * N + q] = (m2 + __builtin_amdgcn_logf(l_tot)) * LN2;
    };

    int it = blockIdx.x;
    q_issue(it);
    int cur = 0;
    __syncthreads();
    for (; it < items; it += G) {
        const int nxt = (it + G < items) ? it + G : it;
        TRACE_STAMP(it, 0);
        q_take();
        const char* Kb = smem + cur * 2 * arr_bytes;
        const char* Vb = Kb + arr_bytes;
        // ---- pass 1: S^T, two 16-key tiles at a time; the operand chunks of the next pair are in flight while this pair's
        // MFMAs run (software-pipelined by hand: left alone hipcc emits read -> wait -> MFMA one fragment at a time)
        f32x4 s[NT];
        const f32x4 zero4f = {0.f, 0.f, 0.f, 0.f};
        bf16x8 ka[2][NKS], kb[2][NKS];
        auto kread = [&](int t, bf16x8 (&dst)[2][NKS]) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) dst[tt][ks] = *reinterpret_cast<const bf16x8*>(Kb + 16 * (t + tt) * R::RB + koff[ks]);
        };
        auto kmma = [&](int t, const bf16x8 (&src)[2][NKS]) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) s[t + tt] = mma16(src[tt][0], qf[0], zero4f);
#pragma unroll
            for (int ks = 1; ks < NKS; ++ks)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) s[t + tt] = mma16(src[tt][ks], qf[ks], s[t + tt]);
        };
        kread(0, ka);
#pragma unroll
        for (int t = 0; t < NT; t += 4) {
            if (t + 2 < NT) kread(t + 2, kb);
            __builtin_amdgcn_sched_barrier(0);
            kmma(t, ka);
            __builtin_amdgcn_sched_barrier(0);
            if (t == 0) { TRACE_STAMP(it, 1); }
            if (t + 2 < NT) {
                if (t + 4 < NT) kread(t + 4, ka);
                __builtin_amdgcn_sched_barrier(0);
                kmma(t + 2, kb);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        TRACE_STAMP(it, 3);
        q_issue(nxt);                                 // (here, not at the head of the item: the memory path is quiet now)
        // ---- softmax over the whole row (registers only); keys >= N sit in the last two tiles
        if (N & 31) {
            int nkeys = N - 4 * g;                    // (opaque per item: hipcc would hoist the lane masks into SGPRs)
            asm volatile("" : "+v"(nkeys));
#pragma unroll
            for (int t = NT - 2; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) s[t][e] = (16 * t + e < nkeys) ? s[t][e] : -INFINITY;
        }
        float mt = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) mt = fmaxf(mt, s[t][e]);
        mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        m2 = mt * sl;
        float ps = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pe = __builtin_amdgcn_exp2f(s[t][e] * sl - m2);
                s[t][e] = pe;
                ps += pe;
            }
        ps += __shfl_xor(ps, 16, 64);
        l_tot = ps + __shfl_xor(ps, 32, 64);
        TRACE_STAMP(it, 4);
        // ---- pass 2: O^T += V^T P^T, 32 keys at a time, the d tiles in two halves: the V^T operand of the next half is in flight
        // while this half's MFMAs run (half-steps keep the operand buffers at 2 x NDT / 2 fragments: the kernel has to fit 128 registers)
        constexpr int NH = NDT / 2;
        bf16x8 va[NH], vb[NH];
        auto vread = [&](int kk, int half, bf16x8 (&dst)[NH]) {
#pragma unroll
            for (int i = 0; i < NH; ++i) {
                union { bf16x4 q4[2]; bf16x8 v; } a;
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    a.q4[r] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (lds_bf16x4_t*)((uint32_t)(uintptr_t)Vb + (32 * kk + 16 * r) * R::RB + voff[NH * half + i]));
                dst[i] = a.v;
            }
        };
        vread(0, 0, va);
#pragma unroll
        for (int kk = 0; kk < NS; ++kk) {
            vread(kk, 1, vb);
            bf16x8 pb;
#pragma unroll
            for (int e = 0; e < 4; ++e) { pb[e] = (bf16_t)s[2 * kk][e]; pb[4 + e] = (bf16_t)s[2 * kk + 1][e]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NH; ++i) o[i] = mma16(va[i], pb, kk == 0 ? zero4f : o[i]);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 1 < NS) vread(kk + 1, 0, va);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NH; ++i) o[NH + i] = mma16(vb[i], pb, kk == 0 ? zero4f : o[NH + i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        it_out = it;
        TRACE_STAMP(it, 5);
        flush();
        TRACE_STAMP(it, 2);
        __syncthreads();      // the loader's fill of the other half has landed and every wave is done with `cur`
        TRACE_STAMP(it, 6);
        cur ^= 1;
    }
}

// phase A step (wave owns 32 queries): NU 32-key sub-tiles starting at kv0
template <int HD, int NU, bool MASK>
__device__ __forceinline__ void bwd_small_q_step(const char* Kt, const char* Vt, int kv0, int N,
                                                 const bf16x8 (&qf)[Cfg<bf16_t, HD>::NKK],
                                                 const bf16x8 (&dof)[Cfg<bf16_t, HD>::NKK], float sl, float lse2,
                                                 float del, f32x16 (&dq)[Cfg<bf16_t, HD>::NDB], int lane) {
    typedef Cfg<bf16_t, HD> C;
    const int l31 = lane & 31, h = lane >> 5;
    f32x16 s[2], dp[2];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[u][r] = 0.f; dp[u][r] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < C::NKK; ++kk) {
            s[u] = mma_chunk(rtile_chunk<bf16_t, HD>(Kt, 32 * u + l31, 2 * kk + h), qf[kk], s[u]);      // S^T[kv][q]
            dp[u] = mma_chunk(rtile_chunk<bf16_t, HD>(Vt, 32 * u + l31, 2 * kk + h), dof[kk], dp[u]);   // dP^T[kv][q]
        }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float p = __builtin_amdgcn_exp2f(s[u][r] * sl - lse2);
            if (MASK) p = (kv0 + 32 * u + acc_row(r, h) < N) ? p : 0.f;
            dp[u][r] = p * (dp[u][r] - del);                                   // dS^T / scale (scale applied to dQ once)
        }
#pragma unroll
    for (int c = 0; c < 2 * NU; ++c) {
        const bf16x8 dsb = pack_chunk((const bf16_t*)nullptr, dp, c);
#pragma unroll
        for (int db = 0; db < C::NDB; ++db) dq[db] = mma_chunk(tr_chunk<HD>(Kt, 32 * db, c, lane), dsb, dq[db]);   // dQ^T[d][q]
    }
}

// phase B step (wave owns 32 keys): NU 32-query sub-tiles starting at q0; lse_t / del_t point at q0 (log2-scaled lse,
// +inf on padded queries so that P = 0 there).  Padded KEYS need no mask: a key is a lane here, and whatever a padded
// lane accumulates stays in its own dK/dV columns, which are never stored.
template <int HD, int NU>
__device__ __forceinline__ void bwd_small_kv_step(const char* Qt, const char* dOt, const float* lse_t, const float* del_t,
                                                  const bf16x8 (&kf)[Cfg<bf16_t, HD>::NKK],
                                                  const bf16x8 (&vf)[Cfg<bf16_t, HD>::NKK], float sl,
                                                  f32x16 (&dk)[Cfg<bf16_t, HD>::NDB], f32x16 (&dv)[Cfg<bf16_t, HD>::NDB],
                                                  int lane) {
    typedef Cfg<bf16_t, HD> C;
    const int l31 = lane & 31, h = lane >> 5;
    f32x16 s[2], dp[2];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[u][r] = 0.f; dp[u][r] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < C::NKK; ++kk) {
            s[u] = mma_chunk(rtile_chunk<bf16_t, HD>(Qt, 32 * u + l31, 2 * kk + h), kf[kk], s[u]);      // S[q][kv]
            dp[u] = mma_chunk(rtile_chunk<bf16_t, HD>(dOt, 32 * u + l31, 2 * kk + h), vf[kk], dp[u]);   // dP[q][kv]
        }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 L = *reinterpret_cast<const f32x4*>(lse_t + 32 * u + 8 * g + 4 * h);
            const f32x4 D = *reinterpret_cast<const f32x4*>(del_t + 32 * u + 8 * g + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                const float p = __builtin_amdgcn_exp2f(s[u][r] * sl - L[e]);
                s[u][r] = p;                                   // P
                dp[u][r] = p * (dp[u][r] - D[e]);              // dS / scale (scale applied to dK once)
            }
        }
#pragma unroll
    for (int c = 0; c < 2 * NU; ++c) {
        const bf16x8 pb = pack_chunk((const bf16_t*)nullptr, s, c);
        const bf16x8 dsb = pack_chunk((const bf16_t*)nullptr, dp, c);
#pragma unroll
        for (int db = 0; db < C::NDB; ++db) {
            dv[db] = mma_chunk(tr_chunk<HD>(dOt, 32 * db, c, lane), pb, dv[db]);    // dV^T[d][kv]
            dk[db] = mma_chunk(tr_chunk<HD>(Qt, 32 * db, c, lane), dsb, dk[db]);    // dK^T[d][kv]
        }
    }
}

// One workgroup per (batch, head); LDS holds {Q, K, V, dO} (4 x 32 KB at N = 197).  Once every wave has lifted its K/V row
// fragments for phase B the K/V tiles are dead, and wave w reuses rows [32w, 32w+32) of them as its private scratch for
// the coalesced row stores of dQ, dK and dV.
template <int HD>
__global__ __launch_bounds__(SM_THREADS) void attn_bwd_small_kernel(const bf16_t* __restrict__ qkv, int64_t ld,
                                                                    const bf16_t* __restrict__ out, int64_t ldo,
                                                                    const bf16_t* __restrict__ dout, int64_t lddo,
                                                                    const float* __restrict__ lse,
                                                                    float* __restrict__ delta,
                                                                    bf16_t* __restrict__ dqkv, int64_t lddq, int N, int H,
                                                                    int hd, float scale) {
    typedef Cfg<bf16_t, HD> C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NS = (N + 31) >> 5, NR = NS * 32;
    char* Qs = smem;
    char* Ks = Qs + NR * C::RROW;
    char* Vs = Ks + NR * C::RROW;
    char* dOs = Vs + NR * C::RROW;
    float* lse_s = reinterpret_cast<float*>(dOs + NR * C::RROW);   // [SM_MAXN]  lse * log2(e), +inf on padded rows
    float* del_s = lse_s + SM_MAXN;                                // [SM_MAXN]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int head = blockIdx.x, b = blockIdx.y;
    const int Cdim = H * hd;
    const bf16_t* qptr = qkv + (int64_t)b * N * ld + head * hd;
    const bf16_t* doptr = dout + (int64_t)b * N * lddo + head * hd;
    const int64_t bh = ((int64_t)b * H + head) * N;
    const int row = 32 * wave + l31;        // the query (phase A) / key (phase B) this lane owns
    const bool row_ok = row < N;
    const int rowc = row_ok ? row : N - 1;
    const bool active = 32 * wave < N;      // wave-uniform
    const float sl = scale * LOG2E;

    const int trace_slot = blockIdx.y * gridDim.x + blockIdx.x;
    (void)trace_slot;
    TRACE_STAMP(trace_slot, 0);
    SmallStage<HD, 4> st;
    {
        const bf16_t* const bases[4] = {qptr, qptr + Cdim, qptr + 2 * Cdim, doptr};
        const int64_t ldv[4] = {ld, ld, ld, lddo};
        st.load(bases, ldv, N, hd, tid);
    }
    u32x4 of[C::NKK];      // O row fragments for delta: the same d-slices as the dO operand fragments
    {
        const bf16_t* optr = out + ((int64_t)b * N + rowc) * ldo + head * hd;
#pragma unroll
        for (int kk = 0; kk < C::NKK; ++kk) {
            const int d = (2 * kk + h) * 8;
            const bool ok = d < hd;
            of[kk] = *reinterpret_cast<const u32x4*>(optr + (ok ? d : 0));
            of[kk] = ok ? of[kk] : zero4();
        }
    }
    const float lse_mine = (tid < N && tid < SM_MAXN) ? lse[bh + tid] * LOG2E : INFINITY;
    {
        char* const tiles[4] = {Qs, Ks, Vs, dOs};
        st.store(tiles, NR, tid, N, hd);
    }
    if (tid < SM_MAXN) lse_s[tid] = lse_mine;
    __syncthreads();
    TRACE_STAMP(trace_slot, 1);

    // ---- phase A: delta and dQ for the 32 queries of this wave
    f32x16 dq[C::NDB];
    if (active) {
        bf16x8 qf[C::NKK], dof[C::NKK];
        float del = 0.f;
#pragma unroll
        for (int kk = 0; kk < C::NKK; ++kk) {
            qf[kk] = rtile_chunk<bf16_t, HD>(Qs, row, 2 * kk + h);
            dof[kk] = rtile_chunk<bf16_t, HD>(dOs, row, 2 * kk + h);
            const u32x4 rd = *reinterpret_cast<const u32x4*>(&dof[kk]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                del += __uint_as_float(of[kk][e] << 16) * __uint_as_float(rd[e] << 16) +
                       __uint_as_float(of[kk][e] & 0xffff0000u) * __uint_as_float(rd[e] & 0xffff0000u);
        }
        del += __shfl_xor(del, 32, 64);
        if (h == 0) {
            del_s[row] = del;
            if (row_ok && delta) delta[bh + row] = del;
        }
        const float lse2 = lse_s[row];      // +inf on padded queries -> P = 0
#pragma unroll
        for (int db = 0; db < C::NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
        int t = 0;
        for (; 32 * (t + 2) <= N; t += 2)
            bwd_small_q_step<HD, 2, false>(Ks + 32 * t * C::RROW, Vs + 32 * t * C::RROW, 32 * t, N, qf, dof, sl, lse2, del, dq, lane);
        if (t + 2 <= NS)
            bwd_small_q_step<HD, 2, true>(Ks + 32 * t * C::RROW, Vs + 32 * t * C::RROW, 32 * t, N, qf, dof, sl, lse2, del, dq, lane);
        else if (t < NS)
            bwd_small_q_step<HD, 1, true>(Ks + 32 * t * C::RROW, Vs + 32 * t * C::RROW, 32 * t, N, qf, dof, sl, lse2, del, dq, lane);
    }
    TRACE_STAMP(trace_slot, 2);
    __syncthreads();      // del_s complete; every phase-A loop over the K/V tiles is finished
    if (!active) {
        __syncthreads();
        return;
    }
    bf16x8 kf[C::NKK], vf[C::NKK];
#pragma unroll
    for (int kk = 0; kk < C::NKK; ++kk) {
        kf[kk] = rtile_chunk<bf16_t, HD>(Ks, row, 2 * kk + h);
        vf[kk] = rtile_chunk<bf16_t, HD>(Vs, row, 2 * kk + h);
    }
    __syncthreads();      // K/V tiles are dead: rows [32 wave, +32) of each are this wave's scratch from here on
    TRACE_STAMP(trace_slot, 3);
    char* scr0 = Ks + 32 * wave * C::RROW;
    char* scr1 = Vs + 32 * wave * C::RROW;
    bf16_t* grow0 = dqkv + ((int64_t)b * N + 32 * wave) * lddq + head * hd;
    store_rows_via_lds<HD>(scr0, dq, scale, grow0, lddq, N - 32 * wave, hd, lane);

    TRACE_STAMP(trace_slot, 4);
    // ---- phase B: dK and dV for the 32 keys of this wave
    f32x16 dk[C::NDB], dv[C::NDB];
#pragma unroll
    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
    int t = 0;
    for (; t + 2 <= NS; t += 2)
        bwd_small_kv_step<HD, 2>(Qs + 32 * t * C::RROW, dOs + 32 * t * C::RROW, lse_s + 32 * t, del_s + 32 * t, kf, vf, sl, dk, dv, lane);
    if (t < NS)
        bwd_small_kv_step<HD, 1>(Qs + 32 * t * C::RROW, dOs + 32 * t * C::RROW, lse_s + 32 * t, del_s + 32 * t, kf, vf, sl, dk, dv, lane);
    TRACE_STAMP(trace_slot, 5);
    store_rows_via_lds<HD>(scr0, dk, scale, grow0 + Cdim, lddq, N - 32 * wave, hd, lane);
    store_rows_via_lds<HD>(scr1, dv, 1.0f, grow0 + 2 * Cdim, lddq, N - 32 * wave, hd, lane);
    TRACE_STAMP(trace_slot, 6);
}

// 16 accumulator rows (lane (row l & 15, g = l >> 4) holds d = 16 dt + 4 g + j) -> bf16 rows of `grow0`, transposed through the
// wave's LDS scratch into whole 128-byte row stores.  The stores are BUFFER stores behind a descriptor that ends with the last
// valid row (rows past it and chunks past hd are dropped by the bounds check, never branched over): the number of memory
// operations is fixed, so a later wait for loads issued BEFORE these stores can be a counted vmcnt instead of vmcnt(0).
template <int HD>
__device__ __forceinline__ void r16_store_rows(char* scr, const f32x4 (&acc)[HD / 16], float mul, bf16_t* __restrict__ grow0,
                                               int64_t ldg, int rows_valid, int hd, int lane) {
    typedef Cfg<bf16_t, HD> C;
    const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) {
        bf16x4 v4;
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[e] = (bf16_t)(acc[dt][e] * mul);
        *reinterpret_cast<bf16x4*>(scr + l15 * C::RROW + (16 * dt + 4 * g) * 2) = v4;
    }
    __builtin_amdgcn_wave_barrier();
    constexpr int RPI = 64 / C::CPR;              // rows per store instruction
    const int chunk = lane % C::CPR, r0 = lane / C::CPR;
    const int nrec = rows_valid > 0 ? (int)(((int64_t)(rows_valid - 1) * ldg + hd) * 2) : 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(grow0, 0, nrec, 0x00020000);
#pragma unroll
    for (int k = 0; k < 16 / RPI; ++k) {
        const int r = r0 + RPI * k;
        const u32x4 v = *reinterpret_cast<const u32x4*>(scr + r * C::RROW + chunk * 16);
        const int off = chunk * 8 < hd ? (int)(r * ldg * 2) + chunk * 16 : 0x7f000000;
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, ME_POL_ATTN_ST);
    }
    __builtin_amdgcn_wave_barrier();
}

// -----------------------------------------------------------------------------------------------------
// Long sequences (N > SM_MAXN), forward: the ring form with K / V STREAMED.  A workgroup (16 waves: up to 14 compute waves of 16
// queries + the two loader waves) owns a block of queries of one (batch, head) and walks the keys in chunks of 128; the loaders
// keep a three-slot LDS ring of {K, V} chunks two chunks ahead of the compute waves (one barrier per chunk: "chunk k has landed
// and everybody is done with chunk k - 1", whose slot the loaders then refill with chunk k + 2).  Online softmax per chunk (the
// row block no longer fits in registers); the same 16 x 16 x 32 products, operand swizzle and hand pipelining as the ring form.
// Replaces the one-workgroup-per-(batch, head) "mid" kernel and the 256-row "chunk" kernel in the forward: those ran 8 waves of
// 32 rows (two per SIMD) with register-staged loads.  Workgroups of one XCD take neighbouring query blocks of the same heads,
// so a head's K / V comes out of HBM once and is re-read from that XCD's L2.
// (Measured and dropped: 64-key chunks with the NEXT chunk's QK^T MFMAs issued ahead of this chunk's softmax arithmetic in every
// wave -- 511 us against 444 at N = 1568: twice the barriers, and the chunk barrier keeps all waves of a SIMD in the same phase;
// skipping the 64-key groups of the last chunk that lie past N with wave-uniform branches -- slower, 502 against 472 us: the joins
// cost the straight-line schedule more than the skipped MFMAs save.)
constexpr int ST_KC = 128;                            // keys per chunk
constexpr int ST_RING = 3;
template <int HD>
__global__ __launch_bounds__(R16_THREADS) void attn_fwd_stream16_kernel(const bf16_t* __restrict__ qkv, int64_t ld,
                                                                        bf16_t* __restrict__ out, int64_t ldo,
                                                                        float* __restrict__ lse, int N, int H, int hd, float scale,
                                                                        int QB, int nqb, int items) {
    typedef Cfg<bf16_t, HD> C;
    typedef RCfg<HD> R;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int arr_bytes = ST_KC * R::RB;          // one array (K or V) of one chunk
    constexpr int slot_bytes = 2 * arr_bytes;
    constexpr int NPC = ST_KC / R::RPI;               // DMA pieces per array and chunk
    constexpr int NTC = ST_KC / 16;                   // 16-key tiles per chunk
    constexpr int NKS = HD / 32, NDT = HD / 16, NH = NDT / 2;
    constexpr int SCR = 16 * C::RROW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int Cdim = H * hd;
    const int G = (int)gridDim.x;
    const int NCH = (N + ST_KC - 1) / ST_KC;
    // XCD-major virtual id: the workgroups of one XCD (blockIdx & 7) take consecutive items = neighbouring query blocks of a head
    const int vid = (G & 7) == 0 ? ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int my_items = vid < items ? (items - vid + G - 1) / G : 0;      // items vid, vid + G, ...
    const bool compute = 16 * wave < QB;              // wave-uniform

    if (!compute) {
        // ---- loader waves: the last two (QB = 224) or four (fewer compute waves) waves; the others only keep the barrier count.
        // With four, each array of a chunk is split between two loaders -- an LDS-DMA piece costs its issuer 60-180 clocks, and 16 of
        // them per 128-key chunk is about what the compute waves of a short query block need for the chunk.
        const int NL = QB <= 12 * 16 ? 4 : 2;
        if (wave < 16 - NL) {
            for (int k = 0; k < my_items * NCH; ++k) __builtin_amdgcn_s_barrier();
            return;
        }
        const int lw = wave - (16 - NL);              // loader index
        const int which = lw & 1;                     // 0: K, 1: V
        const int part = lw >> 1, nparts = NL >> 1;   // which pieces of the array
        const int rg = (lane * 16) / R::RB, pos = ((lane * 16) % R::RB) / 16;
        const int csrc = pos ^ r16_swz<R::CPR>(rg);
        const int dma_voff = (csrc * 8 < hd) ? rg * (int)ld * 2 + csrc * 16 : 0x7f000000;
        const int dma_gstep = R::RPI * (int)ld * 2;
        const int rec_bytes = (int)(((int64_t)(N - 1) * ld + hd) * 2);
        const int total = my_items * NCH;
        // chunk j of the workgroup's stream = chunk j % NCH of its (j / NCH)-th item
        auto fill = [&](int j) {
            if (j >= total) return;
            const int it = vid + (j / NCH) * G, c = j % NCH;
            const int bh = it / nqb;
            const bf16_t* base = qkv + (int64_t)(bh / H) * N * ld + (bh % H) * hd + (1 + which) * Cdim;
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base), 0, rec_bytes, 0x00020000);
            char* dst = smem + (j % ST_RING) * slot_bytes + which * arr_bytes;
            const int voff = dma_voff + c * NPC * dma_gstep;
            if (nparts == 1) {
#pragma unroll
                for (int i = 0; i < NPC; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_dma_t*)(dst + i * 1024), 16, voff + i * dma_gstep, 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < NPC / 2; ++i) {
                    const int p = 2 * i + part;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_dma_t*)(dst + p * 1024), 16, voff + p * dma_gstep, 0, 0, 0);
                }
            }
        };
        fill(0);
        fill(1);
        for (int k = 0; k < total; ++k) {
            // chunk k has landed once at most chunk k + 1's pieces (of this loader) are outstanding (in-order retirement)
            if (k + 1 >= total) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (nparts == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPC) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPC / 2) : "memory");
            __builtin_amdgcn_s_barrier();             // chunk k ready; everybody is done with chunk k - 1
            fill(k + 2);                              // ... whose slot takes chunk k + 2
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ---- compute waves
    char* scr = smem + ST_RING * slot_bytes + wave * SCR;
    const float sl = scale * LOG2E;
    int koff[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) koff[ks] = l15 * R::RB + 16 * ((4 * ks + g) ^ r16_swz<R::CPR>(l15));
    int voff[NDT];
    {
        const int rr = 4 * g + (l15 >> 2);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
            voff[dt] = rr * R::RB + 16 * ((2 * dt + ((l15 & 3) >> 1)) ^ r16_swz<R::CPR>(rr)) + 8 * (l15 & 1);
    }
    const f32x4 zero4f = {0.f, 0.f, 0.f, 0.f};
    // the wave's own query rows: item ii + 1's are fetched under item ii's last chunk
    u32x4 qn[NKS];
    auto q_issue = [&](int it) {
        const int bh = it / nqb, qb = it % nqb;
        const int q = qb * QB + 16 * wave + l15;
        const bf16_t* qptr = qkv + ((int64_t)(bh / H) * N + (q < N ? q : N - 1)) * ld + (bh % H) * hd;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d = 32 * ks + 8 * g;
            qn[ks] = *reinterpret_cast<const u32x4*>(qptr + (d < hd ? d : 0));
        }
    };
    if (my_items > 0) q_issue(vid);
    int j = 0;                                        // chunk index in the workgroup's stream
    for (int ii = 0; ii < my_items; ++ii) {
        const int it = vid + ii * G;
        const int bh = it / nqb, qb = it % nqb;
        const int b = bh / H, head = bh % H;
        const int q0 = qb * QB + 16 * wave;           // first query of this wave
        const int q = q0 + l15;
        bf16x8 qf[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const u32x4 raw = 32 * ks + 8 * g < hd ? qn[ks] : zero4();      // chunks of d past hd are zeros
            qf[ks] = *reinterpret_cast<const bf16x8*>(&raw);
        }
        f32x4 o[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[dt] = zero4f;
        float m_run = -INFINITY, l_run = 0.f;         // l_run: this lane's own keys only (summed over the four key groups at the end)
        for (int c = 0; c < NCH; ++c, ++j) {
            __syncthreads();                          // chunk j has landed (the loaders waited for it); see the loader loop
            if (c == NCH - 1) q_issue(ii + 1 < my_items ? it + G : it);
            const char* Kb = smem + (j % ST_RING) * slot_bytes;
            const char* Vb = Kb + arr_bytes;
            const int nkeys = N - c * ST_KC;          // valid keys of this chunk (>= 1); wave-uniform
            // ---- S^T for the chunk's 16-key tiles, two at a time, operands one pair ahead
            f32x4 s[NTC];
            bf16x8 ka[2][NKS], kb[2][NKS];
            auto kread = [&](int t, bf16x8 (&dst)[2][NKS]) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) dst[tt][ks] = *reinterpret_cast<const bf16x8*>(Kb + 16 * (t + tt) * R::RB + koff[ks]);
            };
            auto kmma = [&](int t, const bf16x8 (&src)[2][NKS]) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) s[t + tt] = mma16(src[tt][0], qf[0], zero4f);
#pragma unroll
                for (int ks = 1; ks < NKS; ++ks)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) s[t + tt] = mma16(src[tt][ks], qf[ks], s[t + tt]);
            };
            kread(0, ka);
#pragma unroll
            for (int t = 0; t < NTC; t += 4) {
                kread(t + 2, kb);
                __builtin_amdgcn_sched_barrier(0);
                kmma(t, ka);
                __builtin_amdgcn_sched_barrier(0);
                if (t + 4 < NTC) kread(t + 4, ka);
                __builtin_amdgcn_sched_barrier(0);
                kmma(t + 2, kb);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- online softmax
            if (nkeys < ST_KC) {
                int nk = nkeys - 4 * g;               // (opaque: hipcc would hoist the lane masks out of the loops into SGPRs)
                asm volatile("" : "+v"(nk));
#pragma unroll
                for (int t = 0; t < NTC; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[t][e] = (16 * t + e < nk) ? s[t][e] : -INFINITY;
            }
            float mt = -INFINITY;
#pragma unroll
            for (int t = 0; t < NTC; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) mt = fmaxf(mt, s[t][e]);
            mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            const float m_new = fmaxf(m_run, mt * sl);     // every chunk holds a valid key: finite from the first chunk on
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float ps = 0.f;
#pragma unroll
            for (int t = 0; t < NTC; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pe = __builtin_amdgcn_exp2f(s[t][e] * sl - m_new);
                    s[t][e] = pe;
                    ps += pe;
                }
            l_run = l_run * alpha + ps;
            m_run = m_new;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) o[dt] *= alpha;
            // ---- O^T += V^T P^T, 32 keys at a time, d tiles in two halves (operands one half ahead)
            bf16x8 va[NH], vb[NH];
            auto vread = [&](int kk, int half, bf16x8 (&dst)[NH]) {
#pragma unroll
                for (int i = 0; i < NH; ++i) {
                    union { bf16x4 q4[2]; bf16x8 v; } a;
#pragma unroll
                    for (int r = 0; r < 2; ++r)
                        a.q4[r] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                            (lds_bf16x4_t*)((uint32_t)(uintptr_t)Vb + (32 * kk + 16 * r) * R::RB + voff[NH * half + i]));
                    dst[i] = a.v;
                }
            };
            vread(0, 0, va);
#pragma unroll
            for (int kk = 0; kk < ST_KC / 32; ++kk) {
                vread(kk, 1, vb);
                bf16x8 pb;
#pragma unroll
                for (int e = 0; e < 4; ++e) { pb[e] = (bf16_t)s[2 * kk][e]; pb[4 + e] = (bf16_t)s[2 * kk + 1][e]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NH; ++i) o[i] = mma16(va[i], pb, o[i]);
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 1 < ST_KC / 32) vread(kk + 1, 0, va);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NH; ++i) o[NH + i] = mma16(vb[i], pb, o[NH + i]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- finish the item: row sums over the four key groups, normalise, store
        l_run += __shfl_xor(l_run, 16, 64);
        l_run += __shfl_xor(l_run, 32, 64);
        const float inv = __builtin_amdgcn_rcpf(l_run);
        r16_store_rows<HD>(scr, o, inv, out + ((int64_t)b * N + q0) * ldo + head * hd, ldo, N - q0, hd, lane);
        if (lse && g == 0 && q < N) lse[((int64_t)b * H + head) * N + q] = (m_run + __builtin_amdgcn_logf(l_run)) * LN2;
    }
}

// a softmax probability from its log2-domain argument: v_exp_f32 with the clamp output modifier (hipcc folds the med3 into it)
__device__ __forceinline__ float r16_p(float arg) { return __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(arg), 0.f, 1.f); }

// four probabilities and their dS / scale (the 32-key dK / dV kernel):  pe = exp2(s * sl - l2),  ds = pe * (dp - d).  ME_R16_PK = 1 (A/B arm, OFF): the
// same on PAIRS -- v_pk_fma_f32 for the argument, v_pk_mul_f32 + v_pk_fma_f32 for dS, 3.5 instead of 5 VALU instructions per score (and, tried with
// it, the same form in every other backward kernel and v_pk_fma_f32 / v_pk_add_f32 in the forward kernels' exp2 + row sum) -- measured SLOWER everywhere (same box,
// profiles/r06_attn_bwd_stream32.txt: N = 197 forward 68.7 -> 74.7 us, backward 212.9 -> 222.7; N = 1568 forward 477 -> 504, backward
// 1 218 -> 1 247): the packed fp32 operations do not issue at the rate of two scalar ones here, and assembling the pairs costs moves.
#ifndef ME_R16_PK
#define ME_R16_PK 0
#endif
__device__ __forceinline__ void r16_pds4(const f32x4& s, const f32x4& dp, float sl, const f32x4& l2, const f32x4& d, f32x4& pe, f32x4& ds) {
#if ME_R16_PK
    const f32x2 slv = {sl, sl};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2 arg = f32x2{s[2 * h], s[2 * h + 1]} * slv - f32x2{l2[2 * h], l2[2 * h + 1]};
        const f32x2 p = {r16_p(arg[0]), r16_p(arg[1])};
        const f32x2 x = p * f32x2{dp[2 * h], dp[2 * h + 1]} - p * f32x2{d[2 * h], d[2 * h + 1]};
        pe[2 * h] = p[0]; pe[2 * h + 1] = p[1];
        ds[2 * h] = x[0]; ds[2 * h + 1] = x[1];
    }
#else
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        pe[e] = r16_p(s[e] * sl - l2[e]);
        ds[e] = pe[e] * (dp[e] - d[e]);
    }
#endif
}

// -----------------------------------------------------------------------------------------------------
// Backward in the ring form (SM_MINN < N <= RS_MAXN), 16 waves, 16 rows per wave (see attn_fwd_ring16_kernel for why).
// LDS holds two 2-array regions, KV = {K, V} and QD = {Q, dO}, unpadded and swizzled (r16_swz).  Per item:
//   phase A (wave = 16 queries):  S^T, dP^T from the KV region + the wave's own Q / dO rows (registers, prefetched from global),
//                                 dS^T = P^T o (dP^T - delta), dQ^T += K^T dS^T;  delta = rowsum(dO o O) on the way in;
//                                 at the end the wave lifts its own 16 K / V rows (phase B's register operands)       -- barrier 1
//   phase B (wave = 16 keys):     S, dP from the QD region + those rows, dV^T += dO^T P, dK^T += Q^T dS                  -- barrier 2
// The KV region is dead during phase B and the QD region during phase A: the two loader waves (14: K then Q, 15: V then dO)
// refill each region by LDS-DMA while the OTHER phase computes -- the whole item's 100 KB arrives under compute, which the
// one-workgroup-per-item kernel above exposes completely (13 k of 40 k clocks per item).  S and dP are computed twice (7 MFMA
// products instead of 5) so that no gradient needs a cross-wave reduction; results are deterministic.
template <int HD, int NS>
__global__ __launch_bounds__(R16_THREADS) void attn_bwd_ring16_kernel(const bf16_t* __restrict__ qkv, int64_t ld,
                                                                      const bf16_t* __restrict__ out, int64_t ldo,
                                                                      const bf16_t* __restrict__ dout, int64_t lddo,
                                                                      const float* __restrict__ lse, float* __restrict__ delta,
                                                                      bf16_t* __restrict__ dqkv, int64_t lddq, int N, int H, int hd,
                                                                      float scale, int items, unsigned* __restrict__ ctr) {
    typedef Cfg<bf16_t, HD> C;
    typedef RCfg<HD> R;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NR = NS * 32;
    constexpr int arr_bytes = NR * R::RB;
    constexpr int NRI = NR / R::RPI;
    constexpr int NKS = HD / 32;
    constexpr int NDT = HD / 16;
    constexpr int SCR = 16 * C::RROW;
    char* const KVr = smem;                           // {K, V}
    char* const QDr = smem + 2 * arr_bytes;           // {Q, dO}
    float* const lse_s = reinterpret_cast<float*>(smem + 4 * arr_bytes);      // [NR] lse * log2(e), +inf on padded queries
    float* const del_s = lse_s + NR;                                          // [NR]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int Cdim = H * hd;
    const int G = (int)gridDim.x;
    const bool active = 16 * wave < N;                // wave-uniform
    // ---- items are CLAIMED (ctr != null; null under hipGraph capture = the static schedule it, it + G, ...): a workgroup's first item
    // is its block index, every further one a draw from a counter -- a workgroup that starts late, because its CU was held by a
    // communication kernel when this one was launched, then simply gets fewer items instead of making the whole launch wait for its
    // full static share (the resident GEMM does the same: gemm3.hip).  Wave 0 draws the next item's id during phase A and leaves it in
    // an LDS word ahead of barrier 1; everybody reads it behind that barrier.  Draws of one launch: (items - G) hits + one miss per
    // workgroup = items; whoever draws the last one zeroes the counter for the next launch on this stream.
    volatile int* const nextw = reinterpret_cast<volatile int*>(smem + 4 * arr_bytes + 2 * NR * (int)sizeof(float) + 16 * SCR);
    // The draw is issued at the head of phase A and its result picked up at the end of it (a returning atomic takes a couple of
    // microseconds; consumed where it is issued, wave 0 would sit on it and hold barrier 1 back -- measured +5 %): the atomic goes out
    // through inline asm with only lane 0 enabled, so that the compiler sees no pending result to wait for; the explicit vmcnt(0)
    // in draw_take() is the wait.
    auto draw_issue = [&]() -> unsigned {            // (wave 0 only)
        unsigned old = 0, one = 1;
        if (ctr) {
            unsigned long long saved;
            asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, %4 sc0\n\ts_mov_b64 exec, %1"
                         : "=&v"(old), "=&s"(saved) : "v"(0u), "v"(one), "s"(ctr) : "memory");
        }
        return old;                                  // valid in lane 0 once the operation has retired
    };
    auto draw_take = [&](unsigned drawn, int it) {   // (wave 0 only)
        int nx = it + G;
        if (ctr) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(drawn) : : "memory");
            const unsigned t = __builtin_amdgcn_readfirstlane(drawn);
            if (t == (unsigned)(items > G ? items : G) - 1u && lane == 0) atomicExch(ctr, 0u);
            nx = G + (int)t;
        }
        if (lane == 0) *nextw = nx;
    };
    if (!active) {
        if (wave < 14) {                              // spare waves: the barrier count only
            __syncthreads();
            for (int it = blockIdx.x; it < items;) {
                __syncthreads();
                it = __builtin_amdgcn_readfirstlane(*nextw);
                __syncthreads();
            }
            return;
        }
        // ---- loader waves: 14 fills the first array of a region (K, Q), 15 the second (V, dO)
        const int which = wave - 14;
        const int rg = (lane * 16) / R::RB, pos = ((lane * 16) % R::RB) / 16;
        const int csrc = pos ^ r16_swz<R::CPR>(rg);
        const bool cok = csrc * 8 < hd;
        const int voff_a = cok ? rg * (int)ld * 2 + csrc * 16 : 0x7f000000;              // rows of qkv
        const int voff_d = cok ? rg * (int)(which ? lddo : ld) * 2 + csrc * 16 : 0x7f000000;      // rows of the QD array of this loader
        const int gstep_a = R::RPI * (int)ld * 2, gstep_d = R::RPI * (int)(which ? lddo : ld) * 2;
        const int rec_a = (int)(((int64_t)(N - 1) * ld + hd) * 2);
        const int rec_d = (int)(((int64_t)(N - 1) * (which ? lddo : ld) + hd) * 2);
        auto fill = [&](const bf16_t* base, int rec, int voff, int gstep, char* dst, int paced) {
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base), 0, rec, 0x00020000);
#pragma unroll
            for (int i = 0; i < NRI; ++i) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_dma_t*)(dst + i * 1024), 16, voff + i * gstep, 0, 0, 0);
                if (paced == 1) __builtin_amdgcn_s_sleep(1);
                if (paced == 2) __builtin_amdgcn_s_sleep(3);
            }
        };
        auto kv_base = [&](int it) { return qkv + (int64_t)(it / H) * N * ld + (it % H) * hd + (1 + which) * Cdim; };
        auto qd_base = [&](int it) {
            return which ? dout + (int64_t)(it / H) * N * lddo + (it % H) * hd : qkv + (int64_t)(it / H) * N * ld + (it % H) * hd;
        };
        int it = blockIdx.x;
        fill(kv_base(it), rec_a, voff_a, gstep_a, KVr + which * arr_bytes, 0);
        __syncthreads();
        while (it < items) {
            fill(qd_base(it), rec_d, voff_d, gstep_d, QDr + which * arr_bytes, 1);
            __syncthreads();                          // barrier 1 (vmcnt(0) first: the fill has landed)
            it = __builtin_amdgcn_readfirstlane(*nextw);      // the next item of this workgroup (>= items: none)
            if (it < items) fill(kv_base(it), rec_a, voff_a, gstep_a, KVr + which * arr_bytes, 2);
            __syncthreads();                          // barrier 2
        }
        return;
    }

    // ---- compute waves
    char* scr = smem + 4 * arr_bytes + 2 * NR * (int)sizeof(float) + wave * SCR;
    const int row = 16 * wave + l15;                  // the query (phase A) / key (phase B) of this lane
    const bool row_ok = row < N;
    const int rowc = row_ok ? row : N - 1;
    const float sl = scale * LOG2E;
    int koff[NKS];                                    // row-operand read: row 16 t + l15, chunk 4 ks + g
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) koff[ks] = l15 * R::RB + 16 * ((4 * ks + g) ^ r16_swz<R::CPR>(l15));
    int troff[NDT];                                   // transposing read: rows 32 kk + 16 r + 4 g + (l15 >> 2), 8 bytes at d = 16 dt + 4 (l15 & 3)
    {
        const int rr = 4 * g + (l15 >> 2);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
            troff[dt] = rr * R::RB + 16 * ((2 * dt + ((l15 & 3) >> 1)) ^ r16_swz<R::CPR>(rr)) + 8 * (l15 & 1);
    }
    for (int i = tid; i < NR; i += 64 * ((N + 15) / 16)) { lse_s[i] = INFINITY; del_s[i] = 0.f; }      // rows no wave owns stay this way

    // own-row operands of the NEXT item's phase A, in flight across this item's phase B
    u32x4 qn[NKS], don[NKS], on[NKS];
    float lse_n = 0.f;
    // piece i of 3 NKS + 1: issued one per phase-B step -- asked for at once (13 waves x 7 loads on top of the loaders' DMA and the
    // dQ stores) the CU's vector-memory path backs up and every wave sits in the issue of its loads for thousands of clocks
    constexpr int NPF = 3 * NKS + 1;
    auto prefetch = [&](int it, int i) {
        const int b = it / H, head = it % H;
        const int ks = i % NKS, arr = i / NKS;
        const int d = 32 * ks + 8 * g, dc = d < hd ? d : 0;
        if (arr == 0) qn[ks] = *reinterpret_cast<const u32x4*>(qkv + ((int64_t)b * N + rowc) * ld + head * hd + dc);
        else if (arr == 1) don[ks] = *reinterpret_cast<const u32x4*>(dout + ((int64_t)b * N + rowc) * lddo + head * hd + dc);
        else if (arr == 2) on[ks] = *reinterpret_cast<const u32x4*>(out + ((int64_t)b * N + rowc) * ldo + head * hd + dc);
        else lse_n = lse[((int64_t)b * H + head) * N + rowc];
    };
    auto rowread = [&](const char* arr, int t, bf16x8 (&dst)[NKS]) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) dst[ks] = *reinterpret_cast<const bf16x8*>(arr + 16 * t * R::RB + koff[ks]);
    };
    auto trread = [&](const char* arr, int kk, int dt) {
        union { bf16x4 q4[2]; bf16x8 v; } a;
#pragma unroll
        for (int r = 0; r < 2; ++r)
            a.q4[r] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)((uint32_t)(uintptr_t)arr + (32 * kk + 16 * r) * R::RB + troff[dt]));
        return a.v;
    };
    const f32x4 zero4f = {0.f, 0.f, 0.f, 0.f};

    int it = blockIdx.x;
#pragma unroll
    for (int i = 0; i < NPF; ++i) prefetch(it, i);
    __syncthreads();
    while (it < items) {
        const int b = it / H, head = it % H;
        const int64_t bh = ((int64_t)b * H + head) * N;
        TRACE_STAMP(it, 0);
        unsigned drawn = 0;
        if (wave == 0) drawn = draw_issue();
        // ---- phase A: this wave's 16 queries.  Own rows: chunks past hd are zeros; delta from the O / dO fragments
        bf16x8 qf[NKS], dof[NKS];
        float del = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const bool ok = 32 * ks + 8 * g < hd;
            const u32x4 qv = ok ? qn[ks] : zero4(), dv4 = ok ? don[ks] : zero4(), ov = ok ? on[ks] : zero4();
            qf[ks] = *reinterpret_cast<const bf16x8*>(&qv);
            dof[ks] = *reinterpret_cast<const bf16x8*>(&dv4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                del += __uint_as_float(ov[e] << 16) * __uint_as_float(dv4[e] << 16) +
                       __uint_as_float(ov[e] & 0xffff0000u) * __uint_as_float(dv4[e] & 0xffff0000u);
        }
        del += __shfl_xor(del, 16, 64);
        del += __shfl_xor(del, 32, 64);
        // +inf on padded queries -> P = 0.  Valid rows keep their exact lse; a padded key (zero K row, s = 0) evaluates to
        // exp2(-lse2), which overflows for a row whose scores are all very negative -- and it only ever multiplies zeros.  Every P is
        // therefore taken through r16_p(): exp2 with the result clamped to [0, 1] (the output modifier of v_exp_f32, no extra
        // instruction), exact for real entries (P <= 1 by construction of lse) and finite for padded ones
        const float lse2 = row_ok ? lse_n * LOG2E : INFINITY;
        if (g == 0) {
            lse_s[row] = lse2;
            del_s[row] = del;
            if (row_ok && delta) delta[bh + row] = del;
        }
        TRACE_STAMP(it, 1);
        f32x4 dq[NDT];
#pragma unroll
        for (int kk = 0; kk < NS; ++kk) {
            bf16x8 ka[2][NKS], va[2][NKS];
            rowread(KVr, 2 * kk, ka[0]); rowread(KVr, 2 * kk + 1, ka[1]);
            rowread(KVr + arr_bytes, 2 * kk, va[0]); rowread(KVr + arr_bytes, 2 * kk + 1, va[1]);
            f32x4 s[2], dp[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) { s[tt] = mma16(ka[tt][0], qf[0], zero4f); dp[tt] = mma16(va[tt][0], dof[0], zero4f); }
#pragma unroll
            for (int ks = 1; ks < NKS; ++ks)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) { s[tt] = mma16(ka[tt][ks], qf[ks], s[tt]); dp[tt] = mma16(va[tt][ks], dof[ks], dp[tt]); }
            bf16x8 kt[NDT];
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) kt[dt] = trread(KVr, kk, dt);
            // keys >= N: their K rows are zeros, so whatever dS holds there adds nothing to dQ -- no mask
            bf16x8 dsb;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pe = r16_p(s[tt][e] * sl - lse2);
                    dsb[4 * tt + e] = (bf16_t)(pe * (dp[tt][e] - del));      // dS^T / scale (scale applied to dQ once)
                }
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) dq[dt] = mma16(kt[dt], dsb, kk == 0 ? zero4f : dq[dt]);
        }
        // this wave's own K / V rows = phase B's register operands (the loaders overwrite the region behind barrier 1)
        bf16x8 kf[NKS], vf[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            kf[ks] = *reinterpret_cast<const bf16x8*>(KVr + 16 * wave * R::RB + koff[ks]);
            vf[ks] = *reinterpret_cast<const bf16x8*>(KVr + arr_bytes + 16 * wave * R::RB + koff[ks]);
        }
        TRACE_STAMP(it, 2);
        if (wave == 0) draw_take(drawn, it);
        __syncthreads();      // barrier 1: KV region dead; {Q, dO} of this item landed; lse_s / del_s complete; next item's id published
        TRACE_STAMP(it, 3);
        const int nid = __builtin_amdgcn_readfirstlane(*nextw);      // (wave-uniform: keep the address arithmetic that follows scalar)
        const int nxt = nid < items ? nid : it;      // (own rows of the next item, or a harmless re-read of this one's)
        bf16_t* grow0 = dqkv + ((int64_t)b * N + 16 * wave) * lddq + head * hd;
        r16_store_rows<HD>(scr, dq, scale, grow0, lddq, N - 16 * wave, hd, lane);
        TRACE_STAMP(it, 4);
        // ---- phase B: this wave's 16 keys.  Padded keys need no mask: a key is a lane (column) here, whatever it accumulates
        // stays in its own dK / dV rows, which are never stored; padded queries have lse_s = +inf -> P = 0
        f32x4 dk[NDT], dv[NDT];
#pragma unroll
        for (int qq = 0; qq < NS; ++qq) {
#pragma unroll
            for (int i = qq * NPF / NS; i < (qq + 1) * NPF / NS; ++i) prefetch(nxt, i);
            f32x4 s[2], dp[2];
            {
                bf16x8 qa[2][NKS];
                rowread(QDr, 2 * qq, qa[0]); rowread(QDr, 2 * qq + 1, qa[1]);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) s[tt] = mma16(qa[tt][0], kf[0], zero4f);
#pragma unroll
                for (int ks = 1; ks < NKS; ++ks)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) s[tt] = mma16(qa[tt][ks], kf[ks], s[tt]);
            }
            {
                bf16x8 da[2][NKS];
                rowread(QDr + arr_bytes, 2 * qq, da[0]); rowread(QDr + arr_bytes, 2 * qq + 1, da[1]);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) dp[tt] = mma16(da[tt][0], vf[0], zero4f);
#pragma unroll
                for (int ks = 1; ks < NKS; ++ks)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) dp[tt] = mma16(da[tt][ks], vf[ks], dp[tt]);
            }
            bf16x8 pb, dsb;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const f32x4 L = *reinterpret_cast<const f32x4*>(lse_s + 16 * (2 * qq + tt) + 4 * g);
                const f32x4 D = *reinterpret_cast<const f32x4*>(del_s + 16 * (2 * qq + tt) + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pe = r16_p(s[tt][e] * sl - L[e]);
                    pb[4 * tt + e] = (bf16_t)pe;
                    dsb[4 * tt + e] = (bf16_t)(pe * (dp[tt][e] - D[e]));      // dS / scale (scale applied to dK once)
                }
            }
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) dv[dt] = mma16(trread(QDr + arr_bytes, qq, dt), pb, qq == 0 ? zero4f : dv[dt]);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) dk[dt] = mma16(trread(QDr, qq, dt), dsb, qq == 0 ? zero4f : dk[dt]);
        }
        TRACE_STAMP(it, 5);
        r16_store_rows<HD>(scr, dk, scale, grow0 + Cdim, lddq, N - 16 * wave, hd, lane);
        r16_store_rows<HD>(scr, dv, 1.0f, grow0 + 2 * Cdim, lddq, N - 16 * wave, hd, lane);
        TRACE_STAMP(it, 6);
        __syncthreads();      // barrier 2: QD region dead; {K, V} of the next item landed
        TRACE_STAMP(it, 7);
        it = nid;
    }
}

// -----------------------------------------------------------------------------------------------------
// Long sequences (N >= ST_BWD_MINN), backward: the two phases of the ring backward as two STREAMING kernels in the form of
// attn_fwd_stream16_kernel (16-row waves, three-slot LDS ring of 128-row chunks kept two chunks ahead by loader waves, one barrier
// per chunk, XCD-major items so that a head's streamed side comes out of HBM once):
//   attn_bwd_dq_stream16_kernel:    a workgroup owns <= 224 queries of a (batch, head) -- own Q / dO / O rows in registers, delta on the
//                                   way in (written out for the second kernel) -- and streams {K, V};       phase A of the ring form
//   attn_bwd_dkdv_stream16_kernel:  a workgroup owns <= 224 keys -- own K / V rows in registers -- and streams {Q, dO} together with the
//                                   chunk's lse / delta (dword LDS-DMA into a small float ring);               phase B of the ring form
// No masks: rows past N arrive as zeros from the bounds-checked DMA (K = V = 0 -> dS meets a zero K row; Q = dO = 0, lse = delta =
// 0 -> P = 1 meets a zero dO row and dS = 0), every P goes through r16_p() and stays finite, and rows a wave owns past N are
// never stored.  Replaces the 8-wave / 32-row chunk kernels (same box, profiles/r04_attn_bwd_stream16_vs_chunk.txt: N = 1568 1 442 vs
// 1 520 us, 1000: 657 vs 711, 592: 409 vs 477, 520: 634 vs 797; 3136: 2 672 vs 2 640).
// Round 6: where a key block fills seven 32-key waves the second phase runs as attn_bwd_dkdv_stream32_kernel (further down; -15 % on that phase).
#ifndef ME_ST_BWD_MINN
#define ME_ST_BWD_MINN 513
#endif
#ifndef ME_ST_BWD_ROWS
#define ME_ST_BWD_ROWS 32                      // keys per compute wave of the streaming dK / dV kernel: 16 (16 waves) or 32 (8 waves)  -- A/B arm
#endif
constexpr int ST_BWD_MINN = ME_ST_BWD_MINN;

// loader waves of the streaming backward kernels: array 0 / 1 of a chunk from (a0, ld0) / (a1, ld1); FL: 128 floats of f0 / f1 per chunk.
// NW waves per workgroup, the last NL of them load (the others past the compute waves keep the barrier count): NL = 4 (two waves per array),
// 2 (one per array) or 1 (one wave fills both arrays -- the 8-wave kernels with seven compute waves).
template <int HD, bool FL, int NW>
__device__ __forceinline__ void st16_bwd_loader(char* smem, char* fring, int wave, int lane, int NL, int total, int NCH, int vid, int G,
                                                int nblk, int H, int N, int hd, const bf16_t* a0, int64_t ld0, int64_t hoff0,
                                                const bf16_t* a1, int64_t ld1, int64_t hoff1, const float* f0, const float* f1) {
    typedef RCfg<HD> R;
    constexpr int arr_bytes = ST_KC * R::RB;
    constexpr int slot_bytes = 2 * arr_bytes;
    constexpr int NPC = ST_KC / R::RPI;
    static_assert(2 * NPC + 4 <= 63, "vmcnt range");
    if (wave < NW - NL) {                             // spare waves: the barrier count only
        for (int k = 0; k < total; ++k) __builtin_amdgcn_s_barrier();
        return;
    }
    const int lw = wave - (NW - NL);
    const int which0 = NL == 1 ? 0 : (lw & 1), nwhich = NL == 1 ? 2 : 1;
    const int part = lw >> 1, nparts = NL == 4 ? 2 : 1;
    const int rg = (lane * 16) / R::RB, pos = ((lane * 16) % R::RB) / 16;
    const int csrc = pos ^ r16_swz<R::CPR>(rg);
    auto fill = [&](int j) {
        if (j >= total) return;
        const int it = vid + (j / NCH) * G, c = j % NCH;
        const int bh = it / nblk;
        for (int w = 0; w < nwhich; ++w) {
            const int which = which0 + w;
            const bf16_t* const arr = which ? a1 : a0;
            const int64_t ldw = which ? ld1 : ld0, hoff = which ? hoff1 : hoff0;
            const int dma_voff = (csrc * 8 < hd) ? rg * (int)ldw * 2 + csrc * 16 : 0x7f000000;
            const int dma_gstep = R::RPI * (int)ldw * 2;
            const int rec_bytes = (int)(((int64_t)(N - 1) * ldw + hd) * 2);
            const bf16_t* base = arr + (int64_t)(bh / H) * N * ldw + (bh % H) * hd + hoff;
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base), 0, rec_bytes, 0x00020000);
            char* dst = smem + (j % ST_RING) * slot_bytes + which * arr_bytes;
            const int voff = dma_voff + c * NPC * dma_gstep;
            if (nparts == 1) {
#pragma unroll
                for (int i = 0; i < NPC; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_dma_t*)(dst + i * 1024), 16, voff + i * dma_gstep, 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < NPC / 2; ++i) {
                    const int p = 2 * i + part;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_dma_t*)(dst + p * 1024), 16, voff + p * dma_gstep, 0, 0, 0);
                }
            }
            if (FL) {
                // 128 floats of this array's statistics: two dword pieces of 64 (one each when two loaders share the array)
                const float* const farr = which ? f1 : f0;
                const __amdgpu_buffer_rsrc_t rf =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(farr + (int64_t)bh * N), 0, N * 4, 0x00020000);
                char* fdst = fring + ((j % ST_RING) * 2 + which) * (ST_KC * 4);
                const int fvoff = (c * ST_KC + lane) * 4;
                if (nparts == 1) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rf, (lds_dma_t*)fdst, 4, fvoff, 0, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rf, (lds_dma_t*)(fdst + 256), 4, fvoff + 256, 0, 0, 0);
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rf, (lds_dma_t*)(fdst + 256 * part), 4, fvoff + 256 * part, 0, 0, 0);
                }
            }
        }
    };
    fill(0);
    fill(1);
    for (int k = 0; k < total; ++k) {
        // chunk k has landed once at most chunk k + 1's pieces (of this loader) are outstanding (in-order retirement)
        if (k + 1 >= total) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (NL == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPC + (FL ? 4 : 0)) : "memory");
        else if (NL == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPC + (FL ? 2 : 0)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPC / 2 + (FL ? 1 : 0)) : "memory");
        __builtin_amdgcn_s_barrier();                 // chunk k ready; everybody is done with chunk k - 1
        fill(k + 2);                                  // ... whose slot takes chunk k + 2
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int HD>
__global__ __launch_bounds__(R16_THREADS) void attn_bwd_dq_stream16_kernel(const bf16_t* __restrict__ qkv, int64_t ld,
                                                                           const bf16_t* __restrict__ out, int64_t ldo,
                                                                           const bf16_t* __restrict__ dout, int64_t lddo,
                                                                           const float* __restrict__ lse, float* __restrict__ delta,
                                                                           bf16_t* __restrict__ dqkv, int64_t lddq, int N, int H, int hd,
                                                                           float scale, int QB, int nqb, int items) {
    typedef Cfg<bf16_t, HD> C;
    typedef RCfg<HD> R;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int arr_bytes = ST_KC * R::RB;
    constexpr int slot_bytes = 2 * arr_bytes;
    constexpr int NKS = HD / 32, NDT = HD / 16;
    constexpr int SCR = 16 * C::RROW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int Cdim = H * hd;
    const int G = (int)gridDim.x;
    const int NCH = (N + ST_KC - 1) / ST_KC;
    const int vid = (G & 7) == 0 ? ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int my_items = vid < items ? (items - vid + G - 1) / G : 0;
    if (16 * wave >= QB) {
        st16_bwd_loader<HD, false, 16>(smem, nullptr, wave, lane, QB <= 12 * 16 ? 4 : 2, my_items * NCH, NCH, vid, G, nqb, H, N, hd, qkv, ld, Cdim, qkv, ld,
                                   2 * Cdim, nullptr, nullptr);
        return;
    }
    char* scr = smem + ST_RING * slot_bytes + wave * SCR;
    const float sl = scale * LOG2E;
    int koff[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) koff[ks] = l15 * R::RB + 16 * ((4 * ks + g) ^ r16_swz<R::CPR>(l15));
    int troff[NDT];
    {
        const int rr = 4 * g + (l15 >> 2);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
            troff[dt] = rr * R::RB + 16 * ((2 * dt + ((l15 & 3) >> 1)) ^ r16_swz<R::CPR>(rr)) + 8 * (l15 & 1);
    }
    auto rowread = [&](const char* arr, int t, bf16x8 (&dst)[NKS]) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) dst[ks] = *reinterpret_cast<const bf16x8*>(arr + 16 * t * R::RB + koff[ks]);
    };
    auto trread = [&](const char* arr, int kk, int dt) {
        union { bf16x4 q4[2]; bf16x8 v; } a;
#pragma unroll
        for (int r = 0; r < 2; ++r)
            a.q4[r] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)((uint32_t)(uintptr_t)arr + (32 * kk + 16 * r) * R::RB + troff[dt]));
        return a.v;
    };
    const f32x4 zero4f = {0.f, 0.f, 0.f, 0.f};
    // the wave's own rows (Q, dO, O, lse): the next item's are fetched under this item's last chunk
    u32x4 qn[NKS], don[NKS], on[NKS];
    float lse_n = 0.f;
    auto own_issue = [&](int it) {
        const int bh = it / nqb, qb = it % nqb;
        const int q = qb * QB + 16 * wave + l15;
        const int qc = q < N ? q : N - 1;
        const int b = bh / H, head = bh % H;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d = 32 * ks + 8 * g, dc = d < hd ? d : 0;
            qn[ks] = *reinterpret_cast<const u32x4*>(qkv + ((int64_t)b * N + qc) * ld + head * hd + dc);
            don[ks] = *reinterpret_cast<const u32x4*>(dout + ((int64_t)b * N + qc) * lddo + head * hd + dc);
            on[ks] = *reinterpret_cast<const u32x4*>(out + ((int64_t)b * N + qc) * ldo + head * hd + dc);
        }
        lse_n = lse[(int64_t)bh * N + qc];
    };
    if (my_items > 0) own_issue(vid);
    int j = 0;
    for (int ii = 0; ii < my_items; ++ii) {
        const int it = vid + ii * G;
        const int bh = it / nqb, qb = it % nqb;
        const int b = bh / H, head = bh % H;
        const int q0 = qb * QB + 16 * wave;
        const int q = q0 + l15;
        const bool row_ok = q < N;
        bf16x8 qf[NKS], dof[NKS];
        float del = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const bool ok = 32 * ks + 8 * g < hd;
            const u32x4 qv = ok ? qn[ks] : zero4(), dv4 = ok ? don[ks] : zero4(), ov = ok ? on[ks] : zero4();
            qf[ks] = *reinterpret_cast<const bf16x8*>(&qv);
            dof[ks] = *reinterpret_cast<const bf16x8*>(&dv4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                del += __uint_as_float(ov[e] << 16) * __uint_as_float(dv4[e] << 16) +
                       __uint_as_float(ov[e] & 0xffff0000u) * __uint_as_float(dv4[e] & 0xffff0000u);
        }
        del += __shfl_xor(del, 16, 64);
        del += __shfl_xor(del, 32, 64);
        const float lse2 = row_ok ? lse_n * LOG2E : INFINITY;      // +inf on padded queries -> P = 0
        if (g == 0 && row_ok) delta[(int64_t)bh * N + q] = del;
        f32x4 dq[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) dq[dt] = zero4f;
        auto chunk = [&](int c) {
            const char* Kb = smem + (j % ST_RING) * slot_bytes;
            const char* Vb = Kb + arr_bytes;
#pragma unroll
            for (int kk = 0; kk < ST_KC / 32; ++kk) {
                bf16x8 ka[2][NKS], va[2][NKS];
                rowread(Kb, 2 * kk, ka[0]); rowread(Kb, 2 * kk + 1, ka[1]);
                rowread(Vb, 2 * kk, va[0]); rowread(Vb, 2 * kk + 1, va[1]);
                f32x4 s[2], dp[2];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) { s[tt] = mma16(ka[tt][0], qf[0], zero4f); dp[tt] = mma16(va[tt][0], dof[0], zero4f); }
#pragma unroll
                for (int ks = 1; ks < NKS; ++ks)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) { s[tt] = mma16(ka[tt][ks], qf[ks], s[tt]); dp[tt] = mma16(va[tt][ks], dof[ks], dp[tt]); }
                bf16x8 kt[NDT];
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) kt[dt] = trread(Kb, kk, dt);
                bf16x8 dsb;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pe = r16_p(s[tt][e] * sl - lse2);
                        dsb[4 * tt + e] = (bf16_t)(pe * (dp[tt][e] - del));      // dS^T / scale (scale applied to dQ once)
                    }
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) dq[dt] = mma16(kt[dt], dsb, dq[dt]);
            }
        };
        for (int c = 0; c + 1 < NCH; ++c, ++j) {
            __syncthreads();                          // chunk j has landed (the loaders waited for it)
            chunk(c);
        }
        __syncthreads();
        own_issue(ii + 1 < my_items ? it + G : it);   // (the last chunk is peeled: the next item's rows are live only under it)
        chunk(NCH - 1);
        ++j;
        r16_store_rows<HD>(scr, dq, scale, dqkv + ((int64_t)b * N + q0) * lddq + head * hd, lddq, N - q0, hd, lane);
    }
}

template <int HD>
__global__ __launch_bounds__(R16_THREADS) void attn_bwd_dkdv_stream16_kernel(const bf16_t* __restrict__ qkv, int64_t ld,
                                                                             const bf16_t* __restrict__ dout, int64_t lddo,
                                                                             const float* __restrict__ lse, const float* __restrict__ delta,
                                                                             bf16_t* __restrict__ dqkv, int64_t lddq, int N, int H, int hd,
                                                                             float scale, int KB, int nkb, int items) {
    typedef Cfg<bf16_t, HD> C;
    typedef RCfg<HD> R;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int arr_bytes = ST_KC * R::RB;
    constexpr int slot_bytes = 2 * arr_bytes;
    constexpr int NKS = HD / 32, NDT = HD / 16;
    constexpr int SCR = 16 * C::RROW;
    constexpr int FRING = ST_RING * 2 * ST_KC * 4;    // {lse, delta} of a chunk, per ring slot
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int Cdim = H * hd;
    const int G = (int)gridDim.x;
    const int NCH = (N + ST_KC - 1) / ST_KC;
    const int vid = (G & 7) == 0 ? ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int my_items = vid < items ? (items - vid + G - 1) / G : 0;
    char* const fring = smem + ST_RING * slot_bytes;
    if (16 * wave >= KB) {
        st16_bwd_loader<HD, true, 16>(smem, fring, wave, lane, KB <= 12 * 16 ? 4 : 2, my_items * NCH, NCH, vid, G, nkb, H, N, hd, qkv, ld, 0, dout, lddo, 0, lse,
                                  delta);
        return;
    }
    char* scr = fring + FRING + wave * SCR;
    const float sl = scale * LOG2E;
    int koff[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) koff[ks] = l15 * R::RB + 16 * ((4 * ks + g) ^ r16_swz<R::CPR>(l15));
    int troff[NDT];
    {
        const int rr = 4 * g + (l15 >> 2);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
            troff[dt] = rr * R::RB + 16 * ((2 * dt + ((l15 & 3) >> 1)) ^ r16_swz<R::CPR>(rr)) + 8 * (l15 & 1);
    }
    auto rowread = [&](const char* arr, int t, bf16x8 (&dst)[NKS]) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) dst[ks] = *reinterpret_cast<const bf16x8*>(arr + 16 * t * R::RB + koff[ks]);
    };
    auto trread = [&](const char* arr, int kk, int dt) {
        union { bf16x4 q4[2]; bf16x8 v; } a;
#pragma unroll
        for (int r = 0; r < 2; ++r)
            a.q4[r] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)((uint32_t)(uintptr_t)arr + (32 * kk + 16 * r) * R::RB + troff[dt]));
        return a.v;
    };
    const f32x4 zero4f = {0.f, 0.f, 0.f, 0.f};
    // the wave's own K / V rows: the next item's are fetched under this item's last chunk
    u32x4 kn[NKS], vn[NKS];
    auto own_issue = [&](int it) {
        const int bh = it / nkb, kb = it % nkb;
        const int k = kb * KB + 16 * wave + l15;
        const bf16_t* kptr = qkv + ((int64_t)(bh / H) * N + (k < N ? k : N - 1)) * ld + (bh % H) * hd + Cdim;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d = 32 * ks + 8 * g, dc = d < hd ? d : 0;
            kn[ks] = *reinterpret_cast<const u32x4*>(kptr + dc);
            vn[ks] = *reinterpret_cast<const u32x4*>(kptr + Cdim + dc);
        }
    };
    if (my_items > 0) own_issue(vid);
    int j = 0;
    for (int ii = 0; ii < my_items; ++ii) {
        const int it = vid + ii * G;
        const int bh = it / nkb, kb = it % nkb;
        const int b = bh / H, head = bh % H;
        const int k0 = kb * KB + 16 * wave;
        bf16x8 kf[NKS], vf[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const bool ok = 32 * ks + 8 * g < hd;
            const u32x4 kv = ok ? kn[ks] : zero4(), vv = ok ? vn[ks] : zero4();
            kf[ks] = *reinterpret_cast<const bf16x8*>(&kv);
            vf[ks] = *reinterpret_cast<const bf16x8*>(&vv);
        }
        f32x4 dk[NDT], dv[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) { dk[dt] = zero4f; dv[dt] = zero4f; }
        auto chunk = [&](int c) {
            const char* Qb = smem + (j % ST_RING) * slot_bytes;
            const char* Db = Qb + arr_bytes;
            const float* lsb = reinterpret_cast<const float*>(fring + (j % ST_RING) * 2 * (ST_KC * 4));
            const float* deb = lsb + ST_KC;
#pragma unroll
            for (int qq = 0; qq < ST_KC / 32; ++qq) {
                f32x4 s[2], dp[2];
                {
                    bf16x8 qa[2][NKS];
                    rowread(Qb, 2 * qq, qa[0]); rowread(Qb, 2 * qq + 1, qa[1]);
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) s[tt] = mma16(qa[tt][0], kf[0], zero4f);
#pragma unroll
                    for (int ks = 1; ks < NKS; ++ks)
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) s[tt] = mma16(qa[tt][ks], kf[ks], s[tt]);
                }
                {
                    bf16x8 da[2][NKS];
                    rowread(Db, 2 * qq, da[0]); rowread(Db, 2 * qq + 1, da[1]);
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) dp[tt] = mma16(da[tt][0], vf[0], zero4f);
#pragma unroll
                    for (int ks = 1; ks < NKS; ++ks)
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) dp[tt] = mma16(da[tt][ks], vf[ks], dp[tt]);
                }
                bf16x8 pb, dsb;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const f32x4 L = *reinterpret_cast<const f32x4*>(lsb + 16 * (2 * qq + tt) + 4 * g);
                    const f32x4 D = *reinterpret_cast<const f32x4*>(deb + 16 * (2 * qq + tt) + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pe = r16_p(s[tt][e] * sl - L[e] * LOG2E);
                        pb[4 * tt + e] = (bf16_t)pe;
                        dsb[4 * tt + e] = (bf16_t)(pe * (dp[tt][e] - D[e]));      // dS / scale (scale applied to dK once)
                    }
                }
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) dv[dt] = mma16(trread(Db, qq, dt), pb, dv[dt]);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) dk[dt] = mma16(trread(Qb, qq, dt), dsb, dk[dt]);
            }
        };
        for (int c = 0; c + 1 < NCH; ++c, ++j) {
            __syncthreads();
            chunk(c);
        }
        __syncthreads();
        own_issue(ii + 1 < my_items ? it + G : it);
        chunk(NCH - 1);
        ++j;
        bf16_t* grow0 = dqkv + ((int64_t)b * N + k0) * lddq + head * hd;
        r16_store_rows<HD>(scr, dk, scale, grow0 + Cdim, lddq, N - k0, hd, lane);
        r16_store_rows<HD>(scr, dv, 1.0f, grow0 + 2 * Cdim, lddq, N - k0, hd, lane);
    }
}

// -----------------------------------------------------------------------------------------------------
// The dK / dV kernel with 32 KEYS PER WAVE (round 6): 8 waves of up to 256 registers instead of 16 of 128 -- seven compute waves own 2 x 16
// keys each (224 keys per workgroup, as before), the eighth loads both arrays.  Every fragment a wave reads from the streamed {Q, dO} chunk
// (row form for S / dP, transposed form for the two gradient products) now feeds TWO 16-key tiles: half the LDS bytes per MFMA -- the
// 16-row kernel reads 16 KB per 20 MFMAs and keeps the LDS array busy 45 % of the time with the matrix pipe at 35 %
// (profiles/r04_attn_pmc_stream.txt) -- and twice the independent MFMA chains per wave, against half the waves issuing softmax arithmetic.
// Same box, profiles/r06_attn_bwd_stream32.txt: N = 1568 (config 5) 852 -> 763 us, 592 (config 4) 443 -> 411 us.  Taken when the key
// blocks fill seven waves (192 < keys per block <= 224); six compute waves on four SIMDs run as slowly as eight (N = 520: +12 %).
// The dQ kernel in the same form (8 waves: 575 vs 562 us at N = 1568; 12 waves of 137 registers, 320-query blocks: 547 vs 538 us; 12 waves with
// the steps staged like here: 573 vs 567 us) does
// not gain -- 12 MFMAs per 12 KB there, and it is not the LDS that paces it -- and stays on 16-row waves.
constexpr int ST32_THREADS = 512;
#ifndef ME_ST32_PIPE
#define ME_ST32_PIPE 2                          // (A/B arms: 1 = the same source order without the scheduling barriers)
#endif

template <int HD>
__global__ __launch_bounds__(ST32_THREADS) void attn_bwd_dkdv_stream32_kernel(const bf16_t* __restrict__ qkv, int64_t ld,
                                                                              const bf16_t* __restrict__ dout, int64_t lddo,
                                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                                              bf16_t* __restrict__ dqkv, int64_t lddq, int N, int H, int hd,
                                                                              float scale, int KB, int nkb, int items) {
    typedef Cfg<bf16_t, HD> C;
    typedef RCfg<HD> R;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int arr_bytes = ST_KC * R::RB;
    constexpr int slot_bytes = 2 * arr_bytes;
    constexpr int NKS = HD / 32, NDT = HD / 16;
    constexpr int SCR = 16 * C::RROW;
    constexpr int FRING = ST_RING * 2 * ST_KC * 4;    // {lse, delta} of a chunk, per ring slot
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int Cdim = H * hd;
    const int G = (int)gridDim.x;
    const int NCH = (N + ST_KC - 1) / ST_KC;
    const int vid = (G & 7) == 0 ? ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int my_items = vid < items ? (items - vid + G - 1) / G : 0;
    char* const fring = smem + ST_RING * slot_bytes;
    const int ncw = KB >> 5;
    if (wave >= ncw) {
        st16_bwd_loader<HD, true, 8>(smem, fring, wave, lane, 8 - ncw >= 4 ? 4 : 8 - ncw >= 2 ? 2 : 1, my_items * NCH, NCH, vid, G, nkb, H, N, hd, qkv,
                                     ld, 0, dout, lddo, 0, lse, delta);
        return;
    }
    char* scr = fring + FRING + wave * SCR;
    const float sl = scale * LOG2E;
    int koff[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) koff[ks] = l15 * R::RB + 16 * ((4 * ks + g) ^ r16_swz<R::CPR>(l15));
    int troff[NDT];
    {
        const int rr = 4 * g + (l15 >> 2);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
            troff[dt] = rr * R::RB + 16 * ((2 * dt + ((l15 & 3) >> 1)) ^ r16_swz<R::CPR>(rr)) + 8 * (l15 & 1);
    }
    auto rowread = [&](const char* arr, int t, bf16x8 (&dst)[NKS]) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) dst[ks] = *reinterpret_cast<const bf16x8*>(arr + 16 * t * R::RB + koff[ks]);
    };
    auto trread = [&](const char* arr, int kk, int dt) {
        union { bf16x4 q4[2]; bf16x8 v; } a;
#pragma unroll
        for (int r = 0; r < 2; ++r)
            a.q4[r] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)((uint32_t)(uintptr_t)arr + (32 * kk + 16 * r) * R::RB + troff[dt]));
        return a.v;
    };
    const f32x4 zero4f = {0.f, 0.f, 0.f, 0.f};
    // the wave's own K / V rows (two 16-key tiles): the next item's are fetched under this item's last chunk
    u32x4 kn[2][NKS], vn[2][NKS];
    auto own_issue = [&](int it) {
        const int bh = it / nkb, kb = it % nkb;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int k = kb * KB + 32 * wave + 16 * r + l15;
            const bf16_t* kptr = qkv + ((int64_t)(bh / H) * N + (k < N ? k : N - 1)) * ld + (bh % H) * hd + Cdim;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int d = 32 * ks + 8 * g, dc = d < hd ? d : 0;
                kn[r][ks] = *reinterpret_cast<const u32x4*>(kptr + dc);
                vn[r][ks] = *reinterpret_cast<const u32x4*>(kptr + Cdim + dc);
            }
        }
    };
    if (my_items > 0) own_issue(vid);
    int j = 0;
    for (int ii = 0; ii < my_items; ++ii) {
        const int it = vid + ii * G;
        const int bh = it / nkb, kb = it % nkb;
        const int b = bh / H, head = bh % H;
        const int k0 = kb * KB + 32 * wave;
        bf16x8 kf[2][NKS], vf[2][NKS];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const bool ok = 32 * ks + 8 * g < hd;
                const u32x4 kv = ok ? kn[r][ks] : zero4(), vv = ok ? vn[r][ks] : zero4();
                kf[r][ks] = *reinterpret_cast<const bf16x8*>(&kv);
                vf[r][ks] = *reinterpret_cast<const bf16x8*>(&vv);
            }
        f32x4 dk[2][NDT], dv[2][NDT];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) { dk[r][dt] = zero4f; dv[r][dt] = zero4f; }
        // one chunk = four 32-query steps, software-pipelined in program order: the S / dP products of step qq + 1 are issued BEFORE the
        // softmax arithmetic of step qq (which then runs under them), the gradient products of step qq behind it
        auto chunk = [&](int c) {
            const char* Qb = smem + (j % ST_RING) * slot_bytes;
            const char* Db = Qb + arr_bytes;
            const float* lsb = reinterpret_cast<const float*>(fring + (j % ST_RING) * 2 * (ST_KC * 4));
            const float* deb = lsb + ST_KC;
            auto sdp = [&](int qq, f32x4 (&s)[2][2], f32x4 (&dp)[2][2]) {      // [query tile of the pair][key tile]
                {
                    bf16x8 qa[2][NKS];
                    rowread(Qb, 2 * qq, qa[0]); rowread(Qb, 2 * qq + 1, qa[1]);
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int r = 0; r < 2; ++r) s[tt][r] = mma16(qa[tt][0], kf[r][0], zero4f);
#pragma unroll
                    for (int ks = 1; ks < NKS; ++ks)
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                            for (int r = 0; r < 2; ++r) s[tt][r] = mma16(qa[tt][ks], kf[r][ks], s[tt][r]);
                }
                {
                    bf16x8 da[2][NKS];
                    rowread(Db, 2 * qq, da[0]); rowread(Db, 2 * qq + 1, da[1]);
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int r = 0; r < 2; ++r) dp[tt][r] = mma16(da[tt][0], vf[r][0], zero4f);
#pragma unroll
                    for (int ks = 1; ks < NKS; ++ks)
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                            for (int r = 0; r < 2; ++r) dp[tt][r] = mma16(da[tt][ks], vf[r][ks], dp[tt][r]);
                }
            };
            f32x4 s[2][2], dp[2][2];
            sdp(0, s, dp);
#pragma unroll
            for (int qq = 0; qq < ST_KC / 32; ++qq) {
                // stage order held by scheduling barriers (the compiler otherwise sinks the next step's products behind this step's and
                // waits on every transposed read pair in turn): next S / dP -> this step's transposed dO fragments + statistics ->
                // softmax arithmetic (under both) -> transposed Q fragments -> dV products (under which those land) -> dK products
                f32x4 s2[2][2], dp2[2][2];
                if (qq + 1 < ST_KC / 32) sdp(qq + 1, s2, dp2);
                if (ME_ST32_PIPE == 2) __builtin_amdgcn_sched_barrier(0);
                bf16x8 td[NDT], tq[NDT];
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) td[dt] = trread(Db, qq, dt);
                f32x4 L[2], D[2];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    L[tt] = *reinterpret_cast<const f32x4*>(lsb + 16 * (2 * qq + tt) + 4 * g);
                    D[tt] = *reinterpret_cast<const f32x4*>(deb + 16 * (2 * qq + tt) + 4 * g);
                }
                if (ME_ST32_PIPE == 2) __builtin_amdgcn_sched_barrier(0);
                bf16x8 pb[2], dsb[2];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const f32x4 L2 = L[tt] * LOG2E;
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        f32x4 pe, ds;
                        r16_pds4(s[tt][r], dp[tt][r], sl, L2, D[tt], pe, ds);      // dS / scale (scale applied to dK once)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { pb[r][4 * tt + e] = (bf16_t)pe[e]; dsb[r][4 * tt + e] = (bf16_t)ds[e]; }
                    }
                }
                if (ME_ST32_PIPE == 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) tq[dt] = trread(Qb, qq, dt);      // (lands under the dV products)
                if (ME_ST32_PIPE == 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                    for (int r = 0; r < 2; ++r) dv[r][dt] = mma16(td[dt], pb[r], dv[r][dt]);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                    for (int r = 0; r < 2; ++r) dk[r][dt] = mma16(tq[dt], dsb[r], dk[r][dt]);
                if (ME_ST32_PIPE == 2) __builtin_amdgcn_sched_barrier(0);
                if (qq + 1 < ST_KC / 32) {
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int r = 0; r < 2; ++r) { s[tt][r] = s2[tt][r]; dp[tt][r] = dp2[tt][r]; }
                }
            }
        };
        for (int c = 0; c + 1 < NCH; ++c, ++j) {
            __syncthreads();
            chunk(c);
        }
        __syncthreads();
        own_issue(ii + 1 < my_items ? it + G : it);
        chunk(NCH - 1);
        ++j;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            bf16_t* grow0 = dqkv + ((int64_t)b * N + k0 + 16 * r) * lddq + head * hd;
            r16_store_rows<HD>(scr, dk[r], scale, grow0 + Cdim, lddq, N - k0 - 16 * r, hd, lane);
            r16_store_rows<HD>(scr, dv[r], 1.0f, grow0 + 2 * Cdim, lddq, N - k0 - 16 * r, hd, lane);
        }
    }
}

// =====================================================================================================
// mid-size sequences (256 < N <= 512: audio spectrogram tokens, 512 x 512 segmentation crops at /32, BASELINE config 3).
// The tiled kernels re-stream Q/dO (or K/V) once per 128-row block of the other side -- four times at N = 512 -- and pay
// two barriers per 64-row tile.  Here one 8-wave workgroup still owns a (batch, head), but LDS holds only TWO arrays at a
// time (2 x 512 rows x 144 B = 147 KB): K/V while a wave walks its (up to two) 32-query groups, then -- backward only --
// the same space is refilled with Q/dO for the pass in which a wave owns its 32-key groups.  Every array is fetched once
// into LDS plus once as row fragments; no barrier inside the key / query loops.  (The same scheme with 4-wave workgroups,
// two per CU, was measured for N <= 256 against the all-four-arrays-resident kernels above: 15-40 % slower there.)
// =====================================================================================================
constexpr int MD_MAXN = 512;

template <int HD, int NTHR, int MAXN>
__global__ __launch_bounds__(NTHR) void attn_fwd_mid_kernel(const bf16_t* __restrict__ qkv, int64_t ld,
                                                                  bf16_t* __restrict__ out, int64_t ldo,
                                                                  float* __restrict__ lse, int N, int H, int hd, float scale) {
    typedef Cfg<bf16_t, HD> C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NS = (N + 31) >> 5, NR = NS * 32;
    char* Ks = smem;
    char* Vs = smem + NR * C::RROW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int head = blockIdx.x, b = blockIdx.y;
    const int Cdim = H * hd;
    const bf16_t* qptr = qkv + (int64_t)b * N * ld + head * hd;
    {
        SmallStage<HD, 2, MAXN, NTHR> st;
        const bf16_t* const bases[2] = {qptr + Cdim, qptr + 2 * Cdim};
        const int64_t ldv[2] = {ld, ld};
        st.load(bases, ldv, N, hd, tid);
        char* const tiles[2] = {Ks, Vs};
        st.store(tiles, NR, tid, N, hd);
    }
    __syncthreads();
    const float sl = scale * LOG2E;
    for (int sub = wave; sub < NS; sub += NTHR / 64) {      // wave-uniform
        const int q = 32 * sub + l31;
        const int qrow = (q < N) ? q : N - 1;
        bf16x8 qf[C::NKK];
#pragma unroll
        for (int kk = 0; kk < C::NKK; ++kk) {
            const int d = (2 * kk + h) * 8;
            const bool ok = d < hd;
            u32x4 raw = *reinterpret_cast<const u32x4*>(qptr + (int64_t)qrow * ld + (ok ? d : 0));
            raw = ok ? raw : zero4();
            qf[kk] = *reinterpret_cast<bf16x8*>(&raw);
        }
        f32x16 o[C::NDB];
#pragma unroll
        for (int db = 0; db < C::NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;
        int t = 0;
        for (; 32 * (t + 2) <= N; t += 2)
            fwd_small_step<HD, 2, false>(Ks + 32 * t * C::RROW, Vs + 32 * t * C::RROW, 32 * t, N, qf, sl, m_run, l_run, o, lane);
        if (t + 2 <= NS)
            fwd_small_step<HD, 2, true>(Ks + 32 * t * C::RROW, Vs + 32 * t * C::RROW, 32 * t, N, qf, sl, m_run, l_run, o, lane);
        else if (t < NS)
            fwd_small_step<HD, 1, true>(Ks + 32 * t * C::RROW, Vs + 32 * t * C::RROW, 32 * t, N, qf, sl, m_run, l_run, o, lane);
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
        if (q < N) {
            bf16_t* orow = out + ((int64_t)b * N + q) * ldo + head * hd;
#pragma unroll
            for (int db = 0; db < C::NDB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = 32 * db + 8 * g + 4 * h;
                    if (d < hd)
                        store_quad<bf16_t>(orow + d, f32x4{o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv,
                                                           o[db][4 * g + 3] * inv});
                }
            if (lse && h == 0) lse[((int64_t)b * H + head) * N + q] = (m_run + __builtin_amdgcn_logf(l_tot)) * LN2;
        }
    }
}

template <int HD, int NTHR, int MAXN>
__global__ __launch_bounds__(NTHR) void attn_bwd_mid_kernel(const bf16_t* __restrict__ qkv, int64_t ld,
                                                                  const bf16_t* __restrict__ out, int64_t ldo,
                                                                  const bf16_t* __restrict__ dout, int64_t lddo,
                                                                  const float* __restrict__ lse, float* __restrict__ delta,
                                                                  bf16_t* __restrict__ dqkv, int64_t lddq, int N, int H,
                                                                  int hd, float scale) {
    typedef Cfg<bf16_t, HD> C;
    constexpr int NWAVE = NTHR / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NS = (N + 31) >> 5, NR = NS * 32;
    char* T0 = smem;                        // K, later Q
    char* T1 = smem + NR * C::RROW;         // V, later dO
    float* lse_s = reinterpret_cast<float*>(T1 + NR * C::RROW);   // [MD_MAXN]  lse * log2(e), +inf on padded rows
    float* del_s = lse_s + MAXN;                                  // [MAXN]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int head = blockIdx.x, b = blockIdx.y;
    const int Cdim = H * hd;
    const bf16_t* qptr = qkv + (int64_t)b * N * ld + head * hd;
    const bf16_t* doptr = dout + (int64_t)b * N * lddo + head * hd;
    const bf16_t* optr = out + (int64_t)b * N * ldo + head * hd;
    const int64_t bh = ((int64_t)b * H + head) * N;
    const float sl = scale * LOG2E;
    {
        SmallStage<HD, 2, MAXN, NTHR> st;
        const bf16_t* const bases[2] = {qptr + Cdim, qptr + 2 * Cdim};
        const int64_t ldv[2] = {ld, ld};
        st.load(bases, ldv, N, hd, tid);
        char* const tiles[2] = {T0, T1};
        st.store(tiles, NR, tid, N, hd);
    }
    for (int i = tid; i < MAXN; i += NTHR) lse_s[i] = i < N ? lse[bh + i] * LOG2E : INFINITY;
    __syncthreads();

    // ---- pass A (K / V resident): delta and dQ, a wave walks its 32-query groups; row fragments come straight from HBM
    for (int sub = wave; sub < NS; sub += NWAVE) {
        const int row = 32 * sub + l31;
        const bool row_ok = row < N;
        const int rowc = row_ok ? row : N - 1;
        bf16x8 qf[C::NKK], dof[C::NKK];
        float del = 0.f;
#pragma unroll
        for (int kk = 0; kk < C::NKK; ++kk) {
            const int d = (2 * kk + h) * 8;
            const bool ok = d < hd;
            const int dc = ok ? d : 0;
            u32x4 rq = *reinterpret_cast<const u32x4*>(qptr + (int64_t)rowc * ld + dc);
            u32x4 rd = *reinterpret_cast<const u32x4*>(doptr + (int64_t)rowc * lddo + dc);
            u32x4 ro = *reinterpret_cast<const u32x4*>(optr + (int64_t)rowc * ldo + dc);
            const bool keep = ok && row_ok;      // padded queries: zero fragments (P = 0 through lse = +inf as well)
            rq = keep ? rq : zero4(); rd = keep ? rd : zero4(); ro = keep ? ro : zero4();
            qf[kk] = *reinterpret_cast<bf16x8*>(&rq);
            dof[kk] = *reinterpret_cast<bf16x8*>(&rd);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                del += __uint_as_float(ro[e] << 16) * __uint_as_float(rd[e] << 16) +
                       __uint_as_float(ro[e] & 0xffff0000u) * __uint_as_float(rd[e] & 0xffff0000u);
        }
        del += __shfl_xor(del, 32, 64);
        if (h == 0) {
            del_s[row] = del;
            if (row_ok && delta) delta[bh + row] = del;
        }
        const float lse2 = lse_s[row];
        f32x16 dq[C::NDB];
#pragma unroll
        for (int db = 0; db < C::NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
        int t = 0;
        for (; 32 * (t + 2) <= N; t += 2)
            bwd_small_q_step<HD, 2, false>(T0 + 32 * t * C::RROW, T1 + 32 * t * C::RROW, 32 * t, N, qf, dof, sl, lse2, del, dq, lane);
        if (t + 2 <= NS)
            bwd_small_q_step<HD, 2, true>(T0 + 32 * t * C::RROW, T1 + 32 * t * C::RROW, 32 * t, N, qf, dof, sl, lse2, del, dq, lane);
        else if (t < NS)
            bwd_small_q_step<HD, 1, true>(T0 + 32 * t * C::RROW, T1 + 32 * t * C::RROW, 32 * t, N, qf, dof, sl, lse2, del, dq, lane);
        if (row_ok) {
            bf16_t* dqrow = dqkv + ((int64_t)b * N + row) * lddq + head * hd;
#pragma unroll
            for (int db = 0; db < C::NDB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = 32 * db + 8 * g + 4 * h;
                    if (d < hd)
                        store_quad<bf16_t>(dqrow + d, f32x4{dq[db][4 * g] * scale, dq[db][4 * g + 1] * scale,
                                                            dq[db][4 * g + 2] * scale, dq[db][4 * g + 3] * scale});
                }
        }
    }
    // ---- hand-over: lift the K / V row fragments of my key groups, then refill the two tiles with Q / dO
    constexpr int MAXG = MAXN / 32 / NWAVE;      // 2
    bf16x8 kf[MAXG][C::NKK], vf[MAXG][C::NKK];
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
        int row = 32 * (wave + NWAVE * g) + l31;
        row = row < NR ? row : NR - 1;
#pragma unroll
        for (int kk = 0; kk < C::NKK; ++kk) {
            kf[g][kk] = rtile_chunk<bf16_t, HD>(T0, row, 2 * kk + h);
            vf[g][kk] = rtile_chunk<bf16_t, HD>(T1, row, 2 * kk + h);
        }
    }
    __syncthreads();      // everyone is done with K / V (pass A loops and the lifts above); del_s is complete
    {
        SmallStage<HD, 2, MAXN, NTHR> st;
        const bf16_t* const bases[2] = {qptr, doptr};
        const int64_t ldv[2] = {ld, lddo};
        st.load(bases, ldv, N, hd, tid);
        char* const tiles[2] = {T0, T1};
        st.store(tiles, NR, tid, N, hd);
    }
    __syncthreads();
    // ---- pass B (Q / dO resident): dK, dV, a wave walks its 32-key groups
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
        const int sub = wave + NWAVE * g;
        if (sub >= NS) break;      // wave-uniform
        const int row = 32 * sub + l31;
        f32x16 dk[C::NDB], dv[C::NDB];
#pragma unroll
        for (int db = 0; db < C::NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
        int t = 0;
        for (; t + 2 <= NS; t += 2)
            bwd_small_kv_step<HD, 2>(T0 + 32 * t * C::RROW, T1 + 32 * t * C::RROW, lse_s + 32 * t, del_s + 32 * t, kf[g], vf[g], sl, dk, dv, lane);
        if (t < NS)
            bwd_small_kv_step<HD, 1>(T0 + 32 * t * C::RROW, T1 + 32 * t * C::RROW, lse_s + 32 * t, del_s + 32 * t, kf[g], vf[g], sl, dk, dv, lane);
        if (row < N) {
            bf16_t* dkrow = dqkv + ((int64_t)b * N + row) * lddq + Cdim + head * hd;
            bf16_t* dvrow = dkrow + Cdim;
#pragma unroll
            for (int db = 0; db < C::NDB; ++db)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int d = 32 * db + 8 * gq + 4 * h;
                    if (d < hd) {
                        store_quad<bf16_t>(dkrow + d, f32x4{dk[db][4 * gq] * scale, dk[db][4 * gq + 1] * scale,
                                                            dk[db][4 * gq + 2] * scale, dk[db][4 * gq + 3] * scale});
                        store_quad<bf16_t>(dvrow + d, f32x4{dv[db][4 * gq], dv[db][4 * gq + 1], dv[db][4 * gq + 2], dv[db][4 * gq + 3]});
                    }
                }
        }
    }
}

// =====================================================================================================
// long sequences (N > 512: video tubelets N = 1568, dense-prediction grids), bf16, head_dim <= 64: the same 8-wave
// building blocks, but the other side of the attention matrix streams through LDS in CHUNKS of 256 rows (74 KB) instead
// of 64-row tiles: two barriers per 256 rows instead of per 64, a workgroup owns 256 rows (so the streamed side is
// re-read N/256 times, not N/128), and inside a chunk the per-wave loops run barrier-free on the step functions above.
//   forward : workgroup = 256 queries, K/V chunks stream (online softmax across chunks)
//   backward: (1) dQ + delta: workgroup = 256 queries, K/V chunks stream; (2) dK/dV: workgroup = 256 keys, Q/dO chunks
//             (+ their lse / delta) stream.  Deterministic, no atomics.
// =====================================================================================================
constexpr int CH_ROWS = 256;

template <int HD>
__global__ __launch_bounds__(SM_THREADS) void attn_fwd_chunk_kernel(const bf16_t* __restrict__ qkv, int64_t ld,
                                                                    bf16_t* __restrict__ out, int64_t ldo,
                                                                    float* __restrict__ lse, int N, int H, int hd, float scale) {
    typedef Cfg<bf16_t, HD> C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + CH_ROWS * C::RROW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int Cdim = H * hd;
    const bf16_t* qptr = qkv + (int64_t)b * N * ld + head * hd;
    const int q = blockIdx.x * CH_ROWS + 32 * wave + l31;
    const bool active = blockIdx.x * CH_ROWS + 32 * wave < N;      // wave-uniform
    const int qrow = (q < N) ? q : N - 1;
    bf16x8 qf[C::NKK];
#pragma unroll
    for (int kk = 0; kk < C::NKK; ++kk) {
        const int d = (2 * kk + h) * 8;
        const bool ok = d < hd;
        u32x4 raw = *reinterpret_cast<const u32x4*>(qptr + (int64_t)qrow * ld + (ok ? d : 0));
        raw = ok ? raw : zero4();
        qf[kk] = *reinterpret_cast<bf16x8*>(&raw);
    }
    f32x16 o[C::NDB];
#pragma unroll
    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sl = scale * LOG2E;
    for (int kv0 = 0; kv0 < N; kv0 += CH_ROWS) {
        const int rows = N - kv0 < CH_ROWS ? N - kv0 : CH_ROWS;
        const int NSc = (rows + 31) >> 5;
        __syncthreads();      // the previous chunk is consumed
        {
            SmallStage<HD, 2> st;
            const bf16_t* const bases[2] = {qptr + Cdim + (int64_t)kv0 * ld, qptr + 2 * Cdim + (int64_t)kv0 * ld};
            const int64_t ldv[2] = {ld, ld};
            st.load(bases, ldv, rows, hd, tid);
            char* const tiles[2] = {Ks, Vs};
            st.store(tiles, NSc * 32, tid, rows, hd);
        }
        __syncthreads();
        if (!active) continue;
        int t = 0;
        for (; 32 * (t + 2) <= rows; t += 2)
            fwd_small_step<HD, 2, false>(Ks + 32 * t * C::RROW, Vs + 32 * t * C::RROW, kv0 + 32 * t, N, qf, sl, m_run, l_run, o, lane);
        if (t + 2 <= NSc)
            fwd_small_step<HD, 2, true>(Ks + 32 * t * C::RROW, Vs + 32 * t * C::RROW, kv0 + 32 * t, N, qf, sl, m_run, l_run, o, lane);
        else if (t < NSc)
            fwd_small_step<HD, 1, true>(Ks + 32 * t * C::RROW, Vs + 32 * t * C::RROW, kv0 + 32 * t, N, qf, sl, m_run, l_run, o, lane);
    }
    if (!active) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q < N) {
        bf16_t* orow = out + ((int64_t)b * N + q) * ldo + head * hd;
#pragma unroll
        for (int db = 0; db < C::NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = 32 * db + 8 * g + 4 * h;
                if (d < hd)
                    store_quad<bf16_t>(orow + d, f32x4{o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv,
                                                       o[db][4 * g + 3] * inv});
            }
        if (lse && h == 0) lse[((int64_t)b * H + head) * N + q] = (m_run + __builtin_amdgcn_logf(l_tot)) * LN2;
    }
}

// backward pass 1: delta and dQ for 256 queries; K/V stream in chunks
template <int HD>
__global__ __launch_bounds__(SM_THREADS) void attn_bwd_dq_chunk_kernel(const bf16_t* __restrict__ qkv, int64_t ld,
                                                                       const bf16_t* __restrict__ out, int64_t ldo,
                                                                       const bf16_t* __restrict__ dout, int64_t lddo,
                                                                       const float* __restrict__ lse, float* __restrict__ delta,
                                                                       bf16_t* __restrict__ dqkv, int64_t lddq, int N, int H,
                                                                       int hd, float scale) {
    typedef Cfg<bf16_t, HD> C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + CH_ROWS * C::RROW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int Cdim = H * hd;
    const bf16_t* qptr = qkv + (int64_t)b * N * ld + head * hd;
    const bf16_t* doptr = dout + (int64_t)b * N * lddo + head * hd;
    const bf16_t* optr = out + (int64_t)b * N * ldo + head * hd;
    const int64_t bh = ((int64_t)b * H + head) * N;
    const int row = blockIdx.x * CH_ROWS + 32 * wave + l31;
    const bool active = blockIdx.x * CH_ROWS + 32 * wave < N;      // wave-uniform
    const bool row_ok = row < N;
    const int rowc = row_ok ? row : N - 1;
    bf16x8 qf[C::NKK], dof[C::NKK];
    float del = 0.f;
#pragma unroll
    for (int kk = 0; kk < C::NKK; ++kk) {
        const int d = (2 * kk + h) * 8;
        const bool ok = d < hd;
        const int dc = ok ? d : 0;
        u32x4 rq = *reinterpret_cast<const u32x4*>(qptr + (int64_t)rowc * ld + dc);
        u32x4 rd = *reinterpret_cast<const u32x4*>(doptr + (int64_t)rowc * lddo + dc);
        u32x4 ro = *reinterpret_cast<const u32x4*>(optr + (int64_t)rowc * ldo + dc);
        const bool keep = ok && row_ok;
        rq = keep ? rq : zero4(); rd = keep ? rd : zero4(); ro = keep ? ro : zero4();
        qf[kk] = *reinterpret_cast<bf16x8*>(&rq);
        dof[kk] = *reinterpret_cast<bf16x8*>(&rd);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            del += __uint_as_float(ro[e] << 16) * __uint_as_float(rd[e] << 16) +
                   __uint_as_float(ro[e] & 0xffff0000u) * __uint_as_float(rd[e] & 0xffff0000u);
    }
    del += __shfl_xor(del, 32, 64);
    if (h == 0 && row_ok) delta[bh + row] = del;
    const float lse2 = row_ok ? lse[bh + row] * LOG2E : INFINITY;      // padded queries: P = 0
    const float sl = scale * LOG2E;
    f32x16 dq[C::NDB];
#pragma unroll
    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
    for (int kv0 = 0; kv0 < N; kv0 += CH_ROWS) {
        const int rows = N - kv0 < CH_ROWS ? N - kv0 : CH_ROWS;
        const int NSc = (rows + 31) >> 5;
        __syncthreads();
        {
            SmallStage<HD, 2> st;
            const bf16_t* const bases[2] = {qptr + Cdim + (int64_t)kv0 * ld, qptr + 2 * Cdim + (int64_t)kv0 * ld};
            const int64_t ldv[2] = {ld, ld};
            st.load(bases, ldv, rows, hd, tid);
            char* const tiles[2] = {Ks, Vs};
            st.store(tiles, NSc * 32, tid, rows, hd);
        }
        __syncthreads();
        if (!active) continue;
        int t = 0;
        for (; 32 * (t + 2) <= rows; t += 2)
            bwd_small_q_step<HD, 2, false>(Ks + 32 * t * C::RROW, Vs + 32 * t * C::RROW, kv0 + 32 * t, N, qf, dof, sl, lse2, del, dq, lane);
        if (t + 2 <= NSc)
            bwd_small_q_step<HD, 2, true>(Ks + 32 * t * C::RROW, Vs + 32 * t * C::RROW, kv0 + 32 * t, N, qf, dof, sl, lse2, del, dq, lane);
        else if (t < NSc)
            bwd_small_q_step<HD, 1, true>(Ks + 32 * t * C::RROW, Vs + 32 * t * C::RROW, kv0 + 32 * t, N, qf, dof, sl, lse2, del, dq, lane);
    }
    if (!active || !row_ok) return;
    bf16_t* dqrow = dqkv + ((int64_t)b * N + row) * lddq + head * hd;
#pragma unroll
    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 32 * db + 8 * g + 4 * h;
            if (d < hd)
                store_quad<bf16_t>(dqrow + d, f32x4{dq[db][4 * g] * scale, dq[db][4 * g + 1] * scale, dq[db][4 * g + 2] * scale,
                                                    dq[db][4 * g + 3] * scale});
        }
}

// backward pass 2: dK and dV for 256 keys; Q / dO (and their lse / delta) stream in chunks
template <int HD>
__global__ __launch_bounds__(SM_THREADS) void attn_bwd_dkdv_chunk_kernel(const bf16_t* __restrict__ qkv, int64_t ld,
                                                                         const bf16_t* __restrict__ dout, int64_t lddo,
                                                                         const float* __restrict__ lse,
                                                                         const float* __restrict__ delta,
                                                                         bf16_t* __restrict__ dqkv, int64_t lddq, int N, int H,
                                                                         int hd, float scale) {
    typedef Cfg<bf16_t, HD> C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem;
    char* dOs = smem + CH_ROWS * C::RROW;
    float* lse_s = reinterpret_cast<float*>(dOs + CH_ROWS * C::RROW);   // [CH_ROWS]
    float* del_s = lse_s + CH_ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int Cdim = H * hd;
    const bf16_t* qptr = qkv + (int64_t)b * N * ld + head * hd;
    const bf16_t* doptr = dout + (int64_t)b * N * lddo + head * hd;
    const int64_t bh = ((int64_t)b * H + head) * N;
    const int row = blockIdx.x * CH_ROWS + 32 * wave + l31;
    const bool active = blockIdx.x * CH_ROWS + 32 * wave < N;      // wave-uniform
    const bool row_ok = row < N;
    const int rowc = row_ok ? row : N - 1;
    bf16x8 kf[C::NKK], vf[C::NKK];
#pragma unroll
    for (int kk = 0; kk < C::NKK; ++kk) {
        const int d = (2 * kk + h) * 8;
        const bool ok = d < hd;
        const int dc = ok ? d : 0;
        u32x4 rk = *reinterpret_cast<const u32x4*>(qptr + Cdim + (int64_t)rowc * ld + dc);
        u32x4 rv = *reinterpret_cast<const u32x4*>(qptr + 2 * Cdim + (int64_t)rowc * ld + dc);
        rk = ok ? rk : zero4(); rv = ok ? rv : zero4();
        kf[kk] = *reinterpret_cast<bf16x8*>(&rk);
        vf[kk] = *reinterpret_cast<bf16x8*>(&rv);
    }
    const float sl = scale * LOG2E;
    f32x16 dk[C::NDB], dv[C::NDB];
#pragma unroll
    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
    for (int q0 = 0; q0 < N; q0 += CH_ROWS) {
        const int rows = N - q0 < CH_ROWS ? N - q0 : CH_ROWS;
        const int NSc = (rows + 31) >> 5;
        __syncthreads();
        {
            SmallStage<HD, 2> st;
            const bf16_t* const bases[2] = {qptr + (int64_t)q0 * ld, doptr + (int64_t)q0 * lddo};
            const int64_t ldv[2] = {ld, lddo};
            st.load(bases, ldv, rows, hd, tid);
            char* const tiles[2] = {Qs, dOs};
            st.store(tiles, NSc * 32, tid, rows, hd);
        }
        if (tid < CH_ROWS) {
            const bool ok = tid < rows;
            lse_s[tid] = ok ? lse[bh + q0 + tid] * LOG2E : INFINITY;      // padded queries: P = 0
            del_s[tid] = ok ? delta[bh + q0 + tid] : 0.f;
        }
        __syncthreads();
        if (!active) continue;
        int t = 0;
        for (; t + 2 <= NSc; t += 2)
            bwd_small_kv_step<HD, 2>(Qs + 32 * t * C::RROW, dOs + 32 * t * C::RROW, lse_s + 32 * t, del_s + 32 * t, kf, vf, sl, dk, dv, lane);
        if (t < NSc)
            bwd_small_kv_step<HD, 1>(Qs + 32 * t * C::RROW, dOs + 32 * t * C::RROW, lse_s + 32 * t, del_s + 32 * t, kf, vf, sl, dk, dv, lane);
    }
    if (!active || !row_ok) return;
    bf16_t* dkrow = dqkv + ((int64_t)b * N + row) * lddq + Cdim + head * hd;
    bf16_t* dvrow = dkrow + Cdim;
#pragma unroll
    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 32 * db + 8 * g + 4 * h;
            if (d < hd) {
                store_quad<bf16_t>(dkrow + d, f32x4{dk[db][4 * g] * scale, dk[db][4 * g + 1] * scale, dk[db][4 * g + 2] * scale,
                                                    dk[db][4 * g + 3] * scale});
                store_quad<bf16_t>(dvrow + d, f32x4{dv[db][4 * g], dv[db][4 * g + 1], dv[db][4 * g + 2], dv[db][4 * g + 3]});
            }
        }
}

template <typename K> void set_smem(K kernel, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

int device_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    }
    return n;
}
constexpr size_t LDS_PER_CU = 160 * 1024;

template <int HD>
int launch_fwd_small(const void* qkv, int64_t ld, void* out, int64_t ldo, float* lse, int B, int N, int H, int hd, float scale,
                     hipStream_t stream) {
    typedef Cfg<bf16_t, HD> C;
    const size_t smem = (size_t)(2 * ((N + 31) / 32 * 32) + SM_THREADS / 2) * C::RROW;      // {K, V} + per-wave [32][HD] scratch
    static OncePerDevice once;
    if (once.need()) { set_smem(attn_fwd_small_kernel<HD>, (size_t)(2 * SM_MAXN + SM_THREADS / 2) * C::RROW); }
    const int64_t items = (int64_t)B * H;
    int per_cu = (int)(LDS_PER_CU / smem);
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    const int64_t slots = (int64_t)device_cus() * per_cu;
    const unsigned grid = (unsigned)(items < slots ? items : slots);
    hipLaunchKernelGGL((attn_fwd_small_kernel<HD>), dim3(grid), dim3(SM_THREADS), smem, stream,
                       reinterpret_cast<const bf16_t*>(qkv), ld, reinterpret_cast<bf16_t*>(out), ldo, lse, N, H, hd, scale,
                       (int)items);
    ME_CHECK_LAUNCH("me_attention_fwd(small)");
    return ME_OK;
}
template <int HD, int NS>
int launch_fwd_ring16_ns(const void* qkv, int64_t ld, void* out, int64_t ldo, float* lse, int B, int N, int H, int hd, float scale,
                         hipStream_t stream) {
    typedef RCfg<HD> R;
    constexpr size_t smem = (size_t)4 * NS * 32 * R::RB + (R16_THREADS / 64) * 16 * Cfg<bf16_t, HD>::RROW;
    static OncePerDevice once;
    if (once.need()) { set_smem(attn_fwd_ring16_kernel<HD, NS>, smem); }
    const int64_t items = (int64_t)B * H;
    const int64_t slots = device_cus();
    const unsigned grid = (unsigned)(items < slots ? items : slots);
    hipLaunchKernelGGL((attn_fwd_ring16_kernel<HD, NS>), dim3(grid), dim3(R16_THREADS), smem, stream,
                       reinterpret_cast<const bf16_t*>(qkv), ld, reinterpret_cast<bf16_t*>(out), ldo, lse, N, H, hd, scale,
                       (int)items);
    ME_CHECK_LAUNCH("me_attention_fwd(ring16)");
    return ME_OK;
}
template <int HD>
int launch_fwd_ring16(const void* qkv, int64_t ld, void* out, int64_t ldo, float* lse, int B, int N, int H, int hd, float scale,
                      hipStream_t stream) {
    switch ((N + 31) / 32) {
        case 3: return launch_fwd_ring16_ns<HD, 3>(qkv, ld, out, ldo, lse, B, N, H, hd, scale, stream);
        case 4: return launch_fwd_ring16_ns<HD, 4>(qkv, ld, out, ldo, lse, B, N, H, hd, scale, stream);
        case 5: return launch_fwd_ring16_ns<HD, 5>(qkv, ld, out, ldo, lse, B, N, H, hd, scale, stream);
        case 6: return launch_fwd_ring16_ns<HD, 6>(qkv, ld, out, ldo, lse, B, N, H, hd, scale, stream);
        default: return launch_fwd_ring16_ns<HD, 7>(qkv, ld, out, ldo, lse, B, N, H, hd, scale, stream);
    }
}
template <int HD>
int launch_bwd_small(const void* qkv, int64_t ld, const void* out, int64_t ldo, const void* dout, int64_t lddo, const float* lse,
                     float* delta, void* dqkv, int64_t lddq, int B, int N, int H, int hd, float scale, hipStream_t stream) {
    typedef Cfg<bf16_t, HD> C;
    const size_t smem = (size_t)4 * ((N + 31) / 32 * 32) * C::RROW + 2 * SM_MAXN * sizeof(float);
    static OncePerDevice once;
    if (once.need()) { set_smem(attn_bwd_small_kernel<HD>, (size_t)4 * SM_MAXN * C::RROW + 2 * SM_MAXN * sizeof(float)); }
    hipLaunchKernelGGL((attn_bwd_small_kernel<HD>), dim3(H, B), dim3(SM_THREADS), smem, stream,
                       reinterpret_cast<const bf16_t*>(qkv), ld, reinterpret_cast<const bf16_t*>(out), ldo,
                       reinterpret_cast<const bf16_t*>(dout), lddo, lse, delta, reinterpret_cast<bf16_t*>(dqkv), lddq, N, H, hd,
                       scale);
    ME_CHECK_LAUNCH("me_attention_bwd(small)");
    return ME_OK;
}

template <int HD>
int launch_fwd_stream16(const void* qkv, int64_t ld, void* out, int64_t ldo, float* lse, int B, int N, int H, int hd, float scale,
                        hipStream_t stream) {
    typedef RCfg<HD> R;
    constexpr size_t smem = (size_t)ST_RING * 2 * ST_KC * R::RB + (R16_THREADS / 64) * 16 * Cfg<bf16_t, HD>::RROW;
    static OncePerDevice once;
    if (once.need()) { set_smem(attn_fwd_stream16_kernel<HD>, smem); }
    // query blocks of (at most) 14 x 16 rows, evened out over the sequence
    const int nqb = (N + 223) / 224;
    const int QB = ((N + nqb - 1) / nqb + 15) / 16 * 16;
    const int64_t items = (int64_t)B * H * nqb;
    const int64_t slots = device_cus();
    const unsigned grid = (unsigned)(items < slots ? items : slots);
    hipLaunchKernelGGL((attn_fwd_stream16_kernel<HD>), dim3(grid), dim3(R16_THREADS), smem, stream,
                       reinterpret_cast<const bf16_t*>(qkv), ld, reinterpret_cast<bf16_t*>(out), ldo, lse, N, H, hd, scale, QB, nqb,
                       (int)items);
    ME_CHECK_LAUNCH("me_attention_fwd(stream16)");
    return ME_OK;
}
template <int HD>
int launch_bwd_stream16(const void* qkv, int64_t ld, const void* out, int64_t ldo, const void* dout, int64_t lddo, const float* lse,
                        float* delta, void* dqkv, int64_t lddq, int B, int N, int H, int hd, float scale, hipStream_t stream) {
    typedef RCfg<HD> R;
    constexpr size_t ring = (size_t)ST_RING * 2 * ST_KC * R::RB, fl = (size_t)ST_RING * 2 * ST_KC * sizeof(float);
    constexpr size_t smem1 = ring + (R16_THREADS / 64) * 16 * Cfg<bf16_t, HD>::RROW;
    constexpr size_t smem2 = smem1 + fl;
    constexpr size_t smem3 = ring + (ST32_THREADS / 64) * 16 * Cfg<bf16_t, HD>::RROW + fl;
    static OncePerDevice once;
    if (once.need()) {
        set_smem(attn_bwd_dq_stream16_kernel<HD>, smem1);
        set_smem(attn_bwd_dkdv_stream16_kernel<HD>, smem2);
        set_smem(attn_bwd_dkdv_stream32_kernel<HD>, smem3);
    }
    // row blocks of (at most) 14 x 16 rows, evened out over the sequence (queries in the first kernel, keys in the second)
    const int nrb = (N + 223) / 224;
    const int RBk = ((N + nrb - 1) / nrb + 15) / 16 * 16;
    const int64_t items = (int64_t)B * H * nrb;
    const int64_t slots = device_cus();
    const unsigned grid = (unsigned)(items < slots ? items : slots);
    hipLaunchKernelGGL((attn_bwd_dq_stream16_kernel<HD>), dim3(grid), dim3(R16_THREADS), smem1, stream,
                       reinterpret_cast<const bf16_t*>(qkv), ld, reinterpret_cast<const bf16_t*>(out), ldo,
                       reinterpret_cast<const bf16_t*>(dout), lddo, lse, delta, reinterpret_cast<bf16_t*>(dqkv), lddq, N, H, hd, scale,
                       RBk, nrb, (int)items);
    ME_CHECK_LAUNCH("me_attention_bwd(dq stream16)");
    // dK / dV: 32-key waves when the key blocks fill seven of them (see attn_bwd_dkdv_stream32_kernel)
    const int KB32 = (RBk + 31) / 32 * 32;
    if (ME_ST_BWD_ROWS == 32 && KB32 == 224) {
        hipLaunchKernelGGL((attn_bwd_dkdv_stream32_kernel<HD>), dim3(grid), dim3(ST32_THREADS), smem3, stream,
                           reinterpret_cast<const bf16_t*>(qkv), ld, reinterpret_cast<const bf16_t*>(dout), lddo, lse, delta,
                           reinterpret_cast<bf16_t*>(dqkv), lddq, N, H, hd, scale, KB32, nrb, (int)items);
        ME_CHECK_LAUNCH("me_attention_bwd(dkdv stream32)");
        return ME_OK;
    }
    hipLaunchKernelGGL((attn_bwd_dkdv_stream16_kernel<HD>), dim3(grid), dim3(R16_THREADS), smem2, stream,
                       reinterpret_cast<const bf16_t*>(qkv), ld, reinterpret_cast<const bf16_t*>(dout), lddo, lse, delta,
                       reinterpret_cast<bf16_t*>(dqkv), lddq, N, H, hd, scale, RBk, nrb, (int)items);
    ME_CHECK_LAUNCH("me_attention_bwd(dkdv stream16)");
    return ME_OK;
}
template <int HD, int NS>
int launch_bwd_ring16_ns(const void* qkv, int64_t ld, const void* out, int64_t ldo, const void* dout, int64_t lddo, const float* lse,
                         float* delta, void* dqkv, int64_t lddq, int B, int N, int H, int hd, float scale, hipStream_t stream) {
    typedef RCfg<HD> R;
    constexpr size_t smem = (size_t)4 * NS * 32 * R::RB + 2 * NS * 32 * sizeof(float) + (R16_THREADS / 64) * 16 * Cfg<bf16_t, HD>::RROW + 16;
    static OncePerDevice once;
    if (once.need()) { set_smem(attn_bwd_ring16_kernel<HD, NS>, smem); }
    unsigned* ctr = me_work_counters(stream);      // (null while capturing: static schedule)
    if (ctr) ctr += 1;                              // word 1 of the stream's set (word 0: the resident GEMM, XCD 0)
    const int64_t items = (int64_t)B * H;
    const int64_t slots = device_cus();
    const unsigned grid = (unsigned)(items < slots ? items : slots);
    hipLaunchKernelGGL((attn_bwd_ring16_kernel<HD, NS>), dim3(grid), dim3(R16_THREADS), smem, stream,
                       reinterpret_cast<const bf16_t*>(qkv), ld, reinterpret_cast<const bf16_t*>(out), ldo,
                       reinterpret_cast<const bf16_t*>(dout), lddo, lse, delta, reinterpret_cast<bf16_t*>(dqkv), lddq, N, H, hd, scale,
                       (int)items, ctr);
    ME_CHECK_LAUNCH("me_attention_bwd(ring16)");
    return ME_OK;
}
template <int HD>
int launch_bwd_ring16(const void* qkv, int64_t ld, const void* out, int64_t ldo, const void* dout, int64_t lddo, const float* lse,
                      float* delta, void* dqkv, int64_t lddq, int B, int N, int H, int hd, float scale, hipStream_t stream) {
    switch ((N + 31) / 32) {
        case 3: return launch_bwd_ring16_ns<HD, 3>(qkv, ld, out, ldo, dout, lddo, lse, delta, dqkv, lddq, B, N, H, hd, scale, stream);
        case 4: return launch_bwd_ring16_ns<HD, 4>(qkv, ld, out, ldo, dout, lddo, lse, delta, dqkv, lddq, B, N, H, hd, scale, stream);
        case 5: return launch_bwd_ring16_ns<HD, 5>(qkv, ld, out, ldo, dout, lddo, lse, delta, dqkv, lddq, B, N, H, hd, scale, stream);
        case 6: return launch_bwd_ring16_ns<HD, 6>(qkv, ld, out, ldo, dout, lddo, lse, delta, dqkv, lddq, B, N, H, hd, scale, stream);
        default: return launch_bwd_ring16_ns<HD, 7>(qkv, ld, out, ldo, dout, lddo, lse, delta, dqkv, lddq, B, N, H, hd, scale, stream);
    }
}
template <int HD, int NTHR, int MAXN>
int launch_fwd_mid(const void* qkv, int64_t ld, void* out, int64_t ldo, float* lse, int B, int N, int H, int hd, float scale,
                   hipStream_t stream) {
    typedef Cfg<bf16_t, HD> C;
    const size_t smem = (size_t)2 * ((N + 31) / 32 * 32) * C::RROW;
    static OncePerDevice once;
    if (once.need()) { set_smem(attn_fwd_mid_kernel<HD, NTHR, MAXN>, (size_t)2 * MAXN * C::RROW); }
    hipLaunchKernelGGL((attn_fwd_mid_kernel<HD, NTHR, MAXN>), dim3(H, B), dim3(NTHR), smem, stream,
                       reinterpret_cast<const bf16_t*>(qkv), ld, reinterpret_cast<bf16_t*>(out), ldo, lse, N, H, hd, scale);
    ME_CHECK_LAUNCH("me_attention_fwd(mid)");
    return ME_OK;
}
template <int HD, int NTHR, int MAXN>
int launch_bwd_mid(const void* qkv, int64_t ld, const void* out, int64_t ldo, const void* dout, int64_t lddo, const float* lse,
                   float* delta, void* dqkv, int64_t lddq, int B, int N, int H, int hd, float scale, hipStream_t stream) {
    typedef Cfg<bf16_t, HD> C;
    const size_t smem = (size_t)2 * ((N + 31) / 32 * 32) * C::RROW + 2 * MAXN * sizeof(float);
    static OncePerDevice once;
    if (once.need()) { set_smem(attn_bwd_mid_kernel<HD, NTHR, MAXN>, (size_t)2 * MAXN * C::RROW + 2 * MAXN * sizeof(float)); }
    hipLaunchKernelGGL((attn_bwd_mid_kernel<HD, NTHR, MAXN>), dim3(H, B), dim3(NTHR), smem, stream,
                       reinterpret_cast<const bf16_t*>(qkv), ld, reinterpret_cast<const bf16_t*>(out), ldo,
                       reinterpret_cast<const bf16_t*>(dout), lddo, lse, delta, reinterpret_cast<bf16_t*>(dqkv), lddq, N, H, hd,
                       scale);
    ME_CHECK_LAUNCH("me_attention_bwd(mid)");
    return ME_OK;
}

template <int HD>
int launch_fwd_chunk(const void* qkv, int64_t ld, void* out, int64_t ldo, float* lse, int B, int N, int H, int hd, float scale,
                     hipStream_t stream) {
    typedef Cfg<bf16_t, HD> C;
    const size_t smem = (size_t)2 * CH_ROWS * C::RROW;
    static OncePerDevice once;
    if (once.need()) { set_smem(attn_fwd_chunk_kernel<HD>, smem); }
    hipLaunchKernelGGL((attn_fwd_chunk_kernel<HD>), dim3((N + CH_ROWS - 1) / CH_ROWS, H, B), dim3(SM_THREADS), smem, stream,
                       reinterpret_cast<const bf16_t*>(qkv), ld, reinterpret_cast<bf16_t*>(out), ldo, lse, N, H, hd, scale);
    ME_CHECK_LAUNCH("me_attention_fwd(chunk)");
    return ME_OK;
}
template <int HD>
int launch_bwd_chunk(const void* qkv, int64_t ld, const void* out, int64_t ldo, const void* dout, int64_t lddo, const float* lse,
                     float* delta, void* dqkv, int64_t lddq, int B, int N, int H, int hd, float scale, hipStream_t stream) {
    typedef Cfg<bf16_t, HD> C;
    const size_t smem1 = (size_t)2 * CH_ROWS * C::RROW, smem2 = smem1 + 2 * CH_ROWS * sizeof(float);
    static OncePerDevice once;
    if (once.need()) {
        set_smem(attn_bwd_dq_chunk_kernel<HD>, smem1);
        set_smem(attn_bwd_dkdv_chunk_kernel<HD>, smem2);
    }
    dim3 grid((N + CH_ROWS - 1) / CH_ROWS, H, B);
    hipLaunchKernelGGL((attn_bwd_dq_chunk_kernel<HD>), grid, dim3(SM_THREADS), smem1, stream, reinterpret_cast<const bf16_t*>(qkv),
                       ld, reinterpret_cast<const bf16_t*>(out), ldo, reinterpret_cast<const bf16_t*>(dout), lddo, lse, delta,
                       reinterpret_cast<bf16_t*>(dqkv), lddq, N, H, hd, scale);
    ME_CHECK_LAUNCH("me_attention_bwd(dq chunk)");
    hipLaunchKernelGGL((attn_bwd_dkdv_chunk_kernel<HD>), grid, dim3(SM_THREADS), smem2, stream, reinterpret_cast<const bf16_t*>(qkv),
                       ld, reinterpret_cast<const bf16_t*>(dout), lddo, lse, delta, reinterpret_cast<bf16_t*>(dqkv), lddq, N, H, hd,
                       scale);
    ME_CHECK_LAUNCH("me_attention_bwd(dkdv chunk)");
    return ME_OK;
}

template <typename T, int HD>
int launch_fwd(const void* qkv, int64_t ld, void* out, int64_t ldo, float* lse, int B, int N, int H, int hd, float scale,
               float p_drop, uint64_t seed, hipStream_t stream) {
    typedef Cfg<T, HD> C;
    const size_t smem = C::R_BYTES + (C::T_BYTES > C::R_BYTES ? C::T_BYTES : C::R_BYTES);   // V tile: transposed (fp32) or row-major (bf16)
    static OncePerDevice once;
    if (once.need()) { set_smem(attn_fwd_kernel<T, HD>, smem); }
    dim3 grid((N + QPB - 1) / QPB, H, B);
    hipLaunchKernelGGL((attn_fwd_kernel<T, HD>), grid, dim3(AT_THREADS), smem, stream, reinterpret_cast<const T*>(qkv), ld,
                       reinterpret_cast<T*>(out), ldo, lse, N, H, hd, scale, p_drop, seed);
    ME_CHECK_LAUNCH("me_attention_fwd");
    return ME_OK;
}

template <typename T, int HD>
int launch_bwd(const void* qkv, int64_t ld, const void* dout, int64_t lddo, const float* lse, const float* delta, void* dqkv,
               int64_t lddq, int B, int N, int H, int hd, float scale, float p_drop, uint64_t seed, hipStream_t stream) {
    typedef Cfg<T, HD> C;
    constexpr size_t TB = TRead<T, HD>::kNeedsTransposedTile ? C::T_BYTES : 0;
    const size_t smem1 = 2 * C::R_BYTES + 2 * TB + 2 * KVT * sizeof(float);
    const size_t smem2 = 2 * C::R_BYTES + TB;
    static OncePerDevice once;
    if (once.need()) {
        set_smem(attn_bwd_dkdv_kernel<T, HD>, smem1);
        set_smem(attn_bwd_dq_kernel<T, HD>, smem2);
    }
    dim3 grid((N + QPB - 1) / QPB, H, B);
    hipLaunchKernelGGL((attn_bwd_dkdv_kernel<T, HD>), grid, dim3(AT_THREADS), smem1, stream,
                       reinterpret_cast<const T*>(qkv), ld, reinterpret_cast<const T*>(dout), lddo, lse, delta,
                       reinterpret_cast<T*>(dqkv), lddq, N, H, hd, scale, p_drop, seed);
    ME_CHECK_LAUNCH("me_attention_bwd(dkdv)");
    hipLaunchKernelGGL((attn_bwd_dq_kernel<T, HD>), grid, dim3(AT_THREADS), smem2, stream,
                       reinterpret_cast<const T*>(qkv), ld, reinterpret_cast<const T*>(dout), lddo, lse, delta,
                       reinterpret_cast<T*>(dqkv), lddq, N, H, hd, scale, p_drop, seed);
    ME_CHECK_LAUNCH("me_attention_bwd(dq)");
    return ME_OK;
}

int check_attn_args(const char* fn, int64_t ld, int B, int N, int H, int hd, int dtype) {
    ME_CHECK_ARG(me_dtype_ok(dtype), "%s: bad dtype", fn);
    ME_CHECK_ARG(B > 0 && N > 0 && H > 0 && hd > 0, "%s: bad shape B=%d N=%d H=%d hd=%d", fn, B, N, H, hd);
    const int E = dtype == ME_BF16 ? 8 : 4;
    ME_CHECK_ARG(hd % E == 0, "%s: head_dim=%d must be a multiple of %d", fn, hd, E);
    ME_CHECK_ARG(hd <= 128, "%s: head_dim=%d > 128 unsupported", fn, hd);
    ME_CHECK_ARG(ld % E == 0, "%s: row stride must be a multiple of %d elements", fn, E);
    ME_CHECK_ARG(B <= 65535 && H <= 65535, "%s: B and H must be <= 65535", fn);
    return ME_OK;
}

}  // namespace

#define ATTN_DISPATCH(FN, ...)                                                                   \
    do {                                                                                         \
        if (dtype == ME_BF16) {                                                                  \
            if (head_dim <= 32) return FN<bf16_t, 32>(__VA_ARGS__);                              \
            if (head_dim <= 64) return FN<bf16_t, 64>(__VA_ARGS__);                              \
            return FN<bf16_t, 128>(__VA_ARGS__);                                                 \
        } else {                                                                                 \
            if (head_dim <= 32) return FN<float, 32>(__VA_ARGS__);                               \
            if (head_dim <= 64) return FN<float, 64>(__VA_ARGS__);                               \
            return FN<float, 128>(__VA_ARGS__);                                                  \
        }                                                                                        \
    } while (0)

extern "C" int me_attention_fwd(const void* qkv, int64_t ld_qkv, void* out, int64_t ld_out, float* lse, int B, int N,
                                int H, int head_dim, float scale, int dtype, float p_drop, uint64_t seed, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ProfScope prof(ME_PROF_ATTN_FWD, dtype, (int64_t)B * H, N, head_dim, stream);
    ME_CHECK_ARG(qkv && out, "me_attention_fwd: null pointer");
    int rc = check_attn_args("me_attention_fwd", ld_qkv, B, N, H, head_dim, dtype);
    if (rc) return rc;
    ME_CHECK_ARG(ld_out % 4 == 0, "me_attention_fwd: ld_out must be a multiple of 4");
    ME_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "me_attention_fwd: p_drop must be in [0, 1)");
    // very short sequences (the Tabular / Graph recipes): one wave per (batch, head), attention_tiny.hip
    if (ME_TINY_ATTN && p_drop == 0.f && N <= SM_MINN && attn_tiny_ok(dtype, ld_qkv, ld_out, B, N, H, head_dim))
        return launch_attn_tiny_fwd(dtype, qkv, ld_qkv, out, ld_out, lse, B, N, H, head_dim, scale, stream);
    if (p_drop == 0.f && dtype == ME_BF16 && head_dim <= 64 && N > SM_MINN && N <= SM_MAXN) {
        if (N <= RS_MAXN && (int64_t)N * ld_qkv * 2 < (int64_t)0x7e000000) {
            if (head_dim <= 32) return launch_fwd_ring16<32>(qkv, ld_qkv, out, ld_out, lse, B, N, H, head_dim, scale, stream);
            return launch_fwd_ring16<64>(qkv, ld_qkv, out, ld_out, lse, B, N, H, head_dim, scale, stream);
        }
        if (head_dim <= 32) return launch_fwd_small<32>(qkv, ld_qkv, out, ld_out, lse, B, N, H, head_dim, scale, stream);
        return launch_fwd_small<64>(qkv, ld_qkv, out, ld_out, lse, B, N, H, head_dim, scale, stream);
    }
    // (head_dim <= 32: half the MFMA work per key for the same softmax arithmetic -- the 32-row kernels below measured faster there)
    if (p_drop == 0.f && dtype == ME_BF16 && head_dim > 32 && head_dim <= 64 && N > SM_MAXN && (int64_t)N * ld_qkv * 2 < (int64_t)0x7e000000 &&
        (int64_t)B * H * ((N + 223) / 224) < (int64_t)0x7fffffff && ld_out % 8 == 0) {
        return launch_fwd_stream16<64>(qkv, ld_qkv, out, ld_out, lse, B, N, H, head_dim, scale, stream);
    }
    if (p_drop == 0.f && dtype == ME_BF16 && head_dim <= 64 && N > SM_MAXN && N <= MD_MAXN) {
        if (head_dim <= 32) return launch_fwd_mid<32, 512, MD_MAXN>(qkv, ld_qkv, out, ld_out, lse, B, N, H, head_dim, scale, stream);
        return launch_fwd_mid<64, 512, MD_MAXN>(qkv, ld_qkv, out, ld_out, lse, B, N, H, head_dim, scale, stream);
    }
    if (p_drop == 0.f && dtype == ME_BF16 && head_dim <= 64 && N > MD_MAXN && B <= 65535) {
        if (head_dim <= 32) return launch_fwd_chunk<32>(qkv, ld_qkv, out, ld_out, lse, B, N, H, head_dim, scale, stream);
        return launch_fwd_chunk<64>(qkv, ld_qkv, out, ld_out, lse, B, N, H, head_dim, scale, stream);
    }
    ATTN_DISPATCH(launch_fwd, qkv, ld_qkv, out, ld_out, lse, B, N, H, head_dim, scale, p_drop, seed, stream);
}

extern "C" int me_attention_bwd(const void* qkv, int64_t ld_qkv, const void* out, int64_t ld_out, const void* dout,
                                int64_t ld_dout, const float* lse, float* delta, void* dqkv, int64_t ld_dqkv, int B,
                                int N, int H, int head_dim, float scale, int dtype, float p_drop, uint64_t seed,
                                void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ProfScope prof(ME_PROF_ATTN_BWD, dtype, (int64_t)B * H, N, head_dim, stream);
    ME_CHECK_ARG(qkv && out && dout && lse && delta && dqkv, "me_attention_bwd: null pointer");
    int rc = check_attn_args("me_attention_bwd", ld_qkv, B, N, H, head_dim, dtype);
    if (rc) return rc;
    const int E = dtype == ME_BF16 ? 8 : 4;
    ME_CHECK_ARG(ld_dout % E == 0 && ld_dqkv % 4 == 0, "me_attention_bwd: bad strides");
    ME_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "me_attention_bwd: p_drop must be in [0, 1)");
    if (ME_TINY_ATTN && p_drop == 0.f && N <= SM_MINN && attn_tiny_ok(dtype, ld_qkv, ld_out, B, N, H, head_dim) && ld_dout % E == 0 && ld_dqkv % E == 0)
        return launch_attn_tiny_bwd(dtype, qkv, ld_qkv, out, ld_out, dout, ld_dout, lse, delta, dqkv, ld_dqkv, B, N, H, head_dim, scale, stream);
    if (p_drop == 0.f && dtype == ME_BF16 && head_dim <= 64 && N > SM_MINN && N <= SM_MAXN && ld_out % 8 == 0) {
        if (N <= RS_MAXN && (int64_t)N * ld_qkv * 2 < (int64_t)0x7e000000 && (int64_t)N * ld_dout * 2 < (int64_t)0x7e000000) {
            if (head_dim <= 32)
                return launch_bwd_ring16<32>(qkv, ld_qkv, out, ld_out, dout, ld_dout, lse, delta, dqkv, ld_dqkv, B, N, H, head_dim, scale, stream);
            return launch_bwd_ring16<64>(qkv, ld_qkv, out, ld_out, dout, ld_dout, lse, delta, dqkv, ld_dqkv, B, N, H, head_dim, scale, stream);
        }
        if (head_dim <= 32)
            return launch_bwd_small<32>(qkv, ld_qkv, out, ld_out, dout, ld_dout, lse, delta, dqkv, ld_dqkv, B, N, H, head_dim,
                                        scale, stream);
        return launch_bwd_small<64>(qkv, ld_qkv, out, ld_out, dout, ld_dout, lse, delta, dqkv, ld_dqkv, B, N, H, head_dim, scale,
                                    stream);
    }
    if (p_drop == 0.f && dtype == ME_BF16 && head_dim > 32 && head_dim <= 64 && N >= ST_BWD_MINN && ld_out % 8 == 0 && ld_dqkv % 8 == 0 &&
        (int64_t)N * ld_qkv * 2 < (int64_t)0x7e000000 && (int64_t)N * ld_dout * 2 < (int64_t)0x7e000000 &&
        (int64_t)B * H * ((N + 223) / 224) < (int64_t)0x7fffffff) {
        return launch_bwd_stream16<64>(qkv, ld_qkv, out, ld_out, dout, ld_dout, lse, delta, dqkv, ld_dqkv, B, N, H, head_dim, scale, stream);
    }
    if (p_drop == 0.f && dtype == ME_BF16 && head_dim <= 64 && N > SM_MAXN && N <= MD_MAXN && ld_out % 8 == 0) {
        if (head_dim <= 32)
            return launch_bwd_mid<32, 512, MD_MAXN>(qkv, ld_qkv, out, ld_out, dout, ld_dout, lse, delta, dqkv, ld_dqkv, B, N, H,
                                                    head_dim, scale, stream);
        return launch_bwd_mid<64, 512, MD_MAXN>(qkv, ld_qkv, out, ld_out, dout, ld_dout, lse, delta, dqkv, ld_dqkv, B, N, H,
                                                head_dim, scale, stream);
    }
    if (p_drop == 0.f && dtype == ME_BF16 && head_dim <= 64 && N > MD_MAXN && ld_out % 8 == 0) {
        if (head_dim <= 32)
            return launch_bwd_chunk<32>(qkv, ld_qkv, out, ld_out, dout, ld_dout, lse, delta, dqkv, ld_dqkv, B, N, H, head_dim, scale,
                                        stream);
        return launch_bwd_chunk<64>(qkv, ld_qkv, out, ld_out, dout, ld_dout, lse, delta, dqkv, ld_dqkv, B, N, H, head_dim, scale,
                                    stream);
    }
    const int64_t rows = (int64_t)B * N;
    const int64_t nw = rows * H;
    const int lph = head_dim / 8;
    if (dtype == ME_BF16 && head_dim % 8 == 0 && (lph & (lph - 1)) == 0 && lph <= 64 && ld_out % 8 == 0 && ld_dout % 8 == 0)
        hipLaunchKernelGGL(attn_delta_vec_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream,
                           reinterpret_cast<const bf16_t*>(out), ld_out, reinterpret_cast<const bf16_t*>(dout), ld_dout, delta, N,
                           H, head_dim, rows);
    else
        hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, stream, out, ld_out, dout, ld_dout,
                           dtype, delta, N, H, head_dim, rows);
    ME_CHECK_LAUNCH("me_attention_bwd(delta)");
    ATTN_DISPATCH(launch_bwd, qkv, ld_qkv, dout, ld_dout, lse, delta, dqkv, ld_dqkv, B, N, H, head_dim, scale, p_drop, seed, stream);
}
