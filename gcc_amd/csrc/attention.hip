// Self attention over the H*W positions of one image (reference Self_Attn, models/SAGAN.py:72-104):
//     energy[i][j] = q_i . k_j ;  A = softmax_j(energy) ;  o_i = sum_j A[i][j] v_j ;  y = gamma * o + x
// q, k (C8 = C/8 channels) and v (C channels) are channel slices of one NHWC bf16 buffer written by the three 1x1 convs.
//
// The N x N matrices never exist in memory (N = 1024 at the generator's 32x32 map: 4 MiB per image and matrix, 268 MiB
// for a batch of 64).  The score tiles are recomputed where they are needed -- their inner dimension is only C/8 -- and
// all products run on the matrix cores (v_mfma_f32_16x16x32_bf16):
//   forward   pass 1: row maximum m_i and row sum l_i of exp(s - m) (online, over 32-key steps) -> `stats` [B][N][2]
//             pass 2: P = exp(s - m_i) / l_i (bf16) ; O += P V
//   backward  with do = gamma dy, D_i = sum_j P[i][j] dP[i][j] (first pass of the query kernel) and dgamma += sum dy o:
//             dP[i][j] = do_i . v_j ; dS = P (dP - D_i) ; dq_i = sum_j dS[i][j] k_j          (kernel per 64 queries)
//             dv_j = sum_i P[i][j] do_i ; dk_j = sum_i dS[i][j] q_i                         (kernel per 64 keys)
//   (the residual branch dx += dy is the caller's: it owns the buffer the three 1x1 data gradients are added into).
//
// Operand layouts.  A wave owns 16 rows; a score tile comes out of the MFMA in the accumulator layout
// (row = 4 * (lane >> 4) + i, column = lane & 15).  Computing the tile TRANSPOSED (keys x queries in the query kernels,
// queries x keys in the key kernel) leaves each lane holding, for ITS row of the next product, eight entries of the
// reduction dimension: two 16-wide tiles give slots e = 0..3 -> tile 0, index 4g + e and e = 4..7 -> tile 1, index
// 4g + e - 4 (g = lane >> 4).  That is a valid A fragment as long as the B operand enumerates the reduction index in
// the same order, so no cross-lane movement is needed; the B operand (v, k, dy or q rows of the 32-step) is staged
// through LDS transposed ([channel][row]), where a lane's eight slots are two 8-byte reads.
#include "common.hpp"

namespace {

constexpr int TSTRIDE = 36;      // bf16 per transposed LDS row: 32 + 4 pad (72 bytes: 8-byte aligned reads, rows 18 banks apart)
constexpr float NEG = -3.0e38f;

struct AttnArgs {
    const bf16_t* qkv; int ldq, qoff, koff, voff;
    const bf16_t* x; int ldx;
    bf16_t* y; int ldy;
    bf16_t* o; int ldo;          // pre-gamma attention output, [B][N][ldo]
    float* stats;                // [B][N][2]: row maximum, row sum of exp(s - max)
    float* A;                    // optional [B][N][N] attention map (what the reference's forward also returns)
    const float* gamma;
    int B, N, C, C8;
    // backward
    const bf16_t* dy; int lddy;
    bf16_t* dqkv; int lddq;      // same slice offsets as qkv
    float* rowdot;               // [B][N]: D_i = sum_j P[i][j] dP[i][j] (written by the query kernel, read by the key kernel)
    float* dgamma;
};

__device__ __forceinline__ bf16x8 ldfrag(const bf16_t* p, bool ok) {
    const i32x4 z = {0, 0, 0, 0};
    const i32x4 v = ok ? *(const i32x4*)p : z;
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 frag_of(const float* f) { return __builtin_bit_cast(bf16x8, pack8(f)); }
// lo[e] = f[e] - float(hi[e]): the part of f the bf16 fragment hi dropped
__device__ __forceinline__ void split_lo(const float* f, const bf16x8& hi, float* lo) {
    float h[8];
    unpack8(__builtin_bit_cast(i32x4, hi), h);
#pragma unroll
    for (int e = 0; e < 8; e++) lo[e] = f[e] - h[e];
}

// 32 rows x 64 channels of an NHWC slice -> registers (thread: row = tid >> 3, 8 channels at (tid & 7) * 8)
__device__ __forceinline__ i32x4 stage_load(const bf16_t* img, int ld, int off, int c0, int width8, int r0, int N) {
    const int r = r0 + (threadIdx.x >> 3), c = c0 + (threadIdx.x & 7) * 8;
    const i32x4 z = {0, 0, 0, 0};
    return (r < N && c < width8) ? *(const i32x4*)(img + (size_t)r * ld + off + c) : z;
}
// ... -> LDS transposed [64 channels][32 rows]
__device__ __forceinline__ void stage_store(bf16_t (*T)[TSTRIDE], const i32x4& v) {
    const int r = threadIdx.x >> 3, ch = (threadIdx.x & 7) * 8;
#pragma unroll
    for (int j = 0; j < 8; j++) T[ch + j][r] = (bf16_t)(((uint32_t)v[j >> 1]) >> (16 * (j & 1)));
}
// B fragment of channel `ch`: reduction slots 4g..4g+3 and 16+4g..16+4g+3 of the staged 32 rows
__device__ __forceinline__ bf16x8 tfrag(const bf16_t (*T)[TSTRIDE], int ch, int g) {
    const i32x2 lo = *(const i32x2*)(&T[ch][g * 4]), hi = *(const i32x2*)(&T[ch][16 + g * 4]);
    const i32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, v);
}

// ---------------------------------------------------------------------------------------------
// grid (ceil(N / 64), ceil(C / 64), B): 4 waves x 16 queries, 64 output channels
template <int DKS>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t VT[2][64][TSTRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, q0 = blockIdx.x * 64 + wave * 16, cb0 = blockIdx.y * 64;
    const int N = a.N, C = a.C, dk8 = ceil8(a.C8), c8 = ceil8(C);
    const bf16_t* base = a.qkv + (size_t)b * N * a.ldq;
    bf16x8 qf[DKS];
#pragma unroll
    for (int kk = 0; kk < DKS; kk++) {
        const int d = kk * 32 + g * 8;
        qf[kk] = ldfrag(base + (size_t)(q0 + lr) * a.ldq + a.qoff + d, q0 + lr < N && d < dk8);
    }
    const int nsteps = cdiv(N, 32);
    // transposed score tiles of one 32-key step: s[4h + i] = q_(q0 + lr) . k_(t0 + 16h + 4g + i)
#define GCC_ATTN_SCORES_T(t0, s)                                                                          \
    do {                                                                                                  \
        _Pragma("unroll") for (int h = 0; h < 2; h++) {                                                   \
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};                                                             \
            const int key = (t0) + h * 16 + lr;                                                           \
            _Pragma("unroll") for (int kk = 0; kk < DKS; kk++) {                                          \
                const int d = kk * 32 + g * 8;                                                            \
                acc = mma(ldfrag(base + (size_t)key * a.ldq + a.koff + d, key < N && d < dk8), qf[kk], acc); \
            }                                                                                             \
            _Pragma("unroll") for (int i = 0; i < 4; i++)                                                 \
                s[h * 4 + i] = ((t0) + h * 16 + g * 4 + i < N) ? acc[i] : NEG;                            \
        }                                                                                                 \
    } while (0)

    float m = NEG, l = 0.f;
    for (int t = 0; t < nsteps; t++) {
        float s[8];
        GCC_ATTN_SCORES_T(t * 32, s);
        float mx = s[0];
#pragma unroll
        for (int e = 1; e < 8; e++) mx = fmaxf(mx, s[e]);
        const float mn = fmaxf(m, mx);
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) sum += s[e] > -1.0e38f ? __expf(s[e] - mn) : 0.f;
        l = l * __expf(m - mn) + sum;
        m = mn;
    }
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        const float m2 = __shfl_xor(m, o, 64), l2 = __shfl_xor(l, o, 64);
        const float mn = fmaxf(m, m2);
        l = l * __expf(m - mn) + l2 * __expf(m2 - mn);
        m = mn;
    }
    const float inv_l = 1.f / l;
    if (blockIdx.y == 0 && g == 0 && q0 + lr < N) {
        float* st = a.stats + ((size_t)b * N + q0 + lr) * 2;
        st[0] = m; st[1] = l;
    }

    f32x4 O[4];
#pragma unroll
    for (int cb = 0; cb < 4; cb++) O[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    i32x4 vreg = stage_load(base, a.ldq, a.voff, cb0, c8, 0, N);
    stage_store(VT[0], vreg);
    __syncthreads();
    for (int t = 0; t < nsteps; t++) {
        if (t + 1 < nsteps) vreg = stage_load(base, a.ldq, a.voff, cb0, c8, (t + 1) * 32, N);
        float s[8], p[8];
        GCC_ATTN_SCORES_T(t * 32, s);
#pragma unroll
        for (int e = 0; e < 8; e++) p[e] = s[e] > -1.0e38f ? __expf(s[e] - m) * inv_l : 0.f;
        if (a.A && blockIdx.y == 0 && q0 + lr < N) {
            float* Ar = a.A + ((size_t)b * N + q0 + lr) * N;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int key = t * 32 + (e >> 2) * 16 + g * 4 + (e & 3);
                if (key < N) Ar[key] = p[e];
            }
        }
        const bf16x8 pf = frag_of(p);
#pragma unroll
        for (int cb = 0; cb < 4; cb++) O[cb] = mma(pf, tfrag(VT[t & 1], cb * 16 + lr, g), O[cb]);
        if (t + 1 < nsteps) stage_store(VT[(t + 1) & 1], vreg);
        __syncthreads();
    }
#undef GCC_ATTN_SCORES_T
    const float gm = a.gamma[0];
#pragma unroll
    for (int cb = 0; cb < 4; cb++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int q = q0 + g * 4 + i, c = cb0 + cb * 16 + lr;
            if (q < N && c < C) {
                const size_t pix = (size_t)b * N + q;
                const bf16_t ob = f2bf(O[cb][i]);      // y is computed from the bf16-stored o, as backward will read it
                a.o[pix * a.ldo + c] = ob;
                a.y[pix * a.ldy + c] = f2bf(gm * bf2f(ob) + bf2f(a.x[pix * a.ldx + c]));
            }
        }
}

// ---------------------------------------------------------------------------------------------
// dgamma += sum dy o.  One wave per row, grid-stride; per-workgroup partial sums go to `rowdot` (free until the query
// kernel runs) and one workgroup adds them up in a fixed order: no atomics, reproducible.
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const AttnArgs a) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rows = a.B * a.N, c8 = ceil8(a.C);
    float tot = 0.f;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        float acc = 0.f;
        for (int c = lane * 8; c < c8; c += 512) {
            float d[8], o[8];
            unpack8(*(const i32x4*)(a.dy + (size_t)row * a.lddy + c), d);
            unpack8(*(const i32x4*)(a.o + (size_t)row * a.ldo + c), o);
#pragma unroll
            for (int j = 0; j < 8; j++) acc += (c + j < a.C) ? d[j] * o[j] : 0.f;
        }
        tot += wave_sum(acc);
    }
    if (lane == 0) red[wave] = tot;
    __syncthreads();
    if (threadIdx.x == 0) a.rowdot[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void attn_bwd_dgamma_kernel(const float* partial, int n, float* dgamma) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) dgamma[0] += red[0] + red[1] + red[2] + red[3];
}

// ---------------------------------------------------------------------------------------------
// dq.  grid (ceil(N / 64), B): 4 waves x 16 queries; CS = 32-channel steps of the do . v products
template <int DKS, int CS>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t KT[2][64][TSTRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, g = lane >> 4;
    const int b = blockIdx.y, q0 = blockIdx.x * 64 + wave * 16;
    const int N = a.N, dk8 = ceil8(a.C8), c8 = ceil8(a.C);
    const bf16_t* base = a.qkv + (size_t)b * N * a.ldq;
    const bool qok = q0 + lr < N;
    bf16x8 qf[DKS], df[CS];
#pragma unroll
    for (int kk = 0; kk < DKS; kk++) {
        const int d = kk * 32 + g * 8;
        qf[kk] = ldfrag(base + (size_t)(q0 + lr) * a.ldq + a.qoff + d, qok && d < dk8);
    }
#pragma unroll
    for (int kk = 0; kk < CS; kk++) {
        const int c = kk * 32 + g * 8;
        df[kk] = ldfrag(a.dy + ((size_t)b * N + q0 + lr) * a.lddy + c, qok && c < c8);
    }
    const float gm = a.gamma[0];
    float m = 0.f, il = 0.f;
    if (qok) {
        const float* st = a.stats + ((size_t)b * N + q0 + lr) * 2;
        m = st[0]; il = 1.f / st[1];
    }
    const int nsteps = cdiv(N, 32);
    // transposed tiles of one 32-key step: p[4h + i], dp[4h + i] for key t0 + 16h + 4g + i (dp = do_i . v_key)
#define GCC_ATTN_P_DP(t0, p, dp)                                                                              \
    do {                                                                                                      \
        _Pragma("unroll") for (int h = 0; h < 2; h++) {                                                       \
            const int key = (t0) + h * 16 + lr;                                                               \
            const bf16_t* kr = base + (size_t)key * a.ldq;                                                    \
            f32x4 s_ = {0.f, 0.f, 0.f, 0.f}, d_ = {0.f, 0.f, 0.f, 0.f};                                       \
            _Pragma("unroll") for (int kk = 0; kk < DKS; kk++) {                                              \
                const int d = kk * 32 + g * 8;                                                                \
                s_ = mma(ldfrag(kr + a.koff + d, key < N && d < dk8), qf[kk], s_);                            \
            }                                                                                                 \
            _Pragma("unroll") for (int kk = 0; kk < CS; kk++) {                                               \
                const int c = kk * 32 + g * 8;                                                                \
                d_ = mma(ldfrag(kr + a.voff + c, key < N && c < c8), df[kk], d_);                             \
            }                                                                                                 \
            _Pragma("unroll") for (int i = 0; i < 4; i++) {                                                   \
                const bool ok = qok && ((t0) + h * 16 + g * 4 + i < N);                                       \
                p[h * 4 + i] = ok ? __expf(s_[i] - m) * il : 0.f;                                             \
                dp[h * 4 + i] = gm * d_[i];                                                                   \
            }                                                                                                 \
        }                                                                                                     \
    } while (0)
    // D_i = sum_j P[i][j] dP[i][j] from the SAME tiles dS is built from: the rows of dS then sum to zero to fp32
    // rounding.  (do_i . o_i with the bf16-stored o is the same number only to 2^-9, and that error is multiplied by
    // the keys' common component -- their bias -- in dq.)
    float D = 0.f;
    for (int t = 0; t < nsteps; t++) {
        float p[8], dp[8];
        GCC_ATTN_P_DP(t * 32, p, dp);
#pragma unroll
        for (int e = 0; e < 8; e++) D += p[e] * dp[e];
    }
    D += __shfl_xor(D, 16, 64);
    D += __shfl_xor(D, 32, 64);
    if (g == 0 && qok) a.rowdot[(size_t)b * N + q0 + lr] = D;
    f32x4 dQ[4];
#pragma unroll
    for (int db = 0; db < 4; db++) dQ[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    i32x4 kreg = stage_load(base, a.ldq, a.koff, 0, dk8, 0, N);
    stage_store(KT[0], kreg);
    __syncthreads();
    for (int t = 0; t < nsteps; t++) {
        if (t + 1 < nsteps) kreg = stage_load(base, a.ldq, a.koff, 0, dk8, (t + 1) * 32, N);
        float p[8], ds[8];
        GCC_ATTN_P_DP(t * 32, p, ds);
#pragma unroll
        for (int e = 0; e < 8; e++) ds[e] = p[e] * (ds[e] - D);
        // dS rows sum to zero (softmax Jacobian) and the keys share their bias: the product cancels, so dS enters as a
        // bf16 pair hi + lo (16 mantissa bits); the second MFMA costs little, the inner dimension being C/8
        float dl[8];
        const bf16x8 dsf = frag_of(ds);
        split_lo(ds, dsf, dl);
        const bf16x8 dsl = frag_of(dl);
#pragma unroll
        for (int db = 0; db < 4; db++)
            if (db * 16 < dk8) {
                const bf16x8 kt = tfrag(KT[t & 1], db * 16 + lr, g);
                dQ[db] = mma(dsf, kt, dQ[db]);
                dQ[db] = mma(dsl, kt, dQ[db]);
            }
        if (t + 1 < nsteps) stage_store(KT[(t + 1) & 1], kreg);
        __syncthreads();
    }
#pragma unroll
    for (int db = 0; db < 4; db++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int q = q0 + g * 4 + i, d = db * 16 + lr;
            if (q < N && d < dk8) a.dqkv[((size_t)b * N + q) * a.lddq + a.qoff + d] = f2bf(dQ[db][i]);
        }
#undef GCC_ATTN_P_DP
}

// ---------------------------------------------------------------------------------------------
// dv, dk.  grid (ceil(N / 64), ceil(C / 64), B): 4 waves x 16 keys, 64 channels of dv; dk by the first channel block
template <int DKS, int CS>
__global__ __launch_bounds__(256) void attn_bwd_k_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t DT[2][64][TSTRIDE];   // dy of the query step, transposed
    __shared__ __attribute__((aligned(16))) bf16_t QT[2][64][TSTRIDE];   // q of the query step, transposed
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, k0 = blockIdx.x * 64 + wave * 16, cb0 = blockIdx.y * 64;
    const int N = a.N, C = a.C, dk8 = ceil8(a.C8), c8 = ceil8(C);
    const bf16_t* base = a.qkv + (size_t)b * N * a.ldq;
    const bf16_t* dyb = a.dy + (size_t)b * N * a.lddy;
    const bool kok = k0 + lr < N;
    const bool want_dk = blockIdx.y == 0;
    bf16x8 kf[DKS], vf[CS];
#pragma unroll
    for (int kk = 0; kk < DKS; kk++) {
        const int d = kk * 32 + g * 8;
        kf[kk] = ldfrag(base + (size_t)(k0 + lr) * a.ldq + a.koff + d, kok && d < dk8);
    }
#pragma unroll
    for (int kk = 0; kk < CS; kk++) {
        const int c = kk * 32 + g * 8;
        vf[kk] = ldfrag(base + (size_t)(k0 + lr) * a.ldq + a.voff + c, kok && c < c8);
    }
    const float gm = a.gamma[0];
    f32x4 dV[4], dK[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { dV[j] = f32x4{0.f, 0.f, 0.f, 0.f}; dK[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int nsteps = cdiv(N, 32);
    i32x4 dreg = stage_load(dyb, a.lddy, 0, cb0, c8, 0, N);
    i32x4 qreg = stage_load(base, a.ldq, a.qoff, 0, dk8, 0, N);
    stage_store(DT[0], dreg);
    stage_store(QT[0], qreg);
    __syncthreads();
    for (int t = 0; t < nsteps; t++) {
        if (t + 1 < nsteps) {
            dreg = stage_load(dyb, a.lddy, 0, cb0, c8, (t + 1) * 32, N);
            qreg = stage_load(base, a.ldq, a.qoff, 0, dk8, (t + 1) * 32, N);
        }
        float p[8], ds[8];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int qrow = t * 32 + h * 16 + lr;               // A operand row of this lane
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < DKS; kk++) {
                const int d = kk * 32 + g * 8;
                s = mma(ldfrag(base + (size_t)qrow * a.ldq + a.qoff + d, qrow < N && d < dk8), kf[kk], s);
            }
#pragma unroll
            for (int kk = 0; kk < CS; kk++) {
                const int c = kk * 32 + g * 8;
                dp = mma(ldfrag(dyb + (size_t)qrow * a.lddy + c, qrow < N && c < c8), vf[kk], dp);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int q = t * 32 + h * 16 + g * 4 + i;        // accumulator row: the query of this entry
                float pv = 0.f, dv = 0.f;
                if (q < N && kok) {
                    const float* st = a.stats + ((size_t)b * N + q) * 2;
                    pv = __expf(s[i] - st[0]) / st[1];
                    dv = pv * (gm * dp[i] - a.rowdot[(size_t)b * N + q]);
                }
                p[h * 4 + i] = pv; ds[h * 4 + i] = dv;
            }
        }
        const bf16x8 pf = frag_of(p), dsf = frag_of(ds);
#pragma unroll
        for (int cb = 0; cb < 4; cb++) dV[cb] = mma(pf, tfrag(DT[t & 1], cb * 16 + lr, g), dV[cb]);
        if (want_dk) {
            float dl[8];
            split_lo(ds, dsf, dl);               // hi + lo pair, as in the query kernel
            const bf16x8 dsl = frag_of(dl);
#pragma unroll
            for (int db = 0; db < 4; db++)
                if (db * 16 < dk8) {
                    const bf16x8 qt = tfrag(QT[t & 1], db * 16 + lr, g);
                    dK[db] = mma(dsf, qt, dK[db]);
                    dK[db] = mma(dsl, qt, dK[db]);
                }
        }
        if (t + 1 < nsteps) {
            stage_store(DT[(t + 1) & 1], dreg);
            stage_store(QT[(t + 1) & 1], qreg);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int key = k0 + g * 4 + i;
            if (key >= N) continue;
            bf16_t* row = a.dqkv + ((size_t)b * N + key) * a.lddq;
            const int c = cb0 + j * 16 + lr, d = j * 16 + lr;
            if (c < c8) row[a.voff + c] = f2bf(c < C ? gm * dV[j][i] : 0.f);
            if (want_dk && d < dk8) row[a.koff + d] = f2bf(dK[j][i]);
        }
}

bool ok_geom(int B, int N, int C, int C8) { return B > 0 && N > 0 && C > 0 && C <= 512 && C8 > 0 && C8 <= 64 && C8 <= C; }

}  // namespace

extern "C" int gcc_attention_fwd(const void* qkv, int ldq, int qoff, int koff, int voff, const void* x, int ldx,
                                 const float* gamma, int B, int N, int C, int C8, void* y, int ldy, void* o, int ldo,
                                 float* stats, float* A, gcc_stream_t stream) {
    GCC_ENTER();
    if (!qkv || !x || !gamma || !y || !o || !stats) return GCC_ERR_BAD_ARG;
    if (!ok_geom(B, N, C, C8)) return GCC_ERR_UNSUPPORTED;
    if ((ldq | qoff | koff | voff | ldx | ldy | ldo) & 7) return GCC_ERR_BAD_ARG;
    AttnArgs a = {};
    a.qkv = (const bf16_t*)qkv; a.ldq = ldq; a.qoff = qoff; a.koff = koff; a.voff = voff;
    a.x = (const bf16_t*)x; a.ldx = ldx; a.y = (bf16_t*)y; a.ldy = ldy; a.o = (bf16_t*)o; a.ldo = ldo;
    a.stats = stats; a.A = A; a.gamma = gamma; a.B = B; a.N = N; a.C = C; a.C8 = C8;
    const dim3 grid(cdiv(N, 64), cdiv(C, 64), B);
    if (C8 <= 32) hipLaunchKernelGGL((attn_fwd_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, a);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

extern "C" int gcc_attention_bwd(const void* qkv, int ldq, int qoff, int koff, int voff, const void* o, int ldo,
                                 const float* stats, const float* gamma, const void* dy, int lddy, int B, int N, int C, int C8,
                                 void* dqkv, int lddq, float* rowdot, float* dgamma, gcc_stream_t stream) {
    GCC_ENTER();
    if (!qkv || !o || !stats || !gamma || !dy || !dqkv || !rowdot) return GCC_ERR_BAD_ARG;
    if (!ok_geom(B, N, C, C8)) return GCC_ERR_UNSUPPORTED;
    if ((ldq | qoff | koff | voff | ldo | lddy | lddq) & 7) return GCC_ERR_BAD_ARG;
    AttnArgs a = {};
    a.qkv = (const bf16_t*)qkv; a.ldq = ldq; a.qoff = qoff; a.koff = koff; a.voff = voff;
    a.o = (bf16_t*)o; a.ldo = ldo; a.stats = (float*)stats; a.gamma = gamma; a.B = B; a.N = N; a.C = C; a.C8 = C8;
    a.dy = (const bf16_t*)dy; a.lddy = lddy; a.dqkv = (bf16_t*)dqkv; a.lddq = lddq; a.rowdot = rowdot; a.dgamma = dgamma;
    hipStream_t st = (hipStream_t)stream;
    int pb = cdiv(B * N, 4);
    if (pb > 512) pb = 512;
    if (dgamma) {
        hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3(pb), dim3(256), 0, st, a);
        GCC_CHECK_LAUNCH();
        hipLaunchKernelGGL(attn_bwd_dgamma_kernel, dim3(1), dim3(256), 0, st, (const float*)rowdot, pb, dgamma);
        GCC_CHECK_LAUNCH();
    }
    const dim3 gq(cdiv(N, 64), B), gk(cdiv(N, 64), cdiv(C, 64), B);
    const int cs = cdiv(C, 32);
#define GCC_ATTN_BWD(DKS, CS)                                                              \
    do {                                                                                   \
        hipLaunchKernelGGL((attn_bwd_q_kernel<DKS, CS>), gq, dim3(256), 0, st, a);         \
        hipLaunchKernelGGL((attn_bwd_k_kernel<DKS, CS>), gk, dim3(256), 0, st, a);         \
    } while (0)
    if (C8 <= 32) {
        if (cs <= 2) GCC_ATTN_BWD(1, 2);
        else if (cs <= 4) GCC_ATTN_BWD(1, 4);
        else if (cs <= 8) GCC_ATTN_BWD(1, 8);
        else GCC_ATTN_BWD(1, 16);
    } else {
        if (cs <= 8) GCC_ATTN_BWD(2, 8);
        else GCC_ATTN_BWD(2, 16);
    }
#undef GCC_ATTN_BWD
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}
