// Self attention over the H*W positions of one image (reference Self_Attn, models/SAGAN.py:72-104):
//     energy[i][j] = q_i . k_j ;  A = softmax_j(energy) ;  o_i = sum_j A[i][j] v_j ;  y = gamma * o + x
// q, k (C8 = C/8 channels) and v (C channels) are channel slices of one NHWC bf16 buffer written by the three 1x1 convs.
// N = H*W <= 1024 (SAGAN: 16..1024), C8 <= 64, C <= 512: a few MFLOP per image -- the kernels are written for HBM / latency,
// not for the matrix cores: one workgroup owns TQ = 8 query (or key) rows, keeps their score rows in LDS, streams k / v
// rows (L2 resident: N * C * 2 bytes per image) and saves A in fp32 for the backward pass.
//
// backward, with do = gamma * dy:
//     dgamma += sum dy * o ;  dV_j = sum_i A[i][j] do_i ;  dA[i][j] = do_i . v_j ;
//     dS[i][j] = A[i][j] * (dA[i][j] - sum_j' A[i][j'] dA[i][j']) ;  dq_i = sum_j dS[i][j] k_j ;  dk_j = sum_i dS[i][j] q_i
// (the residual branch dx += dy is the caller's: it owns the buffer the three 1x1 data gradients are added into).
#include "common.hpp"

namespace {

constexpr int TQ = 8;
constexpr int NMAX = 1024;
constexpr int CMAX = 512;
constexpr int C8MAX = 64;

struct AttnArgs {
    const bf16_t* qkv; int ldq, qoff, koff, voff;
    const bf16_t* x; int ldx;
    bf16_t* y; int ldy;
    bf16_t* o; int ldo;          // pre-gamma attention output, [B][N][ldo]
    float* A;                    // [B][N][N]
    const float* gamma;
    int N, C, C8;
    // backward
    const bf16_t* dy; int lddy;
    float* dS;                   // [B][N][N]
    bf16_t* dqkv; int lddq;      // same slice offsets as qkv
    float* dgamma;
};

__device__ __forceinline__ float block_sum256(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs a) {
    __shared__ float S[TQ][NMAX];
    __shared__ float Q[TQ][C8MAX];
    __shared__ float O[TQ][CMAX];
    const int b = blockIdx.y, i0 = blockIdx.x * TQ, tid = threadIdx.x;
    const int N = a.N, C = a.C, C8 = a.C8;
    const bf16_t* base = a.qkv + (size_t)b * N * a.ldq;
    for (int e = tid; e < TQ * C8; e += 256) {
        const int i = e / C8, c = e - i * C8;
        Q[i][c] = (i0 + i < N) ? bf2f(base[(size_t)(i0 + i) * a.ldq + a.qoff + c]) : 0.f;
    }
    for (int e = tid; e < TQ * C; e += 256) O[e / C][e % C] = 0.f;
    __syncthreads();
    // scores
    for (int j = tid; j < N; j += 256) {
        const bf16_t* kr = base + (size_t)j * a.ldq + a.koff;
        float acc[TQ];
#pragma unroll
        for (int i = 0; i < TQ; i++) acc[i] = 0.f;
        for (int c = 0; c < C8; c++) {
            const float kv = bf2f(kr[c]);
#pragma unroll
            for (int i = 0; i < TQ; i++) acc[i] += Q[i][c] * kv;
        }
#pragma unroll
        for (int i = 0; i < TQ; i++) S[i][j] = acc[i];
    }
    __syncthreads();
    // softmax: wave w handles rows w and w + 4
    const int wv = tid >> 6, lane = tid & 63;
    for (int i = wv; i < TQ; i += 4) {
        float m = -3.0e38f;
        for (int j = lane; j < N; j += 64) m = fmaxf(m, S[i][j]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float s = 0.f;
        for (int j = lane; j < N; j += 64) {
            const float e = __expf(S[i][j] - m);
            S[i][j] = e;
            s += e;
        }
        s = wave_sum(s);
        const float inv = 1.f / s;
        float* Arow = a.A + ((size_t)b * N + (i0 + i)) * N;
        for (int j = lane; j < N; j += 64) {
            const float p = S[i][j] * inv;
            S[i][j] = p;
            if (i0 + i < N) Arow[j] = p;
        }
    }
    __syncthreads();
    // o = A v: thread = (channel chunk of 8, key group)
    const int nch = (C + 7) / 8;
    const int ngrp = 256 / nch > 0 ? 256 / nch : 1;
    if (tid < nch * ngrp) {
        const int ch = tid % nch, grp = tid / nch;
        float acc[TQ][8];
#pragma unroll
        for (int i = 0; i < TQ; i++)
#pragma unroll
            for (int c = 0; c < 8; c++) acc[i][c] = 0.f;
        for (int j = grp; j < N; j += ngrp) {
            float vv[8];
            unpack8(*(const i32x4*)(base + (size_t)j * a.ldq + a.voff + ch * 8), vv);
#pragma unroll
            for (int i = 0; i < TQ; i++) {
                const float p = S[i][j];
#pragma unroll
                for (int c = 0; c < 8; c++) acc[i][c] += p * vv[c];
            }
        }
#pragma unroll
        for (int i = 0; i < TQ; i++)
#pragma unroll
            for (int c = 0; c < 8; c++)
                if (ch * 8 + c < C) atomicAdd(&O[i][ch * 8 + c], acc[i][c]);
    }
    __syncthreads();
    const float g = a.gamma[0];
    for (int e = tid; e < TQ * nch; e += 256) {
        const int i = e / nch, ch = e - i * nch;
        if (i0 + i >= N) continue;
        const size_t pix = (size_t)b * N + i0 + i;
        float xv[8], ov[8], yv[8];
        unpack8(*(const i32x4*)(a.x + pix * a.ldx + ch * 8), xv);
#pragma unroll
        for (int c = 0; c < 8; c++) {
            ov[c] = (ch * 8 + c < C) ? O[i][ch * 8 + c] : 0.f;
        }
        const i32x4 ob = pack8(ov);
        unpack8(ob, ov);                           // y is computed from the bf16-stored o, as backward will read it
#pragma unroll
        for (int c = 0; c < 8; c++) yv[c] = (ch * 8 + c < C) ? g * ov[c] + xv[c] : 0.f;
        *(i32x4*)(a.o + pix * a.ldo + ch * 8) = ob;
        *(i32x4*)(a.y + pix * a.ldy + ch * 8) = pack8(yv);
    }
}

// ---------------------------------------------------------------------------------------------
// per query block: dgamma, dA, dS (-> global), dq
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const AttnArgs a) {
    __shared__ float S[TQ][NMAX];          // dA then dS
    __shared__ float DO[TQ][CMAX];         // gamma * dy
    __shared__ float red[4];
    __shared__ float rowdot[TQ];
    const int b = blockIdx.y, i0 = blockIdx.x * TQ, tid = threadIdx.x;
    const int N = a.N, C = a.C, C8 = a.C8;
    const bf16_t* base = a.qkv + (size_t)b * N * a.ldq;
    const float g = a.gamma[0];
    float dg = 0.f;
    for (int e = tid; e < TQ * C; e += 256) {
        const int i = e / C, c = e - i * C;
        float d = 0.f;
        if (i0 + i < N) {
            const size_t pix = (size_t)b * N + i0 + i;
            d = bf2f(a.dy[pix * a.lddy + c]);
            dg += d * bf2f(a.o[pix * a.ldo + c]);
        }
        DO[i][c] = g * d;
    }
    const float dgs = block_sum256(dg, red);
    if (tid == 0 && a.dgamma) atomicAdd(a.dgamma, dgs);
    __syncthreads();
    // dA[i][j] = do_i . v_j
    for (int j = tid; j < N; j += 256) {
        const bf16_t* vr = base + (size_t)j * a.ldq + a.voff;
        float acc[TQ];
#pragma unroll
        for (int i = 0; i < TQ; i++) acc[i] = 0.f;
        for (int c0 = 0; c0 < C; c0 += 8) {
            float vv[8];
            unpack8(*(const i32x4*)(vr + c0), vv);
#pragma unroll
            for (int c = 0; c < 8; c++) {
                if (c0 + c < C) {
#pragma unroll
                    for (int i = 0; i < TQ; i++) acc[i] += DO[i][c0 + c] * vv[c];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TQ; i++) S[i][j] = acc[i];
    }
    __syncthreads();
    const int wv = tid >> 6, lane = tid & 63;
    for (int i = wv; i < TQ; i += 4) {
        const float* Arow = a.A + ((size_t)b * N + (i0 + i < N ? i0 + i : 0)) * N;
        float s = 0.f;
        for (int j = lane; j < N; j += 64) s += Arow[j] * S[i][j];
        s = wave_sum(s);
        float* dSrow = a.dS + ((size_t)b * N + (i0 + i < N ? i0 + i : 0)) * N;
        for (int j = lane; j < N; j += 64) {
            const float d = (i0 + i < N) ? Arow[j] * (S[i][j] - s) : 0.f;
            S[i][j] = d;
            if (i0 + i < N) dSrow[j] = d;
        }
    }
    __syncthreads();
    // dq[i][c] = sum_j dS[i][j] k[j][c]: thread = (i, c), 8 * C8 <= 512 items
    for (int e = tid; e < TQ * ceil8(C8); e += 256) {
        const int i = e / ceil8(C8), c = e - i * ceil8(C8);
        if (i0 + i >= N) continue;
        float acc = 0.f;
        if (c < C8)
            for (int j = 0; j < N; j++) acc += S[i][j] * bf2f(base[(size_t)j * a.ldq + a.koff + c]);
        a.dqkv[((size_t)b * N + i0 + i) * a.lddq + a.qoff + c] = f2bf(acc);
    }
}

// per key block: dv, dk
__global__ __launch_bounds__(256) void attn_bwd_k_kernel(const AttnArgs a) {
    __shared__ float DV[TQ][CMAX];
    __shared__ float DK[TQ][C8MAX];
    const int b = blockIdx.y, j0 = blockIdx.x * TQ, tid = threadIdx.x;
    const int N = a.N, C = a.C, C8 = a.C8;
    const bf16_t* base = a.qkv + (size_t)b * N * a.ldq;
    const float g = a.gamma[0];
    for (int e = tid; e < TQ * C; e += 256) DV[e / C][e % C] = 0.f;
    for (int e = tid; e < TQ * C8MAX; e += 256) DK[e / C8MAX][e % C8MAX] = 0.f;
    __syncthreads();
    const int nch = (C + 7) / 8;
    const int ngrp = 256 / nch > 0 ? 256 / nch : 1;
    if (tid < nch * ngrp) {
        const int ch = tid % nch, grp = tid / nch;
        float acc[TQ][8];
#pragma unroll
        for (int jj = 0; jj < TQ; jj++)
#pragma unroll
            for (int c = 0; c < 8; c++) acc[jj][c] = 0.f;
        for (int i = grp; i < N; i += ngrp) {
            const size_t pix = (size_t)b * N + i;
            float dv[8];
            unpack8(*(const i32x4*)(a.dy + pix * a.lddy + ch * 8), dv);
            const float* Arow = a.A + pix * N + j0;
#pragma unroll
            for (int jj = 0; jj < TQ; jj++) {
                const float p = (j0 + jj < N) ? Arow[jj] : 0.f;
#pragma unroll
                for (int c = 0; c < 8; c++) acc[jj][c] += p * dv[c];
            }
        }
#pragma unroll
        for (int jj = 0; jj < TQ; jj++)
#pragma unroll
            for (int c = 0; c < 8; c++)
                if (ch * 8 + c < C) atomicAdd(&DV[jj][ch * 8 + c], acc[jj][c]);
    }
    // dk[j][c] = sum_i dS[i][j] q[i][c]: thread = (c, query group)
    {
        const int c = tid % C8MAX, grp = tid / C8MAX, ng = 256 / C8MAX;      // 64 channels x 4 groups
        if (c < C8) {
            float acc[TQ];
#pragma unroll
            for (int jj = 0; jj < TQ; jj++) acc[jj] = 0.f;
            for (int i = grp; i < N; i += ng) {
                const float qv = bf2f(base[(size_t)i * a.ldq + a.qoff + c]);
                const float* dSrow = a.dS + ((size_t)b * N + i) * N + j0;
#pragma unroll
                for (int jj = 0; jj < TQ; jj++) acc[jj] += ((j0 + jj < N) ? dSrow[jj] : 0.f) * qv;
            }
#pragma unroll
            for (int jj = 0; jj < TQ; jj++) atomicAdd(&DK[jj][c], acc[jj]);
        }
    }
    __syncthreads();
    for (int e = tid; e < TQ * nch; e += 256) {
        const int jj = e / nch, ch = e - jj * nch;
        if (j0 + jj >= N) continue;
        float ov[8];
#pragma unroll
        for (int c = 0; c < 8; c++) ov[c] = (ch * 8 + c < C) ? g * DV[jj][ch * 8 + c] : 0.f;
        *(i32x4*)(a.dqkv + ((size_t)b * N + j0 + jj) * a.lddq + a.voff + ch * 8) = pack8(ov);
    }
    for (int e = tid; e < TQ * ceil8(C8); e += 256) {
        const int jj = e / ceil8(C8), c = e - jj * ceil8(C8);
        if (j0 + jj >= N) continue;
        a.dqkv[((size_t)b * N + j0 + jj) * a.lddq + a.koff + c] = f2bf(c < C8 ? DK[jj][c] : 0.f);
    }
}

bool ok_geom(int B, int N, int C, int C8) { return B > 0 && N > 0 && N <= NMAX && C > 0 && C <= CMAX && C8 > 0 && C8 <= C8MAX; }

}  // namespace

extern "C" int gcc_attention_fwd(const void* qkv, int ldq, int qoff, int koff, int voff, const void* x, int ldx,
                                 const float* gamma, int B, int N, int C, int C8, void* y, int ldy, void* o, int ldo,
                                 float* A, gcc_stream_t stream) {
    GCC_ENTER();
    if (!qkv || !x || !gamma || !y || !o || !A) return GCC_ERR_BAD_ARG;
    if (!ok_geom(B, N, C, C8)) return GCC_ERR_UNSUPPORTED;
    if ((ldq | qoff | koff | voff | ldx | ldy | ldo) & 7) return GCC_ERR_BAD_ARG;
    AttnArgs a = {};
    a.qkv = (const bf16_t*)qkv; a.ldq = ldq; a.qoff = qoff; a.koff = koff; a.voff = voff;
    a.x = (const bf16_t*)x; a.ldx = ldx; a.y = (bf16_t*)y; a.ldy = ldy; a.o = (bf16_t*)o; a.ldo = ldo; a.A = A;
    a.gamma = gamma; a.N = N; a.C = C; a.C8 = C8;
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(cdiv(N, TQ), B), dim3(256), 0, (hipStream_t)stream, a);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

extern "C" int gcc_attention_bwd(const void* qkv, int ldq, int qoff, int koff, int voff, const void* o, int ldo,
                                 const float* A, const float* gamma, const void* dy, int lddy, int B, int N, int C, int C8,
                                 void* dqkv, int lddq, float* dS, float* dgamma, gcc_stream_t stream) {
    GCC_ENTER();
    if (!qkv || !o || !A || !gamma || !dy || !dqkv || !dS) return GCC_ERR_BAD_ARG;
    if (!ok_geom(B, N, C, C8)) return GCC_ERR_UNSUPPORTED;
    if ((ldq | qoff | koff | voff | ldo | lddy | lddq) & 7) return GCC_ERR_BAD_ARG;
    AttnArgs a = {};
    a.qkv = (const bf16_t*)qkv; a.ldq = ldq; a.qoff = qoff; a.koff = koff; a.voff = voff;
    a.o = (bf16_t*)o; a.ldo = ldo; a.A = (float*)A; a.gamma = gamma; a.N = N; a.C = C; a.C8 = C8;
    a.dy = (const bf16_t*)dy; a.lddy = lddy; a.dS = dS; a.dqkv = (bf16_t*)dqkv; a.lddq = lddq; a.dgamma = dgamma;
    hipLaunchKernelGGL(attn_bwd_q_kernel, dim3(cdiv(N, TQ), B), dim3(256), 0, (hipStream_t)stream, a);
    GCC_CHECK_LAUNCH();
    hipLaunchKernelGGL(attn_bwd_k_kernel, dim3(cdiv(N, TQ), B), dim3(256), 0, (hipStream_t)stream, a);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}
