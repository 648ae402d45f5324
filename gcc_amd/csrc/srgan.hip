// SRGAN-specific streaming kernels (reference models/SRGAN.py, models/GANLoss.py:95-145): PReLU with a learnable scalar
// slope (optionally fused with PixelShuffle(2)), 2x2 max pooling of the VGG stack, global average pool + Linear head of
// the discriminator.  All HBM-bound, NHWC bf16, 16-byte accesses.
#include "common.hpp"

namespace {

__device__ __forceinline__ float block_sum256(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// ---- PReLU (+ PixelShuffle(2)) -------------------------------------------------------------------
// r == 1: y[p][c] = prelu(x[p][c]).
// r == 2: x is [N][H][W][4C] (conv output), y is [N][2H][2W][C]:  y[n][2h+i][2w+j][c] = prelu(x[n][h][w][4c + 2i + j])
//         (nn.PixelShuffle(2) then nn.PReLU(): the slope is one scalar, so the order does not matter).
// One thread handles 8 output channels of one input pixel (for r == 2: 32 input channels -> 4 output pixels).
struct PreluArgs {
    const bf16_t* x; int ldx; bf16_t* y; int ldy; const float* slope; int C; int N, H, W, r;
    const bf16_t* dy; int lddy; bf16_t* dx; int lddx; float* dslope;
    float* partial; unsigned* counter;        // backward with a workspace: per-block partial sums of dslope, arrival counter
};

template <bool BWD>
__global__ __launch_bounds__(256) void prelu_kernel(const PreluArgs a) {
    __shared__ float sh[4];
    const float s = a.slope[0];
    const int CH = (a.C + 7) / 8;
    const size_t pixels = (size_t)a.N * a.H * a.W;
    const size_t total = pixels * CH;
    float ds = 0.f;
    if (BWD && a.r == 1) {
        // the backward of the plain form: two items per trip, all four loads issued before any is used (the launch is capped at
        // PRELU_BWD_BLOCKS workgroups -- every one of them ends in an atomic on ONE counter, ~10 ns each: 4096 of them were 40 of the
        // 56 us this kernel took on SRGAN's 19 MB trunk tensors -- so a thread walks several items)
        const size_t step = (size_t)gridDim.x * 256;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += 2 * step) {
            const size_t i1 = i + step;
            const bool has1 = i1 < total;
            const size_t pix0 = i / CH, pix1 = has1 ? i1 / CH : pix0;
            const int c00 = (int)(i - pix0 * CH) * 8, c01 = has1 ? (int)(i1 - pix1 * CH) * 8 : c00;
            const i32x4 rx0 = *(const i32x4*)(a.x + pix0 * a.ldx + c00), rg0 = *(const i32x4*)(a.dy + pix0 * a.lddy + c00);
            const i32x4 rx1 = *(const i32x4*)(a.x + pix1 * a.ldx + c01), rg1 = *(const i32x4*)(a.dy + pix1 * a.lddy + c01);
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (u == 1 && !has1) break;
                float xv[8], gv[8], ov[8];
                unpack8(u ? rx1 : rx0, xv);
                unpack8(u ? rg1 : rg0, gv);
                const int c0 = u ? c01 : c00;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const bool live = c0 + j < a.C;
                    ov[j] = live ? (xv[j] > 0.f ? gv[j] : s * gv[j]) : 0.f;
                    if (live && xv[j] <= 0.f) ds += gv[j] * xv[j];
                }
                *(i32x4*)(a.dx + (u ? pix1 : pix0) * a.lddx + c0) = pack8(ov);
            }
        }
    } else
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t pix = i / CH;
        const int c0 = (int)(i - pix * CH) * 8;
        if (a.r == 1) {
            float xv[8], ov[8];
            unpack8(*(const i32x4*)(a.x + pix * a.ldx + c0), xv);
            if (!BWD) {
#pragma unroll
                for (int j = 0; j < 8; j++) ov[j] = (c0 + j < a.C) ? (xv[j] > 0.f ? xv[j] : s * xv[j]) : 0.f;
                *(i32x4*)(a.y + pix * a.ldy + c0) = pack8(ov);
            } else {
                float gv[8];
                unpack8(*(const i32x4*)(a.dy + pix * a.lddy + c0), gv);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const bool live = c0 + j < a.C;
                    ov[j] = live ? (xv[j] > 0.f ? gv[j] : s * gv[j]) : 0.f;
                    if (live && xv[j] <= 0.f) ds += gv[j] * xv[j];
                }
                *(i32x4*)(a.dx + pix * a.lddx + c0) = pack8(ov);
            }
        } else {
            const int w = (int)(pix % a.W);
            const size_t nh = pix / a.W;
            const int h = (int)(nh % a.H);
            const size_t n = nh / a.H;
            // 32 input channels 4*c0 .. 4*c0+31
            float xv[32];
#pragma unroll
            for (int q = 0; q < 4; q++) unpack8(*(const i32x4*)(a.x + pix * a.ldx + 4 * c0 + 8 * q), xv + 8 * q);
            float dxv[32];
#pragma unroll
            for (int sp = 0; sp < 4; sp++) {            // sub-pixel (i, j) = (sp >> 1, sp & 1)
                const size_t opix = (n * (2 * a.H) + 2 * h + (sp >> 1)) * (size_t)(2 * a.W) + 2 * w + (sp & 1);
                if (!BWD) {
                    float ov[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const float v = xv[4 * j + sp];
                        ov[j] = (c0 + j < a.C) ? (v > 0.f ? v : s * v) : 0.f;
                    }
                    *(i32x4*)(a.y + opix * a.ldy + c0) = pack8(ov);
                } else {
                    float gv[8];
                    unpack8(*(const i32x4*)(a.dy + opix * a.lddy + c0), gv);
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const float v = xv[4 * j + sp];
                        const bool live = c0 + j < a.C;
                        dxv[4 * j + sp] = live ? (v > 0.f ? gv[j] : s * gv[j]) : 0.f;
                        if (live && v <= 0.f) ds += gv[j] * v;
                    }
                }
            }
            if (BWD) {
#pragma unroll
                for (int q = 0; q < 4; q++) *(i32x4*)(a.dx + pix * a.lddx + 4 * c0 + 8 * q) = pack8(dxv + 8 * q);
            }
        }
    }
    if (BWD && a.dslope) {
        const float t = block_sum256(ds, sh);
        if (!a.partial) {                               // no workspace: the sum's order is the blocks' arrival order
            if (threadIdx.x == 0) atomicAdd(a.dslope, t);
            return;
        }
        // Deterministic form (round 3: an eager and a replayed iteration must agree to the bit, and so must two runs): every
        // block leaves its partial (write-through store), the last one to arrive folds them in index order and adds the total
        // to the gradient -- one block, one read-modify-write.  The counter is zero before and after the launch.
        typedef __attribute__((address_space(1))) unsigned int gu32;
        gu32* cnt = (gu32*)a.counter;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.partial, 0, gridDim.x * 4, 0x00020000);
        // arrivals in two levels (an atomic on one address costs ~10 ns per arrival whatever the workgroup count: 1024 arrivals on
        // one word were 10 of this kernel's 24 us): 32 counters, one per blockIdx.x & 31, and the last arrival of each moves on
        // to the top word.  Words 0 (top) and 1..32 of the workspace head; all of them zero before and after the launch.
        if (threadIdx.x == 0) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(t), rs, blockIdx.x * 4, 0, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned sub = blockIdx.x & 31u;
            const unsigned nsub = (gridDim.x - sub + 31u) / 32u, ntop = gridDim.x < 32u ? gridDim.x : 32u;
            unsigned lastf = 0;
            if (__hip_atomic_fetch_add(cnt + 1 + sub, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsub - 1) {
                __hip_atomic_store(cnt + 1 + sub, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lastf = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ntop - 1 ? 1u : 0u;
            }
            sh[0] = __uint_as_float(lastf);
        }
        __syncthreads();
        const bool last = __float_as_uint(sh[0]) == 1u;
        __syncthreads();
        if (!last) return;
        float acc = 0.f;
        for (unsigned b = threadIdx.x; b < gridDim.x; b += 256)
            acc += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, b * 4, 0, 16));
        const float total = block_sum256(acc, sh);
        if (threadIdx.x == 0) {
            a.dslope[0] += total;
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- MaxPool2d(2, 2) ------------------------------------------------------------------------------
// backward routes dy to the first maximum of the window in (h, w) scan order, as ATen's max_pool2d_with_indices does
// relu_mask (backward): x is the output of a ReLU -- the gradient that reaches a position whose x is not positive is dropped here, so
// that the ReLU's own backward pass over the (four times larger) input map is not needed
template <bool BWD>
__global__ __launch_bounds__(256) void maxpool_kernel(const bf16_t* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy,
                                                      const bf16_t* __restrict__ dy, int lddy, bf16_t* __restrict__ dx, int lddx,
                                                      int N, int Ho, int Wo, int C, int relu_mask) {
    const int CH = (C + 7) / 8;
    const size_t total = (size_t)N * Ho * Wo * CH;
    const int Hi = 2 * Ho, Wi = 2 * Wo;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t opix = i / CH;
        const int c0 = (int)(i - opix * CH) * 8;
        const int wo = (int)(opix % Wo);
        const size_t t = opix / Wo;
        const int ho = (int)(t % Ho);
        const size_t n = t / Ho;
        float v[4][8];
        size_t ip[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            ip[q] = (n * Hi + 2 * ho + (q >> 1)) * (size_t)Wi + 2 * wo + (q & 1);
            unpack8(*(const i32x4*)(x + ip[q] * ldx + c0), v[q]);
        }
        if (!BWD) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = fmaxf(fmaxf(v[0][j], v[1][j]), fmaxf(v[2][j], v[3][j]));
            *(i32x4*)(y + opix * ldy + c0) = pack8(o);
        } else {
            float g[8], o[4][8];
            unpack8(*(const i32x4*)(dy + opix * lddy + c0), g);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                int best = 0;
#pragma unroll
                for (int q = 1; q < 4; q++)
                    if (v[q][j] > v[best][j]) best = q;
                const float gj = (relu_mask && !(v[best][j] > 0.f)) ? 0.f : g[j];
#pragma unroll
                for (int q = 0; q < 4; q++) o[q][j] = (q == best) ? gj : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; q++) *(i32x4*)(dx + ip[q] * lddx + c0) = pack8(o[q]);
        }
    }
}

// ---- AdaptiveAvgPool2d((1,1)) + Linear(C, 1) -------------------------------------------------------
// one block per (image, 64-channel group): thread -> (8-channel chunk = tid & 7, pixel lane = tid >> 3): 16-byte loads, eight neighbouring
// threads read 128 contiguous bytes of a pixel.  (Round 5: the first form -- one block per channel, 2-byte loads at the pixel stride --
// took 198 us for the 24 x 24 x 512 map of the 96 -> 384 discriminators: 77 launches, 2 % of that step.)
__global__ __launch_bounds__(256) void pool_mean_kernel(const bf16_t* __restrict__ x, int ldx, int C, int HW, float* __restrict__ pooled) {
    __shared__ float sh[32][65];
    const int n = blockIdx.y, c0 = blockIdx.x * 64 + (threadIdx.x & 7) * 8, pl = threadIdx.x >> 3;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < C) {                                        // ldx is a multiple of 8 and >= ceil8(C): the 16 bytes are inside the pixel's row
        for (int p = pl; p < HW; p += 32) {
            float f[8];
            unpack8(*(const i32x4*)(x + ((size_t)n * HW + p) * ldx + c0), f);
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] += f[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) sh[pl][(threadIdx.x & 7) * 8 + j] = acc[j];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = blockIdx.x * 64 + threadIdx.x;
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 32; q++) t += sh[q][threadIdx.x];
        if (c < C) pooled[(size_t)n * C + c] = t / (float)HW;
    }
}
__global__ __launch_bounds__(256) void linear_head_kernel(const float* __restrict__ pooled, const float* __restrict__ w, const float* __restrict__ b,
                                                          int C, bf16_t* __restrict__ logit, int ldl) {
    __shared__ float sh[4];
    const int n = blockIdx.x;
    float acc = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) acc += pooled[(size_t)n * C + c] * w[c];
    const float t = block_sum256(acc, sh);
    if (threadIdx.x == 0) logit[(size_t)n * ldl] = f2bf(t + b[0]);
}
// dx[n][p][c] = dlogit[n] * w[c] / HW
__global__ __launch_bounds__(256) void pool_head_bwd_kernel(const bf16_t* __restrict__ dlogit, int ldl, const float* __restrict__ w, int C, int HW,
                                                            int N, bf16_t* __restrict__ dx, int lddx) {
    const int CH = (C + 7) / 8;
    const size_t total = (size_t)N * HW * CH;
    const float inv = 1.f / (float)HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t pix = i / CH;
        const int c0 = (int)(i - pix * CH) * 8;
        const float g = bf2f(dlogit[(pix / HW) * ldl]) * inv;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = (c0 + j < C) ? g * w[c0 + j] : 0.f;
        *(i32x4*)(dx + pix * lddx + c0) = pack8(o);
    }
}
// dw[c] += sum_n dlogit[n] * pooled[n][c] ; db += sum_n dlogit[n]
__global__ __launch_bounds__(256) void linear_head_wgrad_kernel(const bf16_t* __restrict__ dlogit, int ldl, const float* __restrict__ pooled, int N,
                                                                int C, float* __restrict__ dw, float* __restrict__ db) {
    for (int c = blockIdx.x * 256 + threadIdx.x; c < C; c += gridDim.x * 256) {
        float acc = 0.f;
        for (int n = 0; n < N; n++) acc += bf2f(dlogit[(size_t)n * ldl]) * pooled[(size_t)n * C + c];
        dw[c] += acc;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && db) {
        float acc = 0.f;
        for (int n = 0; n < N; n++) acc += bf2f(dlogit[(size_t)n * ldl]);
        db[0] += acc;
    }
}

constexpr int PRELU_BWD_BLOCKS = 1024;
int grid_for(size_t items, int cap = 4096) {
    size_t b = (items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > (size_t)cap ? cap : b));
}

}  // namespace

extern "C" int gcc_prelu(int backward, const void* x, int ldx, const float* slope, int C, int N, int H, int W, int shuffle,
                         void* y, int ldy, const void* dy, int lddy, void* dx, int lddx, float* dslope, void* workspace,
                         size_t workspace_bytes, gcc_stream_t stream) {
    GCC_ENTER();
    if (!x || !slope || C <= 0 || N <= 0 || H <= 0 || W <= 0 || (shuffle != 1 && shuffle != 2)) return GCC_ERR_BAD_ARG;
    if (backward ? (!dy || !dx) : !y) return GCC_ERR_BAD_ARG;
    if ((ldx & 7) || (!backward && (ldy & 7)) || (backward && ((lddy | lddx) & 7))) return GCC_ERR_BAD_ARG;
    if (shuffle == 2 && (C & 7)) return GCC_ERR_UNSUPPORTED;          // 4C input channels are read as whole 32-channel groups
    PreluArgs a;
    a.x = (const bf16_t*)x; a.ldx = ldx; a.y = (bf16_t*)y; a.ldy = ldy; a.slope = slope; a.C = C; a.N = N; a.H = H; a.W = W;
    a.r = shuffle; a.dy = (const bf16_t*)dy; a.lddy = lddy; a.dx = (bf16_t*)dx; a.lddx = lddx; a.dslope = dslope;
    const size_t items = (size_t)N * H * W * ((C + 7) / 8);
    a.partial = nullptr; a.counter = nullptr;
    if (backward && dslope && workspace && workspace_bytes >= 256 + 4 * (size_t)grid_for(items)) {
        a.counter = (unsigned*)workspace;
        a.partial = (float*)((char*)workspace + 256);
    }
    if (backward) hipLaunchKernelGGL(prelu_kernel<true>, dim3(grid_for(items, shuffle == 1 ? PRELU_BWD_BLOCKS : 4096)), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(prelu_kernel<false>, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, a);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

extern "C" int gcc_maxpool2x2(int backward, const void* x, int ldx, void* y, int ldy, const void* dy, int lddy, void* dx,
                              int lddx, int N, int Ho, int Wo, int C, gcc_stream_t stream) {
    GCC_ENTER();
    if (!x || N <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || (ldx & 7)) return GCC_ERR_BAD_ARG;
    if (backward ? (!dy || !dx || ((lddy | lddx) & 7)) : (!y || (ldy & 7))) return GCC_ERR_BAD_ARG;
    const size_t items = (size_t)N * Ho * Wo * ((C + 7) / 8);
    if (backward)
        hipLaunchKernelGGL(maxpool_kernel<true>, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx,
                           (bf16_t*)nullptr, 0, (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, N, Ho, Wo, C, backward == 2 ? 1 : 0);
    else
        hipLaunchKernelGGL(maxpool_kernel<false>, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx,
                           (bf16_t*)y, ldy, (const bf16_t*)nullptr, 0, (bf16_t*)nullptr, 0, N, Ho, Wo, C, 0);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

extern "C" int gcc_pool_linear_fwd(const void* x, int ldx, int N, int HW, int C, const float* w, const float* b, float* pooled,
                                   void* logit, int ldl, gcc_stream_t stream) {
    GCC_ENTER();
    if (!x || !w || !b || !pooled || !logit || N <= 0 || HW <= 0 || C <= 0 || (ldx & 7)) return GCC_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pool_mean_kernel, dim3((C + 63) / 64, N), dim3(256), 0, st, (const bf16_t*)x, ldx, C, HW, pooled);
    GCC_CHECK_LAUNCH();
    hipLaunchKernelGGL(linear_head_kernel, dim3(N), dim3(256), 0, st, (const float*)pooled, w, b, C, (bf16_t*)logit, ldl);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

extern "C" int gcc_pool_linear_bwd(const void* dlogit, int ldl, const float* w, const float* pooled, int N, int HW, int C,
                                   void* dx, int lddx, float* dw, float* db, gcc_stream_t stream) {
    GCC_ENTER();
    if (!dlogit || !w || !pooled || N <= 0 || HW <= 0 || C <= 0) return GCC_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dx) {
        if (lddx & 7) return GCC_ERR_BAD_ARG;
        hipLaunchKernelGGL(pool_head_bwd_kernel, dim3(grid_for((size_t)N * HW * ((C + 7) / 8))), dim3(256), 0, st,
                           (const bf16_t*)dlogit, ldl, w, C, HW, N, (bf16_t*)dx, lddx);
        GCC_CHECK_LAUNCH();
    }
    if (dw) {
        hipLaunchKernelGGL(linear_head_wgrad_kernel, dim3((C + 255) / 256), dim3(256), 0, st, (const bf16_t*)dlogit, ldl, pooled, N,
                           C, dw, db);
        GCC_CHECK_LAUNCH();
    }
    return GCC_OK;
}
