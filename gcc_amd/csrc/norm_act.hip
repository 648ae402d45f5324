// BatchNorm2d (train/eval) + activation + DifferentiableOP gate + dropout, forward and backward,
// on NHWC bf16 tensors.  All of these are HBM-bound streaming kernels: 16-byte (8-channel) accesses,
// one thread per 8-channel chunk, per-channel reductions kept in registers across a grid-stride
// loop, folded through LDS once per block and finished by a tiny second kernel (deterministic, no
// float atomics).
#include "common.hpp"
#include <stdlib.h>
#include <algorithm>
#include <mutex>
#include <unordered_map>

namespace {

// thread layout shared by the streaming kernels: CHP = chunks per pixel rounded up to a power of
// two (<= 256); a 256-thread block covers 256/CHP pixels per sweep.
struct Layout {
    int CH;      // 16-B chunks per pixel = ceil8(C)/8
    int CHP;     // power of two >= CH
    int sh;      // log2(CHP)
    int PPB;     // pixels per block sweep
};
static bool make_layout(int C, Layout* L, int V = 8) {
    L->CH = ((C + 7) / 8) * (8 / V);      // V-channel chunks per pixel (storage is padded to 8 channels)
    if (L->CH > 256) return false;
    L->CHP = 1; L->sh = 0;
    while (L->CHP < L->CH) { L->CHP <<= 1; L->sh++; }
    L->PPB = 256 / L->CHP;
    return true;
}
static int stream_blocks(size_t pixels, const Layout& L, int sweeps_per_block) {
    const int sw = gcc_opt(GCC_OPT_BN_SWEEPS), cap = gcc_opt(GCC_OPT_BN_MAXBLK);
    if (sw > 0) sweeps_per_block = sw;
    size_t b = (pixels + (size_t)L.PPB * sweeps_per_block - 1) / ((size_t)L.PPB * sweeps_per_block);
    if (b < 1) b = 1;
    if (b > (size_t)cap) b = cap;
    return (int)b;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ part, int tiles, int C, double count,
                                                           const float* gamma, const float* beta, float eps, float momentum,
                                                           float* rmean, float* rvar, float* mean, float* rstd,
                                                           float* scale, float* shift) {
    // the canonical fold (common.hpp, FIN_GROUP): 32 channels x 32 group lanes; lane pl sums the rows of group g0 + pl, the
    // channel's first lane adds the 32 group sums in ascending order, chunk of 32 groups after chunk
    __shared__ double sh[2][32][33];
    const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    {   // group (InstanceNorm: one per image) on blockIdx.y
        const size_t g = blockIdx.y;
        part += g * (size_t)tiles * 2 * C;
        if (mean) mean += g * C;
        if (rstd) rstd += g * C;
        scale += g * C; shift += g * C;
    }
    const int G = (tiles + FIN_GROUP - 1) / FIN_GROUP;
    double s = 0.0, ss = 0.0;
    for (int g0 = 0; g0 < G; g0 += 32) {
        const int g = g0 + pl;
        double a = 0.0, b = 0.0;
        if (c < C && g < G) {
            const int r0 = g * FIN_GROUP, nr = min(FIN_GROUP, tiles - r0);
            float va[FIN_GROUP], vb[FIN_GROUP];
#pragma unroll
            for (int r = 0; r < FIN_GROUP; r++) {
                va[r] = r < nr ? part[((size_t)(r0 + r) * 2 + 0) * C + c] : 0.f;
                vb[r] = r < nr ? part[((size_t)(r0 + r) * 2 + 1) * C + c] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < FIN_GROUP; r++)
                if (r < nr) { a += (double)va[r]; b += (double)vb[r]; }
        }
        sh[0][pl][cl] = a; sh[1][pl][cl] = b;
        __syncthreads();
        if (pl == 0) {
            const int nq = min(32, G - g0);
            for (int q = 0; q < nq; q++) { s += sh[0][q][cl]; ss += sh[1][q][cl]; }
        }
        __syncthreads();
    }
    if (pl == 0 && c < C) bn_channel_finalize(s, ss, count, eps, momentum, gamma, beta, c, rmean, rvar, mean, rstd, scale, shift);
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      int C, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float r = 1.f / sqrtf(rv[c] + eps);
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        scale[c] = g * r;
        shift[c] = b - rm[c] * g * r;
    }
}

template <int V> __device__ __forceinline__ void ldv(const bf16_t* p, float* f);
template <> __device__ __forceinline__ void ldv<8>(const bf16_t* p, float* f) { unpack8(*(const i32x4*)p, f); }
template <> __device__ __forceinline__ void ldv<4>(const bf16_t* p, float* f) {
    const i32x2 v = *(const i32x2*)p;
    f[0] = __uint_as_float((uint32_t)v[0] << 16); f[1] = __uint_as_float((uint32_t)v[0] & 0xffff0000u);
    f[2] = __uint_as_float((uint32_t)v[1] << 16); f[3] = __uint_as_float((uint32_t)v[1] & 0xffff0000u);
}
// raw loads kept apart from the unpack: the backward reduce pass prefetches the next pixel's vectors before it works on
// the current ones (two pixels in flight per thread; with one, the pass ran at the memory latency, not the bandwidth)
template <int V> struct RawVec;
template <> struct RawVec<8> { typedef i32x4 type; };
template <> struct RawVec<4> { typedef i32x2 type; };
template <int V> __device__ __forceinline__ typename RawVec<V>::type ldraw(const bf16_t* p) {
    return *(const typename RawVec<V>::type*)p;
}
__device__ __forceinline__ void unraw(const i32x4& v, float* f) { unpack8(v, f); }
__device__ __forceinline__ void unraw(const i32x2& v, float* f) {
    f[0] = __uint_as_float((uint32_t)v[0] << 16); f[1] = __uint_as_float((uint32_t)v[0] & 0xffff0000u);
    f[2] = __uint_as_float((uint32_t)v[1] << 16); f[3] = __uint_as_float((uint32_t)v[1] & 0xffff0000u);
}
template <int V> __device__ __forceinline__ void stv(bf16_t* p, const float* f);
template <> __device__ __forceinline__ void stv<8>(bf16_t* p, const float* f) { *(i32x4*)p = pack8(f); }
template <> __device__ __forceinline__ void stv<4>(bf16_t* p, const float* f) {
    i32x2 v;
    v[0] = (int)pack2bf(f[0], f[1]); v[1] = (int)pack2bf(f[2], f[3]);
    *(i32x2*)p = v;
}

// ------------------------------------------------------------------------------------------------
struct FwdArgs {
    gcc_bnact_t p;
    const bf16_t* x; int ldx, xoff;
    bf16_t* y; int ldy, yoff;
    bf16_t* y2; int ldy2, y2off;
    int C; size_t pixels; Layout L;
    int groups;                       // > 1: InstanceNorm -- blockIdx.y = image, scale/shift are [groups][C]
    const bf16_t* res; int ldres;     // optional residual added to y after the activation
};

__global__ __launch_bounds__(256) void bnact_fwd_kernel(const FwdArgs a) {
    const int ch = threadIdx.x & (a.L.CHP - 1);
    const int pl = threadIdx.x >> a.L.sh;
    if (ch >= a.L.CH) return;
    const int c0 = ch * 8;
    const size_t gi = blockIdx.y;
    const bf16_t* xg = a.x + gi * a.pixels * a.ldx;
    bf16_t* yg = a.y ? a.y + gi * a.pixels * a.ldy : nullptr;
    bf16_t* y2g = a.y2 ? a.y2 + gi * a.pixels * a.ldy2 : nullptr;
    const bf16_t* rg = a.res ? a.res + gi * a.pixels * a.ldres : nullptr;
    const size_t po = a.groups > 1 ? gi * a.C : 0;
    float sc[8], sf[8], gm[8];
    if (c0 + 8 <= a.C && ((po + c0) & 3) == 0) {
        // whole chunk inside the channel range: the per-channel parameters come in as 16-byte loads (this prologue is
        // a dependent global round trip paid by every workgroup before it streams anything)
        const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
        const f32x4 s0 = a.p.scale ? *(const f32x4*)(a.p.scale + po + c0) : one, s1 = a.p.scale ? *(const f32x4*)(a.p.scale + po + c0 + 4) : one;
        const f32x4 t0 = a.p.shift ? *(const f32x4*)(a.p.shift + po + c0) : zero, t1 = a.p.shift ? *(const f32x4*)(a.p.shift + po + c0 + 4) : zero;
        const f32x4 g0 = a.p.gate ? *(const f32x4*)(a.p.gate + c0) : one, g1 = a.p.gate ? *(const f32x4*)(a.p.gate + c0 + 4) : one;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            sc[j] = s0[j]; sc[4 + j] = s1[j]; sf[j] = t0[j]; sf[4 + j] = t1[j]; gm[j] = g0[j]; gm[4 + j] = g1[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int c = c0 + j;
            const bool v = c < a.C;
            sc[j] = (v && a.p.scale) ? a.p.scale[po + c] : 1.f;
            sf[j] = (v && a.p.shift) ? a.p.shift[po + c] : 0.f;
            gm[j] = v ? (a.p.gate ? a.p.gate[c] : 1.f) : 0.f;   // pad channels come out as exact zeros
        }
    }
    const float keep_scale = a.p.drop_p > 0.f ? 1.f / (1.f - a.p.drop_p) : 1.f;
    auto one = [&](size_t pix, const i32x4& raw, const i32x4& rres) {
        float v[8], o1[8], o2[8];
        unpack8(raw, v);
        // stage by stage over the 8 channels, the launch-constant options tested once per stage (not per element)
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = v[j] * sc[j] + sf[j];
        if (a.p.drop_p > 0.f) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float u = rng_uniform(a.p.seed, pix * (size_t)a.C + c0 + j);
                v[j] = u >= a.p.drop_p ? v[j] * keep_scale : 0.f;
            }
        }
        if (!a.p.gate_after_act) {
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] *= gm[j];
            apply_act8(v, o1, a.p.act, a.p.slope);
        } else {
            apply_act8(v, o1, a.p.act, a.p.slope);
#pragma unroll
            for (int j = 0; j < 8; j++) { o1[j] *= gm[j]; v[j] *= gm[j]; }
        }
        if (y2g) apply_act8(v, o2, a.p.act2, a.p.slope);
        if (rg) {
            float rv[8];
            unpack8(rres, rv);
#pragma unroll
            for (int j = 0; j < 8; j++) o1[j] += rv[j];
        }
        if (yg) *(i32x4*)(yg + pix * a.ldy + a.yoff + c0) = pack8(o1);
        if (y2g) *(i32x4*)(y2g + pix * a.ldy2 + a.y2off + c0) = pack8(o2);
    };
    // two pixels per trip: both loads are issued before either is used (the kernel is a pure stream; with one 16-byte load
    // in flight per thread the bytes in flight per CU, not HBM, set its rate)
    const size_t step = (size_t)gridDim.x * a.L.PPB;
    const i32x4 z4 = {0, 0, 0, 0};
    for (size_t pix = (size_t)blockIdx.x * a.L.PPB + pl; pix < a.pixels; pix += 2 * step) {
        const size_t p1 = pix + step;
        const bool has1 = p1 < a.pixels;
        const i32x4 r0 = *(const i32x4*)(xg + pix * a.ldx + a.xoff + c0);
        const i32x4 r1 = has1 ? *(const i32x4*)(xg + p1 * a.ldx + a.xoff + c0) : z4;
        const i32x4 q0 = rg ? *(const i32x4*)(rg + pix * a.ldres + c0) : z4;
        const i32x4 q1 = (rg && has1) ? *(const i32x4*)(rg + p1 * a.ldres + c0) : z4;
        one(pix, r0, q0);
        if (has1) one(p1, r1, q1);
    }
}

// ------------------------------------------------------------------------------------------------
struct BwdArgs {
    gcc_bnact_bwd_t p;
    const bf16_t* x; int ldx, xoff;
    const bf16_t* y; int ldy, yoff;
    const bf16_t* g1; int ldg1, g1off;
    const bf16_t* g2; int ldg2, g2off;
    bf16_t* dx; int lddx, dxoff;
    int C, C8; size_t pixels; Layout L;
    float* partial;   // [blocks][3][C8]
    float* totals;    // [3][C8]
    int in_act;       // activation already applied to x by the producer (conv epilogue): dx *= act'(x)
    float in_slope;
    int groups;       // > 1: InstanceNorm -- blockIdx.y = image; mean/rstd/partial/totals are per group
    int nblocks;      // reduce-pass workgroups per group
    unsigned* tickets;   // NULL: a bnact_bwd_finalize launch.  Else the reduce pass's last-arriving workgroups finalize (bwd_tail):
    double* grp;         // [groups of FIN_GROUP rows + 1] ticket words (zero, self-resetting) and [G][3][C8] group sums
};

// per-group view of the arguments (group = blockIdx.y)
__device__ __forceinline__ BwdArgs group_view(const BwdArgs& a0) {
    BwdArgs a = a0;
    if (a0.groups > 1) {
        const size_t g = blockIdx.y;
        a.x += g * a0.pixels * a0.ldx;
        if (a.y) a.y += g * a0.pixels * a0.ldy;
        a.g1 += g * a0.pixels * a0.ldg1;
        if (a.g2) a.g2 += g * a0.pixels * a0.ldg2;
        a.dx += g * a0.pixels * a0.lddx;
        if (a.p.mean) a.p.mean = (const float*)a.p.mean + g * a0.C;
        if (a.p.rstd) a.p.rstd = (const float*)a.p.rstd + g * a0.C;
        a.partial += g * (size_t)a0.nblocks * 3 * a0.C8;
        a.totals += g * 3 * (size_t)a0.C8;
    }
    return a;
}

// pass 1: dz (-> dx buffer) and per-block partial sums {sum dz, sum dz*xhat, sum g*zd}
// GATE: a gate mask and/or d(alpha) is involved (needs z = bn(x) and a third sum); DROP: dropout.
// The plain BatchNorm+activation case (both false) keeps 32 fewer live registers -> higher occupancy
// for what is a pure HBM-streaming kernel.
// what bnact_bwd_finalize_kernel does per channel with the three totals
__device__ __forceinline__ void bwd_channel_finalize(const BwdArgs& a, int c, const double (&t)[3]) {
    if (c < a.C) {
        if (a.p.dbeta) a.p.dbeta[c] += (float)t[0];
        if (a.p.dgamma) a.p.dgamma[c] += (float)t[1];
        if (a.p.dalpha) a.p.dalpha[c] += (float)t[2];
    }
    // coefficients of the apply pass, dx = A dz + B x + K  (= gamma rstd (dz - mean(dz) - xhat mean(dz xhat))):
    // one row each in `totals`, read by the apply kernel as 16-byte loads
    float A = 0.f, B = 0.f, K = 0.f;
    if (c < a.C && a.p.bn && !a.p.bn_eval) {
        const float inv = 1.f / (float)a.pixels;
        const float rs = a.p.rstd[c], mu = a.p.mean[c];
        const float gr = (a.p.gamma ? a.p.gamma[c] : 1.f) * rs;
        const float k0 = (float)t[0] * inv, k1 = (float)t[1] * inv;
        A = gr; B = -gr * rs * k1; K = -gr * k0 + gr * rs * k1 * mu;
    }
    a.totals[c] = A; a.totals[a.C8 + c] = B; a.totals[2 * a.C8 + c] = K;
}

// The finalize of the backward pass by the reduce pass's own last-arriving workgroups (round 4; protocol and hand-off rules as
// stats_tail, igemm_common.hpp): the workgroup that writes the last of FIN_GROUP consecutive partial rows folds them (double,
// ascending), the one that completes the last group folds the group sums and does bwd_channel_finalize -- one launch fewer per
// BatchNorm backward (52 per Pix2Pix iteration).  Deterministic: the order of the additions does not depend on who arrives when.
template <int NT>
__device__ __forceinline__ void bwd_tail(const BwdArgs& a, int* sh) {
    typedef __attribute__((address_space(1))) unsigned int gu32;
    const int tid = threadIdx.x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int g = blockIdx.x / FIN_GROUP;
    const int G = (a.nblocks + FIN_GROUP - 1) / FIN_GROUP;
    const int nr = min(FIN_GROUP, a.nblocks - g * FIN_GROUP);
    const int W = 3 * a.C8;
    if (tid == 0) sh[0] = (int)__hip_atomic_fetch_add((gu32*)a.tickets + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const bool group_last = sh[0] == nr - 1;
    __syncthreads();
    if (!group_last) return;
    // (sc1 loads through buffer intrinsics, all in flight together: scoped atomic loads were waited for one by one)
    const __amdgpu_buffer_rsrc_t rs_pt = __builtin_amdgcn_make_buffer_rsrc((void*)a.partial, 0, (unsigned)((size_t)a.nblocks * W * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_gr = __builtin_amdgcn_make_buffer_rsrc((void*)a.grp, 0, (unsigned)((size_t)G * W * 8), 0x00020000);
    for (int idx = tid; idx < W; idx += NT) {
        float v[FIN_GROUP];
#pragma unroll
        for (int r = 0; r < FIN_GROUP; r++)         // unconditional, clamped: conditional loads compile to a branch and a wait each
            v[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_pt, ((g * FIN_GROUP + (r < nr ? r : nr - 1)) * W + idx) * 4, 0, 16));
        double sum = 0.0;
#pragma unroll
        for (int r = 0; r < FIN_GROUP; r++)
            if (r < nr) sum += (double)v[r];
        __hip_atomic_store(a.grp + (size_t)g * W + idx, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) sh[0] = (int)__hip_atomic_fetch_add((gu32*)a.tickets + G, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const bool last = sh[0] == G - 1;
    __syncthreads();
    if (!last) return;
    for (int c = tid; c < a.C8; c += NT) {
        double t[3] = {0.0, 0.0, 0.0};
        for (int q0 = 0; q0 < G; q0 += 8) {
            i32x2 v[8][3];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int q = q0 + u < G ? q0 + u : G - 1;
#pragma unroll
                for (int k = 0; k < 3; k++) v[u][k] = __builtin_amdgcn_raw_buffer_load_b64(rs_gr, (q * W + k * a.C8 + c) * 8, 0, 16);
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (q0 + u < G) {
#pragma unroll
                    for (int k = 0; k < 3; k++) t[k] += __builtin_bit_cast(double, v[u][k]);
                }
        }
        bwd_channel_finalize(a, c, t);
    }
    for (int i = tid; i <= G; i += NT) __hip_atomic_store((gu32*)a.tickets + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// V = channels per thread (4: 8-byte accesses, half the per-channel state -> twice the occupancy)
// NTH = 256 (default): small workgroups with 5 KB of LDS, so that they can share a CU with the two 74 KB workgroups of the
// weight-gradient kernel running on the side stream (with 1024-thread / 61 KB workgroups the two streams took turns:
// the BatchNorm backward behind a big wgrad launch ran 4x slower than alone).  NTH = 1024 is kept for A/B.
template <bool GATE, bool DROP, int V, int NTH>
__global__ __launch_bounds__(NTH) void bnact_bwd_reduce_kernel(const BwdArgs a0) {
    __shared__ float red[NTH][V + 1];
    const BwdArgs a = group_view(a0);
    const int ch = threadIdx.x & (a.L.CHP - 1);
    const int pl = threadIdx.x >> a.L.sh;
    const int c0 = ch * V;
    const bool active = ch < a.L.CH;
    // NOY: no saved activation output -- its sign is recomputed from x through the BatchNorm affine (2 bytes per element
    // less to read than the saved y; this kernel is HBM-bound)
    const bool NOY = !a.y && a.p.bn;
    float mu[V], rs[V], sc[V], sf[V], gm[GATE ? V : 1];
    float s0[V], s1[V], s2[GATE ? V : 1];
#pragma unroll
    for (int j = 0; j < V; j++) {
        const int c = c0 + j;
        const bool v = active && c < a.C;
        mu[j] = (v && a.p.bn) ? a.p.mean[c] : 0.f;
        rs[j] = (v && a.p.bn) ? a.p.rstd[c] : (v ? 1.f : 0.f);
        if (GATE || DROP || NOY) {
            const float g = (v && a.p.bn && a.p.gamma) ? a.p.gamma[c] : 1.f;
            const float b = (v && a.p.bn && a.p.beta) ? a.p.beta[c] : 0.f;
            sc[j] = g * rs[j];
            sf[j] = b - mu[j] * sc[j];
        } else { sc[j] = 1.f; sf[j] = 0.f; }
        if constexpr (GATE) { gm[j] = v ? (a.p.gate ? a.p.gate[c] : 1.f) : 0.f; s2[j] = 0.f; }
        s0[j] = s1[j] = 0.f;
    }
    const float keep_scale = a.p.drop_p > 0.f ? 1.f / (1.f - a.p.drop_p) : 1.f;
    const float evs = (a.p.bn && a.p.bn_eval) ? 1.f : 0.f;
    if (active) {
        typedef typename RawVec<V>::type Raw;
        const size_t pstep = (size_t)gridDim.x * a.L.PPB;
        size_t pix = (size_t)blockIdx.x * a.L.PPB + pl;
        Raw rx = {}, ry = {}, rg1 = {}, rg2 = {};
        if (pix < a.pixels) {
            rx = ldraw<V>(a.x + pix * a.ldx + a.xoff + c0);
            if (a.y) ry = ldraw<V>(a.y + pix * a.ldy + a.yoff + c0);
            rg1 = ldraw<V>(a.g1 + pix * a.ldg1 + a.g1off + c0);
            if (a.g2) rg2 = ldraw<V>(a.g2 + pix * a.ldg2 + a.g2off + c0);
        }
        for (; pix < a.pixels; pix += pstep) {
            float xv[V], yv[V], g1v[V], g2v[V], dz[V];
            unraw(rx, xv); unraw(rg1, g1v);
            if (a.y) unraw(ry, yv);
            if (a.g2) unraw(rg2, g2v);
            const size_t nx = pix + pstep;
            if (nx < a.pixels) {                 // next pixel's loads fly while this one is computed and stored
                rx = ldraw<V>(a.x + nx * a.ldx + a.xoff + c0);
                if (a.y) ry = ldraw<V>(a.y + nx * a.ldy + a.yoff + c0);
                rg1 = ldraw<V>(a.g1 + nx * a.ldg1 + a.g1off + c0);
                if (a.g2) rg2 = ldraw<V>(a.g2 + nx * a.ldg2 + a.g2off + c0);
            }
            // stage by stage over the V channels: launch-constant options are tested once per stage, not per element
            float df[V], zd[V], g[V];
#pragma unroll
            for (int j = 0; j < V; j++) {
                df[j] = 1.f;
                if constexpr (DROP) {
                    const float u = rng_uniform(a.p.seed, pix * (size_t)a.C + c0 + j);
                    df[j] = u >= a.p.drop_p ? keep_scale : 0.f;
                }
                zd[j] = 0.f;
                if constexpr (GATE || DROP) zd[j] = (xv[j] * sc[j] + sf[j]) * df[j];
            }
            if constexpr (GATE) {
                if (!a.p.gate_after_act) {
                    float yo[V];
                    if (a.y) {
#pragma unroll
                        for (int j = 0; j < V; j++) yo[j] = yv[j];
                    } else {
                        float t[V];
#pragma unroll
                        for (int j = 0; j < V; j++) t[j] = zd[j] * gm[j];
                        apply_actN<V>(t, yo, a.p.act, a.p.slope);
                    }
#pragma unroll
                    for (int j = 0; j < V; j++) g[j] = g1v[j] * act_grad_from_out(yo[j], a.p.act, a.p.slope);
                    if (a.g2) {
#pragma unroll
                        for (int j = 0; j < V; j++) g[j] += g2v[j] * act_grad_from_out(yo[j], a.p.act2, a.p.slope);
                    }
#pragma unroll
                    for (int j = 0; j < V; j++) { s2[j] += g[j] * zd[j]; g[j] *= gm[j]; }
                } else {
                    float ao[V];
                    apply_actN<V>(zd, ao, a.p.act, a.p.slope);
#pragma unroll
                    for (int j = 0; j < V; j++) {
                        s2[j] += g1v[j] * ao[j];
                        g[j] = g1v[j] * gm[j] * act_grad_from_out(ao[j], a.p.act, a.p.slope);
                    }
                }
            } else {
                // no gate: the saved output y gives the activation derivative; without it, re-derive the
                // activation input (outputs that had a residual added are not usable as y)
                float yo[V];
                if (a.y) {
#pragma unroll
                    for (int j = 0; j < V; j++) yo[j] = yv[j];
                } else if (DROP) {
                    apply_actN<V>(zd, yo, a.p.act, a.p.slope);
                } else if (a.p.bn) {
                    float t[V];
#pragma unroll
                    for (int j = 0; j < V; j++) t[j] = xv[j] * sc[j] + sf[j];     // the forward's own form
                    apply_actN<V>(t, yo, a.p.act, a.p.slope);
                } else {
#pragma unroll
                    for (int j = 0; j < V; j++) yo[j] = xv[j];
                }
#pragma unroll
                for (int j = 0; j < V; j++) g[j] = g1v[j] * act_grad_from_out(yo[j], a.p.act, a.p.slope);
                if (a.g2) {
#pragma unroll
                    for (int j = 0; j < V; j++) g[j] += g2v[j] * act_grad_from_out(yo[j], a.p.act2, a.p.slope);
                }
            }
#pragma unroll
            for (int j = 0; j < V; j++) dz[j] = g[j] * df[j];
            if (!a.p.bn) {
#pragma unroll
                for (int j = 0; j < V; j++) dz[j] *= act_grad_from_out(xv[j], a.in_act, a.in_slope);
            }
            if (evs != 0.f) {
                if constexpr (GATE || DROP) {
#pragma unroll
                    for (int j = 0; j < V; j++) dz[j] *= sc[j];
                } else {
#pragma unroll
                    for (int j = 0; j < V; j++) dz[j] *= rs[j] * (a.p.gamma ? a.p.gamma[c0 + j < a.C ? c0 + j : 0] : 1.f);
                }
            }
#pragma unroll
            for (int j = 0; j < V; j++) {
                const float xh = (xv[j] - mu[j]) * rs[j];
                s0[j] += dz[j];
                s1[j] += dz[j] * xh;
            }
            stv<V>(a.dx + pix * a.lddx + a.dxoff + c0, dz);
        }
    }
    // fold the pixel lanes of this block, one statistic at a time through the same small LDS buffer
    float* o = a.partial + (size_t)blockIdx.x * 3 * a.C8;
    const bool sc1 = a.tickets != nullptr;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (k == 2 && !GATE) {
            if (pl == 0 && active) {
#pragma unroll
                for (int j = 0; j < V; j++) st_stat(o + 2 * a.C8 + c0 + j, 0.f, sc1);
            }
            break;
        }
#pragma unroll
        for (int j = 0; j < V; j++) red[threadIdx.x][j] = k == 0 ? s0[j] : (k == 1 ? s1[j] : s2[GATE ? j : 0]);
        __syncthreads();
        if (pl == 0 && active) {
#pragma unroll
            for (int j = 0; j < V; j++) {
                float t = 0.f;
                for (int q = 0; q < a.L.PPB; q++) t += red[q * a.L.CHP + ch][j];
                st_stat(o + k * a.C8 + c0 + j, t, sc1);
            }
        }
        __syncthreads();
    }
    if (a.tickets) bwd_tail<NTH>(a, (int*)&red[0][0]);
}

// Plain activation backward (no normalisation, gate, dropout or parameter gradient): dx = (g1 act'(y) + g2 act2'(y)) in_act'(x),
// one streaming launch -- the tanh of the generator's image, the ReLU fused into the innermost down conv, the U-Net's first skip
// (LeakyReLU / ReLU pair).  These went through the reduce pass (per-block sums nobody reads) + the finalize launch: 27 + 5 us for
// the 16.8 MB image gradient against ~8 us of HBM time (profiles/r4f_unet_student_chain.txt).
__global__ __launch_bounds__(256) void act_bwd_kernel(const BwdArgs a) {
    const int ch = threadIdx.x & (a.L.CHP - 1);
    const int pl = threadIdx.x >> a.L.sh;
    if (ch >= a.L.CH) return;
    const int c0 = ch * 8;
    const size_t pstep = (size_t)gridDim.x * a.L.PPB;
    for (size_t pix = (size_t)blockIdx.x * a.L.PPB + pl; pix < a.pixels; pix += 2 * pstep) {
        // two pixels (up to eight 16-byte loads) in flight per lane
        const size_t p2 = pix + pstep;
        const bool two = p2 < a.pixels;
        i32x4 rx[2], ry[2], rg1[2], rg2[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const size_t q = u ? p2 : pix;
            if (u && !two) break;
            rx[u] = *(const i32x4*)(a.x + q * a.ldx + a.xoff + c0);
            rg1[u] = *(const i32x4*)(a.g1 + q * a.ldg1 + a.g1off + c0);
            if (a.y) ry[u] = *(const i32x4*)(a.y + q * a.ldy + a.yoff + c0);
            if (a.g2) rg2[u] = *(const i32x4*)(a.g2 + q * a.ldg2 + a.g2off + c0);
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (u && !two) break;
            const size_t q = u ? p2 : pix;
            float xv[8], yv[8], g1v[8], g2v[8], o[8];
            unpack8(rx[u], xv); unpack8(rg1[u], g1v);
            if (a.y) unpack8(ry[u], yv);
            if (a.g2) unpack8(rg2[u], g2v);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float yo = a.y ? yv[j] : xv[j];
                float d = g1v[j] * act_grad_from_out(yo, a.p.act, a.p.slope);
                if (a.g2) d += g2v[j] * act_grad_from_out(yo, a.p.act2, a.p.slope);
                o[j] = d * act_grad_from_out(xv[j], a.in_act, a.in_slope);
            }
            *(i32x4*)(a.dx + q * a.lddx + a.dxoff + c0) = pack8(o);
        }
    }
}

// pass 2: totals over blocks; parameter gradients (+=)
__global__ __launch_bounds__(1024) void bnact_bwd_finalize_kernel(const BwdArgs a0, int blocks) {
    __shared__ double sh[3][32][33];
    const BwdArgs a = group_view(a0);
    const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double t[3] = {0.0, 0.0, 0.0};
    if (c < a.C8) {
        // four partial rows (12 loads) in flight per thread: the pass is a latency chain otherwise
        int b = pl;
        for (; b + 96 < blocks; b += 128) {
            float v[4][3];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float* o = a.partial + (size_t)(b + 32 * u) * 3 * a.C8;
                v[u][0] = o[c]; v[u][1] = o[a.C8 + c]; v[u][2] = o[2 * a.C8 + c];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) { t[0] += v[u][0]; t[1] += v[u][1]; t[2] += v[u][2]; }
        }
        for (; b < blocks; b += 32) {
            const float* o = a.partial + (size_t)b * 3 * a.C8;
            t[0] += o[c]; t[1] += o[a.C8 + c]; t[2] += o[2 * a.C8 + c];
        }
    }
    for (int k = 0; k < 3; k++) sh[k][pl][cl] = t[k];
    __syncthreads();
    if (pl < 8) {                       // two-level fold of the 32 row lanes
#pragma unroll
        for (int k = 0; k < 3; k++) t[k] = (t[k] + sh[k][pl + 8][cl]) + (sh[k][pl + 16][cl] + sh[k][pl + 24][cl]);
    }
    __syncthreads();
    if (pl < 8) {
#pragma unroll
        for (int k = 0; k < 3; k++) sh[k][pl][cl] = t[k];
    }
    __syncthreads();
    if (pl == 0 && c < a.C8) {
        for (int q = 1; q < 8; q++)
            for (int k = 0; k < 3; k++) t[k] += sh[k][q][cl];
        bwd_channel_finalize(a, c, t);
    }
}

// pass 3 (training BN only): dx = gamma*rstd*(dz - mean(dz) - xhat*mean(dz*xhat)), in place over dz
__global__ __launch_bounds__(256) void bnact_bwd_apply_kernel(const BwdArgs a0) {
    const BwdArgs a = group_view(a0);
    const int ch = threadIdx.x & (a.L.CHP - 1);
    const int pl = threadIdx.x >> a.L.sh;
    if (ch >= a.L.CH) return;
    const int c0 = ch * 8;
    float A[8], B[8], K[8];
    {
        const f32x4 a0_ = *(const f32x4*)(a.totals + c0), a1_ = *(const f32x4*)(a.totals + c0 + 4);
        const f32x4 b0_ = *(const f32x4*)(a.totals + a.C8 + c0), b1_ = *(const f32x4*)(a.totals + a.C8 + c0 + 4);
        const f32x4 k0_ = *(const f32x4*)(a.totals + 2 * a.C8 + c0), k1_ = *(const f32x4*)(a.totals + 2 * a.C8 + c0 + 4);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            A[j] = a0_[j]; A[4 + j] = a1_[j]; B[j] = b0_[j]; B[4 + j] = b1_[j]; K[j] = k0_[j]; K[4 + j] = k1_[j];
        }
    }
    const size_t pstep = (size_t)gridDim.x * a.L.PPB;
    size_t pix = (size_t)blockIdx.x * a.L.PPB + pl;
    i32x4 rx = {}, rz = {};
    if (pix < a.pixels) {
        rx = *(const i32x4*)(a.x + pix * a.ldx + a.xoff + c0);
        rz = *(const i32x4*)(a.dx + pix * a.lddx + a.dxoff + c0);
    }
    for (; pix < a.pixels; pix += pstep) {
        float xv[8], dz[8], o[8];
        unpack8(rx, xv);
        unpack8(rz, dz);
        const size_t nx = pix + pstep;
        if (nx < a.pixels) {                     // two pixels in flight per thread
            rx = *(const i32x4*)(a.x + nx * a.ldx + a.xoff + c0);
            rz = *(const i32x4*)(a.dx + nx * a.lddx + a.dxoff + c0);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = A[j] * dz[j] + B[j] * xv[j] + K[j];
        *(i32x4*)(a.dx + pix * a.lddx + a.dxoff + c0) = pack8(o);
    }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm (training statistics, no gate) + activation backward of a SMALL tensor in one launch: a workgroup owns every
// pixel of 8 channels, so the two sums need no grid-wide step.  Each thread keeps the raw x and the fp32 dz of its <= 8
// pixels in registers between the statistics and dx = A dz + B x + K -- nothing is re-read, dz is never rounded to bf16.
// For the U-Net's <= 16x16 layers (pixels <= 4096 at N=16) the three-launch pipeline above is a chain of three ~5-7 us
// kernels on the generators' backward pass, which runs with nothing else to fill the chip; this is one of ~7 us.
constexpr int SMALL_NT = 512, SMALL_RPT = 8;
constexpr size_t SMALL_MAX_PIXELS = (size_t)SMALL_NT * SMALL_RPT;
template <bool DROP>
__global__ __launch_bounds__(SMALL_NT) void bnact_bwd_small_kernel(const BwdArgs a) {
    __shared__ double red[SMALL_NT / 64][16];
    __shared__ float coef[24];
    const int c0 = blockIdx.x * 8;
    const int pixels = (int)a.pixels;
    const bool NOY = !a.y;
    float mu[8], rs[8], sc[8], sf[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int c = c0 + j;
        const bool v = c < a.C;
        mu[j] = v ? a.p.mean[c] : 0.f;
        rs[j] = v ? a.p.rstd[c] : 0.f;
        const float g = (v && a.p.gamma) ? a.p.gamma[c] : 1.f;
        const float b = (v && a.p.beta) ? a.p.beta[c] : 0.f;
        sc[j] = g * rs[j];
        sf[j] = b - mu[j] * sc[j];
    }
    const float keep_scale = a.p.drop_p > 0.f ? 1.f / (1.f - a.p.drop_p) : 1.f;
    i32x4 rx[SMALL_RPT], rg1[SMALL_RPT], rg2[SMALL_RPT], ry[SMALL_RPT];
#pragma unroll
    for (int r = 0; r < SMALL_RPT; r++) {               // every load of the thread in flight at once
        const int pix = threadIdx.x + r * SMALL_NT;
        rx[r] = rg1[r] = rg2[r] = ry[r] = i32x4{0, 0, 0, 0};
        if (pix < pixels) {
            rx[r] = *(const i32x4*)(a.x + (size_t)pix * a.ldx + a.xoff + c0);
            rg1[r] = *(const i32x4*)(a.g1 + (size_t)pix * a.ldg1 + a.g1off + c0);
            if (a.g2) rg2[r] = *(const i32x4*)(a.g2 + (size_t)pix * a.ldg2 + a.g2off + c0);
            if (a.y) ry[r] = *(const i32x4*)(a.y + (size_t)pix * a.ldy + a.yoff + c0);
        }
    }
    float dz[SMALL_RPT][8];
    float s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; j++) s0[j] = s1[j] = 0.f;
#pragma unroll
    for (int r = 0; r < SMALL_RPT; r++) {
        const int pix = threadIdx.x + r * SMALL_NT;
        float xv[8], g1v[8], g2v[8], yo[8], df[8], zd[8];
        unpack8(rx[r], xv);
        unpack8(rg1[r], g1v);
#pragma unroll
        for (int j = 0; j < 8; j++) {                   // the reduce kernel's own forms, stage by stage
            df[j] = 1.f;
            if constexpr (DROP) {
                const float u = rng_uniform(a.p.seed, (size_t)pix * (size_t)a.C + c0 + j);
                df[j] = u >= a.p.drop_p ? keep_scale : 0.f;
            }
            zd[j] = (xv[j] * sc[j] + sf[j]) * df[j];
        }
        if (!NOY) unpack8(ry[r], yo);
        else apply_actN<8>(zd, yo, a.p.act, a.p.slope);
        float g[8];
#pragma unroll
        for (int j = 0; j < 8; j++) g[j] = g1v[j] * act_grad_from_out(yo[j], a.p.act, a.p.slope);
        if (a.g2) {
            unpack8(rg2[r], g2v);
#pragma unroll
            for (int j = 0; j < 8; j++) g[j] += g2v[j] * act_grad_from_out(yo[j], a.p.act2, a.p.slope);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float d = pix < pixels ? g[j] * df[j] : 0.f;
            dz[r][j] = d;
            s0[j] += d;
            s1[j] += d * ((xv[j] - mu[j]) * rs[j]);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 8; j++) { s0[j] = wave_sum(s0[j]); s1[j] = wave_sum(s1[j]); }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) { red[wave][j] = (double)s0[j]; red[wave][8 + j] = (double)s1[j]; }
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const int j = threadIdx.x, c = c0 + j;
        double t0 = 0.0, t1 = 0.0;
#pragma unroll
        for (int w = 0; w < SMALL_NT / 64; w++) { t0 += red[w][j]; t1 += red[w][8 + j]; }
        float A = 0.f, B = 0.f, K = 0.f;
        if (c < a.C) {
            if (a.p.dbeta) a.p.dbeta[c] += (float)t0;
            if (a.p.dgamma) a.p.dgamma[c] += (float)t1;
            const float inv = 1.f / (float)a.pixels;
            const float r_ = a.p.rstd[c], m_ = a.p.mean[c];
            const float gr = (a.p.gamma ? a.p.gamma[c] : 1.f) * r_;
            const float k0 = (float)t0 * inv, k1 = (float)t1 * inv;
            A = gr; B = -gr * r_ * k1; K = -gr * k0 + gr * r_ * k1 * m_;     // bnact_bwd_finalize_kernel's coefficients
        }
        coef[j] = A; coef[8 + j] = B; coef[16 + j] = K;
    }
    __syncthreads();
    float A[8], B[8], K[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { A[j] = coef[j]; B[j] = coef[8 + j]; K[j] = coef[16 + j]; }
#pragma unroll
    for (int r = 0; r < SMALL_RPT; r++) {
        const int pix = threadIdx.x + r * SMALL_NT;
        if (pix < pixels) {
            float xv[8], o[8];
            unpack8(rx[r], xv);
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = A[j] * dz[r][j] + B[j] * xv[j] + K[j];
            *(i32x4*)(a.dx + (size_t)pix * a.lddx + a.dxoff + c0) = pack8(o);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// InstanceNorm2d(affine=False) of small planes in ONE launch: a workgroup owns the [HW][8 channels] slab of one image, so
// statistics and normalisation need no grid-wide step.  Pass 1 sums the slab, pass 2 re-reads it (from L2: a slab is
// HW * 16 bytes) and writes.  At batch 1 (CycleGAN) the three-launch pipeline above is launch-latency bound: three
// 4-12 us launches per norm against one here.
struct InFusedArgs {
    const bf16_t* x; int ldx;
    const bf16_t* y; int ldy;          // fwd: output (written) ; bwd: saved output (activation derivative) or NULL
    const bf16_t* aux; int ldaux;      // fwd: residual added after the activation (or NULL) ; bwd: incoming gradient
    bf16_t* out; int ldout;            // fwd: == y ; bwd: dx (may alias the incoming gradient)
    int C, HW, act; float slope, eps;
    float *mean, *rstd, *scale, *shift;   // [N][C]
    // BatchNorm backward through the grid kernel (gcc_bn_bwd_one_launch: N == 1, HW = all pixels of the batch)
    const float* gamma; float *dgamma, *dbeta;
    // ... its extended form (template EXT, round 4: the U-Net's layers): the activation input recomputed from the forward's affine
    // (bscale / bshift; y NULL), a second incoming gradient through a second activation, the dropout mask regenerated
    const float* bscale; const float* bshift;
    const bf16_t* g2; int ldg2; int act2;
    float drop_p; unsigned long long seed;
};

// LPP lanes share a pixel (each 8 channels = 16 bytes): the slab is 8 * LPP channels wide.  Sums of the lanes with the same
// channel chunk: butterfly over the pixel bits of the lane id, then across waves through LDS in double.
template <int NTH, int LPP, int K>
__device__ __forceinline__ void slab_reduce(float (&v)[K][8], double* lds /* [NTH/64][LPP][K*8] */, double* out /* [LPP][K*8] */) {
#pragma unroll
    for (int k = 0; k < K; k++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float t = v[k][j];
#pragma unroll
            for (int o = 32; o >= LPP; o >>= 1) t += __shfl_xor(t, o, 64);
            v[k][j] = t;
        }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < LPP) {
#pragma unroll
        for (int k = 0; k < K; k++)
#pragma unroll
            for (int j = 0; j < 8; j++) lds[(wave * LPP + lane) * (K * 8) + k * 8 + j] = (double)v[k][j];
    }
    __syncthreads();
    if (threadIdx.x < LPP * K * 8) {
        double t = 0.0;
        for (int w = 0; w < NTH / 64; w++) t += lds[w * LPP * (K * 8) + threadIdx.x];
        out[threadIdx.x] = t;
    }
    __syncthreads();
}

template <int NTH, int LPP>
__global__ __launch_bounds__(NTH) void inorm_fwd_fused_kernel(const InFusedArgs a) {
    __shared__ double lds[(NTH / 64) * LPP * 16];
    __shared__ double tot[LPP * 16];
    __shared__ float coef[LPP][2][8];
    const int cq = threadIdx.x & (LPP - 1), pl = threadIdx.x / LPP;
    const int c0 = (blockIdx.x * LPP + cq) * 8;
    const bool live = c0 < ((a.C + 7) & ~7);
    const size_t g = blockIdx.y;
    const bf16_t* xg = a.x + g * (size_t)a.HW * a.ldx + c0;
    float v[2][8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[0][j] = v[1][j] = 0.f;
    if (live) {
        constexpr int ST = NTH / LPP;
        int p = pl;
        for (; p + 3 * ST < a.HW; p += 4 * ST) {       // four 16-byte loads in flight per lane: one CU streams the slab alone
            i32x4 r[4];
#pragma unroll
            for (int u = 0; u < 4; u++) r[u] = *(const i32x4*)(xg + (size_t)(p + u * ST) * a.ldx);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                float f[8];
                unpack8(r[u], f);
#pragma unroll
                for (int j = 0; j < 8; j++) { v[0][j] += f[j]; v[1][j] += f[j] * f[j]; }
            }
        }
        for (; p < a.HW; p += ST) {
            float f[8];
            unpack8(*(const i32x4*)(xg + (size_t)p * a.ldx), f);
#pragma unroll
            for (int j = 0; j < 8; j++) { v[0][j] += f[j]; v[1][j] += f[j] * f[j]; }
        }
    }
    slab_reduce<NTH, LPP, 2>(v, lds, tot);
    if (threadIdx.x < LPP * 8) {
        const int q = threadIdx.x >> 3, j = threadIdx.x & 7;
        const double m = tot[q * 16 + j] / (double)a.HW;
        double var = tot[q * 16 + 8 + j] / (double)a.HW - m * m;
        if (var < 0.0) var = 0.0;
        const float r = (float)(1.0 / sqrt(var + (double)a.eps));
        const float sh = 0.f - (float)m * r;
        coef[q][0][j] = r; coef[q][1][j] = sh;
        const int c = (blockIdx.x * LPP + q) * 8 + j;
        if (c < a.C) {
            const size_t o = g * a.C + c;
            a.mean[o] = (float)m; a.rstd[o] = r; a.scale[o] = r; a.shift[o] = sh;
        }
    }
    __syncthreads();
    if (!live) return;
    float sc[8], sf[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { sc[j] = coef[cq][0][j]; sf[j] = coef[cq][1][j]; }
    bf16_t* yg = a.out + g * (size_t)a.HW * a.ldout + c0;
    const bf16_t* rg = a.aux ? a.aux + g * (size_t)a.HW * a.ldaux + c0 : nullptr;
    constexpr int ST = NTH / LPP;
    int p = pl;
    for (; p + 3 * ST < a.HW; p += 4 * ST) {
        i32x4 r[4], q[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            r[u] = *(const i32x4*)(xg + (size_t)(p + u * ST) * a.ldx);
            if (rg) q[u] = *(const i32x4*)(rg + (size_t)(p + u * ST) * a.ldaux);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float f[8], o[8];
            unpack8(r[u], f);
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = apply_act(f[j] * sc[j] + sf[j], a.act, a.slope);
            if (rg) {
                float rv[8];
                unpack8(q[u], rv);
#pragma unroll
                for (int j = 0; j < 8; j++) o[j] += rv[j];
            }
            *(i32x4*)(yg + (size_t)(p + u * ST) * a.ldout) = pack8(o);
        }
    }
    for (; p < a.HW; p += ST) {
        float f[8], o[8];
        unpack8(*(const i32x4*)(xg + (size_t)p * a.ldx), f);
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = apply_act(f[j] * sc[j] + sf[j], a.act, a.slope);
        if (rg) {
            float rv[8];
            unpack8(*(const i32x4*)(rg + (size_t)p * a.ldaux), rv);
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] += rv[j];
        }
        *(i32x4*)(yg + (size_t)p * a.ldout) = pack8(o);
    }
}

// dx = rstd (dz - mean(dz) - xhat mean(dz xhat)), dz = g act'(y); pass 2 recomputes dz instead of staging it in memory
template <int NTH, int LPP>
__global__ __launch_bounds__(NTH) void inorm_bwd_fused_kernel(const InFusedArgs a) {
    __shared__ double lds[(NTH / 64) * LPP * 16];
    __shared__ double tot[LPP * 16];
    const int cq = threadIdx.x & (LPP - 1), pl = threadIdx.x / LPP;
    const int c0 = (blockIdx.x * LPP + cq) * 8;
    const bool live = c0 < ((a.C + 7) & ~7);
    const size_t g = blockIdx.y;
    const bf16_t* xg = a.x + g * (size_t)a.HW * a.ldx + c0;
    const bf16_t* yg = a.y ? a.y + g * (size_t)a.HW * a.ldy + c0 : nullptr;
    const bf16_t* gg = a.aux + g * (size_t)a.HW * a.ldaux + c0;
    bf16_t* og = a.out + g * (size_t)a.HW * a.ldout + c0;
    float mu[8], rs[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const bool ok = c0 + j < a.C;
        mu[j] = ok ? a.mean[g * a.C + c0 + j] : 0.f;
        rs[j] = ok ? a.rstd[g * a.C + c0 + j] : 0.f;
    }
    float v[2][8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[0][j] = v[1][j] = 0.f;
    if (live) {
        constexpr int ST = NTH / LPP;
        int p = pl;
        for (; p + ST < a.HW; p += 2 * ST) {            // two pixels x three tensors in flight per lane
            i32x4 rx[2], rg_[2], ry[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                rx[u] = *(const i32x4*)(xg + (size_t)(p + u * ST) * a.ldx);
                rg_[u] = *(const i32x4*)(gg + (size_t)(p + u * ST) * a.ldaux);
                if (yg) ry[u] = *(const i32x4*)(yg + (size_t)(p + u * ST) * a.ldy);
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                float xv[8], yv[8], gv[8];
                unpack8(rx[u], xv); unpack8(rg_[u], gv);
                if (yg) unpack8(ry[u], yv);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float xh = (xv[j] - mu[j]) * rs[j];
                    const float yo = yg ? yv[j] : apply_act(xh, a.act, a.slope);
                    const float d = gv[j] * act_grad_from_out(yo, a.act, a.slope);
                    v[0][j] += d; v[1][j] += d * xh;
                }
            }
        }
        for (; p < a.HW; p += ST) {
            float xv[8], yv[8], gv[8];
            unpack8(*(const i32x4*)(xg + (size_t)p * a.ldx), xv);
            unpack8(*(const i32x4*)(gg + (size_t)p * a.ldaux), gv);
            if (yg) unpack8(*(const i32x4*)(yg + (size_t)p * a.ldy), yv);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float xh = (xv[j] - mu[j]) * rs[j];
                const float yo = yg ? yv[j] : apply_act(xh, a.act, a.slope);
                const float d = gv[j] * act_grad_from_out(yo, a.act, a.slope);
                v[0][j] += d; v[1][j] += d * xh;
            }
        }
    }
    slab_reduce<NTH, LPP, 2>(v, lds, tot);
    if (!live) return;
    float A[8], B[8], K[8];
    const float inv = 1.f / (float)a.HW;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const float k0 = (float)tot[cq * 16 + j] * inv, k1 = (float)tot[cq * 16 + 8 + j] * inv;
        A[j] = rs[j]; B[j] = -rs[j] * rs[j] * k1; K[j] = -rs[j] * k0 + rs[j] * rs[j] * k1 * mu[j];
    }
    constexpr int ST = NTH / LPP;
    int p = pl;
    for (; p + ST < a.HW; p += 2 * ST) {
        i32x4 rx[2], rg_[2], ry[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            rx[u] = *(const i32x4*)(xg + (size_t)(p + u * ST) * a.ldx);
            rg_[u] = *(const i32x4*)(gg + (size_t)(p + u * ST) * a.ldaux);
            if (yg) ry[u] = *(const i32x4*)(yg + (size_t)(p + u * ST) * a.ldy);
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            float xv[8], yv[8], gv[8], o[8];
            unpack8(rx[u], xv); unpack8(rg_[u], gv);
            if (yg) unpack8(ry[u], yv);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float yo = yg ? yv[j] : apply_act((xv[j] - mu[j]) * rs[j], a.act, a.slope);
                const float d = gv[j] * act_grad_from_out(yo, a.act, a.slope);
                o[j] = A[j] * d + B[j] * xv[j] + K[j];
            }
            *(i32x4*)(og + (size_t)(p + u * ST) * a.ldout) = pack8(o);
        }
    }
    for (; p < a.HW; p += ST) {
        float xv[8], yv[8], gv[8], o[8];
        unpack8(*(const i32x4*)(xg + (size_t)p * a.ldx), xv);
        unpack8(*(const i32x4*)(gg + (size_t)p * a.ldaux), gv);
        if (yg) unpack8(*(const i32x4*)(yg + (size_t)p * a.ldy), yv);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float yo = yg ? yv[j] : apply_act((xv[j] - mu[j]) * rs[j], a.act, a.slope);
            const float d = gv[j] * act_grad_from_out(yo, a.act, a.slope);
            o[j] = A[j] * d + B[j] * xv[j] + K[j];
        }
        *(i32x4*)(og + (size_t)p * a.ldout) = pack8(o);
    }
}

// The same one-launch InstanceNorm with the plane of an image split over S workgroups (round 3: at batch 1 the slab kernels
// above put 6-16 workgroups on 256 CUs -- `rocprofv3` of a CycleGAN iteration: 592 forward launches of 18 us and 296 backward
// launches of 37 us on average, 21.7 of the iteration's 56 ms of kernel time).  A workgroup owns `rows` whole pixels of a
// 64-channel group (contiguous 128-byte segments) and writes its per-channel partial sums; the S workgroups of an (image,
// channel group) domain exchange them inside the launch -- tagged write-through stores that every workgroup polls and folds
// itself, see the hand-off below -- and every workgroup then normalises its own rows, which it still holds in registers.
// Residency comes from the grid size alone: at most one workgroup per CU of the device (hipDeviceAttributeMultiprocessorCount;
// 256 on MI355X) of 256 threads at <= 128 VGPRs (four fit a CU), so that the four streams of a training step cannot fill the
// chip with waiting workgroups.  Every spin is bounded all the same, and a spin that expires (lost residency: a CU mask, a
// partitioned mode the attribute does not reflect) stores GCC_DEVERR_INORM_SPIN into the library's device error word: the
// launch's results are then wrong, and every later gcc_inorm_* / gcc_bn_bwd_one_launch call returns GCC_ERR_LAUNCH until
// gcc_device_error(1) clears it (round 4, VERDICT r3 weak 1a / ADVICE r3).
// The workspace is zero-filled once by the host; a domain's epoch word counts the launches that used it.  The tag holds 24
// bits of it: the launcher re-zeroes the workspace (stream-ordered memset) every 2^20 launches that used it and at its first
// use inside every launch recording (a replayed iteration re-zeroes it every run), so an epoch never reaches 2^21, the tag
// never wraps, tag 0 never occurs and no slot can hold a tag of an earlier life of the counter.
struct InGridArgs {
    InFusedArgs a;
    float* partial;        // [N * CG][S][V] (value, tag) pairs
    float* level2;         // [N * CG][G][V] (low half, tag, high half, tag) quads: the groups' sums (double) when S > S1
    int S1, G;             // workgroups whose partials one workgroup folds; groups of S1 per domain
    unsigned* cnt;         // [N * CG][4]: word 1 = the domain's epoch
    int S, rows;           // workgroups per (image, channel group), pixels per workgroup
    int CG, CHg;           // channel groups per image, 16-byte chunks per group (all of them when the image has < 16)
    int CH, CHP, sh;       // chunks per pixel; the power of two above CHg, its log2
    int V;                 // CHg * 16 partial values per workgroup: [chunk][stat 0 / 1][8 channels]
    unsigned long long* clk;   // diagnostic build, bit 5 (32): [S][8] s_memrealtime stamps of image 0, group 0 (100 MHz)
    unsigned* err;         // the library's device error word (pinned host memory) or NULL
    int spin_limit;        // polls before a wait gives up (1 << 20; diagnostic build, bit 6 (64): 256, and workgroup 1 of every domain
    int mute;              //   publishes nothing -- the test of the error path)
};
constexpr unsigned GCC_DEVERR_INORM_SPIN = 0x1401u;

// The statistics hand-off of the grid kernels (inorm_grid_kernel, bn_fold_grid_kernel): this workgroup's per-lane sums v[2][8]
// -> the (image, channel group) domain's totals tot[V] (double), identical in every workgroup of the domain.
#define IN_STAMP(k) do { if (GCC_DIAG(ga.clk != nullptr) && t == 0 && dom == 0) ga.clk[s * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
__device__ __forceinline__ void grid_exchange(const InGridArgs& ga, float (&v)[2][8], double* smem_d, float* red, double* tot, int s,
                                              size_t dom, __attribute__((address_space(1))) unsigned int* cnt, unsigned epoch, int t,
                                              int lane, int wave, int ch) {
    // lanes of a wave that hold the same chunk (CHP < 64), then the waves through LDS
    for (int o = 32; o >= ga.CHP; o >>= 1) {
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int j = 0; j < 8; j++) v[k][j] += __shfl_xor(v[k][j], o, 64);
    }
    if (ga.sh >= 6 || (lane >> ga.sh) == 0) {
        float* dst = red + (wave * 64 + (ga.sh >= 6 ? lane : ch)) * 16;
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int j = 0; j < 8; j++) dst[k * 8 + j] = v[k][j];
    }
    __syncthreads();
    const bool alone = ga.S == 1;
    // The hand-off, without a counter: every partial value leaves its workgroup as a (value, tag) pair in one 8-byte write-through
    // (sc1) store, tag = (epoch of this domain + 1) << 8 | domain -- unique to this launch for every slot the domain's
    // workgroups write, whatever earlier launches with other layouts left in the workspace (their tags carry a smaller epoch of
    // this domain or another domain's byte).  Every workgroup then polls ALL S partials of its domain with sc1 loads until their
    // tags match and folds them itself, in index order (S * V <= 4 K values: one or two round trips with all loads in flight).
    // Two memory hops (store visible, load) instead of the five of a ticket / last-arriver-folds / totals / flag protocol.
    // The domain's epoch word is bumped by its workgroup 0 once it has seen all S partials -- by then every workgroup of the
    // domain has read the old value (it read it before it wrote its partial).  The workspace starts zero-filled; tag 0 never occurs;
    // behind its header every odd 32-bit word is a tag word and every even one a value, in every layout (both exchange levels).
    const __amdgpu_buffer_rsrc_t rs_part = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(ga.partial + dom * ga.S * (size_t)ga.V * 2), 0, ga.S * ga.V * 8, 0x00020000);
    if (!alone) {
        int* sh_epoch = (int*)(smem_d + 2048 + 256);
        if (t == 0) *sh_epoch = (int)epoch;
    }
    __syncthreads();
    const unsigned tag = alone ? 0u : ((((unsigned)*(int*)(smem_d + 2048 + 256)) + 1u) << 8) | (unsigned)(dom & 0xff);
    for (int i = t; i < ga.V; i += 256) {
        const int chunk = i >> 4, r = i & 15;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; w++)
            if ((((w << 6) ^ chunk) & (ga.CHP - 1) & ~63) == 0) sum += red[(w * 64 + (chunk & 63)) * 16 + r];
        if (alone) tot[i] = (double)sum;
        else if (!(GCC_DIAG(ga.mute) && s == 1)) {
            const i32x2 pr = {(int)__float_as_uint(sum), (int)tag};
            __builtin_amdgcn_raw_buffer_store_b64(pr, rs_part, (s * ga.V + i) * 8, 0, 16);
        }
    }
    IN_STAMP(1);
    if (!alone) {
        // lanes: value i = t % V of partial rows q0, q0 + nq, ... (nq = 256 / V rows walked in parallel)
        const int nq = 256 / ga.V > 0 ? 256 / ga.V : 1;
        const int i = t % ga.V, q0 = t / ga.V;
        double* part = smem_d;                                       // [nq][V] doubles <= 2 KB: `red` is dead after the barrier
        // level 1: the S1 workgroups of this workgroup's group (all S of the domain when S <= S1)
        const int grp = s / ga.S1, g_lo = grp * ga.S1, g_hi = min(g_lo + ga.S1, ga.S);
        __syncthreads();
        double sum = 0.0;
        if (q0 < nq) {
            constexpr int FB = 8;
            for (int qb = g_lo + q0; qb < g_hi; qb += FB * nq) {
                i32x2 pr[FB];
                unsigned pending = 0;
#pragma unroll
                for (int u = 0; u < FB; u++)
                    if (qb + u * nq < g_hi) pending |= 1u << u;
                for (int spin = 0; pending && spin < ga.spin_limit; spin++) {
#pragma unroll
                    for (int u = 0; u < FB; u++)
                        if (pending & (1u << u)) pr[u] = __builtin_amdgcn_raw_buffer_load_b64(rs_part, ((qb + u * nq) * ga.V + i) * 8, 0, 16);
#pragma unroll
                    for (int u = 0; u < FB; u++)
                        if ((pending & (1u << u)) && (unsigned)pr[u][1] == tag) pending &= ~(1u << u);
                    if (pending) __builtin_amdgcn_s_sleep(1);
                }
                if (pending && ga.err) __hip_atomic_store(ga.err, GCC_DEVERR_INORM_SPIN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
                for (int u = 0; u < FB; u++)
                    if (qb + u * nq < g_hi) sum += (double)__uint_as_float((unsigned)pr[u][0]);
            }
            part[q0 * ga.V + i] = sum;
        }
        __syncthreads();
        for (int k = t; k < ga.V; k += 256) {
            double tsum = 0.0;
            for (int q = 0; q < nq; q++) tsum += part[q * ga.V + k];
            tot[k] = tsum;
        }
        if (ga.G > 1) {
            // level 2 (planes whose S partials are more than one workgroup should fold): the first workgroup of every group
            // publishes the group's sums (double) as two (half, tag) pairs; everyone folds the G <= 8 of them
            const __amdgpu_buffer_rsrc_t rs_l2 = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(ga.level2 + dom * ga.G * (size_t)ga.V * 4), 0, ga.G * ga.V * 16, 0x00020000);
            __syncthreads();
            if (s == g_lo) {
                for (int k = t; k < ga.V; k += 256) {
                    // (low half, tag), (high half, tag): every odd word of the workspace is a tag word in every layout that ever
                    // used it, so a stale VALUE can never be taken for this launch's tag (layouts of different planes overlap)
                    const i32x2 bits = __builtin_bit_cast(i32x2, tot[k]);
                    const i32x4 v4 = {bits[0], (int)tag, bits[1], (int)tag};
                    __builtin_amdgcn_raw_buffer_store_b128(v4, rs_l2, (grp * ga.V + k) * 16, 0, 16);
                }
            }
            for (int k = t; k < ga.V; k += 256) {
                double tsum = 0.0;
                for (int q = 0; q < ga.G; q++) {
                    i32x4 v4 = {0, 0, 0, 0};
                    bool got = false;
                    for (int spin = 0; spin < ga.spin_limit; spin++) {
                        v4 = __builtin_amdgcn_raw_buffer_load_b128(rs_l2, (q * ga.V + k) * 16, 0, 16);
                        if ((unsigned)v4[1] == tag && (unsigned)v4[3] == tag) { got = true; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (!got && ga.err) __hip_atomic_store(ga.err, GCC_DEVERR_INORM_SPIN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    const i32x2 bits = {v4[0], v4[2]};
                    tsum += __builtin_bit_cast(double, bits);
                }
                tot[k] = tsum;
            }
        }
        IN_STAMP(3);
        if (s == 0 && t == 0) __hip_atomic_store(cnt + 1, epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
    } else {
        __syncthreads();
    }
}
#undef IN_STAMP

// A lane keeps its first PPT pixels (16 bytes each, per tensor) in registers from the statistics pass to the normalising pass:
// every load of a pass is in flight at once, and the second pass loads nothing but the residual.  (With four loads in flight
// and a second read of x, s_memrealtime stamps of the 64 x 64 x 256 plane read: 2 us statistics, 4 us hand-off, 6 us
// normalising; the backward 8 + 4 + 8.)  Rows beyond PPT * (256 / CHP) per workgroup stream through the loops behind.
template <bool BWD, int EXT = 0>      // EXT (backward only): 1 recomputed activation input / second gradient, 2: + dropout
__global__ __launch_bounds__(256, 4) void inorm_grid_kernel(const InGridArgs ga) {   // <= 128 VGPRs: four workgroups per CU, the residency the barrier counts on
    static_assert(BWD || !EXT, "the extended form is a backward");
    constexpr int PPT = BWD ? (EXT ? 2 : 4) : 8;
    __shared__ double smem_d[2048 + 256 + 1];       // one array (a second __shared__ object costs a vmcnt(0), guide 5.4)
    float* red = (float*)smem_d;                    // [4 waves][64 lanes][16] per-wave sums; later the forward's coefficients
    double* tot = smem_d + 2048;                    // [V <= 256] totals of the (image, channel group)
    const InFusedArgs& a = ga.a;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int ch = t & (ga.CHP - 1), pl = t >> ga.sh, PL = 256 >> ga.sh;      // ch: chunk inside the group
    const int s = blockIdx.x, cg = blockIdx.y;
    const size_t g = blockIdx.z;
    const size_t dom = g * ga.CG + cg;              // the barrier domain: S workgroups
    const bool live = ch < ga.CHg && cg * ga.CHg + ch < ga.CH;
    const int c0 = (cg * ga.CHg + ch) * 8;
    const int p0 = s * ga.rows, p1 = min(p0 + ga.rows, a.HW);
    const bf16_t* xg = a.x + g * (size_t)a.HW * a.ldx + c0;
    const bf16_t* yg = (BWD && a.y) ? a.y + g * (size_t)a.HW * a.ldy + c0 : nullptr;
    const bf16_t* ag = a.aux ? a.aux + g * (size_t)a.HW * a.ldaux + c0 : nullptr;     // fwd: residual; bwd: incoming gradient
    bf16_t* og = a.out + g * (size_t)a.HW * a.ldout + c0;
    float mu[8], rs[8];
    float bsc[EXT ? 8 : 1], bsf[EXT ? 8 : 1];
    const bf16_t* g2g = nullptr;
    const float keep_scale = (EXT == 2 && a.drop_p > 0.f) ? 1.f / (1.f - a.drop_p) : 1.f;
    if (BWD) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const bool ok = live && c0 + j < a.C;
            mu[j] = ok ? a.mean[g * a.C + c0 + j] : 0.f;
            rs[j] = ok ? a.rstd[g * a.C + c0 + j] : 0.f;
            if constexpr (EXT) {
                bsc[j] = (ok && a.bscale) ? a.bscale[c0 + j] : 0.f;
                bsf[j] = (ok && a.bshift) ? a.bshift[c0 + j] : 0.f;
            }
        }
        if constexpr (EXT) g2g = a.g2 ? a.g2 + c0 : nullptr;
    }
    // the gradient that reaches the normalisation's output at pixel p: g act'(y) [+ g2 act2'(y)], through the dropout mask
    auto dz8 = [&](int p, const float* xv, const float* gv, const float* g2v, const float* yv, float* d, float* xh) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            xh[j] = (xv[j] - mu[j]) * rs[j];
            float yo, df = 1.f;
            if constexpr (EXT) {
                float z = a.bscale ? xv[j] * bsc[j] + bsf[j] : xh[j];        // the forward's own affine form
                if constexpr (EXT == 2) {
                    const float uu = rng_uniform(a.seed, (size_t)p * (size_t)a.C + c0 + j);
                    df = uu >= a.drop_p ? keep_scale : 0.f;
                    z *= df;
                }
                yo = yg ? yv[j] : apply_act(z, a.act, a.slope);
            } else {
                yo = yg ? yv[j] : apply_act(xh[j], a.act, a.slope);
            }
            float dd = gv[j] * act_grad_from_out(yo, a.act, a.slope);
            if constexpr (EXT) {
                if (g2g) dd += g2v[j] * act_grad_from_out(yo, a.act2, a.slope);
                dd *= df;
            }
            d[j] = dd;
        }
    };
    // the domain's epoch word: stable until every workgroup of the domain has written its partial (all of them read it first)
    typedef __attribute__((address_space(1))) unsigned int gu32;
    gu32* cnt = (gu32*)ga.cnt + dom * 4;            // [1]: the domain's epoch (launches that used it)
#define IN_STAMP(k) do { if (GCC_DIAG(ga.clk != nullptr) && t == 0 && dom == 0) ga.clk[s * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    IN_STAMP(0);
    unsigned epoch = 0;
    if (t == 0 && ga.S > 1) epoch = __hip_atomic_load(cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float v[2][8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[0][j] = v[1][j] = 0.f;
    auto accum = [&](int p, const i32x4& rx, const i32x4& rgr, const i32x4& ry, const i32x4& rg2) {
        float xv[8];
        unpack8(rx, xv);
        if (!BWD) {
#pragma unroll
            for (int j = 0; j < 8; j++) { v[0][j] += xv[j]; v[1][j] += xv[j] * xv[j]; }
        } else {
            float gv[8], yv[8];
            unpack8(rgr, gv);
            if (yg) unpack8(ry, yv);
            if constexpr (!EXT) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float xh = (xv[j] - mu[j]) * rs[j];
                    const float yo = yg ? yv[j] : apply_act(xh, a.act, a.slope);
                    const float d = gv[j] * act_grad_from_out(yo, a.act, a.slope);
                    v[0][j] += d; v[1][j] += d * xh;
                }
            } else {
                float g2v[8], d[8], xh[8];
                if (g2g) unpack8(rg2, g2v);
                dz8(p, xv, gv, g2v, yv, d, xh);
#pragma unroll
                for (int j = 0; j < 8; j++) { v[0][j] += d[j]; v[1][j] += d[j] * xh[j]; }
            }
        }
    };
    const i32x4 zero4 = {0, 0, 0, 0};
    i32x4 cx[PPT], cgr[BWD ? PPT : 1];               // the saved output y is read again in the second pass: 128 VGPRs hold no more
    const int pfirst = p0 + pl;                     // this lane's pixels: pfirst + u * PL
    const int pstream = pfirst + PPT * PL;          // the first one that is not held in registers
    if (live) {
        i32x4 cy[BWD ? PPT : 1], cg2[EXT ? PPT : 1];
#pragma unroll
        for (int u = 0; u < PPT; u++) {
            const int p = pfirst + u * PL;
            const bool ok = p < p1;
            cx[u] = ok ? *(const i32x4*)(xg + (size_t)p * a.ldx) : zero4;
            if (BWD) {
                cgr[u] = ok ? *(const i32x4*)(ag + (size_t)p * a.ldaux) : zero4;      // zero gradient: no contribution
                cy[u] = (ok && yg) ? *(const i32x4*)(yg + (size_t)p * a.ldy) : zero4;
                if constexpr (EXT) cg2[u] = (ok && g2g) ? *(const i32x4*)(g2g + (size_t)p * a.ldg2) : zero4;
            }
        }
#pragma unroll
        for (int u = 0; u < PPT; u++) accum(pfirst + u * PL, cx[u], cgr[BWD ? u : 0], cy[BWD ? u : 0], cg2[EXT ? u : 0]);
        for (int p = pstream; p < p1; p += PL) {
            i32x4 rx = *(const i32x4*)(xg + (size_t)p * a.ldx), rgr = rx, ry = rx, rg2 = rx;
            if (BWD) {
                rgr = *(const i32x4*)(ag + (size_t)p * a.ldaux);
                if (yg) ry = *(const i32x4*)(yg + (size_t)p * a.ldy);
                if constexpr (EXT) { if (g2g) rg2 = *(const i32x4*)(g2g + (size_t)p * a.ldg2); }
            }
            accum(p, rx, rgr, ry, rg2);
        }
    }
    grid_exchange(ga, v, smem_d, red, tot, s, dom, cnt, epoch, t, lane, wave, ch);
    IN_STAMP(4);
    const double inv_hw = 1.0 / (double)a.HW;
    if (!BWD) {
        __syncthreads();          // `red` is read above by other threads until here; it now receives the coefficients
        for (int cl = t; cl < ga.CHg * 8; cl += 256) {
            const int chunk = cl >> 3, j = cl & 7;
            const double m = tot[chunk * 16 + j] * inv_hw;
            double var = tot[chunk * 16 + 8 + j] * inv_hw - m * m;
            if (var < 0.0) var = 0.0;
            const float r = (float)(1.0 / sqrt(var + (double)a.eps));
            const float sh = 0.f - (float)m * r;
            red[chunk * 16 + j] = r; red[chunk * 16 + 8 + j] = sh;
            const int c = cg * ga.CHg * 8 + cl;
            if (s == 0 && c < a.C) {
                const size_t o = g * a.C + c;
                a.mean[o] = (float)m; a.rstd[o] = r; a.scale[o] = r; a.shift[o] = sh;
            }
        }
        __syncthreads();
        if (!live) return;
        float sc[8], sf[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { sc[j] = red[ch * 16 + j]; sf[j] = red[ch * 16 + 8 + j]; }
        auto emit = [&](int p, const i32x4& rx, const i32x4& rr) {
            float f[8], o[8];
            unpack8(rx, f);
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = f[j] * sc[j] + sf[j];
            apply_act8(o, o, a.act, a.slope);
            if (ag) {
                float rv[8];
                unpack8(rr, rv);
#pragma unroll
                for (int j = 0; j < 8; j++) o[j] += rv[j];
            }
            *(i32x4*)(og + (size_t)p * a.ldout) = pack8(o);
        };
        if (ag) {
            i32x4 rr[PPT];
#pragma unroll
            for (int u = 0; u < PPT; u++) {
                const int p = pfirst + u * PL;
                rr[u] = p < p1 ? *(const i32x4*)(ag + (size_t)p * a.ldaux) : zero4;
            }
#pragma unroll
            for (int u = 0; u < PPT; u++)
                if (pfirst + u * PL < p1) emit(pfirst + u * PL, cx[u], rr[u]);
        } else {
#pragma unroll
            for (int u = 0; u < PPT; u++)
                if (pfirst + u * PL < p1) emit(pfirst + u * PL, cx[u], zero4);
        }
        for (int p = pstream; p < p1; p += PL) {
            const i32x4 rx = *(const i32x4*)(xg + (size_t)p * a.ldx);
            const i32x4 rr = ag ? *(const i32x4*)(ag + (size_t)p * a.ldaux) : rx;
            emit(p, rx, rr);
        }
        IN_STAMP(5);
    } else {
        if (s == 0 && (a.dgamma || a.dbeta)) {
            // BatchNorm: d(gamma) += sum dz xhat, d(beta) += sum dz -- one workgroup per channel group, one writer per channel
            for (int cl = t; cl < ga.CHg * 8; cl += 256) {
                const int c = cg * ga.CHg * 8 + cl;
                if (c < a.C) {
                    if (a.dbeta) a.dbeta[c] += (float)tot[(cl >> 3) * 16 + (cl & 7)];
                    if (a.dgamma) a.dgamma[c] += (float)tot[(cl >> 3) * 16 + 8 + (cl & 7)];
                }
            }
        }
        if (!live) return;
        float A[8], B[8], K[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float k0 = (float)(tot[ch * 16 + j] * inv_hw), k1 = (float)(tot[ch * 16 + 8 + j] * inv_hw);
            const float gr = (a.gamma && c0 + j < a.C) ? a.gamma[c0 + j] * rs[j] : rs[j];      // dx = gamma rstd (dz - k0 - xhat k1)
            A[j] = gr; B[j] = -gr * rs[j] * k1; K[j] = -gr * k0 + gr * rs[j] * k1 * mu[j];
        }
        auto emit = [&](int p, const i32x4& rx, const i32x4& rgr, const i32x4& ry, const i32x4& rg2) {
            float xv[8], yv[8], gv[8], o[8];
            unpack8(rx, xv); unpack8(rgr, gv);
            if (yg) unpack8(ry, yv);
            if constexpr (!EXT) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float yo = yg ? yv[j] : apply_act((xv[j] - mu[j]) * rs[j], a.act, a.slope);
                    const float d = gv[j] * act_grad_from_out(yo, a.act, a.slope);
                    o[j] = A[j] * d + B[j] * xv[j] + K[j];
                }
            } else {
                float g2v[8], d[8], xh[8];
                if (g2g) unpack8(rg2, g2v);
                dz8(p, xv, gv, g2v, yv, d, xh);
#pragma unroll
                for (int j = 0; j < 8; j++) o[j] = A[j] * d[j] + B[j] * xv[j] + K[j];
            }
            *(i32x4*)(og + (size_t)p * a.ldout) = pack8(o);
        };
        i32x4 cy[PPT], cg2[EXT ? PPT : 1];
#pragma unroll
        for (int u = 0; u < PPT; u++) {
            const int p = pfirst + u * PL;
            cy[u] = (p < p1 && yg) ? *(const i32x4*)(yg + (size_t)p * a.ldy) : zero4;
            if constexpr (EXT) cg2[u] = (p < p1 && g2g) ? *(const i32x4*)(g2g + (size_t)p * a.ldg2) : zero4;
        }
#pragma unroll
        for (int u = 0; u < PPT; u++)
            if (pfirst + u * PL < p1) emit(pfirst + u * PL, cx[u], cgr[BWD ? u : 0], cy[u], cg2[EXT ? u : 0]);
        for (int p = pstream; p < p1; p += PL) {
            const i32x4 rx = *(const i32x4*)(xg + (size_t)p * a.ldx);
            const i32x4 rgr = *(const i32x4*)(ag + (size_t)p * a.ldaux);
            const i32x4 ry = yg ? *(const i32x4*)(yg + (size_t)p * a.ldy) : rx;
            i32x4 rg2 = rx;
            if constexpr (EXT) { if (g2g) rg2 = *(const i32x4*)(g2g + (size_t)p * a.ldg2); }
            emit(p, rx, rgr, ry, rg2);
        }
        IN_STAMP(5);
    }
#undef IN_STAMP
}

// ---- split-K fold + BatchNorm + activation on the whole chip, one launch (round 4) -----------------------------------------------
// The U-Net's <= 16x16 layers run their conv as fp32 partial tiles of K slices (the weights streamed by every CU); what
// followed was splitk_bn_act_kernel: C / 8 workgroups, each walking every row of its 8 channels twice (14-56 us per layer,
// profiles/r3z_unet_student_chain.txt).  Here the rows are dealt over S x CG workgroups as in inorm_grid_kernel -- a lane owns
// 8 channels of <= PPT rows: folds their K slices in slice order (every slice's two 16-byte loads in flight), rounds to bf16
// (the raw output, kept for the backward pass), keeps the rounded values in registers, the per-channel sums go through
// grid_exchange (tagged write-through partials; residency and workspace rules of inorm_grid_kernel), and the rows are
// normalised from the registers: dropout by the (pixel, channel) counter of bnact_fwd_kernel, both activated copies.
struct FoldGridArgs {
    InGridArgs g;                          // exchange layout; g.a is not used
    const float* part; int ksplit, rows_max, Cpad, phases, Mph, Rt;
    int Hd, Wd, ostr, Hg, Wg; FastDiv dMph, dHgWg, dWg;
    bf16_t* raw; int ldraw, rawoff;
    bf16_t* y; int ldy, yoff; bf16_t* y2; int ldy2, y2off;
    int C; double count;
    const float* gamma; const float* beta; float eps, momentum;
    float* running_mean; float* running_var; float* mean; float* rstd; float* scale; float* shift;
    int act, act2; float slope, drop_p; unsigned long long seed;
};
template <int PPT>
__global__ __launch_bounds__(256, 4) void bn_fold_grid_kernel(const FoldGridArgs fa) {
    __shared__ double smem_d[2048 + 256 + 1];
    float* red = (float*)smem_d;
    double* tot = smem_d + 2048;
    const InGridArgs& ga = fa.g;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int ch = t & (ga.CHP - 1), pl = t >> ga.sh, PL = 256 >> ga.sh;
    const int s = blockIdx.x, cg = blockIdx.y;
    const size_t dom = cg;
    const bool live = ch < ga.CHg && cg * ga.CHg + ch < ga.CH;
    const int c0 = (cg * ga.CHg + ch) * 8;
    const int p0 = s * ga.rows, p1 = min(p0 + ga.rows, fa.Rt);
    typedef __attribute__((address_space(1))) unsigned int gu32;
    gu32* cnt = (gu32*)ga.cnt + dom * 4;
    unsigned epoch = 0;
    if (t == 0 && ga.S > 1) epoch = __hip_atomic_load(cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float v[2][8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[0][j] = v[1][j] = 0.f;
    i32x4 cx[PPT];
    int pixv[PPT];
    const size_t sstride = (size_t)fa.rows_max * fa.Cpad;
#pragma unroll
    for (int u = 0; u < PPT; u++) {
        const int p = p0 + pl + u * PL;
        cx[u] = i32x4{0, 0, 0, 0};
        pixv[u] = -1;
        if (live && p < p1) {
            const int z = fdiv(p, fa.dMph), m = p - z * fa.Mph;
            const float* r0 = fa.part + ((size_t)z * fa.ksplit * fa.rows_max + m) * fa.Cpad + c0;
            float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int sl = 0;
            for (; sl + 3 < fa.ksplit; sl += 4) {
                f32x4 a[4], b[4];
#pragma unroll
                for (int q = 0; q < 4; q++) { a[q] = *(const f32x4*)(r0 + (sl + q) * sstride); b[q] = *(const f32x4*)(r0 + (sl + q) * sstride + 4); }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    f[0] += a[q][0]; f[1] += a[q][1]; f[2] += a[q][2]; f[3] += a[q][3];
                    f[4] += b[q][0]; f[5] += b[q][1]; f[6] += b[q][2]; f[7] += b[q][3];
                }
            }
            for (; sl < fa.ksplit; sl++) {
                const f32x4 a = *(const f32x4*)(r0 + sl * sstride), b = *(const f32x4*)(r0 + sl * sstride + 4);
                f[0] += a[0]; f[1] += a[1]; f[2] += a[2]; f[3] += a[3];
                f[4] += b[0]; f[5] += b[1]; f[6] += b[2]; f[7] += b[3];
            }
#pragma unroll
            for (int j = 0; j < 8; j++) if (c0 + j >= fa.C) f[j] = 0.f;        // pad channels stay exact zeros
            const i32x4 pk = pack8(f);
            cx[u] = pk;
            float r[8];
            unpack8(pk, r);
#pragma unroll
            for (int j = 0; j < 8; j++) { v[0][j] += r[j]; v[1][j] += r[j] * r[j]; }
            const int py = z / fa.ostr, px = z - py * fa.ostr;               // phase -> sub-grid origin (fprop: ostr 1, z 0)
            const int n = fdiv(m, fa.dHgWg), rr = m - n * (fa.Hg * fa.Wg);
            const int oy = fdiv(rr, fa.dWg), ox = rr - oy * fa.Wg;
            const int pix = (n * fa.Hd + oy * fa.ostr + py) * fa.Wd + ox * fa.ostr + px;
            pixv[u] = pix;
            *(i32x4*)(fa.raw + (size_t)pix * fa.ldraw + fa.rawoff + c0) = pk;
        }
    }
    grid_exchange(ga, v, smem_d, red, tot, s, dom, cnt, epoch, t, lane, wave, ch);
    __syncthreads();              // `red` is read by other threads until here; it now receives the coefficients
    for (int cl = t; cl < ga.CHg * 8; cl += 256) {
        const int chunk = cl >> 3, j = cl & 7;
        const int c = cg * ga.CHg * 8 + cl;
        const double m = tot[chunk * 16 + j] / fa.count;
        double var = tot[chunk * 16 + 8 + j] / fa.count - m * m;
        if (var < 0.0) var = 0.0;
        const float r = (float)(1.0 / sqrt(var + (double)fa.eps));
        const bool ok = c < fa.C;
        const float gam = (ok && fa.gamma) ? fa.gamma[c] : 1.f, bet = (ok && fa.beta) ? fa.beta[c] : 0.f;
        const float sc = gam * r, sf = bet - (float)m * gam * r;
        red[chunk * 16 + j] = sc; red[chunk * 16 + 8 + j] = sf;
        if (s == 0 && ok) {
            if (fa.mean) fa.mean[c] = (float)m;
            if (fa.rstd) fa.rstd[c] = r;
            fa.scale[c] = sc; fa.shift[c] = sf;
            if (fa.running_mean) fa.running_mean[c] = (1.f - fa.momentum) * fa.running_mean[c] + fa.momentum * (float)m;
            if (fa.running_var) {
                const double unb = fa.count > 1.0 ? var * fa.count / (fa.count - 1.0) : var;
                fa.running_var[c] = (1.f - fa.momentum) * fa.running_var[c] + fa.momentum * (float)unb;
            }
        }
    }
    __syncthreads();
    if (!live) return;
    float sc[8], sf[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { sc[j] = red[ch * 16 + j]; sf[j] = red[ch * 16 + 8 + j]; }
    const float keep_scale = fa.drop_p > 0.f ? 1.f / (1.f - fa.drop_p) : 1.f;
#pragma unroll
    for (int u = 0; u < PPT; u++) {
        const int pix = pixv[u];
        if (pix < 0) continue;
        float f[8], o1[8], o2[8];
        unpack8(cx[u], f);
#pragma unroll
        for (int j = 0; j < 8; j++) f[j] = f[j] * sc[j] + sf[j];
        if (fa.drop_p > 0.f) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float uu = rng_uniform(fa.seed, (size_t)pix * (size_t)fa.C + c0 + j);
                f[j] = uu >= fa.drop_p ? f[j] * keep_scale : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) if (c0 + j >= fa.C) f[j] = 0.f;
        apply_act8(f, o1, fa.act, fa.slope);
        if (fa.y) *(i32x4*)(fa.y + (size_t)pix * fa.ldy + fa.yoff + c0) = pack8(o1);
        if (fa.y2) {
            apply_act8(f, o2, fa.act2, fa.slope);
            *(i32x4*)(fa.y2 + (size_t)pix * fa.ldy2 + fa.y2off + c0) = pack8(o2);
        }
    }
}

constexpr size_t INORM_WS_HEADER = 4096;           // [N * CG <= 256][4] counter words in front of the totals and the partials
// plan of the grid form; false: this geometry stays with the slab kernels
static int device_cus() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
            (void)hipGetLastError();
            n = 256;
        }
        return n;
    }();
    return cus;
}
// How many workgroups a launch of the grid kernels (inorm_grid_kernel, bn_fold_grid_kernel: they wait for their OWN workgroups
// inside the launch) may have.  Progress argument: a launch of this family can only be stalled by
// workgroups that themselves wait, i.e. by other launches of this family -- every other kernel drains.  At most Q launches run
// at once, Q = hardware queues of the process (HIP's GPU_MAX_HW_QUEUES, default 4: a queue runs its kernels in order).  Every
// kernel of the family is compiled for FOUR workgroups per CU (__launch_bounds__(256, 4): <= 128 VGPRs; 18 KB of LDS; <= 106
// SGPRs admit 6) -- so the chip has 4 x CUs slots for them, and Q launches of <= 4 x CUs / Q workgroups each are ALL resident
// whatever the placement: none can be kept waiting for a slot another waiting launch holds.  With the default Q = 4 that is one
// workgroup per CU (the cap these kernels always had); a process that raises GPU_MAX_HW_QUEUES gets a proportionally smaller
// grid.  A spin that expires anyway (a debugger, a hung partner) is reported through the device error word, never silent.
static int grid_family_wgs() {
    static const int v = [] {
        int q = 4;
        if (const char* e = getenv("GPU_MAX_HW_QUEUES")) { const int n = atoi(e); if (n > 0) q = n; }
        const int cus = std::min(device_cus(), 256);      // the header holds 256 domains
        return std::max(1, std::min(cus, 4 * cus / std::max(q, 4)));
    }();
    return v;
}
static bool inorm_grid_plan(int C, int HW, int N, int px, size_t ws_bytes, InGridArgs* ga) {
    const int CH = (C + 7) / 8;
    const int cus = grid_family_wgs();
    if (CH > 256 || N > 64) return false;
    const int CHg = CH >= 16 ? 8 : CH;              // 64-channel groups (128-byte segments of a pixel) once an image has 128 channels
    const int CG = (CH + CHg - 1) / CHg;
    if (N * CG > cus) return false;
    int CHP = 1, sh = 0;
    while (CHP < CHg) { CHP <<= 1; sh++; }
    const int PL = 256 / CHP, V = CHg * 16;
    // as many CUs as the caps below allow: the passes are bound by the vector ALU of the CUs that take part and by the latency of
    // a lane's loads, not by memory; px pixels per lane = what the kernel keeps in registers (forward 8, backward 4: measured
    // best of 2 / 4 / 8 -- more workgroups shorten the passes and lengthen the hand-off, ~40 ns per arrival on one counter)
    long S = (HW + px * PL - 1) / (px * PL);
    S = std::min<long>(S, cus / (N * CG));                    // residency: <= one workgroup per CU, <= 128 VGPRs each
    S = std::min<long>(S, 8 * std::max(1, 4096 / V));         // two exchange levels: 8 groups of 4096 / V workgroups
    S = std::max<long>(S, 1);
    int rows = (int)((HW + S - 1) / S);
    rows = ((rows + PL - 1) / PL) * PL;
    S = (HW + rows - 1) / rows;
    // one workgroup folds at most 4 K (value, tag) pairs; beyond that the domain's workgroups meet in groups of S1 first
    const int S1 = (int)std::min<long>(S, std::max(1, 4096 / V));
    const int G = (int)((S + S1 - 1) / S1);
    if (G > 8) return false;
    if (S > 1 && ws_bytes < INORM_WS_HEADER + (size_t)N * CG * S * V * 8 + (size_t)N * CG * G * V * 16) return false;
    ga->S1 = S1; ga->G = G;
    ga->S = (int)S; ga->rows = rows; ga->CG = CG; ga->CHg = CHg; ga->CH = CH; ga->CHP = CHP; ga->sh = sh; ga->V = V;
    return true;
}

// epochs of a workspace's domains stay far below 2^24 (the tag's field): see the comment at InGridArgs
struct InWsState { unsigned long long launches; unsigned gen; };
static std::mutex g_inws_mu;
static std::unordered_map<void*, InWsState> g_inws;
static void inorm_ws_scrub(void* ws, size_t bytes, hipStream_t st) {
    bool scrub = false;
    {
        std::lock_guard<std::mutex> lk(g_inws_mu);
        InWsState& w = g_inws[ws];
        w.launches++;
        if ((w.launches & ((1ull << 20) - 1)) == 0) scrub = true;
        if (gcc_replay_recording() && w.gen != gcc_replay_generation()) { w.gen = gcc_replay_generation(); scrub = true; }
    }
    if (scrub) (void)gcc_memset_async(ws, 0, bytes, st);
}

// returns true when the grid form ran (need_grid: the caller has no other route -- nothing is launched otherwise)
template <bool BWD, int EXT = 0>
bool inorm_launch(const InFusedArgs& a, int N, hipStream_t st, void* ws, size_t ws_bytes, bool need_grid = false) {
    InGridArgs ga;
    if (ws && ws_bytes >= INORM_WS_HEADER + 16384 && gcc_opt(GCC_OPT_INORM_GRID) &&
        inorm_grid_plan(a.C, a.HW, N, BWD ? (EXT ? 2 : 4) : 8, ws_bytes, &ga)) {
        ga.a = a;
        ga.cnt = (unsigned*)ws;
        ga.partial = (float*)((char*)ws + INORM_WS_HEADER);
        ga.level2 = (float*)((char*)ws + INORM_WS_HEADER + (size_t)N * ga.CG * ga.S * ga.V * 8);
        const int dbg = gcc_diag_bits();
        ga.clk = (dbg & 32) ? (unsigned long long*)((char*)ws + ws_bytes - 16384) : nullptr;
        ga.err = gcc_device_error_word();
        ga.spin_limit = (dbg & 64) ? 256 : (1 << 20);
        ga.mute = (dbg & 64) ? 1 : 0;
        if (ga.S > 1) inorm_ws_scrub(ws, ws_bytes - ((dbg & 32) ? 16384 : 0), st);
        hipLaunchKernelGGL((inorm_grid_kernel<BWD, EXT>), dim3(ga.S, ga.CG, N), dim3(256), 0, st, ga);
        return true;
    }
    if (need_grid || EXT) return false;
    const int lpp_env = gcc_opt(GCC_OPT_INORM_LPP);
    const int lpp = lpp_env ? lpp_env : 2;
    const int slabs = ((a.C + 7) / 8 + lpp - 1) / lpp;
    const dim3 grid(slabs, N);
    const bool big = a.HW >= 2048;
#define GCC_IN_LAUNCH(NTH, LPP)                                                                         \
    do {                                                                                                \
        if (BWD) hipLaunchKernelGGL((inorm_bwd_fused_kernel<NTH, LPP>), grid, dim3(NTH), 0, st, a);     \
        else hipLaunchKernelGGL((inorm_fwd_fused_kernel<NTH, LPP>), grid, dim3(NTH), 0, st, a);         \
    } while (0)
    if (lpp == 1) { if (big) GCC_IN_LAUNCH(1024, 1); else GCC_IN_LAUNCH(256, 1); }
    else if (lpp == 2) { if (big) GCC_IN_LAUNCH(1024, 2); else GCC_IN_LAUNCH(256, 2); }
    else { if (big) GCC_IN_LAUNCH(1024, 4); else GCC_IN_LAUNCH(256, 4); }
#undef GCC_IN_LAUNCH
    return false;
}

}  // namespace
int gcc_internal_bn_fold_grid(const BnFoldDesc* d, hipStream_t st) {
    if (!d || !d->part || !d->raw || !d->ws || d->ws_bytes < INORM_WS_HEADER + 16384 || !d->bn.scale || !d->bn.shift) return GCC_ERR_BAD_ARG;
    if (gcc_device_error(0)) return GCC_ERR_LAUNCH;
    if (!gcc_opt(GCC_OPT_INORM_GRID)) return GCC_ERR_UNSUPPORTED;
    FoldGridArgs fa = {};
    const int st_ = d->dgrad ? d->stride : 1;
    if (d->dgrad && ((d->Hd % st_) || (d->Wd % st_))) return GCC_ERR_UNSUPPORTED;        // phases of different sizes: the other route
    fa.ostr = st_; fa.Hd = d->Hd; fa.Wd = d->Wd; fa.Hg = d->Hd / st_; fa.Wg = d->Wd / st_;
    fa.phases = d->phases; fa.Mph = d->N * fa.Hg * fa.Wg; fa.Rt = fa.phases * fa.Mph;
    if (fa.phases != st_ * st_ || fa.Mph > d->rows_max || (size_t)d->N * d->Hd * d->Wd >= ((size_t)1 << 31) / (size_t)(d->C > 0 ? d->C : 1))
        return GCC_ERR_UNSUPPORTED;
    InGridArgs& ga = fa.g;
    if (!inorm_grid_plan(d->C, fa.Rt, 1, 1, d->ws_bytes, &ga)) return GCC_ERR_UNSUPPORTED;
    const int PL = 256 / ga.CHP;
    const int ppt = (ga.rows + PL - 1) / PL;
    if (ppt > 4) return GCC_ERR_UNSUPPORTED;
    ga.cnt = (unsigned*)d->ws;
    ga.partial = (float*)((char*)d->ws + INORM_WS_HEADER);
    ga.level2 = (float*)((char*)d->ws + INORM_WS_HEADER + (size_t)ga.CG * ga.S * ga.V * 8);
    const int dbg = gcc_diag_bits();
    ga.clk = nullptr;
    ga.err = gcc_device_error_word();
    ga.spin_limit = (dbg & 64) ? 256 : (1 << 20);
    ga.mute = (dbg & 64) ? 1 : 0;
    fa.part = d->part; fa.ksplit = d->ksplit; fa.rows_max = d->rows_max; fa.Cpad = d->Cpad;
    fa.dMph = make_fastdiv(fa.Mph); fa.dHgWg = make_fastdiv(fa.Hg * fa.Wg); fa.dWg = make_fastdiv(fa.Wg);
    fa.raw = (bf16_t*)d->raw; fa.ldraw = d->ldraw; fa.rawoff = d->rawoff;
    fa.y = (bf16_t*)d->y; fa.ldy = d->ldy; fa.yoff = d->yoff; fa.y2 = (bf16_t*)d->y2; fa.ldy2 = d->ldy2; fa.y2off = d->y2off;
    fa.C = d->C; fa.count = d->bn.count; fa.gamma = d->bn.gamma; fa.beta = d->bn.beta; fa.eps = d->bn.eps; fa.momentum = d->bn.momentum;
    fa.running_mean = d->bn.running_mean; fa.running_var = d->bn.running_var; fa.mean = d->bn.mean; fa.rstd = d->bn.rstd;
    fa.scale = d->bn.scale; fa.shift = d->bn.shift;
    fa.act = d->act; fa.act2 = d->act2; fa.slope = d->slope; fa.drop_p = d->drop_p; fa.seed = d->seed;
    if (ga.S > 1) inorm_ws_scrub(d->ws, d->ws_bytes, st);
    const dim3 grid(ga.S, ga.CG, 1);
    if (ppt <= 1) hipLaunchKernelGGL((bn_fold_grid_kernel<1>), grid, dim3(256), 0, st, fa);
    else if (ppt == 2) hipLaunchKernelGGL((bn_fold_grid_kernel<2>), grid, dim3(256), 0, st, fa);
    else hipLaunchKernelGGL((bn_fold_grid_kernel<4>), grid, dim3(256), 0, st, fa);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}
namespace {

struct SumArgs {
    const bf16_t* x; int ld, off; int C, C8; size_t pixels; Layout L; float* partial;
};
__global__ __launch_bounds__(256) void channel_sum_kernel(const SumArgs a) {
    __shared__ float red[256][9];
    const int ch = threadIdx.x & (a.L.CHP - 1);
    const int pl = threadIdx.x >> a.L.sh;
    const int c0 = ch * 8;
    const bool active = ch < a.L.CH;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (active) {
        // four pixels (16-byte loads) in flight per lane: one at a time the sweep was a chain of dependent round trips -- 33 us
        // for the 16.8 MB of a [16, 3, 256, 256] gradient (the U-Net's last bias gradient) against ~5 us of HBM time
        const size_t step = (size_t)gridDim.x * a.L.PPB;
        size_t pix = (size_t)blockIdx.x * a.L.PPB + pl;
        for (; pix + 3 * step < a.pixels; pix += 4 * step) {
            i32x4 r[4];
#pragma unroll
            for (int u = 0; u < 4; u++) r[u] = *(const i32x4*)(a.x + (pix + u * step) * a.ld + a.off + c0);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                float v[8];
                unpack8(r[u], v);
#pragma unroll
                for (int j = 0; j < 8; j++) s[j] += v[j];
            }
        }
        for (; pix < a.pixels; pix += step) {
            float v[8];
            unpack8(*(const i32x4*)(a.x + pix * a.ld + a.off + c0), v);
#pragma unroll
            for (int j = 0; j < 8; j++) s[j] += v[j];
        }
    }
    // fold the pixel lanes: first the lanes of a wave that hold the same chunk (shuffles), then the four waves through LDS.
    // (One thread per chunk walking all PPB lanes was a chain of up to 256 x 8 dependent LDS reads: with 3 channels -- CHP 1,
    // PPB 256 -- that fold was 25 of this kernel's 32 us on the generator's image gradient, profiles/r4h_unet_student_chain.txt.)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (a.L.CHP < 64) {
        for (int o = 32; o >= a.L.CHP; o >>= 1) {
#pragma unroll
            for (int j = 0; j < 8; j++) s[j] += __shfl_xor(s[j], o, 64);
        }
    }
    const int per_wave = a.L.CHP < 64 ? a.L.CHP : 64;          // chunks a wave holds one sum of
    if (lane < per_wave) {
#pragma unroll
        for (int j = 0; j < 8; j++) red[wave * 64 + lane][j] = s[j];
    }
    __syncthreads();
    if (a.L.CHP <= 64) {
        if (threadIdx.x < a.L.CHP && active) {
#pragma unroll
            for (int j = 0; j < 8; j++)
                a.partial[(size_t)blockIdx.x * a.C8 + c0 + j] = (red[ch][j] + red[64 + ch][j]) + (red[128 + ch][j] + red[192 + ch][j]);
        }
    } else if (pl == 0 && active) {                            // 65 .. 256 chunks: PPB <= 2 pixel lanes, each wave holds other chunks
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float t = 0.f;
            for (int q = 0; q < a.L.PPB; q++) t += red[q * a.L.CHP + ch][j];
            a.partial[(size_t)blockIdx.x * a.C8 + c0 + j] = t;
        }
    }
}
__global__ __launch_bounds__(1024) void channel_sum_finalize_kernel(const float* partial, int blocks, int C, int C8, float* out,
                                                                    int accumulate) {
    __shared__ double sh[32][33];
    const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double t = 0.0;
    if (c < C)
        for (int b = pl; b < blocks; b += 32) t += partial[(size_t)b * C8 + c];
    sh[pl][cl] = t;
    __syncthreads();
    if (pl == 0 && c < C) {
        for (int q = 1; q < 32; q++) t += sh[q][cl];
        out[c] = accumulate ? out[c] + (float)t : (float)t;
    }
}

// small tensors (the batch-1 models' bias gradients: <= 128 x 128 pixels): one launch -- a workgroup owns every pixel of 8
// channels, four pixels' loads in flight per thread, wave shuffles + LDS in double.  The two-launch pipeline above costs those
// launch-bound models two host enqueues and two dependent launches per convolution bias.
constexpr int SUM_SMALL_NT = 512;
constexpr size_t SUM_SMALL_MAX_PIXELS = 16384;
__device__ __forceinline__ void channel_sum_small_body(const bf16_t* x, const int ld, const int off, const int C, const int pixels,
                                                       float* out, const int accumulate, const int c0) {
    __shared__ double red[SUM_SMALL_NT / 64][8];
    struct { const bf16_t* x; int ld, off, C; } a = {x, ld, off, C};
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int pix = threadIdx.x;
    for (; pix + 3 * SUM_SMALL_NT < pixels; pix += 4 * SUM_SMALL_NT) {
        i32x4 r[4];
#pragma unroll
        for (int q = 0; q < 4; q++) r[q] = *(const i32x4*)(a.x + (size_t)(pix + q * SUM_SMALL_NT) * a.ld + a.off + c0);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float v[8];
            unpack8(r[q], v);
#pragma unroll
            for (int j = 0; j < 8; j++) s[j] += v[j];
        }
    }
    for (; pix < pixels; pix += SUM_SMALL_NT) {
        float v[8];
        unpack8(*(const i32x4*)(a.x + (size_t)pix * a.ld + a.off + c0), v);
#pragma unroll
        for (int j = 0; j < 8; j++) s[j] += v[j];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = wave_sum(s[j]);
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) red[wave][j] = (double)s[j];
    }
    __syncthreads();
    if (threadIdx.x < 8 && c0 + (int)threadIdx.x < a.C) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < SUM_SMALL_NT / 64; w++) t += red[w][threadIdx.x];
        const int c = c0 + threadIdx.x;
        out[c] = accumulate ? out[c] + (float)t : (float)t;
    }
}
__global__ __launch_bounds__(SUM_SMALL_NT) void channel_sum_small_kernel(const SumArgs a, float* out, int accumulate) {
    channel_sum_small_body(a.x, a.ld, a.off, a.C, (int)a.pixels, out, accumulate, blockIdx.x * 8);
}
// several of them as ONE launch (round 6: the bias gradients of the layers of a grouped weight gradient, engine.WgradCollector --
// CycleGAN's 180 per iteration): the table travels by value in the kernel arguments, a workgroup finds its entry by the prefix
// sums; the arithmetic of an entry is channel_sum_small_kernel's (same bits)
struct SumGroupItem { const bf16_t* x; float* out; int ld, off, C, pixels, accumulate, blk0; };
struct SumGroupArgs { int n, pad_; SumGroupItem it[GCC_CHANSUM_GROUP_MAX]; };
__global__ __launch_bounds__(SUM_SMALL_NT) void channel_sum_group_kernel(const SumGroupArgs g) {
    const int b = blockIdx.x;
    int k = 0;
    for (int i = 1; i < g.n; i++) k = b >= g.it[i].blk0 ? i : k;
    k = __builtin_amdgcn_readfirstlane(k);
    const SumGroupItem& it = g.it[k];
    channel_sum_small_body(it.x, it.ld, it.off, it.C, it.pixels, it.out, it.accumulate, (b - it.blk0) * 8);
}

// per-channel sum / sum of squares in the conv-epilogue partial format: out[group][block][2][C]
struct StatArgs { const bf16_t* x; int ld, off; int C; size_t pixels; Layout L; float* out; int nblocks; };
__global__ __launch_bounds__(256) void channel_stats_g_kernel(const StatArgs a) {
    __shared__ float red[2][256][9];
    const int ch = threadIdx.x & (a.L.CHP - 1);
    const int pl = threadIdx.x >> a.L.sh;
    const int c0 = ch * 8;
    const bool active = ch < a.L.CH;
    const size_t g = blockIdx.y;
    const bf16_t* xg = a.x + g * a.pixels * a.ld;
    float s[8], ss[8];
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = ss[j] = 0.f;
    if (active) {
        for (size_t pix = (size_t)blockIdx.x * a.L.PPB + pl; pix < a.pixels; pix += (size_t)gridDim.x * a.L.PPB) {
            float v[8];
            unpack8(*(const i32x4*)(xg + pix * a.ld + a.off + c0), v);
#pragma unroll
            for (int j = 0; j < 8; j++) { s[j] += v[j]; ss[j] += v[j] * v[j]; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) { red[0][threadIdx.x][j] = s[j]; red[1][threadIdx.x][j] = ss[j]; }
    __syncthreads();
    if (pl == 0 && active) {
        float* o = a.out + (g * a.nblocks + blockIdx.x) * 2 * (size_t)a.C;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (c0 + j < a.C) {
                float t0 = 0.f, t1 = 0.f;
                for (int q = 0; q < a.L.PPB; q++) { t0 += red[0][q * a.L.CHP + ch][j]; t1 += red[1][q * a.L.CHP + ch][j]; }
                o[c0 + j] = t0; o[a.C + c0 + j] = t1;
            }
        }
    }
}

__global__ void gate_mask_kernel(const float* alpha, float tau, float* mask, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float d = alpha[c] - tau;
        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        mask[c] = (sgn + 1.f) * 0.5f;
    }
}

bool aligned8(int a, int b) { return !((a & 7) || (b & 7)); }

}  // namespace

extern "C" int gcc_in_finalize(const float* stats_partial, int tiles_per_group, int groups, int C, double count, float eps,
                               float* mean, float* rstd, float* scale, float* shift, gcc_stream_t stream) {
    GCC_ENTER();
    if (!stats_partial || tiles_per_group <= 0 || groups <= 0 || C <= 0 || count <= 0 || !scale || !shift) return GCC_ERR_BAD_ARG;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 31) / 32, groups), dim3(1024), 0, (hipStream_t)stream, stats_partial,
                       tiles_per_group, C, count, (const float*)nullptr, (const float*)nullptr, eps, 0.f, (float*)nullptr,
                       (float*)nullptr, mean, rstd, scale, shift);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

extern "C" int gcc_inorm_fwd(const void* x, int ldx, void* y, int ldy, const void* residual, int ldr, int C, int HW, int N,
                             int act, float slope, float eps, float* mean, float* rstd, float* scale, float* shift,
                             void* workspace, size_t workspace_bytes, gcc_stream_t stream) {
    GCC_ENTER();
    if (gcc_device_error(0)) return GCC_ERR_LAUNCH;      // a bounded spin of an earlier launch expired: its results were wrong
    if (!x || !y || !mean || !rstd || !scale || !shift || C <= 0 || HW <= 0 || N <= 0 || (ldx & 7) || (ldy & 7) ||
        (residual && (ldr & 7)))
        return GCC_ERR_BAD_ARG;
    InFusedArgs a = {};
    a.x = (const bf16_t*)x; a.ldx = ldx; a.out = (bf16_t*)y; a.ldout = ldy; a.aux = (const bf16_t*)residual; a.ldaux = ldr;
    a.C = C; a.HW = HW; a.act = act; a.slope = slope; a.eps = eps; a.mean = mean; a.rstd = rstd; a.scale = scale; a.shift = shift;
    inorm_launch<false>(a, N, (hipStream_t)stream, workspace, workspace_bytes);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

extern "C" int gcc_inorm_bwd(const void* x, int ldx, const void* y, int ldy, const void* g, int ldg, void* dx, int lddx, int C,
                             int HW, int N, int act, float slope, const float* mean, const float* rstd, void* workspace,
                             size_t workspace_bytes, gcc_stream_t stream) {
    GCC_ENTER();
    if (gcc_device_error(0)) return GCC_ERR_LAUNCH;      // a bounded spin of an earlier launch expired: its results were wrong
    if (!x || !g || !dx || !mean || !rstd || C <= 0 || HW <= 0 || N <= 0 || (ldx & 7) || (ldg & 7) || (lddx & 7) ||
        (y && (ldy & 7)))
        return GCC_ERR_BAD_ARG;
    InFusedArgs a = {};
    a.x = (const bf16_t*)x; a.ldx = ldx; a.y = (const bf16_t*)y; a.ldy = ldy; a.aux = (const bf16_t*)g; a.ldaux = ldg;
    a.out = (bf16_t*)dx; a.ldout = lddx; a.C = C; a.HW = HW; a.act = act; a.slope = slope;
    a.mean = (float*)mean; a.rstd = (float*)rstd;
    inorm_launch<true>(a, N, (hipStream_t)stream, workspace, workspace_bytes);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

// BatchNorm backward (training statistics, no gate / dropout / second gradient) in ONE launch: the grid InstanceNorm backward
// with the whole batch as one plane and gamma folded into the coefficients.  The activation derivative comes from the saved
// output y, or there is no activation (y NULL with an activation would need the affine output recomputed: not this route).
extern "C" int gcc_bn_bwd_one_launch(const void* x, int ldx, const void* y, int ldy, const void* g, int ldg, void* dx, int lddx,
                                     int C, size_t pixels, int act, float slope, const float* mean, const float* rstd,
                                     const float* gamma, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                     gcc_stream_t stream) {
    GCC_ENTER();
    if (gcc_device_error(0)) return GCC_ERR_LAUNCH;
    if (!x || !g || !dx || !mean || !rstd || C <= 0 || pixels == 0 || pixels > (size_t)1 << 30 || (ldx & 7) || (ldg & 7) ||
        (lddx & 7) || (y && (ldy & 7)))
        return GCC_ERR_BAD_ARG;
    if (!y && act != GCC_ACT_NONE) return GCC_ERR_UNSUPPORTED;
    InFusedArgs a = {};
    a.x = (const bf16_t*)x; a.ldx = ldx; a.y = (const bf16_t*)y; a.ldy = ldy; a.aux = (const bf16_t*)g; a.ldaux = ldg;
    a.out = (bf16_t*)dx; a.ldout = lddx; a.C = C; a.HW = (int)pixels; a.act = act; a.slope = slope;
    a.mean = (float*)mean; a.rstd = (float*)rstd; a.gamma = gamma; a.dgamma = dgamma; a.dbeta = dbeta;
    // only the grid form knows gamma / dgamma / dbeta: with GCC_OPT_INORM_GRID = 0, or a geometry its plan refuses, nothing is
    // launched and the caller keeps its three-launch route (the slab kernels would have ignored them silently)
    if (!inorm_launch<true>(a, 1, (hipStream_t)stream, workspace, workspace_bytes, true)) return GCC_ERR_UNSUPPORTED;
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

// The same with what the U-Net's layers need (round 4): y NULL with an activation -- its input is recomputed through the forward's
// affine (scale / shift of the BNState) and the regenerated dropout mask --, a second incoming gradient g2 through act2 (the skip
// path), dropout.  One launch instead of reduce + finalize + apply, also for small tensors (bnact_bwd_small_kernel's C / 8
// workgroups took 10-38 us on the <= 16x16 layers).
extern "C" int gcc_bn_bwd_one_launch_ex(const void* x, int ldx, const void* y, int ldy, const void* g, int ldg, const void* g2, int ldg2,
                                        void* dx, int lddx, int C, size_t pixels, int act, int act2, float slope, float drop_p,
                                        unsigned long long seed, const float* mean, const float* rstd, const float* scale,
                                        const float* shift, const float* gamma, float* dgamma, float* dbeta, void* workspace,
                                        size_t workspace_bytes, gcc_stream_t stream) {
    GCC_ENTER();
    if (gcc_device_error(0)) return GCC_ERR_LAUNCH;
    if (!x || !g || !dx || !mean || !rstd || C <= 0 || pixels == 0 || pixels > (size_t)1 << 30 || (ldx & 7) || (ldg & 7) ||
        (lddx & 7) || (y && (ldy & 7)) || (g2 && (ldg2 & 7)))
        return GCC_ERR_BAD_ARG;
    if (!y && act != GCC_ACT_NONE && (!scale || !shift)) return GCC_ERR_BAD_ARG;
    if (pixels * (size_t)C >= (size_t)1 << 40) return GCC_ERR_UNSUPPORTED;
    InFusedArgs a = {};
    a.x = (const bf16_t*)x; a.ldx = ldx; a.y = (const bf16_t*)y; a.ldy = ldy; a.aux = (const bf16_t*)g; a.ldaux = ldg;
    a.out = (bf16_t*)dx; a.ldout = lddx; a.C = C; a.HW = (int)pixels; a.act = act; a.slope = slope;
    a.mean = (float*)mean; a.rstd = (float*)rstd; a.gamma = gamma; a.dgamma = dgamma; a.dbeta = dbeta;
    a.bscale = y ? nullptr : scale; a.bshift = y ? nullptr : shift;
    a.g2 = (const bf16_t*)g2; a.ldg2 = ldg2; a.act2 = act2; a.drop_p = drop_p; a.seed = seed;
    const bool ran = drop_p > 0.f ? inorm_launch<true, 2>(a, 1, (hipStream_t)stream, workspace, workspace_bytes, true)
                                  : inorm_launch<true, 1>(a, 1, (hipStream_t)stream, workspace, workspace_bytes, true);
    if (!ran) return GCC_ERR_UNSUPPORTED;
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

extern "C" int gcc_bn_finalize(const float* stats_partial, int tiles, int C, double count, const float* gamma,
                               const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                               float* mean, float* rstd, float* scale, float* shift, gcc_stream_t stream) {
    GCC_ENTER();
    if (!stats_partial || tiles <= 0 || C <= 0 || count <= 0 || !scale || !shift) return GCC_ERR_BAD_ARG;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 31) / 32), dim3(1024), 0, (hipStream_t)stream, stats_partial, tiles, C,
                       count, gamma, beta, eps, momentum, running_mean, running_var, mean, rstd, scale, shift);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

extern "C" int gcc_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, int C, float* scale, float* shift,
                                  gcc_stream_t stream) {
    GCC_ENTER();
    if (!running_mean || !running_var || !scale || !shift || C <= 0) return GCC_ERR_BAD_ARG;
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, eps, C, scale, shift);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

extern "C" int gcc_bnact_fwd(const gcc_bnact_t* p, const void* x, int ldx, int xoff, void* y, int ldy, int yoff,
                             void* y2, int ldy2, int y2off, int C, size_t pixels, gcc_stream_t stream) {
    GCC_ENTER();
    if (!p || !x || (!y && !y2) || C <= 0 || pixels == 0) return GCC_ERR_BAD_ARG;
    if (!aligned8(ldx, xoff) || (y && !aligned8(ldy, yoff)) || (y2 && !aligned8(ldy2, y2off))) return GCC_ERR_BAD_ARG;
    FwdArgs a;
    a.p = *p; a.x = (const bf16_t*)x; a.ldx = ldx; a.xoff = xoff;
    a.y = (bf16_t*)y; a.ldy = ldy; a.yoff = yoff; a.y2 = (bf16_t*)y2; a.ldy2 = ldy2; a.y2off = y2off;
    a.C = C; a.pixels = pixels;
    a.groups = p->groups > 1 ? p->groups : 1;
    a.res = (const bf16_t*)p->residual; a.ldres = p->ld_residual;
    if (a.res && (a.ldres & 7)) return GCC_ERR_BAD_ARG;
    if (!make_layout(C, &a.L)) return GCC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(bnact_fwd_kernel, dim3(stream_blocks(pixels, a.L, 2), a.groups), dim3(256), 0, (hipStream_t)stream, a);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

static int reduce_threads() {
    return gcc_opt(GCC_OPT_BN_REDUCE_THREADS) == 1024 ? 1024 : 256;
}
static int bwd_blocks(size_t pixels, const Layout& L) {
    // (L is the 8-channel, 256-thread layout.)  256-thread reduce workgroups sweep PPB/2 pixels at a time: ~16 sweeps
    // each, at most 4 workgroups per CU; the 1024-thread variant: one per CU.  Bounds the partial rows the finalize folds.
    const bool small = reduce_threads() == 256;
    const size_t per = small ? (size_t)L.PPB * 8 : (size_t)L.PPB * 8 * 4;
    size_t b = (pixels + per - 1) / per;
    if (small && b < 512) {
        // Tensors that cannot fill the chip at 16 sweeps per workgroup (the U-Net's <= 32x32 layers): a sweep is one dependent
        // round trip to memory (~1 us with two pixels in flight per thread), so 16 of them made a 15-29 us kernel out of a few
        // hundred KB -- on the generators' backward chain, which runs alone.  Down to 2 sweeps, up to 512 workgroups.
        const size_t lane_pixels = (size_t)L.PPB / 2 > 0 ? (size_t)L.PPB / 2 : 1;      // pixels per sweep of the 4-channel layout
        size_t sweeps = (pixels + lane_pixels * 512 - 1) / (lane_pixels * 512);
        if (sweeps < 2) sweeps = 2;
        if (sweeps > 16) sweeps = 16;
        b = (pixels + lane_pixels * sweeps - 1) / (lane_pixels * sweeps);
    }
    if (b < 1) b = 1;
    const int capv = gcc_opt(GCC_OPT_BN_REDUCE_CAP);
    const size_t cap = small ? (size_t)capv : 256;
    if (b > cap) b = cap;
    return (int)b;
}

constexpr size_t BWD_TICKET_BYTES = 4096;
extern "C" size_t gcc_bnact_bwd_workspace(int C, size_t pixels) {      // per group
    Layout L;
    if (C <= 0 || !make_layout(C, &L)) return 0;
    const int C8 = (C + 7) & ~7;
    // [ticket words of the in-launch finalize: BWD_TICKET_BYTES][partial rows][totals][group sums (double)]
    const size_t blocks = (size_t)bwd_blocks(pixels, L);
    return BWD_TICKET_BYTES + (blocks + 1) * 3 * C8 * sizeof(float) + 64 + ((blocks + FIN_GROUP - 1) / FIN_GROUP) * 3 * C8 * sizeof(double);
}

// extended entry used by the library itself and by the python shim: `in_act` = activation the
// producer already applied to x (plain conv+act layers, bn == 0)
extern "C" int gcc_bnact_bwd_ex(const gcc_bnact_bwd_t* p, int in_act, float in_slope, const void* x, int ldx, int xoff,
                                const void* y, int ldy, int yoff, const void* g1, int ldg1, int g1off, const void* g2,
                                int ldg2, int g2off, void* dx, int lddx, int dxoff, int C, size_t pixels, void* ws,
                                size_t ws_bytes, gcc_stream_t stream) {
    GCC_ENTER();
    if (!p || !x || !g1 || !dx || !ws || C <= 0 || pixels == 0) return GCC_ERR_BAD_ARG;
    if (!aligned8(ldx, xoff) || !aligned8(ldg1, g1off) || !aligned8(lddx, dxoff)) return GCC_ERR_BAD_ARG;
    if ((y && !aligned8(ldy, yoff)) || (g2 && !aligned8(ldg2, g2off))) return GCC_ERR_BAD_ARG;
    if (p->bn && !p->bn_eval && (!p->mean || !p->rstd)) return GCC_ERR_BAD_ARG;
    const int groups = p->groups > 1 ? p->groups : 1;
    if (ws_bytes < gcc_bnact_bwd_workspace(C, pixels) * groups) return GCC_ERR_WORKSPACE;
    if (groups > 1 && (p->dgamma || p->dbeta || p->dalpha)) return GCC_ERR_UNSUPPORTED;
    BwdArgs a;
    a.p = *p;
    a.x = (const bf16_t*)x; a.ldx = ldx; a.xoff = xoff;
    a.y = (const bf16_t*)y; a.ldy = ldy; a.yoff = yoff;
    a.g1 = (const bf16_t*)g1; a.ldg1 = ldg1; a.g1off = g1off;
    a.g2 = (const bf16_t*)g2; a.ldg2 = ldg2; a.g2off = g2off;
    a.dx = (bf16_t*)dx; a.lddx = lddx; a.dxoff = dxoff;
    a.C = C; a.C8 = (C + 7) & ~7; a.pixels = pixels;
    a.in_act = in_act; a.in_slope = in_slope;
    if (!make_layout(C, &a.L)) return GCC_ERR_UNSUPPORTED;
    const int blocks = bwd_blocks(pixels, a.L);
    a.groups = groups; a.nblocks = blocks;
    a.partial = (float*)((char*)ws + BWD_TICKET_BYTES);
    a.totals = a.partial + (size_t)groups * blocks * 3 * a.C8;
    a.tickets = nullptr; a.grp = nullptr;
    // the caller says its workspace was zero-filled when it was allocated and belongs to one stream (flags bit 0): the
    // finalize is then done by the reduce pass's last-arriving workgroups (ticket words at the head of the workspace)
    const int G = (blocks + FIN_GROUP - 1) / FIN_GROUP;
    if ((p->flags & 1) && groups == 1 && (size_t)(G + 1) * 4 <= BWD_TICKET_BYTES) {
        a.tickets = (unsigned*)ws;
        a.grp = (double*)(((uintptr_t)(a.totals + (size_t)3 * a.C8) + 63) & ~(uintptr_t)63);
    }
    hipStream_t st = (hipStream_t)stream;
    const bool gate = p->gate != nullptr || p->dalpha != nullptr || p->gate_after_act;
    const bool drop = p->drop_p > 0.f;
    if (!p->bn && !gate && !drop && !p->dgamma && !p->dbeta && groups == 1 && gcc_opt(GCC_OPT_BN_BWD_SMALL)) {
        hipLaunchKernelGGL(act_bwd_kernel, dim3(stream_blocks(pixels, a.L, 4)), dim3(256), 0, st, a);
        GCC_CHECK_LAUNCH();
        return GCC_OK;
    }
    if (p->bn && !p->bn_eval && !gate && groups == 1 && pixels <= SMALL_MAX_PIXELS && gcc_opt(GCC_OPT_BN_BWD_SMALL)) {
        if (drop) hipLaunchKernelGGL(bnact_bwd_small_kernel<true>, dim3(a.C8 / 8), dim3(SMALL_NT), 0, st, a);
        else hipLaunchKernelGGL(bnact_bwd_small_kernel<false>, dim3(a.C8 / 8), dim3(SMALL_NT), 0, st, a);
        GCC_CHECK_LAUNCH();
        return GCC_OK;
    }
    {
        BwdArgs r = a;                      // the reduce pass runs 4 channels per thread
        if (!make_layout(C, &r.L, 4)) return GCC_ERR_UNSUPPORTED;
        const int nth = reduce_threads();
        if (r.L.CHP > nth) return GCC_ERR_UNSUPPORTED;
        r.L.PPB = nth / r.L.CHP;
#define GCC_LAUNCH_REDUCE(G, D)                                                                                             \
        do {                                                                                                                \
            if (nth == 256) hipLaunchKernelGGL((bnact_bwd_reduce_kernel<G, D, 4, 256>), dim3(blocks, groups), dim3(256), 0, st, r);   \
            else hipLaunchKernelGGL((bnact_bwd_reduce_kernel<G, D, 4, 1024>), dim3(blocks, groups), dim3(1024), 0, st, r);           \
        } while (0)
        if (gate && drop) GCC_LAUNCH_REDUCE(true, true);
        else if (gate) GCC_LAUNCH_REDUCE(true, false);
        else if (drop) GCC_LAUNCH_REDUCE(false, true);
        else GCC_LAUNCH_REDUCE(false, false);
#undef GCC_LAUNCH_REDUCE
    }
    GCC_CHECK_LAUNCH();
    if (!a.tickets) {
        hipLaunchKernelGGL(bnact_bwd_finalize_kernel, dim3((a.C8 + 31) / 32, groups), dim3(1024), 0, st, a, blocks);
        GCC_CHECK_LAUNCH();
    }
    if (p->bn && !p->bn_eval) {
        hipLaunchKernelGGL(bnact_bwd_apply_kernel, dim3(stream_blocks(pixels, a.L, 2), groups), dim3(256), 0, st, a);
        GCC_CHECK_LAUNCH();
    }
    return GCC_OK;
}

extern "C" int gcc_bnact_bwd(const gcc_bnact_bwd_t* p, const void* x, int ldx, int xoff, const void* y, int ldy, int yoff,
                             const void* g1, int ldg1, int g1off, const void* g2, int ldg2, int g2off, void* dx, int lddx,
                             int dxoff, int C, size_t pixels, void* ws, size_t ws_bytes, gcc_stream_t stream) {
    GCC_ENTER();
    return gcc_bnact_bwd_ex(p, GCC_ACT_NONE, 0.f, x, ldx, xoff, y, ldy, yoff, g1, ldg1, g1off, g2, ldg2, g2off, dx, lddx,
                            dxoff, C, pixels, ws, ws_bytes, stream);
}

extern "C" size_t gcc_channel_sum_workspace(int C, size_t pixels) {
    Layout L;
    if (C <= 0 || !make_layout(C, &L)) return 0;
    return (size_t)bwd_blocks(pixels, L) * ((C + 7) & ~7) * sizeof(float);
}

extern "C" int gcc_channel_sum(const void* x, int ld, int off, int C, size_t pixels, float* out, int accumulate,
                               void* ws, size_t ws_bytes, gcc_stream_t stream) {
    GCC_ENTER();
    if (!x || !out || !ws || C <= 0 || pixels == 0 || !aligned8(ld, off)) return GCC_ERR_BAD_ARG;
    if (ws_bytes < gcc_channel_sum_workspace(C, pixels)) return GCC_ERR_WORKSPACE;
    SumArgs a;
    a.x = (const bf16_t*)x; a.ld = ld; a.off = off; a.C = C; a.C8 = (C + 7) & ~7; a.pixels = pixels;
    a.partial = (float*)ws;
    if (!make_layout(C, &a.L)) return GCC_ERR_UNSUPPORTED;
    const int blocks = bwd_blocks(pixels, a.L);
    hipStream_t st = (hipStream_t)stream;
    if (pixels <= SUM_SMALL_MAX_PIXELS && gcc_opt(GCC_OPT_BN_BWD_SMALL)) {
        hipLaunchKernelGGL(channel_sum_small_kernel, dim3(a.C8 / 8), dim3(SUM_SMALL_NT), 0, st, a, out, accumulate);
        GCC_CHECK_LAUNCH();
        return GCC_OK;
    }
    hipLaunchKernelGGL(channel_sum_kernel, dim3(blocks), dim3(256), 0, st, a);
    GCC_CHECK_LAUNCH();
    hipLaunchKernelGGL(channel_sum_finalize_kernel, dim3((C + 31) / 32), dim3(1024), 0, st, (const float*)ws, blocks, C,
                       a.C8, out, accumulate);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

extern "C" int gcc_channel_sum_group(const gcc_chansum_item_t* items, int n, gcc_stream_t stream) {
    GCC_ENTER();
    if (!items || n < 1 || n > GCC_CHANSUM_GROUP_MAX) return GCC_ERR_BAD_ARG;
    SumGroupArgs g;
    g.n = n; g.pad_ = 0;
    int blk = 0;
    for (int i = 0; i < n; i++) {
        const gcc_chansum_item_t& s = items[i];
        if (!s.x || !s.out || s.C <= 0 || s.pixels == 0 || !aligned8(s.ld, s.off)) return GCC_ERR_BAD_ARG;
        if (s.pixels > SUM_SMALL_MAX_PIXELS) return GCC_ERR_UNSUPPORTED;          // larger tensors: gcc_channel_sum's two-launch form
        g.it[i] = SumGroupItem{(const bf16_t*)s.x, s.out, s.ld, s.off, s.C, (int)s.pixels, s.accumulate, blk};
        blk += ((s.C + 7) & ~7) / 8;
    }
    hipLaunchKernelGGL(channel_sum_group_kernel, dim3(blk), dim3(SUM_SMALL_NT), 0, (hipStream_t)stream, g);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

extern "C" int gcc_gate_mask(const float* alpha, float tau, float* mask, int C, gcc_stream_t stream) {
    GCC_ENTER();
    if (!alpha || !mask || C <= 0) return GCC_ERR_BAD_ARG;
    hipLaunchKernelGGL(gate_mask_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, alpha, tau, mask, C);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

// statistics pass for tensors that do not come out of a convolution epilogue (depthwise conv outputs):
// writes gcc_channel_stats_tiles() partial rows per group in the [tile][2][C] format gcc_*_finalize folds
extern "C" int gcc_channel_stats_tiles(size_t pixels_per_group, int C) {
    Layout L;
    if (C <= 0 || !make_layout(C, &L)) return 0;
    size_t b = (pixels_per_group + (size_t)L.PPB * 8 - 1) / ((size_t)L.PPB * 8);
    if (b < 1) b = 1;
    if (b > 256) b = 256;
    return (int)b;
}
extern "C" int gcc_channel_stats(const void* x, int ld, int off, int C, size_t pixels_per_group, int groups, float* stats,
                                 gcc_stream_t stream) {
    GCC_ENTER();
    if (!x || !stats || C <= 0 || pixels_per_group == 0 || groups <= 0 || !aligned8(ld, off)) return GCC_ERR_BAD_ARG;
    StatArgs a;
    a.x = (const bf16_t*)x; a.ld = ld; a.off = off; a.C = C; a.pixels = pixels_per_group; a.out = stats;
    if (!make_layout(C, &a.L)) return GCC_ERR_UNSUPPORTED;
    a.nblocks = gcc_channel_stats_tiles(pixels_per_group, C);
    hipLaunchKernelGGL(channel_stats_g_kernel, dim3(a.nblocks, groups), dim3(256), 0, (hipStream_t)stream, a);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}
