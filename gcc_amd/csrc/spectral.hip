// Spectral normalisation of a conv weight (reference models/SAGAN.py:17-70) on the fp32 master.
//
// The master W_bar is a 4-D nn.Parameter [R][C][k][k] stored channels_last (k > 1): element (r, c, t) sits at
// r*C*T + t*C + c (T = k*k taps).  The reference flattens it as w.view(R, -1), i.e. column j = c*T + t; u has R
// entries, v has C*T entries in THAT order (state_dict compatible).  All kernels below walk the physical layout
// (coalesced) and translate to the logical column index where v is touched.
//
//   power iteration (every forward, train or eval):
//       vt = W^T u ;  v = vt / (|vt| + 1e-12) ;  t = W v ;  u = t / (|t| + 1e-12) ;  sigma = u . t
//       W_eff = W_bar / sigma                      (fp32, same layout; packed to bf16 by gcc_pack_weights)
//   gradient through W_eff = W_bar / sigma(W_bar), with G = dL/dW_eff:
//       inner = <G, W_bar> ;  dL/dsigma = -inner / sigma^2
//       dW_bar += G / sigma + dL/dsigma * u v^T
//       du     += dL/dsigma * t        (t = W_bar v of the forward call the gradient belongs to)
//       dv     += dL/dsigma * W_bar^T u
//   u, v in the gradient are the LIVE vectors (latest power iteration), sigma and t are the forward call's own: this is
//   what the reference's autograd evaluates, because `u.data = ...` re-points the tensors earlier graphs saved.
#include "common.hpp"

namespace {

struct SnArgs {
    const float* w;       // W_bar
    float* u;             // [R]
    float* v;             // [C*T], logical order
    int R, C, T;
    float* vt;            // workspace [C*T], physical order
    float* t;             // [R]  (kept per forward call)
    float* scal;          // [4]: 0 |vt|^2, 1 |t|^2, 2 sigma, 3 inner
    float* w_eff;         // [R*C*T]
    float* sigma_out;     // the caller's copy of sigma (gcc_spectral_power_iteration_pack), or NULL (copied by the entry point)
};

__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) s += sh[i];
    return s;
}

// vt[p] = sum_r W[r][p] * u[r] * scale_by   (p physical column).
// A workgroup owns 64 columns; its 16 waves split the rows (a wave reads 256 contiguous bytes of one row), so the serial
// depth is R / 16 and the sum order is fixed.
__device__ __forceinline__ void sn_wtu_body(const float* __restrict__ w, const float* __restrict__ u, int R, int K,
                                            float* __restrict__ vt, float scale_by, int block, float (*red)[65]) {
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int p = block * 64 + cl;
    float a0 = 0.f, a1 = 0.f;
    if (p < K) {
        int r = rg;
        for (; r + 16 < R; r += 32) {
            a0 += w[(size_t)r * K + p] * u[r];
            a1 += w[(size_t)(r + 16) * K + p] * u[r + 16];
        }
        if (r < R) a0 += w[(size_t)r * K + p] * u[r];
    }
    red[rg][cl] = a0 + a1;
    __syncthreads();
    if (rg == 0) {
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < 16; q++) acc += red[q][cl];
        acc *= scale_by;
        if (p < K) vt[p] = acc;
    }
}
__global__ __launch_bounds__(1024) void sn_wtu_kernel(const float* __restrict__ w, const float* __restrict__ u, int R, int K,
                                                      float* __restrict__ vt, float scale_by) {
    __shared__ float red[16][65];
    sn_wtu_body(w, u, R, K, vt, scale_by, blockIdx.x, red);
}

// t[r] = sum_p W[r][p] * vt[p] / (|vt| + eps).  Every row's workgroup sums |vt|^2 itself, in the same fixed order: no
// atomics anywhere in the power iteration, so sigma -- and every weight divided by it -- is reproducible bit for bit.
__device__ __forceinline__ void sn_wv_body(const float* __restrict__ w, const float* __restrict__ vt, int K, float* scal,
                                           float* __restrict__ t, int r, float* sh) {
    float acc = 0.f, n2 = 0.f;
    for (int p = threadIdx.x; p < K; p += 256) {
        const float x = vt[p];
        acc += w[(size_t)r * K + p] * x;
        n2 += x * x;
    }
    n2 = block_sum(n2, sh);
    const float s = block_sum(acc, sh) / (sqrtf(n2) + 1e-12f);
    if (threadIdx.x == 0) {
        t[r] = s;
        if (r == 0) scal[0] = n2;
    }
}
__global__ __launch_bounds__(256) void sn_wv_kernel(const float* __restrict__ w, const float* __restrict__ vt, int K,
                                                    float* scal, float* __restrict__ t) {
    __shared__ float sh[4];
    sn_wv_body(w, vt, K, scal, t, blockIdx.x, sh);
}

// u, v, sigma
__device__ __forceinline__ void sn_finalize_body(const SnArgs& a, int block, int nblk, float* sh) {
    const int K = a.C * a.T;
    float t2 = 0.f;
    for (int r = threadIdx.x; r < a.R; r += 256) t2 += a.t[r] * a.t[r];
    t2 = block_sum(t2, sh);                      // |t|^2, the same bits in every workgroup
    const float inv_v = 1.f / (sqrtf(a.scal[0]) + 1e-12f);
    const float inv_t = 1.f / (sqrtf(t2) + 1e-12f);
    for (int p = block * 256 + threadIdx.x; p < K; p += nblk * 256) {
        const int tt = p / a.C, c = p - tt * a.C;
        a.v[c * a.T + tt] = a.vt[p] * inv_v;
    }
    if (block == 0) {
        for (int r = threadIdx.x; r < a.R; r += 256) a.u[r] = a.t[r] * inv_t;
        if (threadIdx.x == 0) {
            a.scal[1] = t2; a.scal[2] = t2 * inv_t;                            // u . t = |t|^2 / (|t| + eps)
            if (a.sigma_out) a.sigma_out[0] = t2 * inv_t;
        }
    }
}
__global__ __launch_bounds__(256) void sn_finalize_kernel(SnArgs a) {
    __shared__ float sh[4];
    sn_finalize_body(a, blockIdx.x, gridDim.x, sh);
}

// W_eff = W_bar / sigma straight into the two bf16 packings the convolutions read (W [R][T][Cp], Wt [C][T][Rp], channel counts
// padded to 8 with zeros): a 32 x 32 tile of one tap per workgroup, transposed through LDS -- the scale launch, the fp32 W_eff
// round trip and the two pack launches of the separate route in one (round 4: the SAGAN iteration is a chain of ~5 us launches;
// this is 3 of the 7 a spectrally normalised convolution's forward spent before its convolution).  Same bits: bf16(w * (1/sigma)).
__device__ __forceinline__ void sn_scale_pack_body(const float* __restrict__ m, const float* scal, int rows, int taps, int cols,
                                                   int colsp, int rowsp, unsigned short* __restrict__ w,
                                                   unsigned short* __restrict__ wt, int bx, int by, int tap, float (*t)[33]) {
    const float inv = 1.f / scal[2];
    const int r0 = by * 32, c0 = bx * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        const float v = (r < rows && c < cols) ? m[((size_t)r * taps + tap) * cols + c] * inv : 0.f;
        t[k][tx] = v;
        if (r < rows && c < colsp) w[((size_t)r * taps + tap) * colsp + c] = f2bf(v);          // columns cols..colsp-1: zeros
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (c < cols && r < rowsp) wt[((size_t)c * taps + tap) * rowsp + r] = f2bf(t[tx][k]);   // rows rows..rowsp-1: zeros
    }
}
__global__ __launch_bounds__(256) void sn_scale_pack_kernel(const float* __restrict__ m, const float* scal, int rows, int taps,
                                                            int cols, int colsp, int rowsp, unsigned short* __restrict__ w,
                                                            unsigned short* __restrict__ wt) {
    __shared__ float t[32][33];
    sn_scale_pack_body(m, scal, rows, taps, cols, colsp, rowsp, w, wt, blockIdx.x, blockIdx.y, blockIdx.z, t);
}

// ---- the power iterations of several layers (every spectrally normalised convolution of one network's forward pass) as FOUR
// launches instead of four per layer (round 6): a workgroup finds its layer by its block number; per layer the arithmetic, its
// order and the workgroup shapes are those of the kernels above, so the results are the same bits.
struct SnGroupItem {
    SnArgs a;
    unsigned short* pw; unsigned short* pwt;
    int K, colsp, rowsp, gx, gy;
    int b1, b2, b3, b4;            // this layer's first block in the four launches
    int n3;                        // its blocks in the third one
};
struct SnGroupArgs { int n; int pad; SnGroupItem it[GCC_SPECTRAL_GROUP_MAX]; };
__device__ __forceinline__ int sn_group_find(const SnGroupArgs& g, int b, int phase) {
    int i = 0;
    for (int q = 1; q < g.n; q++) {
        const int first = phase == 1 ? g.it[q].b1 : (phase == 2 ? g.it[q].b2 : (phase == 3 ? g.it[q].b3 : g.it[q].b4));
        if (b >= first) i = q;
    }
    return i;
}
__global__ __launch_bounds__(1024) void sn_wtu_group_kernel(const SnGroupArgs g) {
    __shared__ float red[16][65];
    const int i = sn_group_find(g, blockIdx.x, 1);
    const SnGroupItem& it = g.it[i];
    sn_wtu_body(it.a.w, it.a.u, it.a.R, it.K, it.a.vt, 1.f, blockIdx.x - it.b1, red);
}
__global__ __launch_bounds__(256) void sn_wv_group_kernel(const SnGroupArgs g) {
    __shared__ float sh[4];
    const int i = sn_group_find(g, blockIdx.x, 2);
    const SnGroupItem& it = g.it[i];
    sn_wv_body(it.a.w, it.a.vt, it.K, it.a.scal, it.a.t, blockIdx.x - it.b2, sh);
}
__global__ __launch_bounds__(256) void sn_finalize_group_kernel(const SnGroupArgs g) {
    __shared__ float sh[4];
    const int i = sn_group_find(g, blockIdx.x, 3);
    const SnGroupItem& it = g.it[i];
    sn_finalize_body(it.a, blockIdx.x - it.b3, it.n3, sh);
}
__global__ __launch_bounds__(256) void sn_scale_pack_group_kernel(const SnGroupArgs g) {
    __shared__ float t[32][33];
    const int i = sn_group_find(g, blockIdx.x, 4);
    const SnGroupItem& it = g.it[i];
    int b = blockIdx.x - it.b4;
    const int bx = b % it.gx; b /= it.gx;
    const int by = b % it.gy, tap = b / it.gy;
    sn_scale_pack_body(it.a.w, it.a.scal, it.a.R, it.a.T, it.a.C, it.colsp, it.rowsp, it.pw, it.pwt, bx, by, tap, t);
}

__global__ __launch_bounds__(256) void sn_scale_kernel(const float* __restrict__ w, const float* scal, float* __restrict__ w_eff,
                                                       size_t n) {
    const float inv = 1.f / scal[2];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) w_eff[i] = w[i] * inv;
}

// <G, W_bar>: per-workgroup partial sums; the consumers add them up in a fixed order (no atomics: reproducible)
__global__ __launch_bounds__(256) void sn_inner_kernel(const float* __restrict__ g, const float* __restrict__ w, size_t n,
                                                       float* partial) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += g[i] * w[i];
    const float s = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__device__ __forceinline__ float inner_total(const float* partial, int nparts, float* sh) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += partial[i];
    return block_sum(acc, sh);
}

// dW_bar += G/sigma + dLds * u[r] * v[logical(p)] ;  du += dLds * t  (block 0)
__global__ __launch_bounds__(256) void sn_grad_kernel(const float* __restrict__ g, const float* __restrict__ u,
                                                      const float* __restrict__ v, const float* __restrict__ t_fwd,
                                                      const float* sigma_fwd, const float* partial, int nparts, int R, int C,
                                                      int T, float* __restrict__ dw, float* __restrict__ du) {
    __shared__ float sh[4];
    const float inner = inner_total(partial, nparts, sh);
    const float sg = sigma_fwd[0];
    const float inv = 1.f / sg, dlds = -inner * inv * inv;
    const int K = C * T;
    const size_t n = (size_t)R * K;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / K), p = (int)(i - (size_t)r * K);
        const int tt = p / C, c = p - tt * C;
        dw[i] += g[i] * inv + dlds * u[r] * v[c * T + tt];
    }
    if (du && blockIdx.x == 0)
        for (int r = threadIdx.x; r < R; r += 256) du[r] += dlds * t_fwd[r];
}

// dv[logical(p)] += dLds * vt[p]   with vt = W_bar^T u (physical order)
__global__ __launch_bounds__(256) void sn_dv_kernel(const float* __restrict__ vt, const float* sigma_fwd, const float* partial,
                                                    int nparts, int C, int T, float* __restrict__ dv) {
    __shared__ float sh[4];
    const float inner = inner_total(partial, nparts, sh);
    const float inv = 1.f / sigma_fwd[0], dlds = -inner * inv * inv;
    const int K = C * T;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < K; p += gridDim.x * 256) {
        const int tt = p / C, c = p - tt * C;
        dv[c * T + tt] += dlds * vt[p];
    }
}

int nblocks(size_t n, int cap = 1024) {
    size_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > (size_t)cap ? cap : b));
}

}  // namespace

extern "C" size_t gcc_spectral_workspace(int R, int C, int T) {
    if (R <= 0 || C <= 0 || T <= 0) return 0;
    return ((size_t)C * T + 64 + 1024) * sizeof(float);     // scalars | W^T u | partial sums of <G, W_bar>
}

extern "C" int gcc_spectral_power_iteration(const float* w_bar, float* u, float* v, int R, int C, int T, float* t_out,
                                            float* sigma_out, float* w_eff, void* ws, size_t ws_bytes, gcc_stream_t stream) {
    GCC_ENTER();
    if (!w_bar || !u || !v || !t_out || !sigma_out || !w_eff || !ws || R <= 0 || C <= 0 || T <= 0) return GCC_ERR_BAD_ARG;
    if (ws_bytes < gcc_spectral_workspace(R, C, T)) return GCC_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int K = C * T;
    float* scal = (float*)ws;                 // [0..3]
    float* vt = scal + 64;
    hipLaunchKernelGGL(sn_wtu_kernel, dim3((K + 63) / 64), dim3(1024), 0, st, w_bar, (const float*)u, R, K, vt, 1.f);
    GCC_CHECK_LAUNCH();
    hipLaunchKernelGGL(sn_wv_kernel, dim3(R), dim3(256), 0, st, w_bar, (const float*)vt, K, scal, t_out);
    GCC_CHECK_LAUNCH();
    SnArgs a;
    a.w = w_bar; a.u = u; a.v = v; a.R = R; a.C = C; a.T = T; a.vt = vt; a.t = t_out; a.scal = scal; a.w_eff = w_eff;
    a.sigma_out = nullptr;
    hipLaunchKernelGGL(sn_finalize_kernel, dim3(nblocks(K, 64)), dim3(256), 0, st, a);
    GCC_CHECK_LAUNCH();
    const size_t n = (size_t)R * K;
    hipLaunchKernelGGL(sn_scale_kernel, dim3(nblocks(n)), dim3(256), 0, st, w_bar, (const float*)scal, w_eff, n);
    GCC_CHECK_LAUNCH();
    if (gcc_memcpy_d2d_async(sigma_out, scal + 2, sizeof(float), st) != hipSuccess) return GCC_ERR_LAUNCH;
    return GCC_OK;
}

extern "C" int gcc_spectral_power_iteration_pack(const float* w_bar, float* u, float* v, int R, int C, int T, float* t_out,
                                                 float* sigma_out, void* w, void* wt, void* ws, size_t ws_bytes,
                                                 gcc_stream_t stream) {
    GCC_ENTER();
    if (!w_bar || !u || !v || !t_out || !sigma_out || !w || !wt || !ws || R <= 0 || C <= 0 || T <= 0) return GCC_ERR_BAD_ARG;
    if (ws_bytes < gcc_spectral_workspace(R, C, T)) return GCC_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int K = C * T;
    float* scal = (float*)ws;
    float* vt = scal + 64;
    hipLaunchKernelGGL(sn_wtu_kernel, dim3((K + 63) / 64), dim3(1024), 0, st, w_bar, (const float*)u, R, K, vt, 1.f);
    GCC_CHECK_LAUNCH();
    hipLaunchKernelGGL(sn_wv_kernel, dim3(R), dim3(256), 0, st, w_bar, (const float*)vt, K, scal, t_out);
    GCC_CHECK_LAUNCH();
    SnArgs a;
    a.w = w_bar; a.u = u; a.v = v; a.R = R; a.C = C; a.T = T; a.vt = vt; a.t = t_out; a.scal = scal; a.w_eff = nullptr;
    a.sigma_out = sigma_out;
    hipLaunchKernelGGL(sn_finalize_kernel, dim3(nblocks(K, 64)), dim3(256), 0, st, a);
    GCC_CHECK_LAUNCH();
    const int colsp = (C + 7) & ~7, rowsp = (R + 7) & ~7;
    hipLaunchKernelGGL(sn_scale_pack_kernel, dim3((colsp + 31) / 32, (rowsp + 31) / 32, T), dim3(256), 0, st, w_bar,
                       (const float*)scal, R, T, C, colsp, rowsp, (unsigned short*)w, (unsigned short*)wt);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

static size_t sn_item_ws(const gcc_sn_item_t& it) { return (gcc_spectral_workspace(it.R, it.C, it.T) + 255) & ~(size_t)255; }
extern "C" size_t gcc_spectral_group_workspace(const gcc_sn_item_t* items, int n) {
    if (!items || n < 1 || n > GCC_SPECTRAL_GROUP_MAX) return 0;
    size_t total = 0;
    for (int i = 0; i < n; i++) {
        if (items[i].R <= 0 || items[i].C <= 0 || items[i].T <= 0) return 0;
        total += sn_item_ws(items[i]);
    }
    return total;
}

extern "C" int gcc_spectral_power_iteration_pack_group(const gcc_sn_item_t* items, int n, void* ws, size_t ws_bytes,
                                                       gcc_stream_t stream) {
    GCC_ENTER();
    if (!items || n < 1 || n > GCC_SPECTRAL_GROUP_MAX || !ws || (((uintptr_t)ws) & 15)) return GCC_ERR_BAD_ARG;
    const size_t need = gcc_spectral_group_workspace(items, n);
    if (!need) return GCC_ERR_BAD_ARG;
    if (ws_bytes < need) return GCC_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    SnGroupArgs g;
    g.n = n; g.pad = 0;
    int b1 = 0, b2 = 0, b3 = 0, b4 = 0;
    char* wsp = (char*)ws;
    for (int i = 0; i < n; i++) {
        const gcc_sn_item_t& q = items[i];
        if (!q.w_bar || !q.u || !q.v || !q.t_out || !q.sigma_out || !q.w || !q.wt) return GCC_ERR_BAD_ARG;
        SnGroupItem& it = g.it[i];
        float* scal = (float*)wsp;
        it.a.w = q.w_bar; it.a.u = q.u; it.a.v = q.v; it.a.R = q.R; it.a.C = q.C; it.a.T = q.T;
        it.a.vt = scal + 64; it.a.t = q.t_out; it.a.scal = scal; it.a.w_eff = nullptr; it.a.sigma_out = q.sigma_out;
        it.pw = (unsigned short*)q.w; it.pwt = (unsigned short*)q.wt;
        it.K = q.C * q.T; it.colsp = (q.C + 7) & ~7; it.rowsp = (q.R + 7) & ~7;
        it.gx = (it.colsp + 31) / 32; it.gy = (it.rowsp + 31) / 32;
        it.b1 = b1; it.b2 = b2; it.b3 = b3; it.b4 = b4;
        it.n3 = nblocks(it.K, 64);
        b1 += (it.K + 63) / 64; b2 += q.R; b3 += it.n3; b4 += it.gx * it.gy * q.T;
        wsp += sn_item_ws(q);
    }
    hipLaunchKernelGGL(sn_wtu_group_kernel, dim3(b1), dim3(1024), 0, st, g);
    GCC_CHECK_LAUNCH();
    hipLaunchKernelGGL(sn_wv_group_kernel, dim3(b2), dim3(256), 0, st, g);
    GCC_CHECK_LAUNCH();
    hipLaunchKernelGGL(sn_finalize_group_kernel, dim3(b3), dim3(256), 0, st, g);
    GCC_CHECK_LAUNCH();
    hipLaunchKernelGGL(sn_scale_pack_group_kernel, dim3(b4), dim3(256), 0, st, g);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

extern "C" int gcc_spectral_grad(const float* g_eff, const float* w_bar, const float* u, const float* v, const float* t_fwd,
                                 const float* sigma_fwd, int R, int C, int T, float* dw_bar, float* du, float* dv, void* ws,
                                 size_t ws_bytes, gcc_stream_t stream) {
    GCC_ENTER();
    if (!g_eff || !w_bar || !u || !v || !t_fwd || !sigma_fwd || !dw_bar || !ws || R <= 0 || C <= 0 || T <= 0) return GCC_ERR_BAD_ARG;
    if (ws_bytes < gcc_spectral_workspace(R, C, T)) return GCC_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int K = C * T;
    const size_t n = (size_t)R * K;
    float* scal = (float*)ws;
    float* vt = scal + 64;
    float* partial = vt + K;
    const int nb = nblocks(n);
    hipLaunchKernelGGL(sn_inner_kernel, dim3(nb), dim3(256), 0, st, g_eff, w_bar, n, partial);
    GCC_CHECK_LAUNCH();
    hipLaunchKernelGGL(sn_grad_kernel, dim3(nb), dim3(256), 0, st, g_eff, u, v, t_fwd, sigma_fwd, (const float*)partial, nb,
                       R, C, T, dw_bar, du);
    GCC_CHECK_LAUNCH();
    if (dv) {
        hipLaunchKernelGGL(sn_wtu_kernel, dim3((K + 63) / 64), dim3(1024), 0, st, w_bar, u, R, K, vt, 1.f);
        GCC_CHECK_LAUNCH();
        hipLaunchKernelGGL(sn_dv_kernel, dim3(nblocks(K, 64)), dim3(256), 0, st, (const float*)vt, sigma_fwd,
                           (const float*)partial, nb, C, T, dv);
        GCC_CHECK_LAUNCH();
    }
    return GCC_OK;
}
