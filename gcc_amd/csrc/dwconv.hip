// MobileResnet pieces that are not GEMM-shaped (reference models/Pix2Pix.py:132-197, 215-262):
//   * ReflectionPad2d as an explicit NHWC copy (forward) and its adjoint (backward: every interior pixel
//     gathers the padded positions that mirror onto it) -- used in front of the two 7x7 convolutions;
//   * depthwise 3x3 convolution with ReflectionPad2d(1) fused: forward, backward-data, backward-weight.
// All HBM-bound streaming kernels: a thread owns a fixed 4-channel slice (its 9x4 taps stay in registers)
// and walks pixels; 8-byte accesses.
#include "common.hpp"

namespace {

__device__ __forceinline__ int reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

__device__ __forceinline__ void ld4(const bf16_t* p, float* f) {
    const i32x2 v = *(const i32x2*)p;
    f[0] = __uint_as_float((uint32_t)v[0] << 16); f[1] = __uint_as_float((uint32_t)v[0] & 0xffff0000u);
    f[2] = __uint_as_float((uint32_t)v[1] << 16); f[3] = __uint_as_float((uint32_t)v[1] & 0xffff0000u);
}
__device__ __forceinline__ void st4(bf16_t* p, const float* f) {
    i32x2 v;
    v[0] = (int)pack2bf(f[0], f[1]); v[1] = (int)pack2bf(f[2], f[3]);
    *(i32x2*)p = v;
}

struct PadArgs {
    const bf16_t* src; bf16_t* dst; int N, H, W, pad, lds, ldd, CH;   // CH = 8-channel chunks
};
// dst[n, y, x] = src[n, refl(y - pad), refl(x - pad)] ; dst is (H+2p) x (W+2p)
__global__ void reflect_pad_kernel(const PadArgs a) {
    const int Hp = a.H + 2 * a.pad, Wp = a.W + 2 * a.pad;
    const size_t total = (size_t)a.N * Hp * Wp * a.CH;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % a.CH);
        size_t p = i / a.CH;
        const int x = (int)(p % Wp); p /= Wp;
        const int y = (int)(p % Hp);
        const int n = (int)(p / Hp);
        const int sy = reflect(y - a.pad, a.H), sx = reflect(x - a.pad, a.W);
        *(i32x4*)(a.dst + ((size_t)(n * Hp + y) * Wp + x) * a.ldd + ch * 8) =
            *(const i32x4*)(a.src + ((size_t)(n * a.H + sy) * a.W + sx) * a.lds + ch * 8);
    }
}
// adjoint: dsrc[n, y, x] = sum of dpad over the padded positions that read (y, x)
__global__ void reflect_pad_bwd_kernel(const PadArgs a) {   // src = d(padded), dst = d(unpadded)
    const int Hp = a.H + 2 * a.pad, Wp = a.W + 2 * a.pad;
    const size_t total = (size_t)a.N * a.H * a.W * a.CH;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % a.CH);
        size_t p = i / a.CH;
        const int x = (int)(p % a.W); p /= a.W;
        const int y = (int)(p % a.H);
        const int n = (int)(p / a.H);
        // padded rows mapping onto y: y+pad always; -y+pad if 1 <= y <= pad; 2(H-1)-y+pad if H-1-pad <= y <= H-2
        int ys[3], xs[3], ny = 0, nx = 0;
        ys[ny++] = y + a.pad;
        if (y >= 1 && y <= a.pad) ys[ny++] = a.pad - y;
        if (y <= a.H - 2 && y >= a.H - 1 - a.pad) ys[ny++] = 2 * (a.H - 1) - y + a.pad;
        xs[nx++] = x + a.pad;
        if (x >= 1 && x <= a.pad) xs[nx++] = a.pad - x;
        if (x <= a.W - 2 && x >= a.W - 1 - a.pad) xs[nx++] = 2 * (a.W - 1) - x + a.pad;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int iy = 0; iy < ny; iy++)
            for (int ix = 0; ix < nx; ix++) {
                float v[8];
                unpack8(*(const i32x4*)(a.src + ((size_t)(n * Hp + ys[iy]) * Wp + xs[ix]) * a.lds + ch * 8), v);
#pragma unroll
                for (int j = 0; j < 8; j++) acc[j] += v[j];
            }
        *(i32x4*)(a.dst + ((size_t)(n * a.H + y) * a.W + x) * a.ldd + ch * 8) = pack8(acc);
    }
}

// ------------------------------------------------------------------------------------------------
struct DwArgs {
    const bf16_t* x; int ldx;       // input  [N,H,W,ldx]
    const bf16_t* dy; int lddy;     // output gradient (backward)
    bf16_t* y; int ldy;             // output / dx
    const float* w;                 // [C][9] fp32 master (nn.Conv2d(C, C, 3, groups=C).weight)
    const float* bias;              // [C] or null
    float* partial;                 // backward-weight: [blocks][10][C4]  (9 taps + bias)
    int N, H, W, C, CH4, CHP, sh, PPB;
};

template <int MODE>   // 0 forward, 1 backward-data
__global__ __launch_bounds__(256) void dwconv_kernel(const DwArgs a) {
    const int ch = threadIdx.x & (a.CHP - 1);
    const int pl = threadIdx.x >> a.sh;
    if (ch >= a.CH4) return;
    const int c0 = ch * 4;
    float w[9][4], b[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const bool v = c0 + j < a.C;
#pragma unroll
        for (int t = 0; t < 9; t++) w[t][j] = v ? a.w[(size_t)(c0 + j) * 9 + t] : 0.f;
        b[j] = (v && a.bias && MODE == 0) ? a.bias[c0 + j] : 0.f;
    }
    const size_t pixels = (size_t)a.N * a.H * a.W;
    for (size_t pix = (size_t)blockIdx.x * a.PPB + pl; pix < pixels; pix += (size_t)gridDim.x * a.PPB) {
        const int x = (int)(pix % a.W);
        const size_t r = pix / a.W;
        const int y = (int)(r % a.H);
        const size_t nb = (r / a.H) * a.H * a.W;
        float acc[4] = {b[0], b[1], b[2], b[3]};
        if (MODE == 0) {
#pragma unroll
            for (int ky = 0; ky < 3; ky++) {
                const int sy = reflect(y + ky - 1, a.H);
#pragma unroll
                for (int kx = 0; kx < 3; kx++) {
                    const int sx = reflect(x + kx - 1, a.W);
                    float v[4];
                    ld4(a.x + (nb + (size_t)sy * a.W + sx) * a.ldx + c0, v);
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[j] += w[ky * 3 + kx][j] * v[j];
                }
            }
        } else if (y >= 2 && y < a.H - 2 && x >= 2 && x < a.W - 2) {
            // interior (no mirrored coordinate lands here, every tap's source exists): the nine loads issue together, same
            // tap order as the general form below -- the border walk with its data-dependent trip counts made this kernel
            // twice as slow as the forward one (18.8 against 8.6 us on the CycleGAN planes; 94 % of a 64 x 64 plane is interior)
#pragma unroll
            for (int ky = 0; ky < 3; ky++) {
#pragma unroll
                for (int kx = 0; kx < 3; kx++) {
                    float v[4];
                    ld4(a.dy + (nb + (size_t)(y + 1 - ky) * a.W + (x + 1 - kx)) * a.lddy + c0, v);
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[j] += w[ky * 3 + kx][j] * v[j];
                }
            }
        } else if (a.H >= 4 && a.W >= 4) {
            // border pixel of a plane with at least four rows and columns: at most two candidate rows (the pixel's own and one
            // mirror) and two candidate columns -- 4 x 9 loads from clamped addresses, all in flight, contributions that do not
            // exist multiplied by zero.  (With branches around every load the border waves, 12 % of a 64 x 64 plane, set the
            // kernel's duration: 13 us against the forward's 6.)
            const int uy[2] = {y, y == 1 ? -1 : a.H}, vx[2] = {x, x == 1 ? -1 : a.W};
            const bool uo[2] = {true, y == 1 || y == a.H - 2}, vo[2] = {true, x == 1 || x == a.W - 2};
#pragma unroll
            for (int iu = 0; iu < 2; iu++)
#pragma unroll
                for (int iv = 0; iv < 2; iv++) {
#pragma unroll
                    for (int ky = 0; ky < 3; ky++) {
                        const int py = uy[iu] + 1 - ky;
                        const int pyc = py < 0 ? 0 : (py >= a.H ? a.H - 1 : py);
#pragma unroll
                        for (int kx = 0; kx < 3; kx++) {
                            const int px = vx[iv] + 1 - kx;
                            const int pxc = px < 0 ? 0 : (px >= a.W ? a.W - 1 : px);
                            const bool ok = uo[iu] && vo[iv] && py == pyc && px == pxc;
                            float v[4];
                            ld4(a.dy + (nb + (size_t)pyc * a.W + pxc) * a.lddy + c0, v);
#pragma unroll
                            for (int j = 0; j < 4; j++) acc[j] += (ok ? w[ky * 3 + kx][j] : 0.f) * v[j];
                        }
                    }
                }
        } else {
            // dx[q] = sum over padded coordinates u that mirror onto q, taps t: w[t] * dy[u + 1 - t].  Candidates in a fixed order
            // (the pixel itself, the mirror of row / column 1 at -1, the mirror of H-2 / W-2 at H / W), fully unrolled with
            // predicates: no data-dependent trip counts (the sums run in the order they always did)
            const int us[3] = {y, -1, a.H}, vs[3] = {x, -1, a.W};
            const bool uok[3] = {true, y == 1, y == a.H - 2}, vok[3] = {true, x == 1, x == a.W - 2};
#pragma unroll
            for (int iu = 0; iu < 3; iu++)
#pragma unroll
                for (int iv = 0; iv < 3; iv++) {
                    if (!(uok[iu] && vok[iv])) continue;
#pragma unroll
                    for (int ky = 0; ky < 3; ky++) {
                        const int py = us[iu] + 1 - ky;
#pragma unroll
                        for (int kx = 0; kx < 3; kx++) {
                            const int px = vs[iv] + 1 - kx;
                            if (py < 0 || py >= a.H || px < 0 || px >= a.W) continue;
                            float v[4];
                            ld4(a.dy + (nb + (size_t)py * a.W + px) * a.lddy + c0, v);
#pragma unroll
                            for (int j = 0; j < 4; j++) acc[j] += w[ky * 3 + kx][j] * v[j];
                        }
                    }
                }
        }
        st4(a.y + pix * a.ldy + c0, acc);
    }
}

// backward-weight: dw[c][t] = sum_pix dy[pix][c] * x[refl(pix + t)][c] ; db[c] = sum dy
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const DwArgs a) {
    __shared__ float red[256][41];
    const int ch = threadIdx.x & (a.CHP - 1);
    const int pl = threadIdx.x >> a.sh;
    const int c0 = ch * 4;
    const bool active = ch < a.CH4;
    float s[10][4];
#pragma unroll
    for (int t = 0; t < 10; t++)
#pragma unroll
        for (int j = 0; j < 4; j++) s[t][j] = 0.f;
    const size_t pixels = (size_t)a.N * a.H * a.W;
    if (active) {
        for (size_t pix = (size_t)blockIdx.x * a.PPB + pl; pix < pixels; pix += (size_t)gridDim.x * a.PPB) {
            const int x = (int)(pix % a.W);
            const size_t r = pix / a.W;
            const int y = (int)(r % a.H);
            const size_t nb = (r / a.H) * a.H * a.W;
            float g[4];
            ld4(a.dy + pix * a.lddy + c0, g);
#pragma unroll
            for (int j = 0; j < 4; j++) s[9][j] += g[j];
#pragma unroll
            for (int ky = 0; ky < 3; ky++) {
                const int sy = reflect(y + ky - 1, a.H);
#pragma unroll
                for (int kx = 0; kx < 3; kx++) {
                    const int sx = reflect(x + kx - 1, a.W);
                    float v[4];
                    ld4(a.x + (nb + (size_t)sy * a.W + sx) * a.ldx + c0, v);
#pragma unroll
                    for (int j = 0; j < 4; j++) s[ky * 3 + kx][j] += g[j] * v[j];
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 10; t++)
#pragma unroll
        for (int j = 0; j < 4; j++) red[threadIdx.x][t * 4 + j] = s[t][j];
    __syncthreads();
    if (pl == 0 && active) {
        const int C4 = a.CH4 * 4;
        for (int k = 0; k < 40; k++) {
            float t = 0.f;
            for (int q = 0; q < a.PPB; q++) t += red[q * a.CHP + ch][k];
            a.partial[((size_t)blockIdx.x * 10 + k / 4) * C4 + c0 + (k & 3)] = t;
        }
    }
}
// grid (ceil(C / 32), 10 taps [9 = bias]); 256 threads = 32 channels x 8 slices of the partial rows; fixed order, so deterministic
__global__ __launch_bounds__(256) void dwconv_wgrad_finalize_kernel(const float* partial, int blocks, int C, int C4, float* dw,
                                                                    float* db) {
    __shared__ double red[8][33];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl, t = blockIdx.y;
    double acc = 0.0;
    if (c < C) {
        const float* p = partial + (size_t)t * C4 + c;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int b = sl;
        for (; b + 24 < blocks; b += 32) {
            a0 += p[(size_t)b * 10 * C4];
            a1 += p[(size_t)(b + 8) * 10 * C4];
            a2 += p[(size_t)(b + 16) * 10 * C4];
            a3 += p[(size_t)(b + 24) * 10 * C4];
        }
        for (; b < blocks; b += 8) a0 += p[(size_t)b * 10 * C4];
        acc = ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
    }
    red[sl][cl] = acc;
    __syncthreads();
    if (sl == 0 && c < C) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < 8; q++) v += red[q][cl];
        if (t < 9) dw[(size_t)c * 9 + t] += (float)v;
        else if (db) db[c] += (float)v;
    }
}

int dw_setup(DwArgs* a, int C) {
    a->C = C;
    a->CH4 = ((C + 7) / 8) * 2;
    if (a->CH4 > 256) return GCC_ERR_UNSUPPORTED;
    a->CHP = 1; a->sh = 0;
    while (a->CHP < a->CH4) { a->CHP <<= 1; a->sh++; }
    a->PPB = 256 / a->CHP;
    return GCC_OK;
}
int dw_blocks(size_t pixels, int PPB, int cap) {
    size_t b = (pixels + (size_t)PPB * 4 - 1) / ((size_t)PPB * 4);
    if (b < 1) b = 1;
    if (b > (size_t)cap) b = cap;
    return (int)b;
}

}  // namespace

extern "C" int gcc_reflect_pad(const void* src, int lds, void* dst, int ldd, int N, int H, int W, int C, int pad,
                               int backward, gcc_stream_t stream) {
    GCC_ENTER();
    if (!src || !dst || N <= 0 || H <= pad || W <= pad || C <= 0 || pad <= 0 || (lds & 7) || (ldd & 7)) return GCC_ERR_BAD_ARG;
    PadArgs a;
    a.src = (const bf16_t*)src; a.dst = (bf16_t*)dst; a.N = N; a.H = H; a.W = W; a.pad = pad; a.lds = lds; a.ldd = ldd;
    a.CH = (C + 7) / 8;
    const size_t total = backward ? (size_t)N * H * W * a.CH : (size_t)N * (H + 2 * pad) * (W + 2 * pad) * a.CH;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (backward) hipLaunchKernelGGL(reflect_pad_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(reflect_pad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

extern "C" int gcc_dwconv3x3_reflect(int mode, const void* x, int ldx, const void* dy, int lddy, void* out, int ldo,
                                     const float* w, const float* bias, int N, int H, int W, int C, gcc_stream_t stream) {
    GCC_ENTER();
    if (!out || !w || N <= 0 || H < 2 || W < 2 || C <= 0 || (ldo & 7)) return GCC_ERR_BAD_ARG;
    if ((mode == 0 && (!x || (ldx & 7))) || (mode == 1 && (!dy || (lddy & 7))) || mode < 0 || mode > 1) return GCC_ERR_BAD_ARG;
    DwArgs a = {};
    a.x = (const bf16_t*)x; a.ldx = ldx; a.dy = (const bf16_t*)dy; a.lddy = lddy; a.y = (bf16_t*)out; a.ldy = ldo;
    a.w = w; a.bias = bias; a.N = N; a.H = H; a.W = W;
    int rc = dw_setup(&a, C);
    if (rc) return rc;
    const int blocks = dw_blocks((size_t)N * H * W, a.PPB, 4096);
    if (mode == 0) hipLaunchKernelGGL((dwconv_kernel<0>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((dwconv_kernel<1>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

extern "C" size_t gcc_dwconv3x3_wgrad_workspace(int N, int H, int W, int C) {
    DwArgs a = {};
    if (dw_setup(&a, C)) return 0;
    return (size_t)dw_blocks((size_t)N * H * W, a.PPB, 512) * 10 * a.CH4 * 4 * sizeof(float);
}

extern "C" int gcc_dwconv3x3_reflect_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw, float* dbias, int N,
                                           int H, int W, int C, void* ws, size_t ws_bytes, gcc_stream_t stream) {
    GCC_ENTER();
    if (!x || !dy || !dw || !ws || N <= 0 || H < 2 || W < 2 || C <= 0 || (ldx & 7) || (lddy & 7)) return GCC_ERR_BAD_ARG;
    if (ws_bytes < gcc_dwconv3x3_wgrad_workspace(N, H, W, C)) return GCC_ERR_WORKSPACE;
    DwArgs a = {};
    a.x = (const bf16_t*)x; a.ldx = ldx; a.dy = (const bf16_t*)dy; a.lddy = lddy; a.partial = (float*)ws;
    a.N = N; a.H = H; a.W = W;
    int rc = dw_setup(&a, C);
    if (rc) return rc;
    const int blocks = dw_blocks((size_t)N * H * W, a.PPB, 512);
    hipLaunchKernelGGL(dwconv_wgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    GCC_CHECK_LAUNCH();
    hipLaunchKernelGGL(dwconv_wgrad_finalize_kernel, dim3((C + 31) / 32, 10), dim3(256), 0, (hipStream_t)stream,
                       (const float*)ws, blocks, C, a.CH4 * 4, dw, dbias);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}
