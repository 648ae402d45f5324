// Evaluation arithmetic of the reference on device tensors; the networks that produce the
// activations / score maps (Inception, DRN) stay external inputs.
//   mIoU   metric/mIoU_score.py:108-109, 163-167, 213   argmax over classes, fast_hist (bincount of n * label + pred)
//   PSNR   models/SRGAN.py:653-657, data/sr_dataset.py:36-37, 58-62   luminance of [-1, 1] images, 4-pixel border cropped
//   FID    metric/fid_score.py:219-284, 327-328   mu = mean, sigma = np.cov(act, rowvar=False),
//          d^2 = |mu1 - mu2|^2 + tr(s1) + tr(s2) - 2 tr(sqrtm(s1 s2))
// Integer results are exact; the f64 results are reductions in a fixed order (no atomics on floating point).
// sqrtm: scipy's Schur method is replaced by the coupled Newton-Schulz iteration on B = s1 s2 + delta I (real
// non-negative spectrum, shifted off zero):
//     Y0 = B / |B|_F, Z0 = I ;  T = (3 I - Z Y) / 2 ;  Y <- Y T ;  Z <- T Z ;  Y -> sqrtm(B / |B|_F)
// three f64 GEMMs per iteration, all on the GPU; only tr(Y) is needed.  Covariances of fewer samples than dimensions
// (500 validation images against 2048 Inception features) make s1 s2 singular, where Z -> B^(-1/2) would diverge: the
// shift delta = shift_rel |A|_F bounds it, and a second solve at 4 delta cancels the k sqrt(delta) the k zero
// eigenvalues add to the trace (Richardson extrapolation in sqrt(delta)); measured 1e-8 relative against scipy on
// full-rank and rank-deficient inputs alike.  The GEMM is an LDS-tiled FMA kernel
// (64 x 64 x 16, 4 x 4 per thread): bound by the f64 vector rate (78 TFLOP/s peak), a once-per-epoch computation.
#include "common.hpp"

namespace {

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ s, int C, size_t HW, size_t total, int* __restrict__ pred) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t n = i / HW, p = i - n * HW;
        const float* b = s + n * C * HW + p;
        float best = b[0];
        int bi = 0;
        for (int c = 1; c < C; c++) {
            const float v = b[(size_t)c * HW];
            // numpy.argmax: first maximum; a NaN counts as the maximum (first NaN wins)
            if (best == best && (v > best || v != v)) { best = v; bi = c; }
        }
        pred[i] = bi;
    }
}

__global__ __launch_bounds__(256) void hist_kernel(const int* __restrict__ pred, const int* __restrict__ label, size_t count, int n,
                                                   unsigned long long* hist) {
    extern __shared__ unsigned int sh[];
    const int nn = n * n;
    for (int i = threadIdx.x; i < nn; i += 256) sh[i] = 0u;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        const int l = label[i];
        if (l >= 0 && l < n) {
            const int p = pred[i];
            if (p >= 0 && p < n) atomicAdd(&sh[n * l + p], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nn; i += 256)
        if (sh[i]) atomicAdd(&hist[i], (unsigned long long)sh[i]);
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_f64(double v, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) s += sh[i];
    return s;
}

__device__ __forceinline__ float luma(const float* img, size_t plane, size_t o) {
    // convert_image('[-1, 1]' -> 'y-channel'), the reference's operation order in fp32
    const float r = 255.f * ((img[o] + 1.f) / 2.f), g = 255.f * ((img[plane + o] + 1.f) / 2.f),
                b = 255.f * ((img[2 * plane + o] + 1.f) / 2.f);
    return (r * 65.481f + g * 128.553f + b * 24.966f) / 255.f + 16.f;
}

__global__ __launch_bounds__(256) void psnr_sse_kernel(const float* __restrict__ fake, const float* __restrict__ real, int N, int H,
                                                       int W, double* partial) {
    __shared__ double sh[4];
    const int h = H - 8, w = W - 8;
    const size_t per = (size_t)h * w, total = per * N, plane = (size_t)H * W;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t n = i / per, r = i - n * per;
        const size_t y = r / w + 4, x = r % w + 4;
        const size_t o = y * W + x;
        const double d = (double)luma(fake + n * 3 * plane, plane, o) - (double)luma(real + n * 3 * plane, plane, o);
        acc += d * d;
    }
    const double s = block_sum_f64(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// SSIM on the luminance channel (skimage.metrics.structural_similarity defaults: 7 x 7 uniform window, K1 = .01, K2 = .03,
// sample covariance NP / (NP - 1), mean of S over the window centres that keep the whole window inside the image), on the
// 4-pixel-cropped y-channel image of models/SRGAN.py:653-661.  One thread per window centre, 49 taps in f64.
__global__ __launch_bounds__(256) void ssim_y_kernel(const float* __restrict__ fake, const float* __restrict__ real, int N, int H,
                                                     int W, double data_range, double* partial) {
    __shared__ double sh[4];
    const int h = H - 8, w = W - 8, hv = h - 6, wv = w - 6;       // y image, valid window centres
    const size_t per = (size_t)hv * wv, total = per * N, plane = (size_t)H * W;
    const double C1 = (0.01 * data_range) * (0.01 * data_range), C2 = (0.03 * data_range) * (0.03 * data_range);
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t n = i / per, r = i - n * per;
        const size_t y0 = r / wv + 4, x0 = r % wv + 4;                // top-left tap in the uncropped image
        const float* f = fake + n * 3 * plane;
        const float* g = real + n * 3 * plane;
        double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
        for (int dy = 0; dy < 7; dy++)
            for (int dx = 0; dx < 7; dx++) {
                const size_t o = (y0 + dy) * W + x0 + dx;
                const double a = (double)luma(g, plane, o), b = (double)luma(f, plane, o);   // X = real, Y = fake (:660)
                sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
            }
        const double ux = sx / 49.0, uy = sy / 49.0, cn = 49.0 / 48.0;
        const double vx = cn * (sxx / 49.0 - ux * ux), vy = cn * (syy / 49.0 - uy * uy), vxy = cn * (sxy / 49.0 - ux * uy);
        acc += ((2.0 * ux * uy + C1) * (2.0 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
    }
    const double s = block_sum_f64(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const double* partial, int n, double scale, double* out, int accumulate) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    const double s = block_sum_f64(acc, sh) * scale;
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + s : s;
}

// ---------------------------------------------------------------------------------------------
// column means of act [n][d] (fp32 or f64) in f64: one thread per column chunk, rows split over blockIdx.y, fixed order
__global__ __launch_bounds__(256) void colsum_kernel(const void* act, int is_f64, int n, int d, int rows_per, double* partial) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= d) return;
    const int r0 = blockIdx.y * rows_per, r1 = min(n, r0 + rows_per);
    double acc = 0.0;
    if (is_f64) { const double* a = (const double*)act; for (int i = r0; i < r1; i++) acc += a[(size_t)i * d + j]; }
    else { const float* a = (const float*)act; for (int i = r0; i < r1; i++) acc += (double)a[(size_t)i * d + j]; }
    partial[(size_t)blockIdx.y * d + j] = acc;
}
__global__ __launch_bounds__(256) void mean_finalize_kernel(const double* partial, int slices, int n, int d, double* mu) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= d) return;
    double acc = 0.0;
    for (int s = 0; s < slices; s++) acc += partial[(size_t)s * d + j];
    mu[j] = acc / (double)n;
}
__global__ __launch_bounds__(256) void center_kernel(const void* act, int is_f64, size_t total, int d, const double* mu, double* xc) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const double v = is_f64 ? ((const double*)act)[i] : (double)((const float*)act)[i];
        xc[i] = v - mu[i % d];
    }
}

// ---------------------------------------------------------------------------------------------
// C[M][N] = alpha * (*alpha_dev) * sum_k a(i,k) b(k,j) + diag * [i == j];  a(i,k) = A[i*sai + k*sak], b(k,j) = B[k*sbk + j*sbj]
struct GemmArgs {
    const double* A; long sai, sak;
    const double* B; long sbk, sbj;
    double* C; int M, N, K;
    double alpha, diag;
    const double* alpha_dev;     // optional device scalar; inv != 0: multiply by 1 / (*alpha_dev)
    int inv;
};
__global__ __launch_bounds__(256) void dgemm_kernel(const GemmArgs g) {
    __shared__ double As[16][65], Bs[16][65];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    double acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) acc[r][c] = 0.0;
    for (int k0 = 0; k0 < g.K; k0 += 16) {
#pragma unroll
        for (int e0 = 0; e0 < 1024; e0 += 256) {
            const int e = e0 + tid;
            int ii, kk;
            if (g.sak == 1) { kk = e & 15; ii = e >> 4; } else { ii = e & 63; kk = e >> 6; }       // contiguous index fastest
            As[kk][ii] = (i0 + ii < g.M && k0 + kk < g.K) ? g.A[(long)(i0 + ii) * g.sai + (long)(k0 + kk) * g.sak] : 0.0;
            int jj, kb;
            if (g.sbk == 1) { kb = e & 15; jj = e >> 4; } else { jj = e & 63; kb = e >> 6; }
            Bs[kb][jj] = (j0 + jj < g.N && k0 + kb < g.K) ? g.B[(long)(k0 + kb) * g.sbk + (long)(j0 + jj) * g.sbj] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            double a[4], b[4];
#pragma unroll
            for (int r = 0; r < 4; r++) { a[r] = As[kk][ty * 4 + r]; b[r] = Bs[kk][tx * 4 + r]; }
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int c = 0; c < 4; c++) acc[r][c] += a[r] * b[c];
        }
        __syncthreads();
    }
    double al = g.alpha;
    if (g.alpha_dev) al = g.inv ? al / g.alpha_dev[0] : al * g.alpha_dev[0];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int i = i0 + ty * 4 + r, j = j0 + tx * 4 + c;
            if (i < g.M && j < g.N) g.C[(size_t)i * g.N + j] = al * acc[r][c] + (i == j ? g.diag : 0.0);
        }
}
void dgemm(hipStream_t st, const double* A, long sai, long sak, const double* B, long sbk, long sbj, double* C, int M, int N, int K,
           double alpha, double diag, const double* alpha_dev = nullptr, int inv = 0) {
    GemmArgs g = {A, sai, sak, B, sbk, sbj, C, M, N, K, alpha, diag, alpha_dev, inv};
    hipLaunchKernelGGL(dgemm_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, st, g);
}

// single-workgroup reductions (fixed order): out[slot] = sqrt(sum x^2) | trace | |a - b|^2
__global__ __launch_bounds__(1024) void reduce_kernel(int mode, const double* x, const double* y, size_t n, int d, double* out) {
    __shared__ double sh[16];
    double acc = 0.0;
    if (mode == 0) for (size_t i = threadIdx.x; i < n; i += 1024) acc += x[i] * x[i];                    // |x|_F^2
    else if (mode == 1) for (int i = threadIdx.x; i < d; i += 1024) acc += x[(size_t)i * d + i];         // trace
    else for (size_t i = threadIdx.x; i < n; i += 1024) { const double t = x[i] - y[i]; acc += t * t; }  // |x - y|^2
    const double s = block_sum_f64(acc, sh);
    if (threadIdx.x == 0) out[0] = mode == 0 ? sqrt(s) : s;
}
// B = A + (mult * |A|_F) I
__global__ void shift_kernel(const double* A, const double* norm, double mult, size_t n2, int d, double* B) {
    const double delta = mult * norm[0];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256)
        B[i] = A[i] + ((i / d == i % d) ? delta : 0.0);
}
__global__ void scale_identity_kernel(const double* B, const double* norm, size_t n2, int d, double* Y, double* Z) {
    const double inv = 1.0 / norm[0];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        Y[i] = B[i] * inv;
        Z[i] = (i / d == i % d) ? 1.0 : 0.0;
    }
}
// scal: 0 |A|_F ; 3 |mu1 - mu2|^2, 4 tr(s1), 5 tr(s2) ; run r (0: shift delta, 1: shift 4 delta): 6+3r |B|_F, 7+3r tr(Y) now,
// 8+3r tr(Y) one step earlier
__global__ void frechet_finish_kernel(const double* scal, int shifted, double* out) {
    const double f1 = scal[7] * sqrt(scal[6]);
    double tr = f1, resid = fabs(scal[7] - scal[8]) / fmax(fabs(scal[7]), 1e-300);
    if (shifted) {
        // tr sqrt(A + delta I) = T + k sqrt(delta) + O(delta) with k the number of zero eigenvalues (rank-deficient
        // covariances: fewer samples than dimensions): two shifts, delta and 4 delta, cancel the sqrt(delta) term
        const double f4 = scal[10] * sqrt(scal[9]);
        tr = 2.0 * f1 - f4;
        resid = fmax(resid, fabs(scal[10] - scal[11]) / fmax(fabs(scal[10]), 1e-300));
    }
    out[0] = scal[3] + scal[4] + scal[5] - 2.0 * tr;
    out[1] = resid;          // relative change of tr(Y) over the last Newton-Schulz step
}

int nblk(size_t n, int cap = 2048) {
    size_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > (size_t)cap ? cap : b));
}

}  // namespace

extern "C" int gcc_argmax_channels(const float* scores, int N, int C, size_t HW, int* pred, gcc_stream_t stream) {
    GCC_ENTER();
    if (!scores || !pred || N <= 0 || C <= 0 || HW == 0) return GCC_ERR_BAD_ARG;
    const size_t total = (size_t)N * HW;
    hipLaunchKernelGGL(argmax_kernel, dim3(nblk(total, 8192)), dim3(256), 0, (hipStream_t)stream, scores, C, HW, total, pred);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

extern "C" int gcc_confusion_hist(const int* pred, const int* label, size_t count, int n, long long* hist, gcc_stream_t stream) {
    GCC_ENTER();
    if (!pred || !label || !hist || n <= 0) return GCC_ERR_BAD_ARG;
    if ((size_t)n * n * sizeof(unsigned int) > 48 * 1024) return GCC_ERR_UNSUPPORTED;
    if (count == 0) return GCC_OK;
    hipLaunchKernelGGL(hist_kernel, dim3(nblk(count, 1024)), dim3(256), (size_t)n * n * sizeof(unsigned int), (hipStream_t)stream, pred,
                       label, count, n, (unsigned long long*)hist);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

extern "C" size_t gcc_psnr_workspace(void) { return 1024 * sizeof(double); }

extern "C" int gcc_psnr_y_sse(const float* fake, const float* real, int N, int H, int W, double* sse, int accumulate, void* ws,
                              size_t ws_bytes, gcc_stream_t stream) {
    GCC_ENTER();
    if (!fake || !real || !sse || !ws || N <= 0 || H <= 8 || W <= 8) return GCC_ERR_BAD_ARG;
    if (ws_bytes < gcc_psnr_workspace()) return GCC_ERR_WORKSPACE;
    const int nb = nblk((size_t)N * (H - 8) * (W - 8), 1024);
    hipLaunchKernelGGL(psnr_sse_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, fake, real, N, H, W, (double*)ws);
    GCC_CHECK_LAUNCH();
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)ws, nb, 1.0, sse, accumulate);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

extern "C" int gcc_ssim_y_sum(const float* fake, const float* real, int N, int H, int W, double* ssim_sum, int accumulate, void* ws,
                             size_t ws_bytes, gcc_stream_t stream) {
    GCC_ENTER();
    if (!fake || !real || !ssim_sum || !ws || N <= 0 || H < 8 + 7 || W < 8 + 7) return GCC_ERR_BAD_ARG;
    if (ws_bytes < gcc_psnr_workspace()) return GCC_ERR_WORKSPACE;
    const int nb = nblk((size_t)N * (H - 14) * (W - 14), 1024);
    hipLaunchKernelGGL(ssim_y_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, fake, real, N, H, W, 255.0, (double*)ws);
    GCC_CHECK_LAUNCH();
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)ws, nb, 1.0, ssim_sum, accumulate);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

extern "C" size_t gcc_activation_stats_workspace(int n, int d) {
    if (n <= 0 || d <= 0) return 0;
    return ((size_t)n * d + (size_t)64 * d) * sizeof(double);      // centred copy | column-sum slices
}

extern "C" int gcc_activation_stats(const void* act, int is_f64, int n, int d, double* mu, double* sigma, void* ws, size_t ws_bytes,
                                    gcc_stream_t stream) {
    GCC_ENTER();
    if (!act || !mu || !sigma || !ws || n < 2 || d <= 0) return GCC_ERR_BAD_ARG;
    if (ws_bytes < gcc_activation_stats_workspace(n, d)) return GCC_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    double* xc = (double*)ws;
    double* part = xc + (size_t)n * d;
    const int slices = n < 64 ? 1 : 64, rows_per = (n + slices - 1) / slices;
    hipLaunchKernelGGL(colsum_kernel, dim3((d + 255) / 256, slices), dim3(256), 0, st, act, is_f64, n, d, rows_per, part);
    GCC_CHECK_LAUNCH();
    hipLaunchKernelGGL(mean_finalize_kernel, dim3((d + 255) / 256), dim3(256), 0, st, (const double*)part, slices, n, d, mu);
    GCC_CHECK_LAUNCH();
    hipLaunchKernelGGL(center_kernel, dim3(nblk((size_t)n * d)), dim3(256), 0, st, act, is_f64, (size_t)n * d, d, (const double*)mu, xc);
    GCC_CHECK_LAUNCH();
    // sigma = Xc^T Xc / (n - 1)
    dgemm(st, xc, 1, d, xc, d, 1, sigma, d, d, n, 1.0 / (double)(n - 1), 0.0);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

extern "C" size_t gcc_frechet_workspace(int d) {
    if (d <= 0) return 0;
    return ((size_t)6 * d * d + 16) * sizeof(double);
}

extern "C" int gcc_frechet_distance(const double* mu1, const double* sigma1, const double* mu2, const double* sigma2, int d,
                                    int iterations, double shift_rel, double* out, void* ws, size_t ws_bytes,
                                    gcc_stream_t stream) {
    GCC_ENTER();
    if (!mu1 || !sigma1 || !mu2 || !sigma2 || !out || !ws || d <= 0 || iterations < 2 || shift_rel < 0.0) return GCC_ERR_BAD_ARG;
    if (ws_bytes < gcc_frechet_workspace(d)) return GCC_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const size_t n2 = (size_t)d * d;
    double* scal = (double*)ws;
    double* A = scal + 16;
    double *Y = A + n2, *Z = Y + n2, *T = Z + n2, *Y2 = T + n2, *Z2 = Y2 + n2;
    dgemm(st, sigma1, d, 1, sigma2, d, 1, A, d, d, d, 1.0, 0.0);
    hipLaunchKernelGGL(reduce_kernel, dim3(1), dim3(1024), 0, st, 0, (const double*)A, (const double*)nullptr, n2, d, scal + 0);
    hipLaunchKernelGGL(reduce_kernel, dim3(1), dim3(1024), 0, st, 2, mu1, mu2, (size_t)d, d, scal + 3);
    hipLaunchKernelGGL(reduce_kernel, dim3(1), dim3(1024), 0, st, 1, sigma1, (const double*)nullptr, n2, d, scal + 4);
    hipLaunchKernelGGL(reduce_kernel, dim3(1), dim3(1024), 0, st, 1, sigma2, (const double*)nullptr, n2, d, scal + 5);
    GCC_CHECK_LAUNCH();
    const int runs = shift_rel > 0.0 ? 2 : 1;
    for (int r = 0; r < runs; r++) {
        double* sc = scal + 6 + 3 * r;
        // T doubles as the home of B = A + delta I until the first product overwrites it
        hipLaunchKernelGGL(shift_kernel, dim3(nblk(n2)), dim3(256), 0, st, (const double*)A, (const double*)scal,
                           shift_rel * (r == 0 ? 1.0 : 4.0), n2, d, T);
        hipLaunchKernelGGL(reduce_kernel, dim3(1), dim3(1024), 0, st, 0, (const double*)T, (const double*)nullptr, n2, d, sc + 0);
        hipLaunchKernelGGL(scale_identity_kernel, dim3(nblk(n2)), dim3(256), 0, st, (const double*)T, (const double*)sc, n2, d, Y, Z);
        GCC_CHECK_LAUNCH();
        for (int it = 0; it < iterations; it++) {
            dgemm(st, Z, d, 1, Y, d, 1, T, d, d, d, -0.5, 1.5);        // T = (3 I - Z Y) / 2
            dgemm(st, Y, d, 1, T, d, 1, Y2, d, d, d, 1.0, 0.0);         // Y <- Y T
            dgemm(st, T, d, 1, Z, d, 1, Z2, d, d, d, 1.0, 0.0);         // Z <- T Z
            double* t = Y; Y = Y2; Y2 = t;
            t = Z; Z = Z2; Z2 = t;
            if (it >= iterations - 2)
                hipLaunchKernelGGL(reduce_kernel, dim3(1), dim3(1024), 0, st, 1, (const double*)Y, (const double*)nullptr, n2, d,
                                   sc + (it == iterations - 1 ? 1 : 2));
        }
        GCC_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(frechet_finish_kernel, dim3(1), dim3(1), 0, st, (const double*)scal, runs == 2 ? 1 : 0, out);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

// =============================================================================================
// Input pipeline (data/aligned_dataset.py:27-56, data/base_dataset.py:63-112): the paired image
// is split, each half resized with PIL's BICUBIC, cropped, flipped, scaled to [0, 1] and normalised to [-1, 1].
// Pillow (the reference's un-pinned image library; 12.2.0 in this image) resamples 8-bit images in two passes of
// integer arithmetic -- horizontal, then vertical, with 22-bit fixed-point coefficients and a rounded, clipped uint8
// intermediate (Resample.c: ImagingResampleHorizontal_8bpc / Vertical_8bpc).  The kernels below reproduce exactly that;
// the coefficient tables (double precision, a few hundred entries) are built by the caller with Pillow's arithmetic
// (gcc_amd/data/__init__.py::resample_coeffs) and passed in.
namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ unsigned char clip8(int v) {
    v >>= PRECISION_BITS;                       // arithmetic shift, as the C code's lookup index
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// dst[y][xx][c] over the columns of src: bounds[xx] = {xmin, count}, coef[xx][0..count)
__global__ __launch_bounds__(256) void resample_h_kernel(const unsigned char* __restrict__ src, size_t pitch, int H, int outW,
                                                         const int* __restrict__ bounds, const int* __restrict__ coef, int ksize,
                                                         unsigned char* __restrict__ dst) {
    const size_t total = (size_t)H * outW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / outW), xx = (int)(i - (size_t)y * outW);
        const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
        const int* k = coef + (size_t)xx * ksize;
        const unsigned char* row = src + (size_t)y * pitch + (size_t)xmin * 3;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int x = 0; x < cnt; x++) {
            const int kv = k[x];
            s0 += row[3 * x] * kv; s1 += row[3 * x + 1] * kv; s2 += row[3 * x + 2] * kv;
        }
        unsigned char* o = dst + i * 3;
        o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
    }
}
// dst[yy][x][c] over the rows of src (tightly packed [H][W][3])
__global__ __launch_bounds__(256) void resample_v_kernel(const unsigned char* __restrict__ src, size_t pitch, int W, int outH,
                                                         const int* __restrict__ bounds, const int* __restrict__ coef, int ksize,
                                                         unsigned char* __restrict__ dst) {
    const size_t total = (size_t)outH * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int yy = (int)(i / W), x = (int)(i - (size_t)yy * W);
        const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
        const int* k = coef + (size_t)yy * ksize;
        const unsigned char* p = src + (size_t)ymin * pitch + (size_t)x * 3;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int y = 0; y < cnt; y++) {
            const int kv = k[y];
            s0 += p[0] * kv; s1 += p[1] * kv; s2 += p[2] * kv;
            p += pitch;
        }
        unsigned char* o = dst + i * 3;
        o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
    }
}
// crop + horizontal flip + ToTensor (/255) + a range conversion, NCHW fp32 (and optionally the model's NHWC bf16 input):
//   form 0  Normalize(mean, std): (v / 255 - mean[c]) / std[c]      (base_dataset.py:108-111; sr_dataset.py:52-56)
//   form 1  convert_image '[-1, 1]': 2 * (v / 255) - 1              (sr_dataset.py:49-50)
//   form 2  '[0, 1]': v / 255
struct ConvArgs { float mean[3], stdv[3]; int form; };
__global__ __launch_bounds__(256) void crop_flip_norm_kernel(const unsigned char* __restrict__ src, size_t pitch, int x0, int y0,
                                                             int ch, int cw, int flip, const ConvArgs cv, float* __restrict__ nchw,
                                                             bf16_t* __restrict__ nhwc, int ld) {
    const size_t total = (size_t)ch * cw;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / cw), x = (int)(i - (size_t)y * cw);
        const int sx = flip ? (cw - 1 - x) : x;
        const unsigned char* p = src + (size_t)(y0 + y) * pitch + (size_t)(x0 + sx) * 3;
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float t = (float)p[c] / 255.f;
            v[c] = cv.form == 0 ? (t - cv.mean[c]) / cv.stdv[c] : (cv.form == 1 ? 2.f * t - 1.f : t);
        }
        if (nchw) {
#pragma unroll
            for (int c = 0; c < 3; c++) nchw[(size_t)c * total + i] = v[c];
        }
        if (nhwc) {
            bf16_t* o = nhwc + i * ld;
#pragma unroll
            for (int c = 0; c < 3; c++) o[c] = f2bf(v[c]);
        }
    }
}

}  // namespace

extern "C" int gcc_resample_u8(const void* src, int in_h, int in_w, size_t pitch, void* dst, int out_h, int out_w,
                               const int* hbounds, const int* hcoef, int hk, const int* vbounds, const int* vcoef, int vk,
                               void* tmp, gcc_stream_t stream) {
    GCC_ENTER();
    if (!src || !dst || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0 || pitch < (size_t)in_w * 3) return GCC_ERR_BAD_ARG;
    const bool need_h = out_w != in_w, need_v = out_h != in_h;
    if ((need_h && (!hbounds || !hcoef || hk <= 0)) || (need_v && (!vbounds || !vcoef || vk <= 0))) return GCC_ERR_BAD_ARG;
    if (need_h && need_v && !tmp) return GCC_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const unsigned char* s = (const unsigned char*)src;
    size_t sp = pitch;
    if (!need_h && !need_v) {        // PIL returns a copy
        if (hipMemcpy2DAsync(dst, (size_t)in_w * 3, src, pitch, (size_t)in_w * 3, in_h, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return GCC_ERR_LAUNCH;
        return GCC_OK;
    }
    if (need_h) {
        unsigned char* o = (unsigned char*)(need_v ? tmp : dst);
        hipLaunchKernelGGL(resample_h_kernel, dim3(nblk((size_t)in_h * out_w, 4096)), dim3(256), 0, st, s, sp, in_h, out_w, hbounds,
                           hcoef, hk, o);
        GCC_CHECK_LAUNCH();
        s = o; sp = (size_t)out_w * 3;
    }
    if (need_v) {
        hipLaunchKernelGGL(resample_v_kernel, dim3(nblk((size_t)out_h * out_w, 4096)), dim3(256), 0, st, s, sp, out_w, out_h, vbounds,
                           vcoef, vk, (unsigned char*)dst);
        GCC_CHECK_LAUNCH();
    }
    return GCC_OK;
}

extern "C" int gcc_crop_convert(const void* src, int H, int W, size_t pitch, int x0, int y0, int crop_h, int crop_w, int flip,
                                int form, const float* mean3, const float* std3, float* nchw, void* nhwc_bf16, int ld,
                                gcc_stream_t stream) {
    GCC_ENTER();
    if (!src || (!nchw && !nhwc_bf16) || crop_h <= 0 || crop_w <= 0 || x0 < 0 || y0 < 0 || x0 + crop_w > W || y0 + crop_h > H ||
        pitch < (size_t)W * 3 || (nhwc_bf16 && (ld < 3)) || form < 0 || form > 2 || (form == 0 && (!mean3 || !std3)))
        return GCC_ERR_BAD_ARG;
    ConvArgs cv = {};
    cv.form = form;
    for (int c = 0; c < 3; c++) { cv.mean[c] = form == 0 ? mean3[c] : 0.f; cv.stdv[c] = form == 0 ? std3[c] : 1.f; }
    hipLaunchKernelGGL(crop_flip_norm_kernel, dim3(nblk((size_t)crop_h * crop_w, 4096)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char*)src, pitch, x0, y0, crop_h, crop_w, flip, cv, nchw, (bf16_t*)nhwc_bf16, ld);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

extern "C" int gcc_crop_flip_normalize(const void* src, int H, int W, size_t pitch, int x0, int y0, int crop_h, int crop_w, int flip,
                                       float* nchw, void* nhwc_bf16, int ld, gcc_stream_t stream) {
    const float half[3] = {0.5f, 0.5f, 0.5f};
    return gcc_crop_convert(src, H, W, pitch, x0, y0, crop_h, crop_w, flip, 0, half, half, nchw, nhwc_bf16, ld, stream);
}
