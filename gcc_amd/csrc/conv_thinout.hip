// Thin-OUTPUT convolutions with a wide kernel: Conv2d(C -> Co, k, stride 1, pad (k - 1) / 2) with Co * k <= 32 -- SRGAN's last
// layer (models/SRGAN.py: conv_block3, 64 / 24 -> 3 channels, 9 x 9, at the HIGH resolution: 2.36 M pixels per 16-image batch at
// 96 -> 384).  As an implicit GEMM over (tap, channel) its 3 output channels are padded to a 16-wide MFMA operand (19 % of the
// matrix work is real) and its weight gradient is a 128 x 128-tile GEMM with M = 3: round 4 measured 1.0 ms forward and 5.4 ms
// for the weight gradient of the teacher's layer (14 TFLOP/s).
//
// The kernels here put the HORIZONTAL taps into the MFMA's free dimension instead: with m = (tx, co) -- k * Co <= 32 columns,
// 27 of 32 real for 9 x 3 -- a row of the input meets ALL its horizontal taps in one 16 x 16 x 32 product, and what is left of
// the kernel's width is a shift along x:
//   weight gradient   dW[co][ty][tx][ci] = sum_{y, x} E_y[x][(tx, co)] * X[y + ty - P][x][ci],   E_y[x][(tx, co)] = dY[y][x - tx + P][co]
//                     E_y is the row of dY expanded along x (a 32-column bf16 image, built in LDS); per row of dY and vertical
//                     tap one [32 x 64 pixels] x [64 pixels x C] product; K = pixels, both operands by transposing LDS reads.
// A workgroup owns a 64-pixel column strip of a band of rows of one image and walks down the rows: the rows of X pass through a
// ring in LDS (LDS-DMA, read once from HBM), the accumulators -- all k vertical taps x 32 x 16 channels per wave -- stay in
// registers for the whole walk, and the per-workgroup partial dW (k x 32 x C floats) is folded by a second small kernel.
// MFMA-bound: 2 k / 16 products per pixel row and wave; 302 MB of X in ~0.1 ms instead of 5 ms.
#include <stdlib.h>
#include <mutex>
#include "common.hpp"
#include "igemm_common.hpp"

namespace {
using gcc_igemm::OOB;

constexpr int TO_SW = 64;          // strip width in pixels
constexpr int TO_MAXK = 9;         // largest kernel side
constexpr int TO_D = 3;            // rows in flight ahead of the one being multiplied (LDS-DMA latency ~1-2 us, a row ~0.4 us)
constexpr int TO_RING = TO_MAXK + TO_D + 1;
constexpr int TO_DRING = TO_D + 1;

struct ThinOutWgradArgs {
    const bf16_t* x; const bf16_t* dy; float* part;
    int N, H, W, ldx, xoff, ldy, yoff, Ci, Co, K, P;
    int Cip;                // channels padded to 32 (the LDS row of a pixel: Cip * 2 bytes, 64 or 128)
    int strips, bands, band_h, units;
    uint32_t x_bytes, dy_bytes;
};

// physical byte offset of logical 16-byte chunk `ch` of pixel row `r` in an LDS image whose rows are RS bytes (64 or 128): the
// 32-byte windows are XOR-swizzled with the row so that the 32 lanes of a transposing read (rows 4g + q, g = 0 / 1, q = 0..3,
// four 8-byte pieces each) fall on 32 distinct bank pairs
template <int RS>
__device__ __forceinline__ int img_off(int r, int ch) {
    const int w = ch >> 1, sub = ch & 1;
    const int pw = RS == 128 ? (w ^ ((r >> 1) & 3)) : (w ^ ((r >> 2) & 1));
    return r * RS + pw * 32 + sub * 16;
}
// 16 x 16 x 32 operand from a [pixel][column] image: lane (g = lane >> 4, i = lane & 15) ends up with column colbase + i and the
// 8 pixel rows {ks*32 + 4g + 0..3, ks*32 + 16 + 4g + 0..3} (the same k order for both operands of a product)
template <int RS>
__device__ __forceinline__ bf16x8 img_tr_frag(const char* img, int ks, int colbase, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const int r0 = ks * 32 + 4 * g + (i >> 2), r1 = r0 + 16;
    const int c = colbase + 4 * (i & 3);                      // first of the 4 columns this lane addresses
    const int ch = c >> 3, inner = (c & 7) * 2;               // 16-byte chunk, byte offset inside it (0 or 8)
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, img + img_off<RS>(r0, ch) + inner));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, img + img_off<RS>(r1, ch) + inner));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// RS = Cip * 2: 128 (33..64 channels: four waves, 16 channels each) or 64 (<= 32 channels: waves 0 and 1 multiply)
// Persistent workgroups (one per CU): a unit = (image, 64-pixel strip, band of rows); a workgroup takes units blockIdx.x,
// blockIdx.x + gridDim.x, ... and keeps ONE set of accumulators over all of them.
template <int RS, int KK>
__global__ __launch_bounds__(256) void thinout_wgrad_kernel(const ThinOutWgradArgs a) {
    constexpr int XROW = TO_SW * RS;                          // bytes of one staged row of X
    constexpr int PIECES = XROW / 1024;                       // LDS-DMA wave-instructions per row (8 or 4)
    constexpr int PPW = PIECES / 4;                           // per wave (2 or 1)
    constexpr int NBLK = RS / 32;                             // 16-channel blocks (4 or 2)
    constexpr int E_BASE = TO_RING * XROW, D_BASE = E_BASE + 2 * TO_SW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sX = smem;                                          // [TO_RING][64 px][RS]
    char* sE = smem + E_BASE;                                 // [2][64 px][64 B]: the expanded dY row
    const bf16_t* sD = (const bf16_t*)(smem + D_BASE);        // [TO_DRING][128 px][8]: raw dY row segments (64 + 2 P pixels used)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int K = KK, P = (KK - 1) / 2;
    const i32x4 rs_x = make_rsrc(a.x, a.x_bytes), rs_dy = make_rsrc(a.dy, a.dy_bytes);
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);

    f32x4 acc[K][2];
#pragma unroll
    for (int t = 0; t < K; t++) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the 8 columns m = 8 c .. 8 c + 7 of the expanded row this thread writes (m = tx * Co + co): element offset of their source
    // inside the raw row relative to pixel px (-1: a padding column, zero)
    const int KC = K * a.Co;                                  // real columns of the expanded row (27)
    int eoff[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int m = (tid & 3) * 8 + e;
        const int tx = m / a.Co, co = m - tx * a.Co;
        eoff[e] = m < KC ? (2 * P - tx) * 8 + co : -1;
    }

    for (int unit = blockIdx.x; unit < a.units; unit += gridDim.x) {
        int b = unit;
        const int band = b % a.bands; b /= a.bands;
        const int strip = b % a.strips;
        const int n = b / a.strips;
        const int xs = strip * TO_SW;
        const int y0 = band * a.band_h, y1 = min(a.H, y0 + a.band_h);

        // ---- staging.  Every iteration issues the SAME number of LDS-DMA pieces per wave (rows outside the image fetch zeros
        // through out-of-range offsets), so that a counted s_waitcnt can leave the newest rows in flight.
        // row yy of X (this strip's 64 pixels x Cip channels) into ring slot yy mod TO_RING: one piece = 1 KiB = 1024 / RS pixel
        // rows; lane -> (row, physical chunk); the lane fetches the LOGICAL chunk that belongs in its physical slot
        auto stage_x = [&](int yy) {
            const bool row_ok = yy >= 0 && yy < a.H;
            const int slot = ((yy % TO_RING) + TO_RING) % TO_RING;
#pragma unroll
            for (int q = 0; q < PPW; q++) {
                const int piece = wave * PPW + q;
                constexpr int CPR = RS / 16;                      // chunks per pixel row
                const int r = piece * (1024 / RS) + lane / CPR;   // strip-local pixel
                const int pc = lane % CPR;                        // physical chunk
                const int pw = pc >> 1, sub = pc & 1;
                const int lw = RS == 128 ? (pw ^ ((r >> 1) & 3)) : (pw ^ ((r >> 2) & 1));
                const int ch = lw * 2 + sub;                      // logical chunk = channels 8 ch .. 8 ch + 7
                const int col = xs + r;
                const bool ok = row_ok && col < a.W && ch * 8 < a.Ci;
                const uint32_t off = ok ? (uint32_t)((((size_t)(n * a.H + yy) * a.W + col) * a.ldx + a.xoff + ch * 8) * 2) : OOB;
                lds_dma16(rs_x, lds0 + slot * XROW + piece * 1024, off);
            }
        };
        // the raw row yy of dY -- columns xs - P .. xs + 63 + P, 16 bytes (8 padded channels) per pixel -- into slot yy mod
        // TO_DRING of sD (wave 0, two pieces of 64 pixels)
        auto stage_dy = [&](int yy) {
            if (wave != 0) return;
            const int slot = ((yy % TO_DRING) + TO_DRING) % TO_DRING;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int j = q * 64 + lane, col = xs - P + j;
                const bool ok = j < TO_SW + 2 * P && yy >= 0 && yy < a.H && col >= 0 && col < a.W;
                const uint32_t off = ok ? (uint32_t)((((size_t)(n * a.H + yy) * a.W + col) * a.ldy + a.yoff) * 2) : OOB;
                lds_dma16(rs_dy, lds0 + D_BASE + slot * 2048 + q * 1024, off);
            }
        };

        // ---- prologue of the unit: rows y0 - P .. y0 + P + D - 1 of X, rows y0 .. y0 + D - 1 of dY ---------------------------
        __syncthreads();                                      // the previous unit's last products have read their operands
        for (int yy = y0; yy < y0 + TO_D; yy++) stage_dy(yy);
        for (int yy = y0 - P; yy < y0 + P + TO_D; yy++) stage_x(yy);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int y = y0; y < y1; y++) {
            const int eb = (y - y0) & 1;
            // everything issued up to iteration y - D has landed (X row y + P, dY row y); the newest D - 1 iterations may fly
            if (y > y0) {
                if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((TO_D - 1) * (PPW + 2)) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((TO_D - 1) * PPW) : "memory");
            }
            __syncthreads();
            stage_dy(y + TO_D);
            stage_x(y + P + TO_D);
            // expand row y: thread -> (pixel px = tid >> 2, 8 columns)
            {
                const int px = tid >> 2, c = tid & 3;
                const bf16_t* d = sD + (size_t)(((y % TO_DRING) + TO_DRING) % TO_DRING) * 1024 + px * 8;
                uint32_t pk[4];
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const bf16_t v0 = eoff[e] >= 0 ? d[eoff[e]] : (bf16_t)0, v1 = eoff[e + 1] >= 0 ? d[eoff[e + 1]] : (bf16_t)0;
                    pk[e >> 1] = (uint32_t)v0 | ((uint32_t)v1 << 16);
                }
                *(i32x4*)(sE + eb * (TO_SW * 64) + img_off<64>(px, c)) = i32x4{(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]};
            }
            __syncthreads();                                  // E(y) complete
            if (wave < NBLK) {
                // no branch on the row: rows outside the image were staged as zeros, so every vertical tap multiplies -- the K
                // operand reads of a k-step are issued together and their latency is paid once (with one wave per SIMD a
                // read -> wait -> two products chain per tap ran 4 x slower)
                const char* E = sE + eb * (TO_SW * 64);
                int s0 = (y - P) % TO_RING;
                s0 = s0 < 0 ? s0 + TO_RING : s0;
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    const bf16x8 e0 = img_tr_frag<64>(E, ks, 0, lane), e1 = img_tr_frag<64>(E, ks, 16, lane);
                    bf16x8 xb[K];
#pragma unroll
                    for (int t = 0; t < K; t++) {
                        const int slot = s0 + t >= TO_RING ? s0 + t - TO_RING : s0 + t;
                        xb[t] = img_tr_frag<RS>(sX + slot * XROW, ks, wave * 16, lane);
                    }
#pragma unroll
                    for (int t = 0; t < K; t++) {
                        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(e0, xb[t], acc[t][0], 0, 0, 0);
                        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(e1, xb[t], acc[t][1], 0, 0, 0);
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the rows fetched past the band's end: nothing may land in the next unit's slots
    }
    // ---- the workgroup's partial: part[wg][ty][m (32)][Cip] ------------------------------------------------------------------
    if (wave < NBLK) {
        const int g = lane >> 4, i = lane & 15;
        float* out = a.part + (size_t)blockIdx.x * K * 32 * a.Cip;
#pragma unroll
        for (int t = 0; t < K; t++)
#pragma unroll
            for (int mb = 0; mb < 2; mb++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    out[((size_t)t * 32 + mb * 16 + 4 * g + r) * a.Cip + wave * 16 + i] = acc[t][mb][r];
    }
}

// ---- forward --------------------------------------------------------------------------------------------------------------
//   U_y[x_in][(tx, co)] = sum_{ty, ci} X[y + ty - P][x_in][ci] * W[co][ty][tx][ci]        one [32 x K C] x [K C x 64 pixels] product per row
//   Y[y][x][co]         = act(bias[co] + sum_tx U_y[x + tx - P][(tx, co)])                   a shift-and-add over the k horizontal taps
// Same walk as the weight gradient (64 input columns per strip = 64 - 2 P output columns, the rows of X through the LDS ring);
// the weight operands -- 2 x K x C / 32 fragments -- stay in registers for the whole launch, the pixel operand is a plain
// ds_read_b128 of the staged row (K = channels is contiguous), U goes through an 8 KB LDS image for the shift.
struct ThinOutFpropArgs {
    const bf16_t* x; const bf16_t* w; bf16_t* y; const float* bias;
    int N, H, W, ldx, xoff, ldy, yoff, Ci, Co;
    int Cip8;               // channels per tap of the packed weights (Ci rounded up to 8)
    int strips, bands, band_h, units, act;
    float slope;
    uint32_t x_bytes, w_bytes;
};

template <int RS, int KK>
__global__ __launch_bounds__(256) void thinout_fprop_kernel(const ThinOutFpropArgs a) {
    constexpr int K = KK, P = (KK - 1) / 2, SWO = TO_SW - 2 * P;
    constexpr int XROW = TO_SW * RS, PIECES = XROW / 1024, PPW = PIECES / 4;
    constexpr int CC = RS / 64;                               // 32-channel k-steps per vertical tap (2 or 1)
    constexpr int U_BASE = TO_RING * XROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sX = smem;                                          // [TO_RING][64 px][RS]
    float* sU = (float*)(smem + U_BASE);                      // [64 px][32 m] fp32 (+ 4 floats of row padding: 144-byte rows)
    constexpr int UROW = 36;

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, i = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const i32x4 rs_x = make_rsrc(a.x, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
    const int KC = K * a.Co;

    // weight operands: fragment (mb, ty, cc): lane (i, g) holds row m = 16 mb + i = (tx, co), k = (ty, channels 32 cc + 8 g .. + 7)
    bf16x8 wf[2][K][CC];
#pragma unroll
    for (int mb = 0; mb < 2; mb++) {
        const int m = mb * 16 + i;
        const int tx = m / a.Co, co = m - tx * a.Co;
#pragma unroll
        for (int ty = 0; ty < K; ty++)
#pragma unroll
            for (int cc = 0; cc < CC; cc++) {
                const int ci = cc * 32 + g * 8;
                const bool ok = m < KC && ci < a.Cip8;
                const uint32_t off = ok ? (uint32_t)((((co * K + ty) * K + tx) * a.Cip8 + ci) * 2) : OOB;
                wf[mb][ty][cc] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0));
            }
    }
    // the shift-and-add: thread -> (pixel px = tid >> 2, taps tx = j, j + 4, j + 8 with j = tid & 3); partial sums meet by shuffles
    const int spx = tid >> 2, sj = tid & 3;
    float bv[3] = {0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
        for (int c = 0; c < 3; c++) bv[c] = c < a.Co ? a.bias[c] : 0.f;
    }

    for (int unit = blockIdx.x; unit < a.units; unit += gridDim.x) {
        int b = unit;
        const int band = b % a.bands; b /= a.bands;
        const int strip = b % a.strips;
        const int n = b / a.strips;
        const int xo = strip * SWO, xs = xo - P;              // first output column, first input column (may be negative: zeros)
        const int y0 = band * a.band_h, y1 = min(a.H, y0 + a.band_h);
        auto stage_x = [&](int yy) {
            const bool row_ok = yy >= 0 && yy < a.H;
            const int slot = ((yy % TO_RING) + TO_RING) % TO_RING;
#pragma unroll
            for (int q = 0; q < PPW; q++) {
                const int piece = wave * PPW + q;
                constexpr int CPR = RS / 16;
                const int r = piece * (1024 / RS) + lane / CPR;
                const int pc = lane % CPR;
                const int pw = pc >> 1, sub = pc & 1;
                const int lw = RS == 128 ? (pw ^ ((r >> 1) & 3)) : (pw ^ ((r >> 2) & 1));
                const int ch = lw * 2 + sub;
                const int col = xs + r;
                const bool ok = row_ok && col >= 0 && col < a.W && ch * 8 < a.Ci;
                const uint32_t off = ok ? (uint32_t)((((size_t)(n * a.H + yy) * a.W + col) * a.ldx + a.xoff + ch * 8) * 2) : OOB;
                lds_dma16(rs_x, lds0 + slot * XROW + piece * 1024, off);
            }
        };
        __syncthreads();
        for (int yy = y0 - P; yy < y0 + P + TO_D; yy++) stage_x(yy);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int y = y0; y < y1; y++) {
            if (y > y0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((TO_D - 1) * PPW) : "memory");
            __syncthreads();                                  // row y + P landed everywhere; the previous row's shift-and-add has read sU
            stage_x(y + P + TO_D);
            int s0 = (y - P) % TO_RING;
            s0 = s0 < 0 ? s0 + TO_RING : s0;
            f32x4 u0 = {0.f, 0.f, 0.f, 0.f}, u1 = {0.f, 0.f, 0.f, 0.f};
            // the K x CC pixel operands of the row as one software pipeline, six LDS reads ahead of the products they feed (the
            // compiler's own schedule keeps one ahead: conv_ring3.hip, profiles/r5_ring3.txt)
            // pixel operand k = ty * CC + cc: lane (i, g) -> pixel 16 wave + i, channels 32 cc + 8 g .. + 7 = 16-byte chunk 4 cc + g
            constexpr int NI = K * CC, PF = NI < 6 ? NI : 6;
            auto rd = [&](int k) {
                const int ty = k / CC, cc = k - ty * CC;
                const int slot = s0 + ty >= TO_RING ? s0 + ty - TO_RING : s0 + ty;
                return *(const bf16x8*)(sX + slot * XROW + img_off<RS>(wave * 16 + i, cc * 4 + g));
            };
            bf16x8 xbv[NI];
#pragma unroll
            for (int k = 0; k < PF; k++) xbv[k] = rd(k);
            __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
            for (int k = 0; k < NI; k++) {
                if (k + PF < NI) xbv[k + PF] = rd(k + PF);
                const int ty = k / CC, cc = k - ty * CC;
                u0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][ty][cc], xbv[k], u0, 0, 0, 0);
                u1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][ty][cc], xbv[k], u1, 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                if (k + PF < NI) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            // lane (i, g) holds U[pixel 16 wave + i][m = 4 g + r] (u0) and [16 + 4 g + r] (u1)
            *(f32x4*)(sU + (wave * 16 + i) * UROW + 4 * g) = u0;
            *(f32x4*)(sU + (wave * 16 + i) * UROW + 16 + 4 * g) = u1;
            __syncthreads();
            float o[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const int tx = sj + 4 * q;
                const int src = spx + tx;                     // U pixel of output pixel P + spx' ... (output local x = spx: input local spx + tx)
                if (tx < K && src < TO_SW) {
#pragma unroll
                    for (int c = 0; c < 3; c++)
                        if (c < a.Co) o[c] += sU[src * UROW + tx * a.Co + c];
                }
            }
#pragma unroll
            for (int c = 0; c < 3; c++) {
                o[c] += __shfl_xor(o[c], 1, 64);
                o[c] += __shfl_xor(o[c], 2, 64);
            }
            // output local x = spx (0 .. SWO - 1) sits at input local x = spx + P: y[x] = sum_tx U[x + tx]  (x + tx - P + P)
            if (sj == 0 && spx < SWO && xo + spx < a.W) {
                float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 3; c++)
                    if (c < a.Co) f[c] = apply_act(o[c] + bv[c], a.act, a.slope);
                *(i32x4*)(a.y + ((size_t)(n * a.H + y) * a.W + xo + spx) * a.ldy + a.yoff) = pack8(f);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// ---- data gradient ------------------------------------------------------------------------------------------------------------
//   dX[y][x][ci] = sum_ty sum_{(tx, co)} E_{y - ty + P}[x][(tx, co)] * W[co][ty][tx][ci]     E: the expanded rows of dY (see the weight gradient)
// One 32-deep k-step per vertical tap: [16 channels x 32] x [32 x 16 pixels]; the expanded rows of the last k rows of dY sit in an
// LDS ring (4 KB each), the weight operands (k fragments per wave = 16 channels) in registers; a wave owns 16 channels of the 64
// pixels of a strip.  dX is written once (the 302 MB of the teacher's layer), dY read ~1.1 times.
struct ThinOutDgradArgs {
    const bf16_t* dy; const bf16_t* wt; bf16_t* dx;
    int N, H, W, ldx, xoff, ldy, yoff, Ci, Co;
    int Cop8;               // output channels per tap of the packed dgrad weights (Co rounded up to 8)
    int strips, bands, band_h, units, nblk;
    uint32_t dy_bytes, wt_bytes;
};

template <int KK>
__global__ __launch_bounds__(256) void thinout_dgrad_kernel(const ThinOutDgradArgs a) {
    constexpr int K = KK, P = (KK - 1) / 2;
    constexpr int ERING = K + 1, EIMG = TO_SW * 64;
    constexpr int D_BASE = ERING * EIMG;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sE = smem;                                          // [ERING][64 px][64 B]
    const bf16_t* sD = (const bf16_t*)(smem + D_BASE);        // [TO_DRING][128 px][8] raw rows of dY

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, i = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const i32x4 rs_dy = make_rsrc(a.dy, a.dy_bytes);
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
    const int KC = K * a.Co;
    const bool wave_on = wave < a.nblk;                       // 16-channel blocks: 4 (<= 64 channels) or fewer

    // weight operands: A[ci = 16 wave + i][k = (tx, co) = 8 g .. 8 g + 7] for every vertical tap, from the dgrad packing
    // Wt[ci][ty * K + tx][Cop8]: eight 2-byte gathers per fragment, once per launch
    bf16x8 wf[K];
    {
        const int ci = wave * 16 + i;
#pragma unroll
        for (int ty = 0; ty < K; ty++) {
            s16x8 v;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int m = g * 8 + e;
                const int tx = m / a.Co, co = m - tx * a.Co;
                const bool ok = m < KC && ci < a.Ci;
                v[e] = ok ? (short)a.wt[((size_t)ci * K * K + ty * K + tx) * a.Cop8 + co] : (short)0;
            }
            wf[ty] = __builtin_bit_cast(bf16x8, v);
        }
    }
    int eoff[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int m = (tid & 3) * 8 + e;
        const int tx = m / a.Co, co = m - tx * a.Co;
        eoff[e] = m < KC ? (2 * P - tx) * 8 + co : -1;
    }

    for (int unit = blockIdx.x; unit < a.units; unit += gridDim.x) {
        int b = unit;
        const int band = b % a.bands; b /= a.bands;
        const int strip = b % a.strips;
        const int n = b / a.strips;
        const int xs = strip * TO_SW;
        const int y0 = band * a.band_h, y1 = min(a.H, y0 + a.band_h);
        auto stage_dy = [&](int yy) {                         // wave 0: the raw row (zeros outside the image) into slot yy mod TO_DRING
            if (wave != 0) return;
            const int slot = ((yy % TO_DRING) + TO_DRING) % TO_DRING;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int j = q * 64 + lane, col = xs - P + j;
                const bool ok = j < TO_SW + 2 * P && yy >= 0 && yy < a.H && col >= 0 && col < a.W;
                const uint32_t off = ok ? (uint32_t)((((size_t)(n * a.H + yy) * a.W + col) * a.ldy + a.yoff) * 2) : OOB;
                lds_dma16(rs_dy, lds0 + D_BASE + slot * 2048 + q * 1024, off);
            }
        };
        auto expand = [&](int yy) {                           // raw row yy -> its expanded image in slot yy mod ERING
            const int px = tid >> 2, c = tid & 3;
            const bf16_t* d = sD + (size_t)(((yy % TO_DRING) + TO_DRING) % TO_DRING) * 1024 + px * 8;
            uint32_t pk[4];
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const bf16_t v0 = eoff[e] >= 0 ? d[eoff[e]] : (bf16_t)0, v1 = eoff[e + 1] >= 0 ? d[eoff[e + 1]] : (bf16_t)0;
                pk[e >> 1] = (uint32_t)v0 | ((uint32_t)v1 << 16);
            }
            const int slot = ((yy % ERING) + ERING) % ERING;
            *(i32x4*)(sE + slot * EIMG + img_off<64>(px, c)) = i32x4{(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]};
        };
        // ---- prologue: raw rows y0 - P .. y0 + P + D - 1; the images of rows y0 - P .. y0 + P - 1 ------------------------------
        __syncthreads();
        for (int yy = y0 - P; yy < y0 + P + TO_D; yy++) {
            // the raw ring holds D + 1 rows: stage, wait, expand in turn for the first 2 P rows, then leave D rows in flight
            stage_dy(yy);
            if (yy < y0 + P) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                expand(yy);
                __syncthreads();
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int y = y0; y < y1; y++) {
            if (y > y0 && wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((TO_D - 1) * 2) : "memory");
            __syncthreads();                                  // raw row y + P landed; the previous row's products are done with the image slot it replaces
            stage_dy(y + P + TO_D);
            expand(y + P);
            __syncthreads();
            if (wave_on) {
                int s0 = (y + P) % ERING;                     // ty = 0 reads the image of row y + P, ty = k - 1 that of row y - P
                f32x4 acc[4];
#pragma unroll
                for (int nb = 0; nb < 4; nb++) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
                // the 4 K pixel operands as one software pipeline, six LDS reads ahead of their products (conv_ring3.hip)
                constexpr int NI = 4 * K, PF = 6;
                auto rd = [&](int k) {
                    const int ty = k >> 2, nb = k & 3;
                    const int slot = s0 - ty < 0 ? s0 - ty + ERING : s0 - ty;
                    return *(const bf16x8*)(sE + slot * EIMG + img_off<64>(nb * 16 + i, g));
                };
                bf16x8 ebv[NI];
#pragma unroll
                for (int k = 0; k < PF; k++) ebv[k] = rd(k);
                __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
                for (int k = 0; k < NI; k++) {
                    if (k + PF < NI) ebv[k + PF] = rd(k + PF);
                    acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[k >> 2], ebv[k], acc[k & 3], 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (k + PF < NI) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                // lane (i, g): pixel 16 nb + i, channels 16 wave + 4 g + r
                const int c0 = wave * 16 + 4 * g;
#pragma unroll
                for (int nb = 0; nb < 4; nb++) {
                    const int col = xs + nb * 16 + i;
                    if (col < a.W && c0 < a.Ci) {
                        const uint32_t lo = pack2bf(acc[nb][0], acc[nb][1]), hi = pack2bf(acc[nb][2], acc[nb][3]);
                        *(i32x2*)(a.dx + ((size_t)(n * a.H + y) * a.W + col) * a.ldx + a.xoff + c0) = i32x2{(int)lo, (int)hi};
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// dw[co][ty * K + tx][ci] (+)= sum over workgroups of part[wg][ty][tx * Co + co][ci]   (fixed order: reproducible)
__global__ __launch_bounds__(256) void thinout_wgrad_fold_kernel(const float* part, float* dw, int wgs, int K, int Co, int Ci, int Cip,
                                                                 int accumulate) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int total = K * K * Co * Ci;
    if (idx >= total) return;
    const int ci = idx % Ci;
    int r = idx / Ci;
    const int co = r % Co; r /= Co;
    const int tx = r % K, ty = r / K;
    const size_t stride = (size_t)K * 32 * Cip;
    const float* p = part + ((size_t)ty * 32 + tx * Co + co) * Cip + ci;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int w = 0;
    for (; w + 4 <= wgs; w += 4) {
        s0 += p[(size_t)w * stride]; s1 += p[(size_t)(w + 1) * stride]; s2 += p[(size_t)(w + 2) * stride]; s3 += p[(size_t)(w + 3) * stride];
    }
    for (; w < wgs; w++) s0 += p[(size_t)w * stride];
    const float s = (s0 + s1) + (s2 + s3);
    float* o = dw + ((size_t)co * K * K + ty * K + tx) * Ci + ci;
    *o = accumulate ? *o + s : s;
}

struct ThinOutPlan { int ok, Cip, strips, bands, band_h, units, wgs; size_t ws_bytes, lds; };
static int thinout_cus() {
    static const int v = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
            (void)hipGetLastError();
            n = 256;
        }
        return n;
    }();
    return v;
}
ThinOutPlan thinout_plan(const gcc_conv_t* c) {
    ThinOutPlan p = {};
    if (!gcc_opt(GCC_OPT_IGEMM_THIN)) return p;
    if (c->KH != c->KW || (c->KH & 1) == 0 || c->KH < 3 || c->KH > TO_MAXK || c->stride != 1 || c->pad != (c->KH - 1) / 2) return p;
    // Co <= 8: the weight- and data-gradient kernels stage dY as ONE 16-byte chunk per pixel (channels 0..7 of the padded row,
    // `sD[..][px][8]`); a 3 x 3 layer with 9 or 10 output channels (Co * KW <= 32) must take the generic kernels
    if (c->Co < 1 || c->Co > 8 || c->Co * c->KW > 32 || c->Ci < 8 || c->Ci > 64 || (c->Ci & 7)) return p;
    if (c->W < 16 || c->H < 1) return p;
    p.Cip = c->Ci > 32 ? 64 : 32;
    p.strips = cdiv(c->W, TO_SW);
    // One workgroup per CU walks units (image, strip, band of rows).  A band costs band_h + 2 P + D staged rows: choose the band
    // count that minimises rows per workgroup = ceil(units / CUs) * (band_h + 2 P + D), bands of at least 2 k rows.
    const int cus = thinout_cus(), cols = c->N * p.strips;
    const int max_bands = c->H / (2 * c->KH) > 0 ? c->H / (2 * c->KH) : 1;
    long best = -1;
    for (int b = 1; b <= max_bands && b <= 64; b++) {
        const int bh = cdiv(c->H, b), nb = cdiv(c->H, bh);
        const long cost = (long)cdiv(cols * nb, cus) * (bh + 2 * c->pad + TO_D);
        if (best < 0 || cost < best) { best = cost; p.band_h = bh; p.bands = nb; }
    }
    p.units = cols * p.bands;
    p.wgs = p.units < cus ? p.units : cus;
    p.ws_bytes = (size_t)p.wgs * c->KH * 32 * p.Cip * sizeof(float);
    p.lds = (size_t)TO_RING * TO_SW * p.Cip * 2 + 2 * TO_SW * 64 + TO_DRING * 2048;
    const size_t xb = (size_t)c->N * c->H * c->W * c->ldx * 2, yb = (size_t)c->N * c->H * c->W * c->ldy * 2;
    if (xb >= OOB || yb >= OOB) return p;
    p.ok = 1;
    return p;
}

}  // namespace

size_t gcc_internal_thinout_wgrad_workspace(const gcc_conv_t* c) {
    const ThinOutPlan p = thinout_plan(c);
    return p.ok ? p.ws_bytes : 0;
}

// returns GCC_ERR_UNSUPPORTED when the geometry is not this route's (the caller takes the generic kernel)
int gcc_internal_thinout_wgrad(const gcc_conv_t* c, const void* x, const void* dy, float* dw, int accumulate, void* ws, size_t ws_bytes,
                               hipStream_t st) {
    const ThinOutPlan p = thinout_plan(c);
    if (!p.ok || !ws || ws_bytes < p.ws_bytes || (((uintptr_t)ws) & 15)) return GCC_ERR_UNSUPPORTED;
    ThinOutWgradArgs a;
    a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.part = (float*)ws;
    a.N = c->N; a.H = c->H; a.W = c->W; a.ldx = c->ldx; a.xoff = c->xoff; a.ldy = c->ldy; a.yoff = c->yoff;
    a.Ci = c->Ci; a.Co = c->Co; a.K = c->KH; a.P = c->pad; a.Cip = p.Cip;
    a.strips = p.strips; a.bands = p.bands; a.band_h = p.band_h; a.units = p.units;
    a.x_bytes = (uint32_t)((size_t)c->N * c->H * c->W * c->ldx * 2);
    a.dy_bytes = (uint32_t)((size_t)c->N * c->H * c->W * c->ldy * 2);
    // one instantiation per (row bytes, kernel side): the vertical taps are unrolled
#define GCC_TO_LAUNCH(RS_, K_)                                                                                                   \
    do {                                                                                                                          \
        static std::once_flag once;                                                                                               \
        static hipError_t attr_err = hipSuccess;                                                                                  \
        std::call_once(once, [] {                                                                                                 \
            attr_err = hipFuncSetAttribute((const void*)thinout_wgrad_kernel<RS_, K_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        });                                                                                                                       \
        if (attr_err != hipSuccess) return GCC_ERR_LAUNCH;                                                                        \
        hipLaunchKernelGGL((thinout_wgrad_kernel<RS_, K_>), dim3(p.wgs), dim3(256), p.lds, st, a);                                \
    } while (0)
    const bool wide = p.Cip == 64;
    switch (c->KH) {
        case 9: if (wide) GCC_TO_LAUNCH(128, 9); else GCC_TO_LAUNCH(64, 9); break;
        case 7: if (wide) GCC_TO_LAUNCH(128, 7); else GCC_TO_LAUNCH(64, 7); break;
        case 5: if (wide) GCC_TO_LAUNCH(128, 5); else GCC_TO_LAUNCH(64, 5); break;
        default: if (wide) GCC_TO_LAUNCH(128, 3); else GCC_TO_LAUNCH(64, 3); break;
    }
#undef GCC_TO_LAUNCH
    GCC_CHECK_LAUNCH();
    const int total = c->KH * c->KW * c->Co * c->Ci;
    hipLaunchKernelGGL(thinout_wgrad_fold_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, (const float*)ws, dw, p.wgs, c->KH, c->Co, c->Ci,
                       p.Cip, accumulate);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

// does a fprop / dgrad call with this geometry and epilogue take the thin-output route?  (gcc_conv_route: 3)
bool gcc_internal_thinout_routed(const gcc_conv_t* c, int dgrad, const gcc_epilogue_t* ep) {
    const ThinOutPlan p = thinout_plan(c);
    if (!p.ok) return false;
    if (ep && (ep->stats_partial || ep->y2 || ep->bn)) return false;
    if (dgrad) return (c->Ci & 3) == 0 && !(ep && (ep->bias || ep->act != GCC_ACT_NONE));
    return c->Co <= 3;
}

// forward route of gcc_conv_fprop (conv_igemm.hip): GCC_ERR_UNSUPPORTED = not this route's geometry / epilogue
int gcc_internal_thinout_fprop(const gcc_conv_t* c, const void* x, const void* w, void* y, const gcc_epilogue_t* ep, hipStream_t st) {
    ThinOutPlan p = thinout_plan(c);
    if (!p.ok || c->Co > 3) return GCC_ERR_UNSUPPORTED;
    if (ep && (ep->stats_partial || ep->y2 || ep->bn)) return GCC_ERR_UNSUPPORTED;
    const int K = c->KH, SWO = TO_SW - 2 * c->pad;
    // strips tile the OUTPUT columns here (64 input columns give 64 - 2 P outputs); same band search as the weight gradient
    p.strips = cdiv(c->W, SWO);
    const int cus = thinout_cus(), cols = c->N * p.strips;
    const int max_bands = c->H / (2 * K) > 0 ? c->H / (2 * K) : 1;
    long best = -1;
    for (int b = 1; b <= max_bands && b <= 64; b++) {
        const int bh = cdiv(c->H, b), nb = cdiv(c->H, bh);
        const long cost = (long)cdiv(cols * nb, cus) * (bh + 2 * c->pad + TO_D);
        if (best < 0 || cost < best) { best = cost; p.band_h = bh; p.bands = nb; }
    }
    p.units = cols * p.bands;
    p.wgs = p.units < cus ? p.units : cus;
    ThinOutFpropArgs a;
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.y = (bf16_t*)y; a.bias = ep ? ep->bias : nullptr;
    a.N = c->N; a.H = c->H; a.W = c->W; a.ldx = c->ldx; a.xoff = c->xoff; a.ldy = c->ldy; a.yoff = c->yoff; a.Ci = c->Ci; a.Co = c->Co;
    a.Cip8 = ceil8(c->Ci);
    a.strips = p.strips; a.bands = p.bands; a.band_h = p.band_h; a.units = p.units;
    a.act = ep ? ep->act : GCC_ACT_NONE; a.slope = ep ? ep->slope : 0.f;
    a.x_bytes = (uint32_t)((size_t)c->N * c->H * c->W * c->ldx * 2);
    a.w_bytes = (uint32_t)((size_t)c->Co * K * K * a.Cip8 * 2);
    const size_t lds = (size_t)TO_RING * TO_SW * p.Cip * 2 + (size_t)TO_SW * 36 * 4;
#define GCC_TO_LAUNCH(RS_, K_)                                                                                                   \
    do {                                                                                                                          \
        static std::once_flag once;                                                                                               \
        static hipError_t attr_err = hipSuccess;                                                                                  \
        std::call_once(once, [] {                                                                                                 \
            attr_err = hipFuncSetAttribute((const void*)thinout_fprop_kernel<RS_, K_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        });                                                                                                                       \
        if (attr_err != hipSuccess) return GCC_ERR_LAUNCH;                                                                        \
        hipLaunchKernelGGL((thinout_fprop_kernel<RS_, K_>), dim3(p.wgs), dim3(256), lds, st, a);                                  \
    } while (0)
    const bool wide = p.Cip == 64;
    switch (K) {
        case 9: if (wide) GCC_TO_LAUNCH(128, 9); else GCC_TO_LAUNCH(64, 9); break;
        case 7: if (wide) GCC_TO_LAUNCH(128, 7); else GCC_TO_LAUNCH(64, 7); break;
        case 5: if (wide) GCC_TO_LAUNCH(128, 5); else GCC_TO_LAUNCH(64, 5); break;
        default: if (wide) GCC_TO_LAUNCH(128, 3); else GCC_TO_LAUNCH(64, 3); break;
    }
#undef GCC_TO_LAUNCH
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

// data-gradient route of gcc_conv_dgrad (conv_igemm.hip): plain dX = conv_backward_data(dY) only (no bias / activation / statistics)
int gcc_internal_thinout_dgrad(const gcc_conv_t* c, const void* dy, const void* wt, void* dx, const gcc_epilogue_t* ep, hipStream_t st) {
    ThinOutPlan p = thinout_plan(c);
    if (!p.ok || (c->Ci & 3)) return GCC_ERR_UNSUPPORTED;
    if (ep && (ep->stats_partial || ep->y2 || ep->bn || ep->bias || ep->act != GCC_ACT_NONE)) return GCC_ERR_UNSUPPORTED;
    const int K = c->KH, cus = thinout_cus();
    // small LDS footprint (k + 1 expanded rows): two workgroups per CU
    const int cols = c->N * p.strips, slots = 2 * cus;
    const int max_bands = c->H / (2 * K) > 0 ? c->H / (2 * K) : 1;
    long best = -1;
    for (int b = 1; b <= max_bands && b <= 64; b++) {
        const int bh = cdiv(c->H, b), nb = cdiv(c->H, bh);
        const long cost = (long)cdiv(cols * nb, slots) * (bh + 2 * c->pad + TO_D);
        if (best < 0 || cost < best) { best = cost; p.band_h = bh; p.bands = nb; }
    }
    p.units = cols * p.bands;
    p.wgs = p.units < slots ? p.units : slots;
    ThinOutDgradArgs a;
    a.dy = (const bf16_t*)dy; a.wt = (const bf16_t*)wt; a.dx = (bf16_t*)dx;
    a.N = c->N; a.H = c->H; a.W = c->W; a.ldx = c->ldx; a.xoff = c->xoff; a.ldy = c->ldy; a.yoff = c->yoff; a.Ci = c->Ci; a.Co = c->Co;
    a.Cop8 = ceil8(c->Co);
    a.strips = p.strips; a.bands = p.bands; a.band_h = p.band_h; a.units = p.units; a.nblk = cdiv(c->Ci, 16);
    a.dy_bytes = (uint32_t)((size_t)c->N * c->H * c->W * c->ldy * 2);
    a.wt_bytes = (uint32_t)((size_t)c->Ci * K * K * a.Cop8 * 2);
    const size_t lds = (size_t)(K + 1) * TO_SW * 64 + TO_DRING * 2048;
    switch (K) {
        case 9: hipLaunchKernelGGL(thinout_dgrad_kernel<9>, dim3(p.wgs), dim3(256), lds, st, a); break;
        case 7: hipLaunchKernelGGL(thinout_dgrad_kernel<7>, dim3(p.wgs), dim3(256), lds, st, a); break;
        case 5: hipLaunchKernelGGL(thinout_dgrad_kernel<5>, dim3(p.wgs), dim3(256), lds, st, a); break;
        default: hipLaunchKernelGGL(thinout_dgrad_kernel<3>, dim3(p.wgs), dim3(256), lds, st, a); break;
    }
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}
