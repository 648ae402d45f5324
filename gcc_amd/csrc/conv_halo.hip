// Implicit-GEMM convolution with the input neighbourhood of a tile resident in LDS ("halo" form) for the k4 s2 p1
// convolutions of the PatchGAN (models/Pix2Pix.py:267-305: L2 128 -> 256, L3 256 -> 512) -- forward, and backward-data.
//
// igemm_kernel (conv_igemm.hip) gathers, for every tap, the 256 pixel rows of its tile again: 16 taps x 256 rows x 128 B per
// 64-channel slice.  With stride 2 the taps (kh, kw), (kh, kw + 2), (kh + 2, kw), (kh + 2, kw + 2) read the SAME input
// sub-grid, one output pixel further right / down: seen through the sub-grid of parity (u, v) = (kh & 1, kw & 1) the convolution
// is a 2 x 2 stride-1 convolution.  So one 64-channel slice of that sub-grid -- (TR + 1) x (TW + 1) pixels for a TR x TW tile of
// output pixels -- is staged ONCE (LDS-DMA, [pixel][64 ch] rows of 128 B, XOR-swizzled like igemm's) and serves four k-steps;
// a k-step's pixel operand is the same image read (a, b) pixels further down / right (16 consecutive pixels of a tile row are
// 16 consecutive LDS rows, which is all the swizzle's conflict-freedom needs).  Pixel staging falls from 1024 to ~330 rows per
// four k-steps; the weights stream as before (one [256][64] tile per k-step, double buffered).
//   Measured motive (profiles/r3n_quarterpix.txt): igemm_kernel with the pixel DMA of three k-steps out of four removed (wrong
//   results, same instruction stream otherwise) runs the L2 forward in 72 us instead of 83, L3's in 103 instead of 114.
//
// Backward-data of the same convolution: output phase (py, px) = parities of the dx pixel reads a 2 x 2 neighbourhood of dy
// (dy rows Y + py - ja, columns X + px - jb for ja, jb in {0, 1}; weight taps kh = 1 - py + 2 ja, kw = 1 - px + 2 jb): the same
// structure with the dy sub-grid staged once per 64 output channels (mode 1: one phase per blockIdx.z).  Where the conv has 128
// input channels (L2's data gradient) a 256-column tile holds BOTH px phases (columns 0..127: px = 0, 128..255: px = 1; mode 2):
// the two phases' output pixels are neighbours in x, so the tile's 256 "channels" are 512 contiguous bytes of dx.
//
// Stride 1 (k4 s1 p1: PatchGAN L4, 512 -> 1024 on 32 x 32 -> 31 x 31; template parameter S1): all 16 taps share one staged
// neighbourhood, (TR + 3) x (TW + 3) pixels of a 64-channel slice serve 16 k-steps (pixel staging 1/11th of igemm_kernel's; the
// forward's positions live on the padded H x W grid, the tile rows / columns past Ho, Wo are computed and dropped).  With a
// 16 x 16 tile and an LDS pitch of 20 rows the two stages of the slice and of the weights are exactly the CU's 160 KiB.  The
// pitch is 4 mod 8, so a row shift by an odd number of sub-grid rows flips bit 2 of row & 7 -- the swizzled chunk index moves by
// 4, i.e. the address by 64 bytes, exactly like the second 32-deep k-slice does: still a compile-time XOR on one base address.
//
// Tile 256 pixels x 256 columns x 64 k, 8 waves (2 along columns x 4 along pixels, 128 x 64 per wave), one workgroup per CU;
// main loop, fragment order and epilogue are igemm_kernel<256, 256>'s.  Geometry must fit exactly (launcher: halo_plan).
#include <mutex>
#include "common.hpp"
#include "igemm_common.hpp"

namespace gcc_igemm {

struct HaloParams {
    const bf16_t* src; const bf16_t* wgt; bf16_t* dst; const float* bias; float* stats;
    int mode;                      // k4 s2 p1: 0 fprop, 1 dgrad (one phase per z), 2 dgrad, both px phases in the tile's columns;
                                   // k4 s1 p1: 3 fprop, 4 dgrad
    int N, Hs, Ws, lds_, soff;     // gather source (x / dy)
    int Hd, Wd, ldd, doff;         // destination
    int Ct, Cout, ldw;             // source channels per tap (multiple of 64), GEMM columns, weight row stride (elements)
    int act; float slope;
    uint32_t src_bytes, wgt_bytes;
    int TR, TW, lgTW;              // tile of TR x TW positions (TR * TW == 256, TW a power of two >= 16)
    int HR, HW, HWp;               // staged sub-grid: (TR + 1) x (TW + 1 or 2) pixels, LDS pitch HWp = HW rounded up to 8 rows
    int npieces;                   // 1-KiB LDS-DMA pieces of a staged slice (8 pixels each): HR * HWp / 8
    int tiles_x, tiles_y;          // tiles per image
    int ntiles;                    // column tiles
    int nchunks;                   // Ct / 64
    int xcd_cols;                  // tile order: 1: every XCD owns one column tile (launcher: ntiles divides 8, whole groups of 8 workgroups)
    int PH, PW;                    // positions that exist (mode 3: Ho x Wo of the padded H x W grid; otherwise the whole grid)
    TailFin fin;                   // BatchNorm finalize by the last-arriving workgroups (igemm_common.hpp); tickets NULL: off
};

constexpr int HB = 256;
constexpr int HALO_MAX_PIECES = 48;                // 6 per wave

// HC: columns of the tile.  256: the round-3 form (8 waves of 64 pixels x 128 columns).  128 (round 4): 8 waves of 64 x 64 -- twice
// the workgroups for the launches whose 256-column tiles cover half the chip (PatchGAN L3 forward, L4 data gradient: 128 of
// 256 CUs) and for 128-column layers; half the weight stage (the pixel slices are staged by both column tiles of a pixel tile).
// KS (stride-1 form): kernel size -- 4 (PatchGAN L4) or, round 4, 3 (k3 s1 p1: the VGG19 layers of SRGAN's perceptual loss at
// 96 -> 384: 9 taps per staged slice, neighbourhood (16 + 2) x (16 + 2) at the same pitch of 20).
template <bool S1, int HC, int KS = 4>
__global__ __launch_bounds__(512) void igemm_halo_kernel(const HaloParams p) {
    using C = Cfg<HB, HC>;
    static_assert(KS == 4 || (S1 && KS == 3), "kernel sizes: 4 (both strides), 3 (stride 1)");
    constexpr int HALO_W_BYTES = HC * BK * 2;        // one weight stage: [HC][64] bf16
    constexpr int WPIECES = HC / 64;                 // 1-KiB weight pieces per wave and k-step
    constexpr int T = S1 ? KS * KS : 4;              // k-steps (taps) a staged slice serves
    extern __shared__ __attribute__((aligned(128))) char smem[];
    char* sW = smem;                                 // weights [2][256][128 B]
    char* sH = smem + 2 * HALO_W_BYTES;              // staged sub-grid slices [2][npieces * 8][128 B]
    const int hbuf = p.npieces * 1024;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave % C::WC, wp = wave / C::WC;
    const int lr = lane & 15, lq = lane >> 4;

    const int nwg = gridDim.x;
    int mt, nt;
    if (p.xcd_cols) {
        // one column tile per XCD (GCC_OPT_HALO_XCD_COLS): workgroup b runs on XCD b & 7; the XCD's L2 then streams 1 / ntiles of the
        // weights (once: its workgroups walk K in step) and the pixel slices are fetched by ntiles XCDs instead of one
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        nt = xcd % p.ntiles;
        mt = xcd / p.ntiles + (8 / p.ntiles) * idx;
    } else {
        const int tile = xcd_remap(blockIdx.x, nwg);
        mt = tile / p.ntiles; nt = tile % p.ntiles;
    }
    const int n0 = nt * HC;
    const int tpi = p.tiles_x * p.tiles_y;
    const int img = mt / tpi, trem = mt - img * tpi;
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const int Y0 = ty * p.TR, X0 = tx * p.TW;
    const int py = p.mode == 0 ? 0 : (p.mode == 1 ? (int)blockIdx.z >> 1 : (int)blockIdx.z);
    const int pxz = p.mode == 1 ? (int)blockIdx.z & 1 : 0;

    // LDS-DMA from inline assembly (common.hpp lds_dma16): invisible to hipcc's wait-count insertion, which otherwise drains the
    // prefetch in front of the fragment reads of the stride-1 form; the loop's own vmcnt(0) + barrier order the data
    const i32x4 rs_src = make_rsrc(p.src, p.src_bytes);
    const i32x4 rs_wgt = make_rsrc(p.wgt, p.wgt_bytes);
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);

    // ---- staged sub-grid: this wave's pieces q = wave + 8 t; a lane's pixel of piece q is LDS row 8 q + (lane >> 3) ------------
    // physical 16-byte chunk lane & 7 of a row holds logical chunk (lane & 7) ^ (row & 7), and row & 7 == lane >> 3
    const int chunk = (lane & 7) ^ (lane >> 3);
    const int sstep = p.mode == 0 ? 2 : 1;           // source pixels per position step
    const float inv_pitch = 1.0f / (float)p.HWp;
    const int img_base = img * p.Hs * p.Ws;
    // stage `s` of the K loop: mode 0: s = (u * 2 + v) * nchunks + ch; modes 1, 2: s = ch
    auto issue_halo = [&](int s, int t, int buf) {
        int oy_s, ox_s, ch;
        if constexpr (S1) {
            ch = s;
            oy_s = ox_s = p.mode == 3 ? -1 : -(KS - 2);          // forward: -pad; data gradient: -(KS - 1 - pad)
        } else if (p.mode == 0) {
            const int uv = s / p.nchunks;
            ch = s - uv * p.nchunks;
            oy_s = (uv >> 1) - 1; ox_s = (uv & 1) - 1;
        } else {
            ch = s;
            oy_s = py - 1; ox_s = p.mode == 1 ? pxz - 1 : -1;
        }
        const int q = wave + 8 * t;
        if (q < p.npieces) {                          // wave-uniform
            int Yl, Xl;
            if constexpr (S1) {                       // pitch 20: a piece's 8 rows may straddle two sub-grid rows
                const int row = q * 8 + (lane >> 3);
                Yl = (int)(((float)row + 0.5f) * inv_pitch);
                Xl = row - Yl * p.HWp;
            } else {                                  // pitch a multiple of 8: the piece's sub-grid row and first column are wave-uniform
                const int ppr = p.HWp >> 3;
                Yl = q / ppr;
                Xl = (q - Yl * ppr) * 8 + (lane >> 3);
            }
            const int y = sstep * (Y0 + Yl) + oy_s, x = sstep * (X0 + Xl) + ox_s;
            const bool ok = Yl < p.HR && Xl < p.HW && (unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws;
            const uint32_t off = ok ? (uint32_t)((((img_base + y * p.Ws + x) * p.lds_ + p.soff + ch * BK) << 1) + chunk * 16) : OOB;
            lds_dma16(rs_src, lds0 + 2 * HALO_W_BYTES + buf * hbuf + q * 1024, off);
        }
    };

    // ---- weights: wave w stages rows (HC / 8) w .. of the [HC][64] tile, WPIECES 1-KiB pieces -----------------------------------
    // (the plan guarantees whole tiles: every row exists)
    const int wr0 = wave * (HC / 8) + (lane >> 3);
    const int w_row0 = (p.mode == 2 ? (wr0 & 127) : n0 + wr0) * p.ldw * 2 + chunk * 16;
    const int px_w = p.mode == 2 ? (wave >> 2) : pxz;      // column half of the rows this wave stages (mode 2)
    auto issue_w = [&](int kt, int buf) {
        const int s = kt / T, j = kt - s * T, ja = j >> 1, jb = j & 1;
        int kh, kw, ch;
        if constexpr (S1) {
            ch = s;
            kh = j / KS; kw = j - kh * KS;
        } else if (p.mode == 0) {
            const int uv = s / p.nchunks;
            ch = s - uv * p.nchunks;
            kh = 2 * ja + (uv >> 1); kw = 2 * jb + (uv & 1);
        } else {
            ch = s;
            kh = 1 - py + 2 * ja; kw = 1 - px_w + 2 * jb;
        }
        const int tapoff = ((kh * KS + kw) * p.Ct + ch * BK) * 2;
#pragma unroll
        for (int i = 0; i < WPIECES; i++) {
            const uint32_t off = (uint32_t)(w_row0 + tapoff + i * (16 * p.ldw));
            lds_dma16(rs_wgt, lds0 + buf * HALO_W_BYTES + (wave * WPIECES + i) * 1024, off);
        }
    };

    // ---- fragment rows -------------------------------------------------------------------------------------------------------------
    // pixel fragment jj of this wave: tile pixels wp * 64 + jj * 16 + lr = 16 consecutive positions of one tile row
    // pixel fragment jj of this wave = tile pixels wp * 64 + jj * 16 + lr: 16 consecutive positions of one tile row, i.e. 16
    // consecutive LDS rows.  With the pitch a multiple of 8, a fragment's row & 7 is (lr + shx) & 7 whatever the fragment and the
    // row shift: one address register per column shift shx (0, 1 or 2), everything else is a scalar byte offset.
    const int prow0 = ((wp * C::TP) >> p.lgTW) * p.HWp + ((wp * C::TP) & (p.TW - 1)) + lr;
    // (S1, pitch 4 mod 8, TW == 16: fragment jj sits jj sub-grid rows below fragment 0, whose tile row wp * 4 is even, so its
    // row & 7 is (lr + shx) & 7 with bit 2 flipped when jj + shy is odd -- the XOR in afrag below.)
    int abase[3];
#pragma unroll
    for (int sx = 0; sx < 3; sx++) abase[sx] = (prow0 + sx) * 128 + ((lq ^ ((lr + sx) & 7)) << 4);
    int pstep[C::PB];                                // scalar bytes: fragment jj starts pstep[jj] after fragment 0
#pragma unroll
    for (int jj = 0; jj < C::PB; jj++) {
        const int a0 = wp * C::TP, a1 = a0 + jj * 16;
        pstep[jj] = (((a1 >> p.lgTW) - (a0 >> p.lgTW)) * p.HWp + ((a1 & (p.TW - 1)) - (a0 & (p.TW - 1)))) * 128;
    }
    const int px_c = p.mode == 2 ? wc : 0;           // column half this wave computes (mode 2)

    f32x4 acc[C::CB][C::PB];
#pragma unroll
    for (int i = 0; i < C::CB; i++)
#pragma unroll
        for (int j = 0; j < C::PB; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // A k-step: both 32-deep slices' fragments pass through registers.  Slice 0: all twelve fragments up front.  Slice 1: the
    // weight fragment i is read right after the four MFMAs that were the last users of slice 0's weight fragment i -- it can
    // take that one's registers -- and the four pixel fragments into registers of their own, one per two weight fragments
    // (128 accumulators + 48 + 16 fragment registers instead of 128 + 96: igemm_kernel<256, 256> sits at the 256-register
    // limit with the second slice held apart, and this kernel has more address arithmetic alive).
    // (fragment addresses are formed as 32-bit LDS addresses: XOR arithmetic through a generic pointer made hipcc emit
    // flat_load_dwordx4, whose waits cover the vector-memory counter too -- the prefetch was drained in front of the MFMAs)
    typedef __attribute__((address_space(3))) const bf16x8* lds_frag_ptr;
    auto compute = [&](int wbuf, int hb, int shy, int shx) {
        uint32_t a = lds0 + 2 * HALO_W_BYTES + hb * hbuf + shy * p.HWp * 128;
        if constexpr (S1) {          // shy, shx are run-time (wave-uniform) here: the 16-tap loop is not unrolled
            const int t = prow0 + shx;
            a += (t << 7) + ((lq ^ (t & 7)) << 4);
            a ^= (shy & 1) * 64;     // odd row shift: chunk index ^ 4 (pitch 4 mod 8)
        } else {
            a += abase[shx];
        }
        const char* w = sW + wbuf * HALO_W_BYTES;
        auto wfrag = [&](int ks, int i) {
            const int row = wc * C::TC + i * 16 + lr;
            return *(const bf16x8*)(w + row * 128 + (((ks * 4 + lq) ^ (row & 7)) << 4));
        };
        auto afrag = [&](int ks, int j) {             // slice 1 = chunk ^ 4: the address differs in bit 6
            const int flip = S1 ? ((ks + j) & 1) : ks;     // fragment j sits j sub-grid rows further down (S1)
            return *(lds_frag_ptr)(uintptr_t)((a + pstep[j]) ^ (uint32_t)(flip * 64));
        };
        bf16x8 fw0[C::CB], fa0[C::PB], fw1[C::CB], fa1[C::PB];
#pragma unroll
        for (int i = 0; i < C::CB; i++) fw0[i] = wfrag(0, i);
#pragma unroll
        for (int j = 0; j < C::PB; j++) fa0[j] = afrag(0, j);
#pragma unroll
        for (int i = 0; i < C::CB; i++) {
#pragma unroll
            for (int j = 0; j < C::PB; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw0[i], fa0[j], acc[i][j], 0, 0, 0);
            fw1[i] = wfrag(1, i);
            if constexpr (C::CB == 8) { if (i & 1) fa1[i >> 1] = afrag(1, i >> 1); }
            else fa1[i] = afrag(1, i);
        }
#pragma unroll
        for (int i = 0; i < C::CB; i++)
#pragma unroll
            for (int j = 0; j < C::PB; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw1[i], fa1[j], acc[i][j], 0, 0, 0);
        static_assert((C::CB == 8 || C::CB == 4) && C::PB == 4, "schedules below: 128 x 64 or 64 x 64 per wave");
        __builtin_amdgcn_sched_group_barrier(0x100, C::CB + C::PB, 0);
        if constexpr (C::CB == 8) {
#pragma unroll
            for (int i = 0; i < C::CB / 2; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, C::PB, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, C::PB, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < C::CB; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, C::PB, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x008, C::CB * C::PB, 0);
    };

    // ---- K loop ----------------------------------------------------------------------------------------------------------------------
    // one barrier per k-step, as in igemm_kernel: [everything issued a step ago has landed for every wave AND everyone left the
    // buffers of step kt - 1] -> issue the weights of step kt + 1 and (in the first three steps of a stage) a third of the next
    // stage's sub-grid slice -> multiply step kt.  A slice is complete a full k-step before its first read.
    const int nstages = (!S1 && p.mode == 0) ? 4 * p.nchunks : p.nchunks;
    const int nk = T * nstages;
#pragma unroll
    for (int t = 0; t < 6; t++) issue_halo(0, t, 0);
    issue_w(0, 0);
    if constexpr (S1) {
        auto step = [&](int s, int j, int wbuf) {                 // wbuf is a literal at both call sites
            const int kt = T * s + j;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < nk) issue_w(kt + 1, wbuf ^ 1);
            if (j < 6 && s + 1 < nstages) issue_halo(s + 1, j, (s + 1) & 1);
            const int kh = j / KS, kw = j - kh * KS;
            compute(wbuf, s & 1, p.mode == 3 ? kh : KS - 1 - kh, p.mode == 3 ? kw : KS - 1 - kw);
        };
        if constexpr ((T & 1) == 0) {
            for (int s = 0; s < nstages; s++) {
#pragma unroll 1
                for (int j = 0; j < T; j += 2) {
                    step(s, j, 0);
                    step(s, j + 1, 1);
                }
            }
        } else {
            // an odd number of taps per slice: the weight stage alternates across slice boundaries -- pairs of k-steps over the
            // flat index (s, j advance as scalars)
            int s = 0, j = 0;
#pragma unroll 1
            for (int kt = 0; kt < nk; kt += 2) {
                step(s, j, 0);
                if (++j == T) { j = 0; ++s; }
                if (kt + 1 < nk) {
                    step(s, j, 1);
                    if (++j == T) { j = 0; ++s; }
                }
            }
        }
    } else {
        for (int s = 0; s < nstages; s++) {
#pragma unroll
            for (int j = 0; j < T; j++) {
                const int kt = T * s + j;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (kt + 1 < nk) issue_w(kt + 1, (kt + 1) & 1);
                if (j < 3 && s + 1 < nstages) {
                    issue_halo(s + 1, 2 * j, (s + 1) & 1);
                    issue_halo(s + 1, 2 * j + 1, (s + 1) & 1);
                }
                const int ja = j >> 1, jb = j & 1;
                const int shy = p.mode == 0 ? ja : 1 - ja;
                const int shx = p.mode == 0 ? jb : (p.mode == 1 ? 1 - jb : px_c + 1 - jb);
                compute(kt & 1, s & 1, shy, shx);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- epilogue: bias + activation + bf16 rounding, LDS transpose, coalesced NHWC stores, BatchNorm partial sums ---------------
    // acc[i][j][r]: column = wc * 128 + i * 16 + 4 lq + r ; tile pixel = wp * 64 + j * 16 + lr
    char* sO = smem;
#pragma unroll
    for (int i = 0; i < C::CB; i++) {
        const int cl = wc * C::TC + i * 16 + 4 * lq;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
            const int cb = p.mode == 2 ? (cl & 127) : n0 + cl;
#pragma unroll
            for (int r = 0; r < 4; r++) bv[r] = (cb + r < p.Cout) ? p.bias[cb + r] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < C::PB; j++) {
            const int pl = wp * C::TP + j * 16 + lr;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = acc[i][j][r] + bv[r];
            apply_actN<4>(v, v, p.act, p.slope);
            if constexpr (S1) {       // positions past the output keep exact zeros (the statistics below sum every row of the tile)
                if (Y0 + (pl >> p.lgTW) >= p.PH || X0 + (pl & (p.TW - 1)) >= p.PW) v[0] = v[1] = v[2] = v[3] = 0.f;
            }
            i32x2 pk;
            pk[0] = (int)pack2bf(v[0], v[1]);
            pk[1] = (int)pack2bf(v[2], v[3]);
            *(i32x2*)(sO + pl * C::OSTRIDE + cl * 2) = pk;
        }
    }
    __syncthreads();
    constexpr int CPR = HC / 8;                  // 16-byte chunks per tile row
    constexpr int NCH = HB * CPR;
    for (int q = tid; q < NCH; q += C::NT) {
        const int row = q / CPR, cch = q - row * CPR;
        const int yl = row >> p.lgTW, xl = row & (p.TW - 1);
        int oy, ox, ch;
        if constexpr (S1) {
            oy = Y0 + yl; ox = X0 + xl; ch = n0 + cch * 8;
            if (oy >= p.PH || ox >= p.PW) continue;
        }
        else if (p.mode == 0) { oy = Y0 + yl; ox = X0 + xl; ch = n0 + cch * 8; }
        else if (p.mode == 1) { oy = 2 * (Y0 + yl) + py; ox = 2 * (X0 + xl) + pxz; ch = n0 + cch * 8; }
        else { oy = 2 * (Y0 + yl) + py; ox = 2 * (X0 + xl) + (cch >> 4); ch = (cch & 15) * 8; }
        const size_t o = ((size_t)(img * p.Hd + oy) * p.Wd + ox) * p.ldd + p.doff + ch;
        *(i32x4*)(p.dst + o) = *(const i32x4*)(sO + row * C::OSTRIDE + cch * 16);
    }
    if (p.stats) {                               // forward only: one row of partial sums per tile, as igemm_kernel<256, 256> writes them
        float* sR = (float*)(smem + HB * C::OSTRIDE);
        constexpr int PARTS = C::NT / HC, ROWS = HB / PARTS;
        const int c = tid % HC, part = tid / HC;
        float s1 = 0.f, s2 = 0.f;
        for (int r = part * ROWS; r < (part + 1) * ROWS; r++) {
            const float v = bf2f(*(const bf16_t*)(sO + r * C::OSTRIDE + c * 2));
            s1 += v; s2 += v * v;
        }
        sR[tid] = s1; sR[C::NT + tid] = s2;
        __syncthreads();
        if (tid < HC && n0 + tid < p.Cout) {
            float ts = 0.f, tss = 0.f;
#pragma unroll
            for (int q = 0; q < PARTS; q++) { ts += sR[q * HC + tid]; tss += sR[C::NT + q * HC + tid]; }
            st_stat(p.stats + ((size_t)mt * 2 + 0) * p.Cout + n0 + tid, ts, p.fin.tickets != nullptr);
            st_stat(p.stats + ((size_t)mt * 2 + 1) * p.Cout + n0 + tid, tss, p.fin.tickets != nullptr);
        }
        if (p.fin.tickets) stats_tail<C::NT>(p.fin, p.stats, p.Cout, mt, (int*)(smem + HB * C::OSTRIDE + 2 * C::NT * 4), tid);
    }
}

// ---- launcher side ------------------------------------------------------------------------------------------------------------------
HaloPlan halo_plan(const gcc_conv_t* c, int dgrad) {
    HaloPlan h = {};
    if (!gcc_opt(GCC_OPT_IGEMM_HALO)) return h;
    const bool k3 = c->KH == 3 && c->KW == 3 && c->pad == 1 && c->stride == 1 && gcc_opt(GCC_OPT_IGEMM_HALO) >= 3;
    if (!k3 && (c->KH != 4 || c->KW != 4 || c->pad != 1)) return h;
    const int Ct = dgrad ? c->Co : c->Ci, Cout = dgrad ? c->Ci : c->Co;
    if (Ct % BK || Ct < BK) return h;
    int gh, gw;                                   // position grid
    if (c->stride == 2) {
        if ((c->H & 1) || (c->W & 1)) return h;
        gh = c->H / 2; gw = c->W / 2;
        if (!dgrad) { if (Cout % 128) return h; h.mode = 0; }
        else if (Cout == 128 && c->plan.halo_hc != 128) h.mode = 2;
        else if (Cout % 128 == 0) h.mode = 1;
        else return h;
        int tw = 256;
        while (tw > gw) tw >>= 1;
        if (tw < 16 || gw % tw) return h;
        h.TW = tw; h.TR = 256 / tw;
        h.HR = h.TR + 1; h.HW = tw + (h.mode == 2 ? 2 : 1);
        h.HWp = (h.HW + 7) & ~7;
        h.phases = h.mode == 0 ? 1 : (h.mode == 1 ? 4 : 2);
    } else if (c->stride == 1) {
        if (gcc_opt(GCC_OPT_IGEMM_HALO) < 2) return h;            // >= 2: the stride-1 form too (3, the default: also its 3 x 3 instantiation)
        if (Cout % 128) return h;
        gh = c->H; gw = c->W;                     // forward: the padded grid (k4: Ho = H - 1 rows exist; k3: all of them)
        h.mode = dgrad ? 4 : 3;
        h.TW = 16; h.TR = 16;
        h.HR = h.HW = 16 + c->KH - 1; h.HWp = 20;
        h.phases = 1;
    } else return h;
    if (gw % h.TW || gh % h.TR) return h;
    h.lgTW = 0;
    while ((1 << h.lgTW) < h.TW) h.lgTW++;
    h.npieces = (h.HR * h.HWp + 7) / 8;
    if (h.npieces > HALO_MAX_PIECES) return h;
    h.tiles_x = gw / h.TW; h.tiles_y = gh / h.TR;
    // columns per tile: 256 (one workgroup per CU, the least CU time per FLOP), or 128 -- twice the workgroups -- where the layer
    // has no 256-column tiling, where 256-column tiles are too few to be routed here at all, or (gcc_conv_t.plan.halo_hc 1; the models'
    // single-stream plan sets it together with plan.pair) where they would cover only half the chip
    const long pix_tiles = (long)c->N * h.tiles_x * h.tiles_y * h.phases;
    const int pref = c->plan.halo_hc;
    h.hc = 256;
    if (h.mode != 2) {
        const bool can256 = Cout % 256 == 0, can128 = Cout % 128 == 0;
        const long w256 = can256 ? pix_tiles * (Cout / 256) : 0;
        if (pref == 128 && can128) h.hc = 128;
        else if (pref == 256 && can256) h.hc = 256;
        else if (!can256 || w256 < plan_or(c->plan.big_min, PLAN_BIG_MIN) || (pref == 1 && w256 < 192)) h.hc = can128 ? 128 : 256;
        if (Cout % h.hc) return h;
    }
    h.ntiles = h.mode == 2 ? 1 : Cout / h.hc;
    const size_t loop = 2 * (size_t)h.hc * BK * 2 + 2 * (size_t)h.npieces * 1024;
    const size_t epi = h.hc == 256 ? (size_t)Cfg<HB, 256>::LDS_BYTES_EPI : (size_t)Cfg<HB, 128>::LDS_BYTES_EPI;
    h.lds = loop > epi ? loop : epi;
    if (h.lds > 160 * 1024) return h;
    h.wgs = pix_tiles * h.ntiles;
    h.ok = 1;
    return h;
}

int launch_halo(const gcc_conv_t* c, int dgrad, const HaloPlan& h, const void* src, const void* w, void* dst, const gcc_epilogue_t* ep,
                const TailFin* fin, hipStream_t st) {
    HaloParams p;
    p.fin = fin ? *fin : TailFin{};
    const int Ho = gcc_conv_out(c->H, c->KH, c->stride, 1), Wo = gcc_conv_out(c->W, c->KW, c->stride, 1);
    p.src = (const bf16_t*)src; p.wgt = (const bf16_t*)w; p.dst = (bf16_t*)dst;
    p.bias = ep ? ep->bias : nullptr; p.stats = ep ? ep->stats_partial : nullptr;
    p.act = ep ? ep->act : GCC_ACT_NONE; p.slope = ep ? ep->slope : 0.f;
    p.mode = h.mode; p.N = c->N;
    if (!dgrad) {
        p.Hs = c->H; p.Ws = c->W; p.lds_ = c->ldx; p.soff = c->xoff; p.Hd = Ho; p.Wd = Wo; p.ldd = c->ldy; p.doff = c->yoff;
        p.Ct = c->Ci; p.Cout = c->Co;
    } else {
        p.Hs = Ho; p.Ws = Wo; p.lds_ = c->ldy; p.soff = c->yoff; p.Hd = c->H; p.Wd = c->W; p.ldd = c->ldx; p.doff = c->xoff;
        p.Ct = c->Co; p.Cout = c->Ci;
    }
    p.ldw = c->KH * c->KW * p.Ct;
    const size_t sb = (size_t)p.N * p.Hs * p.Ws * p.lds_ * 2, wb = (size_t)p.Cout * p.ldw * 2;
    const size_t db = (size_t)p.N * p.Hd * p.Wd * p.ldd * 2;
    if (sb >= OOB || wb >= OOB || db >= (size_t)1 << 32) return -1;
    p.src_bytes = (uint32_t)sb; p.wgt_bytes = (uint32_t)wb;
    p.TR = h.TR; p.TW = h.TW; p.lgTW = h.lgTW; p.HR = h.HR; p.HW = h.HW; p.HWp = h.HWp; p.npieces = h.npieces;
    p.tiles_x = h.tiles_x; p.tiles_y = h.tiles_y; p.ntiles = h.ntiles; p.nchunks = p.Ct / BK;
    p.PH = h.mode == 3 ? Ho : h.tiles_y * h.TR; p.PW = h.mode == 3 ? Wo : h.tiles_x * h.TW;
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute((const void*)igemm_halo_kernel<false, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)igemm_halo_kernel<true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)igemm_halo_kernel<false, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)igemm_halo_kernel<true, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)igemm_halo_kernel<true, 256, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)igemm_halo_kernel<true, 128, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const dim3 grid((unsigned)(c->N * h.tiles_x * h.tiles_y * h.ntiles), 1, h.phases);
    {
        const int mode = gcc_opt(GCC_OPT_HALO_XCD_COLS);       // 1: the stride-1 form (L4: 16.8 MB of weights); 2: every form
        const unsigned pt = grid.x / h.ntiles;
        p.xcd_cols = (mode >= 2 || (mode == 1 && h.mode >= 3)) && h.ntiles >= 2 && 8 % h.ntiles == 0 && grid.x % 8 == 0 &&
                     pt % (8 / h.ntiles) == 0;
    }
    if (h.mode >= 3 && c->KH == 3) {
        if (h.hc == 256) hipLaunchKernelGGL((igemm_halo_kernel<true, 256, 3>), grid, dim3(512), h.lds, st, p);
        else hipLaunchKernelGGL((igemm_halo_kernel<true, 128, 3>), grid, dim3(512), h.lds, st, p);
    } else if (h.hc == 256) {
        if (h.mode >= 3) hipLaunchKernelGGL((igemm_halo_kernel<true, 256>), grid, dim3(512), h.lds, st, p);
        else hipLaunchKernelGGL((igemm_halo_kernel<false, 256>), grid, dim3(512), h.lds, st, p);
    } else {
        if (h.mode >= 3) hipLaunchKernelGGL((igemm_halo_kernel<true, 128>), grid, dim3(512), h.lds, st, p);
        else hipLaunchKernelGGL((igemm_halo_kernel<false, 128>), grid, dim3(512), h.lds, st, p);
    }
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

}  // namespace gcc_igemm
