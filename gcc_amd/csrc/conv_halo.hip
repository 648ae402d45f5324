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
// Tile 256 pixels x 256 columns x 64 k, 8 waves (2 along columns x 4 along pixels, 128 x 64 per wave), one workgroup per CU;
// main loop, fragment order and epilogue are igemm_kernel<256, 256>'s.  Geometry must fit exactly (launcher: halo_plan).
#include <mutex>
#include "common.hpp"
#include "igemm_common.hpp"

namespace gcc_igemm {

struct HaloParams {
    const bf16_t* src; const bf16_t* wgt; bf16_t* dst; const float* bias; float* stats;
    int mode;                      // 0 fprop, 1 dgrad (one phase per z), 2 dgrad, both px phases in the tile's columns
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
};

constexpr int HB = 256, HC = 256;
using HCfg = Cfg<HB, HC>;
constexpr int HALO_W_BYTES = HC * BK * 2;          // one weight stage: [256][64] bf16
constexpr int HALO_MAX_PIECES = 48;                // 6 per wave

__global__ __launch_bounds__(512) void igemm_halo_kernel(const HaloParams p) {
    using C = HCfg;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sW = smem;                                 // weights [2][256][128 B]
    char* sH = smem + 2 * HALO_W_BYTES;              // staged sub-grid slices [2][npieces * 8][128 B]
    const int hbuf = p.npieces * 1024;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave % C::WC, wp = wave / C::WC;
    const int lr = lane & 15, lq = lane >> 4;

    const int nwg = gridDim.x;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int mt = tile / p.ntiles, nt = tile % p.ntiles;
    const int n0 = nt * HC;
    const int tpi = p.tiles_x * p.tiles_y;
    const int img = mt / tpi, trem = mt - img * tpi;
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const int Y0 = ty * p.TR, X0 = tx * p.TW;
    const int py = p.mode == 0 ? 0 : (p.mode == 1 ? (int)blockIdx.z >> 1 : (int)blockIdx.z);
    const int pxz = p.mode == 1 ? (int)blockIdx.z & 1 : 0;

    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wgt = __builtin_amdgcn_make_buffer_rsrc((void*)p.wgt, 0, p.wgt_bytes, 0x00020000);

    // ---- staged sub-grid: this wave's pieces q = wave + 8 t; a lane's pixel of piece q is LDS row 8 q + (lane >> 3) ------------
    // physical 16-byte chunk lane & 7 of a row holds logical chunk (lane & 7) ^ (row & 7), and row & 7 == lane >> 3
    const int chunk = (lane & 7) ^ (lane >> 3);
    const int sstep = p.mode == 0 ? 2 : 1;           // source pixels per position step
    const int img_base = img * p.Hs * p.Ws;
    // stage `s` of the K loop: mode 0: s = (u * 2 + v) * nchunks + ch; modes 1, 2: s = ch
    auto issue_halo = [&](int s, int t, int buf) {
        int oy_s, ox_s, ch;
        if (p.mode == 0) {
            const int uv = s / p.nchunks;
            ch = s - uv * p.nchunks;
            oy_s = (uv >> 1) - 1; ox_s = (uv & 1) - 1;
        } else {
            ch = s;
            oy_s = py - 1; ox_s = p.mode == 1 ? pxz - 1 : -1;
        }
        const int q = wave + 8 * t;
        if (q < p.npieces) {                          // wave-uniform
            // a sub-grid row occupies HWp (a multiple of 8) LDS rows: the piece's sub-grid row and first column are wave-uniform
            const int ppr = p.HWp >> 3;
            const int Yl = q / ppr;
            const int Xl = (q - Yl * ppr) * 8 + (lane >> 3);
            const int y = sstep * (Y0 + Yl) + oy_s, x = sstep * (X0 + Xl) + ox_s;
            const bool ok = Xl < p.HW && (unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws;
            const uint32_t off = ok ? (uint32_t)((((img_base + y * p.Ws + x) * p.lds_ + p.soff + ch * BK) << 1) + chunk * 16) : OOB;
            char* dst = sH + buf * hbuf + q * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, LDS_PTR(void, dst), 16, off, 0, 0, 0);
        }
    };

    // ---- weights: wave w stages rows 32 w .. 32 w + 31 of the [256][64] tile, four 1-KiB pieces --------------------------------
    // (the plan guarantees whole tiles: every row exists)
    const int wr0 = wave * 32 + (lane >> 3);
    const int w_row0 = (p.mode == 2 ? (wr0 & 127) : n0 + wr0) * p.ldw * 2 + chunk * 16;
    const int px_w = p.mode == 2 ? (wave >> 2) : pxz;      // column half of the rows this wave stages (mode 2)
    auto issue_w = [&](int kt, int buf) {
        const int s = kt >> 2, j = kt & 3, ja = j >> 1, jb = j & 1;
        int kh, kw, ch;
        if (p.mode == 0) {
            const int uv = s / p.nchunks;
            ch = s - uv * p.nchunks;
            kh = 2 * ja + (uv >> 1); kw = 2 * jb + (uv & 1);
        } else {
            ch = s;
            kh = 1 - py + 2 * ja; kw = 1 - px_w + 2 * jb;
        }
        const int tapoff = ((kh * 4 + kw) * p.Ct + ch * BK) * 2;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t off = (uint32_t)(w_row0 + tapoff + i * (16 * p.ldw));
            char* dst = sW + buf * HALO_W_BYTES + (wave * 4 + i) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, LDS_PTR(void, dst), 16, off, 0, 0, 0);
        }
    };

    // ---- fragment rows -------------------------------------------------------------------------------------------------------------
    // pixel fragment jj of this wave: tile pixels wp * 64 + jj * 16 + lr = 16 consecutive positions of one tile row
    // pixel fragment jj of this wave = tile pixels wp * 64 + jj * 16 + lr: 16 consecutive positions of one tile row, i.e. 16
    // consecutive LDS rows.  With the pitch a multiple of 8, a fragment's row & 7 is (lr + shx) & 7 whatever the fragment and the
    // row shift: one address register per column shift shx (0, 1 or 2), everything else is a scalar byte offset.
    const int prow0 = ((wp * C::TP) >> p.lgTW) * p.HWp + ((wp * C::TP) & (p.TW - 1)) + lr;
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
    auto compute = [&](int wbuf, int hb, int shy, int shx) {
        const char* a = sH + (hb * hbuf + shy * p.HWp * 128) + abase[shx];
        const char* w = sW + wbuf * HALO_W_BYTES;
        auto wfrag = [&](int ks, int i) {
            const int row = wc * C::TC + i * 16 + lr;
            return *(const bf16x8*)(w + row * 128 + (((ks * 4 + lq) ^ (row & 7)) << 4));
        };
        auto afrag = [&](int ks, int j) {             // slice 1 = chunk ^ 4: the address differs in bit 6
            return *(const bf16x8*)((const char*)((uintptr_t)(a + pstep[j]) ^ (uintptr_t)(ks * 64)));
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
            if (i & 1) fa1[i >> 1] = afrag(1, i >> 1);
        }
#pragma unroll
        for (int i = 0; i < C::CB; i++)
#pragma unroll
            for (int j = 0; j < C::PB; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw1[i], fa1[j], acc[i][j], 0, 0, 0);
        static_assert(C::CB == 8 && C::PB == 4, "schedule below is written for 128 x 64 per wave");
        __builtin_amdgcn_sched_group_barrier(0x100, C::CB + C::PB, 0);
#pragma unroll
        for (int i = 0; i < C::CB / 2; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, C::PB, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, C::PB, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, C::CB * C::PB, 0);
    };

    // ---- K loop ----------------------------------------------------------------------------------------------------------------------
    // one barrier per k-step, as in igemm_kernel: [everything issued a step ago has landed for every wave AND everyone left the
    // buffers of step kt - 1] -> issue the weights of step kt + 1 and (in the first three steps of a stage) a third of the next
    // stage's sub-grid slice -> multiply step kt.  A slice is complete a full k-step before its first read.
    const int nstages = p.mode == 0 ? 4 * p.nchunks : p.nchunks;
    const int nk = 4 * nstages;
#pragma unroll
    for (int t = 0; t < 6; t++) issue_halo(0, t, 0);
    issue_w(0, 0);
    for (int s = 0; s < nstages; s++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int kt = 4 * s + j;
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
        if (p.mode == 0) { oy = Y0 + yl; ox = X0 + xl; ch = n0 + cch * 8; }
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
            p.stats[((size_t)mt * 2 + 0) * p.Cout + n0 + tid] = ts;
            p.stats[((size_t)mt * 2 + 1) * p.Cout + n0 + tid] = tss;
        }
    }
}

// ---- launcher side ------------------------------------------------------------------------------------------------------------------
HaloPlan halo_plan(const gcc_conv_t* c, int dgrad) {
    HaloPlan h = {};
    if (!gcc_opt(GCC_OPT_IGEMM_HALO)) return h;
    if (c->KH != 4 || c->KW != 4 || c->stride != 2 || c->pad != 1 || (c->H & 1) || (c->W & 1)) return h;
    const int Ho = c->H / 2, Wo = c->W / 2;
    const int Ct = dgrad ? c->Co : c->Ci, Cout = dgrad ? c->Ci : c->Co;
    if (Ct % BK || Ct < BK) return h;
    if (!dgrad) { if (Cout % HC) return h; h.mode = 0; }
    else if (Cout == 128) h.mode = 2;
    else if (Cout % HC == 0) h.mode = 1;
    else return h;
    int tw = 256;
    while (tw > Wo) tw >>= 1;
    if (tw < 16 || Wo % tw) return h;
    const int tr = 256 / tw;
    if (Ho % tr) return h;
    h.TW = tw; h.TR = tr;
    h.lgTW = 0;
    while ((1 << h.lgTW) < tw) h.lgTW++;
    h.HR = tr + 1; h.HW = tw + (h.mode == 2 ? 2 : 1);
    h.HWp = (h.HW + 7) & ~7;
    h.npieces = h.HR * h.HWp / 8;
    if (h.npieces > HALO_MAX_PIECES) return h;
    h.tiles_x = Wo / tw; h.tiles_y = Ho / tr;
    h.ntiles = h.mode == 2 ? 1 : Cout / HC;
    h.phases = h.mode == 0 ? 1 : (h.mode == 1 ? 4 : 2);
    const size_t loop = 2 * (size_t)HALO_W_BYTES + 2 * (size_t)h.npieces * 1024;
    const size_t epi = (size_t)HCfg::LDS_BYTES_EPI;
    h.lds = loop > epi ? loop : epi;
    if (h.lds > 160 * 1024) return h;
    h.wgs = (long)c->N * h.tiles_x * h.tiles_y * h.ntiles * h.phases;
    h.ok = 1;
    return h;
}

int launch_halo(const gcc_conv_t* c, int dgrad, const HaloPlan& h, const void* src, const void* w, void* dst, const gcc_epilogue_t* ep,
                hipStream_t st) {
    HaloParams p;
    const int Ho = c->H / 2, Wo = c->W / 2;
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
    p.ldw = 16 * p.Ct;
    const size_t sb = (size_t)p.N * p.Hs * p.Ws * p.lds_ * 2, wb = (size_t)p.Cout * p.ldw * 2;
    const size_t db = (size_t)p.N * p.Hd * p.Wd * p.ldd * 2;
    if (sb >= OOB || wb >= OOB || db >= (size_t)1 << 32) return -1;
    p.src_bytes = (uint32_t)sb; p.wgt_bytes = (uint32_t)wb;
    p.TR = h.TR; p.TW = h.TW; p.lgTW = h.lgTW; p.HR = h.HR; p.HW = h.HW; p.HWp = h.HWp; p.npieces = h.npieces;
    p.tiles_x = h.tiles_x; p.tiles_y = h.tiles_y; p.ntiles = h.ntiles; p.nchunks = p.Ct / BK;
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute((const void*)igemm_halo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const dim3 grid((unsigned)(c->N * h.tiles_x * h.tiles_y * h.ntiles), 1, h.phases);
    hipLaunchKernelGGL(igemm_halo_kernel, grid, dim3(512), h.lds, st, p);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

}  // namespace gcc_igemm
