// 3 x 3, stride 1, pad 1 convolutions between NARROW layers (8..64 channels on either side) at a large spatial size: the 33
// Conv2d(64, 64, 3, 1, 1) of SRGAN's SRResNet trunk (models/SRGAN.py:59-81, 16 residual blocks + the conv behind them; the
// student's pruned widths) and VGG19's conv1_2 at the high resolution (models/SRGAN.py: the content loss).  As an implicit GEMM
// these are M = 147 k .. 2.4 M pixels x N = 64 x K = 576: the 128 x 64 tiles of igemm_kernel re-stage every input pixel nine times
// through LDS for a K loop of 18 steps and ran at ~6 % of the matrix peak (round 5: 12 % of the serialized kernel time of the
// 96 -> 384 step).
//
// Same walk as conv_thinout.hip: a workgroup owns a strip of 32 or 64 output columns of a band of rows of one image and walks
// down the rows.  The rows of the input pass ONCE through a ring in LDS (LDS-DMA, XOR-swizzled [pixel][channel] images, three rows
// in flight behind a counted s_waitcnt); all nine taps x both 32-channel halves of the weights of a wave's 16 / 32 output channels
// stay in registers for the whole launch (A operand: [16 channels] x [32 source channels] per tap); a horizontal tap is a shift of
// the pixel operand's LDS address, a vertical tap another slot of the ring.  A row of output is rounded, staged through LDS and
// written as full 128-byte lines; BatchNorm partial sums (of the rounded values) are kept per thread over the whole walk and
// leave as ONE row per workgroup.  The data gradient is the same kernel over dY with the taps mirrored (flip) and the dgrad packing.
#include <stdlib.h>
#include <mutex>
#include "common.hpp"
#include "igemm_common.hpp"

namespace {
using gcc_igemm::OOB;
using gcc_igemm::TailFin;

constexpr int R3_D = 3;                 // rows in flight ahead of the row the products need
constexpr int R3_RING = R3_D + 3;       // + the three rows being multiplied
constexpr int R3_OROW = 128;            // bytes of one pixel of the staged output row (<= 64 channels)

struct Ring3Args {
    const bf16_t* src; const bf16_t* w; bf16_t* dst; const float* bias; float* stats;
    int N, H, W, lds_, soff, ldd, doff;
    int Cs, Cd;             // source / destination channels
    int Cs8;                // source channels per tap of the packed weights (rounded up to 8)
    int Cstat;              // channels of a statistics row (= Cd)
    int flip;               // 1: taps mirrored (data gradient)
    int act; float slope;
    int strips, bands, band_h, units;
    uint32_t src_bytes, dst_bytes;
    TailFin fin;            // tickets != NULL: the BatchNorm behind the conv is finalized by the last-arriving workgroups (igemm_common.hpp)
};

// physical byte offset of logical 16-byte chunk `ch` of pixel row `r` (rows of RS = 64 or 128 bytes): 32-byte windows XOR-swizzled
// with the row (conv_thinout.hip's image layout)
template <int RS>
__device__ __forceinline__ int r3_off(int r, int ch) {
    const int w = ch >> 1, sub = ch & 1;
    const int pw = RS == 128 ? (w ^ ((r >> 1) & 3)) : (w ^ ((r >> 2) & 1));
    return r * RS + pw * 32 + sub * 16;
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// RS: bytes of a staged source pixel (64: <= 32 channels, 128: <= 64).  NBW: 16-pixel blocks per wave (strip = 32 NBW pixels).
// DB: 32-channel halves of the destination (waves 0/1 -> wm: channels 16 DB wm ..; waves -> wn = wave >> 1: pixels 16 NBW wn ..)
template <int RS, int NBW, int DB>
__global__ __launch_bounds__(256, 2) void ring3_kernel(const Ring3Args a) {
    constexpr int SW = 32 * NBW;                                  // output pixels of a strip
    constexpr int PXR = 1024 / RS;                                // pixels per LDS-DMA piece
    constexpr int SWP = ((SW + 2 + PXR - 1) / PXR) * PXR;         // staged pixels: SW + 2, rounded up to whole pieces
    constexpr int PIECES = SWP / PXR;
    constexpr int MAXP = (PIECES + 3) / 4;                        // pieces every wave issues per row (the surplus lands in a dump area)
    constexpr int ROWB = PIECES * 1024;
    constexpr int CC = RS / 64;                                   // 32-channel k-steps of the source
    constexpr int MBW = DB;                                       // 16-channel destination blocks per wave
    constexpr int NCH = DB * 4;                                   // 16-byte chunks of an output pixel
    constexpr int ITEMS = SW * NCH;                               // chunks of an output row
    constexpr int S = ITEMS >= 256 ? ITEMS / 256 : 1;             // stores per thread and row (always issued: out-of-range ones are dropped)
    constexpr int O_BASE = R3_RING * ROWB, DUMP = O_BASE + SW * R3_OROW;
    constexpr int CPR = RS / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sX = smem;                                              // [R3_RING][SWP px][RS]
    char* sO = smem + O_BASE;                                     // [SW px][128 B], 16-byte chunks XOR-ed with the pixel

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, i = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const i32x4 rs_src = make_rsrc(a.src, a.src_bytes);
    const __amdgpu_buffer_rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc((void*)a.dst, 0, a.dst_bytes, 0x00020000);
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);

    // weight operands, once per launch: A[dst channel 16 (MBW wm + mb) + i][source channels 32 cc + 8 g .. + 7] of every tap
    bf16x8 wf[MBW][9][CC];
    float bv[MBW][4];
#pragma unroll
    for (int mb = 0; mb < MBW; mb++) {
        const int co = (wm * MBW + mb) * 16 + i;
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int wt = a.flip ? 8 - t : t;
#pragma unroll
            for (int cc = 0; cc < CC; cc++) {
                const int ch = cc * 32 + 8 * g;
                const bool ok = co < a.Cd && ch < a.Cs8;
                const bf16x8 z = {};
                wf[mb][t][cc] = ok ? *(const bf16x8*)(a.w + ((size_t)co * 9 + wt) * a.Cs8 + ch) : z;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int c = (wm * MBW + mb) * 16 + 4 * g + r;
            bv[mb][r] = (a.bias && c < a.Cd) ? a.bias[c] : 0.f;
        }
    }
    float st_s[8], st_q[8];
#pragma unroll
    for (int j = 0; j < 8; j++) st_s[j] = st_q[j] = 0.f;

    for (int unit = blockIdx.x; unit < a.units; unit += gridDim.x) {
        int b = unit;
        const int band = b % a.bands; b /= a.bands;
        const int strip = b % a.strips;
        const int n = b / a.strips;
        const int xs = strip * SW;
        const int y0 = band * a.band_h, y1 = min(a.H, y0 + a.band_h);

        // row yy of the source (columns xs - 1 .. xs + SWP - 2) into slot yy mod R3_RING; rows outside the image or past the band's
        // lower halo fetch zeros through out-of-range offsets -- every call issues MAXP pieces per wave
        auto stage = [&](int yy) {
            const bool row_ok = yy >= 0 && yy < a.H && yy <= y1;
            const int slot = ((yy % R3_RING) + R3_RING) % R3_RING;
#pragma unroll
            for (int q = 0; q < MAXP; q++) {
                const int piece = q * 4 + wave;
                const bool pok = piece < PIECES;
                const int r = piece * PXR + lane / CPR;           // staged pixel
                const int pc = lane % CPR;                        // physical chunk
                const int pw = pc >> 1, sub = pc & 1;
                const int lw = RS == 128 ? (pw ^ ((r >> 1) & 3)) : (pw ^ ((r >> 2) & 1));
                const int ch = lw * 2 + sub;                      // logical chunk: channels 8 ch .. 8 ch + 7
                const int col = xs - 1 + r;
                const bool ok = pok && row_ok && col >= 0 && col < a.W && r < SW + 2 && ch * 8 < a.Cs;
                const uint32_t off = ok ? (uint32_t)((((size_t)(n * a.H + yy) * a.W + col) * a.lds_ + a.soff + ch * 8) * 2) : OOB;
                lds_dma16(rs_src, lds0 + (pok ? slot * ROWB + piece * 1024 : DUMP), off);
            }
        };

        __syncthreads();                                          // the previous unit's reads of the ring are done
        for (int yy = y0 - 1; yy <= y0 + R3_D; yy++) stage(yy);
        for (int y = y0; y < y1; y++) {
            // row y + 1 landed?  younger than its pieces: the D - 1 rows staged after it and the stores of the last min(t, D) rows
            const int t = y - y0;
            if (t == 0) wait_vm<(R3_D - 1) * MAXP>();
            else if (t == 1) wait_vm<(R3_D - 1) * MAXP + S>();
            else if (t == 2) wait_vm<(R3_D - 1) * MAXP + 2 * S>();
            else wait_vm<(R3_D - 1) * MAXP + R3_D * S>();
            __syncthreads();                                      // everyone's pieces; row y - 2's slot and the output row are free
            stage(y + 1 + R3_D);
            f32x4 acc[MBW][NBW];
#pragma unroll
            for (int mb = 0; mb < MBW; mb++)
#pragma unroll
                for (int nb = 0; nb < NBW; nb++) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int s0 = ((y - 1) % R3_RING + R3_RING) % R3_RING;
            const char* rowp[3];
#pragma unroll
            for (int ty = 0; ty < 3; ty++) rowp[ty] = sX + (s0 + ty >= R3_RING ? s0 + ty - R3_RING : s0 + ty) * ROWB;
            // the 9 x CC x NBW pixel operands of the row in one software pipeline: operand k + PF is read while the products of
            // operand k issue (a 16-byte LDS read returns after ~100 cycles, two products cover 32: with one read ahead -- the
            // compiler's own schedule -- a wave multiplied a third of the time, profiles/r5_ring3.txt)
            // pixel operand k = ((ty * 3 + tx) * CC + cc) * NBW + nb: lane (i, g) -> staged pixel 16 (NBW wn + nb) + i + tx,
            // channels 32 cc + 8 g .. + 7
            constexpr int NI = 9 * CC * NBW, PF = NI < 6 ? NI : 6;
            auto rd = [&](int k) {
                const int nb = k % NBW, cc = (k / NBW) % CC, tap = k / (NBW * CC), ty = tap / 3, tx = tap - 3 * ty;
                return *(const bf16x8*)(rowp[ty] + r3_off<RS>((wn * NBW + nb) * 16 + i + tx, cc * 4 + g));
            };
            bf16x8 xb[NI];
#pragma unroll
            for (int k = 0; k < PF; k++) xb[k] = rd(k);
            __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
            for (int k = 0; k < NI; k++) {
                if (k + PF < NI) xb[k + PF] = rd(k + PF);
                const int nb = k % NBW, cc = (k / NBW) % CC, tap = k / (NBW * CC);
#pragma unroll
                for (int mb = 0; mb < MBW; mb++)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[mb][tap][cc], xb[k], acc[mb][nb], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MBW, 0);
                if (k + PF < NI) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            // lane (i, g): pixel 16 (NBW wn + nb) + i, channels 16 (MBW wm + mb) + 4 g + r
#pragma unroll
            for (int mb = 0; mb < MBW; mb++) {
                const int chunk = (wm * MBW + mb) * 2 + (g >> 1);
#pragma unroll
                for (int nb = 0; nb < NBW; nb++) {
                    const int px = (wn * NBW + nb) * 16 + i;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) v[r] = apply_act(acc[mb][nb][r] + bv[mb][r], a.act, a.slope);
                    *(i32x2*)(sO + px * R3_OROW + ((chunk ^ (px & 7)) << 4) + (g & 1) * 8) = i32x2{(int)pack2bf(v[0], v[1]), (int)pack2bf(v[2], v[3])};
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < S; q++) {
                const int e = q * 256 + tid;
                const int px = e / NCH, chunk = e % NCH;
                const bool ok = e < ITEMS && xs + px < a.W && chunk * 8 < a.Cd;
                const i32x4 v = *(const i32x4*)(sO + (px & (SW - 1)) * R3_OROW + ((chunk ^ (px & 7)) << 4));
                if (a.stats && ok) {
                    float f[8];
                    unpack8(v, f);
#pragma unroll
                    for (int j = 0; j < 8; j++) { st_s[j] += f[j]; st_q[j] += f[j] * f[j]; }
                }
                const uint32_t off = ok ? (uint32_t)((((size_t)(n * a.H + y) * a.W + xs + px) * a.ldd + a.doff + chunk * 8) * 2) : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_dst, off, 0, 0);
            }
        }
        wait_vm<0>();
    }
    if (a.stats) {
        // one row of partial sums per workgroup: thread tid owns channels 8 (tid mod NCH) .. + 7 over its pixels
        __syncthreads();
        float* sR = (float*)smem;
#pragma unroll
        for (int j = 0; j < 8; j++) { sR[tid * 16 + j] = st_s[j]; sR[tid * 16 + 8 + j] = st_q[j]; }
        __syncthreads();
        if (tid < a.Cd) {
            const int chunk = tid >> 3, j = tid & 7;
            float ts = 0.f, tq = 0.f;
            for (int t = chunk; t < 256; t += NCH) { ts += sR[t * 16 + j]; tq += sR[t * 16 + 8 + j]; }
            st_stat(a.stats + ((size_t)blockIdx.x * 2 + 0) * a.Cstat + tid, ts, a.fin.tickets != nullptr);
            st_stat(a.stats + ((size_t)blockIdx.x * 2 + 1) * a.Cstat + tid, tq, a.fin.tickets != nullptr);
        }
        if (a.fin.tickets) gcc_igemm::stats_tail<256>(a.fin, a.stats, a.Cstat, blockIdx.x, (int*)(smem + 256 * 16 * 4), tid);
    }
}

struct Ring3Plan { int ok, rs, nbw, db, strips, bands, band_h, units, wgs; size_t lds; };
int ring3_cus() {
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
constexpr size_t R3_MIN_PIXELS = 8192;      // below this the launch is latency whatever the kernel: igemm_kernel's split-K routes keep it

Ring3Plan ring3_plan(const gcc_conv_t* c, int dgrad) {
    Ring3Plan p = {};
    if (!gcc_opt(GCC_OPT_IGEMM_THIN)) return p;
    if (c->KH != 3 || c->KW != 3 || c->stride != 1 || c->pad != 1) return p;
    const int Cs = dgrad ? c->Co : c->Ci, Cd = dgrad ? c->Ci : c->Co;
    if (Cs < 8 || Cs > 64 || (Cs & 7) || Cd < 8 || Cd > 64) return p;
    if (c->W < 16 || c->H < 2 || (size_t)c->N * c->H * c->W < R3_MIN_PIXELS) return p;
    const int lds_ = dgrad ? c->ldy : c->ldx, ldd = dgrad ? c->ldx : c->ldy;
    const size_t px = (size_t)c->N * c->H * c->W;
    if (px * lds_ * 2 >= OOB || px * ldd * 2 >= OOB) return p;
    p.rs = Cs > 32 ? 128 : 64;
    p.db = Cd > 32 ? 2 : 1;
    // strip width: 64 columns unless the last strip would be at most half used (W = 96: three strips of 32)
    const int rem = c->W % 64;
    p.nbw = (rem > 0 && rem <= 32) ? 1 : 2;
    const int sw = 32 * p.nbw;
    p.strips = cdiv(c->W, sw);
    const int pxr = 1024 / p.rs, swp = cdiv(sw + 2, pxr) * pxr;
    p.lds = (size_t)R3_RING * swp * p.rs + (size_t)sw * R3_OROW + 1024;
    if (p.lds < 256 * 16 * 4 + 16) p.lds = 256 * 16 * 4 + 16; // the statistics fold at the end of the launch + the ticket word of stats_tail
    // bands: the walk of a unit costs band_h + 2 + D row times; two workgroups per CU
    const int cols = c->N * p.strips, slots = 2 * ring3_cus();
    long best = -1;
    for (int nbands = 1; nbands <= c->H / 2 && nbands <= 128; nbands++) {
        const int bh = cdiv(c->H, nbands), nb = cdiv(c->H, bh);
        const long cost = (long)cdiv(cols * nb, slots) * (bh + 2 + R3_D);
        if (best < 0 || cost < best) { best = cost; p.band_h = bh; p.bands = nb; }
    }
    p.units = cols * p.bands;
    p.wgs = p.units < slots ? p.units : slots;
    p.ok = 1;
    return p;
}
bool ring3_epilogue_ok(const gcc_epilogue_t* ep) {
    if (!ep) return true;
    if (ep->y2) return false;
    if (ep->bn && !ep->stats_partial) return false;
    return true;
}
}  // namespace

// statistics rows a call on this route writes (0: not this route's geometry): one per workgroup
int gcc_internal_ring3_rows(const gcc_conv_t* c, int dgrad) {
    const Ring3Plan p = ring3_plan(c, dgrad);
    return p.ok ? p.wgs : 0;
}
bool gcc_internal_ring3_routed(const gcc_conv_t* c, int dgrad, const gcc_epilogue_t* ep) {
    return ring3_plan(c, dgrad).ok && ring3_epilogue_ok(ep);
}

// route of gcc_conv_fprop / gcc_conv_dgrad (conv_igemm.hip): GCC_ERR_UNSUPPORTED = not this route's geometry / epilogue
int gcc_internal_ring3(const gcc_conv_t* c, int dgrad, const void* src, const void* w, void* dst, const gcc_epilogue_t* ep, hipStream_t st) {
    const Ring3Plan p = ring3_plan(c, dgrad);
    if (!p.ok || !ring3_epilogue_ok(ep)) return GCC_ERR_UNSUPPORTED;
    Ring3Args a;
    a.src = (const bf16_t*)src; a.w = (const bf16_t*)w; a.dst = (bf16_t*)dst;
    a.bias = ep ? ep->bias : nullptr; a.stats = ep ? ep->stats_partial : nullptr;
    a.N = c->N; a.H = c->H; a.W = c->W;
    if (!dgrad) { a.lds_ = c->ldx; a.soff = c->xoff; a.ldd = c->ldy; a.doff = c->yoff; a.Cs = c->Ci; a.Cd = c->Co; }
    else { a.lds_ = c->ldy; a.soff = c->yoff; a.ldd = c->ldx; a.doff = c->xoff; a.Cs = c->Co; a.Cd = c->Ci; }
    a.Cs8 = ceil8(a.Cs); a.Cstat = a.Cd; a.flip = dgrad ? 1 : 0;
    a.act = ep ? ep->act : GCC_ACT_NONE; a.slope = ep ? ep->slope : 0.f;
    a.strips = p.strips; a.bands = p.bands; a.band_h = p.band_h; a.units = p.units;
    const size_t px = (size_t)c->N * c->H * c->W;
    a.src_bytes = (uint32_t)(px * a.lds_ * 2); a.dst_bytes = (uint32_t)(px * a.ldd * 2);
    // the BatchNorm behind the conv: folded by the last-arriving workgroups where the caller gave a tail workspace (the rows are one
    // per workgroup, one workgroup per row), by a gcc_bn_finalize launch otherwise -- same canonical order, same bits
    a.fin = TailFin{};
    const gcc_bn_t* bn = (ep && ep->bn && a.stats) ? ep->bn : nullptr;
    bool tail = false;
    if (bn && bn->finalize_in_launch && bn->tail_ws && (((uintptr_t)bn->tail_ws) & 15) == 0) {
        const size_t need = gcc_igemm::tail_ws_bytes(p.wgs, a.Cd);
        if (need <= gcc_igemm::TAIL_TICKET_BYTES + gcc_igemm::TAIL_GROUP_BYTES && bn->tail_ws_bytes >= gcc_igemm::TAIL_TICKET_BYTES + gcc_igemm::TAIL_GROUP_BYTES) {
            a.fin.tickets = (unsigned*)bn->tail_ws; a.fin.grp = (double*)((char*)bn->tail_ws + gcc_igemm::TAIL_TICKET_BYTES);
            a.fin.rows = p.wgs; a.fin.wgs_per_row = 1; a.fin.count = bn->count; a.fin.eps = bn->eps; a.fin.momentum = bn->momentum;
            a.fin.gamma = bn->gamma; a.fin.beta = bn->beta; a.fin.running_mean = bn->running_mean; a.fin.running_var = bn->running_var;
            a.fin.mean = bn->mean; a.fin.rstd = bn->rstd; a.fin.scale = bn->scale; a.fin.shift = bn->shift;
            tail = true;
        }
    }
#define GCC_R3_LAUNCH(RS_, NBW_, DB_)                                                                                            \
    do {                                                                                                                          \
        static std::once_flag once;                                                                                               \
        static hipError_t attr_err = hipSuccess;                                                                                  \
        std::call_once(once, [] {                                                                                                 \
            attr_err = hipFuncSetAttribute((const void*)ring3_kernel<RS_, NBW_, DB_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        });                                                                                                                       \
        if (attr_err != hipSuccess) return GCC_ERR_LAUNCH;                                                                        \
        hipLaunchKernelGGL((ring3_kernel<RS_, NBW_, DB_>), dim3(p.wgs), dim3(256), p.lds, st, a);                                 \
    } while (0)
    const int key = (p.rs == 128 ? 4 : 0) + (p.nbw == 2 ? 2 : 0) + (p.db == 2 ? 1 : 0);
    switch (key) {
        case 7: GCC_R3_LAUNCH(128, 2, 2); break;
        case 6: GCC_R3_LAUNCH(128, 2, 1); break;
        case 5: GCC_R3_LAUNCH(128, 1, 2); break;
        case 4: GCC_R3_LAUNCH(128, 1, 1); break;
        case 3: GCC_R3_LAUNCH(64, 2, 2); break;
        case 2: GCC_R3_LAUNCH(64, 2, 1); break;
        case 1: GCC_R3_LAUNCH(64, 1, 2); break;
        default: GCC_R3_LAUNCH(64, 1, 1); break;
    }
#undef GCC_R3_LAUNCH
    GCC_CHECK_LAUNCH();
    if (bn && !tail) {
        return gcc_bn_finalize(a.stats, p.wgs, a.Cd, bn->count, bn->gamma, bn->beta, bn->eps, bn->momentum, bn->running_mean, bn->running_var,
                               bn->mean, bn->rstd, bn->scale, bn->shift, (gcc_stream_t)st);
    }
    return GCC_OK;
}
