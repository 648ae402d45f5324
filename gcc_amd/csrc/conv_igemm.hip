// Implicit-GEMM gather convolution on MFMA (bf16 in, fp32 accumulate) for gfx950.
//
// One kernel serves Conv2d forward and backward-data (== ConvTranspose2d forward): for every output
// pixel it gathers TA x TB taps of Ct channels from an NHWC source and contracts them with a weight
// panel whose K index is (tap, channel) -- K is contiguous within a tap for both operands, so both
// LDS tiles are [row][64 k] images filled by 16-byte buffer loads (hardware range check supplies
// the zero padding) and read back with ds_read_b128 through an XOR swizzle.
//
// Tile: 128 output pixels x BC output channels x 64 k per step, 256 threads = 4 waves.
// MFMA operand roles are swapped (A = weights, B = pixels) so that each lane ends up holding
// 4 consecutive output channels of one pixel; the epilogue (bias, activation, bf16 rounding,
// optional BatchNorm partial sums) goes through LDS and leaves as coalesced 16-byte NHWC stores.
#include <stdlib.h>
#include <mutex>
#include "common.hpp"
#include "igemm_common.hpp"

// named (not anonymous) namespace: hipcc fails to emit the host stub of a kernel template with internal
// linkage whose body holds lambdas inside an `if constexpr` branch
// GCC_IGEMM_ROT=1: the k loop rotated by half a step (see the loop).  Measured and NOT the default (profiles/r03_l_clock_probe.txt,
// same box, in-kernel clock stamps): the 256x256 loop takes 4 % fewer cycles per k-step (2828 -> 2720) and a launch that is not
// power-limited gains them (half-chip L4 dgrad 315.7 -> 304.0 us, zero-filled L4 fprop 166.5 -> 160.6 us), but on random data at
// full chip the clock falls from 2.07 to 2.02 GHz and the launch time does not move (191.5 -> 191.0 us), the short-K and the
// 128-pixel-tile launches get slower (L2 fprop 62.4 -> 64.0 us; the kernel spills 94 VGPRs), and the step loses 0.4 %
// (897.7 against 901.6 images/s, igemm 769 against 789 TFLOP/s).
#ifndef GCC_IGEMM_ROT
#define GCC_IGEMM_ROT 0
#endif

namespace gcc_igemm {

// GLDS = true : tiles are staged global -> LDS directly (buffer_load ... lds, 1 KiB per wave-instruction,
//               XOR swizzle applied on the per-lane SOURCE address, zero fill by the descriptor range
//               check), one barrier per k-step: no staging VGPRs, no ds_write traffic.
// GLDS = false: register-staged variant (kept for A/B measurement).
#ifdef GCC_CLOCK_PROBE
// Diagnostic build only (scratch/probe_clock.py; never part of libgcc_hip.so): the shader clock the chip holds inside the
// main loop = d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, 'DVFS give-back' item 6), one stamp pair per
// workgroup around the k loop.
__device__ unsigned long long g_clock_probe[4096][8];
#endif

// NS: LDS stages of the k loop.  2: the round-1 loop (one k-step in flight behind the one being multiplied).  3 (round 4; 128-pixel
// tiles of 32 / 64 columns on the uniform-tap path): two k-steps in flight.  The U-Net's mid layers run these tiles on 256-512
// workgroups with 8-32 k-steps of 8-16 MFMAs per wave: a k-step was ~0.85 us of which ~0.06 us is matrix work -- the step waits
// for the loads it issued one step earlier (profiles/r4f_unet_student_chain.txt: d3 27 us for 32 steps).
template <int BP, int BC, bool GLDS, bool UT, int NS = 2>
__global__ __launch_bounds__((BP / 32) * 64) void igemm_kernel(const IgemmParams p) {
    using C = Cfg<BP, BC>;
    static_assert(GLDS || BP == 128, "register staging is only kept for the 128-pixel tile");
    static_assert(NS == 2 || (GLDS && UT && BP == 128 && (BC == 32 || BC == 64)), "deeper loops: uniform-tap 128 x {32, 64} tiles");
    constexpr int NT = C::NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                          // pixels  [NS][BP][128 B]
    char* sW = smem + NS * BP * BK * 2;       // weights [NS][BC][128 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: LDS-DMA bases stay scalar
    const int wc = wave % C::WC;
    const int wp = wave / C::WC;
#ifdef GCC_CLOCK_PROBE
    const unsigned long long pe0 = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- per-phase geometry -----------------------------------------------------------------
    int py = 0, px = 0, Hg, Wg, sy, TA, TB, dy0, dx0, dstep, kh0, kw0, kstep, ostr;
    if (!p.dgrad) {
        Hg = p.Hd; Wg = p.Wd; sy = p.stride; TA = p.KH; TB = p.KW;
        dy0 = -p.pad; dx0 = -p.pad; dstep = 1; kh0 = 0; kw0 = 0; kstep = 1; ostr = 1;
    } else {
        const int s = p.stride;
        py = blockIdx.z / s; px = blockIdx.z % s;
        kh0 = (py + p.pad) % s; kw0 = (px + p.pad) % s;
        TA = (p.KH - kh0 + s - 1) / s; TB = (p.KW - kw0 + s - 1) / s;
        dy0 = (py + p.pad - kh0) / s; dx0 = (px + p.pad - kw0) / s;
        dstep = -1; kstep = s; sy = 1; ostr = s;
        Hg = (p.Hd - py + s - 1) / s; Wg = (p.Wd - px + s - 1) / s;
    }
    const int M = p.N * Hg * Wg;
    const int Ktot = TA * TB * p.Ct;
    const int nk_all = (Ktot + BK - 1) / BK;
    const int ks_idx = p.ksplit > 1 ? blockIdx.y : 0;
    const int kbeg = ks_idx * p.kper;
    const int nk = p.ksplit > 1 ? min(p.kper, nk_all - kbeg) : nk_all;

    const int nwg = gridDim.x;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int mt = tile / p.ntiles;
    const int nt = tile % p.ntiles;
    const int m0 = mt * BP;
    const int n0 = nt * BC;
    if (m0 >= M) {         // smaller phase (odd sizes): uniform exit, no barrier reached yet
        // pair split: both K halves of an empty tile land here; only half 0 counts (a full tile reaches the epilogue once,
        // through its second-arriving half, and make_tail's wgs_per_row = ntiles assumes one arrival per tile)
        if (p.stats && (!p.pair || blockIdx.y == 0)) {
            const int trow = blockIdx.z * p.mtiles_max + mt;
            if (tid < BC && n0 + tid < p.Cout) {
                st_stat(p.stats + ((size_t)trow * 2 + 0) * p.Cout + n0 + tid, 0.f, p.fin.tickets != nullptr);
                st_stat(p.stats + ((size_t)trow * 2 + 1) * p.Cout + n0 + tid, 0.f, p.fin.tickets != nullptr);
            }
            if (p.fin.tickets) stats_tail<NT>(p.fin, p.stats, p.Cout, trow, (int*)smem, tid);     // its row counts like any other
        }
        return;
    }

    const int bidx = p.ksplit > 1 ? 0 : blockIdx.y;
    const bf16_t* srcp = p.src + (size_t)bidx * p.src_bstride;
    const bf16_t* wgtp = p.wgt + (size_t)bidx * p.wgt_bstride;
    bf16_t* dstp = p.dst + (size_t)bidx * p.dst_bstride;
    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)srcp, 0, p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wgt = __builtin_amdgcn_make_buffer_rsrc((void*)wgtp, 0, p.wgt_bytes, 0x00020000);

    // ---- per-thread gather rows (4 pixel rows, fixed 16-B chunk column) ----------------------
    // register staging: row = (tid>>3) + 32*i, chunk = tid&7 (swizzled on the LDS write)
    // LDS-DMA        : wave-instruction q = wave*4+i covers rows 8q..8q+7; lane -> row 8q + (lane>>3),
    //                  physical chunk lane&7, which must hold logical chunk (lane&7) ^ (row&7)
    const int chunk = GLDS ? ((lane & 7) ^ (lane >> 3)) : (tid & 7);
    constexpr int AI = C::AI;
    static_assert(AI <= 8 && (GLDS || AI == 4), "pixel staging instructions per wave");
    int a_off[8], a_iy[8], a_ix[8];      // byte offset of the row's (iy0, ix0) pixel (+ chunk), and iy0 / ix0 (first AI used)
#pragma unroll
    for (int i = 0; i < AI; i++) {
        const int rloc = GLDS ? ((wave * AI + i) * 8 + (lane >> 3)) : ((tid >> 3) + 32 * i);
        const int m = m0 + rloc;
        if (m < M) {
            const int n = m / (Hg * Wg);
            const int r = m - n * (Hg * Wg);
            const int oy = r / Wg;
            const int ox = r - oy * Wg;
            a_iy[i] = oy * sy + dy0;
            a_ix[i] = ox * sy + dx0;
            a_off[i] = (((n * p.Hs + a_iy[i]) * p.Ws + a_ix[i]) * p.lds_ + p.soff) * 2 + (UT ? chunk * 16 : 0);
        } else {
            a_off[i] = 0; a_iy[i] = -(1 << 28); a_ix[i] = 0;   // always out of range -> zeros
        }
    }
    // weight rows of this thread
    constexpr int WPW = C::WPW;                        // LDS-DMA weight instructions per wave (0: waves < WI issue one)
    constexpr int W_N = GLDS ? (WPW > 0 ? WPW : 1) : C::W_CHUNKS;
    int w_off[W_N];
    bool w_ok[W_N];
#pragma unroll
    for (int i = 0; i < W_N; i++) {
        const int wr = GLDS ? ((wave * (WPW > 0 ? WPW : 1) + i) * 8 + (lane >> 3)) : ((tid >> 3) + 32 * i);
        const int row = n0 + wr;
        w_ok[i] = row < p.Cout && (GLDS || BC >= 32 || wr < BC);
        w_off[i] = row * p.ldw * 2 + (UT ? chunk * 16 : 0);
    }

    // K walk without divisions: (ta, tb) = tap coordinates, cc = channel offset inside the tap.
    // UT (Ct % 64 == 0): one tap per k-step, wave-uniform state (scalar registers);
    // otherwise every thread walks the tap of its own 8-channel chunk.
    const int tap_row_bytes = p.Ws * p.lds_ * 2 * dstep;   // bytes per +1 in `ta`
    const int tap_col_bytes = p.lds_ * 2 * dstep;          // bytes per +1 in `tb`
    int ta, tb, cc;
    {
        const int k0 = kbeg * BK + (UT ? 0 : chunk * 8);
        const int tap0 = k0 / p.Ct;
        cc = k0 - tap0 * p.Ct;
        ta = tap0 / TB;
        tb = tap0 - ta * TB;
    }

    i32x4 ra[GLDS ? 1 : 4];
    i32x4 rw[GLDS ? 1 : C::W_CHUNKS];

    // issue the global loads of the next k-step (in order); GLDS: straight into LDS stage `stage`
    auto issue_loads = [&](int stage) {
        const bool kval = ta < TA;
        const int dyo = ta * dstep, dxo = tb * dstep;
        const int pix_off = ta * tap_row_bytes + tb * tap_col_bytes + cc * 2;
        const int wt_off = (((kh0 + ta * kstep) * p.KW + (kw0 + tb * kstep)) * p.Ct + cc) * 2;
#pragma unroll
        for (int i = 0; i < AI; i++) {
            const bool ok = kval && (unsigned)(a_iy[i] + dyo) < (unsigned)p.Hs && (unsigned)(a_ix[i] + dxo) < (unsigned)p.Ws;
            const uint32_t off = ok ? (uint32_t)(a_off[i] + pix_off) : OOB;
            if constexpr (GLDS) {
                char* dst = sA + stage * (BP * BK * 2) + (wave * AI + i) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, LDS_PTR(void, dst), 16, off, 0, 0, 0);
            } else {
                ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, off, 0, 0);
            }
        }
        if constexpr (GLDS) {
            if (WPW > 0 || wave < C::WI) {       // wave-uniform
#pragma unroll
                for (int i = 0; i < W_N; i++) {
                    const uint32_t off = (kval && w_ok[i]) ? (uint32_t)(w_off[i] + wt_off) : OOB;
                    char* dst = sW + stage * (BC * BK * 2) + (wave * (WPW > 0 ? WPW : 1) + i) * 1024;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, LDS_PTR(void, dst), 16, off, 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < C::W_CHUNKS; i++) {
                const uint32_t off = (kval && w_ok[i]) ? (uint32_t)(w_off[i] + wt_off) : OOB;
                rw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_wgt, off, 0, 0);
            }
        }
        // advance to the next k-step
        cc += BK;
        if constexpr (UT) {
            if (cc == p.Ct) { cc = 0; if (++tb == TB) { tb = 0; ++ta; } }
        } else {
            while (cc >= p.Ct) { cc -= p.Ct; if (++tb == TB) { tb = 0; ++ta; } }
        }
    };

    auto write_lds = [&](int stage) {
        if constexpr (!GLDS) {
            char* a = sA + stage * (BP * BK * 2);
            char* w = sW + stage * (BC * BK * 2);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int row = (tid >> 3) + 32 * i;
                *(i32x4*)(a + row * 128 + ((chunk ^ (row & 7)) << 4)) = ra[i];
            }
#pragma unroll
            for (int i = 0; i < C::W_CHUNKS; i++) {
                const int row = (tid >> 3) + 32 * i;
                if (BC >= 32 || row < BC) *(i32x4*)(w + row * 128 + ((chunk ^ (row & 7)) << 4)) = rw[i];
            }
        }
    };

    f32x4 acc[C::CB][C::PB];
#pragma unroll
    for (int i = 0; i < C::CB; i++)
#pragma unroll
        for (int j = 0; j < C::PB; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int lr = lane & 15;
    const int lq = lane >> 4;
    auto compute = [&](int cur) {
        const char* a = sA + cur * (BP * BK * 2);
        const char* w = sW + cur * (BC * BK * 2);
        // Both k-slices (2 x 32) of the stage are held in registers: the fragment reads of slice 1 are issued while
        // the MFMAs of slice 0 run (the compiler's own order reads two fragments, waits, issues four MFMAs -- LDS
        // latency exposed at every group).  sched_group_barrier pins the interleave: all reads of slice 0, then one
        // read of slice 1 per MFMA_PER_READ MFMAs, then the rest.
        bf16x8 fw[2][C::CB], fa[2][C::PB];
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
#pragma unroll
            for (int i = 0; i < C::CB; i++) {
                const int row = wc * C::TC + i * 16 + lr;
                fw[ks][i] = *(const bf16x8*)(w + row * 128 + (((ks * 4 + lq) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < C::PB; j++) {
                const int row = wp * C::TP + j * 16 + lr;
                fa[ks][j] = *(const bf16x8*)(a + row * 128 + (((ks * 4 + lq) ^ (row & 7)) << 4));
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int i = 0; i < C::CB; i++)
#pragma unroll
                for (int j = 0; j < C::PB; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ks][i], fa[ks][j], acc[i][j], 0, 0, 0);
        constexpr int READS = C::CB + C::PB;                 // fragment reads per k-slice
        constexpr int MFMAS = C::CB * C::PB;                 // MFMAs per k-slice
        constexpr int MPR = MFMAS / READS > 0 ? MFMAS / READS : 1;
        __builtin_amdgcn_sched_group_barrier(0x100, READS, 0);          // slice 0 fragments
#pragma unroll
        for (int r = 0; r < READS; r++) {
            __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);        // MFMAs of slice 0 ...
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);          // ... covering one read of slice 1
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * MFMAS - READS * MPR, 0);
    };

    if constexpr (GLDS && UT) {
        // Uniform-tap fast path: a tap spans Ct/64 consecutive k-steps, so the per-row source offsets
        // (bounds checks included) are computed once per tap and then just advance by 128 bytes per
        // k-step; the loop body is 8 LDS-DMA + 8 adds + 16 ds_read + 32 MFMA per wave.
        static_assert(C::WN_GLDS <= 8, "weight staging instructions per wave");
        uint32_t cur_a[8], cur_w[8];   // literal bound on purpose: with a template-dependent bound hipcc (ROCm 7.2)
                                       // silently drops the host stub of this instantiation
        int left;                                  // k-steps left in the current tap (scalar)
        auto load_tap = [&]() {
            const bool kval = ta < TA;
            const int dyo = ta * dstep, dxo = tb * dstep;
            const int pix_off = ta * tap_row_bytes + tb * tap_col_bytes + cc * 2;
            const int wt_off = (((kh0 + ta * kstep) * p.KW + (kw0 + tb * kstep)) * p.Ct + cc) * 2;
#pragma unroll
            for (int i = 0; i < AI; i++) {
                const bool ok = kval && (unsigned)(a_iy[i] + dyo) < (unsigned)p.Hs && (unsigned)(a_ix[i] + dxo) < (unsigned)p.Ws;
                cur_a[i] = ok ? (uint32_t)(a_off[i] + pix_off) : OOB;
            }
#pragma unroll
            for (int i = 0; i < W_N; i++) cur_w[i] = (kval && w_ok[i]) ? (uint32_t)(w_off[i] + wt_off) : OOB;
            left = (p.Ct - cc) / BK;
        };
        auto issue = [&](int stage, bool pixels = true) {
#pragma unroll
            for (int i = 0; i < AI; i++) {
                char* dst = sA + stage * (BP * BK * 2) + (wave * AI + i) * 1024;
                if (pixels) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, LDS_PTR(void, dst), 16, cur_a[i], 0, 0, 0);
                cur_a[i] += BK * 2;
            }
            if (WPW > 0 || wave < C::WI) {
#pragma unroll
                for (int i = 0; i < W_N; i++) {
                    char* dst = sW + stage * (BC * BK * 2) + (wave * (WPW > 0 ? WPW : 1) + i) * 1024;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, LDS_PTR(void, dst), 16, cur_w[i], 0, 0, 0);
                    cur_w[i] += BK * 2;
                }
            }
        };
        auto next_step = [&]() {
            if (--left == 0) {                      // wave-uniform, once per tap
                cc = 0;
                if (++tb == TB) { tb = 0; ++ta; }
                load_tap();
            }
        };
#ifdef GCC_CLOCK_PROBE
        const unsigned long long pt0 = __builtin_amdgcn_s_memtime(), pr0 = __builtin_amdgcn_s_memrealtime();
#endif
        if constexpr (NS > 2) {
            // NS - 1 k-steps in flight.  LDS-DMA from inline assembly (lds_dma16): through the builtin hipcc would put
            // s_waitcnt vmcnt(0) in front of the fragment reads of the stage being multiplied (it counts a pending DMA as an LDS
            // write it cannot prove disjoint) and drain the steps behind it; here the counted wait + barrier below order the data.
            constexpr int PER_STEP = AI + W_N;                     // DMA instructions per wave and k-step (uniform for BC 32 / 64)
            static_assert(C::WPW > 0, "every wave stages weights");
            const i32x4 rsv_src = make_rsrc(srcp, p.src_bytes), rsv_wgt = make_rsrc(wgtp, p.wgt_bytes);
            const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
            auto issue_n = [&](int stage) {
#pragma unroll
                for (int i = 0; i < AI; i++) {
                    lds_dma16(rsv_src, lds0 + stage * (BP * BK * 2) + (wave * AI + i) * 1024, cur_a[i]);
                    cur_a[i] += BK * 2;
                }
#pragma unroll
                for (int i = 0; i < W_N; i++) {
                    lds_dma16(rsv_wgt, lds0 + NS * BP * BK * 2 + stage * (BC * BK * 2) + (wave * WPW + i) * 1024, cur_w[i]);
                    cur_w[i] += BK * 2;
                }
            };
            load_tap();
#pragma unroll
            for (int st = 0; st < NS - 1; st++) { issue_n(st); next_step(); }
            int stage = 0, fill = NS - 1;                          // stage of step kt; stage the next issue goes to
            for (int kt = 0; kt < nk; kt++) {
                // step kt has landed once at most the NS - 2 younger steps are outstanding (vmcnt counts in issue order)
                if constexpr (NS == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STEP) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_STEP) : "memory");
                __syncthreads();                                     // ... for every wave, and everyone left the stage of step kt - 1
                issue_n(fill);                                       // past the end of K: out-of-range offsets, zero fill, unused
                next_step();
                compute(stage);
                stage = stage + 1 == NS ? 0 : stage + 1;
                fill = fill + 1 == NS ? 0 : fill + 1;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else {
        load_tap();
        issue(0);
        next_step();
#if GCC_IGEMM_ROT
        // The k loop rotated by half a step: the barrier sits between the two 32-deep k-slices of a stage, so that the
        // fragment reads that follow it (slice 0 of the NEXT stage) run under the MFMAs of slice 1 instead of in front of
        // the step's first MFMA -- with the barrier at the top of the step both waves of a SIMD wait for their first twelve
        // ds_read_b128 at the same time and the matrix pipe idles (in-kernel clock stamps, profiles/r03_l_*: 72 % busy).
        //   A: MFMAs of slice 0 | reads of slice 1 (stage s)         -> nobody reads stage s any more
        //   B: my DMA of stage s^1 landed; barrier                     -> everybody's has, stage s is free
        //   C: DMA of step kt+2 into stage s
        //   D: MFMAs of slice 1 | reads of slice 0 of stage s^1
        // A DMA issued at C is waited for at B of the next step: one full step in flight, as before.
        bf16x8 f0w[C::CB], f0a[C::PB], f1w[C::CB], f1a[C::PB];
        auto read_slice = [&](int stage, int ks, bf16x8 (&fw)[C::CB], bf16x8 (&fa)[C::PB]) {
            const char* a = sA + stage * (BP * BK * 2);
            const char* w = sW + stage * (BC * BK * 2);
#pragma unroll
            for (int i = 0; i < C::CB; i++) {
                const int row = wc * C::TC + i * 16 + lr;
                fw[i] = *(const bf16x8*)(w + row * 128 + (((ks * 4 + lq) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < C::PB; j++) {
                const int row = wp * C::TP + j * 16 + lr;
                fa[j] = *(const bf16x8*)(a + row * 128 + (((ks * 4 + lq) ^ (row & 7)) << 4));
            }
        };
        auto mfma_slice = [&](const bf16x8 (&fw)[C::CB], const bf16x8 (&fa)[C::PB]) {
#pragma unroll
            for (int i = 0; i < C::CB; i++)
#pragma unroll
                for (int j = 0; j < C::PB; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        };
        constexpr int R_READS = C::CB + C::PB, R_MFMAS = C::CB * C::PB;
        constexpr int R_MPR = R_MFMAS / R_READS > 0 ? R_MFMAS / R_READS : 1;
        auto pin = [&]() {                          // one fragment read per R_MPR MFMAs, then the remaining MFMAs
#pragma unroll
            for (int r = 0; r < R_READS; r++) {
                __builtin_amdgcn_sched_group_barrier(0x008, R_MPR, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, R_MFMAS - R_READS * R_MPR > 0 ? R_MFMAS - R_READS * R_MPR : 0, 0);
        };
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        issue(1);                                   // past the end of K the offsets are out of range: zero fill, unused
        next_step();
        read_slice(0, 0, f0w, f0a);
        for (int kt = 0; kt < nk; kt++) {
            const int cur = kt & 1;
            read_slice(cur, 1, f1w, f1a);
            mfma_slice(f0w, f0a);
            pin();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            issue(cur);
            next_step();                            // (rare) tap change: its VALU work hides under the MFMAs below
            __builtin_amdgcn_sched_barrier(0);
            read_slice(cur ^ 1, 0, f0w, f0a);       // the last step reads a stage nobody uses
            mfma_slice(f1w, f1a);
            pin();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#else
        for (int kt = 0; kt < nk; kt++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (!(GCC_DIAG(p.debug) & 2)) {
                issue((kt + 1) & 1, !(GCC_DIAG(p.debug) & 8) || (kt & 3) == 0);    // past the end of K the offsets are out of range: zero fill, unused
                if (!(GCC_DIAG(p.debug) & 4)) next_step();    // (rare) tap change: its VALU work hides under the MFMAs below
                else {
#pragma unroll
                    for (int i = 0; i < AI; i++) cur_a[i] -= BK * 2;
#pragma unroll
                    for (int i = 0; i < W_N; i++) cur_w[i] -= BK * 2;
                }
            }
            compute(kt & 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#endif
        }
#ifdef GCC_CLOCK_PROBE
        {
            const unsigned long long pt1 = __builtin_amdgcn_s_memtime(), pr1 = __builtin_amdgcn_s_memrealtime();
            const unsigned b = blockIdx.z * gridDim.x + blockIdx.x;
            if (tid == 0 && b < 4096) {
                g_clock_probe[b][0] = pt1 - pt0; g_clock_probe[b][1] = pr1 - pr0; g_clock_probe[b][2] = (unsigned long long)nk;
                g_clock_probe[b][3] = 1;
                g_clock_probe[b][4] = pe0; g_clock_probe[b][5] = pr0; g_clock_probe[b][6] = pr1;
            }
        }
#endif
    } else if constexpr (GLDS) {
        // one barrier per k-step: [tile kt landed for every wave AND everyone left tile kt-1] ->
        // issue tile kt+1 into the buffer tile kt-1 occupied -> compute tile kt while it flies
#ifdef GCC_CLOCK_PROBE
        const unsigned long long qr0 = __builtin_amdgcn_s_memrealtime();
#endif
        issue_loads(0);
        for (int kt = 0; kt < nk; kt++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < nk) issue_loads((kt + 1) & 1);
            compute(kt & 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#ifdef GCC_CLOCK_PROBE
        {
            const unsigned long long qr1 = __builtin_amdgcn_s_memrealtime();
            const unsigned b = blockIdx.z * gridDim.x + blockIdx.x;
            if (tid == 0 && b < 4096) {
                g_clock_probe[b][0] = 0; g_clock_probe[b][1] = qr1 - qr0; g_clock_probe[b][2] = (unsigned long long)nk;
                g_clock_probe[b][3] = 1;
                g_clock_probe[b][4] = pe0; g_clock_probe[b][5] = qr0; g_clock_probe[b][6] = qr1;
            }
        }
#endif
    } else {
        // registers hold tile t+1 while LDS[t&1] is consumed
        issue_loads(0);
        write_lds(0);
        if (nk > 1) issue_loads(0);
        __syncthreads();
        for (int kt = 0; kt < nk; kt++) {
            const int cur = kt & 1;
            if (kt + 1 < nk) {
                write_lds(cur ^ 1);                 // tile kt+1 (its loads were issued one step ago)
                if (kt + 2 < nk) issue_loads(0);
            }
            compute(cur);
            __syncthreads();
        }
    }

    if constexpr (BP == 256 && BC == 256) {
        if (p.pair) {
            // Two workgroups own this tile, one per K half (in-launch split-K hand-off, cdna_hip_programming.md Guideline 16,
            // recipe R1): write-through (sc1) slab stores, every storing wave drains, one lane publishes the flag; the other
            // half polls that one word, one agent-scope acquire, sc1 slab loads.  a + b == b + a: the result does not
            // depend on which half arrives first.
            typedef __attribute__((address_space(1))) unsigned int gu32;
            const int tile_id = (int)(blockIdx.z * gridDim.x + blockIdx.x);
            gu32* fl = (gu32*)(p.pair_flags + 2 * tile_id);
            int* sh = (int*)smem;                           // the loop's LDS is free: every wave passed its last barrier
            if (tid == 0) sh[0] = (int)__hip_atomic_fetch_add(fl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const int ticket = sh[0];
            __syncthreads();
            const __amdgpu_buffer_rsrc_t rs_slab =
                __builtin_amdgcn_make_buffer_rsrc((void*)(p.pair_slab + (size_t)tile_id * (BP * BC)), 0, BP * BC * 4, 0x00020000);
            if (ticket == 0) {
#pragma unroll
                for (int i = 0; i < C::CB; i++)
#pragma unroll
                    for (int j = 0; j < C::PB; j++)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, acc[i][j]), rs_slab,
                                                               ((i * C::PB + j) * NT + tid) * 16, 0, 16);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(fl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            if (tid == 0) {
                while (__hip_atomic_load(fl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(4);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < C::CB; i++)
#pragma unroll
                for (int j = 0; j < C::PB; j++) {
                    const f32x4 o = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_slab, ((i * C::PB + j) * NT + tid) * 16, 0, 16));
                    acc[i][j] += o;
                }
        }
    }
    igemm_epilogue<C, BP, BC>(p, acc, smem, tid, lr, lq, wc, wp, m0, n0, M, Hg, Wg, ostr, py, px, mt, ks_idx, dstp);
#ifdef GCC_CLOCK_PROBE
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned b = blockIdx.z * gridDim.x + blockIdx.x;
        if (tid == 0 && b < 4096) g_clock_probe[b][7] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}


// finish a split-K launch: sum the K slices, bias + activation, bf16 NHWC store (same pixel map)
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const IgemmParams p) {
    int py = 0, px = 0, Hg, Wg, ostr = 1;
    if (!p.dgrad) { Hg = p.Hd; Wg = p.Wd; }
    else {
        const int s = p.stride;
        py = blockIdx.z / s; px = blockIdx.z % s; ostr = s;
        Hg = (p.Hd - py + s - 1) / s; Wg = (p.Wd - px + s - 1) / s;
    }
    const int M = p.N * Hg * Wg;
    const int CH = ceil8(p.Cout) / 8;
    const size_t total = (size_t)M * CH;
    const float* base = p.partial + (size_t)blockIdx.z * p.ksplit * p.rows_max * p.Cpad;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
        const int m = (int)(q / CH);
        const int c0 = (int)(q - (size_t)m * CH) * 8;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const float* r0 = base + (size_t)m * p.Cpad + c0;
        const size_t sstride = (size_t)p.rows_max * p.Cpad;
        int sl = 0;
        for (; sl + 3 < p.ksplit; sl += 4) {            // four slices (8 loads) in flight
            f32x4 a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { a[u] = *(const f32x4*)(r0 + (sl + u) * sstride); b[u] = *(const f32x4*)(r0 + (sl + u) * sstride + 4); }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                v[0] += a[u][0]; v[1] += a[u][1]; v[2] += a[u][2]; v[3] += a[u][3];
                v[4] += b[u][0]; v[5] += b[u][1]; v[6] += b[u][2]; v[7] += b[u][3];
            }
        }
        for (; sl < p.ksplit; sl++) {
            const f32x4 a = *(const f32x4*)(r0 + sl * sstride), b = *(const f32x4*)(r0 + sl * sstride + 4);
            v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3];
            v[4] += b[0]; v[5] += b[1]; v[6] += b[2]; v[7] += b[3];
        }
        if (p.bias) {
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] += c0 + j < p.Cout ? p.bias[c0 + j] : 0.f;
        }
        apply_act8(v, v, p.act, p.slope);
        if (c0 + 8 > p.Cout) {
#pragma unroll
            for (int j = 0; j < 8; j++) if (c0 + j >= p.Cout) v[j] = 0.f;
        }
        const int n = m / (Hg * Wg);
        const int r = m - n * (Hg * Wg);
        const int oy = r / Wg, ox = r - oy * Wg;
        const size_t o = ((size_t)(n * p.Hd + oy * ostr + py) * p.Wd + (ox * ostr + px)) * p.ldd + p.doff + c0;
        *(i32x4*)(p.dst + o) = pack8(v);
    }
}

// finish a split-K launch whose output feeds a BatchNorm, in ONE kernel on the whole chip (round 4): a workgroup owns `rpw`
// consecutive rows of a phase x all channels -- folds the K slices (+ bias, activation), stores the bf16-rounded rows, writes
// the statistic row of ITS rows and, with p.fin, takes part in the finalize by the last arrivers (stats_tail).  Replaces
// splitk_epilogue + channel_stats (+ bn_finalize), and splitk_bn_act_kernel's C / 8 workgroups that walked every row twice
// (14-56 us on the U-Net's <= 16x16 layers, profiles/r3z_unet_student_chain.txt).  Statistic rows: phases x wpp.
struct FoldStatsArgs { IgemmParams p; int rpw, wpp, CHP, sh; };
__global__ __launch_bounds__(256) void splitk_fold_stats_kernel(const FoldStatsArgs a) {
    __shared__ float red[256 * 16];
    __shared__ int ticket;
    const IgemmParams& p = a.p;
    int py = 0, px = 0, Hg, Wg, ostr = 1;
    if (!p.dgrad) { Hg = p.Hd; Wg = p.Wd; }
    else {
        const int s = p.stride;
        py = blockIdx.z / s; px = blockIdx.z % s; ostr = s;
        Hg = (p.Hd - py + s - 1) / s; Wg = (p.Wd - px + s - 1) / s;
    }
    const int M = p.N * Hg * Wg;
    const int CH = ceil8(p.Cout) / 8;
    const int tid = threadIdx.x;
    const int chl = tid & (a.CHP - 1), pl = tid >> a.sh, PL = 256 >> a.sh;
    const int m_lo = blockIdx.x * a.rpw, m_hi = min(m_lo + a.rpw, M);
    const float* base = p.partial + (size_t)blockIdx.z * p.ksplit * p.rows_max * p.Cpad;
    const size_t sstride = (size_t)p.rows_max * p.Cpad;
    float s[8], ss[8];
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = ss[j] = 0.f;
    if (chl < CH) {
        const int c0 = chl * 8;
        for (int m = m_lo + pl; m < m_hi; m += PL) {
            float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const float* r0 = base + (size_t)m * p.Cpad + c0;
            int sl = 0;
            for (; sl + 3 < p.ksplit; sl += 4) {            // four slices (8 loads) in flight, added in slice order
                f32x4 u[4], w[4];
#pragma unroll
                for (int q = 0; q < 4; q++) { u[q] = *(const f32x4*)(r0 + (sl + q) * sstride); w[q] = *(const f32x4*)(r0 + (sl + q) * sstride + 4); }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    v[0] += u[q][0]; v[1] += u[q][1]; v[2] += u[q][2]; v[3] += u[q][3];
                    v[4] += w[q][0]; v[5] += w[q][1]; v[6] += w[q][2]; v[7] += w[q][3];
                }
            }
            for (; sl < p.ksplit; sl++) {
                const f32x4 u = *(const f32x4*)(r0 + sl * sstride), w = *(const f32x4*)(r0 + sl * sstride + 4);
                v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
                v[4] += w[0]; v[5] += w[1]; v[6] += w[2]; v[7] += w[3];
            }
            if (p.bias) {
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] += c0 + j < p.Cout ? p.bias[c0 + j] : 0.f;
            }
            apply_act8(v, v, p.act, p.slope);
#pragma unroll
            for (int j = 0; j < 8; j++) if (c0 + j >= p.Cout) v[j] = 0.f;          // pad channels stay exact zeros
            const i32x4 pk = pack8(v);
            float r[8];
            unpack8(pk, r);
#pragma unroll
            for (int j = 0; j < 8; j++) { s[j] += r[j]; ss[j] += r[j] * r[j]; }
            const int n = m / (Hg * Wg);
            const int rr = m - n * (Hg * Wg);
            const int oy = rr / Wg, ox = rr - oy * Wg;
            const size_t o = ((size_t)(n * p.Hd + oy * ostr + py) * p.Wd + (ox * ostr + px)) * p.ldd + p.doff + c0;
            *(i32x4*)(p.dst + o) = pk;
        }
    }
    if (!p.stats) return;
#pragma unroll
    for (int j = 0; j < 8; j++) { red[tid * 16 + j] = s[j]; red[tid * 16 + 8 + j] = ss[j]; }
    __syncthreads();
    const int trow = blockIdx.z * a.wpp + blockIdx.x;
    const bool sc1 = p.fin.tickets != nullptr;
    for (int idx = tid; idx < 2 * p.Cout; idx += 256) {      // row lanes folded in ascending order
        const int w = idx >= p.Cout ? 1 : 0, c = idx - w * p.Cout;
        float t = 0.f;
        for (int q = 0; q < PL; q++) t += red[((q << a.sh) + (c >> 3)) * 16 + w * 8 + (c & 7)];
        st_stat(p.stats + ((size_t)trow * 2 + w) * p.Cout + c, t, sc1);
    }
    if (p.fin.tickets) stats_tail<256>(p.fin, p.stats, p.Cout, trow, &ticket, tid);
}
// statistic rows of that kernel for a geometry: ~256 workgroups over all phases, never fewer per phase than the unsplit
// launch's 128-row tiles (so that either route fits the rows gcc_conv_stat_tiles promised)
static int fold_wpp(size_t max_rows, int phases) {
    const long mt = (long)((max_rows + 127) / 128);
    long w = 256 / phases;
    if (w < mt) w = mt;
    if (w > (long)max_rows) w = (long)max_rows;
    return (int)(w < 1 ? 1 : w);
}

// per-channel sum / sum of squares of an NHWC bf16 tensor -> stats[0][2][C] (one 8-channel chunk per block)
__global__ __launch_bounds__(256) void channel_stats_kernel(const bf16_t* __restrict__ x, int ld, int off, int C, size_t pixels,
                                                            float* __restrict__ stats, int tiles) {
    __shared__ float red[4][16];
    const int c0 = blockIdx.x * 8;
    float s[8], ss[8];
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = ss[j] = 0.f;
    for (size_t pix = threadIdx.x; pix < pixels; pix += 256) {
        float v[8];
        unpack8(*(const i32x4*)(x + pix * ld + off + c0), v);
#pragma unroll
        for (int j = 0; j < 8; j++) { s[j] += v[j]; ss[j] += v[j] * v[j]; }
    }
    // wave shuffle sums, then the waves through LDS (16 threads walking 256 LDS entries each cost ~5 us)
#pragma unroll
    for (int j = 0; j < 8; j++) { s[j] = wave_sum(s[j]); ss[j] = wave_sum(ss[j]); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) { red[wave][j] = s[j]; red[wave][8 + j] = ss[j]; }
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        const int j = threadIdx.x & 7, w = threadIdx.x >> 3;
        const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (c0 + j < C) stats[(size_t)w * C + c0 + j] = t;
    }
    // rows of the other tiles of the [tiles][2][C] table: zeros (the whole tensor's sums sit in row 0)
    for (int q = threadIdx.x; q < (tiles - 1) * 16; q += 256) {
        const int row = 1 + (q >> 4), w = (q >> 3) & 1, j = q & 7;
        if (c0 + j < C) stats[((size_t)row * 2 + w) * C + c0 + j] = 0.f;
    }
}

struct SplitPlan { int ksplit, kper; };
// split only launches that cannot fill the chip and have a long K loop
static SplitPlan plan_ksplit(long blocks, int nk, int max_slices = 1 << 30) {
    SplitPlan sp = {1, nk};
    if (const int f = gcc_opt(GCC_OPT_IGEMM_FORCE_KSPLIT)) {          // tuning hook
        const int s = f > nk ? nk : f;
        sp.kper = (nk + s - 1) / s;
        sp.ksplit = (nk + sp.kper - 1) / sp.kper;
        return sp;
    }
    if (blocks >= 128 || nk < 16) return sp;
    int s = (int)((1024 + blocks - 1) / blocks);
    if (s > nk / 4) s = nk / 4;
    if (s > max_slices) s = max_slices;
    if (s < 2) return sp;
    sp.kper = (nk + s - 1) / s;
    sp.ksplit = (nk + sp.kper - 1) / sp.kper;
    if (sp.ksplit < 2) { sp.ksplit = 1; sp.kper = nk; }
    return sp;
}

static bool use_glds() { return gcc_opt(GCC_OPT_IGEMM_GLDS) != 0; }

template <int BP, int BC>
int launch(const IgemmParams& p, int phases, int batch, hipStream_t st) {
    using C = Cfg<BP, BC>;
    static std::once_flag attr_once;       // one-time kernel attribute (idempotent; once per instantiation)
    std::call_once(attr_once, [] {
        hipFuncSetAttribute((const void*)igemm_kernel<BP, BC, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        hipFuncSetAttribute((const void*)igemm_kernel<BP, BC, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if constexpr (BP == 128)
            hipFuncSetAttribute((const void*)igemm_kernel<BP, BC, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    });
    dim3 grid(p.mtiles_max * p.ntiles, p.ksplit > 1 ? p.ksplit : batch, phases);
    const bool ut = (p.Ct % BK) == 0;
    bool launched = false;
    if constexpr (BP == 128 && (BC == 32 || BC == 64)) {
        // three LDS stages (two k-steps in flight) for the short-tile layers: GCC_OPT_IGEMM_STAGES
        if (ut && use_glds() && gcc_opt(GCC_OPT_IGEMM_STAGES) >= 3 && !(GCC_DIAG(p.debug) & 14)) {
            constexpr int LDS3 = 3 * (BP + BC) * BK * 2 > C::LDS_BYTES_EPI ? 3 * (BP + BC) * BK * 2 : C::LDS_BYTES_EPI;
            static std::once_flag attr3;
            std::call_once(attr3, [] {
                hipFuncSetAttribute((const void*)igemm_kernel<BP, BC, true, true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS3);
            });
            hipLaunchKernelGGL((igemm_kernel<BP, BC, true, true, 3>), grid, dim3(C::NT), LDS3, st, p);
            launched = true;
        }
    }
    if constexpr (BP == 128) {
        if (!use_glds()) {
            hipLaunchKernelGGL((igemm_kernel<BP, BC, false, false>), grid, dim3(C::NT), C::LDS_BYTES, st, p);
            launched = true;
        }
    }
    if (!launched) {
        if (ut)
            hipLaunchKernelGGL((igemm_kernel<BP, BC, true, true>), grid, dim3(C::NT), C::LDS_BYTES, st, p);
        else
            hipLaunchKernelGGL((igemm_kernel<BP, BC, true, false>), grid, dim3(C::NT), C::LDS_BYTES, st, p);
    }
    GCC_CHECK_LAUNCH();
    if (p.ksplit > 1 && !p.raw_partial && !p.pair) {
        const size_t total = (size_t)p.rows_max * (ceil8(p.Cout) / 8);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks, 1, phases), dim3(256), 0, st, p);
        GCC_CHECK_LAUNCH();
    }
    return GCC_OK;
}

// Tile choice (shared by the launcher, gcc_conv_stat_tiles and gcc_conv_workspace).
struct TilePlan { int BP, BC, ntiles, mtiles, max_slices, pair; };
// does a launch take the LDS-resident-neighbourhood kernel (conv_halo.hip)?  With statistics its tile rows must be the ones
// gcc_conv_stat_tiles() promised: the stride-2 form keeps the 256-pixel plan's rows, the stride-1 form has its own (and
// gcc_conv_stat_tiles reports them).
static bool halo_routed(const gcc_conv_plan_t& pl, const HaloPlan& h, int dgrad, bool with_stats, int plan_bp) {
    if (!h.ok || h.wgs < plan_or(pl.big_min, PLAN_BIG_MIN)) return false;
    if (!with_stats) return true;
    if (dgrad) return false;
    return h.mode == 3 || plan_bp == 256;
}
static TilePlan select_tile(const gcc_conv_plan_t& pl, size_t max_rows, int Cout, int phases, int nk, int batch) {
    TilePlan t;
    t.BP = 128;
    t.max_slices = 1 << 30;
    t.pair = 0;
    if (Cout > 64) { t.BC = 128; t.ntiles = cdiv(Cout, 128); }
    else if (Cout > 32) { t.BC = 64; t.ntiles = 1; }
    else if (Cout > 16) { t.BC = 32; t.ntiles = 1; }
    else { t.BC = 16; t.ntiles = 1; }
    if (gcc_opt(GCC_OPT_IGEMM_NARROW) && batch == 1 && t.BC > 32) {
        // Mid-size layers (the U-Nets at 32 x 32 .. 8 x 8): 128 x 128 tiles leave most CUs without a workgroup.  Measured
        // per shape inside the step (profiles/r02_j, r02_m):  (a) K loop of <= 64 steps: the widest tile that still gives
        // every CU a workgroup, un-split (one kernel instead of partials + fold + statistics);  (b) longer loops on >= 8
        // M tiles: 64-channel tiles, K split at most 8 ways by plan_ksplit (the 128-wide plan splits 16-32 ways there:
        // tens of MB of fp32 partials).  Smaller launches keep the 128-wide tile with its deep K split: they stream their
        // weights cold from HBM and want every CU loading.
        const long mt = (long)((max_rows + 127) / 128) * phases;
        if (nk <= 64) {
            int bc = t.BC;
            while (bc > 32 && mt * cdiv(Cout, bc) < 256) bc >>= 1;
            if (mt * cdiv(Cout, bc) >= 256) { t.BC = bc; t.ntiles = cdiv(Cout, bc); }
        } else if (mt >= 8 && mt * cdiv(Cout, t.BC) < 128) {
            t.BC = 64; t.ntiles = cdiv(Cout, 64); t.max_slices = 8;
        }
    }
    if (const int f = gcc_opt(GCC_OPT_IGEMM_FORCE_BC)) {               // tuning hook
        if (f == 16 || f == 32 || f == 64 || f == 128) { t.BC = f; t.ntiles = cdiv(Cout, f); }
    }
    // the plan (gcc_conv_t.plan): which tile families are allowed, minimum number of 256-pixel tiles, minimum K depth
    const int g_big_tiles = plan_or(pl.tile_families, PLAN_TILE_FAMILIES) - 1, g_big_min = plan_or(pl.big_min, PLAN_BIG_MIN),
              g_big_nk = plan_or(pl.big_nk, PLAN_BIG_NK);
    // 256-pixel tiles (one 8-wave workgroup per CU) when they still fill the chip and the K loop is
    // long enough to amortise the un-overlapped prologue / epilogue of a lone workgroup
    if (g_big_tiles && use_glds() && batch == 1 && Cout >= 128 && nk >= g_big_nk) {
        const long m256 = (long)((max_rows + 255) / 256);
        if (g_big_tiles >= 2 && Cout % 256 == 0 && m256 * (Cout / 256) * phases >= g_big_min) {
            t.BP = 256; t.BC = 256; t.ntiles = Cout / 256;
            // fewer than ~3/4 of a chip of one-per-CU workgroups and a long loop: two workgroups per tile, one per K half
            t.pair = (pl.pair == 1 && m256 * (Cout / 256) * phases < 192 && nk >= 48) ? 1 : 0;
        } else if (m256 * cdiv(Cout, 128) * phases >= g_big_min) {
            t.BP = 256; t.BC = 128; t.ntiles = cdiv(Cout, 128);
        }
    }
    t.mtiles = (int)((max_rows + t.BP - 1) / t.BP);
    return t;
}
constexpr size_t PAIR_FLAG_BYTES = 4096;          // ticket / ready words of up to 512 tiles, a block of its own at the workspace's start
static size_t pair_workspace(size_t tiles) { return tiles > 512 ? ~(size_t)0 : PAIR_FLAG_BYTES + tiles * (size_t)(256 * 256 * 4); }
static int conv_nk(const gcc_conv_t* c, int dgrad) {
    const int taps_max = dgrad ? cdiv(c->KH, c->stride) * cdiv(c->KW, c->stride) : c->KH * c->KW;
    return cdiv(taps_max * ceil8(dgrad ? c->Co : c->Ci), BK);
}
static size_t conv_max_rows(const gcc_conv_t* c, int dgrad) {
    if (!dgrad) return (size_t)c->N * gcc_conv_out(c->H, c->KH, c->stride, c->pad) * gcc_conv_out(c->W, c->KW, c->stride, c->pad);
    return (size_t)c->N * cdiv(c->H, c->stride) * cdiv(c->W, c->stride);
}

int check_conv(const gcc_conv_t* c) {
    if (!c) return GCC_ERR_BAD_ARG;
    if (c->N <= 0 || c->H <= 0 || c->W <= 0 || c->Ci <= 0 || c->Co <= 0 || c->KH <= 0 || c->KW <= 0 ||
        c->stride <= 0 || c->pad < 0)
        return GCC_ERR_BAD_ARG;
    if ((c->ldx & 7) || (c->xoff & 7) || (c->ldy & 7) || (c->yoff & 7)) return GCC_ERR_BAD_ARG;
    if (c->ldx < c->xoff + ceil8(c->Ci) || c->ldy < c->yoff + ceil8(c->Co)) return GCC_ERR_BAD_ARG;
    if (gcc_conv_out(c->H, c->KH, c->stride, c->pad) <= 0 || gcc_conv_out(c->W, c->KW, c->stride, c->pad) <= 0)
        return GCC_ERR_BAD_ARG;
    return GCC_OK;
}

}  // namespace gcc_igemm
using namespace gcc_igemm;

// ---------------------------------------------------------------------------------------------
// Single-output-channel convolutions (the PatchGAN head, 1024 -> 1, k4 s1 p1: 80 us per launch as a 128x16 implicit
// GEMM because the gather re-reads the 31 MB input once per tap).  Reformulated so that the input is read once:
//   fprop : T[pixel][tap] = x[pixel][:] . w[tap][:]   (a 1x1 conv with `taps` output channels, fp32 partial tiles)
//           y[oy][ox] = bias + sum_tap T[(oy*s + kh - pad, ox*s + kw - pad)][tap]          (head_tapsum_kernel)
//   wgrad : G[pixel][tap*8] = dy at the output position that tap connects the pixel to   (head_gather_kernel);
//           dW[tap][:] = sum_pixels G[pixel][tap*8] x[pixel][:], a 1x1 weight gradient   (conv_wgrad.hip)
//   (backward-data stays on the generic path: it is bound by writing dx and was measured no faster this way)
struct HeadArgs {
    const float* partial; int ksplit, rows_max, Cpad;
    const float* bias; int act; float slope;
    bf16_t* dst; int ldd, doff;
    const bf16_t* dy; int lddy, dyoff;
    bf16_t* g;
    int N, H, W, Ho, Wo, KH, KW, stride, pad;
};
// 16 lanes per output (one per tap, taps <= 16), each summing its tap's split-K partials, then a 16-lane shuffle sum:
// one thread per output walked taps x ksplit dependent loads (21 us for 14400 outputs)
__global__ __launch_bounds__(256) void head_tapsum_kernel(const HeadArgs a) {
    const size_t total = (size_t)a.N * a.Ho * a.Wo;
    const int tap = threadIdx.x & 15;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    for (size_t q0 = (size_t)blockIdx.x * 16; q0 < total; q0 += (size_t)gridDim.x * 16) {
        const size_t q = q0 + (threadIdx.x >> 4);
        float acc = 0.f;
        if (q < total && kh < a.KH) {
            const int ox = (int)(q % a.Wo);
            const size_t t = q / a.Wo;
            const int oy = (int)(t % a.Ho);
            const size_t n = t / a.Ho;
            const int iy = oy * a.stride + kh - a.pad, ix = ox * a.stride + kw - a.pad;
            if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
                const size_t row = (n * a.H + iy) * (size_t)a.W + ix;
                for (int s = 0; s < a.ksplit; s++) acc += a.partial[((size_t)s * a.rows_max + row) * a.Cpad + tap];
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 16);
        if (tap == 0 && q < total) {
            float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            o[0] = apply_act(acc + (a.bias ? a.bias[0] : 0.f), a.act, a.slope);
            *(i32x4*)(a.dst + q * a.ldd + a.doff) = pack8(o);
        }
    }
}
// G[pixel][tap*8 + 0] = dy[n][oy][ox] with oy*s + kh - pad == iy (if such an output exists), zeros elsewhere
__global__ __launch_bounds__(256) void head_gather_kernel(const HeadArgs a) {
    const int taps = a.KH * a.KW;
    const size_t total = (size_t)a.N * a.H * a.W * taps;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
        const int tap = (int)(q % taps);
        const size_t pix = q / taps;
        const int ix = (int)(pix % a.W);
        const size_t t = pix / a.W;
        const int iy = (int)(t % a.H);
        const size_t n = t / a.H;
        const int kh = tap / a.KW, kw = tap - kh * a.KW;
        const int ny = iy + a.pad - kh, nx = ix + a.pad - kw;
        float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (ny >= 0 && nx >= 0 && ny % a.stride == 0 && nx % a.stride == 0) {
            const int oy = ny / a.stride, ox = nx / a.stride;
            if (oy < a.Ho && ox < a.Wo) o[0] = bf2f(a.dy[((n * a.Ho + oy) * (size_t)a.Wo + ox) * a.lddy + a.dyoff]);
        }
        *(i32x4*)(a.g + q * 8) = pack8(o);
    }
}

static bool head_enabled() { return gcc_opt(GCC_OPT_IGEMM_HEAD) != 0; }
// rows of the tap matrix / bytes of the gathered dy matrix
static size_t head_rows(const gcc_conv_t* c) { return (size_t)c->N * c->H * c->W; }
static bool head_shape(const gcc_conv_t* c) {
    return c->Co == 1 && c->KH * c->KW >= 4 && c->KH * c->KW <= 16 && ceil8(c->Ci) >= 256 && head_enabled();
}
size_t gcc_internal_head_gather_bytes(const gcc_conv_t* c) { return head_shape(c) ? head_rows(c) * c->KH * c->KW * 8 * 2 : 0; }
int gcc_internal_head_gather(const gcc_conv_t* c, const void* dy, void* g, hipStream_t st) {
    HeadArgs a = {};
    a.dy = (const bf16_t*)dy; a.lddy = c->ldy; a.dyoff = c->yoff; a.g = (bf16_t*)g;
    a.N = c->N; a.H = c->H; a.W = c->W; a.KH = c->KH; a.KW = c->KW; a.stride = c->stride; a.pad = c->pad;
    a.Ho = gcc_conv_out(c->H, c->KH, c->stride, c->pad); a.Wo = gcc_conv_out(c->W, c->KW, c->stride, c->pad);
    const size_t total = head_rows(c) * c->KH * c->KW;
    size_t b = (total + 255) / 256;
    hipLaunchKernelGGL(head_gather_kernel, dim3((unsigned)(b > 4096 ? 4096 : b)), dim3(256), 0, st, a);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}
static size_t head_fprop_workspace(const gcc_conv_t* c, int* ksplit_out) {
    const size_t rows = head_rows(c);
    const int nk = cdiv(ceil8(c->Ci), BK);
    const SplitPlan sp = plan_ksplit((long)cdiv((int)rows, 128), nk);
    if (ksplit_out) *ksplit_out = sp.ksplit;
    return (size_t)sp.ksplit * rows * 16 * sizeof(float);
}

// ---------------------------------------------------------------------------------------------
// Thin-input convolutions (<= 8 input channels: the image layers 3->ngf / 6->ndf, and the data gradient of the
// ConvTranspose that produces the image).  K = taps x 8 is at most two k-steps of the implicit GEMM, whose launch is
// then all prologue and epilogue (45 us for 6->128 at 256x256 against 15 us of HBM time).  Here a wave owns 16 output
// pixels at a time and no LDS is involved: with one tap = 8 padded channels = 16 bytes, a 16x16x32 MFMA operand
// (lane = pixel, 8 consecutive k) IS one 16-byte load of x per lane (tap kc*4 + lane/16), the weights of up to 128
// output channels stay in registers for the whole launch, and the product is formed transposed (rows = output
// channels, permuted so that a lane ends up with 8 consecutive channels of its pixel = one 16-byte store).
struct ThinArgs {
    const bf16_t* x; const bf16_t* w; bf16_t* y; const float* bias;
    int act; float slope;
    int H, W, Ho, Wo, KW, stride, pad, taps;
    int ldx, xoff, ldy, yoff, Co;
    uint32_t x_bytes;
    int total;            // N * Ho * Wo
    FastDiv dHoWo, dWo;
    // second output written by the same launch (round 4: gcc_epilogue_t.y2): f(y) of the rounded first output --
    // mode 1: relu(y) (the U-Net's first skip: lin = lrelu(e0) for the next down conv, relu(e0) into the concat buffer; exact,
    // relu(v) == max(lrelu(v), 0)); mode 2: y * gate[c] (the selective-activation PatchGAN's first gate, m in {0, .5, 1})
    bf16_t* y2; int ldy2, y2off, mode2; const float* gate;
};

template <int NJ>      // 32-channel groups per workgroup column
__global__ __launch_bounds__(256) void thin_fprop_kernel(const ThinArgs a) {
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int co_base = blockIdx.y * (NJ * 32);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (uint32_t)(a.Co * a.taps * 16), 0x00020000);

    // weight operand: tile (t, h) row m = 4q + r holds channel co_base + 32t + 8q + 4h + r.  The NJ*8 operands live in
    // LDS in fragment order ([operand][lane] x 16 bytes: a wave reads 1 KiB contiguous per MFMA), loaded once per
    // workgroup: registers stay free for occupancy, which is what hides the load / store latencies here.
    __shared__ __attribute__((aligned(16))) char wlds[NJ * 8 * 1024];
#pragma unroll
    for (int t = 0; t < NJ; t++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int kc = wave;
            const int co = co_base + 32 * t + 8 * (i >> 2) + 4 * h + (i & 3);
            const int tap = kc * 4 + g;
            const uint32_t off = (co < a.Co && tap < a.taps) ? (uint32_t)((co * a.taps + tap) * 16) : OOB;
            *(i32x4*)(wlds + (((t * 2 + h) * 4 + kc) * 64 + lane) * 16) = __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0);
        }
    __syncthreads();
    float bv[NJ][8];
#pragma unroll
    for (int t = 0; t < NJ; t++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int co = co_base + 32 * t + 8 * g + e;
            bv[t][e] = (a.bias && co < a.Co) ? a.bias[co] : 0.f;
        }
    // out = v > 0 ? v : v * neg  covers none (1), ReLU (0) and LeakyReLU (slope); tanh layers take the generic kernel
    const float neg = a.act == GCC_ACT_LRELU ? a.slope : (a.act == GCC_ACT_RELU ? 0.f : 1.f);
    int tdy[4], tdx[4];
#pragma unroll
    for (int kc = 0; kc < 4; kc++) {
        const int tap = kc * 4 + g;
        const int kh = tap / a.KW;
        tdy[kc] = tap < a.taps ? kh - a.pad : -(1 << 28);
        tdx[kc] = tap - kh * a.KW - a.pad;
    }

    // tiles past the end fetch nothing (out-of-range offsets return zeros without touching memory)
    auto fetch = [&](int tile, i32x4* xb) {
        const int q = tile * 16 + i;
        const bool qv = q < a.total;
        const int n = fdiv(qv ? q : 0, a.dHoWo);
        const int r = (qv ? q : 0) - n * (a.Ho * a.Wo);
        const int oy = fdiv(r, a.dWo);
        const int ox = r - oy * a.Wo;
#pragma unroll
        for (int kc = 0; kc < 4; kc++) {
            const int iy = oy * a.stride + tdy[kc], ix = ox * a.stride + tdx[kc];
            const bool ok = qv && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const uint32_t off = ok ? (uint32_t)((((n * a.H + iy) * a.W + ix) * a.ldx + a.xoff) * 2) : OOB;
            xb[kc] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0);
        }
    };

    constexpr int ROWB = NJ * 64 + 16;
    __shared__ __attribute__((aligned(16))) char stage_all[4][16 * ROWB];
    char* stage = stage_all[wave];
    // the gate of the second output (Y2_GATE: the masked PatchGAN's first layer): a lane stores the same eight channels of every
    // pixel it stores, so their mask values are read ONCE -- read per stored chunk inside the loop (eight dependent global loads
    // in front of every second store) they made the two-output launch take 84 us against 36 for one output (N = 16, 6 -> 128 at
    // 256 x 256: 67 MB more to write; scratch/r6/bench_d_l1.py)
    float gatev[8];
    {
        const int cg = co_base + (lane % (NJ * 4)) * 8;
#pragma unroll
        for (int e = 0; e < 8; e++) gatev[e] = (a.y2 && a.mode2 != 1 && cg + e < a.Co) ? a.gate[cg + e] : 0.f;
    }

    auto compute = [&](int tile, const i32x4* xb) {
        f32x4 acc[NJ][2];
#pragma unroll
        for (int t = 0; t < NJ; t++)
#pragma unroll
            for (int h = 0; h < 2; h++) acc[t][h] = f32x4{0.f, 0.f, 0.f, 0.f};
        // weight operand k = (t * 2 + h) * 4 + kc, one per product: a software pipeline with six LDS reads ahead of their products
        // (round 5, as conv_ring3.hip: left alone the compiler keeps one read ahead and every product waits for its operand)
        constexpr int NI = NJ * 8, PF = 6;
        bf16x8 wv[NI];
#pragma unroll
        for (int k = 0; k < PF; k++) wv[k] = *(const bf16x8*)(wlds + (k * 64 + lane) * 16);
        __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
        for (int k = 0; k < NI; k++) {
            if (k + PF < NI) wv[k + PF] = *(const bf16x8*)(wlds + ((k + PF) * 64 + lane) * 16);
            acc[k >> 3][(k >> 2) & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[k], __builtin_bit_cast(bf16x8, xb[k & 3]), acc[k >> 3][(k >> 2) & 1], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (k + PF < NI) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        // lane (i, g) holds pixel tile*16 + i, channels co_base + 32t + 8g + {0..7} (h = 0: first four, h = 1: last four):
        // a store from here would touch 16 rows x 64 bytes.  The wave's [16 pixels][NJ*64 bytes] block goes through a
        // wave-private LDS image (row stride padded by 16 bytes: conflict-free both ways) so that each store instruction
        // writes whole rows: 64 / (NJ*4) pixels x NJ*64 contiguous bytes.
#pragma unroll
        for (int t = 0; t < NJ; t++) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float v0 = acc[t][0][e] + bv[t][e], v1 = acc[t][1][e] + bv[t][4 + e];
                o[e] = fmaxf(v0, v0 * neg);             // neg in [0, 1]: v for v > 0, v * neg below
                o[4 + e] = fmaxf(v1, v1 * neg);
            }
            const int co = co_base + 32 * t + 8 * g;
            if (a.Co & 7) {                                 // padding channels of the 8-wide group stay zero
#pragma unroll
                for (int e = 0; e < 8; e++)
                    if (co + e >= a.Co) o[e] = 0.f;
            }
            *(i32x4*)(stage + i * ROWB + t * 64 + g * 16) = pack8(o);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        constexpr int CPR = NJ * 4, PPI = 64 / CPR;       // 16-byte chunks per row, pixels per store instruction
        const int c = lane % CPR, pl = lane / CPR;
        const int co = co_base + c * 8;
#pragma unroll
        for (int sidx = 0; sidx < 16 / PPI; sidx++) {
            const int pix = sidx * PPI + pl;
            const i32x4 v = *(const i32x4*)(stage + pix * ROWB + c * 16);
            const int q = tile * 16 + pix;
            if (q < a.total && co < a.Co) {
                *(i32x4*)(a.y + (size_t)q * a.ldy + a.yoff + co) = v;
                if (a.y2) {
                    float f[8];
                    unpack8(v, f);
                    if (a.mode2 == 1) {
#pragma unroll
                        for (int e = 0; e < 8; e++) f[e] = fmaxf(f[e], 0.f);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; e++) f[e] *= gatev[e];
                    }
                    *(i32x4*)(a.y2 + (size_t)q * a.ldy2 + a.y2off + co) = pack8(f);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the image is rewritten by the next tile
    };

    // persistent waves, two pixel tiles in flight behind the one being multiplied; three statically named buffers
    // (a rotating array would make the compiler wait for the newest loads at every iteration)
    const int ntiles = (a.total + 15) >> 4;
    const int nw = gridDim.x * 4;
    int tile = blockIdx.x * 4 + wave;
    i32x4 b0[4], b1[4], b2[4];
    fetch(tile, b0);
    fetch(tile + nw, b1);
    while (tile < ntiles) {
        fetch(tile + 2 * nw, b2);
        compute(tile, b0);
        tile += nw;
        if (tile >= ntiles) break;
        fetch(tile + 2 * nw, b0);
        compute(tile, b1);
        tile += nw;
        if (tile >= ntiles) break;
        fetch(tile + 2 * nw, b1);
        compute(tile, b2);
        tile += nw;
    }
}

static bool thin_enabled() { return gcc_opt(GCC_OPT_IGEMM_THIN) != 0; }
// launch_thin's own size limits (32-bit buffer offsets of the input, the pixel counter) are part of the shape test: what
// gcc_conv_y2_supported() promises is exactly what launch_thin() takes
static bool thin_sizes_ok(const gcc_conv_t* c) {
    const size_t xb = (size_t)c->N * c->H * c->W * c->ldx * 2;
    const size_t outs = (size_t)c->N * gcc_conv_out(c->H, c->KH, c->stride, c->pad) * gcc_conv_out(c->W, c->KW, c->stride, c->pad);
    return xb < OOB && outs < ((size_t)1 << 30);
}
static bool thin_shape(const gcc_conv_t* c) {
    return ceil8(c->Ci) == 8 && c->KH * c->KW <= 16 && c->Co >= 16 && (size_t)c->Co * c->KH * c->KW * 16 < OOB && thin_enabled() &&
           thin_sizes_ok(c);
}
static int launch_thin(const gcc_conv_t* c, const void* x, const void* w, void* y, const gcc_epilogue_t* ep, hipStream_t st) {
    ThinArgs a;
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.y = (bf16_t*)y;
    a.bias = ep ? ep->bias : nullptr; a.act = ep ? ep->act : GCC_ACT_NONE; a.slope = ep ? ep->slope : 0.f;
    a.H = c->H; a.W = c->W; a.KW = c->KW; a.stride = c->stride; a.pad = c->pad; a.taps = c->KH * c->KW;
    a.Ho = gcc_conv_out(c->H, c->KH, c->stride, c->pad); a.Wo = gcc_conv_out(c->W, c->KW, c->stride, c->pad);
    a.ldx = c->ldx; a.xoff = c->xoff; a.ldy = c->ldy; a.yoff = c->yoff; a.Co = c->Co;
    a.y2 = nullptr; a.ldy2 = a.y2off = a.mode2 = 0; a.gate = nullptr;
    if (ep && ep->y2) {
        a.y2 = (bf16_t*)ep->y2; a.ldy2 = ep->ldy2; a.y2off = ep->y2off; a.mode2 = ep->y2_mode; a.gate = ep->y2_gate;
    }
    const size_t xb = (size_t)c->N * c->H * c->W * c->ldx * 2, outs = (size_t)c->N * a.Ho * a.Wo;
    if (xb >= OOB || outs >= (size_t)1 << 30) return -1;
    a.x_bytes = (uint32_t)xb; a.total = (int)outs;
    a.dHoWo = make_fastdiv(a.Ho * a.Wo); a.dWo = make_fastdiv(a.Wo);
    const int ntiles = (a.total + 15) / 16;
    // persistent waves: weights are fetched once per wave, so no more workgroups than fill the chip twice
    // persistent waves: three workgroups per CU (the 128-channel form's 50 KB of LDS and 133 VGPRs allow exactly that;
    // measured best for the narrower forms too)
    const int nj = c->Co > 64 ? 4 : (c->Co > 32 ? 2 : 1);
    const int cols = cdiv(c->Co, nj * 32);
    int wgs = cdiv(ntiles, 4);
    if (wgs > 768) wgs = 768;
    const dim3 grid(wgs, cols);
    if (nj == 4) hipLaunchKernelGGL(thin_fprop_kernel<4>, grid, dim3(256), 0, st, a);
    else if (nj == 2) hipLaunchKernelGGL(thin_fprop_kernel<2>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(thin_fprop_kernel<1>, grid, dim3(256), 0, st, a);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

// Thin-output backward-data / ConvTranspose forward (<= 8 channels out, k4 s2 p1: the image-producing ConvTranspose of
// the generators and the data gradient of the first discriminator layer).  As a 128 x 16 implicit GEMM each of the four
// phases re-gathers dy and pays its own prologue (80 us for 128 -> 6 at 256 x 256 against 15 us of HBM time).  Here a
// wave owns 16 horizontally adjacent output PAIRS (iy, 2j) / (iy, 2j+1) of one row parity: the 16 MFMA rows are
// (pixel of the pair, channel), the contraction runs over the 2 x 3 dy pixels the pair touches x Co (a third of the weight
// operand is structural zeros), the dy operand is one 16-byte load per lane straight from global memory, and the weight
// operands of the row parity sit in LDS in fragment order.  Bound by the texture path (every dy byte is requested three
// times per output pixel): 43 -> 17 us for 32 channels in, 55 -> 48 for 64, slower than the implicit GEMM for 128 (95 vs
// 80 us), so it serves Co <= 64; the wide case needs dy rows staged in LDS (a ring of rows per workgroup) -- open.
struct ThinDgradArgs {
    const bf16_t* dy; const bf16_t* wt; bf16_t* dx; const float* bias;
    int act; float slope;
    int N, H, W, Ho, Wo, ldy, yoff, ldx, xoff, Ci, Co8;
    uint32_t dy_bytes, wt_bytes;
    int tiles_per_row, ntiles;        // per row parity: N * (H/2) * tiles_per_row
    FastDiv dTpr, dHh;
};

template <int NCC>      // 32-channel chunks of Co
__global__ __launch_bounds__(256) void thin_dgrad_k4s2_kernel(const ThinDgradArgs a) {
    constexpr int NK = 6 * NCC;
    __shared__ __attribute__((aligned(16))) char wlds[NK * 1024];
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int py = blockIdx.y;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);
    // operand f = (u, b, cc): dy row r + dr(u), dy column j + b (b = -1, 0, +1), channels 32cc + 8g ..
    //   row parity 0 (iy = 2r)  : kh = 1 -> oy = r, kh = 3 -> oy = r - 1;   parity 1 (iy = 2r + 1): kh = 0 -> r + 1, kh = 2 -> r
    //   pixel 0 (ix = 2j)       : kw = 1 -> ox = j, kw = 3 -> ox = j - 1;   pixel 1 (ix = 2j + 1) : kw = 0 -> j + 1, kw = 2 -> j
    for (int f = wave; f < NK; f += 4) {
        const int u = f / (3 * NCC), b = (f / NCC) % 3 - 1, cc = f % NCC;
        const int px = i >> 3, ci = i & 7;
        const int kh = py == 0 ? (u == 0 ? 1 : 3) : (u == 0 ? 0 : 2);
        const int kw = px == 0 ? (b == 0 ? 1 : (b < 0 ? 3 : -1)) : (b > 0 ? 0 : (b == 0 ? 2 : -1));
        const int co = 32 * cc + 8 * g;
        const uint32_t off = (kw >= 0 && ci < a.Ci && co < a.Co8) ? (uint32_t)((((ci * 16 + kh * 4 + kw) * a.Co8) + co) * 2) : OOB;
        *(i32x4*)(wlds + (f * 64 + lane) * 16) = __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0);
    }
    __syncthreads();
    const int ci0 = (g & 1) * 4, pxo = g >> 1;
    float bv[4];
#pragma unroll
    for (int e = 0; e < 4; e++) bv[e] = (a.bias && ci0 + e < a.Ci) ? a.bias[ci0 + e] : 0.f;

    const int nw = gridDim.x * 4;
    for (int tile = blockIdx.x * 4 + wave; tile < a.ntiles; tile += nw) {
        const int nr = fdiv(tile, a.dTpr);               // n * (H/2) + r
        const int jt = tile - nr * a.tiles_per_row;
        const int n = fdiv(nr, a.dHh), r = nr - n * (a.H >> 1);
        const int j = jt * 16 + i;
        i32x4 yb[NK];
#pragma unroll
        for (int f = 0; f < NK; f++) {
            const int u = f / (3 * NCC), b = (f / NCC) % 3 - 1, cc = f % NCC;
            const int oy = r + (py == 0 ? (u == 0 ? 0 : -1) : (u == 0 ? 1 : 0));
            const int ox = j + b;
            const int co = 32 * cc + 8 * g;
            const bool ok = (unsigned)oy < (unsigned)a.Ho && (unsigned)ox < (unsigned)a.Wo && co < a.Co8;
            const uint32_t off = ok ? (uint32_t)((((n * a.Ho + oy) * a.Wo + ox) * a.ldy + a.yoff + co) * 2) : OOB;
            yb[f] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, off, 0, 0);
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};       // two chains: MFMA back-to-back dependency
        // (round 5: pipelining these weight reads six deep, as in thin_fprop_kernel, made the 64-channel layer SLOWER, 51 -> 58 us:
        // the twelve global loads of the tile are what the wave waits for, and more live registers cut the waves that hide them)
#pragma unroll
        for (int f = 0; f < NK; f += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(wlds + (f * 64 + lane) * 16),
                                                           __builtin_bit_cast(bf16x8, yb[f]), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(wlds + ((f + 1) * 64 + lane) * 16),
                                                           __builtin_bit_cast(bf16x8, yb[f + 1]), acc1, 0, 0, 0);
        }
        // lane (i, g): output pixel (iy = 2r + py, ix = 2j + (g >> 1)), channels 4 (g & 1) + {0..3}
        const int ix = 2 * j + pxo;
        if (ix < a.W) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                o[e] = apply_act(acc0[e] + acc1[e] + bv[e], a.act, a.slope);
                if (ci0 + e >= a.Ci) o[e] = 0.f;
            }
            i32x2 v = {(int)pack2bf(o[0], o[1]), (int)pack2bf(o[2], o[3])};
            *(i32x2*)(a.dx + ((size_t)(n * a.H + 2 * r + py) * a.W + ix) * a.ldx + a.xoff + ci0) = v;
        }
    }
}

// The wide case (65 .. 128 channels in: the first PatchGAN layer's data gradient at ndf 128, the teacher generator's last
// ConvTranspose): the texture-path form above requests every dy byte three times per output pixel and loses to the implicit
// GEMM there.  Here the dy rows are staged ONCE by LDS-DMA ([Wo][256 B] per row, the sixteen 16-byte chunks of a pixel
// XOR-swizzled with the pixel index on the source side, zero rows / channels by the descriptor range check), and every MFMA
// operand of the two output rows a row pair (r, r + 1) feeds is a conflict-free ds_read_b128 out of them.
constexpr int THIN_WIDE_RB = 256;                 // bytes per staged pixel (128 channels)
// One 8-wave workgroup per CU walks `rows_per_wg` consecutive r of one image with a ring of three staged dy rows: row r + 2
// flies (LDS-DMA) while the output rows of r are computed; the weight fragments of both row parities sit in LDS in fragment
// order (48 KB, filled once per workgroup).  Two waves per SIMD hide the ds_read latency (with the weights in registers --
// 192 VGPRs, one wave per SIMD -- every ds_read -> MFMA pair ran exposed: 55 us); dy is read from L2 / HBM about once.
constexpr int THIN_WIDE_NT = 512;
__global__ __launch_bounds__(THIN_WIDE_NT) void thin_dgrad_wide_kernel(const ThinDgradArgs a, int rows_per_wg, int wgs_per_image) {
    constexpr int NCC = 4, NK = 6 * NCC, NW = THIN_WIDE_NT / 64;
    extern __shared__ __attribute__((aligned(16))) char smem_w[];
    char* wlds = smem_w;                                                 // [2 parities][NK][1 KiB] weight fragments
    char* ylds = smem_w + 2 * NK * 1024;                                 // [3 ring slots][Wo + 4][THIN_WIDE_RB]
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.x / wgs_per_image, wb = blockIdx.x % wgs_per_image;
    const int r_first = wb == 0 ? -1 : wb * rows_per_wg;                 // r = -1: output row 0 (dy rows -1 (zeros) and 0)
    int r_last = (wb + 1) * rows_per_wg - 1;
    if (r_last > a.Ho - 1) r_last = a.Ho - 1;
    if (r_first > r_last) return;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);
    // a staged row holds pixels -2 .. Wo + 1 (two zero pixels on each side: the taps at ox = -1 and ox = Wo read zeros instead
    // of being predicated); staged pixel p' = ox + 2.  dy row y -> ring slot (y + 1) % 3.  Lane L of piece q lands at
    // q * 1024 + L * 16, which must hold chunk slot ^ (p' & 15) of pixel p'.
    const int row_bytes = (a.Wo + 4) * THIN_WIDE_RB;
    const int pieces = row_bytes / 1024;                                 // 1 KiB = 4 pixels per wave instruction
    auto stage_row = [&](int y) {
        char* base = ylds + ((y + 1) % 3) * row_bytes;
        const bool row_ok = (unsigned)y < (unsigned)a.Ho;
        for (int q = wave; q < pieces; q += NW) {
            const int o = q * 1024 + lane * 16;
            const int pp = o / THIN_WIDE_RB, slot = (o % THIN_WIDE_RB) >> 4;
            const int co = ((slot ^ (pp & 15)) << 3);
            const int ox = pp - 2;
            const bool ok = row_ok && (unsigned)ox < (unsigned)a.Wo && co < a.Co8;
            const uint32_t off = ok ? (uint32_t)((((n * a.Ho + y) * a.Wo + ox) * a.ldy + a.yoff + co) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, LDS_PTR(void, base + q * 1024), 16, off, 0, 0, 0);
        }
    };
    stage_row(r_first);
    stage_row(r_first + 1);
    // weight fragments (operand f = (u, b, cc) as in thin_dgrad_k4s2_kernel), parity py = f2 / NK
    for (int f2 = wave; f2 < 2 * NK; f2 += NW) {
        const int py = f2 / NK, f = f2 - py * NK;
        const int u = f / (3 * NCC), b = (f / NCC) % 3 - 1, cc = f % NCC;
        const int px = i >> 3, ci = i & 7;
        const int kh = py == 0 ? (u == 0 ? 1 : 3) : (u == 0 ? 0 : 2);
        const int kw = px == 0 ? (b == 0 ? 1 : (b < 0 ? 3 : -1)) : (b > 0 ? 0 : (b == 0 ? 2 : -1));
        const int co = 32 * cc + 8 * g;
        const uint32_t off = (kw >= 0 && ci < a.Ci && co < a.Co8) ? (uint32_t)((((ci * 16 + kh * 4 + kw) * a.Co8) + co) * 2) : OOB;
        *(i32x4*)(wlds + (f2 * 64 + lane) * 16) = __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0);
    }
    // per-lane byte offsets of the 12 (b, cc) operands inside a staged row, for tile 0; a tile adds 16 pixels = 4096 bytes
    // (16 pixels keep p' & 15, so the swizzle term is tile-independent)
    int yoffs[3][NCC];
#pragma unroll
    for (int bb = 0; bb < 3; bb++)
#pragma unroll
        for (int cc = 0; cc < NCC; cc++) {
            const int pp = i + (bb - 1) + 2;
            yoffs[bb][cc] = pp * THIN_WIDE_RB + (((4 * cc + g) ^ (pp & 15)) << 4);
        }
    const int ci0 = (g & 1) * 4, pxo = g >> 1;
    float bv[4];
#pragma unroll
    for (int e = 0; e < 4; e++) bv[e] = (a.bias && ci0 + e < a.Ci) ? a.bias[ci0 + e] : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll 1
    for (int r = r_first; r <= r_last; r++) {
        if (r < r_last) stage_row(r + 2);                      // into the slot of row r - 1, which nobody reads any more
        const char* row_u0 = ylds + ((r + 2) % 3) * row_bytes;   // u == 0: dy row r + 1 (both parities)
        const char* row_u1 = ylds + ((r + 1) % 3) * row_bytes;   // u == 1: dy row r
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            const int py = pass == 0 ? 1 : 0;                   // output row 2r + 1 (parity 1 of r), then 2r + 2 (parity 0 of r + 1)
            const int iy = pass == 0 ? 2 * r + 1 : 2 * r + 2;
            if ((unsigned)iy >= (unsigned)a.H) continue;        // workgroup-uniform: the rows above / below the image
            const char* wp = wlds + (py * NK * 64 + lane) * 16;
            for (int jt = wave; jt < a.tiles_per_row; jt += NW) {
                const char* t0 = row_u0 + jt * (16 * THIN_WIDE_RB);
                const char* t1 = row_u1 + jt * (16 * THIN_WIDE_RB);
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                // both operands of a product come from LDS: a software pipeline with four pairs of reads ahead of their products (round 5)
                constexpr int PF = 4;
                bf16x8 yv[NK], wv[NK];
                auto rdy = [&](int ff) {
                    const int u = ff / (3 * NCC), bb = (ff / NCC) % 3, cc = ff % NCC;
                    return *(const bf16x8*)((u == 0 ? t0 : t1) + yoffs[bb][cc]);
                };
#pragma unroll
                for (int f = 0; f < PF; f++) { yv[f] = rdy(f); wv[f] = *(const bf16x8*)(wp + f * 1024); }
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * PF, 0);
#pragma unroll
                for (int f = 0; f < NK; f++) {
                    if (f + PF < NK) { yv[f + PF] = rdy(f + PF); wv[f + PF] = *(const bf16x8*)(wp + (f + PF) * 1024); }
                    if (f & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[f], yv[f], acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[f], yv[f], acc0, 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (f + PF < NK) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                const int ix = 2 * (jt * 16 + i) + pxo;
                if (ix < a.W) {
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        o[e] = apply_act(acc0[e] + acc1[e] + bv[e], a.act, a.slope);
                        if (ci0 + e >= a.Ci) o[e] = 0.f;
                    }
                    i32x2 v = {(int)pack2bf(o[0], o[1]), (int)pack2bf(o[2], o[3])};
                    *(i32x2*)(a.dx + ((size_t)(n * a.H + iy) * a.W + ix) * a.ldx + a.xoff + ci0) = v;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // row r + 2 has landed (and this wave's stores have left)
        __syncthreads();                                          // everybody is done with rows r, r + 1
    }
}

static bool thin_dgrad_wide_shape(const gcc_conv_t* c) {
    return ceil8(c->Ci) == 8 && c->KH == 4 && c->KW == 4 && c->stride == 2 && c->pad == 1 && !(c->H & 1) && !(c->W & 1) &&
           c->Co > 64 && c->Co <= 128 && ((c->W / 2) & 3) == 0 && (size_t)(c->W / 2 + 4) * THIN_WIDE_RB * 3 + 48 * 1024 <= 156 * 1024 &&
           gcc_opt(GCC_OPT_IGEMM_THIN) >= 1 && gcc_opt(GCC_OPT_IGEMM_THIN) != 2;
}
static bool thin_dgrad_shape(const gcc_conv_t* c) {
    if (thin_dgrad_wide_shape(c)) return true;
    return ceil8(c->Ci) == 8 && c->KH == 4 && c->KW == 4 && c->stride == 2 && c->pad == 1 && !(c->H & 1) && !(c->W & 1) &&
           c->Co >= 16 && c->Co <= 64 && thin_enabled();
}
static int launch_thin_dgrad(const gcc_conv_t* c, const void* dy, const void* wt, void* dx, const gcc_epilogue_t* ep, hipStream_t st) {
    ThinDgradArgs a;
    a.dy = (const bf16_t*)dy; a.wt = (const bf16_t*)wt; a.dx = (bf16_t*)dx;
    a.bias = ep ? ep->bias : nullptr; a.act = ep ? ep->act : GCC_ACT_NONE; a.slope = ep ? ep->slope : 0.f;
    a.N = c->N; a.H = c->H; a.W = c->W; a.Ho = c->H / 2; a.Wo = c->W / 2;
    a.ldy = c->ldy; a.yoff = c->yoff; a.ldx = c->ldx; a.xoff = c->xoff; a.Ci = c->Ci; a.Co8 = ceil8(c->Co);
    const size_t yb = (size_t)c->N * a.Ho * a.Wo * c->ldy * 2, wb = (size_t)c->Ci * 16 * a.Co8 * 2;
    const size_t xb = (size_t)c->N * c->H * c->W * c->ldx * 2;
    if (yb >= OOB || wb >= OOB || xb >= OOB) return -1;
    a.dy_bytes = (uint32_t)yb; a.wt_bytes = (uint32_t)wb;
    a.tiles_per_row = cdiv(a.Wo, 16);
    a.ntiles = c->N * a.Ho * a.tiles_per_row;
    a.dTpr = make_fastdiv(a.tiles_per_row); a.dHh = make_fastdiv(a.Ho);
    if (thin_dgrad_wide_shape(c)) {
        const size_t lds = (size_t)3 * (a.Wo + 4) * THIN_WIDE_RB + 48 * 1024;
        static std::once_flag wide_once;
        std::call_once(wide_once, [] {
            (void)hipFuncSetAttribute((const void*)thin_dgrad_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        });
        // one workgroup per CU (a 3-row ring is 96 KB at Wo = 128): as many workgroups per image as give the chip one round
        int wpi = 256 / c->N;
        if (wpi < 1) wpi = 1;
        if (wpi > a.Ho) wpi = a.Ho;
        const int rows_per_wg = cdiv(a.Ho, wpi);
        wpi = cdiv(a.Ho, rows_per_wg);
        hipLaunchKernelGGL(thin_dgrad_wide_kernel, dim3(c->N * wpi), dim3(THIN_WIDE_NT), lds, st, a, rows_per_wg, wpi);
        GCC_CHECK_LAUNCH();
        return GCC_OK;
    }
    const int ncc = a.Co8 > 32 ? 2 : 1;
    int wgs = cdiv(a.ntiles, 4);
    if (wgs > 768) wgs = 768;
    const dim3 grid(wgs, 2);
    if (ncc == 2) hipLaunchKernelGGL(thin_dgrad_k4s2_kernel<2>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(thin_dgrad_k4s2_kernel<1>, grid, dim3(256), 0, st, a);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

// internal entry (also used by distill.hip): `batch` independent problems, strides in elements
extern "C" int gcc_conv_y2_supported(const gcc_conv_t* c, int dgrad, const gcc_epilogue_t* ep) {
    if (check_conv(c) || dgrad || !ep || !thin_shape(c)) return 0;
    if (ep->stats_partial || ep->act == GCC_ACT_TANH || (ep->act == GCC_ACT_LRELU && (ep->slope < 0.f || ep->slope > 1.f))) return 0;
    if ((ep->ldy2 & 7) || (ep->y2off & 7)) return 0;
    if (ep->y2_mode == 1) return ep->act == GCC_ACT_LRELU || ep->act == GCC_ACT_NONE;      // relu(y) == relu(pre-activation) for these
    if (ep->y2_mode == 2) return ep->y2_gate != nullptr;
    return 0;
}

int gcc_internal_thinout_fprop(const gcc_conv_t* c, const void* x, const void* w, void* y, const gcc_epilogue_t* ep, hipStream_t st);
int gcc_internal_thinout_dgrad(const gcc_conv_t* c, const void* dy, const void* wt, void* dx, const gcc_epilogue_t* ep, hipStream_t st);
// conv_ring3.hip: 3 x 3 stride-1 layers between <= 64-channel tensors at a large spatial size (SRGAN's trunk, VGG19 conv1_2)
int gcc_internal_ring3(const gcc_conv_t* c, int dgrad, const void* src, const void* w, void* dst, const gcc_epilogue_t* ep, hipStream_t st);
int gcc_internal_ring3_rows(const gcc_conv_t* c, int dgrad);
bool gcc_internal_ring3_routed(const gcc_conv_t* c, int dgrad, const gcc_epilogue_t* ep);
int gcc_internal_igemm(const gcc_conv_t* c, int dgrad, const void* src, const void* w, void* dst, const gcc_epilogue_t* ep,
                       int batch, long src_bstride, long wgt_bstride, long dst_bstride, hipStream_t st) {
    GCC_ENTER();
    int rc = check_conv(c);
    if (rc) return rc;
    if (!src || !w || !dst) return GCC_ERR_BAD_ARG;
    const int Ho = gcc_conv_out(c->H, c->KH, c->stride, c->pad);
    const int Wo = gcc_conv_out(c->W, c->KW, c->stride, c->pad);
    if (batch == 1 && head_shape(c) && ep && ep->workspace && !ep->stats_partial && (((uintptr_t)ep->workspace) & 15) == 0) {
        const int taps = c->KH * c->KW;
        if (!dgrad) {
            int ks = 1;
            const size_t need = head_fprop_workspace(c, &ks);
            if (need <= ep->workspace_bytes) {
                // T = x . W16^T as fp32 partial tiles: the fprop packing's row 0 is [taps][Ci8], i.e. a 1x1 weight with `taps` rows
                IgemmParams q;
                q.src = (const bf16_t*)src; q.wgt = (const bf16_t*)w; q.dst = (bf16_t*)dst; q.bias = nullptr; q.stats = nullptr;
                q.act = GCC_ACT_NONE; q.slope = 0.f;
                q.N = c->N; q.KH = 1; q.KW = 1; q.stride = 1; q.pad = 0; q.dgrad = 0;
                q.Hs = c->H; q.Ws = c->W; q.lds_ = c->ldx; q.soff = c->xoff; q.Hd = c->H; q.Wd = c->W; q.ldd = c->ldy; q.doff = c->yoff;
                q.Ct = ceil8(c->Ci); q.Cout = taps; q.ldw = q.Ct;
                const size_t rows = head_rows(c);
                const size_t sb = rows * (size_t)q.lds_ * 2, wb = (size_t)taps * q.ldw * 2;
                if (sb < OOB && wb < OOB) {
                    q.src_bytes = (uint32_t)sb; q.wgt_bytes = (uint32_t)wb;
                    q.src_bstride = q.wgt_bstride = q.dst_bstride = 0;
                    q.ntiles = 1; q.mtiles_max = cdiv((int)rows, 128);
                    const int nk = cdiv(q.Ct, BK);
                    q.ksplit = ks; q.kper = cdiv(nk, ks); q.partial = (float*)ep->workspace; q.rows_max = (int)rows; q.Cpad = 16;
                    q.raw_partial = 1;
                    int rc2 = launch<128, 16>(q, 1, 1, st);
                    if (rc2) return rc2;
                    HeadArgs a = {};
                    a.partial = q.partial; a.ksplit = q.ksplit; a.rows_max = q.rows_max; a.Cpad = 16;
                    a.bias = ep->bias; a.act = ep->act; a.slope = ep->slope;
                    a.dst = (bf16_t*)dst; a.ldd = c->ldy; a.doff = c->yoff;
                    a.N = c->N; a.H = c->H; a.W = c->W; a.Ho = Ho; a.Wo = Wo; a.KH = c->KH; a.KW = c->KW; a.stride = c->stride; a.pad = c->pad;
                    const size_t outs = (size_t)c->N * Ho * Wo;
                    hipLaunchKernelGGL(head_tapsum_kernel, dim3((unsigned)(outs + 15 > 16 * 4096 ? 4096 : (outs + 15) / 16)), dim3(256), 0, st, a);
                    GCC_CHECK_LAUNCH();
                    return GCC_OK;
                }
            }
        }
    }
    if (batch == 1 && dgrad && thin_dgrad_shape(c) && !(ep && ep->stats_partial)) {
        const int rc2 = launch_thin_dgrad(c, src, w, dst, ep, st);
        if (rc2 >= 0) return rc2;
    }
    if (batch == 1) {      // wide kernel, <= 3 output channels (SRGAN's last layer): conv_thinout.hip
        const int rc3 = dgrad ? gcc_internal_thinout_dgrad(c, src, w, dst, ep, st) : gcc_internal_thinout_fprop(c, src, w, dst, ep, st);
        if (rc3 != GCC_ERR_UNSUPPORTED) return rc3;
        const int rc4 = gcc_internal_ring3(c, dgrad, src, w, dst, ep, st);      // its own statistics rows + finalize (gcc_conv_stat_tiles agrees)
        if (rc4 != GCC_ERR_UNSUPPORTED) return rc4;
    }
    if (ep && ep->y2) {          // a second output: the thin forward route only (gcc_conv_y2_supported says so beforehand)
        if (!gcc_conv_y2_supported(c, dgrad, ep)) return GCC_ERR_UNSUPPORTED;
    }
    if (batch == 1 && !dgrad && thin_shape(c) &&
        !(ep && (ep->stats_partial || ep->act == GCC_ACT_TANH || (ep->act == GCC_ACT_LRELU && (ep->slope < 0.f || ep->slope > 1.f))))) {
        const int rc2 = launch_thin(c, src, w, dst, ep, st);
        if (rc2 >= 0) return rc2;
    }
    if (ep && ep->y2) return GCC_ERR_UNSUPPORTED;
    IgemmParams p;
    p.src = (const bf16_t*)src; p.wgt = (const bf16_t*)w; p.dst = (bf16_t*)dst;
    p.bias = ep ? ep->bias : nullptr;
    p.stats = ep ? ep->stats_partial : nullptr;
    p.act = ep ? ep->act : GCC_ACT_NONE;
    p.slope = ep ? ep->slope : 0.f;
    p.N = c->N; p.KH = c->KH; p.KW = c->KW; p.stride = c->stride; p.pad = c->pad; p.dgrad = dgrad;
    size_t src_pix, max_rows;
    if (!dgrad) {
        p.Hs = c->H; p.Ws = c->W; p.lds_ = c->ldx; p.soff = c->xoff;
        p.Hd = Ho; p.Wd = Wo; p.ldd = c->ldy; p.doff = c->yoff;
        p.Ct = ceil8(c->Ci); p.Cout = c->Co;
        max_rows = (size_t)c->N * Ho * Wo;
    } else {
        p.Hs = Ho; p.Ws = Wo; p.lds_ = c->ldy; p.soff = c->yoff;
        p.Hd = c->H; p.Wd = c->W; p.ldd = c->ldx; p.doff = c->xoff;
        p.Ct = ceil8(c->Co); p.Cout = c->Ci;
        const int s = c->stride;
        max_rows = (size_t)c->N * cdiv(c->H, s) * cdiv(c->W, s);
        // the last `pad` rows/cols of dx of a strided conv may receive no contribution when
        // (H + 2*pad - KH) % stride != 0; they still belong to a phase and are written (zeros).
    }
    src_pix = (size_t)p.N * p.Hs * p.Ws;
    p.ldw = c->KH * c->KW * p.Ct;
    const size_t sb = src_pix * (size_t)p.lds_ * 2, wb = (size_t)p.Cout * p.ldw * 2;
    const size_t db = (size_t)p.N * p.Hd * p.Wd * p.ldd * 2;
    if (sb >= OOB || wb >= OOB || db >= (size_t)1 << 32) return GCC_ERR_UNSUPPORTED;
    p.src_bytes = (uint32_t)sb; p.wgt_bytes = (uint32_t)wb;
    p.src_bstride = src_bstride; p.wgt_bstride = wgt_bstride; p.dst_bstride = dst_bstride;
    if (batch < 1 || (batch > 1 && p.stats)) return GCC_ERR_BAD_ARG;
    const int phases = dgrad ? c->stride * c->stride : 1;
    if (dgrad && (c->KH < c->stride || c->KW < c->stride)) return GCC_ERR_UNSUPPORTED;
    const TilePlan tp = select_tile(c->plan, max_rows, p.Cout, phases, conv_nk(c, dgrad), batch);
    // BatchNorm behind this conv (ep->bn): its coefficients are final when this call returns -- folded by the last-arriving
    // workgroups of the launch that writes the statistic rows where the route can (stats_tail), by a gcc_bn_finalize launch
    // otherwise.  Every route writes (or zero-fills) the gcc_conv_stat_tiles() rows the caller allocated.
    const gcc_bn_t* bnf = (ep && ep->bn && p.stats && batch == 1) ? ep->bn : nullptr;
    const int rows_alloc = p.stats ? gcc_conv_stat_tiles(c, dgrad) : 0;
    auto make_tail = [&](int rows, int wgs_per_row, TailFin* f) -> bool {
        *f = TailFin{};
        if (!bnf || !bnf->finalize_in_launch || !bnf->tail_ws || (((uintptr_t)bnf->tail_ws) & 15) || rows != rows_alloc) return false;
        if (tail_ws_bytes(rows, p.Cout) > TAIL_TICKET_BYTES + TAIL_GROUP_BYTES || bnf->tail_ws_bytes < TAIL_TICKET_BYTES + TAIL_GROUP_BYTES) return false;
        f->tickets = (unsigned*)bnf->tail_ws; f->grp = (double*)((char*)bnf->tail_ws + TAIL_TICKET_BYTES);
        f->rows = rows; f->wgs_per_row = wgs_per_row; f->count = bnf->count; f->eps = bnf->eps; f->momentum = bnf->momentum;
        f->gamma = bnf->gamma; f->beta = bnf->beta; f->running_mean = bnf->running_mean; f->running_var = bnf->running_var;
        f->mean = bnf->mean; f->rstd = bnf->rstd; f->scale = bnf->scale; f->shift = bnf->shift;
        return true;
    };
    float* const stats_rows = p.stats;             // (p.stats itself is cleared on the split route)
    auto finalize_after = [&]() -> int {          // the separate launch, over the same rows in the same order
        if (!bnf) return GCC_OK;
        return gcc_bn_finalize(stats_rows, rows_alloc, p.Cout, bnf->count, bnf->gamma, bnf->beta, bnf->eps, bnf->momentum,
                               bnf->running_mean, bnf->running_var, bnf->mean, bnf->rstd, bnf->scale, bnf->shift, (gcc_stream_t)st);
    };
    if (batch == 1) {
        // k4 s2 p1 layers whose geometry fits: the tile's input neighbourhood staged once per 64 channels (conv_halo.hip).  With
        // statistics the tile rows must be the ones gcc_conv_stat_tiles() promised (the 256-pixel plan's).
        const HaloPlan h = halo_plan(c, dgrad);
        if (halo_routed(c->plan, h, dgrad, p.stats != nullptr, tp.BP)) {
            TailFin fin;
            const bool tail = p.stats && make_tail(c->N * h.tiles_x * h.tiles_y, h.ntiles, &fin);
            const int rc2 = launch_halo(c, dgrad, h, src, w, dst, ep, tail ? &fin : nullptr, st);
            if (rc2 >= 0) return (rc2 == GCC_OK && !tail) ? finalize_after() : rc2;
        }
    }
    const int BC = tp.BC;
    p.ntiles = tp.ntiles;
    p.mtiles_max = tp.mtiles;
    // ---- split-K decision (needs caller workspace; without it the launch simply is not split) ------
    p.ksplit = 1; p.kper = 0; p.partial = nullptr; p.rows_max = (int)max_rows; p.Cpad = p.ntiles * BC; p.raw_partial = 0;
    p.pair = 0; p.pair_slab = nullptr; p.pair_flags = nullptr;
    p.debug = gcc_diag_bits();
    if (tp.pair && batch == 1 && ep && ep->workspace && (((uintptr_t)ep->workspace) & 15) == 0) {
        const size_t tiles = (size_t)tp.mtiles * tp.ntiles * phases;
        if (pair_workspace(tiles) <= ep->workspace_bytes) {
            p.pair = 1; p.ksplit = 2; p.kper = cdiv(conv_nk(c, dgrad), 2);
            p.pair_flags = (unsigned int*)ep->workspace;
            p.pair_slab = (float*)((char*)ep->workspace + PAIR_FLAG_BYTES);
            if (gcc_memset_async(ep->workspace, 0, PAIR_FLAG_BYTES, st) != hipSuccess) return GCC_ERR_LAUNCH;
        }
    }
    float* stats_out = p.stats;
    bool fold_stats = false;         // split launch + splitk_fold_stats_kernel (statistics rows: phases x fold_wpp)
    if (batch == 1 && tp.BP == 128 && ep && ep->workspace) {
        SplitPlan sp = plan_ksplit((long)p.mtiles_max * p.ntiles * phases, conv_nk(c, dgrad), tp.max_slices);
        if (sp.ksplit > 4 && stats_out) {
            // a BatchNorm layer's slices are all read back by the fold kernel: cap them at GCC_OPT_FUSE_BN_PARTIAL_KB of partial
            // tiles, never below four (fewer leave the partial-tile launch with a handful of workgroups)
            const size_t per_slice = (size_t)phases * max_rows * p.Cpad * sizeof(float);
            const size_t cap = (size_t)gcc_opt(GCC_OPT_FUSE_BN_PARTIAL_KB) * 1024;
            const int nk = conv_nk(c, dgrad);
            int ks = sp.ksplit;
            while (ks > 4 && per_slice * ks > cap) ks = (ks + 1) / 2;
            if (ks != sp.ksplit) { sp.kper = cdiv(nk, ks); sp.ksplit = cdiv(nk, sp.kper); }
        }
        const size_t need = (size_t)phases * sp.ksplit * max_rows * p.Cpad * sizeof(float);
        if (sp.ksplit > 1 && need <= ep->workspace_bytes && (((uintptr_t)ep->workspace) & 15) == 0) {
            p.ksplit = sp.ksplit; p.kper = sp.kper; p.partial = (float*)ep->workspace;
            p.stats = nullptr;       // statistics are taken while the slices are folded
            fold_stats = stats_out && ceil8(p.Cout) / 8 <= 256 && rows_alloc == fold_wpp(max_rows, phases) * phases;
            if (fold_stats) p.raw_partial = 1;
        }
    }
    TailFin fin_direct;
    bool tail_direct = false;
    if (p.stats) {                   // un-split launch writes mtiles x phases rows itself
        const int rows = p.mtiles_max * phases;
        if (rows != rows_alloc && gcc_memset_async(stats_out, 0, (size_t)rows_alloc * 2 * p.Cout * sizeof(float), st) != hipSuccess)
            return GCC_ERR_LAUNCH;
        tail_direct = make_tail(rows, p.ntiles, &fin_direct);
        if (tail_direct) p.fin = fin_direct;
    }
    if (tp.BP == 256) {
        rc = BC == 256 ? launch<256, 256>(p, phases, batch, st) : launch<256, 128>(p, phases, batch, st);
    } else {
        switch (BC) {
            case 128: rc = launch<128, 128>(p, phases, batch, st); break;
            case 64: rc = launch<128, 64>(p, phases, batch, st); break;
            case 32: rc = launch<128, 32>(p, phases, batch, st); break;
            default: rc = launch<128, 16>(p, phases, batch, st); break;
        }
    }
    if (rc) return rc;
    if (fold_stats) {
        FoldStatsArgs a;
        a.p = p; a.p.stats = stats_out;
        a.wpp = fold_wpp(max_rows, phases);
        a.rpw = (int)((max_rows + a.wpp - 1) / a.wpp);
        const int CH = ceil8(p.Cout) / 8;
        a.CHP = 1; a.sh = 0;
        while (a.CHP < CH) { a.CHP <<= 1; a.sh++; }
        TailFin fin;
        const bool tail = make_tail(a.wpp * phases, 1, &fin);
        a.p.fin = tail ? fin : TailFin{};
        hipLaunchKernelGGL(splitk_fold_stats_kernel, dim3(a.wpp, 1, phases), dim3(256), 0, st, a);
        GCC_CHECK_LAUNCH();
        return tail ? GCC_OK : finalize_after();
    }
    if (p.ksplit > 1 && stats_out && !p.pair) {
        const size_t pixels = (size_t)p.N * p.Hd * p.Wd;
        hipLaunchKernelGGL(channel_stats_kernel, dim3(ceil8(p.Cout) / 8), dim3(256), 0, st, p.dst, p.ldd, p.doff, p.Cout,
                           pixels, stats_out, rows_alloc);
        GCC_CHECK_LAUNCH();
        return finalize_after();
    }
    return tail_direct ? GCC_OK : finalize_after();
}

#ifdef GCC_CLOCK_PROBE
extern "C" int gcc_probe_read(unsigned long long* dst, int clear) {
    hipError_t e = hipMemcpyFromSymbol(dst, HIP_SYMBOL(gcc_igemm::g_clock_probe), sizeof(unsigned long long) * 4096 * 8);
    if (e == hipSuccess && clear) {
        void* sym = nullptr;
        if (hipGetSymbolAddress(&sym, HIP_SYMBOL(gcc_igemm::g_clock_probe)) == hipSuccess) e = hipMemset(sym, 0, sizeof(unsigned long long) * 4096 * 8);
    }
    return e == hipSuccess ? 0 : -1;
}
#endif

// which kernel family a fprop / dgrad call with this geometry and epilogue runs on (same predicates as the dispatch in
// gcc_internal_igemm): 0 igemm_kernel, 1 thin_fprop / thin_dgrad, 2 the single-output-channel head route
bool gcc_internal_thinout_routed(const gcc_conv_t* c, int dgrad, const gcc_epilogue_t* ep);
extern "C" int gcc_conv_route(const gcc_conv_t* c, int dgrad, const gcc_epilogue_t* ep) {
    if (check_conv(c)) return -1;
    if (gcc_internal_thinout_routed(c, dgrad, ep) && !(!dgrad && head_shape(c))) return 3;
    if (gcc_internal_ring3_routed(c, dgrad, ep) && !(!dgrad && head_shape(c)) && !(dgrad && thin_dgrad_shape(c) && !(ep && ep->stats_partial)))
        return 4;
    if (dgrad && thin_dgrad_shape(c) && !(ep && ep->stats_partial)) return 1;
    if (!dgrad && thin_shape(c) &&
        !(ep && (ep->stats_partial || ep->act == GCC_ACT_TANH || (ep->act == GCC_ACT_LRELU && (ep->slope < 0.f || ep->slope > 1.f)))))
        return 1;
    if (!dgrad && head_shape(c) && ep && ep->workspace && !ep->stats_partial && (((uintptr_t)ep->workspace) & 15) == 0 &&
        head_fprop_workspace(c, nullptr) <= ep->workspace_bytes)
        return 2;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// conv + BatchNorm(training) + activation in one call (include/gcc_hip.h gcc_conv_bn_act)
namespace gcc_igemm {
struct FoldBnArgs {
    IgemmParams p;                 // geometry of the split launch (partial, ksplit, rows_max, Cpad, dst = raw output)
    gcc_bn_t bn;
    int act, act2; float slope, drop_p; uint64_t seed;
    bf16_t* y; int ldy, yoff;
    bf16_t* y2; int ldy2, y2off;
    int phases;
};
// one workgroup per 8 output channels, every row of them: fold the K slices (bf16-rounded raw output, kept for the backward
// pass), f64 statistics of the rounded values, finalize (as bn_finalize_kernel), normalise + dropout + activation(s) (as
// bnact_fwd_kernel: same affine form, same dropout counter pix * C + channel, so the backward pass regenerates the mask)
// NT = 1024: the grid is C/8 workgroups (32-64 on these layers) and the walk over rows x K slices is a chain of dependent
// round trips -- 256 threads took 50 us on a 16x16 layer (4096 rows x 8 slices); four times the threads and four slices'
// loads in flight per thread (same summation order) bring the kernel back under the launches it replaces.
template <int NT>
__global__ __launch_bounds__(NT) void splitk_bn_act_kernel(const FoldBnArgs a) {
    __shared__ double red[NT / 64][16];
    __shared__ float coef[16];
    const IgemmParams& p = a.p;
    const int c0 = blockIdx.x * 8;
    const int C = p.Cout;
    const size_t sstride = (size_t)p.rows_max * p.Cpad;
    double s[8], ss[8];
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = ss[j] = 0.0;
    for (int z = 0; z < a.phases; z++) {
        int py = 0, px = 0, Hg, Wg, ostr = 1;
        if (!p.dgrad) { Hg = p.Hd; Wg = p.Wd; }
        else {
            const int st = p.stride;
            py = z / st; px = z % st; ostr = st;
            Hg = (p.Hd - py + st - 1) / st; Wg = (p.Wd - px + st - 1) / st;
        }
        const int M = p.N * Hg * Wg;
        const float* base = p.partial + (size_t)z * p.ksplit * sstride;
        for (int m = threadIdx.x; m < M; m += NT) {
            float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const float* r0 = base + (size_t)m * p.Cpad + c0;
            int sl = 0;
            for (; sl + 4 <= p.ksplit; sl += 4) {
                f32x4 u[4], w[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    u[q] = *(const f32x4*)(r0 + (sl + q) * sstride);
                    w[q] = *(const f32x4*)(r0 + (sl + q) * sstride + 4);
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    v[0] += u[q][0]; v[1] += u[q][1]; v[2] += u[q][2]; v[3] += u[q][3];
                    v[4] += w[q][0]; v[5] += w[q][1]; v[6] += w[q][2]; v[7] += w[q][3];
                }
            }
            for (; sl < p.ksplit; sl++) {
                const f32x4 u = *(const f32x4*)(r0 + sl * sstride), w = *(const f32x4*)(r0 + sl * sstride + 4);
                v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
                v[4] += w[0]; v[5] += w[1]; v[6] += w[2]; v[7] += w[3];
            }
#pragma unroll
            for (int j = 0; j < 8; j++) if (c0 + j >= C) v[j] = 0.f;
            const i32x4 pk = pack8(v);
            float r[8];
            unpack8(pk, r);
#pragma unroll
            for (int j = 0; j < 8; j++) { s[j] += (double)r[j]; ss[j] += (double)r[j] * (double)r[j]; }
            const int n = m / (Hg * Wg);
            const int rr = m - n * (Hg * Wg);
            const int oy = rr / Wg, ox = rr - oy * Wg;
            const size_t pix = (size_t)(n * p.Hd + oy * ostr + py) * p.Wd + (ox * ostr + px);
            *(i32x4*)(p.dst + pix * p.ldd + p.doff + c0) = pk;
        }
    }
    // block reduction: wave shuffles, then the waves through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 8; j++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s[j] += __shfl_xor(s[j], o, 64); ss[j] += __shfl_xor(ss[j], o, 64); }
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) { red[wave][j] = s[j]; red[wave][8 + j] = ss[j]; }
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const int j = threadIdx.x, c = c0 + j;
        float sc = 1.f, sf = 0.f;
        if (c < C) {
            double sum = 0.0, sq = 0.0;
#pragma unroll
            for (int wv = 0; wv < NT / 64; wv++) { sum += red[wv][j]; sq += red[wv][8 + j]; }
            const double mean = sum / a.bn.count;
            double var = sq / a.bn.count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float r = (float)(1.0 / sqrt(var + (double)a.bn.eps));
            const float g = a.bn.gamma ? a.bn.gamma[c] : 1.f, b = a.bn.beta ? a.bn.beta[c] : 0.f;
            if (a.bn.mean) a.bn.mean[c] = (float)mean;
            if (a.bn.rstd) a.bn.rstd[c] = r;
            sc = g * r;
            sf = b - (float)mean * g * r;
            a.bn.scale[c] = sc;
            a.bn.shift[c] = sf;
            if (a.bn.running_mean) a.bn.running_mean[c] = (1.f - a.bn.momentum) * a.bn.running_mean[c] + a.bn.momentum * (float)mean;
            if (a.bn.running_var) {
                const double unb = a.bn.count > 1.0 ? var * a.bn.count / (a.bn.count - 1.0) : var;
                a.bn.running_var[c] = (1.f - a.bn.momentum) * a.bn.running_var[c] + a.bn.momentum * (float)unb;
            }
        }
        coef[j] = sc; coef[8 + j] = sf;
    }
    __syncthreads();
    float sc[8], sf[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { sc[j] = coef[j]; sf[j] = coef[8 + j]; }
    const float keep_scale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const size_t pixels = (size_t)p.N * p.Hd * p.Wd;
    // second pass over the raw output: every thread re-reads exactly the pixels it wrote itself
    for (int z = 0; z < a.phases; z++) {
        int py = 0, px = 0, Hg, Wg, ostr = 1;
        if (!p.dgrad) { Hg = p.Hd; Wg = p.Wd; }
        else {
            const int st = p.stride;
            py = z / st; px = z % st; ostr = st;
            Hg = (p.Hd - py + st - 1) / st; Wg = (p.Wd - px + st - 1) / st;
        }
        const int M = p.N * Hg * Wg;
        for (int m = threadIdx.x; m < M; m += NT) {
            const int n = m / (Hg * Wg);
            const int rr = m - n * (Hg * Wg);
            const int oy = rr / Wg, ox = rr - oy * Wg;
            const size_t pix = (size_t)(n * p.Hd + oy * ostr + py) * p.Wd + (ox * ostr + px);
            float v[8], o1[8], o2[8];
            unpack8(*(const i32x4*)(p.dst + pix * p.ldd + p.doff + c0), v);
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = v[j] * sc[j] + sf[j];
            if (a.drop_p > 0.f) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float u = rng_uniform(a.seed, pix * (size_t)C + c0 + j);
                    v[j] = u >= a.drop_p ? v[j] * keep_scale : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) if (c0 + j >= C) v[j] = 0.f;       // pad channels stay exact zeros
            apply_act8(v, o1, a.act, a.slope);
            if (a.y) *(i32x4*)(a.y + pix * a.ldy + a.yoff + c0) = pack8(o1);
            if (a.y2) {
                apply_act8(v, o2, a.act2, a.slope);
                *(i32x4*)(a.y2 + pix * a.ldy2 + a.y2off + c0) = pack8(o2);
            }
        }
    }
    (void)pixels;
}
constexpr size_t FOLD_BN_MAX_ROWS = 4096;      // rows (all phases) a 256-thread workgroup walks twice
}  // namespace gcc_igemm

static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
extern "C" size_t gcc_conv_workspace(const gcc_conv_t* c, int dgrad);
extern "C" int gcc_conv_stat_tiles(const gcc_conv_t* c, int dgrad);
extern "C" size_t gcc_conv_bn_act_workspace(const gcc_conv_t* c, int dgrad) {
    if (check_conv(c)) return 0;
    const int Cout = dgrad ? c->Ci : c->Co;
    return al256(gcc_conv_workspace(c, dgrad)) + al256((size_t)gcc_conv_stat_tiles(c, dgrad) * 2 * Cout * sizeof(float));
}

extern "C" int gcc_conv_bn_act(const gcc_conv_t* c, int dgrad, const void* x, const void* w, void* y_raw, const gcc_bn_t* bn,
                               const gcc_bnact_t* act, void* y, int ldy, int yoff, void* y2, int ldy2, int y2off, void* ws,
                               size_t ws_bytes, gcc_stream_t stream) {
    GCC_ENTER();
    int rc = check_conv(c);
    if (rc) return rc;
    if (!x || !w || !y_raw || !bn || !act || (!y && !y2) || !bn->scale || !bn->shift || bn->count <= 0 || !ws) return GCC_ERR_BAD_ARG;
    if ((y && ((ldy & 7) || (yoff & 7))) || (y2 && ((ldy2 & 7) || (y2off & 7)))) return GCC_ERR_BAD_ARG;
    if (ws_bytes < gcc_conv_bn_act_workspace(c, dgrad) || (((uintptr_t)ws) & 15)) return GCC_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int Cout = dgrad ? c->Ci : c->Co;
    const int phases = dgrad ? c->stride * c->stride : 1;
    const size_t split_bytes = al256(gcc_conv_workspace(c, dgrad));
    float* stats = (float*)((char*)ws + split_bytes);
    const int Ho = gcc_conv_out(c->H, c->KH, c->stride, c->pad), Wo = gcc_conv_out(c->W, c->KW, c->stride, c->pad);
    const size_t out_pixels = dgrad ? (size_t)c->N * c->H * c->W : (size_t)c->N * Ho * Wo;
    const int out_ld = dgrad ? c->ldx : c->ldy, out_off = dgrad ? c->xoff : c->yoff;
    // which plan would the conv take?  (same predicates as gcc_internal_igemm)
    const size_t max_rows = conv_max_rows(c, dgrad);
    const int nk = conv_nk(c, dgrad);
    const TilePlan tp = select_tile(c->plan, max_rows, Cout, phases, nk, 1);
    SplitPlan sp = tp.BP == 128 ? plan_ksplit((long)tp.mtiles * tp.ntiles * phases, nk, tp.max_slices) : SplitPlan{1, nk};
    // The K split is sized to fill the chip with the partial-tile launch, but every slice is one more fp32 copy of the output
    // that splitk_bn_act_kernel -- C / 8 workgroups -- has to read back: the student's 16x16 -> 8x8 layer wrote and folded 16
    // slices = 16.8 MB for a 0.5 MB output, 59 us of fold behind 11 us of MFMA work (profiles/r3z_unet_student_chain.txt).
    // Cap the slices at GCC_OPT_FUSE_BN_PARTIAL_KB of partial tiles (default 4 MB), never below four (fewer leave the partial-tile
    // launch with a handful of workgroups: student forward 628 -> 569 us, teacher 836 -> 796 with 8 MB: profiles/r3z_*).
    if (sp.ksplit > 4 && max_rows * phases <= FOLD_BN_MAX_ROWS) {
        const size_t per_slice = (size_t)phases * max_rows * tp.ntiles * tp.BC * sizeof(float);
        const size_t cap = (size_t)gcc_opt(GCC_OPT_FUSE_BN_PARTIAL_KB) * 1024;
        int ks = sp.ksplit;
        while (ks > 4 && per_slice * ks > cap) ks = (ks + 1) / 2;
        if (ks != sp.ksplit) { sp.kper = cdiv(nk, ks); sp.ksplit = cdiv(nk, sp.kper); }
    }
    const bool routed = (dgrad && thin_dgrad_shape(c)) || (!dgrad && thin_shape(c)) || head_shape(c);
    const int fuse = gcc_opt(GCC_OPT_FUSE_BN);
    const bool grid_ws = bn->tail_ws && bn->tail_ws_bytes >= GCC_TAIL_WORKSPACE_BYTES && (((uintptr_t)bn->tail_ws) & 15) == 0;
    if (!routed && sp.ksplit > 1 && max_rows * phases <= FOLD_BN_MAX_ROWS && (fuse == 1 || fuse == 3)) {
        // split launch with raw partial tiles, then the fused fold + statistics + finalize + normalise kernel
        FoldBnArgs a;
        IgemmParams& p = a.p;
        p.src = (const bf16_t*)x; p.wgt = (const bf16_t*)w; p.dst = (bf16_t*)y_raw; p.bias = nullptr; p.stats = nullptr;
        p.act = GCC_ACT_NONE; p.slope = 0.f;
        p.N = c->N; p.KH = c->KH; p.KW = c->KW; p.stride = c->stride; p.pad = c->pad; p.dgrad = dgrad;
        if (!dgrad) {
            p.Hs = c->H; p.Ws = c->W; p.lds_ = c->ldx; p.soff = c->xoff; p.Hd = Ho; p.Wd = Wo; p.ldd = c->ldy; p.doff = c->yoff;
            p.Ct = ceil8(c->Ci); p.Cout = c->Co;
        } else {
            p.Hs = Ho; p.Ws = Wo; p.lds_ = c->ldy; p.soff = c->yoff; p.Hd = c->H; p.Wd = c->W; p.ldd = c->ldx; p.doff = c->xoff;
            p.Ct = ceil8(c->Co); p.Cout = c->Ci;
            if (c->KH < c->stride || c->KW < c->stride) return GCC_ERR_UNSUPPORTED;
        }
        p.ldw = c->KH * c->KW * p.Ct;
        const size_t sb = (size_t)p.N * p.Hs * p.Ws * p.lds_ * 2, wb = (size_t)p.Cout * p.ldw * 2;
        if (sb >= OOB || wb >= OOB) return GCC_ERR_UNSUPPORTED;
        p.src_bytes = (uint32_t)sb; p.wgt_bytes = (uint32_t)wb;
        p.src_bstride = p.wgt_bstride = p.dst_bstride = 0;
        p.ntiles = tp.ntiles; p.mtiles_max = tp.mtiles;
        p.ksplit = sp.ksplit; p.kper = sp.kper; p.partial = (float*)ws; p.rows_max = (int)max_rows; p.Cpad = tp.ntiles * tp.BC;
        p.raw_partial = 1; p.pair = 0; p.pair_slab = nullptr; p.pair_flags = nullptr;
        if ((size_t)phases * sp.ksplit * max_rows * p.Cpad * sizeof(float) > split_bytes) return GCC_ERR_WORKSPACE;
        switch (tp.BC) {
            case 128: rc = launch<128, 128>(p, phases, 1, st); break;
            case 64: rc = launch<128, 64>(p, phases, 1, st); break;
            case 32: rc = launch<128, 32>(p, phases, 1, st); break;
            default: rc = launch<128, 16>(p, phases, 1, st); break;
        }
        if (rc) return rc;
        if (fuse == 3 && grid_ws) {
            // the slices folded, the statistics exchanged and the rows normalised by ONE kernel on the whole chip
            BnFoldDesc d = {};
            d.part = p.partial; d.ksplit = p.ksplit; d.rows_max = p.rows_max; d.Cpad = p.Cpad; d.phases = phases;
            d.N = p.N; d.Hd = p.Hd; d.Wd = p.Wd; d.stride = c->stride; d.dgrad = dgrad;
            d.raw = y_raw; d.ldraw = p.ldd; d.rawoff = p.doff;
            d.y = y; d.ldy = ldy; d.yoff = yoff; d.y2 = y2; d.ldy2 = ldy2; d.y2off = y2off;
            d.C = Cout; d.bn = *bn; d.act = act->act; d.act2 = act->act2; d.slope = act->slope; d.drop_p = act->drop_p; d.seed = act->seed;
            d.ws = (char*)bn->tail_ws + TAIL_TICKET_BYTES + TAIL_GROUP_BYTES; d.ws_bytes = GCC_INORM_WORKSPACE_BYTES;
            const int rc3 = gcc_internal_bn_fold_grid(&d, st);
            if (rc3 != GCC_ERR_UNSUPPORTED) return rc3;
        }
        a.bn = *bn; a.act = act->act; a.act2 = act->act2; a.slope = act->slope; a.drop_p = act->drop_p; a.seed = act->seed;
        a.y = (bf16_t*)y; a.ldy = ldy; a.yoff = yoff; a.y2 = (bf16_t*)y2; a.ldy2 = ldy2; a.y2off = y2off; a.phases = phases;
        hipLaunchKernelGGL(splitk_bn_act_kernel<1024>, dim3(ceil8(Cout) / 8), dim3(1024), 0, st, a);
        GCC_CHECK_LAUNCH();
        return GCC_OK;
    }
    // default (GCC_OPT_FUSE_BN 2, round 4): the conv with statistics in its epilogue -- a split layer: partial tiles, then
    // splitk_fold_stats_kernel on the whole chip -- finalized by the last-arriving workgroups of the launch that wrote the rows
    // (bn->tail_ws; else a gcc_bn_finalize launch), then the normalising pass: two launches per layer, three on split layers.
    // GCC_OPT_FUSE_BN 0: the same without the in-launch finalize.
    gcc_bn_t bnf = *bn;
    if (!gcc_opt(GCC_OPT_FUSE_BN)) { bnf.tail_ws = nullptr; bnf.tail_ws_bytes = 0; }
    gcc_epilogue_t ep = {nullptr, GCC_ACT_NONE, 0.f, stats, split_bytes ? ws : nullptr, split_bytes, &bnf, nullptr, 0, 0, 0, nullptr};
    rc = gcc_internal_igemm(c, dgrad, x, w, y_raw, &ep, 1, 0, 0, 0, st);
    if (rc) return rc;
    gcc_bnact_t q = *act;
    q.scale = bn->scale; q.shift = bn->shift; q.gate = nullptr; q.gate_after_act = 0; q.groups = 1; q.residual = nullptr; q.ld_residual = 0;
    return gcc_bnact_fwd(&q, y_raw, out_ld, out_off, y, ldy, yoff, y2, ldy2, y2off, Cout, out_pixels, stream);
}

extern "C" size_t gcc_conv_workspace(const gcc_conv_t* c, int dgrad) {
    if (check_conv(c)) return 0;
    if (head_shape(c) && !dgrad) return head_fprop_workspace(c, nullptr);
    const int phases = dgrad ? c->stride * c->stride : 1;
    const size_t max_rows = conv_max_rows(c, dgrad);
    const int nk = conv_nk(c, dgrad);
    const TilePlan tp = select_tile(c->plan, max_rows, dgrad ? c->Ci : c->Co, phases, nk, 1);
    if (tp.pair) return pair_workspace((size_t)tp.mtiles * tp.ntiles * phases);
    if (tp.BP != 128) return 0;
    const SplitPlan sp = plan_ksplit((long)tp.mtiles * tp.ntiles * phases, nk, tp.max_slices);
    if (sp.ksplit <= 1) return 0;
    return (size_t)phases * sp.ksplit * max_rows * tp.ntiles * tp.BC * sizeof(float);
}

extern "C" int gcc_conv_tile(const gcc_conv_t* c, int dgrad) {
    if (check_conv(c)) return 0;
    const int phases = dgrad ? c->stride * c->stride : 1;
    const TilePlan tp = select_tile(c->plan, conv_max_rows(c, dgrad), dgrad ? c->Ci : c->Co, phases, conv_nk(c, dgrad), 1);
    return tp.BP * 1000 + tp.BC;
}

extern "C" int gcc_conv_stat_tiles(const gcc_conv_t* c, int dgrad) {
    GCC_ENTER();
    if (check_conv(c)) return 0;
    if (const int r3 = gcc_internal_ring3_rows(c, dgrad)) return r3;      // one row per workgroup of the ring-walk route (conv_ring3.hip)
    const int phases = dgrad ? c->stride * c->stride : 1;
    const TilePlan tp = select_tile(c->plan, conv_max_rows(c, dgrad), dgrad ? c->Ci : c->Co, phases, conv_nk(c, dgrad), 1);
    const HaloPlan h = halo_plan(c, dgrad);
    if (halo_routed(c->plan, h, dgrad, true, tp.BP)) return c->N * h.tiles_x * h.tiles_y;
    if (tp.BP == 128 && ceil8(dgrad ? c->Ci : c->Co) / 8 <= 256) {
        // layers whose K loop is split over workgroups: the rows of splitk_fold_stats_kernel (an un-split fallback zero-fills them)
        const SplitPlan sp = plan_ksplit((long)tp.mtiles * tp.ntiles * phases, conv_nk(c, dgrad), tp.max_slices);
        if (sp.ksplit > 1) return fold_wpp(conv_max_rows(c, dgrad), phases) * phases;
    }
    return tp.mtiles * phases;
}

extern "C" int gcc_conv_fprop(const gcc_conv_t* c, const void* x, const void* w, void* y,
                              const gcc_epilogue_t* ep, gcc_stream_t stream) {
    GCC_ENTER();
    return gcc_internal_igemm(c, 0, x, w, y, ep, 1, 0, 0, 0, (hipStream_t)stream);
}

extern "C" int gcc_conv_dgrad(const gcc_conv_t* c, const void* dy, const void* wt, void* dx,
                              const gcc_epilogue_t* ep, gcc_stream_t stream) {
    GCC_ENTER();
    return gcc_internal_igemm(c, 1, dy, wt, dx, ep, 1, 0, 0, 0, (hipStream_t)stream);
}
