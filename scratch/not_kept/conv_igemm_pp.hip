// 256 x 256 implicit-GEMM convolution tile with a two-group ("ping-pong") schedule for gfx950.
//
// Same GEMM view, LDS images, LDS-DMA staging and epilogue as igemm_kernel<256, 256> (conv_igemm.hip), different main
// loop.  There, all 8 waves of the workgroup walk a k-step in lock step: after the per-step barrier every wave first
// reads its fragments from LDS (nothing to multiply yet), then every wave multiplies (LDS idle) -- measured 2.0 us per
// 64-deep k-step against 1.0 us of MFMA work.  Here the waves form two groups (waves 0-3 / 4-7: one wave of each group
// per SIMD) that run the same program offset by one barrier: a k-step is 4 phases, each phase = [LDS fragment reads +
// 2 LDS-DMA instructions] barrier [16 MFMAs on one 64-channel x 32-pixel quadrant of the wave's 128 x 64 block] barrier,
// so one group's MFMA half-phase is the other group's load half-phase and each SIMD's matrix pipe always has a wave
// with operands in registers (the schedule of the guide's 256^2 8-phase GEMM, cdna_hip_programming.md section 5).
//
// Staging runs ahead by slot: a k-step's 64 KB (pixels P0 | P1, weights W0 | W1: four 128-row half-tiles) are staged one
// half-tile per phase, by all 8 waves, two 1-KiB LDS-DMA instructions each:
//     phase 1 of step t: P1(t+1) -> other stage     phase 3: W1(t+1) -> other stage
//     phase 2          : W0(t+1) -> other stage     phase 4: P0(t+2) -> this step's own stage, then s_waitcnt vmcnt(2)
// Fragment reads of step t: phase 1 pixels B0 + weights A0, phase 2 B1, phase 3 A1 (phase 4 multiplies A1 x B0 from
// registers), so the pixel half-tiles of a stage are last read in phase 2 and the weight half-tiles in phase 3.
// Hazards (barrier b_k = k-th release; group 0 runs phase p between b_{2p-2} and b_{2p}, group 1 one barrier later):
//   WAR  a slot read in phase p is re-staged in phase >= p + 2 (group 1's reads of phase p retire before b_{2p+1}; group
//        0 issues the DMA of phase p + 2 after b_{2p+2}).
//   RAW  the DMA of step t+1 is waited for (counted vmcnt, never 0 inside the loop) in phase 4 of step t by the issuing
//        wave, which then passes a barrier before any wave issues the reads of phase 1 of step t+1.
// The weight half-tiles (re-read by every pixel tile of the launch: L2 resident) take the short lead (1-2 phases), the
// pixel half-tiles, which come from HBM / Infinity Cache, 4-5 phases.
#include <mutex>
#include "common.hpp"
#include "igemm_common.hpp"

namespace gcc_igemm {

__global__ __launch_bounds__(512) void igemm_pp_kernel(const IgemmParams p) {
    using C = Cfg<256, 256>;
    constexpr int BP = 256, BC = 256;
    constexpr int STAGE = BP * BK * 2;                    // bytes of one operand tile of one stage (32 KiB)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                                      // pixels  [2][256][128 B]
    char* sW = smem + 2 * STAGE;                          // weights [2][256][128 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave % C::WC;                          // 128-channel half of the tile
    const int wp = wave / C::WC;                          // 64-pixel quarter of the tile
    const int group = wave >> 2;

    // ---- per-phase geometry (as igemm_kernel) -------------------------------------------------
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
    const int nk = (TA * TB * p.Ct + BK - 1) / BK;

    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = tile / p.ntiles;
    const int nt = tile % p.ntiles;
    const int m0 = mt * BP;
    const int n0 = nt * BC;
    if (m0 >= M) {         // smaller phase (odd sizes): uniform exit, no barrier reached yet
        if (p.stats && tid < BC && n0 + tid < p.Cout) {
            const int trow = blockIdx.z * p.mtiles_max + mt;
            p.stats[((size_t)trow * 2 + 0) * p.Cout + n0 + tid] = 0.f;
            p.stats[((size_t)trow * 2 + 1) * p.Cout + n0 + tid] = 0.f;
        }
        return;
    }
    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wgt = __builtin_amdgcn_make_buffer_rsrc((void*)p.wgt, 0, p.wgt_bytes, 0x00020000);

    // ---- staging rows of this lane: half-tile h (rows 128 h ..), instruction wave * 2 + i covers 8 rows, lane -> row +
    //      (lane >> 3), physical 16-byte chunk lane & 7 holding logical chunk (lane & 7) ^ (row & 7)
    const int chunk = (lane & 7) ^ (lane >> 3);
    int a_off[4], a_iy[4], a_ix[4], w_off[4];
    bool w_ok[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int rloc = (q >> 1) * 128 + (wave * 2 + (q & 1)) * 8 + (lane >> 3);
        const int m = m0 + rloc;
        if (m < M) {
            const int n = m / (Hg * Wg);
            const int r = m - n * (Hg * Wg);
            const int oy = r / Wg;
            const int ox = r - oy * Wg;
            a_iy[q] = oy * sy + dy0;
            a_ix[q] = ox * sy + dx0;
            a_off[q] = (((n * p.Hs + a_iy[q]) * p.Ws + a_ix[q]) * p.lds_ + p.soff) * 2 + chunk * 16;
        } else {
            a_off[q] = 0; a_iy[q] = -(1 << 28); a_ix[q] = 0;   // always out of range -> zeros
        }
        const int row = n0 + rloc;
        w_ok[q] = row < p.Cout;
        w_off[q] = row * p.ldw * 2 + chunk * 16;
    }
    const int tap_row_bytes = p.Ws * p.lds_ * 2 * dstep;
    const int tap_col_bytes = p.lds_ * 2 * dstep;

    // K walk (one tap per Ct / 64 k-steps: wave-uniform scalar state).  Two cursors: P0 is staged a phase ahead of the
    // other three half-tiles of its step.
    struct KPos { int ta, tb, cc; };
    auto advance = [&](KPos& k) {
        k.cc += BK;
        if (k.cc == p.Ct) { k.cc = 0; if (++k.tb == TB) { k.tb = 0; ++k.ta; } }
    };
    auto stage_P = [&](int h, int st, const KPos& k) {
        const bool kval = k.ta < TA;
        const int dyo = k.ta * dstep, dxo = k.tb * dstep;
        const int pix_off = k.ta * tap_row_bytes + k.tb * tap_col_bytes + k.cc * 2;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int q = h * 2 + i;
            const bool ok = kval && (unsigned)(a_iy[q] + dyo) < (unsigned)p.Hs && (unsigned)(a_ix[q] + dxo) < (unsigned)p.Ws;
            const uint32_t off = ok ? (uint32_t)(a_off[q] + pix_off) : OOB;
            char* dst = sA + st * STAGE + h * (STAGE / 2) + (wave * 2 + i) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, LDS_PTR(void, dst), 16, off, 0, 0, 0);
        }
    };
    auto stage_W = [&](int h, int st, const KPos& k) {
        const bool kval = k.ta < TA;
        const int wt_off = (((kh0 + k.ta * kstep) * p.KW + (kw0 + k.tb * kstep)) * p.Ct + k.cc) * 2;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int q = h * 2 + i;
            const uint32_t off = (kval && w_ok[q]) ? (uint32_t)(w_off[q] + wt_off) : OOB;
            char* dst = sW + st * STAGE + h * (STAGE / 2) + (wave * 2 + i) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, LDS_PTR(void, dst), 16, off, 0, 0, 0);
        }
    };

    f32x4 acc[C::CB][C::PB];
#pragma unroll
    for (int i = 0; i < C::CB; i++)
#pragma unroll
        for (int j = 0; j < C::PB; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int lr = lane & 15;
    const int lq = lane >> 4;
    // fragment addresses: row = base + 16 i + lr, so row & 7 = lr & 7 and the swizzled chunk is a per-lane constant per k-slice
    const int ck0 = ((0 + lq) ^ (lr & 7)) << 4, ck1 = ((4 + lq) ^ (lr & 7)) << 4;
    const int w_row0 = (wc * C::TC + lr) * 128, p_row0 = (wp * C::TP + lr) * 128;
    bf16x8 fw[2][4], fp[2][2][2];          // weights: [k-slice][fragment of the current 64-channel half]; pixels: [32-pixel half][k-slice][fragment]

    auto read_W = [&](int a, int st) {
        const char* w = sW + st * STAGE + w_row0 + a * (64 * 128);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            fw[0][i] = *(const bf16x8*)(w + i * 2048 + ck0);
            fw[1][i] = *(const bf16x8*)(w + i * 2048 + ck1);
        }
    };
    auto read_P = [&](int b, int st) {
        const char* a = sA + st * STAGE + p_row0 + b * (32 * 128);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            fp[b][0][j] = *(const bf16x8*)(a + j * 2048 + ck0);
            fp[b][1][j] = *(const bf16x8*)(a + j * 2048 + ck1);
        }
    };
    // the MFMA half of a phase: [barrier] wait for this phase's fragment reads, 16 MFMAs on quadrant (a, b), [barrier]
    auto mfma_quadrant = [&](int a, int b) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[a * 4 + i][b * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ks][i], fp[b][ks][j], acc[a * 4 + i][b * 2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: step 0 entirely, P0 of step 1 ---------------------------------------------------
    KPos kR = {0, 0, 0}, kP0;
    stage_P(0, 0, kR); stage_P(1, 0, kR); stage_W(0, 0, kR); stage_W(1, 0, kR);
    advance(kR);                        // kR: step 1 (P1 / W0 / W1 of the next step)
    stage_P(0, 1, kR);
    kP0 = kR; advance(kP0);             // kP0: step 2
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();       // step 0 has landed for every wave
    if (group == 1) __builtin_amdgcn_s_barrier();       // stagger: group 1 runs one barrier behind group 0

    for (int t = 0; t < nk; t++) {
        const int s = t & 1;
        // phase 1
        read_P(0, s); read_W(0, s);
        stage_P(1, s ^ 1, kR);
        mfma_quadrant(0, 0);
        // phase 2
        read_P(1, s);
        stage_W(0, s ^ 1, kR);
        mfma_quadrant(0, 1);
        // phase 3
        read_W(1, s);
        stage_W(1, s ^ 1, kR);
        advance(kR);
        mfma_quadrant(1, 1);
        // phase 4: no fragment reads; P0 of step t + 2 into this step's own stage; step t + 1 must have landed
        stage_P(0, s, kP0);
        advance(kP0);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        mfma_quadrant(1, 0);
    }
    if (group == 0) __builtin_amdgcn_s_barrier();       // balance the stagger
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the zero-fill stagings past the end of K target LDS the epilogue reuses
    __syncthreads();

    igemm_epilogue<C, BP, BC>(p, acc, smem, tid, lr, lq, wc, wp, m0, n0, M, Hg, Wg, ostr, py, px, mt, 0, p.dst);
}

int launch_igemm_pp(const IgemmParams& p, int phases, hipStream_t st) {
    using C = Cfg<256, 256>;
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        hipFuncSetAttribute((const void*)igemm_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    });
    dim3 grid(p.mtiles_max * p.ntiles, 1, phases);
    hipLaunchKernelGGL(igemm_pp_kernel, grid, dim3(C::NT), C::LDS_BYTES, st, p);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

}  // namespace gcc_igemm
