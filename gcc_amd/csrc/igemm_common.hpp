// Shared pieces of the implicit-GEMM convolution kernels (conv_igemm.hip, conv_igemm_pp.hip): launch parameters, tile
// constants and the epilogue (bias + activation + bf16 rounding, NHWC store through an LDS transpose, BatchNorm partial sums).
#pragma once
#include "common.hpp"

// named (not anonymous) namespace: hipcc fails to emit the host stub of a kernel template with internal
// linkage whose body holds lambdas inside an `if constexpr` branch
namespace gcc_igemm {

// BatchNorm finalize inside the producing launch (round 4): the workgroups that write the last row of a group of FIN_GROUP
// statistic rows fold that group, the one that completes the last group folds the group sums and finalizes -- the canonical
// order of common.hpp, so the coefficients are bit for bit those of a gcc_bn_finalize launch over the same rows, which this
// replaces (120 launches of ~5 us per Pix2Pix iteration, each a dependent hop on a chain of small kernels).  No workgroup waits
// for another: arrival is counted with returning agent-scope atomics (tickets), the workgroup told "you are last" does the
// fold.  Hand-off as in MI355X_MICROARCH.md, 'Valid forms' table row 1: every handed-off value is stored sc1 (write-through),
// every storing wave drains (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane adds to the counter; the workgroup whose add
// came last loads them sc1 behind a workgroup barrier.  The ticket words start zero (caller-provided workspace, zero-filled
// once) and the last arriver leaves them zero; the workspace belongs to ONE stream (its launches are ordered).
struct TailFin {
    unsigned* tickets;      // NULL: off.  [groups + 1]
    double* grp;            // [groups][2][Cout] group sums
    int rows;               // statistic rows of the launch (tile rows x phases)
    int wgs_per_row;        // workgroups that contribute to one row (column tiles)
    double count;
    float eps, momentum;
    const float* gamma; const float* beta;
    float* running_mean; float* running_var; float* mean; float* rstd; float* scale; float* shift;
};
constexpr size_t TAIL_TICKET_BYTES = 4096;                 // up to 1023 groups = 16368 statistic rows
constexpr size_t TAIL_GROUP_BYTES = (size_t)4 << 20;         // group sums; behind them: the grid fold kernel's workspace (GCC_TAIL_WORKSPACE_BYTES)
static inline size_t tail_ws_bytes(int rows, int Cout) {
    const size_t G = (size_t)(rows + FIN_GROUP - 1) / FIN_GROUP;
    return G + 1 > TAIL_TICKET_BYTES / 4 ? ~(size_t)0 : TAIL_TICKET_BYTES + G * 2 * (size_t)Cout * sizeof(double);
}
// called by EVERY thread of a workgroup after the workgroup's statistic row `trow` has been stored (sc1); sh: one LDS int
// nobody else touches until the call returns.  NT threads.
template <int NT>
__device__ __forceinline__ void stats_tail(const TailFin& f, const float* stats, int Cout, int trow, int* sh, int tid) {
    typedef __attribute__((address_space(1))) unsigned int gu32;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int g = trow / FIN_GROUP;
    const int G = (f.rows + FIN_GROUP - 1) / FIN_GROUP;
    const int nr = min(FIN_GROUP, f.rows - g * FIN_GROUP);
    if (tid == 0) sh[0] = (int)__hip_atomic_fetch_add((gu32*)f.tickets + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const bool group_last = sh[0] == nr * f.wgs_per_row - 1;
    __syncthreads();
    if (!group_last) return;
    // (sc1 loads through buffer intrinsics: the compiler keeps them in flight together; scoped atomic loads were waited for one
    // by one -- a 16-deep chain of round trips per fold, 9-21 us per conv launch: profiles/r4c_*)
    const __amdgpu_buffer_rsrc_t rs_st = __builtin_amdgcn_make_buffer_rsrc((void*)stats, 0, (unsigned)((size_t)f.rows * 2 * Cout * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_gr = __builtin_amdgcn_make_buffer_rsrc((void*)f.grp, 0, (unsigned)((size_t)G * 2 * Cout * 8), 0x00020000);
    for (int idx = tid; idx < 2 * Cout; idx += NT) {
        const int w = idx >= Cout ? 1 : 0, c = idx - w * Cout;
        float v[FIN_GROUP];
        // (unconditional loads, rows past the group clamped: a `r < nr ? load : 0` compiles to sixteen branches with a wait each)
#pragma unroll
        for (int r = 0; r < FIN_GROUP; r++)
            v[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_st, (((g * FIN_GROUP + (r < nr ? r : nr - 1)) * 2 + w) * Cout + c) * 4, 0, 16));
        double sum = 0.0;
#pragma unroll
        for (int r = 0; r < FIN_GROUP; r++)
            if (r < nr) sum += (double)v[r];
        __hip_atomic_store(f.grp + ((size_t)g * 2 + w) * Cout + c, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) sh[0] = (int)__hip_atomic_fetch_add((gu32*)f.tickets + G, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const bool last = sh[0] == G - 1;
    __syncthreads();
    if (!last) return;
    for (int c = tid; c < Cout; c += NT) {
        double s = 0.0, ss = 0.0;
        for (int q0 = 0; q0 < G; q0 += 16) {                // sixteen groups' loads in flight, added in ascending order
            i32x2 a[16], b[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int q = q0 + u < G ? q0 + u : G - 1;
                a[u] = __builtin_amdgcn_raw_buffer_load_b64(rs_gr, ((q * 2 + 0) * Cout + c) * 8, 0, 16);
                b[u] = __builtin_amdgcn_raw_buffer_load_b64(rs_gr, ((q * 2 + 1) * Cout + c) * 8, 0, 16);
            }
#pragma unroll
            for (int u = 0; u < 16; u++)
                if (q0 + u < G) { s += __builtin_bit_cast(double, a[u]); ss += __builtin_bit_cast(double, b[u]); }
        }
        bn_channel_finalize(s, ss, f.count, f.eps, f.momentum, f.gamma, f.beta, c, f.running_mean, f.running_var, f.mean, f.rstd,
                            f.scale, f.shift);
    }
    for (int i = tid; i <= G; i += NT) __hip_atomic_store((gu32*)f.tickets + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct IgemmParams {
    const bf16_t* src;   // gather source (x for fprop, dy for dgrad)
    const bf16_t* wgt;   // packed weights, rows = output channels of this GEMM
    bf16_t* dst;
    const float* bias;
    float* stats;        // [tiles][2][Cout] or null
    int N;
    int Hs, Ws, lds_, soff;    // source spatial dims, pixel stride, channel offset
    int Hd, Wd, ldd, doff;     // destination tensor dims
    int Ct;                    // channels per tap (multiple of 8)
    int Cout;                  // GEMM rows (logical output channels)
    int KH, KW, stride, pad;
    int dgrad;                 // 0: fprop gather, 1: backward-data gather (phases = stride^2)
    int ldw;                   // weight row stride (elements) = KH*KW*Ct
    int act;
    float slope;
    uint32_t src_bytes, wgt_bytes;
    int mtiles_max;            // M tiles of the largest phase (grid sizing / stats rows per phase)
    int ntiles;
    // optional batch of independent problems on blockIdx.y (per-image 1x1 products of the gram loss)
    long src_bstride, dst_bstride, wgt_bstride;   // elements
    // split-K (small grids: U-Net bottleneck, 1-channel PatchGAN head): blockIdx.y = K slice, fp32
    // partial tiles go to `partial` [phase][slice][rows_max][Cpad]; splitk_epilogue_kernel finishes
    int ksplit, kper;
    float* partial;
    int rows_max, Cpad;
    int raw_partial;     // leave the fp32 partial tiles to the caller (no splitk_epilogue_kernel)
    // pair split (256 x 256 tiles that cannot fill the chip): blockIdx.y = K half; the half that finishes first parks its
    // accumulators in `pair_slab` (register order) and signals, the other adds them and runs the epilogue
    int pair;
    float* pair_slab;            // [tiles][256 * 256] fp32
    unsigned int* pair_flags;    // [tiles][2]: ticket, ready -- zeroed by the launcher before every launch
    int debug = 0;               // diagnostic-build ablations (common.hpp: GCC_DIAG) (timing diagnostics only)
    TailFin fin = {};            // BatchNorm finalize by the last-arriving workgroups (tickets NULL: a separate gcc_bn_finalize)
};

constexpr int BK = 64;   // k per step

// conv_halo.hip: k4 s2 p1 convolutions with the tile's input neighbourhood resident in LDS
struct HaloPlan { int ok, mode, TR, TW, lgTW, HR, HW, HWp, npieces, tiles_x, tiles_y, ntiles, phases, hc; size_t lds; long wgs; };
HaloPlan halo_plan(const gcc_conv_t* c, int dgrad);
struct TailFin;
int launch_halo(const gcc_conv_t* c, int dgrad, const HaloPlan& h, const void* src, const void* w, void* dst, const gcc_epilogue_t* ep,
                const TailFin* fin, hipStream_t st);
constexpr uint32_t OOB = 0x7FFFFFF0u;

// BP pixels x BC channels per workgroup.  BP = 128: 4 waves (2 workgroups per CU); BP = 256: 8 waves,
// one workgroup per CU -- the big tiles halve the L2 -> LDS traffic per MFMA (128x128x64 needs 64 FLOP/B,
// i.e. ~39 TB/s of L2 bandwidth at the 2.5 PF peak, more than the 8 L2s deliver; 256x256 needs half).
template <int BP, int BC>
struct Cfg {
    // 128-pixel tiles: 4 waves (2 workgroups per CU); 256-pixel tiles: 8 waves (256x256: 128x64 per wave).
    // Measured alternative for 256x256 (kept expressible through WAVES / AI): 4 waves of 128x128, one per SIMD with
    // the whole AGPR file as accumulators -- a third less LDS fragment traffic per k-step, but with a single wave per
    // SIMD nothing covers the barrier and the exposed first fragment reads of each k-step: 670 / 934 TFLOP/s on the
    // PatchGAN L2 / L4 forward shapes against 936 / 1045 with 8 waves.  A fully pipelined version of it (4-stage LDS
    // ring of 32-deep k-steps, fragments of stage t+1 read under the MFMAs of stage t) was also measured: 473 / 631 --
    // 64-byte LDS-DMA rows double the number of global requests per byte and the 128x128 blocks spill.
    static constexpr int WAVES = BP / 32;
    static constexpr int NT = WAVES * 64;
    static constexpr int AI = (BP / 8) / WAVES;            // 1-KiB pixel staging instructions per wave and k-step (4 or 8)
    static constexpr int WC = (BC >= 128) ? 2 : 1;        // waves along channels
    static constexpr int WP = WAVES / WC;                  // waves along pixels
    static constexpr int TC = BC / WC;                     // channels per wave
    static constexpr int TP = BP / WP;                     // pixels per wave
    static constexpr int CB = TC / 16;
    static constexpr int PB = TP / 16;
    static constexpr int WI = BC / 8;                      // 1-KiB weight staging instructions per k-step
    static constexpr int WPW = WI / WAVES;                 // ... per wave (0: the first WI waves issue one)
    static constexpr int WN_GLDS = WPW > 0 ? WPW : 1;      // weight staging instructions a wave issues per k-step
    static constexpr int W_CHUNKS = (BC * 8 + NT - 1) / NT;  // register path: 16-B weight chunks per thread
    static constexpr int LDS_BYTES_LOOP = 2 * (BP + BC) * BK * 2;
    static constexpr int OSTRIDE = BC * 2 + 16;            // epilogue tile row stride (bytes)
    static constexpr int LDS_BYTES_EPI = BP * OSTRIDE + 2 * NT * 4 + 16;      // + the ticket word of stats_tail
    static constexpr int LDS_BYTES = LDS_BYTES_LOOP > LDS_BYTES_EPI ? LDS_BYTES_LOOP : LDS_BYTES_EPI;
};

// epilogue shared by the igemm kernels: fp32 partial tiles (split-K / raw route) or bias + activation + bf16 rounding,
// NHWC store through an LDS transpose, BatchNorm partial statistics.  C carries the tile constants (NT, CB, PB, TC, TP,
// OSTRIDE); acc[i][j][r]: channel = wc*TC + i*16 + 4*lq + r ; pixel = wp*TP + j*16 + lr
template <class C, int BP, int BC>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& p, f32x4 (&acc)[C::CB][C::PB], char* smem, int tid, int lr,
                                               int lq, int wc, int wp, int m0, int n0, int M, int Hg, int Wg, int ostr, int py,
                                               int px, int mt, int ks_idx, bf16_t* dstp) {
    constexpr int NT = C::NT;
    // ---- epilogue ------------------------------------------------------------------------------
    // acc[i][j][r]: channel = wc*TC + i*16 + 4*lq + r ; pixel = wp*TP + j*16 + lr
    if (p.partial) {      // split-K slices, or the raw fp32 route of the single-output-channel head (ksplit may be 1)
        float* part = p.partial + ((size_t)(blockIdx.z * p.ksplit + ks_idx) * p.rows_max) * p.Cpad;
#pragma unroll
        for (int j = 0; j < C::PB; j++) {
            const int m = m0 + wp * C::TP + j * 16 + lr;
            if (m < M) {
#pragma unroll
                for (int i = 0; i < C::CB; i++) {
                    const int cl = n0 + wc * C::TC + i * 16 + 4 * lq;
                    *(f32x4*)(part + (size_t)m * p.Cpad + cl) = acc[i][j];
                }
            }
        }
        return;
    }
    char* sO = smem;
#pragma unroll
    for (int i = 0; i < C::CB; i++) {
        const int cl = wc * C::TC + i * 16 + 4 * lq;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int r = 0; r < 4; r++) bv[r] = (n0 + cl + r < p.Cout) ? p.bias[n0 + cl + r] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < C::PB; j++) {
            const int pl = wp * C::TP + j * 16 + lr;
            // rows past the end of the phase are padding: keep them exact zeros (the statistics below sum every row
            // of the tile, and a bias would otherwise leak into them)
            const bool live = !p.stats || (m0 + pl < M);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = acc[i][j][r] + bv[r];
            apply_actN<4>(v, v, p.act, p.slope);
            if (!live) v[0] = v[1] = v[2] = v[3] = 0.f;
            i32x2 pk;
            pk[0] = (int)pack2bf(v[0], v[1]);
            pk[1] = (int)pack2bf(v[2], v[3]);
            *(i32x2*)(sO + pl * C::OSTRIDE + cl * 2) = pk;
        }
    }
    __syncthreads();

    // coalesced NHWC stores: 16-B chunks, consecutive threads -> consecutive channels of a pixel
    constexpr int CPR = BC / 8;                 // chunks per pixel row
    constexpr int NCH = BP * CPR;
    const int cend = ceil8(p.Cout);
    for (int q = tid; q < NCH; q += NT) {
        const int row = q / CPR;
        const int cch = q - row * CPR;
        const int m = m0 + row;
        const int ch = n0 + cch * 8;
        if (m < M && ch < cend) {
            const int n = m / (Hg * Wg);
            const int r = m - n * (Hg * Wg);
            const int oy = r / Wg;
            const int ox = r - oy * Wg;
            const size_t o = ((size_t)(n * p.Hd + oy * ostr + py) * p.Wd + (ox * ostr + px)) * p.ldd + p.doff + ch;
            *(i32x4*)(dstp + o) = *(const i32x4*)(sO + row * C::OSTRIDE + cch * 16);
        }
    }

    // BatchNorm partial statistics of the rounded outputs (rows >= M are exact zeros)
    if (p.stats) {
        float* sR = (float*)(smem + BP * C::OSTRIDE);
        constexpr int PARTS = NT / BC;
        constexpr int ROWS = BP / PARTS;
        const int c = tid % BC;
        const int part = tid / BC;
        float s = 0.f, ss = 0.f;
        if (part < PARTS) {
            for (int r = part * ROWS; r < (part + 1) * ROWS; r++) {
                const float v = bf2f(*(const bf16_t*)(sO + r * C::OSTRIDE + c * 2));
                s += v; ss += v * v;
            }
        }
        sR[tid] = s; sR[NT + tid] = ss;
        __syncthreads();
        if (tid < BC && n0 + tid < p.Cout) {
            float ts = 0.f, tss = 0.f;
#pragma unroll
            for (int q = 0; q < PARTS; q++) { ts += sR[q * BC + tid]; tss += sR[NT + q * BC + tid]; }
            const int trow = blockIdx.z * p.mtiles_max + mt;
            st_stat(p.stats + ((size_t)trow * 2 + 0) * p.Cout + n0 + tid, ts, p.fin.tickets != nullptr);
            st_stat(p.stats + ((size_t)trow * 2 + 1) * p.Cout + n0 + tid, tss, p.fin.tickets != nullptr);
        }
        if (p.fin.tickets) stats_tail<NT>(p.fin, p.stats, p.Cout, blockIdx.z * p.mtiles_max + mt, (int*)(smem + BP * C::OSTRIDE + 2 * NT * 4), tid);
    }
}
}  // namespace gcc_igemm
