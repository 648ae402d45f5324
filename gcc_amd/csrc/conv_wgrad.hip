// Weight gradient of a convolution on MFMA for gfx950:
//     dW[co][tap][ci] = sum over output pixels m of dY[m][co] * X[pixel(m, tap)][ci]
// The contraction runs over pixels, which is the slow (strided) dimension of both NHWC operands, so
// both LDS tiles are [64 pixels][128 channels] images (filled by LDS-DMA, zero padding by the hardware
// range check) and the MFMA fragments are fetched with the CDNA4 transposing read ds_read_b64_tr_b16.
// The 32-byte windows of a row are XOR-swizzled with the row number so that the 8 pixel rows one
// half-wave touches per transposed read land on 8 distinct 32-B bank groups (conflict-free).
//
// Tile: 128 (tap,ci) columns x 128 output channels, 64 pixels per step, 256 threads = 2x2 waves; or 256 x 256 with
// 8 waves of 128 x 64 for large outputs (WCfg).
// Split-K over pixels: every split writes an fp32 slab, gcc_wgrad_reduce folds the slabs into the
// fp32 master-layout gradient (deterministic; no float atomics).
#include <mutex>
#include <string.h>
#include "common.hpp"

namespace {

struct WgradParams {
    const bf16_t* x;
    const bf16_t* dy;
    float* out;          // slabs [splits][Co][ncols] (ncols = taps*Cip) or direct dW when splits==1 && direct
    int N, H, W, ldx, xoff, Cip, Ci;
    int Ho, Wo, ldy, yoff, Co;
    int KH, KW, stride, pad;
    int ncols;           // taps * Cip
    int M;               // N*Ho*Wo
    int ksteps_per_split;
    int col_tiles, co_tiles;
    uint32_t x_bytes, dy_bytes;
    FastDiv dHW, dW, dCip, dKW;
    int batch;                       // independent problems on blockIdx.y (gram matrices per image)
    long x_bstride, dy_bstride;      // elements
    int direct;                      // 1: no split -> write (accumulate) straight into dW, no slab / fold pass
    int accumulate;
    float* dw;
    int rowmode;                     // how issue_loads finds a piece's pixels: 0 every lane decomposes its pixel every step, 1 LDS pixel table
    int debug;                       // diagnostic-build ablations (common.hpp: GCC_DIAG) (timing diagnostics only; results are wrong)
};

constexpr int TP = 64;          // pixels per step
constexpr uint32_t OOB = 0x7FFFFFF0u;

// Tile configurations.  BIG = false: 128 columns x 128 output channels, 4 waves of 64 x 64 (two workgroups per CU).
// BIG = true: 256 x 256, 8 waves of 128 (columns) x 64 (output channels) -- 0.375 transposed fragments per MFMA instead
// of 0.5, one workgroup per CU; used when the output is large enough that the extra pixel splits stay few.
template <bool BIG>
struct WCfg {
    static constexpr int WAVES = BIG ? 8 : 4;
    static constexpr int NT = WAVES * 64;
    static constexpr int TCOL = BIG ? 256 : 128;
    static constexpr int TCO = BIG ? 256 : 128;
    static constexpr int RS = TCOL * 2;                 // LDS row stride (bytes): no padding, rows are laid down by 1-KiB LDS-DMA pieces
    static constexpr int TILE_BYTES = TP * RS;
    static constexpr int CPR = RS / 16;                 // 16-byte chunks per row (16 or 32)
    static constexpr int RPP = 64 / CPR;                // pixel rows per 1-KiB piece (4 or 2)
    static constexpr int WCOL = BIG ? 128 : 64;         // columns per wave
    static constexpr int FI = WCOL / 16;                // column fragments per wave (4 or 8)
    static constexpr int FJ = 4;                        // output-channel fragments per wave (64 channels)
    // pixel table (rowmode 1): two chunks of CH k-steps x 64 pixels x {gather base, validity mask}
    static constexpr int CH = BIG ? 16 : 8;
    static constexpr int TBL_BYTES = 2 * CH * TP * 8;
    static constexpr int LDS_BYTES = 4 * TILE_BYTES + TBL_BYTES;
};

// Both panels are [64 pixels][TCOL channels] images filled by LDS-DMA (buffer_load ... lds: one wave instruction lays
// down 1 KiB = RPP pixel rows; no staging registers, no ds_write).  The 32-byte windows of a row are XOR-swizzled with
// the row number (on the SOURCE side: a lane fetches the channels that belong in its physical slot), so that the 8 rows
// a half-wave touches per transposing read fall on 8 distinct bank groups.
__device__ __forceinline__ int phys_col_bytes(int row, int col_bytes) {          // col_bytes: logical byte offset in the row
    return ((((col_bytes >> 5) ^ (row & 7)) << 5) | (col_bytes & 31));
}

template <int RS>
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int ks, int colbase, int lane) {
    // 16x16x32 operand from a [pixel][channel] image: lane (g = lane>>4, i = lane&15) ends up with
    // channel colbase+i and the 8 pixel rows {ks*32 + 4g + 0..3, ks*32 + 16 + 4g + 0..3}.
    const int g = lane >> 4, i = lane & 15;
    const int r0 = ks * 32 + 4 * g + (i >> 2), r1 = r0 + 16;
    const int cb = (colbase + 4 * (i & 3)) * 2;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + r0 * RS + phys_col_bytes(r0, cb)));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + r1 * RS + phys_col_bytes(r1, cb)));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// One workgroup's share of a weight gradient: output tile `tile` (column tile fastest), pixel split `split`, problem `bidx` of a
// batched call.  wgrad_kernel hands it its own block indices; wgrad_group_kernel (below) the ones of a table entry.
template <bool BIG, bool TABLE>
__device__ __forceinline__ void wgrad_body(const WgradParams& p, const int tile, const int split, const int bidx, char* smem) {
    using C = WCfg<BIG>;
    constexpr int RS = C::RS, TILE_BYTES = C::TILE_BYTES, FI = C::FI, FJ = C::FJ;
    char* sX = smem;                       // [2][64][RS]
    char* sY = smem + 2 * TILE_BYTES;      // [2][64][RS]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave & 1, wb = wave >> 1;               // 2 column groups x (2 or 4) output-channel groups

    const int ct = tile % p.col_tiles;     // column tile fastest: neighbours share the dY panel
    const int ot = tile / p.col_tiles;
    const int col0 = ct * C::TCOL, co0 = ot * C::TCO;
    const int k_begin = split * p.ksteps_per_split;
    int k_end = k_begin + p.ksteps_per_split;
    const int ksteps_total = (p.M + TP - 1) / TP;
    if (k_end > ksteps_total) k_end = ksteps_total;

    const i32x4 rs_x = make_rsrc(p.x + (size_t)bidx * p.x_bstride, p.x_bytes);
    const i32x4 rs_y = make_rsrc(p.dy + (size_t)bidx * p.dy_bstride, p.dy_bytes);
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);

    // LDS-DMA pieces of this wave: piece i (0..3) of a panel covers pixel rows RPP * (4 * wave + i) + rsub.  The physical
    // 16-byte chunk pch of a row holds the logical chunk whose 32-byte window is XORed with (row & 7); a lane meets at
    // most four values of row & 7 (one per piece), each with a fixed (tap, channel).
    const int rsub = lane / C::CPR, pch = lane % C::CPR;
    int q_tap_dy[4], q_tap_dx[4], q_cx[4], q_coy[4];
    bool q_colok[4], q_cook[4];
    // rowmode 1: byte offsets relative to the table's gather base / the step's first dy row, and the mask bits a lane needs
    int cxl[4], cyl[4];
    uint32_t xbits[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r7 = (C::RPP * (4 * wave + i) + rsub) & 7;
        const int lch = (((pch >> 1) ^ r7) << 1) | (pch & 1);        // logical 16-byte chunk
        const int q = col0 + lch * 8;
        q_colok[i] = q < p.ncols;
        const int tap = fdiv(q, p.dCip);
        q_cx[i] = q - tap * p.Cip;
        const int kh = fdiv(tap, p.dKW);
        const int kw = tap - kh * p.KW;
        q_tap_dy[i] = kh - p.pad; q_tap_dx[i] = kw - p.pad;
        q_coy[i] = co0 + lch * 8;
        q_cook[i] = q_coy[i] < ((p.Co + 7) & ~7);
        cxl[i] = ((kh * p.W + kw) * p.ldx + q_cx[i]) * 2;
        // a lane with no dy channels to load: an offset beyond 2 GB, which the descriptor's range check answers with zeros
        cyl[i] = q_cook[i] ? ((C::RPP * (4 * wave + i) + rsub) * p.ldy + p.yoff + q_coy[i]) * 2 : (int)0x80000000u;
        // bit 15 is never set in a table mask: a lane with no x columns to load can never match
        xbits[i] = q_colok[i] ? (0x80000000u | (1u << (kh & 15)) | (1u << (16 + (kw & 15)))) : 0x80008000u;
    }

    // A piece's pixels are m = kstep * 64 + RPP * (4 * wave + i) + rsub.  Decomposing m every step (two divisions by launch
    // constants, the gather address, bounds) cost ~110 vector + ~230 scalar instructions per wave and k-step beside 32 or 64
    // MFMAs, against 34 + 31 in igemm_kernel, whose offsets advance by a constant (rocprofv3 SQ_INSTS_*, profiles/r3h_pmc.txt);
    // neither pipe is free beside an MFMA stream: the L4 weight gradient took 245 us with this arithmetic and cache-hot loads
    // against 178 us with neither (profiles/r3c_ablate.txt), whether the arithmetic ran in front of the step's MFMAs, pinned
    // between them, or on the scalar unit.  rowmode 1: every pixel of the workgroup's range is decomposed ONCE, by one thread,
    // into an LDS table entry {byte offset of x[n][oy*s - pad][ox*s - pad], mask: bit kh = row oy*s - pad + kh inside the image,
    // bit 16 + kw = column inside, bit 31 = pixel exists}; a lane's step is one 8-byte LDS read, an add and a mask test per
    // piece.  The table is a ring of two chunks of CH steps, the next-but-one chunk is rebuilt once per CH steps.
    char* tbl = smem + 4 * TILE_BYTES;
    const int nk = k_end - k_begin;
    auto build_chunk = [&](int c) {
        for (int e = tid; e < C::CH * TP; e += C::NT) {
            const int step = c * C::CH + (e >> 6);
            const int m = (k_begin + step) * TP + (e & 63);
            i32x2 ent = {0, 0};
            if (step < nk && m < p.M) {
                const int n = fdiv(m, p.dHW);
                const int r = m - n * (p.Ho * p.Wo);
                const int oy = fdiv(r, p.dW);
                const int ox = r - oy * p.Wo;
                const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
                uint32_t mask = 0x80000000u;
                for (int k = 0; k < p.KH; k++) mask |= ((unsigned)(iy0 + k) < (unsigned)p.H) ? (1u << k) : 0u;
                for (int k = 0; k < p.KW; k++) mask |= ((unsigned)(ix0 + k) < (unsigned)p.W) ? (1u << (16 + k)) : 0u;
                ent[0] = (((n * p.H + iy0) * p.W + ix0) * p.ldx + p.xoff) * 2;
                ent[1] = (int)mask;
            }
            *(i32x2*)(tbl + ((c & 1) * (C::CH * TP) + e) * 8) = ent;
        }
    };
    const int tl = (C::RPP * 4 * wave + rsub) * 8;       // this lane's first entry inside a step's 64
    auto issue_loads = [&](int kstep, int stage) {       // kstep relative to k_begin
        const char* te = tbl + (kstep & (2 * C::CH - 1)) * (TP * 8) + tl;
        const int sby = (k_begin + kstep) * TP * p.ldy * 2;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t offx, offy;
            if constexpr (!TABLE) {
                const int row = C::RPP * (4 * wave + i) + rsub;
                const int m = (k_begin + kstep) * TP + row;
                const bool mv = m < p.M;
                const int n = fdiv(m, p.dHW);
                const int r = m - n * (p.Ho * p.Wo);
                const int oy = fdiv(r, p.dW);
                const int ox = r - oy * p.Wo;
                const int iy = oy * p.stride + q_tap_dy[i], ix = ox * p.stride + q_tap_dx[i];
                const bool okx = mv && q_colok[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                offx = okx ? (uint32_t)((((n * p.H + iy) * p.W + ix) * p.ldx + p.xoff + q_cx[i]) * 2) : OOB;
                const bool oky = mv && q_cook[i];
                offy = oky ? (uint32_t)((m * p.ldy + p.yoff + q_coy[i]) * 2) : OOB;
            } else {
                const i32x2 ent = *(const i32x2*)(te + i * (C::RPP * 8));
                const uint32_t mask = (uint32_t)ent[1];
                offx = (mask & xbits[i]) == xbits[i] ? (uint32_t)(ent[0] + cxl[i]) : OOB;
                offy = (int)mask < 0 ? (uint32_t)(sby + cyl[i]) : OOB;
            }
            const uint32_t dx = lds0 + stage * TILE_BYTES + (4 * wave + i) * 1024;
            lds_dma16(rs_x, dx, offx);
            lds_dma16(rs_y, dx + 2 * TILE_BYTES, offy);
        }
    };

    f32x4 acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; i++)
#pragma unroll
        for (int j = 0; j < FJ; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nk > 0) {
        if constexpr (TABLE) {
            build_chunk(0);
            build_chunk(1);
            __syncthreads();
        }
        // one barrier per step: [tile kt landed for every wave AND everyone left tile kt-1] -> issue tile kt+1 into the
        // stage tile kt-1 occupied -> compute tile kt while it flies
        issue_loads(0, 0);
        for (int kt = 0; kt < nk; kt++) {
            const int cur = kt & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < nk && !(GCC_DIAG(p.debug) & 2)) issue_loads((GCC_DIAG(p.debug) & 4) ? 0 : kt + 1, cur ^ 1);
            // chunk (kt+1)/CH - 1 was last read a step ago (behind this step's barrier); its slot takes chunk (kt+1)/CH + 1
            if constexpr (TABLE) {
                if (((kt + 1) & (C::CH - 1)) == 0) build_chunk(((kt + 1) / C::CH) + 1);
            }
            const char* tx = sX + cur * TILE_BYTES;
            const char* ty = sY + cur * TILE_BYTES;
            // both k-slices in registers; the transposing reads of slice 1 are issued under the MFMAs of slice 0
            // (sched_group_barrier pins the order: the compiler otherwise reads, waits, and only then multiplies)
            bf16x8 fx[2][FI], fy[2][FJ];
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
#pragma unroll
                for (int i = 0; i < FI; i++) fx[ks][i] = tr_frag<RS>(tx, ks, wa * C::WCOL + i * 16, lane);
#pragma unroll
                for (int j = 0; j < FJ; j++) fy[ks][j] = tr_frag<RS>(ty, ks, wb * 64 + j * 16, lane);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
#pragma unroll
                for (int i = 0; i < FI; i++)
#pragma unroll
                    for (int j = 0; j < FJ; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[ks][i], fy[ks][j], acc[i][j], 0, 0, 0);
            constexpr int READS = 2 * (FI + FJ);                 // transposing reads per k-slice
            constexpr int MFMAS = FI * FJ;
            constexpr int MPR = MFMAS / (READS / 2) > 0 ? MFMAS / (READS / 2) : 1;      // MFMAs per pair of reads
            __builtin_amdgcn_sched_group_barrier(0x100, READS, 0);          // slice 0
#pragma unroll
            for (int r = 0; r < READS / 2; r++) {
                __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * MFMAS - (READS / 2) * MPR, 0);
        }
    }

    // acc[i][j][r] = dW[co = co0 + wb*64 + j*16 + (lane&15)][col = col0 + wa*WCOL + i*16 + 4*(lane>>4) + r]
    float* slab = p.direct ? p.dw + (size_t)bidx * p.Co * p.ncols
                           : p.out + ((size_t)split * p.batch + bidx) * p.Co * p.ncols;
    const bool rmw = p.direct && p.accumulate;
#pragma unroll
    for (int j = 0; j < FJ; j++) {
        const int co = co0 + wb * 64 + j * 16 + (lane & 15);
        if (co < p.Co) {
#pragma unroll
            for (int i = 0; i < FI; i++) {
                const int col = col0 + wa * C::WCOL + i * 16 + 4 * (lane >> 4);
                if (col < p.ncols) {
                    f32x4* d = (f32x4*)(slab + (size_t)co * p.ncols + col);
                    f32x4 v = acc[i][j];
                    if (rmw) { const f32x4 o = *d; v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3]; }
                    *d = v;
                }
            }
        }
    }
}

template <bool BIG, bool TABLE>
__global__ __launch_bounds__(WCfg<BIG>::NT) void wgrad_kernel(const WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    wgrad_body<BIG, TABLE>(p, xcd_remap(blockIdx.x, gridDim.x), blockIdx.z, blockIdx.y, smem);
}

// ---- grouped weight gradients (round 6) -------------------------------------------------------------------------------------------
// The backward pass of a generator is a chain of small layers: 14 regular weight gradients per U-Net pass (the headline step runs
// the student's and the teacher's), each a launch of 4-128 tiles x a few pixel splits plus a fold launch -- 58 + ~45 launches per
// iteration at 13 % matrix-pipe utilisation, none of which fills the chip.  A GROUP runs them as ONE launch
// over a table: workgroup b finds its entry by the table's prefix sums (<= GCC_WGRAD_GROUP_MAX entries, a uniform scan), takes the
// entry's WgradParams and runs wgrad_body on its (tile, split).  Pixel splits are chosen for the group as a whole (equal k-steps per
// workgroup over all entries, longest workgroups first), split entries write fp32 slabs, and ONE fold launch (wgrad_group_reduce_kernel)
// adds each entry's slabs in slab order into its master-layout gradient: no float atomics, same bits run after run.
constexpr int GROUP_MAX = GCC_WGRAD_GROUP_MAX;
struct GroupReduce { const float* slabs; float* dw; unsigned n4; int splits; int accumulate; int blk0; };
struct WgradGroupTable {
    int magic, n_items, total_wgs, n_reduce, reduce_blocks, pad_[3];
    int wg_prefix[GROUP_MAX + 1];
    int tiles[GROUP_MAX];
    GroupReduce red[GROUP_MAX];
    WgradParams items[GROUP_MAX];
};
constexpr int GROUP_MAGIC = 0x47525036;

__global__ __launch_bounds__(WCfg<false>::NT) void wgrad_group_kernel(const WgradGroupTable* __restrict__ t) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x, n = t->n_items;
    int it = 0;
    for (int i = 1; i < n; i++) it = b >= t->wg_prefix[i] ? i : it;        // uniform: every operand is a scalar load
    it = __builtin_amdgcn_readfirstlane(it);
    const int li = b - t->wg_prefix[it], tiles = t->tiles[it];
    const int split = li / tiles, tile = li - split * tiles;
    const WgradParams p = t->items[it];
    wgrad_body<false, true>(p, tile, split, 0, smem);
}

__global__ __launch_bounds__(256) void wgrad_group_reduce_kernel(const WgradGroupTable* __restrict__ t) {
    const int b = blockIdx.x, n = t->n_reduce;
    int it = 0;
    for (int i = 1; i < n; i++) it = b >= t->red[i].blk0 ? i : it;
    it = __builtin_amdgcn_readfirstlane(it);
    const GroupReduce r = t->red[it];
    const size_t i = (size_t)(b - r.blk0) * 256 + threadIdx.x;
    if (i >= r.n4) return;
    const f32x4* src = (const f32x4*)r.slabs + i;
    // the launch's own sum first (slabs in order, eight loads in flight), then ONE add to what the buffer holds (as wgrad_reduce_flat_kernel)
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int z = 0;
    for (; z + 8 <= r.splits; z += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = src[(size_t)(z + u) * r.n4];
#pragma unroll
        for (int u = 0; u < 8; u++) { s[0] += v[u][0]; s[1] += v[u][1]; s[2] += v[u][2]; s[3] += v[u][3]; }
    }
    for (; z < r.splits; z++) { const f32x4 v = src[(size_t)z * r.n4]; s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3]; }
    f32x4* d = (f32x4*)r.dw + i;
    if (r.accumulate) { const f32x4 o = *d; s[0] += o[0]; s[1] += o[1]; s[2] += o[2]; s[3] += o[3]; }
    *d = s;
}

// regular widths (Ci % 8 == 0: slab layout == dW layout): 16-byte fold.  256 threads = 64 outputs
// (float4) x 4 split lanes, so that layers with a small dW and hundreds of pixel splits (first
// layers: 262144 pixels) still spread over many workgroups and short dependent chains.
__global__ __launch_bounds__(256) void wgrad_reduce_vec_kernel(const float* __restrict__ slabs, float* __restrict__ dw,
                                                               int splits, size_t n4, int accumulate) {
    __shared__ f32x4 sh[4][64];
    const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
    for (size_t base = (size_t)blockIdx.x * 64; base < n4; base += (size_t)gridDim.x * 64) {
        const size_t i = base + o;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (i < n4) {
            for (int z = sl; z < splits; z += 4) {
                const f32x4 v = ((const f32x4*)slabs)[(size_t)z * n4 + i];
                s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
            }
        }
        sh[sl][o] = s;
        __syncthreads();
        if (sl == 0 && i < n4) {
            for (int q = 1; q < 4; q++) { const f32x4 v = sh[q][o]; s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3]; }
            if (accumulate) { const f32x4 ov = ((const f32x4*)dw)[i]; s[0] += ov[0]; s[1] += ov[1]; s[2] += ov[2]; s[3] += ov[3]; }
            ((f32x4*)dw)[i] = s;
        }
        __syncthreads();
    }
}

// few slabs (<= 8: the usual case, large dW): one thread per float4 output, every slab load and the accumulate load in
// flight at once, no LDS exchange and no barrier
__global__ __launch_bounds__(256) void wgrad_reduce_flat_kernel(const float* __restrict__ slabs, float* __restrict__ dw,
                                                                int splits, size_t n4, int accumulate) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 v[8];
#pragma unroll
        for (int z = 0; z < 8; z++)
            if (z < splits) v[z] = ((const f32x4*)slabs)[(size_t)z * n4 + i];
        // the launch's own sum first, then ONE add to what the buffer holds: a gradient accumulated over several passes is then
        // the same bits whether the passes add to one buffer in turn or to two buffers that are added afterwards
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (accumulate) o = ((const f32x4*)dw)[i];
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int z = 0; z < 8; z++)
            if (z < splits) { s[0] += v[z][0]; s[1] += v[z][1]; s[2] += v[z][2]; s[3] += v[z][3]; }
        if (accumulate) { s[0] += o[0]; s[1] += o[1]; s[2] += o[2]; s[3] += o[3]; }
        ((f32x4*)dw)[i] = s;
    }
}

// irregular widths (Ci % 8 != 0: 3- and 6-channel first layers): dW[co][tap][ci] (+)= sum_z slab[z][co][tap][cip]
// OL outputs x SL split lanes per workgroup.  The image layers have few outputs and hundreds of slabs: with 16 split
// lanes and the slab loop unrolled (independent loads in flight) the fold is no longer a chain of dependent loads
// (36 us for 6 -> 128 at 256 x 256, twice the MFMA kernel it follows, before).
template <int OL, int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int splits,
                                                           int Co, int taps, int Ci, int Cip, int accumulate, int Cop,
                                                           int row_split, int col_split) {
    static_assert(OL * SL == 256, "one workgroup");
    __shared__ float sh[SL][OL];
    const size_t total = (size_t)Co * taps * Ci;
    const size_t slab = (size_t)Cop * taps * Cip;
    const int o = threadIdx.x % OL, sl = threadIdx.x / OL;
    for (size_t base = (size_t)blockIdx.x * OL; base < total; base += (size_t)gridDim.x * OL) {
        const size_t i = base + o;
        float s = 0.f;
        if (i < total) {
            const int c = (int)(i % Ci);
            const size_t rt = i / Ci;
            const int tap = (int)(rt % taps), r = (int)(rt / taps);
            const float* src = slabs + ((size_t)seg_to_phys(r, Co, row_split) * taps + tap) * Cip + seg_to_phys(c, Ci, col_split);
            int z = sl;
            for (; z + 7 * SL < splits; z += 8 * SL) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = src[(size_t)(z + u * SL) * slab];
#pragma unroll
                for (int u = 0; u < 8; u++) s += v[u];
            }
            for (; z < splits; z += SL) s += src[(size_t)z * slab];
        }
        sh[sl][o] = s;
        __syncthreads();
        if (sl == 0 && i < total) {
#pragma unroll
            for (int q = 1; q < SL; q++) s += sh[q][o];
            dw[i] = accumulate ? dw[i] + s : s;
        }
        __syncthreads();
    }
}


// ---- tap-stationary weight gradient: k4 s1 p1, channels in multiples of 64 (round 4) ------------------------------------------
// wgrad_kernel above stages, per 64 pixels, one [64][256] x image PER COLUMN TILE: the 16 taps of a 4x4 stride-1 filter read the
// same input pixels shifted by (kh, kw), yet every (tap, ci) column block fetches its own copy, and a workgroup moves 64 KB
// through L2 -> LDS per 512 MFMAs; the L4 weight gradient of the discriminators (512 -> 1024 at 32 x 32, 1.67 ms of the
// production step) ran at the LDS-DMA rate, not the MFMA rate.  Here the output tile is 64 input channels x 64
// output channels x ALL 16 taps, and the pixels arrive as 8 x 16 blocks of output positions: one [128][64] dy image and one
// [11 x 19 halo][64] x image per block; a tap is a row shift (kh * pitch + kw) into the halo image, so the 16 taps share one
// staged copy -- 49 KB per 1024 MFMAs, 2.6 times fewer bytes per MFMA.  8 waves: wave w owns taps {2w, 2w + 1} (one kh, two
// kw) of the whole 64 x 64 channel tile.  Three LDS stages, two blocks in flight (counted vmcnt), one barrier per block.
// Rows are 128 bytes (64 channels); the four 32-byte windows of a row are XOR-swizzled with bits 1-2 of the row number, so the
// 8 consecutive rows a half-wave touches per transposing read fall on 8 distinct bank groups whatever the tap shift (the
// halo pitch, 24 rows, is a multiple of 8: kh never changes those bits).
// KS = 3 (round 6: k3 s1 p1, the 33 Conv2d(64, 64, 3, 1, 1) of SRGAN's SRResNet trunk at 96 x 96 x 16 images and the stride-1 layers
// of its discriminator): nine taps share the [10 x 18 halo][64] x image; nine waves, one tap each (64 x 64 channels: 16 MFMA tiles).
template <int KS>
struct TsCfg {
    static constexpr int TR = 8, TW = 16, PX = TR * TW;           // output positions per block
    static constexpr int HP = 24, HW = TW + KS - 1, HR = TR + KS - 1;   // halo pitch (rows of the LDS image per halo line), used width, lines
    static constexpr int XROWS = HR * HP;                         // 264 / 240
    static constexpr int XP = XROWS / 8, YP = PX / 8;             // 1-KiB LDS-DMA pieces: 33 / 30 + 16
    static constexpr int RSB = 128;                               // bytes per LDS row
    static constexpr int Y_BYTES = PX * RSB, X_BYTES = XROWS * RSB;
    static constexpr int STAGE = Y_BYTES + X_BYTES;               // 50176 / 47104
    static constexpr int NS = 3;
    static constexpr int LDS_BYTES = NS * STAGE;                  // 150528 / 141312: one workgroup per CU
    static constexpr int NW = KS == 4 ? 8 : 9;                    // waves
    static constexpr int TPW = KS == 4 ? 2 : 1;                   // taps per wave
    static constexpr int NT = NW * 64;
    static constexpr int YJ = (YP + NW - 1) / NW, XJ = (XP + NW - 1) / NW;      // pieces per wave: 2 + 5 / 2 + 4 (the last ones partly)
};
namespace ts {
constexpr int TR = 8, TW = 16;
}

struct TsParams {
    const bf16_t* x;
    const bf16_t* dy;
    float* out;          // slabs [splits][Co][16 * Ci]
    float* dw;
    int N, H, W, ldx, xoff, Ci;
    int Ho, Wo, ldy, yoff, Co;
    int ncols;           // 16 * Ci
    int ci_tiles, co_tiles;
    int TY, TX;          // blocks per image
    int blocks, blocks_per_split;
    uint32_t x_bytes, dy_bytes;
    FastDiv dT, dTX;
    int direct, accumulate;
};

__device__ __forceinline__ bf16x8 ts_frag(const char* a, int off1) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a + off1));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <int KS>
__global__ __launch_bounds__(TsCfg<KS>::NT) void wgrad_ts_kernel(const TsParams p) {
    using C = TsCfg<KS>;
    constexpr int TR = C::TR, TW = C::TW, HP = C::HP, HW = C::HW, RSB = C::RSB, Y_BYTES = C::Y_BYTES, STAGE = C::STAGE, NS = C::NS;
    constexpr int NW = C::NW, TPW = C::TPW, YJ = C::YJ, XJ = C::XJ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = tile % p.ci_tiles, ot = tile / p.ci_tiles;       // input-channel tile fastest: neighbours share the dy panel
    const int ci0 = ct * 64, co0 = ot * 64;
    const int split = blockIdx.z;
    const int b_begin = split * p.blocks_per_split;
    int b_end = b_begin + p.blocks_per_split;
    if (b_end > p.blocks) b_end = p.blocks;
    const int nb = b_end - b_begin;

    const i32x4 rs_x = make_rsrc(p.x, p.x_bytes);
    const i32x4 rs_y = make_rsrc(p.dy, p.dy_bytes);
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);

    // LDS-DMA pieces of this wave: dy pieces wave + NW j (< 16), x pieces wave + NW j (< 33 / 30).
    // Lane (rsub, pch) of a piece lays down the physical 16-byte chunk pch of row 8 * piece + rsub: it fetches the logical chunk
    // whose 32-byte window is XORed with bits 1-2 of the row.
    const int rsub = lane >> 3, pch = lane & 7;
    int yl[YJ], ypy[YJ], ypx[YJ];
#pragma unroll
    for (int j = 0; j < YJ; j++) {
        const int row = 8 * (wave + NW * j) + rsub;
        const int lch = (((pch >> 1) ^ ((row >> 1) & 3)) << 1) | (pch & 1);
        ypy[j] = row >> 4; ypx[j] = row & 15;
        yl[j] = ((ypy[j] * p.Wo + ypx[j]) * p.ldy + p.yoff + co0 + lch * 8) * 2;
    }
    int xl[XJ], xhy[XJ], xhx[XJ];
#pragma unroll
    for (int j = 0; j < XJ; j++) {
        const int row = 8 * (wave + NW * j) + rsub;
        const int lch = (((pch >> 1) ^ ((row >> 1) & 3)) << 1) | (pch & 1);
        xhy[j] = row / HP; xhx[j] = row - xhy[j] * HP;
        xl[j] = ((xhy[j] * p.W + xhx[j]) * p.ldx + p.xoff + ci0 + lch * 8) * 2;
        if (xhx[j] >= HW) xhx[j] = 1 << 20;            // the unused rows of a halo line: never inside the image
    }
    // pieces this wave issues per block (wave-uniform): what "one block may still fly" means to its vmcnt
    const int mine = ((C::YP - 1 - wave) / NW + 1) + ((C::XP - 1 - wave) / NW + 1);
    auto issue = [&](int b, int stage) {                // b: absolute block number (wave-uniform)
        const int n = fdiv(b, p.dT);
        const int t = b - n * (p.TY * p.TX);
        const int ty = fdiv(t, p.dTX);
        const int Y0 = ty * TR, X0 = (t - ty * p.TX) * TW;
        const int ybase = ((n * p.Ho + Y0) * p.Wo + X0) * p.ldy * 2;
        const int xbase = ((n * p.H + Y0 - 1) * p.W + X0 - 1) * p.ldx * 2;
        const uint32_t d0 = lds0 + stage * STAGE;
#pragma unroll
        for (int j = 0; j < YJ; j++) {
            if ((j + 1) * NW <= C::YP || wave + NW * j < C::YP) {
                const bool ok = Y0 + ypy[j] < p.Ho && X0 + ypx[j] < p.Wo;
                lds_dma16(rs_y, d0 + (wave + NW * j) * 1024, ok ? (uint32_t)(ybase + yl[j]) : OOB);
            }
        }
#pragma unroll
        for (int j = 0; j < XJ; j++) {
            if ((j + 1) * NW <= C::XP || wave + NW * j < C::XP) {
                const bool ok = (unsigned)(Y0 - 1 + xhy[j]) < (unsigned)p.H && (unsigned)(X0 - 1 + xhx[j]) < (unsigned)p.W;
                lds_dma16(rs_x, d0 + Y_BYTES + (wave + NW * j) * 1024, ok ? (uint32_t)(xbase + xl[j]) : OOB);
            }
        }
    };

    // fragment addresses (tr_frag's lane pattern: lane (g, i) reads row 4g + (i >> 2) of a 16-row group, 8 bytes at channel
    // 16t + 4 (i & 3)); the rows of a k-slice are positions (py = 2 ks [+ 1], px = 4g + (i >> 2)) of the block
    const int g = lane >> 4, li = lane & 15;
    const int px = 4 * g + (li >> 2);
    const int kh = KS == 4 ? wave >> 1 : wave / 3, kw0 = KS == 4 ? 2 * (wave & 1) : wave - 3 * (wave / 3);
    int ya[4], xa[TPW][4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        ya[t] = px * RSB + ((t ^ ((px >> 1) & 3)) << 5) + 8 * (li & 3);
#pragma unroll
        for (int tj = 0; tj < TPW; tj++) {
            const int hrow = kh * HP + px + kw0 + tj;
            xa[tj][t] = Y_BYTES + hrow * RSB + ((t ^ ((hrow >> 1) & 3)) << 5) + 8 * (li & 3);
        }
    }

    f32x4 acc[TPW][4][4];
#pragma unroll
    for (int a = 0; a < TPW; a++)
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[a][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nb > 0) {
        issue(b_begin, 0);
        if (nb > 1) issue(b_begin + 1, 1);
        int cur = 0, fill = 2;
        for (int b = 0; b < nb; b++) {
            // block b landed (block b + 1 may still fly: `mine` pieces of this wave)
            if (b + 1 < nb) {
                if (mine == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                else if (mine == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if (mine == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();                     // ... for every wave, and every wave has left block b - 1: its stage takes b + 2
            if (b + 2 < nb) issue(b_begin + b + 2, fill);
            const char* st = smem + cur * STAGE;
#pragma unroll
            for (int pp = 0; pp < 2; pp++) {
                bf16x8 fx[2][TPW * 4], fy[2][4];
#pragma unroll
                for (int k2 = 0; k2 < 2; k2++) {
                    const int ks = 2 * pp + k2;
#pragma unroll
                    for (int tj = 0; tj < TPW; tj++)
#pragma unroll
                        for (int t = 0; t < 4; t++)
                            fx[k2][tj * 4 + t] = ts_frag(st + xa[tj][t] + 2 * ks * (HP * RSB), HP * RSB);
#pragma unroll
                    for (int t = 0; t < 4; t++) fy[k2][t] = ts_frag(st + ya[t] + 2 * ks * (TW * RSB), TW * RSB);
                }
#pragma unroll
                for (int k2 = 0; k2 < 2; k2++)
#pragma unroll
                    for (int i = 0; i < TPW * 4; i++)
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            acc[i >> 2][i & 3][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[k2][i], fy[k2][j], acc[i >> 2][i & 3][j], 0, 0, 0);
                if constexpr (KS == 4) {
                    // slice 1's transposing reads under slice 0's MFMAs (as in wgrad_kernel)
                    __builtin_amdgcn_sched_group_barrier(0x100, 24, 0);
#pragma unroll
                    for (int r = 0; r < 12; r++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 40, 0);
                } else {
                    // 32 transposing reads, 32 MFMAs: slice 0's reads first, slice 1's under slice 0's MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
#pragma unroll
                    for (int r = 0; r < 8; r++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
                }
            }
            cur = cur == NS - 1 ? 0 : cur + 1;
            fill = fill == NS - 1 ? 0 : fill + 1;
        }
    }

    // acc[tj][i][j][r] = dW[co0 + 16 j + (lane & 15)][tap TPW wave + tj][ci0 + 16 i + 4 (lane >> 4) + r]
    float* slab = p.direct ? p.dw : p.out + (size_t)split * p.Co * p.ncols;
    const bool rmw = p.direct && p.accumulate;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int co = co0 + j * 16 + (lane & 15);
#pragma unroll
        for (int tj = 0; tj < TPW; tj++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int col = (TPW * wave + tj) * p.Ci + ci0 + i * 16 + 4 * (lane >> 4);
                f32x4* d = (f32x4*)(slab + (size_t)co * p.ncols + col);
                f32x4 v = acc[tj][i][j];
                if (rmw) { const f32x4 o = *d; v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3]; }
                *d = v;
            }
    }
}

// the split plan of the tap-stationary route; 0: the geometry (or GCC_OPT_WGRAD_TS) keeps the layer on wgrad_kernel
int ts_plan(const gcc_conv_t* c, int batch, int* blocks_per_split) {
    const int mode = gcc_opt(GCC_OPT_WGRAD_TS);
    const bool k4 = c->KH == 4 && c->KW == 4, k3 = c->KH == 3 && c->KW == 3;
    if (!mode || batch != 1 || !(k4 || k3) || c->stride != 1 || c->pad != 1) return 0;
    if ((c->Ci & 63) || (c->Co & 63) || c->H < 2 || c->W < 2) return 0;
    const int Ho = k4 ? c->H - 1 : c->H, Wo = k4 ? c->W - 1 : c->W;
    const int blocks = c->N * cdiv(Ho, ts::TR) * cdiv(Wo, ts::TW);
    const int tiles = (c->Ci / 64) * (c->Co / 64);
    const int target = plan_or(c->plan.wgrad_wgs_big, PLAN_WGRAD_WGS_BIG);
    int splits = tiles >= (target * 25) / 32 ? 1 : cdiv(target, tiles);
    if (mode == 1) {
        if (k4) {
            // worth it where a workgroup streams enough blocks for the three-stage loop, and the tiles alone nearly fill the launch
            const int max_splits = blocks / 16;
            if (max_splits < 1 || tiles < 32) return 0;
            if (splits > max_splits) splits = max_splits;
        } else {
            // k3: a layer of few tiles (SRGAN's trunk: ONE 64 x 64 tile, 1152 blocks) fills the launch with pixel splits -- at least 8
            // blocks each: every split is a [Co][9 Ci] fp32 slab to write and fold, and the launch runs beside the pass's chain
            // (and at least 128 workgroups: at SRGAN's 24 x 24 training crop the trunk layer has 96 blocks -- 12 workgroups of 8 -- and
            // the column-tiled kernel's ~100 workgroups are faster: 24 -> 96 step 12.3 -> 12.8 ms with this route, profiles/r6_wgrad_ts3.txt)
            const int max_splits = blocks / 8;
            if (splits > max_splits) splits = max_splits;
            if (splits < 1 || (long)tiles * splits < 128) return 0;
        }
    } else if (splits > blocks) {
        splits = blocks;                                   // 2: forced wherever the geometry fits (tests)
    }
    const int per = cdiv(blocks, splits);
    *blocks_per_split = per;
    return cdiv(blocks, per);
}

int plan_splits(const gcc_conv_t* c, int batch, int* ksteps_per_split, bool* big_out = nullptr) {
    const int Ho = gcc_conv_out(c->H, c->KH, c->stride, c->pad), Wo = gcc_conv_out(c->W, c->KW, c->stride, c->pad);
    const long M = (long)c->N * Ho * Wo;
    const int ksteps = (int)((M + TP - 1) / TP);
    const int ncols = c->KH * c->KW * ceil8(c->Ci);
    // 256 x 256 tiles (one workgroup per CU): large regular outputs only, where at most ~8 pixel splits fill the chip
    const int big_mode = gcc_opt(GCC_OPT_WGRAD_BIG);
    const int tiles_big = cdiv(ncols, 256) * cdiv(c->Co, 256);
    bool big = big_mode && batch == 1 && (c->Ci & 7) == 0 && c->Co >= 256 && ncols >= 256 && tiles_big >= gcc_opt(GCC_OPT_WGRAD_BIG_MIN_TILES) &&
               ksteps >= 64;
    int splits;
    if (big) {
        const int target = plan_or(c->plan.wgrad_wgs_big, PLAN_WGRAD_WGS_BIG);
        splits = tiles_big >= (target * 25) / 32 ? 1 : cdiv(target, tiles_big);
        const int max_splits = ksteps / 16 > 0 ? ksteps / 16 : 1;
        if (splits > max_splits) splits = max_splits;
    } else {
        const int tiles = cdiv(ncols, 128) * cdiv(c->Co, 128) * batch;
        // enough tiles to fill the chip (2 workgroups per CU resident): no split, dW written directly;
        // otherwise split the pixel range so that ~512 workgroups exist, >= 8 k-steps (512 pixels) each
        const int target = plan_or(c->plan.wgrad_wgs, PLAN_WGRAD_WGS);
        splits = tiles >= (target * 3) / 8 ? 1 : cdiv(target, tiles);
        const int max_splits = ksteps / 8 > 0 ? ksteps / 8 : 1;
        if (splits > max_splits) splits = max_splits;
    }
    if (splits < 1) splits = 1;
    const int per = cdiv(ksteps, splits);
    splits = cdiv(ksteps, per);
    *ksteps_per_split = per;
    if (big_out) *big_out = big;
    return splits;
}

}  // namespace

size_t gcc_internal_wgrad_workspace(const gcc_conv_t* c, int batch) {
    if (!c || c->Ci <= 0 || c->Co <= 0 || batch < 1) return 0;
    int per, tper;
    int splits = plan_splits(c, batch, &per);
    const int tsplits = ts_plan(c, batch, &tper);          // the tap-stationary route's split count may differ
    if (tsplits > splits) splits = tsplits;
    return (size_t)splits * batch * c->Co * c->KH * c->KW * ceil8(c->Ci) * sizeof(float);
}

// single-output-channel head (conv_igemm.hip): dW[0][tap][:] = sum_pixels G[pixel][tap*8] * x[pixel][:], a 1x1 weight
// gradient with the gathered dy matrix G as its "dy"; rows tap*8 of that [taps*8][Ci] result are the master's rows
size_t gcc_internal_head_gather_bytes(const gcc_conv_t* c);
int gcc_internal_head_gather(const gcc_conv_t* c, const void* dy, void* g, hipStream_t st);
static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
static gcc_conv_t head_conv1x1(const gcc_conv_t* c) {
    const int taps = c->KH * c->KW;
    gcc_conv_t c1 = {c->N, c->H, c->W, c->Ci, taps * 8, 1, 1, 1, 0, c->ldx, c->xoff, taps * 8, 0};
    return c1;
}
static size_t head_wgrad_workspace(const gcc_conv_t* c) {
    const size_t g = gcc_internal_head_gather_bytes(c);
    if (!g) return 0;
    const gcc_conv_t c1 = head_conv1x1(c);
    return al256(g) + al256((size_t)c1.Co * ceil8(c->Ci) * sizeof(float)) + gcc_internal_wgrad_workspace(&c1, 1);
}
__global__ void head_rows_fold_kernel(const float* __restrict__ big, float* __restrict__ dw, int taps, int Cip, int accumulate) {
    const int n = taps * Cip;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int t = i / Cip, cc = i - t * Cip;
        const float v = big[(size_t)(t * 8) * Cip + cc];
        dw[i] = accumulate ? dw[i] + v : v;
    }
}

// conv_thinout.hip: wide kernels with <= 3 output channels (SRGAN's last 9 x 9 layer)
size_t gcc_internal_thinout_wgrad_workspace(const gcc_conv_t* c);
int gcc_internal_thinout_wgrad(const gcc_conv_t* c, const void* x, const void* dy, float* dw, int accumulate, void* ws, size_t ws_bytes,
                               hipStream_t st);

extern "C" size_t gcc_conv_wgrad_workspace(const gcc_conv_t* c) {
    const size_t h = c ? head_wgrad_workspace(c) : 0;
    if (h) return h;
    const size_t t = c ? gcc_internal_thinout_wgrad_workspace(c) : 0, g = gcc_internal_wgrad_workspace(c, 1);
    return t > g ? t : g;
}

// batched form: problem b reads x + b*x_bstride, dy + b*dy_bstride (elements; c->N images each) and
// writes dw + b*Co*taps*Ci
int gcc_internal_wgrad(const gcc_conv_t* c, const void* x, const void* dy, float* dw, int accumulate, void* ws,
                       size_t ws_bytes, int batch, long x_bstride, long dy_bstride, hipStream_t st, int rows_l,
                       int cols_l, int row_split, int col_split) {
    GCC_ENTER();
    if (!c || !x || !dy || !dw || !ws) return GCC_ERR_BAD_ARG;
    if (c->N <= 0 || c->H <= 0 || c->W <= 0 || c->Ci <= 0 || c->Co <= 0 || c->KH <= 0 || c->KW <= 0 ||
        c->stride <= 0 || c->pad < 0)
        return GCC_ERR_BAD_ARG;
    if ((c->ldx & 7) || (c->xoff & 7) || (c->ldy & 7) || (c->yoff & 7)) return GCC_ERR_BAD_ARG;
    const int Ho = gcc_conv_out(c->H, c->KH, c->stride, c->pad), Wo = gcc_conv_out(c->W, c->KW, c->stride, c->pad);
    if (Ho <= 0 || Wo <= 0) return GCC_ERR_BAD_ARG;
    if (batch < 1 || ws_bytes < gcc_internal_wgrad_workspace(c, batch)) return GCC_ERR_WORKSPACE;
    WgradParams p;
    p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.out = (float*)ws;
    p.N = c->N; p.H = c->H; p.W = c->W; p.ldx = c->ldx; p.xoff = c->xoff; p.Ci = c->Ci; p.Cip = ceil8(c->Ci);
    p.Ho = Ho; p.Wo = Wo; p.ldy = c->ldy; p.yoff = c->yoff; p.Co = c->Co;
    p.KH = c->KH; p.KW = c->KW; p.stride = c->stride; p.pad = c->pad;
    p.ncols = c->KH * c->KW * p.Cip;
    const size_t M = (size_t)c->N * Ho * Wo;
    const size_t xb = (size_t)c->N * c->H * c->W * c->ldx * 2, yb = M * c->ldy * 2;
    if (xb >= OOB || yb >= OOB || M >= (1u << 30)) return GCC_ERR_UNSUPPORTED;
    p.M = (int)M; p.x_bytes = (uint32_t)xb; p.dy_bytes = (uint32_t)yb;
    bool big = false;
    const int splits = plan_splits(c, batch, &p.ksteps_per_split, &big);
    p.batch = batch; p.x_bstride = x_bstride; p.dy_bstride = dy_bstride;
    const bool seg = rows_l > 0;      // c holds physical sizes, dw is [rows_l][taps][cols_l]
    const bool regular = !seg && (c->Ci & 7) == 0 && (((uintptr_t)dw) & 15) == 0;
    big = big && regular;              // concatenated / unaligned gradients keep the 128 x 128 tiling (same split plan)
    p.direct = (splits == 1 && regular) ? 1 : 0;
    p.accumulate = accumulate; p.dw = dw;
    p.debug = gcc_diag_bits();
    {
        p.rowmode = (gcc_opt(GCC_OPT_WGRAD_ROW_TABLE) && c->KH <= 15 && c->KW <= 15) ? 1 : 0;
    }
    int tper = 0;
    const int tsplits = regular ? ts_plan(c, batch, &tper) : 0;
    if (tsplits > 0) {
        TsParams t;
        t.x = p.x; t.dy = p.dy; t.out = (float*)ws; t.dw = dw;
        t.N = c->N; t.H = c->H; t.W = c->W; t.ldx = c->ldx; t.xoff = c->xoff; t.Ci = c->Ci;
        t.Ho = Ho; t.Wo = Wo; t.ldy = c->ldy; t.yoff = c->yoff; t.Co = c->Co;
        t.ncols = p.ncols;
        t.ci_tiles = c->Ci / 64; t.co_tiles = c->Co / 64;
        t.TY = cdiv(Ho, ts::TR); t.TX = cdiv(Wo, ts::TW);
        t.blocks = c->N * t.TY * t.TX; t.blocks_per_split = tper;
        t.x_bytes = p.x_bytes; t.dy_bytes = p.dy_bytes;
        t.dT = make_fastdiv(t.TY * t.TX); t.dTX = make_fastdiv(t.TX);
        t.direct = tsplits == 1 ? 1 : 0; t.accumulate = accumulate;
        static std::once_flag ts_once;
        static bool ts_attr_ok = false;
        std::call_once(ts_once, [] {
            ts_attr_ok = hipFuncSetAttribute((const void*)wgrad_ts_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, TsCfg<4>::LDS_BYTES) == hipSuccess &&
                         hipFuncSetAttribute((const void*)wgrad_ts_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, TsCfg<3>::LDS_BYTES) == hipSuccess;
        });
        if (!ts_attr_ok) return GCC_ERR_LAUNCH;
        if (c->KH == 4)
            hipLaunchKernelGGL(wgrad_ts_kernel<4>, dim3(t.ci_tiles * t.co_tiles, 1, tsplits), dim3(TsCfg<4>::NT), TsCfg<4>::LDS_BYTES, st, t);
        else
            hipLaunchKernelGGL(wgrad_ts_kernel<3>, dim3(t.ci_tiles * t.co_tiles, 1, tsplits), dim3(TsCfg<3>::NT), TsCfg<3>::LDS_BYTES, st, t);
        GCC_CHECK_LAUNCH();
        if (t.direct) return GCC_OK;
        const size_t n4 = (size_t)c->Co * p.ncols / 4;
        if (tsplits <= 8) {
            int blocks = (int)((n4 + 255) / 256);
            if (blocks > 8192) blocks = 8192;
            hipLaunchKernelGGL(wgrad_reduce_flat_kernel, dim3(blocks), dim3(256), 0, st, (const float*)ws, dw, tsplits, n4, accumulate);
        } else {
            int blocks = (int)((n4 + 63) / 64);
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL(wgrad_reduce_vec_kernel, dim3(blocks), dim3(256), 0, st, (const float*)ws, dw, tsplits, n4, accumulate);
        }
        GCC_CHECK_LAUNCH();
        return GCC_OK;
    }
    const int tcol = big ? 256 : 128;
    p.col_tiles = cdiv(p.ncols, tcol); p.co_tiles = cdiv(c->Co, tcol);
    p.dHW = make_fastdiv(Ho * Wo); p.dW = make_fastdiv(Wo); p.dCip = make_fastdiv(p.Cip); p.dKW = make_fastdiv(c->KW);
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        hipFuncSetAttribute((const void*)wgrad_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, WCfg<false>::LDS_BYTES);
        hipFuncSetAttribute((const void*)wgrad_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, WCfg<true>::LDS_BYTES);
        hipFuncSetAttribute((const void*)wgrad_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, WCfg<false>::LDS_BYTES);
        hipFuncSetAttribute((const void*)wgrad_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, WCfg<true>::LDS_BYTES);
    });
    const dim3 grid(p.col_tiles * p.co_tiles, batch, splits);
    if (big && p.rowmode)
        hipLaunchKernelGGL((wgrad_kernel<true, true>), grid, dim3(WCfg<true>::NT), WCfg<true>::LDS_BYTES, st, p);
    else if (big)
        hipLaunchKernelGGL((wgrad_kernel<true, false>), grid, dim3(WCfg<true>::NT), WCfg<true>::LDS_BYTES, st, p);
    else if (p.rowmode)
        hipLaunchKernelGGL((wgrad_kernel<false, true>), grid, dim3(WCfg<false>::NT), WCfg<false>::LDS_BYTES, st, p);
    else
        hipLaunchKernelGGL((wgrad_kernel<false, false>), grid, dim3(WCfg<false>::NT), WCfg<false>::LDS_BYTES, st, p);
    GCC_CHECK_LAUNCH();
    if (p.direct) return GCC_OK;
    const size_t total = (size_t)batch * c->Co * c->KH * c->KW * c->Ci;
    if (regular) {
        const size_t n4 = total / 4;
        if (splits <= 8) {
            int blocks = (int)((n4 + 255) / 256);
            if (blocks > 8192) blocks = 8192;
            hipLaunchKernelGGL(wgrad_reduce_flat_kernel, dim3(blocks), dim3(256), 0, st, (const float*)ws, dw, splits, n4,
                               accumulate);
        } else {
            int blocks = (int)((n4 + 63) / 64);
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL(wgrad_reduce_vec_kernel, dim3(blocks), dim3(256), 0, st, (const float*)ws, dw, splits, n4,
                               accumulate);
        }
    } else {
        const size_t outs = seg ? (size_t)rows_l * c->KH * c->KW * cols_l : total;
        const bool deep = splits >= 32;           // many slabs, few outputs: 16 outputs x 16 split lanes
        int blocks = (int)((outs + (deep ? 15 : 63)) / (deep ? 16 : 64));
        if (blocks > 4096) blocks = 4096;
        const int rows = seg ? rows_l : batch * c->Co, cols = seg ? cols_l : c->Ci, cop = seg ? c->Co : batch * c->Co;
        const int rsp = seg ? row_split : 0, csp = seg ? col_split : 0;
        if (deep)
            hipLaunchKernelGGL((wgrad_reduce_kernel<16, 16>), dim3(blocks), dim3(256), 0, st, (const float*)ws, dw, splits, rows,
                               c->KH * c->KW, cols, p.Cip, accumulate, cop, rsp, csp);
        else
            hipLaunchKernelGGL((wgrad_reduce_kernel<64, 4>), dim3(blocks), dim3(256), 0, st, (const float*)ws, dw, splits, rows,
                               c->KH * c->KW, cols, p.Cip, accumulate, cop, rsp, csp);
    }
    GCC_CHECK_LAUNCH();
    return GCC_OK;
}

extern "C" int gcc_conv_wgrad(const gcc_conv_t* c, const void* x, const void* dy, float* dw, int accumulate,
                              void* ws, size_t ws_bytes, gcc_stream_t stream) {
    GCC_ENTER();
    const size_t hw = c ? head_wgrad_workspace(c) : 0;
    if (hw && ws && ws_bytes >= hw && x && dy && dw && (c->Ci & 7) == 0) {
        hipStream_t st = (hipStream_t)stream;
        char* base = (char*)ws;
        const gcc_conv_t c1 = head_conv1x1(c);
        const size_t gbytes = al256(gcc_internal_head_gather_bytes(c));
        const size_t bbytes = al256((size_t)c1.Co * c->Ci * sizeof(float));
        int rc = gcc_internal_head_gather(c, dy, base, st);
        if (rc) return rc;
        float* big = (float*)(base + gbytes);
        rc = gcc_internal_wgrad(&c1, x, base, big, 0, base + gbytes + bbytes, ws_bytes - gbytes - bbytes, 1, 0, 0, st, 0, 0, 0, 0);
        if (rc) return rc;
        const int taps = c->KH * c->KW;
        hipLaunchKernelGGL(head_rows_fold_kernel, dim3((taps * c->Ci + 255) / 256), dim3(256), 0, st, (const float*)big, dw, taps, c->Ci,
                           accumulate);
        GCC_CHECK_LAUNCH();
        return GCC_OK;
    }
    if (c && x && dy && dw && ws && (((uintptr_t)dw) & 3) == 0) {
        const int rc = gcc_internal_thinout_wgrad(c, x, dy, dw, accumulate, ws, ws_bytes, (hipStream_t)stream);
        if (rc != GCC_ERR_UNSUPPORTED) return rc;
    }
    return gcc_internal_wgrad(c, x, dy, dw, accumulate, ws, ws_bytes, 1, 0, 0, (hipStream_t)stream, 0, 0, 0, 0);
}

extern "C" int gcc_conv_wgrad_seg(const gcc_conv_t* c, const void* x, const void* dy, float* dw, int rows, int cols,
                                  int row_split, int col_split, int accumulate, void* ws, size_t ws_bytes,
                                  gcc_stream_t stream) {
    GCC_ENTER();
    if (!c || rows <= 0 || cols <= 0) return GCC_ERR_BAD_ARG;
    if (c->Co != seg_phys_size(rows, row_split) || c->Ci != seg_phys_size(cols, col_split)) return GCC_ERR_BAD_ARG;
    return gcc_internal_wgrad(c, x, dy, dw, accumulate, ws, ws_bytes, 1, 0, 0, (hipStream_t)stream, rows, cols, row_split,
                              col_split);
}

// ---- grouped weight gradients: host side ------------------------------------------------------------------------------------------
namespace {
struct GroupPlan {
    int n, total_wgs;
    int per[GROUP_MAX], splits[GROUP_MAX], tiles[GROUP_MAX], order[GROUP_MAX];
    size_t slab_off[GROUP_MAX], ws_bytes;
};
int group_cus() {
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
bool group_item_ok(const gcc_wgrad_item_t* it) {
    const gcc_conv_t* c = &it->c;
    if (c->N <= 0 || c->H <= 0 || c->W <= 0 || c->Ci <= 0 || c->Co <= 0 || c->KH <= 0 || c->KW <= 0 || c->stride <= 0 || c->pad < 0) return false;
    if ((c->Ci & 7) || (c->ldx & 7) || (c->xoff & 7) || (c->ldy & 7) || (c->yoff & 7)) return false;      // regular widths only
    if (c->KH > 15 || c->KW > 15) return false;                       // the pixel table's tap masks
    const int Ho = gcc_conv_out(c->H, c->KH, c->stride, c->pad), Wo = gcc_conv_out(c->W, c->KW, c->stride, c->pad);
    if (Ho <= 0 || Wo <= 0) return false;
    const size_t M = (size_t)c->N * Ho * Wo;
    if ((size_t)c->N * c->H * c->W * c->ldx * 2 >= OOB || M * c->ldy * 2 >= OOB || M >= (1u << 30)) return false;
    if (head_wgrad_workspace(c) || gcc_internal_thinout_wgrad_workspace(c)) return false;      // routes of their own
    int tper = 0;
    if (ts_plan(c, 1, &tper) > 0) return false;                       // the tap-stationary route: a launch of its own
    return true;
}
// Pixel splits for the group as a whole: every workgroup gets about S k-steps, S such that the group is ~2 rounds of the chip's
// two-per-CU slots (16 <= S <= 256: below, a workgroup is all prologue; above, the tail of the launch is one long workgroup -- and
// every split is a slab of fp32 traffic: SRGAN's 32 trunk layers at 96 x 96 x 16 images would write 170 MB with S = 64)
int group_plan(const gcc_wgrad_item_t* items, int n, GroupPlan* g) {
    if (!items || n < 1 || n > GROUP_MAX) return GCC_ERR_BAD_ARG;
    long work = 0;
    int ksteps[GROUP_MAX];
    for (int i = 0; i < n; i++) {
        if (!group_item_ok(&items[i])) return GCC_ERR_UNSUPPORTED;
        const gcc_conv_t* c = &items[i].c;
        const long M = (long)c->N * gcc_conv_out(c->H, c->KH, c->stride, c->pad) * gcc_conv_out(c->W, c->KW, c->stride, c->pad);
        ksteps[i] = (int)((M + TP - 1) / TP);
        g->tiles[i] = cdiv(c->KH * c->KW * c->Ci, 128) * cdiv(c->Co, 128);
        work += (long)g->tiles[i] * ksteps[i];
    }
    long S = cdiv(work, (long)4 * group_cus());
    S = S < 16 ? 16 : (S > 256 ? 256 : S);
    g->n = n; g->total_wgs = 0; g->ws_bytes = 0;
    for (int i = 0; i < n; i++) {
        int splits = cdiv(ksteps[i], (int)S);
        const int per = cdiv(ksteps[i], splits);
        splits = cdiv(ksteps[i], per);
        g->per[i] = per; g->splits[i] = splits;
        g->total_wgs += g->tiles[i] * splits;
        g->slab_off[i] = g->ws_bytes;
        if (splits > 1) {
            const gcc_conv_t* c = &items[i].c;
            g->ws_bytes += al256((size_t)splits * c->Co * c->KH * c->KW * c->Ci * sizeof(float));
        }
        g->order[i] = i;
    }
    // longest workgroups first (insertion sort, stable: ties keep the caller's order)
    for (int a = 1; a < n; a++) {
        const int v = g->order[a];
        int b = a;
        while (b > 0 && g->per[g->order[b - 1]] < g->per[v]) { g->order[b] = g->order[b - 1]; b--; }
        g->order[b] = v;
    }
    if (g->ws_bytes == 0) g->ws_bytes = 256;          // (a group without a split entry still gets a non-null workspace)
    return GCC_OK;
}
}  // namespace

extern "C" size_t gcc_conv_wgrad_group_table_bytes(void) { return sizeof(WgradGroupTable); }

extern "C" size_t gcc_conv_wgrad_group_workspace(const gcc_wgrad_item_t* items, int n) {
    GroupPlan g;
    return group_plan(items, n, &g) == GCC_OK ? g.ws_bytes : 0;
}

extern "C" int gcc_conv_wgrad_group_prepare(const gcc_wgrad_item_t* items, int n, void* ws, size_t ws_bytes, void* table_host) {
    GroupPlan g;
    const int rc = group_plan(items, n, &g);
    if (rc) return rc;
    if (!table_host || !ws || (((uintptr_t)ws) & 15)) return GCC_ERR_BAD_ARG;
    if (ws_bytes < g.ws_bytes) return GCC_ERR_WORKSPACE;
    WgradGroupTable* t = (WgradGroupTable*)table_host;
    memset(t, 0, sizeof(*t));
    t->magic = GROUP_MAGIC; t->n_items = n; t->total_wgs = g.total_wgs;
    int wg = 0, nred = 0, rblk = 0;
    for (int k = 0; k < n; k++) {
        const int i = g.order[k];
        const gcc_wgrad_item_t* it = &items[i];
        if (!it->x || !it->dy || !it->dw || (((uintptr_t)it->dw) & 15)) return GCC_ERR_BAD_ARG;
        const gcc_conv_t* c = &it->c;
        const int Ho = gcc_conv_out(c->H, c->KH, c->stride, c->pad), Wo = gcc_conv_out(c->W, c->KW, c->stride, c->pad);
        WgradParams& p = t->items[k];
        p.x = (const bf16_t*)it->x; p.dy = (const bf16_t*)it->dy;
        p.out = (float*)((char*)ws + g.slab_off[i]);
        p.N = c->N; p.H = c->H; p.W = c->W; p.ldx = c->ldx; p.xoff = c->xoff; p.Ci = c->Ci; p.Cip = c->Ci;
        p.Ho = Ho; p.Wo = Wo; p.ldy = c->ldy; p.yoff = c->yoff; p.Co = c->Co;
        p.KH = c->KH; p.KW = c->KW; p.stride = c->stride; p.pad = c->pad;
        p.ncols = c->KH * c->KW * c->Ci;
        const size_t M = (size_t)c->N * Ho * Wo;
        p.M = (int)M;
        p.x_bytes = (uint32_t)((size_t)c->N * c->H * c->W * c->ldx * 2); p.dy_bytes = (uint32_t)(M * c->ldy * 2);
        p.ksteps_per_split = g.per[i];
        p.col_tiles = cdiv(p.ncols, 128); p.co_tiles = cdiv(c->Co, 128);
        p.dHW = make_fastdiv(Ho * Wo); p.dW = make_fastdiv(Wo); p.dCip = make_fastdiv(p.Cip); p.dKW = make_fastdiv(c->KW);
        p.batch = 1; p.x_bstride = 0; p.dy_bstride = 0;
        p.direct = g.splits[i] == 1 ? 1 : 0;
        p.accumulate = it->accumulate; p.dw = it->dw;
        p.rowmode = 1; p.debug = 0;
        t->wg_prefix[k] = wg;
        t->tiles[k] = g.tiles[i];
        wg += g.tiles[i] * g.splits[i];
        if (g.splits[i] > 1) {
            GroupReduce& r = t->red[nred++];
            r.slabs = p.out; r.dw = it->dw; r.splits = g.splits[i]; r.accumulate = it->accumulate;
            r.n4 = (unsigned)((size_t)c->Co * p.ncols / 4);
            r.blk0 = rblk;
            rblk += (int)((r.n4 + 255) / 256);
        }
    }
    t->wg_prefix[n] = wg;
    t->n_reduce = nred; t->reduce_blocks = rblk;
    return GCC_OK;
}

extern "C" int gcc_conv_wgrad_group_run(const void* table_dev, const void* table_host, gcc_stream_t stream) {
    GCC_ENTER();
    const WgradGroupTable* t = (const WgradGroupTable*)table_host;
    if (!table_dev || !t || t->magic != GROUP_MAGIC || t->n_items < 1 || t->n_items > GROUP_MAX || t->total_wgs < 1) return GCC_ERR_BAD_ARG;
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute((const void*)wgrad_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WCfg<false>::LDS_BYTES);
    });
    if (attr_err != hipSuccess) return GCC_ERR_LAUNCH;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(wgrad_group_kernel, dim3(t->total_wgs), dim3(WCfg<false>::NT), WCfg<false>::LDS_BYTES, st, (const WgradGroupTable*)table_dev);
    GCC_CHECK_LAUNCH();
    if (t->reduce_blocks > 0) {
        hipLaunchKernelGGL(wgrad_group_reduce_kernel, dim3(t->reduce_blocks), dim3(256), 0, st, (const WgradGroupTable*)table_dev);
        GCC_CHECK_LAUNCH();
    }
    return GCC_OK;
}
