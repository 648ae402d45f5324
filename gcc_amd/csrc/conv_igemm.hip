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
#include "common.hpp"

namespace {

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
};

constexpr int BP = 128;  // pixels per tile
constexpr int BK = 64;   // k per step
constexpr uint32_t OOB = 0x7FFFFFF0u;

template <int BC>
struct Cfg {
    static constexpr int WC = (BC >= 128) ? 2 : 1;        // waves along channels
    static constexpr int WP = 4 / WC;                      // waves along pixels
    static constexpr int TC = BC / WC;                     // channels per wave
    static constexpr int TP = BP / WP;                     // pixels per wave
    static constexpr int CB = TC / 16;
    static constexpr int PB = TP / 16;
    static constexpr int W_CHUNKS = (BC * 8 + 255) / 256;  // 16-B weight chunks per thread per step
    static constexpr int LDS_BYTES_LOOP = 2 * (BP + BC) * BK * 2;
    static constexpr int OSTRIDE = BC * 2 + 16;            // epilogue tile row stride (bytes)
    static constexpr int LDS_BYTES_EPI = BP * OSTRIDE + 2 * 256 * 4;
    static constexpr int LDS_BYTES = LDS_BYTES_LOOP > LDS_BYTES_EPI ? LDS_BYTES_LOOP : LDS_BYTES_EPI;
};

template <int BC>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p) {
    using C = Cfg<BC>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                          // pixels  [2][BP][128 B]
    char* sW = smem + 2 * BP * BK * 2;        // weights [2][BC][128 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wc = wave % C::WC;
    const int wp = wave / C::WC;

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
    const int nk = (Ktot + BK - 1) / BK;

    const int nwg = gridDim.x;
    const int tile = xcd_remap(blockIdx.x, nwg);
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

    const bf16_t* srcp = p.src + (size_t)blockIdx.y * p.src_bstride;
    const bf16_t* wgtp = p.wgt + (size_t)blockIdx.y * p.wgt_bstride;
    bf16_t* dstp = p.dst + (size_t)blockIdx.y * p.dst_bstride;
    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)srcp, 0, p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wgt = __builtin_amdgcn_make_buffer_rsrc((void*)wgtp, 0, p.wgt_bytes, 0x00020000);

    // ---- per-thread gather rows (4 pixel rows, fixed 16-B chunk column) ----------------------
    const int chunk = tid & 7;
    int a_base[4], a_iy[4], a_ix[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int m = m0 + (tid >> 3) + 32 * i;
        if (m < M) {
            const int n = m / (Hg * Wg);
            const int r = m - n * (Hg * Wg);
            const int oy = r / Wg;
            const int ox = r - oy * Wg;
            a_base[i] = n * p.Hs * p.Ws;
            a_iy[i] = oy * sy + dy0;
            a_ix[i] = ox * sy + dx0;
        } else {
            a_base[i] = 0; a_iy[i] = -(1 << 28); a_ix[i] = 0;   // always out of range -> zeros
        }
    }
    // weight rows of this thread
    int w_row[C::W_CHUNKS];
#pragma unroll
    for (int i = 0; i < C::W_CHUNKS; i++) w_row[i] = (tid >> 3) + 32 * i;

    const bool uniform_tap = (p.Ct % BK) == 0;

    i32x4 ra[4];
    i32x4 rw[C::W_CHUNKS];

    auto issue_loads = [&](int kt) {
        const int k = kt * BK + chunk * 8;
        int tap, cc;
        if (uniform_tap) {
            const int kb = kt * BK;
            tap = kb / p.Ct;               // wave-uniform
            cc = kb - tap * p.Ct + chunk * 8;
        } else {
            tap = k / p.Ct;
            cc = k - tap * p.Ct;
        }
        const bool kval = k < Ktot;
        const int a = tap / TB;
        const int b = tap - a * TB;
        const int dy = a * dstep;
        const int dx = b * dstep;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int iy = a_iy[i] + dy;
            const int ix = a_ix[i] + dx;
            const bool ok = kval && (unsigned)iy < (unsigned)p.Hs && (unsigned)ix < (unsigned)p.Ws;
            const uint32_t off = ok ? (uint32_t)(((a_base[i] + iy * p.Ws + ix) * p.lds_ + p.soff + cc) * 2) : OOB;
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, off, 0, 0);
        }
        const int wtap = (kh0 + a * kstep) * p.KW + (kw0 + b * kstep);
#pragma unroll
        for (int i = 0; i < C::W_CHUNKS; i++) {
            const int row = n0 + w_row[i];
            const bool ok = kval && row < p.Cout && (BC >= 32 || w_row[i] < BC);
            const uint32_t off = ok ? (uint32_t)((row * p.ldw + wtap * p.Ct + cc) * 2) : OOB;
            rw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_wgt, off, 0, 0);
        }
    };

    auto write_lds = [&](int stage) {
        char* a = sA + stage * (BP * BK * 2);
        char* w = sW + stage * (BC * BK * 2);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = (tid >> 3) + 32 * i;
            *(i32x4*)(a + row * 128 + ((chunk ^ (row & 7)) << 4)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < C::W_CHUNKS; i++) {
            const int row = w_row[i];
            if (BC >= 32 || row < BC) *(i32x4*)(w + row * 128 + ((chunk ^ (row & 7)) << 4)) = rw[i];
        }
    };

    f32x4 acc[C::CB][C::PB];
#pragma unroll
    for (int i = 0; i < C::CB; i++)
#pragma unroll
        for (int j = 0; j < C::PB; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- pipeline: registers hold tile t+1 while LDS[t&1] is consumed --------------------------
    issue_loads(0);
    write_lds(0);
    if (nk > 1) issue_loads(1);
    __syncthreads();

    const int lr = lane & 15;
    const int lq = lane >> 4;
    for (int kt = 0; kt < nk; kt++) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            write_lds(cur ^ 1);                 // tile kt+1 (its loads were issued one step ago)
            if (kt + 2 < nk) issue_loads(kt + 2);
        }
        const char* a = sA + cur * (BP * BK * 2);
        const char* w = sW + cur * (BC * BK * 2);
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8 fw[C::CB], fa[C::PB];
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
#pragma unroll
            for (int i = 0; i < C::CB; i++)
#pragma unroll
                for (int j = 0; j < C::PB; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue ------------------------------------------------------------------------------
    // acc[i][j][r]: channel = wc*TC + i*16 + 4*lq + r ; pixel = wp*TP + j*16 + lr
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
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = apply_act(acc[i][j][r] + bv[r], p.act, p.slope);
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
    for (int q = tid; q < NCH; q += 256) {
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
        constexpr int PARTS = 256 / BC < 1 ? 1 : 256 / BC;
        constexpr int ROWS = BP / PARTS;
        const int c = tid % BC;
        const int part = tid / BC;
        float s = 0.f, ss = 0.f;
        if (BC >= 256 || part < PARTS) {
            for (int r = part * ROWS; r < (part + 1) * ROWS; r++) {
                const float v = bf2f(*(const bf16_t*)(sO + r * C::OSTRIDE + c * 2));
                s += v; ss += v * v;
            }
        }
        sR[tid] = s; sR[256 + tid] = ss;
        __syncthreads();
        if (tid < BC && n0 + tid < p.Cout) {
            float ts = 0.f, tss = 0.f;
#pragma unroll
            for (int q = 0; q < PARTS; q++) { ts += sR[q * BC + tid]; tss += sR[256 + q * BC + tid]; }
            const int trow = blockIdx.z * p.mtiles_max + mt;
            p.stats[((size_t)trow * 2 + 0) * p.Cout + n0 + tid] = ts;
            p.stats[((size_t)trow * 2 + 1) * p.Cout + n0 + tid] = tss;
        }
    }
}

template <int BC>
int launch(const IgemmParams& p, int phases, int batch, hipStream_t st) {
    using C = Cfg<BC>;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)igemm_kernel<BC>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        attr_set = true;
    }
    dim3 grid(p.mtiles_max * p.ntiles, batch, phases);
    hipLaunchKernelGGL(igemm_kernel<BC>, grid, dim3(256), C::LDS_BYTES, st, p);
    GCC_CHECK_LAUNCH();
    return GCC_OK;
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

}  // namespace

// internal entry (also used by distill.hip): `batch` independent problems, strides in elements
int gcc_internal_igemm(const gcc_conv_t* c, int dgrad, const void* src, const void* w, void* dst, const gcc_epilogue_t* ep,
                       int batch, long src_bstride, long wgt_bstride, long dst_bstride, hipStream_t st) {
    GCC_ENTER();
    int rc = check_conv(c);
    if (rc) return rc;
    if (!src || !w || !dst) return GCC_ERR_BAD_ARG;
    const int Ho = gcc_conv_out(c->H, c->KH, c->stride, c->pad);
    const int Wo = gcc_conv_out(c->W, c->KW, c->stride, c->pad);
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
    p.mtiles_max = (int)((max_rows + BP - 1) / BP);
    p.src_bstride = src_bstride; p.wgt_bstride = wgt_bstride; p.dst_bstride = dst_bstride;
    if (batch < 1 || (batch > 1 && p.stats)) return GCC_ERR_BAD_ARG;
    const int phases = dgrad ? c->stride * c->stride : 1;
    if (dgrad && (c->KH < c->stride || c->KW < c->stride)) return GCC_ERR_UNSUPPORTED;
    if (p.Cout > 64) { p.ntiles = cdiv(p.Cout, 128); return launch<128>(p, phases, batch, st); }
    if (p.Cout > 32) { p.ntiles = 1; return launch<64>(p, phases, batch, st); }
    if (p.Cout > 16) { p.ntiles = 1; return launch<32>(p, phases, batch, st); }
    p.ntiles = 1;
    return launch<16>(p, phases, batch, st);
}

extern "C" int gcc_conv_stat_tiles(const gcc_conv_t* c, int dgrad) {
    GCC_ENTER();
    if (check_conv(c)) return 0;
    if (!dgrad) {
        const int Ho = gcc_conv_out(c->H, c->KH, c->stride, c->pad), Wo = gcc_conv_out(c->W, c->KW, c->stride, c->pad);
        return (int)(((size_t)c->N * Ho * Wo + BP - 1) / BP);
    }
    const int s = c->stride;
    return (int)(((size_t)c->N * cdiv(c->H, s) * cdiv(c->W, s) + BP - 1) / BP) * s * s;
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
