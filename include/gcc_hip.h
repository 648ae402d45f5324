/*
 * gcc_hip.h -- C ABI of libgcc_hip.so: the MI355X (gfx950) kernels under the GCC training step.
 *
 * The reference (SJLeo/GCC) has no FFI layer: its hot path dispatches torch/ATen operators from
 * models/Pix2Pix.py.  Each entry point below names the reference call site(s) whose ATen operator
 * it replaces (paths relative to the reference root).  Conventions:
 *   - plain C, no torch types; every pointer is a DEVICE pointer owned by the caller (activations,
 *     weights, workspaces); the library allocates nothing.  What a launch does is decided by its arguments
 *     (the tile plan travels in gcc_conv_t.plan); the only process-wide state is the table of A/B tuning hooks below
 *     (gcc_set_option: atomics, defaults read once from GCC_* environment variables; the product leaves them alone), the one-time hipFuncSetAttribute of the kernels that use > 64 KB of LDS, and the RCCL entry
 *     points resolved on the first gcc_comm_* call (communicators themselves are explicit objects the caller owns);
 *   - every kernel is enqueued on the caller's `stream` and never synchronises;
 *   - return value: 0 = GCC_OK, negative = error (see gcc_strerror); no exceptions, no abort;
 *   - activations are NHWC bf16 (a PyTorch channels_last tensor): element (n,h,w,c) lives at
 *     ((n*H+h)*W+w)*ld + off + c, `ld` (pixel stride, elements) and `off` multiples of 8, and the
 *     bytes up to the next multiple of 8 channels are readable and zero;
 *   - conv weights are bf16 in two packings made by gcc_pack_weights from the fp32 master
 *     (a channels_last nn.Parameter, physical [Co][KH][KW][Ci]):  W  = [Co][KH*KW][Cip]  (fprop)
 *     and Wt = [Ci][KH*KW][Cop] (dgrad), Cip/Cop = channels rounded up to 8, zero filled;
 *   - fp32 everywhere else (statistics, losses, gradients of parameters, optimizer state).
 */
#ifndef GCC_HIP_H
#define GCC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gcc_stream_t; /* hipStream_t */

enum {
    GCC_OK = 0,
    GCC_ERR_BAD_ARG = -1,      /* null pointer / non-positive size / misaligned ld or offset */
    GCC_ERR_UNSUPPORTED = -2,  /* geometry outside what the kernels implement */
    GCC_ERR_WORKSPACE = -3,    /* workspace too small */
    GCC_ERR_LAUNCH = -4        /* hipGetLastError() after launch was not hipSuccess */
};

enum { GCC_ACT_NONE = 0, GCC_ACT_LRELU = 1, GCC_ACT_RELU = 2, GCC_ACT_TANH = 3 };

/* ABI generation of this header: bumped whenever a struct layout, an enum numbering or a prototype below changes.  gcc_version()
 * of the library a host loads must return exactly this number (gcc_amd/_lib.py refuses any other; an external host should check it
 * the same way): a stale .so reads gcc_conv_t.plan past its struct and sets the wrong option ids without any error. */
#define GCC_HIP_ABI 603

const char* gcc_strerror(int code);
int gcc_version(void); /* == GCC_HIP_ABI of the header the library was built from */
/* kernel launches the library has made in this process so far (reset != 0: return the count and start again from zero) */
long long gcc_launch_count(int reset);
/* Device-side error word.  The kernels that wait for other workgroups inside a launch (the grid InstanceNorm's tagged exchange,
 * gcc_inorm_fwd / gcc_inorm_bwd / gcc_bn_bwd_one_launch) bound every spin; a spin that expires stores a non-zero code into a
 * pinned host word, the launch's results are wrong, and every later call of those entry points returns GCC_ERR_LAUNCH.
 * Returns the word (0: no error; 0x1401: InstanceNorm exchange timed out); clear != 0 resets it.  Never synchronises. */
int gcc_device_error(int clear);

/* ---------------------------------------------------------------------------------------------
 * Tuning hooks: which kernel family a geometry is routed to.  Every option only selects between kernels that compute the same
 * result (the parity tests run the cases under each).  These are A/B switches for measurements and tests -- NOT the way a host
 * chooses its schedule: everything the model classes switch at run time (tile families, pair split, halo columns, weight-
 * gradient split targets) travels with the call in gcc_conv_t.plan, and the product runs with every hook at its default
 * (bench.py prints the vector and refuses to run otherwise).  Defaults come from the environment variable of the same name
 * (GCC_<NAME>) when it is set at first use.  gcc_set_option(id, value): value < 0 restores the default; returns the previous
 * value (>= 0) or GCC_ERR_BAD_ARG.  Process-wide atomics: set them while nothing is being launched.
 * The shipped library holds no diagnostic ablation: the "results are wrong" switches of the earlier rounds (main loops without
 * staging loads, a grid exchange made to time out) exist only in the GCC_DIAG_BUILD variant (csrc/build.sh: libgcc_hip_diag.so,
 * same ABI plus gcc_diag_set(bits)), which tests and probes load explicitly through GCC_HIP_LIB.
 * ------------------------------------------------------------------------------------------- */
enum {
    GCC_OPT_IGEMM_GLDS = 0,       /* 1 (default): LDS-DMA staging; 0: register-staged 128-pixel tiles */
    GCC_OPT_IGEMM_HEAD,         /* 1 (default): single-output-channel head route */
    GCC_OPT_IGEMM_THIN,         /* 1 (default): thin image-layer kernels; 2: without the LDS-staged wide (65..128 channel) data-gradient form; 0: none */
    GCC_OPT_WGRAD_BIG,          /* 1 (default): 256x256 weight-gradient tiles on the large layers */
    GCC_OPT_BN_SWEEPS,          /* 0 (default): per-kernel choice of sweeps per streaming workgroup */
    GCC_OPT_BN_MAXBLK,          /* cap on streaming workgroups (default 2048) */
    GCC_OPT_BN_REDUCE_THREADS,  /* 256 (default) or 1024 threads per BatchNorm-backward reduce workgroup */
    GCC_OPT_BN_REDUCE_CAP,      /* cap on those workgroups (default 1024) */
    GCC_OPT_INORM_LPP,          /* 0 (default): automatic lanes per pixel of the one-launch InstanceNorm */
    GCC_OPT_IGEMM_FORCE_BC,     /* tuning: 0 (default) automatic; 16 / 32 / 64 / 128: channel width of the 128-pixel tiles */
    GCC_OPT_IGEMM_FORCE_KSPLIT, /* tuning: 0 (default) automatic; n >= 1: K slices of a 128-pixel-tile launch (1 = never split) */
    GCC_OPT_IGEMM_NARROW,       /* 1 (default): 128-pixel tiles narrow to 64 / 32 channels until the launch has >= 256 workgroups, and
                                   K is split only for loops of >= 48 steps (0: the round-1 plan) */
    GCC_OPT_WGRAD_BIG_MIN_TILES,/* minimum number of 256x256 output tiles for the big weight-gradient tiling (default 8: from the PatchGAN's 128 -> 256 layer up; 32 until round 6: +0.3 % on the step, profiles/r6_ab_wgrad_big_min_tiles.txt) */
    GCC_OPT_FUSE_BN,            /* gcc_conv_bn_act: 3 (default; profiles/r4_summary.md): a split layer's K slices folded, statistics exchanged inside the
                                   launch and rows normalised by one kernel on the whole chip (bn_fold_grid_kernel; needs gcc_bn_t.tail_ws; other
                                   layers as 1); 2: BatchNorm finalized by the last-arriving workgroups of the launch that writes the statistic
                                   rows (gcc_bn_t.tail_ws), split layers folded by one full-chip kernel; 1: the round-2 form (a split layer's
                                   partials, statistics, finalize and normalise in one kernel of C / 8 workgroups); 0: separate launches */
    GCC_OPT_BN_BWD_SMALL,       /* 1 (default): gcc_bnact_bwd of <= 4096 pixels (training BatchNorm, no gate) runs as one kernel instead of
                                   three, gcc_channel_sum of <= 16384 pixels as one instead of two */
    GCC_OPT_WGRAD_ROW_TABLE,    /* 1 (default): the weight-gradient kernel decomposes each pixel of a workgroup's range once, into an LDS table
                                   (gather base + validity mask), instead of in every lane at every k-step (0: the round-2 form;
                                   kernels taller or wider than 15 taps always take that form) */
    GCC_OPT_IGEMM_HALO,         /* 3 (default): also k3 s1 p1 convolutions on a 16-divisible grid (9 taps per staged slice: the VGG19 layers of SRGAN's
                                   perceptual loss, 10-25 % per layer, profiles/r4as_halo_3x3.txt); 2: also k4 s1 p1 convolutions on a 16-divisible grid (16 taps per staged slice); 1: k4 s2 p1 convolutions whose geometry fits (channels per tap a multiple of 64, 256 output channels
                                   per tile, output rows that tile 256 pixels) stage each 64-channel slice of the input neighbourhood of a
                                   256-pixel tile ONCE in LDS and serve the taps that share it from there (conv_halo.hip); 0: the gather
                                   kernel re-stages the pixels for every tap */
    GCC_OPT_FUSE_BN_PARTIAL_KB, /* gcc_conv_bn_act: cap (KB of fp32 partial tiles, default 4096) on the K split of the layers whose fold + statistics +
                                   normalise run as one kernel: every slice is another copy of the output that kernel reads back */
    GCC_OPT_INORM_GRID,         /* 1 (default): gcc_inorm_fwd / _bwd with a workspace split an image's plane over workgroups (in-launch barrier); 0: slab kernels */
    GCC_OPT_IGEMM_STAGES,       /* 3 (default): the 128-pixel x 32 / 64-column tiles (uniform taps) keep two k-steps of LDS-DMA in flight behind
                                   the one being multiplied (three LDS stages); 2: one (the round-1 loop) */
    GCC_OPT_WGRAD_TS,           /* 1 (default): k4 s1 p1 weight gradients with channels in multiples of 64, >= 32 channel tiles and >= 16 pixel
                                   blocks per split, and k3 s1 p1 ones with channels in multiples of 64 and >= 128 workgroups of >= 8 blocks, run
                                   tap-stationary (wgrad_ts_kernel<4 / 3>: the 16 / 9 taps share one staged halo image of the input);
                                   2: wherever the geometry fits (tests); 0: wgrad_kernel only */
    GCC_OPT_HALO_XCD_COLS,      /* tile order of igemm_halo_kernel: 0: a pixel tile's column tiles are neighbours on one XCD (its L2 fetches the pixel
                                   slices once, every XCD streams all the weights); 1 (default): the stride-1 form gives every XCD one column tile
                                   (1 / ntiles of the weights per L2, pixel slices fetched by ntiles XCDs: the L4 forward fetches 108 MB instead of
                                   152 MB, same duration -- profiles/r4q_halo_xcd_cols.txt); 2: every form */
    GCC_OPT_COUNT_
};
int gcc_set_option(int id, int value);
int gcc_get_option(int id);
int gcc_options_default(void);      /* 1: every hook holds its built-in default */


/* ---------------------------------------------------------------------------------------------
 * Convolution geometry.  One descriptor serves Conv2d and ConvTranspose2d: a ConvTranspose2d
 * (models/Pix2Pix.py:40-56) is described by the Conv2d it is the adjoint of (big image = conv
 * input, small image = conv output) and run "backwards" (its forward = gcc_conv_dgrad, its
 * input-gradient = gcc_conv_fprop, its weight-gradient = gcc_conv_wgrad with x := grad of its
 * output and dy := its input).
 * ------------------------------------------------------------------------------------------- */
/* The tile plan of a convolution call travels WITH the call (round 5: it used to be process-wide option state, which two
 * models -- or two host threads -- with different schedules had to re-apply in turns).  Every field: 0 = the library's default.
 * The functions that size workspaces or statistic rows (gcc_conv_workspace, gcc_conv_stat_tiles, gcc_conv_bn_act_workspace,
 * gcc_conv_wgrad_workspace ...) read the plan of the descriptor they are given: pass them the descriptor the launch gets. */
typedef struct {
    int tile_families;  /* fprop / dgrad: 1: 128-pixel tiles only; 2: + 256x128; 3 (default): + 256x256 */
    int big_min;        /* minimum number of 256-pixel tiles of a launch (default 120: half a chip of one-per-CU workgroups; the rest of
                           the CUs run the other streams' kernels -- measured +2.7 % on the step against 200, profiles/r02_e) */
    int big_nk;         /* minimum K depth in 64-steps for 256-pixel tiles (default 24) */
    int pair;           /* 1: a 256x256-tile launch of < 192 tiles with >= 48 K steps runs two workgroups per tile (one per K half,
                           combined inside the launch through the caller's workspace) and fills the chip by itself: the plan for a
                           launch that has the chip to itself (single-stream schedules: 0.317 against 0.281 of the bf16 peak over the
                           step's igemm launches).  0 (default): one workgroup per tile -- less CU time per launch, the free CUs run
                           the other streams' kernels (the multi-stream production schedule: +1 % on the step) */
    int halo_hc;        /* columns per tile of igemm_halo_kernel: 0 (default): 256 where the layer tiles by 256 and such tiles are
                           enough to be routed (big_min), else 128; 1: also 128 where 256-column tiles would cover less than 3/4 of the
                           chip (PatchGAN L3 forward, L4 data gradient: 128 workgroups) -- the plan for a launch that has the chip to
                           itself, chosen together with `pair`; 128 / 256: forced (tests, A/B) */
    int wgrad_wgs_big;  /* workgroups a split 256x256 weight-gradient launch aims at (default 256) */
    int wgrad_wgs;      /* ... a split 128x128 weight-gradient launch (default 512) */
} gcc_conv_plan_t;

typedef struct {
    int N;            /* batch */
    int H, W;         /* conv INPUT spatial size */
    int Ci, Co;       /* logical channels */
    int KH, KW, stride, pad;
    int ldx, xoff;    /* conv-input tensor pixel stride / channel offset (elements) */
    int ldy, yoff;    /* conv-output tensor pixel stride / channel offset */
    gcc_conv_plan_t plan;   /* zero-filled: the library's defaults */
} gcc_conv_t;

static inline int gcc_conv_out(int in, int k, int stride, int pad) { return (in + 2 * pad - k) / stride + 1; }

#define GCC_INORM_WORKSPACE_BYTES ((size_t)4096 + ((size_t)3 << 19))      /* gcc_inorm_fwd / _bwd, see there */
/* A BatchNorm2d (training statistics) that follows a convolution: nn.BatchNorm2d at models/Pix2Pix.py:34, 44-64, 286-298,
 * 320-341.  Handed to the conv (gcc_epilogue_t.bn, gcc_conv_bn_act) its coefficients are final when the call returns: where
 * the launch that writes the statistic rows can, its last-arriving workgroups fold them (no gcc_bn_finalize launch on the
 * chain; same bits as that launch: one canonical summation order) -- this needs `tail_ws`: GCC_TAIL_WORKSPACE_BYTES bytes,
 * zero-filled ONCE by the caller when it is allocated, used by ONE stream (its calls are ordered) and by nothing else; the
 * library leaves its counter words zero after every launch.  tail_ws NULL (or too small for the layer): a gcc_bn_finalize
 * launch inside the call instead. */
#define GCC_TAIL_WORKSPACE_BYTES ((size_t)4096 + ((size_t)4 << 20) + GCC_INORM_WORKSPACE_BYTES)   /* the last part: gcc_conv_bn_act's grid fold kernel */
typedef struct {
    const float* gamma; const float* beta;          /* [C] */
    float eps, momentum;
    double count;                                   /* N * H * W of the conv output */
    float* running_mean; float* running_var;        /* [C] or NULL */
    float* mean; float* rstd; float* scale; float* shift;   /* [C] outputs (saved for the backward pass) */
    void* tail_ws; size_t tail_ws_bytes;            /* see above; may be NULL / 0 */
    int finalize_in_launch;                         /* 1: the launch that writes the statistic rows may finalize (needs tail_ws); 0: always a
                                                       gcc_bn_finalize launch (tail_ws then only serves gcc_conv_bn_act's grid fold kernel) */
    int pad_;
} gcc_bn_t;

/* fused epilogue of fprop / dgrad: out = act(acc + bias[c]); optional per-tile BatchNorm partial
 * statistics (sum, sum of squares of the bf16-rounded outputs) for gcc_bn_finalize. */
typedef struct {
    const float* bias;    /* [channels] or NULL */
    int act;              /* GCC_ACT_* */
    float slope;          /* LeakyReLU slope */
    float* stats_partial; /* NULL, or [gcc_conv_stat_tiles()][2][channels] fp32 */
    void* workspace;      /* NULL, or gcc_conv_workspace() bytes: lets small-grid launches (U-Net bottleneck,
                             1-channel PatchGAN head) split their K loop over more workgroups */
    size_t workspace_bytes;
    const gcc_bn_t* bn;   /* NULL, or (with stats_partial) the BatchNorm behind this conv: finalized inside the call */
    /* NULL, or a second output written by the same launch: y2[.., y2off + c] = f(out) -- y2_mode 1: relu(out) (the in-place
     * LeakyReLU / ReLU pair the first U-Net skip is read through, models/Pix2Pix.py:33, 50, 77); 2: out * y2_gate[c] (the first
     * DifferentiableOP of the masked PatchGAN, models/Pix2Pix.py:320-322).  Served where gcc_conv_y2_supported() says 1 (the
     * thin image-layer forward route); GCC_ERR_UNSUPPORTED otherwise -- callers keep a gcc_bnact_fwd for that case. */
    void* y2; int ldy2, y2off, y2_mode; const float* y2_gate;
} gcc_epilogue_t;

/* scratch a fprop (dgrad=0) / dgrad (dgrad=1) launch can use: split-K partial tiles of small grids, or the hand-off slabs of a
 * pair-split 256x256-tile launch; 0 when the launch needs none.  Without it the launch simply runs un-split. */
size_t gcc_conv_workspace(const gcc_conv_t* c, int dgrad);

/* number of partial-statistics rows a fprop/dgrad call writes or zero-fills (one per pixel tile of the route the plan picks) */
int gcc_conv_stat_tiles(const gcc_conv_t* c, int dgrad);

/* kernel family a fprop / dgrad call runs on: 0 igemm_kernel, 1 the thin image-layer kernels (<= 8 channels on the
 * image side), 2 the single-output-channel head route, 3 the thin-output kernels for wide filters with <= 3 output channels
 * (conv_thinout.hip, round 5), 4 the ring-walk kernel for 3 x 3 stride-1 layers between <= 64-channel tensors (conv_ring3.hip,
 * round 5); < 0 for an invalid geometry.  Introspection for profilers. */
int gcc_conv_route(const gcc_conv_t* c, int dgrad, const gcc_epilogue_t* ep);
/* 1 when a call with this geometry / epilogue can write ep->y2 (see gcc_epilogue_t) */
int gcc_conv_y2_supported(const gcc_conv_t* c, int dgrad, const gcc_epilogue_t* ep);
/* tile the current plan picks for a geometry on the igemm_kernel route: BP * 1000 + BC (e.g. 256256 = 256 pixels x
 * 256 channels); 0 for an invalid geometry.  Introspection for tests and profilers. */
int gcc_conv_tile(const gcc_conv_t* c, int dgrad);

/* y = conv(x, W) (+bias, act).  Replaces aten::convolution at models/Pix2Pix.py:31-32, 280-300,
 * 320-343, 407-409 (F.conv2d / nn.Conv2d.forward).  x: [N,H,W,ldx]  w: W packing  y: [N,Ho,Wo,ldy] */
int gcc_conv_fprop(const gcc_conv_t* c, const void* x, const void* w, void* y,
                   const gcc_epilogue_t* ep, gcc_stream_t stream);

/* dx = conv_backward_data(dy, Wt) (+bias, act) -- also ConvTranspose2d.forward.  Replaces
 * aten::convolution_backward (grad_input) for the layers above and aten::convolution(transposed)
 * at models/Pix2Pix.py:40-56.  dy: [N,Ho,Wo,ldy]  wt: Wt packing  dx: [N,H,W,ldx] */
int gcc_conv_dgrad(const gcc_conv_t* c, const void* dy, const void* wt, void* dx,
                   const gcc_epilogue_t* ep, gcc_stream_t stream);

/* dW (+)= conv_backward_weight(x, dy) into the fp32 master layout [Co][KH*KW][Ci].  Replaces
 * aten::convolution_backward (grad_weight).  ws: workspace of gcc_conv_wgrad_workspace() bytes
 * (split-K slabs); accumulate != 0 adds into dw (PyTorch .grad accumulation semantics). */
size_t gcc_conv_wgrad_workspace(const gcc_conv_t* c);
int gcc_conv_wgrad(const gcc_conv_t* c, const void* x, const void* dy, float* dw, int accumulate,
                   void* ws, size_t ws_bytes, gcc_stream_t stream);

/* A GROUP of weight gradients as one launch (+ one fold launch): the backward pass of a generator yields a dozen small layers whose
 * weight gradients each fill a fraction of the chip (models/Pix2Pix.py:20-130: 14 regular ones per U-Net pass; aten runs one
 * convolution_backward per layer at :576 loss_G.backward()).  Entry i is what gcc_conv_wgrad(&c, x, dy, dw, accumulate, ..) would
 * compute; the pixel splits are planned for the group as a whole.  Only regular entries (Ci % 8 == 0, dw 16-byte aligned, no head /
 * thin-output geometry): gcc_conv_wgrad_group_workspace returns 0 for a group that holds any other, and the caller keeps such layers
 * on gcc_conv_wgrad.  Protocol (the library allocates nothing and copies nothing to the device):
 *   bytes = gcc_conv_wgrad_group_workspace(items, n)                      -- DEVICE workspace for the slabs (persistent: its address
 *                                                                            is part of the table)
 *   gcc_conv_wgrad_group_prepare(items, n, ws, bytes, table_host)         -- fills gcc_conv_wgrad_group_table_bytes() bytes of HOST
 *                                                                            memory; the caller copies them to device memory ONCE
 *   gcc_conv_wgrad_group_run(table_dev, table_host, stream)               -- every iteration: two launches
 * A table stays valid while the pointers and geometries it was prepared from do.  Fixed summation order: same bits run after run. */
#define GCC_WGRAD_GROUP_MAX 32
typedef struct {
    gcc_conv_t c;
    const void* x;
    const void* dy;
    float* dw;
    int accumulate;
} gcc_wgrad_item_t;
size_t gcc_conv_wgrad_group_table_bytes(void);
size_t gcc_conv_wgrad_group_workspace(const gcc_wgrad_item_t* items, int n);
int gcc_conv_wgrad_group_prepare(const gcc_wgrad_item_t* items, int n, void* ws, size_t ws_bytes, void* table_host);
int gcc_conv_wgrad_group_run(const void* table_dev, const void* table_host, gcc_stream_t stream);

/* weight gradient when channel dimensions are concatenations (see gcc_pack_desc_t): `c` describes the
 * PHYSICAL problem (c->Co = padded rows, c->Ci = padded cols); dw is the logical master-layout gradient
 * [rows][taps][cols]. */
int gcc_conv_wgrad_seg(const gcc_conv_t* c, const void* x, const void* dy, float* dw, int rows, int cols,
                       int row_split, int col_split, int accumulate, void* ws, size_t ws_bytes, gcc_stream_t stream);

/* fp32 master [rows][taps][cols] -> bf16 W [rows][taps][ceil8(cols)] and Wt [cols][taps][ceil8(rows)].
 * Either output may be NULL. */
int gcc_pack_weights(const float* master, int rows, int taps, int cols, void* w, void* wt,
                     gcc_stream_t stream);

/* multi-tensor form: DEVICE arrays built once per optimizer group.  kind 0: W chunk `a` (2048 output
 * elements); kind 1: one 32x32 tile (row block b, column block c) of tap `a` for Wt; kind 2 (unsplit tensors with
 * cols % 4 == 0 and a 16-byte aligned master): W and Wt of one 64x64 tile of tap `a` from a single read of the master. */
/* rows / cols are the master's logical sizes; row_split / col_split (0 = none) say that the dimension is
 * a concatenation whose first part has that many channels: each part is padded to 8 channels in the
 * packings (W = [rowsp][taps][colsp], Wt = [colsp][taps][rowsp], rowsp/colsp = padded physical sizes),
 * matching activation buffers in which the two tensors sit in 8-aligned channel slices. */
typedef struct { const float* master; void* w; void* wt; int rows, taps, cols, colsp, rowsp, row_split, col_split, pad_; } gcc_pack_desc_t;
typedef struct { int tensor, kind, a, b, c, pad_; } gcc_pack_item_t;
int gcc_pack_weights_multi(const gcc_pack_desc_t* descs, const gcc_pack_item_t* items, int nitems,
                           gcc_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Tensor packing between the public NCHW fp32 surface and NHWC bf16.
 * Replaces torch.cat((real_A, fake_B), 1) at models/Pix2Pix.py:467,471,494,499,516 and the
 * host->device staging of set_input (:456-457).
 * ------------------------------------------------------------------------------------------- */
/* dst[n,h,w,dstoff + c] = bf16(src[n,c,h,w]) for c < C ; channels [C, Cfill) zero-filled. */
int gcc_nchw_f32_to_nhwc_bf16(const float* src, void* dst, int N, int C, int H, int W, int ld, int off,
                              int Cfill, gcc_stream_t stream);
int gcc_nhwc_bf16_to_nchw_f32(const void* src, float* dst, int N, int C, int H, int W, int ld, int off,
                              gcc_stream_t stream);
/* channel-slice copy between NHWC bf16 tensors: dst[.., doff+c] = src[.., soff+c], c < C (C%8==0 not
 * required; [C, Cfill) zero-filled in dst). */
int gcc_nhwc_copy(const void* src, int lds, int soff, void* dst, int ldd, int doff, int C, int Cfill,
                  size_t pixels, gcc_stream_t stream);
/* channel concatenation of two thin tensors into one 8-wide group: dst[.., doff + c] = a[.., aoff + c] (c < Ca), then
 * b[.., boff + c] (c < Cb), zeros up to 8 (Ca + Cb <= 8).  Replaces torch.cat((real_A, fake_B), 1) at
 * models/Pix2Pix.py:466-470, 483 (the discriminator's conditional input). */
int gcc_nhwc_pack_pair(const void* a, int lda, int aoff, const void* b, int ldb, int boff, void* dst, int ldd, int doff,
                       int Ca, int Cb, size_t pixels, gcc_stream_t stream);
/* dst[.., doff+c] += src[.., soff+c]  (gradient fan-in of fake_B: models/Pix2Pix.py:516-550) */
int gcc_nhwc_add(const void* src, int lds, int soff, void* dst, int ldd, int doff, int C, size_t pixels,
                 gcc_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * BatchNorm2d (training / eval) fused with activation, DifferentiableOP gate and dropout.
 * Replaces aten::native_batch_norm(+_backward), leaky_relu_, relu_, bernoulli_/mul (Dropout) and
 * the mask multiply of models/DifferentiableOp.py:44-49 at models/Pix2Pix.py:33-36, 57-64, 286-298,
 * 320-341.
 * ------------------------------------------------------------------------------------------- */
/* Reduce the per-tile partial sums over `count` pixels into batch statistics; writes
 * mean/rstd (saved for backward), scale = gamma*rstd, shift = beta - mean*scale and updates
 * running_mean/var (momentum, unbiased variance) when they are non-NULL. */
int gcc_bn_finalize(const float* stats_partial, int tiles, int C, double count, const float* gamma,
                    const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                    float* mean, float* rstd, float* scale, float* shift, gcc_stream_t stream);
/* InstanceNorm2d(affine=False, no running statistics; models/Pix2Pix.py:201, models/CycleGAN.py:145): per image
 * and channel mean / rstd from [groups][tiles_per_group][2][C] partial sums (conv epilogue or gcc_channel_stats) */
int gcc_in_finalize(const float* stats_partial, int tiles_per_group, int groups, int C, double count, float eps,
                    float* mean, float* rstd, float* scale, float* shift, gcc_stream_t stream);
/* The same InstanceNorm (+ activation, + residual added after it) in one launch: statistics, mean / rstd / scale / shift [N][C]
 * written for the backward, y = act((x - mean) rstd) + residual (models/CycleGAN.py:77-138 at batch 1).  With a workspace the
 * plane of an image is split over up to 256 / N workgroups (pixel ranges x 64-channel groups) that meet at an in-launch barrier; without one (NULL) a workgroup
 * owns a whole (image, 16-channel slab).  Workspace contract: GCC_INORM_WORKSPACE_BYTES bytes, zero-filled once by the
 * caller when it is allocated, used by ONE stream (calls on it are ordered) and by nothing else. */
int gcc_inorm_fwd(const void* x, int ldx, void* y, int ldy, const void* residual, int ld_residual, int C, int HW, int N,
                  int act, float slope, float eps, float* mean, float* rstd, float* scale, float* shift,
                  void* workspace, size_t workspace_bytes, gcc_stream_t stream);
/* its backward: dx = rstd (dz - mean(dz) - xhat mean(dz xhat)) with dz = g act'(y) (y NULL: the activation output is
 * recomputed from x); dx may alias g. */
int gcc_inorm_bwd(const void* x, int ldx, const void* y, int ldy, const void* g, int ldg, void* dx, int lddx, int C, int HW,
                  int N, int act, float slope, const float* mean, const float* rstd, void* workspace, size_t workspace_bytes,
                  gcc_stream_t stream);
/* BatchNorm backward with training statistics and nothing else fused (no gate, dropout or second gradient) in ONE launch
 * -- the SRResNet / SAGAN-generator blocks (models/SRGAN.py:19-66, models/SAGAN.py:77-140): dx = gamma rstd (dz - mean(dz) -
 * xhat mean(dz xhat)), dz = g act'(y) (y NULL: no activation), dgamma += sum dz xhat, dbeta += sum dz (either may be NULL).
 * workspace: as for gcc_inorm_fwd (the same one may be shared on a stream).  GCC_ERR_UNSUPPORTED when the geometry does not
 * fit the grid form: call gcc_bnact_bwd instead. */
int gcc_bn_bwd_one_launch(const void* x, int ldx, const void* y, int ldy, const void* g, int ldg, void* dx, int lddx, int C,
                          size_t pixels, int act, float slope, const float* mean, const float* rstd, const float* gamma,
                          float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, gcc_stream_t stream);
/* ... with the pieces the U-Net's layers need (models/Pix2Pix.py:33-64): y NULL with an activation (its input is recomputed through
 * the forward's affine form `scale` / `shift` and the regenerated dropout mask), a second incoming gradient g2 through act2 (the
 * skip path: g act'(y) + g2 act2'(y)), Dropout(drop_p) with the (seed, pixel * C + channel) counter of gcc_bnact_fwd.  Any
 * tensor size; GCC_ERR_UNSUPPORTED as above. */
int gcc_bn_bwd_one_launch_ex(const void* x, int ldx, const void* y, int ldy, const void* g, int ldg, const void* g2, int ldg2,
                             void* dx, int lddx, int C, size_t pixels, int act, int act2, float slope, float drop_p,
                             unsigned long long seed, const float* mean, const float* rstd, const float* scale, const float* shift,
                             const float* gamma, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                             gcc_stream_t stream);
int gcc_channel_stats_tiles(size_t pixels_per_group, int C);
int gcc_channel_stats(const void* x, int ld, int off, int C, size_t pixels_per_group, int groups, float* stats,
                      gcc_stream_t stream);
/* eval mode: scale/shift from running statistics */
int gcc_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, int C, float* scale, float* shift,
                       gcc_stream_t stream);

typedef struct {
    const float* scale;   /* [C] or NULL (identity) */
    const float* shift;   /* [C] or NULL */
    const float* gate;    /* [C] gate mask m (0, .5, 1) or NULL */
    int gate_after_act;   /* 0: act(gate(bn(x)))  (models/Pix2Pix.py:327-332) ; 1: gate(act(x)) (:320-322) */
    int act;              /* activation written to y */
    float slope;
    int act2;             /* activation written to y2 (if y2 != NULL): y2 = act2(gate(bn(x))) */
    float drop_p;         /* dropout probability (0 = off); applied after bn, before act (U-Net up path) */
    uint64_t seed;        /* counter-based RNG: keep = hash(seed, element index) >= p */
    int groups;           /* <= 1: BatchNorm.  G > 1: InstanceNorm over G images -- `pixels` is per image, scale/shift
                             are [G][C] (gcc_in_finalize), x/y/y2/residual hold the G images back to back */
    int ld_residual;
    const void* residual; /* optional NHWC bf16 tensor added to y after the activation (ResNet block) */
} gcc_bnact_t;

/* y[.., yoff+c] = act(...) ; optional second output y2 (e.g. the ReLU'd copy that lands in the
 * concat buffer).  x/y/y2 are NHWC bf16 with their own ld/off. */
int gcc_bnact_fwd(const gcc_bnact_t* p, const void* x, int ldx, int xoff, void* y, int ldy, int yoff,
                  void* y2, int ldy2, int y2off, int C, size_t pixels, gcc_stream_t stream);

/* Conv2d / ConvTranspose2d + BatchNorm2d(training statistics) + activation (+ dropout, + a second activated copy) as ONE call:
 * the U-Net layer `conv -> BatchNorm2d -> LeakyReLU/ReLU[/Dropout]` of models/Pix2Pix.py:31-36, 40-64 (conv = fprop for
 * dgrad == 0, backward-data / ConvTranspose forward for dgrad == 1, as gcc_conv_fprop / gcc_conv_dgrad; bias-free).
 *   y_raw (the conv's own output, kept for the backward pass), then batch statistics of its bf16-rounded values, mean / rstd /
 *   scale / shift and the running-statistics update as gcc_bn_finalize, then y = act(drop(bn(y_raw))) and, if y2 != NULL,
 *   y2 = act2(drop(bn(y_raw))) as gcc_bnact_fwd (`act`: its scale / shift / gate / residual / groups fields are ignored).
 * Layers whose grid is split over K (the U-Net's <= 16x16 layers: a handful of output rows, thousands of K) run two launches --
 * the fp32 partial tiles, then one kernel that folds them, takes the statistics (f64), finalises and normalises: a workgroup
 * per 8 output channels owns every row of them -- instead of five (partials, fold, channel statistics, finalize, normalise);
 * every other layer runs the ordinary three kernels inside this one call.  ws: gcc_conv_bn_act_workspace() bytes. */
size_t gcc_conv_bn_act_workspace(const gcc_conv_t* c, int dgrad);
int gcc_conv_bn_act(const gcc_conv_t* c, int dgrad, const void* x, const void* w, void* y_raw, const gcc_bn_t* bn,
                    const gcc_bnact_t* act, void* y, int ldy, int yoff, void* y2, int ldy2, int y2off, void* ws,
                    size_t ws_bytes, gcc_stream_t stream);

/* Backward of y = act(gate(bn(x))) [dropout] given up to two upstream gradients:
 *   g  = g1 * act'(y)  +  g2 * act2'(y)     (g2 from the skip/concat path, may be NULL)
 *   dz = g * m[c]  (gate)       dalpha[c] = sum g * z   (z = bn(x), STE, no mask factor)
 *   dgamma[c] = sum dz*xhat, dbeta[c] = sum dz, dx = scale*(dz - mean(dz) - xhat*mean(dz*xhat))
 * Two launches: reduce (writes dz into `dz` and per-block partials into ws) then apply (dx over dz).
 * bn == 0 : no normalisation (plain act backward: dx = g*m, no statistics). */
typedef struct {
    int bn;                  /* 1: through BatchNorm (training statistics) */
    int bn_eval;             /* 1: BN used running stats (dx = dz*scale) */
    const float* mean;       /* saved mean [C] */
    const float* rstd;       /* saved rstd [C] */
    const float* gamma;      /* [C] */
    const float* beta;       /* [C] */
    const float* gate;       /* mask [C] or NULL */
    int gate_after_act;
    int act; float slope;    /* activation of y  (y is what act' is evaluated on, in-place semantics) */
    int act2;                /* activation of the second consumer (g2) */
    float drop_p; uint64_t seed;
    float* dgamma; float* dbeta; float* dalpha;  /* [C] fp32, accumulated into (+=) ; any may be NULL */
    int groups;              /* G > 1: InstanceNorm backward (mean/rstd [G][C], no parameter gradients); workspace is
                                G * gcc_bnact_bwd_workspace() */
    int flags;               /* bit 0: the workspace was zero-filled when it was allocated, is used by ONE stream and by nothing but
                                gcc_bnact_bwd calls -- the finalize step then runs inside the reduce launch (its last-arriving
                                workgroups fold the partial rows; the library leaves the head of the workspace zero): two launches
                                instead of three */
} gcc_bnact_bwd_t;

size_t gcc_bnact_bwd_workspace(int C, size_t pixels);
int gcc_bnact_bwd(const gcc_bnact_bwd_t* p, const void* x, int ldx, int xoff, const void* y, int ldy, int yoff,
                  const void* g1, int ldg1, int g1off, const void* g2, int ldg2, int g2off,
                  void* dx, int lddx, int dxoff, int C, size_t pixels, void* ws, size_t ws_bytes,
                  gcc_stream_t stream);

/* same, for a layer whose producer already applied `in_act` to x (conv epilogue): dx *= in_act'(x) */
int gcc_bnact_bwd_ex(const gcc_bnact_bwd_t* p, int in_act, float in_slope, const void* x, int ldx, int xoff,
                     const void* y, int ldy, int yoff, const void* g1, int ldg1, int g1off, const void* g2,
                     int ldg2, int g2off, void* dx, int lddx, int dxoff, int C, size_t pixels, void* ws,
                     size_t ws_bytes, gcc_stream_t stream);

/* ReflectionPad2d (explicit copy; backward = adjoint gather) and the depthwise 3x3 convolution with
 * ReflectionPad2d(1) of MobileResnetBlock / SeparableConv2d (models/Pix2Pix.py:132-197).
 * gcc_dwconv3x3_reflect mode 0: out = conv(x) + bias ; mode 1: out = d(x) from dy.  w: fp32 [C][9] master. */
int gcc_reflect_pad(const void* src, int lds, void* dst, int ldd, int N, int H, int W, int C, int pad, int backward,
                    gcc_stream_t stream);
int gcc_dwconv3x3_reflect(int mode, const void* x, int ldx, const void* dy, int lddy, void* out, int ldo, const float* w,
                          const float* bias, int N, int H, int W, int C, gcc_stream_t stream);
size_t gcc_dwconv3x3_wgrad_workspace(int N, int H, int W, int C);
int gcc_dwconv3x3_reflect_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw, float* dbias, int N, int H,
                                int W, int C, void* ws, size_t ws_bytes, gcc_stream_t stream);

/* per-channel sum over pixels of an NHWC bf16 tensor (bias gradients): out[c] (+)= sum x[..,c] */
int gcc_channel_sum(const void* x, int ld, int off, int C, size_t pixels, float* out, int accumulate,
                    void* ws, size_t ws_bytes, gcc_stream_t stream);
size_t gcc_channel_sum_workspace(int C, size_t pixels);
/* several channel sums of <= 16384 pixels each (the bias gradients of the layers of one grouped weight gradient) as ONE launch;
 * entry i is gcc_channel_sum(x, ld, off, C, pixels, out, accumulate, ..) bit for bit.  n <= GCC_CHANSUM_GROUP_MAX;
 * GCC_ERR_UNSUPPORTED when an entry has more pixels (the caller keeps it on gcc_channel_sum). */
#define GCC_CHANSUM_GROUP_MAX 24
typedef struct {
    const void* x;
    int ld, off, C;
    size_t pixels;
    float* out;
    int accumulate;
} gcc_chansum_item_t;
int gcc_channel_sum_group(const gcc_chansum_item_t* items, int n, gcc_stream_t stream);

/* gate mask m = (sign(alpha - tau) + 1) / 2.  models/DifferentiableOp.py:25-26,58-59 */
int gcc_gate_mask(const float* alpha, float tau, float* mask, int C, gcc_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Losses (forward value + gradient in one pass; scalars are fp32 device words).
 * ------------------------------------------------------------------------------------------- */
/* GANLoss.__call__, models/GANLoss.py:38-59.  mode 0 hinge, 1 lsgan, 2 vanilla(BCE logits), 3 wgangp.
 * pred: NHWC bf16 [pixels][ld] channel `off` (PatchGAN map, 1 channel).  loss (+)= weight * L ;
 * dpred (may be NULL) = weight * dL/dpred written as bf16 with the same layout. */
int gcc_gan_loss(int mode, int target_is_real, int for_discriminator, const void* pred, int ld, int off,
                 size_t pixels, float weight, float* loss, int accumulate, void* dpred,
                 void* ws, size_t ws_bytes, gcc_stream_t stream);
/* same loss; `loss` receives the unweighted value, dpred (+)= grad_weight * (*grad_weight_dev) * dL/dpred.
 * grad_weight_dev (device scalar, may be NULL) carries data-dependent signs of the arch loss without
 * a host round trip (models/Pix2Pix.py:479-487). */
int gcc_gan_loss_ex(int mode, int target_is_real, int for_discriminator, const void* pred, int ld, int off,
                    size_t pixels, float* loss, void* dpred, float grad_weight, const float* grad_weight_dev,
                    int dpred_accumulate, gcc_stream_t stream);
/* arch-step scalars: loss = | |Lfr-Lf| - dT | + w (Lr+Lf) and its partial derivatives c_fr, c_f
 * (w = 1/2: models/Pix2Pix.py:505-509, models/CycleGAN.py:410-414; w = 1: models/SAGAN.py:388-389) */
int gcc_arch_coeffs(const float* Lfr, const float* Lf, const float* Lr, const float* dT, float real_fake_weight,
                    float* loss, float* c_fr, float* c_f, gcc_stream_t stream);
/* mean |a-b| * weight (nn.L1Loss, models/Pix2Pix.py:520) over C channels; da = weight*sign(a-b)/count */
int gcc_l1_loss(const void* a, int lda, int aoff, const void* b, int ldb, int boff, int C, size_t pixels,
                float weight, float* loss, int accumulate, void* da, int ldda, int daoff,
                void* ws, size_t ws_bytes, gcc_stream_t stream);
/* nn.MSELoss (models/SRGAN.py:447, 457): weight * mean((a-b)^2); da = weight * 2 (a-b) / count */
int gcc_mse_loss(const void* a, int lda, int aoff, const void* b, int ldb, int boff, int C, size_t pixels,
                 float weight, float* loss, int accumulate, void* da, int ldda, int daoff,
                 void* ws, size_t ws_bytes, gcc_stream_t stream);
size_t gcc_loss_workspace(size_t pixels, int C);

/* Feature distillation, models/Pix2Pix.py:537-548 + :733-740, for one feature pair:
 *   gram(f) = F F^T /(c h w) per image (F = [C][HW]);  Lg = sqrt(mse(gram(f), gram(t))) ;
 *   Lc = sqrt(mse(f, t)).   f,t: NHWC bf16 [N][HW][ld].
 * squared != 0: the CycleGAN form (models/CycleGAN.py:516-517), plain MSE terms without the square root.
 * gcc_distill_fwd writes the two scalars (unweighted) into out[0..1] and keeps what backward needs
 * in ws; gcc_distill_bwd writes df = wg*dLg/df + wc*dLc/df (bf16, same layout as f). */
size_t gcc_distill_workspace(int N, int C, int HW);
int gcc_distill_fwd(const void* f, int ldf, int foff, const void* t, int ldt, int toff, int N, int C, int HW,
                    int squared, float* out2, void* ws, size_t ws_bytes, gcc_stream_t stream);
int gcc_distill_bwd(const void* f, int ldf, int foff, const void* t, int ldt, int toff, int N, int C, int HW,
                    int squared, float wg, float wc, void* df, int lddf, int dfoff, void* ws, size_t ws_bytes,
                    gcc_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * SAGAN (models/SAGAN.py).
 * Spectral normalisation, SpectralNorm._update_u_v (:25-38), on the fp32 master W_bar [R][C][k][k] (channels_last for
 * k > 1; the reference's w.view(R, -1) column order c*T + t is kept for v, T = k*k): one power iteration that
 * overwrites u [R] and v [C*T], t_out = W_bar v (kept by the caller for the gradient of u), sigma = u . t_out, and
 * W_eff = W_bar / sigma (fp32, same layout; feed it to gcc_pack_weights).  Runs on every forward, eval included.
 * gcc_spectral_grad folds G = dL/dW_eff into the master's gradient:
 *   dW_bar += G/sigma + s u v^T,  du += s t_fwd,  dv += s W_bar^T u,   s = -<G, W_bar>/sigma^2
 * with the LIVE u, v and the forward call's own sigma / t (what the reference's autograd evaluates, see
 * csrc/spectral.hip).  du / dv may be NULL (generator: u, v are never trained; discriminator: set_requires_grad
 * switches them on, :513). */
size_t gcc_spectral_workspace(int R, int C, int T);
int gcc_spectral_power_iteration(const float* w_bar, float* u, float* v, int R, int C, int T, float* t_out,
                                 float* sigma_out, float* w_eff, void* ws, size_t ws_bytes, gcc_stream_t stream);
/* the same power iteration with W_eff written straight into the two bf16 packings of gcc_pack_weights (w: [R][T][C padded to 8],
 * wt: [C][T][R padded to 8]) instead of an fp32 tensor: 4 launches instead of 7 for what a spectrally normalised convolution's
 * forward does before its convolution (SpectralNorm.forward, models/SAGAN.py:56-70); same bits as the separate calls */
int gcc_spectral_power_iteration_pack(const float* w_bar, float* u, float* v, int R, int C, int T, float* t_out,
                                      float* sigma_out, void* w, void* wt, void* ws, size_t ws_bytes, gcc_stream_t stream);
/* the power iterations + packings of several layers (every spectrally normalised convolution of one network's forward pass:
 * the iterations depend on the weights alone, not on the activations) as FOUR launches instead of four per layer; entry i is
 * gcc_spectral_power_iteration_pack(w_bar, u, v, R, C, T, t_out, sigma_out, w, wt, ..) bit for bit.  n <= GCC_SPECTRAL_GROUP_MAX;
 * ws: gcc_spectral_group_workspace(items, n) bytes (every layer its own scratch), 16-byte aligned. */
#define GCC_SPECTRAL_GROUP_MAX 8
typedef struct {
    const float* w_bar;
    float* u;
    float* v;
    int R, C, T;
    float* t_out;
    float* sigma_out;
    void* w;
    void* wt;
} gcc_sn_item_t;
size_t gcc_spectral_group_workspace(const gcc_sn_item_t* items, int n);
int gcc_spectral_power_iteration_pack_group(const gcc_sn_item_t* items, int n, void* ws, size_t ws_bytes, gcc_stream_t stream);
int gcc_spectral_grad(const float* g_eff, const float* w_bar, const float* u, const float* v, const float* t_fwd,
                      const float* sigma_fwd, int R, int C, int T, float* dw_bar, float* du, float* dv, void* ws,
                      size_t ws_bytes, gcc_stream_t stream);
/* Self attention, Self_Attn.forward (:72-104), per image over N = H*W <= 1024 positions: q, k (C8 channels) and v
 * (C <= 512 channels) are channel slices (qoff / koff / voff) of one NHWC bf16 buffer; y = gamma * softmax(q^T k) v + x.
 * The N x N score / attention matrices are never stored: the kernels recompute score tiles on the matrix cores (their
 * inner dimension is C/8).  Saved for backward: o (pre-gamma output, bf16 [B][N][ldo]) and stats (fp32 [B][N][2]: row
 * maximum and row sum of exp(s - max)).  A: optional fp32 [B][N][N] attention map -- the reference's forward returns it
 * (models/SAGAN.py:103), SAGANModel drops it; NULL skips the write.
 * gcc_attention_bwd: dq / dk / dv into the same slices of dqkv, dgamma (+=); rowdot is an fp32 [B][N] scratch.  The
 * residual branch (dx += dy) belongs to the caller. */
int gcc_attention_fwd(const void* qkv, int ldq, int qoff, int koff, int voff, const void* x, int ldx,
                      const float* gamma, int B, int N, int C, int C8, void* y, int ldy, void* o, int ldo,
                      float* stats, float* A, gcc_stream_t stream);
int gcc_attention_bwd(const void* qkv, int ldq, int qoff, int koff, int voff, const void* o, int ldo,
                      const float* stats, const float* gamma, const void* dy, int lddy, int B, int N, int C, int C8,
                      void* dqkv, int lddq, float* rowdot, float* dgamma, gcc_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * SRGAN (models/SRGAN.py, models/GANLoss.py:95-145).
 * gcc_prelu: nn.PReLU() with its single learnable slope (device scalar), optionally fused with the nn.PixelShuffle(2)
 * in front of it (SubPixelConvolutionalBlock, :68-99): shuffle == 2 reads x [N][H][W][4C] and writes y [N][2H][2W][C],
 * y[n][2h+i][2w+j][c] = prelu(x[n][h][w][4c+2i+j]).  backward != 0: dx (layout of x) from dy (layout of y), and
 * dslope (+=, may be NULL: the distillation optimizer of the reference leaves the PReLU slopes out, :349-352); with a
 * workspace (GCC_PRELU_WORKSPACE_BYTES, zero-filled once by the caller, one per stream; its first 33 words -- the arrival counters -- are zero again after
 * every call) the sum behind dslope is formed in a fixed order -- bit-reproducible; without one (NULL) by atomic adds.
 * gcc_maxpool2x2: nn.MaxPool2d(2, 2) of the VGG stack; backward (1) routes to the first maximum in scan order; backward == 2: x is
 * the output of a ReLU and the ReLU's backward is applied on the way (a gradient whose maximum is not positive is dropped).
 * gcc_pool_linear_*: AdaptiveAvgPool2d((1,1)) + Linear(C, 1) of the discriminators (:245-262): pooled [N][C] fp32 is
 * kept for the backward pass, logit is bf16 [N][ldl] (one pixel per image, as gcc_gan_loss reads it). */
int gcc_prelu(int backward, const void* x, int ldx, const float* slope, int C, int N, int H, int W, int shuffle,
              void* y, int ldy, const void* dy, int lddy, void* dx, int lddx, float* dslope, void* workspace,
              size_t workspace_bytes, gcc_stream_t stream);
#define GCC_PRELU_WORKSPACE_BYTES ((size_t)256 + 4 * 4096)
int gcc_maxpool2x2(int backward, const void* x, int ldx, void* y, int ldy, const void* dy, int lddy, void* dx, int lddx,
                   int N, int Ho, int Wo, int C, gcc_stream_t stream);
int gcc_pool_linear_fwd(const void* x, int ldx, int N, int HW, int C, const float* w, const float* b, float* pooled,
                        void* logit, int ldl, gcc_stream_t stream);
int gcc_pool_linear_bwd(const void* dlogit, int ldl, const float* w, const float* pooled, int N, int HW, int C,
                        void* dx, int lddx, float* dw, float* db, gcc_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Optimizer: multi-tensor Adam (torch.optim.Adam, no weight decay; models/Pix2Pix.py:382,415,
 * 430-431) with the L1-sparsity sub-gradient of L1_sparsity() (:554-563) fused in.
 * `tensors` / `chunks` are DEVICE arrays built once per optimizer; step is the 1-based step count.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    float* p; const float* g; float* m; float* v;
    int64_t numel;
    float l1;        /* grad += l1 * sign(p) before the update (lambda_weight / lambda_scale) */
    float grad_scale;/* grad *= grad_scale first (1/world_size after a sum all-reduce) */
} gcc_adam_tensor_t;
/* work list: one 256-thread workgroup per chunk of `chunk_elems` consecutive elements of one tensor */
typedef struct { int tensor; int pad_; int64_t offset; } gcc_adam_chunk_t;
int gcc_adam_step(const gcc_adam_tensor_t* tensors, const gcc_adam_chunk_t* chunks, int nchunks, int chunk_elems,
                  float lr, float beta1, float beta2, float eps, int step, gcc_stream_t stream);

/* misc fp32 helpers */
/* device-side scalar algebra of the arch step (models/Pix2Pix.py:484-486, 505-511):
 * op 0: out = |a-b| ; op 1: out = k0*|a-b| + k1*c ; op 2: out = a + k0*b */
int gcc_scalar_op(int op, const float* a, const float* b, const float* c, float k0, float k1, float* out,
                  gcc_stream_t stream);
int gcc_fill_f32(float* p, float v, size_t n, gcc_stream_t stream);
int gcc_clamp_f32(float* p, float lo, float hi, size_t n, gcc_stream_t stream);
/* dst[i] += src[i]: folds a gradient accumulated apart (the architecture step's second discriminator pass, run beside the first
 * on another stream: models/Pix2Pix.py:496-511 accumulates both into DifferentiableOP.alpha.grad) in the reference's order */
int gcc_add_f32(float* dst, const float* src, size_t n, gcc_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Evaluation arithmetic.  The evaluator networks (Inception, DRN) stay external: these entry
 * points take their outputs as device tensors.
 * gcc_argmax_channels: scores NCHW fp32 [N][C][HW] -> class index int32 [N][HW], numpy.argmax semantics
 *   (metric/mIoU_score.py:212 `final.argmax(axis=1)`).
 * gcc_confusion_hist: fast_hist (metric/mIoU_score.py:163-167): hist[n * label + pred] += 1 where 0 <= label < n; int64
 *   [n][n], accumulated (+=) so that a whole evaluation set sums into one matrix.  Exact.
 * gcc_psnr_y_sse: sum of squared luminance differences of two NCHW fp32 images in [-1, 1] with the 4-pixel border
 *   cropped (models/SRGAN.py:653-657: convert_image(..., 'y-channel'), data/sr_dataset.py:36-37, 58-62); PSNR =
 *   10 log10(255^2 N (H-8)(W-8) / sse).  f64 accumulation in a fixed order.
 * gcc_ssim_y_sum: sum over the N images and over all window centres of the SSIM map of the same luminance images
 *   (models/SRGAN.py:659-661, skimage.metrics.structural_similarity(real_y, fake_y, data_range=255.) with its defaults: 7 x 7
 *   uniform window, K1 = .01, K2 = .03, sample covariance, border of 3 cropped); SSIM = sum / (N (H-14)(W-14)).  f64.
 *   Workspace as gcc_psnr_workspace().  (skimage is an un-pinned dependency that the build image lacks: parity unpinned.)
 * gcc_activation_stats: mu = mean(act, 0), sigma = np.cov(act, rowvar=False) in f64 (metric/fid_score.py:327-328) of
 *   activations [n][d] (fp32 or f64, row major).
 * gcc_frechet_distance: metric/fid_score.py:219-284; out[0] = |mu1-mu2|^2 + tr(s1) + tr(s2) - 2 tr(sqrtm(s1 s2)) with
 *   the matrix square root by `iterations` coupled Newton-Schulz steps on the GPU (f64) of s1 s2 + delta I, delta =
 *   shift_rel |s1 s2|_F, solved at delta and 4 delta and extrapolated to delta = 0 (shift_rel = 0: one unshifted solve,
 *   full-rank inputs only); out[1] = relative change of the trace over the last step (the caller's convergence check).
 * --------------------------------------------------------------------------------------------- */
int gcc_argmax_channels(const float* scores, int N, int C, size_t HW, int* pred, gcc_stream_t stream);
int gcc_confusion_hist(const int* pred, const int* label, size_t count, int n, long long* hist, gcc_stream_t stream);
size_t gcc_psnr_workspace(void);
int gcc_psnr_y_sse(const float* fake, const float* real, int N, int H, int W, double* sse, int accumulate, void* ws,
                   size_t ws_bytes, gcc_stream_t stream);
int gcc_ssim_y_sum(const float* fake, const float* real, int N, int H, int W, double* ssim_sum, int accumulate, void* ws,
                   size_t ws_bytes, gcc_stream_t stream);
size_t gcc_activation_stats_workspace(int n, int d);
int gcc_activation_stats(const void* act, int is_f64, int n, int d, double* mu, double* sigma, void* ws, size_t ws_bytes,
                         gcc_stream_t stream);
size_t gcc_frechet_workspace(int d);
int gcc_frechet_distance(const double* mu1, const double* sigma1, const double* mu2, const double* sigma2, int d,
                         int iterations, double shift_rel, double* out, void* ws, size_t ws_bytes, gcc_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Input pipeline (data/aligned_dataset.py:27-56, data/base_dataset.py:63-112) on decoded
 * 8-bit RGB images in device memory ([H][W][3] rows of `pitch` bytes; a half of the paired A|B image is addressed by
 * offsetting `src`).
 * gcc_resample_u8: PIL's Image.resize(..., BICUBIC) -- any of its separable filters, the coefficient tables decide --
 *   bit for bit: horizontal pass, uint8 intermediate, vertical pass, 22-bit fixed-point coefficients (hcoef [out_w][hk],
 *   hbounds [out_w][2] = first source column and tap count; likewise v*), tmp = in_h * out_w * 3 bytes when both passes
 *   run.  dst is tightly packed [out_h][out_w][3].
 * gcc_crop_flip_normalize: __crop + __flip + ToTensor + Normalize((.5,.5,.5),(.5,.5,.5)): NCHW fp32 [3][crop_h][crop_w]
 *   and / or the NHWC bf16 model input (pixel stride ld, channels 0..2 written).
 * --------------------------------------------------------------------------------------------- */
int gcc_resample_u8(const void* src, int in_h, int in_w, size_t pitch, void* dst, int out_h, int out_w, const int* hbounds,
                    const int* hcoef, int hk, const int* vbounds, const int* vcoef, int vk, void* tmp, gcc_stream_t stream);
int gcc_crop_flip_normalize(const void* src, int H, int W, size_t pitch, int x0, int y0, int crop_h, int crop_w, int flip,
                            float* nchw, void* nhwc_bf16, int ld, gcc_stream_t stream);
/* the same with the range conversion chosen: form 0 Normalize(mean3, std3) (host arrays; ImageNet statistics for the SRGAN
 * low-resolution input, data/sr_dataset.py:52-56), form 1 convert_image '[-1, 1]' = 2 (v / 255) - 1 (:49-50), form 2 v / 255 */
int gcc_crop_convert(const void* src, int H, int W, size_t pitch, int x0, int y0, int crop_h, int crop_w, int flip, int form,
                     const float* mean3, const float* std3, float* nchw, void* nhwc_bf16, int ld, gcc_stream_t stream);

/* ---- gradient exchange ---------------------------------------------------------------------------------
 * The reference trains on one device (models/Pix2Pix.py:356); the data-parallel path sums the five optimizers' flat fp32
 * gradient buffers over ranks before each `optimizer.step()` (train.py has no counterpart: this is the exchange a
 * multi-GPU launch of it needs).  An explicit communicator over RCCL, one per process (one process per GPU), created from an
 * id that rank 0 makes and the host distributes by its own means.  gcc_amd's Python host uses torch.distributed (the same
 * RCCL) instead; these entry points serve a host without one.  RCCL is loaded on first use (GCC_ERR_UNSUPPORTED if absent). */
#define GCC_COMM_ID_BYTES 128
typedef struct gcc_comm gcc_comm_t;
int gcc_comm_unique_id(void* id /* [GCC_COMM_ID_BYTES], host */);
int gcc_comm_init(gcc_comm_t** comm, int rank, int world, const void* id);   /* collective; binds to the current device */
/* in-place sum over ranks of count fp32 values, ordered on `stream` like a kernel; the caller applies 1 / world in its
 * optimizer step (gcc_adam_tensor_t.grad_scale) and picks the bucket size (one call per bucket) */
int gcc_comm_allreduce_sum_f32(gcc_comm_t* comm, float* buf, size_t count, gcc_stream_t stream);
/* the same over bf16 values: a gradient bucket cast by gcc_cast_f32_bf16, summed, cast back by gcc_cast_bf16_f32 moves half the
 * bytes over xGMI; every rank receives the same sums, so replicas stay identical.  Both all-reduce entry
 * points are part of a launch recording (gcc_replay_begin) made on the calling thread; a recording that holds one is replayed
 * from one host thread, in the recorded order (collectives of a communicator must be issued in one order on every rank). */
int gcc_comm_allreduce_sum_bf16(gcc_comm_t* comm, void* buf, size_t count, gcc_stream_t stream);
int gcc_cast_f32_bf16(const float* src, void* dst, size_t n, gcc_stream_t stream);      /* src 16-byte, dst 8-byte aligned */
int gcc_cast_bf16_f32(const void* src, float* dst, size_t n, gcc_stream_t stream);
int gcc_comm_rank(const gcc_comm_t* comm);
int gcc_comm_world(const gcc_comm_t* comm);
int gcc_comm_count(const gcc_comm_t* comm);      /* ncclCommCount of the communicator (>= 1), or a negative error code */
int gcc_comm_destroy(gcc_comm_t* comm);
/* RCCL's own message for the calling thread's last failing gcc_comm_* call ("" if none failed): a GCC_ERR_LAUNCH from this
 * group otherwise hides which ncclResult it was */
const char* gcc_comm_last_error(void);

/* CycleGAN's image history (utils/image_pool.py:5-54) on the device: gcc_write_i32 stores n <= 16 ints taken BY VALUE (the
 * host's random draws: two per image, mode 0 pass through / 1 store and pass through / 2 swap with slot, and the slot);
 * gcc_image_pool_query moves the pixels: images / out [N][HW][8] bf16, pool [slots][HW][8], sel device [N][2]. */
int gcc_write_i32(int* dst, const int* values, int n, gcc_stream_t stream);
int gcc_image_pool_query(const void* images, void* out, void* pool, const int* sel, int N, size_t HW, int slots,
                         gcc_stream_t stream);

/* ---- launch replay (gcc_amd/csrc/replay.hip) -------------------------------------------------------------------------
 * The reference leaves the host side of an iteration to PyTorch's eager dispatcher (train.py:128-140 calls
 * model.optimize_parameters() per batch); here an iteration is a few thousand launches of this library, and for the small
 * models the HOST's launch rate is the bound.  gcc_replay_begin starts recording on the calling thread: every launch of
 * the library made by that thread (and its memsets / copies, and gcc_event_record / gcc_stream_wait_event) still executes,
 * and is written down with its argument values.  gcc_replay_end closes the recording; gcc_replay_run issues it again.
 * Contract (as for a captured graph): all pointer arguments stay valid and keep their meaning -- inputs are fed through
 * persistent buffers, the iteration's temporaries live in a pool that nothing else allocates from -- and scalars that change
 * from iteration to iteration are patched: gcc_replay_tag_next(tag) marks the next launch the thread records,
 * gcc_replay_patch overwrites one argument of every launch with that tag (gcc_adam_step's launch takes, in this order,
 * tensors, chunks, chunk_elems, lr, beta1, beta2, eps, bias_correction1, sqrt(bias_correction2): gcc_adam_factors gives
 * the last two for a step).  threads > 1 at gcc_replay_end: gcc_replay_run issues each HIP stream's launches from its own
 * host thread (an event wait is issued only after the record it saw while recording).  Not re-entrant per handle. */
typedef struct gcc_replay gcc_replay_t;
int gcc_replay_begin(gcc_replay_t** out);
int gcc_replay_end(gcc_replay_t* r, int threads);
int gcc_replay_run(gcc_replay_t* r);
int gcc_replay_tag_next(int tag);
int gcc_replay_patch(gcc_replay_t* r, int tag, int arg_index, const void* value, size_t bytes);
long long gcc_replay_info(const gcc_replay_t* r, int what);   /* 0 entries, 1 kernel launches, 2 streams, 3 threads, 4 argument bytes */
int gcc_replay_destroy(gcc_replay_t* r);
/* events of the library (hipEventDisableTiming), recorded / waited for through it so that a recording sees them */
int gcc_event_create(void** event);
int gcc_event_destroy(void* event);
int gcc_event_record(void* event, gcc_stream_t stream);
int gcc_stream_wait_event(gcc_stream_t stream, void* event);
/* out[0] = 1 - beta1^step, out[1] = sqrt(1 - beta2^step) as gcc_adam_step passes them to its kernel */
int gcc_adam_factors(float beta1, float beta2, int step, float* out2);

#ifdef __cplusplus
}
#endif
#endif /* GCC_HIP_H */
