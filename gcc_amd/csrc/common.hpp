// Shared device helpers for the gfx950 kernels of libgcc_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <tuple>
#include <type_traits>
#include <utility>
#include "../../include/gcc_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef unsigned short bf16_t;  // raw bf16 bits

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even, NaN preserved by the hardware convert (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
__device__ __forceinline__ void unpack8(const i32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t u = (uint32_t)v[i];
        f[2 * i] = __uint_as_float(u << 16);
        f[2 * i + 1] = __uint_as_float(u & 0xffff0000u);
    }
}
__device__ __forceinline__ i32x4 pack8(const float* f) {
    i32x4 v;
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = (int)pack2bf(f[2 * i], f[2 * i + 1]);
    return v;
}

// `act` is a launch constant: the slope selection below is scalar (loop-invariant) work and the only branch is the uniform
// tanh test.  (A per-element `switch` compiled to a chain of scalar branches per element: the streaming kernels were bound by
// instruction issue, not by HBM -- bnact_fwd ran 5 us behind a plain copy of the same bytes.)
__device__ __forceinline__ float act_neg_slope(int act, float slope) {
    return act == GCC_ACT_LRELU ? slope : (act == GCC_ACT_RELU ? 0.f : 1.f);
}
__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    if (act == GCC_ACT_TANH) return tanhf(v);
    const float neg = act_neg_slope(act, slope);
    return v > 0.f ? v : v * neg;
}
// N channels at once: the tanh test leaves the element loop
template <int N>
__device__ __forceinline__ void apply_actN(const float* v, float* o, int act, float slope) {
    if (act == GCC_ACT_TANH) {
#pragma unroll
        for (int j = 0; j < N; j++) o[j] = tanhf(v[j]);
    } else {
        const float neg = act_neg_slope(act, slope);
#pragma unroll
        for (int j = 0; j < N; j++) o[j] = v[j] > 0.f ? v[j] : v[j] * neg;
    }
}
__device__ __forceinline__ void apply_act8(const float* v, float* o, int act, float slope) {
    if (act == GCC_ACT_TANH) {
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = tanhf(v[j]);
    } else {
        const float neg = act_neg_slope(act, slope);
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = v[j] > 0.f ? v[j] : v[j] * neg;
    }
}
// derivative of the activation evaluated on its OUTPUT y (the reference's in-place activations
// differentiate through the result: leaky_relu_backward(self_is_result), threshold_backward).
__device__ __forceinline__ float act_grad_from_out(float y, int act, float slope) {
    const float neg = act_neg_slope(act, slope);
    const float lin = y > 0.f ? 1.f : neg, th = 1.f - y * y;
    return act == GCC_ACT_TANH ? th : lin;
}

// counter-based RNG for dropout: splitmix64 of (seed, index) -> uniform [0,1)
__device__ __forceinline__ float rng_uniform(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// division by a launch-constant divisor: one multiply-high and a shift (exact for 0 <= n < 2^31)
struct FastDiv {
    uint32_t mul, shr;
    int d;
};
static FastDiv make_fastdiv(int d) {
    FastDiv f; f.d = d;
    if (d == 1) { f.mul = 0; f.shr = 0; return f; }
    int lg = 0;
    while ((1ll << lg) < d) lg++;
    const int p = 31 + lg;
    f.mul = (uint32_t)((((unsigned long long)1 << p) + d - 1) / d);
    f.shr = p - 32;
    return f;
}
__device__ __forceinline__ int fdiv(int n, const FastDiv& f) {
    return f.d == 1 ? n : (int)(__umulhi((uint32_t)n, f.mul) >> f.shr);
}

// process-wide tuning hooks (include/gcc_hip.h: gcc_set_option); defined in misc.hip
int gcc_opt(int id);
// per-call tile plan (gcc_conv_t.plan): a field that is 0 means the library's default
static inline int plan_or(int v, int def) { return v ? v : def; }
constexpr int PLAN_TILE_FAMILIES = 3, PLAN_BIG_MIN = 120, PLAN_BIG_NK = 24, PLAN_WGRAD_WGS_BIG = 256, PLAN_WGRAD_WGS = 512;
// Diagnostic ablations (timing probes whose RESULTS ARE WRONG, the test of the device error word) are compiled only into the
// GCC_DIAG_BUILD variant of the library (build.sh: libgcc_hip_diag.so); in the shipped one the bits are the constant 0 and the
// branches on them do not exist.  Bits: 2 the main loops issue no staging loads after the first step, 4 they re-load the first
// step's addresses, 8 weights every fourth step only, 32 s_memrealtime stamps of the grid InstanceNorm, 64 its exchange is made
// to time out (256 polls, workgroup 1 of every domain publishes nothing).
#ifdef GCC_DIAG_BUILD
int gcc_diag_bits();
#define GCC_DIAG(x) (x)
#else
static inline int gcc_diag_bits() { return 0; }
#define GCC_DIAG(x) 0
#endif
// Every kernel launch of the library goes through gcc_launch: it is counted (gcc_launch_count: bench.py reports launches per
// step) and, while the calling thread records (gcc_replay_begin, replay.hip), written down with its argument values so that
// gcc_replay_run can issue it again without the host code in front of it.  The launch itself is hipLaunchKernel on the kernel's
// host stub -- what `kernel<<<...>>>(...)` compiles to, minus the push / pop of the call configuration.
void gcc_count_launch();
struct GccLaunchRec {
    const void* func; dim3 grid, block; unsigned shmem; hipStream_t stream;
    int nargs; void* const* args; const unsigned* sizes;
};
bool gcc_replay_recording();
unsigned gcc_replay_generation();          // recordings begun so far in this process (replay.hip)
// pinned host word a waiting kernel stores a code into when a bounded spin expires (misc.hip; NULL if it could not be mapped)
unsigned* gcc_device_error_word();
extern "C" int gcc_device_error(int clear);
void gcc_replay_record_kernel(const GccLaunchRec& rec);
// the few non-kernel stream operations of the library, recorded the same way
hipError_t gcc_memset_async(void* dst, int value, size_t bytes, hipStream_t st);
hipError_t gcc_memcpy_d2d_async(void* dst, const void* src, size_t bytes, hipStream_t st);
// the C ABI's own all-reduce (comm.hip) as a recordable stream operation: while the calling thread records, the call is written
// down (communicator, buffer, count, element type 0 fp32 / 1 bf16, stream) and gcc_replay_run issues it again in recorded order
void gcc_replay_record_allreduce(void* comm, void* buf, size_t count, int dtype, hipStream_t st);
int gcc_internal_comm_allreduce(void* comm, void* buf, size_t count, int dtype, hipStream_t st);

template <typename T> struct GccArgBox { T v; };
template <typename Tuple, size_t... I>
static inline void gcc_fill_arg_ptrs(Tuple& t, void** ptrs, unsigned* sizes, std::index_sequence<I...>) {
    ((ptrs[I] = (void*)&std::get<I>(t).v, sizes[I] = (unsigned)sizeof(std::get<I>(t).v)), ...);
}
template <typename... P, typename... A>
static inline void gcc_launch(void (*kernel)(P...), dim3 grid, dim3 block, unsigned shmem, hipStream_t st, A&&... a) {
    static_assert(sizeof...(P) == sizeof...(A), "kernel launched with the wrong number of arguments");
    gcc_count_launch();
    // the arguments converted to the kernel's parameter types, one box each (hipLaunchKernel takes their addresses)
    constexpr int NA = (int)sizeof...(P);
    void* ptrs[NA > 0 ? NA : 1];
    unsigned sizes[NA > 0 ? NA : 1];
    auto boxes = std::tuple<GccArgBox<typename std::remove_cv<P>::type>...>{GccArgBox<typename std::remove_cv<P>::type>{static_cast<P>(a)}...};
    gcc_fill_arg_ptrs(boxes, ptrs, sizes, std::make_index_sequence<sizeof...(P)>{});
    if (__builtin_expect(gcc_replay_recording(), 0))
        gcc_replay_record_kernel(GccLaunchRec{(const void*)kernel, grid, block, shmem, st, NA, ptrs, sizes});
    (void)hipLaunchKernel((const void*)kernel, grid, block, ptrs, shmem, st);
}
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...) \
    gcc_launch((kernelName), dim3(numBlocks), dim3(numThreads), (unsigned)(memPerBlock), (streamId), __VA_ARGS__)

__host__ __device__ static inline int ceil8(int v) { return (v + 7) & ~7; }
__host__ __device__ static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

#include <stdio.h>
// A channel dimension may be the concatenation of two tensors (U-Net skip | up path).  In memory each
// part is padded to 8 channels, so logical channel l of an n-channel dimension whose first part has
// `split` channels lives at l (+ ceil8(split) - split when l >= split).  split <= 0 or >= n: one part.
__host__ __device__ static inline int seg_phys_size(int n, int split) {
    return (split > 0 && split < n) ? ceil8(split) + ceil8(n - split) : ceil8(n);
}
__host__ __device__ static inline int seg_to_phys(int l, int n, int split) {
    return (split > 0 && split < n && l >= split) ? l + ceil8(split) - split : l;
}
__host__ __device__ static inline int seg_to_logical(int p, int n, int split) {   // -1: padding
    if (split <= 0 || split >= n) return p < n ? p : -1;
    if (p < split) return p;
    const int s8 = ceil8(split);
    if (p < s8) return -1;
    const int l = p - (s8 - split);
    return l < n ? l : -1;
}

#define GCC_CHECK_LAUNCH()                                                                      \
    do {                                                                                        \
        hipError_t e_ = hipGetLastError();                                                      \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "[libgcc_hip] %s:%d launch failed: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return GCC_ERR_LAUNCH;                                                              \
        }                                                                                       \
    } while (0)
// errors left behind by other users of the HIP runtime in this thread (e.g. a failed capability
// probe) must not be reported as ours: every entry point starts from a clean error state
#define GCC_ENTER() (void)hipGetLastError()

// LDS-DMA (buffer_load_dwordx4 ... lds: 1 KiB per wave, lane l lands at lds_base + 16 l) issued from inline assembly.  Through
// the builtin hipcc (ROCm 7.2) counts the DMA as a pending LDS write and, in front of the ds_read_b64_tr_b16 fragment reads
// (an intrinsic it has no alias information for), emitted `s_waitcnt vmcnt(0)`: every k-step waited for the loads it had just
// issued before touching the OTHER stage -- the prefetch never overlapped the MFMAs (L4 weight gradient 314 us with loads from
// fixed, cache-hot addresses against 201 us without loads, profiles/r3f_wgrad_scalar.txt).  An asm statement is opaque to that
// bookkeeping (cdna_hip_programming.md 5.7 item 1); the loop's own `s_waitcnt vmcnt(0)` + barrier in front of the stage's first
// read is what orders the data.  M0 (the DMA's LDS base) is written in the same statement that uses it; nothing else in this
// kernel depends on M0 (no LDS-DMA builtin is left).
__device__ __forceinline__ void lds_dma16(const i32x4& rsrc, uint32_t lds_base, uint32_t voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_base), "v"(voff), "s"(rsrc) : "memory", "m0");
}
__device__ __forceinline__ i32x4 make_rsrc(const void* ptr, uint32_t bytes) {
    const uint64_t a = (uint64_t)ptr;
    return i32x4{(int)(uint32_t)a, (int)((uint32_t)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}

// ---- BatchNorm statistics: the canonical fold of the per-tile partial sums (round 4) ---------------------------------------
// Every producer of training statistics (the conv epilogues, the split-K fold kernel, gcc_channel_stats) writes rows of fp32
// partial sums [row][2][C].  Their fold has ONE order, whoever performs it -- bn_finalize_kernel (norm_act.hip) or the
// last-arriving workgroups of the producing launch (stats_tail, igemm_common.hpp): rows are summed in double, ascending,
// inside groups of FIN_GROUP consecutive rows; the group sums are summed in double, ascending.  Same bits either way, which is
// what lets a pass that ran ahead of its place (the early D(real) pass) replay its finalize later with the running-statistics
// update, and lets tests compare the two routes bit for bit.
constexpr int FIN_GROUP = 16;
__device__ __forceinline__ void bn_channel_finalize(double s, double ss, double count, float eps, float momentum, const float* gamma,
                                                    const float* beta, int c, float* rmean, float* rvar, float* mean, float* rstd,
                                                    float* scale, float* shift) {
    const double m = s / count;
    double var = ss / count - m * m;
    if (var < 0.0) var = 0.0;
    const float r = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    if (mean) mean[c] = (float)m;
    if (rstd) rstd[c] = r;
    scale[c] = g * r;
    shift[c] = b - (float)m * g * r;
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
    if (rvar) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
}

// a partial sum another workgroup of the same launch may read (stats_tail, igemm_common.hpp; bwd_tail, norm_act.hip): sc1 = write-through
__device__ __forceinline__ void st_stat(float* p, float v, bool sc1) {
    if (sc1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// gcc_conv_bn_act's split layers, second launch (norm_act.hip bn_fold_grid_kernel; GCC_OPT_FUSE_BN 3): the K slices of the
// partial-tile launch folded, BatchNorm statistics exchanged inside the launch, every row normalised from registers
struct BnFoldDesc {
    const float* part; int ksplit, rows_max, Cpad, phases;        // [phase][slice][rows_max][Cpad] fp32
    int N, Hd, Wd, stride, dgrad;                                  // output geometry; dgrad: rows of phase (py, px) are the sub-grid
    void* raw; int ldraw, rawoff;                                  // bf16 raw output (kept for the backward pass)
    void* y; int ldy, yoff; void* y2; int ldy2, y2off;
    int C; gcc_bn_t bn; int act, act2; float slope, drop_p; unsigned long long seed;
    void* ws; size_t ws_bytes;                                     // GCC_INORM_WORKSPACE_BYTES, zero-filled once, one stream
};
int gcc_internal_bn_fold_grid(const BnFoldDesc* d, hipStream_t st);      // GCC_ERR_UNSUPPORTED: geometry outside the plan

// XCD-aware remap of a linear workgroup id (8 XCDs, round-robin dispatch): logical tiles that are
// adjacent end up on the same XCD (shared L2).  Bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
