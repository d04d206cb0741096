// YAMNet (MobileNetV1) body on gfx950: embedders/yamnet/yamnet.py:36-106 with the BatchNorms
// (yamnet.py:26-33, scale=False, eps=1e-4) folded into the convolution weights at load time.
// All activations are NHWC float32, exactly the reference's layout.
// TF "SAME" for an even extent with stride 2 pads 0 before / 1 after; stride 1 pads 1 / 1.
//
// Default path (7 launches per pass): stem_reg_kernel (stemreg.hip: layers 1-3), l4_window_kernel (here),
// sep_mid_kernel (sepmid.hip: pointwise 5 + layers 6-7), sep_chip_kernel (sepchip.hip: layers 8-12 + depthwise 13),
// tail_gemm_kernel twice (septail.hip: pointwise 13 + depthwise 14, pointwise 14 + pool), pool_head_kernel<1>.
// The kernels of this file:
//   stem3_kernel         layers 1-3 as one kernel, a workgroup per row block (the default of rounds 2-4; bd_set_fusion stem = 5)
//   l4_window_kernel     layer 4 + depthwise 5: persistent workgroups walk whole windows two map rows at a time
//   pw_res_kernel        the 1x1 convolutions of layers 5 and 7: persistent, weights in registers (separable = 10; one kernel per op)
//   sep_ws_kernel        wave-specialised 96 x 256 tiles (4 producer + 4 MFMA waves, slab ring by LDS-DMA): NDW = 1 = layer 6 +
//                        depthwise 7 (separable = 10); PWO = a plain 1x1 convolution of a wide layer (one kernel per op)
//   pool_head_kernel<1>  Dense(1024 -> n_classes) on the pooled embeddings
// Reference kernels, one per op (the fused ones are tested bit for bit against them; they are also the exact-f32 mode's tail):
//   conv1_kernel, depthwise_kernel, pointwise_f16x3_kernel (split-f16), pointwise_kernel (exact-f32 MFMA), pool_head_kernel<6>
// Workgroup -> tile mapping is XCD-aware (tile_of).  The launch path reads no environment and keeps no mutable state besides
// the once-per-device dynamic-LDS attribute flags.
#include "bd_internal.h"
#include <mutex>
#include <type_traits>

namespace bd {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Range guard of the f16 arithmetic modes.  Every activation is split as hi = f16(a), lo = f16(a - hi): beyond the f16
// range (65 504) hi is +inf and the result is garbage that the following ReLU can even hide (max(NaN, 0) = 0).  The
// kernels keep a running max |a| of what they convert (two v_max3 per four values) and raise the engine's sticky flag
// when it is out of range; the host reads the flag with the results and repeats the chunk in exact-f32 mode.
constexpr float kF16Max = 65504.0f;
__device__ __forceinline__ float range_of(float m, float4 v) {
    return fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fmaxf(fabsf(v.y), fabsf(v.z)), fabsf(v.w)));
}
typedef float v4f_range __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float range_of(float m, v4f_range v) {
    return fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fmaxf(fabsf(v.y), fabsf(v.z)), fabsf(v.w)));
}
__device__ __forceinline__ void range_report(float m, unsigned* __restrict__ flag) {
    if (flag && !(m <= kF16Max)) *flag = 1u;  // also true for NaN (flag == nullptr: the handle-less debug entry points)
}

// hipFuncSetAttribute(max dynamic LDS) once per kernel instantiation and device; safe when several analyzer threads
// (one engine each, src/inference/worker.py:21) make their first launch at the same time.
constexpr int kMaxDevices = 64;
template <typename Kernel>
void allow_dynamic_lds(Kernel kernel, int bytes, std::once_flag (&once)[kMaxDevices]) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & (kMaxDevices - 1)], [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    });
}

// --------------------------------------------------------------------------- conv1
constexpr int kC1Rows = 4;   // output rows per workgroup

__global__ __launch_bounds__(256) void conv1_kernel(const float* __restrict__ logmel, int patch_step,
                                                    const WindowMap map, int w0,
                                                    const float* __restrict__ w9x32,
                                                    const float* __restrict__ b32,
                                                    float* __restrict__ out) {
    // grid: (48 / kC1Rows, windows); thread = (ow = tid >> 3, c4 = tid & 7)
    const int tid = threadIdx.x;
    const int c4 = tid & 7;
    const int ow = tid >> 3;
    const int win = blockIdx.y;
    const float* patch = logmel + window_frame(map, w0 + win, patch_step) * BD_MEL_BANDS;

    float4 wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t] = reinterpret_cast<const float4*>(w9x32 + t * 32)[c4];
    const float4 bias = reinterpret_cast<const float4*>(b32)[c4];

#pragma unroll
    for (int rr = 0; rr < kC1Rows; ++rr) {
        const int oh = blockIdx.x * kC1Rows + rr;
        float4 acc = bias;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = 2 * oh + kh;                      // pad_top = 0
            if (ih < BD_PATCH_FRAMES) {
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int iw = 2 * ow + kw;              // pad_left = 0
                    const float v = iw < BD_MEL_BANDS ? patch[ih * BD_MEL_BANDS + iw] : 0.0f;
                    const float4 w = wt[kh * 3 + kw];
                    acc.x = fmaf(v, w.x, acc.x);
                    acc.y = fmaf(v, w.y, acc.y);
                    acc.z = fmaf(v, w.z, acc.z);
                    acc.w = fmaf(v, w.w, acc.w);
                }
            }
        }
        acc.x = fmaxf(acc.x, 0.0f);
        acc.y = fmaxf(acc.y, 0.0f);
        acc.z = fmaxf(acc.z, 0.0f);
        acc.w = fmaxf(acc.w, 0.0f);
        reinterpret_cast<float4*>(out + (((size_t)win * 48 + oh) * 32 + ow) * 32)[c4] = acc;
    }
}

// --------------------------------------------------------------------------- depthwise
// One thread = OWB consecutive outputs of one row for one float4 of channels: the 3 x ((OWB-1)*S+3)
// input patch is loaded once into registers and reused by the OWB outputs (4.5 instead of 9 loads per
// output at stride 1), the 9 taps are loaded once per thread.  Channels are the fastest thread index,
// so a wavefront reads/writes whole 128-byte-or-longer runs of the NHWC rows.
// TF SAME: stride 1 pads 1 before; stride 2 on the even extents used here pads 0 before, 1 after.
template <int STRIDE, int OWB>
__global__ __launch_bounds__(256) void depthwise_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        const float* __restrict__ w9xc,
                                                        const float* __restrict__ bias, int windows, int H,
                                                        int W, int C, int OH, int OW,
                                                        unsigned* __restrict__ amax) {
    constexpr int PAD = STRIDE == 1 ? 1 : 0;
    float cmax = 0.0f;                       // calibration pass (amax != null): largest output of this thread
    constexpr int NCOL = (OWB - 1) * STRIDE + 3;
    const int c4n = C >> 2;
    const int owg = OW / OWB;
    const long long total = (long long)windows * OH * owg * c4n;
    // workgroups go to the XCDs round-robin by ID: give every XCD a contiguous run of the index space, so that
    // IDs b and b + 8 (same L2, dispatched together) are neighbours and the input rows they share are fetched once
    const unsigned per_ = gridDim.x >> 3;
    const unsigned bid = blockIdx.x < per_ * 8 ? (blockIdx.x & 7) * per_ + (blockIdx.x >> 3) : blockIdx.x;
    for (long long i = bid * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
        const int c4 = (int)(i % c4n);
        long long t = i / c4n;
        const int og = (int)(t % owg);
        t /= owg;
        const int oh = (int)(t % OH);
        const long long n = t / OH;
        const int ow0 = og * OWB;
        const float4* src = reinterpret_cast<const float4*>(in + (size_t)n * H * W * C) + c4;
        float4 wt[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wt[k] = reinterpret_cast<const float4*>(w9xc + k * C)[c4];
        const float4 b = reinterpret_cast<const float4*>(bias)[c4];
        float4 acc[OWB];
#pragma unroll
        for (int o = 0; o < OWB; ++o) acc[o] = b;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * STRIDE + kh - PAD;
            const bool row_ok = ih >= 0 && ih < H;
            float4 col[NCOL];
#pragma unroll
            for (int c = 0; c < NCOL; ++c) {
                const int iw = ow0 * STRIDE + c - PAD;
                col[c] = (row_ok && iw >= 0 && iw < W) ? src[((size_t)ih * W + iw) * c4n]
                                                       : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int o = 0; o < OWB; ++o)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float4 v = col[o * STRIDE + kw];
                    const float4 w = wt[kh * 3 + kw];
                    acc[o].x = fmaf(v.x, w.x, acc[o].x);
                    acc[o].y = fmaf(v.y, w.y, acc[o].y);
                    acc[o].z = fmaf(v.z, w.z, acc[o].z);
                    acc[o].w = fmaf(v.w, w.w, acc[o].w);
                }
        }
        float4* dst = reinterpret_cast<float4*>(out + (((size_t)n * OH + oh) * OW + ow0) * C) + c4;
#pragma unroll
        for (int o = 0; o < OWB; ++o) {
            float4 r = acc[o];
            r.x = fmaxf(r.x, 0.0f);
            r.y = fmaxf(r.y, 0.0f);
            r.z = fmaxf(r.z, 0.0f);
            r.w = fmaxf(r.w, 0.0f);
            dst[(size_t)o * c4n] = r;
            cmax = fmaxf(fmaxf(cmax, r.x), fmaxf(fmaxf(r.y, r.z), r.w));
        }
    }
    if (amax) {                              // outputs are >= 0 (or NaN, which fmaxf drops): float bits order like unsigned
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, o, 64));
        if ((threadIdx.x & 63) == 0 && cmax > 0.0f) atomicMax(amax, __float_as_uint(cmax));
    }
}

// dst = src * factor (the stage tap of a depthwise output in the f16 modes: factor = 2^-act_exp, exact)
__global__ __launch_bounds__(256) void scale_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n,
                                                         float factor) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) dst[i] = src[i] * factor;
}

// --------------------------------------------------------------------------- pointwise GEMM
// C[m][n] = relu(sum_k A[m][k] * Wt[n][k] + bias[n]).  A = NHWC activations flattened to
// [rows = windows*H*W][K = Cin]; Wt = folded kernel stored [Cout][Cin] so that both operands
// have K contiguous and one ds_read_b128 feeds four MFMA k-steps.
//
// Workgroup = WGM x WGN waves on a BM x BN tile, BK = 32 per LDS stage, register-staged double
// buffering: the global_load_dwordx4 of stage t+1 are issued before the MFMAs of stage t and
// written to the other LDS buffer after them, one barrier per stage.
// v_mfma_f32_32x32x2_f32 operand map: lane l supplies A[i = l & 31][k = l >> 5] and
// B[k = l >> 5][j = l & 31]; which k a step covers is free as long as both operands agree, so
// lane-half h takes k = 8*s + 4*h + j for the j-th MFMA of super-step s (one b128 per operand).
// Rows past M are loaded from row M-1 (always in bounds) and never stored.
#ifndef BD_PW_ABLATE
#define BD_PW_ABLATE 0      // developer builds: 1 = no global loads in the loop, 2 = no fragment reads, 4 = no MFMAs, 8 = no stores
#endif
constexpr int kBK = 32;
constexpr int kLds = kBK + 4;   // row stride in floats: 144 B = odd multiple of 16 B -> conflict-free b128
// an operand fragment from LDS (developer build 2: a register value instead; by value, so that no address escapes)
typedef float pw4 __attribute__((ext_vector_type(4)));   // native vectors: SSA values, nothing for the compiler to keep in memory
// operand pointers that are re-pointed inside the tile loop: the address space is lost through the loop's phi nodes and the
// loads become flat_load (counted in lgkmcnt too: every LDS wait then waits for them) unless it is spelled out
typedef const __attribute__((address_space(1))) float* pw_gptr;
typedef const __attribute__((address_space(1))) pw4* pw_gptr4;
__device__ __forceinline__ pw4 pw_frag(const float* p, pw4 instead) {
    if (BD_PW_ABLATE & 2) return instead;
    return *reinterpret_cast<const pw4*>(p);
}

// With NH > 0 the kernel also applies the NEXT layer's depthwise 3x3 (stride NS, TF SAME padding) to its output, an NH x NW
// map per window: a 96-row tile is whole windows (one 12 x 8 map, four 6 x 4 maps or sixteen 3 x 2 maps), so the tile goes
// through LDS as f32 (bias + ReLU applied) and what reaches HBM - `C` - is the depthwise output [windows][NH / NS][NW / NS][N];
// the 1x1 output itself never exists.  depthwise_kernel's chain (shift, the taps in (kh, kw) order, zeros outside the map).
template <int BM, int BN, int WGM, int WGN, int NH = 0, int NW = 0, int NS = 1>
__global__ __launch_bounds__(WGM* WGN * 64, (NH > 0 ? 2 : 1)) void pointwise_kernel(const float* __restrict__ A,
                                                                  const float* __restrict__ Wt,
                                                                  const float* __restrict__ bias,
                                                                  float* __restrict__ C, long long M, int N,
                                                                  int K, int tiles_n, long long tiles, int xcd_map,
                                                                  const float* __restrict__ ndw_w,
                                                                  const float* __restrict__ ndw_b, int windows) {
    constexpr int NT = WGM * WGN * 64;
    constexpr bool NDW = NH > 0;
    static_assert(!NDW || (BM % (NH * NW) == 0 && BN == 128 && NT % 32 == 0), "whole windows per tile");
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int RPP = NT / 8;                  // tile rows covered by one float4-per-thread pass
    constexpr int LA = BM / RPP, LB = BN / RPP;  // float4 global loads per thread per stage
    static_assert(BM % RPP == 0 && BN % RPP == 0 && WM % 32 == 0 && WN % 32 == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;                      // [2][BM * kLds]
    float* const Bs = smem + 2 * BM * kLds;      // [2][BN * kLds]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / WGN, wc = wave % WGN;
    const int lrow = tid >> 3;   // 0 .. RPP-1
    const int lc4 = tid & 7;     // float4 column within the 32-wide k slab
    const int st_off = lrow * kLds + lc4 * 4;
    const int frow = lane & 31;
    const int fk = (lane >> 5) * 4;
    const int half = lane >> 5;
    const float* const a_frag = As + (wr * WM + frow) * kLds + fk;
    const float* const b_frag = Bs + (wc * WN + frow) * kLds + fk;

    // Persistent workgroups walk tiles blockIdx.x, + gridDim.x, ... as ONE stream of K stages: the first stage of the next
    // tile is loaded behind the last product of this one, and this tile's stores drain behind the next tile's products (a
    // workgroup that ends on its stores holds its CU share until they retire: 14 of 124 us on a 512 -> 512 layer).
    // Tile order: the n-tiles of one m-tile are neighbours; with xcd_map (m-tiles a multiple of 8) they sit on ONE XCD
    // (workgroups go to the XCDs round-robin by ID and the grid is a multiple of 8), so the A rows they share come from one L2.
#define BD_PW_ORIGIN(T_, M0_, N0_)                                                   \
    {                                                                                \
        long long tile_m_;                                                           \
        int tile_n_;                                                                 \
        if (xcd_map) {                                                               \
            const long long idx_ = (T_) >> 3;                                        \
            tile_n_ = (int)(idx_ % tiles_n);                                         \
            tile_m_ = (idx_ / tiles_n) * 8 + ((T_) & 7);                             \
        } else {                                                                     \
            tile_m_ = (T_) / tiles_n;                                                \
            tile_n_ = (int)((T_) % tiles_n);                                         \
        }                                                                            \
        M0_ = tile_m_ * BM;                                                          \
        N0_ = tile_n_ * BN;                                                          \
    }
    // rows past M are loaded from row M-1 (always in bounds) and never stored
#define BD_PW_POINT(M0_, N0_)                                                        \
    {                                                                                \
        _Pragma("unroll") for (int i = 0; i < LA; ++i) {                             \
            long long m_ = (M0_) + lrow + RPP * i;                                   \
            m_ = m_ < M ? m_ : M - 1;                                                \
            ap[i] = (pw_gptr)A + (size_t)m_ * K + lc4 * 4;                           \
        }                                                                            \
        _Pragma("unroll") for (int i = 0; i < LB; ++i) bp[i] = (pw_gptr)Wt + (size_t)((N0_) + lrow + RPP * i) * K + lc4 * 4; \
    }
    pw_gptr ap[LA];
    pw_gptr bp[LB];

    long long t = blockIdx.x;
    if (t >= tiles) return;
    long long m0;
    int n0;
    BD_PW_ORIGIN(t, m0, n0)
    BD_PW_POINT(m0, n0)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    pw4 ra[LA], rb[LB];
#pragma unroll
    for (int i = 0; i < LA; ++i) ra[i] = *(pw_gptr4)(ap[i]);
#pragma unroll
    for (int i = 0; i < LB; ++i) rb[i] = *(pw_gptr4)(bp[i]);
#pragma unroll
    for (int i = 0; i < LA; ++i) *reinterpret_cast<pw4*>(As + st_off + RPP * i * kLds) = ra[i];
#pragma unroll
    for (int i = 0; i < LB; ++i) *reinterpret_cast<pw4*>(Bs + st_off + RPP * i * kLds) = rb[i];
    __syncthreads();

#define BD_PW_COMPUTE(BUF)                                                                              \
    {                                                                                                   \
        const float* as_ = a_frag + (BUF) * BM * kLds;                                                  \
        const float* bs_ = b_frag + (BUF) * BN * kLds;                                                  \
        _Pragma("unroll") for (int s = 0; s < kBK / 8; ++s) {                                           \
            pw4 av[TM], bv[TN];                                                                      \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) av[i] = pw_frag(as_ + i * 32 * kLds + s * 8, ra[0]); \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) bv[j] = pw_frag(bs_ + j * 32 * kLds + s * 8, rb[0]); \
            if (BD_PW_ABLATE & 4) {                                                                     \
                _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) \
                    acc[i][j][s] += av[i].x * bv[j].y;                                                  \
            } else                                                                                      \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) { \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[j].x, acc[i][j], 0, 0, 0); \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[j].y, acc[i][j], 0, 0, 0); \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[j].z, acc[i][j], 0, 0, 0); \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[j].w, acc[i][j], 0, 0, 0); \
            }                                                                                           \
        }                                                                                               \
    }

    const int nk = K / kBK;
    int buf = 0;
    for (;;) {
        const long long tn = t + gridDim.x;
        const bool more = tn < tiles;
        long long nm0 = 0;
        int nn0 = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const bool last = kt + 1 == nk;
            const bool feed = !last || more;     // a stage follows this one: load it now, store it to LDS after the product
            int koff = (kt + 1) * kBK;
            if (last) {                          // ... the first stage of the next tile
                koff = 0;
                if (more) {
                    BD_PW_ORIGIN(tn, nm0, nn0)
                    BD_PW_POINT(nm0, nn0)
                }
            }
            // unconditional (the very last stage of a workgroup re-reads its tile's first stage and drops it): loads under a
            // branch make the compiler wait for them at the join, in FRONT of the product they are meant to hide behind
            if (!(BD_PW_ABLATE & 1)) {
#pragma unroll
                for (int i = 0; i < LA; ++i) ra[i] = *(pw_gptr4)(ap[i] + koff);
#pragma unroll
                for (int i = 0; i < LB; ++i) rb[i] = *(pw_gptr4)(bp[i] + koff);
            }
            asm volatile("" ::: "memory");       // keeps the loads HERE: their only use is under `feed`, and the compiler sinks them there
            BD_PW_COMPUTE(buf)
            if (feed && !(NDW && last)) {        // (with the depthwise epilogue the tile passes through these buffers first)
                float* an = As + (buf ^ 1) * BM * kLds + st_off;
                float* bn = Bs + (buf ^ 1) * BN * kLds + st_off;
#pragma unroll
                for (int i = 0; i < LA; ++i) *reinterpret_cast<pw4*>(an + RPP * i * kLds) = ra[i];
#pragma unroll
                for (int i = 0; i < LB; ++i) *reinterpret_cast<pw4*>(bn + RPP * i * kLds) = rb[i];
                __syncthreads();
            }
            buf ^= 1;
        }
        if constexpr (NDW) {
            constexpr int PWN = NH * NW, WPT = BM / PWN;         // positions per window, windows per tile
            constexpr int OH = NH / (NS ? NS : 1), OW = NW / (NS ? NS : 1), OPW = OH * OW;
            constexpr int PAD = NS == 1 ? 1 : 0;
            constexpr int PS = BN + 4;                           // row stride of the f32 tile
            constexpr int TASKS = WPT * OPW * (BN / 4);
            float* const P = smem;                               // [BM + 1][PS], over the stage buffers
            const int c4 = tid & (BN / 4 - 1);                   // this thread's channel quad (NT is a multiple of 32)
            pw4 tw[9];
            pw4 tb = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < 9; ++k) tw[k] = *(pw_gptr4)((pw_gptr)ndw_w + (size_t)k * N + n0 + c4 * 4);
            tb = *(pw_gptr4)((pw_gptr)ndw_b + n0 + c4 * 4);
            __syncthreads();                                     // every wave has read its last fragments
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float b = bias[n0 + wc * WN + j * 32 + frow];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        P[(wr * WM + i * 32 + 4 * half + (r & 3) + 8 * (r >> 2)) * PS + wc * WN + j * 32 + frow] = fmaxf(acc[i][j][r] + b, 0.0f);
                        acc[i][j][r] = 0.0f;
                    }
            }
            if (tid < BN / 4) *reinterpret_cast<pw4*>(P + BM * PS + tid * 4) = pw4{0.0f, 0.0f, 0.0f, 0.0f};   // what a tap outside the map reads
            __syncthreads();
            const long long win0 = m0 / PWN;
            const bool whole = win0 + WPT <= windows;            // (stores under a branch each wait for the one before)
            if constexpr (NS == 1) {
                // stride 1: a thread owns a COLUMN of a window's map (and a channel quad) and walks the input rows once - each row
                // is tap row 0 of the output below it, 1 of its own, 2 of the one above - so an output's chain still runs
                // in (kh, kw) order, with 3 NH LDS reads per NH outputs instead of 9 NH
                constexpr int ITEMS = WPT * NW * (BN / 4);
                static_assert(ITEMS % NT == 0, "whole rounds of column items");
#pragma unroll 1
                for (int u = 0; u < ITEMS / NT; ++u) {
                    const int q = (tid + NT * u) / (BN / 4);
                    const int col = q % NW, w = q / NW;
                    pw4 o[NH];
#pragma unroll
                    for (int r = 0; r < NH; ++r) o[r] = tb;
#pragma unroll
                    for (int r = -1; r <= NH; ++r) {
                        pw4 v[3];
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const int cc = col - 1 + kw;
                            const bool ok = r >= 0 && r < NH && cc >= 0 && cc < NW;
                            v[kw] = *reinterpret_cast<const pw4*>(P + (ok ? w * PWN + r * NW + cc : BM) * PS + c4 * 4);    // row BM: zeros
                        }
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            if (r + 1 < NH) o[r + 1 < NH ? r + 1 : 0] = __builtin_elementwise_fma(v[kw], tw[kw], o[r + 1 < NH ? r + 1 : 0]);
                        }
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            if (r >= 0 && r < NH) o[r >= 0 && r < NH ? r : 0] = __builtin_elementwise_fma(v[kw], tw[3 + kw], o[r >= 0 && r < NH ? r : 0]);
                        }
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            if (r >= 1) o[r >= 1 ? r - 1 : 0] = __builtin_elementwise_fma(v[kw], tw[6 + kw], o[r >= 1 ? r - 1 : 0]);
                        }
                    }
                    if (whole || win0 + w < windows) {
                        float* const dst = C + (((size_t)(win0 + w) * OH) * OW + col) * N + n0 + c4 * 4;
#pragma unroll
                        for (int r = 0; r < NH; ++r) {
                            pw4 a = o[r];
                            a.x = fmaxf(a.x, 0.0f);
                            a.y = fmaxf(a.y, 0.0f);
                            a.z = fmaxf(a.z, 0.0f);
                            a.w = fmaxf(a.w, 0.0f);
                            *reinterpret_cast<pw4*>(dst + (size_t)r * OW * N) = a;
                        }
                    }
                }
            } else {
#pragma unroll 3
                for (int u = 0; u < (TASKS + NT - 1) / NT; ++u) {
                    const int idx = tid + NT * u;
                    const int pos = idx / (BN / 4);
                    const int w = pos / OPW, o = pos % OPW, oh = o / OW, ow = o % OW;
                    pw4 a = tb;
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const int ih = oh * NS + kh - PAD, iw = ow * NS + kw - PAD;
                            const bool ok = ih >= 0 && ih < NH && iw >= 0 && iw < NW && (TASKS % NT == 0 || idx < TASKS);
                            const pw4 v = *reinterpret_cast<const pw4*>(P + (ok ? (w * PWN + ih * NW + iw) : BM) * PS + c4 * 4);   // row BM: zeros
                            a = __builtin_elementwise_fma(v, tw[kh * 3 + kw], a);
                        }
                    a.x = fmaxf(a.x, 0.0f);
                    a.y = fmaxf(a.y, 0.0f);
                    a.z = fmaxf(a.z, 0.0f);
                    a.w = fmaxf(a.w, 0.0f);
                    float* const dst = C + (((size_t)(win0 + w) * OH + oh) * OW + ow) * N + n0 + c4 * 4;
                    if (whole && TASKS % NT == 0) *reinterpret_cast<pw4*>(dst) = a;
                    else if ((TASKS % NT == 0 || idx < TASKS) && win0 + w < windows) *reinterpret_cast<pw4*>(dst) = a;
                }
            }
            if (more) {                                          // the next tile's first stage, held in registers since the last product
                __syncthreads();                                 // the tile has been read
                float* an = As + buf * BM * kLds + st_off;
                float* bn = Bs + buf * BN * kLds + st_off;
#pragma unroll
                for (int i = 0; i < LA; ++i) *reinterpret_cast<pw4*>(an + RPP * i * kLds) = ra[i];
#pragma unroll
                for (int i = 0; i < LB; ++i) *reinterpret_cast<pw4*>(bn + RPP * i * kLds) = rb[i];
                __syncthreads();
            }
        } else {
        // epilogue: C/D map of the 32x32 tile: col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5).  A tile inside
            // M stores without a branch per row: under a branch each store gets its own s_waitcnt vmcnt(0) (for the bias load),
            // which on gfx9 also waits for the store before it - sixteen memory round trips in a row per 32 x 32 tile.
            const bool whole = m0 + BM <= M;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wc * WN + j * 32 + frow;
                const float b = bias[n];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const long long mb = m0 + wr * WM + i * 32 + 4 * half;
                    float* const crow = C + (size_t)mb * N + n;
                    if (whole && !(BD_PW_ABLATE & 8)) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) crow[(size_t)((r & 3) + 8 * (r >> 2)) * N] = fmaxf(acc[i][j][r] + b, 0.0f);
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const long long m = mb + (r & 3) + 8 * (r >> 2);
                            if (m < M && (!(BD_PW_ABLATE & 8) || acc[i][j][r] == 12345.678f)) C[(size_t)m * N + n] = fmaxf(acc[i][j][r] + b, 0.0f);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
                }
            }
        }
        if (!more) break;
        t = tn;
        m0 = nm0;
        n0 = nn0;
    }
#undef BD_PW_COMPUTE
#undef BD_PW_ORIGIN
#undef BD_PW_POINT
}

// --------------------------------------------------------------------------- pointwise GEMM, split-f16
// Same contraction on the f16 matrix cores (v_mfma_f32_32x32x16_f16, 16x the f32-MFMA rate) without
// giving up f32 accuracy: every f32 operand x is carried as two halves x = hi + lo with
// hi = f16(x), lo = f16(x - hi)  (22 significand bits), and
//     a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi      (the dropped a_lo*b_lo term is ~2^-22 relative)
// Products of two f16 are exact in f32 and the MFMA accumulates in f32, so the result differs from the
// f32 chain by ~1e-7 relative — measured on the whole network: same max |dlogit| vs the f64 oracle as
// the exact-f32 kernel.  Weights are split once on the host; activations are split while they are staged
// from HBM into LDS.
//
// LDS tiles are [rows][32 f16] = 64-byte rows with no padding; the 16-byte slot index is XORed with
// (row >> 2) & 3 so that the 16 rows a ds_read_b128 lane group touches land on 16 different slots of
// the 256-byte bank row.  Operand map of v_mfma_f32_32x32x16_f16: lane l supplies
// A[row l & 31][k = 8*(l >> 5) + j] and B[k = 8*(l >> 5) + j][col l & 31], j = 0..7.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int swz64(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

// a = hi + lo with hi = f16(a) and lo = f16(a - hi), four values at a time.  The difference and its rounding are ONE
// v_fma_mixlo/mixhi_f16 per value (fma(hi as f16, -1, a as f32), rounded to f16 into one half of the result): six
// instructions per four values where the convert / subtract / convert form took eleven.  a - hi is exact in f32, so the
// bits are those of (_Float16)(a - (float)hi).
__device__ __forceinline__ void split_f16(float x, float y, float z, float w, f16x4& hi, f16x4& lo) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const f16x2 h0 = {(_Float16)x, (_Float16)y}, h1 = {(_Float16)z, (_Float16)w};
    f16x2 l0, l1;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(h0), "v"(x));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l0) : "v"(h0), "v"(y));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(h1), "v"(z));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l1) : "v"(h1), "v"(w));
    hi[0] = h0[0]; hi[1] = h0[1]; hi[2] = h1[0]; hi[3] = h1[1];
    lo[0] = l0[0]; lo[1] = l0[1]; lo[2] = l1[0]; lo[3] = l1[1];
}

// Epilogue of a 32x32 tile accumulated TRANSPOSED (weights fed as the MFMA "A" operand, activations as
// "B"): lane l then owns output row m = l & 31 and, per register quad g, the four consecutive channels
// n = 8 g + 4 (l >> 5) + 0..3 - one 16-byte store per quad instead of four 4-byte ones.  Products and
// their k order are those of the untransposed form.
__device__ __forceinline__ void store_tile_t(const f32x16& acc, const float* __restrict__ unscale_n,
                                             const float* __restrict__ bias_n, float* crow, bool live, int half) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(bias_n + 8 * g + 4 * half);
        const float4 u = *reinterpret_cast<const float4*>(unscale_n + 8 * g + 4 * half);
        float4 v;
        v.x = fmaxf(fmaf(acc[4 * g + 0], u.x, b.x), 0.0f);
        v.y = fmaxf(fmaf(acc[4 * g + 1], u.y, b.y), 0.0f);
        v.z = fmaxf(fmaf(acc[4 * g + 2], u.z, b.z), 0.0f);
        v.w = fmaxf(fmaf(acc[4 * g + 3], u.w, b.w), 0.0f);
        if (live) *reinterpret_cast<float4*>(crow + 8 * g + 4 * half) = v;
    }
}

template <int BM, int BN, int WGM, int WGN, bool PLAIN>
__global__ __launch_bounds__(WGM* WGN * 64) void pointwise_f16x3_kernel(
    const float* __restrict__ A, const _Float16* __restrict__ Whi, const _Float16* __restrict__ Wlo,
    const float* __restrict__ unscale, const float* __restrict__ bias, float* __restrict__ C, long long M, int N, int K,
    int tiles_n, unsigned* __restrict__ range_flag) {
    float rmax = 0.0f;
    constexpr int NT = WGM * WGN * 64;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int RPP = NT / 8;
    constexpr int LA = BM / RPP;            // float4 loads of A per thread per stage
    constexpr int BCH = BN * 4 / NT;        // 16-byte chunks per thread per stage, for each of W_hi / W_lo
    static_assert(BM % RPP == 0 && (BN * 4) % NT == 0 && WM % 32 == 0 && WN % 32 == 0, "tile shape");
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* const Ah = smem_raw;                    // [2][A_BYTES]
    char* const Al = Ah + 2 * A_BYTES;
    char* const Bh = Al + 2 * A_BYTES;            // [2][B_BYTES]
    char* const Bl = Bh + 2 * B_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / WGN, wc = wave % WGN;
    const long long tile_m = blockIdx.x / tiles_n;
    const int tile_n = blockIdx.x % tiles_n;
    const long long m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    const int lrow = tid >> 3;
    const int lc4 = tid & 7;
    const float* ap[LA];
    int a_st[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        long long m = m0 + lrow + RPP * i;
        m = m < M ? m : M - 1;
        ap[i] = A + (size_t)m * K + lc4 * 4;
        a_st[i] = swz64(lrow + RPP * i, lc4 >> 1) + (lc4 & 1) * 8;
    }
    const _Float16* bph[BCH];
    const _Float16* bpl[BCH];
    int b_st[BCH];
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
        const int id = tid + NT * i;
        const int row = id >> 2, slot = id & 3;
        bph[i] = Whi + (size_t)(n0 + row) * K + slot * 8;
        bpl[i] = Wlo + (size_t)(n0 + row) * K + slot * 8;
        b_st[i] = swz64(row, slot);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float4 ra[LA];
    uint4 rbh[BCH], rbl[BCH];

#define BD_F16_LOAD(KOFF)                                                                                \
    {                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < LA; ++i) ra[i] = *reinterpret_cast<const float4*>(ap[i] + (KOFF)); \
        _Pragma("unroll") for (int i = 0; i < BCH; ++i) {                                                \
            rbh[i] = *reinterpret_cast<const uint4*>(bph[i] + (KOFF));                                   \
            rbl[i] = *reinterpret_cast<const uint4*>(bpl[i] + (KOFF));                                   \
        }                                                                                                \
    }
#define BD_F16_STORE(BUF)                                                                                \
    {                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < LA; ++i) {                                                 \
            const float4 v = ra[i];                                                                      \
            rmax = range_of(rmax, v);                                                                    \
            f16x4 hi, lo;                                                                                \
            split_f16(v.x, v.y, v.z, v.w, hi, lo);                                                      \
            *reinterpret_cast<f16x4*>(Ah + (BUF) * A_BYTES + a_st[i]) = hi;                              \
            *reinterpret_cast<f16x4*>(Al + (BUF) * A_BYTES + a_st[i]) = lo;                              \
        }                                                                                                \
        _Pragma("unroll") for (int i = 0; i < BCH; ++i) {                                                \
            *reinterpret_cast<uint4*>(Bh + (BUF) * B_BYTES + b_st[i]) = rbh[i];                          \
            *reinterpret_cast<uint4*>(Bl + (BUF) * B_BYTES + b_st[i]) = rbl[i];                          \
        }                                                                                                \
    }

    const int frow = lane & 31;
    const int fh = lane >> 5;
#define BD_F16_COMPUTE(BUF)                                                                              \
    {                                                                                                    \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                  \
            f16x8 ah[TM], al[TM], bh[TN], bl[TN];                                                        \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                             \
                const int off = (BUF) * A_BYTES + swz64(wr * WM + i * 32 + frow, 2 * s + fh);            \
                ah[i] = *reinterpret_cast<const f16x8*>(Ah + off);                                       \
                al[i] = *reinterpret_cast<const f16x8*>(Al + off);                                       \
            }                                                                                            \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                             \
                const int off = (BUF) * B_BYTES + swz64(wc * WN + j * 32 + frow, 2 * s + fh);            \
                bh[j] = *reinterpret_cast<const f16x8*>(Bh + off);                                       \
                bl[j] = *reinterpret_cast<const f16x8*>(Bl + off);                                       \
            }                                                                                            \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) { \
                if constexpr (!PLAIN) {                                                                  \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al[i], acc[i][j], 0, 0, 0); \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah[i], acc[i][j], 0, 0, 0); \
                }                                                                                        \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);    \
            }                                                                                            \
        }                                                                                                \
    }

    BD_F16_LOAD(0)
    BD_F16_STORE(0)
    __syncthreads();
    const int nk = K / 32;
    for (int kt = 0; kt + 1 < nk; ++kt) {
        const int buf = kt & 1;
        BD_F16_LOAD((kt + 1) * 32)
        BD_F16_COMPUTE(buf)
        BD_F16_STORE(buf ^ 1)
        __syncthreads();
    }
    BD_F16_COMPUTE((nk - 1) & 1)
#undef BD_F16_LOAD
#undef BD_F16_STORE
#undef BD_F16_COMPUTE

    // transposed accumulators: lane owns row m0 + .. + (lane & 31), 16-byte stores (store_tile_t)
    const int half = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const long long m = m0 + wr * WM + i * 32 + frow;
        float* crow = C + (size_t)(m < M ? m : 0) * N;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nb = n0 + wc * WN + j * 32;
            store_tile_t(acc[i][j], unscale + nb, bias + nb, crow + nb, m < M, half);
        }
    }
    range_report(rmax, range_flag);
}

template <int BM, int BN, int WGM, int WGN, bool PLAIN>
void launch_pw16_(const float* A, const _Float16* Whi, const _Float16* Wlo, const float* unscale, const float* bias, float* C,
                  long long M, int N, int K, unsigned* range_flag, hipStream_t stream) {
    constexpr int NT = WGM * WGN * 64;
    constexpr size_t lds = 2u * 2u * (BM + BN) * 64;
    static std::once_flag lds_once[kMaxDevices];
    allow_dynamic_lds(&pointwise_f16x3_kernel<BM, BN, WGM, WGN, PLAIN>, (int)lds, lds_once);
    const int tiles_n = N / BN;
    const long long tiles = ((M + BM - 1) / BM) * tiles_n;
    hipLaunchKernelGGL((pointwise_f16x3_kernel<BM, BN, WGM, WGN, PLAIN>), dim3((unsigned)tiles), dim3(NT), lds, stream, A,
                       Whi, Wlo, unscale, bias, C, M, N, K, tiles_n, range_flag);
}

template <int BM, int BN, int WGM, int WGN>
void launch_pw16(const float* A, const _Float16* Whi, const _Float16* Wlo, const float* unscale, const float* bias, float* C,
                 long long M, int N, int K, bool plain, unsigned* range_flag, hipStream_t stream) {
    if (plain) launch_pw16_<BM, BN, WGM, WGN, true>(A, Whi, Wlo, unscale, bias, C, M, N, K, range_flag, stream);
    else launch_pw16_<BM, BN, WGM, WGN, false>(A, Whi, Wlo, unscale, bias, C, M, N, K, range_flag, stream);
}

template <int BM, int BN, int WGM, int WGN, int NH = 0, int NW = 0, int NS = 1>
void launch_pw(const float* A, const float* Wt, const float* bias, float* C, long long M, int N, int K,
               hipStream_t stream, const float* ndw_w = nullptr, const float* ndw_b = nullptr, int windows = 0) {
    constexpr int NT = WGM * WGN * 64;
    constexpr size_t lds = 2u * (BM + BN) * kLds * sizeof(float);
    static_assert(NH == 0 || lds >= (size_t)(BM + 1) * (BN + 4) * sizeof(float), "the f32 tile of the depthwise epilogue (+ a row of zeros) fits the stage buffers");
    static std::once_flag lds_once[kMaxDevices];
    allow_dynamic_lds(&pointwise_kernel<BM, BN, WGM, WGN, NH, NW, NS>, (int)lds, lds_once);
    const int tiles_n = N / BN;
    const long long tiles_m = (M + BM - 1) / BM;
    const long long tiles = tiles_m * tiles_n;
    // persistent: as many workgroups as fit the chip at once (by LDS: 160 KB per CU; by waves: 8 per SIMD), a multiple of 8
    int per_cu = (int)(160 * 1024 / lds);
    if (per_cu > 2048 / NT) per_cu = 2048 / NT;
    if (per_cu < 1) per_cu = 1;
    long long grid = (long long)cu_count() * per_cu / 8 * 8;
    if (grid > tiles) grid = tiles;
    if (grid < 1) grid = 1;                   // (a device with fewer than 8 resident workgroups: never a zero-sized launch)
    const int xcd_map = tiles_m % 8 == 0 && grid % 8 == 0;
    hipLaunchKernelGGL((pointwise_kernel<BM, BN, WGM, WGN, NH, NW, NS>), dim3((unsigned)grid), dim3(NT), lds, stream, A, Wt,
                       bias, C, M, N, K, tiles_n, tiles, xcd_map, ndw_w, ndw_b, windows);
}

// Workgroup (or persistent tile index) -> (row tile, column tile).  Workgroups go to the 8 XCDs round-robin by ID
// (measured: FETCH_SIZE of a K = N = 512 layer is 53.5 MiB per launch when its two column tiles are IDs b, b+1 /
// b+2 / b+4 apart and 32.6 MiB when they are 8, 16, 32 or 64 apart), each XCD has its own L2, and the column tiles
// of one row tile read the same input slab.  IDs b and b + 8 - same XCD, dispatched together - are therefore made
// the column tiles of one row tile, so the second read of the slab is an L2 hit instead of an HBM fetch.
__device__ __forceinline__ void tile_of(unsigned b, unsigned tiles_m, unsigned tn, unsigned& tile_m, unsigned& tile_n) {
    const unsigned full = tiles_m & ~7u;                         // row tiles covered by whole groups of 8
    if (tn > 1 && b < full * tn) {
        tile_n = (b >> 3) % tn;
        tile_m = (b / (8 * tn)) * 8 + (b & 7);
    } else {
        const unsigned r = tn > 1 ? b - full * tn : b;
        tile_m = (tn > 1 ? full : 0) + r / tn;
        tile_n = r % tn;
    }
}

// --------------------------------------------------------------------------- wave-specialised 96 x 256 tile kernel
// Round 1-2's fused separable kernel, reduced in round 6 to the two forms the tree still runs (every option of its tuning
// history - weights staged through LDS, register-staged slabs, 64-channel stages, 64-row tiles, band tiles, the pool
// epilogue, the clock trace - is in git history and DESIGN_HISTORY.md 4.3 / 4.4):
//   PWO = 1           pointwise only (the 1 x 1 convolution of a layer whose depthwise has been applied elsewhere): the default
//                     path's pointwise 13, and the wide layers of the one-kernel-per-op path
//   NDW = 1           depthwise inside the GEMM + the NEXT layer's stride-2 depthwise in the epilogue (layer 6 + depthwise 7
//                     behind bd_set_fusion separable = 7 / 10; whole windows per tile)
// A workgroup is 8 waves; waves 4-7 are PRODUCERS (the f32 input slab arrives by LDS-DMA into a ring of three, they run
// the depthwise on the VALU - or just split the slab - and write the split-f16 A tile of stage k + 1) and waves 0-3 are
// CONSUMERS (weight fragments straight from the fragment-ordered copy into registers, one stage ahead; the MFMAs of
// stage k).  Waves w and w + 4 share a SIMD; one barrier per 32-channel stage.  BM = 96 output positions (3 MFMA row
// tiles: 4 / 1 / 16 whole windows of the 6 x 4 / 12 x 8 / 3 x 2 maps), BN = 256 output channels.  Arithmetic order is that
// of the unfused kernels: bit-identical.
template <int NDW, int PWO, bool PLAIN>
__global__ __launch_bounds__(512, 2) void sep_ws_kernel(
    const float* __restrict__ X, const float* __restrict__ dw_w, const float* __restrict__ dw_b,
    const _Float16* __restrict__ Whi, const _Float16* __restrict__ Wlo, const float* __restrict__ pw_u,
    const float* __restrict__ pw_b,
    float* __restrict__ Cout, long long M, int N, int K, int H, int W, int tiles_n,
    const float* __restrict__ ndw_w, const float* __restrict__ ndw_b, float* __restrict__ out2,
    unsigned* __restrict__ range_flag) {
    float rmax = 0.0f;                        // largest |activation| this thread has split into f16 halves (producers)
    static_assert((NDW == 0 || NDW == 1) && (PWO == 0 || NDW == 0), "pointwise only, or a fused layer with the next depthwise");
    constexpr int BN = 256, XPMAX = 96, BM = 96;
    constexpr int NX = 3;                    // slab buffers: a ring of three, filled by DMA
    constexpr int WN = BN / 4;               // consumer wave tile: BM x WN
    constexpr int TM = BM / 32, TN = WN / 32;
    constexpr int LA = BM / 32;              // depthwise outputs (x4 channels) per producer thread per stage
    constexpr int XS_FLOATS = (XPMAX + 1) * 32;   // + the zero row
    constexpr int A_BYTES = BM * 64;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* const Xs = reinterpret_cast<float*>(smem_raw);             // [3][XS_FLOATS]
    char* const Ah = reinterpret_cast<char*>(Xs + NX * XS_FLOATS);     // [2][A_BYTES]
    char* const Al = Ah + 2 * A_BYTES;
    float* const Wall = reinterpret_cast<float*>(Al + 2 * A_BYTES);    // [10][K] depthwise taps + shift of all K channels

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned tile_m, tile_n;
    tile_of(blockIdx.x, (unsigned)((M + BM - 1) / BM), (unsigned)tiles_n, tile_m, tile_n);
    const long long m0 = (long long)tile_m * BM;
    const int n0 = (int)tile_n * BN;
    const int nk = K / 32;                    // stages; >= 4 (launcher)

    // input slab of this tile: rows [x_lo, x_lo + x_cnt) of X (whole windows, or a band of rows plus its halo rows)
    const int P = H * W;
    long long x_lo;
    int x_cnt;
    if (PWO || P < BM) {                      // no depthwise, or whole windows: the slab is the tile's own rows
        x_lo = m0;
        x_cnt = (int)((M - m0) < BM ? (M - m0) : BM);
    } else {
        // (32-bit arithmetic: the launcher guarantees M < 2^31, and 64-bit division is a ~1000-cycle routine)
        const unsigned m0u = (unsigned)m0;
        const long long n = m0u / (unsigned)P;
        const int oh_a = (int)(m0u % (unsigned)P) / W;
        const int oh_b = oh_a + BM / W;
        const int r0 = oh_a > 0 ? oh_a - 1 : 0;
        const int r1 = oh_b < H ? oh_b + 1 : H;
        x_lo = (n * H + r0) * W;
        x_cnt = (r1 - r0) * W;
    }
    if (wave >= 4) {
        // ================================================================= producers
        const int pt = tid - 256;
        const int lrow = pt >> 3, lc4 = pt & 7;
        int xt[(LA + 2) * 3];
        int a_st[LA];
        {
            // a thread owns LA vertically adjacent outputs (same column, rows oh0 .. oh0+LA-1) of 4 channels: the
            // 3 x 3 neighbourhoods overlap, so it reads (LA+2) x 3 slab values instead of LA x 9
            // (W and the groups per window G = P / LA are powers of two - checked by the launcher - so this index
            //  arithmetic is shifts; as divisions it was a visible part of the ~1900-cycle table set-up)
            const int slot = lrow;
            const int lw = 31 - __builtin_clz(W);
            int wl = 0, g = slot;
            if (P < BM) {
                const int lg = 31 - __builtin_clz(P / LA);
                wl = slot >> lg;
                g = slot & ((1 << lg) - 1);
            }
            const int og = g >> lw, ow = g & (W - 1);
            const int ml0 = wl * P + LA * og * W + ow;
            const int oh0 = (P >= BM ? (int)((unsigned)m0 % (unsigned)P) / W : 0) + LA * og;
            const int xc0 = (int)(m0 + ml0 - x_lo);
#pragma unroll
            for (int r = 0; r < LA + 2; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int ih = oh0 - 1 + r, iw = ow - 1 + c;
                    const bool ok = ih >= 0 && ih < H && iw >= 0 && iw < W;
                    xt[r * 3 + c] = (ok ? xc0 + (r - 1) * W + (c - 1) : XPMAX) * 32 + lc4 * 4;
                }
#pragma unroll
            for (int i = 0; i < LA; ++i) a_st[i] = swz64(ml0 + i * W, lc4 >> 1) + (lc4 & 1) * 8;
        }
        if (pt < 8 * NX) *reinterpret_cast<v4f*>(Xs + (pt >> 3) * XS_FLOATS + XPMAX * 32 + (pt & 7) * 4) = v4f{0.f, 0.f, 0.f, 0.f};

        int kch_ = 0;                         // first channel of the block the depthwise works on
        (void)kch_;
#define BD_P_DW(XB, AB)                                                                                   \
    {                                                                                                     \
        const float* xs_ = Xs + (XB) * XS_FLOATS;                                                         \
        const float* ws_ = Wall + kch_ + lc4 * 4;                                                         \
        v4f wt[9];                                                                                        \
        _Pragma("unroll") for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const v4f*>(ws_ + t * K); \
        const v4f bias4 = *reinterpret_cast<const v4f*>(ws_ + 9 * K);                                     \
        v4f xv[(LA + 2) * 3];                                                                             \
        _Pragma("unroll") for (int t = 0; t < (LA + 2) * 3; ++t)                                          \
            xv[t] = *reinterpret_cast<const v4f*>(xs_ + xt[t]);                                           \
        _Pragma("unroll") for (int i = 0; i < LA; ++i) {                                                  \
            v4f a4 = bias4;                                                                               \
            _Pragma("unroll") for (int t = 0; t < 9; ++t)                                                 \
                a4 = __builtin_elementwise_fma(xv[i * 3 + t], wt[t], a4);   /* v_pk_fma_f32: two IEEE fmas per issue */ \
            a4.x = fmaxf(a4.x, 0.0f); a4.y = fmaxf(a4.y, 0.0f); a4.z = fmaxf(a4.z, 0.0f); a4.w = fmaxf(a4.w, 0.0f); \
            rmax = range_of(rmax, a4);                                                                    \
            f16x4 hi, lo;                                                                                 \
            split_f16(a4.x, a4.y, a4.z, a4.w, hi, lo);                                                  \
            *reinterpret_cast<f16x4*>(Ah + (AB) * A_BYTES + a_st[i]) = hi;                                \
            *reinterpret_cast<f16x4*>(Al + (AB) * A_BYTES + a_st[i]) = lo;                                \
        }                                                                                                 \
    }
        {
            // ---- slabs and taps by LDS-DMA into a ring of three, three stages ahead.  One global_load_lds_dwordx4
            // moves 8 slab rows (lane l -> row l >> 3, 16-byte chunk l & 7; LDS address = base + 16 l, exactly the
            // [row][32] layout); the 4 producer waves take the 8-row groups round-robin.  No VGPRs, no ds_write, and -
            // the point - the slab has two full stages to arrive: the barrier waits with a COUNTED vmcnt (everything but
            // the newest stage's DMA), where __syncthreads() would drain to 0 and expose the ~3000-cycle memory latency.
            // Producers issue no other vector-memory operation, so the count is exact.
            constexpr int NG = XPMAX / 8, GPW = NG / 4, ND = GPW;
            const int pw = wave - 4;
            // LDS-DMA is serialised on M0 (the LDS base): a DMA to a new base waits for the previous one to finish,
            // ~300 cycles each.  So a wave takes GPW CONSECUTIVE 8-row groups and reaches them through the
            // instruction's immediate offset, which is added to both addresses - the global pointer is biased
            // back by the same amount - and M0 is written once per stage.
            const float* xsrc[GPW];
#pragma unroll
            for (int q = 0; q < GPW; ++q) {
                int row = 8 * (GPW * pw + q) + (lane >> 3);
                row = row < x_cnt ? row : x_cnt - 1;
                xsrc[q] = X + (size_t)(x_lo + row) * K + (lane & 7) * 4 - 256 * q;
            }
#define BD_X_DMA1(Q, KOFF, XB)                                                                            \
    if constexpr ((Q) < GPW)                                                                              \
        __builtin_amdgcn_global_load_lds(                                                                 \
            (const __attribute__((address_space(1))) void*)(xsrc[(Q) < GPW ? (Q) : 0] + (KOFF)),          \
            (__attribute__((address_space(3))) void*)(Xs + (XB) * XS_FLOATS + GPW * pw * 256), 16, 1024 * (Q), 0);
#define BD_X_DMA(KOFF, XB)                                                                                \
    {                                                                                                     \
        BD_X_DMA1(0, KOFF, XB)                                                                            \
        BD_X_DMA1(1, KOFF, XB)                                                                            \
        BD_X_DMA1(2, KOFF, XB)                                                                            \
        BD_X_DMA1(3, KOFF, XB)                                                                            \
    }
#define BD_P_SYNC(KEEP)                                                                                   \
    {                                                                                                     \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
        __builtin_amdgcn_s_barrier();                                                                     \
        asm volatile("" ::: "memory");                                                                    \
    }
            // PWO: the A tile is the slab itself, split into f16 hi + lo (rows lrow + 32 i, channels 4 lc4 ..)
#define BD_P_CVT(XB, AB)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < LA; ++i) {                                                      \
        const v4f a4 = *reinterpret_cast<const v4f*>(Xs + (XB) * XS_FLOATS + (lrow + 32 * i) * 32 + lc4 * 4); \
        rmax = range_of(rmax, a4);                                                                        \
        f16x4 hi, lo;                                                                                     \
        split_f16(a4.x, a4.y, a4.z, a4.w, hi, lo);                                                      \
        const int st_ = swz64(lrow + 32 * i, lc4 >> 1) + (lc4 & 1) * 8;                                   \
        *reinterpret_cast<f16x4*>(Ah + (AB) * A_BYTES + st_) = hi;                                        \
        *reinterpret_cast<f16x4*>(Al + (AB) * A_BYTES + st_) = lo;                                        \
    }
#define BD_P_WORK(XB, AB)                                                                                 \
    if constexpr (PWO) { BD_P_CVT(XB, AB) } else { BD_P_DW(XB, AB) }
            // first the three slab requests, then the taps + shift of all K channels ([10][K] floats, ordinary loads):
            // the compiler drains vmcnt before the first tap is written to LDS, which also covers the slabs - one
            // memory round trip for the whole prologue instead of two
            if constexpr (NDW == 1) {
                // the NEXT layer's taps + shift of this tile's BN columns for the epilogue, [10][BN] behind the f32 tile: ten 1 KB
                // rows by LDS-DMA, issued before the slabs so that the counted waits below cover them
                float* const Nw = reinterpret_cast<float*>(smem_raw + (size_t)BM * (BN + 4) * 4);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int t = pw + 4 * c;
                    if (t < 10) {
                        const float* src = (t < 9 ? ndw_w + (size_t)t * N : ndw_b) + n0 + 4 * lane;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                         (__attribute__((address_space(3))) void*)(Nw + t * BN), 16, 0, 0);
                    }
                }
            }
            BD_X_DMA(0, 0)
            BD_X_DMA(32, 1)
            BD_X_DMA(64, 2)
            if constexpr (!PWO) {
                // dw_w is [9][K] contiguous, dw_b [K]: as float4 items i < 9 K / 4 resp. the rest, Wall has the same
                // flat layout.  All loads are issued before the first write (a plain loop made five serial round trips).
                constexpr int TI = 10;            // items per thread at K = 1024
                const int n_w = 9 * (K / 4), n_all = 10 * (K / 4);
                v4f tw_[TI];
#pragma unroll
                for (int j = 0; j < TI; ++j) {
                    const int i = pt + 256 * j;
                    if (i < n_all) tw_[j] = *reinterpret_cast<const v4f*>(i < n_w ? dw_w + 4 * (size_t)i : dw_b + 4 * (size_t)(i - n_w));
                }
#pragma unroll
                for (int j = 0; j < TI; ++j) {
                    const int i = pt + 256 * j;
                    if (i < n_all) *reinterpret_cast<v4f*>(Wall + 4 * (size_t)i) = tw_[j];
                }
            }
            BD_P_SYNC(2 * ND)                 // slab 0 has landed, the taps are written
            BD_P_WORK(0, 0)
            BD_P_SYNC(ND)                     // A[0] written; slab 1 has landed
            int rs = 1;                       // ring slot of slab k+1
            int k = 0;
            for (; k + 3 < nk; ++k) {         // stage k: slab k+3 replaces slab k (consumed during stage k-1)
                const int r3 = rs == 0 ? 2 : rs - 1;
                BD_X_DMA((k + 3) * 32, r3)
                kch_ = (k + 1) * 32;
                BD_P_WORK(rs, (k + 1) & 1)
                BD_P_SYNC(ND)                 // slab k+2 has landed, slab k+3 stays in flight
                rs = rs == 2 ? 0 : rs + 1;
            }
            for (; k + 1 < nk; ++k) {         // the last two depthwise stages: nothing left to request
                kch_ = (k + 1) * 32;
                BD_P_WORK(rs, (k + 1) & 1)
                BD_P_SYNC(0)
                rs = rs == 2 ? 0 : rs + 1;
            }
            BD_P_SYNC(0)                      // consumers' last MFMA stage
#undef BD_X_DMA
#undef BD_X_DMA1
#undef BD_P_SYNC
#undef BD_P_WORK
#undef BD_P_CVT
        }
#undef BD_P_DW
    } else {
    // ===================================================================== consumers
    const int wc = wave;                      // column block of this wave
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int frow = lane & 31;
    const int fh = lane >> 5;
    {
        // The consumers do not stage the weights through LDS: with the 1 x 4 consumer layout every wave owns its own WN
        // output columns, so a weight fragment is used by exactly one wave: each lane loads its MFMA B fragments (16 bytes of
        // hi, 16 of lo per k-step and column tile) straight from global/L2 into a double-buffered register set, one stage
        // ahead.  Whi / Wlo are the fragment-order copies (SepLayer::pw_fhi / pw_flo): a load is one contiguous KiB per wave.
        // fragment pointers: column tile j -> row n0 + wc*WN + 32 j + frow of W^T, k offset 8 (2 s + fh)
        const _Float16* wph[TN];
        const _Float16* wpl[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const size_t frag = ((size_t)((n0 + wc * WN) / 32 + j) * (K / 16) * 64 + lane) * 8;
            wph[j] = Whi + frag;
            wpl[j] = Wlo + frag;
        }
        f16x8 b0h[TN][2], b0l[TN][2], b1h[TN][2], b1l[TN][2];    // fragments of an even / an odd stage
#define BD_W_LOAD(BH, BL, KOFF)                                                                           \
    {                                                                                                     \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) _Pragma("unroll") for (int s = 0; s < 2; ++s) {    \
            BH[j][s] = *reinterpret_cast<const f16x8*>(wph[j] + (KOFF) * 32 + 512 * s);                   \
            BL[j][s] = *reinterpret_cast<const f16x8*>(wpl[j] + (KOFF) * 32 + 512 * s);                   \
        }                                                                                                 \
    }
        // one stage = 2 TM steps (k16 step s x row tile i) of 3 TN MFMAs; the A fragments of step n+1 are requested
        // before the MFMAs of step n are issued (reading all TM pairs of a k16 step and then waiting exposed two
        // LDS latencies per stage)
#define BD_W_AFRAG(AH, AL, BUF, N)                                                                        \
    {                                                                                                     \
        const int off = (BUF) * A_BYTES + swz64(((N) % TM) * 32 + frow, 2 * ((N) / TM) + fh);             \
        AH = *reinterpret_cast<const f16x8*>(Ah + off);                                                   \
        AL = *reinterpret_cast<const f16x8*>(Al + off);                                                   \
    }
#define BD_W_MFMA(BUF, BH, BL)                                                                            \
    {                                                                                                     \
        f16x8 ahx[2], alx[2];                                                                             \
        BD_W_AFRAG(ahx[0], alx[0], BUF, 0)                                                                \
        _Pragma("unroll") for (int n = 0; n < 2 * TM; ++n) {                                              \
            if (n + 1 < 2 * TM) BD_W_AFRAG(ahx[(n + 1) & 1], alx[(n + 1) & 1], BUF, n + 1)                \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                              \
                if constexpr (!PLAIN) {                                                                   \
                    acc[n % TM][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alx[n & 1], BH[j][n / TM], acc[n % TM][j], 0, 0, 0); \
                    acc[n % TM][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahx[n & 1], BL[j][n / TM], acc[n % TM][j], 0, 0, 0); \
                }                                                                                         \
                acc[n % TM][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahx[n & 1], BH[j][n / TM], acc[n % TM][j], 0, 0, 0); \
            }                                                                                             \
        }                                                                                                 \
    }
        // stage t = 32 channels; its A block is ring slot t & 1, its fragments b0 (t even) / b1 (t odd)
        const int T = K / 32;                 // even, >= 4
        BD_W_LOAD(b0h, b0l, 0)
        BD_W_LOAD(b1h, b1l, 32)
        __syncthreads();
        __syncthreads();
        int t = 0;
        for (; t + 2 < T; t += 2) {
            BD_W_MFMA(t & 1, b0h, b0l)
            BD_W_LOAD(b0h, b0l, (t + 2) * 32)
            __syncthreads();
            BD_W_MFMA((t + 1) & 1, b1h, b1l)
            BD_W_LOAD(b1h, b1l, (t + 3) * 32)
            __syncthreads();
        }
        BD_W_MFMA(t & 1, b0h, b0l)
        __syncthreads();
        BD_W_MFMA((t + 1) & 1, b1h, b1l)
        __syncthreads();
#undef BD_W_AFRAG
#undef BD_W_LOAD
#undef BD_W_MFMA
    }

    // bias + ReLU into an f32 tile in LDS (every stage buffer is dead after the last barrier)
    float* const Ct = reinterpret_cast<float*>(smem_raw);          // [BM][BN + 4]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nl = wc * WN + j * 32 + frow;
        const float b = pw_b[n0 + nl], u = pw_u[n0 + nl];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = i * 32 + 4 * fh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = mb + (r & 3) + 8 * (r >> 2);
                Ct[ml * (BN + 4) + nl] = fmaxf(fmaf(acc[i][j][r], u, b), 0.0f);
            }
        }
    }
    }   // consumers

    __syncthreads();
    if constexpr (NDW == 1) {
        // ---- next layer's depthwise (stride 2) on the tile: windows are whole, so every tap is in LDS ----
        // A wave's 64 lanes are the 64 channel quads of ONE output position (8 waves x 3 positions = the tile's 24), so the
        // position, its padding tests and its row arithmetic are scalar; the two maps this runs on (12 x 8: layer 6, 6 x 4:
        // layer 12 on the test-hook path) are compile-time cases, so no division survives; taps and shift were brought to
        // LDS by the producers' prologue.  (Round 2's form - a position per thread with run-time divisions and nine divergent
        // padding branches - was a third of a layer-6 tile's time.)
        const float* Ct = reinterpret_cast<const float*>(smem_raw);
        constexpr int C4 = BN / 4, CTW = BN + 4;
        static_assert(C4 == 64, "a wave per output position");
        const float* Nw = reinterpret_cast<const float*>(smem_raw + (size_t)BM * (BN + 4) * 4);
        const int c4 = tid & 63;
        const int slot = __builtin_amdgcn_readfirstlane(tid >> 6);
        v4f wt[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const v4f*>(Nw + t * BN + c4 * 4);
        const v4f shift = *reinterpret_cast<const v4f*>(Nw + 9 * BN + c4 * 4);
        const unsigned m0u = (unsigned)m0;
        auto positions = [&](auto hh_, auto ww_) {
            constexpr int HH = decltype(hh_)::value, WW = decltype(ww_)::value, PP = HH * WW;
            constexpr int OW2 = WW / 2, P2 = (HH / 2) * OW2, NPOS = (BM / PP) * P2;
            static_assert(NPOS == 24, "three output positions per wave");
#pragma unroll
            for (int pp = 0; pp < NPOS; pp += 8) {
                const int ps = pp + slot;
                const int wl = ps / P2, pos2 = ps % P2;
                if (m0 + (long long)wl * PP >= M) continue;
                const int oh = pos2 / OW2, ow = pos2 % OW2;
                const float* base = Ct + (wl * PP + 2 * oh * WW + 2 * ow) * CTW + c4 * 4;
                v4f acc = shift;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        v4f v = {0.f, 0.f, 0.f, 0.f};        // SAME padding: 0 before, 1 after - a wave-uniform test
                        if (2 * oh + kh < HH && 2 * ow + kw < WW) v = *reinterpret_cast<const v4f*>(base + (kh * WW + kw) * CTW);
                        acc = __builtin_elementwise_fma(v, wt[kh * 3 + kw], acc);
                    }
                acc.x = fmaxf(acc.x, 0.0f);
                acc.y = fmaxf(acc.y, 0.0f);
                acc.z = fmaxf(acc.z, 0.0f);
                acc.w = fmaxf(acc.w, 0.0f);
                const long long row2 = (long long)(m0u / (unsigned)PP + wl) * P2 + pos2;
                *reinterpret_cast<v4f*>(out2 + (size_t)row2 * N + n0 + c4 * 4) = acc;
            }
        };
        if (H == 12) positions(std::integral_constant<int, 12>{}, std::integral_constant<int, 8>{});
        else positions(std::integral_constant<int, 6>{}, std::integral_constant<int, 4>{});
    } else {
        // ---- all 8 waves: tile -> HBM as whole rows, 16 bytes per lane ----
        const float* Ct = reinterpret_cast<const float*>(smem_raw);
        constexpr int C4 = BN / 4;                                   // float4 per tile row
#pragma unroll
        for (int it = 0; it < BM * C4 / 512; ++it) {
            const int id = tid + 512 * it;
            const int ml = id / C4, c4 = id % C4;
            const long long m = m0 + ml;
            if (m < M)
                *reinterpret_cast<v4f*>(Cout + (size_t)m * N + n0 + c4 * 4) =
                    *reinterpret_cast<const v4f*>(Ct + ml * (BN + 4) + c4 * 4);
        }
    }
    range_report(rmax, range_flag);
}

template <int NDW, int PWO, bool PLAIN = false>
void launch_sep_ws(const float* X, const SepLayer& L, float* out, long long M, hipStream_t stream,
                   const SepLayer* next = nullptr) {
    if constexpr (!PLAIN) {                   // mode 2: the same kernel with one MFMA per product
        if (L.pw_mode == 2) return launch_sep_ws<NDW, PWO, true>(X, L, out, M, stream, next);
    }
    constexpr int BN = 256, XPMAX = 96, BM = 96;
    constexpr size_t lds_pipe0 = 3u * (XPMAX + 1) * 128 + 2u * 2u * BM * 64;
    const size_t lds_pipe = lds_pipe0 + (!PWO ? (size_t)40 * L.cin : 0);   // + taps and shift of all input channels
    constexpr size_t lds_tile = (size_t)BM * (BN + 4) * 4 + (NDW == 1 ? 40u * BN : 0u);   // NDW = 1: + the next layer's taps and shift
    const size_t lds = lds_pipe > lds_tile ? lds_pipe : lds_tile;
    constexpr size_t lds_pipe_max = lds_pipe0 + (!PWO ? 40u * 1024u : 0u);          // the widest layer has 1024 input channels
    constexpr size_t lds_max = lds_pipe_max > lds_tile ? lds_pipe_max : lds_tile;
    static std::once_flag lds_once[kMaxDevices];
    allow_dynamic_lds(&sep_ws_kernel<NDW, PWO, PLAIN>, (int)lds_max, lds_once);
    const int tiles_n = L.cout / BN;
    const long long tiles = ((M + BM - 1) / BM) * tiles_n;
    hipLaunchKernelGGL((sep_ws_kernel<NDW, PWO, PLAIN>), dim3((unsigned)tiles), dim3(512), lds, stream, X, dw_w_of(L),
                       dw_b_of(L), static_cast<const _Float16*>(L.pw_fhi), static_cast<const _Float16*>(L.pw_flo), L.pw_u, L.pw_b,
                       out, M, L.cout, L.cin, L.h_out, L.w_out, tiles_n, next ? dw_w_of(*next) : nullptr,
                       next ? dw_b_of(*next) : nullptr, out, L.range_flag);
}

// --------------------------------------------------------------------------- pointwise with the weights in registers
// LDS operations the compiler must not reorder or wait for on its own: the kernel below counts them (lgkmcnt).
__device__ __forceinline__ unsigned pw_lds_addr(const void* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}
template <int OFFSET>
__device__ __forceinline__ f16x8 pw_lds_frag(unsigned addr) {
    f16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFFSET) : "memory");
    return v;
}
__device__ __forceinline__ void pw_lds_store64(unsigned addr, f16x4 v) {
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ f16x8 pw_landed(f16x8 v) {       // after the wait that covers the read: later uses stay behind it
    asm volatile("" : "+v"(v));
    return v;
}
template <int N>
__device__ __forceinline__ void pw_lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
template <int I, int E, typename F>
__device__ __forceinline__ void static_for_pw(F&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for_pw<I + 1, E>(f);
    }
}
// Issue order of a tile's LDS operations in pw_res_kernel: fragments of steps 0 and 1 (R reads each: hi, lo), then per
// step q the fragments of step q + 2 and, at V evenly spaced steps, the R stores of one split item; pending(q) is how
// many of them may still be in flight when the fragments of step q are needed.
template <int K16, int V, int R>
struct PwResSchedule {
    static constexpr int split_at(int q) {                      // item index whose stores follow the reads of step q, or -1
        for (int j = 0; j < V; ++j)
            if (q == j * K16 / (V > 0 ? V : 1) + 1) return j;
        return -1;
    }
    static constexpr int issued_at_step(int u) { return (u + 2 < K16 ? R : 0) + (split_at(u) >= 0 ? R : 0); }
    static constexpr int pending(int q) {
        int upto_wait = 2 * R;
        for (int u = 0; u <= q; ++u) upto_wait += issued_at_step(u);
        int through_read = q < 2 ? R * (q + 1) : 2 * R;
        if (q >= 2) {
            for (int u = 0; u < q - 2; ++u) through_read += issued_at_step(u);
            through_read += R;
        }
        return upto_wait - through_read;
    }
};

// The 1x1 convolutions of layers 5 (128 -> 256) and 7 (256 -> 512) have so few input channels that a wave's share of the
// split-f16 weights - 32 output columns x K, hi and lo - fits its register file: 64 VGPRs at K = 128, 128 at K = 256.
// A workgroup is 8 equal waves, wave w owning columns 32 w .. 32 w + 31 of a 256-column block; it is PERSISTENT
// (one per CU), loads its weight fragments once and then walks 32-row tiles of the input: all waves split the
// next tile into f16 hi + lo in LDS (rows requested three tiles ahead, straight into registers), every wave runs the
// 3 K / 16 MFMAs of its column tile on the current one and writes bias + ReLU from the accumulators (a store
// instruction covers two 128-byte row segments).  No weight traffic after the prologue, no pipeline fill per tile, one
// barrier per tile.  Same products in the same order as pointwise_f16x3_kernel: bit-identical.
template <int K16, bool PLAIN>
__global__ __launch_bounds__(512, 2) void pw_res_kernel(const float* __restrict__ X, const _Float16* __restrict__ Wfhi,
                                                         const _Float16* __restrict__ Wflo, const float* __restrict__ unscale,
                                                         const float* __restrict__ bias,
                                                         float* __restrict__ C, int M, int N, int tiles_n,
                                                         unsigned* __restrict__ range_flag) {
    constexpr int K = 16 * K16;
    constexpr int V = K16 / 4;                // float4 items per thread and tile: 32 rows x K / 4 over 512 threads
    constexpr int RP = 2048 / K;              // rows the 512 threads cover per item
    constexpr int ROWB = 2 * K;               // bytes of a row of one f16 half
    constexpr int HALF = 32 * ROWB;
    constexpr int R = PLAIN ? 1 : 2;          // LDS operations per fragment / per split item
    using S = PwResSchedule<K16, V, R>;
    // [2 buffers][hi, lo][32 rows][K] f16; the 16-byte chunks of a row are XOR-swizzled by the row number
    extern __shared__ __attribute__((aligned(1024))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fh = lane >> 5;

    // column block h and row stream r of this workgroup: IDs b and b + 8 (same XCD, dispatched together) are the
    // column blocks of the same row tiles, so the second read of a tile is an L2 hit
    const unsigned b = blockIdx.x, tn = (unsigned)tiles_n;
    const unsigned h = (b >> 3) % tn;
    const int r = (int)((b / (8 * tn)) * 8 + (b & 7));
    const int streams = (int)(gridDim.x / tn);
    const int n_tiles = (M + 31) >> 5;
    if (r >= n_tiles) return;
    const int ct = (int)h * 8 + wave;         // column tile of this wave

    // split items: thread -> (row r0 + RP j, channels 4 c4 ..) of a tile
    const int r0 = tid / (K / 4), c4 = tid % (K / 4);
    const float* const xcol = X + c4 * 4;
    const unsigned lds0 = pw_lds_addr(smem_raw);
    const unsigned st0 = lds0 + r0 * ROWB + (((c4 >> 1) ^ (r0 & 15)) << 4) + (c4 & 1) * 8;
    // fragment (row frow, k 16 q + 8 fh ..): chunk (2 q + fh) ^ (frow & 15), i.e. fr0 ^ (q << 5) as a byte address
    const unsigned fr0 = lds0 + frow * ROWB + ((fh ^ (frow & 15)) << 4);
    v4f rx[2][V];
    float rmax = 0.0f;
#define BD_R_LOAD(DST, TILE)                                                                              \
    _Pragma("unroll") for (int j = 0; j < V; ++j) {                                                       \
        int row_ = 32 * (TILE) + r0 + RP * j;                                                             \
        row_ = row_ < M ? row_ : M - 1;                                                                   \
        DST[j] = *reinterpret_cast<const v4f*>(xcol + (size_t)row_ * K);                                  \
    }
#define BD_R_SPLIT1(SRC, J, BUF)                                                                          \
    {                                                                                                     \
        const v4f a4 = SRC[J];                                                                            \
        rmax = range_of(rmax, a4);                                                                        \
        f16x4 hi, lo;                                                                                     \
        split_f16(a4.x, a4.y, a4.z, a4.w, hi, lo);                                                      \
        const unsigned st_ = (st0 ^ (((RP * (J)) & 15) << 4)) + RP * (J) * ROWB + (BUF) * 2 * HALF;       \
        pw_lds_store64(st_, hi);                                                                          \
        if constexpr (!PLAIN) pw_lds_store64(st_ + HALF, lo);                                             \
    }
    int t = r;
    BD_R_LOAD(rx[0], t)
    BD_R_LOAD(rx[1], t + streams)

    f16x8 bh[K16], bl[K16];
#pragma unroll
    for (int q = 0; q < K16; ++q) {
        const size_t frag = ((size_t)(ct * K16 + q) * 64 + lane) * 8;
        bh[q] = *reinterpret_cast<const f16x8*>(Wfhi + frag);
        if constexpr (!PLAIN) bl[q] = *reinterpret_cast<const f16x8*>(Wflo + frag);
    }
    const int col = ct * 32 + frow;
    const float bcol = bias[col], ucol = unscale[col];

#pragma unroll
    for (int j = 0; j < V; ++j) BD_R_SPLIT1(rx[0], j, 0)
    BD_R_LOAD(rx[0], t + 2 * streams)
    // the weights have to be in their registers HERE: left to the compiler, their waits land between the MFMAs of the
    // loop, where in steady state they wait for the previous tile's stores instead
#pragma unroll
    for (int q = 0; q < K16; ++q) {
        bh[q] = pw_landed(bh[q]);
        if constexpr (!PLAIN) bl[q] = pw_landed(bl[q]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    // during tile i register set i & 1 holds the rows of tile i + 2 and the other set those of tile i + 1, which are
    // split now and replaced by the request for tile i + 3; the loop is unrolled by two so that the sets are named statically
    auto tile = [&](auto pc) {
        constexpr int p = decltype(pc)::value;                 // = i & 1: LDS buffer of this tile
        constexpr int buf = p;
        // A fragments through a ring of three k-steps, requested two steps ahead: the wave's LDS operations complete in
        // order, so "the fragments of step q have landed" is a count of what was issued after them (PwResSchedule).
        // The rows of the next tile are split into the other buffer in V pieces placed between the MFMAs (past the last
        // tile they are clamped copies nobody reads); once the last piece is taken its registers take the request for
        // the rows two tiles ahead.
        const unsigned ab = fr0 + buf * 2 * HALF;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
        f16x8 fa[3][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            fa[q][0] = pw_lds_frag<0>(ab ^ (q << 5));
            if constexpr (!PLAIN) fa[q][1] = pw_lds_frag<HALF>(ab ^ (q << 5));
        }
        static_for_pw<0, K16>([&](auto qi) {
            constexpr int q = decltype(qi)::value;
            if constexpr (q + 2 < K16) {
                fa[(q + 2) % 3][0] = pw_lds_frag<0>(ab ^ ((q + 2) << 5));
                if constexpr (!PLAIN) fa[(q + 2) % 3][1] = pw_lds_frag<HALF>(ab ^ ((q + 2) << 5));
            }
            if constexpr (S::split_at(q) >= 0) {
                constexpr int j = S::split_at(q) >= 0 ? S::split_at(q) : 0;
                BD_R_SPLIT1(rx[p ^ 1], j, buf ^ 1)
            }
            pw_lds_wait<S::pending(q)>();
            const f16x8 ah = pw_landed(fa[q % 3][0]);
            if constexpr (!PLAIN) {
                const f16x8 al = pw_landed(fa[q % 3][1]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[q], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[q], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[q], acc, 0, 0, 0);
        });
        BD_R_LOAD(rx[p ^ 1], t + 3 * streams)      // set p ^ 1 held tile i + 1 (split above): now tile i + 3
        const int row0 = 32 * t + 4 * fh;
        float* const crow = C + (size_t)row0 * N + col;
        if (32 * t + 32 <= M) {
#pragma unroll
            for (int e = 0; e < 16; ++e) crow[(size_t)((e & 3) + 8 * (e >> 2)) * N] = fmaxf(fmaf(acc[e], ucol, bcol), 0.0f);
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (row0 + (e & 3) + 8 * (e >> 2) < M) crow[(size_t)((e & 3) + 8 * (e >> 2)) * N] = fmaxf(fmaf(acc[e], ucol, bcol), 0.0f);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        t += streams;
    };
    for (;;) {
        tile(std::integral_constant<int, 0>{});
        if (t >= n_tiles) break;
        tile(std::integral_constant<int, 1>{});
        if (t >= n_tiles) break;
    }
#undef BD_R_LOAD
#undef BD_R_SPLIT1
    range_report(rmax, range_flag);
}

template <int K16, bool PLAIN = false>
void launch_pw_res(const float* X, const SepLayer& L, float* out, int M, hipStream_t stream) {
    if constexpr (!PLAIN) {
        if (L.pw_mode == 2) return launch_pw_res<K16, true>(X, L, out, M, stream);
    }
    constexpr int lds = 2 * 2 * 32 * 32 * K16;
    static std::once_flag lds_once[kMaxDevices];
    allow_dynamic_lds(&pw_res_kernel<K16, PLAIN>, lds, lds_once);
    const int tiles_n = L.cout / 256;
    const int n_tiles = (M + 31) / 32;
    // one workgroup per CU (256 on MI355X), in whole groups of 8 row streams x tiles_n column blocks
    int streams = 256 / tiles_n;
    if (streams > n_tiles) streams = (n_tiles + 7) / 8 * 8;
    hipLaunchKernelGGL((pw_res_kernel<K16, PLAIN>), dim3((unsigned)(streams * tiles_n)), dim3(512), lds, stream, X,
                       static_cast<const _Float16*>(L.pw_fhi), static_cast<const _Float16*>(L.pw_flo), L.pw_u, L.pw_b, out, M,
                       L.cout, tiles_n, L.range_flag);
}

// --------------------------------------------------------------------------- layer 4 (+ depthwise 5), whole windows per workgroup
// Layer 4 is the widest map (24 x 16) with the fewest channels (128 -> 128): as 96-row tiles of the generic kernel it is
// four short stages per tile behind a full pipeline fill, and the band tiles that carry depthwise 5 in their epilogue
// compute every other row pair twice.  Here a workgroup is persistent (one per CU), owns WINDOWS b, b + G, .. and walks
// each top to bottom in twelve steps of two map rows (= one 32-row MFMA tile, all 128 input channels at once):
//   waves 0-3 (matrix side)  the split-f16 weights of their 32 output channels live in registers for the whole launch
//                            (64 VGPRs); per step they move two input rows global -> registers -> LDS ring (requested
//                            four steps ahead), run the 24 MFMAs of the row tile the vector side finished in the previous
//                            step, and - round 6 - run depthwise 5 (stride 2) on the tile where it lies, in the
//                            accumulators: lane (channel, half h) holds columns {0-3, 8-11} + 4 h of the tile's two map
//                            rows, i.e. everything four of the eight output columns need but the one column behind each
//                            group of four, which one v_permlane32_swap per group and row brings from the other half.
//                            One output row per step; its third input row arrives a step later - the partial sums wait in
//                            registers, the order of the nine FMAs is unchanged.  (Until round 5 the even-column vector
//                            waves did this from an f32 copy of the tile in LDS: the vector side was the kernel's bound
//                            - 2 550 of its cycles per step against 1 770 - and without that work the kernel takes 54
//                            instead of 69 us, gpurun_out/r06/abl_l4_nodw5.log.)
//   waves 4-11 (vector side) a thread owns four channels (its taps and shift stay in registers) and one map column: the
//                            3x3 depthwise of the step's two outputs from a register window of 4 x 3 inputs that slides
//                            down the map (6 LDS reads per step), split into the A tile of the next MFMA step.  Two vector
//                            waves and one matrix wave per SIMD.
// so nothing is computed twice, layer 4's own output never exists, and the input is read once.  The row tiles of a
// workgroup's windows form ONE stream (tile T = 12 i + s): step K runs the depthwise of tile K and the MFMAs + depthwise 5 of
// tile K - 1, so the pipeline fills once per launch, not once per window; the top and bottom rows of
// a window take zeros instead of their neighbours' rows.  One barrier per step.  Arithmetic order per element equals
// depthwise_kernel / pointwise_f16x3_kernel: bit-identical to the unfused path.
template <bool PLAIN>
__global__ __launch_bounds__(768) void l4_window_kernel(
    const float* __restrict__ X, const float* __restrict__ dw_w, const float* __restrict__ dw_b,
    const _Float16* __restrict__ Wfhi, const _Float16* __restrict__ Wflo, const float* __restrict__ pw_u,
    const float* __restrict__ pw_b,
    const float* __restrict__ ndw_w, const float* __restrict__ ndw_b, float* __restrict__ out, int windows,
    unsigned* __restrict__ range_flag) {
    constexpr int H = 24, W = 16, C = 128, K16 = 8, STEPS = H / 2;
    constexpr int COL_B = C * 4;                       // bytes of one map position, f32
    constexpr int ROW_B = (W + 1) * COL_B;             // ring slot of a map row: 16 columns + a zero column
    constexpr int RING0 = COL_B;                       // a zero column in front of slot 0 (column -1 of slot 0)
    constexpr int A0 = RING0 + 8 * ROW_B;              // A tile [2 buffers][hi, lo][32 rows][128 f16], chunks XOR-swizzled
    constexpr int A_HALF = 32 * 2 * C, A_BUF = 2 * A_HALF;
    constexpr int T5 = A0 + 2 * A_BUF;                 // depthwise 5's taps and shift per channel, [128][12] f32 (10 used): as registers
                                                       // of the matrix waves they would not fit beside the weights (168 per lane)
    constexpr int O5 = T5 + C * 12 * 4;                // finished depthwise-5 rows on their way out, [2 steps][2 rows][8][128] f32: the
                                                       // matrix waves wait for their input rows with a counted vmcnt, which a store
                                                       // of their own in between turns into vmcnt(0) - the vector waves store
    constexpr int O5_ROW = (W / 2) * C * 4, O5_BUF = 2 * O5_ROW;
    constexpr size_t WIN_IN = (size_t)H * W * C, WIN_OUT = (size_t)STEPS * (W / 2) * C;
    static_assert(A0 % 512 == 0, "fragment addresses are formed by XOR");
    static_assert((2 * STEPS) % 8 == 0 && STEPS % 2 == 0, "ring slots and buffer parities carry over from window to window");
    extern __shared__ __attribute__((aligned(1024))) char smem_raw[];
    char* const smem = smem_raw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x, G = gridDim.x;
    if (b >= windows) return;
    const int NT = STEPS * ((windows - b + G - 1) / G);          // row tiles of this workgroup

    // the zero columns: in front of the ring and column 16 of every ring slot
    for (int i = tid; i < 9 * 32; i += 768) {
        const int z = i >> 5;
        *reinterpret_cast<v4f*>(smem + (z == 0 ? 0 : RING0 + (z - 1) * ROW_B + W * COL_B) + (i & 31) * 16) = v4f{0.f, 0.f, 0.f, 0.f};
    }
    for (int i = tid; i < 10 * C; i += 768) {          // taps t = 0 .. 8 and the shift (t = 9) of channel c at [c][t]
        const int t = i / C, c = i - t * C;
        reinterpret_cast<float*>(smem + T5)[c * 12 + t] = t < 9 ? ndw_w[t * C + c] : ndw_b[c];
    }
    float rmax = 0.0f;

    if (wave < 4) {
        // ================================================================= matrix side
        const int frow = lane & 31, fh = lane >> 5;
        const int c4 = tid & 31, col_lo = tid >> 5;    // slab items: 16-byte chunk c4 of columns col_lo and col_lo + 8
        const float* const xt = X + (size_t)b * WIN_IN + (size_t)col_lo * C + c4 * 4;
        char* const ring_t = smem + RING0 + col_lo * COL_B + c4 * 16;
        v4f rs[2][4];                                  // two row pairs in flight
        // row pair J of the stream = rows 2 j, 2 j + 1 of the workgroup's window i (J = 12 i + j), ring slots (2 J + row) & 7
#define BD_L4_LOAD(DST, J)                                                                                \
    {                                                                                                     \
        const int i_ = (J) / STEPS, j_ = (J) - i_ * STEPS;                                                \
        const float* const src_ = xt + (size_t)i_ * G * WIN_IN + (size_t)j_ * (2 * W * C);                \
        _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                     \
            DST[u] = *reinterpret_cast<const v4f*>(src_ + ((u >> 1) * W + 8 * (u & 1)) * C);              \
    }
#define BD_L4_STORE(SRC, J)                                                                               \
    _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                         \
        *reinterpret_cast<v4f*>(ring_t + ((2 * (J) + (u >> 1)) & 7) * ROW_B + 8 * (u & 1) * COL_B) = SRC[u];
        BD_L4_LOAD(rs[0], 0)
        BD_L4_LOAD(rs[1], 1)
        f16x8 bh[K16], bl[K16];
#pragma unroll
        for (int q = 0; q < K16; ++q) {
            const size_t frag = ((size_t)(wave * K16 + q) * 64 + lane) * 8;
            bh[q] = *reinterpret_cast<const f16x8*>(Wfhi + frag);
            if constexpr (!PLAIN) bl[q] = *reinterpret_cast<const f16x8*>(Wflo + frag);
        }
        const int ncol = 32 * wave + frow;
        const float bcol = pw_b[ncol], ucol = pw_u[ncol];
        // depthwise 5 of this lane's channel: the four partial sums (output columns 2 fh, 2 fh + 1, 4 + 2 fh, 5 + 2 fh); its taps
        // and shift are read from LDS when a step needs them
        const float* const t5 = reinterpret_cast<const float*>(smem + T5) + ncol * 12;
        float acc5[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        float* const o5 = reinterpret_cast<float*>(smem + O5) + (2 * fh) * C + ncol;
        int s5 = 0;                                    // row tile within its window, of the tile the MFMAs work on
        BD_L4_STORE(rs[0], 0)
        BD_L4_STORE(rs[1], 1)
        BD_L4_LOAD(rs[0], 2)
        BD_L4_LOAD(rs[1], 3)
#pragma unroll
        for (int q = 0; q < K16; ++q) {                // the weights are in their registers before the loop (see pw_res_kernel)
            bh[q] = pw_landed(bh[q]);
            if constexpr (!PLAIN) bl[q] = pw_landed(bl[q]);
        }
        // fragment (row frow, k 16 q + 8 fh ..) sits in chunk (2 q + fh) ^ (frow & 15) of its row: fr0 ^ (q << 5)
        const unsigned fr0 = pw_lds_addr(smem) + (unsigned)(A0 + frow * 2 * C + ((fh ^ (frow & 15)) << 4));
        // accumulator element e is tile row (e & 3) + 8 (e >> 2) + 4 fh: map row m >> 4, column m & 15
        __syncthreads();
        auto step = [&](auto pc, int k) {
            constexpr int p = decltype(pc)::value;     // k & 1
            f32x16 acc;
            if (k >= 1 && k <= NT) {
                // row tile k - 1: A tile buffer (k - 1) & 1 = p ^ 1.  A fragments through a ring of three k-steps, requested
                // two steps ahead, with counted waits (pw_res_kernel)
                const unsigned ab = fr0 + (unsigned)((p ^ 1) * A_BUF);
                using S = PwResSchedule<K16, 0, PLAIN ? 1 : 2>;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
                f16x8 fa[3][2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    fa[q][0] = pw_lds_frag<0>(ab ^ (q << 5));
                    if constexpr (!PLAIN) fa[q][1] = pw_lds_frag<A_HALF>(ab ^ (q << 5));
                }
                static_for_pw<0, K16>([&](auto qi) {
                    constexpr int q = decltype(qi)::value;
                    if constexpr (q + 2 < K16) {
                        fa[(q + 2) % 3][0] = pw_lds_frag<0>(ab ^ ((q + 2) << 5));
                        if constexpr (!PLAIN) fa[(q + 2) % 3][1] = pw_lds_frag<A_HALF>(ab ^ ((q + 2) << 5));
                    }
                    pw_lds_wait<S::pending(q)>();
                    const f16x8 ah = pw_landed(fa[q % 3][0]);
                    if constexpr (!PLAIN) {
                        const f16x8 al = pw_landed(fa[q % 3][1]);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[q], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[q], acc, 0, 0, 0);
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[q], acc, 0, 0, 0);
                });
            }
            // the input rows two steps ahead into the ring, a request for those four steps ahead (under the last MFMAs)
            if (k + 2 < NT) { BD_L4_STORE(rs[p], k + 2) }
            if (k + 4 < NT) BD_L4_LOAD(rs[p], k + 4)
            if (k >= 1 && k <= NT) {
                // ---- bias + ReLU, then depthwise 5 on the tile's two map rows 2 s5, 2 s5 + 1: y[r][0 .. 7] = this lane's columns
                // {0-3, 8-11} + 4 fh of row r; nb[r][g] = the column behind group g (column 4 / 12 for half 0: the other half's
                // first of that group; column 8 / 16 for half 1: half 0's first of its second group / the zero padding)
                float y[2][8], nb[2][2];
                const v4f t5a = *reinterpret_cast<const v4f*>(t5), t5b = *reinterpret_cast<const v4f*>(t5 + 4), t5c = *reinterpret_cast<const v4f*>(t5 + 8);
                const float w5[9] = {t5a.x, t5a.y, t5a.z, t5a.w, t5b.x, t5b.y, t5b.z, t5b.w, t5c.x};
                const float b5 = t5c.y;
#pragma unroll
                for (int e = 0; e < 16; ++e) y[e >> 3][e & 7] = fmaxf(fmaf(acc[e], ucol, bcol), 0.0f);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    // v_permlane32_swap vdst, src trades lanes 32-63 of vdst against lanes 0-31 of src: with vdst = own column 0
                    // of the group pair and src = own first column of the second group, half 1 finds half 0's column 8 in vdst
                    // and half 0 finds half 1's column 4 in src; a second swap brings half 1's column 12 to half 0
                    float va = y[r][0], wa = y[r][4], vb = y[r][4], wb = 0.0f;
                    asm("" : "+v"(va), "+v"(wa), "+v"(vb), "+v"(wb));
                    const auto s1 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, va), __builtin_bit_cast(unsigned, wa), false, false);
                    const auto s2 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, vb), __builtin_bit_cast(unsigned, wb), false, false);
                    nb[r][0] = __builtin_bit_cast(float, fh ? (unsigned)s1[0] : (unsigned)s1[1]);
                    nb[r][1] = fh ? 0.0f : __builtin_bit_cast(float, (unsigned)s2[1]);
                }
                // output t of this lane (t = 0, 1: group 0, t = 2, 3: group 1) reads, per input row, in[t][0 .. 2]
#define BD_L4_IN(R, T, KW) ((T) == 0 ? y[R][KW] : (T) == 1 ? ((KW) < 2 ? y[R][2 + (KW)] : nb[R][0]) : (T) == 2 ? y[R][4 + (KW)] : ((KW) < 2 ? y[R][6 + (KW)] : nb[R][1]))
                float* const orow = o5 + (p ^ 1) * (O5_BUF / 4);        // (tile k - 1: buffer of its parity; row slot 0, the last tile's second row slot 1)
                if (s5 > 0) {                          // finishes output row s5 - 1: its kh = 2 row is map row 2 s5
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) acc5[t] = fmaf(BD_L4_IN(0, t, kw), w5[6 + kw], acc5[t]);
                        orow[((t >> 1) * 4 + (t & 1)) * C] = fmaxf(acc5[t], 0.0f);
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {          // starts output row s5: kh = 0, 1
                    acc5[t] = b5;
#pragma unroll
                    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) acc5[t] = fmaf(BD_L4_IN(kh, t, kw), w5[3 * kh + kw], acc5[t]);
                }
                if (s5 == STEPS - 1) {                 // map row 24 is the zero padding (multiplied, as depthwise_kernel does)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) acc5[t] = fmaf(0.0f, w5[6 + kw], acc5[t]);
                        orow[((W / 2) + (t >> 1) * 4 + (t & 1)) * C] = fmaxf(acc5[t], 0.0f);
                    }
                    s5 = 0;
                } else {
                    ++s5;
                }
#undef BD_L4_IN
            }
            __syncthreads();
        };
        for (int k = 0; k < NT + 2; k += 2) {          // (steps 0 .. NT + 1: NT is even; the last one is idle on both sides)
            step(std::integral_constant<int, 0>{}, k);
            step(std::integral_constant<int, 1>{}, k + 1);
        }
#undef BD_L4_LOAD
#undef BD_L4_STORE
    } else {
        // ================================================================= vector side
        // wave v = 0..7, half-wave hi: channels 4 c4 .. of ONE map column - the even columns in waves 0-3, the odd ones in waves 4-7
        const int v = wave - 4, c4 = lane & 31;
        const int col = v < 4 ? 2 * (2 * v + (lane >> 5)) : 2 * (2 * (v - 4) + (lane >> 5)) + 1;
        v4f w4[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) w4[t] = *reinterpret_cast<const v4f*>(dw_w + t * C + c4 * 4);
        const v4f b4 = *reinterpret_cast<const v4f*>(dw_b + c4 * 4);
        // input row r, columns col - 1 .. col + 1 (column -1 is the zero column in front, column 16 the one behind)
        const char* const xin = smem + RING0 + (col - 1) * COL_B + c4 * 16;
        // A tile: row m = 16 rr + col, channels 4 c4 ..: 8 bytes of chunk c4 >> 1
        int a_st[2];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int m = 16 * rr + col;
            a_st[rr] = A0 + m * 2 * C + (((c4 >> 1) ^ (m & 15)) << 4) + (c4 & 1) * 8;
        }
        v4f xr[4][3];                                  // input rows 2 s - 1 .. 2 s + 2 at [(2 p + i) & 3], three columns
        int s4 = 0;                                    // row tile within its window
        // waves v < 4 also carry the finished depthwise-5 rows from LDS to global memory, 16 bytes per lane: step k stores what the
        // matrix side finished in step k - 1 with tile k - 2 (row s5 - 1 of window i5; behind a window's last tile also row 11)
        const int o_lane = (v & 3) * 64 + lane;        // float4 index within a row of [8][128] f32
        float* const ot = out + (size_t)b * WIN_OUT + (size_t)o_lane * 4;
        int s5 = 0, i5 = 0;
        __syncthreads();
        auto step = [&](auto pc, int k) {
            constexpr int p = decltype(pc)::value;     // k & 1
            if (v < 4 && k >= 2) {
                const char* const ob = smem + O5 + p * O5_BUF + o_lane * 16;          // (tile k - 2: buffer of its parity)
                float* const orow = ot + (size_t)i5 * G * WIN_OUT;
                if (s5 > 0) *reinterpret_cast<v4f*>(orow + (size_t)(s5 - 1) * (W / 2) * C) = *reinterpret_cast<const v4f*>(ob);
                if (s5 == STEPS - 1) {
                    *reinterpret_cast<v4f*>(orow + (size_t)s5 * (W / 2) * C) = *reinterpret_cast<const v4f*>(ob + O5_ROW);
                    s5 = 0;
                    ++i5;
                } else {
                    ++s5;
                }
            }
            if (k < NT) {
                // ---- depthwise 4 of map rows 2 s4, 2 s4 + 1 -> A tile buffer p (ring slots continue across windows)
                if (s4 == 0) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        xr[0][c] = v4f{0.f, 0.f, 0.f, 0.f};
                        xr[1][c] = *reinterpret_cast<const v4f*>(xin + c * COL_B);
                    }
                }
                {
                    const char* const r2 = xin + ((2 * k + 1) & 7) * ROW_B;
#pragma unroll
                    for (int c = 0; c < 3; ++c) xr[(2 * p + 2) & 3][c] = *reinterpret_cast<const v4f*>(r2 + c * COL_B);
                }
                if (s4 + 1 < STEPS) {
                    const char* const r3 = xin + ((2 * k + 2) & 7) * ROW_B;
#pragma unroll
                    for (int c = 0; c < 3; ++c) xr[(2 * p + 3) & 3][c] = *reinterpret_cast<const v4f*>(r3 + c * COL_B);
                } else {
#pragma unroll
                    for (int c = 0; c < 3; ++c) xr[(2 * p + 3) & 3][c] = v4f{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    v4f a4 = b4;
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw)
                            a4 = __builtin_elementwise_fma(xr[(2 * p + rr + kh) & 3][kw], w4[kh * 3 + kw], a4);
                    a4.x = fmaxf(a4.x, 0.0f); a4.y = fmaxf(a4.y, 0.0f); a4.z = fmaxf(a4.z, 0.0f); a4.w = fmaxf(a4.w, 0.0f);
                    rmax = range_of(rmax, a4);
                    f16x4 hi, lo;
                    split_f16(a4.x, a4.y, a4.z, a4.w, hi, lo);
                    *reinterpret_cast<f16x4*>(smem + a_st[rr] + p * A_BUF) = hi;
                    if constexpr (!PLAIN) *reinterpret_cast<f16x4*>(smem + a_st[rr] + p * A_BUF + A_HALF) = lo;
                }
                s4 = s4 + 1 == STEPS ? 0 : s4 + 1;
            }
            __syncthreads();
        };
        for (int k = 0; k < NT + 2; k += 2) {
            step(std::integral_constant<int, 0>{}, k);
            step(std::integral_constant<int, 1>{}, k + 1);
        }
    }
    range_report(rmax, range_flag);
}

template <bool PLAIN = false>
void launch_l4_window(const float* X, const SepLayer& L, const SepLayer& next, float* out, int windows, hipStream_t stream) {
    if constexpr (!PLAIN) {
        if (L.pw_mode == 2) return launch_l4_window<true>(X, L, next, out, windows, stream);
    }
    constexpr int lds = 512 + 8 * 17 * 512 + 2 * 2 * 32 * 256 + 128 * 12 * 4 + 2 * 2 * 8 * 128 * 4;
    static std::once_flag lds_once[kMaxDevices];
    allow_dynamic_lds(&l4_window_kernel<PLAIN>, lds, lds_once);
    const int grid = windows < 256 ? windows : 256;           // one persistent workgroup per CU
    hipLaunchKernelGGL((l4_window_kernel<PLAIN>), dim3((unsigned)grid), dim3(768), lds, stream, X, dw_w_of(L), dw_b_of(L),
                       static_cast<const _Float16*>(L.pw_fhi), static_cast<const _Float16*>(L.pw_flo), L.pw_u, L.pw_b, dw_w_of(next),
                       dw_b_of(next), out, windows, L.range_flag);
}

// --------------------------------------------------------------------------- fused stem + layer-3 depthwise
// Layers 1-2 (conv 3x3 s2 -> depthwise 3x3 -> pointwise 32 -> 64) and the stride-2 depthwise of layer 3 (yamnet.py:77-80)
// in one kernel, in the arithmetic order of conv1_kernel, depthwise_kernel and the split-f16 pointwise kernel: the layer-2 output
// (the largest tensor of the network, 402 MB per 1024 windows) is never written.  A workgroup owns TWO
// output rows of layer 3's depthwise in one window; they need five layer-2 rows (one is shared with the
// neighbouring workgroup and computed twice), which need seven conv1 rows and fifteen log-mel rows.
//   A  log-mel band -> LDS                       B  conv1 band (7 rows)  -> LDS
//   C  depthwise 2 (5 rows) -> split-f16 A tile   D  [160][32] x [32][64] on the matrix cores
//   E  bias + ReLU -> f32 tile P[160][64] in LDS (rows past the map's edge are the zero padding)
//   F  depthwise 3 (stride 2, SAME = pad 0 before / 1 after) on P -> split-f16 A tile [32][64] in LDS: neither the layer-2
//   output nor the layer-3 depthwise output (100 MB per 1024 windows) touch HBM   G  [32][64] x [64][128] on the matrix cores (wave w:
//   columns 32 w .. 32 w + 31, weights as register fragments from the fragment-order copy)   H  bias + ReLU -> HBM
// Arithmetic order per element equals conv1_kernel / depthwise_kernel / pointwise_f16x3_kernel.
template <bool PLAIN>
__global__ __launch_bounds__(256, 3) void stem3_kernel(const float* __restrict__ logmel, int patch_step,
                                                    const WindowMap map, int w0,
                                                    const float* __restrict__ c1_w, const float* __restrict__ c1_b,
                                                    const float* __restrict__ dw2_w, const float* __restrict__ dw2_b,
                                                    const _Float16* __restrict__ Whi, const _Float16* __restrict__ Wlo,
                                                    const float* __restrict__ pw_u, const float* __restrict__ pw_b,
                                                    const float* __restrict__ dw3_w,
                                                    const float* __restrict__ dw3_b, float* __restrict__ out,
                                                    const _Float16* __restrict__ W3fhi, const _Float16* __restrict__ W3flo,
                                                    const float* __restrict__ pw3_u, const float* __restrict__ pw3_b,
                                                    unsigned* __restrict__ range_flag) {
    float rmax = 0.0f;
    constexpr int R2 = 5;                       // layer-2 rows in the tile
    constexpr int C1R = R2 + 2;                 // conv1 rows incl. halo: 7
    constexpr int LMR = 2 * C1R + 1;            // log-mel rows: 15
    constexpr int BM = R2 * 32;                 // 160 GEMM rows
    constexpr int PW = 68;                      // padded row of the f32 output tile
    // LDS carve-up (50 944 B -> three workgroups per CU): the log-mel band is dead once the conv band exists,
    // so it shares the A tile's bytes; the f32 output tile P overlays everything from phase E on.
    constexpr int OFF_C1 = 0;
    constexpr int OFF_AH = OFF_C1 + C1R * 34 * 32 * 4;              // 30464
    constexpr int OFF_AL = OFF_AH + BM * 64;                        // 40704
    constexpr int P_BYTES = BM * PW * 4;                            // 43520
    constexpr int OFF_A3H = P_BYTES;                                // PW3: layer-3 A tile, [2 halves of 32 k][32 rows][64 B]
    constexpr int OFF_A3L = OFF_A3H + 2 * 32 * 64;
    constexpr int LDS_BYTES = OFF_A3L + 2 * 32 * 64;               // 51712; P aliases from 0
    static_assert(OFF_AL + BM * 64 <= LDS_BYTES, "pipeline buffers must fit");
    static_assert(LMR * 68 * 4 <= 2 * BM * 64, "log-mel band must fit in the A tile it aliases");
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    float (*s_lm)[68] = reinterpret_cast<float (*)[68]>(smem + OFF_AH);
    float (*s_c1)[34][32] = reinterpret_cast<float (*)[34][32]>(smem + OFF_C1);
    char* const s_ah = smem + OFF_AH;
    char* const s_al = smem + OFF_AL;
    float* const P = reinterpret_cast<float*>(smem);               // [BM][PW], valid from phase E on

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int win = blockIdx.y;
    const int ob = blockIdx.x;                  // 0..11: depthwise-3 rows 2 ob, 2 ob + 1
    const int r0 = 4 * ob;                      // first layer-2 row of the tile
    const float* patch = logmel + window_frame(map, w0 + win, patch_step) * BD_MEL_BANDS;

    // this lane's pointwise weight fragments (phase D)
    f16x8 wbh[2], wbl[2];
    {
        const int wrow = (wave & 1) * 32 + (lane & 31);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int koff = (2 * s2 + (lane >> 5)) * 8;
            wbh[s2] = *reinterpret_cast<const f16x8*>(Whi + wrow * 32 + koff);
            wbl[s2] = *reinterpret_cast<const f16x8*>(Wlo + wrow * 32 + koff);
        }
    }


    // every phase's weights are requested one phase ahead (a phase used to begin with a global round trip)
    const int c4 = tid & 7;
    const int col = tid >> 3;
    v4f c1wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) c1wt[t] = *reinterpret_cast<const v4f*>(c1_w + t * 32 + c4 * 4);
    const v4f c1bias = *reinterpret_cast<const v4f*>(c1_b + c4 * 4);
    // ---- A: log-mel rows 2 (r0 - 1) .. +14, zero halo columns of the conv1 band ----
    for (int i = tid; i < LMR * 17; i += 256) {
        const int j = i / 17, q = i % 17;
        const int ih = 2 * r0 - 2 + j;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < 16 && ih >= 0 && ih < BD_PATCH_FRAMES) v = reinterpret_cast<const float4*>(patch + ih * BD_MEL_BANDS)[q];
        *reinterpret_cast<float4*>(&s_lm[j][q * 4]) = v;
    }
    for (int i = tid; i < C1R * 2 * 8; i += 256) {
        const int r = i / 16, side = (i >> 3) & 1, c4 = i & 7;
        *reinterpret_cast<float4*>(&s_c1[r][side ? 33 : 0][c4 * 4]) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();

    // ---- B: conv1 rows r0 - 1 .. r0 + 5 ----
    // (consecutive conv1 rows share a log-mel row: a rolling window reads 45 values instead of 63; the four
    //  channels of a tap are two packed fmas)
    v4f d2wt[9];                                // depthwise-2 taps: in flight during the conv1 phase
#pragma unroll
    for (int t = 0; t < 9; ++t) d2wt[t] = *reinterpret_cast<const v4f*>(dw2_w + t * 32 + c4 * 4);
    const v4f d2bias = *reinterpret_cast<const v4f*>(dw2_b + c4 * 4);
    {
        const v4f (&wt)[9] = c1wt;
        const v4f bias = c1bias;
        // a tap row past the patch (log-mel row 96: SAME padding) is skipped, as conv1_kernel does; only the
        // last row block of a window can meet one, so the check lives in its own copy of the loop
        // (as a per-tap condition the compiler turns it into 252 selects)
#define BD_STEM3_CONV1(CHECK)                                                                             \
    {                                                                                                     \
        float lm[3][3];                                                                                   \
        const v4f zero4 = {0.f, 0.f, 0.f, 0.f};                                                           \
        _Pragma("unroll") for (int kw = 0; kw < 3; ++kw) lm[0][kw] = s_lm[0][2 * col + kw];               \
        _Pragma("unroll") for (int i = 0; i < C1R; ++i) {                                                 \
            const int c1r = r0 - 1 + i;                                                                   \
            _Pragma("unroll") for (int kh = 1; kh < 3; ++kh)                                              \
                _Pragma("unroll") for (int kw = 0; kw < 3; ++kw) lm[kh][kw] = s_lm[2 * i + kh][2 * col + kw]; \
            if (!(CHECK) || (c1r >= 0 && c1r < 48)) {   /* the same for the whole workgroup: a scalar branch */ \
                v4f acc = bias;                                                                           \
                _Pragma("unroll") for (int kh = 0; kh < 3; ++kh) {                                        \
                    if (CHECK && 2 * c1r + kh >= BD_PATCH_FRAMES) continue;                               \
                    _Pragma("unroll") for (int kw = 0; kw < 3; ++kw) {                                    \
                        const float v = lm[kh][kw];                                                       \
                        acc = __builtin_elementwise_fma(v4f{v, v, v, v}, wt[kh * 3 + kw], acc);           \
                    }                                                                                     \
                }                                                                                         \
                v4f r4;                                                                                   \
                r4.x = fmaxf(acc.x, 0.0f);                                                                \
                r4.y = fmaxf(acc.y, 0.0f);                                                                \
                r4.z = fmaxf(acc.z, 0.0f);                                                                \
                r4.w = fmaxf(acc.w, 0.0f);                                                                \
                *reinterpret_cast<v4f*>(&s_c1[i][col + 1][c4 * 4]) = r4;                                  \
            } else {                             /* a row above or below the map: the depthwise's zero padding */ \
                *reinterpret_cast<v4f*>(&s_c1[i][col + 1][c4 * 4]) = zero4;                               \
            }                                                                                             \
            _Pragma("unroll") for (int kw = 0; kw < 3; ++kw) lm[0][kw] = lm[2][kw];                       \
        }                                                                                                 \
    }
        // (only the first and the last row block of a window have conv1 rows outside the map or tap rows outside the patch)
        if (ob == 0 || 2 * (r0 + C1R - 2) + 2 >= BD_PATCH_FRAMES || r0 + C1R - 2 >= 48) BD_STEM3_CONV1(true)
        else BD_STEM3_CONV1(false)
#undef BD_STEM3_CONV1
    }
    __syncthreads();

    // ---- C: depthwise 2 for rows r0 .. r0 + 4 -> split-f16 A tile [160][32] ----
    // (rolling window over the conv1 band: 21 LDS reads instead of 45)
    {
        const v4f (&wt)[9] = d2wt;
        const v4f bias = d2bias;
        v4f cv[3][3];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) cv[kh][kw] = *reinterpret_cast<const v4f*>(&s_c1[kh][col + kw][c4 * 4]);
#pragma unroll
        for (int r = 0; r < R2; ++r) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) cv[2][kw] = *reinterpret_cast<const v4f*>(&s_c1[r + 2][col + kw][c4 * 4]);
            v4f acc = bias;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) acc = __builtin_elementwise_fma(cv[kh][kw], wt[kh * 3 + kw], acc);
            acc.x = fmaxf(acc.x, 0.0f);
            acc.y = fmaxf(acc.y, 0.0f);
            acc.z = fmaxf(acc.z, 0.0f);
            acc.w = fmaxf(acc.w, 0.0f);
            rmax = range_of(rmax, acc);
            f16x4 hi, lo;
            split_f16(acc.x, acc.y, acc.z, acc.w, hi, lo);
            const int off = swz64(r * 32 + col, c4 >> 1) + (c4 & 1) * 8;
            *reinterpret_cast<f16x4*>(s_ah + off) = hi;
            *reinterpret_cast<f16x4*>(s_al + off) = lo;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                cv[0][kw] = cv[1][kw];
                cv[1][kw] = cv[2][kw];
            }
        }
    }
    __syncthreads();

    v4f d3wt[9];                                // depthwise-3 taps (channels 4 (tid & 15) ..): in flight during D and E
#pragma unroll
    for (int t = 0; t < 9; ++t) d3wt[t] = *reinterpret_cast<const v4f*>(dw3_w + t * 64 + (tid & 15) * 4);
    const v4f d3bias = *reinterpret_cast<const v4f*>(dw3_b + (tid & 15) * 4);
    // ---- D: GEMM.  Waves (wr, wc): column tile wc; row tiles wr, wr + 2 and, for wr == 0, 4 ----
    const int wr = wave >> 1, wc = wave & 1;
    const int frow = lane & 31, fh = lane >> 5;
    f32x16 acc2[3];                             // (the first MFMA of a tile takes a literal zero: no 48 moves to clear them)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int rt = wr + 2 * i;                                  // row tile 0..4 (5 = none)
            if (rt < R2) {
                const int off = swz64(rt * 32 + frow, 2 * s2 + fh);
                const f16x8 ah = *reinterpret_cast<const f16x8*>(s_ah + off);
                const f16x8 al = *reinterpret_cast<const f16x8*>(s_al + off);
                f32x16 c = acc2[i];
                if (s2 == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) c[r] = 0.0f;
                }
                // operands swapped: the accumulators hold the TRANSPOSED tile (lane = position, four consecutive
                // channels per register quad), so phase E writes 16 bytes at a time; same products, same k order
                if constexpr (!PLAIN) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wbh[s2], al, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wbl[s2], ah, c, 0, 0, 0);
                }
                acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wbh[s2], ah, c, 0, 0, 0);
            }
        }
    }
    __syncthreads();   // every wave is done with the A tile, the conv band and the log-mel band: P may overwrite them

    // ---- E: bias + ReLU -> P; layer-2 rows past row 47 are the depthwise's zero padding ----
    {
        // transposed accumulators: lane -> position rt * 32 + frow; registers 4 g .. 4 g + 3 -> channels
        // wc * 32 + 8 g + 4 fh + (0..3)
        v4f b4[4], u4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            b4[g] = *reinterpret_cast<const v4f*>(pw_b + wc * 32 + 8 * g + 4 * fh);
            u4[g] = *reinterpret_cast<const v4f*>(pw_u + wc * 32 + 8 * g + 4 * fh);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int rt = wr + 2 * i;
            if (rt < R2) {
                const bool live = __builtin_amdgcn_readfirstlane((int)(r0 + rt < 48)) != 0;    // the same for the whole wave
                float* prow = P + (rt * 32 + frow) * PW + wc * 32 + 4 * fh;
                if (live) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        // (vector fma: two v_pk_fma_f32 instead of four v_fma_f32; the same IEEE operations)
                        v4f v = __builtin_elementwise_fma(v4f{acc2[i][4 * g + 0], acc2[i][4 * g + 1], acc2[i][4 * g + 2], acc2[i][4 * g + 3]},
                                                          u4[g], b4[g]);
                        v.x = fmaxf(v.x, 0.0f);
                        v.y = fmaxf(v.y, 0.0f);
                        v.z = fmaxf(v.z, 0.0f);
                        v.w = fmaxf(v.w, 0.0f);
                        *reinterpret_cast<v4f*>(prow + 8 * g) = v;
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) *reinterpret_cast<v4f*>(prow + 8 * g) = v4f{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
    }
    __syncthreads();

    f16x8 w3h[4], w3l[4];                       // this lane's layer-3 weight fragments, k16 steps 0..3 (in flight during F)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const size_t f = ((size_t)(wave * 4 + q) * 64 + lane) * 8;
        w3h[q] = *reinterpret_cast<const f16x8*>(W3fhi + f);
        w3l[q] = *reinterpret_cast<const f16x8*>(W3flo + f);
    }
    // ---- F: depthwise 3, stride 2: out[o][ow][c] from P rows 2o + kh, columns 2ow + kw (column 32 = padding) ----
    // 512 tasks: o (2) x ow (16) x c4 (16); a thread keeps its column and channels in both (o = it).  Every tap is an
    // immediate offset from one pointer; the tap right of column 31 (ow = 15, kw = 2) is read like the others and replaced by
    // the zero padding afterwards (what it reads - the next row, or for the last one the bytes after P - is never used)
    const int c16 = tid & 15, ow = (tid >> 4) & 15;
    const float* const pcol = P + (2 * ow) * PW + c16 * 4;
    const bool right_edge = ow == 15;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int o = it;
        v4f acc = d3bias;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                v4f v = *reinterpret_cast<const v4f*>(pcol + ((2 * o + kh) * 32 + kw) * PW);
                if (kw == 2) {
                    v.x = right_edge ? 0.0f : v.x;
                    v.y = right_edge ? 0.0f : v.y;
                    v.z = right_edge ? 0.0f : v.z;
                    v.w = right_edge ? 0.0f : v.w;
                }
                acc = __builtin_elementwise_fma(v, d3wt[kh * 3 + kw], acc);
            }
        acc.x = fmaxf(acc.x, 0.0f);
        acc.y = fmaxf(acc.y, 0.0f);
        acc.z = fmaxf(acc.z, 0.0f);
        acc.w = fmaxf(acc.w, 0.0f);
        rmax = range_of(rmax, acc);
        f16x4 hi, lo;
        split_f16(acc.x, acc.y, acc.z, acc.w, hi, lo);
        const int c = c16 & 7;
        const int off = (c16 >> 3) * 32 * 64 + swz64(o * 16 + ow, c >> 1) + (c & 1) * 8;
        *reinterpret_cast<f16x4*>(smem + OFF_A3H + off) = hi;
        *reinterpret_cast<f16x4*>(smem + OFF_A3L + off) = lo;
    }
    {
        __syncthreads();
        // ---- G: [32][64] x [64][128], one 32 x 32 tile per wave ----
        f32x16 acc3;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[r] = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int off = (q >> 1) * 32 * 64 + swz64(frow, 2 * (q & 1) + fh);
            const f16x8 ah = *reinterpret_cast<const f16x8*>(smem + OFF_A3H + off);
            const f16x8 al = *reinterpret_cast<const f16x8*>(smem + OFF_A3L + off);
            if constexpr (!PLAIN) {
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, w3h[q], acc3, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w3l[q], acc3, 0, 0, 0);
            }
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w3h[q], acc3, 0, 0, 0);
        }
        // ---- H: bias + ReLU, [32][128] block of the layer-3 output (rows are consecutive NHWC positions) ----
        float* dst3 = out + (((size_t)win * 24 + 2 * ob) * 16) * 128;
        const int n = 32 * wave + frow;
        const float b = pw3_b[n], u = pw3_u[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = 4 * fh + (r & 3) + 8 * (r >> 2);
            typedef float v2f_ __attribute__((ext_vector_type(2)));
            const v2f_ t2 = __builtin_elementwise_fma(v2f_{acc3[r & ~1], acc3[r | 1]}, v2f_{u, u}, v2f_{b, b});   // one v_pk_fma_f32 per two outputs
            dst3[(size_t)m * 128 + n] = fmaxf((r & 1) ? t2.y : t2.x, 0.0f);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    range_report(rmax, range_flag);
}

// --------------------------------------------------------------------------- pool + head
template <int POOL>   // 6: act = [window][6][1024] is pooled here; 1: act = [window][1024] is already the pooled embedding
__global__ __launch_bounds__(256) void pool_head_kernel(const float* __restrict__ act,
                                                        const float* __restrict__ head_wt,
                                                        const float* __restrict__ head_b, int n_classes,
                                                        float* __restrict__ emb, float* __restrict__ logits) {
    // one workgroup per window; act = [window][6][1024]
    __shared__ float s_part[4][BD_MAX_CLASSES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const size_t win = blockIdx.x;
    const float4* src = reinterpret_cast<const float4*>(act + win * POOL * BD_EMBEDDING_SIZE);
    float4 s = src[tid];
    if constexpr (POOL > 1) {
#pragma unroll
        for (int p = 1; p < POOL; ++p) {
            const float4 v = src[p * (BD_EMBEDDING_SIZE / 4) + tid];
            s.x += v.x;
            s.y += v.y;
            s.z += v.z;
            s.w += v.w;
        }
        s.x /= (float)POOL;
        s.y /= (float)POOL;
        s.z /= (float)POOL;
        s.w /= (float)POOL;
        if (emb) reinterpret_cast<float4*>(emb + win * BD_EMBEDDING_SIZE)[tid] = s;
    }
    if (!logits) return;

    // classes in groups of 16: all weight rows of a group are in flight together and the 16 butterflies are
    // independent (one class at a time was a chain of 13 dependent L2 round trips, 13 us for 4 MB of input)
    for (int c0 = 0; c0 < n_classes; c0 += 16) {
        float4 w[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int c = c0 + j < n_classes ? c0 + j : n_classes - 1;
            w[j] = reinterpret_cast<const float4*>(head_wt + (size_t)c * BD_EMBEDDING_SIZE)[tid];
        }
        float p[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) p[j] = fmaf(s.x, w[j].x, fmaf(s.y, w[j].y, fmaf(s.z, w[j].z, s.w * w[j].w)));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int j = 0; j < 16; ++j) p[j] += __shfl_xor(p[j], o, 64);
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (lane == 0 && c0 + j < n_classes) s_part[wave][c0 + j] = p[j];
    }
    __syncthreads();
    if (tid < n_classes)
        logits[win * n_classes + tid] =
            ((s_part[0][tid] + s_part[1][tid]) + (s_part[2][tid] + s_part[3][tid])) + head_b[tid];
}

}  // namespace

void launch_conv1(const float* logmel, int patch_step, const WindowMap& map, int w0, int windows, const float* w9x32,
                  const float* b32, float* out, hipStream_t stream) {
    if (windows <= 0) return;
    hipLaunchKernelGGL(conv1_kernel, dim3(48 / kC1Rows, windows), dim3(256), 0, stream, logmel, patch_step, map, w0,
                       w9x32, b32, out);
}

template <int STRIDE, int OWB>
static void launch_dw(const float* in, float* out, int windows, const SepLayer& L, hipStream_t stream) {
    const long long total = (long long)windows * L.h_out * (L.w_out / OWB) * (L.cin / 4);
    const long long blocks = (total + 255) / 256;
    const int grid = (int)(blocks < (1 << 20) ? blocks : (1 << 20));
    hipLaunchKernelGGL((depthwise_kernel<STRIDE, OWB>), dim3(grid), dim3(256), 0, stream, in, out, dw_w_of(L), dw_b_of(L),
                       windows, L.h_in, L.w_in, L.cin, L.h_out, L.w_out, L.amax);
}

void launch_scale_copy(const float* src, float* dst, int64_t n, float factor, hipStream_t stream) {
    if (n <= 0) return;
    const long long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(scale_copy_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, stream, src, dst,
                       (long long)n, factor);
}

void launch_depthwise(const float* in, float* out, int windows, const SepLayer& L, hipStream_t stream) {
    if (windows <= 0) return;
    const bool wide = L.w_out % 4 == 0;
    if (L.stride == 1) {
        if (wide) launch_dw<1, 4>(in, out, windows, L, stream);
        else launch_dw<1, 2>(in, out, windows, L, stream);
    } else {
        if (wide) launch_dw<2, 4>(in, out, windows, L, stream);
        else launch_dw<2, 2>(in, out, windows, L, stream);
    }
}

// variant: 0 = pick by shape; otherwise an explicit tile (test / tuning hook)
int launch_pointwise_variant(const float* A, const float* Wt, const float* bias, float* C, long long M, int N,
                             int K, int variant, hipStream_t stream) {
    if (M <= 0) return 0;
    if (K % kBK != 0 || N % 64 != 0) return -1;
    if (variant == 0) {
        // tools/gemm_sweep.py 1024 ... f32 on MI355X (round 4, persistent kernel): the first tile of the shape's list that
        // gives every CU a tile; 64 x 64 tiles for small batches
        struct Opt { int variant, bm, bn; };
        static const Opt n64[] = {{8, 64, 64}};
        static const Opt n128_k64[] = {{1, 128, 128}, {8, 64, 64}};
        static const Opt n128[] = {{9, 96, 128}, {8, 64, 64}};
        static const Opt n256[] = {{7, 128, 256}, {9, 96, 128}, {8, 64, 64}};
        static const Opt wide[] = {{9, 96, 128}, {8, 64, 64}};
        const Opt* list = wide;
        int count = 2;
        if (N == 64) { list = n64; count = 1; }
        else if (N == 128 && K <= 64) { list = n128_k64; count = 2; }
        else if (N == 128) { list = n128; count = 2; }
        else if (N == 256) { list = n256; count = 3; }
        variant = 8;
        for (int i = 0; i < count; ++i) {
            if (N % list[i].bn) continue;
            const long long tiles = ((M + list[i].bm - 1) / list[i].bm) * (N / list[i].bn);
            if (tiles >= 256 || i == count - 1) { variant = list[i].variant; break; }
        }
    }
    switch (variant) {
        case 1: if (N % 128) return -1; launch_pw<128, 128, 2, 2>(A, Wt, bias, C, M, N, K, stream); break;
        case 2: launch_pw<128, 64, 2, 2>(A, Wt, bias, C, M, N, K, stream); break;
        case 3: if (N % 128) return -1; launch_pw<256, 128, 4, 2>(A, Wt, bias, C, M, N, K, stream); break;
        case 4: if (N % 128) return -1; launch_pw<192, 128, 2, 2>(A, Wt, bias, C, M, N, K, stream); break;
        case 5: launch_pw<256, 64, 4, 1>(A, Wt, bias, C, M, N, K, stream); break;
        case 6: if (N % 128) return -1; launch_pw<64, 128, 1, 4>(A, Wt, bias, C, M, N, K, stream); break;
        case 7: if (N % 256) return -1; launch_pw<128, 256, 2, 4>(A, Wt, bias, C, M, N, K, stream); break;
        case 8: launch_pw<64, 64, 2, 2>(A, Wt, bias, C, M, N, K, stream); break;
        case 9: if (N % 128) return -1; launch_pw<96, 128, 1, 4>(A, Wt, bias, C, M, N, K, stream); break;
        case 10: launch_pw<96, 64, 1, 2>(A, Wt, bias, C, M, N, K, stream); break;
        default: return -1;
    }
    return 0;
}

// Exact-f32 1x1 convolution of layer L with the depthwise of the NEXT layer applied in the kernel's epilogue (96 x 128 tiles of
// whole windows): in = L's depthwise output [windows][h][w][cin], out = the next layer's depthwise output.  false = shape not
// covered (the caller runs the two kernels).
bool launch_pointwise_next_dw_f32(const float* in, float* out, int windows, const SepLayer& L, const SepLayer& Ln,
                                  hipStream_t stream) {
    if (windows <= 0) return true;
    if (L.cout % 128 != 0 || L.cin % kBK != 0 || Ln.cin != L.cout || Ln.h_in != L.h_out || Ln.w_in != L.w_out) return false;
    const long long M = (long long)windows * L.h_out * L.w_out;
    const int h = L.h_out, w = L.w_out, st = Ln.stride;
#define BD_NDW_CASE(H_, W_, S_)                                                                                         \
    if (h == H_ && w == W_ && st == S_) {                                                                               \
        launch_pw<96, 128, 1, 4, H_, W_, S_>(in, L.pw_wt, L.pw_b, out, M, L.cout, L.cin, stream, Ln.dw_w, Ln.dw_b, windows); \
        return true;                                                                                                    \
    }
    BD_NDW_CASE(12, 8, 1)
    BD_NDW_CASE(12, 8, 2)
    BD_NDW_CASE(6, 4, 1)
    BD_NDW_CASE(6, 4, 2)
#undef BD_NDW_CASE
    return false;
}

// Tile choice for the split-f16 kernel, from tools/gemm_sweep.py on MI355X at 1024 windows; when the
// batch is too small to give every CU a tile of the preferred shape, fall back to smaller tiles.
static int pick_f16x3_variant(long long M, int N, int K) {
    struct Opt { int variant, bm, bn; };
    static const Opt big_k[] = {{9, 256, 256}, {7, 128, 256}, {6, 64, 128}, {8, 64, 64}};     // K >= 256, N >= 512
    static const Opt mid[] = {{7, 128, 256}, {1, 128, 128}, {6, 64, 128}, {8, 64, 64}};        // N == 256
    static const Opt n128[] = {{3, 256, 128}, {1, 128, 128}, {6, 64, 128}, {8, 64, 64}};       // N == 128
    static const Opt n64[] = {{2, 128, 64}, {8, 64, 64}};                                       // N == 64
    static const Opt deep[] = {{2, 128, 64}, {6, 64, 128}, {8, 64, 64}};                        // N == 1024
    const Opt* list;
    int count;
    if (N == 64) { list = n64; count = 2; }
    else if (N == 128) { list = n128; count = 4; }
    else if (N == 256) { list = mid; count = 4; }
    else if (N >= 1024) { list = deep; count = 3; }
    else { list = big_k; count = 4; }
    for (int i = 0; i < count; ++i) {
        if (N % list[i].bn) continue;
        const long long tiles = ((M + list[i].bm - 1) / list[i].bm) * (N / list[i].bn);
        if (tiles >= 256 || i == count - 1) return list[i].variant;
    }
    (void)K;
    return 8;
}

int launch_pointwise_f16x3_variant(const float* A, const void* Whi, const void* Wlo, const float* unscale, const float* bias,
                                   float* C, long long M, int N, int K, int variant, hipStream_t stream, bool plain,
                                   unsigned* range_flag) {
    if (M <= 0) return 0;
    if (K % 32 != 0 || N % 64 != 0) return -1;
    const _Float16* wh = static_cast<const _Float16*>(Whi);
    const _Float16* wl = static_cast<const _Float16*>(Wlo);
    if (variant == 0) variant = pick_f16x3_variant(M, N, K);
    switch (variant) {
        case 1: if (N % 128) return -1; launch_pw16<128, 128, 2, 2>(A, wh, wl, unscale, bias, C, M, N, K, plain, range_flag, stream); break;
        case 2: launch_pw16<128, 64, 2, 2>(A, wh, wl, unscale, bias, C, M, N, K, plain, range_flag, stream); break;
        case 3: if (N % 128) return -1; launch_pw16<256, 128, 4, 2>(A, wh, wl, unscale, bias, C, M, N, K, plain, range_flag, stream); break;
        case 4: if (N % 256) return -1; launch_pw16<128, 256, 2, 2>(A, wh, wl, unscale, bias, C, M, N, K, plain, range_flag, stream); break;
        case 5: launch_pw16<256, 64, 4, 1>(A, wh, wl, unscale, bias, C, M, N, K, plain, range_flag, stream); break;
        case 6: if (N % 128) return -1; launch_pw16<64, 128, 1, 4>(A, wh, wl, unscale, bias, C, M, N, K, plain, range_flag, stream); break;
        case 7: if (N % 256) return -1; launch_pw16<128, 256, 2, 4>(A, wh, wl, unscale, bias, C, M, N, K, plain, range_flag, stream); break;
        case 8: launch_pw16<64, 64, 2, 2>(A, wh, wl, unscale, bias, C, M, N, K, plain, range_flag, stream); break;
        case 9: if (N % 256) return -1; launch_pw16<256, 256, 4, 2>(A, wh, wl, unscale, bias, C, M, N, K, plain, range_flag, stream); break;
        default: return -1;
    }
    return 0;
}

// Pointwise 1x1 convolution on the wave-specialised kernel: the producers only split the input rows into
// f16 hi + lo (no depthwise).  For the stride-2 layers, whose depthwise runs as its own kernel or in the
// previous layer's epilogue.  Same products in the same order as pointwise_f16x3_kernel: bit-identical.
bool launch_pointwise_ws(const float* in, float* out, int64_t rows, const SepLayer& L, hipStream_t stream) {
    if (rows <= 0 || rows >= (1LL << 31) || L.cin < 128 || L.cin % 64 != 0 || L.cout % 256 != 0) return false;
    // layers 5 and 7: few enough input channels for the weights to live in registers (variant 10 keeps the tile kernel)
    if (L.pw_variant16 != 10 && (L.cin == 128 || L.cin == 256) && L.cout <= 2048) {
        if (L.cin == 128) launch_pw_res<8>(in, L, out, (int)rows, stream);
        else launch_pw_res<16>(in, L, out, (int)rows, stream);
        return true;
    }
    // (96 x 128 tiles with two workgroups per CU measured the same: 33.1 vs 32.6 us on layer 7)
    launch_sep_ws<0, 1>(in, L, out, rows, stream);
    return true;
}

void launch_pointwise(const float* in, float* out, int64_t rows, const SepLayer& L, hipStream_t stream) {
    const bool f16 = L.pw_mode == 1 || L.pw_mode == 2;
    if (f16 && (L.pw_variant16 == 0 || L.pw_variant16 >= 10) && launch_pointwise_ws(in, out, rows, L, stream)) return;
    if (f16)
        launch_pointwise_f16x3_variant(in, L.pw_whi, L.pw_wlo, L.pw_u, L.pw_b, out, rows, L.cout, L.cin, L.pw_variant16,
                                       stream, L.pw_mode == 2, L.range_flag);
    else
        launch_pointwise_variant(in, L.pw_wt, L.pw_b, out, rows, L.cout, L.cin, L.pw_variant, stream);
}

// Fused depthwise+pointwise of layer L followed by the stride-2 depthwise of the NEXT layer; `out` receives
// that depthwise's output [windows][H/2][W/2][L.cout].  Only for whole-window tiles (12x8 and 6x4 maps).
bool launch_separable_fused_next_dw(const float* in, float* out, int windows, const SepLayer& L, const SepLayer& next,
                                    hipStream_t stream) {
    const int P = L.h_out * L.w_out;
    if (L.stride == 1 && next.stride == 2 && windows > 0 && P == 384 && L.w_out == 16 && L.cin == 128 && L.cout == 128 &&
        next.cin == 128) {                        // layer 4 + depthwise 5: a window per workgroup
        launch_l4_window(in, L, next, out, windows, stream);
        return true;
    }
    if (L.stride != 1 || next.stride != 2 || windows <= 0 || L.cin < 128 || L.cout % 256 != 0) return false;
    // (the epilogue's position arithmetic is compiled for these two maps)
    if (!((L.h_out == 12 && L.w_out == 8) || (L.h_out == 6 && L.w_out == 4)) || next.cin != L.cout) return false;
    const long long M = (long long)windows * P;
    if (M >= (1LL << 31)) return false;       // the kernel's tile arithmetic is 32-bit
    launch_sep_ws<1, 0>(in, L, out, M, stream, &next);
    return true;
}

// The run of stride-1 512 -> 512 layers on the 6 x 4 map (layers 8-11) extended by the layer that closes it - the stride-1
// 512 -> 512 layer whose successor is a stride-2 one (layer 12) - with that successor's depthwise in the epilogue: layers 8-12 +
// depthwise 13 as ONE launch of the on-chip kernel (sepchip.hip), a -> b = [windows][3][2][512] (planes: as f16 hi / lo planes,
// what septail.hip reads).  Returns the number of layers of L it ran (5) or 0 (the caller goes on layer by layer).
int launch_separable_run_next_dw(const float* a, float* b, int windows, const SepLayer* L, int max_layers, hipStream_t stream,
                                 bool planes) {
    int n = 0;
    while (n < 5 && n + 1 < max_layers) {
        const SepLayer& l = L[n];
        if (l.stride != 1 || l.cin != 512 || l.cout != 512 || l.h_out != 6 || l.w_out != 4 || l.pw_mode != L[0].pw_mode) return 0;
        ++n;
        if (L[n].stride != 1) break;
    }
    if (n < 2 || n > 5 || n >= max_layers || L[n].stride != 2 || L[n].cin != 512 || windows <= 0 || (long long)windows * 24 >= (1LL << 31))
        return 0;
    return launch_separable_chip(a, b, windows, L, n, stream, &L[n], planes) ? n : 0;
}

// Layers 1-3 complete: out = [windows][24][16][128], the layer-3 output.
void launch_stem4(const float* logmel, int patch_step, const WindowMap& map, int w0, int windows, const float* c1_w,
                  const float* c1_b, const SepLayer& L2, const SepLayer& L3, float* out, hipStream_t stream) {
    if (windows <= 0) return;
#define BD_STEM4(PLAIN)                                                                                              \
    hipLaunchKernelGGL((stem3_kernel<PLAIN>), dim3(12, windows), dim3(256), 0, stream, logmel, patch_step, map, w0,   \
                       c1_w, c1_b, dw_w_of(L2), dw_b_of(L2), static_cast<const _Float16*>(L2.pw_whi),                     \
                       static_cast<const _Float16*>(L2.pw_wlo), L2.pw_u, L2.pw_b, dw_w_of(L3), dw_b_of(L3), out,            \
                       static_cast<const _Float16*>(L3.pw_fhi), static_cast<const _Float16*>(L3.pw_flo), L3.pw_u, L3.pw_b,  \
                       L2.range_flag)
    if (L2.pw_mode == 2) BD_STEM4(true);
    else BD_STEM4(false);
#undef BD_STEM4
}

void launch_pool_head(const float* act, int windows, const float* head_wt, const float* head_b,
                      int n_classes, float* emb, float* logits, hipStream_t stream) {
    if (windows <= 0) return;
    hipLaunchKernelGGL(pool_head_kernel<6>, dim3(windows), dim3(256), 0, stream, act, head_wt, head_b,
                       n_classes, emb, logits);
}

// Dense head on embeddings that are already pooled: pooled = [windows][1024].
void launch_head(const float* pooled, int windows, const float* head_wt, const float* head_b, int n_classes,
                 float* logits, hipStream_t stream) {
    if (windows <= 0 || !logits) return;
    hipLaunchKernelGGL(pool_head_kernel<1>, dim3(windows), dim3(256), 0, stream, pooled, head_wt, head_b,
                       n_classes, static_cast<float*>(nullptr), logits);
}

}  // namespace bd
