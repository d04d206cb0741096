// Layers 8-11 of YAMNet (yamnet.py:86-89: four separable layers of ONE shape, 512 -> 512 channels on the 6 x 4 map, stride 1)
// as one launch in which only the run's first input and last output touch global memory (round 5; VERDICT r4 next #1a).
//
// The hand-off between two layers is a [96 rows = 4 windows][512] tile: 196 KB as f32 or as the next layer's (hi, lo) f16
// operand - more than the CU's 160 KB of LDS.  It does fit the CU when the register file carries part of it:
//
//   * 8 waves, two per SIMD, 256 VGPRs each, all of one kind (no producer waves).  Wave w owns the output columns of the
//     32-column blocks w and w + 8: 3 row tiles x 2 column tiles of v_mfma_f32_32x32x16_f16 = 96 accumulator registers.
//   * At the end of a layer the accumulators ARE the tile: lane (c, h) holds column c and, per 32-row tile, rows
//     {0-3, 8-11, 16-19, 24-27} + 4 h, i.e. the even (h = 0) or odd (h = 1) rows of the stacked 24 x 4 map of the four windows.
//     One v_permlane32_swap per register pair trades the odd rows of windows 0-1 against the even rows of windows 2-3, after
//     which lane (c, h) holds windows 2 h and 2 h + 1 of channel c completely: the next layer's depthwise 3 x 3 runs in
//     registers with compile-time neighbours (no LDS tap reads, no slab, no padding tests), two windows at a time in
//     v_pk_fma_f32.  Taps and shift are per-lane values (one channel per lane): ten global dwords per column block.
//   * Its outputs, split into f16 hi + lo, are the next layer's A operand: stage s (32 input channels) of the K loop is the
//     column block s of the previous layer, so wave w publishes stages w and w + 8 into an LDS ring of 13 stage tiles
//     (13 x 12.1 KB = 157.6 KB); stages 13-15 stay PENDING in the registers of waves 5-7 (48 packed dwords each) until stages
//     0-2 have been consumed, and then take their slots.  Four workgroup barriers per layer (tile published / slots 0-2
//     free / pending published / tile consumed) instead of one per stage: between them the eight waves drift freely and two
//     matrix waves per SIMD cover each other's LDS and L2 latency.
//   * Weights as in sep_w12_kernel: B fragments straight from the fragment-ordered copy (L2) into registers, two k16 steps
//     ahead.
//
// Arithmetic is that of sep_w12_kernel / depthwise_kernel + pointwise_f16x3_kernel bit for bit: the depthwise sums shift +
// taps in row-major tap order with fmaf, ReLU, the split, and per accumulator the products lo*hi, hi*lo, hi*hi of k16 step
// q = 0..31 in ascending order; which wave owns which column block changes nothing.  Taps that fall outside the map are skipped
// instead of multiplied by zero: fma(0, w, a) == a for every a but -0.0, and a sum that starts at a float shift is never -0.0
// unless the shift is (then the result differs in the sign of a zero that the ReLU absorbs).
#include "bd_internal.h"

#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace bd {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr float kF16MaxChip = 65504.0f;

// LDS image of one stage's A tile: two halves (hi, lo) of [96 rows][32 k] f16 = 64-byte rows, the 16-byte slot of a row
// XORed with (row >> 2) & 3 (sep_w12_kernel's swizzle: the 8 rows a ds_read_b128 lane group touches land in different
// banks), and rows 48.. pushed back by one row: the two half-waves of a publishing wave write rows r and r + 48 at once,
// which would otherwise meet in the same 16 banks.
constexpr int kChipHalfBytes = 97 * 64;
constexpr int kChipSlotBytes = 2 * kChipHalfBytes;

struct ChipChain {
    const float* dw_w[4];          // [9][512] depthwise taps * bn scale * 2^act_exp
    const float* dw_b[4];          // [512]
    const _Float16* whi[4];        // pointwise weights, MFMA B-fragment order [512/32][512/16][64][8]
    const _Float16* wlo[4];
    const float* pw_u[4];          // [512] epilogue factors
    const float* pw_b[4];          // [512]
};
// Pointer `field` of layer `li`, read from the kernel-argument segment with a scalar load (the chain is the kernel's FIRST
// argument, i.e. at offset 0).  Indexing the by-value argument with a run-time layer makes a scratch copy of it; selecting
// among its 24 pointers keeps all of them in scalar registers for the whole kernel (144 of them spilled to lanes).
template <typename T>
__device__ __forceinline__ const T* chain_ptr(int field, int li) {
    typedef const __attribute__((address_space(4))) unsigned long long* kptr;
    const kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    return reinterpret_cast<const T*>(ka[field * 4 + li]);
}
static_assert(sizeof(ChipChain) == 24 * 8, "six tables of four pointers");

template <bool PLAIN, int NSLOT, bool TRACE = false>
__global__ __launch_bounds__(512, 2) void sep_chip_kernel(const ChipChain ch, const float* X, float* Y, int nl,
                                                           long long M, unsigned* __restrict__ range_flag,
                                                           unsigned long long* __restrict__ dbg = nullptr) {
    static_assert(NSLOT >= 9 && NSLOT <= 16, "ring size");
    constexpr int K = 512, KQ = K / 16;
    constexpr int NPEND = 16 - NSLOT;              // stages that wait in registers; their owners are waves 8 - NPEND .. 7
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wc = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fh = lane >> 5;
    const long long m0 = (long long)blockIdx.x * 96;
    float rmax = 0.0f;
    // developer aid (-DBD_KERNEL_TRACE build, BD_WS_TRACE=7): shader-clock stamps of waves 0 and 5 of workgroup 0 at every phase edge
    int tsn = 0;
#define CHIP_TS()                                                                                         \
    if constexpr (TRACE) {                                                                                \
        if (blockIdx.x == 0 && lane == 0 && (wc == 0 || wc == 5) && tsn < 64)                             \
            dbg[(wc == 5 ? 64 : 0) + tsn] = __builtin_amdgcn_s_memtime();                                 \
        ++tsn;                                                                                            \
    }
    CHIP_TS()

    // publisher: byte offset of this lane's k (= frow) in a row whose swizzle key is m; rows 48 fh + r' follow as immediates
    int wb[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) wb[m] = fh * (48 * 64 + 64) + ((((frow >> 3) ^ m)) << 4) + 2 * (frow & 7);
    // reader: lane (frow, fh) supplies A[row 32 i + frow][k = 16 s + 8 fh ..]; the key (row >> 2) & 3 is the same for all i
    int ro[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int row = 32 * i + frow;
            ro[i][s] = row * 64 + (row >= 48 ? 64 : 0) + ((((2 * s + fh) ^ ((frow >> 2) & 3))) << 4);
        }

    f32x16 acc[3][2];
    unsigned pend[48];

    // Depthwise 3 x 3 + shift + ReLU + split of column block J (stage ST) of the tile held as in2[y][x] = (window 2 fh,
    // window 2 fh + 1) at map position (y, x) of channel c: 48 outputs per lane, published into slot ST of the ring, or -
    // always for J = 1, only waves 8 - NPEND .. 7 use them - kept as packed (hi | lo << 16) dwords in pend[].
#define CHIP_DW(J, ST, DW_W, DW_B)                                                                        \
    {                                                                                                     \
        float wt[9];                                                                                      \
        _Pragma("unroll") for (int t = 0; t < 9; ++t) wt[t] = (DW_W)[t * K + 32 * (ST) + frow];           \
        const float shift = (DW_B)[32 * (ST) + frow];                                                     \
        char* const slot = sm + ((J) == 0 ? (ST) : 0) * kChipSlotBytes;                                   \
        _Pragma("unroll") for (int y = 0; y < 6; ++y)                                                     \
            _Pragma("unroll") for (int x = 0; x < 4; ++x) {                                               \
                v2f a = {shift, shift};                                                                   \
                _Pragma("unroll") for (int kh = 0; kh < 3; ++kh)                                          \
                    _Pragma("unroll") for (int kw = 0; kw < 3; ++kw) {                                    \
                        const int iy = y + kh - 1, ix = x + kw - 1;                                       \
                        if (iy < 0 || iy >= 6 || ix < 0 || ix >= 4) continue;                             \
                        a = __builtin_elementwise_fma(in2[iy][ix], v2f{wt[kh * 3 + kw], wt[kh * 3 + kw]}, a); \
                    }                                                                                     \
                _Pragma("unroll") for (int w = 0; w < 2; ++w) {                                           \
                    const float v = fmaxf(w ? a.y : a.x, 0.0f);                                           \
                    rmax = fmaxf(rmax, v);                                                                \
                    unsigned pk = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)v);              \
                    asm("v_fma_mixhi_f16 %0, %0, -1.0, %1 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(pk) : "v"(v)); \
                    const int rl = 24 * w + 4 * y + x;      /* row 48 fh + rl; key ((48 fh + rl) >> 2) & 3 */ \
                    if ((J) == 1) pend[rl] = pk;                                                          \
                    else CHIP_PUT(slot, rl, pk)                                                           \
                }                                                                                         \
            }                                                                                             \
        if ((J) == 1 && (ST) < NSLOT) {           /* one wave-uniform branch */                           \
            char* const slot1 = sm + (ST) * kChipSlotBytes;                                               \
            _Pragma("unroll") for (int rl = 0; rl < 48; ++rl) CHIP_PUT(slot1, rl, pend[rl])               \
        }                                                                                                 \
    }
#define CHIP_PUT(SLOT, RL, PK)                                                                            \
    {                                                                                                     \
        char* const p_ = (SLOT) + wb[((RL) >> 2) & 3] + (RL) * 64;                                        \
        *reinterpret_cast<unsigned short*>(p_) = (unsigned short)(PK);                                    \
        *reinterpret_cast<unsigned short*>(p_ + kChipHalfBytes) = (unsigned short)((PK) >> 16);           \
    }

    // ---------------------------------------------------------------------- layer 0: its depthwise reads the run's input
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int st = j ? wc + 8 : wc;
        const int c = 32 * st + frow;
        v2f in2[6][4];
#pragma unroll
        for (int y = 0; y < 6; ++y)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                long long ra = m0 + 48 * fh + 4 * y + x, rb = ra + 24;
                ra = ra < M ? ra : M - 1;
                rb = rb < M ? rb : M - 1;
                in2[y][x].x = X[(size_t)ra * K + c];
                in2[y][x].y = X[(size_t)rb * K + c];
            }
        if (j == 0) CHIP_DW(0, st, ch.dw_w[0], ch.dw_b[0])
        else CHIP_DW(1, st, ch.dw_w[0], ch.dw_b[0])
        CHIP_TS()
    }

    for (int li = 0; li < nl; ++li) {
        const _Float16* const Wfhi = chain_ptr<_Float16>(2, li);
        const _Float16* const Wflo = chain_ptr<_Float16>(3, li);
        __syncthreads();                          // stages 0 .. NSLOT - 1 of layer li published
        CHIP_TS()

        // ------------------------------------------------------------------ 1 x 1 convolution of layer li
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        // B fragments: column tile (wc, wc + 8), k16 step q -> ((tile * KQ + q) * 64 + lane) * 8 halves
        const _Float16* const wbh0 = Wfhi + ((size_t)wc * KQ * 64 + lane) * 8;
        const _Float16* const wbl0 = Wflo + ((size_t)wc * KQ * 64 + lane) * 8;
        constexpr int jstep = 8 * KQ * 512;       // halves between column tiles wc and wc + 8
        f16x8 bh0[2], bl0[2], bh1[2], bl1[2];
#define CHIP_BLOAD(BH, BL, Q)                                                                             \
    {                                                                                                     \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                   \
            BH[j] = *reinterpret_cast<const f16x8*>(wbh0 + j * jstep + (Q) * 512);                        \
            BL[j] = *reinterpret_cast<const f16x8*>(wbl0 + j * jstep + (Q) * 512);                        \
        }                                                                                                 \
    }
#define CHIP_ALOAD(AH, AL, S, I)                                                                          \
    {                                                                                                     \
        AH = *reinterpret_cast<const f16x8*>(abase + ro[I][S]);                                           \
        AL = *reinterpret_cast<const f16x8*>(abase + ro[I][S] + kChipHalfBytes);                          \
    }
#define CHIP_STEP(I, AH, AL, BH, BL)                                                                      \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                       \
        if constexpr (!PLAIN) {                                                                           \
            acc[I][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL, BH[j], acc[I][j], 0, 0, 0);            \
            acc[I][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BL[j], acc[I][j], 0, 0, 0);            \
        }                                                                                                 \
        acc[I][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BH[j], acc[I][j], 0, 0, 0);                \
    }
#define CHIP_STAGES(FROM, TO)                                                                             \
    for (int kk = (FROM); kk < (TO); ++kk) {                                                              \
        const char* const abase = sm + (kk < NSLOT ? kk : kk - NSLOT) * kChipSlotBytes;                   \
        f16x8 ah0, al0, ah1, al1;                                                                         \
        CHIP_ALOAD(ah0, al0, 0, 0)                                                                        \
        CHIP_ALOAD(ah1, al1, 0, 1)                                                                        \
        CHIP_STEP(0, ah0, al0, bh0, bl0)                                                                  \
        CHIP_ALOAD(ah0, al0, 0, 2)                                                                        \
        CHIP_STEP(1, ah1, al1, bh0, bl0)                                                                  \
        CHIP_ALOAD(ah1, al1, 1, 0)                                                                        \
        CHIP_STEP(2, ah0, al0, bh0, bl0)                                                                  \
        if (2 * kk + 2 < KQ) CHIP_BLOAD(bh0, bl0, 2 * kk + 2)                                             \
        CHIP_ALOAD(ah0, al0, 1, 1)                                                                        \
        CHIP_STEP(0, ah1, al1, bh1, bl1)                                                                  \
        CHIP_ALOAD(ah1, al1, 1, 2)                                                                        \
        CHIP_STEP(1, ah0, al0, bh1, bl1)                                                                  \
        CHIP_STEP(2, ah1, al1, bh1, bl1)                                                                  \
        if (2 * kk + 3 < KQ) CHIP_BLOAD(bh1, bl1, 2 * kk + 3)                                             \
    }
        CHIP_BLOAD(bh0, bl0, 0)
        CHIP_BLOAD(bh1, bl1, 1)
        CHIP_STAGES(0, NPEND)
        CHIP_TS()
        if constexpr (NPEND > 0) {
            __syncthreads();                      // every wave has read stages 0 .. NPEND - 1: their slots are free
            CHIP_TS()
            if (wc >= 8 - NPEND) {
                char* const slot = sm + (wc + 8 - NSLOT) * kChipSlotBytes;
#pragma unroll
                for (int rl = 0; rl < 48; ++rl) CHIP_PUT(slot, rl, pend[rl])
            }
        }
        CHIP_TS()
        CHIP_STAGES(NPEND, NSLOT)
        CHIP_TS()
        if constexpr (NPEND > 0) __syncthreads();  // pending stages published
        CHIP_TS()
        CHIP_STAGES(NSLOT, 16)
        CHIP_TS()
#undef CHIP_STAGES
#undef CHIP_STEP
#undef CHIP_ALOAD
#undef CHIP_BLOAD
        __syncthreads();                          // the ring is free for the next layer's tile
        CHIP_TS()
        if (li + 1 == nl) break;

        // ------------------------------------------------------------------ depthwise of layer li + 1 on the accumulators
        const float* const dw_w = chain_ptr<float>(0, li + 1);
        const float* const dw_b = chain_ptr<float>(1, li + 1);
        const float* const pu = chain_ptr<float>(4, li);
        const float* const pb = chain_ptr<float>(5, li);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int st = j ? wc + 8 : wc;       // stage of layer li + 1 = column block of layer li
            const float u = pu[32 * st + frow], b = pb[32 * st + frow];
            // stacked map row R = 8 i + 2 (r >> 2) + owner half, x = r & 3.  Pair q (rows 2 q, 2 q + 1) of windows 0-1 is
            // accumulator quad (q >> 2, q & 3); the same pair of windows 2-3 is quad ((q + 6) >> 2, (q + 6) & 3).
            float ev[12][4];                      // this lane's two windows, local map rows 0..11
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const float a01 = fmaxf(fmaf(acc[q >> 2][j][4 * (q & 3) + x], u, b), 0.0f);
                    const float a23 = fmaxf(fmaf(acc[(q + 6) >> 2][j][4 * ((q + 6) & 3) + x], u, b), 0.0f);
                    // v_permlane32_swap vdst, src: lanes 32-63 of vdst <-> lanes 0-31 of src
                    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a01),
                                                                    __builtin_bit_cast(unsigned, a23), false, false);
                    ev[2 * q][x] = __builtin_bit_cast(float, (unsigned)r[0]);
                    ev[2 * q + 1][x] = __builtin_bit_cast(float, (unsigned)r[1]);
                }
            v2f in2[6][4];
#pragma unroll
            for (int y = 0; y < 6; ++y)
#pragma unroll
                for (int x = 0; x < 4; ++x) in2[y][x] = v2f{ev[y][x], ev[6 + y][x]};
            if (j == 0) CHIP_DW(0, st, dw_w, dw_b)
            else CHIP_DW(1, st, dw_w, dw_b)
            CHIP_TS()
        }
    }
#undef CHIP_DW
#undef CHIP_PUT

    // ---------------------------------------------------------------------- the run's output: bias + ReLU from the accumulators
    {
        const float* const pu = chain_ptr<float>(4, nl - 1);
        const float* const pb = chain_ptr<float>(5, nl - 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = 32 * (j ? wc + 8 : wc) + frow;
            const float u = pu[c], b = pb[c];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long m = m0 + 32 * i + 8 * (r >> 2) + 4 * fh + (r & 3);
                    if (m < M) Y[(size_t)m * K + c] = fmaxf(fmaf(acc[i][j][r], u, b), 0.0f);
                }
        }
    }
    if (range_flag && !(rmax <= kF16MaxChip)) *range_flag = 1u;
    if constexpr (TRACE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CHIP_TS()
#undef CHIP_TS
}

constexpr int kMaxDevicesChip = 64;

template <bool PLAIN>
void launch_chip(const float* in, float* out, const SepLayer* L, int nl, long long M, hipStream_t stream) {
    ChipChain ch{};
    for (int i = 0; i < nl; ++i) {
        ch.dw_w[i] = dw_w_of(L[i]);
        ch.dw_b[i] = dw_b_of(L[i]);
        ch.whi[i] = static_cast<const _Float16*>(L[i].pw_fhi);
        ch.wlo[i] = static_cast<const _Float16*>(L[i].pw_flo);
        ch.pw_u[i] = L[i].pw_u;
        ch.pw_b[i] = L[i].pw_b;
    }
    constexpr int NSLOT = 13;
    constexpr int lds = NSLOT * kChipSlotBytes;
    static_assert(lds <= 160 * 1024, "the ring must fit the CU's LDS");
    static std::once_flag once[kMaxDevicesChip];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & (kMaxDevicesChip - 1)], [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sep_chip_kernel<PLAIN, NSLOT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    });
    const long long tiles = (M + 95) / 96;
#ifdef BD_KERNEL_TRACE      // developer build only: BD_WS_TRACE=7 stamps the phases of workgroup 0 (waves 0 and 5)
    const char* tr = getenv("BD_WS_TRACE");
    if (tr && tr[0] == '7') {
        static unsigned long long* dbg = nullptr;
        static int shots = 0;
        if (!dbg) {
            (void)hipMalloc(&dbg, 128 * 8);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sep_chip_kernel<PLAIN, NSLOT, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        }
        (void)hipMemsetAsync(dbg, 0, 128 * 8, stream);
        hipLaunchKernelGGL((sep_chip_kernel<PLAIN, NSLOT, true>), dim3((unsigned)tiles), dim3(512), lds, stream, ch, in, out, nl,
                           M, L[0].range_flag, dbg);
        (void)hipStreamSynchronize(stream);
        unsigned long long h[128];
        (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        if (++shots == 8)
            for (int w = 0; w < 2; ++w) {
                fprintf(stderr, "[trace] on-chip run, wave %d: cycles between stamps (start | dw0 j0 j1 | per layer: B1, stages 0-2, B2, pend, "
                                "stages 3-12, B3, stages 13-15, B4, [dw j0, dw j1] | stores):", w ? 5 : 0);
                for (int i = 1; i < 64 && h[w * 64 + i]; ++i) fprintf(stderr, " %llu", h[w * 64 + i] - h[w * 64 + i - 1]);
                fprintf(stderr, "\n");
            }
        return;
    }
#endif
    hipLaunchKernelGGL((sep_chip_kernel<PLAIN, NSLOT>), dim3((unsigned)tiles), dim3(512), lds, stream, ch, in, out, nl, M,
                       L[0].range_flag, (unsigned long long*)nullptr);
}

}  // namespace

// A run of stride-1 512 -> 512 layers on the 6 x 4 map with the tiles between its layers kept on the CU: reads `in`, writes
// `out`.  They may be the same buffer: a workgroup has read all rows of its tile (a tail tile's clamped rows are its own)
// before it writes any.  The caller has checked the shapes (launch_separable_run).
void launch_separable_chip(const float* in, float* out, int windows, const SepLayer* L, int nl, hipStream_t stream) {
    const long long M = (long long)windows * 24;
    if (L[0].pw_mode == 2) launch_chip<true>(in, out, L, nl, M, stream);
    else launch_chip<false>(in, out, L, nl, M, stream);
}

}  // namespace bd
