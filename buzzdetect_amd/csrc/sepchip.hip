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
//   * Weights as in the 12-wave kernel of rounds 2-5: B fragments straight from the fragment-ordered copy (L2) into registers, two k16 steps
//     ahead.
//
// Arithmetic is that of depthwise_kernel + pointwise_f16x3_kernel bit for bit: the depthwise sums shift +
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
// XORed with (row >> 2) & 3 (the swizzle of the round 2-5 kernels: the 8 rows a ds_read_b128 lane group touches land in different
// banks), and rows 48.. pushed back by one row: the two half-waves of a publishing wave write rows r and r + 48 at once,
// which would otherwise meet in the same 16 banks.
constexpr int kChipHalfBytes = 97 * 64;
constexpr int kChipSlotBytes = 2 * kChipHalfBytes;

constexpr int kChipMaxLayers = 5;
struct ChipChain {
    const float* dw_w[kChipMaxLayers];          // [9][512] depthwise taps * bn scale * 2^act_exp ([512] shift behind them)
    const float* dw_b[kChipMaxLayers];          // [512]
    const _Float16* whi[kChipMaxLayers];        // pointwise weights, MFMA B-fragment order [512/32][512/16][64][8]
    const _Float16* wlo[kChipMaxLayers];
    const float* pw_u[kChipMaxLayers];          // [512] epilogue factors
    const float* pw_b[kChipMaxLayers];          // [512]
    const float* ndw_w;                         // NDW: taps [9][512] + shift [512] of the stride-2 depthwise behind the run
};
// Pointer `field` of layer `li`, read from the kernel-argument segment with a scalar load (the chain is the kernel's FIRST
// argument, i.e. at offset 0).  Indexing the by-value argument with a run-time layer makes a scratch copy of it; selecting
// among its 30 pointers keeps all of them in scalar registers for the whole kernel (144 of them spilled to lanes).
template <typename T>
__device__ __forceinline__ const T* chain_ptr(int field, int li) {
    typedef const __attribute__((address_space(4))) unsigned long long* kptr;
    const kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    return reinterpret_cast<const T*>(ka[field * kChipMaxLayers + li]);
}
static_assert(sizeof(ChipChain) == (6 * kChipMaxLayers + 1) * 8, "six tables of five pointers + one");

// NDW (layer 12 closes the run and layer 13 is a stride-2 layer): the run's output is not written; layer 13's depthwise
// (3 x 3, stride 2, SAME = pad 0 before / 1 after on the 6 x 4 map: outputs 3 x 2) is applied to it in registers - after the
// same half-wave swap as between the layers a lane holds whole windows of its channel - and only [windows][3][2][512] goes
// to Y: the arithmetic of depthwise_kernel (shift, then the taps in row-major order with fmaf, ReLU).
// PLANES (with NDW): the depthwise-13 output leaves as the two f16 halves the tail's matrix kernel reads (septail.hip): hi plane
// [windows * 6][512] at Y, lo plane behind it - the split and the range guard of the kernel that would otherwise read it as f32.
template <bool PLAIN, int NSLOT, bool TRACE = false, bool NDW = false, bool PLANES = false>
__global__ __launch_bounds__(512, 2) void sep_chip_kernel(const ChipChain ch, const float* X, float* Y, int nl,
                                                           long long M, unsigned* __restrict__ range_flag,
                                                           unsigned long long* __restrict__ dbg = nullptr, int tune = 0) {
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

    // publisher: lane (frow = k, fh) writes element k of rows 48 fh + r'.  Byte offset of k in a row whose swizzle key is m:
    // (((k >> 3) ^ m) << 4) + 2 (k & 7) = wb0 ^ (m << 4) (bits 4-5 of everything else in wb0 are zero); rows follow as immediates
    const int wb0 = fh * (48 * 64 + 64) + ((frow >> 3) << 4) + 2 * (frow & 7);
    // reader: lane (frow, fh) supplies A[row 32 i + frow][k = 16 s + 8 fh ..] from slot (2 s + fh) ^ key, key = (row >> 2) & 3 =
    // (frow >> 2) & 3 for every i; s = 1 is the s = 0 address with bit 5 flipped.  Row tile 2 is row tile 0 + 4096 + 64 (an
    // immediate); row tile 1 straddles row 48 (the 64-byte skew starts at frow = 16) and keeps its own register.
    const int ra0 = frow * 64 + ((fh ^ ((frow >> 2) & 3)) << 4);
    const int ra1 = ra0 + 2048 + (frow >= 16 ? 64 : 0);
    const unsigned lane16 = lane * 16, c4 = frow * 4;

    f32x16 acc[3][2];
    unsigned pend[48], out0[48];

    // Depthwise 3 x 3 + shift + ReLU + split of column block J (stage ST) of the tile held as in2[y][x] = (window 2 fh,
    // window 2 fh + 1) at map position (y, x) of channel c: 48 outputs per lane, published into slot ST of the ring, or -
    // always for J = 1, only waves 8 - NPEND .. 7 use them - kept as packed (hi | lo << 16) dwords in pend[].
#define CHIP_DW(J, ST, TAPS)                                                                              \
    {                                                                                                     \
        float wt[9];                                                                                      \
        _Pragma("unroll") for (int t = 0; t < 9; ++t)                                                     \
            wt[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(TAPS, c4, (t * K + 32 * (ST)) * 4, 0)); \
        const float shift = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(TAPS, c4, (9 * K + 32 * (ST)) * 4, 0)); \
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
                    unsigned pk = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)v);              \
                    /* lo half = f16(v - hi) and the range guard's running maximum, in ONE ordered statement: as a plain */ \
                    /* fmaxf chain the compiler sums the maxima up at the end and keeps (spills) all 48 values until then */ \
                    asm volatile("v_fma_mixhi_f16 %0, %0, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_max_f32 %1, %1, %2" \
                                 : "+v"(pk), "+v"(rmax) : "v"(v));                                        \
                    const int rl = 24 * w + 4 * y + x;      /* row 48 fh + rl */                          \
                    if ((J) == 1) pend[rl] = pk;                                                          \
                    else out0[rl] = pk;                                                                   \
                }                                                                                         \
            }                                                                                             \
    }
    // ... and their publication: column block 0 (stage wc) always, column block 1 (stage wc + 8) when its slot exists now
#define CHIP_PUBLISH()                                                                                    \
    {                                                                                                     \
        int wbl = wb0;       /* the four swizzle variants of the address are formed per use: as values of the whole kernel */ \
        asm volatile("" : "+v"(wbl));                      /* they were spilled and reloaded behind vmcnt(0) waits */ \
        char* const slot0 = sm + wc * kChipSlotBytes;                                                     \
        _Pragma("unroll") for (int rl = 0; rl < 48; ++rl) CHIP_PUT(slot0, rl, out0[rl])                   \
        if (wc + 8 < NSLOT) {                     /* one wave-uniform branch */                           \
            char* const slot1 = sm + (wc + 8) * kChipSlotBytes;                                           \
            _Pragma("unroll") for (int rl = 0; rl < 48; ++rl) CHIP_PUT(slot1, rl, pend[rl])               \
        }                                                                                                 \
    }
#define CHIP_PUT(SLOT, RL, PK)                                                                            \
    {                                                                                                     \
        char* const p_ = (SLOT) + (wbl ^ ((((RL) >> 2) & 3) << 4)) + (RL) * 64;                           \
        *reinterpret_cast<unsigned short*>(p_) = (unsigned short)(PK);                                    \
        *reinterpret_cast<unsigned short*>(p_ + kChipHalfBytes) = (unsigned short)((PK) >> 16);           \
    }
    // taps [9][512] and shift [512] of a layer are one [10][512] table (engine.hip lays dw_b16 behind dw_w16; the launcher checks)
#define CHIP_TAPS_RSRC(DW_W) __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DW_W), 0, 10 * K * 4, 0x00020000)

    // ---------------------------------------------------------------------- layer 0: its depthwise reads the run's input
    // through a buffer resource that covers exactly this tile's valid rows: a row past the end of the batch (tail tile)
    // reads as zero, with no clamping code, and every load is resource + one per-lane offset register + a scalar row offset
    const long long rows_left = M - m0;
    const unsigned tile_bytes = (unsigned)(rows_left < 96 ? rows_left : 96) * (K * 4);
    {
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X) + (size_t)m0 * K, 0, tile_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t taps0 = CHIP_TAPS_RSRC(ch.dw_w[0]);
        const unsigned xo = (48u * fh * K) * 4 + c4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int st = j ? wc + 8 : wc;
            v2f in2[6][4];
#pragma unroll
            for (int y = 0; y < 6; ++y)
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    in2[y][x].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xo, ((4 * y + x) * K + 32 * st) * 4, 0));
                    in2[y][x].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xo, ((24 + 4 * y + x) * K + 32 * st) * 4, 0));
                }
            if (j == 0) CHIP_DW(0, st, taps0)
            else CHIP_DW(1, st, taps0)
            CHIP_TS()
            __builtin_amdgcn_sched_barrier(0);    // (the second block's 48 loads hoisted over the first's sums cost a spill)
        }
        CHIP_PUBLISH()
    }

    for (int li = 0; li < nl; ++li) {
        const _Float16* const Wfhi = chain_ptr<_Float16>(2, li);
        const _Float16* const Wflo = chain_ptr<_Float16>(3, li);
        // B fragments: column tile (wc, wc + 8), k16 step q -> ((tile * KQ + q) * 64 + lane) * 16 bytes: resource + lane * 16 in
        // one register + a scalar offset.  The first k16 step's are requested in FRONT of the barrier that publishes the A
        // operand (round 6): behind it all eight waves would wait out the same L2 round trip at once
        const __amdgpu_buffer_rsrc_t bhr = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(Wfhi), 0, K * K * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t blr = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(Wflo), 0, K * K * 2, 0x00020000);
        const int btile = wc * (KQ * 1024);       // bytes of a column tile's fragments: KQ k16 steps of 1 KB
        constexpr int jstep = 8 * KQ * 1024;      // bytes between column tiles wc and wc + 8
        f16x8 bh0[2], bl0[2], bh1[2], bl1[2];
#define CHIP_LB(R, B, Q, J) B[J] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(R, lane16, btile + (J) * jstep + (Q) * 1024, 0));
#define CHIP_BLOAD(BH, BL, Q) { CHIP_LB(bhr, BH, Q, 0) CHIP_LB(blr, BL, Q, 0) CHIP_LB(bhr, BH, Q, 1) CHIP_LB(blr, BL, Q, 1) }
        CHIP_BLOAD(bh0, bl0, 0)
        __syncthreads();                          // stages 0 .. NSLOT - 1 of layer li published
        CHIP_TS()

        // ------------------------------------------------------------------ 1 x 1 convolution of layer li
        // (a zero the compiler cannot form early: as plain constants the 96 clears were hoisted above the barrier into the
        //  depthwise phase, where the accumulators' registers are what its 96 results live in - 43 spills)
        float zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = zero;
        // ---- the K loop.  One stage = 6 steps (k16 step s = 0, 1 x row tile i = 0..2) of 6 MFMAs (2 in the plain-f16 mode).
        // Every memory request sits in its own MFMA gap, pinned there by a scheduling barrier (left alone the compiler
        // bunches the four B loads of a k16 step at the top of the stage - an in-order wave then issues no MFMA while they
        // go out - and sinks the A reads to just before their use):
        //   A fragments   a ring of three (hi, lo) pairs, step t uses pair t mod 3 and requests the pair of step t + 2 (the one
        //                 step t - 1 has finished with): two steps = 12 MFMAs between request and use, across stages too;
        //   B fragments   two sets (k16 step 0 / 1 of a stage); a register is requested again in the gap after the last MFMA
        //                 that reads it, three steps = 18 MFMAs before its next use.
#define CHIP_LAH(A, S, I) A = *reinterpret_cast<const f16x8*>(abase + (((I) == 1 ? ra1 : ra0) ^ ((S) << 5)) + ((I) == 2 ? 4096 + 64 : 0));
#define CHIP_LAL(A, S, I) A = *reinterpret_cast<const f16x8*>(abase + (((I) == 1 ? ra1 : ra0) ^ ((S) << 5)) + ((I) == 2 ? 4096 + 64 : 0) + kChipHalfBytes);
#define CHIP_MF(I, J, A, B) acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc[I][J], 0, 0, 0);
#define CHIP_SB __builtin_amdgcn_sched_barrier(0);
#define CHIP_NOP
        // a step: row tile I against both column tiles; M1 .. M6 = the request (or nothing) that follows each MFMA
#define CHIP_STEP(I, AH, AL, BH, BL, M1, M2, M3, M4, M5, M6)                                              \
    if constexpr (!PLAIN) {                                                                               \
        CHIP_MF(I, 0, AL, BH[0]) M1 CHIP_SB                                                               \
        CHIP_MF(I, 0, AH, BL[0]) M2 CHIP_SB                                                               \
    } else {                                                                                              \
        M1 M2                                                                                             \
    }                                                                                                     \
    CHIP_MF(I, 0, AH, BH[0]) M3 CHIP_SB                                                                   \
    if constexpr (!PLAIN) {                                                                               \
        CHIP_MF(I, 1, AL, BH[1]) M4 CHIP_SB                                                               \
        CHIP_MF(I, 1, AH, BL[1]) M5 CHIP_SB                                                               \
    } else {                                                                                              \
        M4 M5                                                                                             \
    }                                                                                                     \
    CHIP_MF(I, 1, AH, BH[1]) M6 CHIP_SB
#define CHIP_STAGES(FROM, TO)                                                                             \
    if ((FROM) < (TO)) {                                                                                  \
        f16x8 ah0, al0, ah1, al1, ah2, al2;                                                               \
        const char* abase = sm + ((FROM) < NSLOT ? (FROM) : (FROM) - NSLOT) * kChipSlotBytes;             \
        CHIP_LAH(ah0, 0, 0) CHIP_LAL(al0, 0, 0) CHIP_LAH(ah1, 0, 1) CHIP_LAL(al1, 0, 1)                   \
        _Pragma("nounroll") for (int kk = (FROM); kk < (TO); ++kk) {                                      \
            const int q1 = 2 * kk + 1, q2 = 2 * kk + 2 < KQ ? 2 * kk + 2 : 0;     /* (the last stage re-reads step 0: unused) */ \
            const char* const anext = sm + (kk + 1 < (TO) ? (kk + 1 < NSLOT ? kk + 1 : kk + 1 - NSLOT) : (kk < NSLOT ? kk : kk - NSLOT)) * kChipSlotBytes; \
            CHIP_STEP(0, ah0, al0, bh0, bl0, CHIP_LB(bhr, bh1, q1, 0), CHIP_LB(blr, bl1, q1, 0), CHIP_LB(bhr, bh1, q1, 1),  \
                      CHIP_LB(blr, bl1, q1, 1), CHIP_LAH(ah2, 0, 2), CHIP_LAL(al2, 0, 2))                 \
            CHIP_STEP(1, ah1, al1, bh0, bl0, CHIP_LAH(ah0, 1, 0), CHIP_LAL(al0, 1, 0), CHIP_NOP, CHIP_NOP, CHIP_NOP, CHIP_NOP) \
            CHIP_STEP(2, ah2, al2, bh0, bl0, CHIP_LAH(ah1, 1, 1), CHIP_LAL(al1, 1, 1), CHIP_LB(blr, bl0, q2, 0),            \
                      CHIP_LB(bhr, bh0, q2, 0), CHIP_NOP, CHIP_LB(blr, bl0, q2, 1))                       \
            CHIP_STEP(0, ah0, al0, bh1, bl1, CHIP_LB(bhr, bh0, q2, 1), CHIP_LAH(ah2, 1, 2), CHIP_LAL(al2, 1, 2), CHIP_NOP, CHIP_NOP, CHIP_NOP) \
            abase = anext;         /* the first two steps of the next stage (the same stage again behind the last: unused) */ \
            CHIP_STEP(1, ah1, al1, bh1, bl1, CHIP_LAH(ah0, 0, 0), CHIP_LAL(al0, 0, 0), CHIP_NOP, CHIP_NOP, CHIP_NOP, CHIP_NOP) \
            CHIP_STEP(2, ah2, al2, bh1, bl1, CHIP_LAH(ah1, 0, 1), CHIP_LAL(al1, 0, 1), CHIP_NOP, CHIP_NOP, CHIP_NOP, CHIP_NOP) \
        }                                                                                                 \
    }
        CHIP_STAGES(0, NPEND)
        CHIP_TS()
        if constexpr (NPEND > 0) {
            __syncthreads();                      // every wave has read stages 0 .. NPEND - 1: their slots are free
            CHIP_TS()
            if (wc >= 8 - NPEND) {
                char* const slot = sm + (wc + 8 - NSLOT) * kChipSlotBytes;
                int wbl = wb0;
                asm volatile("" : "+v"(wbl));
#pragma unroll
                for (int rl = 0; rl < 48; ++rl) CHIP_PUT(slot, rl, pend[rl])
            }
        }
        CHIP_TS()
        // ... and the other waves wait for them HERE, not at stage NSLOT: a wave that publishes beside a neighbour's
        // back-to-back MFMAs gets an LDS write through every ~90 cycles (9 000 cycles for its 96, measured) and everyone
        // waited for it ten stages later; with the whole workgroup at the barrier the 96 writes take ~1 000
        if constexpr (NPEND > 0) __syncthreads();  // pending stages published
        CHIP_TS()
        CHIP_STAGES(NPEND, 16)
        CHIP_TS()
#undef CHIP_STAGES
#undef CHIP_STEP
#undef CHIP_BLOAD
#undef CHIP_LB
#undef CHIP_LAH
#undef CHIP_LAL
#undef CHIP_MF
#undef CHIP_SB
#undef CHIP_NOP
        if (li + 1 == nl) break;

        // ------------------------------------------------------------------ depthwise of layer li + 1 on the accumulators
        const __amdgpu_buffer_rsrc_t taps = CHIP_TAPS_RSRC(chain_ptr<float>(0, li + 1));
        const __amdgpu_buffer_rsrc_t pur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(chain_ptr<float>(4, li)), 0, K * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t pbr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(chain_ptr<float>(5, li)), 0, K * 4, 0x00020000);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int st = j ? wc + 8 : wc;       // stage of layer li + 1 = column block of layer li
            const float u = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pur, c4, 128 * st, 0));
            const float b = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pbr, c4, 128 * st, 0));
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
            if (j == 0) CHIP_DW(0, st, taps)
            else CHIP_DW(1, st, taps)
            CHIP_TS()
        }
        // The barrier that frees the ring stands HERE, behind the depthwise: the two matrix waves of a SIMD do not finish a
        // layer's K loop together (the older one gets the pipe first), and everything above needs only the wave's own
        // accumulators - so the early wave's vector work runs beside the late wave's last MFMAs, and the late wave's has the
        // vector unit to itself, instead of both waiting and then sharing it.
        __syncthreads();                          // every wave has read every stage: the ring is free for this layer's tile
        CHIP_TS()
        CHIP_PUBLISH()
    }
#undef CHIP_DW
#undef CHIP_PUBLISH
#undef CHIP_TAPS_RSRC
#undef CHIP_PUT

    // ---------------------------------------------------------------------- the run's output: bias + ReLU from the accumulators
    // (stores past the tile's valid rows are dropped by the resource's range check: no per-store masks)
    {
        const __amdgpu_buffer_rsrc_t pur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(chain_ptr<float>(4, nl - 1)), 0, K * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t pbr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(chain_ptr<float>(5, nl - 1)), 0, K * 4, 0x00020000);
        if constexpr (!NDW) {
            const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(Y + (size_t)m0 * K, 0, tile_bytes, 0x00020000);
            const unsigned yo = (4u * fh * K) * 4 + c4;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int st = j ? wc + 8 : wc;
                const float u = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pur, c4, 128 * st, 0));
                const float b = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pbr, c4, 128 * st, 0));
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(fmaf(acc[i][j][r], u, b), 0.0f)), yrs, yo,
                                                              ((32 * i + 8 * (r >> 2) + (r & 3)) * K + 32 * st) * 4, 0);
            }
        } else {
            // the tile's windows are rows m0 / 24 .. + 3 of the [windows][6][512] output: 6 output rows per window, a quarter
            // of the tile's bytes
            const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(Y + (size_t)(m0 / 4) * K, 0, tile_bytes / 4, 0x00020000);
            _Float16* const yh = reinterpret_cast<_Float16*>(Y);
            const __amdgpu_buffer_rsrc_t yhr = __builtin_amdgcn_make_buffer_rsrc(yh + (size_t)(m0 / 4) * K, 0, tile_bytes / 8, 0x00020000);
            const __amdgpu_buffer_rsrc_t ylr = __builtin_amdgcn_make_buffer_rsrc(yh + (size_t)(M / 4 + m0 / 4) * K, 0, tile_bytes / 8, 0x00020000);
            const __amdgpu_buffer_rsrc_t ntaps = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ch.ndw_w), 0, 10 * K * 4, 0x00020000);
            const unsigned yo = (12u * fh * K) * 4 + c4;            // this lane's windows 2 fh, 2 fh + 1: output rows 12 fh ..
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int st = j ? wc + 8 : wc;
                const float u = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pur, c4, 128 * st, 0));
                const float b = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pbr, c4, 128 * st, 0));
                float wt[9];
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    wt[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ntaps, c4, (t * K + 32 * st) * 4, 0));
                const float shift = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ntaps, c4, (9 * K + 32 * st) * 4, 0));
                float ev[12][4];
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        const float a01 = fmaxf(fmaf(acc[q >> 2][j][4 * (q & 3) + x], u, b), 0.0f);
                        const float a23 = fmaxf(fmaf(acc[(q + 6) >> 2][j][4 * ((q + 6) & 3) + x], u, b), 0.0f);
                        const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a01),
                                                                        __builtin_bit_cast(unsigned, a23), false, false);
                        ev[2 * q][x] = __builtin_bit_cast(float, (unsigned)r[0]);
                        ev[2 * q + 1][x] = __builtin_bit_cast(float, (unsigned)r[1]);
                    }
#pragma unroll
                for (int oy = 0; oy < 3; ++oy)
#pragma unroll
                    for (int ox = 0; ox < 2; ++ox) {
                        v2f a = {shift, shift};
#pragma unroll
                        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                            for (int kw = 0; kw < 3; ++kw) {
                                const int iy = 2 * oy + kh, ix = 2 * ox + kw;
                                if (iy >= 6 || ix >= 4) continue;            // SAME padding of a stride-2 layer: one row / column behind the map
                                a = __builtin_elementwise_fma(v2f{ev[iy][ix], ev[6 + iy][ix]}, v2f{wt[kh * 3 + kw], wt[kh * 3 + kw]}, a);
                            }
#pragma unroll
                        for (int w = 0; w < 2; ++w) {
                            const float o = fmaxf(w ? a.y : a.x, 0.0f);
                            if constexpr (PLANES) {
                                unsigned pk = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)o);
                                asm volatile("v_fma_mixhi_f16 %0, %0, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_max_f32 %1, %1, |%2|"
                                             : "+v"(pk), "+v"(rmax) : "v"(o));
                                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)pk, yhr, yo >> 1, ((6 * w + 2 * oy + ox) * K + 32 * st) * 2, 0);
                                if constexpr (!PLAIN)
                                    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(pk >> 16), ylr, yo >> 1,
                                                                          ((6 * w + 2 * oy + ox) * K + 32 * st) * 2, 0);
                            } else {
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), yrs, yo,
                                                                      ((6 * w + 2 * oy + ox) * K + 32 * st) * 4, 0);
                            }
                        }
                    }
            }
        }
    }
    if (range_flag && !(rmax <= kF16MaxChip)) *range_flag = 1u;
    if constexpr (TRACE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CHIP_TS()
#undef CHIP_TS
}

constexpr int kMaxDevicesChip = 64;

template <bool PLAIN, bool NDW, bool PLANES = false>
void launch_chip(const float* in, float* out, const SepLayer* L, int nl, long long M, const float* ndw_w, hipStream_t stream) {
    ChipChain ch{};
    for (int i = 0; i < nl; ++i) {
        ch.dw_w[i] = dw_w_of(L[i]);
        ch.dw_b[i] = dw_b_of(L[i]);
        ch.whi[i] = static_cast<const _Float16*>(L[i].pw_fhi);
        ch.wlo[i] = static_cast<const _Float16*>(L[i].pw_flo);
        ch.pw_u[i] = L[i].pw_u;
        ch.pw_b[i] = L[i].pw_b;
    }
    ch.ndw_w = ndw_w;
    constexpr int NSLOT = 13;
    constexpr int lds = NSLOT * kChipSlotBytes;
    static_assert(lds <= 160 * 1024, "the ring must fit the CU's LDS");
    static std::once_flag once[kMaxDevicesChip];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & (kMaxDevicesChip - 1)], [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sep_chip_kernel<PLAIN, NSLOT, false, NDW, PLANES>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    });
    const long long tiles = (M + 95) / 96;
    int tune = 0;
#ifdef BD_KERNEL_TRACE      // developer build only: BD_CHIP_TUNE = policy under test; BD_WS_TRACE=7 stamps workgroup 0
    if (const char* tn = getenv("BD_CHIP_TUNE")) tune = atoi(tn);
    const char* tr = getenv("BD_WS_TRACE");
    if (tr && tr[0] == '7' && !PLANES) {
        static unsigned long long* dbg = nullptr;
        static int shots = 0;
        if (!dbg) (void)hipMalloc(&dbg, 128 * 8);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sep_chip_kernel<PLAIN, NSLOT, true, NDW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipMemsetAsync(dbg, 0, 128 * 8, stream);
        hipLaunchKernelGGL((sep_chip_kernel<PLAIN, NSLOT, true, NDW>), dim3((unsigned)tiles), dim3(512), lds, stream, ch, in, out, nl,
                           M, L[0].range_flag, dbg, tune);
        (void)hipStreamSynchronize(stream);
        unsigned long long h[128];
        (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        if (++shots == 8)
            for (int w = 0; w < 2; ++w) {
                fprintf(stderr, "[trace] on-chip run of %d layers, wave %d: cycles between stamps (start | dw0 j0 j1 | per layer: B1, stages 0-2, B2, pend, "
                                "B3, stages 3-15, [dw j0, dw j1, B4] | stores):", nl, w ? 5 : 0);
                for (int i = 1; i < 64 && h[w * 64 + i]; ++i) fprintf(stderr, " %llu", h[w * 64 + i] - h[w * 64 + i - 1]);
                fprintf(stderr, "\n");
            }
        return;
    }
#endif
    hipLaunchKernelGGL((sep_chip_kernel<PLAIN, NSLOT, false, NDW, PLANES>), dim3((unsigned)tiles), dim3(512), lds, stream, ch, in, out, nl, M,
                       L[0].range_flag, (unsigned long long*)nullptr, tune);
}

}  // namespace

// A run of stride-1 512 -> 512 layers on the 6 x 4 map with the tiles between its layers kept on the CU: reads `in`, writes
// `out`.  They may be the same buffer: a workgroup has read all rows of its tile before it writes any.  The caller has
// checked the shapes (launch_separable_run_next_dw).  With `next` (the stride-2 layer behind the run) the run's output is not written:
// next's depthwise is applied in the epilogue and out = [windows][3][2][512] (then `out` must not be `in`: the tiles' rows
// differ).  False (nothing launched) when a layer's shift table does not follow its taps (the kernel reads both through
// one [10][512] resource; engine.hip lays dw_b16 behind dw_w16).
// planes (with next): the output leaves as f16 hi / lo planes [windows * 6][512] (hi at out, lo behind it), what septail.hip reads.
bool launch_separable_chip(const float* in, float* out, int windows, const SepLayer* L, int nl, hipStream_t stream,
                           const SepLayer* next, bool planes) {
    if (planes && !next) return false;
    if (nl < 1 || nl > kChipMaxLayers) return false;
    for (int i = 0; i < nl; ++i)
        if (dw_b_of(L[i]) != dw_w_of(L[i]) + 9 * 512) return false;
    if (next && (dw_b_of(*next) != dw_w_of(*next) + 9 * 512 || next->cin != 512 || next->stride != 2 || in == out)) return false;
    const long long M = (long long)windows * 24;
    const float* nw = next ? dw_w_of(*next) : nullptr;
    if (planes) {
        if (L[0].pw_mode == 2) launch_chip<true, true, true>(in, out, L, nl, M, nw, stream);
        else launch_chip<false, true, true>(in, out, L, nl, M, nw, stream);
        return true;
    }
    if (L[0].pw_mode == 2) {
        if (next) launch_chip<true, true>(in, out, L, nl, M, nw, stream);
        else launch_chip<true, false>(in, out, L, nl, M, nw, stream);
    } else {
        if (next) launch_chip<false, true>(in, out, L, nl, M, nw, stream);
        else launch_chip<false, false>(in, out, L, nl, M, nw, stream);
    }
    return true;
}

}  // namespace bd
