// The tail of YAMNet as two launches of ONE matrix kernel (round 6): pointwise 13 (512 -> 1024 on the 3 x 2 map) with layer
// 14's depthwise in its epilogue, and pointwise 14 (1024 -> 1024) with the average pool in its epilogue (yamnet.py:91-92,104).
//
// Until round 6: sep_ws_kernel<0, 1> (24.9 us per 1024 windows) and the 12-wave kernel of rounds 2-5 (61.1 us).  The second ran
// 128 workgroups on 256 CUs - its 96 x 512 tiles are all the 6144 x 1024 output has - with two matrix waves and a depthwise
// producer per SIMD in step behind one barrier per 32 input channels: 3 300 cycles per stage for 2 304 of matrix work.
//
// Here every SIMD of the chip owns ONE 96 x 64 wave tile of the output (6144 x 1024 = 1024 wave tiles = 256 CUs x 4 SIMDs) and
// nothing else runs beside it:
//   * a workgroup is 4 waves on a 96 x 256 tile; wave w owns columns 64 w .. 64 w + 63: 3 x 2 tiles of
//     v_mfma_f32_32x32x16_f16, 96 accumulator registers (one wave per SIMD: 512 registers are there to be used);
//   * the A operand arrives as what the matrix instruction reads: two f16 planes (hi, lo) [rows][K], written by the kernel in
//     front (sep_chip_kernel's depthwise-13 epilogue / this kernel's own depthwise-14 epilogue).  A stage is 64 input channels
//     (24 KB); the four waves bring it in by LDS-DMA (buffer_load_dwordx4 ... lds, six per wave and stage), three stages ahead,
//     into a ring of four.  No wave converts, publishes or computes a depthwise inside the K loop;
//   * one workgroup barrier per stage, in the MIDDLE of the stage before: behind it stage t + 1 has landed for every wave and
//     every wave has left stage t - 1, whose slot takes the DMA of stage t + 3 (its six instructions spread over the next step).  The fragments of the next k16 step are
//     always requested a step ahead, across stage ends too;
//   * B fragments straight from the fragment-ordered weights (L2) into registers, one stage (four k16 steps) ahead;
//   * which row of the tile an LDS row holds is free (a DMA lane reads any address): LDS row 32 i + 8 q + 4 h + e holds tile
//     row 48 h + 16 i + 4 q + e, so that lane (column c, half h) finds in its 48 accumulators, in order, the six positions of
//     windows 8 h .. 8 h + 7 of its channel.  Depthwise 14 (3 x 3 on the 3 x 2 map) and the average pool then run on
//     registers with compile-time neighbours - no swap, no LDS, no padding test.
//   * workgroup -> tile: the XCDs take workgroup IDs round-robin; XCD x works on column tiles 2 (x & 1), 2 (x & 1) + 1 (2 MB of
//     weights in its L2) and on a quarter of the row tiles, the two column tiles of a row tile next to each other in time.
//
// Arithmetic per element is that of the kernels it replaces (depthwise_kernel, pointwise_f16x3_kernel, pool_head_kernel): per
// accumulator the products lo*hi, hi*lo, hi*hi of k16 step q = 0 .. K/16 - 1 in ascending order, relu(fma(acc, u, b)); the
// depthwise = shift, then the taps inside the map in row-major order with fmaf, ReLU, the range guard's maximum, the split;
// the pool = the six positions summed in order, divided by 6.  (Taps outside the map are skipped: sepchip.hip on why that is
// the zero-multiplying tap's result.)
#include "bd_internal.h"

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace bd {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr float kF16MaxTail = 65504.0f;
constexpr int kTailN = 1024;                   // output channels of both layers
constexpr int kTailPlane = 96 * 128;           // one f16 half of a stage: 96 rows x 64 input channels
constexpr int kTailStage = 2 * kTailPlane;     // 24 576
constexpr int kTailRing = 4;                   // stages in LDS: the DMA runs kTailRing - 1 stages ahead (3: measures the same)
constexpr int kTailLds = kTailRing * kTailStage;   // 98 304

struct TailArgs {
    const _Float16 *ahi, *alo;                 // A planes [M][K]
    const _Float16 *bhi, *blo;                 // weights in MFMA B-fragment order [1024 / 32][K / 16][64][8]
    const float *pu, *pb;                      // epilogue factor / shift per output channel
    const float* taps;                         // depthwise epilogue: [9][1024] taps * 2^act_exp, [1024] shift behind them
    _Float16 *ohi, *olo;                       // depthwise epilogue: output planes [M][1024]
    float* pooled;                             // pool epilogue: [windows][1024]
};

#define TAIL_RSRC(P, BYTES) __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(P)), 0, (BYTES), 0x00020000)

// EPI 0: + depthwise 14, written as planes; EPI 1: + average pool
template <int K, int EPI, bool PLAIN, bool TRACE = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void tail_gemm_kernel(const TailArgs a, int M, int windows,
                                                                                                 unsigned* __restrict__ range_flag,
                                                                                                 unsigned long long* __restrict__ dbg = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    // developer aid (-DBD_KERNEL_TRACE build, BD_WS_TRACE=9): shader-clock stamps of wave 0 of workgroups 0 and 101
    int tsn = 0;
#define TAIL_TS()                                                                                         \
    if constexpr (TRACE) {                                                                                \
        if ((blockIdx.x == 0 || blockIdx.x == 101) && threadIdx.x == 0 && tsn < 32)                       \
            dbg[(blockIdx.x ? 32 : 0) + tsn] = __builtin_amdgcn_s_memtime();                              \
        ++tsn;                                                                                            \
    }
    if constexpr (TRACE) {
        if (blockIdx.x == 0 && threadIdx.x == 0) dbg[30] = wall_clock64();
    }
    TAIL_TS()
    constexpr int N = kTailN, NST = K / 64, KQ = K / 16;
    constexpr int NB = PLAIN ? 2 : 4;          // B loads of a k16 step
    constexpr int ND = PLAIN ? 3 : 6;          // DMA instructions of a wave and stage
    const int tid = threadIdx.x, lane = tid & 63;
    const int wc = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fh = lane >> 5;
    // tile of this workgroup (header): XCD x = ID & 7
    const int xcd = blockIdx.x & 7, kx = blockIdx.x >> 3;
    const int rt = 4 * (kx >> 1) + (xcd >> 1);
    const int m0 = rt * 96;
    if (m0 >= M) return;
    const int n0 = (2 * (xcd & 1) + (kx & 1)) * 256;
    const int x_cnt = M - m0 < 96 ? M - m0 : 96;

    // ---- DMA: lane L = 64 c + lane of a plane's 768 writes LDS bytes 16 L ..: row rho = L >> 3, slot L & 7
    const __amdgpu_buffer_rsrc_t ahr = TAIL_RSRC(a.ahi, (unsigned)M * K * 2), alr = TAIL_RSRC(a.alo, (unsigned)M * K * 2);
    unsigned dvo[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int L = (3 * wc + t) * 64 + lane;
        const int rho = L >> 3, ls = (L & 7) ^ ((rho >> 1) & 7);
        const int g = 48 * ((rho >> 2) & 1) + 16 * (rho >> 5) + 4 * ((rho >> 3) & 3) + (rho & 3);
        int row = m0 + g;
        row = row < M ? row : M - 1;           // rows past the tile's last: any valid row (their outputs are dropped)
        dvo[t] = (unsigned)row * (K * 2) + ls * 16;
    }
    // piece t of stage s: this wave's 1 KB of the hi plane (lo = false) or the lo plane
    auto dma1 = [&](int s, int t, bool lo) {
        char* const base = sm + (s % kTailRing) * kTailStage + 3 * wc * 1024 + t * 1024;
        if (lo) __builtin_amdgcn_raw_ptr_buffer_load_lds(alr, (__attribute__((address_space(3))) void*)(base + kTailPlane), 16, dvo[t], s * 128, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(ahr, (__attribute__((address_space(3))) void*)base, 16, dvo[t], s * 128, 0, 0);
    };
    auto dma = [&](int s) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            dma1(s, t, false);
            if constexpr (!PLAIN) dma1(s, t, true);
        }
    };
    // ---- A fragments: lane (frow, fh), row tile i, k16 step s of a stage: row 32 i + frow, slot (2 s + fh) ^ ((frow >> 1) & 7) - two
    //      128-byte rows share the 64 banks, and a ds_read_b128 serves the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32):
    //      with that key the sixteen rows of a group lie on sixteen different 16-byte slots (row & 7: eight, 2 x the LDS cycles);
    //      two bases per step (ring slots 0-1 / 2-3: a ds_read's immediate offset ends at 64 KB)
    unsigned aro[2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        aro[0][s] = frow * 128 + (((2 * s + fh) ^ ((frow >> 1) & 7)) << 4);
        aro[1][s] = aro[0][s] + 2 * kTailStage;
    }
    // ---- B fragments.  The wave's two column tiles are the EVEN (j = 0) and the ODD (j = 1) channels of its 64, so that a lane's
    //      two channels are neighbours in memory (packed stores in the epilogue): lane (frow, fh) of tile j supplies column
    //      c = 2 frow + j of the 64, i.e. lane (c & 31, fh) of the stored fragment of column tile n0 / 32 + 2 wc + (c >> 5)
    const __amdgpu_buffer_rsrc_t bhr = TAIL_RSRC(a.bhi, N * K * 2), blr = TAIL_RSRC(a.blo, N * K * 2);
    const int sb0 = (n0 / 32 + 2 * wc) * KQ * 1024;
    unsigned bvo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bvo[j] = (frow >> 4) * (KQ * 1024) + ((((2 * frow + j) & 31) + 32 * fh) << 4);
    f16x8 bh[4][2], bl[4][2];
    // epilogue constants of this lane's two channels, requested now
    const int ch0 = n0 + 64 * wc + 2 * frow;       // ... and ch0 + 1
    const v2f u2 = *reinterpret_cast<const v2f*>(a.pu + ch0), b2 = *reinterpret_cast<const v2f*>(a.pb + ch0);
    v2f wt2[9], shift2;
    if constexpr (EPI == 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t) wt2[t] = *reinterpret_cast<const v2f*>(a.taps + t * N + ch0);
        shift2 = *reinterpret_cast<const v2f*>(a.taps + 9 * N + ch0);
    }

    dma(0);
    asm volatile("" ::: "memory");             // (the count below: everything behind stage 0's DMA stays behind it)
#pragma unroll
    for (int s = 1; s < kTailRing - 1; ++s) dma(s);
#define TAIL_BLOAD(S, J, Q)                                                                               \
    {                                                                                                     \
        bh[S][J] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(bhr, bvo[J], sb0 + (Q) * 1024, 0)); \
        if constexpr (!PLAIN)                                                                             \
            bl[S][J] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(blr, bvo[J], sb0 + (Q) * 1024, 0)); \
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        TAIL_BLOAD(s, 0, s)
        TAIL_BLOAD(s, 1, s)
    }
    f32x16 acc[3][2];
    {
        float zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = zero;
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NB + (kTailRing - 2) * ND) : "memory");      // stage 0 has landed (this wave's part)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    TAIL_TS()

    // The K loop, unrolled, in the order it is written (a scheduling barrier behind every matrix instruction: left to itself the
    // compiler moves every load to just in front of its use).  k16 step g: the matrix instructions of one product at a time -
    // an accumulator's own order is lo*hi, hi*lo, hi*hi, and two instructions on one accumulator are five others apart; behind
    // the first ones, one memory instruction each: the A fragments of step g + 1 (other register set), then the B fragments of
    // step g + 3 into the registers step g - 1 has left.
    f16x8 ah[2][3], al[2][3];
#define TAIL_ALOAD(BUF, G, I, LO)                                                                         \
    {                                                                                                     \
        const char* const p_ = sm + aro[(((G) >> 2) % kTailRing) >> 1][(G) & 3] + (((((G) >> 2) % kTailRing) & 1) * kTailStage + (I) * 4096 + (LO) * kTailPlane); \
        if constexpr (LO) al[BUF][I] = *reinterpret_cast<const f16x8*>(p_);                               \
        else ah[BUF][I] = *reinterpret_cast<const f16x8*>(p_);                                            \
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        TAIL_ALOAD(0, 0, i, 0)
        if constexpr (!PLAIN) TAIL_ALOAD(0, 0, i, 1)
    }
    __builtin_amdgcn_sched_barrier(0);
    constexpr int NMF = PLAIN ? 6 : 18;        // matrix instructions of a step
    constexpr int NAL = PLAIN ? 3 : 6;         // A loads of a step
#pragma unroll
    for (int g = 0; g < KQ; ++g) {
        const int cur = g & 1, nxt = cur ^ 1, bs = g & 3, br = (g + 3) & 3;
        const bool la = g + 1 < KQ, lb = g >= 1 && g + 3 < KQ;
#pragma unroll
        for (int k = 0; k < NMF; ++k) {
            const int p = PLAIN ? 2 : k / 6, i = (k % 6) >> 1, j = k & 1;
            if (p == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur][i], bh[bs][j], acc[i][j], 0, 0, 0);
            else if (p == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][i], bl[bs][j], acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][i], bh[bs][j], acc[i][j], 0, 0, 0);
            if (k < NAL) {
                if (la) {
                    if constexpr (PLAIN) {
                        TAIL_ALOAD(nxt, g + 1, k, 0)
                    } else {
                        if (k & 1) TAIL_ALOAD(nxt, g + 1, k >> 1, 1)
                        else TAIL_ALOAD(nxt, g + 1, k >> 1, 0)
                    }
                }
            } else if (k < NAL + 2) {
                if (lb) TAIL_BLOAD(br, k - NAL, g + 3)
            } else if ((g & 3) == 2 && (g >> 2) + kTailRing - 1 < NST) {
                // the DMA of stage t + 3 (its slot is free since the barrier behind step 1), a piece behind each of the next matrix
                // instructions: issued in one go behind the barrier by all four waves, they held the matrix pipe up
                if constexpr (PLAIN) {
                    if (k == NAL + 2) dma((g >> 2) + kTailRing - 1);
                } else if (k < NAL + 2 + ND) {
                    dma1((g >> 2) + kTailRing - 1, (k - NAL - 2) >> 1, (k - NAL - 2) & 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if ((g & 3) == 1 && (g >> 2) + 1 < NST) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NB) : "memory");       // this wave's part of the next stage has landed
            __builtin_amdgcn_s_barrier();                                       // ... everybody's; and the stage before is read
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (TRACE) {
            if ((g & 3) == 3) { TAIL_TS() }
        }
    }
#undef TAIL_ALOAD
#undef TAIL_BLOAD

    // ---------------------------------------------------------------------- epilogue: the lane's accumulators in order are
    // positions 0 .. 5 of windows 8 fh .. 8 fh + 7 (tile rows 48 fh + n), channels ch0 and ch0 + 1
    v2f yv[48];
#pragma unroll
    for (int n = 0; n < 48; ++n) {
        const v2f c2 = {acc[n >> 4][0][n & 15], acc[n >> 4][1][n & 15]};
        const v2f y = __builtin_elementwise_fma(c2, u2, b2);
        yv[n] = v2f{fmaxf(y.x, 0.0f), fmaxf(y.y, 0.0f)};
    }
    if constexpr (EPI == 1) {
        const __amdgpu_buffer_rsrc_t pr = TAIL_RSRC(a.pooled, (unsigned)windows * N * 4);
        const unsigned po = (unsigned)((m0 / 6 + 8 * fh) * N + ch0) * 4;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            float sx = yv[6 * w].x, sy = yv[6 * w].y;
#pragma unroll
            for (int p = 1; p < 6; ++p) {
                sx += yv[6 * w + p].x;
                sy += yv[6 * w + p].y;
            }
            const float six = 6.0f;
            sx /= six;
            sy /= six;
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, sx), __builtin_bit_cast(unsigned, sy)}, pr, po, w * N * 4, 0);
        }
    } else {
        float rmax = 0.0f;
        const __amdgpu_buffer_rsrc_t ohr = TAIL_RSRC(a.ohi + (size_t)m0 * N, (unsigned)x_cnt * N * 2);
        const __amdgpu_buffer_rsrc_t olr = TAIL_RSRC(a.olo + (size_t)m0 * N, (unsigned)x_cnt * N * 2);
        const unsigned oo = (unsigned)(48 * fh * N + ch0) * 2;
#pragma unroll
        for (int w = 0; w < 8; ++w)
#pragma unroll
            for (int oy = 0; oy < 3; ++oy)
#pragma unroll
                for (int ox = 0; ox < 2; ++ox) {
                    v2f s2 = shift2;
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const int iy = oy + kh - 1, ix = ox + kw - 1;
                            if (iy < 0 || iy >= 3 || ix < 0 || ix >= 2) continue;
                            s2 = __builtin_elementwise_fma(yv[6 * w + 2 * iy + ix], wt2[kh * 3 + kw], s2);
                        }
                    // (o.x, o.y) -> hi halves packed, lo halves packed, the range guard's running maximum: one ordered statement
                    const float ox_ = fmaxf(s2.x, 0.0f), oy_ = fmaxf(s2.y, 0.0f);
                    unsigned hi2, lo2;
                    asm volatile("v_cvt_pk_f16_f32 %0, %3, %4\n\t"
                                 "v_fma_mixlo_f16 %1, %0, -1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                                 "v_fma_mixhi_f16 %1, %0, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                                 "v_max3_f32 %2, %2, |%3|, |%4|"
                                 : "=&v"(hi2), "=&v"(lo2), "+v"(rmax)
                                 : "v"(ox_), "v"(oy_));
                    const int so = (6 * w + 2 * oy + ox) * N * 2;
                    __builtin_amdgcn_raw_buffer_store_b32(hi2, ohr, oo, so, 0);
                    if constexpr (!PLAIN) __builtin_amdgcn_raw_buffer_store_b32(lo2, olr, oo, so, 0);
                }
        if (range_flag && !(rmax <= kF16MaxTail)) *range_flag = 1u;
    }
    TAIL_TS()
    if constexpr (TRACE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TAIL_TS()
        if (blockIdx.x == 0 && threadIdx.x == 0) dbg[31] = wall_clock64();
    }
#undef TAIL_TS
}

// ---------------------------------------------------------------------------------------------------------------- exact-f32 mode
// The same kernel on v_mfma_f32_32x32x2_f32 (bd_set_pointwise_mode 0).  What differs: the A operand is the f32 activation itself
// ([rows][K], what sep_chip_f32_kernel and this kernel's own depthwise epilogue write), a stage is 64 k of it = 256-byte rows (the
// same 24 KB), one ds_read_b128 per row tile and super-step of 8 k, B fragments from the weights in the f32 instruction's fragment
// order (SepLayer::pw_ffrag, even / odd channel tiles by lane address as above), three super-steps ahead; a super-step is 24 matrix
// instructions of 64 cycles.  Per accumulator the k pairs {8 s + e, 8 s + 4 + e}, e = 0..3, of super-step s in ascending order
// (pointwise_kernel's operand map), then acc + shift, ReLU: the bits of pointwise_kernel with the next depthwise / the pool in its
// epilogue, which these two launches replace (0.61 / 0.71 of the f32 matrix peak: a 96 x 128 tile per 4 waves, its A and B tiles
// through LDS behind two barriers per 32 k).
struct TailArgsF32 {
    const float* a;                            // A [M][K]
    const float* bfrag;                        // weights in fragment order [1024 / 32][K / 8][64][4]
    const float* pb;                           // shift per output channel
    const float* taps;                         // depthwise epilogue: [9][1024] taps; shift: the layer's dw_b
    const float* tshift;
    float* out;                                // depthwise epilogue: [M][1024]
    float* pooled;                             // pool epilogue: [windows][1024]
};

template <int K, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void tail_gemm_f32_kernel(const TailArgsF32 a, int M, int windows) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    typedef float v4f __attribute__((ext_vector_type(4)));
    constexpr int N = kTailN, NST = K / 64, KS = K / 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wc = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fh = lane >> 5;
    const int xcd = blockIdx.x & 7, kx = blockIdx.x >> 3;
    const int rt = 4 * (kx >> 1) + (xcd >> 1);
    const int m0 = rt * 96;
    if (m0 >= M) return;
    const int n0 = (2 * (xcd & 1) + (kx & 1)) * 256;
    const int x_cnt = M - m0 < 96 ? M - m0 : 96;
    // ---- DMA: lane L = 64 c + lane of a stage's 1536 writes LDS bytes 16 L ..: row rho = L >> 4, slot L & 15 (XORed with rho & 15)
    const __amdgpu_buffer_rsrc_t ar = TAIL_RSRC(a.a, (unsigned)M * K * 4);
    unsigned dvo[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int L = (6 * wc + t) * 64 + lane;
        const int rho = L >> 4, ls = (L & 15) ^ (rho & 15);
        const int g = 48 * ((rho >> 2) & 1) + 16 * (rho >> 5) + 4 * ((rho >> 3) & 3) + (rho & 3);
        int row = m0 + g;
        row = row < M ? row : M - 1;
        dvo[t] = (unsigned)row * (K * 4) + ls * 16;
    }
    auto dma1 = [&](int s, int t) {
        char* const base = sm + (s % kTailRing) * kTailStage + (6 * wc + t) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ar, (__attribute__((address_space(3))) void*)base, 16, dvo[t], s * 256, 0, 0);
    };
    auto dma = [&](int s) {
#pragma unroll
        for (int t = 0; t < 6; ++t) dma1(s, t);
    };
    // ---- A fragments: lane (frow, fh), row tile i, super-step s of a stage: row 32 i + frow, slot (2 s + fh) ^ (frow & 15)
    unsigned aro[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) aro[s] = frow * 256 + (((2 * s + fh) ^ (frow & 15)) << 4);
    const __amdgpu_buffer_rsrc_t br = TAIL_RSRC(a.bfrag, N * K * 4);
    const int sb0 = (n0 / 32 + 2 * wc) * KS * 1024;
    unsigned bvo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bvo[j] = (frow >> 4) * (KS * 1024) + ((((2 * frow + j) & 31) + 32 * fh) << 4);
    const int ch0 = n0 + 64 * wc + 2 * frow;
    const v2f b2 = *reinterpret_cast<const v2f*>(a.pb + ch0);
    v2f wt2[9], shift2;
    if constexpr (EPI == 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t) wt2[t] = *reinterpret_cast<const v2f*>(a.taps + t * N + ch0);
        shift2 = *reinterpret_cast<const v2f*>(a.tshift + ch0);
    }
    dma(0);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int s = 1; s < kTailRing - 1; ++s) dma(s);
    v4f bf[4][2];
#define TAIL_BLOAD32(S, J, Q) bf[S][J] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(br, bvo[J], sb0 + (Q) * 1024, 0));
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        TAIL_BLOAD32(s, 0, s)
        TAIL_BLOAD32(s, 1, s)
    }
    f32x16 acc[3][2];
    {
        float zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = zero;
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + (kTailRing - 2) * 6) : "memory");      // stage 0 has landed (this wave's part)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    v4f av[2][3];
    // fragments of super-step S of the stage in ring slot SLOT (run time)
#define TAIL_ALOAD32(BUF, SLOT, S, I) \
    av[BUF][I] = *reinterpret_cast<const v4f*>(sm + aro[S] + (SLOT) * kTailStage + (I) * 8192);
#pragma unroll
    for (int i = 0; i < 3; ++i) TAIL_ALOAD32(0, 0, 0, i)
    __builtin_amdgcn_sched_barrier(0);
    // A stage = eight super-steps, written out (a run-time loop over the stages: 3 072 matrix instructions do not unroll).  Super-
    // step s: element e of every fragment in turn (an accumulator's instructions are six others apart); behind the first matrix
    // instructions one memory instruction each: A of the next super-step, B of the third next (ring of four), and - in the super-
    // step behind the stage's barrier - the six DMA instructions of stage t + 3.  LAST: the stage behind which nothing follows.
    auto stage = [&](int t, auto last_c) {
        constexpr bool LAST = decltype(last_c)::value;
        const int slot = t & (kTailRing - 1), slotn = (t + 1) & (kTailRing - 1);
        const bool dm = t + kTailRing - 1 < NST;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int cur = s & 1, nxt = cur ^ 1, bs = s & 3, brl = (s + 3) & 3;
            const bool la = !LAST || s < 7, lb = !LAST || s < 5;
#pragma unroll
            for (int k = 0; k < 24; ++k) {
                const int e = k / 6, i = (k % 6) >> 1, j = k & 1;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i][e], bf[bs][j][e], acc[i][j], 0, 0, 0);
                if (k < 3) {
                    if (la) {
                        if (s < 7) { TAIL_ALOAD32(nxt, slot, s + 1, k) } else { TAIL_ALOAD32(nxt, slotn, 0, k) }
                    }
                } else if (k < 5) {
                    if (lb) TAIL_BLOAD32(brl, k - 3, 8 * t + s + 3)
                } else if (k < 11 && s == 4 && !LAST) {
                    if (dm) dma1(t + kTailRing - 1, k - 5);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (s == 3 && !LAST) {
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                // this wave's part of the next stage has landed
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
#pragma unroll 1
    for (int t = 0; t + 1 < NST; ++t) stage(t, std::false_type{});
    stage(NST - 1, std::true_type{});
#undef TAIL_ALOAD32
#undef TAIL_BLOAD32
    // ---- epilogue (as above): positions 0 .. 5 of windows 8 fh .. 8 fh + 7, channels ch0, ch0 + 1
    v2f yv[48];
#pragma unroll
    for (int n = 0; n < 48; ++n) {
        const v2f y = v2f{acc[n >> 4][0][n & 15], acc[n >> 4][1][n & 15]} + b2;
        yv[n] = v2f{fmaxf(y.x, 0.0f), fmaxf(y.y, 0.0f)};
    }
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    if constexpr (EPI == 1) {
        const __amdgpu_buffer_rsrc_t pr = TAIL_RSRC(a.pooled, (unsigned)windows * N * 4);
        const unsigned po = (unsigned)((m0 / 6 + 8 * fh) * N + ch0) * 4;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            float sx = yv[6 * w].x, sy = yv[6 * w].y;
#pragma unroll
            for (int p = 1; p < 6; ++p) {
                sx += yv[6 * w + p].x;
                sy += yv[6 * w + p].y;
            }
            const float six = 6.0f;
            sx /= six;
            sy /= six;
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, sx), __builtin_bit_cast(unsigned, sy)}, pr, po, w * N * 4, 0);
        }
    } else {
        const __amdgpu_buffer_rsrc_t orr = TAIL_RSRC(a.out + (size_t)m0 * N, (unsigned)x_cnt * N * 4);
        const unsigned oo = (unsigned)(48 * fh * N + ch0) * 4;
#pragma unroll
        for (int w = 0; w < 8; ++w)
#pragma unroll
            for (int oy = 0; oy < 3; ++oy)
#pragma unroll
                for (int ox = 0; ox < 2; ++ox) {
                    v2f s2 = shift2;
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const int iy = oy + kh - 1, ix = ox + kw - 1;
                            if (iy < 0 || iy >= 3 || ix < 0 || ix >= 2) continue;
                            s2 = __builtin_elementwise_fma(yv[6 * w + 2 * iy + ix], wt2[kh * 3 + kw], s2);
                        }
                    const float ox_ = fmaxf(s2.x, 0.0f), oy_ = fmaxf(s2.y, 0.0f);
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, ox_), __builtin_bit_cast(unsigned, oy_)}, orr, oo,
                                                          (6 * w + 2 * oy + ox) * N * 4, 0);
                }
    }
}

template <int K, int EPI>
void launch_tail_f32(const TailArgsF32& a, int windows, hipStream_t stream) {
    static std::once_flag once[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & 63], [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_gemm_f32_kernel<K, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, kTailLds);
    });
    const int M = windows * 6;
    const int nrt4 = ((M + 95) / 96 + 3) / 4 * 4;
    hipLaunchKernelGGL((tail_gemm_f32_kernel<K, EPI>), dim3(4 * nrt4), dim3(256), kTailLds, stream, a, M, windows);
}

template <int K, int EPI, bool PLAIN>
void launch_tail(const TailArgs& a, int windows, unsigned* range_flag, hipStream_t stream) {
    static std::once_flag once[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & 63], [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_gemm_kernel<K, EPI, PLAIN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  kTailLds);
    });
    const int M = windows * 6;
    const int nrt4 = ((M + 95) / 96 + 3) / 4 * 4;      // row tiles, rounded up to the four XCD pairs (workgroups past M leave at once)
#ifdef BD_KERNEL_TRACE      // developer build only: BD_WS_TRACE=9 stamps wave 0 of workgroups 0 and 101
    const char* tr = getenv("BD_WS_TRACE");
    if (tr && tr[0] == '9') {
        static unsigned long long* dbg = nullptr;
        static int shots[2] = {0, 0};
        if (!dbg) (void)hipMalloc(&dbg, 64 * 8);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_gemm_kernel<K, EPI, PLAIN, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  kTailLds);
        (void)hipMemsetAsync(dbg, 0, 64 * 8, stream);
        hipLaunchKernelGGL((tail_gemm_kernel<K, EPI, PLAIN, true>), dim3(4 * nrt4), dim3(256), kTailLds, stream, a, M, windows, range_flag, dbg);
        (void)hipStreamSynchronize(stream);
        unsigned long long h[64];
        (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        if (++shots[EPI] == 8)
            for (int w = 0; w < 2; ++w) {
                fprintf(stderr, "[trace] tail K = %d, workgroup %d: cycles between stamps (prologue | one per stage | epilogue | stores land):", K,
                        w ? 101 : 0);
                for (int i = 1; i < 32 && h[w * 32 + i]; ++i) fprintf(stderr, " %llu", h[w * 32 + i] - h[w * 32 + i - 1]);
                fprintf(stderr, "\n");
                if (!w) fprintf(stderr, "[trace] ... workgroup 0 lived %llu ticks of the 100 MHz clock\n", h[31] - h[30]);
            }
        return;
    }
#endif
    hipLaunchKernelGGL((tail_gemm_kernel<K, EPI, PLAIN>), dim3(4 * nrt4), dim3(256), kTailLds, stream, a, M, windows, range_flag);
}

}  // namespace

// Planes: a [rows][C] activation as two f16 halves, hi at p, lo at p + rows * C (halves): 4 bytes per element, the f32 buffer's size.
//
// Pointwise 13 + depthwise 14: in = depthwise-13 output as planes [windows * 6][512] (sep_chip_kernel with planes = true), out =
// depthwise-14 output as planes [windows * 6][1024].  False (nothing launched) when shapes, mode or table layout are not the
// ones the kernel is built for.
bool tail_supported(const SepLayer& L13, const SepLayer& L14) {
    if (L13.cin != 512 || L13.cout != 1024 || L13.h_out != 3 || L13.w_out != 2 || L14.cin != 1024 || L14.cout != 1024 || L14.stride != 1 ||
        L14.h_out != 3 || L14.w_out != 2)
        return false;
    return L13.pw_mode != 0 && L14.pw_mode == L13.pw_mode && dw_b_of(L14) == dw_w_of(L14) + 9 * 1024;
}

bool launch_tail_pw13_dw14(const void* in, void* out, int windows, const SepLayer& L13, const SepLayer& L14, hipStream_t stream) {
    if (windows <= 0 || windows > (1 << 18) || in == out || !tail_supported(L13, L14)) return false;
    const size_t rows = (size_t)windows * 6;
    TailArgs a{};
    a.ahi = static_cast<const _Float16*>(in);
    a.alo = a.ahi + rows * 512;
    a.bhi = static_cast<const _Float16*>(L13.pw_fhi);
    a.blo = static_cast<const _Float16*>(L13.pw_flo);
    a.pu = L13.pw_u;
    a.pb = L13.pw_b;
    a.taps = dw_w_of(L14);
    a.ohi = static_cast<_Float16*>(out);
    a.olo = a.ohi + rows * 1024;
    if (L13.pw_mode == 2) launch_tail<512, 0, true>(a, windows, L14.range_flag, stream);
    else launch_tail<512, 0, false>(a, windows, L14.range_flag, stream);
    return true;
}

// Pointwise 14 + average pool: in = depthwise-14 output as planes [windows * 6][1024], pooled = [windows][1024] f32.
bool launch_tail_pw14_pool(const void* in, float* pooled, int windows, const SepLayer& L14, hipStream_t stream) {
    if (windows <= 0 || windows > (1 << 18)) return false;
    if (L14.cin != 1024 || L14.cout != 1024 || L14.h_out != 3 || L14.w_out != 2 || L14.pw_mode == 0) return false;
    const size_t rows = (size_t)windows * 6;
    TailArgs a{};
    a.ahi = static_cast<const _Float16*>(in);
    a.alo = a.ahi + rows * 1024;
    a.bhi = static_cast<const _Float16*>(L14.pw_fhi);
    a.blo = static_cast<const _Float16*>(L14.pw_flo);
    a.pu = L14.pw_u;
    a.pb = L14.pw_b;
    a.pooled = pooled;
    if (L14.pw_mode == 2) launch_tail<1024, 1, true>(a, windows, nullptr, stream);
    else launch_tail<1024, 1, false>(a, windows, nullptr, stream);
    return true;
}

// The exact-f32 mode's two launches: in = depthwise-13 output [windows * 6][512] f32 (sep_chip_f32_kernel), mid = depthwise-14 output
// [windows * 6][1024] f32, pooled = [windows][1024].  False (nothing launched) when a shape or the fragment-ordered weights are missing.
bool tail_f32_supported(const SepLayer& L13, const SepLayer& L14) {
    return L13.cin == 512 && L13.cout == 1024 && L13.h_out == 3 && L13.w_out == 2 && L14.cin == 1024 && L14.cout == 1024 && L14.stride == 1 &&
           L14.h_out == 3 && L14.w_out == 2 && L13.pw_ffrag && L14.pw_ffrag;
}
bool launch_tail_f32(const float* in, float* mid, float* pooled, int windows, const SepLayer& L13, const SepLayer& L14, hipStream_t stream,
                     int which) {
    if (windows <= 0 || windows > (1 << 17) || !tail_f32_supported(L13, L14)) return false;      // (32-bit byte offsets into [6 windows][1024] f32)
    TailArgsF32 a{};
    if (which == 0) {
        a.a = in;
        a.bfrag = L13.pw_ffrag;
        a.pb = L13.pw_b;
        a.taps = L14.dw_w;
        a.tshift = L14.dw_b;
        a.out = mid;
        launch_tail_f32<512, 0>(a, windows, stream);
    } else {
        a.a = mid;
        a.bfrag = L14.pw_ffrag;
        a.pb = L14.pw_b;
        a.pooled = pooled;
        launch_tail_f32<1024, 1>(a, windows, stream);
    }
    return true;
}

}  // namespace bd
