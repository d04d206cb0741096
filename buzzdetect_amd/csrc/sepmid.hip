// The middle of YAMNet as one launch whose tiles stay on the CU (round 5): pointwise 5 (128 -> 256 on the 12 x 8 map) ->
// layer 6 (depthwise 3 x 3 + pointwise 256 -> 256) -> depthwise 7 (3 x 3, stride 2 -> 6 x 4) -> pointwise 7 (256 -> 512)
// (yamnet.py:83-85).  Until now four launches (pw_res_kernel, sep_ws_kernel with the next depthwise in its epilogue,
// pw_res_kernel) that hand [96][256] + [24][256] f32 per window through global memory: 250 MB per 1024 windows.
//
// A tile is ONE window (96 positions of the 12 x 8 map).  The scheme is sepchip.hip's:
//   * 8 MFMA waves (two per SIMD, 256 VGPRs), wave w owns output column tile w (layers 5, 6: 256 columns = 8 tiles; layer 7: 512
//     columns, tiles w and w + 8);
//   * an A operand is published into LDS one 32-channel STAGE at a time (split-f16 hi | lo, sepchip.hip's image: 64-byte rows,
//     16-byte slots XORed with (row >> 2) & 3, rows 48.. skewed by 64 bytes) - stage s of a layer is column tile s of the layer
//     before, i.e. wave s's own output;
//   * the tile's row order in LDS is NOT the map's: position (y, x) sits at row 48 (x >> 2) + 4 y + (x & 3).  With that order a
//     lane (channel c, half h) of the 32 x 32 accumulator layout holds, after one v_permlane32_swap per register pair, the
//     columns 4 h .. 4 h + 3 of ALL twelve map rows of its channel; the one column it lacks for a 3 x 3 window (x = 4 h - 1 or
//     4 h + 4: the other half-wave's edge) comes with one more swap per map row, columns outside the map are a select to zero.
//     The depthwise of layer 6 and the stride-2 depthwise of layer 7 then run in registers with compile-time neighbours;
//   * K <= 256, so a whole A operand fits the ring (8 stages = 97 KB) - no pending stages.  A5 (4 stages) and A6 (8) share the
//     ring; A7 has its own 48 KB and collects TWO windows (24 + 24 rows) before layer 7 runs: with one window's 24 rows per
//     pass layer 7 streamed its 512 KB of weights per window from L2 - 85 B/clk/CU, more than the path delivers - and took 13.5 k
//     cycles for 6 k of matrix work.  LDS = 97 + 48 + 1 KB, one persistent workgroup per CU walking a contiguous run of windows.
// Same-box A/B at 938 windows per launch (bench.py --per-slot, three alternations): 99.3 / 98.9 / 99.7 us against 31.3 + 64.2 +
// 28.8 us for the launches it replaces, 1.844 against 1.779 M windows/s for the job (one window per layer-7 pass: 106.4 us).
// Arithmetic per element is that of the kernels it replaces (pw_res_kernel, sep_ws_kernel<NDW = 1>): products lo*hi, hi*lo,
// hi*hi per k16 step in ascending order, relu(fma(acc, u, b)), depthwise = shift then taps in row-major order with fmaf, the
// range guard's maximum over everything that is split.  Taps outside the map are skipped or multiply a zero (sepchip.hip).
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

constexpr float kF16MaxMid = 65504.0f;
constexpr int kHalfA = 97 * 64;                 // one f16 half of a 96-row stage (rows 48.. one row further)
constexpr int kSlotA = 2 * kHalfA;              // 12416
constexpr int kHalf7 = 48 * 64;                 // layer 7's A operand: the 24 rows of TWO windows per stage
constexpr int kSlot7 = 2 * kHalf7;              // 6144
constexpr int kOffA7 = 8 * kSlotA;              // A7 behind the ring A5 and A6 share
constexpr int kMidLds = kOffA7 + 8 * kSlot7 + 1024;   // 149504 (+ 1 KB: the second row tile of layer 7 reads 16 rows past a stage)
static_assert(kMidLds <= 160 * 1024, "one workgroup per CU");

struct MidArgs {
    const _Float16 *w5h, *w5l, *w6h, *w6l, *w7h, *w7l;     // MFMA B-fragment order [cout / 32][cin / 16][64][8]
    const float *u5, *b5, *u6, *b6, *u7, *b7;              // epilogue factor / shift per output channel
    const float *dw6, *dw7;                                // [9][256] taps * 2^act_exp followed by [256] shift
};

#define MID_RSRC(P, BYTES) __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(P)), 0, (BYTES), 0x00020000)
#define MID_LD32(R, VOFF, SOFF) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(R, VOFF, SOFF, 0))
#define MID_LD128(R, VOFF, SOFF) __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(R, VOFF, SOFF, 0))

template <bool PLAIN, bool TRACE>
__global__ __launch_bounds__(512, 2) void sep_mid_kernel(const MidArgs a, const float* __restrict__ X, float* __restrict__ Y, int windows,
                                                          unsigned* __restrict__ range_flag, unsigned long long* __restrict__ dbg) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wc = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fh = lane >> 5;
    float rmax = 0.0f;
    int tsn = 0;
#define MID_TS()                                                                                          \
    if constexpr (TRACE) {                                                                                \
        if (blockIdx.x == 0 && lane == 0 && (wc == 0 || wc == 5) && tsn < 64)                             \
            dbg[(wc == 5 ? 64 : 0) + tsn] = __builtin_amdgcn_s_memtime();                                 \
        ++tsn;                                                                                            \
    }
    // publisher (96-row stages): lane (k = frow, fh) writes rows 48 fh + rl; byte offset of k in a row whose key is m = wb0 ^ (m << 4)
    const int wb0 = fh * (48 * 64 + 64) + ((frow >> 3) << 4) + 2 * (frow & 7);
    // publisher (layer 7's 48-row stages): rows 24 half + 4 oy + 2 fh + j
    const int wb7 = fh * 128 + ((frow >> 3) << 4) + 2 * (frow & 7);
    // reader: lane (frow, fh) supplies A[row 32 i + frow][k = 16 s + 8 fh ..]: slot (2 s + fh) ^ key, key = (frow >> 2) & 3
    const int ra0 = frow * 64 + ((fh ^ ((frow >> 2) & 3)) << 4);
    const int ra1 = ra0 + 2048 + (frow >= 16 ? 64 : 0);
    const unsigned lane16 = lane * 16, c4 = frow * 4;


#define MID_PUT(BASE, RL, PK)                                                                             \
    {                                                                                                     \
        char* const p_ = (BASE) + (wbl ^ ((((RL) >> 2) & 3) << 4)) + (RL) * 64;                           \
        *reinterpret_cast<unsigned short*>(p_) = (unsigned short)(PK);                                    \
        *reinterpret_cast<unsigned short*>(p_ + kHalfA) = (unsigned short)((PK) >> 16);                   \
    }
    // v -> (hi | lo << 16), the range guard's running maximum in the same ordered statement (sepchip.hip)
#define MID_SPLIT(V, PK)                                                                                  \
    unsigned PK = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)(V));                            \
    asm volatile("v_fma_mixhi_f16 %0, %0, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_max_f32 %1, %1, |%2|"   \
                 : "+v"(PK), "+v"(rmax) : "v"(V));
    // one k16 step of a K loop: NI row tiles x NJ column tiles, A from LDS at ABASE (row tile i: ra0 / ra1 / ra0 + 4160), B in BH / BL
#define MID_MMA(ACC, AH, AL, BHV, BLV)                                                                    \
    if constexpr (!PLAIN) {                                                                               \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL, BHV, ACC, 0, 0, 0);                              \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BLV, ACC, 0, 0, 0);                              \
    }                                                                                                     \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BHV, ACC, 0, 0, 0);

    // accumulators of a layer -> this lane's half of the map: relu(fma(acc, u, b)), then the half-wave swap; ev[y][e] = position
    // (y, 4 fh + e).  LDS row of accumulator (i, r, half H) is 32 i + 8 (r >> 2) + 4 H + (r & 3) = 48 h + 4 y + e: pair q of
    // x < 4 rows (2 q, 2 q + 1) is quad (q >> 2, q & 3), the same pair of x >= 4 is quad ((q + 6) >> 2, (q + 6) & 3).
#define MID_TILE_TO_MAP(ACC, U, B, EV)                                                                    \
    _Pragma("unroll") for (int q = 0; q < 6; ++q)                                                         \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                   \
            const float lo_ = fmaxf(fmaf(ACC[q >> 2][4 * (q & 3) + e], U, B), 0.0f);                      \
            const float hi_ = fmaxf(fmaf(ACC[(q + 6) >> 2][4 * ((q + 6) & 3) + e], U, B), 0.0f);          \
            const auto r_ = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, lo_), __builtin_bit_cast(unsigned, hi_), false, false); \
            EV[2 * q][e] = __builtin_bit_cast(float, (unsigned)r_[0]);                                    \
            EV[2 * q + 1][e] = __builtin_bit_cast(float, (unsigned)r_[1]);                                \
        }
    // the column this half lacks: x = 4 fh - 1 (LH, zero for fh = 0) and x = 4 fh + 4 (RH, zero for fh = 1), per map row
#define MID_HALO(EV, LH, RH)                                                                              \
    _Pragma("unroll") for (int y = 0; y < 12; ++y) {                                                      \
        const auto r_ = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, EV[y][0]), __builtin_bit_cast(unsigned, EV[y][3]), false, false); \
        LH[y] = fh ? __builtin_bit_cast(float, (unsigned)r_[0]) : 0.0f;       /* upper half: the lower half's column 3 */ \
        RH[y] = fh ? 0.0f : __builtin_bit_cast(float, (unsigned)r_[1]);       /* lower half: the upper half's column 4 */ \
    }

    // ... as row pairs for the packed depthwise
#define MID_ROW_PAIRS(EV, LH, RH, IN2)                                                                    \
    _Pragma("unroll") for (int c = 0; c < 6; ++c) {                                                       \
        _Pragma("unroll") for (int r = 0; r < 6; ++r)                                                     \
            IN2[1 + r][c] = c == 0 ? v2f{LH[r], LH[r + 6]} : c == 5 ? v2f{RH[r], RH[r + 6]} : v2f{EV[r][c - 1], EV[r + 6][c - 1]}; \
        IN2[0][c] = v2f{0.0f, c == 0 ? LH[5] : c == 5 ? RH[5] : EV[5][c - 1]};                            \
        IN2[7][c] = v2f{c == 0 ? LH[6] : c == 5 ? RH[6] : EV[6][c - 1], 0.0f};                            \
    }

    // this wave's share of a window's input (the depthwise-5 output): stage wc & 3, rows 24 (wc >> 2) .. + 23 of the lane's half.
    // Requested a layer ahead of its use: the loads of window n + 1 fly behind the depthwise 7 (and pointwise 7) of window n.
    float vin[24];
    auto fetch_window = [&](int win) {
        const __amdgpu_buffer_rsrc_t xr = MID_RSRC(X + (size_t)win * 96 * 128, 96 * 128 * 4);
        const int st = wc & 3, rl0 = 24 * (wc >> 2);
        const unsigned vo = (4u * fh * 128) * 4 + c4;
#pragma unroll
        for (int t = 0; t < 24; ++t) {
            const int rl = rl0 + t;                                        // (wave-uniform; map row rl >> 2, column 4 fh + (rl & 3))
            vin[t] = MID_LD32(xr, vo, ((8 * (rl >> 2) + (rl & 3)) * 128 + 32 * st) * 4);
        }
    };
    // layer 7's B fragments (column tiles wc, wc + 8), a ring of three k16 steps; the first two are requested by the window that
    // completes A7, in front of its last barrier
    f16x8 b7h[3][2], b7l[3][2];
    const __amdgpu_buffer_rsrc_t r7h = MID_RSRC(a.w7h, 256 * 512 * 2), r7l = MID_RSRC(a.w7l, 256 * 512 * 2);
    // one window through pointwise 5, layer 6 and depthwise 7; its 24 depthwise-7 rows land at rows 24 HALF .. of A7
    auto window_to_a7 = [&](auto half_c, auto last_c, int win, int win_next) {
        constexpr int HALF = decltype(half_c)::value;
        constexpr bool LAST = decltype(last_c)::value;     // layer 7 runs behind this window
        // ---- A5: split, stage wc & 3 of the ring
        {
            const int st = wc & 3;
            int wbl = wb0;
            asm volatile("" : "+v"(wbl));
            char* const slot = sm + st * kSlotA;
            if (wc < 4) {
#pragma unroll
                for (int t = 0; t < 24; ++t) {
                    MID_SPLIT(vin[t], pk)
                    MID_PUT(slot, t, pk)
                }
            } else {
#pragma unroll
                for (int t = 0; t < 24; ++t) {
                    MID_SPLIT(vin[t], pk)
                    MID_PUT(slot, 24 + t, pk)
                }
            }
        }
        // (the B fragments of a K loop's first two k16 steps are requested in FRONT of the barrier that publishes its A operand:
        //  behind it every wave of the workgroup would wait out the same L2 round trip at once - round 6)
        f16x8 bh[3], bl[3];                       // B fragments two k16 steps (18 MFMAs) ahead of their use
        const __amdgpu_buffer_rsrc_t r5h = MID_RSRC(a.w5h, 128 * 256 * 2), r5l = MID_RSRC(a.w5l, 128 * 256 * 2);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bh[q] = MID_LD128(r5h, lane16, (wc * 8 + q) * 1024);
            bl[q] = MID_LD128(r5l, lane16, (wc * 8 + q) * 1024);
        }
        __syncthreads();                          // A5 published
        MID_TS()
        // ---- pointwise 5: [96][128] x [128][256], column tile wc
        f32x16 acc[3];
        float zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = zero;
        {
            constexpr int KQ = 8;
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                if (q + 2 < KQ) {
                    bh[(q + 2) % 3] = MID_LD128(r5h, lane16, (wc * KQ + q + 2) * 1024);
                    bl[(q + 2) % 3] = MID_LD128(r5l, lane16, (wc * KQ + q + 2) * 1024);
                }
                const char* const ab = sm + (q >> 1) * kSlotA;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const char* const ap = ab + ((i == 1 ? ra1 : ra0) ^ ((q & 1) << 5)) + (i == 2 ? 4096 + 64 : 0);
                    const f16x8 ah = *reinterpret_cast<const f16x8*>(ap), al = *reinterpret_cast<const f16x8*>(ap + kHalfA);
                    MID_MMA(acc[i], ah, al, bh[q % 3], bl[q % 3])
                }
            }
        }
        MID_TS()
        // ---- depthwise 6 in registers; its outputs wait (packed) for the ring: A6 takes the slots A5 is still read from
        const __amdgpu_buffer_rsrc_t r6h = MID_RSRC(a.w6h, 256 * 256 * 2), r6l = MID_RSRC(a.w6l, 256 * 256 * 2);
        {
            unsigned out6[48];
            {
                const __amdgpu_buffer_rsrc_t ur = MID_RSRC(a.u5, 1024), br = MID_RSRC(a.b5, 1024), tr = MID_RSRC(a.dw6, 10 * 256 * 4);
                const float u = MID_LD32(ur, c4, 128 * wc), b = MID_LD32(br, c4, 128 * wc);
                float wt[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) wt[t] = MID_LD32(tr, c4, (t * 256 + 32 * wc) * 4);
                const float shift = MID_LD32(tr, c4, (9 * 256 + 32 * wc) * 4);
                float ev[12][4], lh[12], rh[12];
                MID_TILE_TO_MAP(acc, u, b, ev)
                MID_HALO(ev, lh, rh)
                // two map rows per v_pk_fma_f32: in2[1 + r][1 + c] = (row r, row r + 6) at column 4 fh + c, r = -1 .. 6, c = -1 .. 4;
                // the rows outside the map are zeros (a tap there adds 0 * w: sepchip.hip on why that is the skipped tap's result)
                v2f in2[8][6];
                MID_ROW_PAIRS(ev, lh, rh, in2)
#pragma unroll
                for (int y = 0; y < 6; ++y)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v2f sacc = {shift, shift};
#pragma unroll
                        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                            for (int kw = 0; kw < 3; ++kw)
                                sacc = __builtin_elementwise_fma(in2[y + kh][e + kw], v2f{wt[kh * 3 + kw], wt[kh * 3 + kw]}, sacc);
#pragma unroll
                        for (int w = 0; w < 2; ++w) {
                            const float o = fmaxf(w ? sacc.y : sacc.x, 0.0f);
                            MID_SPLIT(o, pk)
                            out6[4 * (y + 6 * w) + e] = pk;
                        }
                    }
            }
            __syncthreads();                      // every wave has read A5: the ring is free for A6
            MID_TS()
#pragma unroll
            for (int q = 0; q < 2; ++q) {         // layer 6's first B fragments, in flight behind the publication and its barrier
                bh[q] = MID_LD128(r6h, lane16, (wc * 16 + q) * 1024);
                bl[q] = MID_LD128(r6l, lane16, (wc * 16 + q) * 1024);
            }
            int wbl = wb0;
            asm volatile("" : "+v"(wbl));
            char* const slot = sm + wc * kSlotA;
#pragma unroll
            for (int rl = 0; rl < 48; ++rl) MID_PUT(slot, rl, out6[rl])
        }
        __syncthreads();                          // A6 published
        MID_TS()
        // ---- pointwise 6: [96][256] x [256][256], column tile wc
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = zero;
        {
            constexpr int KQ = 16;
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                if (q + 2 < KQ) {
                    bh[(q + 2) % 3] = MID_LD128(r6h, lane16, (wc * KQ + q + 2) * 1024);
                    bl[(q + 2) % 3] = MID_LD128(r6l, lane16, (wc * KQ + q + 2) * 1024);
                }
                const char* const ab = sm + (q >> 1) * kSlotA;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const char* const ap = ab + ((i == 1 ? ra1 : ra0) ^ ((q & 1) << 5)) + (i == 2 ? 4096 + 64 : 0);
                    const f16x8 ah = *reinterpret_cast<const f16x8*>(ap), al = *reinterpret_cast<const f16x8*>(ap + kHalfA);
                    MID_MMA(acc[i], ah, al, bh[q % 3], bl[q % 3])
                }
            }
        }
        MID_TS()
        // ---- depthwise 7 (stride 2) in registers -> rows 24 HALF .. of A7, stage wc; the next window's input is requested here
        if (win_next >= 0) fetch_window(win_next);
        {
            const __amdgpu_buffer_rsrc_t ur = MID_RSRC(a.u6, 1024), br = MID_RSRC(a.b6, 1024), tr = MID_RSRC(a.dw7, 10 * 256 * 4);
            const float u = MID_LD32(ur, c4, 128 * wc), b = MID_LD32(br, c4, 128 * wc);
            float wt[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) wt[t] = MID_LD32(tr, c4, (t * 256 + 32 * wc) * 4);
            const float shift = MID_LD32(tr, c4, (9 * 256 + 32 * wc) * 4);
            float ev[12][4], lh[12], rh[12];
            MID_TILE_TO_MAP(acc, u, b, ev)
            MID_HALO(ev, lh, rh)
            int wbl = wb7;
            asm volatile("" : "+v"(wbl));
            char* const slot = sm + kOffA7 + wc * kSlot7;
            // output (oy, ox = 2 fh + j) reads map rows 2 oy + kh, columns 2 ox + kw = 4 fh + 2 j + kw; SAME padding of a
            // stride-2 layer: one row / column BEHIND the map (row 12; column 8 = the upper half's right halo = 0)
            // two output rows (oy, oy + 3) per v_pk_fma_f32: their input rows 2 oy + kh and 2 oy + kh + 6 are one row pair
            v2f in2[8][6];
            MID_ROW_PAIRS(ev, lh, rh, in2)
#pragma unroll
            for (int oy = 0; oy < 3; ++oy)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    v2f sacc = {shift, shift};
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw)
                            sacc = __builtin_elementwise_fma(in2[1 + 2 * oy + kh][1 + 2 * j + kw], v2f{wt[kh * 3 + kw], wt[kh * 3 + kw]}, sacc);
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        const int oyw = oy + 3 * w;
                        const float o = fmaxf(w ? sacc.y : sacc.x, 0.0f);
                        MID_SPLIT(o, pk)
                        // row 24 HALF + 4 oy + 2 fh + j of the 48-row stage: key (row >> 2) & 3 = (6 HALF + oy) & 3
                        char* const p_ = slot + (wbl ^ (((6 * HALF + oyw) & 3) << 4)) + (24 * HALF + 4 * oyw + j) * 64;
                        *reinterpret_cast<unsigned short*>(p_) = (unsigned short)pk;
                        *reinterpret_cast<unsigned short*>(p_ + kHalf7) = (unsigned short)(pk >> 16);
                    }
                }
        }
        if constexpr (LAST) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    b7h[q][j] = MID_LD128(r7h, lane16, ((wc + 8 * j) * 16 + q) * 1024);
                    b7l[q][j] = MID_LD128(r7l, lane16, ((wc + 8 * j) * 16 + q) * 1024);
                }
        }
        __syncthreads();                          // A7's rows of this window published; every wave has read A6
        MID_TS()
    };
    // pointwise 7 over the NW windows collected in A7: [24 NW (32 NW)][256] x [256][512], column tiles wc, wc + 8
    auto layer7 = [&](auto nw_c, int win) {
        constexpr int NW = decltype(nw_c)::value;
        f32x16 c7[NW][2];
        float zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
#pragma unroll
        for (int i = 0; i < NW; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) c7[i][j][r] = zero;
        constexpr int KQ = 16;
        const int ra7 = frow * 64 + ((fh ^ ((frow >> 2) & 3)) << 4);          // row frow (+ 32): key (row >> 2) & 3
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            if (q + 2 < KQ) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    b7h[(q + 2) % 3][j] = MID_LD128(r7h, lane16, ((wc + 8 * j) * KQ + q + 2) * 1024);
                    b7l[(q + 2) % 3][j] = MID_LD128(r7l, lane16, ((wc + 8 * j) * KQ + q + 2) * 1024);
                }
            }
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const char* const ap = sm + kOffA7 + (q >> 1) * kSlot7 + (ra7 ^ ((q & 1) << 5)) + i * 2048;
                const f16x8 ah = *reinterpret_cast<const f16x8*>(ap), al = *reinterpret_cast<const f16x8*>(ap + kHalf7);
#pragma unroll
                for (int j = 0; j < 2; ++j) { MID_MMA(c7[i][j], ah, al, b7h[q % 3][j], b7l[q % 3][j]) }
            }
        }
        MID_TS()
        // bias + ReLU: accumulator (i, r, half fh) is row 32 i + 8 (r >> 2) + 4 fh + (r & 3) of the NW x 24 output rows of the
        // windows win, win + 1 (consecutive in Y); rows past them are dropped by the resource's range
        const __amdgpu_buffer_rsrc_t ur = MID_RSRC(a.u7, 2048), br = MID_RSRC(a.b7, 2048);
        const __amdgpu_buffer_rsrc_t yr = MID_RSRC(Y + (size_t)win * 24 * 512, NW * 24 * 512 * 4);
        const unsigned yo = (4u * fh * 512) * 4 + c4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float u = MID_LD32(ur, c4, 128 * (wc + 8 * j)), b = MID_LD32(br, c4, 128 * (wc + 8 * j));
#pragma unroll
            for (int i = 0; i < NW; ++i)
#pragma unroll
                for (int r = 0; r < (NW == 1 ? 12 : 16); ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(fmaf(c7[i][j][r], u, b), 0.0f)), yr, yo,
                                                          ((32 * i + 8 * (r >> 2) + (r & 3)) * 512 + 32 * (wc + 8 * j)) * 4, 0);
        }
        MID_TS()
    };

    MID_TS()
    // a contiguous run of windows per workgroup, taken two at a time (the last one alone when the run is odd)
    const int w_begin = (int)(((long long)blockIdx.x * windows) / gridDim.x);
    const int w_end = (int)(((long long)(blockIdx.x + 1) * windows) / gridDim.x);
    if (w_begin < w_end) fetch_window(w_begin);
    for (int win = w_begin; win < w_end; win += 2) {
        const bool two = win + 1 < w_end;
        if (two) {
            window_to_a7(std::integral_constant<int, 0>{}, std::false_type{}, win, win + 1);
            window_to_a7(std::integral_constant<int, 1>{}, std::true_type{}, win + 1, win + 2 < w_end ? win + 2 : -1);
            layer7(std::integral_constant<int, 2>{}, win);
        } else {
            window_to_a7(std::integral_constant<int, 0>{}, std::true_type{}, win, -1);
            layer7(std::integral_constant<int, 1>{}, win);
        }
    }
    if (range_flag && !(rmax <= kF16MaxMid)) *range_flag = 1u;
#undef MID_TS
#undef MID_PUT
#undef MID_SPLIT
#undef MID_MMA
#undef MID_TILE_TO_MAP
#undef MID_HALO
}

template <bool PLAIN>
void launch_mid(const float* in, float* out, int windows, const SepLayer& L5, const SepLayer& L6, const SepLayer& L7, hipStream_t stream) {
    MidArgs a{};
    a.w5h = static_cast<const _Float16*>(L5.pw_fhi);
    a.w5l = static_cast<const _Float16*>(L5.pw_flo);
    a.w6h = static_cast<const _Float16*>(L6.pw_fhi);
    a.w6l = static_cast<const _Float16*>(L6.pw_flo);
    a.w7h = static_cast<const _Float16*>(L7.pw_fhi);
    a.w7l = static_cast<const _Float16*>(L7.pw_flo);
    a.u5 = L5.pw_u; a.b5 = L5.pw_b;
    a.u6 = L6.pw_u; a.b6 = L6.pw_b;
    a.u7 = L7.pw_u; a.b7 = L7.pw_b;
    a.dw6 = dw_w_of(L6);
    a.dw7 = dw_w_of(L7);
    static std::once_flag once[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & 63], [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sep_mid_kernel<PLAIN, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kMidLds);
    });
    int grid = cu_count();
    if (grid > windows) grid = windows;
#ifdef BD_KERNEL_TRACE      // developer build only: BD_WS_TRACE=8 stamps the phases of workgroup 0 (waves 0 and 5)
    const char* tr = getenv("BD_WS_TRACE");
    if (tr && tr[0] == '8') {
        static unsigned long long* dbg = nullptr;
        static int shots = 0;
        if (!dbg) (void)hipMalloc(&dbg, 128 * 8);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sep_mid_kernel<PLAIN, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kMidLds);
        (void)hipMemsetAsync(dbg, 0, 128 * 8, stream);
        hipLaunchKernelGGL((sep_mid_kernel<PLAIN, true>), dim3(grid), dim3(512), kMidLds, stream, a, in, out, windows, L5.range_flag, dbg);
        (void)hipStreamSynchronize(stream);
        unsigned long long h[128];
        (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        if (++shots == 8)
            for (int w = 0; w < 2; ++w) {
                fprintf(stderr, "[trace] mid run, wave %d: cycles between stamps (per window: A5+B1, K5, dw6+Bx, publish+B2, K6, dw7+B3; per pair: K7, stores):", w ? 5 : 0);
                for (int i = 1; i < 64 && h[w * 64 + i]; ++i) fprintf(stderr, " %llu", h[w * 64 + i] - h[w * 64 + i - 1]);
                fprintf(stderr, "\n");
            }
        return;
    }
#endif
    hipLaunchKernelGGL((sep_mid_kernel<PLAIN, false>), dim3(grid), dim3(512), kMidLds, stream, a, in, out, windows, L5.range_flag,
                       (unsigned long long*)nullptr);
}

}  // namespace

// Pointwise 5 -> layer 6 -> depthwise 7 -> pointwise 7 as one launch: in = depthwise-5 output [windows][12][8][128] (what
// l4_window_kernel writes), out = layer-7 output [windows][6][4][512].  L5, L6, L7 = layers 5, 6, 7.  False (nothing launched)
// when the shapes or the table layouts are not the ones the kernel is built for.
bool launch_separable_mid(const float* in, float* out, int windows, const SepLayer& L5, const SepLayer& L6, const SepLayer& L7,
                          hipStream_t stream) {
    if (windows <= 0 || in == out) return false;
    if (L5.cin != 128 || L5.cout != 256 || L5.h_out != 12 || L5.w_out != 8 || L6.cin != 256 || L6.cout != 256 || L6.stride != 1 ||
        L6.h_out != 12 || L6.w_out != 8 || L7.cin != 256 || L7.cout != 512 || L7.stride != 2 || L7.h_out != 6 || L7.w_out != 4)
        return false;
    if (L5.pw_mode == 0 || L6.pw_mode != L5.pw_mode || L7.pw_mode != L5.pw_mode) return false;
    if (dw_b_of(L6) != dw_w_of(L6) + 9 * 256 || dw_b_of(L7) != dw_w_of(L7) + 9 * 256) return false;
    if (L5.pw_mode == 2) launch_mid<true>(in, out, windows, L5, L6, L7, stream);
    else launch_mid<false>(in, out, windows, L5, L6, L7, stream);
    return true;
}

}  // namespace bd
