// Pointwise 5 -> layer 6 -> layer 7 of YAMNet as ONE launch in the exact-f32 mode (bd_set_pointwise_mode 0): sepmid.hip's scheme -
// one window per tile, accumulators -> depthwise 6 / stride-2 depthwise 7 in registers -> LDS stage ring, layer 7 over window
// pairs - with every product on v_mfma_f32_32x32x2_f32 (round 5).  Until now three launches of pointwise_kernel<96, 128, ..> with the
// next depthwise in their epilogues (71 + 111 + 64 us per 938 windows, 0.55 - 0.68 of the f32-MFMA peak: four to eight K stages
// per tile behind a full pipeline fill, the tile through LDS and HBM between them).
//
// What differs from sepmid.hip is what differs between sepchipf32.hip and sepchip.hip: a stage tile is [rows][32 k] f32 in
// 128-byte rows (the same bytes as the (hi, lo) f16 pair), 16-byte chunk XORed with (row >> 1) & 7, rows 48.. of a 96-row stage
// one row further; a published value is ONE ds_write_b32; per super-step of 8 k one ds_read_b128 per row tile and one 16-byte
// global load per column tile from the fragment-ordered f32 weights (SepLayer::pw_ffrag), then four matrix instructions per
// (row tile, column tile).  Arithmetic is depthwise_kernel + pointwise_kernel's bit for bit (sepchipf32.hip);
// tests/test_gpu_parity.py::test_fused_f32_mode_equals_one_kernel_per_op covers it.
#include "bd_internal.h"

#include <mutex>
#include <type_traits>

namespace bd {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kRowB = 128;                      // a stage row: 32 k as f32
constexpr int kSlotA = 97 * kRowB;              // 12416: a 96-row stage (rows 48.. one row further)
constexpr int kSlot7 = 48 * kRowB;              // 6144: layer 7's A operand, the 24 rows of TWO windows per stage
constexpr int kOffA7 = 8 * kSlotA;              // A7 behind the ring A5 and A6 share
constexpr int kMidF32Lds = kOffA7 + 8 * kSlot7 + 2048;   // 150528 (+ 2 KB: the second row tile of layer 7 reads 16 rows past a stage)
static_assert(kMidF32Lds <= 160 * 1024, "one workgroup per CU");

struct MidF32Args {
    const float *w5, *w6, *w7;                  // fragment order [cout / 32][cin / 8][64][4]
    const float *b5, *b6, *b7;                  // shift per output channel
    const float *dw6, *dw7;                     // [9][256] taps followed by [256] shift
};

#define MF_RSRC(P, BYTES) __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(P)), 0, (BYTES), 0x00020000)
#define MF_LD32(R, VOFF, SOFF) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(R, VOFF, SOFF, 0))
#define MF_LD128(R, VOFF, SOFF) __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(R, VOFF, SOFF, 0))

__global__ __launch_bounds__(512, 2) void sep_mid_f32_kernel(const MidF32Args a, const float* __restrict__ X, float* __restrict__ Y,
                                                              int windows) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wc = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fh = lane >> 5;
    // publisher (96-row stages): lane (k = frow, fh) writes rows 48 fh + rl, key (rl >> 1) & 7 for both halves
    const int wb0 = fh * (48 * kRowB + kRowB) + ((frow >> 2) << 4) + 4 * (frow & 3);
    // publisher (layer 7's 48-row stages): rows 24 half + 4 oy + 2 fh + j: key ((12 half + 2 oy) & 7) | fh - the lane's part here
    const int wb7 = (fh * 2 * kRowB + ((frow >> 2) << 4) + 4 * (frow & 3)) ^ (fh << 4);
    // reader: lane (frow, fh), super-step s of a stage: chunk (2 s + fh) ^ key, key = (frow >> 1) & 7 for every row tile
    const int ra0 = frow * kRowB + ((fh ^ ((frow >> 1) & 7)) << 4);
    const int ra1 = ra0 + 32 * kRowB + (frow >= 16 ? kRowB : 0);
    const unsigned lane16 = lane * 16, c4 = frow * 4;

#define MF_PUT(BASE, RL, V) *reinterpret_cast<float*>((BASE) + (wbl ^ ((((RL) >> 1) & 7) << 4)) + (RL) * kRowB) = (V);
    // a super-step of a K loop on one accumulator: the k pairs {8 s + j, 8 s + 4 + j}, j = 0..3 (pointwise_kernel's operand map)
#define MF_MMA(ACC, AV, BV)                                                                               \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.x, BV.x, ACC, 0, 0, 0);                                 \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.y, BV.y, ACC, 0, 0, 0);                                 \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.z, BV.z, ACC, 0, 0, 0);                                 \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.w, BV.w, ACC, 0, 0, 0);
    // accumulators of a layer -> this lane's half of the map: relu(acc + b), then the half-wave swap (sepmid.hip)
#define MF_TILE_TO_MAP(ACC, B, EV)                                                                        \
    _Pragma("unroll") for (int q = 0; q < 6; ++q)                                                         \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                   \
            const float lo_ = fmaxf(ACC[q >> 2][4 * (q & 3) + e] + B, 0.0f);                              \
            const float hi_ = fmaxf(ACC[(q + 6) >> 2][4 * ((q + 6) & 3) + e] + B, 0.0f);                  \
            const auto r_ = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, lo_), __builtin_bit_cast(unsigned, hi_), false, false); \
            EV[2 * q][e] = __builtin_bit_cast(float, (unsigned)r_[0]);                                    \
            EV[2 * q + 1][e] = __builtin_bit_cast(float, (unsigned)r_[1]);                                \
        }
#define MF_HALO(EV, LH, RH)                                                                               \
    _Pragma("unroll") for (int y = 0; y < 12; ++y) {                                                      \
        const auto r_ = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, EV[y][0]), __builtin_bit_cast(unsigned, EV[y][3]), false, false); \
        LH[y] = fh ? __builtin_bit_cast(float, (unsigned)r_[0]) : 0.0f;                                   \
        RH[y] = fh ? 0.0f : __builtin_bit_cast(float, (unsigned)r_[1]);                                   \
    }
#define MF_ROW_PAIRS(EV, LH, RH, IN2)                                                                     \
    _Pragma("unroll") for (int c = 0; c < 6; ++c) {                                                       \
        _Pragma("unroll") for (int r = 0; r < 6; ++r)                                                     \
            IN2[1 + r][c] = c == 0 ? v2f{LH[r], LH[r + 6]} : c == 5 ? v2f{RH[r], RH[r + 6]} : v2f{EV[r][c - 1], EV[r + 6][c - 1]}; \
        IN2[0][c] = v2f{0.0f, c == 0 ? LH[5] : c == 5 ? RH[5] : EV[5][c - 1]};                            \
        IN2[7][c] = v2f{c == 0 ? LH[6] : c == 5 ? RH[6] : EV[6][c - 1], 0.0f};                            \
    }

    // this wave's share of a window's input (the depthwise-5 output): stage wc & 3, rows 24 (wc >> 2) .. + 23 of the lane's half
    float vin[24];
    auto fetch_window = [&](int win) {
        const __amdgpu_buffer_rsrc_t xr = MF_RSRC(X + (size_t)win * 96 * 128, 96 * 128 * 4);
        const int st = wc & 3, rl0 = 24 * (wc >> 2);
        const unsigned vo = (4u * fh * 128) * 4 + c4;
#pragma unroll
        for (int t = 0; t < 24; ++t) {
            const int rl = rl0 + t;
            vin[t] = MF_LD32(xr, vo, ((8 * (rl >> 2) + (rl & 3)) * 128 + 32 * st) * 4);
        }
    };
    // a K loop of KS super-steps over the 96-row stages of the ring, column tile wc of the weights at WR
#define MF_KLOOP96(WR, KS, ACC)                                                                           \
    {                                                                                                     \
        v4f bv[3];                                /* B fragments two super-steps ahead of their use */    \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) bv[q] = MF_LD128(WR, lane16, (wc * (KS) + q) * 1024); \
        _Pragma("unroll") for (int q = 0; q < (KS); ++q) {                                                \
            if (q + 2 < (KS)) bv[(q + 2) % 3] = MF_LD128(WR, lane16, (wc * (KS) + q + 2) * 1024);         \
            const char* const ab = sm + (q >> 2) * kSlotA;                                                \
            _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                               \
                const v4f av = *reinterpret_cast<const v4f*>(ab + ((i == 1 ? ra1 : ra0) ^ ((q & 3) << 5)) + (i == 2 ? 64 * kRowB + kRowB : 0)); \
                MF_MMA(ACC[i], av, bv[q % 3])                                                             \
            }                                                                                             \
        }                                                                                                 \
    }
    // one window through pointwise 5, layer 6 and depthwise 7; its 24 depthwise-7 rows land at rows 24 HALF .. of A7
    auto window_to_a7 = [&](auto half_c, int win_next) {
        constexpr int HALF = decltype(half_c)::value;
        // ---- A5: stage wc & 3 of the ring
        {
            int wbl = wb0;
            asm volatile("" : "+v"(wbl));
            char* const slot = sm + (wc & 3) * kSlotA;
            if (wc < 4) {
#pragma unroll
                for (int t = 0; t < 24; ++t) MF_PUT(slot, t, vin[t])
            } else {
#pragma unroll
                for (int t = 0; t < 24; ++t) MF_PUT(slot, 24 + t, vin[t])
            }
        }
        __syncthreads();                          // A5 published
        // ---- pointwise 5: [96][128] x [128][256], column tile wc
        f32x16 acc[3];
        float zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = zero;
        {
            const __amdgpu_buffer_rsrc_t r5 = MF_RSRC(a.w5, 128 * 256 * 4);
            MF_KLOOP96(r5, 16, acc)
        }
        // ---- depthwise 6 in registers; its outputs wait for the ring: A6 takes the slots A5 is still read from
        {
            float out6[48];
            {
                const __amdgpu_buffer_rsrc_t br = MF_RSRC(a.b5, 1024), tr = MF_RSRC(a.dw6, 10 * 256 * 4);
                const float b = MF_LD32(br, c4, 128 * wc);
                float wt[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) wt[t] = MF_LD32(tr, c4, (t * 256 + 32 * wc) * 4);
                const float shift = MF_LD32(tr, c4, (9 * 256 + 32 * wc) * 4);
                float ev[12][4], lh[12], rh[12];
                MF_TILE_TO_MAP(acc, b, ev)
                MF_HALO(ev, lh, rh)
                v2f in2[8][6];                    // two map rows (y, y + 6) per v_pk_fma_f32 (sepmid.hip)
                MF_ROW_PAIRS(ev, lh, rh, in2)
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
                        out6[4 * y + e] = fmaxf(sacc.x, 0.0f);
                        out6[4 * (y + 6) + e] = fmaxf(sacc.y, 0.0f);
                    }
            }
            __syncthreads();                      // every wave has read A5: the ring is free for A6
            int wbl = wb0;
            asm volatile("" : "+v"(wbl));
            char* const slot = sm + wc * kSlotA;
#pragma unroll
            for (int rl = 0; rl < 48; ++rl) MF_PUT(slot, rl, out6[rl])
        }
        __syncthreads();                          // A6 published
        // ---- pointwise 6: [96][256] x [256][256], column tile wc
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = zero;
        {
            const __amdgpu_buffer_rsrc_t r6 = MF_RSRC(a.w6, 256 * 256 * 4);
            MF_KLOOP96(r6, 32, acc)
        }
        // ---- depthwise 7 (stride 2) in registers -> rows 24 HALF .. of A7, stage wc; the next window's input is requested here
        if (win_next >= 0) fetch_window(win_next);
        {
            const __amdgpu_buffer_rsrc_t br = MF_RSRC(a.b6, 1024), tr = MF_RSRC(a.dw7, 10 * 256 * 4);
            const float b = MF_LD32(br, c4, 128 * wc);
            float wt[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) wt[t] = MF_LD32(tr, c4, (t * 256 + 32 * wc) * 4);
            const float shift = MF_LD32(tr, c4, (9 * 256 + 32 * wc) * 4);
            float ev[12][4], lh[12], rh[12];
            MF_TILE_TO_MAP(acc, b, ev)
            MF_HALO(ev, lh, rh)
            int wbl = wb7;
            asm volatile("" : "+v"(wbl));
            char* const slot = sm + kOffA7 + wc * kSlot7;
            v2f in2[8][6];                        // two output rows (oy, oy + 3) per v_pk_fma_f32
            MF_ROW_PAIRS(ev, lh, rh, in2)
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
                        // row 24 HALF + 4 oyw + 2 fh + j of the 48-row stage: key (row >> 1) & 7 = ((12 HALF + 2 oyw) & 7) | fh
                        *reinterpret_cast<float*>(slot + (wbl ^ (((12 * HALF + 2 * oyw) & 7) << 4)) + (24 * HALF + 4 * oyw + j) * kRowB) =
                            fmaxf(w ? sacc.y : sacc.x, 0.0f);
                    }
                }
        }
        __syncthreads();                          // A7's rows of this window published; every wave has read A6
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
        constexpr int KS = 32;
        const __amdgpu_buffer_rsrc_t r7 = MF_RSRC(a.w7, 256 * 512 * 4);
        v4f bv[3][2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[q][j] = MF_LD128(r7, lane16, ((wc + 8 * j) * KS + q) * 1024);
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            if (q + 2 < KS) {
#pragma unroll
                for (int j = 0; j < 2; ++j) bv[(q + 2) % 3][j] = MF_LD128(r7, lane16, ((wc + 8 * j) * KS + q + 2) * 1024);
            }
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const v4f av = *reinterpret_cast<const v4f*>(sm + kOffA7 + (q >> 2) * kSlot7 + (ra0 ^ ((q & 3) << 5)) + i * 32 * kRowB);
#pragma unroll
                for (int j = 0; j < 2; ++j) { MF_MMA(c7[i][j], av, bv[q % 3][j]) }
            }
        }
        // shift + ReLU: accumulator (i, r, half fh) is row 32 i + 8 (r >> 2) + 4 fh + (r & 3) of the NW x 24 output rows of the
        // windows win, win + 1 (consecutive in Y); rows past them are dropped by the resource's range
        const __amdgpu_buffer_rsrc_t br = MF_RSRC(a.b7, 2048);
        const __amdgpu_buffer_rsrc_t yr = MF_RSRC(Y + (size_t)win * 24 * 512, NW * 24 * 512 * 4);
        const unsigned yo = (4u * fh * 512) * 4 + c4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float b = MF_LD32(br, c4, 128 * (wc + 8 * j));
#pragma unroll
            for (int i = 0; i < NW; ++i)
#pragma unroll
                for (int r = 0; r < (NW == 1 ? 12 : 16); ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(c7[i][j][r] + b, 0.0f)), yr, yo,
                                                          ((32 * i + 8 * (r >> 2) + (r & 3)) * 512 + 32 * (wc + 8 * j)) * 4, 0);
        }
    };

    // a contiguous run of windows per workgroup, taken two at a time (the last one alone when the run is odd)
    const int w_begin = (int)(((long long)blockIdx.x * windows) / gridDim.x);
    const int w_end = (int)(((long long)(blockIdx.x + 1) * windows) / gridDim.x);
    if (w_begin < w_end) fetch_window(w_begin);
    for (int win = w_begin; win < w_end; win += 2) {
        const bool two = win + 1 < w_end;
        window_to_a7(std::integral_constant<int, 0>{}, win + 1 < w_end ? win + 1 : -1);
        if (two) {
            window_to_a7(std::integral_constant<int, 1>{}, win + 2 < w_end ? win + 2 : -1);
            layer7(std::integral_constant<int, 2>{}, win);
        } else {
            layer7(std::integral_constant<int, 1>{}, win);
        }
    }
#undef MF_PUT
#undef MF_MMA
#undef MF_TILE_TO_MAP
#undef MF_HALO
#undef MF_ROW_PAIRS
#undef MF_KLOOP96
}

}  // namespace

// Exact-f32 mode: pointwise 5 -> layer 6 -> depthwise 7 -> pointwise 7 as one launch: in = depthwise-5 output [windows][12][8][128]
// (what l4_f32_kernel writes), out = layer-7 output [windows][6][4][512].  False (nothing launched) when the shapes, the table
// layouts or the fragment-ordered weights are not what the kernel is built for.
bool launch_separable_mid_f32(const float* in, float* out, int windows, const SepLayer& L5, const SepLayer& L6, const SepLayer& L7,
                              hipStream_t stream) {
    if (windows <= 0 || in == out) return false;
    if (L5.cin != 128 || L5.cout != 256 || L5.h_out != 12 || L5.w_out != 8 || L6.cin != 256 || L6.cout != 256 || L6.stride != 1 ||
        L6.h_out != 12 || L6.w_out != 8 || L7.cin != 256 || L7.cout != 512 || L7.stride != 2 || L7.h_out != 6 || L7.w_out != 4)
        return false;
    if (!L5.pw_ffrag || !L6.pw_ffrag || !L7.pw_ffrag) return false;
    if (L6.dw_b != L6.dw_w + 9 * 256 || L7.dw_b != L7.dw_w + 9 * 256) return false;
    MidF32Args a{};
    a.w5 = L5.pw_ffrag; a.w6 = L6.pw_ffrag; a.w7 = L7.pw_ffrag;
    a.b5 = L5.pw_b; a.b6 = L6.pw_b; a.b7 = L7.pw_b;
    a.dw6 = L6.dw_w;
    a.dw7 = L7.dw_w;
    static std::once_flag once[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & 63], [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sep_mid_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kMidF32Lds);
    });
    int grid = cu_count();
    if (grid > windows) grid = windows;
    hipLaunchKernelGGL(sep_mid_f32_kernel, dim3(grid), dim3(512), kMidF32Lds, stream, a, in, out, windows);
    return true;
}

}  // namespace bd
