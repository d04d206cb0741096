// Layers 8-12 of YAMNet + the depthwise of layer 13 as ONE launch in the exact-f32 mode (bd_set_pointwise_mode 0): sepchip.hip's
// scheme - the tile between two 512 -> 512 layers stays on the CU: accumulators -> depthwise in registers -> an LDS ring of 13
// stage tiles + 3 stages pending in registers - with every product on v_mfma_f32_32x32x2_f32 (round 5).
//
// Why it pays more here than in the split-f16 mode: the f32 matrix instruction is 16/3 x slower per product, so a layer's K loop
// is 196 k cycles per tile and SIMD against the 20 k of depthwise, publication and barriers around it - the matrix pipe is busy
// ~90 % of the launch where pointwise_kernel<96, 128, ..> with the next depthwise in its epilogue (one launch per layer, the
// tile through LDS and HBM) reaches 67-70 %.
//
// What differs from sepchip.hip:
//   * a stage tile is [96 rows][32 k] f32 = 128-byte rows (the same 12 416 bytes: rows 48.. pushed back by one row so that the
//     two half-waves of a publishing wave never meet in a bank); the 16-byte chunk of a row is XORed with (row >> 1) & 7
//     (stem3_f32_kernel's swizzle: the sixteen rows of a ds_read_b128 lane group on sixteen different slots of the bank row);
//   * a published value is ONE ds_write_b32 (no split, no range guard);
//   * the K loop: per super-step of 8 k one ds_read_b128 per row tile and one 16-byte global load per column tile from the
//     weights in fragment order (SepLayer::pw_ffrag: a wave's load is one contiguous KiB), then 3 x 2 x 4 matrix instructions;
//     a super-step is 1 536 cycles of matrix work, so the loads of the next one are simply requested in front of it.
// Arithmetic is that of depthwise_kernel + pointwise_kernel bit for bit: the depthwise sums shift + taps in row-major tap order
// with fmaf, ReLU; per accumulator the k pairs {8 s + j, 8 s + 4 + j}, j = 0..3, of super-step s = 0..63 in ascending order
// (pointwise_kernel's operand map); epilogue acc + shift, ReLU.  Taps outside the map are skipped (sepchip.hip on why that is
// the multiplied zero's result).  tests/test_gpu_parity.py::test_fused_f32_mode_equals_one_kernel_per_op covers it.
#include "bd_internal.h"

#include <mutex>

namespace bd {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kF32RowBytes = 128;
constexpr int kF32SlotBytes = 97 * kF32RowBytes;       // 12 416

constexpr int kF32MaxLayers = 5;
struct ChipChainF32 {
    const float* dw_w[kF32MaxLayers];           // [9][512] depthwise taps ([512] shift behind them)
    const float* wfrag[kF32MaxLayers];          // pointwise weights, fragment order [512/32][512/8][64][4]
    const float* pw_b[kF32MaxLayers];           // [512]
    const float* ndw_w;                         // taps [9][512] + shift [512] of the stride-2 depthwise behind the run
};
static_assert(sizeof(ChipChainF32) == (3 * kF32MaxLayers + 1) * 8, "three tables of five pointers + one");
// pointer `field` of layer `li` by a scalar load from the kernel-argument segment (the chain is the FIRST argument; sepchip.hip)
template <typename T>
__device__ __forceinline__ const T* chain_f32_ptr(int field, int li) {
    typedef const __attribute__((address_space(4))) unsigned long long* kptr;
    const kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    return reinterpret_cast<const T*>(ka[field * kF32MaxLayers + li]);
}

// DW0: the run's input is the first layer's depthwise OUTPUT (the launch in front applied it in its epilogue); else its input.
template <int NSLOT, bool NDW, bool DW0>
__global__ __launch_bounds__(512, 2) void sep_chip_f32_kernel(const ChipChainF32 ch, const float* X, float* Y, int nl, long long M) {
    static_assert(NSLOT >= 9 && NSLOT <= 16, "ring size");
    constexpr int K = 512, KS = K / 8;             // super-steps of 8 k
    constexpr int NPEND = 16 - NSLOT;
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wc = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fh = lane >> 5;
    const long long m0 = (long long)blockIdx.x * 96;

    // publisher: lane (k = frow, fh) writes element k of rows 48 fh + rl; key of row 48 fh + rl = (rl >> 1) & 7 for both halves
    const int wb0 = fh * (48 * kF32RowBytes + kF32RowBytes) + ((frow >> 2) << 4) + 4 * (frow & 3);
    // reader: lane (frow, fh), super-step s of a stage: chunk (2 s + fh) ^ key, key = (frow >> 1) & 7 for every row tile
    const int ra0 = frow * kF32RowBytes + ((fh ^ ((frow >> 1) & 7)) << 4);
    const int ra1 = ra0 + 32 * kF32RowBytes + (frow >= 16 ? kF32RowBytes : 0);
    const unsigned lane16 = lane * 16, c4 = frow * 4;

    f32x16 acc[3][2];
    float pend[48], out0[48];

#define F32_RSRC(P, BYTES) __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), 0, BYTES, 0x00020000)
#define F32_LD(R, VO, SO) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(R, VO, SO, 0))
    // depthwise 3 x 3 + shift + ReLU of column block J (stage ST) of the tile held as in2[y][x] = (window 2 fh, window 2 fh + 1)
#define F32_DW(J, ST, TAPS)                                                                               \
    {                                                                                                     \
        float wt[9];                                                                                      \
        _Pragma("unroll") for (int t = 0; t < 9; ++t) wt[t] = F32_LD(TAPS, c4, (t * K + 32 * (ST)) * 4);  \
        const float shift = F32_LD(TAPS, c4, (9 * K + 32 * (ST)) * 4);                                    \
        _Pragma("unroll") for (int y = 0; y < 6; ++y)                                                     \
            _Pragma("unroll") for (int x = 0; x < 4; ++x) {                                               \
                v2f a = {shift, shift};                                                                   \
                _Pragma("unroll") for (int kh = 0; kh < 3; ++kh)                                          \
                    _Pragma("unroll") for (int kw = 0; kw < 3; ++kw) {                                    \
                        const int iy = y + kh - 1, ix = x + kw - 1;                                       \
                        if (iy < 0 || iy >= 6 || ix < 0 || ix >= 4) continue;                             \
                        a = __builtin_elementwise_fma(in2[iy][ix], v2f{wt[kh * 3 + kw], wt[kh * 3 + kw]}, a); \
                    }                                                                                     \
                if ((J) == 1) {                                                                           \
                    pend[4 * y + x] = fmaxf(a.x, 0.0f);                                                   \
                    pend[24 + 4 * y + x] = fmaxf(a.y, 0.0f);                                              \
                } else {                                                                                  \
                    out0[4 * y + x] = fmaxf(a.x, 0.0f);                                                   \
                    out0[24 + 4 * y + x] = fmaxf(a.y, 0.0f);                                              \
                }                                                                                         \
            }                                                                                             \
    }
#define F32_PUT(SLOT, RL, V) *reinterpret_cast<float*>((SLOT) + (wbl ^ ((((RL) >> 1) & 7) << 4)) + (RL) * kF32RowBytes) = (V);
#define F32_PUBLISH()                                                                                     \
    {                                                                                                     \
        int wbl = wb0;                                                                                    \
        asm volatile("" : "+v"(wbl));                                                                     \
        char* const slot0 = sm + wc * kF32SlotBytes;                                                      \
        _Pragma("unroll") for (int rl = 0; rl < 48; ++rl) F32_PUT(slot0, rl, out0[rl])                    \
        if (wc + 8 < NSLOT) {                                                                             \
            char* const slot1 = sm + (wc + 8) * kF32SlotBytes;                                            \
            _Pragma("unroll") for (int rl = 0; rl < 48; ++rl) F32_PUT(slot1, rl, pend[rl])                \
        }                                                                                                 \
    }

    // ---------------------------------------------------------------------- the run's input (a resource over the tile's valid rows:
    // a row past the end of the batch reads as zero)
    const long long rows_left = M - m0;
    const unsigned tile_bytes = (unsigned)(rows_left < 96 ? rows_left : 96) * (K * 4);
    {
        const __amdgpu_buffer_rsrc_t xrs = F32_RSRC(X + (size_t)m0 * K, tile_bytes);
        const __amdgpu_buffer_rsrc_t taps0 = F32_RSRC(ch.dw_w[0], 10 * K * 4);
        const unsigned xo = (48u * fh * K) * 4 + c4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int st = j ? wc + 8 : wc;
            if constexpr (DW0) {
#pragma unroll
                for (int rl = 0; rl < 48; ++rl) {
                    const float v = F32_LD(xrs, xo, (rl * K + 32 * st) * 4);
                    if (j) pend[rl] = v;
                    else out0[rl] = v;
                }
            } else {
                v2f in2[6][4];
#pragma unroll
                for (int y = 0; y < 6; ++y)
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        in2[y][x].x = F32_LD(xrs, xo, ((4 * y + x) * K + 32 * st) * 4);
                        in2[y][x].y = F32_LD(xrs, xo, ((24 + 4 * y + x) * K + 32 * st) * 4);
                    }
                if (j == 0) F32_DW(0, st, taps0)
                else F32_DW(1, st, taps0)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        F32_PUBLISH()
    }

    for (int li = 0; li < nl; ++li) {
        const float* const Wf = chain_f32_ptr<float>(1, li);
        // B fragments: column tile (wc, wc + 8), super-step S -> ((tile * KS + S) * 64 + lane) * 16 bytes.  The first super-step's
        // are requested in FRONT of the barrier that publishes the A operand (round 6, as in sepchip.hip)
        const __amdgpu_buffer_rsrc_t br = F32_RSRC(Wf, K * K * 4);
        const int btile = wc * (KS * 1024);
        constexpr int jstep = 8 * KS * 1024;
#define F32_LB(S, J) __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(br, lane16, btile + (J) * jstep + (S) * 1024, 0))
        v4f bn[2];                                // the B fragments of the next super-step
        bn[0] = F32_LB(0, 0);
        bn[1] = F32_LB(0, 1);
        __syncthreads();                          // stages 0 .. NSLOT - 1 of layer li published
        float zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));      // (a zero the compiler cannot form early: sepchip.hip)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = zero;
        // stages FROM .. TO - 1: four super-steps each
#define F32_STAGES(FROM, TO)                                                                              \
    _Pragma("nounroll") for (int kk = (FROM); kk < (TO); ++kk) {                                          \
        const char* const abase = sm + (kk < NSLOT ? kk : kk - NSLOT) * kF32SlotBytes;                    \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                   \
            const v4f b0 = bn[0], b1 = bn[1];                                                             \
            const int sn = 4 * kk + s + 1 < KS ? 4 * kk + s + 1 : 0;        /* (behind the last one: super-step 0 again, unused) */ \
            bn[0] = F32_LB(sn, 0);                                                                        \
            bn[1] = F32_LB(sn, 1);                                                                        \
            v4f av[3];                                                                                    \
            av[0] = *reinterpret_cast<const v4f*>(abase + (ra0 ^ (s << 5)));                              \
            av[1] = *reinterpret_cast<const v4f*>(abase + (ra1 ^ (s << 5)));                              \
            av[2] = *reinterpret_cast<const v4f*>(abase + (ra0 ^ (s << 5)) + 64 * kF32RowBytes + kF32RowBytes); \
            _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                               \
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, b0.x, acc[i][0], 0, 0, 0);      \
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, b0.y, acc[i][0], 0, 0, 0);      \
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, b0.z, acc[i][0], 0, 0, 0);      \
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, b0.w, acc[i][0], 0, 0, 0);      \
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, b1.x, acc[i][1], 0, 0, 0);      \
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, b1.y, acc[i][1], 0, 0, 0);      \
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, b1.z, acc[i][1], 0, 0, 0);      \
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, b1.w, acc[i][1], 0, 0, 0);      \
            }                                                                                             \
        }                                                                                                 \
    }
        F32_STAGES(0, NPEND)
        if constexpr (NPEND > 0) {
            __syncthreads();                      // every wave has read stages 0 .. NPEND - 1: their slots are free
            if (wc >= 8 - NPEND) {
                char* const slot = sm + (wc + 8 - NSLOT) * kF32SlotBytes;
                int wbl = wb0;
                asm volatile("" : "+v"(wbl));
#pragma unroll
                for (int rl = 0; rl < 48; ++rl) F32_PUT(slot, rl, pend[rl])
            }
            __syncthreads();                      // pending stages published
        }
        F32_STAGES(NPEND, 16)
#undef F32_STAGES
#undef F32_LB
        if (li + 1 == nl) break;

        // ------------------------------------------------------------------ depthwise of layer li + 1 on the accumulators
        const __amdgpu_buffer_rsrc_t taps = F32_RSRC(chain_f32_ptr<float>(0, li + 1), 10 * K * 4);
        const __amdgpu_buffer_rsrc_t pbr = F32_RSRC(chain_f32_ptr<float>(2, li), K * 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int st = j ? wc + 8 : wc;       // stage of layer li + 1 = column block of layer li
            const float b = F32_LD(pbr, c4, 128 * st);
            // stacked map row R = 8 i + 2 (r >> 2) + owner half, x = r & 3 (sepchip.hip): pair q of windows 0-1 is accumulator
            // quad (q >> 2, q & 3), the same pair of windows 2-3 is quad ((q + 6) >> 2, (q + 6) & 3)
            float ev[12][4];
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const float a01 = fmaxf(acc[q >> 2][j][4 * (q & 3) + x] + b, 0.0f);
                    const float a23 = fmaxf(acc[(q + 6) >> 2][j][4 * ((q + 6) & 3) + x] + b, 0.0f);
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
            if (j == 0) F32_DW(0, st, taps)
            else F32_DW(1, st, taps)
        }
        __syncthreads();                          // every wave has read every stage: the ring is free for this layer's tile
        F32_PUBLISH()
    }
#undef F32_DW
#undef F32_PUBLISH
#undef F32_PUT

    // ---------------------------------------------------------------------- the run's output: shift + ReLU from the accumulators
    {
        const __amdgpu_buffer_rsrc_t pbr = F32_RSRC(chain_f32_ptr<float>(2, nl - 1), K * 4);
        if constexpr (!NDW) {
            const __amdgpu_buffer_rsrc_t yrs = F32_RSRC(Y + (size_t)m0 * K, tile_bytes);
            const unsigned yo = (4u * fh * K) * 4 + c4;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int st = j ? wc + 8 : wc;
                const float b = F32_LD(pbr, c4, 128 * st);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(acc[i][j][r] + b, 0.0f)), yrs, yo,
                                                              ((32 * i + 8 * (r >> 2) + (r & 3)) * K + 32 * st) * 4, 0);
            }
        } else {
            // the tile's windows are rows m0 / 24 .. + 3 of the [windows][6][512] output
            const __amdgpu_buffer_rsrc_t yrs = F32_RSRC(Y + (size_t)(m0 / 4) * K, tile_bytes / 4);
            const __amdgpu_buffer_rsrc_t ntaps = F32_RSRC(ch.ndw_w, 10 * K * 4);
            const unsigned yo = (12u * fh * K) * 4 + c4;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int st = j ? wc + 8 : wc;
                const float b = F32_LD(pbr, c4, 128 * st);
                float wt[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) wt[t] = F32_LD(ntaps, c4, (t * K + 32 * st) * 4);
                const float shift = F32_LD(ntaps, c4, (9 * K + 32 * st) * 4);
                float ev[12][4];
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        const float a01 = fmaxf(acc[q >> 2][j][4 * (q & 3) + x] + b, 0.0f);
                        const float a23 = fmaxf(acc[(q + 6) >> 2][j][4 * ((q + 6) & 3) + x] + b, 0.0f);
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
                        for (int w = 0; w < 2; ++w)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(w ? a.y : a.x, 0.0f)), yrs, yo,
                                                                  ((6 * w + 2 * oy + ox) * K + 32 * st) * 4, 0);
                    }
            }
        }
    }
#undef F32_RSRC
#undef F32_LD
}

template <bool NDW, bool DW0>
void launch_chip_f32(const float* in, float* out, const SepLayer* L, int nl, long long M, const float* ndw_w, hipStream_t stream) {
    ChipChainF32 ch{};
    for (int i = 0; i < nl; ++i) {
        ch.dw_w[i] = L[i].dw_w;
        ch.wfrag[i] = L[i].pw_ffrag;
        ch.pw_b[i] = L[i].pw_b;
    }
    ch.ndw_w = ndw_w;
    constexpr int NSLOT = 13;
    constexpr int lds = NSLOT * kF32SlotBytes;
    static_assert(lds <= 160 * 1024, "the ring must fit the CU's LDS");
    static std::once_flag once[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & 63], [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sep_chip_f32_kernel<NSLOT, NDW, DW0>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    });
    const long long tiles = (M + 95) / 96;
    hipLaunchKernelGGL((sep_chip_f32_kernel<NSLOT, NDW, DW0>), dim3((unsigned)tiles), dim3(512), lds, stream, ch, in, out, nl, M);
}

}  // namespace

// Exact-f32 mode: a run of stride-1 512 -> 512 layers on the 6 x 4 map with the tiles between its layers kept on the CU.
// `in` = the first layer's depthwise OUTPUT if dw0_done (the launch in front applied it), else its input; with `next` (the
// stride-2 layer behind the run) next's depthwise is applied in the epilogue and out = [windows][3][2][512], else out = the last
// layer's output (in and out may then be the same buffer).  False (nothing launched): shape not covered, a layer without its
// fragment-ordered weights, or a shift table that does not follow its taps (one [10][512] resource reads both).
bool launch_separable_chip_f32(const float* in, float* out, int windows, const SepLayer* L, int nl, hipStream_t stream,
                               const SepLayer* next, bool dw0_done) {
    if (nl < 1 || nl > kF32MaxLayers || windows <= 0) return false;
    for (int i = 0; i < nl; ++i)
        if (L[i].cin != 512 || L[i].cout != 512 || L[i].stride != 1 || L[i].h_in != 6 || L[i].w_in != 4 || !L[i].pw_ffrag ||
            L[i].dw_b != L[i].dw_w + 9 * 512)
            return false;
    if (next && (next->dw_b != next->dw_w + 9 * 512 || next->cin != 512 || next->stride != 2 || in == out)) return false;
    const long long M = (long long)windows * 24;
    const float* nw = next ? next->dw_w : nullptr;
    if (next) {
        if (dw0_done) launch_chip_f32<true, true>(in, out, L, nl, M, nw, stream);
        else launch_chip_f32<true, false>(in, out, L, nl, M, nw, stream);
    } else {
        if (dw0_done) launch_chip_f32<false, true>(in, out, L, nl, M, nw, stream);
        else launch_chip_f32<false, false>(in, out, L, nl, M, nw, stream);
    }
    return true;
}

}  // namespace bd
