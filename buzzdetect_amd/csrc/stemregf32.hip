// Layers 1-3 of YAMNet in the exact-f32 mode (bd_set_pointwise_mode 0) with the layer-2 tile handed to layer 3's depthwise in
// REGISTERS: stemreg.hip's kernel with f32 A tiles ([rows][32 or 64 k] f32, 16-byte chunks XOR-swizzled as in stem3_f32_kernel)
// and the two 1x1 convolutions on v_mfma_f32_32x32x2_f32 (round 5).  Everything else - the runs of tiles walked bottom-up, the
// row order of layer 2's A operand that leaves a lane with two rows x 16 columns of its channel, the third row by one
// v_permlane32_swap per column, the carried row, the halo column through LDS, and round 6's four conv1 rows per tile (the two
// kept rows copied inside the band), even-columns-first accumulators with the depthwise of layer 3 on register pairs
// (v_pk_fma_f32) and the taps of depthwise 2 in LDS - is stemreg.hip's; read that file's header first.
// Arithmetic per element is stem3_f32_kernel's, i.e. conv1_kernel / depthwise_kernel / pointwise_kernel's (k pairs {8 s + j,
// 8 s + 4 + j}, j = 0..3, of super-step s in ascending order; epilogue acc + shift, ReLU): bit-identical
// (tests/test_gpu_parity.py::test_fused_f32_mode_equals_one_kernel_per_op).
#include "bd_internal.h"

#include <type_traits>

namespace bd {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// ---- LDS map (bytes): 53 248, three workgroups per CU ----
constexpr int OFF_C1 = 0;                        // conv1 band [6][34][32] f32
constexpr int OFF_A2 = 6 * 34 * 32 * 4;          // 26112: A tile of layer 2 [128 rows][32 k] f32 (128-byte rows, chunk ^ ((row >> 1) & 7));
                                                 //        the log-mel band [13][68] f32 before it
constexpr int OFF_A3 = OFF_A2 + 128 * 128;       // 42496: A tile of layer 3 [32 rows][64 k] f32 (256-byte rows, chunk ^ (row & 15))
constexpr int OFF_HALO = OFF_C1 + (2 * 34 + 1) * 32 * 4;   // column 16 of the three input rows of a half, [column tile][half][3][32 channels] f32: in conv
                                                 // band row 2 (from its column 0 on), dead from the third barrier of a tile to the next tile's phase B
                                                 // (stemreg.hip)
constexpr int OFF_D2W = OFF_A3 + 32 * 256;       // 50688: taps and shift of depthwise 2, [10][32] f32, for the whole run
constexpr int OFF_C1W = OFF_D2W + 10 * 32 * 4;   // 51968: taps and shift of conv1, likewise (as global loads at the top of a tile they stood behind
                                                 // the previous tile's output stores: vmcnt(0), the stores' whole round trip - stemreg.hip)
constexpr int kRegF32Lds = OFF_C1W + 10 * 32 * 4;        // 53248

__global__ __launch_bounds__(256, 3) void stem_reg_f32_kernel(const float* __restrict__ logmel, int patch_step, const WindowMap map, int w0,
                                                              const float* __restrict__ c1_w, const float* __restrict__ c1_b,
                                                              const float* __restrict__ dw2_w, const float* __restrict__ dw2_b,
                                                              const float* __restrict__ W2, const float* __restrict__ pw2_b,
                                                              const float* __restrict__ dw3_w, const float* __restrict__ dw3_b,
                                                              const float* __restrict__ W3, const float* __restrict__ pw3_b,
                                                              float* __restrict__ out, int windows) {
    __shared__ __attribute__((aligned(16))) char smem[kRegF32Lds];
    float (*s_lm)[68] = reinterpret_cast<float (*)[68]>(smem + OFF_A2);
    float (*s_c1)[34][32] = reinterpret_cast<float (*)[34][32]>(smem + OFF_C1);
    char* const s_a2 = smem + OFF_A2;
    float* const s_halo = reinterpret_cast<float*>(smem + OFF_HALO);
    const float* const s_d2 = reinterpret_cast<const float*>(smem + OFF_D2W);
    const float* const s_c1w = reinterpret_cast<const float*>(smem + OFF_C1W);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c4 = tid & 7, col = tid >> 3;               // vector phases: channel quad, map column
    const int g = wave >> 1, wc = wave & 1;               // matrix phases of layer 2: column group, column tile
    const int frow = lane & 31, fh = lane >> 5;

    // global address space spelled out: a pointer that went through an asm is a flat one otherwise (sepf32.hip)
    typedef const __attribute__((address_space(1))) float* gptr;
    typedef const __attribute__((address_space(1))) v4f* gptr4;
    gptr pc1w = (gptr)c1_w, pc1b = (gptr)c1_b, pd2w = (gptr)dw2_w, pd2b = (gptr)dw2_b, pw2 = (gptr)W2, pb2 = (gptr)pw2_b,
         pd3w = (gptr)dw3_w, pd3b = (gptr)dw3_b, pw3 = (gptr)W3, pb3 = (gptr)pw3_b;

    // this workgroup's run of tiles; tile t = window t / 12, band 11 - t % 12 (bottom band first)
    const long long total = 12ll * windows;
    const int t_begin = (int)(blockIdx.x * total / gridDim.x), t_end = (int)((blockIdx.x + 1) * total / gridDim.x);

    // the log-mel band of a tile (one float4 per thread) is requested a tile ahead, in front of the previous tile's output
    // stores (a load issued behind them could only be waited for together with them: sepf32.hip)
    float4 lmv;
    bool lm_ok = false;
    auto prefetch = [&](int win, int r_first) {
        const float* patch = logmel + window_frame(map, w0 + win, patch_step) * BD_MEL_BANDS;
        const int j = tid >> 4, q = tid & 15;              // band row, float4 of the row (rows 13 .. 15: nobody's)
        const int ih = 2 * r_first - 2 + j;
        lm_ok = j < 13 && ih >= 0 && ih < BD_PATCH_FRAMES;
        const int ihc = ih < 0 ? 0 : ih >= BD_PATCH_FRAMES ? BD_PATCH_FRAMES - 1 : ih;
        lmv = reinterpret_cast<const float4*>(patch + ihc * BD_MEL_BANDS)[q];
    };

    // ---- layer-2 rows r_first .. r_first + 3 of a window -> ev[t][k]: this lane's channel (32 wc + frow) at row 2 fh + t, as
    //      PAIRS of columns: k < 4: columns 16 g + 4 k, + 4 k + 2 (the even ones), k >= 4: 16 g + 4 (k - 4) + 1, + 3 (the odd ones)
    //      (phases A - D and the 1x1 convolution's epilogue).  c1_new: how many of the C1R conv1 rows are computed here - 4 when
    //      the tile below left rows r_first + 3, r_first + 4 in band rows 4, 5, else all
    v2f ev[2][8];
    auto front = [&](auto rows_c, int r_first, int c1_new) {
        constexpr int ROWS = decltype(rows_c)::value;      // 4, or 1: only row r_first (what a run that starts inside a window needs)
        constexpr int C1R = ROWS + 2, LMR = 2 * C1R + 1;
        // ---- A: log-mel rows 2 (r_first - 1) .. + 12 (prefetched) ----
        if (tid < LMR * 16) {
            float4 v = lmv;
            if (!lm_ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&s_lm[tid >> 4][(tid & 15) * 4]) = v;
        } else if (tid >= 256 - LMR) {                     // mel band 64 of every row: the zero to the right of the patch (the band
            float z;                                       // shares its bytes with the A tile; the zero is made here: stemreg.hip)
            asm volatile("v_mov_b32 %0, 0" : "=v"(z));
            *reinterpret_cast<float4*>(&s_lm[255 - tid][64]) = make_float4(z, z, z, z);
        }
        __syncthreads();                                   // (the halo columns of the conv1 band were zeroed once, before the run)
        v4f c1wt[9];                                       // (behind the barrier: the run's first tile finds the taps written)
#pragma unroll
        for (int t = 0; t < 9; ++t) c1wt[t] = *reinterpret_cast<const v4f*>(s_c1w + t * 32 + c4 * 4);
        const v4f c1bias = *reinterpret_cast<const v4f*>(s_c1w + 9 * 32 + c4 * 4);
        // ---- B: conv1 rows r_first - 1 .. r_first + 4 (conv1_kernel's chain: taps in (kh, kw) order, a tap row past the patch
        //         skipped; a conv1 row outside the map is the depthwise's zero padding) ----
        {
            float lm[3][3];
            const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) lm[0][kw] = s_lm[0][2 * col + kw];
#pragma unroll
            for (int i = 0; i < C1R; ++i) {
                if (i >= c1_new) break;              // (uniform) rows r_first + 3, r_first + 4 are the tile below's rows -1, 0
                const int c1r = r_first - 1 + i;
#pragma unroll
                for (int kh = 1; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) lm[kh][kw] = s_lm[2 * i + kh][2 * col + kw];
                if (c1r >= 0 && c1r < 48) {          // the same for the whole workgroup: a scalar branch
                    v4f acc = c1bias;
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {
                        if (2 * c1r + kh >= BD_PATCH_FRAMES) continue;
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const float v = lm[kh][kw];
                            acc = __builtin_elementwise_fma(v4f{v, v, v, v}, c1wt[kh * 3 + kw], acc);
                        }
                    }
                    v4f r4;
                    r4.x = fmaxf(acc.x, 0.0f);
                    r4.y = fmaxf(acc.y, 0.0f);
                    r4.z = fmaxf(acc.z, 0.0f);
                    r4.w = fmaxf(acc.w, 0.0f);
                    *reinterpret_cast<v4f*>(&s_c1[i][col + 1][c4 * 4]) = r4;
                } else {
                    *reinterpret_cast<v4f*>(&s_c1[i][col + 1][c4 * 4]) = zero4;
                }
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) lm[0][kw] = lm[2][kw];
            }
        }
        __syncthreads();
        // ---- C: depthwise 2, four rows -> f32 A tile; position (y, x) at row 32 (2 (x >> 4) + (y & 1)) + 8 (r >> 2)
        //         + 4 (y >> 1) + (r & 3), r = the accumulator register of column x & 15 (even columns in r = 0 .. 7) ----
        {
            v4f d2wt[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) d2wt[t] = *reinterpret_cast<const v4f*>(s_d2 + t * 32 + c4 * 4);
            const v4f d2bias = *reinterpret_cast<const v4f*>(s_d2 + 9 * 32 + c4 * 4);
            v4f cv[3][3];
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) cv[kh][kw] = *reinterpret_cast<const v4f*>(&s_c1[kh][col + kw][c4 * 4]);
            const int racc = ((col & 1) << 3) + ((col & 15) >> 1);            // accumulator register of this column: evens first
            const int rbase = 64 * (col >> 4) + 8 * (racc >> 2) + (racc & 3);
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) cv[2][kw] = *reinterpret_cast<const v4f*>(&s_c1[r + 2][col + kw][c4 * 4]);
                v4f acc = d2bias;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) acc = __builtin_elementwise_fma(cv[kh][kw], d2wt[kh * 3 + kw], acc);
                acc.x = fmaxf(acc.x, 0.0f);
                acc.y = fmaxf(acc.y, 0.0f);
                acc.z = fmaxf(acc.z, 0.0f);
                acc.w = fmaxf(acc.w, 0.0f);
                const int row = rbase + 32 * (r & 1) + 4 * (r >> 1);
                *reinterpret_cast<v4f*>(s_a2 + row * 128 + ((c4 ^ ((row >> 1) & 7)) << 4)) = acc;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    cv[0][kw] = cv[1][kw];
                    cv[1][kw] = cv[2][kw];
                }
            }
        }
        // this lane's layer-2 weights: output channel 32 wc + frow, k = 8 q + 4 fh .. + 3
        v4f w2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) w2[q] = *(gptr4)(pw2 + (size_t)(wc * 32 + frow) * 32 + 8 * q + 4 * fh);
        __syncthreads();
        if constexpr (ROWS == 4) {
            // conv1 rows r_first - 1, r_first are rows r_first' + 3, r_first' + 4 of the tile above: into band rows 4, 5 (every
            // thread moves the two values it wrote in phase B; the next reader is the tile above's phase C, two barriers on)
            const v4f k0 = *reinterpret_cast<const v4f*>(&s_c1[0][col + 1][c4 * 4]);
            const v4f k1 = *reinterpret_cast<const v4f*>(&s_c1[1][col + 1][c4 * 4]);
            *reinterpret_cast<v4f*>(&s_c1[4][col + 1][c4 * 4]) = k0;
            *reinterpret_cast<v4f*>(&s_c1[5][col + 1][c4 * 4]) = k1;
        }
        // ---- D: [128][32] x [32][64]: wave (g, wc) = row tiles 2 g, 2 g + 1 against column tile wc; lane = output channel ----
        const float b2 = pb2[wc * 32 + frow];
        f32x16 acc2[2];
#pragma unroll
        for (int t = 0; t < (ROWS == 1 ? 1 : 2); ++t) {    // (ROWS == 1: row 0 is accumulator 0 of half 0; the A tile's other rows are stale)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[t][r] = 0.0f;
            const int row = (2 * g + t) * 32 + frow;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const v4f a4 = *reinterpret_cast<const v4f*>(s_a2 + row * 128 + (((2 * q + fh) ^ ((row >> 1) & 7)) << 4));
                acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, w2[q].x, acc2[t], 0, 0, 0);
                acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, w2[q].y, acc2[t], 0, 0, 0);
                acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, w2[q].z, acc2[t], 0, 0, 0);
                acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, w2[q].w, acc2[t], 0, 0, 0);
            }
        }
        // epilogue: accumulator (t, r, half fh) is row 32 (2 g + t) + 8 (r >> 2) + 4 fh + (r & 3) = position (2 fh + t, 16 g + column
        // of register r)
#pragma unroll
        for (int t = 0; t < (ROWS == 1 ? 1 : 2); ++t)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const v2f v = v2f{acc2[t][2 * k], acc2[t][2 * k + 1]} + v2f{b2, b2};
                ev[t][k] = v2f{fmaxf(v.x, 0.0f), fmaxf(v.y, 0.0f)};
            }
    };

    v2f carry[8];                                          // half 0: row 0 of the tile below (= row 4 of this one), in ev's pairs
    if (t_begin < t_end) {
        if (tid >= 96 && tid < 96 + 160) {                 // conv1's and depthwise 2's taps [9][32] and shift [32] into LDS for the whole run
            const int which = (tid - 96) / 80, row = ((tid - 96) % 80) >> 3, cc = tid & 7;
            gptr w = which ? pd2w : pc1w, bs = which ? pd2b : pc1b;
            *reinterpret_cast<v4f*>(smem + (which ? OFF_D2W : OFF_C1W) + (row * 32 + cc * 4) * 4) = row < 9 ? *(gptr4)(w + row * 32 + cc * 4) : *(gptr4)(bs + cc * 4);
        }
        if (tid < 6 * 2 * 8) {                             // columns -1 and 32 of the conv1 band: zero for the whole run
            const int r = tid / 16, side = (tid >> 3) & 1, cc = tid & 7;
            *reinterpret_cast<float4*>(&s_c1[r][side ? 33 : 0][cc * 4]) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        {
            const int win0 = t_begin / 12, ob0 = 11 - t_begin % 12;
            if (ob0 != 11) {                               // the run starts inside a window: the front half of the tile below
                prefetch(win0, 4 * ob0 + 4);
                front(std::integral_constant<int, 1>{}, 4 * ob0 + 4, 3);
#pragma unroll
                for (int k = 0; k < 8; ++k) carry[k] = ev[0][k];
                __syncthreads();                           // (its A tile has been read: the next front pass may write the band)
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) carry[k] = v2f{0.0f, 0.0f};      // row 48 of a window is the zero padding
            }
            prefetch(win0, 4 * ob0);
        }
#pragma unroll 1
        for (int t = t_begin; t < t_end; ++t) {
            asm volatile("" : "+s"(pc1w), "+s"(pc1b), "+s"(pd2w), "+s"(pd2b), "+s"(pw2), "+s"(pb2));
            asm volatile("" : "+s"(pd3w), "+s"(pd3b), "+s"(pw3), "+s"(pb3));
            const int win = t / 12, ob = 11 - t % 12;
            // (the tile below - same window, same run - has left conv1 rows 4 ob + 3, 4 ob + 4 in band rows 4, 5)
            front(std::integral_constant<int, 4>{}, 4 * ob, (t == t_begin || ob == 11) ? 6 : 4);

            // ---- the third input row of each half: row 2 (the other half's first row) for half 0, the carried row 4 for half 1:
            //      v_permlane32_swap vdst, src trades lanes 32-63 of vdst against lanes 0-31 of src ----
            v2f x2[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                // (the operands go through an empty asm: stemreg.hip - with the two elements of a vector as operands this compiler
                //  merges the two swaps into one)
                float e0 = ev[0][k].x, e1 = ev[0][k].y, c0 = carry[k].x, c1 = carry[k].y;
                asm("" : "+v"(e0), "+v"(e1), "+v"(c0), "+v"(c1));
                const auto s0 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, e0), __builtin_bit_cast(unsigned, c0), false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, e1), __builtin_bit_cast(unsigned, c1), false, false);
                x2[k] = v2f{__builtin_bit_cast(float, fh ? (unsigned)s0[0] : (unsigned)s0[1]),
                            __builtin_bit_cast(float, fh ? (unsigned)s1[0] : (unsigned)s1[1])};
            }
            // column 16 of the three rows goes from group 1 to group 0 through LDS
            if (g == 1) {
                float* const hw = s_halo + ((wc * 2 + fh) * 3) * 32 + frow;
                hw[0] = ev[0][0].x;
                hw[32] = ev[1][0].x;
                hw[64] = x2[0].x;
            }
            // this lane's layer-3 taps (channel 32 wc + frow), in flight behind the barrier
            float d3w[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) d3w[k] = pd3w[k * 64 + wc * 32 + frow];
            const float d3b = pd3b[wc * 32 + frow];
            __syncthreads();                               // halo column written; every wave has read the A tile of layer 2
            float hal[3] = {0.0f, 0.0f, 0.0f};             // column 16 g + 16: the zero padding for g = 1
            if (g == 0) {
                const float* const hr = s_halo + ((wc * 2 + fh) * 3) * 32 + frow;
                hal[0] = hr[0];
                hal[1] = hr[32];
                hal[2] = hr[64];
            }
            // ---- F: depthwise 3, stride 2, in registers: outputs (row fh, columns 8 g + 2 k, + 2 k + 1) from rows 2 fh + kh, columns
            //         16 g + 4 k + kw and 16 g + 4 k + 2 + kw (stemreg.hip) -> f32 A tile of layer 3 ----
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v2f a = v2f{d3b, d3b};
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const v2f* const row = kh == 0 ? ev[0] : kh == 1 ? ev[1] : x2;
                    const v2f shifted = v2f{row[k].y, k < 3 ? row[k < 3 ? k + 1 : 0].x : hal[kh]};
                    a = __builtin_elementwise_fma(row[k], v2f{d3w[kh * 3], d3w[kh * 3]}, a);
                    a = __builtin_elementwise_fma(row[4 + k], v2f{d3w[kh * 3 + 1], d3w[kh * 3 + 1]}, a);
                    a = __builtin_elementwise_fma(shifted, v2f{d3w[kh * 3 + 2], d3w[kh * 3 + 2]}, a);
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int pos = fh * 16 + 8 * g + 2 * k + e, k3 = 32 * wc + frow;
                    *reinterpret_cast<float*>(smem + OFF_A3 + pos * 256 + (((k3 >> 2) ^ (pos & 15)) << 4) + (k3 & 3) * 4) = fmaxf(e ? a.y : a.x, 0.0f);
                }
            }
            // what the tile above takes over: this tile's first row (the next window's bottom tile: zeros, its padding row)
#pragma unroll
            for (int k = 0; k < 8; ++k) carry[k] = ob == 0 ? v2f{0.0f, 0.0f} : ev[0][k];
            // the next tile's log-mel band (the last tile of the run: its own again), in flight behind the matrix phase
            {
                const int tn = t + 1 < t_end ? t + 1 : t;
                prefetch(tn / 12, 4 * (11 - tn % 12));
            }
            // this lane's layer-3 weights (behind the depthwise: 32 registers that would not fit beside its inputs)
            v4f w3[8];                                      // output channel 32 wave + frow, k = 8 q + 4 fh .. + 3
#pragma unroll
            for (int q = 0; q < 8; ++q) w3[q] = *(gptr4)(pw3 + (size_t)(32 * wave + frow) * 64 + 8 * q + 4 * fh);
            const int n3 = 32 * wave + frow;
            const float b3 = pb3[n3];
            __syncthreads();                               // the A tile of layer 3 is complete
            // ---- G: [32][64] x [64][128], one 32 x 32 tile per wave ----
            f32x16 acc3;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[r] = 0.0f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const v4f a4 = *reinterpret_cast<const v4f*>(smem + OFF_A3 + frow * 256 + (((2 * q + fh) ^ (frow & 15)) << 4));
                acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, w3[q].x, acc3, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, w3[q].y, acc3, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, w3[q].z, acc3, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, w3[q].w, acc3, 0, 0, 0);
            }
            // the loads are waited for HERE, in front of the stores (sepf32.hip: vmcnt counts both kinds, out of order)
            asm volatile("" : "+v"(lmv.x), "+v"(lmv.y), "+v"(lmv.z), "+v"(lmv.w));
            // ---- H: bias + ReLU, [32][128] block of the layer-3 output (rows are consecutive NHWC positions) ----
            float* dst3 = out + (((size_t)win * 24 + 2 * ob) * 16) * 128;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 4 * fh + (r & 3) + 8 * (r >> 2);
                dst3[(size_t)m * 128 + n3] = fmaxf(acc3[r] + b3, 0.0f);
            }
            // (no barrier: the next tile's phases A and B write the log-mel band and the conv band, which lie below the A tile
            //  of layer 3; its phase C - the layer-2 A tile - and the halo words are one / three barriers away)
        }
    }
}

}  // namespace

// Exact-f32 mode: layers 1-3 complete with the layer-2 tile handed over in registers: out = [windows][24][16][128].
void launch_stem_reg_f32(const float* logmel, int patch_step, const WindowMap& map, int w0, int windows, const float* c1_w,
                         const float* c1_b, const SepLayer& L2, const SepLayer& L3, float* out, hipStream_t stream) {
    if (windows <= 0) return;
    long long grid = 3ll * cu_count();                      // three workgroups per CU, each with a contiguous run of the 12 x windows tiles
    if (grid > 12ll * windows) grid = 12ll * windows;
    hipLaunchKernelGGL(stem_reg_f32_kernel, dim3((unsigned)grid), dim3(256), 0, stream, logmel, patch_step, map, w0, c1_w, c1_b, L2.dw_w,
                       L2.dw_b, L2.pw_wt, L2.pw_b, L3.dw_w, L3.dw_b, L3.pw_wt, L3.pw_b, out, windows);
}

}  // namespace bd
