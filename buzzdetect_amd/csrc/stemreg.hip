// Layers 1-3 of YAMNet (yamnet.py:79-81: conv 3x3 s2 -> [dw 3x3, pw 32->64] -> [dw 3x3 s2, pw 64->128]) with the layer-2 output
// kept in REGISTERS between its 1x1 convolution and the stride-2 depthwise of layer 3 (round 5), the way sepmid.hip / sepchip.hip
// hand a tile from a layer to the next depthwise.
//
// stem3_kernel writes the layer-2 tile (160 positions x 64 channels, f32) to LDS, waits, and reads it back nine times for the
// depthwise of layer 3; a fifth of the tile is the row it shares with the tile below, computed twice.  Here
//   * a workgroup (4 waves, three per CU as before) walks a RUN of tiles, every window from its bottom tile up; a tile is FOUR
//     new layer-2 rows (128 positions: two matrix tiles per wave, no wave idles) and the fifth row the depthwise needs - the
//     first row of the tile below - is still in registers from that tile (a run that starts inside a window computes that one
//     row first; a window's bottom tile takes zeros: the padding row);
//   * the rows of the A operand are not in map order: position (y, x) sits at row 32 (2 (x >> 4) + (y & 1)) + 8 (r >> 2)
//     + 4 (y >> 1) + (r & 3) with r = r(x & 15) the accumulator register of the column (below), so that in the accumulator
//     layout of v_mfma_f32_32x32x16_f16 lane (channel c, half h) of wave (g, column tile) holds rows 2 h, 2 h + 1 at columns
//     16 g .. 16 g + 15 of ITS channel: 32 values in two accumulators;
//   * the depthwise of layer 3 (stride 2, SAME = pad 0 before / 1 after) then runs in registers: half h computes output row h,
//     columns 8 g .. 8 g + 7; its third input row is the other half's first row (h = 0) or the row carried from the tile below
//     (h = 1) - ONE v_permlane32_swap per column delivers both - and the one column a wave group lacks (column 16, for g = 0)
//     crosses through 1.5 KB of LDS;
//   * its outputs go straight into the split-f16 A tile of layer 3's 1x1 convolution, which runs as in stem3_kernel.
// No f32 tile in LDS, no kept row in LDS, five (round 6: four) barriers per tile instead of eight, 1.5 x fewer conv1 / 1.25 x fewer layer-2
// rows than stem3_kernel.
// Round 6 (VERDICT r5 next #1b: fewer vector instructions):
//   * a tile's depthwise needs conv1 rows 4 ob - 1 .. 4 ob + 4, of which rows 4 ob + 3, 4 ob + 4 are rows -1, 0 of the tile
//     below: that tile copies them into band rows 4, 5 once its own depthwise has read the band (two 16-byte LDS reads and
//     writes per thread) - four conv1 rows per tile instead of six (the first tile of a run and a window's bottom tile
//     compute all six);
//   * the columns of a wave group sit in the accumulators of layer 2 EVEN COLUMNS FIRST (register r = column 2 r for r < 8,
//     column 2 (r - 8) + 1 behind), so that the stride-2 depthwise of layer 3 finds the operands of two neighbouring outputs
//     (columns 2 j + kw and 2 j + 2 + kw) in one aligned register pair for kw = 0, 1 and runs as v_pk_fma_f32 (36 + 12 pair
//     moves instead of 72 v_fma_f32 per tile and lane); the 1x1 epilogue of layer 2 is packed the same way.  Arithmetic per element is stem3_kernel's, i.e. conv1_kernel / depthwise_kernel /
// pointwise_f16x3_kernel's (taps in row-major order with fmaf, zeros outside the map multiplied, products lo*hi, hi*lo, hi*hi per
// k16 step): bit-identical (tests/test_gpu_parity.py::test_fused_stem_is_bit_identical_to_unfused).
#include "bd_internal.h"

#include <type_traits>

namespace bd {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr float kF16MaxReg = 65504.0f;

__device__ __forceinline__ int rg_swz64(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }
__device__ __forceinline__ float rg_range(float m, v4f v) {
    return fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fmaxf(fabsf(v.y), fabsf(v.z)), fabsf(v.w)));
}
// a = hi + lo, hi = f16(a), lo = f16(a - hi): cnn.hip's split_f16 (one v_fma_mix per low half)
__device__ __forceinline__ void rg_split(v4f a, f16x4& hi, f16x4& lo) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const f16x2 h0 = {(_Float16)a.x, (_Float16)a.y}, h1 = {(_Float16)a.z, (_Float16)a.w};
    f16x2 l0, l1;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(h0), "v"(a.x));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l0) : "v"(h0), "v"(a.y));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(h1), "v"(a.z));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l1) : "v"(h1), "v"(a.w));
    hi[0] = h0[0]; hi[1] = h0[1]; hi[2] = h1[0]; hi[3] = h1[1];
    lo[0] = l0[0]; lo[1] = l0[1]; lo[2] = l1[0]; lo[3] = l1[1];
}

// ---- LDS map (bytes): 53 248, three workgroups per CU ----
constexpr int OFF_C1 = 0;                        // conv1 band [6][34][32] f32
constexpr int OFF_AH = 6 * 34 * 32 * 4;          // 26112: split-f16 A tile of layer 2 [128 rows][64 B]; the log-mel band [13][68] f32 before it
constexpr int OFF_AL = OFF_AH + 128 * 64;        // 34304
constexpr int OFF_A3H = OFF_AL + 128 * 64;       // 42496: split-f16 A tile of layer 3, [2 halves of 32 k][32 rows][64 B]
constexpr int OFF_A3L = OFF_A3H + 2 * 32 * 64;   // 46592
constexpr int OFF_HALO = OFF_C1 + (2 * 34 + 1) * 32 * 4;   // column 16 of the three input rows of a half, [column tile][half][3][32 channels] f32: in
                                                 // conv band row 2 (from its column 0 on, which phase B rewrites - not the zero column -1), dead from
                                                 // the third barrier of a tile to the next tile's phase B
constexpr int OFF_D2W = OFF_A3L + 2 * 32 * 64;   // 50688: taps and shift of depthwise 2, [10][32] f32, for the whole run (as registers
                                                 // requested a phase ahead they were the peak of the register pressure)
constexpr int OFF_C1W = OFF_D2W + 10 * 32 * 4;   // 51968: taps and shift of conv1, [10][32] f32, for the whole run: as global loads at the top
                                                 // of a tile they stood right behind the previous tile's output stores, and a wave can
                                                 // only wait for a load behind stores with vmcnt(0) - the stores' whole round trip
constexpr int kRegLds = OFF_C1W + 10 * 32 * 4;   // 53248

template <bool PLAIN>
__global__ __launch_bounds__(256, 3) void stem_reg_kernel(const float* __restrict__ logmel, int patch_step, const WindowMap map, int w0,
                                                          const float* __restrict__ c1_w, const float* __restrict__ c1_b,
                                                          const float* __restrict__ dw2_w, const float* __restrict__ dw2_b,
                                                          const _Float16* __restrict__ Whi, const _Float16* __restrict__ Wlo,
                                                          const float* __restrict__ pw_u, const float* __restrict__ pw_b,
                                                          const float* __restrict__ dw3_w, const float* __restrict__ dw3_b,
                                                          float* __restrict__ out, const _Float16* __restrict__ W3fhi,
                                                          const _Float16* __restrict__ W3flo, const float* __restrict__ pw3_u,
                                                          const float* __restrict__ pw3_b, unsigned* __restrict__ range_flag, int windows) {
    __shared__ __attribute__((aligned(16))) char smem[kRegLds];
    float (*s_lm)[68] = reinterpret_cast<float (*)[68]>(smem + OFF_AH);
    float (*s_c1)[34][32] = reinterpret_cast<float (*)[34][32]>(smem + OFF_C1);
    char* const s_ah = smem + OFF_AH;
    char* const s_al = smem + OFF_AL;
    float* const s_halo = reinterpret_cast<float*>(smem + OFF_HALO);
    const float* const s_d2 = reinterpret_cast<const float*>(smem + OFF_D2W);
    const float* const s_c1w = reinterpret_cast<const float*>(smem + OFF_C1W);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c4 = tid & 7, col = tid >> 3;               // vector phases: channel quad, map column
    const int g = wave >> 1, wc = wave & 1;               // matrix phases of layer 2: column group, column tile
    const int frow = lane & 31, fh = lane >> 5;
    float rmax = 0.0f;

    // global address space spelled out: a pointer that went through an asm is a flat one otherwise (sepf32.hip)
    typedef const __attribute__((address_space(1))) float* gptr;
    typedef const __attribute__((address_space(1))) v4f* gptr4;
    typedef const __attribute__((address_space(1))) f16x8* gptrh;
    gptr pc1w = (gptr)c1_w, pc1b = (gptr)c1_b, pd2w = (gptr)dw2_w, pd2b = (gptr)dw2_b, pu2 = (gptr)pw_u, pb2 = (gptr)pw_b,
         pd3w = (gptr)dw3_w, pd3b = (gptr)dw3_b, pu3 = (gptr)pw3_u, pb3 = (gptr)pw3_b;
    const __attribute__((address_space(1))) _Float16 *pwh = (const __attribute__((address_space(1))) _Float16*)Whi,
                                                    *pwl = (const __attribute__((address_space(1))) _Float16*)Wlo,
                                                    *pw3h = (const __attribute__((address_space(1))) _Float16*)W3fhi,
                                                    *pw3l = (const __attribute__((address_space(1))) _Float16*)W3flo;

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

    // ---- A: the prefetched log-mel rows 2 (r_first - 1) .. + 12 of a tile into LDS.  The band shares its bytes with the A tile of
    //      layer 2, so it is written when that tile has been read - behind the FOURTH barrier of the tile before - and the fifth
    //      barrier of that tile makes it visible: a tile costs four barriers, not five (round 6)
    auto band_to_lds = [&](int lmr) {
        if (tid < lmr * 16) {
            float4 v = lmv;
            if (!lm_ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&s_lm[tid >> 4][(tid & 15) * 4]) = v;
        } else if (tid >= 256 - lmr) {                     // mel band 64 of every row: the zero to the right of the patch (the band
            float z;                                       // shares its bytes with the A tile).  (A zero made HERE: hoisted out of
            asm volatile("v_mov_b32 %0, 0" : "=v"(z));     // the tile loop, four registers of zeros end up as a spill)
            *reinterpret_cast<float4*>(&s_lm[255 - tid][64]) = make_float4(z, z, z, z);
        }
    };

    // ---- layer-2 rows r_first .. r_first + 3 of a window -> ev[t][k]: this lane's channel (32 wc + frow) at row 2 fh + t, as
    //      PAIRS of columns: k < 4: columns 16 g + 4 k, + 4 k + 2 (the even ones), k >= 4: 16 g + 4 (k - 4) + 1, + 3 (the odd ones)
    //      (phases A - D and the 1x1 convolution's epilogue)
    v2f ev[2][8];
    // c1_new: how many of the C1R conv1 rows are computed here - 4 when the tile below left rows r_first + 3, r_first + 4 in band
    // rows 4, 5, else all
    auto front = [&](auto rows_c, int r_first, int c1_new) {
        constexpr int ROWS = decltype(rows_c)::value;      // 4, or 1: only row r_first (what a run that starts inside a window needs)
        constexpr int C1R = ROWS + 2;
        v4f c1wt[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) c1wt[t] = *reinterpret_cast<const v4f*>(s_c1w + t * 32 + c4 * 4);
        const v4f c1bias = *reinterpret_cast<const v4f*>(s_c1w + 9 * 32 + c4 * 4);
        // (phase A - the tile's log-mel band into LDS - has happened a tile earlier: band_to_lds below)
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
                float (*const c1row)[32] = s_c1[i];
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
                    *reinterpret_cast<v4f*>(&c1row[col + 1][c4 * 4]) = r4;
                } else {
                    *reinterpret_cast<v4f*>(&c1row[col + 1][c4 * 4]) = zero4;
                }
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) lm[0][kw] = lm[2][kw];
            }
        }
        __syncthreads();
        // this lane's layer-2 weight fragments (output channel 32 wc + frow, k = 16 s + 8 fh ..), in flight behind phase C
        f16x8 wbh[2], wbl[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int koff = (wc * 32 + frow) * 32 + (2 * s2 + fh) * 8;
            wbh[s2] = *(gptrh)(pwh + koff);
            wbl[s2] = *(gptrh)(pwl + koff);
        }
        // ---- C: depthwise 2, four rows -> split-f16 A tile; position (y, x) at row 32 (2 (x >> 4) + (y & 1)) + 8 (r >> 2)
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
                rmax = rg_range(rmax, acc);
                f16x4 hi, lo;
                rg_split(acc, hi, lo);
                const int row = rbase + 32 * (r & 1) + 4 * (r >> 1);
                const int off = rg_swz64(row, c4 >> 1) + (c4 & 1) * 8;
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
        if constexpr (ROWS == 4) {
            // conv1 rows r_first - 1, r_first are rows r_first' + 3, r_first' + 4 of the tile above: into band rows 4, 5 (every
            // thread moves the two values it wrote in phase B; the next reader is the tile above's phase C, two barriers on)
            const v4f k0 = *reinterpret_cast<const v4f*>(&s_c1[0][col + 1][c4 * 4]);
            const v4f k1 = *reinterpret_cast<const v4f*>(&s_c1[1][col + 1][c4 * 4]);
            *reinterpret_cast<v4f*>(&s_c1[4][col + 1][c4 * 4]) = k0;
            *reinterpret_cast<v4f*>(&s_c1[5][col + 1][c4 * 4]) = k1;
        }
        // ---- D: [128][32] x [32][64]: wave (g, wc) = row tiles 2 g, 2 g + 1 against column tile wc; lane = output channel ----
        const float u2 = pu2[wc * 32 + frow], b2 = pb2[wc * 32 + frow];
        f32x16 acc2[2];
#pragma unroll
        for (int t = 0; t < (ROWS == 1 ? 1 : 2); ++t) {    // (ROWS == 1: row 0 is accumulator 0 of half 0; the A tile's other rows are stale)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[t][r] = 0.0f;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int off = rg_swz64((2 * g + t) * 32 + frow, 2 * s2 + fh);
                const f16x8 ah = *reinterpret_cast<const f16x8*>(s_ah + off);
                const f16x8 al = *reinterpret_cast<const f16x8*>(s_al + off);
                if constexpr (!PLAIN) {
                    acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wbh[s2], acc2[t], 0, 0, 0);
                    acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wbl[s2], acc2[t], 0, 0, 0);
                }
                acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wbh[s2], acc2[t], 0, 0, 0);
            }
        }
        // epilogue: accumulator (t, r, half fh) is row 32 (2 g + t) + 8 (r >> 2) + 4 fh + (r & 3) = position (2 fh + t, 16 g + column
        // of register r)
#pragma unroll
        for (int t = 0; t < (ROWS == 1 ? 1 : 2); ++t)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const v2f v = __builtin_elementwise_fma(v2f{acc2[t][2 * k], acc2[t][2 * k + 1]}, v2f{u2, u2}, v2f{b2, b2});
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
                band_to_lds(7);
                __syncthreads();
                front(std::integral_constant<int, 1>{}, 4 * ob0 + 4, 3);
#pragma unroll
                for (int k = 0; k < 8; ++k) carry[k] = ev[0][k];
                __syncthreads();                           // (its A tile has been read: the band of the first tile may be written)
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) carry[k] = v2f{0.0f, 0.0f};      // row 48 of a window is the zero padding
            }
            prefetch(win0, 4 * ob0);
            band_to_lds(13);
            __syncthreads();                               // the first tile's band (and the tables above) are in LDS
        }
#pragma unroll 1
        for (int t = t_begin; t < t_end; ++t) {
            asm volatile("" : "+s"(pc1w), "+s"(pc1b), "+s"(pd2w), "+s"(pd2b), "+s"(pu2), "+s"(pb2), "+s"(pwh), "+s"(pwl));
            asm volatile("" : "+s"(pd3w), "+s"(pd3b), "+s"(pu3), "+s"(pb3), "+s"(pw3h), "+s"(pw3l));
            const int win = t / 12, ob = 11 - t % 12;
            {   // the next tile's log-mel band (the last tile of the run: its own again): consumed behind this tile's fourth barrier
                const int tn = t + 1 < t_end ? t + 1 : t;
                prefetch(tn / 12, 4 * (11 - tn % 12));
            }
            if (ob == 11 && t != t_begin) {                // a new window: below its bottom tile lies the zero padding, not the
#pragma unroll                                             // window before (uniform branch, one tile in twelve)
                for (int k = 0; k < 8; ++k) carry[k] = v2f{0.0f, 0.0f};
            }
            // (the tile below - same window, same run - has left conv1 rows 4 ob + 3, 4 ob + 4 in band rows 4, 5)
            front(std::integral_constant<int, 4>{}, 4 * ob, (t == t_begin || ob == 11) ? 6 : 4);

            // ---- the third input row of each half: row 2 (the other half's first row) for half 0, the carried row 4 for half 1:
            //      v_permlane32_swap vdst, src trades lanes 32-63 of vdst against lanes 0-31 of src ----
            v2f x2[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                // (the operands go through an empty asm: with the two elements of a vector as operands this compiler merges
                //  the two swaps into ONE and hands its result to both - /tmp-sized reproducer in DESIGN.md 4.3)
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
            // this lane's layer-3 taps (channel 32 wc + frow) and layer-3 weight fragments (in flight behind the barrier)
            float d3w[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) d3w[k] = pd3w[k * 64 + wc * 32 + frow];
            const float d3b = pd3b[wc * 32 + frow];
            f16x8 w3h[4], w3l[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w3h[q] = *(gptrh)(pw3h + ((size_t)(wave * 4 + q) * 64 + lane) * 8);
            __syncthreads();                               // halo column written; every wave has read the A tile of layer 2
            band_to_lds(13);                               // ... whose bytes take the next tile's log-mel band
            float hal[3] = {0.0f, 0.0f, 0.0f};             // column 16 g + 16: the zero padding for g = 1
            if (g == 0) {
                const float* const hr = s_halo + ((wc * 2 + fh) * 3) * 32 + frow;
                hal[0] = hr[0];
                hal[1] = hr[32];
                hal[2] = hr[64];
            }
            // ---- F: depthwise 3, stride 2, in registers: outputs (row fh, columns 8 g + 2 k, + 2 k + 1) from rows 2 fh + kh, columns
            //         16 g + 4 k + kw and 16 g + 4 k + 2 + kw: for kw = 0 the even pair k, for kw = 1 the odd pair k, for kw = 2 the
            //         pair (even pair k's second column, even pair k + 1's first - or column 16) -> split-f16 A tile of layer 3 ----
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v2f a = v2f{d3b, d3b};
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const v2f* const row = kh == 0 ? ev[0] : kh == 1 ? ev[1] : x2;
                    v2f shifted;                         // (second column of pair k, first of pair k + 1 - or column 16): one v_pk_mov_b32
                    if (k < 3) asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(shifted) : "v"(row[k]), "v"(row[k < 3 ? k + 1 : 0]));
                    else shifted = v2f{row[k].y, hal[kh]};
                    a = __builtin_elementwise_fma(row[k], v2f{d3w[kh * 3], d3w[kh * 3]}, a);
                    a = __builtin_elementwise_fma(row[4 + k], v2f{d3w[kh * 3 + 1], d3w[kh * 3 + 1]}, a);
                    a = __builtin_elementwise_fma(shifted, v2f{d3w[kh * 3 + 2], d3w[kh * 3 + 2]}, a);
                }
                {
                    // both hi halves by one v_cvt_pk_f16_f32, lo = f16(v - hi) per value, the range guard's maximum by one v_max3_f32
                    const float v0 = fmaxf(a.x, 0.0f), v1 = fmaxf(a.y, 0.0f);
                    unsigned pkh, pkl;
                    asm volatile("v_cvt_pk_f16_f32 %0, %3, %4\n\t"
                                 "v_fma_mixlo_f16 %1, %0, -1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                                 "v_fma_mixhi_f16 %1, %0, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                                 "v_max3_f32 %2, %2, %3, %4"
                                 : "=&v"(pkh), "=&v"(pkl), "+v"(rmax) : "v"(v0), "v"(v1));
                    const int off0 = wc * (32 * 64) + rg_swz64(fh * 16 + 8 * g + 2 * k, frow >> 3) + (frow & 7) * 2;
                    const int off1 = wc * (32 * 64) + rg_swz64(fh * 16 + 8 * g + 2 * k + 1, frow >> 3) + (frow & 7) * 2;
                    *reinterpret_cast<unsigned short*>(smem + OFF_A3H + off0) = (unsigned short)pkh;
                    *reinterpret_cast<unsigned short*>(smem + OFF_A3H + off1) = (unsigned short)(pkh >> 16);
                    *reinterpret_cast<unsigned short*>(smem + OFF_A3L + off0) = (unsigned short)pkl;
                    *reinterpret_cast<unsigned short*>(smem + OFF_A3L + off1) = (unsigned short)(pkl >> 16);
                }
            }
            // what the tile above takes over: this tile's first row
#pragma unroll
            for (int k = 0; k < 8; ++k) carry[k] = ev[0][k];
            // (the low halves of the weight fragments only now: with them in flight across phase F the kernel needs more than
            //  the 168 registers three workgroups per CU leave a lane)
            if constexpr (!PLAIN) {
#pragma unroll
                for (int q = 0; q < 4; ++q) w3l[q] = *(gptrh)(pw3l + ((size_t)(wave * 4 + q) * 64 + lane) * 8);
            }
            const int n3 = 32 * wave + frow;
            const float b3 = pb3[n3], u3 = pu3[n3];
            __syncthreads();                               // the A tile of layer 3 is complete
            // ---- G: [32][64] x [64][128], one 32 x 32 tile per wave ----
            f32x16 acc3;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[r] = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int off = (q >> 1) * 32 * 64 + rg_swz64(frow, 2 * (q & 1) + fh);
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
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 4 * fh + (r & 3) + 8 * (r >> 2);
                const v2f t2v = __builtin_elementwise_fma(v2f{acc3[r & ~1], acc3[r | 1]}, v2f{u3, u3}, v2f{b3, b3});
                dst3[(size_t)m * 128 + n3] = fmaxf((r & 1) ? t2v.y : t2v.x, 0.0f);
            }
            // (no barrier: the next tile's phase B writes the conv band, which lies below the A tile of layer 3; its phase C - the
            //  layer-2 A tile - and the halo words are one / three barriers away)
        }
    }
    if (range_flag && !(rmax <= kF16MaxReg)) *range_flag = 1u;
}

}  // namespace

// Layers 1-3 complete with the layer-2 tile handed over in registers: out = [windows][24][16][128], the layer-3 output.
void launch_stem_reg(const float* logmel, int patch_step, const WindowMap& map, int w0, int windows, const float* c1_w,
                     const float* c1_b, const SepLayer& L2, const SepLayer& L3, float* out, hipStream_t stream) {
    if (windows <= 0) return;
    long long grid = 3ll * cu_count();                      // three workgroups per CU, each with a contiguous run of the 12 x windows tiles
    if (grid > 12ll * windows) grid = 12ll * windows;
#define BD_STEM_REG(PLAIN)                                                                                           \
    hipLaunchKernelGGL((stem_reg_kernel<PLAIN>), dim3((unsigned)grid), dim3(256), 0, stream, logmel, patch_step, map, w0, c1_w,    \
                       c1_b, dw_w_of(L2), dw_b_of(L2), static_cast<const _Float16*>(L2.pw_whi),                           \
                       static_cast<const _Float16*>(L2.pw_wlo), L2.pw_u, L2.pw_b, dw_w_of(L3), dw_b_of(L3), out,            \
                       static_cast<const _Float16*>(L3.pw_fhi), static_cast<const _Float16*>(L3.pw_flo), L3.pw_u, L3.pw_b,  \
                       L2.range_flag, windows)
    if (L2.pw_mode == 2) BD_STEM_REG(true);
    else BD_STEM_REG(false);
#undef BD_STEM_REG
}

}  // namespace bd
