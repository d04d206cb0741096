// Exact-f32 mode (bd_set_pointwise_mode 0), the PREVIOUS form of its two front kernels, kept for same-box A/B behind
// bd_set_fusion stem = 5: stem3_f32_kernel (layers 1-3) and l4_f32_kernel (layer 4 + depthwise 5), both with their tiles through
// LDS.  The mode's default launch set is stemregf32.hip, l4regf32.hip, sepmidf32.hip, sepchipf32.hip + pointwise_kernel with an
// epilogue for layers 13 / 14.  (Round 4's one-kernel-per-separable-layer form, sepf32_kernel behind bd_set_fusion separable = 6,
// was measured slower than that set - 0.72 vs 0.85 M windows/s - and removed in round 6; the code is refused with BD_EINVAL.)
//
// The mode's products are v_mfma_f32_32x32x2_f32: exact f32, 64 cycles per instruction and SIMD, 1/16 of the f16 rate.  Every
// output is the same chain of IEEE operations as conv1_kernel / depthwise_kernel / pointwise_kernel produce: bit-identical
// (tests/test_gpu_parity.py::test_fused_f32_mode_equals_one_kernel_per_op).
#include "bd_internal.h"

#include <cstdlib>
#include <type_traits>

#include <mutex>

namespace bd {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef float v4f __attribute__((ext_vector_type(4)));

// Layers 1-3 in exact f32 as ONE kernel: the structure of stem3_kernel<true> (cnn.hip: a tile is two output rows of layer 3
// in one window; log-mel band -> conv1 band -> depthwise 2 -> 1x1 32 -> 64 -> f32 tile P -> depthwise 3 (stride 2) -> 1x1
// 64 -> 128 -> HBM; 51.7 KB of LDS, three workgroups per CU) with the two 1x1 convolutions on v_mfma_f32_32x32x2_f32 from
// f32 tiles, every element through the chain of IEEE operations of conv1_kernel, depthwise_kernel and pointwise_kernel (k in
// that kernel's order: lane half h takes k = 8 s + 4 h + j for the j-th instruction of super-step s).  Neither the conv1
// output (201 MB per 1024 windows), the layer-2 output (402 MB) nor the two depthwise outputs reach HBM.
// A tile's two depthwise-3 rows need the layer-2 rows 4 ob .. 4 ob + 4, the last of which is the first row of the tile below.
// As in l4_f32_kernel, a workgroup walks a run of tiles, every window from its bottom tile UP, and the tile below leaves that
// row in LDS (the bottom tile of a window: the zero padding): a tile computes FOUR layer-2 rows = 128 positions from six conv1
// rows (round 5; before: five rows from seven, 1.25 x the layer's MFMAs and depthwise work, 1.17 x conv1's).  A run that starts
// inside a window first computes that one row alone (ROWS = 1).
// LDS tiles read as MFMA operands are [row][16-byte chunk] with the chunk index XORed so that the sixteen rows of a
// ds_read_b128 lane group land on sixteen different slots of the 256-byte bank row:
//   A2 [128][32 f32] (128-byte rows: two per bank row)   chunk ^ ((row >> 1) & 7)
//   A3 [ 32][64 f32] (256-byte rows: one per bank row)   chunk ^ (row & 15)
__global__ __launch_bounds__(256, 3) void stem3_f32_kernel(const float* __restrict__ logmel, int patch_step, const WindowMap map,
                                                           int w0, const float* __restrict__ c1_w, const float* __restrict__ c1_b,
                                                           const float* __restrict__ dw2_w, const float* __restrict__ dw2_b,
                                                           const float* __restrict__ W2, const float* __restrict__ pw2_b,
                                                           const float* __restrict__ dw3_w, const float* __restrict__ dw3_b,
                                                           const float* __restrict__ W3, const float* __restrict__ pw3_b,
                                                           float* __restrict__ out, int windows, int run) {
    constexpr int PW = 68;                      // padded row of the f32 output tile
    constexpr int OFF_C1 = 0;                   // conv1 band [6][34][32] f32
    constexpr int OFF_A2 = 6 * 34 * 32 * 4;     // 26112: A2 [128][32] f32; the log-mel band [13][68] sits here before it
    constexpr int P_BYTES = 128 * PW * 4;       // 34816: P [128][68] f32 aliases from 0 after the first product
    constexpr int OFF_A3 = P_BYTES;             // [32][64] f32 (over A2, which has been consumed by then)
    constexpr int OFF_KEEP = OFF_A3 + 32 * 64 * 4;   // 43008: the layer-2 row the tile above needs, [32][68] f32
    constexpr int LDS_BYTES = OFF_KEEP + 32 * PW * 4;   // 51712
    static_assert(OFF_A2 + 128 * 128 <= OFF_KEEP && OFF_A2 + 13 * 68 * 4 <= OFF_A3, "the tiles share the f16 kernel's carve-up");
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    float (*s_lm)[68] = reinterpret_cast<float (*)[68]>(smem + OFF_A2);
    float (*s_c1)[34][32] = reinterpret_cast<float (*)[34][32]>(smem + OFF_C1);
    char* const s_a2 = smem + OFF_A2;
    float* const P = reinterpret_cast<float*>(smem);
    float* const keep = reinterpret_cast<float*>(smem + OFF_KEEP);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int frow = lane & 31, fh = lane >> 5;
    const int c4 = tid & 7;
    const int col = tid >> 3;

    // this workgroup's run of tiles; tile g = window g / 12, band 11 - g % 12 (bottom band first).  run > 0: runs of that many
    // tiles (a divisor of 12), one per workgroup; run == 0: the tiles dealt evenly to the grid
    const long long total = 12ll * windows;
    const int g_begin = run > 0 ? (int)blockIdx.x * run : (int)(blockIdx.x * total / gridDim.x);
    const int g_end = run > 0 ? g_begin + run : (int)((blockIdx.x + 1) * total / gridDim.x);

    // The per-lane constants of a phase are loaded in front of that phase, every tile again (L1 / L2 hits): kept across the
    // tile loop they are 168 registers.  The pointers go through an empty asm per tile so that the loads stay where they are.
    // (global address space spelled out: a pointer that went through an asm is a flat one otherwise, and flat loads count as
    //  LDS operations too)
    typedef const __attribute__((address_space(1))) float* gptr;
    typedef const __attribute__((address_space(1))) v4f* gptr4;
    gptr pc1w = (gptr)c1_w, pc1b = (gptr)c1_b, pd2w = (gptr)dw2_w, pd2b = (gptr)dw2_b, pw2 = (gptr)W2, pb2 = (gptr)pw2_b,
         pd3w = (gptr)dw3_w, pd3b = (gptr)dw3_b, pw3 = (gptr)W3, pb3 = (gptr)pw3_b;

    // What a tile's first phases need from global memory - its log-mel band (one float4 per thread), the conv1 taps, the
    // layer-2 weights - is requested a tile AHEAD, in front of the previous tile's output stores: a load issued behind those
    // stores could only be waited for together with them (vmcnt counts both, in order).  Loads without a branch: clamped
    // address, zeroed when stored to LDS.
    float4 lmv;
    bool lm_ok = false;
    v4f w2[4], c1wt[9], c1bias;
    auto prefetch = [&](int win, int r_first, int lmr) {
        const float* patch = logmel + window_frame(map, w0 + win, patch_step) * BD_MEL_BANDS;
        const int j = tid / 17, q = tid % 17;
        const int ih = 2 * r_first - 2 + j;
        lm_ok = j < lmr && q < 16 && ih >= 0 && ih < BD_PATCH_FRAMES;
        const int ihc = ih < 0 ? 0 : ih >= BD_PATCH_FRAMES ? BD_PATCH_FRAMES - 1 : ih;
        lmv = reinterpret_cast<const float4*>(patch + ihc * BD_MEL_BANDS)[q < 16 ? q : 15];
        // this lane's layer-2 weights (phase D): channel wc * 32 + frow, k = 8 s + 4 fh .. + 3
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) w2[q2] = *(gptr4)(pw2 + (size_t)(wc * 32 + frow) * 32 + 8 * q2 + 4 * fh);
#pragma unroll
        for (int t = 0; t < 9; ++t) c1wt[t] = *(gptr4)(pc1w + t * 32 + c4 * 4);
        c1bias = *(gptr4)(pc1b + c4 * 4);
    };
    auto arrived = [&]() {                      // a use of everything prefetch() requested: the compiler waits for it HERE
        asm volatile("" : "+v"(lmv.x), "+v"(lmv.y), "+v"(lmv.z), "+v"(lmv.w), "+v"(c1bias));
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(w2[q]));
#pragma unroll
        for (int t = 0; t < 9; ++t) asm volatile("" : "+v"(c1wt[t]));
    };
    // layer-2 rows r_first .. r_first + ROWS - 1 of window win (ROWS = 4: -> P; ROWS = 1: -> the keep row): phases A - E
    f32x16 acc2[2];
    auto rows_to_lds = [&](auto rows_c, int r_first) {
        constexpr int ROWS = decltype(rows_c)::value;
        constexpr int C1R = ROWS + 2, LMR = 2 * C1R + 1;
        // ---- A: log-mel rows 2 (r_first - 1) .. + LMR - 1 (prefetched), zero halo columns of the conv1 band ----
        if (tid < LMR * 17) {
            float4 v = lmv;
            if (!lm_ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&s_lm[tid / 17][(tid % 17) * 4]) = v;
        }
        for (int i = tid; i < C1R * 2 * 8; i += 256) {
            const int r = i / 16, side = (i >> 3) & 1, cc = i & 7;
            *reinterpret_cast<float4*>(&s_c1[r][side ? 33 : 0][cc * 4]) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();

        // ---- B: conv1 rows r_first - 1 .. r_first + ROWS (the arithmetic of conv1_kernel: taps in (kh, kw) order, a tap row
        //         past the patch skipped; a conv1 row outside the map is the depthwise's zero padding) ----
        v4f d2wt[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) d2wt[t] = *(gptr4)(pd2w + t * 32 + c4 * 4);
        const v4f d2bias = *(gptr4)(pd2b + c4 * 4);
        {
            float lm[3][3];
            const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) lm[0][kw] = s_lm[0][2 * col + kw];
#pragma unroll
            for (int i = 0; i < C1R; ++i) {
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

        // ---- C: depthwise 2 for the ROWS rows -> f32 A tile [32 ROWS][32] (rolling window over the conv1 band) ----
        {
            v4f cv[3][3];
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) cv[kh][kw] = *reinterpret_cast<const v4f*>(&s_c1[kh][col + kw][c4 * 4]);
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
                const int row = r * 32 + col;
                *reinterpret_cast<v4f*>(s_a2 + row * 128 + ((c4 ^ ((row >> 1) & 7)) << 4)) = acc;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    cv[0][kw] = cv[1][kw];
                    cv[1][kw] = cv[2][kw];
                }
            }
        }
        __syncthreads();

        // ---- D: [32 ROWS][32] x [32][64].  Waves (wr, wc): column tile wc; row tiles wr and wr + 2.  Weights as the A operand,
        //         activations as B: the accumulators hold the transposed tile (lane = position, a register quad = four
        //         consecutive channels), so phase E writes 16 bytes at a time; the same products in the same k order ----
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rt = wr + 2 * i;                                  // row tile 0..3
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][r] = 0.0f;
            if (rt < ROWS) {
                const int row = rt * 32 + frow;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v4f a4 = *reinterpret_cast<const v4f*>(s_a2 + row * 128 + (((2 * q + fh) ^ ((row >> 1) & 7)) << 4));
                    acc2[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[q].x, a4.x, acc2[i], 0, 0, 0);
                    acc2[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[q].y, a4.y, acc2[i], 0, 0, 0);
                    acc2[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[q].z, a4.z, acc2[i], 0, 0, 0);
                    acc2[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[q].w, a4.w, acc2[i], 0, 0, 0);
                }
            }
        }
        if (ROWS == 4) __syncthreads();   // every wave is done with the A tile, the conv band and the log-mel band: P may overwrite them
                                          // (ROWS == 1 writes the keep row, which nobody reads here)
        // ---- E: bias + ReLU -> P / the keep row (pointwise_kernel's epilogue: acc + bias, ReLU) ----
        {
            v4f b4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) b4[g] = *(gptr4)(pb2 + wc * 32 + 8 * g + 4 * fh);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rt = wr + 2 * i;
                if (rt < ROWS) {
                    float* prow = (ROWS == 1 ? keep : P) + (rt * 32 + frow) * PW + wc * 32 + 4 * fh;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        v4f v;
                        v.x = fmaxf(acc2[i][4 * g + 0] + b4[g].x, 0.0f);
                        v.y = fmaxf(acc2[i][4 * g + 1] + b4[g].y, 0.0f);
                        v.z = fmaxf(acc2[i][4 * g + 2] + b4[g].z, 0.0f);
                        v.w = fmaxf(acc2[i][4 * g + 3] + b4[g].w, 0.0f);
                        *reinterpret_cast<v4f*>(prow + 8 * g) = v;
                        if (i == 0) {               // row tile 0 (waves 0, 1) is the row the tile above takes over
                            acc2[0][4 * g + 0] = v.x;
                            acc2[0][4 * g + 1] = v.y;
                            acc2[0][4 * g + 2] = v.z;
                            acc2[0][4 * g + 3] = v.w;
                        }
                    }
                }
            }
        }
        __syncthreads();
    };

    if (g_begin < g_end && g_begin < total) {
        {
            const int win0 = g_begin / 12, ob0 = 11 - g_begin % 12;
            if (ob0 != 11) {
                prefetch(win0, 4 * ob0 + 4, 7);
                rows_to_lds(std::integral_constant<int, 1>{}, 4 * ob0 + 4);           // the run starts inside a window
            } else {                                                                   // row 48 of a window is the zero padding
                for (int i = tid; i < 32 * PW / 4; i += 256) reinterpret_cast<float4*>(keep)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            prefetch(win0, 4 * ob0, 13);
            arrived();                              // (nothing in flight at the head of the tile loop, from either side)
        }
        const int g_stop = g_end < total ? g_end : (int)total;
#pragma unroll 1
        for (int g = g_begin; g < g_stop; ++g) {
            asm volatile("" : "+s"(pc1w), "+s"(pc1b), "+s"(pd2w), "+s"(pd2b), "+s"(pw2), "+s"(pb2));
            asm volatile("" : "+s"(pd3w), "+s"(pd3b), "+s"(pw3), "+s"(pb3));
            const int win = g / 12, ob = 11 - g % 12;
            rows_to_lds(std::integral_constant<int, 4>{}, 4 * ob);

            v4f d3wt[9];                                // depthwise-3 taps (channels 4 (tid & 15) ..)
#pragma unroll
            for (int t = 0; t < 9; ++t) d3wt[t] = *(gptr4)(pd3w + t * 64 + (tid & 15) * 4);
            const v4f d3bias = *(gptr4)(pd3b + (tid & 15) * 4);
            // this lane's layer-3 weights (phase G): channel 32 wave + frow, k = 8 s + 4 fh .. + 3 (in flight during F)
            v4f w3[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) w3[q] = *(gptr4)(pw3 + (size_t)(32 * wave + frow) * 64 + 8 * q + 4 * fh);
            // ---- F: depthwise 3, stride 2: out[o][ow][c] from rows 2o + kh (row 4: the kept one), columns 2ow + kw (column 32 =
            //         padding) -> f32 A3 tile ----
            const int c16 = tid & 15, ow = (tid >> 4) & 15;
            const bool right_edge = ow == 15;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int o = it;
                v4f acc = d3bias;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int pr = 2 * o + kh;
                        const float* const prow = pr == 4 ? keep : P + pr * 32 * PW;
                        v4f v = *reinterpret_cast<const v4f*>(prow + (2 * ow + kw) * PW + c16 * 4);
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
                const int row = o * 16 + ow;
                *reinterpret_cast<v4f*>(smem + OFF_A3 + row * 256 + ((c16 ^ (row & 15)) << 4)) = acc;
            }
            __syncthreads();
            // the row the next tile takes over: this tile's first one (zeros when the next tile is the bottom of a window);
            // nobody reads the keep row before the next tile's phase F
            if (wr == 0) {
                const bool bottom_next = ob == 0;
                float* krow = keep + frow * PW + wc * 32 + 4 * fh;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    v4f v = {acc2[0][4 * g4 + 0], acc2[0][4 * g4 + 1], acc2[0][4 * g4 + 2], acc2[0][4 * g4 + 3]};
                    if (bottom_next) v = v4f{0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<v4f*>(krow + 8 * g4) = v;
                }
            }
            // the next tile's first loads (the last tile of the run: its own again), in flight behind the matrix instructions
            const int n = 32 * wave + frow;
            const float b = pb3[n];
            {
                const int gn = g + 1 < g_stop ? g + 1 : g;
                prefetch(gn / 12, 4 * (11 - gn % 12), 13);
            }
            asm volatile("" ::: "memory");                  // issued HERE, not sunk behind the matrix instructions
            // ---- G: [32][64] x [64][128], one 32 x 32 tile per wave (activations as the A operand) ----
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
            // The loads are waited for HERE, in front of the stores: vmcnt counts loads and stores together and the two kinds
            // retire out of order with respect to each other, so a load with stores in flight in front of its first use can only
            // be waited for with vmcnt(0) - the next tile would open by waiting for this tile's stores to reach memory.
            arrived();
            // ---- H: bias + ReLU, [32][128] block of the layer-3 output (rows are consecutive NHWC positions) ----
            float* dst3 = out + (((size_t)win * 24 + 2 * ob) * 16) * 128;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 4 * fh + (r & 3) + 8 * (r >> 2);
                dst3[(size_t)m * 128 + n] = fmaxf(acc3[r] + b, 0.0f);
            }
            // (no barrier: the next tile's phases A and B write the log-mel and conv bands, which end below A3, and phase C is
            //  two barriers away)
        }
    }
}

// Layer 4 (128 -> 128 on the 24 x 16 map) + the stride-2 depthwise of layer 5 in exact f32 as ONE kernel, in the manner of
// the stem: a tile is two output rows of depthwise 5 in one window; they need the five layer-4 rows 4 ob .. 4 ob + 4, the last
// of which is the first row of the tile below.  Persistent workgroups (two per CU, 74 KB of LDS each) walk a contiguous run of
// tiles, each window from its bottom tile UP, so that row 4 ob + 4 is already there: the tile below left it in LDS (the window's
// bottom tile: the zero padding).  A tile therefore computes FOUR rows = 64 positions = exactly two 32-row matrix tiles, from
// six input rows (round 5; until then every tile computed all five rows as three matrix tiles: 1.5 x the layer's MFMAs).  Only
// the first tile of a run that starts inside a window has nothing to take over and runs the five-row form (NT = 3).
// The 128 input channels go through in four chunks of 32, software-pipelined with ONE barrier per chunk (two band buffers, two
// A buffers):
//   chunk k     v_mfma_f32_32x32x2_f32 on A buffer k & 1, weights as the A operand (lane = position in the accumulators),
//               k ascending; then depthwise 4 of chunk k + 1: band buffer -> f32 A tile [32 NT][32] (16-byte chunks XORed with
//               (row >> 1) & 7); the band [6 (7)][18][32] of chunk k + 2 (zero halo columns, zero rows outside the map) is in
//               flight from HBM into registers meanwhile - for the last two chunks that is the NEXT tile's first band
//   then        bias + ReLU -> f32 tile P[64 (80)][132], depthwise 5 on P and the kept row -> HBM; the tile's first row goes to
//               the keep buffer behind the next barrier (from the registers of the lanes that hold it)
// Both depthwise layers' taps sit in LDS for the life of the workgroup.  The chain of IEEE operations per element is that
// of depthwise_kernel, pointwise_kernel, depthwise_kernel: bit-identical to the three kernels it replaces (316 us per 1024
// windows), whose two intermediate tensors (201 MB each) never exist.
__global__ __launch_bounds__(256, 2) void l4_f32_kernel(const float* __restrict__ X, const float* __restrict__ dw4_w,
                                                        const float* __restrict__ dw4_b, const float* __restrict__ W4,
                                                        const float* __restrict__ pw4_b, const float* __restrict__ dw5_w,
                                                        const float* __restrict__ dw5_b, float* __restrict__ out, int windows) {
    constexpr int H = 24, W = 16, C = 128;
    constexpr int KC = 32, PW = C + 4;
    constexpr int BAND_BYTES = 7 * 18 * KC * 4, A_BYTES = 80 * KC * 4;   // 16128, 10240; two of each (chunk k + 1 is prepared
    constexpr int OFF_BAND = 0, OFF_A = 2 * BAND_BYTES;                  // while chunk k is multiplied; rows 80 .. 95 of the
                                                                         // five-row form's third matrix tile read what follows)
    constexpr int P_BYTES = 80 * PW * 4;                            // 42240, aliases from 0 after the product
    constexpr int LDS_BYTES = OFF_A + 2 * A_BYTES;                  // 52736
    static_assert(P_BYTES <= LDS_BYTES, "P aliases the band and A buffers");
    constexpr int OFF_TAPS = LDS_BYTES;                             // both depthwise layers' taps and shifts: [10][128] f32 each
    constexpr int OFF_KEEP = OFF_TAPS + 2 * 10 * C * 4;             // the layer-4 row the tile above needs: [16][132] f32
    __shared__ __attribute__((aligned(16))) char smem[OFF_KEEP + W * PW * 4 + 2048];   // (+ 2 KB: rows 80 .. 95 of A buffer 1 end
                                                                                       //  past the keep row)
    float* const P = reinterpret_cast<float*>(smem);
    float* const s_t4 = reinterpret_cast<float*>(smem + OFF_TAPS);  // rows 0 .. 8 the taps, row 9 the shift
    float* const s_t5 = s_t4 + 10 * C;
    float* const keep = reinterpret_cast<float*>(smem + OFF_KEEP);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        // 320 float4 per table: every thread one, the first wave a second one; every load first, then the LDS writes
        auto src4 = [&](int i) { return i < 9 * C / 4 ? reinterpret_cast<const float4*>(dw4_w)[i] : reinterpret_cast<const float4*>(dw4_b)[i - 9 * C / 4]; };
        auto src5 = [&](int i) { return i < 9 * C / 4 ? reinterpret_cast<const float4*>(dw5_w)[i] : reinterpret_cast<const float4*>(dw5_b)[i - 9 * C / 4]; };
        const float4 t4a = src4(tid), t5a = src5(tid);
        float4 t4b = t4a, t5b = t5a;
        if (tid < 64) {
            t4b = src4(tid + 256);
            t5b = src5(tid + 256);
        }
        reinterpret_cast<float4*>(s_t4)[tid] = t4a;
        reinterpret_cast<float4*>(s_t5)[tid] = t5a;
        if (tid < 64) {
            reinterpret_cast<float4*>(s_t4)[tid + 256] = t4b;
            reinterpret_cast<float4*>(s_t5)[tid + 256] = t5b;
        }
    }
    const int frow = lane & 31, fh = lane >> 5;
    // this lane's layer-4 weights: output channel 32 wave + frow, k = 8 s + 4 fh .. + 3 (64 KB in all: they stay in L2 and are
    // fetched per chunk, 16 registers instead of 64)
    const float* const w4row = W4 + (size_t)(32 * wave + frow) * C + 4 * fh;
    const int c4 = tid & 7, pcol = (tid >> 3) & 15, phalf = tid >> 7;     // depthwise 4: channel quad, map column, row parity
    // this workgroup's run of tiles; tile g = window g / 6, band 5 - g % 6 (bottom band first)
    const long long total = 6ll * windows;
    const int g_begin = (int)(blockIdx.x * total / gridDim.x), g_end = (int)((blockIdx.x + 1) * total / gridDim.x);
    // the band of a chunk is fetched into registers ahead of its use (in flight behind the depthwise and the product; the
    // first band of the NEXT tile behind this tile's last chunks and epilogue)
    // Loads WITHOUT a branch (clamped address, zeroed when stored to LDS): a load under a branch cannot be counted by the
    // compiler, which then waits for everything in flight - these loads included - before the first use of the weights that
    // were loaded ahead of them, i.e. in front of the matrix instructions the loads are meant to hide behind.
    float4 band[4];
    unsigned band_ok = 0;
    auto fetch_band = [&](int win, int ob, int kc, int rows) {        // rows = 6, or 7 for the five-row form
        const float* const xin = X + (size_t)win * H * W * C;
        band_ok = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + 256 * u;
            const int cc = i & 7, pos = i >> 3;
            const int row = pos / 18, col = pos - row * 18;
            const int ih = 4 * ob - 1 + row, iw = col - 1;
            const bool ok = row < rows && ih >= 0 && ih < H && iw >= 0 && iw < W;
            band_ok |= ok ? 1u << u : 0u;
            const int ihc = ih < 0 ? 0 : ih >= H ? H - 1 : ih, iwc = iw < 0 ? 0 : iw >= W ? W - 1 : iw;
            band[u] = *reinterpret_cast<const float4*>(xin + ((size_t)ihc * W + iwc) * C + kc * KC + cc * 4);
        }
    };
    auto put_band = [&](int kc) {                       // registers -> band buffer kc & 1
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + 256 * u;
            float4 v = band[u];
            if (!((band_ok >> u) & 1)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < 7 * 18 * 8) *reinterpret_cast<float4*>(smem + OFF_BAND + (kc & 1) * BAND_BYTES + i * 16) = v;
        }
    };
    v4f w4[2][4];                                       // the layer's weights of this chunk and the next (from L2)
#pragma unroll
    for (int q = 0; q < 4; ++q) w4[0][q] = *reinterpret_cast<const v4f*>(w4row + 8 * q);

    // one tile in its NT-matrix-tile form: NT = 2 takes row 4 ob + 4 from the keep buffer, NT = 3 computes it
    auto tile_body = [&](auto nt_c, int win, int ob, int nwin, int nob, bool more) {
        constexpr int NT = decltype(nt_c)::value;
        constexpr int R4 = NT + 2;                      // layer-4 rows computed: 4 or 5
        // depthwise 4 of chunk kc (depthwise_kernel's chain: shift, then the taps in (kh, kw) order, zeros outside the map):
        // band buffer kc & 1 -> A buffer kc & 1
        auto depthwise4 = [&](int kc) {
            const float (*s_x)[18][KC] = reinterpret_cast<const float (*)[18][KC]>(smem + OFF_BAND + (kc & 1) * BAND_BYTES);
            char* const s_a = smem + OFF_A + (kc & 1) * A_BYTES;
            v4f a[NT];
#pragma unroll
            for (int it = 0; it < NT; ++it) a[it] = *reinterpret_cast<const v4f*>(s_t4 + 9 * C + kc * KC + c4 * 4);
#pragma unroll 1
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const v4f wt = *reinterpret_cast<const v4f*>(s_t4 + (kh * 3 + kw) * C + kc * KC + c4 * 4);
#pragma unroll
                    for (int it = 0; it < NT; ++it) {
                        const int r = 2 * it + phalf;   // layer-4 row within the tile: 0, 2 (, 4) / 1, 3 (five-row form: row 5 is a dummy)
                        a[it] = __builtin_elementwise_fma(*reinterpret_cast<const v4f*>(&s_x[r < R4 ? r + kh : kh][pcol + kw][c4 * 4]), wt, a[it]);
                    }
                }
#pragma unroll
            for (int it = 0; it < NT; ++it) {
                const int r = 2 * it + phalf;
                if (r < R4) {
                    v4f v = a[it];
                    v.x = fmaxf(v.x, 0.0f);
                    v.y = fmaxf(v.y, 0.0f);
                    v.z = fmaxf(v.z, 0.0f);
                    v.w = fmaxf(v.w, 0.0f);
                    const int row = r * W + pcol;
                    *reinterpret_cast<v4f*>(s_a + row * 128 + ((c4 ^ ((row >> 1) & 7)) << 4)) = v;
                }
            }
        };
        const int r0 = 4 * ob;                          // first layer-4 row of the tile
        f32x16 acc[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        put_band(0);
        fetch_band(win, ob, 1, R4 + 2);
        __syncthreads();                                // band 0 (and, the first time, the taps) are in LDS
        depthwise4(0);
#pragma unroll                                   // (unrolled: which band to fetch is then known at compile time - see fetch_band)
        for (int kc = 0; kc < C / KC; ++kc) {
            // this lane's weights of the NEXT chunk (of chunk 0 again behind the last one: the next tile starts with them)
#pragma unroll
            for (int q = 0; q < 4; ++q) w4[(kc + 1) & 1][q] = *reinterpret_cast<const v4f*>(w4row + ((kc + 1) & 3) * KC + 8 * q);
            if (kc + 1 < C / KC) put_band(kc + 1);      // its buffer was last read two chunks ago
            __syncthreads();                            // A tile kc and band kc + 1 are complete; A tile kc - 1 has been read
            if (kc + 2 < C / KC) fetch_band(win, ob, kc + 2, R4 + 2);
            else if (kc + 2 == C / KC) fetch_band(more ? nwin : win, more ? nob : ob, 0, 6);   // the next tile's first band (or a
                                                        // dummy): stays in registers through the last chunk and the epilogue
            asm volatile("" ::: "memory");              // issued HERE, in front of the matrix instructions, not sunk behind them
            // ---- [32 NT][32] x [32][128]: wave w = output channels 32 w .. ----
            const char* const s_a = smem + OFF_A + (kc & 1) * A_BYTES;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int row = i * 32 + frow;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v4f a4 = *reinterpret_cast<const v4f*>(s_a + row * 128 + (((2 * q + fh) ^ ((row >> 1) & 7)) << 4));
                    const v4f ww = w4[kc & 1][q];
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ww.x, a4.x, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ww.y, a4.y, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ww.z, a4.z, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ww.w, a4.w, acc[i], 0, 0, 0);
                }
            }
            if (kc + 1 < C / KC) depthwise4(kc + 1);
        }
        __syncthreads();   // every wave is done with the band and the A tile: P may overwrite them

        // ---- bias + ReLU -> P (transposed accumulators: lane = position 32 i + frow, registers 4 g .. 4 g + 3 = channels
        //      32 wave + 8 g + 4 fh + (0..3)); layer-4 rows past row 23 are depthwise 5's zero padding ----
        v4f row0[4];                                    // the tile's first row (lanes frow < 16 of matrix tile 0): the tile above needs it
        {
            v4f b4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) b4[g] = *reinterpret_cast<const v4f*>(pw4_b + 32 * wave + 8 * g + 4 * fh);
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int pos = i * 32 + frow;
                if (pos < R4 * W) {
                    const bool live = r0 + pos / W < H;
                    float* prow = P + pos * PW + 32 * wave + 4 * fh;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        v4f v = {0.f, 0.f, 0.f, 0.f};
                        if (live) {
                            v.x = fmaxf(acc[i][4 * g + 0] + b4[g].x, 0.0f);
                            v.y = fmaxf(acc[i][4 * g + 1] + b4[g].y, 0.0f);
                            v.z = fmaxf(acc[i][4 * g + 2] + b4[g].z, 0.0f);
                            v.w = fmaxf(acc[i][4 * g + 3] + b4[g].w, 0.0f);
                        }
                        *reinterpret_cast<v4f*>(prow + 8 * g) = v;
                        if (i == 0) row0[g] = v;
                    }
                }
            }
        }
        __syncthreads();
        // ---- depthwise 5, stride 2 (SAME: pad 0 before, 1 after): out[o][ow][c] from P rows 2 o + kh, columns 2 ow + kw ----
        {
            const int c32 = tid & 31, ow = tid >> 5;    // 8 output columns x 32 channel quads; o = it
            const bool right_edge = ow == 7;            // the tap right of column 15 is the zero padding
            float* dst = out + (((size_t)win * 12 + 2 * ob) * 8) * C;
            v4f a[2];
            a[0] = a[1] = *reinterpret_cast<const v4f*>(s_t5 + 9 * C + c32 * 4);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const v4f wt = *reinterpret_cast<const v4f*>(s_t5 + (kh * 3 + kw) * C + c32 * 4);
                    const int col = 2 * ow + kw;
#pragma unroll
                    for (int o = 0; o < 2; ++o) {
                        const int pr = 2 * o + kh;      // row 4: the kept one (four-row form)
                        const float* const prow = (NT == 2 && pr == 4) ? keep : P + pr * W * PW;
                        v4f v = *reinterpret_cast<const v4f*>(prow + (col < W ? col : W - 1) * PW + c32 * 4);
                        if (kw == 2) {
                            v.x = right_edge ? 0.0f : v.x;
                            v.y = right_edge ? 0.0f : v.y;
                            v.z = right_edge ? 0.0f : v.z;
                            v.w = right_edge ? 0.0f : v.w;
                        }
                        a[o] = __builtin_elementwise_fma(v, wt, a[o]);
                    }
                }
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                v4f v = a[o];
                v.x = fmaxf(v.x, 0.0f);
                v.y = fmaxf(v.y, 0.0f);
                v.z = fmaxf(v.z, 0.0f);
                v.w = fmaxf(v.w, 0.0f);
                *reinterpret_cast<v4f*>(dst + ((size_t)o * 8 + ow) * C + c32 * 4) = v;
            }
        }
        __syncthreads();                                // P and the kept row have been read
        // the row the next tile takes over: this tile's first one - zeros if the next tile is the bottom of a window
        // (nobody reads the keep buffer before the next tile's depthwise 5, four barriers from here)
        if (frow < 16) {
            const bool bottom_next = ob == 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                v4f v = row0[g];
                if (bottom_next) v = v4f{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<v4f*>(keep + frow * PW + 32 * wave + 4 * fh + 8 * g) = v;
            }
        }
    };

    if (g_begin < g_end) {
        const int win0 = g_begin / 6, ob0 = 5 - g_begin % 6;
        const bool cold = ob0 != 5;                     // the run starts inside a window: its first tile computes all five rows
        fetch_band(win0, ob0, 0, cold ? 7 : 6);
        if (!cold && frow < 16) {                       // row 24 of a window is depthwise 5's zero padding
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<v4f*>(keep + frow * PW + 32 * wave + 4 * fh + 8 * g) = v4f{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();                                // the taps are in LDS
#pragma unroll 1
        for (int g = g_begin; g < g_end; ++g) {
            const int win = g / 6, ob = 5 - g % 6;
            const bool more = g + 1 < g_end;
            const int nwin = (g + 1) / 6, nob = 5 - (g + 1) % 6;
            if (g == g_begin && cold) tile_body(std::integral_constant<int, 3>{}, win, ob, nwin, nob, more);
            else tile_body(std::integral_constant<int, 2>{}, win, ob, nwin, nob, more);
        }
    }
}

}  // namespace

// Layers 1-3 in exact f32 as one kernel: out = [windows][24][16][128], the layer-3 output.
void launch_stem_f32(const float* logmel, int patch_step, const WindowMap& map, int w0, int windows, const float* c1_w,
                     const float* c1_b, const SepLayer& L2, const SepLayer& L3, float* out, hipStream_t stream) {
    if (windows <= 0) return;
    // three workgroups per CU, each with a contiguous run of the 12 x windows tiles (BD_STEM_F32_RUN = 2 / 3 / 4 / 6 / 12 in
    // a developer build: one workgroup per run of that many tiles instead)
    int run = 0;
#ifdef BD_KERNEL_TRACE
    if (const char* ev = getenv("BD_STEM_F32_RUN")) run = atoi(ev);
    if (run < 0 || (run > 0 && 12 % run)) run = 0;
#endif
    long long grid = run > 0 ? 12ll * windows / run : 3ll * cu_count();
    if (grid > 12ll * windows) grid = 12ll * windows;
    hipLaunchKernelGGL(stem3_f32_kernel, dim3((unsigned)grid), dim3(256), 0, stream, logmel, patch_step, map, w0, c1_w, c1_b, L2.dw_w,
                       L2.dw_b, L2.pw_wt, L2.pw_b, L3.dw_w, L3.dw_b, L3.pw_wt, L3.pw_b, out, windows, run);
}

// Layer 4 + the depthwise of layer 5 in exact f32 as one kernel: in = [windows][24][16][128], out = [windows][12][8][128].
bool launch_l4_f32(const float* in, float* out, int windows, const SepLayer& L4, const SepLayer& L5, hipStream_t stream) {
    if (windows <= 0) return true;
    if (L4.cin != 128 || L4.cout != 128 || L4.h_in != 24 || L4.w_in != 16 || L4.stride != 1 || L5.cin != 128 || L5.stride != 2)
        return false;
    // persistent: two workgroups per CU (74 KB of LDS each), each with a contiguous run of the 6 x windows tiles
    int grid = 2 * cu_count();
    if (grid > 6 * windows) grid = 6 * windows;
    hipLaunchKernelGGL(l4_f32_kernel, dim3(grid), dim3(256), 0, stream, in, L4.dw_w, L4.dw_b, L4.pw_wt, L4.pw_b, L5.dw_w, L5.dw_b,
                       out, windows);
    return true;
}

}  // namespace bd
