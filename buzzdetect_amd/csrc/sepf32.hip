// (Round 5: the mode's default launch set is stemregf32.hip, l4regf32.hip, sepmidf32.hip, sepchipf32.hip + pointwise_kernel with
//  an epilogue for layers 13 / 14; this file's stem3_f32_kernel / l4_f32_kernel run behind bd_set_fusion stem = 5, sepf32_kernel
//  behind separable = 6.)
// Exact-f32 mode (bd_set_pointwise_mode 0), fused: one kernel per separable layer (yamnet.py:52-74: depthwise 3x3 + BN +
// ReLU, 1x1 convolution + BN + ReLU) instead of depthwise_kernel + pointwise_kernel with the depthwise output through HBM.
//
// The mode's products are v_mfma_f32_32x32x2_f32: exact f32, 64 cycles per instruction and SIMD, 1/16 of the f16 rate.  A
// layer of this network is then bound by its matrix instructions (the 512 -> 512 layers: 12.9 GFLOP = 82 us at the 157.3
// TFLOP/s peak) with an order of magnitude of slack everywhere else, so the fused form needs none of the machinery of the
// split-f16 kernels (wave roles, LDS-DMA rings, counted waits): a workgroup
//   1. tabulates its BM output positions (window, row, column -> input offset and which of the nine taps exist: TF SAME,
//      stride 1 pads 1 before, stride 2 pads 0 before / 1 after),
//   2. computes the depthwise + shift + ReLU of those positions for ALL input channels straight from global memory (the
//      nine taps of neighbouring positions overlap: L1 / L2 serve them) into an f32 tile A[BM][Cin] in LDS - in the tap
//      order and with the zero-padding FMAs of depthwise_kernel,
//   3. multiplies the tile by the layer's [Cout][Cin] kernel: eight waves as WGM x WGN, a wave owns 32 TM rows x 64
//      columns, A fragments by ds_read_b128 (row stride Cin + 4 floats: an odd number of 16-byte slots, conflict-free), B
//      fragments by 16-byte global loads from the [Cout][Cin] kernel one super-step ahead, k taken in pointwise_kernel's
//      order (lane half h takes k = 8 s + 4 h + j for the j-th instruction of super-step s),
//   4. adds the shift, applies ReLU and stores NHWC rows.
// Every output is the same chain of IEEE operations as depthwise_kernel + pointwise_kernel produce: bit-identical
// (tests/test_gpu_parity.py::test_fused_f32_mode_equals_one_kernel_per_op).  With all of Cout in one workgroup (two
// column halves for the 1024-channel layers) the depthwise is computed once per position.
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

template <int TM, int TN, int WGM, int WGN, bool MULTI>   // MULTI: Cin > KC, the tile is produced and multiplied a chunk of K at a time
__global__ __launch_bounds__(512, 2) void sepf32_kernel(const float* __restrict__ X, const float* __restrict__ dw_w,
                                                        const float* __restrict__ dw_b, const float* __restrict__ Wt,
                                                        const float* __restrict__ bias, float* __restrict__ C, long long M, int Cin,
                                                        int Cout, int H, int W, int OH, int OW, int stride, int KC) {
    static_assert(WGM * WGN == 8, "eight waves");
    constexpr int BM = 32 * TM * WGM;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int2* const rowtab = reinterpret_cast<int2*>(smem);                 // [BM]: (input offset of tap (0, 0) in floats, tap mask)
    float* const As = reinterpret_cast<float*>(smem + BM * sizeof(int2));   // [BM][KC + 4]: one chunk of KC input channels
    const int lda = KC + 4;
    const int tid = threadIdx.x;
    const long long m0 = (long long)blockIdx.x * BM;

    // ---- 1. the tile's output positions ----
    const int pad = stride == 1 ? 1 : 0;
    for (int r = tid; r < BM; r += 512) {
        const long long m = m0 + r;
        int2 e = make_int2(0, 0);
        if (m < M) {
            const int P = OH * OW;
            const int win = (int)(m / P), p = (int)(m - (long long)win * P);
            const int oh = p / OW, ow = p - oh * OW;
            const int ih0 = oh * stride - pad, iw0 = ow * stride - pad;
            int mask = 0;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
                    if (ih0 + kh >= 0 && ih0 + kh < H && iw0 + kw >= 0 && iw0 + kw < W) mask |= 1 << (kh * 3 + kw);
            e = make_int2(((win * H + ih0) * W + iw0) * Cin, mask);
        }
        rowtab[r] = e;
    }
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WGN, wc = wave % WGN;
    const int frow = lane & 31, fk = (lane >> 5) * 4;
    const int n0 = (int)blockIdx.y * (32 * TN * WGN) + wc * (32 * TN);
    const float* const a_frag = As + (size_t)(wr * 32 * TM + frow) * lda + fk;
    f32x16 acc[TM][TN];
    if constexpr (MULTI) {                         // (one chunk: the accumulators start their lives after the depthwise phase)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    }
    const int c4n = KC >> 2;                       // a power of two <= 64 .. 256: a thread keeps its channel quad for the chunk
    const int c4 = tid & (c4n - 1);
    const int rstep = 512 / c4n;                   // rows covered per pass of the workgroup

    // K in chunks of KC input channels (the 512- and 1024-channel layers: 256 at a time, so that 96 rows fit the LDS and 1024
    // windows are exactly one workgroup per CU); within and across chunks every accumulator takes its k in ascending order
    for (int k0 = 0; k0 < (MULTI ? Cin : 1); k0 += KC) {
        // ---- 2. depthwise + shift + ReLU of the tile, channels k0 .. k0 + KC, into LDS ----
        float4 wt[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[t] = reinterpret_cast<const float4*>(dw_w + (size_t)t * Cin + k0)[c4];
        const float4 shift = reinterpret_cast<const float4*>(dw_b + k0)[c4];
        __syncthreads();                           // the table is written / the previous chunk's tile has been multiplied
        // rows whose 9 x 16-byte loads a thread has in flight together: what the register file leaves beside the accumulators
        constexpr int RB = !MULTI ? 4 : (TM * TN <= 3 ? 3 : 2);
#pragma unroll 1
        for (int r0 = tid / c4n; r0 < BM; r0 += RB * rstep) {
            float4 v[RB][9];
#pragma unroll
            for (int g = 0; g < RB; ++g) {
                const int r = r0 + g * rstep;
                const int2 e = rowtab[r < BM ? r : r0];
                const float* src = X + e.x + k0 + c4 * 4;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int t = kh * 3 + kw;
                        v[g][t] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (e.y >> t & 1) v[g][t] = *reinterpret_cast<const float4*>(src + (kh * W + kw) * Cin);
                    }
            }
#pragma unroll
            for (int g = 0; g < RB; ++g) {
                const int r = r0 + g * rstep;
                float4 a = shift;
#pragma unroll
                for (int t = 0; t < 9; ++t) {      // (kh, kw) order, zeros outside the map: depthwise_kernel's chain
                    a.x = fmaf(v[g][t].x, wt[t].x, a.x);
                    a.y = fmaf(v[g][t].y, wt[t].y, a.y);
                    a.z = fmaf(v[g][t].z, wt[t].z, a.z);
                    a.w = fmaf(v[g][t].w, wt[t].w, a.w);
                }
                a.x = fmaxf(a.x, 0.0f);
                a.y = fmaxf(a.y, 0.0f);
                a.z = fmaxf(a.z, 0.0f);
                a.w = fmaxf(a.w, 0.0f);
                if (r < BM) *reinterpret_cast<float4*>(As + (size_t)r * lda + c4 * 4) = a;
            }
        }
        __syncthreads();
        if constexpr (!MULTI) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        }

        // ---- 3. the 1x1 convolution over this chunk on v_mfma_f32_32x32x2_f32 ----
        // B fragments: lane (column n0 + 32 j + frow, half h) needs k = 8 s + 4 h .. + 3 of its kernel row for super-step s.
        // The four super-steps of a 32-wide k block are one 128-byte line of that row (h = 0 and h = 1 take alternate
        // 16-byte pieces): they are requested together - every line is fetched once and used whole - and a block ahead,
        // AFTER the block's first super-step: the wait in front of a block then meets loads that are three super-steps old.
        const float* const b_frag = Wt + (size_t)(n0 + frow) * Cin + k0 + fk;
        const int nq = KC >> 5;                    // blocks of 32 k = 4 super-steps
        float4 bv[4][TN], bn[4][TN];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[u][j] = *reinterpret_cast<const float4*>(b_frag + (size_t)j * 32 * Cin + u * 8);
#pragma unroll 1
        for (int q = 0; q < nq; ++q) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float4 av[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const float4*>(a_frag + (size_t)i * 32 * lda + q * 32 + u * 8);
                // instruction c of every accumulator before instruction c + 1 of any (each accumulator still takes its k in
                // pointwise_kernel's order - lane half h: k = 8 s + 4 h + c - bit-identical)
#define BD_SEPF32_STEP(CMP)                                                                                        \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].CMP, bv[u][j].CMP, acc[i][j], 0, 0, 0);
                BD_SEPF32_STEP(x)
                BD_SEPF32_STEP(y)
                BD_SEPF32_STEP(z)
                BD_SEPF32_STEP(w)
#undef BD_SEPF32_STEP
                if (u == 0 && q + 1 < nq) {
#pragma unroll
                    for (int uu = 0; uu < 4; ++uu)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            bn[uu][j] = *reinterpret_cast<const float4*>(b_frag + (size_t)j * 32 * Cin + (q + 1) * 32 + uu * 8);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[u][j] = bn[u][j];
        }
    }

    // ---- 4. shift + ReLU, NHWC rows (C/D map of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) ----
    // A tile inside M stores without a branch per row: under a branch each store gets its own s_waitcnt vmcnt(0) (for the
    // bias load), which on gfx9 also waits for the store before it - sixteen memory round trips in a row per 32 x 32 tile.
    const int half = lane >> 5;
    const bool whole = m0 + BM <= M;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + j * 32 + frow;
        const float b = bias[n];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const long long mb = m0 + wr * 32 * TM + i * 32 + 4 * half;
            if (whole) {
                float* const crow = C + (size_t)mb * Cout + n;
#pragma unroll
                for (int r = 0; r < 16; ++r) crow[(size_t)((r & 3) + 8 * (r >> 2)) * Cout] = fmaxf(acc[i][j][r] + b, 0.0f);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long m = mb + (r & 3) + 8 * (r >> 2);
                    if (m < M) C[(size_t)m * Cout + n] = fmaxf(acc[i][j][r] + b, 0.0f);
                }
            }
        }
    }
}

template <int TM, int TN, int WGM, int WGN, bool MULTI>
void launch_one(const float* in, float* out, int windows, const SepLayer& L, int kc, hipStream_t stream) {
    constexpr int BM = 32 * TM * WGM;
    const long long M = (long long)windows * L.h_out * L.w_out;
    const size_t lds = (size_t)BM * sizeof(int2) + (size_t)BM * (kc + 4) * sizeof(float);
    static std::once_flag once[16];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & 15], [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sepf32_kernel<TM, TN, WGM, WGN, MULTI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)(L.cout / (32 * TN * WGN)));
    hipLaunchKernelGGL((sepf32_kernel<TM, TN, WGM, WGN, MULTI>), grid, dim3(512), lds, stream, in, L.dw_w, L.dw_b, L.pw_wt, L.pw_b, out,
                       M, L.cin, L.cout, L.h_in, L.w_in, L.h_out, L.w_out, L.stride, kc);
}

// Tile shape per layer, chosen so that a 1024-window pass is a whole number of rounds of the 256 CUs (in brackets: rows per
// workgroup x columns, workgroups, LDS):
//   32 -> 64 (layer 2)          2 x 2 tiles per wave, 8 x 1 waves   [512 x 64,   3072, 78 KB: two per CU, six rounds]
//   64 -> 128 (3), 128 -> 128 (4)          4 x 2 waves              [256 x 128,  1536, 72 / 137 KB: three / six rounds]
//   128 -> 256 (5), 256 -> 256 (6)         2 x 4 waves              [128 x 256,   768, 68 / 134 KB]
//   256 -> 512 (7), 512 -> 512 (8 - 12)    3 x 2 tiles, 1 x 8 waves [ 96 x 512,   256, 101 KB, K in chunks of 256: one round]
//   512 -> 1024 (13), 1024 -> 1024 (14)    3 x 1 tiles, 1 x 8 waves [ 96 x 256,   256 (64 row tiles x 4 column quarters)]
static int sep_f32_config(const SepLayer& L, int windows) {     // 0: not a shape of this kernel; else a config number
    if (L.cin < 32 || L.cin > 1024 || (L.cin & (L.cin - 1)) || L.cout % 64) return 0;
    if ((long long)windows * L.h_in * L.w_in * L.cin >= (1LL << 31)) return 0;          // offsets are 32-bit
    if (L.cout == 64 && L.cin <= 64) return 1;
    if (L.cout == 128 && L.cin <= 128) return 2;
    if (L.cout == 256 && L.cin <= 256) return 3;
    if (L.cout == 512) return 4;
    if (L.cout == 1024) return 5;
    return 0;
}

}  // namespace

bool sep_f32_ok(const SepLayer& L, int windows) { return sep_f32_config(L, windows) != 0; }

// Layer L in exact f32 as one kernel: in = [windows][h_in][w_in][cin], out = [windows][h_out][w_out][cout].  False when the
// shape is not one of the network's (the caller then runs depthwise_kernel + pointwise_kernel).
bool launch_sep_f32(const float* in, float* out, int windows, const SepLayer& L, hipStream_t stream) {
    if (windows <= 0) return true;
    const int kc = L.cin < 256 ? L.cin : 256;
    const bool multi = L.cin > kc;
    switch (sep_f32_config(L, windows)) {
        case 1: launch_one<2, 2, 8, 1, false>(in, out, windows, L, kc, stream); return true;
        case 2: launch_one<2, 2, 4, 2, false>(in, out, windows, L, kc, stream); return true;
        case 3: launch_one<2, 2, 2, 4, false>(in, out, windows, L, kc, stream); return true;
        case 4:
            if (multi) launch_one<3, 2, 1, 8, true>(in, out, windows, L, kc, stream);
            else launch_one<3, 2, 1, 8, false>(in, out, windows, L, kc, stream);
            return true;
        case 5: launch_one<3, 1, 1, 8, true>(in, out, windows, L, kc, stream); return true;
        default: return false;
    }
}

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
