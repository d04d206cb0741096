// Layers 1-3 of YAMNet (yamnet.py:79-81: conv 3x3 s2 -> [dw 3x3, pw 32->64] -> [dw 3x3 s2, pw 64->128]) as one kernel that
// WALKS a window instead of cutting it into twelve independent row blocks (round 5; VERDICT r4 next #1b).
//
// stem3_kernel gives every pair of layer-3 rows its own workgroup, which recomputes what its neighbours also compute: 7 conv1
// rows and 5 layer-2 rows per 4 it advances (1.75 x / 1.25 x the arithmetic).  Here a workgroup owns a THIRD of a window (four
// steps of two layer-3 rows) and carries the overlap in LDS from step to step: the last two conv1 rows stay in the first two
// slots of the conv band, the last layer-2 row stays where it is (the five-row f32 tile is walked alternately down and up, so
// the row both steps need never moves).  A warm-up in front of the first step makes the three conv1 rows and the one layer-2
// row a step expects to find.  Per window: 57 conv1 rows and 51 layer-2 rows instead of 84 and 60 (48 are needed).
//
// 512 threads (8 waves), two workgroups per CU (79.5 KB of LDS each): the steps of a walk are serial, so the workgroup is
// twice stem3_kernel's to keep as many waves per CU; every phase has exactly one item per thread or per wave:
//   A  log-mel rows of the step's new conv1 rows -> LDS (9 rows; the band aliases the A tile)
//   B  conv1, 4 new rows: thread = (2 rows, column, 4 channels), rolling 3 x 3 window over the log-mel band
//   C  depthwise 2, 4 new rows -> split-f16 A tile: thread = (2 rows, column, 4 channels), rolling window over the conv band
//   D  1x1 conv 32 -> 64 on the matrix cores, transposed: wave = (row tile, column tile); the other lanes' spare issue slots
//      move the last two conv rows to the front of the band
//   E  bias + ReLU -> the f32 tile (4 of its 5 rows; they overlay the conv band's new rows and the A tile)
//   F  depthwise 3 (stride 2), 2 rows -> split-f16 A tile of layer 3: one output per thread
//   G  1x1 conv 64 -> 128: waves 0-3, one 32 x 32 tile each;  H  bias + ReLU -> global
// Arithmetic per element is stem3_kernel's, i.e. conv1_kernel / depthwise_kernel / pointwise_f16x3_kernel's: bit-identical.
//
// MEASURED AND NOT THE DEFAULT (round 5, same box, three-stream loop, per 938-window launch): stem3_kernel 130.0-132.6 us and
// 1.763-1.771 M windows/s; this kernel 166 us / 1.665 M walking thirds (4 steps), 156 us / 1.705 M halves (6), 155 us / 1.710 M
// whole windows (12, what is compiled in).  A quarter fewer vector instructions, and slower: a walk is 6 barrier intervals per
// step (+ 5 of warm-up) that nothing overlaps inside the workgroup, on two 8-wave workgroups per CU; the block kernel's nine
// intervals per item hide behind two other workgroups of the same CU.  It stays as bd_set_fusion stem = 4 (bit-identity:
// tests/test_gpu_parity.py::test_fused_stem_is_bit_identical_to_unfused); what it would need to win is phases of different
// steps in flight at once (conv1 of step s + 1 beside the matrix work of step s on other waves), not fewer instructions.
#include "bd_internal.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace bd {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr float kF16MaxStem = 65504.0f;

__device__ __forceinline__ int sr_swz64(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }
__device__ __forceinline__ float sr_range(float m, v4f v) {
    return fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fmaxf(fabsf(v.y), fabsf(v.z)), fabsf(v.w)));
}
// a = hi + lo, hi = f16(a), lo = f16(a - hi): cnn.hip's split_f16 (one v_fma_mix per low half)
__device__ __forceinline__ void sr_split(v4f a, f16x4& hi, f16x4& lo) {
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
__device__ __forceinline__ v4f sr_relu(v4f v) {
    v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f);
    return v;
}

// ---- LDS map (bytes) ----
constexpr int kRowC = 34 * 32 * 4;               // a conv1 row: 32 columns + a zero halo column each side, 32 channels: 4352
constexpr int kPW = 68;                          // padded row of the f32 layer-2 tile: 64 channels + 4
constexpr int kRowP = 32 * kPW * 4;              // a layer-2 row of that tile: 8704
constexpr int OFF_C = 0;                         // conv band: slots 0-1 the rows carried over, slots 2-5 the step's new rows
constexpr int OFF_AH = OFF_C + 6 * kRowC;        // 26112: split-f16 A tile of layer 2, 4 rows x 32 positions, hi | lo
constexpr int OFF_AL = OFF_AH + 4 * 32 * 64;     // 34304   (the log-mel band aliases it: 9 rows x 272 B)
constexpr int OFF_PM = OFF_C + 2 * kRowC;        // 8704: rows 1-3 of the f32 tile overlay conv slots 2-5 and the A tile's head
constexpr int OFF_P0 = OFF_AL + 4 * 32 * 64;     // 42496: row 0 and row 4 of the f32 tile survive from step to step
constexpr int OFF_P4 = OFF_P0 + kRowP;           // 51200
constexpr int OFF_A3H = OFF_P4 + kRowP;          // 59904: split-f16 A tile of layer 3, [2 halves of 32 k][32 rows][64 B], hi | lo
constexpr int OFF_A3L = OFF_A3H + 2 * 32 * 64;
constexpr int OFF_T1 = OFF_A3L + 2 * 32 * 64;    // 68096: conv1 taps [9][32] + shift [32]
constexpr int OFF_T2 = OFF_T1 + 10 * 32 * 4;     // depthwise-2 taps [9][32] + shift [32]
constexpr int OFF_T3 = OFF_T2 + 10 * 32 * 4;     // depthwise-3 taps [9][64] + shift [64]
constexpr int OFF_W2 = OFF_T3 + 10 * 64 * 4;     // 73216: layer-2 1x1 weights as MFMA fragments [column tile][k16 step][hi, lo][64 lanes][16 B]
constexpr int kStemLds = OFF_W2 + 2 * 2 * 2 * 1024;   // 81408
static_assert(OFF_PM + 3 * kRowP <= OFF_P0, "the three moving rows of the f32 tile stay inside the conv band + A tile");
static_assert(9 * kPW * 4 <= 2 * 4 * 32 * 64, "log-mel band fits the A tile it aliases");
static_assert(2 * kStemLds <= 160 * 1024, "two workgroups per CU");

__device__ __forceinline__ constexpr int p_slot_off(int slot) {      // byte offset of row `slot` of the f32 tile
    return slot == 0 ? OFF_P0 : slot == 4 ? OFF_P4 : OFF_PM + (slot - 1) * kRowP;
}

template <bool PLAIN, int STEPS>
__global__ __launch_bounds__(512, 4) void stem_roll_kernel(const float* __restrict__ logmel, int patch_step, const WindowMap map, int w0,
                                                           const float* __restrict__ c1_w, const float* __restrict__ c1_b,
                                                           const float* __restrict__ dw2_w, const float* __restrict__ dw2_b,
                                                           const _Float16* __restrict__ Whi, const _Float16* __restrict__ Wlo,
                                                           const float* __restrict__ pw_u, const float* __restrict__ pw_b,
                                                           const float* __restrict__ dw3_w, const float* __restrict__ dw3_b,
                                                           float* __restrict__ out, const _Float16* __restrict__ W3fhi,
                                                           const _Float16* __restrict__ W3flo, const float* __restrict__ pw3_u,
                                                           const float* __restrict__ pw3_b, unsigned* __restrict__ range_flag) {
    static_assert(12 % STEPS == 0, "whole walks per window");
    __shared__ __attribute__((aligned(16))) char smem[kStemLds];
    float (*const s_lm)[kPW] = reinterpret_cast<float (*)[kPW]>(smem + OFF_AH);
    float (*const s_c1)[34][32] = reinterpret_cast<float (*)[34][32]>(smem + OFF_C);
    const float* const t1 = reinterpret_cast<const float*>(smem + OFF_T1);
    const float* const t2 = reinterpret_cast<const float*>(smem + OFF_T2);
    const float* const t3 = reinterpret_cast<const float*>(smem + OFF_T3);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 31, fh = lane >> 5;
    const int c4 = tid & 7, col = (tid >> 3) & 31, half = tid >> 8;
    const int win = blockIdx.y;
    const int g0 = blockIdx.x * STEPS;            // first step of this walk: layer-3 rows 2 g, 2 g + 1
    const float* const patch = logmel + window_frame(map, w0 + win, patch_step) * BD_MEL_BANDS;
    float rmax = 0.0f;

    // ---- once per walk: the three layers' taps into LDS, the zero halo columns of the conv band, this lane's weights ----
    for (int i = tid; i < 10 * 32 + 10 * 32 + 10 * 64; i += 512) {
        float v;
        if (i < 320) v = i < 288 ? c1_w[i] : c1_b[i - 288];
        else if (i < 640) v = i - 320 < 288 ? dw2_w[i - 320] : dw2_b[i - 320 - 288];
        else v = i - 640 < 576 ? dw3_w[i - 640] : dw3_b[i - 640 - 576];
        reinterpret_cast<float*>(smem + OFF_T1)[i] = v;
    }
    for (int i = tid; i < 3 * 2 * 8; i += 512) {          // halo columns of the warm-up's slots 0-2 (a step writes its own)
        const int r = i / 16, side = (i >> 3) & 1, q = i & 7;
        *reinterpret_cast<v4f*>(&s_c1[r][side ? 33 : 0][q * 4]) = v4f{0.f, 0.f, 0.f, 0.f};
    }
    // layer-2 weights -> LDS in fragment order (transposed product: lane (frow, fh) of column tile wc supplies output channel
    // 32 wc + frow, k = 8 (2 s + fh) ..): one 16-byte fragment per thread and half - as registers of the whole walk they spilled
    {
        const int l = tid & 63, s2 = (tid >> 6) & 1, wcw = (tid >> 7) & 1, hl = tid >> 8;
        const int o = (wcw * 32 + (l & 31)) * 32 + (2 * s2 + (l >> 5)) * 8;
        *reinterpret_cast<f16x8*>(smem + OFF_W2 + ((wcw * 2 + s2) * 2 + hl) * 1024 + l * 16) =
            *reinterpret_cast<const f16x8*>((hl ? Wlo : Whi) + o);
    }

    // log-mel rows lm0 .. lm0 + n - 1 of the patch -> band rows 0 .. n - 1 (rows outside the patch and the 4 pad columns: zero)
    auto load_lm = [&](int lm0, int n) {
        for (int i = tid; i < n * 17; i += 512) {
            const int j = i / 17, q = i % 17;
            const int ih = lm0 + j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < 16 && ih >= 0 && ih < BD_PATCH_FRAMES) v = reinterpret_cast<const float4*>(patch + ih * BD_MEL_BANDS)[q];
            *reinterpret_cast<float4*>(&s_lm[j][q * 4]) = v;
        }
    };
    // conv1 rows c .. c + N - 1 (N <= 2) of this thread's column and channels from band rows lmr .. of the log-mel band into conv
    // slots; a row outside the map is the depthwise's zero padding, a tap row behind the patch (log-mel row 96) is skipped
    auto conv_rows = [&](auto n_c, int c, int lmr, int slot) {
        constexpr int N = decltype(n_c)::value;
        v4f wt[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const v4f*>(t1 + t * 32 + c4 * 4);
        const v4f bias = *reinterpret_cast<const v4f*>(t1 + 288 + c4 * 4);
        float lm[3][3];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) lm[0][kw] = s_lm[lmr][2 * col + kw];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int c1r = c + i;
#pragma unroll
            for (int kh = 1; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) lm[kh][kw] = s_lm[lmr + 2 * i + kh][2 * col + kw];
            v4f r4 = {0.f, 0.f, 0.f, 0.f};
            if (c1r >= 0 && c1r < 48) {          // the same for the whole workgroup: a scalar branch
                v4f acc = bias;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    if (2 * c1r + kh >= BD_PATCH_FRAMES) continue;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const float v = lm[kh][kw];
                        acc = __builtin_elementwise_fma(v4f{v, v, v, v}, wt[kh * 3 + kw], acc);
                    }
                }
                r4 = sr_relu(acc);
            }
            *reinterpret_cast<v4f*>(&s_c1[slot + i][col + 1][c4 * 4]) = r4;
            // the zero halo columns are written with every row: slots 2-5 lie under the f32 tile of the previous step
            if (col == 0) *reinterpret_cast<v4f*>(&s_c1[slot + i][0][c4 * 4]) = v4f{0.f, 0.f, 0.f, 0.f};
            if (col == 31) *reinterpret_cast<v4f*>(&s_c1[slot + i][33][c4 * 4]) = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) lm[0][kw] = lm[2][kw];
        }
    };
    // depthwise 2 for N consecutive layer-2 rows: row i reads conv slots s0 + i .. s0 + i + 2 and writes A-tile row a0 + i
    auto dw2_rows = [&](auto n_c, const int (&slots)[4], int a0) {
        constexpr int N = decltype(n_c)::value;
        v4f wt[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const v4f*>(t2 + t * 32 + c4 * 4);
        const v4f bias = *reinterpret_cast<const v4f*>(t2 + 288 + c4 * 4);
        v4f cv[3][3];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) cv[kh][kw] = *reinterpret_cast<const v4f*>(&s_c1[slots[kh]][col + kw][c4 * 4]);
#pragma unroll
        for (int r = 0; r < N; ++r) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) cv[2][kw] = *reinterpret_cast<const v4f*>(&s_c1[slots[r + 2]][col + kw][c4 * 4]);
            v4f acc = bias;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) acc = __builtin_elementwise_fma(cv[kh][kw], wt[kh * 3 + kw], acc);
            acc = sr_relu(acc);
            rmax = sr_range(rmax, acc);
            f16x4 hi, lo;
            sr_split(acc, hi, lo);
            const int off = sr_swz64((a0 + r) * 32 + col, c4 >> 1) + (c4 & 1) * 8;
            *reinterpret_cast<f16x4*>(smem + OFF_AH + off) = hi;
            *reinterpret_cast<f16x4*>(smem + OFF_AL + off) = lo;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                cv[0][kw] = cv[1][kw];
                cv[1][kw] = cv[2][kw];
            }
        }
    };
    // 1x1 conv 32 -> 64 of A-tile row `rt` (32 positions) for this wave's column tile, transposed accumulators, then bias + ReLU
    // into the f32 tile at byte offset p_off (zeros if the layer-2 row lies below the map: depthwise 3's padding)
    auto gemm2 = [&](int rt, f32x16& c) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int off = sr_swz64(rt * 32 + frow, 2 * s2 + fh);
            const f16x8 ah = *reinterpret_cast<const f16x8*>(smem + OFF_AH + off);
            const f16x8 al = *reinterpret_cast<const f16x8*>(smem + OFF_AL + off);
            const char* const wf = smem + OFF_W2 + ((wave & 1) * 2 + s2) * 2048 + lane * 16;
            const f16x8 wbh = *reinterpret_cast<const f16x8*>(wf), wbl = *reinterpret_cast<const f16x8*>(wf + 1024);
            if (s2 == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = 0.0f;
            }
            if constexpr (!PLAIN) {
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wbh, al, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wbl, ah, c, 0, 0, 0);
            }
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wbh, ah, c, 0, 0, 0);
        }
    };
    auto tile_out = [&](const f32x16& c, int p_off, bool live) {
        const int wc = wave & 1;
        float* prow = reinterpret_cast<float*>(smem + p_off) + frow * kPW + wc * 32 + 4 * fh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            v4f v = {0.f, 0.f, 0.f, 0.f};
            if (live) {
                const v4f b4 = *reinterpret_cast<const v4f*>(pw_b + wc * 32 + 8 * g + 4 * fh);
                const v4f u4 = *reinterpret_cast<const v4f*>(pw_u + wc * 32 + 8 * g + 4 * fh);
                v = sr_relu(__builtin_elementwise_fma(v4f{c[4 * g + 0], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]}, u4, b4));
            }
            *reinterpret_cast<v4f*>(prow + 8 * g) = v;
        }
    };

    // ---- warm-up: conv1 rows 4 g0 - 1, 4 g0, 4 g0 + 1 (slots 2, 0, 1) and layer-2 row 4 g0 (row 0 of the f32 tile) ----
    load_lm(8 * g0 - 2, 7);
    __syncthreads();
    if (tid < 256) {                             // rows 4 g0 - 1 (slot 2) and 4 g0 (slot 0): one thread, rolling window
        {
            v4f wt[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const v4f*>(t1 + t * 32 + c4 * 4);
            const v4f bias = *reinterpret_cast<const v4f*>(t1 + 288 + c4 * 4);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c1r = 4 * g0 - 1 + i;
                v4f r4 = {0.f, 0.f, 0.f, 0.f};
                if (c1r >= 0 && c1r < 48) {
                    v4f acc = bias;
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const float v = s_lm[2 * i + kh][2 * col + kw];
                            acc = __builtin_elementwise_fma(v4f{v, v, v, v}, wt[kh * 3 + kw], acc);
                        }
                    r4 = sr_relu(acc);
                }
                *reinterpret_cast<v4f*>(&s_c1[i == 0 ? 2 : 0][col + 1][c4 * 4]) = r4;
            }
        }
    } else {
        conv_rows(std::integral_constant<int, 1>{}, 4 * g0 + 1, 4, 1);
    }
    __syncthreads();
    if (tid < 256) {
        const int slots[4] = {2, 0, 1, 0};
        dw2_rows(std::integral_constant<int, 1>{}, slots, 0);
    }
    __syncthreads();
    {
        f32x16 c;
        if (wave < 2) gemm2(0, c);
        __syncthreads();                         // (the A tile is about to be rewritten by the first step's log-mel band)
        if (wave < 2) tile_out(c, OFF_P0, true);
    }

    float* const dst_win = out + (size_t)win * 24 * 16 * 128;
    // ---- the walk ----
    auto step = [&](auto odd_c, int g) {
        constexpr int ODD = decltype(odd_c)::value;          // even steps walk the f32 tile down (old row = slot 0), odd ones up
        // A: log-mel rows of conv1 rows 4 g + 2 .. 4 g + 5
        load_lm(8 * g + 4, 9);
        __syncthreads();
        // B: conv1, two new rows per thread (slots 2 + 2 half ..)
        conv_rows(std::integral_constant<int, 2>{}, 4 * g + 2 + 2 * half, 4 * half, 2 + 2 * half);
        __syncthreads();
        // C: depthwise 2, layer-2 rows 4 g + 1 + (2 half, 2 half + 1): conv slots 2 half .. 2 half + 3
        {
            const int slots[4] = {2 * half, 2 * half + 1, 2 * half + 2, 2 * half + 3};
            dw2_rows(std::integral_constant<int, 2>{}, slots, 2 * half);
        }
        __syncthreads();
        // D: 1x1 conv 32 -> 64, wave = (row tile wave >> 1, column tile wave & 1); meanwhile the last two conv rows move to the
        // front of the band (slots 4, 5 -> 0, 1: everyone has read slots 0, 1 - the barrier above)
        f32x16 c2;
        gemm2(wave >> 1, c2);
        for (int i = tid; i < 2 * 34 * 8; i += 512)
            reinterpret_cast<v4f*>(smem + OFF_C)[i] = reinterpret_cast<const v4f*>(smem + OFF_C + 4 * kRowC)[i];
        __syncthreads();                         // the A tile and the conv band's new rows may be overwritten
        // E: bias + ReLU -> f32 tile: layer-2 row 4 g + 1 + rt is logical row 1 + rt: slot 1 + rt walking down, 3 - rt walking up
        {
            const int rt = wave >> 1;
            const int slot = ODD ? 3 - rt : 1 + rt;
            const int p_off = slot == 0 ? OFF_P0 : slot == 4 ? OFF_P4 : OFF_PM + (slot - 1) * kRowP;
            tile_out(c2, p_off, 4 * g + 1 + rt < 48);
        }
        __syncthreads();
        // layer-3 weights, fragment order: column tile wave, k16 steps 0..3: requested per step, in flight behind phase F (kept
        // for the whole walk they cost 32 registers of the 128 and the kernel spilled)
        f16x8 w3h[4], w3l[4];
        if (wave < 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const size_t f = ((size_t)(wave * 4 + q) * 64 + lane) * 8;
                w3h[q] = *reinterpret_cast<const f16x8*>(W3fhi + f);
                w3l[q] = *reinterpret_cast<const f16x8*>(W3flo + f);
            }
        }
        // F: depthwise 3, stride 2: one output (row o, column ow, 4 channels) per thread from logical rows 2 o + kh
        {
            const int c16 = tid & 15, ow = (tid >> 4) & 15, o = tid >> 8;
            v4f wt[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const v4f*>(t3 + t * 64 + c16 * 4);
            v4f acc = *reinterpret_cast<const v4f*>(t3 + 576 + c16 * 4);
            const bool right_edge = ow == 15;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                // logical row 2 o + kh -> slot (o is this thread's: both candidates are compile-time, picked by a select)
                const int la = kh, lb = 2 + kh;                                  // o = 0 / o = 1
                const int offa = p_slot_off(ODD ? 4 - la : la), offb = p_slot_off(ODD ? 4 - lb : lb);
                const float* const prow = reinterpret_cast<const float*>(smem + (o ? offb : offa)) + (2 * ow) * kPW + c16 * 4;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    v4f v = *reinterpret_cast<const v4f*>(prow + kw * kPW);
                    if (kw == 2) {               // the tap right of column 31 is the zero padding (what was read there is unused)
                        v.x = right_edge ? 0.0f : v.x;
                        v.y = right_edge ? 0.0f : v.y;
                        v.z = right_edge ? 0.0f : v.z;
                        v.w = right_edge ? 0.0f : v.w;
                    }
                    acc = __builtin_elementwise_fma(v, wt[kh * 3 + kw], acc);
                }
            }
            acc = sr_relu(acc);
            rmax = sr_range(rmax, acc);
            f16x4 hi, lo;
            sr_split(acc, hi, lo);
            const int cc = c16 & 7;
            const int off = (c16 >> 3) * 32 * 64 + sr_swz64(o * 16 + ow, cc >> 1) + (cc & 1) * 8;
            *reinterpret_cast<f16x4*>(smem + OFF_A3H + off) = hi;
            *reinterpret_cast<f16x4*>(smem + OFF_A3L + off) = lo;
        }
        __syncthreads();
        // G + H: [32][64] x [64][128], one 32 x 32 tile per wave 0-3; bias + ReLU -> layer-3 rows 2 g, 2 g + 1
        if (wave < 4) {
            f32x16 acc3;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[r] = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int off = (q >> 1) * 32 * 64 + sr_swz64(frow, 2 * (q & 1) + fh);
                const f16x8 ah = *reinterpret_cast<const f16x8*>(smem + OFF_A3H + off);
                const f16x8 al = *reinterpret_cast<const f16x8*>(smem + OFF_A3L + off);
                if constexpr (!PLAIN) {
                    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, w3h[q], acc3, 0, 0, 0);
                    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w3l[q], acc3, 0, 0, 0);
                }
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w3h[q], acc3, 0, 0, 0);
            }
            float* const dst3 = dst_win + (size_t)(2 * g) * 16 * 128;
            const int n = 32 * wave + frow;
            const float b = pw3_b[n], u = pw3_u[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 4 * fh + (r & 3) + 8 * (r >> 2);
                const v2f t2v = __builtin_elementwise_fma(v2f{acc3[r & ~1], acc3[r | 1]}, v2f{u, u}, v2f{b, b});
                dst3[(size_t)m * 128 + n] = fmaxf((r & 1) ? t2v.y : t2v.x, 0.0f);
            }
        }
        // (no barrier here: the next step's phase A writes the log-mel band = the A tile, last read in D; its phase B the conv
        //  slots 2-5 = rows of the f32 tile that F has read - and F's barrier is behind us)
    };
#pragma unroll 1
    for (int s = 0; s < STEPS; s += 2) {
        step(std::integral_constant<int, 0>{}, g0 + s);
        step(std::integral_constant<int, 1>{}, g0 + s + 1);
    }
    if (range_flag && !(rmax <= kF16MaxStem)) *range_flag = 1u;
}

}  // namespace

// Layers 1-3 complete by walking thirds of a window (stem_roll_kernel): out = [windows][24][16][128], the layer-3 output.
void launch_stem_roll(const float* logmel, int patch_step, const WindowMap& map, int w0, int windows, const float* c1_w,
                      const float* c1_b, const SepLayer& L2, const SepLayer& L3, float* out, hipStream_t stream) {
    if (windows <= 0) return;
#ifndef BD_STEM_STEPS
#define BD_STEM_STEPS 12
#endif
    constexpr int STEPS = BD_STEM_STEPS;      // (developer builds try 2, 6, 12: tools/ab_lib.sh)
#define BD_STEM_ROLL(PLAIN)                                                                                          \
    hipLaunchKernelGGL((stem_roll_kernel<PLAIN, STEPS>), dim3(12 / STEPS, windows), dim3(512), 0, stream, logmel, patch_step, map, \
                       w0, c1_w, c1_b, dw_w_of(L2), dw_b_of(L2), static_cast<const _Float16*>(L2.pw_whi),                 \
                       static_cast<const _Float16*>(L2.pw_wlo), L2.pw_u, L2.pw_b, dw_w_of(L3), dw_b_of(L3), out,            \
                       static_cast<const _Float16*>(L3.pw_fhi), static_cast<const _Float16*>(L3.pw_flo), L3.pw_u, L3.pw_b,  \
                       L2.range_flag)
    if (L2.pw_mode == 2) BD_STEM_ROLL(true);
    else BD_STEM_ROLL(false);
#undef BD_STEM_ROLL
}

}  // namespace bd
