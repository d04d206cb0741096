// Layer 4 (128 -> 128 on the 24 x 16 map) + the stride-2 depthwise of layer 5 in the exact-f32 mode with the layer-4 tile handed
// to depthwise 5 in REGISTERS (round 5): l4_f32_kernel's pipeline up to the matrix products - the input band of a 32-channel
// chunk -> depthwise 4 -> f32 A tile -> v_mfma_f32_32x32x2_f32, four chunks, one barrier each - and stemreg.hip's hand-over behind
// them:
//   * a tile is four layer-4 rows x 16 columns = 64 positions; wave w owns output channels 32 w .. + 31 and BOTH row tiles, and
//     position (y, x) sits at A row 32 (y & 1) + 8 (x >> 2) + 4 (y >> 1) + (x & 3), so that lane (channel c, half h) holds rows
//     2 h, 2 h + 1 of its channel, all 16 columns, in its two accumulators;
//   * depthwise 5 (stride 2, SAME = pad 0 before / 1 after) runs in registers: half h computes output row h; its third input row is
//     the other half's first row (h = 0) or the first row of the tile BELOW, carried in registers from that tile (h = 1): one
//     v_permlane32_swap per column delivers both.  A workgroup walks a run of tiles, every window from its bottom tile up; a run
//     that starts inside a window computes the one row it lacks first; a window's bottom tile takes zeros (the padding row);
//   * the f32 tile P, the kept row and three barriers per tile are gone, and with them 20 KB of LDS (53 KB per workgroup).  Three
//     workgroups per CU would fit the LDS but not the registers (182 per lane without spills; at the 168 of three waves per SIMD
//     the kernel spills and was slower): two per CU as before, 136-139 vs 142-143 us per 938 windows (same box).
// Arithmetic per element is l4_f32_kernel's, i.e. depthwise_kernel, pointwise_kernel, depthwise_kernel's: bit-identical
// (tests/test_gpu_parity.py::test_fused_f32_mode_equals_one_kernel_per_op).
#include "bd_internal.h"

#include <type_traits>

namespace bd {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kH = 24, kW = 16, kC = 128, kKC = 32;
constexpr int kBandBytes = 6 * 18 * kKC * 4;              // 13824: six input rows x (16 + 2 halo) columns x 32 channels
constexpr int kABytes = 64 * kKC * 4;                     // 8192: A tile [64 rows][32 k] f32, chunk ^ ((row >> 1) & 7)
constexpr int kOffBand = 0, kOffA = 2 * kBandBytes;       // two of each: chunk k + 1 is prepared while chunk k is multiplied
constexpr int kOffTaps = kOffA + 2 * kABytes;             // 44032: both depthwise layers' taps + shifts, [10][128] f32 each
constexpr int kL4RegLds = kOffTaps + 2 * 10 * kC * 4;     // 54272

__global__ __launch_bounds__(256, 2) void l4_reg_f32_kernel(const float* __restrict__ X, const float* __restrict__ dw4_w,
                                                            const float* __restrict__ dw4_b, const float* __restrict__ W4,
                                                            const float* __restrict__ pw4_b, const float* __restrict__ dw5_w,
                                                            const float* __restrict__ dw5_b, float* __restrict__ out, int windows) {
    __shared__ __attribute__((aligned(16))) char smem[kL4RegLds];
    float* const s_t4 = reinterpret_cast<float*>(smem + kOffTaps);  // rows 0 .. 8 the taps, row 9 the shift
    float* const s_t5 = s_t4 + 10 * kC;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        // 320 float4 per table: every thread one, the first wave a second one; every load first, then the LDS writes
        auto src4 = [&](int i) { return i < 9 * kC / 4 ? reinterpret_cast<const float4*>(dw4_w)[i] : reinterpret_cast<const float4*>(dw4_b)[i - 9 * kC / 4]; };
        auto src5 = [&](int i) { return i < 9 * kC / 4 ? reinterpret_cast<const float4*>(dw5_w)[i] : reinterpret_cast<const float4*>(dw5_b)[i - 9 * kC / 4]; };
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
    // (this lane's layer-4 weights: output channel 32 wave + frow, k = 8 s + 4 fh .. + 3; they stay in L2, fetched per chunk)
    const int c4 = tid & 7, pcol = (tid >> 3) & 15, phalf = tid >> 7;     // depthwise 4: channel quad, map column, row parity
    const int arow0 = 8 * (pcol >> 2) + (pcol & 3);                        // A row of position (y, pcol): arow0 + 32 (y & 1) + 4 (y >> 1)
    // this workgroup's run of tiles; tile t = window t / 6, band 5 - t % 6 (bottom band first)
    const long long total = 6ll * windows;
    const int t_begin = (int)(blockIdx.x * total / gridDim.x), t_end = (int)((blockIdx.x + 1) * total / gridDim.x);

    // the band of a chunk is fetched into registers ahead of its use; loads WITHOUT a branch (clamped address, zeroed when
    // stored to LDS): l4_f32_kernel on why
    float4 band[4];
    unsigned band_ok = 0;
    // (global memory through buffer resources: one 32-bit offset register per access instead of a 64-bit address)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, (unsigned)((size_t)windows * kH * kW * kC * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W4), 0, kC * kC * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(out, 0, (unsigned)((size_t)windows * 12 * 8 * kC * 4), 0x00020000);
    const unsigned w4off = ((32 * wave + frow) * kC + 4 * fh) * 4;
    auto fetch_band = [&](int win, int r_first, int kc, int rows) {     // input rows r_first - 1 .. r_first + rows - 2
        const unsigned xin = (unsigned)win * (kH * kW * kC * 4);
        band_ok = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + 256 * u;
            const int cc = i & 7, pos = i >> 3;
            const int row = pos / 18, col = pos - row * 18;
            const int ih = r_first - 1 + row, iw = col - 1;
            const bool ok = row < rows && ih >= 0 && ih < kH && iw >= 0 && iw < kW;
            band_ok |= ok ? 1u << u : 0u;
            const int ihc = ih < 0 ? 0 : ih >= kH ? kH - 1 : ih, iwc = iw < 0 ? 0 : iw >= kW ? kW - 1 : iw;
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(xrs, xin + (unsigned)(((ihc * kW + iwc) * kC + kc * kKC + cc * 4) * 4), 0, 0);
            band[u] = __builtin_bit_cast(float4, v);
        }
    };
    auto put_band = [&](int kc) {                       // registers -> band buffer kc & 1
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + 256 * u;
            float4 v = band[u];
            if (!((band_ok >> u) & 1)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < 6 * 18 * 8) *reinterpret_cast<float4*>(smem + kOffBand + (kc & 1) * kBandBytes + i * 16) = v;
        }
    };
    v4f w4[4];                                          // the layer's weights of the chunk to come (from L2), requested behind the
#pragma unroll                                          // matrix instructions of the chunk before: in flight during its depthwise
    for (int q = 0; q < 4; ++q) w4[q] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(wrs, w4off, 32 * q, 0));

    // ---- layer-4 rows r_first .. r_first + ROWS - 1 of a window -> ev[t][x]: this lane's channel at row 2 fh + t, column x.
    //      The first chunk's band is in registers on entry; on exit the first band of (next_win, next_r_first, next_rows) is.
    float ev[2][16];
    auto rows_to_regs = [&](auto rows_c, int win, int r_first, int next_win, int next_r_first, int next_rows) {
        constexpr int ROWS = decltype(rows_c)::value;   // 4, or 1: only row r_first (what a run that starts inside a window lacks)
        constexpr int NT = ROWS == 1 ? 1 : 2;
        // depthwise 4 of chunk kc (depthwise_kernel's chain: shift, then the taps in (kh, kw) order, zeros outside the map):
        // band buffer kc & 1 -> A buffer kc & 1
        auto depthwise4 = [&](int kc) {
            const float (*s_x)[18][kKC] = reinterpret_cast<const float (*)[18][kKC]>(smem + kOffBand + (kc & 1) * kBandBytes);
            char* const s_a = smem + kOffA + (kc & 1) * kABytes;
            v4f a[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) a[it] = *reinterpret_cast<const v4f*>(s_t4 + 9 * kC + kc * kKC + c4 * 4);
#pragma unroll 1
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const v4f wt = *reinterpret_cast<const v4f*>(s_t4 + (kh * 3 + kw) * kC + kc * kKC + c4 * 4);
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int r = 2 * it + phalf;   // layer-4 row within the tile (ROWS == 1: only r == 0 is real)
                        a[it] = __builtin_elementwise_fma(*reinterpret_cast<const v4f*>(&s_x[r < ROWS ? r + kh : kh][pcol + kw][c4 * 4]), wt, a[it]);
                    }
                }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int r = 2 * it + phalf;
                if (r < ROWS) {
                    v4f v = a[it];
                    v.x = fmaxf(v.x, 0.0f);
                    v.y = fmaxf(v.y, 0.0f);
                    v.z = fmaxf(v.z, 0.0f);
                    v.w = fmaxf(v.w, 0.0f);
                    const int row = arow0 + 32 * (r & 1) + 4 * (r >> 1);
                    *reinterpret_cast<v4f*>(s_a + row * 128 + ((c4 ^ ((row >> 1) & 7)) << 4)) = v;
                }
            }
        };
        f32x16 acc[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        put_band(0);
        fetch_band(win, r_first, 1, ROWS + 2);
        __syncthreads();                                // band 0 (and, the first time, the taps) are in LDS; the A tiles are free
        depthwise4(0);
#pragma unroll                                   // (unrolled: which band to fetch is then known at compile time)
        for (int kc = 0; kc < kC / kKC; ++kc) {
            if (kc + 1 < kC / kKC) put_band(kc + 1);    // its buffer was last read two chunks ago
            __syncthreads();                            // A tile kc and band kc + 1 are complete; A tile kc - 1 has been read
            if (kc + 2 < kC / kKC) fetch_band(win, r_first, kc + 2, ROWS + 2);
            else if (kc + 2 == kC / kKC) fetch_band(next_win, next_r_first, 0, next_rows);   // the next pass's first band (stays in
                                                        // registers through the last chunk and the depthwise behind it)
            asm volatile("" ::: "memory");              // issued HERE, in front of the matrix instructions, not sunk behind them
            // ---- [32 NT][32] x [32][128]: wave w = output channels 32 w .., activations as the A operand (lane = channel) ----
            const char* const s_a = smem + kOffA + (kc & 1) * kABytes;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int row = i * 32 + frow;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v4f a4 = *reinterpret_cast<const v4f*>(s_a + row * 128 + (((2 * q + fh) ^ ((row >> 1) & 7)) << 4));
                    const v4f ww = w4[q];
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, ww.x, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, ww.y, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, ww.z, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, ww.w, acc[i], 0, 0, 0);
                }
            }
            // this lane's weights of the NEXT chunk (of chunk 0 again behind the last one: the next pass starts with them)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) w4[q] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(wrs, w4off, (((kc + 1) & 3) * kKC + 8 * q) * 4, 0));
            if (kc + 1 < kC / kKC) depthwise4(kc + 1);
        }
        // epilogue: accumulator (t, r, half fh) is A row 32 t + 8 (r >> 2) + 4 fh + (r & 3) = position (2 fh + t, r)
        const float b4 = pw4_b[32 * wave + frow];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) ev[t][r] = fmaxf(acc[t][r] + b4, 0.0f);
    };

    float carry[16];                                    // half 0: row 0 of the tile below (= row 4 of this one), column r
    if (t_begin < t_end) {
        {
            const int win0 = t_begin / 6, ob0 = 5 - t_begin % 6;
            if (ob0 != 5) {                             // the run starts inside a window: row 0 of the tile below, alone
                fetch_band(win0, 4 * ob0 + 4, 0, 3);
                rows_to_regs(std::integral_constant<int, 1>{}, win0, 4 * ob0 + 4, win0, 4 * ob0, 6);
#pragma unroll
                for (int r = 0; r < 16; ++r) carry[r] = ev[0][r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) carry[r] = 0.0f;      // row 24 of a window is depthwise 5's zero padding
                fetch_band(win0, 4 * ob0, 0, 6);
            }
        }
#pragma unroll 1
        for (int t = t_begin; t < t_end; ++t) {
            const int win = t / 6, ob = 5 - t % 6;
            const int tn = t + 1 < t_end ? t + 1 : t;   // (the last tile of the run fetches its own first band again: unused)
            rows_to_regs(std::integral_constant<int, 4>{}, win, 4 * ob, tn / 6, 4 * (5 - tn % 6), 6);

            // ---- depthwise 5, stride 2, in registers: half fh computes output row fh, columns 0 .. 7, from rows 2 fh + kh (the
            //      third one: the other half's first row / the carried row, one v_permlane32_swap per column) and columns 2 ow + kw
            //      (column 16: the zero padding) ----
            float x2[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, ev[0][r]), __builtin_bit_cast(unsigned, carry[r]),
                                                                 false, false);
                x2[r] = __builtin_bit_cast(float, fh ? (unsigned)sw[0] : (unsigned)sw[1]);
                carry[r] = ob == 0 ? 0.0f : ev[0][r];   // what the tile above takes over (the next window's bottom tile: zeros)
            }
            const int ch = 32 * wave + frow;
            float wt[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) wt[k] = s_t5[k * kC + ch];
            const float shift = s_t5[9 * kC + ch];
            // the next pass's first band is waited for HERE, in front of the stores (sepf32.hip: vmcnt counts loads and stores
            // together and the two retire out of order)
#pragma unroll
            for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(band[u].x), "+v"(band[u].y), "+v"(band[u].z), "+v"(band[u].w));
            const unsigned dst = (unsigned)((((win * 12 + 2 * ob + fh) * 8) * kC + ch) * 4);
#pragma unroll
            for (int ow = 0; ow < 8; ++ow) {
                float a = shift;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int ix = 2 * ow + kw;
                        const float v = ix < 16 ? (kh == 0 ? ev[0][ix < 16 ? ix : 0] : kh == 1 ? ev[1][ix < 16 ? ix : 0] : x2[ix < 16 ? ix : 0]) : 0.0f;
                        a = fmaf(v, wt[kh * 3 + kw], a);
                    }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(a, 0.0f)), yrs, dst, ow * kC * 4, 0);
            }
        }
    }
}

}  // namespace

// Exact-f32 mode: layer 4 + the depthwise of layer 5 with the layer-4 tile handed over in registers:
// in = [windows][24][16][128], out = [windows][12][8][128].  False: shape not covered.
bool launch_l4_reg_f32(const float* in, float* out, int windows, const SepLayer& L4, const SepLayer& L5, hipStream_t stream) {
    if (windows <= 0) return true;
    if (L4.cin != 128 || L4.cout != 128 || L4.h_in != 24 || L4.w_in != 16 || L4.stride != 1 || L5.cin != 128 || L5.stride != 2)
        return false;
    // persistent: two workgroups per CU (182 registers per lane: LDS would take three), each with a contiguous run of the tiles
    long long grid = 2ll * cu_count();
    if (grid > 6ll * windows) grid = 6ll * windows;
    hipLaunchKernelGGL(l4_reg_f32_kernel, dim3((unsigned)grid), dim3(256), 0, stream, in, L4.dw_w, L4.dw_b, L4.pw_wt, L4.pw_b, L5.dw_w,
                       L5.dw_b, out, windows);
    return true;
}

}  // namespace bd
