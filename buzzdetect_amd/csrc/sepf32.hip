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

#include <mutex>

namespace bd {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
    const int half = lane >> 5;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + j * 32 + frow;
        const float b = bias[n];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const long long mb = m0 + wr * 32 * TM + i * 32 + 4 * half;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = mb + (r & 3) + 8 * (r >> 2);
                if (m < M) C[(size_t)m * Cout + n] = fmaxf(acc[i][j][r] + b, 0.0f);
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

}  // namespace bd
