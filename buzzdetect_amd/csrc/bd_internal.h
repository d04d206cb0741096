// Internal declarations shared by the gfx950 kernels and the C-ABI host layer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/buzzdetect_hip.h"

namespace bd {

constexpr int kMelNonZero = 461; // non-zero entries of the [257,64] YAMNet mel matrix (both graph variants)

// Sparsity pattern of tf.signal.linear_to_mel_weight_matrix(64, 257, 16000, 125, 7500) as baked into the
// reference's graphs (features.py:50-55): band m is non-zero on bins
// [kMelStart[m], kMelStart[m] + kMelLen[m]).  Bins 0-4 and 240-256 feed no band.  bd_create checks the
// matrix it is given against this pattern; the front-end kernel is unrolled over it.
constexpr int kMelStart[BD_MEL_BANDS] = {
    5, 5, 6, 7, 9, 10, 11, 12, 13, 14, 16, 17, 18, 20, 21, 23, 25, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 47, 49, 52, 54, 57,
    60, 63, 66, 69, 72, 76, 79, 83, 87, 90, 95, 99, 103, 108, 112, 117, 122, 127, 133, 138, 144, 150, 156, 162, 169, 176, 183,
    190, 198, 206, 214, 223};
constexpr int kMelLen[BD_MEL_BANDS] = {
    1, 2, 3, 3, 2, 2, 2, 2, 3, 3, 2, 3, 3, 3, 4, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 6, 6,
    6, 6, 6, 7, 7, 7, 8, 7, 8, 9, 8, 9, 9, 9, 10, 10, 11, 11, 11, 12, 12, 12, 13, 14, 14, 14, 15, 16, 16, 17, 17, 17};
constexpr int mel_offset(int band) {          // index of band's first weight in FeTables::melw
    int o = 0;
    for (int m = 0; m < band; ++m) o += kMelLen[m];
    return o;
}
static_assert(mel_offset(BD_MEL_BANDS) == kMelNonZero, "mel pattern tables disagree");

// Constant tables of the front end, built on the host at bd_create (engine.hip) and kept in
// device memory.
struct FeTables {
    float  hann[BD_STFT_WINDOW + 16];     // periodic Hann, evaluated in float32 like tf.signal.hann_window; zero past 400
    float2 tw256[256];                    // exp(-2*pi*i*k/256)
    float2 tw512[BD_SPECTRUM_BINS + 1];   // exp(-2*pi*i*k/512), k = 0..256 (+1 pad)
    float  melw[kMelNonZero + 19];        // band-major non-zeros: melw[mel_offset(m) + j] = mel[kMelStart[m] + j][m]; zero
                                          // padding so that every wave may load 48 from its 16-byte boundary
};

// Which log-mel frames a window reads when one launch covers several chunks: the chunks' log-mel rows are
// packed back to back; window w of chunk c (win_start[c] <= w < win_start[c + 1]) is frames
// [frame_base[c] + (w - win_start[c]) * step, + 96).  Passed to the stem kernels by value.
constexpr int kMaxBatchChunks = 64;
struct WindowMap {
    int n_chunks;
    int win_start[kMaxBatchChunks + 1];
    int frame_base[kMaxBatchChunks];
};

__device__ __forceinline__ long long window_frame(const WindowMap& m, int w, int step) {
    int c = 0;
    while (c + 1 < m.n_chunks && w >= m.win_start[c + 1]) ++c;
    return (long long)m.frame_base[c] + (long long)(w - m.win_start[c]) * step;
}

// One separable layer (yamnet.py:52-74) after BatchNorm folding.
struct SepLayer {
    int cin, cout, stride;
    int h_in, w_in, h_out, w_out;
    const float* dw_w;   // [9][cin]   depthwise taps * bn scale
    const float* dw_b;   // [cin]      beta - mean * scale
    const float* pw_wt;  // [cout][cin] pointwise kernel transposed * bn scale (K contiguous)
    const float* pw_b;   // [cout]
    int pw_variant;      // tile choice for the exact-f32 kernel (0 = by shape)
    const void* pw_whi;  // [cout][cin] f16: high half of pw_wt
    const void* pw_wlo;  // [cout][cin] f16: f16(pw_wt - high)
    const void* pw_fhi;  // pw_whi in MFMA B-fragment order: [cout/32][cin/16][64 lanes][8]
    const void* pw_flo;  // pw_wlo, same order
    const float* pw_ffrag;  // pw_wt in the f32 matrix instruction's B-fragment order: [cout/32][cin/8][64 lanes][4] (sepchipf32.hip)
    int pw_variant16;    // tile choice for the split-f16 kernel (0 = by shape)
    int pw_mode;         // 0 = exact f32 MFMA, 1 = split-f16 MFMA (3 products), 2 = plain f16 MFMA (1 product)
    unsigned* range_flag;  // the engine's sticky "an activation left the f16 range" word (modes 1 and 2)
    // Operand scaling of the f16 modes.  Both scales are exact powers of two, so nothing is rounded by them:
    //   activations  the depthwise that produces this layer's GEMM input runs with taps and shift multiplied by 2^act_exp
    //                (dw_w16 / dw_b16): its output is 2^act_exp times the true activation, bit for bit, and lands where f16
    //                hi + lo carry 22 bits (act_exp comes from a calibration pass in exact f32, bd_create / bd_calibrate);
    //   weights      output channel n of pw_whi / pw_wlo / pw_fhi / pw_flo is the split of pw_wt[n][:] * 2^row_exp[n], the
    //                row's largest magnitude in [2^12, 2^13);
    //   epilogue     out = relu(fma(acc, pw_u[n], pw_b[n])),  pw_u[n] = 2^-(act_exp + row_exp[n]).
    const float* dw_w16; // [9][cin]  dw_w * 2^act_exp
    const float* dw_b16; // [cin]     dw_b * 2^act_exp
    const float* pw_u;   // [cout]
    int act_exp;
    unsigned* amax;      // calibration pass only: device word that takes max |depthwise output| (as float bits); else null
};

// compute units of the current device (asked once per device; racing first calls write the same value)
inline int cu_count() {
    static int cus_of[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    int& c = cus_of[dev & 63];
    if (c == 0) {
        int v = 0;
        c = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return c;
}

// taps / shift of the depthwise that feeds L's pointwise convolution, for the arithmetic mode L runs in
inline const float* dw_w_of(const SepLayer& L) { return L.pw_mode != 0 ? L.dw_w16 : L.dw_w; }
inline const float* dw_b_of(const SepLayer& L) { return L.pw_mode != 0 ? L.dw_b16 : L.dw_b; }

// The soxr_hq-class resampler as a matrix-core product (resample.hip): geometry of one rate ratio.  P = 32 NB outputs and D
// input samples make one period (P down = D up, D a multiple of 8); output (m, b, n) = sum_e x[m D + boff[b] + e] G_b[e][n].
struct FirPlan {
    int up, down, P, D, NB;
    int kq, mt;                // k-steps of 16 per wave (four waves split K), 32-row tiles per workgroup
    int contiguous;            // one phase block: the rows overlap and the span is staged once
    int RS;                    // LDS row stride of the Toeplitz view in 16-byte chunks (odd)
    unsigned skew_magic;       // contiguous staging skips one chunk every D / 8: floor(q / (D / 8)) = umulhi(q, skew_magic); 0 = none
    int a_bytes, lds_bytes;    // one f16 half of the staged signal / the workgroup's dynamic LDS
    float unscale[2];          // 2^-(filter scale + input scale) for 16-bit / float PCM
    const void* gfrag;         // device [NB][4 kq][hi, lo][64 lanes][8] f16: B fragments of v_mfma_f32_32x32x16_f16
    const int* koff;           // device [4 kq]: LDS byte offset of a k-step within a row
    const int* boff;           // device [NB]: first input sample of a phase block relative to its period (multiple of 8)
};
struct FirPlanHost {
    FirPlan plan;
    std::vector<uint16_t> gfrag;
    std::vector<int> koff, boff;
};
bool fir_plan_build(int up, int down, const double* taps, int half, FirPlanHost* out);
void launch_fir_mfma(const void* in, bool s16, int64_t n_in, int channels, const FirPlan& plan, float* out, int64_t n_out,
                     hipStream_t stream);

// ---- launchers (each enqueues exactly one kernel on `stream`) ----
void launch_logmel(const float* pcm, int64_t n_valid, int64_t n_frames, float* logmel,
                   const FeTables* tables, hipStream_t stream);
void launch_resample(const void* in, bool s16, int64_t n_in, int channels, const float* taps, int half, int up,
                     int down, float* out, int64_t n_out, hipStream_t stream);
bool resample_span_fits(int half, int up, int down);       // frontend.hip: can resample_kernel stage one output's span?
void launch_patches(const float* logmel, int64_t n_windows, int patch_step, float* patches,
                    hipStream_t stream);
void launch_conv1(const float* logmel, int patch_step, const WindowMap& map, int w0, int windows, const float* w9x32,
                  const float* b32, float* out, hipStream_t stream);
void launch_depthwise(const float* in, float* out, int windows, const SepLayer& L, hipStream_t stream);
void launch_pointwise(const float* in, float* out, int64_t rows, const SepLayer& L, hipStream_t stream);
int launch_pointwise_variant(const float* A, const float* Wt, const float* bias, float* C, long long M, int N,
                             int K, int variant, hipStream_t stream);
int launch_pointwise_f16x3_variant(const float* A, const void* Whi, const void* Wlo, const float* unscale, const float* bias,
                                   float* C, long long M, int N, int K, int variant, hipStream_t stream, bool plain = false,
                                   unsigned* range_flag = nullptr);
void launch_scale_copy(const float* src, float* dst, int64_t n, float factor, hipStream_t stream);
bool launch_pointwise_next_dw_f32(const float* in, float* out, int windows, const SepLayer& L, const SepLayer& Ln,
                                  hipStream_t stream);
bool launch_l4_reg_f32(const float* in, float* out, int windows, const SepLayer& L4, const SepLayer& L5, hipStream_t stream);
bool launch_separable_chip(const float* in, float* out, int windows, const SepLayer* L, int nl, hipStream_t stream,
                           const SepLayer* next = nullptr, bool planes = false);
bool tail_supported(const SepLayer& L13, const SepLayer& L14);      // septail.hip: would the two launches below run?
bool launch_tail_pw13_dw14(const void* in, void* out, int windows, const SepLayer& L13, const SepLayer& L14, hipStream_t stream);   // septail.hip
bool launch_tail_pw14_pool(const void* in, float* pooled, int windows, const SepLayer& L14, hipStream_t stream);
bool tail_f32_supported(const SepLayer& L13, const SepLayer& L14);
// which = 0: pointwise 13 + depthwise 14 (in -> mid); 1: pointwise 14 + pool (mid -> pooled)
bool launch_tail_f32(const float* in, float* mid, float* pooled, int windows, const SepLayer& L13, const SepLayer& L14, hipStream_t stream,
                     int which);
bool launch_separable_chip_f32(const float* in, float* out, int windows, const SepLayer* L, int nl, hipStream_t stream,
                               const SepLayer* next, bool dw0_done);
bool launch_separable_mid_f32(const float* in, float* out, int windows, const SepLayer& L5, const SepLayer& L6, const SepLayer& L7,
                              hipStream_t stream);   // sepchip.hip
bool launch_separable_mid(const float* in, float* out, int windows, const SepLayer& L5, const SepLayer& L6, const SepLayer& L7,
                          hipStream_t stream);       // sepmid.hip
int launch_separable_run_next_dw(const float* a, float* b, int windows, const SepLayer* L, int max_layers, hipStream_t stream,
                                 bool planes = false);
bool launch_separable_fused_next_dw(const float* in, float* out, int windows, const SepLayer& L, const SepLayer& next,
                                    hipStream_t stream);
void launch_stem_reg(const float* logmel, int patch_step, const WindowMap& map, int w0, int windows, const float* c1_w,
                     const float* c1_b, const SepLayer& L2, const SepLayer& L3, float* out, hipStream_t stream);
void launch_stem_reg_f32(const float* logmel, int patch_step, const WindowMap& map, int w0, int windows, const float* c1_w,
                         const float* c1_b, const SepLayer& L2, const SepLayer& L3, float* out, hipStream_t stream);   // stemregf32.hip
void launch_stem4(const float* logmel, int patch_step, const WindowMap& map, int w0, int windows, const float* c1_w,
                  const float* c1_b,
                  const SepLayer& L2, const SepLayer& L3, float* out, hipStream_t stream);
void launch_pool_head(const float* act, int windows, const float* head_wt, const float* head_b,
                      int n_classes, float* emb, float* logits, hipStream_t stream);
void launch_head(const float* pooled, int windows, const float* head_wt, const float* head_b, int n_classes,
                 float* logits, hipStream_t stream);

}  // namespace bd
