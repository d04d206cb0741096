// Internal declarations shared by the gfx950 kernels and the C-ABI host layer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/buzzdetect_hip.h"

namespace bd {

constexpr int kMelMaxLen = 18;   // longest run of non-zero bins a mel band may have (kept in registers; YAMNet: 17)

// Constant tables of the front end, built on the host at bd_create (engine.hip) and kept in
// device memory; every workgroup stages them into LDS once.
struct FeTables {
    float  hann[BD_STFT_WINDOW];          // periodic Hann, evaluated in float32 like tf.signal.hann_window
    float2 tw256[256];                    // exp(-2*pi*i*k/256)
    float2 tw512[BD_SPECTRUM_BINS + 1];   // exp(-2*pi*i*k/512), k = 0..256 (+1 pad)
    int    band_start[BD_MEL_BANDS];      // first non-zero bin of each mel band
    int    band_len[BD_MEL_BANDS];        // number of consecutive non-zero bins
    int    max_len;
    int    pad_[3];
    float  band_w[kMelMaxLen][BD_MEL_BANDS];  // band_w[j][m] = mel[band_start[m] + j][m]
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
    int pw_variant16;    // tile choice for the split-f16 kernel (0 = by shape)
    int pw_mode;         // 0 = exact f32 MFMA, 1 = split-f16 MFMA (3 products)
};

// ---- launchers (each enqueues exactly one kernel on `stream`) ----
void launch_logmel(const float* pcm, int64_t n_valid, int64_t n_frames, float* logmel,
                   const FeTables* tables, hipStream_t stream, int variant = 0);
void launch_resample(const void* in, bool s16, int64_t n_in, int channels, const float* taps, int half, int up,
                     int down, float* out, int64_t n_out, hipStream_t stream);
void launch_patches(const float* logmel, int64_t n_windows, int patch_step, float* patches,
                    hipStream_t stream);
void launch_conv1(const float* logmel, int patch_step, const WindowMap& map, int w0, int windows, const float* w9x32,
                  const float* b32, float* out, hipStream_t stream);
void launch_depthwise(const float* in, float* out, int windows, const SepLayer& L, hipStream_t stream);
void launch_pointwise(const float* in, float* out, int64_t rows, const SepLayer& L, hipStream_t stream);
int launch_pointwise_variant(const float* A, const float* Wt, const float* bias, float* C, long long M, int N,
                             int K, int variant, hipStream_t stream);
int launch_pointwise_f16x3_variant(const float* A, const void* Whi, const void* Wlo, const float* bias, float* C,
                                   long long M, int N, int K, int variant, hipStream_t stream);
bool launch_separable_fused(const float* in, float* out, int windows, const SepLayer& L, int variant,
                            hipStream_t stream);
bool launch_separable_fused_next_dw(const float* in, float* out, int windows, const SepLayer& L, const SepLayer& next,
                                    hipStream_t stream);
void launch_stem(const float* logmel, int patch_step, const WindowMap& map, int w0, int windows, const float* c1_w,
                 const float* c1_b,
                 const SepLayer& L2, float* out, hipStream_t stream);
void launch_stem3(const float* logmel, int patch_step, const WindowMap& map, int w0, int windows, const float* c1_w,
                  const float* c1_b,
                  const SepLayer& L2, const SepLayer& L3, float* out, hipStream_t stream);
void launch_stem4(const float* logmel, int patch_step, const WindowMap& map, int w0, int windows, const float* c1_w,
                  const float* c1_b,
                  const SepLayer& L2, const SepLayer& L3, float* out, hipStream_t stream);
void launch_pool_head(const float* act, int windows, const float* head_wt, const float* head_b,
                      int n_classes, float* emb, float* logits, hipStream_t stream);
void launch_head(const float* pooled, int windows, const float* head_wt, const float* head_b, int n_classes,
                 float* logits, hipStream_t stream);
bool launch_separable_fused_pool(const float* in, float* pooled, int windows, const SepLayer& L, hipStream_t stream);

}  // namespace bd
