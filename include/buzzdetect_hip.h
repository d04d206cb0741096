/*
 * buzzdetect_hip.h — C ABI of the MI355X (gfx950) engine for buzzdetect's analyze hot path.
 *
 * The reference (OSU-Bee-Lab/buzzdetect) is pure Python on TensorFlow; it has no FFI of its
 * own.  These entry points are what a binding for the hot path would call, one per reference
 * call site (paths relative to the reference checkout):
 *
 *   bd_create / bd_destroy   <- YamnetK2.initialize          embedders/yamnet_k2/embedder.py:14-24
 *                               EmbedderYamnet.initialize    embedders/yamnet/embedder.py:25-31
 *                               ModelGeneralV3.initialize    models/model_general_v3/model.py:11-16
 *   bd_padded_length         <- pad_waveform                 embedders/yamnet/features.py:82-108
 *   bd_num_frames            <- tf.signal.stft framing       embedders/yamnet/features.py:42-46
 *   bd_num_windows           <- tf.signal.frame(axis=0)      embedders/yamnet/features.py:65-76
 *   bd_resample              <- WorkerStreamer.queue_chunk: np.mean(axis=1) + librosa.resample
 *                                                            src/stream/worker.py:116-128
 *   bd_frontend              <- waveform_to_log_mel_spectrogram_patches (log-mel output)
 *                                                            embedders/yamnet/features.py:22-58
 *   bd_patches               <- ... (patch output)           embedders/yamnet/features.py:65-79
 *   bd_embed                 <- YamnetK2.embed / EmbedderYamnet.embed
 *                                                            embedders/yamnet_k2/embedder.py:27-37
 *                                                            embedders/yamnet/embedder.py:33-44
 *   bd_predict               <- ModelGeneralV3.predict       models/model_general_v3/model.py:18-31
 *   bd_predict_batch         <- WorkerInferer.process_chunk over several queued chunks
 *   bd_predict_chunks           (the same with one pointer per chunk: AssignChunk.samples as the streamer made them,
 *                                src/stream/worker.py:129-133, and this call's own range word for the writer thread
 *                                that reads the results later, src/write/worker.py:69)
 *                                                            src/inference/worker.py:71-92, src/analyze.py:218-253
 *   bd_calibrate             <- (no counterpart: TensorFlow computes the 1x1 convolutions in float32,
 *                                embedders/yamnet/yamnet.py:64-70; the f16 matrix path needs operand scales)
 *   bd_stage_tap             <- (test hook) any intermediate activation of yamnet()
 *                                                            embedders/yamnet/yamnet.py:96-103
 *
 * Conventions
 *   - Every function returns 0 on success or a negative BD_E* code; bd_last_error() gives the
 *     text for the calling thread.  The count helpers return the count, or a negative code.
 *   - The caller owns every device buffer (PyTorch allocates them); the library allocates only
 *     the folded weights/tables at bd_create.  No hidden synchronisation: work is enqueued on
 *     the stream passed in (a hipStream_t cast to void*; NULL = the legacy default stream).
 *   - One handle per host thread / per GPU (the reference builds one model per analyzer
 *     thread, src/inference/worker.py:21,78).  A handle is not thread-safe.
 *   - All device buffers are float32, dense, row-major.  Pointers need 16-byte alignment.
 *   - There is no CPU fallback: without a usable HIP device bd_create fails.
 */
#ifndef BUZZDETECT_HIP_H
#define BUZZDETECT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BD_ABI_VERSION 5

#if defined(__GNUC__)
#define BD_API __attribute__((visibility("default")))
#else
#define BD_API
#endif

/* sizes fixed by the reference architecture (embedders/yamnet/params.py, yamnet.py:77-93) */
#define BD_SAMPLE_RATE        16000
#define BD_MIN_SAMPLES        15600      /* int((0.96 + 0.025 - 0.010) * 16000) */
#define BD_STFT_WINDOW        400
#define BD_STFT_HOP           160
#define BD_FFT_LENGTH         512
#define BD_SPECTRUM_BINS      257
#define BD_MEL_BANDS          64
#define BD_PATCH_FRAMES       96
#define BD_EMBEDDING_SIZE     1024
#define BD_EMBEDDER_BLOB_FLOATS 3217344  /* payload of variables.data-00000-of-00001 / 4 */
#define BD_MAX_CLASSES        64
#define BD_NUM_STAGES         27         /* conv1, then (depthwise, pointwise) x 13 */

#define BD_OK                 0
#define BD_EINVAL            (-1)        /* bad argument (null, negative, misaligned, hop <= 0 ...) */
#define BD_ENODEVICE         (-2)        /* no usable HIP device / wrong architecture */
#define BD_EHIP              (-3)        /* a HIP runtime call failed */
#define BD_EWORKSPACE        (-4)        /* workspace too small for this call */
#define BD_ERANGE            (-5)        /* chunk too long: float32 ceil of pad_waveform is not exact (n >= 2^24) */
#define BD_EWEIGHTS          (-6)        /* weight blob / mel matrix not in the expected form */

typedef struct bd_engine* bd_handle;

/* Host-side weights, all in the reference's own layouts. */
typedef struct bd_weights {
    const float* embedder_blob;   /* BD_EMBEDDER_BLOB_FLOATS f32: the TensorBundle payload, unchanged
                                     (conv1 kernel [3,3,1,32], BN beta/mean/var, then per separable
                                     layer: depthwise [3,3,C,1], BN x3, pointwise [1,1,Cin,Cout], BN x3) */
    int64_t      embedder_floats; /* must equal BD_EMBEDDER_BLOB_FLOATS */
    const float* mel;             /* [257][64] linear-to-mel matrix as baked into the SavedModel graph */
    const float* head_kernel;     /* [1024][n_classes] Dense kernel (Keras layout), or NULL: embed only */
    const float* head_bias;       /* [n_classes] */
    int32_t      n_classes;       /* 0..BD_MAX_CLASSES */
} bd_weights;

BD_API int bd_abi_version(void);
BD_API const char* bd_last_error(void);

BD_API int bd_create(bd_handle* out, int device, const bd_weights* weights);
BD_API int bd_destroy(bd_handle h);

/* windows processed per pass through the CNN (activations of one pass stay cache-resident);
   0 restores the default. */
BD_API int bd_set_group_windows(bd_handle h, int32_t windows);

/* ---- index arithmetic (host only; bit-exact restatement incl. the float32 ceil) ---- */
BD_API int64_t bd_padded_length(int64_t n_samples, int32_t hop_samples);
BD_API int64_t bd_num_frames(int64_t n_samples, int32_t hop_samples);
BD_API int64_t bd_num_windows(int64_t n_samples, int32_t hop_samples, int32_t patch_step);

/* bytes of scratch bd_embed / bd_predict / bd_stage_tap need for such a chunk */
BD_API int64_t bd_workspace_bytes(bd_handle h, int64_t n_samples, int32_t hop_samples, int32_t patch_step);

/* ---- device entry points ---- */

/* pcm_dev[n_samples] -> logmel_dev[bd_num_frames][64]; samples past n_samples read as zero. */
BD_API int bd_frontend(bd_handle h, const float* pcm_dev, int64_t n_samples, int32_t hop_samples,
                float* logmel_dev, void* stream);

/* Streamer stage (src/stream/worker.py:116-128): mean over channels (float32, as np.mean computes it), then rational
   resample rate_in -> rate_out (up/down = the reduced rate ratio): y[j] = sum_i mono[i] h[j down - i up + half].
   The reference calls librosa.resample with its default res_type soxr_hq.  quality selects the low-pass h:
     BD_RESAMPLE_HQ (default)  the filter class of libsoxr's HQ recipe - linear phase, flat (< 1e-4 dB) to 0.9136 of the lower
                               Nyquist, >= 120 dB from that Nyquist on, unity gain, no net delay, ceil(n up / down) outputs -
                               as one Kaiser-windowed sinc (569 taps for 48 -> 16 kHz); integer decimations and every ratio
                               whose band fits four waves' registers run on the matrix cores (resample.hip), the rest on the
                               vector kernel.  libsoxr's source is absent from the reference checkout and it realises the
                               response as a cascade: same class, not the same bits ("parity unpinned").
     BD_RESAMPLE_SCIPY         rounds 1-3's filter: scipy.signal.resample_poly's default, Kaiser(5.0) windowed sinc of
                               20*max(up,down)+1 taps (about -30 dB one kHz into the stop band: NOT the reference's class).
   in_dev is [n_in][channels] interleaved; out_dev gets bd_resample_length() samples.  bd_resample_taps returns the tap count
   and (optionally) the float32 taps, so that a host restatement can use the identical filter. */
#define BD_RESAMPLE_SCIPY 0
#define BD_RESAMPLE_HQ    1
BD_API int64_t bd_resample_length(int64_t n_in, int32_t rate_in, int32_t rate_out);
BD_API int bd_resample_taps(int32_t rate_in, int32_t rate_out, int32_t quality, float* taps_host, int64_t capacity,
                            int32_t* up, int32_t* down, int32_t* half);
BD_API int bd_set_resample_quality(bd_handle h, int32_t quality);
/* Test hook (host only): the matrix-core plan of a rate ratio at BD_RESAMPLE_HQ (resample.hip).  Returns 1 and fills
   geometry[14] = {up, down, P, D, NB, kq, mt, contiguous, RS, a_bytes, lds_bytes, half, bits of unscale_s16, bits of
   unscale_f32}, boff[NB] and gfrag[NB][4 kq][2][64][8] (f16 bits) when the ratio runs on fir_mfma_kernel, 0 when it runs on
   the vector kernel.  tests/test_resample.py multiplies the fragments out on the CPU against the oracle's filter. */
BD_API int bd_debug_fir_plan(int32_t rate_in, int32_t rate_out, int32_t* geometry, int32_t* boff, int64_t boff_capacity,
                             uint16_t* gfrag, int64_t gfrag_capacity);
/* Host only: 1 when bd_resample / bd_resample_s16 accept rate_in -> rate_out at `quality` (BD_RESAMPLE_*), 0 when they refuse
   it with BD_EINVAL - the ratio does not reduce to <= 4096, or its low-pass is longer than the span the vector kernel
   stages per tile and the matrix-core form does not fit either (HQ: down / up > ~43, e.g. 768 kHz -> 16 kHz).  The
   reference resamples any rate (librosa.resample, src/stream/worker.py:128); such a file has to be decimated in two steps. */
BD_API int bd_resample_supported(int32_t rate_in, int32_t rate_out, int32_t quality);
BD_API int bd_resample(bd_handle h, const float* in_dev, int64_t n_in, int32_t channels, int32_t rate_in,
                       int32_t rate_out, float* out_dev, void* stream);
/* the same from 16-bit PCM (value / 32768, libsndfile's float convention): half the PCIe bytes; with
   rate_in == rate_out and one channel it is a plain s16 -> f32 conversion on the device */
BD_API int bd_resample_s16(bd_handle h, const int16_t* in_dev, int64_t n_in, int32_t channels, int32_t rate_in,
                           int32_t rate_out, float* out_dev, void* stream);

/* Host only (no device, no handle): the result rows of one chunk as the CSV text the reference's writer produces
   (src/write/formatting.py:31-50 add_time + round(2), then DataFrame.to_csv in src/write/worker.py:67-87): per row
   `start`, then the kept activation columns, each rounded to two decimals as numpy's float32 round does and written in
   shortest form with at least one decimal ("-1.28", "0.5", "3.0", "-0.0"), '\n' line ends, no header.
     values[n_rows][row_stride] float32 logits, n_cols of them valid per row; keep[n_keep] = the columns to write, in order
     (n_keep == 0: all n_cols); starts[n_rows] = the rows' start times, already rounded (framing.window_starts);
     out / capacity: at least n_rows * (10 * columns + 12) + 8 bytes.
   Returns the number of bytes written; BD_ERANGE when a value has no two-decimal fixed form below 100 000 (non-finite or
   huge - the caller then formats that chunk the slow way, as buzzdetect_amd/fastcsv.py does); BD_EWORKSPACE / BD_EINVAL. */
BD_API int64_t bd_format_rows(const float* values, int64_t n_rows, int32_t n_cols, int64_t row_stride, const int32_t* keep,
                              int32_t n_keep, const double* starts, char* out, int64_t capacity);

/* ---- the streamer's way onto the device (round 6; ABI 5) ----
   The reference's streamer reads a chunk with soundfile and hands the analyzer a host array (src/stream/worker.py:109-135);
   here a chunk travels file -> device in pieces through a few small page-locked buffers, so that a process page-locks
   n_stage * stage_bytes per reader thread once (16 MB) instead of one chunk-sized buffer per chunk in flight (~1 GB at
   0.07-0.25 s per GB inside the first analyze() call) and the reader thread holds no interpreter lock
   while it works.  One stager per reader thread; not thread-safe.
     bd_stager_create   page-locks n_stage (1..4) buffers of stage_bytes on `device`, one event per buffer
     bd_stager_read     nbytes from file descriptor fd at `offset` (pread; short at the end of the file) -> dev[0 .. n), each
                        piece copied by hipMemcpyAsync on `stream` behind the read of the next; returns the bytes read and
                        enqueued.  The caller orders its kernels behind the copies with an event recorded on `stream`.
     bd_stager_acquire  the next buffer, once the copy that last read it has completed: index and host pointer (for sample
                        formats that are converted on the host before they travel)
     bd_stager_submit   copy the first nbytes of buffer `index` to dev on `stream` */
typedef struct bd_stager* bd_stager_t;
BD_API int bd_stager_create(bd_stager_t* out, int32_t device, int64_t stage_bytes, int32_t n_stage);
BD_API int bd_stager_destroy(bd_stager_t st);
BD_API int64_t bd_stager_read(bd_stager_t st, int32_t fd, int64_t offset, int64_t nbytes, void* dev, void* stream);
BD_API int bd_stager_acquire(bd_stager_t st, int32_t* index, void** host);
BD_API int bd_stager_submit(bd_stager_t st, int32_t index, int64_t nbytes, void* dev, void* stream);

/* logmel_dev[n_frames][64] -> patches_dev[W][96][64], W = 1 + (n_frames - 96) / patch_step. */
BD_API int bd_patches(bd_handle h, const float* logmel_dev, int64_t n_frames, int32_t patch_step,
               float* patches_dev, void* stream);

/* pcm_dev[n_samples] -> emb_dev[W][1024] */
BD_API int bd_embed(bd_handle h, const float* pcm_dev, int64_t n_samples, int32_t hop_samples,
             int32_t patch_step, void* workspace_dev, int64_t workspace_bytes,
             float* emb_dev, void* stream);

/* pcm_dev[n_samples] -> logits_dev[W][n_classes]; emb_dev may be NULL. */
BD_API int bd_predict(bd_handle h, const float* pcm_dev, int64_t n_samples, int32_t hop_samples,
               int32_t patch_step, void* workspace_dev, int64_t workspace_bytes,
               float* emb_dev, float* logits_dev, void* stream);

/* Several chunks in one launch set: the analyze-side batching of src/analyze.py:218-253 +
   src/inference/worker.py:71-74 (the reference calls predict once per chunk; here windows of up to 64 chunks
   share every kernel launch).  pcm_dev holds the chunks back to back, chunk c has chunk_samples[c] samples
   (host array) and is padded on its own exactly as in a
   single-chunk call (hazard H1), so the rows equal those of n_chunks separate bd_predict calls, bit for bit.
   Outputs are [sum of windows][1024] / [sum of windows][n_classes]; either may be NULL. */
BD_API int64_t bd_batch_num_windows(const int64_t* chunk_samples, int32_t n_chunks, int32_t hop_samples,
                                    int32_t patch_step, int64_t* per_chunk_windows /* may be NULL */);
BD_API int64_t bd_batch_workspace_bytes(bd_handle h, const int64_t* chunk_samples, int32_t n_chunks,
                                        int32_t hop_samples, int32_t patch_step);
BD_API int bd_predict_batch(bd_handle h, const float* pcm_dev, const int64_t* chunk_samples, int32_t n_chunks,
                            int32_t hop_samples, int32_t patch_step, void* workspace_dev, int64_t workspace_bytes,
                            float* emb_dev, float* logits_dev, void* stream);
/* The general form.  chunk_pcm_dev: HOST array of n_chunks device pointers, one per chunk (4-byte aligned; the chunks need
   not be adjacent, so nothing is concatenated on the device).  mode: arithmetic of the 1x1 convolutions for THIS call
   only, -1 = the handle's (bd_set_pointwise_mode) - the exact-f32 repeat of a flagged chunk does not touch the handle's
   state.  range_word: int32 in device memory or in PINNED (device-mapped) host memory, or NULL.  When given, this call's
   kernels raise THAT word (they store 1 into it; the caller zeroes it beforehand) instead of the engine's sticky one
   (see bd_range_flag): the word says whether THIS call left the f16 range, however many other calls are queued behind it
   when somebody finally looks - the reference reads results on its writer thread while the analyzer has already
   enqueued the next chunk.  No copy and no reset is enqueued: a pinned word costs nothing unless a chunk overflows.
   Read it after an event recorded behind the call. */
BD_API int bd_predict_chunks(bd_handle h, const float* const* chunk_pcm_dev, const int64_t* chunk_samples, int32_t n_chunks,
                             int32_t hop_samples, int32_t patch_step, void* workspace_dev, int64_t workspace_bytes,
                             float* emb_dev, float* logits_dev, int32_t mode, int32_t* range_word, void* stream);

/* Test hook: run the path up to CNN stage `stage` (0 = conv1 output, 2k-1 / 2k = depthwise /
   pointwise output of layer k+1) for the first `windows` windows and copy that NHWC activation
   to out_dev ([windows][H][W][C] floats; bd_stage_shape gives H, W, C). */
BD_API int bd_stage_shape(int32_t stage, int32_t* h, int32_t* w, int32_t* c);
BD_API int bd_stage_tap(bd_handle h, const float* pcm_dev, int64_t n_samples, int32_t hop_samples,
                 int32_t patch_step, void* workspace_dev, int64_t workspace_bytes,
                 int32_t stage, int32_t windows, float* out_dev, void* stream);

/* Tuning / test hooks for the pointwise (1x1 convolution) GEMM, yamnet.py:64-70 after BN folding:
   c[m][n] = relu(sum_k a[m][k] * wt[n][k] + bias[n]); k % 32 == 0, n % 64 == 0.
   variant 0 = the library's choice for the shape; 1..8 = explicit tile shapes (see cnn.hip). */
BD_API int bd_debug_pointwise(const float* a_dev, const float* wt_dev, const float* bias_dev, float* c_dev,
                              int64_t m, int32_t n, int32_t k, int32_t variant, void* stream);
/* Per layer, in the f16 modes: 0 = the library's choice (layers 5 and 7: weights in registers, pw_res_kernel; 128+ input
   channels otherwise: the wave-specialised tile kernel; else the plain split-f16 GEMM); 1..9 = tile shapes of the plain GEMM;
   10 = always the wave-specialised tile kernel; 11 = as 0.  All variants give the same bits. */
BD_API int bd_set_pointwise_variant(bd_handle h, int32_t layer /* 2..14 */, int32_t variant);

/* Arithmetic of the 1x1 convolutions: 0 = v_mfma_f32_32x32x2_f32 (exact f32 products),
   1 = split-f16 (default): every f32 operand carried as hi + lo halves, three f16 MFMAs per product,
   f32 accumulate; same accuracy class as f32, ~5x the matrix-core rate;
   2 = plain f16 operands, ONE MFMA per product, f32 accumulate (BASELINE config 5's arithmetic): outside the
   reference's 1e-4 logit tolerance by design (~1e-3), for callers that trade accuracy for rate. */
BD_API int bd_set_pointwise_mode(bd_handle h, int32_t mode);
/* Operand scaling of modes 1 and 2.  f16 has 5 exponent bits: hi = f16(x), lo = f16(x - hi) carry x to 22 bits only
   while lo is a normal number (|x| >= 2^-3) and hi is finite (|x| <= 65 504).  Both operands of every 1x1 convolution are
   therefore moved into that window by exact powers of two: each output channel of the folded kernel so that its largest
   weight lies in [2^12, 2^13) (fixed at bd_create), each layer's input activations so that the largest value a
   calibration pass saw lies in [2^8, 2^9) (folded into the taps and shift of the depthwise that produces them, so it
   costs no instruction); the epilogue multiplies the accumulator by the inverse power of two.  Nothing is rounded by the
   scales, so the results do not depend on the scale BatchNorm folding leaves a layer at.
   bd_create calibrates on a built-in signal (silence, noise at five levels, tones, clicks, a chirp, square waves: the
   envelope of what [-1, 1] PCM does to a LOG-mel input) with exact-f32 arithmetic.  bd_calibrate runs the same pass over
   caller-supplied audio and widens the per-layer maxima (they never shrink).  It waits for `stream` and rewrites device
   tables with blocking copies: call it while nothing else is in flight on the handle.
   bd_get_scales: act_exp[13] / act_max[13] for layers 2..14 (either may be NULL); returns 13.
   bd_set_activation_exponents: test hook that overrides the calibrated exponents (-60..60), e.g. to drive a layer out of
   range and exercise the range word. */
BD_API int bd_calibrate(bd_handle h, const float* pcm_dev, int64_t n_samples, int32_t hop_samples, int32_t patch_step,
                        void* workspace_dev, int64_t workspace_bytes, void* stream);
BD_API int bd_get_scales(bd_handle h, int32_t* act_exp, float* act_max);
BD_API int bd_set_activation_exponents(bd_handle h, const int32_t* act_exp);
/* An activation beyond the calibrated headroom (scaled value > 65 504; the network has plain ReLU, yamnet.py:36-74, nothing
   bounds it) would become +inf and the result garbage.  The kernels track the largest magnitude they convert and set a
   sticky per-engine device word when it leaves the f16 range.  bd_range_flag copies that word to *flag_host (1 = some
   launch since the last reset overflowed: repeat those chunks in mode 0) and, with reset != 0, clears it.  It waits for
   `stream` - call it where the results are read anyway.  bd_predict_chunks hands out the word per call instead. */
BD_API int bd_range_flag(bd_handle h, int32_t* flag_host, int32_t reset, void* stream);
/* The same without waiting: enqueues a copy of the word to `dst` (device memory or PINNED host memory) on `stream`,
   then, with reset != 0, its clearing.  For pipelines that read results through their own events. */
BD_API int bd_range_flag_copy(bd_handle h, int32_t* dst, int32_t reset, void* stream);

/* Kernel fusion (both on by default; 0 = one kernel per op, the layout the stage taps use and the reference every fused
   kernel is compared against bit for bit).  Round 6 cut the codes to: per layer group the default, one kernel per op, and at
   most ONE previous form for same-box A/B; everything else is refused with BD_EINVAL.
   stem == 3       (default) layers 1-3 as one kernel that reads log-mel patches and writes the [24][16][128] layer-3 output
                   (the 402 MB layer-2 tensor never reaches HBM); profile slot 5.  The kernel hands the layer-2 tile to layer 3's
                   depthwise in REGISTERS and carries the rows two tiles share (stemreg.hip; in the exact-f32 mode the same
                   scheme on v_mfma_f32_32x32x2_f32, stemregf32.hip);
   stem == 5       as 3 on the kernel of rounds 2-4 (a workgroup per row block, the layer-2 tile through LDS: stem3_kernel;
                   split-f16 modes only - the exact-f32 mode has the one form);
   separable == 1  (default) layer 4 + depthwise 5 a window per workgroup (l4_window_kernel); pointwise 5, layer 6, depthwise 7
                   and pointwise 7 one on-chip launch, a window per tile (sepmid.hip; timed in layer 7's pointwise slot);
                   layers 8-12 + the stride-2 depthwise of layer 13 ONE launch in which every workgroup takes its four windows
                   through the five layers with the tiles between the layers kept on the CU - accumulators -> depthwise in
                   registers -> LDS ring (sepchip.hip; timed in layer 12's pointwise slot), its output written as the f16 hi / lo
                   planes the next launch reads; pointwise 13 with depthwise 14 on its accumulators and pointwise 14 with the
                   average pool, two launches of one matrix kernel that gives every SIMD of the chip one 96 x 64 tile
                   (septail.hip).  The exact-f32 mode (bd_set_pointwise_mode 0): layers 1-3 one f32-MFMA
                   kernel, layer 4 + depthwise 5 another, the two on-chip launches with f32 stage tiles (sepmidf32.hip,
                   sepchipf32.hip), layers 13 / 14 as 1x1 kernels with the next depthwise / the pool in their epilogue;
   separable == 10 as 1 with layers 5-7 on the four kernels of round 4 (pointwise 5, layer 6 + depthwise 7, pointwise 7).
   Removed in round 6 (BD_EINVAL): stem 2 (layers 1-2 + depthwise 3 only), stem 4 (the walk of stemroll.hip); separable 2
   (layer 4 as band tiles), 3 (a launch per layer for layers 8-11), 4 / 5 (layer 12 / 14 on the 8-wave kernel), 6 (one
   exact-f32 kernel per separable layer, sepf32.hip), 7 (layers 8-11 as the round-3 run through global memory, layers 12 and
   14 on the 12-wave kernel: deleted with that kernel when layers 13 / 14 moved to septail.hip), 8 (the on-chip run ending at
   layer 11), 9 / 12 (plain fused layers).
   With stem == 0 or during calibration / stage taps inside a fused group: one kernel per op.  Fused and unfused paths give
   bit-identical results. */
BD_API int bd_set_fusion(bd_handle h, int32_t stem, int32_t separable);
/* whi/wlo: [n][k] f16 halves of wt * scale[n] (wt[n][:] * scale[n] ~= whi[n][:] + wlo[n][:]); unscale[n] = 1 / (scale[n] *
   the scale the caller applied to a): c = relu(fma(acc, unscale[n], bias[n])) */
BD_API int bd_debug_pointwise_f16x3(const float* a_dev, const void* whi_dev, const void* wlo_dev,
                                    const float* unscale_dev, const float* bias_dev, float* c_dev, int64_t m, int32_t n,
                                    int32_t k, int32_t variant, void* stream);

/* ---- per-stage timing (HIP events on the caller's stream) ----
   With profiling on, one event is recorded at the head of every bd_predict/bd_embed call and one after
   each kernel launch; the interval between consecutive events is charged to the later launch's slot.
   bd_profile_read synchronises on them and accumulates per slot: 0 = front end, 1 = conv1,
   2..27 = stages 1..26 (depthwise/pointwise alternating), 28 = pool+head.
   ms[i] += elapsed, launches[i] += count; returns the number of slots (29). */
#define BD_PROFILE_SLOTS 29
BD_API int bd_profile_enable(bd_handle h, int32_t on);
BD_API int bd_profile_read(bd_handle h, double* ms, int64_t* launches, int32_t slots);

#ifdef __cplusplus
}
#endif
#endif /* BUZZDETECT_HIP_H */
