// C-ABI host layer of libbuzzdetect_hip.so (include/buzzdetect_hip.h): weight folding, the
// per-chunk launch plan, index arithmetic and per-stage event timing.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <memory>
#include <string>
#include <vector>

#include "bd_internal.h"

namespace {

thread_local std::string g_error;

int fail(int code, const std::string& msg) {
    g_error = msg;
    return code;
}

#define BD_HIP(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(BD_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));            \
    } while (0)

// (stride, filters): embedders/yamnet/yamnet.py:77-93
const int kLayerDefs[14][2] = {{2, 32},  {1, 64},  {2, 128}, {1, 128}, {2, 256}, {1, 256},  {2, 512},
                               {1, 512}, {1, 512}, {1, 512}, {1, 512}, {1, 512}, {2, 1024}, {1, 1024}};

constexpr int64_t kFloatsA = 98304;   // largest activation per window held in buffer A (layer 2 output 48x32x64)
constexpr int64_t kFloatsB = 49152;   // largest per window in buffer B (layer-2 depthwise output 48x32x32)
constexpr int kDefaultGroup = 1024;

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// One event is recorded after every launch; the time between consecutive events on the stream is
// charged to the later one's slot (slot < 0 = the marker at the head of a call).  Event pairs around
// each launch leave the gaps between pairs unattributed, and on this stack they under-read some
// kernels by tens of microseconds against rocprofv3.
struct Event2 {
    hipEvent_t ev;
    int slot;
};

}  // namespace

struct bd_engine {
    int device = 0;
    int n_classes = 0;
    int group_windows = kDefaultGroup;
    int pointwise_mode = 1;           // 0 = exact f32 MFMA, 1 = split-f16 MFMA, 2 = plain f16 MFMA
    unsigned* d_range_flag = nullptr; // sticky: an activation exceeded the f16 range in mode 1 / 2 (bd_range_flag)
    // bd_set_fusion (round 6: per layer group the default, one kernel per op, and at most ONE previous form):
    bool fuse_stem = true;            // layers 1-3 as one kernel (stem != 0)
    bool stem_reg = true;             // ... with the layer-2 tile handed over in registers (stemreg.hip; stem = 3, default); 5: through LDS
    bool fuse_sep = true;             // layers 4-14 on the fused kernels (separable != 0)
    bool chip_mid = true;             // pointwise 5 -> layer 6 -> depthwise 7 -> pointwise 7 as one on-chip launch (sepmid.hip);
                                      // separable = 10: the four kernels of round 4.  (Layers 8-12 + depthwise 13 as one on-chip launch
                                      // - sepchip.hip - and layers 13 / 14 on septail.hip's kernel have no other fused form left.)
    float* d_pool = nullptr;          // one allocation for every folded tensor
    bd::FeTables* d_tables = nullptr;
    // operand scaling of the f16 modes (bd_internal.h, SepLayer): host copies of what the scaled tensors are made from
    std::vector<float> h_dw[13];      // [10][cin]: folded depthwise taps, then the shift (unscaled)
    std::vector<int> row_exp[13];     // [cout]: power-of-two exponent the pointwise row was scaled by before its f16 split
    size_t off_dw16[13] = {0}, off_pw_u[13] = {0};   // float offsets of dw_w16 (dw_b16 follows at + 9 cin) / pw_u in d_pool
    float act_max[13] = {0};          // largest depthwise output seen by the calibration passes (exact-f32 arithmetic)
    int act_exp[13] = {0};
    unsigned* d_amax = nullptr;       // [16] calibration words
    const float* conv1_w = nullptr;   // [9][32]
    const float* conv1_b = nullptr;   // [32]
    bd::SepLayer sep[13];
    const float* head_wt = nullptr;   // [n_classes][1024]
    const float* head_b = nullptr;
    // resampler: the filters of every (up, down) pair used so far, kept on the device until bd_destroy (nothing is
    // freed or re-uploaded on a rate change: hipFree would synchronise the device under the caller's streams)
    struct Taps {
        int up, down, quality, half;
        float* dev;                   // the taps as float32 (the vector kernels), then, 16-byte aligned, the matrix-core plan's
                                      // fragment-ordered filter and its two index tables (bd::FirPlan points into this block)
        std::vector<float> host;      // source of the asynchronous upload; must outlive it
        bool has_plan;                // the ratio fits fir_mfma_kernel (resample.hip)
        bd::FirPlanHost fir;
        hipStream_t upload_stream;    // a launch on any OTHER stream first waits for `uploaded`
        hipEvent_t uploaded;
    };
    std::vector<Taps> taps;
    int resample_quality = BD_RESAMPLE_HQ;
    // profiling
    bool profiling = false;
    std::vector<Event2> pending;
    std::vector<Event2> free_events;
    double ms[BD_PROFILE_SLOTS] = {0};
    int64_t launches[BD_PROFILE_SLOTS] = {0};
};

namespace {

// Records the marker event of one launch (or, with slot < 0, the head marker of a call).
struct Scope {
    bd_engine* e;
    hipStream_t s;
    int slot;
    bool on;
    static void mark(bd_engine* e, hipStream_t s, int slot) {
        Event2 ev;
        if (!e->free_events.empty()) {
            ev = e->free_events.back();
            e->free_events.pop_back();
        } else if (hipEventCreate(&ev.ev) != hipSuccess) {
            return;
        }
        ev.slot = slot;
        (void)hipEventRecord(ev.ev, s);
        e->pending.push_back(ev);
    }
    Scope(bd_engine* e_, hipStream_t s_, int slot_) : e(e_), s(s_), slot(slot_), on(e_->profiling) {}
    ~Scope() {
        if (on) mark(e, s, slot);
    }
};

// Developer build (-DBD_KERNEL_TRACE) only: BD_REPEAT_SLOT=<profile slot> BD_REPEAT_N=<n> launches that slot's kernel n times
// per pass (same operands: the kernels are idempotent), so that one kernel dominates a run - its sustained clock and the
// board power it draws alone can then be read from hwmon (tools/power_profile.py).  The shipped library reads no environment.
#ifdef BD_KERNEL_TRACE
static int repeat_of(int slot) {
    static const int want = getenv("BD_REPEAT_SLOT") ? atoi(getenv("BD_REPEAT_SLOT")) : -1;
    static const int n = getenv("BD_REPEAT_N") ? atoi(getenv("BD_REPEAT_N")) : 1;
    return slot == want ? n : 1;
}
#define BD_REPEAT_EXTRA(slot) for (int rep_ = repeat_of(slot) - 1; rep_ > 0; --rep_)
#else
#define BD_REPEAT_EXTRA(slot) if (false)
#endif

// ---- index arithmetic: embedders/yamnet/features.py:82-108, :42-46, :65-76 ----
int64_t padded_length(int64_t n, int32_t hop) {
    const int64_t min_samples = BD_MIN_SAMPLES;
    int64_t pad = min_samples - n > 0 ? min_samples - n : 0;
    const int64_t num = n > min_samples ? n : min_samples;
    const int64_t after = num - min_samples;
    // tf.cast(tf.math.ceil(tf.cast(after, f32) / tf.cast(hop, f32)), int32): float32 on purpose
    const volatile float q = (float)after / (float)hop;
    const int64_t hops = (int64_t)ceilf(q);
    pad += (int64_t)hop * hops - after;
    return n + pad;
}

struct Geometry {
    int64_t n_padded, n_frames, n_windows;
};

int geometry(int64_t n, int32_t hop, int32_t step, Geometry* g) {
    if (n < 0) return fail(BD_EINVAL, "n_samples must be >= 0");
    if (hop <= 0) return fail(BD_EINVAL, "hop_samples must be > 0");
    if (n >= (1LL << 24))
        return fail(BD_ERANGE,
                    "chunk of 2^24 samples or more: the float32 ceil in pad_waveform (features.py:100-102) is "
                    "no longer exact; split the chunk");
    g->n_padded = padded_length(n, hop);
    if (g->n_padded < n) return fail(BD_ERANGE, "pad_waveform would need negative padding");
    g->n_frames = g->n_padded >= BD_STFT_WINDOW ? 1 + (g->n_padded - BD_STFT_WINDOW) / BD_STFT_HOP : 0;
    if (step > 0)
        g->n_windows = g->n_frames >= BD_PATCH_FRAMES ? 1 + (g->n_frames - BD_PATCH_FRAMES) / step : 0;
    else
        g->n_windows = 0;
    return BD_OK;
}

// ---- resampler design (restated in oracle/resample_oracle.py, tests/test_resample.py holds the two together) ----
//   quality BD_RESAMPLE_HQ (default): the filter CLASS of libsoxr's HQ recipe, which is what the reference's
//     librosa.resample(...) runs (src/stream/worker.py:128, res_type soxr_hq): linear phase, pass band to
//     1 - 0.05 / TO_3dB(rej) = 0.9136 of the lower Nyquist (rej = 20 bits * 6.02 dB), stop band from that Nyquist on,
//     as one Kaiser-windowed sinc at the up-sampled rate: -6 dB point midway between the band edges, design attenuation
//     125 dB, beta = 0.1102 (A - 8.7), N = (A - 7.95) / (2.285 dw) + 1 taps (569 for 48 -> 16 kHz);
//   quality BD_RESAMPLE_SCIPY: rounds 1-3's filter, scipy.signal.resample_poly's default:
//     half = 10 * max(up, down); h = firwin(2 half + 1, 1 / max(up, down), window = ('kaiser', 5.0)) * up
double bessel_i0(double x) {
    double sum = 1.0, term = 1.0;
    for (int k = 1; k < 256; ++k) {
        term *= (x / (2.0 * k)) * (x / (2.0 * k));
        sum += term;
        if (term < 1e-18 * sum) break;
    }
    return sum;
}

void rational_ratio(int32_t rate_in, int32_t rate_out, int* up, int* down) {
    int a = rate_out, b = rate_in;
    while (b) {
        const int t = a % b;
        a = b;
        b = t;
    }
    *up = rate_out / a;
    *down = rate_in / a;
}

// unity-DC-gain Kaiser-windowed sinc of 2 half + 1 taps, cutoff (-6 dB) as a fraction of Nyquist, times `gain`
std::vector<double> kaiser_lowpass(int half, double cutoff, double beta, double gain) {
    const int n = 2 * half + 1;
    std::vector<double> h(n);
    double sum = 0.0;
    const double i0b = bessel_i0(beta);
    for (int i = 0; i < n; ++i) {
        const double m = i - half;
        const double x = cutoff * m;
        const double sinc = m == 0 ? 1.0 : std::sin(M_PI * x) / (M_PI * x);
        const double r = half > 0 ? m / half : 0.0;
        const double w = bessel_i0(beta * std::sqrt(r * r < 1.0 ? 1.0 - r * r : 0.0)) / i0b;
        h[i] = cutoff * sinc * w;
        sum += h[i];
    }
    for (int i = 0; i < n; ++i) h[i] = h[i] / sum * gain;
    return h;
}

std::vector<double> design_taps(int up, int down, int quality, int* half_out) {
    const int max_rate = up > down ? up : down;
    if (quality == BD_RESAMPLE_SCIPY) {
        const int half = 10 * max_rate;
        *half_out = half;
        return kaiser_lowpass(half, 1.0 / max_rate, 5.0, (double)up);
    }
    if (max_rate == 1) {                           // equal rates: a copy
        *half_out = 0;
        return std::vector<double>(1, 1.0);
    }
    const double rej = 20.0 * 20.0 * std::log10(2.0);                       // libsoxr HQ: 20-bit precision
    const double to_3db = (1.6e-6 * rej - 7.5e-4) * rej + 0.646;
    const double fp = (1.0 - 0.05 / to_3db) / max_rate, fs = 1.0 / max_rate;   // fractions of the up-sampled Nyquist
    const double att = 125.0;
    const double beta = 0.1102 * (att - 8.7);
    const int n = (int)std::ceil((att - 7.95) / (2.285 * M_PI * (fs - fp))) + 1;
    const int half = n / 2;
    *half_out = half;
    return kaiser_lowpass(half, 0.5 * (fp + fs), beta, (double)up);
}

bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }

// ---- weight preparation ----
struct BnFold {
    std::vector<double> scale, shift;
};

BnFold fold_bn(const float* beta, const float* mean, const float* var, int c) {
    // keras BatchNormalization inference with scale=False: (x - mean) * rsqrt(var + eps) + beta
    BnFold f;
    f.scale.resize(c);
    f.shift.resize(c);
    for (int i = 0; i < c; ++i) {
        const double s = 1.0 / std::sqrt((double)var[i] + 1e-4);
        f.scale[i] = s;
        f.shift[i] = (double)beta[i] - (double)mean[i] * s;
    }
    return f;
}

int build_tables(const float* mel, bd::FeTables* t) {
    std::memset(t, 0, sizeof(*t));
    // tf.signal.hann_window(400, periodic=True, dtype=float32): float32 arithmetic throughout
    for (int k = 0; k < BD_STFT_WINDOW; ++k) {
        const float arg = (float)(2.0 * M_PI) * (float)k / (float)BD_STFT_WINDOW;
        t->hann[k] = 0.5f - 0.5f * (float)std::cos((double)arg);
    }
    for (int k = 0; k < 256; ++k) {
        const double a = -2.0 * M_PI * k / 256.0;
        t->tw256[k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    for (int k = 0; k <= BD_SPECTRUM_BINS; ++k) {
        const double a = -2.0 * M_PI * k / 512.0;
        t->tw512[k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    for (int m = 0; m < BD_MEL_BANDS; ++m) {
        const int first = bd::kMelStart[m], len = bd::kMelLen[m];
        for (int k = 0; k < BD_SPECTRUM_BINS; ++k) {
            const float w = mel[k * BD_MEL_BANDS + m];
            if (!std::isfinite(w)) return fail(BD_EWEIGHTS, "mel matrix has a non-finite entry");
            if (w != 0.0f && (k < first || k >= first + len))
                return fail(BD_EWEIGHTS, "mel matrix has a non-zero outside the 64-band YAMNet filterbank pattern "
                                         "(linear_to_mel_weight_matrix(64, 257, 16000, 125, 7500)); not a YAMNet front end");
        }
        for (int j = 0; j < len; ++j) t->melw[bd::mel_offset(m) + j] = mel[(first + j) * BD_MEL_BANDS + m];
    }
    return BD_OK;
}

int apply_scales(bd_engine* e);
int calibrate_builtin(bd_engine* e);

}  // namespace

extern "C" {

int bd_abi_version(void) { return BD_ABI_VERSION; }

const char* bd_last_error(void) { return g_error.c_str(); }

int64_t bd_padded_length(int64_t n_samples, int32_t hop_samples) {
    Geometry g;
    const int rc = geometry(n_samples, hop_samples, 0, &g);
    return rc < 0 ? rc : g.n_padded;
}

int64_t bd_num_frames(int64_t n_samples, int32_t hop_samples) {
    Geometry g;
    const int rc = geometry(n_samples, hop_samples, 0, &g);
    return rc < 0 ? rc : g.n_frames;
}

int64_t bd_num_windows(int64_t n_samples, int32_t hop_samples, int32_t patch_step) {
    if (patch_step <= 0) return fail(BD_EINVAL, "patch_step must be > 0");
    Geometry g;
    const int rc = geometry(n_samples, hop_samples, patch_step, &g);
    return rc < 0 ? rc : g.n_windows;
}

int bd_stage_shape(int32_t stage, int32_t* h, int32_t* w, int32_t* c) {
    if (stage < 0 || stage >= BD_NUM_STAGES || !h || !w || !c) return fail(BD_EINVAL, "bad stage");
    int hh = 96, ww = 64, cc = 1;
    // stage 0 = conv1; stage 2k-1 = depthwise of layer k+1, stage 2k = its pointwise
    hh = 48, ww = 32, cc = 32;
    for (int s = 1; s <= stage; ++s) {
        const int layer = (s + 1) / 2;   // index into kLayerDefs (1..13)
        if (s & 1) {                     // depthwise
            if (kLayerDefs[layer][0] == 2) {
                hh /= 2;
                ww /= 2;
            }
        } else {
            cc = kLayerDefs[layer][1];
        }
    }
    *h = hh;
    *w = ww;
    *c = cc;
    return BD_OK;
}

int bd_create(bd_handle* out, int device, const bd_weights* w) {
    if (!out || !w) return fail(BD_EINVAL, "bd_create: null argument");
    *out = nullptr;
    if (!w->embedder_blob || !w->mel) return fail(BD_EINVAL, "bd_create: embedder_blob and mel are required");
    if (w->embedder_floats != BD_EMBEDDER_BLOB_FLOATS)
        return fail(BD_EWEIGHTS, "bd_create: embedder blob must hold exactly 3217344 floats "
                                 "(payload of variables.data-00000-of-00001)");
    if (w->n_classes < 0 || w->n_classes > BD_MAX_CLASSES) return fail(BD_EINVAL, "bd_create: n_classes out of range");
    if (w->n_classes > 0 && (!w->head_kernel || !w->head_bias))
        return fail(BD_EINVAL, "bd_create: head_kernel/head_bias missing");

    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(BD_ENODEVICE, "bd_create: no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= count) return fail(BD_ENODEVICE, "bd_create: device index out of range");
    hipDeviceProp_t prop;
    BD_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(BD_ENODEVICE, std::string("bd_create: kernels are built for gfx950 only, device is ") +
                                      prop.gcnArchName);
    BD_HIP(hipSetDevice(device));

    // ---- fold + lay out every tensor in one host staging buffer ----
    std::vector<float> host;
    host.reserve(BD_EMBEDDER_BLOB_FLOATS + 64 * 1024);
    auto reserve = [&](size_t n) {
        const size_t off = align_up((int64_t)host.size(), 64);
        host.resize(off + n, 0.0f);
        return off;
    };
    const float* p = w->embedder_blob;
    size_t off_conv1_w, off_conv1_b;
    size_t off_dw_w[13], off_dw_b[13], off_pw_w[13], off_pw_b[13], off_pw_hi[13], off_pw_lo[13], off_pw_fhi[13], off_pw_flo[13], off_pw_ffrag[13];
    size_t off_dw16[13], off_pw_u[13];
    std::vector<int> row_exps[13];
    {
        const int c = kLayerDefs[0][1];
        const float* kern = p;   // [3][3][1][32]
        const BnFold f = fold_bn(p + 9 * c, p + 10 * c, p + 11 * c, c);
        p += 12 * c;
        off_conv1_w = reserve(9 * c);
        off_conv1_b = reserve(c);
        for (int t = 0; t < 9; ++t)
            for (int i = 0; i < c; ++i) host[off_conv1_w + t * c + i] = (float)((double)kern[t * c + i] * f.scale[i]);
        for (int i = 0; i < c; ++i) host[off_conv1_b + i] = (float)f.shift[i];
    }
    int cin = kLayerDefs[0][1];
    for (int l = 0; l < 13; ++l) {
        const int cout = kLayerDefs[l + 1][1];
        const float* dw = p;   // [3][3][cin][1]
        const BnFold fd = fold_bn(p + 9 * cin, p + 10 * cin, p + 11 * cin, cin);
        p += 12 * (size_t)cin;
        const float* pw = p;   // [1][1][cin][cout]
        p += (size_t)cin * cout;
        const BnFold fp = fold_bn(p, p + cout, p + 2 * cout, cout);
        p += 3 * (size_t)cout;
        off_dw_w[l] = reserve(9 * (size_t)cin);
        off_dw_b[l] = reserve(cin);
        off_pw_w[l] = reserve((size_t)cin * cout);
        off_pw_b[l] = reserve(cout);
        for (int t = 0; t < 9; ++t)
            for (int i = 0; i < cin; ++i)
                host[off_dw_w[l] + (size_t)t * cin + i] = (float)((double)dw[(size_t)t * cin + i] * fd.scale[i]);
        for (int i = 0; i < cin; ++i) host[off_dw_b[l] + i] = (float)fd.shift[i];
        for (int n = 0; n < cout; ++n)
            for (int k = 0; k < cin; ++k)
                host[off_pw_w[l] + (size_t)n * cin + k] = (float)((double)pw[(size_t)k * cout + n] * fp.scale[n]);
        for (int n = 0; n < cout; ++n) host[off_pw_b[l] + n] = (float)fp.shift[n];
        // split-f16 copy of the folded pointwise kernel.  Every output channel (row of W^T) is first multiplied by the
        // power of two that puts its largest magnitude into [2^12, 2^13): then hi = f16(w) and lo = f16(w - hi) are both
        // NORMAL f16 numbers for every weight within 2^-15 of the row's largest, i.e. w = hi + lo to 22 bits whatever the
        // scale BatchNorm folding left the row at (unscaled, lo is subnormal - absolute quantum 2^-24 - below |w| = 0.125).
        // The epilogue multiplies the accumulator by the inverse power of two (SepLayer::pw_u): nothing is rounded.
        const size_t nw = (size_t)cin * cout;
        off_pw_hi[l] = reserve((nw + 1) / 2);
        off_pw_lo[l] = reserve((nw + 1) / 2);
        row_exps[l].assign(cout, 0);
        {
            _Float16* hi = reinterpret_cast<_Float16*>(host.data() + off_pw_hi[l]);
            _Float16* lo = reinterpret_cast<_Float16*>(host.data() + off_pw_lo[l]);
            const float* wf = host.data() + off_pw_w[l];
            for (int n = 0; n < cout; ++n) {
                float wmax = 0.0f;
                for (int k = 0; k < cin; ++k) {
                    const float a = std::fabs(wf[(size_t)n * cin + k]);
                    if (!(a <= 3.0e38f)) return fail(BD_EWEIGHTS, "bd_create: non-finite value after BatchNorm folding");
                    wmax = a > wmax ? a : wmax;
                }
                int ex = 0;
                if (wmax > 0.0f) {
                    (void)std::frexp(wmax, &ex);                  // wmax = m 2^ex, 0.5 <= m < 1
                    ex = 13 - ex;
                    ex = ex > 60 ? 60 : (ex < -60 ? -60 : ex);
                }
                row_exps[l][n] = ex;
                for (int k = 0; k < cin; ++k) {
                    const float w = std::ldexp(wf[(size_t)n * cin + k], ex);
                    const _Float16 h = (_Float16)w;
                    hi[(size_t)n * cin + k] = h;
                    lo[(size_t)n * cin + k] = (_Float16)(w - (float)h);
                }
            }
        }
        off_dw16[l] = reserve(10 * (size_t)cin);
        off_pw_u[l] = reserve(cout);
        // the same halves in MFMA fragment order (v_mfma_f32_32x32x16_f16 B operand): for a 32-channel tile t
        // and a 16-deep k step q, lane l holds W[32 t + l % 32][16 q + 8 (l / 32) .. + 8], so a wave's fragment
        // load is 1 KiB contiguous:  frag[((t * K/16 + q) * 64 + l) * 8 + e]
        off_pw_fhi[l] = reserve((nw + 1) / 2);
        off_pw_flo[l] = reserve((nw + 1) / 2);
        {
            const _Float16* hi = reinterpret_cast<const _Float16*>(host.data() + off_pw_hi[l]);
            const _Float16* lo = reinterpret_cast<const _Float16*>(host.data() + off_pw_lo[l]);
            _Float16* fhi = reinterpret_cast<_Float16*>(host.data() + off_pw_fhi[l]);
            _Float16* flo = reinterpret_cast<_Float16*>(host.data() + off_pw_flo[l]);
            const int kq = cin / 16;
            for (int n = 0; n < cout; ++n)
                for (int k = 0; k < cin; ++k) {
                    const int lane = (n & 31) + 32 * ((k & 15) >> 3);
                    const size_t dst = (((size_t)(n >> 5) * kq + (k >> 4)) * 64 + lane) * 8 + (k & 7);
                    fhi[dst] = hi[(size_t)n * cin + k];
                    flo[dst] = lo[(size_t)n * cin + k];
                }
        }
        // the f32 kernel in the fragment order of v_mfma_f32_32x32x2_f32 as the exact-f32 on-chip run reads it (sepchipf32.hip):
        // for a 32-channel tile t and a super-step S of 8 k, lane l holds W[32 t + l % 32][8 S + 4 (l / 32) .. + 4]
        off_pw_ffrag[l] = 0;
        if (cin >= 128) {                                          // layers 5-14: what the exact-f32 on-chip runs and the tail kernel cover (12 MB)
            off_pw_ffrag[l] = reserve(nw);
            const float* wf = host.data() + off_pw_w[l];
            float* ff = host.data() + off_pw_ffrag[l];
            const int ks = cin / 8;
            for (int n = 0; n < cout; ++n)
                for (int k = 0; k < cin; ++k) {
                    const int lane = (n & 31) + 32 * ((k & 7) >> 2);
                    ff[(((size_t)(n >> 5) * ks + (k >> 3)) * 64 + lane) * 4 + (k & 3)] = wf[(size_t)n * cin + k];
                }
        }
        cin = cout;
    }
    if (p - w->embedder_blob != BD_EMBEDDER_BLOB_FLOATS) return fail(BD_EWEIGHTS, "internal: blob walk mismatch");
    size_t off_head_w = 0, off_head_b = 0;
    if (w->n_classes > 0) {
        off_head_w = reserve((size_t)w->n_classes * BD_EMBEDDING_SIZE);
        off_head_b = reserve(BD_MAX_CLASSES);
        for (int c = 0; c < w->n_classes; ++c)
            for (int k = 0; k < BD_EMBEDDING_SIZE; ++k)
                host[off_head_w + (size_t)c * BD_EMBEDDING_SIZE + k] = w->head_kernel[(size_t)k * w->n_classes + c];
        for (int c = 0; c < w->n_classes; ++c) host[off_head_b + c] = w->head_bias[c];
    }
    {
        bool finite = true;
        for (size_t i = off_conv1_w; i < off_conv1_b + 32; ++i) finite = finite && std::isfinite(host[i]);
        int c = kLayerDefs[0][1];
        for (int l = 0; l < 13; ++l) {
            const int co = kLayerDefs[l + 1][1];
            for (size_t i = 0; i < 9 * (size_t)c; ++i) finite = finite && std::isfinite(host[off_dw_w[l] + i]);
            for (int i = 0; i < c; ++i) finite = finite && std::isfinite(host[off_dw_b[l] + i]);
            for (size_t i = 0; i < (size_t)c * co; ++i) finite = finite && std::isfinite(host[off_pw_w[l] + i]);
            for (int i = 0; i < co; ++i) finite = finite && std::isfinite(host[off_pw_b[l] + i]);
            c = co;
        }
        if (!finite) return fail(BD_EWEIGHTS, "bd_create: non-finite value after BatchNorm folding");
    }

    bd::FeTables tables;
    int rc = build_tables(w->mel, &tables);
    if (rc < 0) return rc;

    bd_engine* e = new bd_engine();
    e->device = device;
    e->n_classes = w->n_classes;
    hipError_t err = hipMalloc(&e->d_pool, host.size() * sizeof(float));
    if (err == hipSuccess) err = hipMalloc(&e->d_tables, sizeof(bd::FeTables));
    if (err == hipSuccess) err = hipMalloc(&e->d_range_flag, 256);
    if (err == hipSuccess) err = hipMemset(e->d_range_flag, 0, 256);
    if (err == hipSuccess) err = hipMalloc(&e->d_amax, 64);
    if (err == hipSuccess) err = hipMemset(e->d_amax, 0, 64);
    if (err == hipSuccess) err = hipMemcpy(e->d_pool, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(e->d_tables, &tables, sizeof(tables), hipMemcpyHostToDevice);
    if (err != hipSuccess) {
        if (e->d_pool) (void)hipFree(e->d_pool);
        if (e->d_tables) (void)hipFree(e->d_tables);
        if (e->d_range_flag) (void)hipFree(e->d_range_flag);
        if (e->d_amax) (void)hipFree(e->d_amax);
        delete e;
        return fail(BD_EHIP, std::string("bd_create: ") + hipGetErrorString(err));
    }
    e->conv1_w = e->d_pool + off_conv1_w;
    e->conv1_b = e->d_pool + off_conv1_b;
    int h = 48, wd = 32;
    cin = kLayerDefs[0][1];
    for (int l = 0; l < 13; ++l) {
        bd::SepLayer& L = e->sep[l];
        L.cin = cin;
        L.cout = kLayerDefs[l + 1][1];
        L.stride = kLayerDefs[l + 1][0];
        L.h_in = h;
        L.w_in = wd;
        L.h_out = (h + L.stride - 1) / L.stride;
        L.w_out = (wd + L.stride - 1) / L.stride;
        L.dw_w = e->d_pool + off_dw_w[l];
        L.dw_b = e->d_pool + off_dw_b[l];
        L.pw_wt = e->d_pool + off_pw_w[l];
        L.pw_b = e->d_pool + off_pw_b[l];
        L.pw_variant = 0;
        L.pw_whi = e->d_pool + off_pw_hi[l];
        L.pw_wlo = e->d_pool + off_pw_lo[l];
        L.pw_fhi = e->d_pool + off_pw_fhi[l];
        L.pw_flo = e->d_pool + off_pw_flo[l];
        L.pw_ffrag = off_pw_ffrag[l] ? e->d_pool + off_pw_ffrag[l] : nullptr;
        L.pw_variant16 = 0;
        L.pw_mode = e->pointwise_mode;
        L.range_flag = e->d_range_flag;
        L.dw_w16 = e->d_pool + off_dw16[l];
        L.dw_b16 = L.dw_w16 + 9 * (size_t)cin;
        L.pw_u = e->d_pool + off_pw_u[l];
        L.act_exp = 0;
        L.amax = nullptr;
        e->off_dw16[l] = off_dw16[l];
        e->off_pw_u[l] = off_pw_u[l];
        e->row_exp[l] = row_exps[l];
        e->h_dw[l].assign(host.begin() + off_dw_w[l], host.begin() + off_dw_w[l] + 9 * (size_t)cin);
        e->h_dw[l].insert(e->h_dw[l].end(), host.begin() + off_dw_b[l], host.begin() + off_dw_b[l] + cin);
        h = L.h_out;
        wd = L.w_out;
        cin = L.cout;
    }
    if (w->n_classes > 0) {
        e->head_wt = e->d_pool + off_head_w;
        e->head_b = e->d_pool + off_head_b;
    }
    // activation scales of the f16 modes from a pass over the built-in calibration signal in exact-f32 arithmetic
    rc = apply_scales(e);
    if (rc == BD_OK) rc = calibrate_builtin(e);
    if (rc < 0) {
        const std::string msg = g_error;
        (void)bd_destroy(e);
        return fail(rc, msg);
    }
    *out = e;
    return BD_OK;
}

int bd_destroy(bd_handle h) {
    if (!h) return BD_OK;
    (void)hipSetDevice(h->device);
    for (auto& ev : h->pending) (void)hipEventDestroy(ev.ev);
    for (auto& ev : h->free_events) (void)hipEventDestroy(ev.ev);
    if (h->d_pool) (void)hipFree(h->d_pool);
    if (h->d_tables) (void)hipFree(h->d_tables);
    if (h->d_range_flag) (void)hipFree(h->d_range_flag);
    for (auto& t : h->taps) {
        (void)hipFree(t.dev);
        if (t.uploaded) (void)hipEventDestroy(t.uploaded);
    }
    if (h->d_amax) (void)hipFree(h->d_amax);
    delete h;
    return BD_OK;
}

int bd_set_group_windows(bd_handle h, int32_t windows) {
    if (!h || windows < 0) return fail(BD_EINVAL, "bd_set_group_windows: bad argument");
    h->group_windows = windows == 0 ? kDefaultGroup : windows;
    return BD_OK;
}

int64_t bd_workspace_bytes(bd_handle h, int64_t n_samples, int32_t hop_samples, int32_t patch_step) {
    if (!h) return fail(BD_EINVAL, "bd_workspace_bytes: null handle");
    if (patch_step <= 0) return fail(BD_EINVAL, "patch_step must be > 0");
    Geometry g;
    const int rc = geometry(n_samples, hop_samples, patch_step, &g);
    if (rc < 0) return rc;
    const int64_t group = g.n_windows < h->group_windows ? g.n_windows : h->group_windows;
    return align_up(g.n_frames * BD_MEL_BANDS * 4, 256) + align_up(group * kFloatsA * 4, 256) +
           align_up(group * kFloatsB * 4, 256) + 256;
}

int bd_frontend(bd_handle h, const float* pcm_dev, int64_t n_samples, int32_t hop_samples, float* logmel_dev,
                void* stream) {
    if (!h || !logmel_dev || (!pcm_dev && n_samples > 0)) return fail(BD_EINVAL, "bd_frontend: null argument");
    if (misaligned(pcm_dev) || misaligned(logmel_dev)) return fail(BD_EINVAL, "bd_frontend: pointers need 16-byte alignment");
    Geometry g;
    const int rc = geometry(n_samples, hop_samples, 0, &g);
    if (rc < 0) return rc;
    BD_HIP(hipSetDevice(h->device));
    if (h->profiling) Scope::mark(h, (hipStream_t)stream, -1);
    {
        Scope sc(h, (hipStream_t)stream, 0);
        bd::launch_logmel(pcm_dev, n_samples, g.n_frames, logmel_dev, h->d_tables, (hipStream_t)stream);
    }
    BD_HIP(hipGetLastError());
    return BD_OK;
}

int64_t bd_resample_length(int64_t n_in, int32_t rate_in, int32_t rate_out) {
    if (n_in < 0 || rate_in <= 0 || rate_out <= 0) return fail(BD_EINVAL, "bd_resample_length: bad argument");
    int up, down;
    rational_ratio(rate_in, rate_out, &up, &down);
    return (n_in * up + down - 1) / down;          // ceil(n_in * up / down), as resample_poly
}

int bd_resample_taps(int32_t rate_in, int32_t rate_out, int32_t quality, float* taps_host, int64_t capacity, int32_t* up,
                     int32_t* down, int32_t* half) {
    if (rate_in <= 0 || rate_out <= 0 || !up || !down || !half) return fail(BD_EINVAL, "bd_resample_taps: bad argument");
    if (quality != BD_RESAMPLE_SCIPY && quality != BD_RESAMPLE_HQ) return fail(BD_EINVAL, "bd_resample_taps: unknown quality");
    int u, d, hl;
    rational_ratio(rate_in, rate_out, &u, &d);
    if (u > 4096 || d > 4096) return fail(BD_EINVAL, "bd_resample_taps: rate ratio does not reduce to <= 4096");
    const std::vector<double> t = design_taps(u, d, quality, &hl);
    *up = u;
    *down = d;
    *half = hl;
    if (taps_host) {
        if (capacity < (int64_t)t.size()) return fail(BD_EINVAL, "bd_resample_taps: buffer too small");
        for (size_t i = 0; i < t.size(); ++i) taps_host[i] = (float)t[i];
    }
    return (int)t.size();
}

int bd_debug_fir_plan(int32_t rate_in, int32_t rate_out, int32_t* geometry, int32_t* boff, int64_t boff_capacity,
                      uint16_t* gfrag, int64_t gfrag_capacity) {
    if (rate_in <= 0 || rate_out <= 0 || !geometry) return fail(BD_EINVAL, "bd_debug_fir_plan: bad argument");
    int up, down, half;
    rational_ratio(rate_in, rate_out, &up, &down);
    if (up > 4096 || down > 4096) return fail(BD_EINVAL, "bd_debug_fir_plan: rate ratio does not reduce to <= 4096");
    if (up == 1 && down == 1) return 0;
    const std::vector<double> hd = design_taps(up, down, BD_RESAMPLE_HQ, &half);
    bd::FirPlanHost f;
    if (!bd::fir_plan_build(up, down, hd.data(), half, &f)) return 0;
    const bd::FirPlan& p = f.plan;
    const int32_t g[12] = {p.up, p.down, p.P, p.D, p.NB, p.kq, p.mt, p.contiguous, p.RS, p.a_bytes, p.lds_bytes, half};
    std::memcpy(geometry, g, sizeof(g));
    std::memcpy(geometry + 12, p.unscale, sizeof(p.unscale));
    if (boff) {
        if (boff_capacity < (int64_t)f.boff.size()) return fail(BD_EINVAL, "bd_debug_fir_plan: boff buffer too small");
        std::memcpy(boff, f.boff.data(), f.boff.size() * sizeof(int));
    }
    if (gfrag) {
        if (gfrag_capacity < (int64_t)f.gfrag.size()) return fail(BD_EINVAL, "bd_debug_fir_plan: gfrag buffer too small");
        std::memcpy(gfrag, f.gfrag.data(), f.gfrag.size() * sizeof(uint16_t));
    }
    return 1;
}

int bd_resample_supported(int32_t rate_in, int32_t rate_out, int32_t quality) {
    if (rate_in <= 0 || rate_out <= 0) return fail(BD_EINVAL, "bd_resample_supported: bad rate");
    if (quality != BD_RESAMPLE_SCIPY && quality != BD_RESAMPLE_HQ) return fail(BD_EINVAL, "bd_resample_supported: unknown quality");
    int up, down, half = 0;
    rational_ratio(rate_in, rate_out, &up, &down);
    if (up > 4096 || down > 4096) return 0;
    if (up == 1 && down == 1) return 1;
    const std::vector<double> hd = design_taps(up, down, quality, &half);
    bd::FirPlanHost f;
    if (quality == BD_RESAMPLE_HQ && bd::fir_plan_build(up, down, hd.data(), half, &f)) return 1;
    if (up == 1 && (down == 2 || down == 3) && half == 10 * down) return 1;
    return bd::resample_span_fits(half, up, down) ? 1 : 0;
}

int bd_set_resample_quality(bd_handle h, int32_t quality) {
    if (!h) return fail(BD_EINVAL, "bd_set_resample_quality: null handle");
    if (quality != BD_RESAMPLE_SCIPY && quality != BD_RESAMPLE_HQ) return fail(BD_EINVAL, "bd_set_resample_quality: unknown quality");
    h->resample_quality = quality;
    return BD_OK;
}

static int resample_any(bd_handle h, const void* in_dev, bool s16, int64_t n_in, int32_t channels, int32_t rate_in,
                        int32_t rate_out, float* out_dev, void* stream) {
    if (!h || !out_dev || (!in_dev && n_in > 0)) return fail(BD_EINVAL, "bd_resample: null argument");
    if (n_in < 0 || channels <= 0 || rate_in <= 0 || rate_out <= 0) return fail(BD_EINVAL, "bd_resample: bad size");
    if (misaligned(in_dev) || misaligned(out_dev)) return fail(BD_EINVAL, "bd_resample: pointers need 16-byte alignment");
    int up, down;
    rational_ratio(rate_in, rate_out, &up, &down);
    if (up > 4096 || down > 4096) return fail(BD_EINVAL, "bd_resample: rate ratio does not reduce to <= 4096");
    BD_HIP(hipSetDevice(h->device));
    const int quality = h->resample_quality;
    const bd_engine::Taps* filt = nullptr;
    if (up != 1 || down != 1) {                            // (up = down = 1: no filter, the kernel only converts)
        for (const auto& t : h->taps)
            if (t.up == up && t.down == down && t.quality == quality) filt = &t;
        if (!filt) {                                       // first use of this ratio: design, upload in stream order
            if (h->taps.size() >= 64) return fail(BD_EINVAL, "bd_resample: more than 64 distinct rate ratios on one engine");
            if (h->taps.capacity() < 64) h->taps.reserve(64);      // entries never move: uploads read from them
            bd_engine::Taps t;
            t.up = up;
            t.down = down;
            t.quality = quality;
            const std::vector<double> hd = design_taps(up, down, quality, &t.half);
            // rounds 1-3's filter keeps its kernels (decimate_kernel / resample_kernel: bit for bit what it was); the long
            // filter runs on the matrix cores where the ratio fits
            t.has_plan = quality == BD_RESAMPLE_HQ && bd::fir_plan_build(up, down, hd.data(), t.half, &t.fir);
            if (!t.has_plan && !(up == 1 && (down == 2 || down == 3) && t.half == 10 * down) &&
                !bd::resample_span_fits(t.half, up, down))
                return fail(BD_EINVAL, "bd_resample: the filter of this rate ratio is longer than the kernel's staged span");
            const size_t n_taps = (hd.size() + 3) / 4 * 4;
            const size_t n_g = t.has_plan ? t.fir.gfrag.size() / 2 : 0;             // in floats
            const size_t n_k = t.has_plan ? (t.fir.koff.size() + 3) / 4 * 4 : 0;
            const size_t n_b = t.has_plan ? (t.fir.boff.size() + 3) / 4 * 4 : 0;
            t.host.assign(n_taps + n_g + n_k + n_b, 0.0f);
            for (size_t i = 0; i < hd.size(); ++i) t.host[i] = (float)hd[i];
            if (t.has_plan) {
                std::memcpy(t.host.data() + n_taps, t.fir.gfrag.data(), t.fir.gfrag.size() * sizeof(uint16_t));
                std::memcpy(t.host.data() + n_taps + n_g, t.fir.koff.data(), t.fir.koff.size() * sizeof(int));
                std::memcpy(t.host.data() + n_taps + n_g + n_k, t.fir.boff.data(), t.fir.boff.size() * sizeof(int));
                std::vector<uint16_t>().swap(t.fir.gfrag);
            }
            t.dev = nullptr;
            t.upload_stream = (hipStream_t)stream;
            t.uploaded = nullptr;
            BD_HIP(hipMalloc(&t.dev, t.host.size() * sizeof(float)));
            if (t.has_plan) {
                t.fir.plan.gfrag = t.dev + n_taps;
                t.fir.plan.koff = reinterpret_cast<const int*>(t.dev + n_taps + n_g);
                t.fir.plan.boff = reinterpret_cast<const int*>(t.dev + n_taps + n_g + n_k);
            }
            if (hipEventCreateWithFlags(&t.uploaded, hipEventDisableTiming) != hipSuccess) {
                (void)hipFree(t.dev);
                return fail(BD_EHIP, "bd_resample: hipEventCreate failed");
            }
            h->taps.push_back(std::move(t));
            filt = &h->taps.back();
            BD_HIP(hipMemcpyAsync(filt->dev, filt->host.data(), filt->host.size() * sizeof(float), hipMemcpyHostToDevice,
                                  (hipStream_t)stream));
            BD_HIP(hipEventRecord(filt->uploaded, (hipStream_t)stream));
        } else if (filt->upload_stream != (hipStream_t)stream) {
            // the taps were uploaded in the order of another stream: nothing else orders this launch behind that copy
            BD_HIP(hipStreamWaitEvent((hipStream_t)stream, filt->uploaded, 0));
        }
    }
    const int64_t n_out = (n_in * up + down - 1) / down;
    if (filt && filt->has_plan)
        bd::launch_fir_mfma(in_dev, s16, n_in, channels, filt->fir.plan, out_dev, n_out, (hipStream_t)stream);
    else
        bd::launch_resample(in_dev, s16, n_in, channels, filt ? filt->dev : nullptr, filt ? filt->half : 0, up, down, out_dev,
                            n_out, (hipStream_t)stream);
    BD_HIP(hipGetLastError());
    return BD_OK;
}

int bd_resample(bd_handle h, const float* in_dev, int64_t n_in, int32_t channels, int32_t rate_in, int32_t rate_out,
                float* out_dev, void* stream) {
    return resample_any(h, in_dev, false, n_in, channels, rate_in, rate_out, out_dev, stream);
}

int bd_resample_s16(bd_handle h, const int16_t* in_dev, int64_t n_in, int32_t channels, int32_t rate_in,
                    int32_t rate_out, float* out_dev, void* stream) {
    return resample_any(h, in_dev, true, n_in, channels, rate_in, rate_out, out_dev, stream);
}

int bd_patches(bd_handle h, const float* logmel_dev, int64_t n_frames, int32_t patch_step, float* patches_dev,
               void* stream) {
    if (!h || !logmel_dev || !patches_dev) return fail(BD_EINVAL, "bd_patches: null argument");
    if (patch_step <= 0 || n_frames < 0) return fail(BD_EINVAL, "bd_patches: bad size");
    if (misaligned(logmel_dev) || misaligned(patches_dev)) return fail(BD_EINVAL, "bd_patches: pointers need 16-byte alignment");
    const int64_t w = n_frames >= BD_PATCH_FRAMES ? 1 + (n_frames - BD_PATCH_FRAMES) / patch_step : 0;
    BD_HIP(hipSetDevice(h->device));
    bd::launch_patches(logmel_dev, w, patch_step, patches_dev, (hipStream_t)stream);
    BD_HIP(hipGetLastError());
    return BD_OK;
}

}  // extern "C"

namespace {

// Geometry of a batch of chunks processed in one launch set (each chunk keeps its own zero padding).
struct BatchPlan {
    bd::WindowMap map;
    int64_t sample_base[bd::kMaxBatchChunks];
    int64_t frames[bd::kMaxBatchChunks];
    int64_t total_frames, total_windows;
};

int plan_batch(const int64_t* chunk_samples, int32_t n_chunks, int32_t hop, int32_t step, BatchPlan* p) {
    if (!chunk_samples || n_chunks <= 0 || n_chunks > bd::kMaxBatchChunks)
        return fail(BD_EINVAL, "batch must hold 1..64 chunks");
    if (step <= 0) return fail(BD_EINVAL, "patch_step must be > 0");
    p->map.n_chunks = n_chunks;
    int64_t samples = 0, frames = 0, windows = 0;
    for (int c = 0; c < n_chunks; ++c) {
        Geometry g;
        const int rc = geometry(chunk_samples[c], hop, step, &g);
        if (rc < 0) return rc;
        p->sample_base[c] = samples;
        p->frames[c] = g.n_frames;
        p->map.win_start[c] = (int)windows;
        p->map.frame_base[c] = (int)frames;
        samples += chunk_samples[c];
        frames += g.n_frames;
        windows += g.n_windows;
        if (frames > (1LL << 30)) return fail(BD_ERANGE, "batch too large");
    }
    p->map.win_start[n_chunks] = (int)windows;
    p->total_frames = frames;
    p->total_windows = windows;
    return BD_OK;
}

int64_t batch_workspace(const bd_engine* e, const BatchPlan& p) {
    const int64_t group = p.total_windows < e->group_windows ? p.total_windows : e->group_windows;
    return align_up(p.total_frames * BD_MEL_BANDS * 4, 256) + align_up(group * kFloatsA * 4, 256) +
           align_up(group * kFloatsB * 4, 256) + 256;
}

// The launch plan of one batch of chunks.  stop_stage < 0: run everything; otherwise stop after that CNN
// stage of the first group and copy it to tap_out.
// chunk_pcm[c]: device pointer of chunk c (4-byte aligned; the packed entry points pass pcm + the samples before it).
// mode: arithmetic of the 1x1 convolutions for THIS call (-1: the handle's).  range_word: the word this call's kernels
// raise instead of the engine's sticky one (device memory, or pinned host memory - the kernels then write across PCIe,
// which only ever happens when a chunk does leave the range), or null.  calibrating: the depthwise kernels report their
// largest output into e->d_amax (exact-f32, one kernel per op).
int run_chunks(bd_engine* e, const float* const* chunk_pcm, const int64_t* chunk_samples, int32_t n_chunks, int32_t hop,
               int32_t step, void* ws, int64_t ws_bytes, float* emb, float* logits, int stop_stage, int tap_windows,
               float* tap_out, int mode_arg, int32_t* range_word, bool calibrating, hipStream_t stream) {
    if (!e) return fail(BD_EINVAL, "null handle");
    if (misaligned(ws) || misaligned(emb) || misaligned(logits) || misaligned(tap_out))
        return fail(BD_EINVAL, "device pointers need 16-byte alignment");
    if (mode_arg < -1 || mode_arg > 2) return fail(BD_EINVAL, "mode must be -1 (the handle's), 0, 1 or 2");
    if (logits && e->n_classes == 0) return fail(BD_EINVAL, "engine was created without a head");
    if (!chunk_pcm) return fail(BD_EINVAL, "null chunk pointer table");
    BatchPlan plan;
    int rc = plan_batch(chunk_samples, n_chunks, hop, step, &plan);
    if (rc < 0) return rc;
    for (int c = 0; c < n_chunks; ++c) {
        if (!chunk_pcm[c] && chunk_samples[c] > 0) return fail(BD_EINVAL, "null pcm pointer");
        if (reinterpret_cast<uintptr_t>(chunk_pcm[c]) & 3u) return fail(BD_EINVAL, "pcm pointers need 4-byte alignment");
    }
    const int mode = calibrating ? 0 : (mode_arg < 0 ? e->pointwise_mode : mode_arg);
    BD_HIP(hipSetDevice(e->device));
    unsigned* flag = e->d_range_flag;
    if (range_word) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, range_word) != hipSuccess || !attr.devicePointer) {
            (void)hipGetLastError();
            return fail(BD_EINVAL, "range_word must be device memory or pinned (device-mapped) host memory");
        }
        flag = static_cast<unsigned*>(attr.devicePointer);
    }
    bd::SepLayer sep[13];                     // this call's view of the layers: its own mode, range word, calibration words
    for (int l = 0; l < 13; ++l) {
        sep[l] = e->sep[l];
        sep[l].pw_mode = mode;
        sep[l].range_flag = flag;
        sep[l].amax = calibrating ? e->d_amax + l : nullptr;
    }
    const int64_t need = batch_workspace(e, plan);
    if (!ws || ws_bytes < need) return fail(BD_EWORKSPACE, "workspace smaller than bd_workspace_bytes()");
    struct { int64_t n_frames, n_windows; } g = {plan.total_frames, plan.total_windows};
    if (stop_stage >= 0 && (tap_windows <= 0 || tap_windows > g.n_windows || tap_windows > e->group_windows))
        return fail(BD_EINVAL, "bd_stage_tap: windows must be in 1..min(n_windows, group)");
    BD_HIP(hipSetDevice(e->device));

    const int64_t group = g.n_windows < e->group_windows ? g.n_windows : e->group_windows;
    char* base = static_cast<char*>(ws);
    float* logmel = reinterpret_cast<float*>(base);
    float* const buf_a0 = reinterpret_cast<float*>(base + align_up(g.n_frames * BD_MEL_BANDS * 4, 256));
    float* const buf_b0 = reinterpret_cast<float*>(reinterpret_cast<char*>(buf_a0) + align_up(group * kFloatsA * 4, 256));

    if (e->profiling) Scope::mark(e, stream, -1);
    for (int c = 0; c < n_chunks; ++c) {       // one front-end launch per chunk: its padding is its own
        Scope sc(e, stream, 0);
        bd::launch_logmel(chunk_pcm[c], chunk_samples[c], plan.frames[c],
                          logmel + (int64_t)plan.map.frame_base[c] * BD_MEL_BANDS, e->d_tables, stream);
        BD_REPEAT_EXTRA(0)
            bd::launch_logmel(chunk_pcm[c], chunk_samples[c], plan.frames[c],
                              logmel + (int64_t)plan.map.frame_base[c] * BD_MEL_BANDS, e->d_tables, stream);
    }
    const float* const lm = logmel;
    for (int64_t w0 = 0; w0 < g.n_windows; w0 += group) {
        const int gw = stop_stage >= 0 ? tap_windows : (int)(g.n_windows - w0 < group ? g.n_windows - w0 : group);
        // buf_a holds the latest conv/pointwise output, buf_b the scratch side; the fused separable
        // layers swap the two (everything after layer 2 fits the smaller buffer), so start each pass
        // from the sized assignment: A = 98 304 floats/window, B = 49 152
        float* buf_a = buf_a0;
        float* buf_b = buf_b0;
        // layers 1-3 run as one fused kernel (split-f16 mode) unless a test taps inside them
        const bool fuse_stem = e->fuse_stem && mode != 0 && (stop_stage < 0 || stop_stage >= 2);
        const float* last = buf_a;
        int64_t last_floats = 0;
        bool stopped = false;
        int first_layer = 0;
        // (a tap inside layers 1-3 runs them one kernel per op)
        const bool fuse_stem3 = fuse_stem && (stop_stage < 0 || stop_stage >= 4);
        int skip_dw_layer = -1;      // loop index of a layer whose depthwise the previous kernel already applied
        bool f32_l4 = false;         // exact-f32 mode: layer 4 + the depthwise of layer 5 as one kernel behind the f32 stem
        if (fuse_stem3) {
            {
                Scope sc(e, stream, 5);      // timed in the slot of pointwise 3 (slots 1-4 stay empty)
                const bool alt = mode != 0 && (stop_stage < 0 || stop_stage == 4);
                auto stem_launch = [&]() {
                    if (e->stem_reg && alt)
                        bd::launch_stem_reg(lm, step, plan.map, (int)w0, gw, e->conv1_w, e->conv1_b, sep[0], sep[1], buf_a, stream);
                    else
                        bd::launch_stem4(lm, step, plan.map, (int)w0, gw, e->conv1_w, e->conv1_b, sep[0], sep[1], buf_a, stream);
                };
                stem_launch();
                BD_REPEAT_EXTRA(5) stem_launch();
            }
            last = buf_a;
            last_floats = (int64_t)gw * 24 * 16 * 128;
            stopped = stop_stage == 4;
            first_layer = 2;
        } else if (mode == 0 && e->fuse_stem && !calibrating && stop_stage < 0) {
            // exact-f32 mode: layers 1-3 as one kernel on v_mfma_f32_32x32x2_f32 (sepf32.hip), bit-identical to the five
            // kernels it replaces (the calibration pass and the stage taps keep one kernel per op)
            {
                Scope sc(e, stream, 5);
                // (the layer-2 tile handed over in registers, stemregf32.hip; the form of rounds 4-5 with its tiles through LDS
                //  - stem3_f32_kernel / l4_f32_kernel, sepf32.hip - was removed in round 6: stem = 5 runs this one too)
                auto stem_launch = [&]() {
                    bd::launch_stem_reg_f32(lm, step, plan.map, (int)w0, gw, e->conv1_w, e->conv1_b, sep[0], sep[1], buf_a, stream);
                };
                stem_launch();
                BD_REPEAT_EXTRA(5) stem_launch();
            }
            last = buf_a;
            last_floats = (int64_t)gw * 24 * 16 * 128;
            first_layer = 2;
            f32_l4 = e->fuse_sep;
        } else {
            {
                Scope sc(e, stream, 1);
                bd::launch_conv1(lm, step, plan.map, (int)w0, gw, e->conv1_w, e->conv1_b, buf_a, stream);
            }
            last = buf_a;
            last_floats = (int64_t)gw * 48 * 32 * 32;
            stopped = stop_stage == 0;
        }
        bool pooled_done = false;    // the last layer's kernel already produced the pooled embeddings
        for (int l = first_layer; l < 13 && !stopped; ++l) {
            const bd::SepLayer& L = sep[l];
            // pointwise 5 -> layer 6 -> depthwise 7 -> pointwise 7 as ONE launch, a window per tile, tiles on the CU (sepmid.hip):
            // reads the depthwise-5 output the layer-4 kernel left in buf_b, writes the layer-7 output into buf_a; timed in
            // layer 7's pointwise slot
            if (l == 3 && skip_dw_layer == 3 && e->fuse_sep && e->chip_mid && mode != 0 && stop_stage < 0 &&
                bd::launch_separable_mid(buf_b, buf_a, gw, sep[3], sep[4], sep[5], stream)) {
                BD_REPEAT_EXTRA(13) (void)bd::launch_separable_mid(buf_b, buf_a, gw, sep[3], sep[4], sep[5], stream);
                l = 5;
                if (e->profiling) Scope::mark(e, stream, 3 + 2 * l);
                last = buf_a;
                last_floats = (int64_t)gw * sep[5].h_out * sep[5].w_out * sep[5].cout;
                continue;
            }
            // layers 8-12 + the stride-2 depthwise of layer 13 as ONE launch whose tiles stay on the CU (sepchip.hip): reads
            // buf_a, writes only [windows][3][2][512] into buf_b; timed in layer 12's pointwise slot
            if (e->fuse_sep && mode != 0 && stop_stage < 0 && skip_dw_layer != l) {
                // ... and with the tail behind it on septail.hip's kernel, that output leaves as f16 hi / lo planes
                const bool planes = l == 6 && gw <= (1 << 18) && bd::tail_supported(sep[11], sep[12]);   // (2^18 windows: the tail kernel's 32-bit offsets)
                const int ran = bd::launch_separable_run_next_dw(buf_a, buf_b, gw, &sep[l], 13 - l, stream, planes);
                if (ran > 0) {
                    BD_REPEAT_EXTRA(3 + 2 * (l + ran - 1)) (void)bd::launch_separable_run_next_dw(buf_a, buf_b, gw, &sep[l], 13 - l, stream, planes);
                    l += ran - 1;
                    if (e->profiling) Scope::mark(e, stream, 3 + 2 * l);
                    if (planes) {
                        // pointwise 13 + depthwise 14 (planes buf_b -> planes buf_a), pointwise 14 + average pool (-> [windows][1024]),
                        // timed in the two layers' pointwise slots
                        float* pooled = emb ? emb + w0 * BD_EMBEDDING_SIZE : buf_b;
                        (void)bd::launch_tail_pw13_dw14(buf_b, buf_a, gw, sep[11], sep[12], stream);
                        BD_REPEAT_EXTRA(25) (void)bd::launch_tail_pw13_dw14(buf_b, buf_a, gw, sep[11], sep[12], stream);
                        if (e->profiling) Scope::mark(e, stream, 25);
                        (void)bd::launch_tail_pw14_pool(buf_a, pooled, gw, sep[12], stream);
                        BD_REPEAT_EXTRA(27) (void)bd::launch_tail_pw14_pool(buf_a, pooled, gw, sep[12], stream);
                        if (e->profiling) Scope::mark(e, stream, 27);
                        if (logits) {
                            Scope sc(e, stream, 28);
                            bd::launch_head(pooled, gw, e->head_wt, e->head_b, e->n_classes, logits + w0 * e->n_classes, stream);
                            BD_REPEAT_EXTRA(28)
                                bd::launch_head(pooled, gw, e->head_wt, e->head_b, e->n_classes, logits + w0 * e->n_classes, stream);
                        }
                        pooled_done = true;
                        break;
                    }
                    skip_dw_layer = l + 1;
                    last = buf_b;
                    last_floats = (int64_t)gw * sep[l + 1].h_out * sep[l + 1].w_out * sep[l].cout;
                    continue;
                }
            }
            // stride-1 layers: depthwise inside the GEMM (split-f16 mode), unless a test taps the depthwise
            // ... and when the NEXT layer is a stride-2 one, its depthwise is applied in that kernel's epilogue
            // (whole-window tiles): the kernel then writes the next layer's depthwise output into buf_b
            if (e->fuse_sep && mode != 0 && l + 1 < 13 && (stop_stage < 0 || stop_stage >= 2 * (l + 1) + 2) &&
                bd::launch_separable_fused_next_dw(buf_a, buf_b, gw, L, sep[l + 1], stream)) {
                BD_REPEAT_EXTRA(3 + 2 * l) (void)bd::launch_separable_fused_next_dw(buf_a, buf_b, gw, L, sep[l + 1], stream);
                if (e->profiling) Scope::mark(e, stream, 3 + 2 * l);
                skip_dw_layer = l + 1;
                last = buf_b;
                last_floats = (int64_t)gw * sep[l + 1].h_out * sep[l + 1].w_out * L.cout;
                continue;
            }
            // exact-f32 mode: pointwise 5 + layer 6 + layer 7 as ONE launch (sepmidf32.hip): depthwise 5 has been applied by
            // l4_f32_kernel (buf_b); the layer-7 output lands in buf_a like any pointwise output
            if (f32_l4 && l == 3 && skip_dw_layer == 3 && e->chip_mid && stop_stage < 0 &&
                bd::launch_separable_mid_f32(buf_b, buf_a, gw, sep[3], sep[4], sep[5], stream)) {
                BD_REPEAT_EXTRA(13) (void)bd::launch_separable_mid_f32(buf_b, buf_a, gw, sep[3], sep[4], sep[5], stream);
                l = 5;
                if (e->profiling) Scope::mark(e, stream, 3 + 2 * l);
                last = buf_a;
                last_floats = (int64_t)gw * 6 * 4 * 512;
                continue;
            }
            // exact-f32 mode: layers 8-12 + the depthwise of layer 13 as ONE launch whose tiles stay on the CU (sepchipf32.hip).
            // Its input is the layer-7 output (buf_a) or, when the launch in front applied depthwise 8 in its epilogue, that
            // (buf_b); layer 13 then starts at its 1x1 convolution on the depthwise-13 output in buf_b
            if (f32_l4 && l == 6 && stop_stage < 0) {
                const bool dw8_done = skip_dw_layer == 6;
                float* const src = dw8_done ? buf_b : buf_a;
                float* const dst = dw8_done ? buf_a : buf_b;
                if (bd::launch_separable_chip_f32(src, dst, gw, &sep[6], 5, stream, &sep[11], dw8_done)) {
                    BD_REPEAT_EXTRA(23) (void)bd::launch_separable_chip_f32(src, dst, gw, &sep[6], 5, stream, &sep[11], dw8_done);
                    l = 10;
                    if (e->profiling) Scope::mark(e, stream, 3 + 2 * l);
                    if (dw8_done) {
                        float* t = buf_a;
                        buf_a = buf_b;
                        buf_b = t;
                    }
                    skip_dw_layer = 11;
                    last = buf_b;
                    last_floats = (int64_t)gw * 3 * 2 * 512;
                    continue;
                }
            }
            // exact-f32 mode: pointwise 13 + depthwise 14 and pointwise 14 + pool on septail.hip's kernel (depthwise 13 has been applied
            // by the on-chip run: buf_b -> buf_a -> [windows][1024]), timed in the two layers' pointwise slots
            if (f32_l4 && l == 11 && skip_dw_layer == 11 && stop_stage < 0 && gw <= (1 << 17) && bd::tail_f32_supported(sep[11], sep[12])) {
                float* pooled = emb ? emb + w0 * BD_EMBEDDING_SIZE : buf_b;
                (void)bd::launch_tail_f32(buf_b, buf_a, pooled, gw, sep[11], sep[12], stream, 0);
                BD_REPEAT_EXTRA(25) (void)bd::launch_tail_f32(buf_b, buf_a, pooled, gw, sep[11], sep[12], stream, 0);
                if (e->profiling) Scope::mark(e, stream, 25);
                (void)bd::launch_tail_f32(buf_b, buf_a, pooled, gw, sep[11], sep[12], stream, 1);
                BD_REPEAT_EXTRA(27) (void)bd::launch_tail_f32(buf_b, buf_a, pooled, gw, sep[11], sep[12], stream, 1);
                if (e->profiling) Scope::mark(e, stream, 27);
                if (logits) {
                    Scope sc(e, stream, 28);
                    bd::launch_head(pooled, gw, e->head_wt, e->head_b, e->n_classes, logits + w0 * e->n_classes, stream);
                    BD_REPEAT_EXTRA(28)
                        bd::launch_head(pooled, gw, e->head_wt, e->head_b, e->n_classes, logits + w0 * e->n_classes, stream);
                }
                pooled_done = true;
                break;
            }
            // exact-f32 mode, behind the f32 stem: layer 4 and layer 5's stride-2 depthwise as one kernel (bit-identical to the
            // three it replaces); layer 5 then starts at its 1x1 convolution
            // (the layer-4 tile handed to depthwise 5 in registers, l4regf32.hip)
            auto l4_launch = [&]() { return bd::launch_l4_reg_f32(buf_a, buf_b, gw, L, sep[3], stream); };
            if (f32_l4 && l == 2 && l4_launch()) {
                BD_REPEAT_EXTRA(3 + 2 * l) (void)l4_launch();
                if (e->profiling) Scope::mark(e, stream, 3 + 2 * l);
                skip_dw_layer = 3;
                last = buf_b;
                last_floats = (int64_t)gw * 12 * 8 * 128;
                continue;
            }
            if (skip_dw_layer != l) {
                Scope sc(e, stream, 2 + 2 * l);
                bd::launch_depthwise(buf_a, buf_b, gw, L, stream);
            }
            last = buf_b;
            last_floats = (int64_t)gw * L.h_out * L.w_out * L.cin;
            if (stop_stage == 2 * l + 1) {
                stopped = true;
                break;
            }
            // exact-f32 mode behind the f32 stem: the 1x1 convolution with the NEXT layer's depthwise in its epilogue (whole
            // windows per 96-row tile; bit-identical to the two kernels): the 1x1 output never reaches HBM and the next
            // layer starts at its own 1x1 convolution.  Timed in this layer's pointwise slot.
            if (f32_l4 && l + 1 < 13 && bd::launch_pointwise_next_dw_f32(buf_b, buf_a, gw, L, sep[l + 1], stream)) {
                BD_REPEAT_EXTRA(3 + 2 * l) (void)bd::launch_pointwise_next_dw_f32(buf_b, buf_a, gw, L, sep[l + 1], stream);
                if (e->profiling) Scope::mark(e, stream, 3 + 2 * l);
                float* t = buf_a;                // the next layer's depthwise output is what buf_b holds from here on
                buf_a = buf_b;
                buf_b = t;
                skip_dw_layer = l + 1;
                last = buf_b;
                last_floats = (int64_t)gw * sep[l + 1].h_out * sep[l + 1].w_out * L.cout;
                continue;
            }
            {
                Scope sc(e, stream, 3 + 2 * l);
                bd::launch_pointwise(buf_b, buf_a, (int64_t)gw * L.h_out * L.w_out, L, stream);
                BD_REPEAT_EXTRA(3 + 2 * l) bd::launch_pointwise(buf_b, buf_a, (int64_t)gw * L.h_out * L.w_out, L, stream);
            }
            last = buf_a;
            last_floats = (int64_t)gw * L.h_out * L.w_out * L.cout;
            if (stop_stage == 2 * l + 2) stopped = true;
        }
        if (stop_stage >= 0) {
            if (mode != 0 && (stop_stage & 1)) {
                // a depthwise output of the f16 modes carries its layer's power-of-two activation scale: hand out the true values
                bd::launch_scale_copy(last, tap_out, last_floats, std::ldexp(1.0f, -sep[(stop_stage - 1) / 2].act_exp), stream);
            } else {
                BD_HIP(hipMemcpyAsync(tap_out, last, last_floats * sizeof(float), hipMemcpyDeviceToDevice, stream));
            }
            break;
        }
        if (!pooled_done) {
            Scope sc(e, stream, 28);
            bd::launch_pool_head(buf_a, gw, e->head_wt, e->head_b, e->n_classes,
                                 emb ? emb + w0 * BD_EMBEDDING_SIZE : nullptr,
                                 logits ? logits + w0 * e->n_classes : nullptr, stream);
        }
    }
    BD_HIP(hipGetLastError());
    return BD_OK;
}

// chunk pointers of the packed entry points: chunk c starts where chunk c - 1 ended
int run_packed(bd_engine* e, const float* pcm, const int64_t* chunk_samples, int32_t n_chunks, int32_t hop, int32_t step,
               void* ws, int64_t ws_bytes, float* emb, float* logits, int stop_stage, int tap_windows, float* tap_out,
               hipStream_t stream) {
    if (!chunk_samples || n_chunks <= 0 || n_chunks > bd::kMaxBatchChunks) return fail(BD_EINVAL, "batch must hold 1..64 chunks");
    if (misaligned(pcm)) return fail(BD_EINVAL, "device pointers need 16-byte alignment");
    const float* ptrs[bd::kMaxBatchChunks];
    int64_t at = 0;
    for (int c = 0; c < n_chunks; ++c) {
        if (chunk_samples[c] < 0) return fail(BD_EINVAL, "n_samples must be >= 0");
        if (!pcm && chunk_samples[c] > 0) return fail(BD_EINVAL, "null pcm pointer");
        ptrs[c] = pcm ? pcm + at : nullptr;
        at += chunk_samples[c];
    }
    return run_chunks(e, ptrs, chunk_samples, n_chunks, hop, step, ws, ws_bytes, emb, logits, stop_stage, tap_windows, tap_out,
                      -1, nullptr, false, stream);
}

// ---- operand scaling of the f16 modes (SepLayer in bd_internal.h) ----

// Exponent s with amax * 2^s in [2^8, 2^9): a factor 128 below the f16 overflow for inputs beyond the calibration set,
// and hi + lo both normal f16 numbers (22-bit operands) for every activation down to 2^-11 of the layer's largest.
int exponent_for(float amax) {
    if (!(amax > 0.0f) || !(amax <= 3.0e38f)) return 0;
    int ex = 0;
    (void)std::frexp(amax, &ex);
    const int s = 9 - ex;
    return s > 60 ? 60 : (s < -60 ? -60 : s);
}

// Rebuilds the scaled depthwise copies and the epilogue factors from e->act_exp.  Blocking copies into the pool: only
// call it while no work that uses the handle is in flight (bd_create, bd_calibrate, bd_set_activation_exponents).
int apply_scales(bd_engine* e) {
    BD_HIP(hipSetDevice(e->device));
    std::vector<float> buf;
    for (int l = 0; l < 13; ++l) {
        bd::SepLayer& L = e->sep[l];
        const int s = e->act_exp[l];
        buf.resize(e->h_dw[l].size());
        for (size_t i = 0; i < buf.size(); ++i) {
            buf[i] = std::ldexp(e->h_dw[l][i], s);
            if (!std::isfinite(buf[i])) return fail(BD_EWEIGHTS, "activation scale overflows a depthwise weight");
        }
        BD_HIP(hipMemcpy(e->d_pool + e->off_dw16[l], buf.data(), buf.size() * sizeof(float), hipMemcpyHostToDevice));
        buf.resize(L.cout);
        for (int n = 0; n < L.cout; ++n) buf[n] = std::ldexp(1.0f, -(s + e->row_exp[l][n]));
        BD_HIP(hipMemcpy(e->d_pool + e->off_pw_u[l], buf.data(), buf.size() * sizeof(float), hipMemcpyHostToDevice));
        L.act_exp = s;
    }
    return BD_OK;
}

// One exact-f32 pass (one kernel per op) over the given chunks with the depthwise kernels reporting their largest
// output; the per-layer maxima only ever grow, the exponents follow them.  Waits for the stream.
int calibrate_on(bd_engine* e, const float* const* chunk_pcm, const int64_t* chunk_samples, int32_t n_chunks, int32_t hop,
                 int32_t step, void* ws, int64_t ws_bytes, hipStream_t stream) {
    BD_HIP(hipSetDevice(e->device));
    BD_HIP(hipMemsetAsync(e->d_amax, 0, 64, stream));
    const int rc = run_chunks(e, chunk_pcm, chunk_samples, n_chunks, hop, step, ws, ws_bytes, nullptr, nullptr, -1, 0, nullptr,
                              0, nullptr, true, stream);
    if (rc < 0) return rc;
    unsigned words[16] = {0};
    BD_HIP(hipMemcpyAsync(words, e->d_amax, 64, hipMemcpyDeviceToHost, stream));
    BD_HIP(hipStreamSynchronize(stream));
    for (int l = 0; l < 13; ++l) {
        float m;
        std::memcpy(&m, &words[l], sizeof(m));
        if (m > e->act_max[l] && m <= 3.0e38f) e->act_max[l] = m;
        e->act_exp[l] = exponent_for(e->act_max[l]);
    }
    return apply_scales(e);
}

// The signal bd_create calibrates on: sixteen 0.96 s windows that span what 16 kHz PCM in [-1, 1] can do to the log-mel
// input (silence = the log(0.001) floor everywhere, full-scale noise = the ceiling everywhere, tones, clicks, a chirp,
// a buzz-like square wave).  The CNN's input is a LOG spectrum, so a recording can leave this envelope only by a small
// factor; the exponents keep a factor 128 in hand and the range word catches whatever goes beyond.
std::vector<float> calibration_signal() {
    const int seg = 15360, nseg = 16;
    std::vector<float> x((size_t)seg * nseg + 240, 0.0f);
    uint32_t lcg = 20260723u;
    auto uni = [&]() {                                     // uniform in [-1, 1)
        lcg = lcg * 1664525u + 1013904223u;
        return (float)((int32_t)lcg) * (1.0f / 2147483648.0f);
    };
    const double w = 2.0 * M_PI / 16000.0;
    for (int g = 0; g < nseg; ++g)
        for (int i = 0; i < seg; ++i) {
            const double t = i;
            double v = 0.0;
            switch (g) {
                case 0: v = 0.0; break;
                case 1: v = uni(); break;
                case 2: v = 0.1 * uni(); break;
                case 3: v = 0.01 * uni(); break;
                case 4: v = 0.001 * uni(); break;
                case 5: v = 0.9 * std::sin(w * 220.0 * t); break;
                case 6: v = 0.5 * std::sin(w * 1000.0 * t) + 0.05 * uni(); break;
                case 7: v = 0.9 * std::sin(w * 4000.0 * t); break;
                case 8: v = 0.7 * std::sin(w * (100.0 + 6900.0 * t / (2.0 * seg)) * t); break;
                case 9: v = i % 1000 == 0 ? 1.0 : 0.0; break;
                case 10: v = std::sin(w * 300.0 * t) >= 0.0 ? 1.0 : -1.0; break;
                case 11: v = 0.3 * uni() * (0.5 + 0.5 * std::sin(w * 200.0 * t)); break;
                case 12: v = 0.5 + 0.0001 * uni(); break;
                case 13: v = 0.8 * std::sin(w * 50.0 * t) + 0.2 * std::sin(w * 150.0 * t); break;
                case 14: v = uni() >= 0.0f ? 1.0 : -1.0; break;
                default: v = uni() * t / seg; break;
            }
            v = v > 1.0 ? 1.0 : (v < -1.0 ? -1.0 : v);
            x[(size_t)g * seg + i] = (float)v;
        }
    return x;
}

int calibrate_builtin(bd_engine* e) {
    const std::vector<float> sig = calibration_signal();
    const int64_t n = (int64_t)sig.size();
    const int32_t hop = 15360, step = 96;
    const int64_t ws_bytes = bd_workspace_bytes(e, n, hop, step);
    if (ws_bytes < 0) return (int)ws_bytes;
    float* d_pcm = nullptr;
    void* d_ws = nullptr;
    hipError_t err = hipMalloc(&d_pcm, sig.size() * sizeof(float));
    if (err == hipSuccess) err = hipMalloc(&d_ws, (size_t)ws_bytes);
    if (err == hipSuccess) err = hipMemcpy(d_pcm, sig.data(), sig.size() * sizeof(float), hipMemcpyHostToDevice);
    int rc = BD_OK;
    if (err != hipSuccess) {
        rc = fail(BD_EHIP, std::string("bd_create (calibration): ") + hipGetErrorString(err));
    } else {
        const float* ptr = d_pcm;
        rc = calibrate_on(e, &ptr, &n, 1, hop, step, d_ws, ws_bytes, nullptr);
    }
    if (d_pcm) (void)hipFree(d_pcm);
    if (d_ws) (void)hipFree(d_ws);
    return rc;
}

}  // namespace

extern "C" {

int bd_embed(bd_handle h, const float* pcm_dev, int64_t n_samples, int32_t hop_samples, int32_t patch_step,
             void* workspace_dev, int64_t workspace_bytes, float* emb_dev, void* stream) {
    if (!emb_dev) return fail(BD_EINVAL, "bd_embed: null output");
    return run_packed(h, pcm_dev, &n_samples, 1, hop_samples, patch_step, workspace_dev, workspace_bytes, emb_dev,
                      nullptr, -1, 0, nullptr, (hipStream_t)stream);
}

int bd_predict(bd_handle h, const float* pcm_dev, int64_t n_samples, int32_t hop_samples, int32_t patch_step,
               void* workspace_dev, int64_t workspace_bytes, float* emb_dev, float* logits_dev, void* stream) {
    if (!logits_dev) return fail(BD_EINVAL, "bd_predict: null output");
    return run_packed(h, pcm_dev, &n_samples, 1, hop_samples, patch_step, workspace_dev, workspace_bytes, emb_dev,
                      logits_dev, -1, 0, nullptr, (hipStream_t)stream);
}

int64_t bd_batch_num_windows(const int64_t* chunk_samples, int32_t n_chunks, int32_t hop_samples, int32_t patch_step,
                             int64_t* per_chunk_windows) {
    BatchPlan plan;
    const int rc = plan_batch(chunk_samples, n_chunks, hop_samples, patch_step, &plan);
    if (rc < 0) return rc;
    if (per_chunk_windows)
        for (int c = 0; c < n_chunks; ++c) per_chunk_windows[c] = plan.map.win_start[c + 1] - plan.map.win_start[c];
    return plan.total_windows;
}

int64_t bd_batch_workspace_bytes(bd_handle h, const int64_t* chunk_samples, int32_t n_chunks, int32_t hop_samples,
                                 int32_t patch_step) {
    if (!h) return fail(BD_EINVAL, "bd_batch_workspace_bytes: null handle");
    BatchPlan plan;
    const int rc = plan_batch(chunk_samples, n_chunks, hop_samples, patch_step, &plan);
    if (rc < 0) return rc;
    return batch_workspace(h, plan);
}

int bd_predict_batch(bd_handle h, const float* pcm_dev, const int64_t* chunk_samples, int32_t n_chunks,
                     int32_t hop_samples, int32_t patch_step, void* workspace_dev, int64_t workspace_bytes,
                     float* emb_dev, float* logits_dev, void* stream) {
    if (!logits_dev && !emb_dev) return fail(BD_EINVAL, "bd_predict_batch: no output requested");
    return run_packed(h, pcm_dev, chunk_samples, n_chunks, hop_samples, patch_step, workspace_dev, workspace_bytes,
                      emb_dev, logits_dev, -1, 0, nullptr, (hipStream_t)stream);
}

int bd_predict_chunks(bd_handle h, const float* const* chunk_pcm_dev, const int64_t* chunk_samples, int32_t n_chunks,
                      int32_t hop_samples, int32_t patch_step, void* workspace_dev, int64_t workspace_bytes, float* emb_dev,
                      float* logits_dev, int32_t mode, int32_t* range_word, void* stream) {
    if (!logits_dev && !emb_dev) return fail(BD_EINVAL, "bd_predict_chunks: no output requested");
    if (!chunk_pcm_dev || !chunk_samples || n_chunks <= 0 || n_chunks > bd::kMaxBatchChunks)
        return fail(BD_EINVAL, "batch must hold 1..64 chunks");
    return run_chunks(h, chunk_pcm_dev, chunk_samples, n_chunks, hop_samples, patch_step, workspace_dev, workspace_bytes,
                      emb_dev, logits_dev, -1, 0, nullptr, mode, range_word, false, (hipStream_t)stream);
}

int bd_calibrate(bd_handle h, const float* pcm_dev, int64_t n_samples, int32_t hop_samples, int32_t patch_step,
                 void* workspace_dev, int64_t workspace_bytes, void* stream) {
    if (!h) return fail(BD_EINVAL, "bd_calibrate: null handle");
    if (!pcm_dev && n_samples > 0) return fail(BD_EINVAL, "bd_calibrate: null pcm pointer");
    if (misaligned(pcm_dev)) return fail(BD_EINVAL, "device pointers need 16-byte alignment");
    return calibrate_on(h, &pcm_dev, &n_samples, 1, hop_samples, patch_step, workspace_dev, workspace_bytes,
                        (hipStream_t)stream);
}

int bd_get_scales(bd_handle h, int32_t* act_exp, float* act_max) {
    if (!h) return fail(BD_EINVAL, "bd_get_scales: null handle");
    for (int l = 0; l < 13; ++l) {
        if (act_exp) act_exp[l] = h->act_exp[l];
        if (act_max) act_max[l] = h->act_max[l];
    }
    return 13;
}

int bd_set_activation_exponents(bd_handle h, const int32_t* act_exp) {
    if (!h || !act_exp) return fail(BD_EINVAL, "bd_set_activation_exponents: null argument");
    for (int l = 0; l < 13; ++l)
        if (act_exp[l] < -60 || act_exp[l] > 60) return fail(BD_EINVAL, "bd_set_activation_exponents: exponent outside -60..60");
    for (int l = 0; l < 13; ++l) h->act_exp[l] = act_exp[l];
    return apply_scales(h);
}

int bd_stage_tap(bd_handle h, const float* pcm_dev, int64_t n_samples, int32_t hop_samples, int32_t patch_step,
                 void* workspace_dev, int64_t workspace_bytes, int32_t stage, int32_t windows, float* out_dev,
                 void* stream) {
    if (!out_dev) return fail(BD_EINVAL, "bd_stage_tap: null output");
    if (stage < 0 || stage >= BD_NUM_STAGES) return fail(BD_EINVAL, "bd_stage_tap: stage out of range");
    return run_packed(h, pcm_dev, &n_samples, 1, hop_samples, patch_step, workspace_dev, workspace_bytes, nullptr,
                      nullptr, stage, windows, out_dev, (hipStream_t)stream);
}

int bd_debug_pointwise(const float* a_dev, const float* wt_dev, const float* bias_dev, float* c_dev, int64_t m,
                       int32_t n, int32_t k, int32_t variant, void* stream) {
    if (!a_dev || !wt_dev || !bias_dev || !c_dev || m < 0) return fail(BD_EINVAL, "bd_debug_pointwise: bad argument");
    if (bd::launch_pointwise_variant(a_dev, wt_dev, bias_dev, c_dev, m, n, k, variant, (hipStream_t)stream) != 0)
        return fail(BD_EINVAL, "bd_debug_pointwise: shape/variant not supported (K % 32, N % tile)");
    BD_HIP(hipGetLastError());
    return BD_OK;
}

int bd_set_pointwise_variant(bd_handle h, int32_t layer, int32_t variant) {
    if (!h || layer < 2 || layer > 14) return fail(BD_EINVAL, "bd_set_pointwise_variant: layer must be 2..14");
    if (h->sep[layer - 2].pw_mode != 0) h->sep[layer - 2].pw_variant16 = variant;
    else h->sep[layer - 2].pw_variant = variant;
    return BD_OK;
}

int bd_set_fusion(bd_handle h, int32_t stem, int32_t separable) {
    if (!h) return fail(BD_EINVAL, "null handle");
    // (removed in round 6 and refused like any unknown code: stem 2 = layers 1-2 + depthwise 3 only, 4 = the walk of stemroll.hip;
    //  separable 2 = layer 4 as band tiles, 3 = one launch per layer for layers 8-11, 4 / 5 = layer 12 / 14 on the 8-wave kernel,
    //  6 = one exact-f32 kernel per separable layer, 7 = the round-3 run of layers 8-11 through global memory + layer 12 / 14 on the
    //  12-wave kernel, 8 = the on-chip run ending at layer 11, 9 / 12 = plain fused layers)
    if (stem != 0 && stem != 3 && stem != 5) return fail(BD_EINVAL, "bd_set_fusion: stem must be 0, 3 or 5");
    if (separable != 0 && separable != 1 && separable != 10)
        return fail(BD_EINVAL, "bd_set_fusion: separable must be 0, 1 or 10");
    h->fuse_stem = stem != 0;
    h->stem_reg = stem == 3;                 // 3 (default): the layer-2 tile handed over in registers (stemreg.hip); 5: through LDS, a
                                             //    workgroup per row block (stem3_kernel, the default until round 5)
    h->fuse_sep = separable != 0;
    h->chip_mid = separable == 1;            // 10: layers 5-7 on the four kernels of round 4
    return BD_OK;
}

int bd_range_flag(bd_handle h, int32_t* flag_host, int32_t reset, void* stream) {
    if (!h || !flag_host) return fail(BD_EINVAL, "bd_range_flag: null argument");
    BD_HIP(hipSetDevice(h->device));
    unsigned v = 0;
    BD_HIP(hipMemcpyAsync(&v, h->d_range_flag, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream));
    BD_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (reset && v) BD_HIP(hipMemsetAsync(h->d_range_flag, 0, sizeof(v), (hipStream_t)stream));
    *flag_host = (int32_t)v;
    return BD_OK;
}

int bd_range_flag_copy(bd_handle h, int32_t* dst, int32_t reset, void* stream) {
    if (!h || !dst) return fail(BD_EINVAL, "bd_range_flag_copy: null argument");
    BD_HIP(hipSetDevice(h->device));
    BD_HIP(hipMemcpyAsync(dst, h->d_range_flag, sizeof(int32_t), hipMemcpyDefault, (hipStream_t)stream));
    if (reset) BD_HIP(hipMemsetAsync(h->d_range_flag, 0, sizeof(int32_t), (hipStream_t)stream));
    return BD_OK;
}

int bd_set_pointwise_mode(bd_handle h, int32_t mode) {
    if (!h || mode < 0 || mode > 2)
        return fail(BD_EINVAL, "bd_set_pointwise_mode: mode must be 0 (f32), 1 (split f16) or 2 (plain f16)");
    h->pointwise_mode = mode;
    for (auto& L : h->sep) L.pw_mode = mode;
    return BD_OK;
}

int bd_debug_pointwise_f16x3(const float* a_dev, const void* whi_dev, const void* wlo_dev, const float* unscale_dev,
                             const float* bias_dev, float* c_dev, int64_t m, int32_t n, int32_t k, int32_t variant,
                             void* stream) {
    if (!a_dev || !whi_dev || !wlo_dev || !unscale_dev || !bias_dev || !c_dev || m < 0)
        return fail(BD_EINVAL, "bd_debug_pointwise_f16x3: bad argument");
    if (bd::launch_pointwise_f16x3_variant(a_dev, whi_dev, wlo_dev, unscale_dev, bias_dev, c_dev, m, n, k, variant,
                                           (hipStream_t)stream) != 0)
        return fail(BD_EINVAL, "bd_debug_pointwise_f16x3: shape/variant not supported");
    BD_HIP(hipGetLastError());
    return BD_OK;
}

int bd_profile_enable(bd_handle h, int32_t on) {
    if (!h) return fail(BD_EINVAL, "null handle");
    h->profiling = on != 0;
    return BD_OK;
}

int bd_profile_read(bd_handle h, double* ms, int64_t* launches, int32_t slots) {
    if (!h || !ms || !launches || slots < BD_PROFILE_SLOTS) return fail(BD_EINVAL, "bd_profile_read: bad argument");
    BD_HIP(hipSetDevice(h->device));
    for (size_t i = 0; i < h->pending.size(); ++i) {
        const Event2& ev = h->pending[i];
        BD_HIP(hipEventSynchronize(ev.ev));
        if (ev.slot >= 0 && i > 0) {
            float t = 0.f;
            BD_HIP(hipEventElapsedTime(&t, h->pending[i - 1].ev, ev.ev));
            h->ms[ev.slot] += t;
            h->launches[ev.slot] += 1;
        }
    }
    for (auto& ev : h->pending) h->free_events.push_back(ev);
    h->pending.clear();
    for (int i = 0; i < BD_PROFILE_SLOTS; ++i) {
        ms[i] = h->ms[i];
        launches[i] = h->launches[i];
        h->ms[i] = 0;
        h->launches[i] = 0;
    }
    return BD_PROFILE_SLOTS;
}

// ---- the streamer's way onto the device: pread -> small pinned buffers -> hipMemcpyAsync (header: bd_stager_*) ----
struct bd_stager {
    int device = 0, n = 0, turn = 0;
    int64_t bytes = 0;
    char* base = nullptr;
    hipEvent_t busy[4] = {nullptr, nullptr, nullptr, nullptr};
    bool used[4] = {false, false, false, false};
};

int bd_stager_create(bd_stager_t* out, int32_t device, int64_t stage_bytes, int32_t n_stage) {
    if (!out || stage_bytes < 4096 || stage_bytes % 4096 || n_stage < 1 || n_stage > 4)
        return fail(BD_EINVAL, "bd_stager_create: stage_bytes must be a multiple of 4096, n_stage 1..4");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count)
        return fail(BD_ENODEVICE, "bd_stager_create: no such HIP device (there is no CPU path)");
    BD_HIP(hipSetDevice(device));
    std::unique_ptr<bd_stager> st(new bd_stager);
    st->device = device;
    st->n = n_stage;
    st->bytes = stage_bytes;
    BD_HIP(hipHostMalloc(reinterpret_cast<void**>(&st->base), (size_t)(stage_bytes * n_stage), hipHostMallocDefault));
    for (int i = 0; i < n_stage; ++i) {
        if (hipEventCreateWithFlags(&st->busy[i], hipEventDisableTiming) != hipSuccess) {
            bd_stager_destroy(st.release());
            return fail(BD_EHIP, "bd_stager_create: hipEventCreate failed");
        }
    }
    *out = st.release();
    return BD_OK;
}

int bd_stager_destroy(bd_stager_t st) {
    if (!st) return BD_OK;
    (void)hipSetDevice(st->device);
    for (int i = 0; i < 4; ++i) {
        if (!st->busy[i]) continue;
        if (st->used[i]) (void)hipEventSynchronize(st->busy[i]);
        (void)hipEventDestroy(st->busy[i]);
    }
    if (st->base) (void)hipHostFree(st->base);
    delete st;
    return BD_OK;
}

int bd_stager_acquire(bd_stager_t st, int32_t* index, void** host) {
    if (!st || !index || !host) return fail(BD_EINVAL, "bd_stager_acquire: null argument");
    const int i = st->turn;
    st->turn = (i + 1) % st->n;
    if (st->used[i]) BD_HIP(hipEventSynchronize(st->busy[i]));      // the copy that last read this buffer
    *index = i;
    *host = st->base + (size_t)i * st->bytes;
    return BD_OK;
}

int bd_stager_submit(bd_stager_t st, int32_t index, int64_t nbytes, void* dev, void* stream) {
    if (!st || index < 0 || index >= st->n || nbytes < 0 || nbytes > st->bytes || (!dev && nbytes))
        return fail(BD_EINVAL, "bd_stager_submit: bad argument");
    if (nbytes == 0) return BD_OK;
    BD_HIP(hipSetDevice(st->device));
    BD_HIP(hipMemcpyAsync(dev, st->base + (size_t)index * st->bytes, (size_t)nbytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    BD_HIP(hipEventRecord(st->busy[index], (hipStream_t)stream));
    st->used[index] = true;
    return BD_OK;
}

int64_t bd_stager_read(bd_stager_t st, int32_t fd, int64_t offset, int64_t nbytes, void* dev, void* stream) {
    if (!st || fd < 0 || offset < 0 || nbytes < 0 || (!dev && nbytes)) return fail(BD_EINVAL, "bd_stager_read: bad argument");
    int64_t done = 0;
    while (done < nbytes) {
        int32_t i = 0;
        void* host = nullptr;
        if (bd_stager_acquire(st, &i, &host) != BD_OK) return BD_EHIP;
        const int64_t want = nbytes - done < st->bytes ? nbytes - done : st->bytes;
        int64_t got = 0;
        while (got < want) {
            const ssize_t r = pread(fd, static_cast<char*>(host) + got, (size_t)(want - got), (off_t)(offset + done + got));
            if (r <= 0) break;                                      // the end of the file (or an error: what was read stands)
            got += r;
        }
        if (bd_stager_submit(st, i, got, static_cast<char*>(dev) + done, stream) != BD_OK) return BD_EHIP;
        done += got;
        if (got < want) break;
    }
    return done;
}

}  // extern "C"
