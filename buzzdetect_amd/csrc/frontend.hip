// Front end of the analyze hot path on gfx950: PCM -> log-mel spectrogram, one kernel.
//
// Restates embedders/yamnet/features.py:22-58 (tf.signal.stft 400/160/512 -> tf.abs ->
// matmul with the [257,64] mel matrix -> log(x + 0.001)) with pad_waveform (features.py:82-108)
// folded in as "samples past n_valid read as zero".  Nothing of [T,400] / [T,257] is ever
// written to HBM: a workgroup stages a run of PCM into LDS, each 64-lane wavefront owns one
// STFT frame at a time and carries it through
//   Hann window -> 256-point complex FFT of the even/odd-packed frame (radix-4 Stockham,
//   4 passes through the wave's own LDS tile) -> real-FFT split -> |X[k]| -> banded mel
//   reduction (lane m owns band m; every band is a short run of bins) -> logf
// and writes one coalesced 256-byte row of the [T,64] output.
//
// Algorithmic HBM traffic: 640 B read (160 new samples) + 256 B written per frame.
#include "bd_internal.h"
#include <cstdio>
#include <cstdlib>

namespace bd {

namespace {

constexpr int kWaves = 4;
constexpr int kFramesPerWave = 4;
constexpr int kGroupFrames = kWaves * kFramesPerWave;                         // 16 frames per pass
constexpr int kGroupSamples = (kGroupFrames - 1) * BD_STFT_HOP + BD_STFT_WINDOW;  // 2800
constexpr int kMelTaps = kMelMaxLen;   // mel weights a lane keeps in registers (bd_create refuses longer bands; YAMNet: 17)
static_assert(kMelTaps % 6 == 0, "the mel loop runs in batches of six");

// Each wavefront works on its own z / mag tile, and LDS executes one wave's DS instructions in order,
// so passes of a frame only need their LDS traffic drained and the compiler kept from reordering
// across the point - not a workgroup barrier.
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// Bank swizzle of the 256-entry float2 FFT tile.  The Stockham passes write with strides 4, 16 and 64 (index =
// 4 l + r, 16 g + k + 4 r, 64 q + k + 16 r) and read contiguously; unswizzled, the strided ds_write_b64 are
// 4-way bank-conflicted per half-wave and the kernel is LDS-bound.  XOR-ing the slot's low five bits with two
// parities of index bits 5..7 makes every write and every contiguous read conflict-free (searched exhaustively
// over the GF(2) maps of those bits; the mirrored read of the real-FFT split keeps a 2-way conflict).
// Because only low bits change, each access pattern is "per-lane base ^ small constant".
__device__ __forceinline__ constexpr int zsw_mask(int b5, int b6, int b7) { return ((b5 ^ b6) * 21) ^ ((b5 ^ b7) * 10); }
__device__ __forceinline__ int zsw(int i) { return i ^ zsw_mask((i >> 5) & 1, (i >> 6) & 1, (i >> 7) & 1); }

// forward DFT-4 of (u0..u3) -> (X0..X3) in place
__device__ __forceinline__ void dft4(float2& u0, float2& u1, float2& u2, float2& u3) {
    const float2 a = make_float2(u0.x + u2.x, u0.y + u2.y);
    const float2 b = make_float2(u0.x - u2.x, u0.y - u2.y);
    const float2 c = make_float2(u1.x + u3.x, u1.y + u3.y);
    const float2 d = make_float2(u1.y - u3.y, -(u1.x - u3.x));   // -i * (u1 - u3)
    u0 = make_float2(a.x + c.x, a.y + c.y);
    u1 = make_float2(b.x + d.x, b.y + d.y);
    u2 = make_float2(a.x - c.x, a.y - c.y);
    u3 = make_float2(b.x - d.x, b.y - d.y);
}

__global__ __launch_bounds__(256, 4) void logmel_kernel(const float* __restrict__ pcm, long long n_valid,
                                                     long long n_frames, float* __restrict__ out,
                                                     const FeTables* __restrict__ tab, unsigned* __restrict__ dbg) {
#define FE_TS(I) if (dbg && blockIdx.x == 3 && threadIdx.x == 0 && group == blockIdx.x && fi == 1) dbg[I] = (unsigned)__builtin_readcyclecounter();
    __shared__ __attribute__((aligned(16))) float s_pcm[kGroupSamples];
    __shared__ __attribute__((aligned(16))) float2 s_z[kWaves][256];
    __shared__ __attribute__((aligned(16))) float s_mag[kWaves][BD_SPECTRUM_BINS + 7];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // Everything a lane needs from the constant tables is fixed for the whole kernel (its butterfly index,
    // its bins, its band), so it lives in registers: per frame the LDS only carries the data itself.
    float2 hann2[4];                      // Hann taps of the lane's four packed input points
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n2 = 2 * (lane + 64 * r);
        hann2[r] = n2 < BD_STFT_WINDOW ? make_float2(tab->hann[n2], tab->hann[n2 + 1]) : make_float2(0.f, 0.f);
    }
    float2 tw[3][3];                      // twiddles of passes p = 4, 16, 64 for inputs 1..3
    {
        int pi = 0;
#pragma unroll
        for (int p = 4; p <= 64; p *= 4, ++pi) {
            const int tstep = (lane & (p - 1)) * (64 / p);
#pragma unroll
            for (int r = 1; r <= 3; ++r) tw[pi][r - 1] = tab->tw256[(r * tstep) & 255];
        }
    }
    float2 tws[4];                        // real-FFT split twiddles exp(-2 pi i k / 512), k = lane + 64 r
#pragma unroll
    for (int r = 0; r < 4; ++r) tws[r] = tab->tw512[lane + 64 * r];
    // swizzled tile indices: contiguous accesses lane + 64 r, pass-1 / p=4 / p=16 write bases, mirrored reads
    const int zi_lin = lane ^ ((lane >> 5) * 31);                    // (zi_lin ^ zsw_mask(0, r & 1, r >> 1)) + 64 r
    const int zi_w1 = zsw(4 * lane);                                 // ^ r
    const int zi_w4 = zsw(((lane & ~3) << 2) + (lane & 3));          // ^ 4 r
    const int zi_w16 = zsw(((lane & ~15) << 2) + (lane & 15));       // ^ 16 r ^ (r >= 2 ? 31 : 0)
#define BD_ZLIN(R) ((zi_lin ^ zsw_mask(0, (R) & 1, (R) >> 1)) + 64 * (R))
    const int band_start = tab->band_start[lane];
    const int band_len = tab->band_len[lane];
    float bw[kMelTaps];                   // the band's mel weights (zero past band_len)
#pragma unroll
    for (int j = 0; j < kMelTaps; ++j) bw[j] = j < band_len ? tab->band_w[j][lane] : 0.0f;

    float2* z = s_z[wave];
    float* mag = s_mag[wave];
    if (lane < 7) mag[BD_SPECTRUM_BINS + lane] = 0.0f;   // padding read by the fixed-length mel loop

    const long long n_groups = (n_frames + kGroupFrames - 1) / kGroupFrames;
    for (long long group = blockIdx.x; group < n_groups; group += gridDim.x) {
        __syncthreads();   // previous pass done with s_pcm
        const long long base = group * (long long)(kGroupFrames * BD_STFT_HOP);
        for (int i = tid; i < kGroupSamples; i += 256) {
            const long long idx = base + i;
            s_pcm[i] = idx < n_valid ? pcm[idx] : 0.0f;
        }
        __syncthreads();

        for (int fi = 0; fi < kFramesPerWave; ++fi) {
            const int fl = wave + kWaves * fi;
            const long long frame = group * kGroupFrames + fl;
            const float* x = s_pcm + fl * BD_STFT_HOP;

            FE_TS(0)
            // ---- pass 1 (p = 1): windowed samples straight from the PCM tile, no twiddles ----
            float2 u[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n2 = 2 * (lane + 64 * r);   // z[n] = x[2n] + i x[2n+1]; zero past 400
                if (n2 < BD_STFT_WINDOW) {
                    const float2 xv = *reinterpret_cast<const float2*>(x + n2);
                    u[r] = make_float2(xv.x * hann2[r].x, xv.y * hann2[r].y);
                } else {
                    u[r] = make_float2(0.0f, 0.0f);
                }
            }
            dft4(u[0], u[1], u[2], u[3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) z[zi_w1 ^ r] = u[r];
            wave_lds_sync();

            FE_TS(1)
            // ---- passes 2..4 (p = 4, 16, 64) ----
            {
                int pi = 0;
#pragma unroll
                for (int p = 4; p <= 64; p *= 4, ++pi) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) u[r] = z[BD_ZLIN(r)];
                    u[1] = cmul(u[1], tw[pi][0]);
                    u[2] = cmul(u[2], tw[pi][1]);
                    u[3] = cmul(u[3], tw[pi][2]);
                    dft4(u[0], u[1], u[2], u[3]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = p == 4 ? (zi_w4 ^ (4 * r)) : p == 16 ? (zi_w16 ^ (16 * r) ^ ((r >> 1) * 31)) : BD_ZLIN(r);
                        z[j] = u[r];
                    }
                    wave_lds_sync();
                }
            }

            FE_TS(2)
            // ---- split the packed transform into the real spectrum, take magnitudes ----
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = lane + 64 * r;
                const float2 zk = z[BD_ZLIN(r)];
                const float2 zm = z[zsw((256 - (lane + 64 * r)) & 255)];
                const float ex = 0.5f * (zk.x + zm.x);
                const float ey = 0.5f * (zk.y - zm.y);
                const float ox = 0.5f * (zk.y + zm.y);    // O = -i/2 * (Zk - conj(Zm))
                const float oy = -0.5f * (zk.x - zm.x);
                const float2 t = tws[r];
                const float xr = ex + (t.x * ox - t.y * oy);
                const float xi = ey + (t.x * oy + t.y * ox);
                // v_sqrt_f32 (1 ulp) instead of the correctly rounded sqrtf: its scale / class-test / select fix-ups were
                // ~56 of a frame's ~245 vector instructions, and |X| only feeds log(mel + 0.001)
                mag[k] = __builtin_amdgcn_sqrtf(xr * xr + xi * xi);
            }
            if (lane == 0) mag[256] = fabsf(z[0].x - z[0].y);
            wave_lds_sync();

            FE_TS(3)
            // ---- banded mel reduction + log ----
            // All kMelTaps reads are in bounds (band_start + 17 <= 257 + 6 of zero padding) and the weight is zero past
            // band_len, where fmaf(m, 0, acc) returns acc exactly (m finite, acc >= +0): so no guards - twenty
            // independent LDS reads in flight, then twenty FMAs.  With a uniform and a per-lane branch around every tap
            // the compiler serialised read -> wait -> fma twenty times (1640 of a frame's 4570 cycles).
            float acc = 0.0f;
#pragma unroll
            for (int j0 = 0; j0 < kMelTaps; j0 += 6) {        // three batches of six: the register budget is 128
                float mv[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) mv[j] = mag[band_start + j0 + j];
#pragma unroll
                for (int j = 0; j < 6; ++j) acc = fmaf(mv[j], bw[j0 + j], acc);
            }
            FE_TS(4)
            if (frame < n_frames) out[frame * BD_MEL_BANDS + lane] = logf(acc + 0.001f);
            wave_lds_sync();   // mag / z reads of this frame retire before the next frame overwrites them
            FE_TS(5)
        }
    }
#undef BD_ZLIN
#undef FE_TS
}

// ---------------------------------------------------------------------------------------------------
// Second formulation of the same transform: 256 = 16 x 16.  Sixteen lanes own one frame (four frames per
// wavefront), lane n2 holds the 16 packed points z[16 n1 + n2] in registers:
//   DFT-16 over n1 in registers -> twiddle W256^(n2 k1) -> ONE transpose through LDS -> DFT-16 over n2 in
//   registers -> Z[k1 + 16 k2] -> natural order in LDS -> real split against Z[256 - k] -> |X| -> LDS ->
//   banded mel with lane = band (as above), one frame after the other -> logf.
// Three wave-level sync points per four frames instead of six per frame; same arithmetic definition
// (results differ from logmel_kernel only by float summation order).
__device__ __forceinline__ float2 c_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 c_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 c_mi(float2 a) { return make_float2(a.y, -a.x); }   // a * (-i)

__device__ __forceinline__ void radix4(float2 u0, float2 u1, float2 u2, float2 u3, float2& y0, float2& y1,
                                       float2& y2, float2& y3) {
    const float2 s0 = c_add(u0, u2), s1 = c_sub(u0, u2), s2 = c_add(u1, u3), s3 = c_mi(c_sub(u1, u3));
    y0 = c_add(s0, s2);
    y1 = c_add(s1, s3);
    y2 = c_sub(s0, s2);
    y3 = c_sub(s1, s3);
}

// in-place forward DFT of 16 points, natural order in and out (two radix-4 stages in registers)
__device__ __forceinline__ void dft16(float2 (&x)[16]) {
    constexpr float C = 0.92387953251128674f, S = 0.38268343236508977f, H = 0.70710678118654752f;
    float2 a[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) radix4(x[q], x[4 + q], x[8 + q], x[12 + q], a[q][0], a[q][1], a[q][2], a[q][3]);
    a[1][1] = cmul(a[1][1], make_float2(C, -S));     // W16^1
    a[1][2] = cmul(a[1][2], make_float2(H, -H));     // W16^2
    a[1][3] = cmul(a[1][3], make_float2(S, -C));     // W16^3
    a[2][1] = cmul(a[2][1], make_float2(H, -H));     // W16^2
    a[2][2] = c_mi(a[2][2]);                         // W16^4 = -i
    a[2][3] = cmul(a[2][3], make_float2(-H, -H));    // W16^6
    a[3][1] = cmul(a[3][1], make_float2(S, -C));     // W16^3
    a[3][2] = cmul(a[3][2], make_float2(-H, -H));    // W16^6
    a[3][3] = cmul(a[3][3], make_float2(-C, S));     // W16^9
#pragma unroll
    for (int r = 0; r < 4; ++r) radix4(a[0][r], a[1][r], a[2][r], a[3][r], x[r], x[r + 4], x[r + 8], x[r + 12]);
}

__global__ __launch_bounds__(256) void logmel16_kernel(const float* __restrict__ pcm, long long n_valid,
                                                       long long n_frames, float* __restrict__ out,
                                                       const FeTables* __restrict__ tab) {
    constexpr int TZ = 17 * 16;                            // transposed tile: rows of 17 (bank padding)
    __shared__ __attribute__((aligned(16))) float s_pcm[kGroupSamples];
    __shared__ __attribute__((aligned(16))) float2 s_t[kWaves][4][TZ];      // transpose, then Z, then |X|
    __shared__ __attribute__((aligned(16))) float2 s_tw512[256];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int fq = lane >> 4;                              // frame within the wave
    const int n2 = lane & 15;                              // also k1 after the transpose

    s_tw512[tid] = tab->tw512[tid];
    float2 hann2[13];                                      // Hann taps of z[16 n1 + n2], n1 = 0..12
#pragma unroll
    for (int n1 = 0; n1 < 13; ++n1) {
        const int e = 2 * (16 * n1 + n2);
        hann2[n1] = e < BD_STFT_WINDOW ? make_float2(tab->hann[e], tab->hann[e + 1]) : make_float2(0.f, 0.f);
    }
    float2 tw[15];                                         // W256^(n2 k1), k1 = 1..15
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) tw[k1 - 1] = tab->tw256[(n2 * k1) & 255];
    const int band_start = tab->band_start[lane];
    const int band_len = tab->band_len[lane];
    float bw[kMelTaps];
#pragma unroll
    for (int j = 0; j < kMelTaps; ++j) bw[j] = j < band_len ? tab->band_w[j][lane] : 0.0f;
    const int max_len = tab->max_len;

    float2* const t_mine = s_t[wave][fq];
    float* const mag_wave = reinterpret_cast<float*>(s_t[wave]);            // [4][2 * TZ] floats, |X| at [f][k]

    const long long n_groups = (n_frames + kGroupFrames - 1) / kGroupFrames;
    for (long long group = blockIdx.x; group < n_groups; group += gridDim.x) {
        __syncthreads();
        const long long base = group * (long long)(kGroupFrames * BD_STFT_HOP);
        for (int i = tid; i < kGroupSamples; i += 256) {
            const long long idx = base + i;
            s_pcm[i] = idx < n_valid ? pcm[idx] : 0.0f;
        }
        __syncthreads();

        const int fl = wave * 4 + fq;                                       // frame of this 16-lane group
        const float* x = s_pcm + fl * BD_STFT_HOP;
        float2 u[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            if (n1 < 13) {
                const int e = 2 * (16 * n1 + n2);
                float2 xv = make_float2(0.f, 0.f);
                if (e < BD_STFT_WINDOW) xv = *reinterpret_cast<const float2*>(x + e);
                u[n1] = make_float2(xv.x * hann2[n1].x, xv.y * hann2[n1].y);
            } else {
                u[n1] = make_float2(0.f, 0.f);
            }
        }
        dft16(u);                                                           // over n1 -> k1
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) u[k1] = cmul(u[k1], tw[k1 - 1]);
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) t_mine[k1 * 17 + n2] = u[k1];
        wave_lds_sync();
#pragma unroll
        for (int m2 = 0; m2 < 16; ++m2) u[m2] = t_mine[n2 * 17 + m2];       // lane k1 = n2 gathers over n2
        wave_lds_sync();                                                    // tile is rewritten below
        dft16(u);                                                           // over n2 -> k2: u[k2] = Z[k1 + 16 k2]
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) t_mine[n2 + 16 * k2] = u[k2];       // natural order
        wave_lds_sync();
        float mg[16];
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            const int k = n2 + 16 * k2;
            const float2 zk = u[k2];
            const float2 zm = t_mine[(256 - k) & 255];
            const float ex = 0.5f * (zk.x + zm.x);
            const float ey = 0.5f * (zk.y - zm.y);
            const float ox = 0.5f * (zk.y + zm.y);
            const float oy = -0.5f * (zk.x - zm.x);
            const float2 t = s_tw512[k];
            const float xr = ex + (t.x * ox - t.y * oy);
            const float xi = ey + (t.x * oy + t.y * ox);
            mg[k2] = sqrtf(xr * xr + xi * xi);
        }
        const float nyq = fabsf(u[0].x - u[0].y);                           // meaningful in lane k1 = 0 only
        wave_lds_sync();                                                    // Z reads done: overwrite with |X|
        float* mag = mag_wave + fq * (2 * TZ);
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) mag[n2 + 16 * k2] = mg[k2];
        if (n2 == 0) mag[256] = nyq;
        if (n2 < 7) mag[257 + n2] = 0.0f;                                   // padding read by the fixed-length loop
        wave_lds_sync();

        // ---- banded mel + log: lane = band, the wave's four frames one after the other ----
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const float* mf = mag_wave + f * (2 * TZ);
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < kMelTaps; ++j) {
                if (j < max_len) {
                    const float m = mf[band_start + j];
                    if (j < band_len) acc = fmaf(m, bw[j], acc);
                }
            }
            const long long frame = group * kGroupFrames + wave * 4 + f;
            if (frame < n_frames) out[frame * BD_MEL_BANDS + lane] = logf(acc + 0.001f);
        }
        wave_lds_sync();
    }
}

__global__ __launch_bounds__(256) void patches_kernel(const float* __restrict__ logmel, long long n_windows,
                                                      int patch_step, float* __restrict__ patches) {
    // tf.signal.frame(axis=0) (features.py:72-76): patch w = frames [w*step, w*step + 96)
    const long long per = BD_PATCH_FRAMES * BD_MEL_BANDS / 4;   // float4 per patch
    const long long total = n_windows * per;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
        const long long w = i / per;
        const long long r = i - w * per;
        const float4 v = reinterpret_cast<const float4*>(logmel + w * patch_step * BD_MEL_BANDS)[r];
        reinterpret_cast<float4*>(patches)[i] = v;
    }
}

// Downmix + rational resample (the streamer's np.mean(axis=1) and librosa.resample, src/stream/worker.py:116-128,
// as one device pass).  y[j] = sum_i mono[i] * h[j*down - i*up + half], h = Kaiser(5.0)-windowed sinc of
// 2*half + 1 taps scaled by `up` (the scipy.signal.resample_poly design; the reference's soxr_hq is a
// different low-pass, so this stage is "parity unpinned" against the reference and pinned against its own
// CPU restatement).  One thread per output sample, taps walked in increasing input index.
__device__ __forceinline__ float pcm_to_float(float v) { return v; }
__device__ __forceinline__ float pcm_to_float(short v) { return (float)v * (1.0f / 32768.0f); }   // as libsndfile's float read

template <typename T>
__global__ __launch_bounds__(256) void resample_kernel(const T* __restrict__ in, long long n_in, int channels,
                                                       const float* __restrict__ h, int half, int up, int down,
                                                       float* __restrict__ out, long long n_out) {
    for (long long j = blockIdx.x * 256LL + threadIdx.x; j < n_out; j += gridDim.x * 256LL) {
        const long long c = j * down;                       // position on the up-sampled grid
        long long i0 = (c - half + up - 1) / up;            // ceil((c - half) / up), c - half may be negative
        if (c - half < 0) i0 = -((half - c) / up);
        long long i1 = (c + half) / up;
        if (i0 < 0) i0 = 0;
        if (i1 > n_in - 1) i1 = n_in - 1;
        float acc = 0.0f;
        for (long long i = i0; i <= i1; ++i) {
            float m = 0.0f;
            if (channels == 1) {
                m = pcm_to_float(in[i]);
            } else if (channels == 2) {
                m = (pcm_to_float(in[2 * i]) + pcm_to_float(in[2 * i + 1])) * 0.5f;
            } else {
                for (int ch = 0; ch < channels; ++ch) m += pcm_to_float(in[i * channels + ch]);
                m = m / (float)channels;
            }
            acc = fmaf(m, h[c - i * up + half], acc);
        }
        out[j] = acc;
    }
}

}  // namespace

void launch_resample(const void* in, bool s16, int64_t n_in, int channels, const float* taps, int half, int up,
                     int down, float* out, int64_t n_out, hipStream_t stream) {
    if (n_out <= 0) return;
    const int64_t blocks = (n_out + 255) / 256;
    const int grid = (int)(blocks < 65536 ? blocks : 65536);
    if (s16)
        hipLaunchKernelGGL(resample_kernel<short>, dim3(grid), dim3(256), 0, stream, static_cast<const short*>(in),
                           (long long)n_in, channels, taps, half, up, down, out, (long long)n_out);
    else
        hipLaunchKernelGGL(resample_kernel<float>, dim3(grid), dim3(256), 0, stream, static_cast<const float*>(in),
                           (long long)n_in, channels, taps, half, up, down, out, (long long)n_out);
}

void launch_logmel(const float* pcm, int64_t n_valid, int64_t n_frames, float* logmel,
                   const FeTables* tables, hipStream_t stream, int variant) {
    if (n_frames <= 0) return;
    const int64_t groups = (n_frames + kGroupFrames - 1) / kGroupFrames;
    const int grid = (int)(groups < 4096 ? groups : 4096);
    if (variant == 1) {
        hipLaunchKernelGGL(logmel16_kernel, dim3(grid), dim3(256), 0, stream, pcm, (long long)n_valid,
                           (long long)n_frames, logmel, tables);
        return;
    }
    static unsigned* dbg = nullptr;           // developer aid: BD_FE_TRACE=1 prints a per-frame phase trace
    static int shots = 0;
    if (!dbg && getenv("BD_FE_TRACE")) (void)hipMalloc(&dbg, 64);
    hipLaunchKernelGGL(logmel_kernel, dim3(grid), dim3(256), 0, stream, pcm, (long long)n_valid,
                       (long long)n_frames, logmel, tables, dbg);
    if (dbg) {
        (void)hipStreamSynchronize(stream);
        unsigned h[8];
        (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        if (++shots == 4)
            fprintf(stderr, "[trace] front end, one frame of one wave (4 waves/SIMD), cycles: pass 1 %u | passes 2-4 %u | split + magnitude %u | mel %u | log + store %u | total %u\n",
                    h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[5] - h[0]);
    }
}

void launch_patches(const float* logmel, int64_t n_windows, int patch_step, float* patches,
                    hipStream_t stream) {
    if (n_windows <= 0) return;
    const int64_t total = n_windows * (BD_PATCH_FRAMES * BD_MEL_BANDS / 4);
    const int64_t blocks = (total + 255) / 256;
    const int grid = (int)(blocks < 8192 ? blocks : 8192);
    hipLaunchKernelGGL(patches_kernel, dim3(grid), dim3(256), 0, stream, logmel, (long long)n_windows,
                       patch_step, patches);
}

}  // namespace bd
