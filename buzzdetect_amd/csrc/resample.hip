// soxr_hq-class downmix + resample on the matrix cores of gfx950 (the streamer's np.mean(axis=1) + librosa.resample,
// src/stream/worker.py:116-128, with librosa's default res_type soxr_hq: linear phase, flat to 0.9136 of the lower Nyquist,
// >= 120 dB from it on - see oracle/resample_oracle.py for the design and what is and is not pinned).
//
// Such a low-pass is ~570 taps per output for 48 -> 16 kHz (the 61-tap filter of rounds 1-3 sat in scalar registers;
// this one is 9 x that): 570 f32 FMAs per 16 input bytes is 70 FLOP per byte, three times past the ridge of the vector
// units (157 TFLOP/s over 8 TB/s), so as vector code the stage would be compute bound at ~180 us per 1024-window batch.
// It is a dense contraction, though.  With P = 32 NB outputs and D input samples per period of the rate ratio
// (P down = D up), output j = m P + 32 b + n of period m, phase block b, column n is
//
//     y[m][b][n] = sum_e  x[m D + off_b + e] * G_b[e][n],      G_b[e][n] = h[(32 b + n) down - (off_b + e) up + half]
//
// i.e. for every phase block a product [periods x K] . [K x 32] whose left operand is just the signal read with a row
// stride of D samples (a Toeplitz view: nothing is gathered or copied per row) and whose right operand is a banded
// rearrangement of the filter, K = 31 down / up + taps per phase (85 % of it non-zero).  Carried in split-f16 arithmetic
// like the 1x1 convolutions: x = hi + lo and G = hi + lo as f16 pairs, three v_mfma_f32_32x32x16_f16 per
// product, f32 accumulate.  16-bit PCM is EXACT in that form (a 16-bit integer, or the half-integer mean of two, is
// hi + lo with no remainder), the filter is carried to 22 bits after an exact power-of-two scale; float PCM is carried to
// 22 bits after a scale of 2^6 (|x| beyond 1023 saturates instead of becoming inf).
//
// fir_mfma_kernel<T, KQ, MT>: a 4-wave workgroup (one wave per SIMD, two workgroups per CU) owns ROWS = 32 MT periods of
// one phase block.
//   stage   the input span of the chunk -> channel mean -> (hi, lo) f16 -> LDS, eight samples (one 16-byte chunk of each
//           half) per thread and step, coalesced 16-byte global loads.  With one phase block (integer decimations, 2:3,
//           1:2 ...) the rows overlap and the span is staged once, contiguously; row r of the Toeplitz view then starts
//           D / 8 chunks after row r - 1, an even number, which would put the sixteen rows of a ds_read_b128 lane group
//           on four bank slots - so one spare chunk is skipped every D / 8 chunks and the row stride becomes odd
//           (conflict-free by the bank rule of MI355X_MICROARCH.md, LDS).  With several phase blocks (44.1 kHz: 40) a
//           row is its own piece of the signal, staged at an odd chunk stride.
//   filter  K is split over the four waves: wave w keeps the (hi, lo) B fragments of ITS KQ k-steps in registers for the
//           whole workgroup (8 KQ VGPRs, loaded once from the fragment-ordered copy, one KiB per wave-instruction) -
//           no filter traffic in the loop at all.
//   loop    per k-step and row tile: two ds_read_b128 (A hi, A lo; the k-step's LDS offset is a scalar) and three
//           MFMAs.  lo products first, hi x hi last.
//   reduce  the four partial [32 MT x 32] tiles meet in LDS (the staged signal is dead by then), are added in wave
//           order, scaled by the inverse power of two and stored as 128-byte row segments.
// Bound: the matrix pipe (3 K / 16 MFMAs per 1024 outputs: 29 us per 1024-window batch of 48 kHz input at 2.1 GHz) and
// HBM (16 B read + 4 B written per output for 48 kHz stereo 16-bit: 45 us at 5.6 TB/s) are within a factor of two of
// each other; DESIGN.md has the measured number.
#include "bd_internal.h"

#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

namespace bd {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kFirWaves = 4;
constexpr int kFirThreads = 64 * kFirWaves;

// x, y, z, w -> (hi, lo) f16 halves, hi = f16(v), lo = f16(v - hi): one packed convert + one v_fma_mix per value
__device__ __forceinline__ void fir_split(float x, float y, float z, float w, f16x4& hi, f16x4& lo) {
    const f16x2 h0 = {(_Float16)x, (_Float16)y}, h1 = {(_Float16)z, (_Float16)w};
    f16x2 l0, l1;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(h0), "v"(x));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l0) : "v"(h0), "v"(y));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(h1), "v"(z));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l1) : "v"(h1), "v"(w));
    hi[0] = h0[0]; hi[1] = h0[1]; hi[2] = h1[0]; hi[3] = h1[1];
    lo[0] = l0[0]; lo[1] = l0[1]; lo[2] = l1[0]; lo[3] = l1[1];
}

// the value a sample is staged as: channel mean times 2^15 (16-bit PCM: exact) or 2^6 (float PCM, saturating)
template <typename T> struct FirIn;
template <> struct FirIn<short> {
    static __device__ __forceinline__ float one(short v) { return (float)v; }
};
template <> struct FirIn<float> {
    static __device__ __forceinline__ float one(float v) { return __builtin_amdgcn_fmed3f(v * 64.0f, -65504.0f, 65504.0f); }
};

// Eight consecutive frames starting at frame i0 (a multiple of 8) -> the eight staged values.
// Interior chunks of mono / stereo input, in two steps so that a thread has several chunks' loads in flight before it
// converts the first (nothing between the loads: no branch, no register copy the compiler would have to wait for):
// fir_fetch issues the 16-byte loads, fir_convert turns the raw words into values.
template <typename T, int CH> struct FirRaw {
    uint4 r[sizeof(T) * CH / 2];
};

template <typename T, int CH>
__device__ __forceinline__ FirRaw<T, CH> fir_fetch(const T* __restrict__ in, long long i0) {
    FirRaw<T, CH> raw;
    const uint4* src = reinterpret_cast<const uint4*>(in + CH * i0);
#pragma unroll
    for (int e = 0; e < (int)(sizeof(T) * CH / 2); ++e) raw.r[e] = src[e];
    return raw;
}

template <typename T, int CH>
__device__ __forceinline__ void fir_convert(const FirRaw<T, CH>& raw, float (&v)[8]) {
    if constexpr (sizeof(T) == 2 && CH == 1) {
        const unsigned w[4] = {raw.r[0].x, raw.r[0].y, raw.r[0].z, raw.r[0].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = (float)(short)(w[e] & 0xffffu);
            v[2 * e + 1] = (float)((int)w[e] >> 16);
        }
    } else if constexpr (sizeof(T) == 2) {
        const unsigned w[8] = {raw.r[0].x, raw.r[0].y, raw.r[0].z, raw.r[0].w, raw.r[1].x, raw.r[1].y, raw.r[1].z, raw.r[1].w};
#pragma unroll
        for (int e = 0; e < 8; ++e)              // (l / 32768 + r / 32768) / 2 * 32768 = (l + r) / 2, exact in float32
            v[e] = (float)((int)(short)(w[e] & 0xffffu) + ((int)w[e] >> 16)) * 0.5f;
    } else if constexpr (CH == 1) {
        const unsigned w[8] = {raw.r[0].x, raw.r[0].y, raw.r[0].z, raw.r[0].w, raw.r[1].x, raw.r[1].y, raw.r[1].z, raw.r[1].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = FirIn<float>::one(__uint_as_float(w[e]));
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {            // float32 mean as the reference computes it, then the exact scale
            const uint4 a = raw.r[e];
            v[2 * e] = FirIn<float>::one((__uint_as_float(a.x) + __uint_as_float(a.y)) * 0.5f);
            v[2 * e + 1] = FirIn<float>::one((__uint_as_float(a.z) + __uint_as_float(a.w)) * 0.5f);
        }
    }
}

// the signal's edges (samples outside it are zero) and inputs of three or more channels: one sample at a time
template <typename T>
__device__ __forceinline__ void fir_slow8(const T* __restrict__ in, long long i0, long long n_in, int channels, float (&v)[8]) {
#pragma unroll 1
    for (int e = 0; e < 8; ++e) {
        const long long i = i0 + e;
        float m = 0.0f;
        if (i >= 0 && i < n_in) {
            if (channels == 1) {
                m = FirIn<T>::one(in[i]);
            } else if (channels == 2) {
                if constexpr (sizeof(T) == 2) m = (float)((int)in[2 * i] + (int)in[2 * i + 1]) * 0.5f;
                else m = FirIn<float>::one((in[2 * i] + in[2 * i + 1]) * 0.5f);
            } else {
                float s = 0.0f;
                for (int ch = 0; ch < channels; ++ch) {
                    if constexpr (sizeof(T) == 2) s += (float)in[i * channels + ch] * (1.0f / 32768.0f);
                    else s += in[i * channels + ch];
                }
                s /= (float)channels;
                if constexpr (sizeof(T) == 2) m = s * 32768.0f;
                else m = FirIn<float>::one(s);
            }
        }
        v[e] = m;
    }
}

__device__ __forceinline__ void fir_put8(char* a_hi, char* a_lo, int phys, const float (&v)[8]) {
    f16x4 h0, l0, h1, l1;
    fir_split(v[0], v[1], v[2], v[3], h0, l0);
    fir_split(v[4], v[5], v[6], v[7], h1, l1);
    f16x8 hv, lv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hv[e] = h0[e]; hv[4 + e] = h1[e];
        lv[e] = l0[e]; lv[4 + e] = l1[e];
    }
#if defined(BD_FIR_ABLATE) && BD_FIR_ABLATE == 2      // developer build: no staging writes
    asm volatile("" ::"v"(hv), "v"(lv), "v"(phys));
#else
    *reinterpret_cast<f16x8*>(a_hi + 16 * phys) = hv;
    *reinterpret_cast<f16x8*>(a_lo + 16 * phys) = lv;
#endif
}

// The staged signal of a workgroup: chunk q (eight samples) -> its first frame and its place in LDS.
template <int CPR>
__device__ __forceinline__ void fir_chunk_at(const FirPlan& p, long long base, int q, long long& i0, int& phys) {
    if (p.contiguous) {
        i0 = base + 8LL * q;
        phys = q + (p.skew_magic ? (int)__umulhi((unsigned)q, p.skew_magic) : 0);
    } else {
        const int row = q / CPR, c = q - row * CPR;
        i0 = base + (long long)row * p.D + 8LL * c;
        phys = row * p.RS + c;
    }
}

// Staging in two halves so that the NEXT chunk's loads fly while this chunk is multiplied: `fetch` issues the loads of a
// thread's first kDepth chunks of the span (interior ones; 16 bytes each, raw, into registers: nothing waits for them),
// `commit` - a whole MFMA phase later - converts them, splits them into (hi, lo) f16 halves and writes LDS, then takes
// whatever is left (edges of the signal one sample at a time, spans longer than kDepth x 256 chunks in groups of four
// loads).  Without it a workgroup has bytes in flight only while it stages, a third of its time, and the stage runs at the
// rate "bytes in flight / memory latency" allows: 2.8 TB/s where a copy reaches 6.3.
template <typename T, int CH, int BUDGET>               // BUDGET: VGPRs the raw samples may occupy beside the filter and the accumulators
struct FirPrefetch {
    static constexpr int kQuads = CH ? (int)(sizeof(T) * CH / 2) : 1;       // 16-byte loads per chunk
    static constexpr int kFit = (BUDGET < 56 ? BUDGET : 56) / (4 * kQuads);
    static constexpr int kDepth = CH == 0 ? 0 : (kFit > 7 ? 7 : kFit < 0 ? 0 : kFit);
    FirRaw<T, (CH ? CH : 1)> raw[kDepth ? kDepth : 1];
    unsigned fast = 0;                                     // bit i: raw[i] holds the thread's i-th chunk

    template <int CPR>
    __device__ __forceinline__ void fetch(const T* __restrict__ in, long long n_in, const FirPlan& p, long long base, int total, int tid) {
        fast = 0;
        if constexpr (CH != 0) {
#pragma unroll
            for (int it = 0; it < kDepth; ++it) {
                const int q = tid + it * kFirThreads;
                long long i0;
                int phys;
                fir_chunk_at<CPR>(p, base, q, i0, phys);
                if (q < total && i0 >= 0 && i0 + 8 <= n_in) {
                    raw[it] = fir_fetch<T, (CH ? CH : 1)>(in, i0);
                    fast |= 1u << it;
                }
            }
        }
    }

    template <int CPR>
    __device__ __forceinline__ void commit(const T* __restrict__ in, long long n_in, int channels, const FirPlan& p, long long base,
                                           int total, char* a_hi, char* a_lo, int tid) {
#pragma unroll
        for (int it = 0; it < kDepth; ++it) {
            const int q = tid + it * kFirThreads;
            if (q < total) {
                long long i0;
                int phys;
                fir_chunk_at<CPR>(p, base, q, i0, phys);
                float v[8];
                if (fast >> it & 1u) {
                    if constexpr (CH != 0) fir_convert<T, (CH ? CH : 1)>(raw[it], v);
                } else {
                    fir_slow8(in, i0, n_in, channels, v);
                }
                fir_put8(a_hi, a_lo, phys, v);
            }
        }
        constexpr int GROUP = kQuads <= 2 ? 4 : 2;          // the rest of a long span: a few chunks' loads in flight together
        for (int q0 = tid + kDepth * kFirThreads; q0 < total; q0 += GROUP * kFirThreads) {
            long long i0[GROUP];
            int phys[GROUP];
            bool interior = true;
#pragma unroll
            for (int gi = 0; gi < GROUP; ++gi) {
                const int q = q0 + gi * kFirThreads;
                fir_chunk_at<CPR>(p, base, q, i0[gi], phys[gi]);
                if (q >= total) {                           // past the span: load the group's first chunk again, store nothing
                    i0[gi] = i0[0];
                    phys[gi] = -1;
                }
                interior = interior && i0[gi] >= 0 && i0[gi] + 8 <= n_in;
            }
            if (CH != 0 && __all(interior)) {
                if constexpr (CH != 0) {
                    FirRaw<T, (CH ? CH : 1)> r4[GROUP];
#pragma unroll
                    for (int gi = 0; gi < GROUP; ++gi) r4[gi] = fir_fetch<T, (CH ? CH : 1)>(in, i0[gi]);
                    // every load of the group is issued before the first value is converted (left alone, the scheduler
                    // sinks each chunk's loads to its conversion: one exposed memory round trip per chunk)
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int gi = 0; gi < GROUP; ++gi) {
                        float v[8];
                        fir_convert<T, (CH ? CH : 1)>(r4[gi], v);
                        if (phys[gi] >= 0) fir_put8(a_hi, a_lo, phys[gi], v);
                    }
                }
            } else {
#pragma unroll 1
                for (int gi = 0; gi < GROUP; ++gi) {
                    if (q0 + gi * kFirThreads >= total) break;
                    float v[8];
                    fir_slow8(in, i0[gi], n_in, channels, v);
                    fir_put8(a_hi, a_lo, phys[gi], v);
                }
            }
        }
    }
};

template <typename T, int KQ, int MT, int CH>
__device__ __forceinline__ void fir_body(const T* __restrict__ in, long long n_in, int channels, const FirPlan& p,
                                         float* __restrict__ out, long long n_out, long long n_chunks, char* smem) {
    constexpr int ROWS = 32 * MT;
    constexpr int KSTOT = kFirWaves * KQ;
    constexpr int CPR = 2 * KSTOT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pb = blockIdx.y;

    // this wave's share of the filter: k-steps [wave KQ, wave KQ + KQ) of phase block pb, fragment order
    f16x8 bh[KQ], bl[KQ];
    {
        const f16x8* g = reinterpret_cast<const f16x8*>(p.gfrag) + ((size_t)pb * KSTOT + (size_t)wave * KQ) * 2 * 64 + lane;
#pragma unroll
        for (int i = 0; i < KQ; ++i) {
            bh[i] = g[(2 * i) * 64];
            bl[i] = g[(2 * i + 1) * 64];
        }
    }
    int koff[KQ];
#pragma unroll
    for (int i = 0; i < KQ; ++i) koff[i] = p.koff[wave * KQ + i];

    char* const a_hi = smem;
    char* const a_lo = smem + p.a_bytes;
    const float unscale = p.unscale[sizeof(T) == 2 ? 0 : 1];
    const int total = p.contiguous ? (ROWS - 1) * (p.D >> 3) + CPR : ROWS * CPR;
    const long long boff = p.boff[pb];
    FirPrefetch<T, CH, 220 - 8 * KQ - 16 * MT - 44> pf;   // 256 registers at two waves per SIMD: filter, accumulators, the fragment ring, ~50 others
    if ((long long)blockIdx.x < n_chunks) pf.template fetch<CPR>(in, n_in, p, (long long)blockIdx.x * ROWS * p.D + boff, total, tid);
    // persistent: the workgroup keeps its share of the filter in registers and walks chunks of ROWS periods
    for (long long chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const long long m0 = chunk * ROWS;
    // ---- stage: channel mean -> (hi, lo) f16 halves in LDS; then the next chunk's loads are put in flight ----
    pf.template commit<CPR>(in, n_in, channels, p, m0 * p.D + boff, total, a_hi, a_lo, tid);
    __syncthreads();
    if (chunk + gridDim.x < n_chunks) pf.template fetch<CPR>(in, n_in, p, (m0 + (long long)gridDim.x * ROWS) * p.D + boff, total, tid);

    // ---- the product: this wave's k-steps over every row tile ----
    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
    int abase[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) abase[t] = 16 * ((t * 32 + (lane & 31)) * p.RS + (lane >> 5));
    // The A fragments run two steps (six MFMAs) ahead of their use in a ring of three (hi, lo) pairs, each request pinned in
    // front of the MFMAs of the step it follows: left alone the compiler keeps ONE pair and every step opens with
    // ds_read_b128 -> s_waitcnt lgkmcnt(0), the whole LDS latency in front of three MFMAs (round 5: the matrix pipe ran at
    // 40 % of its pace inside this loop).
    {
        constexpr int NSTEP = KQ * MT;
        f16x8 ah[3], al[3];
#if defined(BD_FIR_ABLATE) && BD_FIR_ABLATE == 1      // developer build: no fragment reads (which LDS access conflicts?)
#define FIR_LA(N) { ah[(N) % 3] = bh[(N) / MT]; al[(N) % 3] = bl[(N) / MT]; asm volatile("" : "+v"(ah[(N) % 3]), "+v"(al[(N) % 3])); }
#else
#define FIR_LA(N)                                                                                         \
    {                                                                                                     \
        ah[(N) % 3] = *reinterpret_cast<const f16x8*>(a_hi + abase[(N) % MT] + koff[(N) / MT]);           \
        al[(N) % 3] = *reinterpret_cast<const f16x8*>(a_lo + abase[(N) % MT] + koff[(N) / MT]);           \
    }
#endif
        FIR_LA(0)
        if constexpr (NSTEP > 1) FIR_LA(1)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NSTEP; ++n) {
            const int i = n / MT, t = n % MT;
            if (n + 2 < NSTEP) FIR_LA(n + 2)
            __builtin_amdgcn_sched_barrier(0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[n % 3], bh[i], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[n % 3], bl[i], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[n % 3], bh[i], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef FIR_LA
    }
    __syncthreads();                                       // every wave is done with the staged signal

    // ---- the four partial tiles meet in LDS: red[wave][tile][register quad][lane], 16 bytes per lane ----
    // (a native 4-vector, so that both sides are ONE ds_write_b128 / ds_read_b128 per lane: with HIP's float4 struct the
    //  compiler split the reads into ds_read_b32 / ds_read2_b32 at a 16-byte lane stride - 8 lanes per bank, 24 conflict cycles
    //  per instruction, 92 % of the kernel's SQ_LDS_BANK_CONFLICT (round 5, tools/fir_conflicts.sh))
    typedef float fir_v4 __attribute__((ext_vector_type(4)));
    fir_v4* const red = reinterpret_cast<fir_v4*>(smem);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#if defined(BD_FIR_ABLATE) && BD_FIR_ABLATE == 3      // developer build: partial tiles not written (the sums below read garbage)
            asm volatile("" ::"v"(acc[t][4 * g]), "v"(acc[t][4 * g + 1]), "v"(acc[t][4 * g + 2]), "v"(acc[t][4 * g + 3]));
#else
            red[((wave * MT + t) * 4 + g) * 64 + lane] = fir_v4{acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
#endif
    __syncthreads();
    const int g = wave;                                    // this wave sums register quad g of every tile
    const int col = lane & 31;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        fir_v4 s = red[((0 * MT + t) * 4 + g) * 64 + lane];
#pragma unroll
        for (int w = 1; w < kFirWaves; ++w) {
            fir_v4 r = red[((w * MT + t) * 4 + g) * 64 + lane];
            asm volatile("" : "+v"(r));                     // (one 16-byte read: not four dword reads the scheduler may spread out)
            s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w;
        }
        // accumulator register 4 g + e of lane l is row e + 8 g + 4 (l >> 5), column l & 31 of the tile
        const long long row0 = m0 + 32 * t + 8 * g + 4 * (lane >> 5);
        const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long long j = ((row0 + e) * p.NB + pb) * 32 + col;
            if (j < n_out) out[j] = sv[e] * unscale;
        }
    }
    __syncthreads();                                       // the partial tiles are read: the next chunk may be staged over them
    }   // chunks
}

template <typename T, int KQ, int MT, int CH>     // CH: 1 = mono, 2 = stereo, 0 = any number of channels (one sample at a time)
__global__ __launch_bounds__(kFirThreads, 2) void fir_mfma_kernel(const T* __restrict__ in, long long n_in, int channels,
                                                                  const FirPlan p, float* __restrict__ out, long long n_out,
                                                                  long long n_chunks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    fir_body<T, KQ, MT, CH>(in, n_in, channels, p, out, n_out, n_chunks, smem);
}

long long ceil_div(long long a, long long b) { return a >= 0 ? (a + b - 1) / b : -((-a) / b); }
long long floor_div(long long a, long long b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }
int gcd_i(int a, int b) {
    while (b) {
        const int t = a % b;
        a = b;
        b = t;
    }
    return a;
}
int lcm_i(int a, int b) { return a / gcd_i(a, b) * b; }

constexpr int kKqChoices[] = {4, 6, 8, 11, 16, 21};

}  // namespace

// Geometry and the fragment-ordered filter of one rate ratio.  h = the 2 half + 1 taps in double precision (already times
// `up`).  Returns false when the ratio does not fit the kernel (filter too long for four waves' registers, LDS).
bool fir_plan_build(int up, int down, const double* h, int half, FirPlanHost* out) {
    const int r = lcm_i(32 / gcd_i(up, 32), 8 / gcd_i(down, 8));
    const long long P = (long long)up * r, D = (long long)down * r;
    if (P > (1 << 20) || D > (1 << 24)) return false;
    const int NB = (int)(P / 32);
    std::vector<int> boff(NB);
    int ks_real = 0;
    for (int b = 0; b < NB; ++b) {
        const long long first = ceil_div(32LL * b * down - half, up);
        const long long off = floor_div(first, 8) * 8;
        const long long last = floor_div((32LL * b + 31) * down + half, up);
        const int k = (int)(last - off + 1);
        boff[b] = (int)off;
        if ((k + 15) / 16 > ks_real) ks_real = (k + 15) / 16;
    }
    int kq = 0;
    for (int c : kKqChoices)
        if (kFirWaves * c >= ks_real) {
            kq = c;
            break;
        }
    if (!kq) return false;
    const int s = (int)(D / 8);
    const bool contiguous = NB == 1;
    FirPlanHost& o = *out;
    o.plan = FirPlan{};
    // rows per workgroup: 128 (64 for the long filters: the accumulators share the register file with the filter) when
    // the staged signal fits the LDS twice over (two workgroups per CU), else 32 (separate row pieces: 44.1 kHz)
    int mt = 0;
    for (int cand : {kq <= 11 ? 4 : 2, 1}) {
        const int rows = 32 * cand, kstot = kFirWaves * kq;
        long long chunks;
        if (contiguous) {
            const long long nch = (long long)(rows - 1) * s + 2 * kstot;
            if (nch >= (1 << 16)) continue;
            chunks = nch + ((s % 2) == 0 ? nch / s + 1 : 0);
        } else {
            chunks = (long long)rows * ((2 * kstot) | 1);
        }
        const long long red_bytes = (long long)kFirWaves * cand * 4 * 64 * 16;
        const long long lds = 32 * chunks > red_bytes ? 32 * chunks : red_bytes;
        if (lds <= (cand == 1 ? 160 : 80) * 1024) {
            mt = cand;
            o.plan.a_bytes = (int)(chunks * 16);
            o.plan.lds_bytes = (int)lds;
            break;
        }
    }
    if (!mt) return false;
    const int kstot = kFirWaves * kq;
    o.plan.up = up;
    o.plan.down = down;
    o.plan.P = (int)P;
    o.plan.D = (int)D;
    o.plan.NB = NB;
    o.plan.kq = kq;
    o.plan.mt = mt;
    o.plan.contiguous = contiguous ? 1 : 0;
    o.koff.resize(kstot);
    if (contiguous) {
        const bool skew = (s % 2) == 0;
        o.plan.RS = skew ? s + 1 : s;
        // floor(q / s) as a multiply-high; exact for q < 2^16 (checked above)
        o.plan.skew_magic = skew ? (unsigned)(((1ULL << 32) + s - 1) / s) : 0u;
        for (int ks = 0; ks < kstot; ++ks) o.koff[ks] = 16 * (2 * ks + (skew ? (2 * ks) / s : 0));
    } else {
        o.plan.RS = (2 * kstot) | 1;
        o.plan.skew_magic = 0;
        for (int ks = 0; ks < kstot; ++ks) o.koff[ks] = 32 * ks;
    }
    // the filter scaled by an exact power of two so that its largest tap lies in [2^10, 2^11), then hi + lo f16 halves
    double hmax = 0.0;
    for (int t = 0; t <= 2 * half; ++t) hmax = std::fabs(h[t]) > hmax ? std::fabs(h[t]) : hmax;
    if (!(hmax > 0.0)) return false;
    int ex;
    (void)std::frexp(hmax, &ex);                            // hmax = f * 2^ex, f in [0.5, 1)
    const int sh = 11 - ex;                                 // hmax * 2^sh in [2^10, 2^11)
    o.plan.unscale[0] = (float)std::ldexp(1.0, -(sh + 15));
    o.plan.unscale[1] = (float)std::ldexp(1.0, -(sh + 6));
    o.boff = boff;
    o.gfrag.assign((size_t)NB * kstot * 2 * 64 * 8, (uint16_t)0);
    _Float16* g = reinterpret_cast<_Float16*>(o.gfrag.data());
    for (int b = 0; b < NB; ++b)
        for (int ks = 0; ks < kstot; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const long long e = 16LL * ks + 8 * (l >> 5) + j;
                    const long long t = (32LL * b + (l & 31)) * down - ((long long)boff[b] + e) * up + half;
                    const double v = (t >= 0 && t <= 2LL * half) ? std::ldexp(h[t], sh) : 0.0;
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)(v - (double)hi);
                    const size_t at = (((size_t)b * kstot + ks) * 2) * 64 * 8 + (size_t)l * 8 + j;
                    g[at] = hi;
                    g[at + 64 * 8] = lo;
                }
    return true;
}

void launch_fir_mfma(const void* in, bool s16, int64_t n_in, int channels, const FirPlan& p, float* out, int64_t n_out,
                     hipStream_t stream) {
    if (n_out <= 0) return;
    const int rows = 32 * p.mt;
    const int64_t periods = (n_out + p.P - 1) / p.P;
    const int64_t n_chunks = (periods + rows - 1) / rows;
    // persistent workgroups: as many as are resident together (two or three per CU by LDS and registers), each walks
    // chunks grid.x apart with its share of the filter in registers
    const int cus = cu_count();
    const int per_cu = p.lds_bytes <= 40 * 1024 ? 3 : p.lds_bytes <= 80 * 1024 ? 2 : 1;
    int64_t gx = (int64_t)cus * per_cu / p.NB;             // (rounded down: one workgroup too many per CU is a whole extra round)
    if (gx > n_chunks) gx = n_chunks;
    if (gx < 1) gx = 1;
    const dim3 grid((unsigned)gx, (unsigned)p.NB);
#define BD_FIR_LAUNCH_CH(T, KQ, MT, CH)                                                                           \
    do {                                                                                                          \
        static std::once_flag once_[16];                                                                          \
        int dev_ = 0;                                                                                             \
        (void)hipGetDevice(&dev_);                                                                                \
        std::call_once(once_[dev_ & 15], [] {                                                                     \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fir_mfma_kernel<T, KQ, MT, CH>),             \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                    \
        });                                                                                                       \
        hipLaunchKernelGGL((fir_mfma_kernel<T, KQ, MT, CH>), grid, dim3(kFirThreads), (size_t)p.lds_bytes, stream, \
                           static_cast<const T*>(in), (long long)n_in, channels, p, out, (long long)n_out,        \
                           (long long)n_chunks);                                                                  \
    } while (0)
#define BD_FIR_LAUNCH(T, KQ, MT)                                                                                  \
    do {                                                                                                          \
        if (channels == 1) BD_FIR_LAUNCH_CH(T, KQ, MT, 1);                                                        \
        else if (channels == 2) BD_FIR_LAUNCH_CH(T, KQ, MT, 2);                                                   \
        else BD_FIR_LAUNCH_CH(T, KQ, MT, 0);                                                                      \
    } while (0)
#define BD_FIR_BY_KQ(T)                                                                                           \
    switch (p.kq * 8 + p.mt) {                                                                                    \
        case 4 * 8 + 4: BD_FIR_LAUNCH(T, 4, 4); break;                                                            \
        case 6 * 8 + 4: BD_FIR_LAUNCH(T, 6, 4); break;                                                            \
        case 8 * 8 + 4: BD_FIR_LAUNCH(T, 8, 4); break;                                                            \
        case 11 * 8 + 4: BD_FIR_LAUNCH(T, 11, 4); break;                                                          \
        case 16 * 8 + 2: BD_FIR_LAUNCH(T, 16, 2); break;                                                          \
        case 21 * 8 + 2: BD_FIR_LAUNCH(T, 21, 2); break;                                                          \
        case 4 * 8 + 1: BD_FIR_LAUNCH(T, 4, 1); break;                                                            \
        case 6 * 8 + 1: BD_FIR_LAUNCH(T, 6, 1); break;                                                            \
        case 8 * 8 + 1: BD_FIR_LAUNCH(T, 8, 1); break;                                                            \
        case 11 * 8 + 1: BD_FIR_LAUNCH(T, 11, 1); break;                                                          \
        case 16 * 8 + 1: BD_FIR_LAUNCH(T, 16, 1); break;                                                          \
        default: BD_FIR_LAUNCH(T, 21, 1); break;                                                                  \
    }
    if (s16) { BD_FIR_BY_KQ(short) } else { BD_FIR_BY_KQ(float) }
#undef BD_FIR_BY_KQ
#undef BD_FIR_LAUNCH
#undef BD_FIR_LAUNCH_CH
}

}  // namespace bd
