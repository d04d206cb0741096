// Front end of the analyze hot path on gfx950: PCM -> log-mel spectrogram, one kernel.
//
// Restates embedders/yamnet/features.py:22-58 (tf.signal.stft 400/160/512 -> tf.abs ->
// matmul with the [257,64] mel matrix -> log(x + 0.001)) with pad_waveform (features.py:82-108)
// folded in as "samples past n_valid read as zero".  Nothing of [T,400] / [T,257] is ever
// written to HBM.
//
// logmel_kernel: one 768-thread workgroup per CU (three waves per SIMD) walks groups of 48 STFT frames.
//   FFT phase   sixteen lanes own a frame, a wave four frames: the frame is read straight from global memory
//               (buffer resource: samples past the end read as zero) as 200 packed complex points
//               z[n] = x[2n] + i x[2n+1] (+56 zeros), Hann-windowed, and transformed as 256 = 16 x 16:
//               DFT-16 over n1 in registers, twiddle, ONE transpose through the wave's own LDS tile, DFT-16
//               over n2 in registers - all on packed f32 pairs (v_pk_add/mul/fma with op_sel / neg modifiers).
//               The upper half of the packed spectrum goes through the tile once more in natural order so
//               that every lane can pick up the mirrored bins Z[256 - k] of half of its own bins and split
//               BOTH X[k] and X[256 - k] out of one (Z[k], Z[256 - k]) pair.  |X| lands in a
//               [48 frames][244] f32 tile.  Wave-level ordering only, no workgroup barrier inside the phase;
//               the next group's samples are requested before the barrier that ends it.
//   mel phase   lane = frame: each of the twelve waves owns a run of mel bands (balanced by non-zeros);
//               a band is a short run of bins, so its weights are wave-uniform and come in through
//               scalar loads - 461 v_fmac per 64 frames instead of 18 per frame and lane, and every LDS
//               read is a conflict-free 16-byte row read.  log(), then the row is staged in the LDS region
//               of the wave that will store it.
//   output      coalesced 256-byte rows of the [T,64] log-mel buffer (buffer stores: rows past T are dropped).
// Measured (profiles/r02_*): 37 us per 98 304 frames = 2.4 TB/s algorithmic; HBM traffic = algorithmic bytes.
//
// Algorithmic HBM traffic: 640 B read (160 new samples) + 256 B written per frame.
#include "bd_internal.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace bd {

namespace {

namespace fe {

constexpr int kWavesG = 12;            // one workgroup per CU, three waves per SIMD: 168 VGPRs keep the per-lane twiddles in
constexpr int kThreads = 64 * kWavesG; // registers and the LDS affords every wave a full four-frame transpose tile
constexpr int kGroup = 4 * kWavesG;    // 48 frames per workgroup pass: every wave transforms four of them
constexpr int kXRow = 36;              // transpose tile: row (frame, k1) = 16 float2 + 16 B of bank padding, in dwords
constexpr int kXFrame = 16 * kXRow;    // 576 dwords per frame, 2304 for the wave's four
constexpr int kZFrame = 2 * 128 + 32;  // natural-order tile of the upper half spectrum: 128 float2 + 128 B bank shift
constexpr int kTile = 2308;            // a wave's LDS region in dwords (9232 B); 2308 = 4 mod 32 spreads the staged
                                       // output rows of the regions over the banks
constexpr int kMagRow = 244;           // |X| tile row: bins 0..243 (5..239 are used); 244 = 52 mod 64 -> 16-byte row reads of
                                       // sixteen consecutive lanes cover all 64 banks
constexpr int kOutRow = 65;
constexpr int kTabRows = 13;           // per-lane Hann pairs in LDS, [row][16] float2 (read once per group, at a quiet moment)
static_assert(4 * kXFrame <= kTile && 4 * kZFrame <= kTile && 4 * kOutRow <= kTile, "a wave's region holds each of its tiles");

// mel bands of wave w in the mel phase: [kCut[w], kCut[w + 1]); balanced on (non-zeros + log) per band under the
// constraint that a run holds at most 48 weights counted from a 16-byte boundary (three s_load_dwordx16)
constexpr int kCut[kWavesG + 1] = {0, 11, 20, 28, 35, 41, 46, 50, 54, 57, 60, 62, 64};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Complex arithmetic on packed f32 pairs (re, im) = one 64-bit register pair.  + - * on v2f compile to v_pk_add /
// v_pk_mul; the forms that swap or negate a half use the VOP3P op_sel / neg modifiers, which the compiler does not
// derive from scalar code (it answered the float2 version of this kernel with ~130 v_mov per four frames).
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v2f add_mi(v2f a, v2f b) {      // a + (-i) b = (a.x + b.y, a.y - b.x)
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ v2f sub_mi(v2f a, v2f b) {      // a - (-i) b = (a.x - b.y, a.y + b.x)
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a * b as two instructions: cmul1 = (-a.y b.y, a.y b.x), cmul2 = (a.x b.x, a.x b.y) + that.  Callers issue the first
// halves of several independent products before the second halves (the second waits on the first otherwise, and the
// compiler pads a dependent pair of asm statements with s_nop).  _v: b in vector registers, _s: b wave-uniform.
__device__ __forceinline__ v2f cmul1_v(v2f a, v2f b) {
    v2f t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    return t;
}
__device__ __forceinline__ v2f cmul2_v(v2f a, v2f b, v2f t) {
    v2f d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(t));
    return d;
}
__device__ __forceinline__ v2f cmul1_s(v2f a, v2f b) {
    v2f t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "s"(b));
    return t;
}
__device__ __forceinline__ v2f cmul2_s(v2f a, v2f b, v2f t) {
    v2f d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(d) : "v"(a), "s"(b), "v"(t));
    return d;
}
__device__ __forceinline__ v2f add_conj(v2f a, v2f b) {    // a + conj(b) = (a.x + b.x, a.y - b.y)
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ v2f odd_part(v2f a, v2f b) {    // -i (a - conj(b)) = (a.y + b.y, b.x - a.x)
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ v2f fma_s(v2f a, v2f s, v2f c) {        // a * s + c, s wave-uniform
    v2f d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(s), "v"(c));
    return d;
}
__device__ __forceinline__ v2f fms_s(v2f a, v2f s, v2f c) {        // a * s - c
    v2f d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(d) : "v"(a), "s"(s), "v"(c));
    return d;
}

// forward DFT-4; with U2MI the third input is given as w where u2 = -i w (a folded W16^4 twiddle)
template <bool U2MI = false>
__device__ __forceinline__ void radix4(v2f u0, v2f u1, v2f u2, v2f u3, v2f& y0, v2f& y1, v2f& y2, v2f& y3) {
    const v2f s0 = U2MI ? add_mi(u0, u2) : u0 + u2;
    const v2f s1 = U2MI ? sub_mi(u0, u2) : u0 - u2;
    const v2f s2 = u1 + u3, d3 = u1 - u3;
    y0 = s0 + s2;
    y2 = s0 - s2;
    y1 = add_mi(s1, d3);
    y3 = sub_mi(s1, d3);
}

// in-place forward DFT of 16 points, natural order in and out (two radix-4 stages in registers)
__device__ __forceinline__ void dft16(v2f (&x)[16]) {
    constexpr float C = 0.92387953251128674f, S = 0.38268343236508977f, H = 0.70710678118654752f;
    const v2f W1 = {C, -S}, W2 = {H, -H}, W3 = {S, -C}, W6 = {-H, -H}, W9 = {-C, S};
    v2f a[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) radix4(x[q], x[4 + q], x[8 + q], x[12 + q], a[q][0], a[q][1], a[q][2], a[q][3]);
    //  a[2][2] * W16^4 = -i a[2][2]: folded into the second stage (U2MI)
    const v2f t11 = cmul1_s(a[1][1], W1), t12 = cmul1_s(a[1][2], W2), t13 = cmul1_s(a[1][3], W3), t21 = cmul1_s(a[2][1], W2);
    const v2f t23 = cmul1_s(a[2][3], W6), t31 = cmul1_s(a[3][1], W3), t32 = cmul1_s(a[3][2], W6), t33 = cmul1_s(a[3][3], W9);
    a[1][1] = cmul2_s(a[1][1], W1, t11);
    a[1][2] = cmul2_s(a[1][2], W2, t12);
    a[1][3] = cmul2_s(a[1][3], W3, t13);
    a[2][1] = cmul2_s(a[2][1], W2, t21);
    a[2][3] = cmul2_s(a[2][3], W6, t23);
    a[3][1] = cmul2_s(a[3][1], W3, t31);
    a[3][2] = cmul2_s(a[3][2], W6, t32);
    a[3][3] = cmul2_s(a[3][3], W9, t33);
    radix4(a[0][0], a[1][0], a[2][0], a[3][0], x[0], x[4], x[8], x[12]);
    radix4(a[0][1], a[1][1], a[2][1], a[3][1], x[1], x[5], x[9], x[13]);
    radix4<true>(a[0][2], a[1][2], a[2][2], a[3][2], x[2], x[6], x[10], x[14]);
    radix4(a[0][3], a[1][3], a[2][3], a[3][3], x[3], x[7], x[11], x[15]);
}

__device__ __forceinline__ unsigned lds_addr(const void* p) {      // byte address inside the workgroup's LDS
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}

template <int OFFSET>       // the caller waits (s_waitcnt lgkmcnt) before using the value: the compiler does not track it
__device__ __forceinline__ v4f lds_read128(unsigned addr) {
    v4f v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFFSET) : "memory");
    return v;
}

// after the s_waitcnt that covers an lds_read128: makes every later use of the value depend on a statement the compiler
// keeps behind that wait (it may otherwise move plain arithmetic on the register above the wait)
__device__ __forceinline__ v4f lds_landed(v4f v) {
    asm volatile("" : "+v"(v));
    return v;
}

__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }   // a wave's DS operations execute in order

typedef float v16f __attribute__((ext_vector_type(16)));

template <int BYTE_OFFSET>      // 16 wave-uniform floats into scalar registers; the caller waits (lgkmcnt) before use
__device__ __forceinline__ v16f scalar_load16(const float* p) {
    v16f v;
    asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(v) : "s"(p), "n"(BYTE_OFFSET) : "memory");
    return v;
}
__device__ __forceinline__ v16f scalar_landed(v16f v) {
    asm volatile("" : "+s"(v));
    return v;
}

// mel phase of wave W: lane = frame; bands [kCut[W], kCut[W + 1]) over that frame's |X| row.  The weights are the same for
// every lane: they come in as scalar loads (the scalar cache holds the 1.8 KB table after the first group), requested
// before the one wait that also covers the |X| row reads, and are used as the scalar operand of v_fmac.  (Requesting them
// ahead of the barrier that ends the FFT phase made the compiler spill the still-empty scalar registers: no gain.)
template <int W>
__device__ __forceinline__ void mel_wave(const float* __restrict__ mag_row, float* __restrict__ out_row,
                                         const float* __restrict__ melw) {
    constexpr int b0 = kCut[W], b1 = kCut[W + 1];
    constexpr int klo = kMelStart[b0] & ~3;
    constexpr int khi = (kMelStart[b1 - 1] + kMelLen[b1 - 1] + 3) & ~3;
    constexpr int wlo = mel_offset(b0) & ~3, n16 = (mel_offset(b1) - wlo + 15) / 16;
    static_assert(n16 <= 3, "a wave's run of bands holds at most 48 weights");
    // the weights first (scalar loads, longest latency), then the |X| row, one wait for both
    v16f w16[n16];
    static_for<0, n16>([&](auto qi) { w16[decltype(qi)::value] = scalar_load16<4 * wlo + 64 * decltype(qi)::value>(melw); });
    v4f m4[(khi - klo) / 4];
    // ds_read_b128 by hand: the compiler narrows float4 loads to the components that are used and then only
    // knows 8-byte alignment (ds_read2_b64: twice the LDS cycles and bank conflicts on the row stride)
    const unsigned m_addr = lds_addr(mag_row + klo);
    static_for<0, (khi - klo) / 4>([&](auto qi) { m4[decltype(qi)::value] = lds_read128<16 * decltype(qi)::value>(m_addr); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    static_for<0, n16>([&](auto qi) { w16[decltype(qi)::value] = scalar_landed(w16[decltype(qi)::value]); });
    static_for<0, (khi - klo) / 4>([&](auto qi) { m4[decltype(qi)::value] = lds_landed(m4[decltype(qi)::value]); });
    const float* m = reinterpret_cast<const float*>(m4);
    static_for<b0, b1>([&](auto bi) {
        constexpr int b = decltype(bi)::value;
        constexpr int off = mel_offset(b) - wlo, st = kMelStart[b] - klo;
        float acc = 0.0f;
        static_for<0, kMelLen[b]>([&](auto ji) {
            constexpr int j = decltype(ji)::value;
            acc = fmaf(m[st + j], w16[(off + j) / 16][(off + j) % 16], acc);
        });
        // acc + 0.001 is in [1e-3, ~1e3]: v_log_f32 (1 ulp, no denormal range) * ln 2
        out_row[b] = __builtin_amdgcn_logf(acc + 0.001f) * 0.69314718055994531f;
    });
}

// BD_FE_TRACE (tools/fe_trace.hip only): s_memtime stamps of two waves of one workgroup, second group it handles
#ifdef BD_FE_TRACE
#define FE_TRACE_ARG , unsigned long long* __restrict__ stamps
#define FE_STAMP(I)                                                                                        \
    if (stamps && blockIdx.x == 5 && group == (int)(blockIdx.x + gridDim.x) && lane == 0 && (wave == 0 || wave == kWavesG - 1)) \
        stamps[(wave ? 32 : 0) + (I)] = __builtin_amdgcn_s_memtime();
#else
#define FE_TRACE_ARG
#define FE_STAMP(I)
#endif

__global__ __launch_bounds__(kThreads) void logmel_kernel(const float* __restrict__ pcm, int n_valid, int n_frames,
                                                          float* __restrict__ out,
                                                          const FeTables* __restrict__ tab FE_TRACE_ARG) {
    __shared__ __attribute__((aligned(16))) float s_mag[kGroup * kMagRow];        //  46 848 B
    __shared__ __attribute__((aligned(16))) float s_x[kWavesG * kTile];           // 110 784 B
    __shared__ __attribute__((aligned(16))) v2f s_tab[kTabRows * 16];             //   1 664 B

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fq = lane >> 4;              // frame of the wave's four
    const int j16 = lane & 15;             // n2 before the transpose, k1 after it

    // Per-lane constants (a lane's n2 / k1 never changes).  Hann taps of z[16 n1 + n2] (zero past sample 400) sit in LDS,
    // [n1][16], and are read once per group; the twiddles stay in registers.
    if (tid < kTabRows * 16) {
        const int r = tid >> 4, c = tid & 15;
        s_tab[tid] = v2f{tab->hann[2 * (16 * r + c)], tab->hann[2 * (16 * r + c) + 1]};
    }
    const v2f* const t_hann = s_tab + j16;
    v2f tw[15];                            // W256^(n2 k1), k1 = 1..15
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) {
        const float2 t = tab->tw256[(j16 * k1) & 255];
        tw[k1 - 1] = v2f{t.x, t.y};
    }
    v2f tws[8];                            // 0.5 * exp(-2 pi i k / 512), k = k1 + 16 k2, k2 = 0..7 (the 1/2 of the split)
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) {
        const float2 t = tab->tw512[j16 + 16 * k2];
        tws[k2] = v2f{0.5f * t.x, 0.5f * t.y};
    }
    const float* const melw = tab->melw;
    const v2f kHalf = {0.5f, 0.5f};

    float* const xw = s_x + wave * kTile;                       // this wave's region
    v2f* const x_wr = reinterpret_cast<v2f*>(xw + fq * kXFrame) + j16;                        // + k1 * (kXRow / 2)
    const float4* const x_rd = reinterpret_cast<const float4*>(xw + fq * kXFrame + j16 * kXRow);
    v2f* const z_wr = reinterpret_cast<v2f*>(xw + fq * kZFrame) + j16;                       // + 16 (k2 - 8)
    const v2f* const z_base = reinterpret_cast<const v2f*>(xw + fq * kZFrame);
    // mirrored bin of (k1, k2) minus 128: k2 = 0: 128 - k1 (k1 = 0: the pair of bin 0 feeds no mel band; read something
    // finite); k2 >= 1: 112 - k1 - 16 (k2 - 1)
    const int zm0 = j16 ? 128 - j16 : 0;
    const int zm1 = 112 - j16;

    const int n_groups = (n_frames + kGroup - 1) / kGroup;
    // PCM goes through a buffer resource whose size is n_valid samples: a dword past the end reads as zero, which
    // IS pad_waveform's zero padding, with no bounds code and no second code path.  Lanes whose thirteenth point
    // lies past sample 400 of the frame (n2 >= 8) read from an offset beyond any buffer instead: zero as well.
    const __amdgpu_buffer_rsrc_t pcm_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pcm), 0, n_valid * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(out, 0, n_frames * (BD_MEL_BANDS * 4), 0x00020000);
    const int off12 = j16 < 8 ? 12 * 128 : 0x7f000000;        // byte offset of the n1 = 12 point relative to the lane's first
    // the 13 packed points of this lane's frame in group g
    auto load_group = [&](v2f (&raw)[13], int g) {
        const int frame = g * kGroup + wave * 4 + fq;
        const int byte0 = (frame * BD_STFT_HOP + 2 * j16) * 4;
#pragma unroll
        for (int n1 = 0; n1 < 12; ++n1)
            raw[n1] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(pcm_rsrc, byte0 + 128 * n1, 0, 0));
        raw[12] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(pcm_rsrc, byte0 + off12, 0, 0));
    };
    // raw[]: samples in flight for the next group; win[]: the windowed points of the group about to run.  A group's
    // samples are requested at the end of the previous group's transform and turned into win[] at a point where nothing
    // younger than them is outstanding (before the output stores: s_waitcnt vmcnt counts loads and stores in order).
    v2f raw[13], win[13];
    __syncthreads();                                           // s_tab
    load_group(raw, blockIdx.x);
#pragma unroll
    for (int n1 = 0; n1 < 13; ++n1) win[n1] = raw[n1] * t_hann[16 * n1];
    for (int group = blockIdx.x; group < n_groups; group += gridDim.x) {
        // ---------------- FFT phase: this wave's four frames ----------------
        {
#ifdef BD_FE_TRACE
            constexpr int round = 0;
#endif
            const int fl = wave * 4 + fq;                      // frame within the group
            v2f u[16];
            FE_STAMP(round * 8 + 0)
#pragma unroll
            for (int n1 = 0; n1 < 13; ++n1) u[n1] = win[n1];
            u[13] = u[14] = u[15] = v2f{0.f, 0.f};
            FE_STAMP(round * 8 + 1)
            dft16(u);                                          // over n1 -> k1
#pragma unroll
            for (int kb = 1; kb < 16; kb += 5) {               // five independent products at a time
                v2f t[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) t[i] = cmul1_v(u[kb + i], tw[kb + i - 1]);
#pragma unroll
                for (int i = 0; i < 5; ++i) u[kb + i] = cmul2_v(u[kb + i], tw[kb + i - 1], t[i]);
            }
            FE_STAMP(round * 8 + 2)
            // transpose through the wave's tile
            v2f v[16];
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) x_wr[k1 * (kXRow / 2)] = u[k1];
            lds_order();
#pragma unroll
            for (int q = 0; q < 8; ++q) {                      // lane k1 gathers its row over n2
                const float4 f = x_rd[q];
                v[2 * q] = v2f{f.x, f.y};
                v[2 * q + 1] = v2f{f.z, f.w};
            }
            lds_order();                                       // the tile is rewritten below
            FE_STAMP(round * 8 + 3)
            dft16(v);                                          // over n2 -> k2: v[k2] = Z[k1 + 16 k2]
            FE_STAMP(round * 8 + 4)
#pragma unroll
            for (int k2 = 8; k2 < 16; ++k2) z_wr[16 * (k2 - 8)] = v[k2];   // only bins >= 128 are ever read back as mirrors
            lds_order();
            // ---- real-FFT split: the pair (Z[k], Z[256 - k]) gives X[k] and X[256 - k]; this lane takes its bins
            //      k = k1 + 16 k2 with k2 < 8, the lane holding 16 - k1 takes the other half of the pairs ----
            float* const mrow = s_mag + fl * kMagRow;
            v2f zm[8];
            zm[0] = z_base[zm0];
#pragma unroll
            for (int k2 = 1; k2 < 8; ++k2) zm[k2] = z_base[zm1 - 16 * (k2 - 1)];
            FE_STAMP(round * 8 + 5)
#pragma unroll
            for (int kb = 0; kb < 8; kb += 4) {                // four pairs at a time
                v2f e2[4], o2[4], t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    e2[i] = add_conj(v[kb + i], zm[kb + i]);   // 2 E[k]
                    o2[i] = odd_part(v[kb + i], zm[kb + i]);   // 2 O[k]
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) t[i] = cmul1_v(o2[i], tws[kb + i]);
#pragma unroll
                for (int i = 0; i < 4; ++i) o2[i] = cmul2_v(o2[i], tws[kb + i], t[i]);       // W^k O[k]
                float mp[4], mq[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const v2f p = fma_s(e2[i], kHalf, o2[i]);  // X[k]
                    const v2f q = fms_s(e2[i], kHalf, o2[i]);  // conj(X[256 - k])
                    mp[i] = fmaf(p.y, p.y, p.x * p.x);
                    mq[i] = fmaf(q.y, q.y, q.x * q.x);
                }
                // v_sqrt_f32 (1 ulp): |X| only feeds log(mel + 0.001)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    mp[i] = __builtin_amdgcn_sqrtf(mp[i]);
                    mq[i] = __builtin_amdgcn_sqrtf(mq[i]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k2 = kb + i;
                    mrow[j16 + 16 * k2] = mp[i];
                    if (k2 > 0) mrow[240 - j16 - 16 * (k2 - 1)] = mq[i];     // bins 129..240
                }
            }
            // bin 128 is its own mirror: X[128] = conj(Z[128])
            if (j16 == 0) mrow[128] = __builtin_amdgcn_sqrtf(fmaf(v[8].y, v[8].y, v[8].x * v[8].x));
            lds_order();
            // request the next group's samples: they arrive under the barriers, the mel phase and the output.  (Also
            // behind the last group: the buffer resource answers zero, and an unconditional definition keeps raw[] and
            // win[] from staying live around the whole loop.)
            load_group(raw, group + (int)gridDim.x);
            FE_STAMP(round * 8 + 6)
        }
        FE_STAMP(16)
        __syncthreads();
        FE_STAMP(17)

        // ---------------- mel phase: lane = frame, wave = run of bands ----------------
        // The staged log-mel row of frame f lives in the region of the wave that transformed (and will store) it:
        // that wave may then go on into its next transform, which rewrites only its own region, without a third barrier.
        if (lane < kGroup) {
            const float* mag_row = s_mag + lane * kMagRow;
            float* out_row = s_x + (lane >> 2) * kTile + (lane & 3) * kOutRow;
            switch (wave) {
                case 0: mel_wave<0>(mag_row, out_row, melw); break;
                case 1: mel_wave<1>(mag_row, out_row, melw); break;
                case 2: mel_wave<2>(mag_row, out_row, melw); break;
                case 3: mel_wave<3>(mag_row, out_row, melw); break;
                case 4: mel_wave<4>(mag_row, out_row, melw); break;
                case 5: mel_wave<5>(mag_row, out_row, melw); break;
                case 6: mel_wave<6>(mag_row, out_row, melw); break;
                case 7: mel_wave<7>(mag_row, out_row, melw); break;
                case 8: mel_wave<8>(mag_row, out_row, melw); break;
                case 9: mel_wave<9>(mag_row, out_row, melw); break;
                case 10: mel_wave<10>(mag_row, out_row, melw); break;
                default: mel_wave<11>(mag_row, out_row, melw); break;
            }
        }
        FE_STAMP(18)
        __syncthreads();
        FE_STAMP(19)

        // the next group: its samples were requested at the end of the transform; consume them BEFORE the stores
#pragma unroll
        for (int n1 = 0; n1 < 13; ++n1) win[n1] = raw[n1] * t_hann[16 * n1];

        // ---------------- output: this wave's four rows of 256 bytes ----------------
        // one 16-byte store per lane (lane -> row lane / 16, bands 4 (lane % 16) ..) through a buffer resource sized
        // n_frames rows: rows of a last, partial group fall outside it and are dropped by the bounds check
        {
            const float* src = xw + fq * kOutRow + 4 * j16;
            v4f o = {src[0], src[1], src[2], src[3]};
            const int byte0 = ((group * kGroup + wave * 4 + fq) * BD_MEL_BANDS + 4 * j16) * 4;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, o), out_rsrc, byte0, 0, 0);
            lds_order();                                       // the row reads stay ahead of the next transform's tile writes
        }
        FE_STAMP(20)
    }
}

}  // namespace fe

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
// CPU restatement).  Taps are walked in increasing input index.
//
// resample_kernel: a workgroup owns a tile of consecutive outputs.  It stages the channel mean of the input span
// the tile needs (coalesced, converted once) and the filter in LDS, then every thread walks its outputs' taps out
// of LDS: each input sample is read from HBM once per tile instead of (taps / down) times, and the filter never.
// convert_kernel: the rate-preserving case (up = down = 1, where resample_poly returns its input): channel mean
// and s16 -> f32 only, 16 bytes stored per lane.
// Algorithmic HBM traffic per output sample: channels * sizeof(T) * down / up read + 4 B written.
__device__ __forceinline__ float pcm_to_float(float v) { return v; }
__device__ __forceinline__ float pcm_to_float(short v) { return (float)v * (1.0f / 32768.0f); }   // as libsndfile's float read

__device__ __forceinline__ float stereo_mean(const short* __restrict__ in, long long i) {      // one 4-byte load per frame
    // (l / 32768 + r / 32768) / 2 with every step exact in float32 = (l + r) / 65536: one conversion instead of two
    const int v = reinterpret_cast<const int*>(in)[i];
    return (float)(((v << 16) >> 16) + (v >> 16)) * (1.0f / 65536.0f);
}
__device__ __forceinline__ float stereo_mean(const float* __restrict__ in, long long i) {      // one 8-byte load per frame
    const float2 v = reinterpret_cast<const float2*>(in)[i];
    return (v.x + v.y) * 0.5f;
}

template <typename T>
__device__ __forceinline__ float mono_at(const T* __restrict__ in, long long i, int channels) {
    if (channels == 1) return pcm_to_float(in[i]);
    if (channels == 2) return stereo_mean(in, i);
    float m = 0.0f;
    for (int ch = 0; ch < channels; ++ch) m += pcm_to_float(in[i * channels + ch]);
    return m / (float)channels;
}

constexpr int kRsThreads = 256;
constexpr int kRsPerThread = 8;                               // outputs per thread
constexpr int kRsTile = kRsThreads * kRsPerThread;            // 2048 outputs per workgroup
constexpr int kRsMaxSpan = 8192;                              // input samples staged per tile (32 KB)
constexpr int kRsMaxTaps = 2048;                              // filter taps staged (8 KB: ratios up to 102); longer filters stay in L2

template <typename T, bool TAPS_IN_LDS>
__global__ __launch_bounds__(kRsThreads) void resample_kernel(const T* __restrict__ in, long long n_in, int channels,
                                                              const float* __restrict__ h, int half, int up, int down,
                                                              float* __restrict__ out, long long n_out, int tile) {
    __shared__ float s_x[kRsMaxSpan];
    __shared__ float s_h[TAPS_IN_LDS ? kRsMaxTaps : 1];
    const int tid = threadIdx.x;
    const long long j0 = (long long)blockIdx.x * tile;
    const long long j1 = j0 + tile < n_out ? j0 + tile : n_out;
    // input span of the tile: i in [ceil((j0*down - half) / up), floor(((j1-1)*down + half) / up)], clamped to the signal
    long long lo = j0 * down - half;
    long long i_lo = lo >= 0 ? (lo + up - 1) / up : -((-lo) / up);
    long long i_hi = ((j1 - 1) * down + half) / up;
    if (i_lo < 0) i_lo = 0;
    if (i_hi > n_in - 1) i_hi = n_in - 1;
    const int span = (int)(i_hi - i_lo + 1);
    for (int k = tid; k < span; k += kRsThreads) s_x[k] = mono_at(in, i_lo + k, channels);
    if (TAPS_IN_LDS)
        for (int k = tid; k < 2 * half + 1; k += kRsThreads) s_h[k] = h[k];
    __syncthreads();
    for (int q = 0; q < kRsPerThread; ++q) {
        const long long j = j0 + tid + q * kRsThreads;
        if (j >= j1 || tid + q * kRsThreads >= tile) break;
        const long long c = j * down;                       // position on the up-sampled grid
        long long a = c - half;
        long long ia = a >= 0 ? (a + up - 1) / up : -((-a) / up);
        long long ib = (c + half) / up;
        if (ia < i_lo) ia = i_lo;
        if (ib > i_hi) ib = i_hi;
        int t = (int)(c - ia * up + half);                  // tap of the first input sample; steps down by `up`
        int k = (int)(ia - i_lo);
        const int ke = (int)(ib - i_lo);
        float acc = 0.0f;
        // eight taps' operands are requested together, then accumulated in input order: the same sum as a one-tap loop
        // (one LDS round trip per eight taps instead of per tap)
        for (; k + 7 <= ke; k += 8, t -= 8 * up) {
            float x[8], w[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                x[e] = s_x[k + e];
                w[e] = TAPS_IN_LDS ? s_h[t - e * up] : h[t - e * up];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(x[e], w[e], acc);
        }
        for (; k <= ke; ++k, t -= up) acc = fmaf(s_x[k], TAPS_IN_LDS ? s_h[t] : h[t], acc);
        out[j] = acc;
    }
}

// decimate_kernel: integer decimation (up = 1: 48 kHz -> 16 kHz, 32 kHz -> 16 kHz), the ratios field recorders produce.
// With up = 1 every output walks the same 20 * DOWN + 1 taps, so the filter is wave-uniform: it sits in scalar register
// pairs and an FMA reads only its samples.  A thread owns kDecR = 7 CONSECUTIVE outputs as three pairs and a single: the
// outputs r and r + 1 of a pair meet tap t at samples i and i + DOWN, which one ds_read2_b32 delivers as a register pair, so
// a pair advances by one v_pk_fma_f32 (tap broadcast through op_sel) per tap and a sample pair serves all three pairs of
// its step: 76 LDS reads, 183 packed and 61 plain FMAs for seven outputs (the general kernel: 2 LDS reads per FMA).  Every
// output still accumulates its taps in increasing input index; samples outside the signal are staged as zeros.  Outputs
// leave through LDS as whole rows.
constexpr int kDecR = 7;                                      // outputs per thread; 7 * DOWN is odd for DOWN = 3: conflict-free reads

template <int SEL>      // acc += x * tap[SEL] in both halves; tap = a pair of wave-uniform filter taps in scalar registers
__device__ __forceinline__ fe::v2f dec_pk_fma(fe::v2f x, fe::v2f tap, fe::v2f acc) {
    if constexpr (SEL == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(x), "s"(tap));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(x), "s"(tap));
    return acc;
}

template <int O0, int O1>      // (x[O0], x[O1]) as one register pair; the caller waits (lgkmcnt) before using it
__device__ __forceinline__ fe::v2f dec_read2(unsigned addr) {
    fe::v2f v;
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(O0), "n"(O1) : "memory");
    return v;
}
__device__ __forceinline__ fe::v2f dec_landed(fe::v2f v) {
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ fe::v2f dec_tap_landed(fe::v2f v) {     // the taps are in their scalar registers from here on
    asm volatile("" : "+s"(v));
    return v;
}
template <int N>
__device__ __forceinline__ void dec_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T, int DOWN>
__global__ __launch_bounds__(256) void decimate_kernel(const T* __restrict__ in, long long n_in, int channels,
                                                       const float* __restrict__ h, float* __restrict__ out, long long n_out) {
    using fe::v2f;
    constexpr int HALF = 10 * DOWN, TAPS = 2 * HALF + 1;
    constexpr int TILE = 256 * kDecR;                          // outputs per workgroup
    constexpr int NX = (kDecR - 1) * DOWN + TAPS;              // samples under a thread's kDecR filters
    constexpr int SPAN = (TILE - 1) * DOWN + TAPS;             // samples under the tile's filters
    constexpr int NP = (TAPS + 1) / 2;                         // tap pairs
    static_assert(kDecR == 7 && SPAN >= TILE, "three pairs and a single; the outputs are staged in the sample buffer");
    __shared__ __attribute__((aligned(16))) float s_x[SPAN + 1];
    const int tid = threadIdx.x;
    const long long j0 = (long long)blockIdx.x * TILE;
    const long long i0 = j0 * DOWN - HALF;                     // input index of s_x[0]
    for (int k = tid; k < SPAN; k += 256) {
        const long long i = i0 + k;
        s_x[k] = i >= 0 && i < n_in ? mono_at(in, i, channels) : 0.0f;
    }
    v2f tap[NP];                                               // (h[2 m], h[2 m + 1]); the filter has an odd number of taps
    fe::static_for<0, NP>([&](auto mi) {
        constexpr int m = decltype(mi)::value;
        if constexpr (2 * m + 1 < TAPS) tap[m] = *reinterpret_cast<const v2f*>(h + 2 * m);
        else tap[m] = v2f{h[2 * m], 0.0f};
    });
    __syncthreads();
    fe::static_for<0, NP>([&](auto mi) { tap[decltype(mi)::value] = dec_tap_landed(tap[decltype(mi)::value]); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned xs = fe::lds_addr(s_x + tid * (kDecR * DOWN));
    v2f acc[3] = {v2f{0.f, 0.f}, v2f{0.f, 0.f}, v2f{0.f, 0.f}};
    float acc6 = 0.0f;
    // sample pairs (x[i], x[i + DOWN]) through a ring, requested DEPTH steps ahead; LDS operations complete in order, so
    // "pair i has landed" is "at most DEPTH (or what is left) requests are in flight"
    constexpr int STEPS = NX - DOWN, DEPTH = 6, RING = 8;
    v2f ring[RING];
    fe::static_for<0, DEPTH>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        ring[i % RING] = dec_read2<i, i + DOWN>(xs);
    });
    fe::static_for<0, STEPS>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        if constexpr (i + DEPTH < STEPS) ring[(i + DEPTH) % RING] = dec_read2<i + DEPTH, i + DEPTH + DOWN>(xs);
        dec_wait<(STEPS - 1 - i < DEPTH ? STEPS - 1 - i : DEPTH)>();
        const v2f xp = dec_landed(ring[i % RING]);
        fe::static_for<0, 3>([&](auto pi) {
            constexpr int p = decltype(pi)::value;
            constexpr int d = i - 2 * p * DOWN;                // position under the filter of the pair's first output
            if constexpr (d >= 0 && d <= 2 * HALF) {
                constexpr int t = 2 * HALF - d;
                acc[p] = dec_pk_fma<t & 1>(xp, tap[t >> 1], acc[p]);
            }
        });
        // output 6 meets sample i + DOWN at this step: the upper half of the pair
        constexpr int d6 = i + DOWN - 6 * DOWN;
        if constexpr (d6 >= 0 && d6 <= 2 * HALF) {
            constexpr int t = 2 * HALF - d6;
            acc6 = fmaf(xp.y, (t & 1) ? tap[t >> 1].y : tap[t >> 1].x, acc6);
        }
    });
    __syncthreads();                                           // everyone is done reading the samples
    s_x[tid * kDecR + 0] = acc[0].x;
    s_x[tid * kDecR + 1] = acc[0].y;
    s_x[tid * kDecR + 2] = acc[1].x;
    s_x[tid * kDecR + 3] = acc[1].y;
    s_x[tid * kDecR + 4] = acc[2].x;
    s_x[tid * kDecR + 5] = acc[2].y;
    s_x[tid * kDecR + 6] = acc6;
    __syncthreads();
    if (j0 + TILE <= n_out) {
        for (int k = tid; k < TILE / 4; k += 256)
            reinterpret_cast<float4*>(out + j0)[k] = reinterpret_cast<const float4*>(s_x)[k];
    } else {
        for (int k = tid; k < TILE && j0 + k < n_out; k += 256) out[j0 + k] = s_x[k];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void convert_kernel(const T* __restrict__ in, long long n, int channels,
                                                      float* __restrict__ out) {
    const long long quads = (n + 3) / 4;
    for (long long q = blockIdx.x * 256LL + threadIdx.x; q < quads; q += gridDim.x * 256LL) {
        const long long i = 4 * q;
        if (i + 3 < n) {
            float4 v;
            v.x = mono_at(in, i, channels);
            v.y = mono_at(in, i + 1, channels);
            v.z = mono_at(in, i + 2, channels);
            v.w = mono_at(in, i + 3, channels);
            *reinterpret_cast<float4*>(out + i) = v;
        } else {
            for (long long k = i; k < n; ++k) out[k] = mono_at(in, k, channels);
        }
    }
}

}  // namespace

void launch_resample(const void* in, bool s16, int64_t n_in, int channels, const float* taps, int half, int up,
                     int down, float* out, int64_t n_out, hipStream_t stream) {
    if (n_out <= 0) return;
    if (up == 1 && down == 1) {                               // same rate: channel mean + conversion only
        const int64_t blocks = ((n_out + 3) / 4 + 255) / 256;
        const int grid = (int)(blocks < 65536 ? blocks : 65536);
        if (s16) hipLaunchKernelGGL(convert_kernel<short>, dim3(grid), dim3(256), 0, stream, static_cast<const short*>(in), (long long)n_out, channels, out);
        else hipLaunchKernelGGL(convert_kernel<float>, dim3(grid), dim3(256), 0, stream, static_cast<const float*>(in), (long long)n_out, channels, out);
        return;
    }
    if (up == 1 && (down == 2 || down == 3) && half == 10 * down) {      // integer decimation: filter in scalar registers
        const int64_t grid = (n_out + 256 * kDecR - 1) / (256 * kDecR);
#define BD_DEC_LAUNCH(T, D)                                                                                     \
    hipLaunchKernelGGL((decimate_kernel<T, D>), dim3((unsigned)grid), dim3(256), 0, stream, static_cast<const T*>(in), \
                       (long long)n_in, channels, taps, out, (long long)n_out)
        if (s16) { if (down == 2) BD_DEC_LAUNCH(short, 2); else BD_DEC_LAUNCH(short, 3); }
        else { if (down == 2) BD_DEC_LAUNCH(float, 2); else BD_DEC_LAUNCH(float, 3); }
#undef BD_DEC_LAUNCH
        return;
    }
    // outputs per workgroup: as many as the staged input span allows (span = tile * down / up + 2 * half / up + 2)
    int64_t tile = ((int64_t)(kRsMaxSpan - 4 - 2 * (int64_t)half / up) * up) / down;
    if (tile > kRsTile) tile = kRsTile;
    if (tile < 1) tile = 1;
    const int64_t grid = (n_out + tile - 1) / tile;
    const bool lds_taps = 2 * half + 1 <= kRsMaxTaps;
#define BD_RS_LAUNCH(T, L)                                                                                      \
    hipLaunchKernelGGL((resample_kernel<T, L>), dim3((unsigned)grid), dim3(kRsThreads), 0, stream,             \
                       static_cast<const T*>(in), (long long)n_in, channels, taps, half, up, down, out,         \
                       (long long)n_out, (int)tile)
    if (s16) { if (lds_taps) BD_RS_LAUNCH(short, true); else BD_RS_LAUNCH(short, false); }
    else { if (lds_taps) BD_RS_LAUNCH(float, true); else BD_RS_LAUNCH(float, false); }
#undef BD_RS_LAUNCH
}

// resample_kernel stages the input span of a tile (tile * down / up + 2 * half / up + a few samples) in s_x[kRsMaxSpan]: a
// filter so long that even a one-output tile does not fit cannot run on it (launch_resample clamps the tile to 1 and the
// kernel would stage past the array).  With the 569-tap-class filter that is down / up > ~43, e.g. 768 kHz -> 16 kHz.
bool resample_span_fits(int half, int up, int down) {
    return 2 * (int64_t)half / up + (int64_t)down / up + 4 <= kRsMaxSpan;
}

void launch_logmel(const float* pcm, int64_t n_valid, int64_t n_frames, float* logmel,
                   const FeTables* tables, hipStream_t stream) {
    if (n_frames <= 0) return;
    const int64_t groups = (n_frames + fe::kGroup - 1) / fe::kGroup;
    const int grid = (int)(groups < 256 ? groups : 256);      // one 156 KB, 12-wave workgroup per CU
#ifdef BD_FE_TRACE
    hipLaunchKernelGGL(fe::logmel_kernel, dim3(grid), dim3(fe::kThreads), 0, stream, pcm, (int)n_valid, (int)n_frames,
                       logmel, tables, (unsigned long long*)nullptr);
#else
    hipLaunchKernelGGL(fe::logmel_kernel, dim3(grid), dim3(fe::kThreads), 0, stream, pcm, (int)n_valid, (int)n_frames,
                       logmel, tables);
#endif
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
