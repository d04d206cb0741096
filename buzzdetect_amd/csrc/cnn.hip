// YAMNet (MobileNetV1) body on gfx950: embedders/yamnet/yamnet.py:36-106 with the BatchNorms
// (yamnet.py:26-33, scale=False, eps=1e-4) folded into the convolution weights at load time.
// All activations are NHWC float32, exactly the reference's layout.
//
//   conv1_kernel      Conv2D 3x3 s2 SAME 1->32 + bias + ReLU, reading log-mel patches in place
//                     (window w = frames [w*step, w*step+96) of the [T,64] spectrogram: the
//                     tf.signal.frame / Reshape of features.py:72-76, yamnet.py:98-100 is index math)
//   depthwise_kernel  DepthwiseConv2D 3x3 s1|s2 SAME + bias + ReLU
//   pointwise_kernel  Conv2D 1x1 + bias + ReLU as C[M,N] = A[M,K] * Wt[N,K]^T on the f32 matrix
//                     cores (v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulate)
//   pool_head_kernel  GlobalAveragePooling2D over 3x2 + Dense(1024 -> n_classes)
//
// TF "SAME" for an even extent with stride 2 pads 0 before / 1 after; stride 1 pads 1 / 1.
#include "bd_internal.h"

namespace bd {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// --------------------------------------------------------------------------- conv1
constexpr int kC1Rows = 4;   // output rows per workgroup

__global__ __launch_bounds__(256) void conv1_kernel(const float* __restrict__ logmel, int patch_step,
                                                    const float* __restrict__ w9x32,
                                                    const float* __restrict__ b32,
                                                    float* __restrict__ out) {
    // grid: (48 / kC1Rows, windows); thread = (ow = tid >> 3, c4 = tid & 7)
    const int tid = threadIdx.x;
    const int c4 = tid & 7;
    const int ow = tid >> 3;
    const int win = blockIdx.y;
    const float* patch = logmel + (size_t)win * patch_step * BD_MEL_BANDS;

    float4 wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t] = reinterpret_cast<const float4*>(w9x32 + t * 32)[c4];
    const float4 bias = reinterpret_cast<const float4*>(b32)[c4];

#pragma unroll
    for (int rr = 0; rr < kC1Rows; ++rr) {
        const int oh = blockIdx.x * kC1Rows + rr;
        float4 acc = bias;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = 2 * oh + kh;                      // pad_top = 0
            if (ih < BD_PATCH_FRAMES) {
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int iw = 2 * ow + kw;              // pad_left = 0
                    const float v = iw < BD_MEL_BANDS ? patch[ih * BD_MEL_BANDS + iw] : 0.0f;
                    const float4 w = wt[kh * 3 + kw];
                    acc.x = fmaf(v, w.x, acc.x);
                    acc.y = fmaf(v, w.y, acc.y);
                    acc.z = fmaf(v, w.z, acc.z);
                    acc.w = fmaf(v, w.w, acc.w);
                }
            }
        }
        acc.x = fmaxf(acc.x, 0.0f);
        acc.y = fmaxf(acc.y, 0.0f);
        acc.z = fmaxf(acc.z, 0.0f);
        acc.w = fmaxf(acc.w, 0.0f);
        reinterpret_cast<float4*>(out + (((size_t)win * 48 + oh) * 32 + ow) * 32)[c4] = acc;
    }
}

// --------------------------------------------------------------------------- depthwise
__global__ __launch_bounds__(256) void depthwise_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        const float* __restrict__ w9xc,
                                                        const float* __restrict__ bias, int windows, int H,
                                                        int W, int C, int OH, int OW, int stride, int pad) {
    const int c4n = C >> 2;
    const long long total = (long long)windows * OH * OW * c4n;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
        const int c4 = (int)(i % c4n);
        long long t = i / c4n;
        const int ow = (int)(t % OW);
        t /= OW;
        const int oh = (int)(t % OH);
        const long long n = t / OH;
        const float* src = in + (size_t)n * H * W * C;
        float4 acc = reinterpret_cast<const float4*>(bias)[c4];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * stride + kh - pad;
            if (ih < 0 || ih >= H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * stride + kw - pad;
                if (iw < 0 || iw >= W) continue;
                const float4 v = reinterpret_cast<const float4*>(src + ((size_t)ih * W + iw) * C)[c4];
                const float4 w = reinterpret_cast<const float4*>(w9xc + (kh * 3 + kw) * C)[c4];
                acc.x = fmaf(v.x, w.x, acc.x);
                acc.y = fmaf(v.y, w.y, acc.y);
                acc.z = fmaf(v.z, w.z, acc.z);
                acc.w = fmaf(v.w, w.w, acc.w);
            }
        }
        acc.x = fmaxf(acc.x, 0.0f);
        acc.y = fmaxf(acc.y, 0.0f);
        acc.z = fmaxf(acc.z, 0.0f);
        acc.w = fmaxf(acc.w, 0.0f);
        reinterpret_cast<float4*>(out)[i] = acc;
    }
}

// --------------------------------------------------------------------------- pointwise GEMM
// C[m][n] = relu(sum_k A[m][k] * Wt[n][k] + bias[n]).  A = NHWC activations flattened to
// [rows = windows*H*W][K = Cin]; Wt = folded kernel stored [Cout][Cin] so that both operands
// have K contiguous and one ds_read_b128 feeds four MFMA k-steps.
//
// Workgroup = 4 waves (2 x 2) on a BM x BN tile, BK = 32 per LDS stage, register-staged
// double buffering (global_load_dwordx4 of tile t+1 in flight while tile t is on the MFMAs).
// v_mfma_f32_32x32x2_f32 operand map: lane l supplies A[i = l & 31][k = l >> 5] and
// B[k = l >> 5][j = l & 31]; the k index of a step is arbitrary as long as both operands
// agree, so lane-half h takes k = 8*s + 4*h + j for the j-th MFMA of super-step s.
constexpr int kBK = 32;
constexpr int kLds = kBK + 4;   // row stride in floats: 144 B = odd multiple of 16 B -> conflict-free b128

template <int BM, int BN>
__global__ __launch_bounds__(256) void pointwise_kernel(const float* __restrict__ A,
                                                        const float* __restrict__ Wt,
                                                        const float* __restrict__ bias,
                                                        float* __restrict__ C, long long M, int N, int K) {
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int LA = BM / 32, LB = BN / 32;   // float4 global loads per thread per stage
    __shared__ __attribute__((aligned(16))) float As[2][BM * kLds];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * kLds];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int n_tiles = N / BN;
    const long long tile_m = blockIdx.x / n_tiles;
    const int tile_n = blockIdx.x % n_tiles;
    const long long m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    const int lrow = tid >> 3;   // 0..31
    const int lc4 = tid & 7;     // float4 column within the 32-wide k slab

    float4 ra[LA], rb[LB];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const long long m = m0 + lrow + 32 * i;
            ra[i] = m < M ? reinterpret_cast<const float4*>(A + (size_t)m * K + k0)[lc4]
                          : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int n = n0 + lrow + 32 * i;
            rb[i] = reinterpret_cast<const float4*>(Wt + (size_t)n * K + k0)[lc4];
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LA; ++i)
            *reinterpret_cast<float4*>(&As[buf][(lrow + 32 * i) * kLds + lc4 * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < LB; ++i)
            *reinterpret_cast<float4*>(&Bs[buf][(lrow + 32 * i) * kLds + lc4 * 4]) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nk = K / kBK;
    gload(0);
    lstore(0);
    __syncthreads();

    const int frow = lane & 31;
    const int fk = (lane >> 5) * 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * kBK);
        const float* as = &As[buf][(wr * WM + frow) * kLds + fk];
        const float* bs = &Bs[buf][(wc * WN + frow) * kLds + fk];
#pragma unroll
        for (int s = 0; s < kBK / 8; ++s) {
            float4 av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const float4*>(as + i * 32 * kLds + s * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const float4*>(bs + j * 32 * kLds + s * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // epilogue: C/D map of the 32x32 tile: col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5)
    const int half = lane >> 5;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wc * WN + j * 32 + frow;
        const float b = bias[n];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const long long mb = m0 + wr * WM + i * 32 + 4 * half;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = mb + (r & 3) + 8 * (r >> 2);
                if (m < M) C[(size_t)m * N + n] = fmaxf(acc[i][j][r] + b, 0.0f);
            }
        }
    }
}

// --------------------------------------------------------------------------- pool + head
__global__ __launch_bounds__(256) void pool_head_kernel(const float* __restrict__ act,
                                                        const float* __restrict__ head_wt,
                                                        const float* __restrict__ head_b, int n_classes,
                                                        float* __restrict__ emb, float* __restrict__ logits) {
    // one workgroup per window; act = [window][6][1024]
    __shared__ float s_part[4][BD_MAX_CLASSES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const size_t win = blockIdx.x;
    const float4* src = reinterpret_cast<const float4*>(act + win * 6 * BD_EMBEDDING_SIZE);
    float4 s = src[tid];
#pragma unroll
    for (int p = 1; p < 6; ++p) {
        const float4 v = src[p * (BD_EMBEDDING_SIZE / 4) + tid];
        s.x += v.x;
        s.y += v.y;
        s.z += v.z;
        s.w += v.w;
    }
    s.x /= 6.0f;
    s.y /= 6.0f;
    s.z /= 6.0f;
    s.w /= 6.0f;
    if (emb) reinterpret_cast<float4*>(emb + win * BD_EMBEDDING_SIZE)[tid] = s;
    if (!logits) return;

    for (int c = 0; c < n_classes; ++c) {
        const float4 w = reinterpret_cast<const float4*>(head_wt + (size_t)c * BD_EMBEDDING_SIZE)[tid];
        float p = fmaf(s.x, w.x, fmaf(s.y, w.y, fmaf(s.z, w.z, s.w * w.w)));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
        if (lane == 0) s_part[wave][c] = p;
    }
    __syncthreads();
    if (tid < n_classes)
        logits[win * n_classes + tid] =
            ((s_part[0][tid] + s_part[1][tid]) + (s_part[2][tid] + s_part[3][tid])) + head_b[tid];
}

}  // namespace

void launch_conv1(const float* logmel, int patch_step, int windows, const float* w9x32, const float* b32,
                  float* out, hipStream_t stream) {
    if (windows <= 0) return;
    hipLaunchKernelGGL(conv1_kernel, dim3(48 / kC1Rows, windows), dim3(256), 0, stream, logmel, patch_step,
                       w9x32, b32, out);
}

void launch_depthwise(const float* in, float* out, int windows, const SepLayer& L, hipStream_t stream) {
    if (windows <= 0) return;
    const long long total = (long long)windows * L.h_out * L.w_out * (L.cin / 4);
    const long long blocks = (total + 255) / 256;
    const int grid = (int)(blocks < (1 << 20) ? blocks : (1 << 20));
    const int pad = L.stride == 1 ? 1 : 0;   // TF SAME, even extents (yamnet: 48x32 ... 6x4)
    hipLaunchKernelGGL(depthwise_kernel, dim3(grid), dim3(256), 0, stream, in, out, L.dw_w, L.dw_b, windows,
                       L.h_in, L.w_in, L.cin, L.h_out, L.w_out, L.stride, pad);
}

void launch_pointwise(const float* in, float* out, int64_t rows, const SepLayer& L, hipStream_t stream) {
    if (rows <= 0) return;
    if (L.cout % 128 == 0) {
        const long long tiles = ((rows + 127) / 128) * (L.cout / 128);
        hipLaunchKernelGGL((pointwise_kernel<128, 128>), dim3((unsigned)tiles), dim3(256), 0, stream, in,
                           L.pw_wt, L.pw_b, out, (long long)rows, L.cout, L.cin);
    } else {
        const long long tiles = ((rows + 127) / 128) * (L.cout / 64);
        hipLaunchKernelGGL((pointwise_kernel<128, 64>), dim3((unsigned)tiles), dim3(256), 0, stream, in,
                           L.pw_wt, L.pw_b, out, (long long)rows, L.cout, L.cin);
    }
}

void launch_pool_head(const float* act, int windows, const float* head_wt, const float* head_b,
                      int n_classes, float* emb, float* logits, hipStream_t stream) {
    if (windows <= 0) return;
    hipLaunchKernelGGL(pool_head_kernel, dim3(windows), dim3(256), 0, stream, act, head_wt, head_b,
                       n_classes, emb, logits);
}

}  // namespace bd
