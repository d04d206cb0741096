// What v_mfma_f32_32x32x2_f32 sustains: waves per SIMD x independent accumulators, operands in registers, nothing else in the loop.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_mfma_f32 tools/ubench_mfma_f32.hip && /tmp/ubench_mfma_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        asm volatile("" : "+v"(a), "+v"(b));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int waves_per_simd, float* out) {
    const int threads = 64 * 4 * waves_per_simd, iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, 10, 1.f, 2.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 256.0 * threads / 64 * iters * 4 * NACC * (32.0 * 32 * 2 * 2);
    printf("%d waves/SIMD, %d accumulators: %.1f TFLOP/s (%.3f ms)\n", waves_per_simd, NACC, flop / ms / 1e9, ms);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4);
    for (int w : {1, 2, 4}) {
        run<1>(w, out);
        run<2>(w, out);
        run<4>(w, out);
        run<6>(w, out);
    }
    return 0;
}
