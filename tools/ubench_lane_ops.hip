// Throughput of the vector / cross-lane / LDS instructions the front end is built from, on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench tools/ubench_lane_ops.hip && /tmp/ubench
// Every kernel runs ITERS x 32 copies of one instruction on eight independent registers per wave, one
// workgroup per CU with W waves per SIMD; the table gives shader cycles per wave-instruction per SIMD
// (s_memtime around the loop, median over waves, divided by W).  Not part of the product.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

constexpr int ITERS = 2000;

template <int OP>
__global__ __launch_bounds__(1024) void bench(unsigned long long* cycles, float* sink) {
    __shared__ float lds[8192];
    const int tid = threadIdx.x;
    float a[8], b = 1.0001f + tid * 1e-7f, c = 1e-6f;
    double d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = 1.0f + 0.01f * i + tid * 1e-6f;
        d[i] = __hiloint2double(__float_as_int(a[i]), __float_as_int(a[i] + 0.5f));
    }
    const double bd = __hiloint2double(__float_as_int(b), __float_as_int(b));
    for (int i = tid; i < 8192; i += blockDim.x) lds[i] = i;
    __syncthreads();
    int addr_b32 = (tid & 63) * 4 + (tid >> 6) * 256;         // conflict-free rows
    int addr_b64 = (tid & 63) * 8 + (tid >> 6) * 512;
    int addr_b128 = (tid & 63) * 16 + (tid >> 6) * 1024;
    int bperm = ((tid * 7 + 3) & 63) * 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            if constexpr (OP == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if constexpr (OP == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i]) : "v"(bd));
                REP8(X)
#undef X
            } else if constexpr (OP == 2) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(bd));
                REP8(X)
#undef X
            } else if constexpr (OP == 3) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(bd));
                REP8(X)
#undef X
            } else if constexpr (OP == 4) {
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[0]), "+v"(a[1]));
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[2]), "+v"(a[3]));
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[4]), "+v"(a[5]));
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[6]), "+v"(a[7]));
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[0]), "+v"(a[2]));
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[1]), "+v"(a[3]));
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[4]), "+v"(a[6]));
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[5]), "+v"(a[7]));
            } else if constexpr (OP == 5) {
                asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[0]), "+v"(a[1]));
                asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[2]), "+v"(a[3]));
                asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[4]), "+v"(a[5]));
                asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[6]), "+v"(a[7]));
                asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[0]), "+v"(a[2]));
                asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[1]), "+v"(a[3]));
                asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[4]), "+v"(a[6]));
                asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[5]), "+v"(a[7]));
            } else if constexpr (OP == 6) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if constexpr (OP == 7) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_ror:4 row_mask:0xf bank_mask:0xa" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if constexpr (OP == 8) {
#define X(i) asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if constexpr (OP == 9) {
#define X(i) asm volatile("v_cndmask_b32_dpp %0, %0, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 7]) : "vcc");
                REP8(X)
#undef X
            } else if constexpr (OP == 10) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
                REP8(X)
#undef X
            } else if constexpr (OP == 11) {
#define X(i) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a[i]) : "v"(bperm));
                REP8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if constexpr (OP == 12) {
#define X(i) asm volatile("ds_swizzle_b32 %0, %0 offset:swizzle(BITMASK_PERM, \"0101p\")" : "+v"(a[i]));
                REP8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if constexpr (OP == 13) {
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
                REP8(X)
#undef X
            } else if constexpr (OP == 14) {
#define X(i) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
                REP8(X)
#undef X
            } else if constexpr (OP == 15) {
#define X(i) asm volatile("ds_read_b64 %0, %1 offset:" #i "*2048" : "=v"(d[i]) : "v"(addr_b64));
                REP8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if constexpr (OP == 16) {
#define X(i) asm volatile("ds_write_b64 %1, %0 offset:" #i "*2048" :: "v"(d[i]), "v"(addr_b64) : "memory");
                REP8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if constexpr (OP == 17) {
#define X(i) asm volatile("ds_write_b32 %1, %0 offset:" #i "*1024" :: "v"(a[i]), "v"(addr_b32) : "memory");
                REP8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if constexpr (OP == 18) {
#define X(i) asm volatile("ds_read_b32 %0, %1 offset:" #i "*1024" : "=v"(a[i]) : "v"(addr_b32));
                REP8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if constexpr (OP == 19) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                REP8(X)
#undef X
            } else if constexpr (OP == 20) {
#define X(i) asm volatile("v_fmac_f32 %0, %1, s20" : "+v"(a[i]) : "v"(b) : "s20");
                REP8(X)
#undef X
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + (float)d[i];
    if (s == 12345.678f) sink[0] = s + lds[addr_b128 & 8191];
    if ((tid & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (tid >> 6)] = t1 - t0;
}

template <int OP>
void run(const char* name) {
    unsigned long long* d_c;
    float* d_s;
    hipMalloc(&d_c, 256 * 16 * sizeof(unsigned long long));
    hipMalloc(&d_s, 64);
    printf("%-44s", name);
    for (int w : {1, 2, 4}) {
        hipLaunchKernelGGL(bench<OP>, dim3(256), dim3(256 * w), 0, 0, d_c, d_s);
        hipLaunchKernelGGL(bench<OP>, dim3(256), dim3(256 * w), 0, 0, d_c, d_s);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 4 * w);
        hipMemcpy(h.data(), d_c, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2];
        printf("  W=%d: %6.2f", w, med / (ITERS * 32.0) / w);
    }
    printf("   cycles / wave-instruction / SIMD\n");
    hipFree(d_c);
    hipFree(d_s);
}

int main() {
    run<0>("v_fma_f32");
    run<19>("v_mul_f32");
    run<20>("v_fmac_f32 (sgpr operand)");
    run<1>("v_pk_fma_f32");
    run<2>("v_pk_add_f32");
    run<3>("v_pk_mul_f32");
    run<10>("v_cndmask_b32");
    run<4>("v_permlane32_swap_b32");
    run<5>("v_permlane16_swap_b32");
    run<6>("v_mov_b32_dpp quad_perm");
    run<7>("v_mov_b32_dpp row_ror:4 bank_mask");
    run<8>("v_add_f32_dpp quad_perm");
    run<9>("v_cndmask_b32_dpp quad_perm");
    run<13>("v_sqrt_f32");
    run<14>("v_log_f32");
    run<11>("ds_bpermute_b32 (x4 SIMDs share the LDS)");
    run<12>("ds_swizzle_b32");
    run<15>("ds_read_b64");
    run<16>("ds_write_b64");
    run<17>("ds_write_b32");
    run<18>("ds_read_b32");
    return 0;
}
