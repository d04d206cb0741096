// What the board's power limit leaves of the paper peaks: sustained rate, shader clock and board power (hwmon) of
//   mfma      back-to-back v_mfma_f32_32x32x16_f16 on every SIMD (1 and 2 waves per SIMD), and the same at reduced duty
//   valu      v_pk_fma_f32 / v_fma_f32 streams
//   lds       ds_read_b128 streams
//   l2        every CU streaming the same L2-resident 2 MB
//   copy      a float4 copy of 2 x 64 MiB (Infinity-Cache resident) and of 2 x 1 GiB (HBM)
// each held for a few seconds.  Not part of the product; evidence for DESIGN.md's "power-bound" section.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_power tools/ubench_power.hip -lpthread && tools/ubench_power [seconds]
#include <hip/hip_runtime.h>
#include <dirent.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                               \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

// GAP: s_nop-filled idle issue slots after every MFMA (0 = back to back)
template <int GAP>
__global__ __launch_bounds__(512) void mfma_kernel(float* sink, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(0.001f * (threadIdx.x + i));
        b[i] = (_Float16)(0.002f * (threadIdx.x ^ i));
    }
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
                if constexpr (GAP > 0) {
#pragma unroll
                    for (int g = 0; g < GAP; ++g) asm volatile("s_nop 15");      // 16 idle cycles each
                }
            }
        }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.678f) sink[0] = s;
}

// the same FLOP per iteration on v_mfma_f32_16x16x32_f16 (sixteen accumulators of four registers)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void mfma16_kernel(float* sink, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(0.001f * (threadIdx.x + i));
        b[i] = (_Float16)(0.002f * (threadIdx.x ^ i));
    }
    f32x4 acc[16];
    for (int j = 0; j < 16; ++j)
        for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
        }
    }
    float s = 0;
    for (int j = 0; j < 16; ++j)
        for (int r = 0; r < 4; ++r) s += acc[j][r];
    if (s == 12345.678f) sink[0] = s;
}

template <bool PACKED>
__global__ __launch_bounds__(512) void valu_kernel(float* sink, int iters) {
    v2f x[8], m = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
    for (int i = 0; i < 8; ++i) x[i] = v2f{1.0f + threadIdx.x * 1e-6f, 1.0f + i * 1e-3f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (PACKED) {
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(m), "v"(c));
                } else {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i].x) : "v"(m.x), "v"(c.x));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i].y) : "v"(m.y), "v"(c.y));
                }
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
    if (s == 12345.678f) sink[0] = s;
}

__global__ __launch_bounds__(512) void lds_kernel(float* sink, int iters) {
    __shared__ float4 buf[4096];      // 64 KB
    for (int i = threadIdx.x; i < 4096; i += 512) buf[i] = float4{1.f * i, 2.f, 3.f, 4.f};
    __syncthreads();
    float4 s = {0, 0, 0, 0};
    unsigned at = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 v = buf[(at + 512 * u) & 4095];
            s.x += v.x;
            s.y += v.y;
            s.z += v.z;
            s.w += v.w;
        }
        at += 64;
    }
    if (s.x + s.y + s.z + s.w == 12345.678f) sink[0] = s.x;
}

// every workgroup streams the same `n` float4 (2 MB: resident in each XCD's L2, far too large for the 32 KB L1) `iters` times
__global__ __launch_bounds__(512) void l2_kernel(const float4* __restrict__ in, float* sink, int n, int iters) {
    float4 s = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < n; i += 512 * 4) {
            const float4 a = in[i], b = in[i + 512], c = in[i + 1024], d = in[i + 1536];
            s.x += a.x + b.x + c.x + d.x;
            s.y += a.y + b.y + c.y + d.y;
            s.z += a.z + b.z + c.z + d.z;
            s.w += a.w + b.w + c.w + d.w;
        }
    }
    if (s.x + s.y + s.z + s.w == 12345.678f) sink[0] = s.x;
}

__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

static std::string hwmon_dir(int dev) {
    char bus[64] = {0};
    CHECK(hipDeviceGetPCIBusId(bus, sizeof bus, dev));
    for (char* p = bus; *p; ++p) *p = (char)tolower(*p);
    DIR* d = opendir("/sys/class/drm");
    if (!d) return "";
    std::string found;
    while (dirent* e = readdir(d)) {
        if (strncmp(e->d_name, "card", 4) != 0 || strchr(e->d_name, '-')) continue;
        const std::string dev_link = std::string("/sys/class/drm/") + e->d_name + "/device";
        char real[512];
        const ssize_t n = readlink(dev_link.c_str(), real, sizeof real - 1);
        if (n <= 0) continue;
        real[n] = 0;
        if (!strstr(real, bus)) continue;
        const std::string hm = dev_link + "/hwmon";
        DIR* h = opendir(hm.c_str());
        if (!h) continue;
        while (dirent* he = readdir(h))
            if (strncmp(he->d_name, "hwmon", 5) == 0) found = hm + "/" + he->d_name;
        closedir(h);
    }
    closedir(d);
    return found;
}

static long read_long(const std::string& path) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return -1;
    long v = -1;
    if (fscanf(f, "%ld", &v) != 1) v = -1;
    fclose(f);
    return v;
}

struct Reading {
    double seconds, mhz, watts;
    long launches;
};

// keeps `launch` going for `seconds`; clock and power are averaged over the second half
template <typename F>
static Reading hold(const std::string& hw, double seconds, F&& launch) {
    std::atomic<bool> stop{false};
    std::vector<std::pair<double, std::pair<long, long>>> samples;
    const auto t0 = std::chrono::steady_clock::now();
    auto now = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    std::thread poll([&] {
        while (!stop.load()) {
            samples.push_back({now(), {read_long(hw + "/freq1_input"), read_long(hw + "/power1_input")}});
            usleep(10000);
        }
    });
    long n = 0;
    hipEvent_t ev[2];
    CHECK(hipEventCreate(&ev[0]));
    CHECK(hipEventCreate(&ev[1]));
    while (now() < seconds) {
        for (int i = 0; i < 8; ++i) launch();
        n += 8;
        CHECK(hipEventRecord(ev[n / 8 % 2], 0));
        CHECK(hipEventSynchronize(ev[(n / 8 + 1) % 2]));      // at most ~16 launches queued
    }
    CHECK(hipDeviceSynchronize());
    const double t1 = now();
    stop = true;
    poll.join();
    double mhz = 0, w = 0;
    int k = 0;
    for (auto& s : samples)
        if (s.first > 0.5 * t1 && s.first < t1 && s.second.first > 0 && s.second.second > 0) {
            mhz += s.second.first / 1e6;
            w += s.second.second / 1e6;
            ++k;
        }
    return {t1, k ? mhz / k : 0, k ? w / k : 0, n};
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    CHECK(hipSetDevice(0));
    const std::string hw = hwmon_dir(0);
    if (hw.empty()) {
        fprintf(stderr, "no hwmon directory for device 0\n");
        return 1;
    }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* sink;
    CHECK(hipMalloc(&sink, 256));
    printf("device %s, %d CUs, hwmon %s, power cap %.0f W\n", prop.name, cus, hw.c_str(), read_long(hw + "/power1_cap") / 1e6);
    usleep(500000);
    printf("idle: sclk %ld MHz, power %.0f W\n", read_long(hw + "/freq1_input") / 1000000, read_long(hw + "/power1_input") / 1e6);

    auto report = [&](const char* name, const Reading& r, double units_per_launch, const char* unit, double scale) {
        printf("%-44s %8.1f %s  sclk %5.0f MHz  power %5.0f W\n", name, units_per_launch * r.launches / r.seconds * scale, unit, r.mhz,
               r.watts);
        fflush(stdout);
    };
    const int it_m = 4000;
    const double flop_mfma = 2.0 * 32 * 32 * 16;
    {
        auto r = hold(hw, seconds, [&] { hipLaunchKernelGGL(mfma_kernel<0>, dim3(cus), dim3(256), 0, 0, sink, it_m); });
        report("mfma f16 32x32x16, 1 wave/SIMD, back to back", r, (double)cus * 4 * it_m * 16 * flop_mfma, "TFLOP/s", 1e-12);
    }
    {
        auto r = hold(hw, seconds, [&] { hipLaunchKernelGGL(mfma_kernel<0>, dim3(cus), dim3(512), 0, 0, sink, it_m); });
        report("mfma f16 32x32x16, 2 waves/SIMD, back to back", r, (double)cus * 8 * it_m * 16 * flop_mfma, "TFLOP/s", 1e-12);
    }
    {       // 32 instructions of 2 * 16 * 16 * 32 FLOP per iteration = the 16 of 2 * 32 * 32 * 16 above
        auto r = hold(hw, seconds, [&] { hipLaunchKernelGGL(mfma16_kernel, dim3(cus), dim3(256), 0, 0, sink, it_m); });
        report("mfma f16 16x16x32, 1 wave/SIMD, back to back", r, (double)cus * 4 * it_m * 16 * flop_mfma, "TFLOP/s", 1e-12);
    }
    {
        auto r = hold(hw, seconds, [&] { hipLaunchKernelGGL(mfma16_kernel, dim3(cus), dim3(512), 0, 0, sink, it_m); });
        report("mfma f16 16x16x32, 2 waves/SIMD, back to back", r, (double)cus * 8 * it_m * 16 * flop_mfma, "TFLOP/s", 1e-12);
    }
    {
        auto r = hold(hw, seconds, [&] { hipLaunchKernelGGL(mfma_kernel<2>, dim3(cus), dim3(256), 0, 0, sink, it_m / 2); });
        report("mfma, 1 wave/SIMD, 32 idle cycles after each", r, (double)cus * 4 * (it_m / 2) * 16 * flop_mfma, "TFLOP/s", 1e-12);
    }
    {
        auto r = hold(hw, seconds, [&] { hipLaunchKernelGGL(mfma_kernel<6>, dim3(cus), dim3(256), 0, 0, sink, it_m / 4); });
        report("mfma, 1 wave/SIMD, 96 idle cycles after each", r, (double)cus * 4 * (it_m / 4) * 16 * flop_mfma, "TFLOP/s", 1e-12);
    }
    const int it_v = 20000;
    {
        auto r = hold(hw, seconds, [&] { hipLaunchKernelGGL(valu_kernel<true>, dim3(cus), dim3(512), 0, 0, sink, it_v); });
        report("v_pk_fma_f32, 2 waves/SIMD", r, (double)cus * 8 * it_v * 32 * 64 * 2 * 2, "TFLOP/s", 1e-12);
    }
    {
        auto r = hold(hw, seconds, [&] { hipLaunchKernelGGL(valu_kernel<false>, dim3(cus), dim3(512), 0, 0, sink, it_v); });
        report("v_fma_f32, 2 waves/SIMD", r, (double)cus * 8 * it_v * 64 * 64 * 2, "TFLOP/s", 1e-12);
    }
    const int it_l = 20000;
    {
        auto r = hold(hw, seconds, [&] { hipLaunchKernelGGL(lds_kernel, dim3(cus), dim3(512), 0, 0, sink, it_l); });
        report("ds_read_b128, 2 waves/SIMD", r, (double)cus * 512 * it_l * 8 * 16, "TB/s", 1e-12);
    }
    {
        const int n = 1 << 17;                 // float4: 2 MB
        float4* in;
        CHECK(hipMalloc(&in, (size_t)n * 16));
        CHECK(hipMemset(in, 1, (size_t)n * 16));
        const int it_2 = 40;
        auto r = hold(hw, seconds, [&] { hipLaunchKernelGGL(l2_kernel, dim3(cus), dim3(512), 0, 0, in, sink, n, it_2); });
        report("global_load_dwordx4 of an L2-resident 2 MB", r, (double)cus * n * 16 * it_2, "TB/s", 1e-12);
        CHECK(hipFree(in));
    }
    {
        const size_t n = (size_t)1 << 22;      // float4: 64 MiB each way - beyond the 8 x 4 MB of L2, inside the 256 MB Infinity Cache
        float4 *in, *out;
        CHECK(hipMalloc(&in, n * 16));
        CHECK(hipMalloc(&out, n * 16));
        CHECK(hipMemset(in, 1, n * 16));
        auto r = hold(hw, seconds, [&] { hipLaunchKernelGGL(copy_kernel, dim3(cus * 16), dim3(256), 0, 0, in, out, n); });
        report("float4 copy, 64 MiB read + 64 MiB written", r, 2.0 * n * 16, "TB/s", 1e-12);
        CHECK(hipFree(in));
        CHECK(hipFree(out));
    }
    {
        const size_t n = (size_t)1 << 26;      // float4: 1 GiB each way
        float4 *in, *out;
        CHECK(hipMalloc(&in, n * 16));
        CHECK(hipMalloc(&out, n * 16));
        CHECK(hipMemset(in, 1, n * 16));
        auto r = hold(hw, seconds, [&] { hipLaunchKernelGGL(copy_kernel, dim3(cus * 16), dim3(256), 0, 0, in, out, n); });
        report("float4 copy, 1 GiB read + 1 GiB written", r, 2.0 * n * 16, "TB/s", 1e-12);
        CHECK(hipFree(in));
        CHECK(hipFree(out));
    }
    return 0;
}
