// Developer aid: the front-end kernel compiled with BD_FE_TRACE, run on random PCM; prints the s_memtime
// stamps of waves 0 and 7 of one workgroup (its second group).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DBD_FE_TRACE -o tools/fe_trace tools/fe_trace.hip && tools/fe_trace
#include "../buzzdetect_amd/csrc/frontend.hip"
#include <cmath>
#include <vector>

int main() {
    const int hop = 15360, n = hop * 1024, frames = 98304;
    std::vector<float> h(n);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) / 8388608.0f - 1.0f) * 0.3f; }
    bd::FeTables t{};
    for (int k = 0; k < 400; ++k) t.hann[k] = 0.5f - 0.5f * std::cos(2.0 * M_PI * k / 400.0);
    for (int k = 0; k < 256; ++k) t.tw256[k] = make_float2(std::cos(-2.0 * M_PI * k / 256), std::sin(-2.0 * M_PI * k / 256));
    for (int k = 0; k <= 257; ++k) t.tw512[k] = make_float2(std::cos(-2.0 * M_PI * k / 512), std::sin(-2.0 * M_PI * k / 512));
    for (int k = 0; k < bd::kMelNonZero; ++k) t.melw[k] = 0.3f;
    float *d_pcm, *d_out; bd::FeTables* d_t; unsigned long long* d_s;
    (void)hipMalloc(&d_pcm, n * 4); (void)hipMalloc(&d_out, (size_t)frames * 64 * 4); (void)hipMalloc(&d_t, sizeof(t)); (void)hipMalloc(&d_s, 64 * 8);
    (void)hipMemcpy(d_pcm, h.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(d_t, &t, sizeof(t), hipMemcpyHostToDevice);
    (void)hipMemset(d_s, 0, 64 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = 0; it < 5; ++it) {
        (void)hipEventRecord(e0);
        for (int r = 0; r < 20; ++r)
            hipLaunchKernelGGL(bd::fe::logmel_kernel, dim3(256), dim3(bd::fe::kThreads), 0, 0, d_pcm, n, frames, d_out, d_t, d_s);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%.1f us per launch (with stamps)\n", ms * 1000 / 20);
    }
    unsigned long long st[64]; (void)hipMemcpy(st, d_s, sizeof(st), hipMemcpyDeviceToHost);
    const char* names[21] = {"r0 start", "r0 windowed+prefetch issued", "r0 dft1+tw", "r0 exchange", "r0 dft2", "r0 mirror", "r0 mags", "", "r1 start", "r1 windowed", "r1 dft1+tw", "r1 exchange", "r1 dft2", "r1 mirror", "r1 mags", "", "fft done", "barrier A", "mel done", "barrier B", "output done"};
    for (int w = 0; w < 2; ++w) {
        printf("wave %d:\n", w ? bd::fe::kWavesG - 1 : 0);
        unsigned long long prev = st[w * 32];
        for (int i = 0; i < 21; ++i) {
            if (!names[i][0]) continue;
            const unsigned long long v = st[w * 32 + i];
            printf("  %-30s +%6lld  (t=%lld)\n", names[i], (long long)(v - prev), (long long)(v - st[w * 32]));
            prev = v;
        }
    }
    return 0;
}
