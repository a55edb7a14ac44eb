// Where a one-utterance process spends its start-up: loading the library (the HIP runtime and the registration of the
// code object), the first HIP call (runtime and device initialisation), the first stft_s call (the code object goes
// to the device, twiddle table, buffers, first launch) and a second call (steady state).
// build: g++ -O2 -o scratch/startup_probe tools/experiments/startup_probe.cpp -ldl ; run: scratch/startup_probe dsp_amd/libdsp_stft_hip.so
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <vector>
#include <cmath>
using clk = std::chrono::steady_clock;
static double ms(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }
int main(int argc, char **argv)
{
    const char *path = argc > 1 ? argv[1] : "dsp_amd/libdsp_stft_hip.so";
    const auto t0 = clk::now();
    void *h = dlopen(path, RTLD_NOW);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
    const auto t1 = clk::now();
    auto set_device = (int (*)(int))dlsym(h, "dsp_set_device");
    auto stft_s = (int (*)(float *, const float *, const float *, size_t, size_t, size_t, size_t, int, int, int, int))dlsym(h, "stft_s");
    auto release = (int (*)())dlsym(h, "dsp_stft_release_buffers");
    if (!set_device || !stft_s) { fprintf(stderr, "symbols missing\n"); return 1; }
    // first HIP call of the process (hipGetDeviceCount inside): runtime + device initialisation
    void *hip = dlopen("libamdhip64.so", RTLD_NOW | RTLD_NOLOAD);
    auto hipGetDeviceCount = hip ? (int (*)(int *))dlsym(hip, "hipGetDeviceCount") : nullptr;
    auto hipFree = hip ? (int (*)(void *))dlsym(hip, "hipFree") : nullptr;
    int n = -1;
    if (hipGetDeviceCount) { (void)hipGetDeviceCount(&n); }
    const auto t2 = clk::now();
    if (hipFree) { (void)hipFree(nullptr); }           // forces the context of device 0
    const auto t3 = clk::now();
    const size_t N = 160000, L = 400, nfft = 512, stp = 160, W = (N + stp / 2) / stp, F = nfft / 2 + 1;
    std::vector<float> x(N), w(L), y(W * F);
    for (size_t i = 0; i < N; ++i) { x[i] = 0.1f * sinf(0.01f * i); }
    for (size_t i = 0; i < L; ++i) { w[i] = 0.5f - 0.5f * cosf(6.283185307f * i / (L - 1)); }
    int rc = stft_s(y.data(), x.data(), w.data(), N, L, nfft, stp, 0, 0, 0, 0);
    const auto t4 = clk::now();
    rc |= stft_s(y.data(), x.data(), w.data(), N, L, nfft, stp, 0, 0, 0, 0);
    const auto t5 = clk::now();
    printf("{\"library\": \"%s\", \"devices\": %d, \"rc\": %d, \"dlopen_ms\": %.2f, \"first_hip_call_ms\": %.2f, \"device_context_ms\": %.2f, \"first_stft_ms\": %.2f, \"second_stft_ms\": %.3f, \"total_ms\": %.2f}\n",
           path, n, rc, ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, t5), ms(t0, t5));
    (void)release; (void)set_device;
    return rc;
}
