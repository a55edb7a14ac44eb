// Host-side paths of libdsp_stft_hip.so under AddressSanitizer / UndefinedBehaviorSanitizer (host code only; the device
// code is the normal build).  Calls the host-pointer entry points the way a C caller of the reference would, at sizes that
// take every staging path: small calls (one blocking copy each way), batches streamed by signals, one long signal
// streamed by ranges of frames (equally spaced frames, tables of frame starts, the reference framing with head frames),
// the single-process multi-GPU forms on the devices present, the argument errors.  Streamed results are compared with
// small calls on the same signals.  Build and run: tools/san/run.sh.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "dsp_stft.h"

static int failures = 0;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

template <typename T> static void fill(std::vector<T> &x, unsigned seed)
{
    unsigned s = seed * 2654435761u + 12345u;
    for (auto &v : x) { s = s * 1664525u + 1013904223u; v = (T)(((int)(s >> 8) % 2001 - 1000) * 1e-4); }
}
static size_t frames(size_t N, size_t L, size_t stp, int snip) { return snip ? 1u + (N - L) / stp : (N + stp / 2u) / stp; }

template <typename T> struct Api;
template <> struct Api<float> {
    static constexpr auto stft = stft_s; static constexpr auto stft_flt = stft_flt_s; static constexpr auto batch = dsp_stft_batch_s;
    static constexpr auto flt_batch = dsp_stft_flt_batch_s; static constexpr auto multi = dsp_stft_multi_s; static constexpr auto flt_multi = dsp_stft_flt_multi_s;
    static constexpr auto hann = hann_s; static constexpr auto pc = dsp_stft_pc_batch_s;
    static constexpr auto strided = dsp_stft_batch_strided_s; static constexpr auto flt_strided = dsp_stft_flt_batch_strided_s;
};
template <> struct Api<double> {
    static constexpr auto stft = stft_d; static constexpr auto stft_flt = stft_flt_d; static constexpr auto batch = dsp_stft_batch_d;
    static constexpr auto flt_batch = dsp_stft_flt_batch_d; static constexpr auto multi = dsp_stft_multi_d; static constexpr auto flt_multi = dsp_stft_flt_multi_d;
    static constexpr auto hann = hann_d; static constexpr auto pc = dsp_stft_pc_batch_d;
    static constexpr auto strided = dsp_stft_batch_strided_d; static constexpr auto flt_strided = dsp_stft_flt_batch_strided_d;
};

template <typename T> static void run(const char *name)
{
    using A = Api<T>;
    const size_t L = 400, nfft = 512, F = nfft / 2 + 1, stp = 160;
    std::vector<T> win(L);
    CHECK(A::hann(win.data(), L, 0) == 0);

    // small calls, every flag
    {
        const size_t N = 16000, W = frames(N, L, stp, 0);
        std::vector<T> x(N), y(W * F);
        fill(x, 1);
        for (int fl = 0; fl < 8; ++fl) { CHECK(A::stft(y.data(), x.data(), win.data(), N, L, nfft, stp, 0, fl & 1, (fl >> 1) & 1, (fl >> 2) & 1) == 0); }
        CHECK(A::stft(y.data(), x.data(), win.data(), N, L, nfft, stp, 1, 0, 0, 0) == 0);
        CHECK(A::stft_flt(y.data(), x.data(), win.data(), N, L, 90, nfft, (T)0, (T)160.5, 1, 0, 1) == 0);
        CHECK(A::stft(y.data(), x.data(), win.data(), N, L, 100, stp, 0, 0, 0, 0) != 0);          // nfft < L
        CHECK(A::stft(y.data(), x.data(), win.data(), N, L, nfft, 0, 0, 0, 0, 0) != 0);            // stp 0
        CHECK(A::stft_flt(y.data(), x.data(), win.data(), N, L, N + 1, nfft, (T)0, (T)160, 0, 0, 0) != 0);   // W > N
    }
    // a batch streamed by signals (above 64 MB of input + output), compared with single calls
    {
        const size_t B = 96, N = 120000, ldx = N + 24, W = frames(N, L, stp, 0);
        std::vector<T> x(B * ldx), y(B * W * F), y1(W * F);
        fill(x, 2);
        CHECK(A::batch(y.data(), x.data(), win.data(), B, N, ldx, L, nfft, stp, 0, 1, 0, 1) == 0);
        for (size_t b : {size_t(0), size_t(47), B - 1}) {
            CHECK(A::stft(y1.data(), x.data() + b * ldx, win.data(), N, L, nfft, stp, 0, 1, 0, 1) == 0);
            CHECK(memcmp(y1.data(), y.data() + b * W * F, W * F * sizeof(T)) == 0);
        }
        CHECK(A::pc(y.data(), x.data(), win.data(), B, N, ldx, L, nfft, stp, 0, 0, 0, 0, (T)0.3, (T)1e-5) == 0);
        const size_t Wf = 700;
        std::vector<T> z(B * Wf * F), z1(Wf * F);
        for (T hop : {(T)160, (T)160.5}) {
            CHECK(A::flt_batch(z.data(), x.data(), win.data(), B, N, ldx, L, Wf, nfft, (T)0, hop, 0, 0, 0) == 0);
            CHECK(A::stft_flt(z1.data(), x.data() + 5 * ldx, win.data(), N, L, Wf, nfft, (T)0, hop, 0, 0, 0) == 0);
            CHECK(memcmp(z1.data(), z.data() + 5 * Wf * F, Wf * F * sizeof(T)) == 0);
        }
        int devs[1] = {0};
        CHECK(A::multi(y.data(), x.data(), win.data(), B, N, ldx, L, nfft, stp, 0, 0, 0, 0, devs, 1) == 0);
        CHECK(A::flt_multi(z.data(), x.data(), win.data(), B, N, ldx, L, Wf, nfft, (T)200, (T)160, 0, 0, 0, devs, 1) == 0);
    }
    // interleaved channels ([N][C]) through the strided host forms: 8 channels (float32: read in place by the kernel), 6 channels
    // (gathered), a block of 4 channels of the 8; against single calls on the channels copied out
    {
        const size_t N = 30000, W = frames(N, L, stp, 0);
        for (size_t C : {size_t(8), size_t(6)}) {
            std::vector<T> xi(N * C), ch(N), y(C * W * F), y1(W * F);
            fill(xi, 4);
            CHECK(A::strided(y.data(), xi.data(), win.data(), C, N, 1, C, L, nfft, stp, 0, 0, 0, 0) == 0);
            for (size_t c : {size_t(0), C - 1}) {
                for (size_t n = 0; n < N; ++n) { ch[n] = xi[n * C + c]; }
                CHECK(A::stft(y1.data(), ch.data(), win.data(), N, L, nfft, stp, 0, 0, 0, 0) == 0);
                double worst = 0, top = 0;
                for (size_t i = 0; i < W * F; ++i) { worst = std::max(worst, std::fabs((double)y1[i] - (double)y[c * W * F + i])); top = std::max(top, (double)y1[i]); }
                CHECK(worst <= 1e-5 * top);
            }
            if (C == 8) {
                std::vector<T> yb(4 * W * F);
                CHECK(A::strided(yb.data(), xi.data() + 4, win.data(), 4, N, 1, C, L, nfft, stp, 0, 1, 0, 1) == 0);
                CHECK(A::flt_strided(yb.data(), xi.data(), win.data(), 4, N, 1, C, L, 150, nfft, (T)200, (T)160, 0, 0, 0) == 0);
            }
            CHECK(A::strided(y.data(), xi.data(), win.data(), C, N, 1, 0, L, nfft, stp, 0, 0, 0, 0) != 0);   // incx 0
        }
    }
    // one long signal streamed by ranges of frames; the head of it must equal a small call on its first samples
    {
        const size_t N = 20000003, W = frames(N, L, stp, 0), Nh = 50000, Wh = 200;
        std::vector<T> x(N), y(W * F), yh(frames(Nh, L, stp, 0) * F);
        fill(x, 3);
        CHECK(A::stft(y.data(), x.data(), win.data(), N, L, nfft, stp, 0, 0, 0, 0) == 0);
        CHECK(A::stft(yh.data(), x.data(), win.data(), Nh, L, nfft, stp, 0, 0, 0, 0) == 0);
        CHECK(memcmp(yh.data(), y.data(), Wh * F * sizeof(T)) == 0);
        for (T c0 : {(T)0, (T)200, (T)-77}) {
            for (T hop : {(T)160, (T)160.25}) {
                const size_t Wf = (size_t)((T)(N - 1000) / hop);
                std::vector<T> z(Wf * F), zh(Wh * F);
                CHECK(A::stft_flt(z.data(), x.data(), win.data(), N, L, Wf, nfft, c0, hop, 0, 0, 0) == 0);
                // (the reference framing reads every frame that starts inside the signal `shift` samples early, and shift
                // depends on c0 and the hop only: the same for the short call)
                CHECK(A::stft_flt(zh.data(), x.data(), win.data(), Nh, L, Wh, nfft, c0, hop, 0, 0, 0) == 0);
                CHECK(memcmp(zh.data(), z.data(), Wh * F * sizeof(T)) == 0);
            }
        }
        CHECK(dsp_stft_release_buffers() == 0);
        // other transform lengths through the same path
        for (size_t n2 : {size_t(256), size_t(1024), size_t(4096), size_t(8192), size_t(100)}) {
            std::vector<T> w2(n2);
            CHECK(A::hann(w2.data(), n2, 0) == 0);
            const size_t N2 = n2 == 100 ? 2000000 : N, W2 = frames(N2, n2, n2 / 4, 0);
            std::vector<T> y2(W2 * (n2 / 2 + 1));
            CHECK(A::stft(y2.data(), x.data(), w2.data(), N2, n2, n2, n2 / 4, 0, 0, 0, 0) == 0);
            bool finite = true;
            for (size_t i = 0; i < y2.size(); i += 997) { finite = finite && std::isfinite((double)y2[i]); }
            CHECK(finite);
        }
    }
    printf("%s: done, %d failure(s) so far\n", name, failures);
}

int main()
{
    run<float>("float32");
    run<double>("float64");
    return failures ? 1 : 0;
}
