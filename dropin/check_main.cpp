// A caller written the way the reference's tools are (src/stft.cpp:16,208-226; src/stft_flt.cpp): it includes the .c
// files by name, calls codee::stft_s / stft_flt_d / fft_power_c with host arrays of its own and treats a non-zero
// return as fatal.  Built against dropin/c instead of the reference's c/ directory (dropin/Makefile); run by
// tests/test_dropin.py, which checks the file it writes against the CPU oracle.
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include "hann.c"
#include "povey.c"
#include "stft.c"
#include "stft_flt.c"
#include "fft_power.c"

static void fill(std::vector<double> &x, uint64_t seed)
{
    // a 64-bit LCG (the test regenerates the same numbers in Python)
    uint64_t s = seed;
    for (auto &v : x) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = ((double)(s >> 11) / 9007199254740992.0 - 0.5) * 0.2; }
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s out.bin\n", argv[0]); return 2; }
    const size_t N = 16000, L = 400, nfft = 512, stp = 160, F = nfft / 2 + 1;
    std::vector<double> xd(N);
    fill(xd, 12345);
    std::vector<float> xs(xd.begin(), xd.end());
    FILE *f = fopen(argv[1], "wb");
    if (!f) { perror("fopen"); return 2; }

    // stft_s, sized by the caller with the reference's formula (src/stft.cpp:179-187)
    std::vector<float> ws(L);
    if (codee::hann_s(ws.data(), L, 0)) { return 1; }
    const size_t W = (N + stp / 2) / stp;
    std::vector<float> Ys(W * F);
    if (codee::stft_s(Ys.data(), xs.data(), ws.data(), N, L, nfft, stp, 0, 1, 0, 0)) { fprintf(stderr, "problem during call to stft_s\n"); return 1; }
    fwrite(ws.data(), sizeof(float), L, f);
    fwrite(Ys.data(), sizeof(float), Ys.size(), f);

    // stft_flt_d with a hop that is no whole number
    std::vector<double> wd(L);
    if (codee::povey_d(wd.data(), L, 0)) { return 1; }
    const size_t Wf = 90;
    std::vector<double> Yd(Wf * F);
    if (codee::stft_flt_d(Yd.data(), xd.data(), wd.data(), N, L, Wf, nfft, 200.0, 160.5, 0, 1, 0)) { fprintf(stderr, "problem during call to stft_flt_d\n"); return 1; }
    fwrite(wd.data(), sizeof(double), L, f);
    fwrite(Yd.data(), sizeof(double), Yd.size(), f);

    // fft_power_c on the first 1000 samples taken as 500 complex values
    std::vector<float> P(500);
    if (codee::fft_power_c(P.data(), xs.data(), 500)) { return 1; }
    fwrite(P.data(), sizeof(float), P.size(), f);

    // an argument error comes back as 1 with one line on stderr, nothing is written
    if (codee::stft_s(Ys.data(), xs.data(), ws.data(), N, 0, nfft, stp, 0, 0, 0, 0) != 1) { fprintf(stderr, "L = 0 was accepted\n"); return 3; }
    fclose(f);
    return 0;
}
