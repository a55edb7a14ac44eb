#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out)
{
    int lane = threadIdx.x;
    int m = __builtin_amdgcn_update_dpp(0, lane, 0x140, 0xf, 0xf, false);      // row_mirror
    int r = __builtin_amdgcn_update_dpp(0, m, 0x121, 0xf, 0xf, false);         // row_ror:1
    int hm = __builtin_amdgcn_update_dpp(0, lane, 0x141, 0xf, 0xf, false);     // row_half_mirror
    int r8 = __builtin_amdgcn_update_dpp(0, lane, 0x128, 0xf, 0xf, false);     // row_ror:8
    out[lane] = m; out[64 + lane] = r; out[128 + lane] = hm; out[192 + lane] = r8;
}
int main()
{
    int *d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *n[4] = {"mirror", "mirror+ror1", "half_mirror", "ror8"};
    for (int t = 0; t < 4; ++t) { printf("%-12s:", n[t]); for (int i = 0; i < 20; ++i) printf(" %d", h[t * 64 + i]); printf("\n"); }
    return 0;
}
