#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *out)
{
    unsigned a = 100 + threadIdx.x, b = 200 + threadIdx.x;
    u2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x] = r.x; out[64 + threadIdx.x] = r.y;
}
int main()
{
    unsigned *d, h[128]; hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("first : lane0=%u lane31=%u lane32=%u lane63=%u\n", h[0], h[31], h[32], h[63]);
    printf("second: lane0=%u lane31=%u lane32=%u lane63=%u\n", h[64], h[95], h[96], h[127]);
    return 0;
}
