// Device helpers shared by the kernels that stage their input spans by LDS-DMA and deal their steps at run time
// (r512_fast_kernel.hip, r512x_kernel.hip, r256_kernel.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace dsp {

// One 16-byte-per-lane LDS-DMA load (1 KiB per wave: lane i's 16 bytes land at lds + 16 i).  Written in
// assembly so that the compiler's wait counting does not see it (it would wait for ALL outstanding memory
// operations, the row stores included, before the first read of the span): the kernels wait for a span
// with a counted s_waitcnt themselves.  lds: wave-uniform LDS byte address.
__device__ __forceinline__ void ring_glds16(const void *g, const unsigned lds)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
__device__ __forceinline__ unsigned ring_lds_addr(const void *p)
{
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p;
}
template <int N> __device__ __forceinline__ void ring_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// Draw of a step: lane 0 alone adds 1 to the workgroup's counter (LDS byte address `cnt`); the old value is in
// lane 0 of `t` once ring_drawn has waited for it.  In assembly for the same reason as the load above: the
// compiler's form waits for the answer on the spot.
__device__ __forceinline__ void ring_draw(unsigned &t, const unsigned cnt)
{
    unsigned long long keep;
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tds_add_rtn_u32 %0, %2, %3\n\ts_mov_b64 exec, %1"
                 : "=&v"(t), "=&s"(keep) : "v"(cnt), "v"(1u) : "memory");
}
// (waits for EVERY LDS operation of the wave, the reads of the span included)
__device__ __forceinline__ int ring_drawn(unsigned &t)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t) :: "memory");
    return (int)__builtin_amdgcn_readfirstlane(t);
}

}  // namespace dsp
