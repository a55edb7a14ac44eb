// Device helpers shared by the kernels that stage their input spans by LDS-DMA and deal their steps at run time
// (r512_fast_kernel.hip, r512x_kernel.hip, r256_kernel.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#ifndef DSP_MASK_RECOMPUTE
#define DSP_MASK_RECOMPUTE 0    // The lane masks of the rows that may reach past the window.  0: the compiler hoists the 32 masks out of
                                // the step loop into scalar registers, which do not hold them all: they spill into the lanes of a vector
                                // register and a select then costs two v_readlane and an s_nop.  1: they are recomputed every step from
                                // one pinned register (a compare per select) -- measured 5 % SLOWER on the 512-point kernel (L 512 / hop
                                // 128: 0.387 -> 0.409 ms) and the same on the 1024 / 2048-point one, so it stays off
#endif

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
    __builtin_assume(p != nullptr);          // (no test for the null pointer in the conversion)
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p;
}
// The same load with its addresses split into wave-uniform parts and small per-lane parts: memory address = base (scalar
// registers) + voff bytes (32 bits per lane), LDS address = lds (scalar) + lds_off (a constant once the call is unrolled).  Three
// instructions -- M0, the wait state that a write of M0 needs, the load -- where the form above takes eight with the 64-bit
// address arithmetic in front of it; a wave issues one instruction of any kind every four cycles, and these kernels spend more
// than half of their time issuing.  M0 is left changed: the compiler keeps nothing in it on this target unless a kernel indexes
// registers or sends messages, which these do not (tools/asm_waits.py checks on the built library that nothing else touches M0
// in a kernel that holds such a load).
#ifndef DSP_GLDS_SPLIT
#define DSP_GLDS_SPLIT 1
#endif
template <typename P> __device__ __forceinline__ const P *ring_uniform(const P *p)     // a wave-uniform pointer, into scalar registers
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const P *>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void ring_glds16_at(const void *base, const unsigned voff, const unsigned lds, const int lds_off)
{
#if DSP_GLDS_SPLIT
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(base), "s"(lds), "n"(lds_off) : "memory", "scc");
#else
    ring_glds16(reinterpret_cast<const char *>(base) + voff, lds + lds_off);
#endif
}
// The counted wait: until at most N memory operations of the wave are outstanding.  Memory operations complete in
// issue order, so with exactly N store instructions issued behind a span's loads this waits for the span and leaves
// the stores in flight.  The count is an assumption about the code the compiler emits between the two points; it is
// checked on the built library (tests/test_asm_waits.py), and DSP_SAFE_WAITS = 1 builds every such wait as a wait for
// all outstanding operations (libdsp_stft_hip_safe.so, compared bit for bit with the counted build on the GPU).
#ifndef DSP_SAFE_WAITS
#define DSP_SAFE_WAITS 0
#endif
// (expcnt(6) marks the instruction for that check: these kernels issue nothing that the export counter counts, so the
// field never waits, and the compiler's own waits never carry it)
template <int N> __device__ __forceinline__ void ring_wait_vm() { asm volatile("s_waitcnt vmcnt(%0) expcnt(6)" ::"n"(DSP_SAFE_WAITS ? 0 : N) : "memory"); }
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

// ---- rows of a step, staged in LDS in output order, to memory at any 4-byte alignment --------------------------
// How a block of `nout` floats at `yblk` lies against 16-byte boundaries: `head` floats (0 to 3) come before the
// first boundary.  The kernels stage value e of the block at stage[e], stage = tile + ring_stage_shift(head), which
// puts the 16-byte boundaries of the OUTPUT on 16-byte boundaries of the LDS.
__device__ __forceinline__ int ring_block_head(const float *yblk)
{
    return __builtin_amdgcn_readfirstlane((int)(((16u - ((unsigned)reinterpret_cast<uintptr_t>(yblk) & 15u)) & 15u) >> 2));
}
__device__ __forceinline__ int ring_stage_shift(const int head) { return (4 - head) & 3; }
// NVI wave stores of whole 16-byte vectors (NVI * 256 >= nout) and one more for the values before and after
// them: NVI + 1 store instructions whatever nout and the alignment, which keeps a counted vmcnt wait exact.
// Surplus lanes repeat a valid store (same value to the same address).  SKIP: diagnostic builds without the stores.
template <int NVI, bool SKIP = false>
__device__ __forceinline__ void ring_store_rows(float *__restrict__ yblk, const float *stage, const int head, const int nout, const int lane)
{
    typedef float f4nt __attribute__((ext_vector_type(4)));
    const int nb = (nout - head) >> 2, tail = nout - head - 4 * nb;
    // (The same split of the addresses as in ring_glds16_at -- a wave-uniform base in scalar registers plus a 32-bit offset per lane,
    // two vector instructions of address arithmetic per store instead of five -- measured 3.5 % SLOWER on the 1024- and 2048-point
    // kernels, 0.456 -> 0.472 and 0.509 -> 0.532 ms: the stores stay on 64-bit addresses per lane.)
    // (unsigned vector numbers: the 64-bit address of a store is then one shift-and-add on the zero-extended number)
    const f4nt *const sv = reinterpret_cast<const f4nt *>(stage + head);
    f4nt *const gv = reinterpret_cast<f4nt *>(yblk + head);
#pragma unroll
    for (int i = 0; i < NVI; ++i) {
        const unsigned v = min((unsigned)lane + 64u * i, (unsigned)(nb - 1));
        const f4nt val = sv[v];
        if constexpr (SKIP) { asm volatile("" :: "v"(val)); }
        else { __builtin_nontemporal_store(val, gv + v); }
    }
    int e = nout - 1;
    if (lane < head) { e = lane; }
    else if (lane >= 4 && lane - 4 < tail) { e = head + 4 * nb + lane - 4; }
    if constexpr (!SKIP) { __builtin_nontemporal_store(stage[e], yblk + e); }
}

// ---- one span into a wave's slot: NV LDS-DMA loads whatever the case (a counted wait relies on that) -----------
// Samples s0 .. s0 + need of the signal at xb (read at s0 - rd_shift under the zero-edge rule).  Inside the signal:
// NV wave loads of 1 KiB, lanes past the last vector that is used (nvec = ceil(need / VEC), VEC = 16 bytes of samples)
// read that vector again.  A span that crosses an edge of its signal is gathered sample by sample (reflection or zeros)
// and NV placeholder loads follow, from the first KiB of `tw` into `dummy`.
// SPARE: the lanes past the last vector read the vector BEFORE it (all of whose samples are needed) instead: the copies that fill the
// rest of the slot then hold nothing but samples of the span proper, for kernels that read the slot past `need` unmasked.
template <int NV, bool SPARE = false, typename T, typename FETCH>
__device__ __forceinline__ void ring_load_span(T *slot, T *dummy, const T *tw, const T *xb, const int s0, const int need,
                                               const int N, const int edge, const int rd_shift, const int lane, FETCH fetch)
{
    constexpr int VEC = 16 / (int)sizeof(T);
    typedef T vecT __attribute__((ext_vector_type(VEC)));
    const int nvec = (need + VEC - 1) / VEC;
    const T *src = ring_uniform(xb + (s0 - (edge ? rd_shift : 0)));
    const bool inside = s0 >= 0 && s0 + VEC * nvec <= N && (edge == 0 || s0 >= rd_shift);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(ring_lds_addr(slot));
    if (inside) {
        const unsigned last = 16u * (unsigned)(nvec - 1);
#pragma unroll
        for (int q = 0; q < NV; ++q) {
#ifdef RING_FAKE_NV
            if (q >= RING_FAKE_NV) { continue; }       // diagnostic builds: fewer loads (wrong results)
#endif
            const unsigned own = 1024u * q + 16u * lane;
            if constexpr (SPARE) { ring_glds16_at(src, own > last ? (last >= 16u ? last - 16u : last) : own, lds0, 1024 * q); }
            else { ring_glds16_at(src, min(own, last), lds0, 1024 * q); }
        }
    } else {
#pragma clang loop unroll(disable)
        for (int q = 0; q < NV; ++q) {
            vecT t;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const int i = (q * 64 + lane) * VEC + v;
                t[v] = (i < need) ? fetch(xb, s0, i) : T(0);
            }
            reinterpret_cast<vecT *>(slot)[q * 64 + lane] = t;
        }
        const unsigned ldsd = __builtin_amdgcn_readfirstlane(ring_lds_addr(dummy));
#pragma unroll
        for (int q = 0; q < NV; ++q) { ring_glds16_at(ring_uniform(tw), 16u * lane, ldsd, 0); }
    }
}

// The same span one load at a time, for kernels that place the NV loads between other work (what an LDS-DMA load costs the
// wave that issues it depends on what else the wave is doing: back to back behind a transform they measured 160 cycles each).
template <typename T> struct RingSpan { const T *src, *xb; int s0, nvec; bool inside; };
template <typename T>
__device__ __forceinline__ RingSpan<T> ring_span_open(const T *xb, const int s0, const int need, const int N, const int edge, const int rd_shift)
{
    constexpr int VEC = 16 / (int)sizeof(T);
    RingSpan<T> r;
    r.xb = xb; r.s0 = s0;
    r.nvec = (need + VEC - 1) / VEC;
    r.src = xb + (s0 - (edge ? rd_shift : 0));
    r.inside = s0 >= 0 && s0 + VEC * r.nvec <= N && (edge == 0 || s0 >= rd_shift);
    return r;
}
template <typename T, typename FETCH>
__device__ __forceinline__ void ring_span_piece(const RingSpan<T> &r, const int q, T *slot, T *dummy, const T *tw, const int need, const int lane, FETCH fetch)
{
    constexpr int VEC = 16 / (int)sizeof(T);
    typedef T vecT __attribute__((ext_vector_type(VEC)));
    if (r.inside) {
        ring_glds16(r.src + min(q * 64 + lane, r.nvec - 1) * VEC, __builtin_amdgcn_readfirstlane(ring_lds_addr(slot + q * 64 * VEC)));
    } else {
        vecT t;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int i = (q * 64 + lane) * VEC + v;
            t[v] = (i < need) ? fetch(r.xb, r.s0, i) : T(0);
        }
        reinterpret_cast<vecT *>(slot)[q * 64 + lane] = t;
        ring_glds16(tw + lane * VEC, __builtin_amdgcn_readfirstlane(ring_lds_addr(dummy)));
    }
}

}  // namespace dsp
