// Device helpers shared by the MFMA convolution kernels of libsrhip.so (gfx950): compile-time loops, LDS / buffer
// accesses written as asm so that their waits can be counted by hand, pinned-operand MFMAs.
#pragma once
#include "srhip_common.h"

namespace {

// Nested generic lambdas only capture what their own body odr-uses outside discarded branches;
// naming a variable here makes every enclosing closure capture it.
#define SRHIP_CAPTURE(x) (void)&(x)

template <int V>
struct IntC { static constexpr int value = V; };
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (B < E) {
        f(IntC<B>{});
        static_for<B + 1, E>(f);
    }
}

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// ---- LDS traffic is written as asm so that the waits can be COUNTED (the compiler waits
// lgkmcnt(0) in front of an asm consumer).  LDS operations of a wave complete in issue order, so
// "all but the N youngest" is exact for them whatever scalar loads the compiler has in flight.
__device__ __forceinline__ bf16x8 lds_read_b128(unsigned addr)
{
    bf16x8 r;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr));
    return r;
}
__device__ __forceinline__ f32x4 lds_read_f32x4(unsigned addr)
{
    f32x4 r;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr));
    return r;
}
// The accumulator of output tile ct lives in the FIXED registers a[16ct : 16ct+15] (physical-register
// asm constraints), so that the epilogue can hand any 4-register quad of it to ds_write_b128 by name:
// no copy through VGPRs and no copy between AGPRs.
#define SRHIP_ACC_REG0 "{a[0:15]}"
#define SRHIP_ACC_REG1 "{a[16:31]}"
template <int CT, int G>
__device__ __forceinline__ void lds_write_acc_quad(unsigned addr, const f32x16& acc)
{
    if constexpr (CT == 0) asm volatile("ds_write_b128 %0, a[%2:%3]" ::"v"(addr), SRHIP_ACC_REG0(acc), "i"(4 * G), "i"(4 * G + 3) : "memory");
    else asm volatile("ds_write_b128 %0, a[%2:%3]" ::"v"(addr), SRHIP_ACC_REG1(acc), "i"(16 + 4 * G), "i"(16 + 4 * G + 3) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait()
{
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N));
}

// One MFMA with the weight fragment pinned to an AGPR (IN_AGPR) or a VGPR.  Written as asm: the
// builtin lets the register allocator park weights in AGPRs and copy them back (4 moves per MFMA).
// FIRST: start of an accumulation chain, C = 0 (inline constant), so no accumulator init exists.
template <bool IN_AGPR, bool FIRST, int CT>
__device__ __forceinline__ void mfma_pinned(f32x16& acc, const bf16x8& w, const bf16x8& b)
{
#define SRHIP_MFMA_CASE(REG)                                                                                          \
    if constexpr (IN_AGPR) {                                                                                          \
        if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=" REG(acc) : "a"(w), "v"(b));   \
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+" REG(acc) : "a"(w), "v"(b));                  \
    } else {                                                                                                          \
        if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=" REG(acc) : "v"(w), "v"(b));   \
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+" REG(acc) : "v"(w), "v"(b));                  \
    }
    if constexpr (CT == 0) { SRHIP_MFMA_CASE(SRHIP_ACC_REG0) } else { SRHIP_MFMA_CASE(SRHIP_ACC_REG1) }
#undef SRHIP_MFMA_CASE
}

// Vector-memory operations (LDS-DMA loads, residual loads, output stores) retire in issue order
// on one counter.  All of them are issued unconditionally (out-of-range lanes are dropped by the
// buffer bounds check, never by a branch), so "all but the N youngest" names exact operations.
template <int N>
__device__ __forceinline__ void vm_wait()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
// Raw buffer descriptor (stride 0, bounds-checked against `bytes`) for asm buffer loads.
__device__ __forceinline__ i32x4 make_srd(const void* base, unsigned bytes)
{
    const unsigned long long a = (unsigned long long)base;
    i32x4 d;
    d[0] = (int)(unsigned)(a & 0xffffffffull);
    d[1] = (int)(unsigned)((a >> 32) & 0xffffull);
    d[2] = (int)bytes;
    d[3] = 0x00020000;
    return d;
}
// Residual fetch hidden from the compiler's wait bookkeeping (it would drain the LDS-DMA queue).
// The LDS-DMA builtin lives in a plain device function: called directly from a generic lambda it
// makes the host pass drop the kernel's launch stub without a diagnostic.
__device__ __forceinline__ void lds_dma_b128(__amdgpu_buffer_rsrc_t rsrc, __attribute__((address_space(3))) void* dst,
                                             int voffset, int soffset)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voffset, soffset, 0, 0);
}

__device__ __forceinline__ void lds_dma_b32(__amdgpu_buffer_rsrc_t rsrc, __attribute__((address_space(3))) void* dst,
                                            int voffset, int soffset)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 4, voffset, soffset, 0, 0);   // lane l's dword lands at dst + 4 l
}
// one dword from LDS, completion left to the caller's lgkmcnt wait and fence
template <int OFF>
__device__ __forceinline__ unsigned lds_read_b32_off(const unsigned addr)
{
    unsigned r;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(OFF));
    return r;
}

__device__ __forceinline__ u32x4 buffer_load_b128_async(const i32x4 srd, unsigned voffset, unsigned soffset)
{
    u32x4 r;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r) : "v"(voffset), "s"(srd), "s"(soffset));
    return r;
}
// 16-byte store whose data registers stay untouched for four more issue cycles.  The store unit reads
// its 128 bits of data after the instruction has issued; the compiler's hazard table exempts buffer stores
// that carry an SGPR offset and happily re-uses the first data register in the very next VALU
// instruction, and with two waves per SIMD that write reaches lanes 12-15 of every 16-lane group before
// the store has read them (seen as stray LDS addresses in the output).
__device__ __forceinline__ void buffer_store_b128_guarded(const u32x4 data, const i32x4 srd, unsigned voffset, unsigned soffset)
{
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 3" ::"v"(data), "v"(voffset), "s"(srd), "s"(soffset) : "memory");
}
constexpr unsigned kOobOffset = 0x80000000u;  // beyond every image: the bounds check drops the lane

}  // namespace
