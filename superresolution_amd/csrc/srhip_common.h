// Shared declarations for the gfx950 kernels of libsrhip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/srhip.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA A/B fragment: 8 bf16 = 4 VGPRs
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---- bf16 helpers (round-to-nearest-even; the cast lowers to v_cvt_pk_bf16_f32 on gfx950) ----
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short v)
{
    return __uint_as_float(((unsigned)v) << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f)
{
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
typedef __bf16 srhip_bf16x2 __attribute__((ext_vector_type(2)));
typedef float srhip_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi)
{
    const srhip_f32x2 f = {lo, hi};
    const srhip_bf16x2 b = __builtin_convertvector(f, srhip_bf16x2);  // one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(unsigned, b);
}
__device__ __forceinline__ float bf16_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf16_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }

// ---- error channel ----
void srhip_set_error(const char* fmt, ...);
int srhip_check_launch(const char* what);           // hipGetLastError -> status
int srhip_num_cus();
constexpr int kSrhipMaxDevices = 64;
int srhip_current_device();                         // hipGetDevice, clamped to [0, kSrhipMaxDevices)

#define SRHIP_REQUIRE(cond, ...)                      \
    do {                                              \
        if (!(cond)) {                                \
            srhip_set_error(__VA_ARGS__);             \
            return SRHIP_ERR_INVALID;                 \
        }                                             \
    } while (0)

static inline int srhip_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }
// Results of asm loads are invisible to the compiler's wait bookkeeping: it believes the destination
// registers hold data as soon as the load has issued and may copy them (register moves at loop exits and
// merges) before the hand-placed wait.  Every such value therefore passes through one fence right after
// the wait that covers it; the marker names the registers in the assembly, and tools/isa_lint.py checks
// that nothing touches them between the load and this marker on any path.
#define SRHIP_LOAD_FENCE(x) asm volatile("; srhip-fence %0" : "+v"(x))

