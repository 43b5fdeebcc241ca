// Shared device/host helpers for libgimhip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/gim_hip.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((address_space(3))) void lds_void_t;

// ---- error plumbing (gim_last_error) ---------------------------------------------------------
void gim_set_error(const char* fmt, ...);
int gim_check_launch(const char* what);

#define GIM_REQUIRE(cond, ...)                       \
    do {                                             \
        if (!(cond)) {                               \
            gim_set_error(__VA_ARGS__);              \
            return GIM_ERR_INVALID;                  \
        }                                            \
    } while (0)

// ---- one-time per-DEVICE kernel attribute setup ------------------------------------------------
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device property: a process that drives several GPUs must
// set it on each of them.  `GimPerDevice flag;  if (flag.needed()) { ...set attributes...; flag.done(); }`
// (one bit per device ordinal; racing threads at worst set the attribute twice, which is harmless).
struct GimPerDevice {
    std::atomic<unsigned long long> mask[4] = {};
    static int dev() { int d = 0; (void)hipGetDevice(&d); return d & 255; }
    bool needed() const { const int d = dev(); return !((mask[d >> 6].load(std::memory_order_acquire) >> (d & 63)) & 1ull); }
    void done() { const int d = dev(); mask[d >> 6].fetch_or(1ull << (d & 63), std::memory_order_release); }
};

// ---- bf16 <-> f32 (round to nearest even, like torch's .to(bfloat16)) ------------------------
__device__ __forceinline__ float bf16_to_f32(unsigned short h) {
    return __uint_as_float(((unsigned)h) << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
// two fp32 -> packed bf16x2 (lo in bits 0-15), round to nearest even: one gfx950 instruction (v_cvt_pk_bf16_f32).
// Written as a vector conversion, NOT as inline asm: the hazard recogniser does not look inside asm statements, and an asm
// v_cvt_pk right behind the v_mfma that produces its operands reads the accumulator before the matrix pipe has written
// it back (measured in fine_fused.hip: wrong even heads in one of two inlined copies of the same code).
typedef __bf16 gim_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float gim_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    const gim_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, gim_bf16x2_t));
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) { return cvt_pk_bf16(lo, hi); }

// generic typed element access used by the small memory-bound kernels
template <bool BF16> struct ElemIO;
template <> struct ElemIO<true> {
    typedef unsigned short type;
    static __device__ __forceinline__ float ld(const void* p, size_t i) { return bf16_to_f32(((const unsigned short*)p)[i]); }
    static __device__ __forceinline__ void st(void* p, size_t i, float v) { ((unsigned short*)p)[i] = f32_to_bf16(v); }
    // 4 consecutive elements (8-byte aligned)
    static __device__ __forceinline__ float4 ld4(const void* p, size_t i) {
        uint2 u = *(const uint2*)((const unsigned short*)p + i);
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                           __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    }
    static __device__ __forceinline__ void st4(void* p, size_t i, float4 v) {
        *(uint2*)((unsigned short*)p + i) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    }
};
template <> struct ElemIO<false> {
    typedef float type;
    static __device__ __forceinline__ float ld(const void* p, size_t i) { return ((const float*)p)[i]; }
    static __device__ __forceinline__ void st(void* p, size_t i, float v) { ((float*)p)[i] = v; }
    static __device__ __forceinline__ float4 ld4(const void* p, size_t i) { return *(const float4*)((const float*)p + i); }
    static __device__ __forceinline__ void st4(void* p, size_t i, float4 v) { *(float4*)((float*)p + i) = v; }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware, bijective block remap (8 XCDs, block b is dispatched to XCD b % 8): gives every XCD a
// contiguous range of logical tile ids so that neighbouring tiles share operand panels in one L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned q = nblk >> 3, r = nblk & 7u;
    const unsigned xcd = bid & 7u, idx = bid >> 3;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
