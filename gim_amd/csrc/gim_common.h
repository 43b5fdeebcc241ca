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

// ---- the 16-bit operand flavour of a translation unit ------------------------------------------------------------------
// The kernels of the gim_loftr path exist in two 16-bit flavours with identical instruction counts: bf16 (8 significand bits,
// fp32 range) and IEEE fp16 (11 bits; |x| < 65504).  Measured on the CPU emulation of the engine's roundings
// (tools/precision_emulation.py, profiles/r03_precision_emulation*.txt): fp16 operands cut the match-set flip rate against the
// fp32 reference from 1.95 % to 0.47 %, because three more significand bits survive every activation store.
// A source file is compiled ONCE PER FLAVOUR (gim_amd/build.py: -DGIM_HALF_KIND=0 / 1): everything that touches 16-bit values
// goes through the helpers below, `GIM_H16` is the dtype tag of the flavour and GIM_FN() names the entry points of the
// fp16 objects (`*_f16`); the bf16 objects carry the public names and forward dtype == GIM_F16 calls.
#ifndef GIM_HALF_KIND
#define GIM_HALF_KIND 0
#endif
#if GIM_HALF_KIND
#define GIM_H16 GIM_F16
#define GIM_FN(name) name##_f16
#else
#define GIM_H16 GIM_BF16
#define GIM_FN(name) name
#endif
typedef _Float16 gim_f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 gim_f16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((ext_vector_type(4))) float gim_f32x4v_t;

// fp32 pair -> packed fp16x2, round to nearest even: one v_cvt_pk_f16_f32
__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {
    const gim_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, gim_f16x2_t));
}
__device__ __forceinline__ float f16_lo(unsigned u) { return (float)__builtin_bit_cast(gim_f16x2_t, u)[0]; }
__device__ __forceinline__ float f16_hi(unsigned u) { return (float)__builtin_bit_cast(gim_f16x2_t, u)[1]; }
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// the flavour of this translation unit
__device__ __forceinline__ unsigned cvt_pk_h16(float lo, float hi) {
#if GIM_HALF_KIND
    return cvt_pk_f16(lo, hi);
#else
    return cvt_pk_bf16(lo, hi);
#endif
}
__device__ __forceinline__ float h16_lo(unsigned u) {
#if GIM_HALF_KIND
    return f16_lo(u);
#else
    return bf16_lo(u);
#endif
}
__device__ __forceinline__ float h16_hi(unsigned u) {
#if GIM_HALF_KIND
    return f16_hi(u);
#else
    return bf16_hi(u);
#endif
}
__device__ __forceinline__ float h16_to_f32(unsigned short h) {
#if GIM_HALF_KIND
    return (float)__builtin_bit_cast(_Float16, h);
#else
    return bf16_to_f32(h);
#endif
}
__device__ __forceinline__ unsigned short f32_to_h16(float f) {
#if GIM_HALF_KIND
    return __builtin_bit_cast(unsigned short, (_Float16)f);
#else
    return f32_to_bf16(f);
#endif
}
// 16-bit output of either kind from a translation unit of either flavour (the bf16 mode's first convolution reads an fp16
// image -- its rounding is half of that mode's error -- and writes bf16); `bf` is wave-uniform
__device__ __forceinline__ unsigned cvt_pk_16(float lo, float hi, bool bf) { return bf ? cvt_pk_bf16(lo, hi) : cvt_pk_f16(lo, hi); }
// MFMAs on 8 x 16-bit operand lanes (carried as bf16x8_t = 8 shorts whatever the flavour)
__device__ __forceinline__ f32x16_t mfma_h16_32x32x16(bf16x8_t a, bf16x8_t b, f32x16_t c) {
#if GIM_HALF_KIND
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gim_f16x8_t, a), __builtin_bit_cast(gim_f16x8_t, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ f32x4_t mfma_h16_16x16x32(bf16x8_t a, bf16x8_t b, f32x4_t c) {
#if GIM_HALF_KIND
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(gim_f16x8_t, a), __builtin_bit_cast(gim_f16x8_t, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}

// generic typed element access used by the small memory-bound kernels (<true>: the 16-bit flavour of the translation unit)
template <bool BF16> struct ElemIO;
template <> struct ElemIO<true> {
    typedef unsigned short type;
    static __device__ __forceinline__ float ld(const void* p, size_t i) { return h16_to_f32(((const unsigned short*)p)[i]); }
    static __device__ __forceinline__ void st(void* p, size_t i, float v) { ((unsigned short*)p)[i] = f32_to_h16(v); }
    // 4 consecutive elements (8-byte aligned)
    static __device__ __forceinline__ float4 ld4(const void* p, size_t i) {
        uint2 u = *(const uint2*)((const unsigned short*)p + i);
        return make_float4(h16_lo(u.x), h16_hi(u.x), h16_lo(u.y), h16_hi(u.y));
    }
    static __device__ __forceinline__ void st4(void* p, size_t i, float4 v) {
        *(uint2*)((unsigned short*)p + i) = make_uint2(cvt_pk_h16(v.x, v.y), cvt_pk_h16(v.z, v.w));
    }
};
template <> struct ElemIO<false> {
    typedef float type;
    static __device__ __forceinline__ float ld(const void* p, size_t i) { return ((const float*)p)[i]; }
    static __device__ __forceinline__ void st(void* p, size_t i, float v) { ((float*)p)[i] = v; }
    static __device__ __forceinline__ float4 ld4(const void* p, size_t i) { return *(const float4*)((const float*)p + i); }
    static __device__ __forceinline__ void st4(void* p, size_t i, float4 v) { *(float4*)((float*)p + i) = v; }
};

// ---- fp16 range guard ---------------------------------------------------------------------------------------------------------------
// IEEE fp16 ends at 65504.  The un-normalised ResNet residual streams are the values of the path that can get there (every other
// stored activation sits behind a BatchNorm or a LayerNorm), and downstream ReLUs scrub the evidence: fmaxf(NaN, 0) = 0, so an
// overflow rarely survives to the outputs as a NaN.  The kernels that STORE a residual stream (bneck_fused.hip, bneck_tail.hip) track
// whether a value they write became inf -- six VALU operations per 16-byte row piece, fp16 flavour only -- and OR 4 into the caller's health word
// (the `health` argument of their entry points -- gim_amd passes the coarse count buffer's word [1], read back with the match count --
// or NULL: no check) when it is beyond the range.  Round 5: an ARGUMENT, not a process-wide registration: two modules on two streams
// of one device do not share it, and a launch that raises leaves nothing registered.
// The check runs on the PACKED row (16 bytes = 8 stored halves) the kernel is about to write, not on the fp32 values: packed
// fp16 maxima fold it into one word with transient registers only, and the verdict is a bool, i.e. a lane mask in two SGPRs.
// (A float running maximum -- or any test on the accumulators -- costs registers where bneck_tail's 256-channel variants have none:
// they sit at 253-256 VGPRs and spilled 12-17 of them, +0.8 ms per step.)  Values are post-ReLU (>= 0): inf is the largest half;
// v_pk_max_f16 drops a NaN operand, which can only come from an inf that was flagged where it was stored.
typedef _Float16 gim_h2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned h16_pk_max(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(gim_h2_t, a), __builtin_bit_cast(gim_h2_t, b)));
}
__device__ __forceinline__ void h16_range_track(bool& any, const uint4 v) {
#if GIM_HALF_KIND
    const unsigned m = h16_pk_max(h16_pk_max(v.x, v.y), h16_pk_max(v.z, v.w));
    any |= (((m & 0x7FFF7FFFu) + 0x04000400u) & 0x80008000u) != 0u;   // a half with all exponent bits set (inf / NaN)
#endif
}
// the same in two steps: fold rows into a packed maximum, test it once and write the flag on the spot (a rare wave-level branch,
// no state carried through the kernel)
__device__ __forceinline__ unsigned h16_range_fold(unsigned m, const uint4 v) {
#if GIM_HALF_KIND
    return h16_pk_max(h16_pk_max(m, v.x), h16_pk_max(h16_pk_max(v.y, v.z), v.w));
#else
    return m;
#endif
}
__device__ __forceinline__ void h16_range_check(int* health, unsigned m) {
#if GIM_HALF_KIND
    if (__builtin_expect((((m & 0x7FFF7FFFu) + 0x04000400u) & 0x80008000u) != 0u, 0))
        if (health != nullptr) atomicOr(health, 4);
#endif
}
// sign-insensitive variant for stores that are not behind a ReLU (gim_conv2d_bn_act with a residual operand): fold |v| as 16-bit
// integers -- a half's magnitude bits order like unsigned integers, inf = 0x7C00 is the largest finite-or-inf pattern, NaNs lie above
typedef unsigned short gim_us2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned h16_range_fold_abs(unsigned m, const uint4 v) {
#if GIM_HALF_KIND
    auto mx = [](unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(gim_us2_t, a), __builtin_bit_cast(gim_us2_t, b))); };
    return mx(mx(m, v.x & 0x7FFF7FFFu), mx(mx(v.y & 0x7FFF7FFFu, v.z & 0x7FFF7FFFu), v.w & 0x7FFF7FFFu));
#else
    return m;
#endif
}
__device__ __forceinline__ void h16_range_flag(int* health, bool any) {
#if GIM_HALF_KIND
    if (health != nullptr && any) atomicOr(health, 4);
#endif
}

// ---- LDS-DMA through inline asm (invisible to hipcc's s_waitcnt bookkeeping) --------------------------------------------------
// hipcc makes the first LDS access behind an LDS-DMA it can see (__builtin_amdgcn_raw_ptr_buffer_load_lds) wait vmcnt(0): it assumes
// every ds_read / ds_write may alias the DMA's destination.  Kernels that keep a DMA in flight ACROSS other LDS work (bneck_tail.hip,
// the row-panel statistics kernel of coarse_match.hip) issue it through this statement and count vmcnt by hand.  One instruction =
// 64 lanes x 16 B -> 1 KiB at LDS byte address `lds_addr` (wave-uniform), lane-linear; M0 carries the LDS address and is written in
// the same statement that reads it (cdna_hip_programming.md section 5.7).  s_nop 4: descriptor SGPRs may come straight out of
// v_readfirstlane (VALU writes SGPR -> VMEM reads it: 5 wait states the compiler cannot insert inside an asm string).
typedef unsigned gim_u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gim_dma16(const gim_u32x4_t rsrc, unsigned lds_addr, unsigned voff) {
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);   // wave-uniform by construction; the "s" constraint needs it provable
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory", "m0");
}
__device__ __forceinline__ gim_u32x4_t gim_make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    gim_u32x4_t r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}

// ---- phase stamps (development builds: GIM_HIPCC_EXTRA=-DGIM_TIMING; tools/kernel_timing.py) -------------------------------------------
// GIM_TT_DECL(name) in a translation unit declares a stamp array [workgroup][wave][slot] and its reader gim_timing_<name>(host, n_wg);
// GIM_TT(name, wave, slot) stores the shader clock (s_memtime) of lane 0; GIM_TT_ACC adds a duration (persistent kernels: per-phase totals).
#ifdef GIM_TIMING
constexpr int GIM_TT_WG = 8192, GIM_TT_W = 8, GIM_TT_N = 16;
#define GIM_TT_DECL(name)                                                                                               \
    __device__ unsigned long long g_tt_##name[GIM_TT_WG][GIM_TT_W][GIM_TT_N];                                              \
    extern "C" int GIM_FN(gim_timing_##name)(unsigned long long* host, int n_wg) {                                       \
        return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tt_##name), (size_t)(n_wg < GIM_TT_WG ? n_wg : GIM_TT_WG) * GIM_TT_W * GIM_TT_N * 8) == hipSuccess ? 0 : -1; \
    }                                                                                                                   \
    extern "C" int GIM_FN(gim_timing_clear_##name)(void) {                                                               \
        void* p_ = nullptr;                                                                                             \
        if (hipGetSymbolAddress(&p_, HIP_SYMBOL(g_tt_##name)) != hipSuccess) return -1;                                  \
        return hipMemset(p_, 0, sizeof(g_tt_##name)) == hipSuccess ? 0 : -1;                                             \
    }
#define GIM_TT(name, wave, slot) do { if (blockIdx.x < GIM_TT_WG && (threadIdx.x & 63) == 0) g_tt_##name[blockIdx.x][wave][slot] = __builtin_readcyclecounter(); } while (0)
#define GIM_TT_SET(name, wave, slot, v) do { if (blockIdx.x < GIM_TT_WG && (threadIdx.x & 63) == 0) g_tt_##name[blockIdx.x][wave][slot] = (v); } while (0)
#define GIM_TT_NOW() __builtin_readcyclecounter()
#else
#define GIM_TT_DECL(name)
#define GIM_TT(name, wave, slot) do { } while (0)
#define GIM_TT_SET(name, wave, slot, v) do { } while (0)
#define GIM_TT_NOW() 0ull
#endif

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
