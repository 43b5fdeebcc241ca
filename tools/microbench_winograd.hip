// Go / no-go micro-benchmark for Winograd F(2x2, 3x3) on the 3x3 layers: the INNER LOOPS only, on LDS-resident synthetic
// operands, no correctness -- an upper bound for what a real kernel could reach, measured next to the library's direct loop on the same box.
//
// Unit of work = one "output unit": 256 output pixels x 256 output channels x 16 input channels of a 3x3 convolution.
//   direct (the library's 256 x 256 tile, igemm_mainloop.h): K = 9 taps x 16 channels = 144 = 2.25 slabs of 64; per slab and wave 4 K steps x
//     (2 A + 4 B fragment reads, 8 MFMAs), 8 LDS-DMA pieces, one wait, one barrier.                       2.25 x 2048 = 4608 MFMA cycles per SIMD
//   Winograd 2-D: 256 px = 64 tiles of 2 x 2; 16 transform positions, each its own [64 tiles x Cin] . [Cin x Cout] product -> 16 LIVE accumulator
//     sets.  One 32 x 32 accumulator per position is 16 x 16 = 256 registers -- everything a wave owns at two waves per SIMD -- so the positions are
//     split over wave pairs (8 accumulators each): a workgroup covers 64 tiles x 64 channels and NO fragment is re-used from registers: every MFMA
//     reads its own A and B fragment from LDS.  Per K step of 16 channels the workgroup stages U (16 positions x 64 channels x 32 B = 32 KiB of
//     transformed filters for 64 x 64 outputs; the direct loop stages 64 KiB of filters + pixels per 256 x 256 outputs and 64 K values) and runs the
//     input transform (per lane-task 4 x 4 pixels x 8 channels: 16 reads, 128 packed-f16 additions, 16 writes of V): per wave and K step 4 LDS-DMA
//     pieces, 4 reads + 32 v_pk_add_f16 + 4 writes of transform work, 8 x (2 fragment reads + 1 MFMA), wait, barrier.
//     An output unit = 4 channel quarters = 4 loop iterations = 2048 MFMA cycles per SIMD (2.25 x fewer than direct).
//     (Favourable omissions: the 10.9 KiB halo of input pixels per K step is not staged, the output transform -- incl. the exchange between the two
//     position halves -- and the 4 x wider epilogue are free, the input transform is counted once per 64 channels although it only depends on the tile.)
//   Winograd 1-D (F(2,3) along x, the three kernel rows direct): 4 positions, K = 3 rows x 16 channels per position; 128 pixel pairs x 128 channels per
//     workgroup, wave = 64 pairs x 32 channels x 4 positions (8 accumulators = 128 registers): per K step and position 2 A fragments (each 2 reads + 4
//     v_pk_add_f16) + 1 B fragment for 2 MFMAs.  1.5 x fewer MFMAs than direct: 3072 MFMA cycles per output unit.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gim_amd/csrc -I include tools/microbench_winograd.hip -o tools/bin/microbench_winograd
//   tools/bin/microbench_winograd [random|relu]        (gpu_check.sh <tag> cmd=...)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../gim_amd/csrc/gim_common.h"

typedef __attribute__((address_space(3))) void lds_t;

__device__ __forceinline__ unsigned pk_add_f16(unsigned a, unsigned b) {
    unsigned r;
    asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ---- direct: the library's loop (R + D + B of tools/microbench_mainloop.hip), 256 x 256 tile, 4 x 2 waves of 64 x 128 ----------------------------
__global__ void __launch_bounds__(512, 2) direct_kernel(const char* __restrict__ src, unsigned window, int slabs, float* __restrict__ out) {
    constexpr int TM = 2, TN = 4, WM = 4, WN = 2, NW = 8, STAGE = 64 * 1024, NDMA = STAGE / 1024 / NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wm = w / WN, wn = w % WN;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_t*)smem);
    const gim_u32x4_t rs = gim_make_rsrc(src, window);
    unsigned off = (unsigned)(((size_t)blockIdx.x * STAGE * 3) % window);
    const unsigned wmask = window - 1;
    for (int i = threadIdx.x; i < 2 * STAGE / 16; i += NW * 64) *(uint4*)(smem + i * 16) = *(const uint4*)(src + ((size_t)i * 16) % window);
    __syncthreads();
    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int buf = 0;
    for (int s = 0; s < slabs; ++s) {
#pragma unroll
        for (int u = 0; u < NDMA; ++u)
            gim_dma16(rs, lds0 + (unsigned)((buf ^ 1) * STAGE + (w * NDMA + u) * 1024), (off + (unsigned)((w * NDMA + u) * 1024 + lane * 16)) & wmask);
        const char* st = smem + buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8_t fa[TM], fb[TN];
            const int sw = ((2 * ks + lh) ^ (l31 & 7)) << 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *(const bf16x8_t*)(st + ((wm * TM + i) * 32 + l31) * 128 + sw);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *(const bf16x8_t*)(st + ((WM * TM + wn * TN + j) * 32 + l31) * 128 + sw);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma_h16_32x32x16(fa[i], fb[j], acc[i][j]);
        }
        off = (off + STAGE * gridDim.x) & wmask;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        buf ^= 1;
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    if (t == 1234.5f) out[threadIdx.x] = t;
}

// ---- Winograd 2-D (see the header).  XF = 0 leaves the input transform out (how much of the time is it?) -----------------------------------------
// 16 positions x one 32 x 32 accumulator = 256 registers = ALL a wave may have at two waves per SIMD, so the positions are split over wave pairs:
// wave = (position half, tile block, channel block) with 8 accumulators; the workgroup covers 64 tiles x 64 channels x 16 positions per K step of
// 16 input channels: 32 KiB of U staged, 32 KiB of V produced by the input transform, 8 MFMAs per wave.  Output unit = 4 channel quarters = 4 iterations.
template <int XF>
__global__ void __launch_bounds__(512, 2) wino2d_kernel(const char* __restrict__ src, unsigned window, int iters, float* __restrict__ out) {
    constexpr int NW = 8, USTAGE = 32 * 1024, VSTAGE = 32 * 1024, NDMA = USTAGE / 1024 / NW;   // 4 pieces per wave and K step
    extern __shared__ __attribute__((aligned(16))) char smem[];                                // [U0 | U1 | V0 | V1]
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), ph = w & 1, wm = (w >> 1) & 1, wn = w >> 2;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_t*)smem);
    const gim_u32x4_t rs = gim_make_rsrc(src, window);
    unsigned off = (unsigned)(((size_t)blockIdx.x * USTAGE * 3) % window);
    const unsigned wmask = window - 1;
    for (int i = threadIdx.x; i < (2 * USTAGE + 2 * VSTAGE) / 16; i += NW * 64) *(uint4*)(smem + i * 16) = *(const uint4*)(src + ((size_t)i * 16) % window);
    __syncthreads();
    f32x16_t acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    int buf = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < NDMA; ++u)
            gim_dma16(rs, lds0 + (unsigned)((buf ^ 1) * USTAGE + (w * NDMA + u) * 1024), (off + (unsigned)((w * NDMA + u) * 1024 + lane * 16)) & wmask);
        const char* us = smem + buf * USTAGE;                       // [16 positions][64 channels][32 B]
        const char* vs = smem + 2 * USTAGE + buf * VSTAGE;          // [16 positions][64 tiles][32 B]
        char* vn = smem + 2 * USTAGE + (buf ^ 1) * VSTAGE;
        if (XF) {
            // this wave's share of the NEXT K step's input transform (64 tiles x 2 channel groups = 128 lane-tasks of 16 reads, 128 packed
            // additions, 16 writes = per wave 4 reads, 32 additions, 4 writes)
            uint4 d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = *(const uint4*)(us + ((w * 4 + k) * 64 + lane) * 16);
            unsigned e[16] = {d[0].x, d[0].y, d[0].z, d[0].w, d[1].x, d[1].y, d[1].z, d[1].w, d[2].x, d[2].y, d[2].z, d[2].w, d[3].x, d[3].y, d[3].z, d[3].w};
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int k = 0; k < 16; ++k) e[k] = pk_add_f16(e[k], e[(k + 1 + r) & 15]);
#pragma unroll
            for (int k = 0; k < 4; ++k) *(uint4*)(vn + ((w * 4 + k) * 64 + lane) * 16) = make_uint4(e[4 * k], e[4 * k + 1], e[4 * k + 2], e[4 * k + 3]);
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int sw = (lh ^ ((l31 >> 1) & 1)) << 4;           // 32-byte rows: two rows share a bank group
            const bf16x8_t fa = *(const bf16x8_t*)(vs + ((ph * 8 + p) * 64 + wm * 32 + l31) * 32 + sw);
            const bf16x8_t fb = *(const bf16x8_t*)(us + ((ph * 8 + p) * 64 + wn * 32 + l31) * 32 + sw);
            acc[p] = mfma_h16_32x32x16(fa, fb, acc[p]);
        }
        off = (off + USTAGE * gridDim.x) & wmask;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        buf ^= 1;
    }
    float t = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[p][r];
    if (t == 1234.5f) out[threadIdx.x] = t;
}

// ---- Winograd 1-D along x: 4 positions, 128 pairs x 128 channels per workgroup, wave = 64 pairs x 32 channels x 4 positions ----------------------
// one loop iteration = K step of 16 channels of ONE kernel row: stages 4 positions x 128 channels x 32 B = 16 KiB of U and 128 pairs' 4 input
// columns (16 KiB of pixels); the transformed A fragments are built in registers from two pixel reads each (V0 = d0 - d2, V1 = d1 + d2, ...)
__global__ void __launch_bounds__(512, 2) wino1d_kernel(const char* __restrict__ src, unsigned window, int iters, float* __restrict__ out) {
    constexpr int NW = 8, STAGE = 32 * 1024, NDMA = STAGE / 1024 / NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wm = w >> 2, wn = w & 3;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_t*)smem);
    const gim_u32x4_t rs = gim_make_rsrc(src, window);
    unsigned off = (unsigned)(((size_t)blockIdx.x * STAGE * 3) % window);
    const unsigned wmask = window - 1;
    for (int i = threadIdx.x; i < 2 * STAGE / 16; i += NW * 64) *(uint4*)(smem + i * 16) = *(const uint4*)(src + ((size_t)i * 16) % window);
    __syncthreads();
    f32x16_t acc[4][2];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][i][r] = 0.f;
    int buf = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < NDMA; ++u)
            gim_dma16(rs, lds0 + (unsigned)((buf ^ 1) * STAGE + (w * NDMA + u) * 1024), (off + (unsigned)((w * NDMA + u) * 1024 + lane * 16)) & wmask);
        const char* ps = smem + buf * STAGE;              // pixels: [128 pairs][4 columns][32 B]
        const char* us = ps + 16 * 1024;                  // U: [4 positions][128 channels][32 B]
        const int sw = (lh ^ ((l31 >> 1) & 1)) << 4;
        uint4 d[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) d[i][c] = *(const uint4*)(ps + (((wm * 2 + i) * 32 + l31) * 4 + c) * 32 + sw);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const bf16x8_t fb = *(const bf16x8_t*)(us + (p * 128 + wn * 32 + l31) * 32 + sw);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint4 x = d[i][p == 0 ? 0 : 1 + (p & 1)], y = d[i][p == 3 ? 3 : 2 - (p >> 1)];
                const uint4 v = make_uint4(pk_add_f16(x.x, y.x), pk_add_f16(x.y, y.y), pk_add_f16(x.z, y.z), pk_add_f16(x.w, y.w));
                acc[p][i] = mfma_h16_32x32x16(__builtin_bit_cast(bf16x8_t, v), fb, acc[p][i]);
            }
        }
        off = (off + STAGE * gridDim.x) & wmask;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        buf ^= 1;
    }
    float t = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) t += acc[p][i][r];
    if (t == 1234.5f) out[threadIdx.x] = t;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename K>
static double time_kernel(K kern, int smem, int grid, const char* src, size_t window, int iters, float* out) {
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, 0, src, (unsigned)window, iters, out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return best * 1e-3;
}

int main(int argc, char** argv) {
    const char* data = argc > 1 ? argv[1] : "relu";
    int dev = 0, ncu = 256, khz = 2400000;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, dev));
    const double ghz = khz * 1e-6;
    const size_t window = (size_t)2 << 20;
    std::vector<unsigned short> h(window / 2);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3000u + ((x >> 16) & 0x0BFFu) + ((x >> 3) & 0x8000u)); }
    if (data[0] == 'r' && data[1] == 'e')
        for (auto& v : h) { v &= 0x7fffu; x = x * 1664525u + 1013904223u; if (x & 0x10000u) v = 0; }
    char* src = nullptr; float* out = nullptr;
    CK(hipMalloc(&src, window)); CK(hipMemcpy(src, h.data(), window, hipMemcpyHostToDevice)); CK(hipMalloc(&out, 4096 * 4));
    printf("device %d: %d CUs, %.2f GHz nominal, %s operands; cycles at the nominal clock per OUTPUT UNIT (256 px x 256 channels x 16 input channels, 3x3)\n", dev, ncu, ghz, data);
    for (int pass = 0; pass < 2; ++pass) {
        const int cus = pass ? ncu / 8 : ncu;
        printf("-- %d CUs%s\n", cus, pass ? " (no power limit)" : "");
        const int n = 4000;
        const double td = time_kernel(direct_kernel, 128 * 1024, cus, src, window, n, out) / n * ghz * 1e9 * 2.25;        // 2.25 slabs per unit
        const double t2 = time_kernel(wino2d_kernel<1>, 128 * 1024, cus, src, window, n, out) / n * ghz * 1e9 * 4.0;      // 4 channel quarters x 1 K step
        const double t2n = time_kernel(wino2d_kernel<0>, 128 * 1024, cus, src, window, n, out) / n * ghz * 1e9 * 4.0;
        // wino1d: an iteration covers 128 pairs = 256 px, 128 channels, 16 input channels of ONE kernel row: unit = 2 channel halves x 3 rows = 6 iterations
        const double t1u = time_kernel(wino1d_kernel, 64 * 1024, cus, src, window, n, out) / n * ghz * 1e9 * 6.0;
        printf("direct 256 x 256 tile (R + D + B)      %8.0f cycles per output unit  (MFMA alone 4608)\n", td);
        printf("Winograd 2-D, best case                %8.0f cycles per output unit  (MFMA alone 2048)  %.2f x direct\n", t2, td / t2);
        printf("Winograd 2-D without input transform   %8.0f cycles per output unit                      %.2f x direct\n", t2n, td / t2n);
        printf("Winograd 1-D (x only), best case       %8.0f cycles per output unit  (MFMA alone 3072)  %.2f x direct\n", t1u, td / t1u);
    }
    return 0;
}
