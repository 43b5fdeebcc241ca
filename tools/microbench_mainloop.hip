// Micro-benchmark of the main loop every MFMA kernel of this library shares, one ingredient at a time (see DESIGN "what the counters said"):
// a 256 x 256 x (64 per slab) tile on one workgroup per CU, operands in a 2 x 64 KiB LDS ring in the kernels' swizzled row layout.
//   bit 0  R  the waves read their fragments from LDS (ds_read_b128) -- otherwise the fragments stay in registers
//   bit 1  D  the next slab is staged by LDS-DMA from an L2-resident window (gim_dma16), waited for at the end of the slab
//   bit 2  B  workgroup barrier per slab
// for wave grids WM x WN with TM x TN 32 x 32 accumulators per wave.  Prints cycles per slab and the fraction of the MFMA peak.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gim_amd/csrc -I include tools/microbench_mainloop.hip -o tools/bin/microbench_mainloop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../gim_amd/csrc/gim_common.h"

typedef __attribute__((address_space(3))) void lds_t;

// NV: that many extra (independent, integer) VALU instructions per K step -- do they hide under the MFMAs?
// DK (how the next slab is staged when MODE has D): 0 LDS-DMA, all instructions at the start of the slab; 1 LDS-DMA, spread over the four K steps;
// 2 global_load_dwordx4 into registers at the start, ds_write_b128 at the end of the slab; 3 LDS-DMA of HALF the slab (is the cost proportional?)
// NL: that many extra LOADER waves issue all the LDS-DMA (the MFMA waves none); PF: fragments of K step k + 1 are read before the MFMAs of step k
// (a second register set) instead of wherever hipcc puts them; PRIO: s_setprio 1 around the MFMAs
// FAR: every FAR-th LDS-DMA instruction of an MFMA wave reads a 2 GiB window instead (first touches that miss L2: HBM latency, little bandwidth)
template <int TM, int TN, int WM, int WN, int MODE, int WG_PER_CU, int DK, int NV, int NL, int PF, int PRIO, int FAR>
__global__ void __launch_bounds__((WM * WN + NL) * 64, ((WM * WN + NL) * WG_PER_CU + 3) / 4)
mainloop_kernel(const char* __restrict__ src, unsigned window, int slabs, float* __restrict__ out, const char* __restrict__ far) {
    constexpr int NW = WM * WN, STAGE = (WM * TM + WN * TN) * 32 * 128, NDMA = STAGE / 1024 / (NL ? NL : NW);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wm = w / WN, wn = w % WN;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_t*)smem);
    const gim_u32x4_t rs = gim_make_rsrc(src, window), rf = gim_make_rsrc(far, 0x80000000u);
    unsigned foff = (unsigned)(((size_t)blockIdx.x * 7919u * 4096u) & 0x7fffffffu);
    unsigned off = (unsigned)(((size_t)blockIdx.x * STAGE * 3) % window);
    // both stages hold finite numbers before the first read
    for (int i = threadIdx.x; i < 2 * STAGE / 16; i += (NW + NL) * 64) *(uint4*)(smem + i * 16) = *(const uint4*)(src + ((size_t)i * 16) % window);
    __syncthreads();
    if (NL && w >= NW) {                                   // loader wave: stage the next slab, wait for it, meet the MFMA waves
        const int lw = w - NW;
        int lbuf = 0;
        const unsigned wmask = window - 1;
        for (int s = 0; s < slabs; ++s) {
#pragma unroll
            for (int u = 0; u < NDMA; ++u)
                gim_dma16(rs, lds0 + (unsigned)((lbuf ^ 1) * STAGE + (lw * NDMA + u) * 1024), (off + (unsigned)((lw * NDMA + u) * 1024 + lane * 16)) & wmask);
            off = (off + STAGE * gridDim.x) & wmask;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            lbuf ^= 1;
        }
        return;
    }
    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8_t fa[TM], fb[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = *(const bf16x8_t*)(smem + ((wm * TM + i) * 32 + l31) * 128 + lh * 16);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = *(const bf16x8_t*)(smem + ((WM * TM + wn * TN + j) * 32 + l31) * 128 + lh * 16);
    int buf = 0;
    unsigned vx = threadIdx.x, vy = blockIdx.x;
    for (int s = 0; s < slabs; ++s) {
        uint4 rv[DK == 2 ? NDMA : 1];
        const unsigned wmask = window - 1;               // the window is a power of two
        auto dma = [&](const int u) __attribute__((always_inline)) {
            if (FAR && u % FAR == 0) {
                gim_dma16(rf, lds0 + (unsigned)((buf ^ 1) * STAGE + (w * NDMA + u) * 1024), (foff + (unsigned)((w * NDMA + u) * 1024 + lane * 16)) & 0x7fffffffu);
                foff += 0x00A00000u;                    // 10 MiB further on every time: never in a cache
            } else
            gim_dma16(rs, lds0 + (unsigned)((buf ^ 1) * STAGE + (w * NDMA + u) * 1024), (off + (unsigned)((w * NDMA + u) * 1024 + lane * 16)) & wmask);
        };
        if ((MODE & 2) && !NL && (DK == 0 || DK == 3)) {
#pragma unroll
            for (int u = 0; u < (DK == 3 ? NDMA / 2 : NDMA); ++u) dma(u);
        }
        if ((MODE & 2) && !NL && DK == 2) {
#pragma unroll
            for (int u = 0; u < NDMA; ++u) rv[u] = *(const uint4*)(src + ((off + (unsigned)((w * NDMA + u) * 1024 + lane * 16)) & wmask));
        }
        const char* st = smem + buf * STAGE;
        auto load = [&](bf16x8_t* a_, bf16x8_t* b_, const int ks) __attribute__((always_inline)) {
            const int sw = ((2 * ks + lh) ^ (l31 & 7)) << 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) a_[i] = *(const bf16x8_t*)(st + ((wm * TM + i) * 32 + l31) * 128 + sw);
#pragma unroll
            for (int j = 0; j < TN; ++j) b_[j] = *(const bf16x8_t*)(st + ((WM * TM + wn * TN + j) * 32 + l31) * 128 + sw);
        };
        auto mma = [&](const bf16x8_t* a_, const bf16x8_t* b_) __attribute__((always_inline)) {
            if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma_h16_32x32x16(a_[i], b_[j], acc[i][j]);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
        };
        if (PF && (MODE & 1)) {
            bf16x8_t ga[TM], gb[TN];
            load(fa, fb, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ks += 2) {
                if ((MODE & 2) && !NL && DK == 1) {
#pragma unroll
                    for (int u = 0; u < NDMA; ++u)
                        if (u * 2 / NDMA == ks / 2) dma(u);
                }
                load(ga, gb, ks + 1);
                __builtin_amdgcn_sched_barrier(0);
                mma(fa, fb);
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 2 < 4) load(fa, fb, ks + 2);
                __builtin_amdgcn_sched_barrier(0);
                mma(ga, gb);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if ((MODE & 2) && !NL && DK == 1) {
#pragma unroll
                for (int u = 0; u < NDMA; ++u)
                    if (u * 4 / NDMA == ks) dma(u);
            }
            if (MODE & 1) load(fa, fb, ks);
#pragma unroll
            for (int v = 0; v < NV; ++v) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(vx) : "v"(vy)); }
            mma(fa, fb);
        }
        }
        if ((MODE & 2) && !NL && DK == 2) {
#pragma unroll
            for (int u = 0; u < NDMA; ++u) *(uint4*)(smem + (buf ^ 1) * STAGE + (w * NDMA + u) * 1024 + lane * 16) = rv[u];
        }
        if ((MODE & 2) && !NL) { off = (off + STAGE * gridDim.x) & wmask; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        if (MODE & 4) __syncthreads();
        buf ^= 1;
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    if (t == 1234.5f || vx == 0x7fffffffu) out[threadIdx.x] = t;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static const char* g_far = nullptr;
template <int TM, int TN, int WM, int WN, int MODE, int WG_PER_CU = 1, int DK = 0, int NV = 0, int NL = 0, int PF = 0, int PRIO = 0, int FAR = 0>
static void run(const char* src, size_t window, float* out, int ncu, double ghz) {
    constexpr int STAGE = (WM * TM + WN * TN) * 32 * 128;
    auto kern = mainloop_kernel<TM, TN, WM, WN, MODE, WG_PER_CU, DK, NV, NL, PF, PRIO, FAR>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));
    const int slabs = 4000, grid = ncu * WG_PER_CU;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3((WM * WN + NL) * 64), 2 * STAGE, 0, src, (unsigned)window, slabs, out, g_far);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double flops = 2.0 * (WM * TM * 32) * (WN * TN * 32) * 64.0 * slabs * grid, s = best * 1e-3, peak = 2500.0 * ncu / 256.0;
    const double mfma_cycles = (double)TM * TN * 4 * 32 * (WM * WN / 4.0) * WG_PER_CU;   // per slab and SIMD at 1024 flop per clock
    printf("tile %3d x %3d  waves %d x %d (%d x %d frags)  wg/cu %d  %c%c%c %s +%2d valu %d loaders%s%s far 1/%d  %7.0f cycles/slab (MFMA alone %5.0f)  %6.1f TFLOP/s = %.3f of the %d CUs' peak\n", WM * TM * 32, WN * TN * 32, WM, WN,
           TM, TN, WG_PER_CU, MODE & 1 ? 'R' : '-', MODE & 2 ? 'D' : '-', MODE & 4 ? 'B' : '-', !(MODE & 2) ? "        " : DK == 0 ? "dma@0   " : DK == 1 ? "dma/4   " : DK == 2 ? "vgpr    " : "dma half", NV, NL, PF ? " prefetch" : "         ", PRIO ? " prio" : "     ", FAR, s / slabs * ghz * 1e9, mfma_cycles, flops / s * 1e-12, flops / s * 1e-12 / peak, ncu);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
    const char* data = argc > 1 ? argv[1] : "random";   // random | relu (half of the A-side values zero, the others positive) | zero
    int dev = 0, ncu = 256, khz = 2400000;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, dev));
    const double ghz = khz * 1e-6;
    const size_t window = (size_t)(argc > 2 ? atoi(argv[2]) : 2) << 20;   // MiB, a power of two: 2 = L2-resident in every XCD, 64 = MALL, 2048 = HBM
    std::vector<unsigned short> h(window / 2);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3000u + ((x >> 16) & 0x0BFFu) + ((x >> 3) & 0x8000u)); }   // fp16 / bf16 of moderate size, both signs
    if (data[0] == 'r' && data[1] == 'e')            // activations behind a ReLU: no negative values, half of them zero (both operands: the window is shared)
        for (auto& v : h) { v &= 0x7fffu; x = x * 1664525u + 1013904223u; if (x & 0x10000u) v = 0; }
    if (data[0] == 'z') for (auto& v : h) v = 0;
    char* src = nullptr; float* out = nullptr;
    CK(hipMalloc(&src, window)); CK(hipMemcpy(src, h.data(), window, hipMemcpyHostToDevice)); CK(hipMalloc(&out, 4096 * 4));
    { char* f = nullptr; CK(hipMalloc(&f, (size_t)2 << 30)); CK(hipMemset(f, 0x3c, (size_t)2 << 30)); g_far = f; }
    printf("device %d: %d CUs, %.2f GHz nominal, %s operands, %zu MiB source window\n", dev, ncu, ghz, data, window >> 20);
    for (int pass = 0; pass < 2; ++pass) {
        const int cus = pass ? ncu / 8 : ncu;
        printf("-- %d CUs%s\n", cus, pass ? " (4 per XCD: no power limit)" : " (power-limited with random operands)");
        run<2, 4, 4, 2, 0>(src, window, out, cus, ghz);
        run<2, 4, 4, 2, 1>(src, window, out, cus, ghz);
        run<2, 4, 4, 2, 1, 1, 0, 0, 0, 1>(src, window, out, cus, ghz);
        run<2, 4, 4, 2, 1, 1, 0, 0, 0, 1, 1>(src, window, out, cus, ghz);
        run<2, 4, 4, 2, 7>(src, window, out, cus, ghz);
        run<2, 4, 4, 2, 7, 1, 0, 0, 0, 0, 0, 8>(src, window, out, cus, ghz);
        run<2, 4, 4, 2, 7, 1, 0, 0, 0, 0, 0, 4>(src, window, out, cus, ghz);
        run<2, 4, 4, 2, 7, 1, 0, 0, 0, 0, 0, 2>(src, window, out, cus, ghz);
        run<2, 4, 4, 2, 7, 1, 0, 0, 0, 1>(src, window, out, cus, ghz);
        run<2, 4, 4, 2, 7, 1, 1, 0, 0, 1>(src, window, out, cus, ghz);
        run<2, 4, 4, 2, 7, 1, 0, 0, 0, 0, 1>(src, window, out, cus, ghz);
        run<2, 4, 4, 2, 7, 1, 0, 0, 4, 0>(src, window, out, cus, ghz);
        run<2, 4, 4, 2, 7, 1, 0, 0, 4, 0, 1>(src, window, out, cus, ghz);
        run<2, 4, 4, 2, 7, 1, 0, 0, 4, 1, 1>(src, window, out, cus, ghz);
        run<2, 4, 4, 2, 7, 1, 0, 0, 2, 0, 1>(src, window, out, cus, ghz);
        run<2, 4, 4, 2, 7, 1, 0, 0, 1, 0, 1>(src, window, out, cus, ghz);
    }
    return 0;
}
