// Micro-benchmark behind DESIGN "what the counters said": how many bytes per clock does ONE CU of gfx950 move from L2 (or HBM) into LDS /
// registers, by which instruction?  Every MFMA kernel of this library stages its operands through LDS; their main loops all landed on
// ~13 B per clock and CU.  This program measures the path alone -- no MFMAs, no address arithmetic worth mentioning:
//   mode 0  buffer_load_dwordx4 ... lds   (LDS-DMA, 16 B per lane: gim_dma16, what the kernels use)
//   mode 1  buffer_load_dword ... lds     (LDS-DMA, 4 B per lane)
//   mode 2  global_load_dwordx4 -> VGPR -> ds_write_b128   (the register path)
//   mode 3  global_load_dwordx4 -> VGPR   (no LDS: the L2 -> CU ceiling)
// over a source window that fits L2 (every workgroup re-reads the same few MiB) or streams from HBM (window >> 256 MiB MALL),
// with U KiB in flight per wave, W waves per workgroup, G workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gim_amd/csrc -I include tools/microbench_staging.hip -o tools/bin/microbench_staging
//   gpurun -- 'tools/bin/microbench_staging'          (tools/gpu_check.sh <tag> cmd=tools/bin/microbench_staging)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../gim_amd/csrc/gim_common.h"

typedef __attribute__((address_space(3))) void lds_t;

__device__ __forceinline__ void dma4(const gim_u32x4_t rsrc, unsigned lds_addr, unsigned voff) {
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory", "m0");
}

// One iteration = every wave stages U KiB (U instructions of 64 lanes x 16 B; mode 1: 4 U instructions of 4 B) and waits for them.
template <int MODE, int U>
__global__ void __launch_bounds__(1024) stage_kernel(const char* __restrict__ src, unsigned window, int iters, unsigned* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_t*)smem) + (unsigned)(w * U * 1024);
    const gim_u32x4_t rs = gim_make_rsrc(src, window);
    const unsigned per_it = (unsigned)(nw * U * 1024);                   // bytes one workgroup stages per iteration
    unsigned off = (unsigned)(((size_t)blockIdx.x * per_it * 7) % window);
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        const unsigned base = off + (unsigned)(w * U * 1024) + (unsigned)(lane * 16);
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) gim_dma16(rs, lds0 + u * 1024, (base + u * 1024) % window);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 4 * U; ++u) dma4(rs, lds0 + u * 256, (off + (unsigned)(w * U * 1024) + u * 256 + lane * 4) % window);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            uint4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = *(const uint4*)(src + (base + u * 1024) % window);
            if (MODE == 2) {
#pragma unroll
                for (int u = 0; u < U; ++u) *(uint4*)(smem + w * U * 1024 + u * 1024 + lane * 16) = v[u];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
            }
        }
        off += per_it * (unsigned)gridDim.x;
        if (off >= window) off %= window;
    }
    if (MODE != 3) {                                                    // keep the LDS contents observable
        __syncthreads();
        acc = *(const uint4*)(smem + ((threadIdx.x * 16) % (nw * U * 1024)));
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE, int U>
static void run(const char* src, size_t window, int waves, int wg_per_cu, unsigned* sink, const char* what, int ncu, double ghz) {
    const int lds = waves * U * 1024;
    CK(hipFuncSetAttribute((const void*)stage_kernel<MODE, U>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int grid = ncu * wg_per_cu;
    const size_t per_it = (size_t)grid * waves * U * 1024;
    int iters = (int)((size_t)(6ull << 30) / per_it);                   // ~6 GiB per launch
    if (iters < 8) iters = 8;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((stage_kernel<MODE, U>), dim3(grid), dim3(waves * 64), lds, 0, src, (unsigned)window, iters, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)per_it * iters, s = best * 1e-3;
    printf("%-28s %-6s U=%-2d waves=%-2d wg/cu=%d  %8.1f GB/s  %6.1f B/clk/CU (at %.1f GHz)\n", what, window > (1u << 28) ? "HBM" : "L2", U, waves, wg_per_cu,
           bytes / s * 1e-9, bytes / s / ncu / (ghz * 1e9), ghz);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main() {
    int dev = 0, ncu = 256, khz = 2400000;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, dev));
    const double ghz = khz * 1e-6;
    const size_t big = (size_t)3 << 30;                                  // 3 GiB: streams from HBM
    char* src = nullptr; unsigned* sink = nullptr;
    CK(hipMalloc(&src, big)); CK(hipMemset(src, 1, big)); CK(hipMalloc(&sink, 64));
    printf("device %d: %d CUs, %.2f GHz\n", dev, ncu, ghz);
    const size_t l2 = (size_t)2 << 20;                                   // 2 MiB window: resident in every XCD's 4 MiB L2
    for (int pass = 0; pass < 2; ++pass) {
        const size_t win = pass ? big : l2;
        run<0, 8>(src, win, 8, 1, sink, "lds-dma x4", ncu, ghz);
        run<0, 8>(src, win, 8, 2, sink, "lds-dma x4", ncu, ghz);
        run<0, 4>(src, win, 16, 1, sink, "lds-dma x4", ncu, ghz);
        run<0, 4>(src, win, 8, 2, sink, "lds-dma x4", ncu, ghz);
        run<0, 2>(src, win, 8, 4, sink, "lds-dma x4", ncu, ghz);
        run<0, 16>(src, win, 4, 2, sink, "lds-dma x4", ncu, ghz);
        run<1, 2>(src, win, 8, 2, sink, "lds-dma x1", ncu, ghz);
        run<2, 8>(src, win, 8, 1, sink, "vgpr + ds_write_b128", ncu, ghz);
        run<2, 8>(src, win, 8, 2, sink, "vgpr + ds_write_b128", ncu, ghz);
        run<2, 4>(src, win, 16, 1, sink, "vgpr + ds_write_b128", ncu, ghz);
        run<2, 4>(src, win, 8, 4, sink, "vgpr + ds_write_b128", ncu, ghz);
        run<2, 16>(src, win, 4, 2, sink, "vgpr + ds_write_b128", ncu, ghz);
        run<3, 8>(src, win, 8, 1, sink, "vgpr only", ncu, ghz);
        run<3, 8>(src, win, 8, 2, sink, "vgpr only", ncu, ghz);
        run<3, 8>(src, win, 8, 4, sink, "vgpr only", ncu, ghz);
        run<3, 16>(src, win, 4, 4, sink, "vgpr only", ncu, ghz);
    }
    return 0;
}
