// Issue rate of the VALU instructions a depthwise convolution could be built from, gfx950: cycles per wave instruction at 1, 2, 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_valu.hip -o /tmp/mb_valu && /tmp/mb_valu
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int KIND>
__global__ void __launch_bounds__(1024) k(float* out, int iters, unsigned long long* cyc) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = (f2){(float)threadIdx.x, (float)i};
    f2 x = {1.0001f, 0.9999f}, y = {1e-6f, -1e-6f};
    unsigned xi = 0x3c003c00u + threadIdx.x, yi = 0x3c003c00u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
#define FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(x.x), "v"(y.x));
#define FMA2(i) asm volatile("v_fma_f32 %0, %2, %3, %0\n\tv_fma_f32 %1, %2, %3, %1" : "+v"(a[i].x), "+v"(a[i].y) : "v"(x.x), "v"(y.x));
#define DOT2H(i) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(xi), "v"(yi));
#define DOT2B(i) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(xi), "v"(yi));
#define DOT2CH(i) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[i].x) : "v"(xi), "v"(yi));
#define DOT2CB(i) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[i].x) : "v"(xi), "v"(yi));
#define PKFMAH(i) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(xi), "v"(yi));
#define PERM(i) asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(xi), "v"(yi));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
#define CVT(i) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[i].x) : "v"(xi));
        if constexpr (KIND == 0) { REP8(PKFMA) REP8(PKFMA) }
        if constexpr (KIND == 1) { REP8(FMA) REP8(FMA) }
        if constexpr (KIND == 2) { REP8(FMA2) }
        if constexpr (KIND == 3) { REP8(DOT2H) REP8(DOT2H) }
        if constexpr (KIND == 4) { REP8(DOT2B) REP8(DOT2B) }
        if constexpr (KIND == 5) { REP8(DOT2CH) REP8(DOT2CH) }
        if constexpr (KIND == 6) { REP8(DOT2CB) REP8(DOT2CB) }
        if constexpr (KIND == 7) { REP8(PKFMAH) REP8(PKFMAH) }
        if constexpr (KIND == 8) { REP8(PERM) REP8(PERM) }
        if constexpr (KIND == 9) { REP8(PKMUL) REP8(PKMUL) }
        if constexpr (KIND == 10) { REP8(CVT) REP8(CVT) }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
void run(const char* name, int waves_per_simd) {
    float* out; unsigned long long* cyc; unsigned long long h;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    const int iters = 32768, threads = 256 * waves_per_simd;   // one workgroup per CU, `waves_per_simd` waves on each SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double ninst = (double)iters * 16;
    printf("%-18s %d wave(s)/SIMD: %7.2f shader cycles per wave instruction (s_memtime of wave 0: %llu over %.0f instr), %.3f ms => %.2f cycles per instr and SIMD at 2.4 GHz\n",
           name, waves_per_simd, (double)h / ninst, h, ninst, ms, ms * 1e-3 * 2.4e9 / (ninst * waves_per_simd));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("v_pk_fma_f32", w); run<1>("v_fma_f32", w); run<2>("2 x v_fma_f32", w); run<3>("v_dot2_f32_f16", w); run<4>("v_dot2_f32_bf16", w);
        run<5>("v_dot2c_f32_f16", w); run<6>("v_dot2c_f32_bf16", w); run<7>("v_pk_fma_f16", w); run<8>("v_perm_b32", w); run<9>("v_pk_mul_f32", w); run<10>("v_cvt_f32_f16", w);
    }
    return 0;
}
