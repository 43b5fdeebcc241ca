// Depthwise 5x5 + BN + ReLU (gim_dwconv5x5_bn_relu) alone on the shapes of one gim_dkm match().  (Round 4 also built ablations of the
// kernel's ingredients -- no FMAs / no global loads / no stores, profiles/r04_dwconv.txt -- through GIM_DW_ABL blocks inside dkm.hip; round 6
// removed those blocks from the product source, git history has them.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_dwconv.hip gim_amd/csrc/runtime.hip -o /tmp/dw && /tmp/dw
#include "../gim_amd/csrc/dkm.hip"
#include <stdio.h>
#include <vector>

int main() {
    struct Shape { int B, H, W, C; const char* what; };
    const Shape shapes[] = {{2, 1152, 1536, 24, "scale 1, upsampling pass"}, {2, 576, 768, 144, "scale 2, upsampling pass"},
                            {2, 288, 384, 569, "scale 4, upsampling pass"}, {2, 144, 192, 1137, "scale 8, upsampling pass"},
                            {2, 336, 448, 144, "scale 2"}, {2, 168, 224, 569, "scale 4"}};
    hipStream_t s;
    hipStreamCreate(&s);
    for (const Shape& sh : shapes) {
        const int cpad = (sh.C + 7) / 8 * 8;
        const size_t n = (size_t)sh.B * sh.H * sh.W * cpad;
        unsigned short *x, *y;
        float *w, *sc, *sf;
        hipMalloc(&x, n * 2); hipMalloc(&y, n * 2);
        hipMalloc(&w, 25 * cpad * 4); hipMalloc(&sc, cpad * 4); hipMalloc(&sf, cpad * 4);
        std::vector<unsigned short> hx(n);
        for (size_t i = 0; i < n; ++i) hx[i] = (unsigned short)(0x3f00 + (i * 2654435761u >> 25));   // bf16 values around 0.5 .. 1
        hipMemcpy(x, hx.data(), n * 2, hipMemcpyHostToDevice);
        std::vector<float> hw(25 * cpad, 0.04f), h1(cpad, 1.f), h0(cpad, 0.f);
        hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(sc, h1.data(), cpad * 4, hipMemcpyHostToDevice);
        hipMemcpy(sf, h0.data(), cpad * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) gim_dwconv5x5_bn_relu(x, w, sc, sf, y, sh.B, sh.H, sh.W, sh.C, sh.C, cpad, cpad, cpad, GIM_BF16, s);
        hipEventRecord(e0, s);
        const int it = 20;
        for (int i = 0; i < it; ++i) gim_dwconv5x5_bn_relu(x, w, sc, sf, y, sh.B, sh.H, sh.W, sh.C, sh.C, cpad, cpad, cpad, GIM_BF16, s);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned short probe[8];
        hipMemcpy(probe, y + ((size_t)(sh.H / 2) * sh.W + sh.W / 2) * cpad, 16, hipMemcpyDeviceToHost);
        const double us = ms / it * 1e3, mb = n * 4 / 1e6;
        printf("%2d x %4d x %4d x %4d (%-24s): %7.1f us  %6.0f MB in + out  %5.2f TB/s   y[mid] = 0x%04x  (%s)\n", sh.B, sh.H, sh.W, sh.C, sh.what, us, mb,
               mb / us / 1e3, probe[0], gim_last_error());
        hipFree(x); hipFree(y); hipFree(w); hipFree(sc); hipFree(sf);
    }
    return 0;
}
