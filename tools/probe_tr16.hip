// ds_read_b64_tr_b16 on gfx950: which 16-bit elements does lane l receive when lane l addresses the 4 elements 4 l .. 4 l + 3?
//   hipcc --offload-arch=gfx950 -O3 tools/probe_tr16.hip -o /tmp/probe_tr16 && /tmp/probe_tr16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + l * 4));
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
int main() {
    short* d; short h[256];
    hipMalloc(&d, 512);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d   (= lane,element of the source rows: %d.%d %d.%d %d.%d %d.%d)\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3],
                                        h[4 * l] / 4, h[4 * l] % 4, h[4 * l + 1] / 4, h[4 * l + 1] % 4, h[4 * l + 2] / 4, h[4 * l + 2] % 4, h[4 * l + 3] / 4, h[4 * l + 3] % 4);
    return 0;
}
