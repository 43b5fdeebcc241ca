// layout probe of v_mfma_f64_16x16x4_f64: which D[i][j] does (lane, register) hold for A lane l = A[l%16][l/16], B lane l = B[l/16][l%16]?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double f64x4_t __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* D) {
    const int l = threadIdx.x;
    f64x4_t acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l % 16) * 4 + l / 16], B[(l / 16) * 16 + l % 16], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = acc[r];
}
int main() {
    double hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 64; ++i) { hA[i] = sin(1.0 + i) ; hB[i] = cos(2.0 + 3 * i); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int kk = 0; kk < 4; ++kk) s += hA[i * 4 + kk] * hB[kk * 16 + j]; ref[i * 16 + j] = s; }
    double *dA, *dB, *dD;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
    hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 5) for (int r = 0; r < 4; ++r) {
        int bi = -1, bj = -1;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (fabs(ref[i * 16 + j] - hD[l * 4 + r]) < 1e-12) { bi = i; bj = j; }
        printf("lane %2d reg %d -> D[%d][%d]\n", l, r, bi, bj);
    }
    return 0;
}
