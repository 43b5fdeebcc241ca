// GP posterior mean of the gim_dkm path (networks/dkm/models/dkm.py:340-370):  mu = K_xy (K_yy + sigma I)^-1 f.
//
// The reference inverts the n x n kernel matrix (n = 2352 at 672x896) with torch.linalg.inv (LU, fp32) and multiplies.
// K_yy + sigma I is symmetric positive definite (exp-cosine kernel + 0.1 I), so here the system is SOLVED:
// blocked right-looking Cholesky A = L L^T and two blocked triangular solves with the 256 right-hand sides, all in
// fp64 (condition number up to ~2e4: an fp32 factorisation and the reference's fp32 LU agree to ~1e-3 only; fp64
// puts this side's error far below the reference's own).  The n^3/3 flops are plain fp64 FMAs in 64 x 64 LDS tiles;
// the cost is dominated by the ~260 small dependent launches (n/64 panel steps), a few ms per call.
//
//   gp_solve(K [B][n][ldk] fp32 (K_yy with sigma already on the diagonal), F [B][n][nrhs] fp32)
//        -> Xt [B][nrhs][npad] fp32 = ((K)^-1 F)^T, zero padded to npad: the [N][K] "weight" layout the igemm takes
//           for the final mu = K_xy X product.
#include "gim_common.h"

namespace {

constexpr int NB = 64;

__global__ void to_f64_kernel(const float* __restrict__ src, double* __restrict__ dst, int rows, int cols, int lds_, int ldd) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)rows * cols) return;
    const size_t r = idx / cols, c = idx - r * cols;
    dst[r * ldd + c] = (double)src[r * lds_ + c];
}

// Cholesky of the nb x nb diagonal block at k0 (lower) and its inverse Linv = L_kk^-1 ([NB][NB] row-major, zero
// padded), by ONE WAVE per matrix: lane i owns row i (factorisation) / column j (inversion).  No workgroup barriers --
// LDS accesses of a wave execute in order.  (A 256-thread version with three __syncthreads per column took 166 us per
// block and ran on 2 CUs while the whole chip waited.)  With Linv every triangular solve against the diagonal block
// becomes a 64-wide GEMM.
__device__ __forceinline__ double readlane_f64(double v, int src) {   // src wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// Second version.  The first one was left-looking column by column with the block in LDS: 64 dependent steps, each a dot
// product whose LDS reads form a latency chain (111 us per block, the longest serial piece of the whole solve).  Now in
// panels of 16 columns: the contribution of all earlier columns to a panel is one batched rank-k update (per k one row
// read + 16 broadcast reads feed 16 independent FMAs, nothing waits on a previous result), the 16 panel columns live in
// registers, and inside the panel pivots / multipliers travel by v_readlane (SGPR broadcast) instead of LDS round trips.
// The inverse uses the same shape: row panels of 16, batched update from the finished rows, 16 in-register steps.
__global__ void __launch_bounds__(64) potrf_diag_kernel(double* __restrict__ A, int n, int k0, int nb, size_t bstride, int* __restrict__ info,
                                                        double* __restrict__ Linv, size_t lstride) {
    __shared__ double T[NB][NB + 1];
    __shared__ double Ti[NB][NB + 1];
    __shared__ double Rinv[NB];
    double* a = A + blockIdx.x * bstride;
    const int lane = threadIdx.x;
    for (int r = 0; r < NB; ++r) T[r][lane] = (r < nb && lane <= r) ? a[(size_t)(k0 + r) * n + k0 + lane] : 0.0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const int i = lane;
#pragma unroll 1
    for (int pb = 0; pb < NB / 16; ++pb) {
        const int c0 = pb * 16;
        double p[16];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) p[jj] = T[i][c0 + jj];
#pragma unroll 2
        for (int k = 0; k < c0; ++k) {
            const double ak = T[i][k];
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) p[jj] = fma(-ak, T[c0 + jj][k], p[jj]);
        }
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int j = c0 + jj;
            double d = readlane_f64(p[jj], j);
            const bool live = j < nb;
            if (live && !(d > 0.0)) { if (lane == 0) info[blockIdx.x] = k0 + j + 1; d = 1.0; }
            if (!live) d = 1.0;
            const double sd = sqrt(d), rinv = 1.0 / sd;
            const double l = !live ? 0.0 : (i == j ? sd : (i > j ? p[jj] * rinv : 0.0));
            p[jj] = l;
            if (i == j) Rinv[j] = live ? rinv : 0.0;
#pragma unroll
            for (int kk = jj + 1; kk < 16; ++kk) p[kk] = fma(-l, readlane_f64(l, c0 + kk), p[kk]);
        }
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) T[i][c0 + jj] = p[jj];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    for (int r = 0; r < nb; ++r)
        if (lane <= r) a[(size_t)(k0 + r) * n + k0 + lane] = T[r][lane];
    // inverse, column j = lane: forward substitution L x = e_j in row panels of 16
    const int j = lane;
#pragma unroll 1
    for (int rb = 0; rb < NB / 16; ++rb) {
        const int r0 = rb * 16;
        double sacc[16];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) sacc[rr] = (r0 + rr == j) ? 1.0 : 0.0;
#pragma unroll 2
        for (int k = 0; k < r0; ++k) {
            const double xk = Ti[k][j];
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) sacc[rr] = fma(-T[r0 + rr][k], xk, sacc[rr]);
        }
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const double xv = sacc[rr] * Rinv[r0 + rr];
            sacc[rr] = xv;
#pragma unroll
            for (int r2 = rr + 1; r2 < 16; ++r2) sacc[r2] = fma(-T[r0 + r2][r0 + rr], xv, sacc[r2]);
        }
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) Ti[r0 + rr][j] = sacc[rr];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    double* li = Linv + blockIdx.x * lstride + (size_t)(k0 / NB) * NB * NB;
    for (int r = 0; r < NB; ++r) li[r * NB + lane] = Ti[r][lane];
}

// C[i][c] = sum_k Aop[i][k] * Bop[k][c]  (same operand conventions as gemm_sub_kernel, K = 64, plain store)
__global__ void __launch_bounds__(256) gemm_set_kernel(double* __restrict__ C, int ldc, const double* __restrict__ Am, int lda, int ta,
                                                       const double* __restrict__ Bm, int ldb, int tb, int M, int N, int kk,
                                                       size_t cstride, size_t astride, size_t bstride2) {
    __shared__ double As[NB][NB + 1];
    __shared__ double Bs[NB][NB + 1];
    const int t = threadIdx.x, i0 = blockIdx.y * NB, c0 = blockIdx.x * NB;
    double* c = C + blockIdx.z * cstride;
    const double* am = Am + blockIdx.z * astride;
    const double* bm = Bm + blockIdx.z * bstride2;
    for (int e = t; e < NB * NB; e += 256) {
        if (ta) { const int k = e / NB, i = e - k * NB; As[i][k] = (k < kk && i0 + i < M) ? am[(size_t)k * lda + i0 + i] : 0.0; }
        else { const int i = e / NB, k = e - i * NB; As[i][k] = (k < kk && i0 + i < M) ? am[(size_t)(i0 + i) * lda + k] : 0.0; }
        if (tb) { const int cc = e / NB, k = e - cc * NB; Bs[k][cc] = (k < kk && c0 + cc < N) ? bm[(size_t)(c0 + cc) * ldb + k] : 0.0; }
        else { const int k = e / NB, cc = e - k * NB; Bs[k][cc] = (k < kk && c0 + cc < N) ? bm[(size_t)k * ldb + c0 + cc] : 0.0; }
    }
    __syncthreads();
    const int ti = (t >> 4) * 4, tc = (t & 15) * 4;
    double acc[4][4] = {};
    for (int k = 0; k < kk; ++k) {
        double av[4], bv[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) { av[x] = As[ti + x][k]; bv[x] = Bs[k][tc + x]; }
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) acc[x][y] = fma(av[x], bv[y], acc[x][y]);
    }
    __syncthreads();  // all reads of the inputs are done: C may alias Am or Bm (in-place panel / right-hand-side update)
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const int i = i0 + ti + x, cc = c0 + tc + y;
            if (i < M && cc < N) c[(size_t)i * ldc + cc] = acc[x][y];
        }
}

// C[i][c] -= sum_k Aop[i][k] * Bop[k][c] on a 64 x 64 tile, K = kk (<= 64).
//   Aop[i][k] = ta ? Am[k*lda + i] : Am[i*lda + k]      Bop[k][c] = tb ? Bm[c*ldb + k] : Bm[k*ldb + c]
// lower_only: skip tiles strictly above the diagonal (symmetric rank-k update of the trailing matrix)
__global__ void __launch_bounds__(256) gemm_sub_kernel(double* __restrict__ C, int ldc, const double* __restrict__ Am, int lda, int ta,
                                                       const double* __restrict__ Bm, int ldb, int tb, int M, int N, int kk,
                                                       int lower_only, size_t cstride, size_t astride, size_t bstride2) {
    if (lower_only && blockIdx.x > blockIdx.y) return;
    __shared__ double As[NB][NB + 1];  // [i][k]
    __shared__ double Bs[NB][NB + 1];  // [k][c]
    const int t = threadIdx.x, i0 = blockIdx.y * NB, c0 = blockIdx.x * NB;
    double* c = C + blockIdx.z * cstride;
    const double* am = Am + blockIdx.z * astride;
    const double* bm = Bm + blockIdx.z * bstride2;
    for (int e = t; e < NB * NB; e += 256) {
        // coalesce along the contiguous source dimension
        if (ta) { const int k = e / NB, i = e - k * NB; As[i][k] = (k < kk && i0 + i < M) ? am[(size_t)k * lda + i0 + i] : 0.0; }
        else { const int i = e / NB, k = e - i * NB; As[i][k] = (k < kk && i0 + i < M) ? am[(size_t)(i0 + i) * lda + k] : 0.0; }
        if (tb) { const int cc = e / NB, k = e - cc * NB; Bs[k][cc] = (k < kk && c0 + cc < N) ? bm[(size_t)(c0 + cc) * ldb + k] : 0.0; }
        else { const int k = e / NB, cc = e - k * NB; Bs[k][cc] = (k < kk && c0 + cc < N) ? bm[(size_t)k * ldb + c0 + cc] : 0.0; }
    }
    __syncthreads();
    const int ti = (t >> 4) * 4, tc = (t & 15) * 4;
    double acc[4][4] = {};
    for (int k = 0; k < kk; ++k) {
        double av[4], bv[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) { av[x] = As[ti + x][k]; bv[x] = Bs[k][tc + x]; }
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) acc[x][y] = fma(av[x], bv[y], acc[x][y]);
    }
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const int i = i0 + ti + x, cc = c0 + tc + y;
            if (i < M && cc < N) c[(size_t)i * ldc + cc] -= acc[x][y];
        }
}

// Xt[b][c][i] = (float) X[b][i][c], zero for i in [n, npad)
__global__ void store_xt_kernel(const double* __restrict__ X, float* __restrict__ Xt, int n, int nrhs, int npad) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z, i0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int i = i0 + r, c = c0 + tx;
        tile[r][tx] = (i < n && c < nrhs) ? (float)X[((size_t)b * n + i) * nrhs + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r, i = i0 + tx;
        if (c < nrhs && i < npad) Xt[((size_t)b * nrhs + c) * npad + i] = tile[tx][r];
    }
}

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" int64_t gim_gp_solve_ws_bytes(int B, int n, int nrhs) {
    const size_t nblk = (n + NB - 1) / NB;
    return (int64_t)(al((size_t)B * n * n * 8) + al((size_t)B * n * nrhs * 8) + al((size_t)B * nblk * NB * NB * 8) +
                     al((size_t)B * NB * (nrhs > n ? nrhs : n) * 8) + 256);
}

extern "C" int gim_gp_solve(const float* K, const float* F, float* Xt, void* ws, int B, int n, int ldk, int nrhs, int npad,
                            gim_stream_t stream) {
    GIM_REQUIRE(K && F && Xt && ws && B > 0 && B <= 16 && n > 0 && nrhs > 0 && ldk >= n && npad >= n, "gp_solve: bad args");
    hipStream_t s = (hipStream_t)stream;
    const size_t nblk_ = (n + NB - 1) / NB;
    double* A = (double*)ws;
    double* R = (double*)((char*)A + al((size_t)B * n * n * 8));
    double* Li = (double*)((char*)R + al((size_t)B * n * nrhs * 8));
    double* Tmp = (double*)((char*)Li + al((size_t)B * nblk_ * NB * NB * 8));   // [B][NB][max(n, nrhs)] staging for in-place products
    int* info = (int*)((char*)Tmp + al((size_t)B * NB * (nrhs > n ? nrhs : n) * 8));
    const size_t as = (size_t)n * n, fs = (size_t)n * nrhs, ls = nblk_ * NB * NB, ts = (size_t)NB * (nrhs > n ? nrhs : n);
    if (hipMemsetAsync(info, 0, 64, s) != hipSuccess) return gim_check_launch("gp_solve memset");
    for (int b = 0; b < B; ++b) {
        hipLaunchKernelGGL(to_f64_kernel, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, s, K + (size_t)b * n * ldk, A + b * as, n, n, ldk, n);
        hipLaunchKernelGGL(to_f64_kernel, dim3((unsigned)(((size_t)n * nrhs + 255) / 256)), dim3(256), 0, s, F + (size_t)b * fs, R + b * fs, n, nrhs, nrhs, nrhs);
    }
    // ---- A = L L^T (lower triangle of A overwritten by L) ----
    for (int k0 = 0; k0 < n; k0 += NB) {
        const int nb = n - k0 < NB ? n - k0 : NB;
        hipLaunchKernelGGL(potrf_diag_kernel, dim3(B), dim3(64), 0, s, A, n, k0, nb, as, info, Li, ls);
        const int m = n - k0 - nb;
        if (m <= 0) break;
        const int tiles = (m + NB - 1) / NB;
        const size_t off = (size_t)(k0 + nb) * n;
        const double* li = Li + (size_t)(k0 / NB) * NB * NB;
        // panel: P <- P L_kk^-T  = P (Linv)^T : Aop = P rows, Bop[k][c] = Linv[c][k]   (in place: one 64-column tile per row tile)
        hipLaunchKernelGGL(gemm_set_kernel, dim3(1, tiles, B), dim3(256), 0, s, A + off + k0, n, A + off + k0, n, 0, li, NB, 1, m, nb, nb, as, as, ls);
        // trailing[i][j] -= sum_k P[i][k] P[j][k]
        hipLaunchKernelGGL(gemm_sub_kernel, dim3(tiles, tiles, B), dim3(256), 0, s, A + off + k0 + nb, n, A + off + k0, n, 0,
                           A + off + k0, n, 1, m, m, nb, 1, as, as, as);
    }
    // ---- L Y = F ----
    const int ctiles = (nrhs + NB - 1) / NB;
    for (int k0 = 0; k0 < n; k0 += NB) {
        const int nb = n - k0 < NB ? n - k0 : NB;
        const double* li = Li + (size_t)(k0 / NB) * NB * NB;
        // Y_k = Linv F_k
        // in place: every workgroup stages its whole 64 x 64 column tile of F_k in LDS before it stores
        hipLaunchKernelGGL(gemm_set_kernel, dim3(ctiles, 1, B), dim3(256), 0, s, R + (size_t)k0 * nrhs, nrhs, li, NB, 0, R + (size_t)k0 * nrhs, nrhs, 0, nb, nrhs, nb, fs, ls, fs);
        const int m = n - k0 - nb;
        if (m <= 0) break;
        hipLaunchKernelGGL(gemm_sub_kernel, dim3(ctiles, (m + NB - 1) / NB, B), dim3(256), 0, s, R + (size_t)(k0 + nb) * nrhs, nrhs,
                           A + (size_t)(k0 + nb) * n + k0, n, 0, R + (size_t)k0 * nrhs, nrhs, 0, m, nrhs, nb, 0, fs, as, fs);
    }
    // ---- L^T X = Y ----
    const int nblk = (n + NB - 1) / NB;
    for (int kb = nblk - 1; kb >= 0; --kb) {
        const int k0 = kb * NB, nb = n - k0 < NB ? n - k0 : NB;
        const double* li = Li + (size_t)kb * NB * NB;
        // X_k = Linv^T Y_k : Aop[i][k] = Linv[k][i]
        hipLaunchKernelGGL(gemm_set_kernel, dim3(ctiles, 1, B), dim3(256), 0, s, R + (size_t)k0 * nrhs, nrhs, li, NB, 1, R + (size_t)k0 * nrhs, nrhs, 0, nb, nrhs, nb, fs, ls, fs);
        if (k0 == 0) break;
        hipLaunchKernelGGL(gemm_sub_kernel, dim3(ctiles, (k0 + NB - 1) / NB, B), dim3(256), 0, s, R, nrhs,
                           A + (size_t)k0 * n, n, 1, R + (size_t)k0 * nrhs, nrhs, 0, k0, nrhs, nb, 0, fs, as, fs);
    }
    hipLaunchKernelGGL(store_xt_kernel, dim3((npad + 63) / 64, (nrhs + 63) / 64, B), dim3(256), 0, s, R, Xt, n, nrhs, npad);
    return gim_check_launch("gp_solve");
}
