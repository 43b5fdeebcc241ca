// GP posterior mean of the gim_dkm path (networks/dkm/models/dkm.py:340-370):  mu = K_xy (K_yy + sigma I)^-1 f.
//
// The reference inverts the n x n kernel matrix (n = 2352 at 672x896) with torch.linalg.inv (LU, fp32) and multiplies.
// K_yy + sigma I is symmetric positive definite (exp-cosine kernel + 0.1 I), so here the system is SOLVED, all in fp64
// (condition number up to ~2e4: an fp32 factorisation and the reference's fp32 LU agree to ~1e-3 only; fp64 puts this
// side's error far below the reference's own):
//   1. blocked right-looking Cholesky A = L L^T -- n/64 dependent steps (diagonal block factored and inverted by one wave,
//      panel and trailing update as 64-wide fp64 GEMM tiles): the serial part, ~3.5 ms at n = 2304;
//   2. L^-1 completed from the inverted diagonal blocks by recursive doubling (two batched GEMMs per level);
//   3. X = L^-T (L^-1 F) as two triangular GEMMs with the 256 / 512 right-hand sides.
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
// Round 3: TWO waves.  The inverse of row panel rb needs rows 16 rb .. 16 rb + 15 of L complete, i.e. factor panels 0 .. rb -- nothing
// more -- so wave 1 inverts panel rb while wave 0 factors panel rb + 1 (progress word in LDS: a wave's LDS accesses execute in order, so
// whoever sees the word sees the panel).  The chain shrinks from factor + inverse to factor + the last inverse panel.
__global__ void __launch_bounds__(128) potrf_diag_kernel(double* __restrict__ A, int n, int k0, int nb, size_t bstride, int* __restrict__ info,
                                                         double* __restrict__ Linv, size_t lstride, int ldl) {
    __shared__ double T[NB][NB + 1];
    __shared__ double Ti[NB][NB + 1];
    __shared__ double Rinv[NB];
    __shared__ int prog;          // factor panels finished by wave 0
    double* a = A + blockIdx.x * bstride;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) prog = 0;
    // all loads of a half block are in flight before the first LDS write (a rolled load -> store loop: 64 dependent round trips);
    // wave w loads rows 32 w .. 32 w + 31
    {
        const int h = wave * 32;
        double tmp[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) tmp[r] = (h + r < nb && lane <= h + r) ? a[(size_t)(k0 + h + r) * n + k0 + lane] : 0.0;
#pragma unroll
        for (int r = 0; r < 32; ++r) T[h + r][lane] = tmp[r];
    }
    __syncthreads();
    if (wave == 0) {
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
                // 1 / sqrt(d) from v_rsq_f64 + two Newton steps (quadratic: ~1e-8 -> ~1e-16 -> below an ulp), sqrt(d) = d * that + one
                // correction -- about ten dependent fp64 operations on the critical path of every column instead of the ~60 of the
                // correctly rounded sqrt() followed by a correctly rounded division
                const double hd = 0.5 * d;
                double rinv = __builtin_amdgcn_rsq(d);
                rinv = rinv * fma(-(hd * rinv), rinv, 1.5);
                rinv = rinv * fma(-(hd * rinv), rinv, 1.5);
                double sd = d * rinv;
                sd = fma(0.5 * rinv, fma(-sd, sd, d), sd);
                const double l = !live ? 0.0 : (i == j ? sd : (i > j ? p[jj] * rinv : 0.0));
                p[jj] = l;
                if (i == j) Rinv[j] = live ? rinv : 0.0;
#pragma unroll
                for (int kk = jj + 1; kk < 16; ++kk) p[kk] = fma(-l, readlane_f64(l, c0 + kk), p[kk]);
            }
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) T[i][c0 + jj] = p[jj];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) *(volatile int*)&prog = pb + 1;
        }
        for (int r = 0; r < nb; ++r)
            if (lane <= r) a[(size_t)(k0 + r) * n + k0 + lane] = T[r][lane];
        return;
    }
    // wave 1: inverse, column j = lane: forward substitution L x = e_j in row panels of 16
    const int j = lane;
#pragma unroll 1
    for (int rb = 0; rb < NB / 16; ++rb) {
        while (*(volatile int*)&prog < rb + 1) __builtin_amdgcn_s_sleep(2);
        asm volatile("" ::: "memory");
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
    // the inverse goes onto the diagonal of the full n x n matrix L^-1 (zero elsewhere; completed by invert_levels below)
    double* li = Linv + blockIdx.x * lstride + (size_t)k0 * ldl + k0;
    for (int r = 0; r < nb; ++r)
        if (lane < nb) li[(size_t)r * ldl + lane] = Ti[r][lane];
}

// 64 x 64 operand tiles -> LDS, coalesced along the contiguous source dimension.  All 32 loads of a thread are issued before
// the first LDS write (the rolled load -> store loop made every tile pay 16 dependent memory round trips).
__device__ __forceinline__ void load_tiles(double (*As)[NB + 1], double (*Bs)[NB + 1], const double* __restrict__ am, int lda, int ta,
                                           const double* __restrict__ bm, int ldb, int tb, int i0, int c0, int M, int N, int kk) {
    constexpr int NL = NB * NB / 256;
    const int t = threadIdx.x;
    double ra[NL], rb[NL];
    if (ta) {
#pragma unroll
        for (int j = 0; j < NL; ++j) { const int e = t + 256 * j, k = e / NB, i = e - k * NB; ra[j] = (k < kk && i0 + i < M) ? am[(size_t)k * lda + i0 + i] : 0.0; }
    } else {
#pragma unroll
        for (int j = 0; j < NL; ++j) { const int e = t + 256 * j, i = e / NB, k = e - i * NB; ra[j] = (k < kk && i0 + i < M) ? am[(size_t)(i0 + i) * lda + k] : 0.0; }
    }
    if (tb) {
#pragma unroll
        for (int j = 0; j < NL; ++j) { const int e = t + 256 * j, cc = e / NB, k = e - cc * NB; rb[j] = (k < kk && c0 + cc < N) ? bm[(size_t)(c0 + cc) * ldb + k] : 0.0; }
    } else {
#pragma unroll
        for (int j = 0; j < NL; ++j) { const int e = t + 256 * j, k = e / NB, cc = e - k * NB; rb[j] = (k < kk && c0 + cc < N) ? bm[(size_t)k * ldb + c0 + cc] : 0.0; }
    }
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int e = t + 256 * j, hi = e / NB, lo = e - hi * NB;
        if (ta) As[lo][hi] = ra[j]; else As[hi][lo] = ra[j];      // ta: (k, i) = (hi, lo); else (i, k) = (hi, lo)
        if (tb) Bs[lo][hi] = rb[j]; else Bs[hi][lo] = rb[j];      // tb: (c, k) = (hi, lo); else (k, c) = (hi, lo)
    }
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
    load_tiles(As, Bs, am, lda, ta, bm, ldb, tb, i0, c0, M, N, kk);
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
    load_tiles(As, Bs, am, lda, ta, bm, ldb, tb, i0, c0, M, N, kk);
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

// General fp64 product on 64 x 64 tiles with a K loop:  C[i][c] = sign * sum_{k in range} Aop[i][k] * Bop[k][c]
//   Aop[i][k] = ta ? A[k*lda + i] : A[i*lda + k]        Bop[k][c] = tb ? B[c*ldb + k] : B[k*ldb + c]
// tri restricts the K range of a tile to where a triangular operand is non-zero:
//   1: A lower triangular (k < i0 + 64)   2: A = transposed lower (k >= i0)   3: B lower triangular (k >= c0)
// blockIdx.z = b * npair + p: matrix b (strides bsA/bsB/bsC) and sub-problem p (strides psA/psB/psC); the last sub-problem
// has Mlast rows.
struct GemmArgs {
    double* C; const double* A; const double* B;
    int ldc, lda, ldb, ta, tb, M, Mlast, N, K, Klast, tri, npair;
    double sign;
    size_t bsC, bsA, bsB, psC, psA, psB;
};

// 128 x 64 block tile on the fp64 MFMA (v_mfma_f64_16x16x4_f64: A lane l = A[l % 16][l / 16], B lane l = B[l / 16][l % 16],
// D lane l, register r = D[4 r + l / 16][l % 16], r = 0..3 -- probed on the device, tools/probe_mfma_f64.hip): wave w owns rows [32 w, 32 w + 32) x all 64 columns = 2 x 4 MFMA tiles,
// K in slabs of 32 = 8 MFMA steps, per step 6 ds_read_b64 feed 8 MFMAs.  Each operand keeps the orientation of its SOURCE
// in LDS, so that the coalesced global rows are stored contiguously: [k][row] with rows padded by 16 doubles when the row
// index is the contiguous source dimension (the two k rows a 32-lane read group touches sit 32 banks apart), [row][k] with a
// stride of 34 doubles when k is (16 rows x 2 k land on 64 distinct banks).  Both the stores and the fragment reads are
// conflict-free in both orientations; a single [k][row] image made the transposing stores 16-way conflicted.
// The first version of this kernel was a VALU product, 8 x 4 per thread, 5.3 FMAs per LDS read: 6.5 TFLOP/s, 2.8 ms of a
// gim_dkm match() (round 1).
typedef double f64x4_t __attribute__((ext_vector_type(4)));
constexpr int GBM = 128, GBN = 64, GKT = 32;

constexpr int GSK = GBM + 16, GSKB = GBN + 16, GSR = GKT + 2;   // row strides: [k][row] images (A / B), [row][k] image

template <bool TA, bool TB>
__global__ void __launch_bounds__(256) gemm_f64_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) double As[TA ? GKT * GSK : GBM * GSR];   // TA: [k][i], else [i][k]
    __shared__ __attribute__((aligned(16))) double Bs[TB ? GBN * GSR : GKT * GSKB];  // TB: [c][k], else [k][c]
    const int b = blockIdx.z / g.npair, p = blockIdx.z - b * g.npair;
    const int M = p == g.npair - 1 ? g.Mlast : g.M;
    const int K = p == g.npair - 1 ? g.Klast : g.K;
    const int t = threadIdx.x, i0 = blockIdx.y * GBM, c0 = blockIdx.x * GBN;
    if (i0 >= M || c0 >= g.N) return;
    double* c = g.C + b * g.bsC + p * g.psC;
    const double* am = g.A + b * g.bsA + p * g.psA;
    const double* bm = g.B + b * g.bsB + p * g.psB;
    int kb = 0, ke = K;
    if (g.tri == 1) ke = min(K, i0 + GBM);
    else if (g.tri == 2) kb = i0;
    else if (g.tri == 3) kb = c0;
    const int wave = t >> 6, lane = t & 63, l15 = lane & 15, lq = lane >> 4;
    f64x4_t acc[2][4];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = (f64x4_t){0.0, 0.0, 0.0, 0.0};
    // Software pipeline over the K slabs: the global loads of slab s + 1 are issued (into registers) before the MFMAs of slab s
    // and written to LDS after them -- single-buffered, every slab exposed a full memory round trip (~1.5 us x 72 slabs per
    // n = 2304 product, whatever the FMA rate: the MFMA version of the inner product alone gained 8 %).
    // Loads coalesce along the contiguous source dimension.
    double ra[GBM * GKT / 256], rb[GBN * GKT / 256];
    auto fetch = [&](const int k0) {
        if (TA) {
#pragma unroll
            for (int j = 0; j < GBM * GKT / 256; ++j) {
                const int e = t + 256 * j, k = e / GBM, i = e - k * GBM;
                ra[j] = (k0 + k < ke && i0 + i < M) ? am[(size_t)(k0 + k) * g.lda + i0 + i] : 0.0;
            }
        } else {
#pragma unroll
            for (int j = 0; j < GBM * GKT / 256; ++j) {
                const int e = t + 256 * j, i = e / GKT, k = e - i * GKT;
                ra[j] = (k0 + k < ke && i0 + i < M) ? am[(size_t)(i0 + i) * g.lda + k0 + k] : 0.0;
            }
        }
        if (TB) {
#pragma unroll
            for (int j = 0; j < GBN * GKT / 256; ++j) {
                const int e = t + 256 * j, cc = e / GKT, k = e - cc * GKT;
                rb[j] = (k0 + k < ke && c0 + cc < g.N) ? bm[(size_t)(c0 + cc) * g.ldb + k0 + k] : 0.0;
            }
        } else {
#pragma unroll
            for (int j = 0; j < GBN * GKT / 256; ++j) {
                const int e = t + 256 * j, k = e / GBN, cc = e - k * GBN;
                rb[j] = (k0 + k < ke && c0 + cc < g.N) ? bm[(size_t)(k0 + k) * g.ldb + c0 + cc] : 0.0;
            }
        }
    };
    if (kb < ke) fetch(kb);
    for (int k0 = kb; k0 < ke; k0 += GKT) {
#pragma unroll
        for (int j = 0; j < GBM * GKT / 256; ++j) {
            const int e = t + 256 * j;
            if (TA) { const int k = e / GBM; As[k * GSK + e - k * GBM] = ra[j]; } else { const int i = e / GKT; As[i * GSR + e - i * GKT] = ra[j]; }
        }
#pragma unroll
        for (int j = 0; j < GBN * GKT / 256; ++j) {
            const int e = t + 256 * j;
            if (TB) { const int cc = e / GKT; Bs[cc * GSR + e - cc * GKT] = rb[j]; } else { const int k = e / GBN; Bs[k * GSKB + e - k * GBN] = rb[j]; }
        }
        __syncthreads();
        if (k0 + GKT < ke) fetch(k0 + GKT);
#pragma unroll
        for (int k4 = 0; k4 < GKT; k4 += 4) {
            double av[2], bv[4];
#pragma unroll
            for (int x = 0; x < 2; ++x) av[x] = TA ? As[(k4 + lq) * GSK + wave * 32 + x * 16 + l15] : As[(wave * 32 + x * 16 + l15) * GSR + k4 + lq];
#pragma unroll
            for (int y = 0; y < 4; ++y) bv[y] = TB ? Bs[(y * 16 + l15) * GSR + k4 + lq] : Bs[(k4 + lq) * GSKB + y * 16 + l15];
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[x], bv[y], acc[x][y], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + wave * 32 + x * 16 + r * 4 + lq;
            if (i >= M) continue;
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                const int cc = c0 + y * 16 + l15;
                if (cc < g.N) c[(size_t)i * g.ldc + cc] = g.sign * acc[x][y][r];
            }
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
    // A (L), full L^-1, product scratch: n x n each; right-hand sides twice: n x nrhs
    return (int64_t)(3 * al((size_t)B * n * n * 8) + 2 * al((size_t)B * n * nrhs * 8) + 256);
}

namespace {
void launch_gemm(hipStream_t s, int B, const GemmArgs& g) {
    const int mt = ((g.M > g.Mlast ? g.M : g.Mlast) + GBM - 1) / GBM, nt = (g.N + GBN - 1) / GBN;
    if (g.ta && g.tb) hipLaunchKernelGGL((gemm_f64_kernel<true, true>), dim3(nt, mt, B * g.npair), dim3(256), 0, s, g);
    else if (g.ta) hipLaunchKernelGGL((gemm_f64_kernel<true, false>), dim3(nt, mt, B * g.npair), dim3(256), 0, s, g);
    else if (g.tb) hipLaunchKernelGGL((gemm_f64_kernel<false, true>), dim3(nt, mt, B * g.npair), dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_f64_kernel<false, false>), dim3(nt, mt, B * g.npair), dim3(256), 0, s, g);
}
}  // namespace

namespace {
// Blocked fp64 Cholesky of A [B][n][n] (lower triangle overwritten by L) and X = A^-1 R for the right-hand sides R [B][n][nrhs]
// (overwritten by X).  Lf: zeroed [B][n][n] (receives L^-1), Tb: [B][n][n] scratch, R2: [B][n][nrhs] scratch, info: zeroed ints.
void chol_solve(hipStream_t s, double* A, double* Lf, double* Tb, double* R, double* R2, int* info, int B, int n, int nrhs) {
    const size_t as = (size_t)n * n, fs = (size_t)n * nrhs;
    // ---- A = L L^T (lower triangle of A overwritten by L); the inverses of the diagonal blocks land on the diagonal of Lf ----
    for (int k0 = 0; k0 < n; k0 += NB) {
        const int nb = n - k0 < NB ? n - k0 : NB;
        hipLaunchKernelGGL(potrf_diag_kernel, dim3(B), dim3(128), 0, s, A, n, k0, nb, as, info, Lf, as, n);
        const int m = n - k0 - nb;
        if (m <= 0) break;
        const int tiles = (m + NB - 1) / NB;
        const size_t off = (size_t)(k0 + nb) * n;
        const double* li = Lf + (size_t)k0 * n + k0;
        // panel: P <- P L_kk^-T  = P (Linv)^T : Aop = P rows, Bop[k][c] = Linv[c][k]   (in place: one 64-column tile per row tile)
        hipLaunchKernelGGL(gemm_set_kernel, dim3(1, tiles, B), dim3(256), 0, s, A + off + k0, n, A + off + k0, n, 0, li, n, 1, m, nb, nb, as, as, as);
        // trailing[i][j] -= sum_k P[i][k] P[j][k]
        hipLaunchKernelGGL(gemm_sub_kernel, dim3(tiles, tiles, B), dim3(256), 0, s, A + off + k0 + nb, n, A + off + k0, n, 0,
                           A + off + k0, n, 1, m, m, nb, 1, as, as, as);
    }
    // ---- L^-1 by doubling: inv([[A, 0], [C, D]]) = [[A^-1, 0], [-D^-1 C A^-1, D^-1]].  Every level is two batched GEMMs over
    // all block pairs (6 levels for n = 2304) instead of the 2 x n/64 dependent small launches of blocked substitution. ----
    for (int sz = NB; sz < n; sz *= 2) {
        const int npair = (n + 2 * sz - 1) / (2 * sz);
        const int last_r1 = (npair - 1) * 2 * sz + sz;                  // first row of the last pair's second half
        const int mlast = last_r1 < n ? (n - last_r1 < sz ? n - last_r1 : sz) : 0;
        GemmArgs g{};
        // T = D^-1 C:  D^-1 = Lf[r1:r2, r1:r2] (lower), C = L[r1:r2, r0:r1]
        g.C = Tb; g.ldc = sz; g.psC = (size_t)sz * sz; g.bsC = as;
        g.A = Lf + (size_t)sz * n + sz; g.lda = n; g.ta = 0; g.psA = (size_t)2 * sz * n + 2 * sz; g.bsA = as;
        g.B = A + (size_t)sz * n; g.ldb = n; g.tb = 0; g.psB = (size_t)2 * sz * n + 2 * sz; g.bsB = as;
        g.M = sz; g.Mlast = mlast; g.N = sz; g.K = sz; g.Klast = mlast; g.tri = 1; g.npair = npair; g.sign = 1.0;
        launch_gemm(s, B, g);
        // Lf[r1:r2, r0:r1] = -T A^-1:  A^-1 = Lf[r0:r1, r0:r1] (lower)
        GemmArgs h{};
        h.C = Lf + (size_t)sz * n; h.ldc = n; h.psC = (size_t)2 * sz * n + 2 * sz; h.bsC = as;
        h.A = Tb; h.lda = sz; h.ta = 0; h.psA = (size_t)sz * sz; h.bsA = as;
        h.B = Lf; h.ldb = n; h.tb = 0; h.psB = (size_t)2 * sz * n + 2 * sz; h.bsB = as;
        h.M = sz; h.Mlast = mlast; h.N = sz; h.K = sz; h.Klast = sz; h.tri = 3; h.npair = npair; h.sign = -1.0;
        launch_gemm(s, B, h);
    }
    // ---- Y = L^-1 F, X = L^-T Y: two triangular GEMMs ----
    {
        GemmArgs g{};
        g.C = R2; g.ldc = nrhs; g.bsC = fs; g.A = Lf; g.lda = n; g.ta = 0; g.bsA = as; g.B = R; g.ldb = nrhs; g.tb = 0; g.bsB = fs;
        g.M = n; g.Mlast = n; g.N = nrhs; g.K = n; g.Klast = n; g.tri = 1; g.npair = 1; g.sign = 1.0;
        launch_gemm(s, B, g);
        GemmArgs h = g;
        h.C = R; h.B = R2; h.ta = 1; h.tri = 2;
        launch_gemm(s, B, h);
    }
}
}  // namespace

extern "C" int gim_gp_solve(const float* K, const float* F, float* Xt, void* ws, int B, int n, int ldk, int nrhs, int npad,
                            gim_stream_t stream) {
    GIM_REQUIRE(K && F && Xt && ws && B > 0 && B <= 16 && n > 0 && nrhs > 0 && ldk >= n && npad >= n, "gp_solve: bad args");
    hipStream_t s = (hipStream_t)stream;
    double* A = (double*)ws;
    double* Lf = (double*)((char*)A + al((size_t)B * n * n * 8));     // L^-1, full lower-triangular matrix
    double* Tb = (double*)((char*)Lf + al((size_t)B * n * n * 8));    // products of the inversion levels
    double* R = (double*)((char*)Tb + al((size_t)B * n * n * 8));
    double* R2 = (double*)((char*)R + al((size_t)B * n * nrhs * 8));
    int* info = (int*)((char*)R2 + al((size_t)B * n * nrhs * 8));
    const size_t as = (size_t)n * n, fs = (size_t)n * nrhs;
    if (hipMemsetAsync(info, 0, 64, s) != hipSuccess) return gim_check_launch("gp_solve memset");
    if (hipMemsetAsync(Lf, 0, (size_t)B * as * 8, s) != hipSuccess) return gim_check_launch("gp_solve memset");
    for (int b = 0; b < B; ++b) {
        hipLaunchKernelGGL(to_f64_kernel, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, s, K + (size_t)b * n * ldk, A + b * as, n, n, ldk, n);
        hipLaunchKernelGGL(to_f64_kernel, dim3((unsigned)(((size_t)n * nrhs + 255) / 256)), dim3(256), 0, s, F + (size_t)b * fs, R + b * fs, n, nrhs, nrhs, nrhs);
    }
    chol_solve(s, A, Lf, Tb, R, R2, info, B, n, nrhs);
    hipLaunchKernelGGL(store_xt_kernel, dim3((npad + 63) / 64, (nrhs + 63) / 64, B), dim3(256), 0, s, R, Xt, n, nrhs, npad);
    return gim_check_launch("gp_solve");
}

// ------------------------------------------------------------------------------------------------------------------
// The whole posterior in fp64 from the fp32 feature rows (the parity mode of the two dense matchers):
//     mu = K_xy (K_yy + sigma I)^-1 f,   K[i][j] = exp((x_i . y_j / (|x_i| |y_j| + eps) - 1) / T)      (dkm.py:135-144, 340-370)
// K_yy + sigma I has a condition number of ~2e4: fp32 rounding of the kernel ENTRIES alone (1e-7 relative) moves mu by ~1e-4 of
// its scale (tests/test_gpu_gp_pins.py: engine with fp32 entries 6e-5 ... 2e-4, the reference's fp32 LU inverse 4e-5 ... 8e-5 from
// an all-fp64 evaluation).  Here entries, factorisation and both products are fp64 (fp64 MFMA GEMM above), so this side sits
// at the exact-arithmetic value of the reference's formula (<= 1e-9 of scale) and the remaining difference to a reference
// run is the reference's own rounding.
namespace {
__global__ void norms_f64_kernel(const double* __restrict__ x, double* __restrict__ out, int rows, int d) {   // one wave per row
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    double acc = 0.0;
    for (int c = lane; c < d; c += 64) { const double v = x[(size_t)row * d + c]; acc += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) out[row] = sqrt(acc);
}
// K[i][j] = exp((K[i][j] / (nx[i] ny[j] + eps) - 1) / T) + (i == j ? diag : 0), in place, [B][n][n]
__global__ void cos_finish_f64_kernel(double* __restrict__ k, const double* __restrict__ nx, const double* __restrict__ ny, int n,
                                      double T, double eps, double diag) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (idx >= (size_t)n * n) return;
    const int i = (int)(idx / n), j = (int)(idx - (size_t)i * n);
    double* p = k + (size_t)b * n * n + idx;
    const double c = *p / (nx[(size_t)b * n + i] * ny[(size_t)b * n + j] + eps);
    *p = exp((c - 1.0) / T) + (i == j ? diag : 0.0);
}
__global__ void to_f32_rows_kernel(const double* __restrict__ src, float* __restrict__ dst, int rows, int cols, int ldd) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)rows * cols) return;
    const size_t r = idx / cols, c = idx - r * cols;
    dst[r * ldd + c] = (float)src[idx];
}
}  // namespace

extern "C" int64_t gim_gp_posterior_f64_ws_bytes(int B, int n, int d, int nrhs) {
    // X, Y rows and norms; K_xy; then the solver's A, L^-1, scratch (n x n each) and two right-hand-side blocks
    return (int64_t)(2 * al((size_t)B * n * d * 8) + 2 * al((size_t)B * n * 8) + 4 * al((size_t)B * n * n * 8) + 2 * al((size_t)B * n * nrhs * 8) + 256);
}

extern "C" int gim_gp_posterior_f64(const float* X, const float* Y, const float* F, float* mu, void* ws, int B, int n, int d,
                                    int ldx, int nrhs, int ld_mu, float T, float eps, float sigma, gim_stream_t stream) {
    GIM_REQUIRE(X && Y && F && mu && ws && B > 0 && B <= 16 && n > 0 && d > 0 && nrhs > 0 && ldx >= d && ld_mu >= nrhs, "gp_posterior_f64: bad args");
    hipStream_t s = (hipStream_t)stream;
    char* w = (char*)ws;
    auto take = [&](size_t bytes) { char* p = w; w += al(bytes); return p; };
    double* Xd = (double*)take((size_t)B * n * d * 8);
    double* Yd = (double*)take((size_t)B * n * d * 8);
    double* nx = (double*)take((size_t)B * n * 8);
    double* ny = (double*)take((size_t)B * n * 8);
    double* Kxy = (double*)take((size_t)B * n * n * 8);
    double* A = (double*)take((size_t)B * n * n * 8);
    double* Lf = (double*)take((size_t)B * n * n * 8);
    double* Tb = (double*)take((size_t)B * n * n * 8);
    double* R = (double*)take((size_t)B * n * nrhs * 8);
    double* R2 = (double*)take((size_t)B * n * nrhs * 8);
    int* info = (int*)w;
    const size_t as = (size_t)n * n, fs = (size_t)n * nrhs, xs = (size_t)n * d;
    if (hipMemsetAsync(info, 0, 64, s) != hipSuccess) return gim_check_launch("gp_posterior_f64 memset");
    if (hipMemsetAsync(Lf, 0, (size_t)B * as * 8, s) != hipSuccess) return gim_check_launch("gp_posterior_f64 memset");
    const unsigned gx = (unsigned)((xs + 255) / 256), gf = (unsigned)((fs + 255) / 256);
    for (int b = 0; b < B; ++b) {
        hipLaunchKernelGGL(to_f64_kernel, dim3(gx), dim3(256), 0, s, X + (size_t)b * n * ldx, Xd + b * xs, n, d, ldx, d);
        hipLaunchKernelGGL(to_f64_kernel, dim3(gx), dim3(256), 0, s, Y + (size_t)b * n * ldx, Yd + b * xs, n, d, ldx, d);
        hipLaunchKernelGGL(to_f64_kernel, dim3(gf), dim3(256), 0, s, F, R + b * fs, n, nrhs, nrhs, nrhs);   // the same f for every direction
    }
    hipLaunchKernelGGL(norms_f64_kernel, dim3((unsigned)((B * n + 3) / 4)), dim3(256), 0, s, Xd, nx, B * n, d);
    hipLaunchKernelGGL(norms_f64_kernel, dim3((unsigned)((B * n + 3) / 4)), dim3(256), 0, s, Yd, ny, B * n, d);
    {   // dot products: K_yy <- Y Y^T, K_xy <- X Y^T
        GemmArgs g{};
        g.C = A; g.ldc = n; g.bsC = as; g.A = Yd; g.lda = d; g.ta = 0; g.bsA = xs; g.B = Yd; g.ldb = d; g.tb = 1; g.bsB = xs;
        g.M = n; g.Mlast = n; g.N = n; g.K = d; g.Klast = d; g.tri = 0; g.npair = 1; g.sign = 1.0;
        launch_gemm(s, B, g);
        g.C = Kxy; g.A = Xd;
        launch_gemm(s, B, g);
    }
    const dim3 ge((unsigned)((as + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(cos_finish_f64_kernel, ge, dim3(256), 0, s, A, ny, ny, n, (double)T, (double)eps, (double)sigma);
    hipLaunchKernelGGL(cos_finish_f64_kernel, ge, dim3(256), 0, s, Kxy, nx, ny, n, (double)T, (double)eps, 0.0);
    chol_solve(s, A, Lf, Tb, R, R2, info, B, n, nrhs);     // R <- (K_yy + sigma I)^-1 f
    {   // mu = K_xy R  (into R2), then fp32 rows
        GemmArgs g{};
        g.C = R2; g.ldc = nrhs; g.bsC = fs; g.A = Kxy; g.lda = n; g.ta = 0; g.bsA = as; g.B = R; g.ldb = nrhs; g.tb = 0; g.bsB = fs;
        g.M = n; g.Mlast = n; g.N = nrhs; g.K = n; g.Klast = n; g.tri = 0; g.npair = 1; g.sign = 1.0;
        launch_gemm(s, B, g);
    }
    for (int b = 0; b < B; ++b)
        hipLaunchKernelGGL(to_f32_rows_kernel, dim3(gf), dim3(256), 0, s, R2 + b * fs, mu + (size_t)b * n * ld_mu, n, nrhs, ld_mu);
    return gim_check_launch("gp_posterior_f64");
}
