// Shared MFMA main loop of the implicit-GEMM kernels (conv / linear / coarse-matching similarity).
// See conv_igemm.hip for the design notes (LDS image, swizzle, LDS-DMA staging, transposed MFMA tile).
#pragma once
#include "gim_common.h"

namespace gim {

constexpr int KTB = 128;  // bytes of K per row per LDS stage

struct MainloopArgs {
    const void* x;      // pixel rows (NHWC), dtype T
    const void* w;      // [npad][kpad] rows, dtype T
    const int* ktab;    // [(nkt + 2) * 8]
    unsigned x_bytes;   // buffer bound of x (offsets >= bound read 0)
    unsigned w_bytes;   // buffer bound of w
    int H, W, Ho, Wo, stride, pad, ldx;
    int kpad;
    int M;              // valid output rows
};

template <int BM, int BN>
constexpr int mainloop_smem_bytes() { return 2 * (BM + BN) * KTB; }

// Accumulates the BM x BN tile at (m0, n0) into acc[TN][TM] (transposed fragments: acc[i][j][rg*4+e] is
// output channel n0 + wn*WTN + i*32 + rg*8 + (lane>>5)*4 + e of pixel m0 + wm*WTM + j*32 + (lane&31)).
// On return all waves have passed a barrier and the LDS stage buffers are free for reuse.
template <int BM, int BN, int WM, int WN, bool BF16, bool LDSDMA>
__device__ __forceinline__ void igemm_mainloop(const MainloopArgs& a, char* smem, const int m0, const int n0,
                                               f32x16_t (&acc)[BN / WN / 32][BM / WM / 32]) {
    constexpr int ES = BF16 ? 2 : 4;
    constexpr int A_BYTES = BM * KTB, B_BYTES = BN * KTB, STAGE = A_BYTES + B_BYTES;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    constexpr int PA = BM / 32, PB = BN / 32;
    static_assert(WM * WN == 4, "4 waves");
    const int M = a.M;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // ---- staging role of this thread: row (t>>3) of each 32-row pass, LDS slot (t&7) -----------
    const int srow = t >> 3, sslot = t & 7;
    const int sgrp = sslot ^ ((srow >> 1) & 7);  // K group (16 B) within the slab fetched into that slot

    int iy0[PA], ix0[PA];
    unsigned pix0[PA];
    {
        const int HoWo = a.Ho * a.Wo;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int m = m0 + i * 32 + srow;
            if (m < M) {
                const int b = m / HoWo, r = m - b * HoWo;
                const int ho = r / a.Wo, wo = r - ho * a.Wo;
                iy0[i] = ho * a.stride - a.pad;
                ix0[i] = wo * a.stride - a.pad;
                pix0[i] = (unsigned)b * (unsigned)(a.H * a.W);
            } else {
                iy0[i] = -(1 << 24);
                ix0[i] = 0;
                pix0[i] = 0;
            }
        }
    }
    const unsigned oobx = a.x_bytes;
    const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)oobx, 0x00020000);
    const unsigned wbytes = a.w_bytes;
    const auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)wbytes, 0x00020000);
    const unsigned wrow = (unsigned)(n0 + srow) * (unsigned)a.kpad * ES + sgrp * 16;

    uint4 ra[LDSDMA ? 1 : PA], rb[LDSDMA ? 1 : PB];

    // `e` = ktab entry of this lane's K group for slab kt; it is fetched one slab ahead (below) so that
    // no dependent global load sits in front of the LDS-DMA issue.  The host pads ktab with two slabs
    // of "invalid" entries, so the look-ahead never needs a bounds check.
    auto stage_issue = [&](int buf, int kt, int e) {
        const int c = e & 0xffff, dx = (e >> 16) & 0xff, dy = (e >> 24) & 0xff;
        char* sA = smem + buf * STAGE;
        char* sB = sA + A_BYTES;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int iy = iy0[i] + dy, ix = ix0[i] + dx;
            const bool ok = (dy != 255) && ((unsigned)iy < (unsigned)a.H) && ((unsigned)ix < (unsigned)a.W);
            const unsigned off = ((pix0[i] + (unsigned)(iy * a.W + ix)) * (unsigned)a.ldx + (unsigned)c) * ES;
            const unsigned voff = ok ? off : oobx;
            if constexpr (LDSDMA) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)(sA + (i * 32 + wave * 8) * KTB), 16, voff, 0, 0, 0);
            } else {
                ra[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, 0, 0));
            }
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const unsigned voff = wrow + (unsigned)(i * 32) * (unsigned)a.kpad * ES + (unsigned)kt * KTB;
            if constexpr (LDSDMA) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(sB + (i * 32 + wave * 8) * KTB), 16, voff, 0, 0, 0);
            } else {
                rb[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, voff, 0, 0));
            }
        }
    };
    auto stage_write = [&](int buf) {
        if constexpr (!LDSDMA) {
            char* sA = smem + buf * STAGE;
            char* sB = sA + A_BYTES;
#pragma unroll
            for (int i = 0; i < PA; ++i) *(uint4*)(sA + (i * 32 + srow) * KTB + sslot * 16) = ra[i];
#pragma unroll
            for (int i = 0; i < PB; ++i) *(uint4*)(sB + (i * 32 + srow) * KTB + sslot * 16) = rb[i];
        }
    };

    // ---- compute role --------------------------------------------------------------------------
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int lswz = (l31 >> 1) & 7;
    const int arow0 = (wm * WTM + l31) * KTB, brow0 = (wn * WTN + l31) * KTB;

#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nkt = a.kpad * ES / KTB;
    int e_nxt = a.ktab[8 + sgrp];
    stage_issue(0, 0, a.ktab[sgrp]);
    stage_write(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const int e_n2 = a.ktab[(kt + 2) * 8 + sgrp];
        if (kt + 1 < nkt) stage_issue(cur ^ 1, kt + 1, e_nxt);
        const char* sA = smem + cur * STAGE;
        const char* sB = sA + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int so = ((2 * ks + lh) ^ lswz) << 4;
            if constexpr (BF16) {
                bf16x8_t fa[TM], fb[TN];
#pragma unroll
                for (int j = 0; j < TM; ++j) fa[j] = *(const bf16x8_t*)(sA + arow0 + j * 32 * KTB + so);
#pragma unroll
                for (int i = 0; i < TN; ++i) fb[i] = *(const bf16x8_t*)(sB + brow0 + i * 32 * KTB + so);
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[i], fa[j], acc[i][j], 0, 0, 0);
            } else {
                f32x4_t fa[TM], fb[TN];
#pragma unroll
                for (int j = 0; j < TM; ++j) fa[j] = *(const f32x4_t*)(sA + arow0 + j * 32 * KTB + so);
#pragma unroll
                for (int i = 0; i < TN; ++i) fb[i] = *(const f32x4_t*)(sB + brow0 + i * 32 * KTB + so);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < TN; ++i)
#pragma unroll
                        for (int j = 0; j < TM; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[i][q], fa[j][q], acc[i][j], 0, 0, 0);
            }
        }
        if (kt + 1 < nkt) stage_write(cur ^ 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        e_nxt = e_n2;
    }

}

}  // namespace gim
