// Shared MFMA building blocks of the implicit-GEMM kernels (conv / linear / coarse-matching similarity).
// See conv_igemm.hip for the design notes (LDS image, swizzle, LDS-DMA staging, transposed MFMA tile).
#pragma once
#include <type_traits>
#include "gim_common.h"

namespace gim {

constexpr int KTB = 128;  // bytes of K per row per LDS stage ("slab")

struct MainloopArgs {
    const void* x;      // pixel rows (NHWC), dtype T
    const void* w;      // [npad][kpad] rows, dtype T
    const int* ktab;    // [(nkt + 2) * 8]: K group -> c | dx<<16 | dy<<24 (dy = 255: padding)
    unsigned x_bytes;   // buffer bound of x (offsets >= bound read 0)
    unsigned w_bytes;   // buffer bound of w
    int H, W, Ho, Wo, stride, pad, ldx;
    int kpad;
    int ldw;            // row stride of w in elements (kpad for packed weights; larger for a run-time row operand)
    int M;              // valid output rows
};

template <int BM, int BN>
constexpr int mainloop_smem_bytes() { return 2 * (BM + BN) * KTB; }

// Per-thread state + steps of one BM x BN tile pipeline.  256 threads = 4 waves (WM x WN).
//   staging role : thread t fetches LDS slot (t & 7) of row (t >> 3) of every 32-row pass; the K group it
//                  fetches into that slot is slot ^ ((row >> 1) & 7)  (source-side swizzle, LDS-DMA is
//                  lane-linear);
//   compute role : wave (wm, wn) owns a (BM/WM) x (BN/WN) sub-tile as TM x TN 32x32 MFMA fragments,
//                  computed transposed: acc[i][j][rg*4+e] = channel n0 + wn*WTN + i*32 + rg*8 + (lane>>5)*4 + e
//                  of pixel m0 + wm*WTM + j*32 + (lane & 31).
template <int BM, int BN, int WM, int WN, bool BF16, bool LDSDMA>
struct Igemm {
    static constexpr int ES = BF16 ? 2 : 4;
    static constexpr int A_BYTES = BM * KTB, B_BYTES = BN * KTB, STAGE = A_BYTES + B_BYTES;
    static constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    static constexpr int NW = WM * WN;            // waves per workgroup (4 or 8)
    static constexpr int RPP = 8 * NW;            // rows staged per pass (one 16-byte slot per thread)
    static constexpr int PA = BM / RPP, PB = BN / RPP;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the staging pass");
    typedef f32x16_t Acc[TN][TM];

    int iy0[PA], ix0[PA];
    unsigned rowoff[PA];  // byte offset of pixel (b, iy0, ix0), channel 0 (mod 2^32; only used when valid)
    unsigned wrow;
    uint4 ra[LDSDMA ? 1 : PA], rb[LDSDMA ? 1 : PB];

    // pixel coordinates of this thread's staging rows for the tile at (m0, n0).  `t`: the staging thread this lane stands in for
    // (its own index in the kernels whose waves stage and compute; a loader wave of igemm_lw_kernel plays two of them)
    __device__ __forceinline__ void decode(const MainloopArgs& a, int m0, int n0) { decode(a, m0, n0, (int)threadIdx.x); }
    __device__ __forceinline__ void decode(const MainloopArgs& a, int m0, int n0, const int t) {
        const int srow = t >> 3, sslot = t & 7;
        const int sgrp = sslot ^ ((srow >> 1) & 7);
        const bool flat = (a.Ho == 1) && (a.H == 1) && (a.stride == 1) && (a.pad == 0);  // pixel index == row index
        const int HoWo = a.Ho * a.Wo;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int m = m0 + i * RPP + srow;
            int y = 0, x = m;
            unsigned pix = 0;
            if (!flat) {
                const int b = m / HoWo, r = m - b * HoWo;
                const int ho = r / a.Wo, wo = r - ho * a.Wo;
                y = ho * a.stride - a.pad;
                x = wo * a.stride - a.pad;
                pix = (unsigned)b * (unsigned)(a.H * a.W);
            }
            iy0[i] = m < a.M ? y : -(1 << 24);
            ix0[i] = x;
            rowoff[i] = (pix + (unsigned)(y * a.W + x)) * (unsigned)a.ldx * ES;
        }
        wrow = (unsigned)(n0 + srow) * (unsigned)a.ldw * ES + sgrp * 16;
    }

    static __device__ __forceinline__ int ktab_index(int kt) { return ktab_index(kt, (int)threadIdx.x); }
    static __device__ __forceinline__ int ktab_index(int kt, const int t) {
        return kt * 8 + ((t & 7) ^ (((t >> 3) >> 1) & 7));
    }

    // issue the loads of slab kt into LDS stage `buf`.  `e` = ktab[ktab_index(kt)], fetched by the caller
    // one slab ahead so that no dependent global load sits in front of the DMA issue.
    __device__ __forceinline__ void stage_issue(const MainloopArgs& a, char* smem, int buf, int kt, int e) {
        stage_issue(a, smem, buf, kt, e, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
    }
    // (`wave`: the staging wave whose 8-row groups this call fills -- wave-uniform)
    __device__ __forceinline__ void stage_issue(const MainloopArgs& a, char* smem, int buf, int kt, int e, const int wave) {
        // `buf` selects the stage at smem + buf * STAGE (2-stage double buffer or a deeper ring)
        const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
        const auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)a.w_bytes, 0x00020000);
        const int c = e & 0xffff, dx = (e >> 16) & 0xff;
        int dy = (e >> 24) & 0xff;
        dy = dy == 255 ? (1 << 28) : dy;  // K padding group: pushes iy out of range for every row
        // tap offset is common to all rows of this thread
        const unsigned tapoff = ((unsigned)(dy * a.W + dx) * (unsigned)a.ldx + (unsigned)c) * ES;
        char* sA = smem + buf * STAGE;
        char* sB = sA + A_BYTES;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const bool ok = ((unsigned)(iy0[i] + dy) < (unsigned)a.H) & ((unsigned)(ix0[i] + dx) < (unsigned)a.W);
            const unsigned voff = ok ? rowoff[i] + tapoff : a.x_bytes;
            if constexpr (LDSDMA) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)(sA + (i * RPP + wave * 8) * KTB), 16, voff, 0, 0, 0);
            } else {
                ra[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, 0, 0));
            }
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const unsigned voff = wrow + (unsigned)(i * RPP) * (unsigned)a.ldw * ES + (unsigned)kt * KTB;
            if constexpr (LDSDMA) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(sB + (i * RPP + wave * 8) * KTB), 16, voff, 0, 0, 0);
            } else {
                rb[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, voff, 0, 0));
            }
        }
    }

    // ---- the same staging, one DMA instruction ("piece") at a time -------------------------------------------------
    // (the ping-pong kernel spreads the pieces over its phases)
    static constexpr int NPIECE = PA + PB;
    struct Tap { unsigned tapoff; int dy, dx; };
    static __device__ __forceinline__ Tap tap_decode(const MainloopArgs& a, int e) {
        Tap t;
        const int c = e & 0xffff;
        t.dx = (e >> 16) & 0xff;
        int dy = (e >> 24) & 0xff;
        t.dy = dy == 255 ? (1 << 28) : dy;
        t.tapoff = ((unsigned)(t.dy * a.W + t.dx) * (unsigned)a.ldx + (unsigned)c) * ES;
        return t;
    }
    __device__ __forceinline__ void issue_piece(const MainloopArgs& a, char* smem, int buf, int kt, const Tap& t, int p) const {
        static_assert(LDSDMA, "piece-wise staging is LDS-DMA only");
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        char* sA = smem + buf * STAGE;
        if (p < PA) {
            const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
            const bool ok = ((unsigned)(iy0[p] + t.dy) < (unsigned)a.H) & ((unsigned)(ix0[p] + t.dx) < (unsigned)a.W);
            const unsigned voff = ok ? rowoff[p] + t.tapoff : a.x_bytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)(sA + (p * RPP + wave * 8) * KTB), 16, voff, 0, 0, 0);
        } else {
            const int i = p - PA;
            const auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)a.w_bytes, 0x00020000);
            const unsigned voff = wrow + (unsigned)(i * RPP) * (unsigned)a.ldw * ES + (unsigned)kt * KTB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(sA + A_BYTES + (i * RPP + wave * 8) * KTB), 16, voff, 0, 0, 0);
        }
    }
    // register-staging path only: write the fetched slab into LDS stage `buf`
    __device__ __forceinline__ void stage_write(char* smem, int buf) {
        if constexpr (!LDSDMA) {
            const int t = threadIdx.x, srow = t >> 3, sslot = t & 7;
            char* sA = smem + buf * STAGE;
            char* sB = sA + A_BYTES;
#pragma unroll
            for (int i = 0; i < PA; ++i) *(uint4*)(sA + (i * RPP + srow) * KTB + sslot * 16) = ra[i];
#pragma unroll
            for (int i = 0; i < PB; ++i) *(uint4*)(sB + (i * RPP + srow) * KTB + sslot * 16) = rb[i];
        }
    }

    // wave -> (wm, wn).  With 8 waves in a 4 x 2 grid, waves w and w + 4 share a SIMD: give them different wn so that
    // every SIMD carries one wave of each column half (the halves can have different numbers of live N fragments)
    static __device__ __forceinline__ void wave_mn(int wave, int& wm, int& wn) {
        if constexpr (WM * WN == 8 && WN == 2) { wm = wave >> 1; wn = (wave ^ (wave >> 2)) & 1; }
        else { wm = wave / WN; wn = wave - wm * WN; }
    }

    // MFMA operand fragments of one 16-byte K step (ks = 0..3 of a slab) and the MFMAs on them.  LIVE <= TN: only the
    // first LIVE channel fragments of the wave are touched (the others hold padding channels >= N: N = 196 in a 256-wide
    // tile leaves the last 32-channel fragment empty).  LIVE is a template argument on purpose: a run-time branch inside
    // the K loop body stops the compiler from overlapping LDS reads and MFMAs (measured: +4 % on the whole forward).
    typedef typename std::conditional<BF16, bf16x8_t, f32x4_t>::type Frag;
    template <int LIVE> struct Frags { Frag a[TM], b[LIVE]; };
    struct FragAddr { const char *sA, *sB; int lh, lswz; };
    static __device__ __forceinline__ FragAddr frag_addr(const char* smem, int buf) {
        const int lane = threadIdx.x & 63;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int l31 = lane & 31;
        int wm, wn;
        wave_mn(wave, wm, wn);
        FragAddr f;
        f.lh = lane >> 5;
        f.lswz = (l31 >> 1) & 7;
        f.sA = smem + buf * STAGE + (wm * WTM + l31) * KTB;
        f.sB = smem + buf * STAGE + A_BYTES + (wn * WTN + l31) * KTB;
        return f;
    }
    template <int LIVE>
    static __device__ __forceinline__ void load_frags(const FragAddr& f, int ks, Frags<LIVE>& r) {
        const int so = ((2 * ks + f.lh) ^ f.lswz) << 4;
#pragma unroll
        for (int j = 0; j < TM; ++j) r.a[j] = *(const Frag*)(f.sA + j * 32 * KTB + so);
#pragma unroll
        for (int i = 0; i < LIVE; ++i) r.b[i] = *(const Frag*)(f.sB + i * 32 * KTB + so);
    }
    template <int LIVE>
    static __device__ __forceinline__ void mma(Acc& acc, const Frags<LIVE>& r) {
        if constexpr (BF16) {
#pragma unroll
            for (int i = 0; i < LIVE; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = mfma_h16_32x32x16(r.b[i], r.a[j], acc[i][j]);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < LIVE; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(r.b[i][q], r.a[j][q], acc[i][j], 0, 0, 0);
        }
    }

    // acc += slab in LDS stage `buf` (the compiler interleaves the reads of one K step with the MFMAs of the previous one)
    template <int LIVE = TN>
    static __device__ __forceinline__ void compute(const char* smem, int buf, Acc& acc) {
        const FragAddr f = frag_addr(smem, buf);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            Frags<LIVE> r;
            load_frags<LIVE>(f, ks, r);
            mma<LIVE>(acc, r);
        }
    }

    // fp32 operands on the 16-bit matrix pipe (round 6; the parity-grade mode at 16-bit MFMA rates): every fp32 value is split in
    // registers into an IEEE-fp16 pair hi = rn16(x), lo = rn16(x - hi) and the slab's products are evaluated as
    //     x w ~= hi_x hi_w + hi_x lo_w + lo_x hi_w          (lo_x lo_w <= 2^-22 |x w| is dropped)
    // i.e. three v_mfma_f32_32x32x16_f16 per 16 K instead of eight v_mfma_f32_32x32x2_f32, fp32 accumulation either way.  Two 16-byte K
    // steps of the fp32 slab (4 + 4 values per lane) make one 16-bit operand fragment; the pairing is the same on both operands, so the
    // MFMA's K positions line up.  `wscale` (a power of two) multiplies the weight operand before the split -- it lifts the weights' low
    // halves out of the fp16 subnormals; the caller divides the accumulators by it (exact) -- and the bias enters pre-multiplied.
    template <int LIVE = TN>
    static __device__ __forceinline__ void compute_split16(const char* smem, int buf, Acc& acc, const float wscale) {
        static_assert(!BF16, "split products are the fp32 operands' path");
        const FragAddr f = frag_addr(smem, buf);
        auto split = [](const f32x4_t& v0, const f32x4_t& v1, const float sc, bf16x8_t& hi, bf16x8_t& lo) __attribute__((always_inline)) {
            unsigned h[4], l[4];
            float x[8] = {v0[0] * sc, v0[1] * sc, v0[2] * sc, v0[3] * sc, v1[0] * sc, v1[1] * sc, v1[2] * sc, v1[3] * sc};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = cvt_pk_f16(x[2 * e], x[2 * e + 1]);
                l[e] = cvt_pk_f16(x[2 * e] - f16_lo(h[e]), x[2 * e + 1] - f16_hi(h[e]));
            }
            hi = __builtin_bit_cast(bf16x8_t, make_uint4(h[0], h[1], h[2], h[3]));
            lo = __builtin_bit_cast(bf16x8_t, make_uint4(l[0], l[1], l[2], l[3]));
        };
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            // (fragments are loaded where they are split: all of a K step's fp32 fragments at once are 48 registers on the 64 x 128 wave tile)
            const int so0 = ((2 * (2 * kp) + f.lh) ^ f.lswz) << 4, so1 = ((2 * (2 * kp + 1) + f.lh) ^ f.lswz) << 4;
            bf16x8_t ah[TM], al[TM];
#pragma unroll
            for (int j = 0; j < TM; ++j)
                split(*(const f32x4_t*)(f.sA + j * 32 * KTB + so0), *(const f32x4_t*)(f.sA + j * 32 * KTB + so1), 1.f, ah[j], al[j]);
#pragma unroll
            for (int i = 0; i < LIVE; ++i) {
                bf16x8_t bh, bl;
                split(*(const f32x4_t*)(f.sB + i * 32 * KTB + so0), *(const f32x4_t*)(f.sB + i * 32 * KTB + so1), wscale, bh, bl);
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    f32x16_t c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gim_f16x8_t, bl), __builtin_bit_cast(gim_f16x8_t, ah[j]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gim_f16x8_t, bh), __builtin_bit_cast(gim_f16x8_t, al[j]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gim_f16x8_t, bh), __builtin_bit_cast(gim_f16x8_t, ah[j]), c, 0, 0, 0);
                    acc[i][j] = c;
                }
            }
        }
    }

    static __device__ __forceinline__ void zero(Acc& acc) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
};

// One whole tile, double-buffered over K (non-persistent use: coarse matching, register-staging fallback).
// On return all waves have passed a barrier and the LDS stage buffers are free for reuse.
template <int BM, int BN, int WM, int WN, bool BF16, bool LDSDMA>
__device__ __forceinline__ void igemm_mainloop(const MainloopArgs& a, char* smem, const int m0, const int n0,
                                               f32x16_t (&acc)[BN / WN / 32][BM / WM / 32]) {
    typedef Igemm<BM, BN, WM, WN, BF16, LDSDMA> G;
    G g;
    g.decode(a, m0, n0);
    G::zero(acc);
    const int nkt = a.kpad * G::ES / KTB;
    int e_nxt = a.ktab[G::ktab_index(1)];
    g.stage_issue(a, smem, 0, 0, a.ktab[G::ktab_index(0)]);
    g.stage_write(smem, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const int e_n2 = a.ktab[G::ktab_index(kt + 2)];
        if (kt + 1 < nkt) g.stage_issue(a, smem, cur ^ 1, kt + 1, e_nxt);
        G::compute(smem, cur, acc);
        if (kt + 1 < nkt) g.stage_write(smem, cur ^ 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        e_nxt = e_n2;
    }
}

}  // namespace gim
