// Back-to-back fusion of a ResNet Bottleneck's tail with the next block's head (planes P = 64: layer1 of the gim_loftr backbone,
// networks/loftr/backbone/resnet.py:109-126) for gfx950:
//
//     t2  = relu(bn2(conv2_3x3(t1)))                  64 -> 64        (resnet.py:113-115)
//     x'  = relu(bn3(conv3_1x1(t2)) + identity)       64 -> 256       (resnet.py:117-124)
//     t1' = relu(bn1'(conv1'_1x1(x')))               256 -> 64        (the NEXT block's resnet.py:109-111; optional)
//
// ONE kernel instead of three implicit-GEMM launches: t2 never exists in memory and x' is not read back for conv1'.
// Per block boundary at 16 x 240 x 320: 2.5 GB of HBM traffic -> 1.6 GB (read t1 + identity, write x' + t1').
//
// The chain stays in REGISTERS.  A workgroup (8 waves) owns an 8 x 32 pixel tile, wave w the 32 pixels of output row w -- for
// all three products.  Every MFMA is computed "transposed" (weights = A operand, pixels = B operand), so a lane ends up with 4
// consecutive channels of ONE pixel per accumulator quad; that is already the shape of the next product's B operand (pixel =
// lane, 8 contraction values per lane) up to the ORDER of the contraction index -- and the order is free: the next layer's
// weights are packed with their K axis permuted to the accumulator layout (packing.py::pack_bneck).  relu / bias / residual are
// applied to the accumulators, two v_cvt_pk per quad make the bf16 operand, and the next v_mfma consumes it.  No LDS round trip,
// no cross-lane movement between the three convolutions.
//
// LDS holds only what is shared between waves: the 10 x 34 pixel halo tile of t1 (LDS-DMA, zero padding from the buffer
// descriptor) and the weights (conv2 72 KiB, conv3 32 KiB, conv1' 32 KiB -- the last one replaces conv2's while conv3 runs),
// plus a 4 KiB per-wave transposition patch so that the identity loads and all stores are 16 bytes per lane on full rows.
#include "gim_common.h"

namespace {

constexpr int P = 64, C4 = 256;
constexpr int TH = 8, TW = 32;                 // output tile (rows x columns); wave w <-> row w
constexpr int HR = TH + 2, HC = TW + 2;        // halo tile of t1
constexpr int HPIX = HR * HC;                  // 340
constexpr int T1ROWS = 344;                    // 43 LDS-DMA pieces of 8 rows
constexpr int OFF_T1 = 0;                                   // [344][128 B]
constexpr int OFF_W2 = T1ROWS * 128;                        // [64][1152 B]            (conv2 phase)
constexpr int OFF_W1 = OFF_W2;                              // [N1 <= 128][512 B]      (after conv2: replaces W2)
constexpr int OFF_SCR = OFF_T1;                             // 8 waves x 4 KiB transposition patches (after conv2: replace T1)
constexpr int OFF_W3 = OFF_W2 + 64 * 1152;                  // [256][128 B]
constexpr int OFF_WDS = OFF_W1 + 64 * 512;                  // [256][128 B] downsample weights behind W1n (DS variant, N1 = 64: after conv2)
constexpr int SMEM = OFF_W3 + 256 * 128;
static_assert(8 * 4096 <= T1ROWS * 128 && 128 * 512 <= 64 * 1152 && 64 * 512 + 256 * 128 <= 64 * 1152 && SMEM <= 160 * 1024, "LDS map");

struct Args {
    const unsigned short* t1;    // [B,H,W,64] bf16
    const unsigned short* res;   // [B,H,W,256] bf16 identity / downsample branch -- DS variant: [B,H,W,64], the block's INPUT
    const unsigned short* wds;   // DS variant: [256][64] bf16 downsample conv (BN folded), K in channel order; its bias is folded into b3
    unsigned short* xo;          // [B,H,W,256] bf16
    unsigned short* t1n;         // [B,H,W,N1] bf16 or NULL
    const unsigned short* w2;    // [64][576] bf16, K = (ky, kx, c)
    const unsigned short* w3;    // [256][64] bf16, K permuted to the accumulator layout
    const unsigned short* w1n;   // [N1][256] bf16, K permuted (N1 = 64: the next block of the same layer, 128: the next layer's first conv1)
    const float* b2;             // [64]
    const float* b3;             // [256]
    const float* b1n;            // [N1]
    int B, H, W;
    unsigned t1_bytes, w2_bytes, w3_bytes, w1n_bytes;
    int* health;                 // fp16 range guard word (gim_common.h) or NULL
};
// DS (first block of the layer, resnet.py:120-124: identity = bn(conv1x1(x))): the 64 -> 256 downsample convolution is conv3 with its K
// axis extended -- x' = relu([W3 | Wds] [t2 ; x] + b3 + bds) -- the lane's 32 pixels of x arrive as four MFMA operand fragments straight
// from global memory (128 B per pixel instead of the identity's 512 B), Wds takes the LDS space conv2's weights leave behind, and the
// downsample launch (0.17 ms at 640x480 batch 8, HBM-bound) with the 629 MB tensor it wrote disappears.

typedef __attribute__((address_space(3))) void lds_t;
GIM_TT_DECL(bneck64)

// 8 rows x 128 B per wave instruction, lane i -> row base + (i >> 3), LDS slot i & 7 <- source slot (i & 7) ^ key(row)
__device__ __forceinline__ int swz_key(int row) { return (row >> 1) & 7; }   // two 128-byte rows share a 256-byte bank row

template <int N1, bool DS = false>   // N1 = 0: no trailing conv1
__global__ void __launch_bounds__(512, 2) bneck64_kernel(const Args a) {
    static_assert(!DS || N1 == 64, "downsample variant: first block of layer 1");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles_x = a.W / TW, tiles_y = a.H / TH;
    const int tile = blockIdx.x;
    const int b = tile / (tiles_x * tiles_y), tr = tile - b * tiles_x * tiles_y;
    const int y0 = (tr / tiles_x) * TH, x0 = (tr % tiles_x) * TW;
    GIM_TT(bneck64, w, 0);

    // ---- LDS-DMA: halo tile of t1 (zero outside the image), conv2 / conv3 weights -----------------------------------
    {
        const auto r1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.t1, 0, (int)a.t1_bytes, 0x00020000);
        const auto rw2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2, 0, (int)a.w2_bytes, 0x00020000);
        const auto rw3 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w3, 0, (int)a.w3_bytes, 0x00020000);
        const int sub = lane >> 3, slot = lane & 7;
        for (int pc = w; pc < T1ROWS / 8; pc += 8) {               // 43 pieces of 8 halo pixels
            const int h = pc * 8 + sub;
            const int hy = h / HC, hx = h - hy * HC;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool ok = h < HPIX && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            const unsigned voff = ok ? (unsigned)(((b * a.H + y) * a.W + x) * 128 + ((slot ^ swz_key(h)) << 4)) : a.t1_bytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_t*)(smem + OFF_T1 + pc * 1024), 16, voff, 0, 0, 0);
        }
        // W2: row n = 1152 B = 72 slots; piece = 64 lanes x 16 B: linear over (n, slot); the XOR acts on the slot's low 3 bits
        for (int pc = w; pc < 64 * 72 / 64; pc += 8) {
            const int idx = pc * 64 + lane, n = idx / 72, s = idx - n * 72;
            const unsigned voff = (unsigned)(n * 1152 + (((s & ~7) | ((s & 7) ^ swz_key(n))) << 4));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw2, (lds_t*)(smem + OFF_W2 + pc * 1024), 16, voff, 0, 0, 0);
        }
        for (int pc = w; pc < 256 * 8 / 64; pc += 8) {              // W3: 256 rows x 8 slots
            const int n = pc * 8 + sub;
            const unsigned voff = (unsigned)(n * 128 + ((slot ^ swz_key(n)) << 4));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw3, (lds_t*)(smem + OFF_W3 + pc * 1024), 16, voff, 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // T1 and W2 have landed; this wave's 4 W3 pieces (issued last) may still fly
    __builtin_amdgcn_s_barrier();
    GIM_TT(bneck64, w, 1);

    // ---- conv2: D[m = out channel][n = pixel of row w] over 9 taps x 4 k16 steps --------------------------------------
    f32x16_t c2[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const float4 bb = *(const float4*)(a.b2 + 32 * f + 8 * rg + 4 * lh);
            c2[f][rg * 4] = bb.x; c2[f][rg * 4 + 1] = bb.y; c2[f][rg * 4 + 2] = bb.z; c2[f][rg * 4 + 3] = bb.w;
        }
    }
    {
        const char* wrow0 = smem + OFF_W2 + l31 * 1152;
        const char* wrow1 = wrow0 + 32 * 1152;
        const int wk = swz_key(l31);                       // rows l31 and l31 + 32 share the key
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
            const int h = (w + dy) * HC + l31 + dx;
            const char* prow = smem + OFF_T1 + h * 128;
            const int pk = swz_key(h);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8_t px = *(const bf16x8_t*)(prow + (((2 * ks + lh) ^ pk) << 4));
                const int ws = ((tap * 8) | ((2 * ks + lh) ^ wk)) << 4;
                const bf16x8_t w0 = *(const bf16x8_t*)(wrow0 + ws), w1 = *(const bf16x8_t*)(wrow1 + ws);
                c2[0] = mfma_h16_32x32x16(w0, px, c2[0]);
                c2[1] = mfma_h16_32x32x16(w1, px, c2[1]);
            }
        }
    }
    GIM_TT(bneck64, w, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // W3
    __syncthreads();   // every wave is done with W2 and T1 (conv1' weights / the patches take their place); W3 is visible
    // identity rows of the first 64-channel pass (8 lanes x 16 B per pixel, 8 pixels per instruction): requested HERE, a whole conv3
    // ahead of their use (they used to be requested right in front of it: one exposed HBM round trip per tile, and with one
    // workgroup per CU nothing else runs meanwhile)
    const size_t prow0 = ((size_t)(b * a.H + y0 + w) * a.W + x0);   // first pixel of this wave's row
    const unsigned short* rp = a.res + (prow0 + (lane >> 3)) * C4 + (lane & 7) * 8;
    uint4 i0 = make_uint4(0u, 0u, 0u, 0u), i1 = i0, i2 = i0, i3 = i0;
    bf16x8_t xs[4];   // DS: the block's input at this lane's pixel, k16 step s: channels 16 s + 8 lh .. + 7
    if constexpr (DS) {
#pragma unroll
        for (int s = 0; s < 4; ++s) xs[s] = *(const bf16x8_t*)(a.res + (prow0 + l31) * P + 16 * s + 8 * lh);
        const gim_u32x4_t rwd = gim_make_rsrc(a.wds, 256 * 128);
        const unsigned wd_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_t*)(smem + OFF_WDS));
#pragma unroll
        for (int k = 0; k < 4; ++k) {                                // Wds: 256 rows x 8 slots, as W3
            const int pc = w + 8 * k, n = pc * 8 + (lane >> 3);
            gim_dma16(rwd, wd_addr + (unsigned)(pc * 1024), (unsigned)(n * 128 + (((lane & 7) ^ swz_key(n)) << 4)));
        }
    } else {
        i0 = *(const uint4*)(rp); i1 = *(const uint4*)(rp + 8 * C4); i2 = *(const uint4*)(rp + 16 * C4); i3 = *(const uint4*)(rp + 24 * C4);
    }
    if constexpr (N1 > 0) {
        // W1n by LDS-DMA through inline asm (gim_dma16): a DMA the compiler can see makes the next LDS access -- conv3's first weight
        // read -- wait vmcnt(0), i.e. for these 32 / 64 KiB and for the identity rows above; the hand-placed wait sits in front of conv1'
        const gim_u32x4_t rw1 = gim_make_rsrc(a.w1n, a.w1n_bytes);
        const unsigned w1_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_t*)(smem + OFF_W1));
#pragma unroll
        for (int k = 0; k < N1 * 32 / 64 / 8; ++k) {                  // W1n: N1 rows x 32 slots (512 B), XOR on the low 4 slot bits
            const int pc = w + 8 * k;
            const int idx = pc * 64 + lane, n = idx >> 5, s = idx & 31;
            const unsigned voff = (unsigned)(n * 512 + (((s & 16) | ((s & 15) ^ (n & 15))) << 4));
            gim_dma16(rw1, w1_addr + (unsigned)(pc * 1024), voff);
        }
    }
    GIM_TT(bneck64, w, 3);
    // ---- relu -> bf16 operand (contraction order = accumulator order) -> conv3 -----------------------------------------
    bf16x8_t t2[4];   // k16 step s = 2f + t: channels 32f + 16t + 8(p >> 2) + 4lh + (p & 3)
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            unsigned u[4];
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
                const int rg = 2 * t + hq;
                u[2 * hq] = cvt_pk_h16(fmaxf(c2[f][rg * 4], 0.f), fmaxf(c2[f][rg * 4 + 1], 0.f));
                u[2 * hq + 1] = cvt_pk_h16(fmaxf(c2[f][rg * 4 + 2], 0.f), fmaxf(c2[f][rg * 4 + 3], 0.f));
            }
            t2[2 * f + t] = __builtin_bit_cast(bf16x8_t, make_uint4(u[0], u[1], u[2], u[3]));
        }
    f32x16_t c3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const float4 bb = *(const float4*)(a.b3 + 32 * j + 8 * rg + 4 * lh);
            c3[j][rg * 4] = bb.x; c3[j][rg * 4 + 1] = bb.y; c3[j][rg * 4 + 2] = bb.z; c3[j][rg * 4 + 3] = bb.w;
        }
    }
    {
        const int wk = swz_key(l31);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bf16x8_t wv = *(const bf16x8_t*)(smem + OFF_W3 + (32 * j + l31) * 128 + (((2 * s + lh) ^ wk) << 4));
                c3[j] = mfma_h16_32x32x16(wv, t2[s], c3[j]);
            }
        if constexpr (DS) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // Wds (and W1n) pieces of this wave, the xs fragments
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bf16x8_t wv = *(const bf16x8_t*)(smem + OFF_WDS + (32 * j + l31) * 128 + (((2 * s + lh) ^ wk) << 4));
                    c3[j] = mfma_h16_32x32x16(wv, xs[s], c3[j]);
                }
        }
    }
    GIM_TT(bneck64, w, 4);
    // ---- + identity, relu; x' out; bf16 operand of conv1' -- in four 64-channel passes through this wave's LDS patch ----
    char* patch = smem + OFF_SCR + w * 4096;          // [32 px][128 B], 16-byte slots XOR (px & 7)
    bf16x8_t xq[16];  // conv1' operand: step s = 2j + t
    // identity rows are fetched one pass ahead.  Four named registers, not an array: behind the "memory"-clobbering waits below an
    // array was kept in scratch (80 B / lane, a vmcnt(0) around every access: the kernel ran 1.7x slower)
    const int psl = (lane & 7), ppx = lane >> 3;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if constexpr (DS) {   // the identity branch is already inside c3
#pragma unroll
            for (int ff = 0; ff < 2; ++ff)
#pragma unroll
                for (int r = 0; r < 16; ++r) c3[2 * q + ff][r] = fmaxf(c3[2 * q + ff][r], 0.f);
        } else {
        *(uint4*)(patch + (ppx) * 128 + ((psl ^ (ppx & 7)) << 4)) = i0;
        *(uint4*)(patch + (ppx + 8) * 128 + ((psl ^ (ppx & 7)) << 4)) = i1;     // (px + 8k) & 7 == px & 7
        *(uint4*)(patch + (ppx + 16) * 128 + ((psl ^ (ppx & 7)) << 4)) = i2;
        *(uint4*)(patch + (ppx + 24) * 128 + ((psl ^ (ppx & 7)) << 4)) = i3;
        if (q < 3) {
            i0 = *(const uint4*)(rp + 64 * (q + 1)); i1 = *(const uint4*)(rp + 8 * C4 + 64 * (q + 1));
            i2 = *(const uint4*)(rp + 16 * C4 + 64 * (q + 1)); i3 = *(const uint4*)(rp + 24 * C4 + 64 * (q + 1));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ff = 0; ff < 2; ++ff) {
            const int j = 2 * q + ff;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const uint2 r = *(const uint2*)(patch + l31 * 128 + (((4 * ff + rg) ^ (l31 & 7)) << 4) + lh * 8);
                c3[j][rg * 4] = fmaxf(c3[j][rg * 4] + h16_lo(r.x), 0.f);
                c3[j][rg * 4 + 1] = fmaxf(c3[j][rg * 4 + 1] + h16_hi(r.x), 0.f);
                c3[j][rg * 4 + 2] = fmaxf(c3[j][rg * 4 + 2] + h16_lo(r.y), 0.f);
                c3[j][rg * 4 + 3] = fmaxf(c3[j][rg * 4 + 3] + h16_hi(r.y), 0.f);
            }
        }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ff = 0; ff < 2; ++ff) {
            const int j = 2 * q + ff;
            unsigned u[8];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                u[2 * rg] = cvt_pk_h16(c3[j][rg * 4], c3[j][rg * 4 + 1]);
                u[2 * rg + 1] = cvt_pk_h16(c3[j][rg * 4 + 2], c3[j][rg * 4 + 3]);
                *(uint2*)(patch + l31 * 128 + (((4 * ff + rg) ^ (l31 & 7)) << 4) + lh * 8) = make_uint2(u[2 * rg], u[2 * rg + 1]);
            }
            xq[2 * j] = __builtin_bit_cast(bf16x8_t, make_uint4(u[0], u[1], u[2], u[3]));
            xq[2 * j + 1] = __builtin_bit_cast(bf16x8_t, make_uint4(u[4], u[5], u[6], u[7]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        unsigned hm = 0u;   // fp16 range guard (gim_common.h): packed maximum of the x' halves this lane stores in this pass
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int px = it * 8 + (lane >> 3), sl = lane & 7;
            const uint4 v = *(const uint4*)(patch + px * 128 + ((sl ^ (px & 7)) << 4));
            *(uint4*)(a.xo + (prow0 + px) * C4 + 64 * q + sl * 8) = v;
            hm = h16_range_fold(hm, v);
        }
        h16_range_check(a.health, hm);   // x': the un-normalised residual stream
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the patch is rewritten by the next pass
        GIM_TT(bneck64, w, 5 + q);
    }
    if constexpr (N1 > 0) {
        // ---- conv1' of the next block: K = 256 in accumulator order, weights from LDS -----------------------------------------
        constexpr int NF = N1 / 32;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        GIM_TT(bneck64, w, 9);
        f32x16_t c1[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float4 bb = *(const float4*)(a.b1n + 32 * f + 8 * rg + 4 * lh);
                c1[f][rg * 4] = bb.x; c1[f][rg * 4 + 1] = bb.y; c1[f][rg * 4 + 2] = bb.z; c1[f][rg * 4 + 3] = bb.w;
            }
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int so = (((2 * s + lh) & 16) | (((2 * s + lh) & 15) ^ (l31 & 15))) << 4;   // rows l31 + 32f share the low-4-bit key
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const bf16x8_t wv = *(const bf16x8_t*)(smem + OFF_W1 + (32 * f + l31) * 512 + so);
                c1[f] = mfma_h16_32x32x16(wv, xq[s], c1[f]);
            }
        }
        GIM_TT(bneck64, w, 10);
#pragma unroll
        for (int h2 = 0; h2 < NF / 2; ++h2) {          // 64 output channels per pass through the patch
#pragma unroll
            for (int ff = 0; ff < 2; ++ff)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    *(uint2*)(patch + l31 * 128 + (((4 * ff + rg) ^ (l31 & 7)) << 4) + lh * 8) =
                        make_uint2(cvt_pk_h16(fmaxf(c1[2 * h2 + ff][rg * 4], 0.f), fmaxf(c1[2 * h2 + ff][rg * 4 + 1], 0.f)),
                                   cvt_pk_h16(fmaxf(c1[2 * h2 + ff][rg * 4 + 2], 0.f), fmaxf(c1[2 * h2 + ff][rg * 4 + 3], 0.f)));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int px = it * 8 + (lane >> 3), sl = lane & 7;
                const uint4 v = *(const uint4*)(patch + px * 128 + ((sl ^ (px & 7)) << 4));
                *(uint4*)(a.t1n + (prow0 + px) * N1 + 64 * h2 + sl * 8) = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    GIM_TT(bneck64, w, 11);
}

}  // namespace

extern "C" int GIM_FN(gim_bneck64_fused_ds)(const void* t1, const void* x_in, void* x_out, void* t1_next, const void* w2, const void* w3,
                                    const void* wds, const void* w1n, const float* b2, const float* b3ds, const float* b1n, int B, int H, int W,
                                    int32_t* health, gim_stream_t stream) {
    GIM_REQUIRE(t1 && x_in && x_out && t1_next && w2 && w3 && wds && w1n && b2 && b3ds && b1n, "bneck64_fused_ds: NULL pointer");
    GIM_REQUIRE(B > 0 && H > 0 && W > 0 && H % TH == 0 && W % TW == 0, "bneck64_fused_ds: H %% 8 == 0 and W %% 32 == 0 required (got %d x %d)", H, W);
    GIM_REQUIRE((int64_t)B * H * W * C4 * 2 < (int64_t)0xFFFFFFF0ll, "bneck64_fused_ds: tensor too large for 32-bit buffer offsets");
    static GimPerDevice attr;
    if (attr.needed()) {
        hipError_t e = hipFuncSetAttribute((const void*)bneck64_kernel<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) { gim_set_error("bneck64_fused_ds: hipFuncSetAttribute(%d B LDS): %s", SMEM, hipGetErrorString(e)); return GIM_ERR_LAUNCH; }
        attr.done();
    }
    Args a;
    a.t1 = (const unsigned short*)t1; a.res = (const unsigned short*)x_in; a.wds = (const unsigned short*)wds; a.xo = (unsigned short*)x_out;
    a.t1n = (unsigned short*)t1_next; a.w2 = (const unsigned short*)w2; a.w3 = (const unsigned short*)w3; a.w1n = (const unsigned short*)w1n;
    a.b2 = b2; a.b3 = b3ds; a.b1n = b1n; a.B = B; a.H = H; a.W = W; a.health = (int*)health;
    a.t1_bytes = (unsigned)((size_t)B * H * W * P * 2); a.w2_bytes = 64 * 1152; a.w3_bytes = 256 * 128; a.w1n_bytes = 64 * 512;
    hipLaunchKernelGGL((bneck64_kernel<64, true>), dim3((unsigned)(B * (H / TH) * (W / TW))), dim3(512), SMEM, (hipStream_t)stream, a);
    return gim_check_launch("bneck64_fused_ds");
}

extern "C" int GIM_FN(gim_bneck64_fused)(const void* t1, const void* res, void* x_out, void* t1_next, const void* w2, const void* w3,
                                 const void* w1n, const float* b2, const float* b3, const float* b1n, int B, int H, int W,
                                 int n_next, int32_t* health, gim_stream_t stream) {
    GIM_REQUIRE(t1 && res && x_out && w2 && w3 && b2 && b3, "bneck64_fused: NULL pointer");
    GIM_REQUIRE((t1_next == nullptr) == (w1n == nullptr) && (t1_next == nullptr || b1n), "bneck64_fused: t1_next, w1n and b1n go together");
    GIM_REQUIRE(t1_next ? (n_next == 64 || n_next == 128) : n_next == 0, "bneck64_fused: n_next must be 64 or 128 with t1_next, 0 without (got %d)", n_next);
    GIM_REQUIRE(B > 0 && H > 0 && W > 0 && H % TH == 0 && W % TW == 0, "bneck64_fused: H %% 8 == 0 and W %% 32 == 0 required (got %d x %d)", H, W);
    GIM_REQUIRE((int64_t)B * H * W * C4 * 2 < (int64_t)0xFFFFFFF0ll, "bneck64_fused: tensor too large for 32-bit buffer offsets");
    static GimPerDevice attr;
    if (attr.needed()) {
        hipError_t e = hipFuncSetAttribute((const void*)bneck64_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)bneck64_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)bneck64_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) { gim_set_error("bneck64_fused: hipFuncSetAttribute(%d B LDS): %s", SMEM, hipGetErrorString(e)); return GIM_ERR_LAUNCH; }
        attr.done();
    }
    Args a;
    a.t1 = (const unsigned short*)t1; a.res = (const unsigned short*)res; a.wds = nullptr; a.xo = (unsigned short*)x_out; a.t1n = (unsigned short*)t1_next;
    a.w2 = (const unsigned short*)w2; a.w3 = (const unsigned short*)w3; a.w1n = (const unsigned short*)w1n;
    a.b2 = b2; a.b3 = b3; a.b1n = b1n; a.B = B; a.H = H; a.W = W; a.health = (int*)health;
    a.t1_bytes = (unsigned)((size_t)B * H * W * P * 2); a.w2_bytes = 64 * 1152; a.w3_bytes = 256 * 128; a.w1n_bytes = (unsigned)n_next * 512;
    const unsigned tiles = (unsigned)(B * (H / TH) * (W / TW));
    if (n_next == 0) hipLaunchKernelGGL(bneck64_kernel<0>, dim3(tiles), dim3(512), SMEM, (hipStream_t)stream, a);
    else if (n_next == 64) hipLaunchKernelGGL(bneck64_kernel<64>, dim3(tiles), dim3(512), SMEM, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(bneck64_kernel<128>, dim3(tiles), dim3(512), SMEM, (hipStream_t)stream, a);
    return gim_check_launch("bneck64_fused");
}
