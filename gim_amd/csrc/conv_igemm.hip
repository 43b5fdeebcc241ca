// Implicit-GEMM convolution / linear layer with fused epilogue for gfx950 (MI355X).
//
//   y[m, n] = act( sum_k A[m,k] * w[n,k] + bias[n] + res[m, n] )
//
// m = output pixel row (b, ho, wo) in NHWC order, n = output channel, k = (dy, dx, c).
// A is never materialised: each 16-byte K group of each pixel row is fetched straight from the NHWC
// activation tensor into LDS by `buffer_load_dwordx4 ... lds` (LDS-DMA); spatial zero padding and
// the K/M tails come for free from the buffer descriptor's bounds check (out-of-range offset -> 0).
//
// Tiling (wave64, MFMA 32x32): a 256-thread workgroup owns a BM x BN output tile, K is walked in
// 128-byte slabs (64 bf16 / 32 fp32 per row), double buffered in LDS.  LDS image per operand is
// [rows][128 B] with the 16-byte slot index XOR-swizzled by ((row >> 1) & 7): LDS-DMA writes are
// lane-linear, so the swizzle is applied to the *source* K group each lane fetches and again on the
// ds_read_b128 side -- conflict-free for the 32x32 fragment pattern (lanes 0-31: rows r..r+31 of one
// slot, lanes 32-63: next slot).
//
// The MFMA computes the transposed tile (weights as the A operand, pixels as B) so that each lane
// ends up with 4 *consecutive channels* of one pixel per accumulator quad -> 8/16-byte NHWC stores.
//
// dtype GIM_H16: v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  dtype GIM_F32: v_mfma_f32_32x32x2_f32,
// bit-equivalent to an fp32 fmaf chain -- the exact-parity mode.
//
// Replaces (reference file:line): networks/loftr/backbone/resnet.py:109-126,230-233,316-327 and the
// nn.Linear calls of networks/loftr/submodules/transformer.py:47-55.
#include "igemm_mainloop.h"

namespace {

using gim::KTB;
GIM_TT_DECL(conv)

// 16-bit output?  (the fp16 objects can also write bf16 -- gim_conv2d_bn_act)
inline bool out_is16(const gim_conv_args& a) { return a.out_dtype == GIM_H16 || (GIM_HALF_KIND && a.out_dtype == GIM_BF16); }

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == GIM_ACT_RELU) return fmaxf(v, 0.f);
    if (act == GIM_ACT_LEAKY) return v > 0.f ? v : 0.01f * v;
    if (act == GIM_ACT_ELU1) return v > 0.f ? v + 1.f : (expf(v) - 1.f) + 1.f;  // elu(x)+1 exactly as torch
    if (act == GIM_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));  // F.gelu (erf form)
    return v;
}

template <int BM, int BN, bool OUT_BF16>
__device__ __forceinline__ void epilogue_rows(const gim_conv_args& a, const float* Ct, int m0, int n0, int M) {
    constexpr int CLD = BN + 4;
    constexpr int G = OUT_BF16 ? 8 : 4;   // channels per 16-byte store
    constexpr int LPR = BN / G;           // lanes per pixel row
    constexpr int RPP = 256 / LPR;        // pixel rows per pass
    const int t = threadIdx.x;
    const int cg = (t % LPR) * G, n = n0 + cg;
    if (n >= a.N) return;
    const int act = (a.act_cols > 0 && n0 >= a.act_cols) ? GIM_ACT_NONE : a.act;
    float bias[G];
#pragma unroll
    for (int e = 0; e < G; ++e) bias[e] = a.bias ? a.bias[n + e] : 0.f;
    unsigned hm = 0u;   // fp16 range guard of a residual stream stored here (gim_common.h)
    for (int r = t / LPR; r < BM; r += RPP) {
        const int m = m0 + r;
        if (m >= M) break;
        float v[G];
#pragma unroll
        for (int e = 0; e < G; e += 4) {
            const float4 c = *(const float4*)(Ct + r * CLD + cg + e);
            v[e] = c.x + bias[e]; v[e + 1] = c.y + bias[e + 1]; v[e + 2] = c.z + bias[e + 2]; v[e + 3] = c.w + bias[e + 3];
        }
        if (a.res) {
            const size_t ro = (size_t)(a.res_mod > 0 ? m % a.res_mod : m) * a.ldres + n;
#pragma unroll
            for (int e = 0; e < G; e += 4) {
                const float4 rr = a.res_dtype == GIM_H16 ? ElemIO<true>::ld4(a.res, ro + e) : ElemIO<false>::ld4(a.res, ro + e);
                v[e] += rr.x; v[e + 1] += rr.y; v[e + 2] += rr.z; v[e + 3] += rr.w;
            }
        }
        if (act != GIM_ACT_NONE) {
#pragma unroll
            for (int e = 0; e < G; ++e) v[e] = apply_act(v[e], act);
        }
        const size_t yo = (size_t)m * a.ldy + n;
        if constexpr (OUT_BF16) {
            uint4 o;
            o.x = cvt_pk_h16(v[0], v[1]); o.y = cvt_pk_h16(v[2], v[3]);
            o.z = cvt_pk_h16(v[4], v[5]); o.w = cvt_pk_h16(v[6], v[7]);
            *(uint4*)((unsigned short*)a.y + yo) = o;
            if (a.res && a.out_dtype == GIM_F16) hm = h16_range_fold_abs(hm, o);   // (fp16 patterns only: the persistent Epilogue's `!out_bf`)
        } else {
            *(float4*)((float*)a.y + yo) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    if constexpr (OUT_BF16) { if (a.out_dtype == GIM_F16) h16_range_check(a.health, hm); }
}

template <int BM, int BN, int WM, int WN, bool BF16, bool LDSDMA>
__global__ void __launch_bounds__(256)
igemm_kernel(const gim_conv_args a, const int mtiles, const int ntiles, const int M) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ES = BF16 ? 2 : 4;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned tile = xcd_remap(blockIdx.x, (unsigned)(mtiles * ntiles));
    const int mt = tile / ntiles, nt = tile - mt * ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave - wm * WN;

    gim::MainloopArgs ml;
    ml.x = a.x; ml.w = a.w; ml.ktab = a.ktab;
    ml.x_bytes = (unsigned)a.x_bytes;
    ml.w_bytes = (unsigned)a.npad * (unsigned)a.kpad * ES;
    ml.H = a.H; ml.W = a.W; ml.Ho = a.Ho; ml.Wo = a.Wo; ml.stride = a.stride; ml.pad = a.pad; ml.ldx = a.ldx;
    ml.kpad = a.kpad; ml.ldw = a.kpad; ml.M = M;
    f32x16_t acc[TN][TM];
    gim::igemm_mainloop<BM, BN, WM, WN, BF16, LDSDMA>(ml, smem, m0, n0, acc);

    // ---- epilogue, phase 1: accumulators -> fp32 tile Ct[pixel][channel] in LDS (stage buffers are free:
    // the main loop ends with a barrier).  Row stride BN+4 floats keeps the 8-lane ds_write_b128 groups
    // (8 consecutive pixels, same channel quad) on distinct banks.
    float* Ct = (float*)smem;
    constexpr int CLD = BN + 4;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int px = wm * WTM + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int ch = wn * WTN + i * 32 + rg * 8 + lh * 4;
                *(float4*)(Ct + px * CLD + ch) = make_float4(acc[i][j][rg * 4 + 0], acc[i][j][rg * 4 + 1],
                                                             acc[i][j][rg * 4 + 2], acc[i][j][rg * 4 + 3]);
            }
    }
    __syncthreads();
    // ---- phase 2: each lane owns one 16-byte channel group of a pixel row -> bias, residual, activation,
    // fully coalesced NHWC stores (a 256-byte bf16 row is written by 16 adjacent lanes).
    if (a.out_dtype == GIM_H16) epilogue_rows<BM, BN, true>(a, Ct, m0, n0, M);
    else epilogue_rows<BM, BN, false>(a, Ct, m0, n0, M);
}

// ------------------------------------------------------------------------------------------------
// Shared epilogue of the persistent kernels (measured: 8-byte per-lane stores at pixel stride cost 4x the
// TCP->TCC requests of full lines and bound the half-resolution 1x1 layers): every wave transposes its
// 64 px x 64 ch sub-tile through a stage buffer the tile has finished with (XOR-swizzled 16-byte slots, no
// extra LDS), so that residual loads and output stores are 16 bytes per lane with a whole 128/256-byte row
// segment per 8/16 adjacent lanes.  Bias enters as the accumulator's initial value; activation kind,
// bounds and dtypes are tile-uniform branches; bf16 rounding is one v_cvt_pk_bf16_f32 per pair.
template <typename G, bool OUT_BF16, bool HAS_RES, bool UPS = false>
struct Epilogue {
    static constexpr int TM = G::TM, TN = G::TN, WTM = G::WTM, WTN = G::WTN, WN = G::WTN == 0 ? 1 : (G::B_BYTES / KTB) / G::WTN;
    static_assert(WTM % 32 == 0 && WTN % 64 == 0, "epilogue transposition works on 32 px x 64 ch passes of the wave tile");
    static constexpr int NH = WTN / 64;            // 64-channel halves of the wave tile
    static_assert(!HAS_RES || NH == 1, "residual prefetch is only built for 64-channel wave tiles");
    static constexpr int OES = OUT_BF16 ? 2 : 4;   // output element size
    static constexpr int RB = 64 * OES;            // bytes of one pixel row of the wave sub-tile
    static constexpr int LPR = RB / 16;            // lanes per row in row layout (8 / 16)
    static constexpr int RPI = 64 / LPR;           // rows per wave instruction (8 / 4)
    static constexpr int NI = 32 / RPI;            // row-layout instructions per 32-pixel pass (4 / 8)
    static constexpr int WAVE_BYTES = 32 * RB;     // transposition tile of one wave
    static_assert(G::NW * WAVE_BYTES <= G::STAGE, "transposition tiles must fit in one stage buffer");
    typedef uint4 Res[(HAS_RES && OUT_BF16) ? TM : 1][(HAS_RES && OUT_BF16) ? NI : 1];

    int wm, wn, l31, lh, rrow, rslot, wave;
    float wscale = 1.f;   // fp32 operands as fp16 hi / lo pairs (Igemm::compute_split16): the weight operand's power-of-two scale -- the bias enters times it, run() divides
    __device__ __forceinline__ Epilogue() {
        const int lane = threadIdx.x & 63;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        l31 = lane & 31; lh = lane >> 5;
        G::wave_mn(wave, wm, wn);
        rrow = lane / LPR; rslot = lane % LPR;
    }

    // acc := bias (the MFMAs accumulate on top of it)
    __device__ __forceinline__ void init_acc(const gim_conv_args& a, typename G::Acc& acc, int n0) const {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.bias) bb = *(const float4*)(a.bias + n0 + wn * WTN + i * 32 + rg * 8 + lh * 4);  // bias is padded to npad
                if constexpr (G::ES == 4) { bb.x *= wscale; bb.y *= wscale; bb.z *= wscale; bb.w *= wscale; }
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    acc[i][j][rg * 4 + 0] = bb.x; acc[i][j][rg * 4 + 1] = bb.y;
                    acc[i][j][rg * 4 + 2] = bb.z; acc[i][j][rg * 4 + 3] = bb.w;
                }
            }
    }

    // coalesced residual rows of the tile -> registers (issued before the tile's last slab is computed)
    // (fp32 rows: not prefetched -- 64 registers held across the last slab's MFMAs spilled 23 registers in round 5's kernel and 267 beside the
    //  split-product loop of round 6; run() loads them pass by pass instead)
    static constexpr bool RES_PREFETCH = HAS_RES && OES == 2;
    __device__ __forceinline__ void prefetch_res(const gim_conv_args& a, Res& rres, int m0, int n0, int M) const {
        if constexpr (RES_PREFETCH) {
            const bool full = (m0 + G::A_BYTES / KTB <= M) && (n0 + G::B_BYTES / KTB <= a.N);
            const int ncol = n0 + wn * WTN + rslot * (16 / OES);
            const bool ncol_ok = full || ncol < a.N;
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int k = 0; k < NI; ++k) {
                    const int m = m0 + wm * WTM + j * 32 + k * RPI + rrow;
                    const bool ok = ncol_ok && (full || m < M);
                    const size_t ro = (size_t)(a.res_mod > 0 ? m % a.res_mod : m) * a.ldres + ncol;
                    rres[j][k] = ok ? *(const uint4*)((const char*)a.res + ro * OES) : make_uint4(0u, 0u, 0u, 0u);
                }
        }
    }

    // UPS: acc += bilinear x2 (align_corners=True) of a.ups -- the FPN's lateral 1x1 conv + F.interpolate + add (resnet.py:321-327) in one launch.
    // Rounds 3-5 blended on the VALU inside the store loop (four neighbours per output from an LDS patch): 29 000 cycles per 256 x 256 tile
    // against 9 000 for the plain epilogue, ~1 750 VALU instructions per wave and tile of which the 16-bit unpacks cost as much as the
    // arithmetic (profiles/r06_halo_variants.txt, r06_valu_rates.txt).  Round 6: the interpolation is 48 more K of the MFMA.  The 32 pixels of
    // a pass lie in one output row (2 ups_w % 32 == 0, checked at launch); they read 2 source rows x <= 18 source columns.  Per pass
    // (32 px x 64 ch) the wave stages those sources by LDS-DMA as rows [s][64 ch] with s = 16 ks + 8 r + t <-> source row r, column
    // 8 ks + t (six 1 KiB pieces, each 8 consecutive columns of one source row), and runs, per 32-channel fragment and k16 step ks,
    //     acc[ch][px] += A[ch][s] * B[s][px]     A = the staged sources read TRANSPOSED (ds_read_b64_tr_b16: a 16-lane group reads
    //                                            [4 sources][16 channels] and every lane receives one channel of the four sources),
    //                                            B = the pixel's bilinear weights (wy of row r = lane half) x (lx0 at column x0, lx1 at x1),
    //                                            built in registers and rounded to the operand type
    // i.e. 6 MFMAs per pass, 24 per tile and wave, in front of the plain epilogue: conv + upsample are summed in fp32 and rounded once
    // (the two-pass path rounds the conv output first).  Two patches per wave (its transposition tile and its half of the stage's second
    // 32 KiB, rows 32..47 in 4 KiB per wave behind the stages): the request of pass p + 2 goes out when pass p is consumed.
    __device__ __forceinline__ void ups_accumulate(const gim_conv_args& a, typename G::Acc& acc, char* stage, char* patch2, int m0, int n0, int M) const {
        static_assert(WAVE_BYTES == 4096 && G::NW * (WAVE_BYTES + 4096) <= G::STAGE, "upsample patches: LDS map");
        const int h = a.ups_h, w = a.ups_w, W2 = 2 * w, H2 = 2 * h;
        const float sy = (float)(h - 1) / (float)(H2 - 1), sx = (float)(w - 1) / (float)(W2 - 1);
        const int lane = threadIdx.x & 63, l15 = lane & 15;
        constexpr int NP = TM * NH;   // passes of the wave tile
        char* const pm[2] = {stage + wave * WAVE_BYTES, stage + G::NW * WAVE_BYTES + wave * 4096};   // rows 0..31 of patch 0 / 1
        char* const pt[2] = {patch2 + wave * 4096, patch2 + wave * 4096 + 2048};                     // rows 32..47
        const gim_u32x4_t ur = gim_make_rsrc(a.ups, (unsigned)((size_t)(M / (4 * h * w)) * h * w * a.ups_ld * 2));
        // (16-byte slot of channel group g in row s: g ^ 2 ((s >> 1) & 1) -- the four rows of a transposed read then cover 128 B of distinct banks)
        auto issue = [&](const int p) __attribute__((always_inline)) {
            const int jj = p / NH, nn = p % NH;
            const int mb = m0 + wm * WTM + jj * 32;
            const int pr = mb / W2, X0 = mb - pr * W2, ib = pr / H2, Y = pr - ib * H2;
            const float fy = sy * Y;
            const int y0 = (int)fy, y1 = y0 + (y0 < h - 1 ? 1 : 0);
            const int xa = (int)(sx * X0);
            const int row8 = lane >> 3;
            const int g = (lane & 7) ^ (2 * ((row8 >> 1) & 1));
            const int nc = n0 + wn * WTN + nn * 64 + g * 8;
#pragma unroll
            for (int q = 0; q < 6; ++q) {   // piece q: source row q & 1, columns 8 (q >> 1) .. + 7 (beyond the row's last pixel: that pixel again, weight 0)
                int c = xa + 8 * (q >> 1) + row8;
                c = c < w - 1 ? c : w - 1;
                const unsigned voff = (unsigned)((((size_t)ib * h + ((q & 1) ? y1 : y0)) * w + c) * a.ups_ld + nc) * 2u;
                char* dst = q < 4 ? pm[p & 1] + q * 1024 : pt[p & 1] + (q - 4) * 1024;
                gim_dma16(ur, (unsigned)(size_t)(__attribute__((address_space(3))) void*)dst, voff);
            }
        };
        issue(0);
        if (NP > 1) issue(1);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int j = p / NH, nh = p % NH;   // compile-time
            // ---- this pass's weights: pixel l31 of the pass, source row = lane half --------------------------------------------
            const int mb = m0 + wm * WTM + j * 32;
            const int pr = mb / W2, uX0 = mb - pr * W2, uY = pr - (pr / H2) * H2;
            const float fy = sy * uY;
            const float ly1 = fy - (float)(int)fy;
            const float wy = lh ? ly1 : 1.f - ly1;
            const float fx = sx * (float)(uX0 + l31);
            const int x0 = (int)fx, xa = (int)(sx * uX0);
            const int x0r = x0 - xa, x1r = x0r + (x0 < w - 1 ? 1 : 0);
            const float lx1 = (fx - (float)x0) * wy, lx0 = wy - lx1;   // (1 - lx1) wy
            bf16x8_t wb[3];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                unsigned u[4];
#pragma unroll
                for (int t = 0; t < 8; t += 2) {
                    const int c = 8 * ks + t;
                    const float w0 = (c == x0r ? lx0 : 0.f) + (c == x1r ? lx1 : 0.f);
                    const float w1 = (c + 1 == x0r ? lx0 : 0.f) + (c + 1 == x1r ? lx1 : 0.f);
                    u[t >> 1] = cvt_pk_h16(w0, w1);
                }
                wb[ks] = __builtin_bit_cast(bf16x8_t, make_uint4(u[0], u[1], u[2], u[3]));
            }
            // ---- the patch has landed (younger in the queue: the six requests of pass p + 1) ---------------------------------------
            if (p + 1 < NP) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            typedef short s4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                const char* base = ks < 2 ? pm[p & 1] + ks * 2048 : pt[p & 1];   // rows 16 ks .. 16 ks + 15
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    // lane -> (source 8 lh + 4 e + (l15 >> 2), channels 32 i + 16 (lane >> 4 & 1) + 4 (l15 & 3) .. + 3) of its 16-lane group's block
                    const int ch = 32 * i + 16 * ((lane >> 4) & 1) + 4 * (l15 & 3);
                    s4 lo4, hi4;
                    {
                        const int row = 8 * lh + (l15 >> 2);
                        const char* q = base + row * 128 + ((((ch >> 3) ^ (2 * ((row >> 1) & 1))) << 4) | ((ch & 7) << 1));
                        lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)q);
                    }
                    {
                        const int row = 8 * lh + 4 + (l15 >> 2);
                        const char* q = base + row * 128 + ((((ch >> 3) ^ (2 * ((row >> 1) & 1))) << 4) | ((ch & 7) << 1));
                        hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)q);
                    }
                    const bf16x8_t av = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                    acc[nh * 2 + i][j] = mfma_h16_32x32x16(av, wb[ks], acc[nh * 2 + i][j]);
                }
            }
            if (p + 2 < NP) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the patch is consumed: the request after next overwrites it
                issue(p + 2);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // patch 0 / 1's first 4 KiB are the transposition tile / nothing the plain epilogue touches
    }

    // residual + activation in accumulator layout, transposition through `stage` (a stage buffer no wave reads
    // any more), coalesced stores.  The caller must barrier before the stage is overwritten again.
    __device__ __forceinline__ void run(const gim_conv_args& a, typename G::Acc& acc, const Res& rres, char* stage,
                                        int m0, int n0, int M, char* patch2 = nullptr) const {
        if constexpr (UPS && OUT_BF16) ups_accumulate(a, acc, stage, patch2, m0, n0, M);
        if constexpr (G::ES == 4) {
            if (wscale != 1.f) {   // wave-uniform
                const float inv = 1.f / wscale;   // a power of two: exact
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] *= inv;
            }
        }
        char* wl = stage + wave * WAVE_BYTES;  // this wave's transposition tile [32 px][RB]
        const bool full = (m0 + G::A_BYTES / KTB <= M) && (n0 + G::B_BYTES / KTB <= a.N);
        const int act = (a.act_cols > 0 && n0 >= a.act_cols) ? GIM_ACT_NONE : a.act;  // tile-uniform
        // kind of a 16-bit output: the flavour of this translation unit, except that the fp16 objects can also write bf16
        // (no residual / upsample operand then: checked at launch)
        const bool out_bf = GIM_HALF_KIND ? a.out_dtype == GIM_BF16 : true;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
#pragma unroll
            for (int nh = 0; nh < NH; ++nh) {  // one 32 px x 64 ch pass
                const int ncol = n0 + wn * WTN + nh * 64 + rslot * (16 / OES);
                const bool ncol_ok = full || ncol < a.N;
                if constexpr (HAS_RES) {
                    if constexpr (RES_PREFETCH) {
#pragma unroll
                        for (int k = 0; k < NI; ++k) {
                            const int row = k * RPI + rrow;
                            *(uint4*)(wl + row * RB + ((rslot ^ (row & 7)) << 4)) = rres[j][k];
                        }
                    } else {
                        uint4 rr[NI];
#pragma unroll
                        for (int k = 0; k < NI; ++k) {
                            const int m = m0 + wm * WTM + j * 32 + k * RPI + rrow;
                            const bool ok = ncol_ok && (full || m < M);
                            const size_t ro = (size_t)(a.res_mod > 0 ? m % a.res_mod : m) * a.ldres + ncol;
                            rr[k] = ok ? *(const uint4*)((const char*)a.res + ro * OES) : make_uint4(0u, 0u, 0u, 0u);
                        }
#pragma unroll
                        for (int k = 0; k < NI; ++k) {
                            const int row = k * RPI + rrow;
                            *(uint4*)(wl + row * RB + ((rslot ^ (row & 7)) << 4)) = rr[k];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            if constexpr (OUT_BF16) {
                                const uint2 u = *(const uint2*)(wl + l31 * RB + (((i * 4 + rg) ^ (l31 & 7)) << 4) + lh * 8);
                                acc[i][j][rg * 4 + 0] += h16_lo(u.x);
                                acc[i][j][rg * 4 + 1] += h16_hi(u.x);
                                acc[i][j][rg * 4 + 2] += h16_lo(u.y);
                                acc[i][j][rg * 4 + 3] += h16_hi(u.y);
                            } else {
                                const float4 rr = *(const float4*)(wl + l31 * RB + (((i * 8 + rg * 2 + lh) ^ (l31 & 7)) << 4));
                                acc[i][j][rg * 4 + 0] += rr.x; acc[i][j][rg * 4 + 1] += rr.y;
                                acc[i][j][rg * 4 + 2] += rr.z; acc[i][j][rg * 4 + 3] += rr.w;
                            }
                        }
                }
                if (act == GIM_ACT_RELU) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[nh * 2 + i][j][r] = fmaxf(acc[nh * 2 + i][j][r], 0.f);
                } else if (act == GIM_ACT_LEAKY) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            acc[nh * 2 + i][j][r] = fmaxf(acc[nh * 2 + i][j][r], 0.f) + 0.01f * fminf(acc[nh * 2 + i][j][r], 0.f);
                } else if (act == GIM_ACT_ELU1) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[nh * 2 + i][j][r] = apply_act(acc[nh * 2 + i][j][r], GIM_ACT_ELU1);
                } else if (act == GIM_ACT_GELU) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[nh * 2 + i][j][r] = apply_act(acc[nh * 2 + i][j][r], GIM_ACT_GELU);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const f32x16_t& v = acc[nh * 2 + i][j];
                        if constexpr (OUT_BF16) {
                            *(uint2*)(wl + l31 * RB + (((i * 4 + rg) ^ (l31 & 7)) << 4) + lh * 8) =
                                make_uint2(cvt_pk_16(v[rg * 4 + 0], v[rg * 4 + 1], out_bf), cvt_pk_16(v[rg * 4 + 2], v[rg * 4 + 3], out_bf));
                        } else {
                            *(float4*)(wl + l31 * RB + (((i * 8 + rg * 2 + lh) ^ (l31 & 7)) << 4)) =
                                make_float4(v[rg * 4 + 0], v[rg * 4 + 1], v[rg * 4 + 2], v[rg * 4 + 3]);
                        }
                    }
                unsigned hm = 0u;   // fp16 range guard: an un-normalised residual stream is stored here (x + identity; gim_common.h)
#pragma unroll
                for (int k = 0; k < NI; ++k) {
                    const int row = k * RPI + rrow;
                    const int m = m0 + wm * WTM + j * 32 + row;
                    const uint4 o = *(const uint4*)(wl + row * RB + ((rslot ^ (row & 7)) << 4));
                    if (ncol_ok && (full || m < M)) {
                        *(uint4*)((char*)a.y + ((size_t)m * a.ldy + ncol) * OES) = o;
                        if constexpr (HAS_RES && OUT_BF16) hm = h16_range_fold_abs(hm, o);
                    }
                }
                if constexpr (HAS_RES && OUT_BF16) { if (!out_bf) h16_range_check(a.health, hm); }
            }
        }
    }
};

// tile list of a persistent block: first, first + step, ... < end, handed out in XCD-contiguous chunks
// (block b runs on XCD b % 8) so that tiles sharing an activation panel are in flight on the same L2.
__device__ __forceinline__ void tile_list(unsigned T, unsigned& first, unsigned& step, unsigned& end) {
    const unsigned nb = gridDim.x, b = blockIdx.x;
    if (nb >= 8 && T >= 16) {
        const unsigned xcd = b & 7u, slot = b >> 3;
        const unsigned q = T >> 3, r = T & 7u;
        const unsigned cstart = xcd * q + (xcd < r ? xcd : r);
        end = cstart + q + (xcd < r ? 1u : 0u);
        step = (nb >> 3) + (xcd < (nb & 7u) ? 1u : 0u);
        first = cstart + slot;
    } else {
        first = b; step = nb; end = T;
    }
}

__device__ __forceinline__ gim::MainloopArgs mainloop_args(const gim_conv_args& a, int M, int ES) {
    gim::MainloopArgs ml;
    ml.x = a.x; ml.w = a.w; ml.ktab = a.ktab;
    ml.x_bytes = (unsigned)a.x_bytes;
    ml.w_bytes = (unsigned)a.npad * (unsigned)a.kpad * ES;
    ml.H = a.H; ml.W = a.W; ml.Ho = a.Ho; ml.Wo = a.Wo; ml.stride = a.stride; ml.pad = a.pad; ml.ldx = a.ldx;
    ml.kpad = a.kpad; ml.ldw = a.kpad; ml.M = M;
    return ml;
}

// ------------------------------------------------------------------------------------------------
// Persistent, cross-tile software-pipelined variant, 4 waves, 2 LDS stages (2 workgroups per CU).
//
// Each workgroup walks a list of output tiles; K slabs of consecutive tiles form ONE flattened stream:
// while slab s is on the MFMAs, the LDS-DMA of slab s+1 (possibly the first slab of the *next* tile) and
// the residual rows of the current tile are already in flight, and the stores of the previous tile's
// epilogue drain in the background.
template <int N> struct IntC { static constexpr int value = N; };

// SKIP: waves whose last 32-channel fragment lies entirely beyond N run a K loop without it (a second copy of the loop,
// selected per tile by a wave-uniform branch; see Igemm::mma)
template <int BM, int BN, int WM, int WN, bool BF16, bool OUT_BF16, bool HAS_RES, bool SKIP = false, bool UPS = false>
__global__ void __launch_bounds__(WM * WN * 64, 2)  // 2 waves per SIMD: 2 x 4-wave or 1 x 8-wave workgroup per CU
igemm_persistent_kernel(const gim_conv_args a, const int mtiles, const int ntiles, const int M) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef gim::Igemm<BM, BN, WM, WN, BF16, true> G;
    typedef Epilogue<G, OUT_BF16, HAS_RES, UPS> E;

    unsigned first, step, end;
    tile_list((unsigned)(mtiles * ntiles), first, step, end);
    if (first >= end) return;
    const gim::MainloopArgs ml = mainloop_args(a, M, G::ES);
    const int nkt = a.kpad * G::ES / KTB;

    E epi;
    if constexpr (!BF16) { if (a.split16) epi.wscale = 4096.f; }
    GIM_TT(conv, epi.wave, 0);
    unsigned long long tt_k = 0, tt_e = 0, tt_n = 0, tt_a = 0, tt_b = 0;   // GIM_TIMING: K-loop / epilogue totals over this workgroup's tiles
    (void)tt_k; (void)tt_e; (void)tt_n; (void)tt_a; (void)tt_b;
    G g, gn;  // staging coordinates of the current / the next tile
    typename G::Acc acc;
    typename E::Res rres;
    int buf = 0;
    int m0 = (int)(first / ntiles) * BM, n0 = (int)(first % ntiles) * BN;
    epi.init_acc(a, acc, n0);
    g.decode(ml, m0, n0);
    g.stage_issue(ml, smem, 0, 0, a.ktab[G::ktab_index(0)]);
    int e_nxt = a.ktab[G::ktab_index(nkt > 1 ? 1 : 0)];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (unsigned tile = first; tile < end; tile += step) {
        const unsigned tile_n = tile + step;
        const bool has_next = tile_n < end;
        const int m0n = (int)(tile_n / ntiles) * BM, n0n = (int)(tile_n % ntiles) * BN;
        if (has_next) gn.decode(ml, m0n, n0n);
        tt_a = GIM_TT_NOW();
        // ---- K loop: only MFMAs touch the accumulators in here ------------------------------------------
        auto kloop = [&](auto live, auto split16) __attribute__((always_inline)) {
            for (int kt = 0; kt < nkt; ++kt) {
                const bool last = kt + 1 == nkt;
                int k2 = kt + 2;
                if (k2 >= nkt) k2 -= nkt;
                if (k2 >= nkt) k2 = 0;  // nkt == 1
                const int e_n2 = a.ktab[G::ktab_index(k2)];
                if (!last) g.stage_issue(ml, smem, buf ^ 1, kt + 1, e_nxt);
                else if (has_next) gn.stage_issue(ml, smem, buf ^ 1, 0, e_nxt);  // first slab of the next tile
                if (last) epi.prefetch_res(a, rres, m0, n0, M);
                if constexpr (decltype(split16)::value != 0) G::template compute_split16<decltype(live)::value>(smem, buf, acc, epi.wscale);
                else G::template compute<decltype(live)::value>(smem, buf, acc);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                buf ^= 1;
                e_nxt = e_n2;
            }
        };
        if constexpr (SKIP && G::TN > 1) {
            // the wave's last channel fragment holds only padding channels (wave-uniform)
            if (n0 + epi.wn * G::WTN + (G::TN - 1) * 32 >= a.N) kloop(IntC<G::TN - 1>(), IntC<BF16 ? 0 : 1>());   // (fp32 operands reach this tile as split launches only)
            else kloop(IntC<G::TN>(), IntC<BF16 ? 0 : 1>());
        } else if constexpr (!BF16) {
            if constexpr (BM == 256 && BN == 256) kloop(IntC<G::TN>(), IntC<1>());   // (only split launches are sent to this tile: dispatch_persistent)
            else if (a.split16) kloop(IntC<G::TN>(), IntC<1>());   // (a second copy of the loop, selected per launch)
            else kloop(IntC<G::TN>(), IntC<0>());
        } else {
            kloop(IntC<G::TN>(), IntC<0>());
        }
        tt_b = GIM_TT_NOW(); tt_k += tt_b - tt_a;
        epi.run(a, acc, rres, smem + (buf ^ 1) * G::STAGE, m0, n0, M, smem + 2 * G::STAGE);  // buf ^ 1: the stage just consumed; UPS: patch rows behind the stages
        epi.init_acc(a, acc, n0n < a.npad ? n0n : 0);
        g = gn;
        m0 = m0n; n0 = n0n;
        __syncthreads();  // the transposition tile lives in a stage buffer the next slab's DMA will overwrite
        tt_e += GIM_TT_NOW() - tt_b; ++tt_n;
    }
    GIM_TT(conv, epi.wave, 1);
    GIM_TT_SET(conv, epi.wave, 4, tt_k); GIM_TT_SET(conv, epi.wave, 5, tt_e); GIM_TT_SET(conv, epi.wave, 6, tt_n);
}


template <int BM, int BN, int WM, int WN, bool BF16, bool OUT_BF16, bool HAS_RES, bool SKIP = false, bool UPS = false>
int launch_persistent(const gim_conv_args& a, hipStream_t stream) {
    constexpr int smem = 2 * (BM + BN) * KTB + (UPS ? WM * WN * 4096 : 0);   // UPS: rows 32..47 of the two upsample patches of every wave (Epilogue::ups_accumulate)
    static_assert(smem <= 160 * 1024, "LDS");
    auto kern = igemm_persistent_kernel<BM, BN, WM, WN, BF16, OUT_BF16, HAS_RES, SKIP, UPS>;
    static GimPerDevice attr_done;
    if (attr_done.needed()) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) {
            gim_set_error("hipFuncSetAttribute(%d B LDS): %s", smem, hipGetErrorString(e));
            return GIM_ERR_LAUNCH;
        }
        attr_done.done();
    }
    const int M = a.B * a.Ho * a.Wo;
    const int mtiles = (M + BM - 1) / BM, ntiles = a.npad / BN;
    const int T = mtiles * ntiles;
    constexpr int NT = WM * WN * 64;
    constexpr int RESIDENT = (NT == 256 ? 2 : 1) * 256;  // workgroups per CU (LDS bound) x 256 CUs
    // one workgroup per resident slot (tile counts differ by at most one).  Until round 5 the grid was ceil(T / rounds) -- every block the
    // same tile count, but e.g. layer 3's 1 200 tiles ran as 400 blocks of 3 on 512 slots: 144 CUs carried six tiles, 112 three.  Same-box
    // A/B of the forward: -0.08 ms per batch-8 step (profiles/r05_launch_order.txt)
    const int grid = T < RESIDENT ? T : RESIDENT;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), smem, stream, a, mtiles, ntiles, M);
    return gim_check_launch("igemm_persistent_kernel");
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with a HALO tile as the pixel operand (bf16 in / out, no residual).
//
// The implicit GEMM above stages every (tap, 64-channel) K slab of the pixel operand separately: the nine taps of a 3x3 conv
// fetch the same input pixels nine times from L2 into LDS, and the PMC / variant experiments show the tile bound by
// what it stages per MFMA, not by its schedule.  Here a workgroup owns an 8 x 32 patch of output pixels of one image: per
// 64-channel chunk it stages the (8+2) x (32+2) halo of input pixels ONCE (340 rows x 128 B = 43 LDS-DMA pieces instead of
// 9 x 32) and the nine taps read their fragments from it at shifted rows -- row(py, px, dy, dx) = (py + dy) * 34 + px + dx.
// The XOR swizzle ((row >> 1) & 7 on the 16-byte slot) stays conflict-free for ANY row offset: a ds_read_b128 lane group
// covers 16 rows whose low four bits are all different, shifted or not.  The weight operand streams as before, one
// [BN][64] slab per K step of four 16-channel sub-steps; K order is (chunk, tap, channel) for the full chunks and
// (tap, channel) for the last, narrower one (Cin = 196: three chunks + nine 16-channel sub-steps packed four to a slab), a
// small per-slab table in LDS tells every sub-step its row shift and its channel offset inside the halo row.
constexpr int HTH = 8, HTW = 32, HW2 = HTW + 2, HROWS = (HTH + 2) * HW2;  // 340 halo rows
constexpr int HPIECES = (HROWS + 7) / 8;                                 // 43 LDS-DMA pieces of 8 rows
constexpr int HA_BYTES = HPIECES * 8 * KTB;                              // 44032 B per halo buffer
constexpr int HNPA = (HPIECES + 7) / 8;                                  // pieces per wave (6)

template <int TN>  // wave tile: 64 pixels x TN * 32 channels; BN = 2 * TN * 32
__global__ void __launch_bounds__(512, 2)
conv3x3_halo_kernel(const gim_conv_args a, const int tiles_x, const int tiles_y, const int ntiles, const int nslab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = 2 * TN * 32, PB = BN / 64;
    typedef gim::Igemm<BM, BN, 4, 2, true, true> G;
    typedef Epilogue<G, true, false> E;
    char* const sAh = smem;                    // [2][HA_BYTES]
    char* const sBb = smem + 2 * HA_BYTES;     // [2][BN * KTB]
    int* const tab = (int*)(sBb + 2 * BN * KTB);

    unsigned first, step, end;
    tile_list((unsigned)(tiles_x * tiles_y * a.B * ntiles), first, step, end);
    if (first >= end) return;
    for (int i = threadIdx.x; i < nslab * 8; i += 512) tab[i] = a.ktab[i];

    const int t = threadIdx.x, lane = t & 63;
    E epi;  // wave, wm, wn, l31, lh, rrow, rslot
    const int wave = epi.wave;
    GIM_TT(conv, wave, 0);
    unsigned long long tt_k = 0, tt_e = 0, tt_n = 0, tt_a = 0, tt_b = 0;
    (void)tt_k; (void)tt_e; (void)tt_n; (void)tt_a; (void)tt_b;
    const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)((unsigned)a.npad * (unsigned)a.kpad * 2u), 0x00020000);

    // ---- per-thread staging coordinates ---------------------------------------------------------------------------------
    unsigned offA[HNPA];   // byte offset of (pixel, channel group) of this thread's slot in piece i, 0xFFFFFFFF = outside the image
    int gch[HNPA];         // first channel (within the chunk) of that slot
    auto decode = [&](const unsigned tile) {
        const int mt = (int)(tile / (unsigned)ntiles);
        const int tx = mt % tiles_x, ty = (mt / tiles_x) % tiles_y, b = mt / (tiles_x * tiles_y);
        const int x0 = tx * HTW, y0 = ty * HTH;
#pragma unroll
        for (int i = 0; i < HNPA; ++i) {
            const int q = wave + 8 * i, r = 8 * q + (lane >> 3);
            const int hy = r / HW2, hx = r - hy * HW2;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool ok = (q < HPIECES) & (r < HROWS) & ((unsigned)y < (unsigned)a.H) & ((unsigned)x < (unsigned)a.W);
            const int g = (lane & 7) ^ ((r >> 1) & 7);
            gch[i] = g * 8;
            offA[i] = ok ? ((unsigned)((b * a.H + y) * a.W + x) * (unsigned)a.ldx + (unsigned)(g * 8)) * 2u : 0xFFFFFFFFu;
        }
    };
    const int cin_pad = a.res_mod;  // halo mode: the channel count of x rows that holds data (cin_pad <= ldx); no residual here
    auto issue_A = [&](const int c0, const int ab) {  // halo of channels [c0, c0 + 64) -> sAh[ab]
#pragma unroll
        for (int i = 0; i < HNPA; ++i) {
            const int q = wave + 8 * i;
            if (q < HPIECES) {
                const bool ok = (offA[i] != 0xFFFFFFFFu) & (c0 + gch[i] < cin_pad);
                const unsigned voff = ok ? offA[i] + (unsigned)c0 * 2u : (unsigned)a.x_bytes;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)(sAh + ab * HA_BYTES + q * 8 * KTB), 16, voff, 0, 0, 0);
            }
        }
    };
    const int srow = t >> 3, sgrp = (t & 7) ^ ((srow >> 1) & 7);
    auto issue_B = [&](const int n0, const int slab, const int bb) {  // weight slab -> sBb[bb]
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const unsigned voff = ((unsigned)(n0 + i * 64 + srow) * (unsigned)a.kpad + (unsigned)(slab * 64 + sgrp * 8)) * 2u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(sBb + bb * BN * KTB + (i * 64 + wave * 8) * KTB), 16, voff, 0, 0, 0);
        }
    };

    typename G::Acc acc;
    unsigned tile = first;
    int n0 = (int)(tile % (unsigned)ntiles) * BN;
    epi.init_acc(a, acc, n0);
    decode(tile);
    issue_A(0, 0);
    issue_B(n0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // fragment addressing
    const int lswz = (epi.l31 >> 1) & 7;
    const int rb0 = (2 * epi.wm) * HW2 + epi.l31;  // halo row of the wave's first pixel row at tap (0, 0)
    int ab = 0, bb = 0;

    for (; tile < end; tile += step) {
        const unsigned tile_n = tile + step;
        const bool has_next = tile_n < end;
        const int n0n = has_next ? (int)(tile_n % (unsigned)ntiles) * BN : n0;
        const int mt = (int)(tile / (unsigned)ntiles);
        auto kloop = [&](auto live) __attribute__((always_inline)) {
        constexpr int LIVE = decltype(live)::value;
        for (int s = 0; s < nslab; ++s) {
            const int* te = tab + s * 8;
            const int flags = __builtin_amdgcn_readfirstlane(te[1]);
            const bool last = s + 1 == nslab;
            // ---- prefetch: next weight slab, and (at the first slab of a chunk) the next halo -------------------------------
            if (!last) issue_B(n0, s + 1, bb ^ 1);
            else if (has_next) issue_B(n0n, 0, bb ^ 1);
            if (flags & 1) {
                const int cnext = __builtin_amdgcn_readfirstlane(te[6]);  // channel base of the next chunk, -1: the tile's last chunk
                if (cnext >= 0) issue_A(cnext, ab ^ 1);
                else if (has_next) { decode(tile_n); issue_A(0, ab ^ 1); }
            }
            // ---- MFMAs of this slab ---------------------------------------------------------------------------------------
            const char* sA = sAh + ab * HA_BYTES;
            const char* sB = sBb + bb * BN * KTB + (epi.wn * G::WTN + epi.l31) * KTB;
            auto load = [&](const int ks, typename G::template Frags<LIVE>& f) __attribute__((always_inline)) {
                const int e = __builtin_amdgcn_readfirstlane(te[2 + ks]);  // row shift | channel sub-step << 8
                const int shift = e & 0xff, ksc = (e >> 8) & 0xff;
#pragma unroll
                for (int j = 0; j < G::TM; ++j) {
                    const int row = rb0 + j * HW2 + shift;
                    f.a[j] = *(const bf16x8_t*)(sA + row * KTB + ((((2 * ksc + epi.lh) ^ (row >> 1)) & 7) << 4));
                }
                const int so = ((2 * ks + epi.lh) ^ lswz) << 4;
#pragma unroll
                for (int i = 0; i < LIVE; ++i) f.b[i] = *(const bf16x8_t*)(sB + i * 32 * KTB + so);
            };
            // (unused sub-steps of the last slab carry zero weights: no branch in here)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {  // (loading step ks + 1 ahead of the MFMAs of step ks by hand compiles to the same schedule)
                typename G::template Frags<LIVE> f;
                load(ks, f);
                G::template mma<LIVE>(acc, f);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            bb ^= 1;
            if (flags & 2) ab ^= 1;  // last slab of its chunk
        }
        };
        // (no fragment skipping here: a second copy of this K loop spills; layers with an all-padding fragment stay on the
        // generic kernel, which skips it)
        tt_a = GIM_TT_NOW();
        kloop(IntC<TN>());
        tt_b = GIM_TT_NOW(); tt_k += tt_b - tt_a;
        // ---- epilogue: activation, bf16, per-wave transposition through the halo buffer just consumed (ab ^ 1), row stores ------
        {
            const int tx = mt % tiles_x, ty = (mt / tiles_x) % tiles_y, b = mt / (tiles_x * tiles_y);
            const int x0 = tx * HTW, y0 = ty * HTH;
            char* wl = (char*)sAh + (ab ^ 1) * HA_BYTES + wave * E::WAVE_BYTES;
            const int act = a.act;
#pragma unroll
            for (int j = 0; j < G::TM; ++j) {
                const int y = y0 + 2 * epi.wm + j;
#pragma unroll
                for (int nh = 0; nh < E::NH; ++nh) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[nh * 2 + i][j][r] = apply_act(acc[nh * 2 + i][j][r], act);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            const f32x16_t& v = acc[nh * 2 + i][j];
                            *(uint2*)(wl + epi.l31 * E::RB + (((i * 4 + rg) ^ (epi.l31 & 7)) << 4) + epi.lh * 8) =
                                make_uint2(cvt_pk_h16(v[rg * 4 + 0], v[rg * 4 + 1]), cvt_pk_h16(v[rg * 4 + 2], v[rg * 4 + 3]));
                        }
                    const int ncol = n0 + epi.wn * G::WTN + nh * 64 + epi.rslot * 8;
#pragma unroll
                    for (int k = 0; k < E::NI; ++k) {
                        const int row = k * E::RPI + epi.rrow, x = x0 + row;
                        const uint4 o = *(const uint4*)(wl + row * E::RB + ((epi.rslot ^ (row & 7)) << 4));
                        if (ncol < a.N && y < a.H && x < a.W)
                            *(uint4*)((char*)a.y + ((size_t)((b * a.H + y) * a.W + x) * a.ldy + ncol) * 2) = o;
                    }
                }
            }
        }
        epi.init_acc(a, acc, n0n);
        n0 = n0n;
        __syncthreads();  // the transposition tiles live in a halo buffer the next chunk's DMA will overwrite
        tt_e += GIM_TT_NOW() - tt_b; ++tt_n;
    }
    GIM_TT(conv, wave, 1);
    GIM_TT_SET(conv, wave, 4, tt_k); GIM_TT_SET(conv, wave, 5, tt_e); GIM_TT_SET(conv, wave, 6, tt_n);
}

template <int TN>
int launch_halo(const gim_conv_args& a, hipStream_t stream) {
    constexpr int BN = 2 * TN * 32;
    const int nslab = a.kpad / 64;
    const int smem = 2 * HA_BYTES + 2 * BN * KTB + nslab * 8 * 4;
    GIM_REQUIRE(smem <= 160 * 1024, "conv3x3 halo: %d slabs do not fit the LDS table", nslab);
    auto kern = conv3x3_halo_kernel<TN>;
    static GimPerDevice attr_done;
    if (attr_done.needed()) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            gim_set_error("hipFuncSetAttribute(160 KiB LDS): %s", hipGetErrorString(e));
            return GIM_ERR_LAUNCH;
        }
        attr_done.done();
    }
    const int tiles_x = (a.W + HTW - 1) / HTW, tiles_y = (a.H + HTH - 1) / HTH, ntiles = a.npad / BN;
    const int T = tiles_x * tiles_y * a.B * ntiles;
    constexpr int RESIDENT = 256;  // one workgroup per CU
    const int grid = T < RESIDENT ? T : RESIDENT;   // (see launch_persistent)
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, stream, a, tiles_x, tiles_y, ntiles, nslab);
    return gim_check_launch("conv3x3_halo_kernel");
}

// Tile-selection thresholds of dispatch_persistent (round-3 sweep, profiles/r03_knob_sweep.txt): the 256 x 256 / 8-wave tile takes a layer
// with at least BIG_MIN_NKT K slabs and BIG_MIN_TILES tiles; its all-padding column fragment is skipped for N <= npad - 32.
// Round 5: the experimental main-loop variants (3-stage ring, ping-pong, loader waves: each exact, each measured no faster -- DESIGN.md
// section 4, "what the counters said") and their GIM_IGEMM_* switches are gone from the library; git history has them.
constexpr int BIG_MIN_TILES = 1024, BIG_MIN_NKT = 4;
// Round 6: ... or between half a round and ONE round of the 256 one-per-CU workgroups (128 <= tiles <= 256).  Such a launch leaves CUs idle on its own, but it is
// what layer 3 of one image chain of the batch-8 benchmark is (150 tiles: the other chain's kernels run on the idle CUs), and alone it costs about what the 600
// 128 x 128 tiles in 1.17 rounds do.  Same box, two chains: 9.59 vs 9.73 ms per step.  Between one and four rounds the smaller tile's finer quantisation wins.
// (deep K only -- 16 slabs, i.e. the 3 x 3 layers: gim_lightglue's 128- and 256-tile Linears, K = 256 / 512, measured 1.5 % slower on the big tile)
static bool big_tile_count(long long tiles, int nkt) { return tiles >= BIG_MIN_TILES || (tiles >= 128 && tiles <= 256 && nkt >= 16); }

template <int BM, int BN, int WM, int WN, bool BF16>
int dispatch_res(const gim_conv_args& a, hipStream_t s) {
    const bool obf = out_is16(a);
    if (a.res) {
        if (obf) return launch_persistent<BM, BN, WM, WN, BF16, true, true>(a, s);
        return launch_persistent<BM, BN, WM, WN, BF16, false, true>(a, s);
    }
    if (obf) return launch_persistent<BM, BN, WM, WN, BF16, true, false>(a, s);
    return launch_persistent<BM, BN, WM, WN, BF16, false, false>(a, s);
}

template <bool BF16>
int dispatch_persistent(const gim_conv_args& a, hipStream_t s) {
    if (a.npad % 128 == 0) {
        // deep-prefetch 256x128 kernel: bf16 output, enough K per tile to be MFMA-bound and enough tiles to fill
        // 256 one-per-CU workgroups a few times over
        const int es = BF16 ? 2 : 4;
        const int nkt = a.kpad * es / KTB;
        const long long M = (long long)a.B * a.Ho * a.Wo;
        const long long T = ((M + 255) / 256) * (a.npad / 128);
        // 256 x 256 tile, 8 waves, 64 x 128 wave tile: twice the MFMAs per wave and slab against nearly the same
        // staging / addressing overhead -- for the MFMA-bound layers (no residual, bf16 out, N % 256 == 0)
        if (a.npad % 256 == 0 && out_is16(a) && !a.res &&
            (a.use_lds_dma == 3 || (nkt >= BIG_MIN_NKT && big_tile_count(((M + 255) / 256) * (a.npad / 256), nkt))))   // 3: the tests' way onto this tile
        {
            // N <= 224 (the FPN's 196-channel layers): the second column half's last fragment is pure padding
            const bool skip = a.N <= a.npad - 32;
            if constexpr (BF16) {
                if (a.ups) return skip ? launch_persistent<256, 256, 4, 2, true, true, false, true, true>(a, s)
                                       : launch_persistent<256, 256, 4, 2, true, true, false, false, true>(a, s);
            }
            if (skip) return launch_persistent<256, 256, 4, 2, BF16, true, false, true>(a, s);
            return launch_persistent<256, 256, 4, 2, BF16, true, false>(a, s);
        }
        // fp32 operands multiplied as fp16 hi / lo pairs (split16): the 256 x 256 tile halves the splits and the staged bytes per MFMA
        if constexpr (!BF16) {
            if (a.split16 && a.npad % 256 == 0 && a.out_dtype == GIM_F32 && !a.res && !a.ups &&
                (a.use_lds_dma == 3 || (nkt >= 2 * BIG_MIN_NKT && big_tile_count(((M + 255) / 256) * (a.npad / 256), nkt))))
                return a.N <= a.npad - 32 ? launch_persistent<256, 256, 4, 2, false, false, false, true>(a, s)   // (196 channels: the all-padding fragment is skipped)
                                          : launch_persistent<256, 256, 4, 2, false, false, false>(a, s);
        }
        // (a 512 x 128 tile with 128 x 64 wave tiles for the N = 128 layers measured slower than 128 x 128: 875 vs 716 us
        //  on 196->128 3x3 -- not built)
        return dispatch_res<128, 128, 2, 2, BF16>(a, s);
    }
    // N = 129 ... 192 on the 256 x 64 tile reads every pixel panel three times (one tile per 64 channels); these layers are memory-bound
    // (DKM's 144-channel refiner blocks at 0.9 M pixels: 510 MB per launch at 2.8 TB/s).  A 256 x 192 tile on 8 waves of 32 px x 192 ch
    // stages the panel once.
    if constexpr (BF16) {
        if (a.npad == 192 && out_is16(a) && !a.res && !a.ups) return launch_persistent<256, 192, 8, 1, true, true, false>(a, s);
    }
    return dispatch_res<256, 64, 4, 1, BF16>(a, s);
}

template <int BM, int BN, int WM, int WN, bool BF16, bool LDSDMA>
int launch_igemm(const gim_conv_args& a, hipStream_t stream) {
    constexpr int stage = 2 * (BM + BN) * KTB, ctile = BM * (BN + 4) * 4;
    constexpr int smem = stage > ctile ? stage : ctile;
    auto kern = igemm_kernel<BM, BN, WM, WN, BF16, LDSDMA>;
    static GimPerDevice attr_done;
    if (attr_done.needed()) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) {
            gim_set_error("hipFuncSetAttribute(%d B LDS): %s", smem, hipGetErrorString(e));
            return GIM_ERR_LAUNCH;
        }
        attr_done.done();
    }
    const int M = a.B * a.Ho * a.Wo;
    const int mtiles = (M + BM - 1) / BM, ntiles = a.npad / BN;
    hipLaunchKernelGGL(kern, dim3((unsigned)(mtiles * ntiles)), dim3(256), smem, stream, a, mtiles, ntiles, M);
    return gim_check_launch("igemm_kernel");
}

template <bool BF16, bool LDSDMA>
int dispatch_tile(const gim_conv_args& a, hipStream_t s) {
    if (a.npad % 128 == 0) return launch_igemm<128, 128, 2, 2, BF16, LDSDMA>(a, s);
    return launch_igemm<256, 64, 4, 1, BF16, LDSDMA>(a, s);
}

}  // namespace

// a->ups is only built into the 256 x 256 / 8-wave bf16 tile: the launch must be one that dispatch_persistent sends there
static bool ups_supported(const gim_conv_args& a) {
    if (a.dtype != GIM_H16 || a.out_dtype != GIM_H16 || a.res || (a.use_lds_dma != 1 && a.use_lds_dma != 3) || a.npad % 256 != 0) return false;
    // output rows are (image, Y, X) with Y < 2 ups_h, X < 2 ups_w whatever geometry the launch states (a 1x1 conv is launched flat)
    const long long Mo = (long long)a.B * a.Ho * a.Wo;
    if (a.ups_h <= 0 || a.ups_w <= 0 || Mo % (4ll * a.ups_h * a.ups_w) != 0 || (2 * a.ups_w) % 32 != 0 || a.ups_ld % 8 != 0 || a.ups_ld < a.N) return false;
    if (a.act_cols != 0 || a.act != GIM_ACT_NONE) return false;   // the upsampled map is added in front of the activation slot: only the FPN's bare lateral conv
    const int nkt = a.kpad * 2 / KTB;
    const long long M = (long long)a.B * a.Ho * a.Wo;
    return a.use_lds_dma == 3 || (nkt >= BIG_MIN_NKT && big_tile_count(((M + 255) / 256) * (a.npad / 256), nkt));
}

#if !GIM_HALF_KIND
extern "C" int gim_conv_ups_supported_f16(const gim_conv_args* ap);
extern "C" int gim_conv2d_bn_act_f16(const gim_conv_args* ap, gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_conv_ups_supported)(const gim_conv_args* ap) {
#if !GIM_HALF_KIND
    if (ap && ap->dtype == GIM_F16) return gim_conv_ups_supported_f16(ap);
#endif
    return ap && ups_supported(*ap) ? 1 : 0;
}

extern "C" int GIM_FN(gim_conv2d_bn_act)(const gim_conv_args* ap, gim_stream_t stream) {
    GIM_REQUIRE(ap, "gim_conv2d_bn_act: NULL args");
#if !GIM_HALF_KIND
    // the fp16 objects of this file: fp16 operands, or fp32 operands with an fp16 output / residual (the dense matchers' fp32 GP products
    // and kernel matrices feeding 16-bit feature maps: round 5, their IEEE-fp16 flavour)
    if (ap->dtype == GIM_F16 || (ap->dtype == GIM_F32 && (ap->out_dtype == GIM_F16 || (ap->res && ap->res_dtype == GIM_F16))))
        return gim_conv2d_bn_act_f16(ap, stream);
    GIM_REQUIRE(ap->out_dtype != GIM_F16 && (!ap->res || ap->res_dtype != GIM_F16), "conv: bf16 operands cannot take an fp16 output / residual");
#endif
    const gim_conv_args& a = *ap;
#if GIM_HALF_KIND
    // the other 16-bit kind as OUTPUT only (the bf16 mode's stem: fp16 image in, bf16 feature map out): the persistent kernels'
    // epilogue packs either kind (out_is16 / Epilogue::run); everything else here works on the flavour's own type
    GIM_REQUIRE(a.out_dtype != GIM_BF16 || (a.dtype == GIM_H16 && !a.res && !a.ups && a.use_lds_dma == 1),
                "conv: fp16 operands with bf16 output: no residual, no upsample operand, LDS-DMA path only");
#endif
    const int es = a.dtype == GIM_H16 ? 2 : 4;
    GIM_REQUIRE(a.dtype == GIM_H16 || a.dtype == GIM_F32, "conv: bad dtype %d", a.dtype);
    GIM_REQUIRE(a.x && a.w && a.y && a.ktab, "conv: NULL x/w/y/ktab");
    GIM_REQUIRE(a.npad > 0 && a.npad % 64 == 0, "conv: npad=%d must be a multiple of 64", a.npad);
    GIM_REQUIRE(a.kpad > 0 && (a.kpad * es) % KTB == 0, "conv: kpad=%d is not a multiple of the %d-byte K slab", a.kpad, KTB);
    GIM_REQUIRE(a.N > 0 && a.N % 4 == 0 && a.N <= a.npad, "conv: N=%d must be a multiple of 4 and <= npad=%d", a.N, a.npad);
    GIM_REQUIRE(a.x_bytes > 0 && a.x_bytes < (int64_t)0xFFFFFFF0ll, "conv: x_bytes=%lld must be < 4 GiB", (long long)a.x_bytes);
    GIM_REQUIRE(a.ldx % (16 / es) == 0, "conv: ldx=%d breaks 16-byte alignment", a.ldx);
    GIM_REQUIRE(a.ldy % 4 == 0 && (!a.res || a.ldres % 4 == 0), "conv: ldy/ldres must be multiples of 4");
    GIM_REQUIRE(!out_is16(a) || (a.N % 8 == 0 && a.ldy % 8 == 0), "conv: 16-bit output needs N and ldy multiples of 8 (16-byte row stores)");
    GIM_REQUIRE(!a.res || (a.res_dtype == a.out_dtype && (a.res_dtype != GIM_H16 || a.ldres % 8 == 0)), "conv: residual must have the output dtype (and ldres %% 8 == 0 for 16-bit rows)");
    GIM_REQUIRE(a.act_cols >= 0 && a.act_cols % 128 == 0, "conv: act_cols=%d must be a multiple of 128", a.act_cols);
    GIM_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0 && a.Ho > 0 && a.Wo > 0 && a.stride > 0, "conv: bad geometry");
    GIM_REQUIRE((int64_t)a.B * a.Ho * a.Wo < (int64_t)0x7fffffff, "conv: too many output rows");
    GIM_REQUIRE(!a.ups || ups_supported(a), "conv: this launch cannot take the fused upsample-add (see gim_conv_ups_supported)");
    hipStream_t s = (hipStream_t)stream;
    if (a.use_lds_dma == 2) {  // 3x3 halo kernel: w / ktab / kpad describe the halo packing (gim_amd/packing.py::pack_halo)
        GIM_REQUIRE(a.dtype == GIM_H16 && a.out_dtype == GIM_H16 && !a.res && a.stride == 1 && a.pad == 1 && a.H == a.Ho && a.W == a.Wo,
                    "conv3x3 halo: 16-bit in / out, stride 1, pad 1, no residual");
        GIM_REQUIRE(a.npad % 128 == 0 && a.kpad % 64 == 0 && a.res_mod > 0 && a.res_mod <= a.ldx && a.act_cols == 0, "conv3x3 halo: bad packing");
        return a.npad % 256 == 0 ? launch_halo<4>(a, s) : launch_halo<2>(a, s);
    }
    if (a.dtype == GIM_H16) return a.use_lds_dma ? dispatch_persistent<true>(a, s) : dispatch_tile<true, false>(a, s);
    return a.use_lds_dma ? dispatch_persistent<false>(a, s) : dispatch_tile<false, false>(a, s);
}
