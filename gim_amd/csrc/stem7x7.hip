// The first convolution of the ResNet-FPN backbone (networks/loftr/backbone/resnet.py:306: conv1 7x7 / stride 2 / pad 3, 3 -> 64,
// + bn1 + relu) as its own kernel for gfx950: round 4.
//
// Through the implicit-GEMM kernel this layer stages 49 taps x 16 B (32 B with split operands) per OUTPUT pixel and K slab after K
// slab of a [64 x 392 / 784] filter bank for 23 GFLOP of work: 0.12 ms per batch-8 step, 0.25 ms on split operands -- 14 x more
// staged bytes than the layer has input (DESIGN "Round 4: what the counters said").  Here
//   * a persistent 8-wave workgroup (one per CU) keeps the WHOLE filter bank in LDS (98 KiB, loaded once) and walks 4 x 32-pixel
//     output tiles; the (2 * 4 + 5) x (2 * 32 + 5) = 13 x 69 input pixels a tile touches are staged ONCE (14 KiB instead of 196 KiB of
//     per-tap gathers), double-buffered by LDS-DMA so that tile t + 1 travels under the MFMAs of tile t;
//   * wave w computes output row w >> 1, channels 32 (w & 1) ... + 31: one 32 x 32 accumulator, one v_mfma_f32_32x32x16 per
//     "virtual tap" -- weights = A operand (lane = output channel), pixels = B operand (lane = output x, the tap only shifts the
//     LDS address), so the accumulator holds 4 consecutive channels of one pixel per quad as everywhere in this library;
//   * a pixel is ONE 16-byte piece in both operand modes.  Plain: [r g b 0 0 0 0 0]; the two K halves of an MFMA are two different
//     taps (25 MFMAs).  Split (the default, packing.pack_stem7x7): [r_hi g_hi b_hi r_lo g_lo b_lo 0 0]; both K halves read the SAME
//     piece, against [w_hi w_hi 0 0] and [w_lo 0 0 0 0 0]: x_hi w_hi + x_lo w_hi + x_hi w_lo in 49 MFMAs -- the image tensor and the
//     staging are no wider than the plain stem's (the implicit-GEMM form needs 16 channels per pixel for the same sum).
// LDS images are XOR-swizzled on the fill side (the filter bank by the host packer, the patch on the DMA source address) so that
// every ds_read_b128 of a fragment is conflict-free: pixels of a fragment are 32 B apart, filter rows 32 B apart.
#include "gim_common.h"

namespace {

constexpr int TH = 4, TW = 32;                       // output tile
constexpr int PR = 2 * TH + 5, PC = 2 * TW + 5;      // input patch: 13 x 69 pixels
constexpr int NPX = PR * PC;                          // 897
constexpr int PATCH_SLOTS = 960;                      // 15 LDS-DMA instructions of 64 pixels
constexpr int PATCH_BYTES = PATCH_SLOTS * 16;
constexpr int OSTAGE = 32 * 64;                       // per-wave output transposition: 32 pixels x 32 channels x 2 B

template <bool SPLIT> struct StemCfg {
    static constexpr int NVT = SPLIT ? 49 : 25;       // MFMAs per accumulator ("virtual taps")
    static constexpr int W_BYTES = NVT * 64 * 32;     // [vt][64 output channels][2 K halves x 16 B]
    static constexpr int OFF_PATCH = W_BYTES, OFF_OUT = OFF_PATCH + 2 * PATCH_BYTES, SMEM = OFF_OUT + 8 * OSTAGE;
};
static_assert(StemCfg<true>::SMEM <= 160 * 1024, "stem7x7: LDS");

struct Args {
    const void* x;          // [B][H][W][8] 16-bit pixels (one 16-byte piece each)
    const void* w;          // StemCfg::W_BYTES, LDS image (packing.pack_stem7x7)
    const float* bias;      // [64]
    void* y;                // [B][Ho][Wo][64] 16-bit
    int B, H, W, Ho, Wo;
    int out_bf;             // output kind: 1 = bf16, 0 = fp16 (the operands are the translation unit's own 16-bit kind)
    unsigned x_bytes;
};

typedef __attribute__((address_space(3))) void lds_t;

template <bool SPLIT>
__global__ void __launch_bounds__(512) stem7x7_kernel(const Args a) {
    typedef StemCfg<SPLIT> C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, lh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned smem_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_t*)smem);
    const gim_u32x4_t rw = gim_make_rsrc(a.w, (unsigned)C::W_BYTES), rx = gim_make_rsrc(a.x, a.x_bytes);
    const int ntx = (a.Wo + TW - 1) / TW, nty = (a.Ho + TH - 1) / TH;
    const int ntiles = a.B * nty * ntx;
    // ---- filter bank -> LDS, once (a linear copy: the host packed the swizzled image) ----
    for (int p = w; p < C::W_BYTES / 1024; p += 8) gim_dma16(rw, smem_addr + (unsigned)(p * 1024), (unsigned)(p * 1024 + lane * 16));
    // ---- input patch of tile `tile` -> patch buffer `buf`: LDS slot s = pr * 69 + pc' holds pixel (pr, pc' ^ ((pc' >> 4) & 1)) of the
    // patch = image pixel (2 ty0 - 3 + pr, 2 tx0 - 3 + pc); outside the image (and beyond the 897 slots) the offset lies beyond the
    // descriptor's bound and reads as zeros: the convolution's padding
    auto issue_patch = [&](const int tile, const int buf) __attribute__((always_inline)) {
        const int b = tile / (nty * ntx), r = tile - b * nty * ntx, ty = r / ntx, tx = r - ty * ntx;
        const int iy0 = 2 * ty * TH - 3, ix0 = 2 * tx * TW - 3;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int q = w + 8 * k;                   // wave-uniform DMA instruction index: 15 of them
            if (q < PATCH_SLOTS / 64) {
                const int s = q * 64 + lane, pr = s / PC, pcs = s - pr * PC;
                const int pc = pcs ^ ((pcs >> 4) & 1);
                const int iy = iy0 + pr, ix = ix0 + pc;
                const bool ok = s < NPX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const unsigned voff = ok ? (unsigned)(((size_t)b * a.H + iy) * a.W + ix) * 16u : a.x_bytes;
                gim_dma16(rx, smem_addr + (unsigned)(C::OFF_PATCH + buf * PATCH_BYTES + q * 1024), voff);
            }
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) issue_patch(tile, 0);
    // this wave's role: output row `orow` of the tile, channels 32 f ... 32 f + 31
    const int orow = w >> 1, f = w & 1;
    // filter fragment of virtual tap vt: row 32 f + l31, K half lh at half slot lh ^ ((row >> 3) & 1)
    const unsigned wfrag = (unsigned)(((32 * f + l31) * 2 + (lh ^ ((l31 >> 3) & 1))) * 16);
    float bq[16];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const float4 b4 = *(const float4*)(a.bias + 32 * f + 8 * rg + 4 * lh);
        bq[4 * rg] = b4.x; bq[4 * rg + 1] = b4.y; bq[4 * rg + 2] = b4.z; bq[4 * rg + 3] = b4.w;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the bias loads have landed AND the compiler knows it (no per-iteration wait that would
                                          // also drain the patch DMA issued inside the loop, see DESIGN "hipcc's wait bookkeeping")
    char* ost = smem + C::OFF_OUT + w * OSTAGE;
    int buf = 0;
    for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // this tile's patch (and, the first time, the filter bank) has landed; the other buffer is free
        const int next = tile + gridDim.x;
        if (next < ntiles) issue_patch(next, buf ^ 1);
        const char* pb = smem + C::OFF_PATCH + buf * PATCH_BYTES;
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bq[r];
#pragma unroll
        for (int vt = 0; vt < C::NVT; ++vt) {
            int tap = SPLIT ? vt : 2 * vt + lh;        // plain: the two K halves are two taps (tap 49 does not exist: zero filter, any pixel)
            tap = tap > 48 ? 48 : tap;
            const int dy = tap / 7, dx = tap - 7 * dy;
            const int pr = 2 * orow + dy, pc = 2 * l31 + dx;
            const int slot = pr * PC + (pc ^ ((pc >> 4) & 1));
            const bf16x8_t px = *(const bf16x8_t*)(pb + slot * 16);
            const bf16x8_t wv = *(const bf16x8_t*)(smem + vt * 2048 + wfrag);
            acc = mfma_h16_32x32x16(wv, px, acc);
        }
        // ---- relu, 16-bit, through the wave's staging buffer into 16-byte row pieces ----
        const int b = tile / (nty * ntx), r = tile - b * nty * ntx, ty = r / ntx, tx = r - ty * ntx;
        const int oy = ty * TH + orow;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const unsigned u0 = cvt_pk_16(fmaxf(acc[4 * rg], 0.f), fmaxf(acc[4 * rg + 1], 0.f), a.out_bf != 0);
            const unsigned u1 = cvt_pk_16(fmaxf(acc[4 * rg + 2], 0.f), fmaxf(acc[4 * rg + 3], 0.f), a.out_bf != 0);
            // pixel l31, channels 8 rg + 4 lh ... + 3 of this fragment: 16-byte piece rg (XOR-swizzled with the pixel), half lh
            *(uint2*)(ost + l31 * 64 + ((rg ^ (l31 & 3)) << 4) + lh * 8) = make_uint2(u0, u1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = k * 64 + lane, px = i >> 2, pc = i & 3;
            const uint4 v = *(const uint4*)(ost + px * 64 + ((pc ^ (px & 3)) << 4));
            const int ox = tx * TW + px;
            if (oy < a.Ho && ox < a.Wo)
                *(uint4*)((char*)a.y + ((((size_t)b * a.Ho + oy) * a.Wo + ox) * 64 + 32 * f + 8 * pc) * 2) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the staging buffer is rewritten by the next tile
    }
}

}  // namespace

extern "C" int64_t GIM_FN(gim_stem7x7_weight_bytes)(int split) { return split ? StemCfg<true>::W_BYTES : StemCfg<false>::W_BYTES; }

#if !GIM_HALF_KIND
extern "C" int gim_stem7x7_f16(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int split, int dtype, int out_dtype, gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_stem7x7)(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int split, int dtype,
                                   int out_dtype, gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16) return gim_stem7x7_f16(x, w, bias, y, B, H, W, split, dtype, out_dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(x && w && bias && y && B > 0 && H > 0 && W > 0, "stem7x7: bad args");
    GIM_REQUIRE(dtype == GIM_H16 && (out_dtype == GIM_BF16 || out_dtype == GIM_F16), "stem7x7: 16-bit operands and output only (dtype %d -> %d)", dtype, out_dtype);
    GIM_REQUIRE((int64_t)B * H * W * 16 < (int64_t)0xFFFFFFF0ll, "stem7x7: image batch too large for 32-bit buffer offsets");
    Args a;
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.B = B; a.H = H; a.W = W;
    a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;     // (H + 2 * 3 - 7) / 2 + 1
    a.out_bf = out_dtype == GIM_BF16; a.x_bytes = (unsigned)((size_t)B * H * W * 16);
    static GimPerDevice attr;
    if (attr.needed()) {
        hipError_t e = hipFuncSetAttribute((const void*)stem7x7_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, StemCfg<true>::SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)stem7x7_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, StemCfg<false>::SMEM);
        if (e != hipSuccess) { gim_set_error("stem7x7: hipFuncSetAttribute: %s", hipGetErrorString(e)); return GIM_ERR_LAUNCH; }
        attr.done();
    }
    int ncu = 256;
    { int dev = 0; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev); }
    const int ntiles = B * ((a.Ho + TH - 1) / TH) * ((a.Wo + TW - 1) / TW);
    const unsigned grid = (unsigned)(ntiles < ncu ? ntiles : ncu);
    if (split) hipLaunchKernelGGL(stem7x7_kernel<true>, dim3(grid), dim3(512), StemCfg<true>::SMEM, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(stem7x7_kernel<false>, dim3(grid), dim3(512), StemCfg<false>::SMEM, (hipStream_t)stream, a);
    return gim_check_launch("stem7x7");
}
