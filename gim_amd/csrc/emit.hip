// Output hand-out of the gim_loftr forward (gfx950): the small device-to-device moves that follow coarse matching.
//
// A HIP-graph replay re-uses its output buffers, so every forward hands out private copies of the M-row match lists
// (b_ids, i_ids, j_ids, m_bids, mkpts0_c, mkpts1_c, mconf) plus an all-false gt_mask: eight torch copy / fill kernels
// (clone, zeros) per forward before -- one launch here.  `gim_pack_matches` is the reporting row
// [pair_id, x0, y0, x1, y1, conf] of gim_amd/runner.py (was: index + 4-way torch.cat + a pageable host-to-device copy).
//
// Replaces (reference file:line): the tensor construction at networks/loftr/utils/coarse_matching.py:236-259 as far as it
// only moves data, and the per-pair metric rows of trainer/lightning.py:258-270 (packed form).
#include "gim_common.h"

namespace {

struct Segs { gim_copy_segs s; };

// grid.y = segment; 16-byte lanes where source, destination and length allow, bytes otherwise.  src == NULL: zero fill.
__global__ void __launch_bounds__(256) copy_segments_kernel(const Segs a) {
    const int k = blockIdx.y;
    const char* src = (const char*)a.s.src[k];
    char* dst = (char*)a.s.dst[k];
    const int64_t n = a.s.bytes[k];
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
    const bool vec = (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
    const int64_t nv = vec ? n >> 4 : 0;
    for (int64_t i = t; i < nv; i += nt) ((uint4*)dst)[i] = src ? ((const uint4*)src)[i] : make_uint4(0u, 0u, 0u, 0u);
    for (int64_t i = (nv << 4) + t; i < n; i += nt) dst[i] = src ? src[i] : (char)0;
}

__global__ void __launch_bounds__(256) pack_matches_kernel(const int64_t* __restrict__ m_bids, const float2* __restrict__ mk0,
                                                           const float2* __restrict__ mk1, const float* __restrict__ conf,
                                                           const int64_t* __restrict__ pair_ids, int64_t pid_base,
                                                           float* __restrict__ out, int M) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int64_t b = m_bids[m];
    const float pid = (float)(pair_ids ? pair_ids[b] : pid_base + b);
    const float2 p0 = mk0[m], p1 = mk1[m];
    float* o = out + (size_t)m * 6;   // 24-byte rows: three 8-byte stores
    *(float2*)(o + 0) = make_float2(pid, p0.x);
    *(float2*)(o + 2) = make_float2(p0.y, p1.x);
    *(float2*)(o + 4) = make_float2(p1.y, conf[m]);
}

}  // namespace

extern "C" int gim_copy_segments(const gim_copy_segs* sp, gim_stream_t stream) {
    GIM_REQUIRE(sp, "gim_copy_segments: NULL args");
    GIM_REQUIRE(sp->n >= 0 && sp->n <= GIM_MAX_COPY_SEGS, "gim_copy_segments: n=%d outside [0, %d]", sp->n, GIM_MAX_COPY_SEGS);
    int64_t mx = 0;
    for (int k = 0; k < sp->n; ++k) {
        GIM_REQUIRE(sp->bytes[k] >= 0 && (sp->bytes[k] == 0 || sp->dst[k]), "gim_copy_segments: segment %d: bad size / NULL dst", k);
        mx = sp->bytes[k] > mx ? sp->bytes[k] : mx;
    }
    if (sp->n == 0 || mx == 0) return GIM_OK;
    Segs a;
    a.s = *sp;
    int64_t blocks = (mx / 16 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    hipLaunchKernelGGL(copy_segments_kernel, dim3((unsigned)blocks, (unsigned)sp->n), dim3(256), 0, (hipStream_t)stream, a);
    return gim_check_launch("copy_segments_kernel");
}

extern "C" int gim_pack_matches(const int64_t* m_bids, const float* mkpts0, const float* mkpts1, const float* mconf,
                                const int64_t* pair_ids, int64_t pid_base, float* out, int M, gim_stream_t stream) {
    GIM_REQUIRE(M >= 0, "gim_pack_matches: M=%d", M);
    if (M == 0) return GIM_OK;
    GIM_REQUIRE(m_bids && mkpts0 && mkpts1 && mconf && out, "gim_pack_matches: NULL pointer");
    hipLaunchKernelGGL(pack_matches_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, m_bids,
                       (const float2*)mkpts0, (const float2*)mkpts1, mconf, pair_ids, pid_base, out, M);
    return gim_check_launch("pack_matches_kernel");
}
