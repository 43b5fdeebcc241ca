"""Thin torch-tensor wrappers over the libgimhip C ABI (device pointers + current stream; torch is only
the allocator / stream owner).  Every function launches HIP kernels from `gim_amd/csrc`; none of them
has a torch or CPU fallback."""
import ctypes
import os

import torch

from . import _lib
from ._lib import ACT_ELU1, ACT_GELU, ACT_LEAKY, ACT_NONE, ACT_RELU, GIM_BF16, GIM_F16, GIM_F32, check, lib  # noqa: F401
from .packing import elem_size, torch_dtype
from .switches import flag

_DT = {torch.float32: GIM_F32, torch.bfloat16: GIM_BF16, torch.float16: GIM_F16}
HALF = (torch.bfloat16, torch.float16)   # the two 16-bit operand kinds: same kernels in two flavours (csrc/gim_common.h)


# When set to a list, every gim_conv2d_bn_act launch is bracketed by HIP events recorded on the launch
# stream and (start, end, algorithmic_flops, label) is appended -- bench.py's live roofline measurement.
PROFILE = None
PROFILE_FUSED = None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Timed:
    """bench.py's live measurement of the fused kernels: HIP events on the launch stream around one C-ABI call, appended to
    PROFILE_FUSED as (start, end, algorithmic_flops, kernel family) when that list is set."""

    def __init__(self, family, flops):
        self.family, self.flops = family, flops

    def __enter__(self):
        if PROFILE_FUSED is not None:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if PROFILE_FUSED is not None:
            self.e1.record()
            PROFILE_FUSED.append((self.e0, self.e1, self.flops, self.family))
        return False


def patch_last_fused_flops(family, flops):
    """bench.py's live measurement: a launch that did not know its work when it was enqueued (fine_fused with a device-side count)"""
    if PROFILE_FUSED:
        for i in range(len(PROFILE_FUSED) - 1, -1, -1):
            if PROFILE_FUSED[i][3] == family:
                e0, e1, _, nm = PROFILE_FUSED[i]
                PROFILE_FUSED[i] = (e0, e1, flops, nm)
                return


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _req_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.GimHipError("gim_amd ops need device (cuda/HIP) tensors: the product path has no CPU fallback")


def gim_dtype(t):
    return _DT[t.dtype]


def _rows(t):
    """[..., C] tensor with a uniform row stride -> (rows, ld)"""
    assert t.stride(-1) == 1
    if t.dim() == 2:
        return t.shape[0], t.stride(0)
    assert t.is_contiguous()
    return t.numel() // t.shape[-1], t.shape[-1]


# ---- layout ------------------------------------------------------------------------------------
def nchw_to_nhwc(src, dst, b_off=0):
    """src [B,C,H,W] fp32 -> dst [Btot,H,W,cpad] (fp32/bf16), images b_off.."""
    _req_cuda(src, dst)
    assert src.dtype == torch.float32 and src.is_contiguous() and dst.is_contiguous()
    B, C, H, W = src.shape
    check(lib.gim_nchw_to_nhwc(_p(src), _p(dst), B, C, H, W, dst.shape[-1], dst.shape[-1], b_off,
                               gim_dtype(dst), _stream()), "gim_nchw_to_nhwc")


def _health(word):
    """the `health` argument of the kernels that store an un-normalised residual stream (include/gim_hip.h, "fp16 range guard"): an
    int32 device tensor (one element) they OR 4 into when an fp16 value they store left the IEEE range, or None"""
    if word is not None:
        _req_cuda(word)
        assert word.dtype == torch.int32 and word.numel() >= 1
    return _p(word)


def nchw_to_nhwc_split(src, dst, b_off=0):
    """src [B,C,H,W] fp32 -> dst [Btot,H,W,ld] 16-bit with channels [hi(C) | lo(C) | hi(C) | 0...] when ld >= 3 C, [hi(C) | lo(C) |
    0...] when 2 C <= ld < 3 C (gim_nchw_to_nhwc_split)"""
    _req_cuda(src, dst)
    assert src.dtype == torch.float32 and src.is_contiguous() and dst.is_contiguous() and dst.dtype in HALF
    B, C, H, W = src.shape
    check(lib.gim_nchw_to_nhwc_split(_p(src), _p(dst), B, C, H, W, dst.shape[-1], b_off, gim_dtype(dst), _stream()),
          "gim_nchw_to_nhwc_split")


def stem7x7(x, ps, out_dtype=None):
    """x [B,H,W,8] 16-bit pixels ([r g b 0...] or, ps.split, [hi(3) lo(3) 0 0]) -> relu(bn1(conv1(x))) [B,Ho,Wo,64] (gim_stem7x7;
    ps = packing.pack_stem7x7)"""
    _req_cuda(x, ps.w, ps.bias)
    assert x.dtype in HALF and x.dtype == ps.w.dtype and x.is_contiguous() and x.shape[-1] == 8
    assert ps.w.numel() * 2 == lib.gim_stem7x7_weight_bytes(int(ps.split))
    B, H, W, _ = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(B, Ho, Wo, 64, dtype=out_dtype or x.dtype, device=x.device)
    with _Timed("stem7x7", 2.0 * B * Ho * Wo * 64 * 49 * ps.cin):   # the layer's algorithmic flops
        check(lib.gim_stem7x7(_p(x), _p(ps.w), _p(ps.bias), _p(y), B, H, W, int(ps.split), gim_dtype(x), gim_dtype(y), _stream()),
              "gim_stem7x7")
    return y


def copy_segments(pairs):
    """[(src or None, dst), ...] contiguous device tensors of equal byte size per pair (src None: dst is zero filled) -> all of
    them in ONE launch (gim_copy_segments); more than 12 pairs go out in batches."""
    pairs = [(s, d) for s, d in pairs if d.numel() > 0]
    for i in range(0, len(pairs), _lib.CopySegs.MAX):
        seg = _lib.CopySegs()
        chunk = pairs[i:i + _lib.CopySegs.MAX]
        for k, (s, d) in enumerate(chunk):
            _req_cuda(s, d)
            assert d.is_contiguous() and (s is None or (s.is_contiguous() and s.numel() * s.element_size() == d.numel() * d.element_size()))
            seg.src[k] = s.data_ptr() if s is not None else None
            seg.dst[k] = d.data_ptr()
            seg.bytes[k] = d.numel() * d.element_size()
        seg.n = len(chunk)
        check(lib.gim_copy_segments(ctypes.byref(seg), _stream()), "gim_copy_segments")


def pack_matches(m_bids, mkpts0, mkpts1, mconf, pair_ids=None, pid_base=0):
    """rows [pair_id, x0, y0, x1, y1, conf] fp32 [M, 6]; pair_id = pair_ids[m_bids] (device int64 tensor) or pid_base + m_bids"""
    _req_cuda(m_bids, mkpts0, mkpts1, mconf, pair_ids)
    M = m_bids.numel()
    assert m_bids.dtype == torch.int64 and mkpts0.dtype == mkpts1.dtype == mconf.dtype == torch.float32
    assert mkpts0.shape == (M, 2) and mkpts1.shape == (M, 2) and mconf.shape == (M,)
    assert pair_ids is None or pair_ids.dtype == torch.int64
    out = torch.empty(M, 6, dtype=torch.float32, device=m_bids.device)
    check(lib.gim_pack_matches(_p(m_bids.contiguous()), _p(mkpts0.contiguous()), _p(mkpts1.contiguous()), _p(mconf.contiguous()),
                               _p(pair_ids), int(pid_base), _p(out), M, _stream()), "gim_pack_matches")
    return out


def nhwc_to_nchw(src, C):
    """src [B,H,W,cstore] -> new fp32 [B,C,H,W] (reference layout, for tests / lazy outputs)"""
    _req_cuda(src)
    B, H, W, ld = src.shape
    out = torch.empty(B, C, H, W, dtype=torch.float32, device=src.device)
    check(lib.gim_nhwc_to_nchw(_p(src), _p(out), B, C, H, W, ld, gim_dtype(src), _stream()), "gim_nhwc_to_nchw")
    return out


# ---- conv / linear --------------------------------------------------------------------------------
def conv_rows(x, pk, geom, y, act=ACT_NONE, res=None, res_mod=0, lds_dma=True, act_cols=0, ups=None, health=None):
    """Generic launch.  x: 2-D row view [rows_in, ldx]; geom = (B, H, W, Ho, Wo); y: 2-D row view.  ups: half-resolution NHWC
    tensor whose bilinear x2 upsampling is added in the epilogue, or a callable that performs that addition as a separate pass when
    the launch cannot take it (returns True when it was fused)."""
    _req_cuda(x, y, res)
    B, H, W, Ho, Wo = geom
    a = _lib.ConvArgs()
    a.x, a.w, a.ktab = x.data_ptr(), pk.w.data_ptr(), pk.ktab.data_ptr()
    a.bias = pk.bias.data_ptr() if pk.bias is not None else None
    a.res = res.data_ptr() if res is not None else None
    a.y = y.data_ptr()
    es = elem_size(pk.dtype)
    assert x.dtype == torch_dtype(pk.dtype), (x.dtype, pk.dtype)
    a.x_bytes = ((B * H * W - 1) * x.stride(0) + pk.cin_pad) * es
    a.B, a.H, a.W, a.Ho, a.Wo = B, H, W, Ho, Wo
    a.stride, a.pad = pk.stride, pk.pad
    a.ldx, a.ldy = x.stride(0), y.stride(0)
    a.ldres = res.stride(0) if res is not None else 0
    a.N, a.npad, a.kpad = pk.n_store, pk.npad, pk.kpad
    a.act, a.res_mod, a.act_cols = act, res_mod, act_cols
    a.dtype, a.out_dtype = pk.dtype, gim_dtype(y)
    a.res_dtype = gim_dtype(res) if res is not None else GIM_F32
    a.use_lds_dma = (3 if FORCE_BIG_TILE else 1) if lds_dma else 0
    a.health = _health(health).value if (health is not None and res is not None) else None
    a.split16 = 1 if (FP32_SPLIT and pk.dtype == GIM_F32 and lds_dma) else 0
    assert y.shape[0] >= B * Ho * Wo and y.shape[1] >= pk.n_store
    if ups is not None:
        a.ups, a.ups_h, a.ups_w, a.ups_ld = ups.data_ptr(), ups.shape[1], ups.shape[2], ups.shape[3]
        if not (UPS_FUSED and ups.dtype in HALF and ups.dtype == y.dtype and lib.gim_conv_ups_supported(ctypes.byref(a))):
            a.ups = None
    fused_ups = ups is not None and a.ups is not None
    if PROFILE is None:
        check(lib.gim_conv2d_bn_act(ctypes.byref(a), _stream()), "gim_conv2d_bn_act")
        return fused_ups
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.gim_conv2d_bn_act(ctypes.byref(a), _stream()), "gim_conv2d_bn_act")
    e1.record()
    flops = 2.0 * B * Ho * Wo * pk.cout * pk.cin * pk.kh * pk.kw  # algorithmic: real channels, no padding
    PROFILE.append((e0, e1, flops, f"{pk.cin}->{pk.cout} k{pk.kh}s{pk.stride} M={B * Ho * Wo}" + (" +ups" if fused_ups else "")))
    return fused_ups


FORCE_BIG_TILE = flag("force_big_tile", False)   # tests: the 256 x 256 tile on every eligible launch, whatever its size (gim_conv_args.use_lds_dma = 3)
# fp32 operands multiplied as IEEE-fp16 hi / lo pairs on the 16-bit MFMA (gim_conv_args.split16: three products per 16 K instead of eight fp32 MFMAs, 2^-22 per
# product); module state read at every launch -- gim_loftr's fp32 mode switches it on around its forward (LoFTR.fp32_split)
FP32_SPLIT = flag("fp32_split_all", False)
UPS_FUSED = flag("ups_fused", True)   # FPN: bilinear x2 + add inside the lateral 1x1 conv's epilogue (False: a second pass over the output)


def conv2d(x, pk, act=ACT_NONE, res=None, out_dtype=None, lds_dma=True, ups=None, health=None):
    """x [B,H,W,cin_pad] NHWC -> new [B,Ho,Wo,n_store].  ups: [B,Ho/2,Wo/2,n_store] -> y += bilinear_x2(ups) (align_corners=True):
    in the conv's epilogue when the launch supports it, otherwise as a second pass (gim_upsample2x_add)."""
    B, H, W, cs = x.shape
    assert cs == pk.cin_pad, (cs, pk)
    Ho = (H + 2 * pk.pad - pk.kh) // pk.stride + 1
    Wo = (W + 2 * pk.pad - pk.kw) // pk.stride + 1
    y = torch.empty(B, Ho, Wo, pk.n_store, dtype=out_dtype or x.dtype, device=x.device)
    r = res.view(-1, res.shape[-1]) if res is not None else None
    geom = (B, H, W, Ho, Wo)
    if pk.kh == 1 and pk.kw == 1 and pk.stride == 1 and pk.pad == 0:
        geom = (1, 1, B * H * W, 1, B * H * W)  # pixel index == row index: the kernel skips the coordinate decode
    if HALO and pk.halo is not None and res is None and lds_dma and x.dtype in HALF and y.dtype == x.dtype \
            and x.is_contiguous() and _halo_pays(pk, B, H, W):
        conv3x3_halo(x, pk, y, act)
        return y
    if ups is not None:
        assert ups.shape == (B, Ho // 2, Wo // 2, pk.n_store) and res is None and ups.is_contiguous()
        if not conv_rows(x.view(-1, cs), pk, geom, y.view(-1, pk.n_store), act, r, 0, lds_dma, ups=ups):
            upsample2x_add(ups, y)
        return y
    conv_rows(x.view(-1, cs), pk, geom, y.view(-1, pk.n_store), act, r, 0, lds_dma, health=health)
    return y


# 3x3 halo kernel (conv_igemm.hip: conv3x3_halo_kernel); module attributes, tests flip them (switches: conv_halo, conv_halo_min_tiles)
HALO = flag("conv_halo", True)
HALO_MIN_TILES = flag("conv_halo_min_tiles", 2000)   # round 3 sweep: below ~2000 tiles (the 1/4-resolution layers) the generic kernel is 5-8 % faster


def _halo_pays(pk, B, H, W):
    """whole 8 x 32 patches, enough of them to fill 256 one-per-CU workgroups twice, and no all-padding 32-channel fragment in
    the tile (N = 196 in a 256-wide tile: the generic kernel skips that fragment, this one does not) -- measured +4..5 % on
    the layers that qualify, -2 % on the 196-channel ones"""
    if H % 8 or W % 32:
        return False
    npad = pk.halo[0].shape[0]
    bn = 256 if npad % 256 == 0 else 128
    if pk.n_store <= npad - 32:     # incl. the 64-channel layers (SuperPoint / VGG19 / ResNet stems) in the 128-wide tile: half of it would be
        return False                # padding -- the generic 256 x 64 tile is 2x faster there (profiles/r03_lightglue_kernel_stats.txt: 552 vs 280 us)
    return B * (H // 8) * (W // 32) * (npad // bn) >= HALO_MIN_TILES


def conv3x3_halo(x, pk, y, act=ACT_NONE):
    """x [B,H,W,cin_pad] bf16 -> y [B,H,W,n_store] bf16 through the halo-tile kernel (3x3, stride 1, pad 1, no residual)"""
    _req_cuda(x, y)
    assert x.is_contiguous() and y.is_contiguous(), "conv3x3_halo addresses pixels as (b*H + y)*W + x rows of width cin_pad"
    B, H, W, cs = x.shape
    w, tab, nslab, bias = pk.halo
    a = _lib.ConvArgs()
    a.x, a.w, a.ktab = x.data_ptr(), w.data_ptr(), tab.data_ptr()
    a.bias = bias.data_ptr() if bias is not None else None
    a.res, a.y = None, y.data_ptr()
    a.x_bytes = ((B * H * W - 1) * cs + pk.cin_pad) * 2
    a.B, a.H, a.W, a.Ho, a.Wo = B, H, W, H, W
    a.stride, a.pad = 1, 1
    a.ldx, a.ldy, a.ldres = cs, y.shape[-1], 0
    a.N, a.npad, a.kpad = pk.n_store, w.shape[0], nslab * 64
    a.act, a.res_mod, a.act_cols = act, pk.cin_pad, 0
    a.dtype = a.out_dtype = gim_dtype(x)
    a.res_dtype = GIM_F32
    a.use_lds_dma = 2
    if PROFILE is None:
        check(lib.gim_conv2d_bn_act(ctypes.byref(a), _stream()), "gim_conv2d_bn_act(halo)")
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.gim_conv2d_bn_act(ctypes.byref(a), _stream()), "gim_conv2d_bn_act(halo)")
    e1.record()
    PROFILE.append((e0, e1, 2.0 * B * H * W * pk.cout * pk.cin * 9, f"{pk.cin}->{pk.cout} k3s1 M={B * H * W} halo"))


def linear(x, pk, y, act=ACT_NONE, lds_dma=True, act_cols=0):
    """x: row view [rows, >=K] (row stride may exceed K), y: row view [rows, >=N].  y = act(x @ W^T);
    act_cols > 0 restricts the activation to output columns < act_cols."""
    rows = x.shape[0]
    conv_rows(x, pk, (1, 1, rows, 1, rows), y, act, None, 0, lds_dma, act_cols)


# ---- elementwise ------------------------------------------------------------------------------------
def upsample2x_add(x, y):
    """y [B,2h,2w,C] += bilinear2x(x [B,h,w,C]) (align_corners=True), in place"""
    _req_cuda(x, y)
    B, h, w, C = x.shape
    assert y.shape == (B, 2 * h, 2 * w, C) and x.dtype == y.dtype
    check(lib.gim_upsample2x_add(_p(x), _p(y), B, h, w, C, C, C, gim_dtype(x), _stream()), "gim_upsample2x_add")


def posenc_add(x, pe, out_f32, out_t):
    """x rows [R, C] (dtype T); pe [hw, C] fp32; out_f32 row view and/or out_t row view"""
    _req_cuda(x, pe, out_f32, out_t)
    R, C = x.shape
    check(lib.gim_posenc_add(_p(x), _p(pe), _p(out_f32), _p(out_t), R, pe.shape[0], C, x.stride(0),
                             out_f32.stride(0) if out_f32 is not None else 4,
                             out_t.stride(0) if out_t is not None else 4, gim_dtype(x), _stream()), "gim_posenc_add")


def layernorm_residual(x, gamma, beta, res, out_f32, out_t, eps=1e-5):
    """x fp32 / bf16 row view [R, C]; out_f32 / out_t row views (either may be None)"""
    _req_cuda(x, gamma, beta, res, out_f32, out_t)
    R, C = x.shape
    dt = gim_dtype(out_t) if out_t is not None else GIM_F32
    check(lib.gim_layernorm_residual(_p(x), _p(gamma), _p(beta), _p(res), _p(out_f32), _p(out_t), R, C,
                                     x.stride(0), res.stride(0) if res is not None else 4,
                                     out_f32.stride(0) if out_f32 is not None else 4,
                                     out_t.stride(0) if out_t is not None else 4, gim_dtype(x), dt, eps, _stream()),
          "gim_layernorm_residual")


# ---- linear attention ---------------------------------------------------------------------------------
def linear_attention(q, k, v, out, nb_q, L, nb_kv, S, H, ws=None, q_mask=None, kv_mask=None):
    """q row view [nb_q*L, C]; k,v row views [nb_kv*S, C]; nb_q == nb_kv.  out row view [nb_q*L, C].
    q_mask [nb_q*L] / kv_mask [nb_kv*S]: uint8 padding masks or None."""
    _req_cuda(q, k, v, out, q_mask, kv_mask)
    assert nb_q == nb_kv
    C = H * (q.shape[1] // H)
    D = q.shape[1] // H
    if H == 8 and D in (16, 32) and L <= 64 and S <= 64:  # short sequences: fused single-launch kernel
        check(lib.gim_linear_attention_short(_p(q), _p(k), _p(v), _p(q_mask), _p(kv_mask), _p(out), nb_q, L, S, H, D,
                                             q.stride(0), k.stride(0), v.stride(0), out.stride(0), gim_dtype(q),
                                             gim_dtype(out), _stream()), "gim_linear_attention_short")
        return ws
    need = lib.gim_linear_attention_ws_bytes(nb_kv, S, H, D)
    if ws is None or ws.numel() * ws.element_size() < need:
        ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=q.device)
    check(lib.gim_linear_attention_kv(_p(k), _p(v), _p(kv_mask), _p(ws), nb_kv, S, H, D, k.stride(0), v.stride(0),
                                      gim_dtype(k), _stream()), "gim_linear_attention_kv")
    check(lib.gim_linear_attention_apply(_p(q), _p(q_mask), _p(ws), _p(out), nb_q, L, S, H, D, q.stride(0), out.stride(0),
                                         gim_dtype(q), gim_dtype(out), _stream()), "gim_linear_attention_apply")
    return ws


# ---- coarse matching -----------------------------------------------------------------------------------
class CoarseResult:
    __slots__ = ("args", "ws", "count", "b_ids", "i_ids", "j_ids", "mconf", "mkpts0_c", "mkpts1_c", "keep")


def coarse_match(feat0, feat1, hw0_c, hw1_c, scale, temperature=0.1, thr=0.2, border_rm=2,
                 scale0=None, scale1=None, mask0=None, mask1=None, count=None, precand_per_row=0):
    """feat0 [N,L,C], feat1 [N,S,C], fp32 or bf16, rows possibly strided (stride(1) = ldf >= C, the same for both; e.g. a
    column range of the transformer's token buffers).  bf16 features run the similarity on the bf16 MFMA (exact products,
    fp32 accumulation).  Returns CoarseResult with cap-sized device buffers; count[0] (device int32) is the number of
    valid leading entries."""
    _req_cuda(feat0, feat1, scale0, scale1)
    assert feat0.dtype == feat1.dtype and feat0.dtype in (torch.float32, torch.bfloat16, torch.float16)
    ldf = feat0.stride(1)
    for f in (feat0, feat1):
        assert f.dim() == 3 and f.stride(2) == 1 and f.stride(1) == ldf and f.stride(0) == f.shape[1] * ldf, (f.shape, f.stride())
    N, L, C = feat0.shape
    S = feat1.shape[1]
    dev = feat0.device
    cap = N * min(L, S)
    r = CoarseResult()
    r.ws = torch.empty(lib.gim_coarse_match_ws_bytes(N, L, S), dtype=torch.uint8, device=dev)
    # [M, health word, per-pair counts].  Bit 1 of the health word is sticky (include/gim_hip.h): a caller that captures this call
    # into a graph passes a `count` buffer it zeroed OUTSIDE the capture, so that replays do not clear the bit
    if count is not None:
        assert count.dtype == torch.int32 and count.numel() >= 2 + N and count.device == dev and count.is_contiguous()
    r.count = torch.zeros(2 + N, dtype=torch.int32, device=dev) if count is None else count
    r.b_ids = torch.empty(cap, dtype=torch.int64, device=dev)
    r.i_ids = torch.empty(cap, dtype=torch.int64, device=dev)
    r.j_ids = torch.empty(cap, dtype=torch.int64, device=dev)
    r.mconf = torch.empty(cap, dtype=torch.float32, device=dev)
    r.mkpts0_c = torch.empty(cap, 2, dtype=torch.float32, device=dev)
    r.mkpts1_c = torch.empty(cap, 2, dtype=torch.float32, device=dev)
    if scale0 is not None:
        scale0 = scale0.to(device=dev, dtype=torch.float32).contiguous()
        scale1 = scale1.to(device=dev, dtype=torch.float32).contiguous()
    if mask0 is not None:
        assert mask0.dtype == torch.uint8 and mask1.dtype == torch.uint8 and mask0.is_contiguous() and mask1.is_contiguous()
        assert mask0.numel() == N * L and mask1.numel() == N * S
    r.keep = (feat0, feat1, scale0, scale1, mask0, mask1)
    a = _lib.CoarseArgs()
    a.feat0, a.feat1 = feat0.data_ptr(), feat1.data_ptr()
    a.scale0 = scale0.data_ptr() if scale0 is not None else None
    a.scale1 = scale1.data_ptr() if scale1 is not None else None
    a.mask0 = mask0.data_ptr() if mask0 is not None else None
    a.mask1 = mask1.data_ptr() if mask1 is not None else None
    a.ws, a.count = r.ws.data_ptr(), r.count.data_ptr()
    a.b_ids, a.i_ids, a.j_ids = r.b_ids.data_ptr(), r.i_ids.data_ptr(), r.j_ids.data_ptr()
    a.mconf, a.mkpts0_c, a.mkpts1_c = r.mconf.data_ptr(), r.mkpts0_c.data_ptr(), r.mkpts1_c.data_ptr()
    a.N, a.L, a.S, a.C = N, L, S, C
    a.h0c, a.w0c, a.h1c, a.w1c = hw0_c[0], hw0_c[1], hw1_c[0], hw1_c[1]
    a.cap, a.temperature, a.thr, a.border_rm, a.scale = cap, temperature, thr, border_rm, scale
    a.feat_dtype, a.ldf = gim_dtype(feat0), ldf
    a.precand_per_row = precand_per_row   # 0: the default capacity; -1: none (tests force the overflow -> recompute fallback)
    r.args = a
    with _Timed("coarse_match", 2.0 * N * L * S * C):   # the similarity GEMM's flops (computed once in the common case)
        check(lib.gim_coarse_match(ctypes.byref(a), _stream()), "gim_coarse_match")
    return r


def coarse_conf_matrix(r):
    """Materialise conf_matrix [N,L,S] from the statistics of a previous coarse_match call."""
    a = r.args
    conf = torch.empty(a.N, a.L, a.S, dtype=torch.float32, device=r.count.device)
    check(lib.gim_coarse_conf_matrix(ctypes.byref(a), _p(conf), _stream()), "gim_coarse_conf_matrix")
    return conf


# ---- fine level ---------------------------------------------------------------------------------------
def fine_gather(feat_f0, feat_f1, b_ids, i_ids, j_ids, M, w0c, w1c, stride, W, out_f32, out_t):
    """feat_f0 [bs,hf0,wf0,C], feat_f1 [bs,hf1,wf1,C] NHWC; out_* row views [2*M*W*W, >=C]"""
    _req_cuda(feat_f0, feat_f1, b_ids, out_f32, out_t)
    _, hf0, wf0, C = feat_f0.shape
    _, hf1, wf1, _ = feat_f1.shape
    assert feat_f0.is_contiguous() and feat_f1.is_contiguous() and feat_f1.shape[3] == C
    check(lib.gim_fine_gather(_p(feat_f0), _p(feat_f1), _p(b_ids), _p(i_ids), _p(j_ids), _p(out_f32), _p(out_t),
                              M, hf0, wf0, hf1, wf1, C, C, w0c, w1c, stride, W,
                              out_f32.stride(0) if out_f32 is not None else 4,
                              out_t.stride(0) if out_t is not None else 4, gim_dtype(feat_f0), _stream()),
          "gim_fine_gather")


def fine_match(f0, f1, mkpts1_c, b_ids, scale1, M, WW, scale, has_scale0):
    """f0/f1 fp32 row views [M*WW, C].  Returns (expec_f [M,3], mkpts1_f [M,2])."""
    _req_cuda(f0, f1, mkpts1_c)
    dev = f0.device
    expec = torch.empty(M, 3, dtype=torch.float32, device=dev)
    mk1 = torch.empty(M, 2, dtype=torch.float32, device=dev)
    if M > 0:
        check(lib.gim_fine_match(_p(f0), _p(f1), _p(mkpts1_c), _p(b_ids), _p(scale1), _p(expec), _p(mk1), M, WW,
                                 f0.shape[1], f0.stride(0), scale, 1 if has_scale0 else 0, _stream()),
              "gim_fine_match")
    return expec, mk1


def _outs(out, shapes, like):
    """output tensors of a fused launch: fresh ones, or the caller's (`out` = a tuple aligned with `shapes`; None entries are
    allocated here) -- contiguous slices of a larger batch when a caller runs image groups separately"""
    res = []
    for i, shp in enumerate(shapes):
        o = out[i] if out is not None else None
        if shp is None:
            res.append(None)
        elif o is None:
            res.append(torch.empty(*shp, dtype=like.dtype, device=like.device))
        else:
            assert tuple(o.shape) == tuple(shp) and o.dtype == like.dtype and o.is_contiguous(), (o.shape, shp)
            res.append(o)
    return res


def bneck64(t1, res, pk, want_next, out=None, health=None):
    """Fused Bottleneck tail (+ the next conv1): t1 [B,H,W,64] bf16, res [B,H,W,256] bf16 -> (x' [B,H,W,256], t1' [B,H,W,N1] or None).
    pk = packing.pack_bneck(...); N1 = 64 (next block of the layer) or 128 (the next layer's first conv1) from the packed weights."""
    _req_cuda(t1, res)
    assert t1.dtype in HALF and res.dtype == t1.dtype and t1.is_contiguous() and res.is_contiguous()
    assert pk[0].dtype == t1.dtype, "bneck64: weights packed for the other 16-bit kind"
    fn = lib.gim_bneck64_fused_f16 if t1.dtype == torch.float16 else lib.gim_bneck64_fused
    B, H, W, _ = t1.shape
    w2, w3, w1n, b2, b3, b1n = pk
    n1 = w1n.shape[0] if (want_next and w1n is not None) else 0
    assert not want_next or n1 in (64, 128)
    xo, t1n = _outs(out, ((B, H, W, 256), (B, H, W, n1) if n1 else None), t1)
    with _Timed("bneck64_fused", 2.0 * B * H * W * (576 * 64 + 64 * 256 + 256 * n1)):
        check(fn(_p(t1), _p(res), _p(xo), _p(t1n), _p(w2), _p(w3), _p(w1n if n1 else None), _p(b2), _p(b3),
                                    _p(b1n if n1 else None), B, H, W, n1, _health(health), _stream()), "gim_bneck64_fused")
    return xo, t1n


def bneck64_ds(t1, x_in, pk, out=None, health=None):
    """First block of layer 1 with its downsample branch computed in the kernel (gim_bneck64_fused_ds): t1 [B,H,W,64] (conv1 output),
    x_in [B,H,W,64] (the block's input) -> (x' [B,H,W,256], t1' [B,H,W,64]); pk = packing.pack_bneck_ds(...)."""
    _req_cuda(t1, x_in)
    assert t1.dtype in HALF and x_in.dtype == t1.dtype and t1.is_contiguous() and x_in.is_contiguous() and x_in.shape == t1.shape
    w2, w3, wds, w1n, b2, b3ds, b1n = pk
    assert w2.dtype == t1.dtype and w1n.shape[0] == 64
    fn = lib.gim_bneck64_fused_ds_f16 if t1.dtype == torch.float16 else lib.gim_bneck64_fused_ds
    B, H, W, _ = t1.shape
    xo, t1n = _outs(out, ((B, H, W, 256), (B, H, W, 64)), t1)
    with _Timed("bneck64_fused", 2.0 * B * H * W * (576 * 64 + 64 * 256 + 64 * 256 + 256 * 64)):
        check(fn(_p(t1), _p(x_in), _p(xo), _p(t1n), _p(w2), _p(w3), _p(wds), _p(w1n), _p(b2), _p(b3ds), _p(b1n), B, H, W, _health(health), _stream()),
              "gim_bneck64_fused_ds")
    return xo, t1n


def bneck_tail(t2, res, pk, act_next=ACT_RELU, store_x=True, out=None, health=None):
    """Bottleneck tail + the next 1x1 convolution in one launch (planes P = 128: layer 2, P = 256: layer 3): t2 [B,H,W,P], res
    [B,H,W,4P] (same 16-bit dtype) -> (x' [B,H,W,4P] or None when store_x is False, t1' [B,H,W,N1]); pk = packing.pack_bneck_tail(...).
    B*H*W must be a multiple of 256."""
    _req_cuda(t2, res)
    assert t2.dtype in HALF and res.dtype == t2.dtype and t2.is_contiguous() and res.is_contiguous()
    w3, w1n, b3, b1n = pk
    assert w3.dtype == t2.dtype, "bneck_tail: weights packed for the other 16-bit kind"
    B, H, W, pl = t2.shape
    M, n1 = B * H * W, w1n.shape[1]
    assert pl in (128, 256) and w3.shape == (4 * pl, pl) and res.shape == (B, H, W, 4 * pl) and M % 256 == 0
    assert store_x or pl == 256
    xo, t1n = _outs(out, ((B, H, W, 4 * pl) if store_x else None, (B, H, W, n1)), t2)
    name = "gim_bneck_tail%d%s" % (pl, "_f16" if t2.dtype == torch.float16 else "")
    with _Timed("bneck_tail", 2.0 * M * (pl * 4 * pl + 4 * pl * n1)):
        check(getattr(lib, name)(_p(t2), _p(res), _p(xo), _p(t1n), _p(w3), _p(w1n), _p(b3), _p(b1n), M, n1, act_next, _health(health), _stream()), name)
    return xo, t1n


def bneck_tail_ds(t2, x_in, pk, act_next=ACT_RELU, out=None, health=None):
    """First block of layer 2 with its downsample branch inside the tail kernel (gim_bneck_tail128_ds): t2 [B,Ho,Wo,128] (conv2 output),
    x_in [B,Hin,Win,256] (the block's input; the stride-2 1x1 downsample convolution reads pixel (2y, 2x)) -> (x' [B,Ho,Wo,512], t1'
    [B,Ho,Wo,128]); pk = packing.pack_bneck_tail(blk, next_conv, next_bn, ..., ds=True)."""
    _req_cuda(t2, x_in)
    assert t2.dtype in HALF and x_in.dtype == t2.dtype and t2.is_contiguous() and x_in.is_contiguous()
    w3, w1n, b3, b1n = pk
    assert w3.dtype == t2.dtype, "bneck_tail_ds: weights packed for the other 16-bit kind"
    B, Ho, Wo, pl = t2.shape
    _, Hin, Win, cd = x_in.shape
    n1 = w1n.shape[1]
    assert pl == 128 and cd == 256 and w3.shape == (512, 384) and w1n.shape == (16, 128, 32) and x_in.shape[0] == B
    assert (Hin - 1) // 2 + 1 == Ho and (Win - 1) // 2 + 1 == Wo and (B * Ho * Wo) % 256 == 0
    xo, t1n = _outs(out, ((B, Ho, Wo, 512), (B, Ho, Wo, n1)), t2)
    fn = lib.gim_bneck_tail128_ds_f16 if t2.dtype == torch.float16 else lib.gim_bneck_tail128_ds
    with _Timed("bneck_tail", 2.0 * B * Ho * Wo * ((pl + cd) * 4 * pl + 4 * pl * n1)):
        check(fn(_p(t2), _p(x_in), _p(xo), _p(t1n), _p(w3), _p(w1n), _p(b3), _p(b1n), B, Ho, Wo, Hin, Win, n1, act_next, _health(health),
                 _stream()), "gim_bneck_tail128_ds")
    return xo, t1n


def token_mlp(msg, xb, x32, weights, ln_params, eps, kv=None, L=0, S=0, q_mask=None, emit=None, q_weights=None):
    """x += norm2(mlp.2(relu(mlp.0(cat[x, norm1(merge(msg))])))) on row views: msg [R, >=256] bf16, xb [R, >=256] bf16 (operand copy
    of x, updated in place), x32 [R, >=256] fp32 (updated in place).  With `kv` (the fp32 state of linear_attention_state) the
    attention's apply step is fused in front: `msg` then holds the elu+1 query rows of R / L sequences, S = source length.
    `emit` = (stream from packing.pack_token_emit, [(out [R, >=256] 16-bit row view, act, row_lo, row_hi), ...]): projection blocks
    act(x_new W_b^T) of the new x, written for the rows of the 64-row tiles that start in [row_lo, row_hi).
    A block entry with a fifth element (ws, nb_kv, nchunk, tile0, src_len) is the K block of a fused (k, v) pair -- the next entry is its V block, `out`
    of both may be None: nothing is written but each tile's partial KV state into the partial area of the linear-attention workspace `ws`
    (kv_state_workspace) of the call that consumes these rows as its source; kv_state_finalize(ws, ...) completes it.
    `q_weights` (with `kv`; pack_token_emit([Wq]) of this layer): `msg` may be None -- the query rows are projected inside the kernel from xb."""
    if msg is None:
        assert q_weights is not None and kv is not None, "token_mlp: msg=None needs q_weights and kv"
        msg = xb
    _req_cuda(msg, xb, x32, weights, ln_params, kv, q_mask, q_weights)
    assert msg.dtype in HALF and xb.dtype == msg.dtype and weights.dtype == msg.dtype and x32.dtype == torch.float32
    f16 = msg.dtype == torch.float16
    assert msg.stride(1) == 1 and xb.stride(1) == 1 and x32.stride(1) == 1 and msg.shape[0] == xb.shape[0] == x32.shape[0]
    R = msg.shape[0]
    flops = 2.0 * R * (256 * 256 + 512 * 512 + 512 * 256 + (32 * 256 if kv is not None else 0))
    em = None
    if q_weights is not None:
        assert kv is not None and q_weights.dtype == msg.dtype and q_weights.numel() == 256 * 256 and q_weights.is_contiguous()
        em = _lib.TokenEmit()
        em.q_weights = q_weights.data_ptr()
        flops += 2.0 * R * 256 * 256
    if emit:
        ew, blocks = emit
        assert 0 < len(blocks) <= _lib.TokenEmit.MAX and ew.dtype == msg.dtype and ew.is_cuda and ew.numel() == len(blocks) * 256 * 256
        em = em or _lib.TokenEmit()
        em.nblk, em.weights = len(blocks), ew.data_ptr()
        flops += _fill_emit_blocks(em, blocks, R, msg.dtype)
    with _Timed("token_mlp", flops):
        if em is None:
            fn = lib.gim_token_mlp_f16 if f16 else lib.gim_token_mlp
            check(fn(_p(msg), _p(xb), _p(x32), _p(weights), _p(ln_params), _p(kv), _p(q_mask), R, 256, L, S,
                     msg.stride(0), xb.stride(0), x32.stride(0), eps, _stream()), "gim_token_mlp")
        else:
            fn = lib.gim_token_mlp_emit_f16 if f16 else lib.gim_token_mlp_emit
            check(fn(_p(msg), _p(xb), _p(x32), _p(weights), _p(ln_params), _p(kv), _p(q_mask), R, 256, L, S,
                     msg.stride(0), xb.stride(0), x32.stride(0), eps, ctypes.byref(em), _stream()), "gim_token_mlp_emit")


def linear_attention_state(k, v, nb_kv, S, H, ws=None, kv_mask=None):
    """First half of linear_attention: KV / Ksum state of nb_kv sequences of S rows.  Returns (ws, need): the fp32 workspace
    (re-used when the one passed in is large enough) and the bytes the call needs of it.  The final state -- per sequence and
    head a D x D KV block followed by the D-vector Ksum, nb_kv*H*(D*D+D) floats -- sits at the START of the workspace, the
    per-chunk partials behind it (gim_linear_attention_kv); `ws` itself is what gim_token_mlp takes as `kv`.  NB the fused apply
    in token_mlp rounds KV to bf16 for the bf16 MFMA, the stand-alone la_apply kernel multiplies it in fp32."""
    _req_cuda(k, v, kv_mask)
    D = k.shape[1] // H
    need = lib.gim_linear_attention_ws_bytes(nb_kv, S, H, D)
    if ws is None or ws.numel() * ws.element_size() < need:
        ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=k.device)
    check(lib.gim_linear_attention_kv(_p(k), _p(v), _p(kv_mask), _p(ws), nb_kv, S, H, D, k.stride(0), v.stride(0),
                                      gim_dtype(k), _stream()), "gim_linear_attention_kv")
    return ws, need


def token_project(xb, emit, pos=None):
    """Projection blocks of the 16-bit rows `xb` [R, >= 256] as they are (gim_token_mlp_emit with project_only): `emit` as in token_mlp --
    the first layer's (k, v) pair handed over as partial KV states instead of a projection GEMM, its rows and la_kv launches.
    `pos` = (feat [R, 256] 16-bit rows, pe [hw, 256] fp32, x32 [R, >= 256] fp32 row view): the positional encoding in front (posenc_add's
    arithmetic) -- the kernel computes the rows feat + pe[row % hw] itself and WRITES them to x32 and xb before it projects them."""
    _req_cuda(xb)
    assert xb.dtype in HALF and xb.stride(1) == 1 and xb.shape[1] >= 256
    ew, blocks = emit
    R = xb.shape[0]
    assert 0 < len(blocks) <= _lib.TokenEmit.MAX and ew.dtype == xb.dtype and ew.is_cuda and ew.numel() == len(blocks) * 256 * 256
    em = _lib.TokenEmit()
    em.nblk, em.weights, em.project_only = len(blocks), ew.data_ptr(), 1
    x32 = None
    if pos is not None:
        feat, pe, x32 = pos
        _req_cuda(feat, pe, x32)
        assert feat.dtype == xb.dtype and feat.shape == (R, 256) and feat.stride(1) == 1 and pe.dtype == torch.float32 and pe.shape[1] == 256 and pe.is_contiguous()
        assert x32.dtype == torch.float32 and x32.shape[0] == R and x32.stride(1) == 1 and x32.shape[1] >= 256
        em.pe_feat, em.pe, em.pe_ld, em.pe_hw = feat.data_ptr(), pe.data_ptr(), feat.stride(0), pe.shape[0]
    _fill_emit_blocks(em, blocks, R, xb.dtype)
    fn = lib.gim_token_mlp_emit_f16 if xb.dtype == torch.float16 else lib.gim_token_mlp_emit
    with _Timed("token_mlp", 2.0 * sum(max(0, min(b[3], R) - b[2]) for b in blocks) * 256 * 256):
        check(fn(None, _p(xb), _p(x32), None, None, None, None, R, 256, 0, 0, 0, xb.stride(0), x32.stride(0) if x32 is not None else 0, 0.0,
                 ctypes.byref(em), _stream()), "gim_token_mlp_emit")


def _fill_emit_blocks(em, blocks, R, dtype):
    """block descriptors of a gim_token_emit (see token_mlp); returns the flops of the blocks"""
    flops = 0.0
    fused_v = False
    for b, blk in enumerate(blocks):
        out, act, lo, hi = blk[:4]
        if len(blk) > 4:      # K of a fused pair
            ws, nb_kv, nchunk, tile0, src_len = blk[4]
            _req_cuda(ws)
            assert ws.dtype == torch.float32 and ws.is_contiguous() and b + 1 < len(blocks) and len(blocks[b + 1]) == 4
            assert hi <= R and lo % 64 == 0 and hi % 64 == 0 and tile0 + (hi - lo) // 64 <= nb_kv * nchunk, "token_mlp: fused KV state: tile range"
            state = nb_kv * 8 * (32 * 32 + 32)
            assert ws.numel() >= state * (nchunk + 1), "token_mlp: KV-state workspace too small (kv_state_workspace)"
            em.kv_part[b], em.kv_nchunk[b], em.kv_tile0[b], em.kv_len[b] = ws.data_ptr() + 4 * state, nchunk, tile0, float(src_len)
            fused_v = True
        elif fused_v:         # V of the pair
            fused_v = False
        else:
            _req_cuda(out)
            assert out.dtype == dtype and out.stride(1) == 1 and out.shape[0] == R and out.shape[1] >= 256
            em.out[b], em.ld[b] = out.data_ptr(), out.stride(0)
        em.act[b], em.row_lo[b], em.row_hi[b] = act, lo, hi
        flops += 2.0 * max(0, min(hi, R) - lo) * 256 * 256
    return flops


def kv_state_workspace(nb_kv, nchunk, device, H=8, D=32):
    """fp32 workspace [state nb_kv x H x (D D + D)][partials nb_kv x H x nchunk x (D D + D)] for partial states written by token_mlp's
    fused (k, v) pairs (gim_linear_attention_ws_bytes_chunks)"""
    return torch.empty(lib.gim_linear_attention_ws_bytes_chunks(nb_kv, H, D, nchunk) // 4, dtype=torch.float32, device=device)


def kv_state_finalize(ws, nb_kv, nchunk, H=8, D=32):
    """state = sum of the partials (gim_linear_attention_finalize); afterwards `ws` is what token_mlp takes as `kv`"""
    _req_cuda(ws)
    assert ws.dtype == torch.float32 and ws.numel() * 4 >= lib.gim_linear_attention_ws_bytes_chunks(nb_kv, H, D, nchunk)
    check(lib.gim_linear_attention_finalize(_p(ws), nb_kv, H, D, nchunk, _stream()), "gim_linear_attention_finalize")
    return ws


def fine_fused(feat_f0, feat_f1, b_ids, i_ids, j_ids, mkpts1_c, scale1, weights, ln_params, M, w0c, w1c, stride, W,
               scale, ln_eps, has_scale0, debug=False, count=None):
    """Whole fine level in one launch (bf16 fine maps).  Returns (expec_f [M,3], mkpts1_f [M,2], fine0, fine1) where
    fine0/fine1 are fp32 [M, W*W, C] dumps of the transformer output when `debug`, else None.
    `count` (device int32 tensor): M is the CAPACITY of the match lists and the kernel processes the first min(M, count[0]) matches --
    the launch does not wait for the host to learn the count (gim_fine_fused_dev); rows beyond it are left unwritten."""
    _req_cuda(feat_f0, feat_f1, b_ids, mkpts1_c, weights, ln_params, count)
    if count is not None:
        assert not debug and count.dtype == torch.int32 and feat_f0.dtype in HALF and weights.dtype == feat_f0.dtype
        assert feat_f0.is_contiguous() and feat_f1.is_contiguous() and b_ids.numel() >= M
        fnd = lib.gim_fine_fused_dev_f16 if feat_f0.dtype == torch.float16 else lib.gim_fine_fused_dev
        _, hf0, wf0, C = feat_f0.shape
        _, hf1, wf1, _ = feat_f1.shape
        expec = torch.empty(M, 3, dtype=torch.float32, device=feat_f0.device)
        mk1 = torch.empty(M, 2, dtype=torch.float32, device=feat_f0.device)
        with _Timed("fine_fused", 0.0) as tm:   # the caller fills in the flops once it knows the count (patch_last_fused_flops)
            check(fnd(_p(feat_f0), _p(feat_f1), _p(b_ids), _p(i_ids), _p(j_ids), _p(mkpts1_c), _p(scale1), _p(weights), _p(ln_params),
                      _p(expec), _p(mk1), M, _p(count), hf0, wf0, hf1, wf1, C, C, w0c, w1c, stride, W, scale, ln_eps,
                      1 if has_scale0 else 0, _stream()), "gim_fine_fused_dev")
        return expec, mk1, None, None
    _req_cuda(feat_f0, feat_f1, b_ids, mkpts1_c, weights, ln_params)
    assert feat_f0.dtype in HALF and feat_f1.dtype == feat_f0.dtype and weights.dtype == feat_f0.dtype
    assert feat_f0.is_contiguous() and feat_f1.is_contiguous()
    fn = lib.gim_fine_fused_f16 if feat_f0.dtype == torch.float16 else lib.gim_fine_fused
    _, hf0, wf0, C = feat_f0.shape
    _, hf1, wf1, _ = feat_f1.shape
    dev = feat_f0.device
    expec = torch.empty(M, 3, dtype=torch.float32, device=dev)
    mk1 = torch.empty(M, 2, dtype=torch.float32, device=dev)
    d0 = torch.empty(M, W * W, C, dtype=torch.float32, device=dev) if debug else None
    d1 = torch.empty(M, W * W, C, dtype=torch.float32, device=dev) if debug else None
    assert weights.numel() * weights.element_size() == lib.gim_fine_fused_weight_bytes()
    with _Timed("fine_fused", 33.6e6 * M):   # 33.6 MFLOP per match
        check(fn(_p(feat_f0), _p(feat_f1), _p(b_ids), _p(i_ids), _p(j_ids), _p(mkpts1_c), _p(scale1),
                                 _p(weights), _p(ln_params), _p(expec), _p(mk1), _p(d0), _p(d1), M, hf0, wf0, hf1, wf1, C, C,
                                 w0c, w1c, stride, W, scale, ln_eps, 1 if has_scale0 else 0, _stream()), "gim_fine_fused")
    return expec, mk1, d0, d1


# ---- gim_lightglue: SuperPoint glue -----------------------------------------------------------------------
def maxpool2x2(x):
    """x [B,H,W,C] NHWC -> new [B,H/2,W/2,C]"""
    _req_cuda(x)
    B, H, W, C = x.shape
    y = torch.empty(B, H // 2, W // 2, C, dtype=x.dtype, device=x.device)
    check(lib.gim_maxpool2x2(_p(x), _p(y), B, H, W, C, C, C, gim_dtype(x), _stream()), "gim_maxpool2x2")
    return y


def sp_scores(logits, B, h, w):
    """logits rows [B*h*w, >=65] -> scores [B, 8h, 8w] fp32"""
    _req_cuda(logits)
    out = torch.empty(B, 8 * h, 8 * w, dtype=torch.float32, device=logits.device)
    check(lib.gim_sp_scores(_p(logits), _p(out), B, h, w, logits.stride(0), gim_dtype(logits), _stream()), "gim_sp_scores")
    return out


def sp_nms(scores, radius, border):
    """scores [B,H,W] fp32 -> new [B,H,W]: kept maxima keep their score, others 0, border -1"""
    _req_cuda(scores)
    assert scores.dtype == torch.float32 and scores.is_contiguous()
    B, H, W = scores.shape
    ws = torch.empty(lib.gim_sp_nms_ws_bytes(B, H, W), dtype=torch.uint8, device=scores.device)
    out = torch.empty_like(scores)
    check(lib.gim_sp_nms(_p(scores), _p(out), _p(ws), B, H, W, radius, border, _stream()), "gim_sp_nms")
    return out


def sp_topk(nms_scores, k, thr):
    """-> (kpts [B,k,2] (x,y) fp32, kscores [B,k], nvalid [B] int32 device)"""
    _req_cuda(nms_scores)
    B, H, W = nms_scores.shape
    dev = nms_scores.device
    ws = torch.empty(lib.gim_sp_topk_ws_bytes(B, H, W), dtype=torch.uint8, device=dev)
    kpts = torch.empty(B, k, 2, dtype=torch.float32, device=dev)
    ksc = torch.empty(B, k, dtype=torch.float32, device=dev)
    nv = torch.empty(B, dtype=torch.int32, device=dev)
    check(lib.gim_sp_topk(_p(nms_scores), _p(ws), _p(kpts), _p(ksc), _p(nv), B, H, W, k, thr, _stream()), "gim_sp_topk")
    return kpts, ksc, nv


def sp_sample_desc(dense, kpts, h, w, out_f32, out_t, cell=8):
    """dense rows [B*h*w, >=256]; kpts [B,K,2]; out_f32 / out_t row views [B*K, >=256] (either may be None)"""
    _req_cuda(dense, kpts, out_f32, out_t)
    B, K, _ = kpts.shape
    check(lib.gim_sp_sample_desc(_p(dense), _p(kpts), _p(out_f32), _p(out_t), B, K, h, w, 256, dense.stride(0),
                                 out_f32.stride(0) if out_f32 is not None else 4,
                                 out_t.stride(0) if out_t is not None else 4, cell, gim_dtype(dense), _stream()),
          "gim_sp_sample_desc")


# ---- gim_lightglue: matcher -------------------------------------------------------------------------------
def lg_posenc(kpts, size_wh, Wr):
    """kpts [B,K,2], size_wh [B,2] fp32 device, Wr [32,2] -> enc [B*K, 64] (cos | sin)"""
    _req_cuda(kpts, size_wh, Wr)
    B, K, _ = kpts.shape
    enc = torch.empty(B * K, 64, dtype=torch.float32, device=kpts.device)
    check(lib.gim_lg_posenc(_p(kpts), _p(size_wh), _p(Wr), _p(enc), B, K, _stream()), "gim_lg_posenc")
    return enc


def lg_rotary(x, enc, ncols):
    """in place on columns [0, ncols) of the row view x"""
    _req_cuda(x, enc)
    check(lib.gim_lg_rotary(_p(x), _p(enc), x.shape[0], ncols, x.stride(0), gim_dtype(x), _stream()), "gim_lg_rotary")


def lg_transpose(src, dst, nb, S, Sp, C):
    """src row view [nb*S, >=C] -> dst [nb, C, Sp] (same dtype)"""
    _req_cuda(src, dst)
    check(lib.gim_lg_transpose(_p(src), _p(dst), nb, S, Sp, C, src.stride(0), gim_dtype(src), _stream()), "gim_lg_transpose")


def sdpa(q, k, vt, out, nb, H, L, S, Sp, kv_shift=0, D=64):
    """q row view [nb*L, ..], k row view [nb*S, ..], vt [nb, H*D, Sp], out row view [nb*L, ..]; head dim D in {64, 128}"""
    _req_cuda(q, k, vt, out)
    assert q.dtype == k.dtype == vt.dtype
    check(lib.gim_sdpa(_p(q), _p(k), _p(vt), _p(out), nb, H, L, S, Sp, D, q.stride(0), k.stride(0), out.stride(0),
                       kv_shift, gim_dtype(q), gim_dtype(out), _stream()), "gim_sdpa")


def layernorm_act(x, gamma, beta, out, act=ACT_NONE, eps=1e-5):
    """x fp32 row view [R, C] -> out row view (fp32 / bf16)"""
    _req_cuda(x, gamma, beta, out)
    R, C = x.shape
    check(lib.gim_layernorm_act(_p(x), _p(gamma), _p(beta), _p(out), R, C, x.stride(0), out.stride(0), act,
                                gim_dtype(out), eps, _stream()), "gim_layernorm_act")


def cast_rows(src, dst):
    """fp32 row view -> dst row view of the same shape (fp32 / bf16)"""
    _req_cuda(src, dst)
    R, C = src.shape
    check(lib.gim_cast_rows(_p(src), _p(dst), R, C, src.stride(0), dst.stride(0), gim_dtype(dst), _stream()), "gim_cast_rows")


class AssignResult:
    __slots__ = ("args", "ws", "matches0", "matches1", "mscores0", "mscores1", "pos", "count", "keep")


def lg_assign(desc0, desc1, md0, md1, match_w, match_b, threshold):
    """desc0 [B,M,256] / desc1 [B,N,256] fp32 (row stride 256), md0/md1 = final_proj outputs (contiguous)."""
    _req_cuda(desc0, desc1, md0, md1, match_w, match_b)
    B, M, C = md0.shape
    N = md1.shape[1]
    dev = md0.device
    assert md0.is_contiguous() and md1.is_contiguous() and md0.dtype == torch.float32
    assert desc0.dtype == torch.float32 and desc0.stride(-1) == 1 and desc0.stride(-2) == desc1.stride(-2)
    r = AssignResult()
    r.ws = torch.empty(lib.gim_lg_assign_ws_bytes(B, M, N, C), dtype=torch.uint8, device=dev)
    r.matches0 = torch.empty(B, M, dtype=torch.int64, device=dev)
    r.matches1 = torch.empty(B, N, dtype=torch.int64, device=dev)
    r.mscores0 = torch.empty(B, M, dtype=torch.float32, device=dev)
    r.mscores1 = torch.empty(B, N, dtype=torch.float32, device=dev)
    r.pos = torch.empty(B, M, dtype=torch.int32, device=dev)
    r.count = torch.empty(B, dtype=torch.int32, device=dev)
    r.keep = (desc0, desc1, md0, md1, match_w, match_b)
    a = _lib.LgAssignArgs()
    a.desc0, a.desc1, a.md0, a.md1 = desc0.data_ptr(), desc1.data_ptr(), md0.data_ptr(), md1.data_ptr()
    a.match_w, a.match_b, a.ws = match_w.data_ptr(), match_b.data_ptr(), r.ws.data_ptr()
    a.matches0, a.matches1 = r.matches0.data_ptr(), r.matches1.data_ptr()
    a.mscores0, a.mscores1 = r.mscores0.data_ptr(), r.mscores1.data_ptr()
    a.pos, a.count = r.pos.data_ptr(), r.count.data_ptr()
    a.B, a.M, a.N, a.C, a.ld_desc, a.threshold = B, M, N, C, desc0.stride(-2), threshold
    r.args = a
    check(lib.gim_lg_assign(ctypes.byref(a), _stream()), "gim_lg_assign")
    return r


def lg_log_assignment(r):
    """Materialise log_assignment [B, M+1, N+1] from a previous lg_assign call's inputs."""
    a = r.args
    out = torch.empty(a.B, a.M + 1, a.N + 1, dtype=torch.float32, device=r.count.device)
    check(lib.gim_lg_log_assignment(ctypes.byref(a), _p(out), _stream()), "gim_lg_log_assignment")
    return out


def lg_emit_matches(r, total, kpts0=None, kpts1=None, scale0=None, scale1=None):
    """Packs the match lists of all pairs (torch.where order).  Returns (matches [total,2] int64, scores [total],
    mkpts0 [total,2] | None, mkpts1, m_bids)."""
    a = r.args
    dev = r.count.device
    matches = torch.empty(total, 2, dtype=torch.int64, device=dev)
    scores = torch.empty(total, dtype=torch.float32, device=dev)
    adapter = kpts0 is not None
    mk0 = torch.empty(total, 2, dtype=torch.float32, device=dev) if adapter else None
    mk1 = torch.empty(total, 2, dtype=torch.float32, device=dev) if adapter else None
    bids = torch.empty(total, dtype=torch.int64, device=dev) if adapter else None
    if total > 0:
        _req_cuda(kpts0, kpts1, scale0, scale1)
        check(lib.gim_lg_emit_matches(_p(r.matches0), _p(r.mscores0), _p(r.pos), _p(r.count), _p(kpts0), _p(kpts1),
                                      _p(scale0), _p(scale1), _p(matches), _p(scores), _p(mk0), _p(mk1), _p(bids),
                                      a.B, a.M, a.N, _stream()), "gim_lg_emit_matches")
    return matches, scores, mk0, mk1, bids


# ---- gim_dkm -------------------------------------------------------------------------------------------------
def maxpool3x3s2(x):
    """x [B,H,W,C] NHWC -> new [B,(H-1)//2+1,(W-1)//2+1,C] (kernel 3, stride 2, padding 1)"""
    _req_cuda(x)
    B, H, W, C = x.shape
    y = torch.empty(B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C, dtype=x.dtype, device=x.device)
    check(lib.gim_maxpool3x3s2(_p(x), _p(y), B, H, W, C, C, C, gim_dtype(x), _stream()), "gim_maxpool3x3s2")
    return y


def resize_bilinear(x, size, out_dtype=None):
    """x [B,h,w,C] NHWC contiguous -> new [B,Ho,Wo,C] (bilinear, align_corners=False)"""
    _req_cuda(x)
    B, h, w, C = x.shape
    y = torch.empty(B, size[0], size[1], C, dtype=out_dtype or x.dtype, device=x.device)
    check(lib.gim_resize_bilinear(_p(x), _p(y), B, h, w, size[0], size[1], C, C, C, gim_dtype(x), gim_dtype(y), _stream()),
          "gim_resize_bilinear")
    return y


def resize_image(img, dst, b_off=0):
    """img [B,C,h,w] fp32 NCHW -> dst[b_off:b_off+B] ([.,Ho,Wo,cpad] NHWC), bilinear align_corners=False"""
    _req_cuda(img, dst)
    B, C, h, w = img.shape
    check(lib.gim_resize_image(_p(img), _p(dst), B, C, h, w, dst.shape[1], dst.shape[2], dst.shape[3], b_off, gim_dtype(dst),
                               _stream()), "gim_resize_image")


def grid_sample(feat, grid, out):
    """feat [B,h,w,C]; grid [B,Ho,Wo,2] fp32; out: row view [B*Ho*Wo, >=C] (may be a channel slice)"""
    _req_cuda(feat, grid, out)
    B, h, w, C = feat.shape
    check(lib.gim_grid_sample(_p(feat), _p(grid), _p(out), B, h, w, grid.shape[1], grid.shape[2], C, C, out.stride(0),
                              gim_dtype(feat), _stream()), "gim_grid_sample")


def dkm_disp_emb(flow, wgt, bias, out):
    """flow [B,h,w,2] fp32; wgt [E,2], bias [E] fp32; out row view [B*h*w, >=E]"""
    _req_cuda(flow, wgt, bias, out)
    B, h, w, _ = flow.shape
    check(lib.gim_dkm_disp_emb(_p(flow), _p(wgt), _p(bias), _p(out), B, h, w, wgt.shape[0], out.stride(0), gim_dtype(out),
                               _stream()), "gim_dkm_disp_emb")


def local_corr(f0, f1, flow, r, out):
    """f0, f1 [B,h,w,C] (C = valid channels, row stride = shape[-1]); out row view [B*h*w, >=(2r+1)^2]"""
    _req_cuda(f0, f1, flow, out)
    B, h, w, C = f0.shape
    check(lib.gim_local_corr(_p(f0), _p(f1), _p(flow), _p(out), B, h, w, C, r, f0.stride(2), f1.stride(2), out.stride(0),
                             gim_dtype(f0), gim_dtype(out), _stream()), "gim_local_corr")


def dwconv5x5_bn_relu(x, wgt, scale, shift, cin, cout):
    """x [B,H,W,ldx]; wgt [25,cpad], scale/shift [cpad] fp32 -> new [B,H,W,cpad]"""
    _req_cuda(x, wgt, scale, shift)
    B, H, W, ldx = x.shape
    cpad = wgt.shape[1]
    y = torch.empty(B, H, W, cpad, dtype=x.dtype, device=x.device)
    check(lib.gim_dwconv5x5_bn_relu(_p(x), _p(wgt), _p(scale), _p(shift), _p(y), B, H, W, cin, cout, cpad, ldx, cpad,
                                    gim_dtype(x), _stream()), "gim_dwconv5x5_bn_relu")
    return y


def dwconv5x5_pw(x, wgt, scale, shift, pw_w, pw_b):
    """One ConvRefiner block in one launch (gim_dwconv5x5_pw): x [B,H,W,cs] 16-bit with cs = 144 (or 24 / 32) stored channels -> y [B,H,W,cs] =
    conv1x1(relu(bn(dwconv5x5(x)))) + bias; wgt [25, cs], scale / shift [cs] fp32 as for dwconv5x5_bn_relu; pw_w [NP, KP] in x's dtype, pw_b [NP] fp32
    (NP, KP = 160, 144 for cs = 144; 32, 32 for cs = 24 / 32)."""
    _req_cuda(x, wgt, scale, shift, pw_w, pw_b)
    B, H, W, cs = x.shape
    npc, kp = (160, 144) if cs == 144 else (32, 32)
    assert x.dtype in HALF and x.is_contiguous() and cs in (24, 32, 144) and pw_w.dtype == x.dtype and tuple(pw_w.shape) == (npc, kp) and pw_b.numel() == npc
    assert wgt.shape == (25, cs) and scale.numel() == cs and shift.numel() == cs
    y = torch.empty(B, H, W, cs, dtype=x.dtype, device=x.device)
    check(lib.gim_dwconv5x5_pw(_p(x), _p(wgt), _p(scale), _p(shift), _p(pw_w), _p(pw_b), _p(y), B, H, W, cs, cs, cs, gim_dtype(x), _stream()), "gim_dwconv5x5_pw")
    return y


def row_norms(x, C):
    """x row view [R, >=C] -> [R] fp32 L2 norms"""
    _req_cuda(x)
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    check(lib.gim_row_norms(_p(x), _p(out), x.shape[0], C, x.stride(0), gim_dtype(x), _stream()), "gim_row_norms")
    return out


def cos_kernel_finish(k, nx, ny, B, n, m, T, eps, diag_add):
    """k: fp32 dot products, rows [B*n, ld]; in place -> exp((k / (nx ny + eps) - 1) / T) (+ diag_add on i == j)"""
    _req_cuda(k, nx, ny)
    check(lib.gim_cos_kernel_finish(_p(k), _p(nx), _p(ny), B, n, m, k.stride(0), T, eps, diag_add, _stream()), "gim_cos_kernel_finish")


def gp_solve(K, F, npad):
    """K [B,n,ld] fp32 (SPD, sigma on the diagonal), F [B,n,nrhs] fp32 -> Xt [B,nrhs,npad] fp32 = (K^-1 F)^T, zero padded"""
    _req_cuda(K, F)
    B, n, ld = K.shape
    nrhs = F.shape[2]
    assert K.is_contiguous() and F.is_contiguous() and K.dtype == torch.float32 and F.dtype == torch.float32
    ws = torch.empty(lib.gim_gp_solve_ws_bytes(B, n, nrhs), dtype=torch.uint8, device=K.device)
    Xt = torch.empty(B, nrhs, npad, dtype=torch.float32, device=K.device)
    check(lib.gim_gp_solve(_p(K), _p(F), _p(Xt), _p(ws), B, n, ld, nrhs, npad, _stream()), "gim_gp_solve")
    return Xt


def gp_posterior_f64(X, Y, F, out, T=0.2, eps=1e-6, sigma=0.1):
    """GP posterior mean, every step in fp64 (the dense matchers' parity mode): X / Y [B,n,d] fp32 query / support rows (contiguous),
    F [n,nrhs] fp32 -> out: fp32 row view [B*n, >= nrhs] receives K_xy (K_yy + sigma I)^-1 F, K = exp((cos - 1) / T)."""
    _req_cuda(X, Y, F, out)
    B, n, d = X.shape
    nrhs = F.shape[1]
    assert X.is_contiguous() and Y.is_contiguous() and F.is_contiguous() and Y.shape == X.shape and F.shape[0] == n
    assert X.dtype == Y.dtype == F.dtype == out.dtype == torch.float32 and out.stride(1) == 1 and out.shape[0] >= B * n and out.shape[1] >= nrhs
    ws = torch.empty(lib.gim_gp_posterior_f64_ws_bytes(B, n, d, nrhs), dtype=torch.uint8, device=X.device)
    check(lib.gim_gp_posterior_f64(_p(X), _p(Y), _p(F), _p(out), _p(ws), B, n, d, d, nrhs, out.stride(0), T, eps, sigma, _stream()),
          "gim_gp_posterior_f64")


def global_avgpool(x, out, c_off):
    """x [B,h,w,C] -> out[b, c_off:c_off+C] (fp32 [B, ldo])"""
    _req_cuda(x, out)
    B, h, w, C = x.shape
    check(lib.gim_global_avgpool(_p(x), _p(out), B, h * w, C, C, out.stride(0), c_off, gim_dtype(x), _stream()), "gim_global_avgpool")


def cab_scale_add(g, x1, x2):
    """sigmoid(g[b,c]) * x2 + x1 (x1 may be None = zeros); x2 [B,h,w,C] -> new tensor"""
    _req_cuda(g, x1, x2)
    B, h, w, C = x2.shape
    out = torch.empty_like(x2)
    check(lib.gim_cab_scale_add(_p(g), _p(x1), _p(x2), _p(out), B, h * w, C, g.stride(0), C, C, C, gim_dtype(x2), _stream()),
          "gim_cab_scale_add")
    return out


def dkm_flow_update(flow, cert, d, sx, sy, cert_init=False, roma_layout=False):
    """flow [B,h,w,2], cert [B,h,w,1] fp32 updated in place from d rows [B*h*w, >=3] = [dcert, dx, dy] (DKM) or
    [dx, dy, dcert] (RoMa, roma.py:579)"""
    _req_cuda(flow, cert, d)
    check(lib.gim_dkm_flow_update(_p(flow), _p(cert), _p(d), flow.numel() // 2, d.stride(0), sx, sy,
                                  (1 if cert_init else 0) | (2 if roma_layout else 0), gim_dtype(d), _stream()), "gim_dkm_flow_update")


def dkm_grid_coords(B, h, w, device):
    flow = torch.empty(B, h, w, 2, dtype=torch.float32, device=device)
    check(lib.gim_dkm_grid_coords(_p(flow), B, h, w, _stream()), "gim_dkm_grid_coords")
    return flow


def dkm_black_mask(im, size):
    """im [1,3,h,w] fp32 NCHW -> uint8 [Ho,Wo]"""
    _req_cuda(im)
    m = torch.empty(size[0], size[1], dtype=torch.uint8, device=im.device)
    check(lib.gim_dkm_black_mask(_p(im), _p(m), im.shape[2], im.shape[3], size[0], size[1], _stream()), "gim_dkm_black_mask")
    return m


def dkm_match_post(flow, cert, low, black0, black1, warp, certainty):
    """flow (f0, f1) [H,W,2], cert / low (c0, c1) [H,W,1] of the two directions -> warp [H,2W,4], certainty [H,2W] (written in place)"""
    _req_cuda(flow[0], cert[0], low[0], black0, black1, warp, certainty)
    H, W, _ = flow[0].shape
    for t in (*flow, *cert, *low, warp, certainty):
        assert t.is_contiguous()
    check(lib.gim_dkm_match_post(_p(flow[0]), _p(flow[1]), _p(cert[0]), _p(cert[1]), _p(low[0]), _p(low[1]), _p(black0), _p(black1),
                                 _p(warp), _p(certainty), H, W, _stream()), "gim_dkm_match_post")


class RowMatrixOperand:
    """A row buffer [N_alloc, K] used as the [N][K] "weight" operand of gim_conv2d_bn_act (same fields as
    packing.PackedConv): y[m, n] = sum_k x[m, k] * w[n, k].  N_alloc must cover n rounded up to 64 rows (rows past n
    only produce columns that are never stored) and K must be a multiple of the 128-byte K slab."""
    _ktabs = {}

    def __init__(self, w, n, k):
        from .packing import KTILE_BYTES, NPAD, elem_size, group_elems
        dt = gim_dtype(w)
        es, g = elem_size(dt), group_elems(dt)
        assert w.dim() == 2 and w.stride(1) == 1 and w.stride(0) == k and (k * es) % KTILE_BYTES == 0, (w.shape, w.stride(), k)
        self.npad = (n + NPAD - 1) // NPAD * NPAD
        assert w.shape[0] >= self.npad, f"operand needs {self.npad} allocated rows, has {w.shape[0]}"
        key = (k, dt, str(w.device))
        if key not in RowMatrixOperand._ktabs:
            nkt = k * es // KTILE_BYTES
            gi = torch.arange((nkt + 2) * 8, dtype=torch.int64) * g
            ent = torch.where(gi < k, gi, torch.full_like(gi, 0xFF000000))
            ent = torch.where(ent >= 2 ** 31, ent - 2 ** 32, ent).to(torch.int32)
            RowMatrixOperand._ktabs[key] = ent.to(w.device)
        self.w, self.bias, self.ktab = w, None, RowMatrixOperand._ktabs[key]
        self.kh = self.kw = 1
        self.stride, self.pad = 1, 0
        self.cin = self.cin_pad = k
        self.cout = n
        self.n_store = (n + 3) // 4 * 4 if dt == GIM_F32 else (n + 7) // 8 * 8
        self.kpad, self.dtype = k, dt


def matmul_nt(x, w_rows, n, y):
    """y[:, :n] = x @ w_rows[:n].T on the igemm (x: row view [M, >=K]; w_rows: [N_alloc, K] contiguous; y: row view)."""
    pk = RowMatrixOperand(w_rows, n, w_rows.shape[1])
    M = x.shape[0]
    conv_rows(x, pk, (1, 1, M, 1, M), y)


def kde(x, std, half=False):
    """x [n,4] fp32 contiguous -> density [n]; half: coordinates rounded to fp16 first (RoMa's kde, roma.py:1018-1023)"""
    _req_cuda(x)
    assert x.dim() == 2 and x.shape[1] == 4 and x.is_contiguous() and x.dtype == torch.float32
    d = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    check(lib.gim_kde(_p(x), _p(d), x.shape[0], -std if half else std, _stream()), "gim_kde")
    return d


def cls_to_flow(logits, B, h, w, ncls):
    """logits rows [B*h*w, >= ncls + 1] fp32 (64x64 anchor classes + certainty) -> (flow [B,h,w,2], certainty [B,h,w,1])"""
    _req_cuda(logits)
    flow = torch.empty(B, h, w, 2, dtype=torch.float32, device=logits.device)
    cert = torch.empty(B, h, w, 1, dtype=torch.float32, device=logits.device)
    check(lib.gim_cls_to_flow(_p(logits), _p(flow), _p(cert), B * h * w, ncls, logits.stride(0), _stream()), "gim_cls_to_flow")
    return flow, cert


def weighted_sample(w, k, seed):
    """k distinct indices drawn with probability ~ w (fp32 [n], >= k positive entries) -> int64 [k], unordered"""
    _req_cuda(w)
    assert w.dim() == 1 and w.is_contiguous() and w.dtype == torch.float32
    n = w.shape[0]
    ws = torch.empty(lib.gim_weighted_sample_ws_bytes(n), dtype=torch.uint8, device=w.device)
    out = torch.empty(k, dtype=torch.int64, device=w.device)
    check(lib.gim_weighted_sample(_p(w), _p(out), _p(ws), n, k, seed & 0xffffffff, _stream()), "gim_weighted_sample")
    return out


def dense_to_pixels(matches, hw0, hw1):
    """matches [n,4] fp32 normalised -> (kpts0 [n,2], kpts1 [n,2]) in pixels of images of size hw0 / hw1 (h, w)"""
    _req_cuda(matches)
    n = matches.shape[0]
    m = matches.contiguous()
    k0 = torch.empty(n, 2, dtype=torch.float32, device=m.device)
    k1 = torch.empty(n, 2, dtype=torch.float32, device=m.device)
    if n == 0:
        return k0, k1
    check(lib.gim_dense_to_pixels(_p(m), _p(k0), _p(k1), n, float(hw0[1]), float(hw0[0]), float(hw1[1]), float(hw1[0]), _stream()),
          "gim_dense_to_pixels")
    return k0, k1
