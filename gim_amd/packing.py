"""Host-side weight pre-packing for libgimhip (runs once per checkpoint / device / precision).

  * eval-mode BatchNorm is folded into the preceding bias-free conv: w' = w * g/sqrt(var+eps),
    b' = beta - mean * g/sqrt(var+eps)   (reference layers: networks/loftr/backbone/resnet.py:90-126,277-289);
  * weights go to the kernel's [npad][kpad] K-contiguous layout with K ordered (ky, kx, c) and channels
    padded to the activation's stored channel count (196 -> 200 for bf16: 16-byte groups);
  * `ktab` maps every 16-byte K group to (dy, dx, c) for the implicit-GEMM gather, plus two trailing
    slabs of "invalid" entries (the kernel prefetches the table one slab ahead).

This is layout plumbing in torch on the host; no matching arithmetic happens here.
"""
import os

import torch

from . import _lib

KTILE_BYTES = _lib.lib.gim_ktile_bytes()
NPAD = _lib.lib.gim_npad_granule()


def torch_dtype(dt):
    return {_lib.GIM_BF16: torch.bfloat16, _lib.GIM_F16: torch.float16, _lib.GIM_F32: torch.float32}[dt]


def is_half(dt):
    """one of the two 16-bit operand kinds (bf16 / IEEE fp16: same kernels, same layouts)"""
    return dt in (_lib.GIM_BF16, _lib.GIM_F16)


def elem_size(dt):
    return 2 if is_half(dt) else 4


def group_elems(dt):
    """elements per 16-byte K group"""
    return 16 // elem_size(dt)


def cstore(c, dt):
    """channels actually stored for a c-channel activation (padded to a 16-byte group)"""
    g = group_elems(dt)
    return (c + g - 1) // g * g


def _round_up(x, m):
    return (x + m - 1) // m * m


class PackedConv:
    """One conv / linear layer in kernel layout."""
    __slots__ = ("w", "bias", "ktab", "kh", "kw", "stride", "pad", "cin", "cin_pad", "cout", "n_store",
                 "npad", "kpad", "dtype", "halo")

    def __repr__(self):
        return (f"PackedConv({self.cin}->{self.cout} k{self.kh} s{self.stride} cin_pad={self.cin_pad} "
                f"n_store={self.n_store} npad={self.npad} kpad={self.kpad} dt={self.dtype})")


def fold_bn(weight, bn):
    """bn = (gamma, beta, running_mean, running_var, eps) or None -> (weight', bias' or None), fp32."""
    w = weight.detach().float()
    if bn is None:
        return w, None
    gamma, beta, mean, var, eps = bn
    s = gamma.detach().float() / torch.sqrt(var.detach().float() + eps)
    return w * s.view(-1, 1, 1, 1), beta.detach().float() - mean.detach().float() * s


def split_channels(cin):
    """channels the split-operand layout stores for a cin-channel image: [hi | lo | hi]"""
    return 3 * cin


def pack_conv_split(weight, bn, dtype, device, stride=1, pad=0):
    """The first convolution on split operands (16-bit dtypes): BatchNorm is folded in fp32 first, then every weight becomes the
    pair w_hi = rn16(w), w_lo = rn16(w - w_hi) and the layer is packed as a 3*Cin-channel convolution [w_hi | w_hi | w_lo] that
    meets the image layout [x_hi | x_lo | x_hi] of gim_nchw_to_nhwc_split: x_hi w_hi + x_lo w_hi + x_hi w_lo (the x_lo w_lo term,
    2^-22 relative, is dropped).  Both halves are exactly representable, so the device-side cast of pack_conv is the identity."""
    assert is_half(dtype)
    w, b = fold_bn(weight, bn)
    td = torch_dtype(dtype)
    w_hi = w.to(td).float()
    w_lo = (w - w_hi).to(td).float()
    w3 = torch.cat([w_hi, w_hi, w_lo], dim=1)
    return pack_conv(w3, None, dtype, device, stride=stride, pad=pad, cin_pad=cstore(w3.shape[1], dtype), bias=b)


class PackedStem:
    """The first convolution for gim_stem7x7: `w` = the kernel's LDS image of the (BatchNorm-folded) filter bank, `bias` fp32 [64]."""

    def __init__(self, w, bias, split, dtype, cin):
        self.w, self.bias, self.split, self.dtype, self.cin = w, bias, bool(split), dtype, cin
        self.cout, self.kh, self.kw, self.stride, self.pad = 64, 7, 7, 2, 3


def stem7x7_image(weight, bn, dtype, split):
    """The filter bank of conv1 (backbone/resnet.py:306, + bn1 folded in fp32) as gim_stem7x7 reads it from LDS, as an fp32 tensor
    [NVT, 64, 2, 8] of values exactly representable in `dtype` (NVT "virtual taps" = MFMAs per accumulator, 64 output channels, 2 K
    halves of 8 channels), BEFORE the half-slot swizzle:
      split: virtual tap = tap (49); half 0 = [w_hi(C) | w_hi(C) | 0], half 1 = [w_lo(C) | 0] against the pixel [x_hi | x_lo | 0] that
             both halves read: x_hi w_hi + x_lo w_hi + x_hi w_lo (the 2^-22 relative x_lo w_lo term is dropped, as in pack_conv_split);
      plain: virtual tap v = taps 2 v (half 0) and 2 v + 1 (half 1; tap 49 does not exist: zeros), each [w(C) | 0].
    Returns (image, bias)."""
    assert is_half(dtype)
    w, b = fold_bn(weight, bn)
    cout, cin, kh, kw = w.shape
    assert (cout, kh, kw) == (64, 7, 7) and 2 * cin <= 8, "gim_stem7x7 is the 7x7, <= 4 -> 64 channel first convolution"
    td = torch_dtype(dtype)
    wt = w.float().cpu().permute(2, 3, 0, 1).reshape(49, 64, cin)          # [tap][cout][cin]
    w_hi = wt.to(td).float()
    if split:
        w_lo = (wt - w_hi).to(td).float()
        img = torch.zeros(49, 64, 2, 8)
        img[:, :, 0, :cin] = w_hi
        img[:, :, 0, cin:2 * cin] = w_hi
        img[:, :, 1, :cin] = w_lo
    else:
        img = torch.zeros(50, 64, 8)
        img[:49, :, :cin] = w_hi
        img = img.reshape(25, 2, 64, 8).permute(0, 2, 1, 3).contiguous()
    bias = b.float().cpu() if b is not None else torch.zeros(64)
    return img, bias


def pack_stem7x7(weight, bn, dtype, device, split=True):
    """-> PackedStem: stem7x7_image with K half h of output channel n stored at half slot h ^ ((n >> 3) & 1) (the kernel's
    conflict-free ds_read_b128 of a 32-row filter fragment), in the 16-bit kind of `dtype`, on `device`."""
    img, bias = stem7x7_image(weight, bn, dtype, split)
    n = torch.arange(64)
    sw = (n >> 3) & 1
    out = torch.empty_like(img)
    out[:, n, sw, :] = img[:, :, 0, :]
    out[:, n, 1 - sw, :] = img[:, :, 1, :]
    cin = weight.shape[1]
    return PackedStem(out.to(torch_dtype(dtype)).contiguous().to(device), bias.to(device), split, dtype, cin)


NPAD128 = True   # (module attribute: tests of the 64-granule path flip it)


def npad_for(cout):
    """rows of the packed weight matrix = what the kernels tile over N.  The granule is 64 (the 256 x 64 tile); a wide layer whose
    64-granule count is odd is padded to the next multiple of 128 when that costs at most 1/8 more tiles, so that it runs on the
    128 x 128 tile (DKM's 569-channel refiner: 576 -> 640 rows; ~725 instead of ~510 TFLOP/s on the layers' useful work)"""
    npad = _round_up(cout, NPAD)
    if NPAD128 and npad % 128 and npad >= 512:
        npad += 64
    return npad


def pack_conv(weight, bn, dtype, device, stride=1, pad=0, cin_pad=None, bias=None):
    """weight [Cout, Cin, kh, kw] (or [out, in] for a Linear).  Returns PackedConv on `device`."""
    if weight.dim() == 2:
        weight = weight[:, :, None, None]
    w, b = fold_bn(weight, bn)
    if bias is not None:
        b = bias.detach().float() if b is None else b + bias.detach().float()
    cout, cin, kh, kw = w.shape
    g = group_elems(dtype)
    es = elem_size(dtype)
    cin_pad = cstore(cin, dtype) if cin_pad is None else cin_pad
    assert cin_pad >= cin and cin_pad % g == 0
    kslab = KTILE_BYTES // es
    k = kh * kw * cin_pad
    kpad = _round_up(k, kslab)
    npad = npad_for(cout)
    wk = torch.zeros(npad, kpad, dtype=torch.float32)
    if kh == 1 and kw == 1:          # Linear / 1x1 conv: one pass (the 300 M-parameter ViT packs in seconds)
        wk[:cout, :cin] = w.reshape(cout, cin).cpu()
    else:
        wp = torch.zeros(npad, kh, kw, cin_pad, dtype=torch.float32)
        wp[:cout, :, :, :cin] = w.permute(0, 2, 3, 1).cpu()
        wk[:, :k] = wp.reshape(npad, k)
    nkt = kpad // kslab
    ngrp = (nkt + 2) * 8
    gidx = torch.arange(ngrp, dtype=torch.int64) * g
    tap = gidx // cin_pad
    c = gidx % cin_pad
    dy, dx = tap // kw, tap % kw
    ent = c | (dx << 16) | (dy << 24)
    ent = torch.where(tap < kh * kw, ent, torch.full_like(ent, 0xFF000000))
    ent = torch.where(ent >= 2 ** 31, ent - 2 ** 32, ent).to(torch.int32)

    p = PackedConv()
    p.w = wk.to(device).to(torch_dtype(dtype)).contiguous()     # rounding (RNE) on the device: same bits, no host pass
    if b is not None:
        bp = torch.zeros(npad, dtype=torch.float32)
        bp[:cout] = b.cpu()
        p.bias = bp.to(device)
    else:
        p.bias = None
    p.ktab = ent.to(device)
    p.kh, p.kw, p.stride, p.pad = kh, kw, stride, pad
    p.cin, p.cin_pad, p.cout = cin, cin_pad, cout
    p.n_store = cstore(cout, dtype)
    p.npad, p.kpad, p.dtype = npad, kpad, dtype
    p.halo = None
    if is_half(dtype) and kh == 3 and kw == 3 and stride == 1 and pad == 1:
        wh, tab, nslab = pack_halo(wp, cin_pad, device, torch_dtype(dtype))
        bh = p.bias
        if bh is not None and wh.shape[0] > npad:      # the halo kernel's N tile is 128 wide: its bias reads cover wh.shape[0] entries
            bh = torch.zeros(wh.shape[0], dtype=torch.float32, device=device)
            bh[:npad] = p.bias
        p.halo = (wh, tab, nslab, bh)
    return p


HALO_W2 = 34  # halo row length of the 8 x 32-pixel patch (gim_amd/csrc/conv_igemm.hip: conv3x3_halo_kernel)


def pack_halo(wp, cin_pad, device, tdt=torch.bfloat16):
    """Second packing of a 3x3 / stride-1 / pad-1 bf16 layer for the halo kernel.  wp: fp32 [npad, 3, 3, cin_pad] (BN folded,
    zero padded).  Returns (w [npad128, nslab * 64] bf16, table int32 [nslab * 8], nslab).  K order: for every full 64-channel
    chunk the nine taps (one slab each), then the remaining channels in 16-channel sub-steps, tap-major, four to a slab.
    Table row per slab: [chunk, flags (1 = first slab of its chunk, 2 = last), 4 x (row shift | channel sub-step << 8),
    channel base of the next chunk (-1: none, first slab only), unused]."""
    npad = wp.shape[0]
    npad = (npad + 127) // 128 * 128
    nw, rem = cin_pad // 64, cin_pad % 64
    nsub = (rem + 15) // 16
    nchunk = nw + (1 if rem else 0)
    cols, table = [], []
    for c in range(nw):
        for t in range(9):
            dy, dx = t // 3, t % 3
            cols.append(wp[:, dy, dx, c * 64:(c + 1) * 64])
            sh = dy * HALO_W2 + dx
            table.append([c, (1 if t == 0 else 0) | (2 if t == 8 else 0)] + [sh | (k << 8) for k in range(4)]
                         + [((c + 1) * 64 if c + 1 < nchunk else -1) if t == 0 else 0, 0])
    if rem:
        steps = [(u // nsub, u % nsub) for u in range(9 * nsub)]
        nsl = (len(steps) + 3) // 4
        for q in range(nsl):
            ent, blk = [], []
            for k in range(4):
                u = 4 * q + k
                if u < len(steps):
                    t, ksc = steps[u]
                    dy, dx = t // 3, t % 3
                    c0 = nw * 64 + ksc * 16
                    w16 = torch.zeros(wp.shape[0], 16)
                    n = max(0, min(16, cin_pad - c0))
                    w16[:, :n] = wp[:, dy, dx, c0:c0 + n]
                    blk.append(w16)
                    ent.append((dy * HALO_W2 + dx) | (ksc << 8))
                else:
                    blk.append(torch.zeros(wp.shape[0], 16))
                    ent.append(0)
            cols.append(torch.cat(blk, 1))
            table.append([nw, (1 if q == 0 else 0) | (2 if q == nsl - 1 else 0)] + ent + [-1 if q == 0 else 0, 0])
    wk = torch.zeros(npad, len(cols) * 64)
    wk[:wp.shape[0]] = torch.cat(cols, 1)
    tab = torch.tensor(table, dtype=torch.int32).reshape(-1)
    return wk.to(device).to(tdt).contiguous(), tab.to(device), len(cols)


def _frag_order(w):
    """[128 out][K] fp32 -> MFMA fragment order [wave = out/32][k16 step][lane = (k/8 % 2)*32 + out%32][8] (flat)."""
    n, k = w.shape
    assert n == 128 and k % 16 == 0
    return w.reshape(4, 32, k // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().reshape(-1)


def pack_fine_fused(layers, device, tdt=torch.bfloat16):
    """Weights of the 2-layer fine LocalFeatureTransformer for gim_fine_fused (see include/gim_hip.h): one bf16 stream
    [Wq | Wk | Wv | Wmerge | mlp.0[:128] | mlp.0[128:] | mlp.2[:, :128] | mlp.2[:, 128:]] per layer in fragment order,
    plus the fp32 LayerNorm parameters [g1 | b1 | g2 | b2] per layer.  `layers`: modules with q_proj/k_proj/v_proj/merge/
    mlp/norm1/norm2 (transformer.py:8-33)."""
    ws, lns = [], []
    for layer in layers:
        f = lambda t: t.detach().float().cpu()  # noqa: E731
        m0, m2 = f(layer.mlp[0].weight), f(layer.mlp[2].weight)
        assert m0.shape == (256, 256) and m2.shape == (128, 256)
        for blk in (f(layer.q_proj.weight), f(layer.k_proj.weight), f(layer.v_proj.weight), f(layer.merge.weight),
                    m0[:128], m0[128:], m2[:, :128], m2[:, 128:]):
            ws.append(_frag_order(blk.contiguous()))
        lns += [f(layer.norm1.weight), f(layer.norm1.bias), f(layer.norm2.weight), f(layer.norm2.bias)]
        assert abs(layer.norm1.eps - layer.norm2.eps) == 0
    w = torch.cat(ws).to(device).to(tdt).contiguous()
    assert w.numel() * 2 == _lib.lib.gim_fine_fused_weight_bytes()
    return w, torch.cat(lns).to(device).contiguous()


def _frag(w, n0, k0):
    """MFMA fragment W[n0:n0+32, k0:k0+16] in lane order: lane = (k/8 % 2) * 32 + n -> 8 consecutive k."""
    return w[n0:n0 + 32, k0:k0 + 16].reshape(32, 2, 8).permute(1, 0, 2).reshape(-1)


def pack_token_mlp(layer, device, tdt=torch.bfloat16):
    """Weights of one coarse LoFTREncoderLayer's token-wise tail for gim_token_mlp: per wave w (output columns 64w.. of merge / mlp.2,
    hidden columns 32w.. of every 128-column quarter) the fragments in the order the kernel consumes them:
      merge:   4 units x (4 k16 steps x 2 column fragments)
      quarter hq = 0..3:  mlp.0 rows 128hq + 32w..: 4 units x 8 k16 steps (K = [x | msg]);  mlp.2 cols 128hq..: 2 units x (4 k16 x 2)
    Returns (bf16 weight stream, fp32 [norm1.weight | norm1.bias | norm2.weight | norm2.bias], eps)."""
    f = lambda t: t.detach().float().cpu()  # noqa: E731
    wm, w0, w2 = f(layer.merge.weight), f(layer.mlp[0].weight), f(layer.mlp[2].weight)
    assert wm.shape == (256, 256) and w0.shape == (512, 512) and w2.shape == (256, 512)
    out = []
    for w in range(4):
        for q in range(4):
            for k in range(4):
                for nf in range(2):
                    out.append(_frag(wm, 64 * w + 32 * nf, 16 * (4 * q + k)))
        for hq in range(4):
            for q in range(4):
                for k in range(8):
                    out.append(_frag(w0, 128 * hq + 32 * w, 16 * (8 * q + k)))
            for q in range(2):
                for k in range(4):
                    for nf in range(2):
                        out.append(_frag(w2, 64 * w + 32 * nf, 128 * hq + 16 * (4 * q + k)))
    wts = torch.cat(out).to(device).to(tdt).contiguous()
    assert wts.numel() * 2 == _lib.lib.gim_token_mlp_weight_bytes()
    ln = torch.cat([f(layer.norm1.weight), f(layer.norm1.bias), f(layer.norm2.weight), f(layer.norm2.bias)]).to(device).contiguous()
    assert layer.norm1.eps == layer.norm2.eps
    return wts, ln, layer.norm1.eps


def pack_token_emit(weights, device, tdt=torch.bfloat16):
    """Projection blocks of gim_token_mlp_emit: `weights` = list of [256, 256] Linear weights (q_proj / k_proj / v_proj of the layer
    that consumes the tokens next).  Per wave w (output columns 64w..) and block the fragments in the merge product's order: 4 units
    x (4 k16 steps x 2 column fragments).  Returns the 16-bit stream [wave][block][unit]."""
    out = []
    ws = [t.detach().float().cpu() for t in weights]
    assert ws and all(tuple(t.shape) == (256, 256) for t in ws)
    for w in range(4):
        for wb in ws:
            for q in range(4):
                for k in range(4):
                    for nf in range(2):
                        out.append(_frag(wb, 64 * w + 32 * nf, 16 * (4 * q + k)))
    return torch.cat(out).to(device).to(tdt).contiguous()


def _acc_order(k):
    """Column permutation that brings a weight's K axis into MFMA accumulator order: position 16s + 8lh + p of the packed row holds
    input channel 32(s >> 1) + 16(s & 1) + 8(p >> 2) + 4lh + (p & 3) -- what lane-half lh of a 32x32 accumulator fragment holds
    for k16 step s after two v_cvt_pk per quad (csrc/bneck_fused.hip)."""
    q = torch.arange(k)
    s_, lh, p = q // 16, (q // 8) % 2, q % 8
    return 32 * (s_ // 2) + 16 * (s_ % 2) + 8 * (p // 4) + 4 * lh + (p % 4)


def pack_bneck(blk, nxt, device, tdt=torch.bfloat16):
    """gim_bneck64_fused operands of Bottleneck `blk` (conv2/bn2, conv3/bn3) and, if given, the next block's conv1/bn1:
    (w2 [64][576] bf16 K=(ky,kx,c), w3 [256][64] bf16 K in accumulator order, w1n [64 or 128][256] bf16 or None, b2, b3, b1n fp32)."""
    bn = lambda m: (m.weight, m.bias, m.running_mean, m.running_var, m.eps)  # noqa: E731
    w2, b2 = fold_bn(blk.conv2.weight, bn(blk.bn2))
    w3, b3 = fold_bn(blk.conv3.weight, bn(blk.bn3))
    assert tuple(w2.shape) == (64, 64, 3, 3) and tuple(w3.shape) == (256, 64, 1, 1) and blk.conv2.stride == (1, 1)
    to = lambda t: t.to(device).to(tdt).contiguous()  # noqa: E731
    w2p = to(w2.permute(0, 2, 3, 1).reshape(64, 576).cpu())
    w3p = to(w3.reshape(256, 64).cpu()[:, _acc_order(64)])
    w1p = b1 = None
    if nxt is not None:
        w1, b1 = fold_bn(nxt.conv1.weight, bn(nxt.bn1))
        n1 = w1.shape[0]
        assert tuple(w1.shape) == (n1, 256, 1, 1) and n1 in (64, 128)
        w1p = to(w1.reshape(n1, 256).cpu()[:, _acc_order(256)])
        b1 = b1.float().to(device).contiguous()
    return w2p, w3p, w1p, b2.float().to(device).contiguous(), b3.float().to(device).contiguous(), b1


def pack_bneck_ds(blk, nxt, device, tdt=torch.bfloat16):
    """gim_bneck64_fused_ds operands: pack_bneck(blk, nxt) plus the block's downsample branch (1x1 conv 64 -> 256 + BN, stride 1) as
    wds [256][64] in channel order, its bias added to conv3's: (w2, w3, wds, w1n, b2, b3 + bds, b1n)."""
    bn = lambda m: (m.weight, m.bias, m.running_mean, m.running_var, m.eps)  # noqa: E731
    w2, w3, w1n, b2, b3, b1n = pack_bneck(blk, nxt, device, tdt)
    conv, norm = blk.downsample[0], blk.downsample[1]
    assert tuple(conv.weight.shape) == (256, 64, 1, 1) and conv.stride == (1, 1) and w1n is not None and w1n.shape[0] == 64
    wd, bd = fold_bn(conv.weight, bn(norm))
    wds = wd.reshape(256, 64).cpu().to(device).to(tdt).contiguous()
    return w2, w3, wds, w1n, b2, (b3 + bd.float().to(device)).contiguous(), b1n


def pack_bneck_tail(blk, next_conv, next_bn, device, tdt=torch.bfloat16, ds=False):
    """gim_bneck_tail128 / 256 operands: conv3 / bn3 of Bottleneck `blk` (planes P = 128 or 256) and the 1x1 convolution that consumes
    the block's output next -- the following block's conv1 with its bn1, or (last block of layer 3) the FPN's layer3_outconv, which has
    no BatchNorm (next_bn = None):
    (w3 [4P][P] K in channel order, w1n [chunks][N1][CH] -- per CH-channel chunk of x' the K axis in accumulator order, CH = 64 for
    P = 128, 32 for P = 256 --, b3, b1n fp32).
    ds=True (gim_bneck_tail128_ds, planes 128): the block's downsample branch (1x1 conv 2P -> 4P, stride 2, + BN) rides along as extra K of
    conv3 -- w3 becomes [4P][P + 2P] = [W3 | Wds] (both in channel order), b3 + bds its bias, and the chunks are 32 channels wide."""
    bn = lambda m: (m.weight, m.bias, m.running_mean, m.running_var, m.eps)  # noqa: E731
    w3, b3 = fold_bn(blk.conv3.weight, bn(blk.bn3))
    w1, b1 = fold_bn(next_conv.weight, bn(next_bn) if next_bn is not None else None)
    c4, pl = w3.shape[0], w3.shape[1]
    n1 = w1.shape[0]
    assert pl in (128, 256) and c4 == 4 * pl and tuple(w1.shape) == (n1, c4, 1, 1) and n1 in ((128, 256) if pl == 128 else (256,))
    ch = 64 if (pl == 128 and not ds) else 32
    to = lambda t: t.to(device).to(tdt).contiguous()  # noqa: E731
    w3m = w3.reshape(c4, pl).cpu()
    if ds:
        conv, norm = blk.downsample[0], blk.downsample[1]
        assert pl == 128 and n1 == 128 and tuple(conv.weight.shape) == (c4, 2 * pl, 1, 1) and conv.stride == (2, 2)
        wd, bd = fold_bn(conv.weight, bn(norm))
        w3m = torch.cat([w3m, wd.reshape(c4, 2 * pl).cpu()], dim=1)
        b3 = b3 + bd
    w3p = to(w3m)
    w1c = w1.reshape(n1, c4 // ch, ch).cpu()[:, :, _acc_order(ch)].permute(1, 0, 2)       # [chunk][n1][ch]
    b1 = b1 if b1 is not None else torch.zeros(n1)
    return w3p, to(w1c), b3.float().to(device).contiguous(), b1.float().to(device).contiguous()
