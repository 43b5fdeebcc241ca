"""The filter-bank image of gim_stem7x7 (round 4: the first convolution's own kernel) on the CPU.  packing.stem7x7_image lays the
BatchNorm-folded 7x7 filters (backbone/resnet.py:306: conv1 + bn1) out per "virtual tap" = one MFMA of the kernel: 64 output channels
x 2 K halves x 8 channels.  The kernel's arithmetic is restated here in torch (fp64 sums of the exactly representable 16-bit
products) and compared with the fp32 layer: the split image reproduces it to 2^-20, the plain image is the convolution of the
rounded operands.  pack_stem7x7's half-slot swizzle is the permutation the kernel's LDS reads undo."""
import pytest
import torch
import torch.nn.functional as F

from gim_amd import _lib
from gim_amd.packing import fold_bn, pack_stem7x7, stem7x7_image

KINDS = [(_lib.GIM_F16, torch.float16), (_lib.GIM_BF16, torch.bfloat16)]


def _layer(seed):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    bn = (torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1, torch.randn(64, generator=g) * 0.1,
          torch.rand(64, generator=g) + 0.5, 1e-5)
    return w, bn


def _pixels(x, td, split):
    """[B,3,H,W] fp32 -> the kernel's 8-channel pixels [B,8,H,W] (values of the 16-bit kind, as fp64)"""
    hi = x.to(td)
    out = torch.zeros(x.shape[0], 8, *x.shape[2:], dtype=torch.float64)
    out[:, :3] = hi.double()
    if split:
        out[:, 3:6] = (x - hi.float()).to(td).double()
    return out


def _kernel_sum(img, px, split):
    """what the MFMA chain of stem7x7_kernel adds up: virtual tap vt, K half h, channel c -> img[vt, n, h, c] * pixel_h(vt)[c]"""
    B, _, H, W = px.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    cols = F.unfold(px, 7, padding=3, stride=2).reshape(B, 8, 49, Ho * Wo)          # [B][c][tap][pixel]
    cols = torch.cat([cols, torch.zeros(B, 8, 1, Ho * Wo, dtype=cols.dtype)], dim=2)  # tap 49 (plain mode's padding tap)
    y = torch.zeros(B, 64, Ho * Wo, dtype=torch.float64)
    for vt in range(img.shape[0]):
        for h in range(2):
            tap = vt if split else 2 * vt + h
            y += torch.einsum("nc,bcp->bnp", img[vt, :, h].double(), cols[:, :, tap])
    return y.reshape(B, 64, Ho, Wo)


@pytest.mark.parametrize("dt,td", KINDS, ids=["fp16", "bf16"])
def test_split_image_reproduces_the_fp32_layer(dt, td):
    w, bn = _layer(5)
    x = torch.rand(2, 3, 21, 30, generator=torch.Generator().manual_seed(6))
    wf, bf = fold_bn(w, bn)
    ref = F.conv2d(x.double(), wf.double(), bf.double(), stride=2, padding=3)
    img, bias = stem7x7_image(w, bn, dt, True)
    assert img.shape == (49, 64, 2, 8) and torch.equal(img.to(td).float(), img)        # exactly representable
    assert torch.equal(img[:, :, 0, 6:], torch.zeros(49, 64, 2)) and torch.equal(img[:, :, 1, 3:], torch.zeros(49, 64, 5))
    got = _kernel_sum(img, _pixels(x, td, True), True) + bias.double()[None, :, None, None]
    eps = 2.0 ** -(11 if td == torch.float16 else 8)
    scale = ref.abs().max()
    assert (got - ref).abs().max() / scale < 8 * eps * eps                              # 3 cross terms of 2^-22 (2^-16), summed
    one = F.conv2d(x.to(td).double(), wf.to(td).double(), bf.double(), stride=2, padding=3)
    assert (one - ref).abs().max() > 50 * (got - ref).abs().max()                       # what the split buys


@pytest.mark.parametrize("dt,td", KINDS, ids=["fp16", "bf16"])
def test_plain_image_is_the_convolution_of_the_rounded_operands(dt, td):
    w, bn = _layer(7)
    x = torch.rand(1, 3, 18, 25, generator=torch.Generator().manual_seed(8)) * 255.0
    wf, bf = fold_bn(w, bn)
    img, bias = stem7x7_image(w, bn, dt, False)
    assert img.shape == (25, 64, 2, 8) and torch.equal(img[24, :, 1], torch.zeros(64, 8))   # tap 49 does not exist
    got = _kernel_sum(img, _pixels(x, td, False), False) + bias.double()[None, :, None, None]
    one = F.conv2d(x.to(td).double(), wf.to(td).double(), bf.double(), stride=2, padding=3)
    assert (got - one).abs().max() <= 1e-9 * one.abs().max()


@pytest.mark.parametrize("split", [True, False], ids=["split", "plain"])
def test_pack_stem7x7_half_slot_swizzle(split):
    w, bn = _layer(9)
    img, bias = stem7x7_image(w, bn, _lib.GIM_F16, split)
    ps = pack_stem7x7(w, bn, _lib.GIM_F16, "cpu", split=split)
    assert ps.w.dtype == torch.float16 and ps.w.is_contiguous() and ps.split == split and ps.cin == 3
    assert ps.w.numel() * 2 == img.shape[0] * 64 * 32                                   # = gim_stem7x7_weight_bytes(split)
    assert torch.equal(ps.bias, bias)
    for n in range(64):
        s = (n >> 3) & 1
        for h in range(2):
            assert torch.equal(ps.w[:, n, h ^ s].float(), img[:, n, h])                 # the kernel reads half h at slot h ^ ((n >> 3) & 1)
