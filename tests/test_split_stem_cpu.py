"""The split-operand first convolution (round 4: `stem_split`) on the CPU: packing.pack_conv_split lays the BatchNorm-folded 7x7
filters out as [w_hi | w_hi | w_lo] over 3 x Cin channels; against the image layout [x_hi | x_lo | x_hi] of gim_nchw_to_nhwc_split
the 16-bit products sum to x * w up to 2^-21 -- where the plainly rounded operands are off by 2^-11 (backbone/resnet.py:306: the
conv1 + bn1 the engine folds).  Pure host logic: the packed tensor is unpacked again and convolved with torch."""
import pytest
import torch
import torch.nn.functional as F

from gim_amd import _lib
from gim_amd.packing import cstore, fold_bn, pack_conv, pack_conv_split, split_channels


def _split_image(x, td):
    hi = x.to(td)
    lo = (x - hi.float()).to(td)
    return torch.cat([hi, lo, hi], dim=1).float()


@pytest.mark.parametrize("dt,td", [(_lib.GIM_F16, torch.float16), (_lib.GIM_BF16, torch.bfloat16)], ids=["fp16", "bf16"])
def test_pack_conv_split_reproduces_the_fp32_convolution(dt, td):
    g = torch.Generator().manual_seed(5)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    bn = (torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1, torch.randn(64, generator=g) * 0.1,
          torch.rand(64, generator=g) + 0.5, 1e-5)
    x = torch.rand(2, 3, 40, 56, generator=g)       # image intensities in [0, 1): a large mean in front of cancelling filters
    wf, bf = fold_bn(w, bn)
    ref = F.conv2d(x.double(), wf.double(), bf.double(), stride=2, padding=3)

    pk = pack_conv_split(w, bn, dt, "cpu", stride=2, pad=3)
    assert pk.cin == split_channels(3) == 9 and pk.cin_pad == cstore(9, dt) == 16 and pk.kh == 7 and pk.stride == 2 and pk.pad == 3
    k = 7 * 7 * pk.cin_pad
    w9 = pk.w[:64, :k].float().reshape(64, 7, 7, pk.cin_pad)[..., :9].permute(0, 3, 1, 2).contiguous()   # K order (ky, kx, c)
    assert torch.equal(pk.w[:64, :k].float().reshape(64, 7, 7, pk.cin_pad)[..., 9:], torch.zeros(64, 7, 7, pk.cin_pad - 9))
    got = F.conv2d(_split_image(x, td).double(), w9.double(), pk.bias[:64].double(), stride=2, padding=3)

    plain = pack_conv(w, bn, dt, "cpu", stride=2, pad=3, cin_pad=cstore(3, dt))
    w3 = plain.w[:64, :7 * 7 * plain.cin_pad].float().reshape(64, 7, 7, plain.cin_pad)[..., :3].permute(0, 3, 1, 2).contiguous()
    one = F.conv2d(x.to(td).double(), w3.double(), plain.bias[:64].double(), stride=2, padding=3)

    scale = ref.abs().max()
    e_split, e_plain = (got - ref).abs().max() / scale, (one - ref).abs().max() / scale
    eps = 2.0 ** (-11 if td == torch.float16 else -8)
    assert e_split < 4 * eps * eps * 16, e_split          # ~2^-21 (2^-15 for bf16 pairs) with head-room for the 147-term sums
    assert e_plain > 50 * e_split, (e_plain, e_split)      # the error the split removes
