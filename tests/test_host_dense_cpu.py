"""CPU: host-side logic of the dense matchers (gim_dkm / gim_roma) that needs no device -- constant tables built once per
shape, weight folding at pack time, padding arithmetic of the caller adapters."""
import math

import torch
import torch.nn.functional as F

import roma_oracle as RO
from gim_amd._lib import GIM_F32
from gim_amd.adapters import get_padding_size
from gim_amd.dkm.dkm import _bn_after_bias
from gim_amd.packing import pack_conv
from gim_amd.roma.roma import VIT_DIM, VIT_GRID, RegressionMatcher


def _unpack(pk):
    """packed [npad][kpad] fp32 weights of a 1x1 / kxk conv -> [cout, cin, kh, kw] + bias"""
    w = pk.w[:pk.cout, :pk.kh * pk.kw * pk.cin_pad].reshape(pk.cout, pk.kh, pk.kw, pk.cin_pad)[..., :pk.cin].permute(0, 3, 1, 2)
    return w.contiguous(), pk.bias[:pk.cout]


def test_conv_bias_then_batchnorm_fold():
    """conv (with bias) -> eval BatchNorm == packed conv with the bias moved into the running mean (VGG19-BN, RRB, proj)"""
    g = torch.Generator().manual_seed(0)
    conv = torch.nn.Conv2d(5, 7, 3, padding=1)
    bn = torch.nn.BatchNorm2d(7).eval()
    with torch.no_grad():
        bn.running_mean.copy_(torch.randn(7, generator=g) * 0.3)
        bn.running_var.copy_(0.5 + torch.rand(7, generator=g))
        bn.weight.copy_(1 + 0.2 * torch.randn(7, generator=g))
        bn.bias.copy_(0.1 * torch.randn(7, generator=g))
    x = torch.randn(2, 5, 9, 11, generator=g)
    pk = pack_conv(conv.weight, _bn_after_bias(bn, conv.bias), GIM_F32, torch.device("cpu"), pad=1, cin_pad=8)
    w, b = _unpack(pk)
    with torch.no_grad():
        ref = bn(conv(x))
        got = F.conv2d(x, w, b, padding=1)
    assert (got - ref).abs().max() < 1e-5


def test_layerscale_fold_and_block_packing():
    """x + ls * (W a + b) == x + W' a + b' with the LayerScale folded into the producing Linear (DINOv2 blocks)"""
    g = torch.Generator().manual_seed(1)
    D = 64
    W, b, ls = torch.randn(D, D, generator=g), torch.randn(D, generator=g), 0.2 + 0.05 * torch.randn(D, generator=g)
    P = {}
    z = torch.zeros(D)
    RegressionMatcher._pack_block(P, "t.", GIM_F32, torch.device("cpu"), (z + 1, z), torch.randn(3 * D, D, generator=g), None, W, b,
                                  (z + 1, z), torch.randn(4 * D, D, generator=g), torch.zeros(4 * D), torch.randn(D, 4 * D, generator=g),
                                  b, ls, ls)
    a = torch.randn(10, D, generator=g)
    wp, bp = _unpack(P["t.proj"])
    assert torch.allclose(a @ wp[:, :, 0, 0].t() + bp, ls * (a @ W.t() + b), atol=1e-5)
    assert P["t.qkv"].bias is None and P["t.fc1"].cout == 4 * D


def test_roma_constant_tables_match_oracle():
    """bicubic position table (dino.py:457-488 incl. the +0.1 trick and the (H, W) argument order) and the GP's Fourier
    features (roma.py:94-108) are built on the host: same values as the oracle's"""
    g = torch.Generator().manual_seed(2)
    m = RegressionMatcher(h=112, w=140)
    dsd = {"pos_embed": 0.02 * torch.randn(1, VIT_GRID ** 2 + 1, VIT_DIM, generator=g), "cls_token": 0.02 * torch.randn(1, 1, VIT_DIM, generator=g)}
    m._dino[0] = dsd
    for hs, ws in ((112, 140), (140, 112), (518, 518)):
        cls_row, patch = m._pos_table(hs, ws, torch.device("cpu"))
        ref = RO.dino_pos_embed(dsd, hs, ws)
        assert torch.equal(patch, ref[0, 1:]) and torch.equal(cls_row, dsd["cls_token"][0, 0] + ref[0, 0])
    sd = RO.make_roma_state_dict(0)
    m.load_state_dict(sd)
    f = m._gp_features(8, 10, torch.device("cpu"))
    ref = torch.cos(8 * math.pi * RO._conv(sd, "decoder.gps.16.pos_conv", RO.grid_coords(1, 8, 10)))[0].permute(1, 2, 0).reshape(80, -1)
    assert torch.equal(f, ref)


def test_padding_size():
    """tools/__init__.py:202-218: pad to the aspect ratio w / h, never shrink"""
    assert get_padding_size(torch.zeros(1, 3, 480, 640), 672, 896) == (640, 480, 0, 0, 0, 0)            # already 4:3
    assert get_padding_size(torch.zeros(1, 3, 100, 150), 128, 160) == (150, 100, 0, 0, 10, 10)
    assert get_padding_size(torch.zeros(1, 3, 300, 200), 672, 672) == (200, 300, 50, 50, 0, 0)
    ow, oh, pl, pr, pt, pb = get_padding_size(torch.zeros(1, 3, 333, 1001), 672, 896)
    assert (ow + pl + pr) / (oh + pt + pb) <= 896 / 672 + 1e-2 and pl == pr == 0 and pt + pb == int(1001 / (896 / 672)) - 333
