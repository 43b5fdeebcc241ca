"""gim_bneck64_fused (conv2 3x3 -> conv3 1x1 + identity -> next block's conv1, chained through MFMA accumulators with K-permuted
weights; resnet.py:109-126) against torch fp32 convolutions with the kernel's rounding points, incl. image-border tiles (zero
padding of the 3x3), several images, and the variant without the trailing conv1."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _blocks(seed, next_planes=64):
    from gim_amd.loftr.loftr import _Bottleneck
    torch.manual_seed(seed)
    blk, nxt = _Bottleneck(256, 64, 1, None), _Bottleneck(256, next_planes, 1, None)
    with torch.no_grad():
        for m in list(blk.modules()) + list(nxt.modules()):
            if isinstance(m, torch.nn.Conv2d):
                torch.nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(0.5 + torch.rand_like(m.weight))
                m.bias.copy_(0.2 * torch.randn_like(m.bias))
                m.running_mean.copy_(0.2 * torch.randn_like(m.running_mean))
                m.running_var.copy_(0.5 + torch.rand_like(m.running_var))
    return blk.eval(), nxt.eval()


KINDS = [torch.bfloat16, torch.float16]   # the two 16-bit flavours of the kernel (csrc/gim_common.h)
KIDS = ["bf16", "fp16"]


def _ref(blk, nxt, t1, res, tdt=torch.bfloat16):
    """fp32 on bf16-rounded operands: folded-BN weights rounded to bf16, t2 and x' rounded to bf16 where the kernel packs them"""
    from gim_amd.packing import fold_bn
    bf = lambda t: t.to(tdt).float()  # noqa: E731
    bn = lambda m: (m.weight, m.bias, m.running_mean, m.running_var, m.eps)  # noqa: E731
    w2, b2 = fold_bn(blk.conv2.weight, bn(blk.bn2))
    w3, b3 = fold_bn(blk.conv3.weight, bn(blk.bn3))
    w1, b1 = fold_bn(nxt.conv1.weight, bn(nxt.bn1))
    t2 = bf(F.relu(F.conv2d(t1, bf(w2), b2, padding=1)))
    x = F.relu(F.conv2d(t2, bf(w3), b3) + res)
    t1n = F.relu(F.conv2d(bf(x), bf(w1), b1))
    return x, t1n


@pytest.mark.parametrize("tdt", KINDS, ids=KIDS)
@pytest.mark.parametrize("B,H,W,with_next", [(1, 8, 32, 64), (2, 24, 64, 64), (1, 16, 96, 0), (2, 16, 64, 128)])
def test_bneck64_fused_matches_reference(B, H, W, with_next, tdt):
    """with_next: output channels of the trailing conv1 (64 = next block of layer1, 128 = layer2's first conv1, 0 = none)"""
    from gim_amd import ops
    from gim_amd.packing import pack_bneck
    blk, nxt = _blocks(H + W, with_next or 64)
    g = torch.Generator().manual_seed(B * H)
    t1 = F.relu(torch.randn(B, 64, H, W, generator=g)).to(tdt)       # post-ReLU like the real conv1 output
    res = torch.randn(B, 256, H, W, generator=g).to(tdt)
    with torch.no_grad():
        x_ref, t1n_ref = _ref(blk, nxt, t1.float(), res.float(), tdt)
    pk = pack_bneck(blk, nxt if with_next else None, "cuda", tdt)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()  # noqa: E731
    xo, t1n = ops.bneck64(nhwc(t1), nhwc(res), pk, bool(with_next))
    torch.cuda.synchronize()
    got = xo.float().cpu().permute(0, 3, 1, 2)
    sc = x_ref.abs().max().item()
    err = (got - x_ref).abs()
    k = 1.0 if tdt == torch.bfloat16 else 0.25
    assert err.max().item() < 2e-2 * k * sc and err.mean().item() < 2e-3 * k * sc, (err.max().item() / sc, err.mean().item() / sc)
    if with_next:
        got1 = t1n.float().cpu().permute(0, 3, 1, 2)
        sc1 = t1n_ref.abs().max().item()
        e1 = (got1 - t1n_ref).abs()
        assert e1.max().item() < 2e-2 * k * sc1 and e1.mean().item() < 2e-3 * k * sc1, (e1.max().item() / sc1, e1.mean().item() / sc1)
    else:
        assert t1n is None


def test_backbone_with_and_without_fused_layer1():
    """whole backbone (bf16): fused layer1 vs one launch per convolution -- same maps up to bf16 re-rounding"""
    from tools import synth_loftr as S
    model, _ = S.synthetic_model("bf16")
    model = model.cuda()
    c0, c1 = S.textured_pairs(2, 128, 192, seed=4)
    outs = {}
    for fused in (True, False):
        model.bneck_fused = fused
        model.debug = {}
        d = {"image0": c0[:, :1].cuda(), "image1": c1[:, :1].cuda(), "color0": c0.cuda(), "color1": c1.cuda()}
        model(d)
        outs[fused] = (model.debug["c0"].float().cpu(), model.debug["f0"].float().cpu())
        model.debug = None
    model.bneck_fused = True
    for a, b in zip(outs[True], outs[False]):
        sc = b.abs().max().item()
        assert (a - b).abs().mean().item() < 3e-3 * sc and (a - b).abs().max().item() < 6e-2 * sc


@pytest.mark.parametrize("tdt", KINDS, ids=KIDS)
@pytest.mark.parametrize("B,H,W", [(1, 8, 32), (2, 24, 64), (1, 16, 96)])
def test_bneck64_fused_ds_matches_reference(B, H, W, tdt):
    """gim_bneck64_fused_ds: first block of layer 1 -- the downsample branch (1x1 conv 64 -> 256 + BN, resnet.py:120-124) runs inside
    the kernel as extra K of conv3; against torch fp32 convolutions with the kernel's rounding points (the identity is NOT rounded to
    16 bits here, unlike the two-launch path: it never leaves the accumulators) and against the two-launch path"""
    from gim_amd import _lib, ops
    from gim_amd.loftr.loftr import _Bottleneck
    from gim_amd.packing import fold_bn, pack_bneck, pack_bneck_ds, pack_conv
    torch.manual_seed(H * W)
    ds = torch.nn.Sequential(torch.nn.Conv2d(64, 256, 1, bias=False), torch.nn.BatchNorm2d(256))
    blk, nxt = _Bottleneck(64, 64, 1, ds), _Bottleneck(256, 64, 1, None)
    with torch.no_grad():
        for m in list(blk.modules()) + list(nxt.modules()):
            if isinstance(m, torch.nn.Conv2d):
                torch.nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(0.5 + torch.rand_like(m.weight)); m.bias.copy_(0.2 * torch.randn_like(m.bias))
                m.running_mean.copy_(0.2 * torch.randn_like(m.running_mean)); m.running_var.copy_(0.5 + torch.rand_like(m.running_var))
    blk, nxt = blk.eval(), nxt.eval()
    g = torch.Generator().manual_seed(B * H)
    x_in = F.relu(torch.randn(B, 64, H, W, generator=g)).to(tdt)
    t1 = F.relu(torch.randn(B, 64, H, W, generator=g)).to(tdt)
    bf = lambda t: t.to(tdt).float()  # noqa: E731
    bn = lambda m: (m.weight, m.bias, m.running_mean, m.running_var, m.eps)  # noqa: E731
    with torch.no_grad():
        w2, b2 = fold_bn(blk.conv2.weight, bn(blk.bn2)); w3, b3 = fold_bn(blk.conv3.weight, bn(blk.bn3))
        wd, bd = fold_bn(blk.downsample[0].weight, bn(blk.downsample[1])); w1, b1 = fold_bn(nxt.conv1.weight, bn(nxt.bn1))
        t2 = bf(F.relu(F.conv2d(t1.float(), bf(w2), b2, padding=1)))
        x_ref = F.relu(F.conv2d(t2, bf(w3), b3) + F.conv2d(x_in.float(), bf(wd), bd))
        t1n_ref = F.relu(F.conv2d(bf(x_ref), bf(w1), b1))
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()  # noqa: E731
    xo, t1n = ops.bneck64_ds(nhwc(t1), nhwc(x_in), pack_bneck_ds(blk, nxt, "cuda", tdt))
    dt = _lib.GIM_F16 if tdt == torch.float16 else _lib.GIM_BF16
    idn = ops.conv2d(nhwc(x_in), pack_conv(blk.downsample[0].weight, bn(blk.downsample[1]), dt, "cuda"))
    xo2, t1n2 = ops.bneck64(nhwc(t1), idn, pack_bneck(blk, nxt, "cuda", tdt), True)
    torch.cuda.synchronize()
    k = 1.0 if tdt == torch.bfloat16 else 0.25
    for got, ref, two in ((xo, x_ref, xo2), (t1n, t1n_ref, t1n2)):
        gv = got.float().cpu().permute(0, 3, 1, 2)
        sc = ref.abs().max().item()
        err = (gv - ref).abs()
        assert err.max().item() < 2e-2 * k * sc and err.mean().item() < 2e-3 * k * sc, (err.max().item() / sc, err.mean().item() / sc)
        d2 = (got.float() - two.float()).abs()
        assert d2.max().item() < 4e-2 * k * sc and d2.mean().item() < 3e-3 * k * sc     # the two-launch path rounds the identity to 16 bits
