"""gim_bneck_tail128 (layer-2 Bottleneck tail: conv3 1x1 + bn3 + identity + relu, fused with the next block's conv1 1x1 + bn1 + relu;
weights streamed through LDS two chunks ahead; resnet.py:109-126) against torch fp32 convolutions with the kernel's rounding points,
both 16-bit flavours, both widths of the trailing conv1, and the whole backbone with / without the fusion."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

KINDS = [torch.bfloat16, torch.float16]
KIDS = ["bf16", "fp16"]


def _blocks(seed, n1, planes=128):
    from gim_amd.loftr.loftr import _Bottleneck
    torch.manual_seed(seed)
    blk, nxt = _Bottleneck(4 * planes, planes, 1, None), _Bottleneck(4 * planes, n1, 1, None)
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


def _ref(blk, nxt, t2, res, tdt, plain_next=False):
    """plain_next: the next convolution has no BatchNorm and no activation (the FPN's layer3_outconv behind layer 3's last block)"""
    from gim_amd.packing import fold_bn
    r = lambda t: t.to(tdt).float()  # noqa: E731
    bn = lambda m: (m.weight, m.bias, m.running_mean, m.running_var, m.eps)  # noqa: E731
    w3, b3 = fold_bn(blk.conv3.weight, bn(blk.bn3))
    w1, b1 = fold_bn(nxt.conv1.weight, None if plain_next else bn(nxt.bn1))
    x = F.relu(F.conv2d(t2, r(w3), b3) + res)
    t1 = F.conv2d(r(x), r(w1), b1)
    return x, (t1 if plain_next else F.relu(t1))


@pytest.mark.parametrize("tdt", KINDS, ids=KIDS)
@pytest.mark.parametrize("B,H,W,planes,n1,plain", [(1, 8, 32, 128, 128, False), (2, 24, 64, 128, 128, False), (3, 16, 80, 128, 128, False),
                                                   (1, 16, 48, 128, 256, False), (1, 8, 32, 256, 256, False), (2, 16, 40, 256, 256, False),
                                                   (1, 16, 32, 256, 256, True), (16, 60, 80, 256, 256, False)])
def test_bneck_tail_matches_reference(B, H, W, planes, n1, plain, tdt):
    """planes 128 = layer 2 (chunks of 64 channels), 256 = layer 3 (chunks of 32); plain = layer3_outconv as the next convolution
    (no BatchNorm, no activation) with x' not stored.  16 x 60 x 80 = the benchmark's layer 3 as ONE launch: 300 tiles of 256 rows, the
    4-wave workgroups (smaller launches take the 8-wave ones: bneck_tail.hip, tail_entry)"""
    from gim_amd import ops
    from gim_amd.packing import pack_bneck_tail
    blk, nxt = _blocks(H + W, n1, planes)
    g = torch.Generator().manual_seed(B * H + n1)
    t2 = F.relu(torch.randn(B, planes, H, W, generator=g)).to(tdt)        # post-ReLU like the real conv2 output
    res = torch.randn(B, 4 * planes, H, W, generator=g).to(tdt)
    with torch.no_grad():
        x_ref, t1_ref = _ref(blk, nxt, t2.float(), res.float(), tdt, plain)
    pk = pack_bneck_tail(blk, nxt.conv1, None if plain else nxt.bn1, "cuda", tdt)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()  # noqa: E731
    xo, t1n = ops.bneck_tail(nhwc(t2), nhwc(res), pk, ops.ACT_NONE if plain else ops.ACT_RELU, store_x=not plain)
    torch.cuda.synchronize()
    k = 1.0 if tdt == torch.bfloat16 else 0.25
    assert plain == (xo is None)
    for got, ref, nm in ((xo, x_ref, "x'"), (t1n, t1_ref, "t1'")):
        if got is None:
            continue
        got = got.float().cpu().permute(0, 3, 1, 2)
        assert torch.isfinite(got).all(), nm
        sc = ref.abs().max().item()
        err = (got - ref).abs()
        assert err.max().item() < 2e-2 * k * sc and err.mean().item() < 2e-3 * k * sc, (nm, err.max().item() / sc, err.mean().item() / sc)


@pytest.mark.parametrize("planes", [128, 256])
def test_bneck_tail_many_tiles_and_repeatability(planes):
    """more workgroups than CUs (the weight stream is shared through L2) and bitwise-identical results on a second launch (the
    hand-counted vmcnt waits leave loads in flight across barriers: a miscount shows as run-to-run differences)"""
    from gim_amd import ops
    from gim_amd.packing import pack_bneck_tail
    n1 = planes
    blk, nxt = _blocks(7, n1, planes)
    g = torch.Generator().manual_seed(3)
    t2 = F.relu(torch.randn(4, 120, 160, planes, generator=g)).to(torch.bfloat16).cuda()    # 300 workgroups
    res = torch.randn(4, 120, 160, 4 * planes, generator=g).to(torch.bfloat16).cuda()
    pk = pack_bneck_tail(blk, nxt.conv1, nxt.bn1, "cuda")
    a = ops.bneck_tail(t2, res, pk)
    b = ops.bneck_tail(t2, res, pk)
    c = ops.bneck_tail(t2[:1].contiguous(), res[:1].contiguous(), pk)
    torch.cuda.synchronize()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.equal(a[0][:1], c[0]) and torch.equal(a[1][:1], c[1])                  # rows are independent of the tile they sit in
    with torch.no_grad():
        x_ref, t1_ref = _ref(blk, nxt, t2[:1].float().cpu().permute(0, 3, 1, 2), res[:1].float().cpu().permute(0, 3, 1, 2), torch.bfloat16)
    sc = t1_ref.abs().max().item()
    assert (c[1].float().cpu().permute(0, 3, 1, 2) - t1_ref).abs().max().item() < 2e-2 * sc


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_backbone_with_and_without_tail_fusion(precision):
    from tools import synth_loftr as S
    model, _ = S.synthetic_model(precision)
    model = model.cuda()
    c0, c1 = S.textured_pairs(2, 128, 256, seed=4)           # quarter resolution 32 x 64: 4 images x 2048 rows = 32 tiles
    outs = {}
    for fused in (True, False):
        model.bneck_tail = fused
        model.debug = {}
        d = {"image0": c0[:, :1].cuda(), "image1": c1[:, :1].cuda(), "color0": c0.cuda(), "color1": c1.cuda()}
        model(d)
        outs[fused] = (model.debug["c0"].float().cpu(), model.debug["f0"].float().cpu())
        model.debug = None
    model.bneck_tail = True
    k = 1.0 if precision == "bf16" else 0.25
    for a, b in zip(outs[True], outs[False]):
        sc = b.abs().max().item()
        assert (a - b).abs().mean().item() < 3e-3 * k * sc and (a - b).abs().max().item() < 6e-2 * k * sc


def _blocks_ds(seed):
    """layer 2's first block (stride 2, 256 -> 128 -> 512 with a downsample branch) and the block whose conv1 follows it"""
    from gim_amd.loftr.loftr import _Bottleneck, _conv
    torch.manual_seed(seed)
    ds = torch.nn.Sequential(_conv(256, 512, 1, 2), torch.nn.BatchNorm2d(512))
    blk, nxt = _Bottleneck(256, 128, 2, ds), _Bottleneck(512, 128, 1, None)
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


@pytest.mark.parametrize("tdt", KINDS, ids=KIDS)
@pytest.mark.parametrize("B,Hin,Win", [(1, 16, 64), (2, 48, 128), (3, 32, 160), (1, 31, 63), (4, 240, 320)],
                         ids=["1x16x64", "2x48x128", "3x32x160", "odd-31x63", "4x240x320-many-tiles"])
def test_bneck_tail_ds_matches_reference(B, Hin, Win, tdt):
    """gim_bneck_tail128_ds (round 5): conv3 + bn3 + the stride-2 downsample branch (as extra K) + relu + the next conv1, against torch fp32
    convolutions with the kernel's rounding points (resnet.py:109-126 with `downsample`); odd input sizes read pixel (2y, 2x) of a
    (2 Ho - 1)-row map; the large case runs more workgroups than CUs, twice, bit-identically (counted waits)."""
    from gim_amd import ops
    from gim_amd.packing import fold_bn, pack_bneck_tail
    blk, nxt = _blocks_ds(Hin + Win)
    Ho, Wo = (Hin - 1) // 2 + 1, (Win - 1) // 2 + 1
    if (B * Ho * Wo) % 256:
        pytest.skip("row count not a multiple of 256: the engine keeps the two launches there")
    g = torch.Generator().manual_seed(B * Hin + 5)
    t2 = F.relu(torch.randn(B, 128, Ho, Wo, generator=g)).to(tdt)
    xin = F.relu(torch.randn(B, 256, Hin, Win, generator=g)).to(tdt)
    r = lambda t: t.to(tdt).float()  # noqa: E731
    bn = lambda m: (m.weight, m.bias, m.running_mean, m.running_var, m.eps)  # noqa: E731
    with torch.no_grad():
        w3, b3 = fold_bn(blk.conv3.weight, bn(blk.bn3))
        wd, bd = fold_bn(blk.downsample[0].weight, bn(blk.downsample[1]))
        w1, b1 = fold_bn(nxt.conv1.weight, bn(nxt.bn1))
        x_ref = F.relu(F.conv2d(t2.float(), r(w3), b3) + F.conv2d(xin.float(), r(wd), bd, stride=2))
        t1_ref = F.relu(F.conv2d(r(x_ref), r(w1), b1))
    pk = pack_bneck_tail(blk, nxt.conv1, nxt.bn1, "cuda", tdt, ds=True)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()  # noqa: E731
    a_t2, a_x = nhwc(t2), nhwc(xin)
    xo, t1n = ops.bneck_tail_ds(a_t2, a_x, pk)
    xo2, t1n2 = ops.bneck_tail_ds(a_t2, a_x, pk)
    torch.cuda.synchronize()
    assert torch.equal(xo, xo2) and torch.equal(t1n, t1n2)
    k = 1.0 if tdt == torch.bfloat16 else 0.25
    for got, ref, nm in ((xo, x_ref, "x'"), (t1n, t1_ref, "t1'")):
        got = got.float().cpu().permute(0, 3, 1, 2)
        assert torch.isfinite(got).all(), nm
        sc = ref.abs().max().item()
        err = (got - ref).abs()
        assert err.max().item() < 2e-2 * k * sc and err.mean().item() < 2e-3 * k * sc, (nm, err.max().item() / sc, err.mean().item() / sc)


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_backbone_with_and_without_tail_ds(precision):
    """the whole backbone with layer 2's downsample branch inside the tail kernel against the three-launch form of that block"""
    from tools import synth_loftr as S
    model, _ = S.synthetic_model(precision)
    model = model.cuda()
    c0, c1 = S.textured_pairs(2, 128, 256, seed=4)
    outs = {}
    for fused in (True, False):
        model.bneck_tail_ds = fused
        model.debug = {}
        d = {"image0": c0[:, :1].cuda(), "image1": c1[:, :1].cuda(), "color0": c0.cuda(), "color1": c1.cuda()}
        model(d)
        outs[fused] = (model.debug["c0"].float().cpu(), model.debug["f0"].float().cpu())
        model.debug = None
    model.bneck_tail_ds = True
    k = 1.0 if precision == "bf16" else 0.25
    for a, b in zip(outs[True], outs[False]):
        sc = b.abs().max().item()
        assert (a - b).abs().mean().item() < 3e-3 * k * sc and (a - b).abs().max().item() < 6e-2 * k * sc
