"""Synthetic, match-rich gim_loftr workloads for tests and bench.py (no checkpoint ships with the reference).

Random-init LoFTR weights collapse the features: the whole pipeline finds ~1 coarse match per pair and the
fine level idles (SURVEY 8c caveat).  Two ingredients make a *random* network behave like a trained one on
synthetic pairs, with no hook inside the product forward:

  * `calibrate_(model)` -- "trained-like" seeded weights: every BatchNorm's running statistics are set to the
    batch statistics of a small textured calibration batch flowing through the backbone (what a training run
    converges them to), and the last BN of every Bottleneck is damped (gamma *= res_gain, the usual
    zero-init-residual practice) so that the network is smooth instead of chaotic.  Plain torch CPU ops on the
    parameter containers of `gim_amd.loftr.LoFTR` -- weight synthesis, not part of the data path.
  * `textured_pairs(...)` -- image1 contains a shifted copy of image0 (shift = whole coarse cells) in a chosen
    fraction of the frame plus pixel noise, so corresponding cells see (almost) the same receptive field.

With res_gain 0.15 and noise 0.02 the reference's own arithmetic (CPU oracle) recovers ~85 % of the planted cells
with confidences spread over (0.2, 1]; `frac` tunes the match count (the reference's gim_loftr dumps average 1480
matches per pair).

Reference lines mirrored by the calibration walk: networks/loftr/backbone/resnet.py:109-126 (Bottleneck),
:230-235 (encoder), :306-329 (FPN).
"""
import torch
import torch.nn.functional as F


def textured(n, h, w, gen):
    """[n,3,h,w] in [0,1]: band-limited noise at 4 octaves (a texture whose patches are distinctive)."""
    out = torch.zeros(n, 3, h, w)
    for s, a in ((16, 0.15), (8, 0.3), (4, 0.5)):
        r = torch.rand(n, 3, (h + s - 1) // s + 1, (w + s - 1) // s + 1, generator=gen)
        out += a * F.interpolate(r, scale_factor=s, mode="bilinear", align_corners=False)[:, :, :h, :w]
    out += 0.25 * torch.rand(n, 3, h, w, generator=gen)
    return (out / 1.2).clamp(0, 1)


def textured_pairs(n, h, w, seed=0, shift=(16, 24), noise=0.02, frac=1.0):
    """(color0, color1) fp32 [n,3,h,w].  The left `frac` of image1's columns is image0 displaced by `shift`
    (dy, dx) pixels (multiples of 8 = whole coarse cells) plus N(0, noise^2); the rest is unrelated texture."""
    g = torch.Generator().manual_seed(seed)
    dy, dx = shift
    canvas = textured(n, h + dy, w + dx, g)
    c0 = canvas[:, :, :h, :w].contiguous()
    c1 = canvas[:, :, dy:h + dy, dx:w + dx].clone()
    if frac < 1.0:
        cut = int(round(w * frac / 8.0)) * 8
        c1[:, :, :, cut:] = textured(n, h, w - cut, g)
    if noise > 0:
        c1 = (c1 + noise * torch.randn(c1.shape, generator=g)).clamp(0, 1)
    return c0, c1.contiguous()


def _bn_calibrate(bn, x):
    m = x.mean((0, 2, 3))
    v = x.var((0, 2, 3), unbiased=False)
    bn.running_mean.copy_(m)
    bn.running_var.copy_(v)
    return F.batch_norm(x, m, v, bn.weight, bn.bias, False, 0.0, bn.eps)


@torch.no_grad()
def calibrate_(model, seed=0, res_gain=0.15, size=(128, 160), affine_jitter=True):
    """In place on a CPU `gim_amd.loftr.LoFTR` (or anything with the same `.backbone` containers).  Returns model."""
    g = torch.Generator().manual_seed(1000 + seed)
    bb = model.backbone
    enc = bb.encode
    if affine_jitter:  # affine parameters away from the identity, so that BN/LN folding mistakes show up
        for m in model.modules():
            if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.LayerNorm)):
                m.weight.copy_(0.75 + 0.5 * torch.rand(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
    for li in (1, 2, 3):
        for blk in getattr(enc, f"layer{li}"):
            blk.bn3.weight.mul_(res_gain)
    c0, c1 = textured_pairs(2, size[0], size[1], seed=77 + seed, noise=0.0)
    x = torch.cat([c0, c1])
    x = F.relu(_bn_calibrate(enc.bn1, F.conv2d(x, enc.conv1.weight, stride=2, padding=3)))
    feats = []
    for li in (1, 2, 3):
        for blk in getattr(enc, f"layer{li}"):
            o = F.relu(_bn_calibrate(blk.bn1, F.conv2d(x, blk.conv1.weight)))
            o = F.relu(_bn_calibrate(blk.bn2, F.conv2d(o, blk.conv2.weight, stride=blk.stride, padding=1)))
            o = _bn_calibrate(blk.bn3, F.conv2d(o, blk.conv3.weight))
            idn = x
            if blk.downsample is not None:
                idn = _bn_calibrate(blk.downsample[1], F.conv2d(x, blk.downsample[0].weight, stride=blk.stride))
            x = F.relu(o + idn)
        feats.append(x)
    x1, x2, x3 = feats
    up = lambda t: F.interpolate(t, scale_factor=2.0, mode="bilinear", align_corners=True)  # noqa: E731
    x3o = F.conv2d(x3, bb.layer3_outconv.weight)
    x2o = F.conv2d(x2, bb.layer2_outconv.weight) + up(x3o)
    t = F.leaky_relu(_bn_calibrate(bb.layer2_outconv2[1], F.conv2d(x2o, bb.layer2_outconv2[0].weight, padding=1)), 0.01)
    x2o = F.conv2d(t, bb.layer2_outconv2[3].weight, padding=1)
    x1o = F.conv2d(x1, bb.layer1_outconv.weight) + up(x2o)
    _bn_calibrate(bb.layer1_outconv2[1], F.conv2d(x1o, bb.layer1_outconv2[0].weight, padding=1))
    if hasattr(model, "_packed"):
        model._packed = None
    return model


def synthetic_model(precision="fp16", seed=0, res_gain=0.15, **cfg_over):
    """Seeded, calibrated `gim_amd.loftr.LoFTR` on the CPU + its reference-keyed state_dict (for the oracle)."""
    from gim_amd.loftr import LoFTR, get_cfg_defaults, lower_config
    torch.manual_seed(seed)
    cfg = lower_config(get_cfg_defaults())["loftr"]
    cfg["precision"] = precision
    cfg.update(cfg_over)
    model = calibrate_(LoFTR(cfg).eval(), seed=seed, res_gain=res_gain)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return model, sd
