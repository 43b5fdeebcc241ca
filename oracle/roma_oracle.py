"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the gim_roma path (kernels R1-R8).

A functional fp32 torch restatement of what the reference computes for `--model gim_roma`
(`RoMa(img_size=[672])` + `RegressionMatcher.match`, symmetric, upsampled predictions):

    VGG19-BN pyramid {1,2,4,8}               networks/roma/roma.py:139-152 (torchvision vgg19_bn.features[:40])
    DINOv2 ViT-L/14 patch tokens -> scale 16 networks/roma/roma.py:583-633, networks/roma/dino.py:322-540
    proj (1x1 conv + BN) per scale           roma.py:1220-1234
    GP.forward (CosKernel, no_cov, 512-d)    roma.py:110-136
    TransformerDecoder (5 ViT blocks, 8 heads x 128, 64x64+1 classes)   roma.py:952-1015
    cls_to_flow_refine                       roma.py:1092-1121
    ConvRefiner.forward                      roma.py:529-580
    local_correlation                        roma.py:1026-1089 (same arithmetic as DKM's)
    Decoder.forward                          roma.py:220-353
    RegressionMatcher.forward_symmetric / match   roma.py:739-755, 816-917

The DINOv2 weights are NOT part of the model's state_dict (`self.dinov2_vitl14 = [dinov2_vitl14]`, roma.py:612): the
reference downloads them in the constructor; here they are a second, separately supplied dict (`dino_sd`).

Parity: PINNED -- `oracle/make_golden_roma.py` builds the reference's own RoMa (xformers / torchvision stand-ins and a
patched `torch.hub.load_state_dict_from_url` in `oracle/ref_shims.py::install_roma`), loads the seeded weights and records
the reference's outputs under tests/golden/roma_*.npz.  Only `tests/` may import this module.
"""
import math

import torch
import torch.nn.functional as F

import dkm_oracle as _DO
from dkm_oracle import BN_EPS, _bn, _conv, cos_kernel, grid_coords, kde, local_correlation  # noqa: F401  (shared arithmetic)

VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M"]      # vgg19_bn.features[:40]
REFINER = {"16": (512, 128, 7), "8": (512, 64, 3), "4": (256, 32, 2), "2": (64, 16, None), "1": (9, 6, None)}
PROJ = {"16": (1024, 512), "8": (512, 512), "4": (256, 256), "2": (128, 64), "1": (64, 9)}
GP_DIM, DEC_DIM, DEC_HEADS, DEC_BLOCKS, CLS_RES, HIDDEN_BLOCKS = 512, 1024, 8, 5, 64, 8
VIT = dict(dim=1024, depth=24, heads=16, patch=14, grid=37)                                 # vit_large, img_size 518


def _refiner_dims(scale):
    c, e, r = REFINER[scale]
    in_dim = 2 * c + e + ((2 * r + 1) ** 2 if r else 0)
    return in_dim, {"2": 128 + 16, "1": 24}.get(scale, in_dim)


# ------------------------------------------------------------------------------------------------ parameters
def roma_param_spec():
    spec = {}

    def conv(name, ci, co, k, groups=1):
        spec[name + ".weight"] = (co, ci // groups, k, k)
        spec[name + ".bias"] = (co,)

    def bn(name, c):
        for s in ("weight", "bias", "running_mean", "running_var"):
            spec[f"{name}.{s}"] = (c,)
        spec[name + ".num_batches_tracked"] = ()

    def lin(name, i, o, bias=True):
        spec[name + ".weight"] = (o, i)
        if bias:
            spec[name + ".bias"] = (o,)

    idx, ci = 0, 3
    for v in VGG_CFG:
        if v == "M":
            idx += 1
            continue
        conv(f"encoder.cnn.layers.{idx}", ci, v, 3)
        bn(f"encoder.cnn.layers.{idx + 1}", v)
        idx, ci = idx + 3, v
    d = "decoder."
    for i in range(DEC_BLOCKS):
        b = f"{d}embedding_decoder.blocks.{i}."
        for nm in ("norm1", "norm2"):
            spec[b + nm + ".weight"] = (DEC_DIM,)
            spec[b + nm + ".bias"] = (DEC_DIM,)
        lin(b + "attn.qkv", DEC_DIM, 3 * DEC_DIM, bias=False)
        lin(b + "attn.proj", DEC_DIM, DEC_DIM)
        lin(b + "mlp.fc1", DEC_DIM, 4 * DEC_DIM)
        lin(b + "mlp.fc2", 4 * DEC_DIM, DEC_DIM)
    lin(d + "embedding_decoder.to_out", DEC_DIM, CLS_RES ** 2 + 1)
    conv(d + "gps.16.pos_conv", 2, GP_DIM, 1)
    for s, (ci_, co_) in PROJ.items():
        conv(f"{d}proj.{s}.0", ci_, co_, 1)
        bn(f"{d}proj.{s}.1", co_)
    for s in REFINER:
        in_dim, hid = _refiner_dims(s)
        r = f"{d}conv_refiner.{s}."
        for nm, c_in in [("block1", in_dim)] + [(f"hidden_blocks.{i}", hid) for i in range(HIDDEN_BLOCKS)]:
            conv(f"{r}{nm}.0", c_in, hid, 5, groups=c_in)
            bn(f"{r}{nm}.1", hid)
            conv(f"{r}{nm}.3", hid, hid, 1)
        conv(r + "out_conv", hid, 3, 1)
        conv(r + "disp_emb", 2, REFINER[s][1], 1)
    return spec


def dino_param_spec():
    D, n = VIT["dim"], VIT["grid"] ** 2
    spec = {"cls_token": (1, 1, D), "pos_embed": (1, n + 1, D), "mask_token": (1, D),
            "patch_embed.proj.weight": (D, 3, 14, 14), "patch_embed.proj.bias": (D,), "norm.weight": (D,), "norm.bias": (D,)}
    for i in range(VIT["depth"]):
        b = f"blocks.{i}."
        for nm in ("norm1", "norm2"):
            spec[b + nm + ".weight"] = (D,)
            spec[b + nm + ".bias"] = (D,)
        spec[b + "attn.qkv.weight"], spec[b + "attn.qkv.bias"] = (3 * D, D), (3 * D,)
        spec[b + "attn.proj.weight"], spec[b + "attn.proj.bias"] = (D, D), (D,)
        spec[b + "ls1.gamma"], spec[b + "ls2.gamma"] = (D,), (D,)
        spec[b + "mlp.fc1.weight"], spec[b + "mlp.fc1.bias"] = (4 * D, D), (4 * D,)
        spec[b + "mlp.fc2.weight"], spec[b + "mlp.fc2.bias"] = (D, 4 * D), (D,)
    return spec


def _fill(spec, g, small=()):
    sd = {}
    for k, shp in spec.items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(0, dtype=torch.long)
        elif k.endswith("running_var"):
            sd[k] = 0.8 + 0.4 * torch.rand(shp, generator=g)
        elif k.endswith("running_mean"):
            sd[k] = 0.05 * torch.randn(shp, generator=g)
        elif k.endswith("gamma"):
            sd[k] = 0.2 + 0.05 * torch.randn(shp, generator=g)
        elif len(shp) == 1 and k.endswith("weight"):
            sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            sd[k] = 0.02 * torch.randn(shp, generator=g)
        elif k in ("cls_token", "pos_embed", "mask_token"):
            sd[k] = 0.02 * torch.randn(shp, generator=g)
        else:
            sd[k] = torch.randn(shp, generator=g) * (1.0 / math.sqrt(math.prod(shp[1:])))
        if any(k.startswith(p) for p in small):
            sd[k] = sd[k] * 0.05
    return sd


def make_roma_state_dict(seed=0, _g=None):
    """the 603 tensors of the module's own state_dict (cheap); same values as make_state_dicts(seed)[0]"""
    g = _g or torch.Generator().manual_seed(seed)
    roma = _fill(roma_param_spec(), g, small=tuple(f"decoder.conv_refiner.{s}.out_conv" for s in REFINER))
    roma["decoder.gps.16.pos_conv.weight"] = roma["decoder.gps.16.pos_conv.weight"] * 0.25
    return roma


def make_state_dicts(seed=0):
    """(roma_sd, dino_sd): seeded stand-in weights; refiner outputs scaled down so the flow stays in range.  The DINOv2
    part continues the same generator stream (304 M values: ~40 s on a few cores)."""
    g = torch.Generator().manual_seed(seed)
    roma = make_roma_state_dict(seed, g)
    dino = _fill(dino_param_spec(), g)
    return roma, dino


# ------------------------------------------------------------------------------------------------ encoders
def vgg_pyramid(sd, x, p="encoder.cnn.layers."):
    """VGG19.forward (roma.py:144-152): features before each max-pool -> {1: 64, 2: 128, 4: 256, 8: 512 channels}"""
    feats, scale, idx = {}, 1, 0
    for v in VGG_CFG:
        if v == "M":
            feats[scale] = x
            scale *= 2
            x = F.max_pool2d(x, 2, 2)
            idx += 1
            continue
        x = F.relu(_bn(sd, f"{p}{idx + 1}", _conv(sd, f"{p}{idx}", x, 1, 1)))
        idx += 3
    return feats


def _ln(sd, name, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def vit_block(sd, p, x, heads, eps, layerscale):
    """dino.py:143-168 in eval mode; attention = softmax(q k^T / sqrt(d)) v (MemEffAttention / Attention, dino.py:74-91,303-319)"""
    B, N, C = x.shape
    qkv = _lin(sd, p + "attn.qkv", _ln(sd, p + "norm1", x, eps)).reshape(B, N, 3, heads, C // heads)
    q, k, v = (t.transpose(1, 2) for t in qkv.unbind(2))
    a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, C)
    a = _lin(sd, p + "attn.proj", a)
    x = x + (a * sd[p + "ls1.gamma"] if layerscale else a)
    m = _lin(sd, p + "mlp.fc2", F.gelu(_lin(sd, p + "mlp.fc1", _ln(sd, p + "norm2", x, eps))))
    return x + (m * sd[p + "ls2.gamma"] if layerscale else m)


def dino_pos_embed(dsd, h, w):
    """interpolate_pos_encoding (dino.py:457-488): bicubic resize of the 37x37 grid with the +0.1 scale-factor trick.
    The reference passes (w, h) = x.shape[2:] = (H, W) of the image, so its `w0` is the row count."""
    pe = dsd["pos_embed"].float()
    N = pe.shape[1] - 1
    g = int(math.sqrt(N))
    h0, w0 = h // 14, w // 14
    if h0 * w0 == N and h == w:
        return pe
    patch = F.interpolate(pe[:, 1:].reshape(1, g, g, -1).permute(0, 3, 1, 2),
                          scale_factor=((h0 + 0.1) / g, (w0 + 0.1) / g), mode="bicubic")
    assert patch.shape[-2:] == (h0, w0)
    return torch.cat((pe[:, :1], patch.permute(0, 2, 3, 1).reshape(1, h0 * w0, -1)), 1)


def dino_patch_features(dsd, x):
    """forward_features(x)['x_norm_patchtokens'] -> [B, 1024, H/14, W/14] (dino.py:490-540, roma.py:624-631)"""
    B, _, H, W = x.shape
    t = F.conv2d(x, dsd["patch_embed.proj.weight"], dsd["patch_embed.proj.bias"], stride=14).flatten(2).transpose(1, 2)
    t = torch.cat((dsd["cls_token"].expand(B, -1, -1), t), 1) + dino_pos_embed(dsd, H, W)
    for i in range(VIT["depth"]):
        t = vit_block(dsd, f"blocks.{i}.", t, VIT["heads"], 1e-6, True)
    t = _ln(dsd, "norm", t, 1e-6)[:, 1:]
    return t.permute(0, 2, 1).reshape(B, VIT["dim"], H // 14, W // 14)


def encoder(sd, dsd, x, upsample=False):
    """CNNandDinov2.forward (roma.py:617-633): VGG pyramid; scale 16 = DINOv2 patch tokens, only in the low-res pass"""
    feats = vgg_pyramid(sd, x)
    if not upsample:
        feats[16] = dino_patch_features(dsd, x)
    return feats


# ------------------------------------------------------------------------------------------------ decoder
def gp_forward(sd, x, y, sigma_noise=0.1):
    """GP.forward, no_cov (roma.py:110-136); 512-d Fourier features"""
    b, c, h1, w1 = x.shape
    _, _, h2, w2 = y.shape
    f = torch.cos(8 * math.pi * _conv(sd, "decoder.gps.16.pos_conv", grid_coords(b, h2, w2)))
    xr, yr, fr = (t.float().flatten(2).transpose(1, 2) for t in (x, y, f))
    if _DO.GP_FP64:   # test switch: the formula in fp64 (see dkm_oracle.GP_FP64)
        return _DO.gp_posterior_fp64(xr, yr, fr, sigma_noise).transpose(1, 2).reshape(b, -1, h1, w1)
    K_inv = torch.linalg.inv(cos_kernel(yr, yr) + sigma_noise * torch.eye(h2 * w2)[None])
    mu = cos_kernel(xr, yr).matmul(K_inv.matmul(fr))
    return mu.transpose(1, 2).reshape(b, -1, h1, w1)


def transformer_decoder(sd, gp_posterior, features):
    """TransformerDecoder.forward (roma.py:982-1015), pos_enc off: tokens = cat(gp, feats) -> 5 blocks -> to_out"""
    x = torch.cat((gp_posterior, features), 1)
    B, C, H, W = x.shape
    t = x.reshape(B, C, H * W).permute(0, 2, 1)
    for i in range(DEC_BLOCKS):
        t = vit_block(sd, f"decoder.embedding_decoder.blocks.{i}.", t, DEC_HEADS, 1e-5, False)
    out = _lin(sd, "decoder.embedding_decoder.to_out", t).permute(0, 2, 1).reshape(B, CLS_RES ** 2 + 1, H, W)
    return out[:, :-1], out[:, -1:]


def cls_to_flow_refine(cls):
    """roma.py:1092-1121: softmax over the 64x64 anchor classes, arg-max and its 4-neighbourhood, weighted anchor mean"""
    B, C, H, W = cls.shape
    res = round(math.sqrt(C))
    lin = torch.linspace(-1 + 1 / res, 1 - 1 / res, steps=res)
    gy, gx = torch.meshgrid(lin, lin, indexing="ij")
    G = torch.stack([gx, gy], -1).reshape(C, 2)
    cls = cls.softmax(dim=1)
    mode = cls.max(dim=1).indices
    index = torch.stack((mode - 1, mode, mode + 1, mode - res, mode + res), 1).clamp(0, C - 1).long()
    nb = torch.gather(cls, 1, index)[..., None]
    flow = sum(nb[:, i] * G[index[:, i]] for i in range(5))
    return flow / nb.sum(1)


def conv_refiner(sd, scale, x, y, flow, scale_factor=1.0):
    """ConvRefiner.forward (roma.py:529-580) -> (displacement [b,2,h,w], certainty [b,1,h,w])"""
    p = f"decoder.conv_refiner.{scale}."
    b, c, hs, ws = x.shape
    _, _, radius = REFINER[scale]
    x_hat = F.grid_sample(y, flow.permute(0, 2, 3, 1), align_corners=False, mode="bilinear")
    emb = _conv(sd, p + "disp_emb", 40 / 32 * scale_factor * (flow - grid_coords(b, hs, ws)))
    parts = [x, x_hat, emb]
    if radius:
        parts.append(local_correlation(x, y, radius, flow))
    d = torch.cat(parts, 1)
    in_dim, hid = _refiner_dims(scale)
    for nm, ci in [("block1", in_dim)] + [(f"hidden_blocks.{i}", hid) for i in range(HIDDEN_BLOCKS)]:
        d = _conv(sd, f"{p}{nm}.0", d, 1, 2, groups=ci)
        d = F.relu(_bn(sd, f"{p}{nm}.1", d))
        d = _conv(sd, f"{p}{nm}.3", d)
    d = _conv(sd, p + "out_conv", d.float())
    return d[:, :-1], d[:, -1:]


def _proj(sd, s, x):
    return _bn(sd, f"decoder.proj.{s}.1", _conv(sd, f"decoder.proj.{s}.0", x))


def decoder(sd, f1, f2, upsample=False, flow=None, certainty=None, scale_factor=1.0):
    """Decoder.forward (roma.py:220-353) -> {scale: {'flow', 'certainty'}}"""
    scales = ["8", "4", "2", "1"] if upsample else ["16", "8", "4", "2", "1"]
    sizes = {s: f1[s].shape[-2:] for s in f1}
    h, w = sizes[1]
    b = f1[1].shape[0]
    coarsest = int(scales[0])
    if not upsample:
        flow = grid_coords(b, *sizes[coarsest])
        certainty = 0.0
    else:
        flow = F.interpolate(flow, size=sizes[coarsest], align_corners=False, mode="bilinear")
        certainty = F.interpolate(certainty, size=sizes[coarsest], align_corners=False, mode="bilinear")
    out = {}
    for s in scales:
        ins = int(s)
        a, c = _proj(sd, s, f1[ins]), _proj(sd, s, f2[ins])
        if s == "16":
            cls, certainty = transformer_decoder(sd, gp_forward(sd, a, c), a)
            flow = cls_to_flow_refine(cls).permute(0, 3, 1, 2)
            out.setdefault(ins, {}).update({"gm_cls": cls, "gm_certainty": certainty})
        dflow, dcert = conv_refiner(sd, s, a, c, flow, scale_factor)
        flow = flow + ins * torch.stack((dflow[:, 0].float() / (4 * w), dflow[:, 1].float() / (4 * h)), 1)
        certainty = certainty + dcert
        out.setdefault(ins, {}).update({"certainty": certainty, "flow": flow})
        if s != "1":
            flow = F.interpolate(flow, size=sizes[ins // 2], mode="bilinear")
            certainty = F.interpolate(certainty, size=sizes[ins // 2], mode="bilinear")
    return out


def forward_symmetric(sd, dsd, im_A, im_B, upsample=False, corresps=None, scale_factor=1.0):
    """RegressionMatcher.forward_symmetric (roma.py:739-755)"""
    pyr = encoder(sd, dsd, torch.cat((im_A, im_B)), upsample)
    swapped = {s: torch.cat((f.chunk(2)[1], f.chunk(2)[0])) for s, f in pyr.items()}
    kw = {} if corresps is None else {"flow": corresps["flow"], "certainty": corresps["certainty"]}
    return decoder(sd, pyr, swapped, upsample=upsample, scale_factor=scale_factor, **kw)


def match(sd, dsd, im_A, im_B, h_resized, w_resized, upsample_res=None, attenuate_cert=True):
    """RegressionMatcher.match, symmetric, non-batched (roma.py:816-917) -> warp [H, 2W, 4], certainty [H, 2W]"""
    def up(t, size):
        return F.interpolate(t, size=size, mode="bilinear", align_corners=False)

    hs, ws = h_resized, w_resized
    cor = forward_symmetric(sd, dsd, up(im_A, (hs, ws)), up(im_B, (hs, ws)))
    if upsample_res is not None:
        hs, ws = upsample_res
    low = 0.0
    if attenuate_cert:
        low = up(cor[16]["certainty"], (hs, ws))
        low = 0.5 * low * (low < 0)
    if upsample_res is not None:
        sf = math.sqrt(upsample_res[0] * upsample_res[1] / (w_resized * h_resized))
        cor = forward_symmetric(sd, dsd, up(im_A, (hs, ws)), up(im_B, (hs, ws)), upsample=True, corresps=cor[1], scale_factor=sf)
    a2b = cor[1]["flow"].permute(0, 2, 3, 1)
    cert = (cor[1]["certainty"] - low).sigmoid()
    qc = grid_coords(1, hs, ws).permute(0, 2, 3, 1)
    wrong = (a2b.abs() > 1).sum(dim=-1) > 0
    cert[wrong[:, None]] = 0

    def black(im):
        m = (im[0, 0] < 0.03125) & (im[0, 1] < 0.03125) & (im[0, 2] < 0.03125)
        return F.interpolate(m.float()[None, None], size=(hs, ws), mode="nearest").bool()
    cert[torch.cat((black(im_A), black(im_B)), 0)] = 0
    a2b = torch.clamp(a2b, -1, 1)
    A, Bm = a2b.chunk(2)
    warp = torch.cat((torch.cat((qc, A), -1), torch.cat((Bm, qc), -1)), 2)
    cert = torch.cat(cert.chunk(2), 3)
    return warp[0], cert[0, 0]


def kde_half(x, std=0.1):
    """roma.py:1018-1023: the KDE of sample() runs on fp16 coordinates"""
    x = x.half().float()   # CPU: emulate the half rounding of the inputs; cdist itself is not available in half on CPU
    return (-torch.cdist(x, x) ** 2 / (2 * std ** 2)).exp().sum(-1)
