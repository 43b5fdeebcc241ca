"""TEST INFRASTRUCTURE ONLY -- CPU restatement (the *oracle*) of the reference's gim_loftr forward path.

Parity status: PINNED.  Every stage below is checked against the reference's own modules
(`/root/reference/networks/loftr/*`, imported through `oracle/ref_shims.py`) by
`oracle/make_golden.py`, which also writes the golden vectors under `tests/golden/` that
`tests/test_oracle_golden.py` replays on every run (CPU, no reference needed).  The reference ships no
checkpoint and no per-stage golden vectors, so the pins are outputs of the
reference itself run in the authoring container on seeded weights and seeded inputs.

What this file is: a functional, stateless restatement in plain fp32 torch CPU ops, driven by a
`state_dict` with the reference's key names.  It is the checker for the HIP path.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the product
(`gim_amd/`) never does.

Each function cites the reference lines it follows.
"""
import math

import torch
import torch.nn.functional as F

INF = 1e9

# effective gim_loftr configuration (`networks/loftr/config.py:3-77` lowered by `misc.py:13-16`)
DEFAULT_CONFIG = {
    "backbone_type": "ResNetFPN",
    "resolution": (8, 2),
    "fine_window_size": 5,
    "fine_concat_coarse_feat": False,
    "resnetfpn": {"initial_dim": 128, "block_dims": [64, 128, 196, 256, 512, 1024]},
    "coarse": {"d_model": 256, "nhead": 8, "layer_names": 4, "attention": "linear"},
    "match_coarse": {
        "thr": 0.2, "border_rm": 2, "match_type": "dual_softmax", "dsmax_temperature": 0.1,
        "skh_iters": 3, "skh_init_bin_score": 1.0, "skh_prefilter": False,
        "train_coarse_percent": 0.2, "train_pad_num_gt_min": 200, "sparse_spvs": False,
    },
    "fine": {"d_model": 128, "nhead": 8, "layer_names": 1, "attention": "linear"},
    "weight": None,
}


# ----------------------------------------------------------------------------------------------
# backbone: ResNet-50 (minus maxpool / layer4) + FPN   -- networks/loftr/backbone/resnet.py
# ----------------------------------------------------------------------------------------------
def _bn(sd, p, x):
    # nn.BatchNorm2d in eval mode (resnet.py:104,108,112; eps = torch default 1e-5)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


def bottleneck(sd, p, x, stride):
    """resnet.py:109-126 (Bottleneck.forward); stride sits on conv2 (resnet.py:100-105)."""
    out = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"])))
    out = F.relu(_bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], stride=stride, padding=1)))
    out = _bn(sd, p + ".bn3", F.conv2d(out, sd[p + ".conv3.weight"]))
    if (p + ".downsample.0.weight") in sd:  # resnet.py:198-202
        identity = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride))
    else:
        identity = x
    return F.relu(out + identity)


def resnet_encode(sd, p, x):
    """resnet.py:230-235: x0 = relu(bn1(conv1)); layer1 (3 blocks), layer2 (4, s2), layer3 (6, s2)."""
    x0 = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], stride=2, padding=3)))
    feats = []
    cur = x0
    for li, (nblk, stride) in enumerate(((3, 1), (4, 2), (6, 2)), start=1):
        for bi in range(nblk):
            cur = bottleneck(sd, f"{p}.layer{li}.{bi}", cur, stride if bi == 0 else 1)
        feats.append(cur)
    return feats  # x1 (1/2, 256ch), x2 (1/4, 512ch), x3 (1/8, 1024ch)


def _outconv2(sd, p, x):
    # nn.Sequential(conv3x3, BatchNorm2d, LeakyReLU(0.01), conv3x3)  resnet.py:277-289
    x = F.conv2d(x, sd[p + ".0.weight"], padding=1)
    x = F.leaky_relu(_bn(sd, p + ".1", x), 0.01)
    return F.conv2d(x, sd[p + ".3.weight"], padding=1)


def backbone(sd, x, p="backbone"):
    """ResNetFPN_8_2.forward, resnet.py:306-329.  x [B,3,H,W] -> (coarse [B,256,H/8,W/8], fine [B,128,H/2,W/2])."""
    x1, x2, x3 = resnet_encode(sd, p + ".encode", x)
    x3_out = F.conv2d(x3, sd[p + ".layer3_outconv.weight"])
    x3_out_2x = F.interpolate(x3_out, scale_factor=2.0, mode="bilinear", align_corners=True)
    x2_out = F.conv2d(x2, sd[p + ".layer2_outconv.weight"])
    x2_out = _outconv2(sd, p + ".layer2_outconv2", x2_out + x3_out_2x)
    x2_out_2x = F.interpolate(x2_out, scale_factor=2.0, mode="bilinear", align_corners=True)
    x1_out = F.conv2d(x1, sd[p + ".layer1_outconv.weight"])
    x1_out = _outconv2(sd, p + ".layer1_outconv2", x1_out + x2_out_2x)
    return x3_out, x1_out


# ----------------------------------------------------------------------------------------------
# position encoding -- networks/loftr/utils/position_encoding.py:11-43 with temp_bug_fix=False
# ----------------------------------------------------------------------------------------------
def position_encoding(d_model, h, w, max_shape=(256, 256)):
    """The *buggy* variant LoFTR is built with (loftr.py:22-24): `-log(1e4)/d_model//2` parses as
    floor(-log(1e4)/d_model) / ... = (-0.036 // 2) = -1.0, so div_term = exp(-1.0 * [0,2,4,...]).
    Positions are 1-based (cumsum of ones).  Returns pe[:, :h, :w] as [1, C, h, w]."""
    pe = torch.zeros((d_model, *max_shape))
    y_position = torch.ones(max_shape).cumsum(0).float().unsqueeze(0)
    x_position = torch.ones(max_shape).cumsum(1).float().unsqueeze(0)
    div_term = torch.exp(torch.arange(0, d_model // 2, 2).float() * (-math.log(10000.0) / d_model // 2))
    div_term = div_term[:, None, None]
    pe[0::4] = torch.sin(x_position * div_term)
    pe[1::4] = torch.cos(x_position * div_term)
    pe[2::4] = torch.sin(y_position * div_term)
    pe[3::4] = torch.cos(y_position * div_term)
    return pe.unsqueeze(0)[:, :, :h, :w]


# ----------------------------------------------------------------------------------------------
# transformer -- submodules/transformer.py, submodules/attentions.py
# ----------------------------------------------------------------------------------------------
def linear_attention(q, k, v, q_mask=None, kv_mask=None, eps=1e-6):
    """LinearAttention.forward, attentions.py:20-47.  q [N,L,H,D], k/v [N,S,H,D]."""
    Q = F.elu(q) + 1
    K = F.elu(k) + 1
    if q_mask is not None:
        Q = Q * q_mask[:, :, None, None]
    if kv_mask is not None:
        K = K * kv_mask[:, :, None, None]
        v = v * kv_mask[:, :, None, None]
    v_length = v.size(1)
    v = v / v_length
    KV = torch.einsum("nshd,nshv->nhdv", K, v)
    Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(dim=1)) + eps)
    return (torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * v_length).contiguous()


def encoder_layer(sd, p, x, source, nhead, x_mask=None, source_mask=None):
    """LoFTREncoderLayer.forward, transformer.py:35-58."""
    bs, d_model = x.size(0), x.size(2)
    dim = d_model // nhead
    q = F.linear(x, sd[p + ".q_proj.weight"]).view(bs, -1, nhead, dim)
    k = F.linear(source, sd[p + ".k_proj.weight"]).view(bs, -1, nhead, dim)
    v = F.linear(source, sd[p + ".v_proj.weight"]).view(bs, -1, nhead, dim)
    msg = linear_attention(q, k, v, x_mask, source_mask)
    msg = F.linear(msg.view(bs, -1, nhead * dim), sd[p + ".merge.weight"])
    msg = F.layer_norm(msg, (d_model,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
    msg = F.linear(torch.cat([x, msg], dim=2), sd[p + ".mlp.0.weight"])
    msg = F.linear(F.relu(msg), sd[p + ".mlp.2.weight"])
    msg = F.layer_norm(msg, (d_model,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
    return x + msg


def local_feature_transformer(sd, p, feat0, feat1, nhead, n_pairs, mask0=None, mask1=None):
    """LocalFeatureTransformer.forward, transformer.py:80-101: layer_names = ['self','cross'] * n_pairs.
    NB the cross layer updates feat0 first and feeds the *updated* feat0 to feat1's call (line 95-96)."""
    for li in range(2 * n_pairs):
        lp = f"{p}.layers.{li}"
        if li % 2 == 0:  # self
            feat0 = encoder_layer(sd, lp, feat0, feat0, nhead, mask0, mask0)
            feat1 = encoder_layer(sd, lp, feat1, feat1, nhead, mask1, mask1)
        else:  # cross
            feat0 = encoder_layer(sd, lp, feat0, feat1, nhead, mask0, mask1)
            feat1 = encoder_layer(sd, lp, feat1, feat0, nhead, mask1, mask0)
    return feat0, feat1


# ----------------------------------------------------------------------------------------------
# coarse matching -- utils/coarse_matching.py
# ----------------------------------------------------------------------------------------------
def conf_matrix_dual_softmax(feat_c0, feat_c1, temperature=0.1, mask_c0=None, mask_c1=None):
    """coarse_matching.py:108-118."""
    c = feat_c0.shape[-1]
    f0, f1 = feat_c0 / c ** .5, feat_c1 / c ** .5
    sim = torch.einsum("nlc,nsc->nls", f0, f1) / temperature
    if mask_c0 is not None:
        sim.masked_fill_(~(mask_c0[..., None] * mask_c1[:, None]).bool(), -INF)
    return F.softmax(sim, 1) * F.softmax(sim, 2)


def _mask_border(m, b, v):
    # coarse_matching.py:9-26 (all four spatial axes, both ends)
    if b <= 0:
        return
    m[:, :b] = v
    m[:, :, :b] = v
    m[:, :, :, :b] = v
    m[:, :, :, :, :b] = v
    m[:, -b:] = v
    m[:, :, -b:] = v
    m[:, :, :, -b:] = v
    m[:, :, :, :, -b:] = v


def _mask_border_with_padding(m, bd, v, p_m0, p_m1):
    # coarse_matching.py:29-44
    if bd <= 0:
        return
    m[:, :bd] = v
    m[:, :, :bd] = v
    m[:, :, :, :bd] = v
    m[:, :, :, :, :bd] = v
    h0s, w0s = p_m0.sum(1).max(-1)[0].int(), p_m0.sum(-1).max(-1)[0].int()
    h1s, w1s = p_m1.sum(1).max(-1)[0].int(), p_m1.sum(-1).max(-1)[0].int()
    for b_idx, (h0, w0, h1, w1) in enumerate(zip(h0s, w0s, h1s, w1s)):
        m[b_idx, h0 - bd:] = v
        m[b_idx, :, w0 - bd:] = v
        m[b_idx, :, :, h1 - bd:] = v
        m[b_idx, :, :, :, w1 - bd:] = v


def get_coarse_match(conf_matrix, hw0_i, hw1_i, hw0_c, hw1_c, thr=0.2, border_rm=2,
                     scale0=None, scale1=None, mask0=None, mask1=None):
    """CoarseMatching.get_coarse_match at inference (coarse_matching.py:149-259, training branch
    :199-234 omitted).  Returns the dict the reference merges into `data`."""
    n = conf_matrix.shape[0]
    h0c, w0c = hw0_c
    h1c, w1c = hw1_c
    mask = conf_matrix > thr
    mask = mask.view(n, h0c, w0c, h1c, w1c).clone()
    if mask0 is None:
        _mask_border(mask, border_rm, False)
    else:
        _mask_border_with_padding(mask, border_rm, False, mask0, mask1)
    mask = mask.view(n, h0c * w0c, h1c * w1c)
    mask = mask \
        * (conf_matrix == conf_matrix.max(dim=2, keepdim=True)[0]) \
        * (conf_matrix == conf_matrix.max(dim=1, keepdim=True)[0])
    mask_v, all_j_ids = mask.max(dim=2)
    b_ids, i_ids = torch.where(mask_v)
    j_ids = all_j_ids[b_ids, i_ids]
    mconf = conf_matrix[b_ids, i_ids, j_ids]

    scale = hw0_i[0] / hw0_c[0]
    s0 = scale * scale0[b_ids] if scale0 is not None else scale
    s1 = scale * scale1[b_ids] if scale1 is not None else scale
    mkpts0_c = torch.stack([i_ids % w0c, i_ids // w0c], dim=1) * s0
    mkpts1_c = torch.stack([j_ids % w1c, j_ids // w1c], dim=1) * s1
    keep = mconf != 0
    return {
        "b_ids": b_ids, "i_ids": i_ids, "j_ids": j_ids,
        "gt_mask": mconf == 0,
        "m_bids": b_ids[keep],
        "mkpts0_c": mkpts0_c[keep],
        "mkpts1_c": mkpts1_c[keep],
        "mconf": mconf[keep],
    }


# ----------------------------------------------------------------------------------------------
# fine level -- submodules/fine_preprocess.py, utils/fine_matching.py
# ----------------------------------------------------------------------------------------------
def fine_preprocess(feat_f0, feat_f1, b_ids, i_ids, j_ids, hw0_c, hw0_f, W=5):
    """FinePreprocess.forward with fine_concat_coarse_feat=False (fine_preprocess.py:29-47).
    Window = F.unfold(kernel W, stride hw0_f[0]//hw0_c[0] (=4), padding W//2) then pick the matches."""
    stride = hw0_f[0] // hw0_c[0]
    c = feat_f0.shape[1]
    if b_ids.shape[0] == 0:
        return torch.empty(0, W * W, c), torch.empty(0, W * W, c)

    def unfold(f):
        u = F.unfold(f, kernel_size=(W, W), stride=stride, padding=W // 2)  # [n, c*ww, l]
        n = u.shape[0]
        return u.view(n, c, W * W, -1).permute(0, 3, 2, 1)  # 'n (c ww) l -> n l ww c'

    return unfold(feat_f0)[b_ids, i_ids], unfold(feat_f1)[b_ids, j_ids]


def fine_matching(feat_f0, feat_f1, mkpts0_c, mkpts1_c, b_ids, n_mconf, hw0_i, hw0_f,
                  scale1=None, has_scale0=False):
    """FineMatching.forward + get_fine_match (fine_matching.py:15-74).  The kornia calls
    (fine_matching.py:49-50) are restated: meshgrid for W=5 is {-1,-.5,0,.5,1}^2 (x fastest) and
    spatial_expectation2d is the heatmap-weighted mean of that grid.
    NB quirk preserved: the scale switch is keyed on 'scale0' in data but multiplies scale1 (line 68)."""
    M, WW, C = feat_f0.shape
    W = int(math.sqrt(WW))
    scale = hw0_i[0] / hw0_f[0]
    if M == 0:
        return {"expec_f": torch.empty(0, 3), "mkpts0_f": mkpts0_c, "mkpts1_f": mkpts1_c}
    picked = feat_f0[:, WW // 2, :]
    sim = torch.einsum("mc,mrc->mr", picked, feat_f1)
    heat = torch.softmax((1. / C ** .5) * sim, dim=1)  # [M, WW]
    lin = torch.linspace(-1, 1, W)
    gy, gx = torch.meshgrid(lin, lin, indexing="ij")
    grid = torch.stack([gx, gy], -1).reshape(1, WW, 2)
    coords = torch.sum(grid * heat[:, :, None], dim=1)  # [M,2] (x,y)
    var = torch.sum(grid ** 2 * heat.view(-1, WW, 1), dim=1) - coords ** 2
    std = torch.sum(torch.sqrt(torch.clamp(var, min=1e-10)), -1)
    expec_f = torch.cat([coords, std.unsqueeze(1)], -1)
    s1 = scale * scale1[b_ids] if has_scale0 else scale
    mkpts1_f = mkpts1_c + (coords * (W // 2) * s1)[:n_mconf]
    return {"expec_f": expec_f, "mkpts0_f": mkpts0_c, "mkpts1_f": mkpts1_f}


# ----------------------------------------------------------------------------------------------
# whole forward -- networks/loftr/loftr.py:43-91
# ----------------------------------------------------------------------------------------------
def loftr_forward(sd, data, config=None):
    """LoFTR.forward restated; mutates and returns `data` with the keys of SURVEY Appendix A2.
    `sd` is a state_dict with the reference's key names (fp32 CPU tensors)."""
    cfg = config or DEFAULT_CONFIG
    data.update({"bs": data["image0"].size(0),
                 "hw0_i": data["image0"].shape[2:], "hw1_i": data["image1"].shape[2:]})
    bs = data["bs"]
    if data["hw0_i"] == data["hw1_i"]:
        fc, ff = backbone(sd, torch.cat([data["color0"], data["color1"]], dim=0))
        (feat_c0, feat_c1), (feat_f0, feat_f1) = fc.split(bs), ff.split(bs)
    else:
        (feat_c0, feat_f0), (feat_c1, feat_f1) = backbone(sd, data["color0"]), backbone(sd, data["color1"])
    data.update({"hw0_c": feat_c0.shape[2:], "hw1_c": feat_c1.shape[2:],
                 "hw0_f": feat_f0.shape[2:], "hw1_f": feat_f1.shape[2:]})
    d_c = cfg["coarse"]["d_model"]
    pe0 = position_encoding(d_c, *feat_c0.shape[2:])
    pe1 = position_encoding(d_c, *feat_c1.shape[2:])
    feat_c0 = (feat_c0 + pe0).flatten(2).transpose(1, 2)  # 'n c h w -> n (h w) c'
    feat_c1 = (feat_c1 + pe1).flatten(2).transpose(1, 2)
    mask_c0 = mask_c1 = None
    if "mask0" in data:
        mask_c0, mask_c1 = data["mask0"].flatten(-2), data["mask1"].flatten(-2)
    feat_c0, feat_c1 = local_feature_transformer(
        sd, "loftr_coarse", feat_c0, feat_c1, cfg["coarse"]["nhead"], cfg["coarse"]["layer_names"],
        mask_c0, mask_c1)
    mc = cfg["match_coarse"]
    conf = conf_matrix_dual_softmax(feat_c0, feat_c1, mc["dsmax_temperature"], mask_c0, mask_c1)
    data["conf_matrix"] = conf
    data.update(get_coarse_match(conf, data["hw0_i"], data["hw1_i"], data["hw0_c"], data["hw1_c"],
                                 mc["thr"], mc["border_rm"], data.get("scale0"), data.get("scale1"),
                                 data.get("mask0"), data.get("mask1")))
    W = cfg["fine_window_size"]
    data["W"] = W
    f0u, f1u = fine_preprocess(feat_f0, feat_f1, data["b_ids"], data["i_ids"], data["j_ids"],
                               data["hw0_c"], data["hw0_f"], W)
    if f0u.size(0) != 0:
        f0u, f1u = local_feature_transformer(sd, "loftr_fine", f0u, f1u, cfg["fine"]["nhead"],
                                             cfg["fine"]["layer_names"])
    data.update(fine_matching(f0u, f1u, data["mkpts0_c"], data["mkpts1_c"], data["b_ids"],
                              len(data["mconf"]), data["hw0_i"], data["hw0_f"],
                              data.get("scale1"), "scale0" in data))
    return data


# ----------------------------------------------------------------------------------------------
# synthetic inputs shared by tests / bench: planted-correspondence coarse features
# ----------------------------------------------------------------------------------------------
def planted_coarse_features(n, hw, c=256, sigma=2.0, eps=0.1, seed=0):
    """f0 ~ N(0, sigma^2); f1 = f0[:, perm] + eps*N(0, sigma^2)  =>  ~3.7k mutual matches per pair at
    60x80.  Returns (f0, f1, perm) with f1[:, k] planted from f0[:, perm[k]]."""
    g = torch.Generator().manual_seed(seed)
    L = hw[0] * hw[1]
    f0 = torch.randn(n, L, c, generator=g) * sigma
    perm = torch.stack([torch.randperm(L, generator=g) for _ in range(n)])
    f1 = torch.gather(f0, 1, perm[:, :, None].expand(-1, -1, c)) + eps * sigma * torch.randn(n, L, c, generator=g)
    return f0, f1, perm


# ----------------------------------------------------------------------------------------------
# seeded weights with the reference's state_dict key names / shapes (no checkpoint ships with the
# reference: `.MISSING_LARGE_BLOBS`).  `oracle/make_golden.py` asserts `load_state_dict(strict=True)`
# into the real reference module accepts exactly this key set.
# ----------------------------------------------------------------------------------------------
def loftr_param_spec(config=None):
    """Ordered [(key, shape, kind)], kind in conv|bn_w|bn_b|bn_rm|bn_rv|bn_n|lin|ln_w|ln_b."""
    cfg = config or DEFAULT_CONFIG
    spec = []

    def conv(p, co, ci, k):
        spec.append((p + ".weight", (co, ci, k, k), "conv"))

    def bn(p, c):
        spec.extend([(p + ".weight", (c,), "bn_w"), (p + ".bias", (c,), "bn_b"),
                     (p + ".running_mean", (c,), "bn_rm"), (p + ".running_var", (c,), "bn_rv"),
                     (p + ".num_batches_tracked", (), "bn_n")])

    e = "backbone.encode"
    conv(e + ".conv1", 64, 3, 7)
    bn(e + ".bn1", 64)
    inpl = 64
    for li, (planes, nblk) in enumerate(((64, 3), (128, 4), (256, 6)), start=1):
        for bi in range(nblk):
            p = f"{e}.layer{li}.{bi}"
            conv(p + ".conv1", planes, inpl, 1); bn(p + ".bn1", planes)
            conv(p + ".conv2", planes, planes, 3); bn(p + ".bn2", planes)
            conv(p + ".conv3", planes * 4, planes, 1); bn(p + ".bn3", planes * 4)
            if bi == 0:
                conv(p + ".downsample.0", planes * 4, inpl, 1); bn(p + ".downsample.1", planes * 4)
            inpl = planes * 4
    bd = cfg["resnetfpn"]["block_dims"]
    conv("backbone.layer3_outconv", bd[3], bd[5], 1)
    conv("backbone.layer2_outconv", bd[3], bd[4], 1)
    conv("backbone.layer2_outconv2.0", bd[3], bd[3], 3); bn("backbone.layer2_outconv2.1", bd[3])
    conv("backbone.layer2_outconv2.3", bd[2], bd[3], 3)
    conv("backbone.layer1_outconv", bd[2], bd[3], 1)
    conv("backbone.layer1_outconv2.0", bd[2], bd[2], 3); bn("backbone.layer1_outconv2.1", bd[2])
    conv("backbone.layer1_outconv2.3", bd[1], bd[2], 3)
    for name, key in (("loftr_coarse", "coarse"), ("loftr_fine", "fine")):
        d = cfg[key]["d_model"]
        for li in range(2 * cfg[key]["layer_names"]):
            p = f"{name}.layers.{li}"
            for lin in ("q_proj", "k_proj", "v_proj", "merge"):
                spec.append((f"{p}.{lin}.weight", (d, d), "lin"))
            spec.append((f"{p}.mlp.0.weight", (2 * d, 2 * d), "lin"))
            spec.append((f"{p}.mlp.2.weight", (d, 2 * d), "lin"))
            spec.extend([(f"{p}.norm1.weight", (d,), "ln_w"), (f"{p}.norm1.bias", (d,), "ln_b"),
                         (f"{p}.norm2.weight", (d,), "ln_w"), (f"{p}.norm2.bias", (d,), "ln_b")])
    return spec


def make_state_dict(seed=0, config=None, gain=1.0):
    """Seeded fp32 weights.  Convs: N(0, 2/(cout*k*k)) (the reference's kaiming fan_out init,
    resnet.py:291-296); Linears: xavier-uniform (transformer.py:75-78); BN/LN affine and BN running
    stats are drawn away from the identity so that folding mistakes show up in parity tests."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape, kind in loftr_param_spec(config):
        if kind == "conv":
            co, ci, k, _ = shape
            sd[key] = torch.randn(shape, generator=g) * (gain * math.sqrt(2.0 / (co * k * k)))
        elif kind == "lin":
            a = gain * math.sqrt(6.0 / (shape[0] + shape[1]))
            sd[key] = (torch.rand(shape, generator=g) * 2 - 1) * a
        elif kind in ("bn_w", "ln_w"):
            sd[key] = 0.75 + 0.5 * torch.rand(shape, generator=g)
        elif kind in ("bn_b", "ln_b", "bn_rm"):
            sd[key] = 0.1 * torch.randn(shape, generator=g)
        elif kind == "bn_rv":
            sd[key] = 0.5 + torch.rand(shape, generator=g)
        elif kind == "bn_n":
            sd[key] = torch.tensor(0, dtype=torch.int64)
    return sd


def seeded_images(n, h, w, seed=1234):
    """color0/color1 ~ U[0,1) fp32 [n,3,h,w] (SURVEY 8d throughput inputs)."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, 3, h, w, generator=g), torch.rand(n, 3, h, w, generator=g)
