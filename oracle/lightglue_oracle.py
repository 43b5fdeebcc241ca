"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the gim_lightglue path.

A functional fp32 torch restatement of what the reference computes for `--model gim_lightglue`:

    SuperPoint._forward                networks/lightglue/superpoint.py:206-354
      simple_nms                       superpoint.py:61-80
      top_k_keypoints                  superpoint.py:83-87
      sample_descriptors (legacy)      superpoint.py:120-137
    LightGlue.forward                  networks/lightglue/models/matchers/lightglue.py:405-545
      normalize_keypoints              lightglue.py:21-33
      LearnableFourierPositionalEncoding  lightglue.py:47-61
      SelfBlock / CrossBlock           lightglue.py:121-211
      MatchAssignment + sigmoid_log_double_softmax   lightglue.py:248-281
      filter_matches                   lightglue.py:284-300
    caller-side adapter                trainer/lightning.py:161-193 (same code in demo.py:472-511)

with the configuration the reference's callers build (`trainer/lightning.py:49-60`, `demo.py:338-349`):
SuperPoint max_num_keypoints 2048, force_num_keypoints, detection_threshold 0.0, nms_radius 3,
remove_borders 4, legacy_sampling; LightGlue 9 layers, 4 heads, d 256, filter_threshold 0.1, flash False,
depth/width confidence -1 (no early stop, no pruning).

Parity: PINNED -- `oracle/make_golden_lightglue.py` runs the reference's own modules (imported through
`oracle/ref_shims.py::install_omegaconf`) on seeded weights/inputs, checks this restatement against them
and commits the vectors under `tests/golden/lg_*.npz`; `tests/test_oracle_golden.py` re-checks them.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this module; the
product (`gim_amd/`) never does.
"""
import math

import torch
import torch.nn.functional as F

SP_CONF = {"nms_radius": 3, "remove_borders": 4, "detection_threshold": 0.0, "max_num_keypoints": 2048,
           "force_num_keypoints": True, "descriptor_dim": 256}
LG_CONF = {"n_layers": 9, "num_heads": 4, "descriptor_dim": 256, "filter_threshold": 0.1}


# --------------------------------------------------------------------------------------------- parameters
def superpoint_param_spec():
    """name -> shape, in `SuperPoint._init` order (superpoint.py:179-204)."""
    c1, c2, c3, c4, c5 = 64, 64, 128, 128, 256
    convs = [("conv1a", 1, c1, 3), ("conv1b", c1, c1, 3), ("conv2a", c1, c2, 3), ("conv2b", c2, c2, 3),
             ("conv3a", c2, c3, 3), ("conv3b", c3, c3, 3), ("conv4a", c3, c4, 3), ("conv4b", c4, c4, 3),
             ("convPa", c4, c5, 3), ("convPb", c5, 65, 1), ("convDa", c4, c5, 3), ("convDb", c5, 256, 1)]
    spec = {}
    for name, ci, co, k in convs:
        spec[name + ".weight"] = (co, ci, k, k)
        spec[name + ".bias"] = (co,)
    return spec


def lightglue_param_spec(n_layers=9, d=256, heads=4):
    """name -> shape, in `LightGlue.__init__` order (lightglue.py:335-355)."""
    spec = {"posenc.Wr.weight": (d // heads // 2, 2)}

    def lin(name, i, o):
        spec[name + ".weight"] = (o, i)
        spec[name + ".bias"] = (o,)

    def ffn(p):
        lin(p + ".ffn.0", 2 * d, 2 * d)
        spec[p + ".ffn.1.weight"] = (2 * d,)
        spec[p + ".ffn.1.bias"] = (2 * d,)
        lin(p + ".ffn.3", 2 * d, d)

    for i in range(n_layers):
        p = f"transformers.{i}.self_attn"
        lin(p + ".Wqkv", d, 3 * d)
        lin(p + ".out_proj", d, d)
        ffn(p)
        p = f"transformers.{i}.cross_attn"
        lin(p + ".to_qk", d, d)
        lin(p + ".to_v", d, d)
        lin(p + ".to_out", d, d)
        ffn(p)
    for i in range(n_layers):
        lin(f"log_assignment.{i}.matchability", d, 1)
        lin(f"log_assignment.{i}.final_proj", d, d)
    for i in range(n_layers - 1):
        lin(f"token_confidence.{i}.token.0", d, 1)
    return spec


def make_state_dicts(seed=0):
    """Seeded stand-in weights (no checkpoint exists in the container): fan-in scaled normals, LayerNorm
    near identity.  Shared by the oracle, the reference (golden generation) and the engine."""
    g = torch.Generator().manual_seed(seed)

    def fill(spec):
        sd = {}
        for k, shp in spec.items():
            if k.endswith("ffn.1.weight"):
                sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
            elif k.endswith(".bias"):
                sd[k] = 0.1 * torch.randn(shp, generator=g)
            else:
                fan_in = math.prod(shp[1:]) if len(shp) > 1 else shp[0]
                sd[k] = torch.randn(shp, generator=g) * (1.4 / math.sqrt(fan_in))
        return sd

    sp = fill(superpoint_param_spec())
    lg = fill(lightglue_param_spec())
    lg["posenc.Wr.weight"] = torch.randn(lg["posenc.Wr.weight"].shape, generator=g)  # gamma=1: std 1 (lightglue.py:53)
    # Random weights scramble the descriptors and leave ~3 matches per pair.  Keep the residual stream close
    # to its input (small last FFN layer) and make the last assignment head near 40*I with a positive
    # matchability bias, so planted correspondences survive and the matching arithmetic is exercised.
    for k in lg:
        if k.endswith("ffn.3.weight") or k.endswith("ffn.3.bias"):
            lg[k] = lg[k] * 0.05
    for i in range(9):
        w = lg[f"log_assignment.{i}.final_proj.weight"]
        lg[f"log_assignment.{i}.final_proj.weight"] = 40.0 * torch.eye(w.shape[0]) + w
        lg[f"log_assignment.{i}.matchability.bias"] = lg[f"log_assignment.{i}.matchability.bias"] + 3.0
    return sp, lg


# --------------------------------------------------------------------------------------------- SuperPoint
def superpoint_dense(sd, image):
    """superpoint.py:206-241: shared VGG encoder, detector head -> scores [B,H,W] (softmax over 65, dustbin
    dropped, 8x8 pixel shuffle), descriptor head -> L2-normalised dense descriptors [B,256,H/8,W/8]."""
    if image.shape[1] == 3:
        w = image.new_tensor([0.299, 0.587, 0.114]).view(1, 3, 1, 1)
        image = (image * w).sum(1, keepdim=True)

    def conv(name, x, pad):
        return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], padding=pad)

    x = F.relu(conv("conv1a", image, 1))
    x = F.relu(conv("conv1b", x, 1))
    x = F.max_pool2d(x, 2, 2)
    x = F.relu(conv("conv2a", x, 1))
    x = F.relu(conv("conv2b", x, 1))
    x = F.max_pool2d(x, 2, 2)
    x = F.relu(conv("conv3a", x, 1))
    x = F.relu(conv("conv3b", x, 1))
    x = F.max_pool2d(x, 2, 2)
    x = F.relu(conv("conv4a", x, 1))
    x = F.relu(conv("conv4b", x, 1))
    logits = conv("convPb", F.relu(conv("convPa", x, 1)), 0)
    prob = F.softmax(logits, 1)[:, :-1]
    b, _, h, w = prob.shape
    scores = prob.permute(0, 2, 3, 1).reshape(b, h, w, 8, 8).permute(0, 1, 3, 2, 4).reshape(b, h * 8, w * 8)
    desc = conv("convDb", F.relu(conv("convDa", x, 1)), 0)
    return scores, F.normalize(desc, p=2, dim=1)


def simple_nms(scores, radius):
    """superpoint.py:61-80: max-pool NMS with two suppression refinements; equal neighbours both survive."""
    def mp(x):
        return F.max_pool2d(x, kernel_size=2 * radius + 1, stride=1, padding=radius)

    zeros = torch.zeros_like(scores)
    keep = scores == mp(scores)
    for _ in range(2):
        supp = mp(keep.float()) > 0
        s2 = torch.where(supp, zeros, scores)
        keep = keep | ((s2 == mp(s2)) & ~supp)
    return torch.where(keep, scores, zeros)


def mask_borders(scores, image_size, border):
    """superpoint.py:247-258: -1 on the first `border` rows/cols and from image_size - border on."""
    scores = scores.clone()
    scores[:, :border] = -1
    scores[:, :, :border] = -1
    for i in range(scores.shape[0]):
        w, h = int(image_size[i][0]), int(image_size[i][1])
        scores[i, h - border:] = -1
        scores[i, :, w - border:] = -1
    return scores


def select_keypoints(scores, thr, k):
    """superpoint.py:260-300: candidates in `torch.where` order, then per image top-k by score (sorted)."""
    b = scores.shape[0]
    idx = torch.where(scores > thr)
    val = scores[idx]
    kpts, sc = [], []
    for i in range(b):
        sel = idx[0] == i
        yx = torch.stack(idx[1:3], -1)[sel]
        s = val[sel]
        if k < len(yx):
            s, top = torch.topk(s, k, dim=0, sorted=True)
            yx = yx[top]
        kpts.append(torch.flip(yx, [1]).float())   # (h, w) -> (x, y), superpoint.py:308
        sc.append(s)
    return kpts, sc


def sample_descriptors_legacy(keypoints, descriptors, s=8):
    """superpoint.py:120-137 (the 'legacy (broken)' sampling the default config keeps): bilinear
    grid_sample with align_corners=True at ((kp - s/2 + 0.5) / (w*s - s/2 - 0.5))*2-1, then L2 normalise."""
    b, c, h, w = descriptors.shape
    kp = keypoints - s / 2 + 0.5
    kp = kp / torch.tensor([(w * s - s / 2 - 0.5), (h * s - s / 2 - 0.5)]).to(kp)[None]
    kp = kp * 2 - 1
    d = F.grid_sample(descriptors, kp.view(b, 1, -1, 2), mode="bilinear", align_corners=True)
    return F.normalize(d.reshape(b, c, -1), p=2, dim=1)


def superpoint_forward(sd, data, conf=SP_CONF):
    """`SuperPoint._forward` for the gim_lightglue configuration.  data: {'image': [B,1|3,H,W]}.
    Quirk kept: the reference overwrites any caller-supplied `image_size` with the canvas size
    (`data["image_size"] = torch.tensor(image.shape[-2:][::-1])[None]`, superpoint.py:207), so the border
    rule always uses the full canvas, and -- that tensor having one row -- the reference itself raises
    IndexError for B > 1 (superpoint.py:254); this restatement applies the canvas size to every image,
    i.e. it equals the reference run image by image.  Returns {'keypoints': [B,K,2] (x,y)+0.5, 'descriptors': [B,K,256]}
    plus the intermediates the stage tests compare ('keypoint_scores_dense', 'nms_scores', 'dense_descriptors',
    'keypoint_scores').  Requires >= K candidates per image (the random padding of `pad_and_stack`
    'random_c', misc.py:44-55, draws from the torch RNG and is restated in the host code, not here)."""
    image = data["image"]
    b = image.shape[0]
    size = torch.tensor(image.shape[-2:][::-1])[None].expand(b, 2)
    dense_scores, dense_desc = superpoint_dense(sd, image)
    nms = simple_nms(dense_scores, conf["nms_radius"])
    nms = mask_borders(nms, size, conf["remove_borders"])
    kpts, sc = select_keypoints(nms, conf["detection_threshold"], conf["max_num_keypoints"])
    k = conf["max_num_keypoints"]
    assert all(len(x) == k for x in kpts), "oracle covers the >=K-candidates case only"
    kpts, sc = torch.stack(kpts, 0), torch.stack(sc, 0)
    desc = sample_descriptors_legacy(kpts.clone(), dense_desc, 8)
    return {"keypoints": kpts + 0.5, "descriptors": desc.transpose(-1, -2), "keypoint_scores": sc,
            "keypoint_scores_dense": dense_scores, "nms_scores": nms, "dense_descriptors": dense_desc}


# ---------------------------------------------------------------------------------------------- LightGlue
def normalize_keypoints(kpts, size):
    """lightglue.py:21-33 with an explicit size [B,2] = (w,h): centre, divide by max(w,h)/2."""
    size = size.to(kpts)
    shift = size / 2
    scale = size.max(-1).values / 2
    return (kpts - shift[..., None, :]) / scale[..., None, None]


def fourier_encoding(sd, kpts):
    """lightglue.py:47-61: Wr (2 -> 32, no bias); [cos; sin] stacked, each repeated x2 along the last dim
    -> [2, B, 1, K, 64]."""
    proj = kpts @ sd["posenc.Wr.weight"].t()
    emb = torch.stack([torch.cos(proj), torch.sin(proj)], 0).unsqueeze(-3)
    return emb.repeat_interleave(2, dim=-1)


def _rotate_half(x):
    x = x.unflatten(-1, (-1, 2))
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).flatten(start_dim=-2)


def _rotary(freqs, t):
    return t * freqs[0] + _rotate_half(t) * freqs[1]


def _linear(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def _ffn(sd, p, x):
    y = _linear(sd, p + ".ffn.0", x)
    y = F.layer_norm(y, (y.shape[-1],), sd[p + ".ffn.1.weight"], sd[p + ".ffn.1.bias"])
    return _linear(sd, p + ".ffn.3", F.gelu(y))


def _softmax_attention(q, k, v):
    """lightglue.py:106-109: with flash=False and torch >= 2 the reference calls
    F.scaled_dot_product_attention on contiguous fp32 q,k,v = softmax(q k^T / sqrt(d)) v."""
    return F.scaled_dot_product_attention(q.contiguous(), k.contiguous(), v.contiguous())


def self_block(sd, p, x, enc, heads=4):
    """lightglue.py:142-156: fused Wqkv whose output feature f = h*(3*dh) + d*3 + {q,k,v}; rotary on q,k."""
    qkv = _linear(sd, p + ".Wqkv", x).unflatten(-1, (heads, -1, 3)).transpose(1, 2)
    q, k, v = qkv[..., 0], qkv[..., 1], qkv[..., 2]
    ctx = _softmax_attention(_rotary(enc, q), _rotary(enc, k), v)
    msg = _linear(sd, p + ".out_proj", ctx.transpose(1, 2).flatten(start_dim=-2))
    return x + _ffn(sd, p, torch.cat([x, msg], -1))


def cross_block(sd, p, x0, x1, heads=4):
    """lightglue.py:183-211 (non-flash branch): one similarity, softmax along both axes."""
    def split(t):
        return t.unflatten(-1, (heads, -1)).transpose(1, 2)

    qk0, qk1 = split(_linear(sd, p + ".to_qk", x0)), split(_linear(sd, p + ".to_qk", x1))
    v0, v1 = split(_linear(sd, p + ".to_v", x0)), split(_linear(sd, p + ".to_v", x1))
    s = (qk0.shape[-1] ** -0.5) ** 0.5
    sim = torch.einsum("bhid,bhjd->bhij", qk0 * s, qk1 * s)
    m0 = torch.einsum("bhij,bhjd->bhid", F.softmax(sim, -1), v1)
    m1 = torch.einsum("bhji,bhjd->bhid", F.softmax(sim.transpose(-2, -1).contiguous(), -1).transpose(-2, -1), v0)
    m0, m1 = (t.transpose(1, 2).flatten(start_dim=-2) for t in (m0, m1))
    m0, m1 = _linear(sd, p + ".to_out", m0), _linear(sd, p + ".to_out", m1)
    return x0 + _ffn(sd, p, torch.cat([x0, m0], -1)), x1 + _ffn(sd, p, torch.cat([x1, m1], -1))


def log_assignment(sd, i, desc0, desc1):
    """lightglue.py:248-276: final_proj / d^(1/4), similarity, log double softmax + matchability."""
    p = f"log_assignment.{i}"
    d = desc0.shape[-1]
    md0, md1 = _linear(sd, p + ".final_proj", desc0) / d ** 0.25, _linear(sd, p + ".final_proj", desc1) / d ** 0.25
    sim = torch.einsum("bmd,bnd->bmn", md0, md1)
    z0, z1 = _linear(sd, p + ".matchability", desc0), _linear(sd, p + ".matchability", desc1)
    b, m, n = sim.shape
    cert = F.logsigmoid(z0) + F.logsigmoid(z1).transpose(1, 2)
    s0 = F.log_softmax(sim, 2)
    s1 = F.log_softmax(sim.transpose(-1, -2).contiguous(), 2).transpose(-1, -2)
    scores = sim.new_zeros((b, m + 1, n + 1))
    scores[:, :m, :n] = s0 + s1 + cert
    scores[:, :-1, -1] = F.logsigmoid(-z0.squeeze(-1))
    scores[:, -1, :-1] = F.logsigmoid(-z1.squeeze(-1))
    return scores, sim


def filter_matches(scores, th):
    """lightglue.py:284-300: mutual argmax of the assignment core, exp(max) > th."""
    max0, max1 = scores[:, :-1, :-1].max(2), scores[:, :-1, :-1].max(1)
    m0, m1 = max0.indices, max1.indices
    i0 = torch.arange(m0.shape[1])[None]
    i1 = torch.arange(m1.shape[1])[None]
    mutual0 = i0 == m1.gather(1, m0)
    mutual1 = i1 == m0.gather(1, m1)
    e0 = max0.values.exp()
    zero = e0.new_tensor(0)
    ms0 = torch.where(mutual0, e0, zero)
    ms1 = torch.where(mutual1, ms0.gather(1, m1), zero)
    valid0 = mutual0 & (ms0 > th)
    valid1 = mutual1 & valid0.gather(1, m1)
    return torch.where(valid0, m0, -1), torch.where(valid1, m1, -1), ms0, ms1


def lightglue_forward(sd, data, conf=LG_CONF):
    """`LightGlue.forward` in eval mode without early stop / pruning (lightglue.py:405-545).  data needs
    keypoints0/1 [B,K,2], descriptors0/1 [B,K,256] and image_size0/1 or resize0/1 given as [B,2] (h,w)
    (the callers pass `resize[:, [1,0]]`-style tensors that forward flips again, lightglue.py:414-415)."""
    kp0, kp1 = data["keypoints0"], data["keypoints1"]
    size0 = (data["image_size0"] if "image_size0" in data else data["resize0"])[:, [1, 0]]
    size1 = (data["image_size1"] if "image_size1" in data else data["resize1"])[:, [1, 0]]
    enc0 = fourier_encoding(sd, normalize_keypoints(kp0, size0))
    enc1 = fourier_encoding(sd, normalize_keypoints(kp1, size1))
    d0, d1 = data["descriptors0"].contiguous(), data["descriptors1"].contiguous()
    h = conf["num_heads"]
    n = conf["n_layers"]
    for i in range(n):
        d0 = self_block(sd, f"transformers.{i}.self_attn", d0, enc0, h)
        d1 = self_block(sd, f"transformers.{i}.self_attn", d1, enc1, h)
        d0, d1 = cross_block(sd, f"transformers.{i}.cross_attn", d0, d1, h)
    scores, sim = log_assignment(sd, n - 1, d0, d1)
    m0, m1, ms0, ms1 = filter_matches(scores, conf["filter_threshold"])
    matches, mscores = [], []
    for k in range(kp0.shape[0]):
        valid = m0[k] > -1
        matches.append(torch.stack([torch.where(valid)[0], m0[k][valid]], -1))
        mscores.append(ms0[k][valid])
    return {"matches0": m0, "matches1": m1, "matching_scores0": ms0, "matching_scores1": ms1,
            "ref_descriptors0": d0[:, None], "ref_descriptors1": d1[:, None], "log_assignment": scores,
            "stop": n, "matches": matches, "scores": mscores,
            "prune0": torch.ones_like(ms0) * n, "prune1": torch.ones_like(ms1) * n}


# --------------------------------------------------------------------------------- caller-side adapter (a15)
def gim_lightglue_inference(sp_sd, lg_sd, data, sp_conf=SP_CONF):
    """`Trainer.gim_lightglue_inference` (trainer/lightning.py:161-193): detector on image0/1 with
    image_size = resize[:, [1,0]], matcher, keypoints * scale, per-pair gather of the matched keypoints.
    Returns the dict the reference writes into `data`."""
    pred = {}
    for s in "01":
        out = superpoint_forward(sp_sd, {"image": data["image" + s], "image_size": data["resize" + s][:, [1, 0]]}, sp_conf)
        pred["keypoints" + s], pred["descriptors" + s] = out["keypoints"], out["descriptors"]
    pred.update(lightglue_forward(lg_sd, {**pred, **data}))
    bs = data["image0"].shape[0]
    k0 = torch.cat([kp * s for kp, s in zip(pred["keypoints0"], data["scale0"][:, None])])
    k1 = torch.cat([kp * s for kp, s in zip(pred["keypoints1"], data["scale1"][:, None])])
    m_bids = torch.nonzero(pred["keypoints0"].sum(dim=2) > -1)[:, 0]
    mt = pred["matches"]
    out0 = torch.cat([k0[m_bids == b][mt[b][..., 0]] for b in range(bs)])
    out1 = torch.cat([k1[m_bids == b][mt[b][..., 1]] for b in range(bs)])
    bids = torch.cat([m_bids[m_bids == b][mt[b][..., 0]] for b in range(bs)])
    return {"hw0_i": data["color0"].shape[2:] if "color0" in data else data["image0"].shape[2:],
            "hw1_i": data["color1"].shape[2:] if "color1" in data else data["image1"].shape[2:],
            "mkpts0_f": out0, "mkpts1_f": out1, "m_bids": bids, "mconf": torch.cat(pred["scores"]), "pred": pred}


def seeded_gray(b, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    # band-limited texture: plain uniform noise gives SuperPoint a flat score map
    base = torch.rand(b, 1, h // 4, w // 4, generator=g)
    img = F.interpolate(base, size=(h, w), mode="bilinear", align_corners=False)
    return (0.7 * img + 0.3 * torch.rand(b, 1, h, w, generator=g)).contiguous()


def planted_descriptors(b, k, seed=0, noise=0.05, frac=0.7):
    """Match-rich LightGlue stage inputs: keypoints U[0,640)x[0,480), unit descriptors; a
    fraction `frac` of image1's points are a permutation of image0's (same position + jitter, same descriptor
    + noise), the rest unrelated.  Random weights still decide what the network does with them; the tests
    use these to exercise attention/assignment arithmetic on non-degenerate data."""
    g = torch.Generator().manual_seed(seed)
    kp0 = torch.rand(b, k, 2, generator=g) * torch.tensor([640.0, 480.0])
    d0 = F.normalize(torch.randn(b, k, 256, generator=g), dim=-1)
    kp1, d1 = torch.empty_like(kp0), torch.empty_like(d0)
    nm = int(frac * k)
    for i in range(b):
        perm = torch.randperm(k, generator=g)
        kp1[i, perm[:nm]] = kp0[i, :nm] + torch.randn(nm, 2, generator=g)
        d1[i, perm[:nm]] = F.normalize(d0[i, :nm] + noise * torch.randn(nm, 256, generator=g), dim=-1)
        kp1[i, perm[nm:]] = torch.rand(k - nm, 2, generator=g) * torch.tensor([640.0, 480.0])
        d1[i, perm[nm:]] = F.normalize(torch.randn(k - nm, 256, generator=g), dim=-1)
    return kp0, d0, kp1.clamp_(min=0), d1
