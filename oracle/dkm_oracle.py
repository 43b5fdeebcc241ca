"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the gim_dkm path (kernels D1-D9).

A functional fp32 torch restatement of what the reference computes for `--model gim_dkm`
(`DKMv3(None, h, w)` + `RegressionMatcher.match` / `.sample`, symmetric, upsampled predictions):

    ResNet50 pyramid {1,2,4,8,16,32}        networks/dkm/models/encoders.py:30-62 (torchvision resnet50, no fc)
    CosKernel, GP.forward (no_cov)           networks/dkm/models/dkm.py:126-144, 340-370
    DFN / RRB / CAB embedding decoder        dkm.py:147-254
    ConvRefiner.forward                      dkm.py:75-123
    local_correlation                        networks/dkm/utils/local_correlation.py:5-40
    Decoder.forward                          dkm.py:451-534
    RegressionMatcher.forward_symmetric      dkm.py:637-646
    RegressionMatcher.match                  dkm.py:654-752
    RegressionMatcher.sample + kde           dkm.py:583-620, networks/dkm/utils/kde.py:17-26
    model definition (widths, radii)         networks/dkm/models/model_zoo/DKMv3.py:5-145
    caller-side adapter                      trainer/lightning.py:134-156

Parity: PINNED -- `oracle/make_golden_dkm.py` builds the reference's own DKMv3 (through
`oracle/ref_shims.py::install_dkm`: empty cv2, torchvision.models.resnet50 rebuilt from the reference's own
ResNet/Bottleneck), loads the seeded weights of `make_state_dict`, runs reference and restatement on the same
seeded images and records the reference's outputs under tests/golden/dkm_*.npz.

The HIP path for this row is not built yet (round 2): this oracle and its golden vectors are the first step.
Only `tests/` may import this module.
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
REFINER = {  # scale -> (feature dim per image, displacement-embedding dim, local-correlation radius)  DKMv3.py:51-110
    "16": (512, 128, 7), "8": (512, 64, 3), "4": (256, 32, 2), "2": (64, 16, None), "1": (3, 6, None)}
GP_DIM, DFN_DIM, FEAT_DIM = 256, 384, 256
HIDDEN_BLOCKS = 8


# ------------------------------------------------------------------------------------------------ parameters
def _refiner_dims(scale):
    c, e, r = REFINER[scale]
    in_dim = 2 * c + e + ((2 * r + 1) ** 2 if r else 0)
    hidden = {"2": 128 + 16, "1": 24}.get(scale, in_dim)   # DKMv3.py:88-110: s2 hidden = in, s1 hidden = 24
    return in_dim, hidden


def dkm_param_spec():
    """name -> (shape, kind) in the reference's `state_dict()` order is not needed; names and shapes are."""
    spec = {}

    def conv(name, ci, co, k, groups=1, bias=True):
        spec[name + ".weight"] = (co, ci // groups, k, k)
        if bias:
            spec[name + ".bias"] = (co,)

    def bn(name, c):
        for s in ("weight", "bias", "running_mean", "running_var"):
            spec[f"{name}.{s}"] = (c,)
        spec[name + ".num_batches_tracked"] = ()

    # encoder.net = torchvision resnet50 without fc
    p = "encoder.net."
    conv(p + "conv1", 3, 64, 7, bias=False)
    bn(p + "bn1", 64)
    inpl = 64
    for li, (planes, nblk) in enumerate(((64, 3), (128, 4), (256, 6), (512, 3)), start=1):
        for bi in range(nblk):
            q = f"{p}layer{li}.{bi}."
            conv(q + "conv1", inpl, planes, 1, bias=False); bn(q + "bn1", planes)
            conv(q + "conv2", planes, planes, 3, bias=False); bn(q + "bn2", planes)
            conv(q + "conv3", planes, planes * 4, 1, bias=False); bn(q + "bn3", planes * 4)
            if bi == 0:
                conv(q + "downsample.0", inpl, planes * 4, 1, bias=False); bn(q + "downsample.1", planes * 4)
            inpl = planes * 4
    # decoder
    d = "decoder."
    for s in ("32", "16"):
        e = d + "embedding_decoder."
        conv(f"{e}feat_input_modules.{s}", 512, FEAT_DIM, 1)
        for nm, ci in ((f"rrb_d.{s}", GP_DIM + FEAT_DIM), (f"rrb_u.{s}", DFN_DIM)):
            conv(f"{e}{nm}.conv1", ci, DFN_DIM, 1)
            conv(f"{e}{nm}.conv2", DFN_DIM, DFN_DIM, 3)
            bn(f"{e}{nm}.bn", DFN_DIM)
            conv(f"{e}{nm}.conv3", DFN_DIM, DFN_DIM, 3)
        conv(f"{e}cab.{s}.conv1", 2 * DFN_DIM, DFN_DIM, 1)
        conv(f"{e}cab.{s}.conv2", DFN_DIM, DFN_DIM, 1)
        conv(f"{e}terminal_module.{s}", DFN_DIM, 3, 1)
        conv(f"{d}gps.{s}.pos_conv", 2, GP_DIM, 1)
    conv(d + "proj.16", 1024, 512, 1)
    conv(d + "proj.32", 2048, 512, 1)
    for s in REFINER:
        in_dim, hid = _refiner_dims(s)
        r = f"{d}conv_refiner.{s}."
        blocks = [("block1", in_dim)] + [(f"hidden_blocks.{i}", hid) for i in range(HIDDEN_BLOCKS)]
        for nm, ci in blocks:
            conv(f"{r}{nm}.0", ci, hid, 5, groups=ci)       # depthwise 5x5 (dw=True), out = multiple of in
            bn(f"{r}{nm}.1", hid)
            conv(f"{r}{nm}.3", hid, hid, 1)
        conv(r + "out_conv", hid, 3, 1)
        conv(r + "disp_emb", 2, REFINER[s][1], 1)
    return spec


def make_state_dict(seed=0):
    """Seeded stand-in weights (no checkpoint in the container): fan-in scaled normals, BatchNorm close to identity
    with non-trivial running statistics; the refiners' output convs are scaled down so that the predicted
    displacements stay small and the flow stays inside the image (the regime real weights operate in)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in dkm_param_spec().items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(0, dtype=torch.long)
        elif k.endswith("running_var"):
            sd[k] = 0.8 + 0.4 * torch.rand(shp, generator=g)
        elif k.endswith("running_mean"):
            sd[k] = 0.05 * torch.randn(shp, generator=g)
        elif (".bn" in k or ".downsample.1" in k or k.split(".")[-2] == "1") and k.endswith("weight") and len(shp) == 1:
            sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            sd[k] = 0.02 * torch.randn(shp, generator=g)
        else:
            fan_in = math.prod(shp[1:])
            sd[k] = torch.randn(shp, generator=g) * (1.0 / math.sqrt(fan_in))
    for s in REFINER:
        for t in ("weight", "bias"):
            sd[f"decoder.conv_refiner.{s}.out_conv.{t}"] *= 0.05
    for s in ("32", "16"):
        sd[f"decoder.gps.{s}.pos_conv.weight"] *= 0.25
    return sd


# ------------------------------------------------------------------------------------------------ layers
def _conv(sd, name, x, stride=1, pad=0, groups=1):
    return F.conv2d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride=stride, padding=pad, groups=groups)


def _bn(sd, name, x):
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"], sd[name + ".bias"],
                        False, 0.0, BN_EPS)


def resnet50_pyramid(sd, x, p="encoder.net."):
    """encoders.py:43-62: feats[1] = image, [2] = relu(bn1(conv1)), maxpool, layer1..4 -> [4],[8],[16],[32]."""
    feats = {1: x}
    x = F.relu(_bn(sd, p + "bn1", _conv(sd, p + "conv1", x, 2, 3)))
    feats[2] = x
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (nblk, stride) in enumerate(((3, 1), (4, 2), (6, 2), (3, 2)), start=1):
        for bi in range(nblk):
            q = f"{p}layer{li}.{bi}."
            st = stride if bi == 0 else 1
            idn = x
            o = F.relu(_bn(sd, q + "bn1", _conv(sd, q + "conv1", x)))
            o = F.relu(_bn(sd, q + "bn2", _conv(sd, q + "conv2", o, st, 1)))     # torchvision v1.5: stride on conv2
            o = _bn(sd, q + "bn3", _conv(sd, q + "conv3", o))
            if bi == 0:
                idn = _bn(sd, q + "downsample.1", _conv(sd, q + "downsample.0", x, st))
            x = F.relu(o + idn)
        feats[2 ** (li + 1)] = x
    return feats


def grid_coords(b, h, w):
    """the normalised pixel-centre grid used everywhere (dkm.py:91-98, 318-331, 437-448): [b,2,h,w], (x, y)"""
    ys = torch.linspace(-1 + 1 / h, 1 - 1 / h, h)
    xs = torch.linspace(-1 + 1 / w, 1 - 1 / w, w)
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    return torch.stack((gx, gy))[None].expand(b, 2, h, w)


def cos_kernel(x, y, T=0.2, eps=1e-6):
    """dkm.py:135-144: exp((cos(x_i, y_j) - 1) / T)"""
    c = torch.einsum("bnd,bmd->bnm", x, y) / (x.norm(dim=-1)[..., None] * y.norm(dim=-1)[:, None] + eps)
    return ((c - 1.0) / T).exp()


# Test switch (never set by the golden-vector generators): evaluate the GP posterior -- the ONE ill-conditioned step of the dense
# matchers, cond(K_yy + sigma I) ~ 2e4 -- with the reference's formula in fp64 instead of its fp32 kernel matrix + fp32 LU inverse.
# fp32 arithmetic is ~1e-4 of mu away from the formula's exact value whatever the operation order (tests/test_gpu_gp_pins.py), so two
# fp32 implementations cannot be compared below that; with GP_FP64 the oracle IS the formula and the engine's parity mode (fp64 GP,
# csrc/gp_solve.hip) is compared with it at 1e-4 end to end, next to the reference arithmetic's own distance from it.
GP_FP64 = False


def gp_posterior_fp64(xr, yr, fr, sigma_noise=0.1, T=0.2, eps=1e-6):
    """mu = K_xy (K_yy + sigma I)^-1 f in fp64 from fp32 rows [b, n, d] / [b, n, c] (cos_kernel + dkm.py:352-362)"""
    x, y, f = xr.double(), yr.double(), fr.double()

    def k(a, b):
        c = torch.einsum("bnd,bmd->bnm", a, b) / (a.norm(dim=-1)[..., None] * b.norm(dim=-1)[:, None] + eps)
        return ((c - 1.0) / T).exp()
    kyy = k(y, y) + sigma_noise * torch.eye(y.shape[1], dtype=torch.float64)[None]
    try:
        sol = torch.linalg.solve(kyy, f)
    except RuntimeError:   # this image's torch CPU LU can fail for n >= 588 ("Pivots given to lu_solve ..."): numpy's LAPACK
        import numpy as np
        sol = torch.from_numpy(np.linalg.solve(kyy.numpy(), f.numpy()))
    return k(x, y).matmul(sol).float()


def gp_forward(sd, scale, x, y, sigma_noise=0.1):
    """GP.forward with no_cov=True (dkm.py:340-370): mu = K_xy (K_yy + sigma I)^-1 f, f = cos(8 pi pos_conv(coords)).
    (The reference also evaluates K_xx and discards it.)"""
    b, c, h1, w1 = x.shape
    _, _, h2, w2 = y.shape
    f = torch.cos(8 * math.pi * _conv(sd, f"decoder.gps.{scale}.pos_conv", grid_coords(b, h2, w2)))
    xr, yr, fr = (t.flatten(2).transpose(1, 2) for t in (x, y, f))
    if GP_FP64:
        return gp_posterior_fp64(xr, yr, fr, sigma_noise).transpose(1, 2).reshape(b, -1, h1, w1)
    K_yy = cos_kernel(yr, yr)
    K_xy = cos_kernel(xr, yr)
    K_inv = torch.linalg.inv(K_yy + sigma_noise * torch.eye(h2 * w2)[None])
    mu = K_xy.matmul(K_inv.matmul(fr))
    return mu.transpose(1, 2).reshape(b, -1, h1, w1)


def rrb(sd, p, x):
    """dkm.py:171-199"""
    x = _conv(sd, p + ".conv1", x)
    r = F.relu(_bn(sd, p + ".bn", _conv(sd, p + ".conv2", x, 1, 1)))
    return F.relu(x + _conv(sd, p + ".conv3", r, 1, 1))


def cab(sd, p, x1, x2):
    """dkm.py:147-168: channel attention from the global average of cat[x1, x2]; x * x2 + x1"""
    g = torch.cat([x1, x2], 1).mean((2, 3), keepdim=True)
    g = torch.sigmoid(_conv(sd, p + ".conv2", F.relu(_conv(sd, p + ".conv1", g))))
    return g * x2 + x1


def dfn(sd, key, embeddings, feats, context):
    """DFN.forward (dkm.py:243-254) -> (flow [b,2,h,w], certainty [b,1,h,w], context)"""
    e = "decoder.embedding_decoder."
    feats = _conv(sd, f"{e}feat_input_modules.{key}", feats)
    emb = rrb(sd, f"{e}rrb_d.{key}", torch.cat([feats, embeddings], 1))
    context = rrb(sd, f"{e}rrb_u.{key}", cab(sd, f"{e}cab.{key}", context, emb))
    preds = _conv(sd, f"{e}terminal_module.{key}", context)
    return preds[:, -2:], preds[:, :-2], context


def local_correlation(f0, f1, r, flow):
    """local_correlation.py:5-40 with `flow` given (corr_in_other): a (2r+1)^2 bilinear window of f1 around the
    flow target of every f0 pixel, dotted with f0, / sqrt(c)."""
    b, c, h, w = f0.shape
    coords = flow.permute(0, 2, 3, 1)
    wy = torch.linspace(-2 * r / h, 2 * r / h, 2 * r + 1)
    wx = torch.linspace(-2 * r / w, 2 * r / w, 2 * r + 1)
    gy, gx = torch.meshgrid(wy, wx, indexing="ij")
    win = torch.stack((gx, gy), -1)[None].expand(b, 2 * r + 1, 2 * r + 1, 2).reshape(b, (2 * r + 1) ** 2, 2)
    coords = (coords[:, :, :, None] + win[:, None, None]).reshape(b, h, w * (2 * r + 1) ** 2, 2)
    wf = F.grid_sample(f1, coords, padding_mode="zeros", align_corners=False)[..., None].reshape(b, c, h, w, (2 * r + 1) ** 2)
    return torch.einsum("bchw,bchwk->bkhw", f0, wf) / (c ** 0.5)


def conv_refiner(sd, scale, x, y, flow):
    """ConvRefiner.forward (dkm.py:75-123) -> (certainty [b,1,h,w], displacement [b,2,h,w])"""
    p = f"decoder.conv_refiner.{scale}."
    b, c, hs, ws = x.shape
    _, emb_dim, radius = REFINER[scale]
    x_hat = F.grid_sample(y, flow.permute(0, 2, 3, 1), align_corners=False)
    emb = _conv(sd, p + "disp_emb", flow - grid_coords(b, hs, ws))
    parts = [x, x_hat, emb]
    if radius:
        parts.append(local_correlation(x, y, radius, flow))
    d = torch.cat(parts, 1)
    in_dim, hid = _refiner_dims(scale)
    for nm, ci in [("block1", in_dim)] + [(f"hidden_blocks.{i}", hid) for i in range(HIDDEN_BLOCKS)]:
        d = _conv(sd, f"{p}{nm}.0", d, 1, 2, groups=ci)
        d = F.relu(_bn(sd, f"{p}{nm}.1", d))
        d = _conv(sd, f"{p}{nm}.3", d)
    d = _conv(sd, p + "out_conv", d)
    return d[:, :-2], d[:, -2:]


def _up(t, size):
    return F.interpolate(t, size=size, align_corners=False, mode="bilinear")


def decoder(sd, f1, f2, upsample=False, dense_flow=None, dense_certainty=None):
    """Decoder.forward (dkm.py:451-534) -> {scale: {'dense_flow', 'dense_certainty'}}"""
    scales = ["8", "4", "2", "1"] if upsample else ["32", "16", "8", "4", "2", "1"]
    sizes = {s: f1[s].shape[-2:] for s in f1}
    h, w = sizes[1]
    b = f1[1].shape[0]
    coarsest = int(scales[0])
    old = torch.zeros(b, DFN_DIM, *sizes[coarsest])
    if not upsample:
        dense_flow = grid_coords(b, *sizes[coarsest])
        dense_certainty = 0.0
    else:
        dense_flow = _up(dense_flow, sizes[coarsest])
        dense_certainty = _up(dense_certainty, sizes[coarsest])
    out = {}
    for s in scales:
        ins = int(s)
        a, c = f1[ins], f2[ins]
        if s in ("16", "32"):
            a, c = _conv(sd, f"decoder.proj.{s}", a), _conv(sd, f"decoder.proj.{s}", c)
            old = _up(old, sizes[ins])
            dense_flow, dense_certainty, old = dfn(sd, s, gp_forward(sd, s, a, c), a, old)
        if s in REFINER:
            dcert, disp = conv_refiner(sd, s, a, c, dense_flow)
            dense_flow = torch.stack((dense_flow[:, 0] + ins * disp[:, 0] / (4 * w),
                                      dense_flow[:, 1] + ins * disp[:, 1] / (4 * h)), 1)
            dense_certainty = dense_certainty + dcert
        out[ins] = {"dense_flow": dense_flow, "dense_certainty": dense_certainty}
        if s != "1":
            dense_flow = _up(dense_flow, sizes[ins // 2])
            dense_certainty = _up(dense_certainty, sizes[ins // 2])
    return out


def forward_symmetric(sd, query, support, upsample=False, corresps=None):
    """RegressionMatcher.forward_symmetric (dkm.py:637-646): one encoder pass on [q; s], decoder on ([q;s], [s;q])."""
    pyr = resnet50_pyramid(sd, torch.cat((query, support)))
    swapped = {s: torch.cat((f.chunk(2)[1], f.chunk(2)[0])) for s, f in pyr.items()}
    return decoder(sd, pyr, swapped, upsample=upsample, **(corresps or {}))


def match(sd, im1, im2, h_resized, w_resized, upsample_res=None):
    """RegressionMatcher.match, symmetric, non-batched, tensor inputs (dkm.py:654-752).
    -> warp [H, 2W, 4] (normalised x0, y0, x1, y1), certainty [H, 2W]"""
    hs, ws = h_resized, w_resized
    q, s = _up(im1, (hs, ws)), _up(im2, (hs, ws))
    cor = forward_symmetric(sd, q, s)
    if upsample_res is not None:
        hs, ws = upsample_res
    low = _up(cor[16]["dense_certainty"], (hs, ws))
    low = 0.5 * low * (low < 0)
    if upsample_res is not None:
        q, s = _up(im1, (hs, ws)), _up(im2, (hs, ws))
        cor = forward_symmetric(sd, q, s, upsample=True, corresps=cor[1])
    q2s = cor[1]["dense_flow"].permute(0, 2, 3, 1)
    cert = (cor[1]["dense_certainty"] - low).sigmoid()
    qc = grid_coords(1, hs, ws).permute(0, 2, 3, 1)
    wrong = (q2s.abs() > 1).sum(dim=-1) > 0
    cert[wrong[:, None]] = 0
    def black(im):
        m = (im[0, 0] < 0.03125) & (im[0, 1] < 0.03125) & (im[0, 2] < 0.03125)
        return F.interpolate(m.float()[None, None], size=(hs, ws), mode="nearest").bool()
    cert[torch.cat((black(im1), black(im2)), 0)] = 0
    q2s = torch.clamp(q2s, -1, 1)
    qts, stq = q2s.chunk(2)
    warp = torch.cat((torch.cat((qc, qts), -1), torch.cat((stq, qc), -1)), 2)
    cert = torch.cat(cert.chunk(2), 3)[:, 0]
    return warp[0], cert[0]


def kde(x, std=0.1):
    """kde.py:17-26"""
    return (-torch.cdist(x, x) ** 2 / (2 * std ** 2)).exp().sum(-1)


def sample(dense_matches, dense_certainty, num, thresh=0.05):
    """RegressionMatcher.sample, 'threshold_balanced' (dkm.py:583-620).  Draws from torch's global RNG exactly where
    the reference does: seed it identically for a bitwise comparison on the same device type."""
    cert = dense_certainty.clone()
    cert_ = dense_certainty.clone()
    cert[cert > thresh] = 1
    matches, cert, cert_ = dense_matches.reshape(-1, 4), cert.reshape(-1), cert_.reshape(-1)
    if not cert.sum():
        cert = cert + 1e-8
    good = torch.multinomial(cert, num_samples=min(4 * num, len(cert)), replacement=False)
    gm, gc = matches[good], cert_[good]
    density = kde(gm, 0.1)
    p = 1 / (density + 1)
    p[density < 10] = 1e-7
    bal = torch.multinomial(p, num_samples=min(num, len(gc)), replacement=False)
    return gm[bal], gc[bal]


def gim_dkm_adapter(sparse_matches, mconf, hw0, hw1):
    """trainer/lightning.py:134-156: normalised -> pixel coordinates, mconf > 0 filter."""
    (h0, w0), (h1, w1) = hw0, hw1
    k0 = torch.stack((w0 * (sparse_matches[:, 0] + 1) / 2, h0 * (sparse_matches[:, 1] + 1) / 2), -1)
    k1 = torch.stack((w1 * (sparse_matches[:, 2] + 1) / 2, h1 * (sparse_matches[:, 3] + 1) / 2), -1)
    mask = mconf > 0
    return {"mkpts0_f": k0[mask], "mkpts1_f": k1[mask], "m_bids": torch.where(mconf[None])[0], "mconf": mconf[mask]}


def seeded_pair(h, w, seed, shift=(6, 10)):
    """A textured image and a shifted + slightly brightened copy (real overlap, so the flow is meaningful)."""
    g = torch.Generator().manual_seed(seed)
    base = F.interpolate(torch.rand(1, 3, h // 8 + 4, w // 8 + 4, generator=g), size=(h + 32, w + 32), mode="bicubic",
                         align_corners=False).clamp(0.05, 1)
    base = (0.8 * base + 0.2 * torch.rand(1, 3, h + 32, w + 32, generator=g)).clamp(0.05, 1)
    im0 = base[:, :, 16:16 + h, 16:16 + w].contiguous()
    im1 = (base[:, :, 16 + shift[0]:16 + shift[0] + h, 16 + shift[1]:16 + shift[1] + w] * 0.95 + 0.02).contiguous()
    return im0, im1
