"""gim_roma on MI355X: the reference's `RoMa(img_size=[672])` / `RegressionMatcher` surface
(`networks/roma/roma.py:636-917, 1124-1266`) over hand-written HIP.

Drop-in contract:
  * `RoMa(img_size, **kwargs)` returns a module whose `state_dict()` has the reference's tensors (`encoder.cnn.layers.*`
    = torchvision vgg19_bn.features[:40], `decoder.embedding_decoder.*`, `decoder.gps.16.*`, `decoder.proj.*`,
    `decoder.conv_refiner.*`), so gim_roma checkpoints load with the reference's prefix rule (`demo.py:365-371`);
  * the DINOv2 ViT-L/14 weights are NOT part of that state_dict in the reference either (`self.dinov2_vitl14 =
    [dinov2_vitl14]`, roma.py:612); the reference downloads them inside the constructor, this module takes them as a
    plain state dict: `RoMa(img_size, dinov2_weights=sd)` or `model.load_dinov2(sd)` (names of `dino.py`'s
    `vit_large`: cls_token, pos_embed, patch_embed.proj.*, blocks.N.*, norm.*);
  * `h_resized, w_resized, upsample_preds, upsample_res, symmetric, sample_thresh, attenuate_cert` are plain
    attributes read at call time;
  * `match(im_A, im_B)` takes [1,3,H,W] fp32 tensors and returns `(warp [Hs, 2Ws, 4], certainty [Hs, 2Ws])`,
    `sample(dense_matches, dense_certainty, num)` returns `([n,4], [n])` like roma.py:680-714.
  * built: symmetric, non-batched matching with or without the upsampling pass (what gim runs).

Every stage of `match()` is a libgimhip launch: VGG / patch-embed / 1x1 convolutions and all Linear layers of the two
transformers on the implicit-GEMM kernel (LayerScale folded into proj / fc2, the residual add in its epilogue, exact
GELU in fc1's epilogue), attention on the flash SDPA kernel (DINOv2 16 x 64, decoder 8 x 128), the GP on the fp32-MFMA
products + fp64 Cholesky of `csrc/gp_solve.hip`, the refiners on `csrc/dkm.hip`.  Constant tables (bicubic-resized
position embedding, the GP's Fourier features) are built once per shape on the host with the reference's own fp32 ops.
No CPU / eager fallback.
"""
import math
import os

import torch

from ..precision import resolve as resolve_precision
from ..switches import flag, tri_flag
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .._lib import ACT_GELU, ACT_NONE, ACT_RELU, GIM_BF16, GIM_F16, GIM_F32, GimHipError
from ..dkm.dkm import _bn_after_bias, balanced_sample
from ..packing import cstore, pack_conv, torch_dtype

VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M"]      # vgg19_bn.features[:40]
REFINER = {"16": (512, 128, 7), "8": (512, 64, 3), "4": (256, 32, 2), "2": (64, 16, None), "1": (9, 6, None)}
PROJ = {"16": (1024, 512), "8": (512, 512), "4": (256, 256), "2": (128, 64), "1": (64, 9)}
GP_DIM, DEC_DIM, DEC_HEADS, DEC_BLOCKS, CLS_RES, HIDDEN_BLOCKS = 512, 1024, 8, 5, 64, 8
VIT_DIM, VIT_DEPTH, VIT_HEADS, VIT_PATCH, VIT_GRID = 1024, 24, 16, 14, 37                      # vit_large, img_size 518


def _refiner_dims(scale):
    c, e, r = REFINER[scale]
    in_dim = 2 * c + e + ((2 * r + 1) ** 2 if r else 0)
    return in_dim, {"2": 128 + 16, "1": 24}.get(scale, in_dim)


# ---------------------------------------------------------------------------------------- parameter containers
class _VGG19(nn.Module):
    def __init__(self):
        super().__init__()
        layers, ci = [], 3
        for v in VGG_CFG:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(ci, v, 3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
                ci = v
        self.layers = nn.ModuleList(layers)


class _Encoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.cnn = _VGG19()


class _Attention(nn.Module):
    def __init__(self, dim, qkv_bias):
        super().__init__()
        self.qkv = nn.Linear(dim, 3 * dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.fc1 = nn.Linear(dim, 4 * dim)
        self.fc2 = nn.Linear(4 * dim, dim)


class _Block(nn.Module):
    """dino.py:114-168 `Block(dim, heads)` as the decoder builds it: qkv_bias=False, no LayerScale"""

    def __init__(self, dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _Attention(dim, False)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _Mlp(dim)


class _TransformerDecoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.blocks = nn.Sequential(*[_Block(DEC_DIM) for _ in range(DEC_BLOCKS)])
        self.to_out = nn.Linear(DEC_DIM, CLS_RES ** 2 + 1)


class _GP(nn.Module):
    def __init__(self):
        super().__init__()
        self.pos_conv = nn.Conv2d(2, GP_DIM, 1)


def _block(ci, co):
    return nn.Sequential(nn.Conv2d(ci, co, 5, 1, 2, groups=ci), nn.BatchNorm2d(co, momentum=0.01), nn.ReLU(inplace=True), nn.Conv2d(co, co, 1))


class _ConvRefiner(nn.Module):
    def __init__(self, scale):
        super().__init__()
        in_dim, hid = _refiner_dims(scale)
        self.block1 = _block(in_dim, hid)
        self.hidden_blocks = nn.Sequential(*[_block(hid, hid) for _ in range(HIDDEN_BLOCKS)])
        self.out_conv = nn.Conv2d(hid, 3, 1)
        self.disp_emb = nn.Conv2d(2, REFINER[scale][1], 1)


class _Decoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.embedding_decoder = _TransformerDecoder()
        self.gps = nn.ModuleDict({"16": _GP()})
        self.proj = nn.ModuleDict({s: nn.Sequential(nn.Conv2d(ci, co, 1, 1), nn.BatchNorm2d(co)) for s, (ci, co) in PROJ.items()})
        self.conv_refiner = nn.ModuleDict({s: _ConvRefiner(s) for s in REFINER})


class _Workspace:
    """scratch rows of one transformer pass (activation dtype T): LayerNorm output, qkv, V^T, attention output, MLP hidden"""

    def __init__(self, nb, n, dim, tdt, dev):
        R = nb * n
        self.Sp = (n + 63) // 64 * 64
        self.xn = torch.empty(R, dim, dtype=tdt, device=dev)
        self.qkv = torch.empty(R, 3 * dim, dtype=tdt, device=dev)
        self.vt = torch.empty(nb, dim, self.Sp, dtype=tdt, device=dev)
        self.att = torch.empty(R, dim, dtype=tdt, device=dev)
        self.hid = torch.empty(R, 4 * dim, dtype=tdt, device=dev)


class RegressionMatcher(nn.Module):
    def __init__(self, h=448, w=448, sample_mode="threshold_balanced", upsample_preds=True, symmetric=True, name=None,
                 attenuate_cert=True, precision=None, dinov2_weights=None):
        super().__init__()
        self.attenuate_cert = attenuate_cert
        self.encoder = _Encoder()
        self.decoder = _Decoder()
        self.name = name
        self.w_resized, self.h_resized = w, h
        self.sample_mode = sample_mode
        self.upsample_preds = upsample_preds
        self.upsample_res = (14 * 16 * 6, 14 * 16 * 6)
        self.symmetric = symmetric
        self.sample_thresh = 0.05
        # round 6: IEEE fp16 is gim_roma's default 16-bit mode (gim_amd/precision.py has the argument; bf16 moves the warp by ~3 px at 560 x 560,
        # fp16 stays < 5e-6 of scale from the fp32 mode: tests/test_gpu_roma.py, tests/test_gpu_dense_fullsize.py); `_fp16_checked` = the
        # output check of match_batch below has passed once for the current weights
        self.precision = resolve_precision(precision, "gim_roma", default="fp16")
        self._fp16_default = precision is None and self.precision == "fp16"
        self._fp16_checked = False
        # GP posterior entirely in fp64 (kernel entries, Cholesky, products; csrc/gp_solve.hip: gim_gp_posterior_f64).  None = in
        # the fp32 parity mode only: the system's condition number (~2e4) turns fp32 rounding of the kernel ENTRIES into ~1e-4 of mu,
        # the one term of the engine's deviation that is not the reference's own (tests/test_gpu_gp_pins.py)
        self.gp_exact = tri_flag("gp_exact")
        # 16-bit modes: the 144- and 24-channel ConvRefiner blocks (scales 2 and 1, both passes) as ONE launch each (gim_dwconv5x5_pw, round 5)
        self.refiner_fused = flag("refiner_fused", True)
        self._dino = [None]          # a list, like roma.py:612: the ViT is not a registered sub-module / not in state_dict()
        self._packed = None
        self._tables = {}
        if dinov2_weights is not None:
            self.load_dinov2(dinov2_weights)

    # ---- weights -------------------------------------------------------------------------------------------------
    def load_dinov2(self, state_dict):
        """DINOv2 ViT-L/14 weights by `dino.py` name; replaces the download of roma.py:596-604."""
        need = ["cls_token", "pos_embed", "patch_embed.proj.weight", "patch_embed.proj.bias", "norm.weight", "norm.bias"]
        for i in range(VIT_DEPTH):
            need += [f"blocks.{i}.{k}" for k in ("norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight",
                                                "attn.proj.bias", "ls1.gamma", "norm2.weight", "norm2.bias", "mlp.fc1.weight",
                                                "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias", "ls2.gamma")]
        missing = [k for k in need if k not in state_dict]
        if missing:
            raise GimHipError(f"DINOv2 weights: {len(missing)} tensors missing, e.g. {missing[:3]}")
        if tuple(state_dict["pos_embed"].shape) != (1, VIT_GRID ** 2 + 1, VIT_DIM):
            raise GimHipError(f"DINOv2 weights: pos_embed {tuple(state_dict['pos_embed'].shape)} is not ViT-L/14 @518")
        self._dino[0] = {k: state_dict[k].detach().float().cpu() for k in need}
        self._fp16_checked = False
        self._packed = None
        self._tables = {}

    def load_state_dict(self, state_dict, *a, **k):
        self._packed = None
        self._fp16_checked = False
        self._tables = {}
        return super().load_state_dict(state_dict, *a, **k)

    def _apply(self, fn, *a, **k):
        self._packed = None
        self._tables = {}
        return super()._apply(fn, *a, **k)

    def get_output_resolution(self):
        return self.upsample_res if self.upsample_preds else (self.h_resized, self.w_resized)

    # ---- one-time packing ----------------------------------------------------------------------------------------
    @staticmethod
    def _pack_block(P, p, dt, device, n1, qkv_w, qkv_b, proj_w, proj_b, n2, fc1_w, fc1_b, fc2_w, fc2_b, ls1=None, ls2=None):
        f = lambda t: t.detach().float().contiguous().to(device)  # noqa: E731
        P[p + "n1"], P[p + "n2"] = (f(n1[0]), f(n1[1])), (f(n2[0]), f(n2[1]))
        P[p + "qkv"] = pack_conv(qkv_w, None, dt, device, bias=qkv_b)
        if ls1 is not None:     # LayerScale (dino.py:155-160): x + gamma * f(x) -> gamma folded into the producing Linear
            proj_w, proj_b = proj_w.detach().float() * ls1.float()[:, None], proj_b.detach().float() * ls1.float()
            fc2_w, fc2_b = fc2_w.detach().float() * ls2.float()[:, None], fc2_b.detach().float() * ls2.float()
        P[p + "proj"] = pack_conv(proj_w, None, dt, device, bias=proj_b)
        P[p + "fc1"] = pack_conv(fc1_w, None, dt, device, bias=fc1_b)
        P[p + "fc2"] = pack_conv(fc2_w, None, dt, device, bias=fc2_b)

    def _prepack(self, device):
        if self._dino[0] is None:
            raise GimHipError("RoMa needs the DINOv2 ViT-L/14 weights: RoMa(img_size, dinov2_weights=sd) or model.load_dinov2(sd)")
        dt = {"bf16": GIM_BF16, "fp16": GIM_F16, "fp32": GIM_F32}[self.precision]
        P = {}
        layers = self.encoder.cnn.layers
        idx = 0
        for v in VGG_CFG:
            if v == "M":
                idx += 1
                continue
            conv, bn = layers[idx], layers[idx + 1]
            P[f"vgg{idx}"] = pack_conv(conv.weight, _bn_after_bias(bn, conv.bias), dt, device, pad=1,
                                       cin_pad=cstore(conv.in_channels, dt))
            idx += 3
        d = self._dino[0]
        P["patch"] = pack_conv(d["patch_embed.proj.weight"], None, dt, device, stride=VIT_PATCH, cin_pad=cstore(3, dt),
                               bias=d["patch_embed.proj.bias"])
        for i in range(VIT_DEPTH):
            b = f"blocks.{i}."
            self._pack_block(P, f"vit{i}.", dt, device, (d[b + "norm1.weight"], d[b + "norm1.bias"]), d[b + "attn.qkv.weight"],
                             d[b + "attn.qkv.bias"], d[b + "attn.proj.weight"], d[b + "attn.proj.bias"],
                             (d[b + "norm2.weight"], d[b + "norm2.bias"]), d[b + "mlp.fc1.weight"], d[b + "mlp.fc1.bias"],
                             d[b + "mlp.fc2.weight"], d[b + "mlp.fc2.bias"], d[b + "ls1.gamma"], d[b + "ls2.gamma"])
        P["vit.norm"] = (d["norm.weight"].to(device), d["norm.bias"].to(device))
        dec = self.decoder
        for i, blk in enumerate(dec.embedding_decoder.blocks):
            self._pack_block(P, f"dec{i}.", dt, device, (blk.norm1.weight, blk.norm1.bias), blk.attn.qkv.weight, blk.attn.qkv.bias,
                             blk.attn.proj.weight, blk.attn.proj.bias, (blk.norm2.weight, blk.norm2.bias), blk.mlp.fc1.weight,
                             blk.mlp.fc1.bias, blk.mlp.fc2.weight, blk.mlp.fc2.bias)
        to_out = dec.embedding_decoder.to_out
        P["to_out"] = pack_conv(to_out.weight, None, dt, device, bias=to_out.bias)
        for s, seq in dec.proj.items():
            P["proj" + s] = pack_conv(seq[0].weight, _bn_after_bias(seq[1], seq[0].bias), dt, device)
        for s, ref in dec.conv_refiner.items():
            in_dim, hid = _refiner_dims(s)
            blocks = [ref.block1] + list(ref.hidden_blocks)
            for i, blk in enumerate(blocks):
                conv, bn, _, pw = blk
                ci = in_dim if i == 0 else hid
                cpad = cstore(hid, dt)
                sc = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
                W = torch.zeros(25, cpad)
                W[:, :hid] = conv.weight.detach().float().reshape(hid, 25).t().cpu()
                scale, shift = torch.zeros(cpad), torch.zeros(cpad)
                scale[:hid] = sc.cpu()
                shift[:hid] = (bn.bias.detach().float() + (conv.bias.detach().float() - bn.running_mean.detach().float()) * sc).cpu()
                P[f"cr{s}.{i}.dw"] = (W.to(device), scale.to(device), shift.to(device), ci, hid)
                P[f"cr{s}.{i}.pw"] = pack_conv(pw.weight, None, dt, device, cin_pad=cpad, bias=pw.bias)
                if dt != GIM_F32 and cpad in (24, 32, 144) and ci == hid:   # refiner blocks that fit one launch (gim_dwconv5x5_pw): dw 5x5 + BN + ReLU + 1x1
                    npc, kp = (160, 144) if cpad == 144 else (32, 32)
                    wf, bf = torch.zeros(npc, kp), torch.zeros(npc)
                    wf[:hid, :hid] = pw.weight.detach().float().reshape(hid, hid).cpu()
                    bf[:hid] = pw.bias.detach().float().cpu()
                    P[f"cr{s}.{i}.pwf"] = (wf.to(device).to(torch_dtype(dt)).contiguous(), bf.to(device))
            P[f"cr{s}.out"] = pack_conv(ref.out_conv.weight, None, dt, device, cin_pad=cstore(hid, dt), bias=ref.out_conv.bias)
            P[f"cr{s}.emb"] = (ref.disp_emb.weight.detach().float().reshape(-1, 2).contiguous().to(device),
                               ref.disp_emb.bias.detach().float().contiguous().to(device))
            P[f"cr{s}.cin_store"] = cstore(in_dim, dt)
        self._packed = (P, dt, device)

    def _gp_features(self, h, w, device):
        """f = cos(8 pi pos_conv(coords)) of GP.get_pos_enc (roma.py:94-108): constant per (h, w); built once on the host
        with the reference's fp32 ops, cached on the device as rows [h*w, 512]."""
        key = ("gp", h, w, str(device))
        if key not in self._tables:
            ys = torch.linspace(-1 + 1 / h, 1 - 1 / h, h)
            xs = torch.linspace(-1 + 1 / w, 1 - 1 / w, w)
            gy, gx = torch.meshgrid(ys, xs, indexing="ij")
            coords = torch.stack((gx, gy))[None]
            pc = self.decoder.gps["16"].pos_conv
            f = torch.cos(8 * math.pi * F.conv2d(coords, pc.weight.detach().float().cpu(), pc.bias.detach().float().cpu()))
            self._tables[key] = f[0].permute(1, 2, 0).reshape(h * w, GP_DIM).contiguous().to(device)
        return self._tables[key]

    def _pos_table(self, hs, ws, device):
        """interpolate_pos_encoding (dino.py:457-488) -> (cls row [1024] = cls_token + pos[0], patch rows [h0*w0, 1024]).
        The reference passes (w, h) = x.shape[2:] = (H, W), so the bicubic scale factors are ((h0 + .1) / 37, (w0 + .1) / 37)
        on the (row, column) axes of the 37 x 37 table."""
        key = ("pos", hs, ws, str(device))
        if key not in self._tables:
            d = self._dino[0]
            pe = d["pos_embed"]
            h0, w0 = hs // VIT_PATCH, ws // VIT_PATCH
            if h0 * w0 == VIT_GRID ** 2 and hs == ws:
                patch = pe[0, 1:]
            else:
                g = VIT_GRID
                patch = F.interpolate(pe[:, 1:].reshape(1, g, g, -1).permute(0, 3, 1, 2),
                                      scale_factor=((h0 + 0.1) / g, (w0 + 0.1) / g), mode="bicubic")
                assert patch.shape[-2:] == (h0, w0)
                patch = patch.permute(0, 2, 3, 1).reshape(h0 * w0, -1)
            self._tables[key] = ((d["cls_token"][0, 0] + pe[0, 0]).contiguous().to(device), patch.contiguous().to(device))
        return self._tables[key]

    # ---- stages ---------------------------------------------------------------------------------------------------
    def _vgg(self, P, x):
        """VGG19.forward (roma.py:144-152): activations before each max-pool -> {1: 64, 2: 128, 4: 256, 8: 512 channels}"""
        feats, scale, idx = {}, 1, 0
        for v in VGG_CFG:
            if v == "M":
                feats[scale] = x
                scale *= 2
                if scale <= 8:
                    x = ops.maxpool2x2(x)
                idx += 1
                continue
            x = ops.conv2d(x, P[f"vgg{idx}"], ACT_RELU)
            idx += 3
        return feats

    @staticmethod
    def _vit_block(P, p, x32, ws, nb, n, heads, eps):
        """one pre-norm transformer block (dino.py:143-168, eval) on the fp32 residual stream x32 [nb*n, C], in place"""
        C = x32.shape[1]
        R = nb * n
        ops.layernorm_act(x32, *P[p + "n1"], ws.xn, ACT_NONE, eps)
        ops.linear(ws.xn, P[p + "qkv"], ws.qkv)
        ops.lg_transpose(ws.qkv[:, 2 * C:], ws.vt, nb, n, ws.Sp, C)
        ops.sdpa(ws.qkv[:, :C], ws.qkv[:, C:2 * C], ws.vt, ws.att, nb, heads, n, n, ws.Sp, D=C // heads)
        ops.conv_rows(ws.att, P[p + "proj"], (1, 1, R, 1, R), x32, ACT_NONE, x32)           # x += [ls1 *] proj(attn)
        ops.layernorm_act(x32, *P[p + "n2"], ws.xn, ACT_NONE, eps)
        ops.linear(ws.xn, P[p + "fc1"], ws.hid, ACT_GELU)
        ops.conv_rows(ws.hid, P[p + "fc2"], (1, 1, R, 1, R), x32, ACT_NONE, x32)            # x += [ls2 *] fc2(gelu(fc1))

    def _dino_features(self, P, dt, x):
        """forward_features(x)['x_norm_patchtokens'] (dino.py:490-540, roma.py:624-631): x NHWC [nb,hs,ws,cpad] ->
        [nb, hs/14, ws/14, 1024]"""
        tdt = torch_dtype(dt)
        nb, hs, ws, cp = x.shape
        if hs % VIT_PATCH or ws % VIT_PATCH:
            raise GimHipError(f"h_resized / w_resized must be multiples of {VIT_PATCH} (DINOv2 patch size), got {(hs, ws)}")
        dev = x.device
        h0, w0 = hs // VIT_PATCH, ws // VIT_PATCH
        n = h0 * w0 + 1
        cls_row, pos = self._pos_table(hs, ws, dev)
        tok = torch.empty(nb, n, VIT_DIM, dtype=torch.float32, device=dev)
        tok[:, 0] = cls_row
        for b in range(nb):      # patch embedding (14x14 / stride 14 conv) + position rows in the epilogue's residual add
            ops.conv_rows(x[b].view(hs * ws, cp), P["patch"], (1, hs, ws, h0, w0), tok[b, 1:], ACT_NONE, pos)
        x32 = tok.view(nb * n, VIT_DIM)
        wsp = _Workspace(nb, n, VIT_DIM, tdt, dev)
        for i in range(VIT_DEPTH):
            self._vit_block(P, f"vit{i}.", x32, wsp, nb, n, VIT_HEADS, 1e-6)
        feat = torch.empty(nb, h0, w0, VIT_DIM, dtype=tdt, device=dev)
        for b in range(nb):
            ops.layernorm_act(tok[b, 1:], *P["vit.norm"], feat[b].view(h0 * w0, VIT_DIM), ACT_NONE, 1e-6)
        return feat

    def _encode(self, P, dt, x, upsample=False):
        """CNNandDinov2.forward (roma.py:617-633)"""
        feats = self._vgg(P, x)
        if not upsample:
            feats[16] = self._dino_features(P, dt, x)
        return feats

    def _project(self, P, dt, feat, s, out32=False):
        """proj[s] = 1x1 conv + BatchNorm (roma.py:1220-1234) -> NHWC [nb,h,w,cstore(co)]"""
        nb, h, w, _ = feat.shape
        if not out32:
            return ops.conv2d(feat, P["proj" + s])
        a32 = torch.zeros(nb * h * w + 64, P["proj" + s].n_store, dtype=torch.float32, device=feat.device)
        ops.linear(feat.view(nb * h * w, feat.shape[3]), P["proj" + s], a32)
        return a32

    def _gp(self, a32, nb, h, w, out):
        """GP.forward, no_cov (roma.py:110-136) for all nb = 2 * pairs directions (image d against image (d + nb/2) % nb).
        a32: fp32 rows [nb*hw (+64 slack), 512] of the projected features; writes mu into `out` (row view [nb*hw, 512])."""
        dev = a32.device
        n = h * w
        half = nb // 2
        exact = (self.precision == "fp32") if self.gp_exact is None else self.gp_exact
        if exact and out.dtype == torch.float32:
            X = a32[:nb * n].view(nb, n, 512)
            ops.gp_posterior_f64(X, X.roll(-half, 0).contiguous(), self._gp_features(h, w, dev), out, 0.2, 1e-6, 0.1)   # support of direction b: image (b + half) % nb
            return
        nrm = ops.row_norms(a32[:nb * n], 512)
        ld = (n + 63) // 64 * 64
        npad = (n + 31) // 32 * 32
        Kyy = torch.zeros(nb, n, ld, dtype=torch.float32, device=dev)
        Kxy = torch.zeros(nb, n, max(ld, npad), dtype=torch.float32, device=dev)
        for b in range(nb):
            o = (b + half) % nb
            ops.matmul_nt(a32[o * n:(o + 1) * n], a32[o * n:], n, Kyy[b])
            ops.matmul_nt(a32[b * n:(b + 1) * n], a32[o * n:], n, Kxy[b])
        ny = nrm.view(nb, n).roll(-half, 0).contiguous().view(-1)
        ops.cos_kernel_finish(Kyy.view(nb * n, ld), ny, ny, nb, n, n, 0.2, 1e-6, 0.1)        # K_yy + sigma_noise I
        ops.cos_kernel_finish(Kxy.view(nb * n, Kxy.shape[2]), nrm, ny, nb, n, n, 0.2, 1e-6, 0.0)
        f = self._gp_features(h, w, dev)
        Xt = ops.gp_solve(Kyy, f[None].expand(nb, n, GP_DIM).contiguous(), npad)
        for b in range(nb):
            ops.matmul_nt(Kxy[b][:, :npad], Xt[b], GP_DIM, out[b * n:(b + 1) * n])

    def _coarse(self, P, dt, feat16):
        """scale 16 of Decoder.forward (roma.py:263-296): proj -> GP -> TransformerDecoder -> cls_to_flow_refine
        -> (projected features a [nb,h,w,512], flow [nb,h,w,2], certainty [nb,h,w,1])"""
        tdt = torch_dtype(dt)
        nb, h, w, _ = feat16.shape
        n, dev = h * w, feat16.device
        a32 = self._project(P, dt, feat16, "16", out32=True)
        x32 = torch.empty(nb * n, DEC_DIM, dtype=torch.float32, device=dev)
        self._gp(a32, nb, h, w, x32[:, :GP_DIM])
        ops.cast_rows(a32[:nb * n], x32[:, GP_DIM:])                       # tokens = cat(gp_posterior, features)
        if dt == GIM_F32:
            a = a32[:nb * n].view(nb, h, w, 512)
        else:
            a = torch.empty(nb, h, w, 512, dtype=tdt, device=dev)
            ops.cast_rows(a32[:nb * n], a.view(nb * n, 512))
        wsp = _Workspace(nb, n, DEC_DIM, tdt, dev)
        for i in range(DEC_BLOCKS):
            self._vit_block(P, f"dec{i}.", x32, wsp, nb, n, DEC_HEADS, 1e-5)
        ops.cast_rows(x32, wsp.xn)
        logits = torch.empty(nb * n, P["to_out"].n_store, dtype=torch.float32, device=dev)
        ops.linear(wsp.xn, P["to_out"], logits)
        flow, cert = ops.cls_to_flow(logits, nb, h, w, CLS_RES ** 2)
        return a, flow, cert

    def _refine(self, P, s, dt, x, y, flow, cert, ins, full_hw, scale_factor):
        """ConvRefiner.forward + the flow / certainty update of Decoder.forward (roma.py:529-580, 318-331)."""
        tdt = torch_dtype(dt)
        b, h, w, _ = x.shape
        c, e, r = REFINER[s]
        in_dim, hid = _refiner_dims(s)
        cs = P[f"cr{s}.cin_store"]
        dev = x.device
        g = 8 if dt in (GIM_BF16, GIM_F16) else 4
        ew, eb = P[f"cr{s}.emb"]
        ew = ew * (40.0 / 32.0 * scale_factor)            # disp_emb(40/32 * scale_factor * (flow - coords)), roma.py:545-547
        if c % g == 0:
            D = torch.zeros(b, h, w, cs, dtype=tdt, device=dev)
            rows = D.view(b * h * w, cs)
            D[..., :c].copy_(x[..., :c])
            ops.grid_sample(y, flow, rows[:, c:2 * c])
            ops.dkm_disp_emb(flow, ew, eb, rows[:, 2 * c:])
            if r:
                ops.local_corr(x, y, flow, r, rows[:, 2 * c + e:])
        else:  # scale 1: 9 projected channels (stored with padding) -> assemble the 24-channel input with copies
            xh = torch.empty(b * h * w, x.shape[3], dtype=tdt, device=dev)
            ops.grid_sample(y, flow, xh)
            emb = torch.empty(b * h * w, cstore(e, dt), dtype=tdt, device=dev)
            ops.dkm_disp_emb(flow, ew, eb, emb)
            D = torch.zeros(b, h, w, cs, dtype=tdt, device=dev)
            D[..., :c].copy_(x[..., :c])
            D[..., c:2 * c].copy_(xh.view(b, h, w, -1)[..., :c])
            D[..., 2 * c:2 * c + e].copy_(emb.view(b, h, w, -1)[..., :e])
        d = D
        for i in range(1 + HIDDEN_BLOCKS):
            W_, sc, sh, ci, co = P[f"cr{s}.{i}.dw"]
            pwf = P.get(f"cr{s}.{i}.pwf") if self.refiner_fused else None
            if pwf is not None and d.shape[3] == W_.shape[1] and d.is_contiguous():
                d = ops.dwconv5x5_pw(d, W_, sc, sh, *pwf)   # the whole block in one launch: the depthwise output never leaves the CU
                continue
            d = ops.dwconv5x5_bn_relu(d, W_, sc, sh, ci, co)
            d = ops.conv2d(d, P[f"cr{s}.{i}.pw"])
        out = torch.empty(b * h * w, P[f"cr{s}.out"].n_store, dtype=torch.float32, device=dev)
        ops.linear(d.view(b * h * w, d.shape[3]), P[f"cr{s}.out"], out)
        ops.dkm_flow_update(flow, cert, out, ins / (4.0 * full_hw[1]), ins / (4.0 * full_hw[0]), roma_layout=True)

    def _decode(self, P, dt, f1, upsample=False, flow=None, cert=None, scale_factor=1.0):
        """Decoder.forward on the symmetric pair (f2 = f1 with the two images swapped) -> {scale: (flow, certainty)}"""
        scales = ["8", "4", "2", "1"] if upsample else ["16", "8", "4", "2", "1"]
        sizes = {s: tuple(f1[s].shape[1:3]) for s in f1}
        full = sizes[1]
        nb = f1[1].shape[0]
        half = nb // 2
        coarsest = int(scales[0])
        if upsample:
            flow = ops.resize_bilinear(flow, sizes[coarsest])
            cert = ops.resize_bilinear(cert, sizes[coarsest])
        out = {}
        for s in scales:
            ins = int(s)
            if s == "16":
                a, flow, cert = self._coarse(P, dt, f1[16])
                out["gm"] = (flow.clone(), cert.clone())
            else:
                a = self._project(P, dt, f1[ins], s)
            self._refine(P, s, dt, a, torch.cat((a[half:], a[:half])), flow, cert, ins, full, scale_factor)
            out[ins] = (flow, cert)
            if s != "1":
                flow = ops.resize_bilinear(flow, sizes[ins // 2])
                cert = ops.resize_bilinear(cert, sizes[ins // 2])
        return out

    def _images(self, dt, im1, im2, hs, ws):
        """[B,3,H,W] x 2 -> NHWC [2B, hs, ws, cpad]: im_A first, then im_B (extract_backbone_features, roma.py:668-678)"""
        B = im1.shape[0]
        x = torch.empty(2 * B, hs, ws, cstore(3, dt), dtype=torch_dtype(dt), device=im1.device)
        ops.resize_image(im1, x, 0)
        ops.resize_image(im2, x, B)
        return x

    @torch.no_grad()
    def match(self, im_A_path, im_B_path, *args, batched=False):
        """RegressionMatcher.match (roma.py:816-917), tensor inputs as gim calls it (`demo.py:433`, `lightning.py:135`):
        [1,3,H,W] x 2 -> (warp [Hs, 2Ws, 4], certainty [Hs, 2Ws])."""
        if batched or not self.symmetric:
            raise NotImplementedError("gim runs RoMa symmetric and non-batched; use match_batch for several pairs")
        if im_A_path.dim() != 4 or im_A_path.shape[0] != 1:
            raise GimHipError(f"match() takes [1,3,H,W] images, got {tuple(im_A_path.shape)}")
        warp, certainty = self.match_batch(im_A_path, im_B_path)
        return warp[0], certainty[0]

    @torch.no_grad()
    def match_batch(self, ims_A, ims_B):
        """B independent pairs in one pass ([B,3,H,W] x 2 -> warp [B,Hs,2Ws,4], certainty [B,Hs,2Ws]); result b equals
        `match(ims_A[b:b+1], ims_B[b:b+1])` (the engine's batching, like gim_amd.dkm)."""
        if not self.symmetric:
            raise NotImplementedError("only symmetric matching is built")
        im1, im2 = ims_A, ims_B
        if not im1.is_cuda:
            raise GimHipError("gim_amd RoMa needs device (cuda/HIP) tensors: there is no CPU fallback")
        if im1.dim() != 4 or im1.shape[1] != 3 or im1.shape != im2.shape or not 1 <= im1.shape[0] <= 4:
            raise GimHipError(f"match takes two [B,3,H,W] batches of equal shape with B <= 4, got {tuple(im1.shape)} / {tuple(im2.shape)}")
        dev = im1.device
        B = im1.shape[0]
        want = {"bf16": GIM_BF16, "fp16": GIM_F16, "fp32": GIM_F32}[self.precision]
        if self._packed is None or self._packed[2] != dev or self._packed[1] != want:
            self._prepack(dev)
        P, dt, _ = self._packed
        im1, im2 = im1.contiguous().float(), im2.contiguous().float()
        hs, ws = self.h_resized, self.w_resized
        cor = self._decode(P, dt, self._encode(P, dt, self._images(dt, im1, im2, hs, ws)))
        stages = {"low": cor}
        if self.upsample_preds:
            hs, ws = self.upsample_res
        if self.attenuate_cert:
            low = ops.resize_bilinear(cor[16][1], (hs, ws))
        else:
            low = torch.zeros(2 * B, hs, ws, 1, dtype=torch.float32, device=dev)
        if self.upsample_preds:
            sf = math.sqrt(self.upsample_res[0] * self.upsample_res[1] / (self.w_resized * self.h_resized))
            pyr_hi = self._encode(P, dt, self._images(dt, im1, im2, hs, ws), upsample=True)
            cor = self._decode(P, dt, pyr_hi, upsample=True, flow=cor[1][0], cert=cor[1][1], scale_factor=sf)
            stages["high"] = cor
        flow, cert = cor[1]
        warp = torch.empty(B, hs, 2 * ws, 4, dtype=torch.float32, device=dev)
        certainty = torch.empty(B, hs, 2 * ws, dtype=torch.float32, device=dev)
        for b in range(B):
            ops.dkm_match_post((flow[b], flow[b + B]), (cert[b], cert[b + B]), (low[b], low[b + B]),
                               ops.dkm_black_mask(im1[b:b + 1], (hs, ws)), ops.dkm_black_mask(im2[b:b + 1], (hs, ws)), warp[b], certainty[b])
        self._debug = stages
        # fp16 as the DEFAULT mode carries a range check of what it hands out (an explicit precision='fp16' is the caller's decision):
        # every stored activation of this engine sits behind a BatchNorm / LayerNorm, so an overflow needs pathological weights -- it then
        # surfaces as inf / nan in the flow or certainty logits (no ReLU between the refiners' last convolution and these outputs).  One
        # reduction + host sync on the first calls after a weight change; a trip switches the module to bf16 for good and re-runs the batch
        if self._fp16_default and self.precision == "fp16" and not self._fp16_checked:
            if bool(torch.isfinite(warp).all()) and bool(torch.isfinite(certainty).all()):
                self._fp16_checked = True
            else:
                import warnings
                warnings.warn("gim_amd RoMa: non-finite outputs in the default fp16 mode (activations beyond 65504?); switching this module to bf16")
                self.precision, self._packed = "bf16", None
                return self.match_batch(ims_A, ims_B)
        return warp, certainty

    @torch.no_grad()
    def sample(self, dense_matches, dense_certainty, num=10000):
        """RegressionMatcher.sample (roma.py:680-714); the KDE runs on fp16-rounded coordinates like roma.py:1018-1023."""
        return balanced_sample(dense_matches, dense_certainty, num, self.sample_mode, self.sample_thresh, kde_half=True)


def RoMa(img_size, pretrained_backbone=False, **kwargs):
    """`networks/roma/roma.py:1124-1266`: img_size = [s] or [h, w]; kwargs go to RegressionMatcher (plus `precision`,
    `dinov2_weights`).  `pretrained_backbone` is ignored: checkpoints are loaded by the caller (`demo.py:365-371`)."""
    assert img_size is not None and isinstance(img_size, list) and len(img_size) <= 2
    if len(img_size) == 1:
        img_size = img_size * 2
    h, w = img_size
    kwargs.pop("device", None)
    return RegressionMatcher(h=h, w=w, **kwargs)


def random_dinov2_weights(dev, seed=0):
    """Synthetic ViT-L/14 weights with the magnitudes of a trained network (LayerScale ~0.2) for benchmarks -- there is no
    checkpoint in the container; generated on `dev` (300 M values), returned on the host by `dino.py` name."""
    g = torch.Generator(device=dev).manual_seed(seed)
    D = VIT_DIM

    def rn(*shape, s=1.0):
        return (torch.randn(*shape, generator=g, device=dev) * s).cpu()
    sd = {"cls_token": rn(1, 1, D, s=0.02), "pos_embed": rn(1, VIT_GRID ** 2 + 1, D, s=0.02),
          "patch_embed.proj.weight": rn(D, 3, 14, 14, s=1 / 24.0), "patch_embed.proj.bias": rn(D, s=0.02),
          "norm.weight": 1 + rn(D, s=0.1), "norm.bias": rn(D, s=0.02)}
    for i in range(VIT_DEPTH):
        b = f"blocks.{i}."
        for nm in ("norm1", "norm2"):
            sd[b + nm + ".weight"], sd[b + nm + ".bias"] = 1 + rn(D, s=0.1), rn(D, s=0.02)
        sd[b + "attn.qkv.weight"], sd[b + "attn.qkv.bias"] = rn(3 * D, D, s=1 / 32.0), rn(3 * D, s=0.02)
        sd[b + "attn.proj.weight"], sd[b + "attn.proj.bias"] = rn(D, D, s=1 / 32.0), rn(D, s=0.02)
        sd[b + "mlp.fc1.weight"], sd[b + "mlp.fc1.bias"] = rn(4 * D, D, s=1 / 32.0), rn(4 * D, s=0.02)
        sd[b + "mlp.fc2.weight"], sd[b + "mlp.fc2.bias"] = rn(D, 4 * D, s=1 / 64.0), rn(D, s=0.02)
        sd[b + "ls1.gamma"], sd[b + "ls2.gamma"] = 0.2 + rn(D, s=0.05), 0.2 + rn(D, s=0.05)
    return sd


@torch.no_grad()
def gim_roma_inference(model, data, num=5000):
    """`Trainer.gim_dkm_inference` (trainer/lightning.py:134-156) -- the same adapter serves gim_roma (lightning.py:125)."""
    from ..dkm.dkm import gim_dkm_inference
    return gim_dkm_inference(model, data, num)
