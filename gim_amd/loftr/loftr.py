"""gim_loftr on MI355X: the reference's `LoFTR` module surface (`networks/loftr/loftr.py:14-99`) over
hand-written HIP.

Drop-in contract kept:
  * `LoFTR(config)` takes the same lower-case dict (`demo.py:333-335`, `trainer/lightning.py:45-46`);
  * `state_dict()` has the reference's 375 keys / shapes, `load_state_dict` strips the `model.` /
    `matcher.` prefixes (`loftr.py:93-99`), so reference checkpoints load unchanged;
  * `model(data)` mutates `data` in place and returns None; keys, dtypes, shapes and insertion order
    follow SURVEY Appendix A2 (`conf_matrix` is produced lazily, see `LazyConfMatrix`).

What is different underneath: the nn.Module tree below only *holds parameters*.  `forward` never calls a
torch op on the data path: weights are pre-packed once (BN folded, K-contiguous, padded) and every stage
is a libgimhip kernel launched through ctypes on torch's current HIP stream.  There is no CPU / eager
fallback: without the HIP library the import fails, without a GPU tensor the call raises.

Precision modes (`config['precision']`, default env GIM_PRECISION or 'fp16'):
  'fp32'  fp32 operands on v_mfma_f32_32x32x2_f32 (exact fp32 products/accumulate) -- the parity mode;
  'bf16'  bf16 operands / fp32 accumulate for the backbone and the transformer GEMMs -- the throughput
          mode BASELINE config 2 names.  The token residual stream stays fp32.  The FIRST convolution reads the image
          as fp16 (`config['stem_fp16']`, default on): rounding the image and the 7x7 filters to 8
          significand bits in front of an edge-detecting (cancelling) convolution is HALF of this mode's deviation from
          the fp32 reference (tools/precision_emulation.py: index flip rate 1.95 % -> 0.98 % with the stem alone on fp16
          operands; same MFMA rate, same bytes) -- plainly rounded: the split (hi + lo) stem of the fp16 mode is off here unless
          `config['stem_split']` asks for it (no measurable parity gain under bf16 storage everywhere else, 0.07 ms per step);
  'fp16'  (default) IEEE fp16 operands / fp32 accumulate everywhere the bf16 mode uses bf16: same kernels in their second flavour
          (csrc/gim_common.h), same instruction counts and bytes (measured 2-3 % slower: lower clocks), 11 instead of 8
          significand bits per stored activation -- index flip rate 0.15-0.3 % against the fp32 oracle where bf16 has 0.7-1.3 %
          (0.47 % in the emulation).  It is the reference's own reduced-precision mode (its attention divides the values by
          their length "prevent fp16 overflow", submodules/attentions.py:42).  Range: |activation| < 65504 (BatchNorm-folded
          ResNet activations and LayerNorm'd tokens are O(1..100)); a checkpoint that overflows shows inf / nan in the
          outputs -- use 'bf16' for it.

Coarse similarity (`config['coarse_sim']`; default = the precision mode):
  'fp32'  similarity of the fp32 tokens with fp32-exact products: on the same features the mutual-NN indices equal the
          reference's fp32 arithmetic.  Always used by the fp32 mode; selectable in bf16 mode.
  'bf16'  (bf16 mode only, its default) similarity of the bf16 operand copy of the tokens on the bf16 MFMA.  Measured on
          match-rich 640x480 pairs against the fp32 oracle (profiles/r02_parity_probe.txt, tests/test_gpu_loftr_fullsize.py):
          index flip rate 1.7-2.5 % with EITHER setting -- the flips come from the bf16 backbone / transformer, the
          similarity's operand rounding adds nothing measurable -- at 0.47 instead of 1.17 ms per batch of 8.
          bench.py reports the flip rate of both settings next to the throughput.
"""
import collections
import math
import os

import torch
import torch.nn as nn

from .. import ops
from ..switches import flag
from .._lib import ACT_ELU1, ACT_LEAKY, ACT_NONE, ACT_RELU, GIM_BF16, GIM_F16, GIM_F32, GimHipError
from ..packing import (PackedStem, cstore, is_half, pack_bneck, pack_bneck_ds, pack_bneck_tail, pack_conv, pack_conv_split, pack_fine_fused,
                       pack_stem7x7, pack_token_mlp, pack_token_emit, split_channels, torch_dtype)

_DT = {"bf16": GIM_BF16, "fp16": GIM_F16, "fp32": GIM_F32}


# --------------------------------------------------------------------------------------------------
# parameter containers with the reference's names (never executed)
# --------------------------------------------------------------------------------------------------
def _conv(ci, co, k, stride=1):
    return nn.Conv2d(ci, co, kernel_size=k, stride=stride, padding=k // 2, bias=False)


class _Bottleneck(nn.Module):
    """parameter layout of resnet.py:71-107 (ResNet v1.5 bottleneck, stride on conv2)"""

    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 1)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv(planes, planes, 3, stride)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = _conv(planes, planes * 4, 1)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample
        self.stride = stride


class _ResNetEncoder(nn.Module):
    """resnet.py:129-167 with Bottleneck,[3,4,6,3]: conv1/bn1 + layer1..3 (no maxpool, no layer4/fc)"""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inpl = 64
        for li, (planes, nblk, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2)), start=1):
            blocks = []
            for bi in range(nblk):
                ds = None
                if bi == 0:
                    ds = nn.Sequential(_conv(inpl, planes * 4, 1, stride), nn.BatchNorm2d(planes * 4))
                blocks.append(_Bottleneck(inpl, planes, stride if bi == 0 else 1, ds))
                inpl = planes * 4
            setattr(self, f"layer{li}", nn.Sequential(*blocks))


class _ResNetFPN_8_2(nn.Module):
    """parameter layout of resnet.py:253-297"""

    def __init__(self, config):
        super().__init__()
        bd = config["block_dims"]
        self.encode = _ResNetEncoder()
        self.layer3_outconv = _conv(bd[5], bd[3], 1)
        self.layer2_outconv = _conv(bd[4], bd[3], 1)
        self.layer2_outconv2 = nn.Sequential(_conv(bd[3], bd[3], 3), nn.BatchNorm2d(bd[3]), nn.LeakyReLU(),
                                             _conv(bd[3], bd[2], 3))
        self.layer1_outconv = _conv(bd[3], bd[2], 1)
        self.layer1_outconv2 = nn.Sequential(_conv(bd[2], bd[2], 3), nn.BatchNorm2d(bd[2]), nn.LeakyReLU(),
                                             _conv(bd[2], bd[1], 3))
        for m in self.modules():  # resnet.py:291-296
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


class _EncoderLayer(nn.Module):
    """parameter layout of transformer.py:8-33"""

    def __init__(self, d_model, nhead):
        super().__init__()
        self.q_proj = nn.Linear(d_model, d_model, bias=False)
        self.k_proj = nn.Linear(d_model, d_model, bias=False)
        self.v_proj = nn.Linear(d_model, d_model, bias=False)
        self.merge = nn.Linear(d_model, d_model, bias=False)
        self.mlp = nn.Sequential(nn.Linear(d_model * 2, d_model * 2, bias=False), nn.ReLU(True),
                                 nn.Linear(d_model * 2, d_model, bias=False))
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)


class _LocalFeatureTransformer(nn.Module):
    """parameter layout of transformer.py:64-78"""

    def __init__(self, config):
        super().__init__()
        if config["attention"] != "linear":
            raise NotImplementedError("gim_loftr uses LinearAttention (config.py:22,41); 'full' is not built")
        self.d_model, self.nhead = config["d_model"], config["nhead"]
        self.layer_names = ["self", "cross"] * config["layer_names"]
        self.layers = nn.ModuleList([_EncoderLayer(self.d_model, self.nhead) for _ in self.layer_names])
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)


class LazyConfMatrix:
    """data['conf_matrix'] (coarse_matching.py:144).  Nothing on the inference path reads it (only the
    training losses do), and at 640x480 batch 8 it is 737 MB, so the engine materialises it on demand:
    `.get()` / `torch.as_tensor(obj.get())` launches the HIP kernel that writes [N,L,S] fp32."""

    def __init__(self, coarse_result, owner=None, generation=None):
        self._r = coarse_result
        self._t = None
        self._owner, self._gen = owner, generation

    def get(self):
        if self._t is None:
            if self._gen is not None and self._owner._generation != self._gen:
                raise RuntimeError("conf_matrix of an earlier forward: with HIP-graph replay the softmax "
                                   "statistics are overwritten by the next call; call .get() before it "
                                   "(or config['graph'] = False)")
            self._t = ops.coarse_conf_matrix(self._r)
        return self._t

    @property
    def shape(self):
        a = self._r.args
        return torch.Size([a.N, a.L, a.S])


def _precision_from(config):
    from ..precision import resolve
    return resolve(config.get("precision"), "loftr", default="fp16")


class LoFTR(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        if config["backbone_type"] != "ResNetFPN" or tuple(config["resolution"]) != (8, 2):
            # backbone/__init__.py:4-11: only ResNetFPN (8,2) is constructible in the reference either
            raise ValueError(f"LOFTR.BACKBONE_TYPE/RESOLUTION {config['backbone_type']} {config['resolution']} not supported.")
        if config["match_coarse"]["match_type"] != "dual_softmax":
            raise NotImplementedError("only match_type='dual_softmax' (the gim_loftr setting) is built")
        if config["fine_concat_coarse_feat"]:
            raise NotImplementedError("fine_concat_coarse_feat=True is not used by gim_loftr and is not built")
        self.precision = _precision_from(config)
        self.coarse_sim = self._check_sim((config.get("coarse_sim") or flag("coarse_sim", "") or self.precision).lower())
        # Switches (gim_amd/switches.py: config keys, or GIM_FLAGS="name=0,..." for A/B runs).  Each keeps the launch sequence a fused
        # kernel replaced as its cross-check; the defaults are what the benchmarks run.
        # 16-bit modes: the first convolution reads fp16 operands also in the bf16 mode (see the module docstring) ...
        self.stem_fp16 = flag("stem_fp16", True, config)
        # ... on split (hi + lo) operands -- image and 7x7 filters carried to 2^-22 instead of 2^-11
        # (profiles/r04_precision_sweep.txt: the stem alone is 57 % of the fp16 mode's mean |dmconf| and 3/4 of its index flips) ...
        # 'auto' (default): in the fp16 mode only.  The bf16 mode rounds every other activation to 8 bits, and the split buys it nothing
        # measurable (bench.py `parity.split_stem` / `plain_stem`: 10-12 vs 13 flips of 1485 in two runs of the round, the same mean |dmconf| 0.0082) for twice the
        # stem's MFMAs: -0.07 ms per step without it (profiles/r05_la_finalize.txt).  True / False force it either way.
        self.stem_split = flag("stem_split", "auto", config)
        # ... on its own kernel (gim_stem7x7: filter bank resident in LDS, every input patch staged once) instead of the implicit GEMM
        self.stem_kernel = flag("stem_kernel", True, config)
        # set once the fp16 mode's range guard tripped and the module fell back to bf16 (see forward)
        self.fp16_overflowed = False
        self.backbone = _ResNetFPN_8_2(config["resnetfpn"])
        self.loftr_coarse = _LocalFeatureTransformer(config["coarse"])
        self.loftr_fine = _LocalFeatureTransformer(config["fine"])
        self.W = config["fine_window_size"]
        self.use_lds_dma = flag("lds_dma", True, config)          # False: the register-staging fallback kernels
        # 16-bit modes: the whole fine level (gather + 2-layer transformer + fine matching) as ONE kernel (fine_fused.hip); False keeps
        # the unfused launch sequence (the only fine path of the fp32 parity mode)
        self.fine_fused = flag("fine_fused", True, config)
        # 16-bit modes, d_model 256: merge -> norm1 -> mlp -> norm2 -> residual of every coarse encoder layer as ONE kernel (token_mlp.hip)
        self.token_fused = flag("token_fused", True, config)
        # fused fine kernel launched with the device-side match count (no host round trip in front of it); False: sync first
        self.fine_dev_count = flag("fine_dev_count", True, config)
        self._count_pin = None
        # the token tail also computes the q / k / v projections its rows feed next (gim_token_mlp_emit): no projection GEMMs
        self.token_emit = flag("token_emit", True, config)
        # 16-bit modes, layer1 (planes 64): conv2 -> conv3 (+identity) -> the next block's conv1 chained through registers (bneck_fused.hip)
        self.bneck_fused = flag("bneck_fused", True, config)
        # 16-bit modes, layers 2 / 3: conv3 (+identity) -> the next block's conv1 in one kernel (bneck_tail.hip)
        self.bneck_tail = flag("bneck_tail", True, config)
        self.bneck_ds = flag("bneck_ds", True, config)   # layer 1's first block: downsample conv inside the fused kernel
        self.bneck_tail_ds = flag("bneck_tail_ds", True, config)   # layer 2's first block: downsample conv (stride 2) inside the tail kernel
        # launch-order experiments over independent images / pairs (same kernels, same arithmetic; see _backbone_trunk, _transformer_emit;
        # profiles/r05_launch_order.txt: two transformer chains -0.2 ms per batch-8 step, the other two do not pay)
        self.depth_groups = flag("depth_groups", 1, config)
        self.l3_chains = flag("l3_chains", 2, config)
        # precision='fp32': the GEMMs' fp32 operands as IEEE-fp16 hi / lo pairs on the 16-bit MFMA (gim_conv_args.split16: three products per 16 K, 2^-22 each)
        # instead of v_mfma_f32_32x32x2_f32 (exact fp32 products; the cross-check)
        self.fp32_split = flag("fp32_split", True, config)
        self.trunk_chains = flag("trunk_chains", False, config)
        self.tf_chains = flag("tf_chains", 2, config)
        # token tails write partial KV states instead of the k / v rows of the next attention (see _transformer_emit; False: rows + la_kv)
        self.kv_fused = flag("kv_fused", True, config)
        # every token tail projects its own queries (no q rows between the calls; False: the updating tail emits them)
        self.q_local = flag("q_local", True, config)
        # the first layer's [k | v] projection as partial KV states too (token kernel, projection only) instead of a GEMM + la_kv launches
        self.kv_init = flag("kv_init", True, config)
        self.pos_fused = flag("pos_fused", True, config)   # ... with the positional encoding added on the fly by that launch (False: gim_posenc_add in front)
        self._packed = None
        self._health = None          # fp16 range guard word of the forward in flight (count[1] of its coarse matching), see _coarse_stage
        self._health_sync_left = 3   # forwards that still wait for the fine kernel to read its health bit at once (fp16 mode)
        self._packed_key = None
        self._pe_cache = {}
        self.debug = None  # set to a dict to capture stage outputs (tests): coarse/fine maps, token features
        # HIP-graph replay of the shape-static part of the forward (config['graph'] = False disables).  A shape is captured
        # the second time it is seen (its first call runs eagerly and doubles as the warm-up), and at most
        # config['graph_cache'] (default 4) graphs -- each owns its static inputs and activation pool -- are kept, LRU.
        self.use_graph = flag("graph", True, config)
        self.graph_cache_size = max(1, flag("graph_cache", 4, config))
        self._graphs = collections.OrderedDict()
        self._seen = collections.OrderedDict()
        self._generation = 0
        if config.get("weight") is not None:
            self.load_state_dict(torch.load(config["weight"], map_location="cpu"))

    # ---- checkpoint surface (loftr.py:93-99) -----------------------------------------------------
    def load_state_dict(self, state_dict, *args, **kwargs):
        for k in list(state_dict.keys()):
            if k.startswith("model."):
                state_dict[k.replace("model.", "", 1)] = state_dict.pop(k)
            if k.startswith("matcher."):
                state_dict[k.replace("matcher.", "", 1)] = state_dict.pop(k)
        self._invalidate()
        return super().load_state_dict(state_dict, *args, **kwargs)

    def _invalidate(self):
        """weights / device / precision changed: packed weights, captured graphs AND the seen-shape counters go (a stale
        counter would send the next forward of a known shape straight into capture with nothing packed)"""
        self._packed = None
        self._health_sync_left = 3
        if hasattr(self, "_graphs"):
            self._graphs.clear()
            self._seen.clear()

    def _apply(self, fn, *a, **k):
        self._invalidate()
        return super()._apply(fn, *a, **k)

    def _check_sim(self, sim):
        """'fp32' (fp32 tokens on the fp32 MFMA) or the mode's own 16-bit type (operand copy of the tokens on the 16-bit MFMA)"""
        if sim not in ("fp32", self.precision):
            raise ValueError(f"coarse_sim must be 'fp32' or the precision mode's own type {self.precision!r}, got {sim!r}")
        return sim

    def _dt(self):
        return _DT[self.precision]

    def _img_dt(self):
        """dtype of the NHWC image tensor = operand type of the first convolution (see the module docstring)"""
        # (the fp16-in / bf16-out convolution exists on the LDS-DMA kernels only: with lds_dma off the stem reads the mode's own type)
        return GIM_F16 if (self.precision == "bf16" and self.stem_fp16 and self.use_lds_dma) else self._dt()

    def _split(self):
        """first convolution on split (hi + lo) operands?  16-bit modes only: to the implicit-GEMM kernel it is a 9-channel
        convolution, to gim_stem7x7 one MFMA per tap on [hi | lo] pixels"""
        if self.precision == "fp32":
            return False
        s = self.stem_split
        if isinstance(s, str):
            s = s.strip().lower()
            return self.precision == "fp16" if s == "auto" else s not in ("0", "false", "no", "off", "")
        return bool(s)

    def _stem_k(self):
        """first convolution through gim_stem7x7?  (16-bit modes; the kernel stages by LDS-DMA, so lds_dma = False turns it off too)"""
        return bool(self.stem_kernel) and self.precision != "fp32" and self.use_lds_dma

    def set_precision(self, precision, coarse_sim=None):
        assert precision in _DT
        self.precision = precision
        self.coarse_sim = self._check_sim(coarse_sim or precision)
        self._invalidate()
        return self

    # ---- weight pre-pack --------------------------------------------------------------------------
    @staticmethod
    def _bn(m):
        return (m.weight, m.bias, m.running_mean, m.running_var, m.eps)

    def _prepack(self, device):
        key = (str(device), self.precision, self._img_dt(), self._split(), self._stem_k())
        if self._packed is not None and self._packed_key == key:
            return self._packed
        dt = self._dt()
        tdt = torch_dtype(dt)
        P = {}
        enc = self.backbone.encode
        idt = self._img_dt()
        if self._stem_k():
            P["stem"] = pack_stem7x7(enc.conv1.weight, self._bn(enc.bn1), idt, device, split=self._split())
        elif self._split():
            P["stem"] = pack_conv_split(enc.conv1.weight, self._bn(enc.bn1), idt, device, stride=2, pad=3)
        else:
            P["stem"] = pack_conv(enc.conv1.weight, self._bn(enc.bn1), idt, device, stride=2, pad=3, cin_pad=cstore(3, idt))
        for li in (1, 2, 3):
            for bi, blk in enumerate(getattr(enc, f"layer{li}")):
                p = f"l{li}.{bi}."
                P[p + "c1"] = pack_conv(blk.conv1.weight, self._bn(blk.bn1), dt, device)
                P[p + "c2"] = pack_conv(blk.conv2.weight, self._bn(blk.bn2), dt, device, stride=blk.stride, pad=1)
                P[p + "c3"] = pack_conv(blk.conv3.weight, self._bn(blk.bn3), dt, device)
                if blk.downsample is not None:
                    P[p + "ds"] = pack_conv(blk.downsample[0].weight, self._bn(blk.downsample[1]), dt, device,
                                            stride=blk.stride)
        if is_half(dt):
            l1 = list(enc.layer1)
            for bi, blk in enumerate(l1):   # the last block's trailing conv1 is layer2's first one (256 -> 128, same resolution)
                P[f"l1.{bi}.fused"] = pack_bneck(blk, l1[bi + 1] if bi + 1 < len(l1) else enc.layer2[0], device, tdt)
            if l1[0].downsample is not None and l1[0].downsample[0].stride == (1, 1) and len(l1) > 1:
                P["l1.0.fused_ds"] = pack_bneck_ds(l1[0], l1[1], device, tdt)   # the first block's downsample conv runs inside the kernel
            # layers 2 / 3: conv3 + identity + relu of block bi with the 1x1 convolution that reads its output next (bneck_tail.hip):
            # the next block's conv1; for layer 2's last block the first conv1 of layer 3 (512 -> 256, same resolution: the stride sits
            # on conv2); for layer 3's last block the FPN's layer3_outconv (no BatchNorm, no activation)
            l2, l3 = list(enc.layer2), list(enc.layer3)

            def tail_fits(blk, nconv):   # the shapes gim_bneck_tail128 / 256 are built for; any other block_dims keeps conv3 + conv1 launches
                pl, c4, n1 = blk.conv3.weight.shape[1], blk.conv3.weight.shape[0], nconv.weight.shape[0]
                return pl in (128, 256) and c4 == 4 * pl and tuple(nconv.weight.shape[1:]) == (c4, 1, 1) and n1 in ((128, 256) if pl == 128 else (256,))

            for bi in range(len(l2)):
                nx = l2[bi + 1] if bi + 1 < len(l2) else (l3[0] if l3 else None)
                if nx is not None and tail_fits(l2[bi], nx.conv1):
                    P[f"l2.{bi}.tail"] = pack_bneck_tail(l2[bi], nx.conv1, nx.bn1, device, tdt)
                    d_ = l2[bi].downsample
                    if (bi == 0 and d_ is not None and d_[0].stride == (2, 2) and tuple(d_[0].weight.shape) == (512, 256, 1, 1)
                            and nx.conv1.weight.shape[0] == 128):   # layer 2's first block: the downsample branch as extra K of conv3
                        P[f"l2.{bi}.tail_ds"] = pack_bneck_tail(l2[bi], nx.conv1, nx.bn1, device, tdt, ds=True)
            for bi in range(len(l3)):
                nconv, nbn = (l3[bi + 1].conv1, l3[bi + 1].bn1) if bi + 1 < len(l3) else (self.backbone.layer3_outconv, None)
                if tail_fits(l3[bi], nconv):
                    P[f"l3.{bi}.tail"] = pack_bneck_tail(l3[bi], nconv, nbn, device, tdt)
        bb = self.backbone
        P["l3o"] = pack_conv(bb.layer3_outconv.weight, None, dt, device)
        P["l2o"] = pack_conv(bb.layer2_outconv.weight, None, dt, device)
        P["l2o2a"] = pack_conv(bb.layer2_outconv2[0].weight, self._bn(bb.layer2_outconv2[1]), dt, device, pad=1)
        P["l2o2b"] = pack_conv(bb.layer2_outconv2[3].weight, None, dt, device, pad=1)
        P["l1o"] = pack_conv(bb.layer1_outconv.weight, None, dt, device)
        P["l1o2a"] = pack_conv(bb.layer1_outconv2[0].weight, self._bn(bb.layer1_outconv2[1]), dt, device, pad=1)
        P["l1o2b"] = pack_conv(bb.layer1_outconv2[3].weight, None, dt, device, pad=1)
        for name, tf in (("c", self.loftr_coarse), ("f", self.loftr_fine)):
            for li, layer in enumerate(tf.layers):
                p = f"{name}{li}."
                P[p + "merge"] = pack_conv(layer.merge.weight, None, dt, device)
                P[p + "q_proj"] = pack_conv(layer.q_proj.weight, None, dt, device)
                # fused projections (one GEMM instead of three / two): self layers [q|k|v], cross layers [k|v]
                P[p + "qkv"] = pack_conv(torch.cat([layer.q_proj.weight, layer.k_proj.weight, layer.v_proj.weight], 0),
                                         None, dt, device)
                P[p + "kv"] = pack_conv(torch.cat([layer.k_proj.weight, layer.v_proj.weight], 0), None, dt, device)
                P[p + "mlp0"] = pack_conv(layer.mlp[0].weight, None, dt, device)
                P[p + "mlp2"] = pack_conv(layer.mlp[2].weight, None, dt, device)
                for nm in ("norm1", "norm2"):
                    ln = getattr(layer, nm)
                    P[p + nm] = (ln.weight.detach().float().to(device).contiguous(),
                                 ln.bias.detach().float().to(device).contiguous(), ln.eps)
        if is_half(dt) and self.loftr_coarse.d_model == 256:
            for li, layer in enumerate(self.loftr_coarse.layers):
                P[f"c{li}.tok"] = pack_token_mlp(layer, device, tdt)
                P[f"c{li}.qtok"] = pack_token_emit([layer.q_proj.weight], device, tdt)   # the call's own query projection (q_local)
                P[f"c{li}.kvtok"] = pack_token_emit([layer.k_proj.weight, layer.v_proj.weight], device, tdt)   # projection-only launch of the initial tokens (kv_init)
            for same_len in (True, False):
                P["c.emit", same_len] = self._emit_plan(self.loftr_coarse, same_len, device, tdt)
                P["c.emitk", same_len] = self._emit_plan(self.loftr_coarse, same_len, device, tdt, emit_q=False)
        fl = self.loftr_fine
        if is_half(dt) and fl.d_model == 128 and fl.nhead == 8 and fl.layer_names == ["self", "cross"] and self.W == 5:
            P["fine_fused"] = pack_fine_fused(fl.layers, device, tdt) + (fl.layers[0].norm1.eps,)
        if self.precision == "fp16":
            # range guard, weight side: a BatchNorm-folded weight beyond the IEEE-fp16 range became inf when it was packed
            bad = [k for k, v in P.items() if hasattr(v, "w") and torch.is_tensor(v.w) and not bool(torch.isfinite(v.w).all())]
            if bad:
                import warnings
                warnings.warn(f"gim_amd LoFTR: folded weights of {bad[:4]}{' ...' if len(bad) > 4 else ''} exceed the IEEE-fp16 range; "
                              "falling back to precision='bf16' for this module")
                self.fp16_overflowed = True
                self.set_precision("bf16", coarse_sim="fp32" if self.coarse_sim == "fp32" else None)   # a caller's fp32 similarity survives the fallback
                return self._prepack(device)
        self._packed, self._packed_key = P, key
        return P

    def _pos_encoding(self, d_model, h, w, device):
        """[h*w, C] fp32 table = pos_encoding buffer (position_encoding.py:22-36, temp_bug_fix=False as in
        loftr.py:22-24) sliced to (h,w) and laid out 'h w c'.  Constant per shape: built once on the host
        with the same fp32 ops as the reference, cached on the device."""
        key = (d_model, h, w, str(device))
        if key not in self._pe_cache:
            if h > 256 or w > 256:
                raise ValueError("coarse map exceeds PositionEncodingSine max_shape (256,256)")
            y_pos = torch.ones(h, w).cumsum(0).float().unsqueeze(0)
            x_pos = torch.ones(h, w).cumsum(1).float().unsqueeze(0)
            div = torch.exp(torch.arange(0, d_model // 2, 2).float() * (-math.log(10000.0) / d_model // 2))
            div = div[:, None, None]
            pe = torch.zeros(d_model, h, w)
            pe[0::4] = torch.sin(x_pos * div)
            pe[1::4] = torch.cos(x_pos * div)
            pe[2::4] = torch.sin(y_pos * div)
            pe[3::4] = torch.cos(y_pos * div)
            self._pe_cache[key] = pe.permute(1, 2, 0).reshape(h * w, d_model).contiguous().to(device)
        return self._pe_cache[key]

    # ---- stages -------------------------------------------------------------------------------------
    def _to_nhwc(self, images, dt, out=None):
        """list of [n_i,3,H,W] fp32 tensors sharing (H,W) -> one NHWC [sum n_i, H, W, cstore(3)] tensor of the compute dtype
        (replaces torch.cat([color0, color1]), loftr.py:60).  `out`: an existing tensor to fill (the HIP graph's static input)."""
        H, W = images[0].shape[2:]
        B = sum(im.shape[0] for im in images)
        split = self._split()
        if out is None:
            # gim_stem7x7 reads one 16-byte piece per pixel ([hi | lo | 0] when split); the implicit-GEMM form of the split needs 3 C
            cs = 8 if self._stem_k() else cstore(split_channels(images[0].shape[1]) if split else 3, dt)
            out = torch.empty(B, H, W, cs, dtype=torch_dtype(dt), device=images[0].device)
        off = 0
        for im in images:
            (ops.nchw_to_nhwc_split if split else ops.nchw_to_nhwc)(im, out, off)
            off += im.shape[0]
        return out

    def _backbone(self, P, x, dt):
        """x: NHWC [B,H,W,cstore(3)] images in the compute dtype.  Returns (x3_out NHWC [B,h8,w8,256], feat_f NHWC [B,h2,w2,128])
        in the compute dtype.  (resnet.py:230-235, 306-329)"""
        x1, x2, x3_out = self._backbone_trunk(P, x, dt)
        return x3_out, self._fpn_fine(P, x1, x2, x3_out)

    def _backbone_trunk(self, P, x, dt):
        """stem + layer1-3 + layer3_outconv (resnet.py:306-320): returns (x1, x2, x3_out) -- the coarse features x3_out are complete
        here; the fine head only needs x1, x2 and x3_out (see _fpn_fine).
        Images are independent up to here, so the batch may run as image groups: `depth_groups` > 1 walks stem -> layer1 -> layer2
        group by group (producer -> consumer tensors of a group stay closer to the Infinity Cache), `l3_chains` > 1 runs layer 3 as
        that many image groups on parallel streams (its 300- / 600-tile launches fill 0.6 / 1.2 rounds of the workgroup slots: two
        unsynchronised chains keep the slots busy -- round 5 measured +-0, round 6 -0.25 ms per batch-8 step on one box, 10.16 -> 9.91 ms,
        and made 2 the default; 4 and 8 chains measure 9.95 / 10.03), `trunk_chains` starts the chains at the stem.  All of them are
        launch-order changes only: same kernels, same arithmetic."""
        B = x.shape[0]
        K = self.l3_chains if (self.debug is None and self.l3_chains > 1 and B % self.l3_chains == 0) else 1
        if K > 1:   # a chain must keep the shapes the fused Bottleneck tails take (256-row tiles at 1/4 and 1/8 resolution), else it would fall back to unfused launches
            half_ = lambda v: (v - 1) // 2 + 1   # noqa: E731
            h4, w4 = half_(half_(x.shape[1])), half_(half_(x.shape[2]))
            if (B // K * h4 * w4) % 256 != 0 or (B // K * half_(h4) * half_(w4)) % 256 != 0:
                K = 1
        whole = K > 1 and self.trunk_chains   # the chains start at the stem instead of at layer 3
        G = K if whole else (self.depth_groups if (self.debug is None and self.depth_groups > 1 and B % self.depth_groups == 0) else 1)
        o3 = None
        if G > 1:
            n = B // G
            half = lambda v: (v - 1) // 2 + 1   # noqa: E731
            H1, W1 = half(x.shape[1]), half(x.shape[2])
            H2, W2 = half(H1), half(W1)
            tdt = torch_dtype(dt)
            x1 = torch.empty(B, H1, W1, P["l1.2.c3"].n_store, dtype=tdt, device=x.device)
            x2 = torch.empty(B, H2, W2, P["l2.3.c3"].n_store, dtype=tdt, device=x.device)
            if self.bneck_tail and "l2.3.tail" in P and (n * H2 * W2) % 256 == 0 and n * H2 * W2 * 1024 < (1 << 32) - 16:   # (_layer's test)
                o3 = torch.empty(B, H2, W2, P["l3.0.c1"].n_store, dtype=tdt, device=x.device)

        def trunk12_group(g):
            sl = slice(g * n, (g + 1) * n)
            a1, a2, ao = self._trunk12(P, x[sl], dt, out=(x1[sl], x2[sl], o3[sl] if o3 is not None else None))
            assert (ao is None) == (o3 is None)
            for dst, src in ((x1, a1), (x2, a2), (o3, ao)):   # a launch that was not a fused one allocated its own output
                if src is not None and src.data_ptr() != dst[sl].data_ptr():
                    ops.copy_segments([(src.contiguous(), dst[sl])])

        if G == 1:
            x1, x2, o3 = self._trunk12(P, x, dt)
        elif not whole:
            for g in range(G):
                trunk12_group(g)
        if K == 1:
            x3, _, x3_out = self._layer(P, 3, 6, x2, o3)
            if x3_out is None:
                x3_out = ops.conv2d(x3, P["l3o"], lds_dma=self.use_lds_dma)
            return x1, x2, x3_out
        n = B // K
        main = torch.cuda.current_stream()
        sides = self._side_streams(x.device, K - 1)
        x3_out = torch.empty(B, (x2.shape[1] - 1) // 2 + 1, (x2.shape[2] - 1) // 2 + 1, P["l3o"].n_store, dtype=x2.dtype, device=x2.device)
        keep = []   # tensors that cross streams stay referenced until the join (the caching allocator re-uses a freed block per stream)

        def chain(g):
            if whole:
                trunk12_group(g)
            sl = slice(g * n, (g + 1) * n)
            x3, _, xo = self._layer(P, 3, 6, x2[sl], o3[sl] if o3 is not None else None, out_last=(None, x3_out[sl]))
            if xo is None:
                xo = ops.conv2d(x3, P["l3o"], lds_dma=self.use_lds_dma)
            if xo.data_ptr() != x3_out[sl].data_ptr():
                ops.copy_segments([(xo.contiguous(), x3_out[sl])])
            keep.append((x3, xo))

        for g in range(1, K):
            sides[g - 1].wait_stream(main)
            with torch.cuda.stream(sides[g - 1]):
                chain(g)
        chain(0)
        for s_ in sides:
            main.wait_stream(s_)
        return x1, x2, x3_out

    def _side_streams(self, dev, n):
        ss = getattr(self, "_sides", None)
        if ss is None or len(ss) < n or ss[0].device != dev:
            ss = self._sides = [torch.cuda.Stream(device=dev) for _ in range(n)]
        return ss[:n]

    def _trunk12(self, P, x, dt, out=None):
        """stem + layer 1 + layer 2 of a batch (or image group) -> (x1, x2, conv1 output of layer 3's first block or None);
        out = (x1, x2, o3) destinations for the fused kernels' outputs (image-group mode)"""
        dma = self.use_lds_dma
        if isinstance(P["stem"], PackedStem):
            x = ops.stem7x7(x, P["stem"], out_dtype=torch_dtype(dt))
        else:
            x = ops.conv2d(x, P["stem"], ACT_RELU, out_dtype=torch_dtype(dt), lds_dma=dma)   # image dtype may be fp16 in bf16 mode
        x1, o, _ = self._layer(P, 1, 3, x, None, out_last=(out[0], None) if out else None)
        x2, o, _ = self._layer(P, 2, 4, x1, o, out_last=(out[1], out[2]) if out else None)
        return x1, x2, o

    def _layer(self, P, li, nblk, x, o, out_last=None):
        """one ResNet layer (resnet.py:109-126, 230-235) on x; o = conv1 output of its first block when the previous layer's last fused
        kernel already produced it.  Returns (x', conv1 output of the NEXT layer's first block or None, x3_out or None: the FPN's
        layer3_outconv when layer 3's last fused tail produced it).  out_last = (x', t1') destinations of the last block's fused
        kernel (either may be None)."""
        dma = self.use_lds_dma
        x3_out = None
        fuse = li == 1 and self.bneck_fused and "l1.0.fused" in P and x.shape[1] % 8 == 0 and x.shape[2] % 32 == 0
        for bi in range(nblk):
            p = f"l{li}.{bi}."
            last = bi == nblk - 1
            outs = out_last if last else None
            if o is None:
                o = ops.conv2d(x, P[p + "c1"], ACT_RELU, lds_dma=dma)
            if fuse and self.bneck_ds and (p + "fused_ds") in P and x.shape[3] == 64 and x.is_contiguous():
                x, o = ops.bneck64_ds(o, x, P[p + "fused_ds"], out=outs, health=self._health)   # ... and the downsample branch: no identity tensor at all
                continue
            rows_ds = x.shape[0] * ((x.shape[1] - 1) // 2 + 1) * ((x.shape[2] - 1) // 2 + 1)   # output rows of a stride-2 block
            if (self.bneck_tail and self.bneck_tail_ds and (p + "tail_ds") in P and x.is_contiguous() and x.shape[3] == 256
                    and rows_ds % 256 == 0 and rows_ds * 512 * 2 < (1 << 32) - 16):   # 32-bit byte offsets into [rows, 512] (its own REQUIRE): oversize batches take the unfused launches
                # layer 2's first block: conv2, then ONE kernel for conv3 + the stride-2 downsample branch (extra K) + relu + the next conv1:
                # no downsample launch, no identity tensor (gim_bneck_tail128_ds)
                o = ops.conv2d(o, P[p + "c2"], ACT_RELU, lds_dma=dma)
                x, o = ops.bneck_tail_ds(o, x, P[p + "tail_ds"], out=outs, health=self._health)
                continue
            idn = ops.conv2d(x, P[p + "ds"], ACT_NONE, lds_dma=dma) if (p + "ds") in P else x
            if fuse:   # conv2 -> conv3 + identity -> the next conv1 (of this layer, or layer2's first), one kernel
                x, o = ops.bneck64(o, idn, P[p + "fused"], True, out=outs, health=self._health)
                continue
            o = ops.conv2d(o, P[p + "c2"], ACT_RELU, lds_dma=dma)
            rows = o.shape[0] * o.shape[1] * o.shape[2]
            # the tail kernel walks 256-row tiles with 32-bit byte offsets into the [rows, 4 P] tensors (its own REQUIREs)
            if self.bneck_tail and (p + "tail") in P and rows % 256 == 0 and rows * 4 * o.shape[3] * 2 < (1 << 32) - 16:
                if li == 3 and last:   # last block: t1' IS x3_out (layer3_outconv), x3 itself is read by nothing else
                    x, x3_out = ops.bneck_tail(o, idn.contiguous(), P[p + "tail"], ACT_NONE, store_x=self.debug is not None, out=outs,
                                               health=self._health)
                    o = None
                else:
                    x, o = ops.bneck_tail(o, idn.contiguous(), P[p + "tail"], out=outs, health=self._health)   # x' and the next block's conv1 output
                continue
            x = ops.conv2d(o, P[p + "c3"], ACT_RELU, res=idn, lds_dma=dma, health=self._health)   # the unfused block stores the stream too
            o = None
        return x, o, x3_out

    def _fpn_fine(self, P, x1, x2, x3_out):
        """the FPN's top-down path to the 1/2-resolution fine features (resnet.py:321-329): six convolutions that nothing of the coarse
        level (position encoding, transformer, coarse matching) depends on"""
        dma = self.use_lds_dma
        # lateral 1x1 conv + F.interpolate(scale_factor=2, bilinear, align_corners=True) of the coarser level + add (resnet.py:
        # 321-327): the upsample-add runs in the conv's epilogue when the launch takes it, else as a second pass over the output
        x2_out = ops.conv2d(x2, P["l2o"], lds_dma=dma, ups=x3_out)
        x2_out = ops.conv2d(ops.conv2d(x2_out, P["l2o2a"], ACT_LEAKY, lds_dma=dma), P["l2o2b"], lds_dma=dma)
        x1_out = ops.conv2d(x1, P["l1o"], lds_dma=dma, ups=x2_out)
        x1_out = ops.conv2d(ops.conv2d(x1_out, P["l1o2a"], ACT_LEAKY, lds_dma=dma), P["l1o2b"], lds_dma=dma)
        return x1_out

    class _TfBuffers:
        """Row buffers of one LocalFeatureTransformer run over R rows of width C.  The q / k / v rows, the message, the pre-LayerNorm
        activations and the hidden layer exist only on the UNFUSED launch paths (masks, debug dumps, fp32 mode without the token kernel):
        they are allocated on first touch -- the default 16-bit coarse path (local queries, fused KV state, first layer's state from the
        projection-only launch) never touches them (two [R, 3C] buffers = 236 MB at batch 8 were allocated, and pinned by the captured
        graph's pool, for nothing: ADVICE r5)."""
        _LAZY = {"QKV": 3, "QKV2": 3, "MSG": 1, "MRG": 1, "HID": 2, "MLP": 1}   # name -> width in units of C

        def __init__(self, R, C, tdt, dev):
            f32 = torch.float32
            self._shape = (R, C, tdt, dev)
            self.X32 = torch.empty(R, C, dtype=f32, device=dev)       # fp32 master of the token features
            self.CAT = torch.empty(R, 2 * C, dtype=tdt, device=dev)   # [x | norm1(message)] GEMM operand
            # QKV / QKV2: [elu(q)+1 | elu(k)+1 | v] row buffers (two alternate by layer parity on the emitting path); MSG; MRG, HID, MLP:
            # pre-LayerNorm activations in the operand dtype (A/B on one box: 14.16-14.26 vs 14.38-14.48 ms with fp32) -- see _LAZY
            self.ws = None
            self.MASK = None  # optional uint8 [R] padding mask aligned with the rows (coarse level only)
            self.pos = None   # [(feature rows, pos-encoding table)] per side when the tokens still lack their positional encoding (coarse level)
            self.pos_all = None

        def __getattr__(self, name):   # only reached when the attribute does not exist yet
            w = type(self)._LAZY.get(name)
            if w is None:
                raise AttributeError(name)
            R, C, tdt, dev = self._shape
            t = torch.empty(R, w * C, dtype=tdt, device=dev)
            setattr(self, name, t)
            return t

    def _encoder_layer(self, P, p, T, xs, ss, nb, L, S, H, have_q=False, with_q_of_source=False):
        """LoFTREncoderLayer.forward (transformer.py:35-58) on row ranges xs (queries) / ss (source).
        Cross layers run as a pair (feat0 against feat1, then feat1 against the updated feat0, transformer.py:95-96): the FIRST call
        also projects the source's queries for the second one (`with_q_of_source`: one [q|k|v] GEMM on the source rows instead of a
        [k|v] GEMM now and a q GEMM later -- those rows do not change in between), which then runs with `have_q`."""
        C = T.X32.shape[1]
        dma = self.use_lds_dma
        x_t, s_t = T.CAT[xs, :C], T.CAT[ss, :C]
        # q/k/v projections with elu(.)+1 (attentions.py:31-32) fused into the epilogue of the q and k columns
        if xs == ss:
            ops.linear(x_t, P[p + "qkv"], T.QKV[xs], ACT_ELU1, dma, act_cols=2 * C)
        else:
            if not have_q:
                ops.linear(x_t, P[p + "q_proj"], T.QKV[xs, :C], ACT_ELU1, dma)
            if with_q_of_source:
                ops.linear(s_t, P[p + "qkv"], T.QKV[ss], ACT_ELU1, dma, act_cols=2 * C)
            else:
                ops.linear(s_t, P[p + "kv"], T.QKV[ss, C:], ACT_ELU1, dma, act_cols=C)
        qm = T.MASK[xs] if T.MASK is not None else None  # x_mask / source_mask (transformer.py:50, attentions.py:35-39)
        km = T.MASK[ss] if T.MASK is not None else None
        fused = self.token_fused and (p + "tok") in P
        if fused and L % 64 == 0 and H == 8 and C == 256:
            # KV / Ksum state of the source, then ONE kernel: attention apply + merge + norm1 + mlp + norm2 + residual
            wts, lnp, eps = P[p + "tok"]
            T.ws, _ = ops.linear_attention_state(T.QKV[ss, C:2 * C], T.QKV[ss, 2 * C:], nb, S, H, T.ws, km)
            ops.token_mlp(T.QKV[xs, :C], T.CAT[xs, :C], T.X32[xs], wts, lnp, eps, kv=T.ws, L=L, S=S, q_mask=qm)
            return
        T.ws = ops.linear_attention(T.QKV[xs, :C], T.QKV[ss, C:2 * C], T.QKV[ss, 2 * C:], T.MSG[xs], nb, L, nb, S, H,
                                    T.ws, qm, km)
        if fused:
            wts, lnp, eps = P[p + "tok"]
            ops.token_mlp(T.MSG[xs], T.CAT[xs, :C], T.X32[xs], wts, lnp, eps)   # x += norm2(mlp(cat[x, norm1(merge(msg))]))
            return
        ops.linear(T.MSG[xs], P[p + "merge"], T.MRG[xs], ACT_NONE, dma)
        g1, b1, e1 = P[p + "norm1"]
        ops.layernorm_residual(T.MRG[xs], g1, b1, None, None, T.CAT[xs, C:], e1)
        ops.linear(T.CAT[xs], P[p + "mlp0"], T.HID[xs], ACT_RELU, dma)
        ops.linear(T.HID[xs], P[p + "mlp2"], T.MLP[xs], ACT_NONE, dma)
        g2, b2, e2 = P[p + "norm2"]
        ops.layernorm_residual(T.MLP[xs], g2, b2, T.X32[xs], T.X32[xs], T.CAT[xs, :C], e2)  # x + message

    @staticmethod
    def _emit_plan(tf, same_len, device, tdt, emit_q=True):
        """Launch plan of a LocalFeatureTransformer whose token tails emit the projections (gim_token_mlp_emit).
        The call sequence (transformer.py:88-99) as (layer, query sides, source sides); after each call, the q / k / v projections
        its rows feed before they are updated again: in a cross layer the first call's rows are the source of the second call (k, v
        of the SAME layer), then the queries / source of the next layer.  Returns (calls, per call (weight stream, [(layer, column
        block 0..2, sides)]) or None, the projections of the initial tokens)."""
        calls = []
        for li, kind in enumerate(tf.layer_names):
            if kind == "self":
                calls += [(li, (0, 1), (0, 1))] if same_len else [(li, (0,), (0,)), (li, (1,), (1,))]
            else:
                calls += [(li, (0,), (1,)), (li, (1,), (0,))]

        def needs_after(ci, sides):
            need = {}   # (layer, column block) -> sides, in first-use order
            for sd in sides:
                for li, xs, ss in calls[ci + 1:]:
                    if sd in ss:
                        need.setdefault((li, 1), []).append(sd)
                        need.setdefault((li, 2), []).append(sd)
                    if sd in xs:   # these rows are updated by that call: nothing later reads the current values
                        if emit_q:   # (emit_q = False: every call projects its own queries, token_mlp.hip "local queries")
                            need.setdefault((li, 0), []).append(sd)
                        break
            return [(li, blk, tuple(sorted(sd))) for (li, blk), sd in need.items()]

        per_call = []
        for ci, (li, xs, _) in enumerate(calls):
            blocks = needs_after(ci, xs)
            assert len(blocks) <= 6
            if not blocks:
                per_call.append(None)
                continue
            wsel = [(tf.layers[l2].q_proj, tf.layers[l2].k_proj, tf.layers[l2].v_proj)[blk].weight for l2, blk, _ in blocks]
            per_call.append((pack_token_emit(wsel, device, tdt), blocks))
        return calls, per_call, needs_after(-1, (0, 1))

    @staticmethod
    def _kv_consumers(calls, per_call, initial):
        """(call index, index of a k block in that call's block list) -> index of the later call whose SOURCE those k / v rows are, for every
        (k, v) pair a token tail may hand over as partial KV states (the v block is the next entry of the list).  A call fed, for any of its
        source sides, by the initial projections keeps the row path: its state comes from one gim_linear_attention_kv launch."""
        fed_by_initial = {(li, sd) for li, blk, sides in initial if blk == 1 for sd in sides}
        out = {}
        for ci, em in enumerate(per_call):
            if em is None:
                continue
            blocks = em[1]
            for bi, (l2, blk, sides) in enumerate(blocks):
                if blk != 1:
                    continue
                assert bi + 1 < len(blocks) and blocks[bi + 1] == (l2, 2, sides), "the v block follows its k block"
                for cj in range(ci + 1, len(calls)):
                    if calls[cj][0] == l2 and set(sides) <= set(calls[cj][2]):
                        if not any((l2, sd) in fed_by_initial for sd in calls[cj][2]):
                            out[(ci, bi)] = cj
                        break
        return out

    def _transformer_emit(self, P, name, tf, T, n0, L, n1, S):
        """LocalFeatureTransformer.forward with every projection after the first layer's computed by the token tail that produced
        its operand rows (see _emit_plan).  Two [R, 3C] projection buffers alternate by layer parity: a tail writes the NEXT layer's
        columns while its own layer's are still being read.
        `tf_chains` > 1 (coarse level, n0 == n1 pairs): pairs are independent sequences, so the pair batch runs as that many chains of
        n0 / chains pairs on parallel streams, each with the per-side launch plan -- a cross-layer call of the whole batch is 600
        64-token tiles on 512 workgroup slots (two rounds, the second 17 % full); unsynchronised chains keep the slots busy."""
        C = T.X32.shape[1]
        H = tf.nhead
        K = self.tf_chains if (name == "c" and self.tf_chains > 1 and n0 == n1 and n0 % self.tf_chains == 0 and self.debug is None
                               and (name + ".emit", False) in P) else 1
        # local queries: every call projects its own q rows from the operand copy of x it loads anyway -- no q blocks, no q rows
        ql = bool(self.q_local) and self.debug is None and (name + ".emitk", False) in P
        calls, per_call, initial = P[name + (".emitk" if ql else ".emit"), L == S and K == 1]
        class _QK:   # the two projection row buffers, by layer parity; allocated when a block is really emitted as rows (_TfBuffers._LAZY)
            def __getitem__(self, i):
                return T.QKV2 if i else T.QKV
        QK = _QK()
        rows = (slice(0, n0 * L), slice(n0 * L, n0 * L + n1 * S))
        rall = slice(0, n0 * L + n1 * S)
        rs = lambda sides: rall if len(sides) == 2 else rows[sides[0]]   # noqa: E731
        # projections of the initial tokens: GEMMs (one per layer and side set; q/k/v column blocks that go together in one launch)
        groups = {}
        for li, blk, sides in initial:
            groups.setdefault((li, sides), []).append(blk)
        fuse_kv = bool(self.kv_fused) and T.MASK is None and self.debug is None
        # ... except a [k | v] group when the states are handed over anyway (`kv_init`): the token kernel in its projection-only form writes the
        # partial states of those rows -- no projection GEMM, no k / v rows, no la_kv launches for the first layer either.  One workspace per
        # side (both sides in one when the sequences are equally long: a self layer over both images reads one contiguous state).
        init_state = {}   # (layer, side) -> (workspace, index of the side's first sequence in it)
        if fuse_kv and self.kv_init and ql:
            for (li, sides), blks in list(groups.items()):
                if sorted(blks) != [1, 2]:
                    continue
                del groups[(li, sides)]
                ew = P[f"{name}{li}.kvtok"]
                parts = [sides] if (len(sides) == 1 or L == S) else [(sd,) for sd in sides]
                for part in parts:
                    r = rs(part)
                    nseq, slen = sum((n0, n1)[sd] for sd in part), (L if part[0] == 0 else S)
                    ws_i = ops.kv_state_workspace(nseq, slen // 64, T.X32.device)
                    nrow = r.stop - r.start
                    # the positional encoding of these rows on the fly (`pos_fused`) when they still lack it and one table serves the launch
                    pos = None
                    if self.pos_fused and T.pos is not None and all(T.pos[sd] is not None for sd in part):
                        if len(part) == 1:
                            pos = (T.pos[part[0]][0], T.pos[part[0]][1], T.X32[r])
                        elif T.pos_all is not None and T.pos[0][1] is T.pos[1][1]:
                            pos = (T.pos_all, T.pos[0][1], T.X32[r])
                        if pos is not None:
                            for sd in part:
                                T.pos[sd] = None
                    if pos is None:
                        self._posenc(T, rows, part)
                    ops.token_project(T.CAT[r, :C], (ew, [(None, ACT_ELU1, 0, nrow, (ws_i, nseq, slen // 64, 0, slen)), (None, ACT_NONE, 0, nrow)]), pos=pos)
                    ops.kv_state_finalize(ws_i, nseq, slen // 64)
                    off = 0
                    for sd in part:
                        init_state[(li, sd)] = (ws_i, off)
                        off += (n0, n1)[sd]
        # Cross-stream lifetime: the chains below run on side streams and read these workspaces (and write the ones in T.keep).  No
        # record_stream() is needed because the owning tensors stay referenced by T (init_state, keep) until _coarse_stage returns, i.e.
        # past the main.wait_stream() join at the end of this function -- the caching allocator cannot hand their blocks out before that
        T.init_state = init_state
        self._posenc(T, rows)   # whatever the projection-only launches did not cover
        for (li, sides), blks in groups.items():
            p, r = f"{name}{li}.", rs(sides)
            x_t, q = T.CAT[r, :C], QK[li & 1]
            blks = sorted(blks)
            if blks == [0, 1, 2]:
                ops.linear(x_t, P[p + "qkv"], q[r], ACT_ELU1, self.use_lds_dma, act_cols=2 * C)
            elif blks == [1, 2]:
                ops.linear(x_t, P[p + "kv"], q[r, C:], ACT_ELU1, self.use_lds_dma, act_cols=C)
            else:
                assert blks == [0], blks
                ops.linear(x_t, P[p + "q_proj"], q[r, :C], ACT_ELU1, self.use_lds_dma)

        # Fused KV state (token_mlp.hip): the k / v rows of a side have one reader, the state reduction of the call they are the source of.
        # A token tail that would emit them writes its tiles' partial states into that call's workspace instead; the call then only sums them
        # (gim_linear_attention_finalize) -- no k / v rows, no la_kv launch.  Not for calls fed by the initial projections, padded inputs
        # (the state reduction masks rows) or the debug dumps.
        consumers = self._kv_consumers(calls, per_call, initial) if fuse_kv else {}

        def run(rows_c, m0, m1, ws):
            """the call sequence on the row ranges (side 0, side 1) of m0 / m1 sequences; returns the KV workspaces it used (the caller keeps
            them referenced until the streams join; [0] is re-usable by the next forward)"""
            both = slice(rows_c[0].start, rows_c[1].stop)
            rc = lambda sides: both if len(sides) == 2 else rows_c[sides[0]]   # noqa: E731
            nseq = lambda sides: m0 + m1 if len(sides) == 2 else (m0, m1)[sides[0]]   # noqa: E731
            kvws = {}   # consumer call index -> workspace its producers fill
            used = []
            for ci, ((li, xs_s, ss_s), em) in enumerate(zip(calls, per_call)):
                xs, ss = rc(xs_s), rc(ss_s)
                nb_src = nseq(ss_s)
                len_q = L if xs_s[0] == 0 else S
                len_src = L if ss_s[0] == 0 else S
                qm = T.MASK[xs] if T.MASK is not None else None
                km = T.MASK[ss] if T.MASK is not None else None
                wts, lnp, eps = P[f"{name}{li}.tok"]
                if ci in kvws:
                    kv = ops.kv_state_finalize(kvws.pop(ci), nb_src, len_src // 64)
                elif all((li, sd) in init_state for sd in ss_s) and len({id(init_state[(li, sd)][0]) for sd in ss_s}) == 1:
                    # the first layer's state, written before the chains forked: this call's sequences are a slice of it
                    ws_i, off = init_state[(li, ss_s[0])]
                    seq0 = off + (ss.start - rows[ss_s[0]].start) // len_src
                    kv = ws_i[seq0 * H * (32 * 32 + 32):]
                else:
                    q = QK[li & 1]
                    ws, _ = ops.linear_attention_state(q[ss, C:2 * C], q[ss, 2 * C:], nb_src, len_src, H, ws, km)
                    kv = ws
                emit = None
                if em is not None:
                    ew, blocks = em
                    spec = []
                    for bi, (l2, blk, sides) in enumerate(blocks):
                        lo, hi = (0, xs.stop - xs.start) if sides == xs_s else ((0, m0 * L) if sides == (0,) else (m0 * L, xs.stop - xs.start))
                        cj = consumers.get((ci, bi if blk == 1 else bi - 1)) if blk >= 1 else None
                        if cj is None:
                            spec.append((QK[l2 & 1][xs, blk * C:(blk + 1) * C], ACT_ELU1 if blk < 2 else ACT_NONE, lo, hi))
                        elif blk == 1:   # K of a fused pair; its V block follows (same consumer, same rows)
                            src = rc(calls[cj][2])
                            slen = L if calls[cj][2][0] == 0 else S
                            if cj not in kvws:
                                kvws[cj] = ops.kv_state_workspace(nseq(calls[cj][2]), slen // 64, T.X32.device)
                                used.append(kvws[cj])
                            spec.append((None, ACT_ELU1, lo, hi, (kvws[cj], nseq(calls[cj][2]), slen // 64, (xs.start + lo - src.start) // 64, slen)))
                        else:
                            assert len(spec[-1]) == 5 and spec[-1][2:4] == (lo, hi)
                            spec.append((None, ACT_NONE, lo, hi))
                    emit = (ew, spec)
                ops.token_mlp(None if ql else QK[li & 1][xs, :C], T.CAT[xs, :C], T.X32[xs], wts, lnp, eps, kv=kv, L=len_q, S=len_src, q_mask=qm, emit=emit,
                              q_weights=P[f"{name}{li}.qtok"] if ql else None)
            assert not kvws
            return [ws] + used

        if K == 1:
            T.keep = run(rows, n0, n1, T.ws)
            T.ws = T.keep[0]
            return
        m = n0 // K
        main = torch.cuda.current_stream()
        sides_ = self._side_streams(T.X32.device, K - 1)
        keep = []   # per-chain workspaces stay referenced until the join
        for g in range(1, K):
            sides_[g - 1].wait_stream(main)
            with torch.cuda.stream(sides_[g - 1]):
                keep.append(run((slice(g * m * L, (g + 1) * m * L), slice(n0 * L + g * m * S, n0 * L + (g + 1) * m * S)), m, m, None))
        keep.append(run((slice(0, m * L), slice(n0 * L, n0 * L + m * S)), m, m, None))
        for s_ in sides_:
            main.wait_stream(s_)
        T.keep = keep   # (every side-stream workspace outlives the join above: see the lifetime note at T.init_state)

    @staticmethod
    def _posenc(T, rows, sides=(0, 1)):
        """pos_encoding + 'n c h w -> n (h w) c' (loftr.py:74-75) of the sides that still lack it: feature rows + table -> X32 and the operand copy"""
        if T.pos is None:
            return
        C = T.X32.shape[1]
        for sd in sides:
            if T.pos[sd] is not None:
                feat, pe = T.pos[sd]
                ops.posenc_add(feat, pe, T.X32[rows[sd]], T.CAT[rows[sd], :C])
                T.pos[sd] = None

    def _transformer(self, P, name, tf, T, n0, L, n1, S):
        """LocalFeatureTransformer.forward (transformer.py:80-101).  Rows [0, n0*L) are feat0's tokens,
        rows [n0*L, n0*L + n1*S) feat1's; n0 == n1 sequences on each side."""
        r0, r1 = slice(0, n0 * L), slice(n0 * L, n0 * L + n1 * S)
        rall = slice(0, n0 * L + n1 * S)
        H = tf.nhead
        if (self.token_emit and self.token_fused and (name + ".emit", L == S) in P and L % 64 == 0 and S % 64 == 0 and H == 8
                and T.X32.shape[1] == 256 and all(f"{name}{li}.tok" in P for li in range(len(tf.layer_names)))):
            return self._transformer_emit(P, name, tf, T, n0, L, n1, S)
        self._posenc(T, (r0, r1))
        for li, kind in enumerate(tf.layer_names):
            p = f"{name}{li}."
            if kind == "self":
                if L == S:  # one launch set for both sides: same weights, independent sequences
                    self._encoder_layer(P, p, T, rall, rall, n0 + n1, L, S, H)
                else:
                    self._encoder_layer(P, p, T, r0, r0, n0, L, L, H)
                    self._encoder_layer(P, p, T, r1, r1, n1, S, S, H)
            else:  # cross: feat0 first, then feat1 against the *updated* feat0 (transformer.py:95-96)
                self._encoder_layer(P, p, T, r0, r1, n0, L, S, H, with_q_of_source=True)
                self._encoder_layer(P, p, T, r1, r0, n1, S, L, H, have_q=True)

    # ---- forward (loftr.py:43-91) -------------------------------------------------------------------
    def _coarse_stage(self, xs, bs, scale0, scale1, mask0=None, mask1=None, count=None):
        """Everything from the NHWC images up to and including coarse matching: a fixed launch sequence with no host sync and
        no data-dependent shape, so it can be captured once per input shape into a HIP graph and replayed.
        xs: [x_all] (both images of all pairs in one [2 bs, H, W, c] tensor: equal image shapes) or [x0, x1] (loftr.py:59-63).
        Returns a dict of device tensors (graph-owned when captured)."""
        dev = xs[0].device
        dt = self._dt()
        # [match count, health word, per-pair counts] of this forward's coarse matching.  The health word doubles as the fp16 range
        # guard of the kernels in front of it (registered here, read back with the count): allocated before the first launch
        if count is None:
            count = torch.zeros(2 + bs, dtype=torch.int32, device=dev)
        # the fp16 range guard of the kernels in front of it is that word too, handed to every launch that stores a residual stream
        self._health = count[1:2] if self.precision == "fp16" else None
        tdt = torch_dtype(dt)
        P = self._prepack(dev)
        cfg = self.config
        if len(xs) == 1:
            # (round 4: the fine head on a second stream beside the coarse level -- a graph with two branches, the head's persistent
            # workgroups filling the CUs the transformer's 600-tile launches leave idle -- measured SLOWER, 10.96 vs 10.70 ms per step:
            # the two branches fight over L2 and LDS instead of complementing each other; one stream it is)
            c_all, f_all = self._backbone(P, xs[0], dt)
            c0, c1 = c_all[:bs], c_all[bs:]
            f0, f1 = f_all[:bs], f_all[bs:]
        else:  # different input shapes (loftr.py:62-63)
            c0, f0 = self._backbone(P, xs[0], dt)
            c1, f1 = self._backbone(P, xs[1], dt)
        hw0_c, hw1_c = c0.shape[1:3], c1.shape[1:3]
        # 2. coarse transformer on pos-encoded tokens (NHWC rows == 'n (h w) c', loftr.py:74-75)
        C = cfg["coarse"]["d_model"]
        L, S = hw0_c[0] * hw0_c[1], hw1_c[0] * hw1_c[1]
        T = self._TfBuffers(bs * (L + S), C, tdt, dev)
        r0, r1 = slice(0, bs * L), slice(bs * L, bs * (L + S))
        if mask0 is not None:  # mask_c0 = mask0.flatten(-2) (loftr.py:78-79); row-aligned with the token buffers
            T.MASK = torch.empty(bs * (L + S), dtype=torch.uint8, device=dev)
            T.MASK[r0].copy_(mask0.reshape(-1))
            T.MASK[r1].copy_(mask1.reshape(-1))
        # (the positional encoding itself runs inside _transformer: in front of the first layer's projections -- the projection-only token
        #  kernel adds it on the fly where it can, gim_posenc_add does it otherwise)
        T.pos = [(c0.reshape(-1, C), self._pos_encoding(C, *hw0_c, dev)), (c1.reshape(-1, C), self._pos_encoding(C, *hw1_c, dev))]
        T.pos_all = c_all.reshape(-1, C) if len(xs) == 1 else None    # both sides' feature rows as one tensor (same image shapes)
        self._transformer(P, "c", self.loftr_coarse, T, bs, L, bs, S)
        # 3. coarse matching (coarse_matching.py:88-259), fused
        mc = cfg["match_coarse"]
        scale = xs[0].shape[1] / hw0_c[0]
        if is_half(dt) and self.coarse_sim == self.precision:
            # opt-in: the operand-dtype copy of the final tokens (written by the last LayerNorm for the next GEMM)
            # feeds the similarity -- bf16 MFMA with fp32 accumulation, not index-exact against the fp32 tokens
            fc0 = T.CAT[r0].view(bs, L, 2 * C)[:, :, :C]
            fc1 = T.CAT[r1].view(bs, S, 2 * C)[:, :, :C]
        else:
            fc0, fc1 = T.X32[r0].view(bs, L, C), T.X32[r1].view(bs, S, C)
        cr = ops.coarse_match(fc0, fc1, hw0_c, hw1_c, scale,
                              mc["dsmax_temperature"], mc["thr"], mc["border_rm"], scale0, scale1,
                              T.MASK[r0] if mask0 is not None else None, T.MASK[r1] if mask0 is not None else None, count=count)
        self._health = None
        return {"c0": c0, "c1": c1, "f0": f0, "f1": f1, "cr": cr,
                "feat_c0": T.X32[r0].view(bs, L, C), "feat_c1": T.X32[r1].view(bs, S, C)}

    def _graph_key(self, color0, color1, scale0, mask0):
        return (tuple(color0.shape), tuple(color1.shape), scale0 is not None, mask0 is not None, self.precision, bool(self.fp32_split),
                self.coarse_sim, self._img_dt(), self._split(), self._stem_k(), str(color0.device))

    def _coarse_stage_graphed(self, key, color0, color1, scale0, scale1, mask0=None, mask1=None):
        """HIP-graph replay of `_coarse_stage` (one graph per input shape / precision).  ~140 kernel launches collapse into one
        graph launch.  The graph's static input is the NHWC image tensor: the two layout kernels that fill it from the caller's
        NCHW images run eagerly in front of the replay, so the images are never copied as such."""
        ent = self._graphs.get(key)
        dt = self._dt()
        same = color0.shape[2:] == color1.shape[2:]
        groups = [[color0, color1]] if same else [[color0], [color1]]
        if ent is None:
            # host-side caches (weight packing does pageable H2D copies, the position table is built on the CPU) must be
            # filled BEFORE capture starts, whatever ran earlier
            self._prepack(color0.device)
            C = self.config["coarse"]["d_model"]
            half = lambda n: (n - 1) // 2 + 1   # noqa: E731  the three stride-2 convs (k7 p3, k3 p1, k3 p1)
            for c in (color0, color1):
                self._pos_encoding(C, half(half(half(c.shape[2]))), half(half(half(c.shape[3]))), c.device)
            sin = [[self._to_nhwc(g, self._img_dt()) for g in groups],
                   scale0.clone().float() if scale0 is not None else None,
                   scale1.clone().float() if scale1 is not None else None,
                   mask0.clone() if mask0 is not None else None, mask1.clone() if mask1 is not None else None]
            # the count buffer of the captured coarse matching: zeroed HERE, outside the capture (its health bit 1 is sticky)
            sin.append(torch.zeros(2 + color0.shape[0], dtype=torch.int32, device=color0.device))
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            # thread_local: other threads (e.g. RCCL's watchdog in multi-GPU runs) may issue HIP calls meanwhile
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                out = self._coarse_stage(sin[0], color0.shape[0], *sin[1:])
            while len(self._graphs) >= self.graph_cache_size:  # LRU eviction frees that graph's pool
                self._graphs.popitem(last=False)
            ent = self._graphs[key] = (graph, sin, out)
        else:
            self._graphs.move_to_end(key)
            for g, x in zip(groups, ent[1][0]):
                self._to_nhwc(g, self._img_dt(), out=x)
        graph, sin, out = ent
        small = []
        if scale0 is not None:
            small += [(scale0, sin[1]), (scale1, sin[2])]
        if mask0 is not None:
            small += [(mask0, sin[3]), (mask1, sin[4])]
        ops.copy_segments(small)
        graph.replay()
        return out

    @torch.no_grad()
    def forward(self, data):
        if self.precision == "fp32" and self.fp32_split and not ops.FP32_SPLIT:
            ops.FP32_SPLIT = True   # (module state of gim_amd.ops, read at every fp32 conv / linear launch)
            try:
                return self._forward(data)
            finally:
                ops.FP32_SPLIT = False
        return self._forward(data)

    def _forward(self, data):
        for k in ("image0", "image1", "color0", "color1"):
            if k not in data:
                raise KeyError(f"LoFTR.forward needs data[{k!r}] (loftr.py:54-63)")
        color0, color1 = data["color0"], data["color1"]
        if not color0.is_cuda:
            raise GimHipError("gim_amd LoFTR runs on the HIP device only (no CPU fallback): move the inputs "
                              "and the module to 'cuda'")
        dev = color0.device
        self._prepack(dev)   # first: packing may still change the mode (fp16 weights out of range -> bf16, see _prepack)
        dt = self._dt()
        tdt = torch_dtype(dt)
        cfg = self.config
        color0 = color0.contiguous().float()
        color1 = color1.contiguous().float()
        scale0, scale1 = data.get("scale0"), data.get("scale1")
        if scale0 is not None:
            scale0 = scale0.to(device=dev, dtype=torch.float32).contiguous()
            scale1 = scale1.to(device=dev, dtype=torch.float32).contiguous()

        mask0 = mask1 = None
        if "mask0" in data:  # [N, h/8, w/8] padding masks, '0' = padded (loftr.py:49-50, 77-79)
            mask0 = data["mask0"].to(device=dev).ne(0).to(torch.uint8).contiguous()
            mask1 = data["mask1"].to(device=dev).ne(0).to(torch.uint8).contiguous()
            exp0 = (color0.shape[0], color0.shape[2] // 8, color0.shape[3] // 8)
            exp1 = (color1.shape[0], color1.shape[2] // 8, color1.shape[3] // 8)
            if tuple(mask0.shape) != exp0 or tuple(mask1.shape) != exp1:
                raise ValueError(f"mask0/mask1 must be [N, H/8, W/8]: got {tuple(mask0.shape)}, {tuple(mask1.shape)}")

        data.update({"bs": data["image0"].size(0),
                     "hw0_i": data["image0"].shape[2:], "hw1_i": data["image1"].shape[2:]})
        bs = data["bs"]
        graphed = False
        if self.use_graph and self.debug is None:
            key = self._graph_key(color0, color1, scale0, mask0)
            if key in self._graphs or self._seen.get(key, 0) >= 1:
                try:
                    st = self._coarse_stage_graphed(key, color0, color1, scale0, scale1, mask0, mask1)
                    graphed = True
                except RuntimeError as e:
                    # only a failed *capture* (HIP graphs unsupported in this environment) lands here: input errors
                    # (ValueError / GimHipError from argument checks) were raised by this shape's eager first call
                    if key in self._graphs or "captur" not in str(e).lower():   # '... when stream is capturing', 'StreamCapture...'
                        raise
                    import warnings
                    warnings.warn(f"gim_amd: HIP graph capture failed ({e!r}); using eager kernel launches")
                    self.use_graph = False
                    self._graphs.clear()
        if not graphed:
            same = color0.shape[2:] == color1.shape[2:]
            idt = self._img_dt()
            xs = [self._to_nhwc([color0, color1], idt)] if same else [self._to_nhwc([color0], idt), self._to_nhwc([color1], idt)]
            st = self._coarse_stage(xs, bs, scale0, scale1, mask0, mask1)
            if self.use_graph and self.debug is None:   # counted only once the eager call went through (bad inputs raise above)
                self._seen[key] = self._seen.get(key, 0) + 1
                while len(self._seen) > 64:
                    self._seen.popitem(last=False)
        c0, c1, f0, f1, cr = st["c0"], st["c1"], st["f0"], st["f1"], st["cr"]
        if self.debug is not None:
            self.debug.update({k: st[k] for k in ("c0", "c1", "f0", "f1", "feat_c0", "feat_c1")})
        hw0_c, hw1_c = c0.shape[1:3], c1.shape[1:3]
        hw0_f, hw1_f = f0.shape[1:3], f1.shape[1:3]
        data.update({"hw0_c": torch.Size(hw0_c), "hw1_c": torch.Size(hw1_c),
                     "hw0_f": torch.Size(hw0_f), "hw1_f": torch.Size(hw1_f)})

        # 4./5. fine level (fine_preprocess.py:29-47, transformer on [M,25,128], fine_matching.py:15-74).  The reference synchronises on
        # the match count first (torch.where, coarse_matching.py:193).  With the fused fine kernel the launch goes out BEFORE the host
        # knows the count: it covers the capacity of the match lists and reads the count on the device, the asynchronous read-back
        # of the count (pinned host word + event, enqueued in front of it) overlaps the kernel -- the GPU used to idle 60-75 us per
        # forward between the read-back and the launch (profiles/r03_m kernel trace).
        fine = None
        dev_count = (self.fine_dev_count and self.fine_fused and self.debug is None and "fine_fused" in self._prepack(dev)
                     and f0.dtype in ops.HALF)
        if self._count_pin is None:
            self._count_pin = torch.empty(2, dtype=torch.int32).pin_memory()
        if dev_count:
            self._count_pin.copy_(cr.count[:2], non_blocking=True)   # [match count, health word of the coarse level]
            ev = torch.cuda.Event()
            ev.record()
            cap = cr.b_ids.numel()
            fine = self._fine_level(f0, f1, cr.b_ids, cr.i_ids, cr.j_ids, cr.mkpts1_c, scale1, "scale0" in data,
                                    hw0_c, hw1_c, data["hw0_i"], True, count=cr.count)
            ev.synchronize()
            M, health = int(self._count_pin[0]), int(self._count_pin[1])
            # bit 1 (non-finite fine-level output) is written by the fine kernel, which is still running: a replayed graph reuses its
            # count buffer, so the word read HERE carries the previous forward's bit (one batch late, no extra sync); eager forwards
            # and the first forwards after a weight / precision change wait for the kernel and look at once
            if self.precision == "fp16" and M > 0 and not health and (not graphed or self._health_sync_left > 0):
                self._health_sync_left = max(0, self._health_sync_left - 1)
                health = int(cr.count[1].item())
            ops.patch_last_fused_flops("fine_fused", 33.6e6 * M)
            fine = (fine[0][:M], fine[1][:M], None, None) if M > 0 else None
            assert M <= cap
        else:
            self._count_pin.copy_(cr.count[:2])  # the one host sync the reference also has
            M, health = int(self._count_pin[0]), int(self._count_pin[1])
        if health:
            if health & 6:
                cr.count[1:2].zero_()   # sticky bits: acknowledged
            if self._range_guard(health):
                return self.forward(data)   # the module is in bf16 now: same inputs, once more
        self._generation += 1
        if not dev_count and M > 0:
            fine = self._fine_level(f0, f1, cr.b_ids[:M], cr.i_ids[:M], cr.j_ids[:M], cr.mkpts1_c[:M], scale1, "scale0" in data,
                                    hw0_c, hw1_c, data["hw0_i"], self.fine_fused)
        # graph replays reuse their output buffers: hand out private copies of the (small) match lists -- one launch for all of
        # them and the all-false gt_mask (mconf == 0 never holds: mconf > thr), not one torch copy kernel each
        if graphed:
            ib = torch.empty(4, M, dtype=torch.int64, device=dev)
            fb = torch.empty(5 * M, dtype=torch.float32, device=dev)
            gt_mask = torch.empty(M, dtype=torch.bool, device=dev)
            b_ids, i_ids, j_ids, m_bids = ib[0], ib[1], ib[2], ib[3]
            mkpts0_c, mkpts1_c, mconf = fb[:2 * M].view(M, 2), fb[2 * M:4 * M].view(M, 2), fb[4 * M:]
            ops.copy_segments([(cr.b_ids[:M], b_ids), (cr.i_ids[:M], i_ids), (cr.j_ids[:M], j_ids), (cr.b_ids[:M], m_bids),
                               (cr.mkpts0_c[:M], mkpts0_c), (cr.mkpts1_c[:M], mkpts1_c), (cr.mconf[:M], mconf), (None, gt_mask)])
        else:
            ib = torch.empty(M, dtype=torch.int64, device=dev)
            gt_mask = torch.empty(M, dtype=torch.bool, device=dev)
            b_ids, i_ids, j_ids, m_bids = cr.b_ids[:M], cr.i_ids[:M], cr.j_ids[:M], ib
            mkpts0_c, mkpts1_c, mconf = cr.mkpts0_c[:M], cr.mkpts1_c[:M], cr.mconf[:M]
            ops.copy_segments([(cr.b_ids[:M], m_bids), (None, gt_mask)])
        data.update({"conf_matrix": LazyConfMatrix(cr, self, self._generation if graphed else None)})   # key order = the reference's
        data.update({"b_ids": b_ids, "i_ids": i_ids, "j_ids": j_ids, "gt_mask": gt_mask, "m_bids": m_bids,
                     "mkpts0_c": mkpts0_c, "mkpts1_c": mkpts1_c, "mconf": mconf})
        data.update({"W": self.W})
        if fine is None:
            data.update({"expec_f": torch.empty(0, 3, device=dev),
                         "mkpts0_f": data["mkpts0_c"], "mkpts1_f": data["mkpts1_c"]})
            return
        expec_f, mkpts1_f, fine0, fine1 = fine
        if self.debug is not None:
            self.debug.update({"fine0": fine0, "fine1": fine1})
        data.update({"expec_f": expec_f, "mkpts0_f": data["mkpts0_c"], "mkpts1_f": mkpts1_f})

    def _range_guard(self, health):
        """A non-finite value reached coarse matching (bit 0) or left the fine level (bit 1).  In the fp16 mode that is the IEEE-fp16
        range (|activation| > 65504 somewhere in the backbone: a checkpoint whose un-normalised ResNet streams run hot): warn, switch
        this module to bf16 for good -- same kernels, fp32's exponent range -- and tell the caller to run the batch again.  In the
        other modes the inputs or weights themselves were not finite: warn only.  The reference computes in fp32 and has no such
        case (networks/loftr/utils/coarse_matching.py:174-195 would return no match for a NaN row, silently)."""
        import warnings
        what = " and ".join(w for b, w in ((4, "a residual-stream value beyond 65504"), (1, "non-finite coarse similarities"),
                                           (2, "non-finite fine-level outputs")) if health & b)
        if self.precision == "fp16":
            warnings.warn(f"gim_amd LoFTR: {what} in the fp16 mode (an activation left the IEEE-fp16 range); "
                          "falling back to precision='bf16' for this module and re-running the batch")
            self.fp16_overflowed = True
            self.set_precision("bf16", coarse_sim="fp32" if self.coarse_sim == "fp32" else None)   # a caller's fp32 similarity survives the fallback
            return True
        warnings.warn(f"gim_amd LoFTR: {what} in the {self.precision} mode: the inputs or the weights are not finite")
        return False

    def _fine_level(self, f0, f1, b_ids, i_ids, j_ids, mkpts1_c, scale1, has_s0, hw0_c, hw1_c, hw0_i, fused, count=None):
        """FinePreprocess + loftr_fine + FineMatching (loftr.py:84-91) for M > 0 matches on the NHWC fine maps f0 / f1.
        Returns (expec_f, mkpts1_f, fine0, fine1); fine0/fine1 = fp32 [M, WW, C] transformer outputs (None unless
        self.debug is set on the fused path)."""
        dev = f0.device
        dt = self._dt()
        M, W, WW, Cf = b_ids.numel(), self.W, self.W * self.W, self.config["fine"]["d_model"]
        P = self._prepack(dev)
        hw0_f, hw1_f = f0.shape[1:3], f1.shape[1:3]
        stride = hw0_f[0] // hw0_c[0]
        fscale = hw0_i[0] / hw0_f[0]
        if fused and "fine_fused" in P and f0.dtype in ops.HALF:
            wts, lnp, eps = P["fine_fused"]
            return ops.fine_fused(f0, f1, b_ids, i_ids, j_ids, mkpts1_c, scale1 if has_s0 else None, wts, lnp, M,
                                  hw0_c[1], hw1_c[1], stride, W, fscale, eps, has_s0, debug=self.debug is not None, count=count)
        assert count is None, "the device-side match count needs the fused fine kernel"
        F = self._TfBuffers(2 * M * WW, Cf, torch_dtype(dt), dev)
        ops.fine_gather(f0, f1, b_ids, i_ids, j_ids, M, hw0_c[1], hw1_c[1], stride, W, F.X32, F.CAT[:, :Cf])
        self._transformer(P, "f", self.loftr_fine, F, M, WW, M, WW)
        expec_f, mkpts1_f = ops.fine_match(F.X32[:M * WW], F.X32[M * WW:], mkpts1_c, b_ids,
                                           scale1 if has_s0 else None, M, WW, fscale, has_s0)
        return expec_f, mkpts1_f, F.X32[:M * WW].view(M, WW, Cf), F.X32[M * WW:].view(M, WW, Cf)
