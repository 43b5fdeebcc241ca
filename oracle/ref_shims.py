"""TEST INFRASTRUCTURE ONLY -- stand-ins that let the *real* reference import in the authoring container.

The reference (`/root/reference`, xuelunshen/gim) imports two third-party packages on the gim_loftr
path that are not installed here and cannot be installed (no network):

  * kornia 0.6.10  -- `networks/loftr/utils/fine_matching.py:5-6` uses
        kornia.geometry.subpix.dsnt.spatial_expectation2d   (line 49)
        kornia.utils.grid.create_meshgrid                    (line 50)
  * yacs           -- `networks/loftr/config.py:1` (`CfgNode`)

  * omegaconf      -- `networks/lightglue/models/base_model.py:8-9`, `models/matchers/lightglue.py:8`
                      (gim_lightglue path; `install_omegaconf()`)

`install()` registers minimal modules under those names in `sys.modules` (restating the published
behaviour of the two kornia functions and a dict-backed CfgNode) and puts `/root/reference` on
`sys.path`, so `oracle/make_golden.py` can run the reference's own modules to pin `oracle/loftr_oracle.py`.

Nothing here is copied from the reference or from kornia/yacs sources; nothing here is imported by the
product (`gim_amd/`).  `/root/reference` does not exist on the GPU box: only `oracle/make_golden.py`
(run by hand in the authoring container) calls `install()`.
"""
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


def _create_meshgrid(height, width, normalized_coordinates=True, device=None, dtype=torch.float32):
    """kornia.utils.grid.create_meshgrid: [1,H,W,2] grid, last dim = (x, y), in [-1,1] if normalised."""
    xs = torch.linspace(0, width - 1, width, device=device, dtype=dtype)
    ys = torch.linspace(0, height - 1, height, device=device, dtype=dtype)
    if normalized_coordinates:
        xs = (xs / (width - 1) - 0.5) * 2
        ys = (ys / (height - 1) - 0.5) * 2
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    return torch.stack([gx, gy], dim=-1).unsqueeze(0)


def _spatial_expectation2d(inp, normalized_coordinates=True):
    """kornia.geometry.subpix.dsnt.spatial_expectation2d: inp [B,N,H,W] (a distribution) -> [B,N,2] (x,y)."""
    b, n, h, w = inp.shape
    grid = _create_meshgrid(h, w, normalized_coordinates, inp.device).to(inp.dtype)
    pos_x = grid[..., 0].reshape(-1)
    pos_y = grid[..., 1].reshape(-1)
    flat = inp.reshape(b, n, -1)
    ex = torch.sum(pos_x * flat, -1, keepdim=True)
    ey = torch.sum(pos_y * flat, -1, keepdim=True)
    return torch.cat([ex, ey], -1).reshape(b, n, 2)


class CfgNode(dict):
    """Dict-backed stand-in for yacs.config.CfgNode (attribute access + clone), enough for
    `networks/loftr/config.py:3-77`."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        out = CfgNode()
        for k, v in self.items():
            out[k] = v.clone() if isinstance(v, CfgNode) else v
        return out


def lower_config(cfg):
    """Restates `networks/loftr/misc.py:13-16` (that module itself imports loguru / pytorch_lightning)."""
    if not isinstance(cfg, CfgNode):
        return cfg
    return {k.lower(): lower_config(v) for k, v in cfg.items()}


def install():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    if "kornia" not in sys.modules:
        k = mod("kornia")
        kg = mod("kornia.geometry")
        ks = mod("kornia.geometry.subpix")
        kd = mod("kornia.geometry.subpix.dsnt")
        ku = mod("kornia.utils")
        kug = mod("kornia.utils.grid")
        kd.spatial_expectation2d = _spatial_expectation2d
        ks.dsnt = kd
        kg.subpix = ks
        k.geometry = kg
        kug.create_meshgrid = _create_meshgrid
        ku.grid = kug
        ku.create_meshgrid = _create_meshgrid
        k.utils = ku
    if "yacs" not in sys.modules:
        y = mod("yacs")
        yc = mod("yacs.config")
        yc.CfgNode = CfgNode
        y.config = yc
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


class DictConfig(dict):
    """Stand-in for omegaconf.DictConfig: a dict with attribute access (nested dicts converted)."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = DictConfig(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class _OmegaConf:
    @staticmethod
    def create(d=None):
        return DictConfig(d or {})

    @staticmethod
    def merge(*cfgs):
        out = DictConfig()
        for c in cfgs:
            for k, v in dict(c).items():
                if isinstance(v, dict) and isinstance(out.get(k), dict):
                    out[k] = _OmegaConf.merge(out[k], v)
                else:
                    out[k] = DictConfig(v) if isinstance(v, dict) else v
        return out

    @staticmethod
    def set_struct(conf, flag):
        pass

    @staticmethod
    def set_readonly(conf, flag):
        pass


def install_omegaconf():
    """Registers the omegaconf stand-in so `networks.lightglue.*` imports."""
    import contextlib
    install()
    if "omegaconf" not in sys.modules:
        m = types.ModuleType("omegaconf")
        m.OmegaConf = _OmegaConf
        m.DictConfig = DictConfig
        m.read_write = lambda conf: contextlib.nullcontext()
        m.open_dict = lambda conf: contextlib.nullcontext()
        sys.modules["omegaconf"] = m


def reference_lightglue_models():
    """(SuperPoint, LightGlue) built exactly as `trainer/lightning.py:49-60` / `demo.py:338-349` build them."""
    install_omegaconf()
    from networks.lightglue.superpoint import SuperPoint
    from networks.lightglue.models.matchers.lightglue import LightGlue
    detector = SuperPoint({"max_num_keypoints": 2048, "force_num_keypoints": True, "detection_threshold": 0.0,
                           "nms_radius": 3, "trainable": False})
    model = LightGlue({"filter_threshold": 0.1, "flash": False, "checkpointed": True})
    return detector.eval(), model.eval()


def install_dkm():
    """Stand-ins that let `networks.dkm.*` import: an empty `cv2` (only host-side pose utilities use
    it), `torchvision.transforms` names that `dkm/utils/utils.py:4-5,178-199` touches at import / construction,
    and `torchvision.models.resnet50` restated from the reference's OWN ResNet/Bottleneck
    (`networks/loftr/backbone/resnet.py:71-167`, a verbatim torchvision architecture with maxpool / layer4 / fc
    commented out) by adding those three members back -- so the parameter names are torchvision's
    (`encoder.net.conv1 ... layer4.2.bn3`), which is what gim_dkm checkpoints carry."""
    import enum
    import torch.nn as nn
    install()
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tr = types.ModuleType("torchvision.transforms")
        trf = types.ModuleType("torchvision.transforms.functional")
        tvm = types.ModuleType("torchvision.models")

        class InterpolationMode(enum.Enum):
            NEAREST = "nearest"
            BILINEAR = "bilinear"
            BICUBIC = "bicubic"

        def _unavailable(*a, **k):
            raise NotImplementedError("torchvision is not installed; the oracle path feeds tensors directly")

        def resnet50(pretrained=False, weights=None, replace_stride_with_dilation=None, **kw):
            from networks.loftr.backbone.resnet import Bottleneck, ResNet
            net = ResNet(Bottleneck, [3, 4, 6, 3], replace_stride_with_dilation=replace_stride_with_dilation)
            net.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
            net.layer4 = net._make_layer(Bottleneck, 512, 3, stride=2)
            net.fc = nn.Linear(2048, 1000)
            return net

        trf.InterpolationMode = InterpolationMode
        tr.InterpolationMode = InterpolationMode
        tr.functional = trf
        tr.Resize = tr.Normalize = _unavailable
        tvm.resnet50 = resnet50
        tvm.resnet18 = _unavailable
        tv.transforms, tv.models = tr, tvm
        for name, m in (("torchvision", tv), ("torchvision.transforms", tr), ("torchvision.transforms.functional", trf),
                        ("torchvision.models", tvm)):
            sys.modules[name] = m


def reference_dkm(h, w, upsample_res=None, **attrs):
    """DKMv3 built and configured the way `trainer/lightning.py:30-37` / `demo.py:328` do (tensor inputs, symmetric)."""
    install_dkm()
    from networks.dkm.models.model_zoo.DKMv3 import DKMv3
    model = DKMv3(None, h, w, upsample_preds=upsample_res is not None)
    if upsample_res is not None:
        model.upsample_res = tuple(upsample_res)
    for k, v in attrs.items():
        setattr(model, k, v)
    return model.eval()


def install_roma(dino_state_dict=None):
    """Stand-ins that let `networks.roma.*` import and construct:
      * `xformers.ops` (memory_efficient_attention = scaled-dot-product attention on [B,N,H,D], unbind, a SwiGLU base
        class) -- without it `dino.py:275` fails at import;
      * `torchvision.models.vgg19_bn` restated from its published configuration "E" with BatchNorm (`roma.py:142` takes
        `.features[:40]`);
      * `torch.hub.load_state_dict_from_url` patched to return `dino_state_dict` (the reference downloads the DINOv2
        ViT-L/14 weights in `CNNandDinov2.__init__`, roma.py:591-595; there is no network here)."""
    import torch.nn as nn
    install_dkm()
    if "xformers" not in sys.modules:
        xf = types.ModuleType("xformers")
        xo = types.ModuleType("xformers.ops")

        def memory_efficient_attention(q, k, v, attn_bias=None):
            assert attn_bias is None
            o = torch.nn.functional.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
            return o.transpose(1, 2)

        class SwiGLU(nn.Module):
            def __init__(self, in_features, hidden_features=None, out_features=None, bias=True):
                super().__init__()

        xo.memory_efficient_attention, xo.unbind, xo.SwiGLU = memory_efficient_attention, torch.unbind, SwiGLU
        xf.ops = xo
        sys.modules["xformers"], sys.modules["xformers.ops"] = xf, xo
    tvm = sys.modules["torchvision.models"]
    if not hasattr(tvm, "vgg19_bn"):
        def vgg19_bn(pretrained=False, **kw):
            cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
            layers, ci = [], 3
            for v in cfg:
                if v == "M":
                    layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
                else:
                    layers += [nn.Conv2d(ci, v, kernel_size=3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
                    ci = v
            m = nn.Module()
            m.features = nn.Sequential(*layers)
            return m
        tvm.vgg19_bn = vgg19_bn
    if dino_state_dict is not None:
        torch.hub.load_state_dict_from_url = lambda *a, **k: dino_state_dict


def reference_roma(h, w, dino_state_dict, upsample_res=None, **attrs):
    """RoMa built the way `demo.py:332` / `trainer/lightning.py:41` do (`RoMa(img_size=[672])`), at a test resolution."""
    install_roma(dino_state_dict)
    from networks.roma.roma import RoMa
    model = RoMa(img_size=[h, w], upsample_preds=upsample_res is not None)
    if upsample_res is not None:
        model.upsample_res = tuple(upsample_res)
    for k, v in attrs.items():
        setattr(model, k, v)
    return model.eval()


def reference_loftr_config():
    """The effective gim_loftr config dict (`demo.py:333-335`: lower_config(get_cfg_defaults())['loftr'])."""
    install()
    from networks.loftr.config import get_cfg_defaults
    return lower_config(get_cfg_defaults())["loftr"]
