"""SuperPoint on MI355X: the reference's module surface (`networks/lightglue/superpoint.py:153-354`) over
hand-written HIP.

Drop-in contract:
  * `SuperPoint(conf_dict)` with the reference's `default_conf` keys; `state_dict()` has the reference's 24
    tensors (conv1a ... convDb), so the `superpoint.`-stripped halves of gim_lightglue checkpoints load
    unchanged (`demo.py:378-386`);
  * `model({'image': [B,1|3,H,W]})` -> `{'keypoints': [B,K,2] (x, y) + 0.5, 'descriptors': [B,K,256]}`;
  * quirks kept: a caller-supplied `image_size` is ignored (the reference overwrites it with the canvas size,
    superpoint.py:207), `legacy_sampling` descriptor interpolation, keypoints padded with uniform random
    points when fewer than `max_num_keypoints` survive (`pad_and_stack(mode='random_c')`, misc.py:44-55).
  * the reference raises IndexError for B > 1 (superpoint.py:254); here a batch is the reference applied to
    each image.

The nn.Module tree only holds parameters; `forward` launches libgimhip kernels (conv_igemm + superpoint.hip)
through ctypes.  No CPU / eager fallback.
"""
import os

import torch

from ..precision import resolve as resolve_precision
import torch.nn as nn

from .. import ops
from .._lib import ACT_RELU, GIM_BF16, GIM_F16, GIM_F32, GimHipError
from ..packing import pack_conv, torch_dtype


class SuperPoint(nn.Module):
    default_conf = {
        "name": None, "trainable": True, "freeze_batch_normalization": False, "timeit": False,
        "has_detector": True, "has_descriptor": True, "descriptor_dim": 256,
        "sparse_outputs": True, "dense_outputs": False, "nms_radius": 4, "refinement_radius": 0,
        "detection_threshold": 0.005, "max_num_keypoints": -1, "max_num_keypoints_val": None,
        "force_num_keypoints": False, "randomize_keypoints_training": False, "remove_borders": 4,
        "legacy_sampling": True,
    }
    required_data_keys = ["image"]

    def __init__(self, conf):
        super().__init__()
        self.conf = c = {**self.default_conf, **dict(conf)}
        if not (c["has_detector"] and c["has_descriptor"] and c["sparse_outputs"]) or c["dense_outputs"]:
            raise NotImplementedError("gim_lightglue uses the sparse detector+descriptor outputs")
        if not c["legacy_sampling"] or c["refinement_radius"] != 0 or c["descriptor_dim"] != 256:
            raise NotImplementedError("only the gim configuration (legacy_sampling, no refinement, 256-d) is built")
        self.precision = resolve_precision(c.get("precision"), "SuperPoint")
        c1, c2, c3, c4, c5 = 64, 64, 128, 128, 256
        for name, ci, co, k in (("conv1a", 1, c1, 3), ("conv1b", c1, c1, 3), ("conv2a", c1, c2, 3), ("conv2b", c2, c2, 3),
                                ("conv3a", c2, c3, 3), ("conv3b", c3, c3, 3), ("conv4a", c3, c4, 3), ("conv4b", c4, c4, 3),
                                ("convPa", c4, c5, 3), ("convPb", c5, 65, 1), ("convDa", c4, c5, 3), ("convDb", c5, 256, 1)):
            setattr(self, name, nn.Conv2d(ci, co, kernel_size=k, stride=1, padding=k // 2))
        if not c["trainable"]:
            for p in self.parameters():
                p.requires_grad = False
        self._packed = None

    def load_state_dict(self, state_dict, *args, **kwargs):
        self._packed = None
        return super().load_state_dict(state_dict, *args, **kwargs)

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    # ---- one-time weight packing ------------------------------------------------------------------------
    def _prepack(self, device):
        dt = {"bf16": GIM_BF16, "fp16": GIM_F16, "fp32": GIM_F32}[self.precision]
        pk = {}

        def conv(name, w, b, cin_pad=None):
            pk[name] = pack_conv(w, None, dt, device, stride=1, pad=w.shape[-1] // 2, cin_pad=cin_pad, bias=b)

        w1 = self.conv1a.weight.detach().float()
        conv("conv1a", w1, self.conv1a.bias)
        # RGB input: image = sum_c rgb_c * (0.299, 0.587, 0.114) (superpoint.py:209-211) folded into the weights
        g = torch.tensor([0.299, 0.587, 0.114], dtype=torch.float32, device=w1.device).view(1, 3, 1, 1)
        conv("conv1a_rgb", w1 * g, self.conv1a.bias)
        for n in ("conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "convPb", "convDb"):
            m = getattr(self, n)
            conv(n, m.weight, m.bias)
        # both heads read conv4b's output: one 128 -> 512 conv (convPa | convDa)
        conv("heads", torch.cat([self.convPa.weight, self.convDa.weight], 0).detach(),
             torch.cat([self.convPa.bias, self.convDa.bias], 0).detach())
        self._packed = (pk, dt, device)

    @torch.no_grad()
    def forward(self, data):
        for key in self.required_data_keys:
            assert key in data, f"Missing key {key} in data"
        image = data["image"]
        if not image.is_cuda:
            raise GimHipError("gim_amd SuperPoint needs device (cuda/HIP) tensors: there is no CPU fallback")
        dev = image.device
        if self._packed is None or self._packed[2] != dev or self._packed[1] != ({"bf16": GIM_BF16, "fp16": GIM_F16, "fp32": GIM_F32}[self.precision]):
            self._prepack(dev)
        pk, dt, _ = self._packed
        tdt = torch_dtype(dt)
        c = self.conf
        B, C, H, W = image.shape
        if C not in (1, 3) or H % 8 or W % 8:
            raise GimHipError(f"SuperPoint: image must be [B,1|3,H,W] with H, W multiples of 8, got {tuple(image.shape)}")
        data["image_size"] = torch.tensor(image.shape[-2:][::-1])[None]  # superpoint.py:207 (callers may read it back)
        first = pk["conv1a_rgb" if C == 3 else "conv1a"]
        x = torch.empty(B, H, W, first.cin_pad, dtype=tdt, device=dev)
        ops.nchw_to_nhwc(image.contiguous().float(), x)
        x = ops.conv2d(x, first, ACT_RELU)
        x = ops.maxpool2x2(ops.conv2d(x, pk["conv1b"], ACT_RELU))
        x = ops.conv2d(x, pk["conv2a"], ACT_RELU)
        x = ops.maxpool2x2(ops.conv2d(x, pk["conv2b"], ACT_RELU))
        x = ops.conv2d(x, pk["conv3a"], ACT_RELU)
        x = ops.maxpool2x2(ops.conv2d(x, pk["conv3b"], ACT_RELU))
        x = ops.conv2d(x, pk["conv4a"], ACT_RELU)
        x = ops.conv2d(x, pk["conv4b"], ACT_RELU)
        h, w = H // 8, W // 8
        heads = ops.conv2d(x, pk["heads"], ACT_RELU).view(B * h * w, 512)
        logits = torch.empty(B * h * w, pk["convPb"].n_store, dtype=torch.float32, device=dev)
        ops.linear(heads[:, :256], pk["convPb"], logits)
        dense = torch.empty(B * h * w, 256, dtype=torch.float32, device=dev)
        ops.linear(heads[:, 256:], pk["convDb"], dense)
        scores = ops.sp_scores(logits, B, h, w)
        nms = ops.sp_nms(scores, int(c["nms_radius"]), int(c["remove_borders"]))
        max_kps = c["max_num_keypoints"]
        if not self.training and c["max_num_keypoints_val"] is not None:
            max_kps = c["max_num_keypoints_val"]
        if max_kps is None or max_kps <= 0:
            raise NotImplementedError("max_num_keypoints must be set (gim uses 2048): the extraction kernel is a top-k")
        kpts, ksc, nvalid = ops.sp_topk(nms, int(max_kps), float(c["detection_threshold"]))
        nv = nvalid.tolist()  # one small read-back (the reference syncs in torch.where, superpoint.py:261)
        if min(nv) < max_kps:
            if c["force_num_keypoints"]:
                # pad_and_stack(mode='random_c'): uniform between the min / max of the found coordinates, per axis
                # (bounds (0, min(w, h)) when nothing was found); scores padded with zeros (superpoint.py:313-326)
                for b, n in enumerate(nv):
                    if n >= max_kps:
                        continue
                    for ax in range(2):
                        lo = kpts[b, :n, ax].min().item() if n > 0 else 0.0
                        hi = kpts[b, :n, ax].max().item() if n > 0 else float(min(W, H))
                        kpts[b, n:, ax] = torch.empty(max_kps - n, device=dev).uniform_(lo, hi)
                    ksc[b, n:] = 0
            elif B == 1:
                kpts, ksc = kpts[:, :nv[0]].contiguous(), ksc[:, :nv[0]].contiguous()
            else:
                raise GimHipError("images of one batch produced different keypoint counts; set force_num_keypoints "
                                  "(the reference's torch.stack fails the same way, superpoint.py:328-329)")
        K = kpts.shape[1]
        desc = torch.empty(B, K, 256, dtype=torch.float32, device=dev)
        if K > 0:
            ops.sp_sample_desc(dense, kpts, h, w, desc.view(B * K, 256), None)
        pred = {"keypoints": kpts + 0.5, "descriptors": desc}
        self._debug = {"scores": scores, "nms": nms, "keypoint_scores": ksc, "nvalid": nv, "dense_raw": dense, "logits": logits}
        return pred
