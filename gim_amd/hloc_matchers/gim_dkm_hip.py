"""hloc dense-matcher plugin `gim_dkm_hip`: the reference's `hloc/matchers/dkm.py` (class `LoFTR(BaseModel)`, :15-154) with the
HIP engine underneath.

Same conf (`weights`: file under `weights/`, `max_num_matches`), same checkpoint handling (:27-38: unwrap `state_dict`, strip
`model.`, drop `encoder.net.fc*`), same data contract:
    model({'image0', 'image1', 'name0', 'name1'}) -> {'keypoints0', 'keypoints1', 'scores', 'batch_indexes'}
incl. the image0/image1 swap ("we refine kpts in image0", :44-55), the semantic-mask blackout read from
`$GIMRECONSTRUCTION/../segment/<name>.npy` (:65-90; skipped when the environment variable / file is absent, or pass
`mask0` / `mask1` arrays in `data`), padding to 672x896, 8192 samples and top-k.  The arithmetic is
`gim_amd.adapters.HlocDenseMatcher` on `gim_amd.dkm.DKMv3`; there is no CPU fallback (`.to('cuda')` as hloc does).
"""
import os
from os.path import join
from pathlib import Path

import numpy as np
import torch

from ..adapters import HlocDenseMatcher
from ..dkm import DKMv3
from .base import BaseModel


def _nearest_resize(mask, hw):
    """cv2.resize(mask, (w, h), interpolation=cv2.INTER_NEAREST) (dkm.py:67-69): src index = floor(dst * src / dst_size)"""
    h, w = hw
    ys = np.minimum((np.arange(h) * (mask.shape[0] / h)).astype(np.int64), mask.shape[0] - 1)
    xs = np.minimum((np.arange(w) * (mask.shape[1] / w)).astype(np.int64), mask.shape[1] - 1)
    return mask[ys][:, xs]


class GimDkmHip(BaseModel):
    default_conf = {
        "weights": None,            # file name under weights/ (reference: 'gim_dkm_100h.ckpt'); None = keep the module's init
        "max_num_matches": None,
        "precision": None,          # 'bf16' (default of the engine) or 'fp32'
    }
    required_inputs = ["image0", "image1"]

    def _init(self, conf):
        self.h, self.w = 672, 896
        kw = {"precision": conf["precision"]} if conf.get("precision") else {}
        model = DKMv3(None, self.h, self.w, upsample_preds=True, **kw)
        if conf.get("weights"):
            path = conf["weights"] if os.path.isabs(conf["weights"]) else join("weights", conf["weights"])
            state_dict = torch.load(path, map_location="cpu")
            if "state_dict" in state_dict.keys():
                state_dict = state_dict["state_dict"]
            for k in list(state_dict.keys()):   # dkm.py:32-37: strip `model.`, drop the ResNet's unused fc head (under either name)
                v = state_dict.pop(k)
                nk = k.replace("model.", "", 1) if k.startswith("model.") else k
                if "encoder.net.fc" not in nk:
                    state_dict[nk] = v
            model.load_state_dict(state_dict)
        self.net = model
        self.adapter = HlocDenseMatcher(model, self.h, self.w, conf["max_num_matches"], 8192)

    def _segment_mask(self, name, hw):
        root = os.environ.get("GIMRECONSTRUCTION")
        if not root or name is None:
            return None
        name = name[0] if isinstance(name, (list, tuple)) else name
        path = Path(root) / ".." / "segment" / "{}.npy".format(name[:-4])
        if not path.exists():
            return None
        mask = np.load(path)
        if mask.shape[:2] != tuple(hw):
            mask = _nearest_resize(mask, hw)
        return mask

    def _forward(self, data):
        d = {"image0": data["image0"], "image1": data["image1"]}
        for i in ("0", "1"):
            m = data.get("mask" + i)
            if m is None:
                m = self._segment_mask(data.get("name" + i), data["image" + i].shape[-2:])
            if m is not None:
                d["mask" + i] = m
        return self.adapter(d)
