"""TEST INFRASTRUCTURE ONLY -- BASELINE config 1: `gim_lightglue` on assets/demo a1.png <-> a2.png through the REFERENCE's own
CPU modules (plumbing run: no checkpoint ships with the reference, so the seeded weights of `make_state_dicts(0)` are used).

Run by hand in the authoring container (needs /root/reference):  python oracle/make_golden_demo.py
Follows demo.py:472-511 with the reference's `SuperPoint` / `LightGlue` (imported through `oracle/ref_shims.py`), on the two
demo images pre-processed by `gim_amd.demo.read_image / preprocess` (OpenCV / torchvision are not installed here, so the
decode + resize is ours on both sides; `resize_max=512` keeps the CPU run to about a minute).  Stores the REFERENCE's outputs in
tests/golden/demo/lightglue_a1_a2.npz; the two PNGs are committed next to it as fixtures.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import lightglue_oracle as O  # noqa: E402
from oracle import ref_shims  # noqa: E402
from gim_amd import demo as D  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "demo")
RESIZE_MAX = 512


@torch.no_grad()
def main():
    ref_shims.install_omegaconf()
    from networks.lightglue.superpoint import SuperPoint
    from networks.lightglue.models.matchers.lightglue import LightGlue
    detector = SuperPoint({"max_num_keypoints": 2048, "force_num_keypoints": True, "detection_threshold": 0.0,
                           "nms_radius": 3, "trainable": False}).eval()
    model = LightGlue({"filter_threshold": 0.1, "flash": False, "checkpointed": True}).eval()
    sp_sd, lg_sd = O.make_state_dicts(0)
    detector.load_state_dict(sp_sd)
    model.load_state_dict(lg_sd)
    p0, p1 = os.path.join(OUT, "a1.png"), os.path.join(OUT, "a2.png")
    image0, scale0 = D.preprocess(D.read_image(p0), resize_max=RESIZE_MAX)
    image1, scale1 = D.preprocess(D.read_image(p1), resize_max=RESIZE_MAX)
    gray0 = D.preprocess(D.read_image(p0, grayscale=True), grayscale=True, resize_max=RESIZE_MAX)[0][None]
    gray1 = D.preprocess(D.read_image(p1, grayscale=True), grayscale=True, resize_max=RESIZE_MAX)[0][None]
    data = dict(color0=image0[None], color1=image1[None], image0=image0[None], image1=image1[None], gray0=gray0, gray1=gray1,
                scale0=torch.tensor(scale0)[None], scale1=torch.tensor(scale1)[None],
                size0=torch.tensor(gray0.shape[-2:][::-1])[None], size1=torch.tensor(gray1.shape[-2:][::-1])[None])
    pred = {}
    pred.update({k + "0": v for k, v in detector({"image": data["gray0"]}).items()})
    pred.update({k + "1": v for k, v in detector({"image": data["gray1"]}).items()})
    pred.update(model({**pred, **data, **{"image_size0": data["size0"], "image_size1": data["size1"]}}))
    kpts0 = torch.cat([kp * s for kp, s in zip(pred["keypoints0"], data["scale0"][:, None])])
    kpts1 = torch.cat([kp * s for kp, s in zip(pred["keypoints1"], data["scale1"][:, None])])
    m_bids = torch.nonzero(pred["keypoints0"].sum(dim=2) > -1)[:, 0]
    matches = pred["matches"]
    kpts0 = torch.cat([kpts0[m_bids == 0][matches[0][..., 0]]])
    kpts1 = torch.cat([kpts1[m_bids == 0][matches[0][..., 1]]])
    mconf = torch.cat(pred["scores"])
    print("gim_lightglue a1<->a2 (reference modules, seeded weights):", gray0.shape, "matches", len(mconf))
    np.savez_compressed(os.path.join(OUT, "lightglue_a1_a2.npz"), resize_max=RESIZE_MAX, shape=np.array(gray0.shape[-2:]),
                        scale0=scale0, scale1=scale1, keypoints0=pred["keypoints0"].numpy(), keypoints1=pred["keypoints1"].numpy(),
                        matches0=pred["matches0"].numpy(), matching_scores0=pred["matching_scores0"].numpy(),
                        mkpts0_f=kpts0.numpy().astype(np.float32), mkpts1_f=kpts1.numpy().astype(np.float32), mconf=mconf.numpy())


if __name__ == "__main__":
    main()
