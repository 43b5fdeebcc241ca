"""TEST INFRASTRUCTURE ONLY -- pins `oracle/lightglue_oracle.py` to the reference's own modules.

Run by hand in the authoring container (needs /root/reference):  python oracle/make_golden_lightglue.py
Imports the reference's SuperPoint / LightGlue through `oracle/ref_shims.py::install_omegaconf`, loads the
seeded weights of `make_state_dicts(0)` into them, runs reference and restatement on the same seeded inputs,
asserts agreement and writes the vectors to tests/golden/lg_*.npz (inputs are regenerated from seeds by the
tests; outputs of the REFERENCE are what is stored).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import lightglue_oracle as O  # noqa: E402
from oracle import ref_shims  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def build_reference(k):
    ref_shims.install_omegaconf()
    from networks.lightglue.superpoint import SuperPoint
    from networks.lightglue.models.matchers.lightglue import LightGlue
    det = SuperPoint({"max_num_keypoints": k, "force_num_keypoints": True, "detection_threshold": 0.0,
                      "nms_radius": 3, "trainable": False}).eval()
    lg = LightGlue({"filter_threshold": 0.1, "flash": False, "checkpointed": True}).eval()
    sp_sd, lg_sd = O.make_state_dicts(0)
    det.load_state_dict(sp_sd)
    lg.load_state_dict(lg_sd)
    return det, lg, sp_sd, lg_sd


def run_det(det, img, size=None):
    """the reference detector image by image (it cannot take B > 1, superpoint.py:207,254), stacked"""
    outs = [det({"image": img[i:i + 1], **({"image_size": size} if size is not None else {})}) for i in range(len(img))]
    return {k: torch.cat([o[k] for o in outs]) for k in ("keypoints", "descriptors")}


def close(a, b, tol, what):
    err = (a - b).abs().max().item()
    assert err <= tol, (what, err)
    return err


@torch.no_grad()
def main():
    K = 128
    det, lg, sp_sd, lg_sd = build_reference(K)
    conf = dict(O.SP_CONF, max_num_keypoints=K)

    # ---- SuperPoint: dense heads, NMS, selection, descriptor sampling ------------------------------------
    from networks.lightglue import superpoint as RSP
    img = O.seeded_gray(2, 96, 128, 11)
    size = torch.tensor([[120, 90]])       # ignored by the reference (superpoint.py:207 overwrites it): pinned here
    ref = run_det(det, img, size)
    mine = O.superpoint_forward(sp_sd, {"image": img, "image_size": size}, conf)
    assert torch.equal(ref["keypoints"], mine["keypoints"])
    e = close(ref["descriptors"], mine["descriptors"], 1e-6, "sp desc")
    nms_ref = RSP.simple_nms(mine["keypoint_scores_dense"], 3)
    assert torch.equal(nms_ref, O.simple_nms(mine["keypoint_scores_dense"], 3))
    print("superpoint: keypoints exact, descriptors", e)
    np.savez_compressed(os.path.join(OUT, "lg_superpoint.npz"), seed=11, shape=np.array([2, 96, 128]), K=K,
                        keypoints=ref["keypoints"].numpy(),
                        descriptors=ref["descriptors"].numpy().astype(np.float32),
                        dense_scores=mine["keypoint_scores_dense"].numpy(), nms_scores=mine["nms_scores"].numpy())

    # RGB input path (superpoint.py:209-211)
    rgb = torch.rand(1, 3, 64, 96, generator=torch.Generator().manual_seed(12))
    det_s = type(det)({"max_num_keypoints": 32, "force_num_keypoints": True, "detection_threshold": 0.0,
                       "nms_radius": 3, "trainable": False}).eval()
    det_s.load_state_dict(sp_sd)
    r2 = det_s({"image": rgb})
    m2 = O.superpoint_forward(sp_sd, {"image": rgb}, dict(conf, max_num_keypoints=32))
    assert torch.equal(r2["keypoints"], m2["keypoints"])
    close(r2["descriptors"], m2["descriptors"], 1e-6, "sp rgb desc")
    np.savez_compressed(os.path.join(OUT, "lg_superpoint_rgb.npz"), seed=12, keypoints=r2["keypoints"].numpy(),
                        descriptors=r2["descriptors"].numpy())

    # ---- LightGlue on planted descriptors ---------------------------------------------------------------------
    kp0, d0, kp1, d1 = O.planted_descriptors(2, K, seed=21)
    resize = torch.tensor([[480, 640], [480, 640]])                  # (h, w) as the ZEB loaders give it
    data = {"keypoints0": kp0, "keypoints1": kp1, "descriptors0": d0, "descriptors1": d1,
            "resize0": resize, "resize1": resize}
    ref = lg(data)
    mine = O.lightglue_forward(lg_sd, data)
    assert torch.equal(ref["matches0"], mine["matches0"]) and torch.equal(ref["matches1"], mine["matches1"])
    e1 = close(ref["log_assignment"], mine["log_assignment"], 2e-4, "log_assignment")
    e2 = close(ref["matching_scores0"], mine["matching_scores0"], 1e-5, "mscores0")
    e3 = close(ref["ref_descriptors0"], mine["ref_descriptors0"], 2e-4, "ref_desc0")
    for a, b in zip(ref["matches"], mine["matches"]):
        assert torch.equal(a, b)
    nm = [int((m > -1).sum()) for m in ref["matches0"]]
    print("lightglue: matches exact", nm, "log_assignment", e1, "scores", e2, "desc", e3)
    np.savez_compressed(os.path.join(OUT, "lg_lightglue.npz"), seed=21, K=K, matches0=ref["matches0"].numpy(),
                        matches1=ref["matches1"].numpy(), matching_scores0=ref["matching_scores0"].numpy(),
                        matching_scores1=ref["matching_scores1"].numpy(),
                        log_assignment=ref["log_assignment"].numpy(),
                        ref_descriptors0=ref["ref_descriptors0"][:, 0].numpy(),
                        ref_descriptors1=ref["ref_descriptors1"][:, 0].numpy())

    # ---- detector + matcher end to end (reference modules chained as lightning.py:165-174 chains them) -------
    img0 = O.seeded_gray(2, 96, 128, 31)
    img1 = torch.roll(img0, shifts=(8, 16), dims=(2, 3)).contiguous()   # same texture shifted by whole cells
    rs = torch.tensor([[96, 128], [96, 128]])
    pred = {}
    pred.update({k + "0": v for k, v in run_det(det, img0, rs[:1, [1, 0]]).items()})
    pred.update({k + "1": v for k, v in run_det(det, img1, rs[:1, [1, 0]]).items()})
    pred.update(lg({**pred, "resize0": rs, "resize1": rs}))
    scale = torch.tensor([[1.5, 2.0], [1.0, 1.25]])
    mine = O.gim_lightglue_inference(sp_sd, lg_sd, {"image0": img0, "image1": img1, "resize0": rs, "resize1": rs,
                                                     "scale0": scale, "scale1": scale}, conf)
    assert torch.equal(pred["matches0"], mine["pred"]["matches0"])
    close(pred["matching_scores0"], mine["pred"]["matching_scores0"], 1e-5, "e2e scores")
    print("e2e: matches", [int((m > -1).sum()) for m in pred["matches0"]])
    np.savez_compressed(os.path.join(OUT, "lg_e2e.npz"), seeds=np.array([31]), shift=np.array([8, 16]), K=K, scale=scale.numpy(),
                        keypoints0=pred["keypoints0"].numpy(), keypoints1=pred["keypoints1"].numpy(),
                        matches0=pred["matches0"].numpy(), matching_scores0=pred["matching_scores0"].numpy(),
                        mkpts0_f=mine["mkpts0_f"].numpy(), mkpts1_f=mine["mkpts1_f"].numpy(),
                        m_bids=mine["m_bids"].numpy(), mconf=mine["mconf"].numpy())


if __name__ == "__main__":
    main()
