"""CPU: oracle/lightglue_oracle.py replayed against the vectors oracle/make_golden_lightglue.py recorded from the
reference's own SuperPoint / LightGlue modules (networks/lightglue/superpoint.py, models/matchers/lightglue.py)."""
import os

import numpy as np
import torch

import lightglue_oracle as O


def _close(a, b, tol=1e-5):
    a, b = torch.as_tensor(np.asarray(a)), torch.as_tensor(np.asarray(b))
    assert a.shape == b.shape
    scale = max(1.0, b.abs().max().item())
    assert (a - b).abs().max().item() <= tol * scale


def test_param_specs_match_reference_counts():
    sp, lg = O.make_state_dicts(0)
    assert len(sp) == 24 and sum(v.numel() for v in sp.values()) == 1300865      # reference SuperPoint
    assert len(lg) == 251 and sum(v.numel() for v in lg.values()) == 11851601    # reference LightGlue (9 layers)


def test_superpoint_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "lg_superpoint.npz"))
    sp, _ = O.make_state_dicts(0)
    b, h, w = g["shape"]
    img = O.seeded_gray(int(b), int(h), int(w), int(g["seed"]))
    with torch.no_grad():
        out = O.superpoint_forward(sp, {"image": img}, dict(O.SP_CONF, max_num_keypoints=int(g["K"])))
    assert np.array_equal(out["keypoints"].numpy(), g["keypoints"])
    _close(out["descriptors"], g["descriptors"])
    _close(out["keypoint_scores_dense"], g["dense_scores"], 1e-6)
    assert np.array_equal(out["nms_scores"].numpy() > 0, g["nms_scores"] > 0)
    # border rule: nothing survives within 4 px of the canvas edge (superpoint.py:247-258)
    kp = out["keypoints"] - 0.5
    assert kp[..., 0].min() >= 4 and kp[..., 0].max() < w - 4 and kp[..., 1].min() >= 4 and kp[..., 1].max() < h - 4


def test_superpoint_rgb_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "lg_superpoint_rgb.npz"))
    sp, _ = O.make_state_dicts(0)
    rgb = torch.rand(1, 3, 64, 96, generator=torch.Generator().manual_seed(int(g["seed"])))
    with torch.no_grad():
        out = O.superpoint_forward(sp, {"image": rgb}, dict(O.SP_CONF, max_num_keypoints=32))
    assert np.array_equal(out["keypoints"].numpy(), g["keypoints"])
    _close(out["descriptors"], g["descriptors"])


def test_lightglue_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "lg_lightglue.npz"))
    _, lg = O.make_state_dicts(0)
    kp0, d0, kp1, d1 = O.planted_descriptors(2, int(g["K"]), seed=int(g["seed"]))
    rs = torch.tensor([[480, 640], [480, 640]])
    with torch.no_grad():
        out = O.lightglue_forward(lg, {"keypoints0": kp0, "keypoints1": kp1, "descriptors0": d0, "descriptors1": d1,
                                       "resize0": rs, "resize1": rs})
    assert np.array_equal(out["matches0"].numpy(), g["matches0"]) and np.array_equal(out["matches1"].numpy(), g["matches1"])
    assert (g["matches0"] > -1).sum() > 150                       # match-rich: the arithmetic is exercised
    # CPU GEMM/SDPA summation order depends on the thread count, and exp() of O(100) logits amplifies it
    _close(out["matching_scores0"], g["matching_scores0"], 1e-3)
    _close(out["matching_scores1"], g["matching_scores1"], 1e-3)
    _close(out["log_assignment"], g["log_assignment"], 1e-4)
    _close(out["ref_descriptors0"][:, 0], g["ref_descriptors0"], 1e-4)
    _close(out["ref_descriptors1"][:, 0], g["ref_descriptors1"], 1e-4)
    # structure of filter_matches (lightglue.py:284-300): matches are mutual and above the 0.1 threshold
    m0, m1 = out["matches0"], out["matches1"]
    for b in range(2):
        i = torch.where(m0[b] > -1)[0]
        assert torch.equal(m1[b][m0[b][i]], i) and (out["matching_scores0"][b][i] > 0.1).all()


def test_e2e_adapter_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "lg_e2e.npz"))
    sp, lg = O.make_state_dicts(0)
    img0 = O.seeded_gray(2, 96, 128, int(g["seeds"][0]))
    img1 = torch.roll(img0, shifts=tuple(int(x) for x in g["shift"]), dims=(2, 3)).contiguous()
    rs = torch.tensor([[96, 128], [96, 128]])
    scale = torch.as_tensor(g["scale"])
    with torch.no_grad():
        out = O.gim_lightglue_inference(sp, lg, {"image0": img0, "image1": img1, "resize0": rs, "resize1": rs,
                                                  "scale0": scale, "scale1": scale},
                                        dict(O.SP_CONF, max_num_keypoints=int(g["K"])))
    assert np.array_equal(out["pred"]["keypoints0"].numpy(), g["keypoints0"])
    assert np.array_equal(out["pred"]["keypoints1"].numpy(), g["keypoints1"])
    assert np.array_equal(out["pred"]["matches0"].numpy(), g["matches0"])
    _close(out["pred"]["matching_scores0"], g["matching_scores0"], 1e-3)
    _close(out["mkpts0_f"], g["mkpts0_f"]); _close(out["mkpts1_f"], g["mkpts1_f"])
    assert np.array_equal(out["m_bids"].numpy(), g["m_bids"])
    # adapter semantics (lightning.py:176-183): matched keypoints scaled per pair
    for b in range(2):
        i = torch.where(out["pred"]["matches0"][b] > -1)[0]
        exp = out["pred"]["keypoints0"][b][i] * scale[b]
        assert torch.equal(out["mkpts0_f"][out["m_bids"] == b], exp)
