"""The reference's entry points as drop-ins (VERDICT r1 missing #4): `demo.py --model` (gim_amd/demo.py) and the hloc matcher
plugin (gim_amd/hloc_matchers/gim_dkm_hip.py).  CPU part: image pre-processing arithmetic, checkpoint key-prefix rules
(demo.py:355-395, hloc/matchers/dkm.py:27-38) on synthetic checkpoints, plugin lookup through `dynamic_load`."""
import os

import numpy as np
import pytest
import torch

import lightglue_oracle as LO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "demo")


def test_preprocess_follows_demo_py():
    from gim_amd import demo as D
    img = D.read_image(os.path.join(GOLD, "a1.png"))
    assert img.shape == (1004, 1004, 3) and img.dtype == np.uint8
    t, scale = D.preprocess(img)                                   # demo.py:171-177: size floored to a multiple of 8
    assert tuple(t.shape) == (3, 1000, 1000) and np.allclose(scale, [1.004, 1.004])
    assert 0.0 <= float(t.min()) and float(t.max()) <= 1.0
    g, s = D.preprocess(D.read_image(os.path.join(GOLD, "a1.png"), grayscale=True), grayscale=True, resize_max=512)
    assert tuple(g.shape) == (1, 512, 512) and np.allclose(s, [1004 / 512, 1004 / 512])
    gold = np.load(os.path.join(GOLD, "lightglue_a1_a2.npz"))
    assert tuple(gold["shape"]) == (512, 512) and np.allclose(gold["scale0"], s)


def test_checkpoint_prefix_rules(tmp_path):
    """a Lightning-style checkpoint ({'state_dict': {'model.*', 'superpoint.*'}}) loads into detector + matcher exactly as
    demo.py:377-395 splits it; a gim_loftr checkpoint with `model.` prefixes loads through LoFTR.load_state_dict"""
    from gim_amd import demo as D
    sp_sd, lg_sd = LO.make_state_dicts(0)
    ck = {"state_dict": {**{"superpoint." + k: v for k, v in sp_sd.items()}, **{"model." + k: v for k, v in lg_sd.items()}}}
    path = str(tmp_path / "gim_lightglue_100h.ckpt")
    torch.save(ck, path)
    model, detector = D.build("gim_lightglue", path, "fp32", device="cpu")
    for k, v in sp_sd.items():
        assert torch.equal(detector.state_dict()[k], v), k
    for k, v in lg_sd.items():
        assert torch.equal(model.state_dict()[k], v), k
    import loftr_oracle as O
    sd = O.make_state_dict(0)
    path = str(tmp_path / "gim_loftr_50h.ckpt")
    torch.save({"state_dict": {"model." + k: v for k, v in sd.items()}}, path)
    model, detector = D.build("gim_loftr", path, "fp32", device="cpu")
    assert detector is None
    for k, v in sd.items():
        assert torch.equal(model.state_dict()[k], v), k
    with pytest.raises(ValueError):
        D.build("gim_sift", None, None, device="cpu")


def test_hloc_plugin_lookup_and_checkpoint(tmp_path, monkeypatch):
    import dkm_oracle as DO
    import gim_amd.hloc_matchers as plugins
    from hloc.utils.base_model import BaseModel, dynamic_load
    cls = dynamic_load(plugins, "gim_dkm_hip")                      # hloc/utils/base_model.py:36-47
    assert issubclass(cls, BaseModel) and cls.required_inputs == ["image0", "image1"]
    sd = DO.make_state_dict(0)
    ck = {"state_dict": {**{"model." + k: v for k, v in sd.items()}, "model.encoder.net.fc.weight": torch.zeros(3, 3)}}
    (tmp_path / "weights").mkdir()
    torch.save(ck, str(tmp_path / "weights" / "gim_dkm_100h.ckpt"))
    monkeypatch.chdir(tmp_path)                                     # the plugin resolves conf['weights'] under weights/ (dkm.py:27)
    m = cls({"weights": "gim_dkm_100h.ckpt", "max_num_matches": 100, "precision": "fp32"})
    assert m.conf["max_num_matches"] == 100 and (m.h, m.w) == (672, 896)
    got = m.net.state_dict()
    for k, v in sd.items():
        assert torch.equal(got[k], v), k
    with pytest.raises(AssertionError):
        m({"image0": torch.zeros(1, 3, 8, 8)})                      # BaseModel.forward checks required_inputs
