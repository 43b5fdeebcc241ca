"""The real-checkpoint hook (tools/real_weights_check.py).

  * with $GIM_WEIGHTS_DIR set (a directory holding the reference's gim_loftr_50h.ckpt): the real check -- the checkpoint loads strictly,
    the fp32 mode reproduces the oracle's match indices on the demo pair, the fp16 mode either stays finite below the stated flip
    bound or trips its range guard and reports bf16; skipped otherwise (no checkpoint ships with the reference, demo.py:385-400);
  * always (GPU): the tool's self-test -- a seeded checkpoint written in the reference's file format ({'state_dict': {'model.*'}})
    goes through the same code path, so the hook itself is known to work the day a checkpoint is there."""
import os

import pytest

pytestmark = pytest.mark.gpu


def _check(rep):
    lo = rep["gim_loftr"]
    assert lo["oracle_matches"] > 50, lo["oracle_matches"]
    m = lo["modes"]
    assert m["fp32"]["ran_as"] == "fp32" and m["fp32"]["flip_rate"] <= 0.002 and m["fp32"].get("max_abs_dmconf", 0.0) <= 1e-3, m["fp32"]
    for prec in ("fp16", "bf16"):
        assert m[prec]["finite"], m[prec]
        assert m[prec]["ran_as"] == prec or (prec == "fp16" and m[prec]["fp16_range_guard_tripped"] and m[prec]["ran_as"] == "bf16")
        assert m[prec]["flip_rate"] <= 0.05, (prec, m[prec])
    a = lo["activation_max"]
    assert a["overall"] > 0 and len(a["largest"]) >= 3
    if a["overall"] < 65504.0 / 4:      # comfortable head-room: the fp16 mode must not have fallen back
        assert not m["fp16"]["fp16_range_guard_tripped"]


def test_hook_selftest_on_a_seeded_checkpoint(tmp_path):
    import numpy as np
    from PIL import Image
    from tools import real_weights_check as R
    from tools import synth_loftr as S
    # the seeded weights only "see" the synthetic textures they were calibrated on: the pair is written as two PNG files
    c0, c1 = S.textured_pairs(1, 480, 640, seed=21, frac=0.6)
    paths = []
    for name, img in (("a.png", c0[0]), ("b.png", c1[0])):
        paths.append(str(tmp_path / name))
        Image.fromarray((img.permute(1, 2, 0).numpy() * 255).round().astype(np.uint8)).save(paths[-1])
    rep = R.main(["--synthetic", str(tmp_path / "w"), "--pair", *paths])
    assert rep is not None and os.path.exists(tmp_path / "w" / "gim_loftr_50h.ckpt")
    _check(rep)


@pytest.mark.skipif(not os.environ.get("GIM_WEIGHTS_DIR"), reason="GIM_WEIGHTS_DIR not set: no real checkpoint on this box")
def test_real_checkpoint():
    from tools import real_weights_check as R
    rep = R.main([])
    assert rep is not None and "gim_loftr" in rep, rep
    _check(rep)
