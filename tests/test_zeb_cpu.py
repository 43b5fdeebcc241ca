"""CPU: gim_amd/zeb.py (AUC scorer + dump I/O) against values computed by the reference's own analysis.py
on a sample of its shipped gim_loftr dumps (oracle/make_golden_zeb.py)."""
import json
import os

import numpy as np

from gim_amd import zeb


def test_auc_matches_reference_scorer(golden_dir):
    d = os.path.join(golden_dir, "zeb")
    exp = json.load(open(os.path.join(d, "expected.json")))
    per, mean = zeb.score_dir(d, "gim_loftr", "50h")
    assert set(per) == set(exp)
    for scene, e in exp.items():
        assert abs(per[scene][5.0] - 100 * e["auc5"]) < 1e-9
        assert abs(per[scene][10.0] - 100 * e["auc10"]) < 1e-9
        assert abs(per[scene][20.0] - 100 * e["auc20"]) < 1e-9
    assert abs(mean[5.0] - 100 * np.mean([e["auc5"] for e in exp.values()])) < 1e-9
    # nan / inf pose errors count as 180 degrees (analysis.py:37-40)
    a = zeb.error_auc([1.0, float("nan"), 2.0], [float("inf"), 1.0, 3.0])
    assert a[5.0] == zeb.error_auc([180.0, 180.0, 2.0], [180.0, 180.0, 3.0])[5.0]


def test_dump_roundtrip_and_row_format(tmp_path, golden_dir):
    src = os.path.join(golden_dir, "zeb", zeb.dump_path("", "gim_loftr", "KITTI", "50h").lstrip("/"))
    cols = zeb.read_dump(src)
    assert list(cols) == zeb.HEADER.split()
    lines = open(src).read().splitlines()[1:]
    out = zeb.dump_path(str(tmp_path), "gim_loftr", "KITTI", "50h")
    assert os.path.basename(out) == "[T] gim_loftr           KITTI 50h.txt"
    zeb.write_dump(out, lines)
    back = open(out).read().splitlines()
    assert back[0] == zeb.HEADER
    uniq = sorted({ln.split()[0]: ln for ln in reversed(lines)}.values(), key=lambda r: r.split()[0])
    assert [b.split()[0] for b in back[1:]] == [u.split()[0] for u in uniq]
    # row formatting of lightning.py:261-270
    epi = np.array([1e-4, 6e-4, 2e-4, 9e-4])
    row = zeb.format_row("s#a#b", 0.5, 0.25, 1.5, 2.5, 3.5, epi, [True, True, False, False])
    assert row == "s#a#b 0.5 0.25 1.5 2.5 3.5 0.5 2 0.5 1"
    assert zeb.format_row("s#a#b", 0, 0, np.inf, np.inf, np.inf, [], []).endswith("0.0 0 0.0 0")


def test_relative_pose_error():
    T = np.eye(4)
    T[:3, 3] = [1.0, 0, 0]
    c, s = np.cos(np.deg2rad(10)), np.sin(np.deg2rad(10))
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    t_err, R_err = zeb.relative_pose_error(T, R, np.array([-1.0, 0, 0]))
    assert abs(R_err - 10) < 1e-6 and abs(t_err) < 1e-6  # sign-ambiguous translation (min(e, 180-e))
