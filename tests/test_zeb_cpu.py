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
    t_err, R_err, t_err2 = zeb.relative_pose_error(T, R, np.array([-1.0, 0, 0]))
    assert abs(R_err - 10) < 1e-6 and abs(t_err) < 1e-6 and abs(t_err2 - 2.0) < 1e-9  # sign-ambiguous translation (min(e, 180-e))


# ---- the sharded evaluation loop (test.py + lightning.test_epoch_end), CPU, world_size 2 over gloo ----
def _synthetic_batches(pairs):
    """two cameras with a known relative pose looking at random 3-D points; `matcher` returns their projections"""
    import torch
    out = []
    for p in pairs:
        g = np.random.default_rng(p)
        K = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]])
        a = 0.1 + 0.01 * p
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        t = np.array([0.5, 0.05 * p, 0.1])
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
        X = np.concatenate([g.uniform(-1, 1, (60, 2)), g.uniform(4, 8, (60, 1))], 1)
        x0 = (K @ X.T).T; x0 = x0[:, :2] / x0[:, 2:]
        X1 = (R @ X.T).T + t
        x1 = (K @ X1.T).T; x1 = x1[:, :2] / x1[:, 2:]
        out.append({"scene_id": ["s"], "pair_names": (["%04d" % p], ["%04d" % (p + 1)]), "K0": torch.tensor(K)[None],
                    "K1": torch.tensor(K)[None], "T_0to1": torch.tensor(T)[None], "covisible0": torch.tensor([0.5]),
                    "covisible1": torch.tensor([0.25]), "_x0": x0, "_x1": x1, "_T": T})
    return out


def _matcher(batch):
    import torch
    batch.update({"mkpts0_f": torch.tensor(batch["_x0"]), "mkpts1_f": torch.tensor(batch["_x1"]),
                  "m_bids": torch.zeros(len(batch["_x0"]), dtype=torch.int64)})


def _oracle_pose(batch_holder):
    def est(p0, p1, K0, K1):  # stands in for cv2 RANSAC (not installed here): returns the ground-truth pose
        T = batch_holder["T"]
        return T[:3, :3], T[:3, 3] * 3.0, np.ones(len(p0), dtype=bool)
    return est


def _zeb_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gim_amd.runner import shard_pairs
    mine = shard_pairs(5, rank, world)
    batches = _synthetic_batches(mine)
    holder = {}

    def matcher(b):
        _matcher(b); holder["T"] = b["_T"]
    rows = zeb.run_scene(matcher, batches, out, rank, world, estimate=_oracle_pose(holder))
    assert len(rows) == 5
    dist.barrier(); dist.destroy_process_group()


def test_run_scene_world2(tmp_path):
    import torch.multiprocessing as mp
    out = zeb.dump_path(str(tmp_path), "gim_loftr", "GL3D", "test")
    mp.spawn(_zeb_worker, args=(2, 29900 + os.getpid() % 1000, out), nprocs=2, join=True)
    cols = zeb.read_dump(out)
    assert cols["identifiers"] == ["s#%04d#%04d" % (p, p + 1) for p in range(5)]      # sorted, no duplicates
    assert all(float(r) < 1e-4 for r in cols["R_errs"]) and all(float(t) < 1e-4 for t in cols["t_errs"])
    assert all(float(x) == 1.0 for x in cols["Bef.Prec"]) and cols["Bef.Num"] == ["60"] * 5  # exact projections: epi err ~ 0
    per, mean = zeb.score_dir(str(tmp_path), "gim_loftr", "test")
    assert per["GL3D"][5.0] > 99.9
    assert zeb.run_scene(None, [], out) is None  # restartable: an existing dump is kept


def test_estimate_pose_runs_opencv_ransac_where_available():
    """`zeb.estimate_pose` (= tools/metrics.py:77-103: cv2.findEssentialMat RANSAC + recoverPose) executed for real
    wherever OpenCV is installed (skipped otherwise -- not in this image): a synthetic two-view scene with a known relative pose,
    20 % gross outliers; the recovered rotation / translation direction must be within a degree."""
    import pytest
    pytest.importorskip("cv2")
    import numpy as np
    from gim_amd import zeb
    rng = np.random.default_rng(0)
    K0 = np.array([[525.0, 0, 320], [0, 525.0, 240], [0, 0, 1]])
    K1 = np.array([[500.0, 0, 310], [0, 500.0, 250], [0, 0, 1]])
    ang = np.deg2rad(8.0)
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t = np.array([0.4, -0.05, 0.1])
    X = np.concatenate([rng.uniform(-2, 2, (400, 2)), rng.uniform(4, 9, (400, 1))], 1)
    p0 = (K0 @ X.T).T
    p1 = (K1 @ (X @ R.T + t).T).T
    k0, k1 = p0[:, :2] / p0[:, 2:], p1[:, :2] / p1[:, 2:]
    k1[:80] = rng.uniform(0, 480, (80, 2))                                      # outliers
    ret = zeb.estimate_pose(k0, k1, K0, K1, 0.5, 0.99999)
    assert ret is not None
    Re, te, inl = ret
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    t_err, R_err, _ = zeb.relative_pose_error(T, Re, te)
    assert R_err < 1.0 and t_err < 1.0, (R_err, t_err)
    assert inl[80:].mean() > 0.9 and inl[:80].mean() < 0.2
    assert zeb.estimate_pose(k0[:4], k1[:4], K0, K1) is None                    # < 5 matches (tools/metrics.py:78-79)
