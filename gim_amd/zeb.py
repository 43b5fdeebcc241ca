"""ZEB scoring and dump I/O (host side, numpy) -- the "pose-AUC@5/10/20" half of the headline metric.

Restates, without Lightning:
  * `analysis.py:34-53` `error_auc` (pose AUC of max(R_err, t_err), nan/inf -> 180 deg) and the per-scene
    read / de-duplicate / score loop of `analysis.py:84-103`;
  * the dump format written by `trainer/lightning.py:258-275`
    ("identifiers covisible0 covisible1 R_errs t_errs t_errs2 Bef.Prec Bef.Num Aft.Prec Aft.Num",
    file name "[T] <weight> <scene:>15> <version>.txt"), so `check.py` / `analysis.py` of the reference keep
    working on dumps produced by this engine;
  * `tools/metrics.py:28-53,107-168` relative pose error and the RANSAC call -- RANSAC stays on the host as north_star
    prescribes: OpenCV when it imports (it does not in the build container), else gim_amd/pose.py (numpy five-point RANSAC +
    recoverPose, the published algorithms behind the two cv2 calls).
Pinned by `tests/test_zeb_cpu.py` against AUC values computed by the reference's own `analysis.py` on a
sample of its shipped dumps (`oracle/make_golden_zeb.py`).
"""
import os

import numpy as np

_trapz = getattr(np, "trapezoid", None) or np.trapz   # np.trapz is deprecated since numpy 2.0

DATASETS = ["GL3D", "BlendedMVS", "ETH3DI", "ETH3DO", "KITTI", "RobotcarWeather", "RobotcarSeason",
            "RobotcarNight", "Multi-FoV", "SceneNetRGBD", "ICL-NUIM", "GTA-SfM"]  # analysis.py:18-31
HEADER = "identifiers covisible0 covisible1 R_errs t_errs t_errs2 Bef.Prec Bef.Num Aft.Prec Aft.Num"


def error_auc(errs0, errs1, thresholds=(5.0, 10.0, 20.0)):
    """analysis.py:34-53 with numeric thresholds (degrees).  Returns {thr: auc in [0,1]}."""
    errs0 = np.array(errs0, dtype=np.float64)
    errs1 = np.array(errs1, dtype=np.float64)
    for e in (errs0, errs1):
        e[np.isnan(e)] = 180
        e[np.isinf(e)] = 180
    errors = np.max(np.stack([errs0, errs1]), axis=0)
    errors = [0] + sorted(list(errors))
    recall = list(np.linspace(0, 1, len(errors)))
    out = {}
    for thr in thresholds:
        thr = float(thr)
        last_index = np.searchsorted(errors, thr)
        y = recall[:last_index] + [recall[last_index - 1]]
        x = errors[:last_index] + [thr]
        out[thr] = float(_trapz(y, x) / thr)
    return out


def dump_path(directory, weight, scene, version):
    return os.path.join(directory, f"[T] {weight} {scene:>15} {version}.txt")  # lightning.py:273


def format_row(identifier, covisible0, covisible1, R_err, t_err, t_err2, epi_errs, inliers, epi_thr=5e-4):
    """One dump line (lightning.py:261-270).  epi_errs: float array [M]; inliers: bool array [M]."""
    epi = np.asarray(epi_errs)
    inl = np.asarray(inliers, dtype=bool) if len(epi) else np.zeros(0, dtype=bool)
    bef = epi < epi_thr
    aft = epi[inl] < epi_thr if len(epi) else np.zeros(0, dtype=bool)
    mean = lambda x: sum(x) / max(len(x), 1)  # noqa: E731
    return (f"{identifier} {covisible0} {covisible1} {R_err} {t_err} {t_err2} "
            f"{mean(bef)} {sum(bef)} {mean(aft)} {sum(aft)}")


def write_dump(path, rows):
    """rows: iterable of already formatted lines; written sorted by identifier, duplicates removed
    (lightning.py:253-255)."""
    uniq = {}
    for r in rows:
        uniq.setdefault(r.split()[0], r)
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w") as f:
        f.write(HEADER + "\n")
        for k in sorted(uniq):
            f.write(uniq[k] + "\n")


def read_dump(path):
    """-> dict of column -> list (first occurrence of each identifier only, analysis.py:95-101)."""
    with open(path) as f:
        lines = f.readlines()
    head = lines[0].split()
    cols = {k: [] for k in head}
    seen = set()
    for ln in lines[1:]:
        x = ln.split()
        if not x or x[0] in seen:
            continue
        seen.add(x[0])
        for k, v in zip(head, x):
            cols[k].append(v)
    return cols


def score_dir(directory, weight, version, thresholds=(5.0, 10.0, 20.0)):
    """AUC per scene and mean over the scenes present, in percent.  -> ({scene: {thr: auc%}}, {thr: mean%})"""
    files = {}
    for d in os.listdir(directory):
        stem = d.rpartition(".txt")[0].split()
        if len(stem) >= 4 and stem[1] == weight and stem[-1] == version:
            files[stem[2]] = d
    per = {}
    for scene in DATASETS:
        if scene not in files:
            continue
        c = read_dump(os.path.join(directory, files[scene]))
        auc = error_auc([float(v) for v in c["R_errs"]], [float(v) for v in c["t_errs"]], thresholds)
        per[scene] = {t: 100.0 * a for t, a in auc.items()}
    mean = {t: float(np.mean([per[s][t] for s in per])) for t in thresholds} if per else {}
    return per, mean


# ---- host-side pose metrics (tools/metrics.py) ---------------------------------------------------
def relative_pose_error(T_0to1, R, t, ignore_gt_t_thr=0.0):
    """tools/metrics.py:10-29: (t_err deg, R_err deg, t_err2) of an estimated pose against T_0to1."""
    t_gt = T_0to1[:3, 3]
    n = np.linalg.norm(t) * np.linalg.norm(t_gt)
    t_err = np.rad2deg(np.arccos(np.clip(np.dot(t, t_gt) / n, -1.0, 1.0)))
    t_err = np.minimum(t_err, 180 - t_err)
    if np.linalg.norm(t_gt) < ignore_gt_t_thr:
        t_err = 0
    r = np.linalg.norm(t_gt) / np.linalg.norm(t)
    t_err2 = np.linalg.norm((t * r - t_gt))
    R_gt = T_0to1[:3, :3]
    cos = np.clip((np.trace(np.dot(R.T, R_gt)) - 1) / 2, -1.0, 1.0)
    return t_err, np.rad2deg(np.abs(np.arccos(cos))), t_err2


def estimate_pose(kpts0, kpts1, K0, K1, thresh=0.5, conf=0.99999):
    """tools/metrics.py:77-103 (findEssentialMat RANSAC + recoverPose).  Host only.  OpenCV when it imports (the reference's own
    call), otherwise the numpy restatement of the same two OpenCV routines in gim_amd/pose.py (`GIM_POSE_BACKEND` forces one)."""
    from . import pose
    if len(kpts0) < 5:
        return None
    kpts0 = (kpts0 - K0[[0, 1], [2, 2]][None]) / K0[[0, 1], [0, 1]][None]
    kpts1 = (kpts1 - K1[[0, 1], [2, 2]][None]) / K1[[0, 1], [0, 1]][None]
    ransac_thr = thresh / np.mean([K0[0, 0], K1[1, 1], K0[0, 0], K1[1, 1]])
    if pose.backend() == "numpy":
        E, mask = pose.find_essential_mat(kpts0, kpts1, ransac_thr, prob=conf)
        if E is None:
            return None
        n, R, t, _ = pose.recover_pose(E, kpts0, kpts1, 1e9, mask=mask)
        return (R, t, mask) if n > 0 else None
    import cv2
    E, mask = cv2.findEssentialMat(kpts0, kpts1, np.eye(3), threshold=ransac_thr, prob=conf, method=cv2.RANSAC)
    if E is None:
        return None
    best, ret = 0, None
    for _E in np.split(E, len(E) / 3):
        n, R, t, _ = cv2.recoverPose(_E, kpts0, kpts1, np.eye(3), 1e9, mask=mask)
        if n > best:
            ret, best = (R, t[:, 0], mask.ravel() > 0), n
    return ret


def symmetric_epipolar_distance(pts0, pts1, E, K0, K1):
    """tools/metrics.py:32-52 (squared symmetric epipolar distance in normalised coordinates), numpy."""
    pts0 = (pts0 - K0[[0, 1], [2, 2]][None]) / K0[[0, 1], [0, 1]][None]
    pts1 = (pts1 - K1[[0, 1], [2, 2]][None]) / K1[[0, 1], [0, 1]][None]
    pts0 = np.concatenate([pts0, np.ones_like(pts0[:, :1])], 1)
    pts1 = np.concatenate([pts1, np.ones_like(pts1[:, :1])], 1)
    Ep0 = pts0 @ E.T
    p1Ep0 = np.sum(pts1 * Ep0, -1)
    Etp1 = pts1 @ E
    return p1Ep0 ** 2 * (1.0 / (Ep0[:, 0] ** 2 + Ep0[:, 1] ** 2) + 1.0 / (Etp1[:, 0] ** 2 + Etp1[:, 1] ** 2))


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def evaluate_batch(batch, estimate=None):
    """Per-pair dump rows of one matched batch = `Trainer.compute_metrics` (lightning.py:101-122) +
    `compute_symmetrical_epipolar_errors` / `compute_pose_errors` (tools/metrics.py:56-74,107-168).
    `batch` needs mkpts0_f, mkpts1_f, m_bids (the matcher's outputs) and K0, K1, T_0to1, scene_id, pair_names,
    covisible0, covisible1 (the ZEB loaders' fields).  `estimate(kpts0, kpts1, K0, K1)` defaults to the cv2
    RANSAC of the reference (thresh 0.5, conf 0.99999 -- the values tools/metrics.py:139 hard-codes)."""
    estimate = estimate or (lambda a, b, k0, k1: estimate_pose(a, b, k0, k1, 0.5, 0.99999))
    m_bids = _np(batch["m_bids"])
    pts0, pts1 = _np(batch["mkpts0_f"]).astype(np.float64), _np(batch["mkpts1_f"]).astype(np.float64)
    K0, K1, T = _np(batch["K0"]).astype(np.float64), _np(batch["K1"]).astype(np.float64), _np(batch["T_0to1"]).astype(np.float64)
    names = list(zip(batch["scene_id"], *batch["pair_names"]))
    rows = []
    for b in range(K0.shape[0]):
        sel = m_bids == b
        p0, p1 = pts0[sel], pts1[sel]
        t = T[b, :3, 3]
        Tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        epi = symmetric_epipolar_distance(p0, p1, Tx @ T[b, :3, :3], K0[b], K1[b]) if len(p0) else np.zeros(0)
        ret = estimate(p0, p1, K0[b], K1[b])
        if ret is None:
            R_err = t_err = t_err2 = np.inf
            inl = np.array([]).astype(bool)
        else:
            R, tt, inl = ret
            t_err, R_err, t_err2 = relative_pose_error(T[b], R, tt, ignore_gt_t_thr=0.0)
        rows.append(format_row("#".join(names[b]), _np(batch["covisible0"])[b], _np(batch["covisible1"])[b],
                               R_err, t_err, t_err2, epi, inl))
    return rows


def run_scene(matcher, batches, out_path, rank=0, world=1, estimate=None, skip_existing=True):
    """The ZEB test loop of `test.py:188-231` + `trainer/lightning.py:243-275` without Lightning.
    Every rank runs `matcher(batch)` (mutates the batch like LoFTR.forward) over ITS batches -- shard with
    `gim_amd.runner.shard_pairs` or a DistributedSampler -- and scores them on the host; the per-pair rows
    (~100 B each) are gathered once at the end and rank 0 writes the dump in the reference's format.
    Restartable like the reference: an existing dump is kept (`test.py:226-228`)."""
    if skip_existing and os.path.exists(out_path):
        return None
    rows = []
    for batch in batches:
        matcher(batch)
        rows.extend(evaluate_batch(batch, estimate))
    if world > 1:
        import torch.distributed as dist
        gathered = [None] * world
        dist.all_gather_object(gathered, rows)
        rows = [r for part in gathered for r in part]
    if rank == 0:
        write_dump(out_path, rows)
    return rows
