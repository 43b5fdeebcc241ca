"""SURVEY 8(f1) end to end on the GPU with a REAL matcher: ZEB scene directory on disk -> `gim_amd.zeb_data.ZebScene` / `collate`
-> gim_loftr engine (bf16, fused kernels, HIP graph) -> `gim_amd.zeb.run_scene` -> dump file in the reference's format ->
`score_dir`.  The scene is synthetic but geometrically consistent: a fronto-parallel textured plane seen by two cameras that
differ by a translation parallel to the image plane, i.e. image1 = image0 shifted by the disparity -- so the ground-truth
essential matrix prices every match (symmetric epipolar distance, tools/metrics.py:56-74) and a translation-only estimator
(OpenCV's RANSAC is not installed on the box: `estimate=` hook of `evaluate_batch`) recovers the pose from the matches."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H, W, F_PX, DEPTH = 480, 640, 500.0, 4.0
SHIFT = (16, 24)  # (dy, dx) pixels: image1[y, x] = image0[y + dy, x + dx]  (tools/synth_loftr.textured_pairs)


def _write_scene(root, n_pairs):
    from PIL import Image
    from tools import synth_loftr as S
    d = os.path.join(root, "GL3D")
    os.makedirs(d)
    c0, c1 = S.textured_pairs(n_pairs, H, W, seed=77, shift=SHIFT, frac=1.0)
    K = np.array([[F_PX, 0, W / 2], [0, F_PX, H / 2], [0, 0, 1.0]])
    # a point seen at x0 in image 0 appears at x1 = x0 - (dx, dy) in image 1:  X1 = X0 + t,  t = -(dx, dy, 0) * DEPTH / f
    T = np.eye(4)
    T[:3, 3] = [-SHIFT[1] * DEPTH / F_PX, -SHIFT[0] * DEPTH / F_PX, 0.0]
    for k in range(n_pairs):
        sid = f"s{k:03d}"
        for name, img in (("a", c0[k]), ("b", c1[k])):
            Image.fromarray((img.permute(1, 2, 0).numpy() * 255).round().astype(np.uint8)).save(os.path.join(d, f"{sid}_{name}.png"))
        vals = ["a.jpg", "b.jpg", "0.9", "0.9"] + [repr(float(v)) for v in K.reshape(-1)] * 2 + [repr(float(v)) for v in T.reshape(-1)]
        with open(os.path.join(d, f"{sid}_0.txt"), "w") as f:
            f.write(" ".join(vals) + "\n")
    return T


def _translation_estimator(k0, k1, K0, K1):
    """R = I, t = direction of the mean normalised displacement (valid for this scene: planar, no rotation); all matches inliers"""
    if len(k0) < 8:
        return None
    n0 = (k0 - K0[[0, 1], [2, 2]]) / K0[[0, 1], [0, 1]]
    n1 = (k1 - K1[[0, 1], [2, 2]]) / K1[[0, 1], [0, 1]]
    t = np.append(np.median(n1 - n0, axis=0), 0.0)
    return np.eye(3), t / np.linalg.norm(t), np.ones(len(k0), dtype=bool)


def test_zeb_scene_to_dump_with_the_loftr_engine(tmp_path):
    from gim_amd import zeb
    from gim_amd.zeb_data import ZebScene, collate
    from tools import synth_loftr as S
    root = str(tmp_path / "zeb")
    n_pairs = 4
    _write_scene(root, n_pairs)
    scene = ZebScene(root, "GL3D", max_resize=640, df=8, padding=False)
    assert len(scene) == n_pairs
    model, _ = S.synthetic_model("bf16")
    model = model.to("cuda:0")
    counts = []

    def matcher(batch):
        for k, v in batch.items():
            if torch.is_tensor(v):
                batch[k] = v.to("cuda:0")
        model(batch)
        counts.append(int(batch["b_ids"].numel()))

    batches = [collate([scene[i], scene[i + 1]]) for i in range(0, n_pairs, 2)]
    out = zeb.dump_path(str(tmp_path / "dump"), "gim_loftr_hip", "GL3D", "test")
    rows = zeb.run_scene(matcher, batches, out, estimate=_translation_estimator)
    assert len(rows) == n_pairs and os.path.exists(out)
    assert all(c >= 2 * 500 for c in counts), counts            # match-rich pairs went through the fine level
    cols = zeb.read_dump(out)
    assert cols["identifiers"] == [f"s{k:03d}#a#b" for k in range(n_pairs)]
    R_err, t_err = [float(v) for v in cols["R_errs"]], [float(v) for v in cols["t_errs"]]
    assert max(R_err) < 1e-6 and max(t_err) < 1.0, (R_err, t_err)   # degrees: sub-pixel matches -> the translation direction to < 1 deg
    assert min(float(v) for v in cols["Bef.Prec"]) > 0.95, cols["Bef.Prec"]   # share of matches with symmetric epipolar distance < 5e-4
    assert min(float(v) for v in cols["Bef.Num"]) >= 500
    per, mean = zeb.score_dir(str(tmp_path / "dump"), "gim_loftr_hip", "test")
    assert mean[5.0] > 90.0, (per, mean)
    assert zeb.run_scene(matcher, batches, out, estimate=_translation_estimator) is None   # restartable: the dump is kept
