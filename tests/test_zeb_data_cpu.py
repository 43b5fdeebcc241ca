"""ZEB pair readers (gim_amd/zeb_data.py) on a synthetic scene directory laid out like `zeb/<scene>/` (datasets/kitti/kitti.py:28-125
and siblings): file discovery, per-scene naming rules, the `read_images` size / scale / padding arithmetic (datasets/utils.py:80-126),
intrinsics / pose parsing, identifiers, and the readers feeding `zeb.run_scene`."""
import os

import numpy as np
import pytest
import torch


def _write_scene(root, scene, sep, names, size=(50, 70)):
    from PIL import Image
    d = os.path.join(root, scene)
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(0)
    for k, (sid, n0, n1, stem0, stem1) in enumerate(names):
        for stem in (stem0, stem1):
            Image.fromarray(rng.integers(0, 255, (size[0], size[1], 3), dtype=np.uint8)).save(os.path.join(d, f"{sid}{sep}{stem}.png"))
        K0 = np.array([[100.0, 0, 35], [0, 100, 25], [0, 0, 1]]) + k
        K1 = K0 + 0.5
        T = np.eye(4)
        T[:3, 3] = [0.1 * (k + 1), 0.2, 0.3]
        vals = [n0, n1, 0.5, 0.25] + K0.ravel().tolist() + K1.ravel().tolist() + T.ravel().tolist()
        with open(os.path.join(d, f"{sid}{sep}{k}.txt"), "w") as f:
            f.write(" ".join(str(v) for v in vals) + "\n")


def test_read_images_arithmetic(tmp_path):
    from gim_amd.zeb_data import get_divisible_wh, get_resized_wh, read_images
    assert get_resized_wh(1226, 370, 640) == (640, 193) and get_divisible_wh(640, 193, 8) == (640, 192)
    assert get_divisible_wh(5, 5, 8) == (8, 8)                                  # max(w // df, 1) * df
    img = np.random.default_rng(1).integers(0, 255, (370, 1226, 3), dtype=np.uint8)
    gray, color, scale, resize, mask = read_images(None, 640, 8, False, img)
    assert gray.shape == (1, 192, 640) and color.shape == (3, 192, 640) and resize == [192, 640] and mask is None
    assert torch.allclose(scale, torch.tensor([1226 / 640, 370 / 192]))
    assert 0 <= gray.min() and gray.max() <= 1 and abs(float(gray.mean()) - float(color.mean())) < 0.02
    gray, color, scale, resize, mask = read_images(None, 640, 8, True, img)      # zero padding to a square + mask
    assert gray.shape == (1, 640, 640) and mask.shape == (640, 640) and int(mask.sum()) == 192 * 640
    assert float(color[:, 192:].abs().max()) == 0.0
    small = img[:100, :120]                                                      # smaller than max_resize: only the df rounding
    g2, _, s2, r2, _ = read_images(None, 640, 8, False, small)
    assert r2 == [96, 120] and torch.allclose(s2, torch.tensor([1.0, 100 / 96]))
    g3, c3, _, _, _ = read_images(None, 640, 8, False, np.ascontiguousarray(img[:192, :640]))   # no resize at all: bit-exact pixels
    assert torch.equal((c3 * 255).round().to(torch.uint8), torch.from_numpy(img[:192, :640]).permute(2, 0, 1))


@pytest.mark.parametrize("scene,sep,n0,stem0,expect_name", [
    ("KITTI", "-", "000012.png", "000012", "000012.png"),          # extension stripped, '.png' re-attached
    ("GL3D", "_", "00000007.jpg", "00000007", "00000007"),          # extension stripped, plain names
    ("RobotcarNight", "_", "1418236221", "1418236221", "1418236221"),
    ("Multi-FoV", "-", "img0042", "img0042", "img/img0042.png"),
    ("ETH3DO", "-", "DSC_0001.JPG", "DSC_0001", "DSC_0001.JPG"),
])
def test_scene_naming_rules(tmp_path, scene, sep, n0, stem0, expect_name):
    from gim_amd.zeb_data import ZebScene
    _write_scene(str(tmp_path), scene, sep, [("s1", n0, n0.replace("1", "2", 1), stem0, stem0.replace("1", "2", 1))])
    ds = ZebScene(str(tmp_path), scene, max_resize=64, df=8, padding=True)
    assert len(ds) == 1
    d = ds[0]
    assert d["pair_names"][0] == expect_name and d["scene_id"] == "s1"
    assert d["image0"].shape == (1, 64, 64) and d["color0"].shape == (3, 64, 64) and d["mask0"].shape == (8, 8)
    assert d["imsize0"].tolist() == [50, 70] and d["resize0"].tolist() == [40, 64]   # 70 -> 64, 50 -> round(45.7) = 46 -> 40
    assert torch.allclose(d["scale0"], torch.tensor([70 / 64, 50 / 40]))
    assert d["K0"].tolist() == [[100.0, 0, 35], [0, 100, 25], [0, 0, 1]] and float(d["T_0to1"][0, 3]) == pytest.approx(0.1)
    assert d["covisible0"] == 0.5 and d["covisible1"] == 0.25
    assert ds.identifier(d) == "#".join(["s1", *d["pair_names"]])
    assert bool(d["mask0"][:5].all()) and not bool(d["mask0"][5:].any())          # 40 of 64 rows valid -> 5 of 8 coarse rows


def test_readers_feed_run_scene(tmp_path):
    """ZebScene -> collate -> zeb.run_scene with a stand-in matcher -> a dump the reference's analysis.py format reads"""
    from gim_amd import zeb
    from gim_amd.zeb_data import ZebScene, collate
    _write_scene(str(tmp_path), "KITTI", "-", [("a", "1.png", "2.png", "1", "2"), ("b", "3.png", "4.png", "3", "4")], size=(48, 64))
    ds = ZebScene(str(tmp_path), "KITTI", max_resize=64, df=8)

    def matcher(batch):   # contract of the real matchers: adds mkpts*_f, m_bids, mconf
        n = 12
        g = torch.Generator().manual_seed(0)
        batch.update({"mkpts0_f": torch.rand(n, 2, generator=g) * 40, "mkpts1_f": torch.rand(n, 2, generator=g) * 40,
                      "m_bids": torch.zeros(n, dtype=torch.long), "mconf": torch.rand(n, generator=g)})

    out = str(tmp_path / "dump" / "x.txt")
    rows = zeb.run_scene(matcher, [collate([ds[i]]) for i in range(len(ds))], out,
                         estimate=lambda k0, k1, K0, K1: None)     # no OpenCV here: the pose leg reports failure
    assert len(rows) == 2 and os.path.exists(out)
    cols = zeb.read_dump(out)
    assert cols["identifiers"] == ["a#1.png#2.png", "b#3.png#4.png"]
