"""ZEB pair readers (host side): the `Dataset.__getitem__` of the reference's twelve ZEB scene classes without torch's
Dataset / Lightning DataModule around them, so that `gim_amd.zeb.run_scene` can stream real ZEB data into the engine.

Reference (file:line):
  * `datasets/utils.py:80-126` `read_images`: load RGB, grey copy, resize so that the longer edge is <= max_resize, round both edges
    down to a multiple of `df`, optional zero padding to a max_resize square (+ mask), scale = original / resized (w, h);
  * `datasets/{kitti,gl3d,eth3d,gtasfm,iclnuim,multifov,robotcar,scenenet}/*.py:28-125`: one pair per `zeb/<scene>/<id>-*.txt`,
    line = `name0 name1 covisible0 covisible1 K0(9) K1(9) T_0to1(16)`, images `<scene_id><sep><name>.png`; the classes differ only
    in the separator, in whether the name's extension is stripped and in the `pair_names` / `dataset_name` they report (table
    `SCENES` below; `datasets/data.py:21-34` maps the twelve ZEB names onto them, BlendedMVS re-using the GL3D class);
  * `datasets/data.py:150-160`: tests use max_resize = args.img_size, df = LOFTR.RESOLUTION[0] = 8, padding per config.

Decoding differs from the reference in one respect only: OpenCV is not available to this engine's environment, so images are
decoded with PIL and resized with torch (`F.interpolate(bilinear, align_corners=False)` = `cv2.resize`'s INTER_LINEAR sampling
grid, in float instead of OpenCV's 11-bit fixed point; grey = ITU-R 601 like `cv2.COLOR_RGB2GRAY`): pixel values can differ by
one 8-bit step.  Everything else -- sizes, scales, intrinsics, poses, identifiers -- is the reference's arithmetic.
"""
import glob
import os
from os.path import join

import numpy as np
import torch
import torch.nn.functional as F

# ZEB scene -> (separator between scene id and image name, strip the name's extension?, dataset_name, pair_names formatter)
_plain = lambda n: n  # noqa: E731
SCENES = {
    "GL3D":            ("_", True,  "GL3D",     _plain),
    "BlendedMVS":      ("_", True,  "GL3D",     _plain),                     # data.py:14,25: the GL3D class
    "ETH3DI":          ("-", True,  "ETH3D",    lambda n: n + ".JPG"),
    "ETH3DO":          ("-", True,  "ETH3D",    lambda n: n + ".JPG"),
    "KITTI":           ("-", True,  "KITTI",    lambda n: n + ".png"),
    "RobotcarWeather": ("_", False, "Robotcar", str),
    "RobotcarSeason":  ("_", False, "Robotcar", str),
    "RobotcarNight":   ("_", False, "Robotcar", str),
    "Multi-FoV":       ("-", False, "MultiFoV", lambda n: f"img/{n}.png"),
    "SceneNetRGBD":    ("-", False, "SceneNet", lambda n: n + ".jpg"),
    "ICL-NUIM":        ("-", False, "ICL-NUIM", lambda n: n + ".jpg"),
    "GTA-SfM":         ("-", False, "GTA-SfM",  _plain),
}


def get_resized_wh(w, h, resize=None):
    """datasets/utils.py:34-40"""
    if resize is not None:
        scale = resize / max(h, w)
        return int(round(w * scale)), int(round(h * scale))
    return w, h


def get_divisible_wh(w, h, df=None):
    """datasets/utils.py:43-54"""
    if df is not None:
        return max((w // df), 1) * df, max((h // df), 1) * df
    return w, h


def read_images(path, max_resize, df, padding, image=None):
    """datasets/utils.py:80-126 -> (gray [1,h,w], color [3,h,w], scale [2] = (w/w_new, h/h_new), resize [h_new, w_new], mask or None)"""
    assert max_resize is not None
    if image is None:
        from PIL import Image
        image = np.asarray(Image.open(path).convert("RGB"))
    t = torch.from_numpy(np.array(image)).permute(2, 0, 1).float()                       # [3,h,w], 0..255 (np.array: writable copy)
    g = (0.299 * t[0] + 0.587 * t[1] + 0.114 * t[2]).round()[None]                       # cv2.COLOR_RGB2GRAY
    w, h = image.shape[1], image.shape[0]
    w_new, h_new = get_resized_wh(w, h, max_resize) if max(w, h) > max_resize else (w, h)
    w_new, h_new = get_divisible_wh(w_new, h_new, df)
    if (w_new, h_new) != (w, h):
        t = F.interpolate(t[None], size=(h_new, w_new), mode="bilinear", align_corners=False)[0].round()
        g = F.interpolate(g[None], size=(h_new, w_new), mode="bilinear", align_corners=False)[0].round()
    scale = torch.tensor([w / w_new, h / h_new], dtype=torch.float)
    mask = None
    if padding:   # pad_bottom_right to a max_resize square (utils.py:57-75)
        assert max_resize >= max(h_new, w_new)
        tp = torch.zeros(3, max_resize, max_resize)
        gp = torch.zeros(1, max_resize, max_resize)
        tp[:, :h_new, :w_new] = t
        gp[:, :h_new, :w_new] = g
        mask = torch.zeros(max_resize, max_resize, dtype=torch.bool)
        mask[:h_new, :w_new] = True
        t, g = tp, gp
    return g / 255, t / 255, scale, [h_new, w_new], mask


class ZebScene:
    """One ZEB scene directory (`<root>/<scene>/`): `len()`, `scene[i]` -> the reference's data dict (one un-batched pair)."""

    def __init__(self, root, scene, max_resize, df=8, padding=False):
        if scene not in SCENES:
            raise KeyError(f"unknown ZEB scene {scene!r}; known: {sorted(SCENES)}")
        self.scene, self.dir = scene, join(root, scene)
        self.sep, self.strip, self.dataset_name, self.fmt = SCENES[scene]
        self.max_resize, self.df, self.padding = max_resize, df, padding
        lines = []
        for path in glob.glob(join(self.dir, "*.txt")):
            with open(path, "r") as f:
                scene_id = os.path.basename(path).rpartition(".")[0].split(self.sep)[0]
                lines.append([scene_id] + f.readline().strip().split())
        self.pairs = sorted(lines)

    def __len__(self):
        return len(self.pairs)

    def __getitem__(self, idx):
        pair = self.pairs[idx]
        scene_id = pair[0]
        n0, n1 = (pair[1].rpartition(".")[0], pair[2].rpartition(".")[0]) if self.strip else (pair[1], pair[2])
        p0 = join(self.dir, f"{scene_id}{self.sep}{n0}.png")
        p1 = join(self.dir, f"{scene_id}{self.sep}{n1}.png")
        from PIL import Image
        im0, im1 = np.asarray(Image.open(p0).convert("RGB")), np.asarray(Image.open(p1).convert("RGB"))
        image0, color0, scale0, resize0, mask0 = read_images(p0, self.max_resize, self.df, self.padding, im0)
        image1, color1, scale1, resize1, mask1 = read_images(p1, self.max_resize, self.df, self.padding, im1)
        f = lambda a, b: torch.tensor(list(map(float, pair[a:b])), dtype=torch.float)  # noqa: E731
        data = {
            "image0": image0, "color0": color0, "imsize0": torch.tensor(im0.shape[:2], dtype=torch.long),
            "resize0": torch.tensor(resize0, dtype=torch.long),
            "image1": image1, "color1": color1, "imsize1": torch.tensor(im1.shape[:2], dtype=torch.long),
            "resize1": torch.tensor(resize1, dtype=torch.long),
            "T_0to1": f(23, 39).reshape(4, 4), "K0": f(5, 14).reshape(3, 3), "K1": f(14, 23).reshape(3, 3),
            "scale0": scale0, "scale1": scale1, "dataset_name": self.dataset_name, "scene_id": scene_id,
            "pair_id": f"{idx}-{idx}", "pair_names": (self.fmt(n0), self.fmt(n1)),
            "covisible0": float(pair[3]), "covisible1": float(pair[4]),
        }
        if mask0 is not None:   # coarse-level padding masks (kitti.py:115-123): nearest down-scale by 1 / df
            m = F.interpolate(torch.stack([mask0, mask1], dim=0)[None].float(), scale_factor=1 / self.df, mode="nearest",
                              recompute_scale_factor=False)[0].bool()
            data.update({"mask0": m[0], "mask1": m[1]})
        return data

    @staticmethod
    def identifier(data):
        """the dump identifier of a pair (trainer/lightning.py:107-109: '#'.join of scene id and the two pair names)"""
        return "#".join([data["scene_id"], *data["pair_names"]])


def collate(items):
    """torch's default_collate semantics for the reference's data dict: tensors stacked, numbers -> tensors, strings -> lists,
    tuples of strings transposed (`pair_names` becomes ([name0 of every pair], [name1 of every pair]), which is what
    `compute_metrics` zips, lightning.py:107)"""
    out = {}
    for k in items[0]:
        v = [it[k] for it in items]
        if torch.is_tensor(v[0]):
            out[k] = torch.stack(v)
        elif isinstance(v[0], (int, float)):
            out[k] = torch.tensor(v, dtype=torch.float64 if isinstance(v[0], float) else torch.int64)
        elif isinstance(v[0], tuple):
            out[k] = [list(x) for x in zip(*v)]
        else:
            out[k] = v
    return out
