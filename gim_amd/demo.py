"""`demo.py --model {gim_dkm, gim_roma, gim_loftr, gim_lightglue}` of the reference, on the HIP engine.

    python -m gim_amd.demo --model gim_lightglue [--weights weights/gim_lightglue_100h.ckpt] \
           [--image0 assets/demo/a1.png --image1 assets/demo/a2.png] [--precision bf16|fp32] [--resize-max N]

Mirrors the reference entry point step by step (reference file:line):
  * `read_image`         demo.py:123-137  (PIL instead of OpenCV: RGB order, ITU-R 601 grey)
  * `preprocess`         demo.py:140-178  (optional area down-scale to `resize_max`, /255, size made divisible by 8 with the
                                            bilinear tensor resize torchvision's `F.resize` performs, scale = original / new)
  * `build`              demo.py:324-400  (constructor arguments per model, checkpoint unwrapping and the per-model key-prefix
                                            rules; without a checkpoint the seeded init of the modules is kept -- plumbing runs)
  * `match_pair`         demo.py:405-511  (dense: pad / match / sample 5000 / pixels / un-pad / in-bounds mask; gim_loftr: the
                                            module's dict; gim_lightglue: detector x2 + matcher + per-pair gather)
  * robust fitting       demo.py:514-517  cv2.findFundamentalMat(USAC_MAGSAC, 1.0 px, 0.999999, 10000) stays on the host (north_star);
                                            without OpenCV: seven-point RANSAC of gim_amd/pose.py with the same parameters
Returns / prints `{mkpts0_f, mkpts1_f, mconf, m_bids}` in pixels of the ORIGINAL images (kpts * scale, demo.py:499-500 and
the `scale0/scale1` the other models get through their own adapters).
"""
import argparse
import os
from os.path import join

import numpy as np
import torch
import torch.nn.functional as F

MODELS = ("gim_dkm", "gim_roma", "gim_loftr", "gim_lightglue")
CKPT = {"gim_dkm": "gim_dkm_100h.ckpt", "gim_roma": "gim_roma_100h.ckpt", "gim_loftr": "gim_loftr_50h.ckpt",
        "gim_lightglue": "gim_lightglue_100h.ckpt"}   # demo.py:328-347


def read_image(path, grayscale=False):
    """demo.py:123-137 -> uint8 [H,W,3] RGB or [H,W] grey"""
    from PIL import Image
    im = Image.open(path)
    return np.asarray(im.convert("L" if grayscale else "RGB"))


def preprocess(image, grayscale=False, resize_max=None, dfactor=8, antialias=False):
    """demo.py:140-178 -> (float tensor [C,H',W'] in [0,1], scale (x, y) = original size / new size).
    Numerics vs the reference when a resize actually happens: (1) the divisible-by-8 resize is torchvision's `F.resize` on a
    TENSOR = bilinear, align_corners=False; the reference pins torchvision 0.13.1 (environment.yaml), whose tensor path does
    NOT antialias (`antialias=None`), hence the default here -- pass `antialias=True` to reproduce a torchvision >= 0.17
    installation, where it became the default; (2) the `resize_max` downscale is cv2 INTER_AREA in the reference: `mode='area'`
    (adaptive average pooling) equals it for integer factors only, non-integer factors differ in the fractional cell weights;
    (3) `read_image` decodes with PIL instead of OpenCV (identical for PNG, +-1 grey level for JPEG)."""
    image = image.astype(np.float32, copy=False)
    size = image.shape[:2][::-1]
    t = torch.from_numpy(image[None] if grayscale else image.transpose(2, 0, 1)).float()
    if resize_max:
        s = resize_max / max(size)
        if s < 1.0:   # resize_image(..., 'cv2_area'): area averaging (= INTER_AREA for integer factors)
            size_new = tuple(int(round(x * s)) for x in size)
            t = F.interpolate(t[None], size=size_new[::-1], mode="area")[0]
    t = t / 255.0
    size_new = tuple(int(x // dfactor * dfactor) for x in t.shape[-2:])
    t = F.interpolate(t[None], size=size_new, mode="bilinear", align_corners=False, antialias=antialias)[0]   # F.resize on a tensor
    scale = np.array(size) / np.array(size_new)[::-1]
    return t, scale


def _load_ckpt(path):
    sd = torch.load(path, map_location="cpu")
    return sd["state_dict"] if "state_dict" in sd.keys() else sd


def build(model_name, weights=None, precision=None, device="cuda", dinov2_weights=None):
    """demo.py:324-400 -> (model, detector or None) on `device`, eval mode"""
    kw = {"precision": precision} if precision else {}
    detector = None
    if model_name == "gim_dkm":
        from .dkm import DKMv3
        model = DKMv3(weights=None, h=672, w=896, **kw)
        if weights:
            sd = _load_ckpt(weights)
            for k in list(sd.keys()):   # demo.py:357-362: strip `model.`, drop the ResNet's unused fc head (under either name)
                v = sd.pop(k)
                nk = k.replace("model.", "", 1) if k.startswith("model.") else k
                if "encoder.net.fc" not in nk:
                    sd[nk] = v
            model.load_state_dict(sd)
    elif model_name == "gim_roma":
        from .roma import RoMa, random_dinov2_weights
        sd = _load_ckpt(weights) if weights else None
        # the reference downloads dinov2_vitl14_pretrain.pth inside the constructor (roma.py:591-595) and hides the ViT from
        # state_dict() (roma.py:612), so the gim checkpoint does not carry it: pass the same file here (no network access in
        # this engine).  Without any checkpoint a seeded synthetic ViT-L/14 stands in (plumbing runs).
        if dinov2_weights is not None:
            dsd = torch.load(dinov2_weights, map_location="cpu") if isinstance(dinov2_weights, str) else dinov2_weights
        elif sd is None:
            dsd = random_dinov2_weights("cpu")
        else:
            raise ValueError("gim_roma with a checkpoint also needs --dinov2-weights dinov2_vitl14_pretrain.pth "
                             "(the reference fetches it from the hub inside RoMa(), roma.py:591-595)")
        model = RoMa(img_size=[672], dinov2_weights=dsd, **kw)
        if sd is not None:
            for k in list(sd.keys()):
                if k.startswith("model."):
                    sd[k.replace("model.", "", 1)] = sd.pop(k)
            model.load_state_dict(sd)
    elif model_name == "gim_loftr":
        from .loftr import LoFTR, get_cfg_defaults, lower_config
        cfg = lower_config(get_cfg_defaults())["loftr"]
        if precision:
            cfg["precision"] = precision
        model = LoFTR(cfg)
        if weights:
            model.load_state_dict(_load_ckpt(weights))
    elif model_name == "gim_lightglue":
        from .lightglue import LightGlue, SuperPoint
        detector = SuperPoint({"max_num_keypoints": 2048, "force_num_keypoints": True, "detection_threshold": 0.0,
                               "nms_radius": 3, "trainable": False, **kw})
        model = LightGlue({"filter_threshold": 0.1, "flash": False, "checkpointed": True, **kw})
        if weights:
            sd = _load_ckpt(weights)
            for k in list(sd.keys()):
                if k.startswith("model."):
                    sd.pop(k)
                elif k.startswith("superpoint."):
                    sd[k.replace("superpoint.", "", 1)] = sd.pop(k)
            detector.load_state_dict(sd)
            sd = _load_ckpt(weights)
            for k in list(sd.keys()):
                if k.startswith("superpoint."):
                    sd.pop(k)
                elif k.startswith("model."):
                    sd[k.replace("model.", "", 1)] = sd.pop(k)
            model.load_state_dict(sd)
    else:
        raise ValueError(f"--model must be one of {MODELS}, got {model_name!r}")
    if detector is not None:
        detector = detector.eval().to(device)
    return model.eval().to(device), detector


@torch.no_grad()
def match_pair(model_name, model, detector, path0, path1, device="cuda", resize_max=None, num=5000):
    """demo.py:405-511 -> dict(mkpts0_f, mkpts1_f, mconf, m_bids, hw0_i, hw1_i); coordinates in pixels of the pre-processed images
    for the dense matchers and gim_loftr (as the reference leaves them), of the original images for gim_lightglue (:499-500)"""
    image0, scale0 = preprocess(read_image(path0), resize_max=resize_max)
    image1, scale1 = preprocess(read_image(path1), resize_max=resize_max)
    image0, image1 = image0.to(device)[None], image1.to(device)[None]
    data = dict(color0=image0, color1=image1, image0=image0, image1=image1)
    if model_name in ("gim_dkm", "gim_roma"):
        from .adapters import dense_demo_inference
        h, w = (672, 896) if model_name == "gim_dkm" else (672, 672)   # demo.py:421-424 (named width, height there)
        kpts0, kpts1, b_ids, mconf = dense_demo_inference(model, image0, image1, h, w, num)
    elif model_name == "gim_loftr":
        model(data)
        kpts0, kpts1, b_ids, mconf = data["mkpts0_f"], data["mkpts1_f"], data["m_bids"], data["mconf"]
    else:
        from .lightglue import gim_lightglue_inference
        gray0 = preprocess(read_image(path0, grayscale=True), grayscale=True, resize_max=resize_max)[0].to(device)[None]
        gray1 = preprocess(read_image(path1, grayscale=True), grayscale=True, resize_max=resize_max)[0].to(device)[None]
        d = dict(image0=gray0, image1=gray1, color0=image0, color1=image1,
                 scale0=torch.tensor(scale0, dtype=torch.float32, device=device)[None],
                 scale1=torch.tensor(scale1, dtype=torch.float32, device=device)[None],
                 image_size0=torch.tensor(gray0.shape[-2:][::-1], device=device)[None],
                 image_size1=torch.tensor(gray1.shape[-2:][::-1], device=device)[None])
        gim_lightglue_inference(detector, model, d)
        kpts0, kpts1, b_ids, mconf = d["mkpts0_f"], d["mkpts1_f"], d["m_bids"], d["mconf"]
    out = {"mkpts0_f": kpts0, "mkpts1_f": kpts1, "m_bids": b_ids, "mconf": mconf,
           "hw0_i": image0.shape[2:], "hw1_i": image1.shape[2:], "scale0": scale0, "scale1": scale1}
    # robust fitting on the host (demo.py:514-517): OpenCV's USAC_MAGSAC when cv2 imports, else plain seven-point RANSAC with the
    # same threshold / confidence / iteration bound (gim_amd/pose.py; `backend` says which one produced the mask)
    if len(kpts0) >= 8:
        from . import pose
        p0, p1 = kpts0.float().cpu().numpy(), kpts1.float().cpu().numpy()
        out["ransac_backend"] = pose.backend()
        if out["ransac_backend"] == "cv2":
            import cv2
            _, mask = cv2.findFundamentalMat(p0, p1, cv2.USAC_MAGSAC, ransacReprojThreshold=1.0, confidence=0.999999, maxIters=10000)
            out["inliers"] = mask.ravel() > 0 if mask is not None else np.zeros(len(p0), dtype=bool)
        else:
            _, mask = pose.find_fundamental_mat(p0, p1, threshold=1.0, prob=0.999999, max_iters=10000)
            out["inliers"] = mask
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--model", choices=MODELS, default="gim_dkm")
    ap.add_argument("--weights", default=None, help="checkpoint path (default: weights/<the reference's file name> if it exists)")
    ap.add_argument("--image0", default=join("assets", "demo", "a1.png"))
    ap.add_argument("--image1", default=join("assets", "demo", "a2.png"))
    ap.add_argument("--precision", choices=["bf16", "fp32"], default=None)
    ap.add_argument("--resize-max", type=int, default=None)
    ap.add_argument("--dinov2-weights", default=None, help="gim_roma: the DINOv2 ViT-L/14 file the reference downloads")
    args = ap.parse_args(argv)
    weights = args.weights
    if weights is None and os.path.exists(join("weights", CKPT[args.model])):
        weights = join("weights", CKPT[args.model])
    if weights is None:
        print(f"gim_amd.demo: no checkpoint ({join('weights', CKPT[args.model])} not found): running on the modules' seeded init")
    model, detector = build(args.model, weights, args.precision, dinov2_weights=args.dinov2_weights)
    out = match_pair(args.model, model, detector, args.image0, args.image1, resize_max=args.resize_max)
    n = len(out["mconf"])
    print(f"{args.model}: {n} matches" + (f", {int(out['inliers'].sum())} inliers ({out['ransac_backend']} RANSAC)" if "inliers" in out else " (fewer than 8: no RANSAC)"))
    for k in range(min(n, 5)):
        a, b = out["mkpts0_f"][k].tolist(), out["mkpts1_f"][k].tolist()
        print(f"  ({a[0]:8.2f}, {a[1]:8.2f}) <-> ({b[0]:8.2f}, {b[1]:8.2f})  conf {float(out['mconf'][k]):.4f}")
    return out


if __name__ == "__main__":
    main()
