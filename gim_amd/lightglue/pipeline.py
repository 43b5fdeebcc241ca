"""The caller-side gim_lightglue adapter (`Trainer.gim_lightglue_inference`, trainer/lightning.py:161-193; the same
code in demo.py:472-511) with the per-pair Python gather loops replaced by the match-emission kernel
(SURVEY 8a row a15 / 8f item 2): detector on image0 / image1, matcher, matched keypoints scaled to
original-image pixels, `m_bids`, `mconf` written into `data`."""
import torch

from .._lib import GimHipError


@torch.no_grad()
def gim_lightglue_inference(detector, model, data):
    """Mutates `data` like the reference: adds hw0_i, hw1_i, mkpts0_f, mkpts1_f, m_bids, mconf.  Needs image0,
    image1 ([B,1,H,W] gray), scale0/1 ([B,2]) and either resize0/1 ([B,2] (h, w), the ZEB adapter, lightning.py:161-193) or
    image_size0/1 ([B,2] (w, h), demo.py:485-511 -- the matcher flips whichever it gets, lightglue.py:414-415);
    color0/1 only provide hw*_i."""
    img0, img1 = data["image0"], data["image1"]
    if not img0.is_cuda:
        raise GimHipError("gim_lightglue_inference needs device tensors: there is no CPU fallback")
    dev = img0.device
    pred = {}
    if img0.shape == img1.shape:
        # one detector launch sequence for both images (the reference calls the detector twice, lightning.py:166-173)
        both = detector({"image": torch.cat([img0, img1], 0)})
        B = img0.shape[0]
        pred["keypoints0"], pred["keypoints1"] = both["keypoints"][:B], both["keypoints"][B:]
        pred["descriptors0"], pred["descriptors1"] = both["descriptors"][:B], both["descriptors"][B:]
    else:
        for s, img in (("0", img0), ("1", img1)):
            # lightning.py:166-173 passes image_size = resize[:, [1, 0]]; demo.py:490-497 passes none (the detector ignores it, quirk A-10)
            extra = {"image_size": data["resize" + s][:, [1, 0]]} if ("resize" + s) in data else {}
            out = detector({"image": img, **extra})
            pred["keypoints" + s], pred["descriptors" + s] = out["keypoints"], out["descriptors"]
    scale0 = data["scale0"].to(device=dev, dtype=torch.float32).contiguous()
    scale1 = data["scale1"].to(device=dev, dtype=torch.float32).contiguous()
    pred.update(model({**pred, **data, "_adapter": (scale0, scale1)}))
    _, mconf, mk0, mk1, bids = pred.pop("_packed")
    data.update({
        "hw0_i": data["color0"].shape[2:] if "color0" in data else img0.shape[2:],
        "hw1_i": data["color1"].shape[2:] if "color1" in data else img1.shape[2:],
        "mkpts0_f": mk0, "mkpts1_f": mk1, "m_bids": bids, "mconf": mconf,
    })
    return pred
