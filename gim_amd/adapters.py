"""Caller-side adapters of the dense matchers: the glue between `match()` / `sample()` and the
`{mkpts0, mkpts1, mconf}` contract, as the reference's two other call sites write it.

  get_padding_size        tools/__init__.py:202-218
  dense_demo_inference    demo.py:420-462        (pad to the model's aspect ratio, match, sample, pixels, un-pad, in-bounds mask)
  HlocDenseMatcher        hloc/matchers/dkm.py:15-154 (image0/image1 swapped, optional class-id masks, 8192 samples, top-k)

(`trainer/lightning.py:134-156`, the ZEB adapter, is `gim_amd.dkm.gim_dkm_inference` / `gim_amd.roma.gim_roma_inference`.)
The models are `gim_amd.dkm.DKMv3(...)` / `gim_amd.roma.RoMa(...)`; everything here works on their device tensors, the only
host synchronisation is the boolean-mask compaction the reference has as well.
"""
import torch
import torch.nn.functional as F

from . import ops


def get_padding_size(image, h, w):
    """tools/__init__.py:202-218: symmetric padding that brings [.., H, W] to the aspect ratio w / h"""
    orig_width, orig_height = image.shape[3], image.shape[2]
    aspect_ratio = w / h
    new_width = max(orig_width, int(orig_height * aspect_ratio))
    new_height = max(orig_height, int(orig_width / aspect_ratio))
    pad_height, pad_width = new_height - orig_height, new_width - orig_width
    pad_top, pad_left = pad_height // 2, pad_width // 2
    return orig_width, orig_height, pad_left, pad_width - pad_left, pad_top, pad_height - pad_top


def _unpad_and_mask(sparse_matches, hw0, hw1, pads0, pads1):
    """normalised -> pixel coordinates of the padded images, minus the padding, plus the in-bounds mask (demo.py:437-457)"""
    ow0, oh0, pl0, _, pt0, _ = pads0
    ow1, oh1, pl1, _, pt1, _ = pads1
    kpts0, kpts1 = ops.dense_to_pixels(sparse_matches, hw0, hw1)
    kpts0 = kpts0 - kpts0.new_tensor((pl0, pt0))[None]
    kpts1 = kpts1 - kpts1.new_tensor((pl1, pt1))[None]
    mask = (kpts0[:, 0] > 0) & (kpts0[:, 1] > 0) & (kpts1[:, 0] > 0) & (kpts1[:, 1] > 0)
    mask = mask & (kpts0[:, 0] <= ow0 - 1) & (kpts1[:, 0] <= ow1 - 1) & (kpts0[:, 1] <= oh0 - 1) & (kpts1[:, 1] <= oh1 - 1)
    return kpts0, kpts1, mask


@torch.no_grad()
def dense_demo_inference(model, image0, image1, h, w, num=5000):
    """demo.py:420-462 for gim_dkm (h, w = 672, 896) and gim_roma (672, 672): [1,3,H,W] images in [0,1] ->
    (kpts0 [M,2], kpts1 [M,2], b_ids [M], mconf [M]) in pixels of the un-padded images"""
    pads0, pads1 = get_padding_size(image0, h, w), get_padding_size(image1, h, w)
    image0_ = F.pad(image0, (pads0[2], pads0[3], pads0[4], pads0[5]))
    image1_ = F.pad(image1, (pads1[2], pads1[3], pads1[4], pads1[5]))
    dense_matches, dense_certainty = model.match(image0_, image1_)
    sparse_matches, mconf = model.sample(dense_matches, dense_certainty, num)
    kpts0, kpts1, mask = _unpad_and_mask(sparse_matches, image0_.shape[-2:], image1_.shape[-2:], pads0, pads1)
    b_ids = torch.where(mconf[None])[0]
    return kpts0[mask], kpts1[mask], b_ids[mask], mconf[mask]


class HlocDenseMatcher(torch.nn.Module):
    """hloc/matchers/dkm.py:15-154 without the file I/O: `forward({'image0', 'image1'[, 'mask0', 'mask1']})` ->
    `{'keypoints0', 'keypoints1', 'scores'[, 'batch_indexes']}`.  The plugin matches the pair in swapped order ("we refine
    kpts in image0", :44-55) and switches the names back at the end.  `mask0/1`: the semantic class-id maps the plugin loads
    from `segment/*.npy` ([H,W], already at the image size); as shipped, pixels whose class id is 0 are blacked out
    (:69-76 -- the person / sky / car exclusion is overwritten)."""

    def __init__(self, net, h=672, w=896, max_num_matches=None, num_samples=8192):
        super().__init__()
        self.net, self.h, self.w = net, h, w
        self.max_num_matches, self.num_samples = max_num_matches, num_samples

    @torch.no_grad()
    def forward(self, data):
        image0, image1 = data["image1"], data["image0"]                      # swapped, like `rename`
        mask0, mask1 = data.get("mask1"), data.get("mask0")
        if mask0 is not None:
            image0 = image0 * (torch.as_tensor(mask0, device=image0.device) != 0)[None, None]
        if mask1 is not None:
            image1 = image1 * (torch.as_tensor(mask1, device=image1.device) != 0)[None, None]
        pads0, pads1 = get_padding_size(image0, self.h, self.w), get_padding_size(image1, self.h, self.w)
        image0 = F.pad(image0, (pads0[2], pads0[3], pads0[4], pads0[5]))
        image1 = F.pad(image1, (pads1[2], pads1[3], pads1[4], pads1[5]))
        dense_matches, dense_certainty = self.net.match(image0, image1)
        sparse_matches, mconf = self.net.sample(dense_matches, dense_certainty, self.num_samples)
        m = mconf > 0
        mconf, sparse_matches = mconf[m], sparse_matches[m].contiguous()
        kpts0, kpts1, mask = _unpad_and_mask(sparse_matches, image0.shape[-2:], image1.shape[-2:], pads0, pads1)
        b_ids = torch.zeros_like(mconf, dtype=torch.long)
        kpts0, kpts1, scores, b_ids = kpts0[mask], kpts1[mask], mconf[mask], b_ids[mask]
        if self.max_num_matches is not None and len(scores) > self.max_num_matches:
            keep = torch.argsort(scores, descending=True)[:self.max_num_matches]
            kpts0, kpts1, scores = kpts0[keep], kpts1[keep], scores[keep]
        # names switched back: the model's first image is the caller's image1
        return {"keypoints0": kpts1, "keypoints1": kpts0, "scores": scores, "batch_indexes": b_ids}
