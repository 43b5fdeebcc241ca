"""Secondary workload: gim_lightglue (SuperPoint x2 + LightGlue + adapter) at 640x480, batch of pairs, one GPU.
    python tools/bench_lightglue.py [--pairs 8] [--steps 10] [--precision bf16] [--kpts 2048]
Prints one JSON line (pairs/s; algorithmic 334 GFLOP/pair from SURVEY 8d) and a per-stage split from HIP events."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=8)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--kpts", type=int, default=2048)
    ap.add_argument("--precision", default="bf16")
    a = ap.parse_args()
    from gim_amd.lightglue import LightGlue, SuperPoint, gim_lightglue_inference
    dev = torch.device("cuda:0")
    torch.manual_seed(0)  # random-init weights of the reference architecture
    det = SuperPoint({"max_num_keypoints": a.kpts, "force_num_keypoints": True, "detection_threshold": 0.0, "nms_radius": 3,
                      "trainable": False, "precision": a.precision}).eval()
    lg = LightGlue({"filter_threshold": 0.1, "flash": False, "checkpointed": True, "precision": a.precision}).eval()
    B = a.pairs
    gg = torch.Generator().manual_seed(1)
    img0 = torch.nn.functional.interpolate(torch.rand(B, 1, 120, 160, generator=gg), size=(480, 640), mode="bilinear")
    img0 = (0.7 * img0 + 0.3 * torch.rand(B, 1, 480, 640, generator=gg)).contiguous().to(dev)
    img1 = torch.roll(img0, shifts=(16, 24), dims=(2, 3)).contiguous()
    rs = torch.tensor([[480, 640]] * B, device=dev)
    sc = torch.ones(B, 2, device=dev)

    def step():
        data = {"image0": img0, "image1": img1, "resize0": rs, "resize1": rs, "scale0": sc, "scale1": sc}
        gim_lightglue_inference(det, lg, data)
        return data["mconf"].shape[0]

    for _ in range(a.warmup):
        nm = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        nm = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    # stage split
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    both = det({"image": torch.cat([img0, img1], 0)})
    ev[1].record()
    lg({"keypoints0": both["keypoints"][:B], "keypoints1": both["keypoints"][B:], "descriptors0": both["descriptors"][:B],
        "descriptors1": both["descriptors"][B:], "resize0": rs, "resize1": rs})
    ev[2].record()
    torch.cuda.synchronize()
    print(json.dumps({"metric": "image-pairs/sec at 640x480 (gim_lightglue)", "value": B / dt, "unit": "pairs/s",
                      "ms_per_step": dt * 1e3, "pairs": B, "keypoints": a.kpts, "precision": a.precision,
                      "matches_per_pair": nm / B, "superpoint_ms": ev[0].elapsed_time(ev[1]),
                      "lightglue_ms": ev[1].elapsed_time(ev[2]),
                      "achieved_tflops": B / dt * 334e9 / 1e12, "data": "synthetic"}))


if __name__ == "__main__":
    main()
