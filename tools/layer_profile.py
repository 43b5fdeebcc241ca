"""Per-launch profile of one gim_loftr forward (HIP events around every gim_conv2d_bn_act launch) plus
whole-forward timing.  Run on the GPU box:  python tools/layer_profile.py [--dma 0|1] [--precision bf16]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gim_amd import ops  # noqa: E402
from gim_amd.loftr import LoFTR, get_cfg_defaults, lower_config  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dma", type=int, default=1)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--batch", type=int, default=8)
args = ap.parse_args()
os.environ["GIM_FLAGS"] = f"lds_dma={args.dma}"
torch.manual_seed(0)
cfg = lower_config(get_cfg_defaults())["loftr"]
cfg["precision"] = args.precision
m = LoFTR(cfg).eval().cuda()
g = torch.Generator().manual_seed(1234)
c0 = torch.rand(args.batch, 3, 480, 640, generator=g).cuda()
c1 = torch.rand(args.batch, 3, 480, 640, generator=g).cuda()


def step():
    d = {"image0": c0[:, :1], "image1": c1[:, :1], "color0": c0, "color1": c1}
    m(d)
    return d


for _ in range(3):
    step()
torch.cuda.synchronize()
print("hip graph:", m.use_graph)
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print(f"forward: {(time.perf_counter() - t0) * 100:.3f} ms/step  (dma={args.dma} {args.precision})")
m.use_graph = False
step()
ops.PROFILE = []
for _ in range(3):
    step()
torch.cuda.synchronize()
prof, ops.PROFILE = ops.PROFILE, None
by = {}
for e0, e1, f, lab in prof:
    ms, fl, n = by.get(lab, (0.0, 0.0, 0))
    by[lab] = (ms + e0.elapsed_time(e1), fl + f, n + 1)
tot = sum(v[0] for v in by.values()) / 3
print(f"igemm total {tot:.3f} ms/step, {sum(v[1] for v in by.values()) / 3 / tot / 1e9:.1f} TFLOP/s, {len(prof) // 3} launches")
print(f"{'layer':38s} {'n':>3s} {'ms/step':>8s} {'us/launch':>9s} {'TFLOP/s':>8s}")
for k, v in sorted(by.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:38s} {v[2] // 3:3d} {v[0] / 3:8.3f} {1e3 * v[0] / v[2]:9.1f} {v[1] / (v[0] * 1e-3) / 1e12:8.1f}")
