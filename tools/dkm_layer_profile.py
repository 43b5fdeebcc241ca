"""HIP events around every gim_conv2d_bn_act launch of one gim_dkm match() (bf16, 672x896 -> 1152x1536): per-shape table."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gim_amd import ops
from gim_amd.dkm import DKMv3
torch.manual_seed(0)
m = DKMv3(None, 672, 896, upsample_preds=True, precision="bf16").eval()
m.upsample_res = (1152, 1536)
g = torch.Generator().manual_seed(1)
im0, im1 = torch.rand(1, 3, 672, 896, generator=g).cuda(), torch.rand(1, 3, 672, 896, generator=g).cuda()
for _ in range(2):
    m.match(im0, im1)
torch.cuda.synchronize()
ops.PROFILE = []
m.match(im0, im1)
torch.cuda.synchronize()
prof, ops.PROFILE = ops.PROFILE, None
by = collections.defaultdict(lambda: [0, 0.0, 0.0])
for e0, e1, f, lab in prof:
    v = by[lab]; v[0] += 1; v[1] += e0.elapsed_time(e1); v[2] += f
tot = sum(v[1] for v in by.values())
print(f"conv / linear launches: {len(prof)}, {tot:.2f} ms")
for lab, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{lab:44s} x{v[0]:3d} {v[1]:7.3f} ms  {v[1] / v[0] * 1e3:7.1f} us each  {v[2] / (v[1] * 1e-3) / 1e12:7.1f} TFLOP/s")
