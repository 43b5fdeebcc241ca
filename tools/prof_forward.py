"""N plain forwards of gim_loftr (batch 8, 640x480, bf16) for rocprofv3 --kernel-trace --stats."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gim_amd.loftr import LoFTR, get_cfg_defaults, lower_config
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
torch.manual_seed(0)
cfg = lower_config(get_cfg_defaults())["loftr"]; cfg["precision"] = "bf16"
m = LoFTR(cfg).eval().cuda()
g = torch.Generator().manual_seed(1234)
c0 = torch.rand(8, 3, 480, 640, generator=g).cuda(); c1 = torch.rand(8, 3, 480, 640, generator=g).cuda()
for _ in range(n):
    m({"image0": c0[:, :1], "image1": c1[:, :1], "color0": c0, "color1": c1})
torch.cuda.synchronize()
