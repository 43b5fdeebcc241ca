import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from gim_amd.loftr import LoFTR, get_cfg_defaults, lower_config
from gim_amd.runner import pack_matches
torch.manual_seed(0)
cfg = lower_config(get_cfg_defaults())["loftr"]; cfg["precision"] = "bf16"
m = LoFTR(cfg).eval().cuda()
g = torch.Generator().manual_seed(1234)
c0 = torch.rand(8, 3, 480, 640, generator=g).cuda(); c1 = torch.rand(8, 3, 480, 640, generator=g).cuda()
def step(pack):
    d = {"image0": c0[:, :1], "image1": c1[:, :1], "color0": c0, "color1": c1}
    m(d)
    if pack: return pack_matches(d, list(range(8)))
for _ in range(3): step(True)
for pack in (False, True, False, True):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step(pack)
    torch.cuda.synchronize(); print("pack", pack, (time.perf_counter() - t0) * 50, "ms/step")
# breakdown of one forward on the host side
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step(True)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
