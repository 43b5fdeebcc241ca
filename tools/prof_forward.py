"""N plain forwards of the benchmarked gim_loftr workload (batch 8, 640x480, default precision mode or argv[2], match-rich synthetic pairs: tools/synth_loftr.py)
for rocprofv3 --kernel-trace / --pmc runs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import synth_loftr as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
prec = sys.argv[2] if len(sys.argv) > 2 else "fp16"   # the bench headline mode (round 6)
m, _ = S.synthetic_model(prec)
m = m.cuda()
c0, c1 = S.textured_pairs(8, 480, 640, seed=1234, frac=0.45)
c0, c1 = c0.cuda(), c1.cuda()
for _ in range(n):
    d = {"image0": c0[:, :1], "image1": c1[:, :1], "color0": c0, "color1": c1}
    m(d)
torch.cuda.synchronize()
print("precision", prec, "matches per pair", d["b_ids"].numel() / 8)
