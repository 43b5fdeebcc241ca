"""Prints the engine-vs-oracle parity numbers of gim_loftr on match-rich synthetic pairs (GPU box).
    python tools/parity_probe.py [H W frac]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import loftr_oracle as O  # noqa: E402
from tools import synth_loftr as S  # noqa: E402
from tools.parity import flip_margins, parity_vs_oracle  # noqa: E402

H, W, frac = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (480, 640, 0.45)
torch.set_num_threads(min(64, os.cpu_count() or 1))
model, sd = S.synthetic_model("fp32")
nb = 2
c0, c1 = S.textured_pairs(nb, H, W, seed=1234, frac=frac)
t = time.time()
with torch.no_grad():
    ref = O.loftr_forward(sd, {"image0": c0[:, :1], "image1": c1[:, :1], "color0": c0, "color1": c1})
print(f"oracle {nb} pairs {time.time() - t:.1f}s, matches {ref['b_ids'].numel()}", flush=True)
q = torch.quantile(ref["mconf"], torch.tensor([0.05, 0.25, 0.5, 0.75]))
print("oracle mconf quantiles", [round(float(x), 3) for x in q])
model = model.cuda()
for prec, sim in (("fp32", "fp32"), ("bf16", "fp32"), ("bf16", "bf16")):
    model.set_precision(prec, sim)
    for use_graph in (False,):
        model.use_graph = use_graph
        d = {"image0": c0[:, :1].cuda(), "image1": c1[:, :1].cuda(), "color0": c0.cuda(), "color1": c1.cuda()}
        model(d)
        torch.cuda.synchronize()
        for b in range(nb):
            p = parity_vs_oracle(d, ref, b, b)
            print(prec, "sim", sim, "pair", b, p, flush=True)
            fm = flip_margins(d, ref, b, b)
            for r in fm[:12]:
                print("    flip i=%d j=%d in_oracle=%s conf=%.5f |conf-thr|=%.2e gap=%.2e" % r)
