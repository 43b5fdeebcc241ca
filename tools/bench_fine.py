"""Fine-level stage (gather + fine transformer + fine matching) at realistic match counts, fused kernel vs the unfused
launch sequence (report the fine stage at M in {500, 1500, 4000} per pair).
    python tools/bench_fine.py [per_pair ...]      GIM_FINE_ONLY=fused|unfused restricts the variants (PMC runs)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gim_amd.loftr import LoFTR, get_cfg_defaults, lower_config

torch.manual_seed(0)
cfg = lower_config(get_cfg_defaults())["loftr"]; cfg["precision"] = "bf16"
m = LoFTR(cfg).eval().cuda()
dev = torch.device("cuda")
bs, hc, wc = 8, 60, 80
g = torch.Generator().manual_seed(0)
f0 = torch.randn(bs, 240, 320, 128, generator=g).to(torch.bfloat16).to(dev)
f1 = torch.randn(bs, 240, 320, 128, generator=g).to(torch.bfloat16).to(dev)
only = os.environ.get("GIM_FINE_ONLY", "")
iters = int(os.environ.get("GIM_FINE_ITERS", "5"))
for per_pair in [int(a) for a in sys.argv[1:]] or (500, 1500, 4000):
    M = per_pair * bs
    b_ids = torch.arange(bs).repeat_interleave(per_pair).to(dev)
    i_ids = torch.randint(0, hc * wc, (M,), generator=g).to(dev).sort().values
    j_ids = torch.randint(0, hc * wc, (M,), generator=g).to(dev)
    mk1 = torch.rand(M, 2, generator=g).to(dev) * 600
    for fused in (True, False):
        if only and only != ("fused" if fused else "unfused"):
            continue
        run = lambda: m._fine_level(f0, f1, b_ids, i_ids, j_ids, mk1, None, False, (hc, wc), (hc, wc), (480, 640), fused)  # noqa: E731
        for _ in range(2): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(iters): run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
        print(f"fine stage {'fused  ' if fused else 'unfused'}: {per_pair} matches/pair x {bs} pairs (M={M}): {dt * 1e3:.3f} ms per batch  "
              f"({33.6e6 * M / dt / 1e12:.1f} TFLOP/s on 33.6 MFLOP/match)", flush=True)
