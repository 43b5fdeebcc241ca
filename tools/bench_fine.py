"""Fine-level stage (gather + fine transformer + fine matching) at realistic match counts.
The random-weight bench has ~1 match per pair, so this stage is measured separately with synthetic matches
(report the fine stage at M in {500, 1500, 4000} per pair)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gim_amd import ops
from gim_amd.loftr import LoFTR, get_cfg_defaults, lower_config

torch.manual_seed(0)
cfg = lower_config(get_cfg_defaults())["loftr"]; cfg["precision"] = "bf16"
m = LoFTR(cfg).eval().cuda()
dev = torch.device("cuda")
P = m._prepack(dev)
bs, hc, wc, W, Cf = 8, 60, 80, 5, 128
g = torch.Generator().manual_seed(0)
f0 = torch.randn(bs, 240, 320, 128, generator=g).to(torch.bfloat16).to(dev)
f1 = torch.randn(bs, 240, 320, 128, generator=g).to(torch.bfloat16).to(dev)
for per_pair in (500, 1500, 4000):
    M = per_pair * bs
    b_ids = torch.arange(bs).repeat_interleave(per_pair).to(dev)
    i_ids = torch.randint(0, hc * wc, (M,), generator=g).to(dev)
    j_ids = torch.randint(0, hc * wc, (M,), generator=g).to(dev)
    mk1 = torch.rand(M, 2, generator=g).to(dev) * 600

    def run():
        F = m._TfBuffers(2 * M * W * W, Cf, torch.bfloat16, dev)
        ops.fine_gather(f0, f1, b_ids, i_ids, j_ids, M, wc, wc, 4, W, F.X32, F.CAT[:, :Cf])
        m._transformer(P, "f", m.loftr_fine, F, M, W * W, M, W * W)
        return ops.fine_match(F.X32[:M * W * W], F.X32[M * W * W:], mk1, b_ids, None, M, W * W, 2.0, False)

    for _ in range(2): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"fine stage: {per_pair} matches/pair x {bs} pairs (M={M}): {dt * 1e3:.3f} ms per batch  "
          f"({33.6e6 * M / dt / 1e12:.1f} TFLOP/s on 33.6 MFLOP/match)")
