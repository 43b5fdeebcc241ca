"""coarse matching microbenchmark (batch 8, 60x80 cells, C=256) for timing / rocprofv3 --pmc.
   --planted : f1 = permuted f0 + noise (~3.8k matches per pair)   --sigma S : feature scale   --thr T"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gim_amd import ops
def opt(name, default):
    return type(default)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default
planted = "--planted" in sys.argv
sigma, thr = opt("--sigma", 2.0 if planted else 0.3), opt("--thr", 0.2)
g = torch.Generator().manual_seed(0)
N, L, C = 8, 4800, 256
f0 = torch.randn(N, L, C, generator=g) * sigma
if planted:
    perm = torch.stack([torch.randperm(L, generator=g) for _ in range(N)])
    f1 = torch.gather(f0, 1, perm[:, :, None].expand(-1, -1, C)) + 0.1 * sigma * torch.randn(N, L, C, generator=g)
else:
    f1 = torch.randn(N, L, C, generator=g) * sigma
f0, f1 = f0.cuda(), f1.cuda()
if "--bf16" in sys.argv:
    f0, f1 = f0.bfloat16(), f1.bfloat16()
for _ in range(3):
    r = ops.coarse_match(f0, f1, (60, 80), (60, 80), 8.0, 0.1, thr)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    r = ops.coarse_match(f0, f1, (60, 80), (60, 80), 8.0, 0.1, thr)
e1.record(); torch.cuda.synchronize()
print(f"coarse_match planted={planted} sigma={sigma} thr={thr}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per call, M={int(r.count[0])}")
