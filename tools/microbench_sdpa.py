"""sdpa microbenchmark: nb sequences x 4 heads, L = S keys, head dim 64.  python tools/microbench_sdpa.py [L] [nb] [iters] [dtype]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gim_amd import ops  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
tdt = torch.float32 if (len(sys.argv) > 4 and sys.argv[4] == "fp32") else torch.bfloat16
dev = torch.device("cuda:0")
qkv = torch.randn(nb * L, 768, device=dev).to(tdt)
vt = torch.empty(nb, 256, L, dtype=tdt, device=dev)
ops.lg_transpose(qkv[:, 512:], vt, nb, L, L, 256)
out = torch.empty(nb * L, 256, dtype=tdt, device=dev)
for _ in range(3):
    ops.sdpa(qkv[:, :256], qkv[:, 256:512], vt, out, nb, 4, L, L, L)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    ops.sdpa(qkv[:, :256], qkv[:, 256:512], vt, out, nb, 4, L, L, L)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
fl = nb * 4 * 4.0 * L * L * 64
print(f"sdpa L={L} nb={nb} {tdt}: {dt * 1e6:.1f} us  {fl / dt / 1e12:.1f} TFLOP/s")
