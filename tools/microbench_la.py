"""linear attention microbenchmark (coarse level: nb sequences x 4800 tokens, 8 heads x 32).  python tools/microbench_la.py [nb] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gim_amd import ops  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
L = 4800
dev = torch.device("cuda:0")
qkv = (torch.rand(nb * L, 768, device=dev) + 0.5).to(torch.bfloat16)
out = torch.empty(nb * L, 256, dtype=torch.bfloat16, device=dev)
ws = None
for _ in range(3):
    ws = ops.linear_attention(qkv[:, :256], qkv[:, 256:512], qkv[:, 512:], out, nb, L, nb, L, 8, ws)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    ws = ops.linear_attention(qkv[:, :256], qkv[:, 256:512], qkv[:, 512:], out, nb, L, nb, L, 8, ws)
e1.record()
torch.cuda.synchronize()
print(f"linear_attention nb={nb}: {e0.elapsed_time(e1) / iters * 1e3:.1f} us per call (kv + finalize + apply)")
