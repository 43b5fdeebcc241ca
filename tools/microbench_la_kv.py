"""KV-state reduction of the coarse-level linear attention alone (gim_linear_attention_kv = la_kv + la_kv_finalize), timed inside a HIP
graph so that launch overhead of the host does not hide the kernels:  python tools/microbench_la_kv.py [reps]
K / V are column blocks of a [rows, 768] projection buffer
as in the forward (cross call: 8 sequences, self call: 16)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gim_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
S, H, C, N = 4800, 8, 256, 20
for kind in (torch.float16, torch.bfloat16):
    for nb in (8, 16):
        qkv = (torch.rand(nb * S, 3 * C, device=dev) + 0.5).to(kind)
        ws = None
        for _ in range(3):
            ws, _ = ops.linear_attention_state(qkv[:, C:2 * C], qkv[:, 2 * C:], nb, S, H, ws)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(N):
                ops.linear_attention_state(qkv[:, C:2 * C], qkv[:, 2 * C:], nb, S, H, ws)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (reps * N) * 1e3
        mb = nb * S * 2 * C * 2 / 1e6
        print(f"{str(kind)[6:]} nb={nb}: {us:.1f} us per call (kv + finalize), "
              f"{mb:.0f} MB of K / V -> {mb / us:.2f} TB/s")
