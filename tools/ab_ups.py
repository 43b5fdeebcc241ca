"""Lateral 1x1 convolution with the fused upsample-add (gim_conv2d_bn_act, a->ups), one layer, for same-box A/Bs of the library
(GIM_LIB=gim_amd/lib/alt/libgimhip.so python tools/ab_ups.py ... against the working tree's build).
    python tools/ab_ups.py [cin cout H W B]        defaults: the benchmark's 256->196 layer at 240 x 320, 16 images"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gim_amd import ops, _lib  # noqa: E402
from gim_amd.packing import cstore, pack_conv, torch_dtype  # noqa: E402

cin, cout, H, W, B = ([int(v) for v in sys.argv[1:6]] + [256, 196, 240, 320, 16][len(sys.argv) - 1:])[:5]
dt = _lib.GIM_F16 if os.environ.get("PREC", "fp16") == "fp16" else _lib.GIM_BF16
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
w = torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
pk = pack_conv(w, None, dt, dev, cin_pad=cstore(cin, dt))
x = torch.randn(B, H, W, pk.cin_pad, generator=g).relu().to(torch_dtype(dt)).to(dev)
u = torch.randn(B, H // 2, W // 2, pk.n_store, generator=g).to(torch_dtype(dt)).to(dev)
if os.environ.get('NOUPS'):
    u = None
for _ in range(3):
    y = ops.conv2d(x, pk, ups=u)
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = ops.conv2d(x, pk, ups=u)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
print(f"{os.environ.get('GIM_LIB', 'new'):36s} {cin}->{cout} +ups M={B * H * W}: {best * 1e3:8.1f} us  checksum {y.float().abs().sum().item():.6e}")
