"""Single-layer microbenchmark of gim_conv2d_bn_act (for rocprofv3 --pmc runs and A/B of variants).
  python tools/microbench_conv.py --cin 64 --cout 256 --k 1 --H 240 --W 320 --B 16 --res 1 --act relu [--dma 0]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gim_amd import ops, _lib  # noqa: E402
from gim_amd.packing import cstore, pack_conv, torch_dtype  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cin", type=int, default=64)
ap.add_argument("--cout", type=int, default=256)
ap.add_argument("--k", type=int, default=1)
ap.add_argument("--stride", type=int, default=1)
ap.add_argument("--H", type=int, default=240)
ap.add_argument("--W", type=int, default=320)
ap.add_argument("--B", type=int, default=16)
ap.add_argument("--res", type=int, default=0)
ap.add_argument("--act", default="relu")
ap.add_argument("--dma", type=int, default=1)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--ldx", type=int, default=0, help="pixel stride of the input in elements (default: cin_pad)")
ap.add_argument("--copy", type=int, default=0, help="also time a plain device copy of the output tensor")
a = ap.parse_args()
dt = _lib.GIM_BF16 if a.precision == "bf16" else _lib.GIM_F32
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
w = torch.randn(a.cout, a.cin, a.k, a.k, generator=g) * (2.0 / (a.cin * a.k * a.k)) ** 0.5
pk = pack_conv(w, None, dt, dev, stride=a.stride, pad=a.k // 2, cin_pad=cstore(a.cin, dt))
x = torch.randn(a.B, a.H, a.W, pk.cin_pad, generator=g).to(torch_dtype(dt)).to(dev)
Ho = (a.H + 2 * (a.k // 2) - a.k) // a.stride + 1
Wo = (a.W + 2 * (a.k // 2) - a.k) // a.stride + 1
res = torch.randn(a.B, Ho, Wo, pk.n_store, generator=g).to(torch_dtype(dt)).to(dev) if a.res else None
act = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "leaky": ops.ACT_LEAKY}[a.act]
if a.ldx:
    xf = torch.zeros(a.B, a.H, a.W, a.ldx, dtype=x.dtype, device=dev)
    xf[..., :pk.cin_pad] = x
    xr = xf.view(-1, a.ldx)[:, :pk.cin_pad]
    yb = torch.empty(a.B, Ho, Wo, pk.n_store, dtype=x.dtype, device=dev)
    _conv = ops.conv2d
    def conv2d(x_, pk_, act_, res=None, lds_dma=True):
        ops.conv_rows(xr, pk_, (a.B, a.H, a.W, Ho, Wo), yb.view(-1, pk_.n_store), act_, res.view(-1, res.shape[-1]) if res is not None else None, 0, lds_dma)
        return yb
    ops.conv2d = conv2d
for _ in range(3):
    y = ops.conv2d(x, pk, act, res=res, lds_dma=bool(a.dma))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    y = ops.conv2d(x, pk, act, res=res, lds_dma=bool(a.dma))
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
M = a.B * Ho * Wo
fl = 2.0 * M * a.cout * a.cin * a.k * a.k
by = x.numel() * x.element_size() + y.numel() * y.element_size() * (2 if a.res else 1)
print(f"{a.cin}->{a.cout} k{a.k}s{a.stride} M={M} res={a.res} dma={a.dma}: {ms * 1e3:.1f} us  "
      f"{fl / ms / 1e9:.1f} TFLOP/s  {by / ms / 1e6:.0f} GB/s (min traffic {by / 1e6:.0f} MB)")
if a.copy:
    z = torch.empty_like(y)
    for _ in range(3):
        z.copy_(y)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.iters):
        z.copy_(y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(f"device copy of {y.numel() * y.element_size() / 1e6:.0f} MB: {ms * 1e3:.1f} us = {2 * y.numel() * y.element_size() / ms / 1e6:.0f} GB/s (r+w)")
