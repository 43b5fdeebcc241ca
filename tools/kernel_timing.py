"""Phase stamps of the fused Bottleneck kernels and the implicit-GEMM kernels inside one eager forward of the benchmark workload
(development build: GIM_HIPCC_EXTRA=-DGIM_TIMING python -m gim_amd.build --force; csrc/gim_common.h: GIM_TT).
    python tools/kernel_timing.py [bf16|fp16]
Every ops.bneck64 / bneck64_ds / bneck_tail / bneck_tail_ds / conv2d call of the forward is followed by a device sync and a read of the
kernel's stamp array; per call: launch time from HIP events and the mean shader cycles per phase over the workgroups (wave 0)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gim_amd import ops, _lib  # noqa: E402
from tools import synth_loftr as S  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
raw = ctypes.CDLL(_lib.LIB_PATH)
sfx = "_f16" if prec == "fp16" else ""


def read(name, nwg):
    fn = getattr(raw, f"gim_timing_{name}{sfx}", None)
    if fn is None:
        raise SystemExit("library built without -DGIM_TIMING")
    nwg = min(nwg, 8192)
    buf = np.zeros((nwg, 8, 16), dtype=np.uint64)
    assert fn(buf.ctypes.data_as(ctypes.c_void_p), nwg) == 0
    return buf.astype(np.int64)


def have(name):
    return hasattr(raw, f"gim_timing_clear_{name}{sfx}")


def clear(name):
    if have(name):
        assert getattr(raw, f"gim_timing_clear_{name}{sfx}")() == 0


def timed(fn, *a, **k):
    torch.cuda.synchronize()
    for nm in ("bneck64", "bneck_tail", "conv"):
        clear(nm)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn(*a, **k)
    e1.record()
    torch.cuda.synchronize()
    return r, e0.elapsed_time(e1) * 1e3


def wrap_bneck64(orig, label):
    def f(t1, *a, **k):
        r, us = timed(orig, t1, *a, **k)
        B, H, W, _ = t1.shape
        nwg = B * (H // 8) * (W // 32)
        t = read("bneck64", nwg)[:, :, :12]
        n1 = r[1].shape[-1] if r[1] is not None else 0
        names = ["loads + DMA issue, wait T1 / W2, barrier", "conv2 (72 MFMA)", "wait W3 + barrier + W1n DMA issue", "pack t2 + conv3 (32 MFMA)",
                 "epilogue pass 0", "epilogue pass 1", "epilogue pass 2", "epilogue pass 3", "wait + barrier", "conv1' MFMAs", "t1' stores"]
        d = np.diff(t[:, 0, :], axis=1)
        tot = t[:, 0, 11] - t[:, 0, 0]
        span = t[:, :, 11].max() - t[:, :, 0].min()
        print(f"\n{label} -> N1={n1}: {us:.1f} us, {nwg} workgroups, {tot.mean():.0f} cycles per workgroup (wave 0), launch span {span} cycles = {span / us:.0f} per us")
        for i in range(11):
            print(f"    {names[i]:36s} {d[:, i].mean():8.0f}  ({100 * d[:, i].mean() / tot.mean():5.1f} %)")
        return r
    return f


def wrap_tail(orig, label):
    def f(t2, *a, **k):
        r, us = timed(orig, t2, *a, **k)
        rows = r[1].shape[0] * r[1].shape[1] * r[1].shape[2]
        P = t2.shape[-1]
        nw = 4 if (P == 256 and rows // 256 < 768) else 8
        nwg = rows // (32 * nw)
        t = read("bneck_tail", nwg)
        w0 = t[:, 0, :]
        tot = w0[:, 3] - w0[:, 0]
        span = t[:, :nw, 3].max() - t[:, :nw, 0].min()
        print(f"\n{label} P={P} N1={r[1].shape[-1]} rows={rows} NW={nw}: {us:.1f} us, {nwg} workgroups, {tot.mean():.0f} cycles per workgroup, launch span {span} = {span / us:.0f} per us")
        for nm, v in (("prologue (t2 loads, biases, first DMA)", w0[:, 1] - w0[:, 0]), ("chunk loop", w0[:, 2] - w0[:, 1]),
                      ("   waits + barrier", w0[:, 4]), ("   conv3 MFMAs", w0[:, 5]), ("   identity/relu/pack/issue/stores", w0[:, 6]), ("   conv1' MFMAs", w0[:, 7]),
                      ("t1' out", w0[:, 3] - w0[:, 2])):
            print(f"    {nm:40s} {v.mean():9.0f}  ({100 * v.mean() / tot.mean():5.1f} %)")
        return r
    return f


_conv = ops.conv2d


def conv2d(x, pk, act=0, res=None, **k):
    (y), us = timed(_conv, x, pk, act, res, **k)
    t = read("conv", 512)
    w0 = t[:, 0, :]
    live = w0[:, 1] > w0[:, 0]
    if live.sum() == 0:
        return y
    w0 = w0[live]
    tot = w0[:, 1] - w0[:, 0]
    B, H, W, cs = x.shape
    lab = f"{cs}->{pk.n_store} k{pk.kh}s{pk.stride} M={y.shape[0] * y.shape[1] * y.shape[2]}" + (" +res" if res is not None else "") + (" +ups" if k.get("ups") is not None else "")
    print(f"conv {lab:44s} {us:8.1f} us  wgs {int(live.sum()):4d} tiles/wg {w0[:, 6].mean():5.1f}  cycles/wg {tot.mean():9.0f}  K-loop {100 * w0[:, 4].mean() / tot.mean():5.1f} %  "
          f"epilogue {100 * w0[:, 5].mean() / tot.mean():5.1f} %  per tile: K {w0[:, 4].sum() / max(1, w0[:, 6].sum()):8.0f} epi {w0[:, 5].sum() / max(1, w0[:, 6].sum()):7.0f} cycles")
    return y


model, _ = S.synthetic_model(prec)
model = model.cuda()
model.use_graph = False
model.tf_chains = 1
c0, c1 = S.textured_pairs(8, 480, 640, seed=1234, frac=0.45)
c0, c1 = c0.cuda(), c1.cuda()
for _ in range(2):
    model({"image0": c0[:, :1], "image1": c1[:, :1], "color0": c0, "color1": c1})
torch.cuda.synchronize()
if have("bneck64"):
    ops.bneck64 = wrap_bneck64(ops.bneck64, "bneck64")
    ops.bneck64_ds = wrap_bneck64(ops.bneck64_ds, "bneck64_ds")
if have("bneck_tail"):
    ops.bneck_tail = wrap_tail(ops.bneck_tail, "bneck_tail")
    ops.bneck_tail_ds = wrap_tail(ops.bneck_tail_ds, "bneck_tail_ds")
if have("conv"):
    ops.conv2d = conv2d
model({"image0": c0[:, :1], "image1": c1[:, :1], "color0": c0, "color1": c1})
torch.cuda.synchronize()
