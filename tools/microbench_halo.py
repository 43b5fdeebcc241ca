"""The 3x3 / stride-1 layers of the gim_loftr forward on the two kernels behind gim_conv2d_bn_act, one process, interleaved rounds:
generic implicit GEMM (persistent 256x256 / 128x128 tiles) and the halo kernel.  (Round 6 timed three re-scheduled K loops with it --
profiles/r06_halo_variants.txt; their kernels are in the git history.)
    python tools/microbench_halo.py [bf16|fp16] [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gim_amd import ops, _lib  # noqa: E402
from gim_amd.packing import cstore, pack_conv, torch_dtype  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dt = _lib.GIM_BF16 if prec == "bf16" else _lib.GIM_F16
dev = torch.device("cuda")
LAYERS = [(196, 196, 16, 240, 320, "leaky"), (196, 128, 16, 240, 320, "none"), (256, 256, 16, 120, 160, "leaky"), (256, 196, 16, 120, 160, "none"),
          (128, 128, 16, 120, 160, "relu"), (256, 256, 16, 60, 80, "relu")]
g = torch.Generator().manual_seed(0)
res = {}
cases = []
for cin, cout, B, H, W, act in LAYERS:
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    pk = pack_conv(w, None, dt, dev, stride=1, pad=1, cin_pad=cstore(cin, dt))
    x = torch.relu(torch.randn(B, H, W, pk.cin_pad, generator=g)).to(torch_dtype(dt)).to(dev)
    x[..., cin:] = 0
    y = torch.empty(B, H, W, pk.n_store, dtype=x.dtype, device=dev)
    cases.append((f"{cin}->{cout} M={B * H * W}", pk, x, y, {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "leaky": ops.ACT_LEAKY}[act], 2.0 * B * H * W * cout * cin * 9))


def run(kind, pk, x, y, act):
    B, H, W, cs = x.shape
    if kind == "generic":
        ops.conv_rows(x.view(-1, cs), pk, (B, H, W, H, W), y.view(-1, pk.n_store), act)
    else:
        ops.conv3x3_halo(x, pk, y, act)


ref = {}
for r in range(rounds + 1):
    for name, pk, x, y, act, fl in cases:
        for kind in ("generic", "halo1"):
            if pk.halo is None:
                continue
            run(kind, pk, x, y, act)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run(kind, pk, x, y, act)
            e1.record()
            torch.cuda.synchronize()
            if r:
                res.setdefault((name, kind), []).append(e0.elapsed_time(e1) / 5 * 1e3)
            if r == 0:
                yy = y.float()
                if kind == "generic":
                    ref[name] = yy.clone()
                else:
                    d = (yy - ref[name]).abs().max().item() / (ref[name].abs().max().item() or 1.0)
                    print(f"{name} {kind}: max deviation from the generic kernel {d:.2e} of scale")
for name, pk, x, y, act, fl in cases:
    row = []
    for kind in ("generic", "halo1"):
        v = res.get((name, kind))
        if v:
            m = sorted(v)[len(v) // 2]
            row.append(f"{kind} {m:7.1f} us {fl / m / 1e6:7.1f} TF")
    print(f"{name:22s} " + " | ".join(row))
