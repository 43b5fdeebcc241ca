"""Phase timing of gim_token_mlp_emit (development build: GIM_HIPCC_EXTRA=-DGIM_TOKEN_TIMING python -c "import __graft_entry__ as g; g.build()").
One coarse self-layer launch of the benchmark (16 sequences x 4800 tokens, attention apply fused in front, 3 projection blocks with
k / v gated to the second half): mean shader cycles per phase and wave over all workgroups, and the launch time."""
import ctypes, glob, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gim_amd import ops
from gim_amd._lib import ACT_ELU1, ACT_NONE
from gim_amd.packing import pack_token_emit, pack_token_mlp
from gim_amd.loftr.loftr import _EncoderLayer

tdt = torch.float16 if "--bf16" not in sys.argv else torch.bfloat16
nblk = 3
for a in sys.argv[1:]:
    if a.startswith("--blocks="):
        nblk = int(a.split("=")[1])
nb, L, C, H = 16, 4800, 256, 8
R = nb * L
torch.manual_seed(0)
layer = _EncoderLayer(256, 8)
wts, ln, eps = pack_token_mlp(layer, "cuda", tdt)
g = torch.Generator().manual_seed(1)
q = (torch.nn.functional.elu(torch.randn(R, C, generator=g)) + 1).to(tdt).cuda()
k = (torch.nn.functional.elu(torch.randn(R, C, generator=g)) + 1).to(tdt).cuda()
v = torch.randn(R, C, generator=g).to(tdt).cuda()
x32 = (torch.randn(R, C, generator=g) * 2).cuda()
cat = torch.zeros(R, 2 * C, dtype=tdt, device="cuda")
cat[:, :C] = x32.to(tdt)
ws, _ = ops.linear_attention_state(k, v, nb, L, H, None, None)
out = torch.empty(R, 3 * C, dtype=tdt, device="cuda")
ew = pack_token_emit([torch.randn(256, 256, generator=g) / 16 for _ in range(max(1, nblk))], "cuda", tdt)
spec = [(out[:, :C], ACT_ELU1, 0, R), (out[:, C:2 * C], ACT_ELU1, R // 2, R), (out[:, 2 * C:], ACT_NONE, R // 2, R)][:nblk]
emit = (ew, spec) if nblk else None
for _ in range(3):
    ops.token_mlp(q, cat[:, :C], x32, wts, ln, eps, kv=ws, L=L, S=L, emit=emit)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.token_mlp(q, cat[:, :C], x32, wts, ln, eps, kv=ws, L=L, S=L, emit=emit)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
print(f"launch: {us:.1f} us for {R // 64} workgroups ({nblk} projection blocks)")
lib = ctypes.CDLL(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gim_amd/lib/libgimhip.so"))[0])
fn = getattr(lib, "gim_token_mlp_timing_f16" if tdt == torch.float16 else "gim_token_mlp_timing", None)
if fn is None:
    print("library built without -DGIM_TOKEN_TIMING"); sys.exit(0)
nwg = R // 64
buf = np.zeros((nwg, 4, 10), dtype=np.uint64)
assert fn(buf.ctypes.data_as(ctypes.c_void_p), nwg) == 0
t = buf.astype(np.int64)
names = ["tile loads + barrier", "attention apply + barrier", "merge (4 units)", "norm1 + A write + barrier", "mlp (24 units)", "barrier",
         "norm2", "residual RMW + x stores", "barrier + A' write", "projection blocks"]
full = t[nwg // 2:]   # the gated half runs all blocks
d = np.diff(full, axis=2)
tot = (full[:, :, 9] - full[:, :, 0])
print(f"workgroups with all blocks: mean {tot.mean():.0f} cycles per wave from first to last stamp (min {tot.min()}, max {tot.max()})")
for i in range(9):
    print(f"  {names[i + 1] if False else names[i]:32s} {d[:, :, i].mean():9.0f} cycles  ({100 * d[:, :, i].mean() / tot.mean():5.1f} %)")
span = t[:, :, 9].max() - t[:, :, 0].min()
print(f"first start to last end over the launch: {span} cycles => {span / us:.0f} cycles per us")
starts = np.sort(t[:, 0, 0] - t[:, 0, 0].min())
print("workgroup start times (cycles), deciles:", [int(starts[int(f * (nwg - 1))]) for f in np.linspace(0, 1, 11)])
