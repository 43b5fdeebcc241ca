"""diagnostic for gim_bneck_tail128: where do wrong values sit (tile / wave / chunk), on a launch with more workgroups than CUs"""
import sys, os, collections
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_bneck_tail import _blocks
from gim_amd import ops
from gim_amd.packing import pack_bneck_tail, fold_bn

blk, nxt = _blocks(7, 128)
g = torch.Generator().manual_seed(3)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
t2 = F.relu(torch.randn(nb, 120, 160, 128, generator=g)).to(torch.bfloat16).cuda()
res = torch.randn(nb, 120, 160, 512, generator=g).to(torch.bfloat16).cuda()
pk = pack_bneck_tail(blk, nxt.conv1, nxt.bn1, "cuda")
bn = lambda m: (m.weight, m.bias, m.running_mean, m.running_var, m.eps)
w3, b3 = fold_bn(blk.conv3.weight, bn(blk.bn3)); w1, b1 = fold_bn(nxt.conv1.weight, bn(nxt.bn1))
r = lambda t: t.to(torch.bfloat16).float()
M = nb * 120 * 160
with torch.no_grad():
    xr = torch.relu(t2.view(M, 128).float() @ r(w3.view(512, 128)).cuda().T + b3.cuda() + res.view(M, 512).float())
    tr = torch.relu(r(xr) @ r(w1.view(128, 512)).cuda().T + b1.cuda())
for it in range(3):
    xo, t1 = ops.bneck_tail(t2, res, pk)
    torch.cuda.synchronize()
    ex = (xo.view(M, 512).float() - xr).abs() > 0.05 * xr.abs().max()
    et = (t1.view(M, 128).float() - tr).abs() > 0.05 * tr.abs().max()
    print(f"run {it}: bad x' elements {int(ex.sum())} rows {int(ex.any(1).sum())}; bad t1' elements {int(et.sum())} rows {int(et.any(1).sum())} of {M}")
    rows = torch.nonzero(ex.any(1)).flatten().cpu()
    if rows.numel():
        tiles = collections.Counter((rows // 256).tolist())
        waves = collections.Counter(((rows % 256) // 32).tolist())
        chunks = collections.Counter((torch.nonzero(ex)[:, 1] // 64).cpu().tolist())
        print("   x': tiles", sorted(tiles.items())[:20], "...", len(tiles), "tiles; waves", sorted(waves.items()), "chunks", sorted(chunks.items()))
        rr = int(rows[0]); cc = torch.nonzero(ex[rr]).flatten()[:8].cpu().tolist()
        print("   first bad row", rr, "cols", cc, "got", xo.view(M, 512)[rr, cc].float().cpu().tolist(), "want", xr[rr, cc].cpu().tolist())
    rows = torch.nonzero(et.any(1)).flatten().cpu()
    if rows.numel():
        tiles = collections.Counter((rows // 256).tolist())
        waves = collections.Counter(((rows % 256) // 32).tolist())
        print("   t1': tiles", sorted(tiles.items())[:20], "...", len(tiles), "tiles; waves", sorted(waves.items()))
