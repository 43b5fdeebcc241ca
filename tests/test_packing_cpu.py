"""Host-side weight packing of the round-3 kernels on the CPU: the layouts the kernels index (no GPU needed)."""
import torch

from gim_amd import packing


def test_pack_token_emit_fragment_layout():
    """[wave][block][unit q][k16 step k][column fragment nf][lane = (k / 8 % 2) * 32 + n][8]: element (n, k) of block b sits where
    token_mlp.hip's projection stage reads it (wave = n / 64, nf = n / 32 % 2, unit = k / 64, step = k / 16 % 4)"""
    g = torch.Generator().manual_seed(0)
    ws = [torch.randn(256, 256, generator=g) for _ in range(3)]
    s = packing.pack_token_emit(ws, "cpu", torch.float32)
    assert s.numel() == 3 * 256 * 256
    v = s.view(4, 3, 4, 4, 2, 2, 32, 8)      # wave, block, unit, k16 step, nf, lane half, lane row, 8 consecutive k
    for b, n, k in [(0, 0, 0), (1, 37, 200), (2, 255, 255), (1, 64, 17), (0, 191, 128)]:
        wv, nf, row = n // 64, (n // 32) % 2, n % 32
        q, st, half, e = k // 64, (k // 16) % 4, (k // 8) % 2, k % 8
        assert v[wv, b, q, st, nf, half, row, e] == ws[b][n, k], (b, n, k)


def test_pack_bneck_ds_folds_the_downsample_branch():
    from gim_amd.loftr.loftr import _Bottleneck
    torch.manual_seed(1)
    ds = torch.nn.Sequential(torch.nn.Conv2d(64, 256, 1, bias=False), torch.nn.BatchNorm2d(256))
    blk, nxt = _Bottleneck(64, 64, 1, ds).eval(), _Bottleneck(256, 64, 1, None).eval()
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.5, 1.5); m.bias.normal_()
    w2, w3, wds, w1n, b2, b3ds, b1n = packing.pack_bneck_ds(blk, nxt, "cpu", torch.float32)
    _, _, _, _, b3, _ = packing.pack_bneck(blk, nxt, "cpu", torch.float32)
    bn = ds[1]
    sc = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    assert torch.allclose(wds, (ds[0].weight.reshape(256, 64) * sc[:, None]).detach(), atol=1e-6)
    assert torch.allclose(b3ds - b3, (bn.bias - bn.running_mean * sc).detach(), atol=1e-6)
    assert tuple(wds.shape) == (256, 64) and tuple(w1n.shape) == (64, 256) and tuple(w2.shape) == (64, 576)
