"""Host-side packing of the 3x3 halo kernel (gim_amd/packing.py::pack_halo): the K order and the per-slab table are exercised by a
CPU emulation of the kernel's addressing -- halo rows of an 8 x 32 patch, row shift and channel sub-step per 16-channel K step --
against F.conv2d.  (The kernel itself: tests/test_gpu_conv_halo.py.)"""
import pytest
import torch
import torch.nn.functional as F


@pytest.mark.parametrize("cin,cout", [(196, 196), (256, 128), (40, 72), (64, 64)])
def test_pack_halo_table_reproduces_conv(cin, cout):
    from gim_amd import _lib
    from gim_amd.packing import HALO_W2, cstore, pack_conv
    g = torch.Generator().manual_seed(3)
    w = torch.randn(cout, cin, 3, 3, generator=g)
    pk = pack_conv(w, None, _lib.GIM_BF16, "cpu", stride=1, pad=1)
    wh, tab, nslab, _ = pk.halo
    tab = tab.reshape(nslab, 8)
    cs = cstore(cin, _lib.GIM_BF16)
    H, W = 8, 32                                   # one patch
    x = torch.zeros(1, H, W, cs)
    x[..., :cin] = torch.randn(1, H, W, cin, generator=g)
    xb = x.to(torch.bfloat16).float()
    halo = torch.zeros(H + 2, HALO_W2, (cs + 63) // 64 * 64)   # zero padded halo, channels padded to whole chunks
    halo[1:H + 1, 1:W + 1, :cs] = xb[0]
    halo = halo.reshape((H + 2) * HALO_W2, -1)
    wf = wh.float()
    out = torch.zeros(H * W, wf.shape[0])
    py, px = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    r0 = (py * HALO_W2 + px).reshape(-1)           # halo row of every output pixel at tap (0, 0)
    nchunk = int(tab[:, 0].max()) + 1
    for s in range(nslab):
        chunk, flags = int(tab[s, 0]), int(tab[s, 1])
        assert bool(flags & 1) == (s == 0 or int(tab[s - 1, 0]) != chunk)
        assert bool(flags & 2) == (s == nslab - 1 or int(tab[s + 1, 0]) != chunk)
        if flags & 1:
            assert int(tab[s, 6]) == ((chunk + 1) * 64 if chunk + 1 < nchunk else -1)
        for ks in range(4):
            e = int(tab[s, 2 + ks])
            shift, ksc = e & 0xff, (e >> 8) & 0xff
            a = halo[r0 + shift][:, chunk * 64 + ksc * 16: chunk * 64 + ksc * 16 + 16]      # [256, 16]
            out += a @ wf[:, s * 64 + ks * 16: s * 64 + ks * 16 + 16].t()
    ref = F.conv2d(xb[..., :cin].permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), padding=1)[0].permute(1, 2, 0).reshape(H * W, cout)
    assert (out[:, :cout] - ref).abs().max() <= 1e-3 * ref.abs().max()
    assert (out[:, cout:] == 0).all()
