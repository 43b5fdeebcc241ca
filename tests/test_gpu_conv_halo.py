"""3x3 halo-tile convolution (conv_igemm.hip: conv3x3_halo_kernel) against the oracle's fp32 conv on bf16-rounded operands and
against the generic implicit-GEMM path (same products, different K order)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [  # B, H, W, cin, cout, act
    (2, 16, 64, 196, 196, "leaky"),   # 3 full chunks + the 16-channel tail, N = 196 in a 256-wide tile
    (1, 24, 32, 256, 256, "relu"),    # 4 full chunks, no tail
    (2, 8, 96, 128, 128, "relu"),     # 128-wide tile variant
    (1, 16, 32, 196, 128, "none"),
    (1, 8, 32, 64, 64, "none"),       # single chunk: the halo of the NEXT tile is prefetched at slab 0
    (3, 40, 64, 40, 72, "relu"),      # tail only (cin < 64), 3 sub-steps per tap, many tiles per workgroup list
]


@pytest.mark.parametrize("kind", ["bf16", "fp16"])
@pytest.mark.parametrize("case", CASES, ids=[f"{c[3]}to{c[4]}_{c[1]}x{c[2]}" for c in CASES])
def test_conv3x3_halo_vs_reference(case, kind):
    from gim_amd import _lib, ops
    from gim_amd.packing import cstore, pack_conv
    B, H, W, cin, cout, act = case
    gdt, tdt, tol = (_lib.GIM_BF16, torch.bfloat16, 1e-2) if kind == "bf16" else (_lib.GIM_F16, torch.float16, 1.5e-3)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    pk = pack_conv(w, None, gdt, dev, stride=1, pad=1, bias=bias)
    assert pk.halo is not None
    cs = cstore(cin, gdt)
    x = torch.zeros(B, H, W, cs)
    x[..., :cin] = torch.randn(B, H, W, cin, generator=g)
    xb = x.to(tdt).to(dev)
    actc = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "leaky": ops.ACT_LEAKY}[act]
    y = torch.full((B, H, W, pk.n_store), float("nan"), dtype=tdt, device=dev)
    ops.conv3x3_halo(xb, pk, y, actc)
    y2 = torch.empty_like(y)
    ops.conv_rows(xb.view(-1, cs), pk, (B, H, W, H, W), y2.view(-1, pk.n_store), actc)
    torch.cuda.synchronize()
    ref = F.conv2d(xb.float().cpu()[..., :cin].permute(0, 3, 1, 2), w.to(tdt).float(), bias, padding=1)
    ref = {"none": lambda v: v, "relu": F.relu, "leaky": lambda v: F.leaky_relu(v, 0.01)}[act](ref).permute(0, 2, 3, 1)
    got = y.float().cpu()
    assert torch.isfinite(got[..., :cout]).all()
    scale = ref.abs().max().item()
    assert (got[..., :cout] - ref).abs().max().item() <= tol * scale              # 16-bit output rounding
    assert (got - y2.float().cpu())[..., :cout].abs().max().item() <= tol * scale  # generic path: same products, other K order
    if pk.n_store > cout:
        assert (got[..., cout:] == 0).all()                                        # pad channels stay exact zeros


@pytest.mark.parametrize("kind", ["bf16", "fp16"])
@pytest.mark.parametrize("case", [(2, 16, 64, 512, 256), (1, 24, 96, 256, 196), (2, 8, 32, 64, 256), (3, 40, 160, 256, 200), (1, 6, 32, 256, 196)],
                         ids=["512to256", "256to196", "64to256_tiny", "256to200_rows_straddle_tiles", "three_source_rows"])
def test_conv1x1_fused_upsample_add(case, kind, monkeypatch):
    """FPN lateral conv + bilinear x2 (align_corners=True) + add (resnet.py:321-327): the launch that carries the upsampled map through the MFMA
    (gim_conv_args.ups; Epilogue::ups_accumulate: sources read transposed from LDS, bilinear weights as the pixel operand, ONE 16-bit rounding of
    conv + upsample) against the two-pass path (conv rounded, then upsample2x_add) and against torch in fp32 on the same 16-bit-valued operands."""
    from gim_amd import _lib, ops
    from gim_amd.packing import cstore, pack_conv
    B, H, W, cin, cout = case
    gdt, tdt, tol = (_lib.GIM_BF16, torch.bfloat16, 1e-2) if kind == "bf16" else (_lib.GIM_F16, torch.float16, 1.5e-3)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    w = torch.randn(cout, cin, 1, 1, generator=g) * (1.0 / cin) ** 0.5
    pk = pack_conv(w, None, gdt, dev)
    cs, ns = cstore(cin, gdt), pk.n_store
    x = torch.zeros(B, H, W, cs); x[..., :cin] = torch.randn(B, H, W, cin, generator=g)
    lo = torch.zeros(B, H // 2, W // 2, ns); lo[..., :cout] = torch.randn(B, H // 2, W // 2, cout, generator=g)
    # a transposed or shifted source would hide in noise: a ramp in x, y, channel and image on top of it
    ramp = (torch.arange(W // 2).view(1, 1, -1, 1) * 0.05 + torch.arange(H // 2).view(1, -1, 1, 1) * 0.11 + torch.arange(cout).view(1, 1, 1, -1) * 0.01
            + torch.arange(B).view(-1, 1, 1, 1) * 0.5)
    lo[..., :cout] += ramp
    xb, lob = x.to(tdt).to(dev), lo.to(tdt).to(dev)
    monkeypatch.setattr(ops, "FORCE_BIG_TILE", True)   # the 256 x 256 tile (the only one that carries the operand) whatever the size
    calls, two_pass = [], ops.upsample2x_add
    monkeypatch.setattr(ops, "upsample2x_add", lambda *a_, **k_: (calls.append(1), two_pass(*a_, **k_))[1])
    monkeypatch.setattr(ops, "UPS_FUSED", True)
    y_f = ops.conv2d(xb, pk, ups=lob)
    assert not calls, "the launch did not take the upsample operand"
    monkeypatch.setattr(ops, "UPS_FUSED", False)
    y_u = ops.conv2d(xb, pk, ups=lob)
    assert len(calls) == 1
    torch.cuda.synchronize()
    conv = F.conv2d(xb.float().cpu()[..., :cin].permute(0, 3, 1, 2), w.to(tdt).float())
    up = F.interpolate(lob.float().cpu()[..., :cout].permute(0, 3, 1, 2), scale_factor=2.0, mode="bilinear", align_corners=True)
    ref = (conv + up).permute(0, 2, 3, 1)
    scale = ref.abs().max().item()
    err = {}
    for name, y in (("fused", y_f), ("two-pass", y_u)):
        got = y.float().cpu()
        assert torch.isfinite(got).all(), name
        err[name] = (got[..., :cout] - ref).abs()
        assert err[name].max().item() <= tol * scale, (name, err[name].max().item() / scale)
        if ns > cout:
            assert (got[..., cout:] == 0).all(), name   # pad channels stay exact zeros
    # one rounding (of conv + upsample, with the bilinear weights rounded to the operand type) is no worse than two
    assert err["fused"].mean().item() <= 1.25 * err["two-pass"].mean().item() + 1e-7 * scale, (err["fused"].mean().item(), err["two-pass"].mean().item())
    assert (y_f.float() - y_u.float()).abs().max().item() <= 2 * tol * scale
