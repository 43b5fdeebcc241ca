"""fp32 operands as IEEE-fp16 hi / lo pairs on the 16-bit MFMA (gim_conv_args.split16, Igemm::compute_split16): x w ~= hi hi + hi lo + lo hi with fp32
accumulation, against a float64 convolution of the same fp32 values and against the exact-product path (v_mfma_f32_32x32x2_f32) of the same launch."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [  # B, H, W, cin, cout, k, stride, res, act
    (2, 24, 32, 64, 128, 3, 1, False, "relu"),
    (1, 16, 32, 256, 256, 1, 1, True, "relu"),
    (1, 17, 23, 40, 72, 3, 2, False, "leaky"),     # ragged rows, K tail, the 256 x 64 tile
    (2, 30, 40, 128, 196, 3, 1, False, "none"),
    (1, 60, 80, 256, 256, 3, 1, False, "relu"),    # layer 3's conv2 at one image
]


@pytest.mark.parametrize("big", [False, True], ids=["tile128", "tile256"])
@pytest.mark.parametrize("case", CASES, ids=[f"{c[3]}to{c[4]}_k{c[5]}s{c[6]}" for c in CASES])
def test_split16_conv_vs_float64(case, big, monkeypatch):
    """big: the 256 x 256 tile (fp32 output, no residual, N padded to 256) takes the split launch whatever its size (gim_conv_args.use_lds_dma = 3)"""
    from gim_amd import _lib, ops
    if big and (case[7] or case[4] not in (196, 256)):
        pytest.skip("the 256 x 256 tile takes no residual and needs npad % 256 == 0")
    monkeypatch.setattr(ops, "FORCE_BIG_TILE", big)
    from gim_amd.packing import cstore, pack_conv
    B, H, W, cin, cout, k, stride, has_res, act = case
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    pk = pack_conv(w, None, _lib.GIM_F32, dev, stride=stride, pad=k // 2, bias=bias)
    cs = cstore(cin, _lib.GIM_F32)
    x = torch.zeros(B, H, W, cs)
    # magnitudes over six decades, half of them zero (post-ReLU maps): small values exercise the fp16 subnormal low halves
    mag = torch.exp(torch.rand(B, H, W, cin, generator=g) * 14.0 - 9.0)
    x[..., :cin] = mag * (torch.rand(B, H, W, cin, generator=g) > 0.5) * torch.sign(torch.randn(B, H, W, cin, generator=g))
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    res = torch.randn(B, Ho, Wo, pk.n_store, generator=g) if has_res else None
    actc = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "leaky": ops.ACT_LEAKY}[act]
    xd, rd = x.to(dev), (res.to(dev) if has_res else None)
    ref = F.conv2d(x[..., :cin].permute(0, 3, 1, 2).double(), w.double(), bias.double(), stride=stride, padding=k // 2)
    if has_res:
        ref = ref + res[..., :cout].permute(0, 3, 1, 2).double()
    ref = {"none": lambda v: v, "relu": F.relu, "leaky": lambda v: F.leaky_relu(v, 0.01)}[act](ref).permute(0, 2, 3, 1)
    scale = ref.abs().max().item()
    err = {}
    for name, on in (("exact", False), ("split", True)):
        monkeypatch.setattr(ops, "FP32_SPLIT", on)
        y = ops.conv2d(xd, pk, actc, res=rd)
        torch.cuda.synchronize()
        got = y.double().cpu()
        assert torch.isfinite(got).all(), name
        err[name] = ((got[..., :cout] - ref).abs().max().item() / scale, (got[..., :cout] - ref).abs().mean().item() / scale)
    print(f"[split16] {case}: exact max {err['exact'][0]:.2e} mean {err['exact'][1]:.2e}; split max {err['split'][0]:.2e} mean {err['split'][1]:.2e}")
    assert err["exact"][0] <= 2e-6
    assert err["split"][0] <= 4e-6 and err["split"][1] <= 4e-7, err


def test_split16_linear_rows():
    """the strided-row linear path (token projections of the fp32 mode)"""
    from gim_amd import _lib, ops
    from gim_amd.packing import pack_conv
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    R, K, N = 4800, 256, 768
    w = torch.randn(N, K, generator=g) * K ** -0.5
    x = torch.randn(R, K, generator=g)
    pk = pack_conv(w.view(N, K, 1, 1), None, _lib.GIM_F32, dev)   # a Linear is a 1x1 convolution over rows
    ref = x.double() @ w.double().t()
    out = {}
    for on in (False, True):
        ops.FP32_SPLIT = on
        try:
            y = torch.empty(R, pk.n_store, device=dev)
            ops.linear(x.to(dev), pk, y, ops.ACT_NONE, True)
            torch.cuda.synchronize()
            out[on] = (y.double().cpu()[:, :N] - ref).abs().max().item() / ref.abs().max().item()
        finally:
            ops.FP32_SPLIT = False
    print(f"[split16] linear {R}x{K}x{N}: exact {out[False]:.2e} split {out[True]:.2e}")
    assert out[False] <= 2e-6 and out[True] <= 4e-6
