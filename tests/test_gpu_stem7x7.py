"""gim_stem7x7 (round 4): conv1 7x7 / stride 2 / pad 3 + bn1 + relu of the backbone (backbone/resnet.py:306) on its own kernel,
through the C-ABI, against an fp64 convolution in torch:
  * the [hi | lo | 0] pixel layout of gim_nchw_to_nhwc_split (ld = 8), bit for bit;
  * plain operands: equal to the fp64 convolution of the ROUNDED operands up to the rounding of the 16-bit output (fp32 accumulate);
  * split operands: equal to the fp64 convolution of the fp32 operands up to the rounding of the 16-bit output;
  * both operand kinds, both output kinds (the bf16 mode reads an fp16 image and writes bf16), ragged sizes (odd H / W, tiles that
    hang over the border, images smaller than one 4 x 32 tile, more tiles than CUs) and the implicit-GEMM path the kernel replaces;
  * the module with and without the kernel: same matches."""
import pytest
import torch
import torch.nn.functional as F

from tools import synth_loftr as S

pytestmark = pytest.mark.gpu
KINDS = [torch.float16, torch.bfloat16]
SIZES = [(2, 64, 96), (1, 37, 45), (3, 5, 7), (1, 1, 1), (2, 480, 640), (1, 9, 300)]


def _layer(seed):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    bn = (torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1, torch.randn(64, generator=g) * 0.1,
          torch.rand(64, generator=g) + 0.5, 1e-5)
    return w, bn


def _ulp(td):
    """spacing of the 16-bit kind relative to a value at the bottom of its binade (round-to-nearest is off by at most half of it)"""
    return 2.0 ** -(10 if td == torch.float16 else 7)


@pytest.mark.parametrize("td", KINDS, ids=["fp16", "bf16"])
def test_split_pixel_layout(td):
    from gim_amd import ops
    g = torch.Generator().manual_seed(3)
    a, b = torch.rand(2, 3, 20, 24, generator=g), torch.rand(1, 3, 20, 24, generator=g) * 255.0
    out = torch.full((3, 20, 24, 8), 7.0, dtype=td, device="cuda")
    ops.nchw_to_nhwc_split(a.cuda(), out, 0)
    ops.nchw_to_nhwc_split(b.cuda(), out, 2)
    torch.cuda.synchronize()
    x = torch.cat([a, b]).permute(0, 2, 3, 1)
    hi = x.to(td)
    lo = (x - hi.float()).to(td)
    assert torch.equal(out.cpu(), torch.cat([hi, lo, torch.zeros(3, 20, 24, 2, dtype=td)], dim=-1))


def _run(x, w, bn, td, split, out_td):
    from gim_amd import _lib, ops
    from gim_amd.packing import pack_stem7x7
    dt = _lib.GIM_F16 if td == torch.float16 else _lib.GIM_BF16
    B, _, H, W = x.shape
    xi = torch.empty(B, H, W, 8, dtype=td, device="cuda")
    (ops.nchw_to_nhwc_split if split else ops.nchw_to_nhwc)(x.cuda(), xi, 0)
    y = ops.stem7x7(xi, pack_stem7x7(w, bn, dt, "cuda", split=split), out_dtype=out_td)
    torch.cuda.synchronize()
    return y


@pytest.mark.parametrize("B,H,W", SIZES, ids=[f"{b}x{h}x{w}" for b, h, w in SIZES])
@pytest.mark.parametrize("split", [True, False], ids=["split", "plain"])
@pytest.mark.parametrize("td", KINDS, ids=["fp16", "bf16"])
def test_stem7x7_against_fp64_convolution(td, split, B, H, W):
    from gim_amd.packing import fold_bn
    w, bn = _layer(8)
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(H * 1000 + W))
    wf, bf = fold_bn(w, bn)
    if split:
        ref = F.conv2d(x.double(), wf.double(), bf.double(), stride=2, padding=3)
    else:
        ref = F.conv2d(x.to(td).double(), wf.to(td).double(), bf.double(), stride=2, padding=3)
    ref = F.relu(ref).permute(0, 2, 3, 1)
    y = _run(x, w, bn, td, split, td)
    assert y.shape == ref.shape and y.dtype == td
    err = (y.double().cpu() - ref).abs()
    # the rounding of the output (half an ulp of the value) + the fp32 accumulation of 147 / 441 products + (split) the dropped lo x lo terms
    # (operands carried to eps^2 = 2^-22 / 2^-16 by the hi + lo pair: ~1e-6 / 6e-5 of the layer's scale)
    tol = 0.5 * _ulp(td) * ref.abs() * 1.01 + (_ulp(td) ** 2 + 2e-6) * ref.abs().max().clamp_min(1.0)
    assert bool((err <= tol).all()), f"max excess {(err - tol).max().item():.3e}"


def test_stem7x7_fp16_operands_bf16_output():
    """the bf16 mode's stem: fp16 image and filters, bf16 activations"""
    from gim_amd.packing import fold_bn
    w, bn = _layer(11)
    x = torch.rand(2, 3, 50, 70, generator=torch.Generator().manual_seed(12))
    wf, bf = fold_bn(w, bn)
    ref = F.relu(F.conv2d(x.double(), wf.double(), bf.double(), stride=2, padding=3)).permute(0, 2, 3, 1)
    y = _run(x, w, bn, torch.float16, True, torch.bfloat16)
    assert y.dtype == torch.bfloat16
    err = (y.double().cpu() - ref).abs()
    assert bool((err <= 0.5 * _ulp(torch.bfloat16) * ref.abs() * 1.01 + 3e-6 * ref.abs().max()).all())    # fp16 operands: 2^-22


@pytest.mark.parametrize("td", KINDS, ids=["fp16", "bf16"])
def test_stem7x7_equals_the_implicit_gemm_path(td):
    """same layer, same split operands, the kernel it replaces: fp32 sums in another order, one rounding to 16 bits"""
    from gim_amd import _lib, ops
    from gim_amd.packing import pack_conv_split
    dt = _lib.GIM_F16 if td == torch.float16 else _lib.GIM_BF16
    w, bn = _layer(13)
    x = torch.rand(2, 3, 96, 128, generator=torch.Generator().manual_seed(14))
    xs = torch.empty(2, 96, 128, 16, dtype=td, device="cuda")
    ops.nchw_to_nhwc_split(x.cuda(), xs, 0)
    ref = ops.conv2d(xs, pack_conv_split(w, bn, dt, "cuda", stride=2, pad=3), _lib.ACT_RELU, out_dtype=td)
    y = _run(x, w, bn, td, True, td)
    d = (y.float() - ref.float()).abs()
    assert bool((d <= _ulp(td) * 1.01 * ref.float().abs() + 1e-5).all())    # at most one 16-bit step apart
    assert (d > 0).float().mean().item() < 0.01                               # and almost everywhere identical


def test_module_with_and_without_the_stem_kernel():
    """both 16-bit modes, with the kernel and on the implicit-GEMM path it replaces: the same matches up to the stem's output rounding"""
    c0, c1 = S.textured_pairs(2, 192, 256, seed=3)
    for prec in ("fp16", "bf16"):
        keys = []
        for k in (True, False):
            m, _ = S.synthetic_model(prec, stem_kernel=k)
            m = m.to("cuda:0")
            assert m._stem_k() == k
            d = {"image0": c0[:, :1].cuda(), "image1": c1[:, :1].cuda(), "color0": c0.cuda(), "color1": c1.cuda()}
            for _ in range(3):          # eager, capture, replay
                m(d)
            torch.cuda.synchronize()
            keys.append(set(zip(d["b_ids"].tolist(), d["i_ids"].tolist(), d["j_ids"].tolist())))
        a, b = keys
        assert len(a) > 200 and len(a ^ b) <= max(2, len(a) // 50), (prec, len(a), len(b), len(a ^ b))
