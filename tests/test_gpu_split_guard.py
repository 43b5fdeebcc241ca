"""Round 4: the split-operand first convolution and the fp16 range guard of gim_loftr.

  * gim_nchw_to_nhwc_split against torch: channels [hi | lo | hi | 0], bit for bit, both 16-bit kinds;
  * the stem (conv1 7x7 / 2 + bn1 + relu, backbone/resnet.py:306) through the split layout against an fp64 convolution: 50 x closer
    than the plainly rounded operands, i.e. at the rounding of its 16-bit OUTPUT;
  * the health word of gim_coarse_match (count[1]): 0 on finite features, bit 0 on NaN / inf features, for the 256-tile statistics
    kernel, the 128-tile kernel and the fp32 path; bit 1 from gim_fine_fused_dev on non-finite fine maps, sticky across calls;
  * end to end: a checkpoint whose first layer overflows IEEE fp16 (conv1 scaled by 1e5) -- the fp16 module warns, switches itself
    to bf16 and returns FINITE outputs for that very batch (no NaN confidences, no silently empty match list)."""
import warnings

import pytest
import torch
import torch.nn.functional as F

import loftr_oracle as O
from tools import synth_loftr as S

pytestmark = pytest.mark.gpu
KINDS = [torch.float16, torch.bfloat16]


@pytest.mark.parametrize("C,ld", [(3, 16), (1, 8), (2, 8), (5, 16)], ids=["rgb", "gray", "c2-generic", "c5-generic"])
@pytest.mark.parametrize("td", KINDS, ids=["fp16", "bf16"])
def test_nchw_to_nhwc_split_layout(td, C, ld):
    from gim_amd import ops
    g = torch.Generator().manual_seed(3)
    a, b = torch.rand(2, C, 20, 24, generator=g), torch.rand(1, C, 20, 24, generator=g) * 255.0
    out = torch.full((3, 20, 24, ld), 7.0, dtype=td, device="cuda")
    ops.nchw_to_nhwc_split(a.cuda(), out, 0)
    ops.nchw_to_nhwc_split(b.cuda(), out, 2)
    torch.cuda.synchronize()
    x = torch.cat([a, b]).permute(0, 2, 3, 1)
    hi = x.to(td)
    lo = (x - hi.float()).to(td)
    ref = torch.cat([hi, lo, hi, torch.zeros(3, 20, 24, ld - 3 * C, dtype=td)], dim=-1)
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("td", KINDS, ids=["fp16", "bf16"])
def test_split_stem_is_exact_to_its_output_rounding(td):
    from gim_amd import _lib, ops
    from gim_amd.packing import cstore, fold_bn, pack_conv, pack_conv_split
    dt = _lib.GIM_F16 if td == torch.float16 else _lib.GIM_BF16
    g = torch.Generator().manual_seed(8)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    bn = (torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1, torch.randn(64, generator=g) * 0.1,
          torch.rand(64, generator=g) + 0.5, 1e-5)
    x = torch.rand(2, 3, 64, 96, generator=g)
    wf, bf = fold_bn(w, bn)
    ref = F.relu(F.conv2d(x.double(), wf.double(), bf.double(), stride=2, padding=3)).permute(0, 2, 3, 1)
    xs = torch.empty(2, 64, 96, 16, dtype=td, device="cuda")
    ops.nchw_to_nhwc_split(x.cuda(), xs, 0)
    xp = torch.empty(2, 64, 96, cstore(3, dt), dtype=td, device="cuda")
    ops.nchw_to_nhwc(x.cuda(), xp, 0)
    # fp32 output: the operand error alone
    ys = ops.conv2d(xs, pack_conv_split(w, bn, dt, "cuda", stride=2, pad=3), _lib.ACT_RELU, out_dtype=torch.float32)
    yp = ops.conv2d(xp, pack_conv(w, bn, dt, "cuda", stride=2, pad=3, cin_pad=cstore(3, dt)), _lib.ACT_RELU, out_dtype=torch.float32)
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    es = (ys[..., :64].double().cpu() - ref).abs().max().item() / scale
    ep = (yp[..., :64].double().cpu() - ref).abs().max().item() / scale
    print(f"stem {td}: split {es:.2e}  plain {ep:.2e} of scale")
    assert es < (3e-6 if td == torch.float16 else 2e-4) and ep > 30 * es, (es, ep)


def _planted(td, N=2, hw=(30, 40), seed=3):
    f0, f1, _ = O.planted_coarse_features(N, hw, sigma=1.0, eps=0.5, seed=seed)
    return (f0.to(td), f1.to(td)) if td is not None else (f0, f1)


@pytest.mark.parametrize("td", [torch.float16, torch.bfloat16, None], ids=["fp16", "bf16", "fp32"])
def test_coarse_health_word(td):
    from gim_amd import ops
    f0, f1 = _planted(td)
    r = ops.coarse_match(f0.cuda(), f1.cuda(), (30, 40), (30, 40), 8.0)
    c = r.count.cpu()
    assert int(c[0]) > 200 and int(c[1]) == 0 and int(c[2:].sum()) == int(c[0])
    for bad in (float("nan"), float("inf")):
        g0 = f0.clone()
        g0[1, 777, 5] = bad
        r = ops.coarse_match(g0.cuda(), f1.cuda(), (30, 40), (30, 40), 8.0)
        assert int(r.count[1]) & 1, (td, bad)
    # a caller-owned count buffer: bit 1 (the fine kernel's) survives the next coarse call, bit 0 is rewritten
    cnt = torch.zeros(2 + 2, dtype=torch.int32, device="cuda")
    cnt[1] = 3
    r = ops.coarse_match(f0.cuda(), f1.cuda(), (30, 40), (30, 40), 8.0, count=cnt)
    assert int(cnt[1]) == 2 and int(cnt[0]) > 200


def test_coarse_health_word_tile128_kernel():
    """the 128 x 128 statistics kernel (the one that serves padding masks) writes the same bit: all-valid masks route 16-bit features to it"""
    from gim_amd import ops
    f0, f1, _ = O.planted_coarse_features(1, (30, 40), sigma=1.0, eps=0.5, seed=3)
    f0, f1 = f0.half(), f1.half()
    m = torch.ones(1200, dtype=torch.uint8, device="cuda")
    r = ops.coarse_match(f0.cuda(), f1.cuda(), (30, 40), (30, 40), 8.0, mask0=m, mask1=m.clone())
    assert int(r.count[1]) == 0 and int(r.count[0]) > 200
    f0[0, 5, 5] = float("inf")
    r = ops.coarse_match(f0.cuda(), f1.cuda(), (30, 40), (30, 40), 8.0, mask0=m, mask1=m.clone())
    assert int(r.count[1]) & 1


def test_fine_kernel_sets_sticky_health_bit():
    model, _ = S.synthetic_model("fp16")
    model = model.to("cuda:0")
    g = torch.Generator().manual_seed(1)
    hc, wc, M = 12, 16, 9
    f0 = torch.randn(1, 4 * hc, 4 * wc, 128, generator=g).half().cuda()
    f1 = torch.randn(1, 4 * hc, 4 * wc, 128, generator=g).half().cuda()
    b = torch.zeros(M, dtype=torch.int64).cuda()
    i = torch.arange(20, 20 + M).cuda()
    j = torch.arange(40, 40 + M).cuda()
    mk = torch.stack([(j % wc).float() * 8, (j // wc).float() * 8], 1)
    cnt = torch.zeros(3, dtype=torch.int32, device="cuda")
    cnt[0] = M
    e, k, _, _ = model._fine_level(f0, f1, b, i, j, mk, None, False, (hc, wc), (hc, wc), (96, 128), True, count=cnt)
    torch.cuda.synchronize()
    assert int(cnt[1]) == 0 and torch.isfinite(e[:M]).all()
    f1[0, 10, 13, 7] = float("inf")       # inside the window of match 0 (j = 40: cell (2, 8) -> fine rows 6..10, columns 30..34)? any window
    f1[0, :, :, 3] = float("inf")         # every window sees it
    model._fine_level(f0, f1, b, i, j, mk, None, False, (hc, wc), (hc, wc), (96, 128), True, count=cnt)
    torch.cuda.synchronize()
    assert int(cnt[1]) & 2


def _fwd(model, c0, c1):
    d = {"image0": c0[:, :1].cuda(), "image1": c1[:, :1].cuda(), "color0": c0.cuda(), "color1": c1.cuda()}
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        model(d)
        torch.cuda.synchronize()
    return d, [str(w.message) for w in rec]


@pytest.mark.parametrize("where", ["stream", "weights"])
def test_fp16_overflow_falls_back_to_bf16_with_finite_outputs(where):
    """`stream`: the second Bottleneck's bn3 shift raised by 1e5 -- the residual stream leaves the fp16 range while every weight stays
    inside (the downstream ReLUs would scrub the NaNs: the range guard of the storing kernel has to see it).  `weights`: conv1
    scaled by 1e7 -- the folded filters themselves are beyond 65504 (caught when they are packed)."""
    model, sd = S.synthetic_model("fp16")
    sd = {k: v.clone() for k, v in sd.items()}
    if where == "stream":
        sd["backbone.encode.layer1.1.bn3.bias"] = sd["backbone.encode.layer1.1.bn3.bias"] + 1e5
    else:
        sd["backbone.encode.conv1.weight"] = sd["backbone.encode.conv1.weight"] * 1e7
    model.load_state_dict(sd)
    model = model.to("cuda:0")
    c0, c1 = S.textured_pairs(2, 192, 256, seed=3)
    d, msgs = _fwd(model, c0, c1)
    assert any("fp16" in m and "bf16" in m for m in msgs), msgs
    if where == "stream":
        assert any("65504" in m for m in msgs), msgs
    assert model.precision == "bf16" and model.fp16_overflowed
    for k in ("mconf", "mkpts0_f", "mkpts1_f", "expec_f"):
        assert torch.isfinite(d[k]).all(), k
    # the same batch straight in bf16 gives the same answer: the guard re-ran it, it did not patch it up
    m2, _ = S.synthetic_model("bf16")
    m2.load_state_dict(sd)
    m2 = m2.to("cuda:0")
    d2, msgs2 = _fwd(m2, c0, c1)
    assert not msgs2, msgs2
    for k in ("b_ids", "i_ids", "j_ids", "mconf", "mkpts1_f"):
        assert torch.equal(d[k], d2[k]), k


@pytest.mark.parametrize("shape,block", [((1, 480, 640), "layer3.2"), ((1, 200, 264), "layer2.1"), ((1, 200, 264), "layer1.1")],
                         ids=["one-640x480-pair-layer3", "odd-size-layer2", "odd-size-layer1"])
def test_fp16_overflow_is_seen_on_the_unfused_residual_path(shape, block):
    """where a Bottleneck does NOT take a fused kernel -- layer 3 of ONE 640x480 pair has 2 x 60 x 80 = 9 600 rows, not a
    multiple of 256; a 200 x 264 image fails layer 1's H % 8 / W % 32 and the tails' row counts -- conv3 + identity + relu is the implicit-GEMM
    kernel's residual epilogue, which now carries the same range check (gim_conv_args.health).  An overflowing residual stream there must
    trip the guard, not return silently wrong matches."""
    n, H, W = shape
    model, sd = S.synthetic_model("fp16")
    sd = {k: v.clone() for k, v in sd.items()}
    key = f"backbone.encode.{block}.bn3.bias"
    sd[key] = sd[key] + 1e5
    model.load_state_dict(sd)
    model = model.to("cuda:0")
    c0, c1 = S.textured_pairs(n, H, W, seed=5)
    d, msgs = _fwd(model, c0, c1)
    assert any("65504" in m for m in msgs), msgs
    assert model.precision == "bf16" and model.fp16_overflowed
    for k in ("mconf", "mkpts0_f", "mkpts1_f"):
        assert torch.isfinite(d[k]).all(), k


def test_fallback_keeps_a_callers_fp32_similarity():
    """the fp16 -> bf16 fallback must not silently replace coarse_sim='fp32' by the 16-bit similarity"""
    model, sd = S.synthetic_model("fp16")
    model.coarse_sim = "fp32"
    sd = {k: v.clone() for k, v in sd.items()}
    sd["backbone.encode.layer1.1.bn3.bias"] = sd["backbone.encode.layer1.1.bn3.bias"] + 1e5
    model.load_state_dict(sd)
    model = model.to("cuda:0")
    c0, c1 = S.textured_pairs(2, 192, 256, seed=3)
    _fwd(model, c0, c1)
    assert model.precision == "bf16" and model.coarse_sim == "fp32"


def test_healthy_checkpoint_never_trips_the_guard():
    m3, _ = S.synthetic_model("fp16")
    m3 = m3.to("cuda:0")
    c0, c1 = S.textured_pairs(2, 192, 256, seed=3)
    for _ in range(4):   # eager, capture, replays
        d3, msgs = _fwd(m3, c0, c1)
        assert not msgs, msgs
    assert m3.precision == "fp16" and not m3.fp16_overflowed and d3["b_ids"].numel() > 200
