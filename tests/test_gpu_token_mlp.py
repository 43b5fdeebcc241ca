"""gim_token_mlp (merge -> norm1 -> mlp.0 -> relu -> mlp.2 -> norm2 -> residual of a coarse LoFTREncoderLayer in one kernel,
transformer.py:52-58) against a plain torch fp32 restatement with the kernel's rounding points (bf16 operands, fp32 accumulation)
and against the unfused launch sequence, incl. a row count that is not a multiple of the 64-row workgroup tile and strided
row views (the operand copy of x lives in the [x | msg] concat buffer)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _layer(seed):
    from gim_amd.loftr.loftr import _EncoderLayer
    torch.manual_seed(seed)
    layer = _EncoderLayer(256, 8)
    with torch.no_grad():
        for p in layer.parameters():
            if p.dim() > 1:
                torch.nn.init.xavier_uniform_(p)
        for ln in (layer.norm1, layer.norm2):
            ln.weight.copy_(0.75 + 0.5 * torch.rand(256))
            ln.bias.copy_(0.1 * torch.randn(256))
    return layer


KINDS = [torch.bfloat16, torch.float16]   # the two 16-bit flavours of the kernel (csrc/gim_common.h)
KIDS = ["bf16", "fp16"]


def _reference(layer, msg, x32, tdt=torch.bfloat16):
    bf = lambda t: t.to(tdt).float()  # noqa: E731
    m = bf(msg) @ bf(layer.merge.weight).T
    m = bf(F.layer_norm(m, (256,), layer.norm1.weight, layer.norm1.bias, layer.norm1.eps))
    h = bf(F.relu(torch.cat([bf(x32), m], 1) @ bf(layer.mlp[0].weight).T))
    o = h @ bf(layer.mlp[2].weight).T
    return x32 + F.layer_norm(o, (256,), layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)


@pytest.mark.parametrize("tdt", KINDS, ids=KIDS)
@pytest.mark.parametrize("R", [64, 200, 4800 * 2 + 7])
def test_token_mlp_matches_reference(R, tdt):
    from gim_amd import ops
    from gim_amd.packing import pack_token_mlp
    layer = _layer(R)
    g = torch.Generator().manual_seed(R)
    msg = (0.5 * torch.randn(R, 256, generator=g)).to(tdt)
    x32 = torch.randn(R, 256, generator=g) * 2.0
    with torch.no_grad():
        ref = _reference(layer, msg.float(), x32, tdt)
    wts, ln, eps = pack_token_mlp(layer, "cuda", tdt)
    cat = torch.zeros(R, 512, dtype=tdt, device="cuda")      # [x | msg] buffer of the engine: x in the left half
    cat[:, :256] = x32.cuda().to(tdt)
    cat[:, 256:] = 7.0                                                   # must stay untouched
    xd = x32.cuda().clone()
    ops.token_mlp(msg.cuda(), cat[:, :256], xd, wts, ln, eps)
    torch.cuda.synchronize()
    got = xd.cpu()
    err = (got - ref).abs()
    k = 1.0 if tdt == torch.bfloat16 else 0.25   # fp16: three more significand bits at every rounding point
    assert err.max() < 3e-2 * k and err.mean() < 2e-3 * k, (err.max().item(), err.mean().item())
    assert torch.equal(cat[:, :256].cpu(), got.to(tdt))       # operand copy of the new x
    assert bool((cat[:, 256:] == 7.0).all())


def test_coarse_transformer_fused_vs_unfused_vs_oracle():
    """8 coarse layers with and without the fused tail on the same tokens; both against the fp32 oracle."""
    import loftr_oracle as O
    from tools import synth_loftr as S
    model, sd = S.synthetic_model("bf16")
    model = model.cuda()
    c0, c1 = S.textured_pairs(2, 96, 128, seed=2)
    outs = {}
    for fused in (True, False):
        model.token_fused = fused
        model.debug = {}
        d = {"image0": c0[:, :1].cuda(), "image1": c1[:, :1].cuda(), "color0": c0.cuda(), "color1": c1.cuda()}
        model(d)
        outs[fused] = (model.debug["feat_c0"].float().cpu(), model.debug["feat_c1"].float().cpu())
        model.debug = None
    model.token_fused = True
    with torch.no_grad():
        fc, _ = O.backbone(sd, torch.cat([c0, c1], 0))
        pe = O.position_encoding(256, 12, 16)
        t = (fc + pe).flatten(2).transpose(1, 2)
        r0, r1 = O.local_feature_transformer(sd, "loftr_coarse", t[:2], t[2:], 8, 4)
    scale = r0.abs().max().item()
    for k, ref in ((0, r0), (1, r1)):
        ef = (outs[True][k] - ref).abs().mean().item() / scale
        eu = (outs[False][k] - ref).abs().mean().item() / scale
        assert ef < 1.5 * eu + 1e-3, (k, ef, eu)     # the fused tail is no further from the oracle than the unfused one
        assert (outs[True][k] - outs[False][k]).abs().mean().item() / scale < 2 * eu + 1e-3


@pytest.mark.parametrize("masked", [False, True])
def test_token_mlp_with_fused_attention_apply(masked):
    """kv != NULL: queries in, the apply step of the linear attention (attentions.py:44-45) runs in the kernel's prologue; against the
    two-launch sequence (gim_linear_attention + gim_token_mlp) and the torch restatement, with and without padding masks."""
    from gim_amd import ops
    from gim_amd.packing import pack_token_mlp
    import loftr_oracle as O
    nb, L, S, H, C = 3, 128, 192, 8, 256
    layer = _layer(5)
    g = torch.Generator().manual_seed(5)
    bf = lambda t: t.to(torch.bfloat16)  # noqa: E731
    q = bf(torch.nn.functional.elu(torch.randn(nb * L, C, generator=g)) + 1)
    k = bf(torch.nn.functional.elu(torch.randn(nb * S, C, generator=g)) + 1)
    v = bf(torch.randn(nb * S, C, generator=g))
    x32 = torch.randn(nb * L, C, generator=g) * 2.0
    qm = km = None
    if masked:
        qm = (torch.rand(nb * L, generator=g) > 0.2).to(torch.uint8)
        km = (torch.rand(nb * S, generator=g) > 0.2).to(torch.uint8)
    wts, ln, eps = pack_token_mlp(layer, "cuda")
    dev = "cuda"
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    qmd, kmd = (qm.to(dev), km.to(dev)) if masked else (None, None)
    # reference sequence: attention output, then the unfused-prologue kernel
    msg = torch.empty(nb * L, C, dtype=torch.bfloat16, device=dev)
    ops.linear_attention(qd, kd, vd, msg, nb, L, nb, S, H, None, qmd, kmd)
    xb_a = x32.to(dev).to(torch.bfloat16).contiguous()
    xa = x32.to(dev).clone()
    ops.token_mlp(msg, xb_a, xa, wts, ln, eps)
    # fused
    ws, _ = ops.linear_attention_state(kd, vd, nb, S, H, None, kmd)
    xb_b = x32.to(dev).to(torch.bfloat16).contiguous()
    xbf = x32.to(dev).clone()
    ops.token_mlp(qd, xb_b, xbf, wts, ln, eps, kv=ws, L=L, S=S, q_mask=qmd)
    torch.cuda.synchronize()
    err = (xbf - xa).abs()
    assert err.max() < 5e-2 and err.mean() < 3e-3, (err.max().item(), err.mean().item())
    # torch restatement of the whole thing
    with torch.no_grad():
        m = O.linear_attention(torch.zeros(0), torch.zeros(0), torch.zeros(0)) if False else None
        Q, K, V = q.float().view(nb, L, H, 32), k.float().view(nb, S, H, 32), v.float().view(nb, S, H, 32)
        if masked:
            Q = Q * qm.view(nb, L, 1, 1)
            K = K * km.view(nb, S, 1, 1)
            V = V * km.view(nb, S, 1, 1)
        KV = torch.einsum("nshd,nshv->nhdv", K, V / S)
        Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(1)) + 1e-6)
        att = (torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * S).reshape(nb * L, C)
        ref = _reference(layer, att, x32)
    err = (xbf.cpu() - ref).abs()
    assert err.max() < 5e-2 and err.mean() < 3e-3, (err.max().item(), err.mean().item())


@pytest.mark.parametrize("tdt", KINDS, ids=KIDS)
@pytest.mark.parametrize("R", [128, 64 * 7 + 9])
def test_token_mlp_projection_blocks(R, tdt):
    """gim_token_mlp_emit: act(x_new W_b^T) of the new rows for up to six [256 x 256] projections (the next layer's q / k / v,
    transformer.py:42-44, elu+1 on q and k: attentions.py:31-32), row-gated per block, into strided column views -- against fp32
    products of the 16-bit x the kernel wrote; the token update itself must not change."""
    from gim_amd import ops
    from gim_amd._lib import ACT_ELU1, ACT_NONE
    from gim_amd.packing import pack_token_emit, pack_token_mlp
    layer = _layer(R)
    g = torch.Generator().manual_seed(R + 1)
    msg = (0.5 * torch.randn(R, 256, generator=g)).to(tdt)
    x32 = torch.randn(R, 256, generator=g) * 2.0
    ws = [torch.randn(256, 256, generator=g) / 16 for _ in range(5)]
    wts, ln, eps = pack_token_mlp(layer, "cuda", tdt)
    ew = pack_token_emit(ws, "cuda", tdt)

    def run(emit):
        cat = torch.zeros(R, 512, dtype=tdt, device="cuda")
        cat[:, :256] = x32.cuda().to(tdt)
        xd = x32.cuda().clone()
        ops.token_mlp(msg.cuda(), cat[:, :256], xd, wts, ln, eps, emit=emit)
        torch.cuda.synchronize()
        return cat, xd

    qa = torch.full((R, 768), 5.0, dtype=tdt, device="cuda")
    qb = torch.full((R, 768), 5.0, dtype=tdt, device="cuda")
    lo = 64 if R > 128 else 0
    spec = [(qa[:, :256], ACT_ELU1, 0, R), (qa[:, 256:512], ACT_ELU1, lo, R), (qa[:, 512:], ACT_NONE, lo, R),
            (qb[:, :256], ACT_ELU1, 0, 64), (qb[:, 512:], ACT_NONE, 0, R)]
    cat0, x0 = run(None)
    cat1, x1 = run((ew, spec))
    assert torch.equal(x0, x1) and torch.equal(cat0, cat1)
    xn = cat1[:, :256].float().cpu()
    r16 = lambda t: t.to(tdt).float()  # noqa: E731
    tol = 4e-2 if tdt == torch.bfloat16 else 6e-3
    for (out, act, a, b), w in zip(spec, ws):
        ref = xn @ r16(w).T
        if act == ACT_ELU1:
            ref = F.elu(ref) + 1
        got = out.float().cpu()
        hi = min(R, (b + 63) // 64 * 64)   # whole tiles
        err = (got[a:hi] - ref[a:hi]).abs()
        assert err.max() < tol * max(1.0, ref.abs().max().item()) and err.mean() < tol / 8, (err.max().item(), err.mean().item())
        assert bool((got[:a] == 5.0).all()) and bool((got[hi:] == 5.0).all())    # rows outside the block's range are untouched
    assert bool((qb[:, 256:512] == 5.0).all())


def test_coarse_transformer_emitted_projections_vs_gemms():
    """The coarse transformer with the projections emitted by the token tails against the same engine with projection GEMMs: same
    operands and products, only the accumulation order inside a 256-long dot product differs."""
    from tools import synth_loftr as S
    model, sd = S.synthetic_model("fp16")
    model = model.cuda()
    c0, c1 = S.textured_pairs(2, 256, 256, seed=4)   # 32 x 32 coarse tokens: a multiple of the 64-row tile
    outs = {}
    for emit in (True, False):
        model.token_emit = emit
        model.debug = {}
        d = {"image0": c0[:, :1].cuda(), "image1": c1[:, :1].cuda(), "color0": c0.cuda(), "color1": c1.cuda()}
        model(d)
        outs[emit] = (model.debug["feat_c0"].float().cpu(), model.debug["feat_c1"].float().cpu(), d["mconf"].numel())
        model.debug = None
    model.token_emit = True
    for k in (0, 1):
        scale = outs[False][k].abs().max().item()
        e = (outs[True][k] - outs[False][k]).abs()
        assert e.max().item() < 2e-2 * scale and e.mean().item() < 1e-3 * scale, (k, e.max().item(), e.mean().item(), scale)
    assert outs[True][2] > 0 and abs(outs[True][2] - outs[False][2]) <= 0.05 * outs[False][2] + 2


def test_emitted_projections_unequal_image_sizes():
    """image0 256 x 256 (1024 coarse tokens) against image1 256 x 384 (1536): the self layers run per side, the plan of
    _emit_plan(same_len=False) -- against the projection-GEMM path on the same inputs"""
    from tools import synth_loftr as S
    model, sd = S.synthetic_model("fp16")
    model = model.cuda()
    c0, _ = S.textured_pairs(2, 256, 256, seed=5)
    c1, _ = S.textured_pairs(2, 256, 384, seed=6)
    outs = {}
    for emit in (True, False):
        model.token_emit = emit
        model.debug = {}
        d = {"image0": c0[:, :1].cuda(), "image1": c1[:, :1].cuda(), "color0": c0.cuda(), "color1": c1.cuda()}
        model(d)
        outs[emit] = (model.debug["feat_c0"].float().cpu(), model.debug["feat_c1"].float().cpu())
        model.debug = None
    model.token_emit = True
    assert outs[True][0].shape[1] == 1024 and outs[True][1].shape[1] == 1536
    for k in (0, 1):
        scale = outs[False][k].abs().max().item()
        e = (outs[True][k] - outs[False][k]).abs()
        assert e.max().item() < 2e-2 * scale and e.mean().item() < 1e-3 * scale, (k, e.max().item(), e.mean().item(), scale)


def test_emitted_projections_with_padding_masks():
    """padding masks (mask0 / mask1, coarse resolution) on the 16-bit path with the projections emitted by the token tails: the masks act
    in the attention state (kv mask) and in the apply step (query mask), the projections are computed for every token as in the
    reference -- against the projection-GEMM path on the same inputs"""
    from tools import synth_loftr as S
    model, sd = S.synthetic_model("fp16")
    model = model.cuda()
    c0, c1 = S.textured_pairs(2, 256, 256, seed=8)
    m0 = torch.zeros(2, 32, 32, dtype=torch.bool); m1 = torch.zeros(2, 32, 32, dtype=torch.bool)
    for b, ((h0, w0), (h1, w1)) in enumerate([((32, 28), (30, 32)), ((24, 32), (32, 32))]):
        m0[b, :h0, :w0] = True; m1[b, :h1, :w1] = True
    outs = {}
    for emit in (True, False):
        model.token_emit = emit
        model.debug = {}
        d = {"image0": c0[:, :1].cuda(), "image1": c1[:, :1].cuda(), "color0": c0.cuda(), "color1": c1.cuda(),
             "mask0": m0.cuda(), "mask1": m1.cuda()}
        model(d)
        outs[emit] = (model.debug["feat_c0"].float().cpu(), model.debug["feat_c1"].float().cpu(), d["b_ids"].numel())
        model.debug = None
    model.token_emit = True
    for k in (0, 1):
        scale = outs[False][k].abs().max().item()
        e = (outs[True][k] - outs[False][k]).abs()
        assert torch.isfinite(outs[True][k]).all()
        assert e.max().item() < 2e-2 * scale and e.mean().item() < 1e-3 * scale, (k, e.max().item(), e.mean().item(), scale)
    assert abs(outs[True][2] - outs[False][2]) <= 0.05 * outs[False][2] + 2


@pytest.mark.parametrize("tdt", KINDS, ids=KIDS)
def test_token_mlp_fused_kv_state(tdt):
    """Fused KV state (round 5): a (k, v) projection pair handed over as per-tile partial states K^T V / S, K^T 1 (attentions.py:38-43)
    instead of rows -- against the row path on the same launch inputs: k / v rows emitted, then gim_linear_attention_kv.  Two producers
    fill one consumer's workspace (4 sequences of 128 rows: sequences 0-1 from the first call, 2-3 from the second, whose pair is row-gated
    to the last two of its three sequences); a plain q block rides along."""
    from gim_amd import ops
    from gim_amd._lib import ACT_ELU1, ACT_NONE
    from gim_amd.packing import pack_token_emit, pack_token_mlp
    L, H = 128, 8
    layer = _layer(11)
    g = torch.Generator().manual_seed(12)
    wq, wk, wv = (torch.randn(256, 256, generator=g) / 16 for _ in range(3))
    wts, ln, eps = pack_token_mlp(layer, "cuda", tdt)
    ew = pack_token_emit([wq, wk, wv], "cuda", tdt)
    launches = [(2 * L, 0, 0), (3 * L, L, 2 * (L // 64))]      # (rows, row_lo of the k / v blocks, consumer-relative first tile)
    ins = [((0.5 * torch.randn(R, 256, generator=g)).to(tdt).cuda(), (torch.randn(R, 256, generator=g) * 2.0).cuda()) for R, _, _ in launches]

    def run(fused, ws):
        outs = []
        for (R, lo, tile0), (msg, x32) in zip(launches, ins):
            cat = torch.zeros(R, 512, dtype=tdt, device="cuda")
            cat[:, :256] = x32.to(tdt)
            xd = x32.clone()
            qkv = torch.full((R, 768), 5.0, dtype=tdt, device="cuda")
            if fused:
                spec = [(qkv[:, :256], ACT_ELU1, 0, R), (None, ACT_ELU1, lo, R, (ws, 4, L // 64, tile0, L)), (None, ACT_NONE, lo, R)]
            else:
                spec = [(qkv[:, :256], ACT_ELU1, 0, R), (qkv[:, 256:512], ACT_ELU1, lo, R), (qkv[:, 512:], ACT_NONE, lo, R)]
            ops.token_mlp(msg, cat[:, :256], xd, wts, ln, eps, emit=(ew, spec))
            outs.append((xd, cat, qkv))
        torch.cuda.synchronize()
        return outs

    rows = run(False, None)
    ws = ops.kv_state_workspace(4, L // 64, "cuda")
    ws.fill_(float("nan"))
    fus = run(True, ws)
    for (x0, c0, q0), (x1, c1, q1) in zip(rows, fus):
        assert torch.equal(x0, x1) and torch.equal(c0, c1) and torch.equal(q0[:, :256], q1[:, :256])
        assert bool((q1[:, 256:] == 5.0).all())                         # the fused pair writes no rows
    ops.kv_state_finalize(ws, 4, L // 64)
    k = torch.cat([rows[0][2][:, 256:512], rows[1][2][L:, 256:512]]).contiguous()
    v = torch.cat([rows[0][2][:, 512:], rows[1][2][L:, 512:]]).contiguous()
    ref, _ = ops.linear_attention_state(k, v, 4, L, H)
    torch.cuda.synchronize()
    n = 4 * H * (32 * 32 + 32)
    got, want = ws[:n].cpu().view(4, H, 1056), ref[:n].cpu().view(4, H, 1056)
    assert torch.isfinite(got).all()
    scale = want.abs().max().item()
    err = (got - want).abs().max().item()
    assert err <= 2e-6 * scale + 1e-6, (err, scale)                      # same 16-bit products, fp32 sums in another order
    # ... and against the definition on the emitted rows
    kf, vf = k.float().cpu().view(4, L, H, 32), v.float().cpu().view(4, L, H, 32)
    kv = torch.einsum("nshd,nshv->nhdv", kf.double(), vf.double() / L).float()
    assert (got[..., :1024].view(4, H, 32, 32) - kv).abs().max().item() <= 1e-5 * scale
    assert (got[..., 1024:] - kf.double().sum(1).float()).abs().max().item() <= 1e-5 * kf.sum(1).abs().max().item()


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
def test_coarse_transformer_fused_kv_state_vs_row_path(precision):
    """the whole forward with the k / v rows handed over as partial states (`kv_fused`, default) and on the row path (k / v rows + la_kv launches),
    EACH AGAINST THE fp32 CPU ORACLE (round 6: until round 5 the two engine paths were compared with each other at "match lists agree to 1 %",
    an engine-vs-engine bound with a hole; the state itself is pinned at 2e-6 by test_token_mlp_fused_kv_state).  Bounds = the full-size
    table of tests/test_gpu_loftr_fullsize.py for the mode (index flip rate, mean |d mconf|), over the four pairs together, + 2 flips for the
    granularity of ~10^3 matches; and the fused path may not sit further from the oracle than the row path does (x 1.5 + noise floor)."""
    import loftr_oracle as LO
    from tools import synth_loftr as S
    from tools.parity import parity_vs_oracle
    model, sd = S.synthetic_model(precision)
    model = model.cuda()
    c0, c1 = S.textured_pairs(4, 256, 320, seed=21, frac=0.5)
    with torch.no_grad():
        ref = LO.loftr_forward(sd, {"image0": c0[:, :1], "image1": c1[:, :1], "color0": c0, "color1": c1})
    c0, c1 = c0.cuda(), c1.cuda()
    max_flip, max_dconf = {"fp16": (0.004, 0.0025), "bf16": (0.026, 0.018)}[precision]
    dev = {}
    for fused in (True, False):
        model.kv_fused = fused
        model._invalidate()
        for _ in range(2):   # eager, then the captured graph
            d = {"image0": c0[:, :1], "image1": c1[:, :1], "color0": c0, "color1": c1}
            model(d)
        torch.cuda.synchronize()
        ps = [parity_vs_oracle(d, ref, b, b) for b in range(4)]
        n_ref = sum(p["oracle_matches"] for p in ps)
        flips = sum(round(p["flip_rate"] * max(1, p["oracle_matches"])) for p in ps)
        dconf = sum(p.get("mean_abs_dmconf", 0.0) * p["common"] for p in ps) / max(1, sum(p["common"] for p in ps))
        dpx = max(p.get("max_abs_dmkpts1_px", 0.0) for p in ps)
        print(precision, "kv_fused", fused, "vs oracle:", n_ref, "matches,", flips, "flips, mean |d mconf|", round(dconf, 5), "max |d mkpts1|", dpx)
        assert n_ref >= 200, n_ref
        assert flips <= max_flip * n_ref + 2, (fused, flips, n_ref)
        assert dconf <= max_dconf and dpx <= 1.0, (fused, dconf, dpx)
        dev[fused] = (flips, dconf)
    model.kv_fused = True
    assert dev[True][0] <= 1.5 * dev[False][0] + 3 and dev[True][1] <= 1.5 * dev[False][1] + 1e-4, dev


@pytest.mark.parametrize("masked", [False, True], ids=["nomask", "qmask"])
@pytest.mark.parametrize("tdt", KINDS, ids=KIDS)
def test_token_mlp_local_queries_are_bit_identical_to_emitted_ones(tdt, masked):
    """Local queries (round 5): a call that projects its own q rows from the operand copy of x (`q_weights`) against the same call fed with
    the q rows the PREVIOUS tail emitted for it -- same units, same MFMA order, same elu + 1: every output bit agrees."""
    from gim_amd import ops
    from gim_amd._lib import ACT_ELU1
    from gim_amd.packing import pack_token_emit, pack_token_mlp
    nb, L, H = 3, 128, 8
    R = nb * L
    g = torch.Generator().manual_seed(31)
    la, lb = _layer(32), _layer(33)
    wq = torch.randn(256, 256, generator=g) / 16
    qw = pack_token_emit([wq], "cuda", tdt)
    wa, lna, epsa = pack_token_mlp(la, "cuda", tdt)
    wb, lnb, epsb = pack_token_mlp(lb, "cuda", tdt)
    msg = (0.5 * torch.randn(R, 256, generator=g)).to(tdt).cuda()
    x32 = (torch.randn(R, 256, generator=g) * 2.0).cuda()
    cat = torch.zeros(R, 512, dtype=tdt, device="cuda")
    cat[:, :256] = x32.to(tdt)
    qrows = torch.zeros(R, 256, dtype=tdt, device="cuda")
    ops.token_mlp(msg, cat[:, :256], x32, wa, lna, epsa, emit=(qw, [(qrows, ACT_ELU1, 0, R)]))     # the updating tail: new x + its q rows
    k = (torch.rand(R, 256, generator=g) + 0.5).to(tdt).cuda()
    v = torch.randn(R, 256, generator=g).to(tdt).cuda()
    kv, _ = ops.linear_attention_state(k, v, nb, L, H)
    qm = None
    if masked:
        qm = (torch.rand(R, generator=g) > 0.2).to(torch.uint8).cuda()
    outs = []
    for local in (False, True):
        c, x = cat.clone(), x32.clone()
        ops.token_mlp(None if local else qrows, c[:, :256], x, wb, lnb, epsb, kv=kv, L=L, S=L, q_mask=qm, q_weights=qw if local else None)
        torch.cuda.synchronize()
        outs.append((c, x))
    assert torch.isfinite(outs[0][1]).all() and not torch.equal(outs[0][1], x32)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
def test_coarse_transformer_local_queries_vs_emitted_queries(precision):
    """the whole forward with local queries (`q_local`, default) against the emitted-q path: identical from the second layer on, the first
    layer's q rows come from the token kernel instead of the initial projection GEMM (another accumulation order inside K = 256)"""
    from tools import synth_loftr as S
    model, _ = S.synthetic_model(precision)
    model = model.cuda()
    c0, c1 = S.textured_pairs(4, 256, 320, seed=22, frac=0.5)
    c0, c1 = c0.cuda(), c1.cuda()
    outs = {}
    for ql in (True, False):
        model.q_local = ql
        model._invalidate()
        for _ in range(2):
            d = {"image0": c0[:, :1], "image1": c1[:, :1], "color0": c0, "color1": c1}
            model(d)
        torch.cuda.synchronize()
        outs[ql] = {k: d[k].cpu() for k in ("b_ids", "i_ids", "j_ids", "mconf")}
    model.q_local = True
    key = lambda o: set(zip(o["b_ids"].tolist(), o["i_ids"].tolist(), o["j_ids"].tolist()))   # noqa: E731
    a, b = key(outs[True]), key(outs[False])
    assert len(b) >= 200 and len(a ^ b) <= 0.01 * len(b) + 2, (len(a), len(b), len(a ^ b))


@pytest.mark.parametrize("tdt", KINDS, ids=KIDS)
def test_token_project_first_layer_state(tdt):
    """Projection only (round 5, `kv_init`): the first layer's (k, v) pair of the initial tokens as partial KV states straight from the rows --
    against the row path it replaces: [k | v] projection GEMM with elu + 1 on k (gim_conv2d_bn_act), then gim_linear_attention_kv."""
    from gim_amd import ops
    from gim_amd._lib import ACT_ELU1, ACT_NONE
    from gim_amd.packing import pack_conv, pack_token_emit
    from gim_amd._lib import GIM_BF16, GIM_F16
    nb, L, H = 5, 192, 8
    R = nb * L
    g = torch.Generator().manual_seed(41)
    wk, wv = (torch.randn(256, 256, generator=g) / 16 for _ in range(2))
    cat = torch.zeros(R, 512, dtype=tdt, device="cuda")                      # the engine's [x | msg] buffer: x rows are a strided view
    cat[:, :256] = torch.randn(R, 256, generator=g).to(tdt).cuda()
    ew = pack_token_emit([wk, wv], "cuda", tdt)
    ws = ops.kv_state_workspace(nb, L // 64, "cuda")
    ws.fill_(float("nan"))
    ops.token_project(cat[:, :256], (ew, [(None, ACT_ELU1, 0, R, (ws, nb, L // 64, 0, L)), (None, ACT_NONE, 0, R)]))
    ops.kv_state_finalize(ws, nb, L // 64)
    kvrows = torch.empty(R, 512, dtype=tdt, device="cuda")
    ops.linear(cat[:, :256], pack_conv(torch.cat([wk, wv], 0), None, GIM_F16 if tdt == torch.float16 else GIM_BF16, "cuda"), kvrows, ACT_ELU1, True, act_cols=256)
    ref, _ = ops.linear_attention_state(kvrows[:, :256], kvrows[:, 256:], nb, L, H)
    torch.cuda.synchronize()
    n = nb * H * (32 * 32 + 32)
    got, want = ws[:n].cpu(), ref[:n].cpu()
    assert torch.isfinite(got).all()
    scale = want.abs().max().item()
    # the rows differ in the last 16-bit digit here and there (another accumulation order inside K = 256 before the rounding), the sums follow
    tol = 2e-3 if tdt == torch.bfloat16 else 3e-4
    assert (got - want).abs().max().item() <= tol * scale, ((got - want).abs().max().item(), scale)
    assert (got - want).abs().mean().item() <= tol / 8 * scale


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
def test_coarse_transformer_first_layer_state_vs_projection_gemm(precision):
    """the whole forward with the first layer's k / v handed over as partial states by the projection-only token kernel (`kv_init`, default)
    against the projection GEMM + la_kv launches"""
    from tools import synth_loftr as S
    model, _ = S.synthetic_model(precision)
    model = model.cuda()
    c0, c1 = S.textured_pairs(4, 256, 320, seed=23, frac=0.5)
    c0, c1 = c0.cuda(), c1.cuda()
    outs = {}
    for ki in (True, False):
        model.kv_init = ki
        model._invalidate()
        for _ in range(2):
            d = {"image0": c0[:, :1], "image1": c1[:, :1], "color0": c0, "color1": c1}
            model(d)
        torch.cuda.synchronize()
        outs[ki] = {k: d[k].cpu() for k in ("b_ids", "i_ids", "j_ids", "mconf")}
    model.kv_init = True
    key = lambda o: set(zip(o["b_ids"].tolist(), o["i_ids"].tolist(), o["j_ids"].tolist()))   # noqa: E731
    a, b = key(outs[True]), key(outs[False])
    assert len(b) >= 200 and len(a ^ b) <= 0.01 * len(b) + 2, (len(a), len(b), len(a ^ b))


def test_padded_inputs_keep_the_row_path_for_k_v_and_agree_with_emitted_queries():
    """padding masks: the state reduction masks k / v ROWS, so the fused KV state is off (rows + la_kv launches) while the queries stay local
    (the query mask acts in the apply step) -- against the same forward with emitted q rows"""
    from tools import synth_loftr as S
    model, _ = S.synthetic_model("fp16")
    model = model.cuda()
    c0, c1 = S.textured_pairs(2, 256, 256, seed=9)
    m0 = torch.zeros(2, 32, 32, dtype=torch.bool); m1 = torch.zeros(2, 32, 32, dtype=torch.bool)
    for b, ((h0, w0), (h1, w1)) in enumerate([((32, 28), (30, 32)), ((24, 32), (32, 32))]):
        m0[b, :h0, :w0] = True; m1[b, :h1, :w1] = True
    outs = {}
    for ql in (True, False):
        model.q_local = ql
        model._invalidate()
        for _ in range(2):
            d = {"image0": c0[:, :1].cuda(), "image1": c1[:, :1].cuda(), "color0": c0.cuda(), "color1": c1.cuda(), "mask0": m0.cuda(), "mask1": m1.cuda()}
            model(d)
        torch.cuda.synchronize()
        outs[ql] = {k: d[k].cpu() for k in ("b_ids", "i_ids", "j_ids", "mconf")}
    model.q_local = True
    key = lambda o: set(zip(o["b_ids"].tolist(), o["i_ids"].tolist(), o["j_ids"].tolist()))   # noqa: E731
    a, b = key(outs[True]), key(outs[False])
    assert len(b) >= 50 and len(a ^ b) <= 0.02 * len(b) + 2, (len(a), len(b), len(a ^ b))
    # masked coarse cells never match (coarse_matching.py:117-119)
    for o in outs.values():
        for bb, i, j in zip(o["b_ids"].tolist(), o["i_ids"].tolist(), o["j_ids"].tolist()):
            assert m0[bb].reshape(-1)[i] and m1[bb].reshape(-1)[j]


@pytest.mark.parametrize("tdt", KINDS, ids=KIDS)
def test_token_project_with_positional_encoding_in_front(tdt):
    """`pos_fused` (round 5): the projection-only launch computes its rows feat + pe[row % hw] itself (gim_posenc_add's arithmetic), writes them to
    the fp32 stream and the operand copy and projects them -- bit for bit what gim_posenc_add followed by the plain launch gives."""
    from gim_amd import ops
    from gim_amd._lib import ACT_ELU1, ACT_NONE
    from gim_amd.packing import pack_token_emit
    nb, L = 3, 192
    R = nb * L
    g = torch.Generator().manual_seed(51)
    wk, wv = (torch.randn(256, 256, generator=g) / 16 for _ in range(2))
    ew = pack_token_emit([wk, wv], "cuda", tdt)
    feat = torch.randn(R, 256, generator=g).to(tdt).cuda()
    pe = torch.randn(L, 256, generator=g).cuda()
    outs = []
    for fused in (False, True):
        x32 = torch.full((R, 256), 3.0, device="cuda")
        cat = torch.full((R, 512), 7.0, dtype=tdt, device="cuda")
        ws = ops.kv_state_workspace(nb, L // 64, "cuda")
        ws.fill_(float("nan"))
        spec = [(None, ACT_ELU1, 0, R, (ws, nb, L // 64, 0, L)), (None, ACT_NONE, 0, R)]
        if fused:
            ops.token_project(cat[:, :256], (ew, spec), pos=(feat, pe, x32))
        else:
            ops.posenc_add(feat, pe, x32, cat[:, :256])
            ops.token_project(cat[:, :256], (ew, spec))
        ops.kv_state_finalize(ws, nb, L // 64)
        torch.cuda.synchronize()
        outs.append((x32, cat, ws[:nb * 8 * 1056].clone()))
    assert torch.isfinite(outs[0][2]).all() and bool((outs[1][1][:, 256:] == 7.0).all())
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert torch.equal(outs[1][0], (feat.float() + pe.repeat(nb, 1)))


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
def test_forward_with_fused_positional_encoding_is_bit_identical(precision):
    """the whole forward with the positional encoding inside the projection-only launch against gim_posenc_add in front of it: same arithmetic"""
    from tools import synth_loftr as S
    model, _ = S.synthetic_model(precision)
    model = model.cuda()
    c0, c1 = S.textured_pairs(4, 256, 320, seed=24, frac=0.5)
    c0, c1 = c0.cuda(), c1.cuda()
    outs = {}
    for pf in (True, False):
        model.pos_fused = pf
        model._invalidate()
        for _ in range(2):
            d = {"image0": c0[:, :1], "image1": c1[:, :1], "color0": c0, "color1": c1}
            model(d)
        torch.cuda.synchronize()
        outs[pf] = {k: d[k].clone() for k in ("b_ids", "i_ids", "j_ids", "mconf", "mkpts0_f", "mkpts1_f")}
    model.pos_fused = True
    assert outs[True]["b_ids"].numel() >= 200
    for k in outs[True]:
        assert torch.equal(outs[True][k], outs[False][k]), k
