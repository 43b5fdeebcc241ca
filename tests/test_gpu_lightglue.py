"""GPU parity tests of the gim_lightglue path (SuperPoint + LightGlue + adapter), through the C ABI, against
oracle/lightglue_oracle.py on the same seeded inputs and against the golden vectors recorded from the
reference's own modules (tests/golden/lg_*.npz).

Bars: keypoint coordinates, arg-max / match indices: exact in fp32 mode (fp32 MFMA); floats 2e-5 of the
output scale in fp32 mode (summation order only); bf16 mode: 6.5e-3 = 2 x the largest value measured on MI355X (3.2e-3 of scale, one
bf16 output rounding; profiles/r04_secondary_measured.txt -- the bound was 1.5e-2 ... 2.5e-2 through round 3)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import lightglue_oracle as O

pytestmark = pytest.mark.gpu
DTS = ["fp32", "bf16"]


def _dev():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return torch.device("cuda:0")


def _tdt(dt):
    return torch.bfloat16 if dt == "bf16" else torch.float32


def _tol(dt):
    return 6.5e-3 if dt == "bf16" else 2e-5


def _close(got, ref, tol, what=""):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    scale = max(1e-6, ref.abs().max().item())
    err = (got - ref).abs().max().item()
    print(f"[close] {what}: {err / scale:.3e} of scale (tol {tol:g})")   # pytest -s: the measured value the tolerance is set from
    assert err <= tol * scale, f"{what}: max|err|={err:.3e} scale={scale:.3e} tol={tol}"


# ------------------------------------------------------------------------------------------- SuperPoint glue
@pytest.mark.parametrize("dt", DTS)
def test_maxpool2x2(dt):
    from gim_amd import ops
    dev = _dev()
    x = torch.randn(2, 12, 20, 64, generator=torch.Generator().manual_seed(1)).to(_tdt(dt))
    y = ops.maxpool2x2(x.to(dev))
    ref = F.max_pool2d(x.float().permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    assert torch.equal(y.float().cpu(), ref)


@pytest.mark.parametrize("dt", DTS)
def test_sp_scores(dt):
    from gim_amd import ops
    dev = _dev()
    B, h, w = 2, 5, 7
    logits = (3 * torch.randn(B * h * w, 72, generator=torch.Generator().manual_seed(2))).to(_tdt(dt))
    got = ops.sp_scores(logits.to(dev), B, h, w)
    lg = logits.float()[:, :65].reshape(B, h, w, 65).permute(0, 3, 1, 2)
    prob = F.softmax(lg, 1)[:, :-1]
    ref = prob.permute(0, 2, 3, 1).reshape(B, h, w, 8, 8).permute(0, 1, 3, 2, 4).reshape(B, h * 8, w * 8)
    _close(got, ref, 2e-6, "sp_scores")


def _score_map(B, H, W, seed, plateau=False):
    g = torch.Generator().manual_seed(seed)
    s = torch.rand(B, H, W, generator=g) * 0.05
    s = F.avg_pool2d(s[:, None], 3, 1, 1)[:, 0].contiguous()      # smooth: maxima spaced like real score maps
    if plateau:
        s[:, 10:14, 20:30] = 0.04                                  # equal neighbours both survive (superpoint.py:63)
    return s


@pytest.mark.parametrize("radius,border,plateau", [(3, 4, False), (4, 4, True), (1, 0, False)])
def test_sp_nms(radius, border, plateau):
    from gim_amd import ops
    dev = _dev()
    s = _score_map(2, 48, 72, 3, plateau)
    got = ops.sp_nms(s.to(dev), radius, border).cpu()
    ref = O.simple_nms(s, radius)
    if border:
        ref = O.mask_borders(ref, torch.tensor([[72, 48]] * 2), border)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("k,thr", [(64, 0.0), (200, 0.0), (4096, 0.0), (50, 0.02)])
def test_sp_topk(k, thr):
    """top-k (sorted, descending) when there are more than k candidates, torch.where order when fewer."""
    from gim_amd import ops
    dev = _dev()
    s = _score_map(3, 64, 96, 4)
    nms = O.mask_borders(O.simple_nms(s, 3), torch.tensor([[96, 64]] * 3), 4)
    kpts, ksc, nv = ops.sp_topk(nms.to(dev), k, thr)
    ref_k, ref_s = O.select_keypoints(nms, thr, k)
    for b in range(3):
        n = len(ref_k[b])
        assert int(nv[b]) == n
        assert torch.equal(kpts[b, :n].cpu(), ref_k[b]), b
        assert torch.equal(ksc[b, :n].cpu(), ref_s[b])
        assert (kpts[b, n:] == 0).all() and (ksc[b, n:] == 0).all()


def test_sp_topk_ties_and_plateau():
    """equal scores: lower flat index first (deterministic); a constant plateau keeps every pixel a candidate"""
    from gim_amd import ops
    dev = _dev()
    nms = torch.zeros(1, 32, 32)
    nms[0, 5:9, 5:9] = 0.5
    nms[0, 20, 3] = 0.7
    kpts, ksc, nv = ops.sp_topk(nms.to(dev), 8, 0.0)
    assert int(nv[0]) == 8
    assert kpts[0, 0].tolist() == [3.0, 20.0] and float(ksc[0, 0]) == pytest.approx(0.7)
    exp = [[5.0 + i, 5.0] for i in range(4)] + [[5.0 + i, 6.0] for i in range(3)]
    assert kpts[0, 1:].tolist() == exp


@pytest.mark.parametrize("dt", ["fp32"])
def test_sp_sample_desc(dt):
    from gim_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    B, h, w, K = 2, 8, 12, 70
    dense = torch.randn(B, 256, h, w, generator=g)
    kp = torch.stack([torch.randint(0, w * 8, (B, K), generator=g), torch.randint(0, h * 8, (B, K), generator=g)], -1).float()
    kp[0, 0] = torch.tensor([0.0, 0.0]); kp[0, 1] = torch.tensor([w * 8 - 1.0, h * 8 - 1.0])   # outside the sample grid
    ref = O.sample_descriptors_legacy(kp.clone(), F.normalize(dense, p=2, dim=1), 8).transpose(-1, -2)
    rows = dense.permute(0, 2, 3, 1).reshape(B * h * w, 256).contiguous().to(dev)
    out = torch.empty(B * K, 256, device=dev)
    ops.sp_sample_desc(rows, kp.to(dev), h, w, out, None)
    _close(out.view(B, K, 256), ref, 2e-6, "sample_desc")


# ------------------------------------------------------------------------------------------- LightGlue pieces
def test_posenc_rotary():
    from gim_amd import ops
    dev = _dev()
    _, lg = O.make_state_dicts(0)
    g = torch.Generator().manual_seed(6)
    B, K = 2, 50
    kp = torch.rand(B, K, 2, generator=g) * torch.tensor([640.0, 480.0])
    size = torch.tensor([[640.0, 480.0], [500.0, 480.0]])
    enc_ref = O.fourier_encoding(lg, O.normalize_keypoints(kp, size))            # [2,B,1,K,64]
    enc = ops.lg_posenc(kp.to(dev), size.to(dev), lg["posenc.Wr.weight"].to(dev))
    _close(enc.view(B, K, 64)[..., :32], enc_ref[0, :, 0, :, 0::2], 1e-5, "cos")
    _close(enc.view(B, K, 64)[..., 32:], enc_ref[1, :, 0, :, 0::2], 1e-5, "sin")
    q = torch.randn(B, K, 768, generator=g)
    x = q.view(B * K, 768).clone().to(dev)
    ops.lg_rotary(x, enc, 512)
    ref = q.clone()
    for h in range(8):
        ref[..., h * 64:(h + 1) * 64] = O._rotary(enc_ref[:, :, 0], q[..., h * 64:(h + 1) * 64])
    _close(x.view(B, K, 768), ref, 1e-5, "rotary")
    assert torch.equal(x.view(B, K, 768)[..., 512:].cpu(), q[..., 512:])


@pytest.mark.parametrize("dt", DTS)
def test_transpose_and_cast(dt):
    from gim_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(7)
    nb, S, C = 3, 100, 256
    Sp = 128
    src = torch.randn(nb * S, 768, generator=g).to(_tdt(dt))
    dst = torch.full((nb, C, Sp), 7.0, dtype=_tdt(dt), device=dev)
    ops.lg_transpose(src.to(dev)[:, 512:], dst, nb, S, Sp, C)
    ref = torch.zeros(nb, C, Sp)
    ref[:, :, :S] = src.float()[:, 512:].reshape(nb, S, C).transpose(1, 2)
    assert torch.equal(dst.float().cpu(), ref)
    x = torch.randn(40, 256, generator=g)
    out = torch.empty(40, 512, dtype=_tdt(dt), device=dev)
    ops.cast_rows(x.to(dev), out[:, :256])
    assert torch.equal(out[:, :256].float().cpu(), x.to(_tdt(dt)).float())


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("L,S,cross", [(128, 128, False), (100, 77, True), (300, 2048, True), (2048, 2048, False), (1, 5, True)])
def test_sdpa(dt, L, S, cross):
    """flash SDPA vs softmax(q k^T / 8) v in fp64 on the kernel's operand precision"""
    from gim_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(8)
    nb, H = 2, 4
    tdt = _tdt(dt)
    q = torch.randn(nb * L, 768, generator=g).to(tdt)
    kv = torch.randn(nb * S, 768, generator=g).to(tdt)
    kv[:, 256:512] *= 2.0          # sharper softmax
    Sp = (S + 63) // 64 * 64
    vt = torch.empty(nb, 256, Sp, dtype=tdt, device=dev)
    kvd = kv.to(dev)
    ops.lg_transpose(kvd[:, 512:], vt, nb, S, Sp, 256)
    out = torch.empty(nb * L, 256, dtype=tdt, device=dev)
    shift = 1 if cross else 0
    ops.sdpa(q.to(dev)[:, :256], kvd[:, 256:512], vt, out, nb, H, L, S, Sp, kv_shift=shift)
    qq = q.double()[:, :256].reshape(nb, L, H, 64).transpose(1, 2)
    kk = kv.double()[:, 256:512].reshape(nb, S, H, 64).transpose(1, 2).roll(-shift, 0)
    vv = kv.double()[:, 512:].reshape(nb, S, H, 64).transpose(1, 2).roll(-shift, 0)
    ref = (torch.softmax(qq @ kk.transpose(-1, -2) / 8.0, -1) @ vv).transpose(1, 2).reshape(nb * L, 256)
    _close(out, ref.float(), 1e-5 if dt == "fp32" else 6.5e-3, f"sdpa L={L} S={S}")


@pytest.mark.parametrize("dt", DTS)
def test_layernorm_gelu(dt):
    from gim_amd import ops
    from gim_amd._lib import ACT_GELU, ACT_NONE
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    x = 2 * torch.randn(77, 512, generator=g)
    gamma, beta = 1 + 0.1 * torch.randn(512, generator=g), 0.1 * torch.randn(512, generator=g)
    for act in (ACT_GELU, ACT_NONE):
        out = torch.empty(77, 512, dtype=_tdt(dt), device=dev)
        ops.layernorm_act(x.to(dev), gamma.to(dev), beta.to(dev), out, act)
        ref = F.layer_norm(x, (512,), gamma, beta)
        ref = F.gelu(ref) if act == ACT_GELU else ref
        _close(out, ref, 5e-3 if dt == "bf16" else 2e-6, "layernorm_act")


def _assign_inputs(B, M, N, seed):
    _, lg = O.make_state_dicts(0)
    g = torch.Generator().manual_seed(seed)
    d0 = torch.randn(B, M, 256, generator=g)
    d1 = torch.randn(B, N, 256, generator=g)
    nm = min(M, N) // 2
    for b in range(B):   # plant correspondences so that many rows have a confident mutual match
        perm = torch.randperm(N, generator=g)[:nm]
        d1[b, perm] = d0[b, :nm] + 0.05 * torch.randn(nm, 256, generator=g)
    return lg, d0, d1


@pytest.mark.parametrize("B,M,N", [(2, 128, 128), (1, 300, 170), (2, 2048, 2048)])
def test_lg_assign(B, M, N):
    """fused assignment vs oracle log_assignment + filter_matches: indices exact, scores 1e-5, lazy matrix 2e-5"""
    from gim_amd import ops
    dev = _dev()
    lg, d0, d1 = _assign_inputs(B, M, N, 10)
    i = 8
    p = f"log_assignment.{i}"
    lg[p + ".final_proj.weight"] = lg[p + ".final_proj.weight"] * 0.1      # keep exp() in range for random descriptors
    scores, _ = O.log_assignment(lg, i, d0, d1)
    m0, m1, ms0, ms1 = O.filter_matches(scores, 0.1)
    md0 = F.linear(d0, lg[p + ".final_proj.weight"], lg[p + ".final_proj.bias"])
    md1 = F.linear(d1, lg[p + ".final_proj.weight"], lg[p + ".final_proj.bias"])
    r = ops.lg_assign(d0.to(dev), d1.to(dev), md0.to(dev).contiguous(), md1.to(dev).contiguous(),
                      lg[p + ".matchability.weight"].reshape(-1).to(dev), lg[p + ".matchability.bias"].to(dev), 0.1)
    assert (m0 > -1).sum() > B * min(M, N) // 8, "test inputs must be match-rich"
    assert torch.equal(r.matches0.cpu(), m0) and torch.equal(r.matches1.cpu(), m1)
    _close(r.mscores0, ms0, 2e-5, "mscores0")
    _close(r.mscores1, ms1, 2e-5, "mscores1")
    assert r.count.tolist() == [(m0[b] > -1).sum().item() for b in range(B)]
    la = ops.lg_log_assignment(r)
    _close(la, scores, 2e-5, "log_assignment")
    total = sum(r.count.tolist())
    kp0, kp1 = torch.rand(B, M, 2) * 600, torch.rand(B, N, 2) * 600
    sc0, sc1 = torch.tensor([[1.5, 2.0]] * B), torch.tensor([[0.5, 1.25]] * B)
    matches, sc, mk0, mk1, bids = ops.lg_emit_matches(r, total, kp0.to(dev), kp1.to(dev), sc0.to(dev), sc1.to(dev))
    exp_m = torch.cat([torch.stack([torch.where(m0[b] > -1)[0], m0[b][m0[b] > -1]], -1) for b in range(B)])
    assert torch.equal(matches.cpu(), exp_m)
    exp_b = torch.cat([torch.full(((m0[b] > -1).sum().item(),), b) for b in range(B)])
    assert torch.equal(bids.cpu(), exp_b)
    exp0 = torch.cat([kp0[b][torch.where(m0[b] > -1)[0]] * sc0[b] for b in range(B)])
    exp1 = torch.cat([kp1[b][m0[b][m0[b] > -1]] * sc1[b] for b in range(B)])
    assert torch.equal(mk0.cpu(), exp0) and torch.equal(mk1.cpu(), exp1)
    _close(sc, torch.cat([ms0[b][m0[b] > -1] for b in range(B)]), 2e-5, "scores list")


# ------------------------------------------------------------------------------------------- whole modules
def _models(precision, K):
    from gim_amd.lightglue import LightGlue, SuperPoint
    sp_sd, lg_sd = O.make_state_dicts(0)
    det = SuperPoint({"max_num_keypoints": K, "force_num_keypoints": True, "detection_threshold": 0.0, "nms_radius": 3,
                      "trainable": False, "precision": precision})
    lg = LightGlue({"filter_threshold": 0.1, "flash": False, "checkpointed": True, "precision": precision})
    det.load_state_dict(sp_sd)
    lg.load_state_dict(lg_sd)
    return det.eval(), lg.eval(), sp_sd, lg_sd


def test_superpoint_golden_fp32(golden_dir):
    """engine (fp32 mode) vs the reference's own SuperPoint outputs"""
    g = np.load(os.path.join(golden_dir, "lg_superpoint.npz"))
    det, _, _, _ = _models("fp32", int(g["K"]))
    b, h, w = (int(v) for v in g["shape"])
    img = O.seeded_gray(b, h, w, int(g["seed"]))
    out = det({"image": img.to(_dev())})
    _close(det._debug["scores"], torch.as_tensor(g["dense_scores"]), 2e-5, "dense scores")
    kp, ref = out["keypoints"].cpu(), torch.as_tensor(g["keypoints"])
    same = (kp == ref).all(-1).float().mean().item()
    # keypoint order is a sort on fp32 scores that agree to ~1e-7 relative: allow a handful of swaps
    assert same >= 0.97, same
    for i in range(b):
        a = set(map(tuple, kp[i].tolist())); r = set(map(tuple, ref[i].tolist()))
        assert len(a & r) >= 0.98 * len(r)
    keep = (kp == ref).all(-1)
    _close(out["descriptors"].cpu()[keep], torch.as_tensor(g["descriptors"])[keep], 5e-5, "descriptors")


def test_superpoint_rgb_and_bf16(golden_dir):
    g = np.load(os.path.join(golden_dir, "lg_superpoint_rgb.npz"))
    det, _, _, _ = _models("fp32", 32)
    rgb = torch.rand(1, 3, 64, 96, generator=torch.Generator().manual_seed(int(g["seed"])))
    out = det({"image": rgb.to(_dev())})
    kp, ref = out["keypoints"].cpu(), torch.as_tensor(g["keypoints"])
    assert (kp == ref).all(-1).float().mean().item() >= 0.9
    det16, _, _, _ = _models("bf16", 32)
    out16 = det16({"image": rgb.to(_dev())})
    assert out16["keypoints"].shape == (1, 32, 2) and torch.isfinite(out16["descriptors"]).all()
    _close(det16._debug["scores"], det._debug["scores"], 0.1, "bf16 score map")


def test_superpoint_few_keypoints_padding():
    """fewer candidates than K: found points first (torch.where order), the rest uniform random inside their
    bounding box, zero scores (pad_and_stack 'random_c', misc.py:44-55)"""
    det, _, _, _ = _models("fp32", 2048)
    img = O.seeded_gray(1, 64, 96, 3)
    out = det({"image": img.to(_dev())})
    n = det._debug["nvalid"][0]
    assert 0 < n < 2048 and out["keypoints"].shape == (1, 2048, 2)
    found = out["keypoints"][0, :n].cpu() - 0.5
    flat = found[:, 1] * 96 + found[:, 0]
    assert (flat[1:] > flat[:-1]).all()                        # torch.where order
    pad = out["keypoints"][0, n:].cpu() - 0.5
    assert (pad[:, 0] >= found[:, 0].min()).all() and (pad[:, 0] <= found[:, 0].max()).all()
    assert (det._debug["keypoint_scores"][0, n:] == 0).all()
    assert torch.isfinite(out["descriptors"]).all()


def test_lightglue_golden_fp32(golden_dir):
    """engine (fp32 mode) vs the reference's own LightGlue outputs on planted descriptors: indices exact"""
    g = np.load(os.path.join(golden_dir, "lg_lightglue.npz"))
    _, lg, _, _ = _models("fp32", 128)
    dev = _dev()
    kp0, d0, kp1, d1 = O.planted_descriptors(2, int(g["K"]), seed=int(g["seed"]))
    rs = torch.tensor([[480, 640], [480, 640]])
    pred = lg({"keypoints0": kp0.to(dev), "keypoints1": kp1.to(dev), "descriptors0": d0.to(dev), "descriptors1": d1.to(dev),
               "resize0": rs.to(dev), "resize1": rs.to(dev)})
    _close(pred["ref_descriptors0"][:, 0], torch.as_tensor(g["ref_descriptors0"]), 1e-4, "ref_descriptors0")
    _close(pred["ref_descriptors1"][:, 0], torch.as_tensor(g["ref_descriptors1"]), 1e-4, "ref_descriptors1")
    assert np.array_equal(pred["matches0"].cpu().numpy(), g["matches0"])
    assert np.array_equal(pred["matches1"].cpu().numpy(), g["matches1"])
    _close(pred["matching_scores0"], torch.as_tensor(g["matching_scores0"]), 5e-3, "matching_scores0")
    _close(pred["log_assignment"].get(), torch.as_tensor(g["log_assignment"]), 2e-4, "log_assignment")
    assert pred["stop"] == 9 and len(pred["matches"]) == 2
    for b in range(2):
        m = g["matches0"][b]
        exp = np.stack([np.nonzero(m > -1)[0], m[m > -1]], -1)
        assert np.array_equal(pred["matches"][b].cpu().numpy(), exp)
    assert pred["prune0"].shape == (2, 128) and float(pred["prune0"][0, 0]) == 9.0


def test_lightglue_unequal_counts_and_bf16():
    """M != N (the stacked row block handles both sets), bf16 mode agrees with the fp32 oracle on most matches"""
    _, lg_sd = O.make_state_dicts(0)
    dev = _dev()
    kp0, d0, kp1, d1 = O.planted_descriptors(1, 192, seed=33)
    kp1, d1 = kp1[:, :150].contiguous(), d1[:, :150].contiguous()
    rs = torch.tensor([[480, 640]])
    data = {"keypoints0": kp0, "keypoints1": kp1, "descriptors0": d0, "descriptors1": d1, "resize0": rs, "resize1": rs}
    ref = O.lightglue_forward(lg_sd, data)
    for prec, need in (("fp32", 1.0), ("bf16", 0.985), ("fp16", 0.99)):   # bf16 measured 0.9948 (one of 192 keypoints differs): 2 x the disagreement; fp16: round 5
        _, lg, _, _ = _models(prec, 128)
        pred = lg({k: v.to(dev) for k, v in data.items()})
        agree = (pred["matches0"].cpu() == ref["matches0"]).float().mean().item()
        print(f"[agree] lightglue {prec} matches0 agreement with the fp32 oracle: {agree:.4f}")
        assert agree >= need, (prec, agree)
        assert pred["matches1"].shape == (1, 150)
    assert (ref["matches0"] > -1).sum() > 50


def test_e2e_pipeline_golden(golden_dir):
    """detector + matcher + adapter (trainer/lightning.py:161-193) vs the golden run of the reference modules"""
    from gim_amd.lightglue import gim_lightglue_inference
    g = np.load(os.path.join(golden_dir, "lg_e2e.npz"))
    det, lg, _, _ = _models("fp32", int(g["K"]))
    dev = _dev()
    img0 = O.seeded_gray(2, 96, 128, int(g["seeds"][0]))
    img1 = torch.roll(img0, shifts=tuple(int(x) for x in g["shift"]), dims=(2, 3)).contiguous()
    rs = torch.tensor([[96, 128], [96, 128]])
    scale = torch.as_tensor(g["scale"])
    data = {"image0": img0.to(dev), "image1": img1.to(dev), "resize0": rs.to(dev), "resize1": rs.to(dev),
            "scale0": scale.to(dev), "scale1": scale.to(dev)}
    pred = gim_lightglue_inference(det, lg, data)
    for k in ("hw0_i", "hw1_i", "mkpts0_f", "mkpts1_f", "m_bids", "mconf"):
        assert k in data
    assert tuple(data["hw0_i"]) == (96, 128)
    same0 = (pred["keypoints0"].cpu() == torch.as_tensor(g["keypoints0"])).all(-1).float().mean().item()
    assert same0 >= 0.95
    # the adapter outputs are consistent with the engine's own matches (exact gather + scale)
    m0 = pred["matches0"].cpu()
    exp0 = torch.cat([pred["keypoints0"][b].cpu()[m0[b] > -1] * scale[b] for b in range(2)])
    exp1 = torch.cat([pred["keypoints1"][b].cpu()[m0[b][m0[b] > -1]] * scale[b] for b in range(2)])
    assert torch.equal(data["mkpts0_f"].cpu(), exp0) and torch.equal(data["mkpts1_f"].cpu(), exp1)
    assert torch.equal(data["m_bids"].cpu(), torch.cat([torch.full(((m0[b] > -1).sum().item(),), b) for b in range(2)]))
    assert data["mconf"].shape[0] == exp0.shape[0]
    if same0 == 1.0 and (pred["keypoints1"].cpu() == torch.as_tensor(g["keypoints1"])).all():
        assert np.array_equal(m0.numpy(), g["matches0"])
        _close(data["mkpts0_f"], torch.as_tensor(g["mkpts0_f"]), 1e-6, "mkpts0_f")


def test_no_cpu_fallback():
    from gim_amd._lib import GimHipError
    det, lg, _, _ = _models("fp32", 32)
    with pytest.raises(GimHipError):
        det({"image": torch.rand(1, 1, 64, 64)})
    with pytest.raises(GimHipError):
        lg({"keypoints0": torch.rand(1, 8, 2), "keypoints1": torch.rand(1, 8, 2), "descriptors0": torch.rand(1, 8, 256),
            "descriptors1": torch.rand(1, 8, 256), "resize0": torch.tensor([[480, 640]]), "resize1": torch.tensor([[480, 640]])})
