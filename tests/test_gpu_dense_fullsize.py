"""Full-size single-pair comparisons of the two dense matchers against their CPU oracles: gim_dkm at
672x896 (+ the 1152x1536 upsampling pass) and gim_roma at 672x672, fp32 mode, seeded weights, same tolerances as the small-size
golden tests (the GP posterior's conditioning sets them: tests/test_gpu_gp_pins.py).  ~16 s of CPU oracle per model on the GPU
box's host (minutes on a small container): GIM_SKIP_SLOW_TESTS=1 skips them.
Measured (round 2): gim_dkm warp max 7.9e-4 / mean 6.5e-5 of scale, certainty max 2.6e-4 -- every value inside the tolerance;
gim_roma warp 99.34 % of the values within 2e-3, mean 8.5e-4, the rest are isolated flipped decisions (max 1.1 of scale).

Round 3: the engine's fp32 mode evaluates the GP posterior in fp64 (gim_gp_posterior_f64), i.e. at the exact value of the
reference's formula.  Two comparisons per model: (a) against the reference arithmetic (fp32 kernel matrix, fp32 LU inverse) at the
tolerance that arithmetic's own rounding sets, and (b) against the SAME oracle with only the GP step evaluated in fp64
(dkm_oracle.GP_FP64) at north_star's 1e-4 -- plus the distance between the two oracles, which is what no fp32 implementation of
the reference can get below."""
import os

import pytest
import torch

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(bool(os.environ.get("GIM_SKIP_SLOW_TESTS")), reason="GIM_SKIP_SLOW_TESTS is set")]


@pytest.fixture(autouse=True)
def _numpy_inverse(monkeypatch):
    """torch.linalg.inv on the CPU can fail in this image's torch build ("Pivots given to lu_solve must all be greater or equal to
    1": observed for the 2352 x 2352 GP matrix, and for smaller ones after torch.set_num_threads() had been called -- which is
    why this file leaves the thread count alone): fall back to numpy's LAPACK for the oracle's fp32 LU inverse when it does"""
    import numpy as np
    real = torch.linalg.inv

    def inv(a):
        try:
            return real(a)
        except RuntimeError:
            if a.is_cuda:
                raise
            return torch.from_numpy(np.linalg.inv(a.detach().numpy())).to(a.dtype)
    monkeypatch.setattr(torch.linalg, "inv", inv)


def _close(got, ref, tol, name, frac=0.999, mean_tol=None):
    """max-norm agreement is not the right bar at this size: the matchers contain hard decisions (RoMa's arg-max over 4096 anchor
    classes, the certainty threshold of the warp refinement) that a 1e-4 difference in the GP posterior can flip at isolated
    pixels.  Required: `frac` of all values within tol x scale of the oracle's, and the mean error below tol x scale / 10."""
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    scale = max(ref.abs().max().item(), 1e-12)
    err = (got - ref).abs() / scale
    inside = (err <= tol).float().mean().item()
    print(f"{name}: max {err.max().item():.3e}  mean {err.mean().item():.3e}  within {tol:g}: {100 * inside:.3f} %")
    assert inside >= frac and err.mean().item() <= (mean_tol if mean_tol is not None else tol / 10), (name, inside, err.mean().item())


def _dist(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    e = (a - b).abs() / max(b.abs().max().item(), 1e-12)
    return e.max().item(), e.mean().item()


def test_dkm_672x896_vs_oracle(monkeypatch):
    import dkm_oracle as O
    from gim_amd.dkm import DKMv3
    sd = O.make_state_dict(0)
    im0, im1 = O.seeded_pair(672, 896, 3)
    with torch.no_grad():
        ref_warp, ref_cert = O.match(sd, im0, im1, 672, 896, None)
        monkeypatch.setattr(O, "GP_FP64", True)
        x_warp, x_cert = O.match(sd, im0, im1, 672, 896, None)       # the formula: GP in fp64, everything else as the reference
        monkeypatch.setattr(O, "GP_FP64", False)
    m = DKMv3(None, 672, 896, upsample_preds=False, precision="fp32")
    m.load_state_dict(sd)
    m = m.eval()
    warp, cert = m.match(im0.to("cuda:0"), im1.to("cuda:0"))
    # (a) against the PINNED reference arithmetic: bounds = 2 x measured (profiles/r03_dense_parity.txt: warp max 2.9e-4 / mean
    # 2.5e-5 of scale, certainty max 8.8e-5 / mean 1.8e-5 -- the distance the reference's own fp32 GP arithmetic keeps from its formula)
    _close(warp, ref_warp, 6e-4, "dkm warp 672x896 vs the reference arithmetic", frac=1.0, mean_tol=5e-5)
    _close(cert, ref_cert, 2e-4, "dkm certainty 672x896 vs the reference arithmetic", frac=1.0, mean_tol=4e-5)
    # (b) against the exact-GP oracle: EVERY value within 2e-5 of scale (north_star: 1e-4; measured max 3.0e-6 / 1.1e-6,
    # profiles/r03_dense_parity.txt)
    _close(warp, x_warp, 2e-5, "dkm warp 672x896 vs the fp64-GP oracle", frac=1.0, mean_tol=2e-6)
    _close(cert, x_cert, 2e-5, "dkm certainty 672x896 vs the fp64-GP oracle", frac=1.0, mean_tol=2e-6)
    (rw_max, rw_mean), (ew_max, ew_mean) = _dist(ref_warp, x_warp), _dist(warp, x_warp)
    print(f"dkm 672x896 warp: reference arithmetic vs fp64-GP oracle max {rw_max:.2e} mean {rw_mean:.2e}; engine vs fp64-GP oracle max {ew_max:.2e} mean {ew_mean:.2e}")
    assert ew_mean <= rw_mean, "the engine's parity mode must sit closer to the formula than the reference's fp32 arithmetic does"


def test_roma_672_vs_oracle(monkeypatch):
    import dkm_oracle as DO
    import roma_oracle as O
    from gim_amd.roma import RoMa
    sd, dsd = O.make_state_dicts(0)
    im0, im1 = DO.seeded_pair(672, 672, 3)
    with torch.no_grad():
        ref_warp, ref_cert = O.match(sd, dsd, im0, im1, 672, 672, None)
        monkeypatch.setattr(DO, "GP_FP64", True)
        x_warp, x_cert = O.match(sd, dsd, im0, im1, 672, 672, None)
        monkeypatch.setattr(DO, "GP_FP64", False)
    m = RoMa([672, 672], precision="fp32", dinov2_weights=dsd)
    m.load_state_dict(sd)
    m = m.eval()
    m.upsample_preds = False
    warp, cert = m.match(im0.to("cuda:0"), im1.to("cuda:0"))
    # RoMa's coarse flow is an arg-max over 64 x 64 anchor classes (roma.py:94-136): with random weights ~0.7 % of the values sit
    # behind a decision that the GP's 1e-4 noise flips
    # (a) against the PINNED reference arithmetic, 2 x measured (profiles/r03_dense_parity.txt): 99.75 % of the warp values within
    # 2e-3 (the rest: arg-max flips caused by the reference's own GP noise, max 1.1 of scale), mean 3.6e-4; certainty max 7.5e-4
    _close(warp, ref_warp, 2e-3, "roma warp 672x672 vs the reference arithmetic", frac=0.995, mean_tol=7.5e-4)
    _close(cert, ref_cert, 1.5e-3, "roma certainty 672x672 vs the reference arithmetic", frac=1.0, mean_tol=1.1e-4)
    # (b) against the exact-GP oracle: the anchor arg-max no longer sees GP noise, so the flipped decisions go away: EVERY value
    # within 2e-5 of scale (measured max 3.6e-7 / 3.0e-6); a flipped arg-max at an isolated pixel would show as a value of O(1)
    _close(warp, x_warp, 2e-5, "roma warp 672x672 vs the fp64-GP oracle", frac=1.0, mean_tol=2e-6)
    _close(cert, x_cert, 2e-5, "roma certainty 672x672 vs the fp64-GP oracle", frac=1.0, mean_tol=4e-6)
    (rw_max, rw_mean), (ew_max, ew_mean) = _dist(ref_warp, x_warp), _dist(warp, x_warp)
    print(f"roma 672 warp: reference arithmetic vs fp64-GP oracle max {rw_max:.2e} mean {rw_mean:.2e}; engine vs fp64-GP oracle max {ew_max:.2e} mean {ew_mean:.2e}")
    assert ew_mean <= rw_mean, "the engine's parity mode must sit closer to the formula than the reference's fp32 arithmetic does"


def test_dkm_672x896_upsample_pass_vs_oracle(monkeypatch):
    """the 1152 x 1536 upsampling pass -- 3.71 of gim_dkm's 5.27 TFLOP per pair (dkm.py:680-714: the second encoder pass
    and the refiners at scales 8 ... 1 on the upsampled flow) -- against dkm_oracle.match with the same pass, at size.  Round 4 only checked
    its swap symmetry in bf16.  Same two comparisons as the low-resolution test: the exact-GP oracle at north_star's 1e-4, the pinned
    reference arithmetic at the distance its own fp32 GP keeps from the formula."""
    import dkm_oracle as O
    from gim_amd.dkm import DKMv3
    sd = O.make_state_dict(0)
    im0, im1 = O.seeded_pair(672, 896, 3)
    up = (1152, 1536)
    with torch.no_grad():
        ref_warp, ref_cert = O.match(sd, im0, im1, 672, 896, up)
        monkeypatch.setattr(O, "GP_FP64", True)
        x_warp, x_cert = O.match(sd, im0, im1, 672, 896, up)
        monkeypatch.setattr(O, "GP_FP64", False)
    m = DKMv3(None, 672, 896, upsample_preds=True, precision="fp32")
    m.upsample_res = up
    m.load_state_dict(sd)
    m = m.eval()
    warp, cert = m.match(im0.to("cuda:0"), im1.to("cuda:0"))
    assert warp.shape == (up[0], 2 * up[1], 4) and cert.shape == (up[0], 2 * up[1])
    # measured on MI355X (round 5): vs the pinned arithmetic warp max 2.7e-4 / mean 2.4e-5, certainty max 8.5e-5 / mean 1.7e-5; vs the
    # fp64-GP oracle warp max 3.1e-6 / mean 2.6e-7, certainty max 1.1e-6 / mean 2.0e-7 -- the upsampling pass adds nothing to either distance
    _close(warp, ref_warp, 6e-4, "dkm warp 1152x1536 (upsampling pass) vs the reference arithmetic", frac=1.0, mean_tol=5e-5)
    _close(cert, ref_cert, 2e-4, "dkm certainty 1152x1536 (upsampling pass) vs the reference arithmetic", frac=1.0, mean_tol=4e-5)
    _close(warp, x_warp, 1e-5, "dkm warp 1152x1536 (upsampling pass) vs the fp64-GP oracle", frac=1.0, mean_tol=1e-6)
    _close(cert, x_cert, 1e-5, "dkm certainty 1152x1536 (upsampling pass) vs the fp64-GP oracle", frac=1.0, mean_tol=1e-6)


def test_dkm_batch4_672x896_vs_single_pair_oracles(monkeypatch):
    """BASELINE config 3 is gim_dkm 672x896 at batch = 4: four pairs in one engine pass (match_batch; the reference's own batched mode asserts
    against the upsampling pass, dkm.py:662, so this is the low-resolution pass) against single-pair oracle calls.  Two DIFFERENT pairs, each at
    two batch positions ([A, B, B, A]: 2 x 16 s of CPU oracle instead of 4 x): every slot against its own pair's oracle, and equal slots
    bit-identical -- a slot's result must not depend on its neighbours or its position."""
    import dkm_oracle as O
    from gim_amd.dkm import DKMv3
    sd = O.make_state_dict(0)
    pa, pb = O.seeded_pair(672, 896, 3, shift=(8, 12)), O.seeded_pair(672, 896, 5, shift=(4, 14))
    monkeypatch.setattr(O, "GP_FP64", True)
    with torch.no_grad():
        ra, rb = O.match(sd, pa[0], pa[1], 672, 896, None), O.match(sd, pb[0], pb[1], 672, 896, None)
    monkeypatch.setattr(O, "GP_FP64", False)
    m = DKMv3(None, 672, 896, upsample_preds=False, precision="fp32")
    m.load_state_dict(sd)
    m = m.eval()
    order = (pa, pb, pb, pa)
    W, C = m.match_batch(torch.cat([p[0] for p in order]).to("cuda:0"), torch.cat([p[1] for p in order]).to("cuda:0"))
    assert W.shape == (4, 672, 2 * 896, 4) and C.shape == (4, 672, 2 * 896)
    for k, (rw, rc) in enumerate((ra, rb, rb, ra)):
        _close(W[k], rw, 2e-5, f"dkm batch-4 slot {k} warp vs the fp64-GP oracle", frac=1.0, mean_tol=2e-6)
        _close(C[k], rc, 2e-5, f"dkm batch-4 slot {k} certainty vs the fp64-GP oracle", frac=1.0, mean_tol=2e-6)
    assert torch.equal(W[0], W[3]) and torch.equal(W[1], W[2]) and torch.equal(C[0], C[3]) and torch.equal(C[1], C[2])


# ---- round 6: gim_roma at BASELINE config 4's size, and its upsampling pass at size -------------------------------------
def _roma_vs_oracles(monkeypatch, size, up, tag, pinned=True):
    """engine (fp32 mode) at `size` x `size` (-> `up` when given) against (a) the pinned reference arithmetic and (b) the same oracle with
    only the GP step in fp64 -- the two comparisons of test_roma_672_vs_oracle / test_dkm_672x896_upsample_pass_vs_oracle"""
    import dkm_oracle as DO
    import roma_oracle as O
    from gim_amd.roma import RoMa
    sd, dsd = O.make_state_dicts(0)
    im0, im1 = DO.seeded_pair(size, size, 3)
    with torch.no_grad():
        if pinned:   # (one CPU pass of the upsampling oracle at 1344 x 1344 is ~1.5 min on the GPU box's host)
            ref_warp, ref_cert = O.match(sd, dsd, im0, im1, size, size, up)
        monkeypatch.setattr(DO, "GP_FP64", True)
        x_warp, x_cert = O.match(sd, dsd, im0, im1, size, size, up)
        monkeypatch.setattr(DO, "GP_FP64", False)
    m = RoMa([size], precision="fp32", dinov2_weights=dsd)
    m.load_state_dict(sd)
    m = m.eval()
    m.upsample_preds = up is not None
    if up is not None:
        m.upsample_res = up
    warp, cert = m.match(im0.to("cuda:0"), im1.to("cuda:0"))
    hs, ws = up if up is not None else (size, size)
    assert warp.shape == (hs, 2 * ws, 4) and cert.shape == (hs, 2 * ws)
    # (a) the pinned reference arithmetic: the anchor arg-max over 64 x 64 classes (roma.py:94-136) flips at isolated pixels under the
    # reference's OWN fp32 GP noise, so a fraction + a mean, as at 672 x 672 (bounds = 2 x measured on MI355X, round 6: see the callers)
    if pinned:
        _close(warp, ref_warp, 2e-3, f"roma warp {tag} vs the reference arithmetic", frac=0.995, mean_tol=7.5e-4)
        _close(cert, ref_cert, 1.5e-3, f"roma certainty {tag} vs the reference arithmetic", frac=0.999, mean_tol=1.5e-4)
    # (b) the exact-GP oracle: EVERY value within 2e-5 of scale (north_star: 1e-4)
    _close(warp, x_warp, 2e-5, f"roma warp {tag} vs the fp64-GP oracle", frac=1.0, mean_tol=2e-6)
    _close(cert, x_cert, 2e-5, f"roma certainty {tag} vs the fp64-GP oracle", frac=1.0, mean_tol=4e-6)
    if pinned:
        (rw_max, rw_mean), (ew_max, ew_mean) = _dist(ref_warp, x_warp), _dist(warp, x_warp)
        print(f"roma {tag} warp: reference arithmetic vs fp64-GP oracle max {rw_max:.2e} mean {rw_mean:.2e}; engine vs fp64-GP oracle max {ew_max:.2e} mean {ew_mean:.2e}")
        assert ew_mean <= rw_mean, "the engine's parity mode must sit closer to the formula than the reference's fp32 arithmetic does"


def test_roma_560_vs_oracle(monkeypatch):
    """BASELINE config 4: gim_roma at 560 x 560 (`RoMa(img_size=[560])`, roma.py:1124-1266; 1600-point GP), low-resolution pass"""
    _roma_vs_oracles(monkeypatch, 560, None, "560x560")


def test_roma_560_upsample_pass_vs_oracle(monkeypatch):
    """... and with the upsampling pass it runs by default (roma.py:658: upsample_res = 1344 x 1344 whatever img_size; roma.py:836-866: the second
    encoder pass + the refiners at scales 8 ... 1 on the upsampled flow -- 5.9 of config 4's 10 TFLOP), never compared with the oracle at size before"""
    _roma_vs_oracles(monkeypatch, 560, (1344, 1344), "560 -> 1344 (upsampling pass)")


def test_roma_672_upsample_pass_vs_oracle(monkeypatch):
    """the configuration gim's own callers run (`RoMa(img_size=[672])`, trainer/lightning.py:38-41, demo.py:332): 672 -> 1344, against the exact-GP
    oracle (the pinned-arithmetic comparison of this size's low-resolution pass is test_roma_672_vs_oracle, of the upsampling pass the 560 test)"""
    _roma_vs_oracles(monkeypatch, 672, (1344, 1344), "672 -> 1344 (upsampling pass)", pinned=False)


def test_roma_default_mode_is_fp16_and_close_to_the_fp32_mode_at_560():
    """round 6: gim_roma's DEFAULT 16-bit mode is IEEE fp16 (gim_amd/precision.py).  At config 4's size, with the upsampling pass: the default
    module against the fp32 parity mode of the same engine, and the bf16 mode on the same pair.  Measured on MI355X (round 6, seeded random
    weights): fp16 99.43 % of the warp values within 2e-3 of scale, mean 8.0e-4 -- the mean is ALL in the 0.57 % of values behind an anchor
    arg-max decision (roma.py:94-136, 4096 classes) that 11-bit storage still flips at isolated pixels (max 1.3 of scale); bf16: mean 1.04e-2,
    13 x further.  VERDICT r5's "< 1e-4" holds for the values outside those flips, not for the mean; fp16 is the default because it is an
    order of magnitude closer at the same speed, which is what this test holds (bounds = 2 x measured)."""
    import dkm_oracle as DO
    import roma_oracle as O
    from gim_amd.roma import RoMa
    sd, dsd = O.make_state_dicts(0)
    im0, im1 = DO.seeded_pair(560, 560, 3)
    out = {}
    for prec in (None, "fp32", "bf16"):
        m = RoMa([560], precision=prec, dinov2_weights=dsd)
        m.load_state_dict(sd)
        m = m.eval()
        if prec is None:
            assert m.precision == "fp16"
        w, c = m.match(im0.to("cuda:0"), im1.to("cuda:0"))
        if prec is None:
            assert m.precision == "fp16" and m._fp16_checked, "the default mode's output range check must have passed"
        out[prec] = (w.float().cpu(), c.float().cpu())
        del m
        torch.cuda.empty_cache()
    scale = out["fp32"][0].abs().max().item()
    e16 = (out[None][0] - out["fp32"][0]).abs() / scale
    ebf = (out["bf16"][0] - out["fp32"][0]).abs() / scale
    c16 = (out[None][1] - out["fp32"][1]).abs().mean().item()
    print(f"roma 560 -> 1344, vs the fp32 mode: default (fp16) warp mean {e16.mean().item():.2e} max {e16.max().item():.2e}, within 2e-3: "
          f"{100 * (e16 <= 2e-3).float().mean().item():.3f} %, certainty mean {c16:.2e}; bf16 warp mean {ebf.mean().item():.2e}")
    inside = e16 <= 2e-3
    print(f"  fp16 values outside 2e-3: {100 * (1 - inside.float().mean().item()):.3f} %; mean over the values inside: {e16[inside].mean().item():.2e}, median {e16.median().item():.2e}")
    assert inside.float().mean().item() >= 0.988 and e16.mean().item() < 1.6e-3 and e16[inside].mean().item() < 1e-4
    assert e16.mean().item() * 6 < ebf.mean().item(), "fp16 must be several times closer than bf16 on this pair (measured 13 x)"
