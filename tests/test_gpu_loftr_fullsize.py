"""Parity of the BENCHMARKED gim_loftr configuration (640x480, the BASELINE config-2 size) against the CPU oracle on
match-rich synthetic pairs (tools/synth_loftr.py: calibrated "trained-like" weights + textured pairs, ~1400 coarse
matches per pair come out of the images -- nothing is injected into the forward).

  * fp32 mode, one 640x480 pair: match set exact (any flip must be marginal in the oracle's own decision, i.e. within
    1e-3 of the threshold / of a mutual-NN tie), coordinates / confidences within 1e-4 (north_star's bar);
  * bf16 mode (the mode bench.py times), batch 8: index flip rate and coordinate deviation against the fp32 oracle,
    with stated bounds -- bf16 operand rounding in the backbone / transformer moves confidences by ~0.02 on average,
    so matches whose confidence sits near thr = 0.2 flip; measured 1.7-2.6 % (both coarse_sim settings);
  * fine level with >= 500 matches: fine transformer output (a8) and fine matching (a9) against the oracle's
    `local_feature_transformer(sd, "loftr_fine", ...)` / `fine_matching` on the same windows.
"""
import pytest
import torch

import loftr_oracle as O
from tools import synth_loftr as S
from tools.parity import flip_margins, parity_vs_oracle

pytestmark = pytest.mark.gpu

H, W = 480, 640


def _data(c0, c1, dev=None):
    d = {"image0": c0[:, :1], "image1": c1[:, :1], "color0": c0, "color1": c1}
    return {k: v.to(dev) for k, v in d.items()} if dev is not None else d


@pytest.fixture(scope="module")
def synth():
    torch.set_num_threads(min(64, torch.get_num_threads()))
    model, sd = S.synthetic_model("fp32")
    return model.to("cuda:0"), sd


@pytest.fixture(scope="module")
def pairs():
    return S.textured_pairs(8, H, W, seed=1234, frac=0.45)


@pytest.fixture(scope="module")
def oracle_two_pairs(synth, pairs):
    _, sd = synth
    c0, c1 = pairs
    with torch.no_grad():
        return O.loftr_forward(sd, _data(c0[:2], c1[:2]))


def test_fp32_640x480_exact_vs_oracle(synth, pairs, oracle_two_pairs):
    model, _ = synth
    model.set_precision("fp32")
    c0, c1 = pairs
    ref = oracle_two_pairs
    d = _data(c0[:2], c1[:2], "cuda:0")
    model(d)
    torch.cuda.synchronize()
    assert ref["b_ids"].numel() >= 2000  # match-rich: ~1400 per pair
    for b in range(2):
        p = parity_vs_oracle(d, ref, b, b)
        print("fp32 640x480 pair", b, p)
        flips = flip_margins(d, ref, b, b)
        # exact, except where the oracle's own decision is a coin toss at fp32 resolution
        assert all(f[4] < 1e-3 or f[5] < 1e-3 for f in flips), flips
        assert len(flips) <= 2, flips
        assert p["max_abs_dmconf"] <= 1e-4, p
        assert p["max_abs_dmkpts1_px"] <= 1e-4 and p["max_abs_dmkpts0_px"] == 0.0, p
        assert p["max_abs_dexpec_f"] <= 1e-4, p
    if not any(flip_margins(d, ref, b, b) for b in range(2)):  # then order and every index are identical as well
        for k in ("b_ids", "i_ids", "j_ids"):
            assert torch.equal(d[k].cpu(), ref[k]), k


# (precision, coarse_sim, stem_fp16, stem_split, bound on the index flip rate, bound on mean |d mconf|): bounds = 2 x the worse of the
# two pairs measured on MI355X.  Plain stem (profiles/r03_parity_modes.txt): bf16 with the fp16 stem 0.67 % / 1.30 % flips, mean
# |d mconf| 0.009; bf16 incl. a bf16 stem (round 2's mode) 1.68 % / 2.61 %, 0.018; fp16 0.27 % / 0.15 %, 0.0024.  Split stem (round 4,
# the default; profiles/r04_parity_modes.txt): see the table there.  The CPU emulation of the engine's roundings
# (tools/precision_emulation.py, profiles/r04_precision_sweep.txt) predicts 0.47 % / 0.0025 for fp16 with a plain stem and
# 0.19 % / 0.0012 with the stem exact.
MODES = [("bf16", "fp32", True, False, 0.026, 0.018), ("bf16", "bf16", True, False, 0.026, 0.018), ("bf16", "bf16", False, False, 0.052, 0.036),
         ("fp16", "fp16", True, False, 0.0055, 0.005), ("fp16", "fp32", True, False, 0.0055, 0.005),
         ("fp16", "fp16", True, True, 0.004, 0.0025), ("bf16", "bf16", True, True, 0.02, 0.016)]


@pytest.mark.parametrize("precision,coarse_sim,stem_fp16,stem_split,max_flip,max_dconf", MODES,
                         ids=[f"{m[0]}-sim{m[1]}-{'stemfp16' if m[2] else 'stembf16'}{'-split' if m[3] else ''}" for m in MODES])
def test_bf16_batch8_flip_rate_vs_oracle(synth, pairs, oracle_two_pairs, precision, coarse_sim, stem_fp16, stem_split, max_flip, max_dconf):
    """The benchmarked 16-bit modes against the fp32 oracle, batch 8 (what bench.py times).  Round 2 (bf16 incl. a bf16 stem)
    measured 1.7-2.6 % flips, 0.006 px mean / 0.33 px max coordinate deviation (profiles/r02_parity_probe.txt)."""
    model, _ = synth
    model.stem_fp16 = stem_fp16
    model.stem_split = stem_split
    model.set_precision(precision, coarse_sim)
    c0, c1 = pairs
    try:
        for _ in range(3):  # eager, capture, replay: the replayed graph is what bench.py times
            d = _data(c0, c1, "cuda:0")
            model(d)
        torch.cuda.synchronize()
        assert len(model._graphs) == 1 and model.precision == precision   # (the fp16 range guard did not trip)
        assert d["b_ids"].numel() >= 8 * 1000
        for b in range(2):
            p = parity_vs_oracle(d, oracle_two_pairs, b, b)
            print(precision, "batch-8 coarse_sim", coarse_sim, "stem_fp16", stem_fp16, "stem_split", stem_split, "pair", b, p)
            assert p["flip_rate"] <= max_flip, p
            assert p["mean_abs_dmkpts1_px"] <= 0.02 and p["max_abs_dmkpts1_px"] <= 1.0, p
            assert p["mean_abs_dmconf"] <= max_dconf, p
            assert torch.isfinite(d["mconf"]).all() and torch.isfinite(d["mkpts1_f"]).all()
    finally:
        model.stem_fp16 = True
        model.stem_split = "auto"
        model.set_precision("fp32")


def test_graph_replay_is_bitwise_eager(synth, pairs):
    model, _ = synth
    model.set_precision("bf16")
    c0, c1 = pairs
    outs = []
    try:
        for _ in range(3):
            d = _data(c0[:4], c1[:4], "cuda:0")
            model(d)
            outs.append({k: d[k].clone() for k in ("b_ids", "i_ids", "j_ids", "mconf", "mkpts1_f", "expec_f")})
        torch.cuda.synchronize()
        for k, v in outs[0].items():  # call 1 ran eagerly, call 3 replayed the captured graph
            assert torch.equal(v, outs[2][k]), k
    finally:
        model.set_precision("fp32")


def test_fine_level_match_rich_vs_oracle(synth):
    """a8 / a9 with >= 500 matches (the fine transformer was only checked on <= 4 matches)."""
    from gim_amd import ops
    model, sd = synth
    model.set_precision("fp32")
    c0, c1 = S.textured_pairs(2, 256, 320, seed=5, frac=1.0)
    model.debug = {}
    try:
        d = _data(c0, c1, "cuda:0")
        model(d)
        torch.cuda.synchronize()
        dbg = model.debug
    finally:
        model.debug = None
    M = d["b_ids"].numel()
    assert M >= 500, M
    b_ids, i_ids, j_ids = d["b_ids"].cpu(), d["i_ids"].cpu(), d["j_ids"].cpu()
    # the engine's own fine maps -> the oracle's window extraction + fine transformer + fine matching
    f0 = ops.nhwc_to_nchw(dbg["f0"].contiguous(), 128).cpu()
    f1 = ops.nhwc_to_nchw(dbg["f1"].contiguous(), 128).cpu()
    with torch.no_grad():
        w0, w1 = O.fine_preprocess(f0, f1, b_ids, i_ids, j_ids, (32, 40), (128, 160), 5)
        t0, t1 = O.local_feature_transformer(sd, "loftr_fine", w0, w1, 8, 1)
        fm = O.fine_matching(t0, t1, d["mkpts0_c"].cpu(), d["mkpts1_c"].cpu(), b_ids, M, (256, 320), (128, 160))
    for got, ref, name in ((dbg["fine0"], t0, "fine0"), (dbg["fine1"], t1, "fine1")):
        err = (got.cpu() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 1e-4, (name, err)
    assert (d["expec_f"].cpu() - fm["expec_f"]).abs().max() < 1e-4
    assert (d["mkpts1_f"].cpu() - fm["mkpts1_f"]).abs().max() < 1e-4
    assert fm["expec_f"][:, :2].abs().max() > 0.05  # the sub-pixel refinement is not degenerate
