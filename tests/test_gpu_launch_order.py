"""Launch-order variants of the gim_loftr forward (gim_amd/loftr/loftr.py: `depth_groups`, `l3_chains`, `tf_chains`): images and
pairs are independent (BatchNorm in eval mode, per-sequence attention: networks/loftr/backbone/resnet.py:306-329,
submodules/transformer.py:80-101), so running the batch as image groups / pair chains on parallel streams must not change a bit
of the output -- same kernels, same per-image and per-sequence arithmetic."""
import pytest
import torch

from tools import synth_loftr as S

pytestmark = pytest.mark.gpu

KEYS = ("b_ids", "i_ids", "j_ids", "mconf", "mkpts0_f", "mkpts1_f", "expec_f")


def _run(model, c0, c1, reps=3):
    out = None
    for _ in range(reps):   # eager, (capture +) replay, replay
        d = {"image0": c0[:, :1], "image1": c1[:, :1], "color0": c0, "color1": c1}
        model(d)
        torch.cuda.synchronize()
        cur = {k: d[k].clone() for k in KEYS}
        if out is not None:
            for k in KEYS:
                assert torch.equal(cur[k], out[k]), ("replay differs from the eager forward", k)
        out = cur
    return out


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
@pytest.mark.parametrize("size", [(256, 320), (480, 640)])
def test_groups_and_chains_are_bit_identical(precision, size):
    H, W = size
    nb = 4
    model, _ = S.synthetic_model(precision)
    model = model.to("cuda:0")
    c0, c1 = S.textured_pairs(nb, H, W, seed=77, frac=0.5)
    c0, c1 = c0.cuda(), c1.cuda()
    model.depth_groups = model.l3_chains = model.tf_chains = 1
    base = _run(model, c0, c1)
    assert base["b_ids"].numel() >= 50 * nb
    for dg, l3, tf in ((2, 1, 1), (1, 2, 1), (1, 1, 2), (4, 2, 2)):
        model.depth_groups, model.l3_chains, model.tf_chains = dg, l3, tf
        model._invalidate()
        got = _run(model, c0, c1)
        for k in KEYS:
            assert torch.equal(got[k], base[k]), (dg, l3, tf, k)
    model.depth_groups = model.l3_chains = model.tf_chains = 1
