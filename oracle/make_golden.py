"""TEST INFRASTRUCTURE ONLY -- pins `oracle/loftr_oracle.py` to the real reference and writes golden vectors.

Run by hand in the authoring container (needs `/root/reference`; does NOT run on the GPU box):

    python oracle/make_golden.py            # checks oracle == reference, rewrites tests/golden/*.npz

For each stage of the gim_loftr path it (1) builds seeded weights with the reference's state_dict
keys (`make_state_dict`) and loads them with strict=True into the *reference's own* module,
(2) runs the reference module and the oracle restatement on the same seeded inputs, (3) asserts they
agree (integers exactly, floats to 1e-5) and (4) stores the *reference's* outputs -- small enough to
commit -- under `tests/golden/`.  `tests/test_oracle_golden.py` regenerates the inputs from the same
seeds and replays the oracle against these files.

Reference entry points exercised (file:line in /root/reference):
  networks/loftr/backbone/resnet.py:306-329       ResNetFPN_8_2.forward
  networks/loftr/utils/position_encoding.py:38-43 PositionEncodingSine.forward
  networks/loftr/submodules/transformer.py:80-101 LocalFeatureTransformer.forward
  networks/loftr/utils/coarse_matching.py:88-259  CoarseMatching.forward
  networks/loftr/submodules/fine_preprocess.py:29-59, utils/fine_matching.py:15-74
  networks/loftr/loftr.py:43-91                   LoFTR.forward (end to end)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402
import loftr_oracle as O  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")


def close(a, b, tol=1e-5, what=""):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.dtype in (torch.int64, torch.bool, torch.int32):
        assert torch.equal(a, b), what
        return 0.0
    err = (a - b).abs().max().item() if a.numel() else 0.0
    scale = max(1.0, b.abs().max().item()) if b.numel() else 1.0
    assert err <= tol * scale, (what, err, scale)
    return err


def main():
    torch.manual_seed(0)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    cfg = ref_shims.reference_loftr_config()
    from networks.loftr.loftr import LoFTR
    os.makedirs(GOLDEN, exist_ok=True)

    sd = O.make_state_dict(seed=0)
    ref = LoFTR(cfg).eval()
    ref.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    assert list(ref.state_dict().keys()) == list(sd.keys())
    report = {}

    with torch.no_grad():
        # ---- backbone ---------------------------------------------------------------------
        c0, c1 = O.seeded_images(1, 64, 96, seed=11)
        x = torch.cat([c0, c1], 0)
        rc, rf = ref.backbone(x)
        oc, of = O.backbone(sd, x)
        report["backbone_c"] = close(oc, rc, 1e-5, "backbone coarse")
        report["backbone_f"] = close(of, rf, 1e-5, "backbone fine")
        np.savez(os.path.join(GOLDEN, "backbone.npz"), seed=11, hw=(64, 96),
                 coarse=rc.numpy(), fine_sub=rf[:, :, ::4, ::4].contiguous().numpy(),
                 fine_sum=rf.double().sum().item(), fine_abs=rf.double().abs().sum().item())

        # ---- position encoding ------------------------------------------------------------
        z = torch.zeros(1, 256, 60, 80)
        rpe = ref.pos_encoding(z)
        ope = O.position_encoding(256, 60, 80)
        report["posenc"] = close(ope, rpe, 0.0, "posenc")
        np.savez(os.path.join(GOLDEN, "posenc.npz"), pe_corner=rpe[0, :, :3, :5].numpy(),
                 pe_last=rpe[0, :, 59, 79].numpy(), pe_sum=rpe.double().sum().item())

        # ---- coarse transformer -----------------------------------------------------------
        g = torch.Generator().manual_seed(21)
        f0 = torch.randn(2, 48, 256, generator=g)
        f1 = torch.randn(2, 48, 256, generator=g)
        r0, r1 = ref.loftr_coarse(f0, f1)
        o0, o1 = O.local_feature_transformer(sd, "loftr_coarse", f0, f1, 8, 4)
        report["coarse_tf"] = max(close(o0, r0, 1e-5, "tf0"), close(o1, r1, 1e-5, "tf1"))
        np.savez(os.path.join(GOLDEN, "coarse_transformer.npz"), seed=21, shape=(2, 48, 256),
                 out0=r0.numpy(), out1=r1.numpy())

        # ---- coarse matching (planted correspondences; with and without scale) -------------
        hw_c = (12, 16)
        hw_i = (96, 128)
        pf0, pf1, _ = O.planted_coarse_features(2, hw_c, sigma=1.0, eps=0.5, seed=31)
        for tag, extra in (("plain", {}),
                           ("scaled", {"scale0": torch.tensor([[1.5, 1.25], [0.5, 2.0]]),
                                       "scale1": torch.tensor([[1.0, 3.0], [2.5, 0.75]])})):
            data = {"hw0_i": torch.Size(hw_i), "hw1_i": torch.Size(hw_i),
                    "hw0_c": torch.Size(hw_c), "hw1_c": torch.Size(hw_c), **extra}
            ref.coarse_matching(pf0, pf1, data)
            conf = O.conf_matrix_dual_softmax(pf0, pf1, 0.1)
            close(conf, data["conf_matrix"], 1e-6, "conf")
            om = O.get_coarse_match(conf, hw_i, hw_i, hw_c, hw_c, 0.2, 2,
                                    extra.get("scale0"), extra.get("scale1"))
            for k in ("b_ids", "i_ids", "j_ids", "m_bids", "mkpts0_c", "mkpts1_c", "mconf", "gt_mask"):
                close(om[k], data[k], 1e-6, "cm " + k)
            report["coarse_match_M_" + tag] = int(data["b_ids"].numel())
            np.savez(os.path.join(GOLDEN, f"coarse_match_{tag}.npz"), seed=31, hw_c=hw_c, hw_i=hw_i,
                     sigma=1.0, eps=0.5,
                     **{k: v.numpy() for k, v in extra.items()},
                     **{k: data[k].numpy() for k in
                        ("b_ids", "i_ids", "j_ids", "m_bids", "mkpts0_c", "mkpts1_c", "mconf")})

        # ---- fine level (preprocess + fine transformer + fine matching) ---------------------
        g = torch.Generator().manual_seed(41)
        hw_f = (48, 64)
        ff0 = torch.randn(2, 128, *hw_f, generator=g)
        ff1 = torch.randn(2, 128, *hw_f, generator=g)
        data = {"hw0_i": torch.Size(hw_i), "hw1_i": torch.Size(hw_i),
                "hw0_c": torch.Size(hw_c), "hw1_c": torch.Size(hw_c),
                "hw0_f": torch.Size(hw_f), "hw1_f": torch.Size(hw_f),
                "scale0": torch.tensor([[1.5, 1.25], [0.5, 2.0]]),
                "scale1": torch.tensor([[1.0, 3.0], [2.5, 0.75]])}
        ref.coarse_matching(pf0, pf1, data)
        # move a few matches onto the map border so the zero-padded unfold windows are exercised
        data["i_ids"][:3] = torch.tensor([0, 15, 16 * 11])
        data["j_ids"][:3] = torch.tensor([16 * 12 - 1, 3, 16 * 5])
        u0, u1 = ref.fine_preprocess(ff0, ff1, None, None, data)
        ou0, ou1 = O.fine_preprocess(ff0, ff1, data["b_ids"], data["i_ids"], data["j_ids"], hw_c, hw_f, 5)
        close(ou0, u0, 0.0, "unfold0"); close(ou1, u1, 0.0, "unfold1")
        t0, t1 = ref.loftr_fine(u0, u1)
        ot0, ot1 = O.local_feature_transformer(sd, "loftr_fine", ou0, ou1, 8, 1)
        report["fine_tf"] = max(close(ot0, t0, 1e-5, "ftf0"), close(ot1, t1, 1e-5, "ftf1"))
        ref.fine_matching(t0, t1, data)
        ofm = O.fine_matching(ot0, ot1, data["mkpts0_c"], data["mkpts1_c"], data["b_ids"],
                              len(data["mconf"]), hw_i, hw_f, data["scale1"], True)
        for k in ("expec_f", "mkpts0_f", "mkpts1_f"):
            report["fine_" + k] = close(ofm[k], data[k], 1e-5, k)
        np.savez(os.path.join(GOLDEN, "fine.npz"), seed=41, hw_f=hw_f,
                 i_ids=data["i_ids"].numpy(), j_ids=data["j_ids"].numpy(), b_ids=data["b_ids"].numpy(),
                 unfold0_first=u0[:4].numpy(), expec_f=data["expec_f"].numpy(),
                 mkpts0_f=data["mkpts0_f"].numpy(), mkpts1_f=data["mkpts1_f"].numpy())

        # ---- padding masks: masked coarse transformer + masked coarse matching (mask_border_with_padding) ----
        def pad_mask(n, h, w, valid):  # bottom/right padding like datasets/utils.py pad_bottom_right
            m = torch.zeros(n, h, w, dtype=torch.bool)
            for b, (vh, vw) in enumerate(valid):
                m[b, :vh, :vw] = True
            return m
        m0 = pad_mask(2, *hw_c, [(12, 13), (9, 16)])
        m1 = pad_mask(2, *hw_c, [(10, 16), (12, 11)])
        g = torch.Generator().manual_seed(61)
        tf0, tf1 = torch.randn(2, 192, 256, generator=g), torch.randn(2, 192, 256, generator=g)
        rm0, rm1 = ref.loftr_coarse(tf0, tf1, m0.flatten(-2), m1.flatten(-2))
        om0, om1 = O.local_feature_transformer(sd, "loftr_coarse", tf0, tf1, 8, 4, m0.flatten(-2), m1.flatten(-2))
        report["masked_tf"] = max(close(om0, rm0, 1e-5, "mtf0"), close(om1, rm1, 1e-5, "mtf1"))
        data = {"hw0_i": torch.Size(hw_i), "hw1_i": torch.Size(hw_i), "hw0_c": torch.Size(hw_c),
                "hw1_c": torch.Size(hw_c), "mask0": m0, "mask1": m1}
        ref.coarse_matching(pf0, pf1, data, mask_c0=m0.flatten(-2), mask_c1=m1.flatten(-2))
        confm = O.conf_matrix_dual_softmax(pf0, pf1, 0.1, m0.flatten(-2), m1.flatten(-2))
        close(confm, data["conf_matrix"], 1e-6, "masked conf")
        omm = O.get_coarse_match(confm, hw_i, hw_i, hw_c, hw_c, 0.2, 2, None, None, m0, m1)
        for k in ("b_ids", "i_ids", "j_ids", "mkpts0_c", "mkpts1_c", "mconf"):
            close(omm[k], data[k], 1e-6, "masked cm " + k)
        report["masked_coarse_M"] = int(data["b_ids"].numel())
        np.savez(os.path.join(GOLDEN, "masked.npz"), seed_tf=61, seed_cm=31, hw_c=hw_c, hw_i=hw_i,
                 valid0=np.array([(12, 13), (9, 16)]), valid1=np.array([(10, 16), (12, 11)]),
                 tf_out0_sub=rm0[:, ::4].numpy(), tf_out1_sub=rm1[:, ::4].numpy(),
                 **{k: data[k].numpy() for k in ("b_ids", "i_ids", "j_ids", "mkpts0_c", "mkpts1_c", "mconf")})

        # ---- end to end (plumbing: keys, dtypes, shapes, few matches with random weights) ----
        for tag, (h, w) in (("e2e_64x96", (64, 96)), ("e2e_96x128", (96, 128))):
            c0, c1 = O.seeded_images(2, h, w, seed=51)
            d_ref = {"image0": c0[:, :1], "image1": c1[:, :1], "color0": c0, "color1": c1}
            ref(d_ref)
            d_or = O.loftr_forward(sd, {"image0": c0[:, :1], "image1": c1[:, :1], "color0": c0, "color1": c1})
            keys = ("b_ids", "i_ids", "j_ids", "m_bids", "mkpts0_c", "mkpts1_c", "mconf", "expec_f",
                    "mkpts0_f", "mkpts1_f")
            for k in keys:
                close(d_or[k], d_ref[k], 1e-5, tag + " " + k)
            close(d_or["conf_matrix"], d_ref["conf_matrix"], 1e-5, tag + " conf")
            report[tag + "_M"] = int(d_ref["b_ids"].numel())
            cm = d_ref["conf_matrix"]
            np.savez(os.path.join(GOLDEN, tag + ".npz"), seed=51, hw=(h, w),
                     conf_rowmax=cm.max(dim=2)[0].numpy(), conf_colmax=cm.max(dim=1)[0].numpy(),
                     key_order=np.array([k for k in d_ref.keys()]),
                     **{k: d_ref[k].numpy() for k in keys})

        # ---- end to end with padding masks ------------------------------------------------------------
        c0, c1 = O.seeded_images(2, 64, 96, seed=71)
        mm0, mm1 = pad_mask(2, 8, 12, [(8, 9), (6, 12)]), pad_mask(2, 8, 12, [(7, 12), (8, 10)])
        c0 = c0 * torch.nn.functional.interpolate(mm0[:, None].float(), scale_factor=8)   # zero-padded pixels
        c1 = c1 * torch.nn.functional.interpolate(mm1[:, None].float(), scale_factor=8)
        d_ref = {"image0": c0[:, :1], "image1": c1[:, :1], "color0": c0, "color1": c1, "mask0": mm0, "mask1": mm1}
        ref(d_ref)
        d_or = O.loftr_forward(sd, {"image0": c0[:, :1], "image1": c1[:, :1], "color0": c0, "color1": c1,
                                    "mask0": mm0, "mask1": mm1})
        for k in ("b_ids", "i_ids", "j_ids", "mconf", "mkpts0_f", "mkpts1_f"):
            close(d_or[k], d_ref[k], 1e-5, "e2e masked " + k)
        close(d_or["conf_matrix"], d_ref["conf_matrix"], 1e-5, "e2e masked conf")
        report["e2e_masked_M"] = int(d_ref["b_ids"].numel())
        cmm = d_ref["conf_matrix"]
        np.savez(os.path.join(GOLDEN, "e2e_masked.npz"), seed=71, hw=(64, 96),
                 valid0=np.array([(8, 9), (6, 12)]), valid1=np.array([(7, 12), (8, 10)]),
                 conf_rowmax=cmm.max(dim=2)[0].numpy(), conf_colmax=cmm.max(dim=1)[0].numpy(),
                 **{k: d_ref[k].numpy() for k in ("b_ids", "i_ids", "j_ids", "mconf", "mkpts0_f", "mkpts1_f")})

    for k, v in report.items():
        print(f"{k:28s} {v}")
    print("golden vectors written to", GOLDEN)


if __name__ == "__main__":
    main()
