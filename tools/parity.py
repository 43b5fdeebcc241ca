"""Match-set parity between the HIP engine and the CPU oracle (used by bench.py's `parity` field and by the
full-size GPU parity tests).  Pure bookkeeping on the two output dicts -- no model code."""
import torch


def match_set(d, b):
    sel = (d["b_ids"] == b).cpu()
    return {(int(i), int(j)) for i, j in zip(d["i_ids"].cpu()[sel].tolist(), d["j_ids"].cpu()[sel].tolist())}


def parity_vs_oracle(d_eng, d_ref, b=0, b_ref=0):
    """Engine outputs of batch element `b` against batch element `b_ref` of the oracle's dict.
    flip_rate = |engine matches XOR oracle matches| / |oracle matches|; float deviations over the common matches."""
    se, sr = match_set(d_eng, b), match_set(d_ref, b_ref)
    common = sorted(se & sr)
    out = {"oracle_matches": len(sr), "engine_matches": len(se), "common": len(common),
           "flip_rate": round(len(se ^ sr) / max(1, len(sr)), 5)}
    if common:
        def index(d, bb):
            sel = (d["b_ids"] == bb).cpu()
            ii, jj = d["i_ids"].cpu()[sel].tolist(), d["j_ids"].cpu()[sel].tolist()
            pos = {(int(i), int(j)): k for k, (i, j) in enumerate(zip(ii, jj))}
            return sel, torch.tensor([pos[c] for c in common])
        sel_e, ie = index(d_eng, b)
        sel_r, ir = index(d_ref, b_ref)
        for key, name in (("mkpts1_f", "dmkpts1_px"), ("mkpts0_f", "dmkpts0_px"), ("mconf", "dmconf")):
            e = d_eng[key].cpu()[sel_e][ie].float()
            r = d_ref[key].cpu()[sel_r][ir].float()
            out["max_abs_" + name] = round(float((e - r).abs().max()), 6)
            out["mean_abs_" + name] = round(float((e - r).abs().mean()), 6)
        if "expec_f" in d_eng and "expec_f" in d_ref:
            e = d_eng["expec_f"].cpu()[sel_e][ie].float()
            r = d_ref["expec_f"].cpu()[sel_r][ir].float()
            out["max_abs_dexpec_f"] = round(float((e - r).abs().max()), 6)
    return out


def flip_margins(d_eng, d_ref, b=0, b_ref=0, thr=0.2):
    """For every flipped match (in exactly one of the two sets): how marginal the oracle's own decision was.
    Returns a list of (i, j, in_oracle, conf, |conf - thr|, row/col runner-up gap) from the oracle's conf_matrix."""
    conf = d_ref["conf_matrix"][b_ref]
    se, sr = match_set(d_eng, b), match_set(d_ref, b_ref)
    rows = []
    for (i, j) in sorted(se ^ sr):
        c = float(conf[i, j])
        row, col = conf[i].clone(), conf[:, j].clone()
        row[j] = -1
        col[i] = -1
        gap = min(c - float(row.max()), c - float(col.max()))
        rows.append((i, j, (i, j) in sr, c, abs(c - thr), gap))
    return rows
