"""Real-checkpoint hook: "unverifiable without weights" becomes one environment variable away.

    GIM_WEIGHTS_DIR=/path/to/weights [GIM_ZEB_DIR=/path/to/zeb] python tools/real_weights_check.py [--pair a.png b.png] [--json out.json]

With `gim_loftr_50h.ckpt` (the reference's file name, demo.py:328-347) in $GIM_WEIGHTS_DIR:
  1. loads it through the engine's `load_state_dict` (prefix rules of loftr.py:93-99 / demo.py:385-400) -- strictly: a key that does
     not fit the 375-tensor surface raises;
  2. runs the demo pair (tests/golden/demo/a1.png <-> a2.png, the reference's assets/demo pair; or --pair) through the fp32 CPU
     oracle (test infrastructure: a restatement of the reference pinned by golden vectors) and through the engine in its three
     modes -- fp16 (default), bf16, fp32 -- and prints, per mode: matches, index flip rate against the oracle, max / mean
     |d mconf| and |d mkpts1|, whether the fp16 range guard tripped (the module then reports bf16);
  3. prints the activation range of every stage of the oracle forward on that pair (max |x| after every conv / linear / ReLU group)
     next to the IEEE-fp16 limit 65504: the head-room of the default mode for THIS checkpoint;
  4. with $GIM_ZEB_DIR (the reference's zeb/ directory layout, datasets/zeb.py): runs every scene found through
     `gim_amd.zeb.run_scene`, writes dumps in the reference's format and prints pose AUC@5/10/20 (needs OpenCV for RANSAC --
     trainer/lightning.py:243-275, tools/metrics.py:77-103; without it only match counts / epipolar precision are reported).
`gim_dkm_100h.ckpt`, when present, goes through `gim_amd.demo.build('gim_dkm')` + `match()` in bf16 and fp32 on the same pair and is
compared with oracle/dkm_oracle.py.
`--synthetic DIR` writes a checkpoint of seeded trained-like weights in the reference's file format into DIR first (self-test of this
tool: tests/test_gpu_real_weights.py runs it on the GPU box, where no real checkpoint exists).
"""
import argparse
import json
import os
import sys
import warnings

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

DEMO = (os.path.join(ROOT, "tests", "golden", "demo", "a1.png"), os.path.join(ROOT, "tests", "golden", "demo", "a2.png"))


def write_synthetic_checkpoint(directory):
    """seeded trained-like gim_loftr weights in the reference's checkpoint format: {'state_dict': {'model.<key>': tensor}}"""
    from tools import synth_loftr as S
    os.makedirs(directory, exist_ok=True)
    _, sd = S.synthetic_model("fp32")
    path = os.path.join(directory, "gim_loftr_50h.ckpt")
    torch.save({"state_dict": {"model." + k: v for k, v in sd.items()}}, path)
    return path


def activation_ranges(sd, data):
    """max |x| per stage of the oracle forward: conv / linear OUTPUTS grouped by the weight's owner (block level)"""
    import loftr_oracle as O
    owner = {id(v): ".".join(k.split(".")[:-1]) for k, v in sd.items() if k.endswith("weight") and v.dim() >= 2}
    rng = {}
    conv0, lin0 = F.conv2d, F.linear

    class FF:
        def __getattr__(self, n):
            return getattr(F, n)

        @staticmethod
        def conv2d(x, w, *a, **k):
            y = conv0(x, w, *a, **k)
            nm = owner.get(id(w), "?")
            rng[nm] = max(rng.get(nm, 0.0), float(y.abs().max()), float(x.abs().max()))
            return y

        @staticmethod
        def linear(x, w, *a, **k):
            y = lin0(x, w, *a, **k)
            nm = owner.get(id(w), "?")
            rng[nm] = max(rng.get(nm, 0.0), float(y.abs().max()), float(x.abs().max()))
            return y

    O.F = FF()
    try:
        with torch.no_grad():
            O.loftr_forward(sd, data)
    finally:
        O.F = F
    return rng


def check_loftr(ckpt, pair, device="cuda:0", resize_max=640):
    import loftr_oracle as O
    from gim_amd import demo
    from tools.parity import parity_vs_oracle
    rep = {"checkpoint": ckpt}
    im0, _ = demo.preprocess(demo.read_image(pair[0]), resize_max=resize_max)   # longer side <= 640 like the ZEB loaders (demo.py itself keeps the file's size)
    im1, _ = demo.preprocess(demo.read_image(pair[1]), resize_max=resize_max)
    c0, c1 = im0[None], im1[None]
    rep["pair"] = [os.path.basename(p) for p in pair]
    rep["image_size"] = [list(c0.shape[2:]), list(c1.shape[2:])]

    def data(dev=None):
        d = {"image0": c0[:, :1], "image1": c1[:, :1], "color0": c0, "color1": c1}
        return {k: v.to(dev) for k, v in d.items()} if dev else d

    sd = demo._load_ckpt(ckpt)
    sd = {(k.replace("model.", "", 1) if k.startswith("model.") else k).replace("matcher.", "", 1): v.float() for k, v in sd.items()}
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    with torch.no_grad():
        ref = O.loftr_forward(sd, data())
    rep["oracle_matches"] = int(ref["b_ids"].numel())
    rng = activation_ranges(sd, data())
    top = sorted(rng.items(), key=lambda kv: -kv[1])[:8]
    rep["activation_max"] = {"overall": max(rng.values()), "fp16_limit": 65504.0, "headroom_x": round(65504.0 / max(rng.values()), 1),
                             "largest": [[k, round(v, 2)] for k, v in top]}
    rep["modes"] = {}
    for prec in ("fp16", "bf16", "fp32"):
        model, _ = demo.build("gim_loftr", ckpt, prec, device=device)
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            for _ in range(2):
                d = data(device)
                model(d)
            torch.cuda.synchronize()
        p = parity_vs_oracle(d, ref, 0, 0)
        p.update({"ran_as": model.precision, "fp16_range_guard_tripped": bool(model.fp16_overflowed),
                  "warnings": [str(w.message)[:160] for w in rec][:3],
                  "finite": bool(torch.isfinite(d["mconf"]).all() and torch.isfinite(d["mkpts1_f"]).all())})
        rep["modes"][prec] = p
        del model
    return rep


def check_dkm(ckpt, pair, device="cuda:0"):
    import dkm_oracle as DO
    from gim_amd import demo
    im0, _ = demo.preprocess(demo.read_image(pair[0]))
    im1, _ = demo.preprocess(demo.read_image(pair[1]))
    rep = {"checkpoint": ckpt, "modes": {}}
    sd = demo._load_ckpt(ckpt)
    sd = {(k.replace("model.", "", 1) if k.startswith("model.") else k): v.float() for k, v in sd.items() if "encoder.net.fc" not in k}
    with torch.no_grad():
        wref, cref = DO.match(sd, im0[None], im1[None], 672, 896, (1152, 1536))
    for prec in ("bf16", "fp32"):
        model, _ = demo.build("gim_dkm", ckpt, prec, device=device)
        model.upsample_res = (1152, 1536)
        warp, cert = model.match(im0[None].to(device), im1[None].to(device))
        rep["modes"][prec] = {"max_abs_dwarp": float((warp.cpu() - wref).abs().max()), "mean_abs_dwarp": float((warp.cpu() - wref).abs().mean()),
                              "max_abs_dcertainty": float((cert.cpu() - cref).abs().max())}
        del model
    return rep


def run_zeb(ckpt, zeb_dir, out_dir, device="cuda:0", precision="fp16", scenes=None, max_pairs=None):
    from gim_amd import demo, zeb
    from gim_amd.zeb_data import ZebScene, collate
    model, _ = demo.build("gim_loftr", ckpt, precision, device=device)
    have_cv2 = True
    try:
        import cv2  # noqa: F401
    except ImportError:
        have_cv2 = False
    rep = {"opencv": have_cv2, "scenes": {}}
    est = None if have_cv2 else (lambda a, b, k0, k1: None)    # without RANSAC every pose error is inf: AUC 0, match statistics still valid

    def matcher(batch):
        for k, v in batch.items():
            if torch.is_tensor(v):
                batch[k] = v.to(device)
        model(batch)

    for scene in (scenes or zeb.DATASETS):
        if not os.path.isdir(os.path.join(zeb_dir, scene)):
            continue
        ds = ZebScene(zeb_dir, scene, max_resize=640, df=8, padding=False)   # TEST_GIM_LOFTR.sh: --max_resize 640? see SURVEY 3.2
        n = len(ds) if max_pairs is None else min(len(ds), max_pairs)
        batches = (collate([ds[i]]) for i in range(n))
        out = zeb.dump_path(out_dir, "gim_loftr_hip", scene, "real")
        rows = zeb.run_scene(matcher, batches, out, estimate=est, skip_existing=False)
        cols = zeb.read_dump(out)
        rep["scenes"][scene] = {"pairs": len(rows), "mean_matches": float(sum(float(v) for v in cols["Bef.Num"]) / max(1, len(rows))),
                                "mean_epipolar_precision": float(sum(float(v) for v in cols["Bef.Prec"]) / max(1, len(rows)))}
    if rep["scenes"]:
        per, mean = zeb.score_dir(out_dir, "gim_loftr_hip", "real")
        rep["auc"] = {"per_scene": per, "mean": mean}
    return rep


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--pair", nargs=2, default=list(DEMO))
    ap.add_argument("--json", default=None)
    ap.add_argument("--synthetic", default=None, metavar="DIR", help="write a seeded checkpoint into DIR and use it as GIM_WEIGHTS_DIR")
    ap.add_argument("--zeb-pairs", type=int, default=None, help="at most this many pairs per ZEB scene")
    ap.add_argument("--resize-max", type=int, default=640, help="longer image side of the gim_loftr check (0: keep the file's size)")
    args = ap.parse_args(argv)
    wdir = os.environ.get("GIM_WEIGHTS_DIR")
    if args.synthetic:
        write_synthetic_checkpoint(args.synthetic)
        wdir = args.synthetic
    if not wdir:
        print("real_weights_check: GIM_WEIGHTS_DIR is not set -- nothing to do (the reference ships no checkpoint; "
              "put gim_loftr_50h.ckpt / gim_dkm_100h.ckpt there)")
        return None
    rep = {"weights_dir": wdir}
    lo = os.path.join(wdir, "gim_loftr_50h.ckpt")
    if os.path.exists(lo):
        rep["gim_loftr"] = check_loftr(lo, args.pair, resize_max=args.resize_max or None)
        zdir = os.environ.get("GIM_ZEB_DIR")
        if zdir:
            rep["zeb"] = run_zeb(lo, zdir, os.path.join(wdir, "zeb_dump_hip"), max_pairs=args.zeb_pairs)
    dk = os.path.join(wdir, "gim_dkm_100h.ckpt")
    if os.path.exists(dk):
        rep["gim_dkm"] = check_dkm(dk, args.pair)
    txt = json.dumps(rep, indent=1, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))
    print(txt)
    if args.json:
        open(args.json, "w").write(txt)
    return rep


if __name__ == "__main__":
    main()
