"""TEST INFRASTRUCTURE ONLY -- pins gim_amd/zeb.py to the reference's scorer.

Takes a deterministic sample of rows of the per-pair dumps the reference ships (`/root/reference/dump/zeb`,
written by `trainer/lightning.py:258-275`), scores them with the reference's OWN `analysis.error_auc`
(`analysis.py:34-53`, thresholds extended to 5/10/20 deg) and stores sample + expected values under
tests/golden/zeb/.  Run by hand in the authoring container."""
import importlib.util
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "zeb")

spec = importlib.util.spec_from_file_location("ref_analysis", os.path.join(REF, "analysis.py"))
ana = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ana)

os.makedirs(OUT, exist_ok=True)
expected = {}
for scene, stride in (("GL3D", 25), ("KITTI", 40), ("RobotcarNight", 15)):
    name = f"[T] gim_loftr {scene:>15} 50h.txt"
    lines = open(os.path.join(REF, "dump", "zeb", name)).readlines()
    rows = lines[1::stride]
    rows = rows + rows[:3]  # duplicates: the scorer must keep the first occurrence only
    with open(os.path.join(OUT, name), "w") as f:
        f.write(lines[0])
        f.writelines(rows)
    seen, R, T = set(), [], []
    for r in rows:
        x = r.split()
        if x[0] in seen:
            continue
        seen.add(x[0])
        R.append(float(x[3])); T.append(float(x[4]))
    auc = ana.error_auc(R, T, ["5.0°", "10.0°", "20.0°"], "auc")
    expected[scene] = {"n": len(R), "auc5": auc["auc@ 5.0°"], "auc10": auc["auc@ 10.0°"], "auc20": auc["auc@ 20.0°"]}
json.dump(expected, open(os.path.join(OUT, "expected.json"), "w"), indent=1)
print(json.dumps(expected, indent=1))
