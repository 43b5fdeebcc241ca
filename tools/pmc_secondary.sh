#!/bin/bash
# HBM traffic per call of the secondary engines (gim_dkm match() 672x896 -> 1152x1536, gim_roma match() at 560 and 672 -> 1344, gim_lightglue
# batch 8): separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE; kernel-trace only), FETCH x2 gfx950 correction, only the dispatches behind
# the marker kernel of tools/prof_secondary.py.   tools/pmc_secondary.sh <tag> [engines...]  -> gpurun_out/traffic_secondary_<tag>.json
tag=${1:-r06}; shift
engines=${@:-dkm roma560 roma672 lightglue}
root=$GRAFT_REPO_ROOT
for e in $engines; do
  for c in FETCH_SIZE WRITE_SIZE; do
    out=$root/gpurun_out/pmcs_${e}_$c; rm -rf $out; mkdir -p $out
    ( cd /tmp && TMPDIR=/tmp timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -o p -- python $root/tools/prof_secondary.py $e 2 ) > $out/log.txt 2>&1
  done
done
python - "$root" "$tag" $engines <<'PY'
import csv, glob, json, sys, collections
root, tag, engines = sys.argv[1], sys.argv[2], sys.argv[3:]
res = {}
for e in engines:
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"{root}/gpurun_out/pmcs_{e}_{c}/**/*counter_collection.csv", recursive=True)
        if not f:
            print("missing", e, c, open(f"{root}/gpurun_out/pmcs_{e}_{c}/log.txt").read()[-600:]); tot = None; break
        rows = [r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == c]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        mark = max((i for i, r in enumerate(rows) if "i0e" in r["Kernel_Name"]), default=None)
        if mark is None:
            print("no marker", e, c); tot = None; break
        tot[c] = (sum(float(r["Counter_Value"]) for r in rows[mark + 1:]), len(rows) - mark - 1)
    if tot:
        calls = 2
        fetch = tot["FETCH_SIZE"][0] * 1024 * 2.0 / calls    # gfx950: FETCH_SIZE counts 64 B per 128-B request
        write = tot["WRITE_SIZE"][0] * 1024 / calls
        res[e] = {"hbm_bytes_per_call": round(fetch + write), "fetch_bytes_per_call": round(fetch), "write_bytes_per_call": round(write),
                  "dispatches_per_call": tot["FETCH_SIZE"][1] // calls}
out = {"what": "HBM bytes per call from the TCC counters, separate rocprofv3 --pmc passes (FETCH_SIZE KiB x 1024 x 2 -- gfx950 counts 64 B per 128-B request --, WRITE_SIZE KiB x 1024), "
               "2 calls behind a marker kernel (tools/prof_secondary.py, tools/pmc_secondary.sh); dkm / roma: one match() of one pair incl. the upsampling pass; lightglue: one batch of 8 pairs",
       "engines": res}
json.dump(out, open(f"{root}/gpurun_out/traffic_secondary_{tag}.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
