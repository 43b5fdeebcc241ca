#!/bin/bash
# HBM traffic of the gim_conv2d_bn_act launches (igemm_* and conv3x3_halo kernels) of one batch-8 forward, from the TCC memory-side counters.
# Two separate rocprofv3 passes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2: they do not fit together), kernel-trace
# only, eager launches (GIM_FLAGS=graph=0).  Writes profiles/traffic_<tag>.json: per-launch average over the igemm kernels.
# MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a
# wide coalesced stream, i.e. reads are under-reported by 2x -> doubled here.
tag=${1:-r01}
root=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  out=$root/gpurun_out/pmc_$c; rm -rf $out; mkdir -p $out
  ( cd /tmp && TMPDIR=/tmp GIM_FLAGS=graph=0 timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -o p -- python $root/tools/prof_forward.py 3 ) > $out/log.txt 2>&1
done
python - "$root" "$tag" <<'PY'
import csv, glob, json, sys, collections
root, tag = sys.argv[1], sys.argv[2]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{root}/gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True)
    if not f:
        print("missing", c, open(f"{root}/gpurun_out/pmc_{c}/log.txt").read()[-1500:]); sys.exit(1)
    tot, n, allk = 0.0, 0, collections.defaultdict(float)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c:
            continue
        nm = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        allk[nm.split("<")[0].split("(")[0][-48:]] += float(r["Counter_Value"])
        if "igemm" in r["Kernel_Name"] or "conv3x3_halo" in r["Kernel_Name"]:  # every kernel behind gim_conv2d_bn_act
            tot += float(r["Counter_Value"]); n += 1
    res[c] = (tot, n)
    res[c + "_all"] = allk
nl = res["FETCH_SIZE"][1]
fetch_b = res["FETCH_SIZE"][0] * 1024 * 2.0   # gfx950 correction (see header)
write_b = res["WRITE_SIZE"][0] * 1024
NF = 3  # forwards in the run (tools/prof_forward.py 3)
per_kernel = {}
for k in set(res["FETCH_SIZE_all"]) | set(res["WRITE_SIZE_all"]):
    per_kernel[k] = round((res["FETCH_SIZE_all"].get(k, 0.0) * 2048 + res["WRITE_SIZE_all"].get(k, 0.0) * 1024) / NF / 1e9, 3)
out = {"workload": "gim_loftr 640x480 batch 8, the headline 16-bit mode (fp16 since round 6; the bf16 flavour moves the same bytes), match-rich synthetic pairs (bench.py workload), 3 forwards, eager launches", "igemm_launches": nl,
       "total_gb_per_forward_all_kernels": round(sum(per_kernel.values()), 2),
       "igemm_gb_per_forward": round((fetch_b + write_b) / NF / 1e9, 2),
       "gb_per_forward_by_kernel": dict(sorted(per_kernel.items(), key=lambda kv: -kv[1])[:12]),
       "fetch_bytes_per_launch": fetch_b / nl, "write_bytes_per_launch": write_b / res["WRITE_SIZE"][1],
       "traffic_bytes_per_launch": fetch_b / nl + write_b / res["WRITE_SIZE"][1],
       "fetch_correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950 counts 64 B per 128-B request)",
       "raw_FETCH_SIZE_KiB_sum": res["FETCH_SIZE"][0], "raw_WRITE_SIZE_KiB_sum": res["WRITE_SIZE"][0]}
json.dump(out, open(f"{root}/gpurun_out/traffic_{tag}.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
