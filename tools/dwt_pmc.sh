#!/bin/bash
# HBM traffic of the DWT launches (profiles/r2_dwt_pmc.json): two separate rocprofv3 counter passes
# (--kernel-trace + one --pmc counter each, as MI355X_MICROARCH.md prescribes), one frame at a time.
# usage (on a GPU box, from the repo root): tools/dwt_pmc.sh  -> writes gpurun_out/<round>_dwt_pmc.json
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
export R=${1:-r3}
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$C
  (cd /tmp && rocprofv3 --kernel-trace --pmc $C -d $ROOT/gpurun_out/pmc_$C -o p --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --no-host-path --no-rate-control --no-dwt-replay > $ROOT/gpurun_out/pmc_$C.log 2>&1)
done
python3 - <<'PY'
import csv, json, collections, re
def collect(counter):
    per = collections.defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(f"gpurun_out/pmc_{counter}/p_counter_collection.csv")):
        m = re.search(r"(dwt_fused_kernel|dwt_level_kernel)", r["Kernel_Name"])
        if m and r["Counter_Name"] == counter:
            per[m.group(1)][0] += float(r["Counter_Value"]); per[m.group(1)][1].add(r["Dispatch_Id"])
    return per
f, w = collect("FETCH_SIZE"), collect("WRITE_SIZE")
frames = len(f["dwt_fused_kernel"][1])
out = {"source": "tools/dwt_pmc.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python3 bench.py --steps 2 --warmup 1 --inflight 1, MI355X gfx950 ROCm 7.2",
       "workload": "8192x8192 16-bit RGB, 9/7 + ICT, 5 levels (fused front end + level 1, then 4 level launches)",
       "units": "counter values are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads)",
       "per_frame": {}}
total = 0.0
for k in ("dwt_fused_kernel", "dwt_level_kernel"):
    fk, wk = f[k][0] / frames, w[k][0] / frames
    out["per_frame"][k] = {"launches": len(f[k][1]) // frames, "FETCH_SIZE_KiB": round(fk, 2), "WRITE_SIZE_KiB": round(wk, 2)}
    total += (2 * fk + wk) * 1024
out["hbm_bytes_per_frame"] = total
out["hbm_bytes_per_launch"] = total / 5
fk, wk = f["dwt_fused_kernel"][0] / frames, w["dwt_fused_kernel"][0] / frames
out["fused_hbm_bytes"] = (2 * fk + wk) * 1024  # the dominant launch (level 1 + front end): what bench.py reports as roofline.traffic
out["fused_algorithmic_bytes"] = 8.0 * 3 * 8192 * 8192
out["fused_minimum_bytes"] = (8.0 + 12.0) * 8192 * 8192  # 8 B/pixel ARGB64 read + 3 x 4 B written
out["algorithmic_bytes_per_frame"] = 8.0 * 3 * 8192 * 8192 * sum(0.25 ** l for l in range(5))
import os
json.dump(out, open("gpurun_out/%s_dwt_pmc.json" % os.environ.get("R", "r3"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
