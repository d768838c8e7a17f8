#!/bin/bash
# Where the Tier-1 kernels spend their wave-cycles (one frame at a time): SQ counters per kernel.
# usage (GPU box, repo root): tools/t1_pmc.sh  -> gpurun_out/r2_t1_pmc.txt
cd "$(dirname "$0")/.."; ROOT=$PWD; export TMPDIR=/tmp
export R=${1:-r3}
rm -rf gpurun_out/pmc_t1a gpurun_out/pmc_t1b
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d $ROOT/gpurun_out/pmc_t1a -o p --output-format csv -- python3 $ROOT/bench.py --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --no-rate-control --no-dwt-replay > $ROOT/gpurun_out/pmc_t1a.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $ROOT/gpurun_out/pmc_t1b -o p --output-format csv -- python3 $ROOT/bench.py --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --no-rate-control --no-dwt-replay > $ROOT/gpurun_out/pmc_t1b.log 2>&1)
python3 - <<'PY' | tee gpurun_out/${R}_t1_pmc.txt
import csv, collections, re, glob, json, os
valu = {}
for d in ("gpurun_out/pmc_t1a", "gpurun_out/pmc_t1b"):
    f = glob.glob(d + "/**/p_counter_collection.csv", recursive=True)
    if not f:
        print("no counters in", d); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        m = re.search(r"(t1_model_kernel|t1_mq2_kernel|t1_mq_scalar_kernel|dwt_fused_kernel|dwt_level_kernel|gather_kernel)", r["Kernel_Name"])
        if m:
            agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"]); n[m.group(1)].add(r["Dispatch_Id"])
    frames = max(1, len(n["dwt_fused_kernel"]))
    for k, c in agg.items():
        print(k, "launches/frame", len(n[k]) // frames, {a: "%.3e" % (b / frames) for a, b in sorted(c.items())})
        if "SQ_INSTS_VALU" in c and k.startswith("t1_"):
            valu[k] = c["SQ_INSTS_VALU"] / frames
if valu:
    json.dump({"source": "tools/t1_pmc.sh (rocprofv3 --pmc SQ_INSTS_VALU, one frame at a time, metric frame)",
               **{k + "_valu": v for k, v in valu.items()}, "valu_per_frame": sum(valu.values())},
              open("gpurun_out/%s_t1_pmc.json" % os.environ.get("R", "r3"), "w"), indent=1)
PY
