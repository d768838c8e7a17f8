#!/bin/bash
# VALU wave-instructions of the context modeller and the coder per C3 frame (rocprofv3 --pmc SQ_INSTS_VALU)
cd "$(dirname "$0")/.."; ROOT=$PWD; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU -d $ROOT/gpurun_out/pmcK -o p --output-format csv -- python3 $ROOT/bench.py --inflight 1 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
python3 - <<PY
import csv,collections,re
agg=collections.defaultdict(float); d=collections.defaultdict(set)
for r in csv.DictReader(open("$ROOT/gpurun_out/pmcK/p_counter_collection.csv")):
    m = re.search(r"(t1_model_kernel|t1_mq2_kernel|dwt_fused_kernel)", r["Kernel_Name"])
    if m: agg[m.group(1)]+=float(r["Counter_Value"]); d[m.group(1)].add(r["Dispatch_Id"])
frames = len(d["dwt_fused_kernel"])
for k in agg: print("  %s VALU per frame %.3e" % (k, agg[k]/frames))
PY
