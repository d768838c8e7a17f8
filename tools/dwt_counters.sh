#!/bin/bash
# What binds the DWT launches: L2 hit / miss, requests and stalls towards the fabric, address translation
# in the vector L1, and the waves' own account (issue, wait), per kernel, one frame at a time.  Separate rocprofv3 passes of
# --kernel-trace + --pmc (a few counters each), as MI355X_MICROARCH.md prescribes.
# PASSES="1 7" limits the run to those passes.
# usage (GPU box, repo root): tools/dwt_counters.sh [round]  -> gpurun_out/<round>_dwt_counters.txt
cd "$(dirname "$0")/.."; ROOT=$PWD; export TMPDIR=/tmp
export R=${1:-r4}
CMD="python3 $ROOT/tools/one_frame.py 8192 2"
i=0
for SET in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
           "TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT" "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM"; do
  i=$((i+1))
  if [ -n "$PASSES" ] && ! echo " $PASSES " | grep -q " $i "; then continue; fi
  rm -rf gpurun_out/pmc_dwt$i
  (cd /tmp && timeout -k 5 90 rocprofv3 --kernel-trace --pmc $SET -d $ROOT/gpurun_out/pmc_dwt$i -o p --output-format csv -- $CMD > $ROOT/gpurun_out/pmc_dwt$i.log 2>&1) || echo "pass $i ($SET) failed: $(tail -2 gpurun_out/pmc_dwt$i.log)"
done
python3 - <<'PY' | tee gpurun_out/${R}_dwt_counters.txt
import csv, collections, re, glob
print("# tools/dwt_counters.sh: per frame (metric frame, one at a time), per kernel; SQ_* in quad-cycles, TCC_EA requests of 32/64/128 B as the counter says")
for d in sorted(glob.glob("gpurun_out/pmc_dwt[0-9]*")):
    f = glob.glob(d + "/**/p_counter_collection.csv", recursive=True)
    if not f:
        print("no counters in", d); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        m = re.search(r"(dwt_fused_kernel|dwt_level_kernel)", r["Kernel_Name"])
        if m:
            agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"]); n[m.group(1)].add(r["Dispatch_Id"])
    frames = max(1, len(n["dwt_fused_kernel"]))
    for k, c in sorted(agg.items()):
        print(k, "launches/frame", len(n[k]) // frames, {a: "%.4g" % (b / frames) for a, b in sorted(c.items())})
PY
