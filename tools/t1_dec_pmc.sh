#!/bin/bash
# SQ counters of the Tier-1 decode kernel (one C2-size decode): instruction mix and stall split.
# usage (GPU box, repo root): tools/t1_dec_pmc.sh [CASE]  -> gpurun_out/r2_t1_dec_pmc.txt
cd "$(dirname "$0")/.."; ROOT=$PWD; export TMPDIR=/tmp
CASE=${1:-C3}
R=${2:-r3}
rm -rf gpurun_out/pmc_dec gpurun_out/pmc_dec2
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_SCA -d $ROOT/gpurun_out/pmc_dec -o p --output-format csv -- python3 $ROOT/tools/decode_bench.py $CASE > $ROOT/gpurun_out/pmc_dec.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU -d $ROOT/gpurun_out/pmc_dec2 -o p --output-format csv -- python3 $ROOT/tools/decode_bench.py $CASE > $ROOT/gpurun_out/pmc_dec2.log 2>&1)
python3 - <<'PY' | tee gpurun_out/${R}_t1_dec_pmc.txt
import csv, collections, re, glob
for d in ("gpurun_out/pmc_dec", "gpurun_out/pmc_dec2"):
  f = glob.glob(d + "/**/p_counter_collection.csv", recursive=True)
  if not f:
    print("no counters in", d); continue
  agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
  for r in csv.DictReader(open(f[0])):
    m = re.search(r"(t1_decode_lanes_kernel|t1_decode_kernel|t1_assemble_lanes_kernel|t1_assemble_kernel|idwt_h_kernel|idwt_v_kernel|decode_output_kernel)", r["Kernel_Name"])
    if m:
        agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"]); n[m.group(1)].add(r["Dispatch_Id"])
  for k, c in agg.items():
    L = len(n[k])
    print(k, "launches", L, {a: "%.3e" % (b / L) for a, b in sorted(c.items())})
PY
