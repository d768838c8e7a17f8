#!/bin/bash
# Everything profiles/ holds for a round, in one GPU call (repo root):  tools/profile_round.sh r2
#   <r>_bench_default.log            the JSON line of the default `python bench.py`
#   <r>_bench_c3_kernel_stats.csv    rocprofv3 --kernel-trace --stats of the same command (live: 3 frames in flight)
#   <r>_bench_c3_alone_kernel_stats.csv   same with J2K_NO_OVERLAP=1 --inflight 1 (one frame at a time)
#   <r>_dwt_pmc.json                 tools/dwt_pmc.sh (FETCH_SIZE / WRITE_SIZE passes)
#   <r>_t1_pmc.txt                   tools/t1_pmc.sh (SQ counters of the Tier-1 kernels)
#   <r>_rate_bench.txt               tools/rate_bench.py + tools/rate_inflight.py (rate-controlled encode of the metric frame)
set -e
R=${1:-r3}
cd "$(dirname "$0")/.."; ROOT=$PWD; export TMPDIR=/tmp
mkdir -p gpurun_out
python3 bench.py > gpurun_out/${R}_bench_default.log 2> gpurun_out/${R}_bench_default.err
rm -rf gpurun_out/${R}_ks gpurun_out/${R}_ks_alone
(cd /tmp && rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/${R}_ks -o k --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --no-host-path --no-rate-control --no-dwt-replay > $ROOT/gpurun_out/${R}_bench_prof.log 2>&1)
(cd /tmp && J2K_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/${R}_ks_alone -o k --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --no-host-path --no-rate-control --no-dwt-replay --inflight 1 --steps 8 --warmup 2 > $ROOT/gpurun_out/${R}_bench_prof_alone.log 2>&1)
cp $(find gpurun_out/${R}_ks -name k_kernel_stats.csv | head -1) gpurun_out/${R}_bench_c3_kernel_stats.csv
cp $(find gpurun_out/${R}_ks_alone -name k_kernel_stats.csv | head -1) gpurun_out/${R}_bench_c3_alone_kernel_stats.csv
tools/dwt_pmc.sh $R > gpurun_out/${R}_dwt_pmc.log 2>&1
tools/t1_pmc.sh $R > /dev/null 2>&1
(python3 tools/rate_bench.py 8192 20; python3 tools/rate_inflight.py 3 4 5 6) 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_rate_bench.txt
tail -1 gpurun_out/${R}_bench_default.log | cut -c1-2200
cat gpurun_out/${R}_bench_c3_kernel_stats.csv | cut -c1-160
