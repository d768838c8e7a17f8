#!/bin/bash
# Kernel timeline of the band-pipelined synchronous call (tools/band_probe.py, 3 frames of one `bands` setting) by
# rocprofv3 --kernel-trace; tools/band_trace.py prints the last frame's launches relative to its first one.
# usage (repo root, GPU box): tools/band_trace.sh [bands] [size]
set -e
B=${1:-8}; S=${2:-8192}
cd "$(dirname "$0")/.."; ROOT=$PWD; export TMPDIR=/tmp
rm -rf gpurun_out/band_trace
(cd /tmp && rocprofv3 --kernel-trace -d $ROOT/gpurun_out/band_trace -o k --output-format csv -- python3 $ROOT/tools/band_probe.py $S 2 $B > $ROOT/gpurun_out/band_trace.log 2>&1)
python3 tools/band_trace.py $(find gpurun_out/band_trace -name k_kernel_trace.csv | head -1) > gpurun_out/r4_band_trace_b${B}.txt
tail -60 gpurun_out/r4_band_trace_b${B}.txt
