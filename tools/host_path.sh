#!/bin/bash
# PCIe-inclusive plug-in path through the C ABI (pageable host frame -> sink), 1..N host threads.
# usage: tools/host_path.sh [size] [thread counts...]
set -e
S=${1:-8192}; shift || true
TH=${@:-1 2 3 4}
cd "$(dirname "$0")/.."
python - <<PY
import sys; sys.path.insert(0, '.')
from j2k_amd import synth
pl = synth.planes($S, $S, 3, 16, 23456); frame, lay = synth.ae_frame(pl, 16)
frame.tofile('/tmp/frame_$S.raw')
PY
g++ -O2 -std=c++17 -Iinclude tools/host_path_bench.cpp -Lj2k_amd -lj2k_hip -Wl,-rpath,$PWD/j2k_amd -lpthread -o /tmp/host_path_bench
for T in $TH; do /tmp/host_path_bench /tmp/frame_$S.raw $S $S $T 6; done
