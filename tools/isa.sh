#!/bin/bash
# The gfx950 ISA of one kernel, as the library is built:  tools/isa.sh t1.hip t1_mq2_kernel > mq2.s
# (no GPU needed).  What round 3's last coder changes came from: instructions per decision counted in the listing,
# scalar mask juggling and loop-carried copies the source does not show (round 3; DESIGN.md section 5, Tier-1 design).
set -e
cd "$(dirname "$0")/.."
src=j2k_amd/csrc/$1; pat=$2
tmp=$(mktemp /tmp/isa.XXXXXX.s)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -S --cuda-device-only -o "$tmp" "$src" 2>/dev/null
awk -v pat="$pat" '
  /^_Z[A-Za-z0-9_]*:/ { on = index($0, pat) > 0 }
  on { print }
  on && /^\.Lfunc_end/ { on = 0 }' "$tmp"
grep -A12 "\.name: *_Z.*$pat" "$tmp" | grep "name:\|vgpr_count\|sgpr_count\|spill" | sed 's/^/; /'
rm -f "$tmp"
