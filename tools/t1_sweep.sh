#!/bin/bash
# usage: tools/t1_sweep.sh VAR v1 v2 ...  -> Tier-1 time for each value of an env knob (one frame at a time)
VAR=$1; shift
for V in "$@"; do
  env $VAR=$V python bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$V', 't1_ms', d['stages_ms']['ms_t1'], 'value', d['value'])"
done
