#!/bin/bash
# usage: tools/dwt_sweep.sh VAR v1 v2 ...   -> bench DWT time / roofline frac for each value of an env knob
VAR=$1; shift
for V in "$@"; do
  env $VAR=$V python bench.py --steps 6 --warmup 2 --inflight ${INFLIGHT:-1} --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$V', 'dwt_ms', d['stages_ms']['ms_dwt'], 'frac', d['roofline']['frac'], 'value', d['value'])"
done
