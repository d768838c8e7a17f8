#!/bin/bash
# usage: tools/ov_sweep.sh "ENV1=.. ENV2=.." ...   -- one bench line (value, ms/step, DWT frac, dwt ms) per env set
for E in "$@"; do
  echo "== $E"
  env $E python bench.py --steps 24 --warmup 6 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('frames_in_flight'))"
done
