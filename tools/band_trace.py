#!/usr/bin/env python3
"""The launches of the last frame in a rocprofv3 kernel trace (CSV), times in ms relative to the frame's first launch."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
# frames are separated by gaps of > 3 ms without a launch start after the previous frame's first DWT launch
starts = [i for i, (s, e, n) in enumerate(ev) if "dwt_fused" in n and (i == 0 or s - max(x[1] for x in ev[:i]) > 500_000)]
first = starts[-1] if starts else 0
t0 = ev[first][0]
short = lambda n: n.replace("j2k_hip::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
for s, e, n in ev[first:]:
    print(f"{(s - t0) / 1e6:8.3f} -> {(e - t0) / 1e6:8.3f}  ({(e - s) / 1e3:9.1f} us)  {short(n)}")
