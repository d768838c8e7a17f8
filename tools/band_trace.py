#!/usr/bin/env python3
"""The launches of the last frame in a rocprofv3 kernel trace (CSV), times in ms relative to the frame's first launch."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
# a band-pipelined frame begins with its (gated) coder launch; the last one in the trace is the frame shown
starts = [i for i, (s, e, n) in enumerate(ev) if "t1_mq2_kernel" in n]
first = starts[-1] if starts else 0
while first > 0 and ev[first][0] - ev[first - 1][0] < 200_000 and "fillBuffer" in ev[first - 1][2]:  # (the memsets right before it)
    first -= 1
t0 = ev[first][0]
short = lambda n: n.replace("j2k_hip::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
for s, e, n in ev[first:]:
    print(f"{(s - t0) / 1e6:8.3f} -> {(e - t0) / 1e6:8.3f}  ({(e - s) / 1e3:9.1f} us)  {short(n)}")
