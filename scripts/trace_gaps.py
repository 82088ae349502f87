#!/usr/bin/env python3
"""Busy time against gaps in a rocprofv3 kernel trace: python scripts/trace_gaps.py <kernel_trace.csv> [first_fraction_to_skip]
For a launch-bound graph (the UNet at one image per call) the question is whether the time is IN the kernels or BETWEEN them: prints the
span, the sum of kernel durations, the sum and mean of the gaps (start of a kernel minus end of its predecessor on the device timeline;
negative = overlap, counted as 0) and the per-kernel-name totals."""
import collections
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5          # the first half holds set-up, warm-up and the eager recording run
rows = rows[int(len(rows) * skip):]
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
gaps = [max(0, rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1)]
big = sum(g for g in gaps if g > 200_000)                        # host-side pauses between forwards, not launch gaps
print(f"{len(rows)} kernels, span {span / 1e6:.3f} ms, in kernels {busy / 1e6:.3f} ms ({100 * busy / span:.1f} %), gaps {sum(gaps) / 1e6:.3f} ms "
      f"(of which pauses > 0.2 ms: {big / 1e6:.3f} ms); mean gap without pauses {(sum(gaps) - big) / max(1, sum(1 for g in gaps if g <= 200_000)) / 1e3:.2f} us")
acc = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    a = acc[n[:110]]
    a[0] += 1
    a[1] += e - s
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t / 1e6:9.3f} ms  {c:6d} x {t / c / 1e3:8.2f} us  {n}")
