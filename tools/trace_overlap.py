"""Reads a rocprofv3 kernel_trace.csv of a bench.py run and reports, for the steps FIRST..LAST (counted
by the launches of the batched forward preprocess, 0-based), how much of the time no / one / two / ...
kernels were running and which kernels were running alone.  Development tool.
usage: python tools/trace_overlap.py path/to/kernel_trace.csv FIRST LAST"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
first, last_step = int(sys.argv[2]), int(sys.argv[3])
marks = sorted(e[0] for e in ev if "preprocess_fwd_batch" in e[2])
t_a, t_b = marks[first], marks[last_step + 1]
ev = [e for e in ev if t_a <= e[0] < t_b]
print("steps %d..%d: %d kernels, %.2f ms per step" % (first, last_step, len(ev), (t_b - t_a) / 1e6 / (last_step + 1 - first)))
pts = []
for i, (a, b, n) in enumerate(ev):
    pts.append((a, 1, i))
    pts.append((b, -1, i))
pts.sort()
active = set()
hist = defaultdict(int)
alone = defaultdict(int)
last = pts[0][0]
for t, d, i in pts:
    dt = t - last
    if dt > 0:
        hist[len(active)] += dt
        if len(active) == 1:
            alone[ev[next(iter(active))][2][:60]] += dt
    last = t
    if d > 0:
        active.add(i)
    else:
        active.discard(i)
tot = sum(hist.values())
print("span %.1f ms; kernels running at once: " % (tot / 1e6) + ", ".join("%d: %.1f%%" % (k, 100.0 * v / tot) for k, v in sorted(hist.items())))
print("sum of kernel durations / span = %.2f" % (sum(b - a for a, b, _ in ev) / tot))
print("running alone (ms):")
for n, v in sorted(alone.items(), key=lambda x: -x[1])[:14]:
    print("  %-62s %.2f" % (n, v / 1e6))
