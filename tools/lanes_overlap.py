"""From a rocprofv3 kernel trace of tools/lanes_probe.py: busy time of each stream, of their union and of their overlap
over the last part of the run (the two-lane section).  usage: python tools/lanes_overlap.py <trace dir>"""
import csv
import glob
import sys
from collections import defaultdict

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        rows += [r for r in csv.DictReader(fh) if "octane::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
bys = defaultdict(list)
for r in rows:
    bys[r["Stream_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
# the two-lane section: from the first kernel of the stream that starts last
starts = {s: v[0][0] for s, v in bys.items()}
print("streams:", {s: len(v) for s, v in bys.items()})
lanes = sorted(bys, key=lambda s: len(bys[s]), reverse=True)[:2]
t0 = max(bys[lanes[1]][0][0], bys[lanes[0]][len(bys[lanes[0]]) - len(bys[lanes[1]])][0]) if len(lanes) == 2 else 0
ev = []
for s in lanes:
    for a, b in bys[s]:
        if b > t0:
            ev += [(max(a, t0), 1, s), (b, -1, s)]
ev.sort()
active = defaultdict(int)
last = t0
busy1 = busy2 = 0
for t, d, s in ev:
    k = sum(1 for x in active.values() if x > 0)
    if k == 1: busy1 += t - last
    elif k >= 2: busy2 += t - last
    active[s] += d
    last = t
span = t_end - t0
print(f"two-lane section {span / 1e6:.2f} ms: exactly one lane running {busy1 / 1e6:.2f} ms, both {busy2 / 1e6:.2f} ms, "
      f"idle {(span - busy1 - busy2) / 1e6:.2f} ms")
