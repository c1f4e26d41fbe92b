"""Per-kernel, per-grid mean durations from a rocprofv3 kernel trace (the solver's kernels only).
usage: python tools/level_times.py <trace dir> [substring filters...]"""
import csv
import glob
import sys
from collections import defaultdict

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
want = sys.argv[2:] or ["fused", "pass_a", "pass_b", "assemble", "flow_update"]
agg = defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "octane::" not in n:
        continue
    short = n.split("octane::")[1].split("(")[0]
    if any(w in short for w in want):
        agg[(short, int(r["Grid_Size_X"]) // 256)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()):
    print(f"{k[0]:34s} grid {k[1]:5d} n={len(v):5d} mean {sum(v) / len(v):8.2f} us  total {sum(v) / 1e3:8.2f} ms")
