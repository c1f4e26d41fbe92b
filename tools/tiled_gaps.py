"""Reads a rocprofv3 kernel trace of tools/tiled_virtual.py and reports, for the PCG passes of the LAST solve in the
trace (the timed one), kernel durations per band stream and the idle time between consecutive passes on that stream
(what a phase boundary costs).
usage: python tools/tiled_gaps.py <dir with *_kernel_trace.csv> [passes per stream to keep = 60]"""
import csv
import glob
import sys
from collections import defaultdict


def med(v):
    v = sorted(v)
    return v[len(v) // 2] / 1e3 if v else float("nan")


def main():
    files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
    keep = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    bys = defaultdict(list)
    for r in rows:
        n = r["Kernel_Name"]
        if "k_pcg_pass" in n:
            bys[r["Stream_Id"]].append(("A" if "k_pcg_pass_a" in n else "B", int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                                        int(r["Grid_Size_X"])))
    t_first, t_last = None, None
    for sid, ks in sorted(bys.items()):
        ks = ks[-keep:]
        durs, gaps = defaultdict(list), defaultdict(list)
        for prev, cur in zip(ks[:-1], ks[1:]):
            durs[cur[0]].append(cur[2] - cur[1])
            gaps[prev[0] + "->" + cur[0]].append(cur[1] - prev[2])
        t_first = ks[0][1] if t_first is None else min(t_first, ks[0][1])
        t_last = ks[-1][2] if t_last is None else max(t_last, ks[-1][2])
        print(f"stream {sid}: last {len(ks)} passes (grids {sorted(set(k[3] // 256 for k in ks))}); median A {med(durs['A']):.1f} us, "
              f"B {med(durs['B']):.1f} us; idle A->B {med(gaps['A->B']):.1f} us, B->A {med(gaps['B->A']):.1f} us")
    if t_first is not None:
        print(f"wall time of those passes: {(t_last - t_first) / 1e3:.1f} us -> {(t_last - t_first) / 1e3 / (keep / 2):.1f} us per PCG iteration")


if __name__ == "__main__":
    main()
